// mfma_loop_probe.hip -- what the LDS -> MFMA inner loop of conv_igemm_mfma can reach WITHOUT any staging loads:
// 8 waves (2 x 4), 256 x 256 tile, wave tile 128 px x 64 ch, K-step 64, two LDS stages alternated, one barrier per K-step.
// Variants: MFMA shape 16x16x32 (the product kernel's) vs 32x32x16; fragments requested phase by phase (as the product
// kernel does) vs all of a K-step first; with / without the barrier.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_loop_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kStage = 64 * 1024;      // W 256 rows x 128 B | P 256 rows x 128 B

// VAR bits: 1 = 32x32x16 MFMA, 2 = all fragments first, 4 = no barrier, 8 = setprio around the MFMA cluster,
// 16 = no LDS reads (fragments loaded once: the bare MFMA issue rate at this occupancy), 32 = stage the NEXT K-step with
// 8 buffer_load ... lds per wave (64 KB per block per K-step, from an L2-resident buffer), drained before the barrier
// NV = integer VALU instructions issued at the top of every K-step (stand-in for the gather's address arithmetic:
// the product kernel spends ~14 per staged pixel row, 8-10 rows per wave)
template <int VAR, int NV = 0>
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters, const char* src, unsigned long long* clk)
{
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 2, wc = wave & 3;
    // fill LDS with pseudo-random fp16 in [-1, 1)
    {
        unsigned s = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int i = tid; i < 2 * kStage / 2; i += 512) {
            s = s * 1664525u + 1013904223u;
            ((_Float16*)smem)[i] = (_Float16)(((int)(s >> 9) & 0xffff) / 32768.f - 1.f);
        }
    }
    __syncthreads();
    constexpr bool M32 = VAR & 1, ALL = VAR & 2, NOBAR = VAR & 4, PRIO = VAR & 8, REG = VAR & 16, LD = VAR & 32, X3L = VAR & 64;
    auto stage_next = [&](int it) __attribute__((always_inline)) {
        if constexpr (LD) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 26, 0x00020000);
            char* dst = smem + ((it + 1) & 1) * kStage;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + (j * 8 + wave) * 1024), 16,
                                                         (unsigned)(lane * 16 + wave * 1024), (unsigned)((((it * 8 + j) * 8192) + (blockIdx.x & 7) * 65536) & ((1 << 21) - 1)), 0, 0);
        }
    };
    auto drain = [&]() __attribute__((always_inline)) {
        if constexpr (LD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    float total = 0.f;
    int av[8];
    for (int j = 0; j < 8; ++j) av[j] = tid * (j + 3);
    auto addr_work = [&](int it) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NV; ++k) av[k & 7] = __builtin_amdgcn_sad_u8(av[k & 7], it + k, av[(k + 3) & 7]);      // one VALU op each, 8 independent chains
    };
    if constexpr (!M32) {
        const int frow = lane & 15, fg = lane >> 4;
        const int rd0 = frow * 128 + (((0 + fg) ^ (frow & 7)) << 4), rd1 = frow * 128 + (((4 + fg) ^ (frow & 7)) << 4);
        const int w_rd = (wc * 64) * 128, p_rd = 256 * 128 + (wp * 128) * 128;
        f32x4 acc[4][8];
        f16x8 ra[2][4], rb[2][8];
        for (int mi = 0; mi < 4; ++mi)
            for (int q = 0; q < 8; ++q) acc[mi][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            const char* sb = smem + (it & 1) * kStage;
            addr_work(it);
            stage_next(it);
            if constexpr (REG) {
                static_assert(!REG || ALL, "register-only variant uses the all-fragments form");
                if (it == 0) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) ra[kk][mi] = *(const f16x8*)(sb + w_rd + mi * 2048 + (kk ? rd1 : rd0));
#pragma unroll
                        for (int q = 0; q < 8; ++q) rb[kk][q] = *(const f16x8*)(sb + p_rd + q * 2048 + (kk ? rd1 : rd0));
                    }
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc[mi][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[kk][mi], rb[kk][q], acc[mi][q], 0, 0, 0);
            } else if constexpr (X3L) {
                // the SPLIT mode's K-step (round 5): the stage is 32 channels as hi (slots 0-3: rd0) and lo (slots 4-7: rd1) granules, three
                // MFMAs per fragment pair (lo*hi, hi*lo, hi*hi), pixel fragments in four phases of two blocks requested a phase ahead --
                // conv_igemm_mfma<256, 256, 2, 4, 2, .., X3>'s loop; 96 MFMAs per wave for the same LDS bytes and staging loads
                f16x8 ah[4], al[4], bh[2][2], bl[2][2];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) al[mi] = *(const f16x8*)(sb + w_rd + mi * 2048 + rd1);
#pragma unroll
                for (int q = 0; q < 2; ++q) bh[0][q] = *(const f16x8*)(sb + p_rd + q * 2048 + rd0);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) ah[mi] = *(const f16x8*)(sb + w_rd + mi * 2048 + rd0);
#pragma unroll
                for (int q = 0; q < 2; ++q) bl[0][q] = *(const f16x8*)(sb + p_rd + q * 2048 + rd1);
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) {
                    if (ph + 1 < 4) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            bh[(ph + 1) & 1][q] = *(const f16x8*)(sb + p_rd + ((ph + 1) * 2 + q) * 2048 + rd0);
                            bl[(ph + 1) & 1][q] = *(const f16x8*)(sb + p_rd + ((ph + 1) * 2 + q) * 2048 + rd1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int q = 0; q < 2; ++q) acc[mi][ph * 2 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mi], bh[ph & 1][q], acc[mi][ph * 2 + q], 0, 0, 0);
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int q = 0; q < 2; ++q) acc[mi][ph * 2 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bl[ph & 1][q], acc[mi][ph * 2 + q], 0, 0, 0);
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int q = 0; q < 2; ++q) acc[mi][ph * 2 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], bh[ph & 1][q], acc[mi][ph * 2 + q], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (ALL) {
                f16x8 a[2][4], b[2][8];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) a[kk][mi] = *(const f16x8*)(sb + w_rd + mi * 2048 + (kk ? rd1 : rd0));
#pragma unroll
                    for (int q = 0; q < 8; ++q) b[kk][q] = *(const f16x8*)(sb + p_rd + q * 2048 + (kk ? rd1 : rd0));
                }
                if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc[mi][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk][mi], b[kk][q], acc[mi][q], 0, 0, 0);
                if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            } else {
                // the product kernel's phases: (kk, half of the pixel blocks), next phase's fragments requested before this phase's MFMAs
                f16x8 a[2][4], b[2][4];
                auto load_a = [&](int kk, f16x8 (&dst)[4]) __attribute__((always_inline)) {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) dst[mi] = *(const f16x8*)(sb + w_rd + mi * 2048 + (kk ? rd1 : rd0));
                };
                auto load_b = [&](int kk, int h, f16x8 (&dst)[4]) __attribute__((always_inline)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) dst[q] = *(const f16x8*)(sb + p_rd + (h * 4 + q) * 2048 + (kk ? rd1 : rd0));
                };
                load_a(0, a[0]);
                load_b(0, 0, b[0]);
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) {
                    const int kk = ph / 2, h = ph % 2;
                    if (ph + 1 < 4) {
                        const int nkk = (ph + 1) / 2, nh = (ph + 1) % 2;
                        if (nh == 0) load_a(nkk, a[nkk & 1]);
                        load_b(nkk, nh, b[(ph + 1) & 1]);
                    }
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[mi][h * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk & 1][mi], b[ph & 1][q], acc[mi][h * 4 + q], 0, 0, 0);
                }
            }
            drain();
            if constexpr (!NOBAR) __builtin_amdgcn_s_barrier();
        }
        for (int mi = 0; mi < 4; ++mi)
            for (int q = 0; q < 8; ++q) total += acc[mi][q][0] + acc[mi][q][1] + acc[mi][q][2] + acc[mi][q][3];
    } else {
        // 32x32x16: A fragment = row (lane & 31), k-chunk of 8 = 2j + (lane >> 5); swizzle granule ^ ((row >> 1) & 7) (conflict-free for
        // the 16-lane groups of ds_read_b128 when 32 consecutive rows are read)
        const int frow = lane & 31, fh = lane >> 5;
        int rd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rd[j] = frow * 128 + (((2 * j + fh) ^ ((frow >> 1) & 7)) << 4);
        const int w_rd = (wc * 64) * 128, p_rd = 256 * 128 + (wp * 128) * 128;
        f32x16 acc[2][4];
        for (int mi = 0; mi < 2; ++mi)
            for (int q = 0; q < 4; ++q)
                for (int r = 0; r < 16; ++r) acc[mi][q][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            const char* sb = smem + (it & 1) * kStage;
            if constexpr (ALL) {
                f16x8 a[4][2], b[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) a[j][mi] = *(const f16x8*)(sb + w_rd + mi * 4096 + rd[j]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[j][q] = *(const f16x8*)(sb + p_rd + q * 4096 + rd[j]);
                }
                if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[mi][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j][mi], b[j][q], acc[mi][q], 0, 0, 0);
                if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            } else {
                f16x8 a[2][2], b[2][4];
                auto load = [&](int j, f16x8 (&da)[2], f16x8 (&db)[4]) __attribute__((always_inline)) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) da[mi] = *(const f16x8*)(sb + w_rd + mi * 4096 + rd[j]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) db[q] = *(const f16x8*)(sb + p_rd + q * 4096 + rd[j]);
                };
                load(0, a[0], b[0]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j + 1 < 4) load(j + 1, a[(j + 1) & 1], b[(j + 1) & 1]);
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[mi][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j & 1][mi], b[j & 1][q], acc[mi][q], 0, 0, 0);
                }
            }
            if constexpr (!NOBAR) __builtin_amdgcn_s_barrier();
        }
        for (int mi = 0; mi < 2; ++mi)
            for (int q = 0; q < 4; ++q)
                for (int r = 0; r < 16; ++r) total += acc[mi][q][r];
    }
    for (int j = 0; j < 8; ++j) total += (float)(av[j] & 1);
    if (total == 12345.678f) out[blockIdx.x * 512 + tid] = total;      // keep the accumulators alive
    if (tid == 0 && blockIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
}

__global__ void fill_random(_Float16* p, size_t n)
{
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 99u;
    for (size_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        s = s * 1664525u + 1013904223u;
        p[i] = (_Float16)(((int)(s >> 9) & 0xffff) / 32768.f - 1.f);
    }
}

template <int VAR, int NV = 0>
static void run(const char* name, float* d_out, int iters)
{
    static char* d_src = nullptr;
    static unsigned long long* d_clk = nullptr;
    if (!d_src) { hipMalloc((void**)&d_src, 1 << 26); hipLaunchKernelGGL(fill_random, dim3(1024), dim3(256), 0, 0, (_Float16*)d_src, (size_t)(1 << 25)); hipMalloc((void**)&d_clk, 16); }
    hipFuncSetAttribute((const void*)probe<VAR, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStage);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<VAR, NV>), dim3(256), dim3(512), 2 * kStage, 0, d_out, iters, d_src, d_clk);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((probe<VAR, NV>), dim3(256), dim3(512), 2 * kStage, 0, d_out, iters, d_src, d_clk);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double flops = ((VAR & 64) ? 3.0 * 2.0 * 256 * 256 * 32 : 2.0 * 256 * 256 * 64) * (double)iters * 256;      // (split loop: ISSUED MFMA work)
    unsigned long long h[2];
    hipMemcpy(h, d_clk, 16, hipMemcpyDeviceToHost);
    // clock64 = shader-clock counter, wall_clock64 = 100 MHz constant counter
    printf("%-58s %8.3f ms  %7.1f TFLOP/s   shader clock %.0f MHz\n", name, best, flops / (best * 1e-3) / 1e12, (double)h[0] / (double)h[1] * 100.0);
}

int main()
{
    float* d_out = nullptr;
    hipMalloc((void**)&d_out, 256 * 512 * sizeof(float));
    const int iters = 4000;
    run<0>("16x16x32, phased reads, barrier per K-step (product)", d_out, iters);
    run<2>("16x16x32, all fragments first, barrier", d_out, iters);
    run<2 | 8>("16x16x32, all fragments first, barrier, setprio", d_out, iters);
    run<4>("16x16x32, phased reads, no barrier", d_out, iters);
    run<2 | 4>("16x16x32, all fragments first, no barrier", d_out, iters);
    run<2 | 16>("16x16x32, NO LDS reads (bare MFMA issue), barrier", d_out, iters);
    run<2 | 16 | 4>("16x16x32, NO LDS reads, no barrier", d_out, iters);
    run<32>("16x16x32 product loop + 8 staging loads per wave", d_out, iters);
    run<2 | 16 | 32>("16x16x32, NO LDS reads + 8 staging loads per wave", d_out, iters);
    run<4 | 32>("16x16x32 product loop + 8 staging loads, no barrier (racy)", d_out, iters);
    run<0, 32>("16x16x32 product loop + 32 VALU per K-step", d_out, iters);
    run<0, 64>("16x16x32 product loop + 64 VALU per K-step", d_out, iters);
    run<0, 128>("16x16x32 product loop + 128 VALU per K-step", d_out, iters);
    run<0, 192>("16x16x32 product loop + 192 VALU per K-step", d_out, iters);
    run<64>("SPLIT loop (3 MFMAs per pair), LDS reads, barrier, no staging", d_out, iters);
    run<64 | 32>("SPLIT loop + 8 staging loads per wave (the decoder's K-step)", d_out, iters);
    run<64 | 32 | 4>("SPLIT loop + 8 staging loads, no barrier (racy)", d_out, iters);
    run<64, 32>("SPLIT loop + 32 VALU per K-step, no staging", d_out, iters);
    run<1>("32x32x16, reads one k-chunk ahead, barrier", d_out, iters);
    run<1 | 2>("32x32x16, all fragments first, barrier", d_out, iters);
    run<1 | 2 | 8>("32x32x16, all fragments first, barrier, setprio", d_out, iters);
    run<1 | 4>("32x32x16, reads one k-chunk ahead, no barrier", d_out, iters);
    run<1 | 2 | 4>("32x32x16, all fragments first, no barrier", d_out, iters);
    hipFree(d_out);
    return 0;
}
