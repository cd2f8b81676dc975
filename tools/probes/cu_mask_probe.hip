// cu_mask_probe.hip -- can two lanes own DISJOINT halves of the chip (hipExtStreamCreateWithCUMask) so that one lane's MFMA-bound
// (power-capped) kernels run beside the other lane's HBM-bound kernels instead of after them?  Measures, on one MI355X:
//   * where the blocks of a masked stream land (XCC id / SE / CU id per block) for two mask patterns
//   * the dense MFMA loop of the 256x256 conv tile (LDS reads + 8 staging loads per wave per K-step, random fp16 data) on the whole
//     chip and on half of it: TFLOP/s and shader clock (the chip is power-capped at ~1.85-2.0 GHz under dense MFMA on 256 CUs)
//   * a streaming read+write kernel on the whole chip and on half of it: GB/s
//   * both at once on disjoint halves, and both at once on unmasked streams (what two lanes do today)
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <map>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int kStage = 64 * 1024;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void whoami(unsigned* out)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[blockIdx.x * 2] = xcc & 0xf;
        out[blockIdx.x * 2 + 1] = hw;
    }
    // keep the block alive for a moment so that blocks spread over the CUs instead of reusing one
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
}

// the product conv loop's shape: 8 waves, 256 x 256 tile, 64 MFMAs + 24 ds_read_b128 + 8 LDS-DMA loads per wave per K-step
__global__ __launch_bounds__(512, 2) void mfma_loop(float* out, int iters, const char* src, unsigned long long* clk)
{
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 2, wc = wave & 3;
    {
        unsigned s = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int i = tid; i < 2 * kStage / 2; i += 512) {
            s = s * 1664525u + 1013904223u;
            ((_Float16*)smem)[i] = (_Float16)(((int)(s >> 9) & 0xffff) / 32768.f - 1.f);
        }
    }
    __syncthreads();
    const int frow = lane & 15, fg = lane >> 4;
    const int rd0 = frow * 128 + (((0 + fg) ^ (frow & 7)) << 4), rd1 = frow * 128 + (((4 + fg) ^ (frow & 7)) << 4);
    const int w_rd = (wc * 64) * 128, p_rd = 256 * 128 + (wp * 128) * 128;
    f32x4 acc[4][8];
    for (int mi = 0; mi < 4; ++mi)
        for (int q = 0; q < 8; ++q) acc[mi][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 26, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        const char* sb = smem + (it & 1) * kStage;
        char* dst = smem + ((it + 1) & 1) * kStage;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + (j * 8 + wave) * 1024), 16,
                                                     (unsigned)(lane * 16 + wave * 1024), (unsigned)((((it * 8 + j) * 8192) + (blockIdx.x & 7) * 65536) & ((1 << 21) - 1)), 0, 0);
        f16x8 a[2][4], b[2][4];
        auto load_a = [&](int kk, f16x8 (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) d[mi] = *(const f16x8*)(sb + w_rd + mi * 2048 + (kk ? rd1 : rd0));
        };
        auto load_b = [&](int kk, int h, f16x8 (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) d[q] = *(const f16x8*)(sb + p_rd + (h * 4 + q) * 2048 + (kk ? rd1 : rd0));
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float total = 0.f;
    for (int mi = 0; mi < 4; ++mi)
        for (int q = 0; q < 8; ++q) total += acc[mi][q][0] + acc[mi][q][3];
    out[blockIdx.x * 512 + tid] = total;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
}

// HBM-bound read + write stream: y = 2 x over n float4, grid-stride, 8 loads in flight per lane
__global__ __launch_bounds__(256) void stream_rw(const float4* x, float4* y, size_t n, int reps)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (int rep = 0; rep < reps; ++rep)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += 4 * stride) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = i + k * stride < n ? x[i + k * stride] : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + k * stride < n) y[i + k * stride] = make_float4(v[k].x * 2.f, v[k].y * 2.f, v[k].z * 2.f, v[k].w * 2.f);
        }
}

__global__ void fill_random(_Float16* p, size_t n)
{
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 99u;
    for (size_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        s = s * 1664525u + 1013904223u;
        p[i] = (_Float16)(((int)(s >> 9) & 0xffff) / 32768.f - 1.f);
    }
}

static void where(const char* name, hipStream_t s, unsigned* d_who, int blocks)
{
    CHK(hipMemsetAsync(d_who, 0xff, blocks * 8, s));
    hipLaunchKernelGGL(whoami, dim3(blocks), dim3(64), 0, s, d_who);
    CHK(hipStreamSynchronize(s));
    std::vector<unsigned> h(blocks * 2);
    CHK(hipMemcpy(h.data(), d_who, blocks * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, int> per_xcc;
    std::map<unsigned long long, int> cus;
    for (int b = 0; b < blocks; ++b) {
        per_xcc[h[b * 2]]++;
        const unsigned hw = h[b * 2 + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus[((unsigned long long)h[b * 2] << 16) | (se << 8) | (sh << 4) | cu]++;
    }
    printf("%-40s blocks per XCC:", name);
    for (auto& kv : per_xcc) printf(" x%u:%d", kv.first, kv.second);
    printf("   distinct (xcc, se, sh, cu): %zu   first blocks' xcc:", cus.size());
    for (int b = 0; b < 16 && b < blocks; ++b) printf(" %u", h[b * 2]);
    printf("\n");
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device: %s, %d CUs\n", prop.name, ncu);
    const int words = (ncu + 31) / 32;
    std::vector<uint32_t> lowhalf(words, 0), highhalf(words, 0), even(words, 0), odd(words, 0);
    for (int i = 0; i < ncu; ++i) {
        (i < ncu / 2 ? lowhalf : highhalf)[i / 32] |= 1u << (i % 32);
        ((i & 1) ? odd : even)[i / 32] |= 1u << (i % 32);
    }
    hipStream_t s_low, s_high, s_even, s_odd, s_a, s_b;
    hipError_t e = hipExtStreamCreateWithCUMask(&s_low, words, lowhalf.data());
    if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(e)); return 2; }
    CHK(hipExtStreamCreateWithCUMask(&s_high, words, highhalf.data()));
    CHK(hipExtStreamCreateWithCUMask(&s_even, words, even.data()));
    CHK(hipExtStreamCreateWithCUMask(&s_odd, words, odd.data()));
    CHK(hipStreamCreateWithFlags(&s_a, hipStreamNonBlocking));
    CHK(hipStreamCreateWithFlags(&s_b, hipStreamNonBlocking));
    unsigned* d_who;
    CHK(hipMalloc((void**)&d_who, 4096 * 8));
    where("unmasked stream, 1024 blocks", s_a, d_who, 1024);
    where("mask bits [0, n/2), 1024 blocks", s_low, d_who, 1024);
    where("mask bits [n/2, n), 1024 blocks", s_high, d_who, 1024);
    where("mask even bits, 1024 blocks", s_even, d_who, 1024);
    where("mask odd bits, 1024 blocks", s_odd, d_who, 1024);

    char* d_src; float* d_out; unsigned long long* d_clk; unsigned long long* d_clk2;
    CHK(hipMalloc((void**)&d_src, 1 << 26));
    hipLaunchKernelGGL(fill_random, dim3(1024), dim3(256), 0, 0, (_Float16*)d_src, (size_t)(1 << 25));
    CHK(hipMalloc((void**)&d_out, 256 * 512 * sizeof(float)));
    CHK(hipMalloc((void**)&d_clk, 16)); CHK(hipMalloc((void**)&d_clk2, 16));
    const size_t n4 = (size_t)1 << 26;                       // 64 M float4 = 1 GiB read + 1 GiB written per pass
    float4 *d_x, *d_y;
    CHK(hipMalloc((void**)&d_x, n4 * 16)); CHK(hipMalloc((void**)&d_y, n4 * 16));
    CHK(hipMemset(d_x, 0x11, n4 * 16));
    CHK(hipFuncSetAttribute((const void*)mfma_loop, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStage));
    CHK(hipDeviceSynchronize());

    auto time_mfma = [&](const char* name, hipStream_t s, int blocks, int iters) {
        hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(512), 2 * kStage, s, d_out, iters, d_src, d_clk);
        CHK(hipStreamSynchronize(s));
        CHK(hipEventRecord(a, s));
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(512), 2 * kStage, s, d_out, iters, d_src, d_clk);
        CHK(hipEventRecord(b, s));
        CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        unsigned long long h[2]; CHK(hipMemcpy(h, d_clk, 16, hipMemcpyDeviceToHost));
        const double fl = 2.0 * 256 * 256 * 64 * (double)iters * blocks;
        printf("%-58s %8.3f ms  %7.1f TFLOP/s  (%5.2f per block)  clock %.0f MHz\n", name, ms, fl / (ms * 1e-3) / 1e12, fl / (ms * 1e-3) / 1e12 / blocks, (double)h[0] / h[1] * 100.0);
    };
    auto time_stream = [&](const char* name, hipStream_t s, int blocks, int reps) {
        hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
        hipLaunchKernelGGL(stream_rw, dim3(blocks), dim3(256), 0, s, d_x, d_y, n4, 1);
        CHK(hipStreamSynchronize(s));
        CHK(hipEventRecord(a, s));
        hipLaunchKernelGGL(stream_rw, dim3(blocks), dim3(256), 0, s, d_x, d_y, n4, reps);
        CHK(hipEventRecord(b, s));
        CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        printf("%-58s %8.3f ms  %7.1f GB/s (read + write)\n", name, ms, 2.0 * n4 * 16 * reps / (ms * 1e-3) / 1e9);
    };
    const int iters = 3000;
    time_mfma("MFMA loop, whole chip (256 blocks)", s_a, ncu, iters);
    time_mfma("MFMA loop, mask [0, n/2) (128 blocks)", s_low, ncu / 2, iters);
    time_mfma("MFMA loop, mask even bits (128 blocks)", s_even, ncu / 2, iters);
    time_stream("stream r+w, whole chip (2048 blocks)", s_a, 8 * ncu, 2);
    time_stream("stream r+w, mask [n/2, n) (1024 blocks)", s_high, 4 * ncu, 2);
    time_stream("stream r+w, mask odd bits (1024 blocks)", s_odd, 4 * ncu, 2);
    time_stream("stream r+w, mask odd bits (2048 blocks)", s_odd, 8 * ncu, 2);

    // both at once
    auto both = [&](const char* name, hipStream_t sm, int mblocks, hipStream_t ss, int sblocks) {
        hipEvent_t a0, a1, b0, b1;
        CHK(hipEventCreate(&a0)); CHK(hipEventCreate(&a1)); CHK(hipEventCreate(&b0)); CHK(hipEventCreate(&b1));
        CHK(hipDeviceSynchronize());
        const int reps = 6;
        CHK(hipEventRecord(a0, sm));
        hipLaunchKernelGGL(mfma_loop, dim3(mblocks), dim3(512), 2 * kStage, sm, d_out, iters, d_src, d_clk);
        CHK(hipEventRecord(a1, sm));
        CHK(hipEventRecord(b0, ss));
        hipLaunchKernelGGL(stream_rw, dim3(sblocks), dim3(256), 0, ss, d_x, d_y, n4, reps);
        CHK(hipEventRecord(b1, ss));
        CHK(hipDeviceSynchronize());
        float ma, mb; CHK(hipEventElapsedTime(&ma, a0, a1)); CHK(hipEventElapsedTime(&mb, b0, b1));
        unsigned long long h[2]; CHK(hipMemcpy(h, d_clk, 16, hipMemcpyDeviceToHost));
        const double fl = 2.0 * 256 * 256 * 64 * (double)iters * mblocks;
        printf("%-44s MFMA %8.3f ms %7.1f TFLOP/s clock %.0f MHz | stream %8.3f ms %7.1f GB/s\n", name, ma, fl / (ma * 1e-3) / 1e12,
               (double)h[0] / h[1] * 100.0, mb, 2.0 * n4 * 16 * reps / (mb * 1e-3) / 1e9);
    };
    both("halves [0,n/2) MFMA | [n/2,n) stream", s_low, ncu / 2, s_high, 4 * ncu);
    both("even MFMA | odd stream", s_even, ncu / 2, s_odd, 4 * ncu);
    both("unmasked: MFMA 256 blocks | stream 2048", s_a, ncu, s_b, 8 * ncu);
    both("unmasked: MFMA 128 blocks | stream 1024", s_a, ncu / 2, s_b, 4 * ncu);
    return 0;
}
