// dec_halo_f16.hip -- the decoder conv at 224 x 224 ("dec4": 3x3 conv over [up2(src0: 128 ch @ 112 x 112), skip: 64 ch @ 224 x 224] -> 64 ch,
// BN / ReLU) with its source halos resident in LDS, plain fp16 mode (kF16), round 4.  The split-mode twin is dec_halo_x3.hip; this
// file states what differs.
//
//   * a stored pixel is half as wide (src0 256 B, skip 128 B), a K-step is 64 channels = two MFMA k-halves (kk = 0, 1): 2 groups x 4
//     taps of src0 + 9 taps of the skip = 17 K-steps, 16 MFMAs per wave each, in the order of the classes' K-step records (the
//     generic kernel's order; every accumulator takes kk = 0 before kk = 1): bit-identical outputs (tests/test_gpu_parity.py)
//   * both halos fit TWICE (2 x 25 KB + 2 x 45 KB): the whole next tile is fetched (LDS-DMA, 10 instructions per wave) at the top of
//     a tile into the other buffer -- one barrier at the top (every wave has left the buffer about to be overwritten) and one in the
//     middle (every wave has waited for its own DMA: the next tile's halo is complete before the first fragment of it is read)
//   * 17 is odd and the fragment / weight registers alternate between two sets: the tile loop is unrolled over TWO tiles (34
//     compile-time steps, buffers 0 and 1); a block with an odd tile count computes its last tile twice (same bytes stored twice)
//   * skip rows are 128 B, half a bank row: a 16-lane ds_read_b128 group (guide: lanes {0-3, 12-15, 20-27} ...: all 16 fragment
//     pixels, fg = 0 for eight of them and 1 for the others) would find only 8 slots.  The skip halo is therefore stored as 256-byte
//     PAIR rows -- halo rows hy and hy + 2 side by side, the two rows a lane group reads from (class-grid rows i0 = 0, 1) -- with
//     granule G of a pixel at slot (G + (hx & ~1)) & 7 of its half: 16 distinct slots per group.  src0 rows are 256 B with the
//     split kernel's rotation (G + 2 hx) & 15
//   * every vector-memory operation of the loop is issued from inline asm and counted by hand, as in the split kernel
#include "internal.h"

namespace sbbseg {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
#define LDS_AS __attribute__((address_space(3)))

constexpr int kS0Instr = 25, kSkInstr = 45;             // wave-instructions (1 KB) per halo: 100 pixels x 256 B; 10 pair rows x 18 x 256 B
constexpr int kS0Bytes = kS0Instr * 1024;
constexpr int kSkBytes = kSkInstr * 1024;
constexpr int kDecHaloF16LdsBytes = 2 * kS0Bytes + 2 * kSkBytes;      // 143 360
constexpr int kSteps = 17;
constexpr int kS0PerWave = 4, kSkPerWave = 6;           // DMA instructions per wave per tile (32 >= 25, 48 >= 45: the surplus repeats)
constexpr int kDma = kS0PerWave + kSkPerWave;
constexpr int kStores = 4;

template <int N> struct IC { static constexpr int value = N; };
template <int B, int E, class F> __device__ __attribute__((always_inline)) inline void static_for(F&& f)
{
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ inline f4_t mma(h8_t a, h8_t b, f4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// kernels.hip's pack_f16x2: saturate, round to nearest even
__device__ inline uint32_t pack_h2(float a, float b)
{
    a = fminf(fmaxf(a, -65504.f), 65504.f);
    b = fminf(fmaxf(b, -65504.f), 65504.f);
    h2_t v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
}

__device__ __attribute__((always_inline)) inline void wload4(u4_t& a, u4_t& b, u4_t& c, u4_t& d, uint32_t voff, u4_t rsrc)
{
    asm volatile("buffer_load_dwordx4 %0, %4, %5, 0 offen\n\t"
                 "buffer_load_dwordx4 %1, %4, %5, 0 offen offset:1024\n\t"
                 "buffer_load_dwordx4 %2, %4, %5, 0 offen offset:2048\n\t"
                 "buffer_load_dwordx4 %3, %4, %5, 0 offen offset:3072"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(voff), "s"(rsrc) : "memory");
}
template <int N> __device__ __attribute__((always_inline)) inline void wait_w(u4_t& a, u4_t& b, u4_t& c, u4_t& d)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
__device__ __attribute__((always_inline)) inline void glds16_hidden(const void* gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

}  // namespace

__global__ __launch_bounds__(512, 2) void dec_halo_f16(const DecHaloParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(LDS_AS char*)smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cls = wave >> 1, mh = wave & 1;
    const int py = cls >> 1, px = cls & 1;
    const int frow = lane & 15, fg = lane >> 4;

    const int H = 2 * p.PH, W = 2 * p.PW;
    const int tiles_x = W / 16, tiles_y = H / 16;
    const int tiles_per_patch = tiles_x * tiles_y;
    // owned-region launch (DecHaloParams::ttab, region.h): the tiles are the table's entries, see dec_halo_x3
    const int n_tiles = p.ttab ? p.n_tab : p.n * tiles_per_patch;
    const __attribute__((address_space(4))) uint32_t* ttab = (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)p.ttab;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, GX = gridDim.x >> 3;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int xcd_lo = xcd * per_xcd, xcd_hi = min(n_tiles, xcd_lo + per_xcd);
    const int my_tiles = xcd_lo + slot < xcd_hi ? (xcd_hi - xcd_lo - slot + GX - 1) / GX : 0;
    if (my_tiles <= 0) return;
    auto tile_at = [&](int it) __attribute__((always_inline)) -> int { return xcd_lo + slot + (it < my_tiles ? it : my_tiles - 1) * GX; };

    // ---- halo DMA.  src0 instruction k: halo pixels 4 k .. 4 k + 3, lane l -> slot l & 15 of pixel 4 k + (l >> 4).  skip instruction k:
    // pair rows 4 k .. 4 k + 3 (row r = pair * 18 + hx), lane l -> half (l >> 3) & 1, slot l & 7 of row 4 k + (l >> 4); pair P holds halo
    // rows hy = 4 (P >> 1) + (P & 1) [half 0] and hy + 2 [half 1]
    int s0_e[kS0PerWave], sk_e[kSkPerWave];                    // hy | hx << 8 | source granule << 16 | instruction << 24
#pragma unroll
    for (int m = 0; m < kS0PerWave; ++m) {
        int k = wave + 8 * m;
        k = k < kS0Instr ? k : k - kS0Instr;
        const int hp = 4 * k + (lane >> 4);
        const int hy = hp / 10, hx = hp - hy * 10;
        s0_e[m] = hy | (hx << 8) | ((((lane & 15) - 2 * hx) & 15) << 16) | (k << 24);
    }
#pragma unroll
    for (int m = 0; m < kSkPerWave; ++m) {
        int k = wave + 8 * m;
        k = k < kSkInstr ? k : k - kSkInstr;
        const int r = 4 * k + (lane >> 4);
        const int P = r / 18, hx = r - P * 18;
        const int hy = 4 * (P >> 1) + (P & 1) + 2 * ((lane >> 3) & 1);        // 18, 19 (pair 8, 9, half 1): no such halo row -> zeros
        sk_e[m] = hy | (hx << 8) | ((((lane & 7) - (hx & ~1)) & 7) << 16) | (k << 24);
    }
    auto tile_coords = [&](int tile, int& n, int& y0, int& x0) __attribute__((always_inline)) {      // patch, origin of the 16 x 16 outputs
        if (ttab) {
            const uint32_t code = ttab[tile];
            n = (int)(code >> 22); y0 = (int)((code >> 11) & 2047u) * 2; x0 = (int)(code & 2047u) * 2;
        } else {
            n = tile / tiles_per_patch;
            const int rem = tile - n * tiles_per_patch;
            const int ty = rem / tiles_x;
            y0 = ty * 16;
            x0 = (rem - ty * tiles_x) * 16;
        }
    };
    auto issue_halos = [&](int tile, int buf) __attribute__((always_inline)) {            // kDma loads per wave
        int n, y0, x0;
        tile_coords(tile, n, y0, x0);
#pragma unroll
        for (int m = 0; m < kS0PerWave; ++m) {
            const int e = s0_e[m];
            const int Y = (y0 >> 1) - 1 + (e & 255), X = (x0 >> 1) - 1 + ((e >> 8) & 255);
            const bool ok = (unsigned)Y < (unsigned)p.PH && (unsigned)X < (unsigned)p.PW;
            const uint32_t off = ok ? (uint32_t)((n * p.PH + Y) * p.PW + X) * 256u + (uint32_t)(((e >> 16) & 15) * 16 + kZeroHeaderBytes) : 0u;
            const uint32_t dst = lds0 + (uint32_t)(buf * kS0Bytes) + (uint32_t)__builtin_amdgcn_readfirstlane(e >> 24) * 1024u;
            glds16_hidden(p.src0 + off, dst);
        }
#pragma unroll
        for (int m = 0; m < kSkPerWave; ++m) {
            const int e = sk_e[m];
            const int hy = e & 255;
            const int Y = y0 - 1 + hy, X = x0 - 1 + ((e >> 8) & 255);
            const bool ok = hy < 18 && (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W;
            const uint32_t off = ok ? (uint32_t)((n * H + Y) * W + X) * 128u + (uint32_t)(((e >> 16) & 15) * 16 + kZeroHeaderBytes) : 0u;
            const uint32_t dst = lds0 + (uint32_t)(2 * kS0Bytes + buf * kSkBytes) + (uint32_t)__builtin_amdgcn_readfirstlane(e >> 24) * 1024u;
            glds16_hidden(p.skip + off, dst);
        }
    };

    // ---- fragment addressing.  Lane (frow, fg) of pixel block ni holds class-grid pixel (i, j) = (2 ni + (frow >> 3), frow & 7).
    //   src0 step (group g, tap (dy, dx)): halo pixel (i + dy + 1, j + dx + 1), granule 8 g + 4 kk + fg, slot (G + 2 hx) & 15; ni + 1 = 20 rows on
    //   skip step (tap (dy, dx)):          halo pixel (hy, hx) = (2 i + dy + 1, 2 j + dx + 1) = (4 ni + e, ...), e = 2 i0 + dy + 1 in 0..5:
    //                                      pair 2 (ni + (e >> 2)) + (e & 1), half (e >> 1) & 1, slot (4 kk + fg + (hx & ~1)) & 7; ni + 1 = 36 rows on
    const int i0 = frow >> 3, j0 = frow & 7;
    int a0 = (i0 + 1) * 10 + j0 + 1, r0 = fg + 2 * (j0 + 1);
    int e1 = 2 * i0 + 1, x1 = 2 * j0 + 1;
    int taps[13];
    {
        const __attribute__((address_space(4))) int* tp = (const __attribute__((address_space(4))) int*)(uintptr_t)(p.taps + cls * 16);
#pragma unroll
        for (int k = 0; k < 13; ++k) taps[k] = tp[k];
    }
    auto frag_addr = [&](int t, int buf, const char*& k0, const char*& k1) __attribute__((always_inline)) {
        if (t < 8) {
            const int g = t >> 2, tp = taps[t & 3];
            const int dy = (tp << 24) >> 24, dx = (tp << 16) >> 24;
            const int hp = a0 + dy * 10 + dx;
            const int sh = (r0 + 2 * dx + 8 * g) & 15;
            const char* base = smem + buf * kS0Bytes + hp * 256;
            k0 = base + (sh << 4);
            k1 = base + (((sh + 4) & 15) << 4);
        } else {
            const int tp = taps[4 + (t - 8)];
            const int dy = (tp << 24) >> 24, dx = (tp << 16) >> 24;
            const int e = e1 + dy, hx = x1 + dx;
            const int row = (2 * (e >> 2) + (e & 1)) * 18 + hx;
            const int sh = (fg + (hx & ~1)) & 7;
            const char* base = smem + 2 * kS0Bytes + buf * kSkBytes + row * 256 + ((e >> 1) & 1) * 128;
            k0 = base + (sh << 4);
            k1 = base + (((sh + 4) & 7) << 4);
        }
    };
    // weights: wfrag = [class][K-step][row block mi 0..3][kk][64 lanes x 16 B]; this wave's row blocks are 2 mh, 2 mh + 1
    u4_t wrsrc;
    {
        const uint64_t base = (uint64_t)(uintptr_t)p.wfrag + (uint64_t)(cls * kSteps * 8 + 2 * mh * 2) * 1024u;
        wrsrc[0] = __builtin_amdgcn_readfirstlane((uint32_t)base);
        wrsrc[1] = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32) & 0xffffu);
        wrsrc[2] = (uint32_t)(kSteps * 8 * 1024);
        wrsrc[3] = 0x00020000u;
    }
    uint32_t wlane = (uint32_t)lane * 16u;

    const int c0 = mh * 32 + fg * 8;
    float ksc[8], ksh[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { ksc[q] = p.scale[c0 + q]; ksh[q] = p.shift[c0 + q]; }
    // (a use the compiler can see: it waits for these loads HERE.  Left to the first epilogue, its wait -- vmcnt(0), it cannot count the
    // asm loads in between -- would drain the hand-counted queue inside the tile loop)
#pragma unroll
    for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(ksc[q]), "+v"(ksh[q]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- prologue: the first tile's halos (buffer 0)
    issue_halos(tile_at(0), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    u4_t w[2][4];                                              // weight ring: step s of the two-tile loop in set s & 1; [m * 2 + kk]
    h8_t b0[2][4], b1[2][4];                                   // pixel fragments (kk = 0, 1) of step s in set s & 1
    auto load_b = [&](int t, int buf, h8_t (&d0)[4], h8_t (&d1)[4]) __attribute__((always_inline)) {
        const char *k0, *k1;
        frag_addr(t, buf, k0, k1);
        const int nstride = t < 8 ? 20 * 256 : 36 * 256;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            d0[ni] = *(const h8_t*)(k0 + ni * nstride);
            d1[ni] = *(const h8_t*)(k1 + ni * nstride);
        }
    };
    wload4(w[0][0], w[0][1], w[0][2], w[0][3], wlane + 0 * 8192u, wrsrc);
    wload4(w[1][0], w[1][1], w[1][2], w[1][3], wlane + 1 * 8192u, wrsrc);
    load_b(0, 0, b0[0], b1[0]);

    for (int it = 0; it < my_tiles; it += 2) {
        f4_t acc[2][4];
        static_for<0, 2 * kSteps>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            constexpr int half = s / kSteps, t = s % kSteps;            // tile it + half lives in buffer `half`
            if constexpr (t == 0) {
                // top of a tile: every wave is through with the other buffer (its last fragments were taken before the previous tile's
                // last MFMAs) -> fetch the tile after this one into it
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                // (the fragment addresses of the 17 steps are tile-invariant: keep the compiler from carrying them -- 2 x 17 registers and as
                // many scalars -- from tile to tile; they are recomputed per step, ~10 VALU / SALU ops beside 16 MFMAs)
                asm volatile("" : "+v"(a0), "+v"(r0), "+v"(e1), "+v"(x1), "+v"(wlane));
#pragma unroll
                for (int k = 0; k < 13; ++k) asm volatile("" : "+s"(taps[k]));
                issue_halos(tile_at(it + half + 1), half ^ 1);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[m][ni] = (f4_t){0.f, 0.f, 0.f, 0.f};
            }
            // pixel fragments of the next step (step 0 of the next tile behind the last: the other buffer, complete since the middle barrier)
            load_b((t + 1) % kSteps, t + 1 < kSteps ? half : half ^ 1, b0[(s + 1) & 1], b1[(s + 1) & 1]);
            // This step's weights were requested two steps ago.  Younger, in issue order: the 4 weight loads of the step before this one, and
            // around the top of a tile the previous tile's 4 stores and the 10 DMA instructions (first tile: no stores yet).
            u4_t (&cw)[4] = w[s & 1];
            if constexpr (t == 0 || t == 1) {
                if (s < kSteps && it == 0) wait_w<4 + kDma>(cw[0], cw[1], cw[2], cw[3]);
                else wait_w<4 + kStores + kDma>(cw[0], cw[1], cw[2], cw[3]);
            } else {
                wait_w<4>(cw[0], cw[1], cw[2], cw[3]);
            }
            const h8_t a[2][2] = {{__builtin_bit_cast(h8_t, cw[0]), __builtin_bit_cast(h8_t, cw[1])},
                                  {__builtin_bit_cast(h8_t, cw[2]), __builtin_bit_cast(h8_t, cw[3])}};
            const h8_t (&x0)[4] = b0[s & 1];
            const h8_t (&x1f)[4] = b1[s & 1];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[m][ni] = mma(a[m][0], x0[ni], acc[m][ni]);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[m][ni] = mma(a[m][1], x1f[ni], acc[m][ni]);
            // this set's weights are spent: step t + 2 (of the next tile behind the last two; always issued: the counts stay constant)
            wload4(cw[0], cw[1], cw[2], cw[3], wlane + (uint32_t)(((t + 2) % kSteps) * 8192), wrsrc);
            if constexpr (t == 7) {
                // every wave has passed a counted wait that covers its DMA of this tile's top (steps >= 2): behind this barrier the other
                // buffer holds the whole next tile
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);

            if constexpr (t == kSteps - 1) {
                // ---- epilogue: y = ReLU(scale * acc + shift) -> fp16, 16 bytes per pixel and 8-channel group.  EXACTLY kStores stores per wave.
                int n, ty0, tx0;
                tile_coords(tile_at(it + half), n, ty0, tx0);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int oy = ty0 + 2 * (2 * ni + i0) + py, ox = tx0 + 2 * j0 + px;
                    float y[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        y[q] = __builtin_fmaf(acc[0][ni][q], ksc[q], ksh[q]);
                        y[4 + q] = __builtin_fmaf(acc[1][ni][q], ksc[4 + q], ksh[4 + q]);
                    }
                    if (p.relu) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) y[q] = fmaxf(y[q], 0.f);
                    }
                    u4_t r;
                    r[0] = pack_h2(y[0], y[1]); r[1] = pack_h2(y[2], y[3]); r[2] = pack_h2(y[4], y[5]); r[3] = pack_h2(y[6], y[7]);
                    uint16_t* dst = (uint16_t*)p.out + ((size_t)(n * H + oy) * W + ox) * 64 + c0;
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(dst), "v"(r) : "memory");      // (s_nop: the store-data WAR wait state hipcc cannot insert behind inline asm)
                }
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

hipError_t launch_dec_halo_f16(const DecHaloParams& p, int num_cus, hipStream_t s)
{
    static bool attr_done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!attr_done[dev & 63]) {
        e = hipFuncSetAttribute((const void*)dec_halo_f16, hipFuncAttributeMaxDynamicSharedMemorySize, kDecHaloF16LdsBytes);
        if (e != hipSuccess) return e;
        attr_done[dev & 63] = true;
    }
    const int n_tiles = p.ttab ? p.n_tab : p.n * (p.PH / 8) * (p.PW / 8);
    if (n_tiles <= 0) return hipSuccess;
    const int grid = ((n_tiles < num_cus ? n_tiles : num_cus) + 7) & ~7;
    hipLaunchKernelGGL(dec_halo_f16, dim3(grid), dim3(512), kDecHaloF16LdsBytes, s, p);
    return hipGetLastError();
}

}  // namespace sbbseg
