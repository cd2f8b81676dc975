// block_x3.hip -- a whole identity ResNet bottleneck block (stage 2 of the sbb nets: 256 -> 64 -> 64 -> 256 channels at 111 x 111)
// per launch in the split-fp16 mode (kF16X3: every activation and weight is hi + lo fp16, three MFMAs per product).
//
//   a = ReLU(BN(conv1x1(x, 256 -> 64)))              phase A, on the 6 x 18 halo of a 4 x 16 output tile
//   b = ReLU(BN(conv3x3(a, 64 -> 64)))               phase B, from the halo tile in LDS
//   y = ReLU(BN(conv1x1(b, 64 -> 256)) + x)          phase C
//
// Run as three launches in the split mode these layers move 4 bytes per element of the 256-channel tensor four times per
// block (2.0 ms per 140 patches, 4.0-4.3 TB/s); fused, x is read once (+ the halo overlap, from L2) and y written once.
// The 16-bit kernel (bottleneck_fused, kernels.hip) does not carry over: doubled operands do not fit its LDS / register
// budget.  What changes here:
//   * tile 4 x 16 output pixels (halo 6 x 18 = 108 pixels): inner x (the residual) is 64 VGPRs per lane
//   * W1 and W3 (hi + lo: 64 KB each) fill the LDS; a (halo, 112 rows x 256 B) and b (64 rows x 256 B) SHARE one 28 KB region:
//     phase B keeps its results in registers until every wave has finished reading a
//   * x arrives straight in MFMA B-fragment registers (pixel = lane & 15, 8 channels per lane, hi and lo), the inner
//     pixels' fragments double as phase C's residual.  The NEXT tile's x is requested into the same registers as soon as the
//     current tile has spent them: the border pixels right after phase A (in flight across phases B and C), the inner pixels
//     K-step by K-step inside phase C, right after the residual add that reads them (a second inner buffer would not fit:
//     x 128 + 3x3 weights 144 VGPRs are live throughout)
//   * the wave's 3x3 weights (hi + lo fragments of its 16 output channels) live in 144 VGPRs
//   * pixel rows in LDS are 256 B = [64 hi][64 lo] in 16 granule slots, slot = (granule + 2 * row) & 15: conflict-free for
//     the 16-lane groups of ds_read_b128 (two k-groups x eight consecutive rows -> the even and the odd slots)
// One block of four waves per CU (up to 512 VGPRs per lane).  Wave w owns inner row w and border tile w (11 of the 44 border
// pixels) in phases A and C, and the 16 output channels of MFMA row block w in phase B.
#include "internal.h"

namespace sbbseg {

typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;
template <int N> struct IC { static constexpr int value = N; };

namespace {

constexpr int kHaloW = 18, kHaloRows = 6 * kHaloW;              // 108 halo pixels
constexpr int kW1Bytes = 8 * 4 * 2 * 1024;                      // [8 kk][4 mi][hi | lo][64 lanes x 16 B]
constexpr int kW3Bytes = 2 * 16 * 2 * 1024;                     // [2 kk][16 mi][hi | lo][64 lanes x 16 B]
constexpr int kABBytes = 112 * 256;                             // a: 108 halo rows + 4 dump rows; b: rows 0..63
constexpr int kCstBytes = (4 * 64 + 2 * 256) * 4;
constexpr int kBlockX3LdsBytes = kW1Bytes + kW3Bytes + kABBytes + kCstBytes;      // 162 816 <= 163 840

__device__ inline f4_t mma(h8_t a, h8_t b, f4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// w * x with both operands split: small terms first
__device__ inline f4_t mma3(h8_t wh, h8_t wl, h8_t xh, h8_t xl, f4_t c) { return mma(wh, xh, mma(wh, xl, mma(wl, xh, c))); }

__device__ inline void split8(const float (&y)[8], h8_t& hi, h8_t& lo)
{
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float v = fminf(fmaxf(y[q], -65504.f), 65504.f);
        const _Float16 h = (_Float16)v;
        hi[q] = h;
        lo[q] = (_Float16)(v - (float)h);
    }
}

}  // namespace

// FULL: the next tile's inner x has its own registers and is requested before phase A (in flight across all three phases);
// !FULL: it is requested K-step by K-step inside phase C into the registers the residual add has just released
// PROJ: the projection block at the head of the stage (64 -> 64 -> 64 -> 256 channels; the planner has folded the shortcut conv into
// the last contraction: y = ReLU(BN(W3' . [b, x]))).  x is 64 channels (2 K-steps): W1 moves to 64 VGPRs, W3' (K = 128: 128 KB as
// hi | lo fragments) takes the whole weight region of the LDS, phase C contracts b (LDS) and the inner x fragments (registers),
// and there is no residual add.
template <bool FULL, bool PROJ>
__global__ __launch_bounds__(256, 1) void block_x3(const BlockParams p)
{
    static_assert(FULL || !PROJ, "the projection block keeps its inner x for the whole of phase C: full next-tile buffer only");
    constexpr int KA = PROJ ? 2 : 8;                            // K-steps (32 channels) of phase A
    constexpr int KC = PROJ ? 4 : 2;                            // K-steps of phase C
    constexpr int PIXB = KA * 128;                              // bytes per stored x pixel (channel groups of [32 hi][32 lo])
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_w1 = smem;                                        // (identity block only)
    char* lds_w3 = PROJ ? smem : lds_w1 + kW1Bytes;
    char* lds_ab = smem + kW1Bytes + kW3Bytes;                  // (PROJ: W3' fills kW1Bytes + kW3Bytes = 128 KB)
    float* cst = (float*)(lds_ab + kABBytes);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int tiles_x = (p.W + 15) / 16, tiles_y = (p.H + 3) / 4;
    const int tiles_per_patch = tiles_x * tiles_y;
    const int n_tiles = p.n * tiles_per_patch;
    // XCD-contiguous walk: XCD x = block % 8 owns tiles [x * per_xcd, (x + 1) * per_xcd) -- vertical halo neighbours share an L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, GX = gridDim.x >> 3;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int xcd_lo = xcd * per_xcd, xcd_hi = min(n_tiles, xcd_lo + per_xcd);
    const int my_tiles = xcd_lo + slot < xcd_hi ? (xcd_hi - xcd_lo - slot + GX - 1) / GX : 0;
    if (my_tiles <= 0) return;
    auto tile_at = [&](int it) __attribute__((always_inline)) -> int { return xcd_lo + slot + it * GX; };

    // ---- one-time: weights and constants to LDS, this wave's 3x3 fragments to registers
    if constexpr (!PROJ)
        for (int i = tid; i < kW1Bytes / 16; i += 256) ((uint4*)lds_w1)[i] = ((const uint4*)p.w1)[i];
    for (int i = tid; i < KC * 16 * 2 * 64; i += 256) ((uint4*)lds_w3)[i] = ((const uint4*)p.w3)[i];
    h8_t w1h[PROJ ? KA : 1][PROJ ? 4 : 1], w1l[PROJ ? KA : 1][PROJ ? 4 : 1];      // PROJ: W1 in registers, [kk][mi]
    if constexpr (PROJ) {
        const uint4* src = (const uint4*)p.w1 + lane;
#pragma unroll
        for (int kk = 0; kk < KA; ++kk)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                w1h[kk][mi] = __builtin_bit_cast(h8_t, src[(size_t)((kk * 4 + mi) * 2 + 0) * 64]);
                w1l[kk][mi] = __builtin_bit_cast(h8_t, src[(size_t)((kk * 4 + mi) * 2 + 1) * 64]);
            }
    }
    if (tid < 64) {
        cst[tid] = p.s1[tid] * p.wmul1; cst[64 + tid] = p.b1[tid]; cst[128 + tid] = p.s2[tid] * p.wmul2; cst[192 + tid] = p.b2[tid];
    }
    cst[256 + tid] = p.s3[tid] * p.wmul3; cst[512 + tid] = p.b3[tid];
    h8_t wfh[9][2], wfl[9][2];                                  // [tap][kk] of MFMA row block `wave`; w2 = [hi | lo][9][2][4 mi][64 lanes]
    {
        const uint4* src = (const uint4*)p.w2 + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const size_t f = (size_t)((t * 2 + kk) * 4 + wave) * 64;
                wfh[t][kk] = __builtin_bit_cast(h8_t, src[f]);
                wfl[t][kk] = __builtin_bit_cast(h8_t, src[(size_t)9 * 2 * 4 * 64 + f]);
            }
    }

    // ---- this lane's two halo pixels: j = 0 inner pixel (row `wave`, column frow), j = 1 border pixel wave * 11 + frow (frow < 11)
    int hy[2], hx[2], hr[2];
    hy[0] = wave + 1; hx[0] = frow + 1; hr[0] = hy[0] * kHaloW + hx[0];
    {
        const int bi = wave * 11 + frow;
        int y, x;
        if (frow >= 11) { y = 6; x = frow - 11; }               // dummies: dump rows 108.., never inside the image
        else if (bi < 18) { y = 0; x = bi; }
        else if (bi < 36) { y = 5; x = bi - 18; }
        else if (bi < 40) { y = 1 + (bi - 36); x = 0; }
        else { y = 1 + (bi - 40); x = 17; }
        hy[1] = y; hx[1] = x; hr[1] = frow >= 11 ? 108 + ((frow - 11) & 3) : y * kHaloW + x;
    }

    // tile-invariant LDS read offsets of phases B and C (no address arithmetic per fragment read): row = frow + rc with rc known at
    // compile time, slot of granule G = (G + 2 row) & 15 = (s0 + D) & 15, s0 = (fg + 2 frow) & 15, D = kk * 4 + 8 * lo + (2 rc & 15)
    const char* rb[8];
    {
        const int s0 = (fg + 2 * frow) & 15;
#pragma unroll
        for (int e = 0; e < 8; ++e) rb[e] = lds_ab + frow * 256 + (((s0 + 2 * e) & 15) << 4);
    }

    // row rotation on the WRITE side: 2 slots per row like the reads (timing probe p.dbg & 1: 1 slot per row -- the 8-lane groups of
    // ds_write_b128 then hit eight different bank quads instead of four twice; the reads no longer find their data: results are wrong)
    const int wrot = SBBSEG_PROBE(p.dbg & 1) ? 1 : 2;
    bool inimg[2];
    uint32_t xoff[2];
    auto locate = [&](int tile, bool (&in)[2], uint32_t (&off)[2]) __attribute__((always_inline)) {
        const int n = tile / tiles_per_patch;
        const int rem = tile - n * tiles_per_patch;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int Y = ty * 4 - 1 + hy[j], X = tx * 16 - 1 + hx[j];
            in[j] = ((unsigned)Y < (unsigned)p.H) & ((unsigned)X < (unsigned)p.W) & (hy[j] < 6);
            // stored pixel = 8 channel groups of [32 hi][32 lo] halves (1 KB); outside the image: the zero header
            off[j] = in[j] ? (uint32_t)((n * p.H + Y) * p.W + X) * (uint32_t)PIXB + (uint32_t)(kZeroHeaderBytes + fg * 16) : 0u;
        }
    };
    h8_t xih[KA], xil[KA], xbh[KA], xbl[KA];                    // inner pixel, border pixel: [K-step] fragments, hi and lo
    h8_t nih[FULL ? KA : 1], nil[FULL ? KA : 1];                // FULL: the next tile's inner pixel
    auto fetch = [&](uint32_t off, h8_t (&dh)[KA], h8_t (&dl)[KA]) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < KA; ++kk) {
            dh[kk] = *(const h8_t*)(p.x + off + kk * 128);          // stored pixel = 8 groups of [32 hi][32 lo]: one 128-byte line per K-step
            dl[kk] = *(const h8_t*)(p.x + off + kk * 128 + 64);
        }
    };

    // (a use the compiler can see: it waits for the one-time weight loads HERE.  Left to their first use inside the tile loop -- phase B for
    // the 3x3 fragments -- its wait is vmcnt(0) on every trip: it drained the next tile's x prefetch, just issued, once per tile)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) asm volatile("" : "+v"(wfh[t][kk]), "+v"(wfl[t][kk]));
    if constexpr (PROJ) {
#pragma unroll
        for (int kk = 0; kk < KA; ++kk)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) asm volatile("" : "+v"(w1h[kk][mi]), "+v"(w1l[kk][mi]));
    }
    locate(tile_at(0), inimg, xoff);
    fetch(xoff[0], xih, xil);
    fetch(xoff[1], xbh, xbl);
    // (again a visible use: the first tile's x is waited for in front of the loop.  The compiler merges the loop entry with the back edge
    // and keeps the SMALLER count per register: with these 32 loads pending at the entry, every wait of phase A came out as if nothing
    // but the later x loads were younger -- vmcnt(15) .. vmcnt(0) in place of vmcnt(47) .. -- i.e. all of phase C's stores had to land)
#pragma unroll
    for (int kk = 0; kk < KA; ++kk) asm volatile("" : "+v"(xih[kk]), "+v"(xil[kk]), "+v"(xbh[kk]), "+v"(xbl[kk]));
    __syncthreads();                                            // weights / constants visible

    for (int it = 0; it < my_tiles; ++it) {
        const int tile = tile_at(it);
        const bool more = it + 1 < my_tiles;
        // The next tile's x is requested UNCONDITIONALLY (the last tile asks for its own again) and the output stores below are
        // buffer stores that drop the lanes outside the image instead of branching around them: with every vector-memory operation
        // of the loop in straight-line code the compiler counts its waits exactly -- phase A's K-step kk waits for the two loads
        // phase C's group kk issued, vmcnt(30 - 4 kk), not for the stores behind them.  (With `if (more)` / `if (live)` around them
        // it could not count the stores and waited for all of them -- and for the loads between them -- before phase A.)
        bool in_next[2];
        uint32_t off_next[2];
        locate(tile_at(more ? it + 1 : it), in_next, off_next);
        if constexpr (FULL) {
#pragma unroll
            for (int kk = 0; kk < KA; ++kk) {
                nih[kk] = *(const h8_t*)(p.x + off_next[0] + kk * 128);
                nil[kk] = *(const h8_t*)(p.x + off_next[0] + kk * 128 + 64);
            }
        }

        // ---- phase A: a[halo pixel][64] = ReLU(s1 * (W1 . x) + b1), zero outside the image
        {
            f4_t acc[4][2];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[mi][j] = (f4_t){0.f, 0.f, 0.f, 0.f};
            // W1 fragments of K-step kk + 1 are requested before the MFMAs of K-step kk (register double buffer): with one wave per
            // SIMD nothing else covers the LDS round trip
            h8_t wh[2][4], wl[2][4];
            auto load_w1 = [&](int kk, h8_t (&dh)[4], h8_t (&dl)[4]) __attribute__((always_inline)) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    if constexpr (PROJ) { dh[mi] = w1h[kk][mi]; dl[mi] = w1l[kk][mi]; }
                    else {
                        dh[mi] = *(const h8_t*)(lds_w1 + ((kk * 4 + mi) * 2 + 0) * 1024 + lane * 16);
                        dl[mi] = *(const h8_t*)(lds_w1 + ((kk * 4 + mi) * 2 + 1) * 1024 + lane * 16);
                    }
                }
            };
            load_w1(0, wh[0], wl[0]);
#pragma unroll
            for (int kk = 0; kk < KA; ++kk) {
                if (kk + 1 < KA) load_w1(kk + 1, wh[(kk + 1) & 1], wl[(kk + 1) & 1]);
                const h8_t (&ch)[4] = wh[kk & 1];
                const h8_t (&cl)[4] = wl[kk & 1];
                // three sweeps over the eight accumulators (small terms first): MFMAs on one accumulator stay 8 instructions apart
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) { acc[mi][0] = mma(cl[mi], xih[kk], acc[mi][0]); acc[mi][1] = mma(cl[mi], xbh[kk], acc[mi][1]); }
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) { acc[mi][0] = mma(ch[mi], xil[kk], acc[mi][0]); acc[mi][1] = mma(ch[mi], xbl[kk], acc[mi][1]); }
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) { acc[mi][0] = mma(ch[mi], xih[kk], acc[mi][0]); acc[mi][1] = mma(ch[mi], xbh[kk], acc[mi][1]); }
                __builtin_amdgcn_sched_barrier(0);
            }
            fetch(off_next[1], xbh, xbl);                       // border x is spent: its registers take the next tile's
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int c0 = s2 * 32 + fg * 8;
                float sc[8], sh[8];
                *(float4*)&sc[0] = *(const float4*)(cst + c0); *(float4*)&sc[4] = *(const float4*)(cst + c0 + 4);
                *(float4*)&sh[0] = *(const float4*)(cst + 64 + c0); *(float4*)&sh[4] = *(const float4*)(cst + 64 + c0 + 4);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float y[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        y[q] = fmaxf(__builtin_fmaf(acc[2 * s2][j][q], sc[q], sh[q]), 0.f);
                        y[4 + q] = fmaxf(__builtin_fmaf(acc[2 * s2 + 1][j][q], sc[4 + q], sh[4 + q]), 0.f);
                    }
                    h8_t vh, vl;
                    split8(y, vh, vl);
                    if (!inimg[j]) { vh = (h8_t){0, 0, 0, 0, 0, 0, 0, 0}; vl = vh; }
                    char* row = lds_ab + hr[j] * 256;
                    *(h8_t*)(row + (((s2 * 4 + fg + wrot * hr[j]) & 15) << 4)) = vh;
                    *(h8_t*)(row + (((8 + s2 * 4 + fg + wrot * hr[j]) & 15) << 4)) = vl;
                }
            }
        }
        __syncthreads();

        // ---- phase B: b[pixel][16 channels of row block `wave`] = ReLU(s2 * conv3x3(a) + b2); kept in registers until a is dead
        f4_t bacc[4];
        {
#pragma unroll
            for (int i = 0; i < 4; ++i) bacc[i] = (f4_t){0.f, 0.f, 0.f, 0.f};
            h8_t fh[2][4], fl[2][4];
            auto load_a = [&](int step, h8_t (&dh)[4], h8_t (&dl)[4]) __attribute__((always_inline)) {
                const int t = step >> 1, kk = step & 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rc = (i + t / 3) * kHaloW + t % 3;
                    const int eh = (kk * 4 + 2 * rc) & 15, el = (8 + kk * 4 + 2 * rc) & 15;
                    dh[i] = *(const h8_t*)(rb[eh >> 1] + rc * 256);
                    dl[i] = *(const h8_t*)(rb[el >> 1] + rc * 256);
                }
            };
            load_a(0, fh[0], fl[0]);
#pragma unroll
            for (int step = 0; step < 18; ++step) {             // (tap, K-step): fragments of step + 1 are in flight during step's MFMAs
                const int t = step >> 1, kk = step & 1;
                if (step + 1 < 18) load_a(step + 1, fh[(step + 1) & 1], fl[(step + 1) & 1]);
                const h8_t (&bh)[4] = fh[step & 1];
                const h8_t (&bl)[4] = fl[step & 1];
#pragma unroll
                for (int i = 0; i < 4; ++i) bacc[i] = mma(wfl[t][kk], bh[i], bacc[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) bacc[i] = mma(wfh[t][kk], bl[i], bacc[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) bacc[i] = mma(wfh[t][kk], bh[i], bacc[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                        // every wave has read a: b may overwrite it
        {
            const int sB = wave >> 1, half = wave & 1;
            const int cB = sB * 32 + fg * 8 + half * 4;
            const float4 sc = *(const float4*)(cst + 128 + cB), sh = *(const float4*)(cst + 192 + cB);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 16 + frow;
                const float v[4] = {fmaxf(__builtin_fmaf(bacc[i][0], sc.x, sh.x), 0.f), fmaxf(__builtin_fmaf(bacc[i][1], sc.y, sh.y), 0.f),
                                    fmaxf(__builtin_fmaf(bacc[i][2], sc.z, sh.z), 0.f), fmaxf(__builtin_fmaf(bacc[i][3], sc.w, sh.w), 0.f)};
                typedef __attribute__((ext_vector_type(4))) _Float16 h4_t;
                h4_t vh, vl;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float c = fminf(v[q], 65504.f);
                    const _Float16 h = (_Float16)c;
                    vh[q] = h;
                    vl[q] = (_Float16)(c - (float)h);
                }
                char* px = lds_ab + row * 256;
                *(h4_t*)(px + (((sB * 4 + fg + wrot * row) & 15) << 4) + half * 8) = vh;
                *(h4_t*)(px + (((8 + sB * 4 + fg + wrot * row) & 15) << 4) + half * 8) = vl;
            }
        }
        __syncthreads();

        // ---- phase C: y[inner row `wave`][256] = ReLU(s3 * (W3 . b) + b3 + x), 32 channels at a time
        {
            const int n = tile / tiles_per_patch;
            const int rem = tile - n * tiles_per_patch;
            const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
            const int oy = ty * 4 + wave, ox = tx * 16 + frow;
            const bool live = oy < p.H && ox < p.W;
            h8_t bh[2], bl[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {                    // row = wave * 16 + frow: 2 * 16 * wave = 0 mod 16
                bh[kk] = *(const h8_t*)(rb[(kk * 4) >> 1] + wave * 16 * 256);
                bl[kk] = *(const h8_t*)(rb[(8 + kk * 4) >> 1] + wave * 16 * 256);
            }
            __syncthreads();                                    // b is in registers everywhere: the next tile's phase A may write a
            // this patch's plane of the output as a buffer: lanes outside the image store at an offset past its end (dropped)
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((char*)p.out + (size_t)n * p.H * p.W * 1024, 0, p.H * p.W * 1024, 0x00020000);
            const uint32_t ooff = live ? (uint32_t)((oy * p.W + ox) * 1024 + fg * 16) : 0x80000000u;
            // channel group s3 + 1's weight fragments and constants are requested before group s3's MFMAs and epilogue
            h8_t w3h[2][KC][2], w3l[2][KC][2];
            // (the constants are requested a group ahead too, except in the FULL form: it has no 16 registers left for the second set --
            // there they are requested in front of the group's own MFMAs)
            constexpr int KCS = FULL && !PROJ ? 1 : 2;
            float4 kc[KCS][4];                                  // scale[c0..c0+7], shift[c0..c0+7]
            auto load_k = [&](int s3, float4 (&dk)[4]) __attribute__((always_inline)) {
                const int c0 = s3 * 32 + fg * 8;
                dk[0] = *(const float4*)(cst + 256 + c0); dk[1] = *(const float4*)(cst + 256 + c0 + 4);
                dk[2] = *(const float4*)(cst + 512 + c0); dk[3] = *(const float4*)(cst + 512 + c0 + 4);
            };
            auto load_c = [&](int s3, h8_t (&dh)[KC][2], h8_t (&dl)[KC][2]) __attribute__((always_inline)) {
#pragma unroll
                for (int kk = 0; kk < KC; ++kk)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        dh[kk][m] = *(const h8_t*)(lds_w3 + ((kk * 16 + 2 * s3 + m) * 2 + 0) * 1024 + lane * 16);
                        dl[kk][m] = *(const h8_t*)(lds_w3 + ((kk * 16 + 2 * s3 + m) * 2 + 1) * 1024 + lane * 16);
                    }
            };
            load_c(0, w3h[0], w3l[0]);
            if constexpr (KCS == 2) load_k(0, kc[0]);
#pragma unroll
            for (int s3 = 0; s3 < 8; ++s3) {
                if (s3 + 1 < 8) {
                    load_c(s3 + 1, w3h[(s3 + 1) & 1], w3l[(s3 + 1) & 1]);
                    if constexpr (KCS == 2) load_k(s3 + 1, kc[(s3 + 1) & 1]);
                }
                if constexpr (KCS == 1) load_k(s3, kc[0]);
                f4_t acc[2] = {(f4_t){0.f, 0.f, 0.f, 0.f}, (f4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int kk = 0; kk < KC; ++kk) {
                    const h8_t (&wh)[2] = w3h[s3 & 1][kk];
                    const h8_t (&wl)[2] = w3l[s3 & 1][kk];
                    // identity block: two K-steps over b.  Projection block: four K-steps in the SOURCE ORDER of the conv it replaces
                    // (p.proj == 1: [b, x]; 2: [x, b]) -- the same accumulation order, hence the same bits
                    const bool from_b = !PROJ || ((kk < 2) != (p.proj == 2));
                    const h8_t qh = from_b ? bh[kk & 1] : xih[(kk & 1) % KA], ql = from_b ? bl[kk & 1] : xil[(kk & 1) % KA];
                    acc[0] = mma(wl[0], qh, acc[0]); acc[1] = mma(wl[1], qh, acc[1]);
                    acc[0] = mma(wh[0], ql, acc[0]); acc[1] = mma(wh[1], ql, acc[1]);
                    acc[0] = mma(wh[0], qh, acc[0]); acc[1] = mma(wh[1], qh, acc[1]);
                }
                float sc[8], sh[8], y[8];
                *(float4*)&sc[0] = kc[s3 & (KCS - 1)][0]; *(float4*)&sc[4] = kc[s3 & (KCS - 1)][1];
                *(float4*)&sh[0] = kc[s3 & (KCS - 1)][2]; *(float4*)&sh[4] = kc[s3 & (KCS - 1)][3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    y[q] = __builtin_fmaf(acc[0][q], sc[q], sh[q]);
                    y[4 + q] = __builtin_fmaf(acc[1][q], sc[4 + q], sh[4 + q]);
                }
                if constexpr (PROJ) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) y[q] = fmaxf(y[q], 0.f);
                } else {
                    // the residual: this lane's x fragment of K-step s3 = channels c0 .. c0 + 7 of its inner pixel (hi + lo is exact in fp32)
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        y[q] = fmaxf(__fadd_rn(y[q], __fadd_rn((float)xih[s3 % KA][q], (float)xil[s3 % KA][q])), 0.f);      // (= add_split8, kernels.hip)
                }
                if constexpr (!FULL) {                          // this K-step's x is spent: its registers take the next tile's
                    xih[s3 % KA] = *(const h8_t*)(p.x + off_next[0] + s3 * 128);
                    xil[s3 % KA] = *(const h8_t*)(p.x + off_next[0] + s3 * 128 + 64);
                }
                h8_t vh, vl;
                split8(y, vh, vl);
                typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, vh), orsrc, ooff + s3 * 128, 0, 0);         // channel group s3: [32 hi][32 lo]
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, vl), orsrc, ooff + s3 * 128 + 64, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        inimg[0] = in_next[0]; inimg[1] = in_next[1];
        if constexpr (FULL) {
#pragma unroll
            for (int kk = 0; kk < KA; ++kk) { xih[kk] = nih[kk]; xil[kk] = nil[kk]; }
        }
    }
}

hipError_t launch_block_x3(const BlockParams& p, int num_cus, hipStream_t s)
{
    const int n_tiles = p.n * ((p.H + 3) / 4) * ((p.W + 15) / 16);
    int grid = n_tiles < num_cus ? n_tiles : num_cus;
    grid = (grid + 7) & ~7;                                     // the XCD-contiguous walk needs a multiple of 8 blocks
    static bool attr_done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!attr_done[dev & 63]) {
        e = hipFuncSetAttribute((const void*)block_x3<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kBlockX3LdsBytes);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)block_x3<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kBlockX3LdsBytes);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)block_x3<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kBlockX3LdsBytes);
        if (e != hipSuccess) return e;
        attr_done[dev & 63] = true;
    }
    if (p.proj) hipLaunchKernelGGL((block_x3<true, true>), dim3(grid), dim3(256), kBlockX3LdsBytes, s, p);
    // identity blocks: the full next-tile buffer is the default since round 4 (its inner x is requested a whole tile ahead; with the
    // constants of phase C single-buffered it fits 256 + 248 registers without scratch: 1.23 -> 1.18 ms per 140 patches); conv variant
    // bit 20 (pq cleared) selects the in-place prefetch for A/B
    else if (p.pq) hipLaunchKernelGGL((block_x3<true, false>), dim3(grid), dim3(256), kBlockX3LdsBytes, s, p);
    else hipLaunchKernelGGL((block_x3<false, false>), dim3(grid), dim3(256), kBlockX3LdsBytes, s, p);
    return hipGetLastError();
}

}  // namespace sbbseg
