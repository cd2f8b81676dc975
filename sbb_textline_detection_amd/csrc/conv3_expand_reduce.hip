// conv3_expand_reduce.hip -- an identity bottleneck block's 3x3 conv, its last 1x1 conv and the NEXT block's first 1x1 conv in one launch
// (round 6; encoder stage 3 at 56 x 56, C = 128):
//
//   b  = ReLU(s2 * (W2 * a) + h2)              a, b: [n][H][W][C], 3x3 / stride 1 / zero padding 1        ("GEMM 0", new here)
//   y  = ReLU(s3 * (W3 . b) + h3 + x)          x, y: [n][H][W][4C]                                         (expand_reduce's GEMM 1)
//   a' = ReLU(s1 * (W1 . y) + h1)              a': [n][H][W][C]                                            (expand_reduce's GEMM 2)
//
// As two launches (conv_igemm_mfma for the 3x3, expand_reduce for the pair) b makes a round trip through HBM, the 3x3 is MFMA-bound
// with the memory system idle and the pair is HBM-bound with the matrix pipe 25 % busy.  Here a block of eight waves owns an 8 x 8
// pixel tile: the 10 x 10 halo of `a` is copied to LDS (LDS-DMA, in pieces of 256 bytes per pixel, rotated as in dec_halo_*), GEMM 0
// contracts it tap by tap into b, which goes from the epilogue registers straight into the LDS image expand_reduce's GEMM 1 reads; the
// halo of the NEXT tile is fetched while GEMM 1 / GEMM 2 of this one run, and the x / y streams of the pair run under GEMM 0 of the next.
// From GEMM 1 on this IS expand_reduce (same K-steps, same order, same MFMA triple, same epilogue arithmetic), and GEMM 0 walks the
// K-steps of the conv it replaces in that conv's order (taken from its K-step records) with its epilogue arithmetic: b, y and a' are
// bit-identical to the three launches (tests/test_gpu_parity.py).  b itself is never written to HBM.
// Every vector-memory operation of the loop is issued from inline asm and waited for by a hand-counted vmcnt (see dec_halo_x3.hip).
// One count is deliberately NOT the tight steady-state one: GEMM 0's first two steps wait `vmcnt(weights of the next step)` although the
// previous tile's a' stores (epilogue 2) are younger than their own weight request and could be left in flight.  The tight count
// `next step + 4 stores` is only right from a block's second tile on: its FIRST tile has no epilogue in front of it, the count is then
// looser than what is in flight, and steps 0 / 1 multiplied weight registers whose loads had not landed whenever those loads missed L2 --
// one tile in ~10^5 in the plain fp16 mode (tools/first_run_probe.py: a whole patch's labels, run to run; with the first tile drained in
// the prologue and the stores allowed for again: 0 of 144, profiles/r06_experiments.md section 7).  `vmcnt(next step)` is right for every
// tile and costs nothing measurable (the stores are acknowledged by then).  expand_reduce had the same tile top since round 4.  And the
// wait is one statement, not an `if (first tile) ... else ...` pair: the `"+v"` ties of two asm statements in two branches make the
// compiler copy the weight registers in front of one of them -- a copy of registers whose loads are still in flight.
#include "internal.h"

namespace sbbseg {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 h4_t;
typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u2_t;
#define LDS_AS __attribute__((address_space(3)))

template <int N> struct IC { static constexpr int value = N; };
template <int B, int E, class F> __device__ __attribute__((always_inline)) inline void static_for(F&& f)
{
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ inline f4_t mma(h8_t a, h8_t b, f4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

template <int N, class V> __device__ inline void split_n(const float (&y)[N], V& hi, V& lo)
{
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const float v = fminf(fmaxf(y[q], -65504.f), 65504.f);
        const _Float16 h = (_Float16)v;
        hi[q] = h;
        lo[q] = (_Float16)(v - (float)h);
    }
}
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
__device__ __attribute__((always_inline)) inline void wload2(u4_t& a, u4_t& b, uint32_t voff, u4_t rsrc)
{
    asm volatile("buffer_load_dwordx4 %0, %2, %3, 0 offen\n\t"
                 "buffer_load_dwordx4 %1, %2, %3, 0 offen offset:1024"
                 : "=&v"(a), "=&v"(b) : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __attribute__((always_inline)) inline void xload2(u4_t& h, u4_t& l, uint32_t voff, u4_t rsrc)
{
    asm volatile("buffer_load_dwordx4 %0, %2, %3, 0 offen\n\t"
                 "buffer_load_dwordx4 %1, %2, %3, 0 offen offset:64"
                 : "=&v"(h), "=&v"(l) : "v"(voff), "s"(rsrc) : "memory");
}
template <int N> __device__ __attribute__((always_inline)) inline void wait4(u4_t& a, u4_t& b, u4_t& c, u4_t& d)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
__device__ __attribute__((always_inline)) inline void glds16_hidden(const void* gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ inline u4_t make_rsrc(const void* base, uint32_t bytes)
{
    u4_t r;
    const uint64_t b = (uint64_t)(uintptr_t)base;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}

constexpr int kHaloPx = 100;                 // 10 x 10
constexpr int kHaloInstr = 25;               // wave-instructions of 4 pixels x 256 B per piece
constexpr int kPieceBytes = kHaloInstr * 1024;

}  // namespace

// C = channels of a, b and a' (128); x and y have 4 C.  X3: split mode (hi | lo planes, K-steps of 32 channels, three MFMAs per product);
// else plain fp16 (K-steps of 64 channels as two k-halves).  Tiles are 8 x 8 pixels: pixel block ni (16 pixels) = tile rows 2 ni, 2 ni + 1;
// lane frow of a fragment = pixel (2 ni + (frow >> 3), frow & 7) -- the index ni * 16 + frow is what expand_reduce calls the tile's pixel.
template <int C, bool X3>
__global__ __launch_bounds__(512, 2) void conv3_expand_reduce(const C3ERParams p)
{
    static_assert(C == 128, "stage 3 only (LDS: the split mode's halo + b + y chunk fill it at C = 128)");
    constexpr int EB = X3 ? 4 : 2;                              // stored bytes per channel
    constexpr int KCH = X3 ? 32 : 64;                           // channels per K-step
    constexpr int KS0 = 9 * C / KCH;                            // K-steps of GEMM 0
    constexpr int KS1 = C / KCH;                                // K-steps of GEMM 1
    constexpr int G2S = 256 / KCH;                              // K-steps of GEMM 2 per chunk
    constexpr int NCH = C / 64;                                 // 256-channel chunks of y
    constexpr int MI2 = C / 128;                                // row blocks of b / a' per wave
    constexpr int LG2 = 2 * MI2;                                // weight loads per K-step of GEMM 0 / GEMM 2
    constexpr int PB = C * EB, PY = 4 * C * EB, PA = C * EB;    // bytes per stored pixel: a / b, y / x, a'
    constexpr int NP = PB / 256;                                // 256-byte pieces of a halo pixel
    constexpr int YROW = 256 * EB;
    constexpr int SLB = PB / 16, SLY = YROW / 16;
    constexpr int kHaloBytes = NP * kPieceBytes, kBBytes = 64 * PB, kYBytes = 64 * YROW;
    constexpr int PER = KS1 + G2S;                              // K-steps of one chunk
    constexpr int STEPS = KS0 + NCH * PER;                      // K-steps per tile, all three GEMMs (even)
    constexpr int kEpi1 = X3 ? 16 : 8, kEpi2 = X3 ? 8 : 4;      // vector-memory operations of the epilogues: y stores + x loads; a' stores
    constexpr int kDma = 4 * NP;                                // halo DMA instructions per wave and tile
    static_assert(STEPS % 2 == 0 && KS0 % 2 == 0, "the weight ring alternates two register sets");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const lds_a = smem;                                   // NP pieces of [100 px][256 B]
    char* const lds_b = smem + kHaloBytes;
    char* const lds_y = lds_b + kBBytes;
    float* const cst = (float*)(lds_y + kYBytes);               // s3 * wmul3 [4C] | h3 [4C] | s1 * wmul1 [C] | h1 [C] | s2 * wmul2 [C] | h2 [C]
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(LDS_AS char*)smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int tiles_x = p.W >> 3, tiles_y = p.H >> 3;
    const int tpp = tiles_x * tiles_y;
    const int n_tiles = p.n * tpp;
    // XCD-contiguous walk: neighbouring tiles share halo lines in one L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, GX = gridDim.x >> 3;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int xcd_lo = xcd * per_xcd, xcd_hi = min(n_tiles, xcd_lo + per_xcd);
    const int my_tiles = xcd_lo + slot < xcd_hi ? (xcd_hi - xcd_lo - slot + GX - 1) / GX : 0;
    if (my_tiles <= 0) return;
    auto tile_at = [&](int it) __attribute__((always_inline)) -> int { return xcd_lo + slot + (it < my_tiles ? it : my_tiles - 1) * GX; };
    // tile -> patch, first pixel row / column
    auto tile_coords = [&](int tile, int& n, int& y0, int& x0) __attribute__((always_inline)) {
        n = tile / tpp;
        const int rem = tile - n * tpp;
        const int ty = rem / tiles_x;
        y0 = ty * 8;
        x0 = (rem - ty * tiles_x) * 8;
    };

    for (int i = tid; i < 4 * C; i += 512) { cst[i] = p.s3[i] * p.wmul3; cst[4 * C + i] = p.h3[i]; }
    for (int i = tid; i < C; i += 512) {
        cst[8 * C + i] = p.s1[i] * p.wmul1; cst[9 * C + i] = p.h1[i];
        cst[10 * C + i] = p.s2[i] * p.wmul2; cst[11 * C + i] = p.h2[i];
    }

    // ---- halo DMA: instruction k of a piece fills halo pixels 4 k .. 4 k + 3 (256 B each); lane l writes slot l & 15 of pixel 4 k + (l >> 4),
    // i.e. fetches granule (slot - 2 hx) & 15 of the piece.  Wave w issues k = w, w + 8, w + 16, w + 24 (past 24: k - 25 again, same bytes to
    // the same place): every wave issues 4 loads per piece
    int halo_e[4];                                              // hy | hx << 8 | source granule << 16 | instruction << 24
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        int k = wave + 8 * m;
        k = k < kHaloInstr ? k : k - kHaloInstr;
        const int hp = 4 * k + (lane >> 4);
        const int hy = hp / 10, hx = hp - hy * 10;
        halo_e[m] = hy | (hx << 8) | ((((lane & 15) - 2 * hx) & 15) << 16) | (k << 24);
    }
    auto issue_halo = [&](int tile) __attribute__((always_inline)) {
        int n, y0, x0;
        tile_coords(tile, n, y0, x0);
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int e = halo_e[m];
                const int Y = y0 - 1 + (e & 255), X = x0 - 1 + ((e >> 8) & 255);
                const bool ok = (unsigned)Y < (unsigned)p.H && (unsigned)X < (unsigned)p.W;
                const uint32_t off = ok ? (uint32_t)((n * p.H + Y) * p.W + X) * (uint32_t)PB + (uint32_t)(pc * 256 + ((e >> 16) & 15) * 16 + kZeroHeaderBytes) : 0u;
                const uint32_t dst = lds0 + (uint32_t)(pc * kPieceBytes) + (uint32_t)__builtin_amdgcn_readfirstlane(e >> 24) * 1024u;
                glds16_hidden(p.a + off, dst);
            }
    };

    const uint32_t npix = (uint32_t)p.n * (uint32_t)p.H * (uint32_t)p.W;
    const u4_t xrsrc = make_rsrc(p.x + kZeroHeaderBytes, npix * (uint32_t)PY);
    const u4_t yrsrc = make_rsrc(p.y + kZeroHeaderBytes, npix * (uint32_t)PY);
    const u4_t arsrc = make_rsrc(p.a2 + kZeroHeaderBytes, npix * (uint32_t)PA);
    // weights: w2frag = [K-step][wave][mi][hi | lo][64 lanes x 16 B], w3frag = [chunk][K-step][wave][m][hi | lo][..] (4 KB per wave and step),
    // w1frag = [chunk][K-step][wave][mi2][hi | lo][..]
    const u4_t w2rsrc = make_rsrc((const char*)p.w2frag + wave * (LG2 * 1024), (uint32_t)(KS0 * 8 * LG2 * 1024));
    const u4_t w3rsrc = make_rsrc((const char*)p.w3frag + wave * 4096, (uint32_t)(NCH * KS1 * 8 * 4096));
    const u4_t w1rsrc = make_rsrc((const char*)p.w1frag + wave * (LG2 * 1024), (uint32_t)(NCH * G2S * 8 * LG2 * 1024));
    uint32_t wlane = (uint32_t)lane * 16u;
    int frv = frow, fgv = fg;

    // lane-constant pieces of the addresses.  A tile's pixel (ni, frow) is image pixel (y0 + 2 ni + (frow >> 3), x0 + (frow & 7))
    int rot = fg + 2 * frow;                                    // b / y images: slot of granule fg of this lane's pixel row, before the K-step's 8 k
    const int i0 = frow >> 3, j0 = frow & 7;
    uint32_t xlane = (uint32_t)((i0 * p.W + j0) * PY + wave * (32 * EB) + fg * 16);
    uint32_t alane = MI2 == 2 ? (uint32_t)((i0 * p.W + j0) * PA + wave * (32 * EB) + fg * 16)
                              : (uint32_t)((i0 * p.W + j0) * PA + (wave >> 1) * (32 * EB) + fg * 16 + (wave & 1) * 8);
    const uint32_t ni_y = (uint32_t)(2 * p.W) * (uint32_t)PY, ni_a = (uint32_t)(2 * p.W) * (uint32_t)PA;      // pixel block ni + 1 = two image rows down
    int a0 = (i0 + 1) * 10 + j0 + 1, r0 = fg + 2 * (j0 + 1);    // halo: pixel index / slot of this lane's centre tap
    // GEMM 0's K-steps (the conv's own order): (dy & 255) | (dx & 255) << 8 | channel group << 16, wave-uniform
    int k0[KS0];
    {
        const __attribute__((address_space(4))) int* kp = (const __attribute__((address_space(4))) int*)(uintptr_t)p.k0;
#pragma unroll
        for (int t = 0; t < KS0; ++t) k0[t] = kp[t];
    }

    u4_t w[2][4];                                               // weight ring: step u in set u & 1
    h8_t bh[2][4], bl[2][4];                                    // pixel fragments of step u in set u & 1
    u4_t xh[4], xl[4];                                          // the residual of the chunk ahead: [pixel block]
    auto issue_w = [&](int u, u4_t (&d)[4]) __attribute__((always_inline)) {             // u in [0, STEPS)
        if (u < KS0) {
            if constexpr (MI2 == 2) wload4(d[0], d[1], d[2], d[3], wlane + (uint32_t)(u * 8 * 4096), w2rsrc);
            else wload2(d[0], d[1], wlane + (uint32_t)(u * 8 * 2048), w2rsrc);
            return;
        }
        const int v = u - KS0, j = v / PER, r = v % PER;
        if (r < KS1) wload4(d[0], d[1], d[2], d[3], wlane + (uint32_t)((j * KS1 + r) * 8 * 4096), w3rsrc);
        else if constexpr (MI2 == 2) wload4(d[0], d[1], d[2], d[3], wlane + (uint32_t)((j * G2S + r - KS1) * 8 * 4096), w1rsrc);
        else wload2(d[0], d[1], wlane + (uint32_t)((j * G2S + r - KS1) * 8 * 2048), w1rsrc);
    };
    auto issue_x = [&](uint32_t pix0, int j) __attribute__((always_inline)) {             // 8 loads (plain mode: 4)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const uint32_t off = xlane + pix0 * (uint32_t)PY + (uint32_t)ni * ni_y + (uint32_t)(j * YROW);
            if constexpr (X3) xload2(xh[ni], xl[ni], off, xrsrc);
            else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(xh[ni]) : "v"(off), "s"(xrsrc) : "memory");
        }
    };
    auto load_frags = [&](const char* base, int row_bytes, int slots, int k, h8_t (&dh)[4], h8_t (&dl)[4]) __attribute__((always_inline)) {
        const int sh = (rot + 8 * k) & (slots - 1), sl = (rot + 8 * k + 4) & (slots - 1);
        const char* a = base + frv * row_bytes;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            dh[ni] = *(const h8_t*)(a + ni * 16 * row_bytes + (sh << 4));
            dl[ni] = *(const h8_t*)(a + ni * 16 * row_bytes + (sl << 4));
        }
    };
    // GEMM 0, K-step t: tap (dy, dx) of channel group cg: piece cg >> 1, granules 8 (cg & 1) + fg (first fragment) and + 4 (second) of halo pixel
    // (2 ni + i0 + dy + 1, j0 + dx + 1); slot = (granule + 2 hx) & 15
    auto load_halo = [&](int t, h8_t (&dh)[4], h8_t (&dl)[4]) __attribute__((always_inline)) {
        const int rec = k0[t];
        const int dy = (rec << 24) >> 24, dx = (rec << 16) >> 24, cg = rec >> 16;
        const int hp = a0 + dy * 10 + dx;
        const int sh = (r0 + 2 * dx + 8 * (cg & 1)) & 15;
        const char* base = lds_a + (cg >> 1) * kPieceBytes + hp * 256;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            dh[ni] = *(const h8_t*)(base + ni * (20 * 256) + (sh << 4));
            dl[ni] = *(const h8_t*)(base + ni * (20 * 256) + (((sh + 4) & 15) << 4));
        }
    };

    // ---- prologue: the first tile's halo and its first chunk of x
    int cn, cy0, cx0;
    tile_coords(tile_at(0), cn, cy0, cx0);
    uint32_t pix0 = (uint32_t)((cn * p.H + cy0) * p.W + cx0);
    issue_halo(tile_at(0));
    issue_x(pix0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        asm volatile("" : "+v"(xh[ni]));
        if constexpr (X3) asm volatile("" : "+v"(xl[ni]));
    }
    __syncthreads();                                            // halo, constants visible
    issue_w(0, w[0]);
    issue_w(1, w[1]);

    for (int it = 0; it < my_tiles; ++it) {
        int nn, ny0, nx0;
        tile_coords(tile_at(it + 1), nn, ny0, nx0);
        const uint32_t pix_next = (uint32_t)((nn * p.H + ny0) * p.W + nx0);
        asm volatile("" : "+v"(rot), "+v"(xlane), "+v"(alane), "+v"(wlane), "+v"(frv), "+v"(fgv), "+v"(a0), "+v"(r0));
#pragma unroll
        for (int t = 0; t < KS0; ++t) asm volatile("" : "+s"(k0[t]));
        f4_t acc2[MI2][4];
        f4_t acc1[2][4];
        static_for<0, STEPS>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            constexpr bool g0 = u < KS0;
            constexpr int v = g0 ? 0 : u - KS0;
            constexpr int j = v / PER, r = v % PER;
            constexpr bool g1 = !g0 && r < KS1;
            constexpr int k = g0 ? u : (g1 ? r : r - KS1);
            constexpr int set = u & 1;
            // loads of the weight request of the step after this one (the only request younger than this step's, but for the bursts below)
            constexpr int un = (u + 1) % STEPS;
            constexpr int Lnext = un < KS0 ? LG2 : (((un - KS0) % PER) < KS1 ? 4 : LG2);
            // vector-memory operations issued between this step's weight request (end of step u - 2) and here, beside that one request:
            //   GEMM 0 steps 0, 1: the previous tile's epilogue 2 (a' stores)      GEMM 1 steps 0, 1 of chunk 0: the next tile's halo DMA
            //   GEMM 2 steps 0, 1: epilogue 1 (y stores + x loads)
            constexpr int extra = (!g0 && !g1 && k < 2) ? kEpi1 : ((g1 && j == 0 && k < 2) ? kDma : 0);
            if constexpr (g0 && k == 0) {
#pragma unroll
                for (int m = 0; m < MI2; ++m)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc2[m][ni] = (f4_t){0.f, 0.f, 0.f, 0.f};      // (GEMM 0's accumulators live in acc2: same shape)
                load_halo(0, bh[set], bl[set]);
            }
            if constexpr (g1 && k == 0) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc1[m][ni] = (f4_t){0.f, 0.f, 0.f, 0.f};
                load_frags(lds_b, PB, SLB, 0, bh[set], bl[set]);
            }
            if constexpr (!g0 && !g1 && k == 0) load_frags(lds_y, YROW, SLY, 0, bh[set], bl[set]);
            // the next step's pixel fragments, inside a phase
            if constexpr (g0 && k + 1 < KS0) load_halo(k + 1, bh[set ^ 1], bl[set ^ 1]);
            if constexpr (g1 && k + 1 < KS1) load_frags(lds_b, PB, SLB, k + 1, bh[set ^ 1], bl[set ^ 1]);
            if constexpr (!g0 && !g1 && k + 1 < G2S) load_frags(lds_y, YROW, SLY, k + 1, bh[set ^ 1], bl[set ^ 1]);
            u4_t (&cw)[4] = w[set];
            if constexpr (g0 && k < 2) {
                // (the a' stores of the previous tile's epilogue 2 sit between this step's weight request and here -- but not on a block's
                //  first tile: they are NOT allowed for in the count, see the header)
                wait4<Lnext>(cw[0], cw[1], cw[2], cw[3]);
            } else {
                wait4<Lnext + extra>(cw[0], cw[1], cw[2], cw[3]);
            }
            const h8_t (&ph)[4] = bh[set];
            const h8_t (&pl)[4] = bl[set];
            if constexpr (g1) {
                const h8_t ah[2] = {__builtin_bit_cast(h8_t, cw[0]), __builtin_bit_cast(h8_t, cw[2])};       // (plain mode: k-half 0)
                const h8_t al[2] = {__builtin_bit_cast(h8_t, cw[1]), __builtin_bit_cast(h8_t, cw[3])};       // (             k-half 1)
                if constexpr (X3) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) acc1[m][ni] = mma(al[m], ph[ni], acc1[m][ni]);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) acc1[m][ni] = mma(ah[m], pl[ni], acc1[m][ni]);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc1[m][ni] = mma(ah[m], ph[ni], acc1[m][ni]);
                if constexpr (!X3) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) acc1[m][ni] = mma(al[m], pl[ni], acc1[m][ni]);
                }
            } else {
                // GEMM 0 and GEMM 2: C / 8 output channels x 64 pixels per wave
                h8_t ah[MI2], al[MI2];
#pragma unroll
                for (int m = 0; m < MI2; ++m) { ah[m] = __builtin_bit_cast(h8_t, cw[2 * m]); al[m] = __builtin_bit_cast(h8_t, cw[2 * m + 1]); }
                if constexpr (X3) {
#pragma unroll
                    for (int m = 0; m < MI2; ++m)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) acc2[m][ni] = mma(al[m], ph[ni], acc2[m][ni]);
#pragma unroll
                    for (int m = 0; m < MI2; ++m)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) acc2[m][ni] = mma(ah[m], pl[ni], acc2[m][ni]);
                }
#pragma unroll
                for (int m = 0; m < MI2; ++m)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc2[m][ni] = mma(ah[m], ph[ni], acc2[m][ni]);
                if constexpr (!X3) {
#pragma unroll
                    for (int m = 0; m < MI2; ++m)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni) acc2[m][ni] = mma(al[m], pl[ni], acc2[m][ni]);
                }
            }
            // this set's weights are spent: step u + 2 (of the next tile behind the last two; always issued: the counts stay constant)
            issue_w((u + 2) % STEPS, cw);
            __builtin_amdgcn_sched_barrier(0);

            if constexpr (g0 && k == KS0 - 1) {
                // ---- epilogue 0: b = ReLU(s2 * acc + h2) -> the LDS image GEMM 1 reads (granule G of pixel p at slot (G + 2 (p & 15)) mod SLB).
                // No vector-memory operation here: the hand-counted queue is untouched
                const int c0 = MI2 == 2 ? wave * 32 + fgv * 8 : (wave >> 1) * 32 + fgv * 8 + (wave & 1) * 4;
                float sc[4 * MI2], sh[4 * MI2];
                *(float4*)&sc[0] = *(const float4*)(cst + 10 * C + c0);
                *(float4*)&sh[0] = *(const float4*)(cst + 11 * C + c0);
                if constexpr (MI2 == 2) {
                    *(float4*)&sc[4] = *(const float4*)(cst + 10 * C + c0 + 4);
                    *(float4*)&sh[4] = *(const float4*)(cst + 11 * C + c0 + 4);
                }
                // byte offset of channel c0 inside the stored pixel: split mode = groups of 32 channels [32 hi][32 lo]
                const int boff = X3 ? (c0 >> 5) * 128 + (c0 & 31) * 2 : c0 * 2;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    float y[4 * MI2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        y[q] = fmaxf(__builtin_fmaf(acc2[0][ni][q], sc[q], sh[q]), 0.f);
                        if constexpr (MI2 == 2) y[4 + q] = fmaxf(__builtin_fmaf(acc2[MI2 - 1][ni][q], sc[4 * (MI2 - 1) + q], sh[4 * (MI2 - 1) + q]), 0.f);
                    }
                    char* row = lds_b + (ni * 16 + frv) * PB;
                    const int s_hi = ((boff >> 4) + 2 * frv) & (SLB - 1), s_lo = (((boff + 64) >> 4) + 2 * frv) & (SLB - 1);
                    if constexpr (X3 && MI2 == 2) {
                        h8_t vh, vl;
                        split_n<8>(y, vh, vl);
                        *(h8_t*)(row + (s_hi << 4)) = vh;
                        *(h8_t*)(row + (s_lo << 4)) = vl;
                    } else if constexpr (X3) {
                        h4_t vh, vl;
                        split_n<4>(y, vh, vl);
                        *(h4_t*)(row + (s_hi << 4) + (boff & 8)) = vh;
                        *(h4_t*)(row + (s_lo << 4) + (boff & 8)) = vl;
                    } else if constexpr (MI2 == 2) {
                        u4_t vv;
                        vv[0] = pack_h2(y[0], y[1]); vv[1] = pack_h2(y[2], y[3]); vv[2] = pack_h2(y[4 % (4 * MI2)], y[5 % (4 * MI2)]); vv[3] = pack_h2(y[6 % (4 * MI2)], y[7 % (4 * MI2)]);
                        *(u4_t*)(row + (s_hi << 4)) = vv;
                    } else {
                        u2_t vv;
                        vv[0] = pack_h2(y[0], y[1]); vv[1] = pack_h2(y[2], y[3]);
                        *(u2_t*)(row + (s_hi << 4) + (boff & 8)) = vv;
                    }
                }
#pragma unroll
                for (int m = 0; m < MI2; ++m)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc2[m][ni] = (f4_t){0.f, 0.f, 0.f, 0.f};      // GEMM 2 accumulates across the chunks from here
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                   // b is complete; every wave has taken its last fragment from the halo
                asm volatile("" ::: "memory");
                issue_halo(tile_at(it + 1));                    // the next tile's halo lands under GEMM 1 / GEMM 2 (kDma loads per wave)
            }
            if constexpr (g1 && k == KS1 - 1) {
                // ---- epilogue 1 (expand_reduce's): y = ReLU(s3 * acc + h3 + x) -> hi | lo: to HBM and into the LDS image of the chunk
                if constexpr (X3)
                    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]), "+v"(xl[0]), "+v"(xl[1]), "+v"(xl[2]), "+v"(xl[3])
                                 : "n"(2 * LG2) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]) : "n"(2 * LG2) : "memory");
                const int c0 = j * 256 + wave * 32 + fgv * 8;
                float sc[8], sh[8];
                *(float4*)&sc[0] = *(const float4*)(cst + c0); *(float4*)&sc[4] = *(const float4*)(cst + c0 + 4);
                *(float4*)&sh[0] = *(const float4*)(cst + 4 * C + c0); *(float4*)&sh[4] = *(const float4*)(cst + 4 * C + c0 + 4);
                const int s_hi = (wave * (X3 ? 8 : 4) + rot) & (SLY - 1), s_lo = (wave * 8 + 4 + rot) & (SLY - 1);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    float y[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        y[q] = __builtin_fmaf(acc1[0][ni][q], sc[q], sh[q]);
                        y[4 + q] = __builtin_fmaf(acc1[1][ni][q], sc[4 + q], sh[4 + q]);
                    }
                    const uint32_t off = xlane + pix0 * (uint32_t)PY + (uint32_t)ni * ni_y + (uint32_t)(j * YROW);
                    char* row = lds_y + (ni * 16 + frv) * YROW;
                    if constexpr (X3) {
                        const h8_t rh = __builtin_bit_cast(h8_t, xh[ni]), rl = __builtin_bit_cast(h8_t, xl[ni]);
#pragma unroll
                        for (int q = 0; q < 8; ++q) y[q] = fmaxf(__fadd_rn(y[q], __fadd_rn((float)rh[q], (float)rl[q])), 0.f);
                        h8_t vh, vl;
                        split_n<8>(y, vh, vl);
                        asm volatile("buffer_store_dwordx4 %1, %0, %3, 0 offen\n\tbuffer_store_dwordx4 %2, %0, %3, 0 offen offset:64\n\ts_nop 1"
                                     :: "v"(off), "v"(vh), "v"(vl), "s"(yrsrc) : "memory");
                        *(h8_t*)(row + (s_hi << 4)) = vh;
                        *(h8_t*)(row + (s_lo << 4)) = vl;
                    } else {
                        const h8_t rr = __builtin_bit_cast(h8_t, xh[ni]);
#pragma unroll
                        for (int q = 0; q < 8; ++q) y[q] = fmaxf(__fadd_rn(y[q], (float)rr[q]), 0.f);
                        u4_t vv;
                        vv[0] = pack_h2(y[0], y[1]); vv[1] = pack_h2(y[2], y[3]); vv[2] = pack_h2(y[4], y[5]); vv[3] = pack_h2(y[6], y[7]);
                        asm volatile("buffer_store_dwordx4 %1, %0, %2, 0 offen\n\ts_nop 1" :: "v"(off), "v"(vv), "s"(yrsrc) : "memory");
                        *(u4_t*)(row + (s_hi << 4)) = vv;
                    }
                }
                // the residual of the chunk after this one (the next tile's first behind the last)
                if constexpr (j + 1 < NCH) issue_x(pix0, j + 1);
                else issue_x(pix_next, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                   // the y chunk is complete
                asm volatile("" ::: "memory");
            }
            if constexpr (!g0 && !g1 && k == G2S - 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                   // every wave has taken its last fragment of this y chunk (and, past GEMM 1's first steps
                asm volatile("" ::: "memory");                  // of chunk 0, has waited for its share of the next tile's halo)
            }
        });

        // ---- epilogue 2: a' = ReLU(s1 * acc + h1) -> hi | lo.  EXACTLY kEpi2 stores per wave.
        {
            const int c0 = MI2 == 2 ? wave * 32 + fgv * 8 : (wave >> 1) * 32 + fgv * 8 + (wave & 1) * 4;
            float sc[4 * MI2], sh[4 * MI2];
            *(float4*)&sc[0] = *(const float4*)(cst + 8 * C + c0);
            *(float4*)&sh[0] = *(const float4*)(cst + 9 * C + c0);
            if constexpr (MI2 == 2) {
                *(float4*)&sc[4] = *(const float4*)(cst + 8 * C + c0 + 4);
                *(float4*)&sh[4] = *(const float4*)(cst + 9 * C + c0 + 4);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const uint32_t off = alane + pix0 * (uint32_t)PA + (uint32_t)ni * ni_a;
                float y[4 * MI2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    y[q] = fmaxf(__builtin_fmaf(acc2[0][ni][q], sc[q], sh[q]), 0.f);
                    if constexpr (MI2 == 2) y[4 + q] = fmaxf(__builtin_fmaf(acc2[MI2 - 1][ni][q], sc[4 * (MI2 - 1) + q], sh[4 * (MI2 - 1) + q]), 0.f);
                }
                if constexpr (X3 && MI2 == 2) {
                    h8_t vh, vl;
                    split_n<8>(y, vh, vl);
                    asm volatile("buffer_store_dwordx4 %1, %0, %3, 0 offen\n\tbuffer_store_dwordx4 %2, %0, %3, 0 offen offset:64\n\ts_nop 1"
                                 :: "v"(off), "v"(vh), "v"(vl), "s"(arsrc) : "memory");
                } else if constexpr (X3) {
                    h4_t vh, vl;
                    split_n<4>(y, vh, vl);
                    asm volatile("buffer_store_dwordx2 %1, %0, %3, 0 offen\n\tbuffer_store_dwordx2 %2, %0, %3, 0 offen offset:64\n\ts_nop 1"
                                 :: "v"(off), "v"(vh), "v"(vl), "s"(arsrc) : "memory");
                } else if constexpr (MI2 == 2) {
                    u4_t vv;
                    vv[0] = pack_h2(y[0], y[1]); vv[1] = pack_h2(y[2], y[3]); vv[2] = pack_h2(y[4 % (4 * MI2)], y[5 % (4 * MI2)]); vv[3] = pack_h2(y[6 % (4 * MI2)], y[7 % (4 * MI2)]);
                    asm volatile("buffer_store_dwordx4 %1, %0, %2, 0 offen\n\ts_nop 1" :: "v"(off), "v"(vv), "s"(arsrc) : "memory");
                } else {
                    u2_t vv;
                    vv[0] = pack_h2(y[0], y[1]); vv[1] = pack_h2(y[2], y[3]);
                    asm volatile("buffer_store_dwordx2 %1, %0, %2, 0 offen\n\ts_nop 1" :: "v"(off), "v"(vv), "s"(arsrc) : "memory");
                }
            }
        }
        pix0 = pix_next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int C, bool X3> static hipError_t launch_one(const C3ERParams& p, int num_cus, hipStream_t s)
{
    constexpr int EB = X3 ? 4 : 2;
    constexpr int lds = (C * EB / 256) * kPieceBytes + 64 * C * EB + 64 * 256 * EB + 12 * C * 4;
    static_assert(lds <= 160 * 1024, "conv3_expand_reduce: LDS");
    static bool attr_done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!attr_done[dev & 63]) {
        e = hipFuncSetAttribute((const void*)conv3_expand_reduce<C, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_done[dev & 63] = true;
    }
    const int n_tiles = p.n * (p.H / 8) * (p.W / 8);
    if (n_tiles <= 0) return hipSuccess;
    const int grid = ((n_tiles < num_cus ? n_tiles : num_cus) + 7) & ~7;
    hipLaunchKernelGGL((conv3_expand_reduce<C, X3>), dim3(grid), dim3(512), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_conv3_expand_reduce(const C3ERParams& p, int num_cus, hipStream_t s)
{
    if (p.C == 128) return p.x3 ? launch_one<128, true>(p, num_cus, s) : launch_one<128, false>(p, num_cus, s);
    return hipErrorInvalidValue;
}

}  // namespace sbbseg
