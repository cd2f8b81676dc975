// expand_reduce_x3.hip -- stages 3 and 4 of the ResNet-50 encoder in the split-fp16 mode (kF16X3): the LAST 1x1 conv of an identity
// bottleneck block ("expand": C -> 4C channels, BN, + residual x, ReLU) and the FIRST 1x1 conv of the next block ("reduce": 4C -> C,
// BN, ReLU) in one launch, round 4.
//
//   y  = ReLU(s3 * (W3 . b) + h3 + x)          b: [M][C], x, y: [M][4C]   (y is still written: the next block's residual)
//   a' = ReLU(s1 * (W1 . y) + h1)              a': [M][C]
//
// As two launches of conv_igemm_mfma these layers are HBM-bound (3.1 - 4.5 TB/s): the expand reads b and x and writes y (4.5 C' bytes
// ... 9 KB per pixel in stage 4), the reduce reads y back (4 KB) and writes a'.  Fused, y goes from the expand's epilogue registers
// into LDS and is contracted there: 5 KB per pixel in place of 7 (stage 3), 10 in place of 14 (stage 4).
//
// A block of eight waves owns 64 consecutive pixels (the convs are pointwise: any 64) and walks the 4C channels of y in chunks of 256:
//   GEMM 1  y chunk [256 ch x 64 px] = W3 rows . b:  wave w owns 32 channels (2 row blocks) x 64 pixels, K = C in steps of 32 channels;
//           b (64 px x C, hi + lo) is resident in LDS for the whole tile (LDS-DMA, refilled for the next tile once the last chunk's GEMM 1
//           is through), W3 is streamed as A fragments straight into registers, two K-steps ahead
//   epilogue 1  BN, + x (loaded a chunk ahead into registers, fragment-shaped), ReLU, hi | lo split: 16-byte stores of y AND the same
//           registers into the LDS image of the chunk (64 px x 256 ch)
//   GEMM 2  a' [C ch x 64 px] += W1[:, chunk] . y chunk:  wave w owns C / 8 output channels x 64 pixels, accumulators live across chunks,
//           8 K-steps per chunk, W1 streamed like W3
//   epilogue 2  BN, ReLU, split, stores of a'
// Both contractions walk the K-steps in the order of the convs they replace, three MFMAs per product (lo*hi, hi*lo, hi*hi), one
// accumulator per output, and state the same epilogue arithmetic: y and a' are bit-identical to the two launches (tests/test_gpu_parity.py).
// (Every asm store ends in `s_nop 1`: the hazard recognizer does not see a store inside inline asm, so it does not insert the wait state the
// ISA wants between a store of more than 64 bits and a VALU write of its data registers -- lanes 12-15 of the first data register came out
// as the NEXT pixel block's values in the plain mode, where the compiler reuses the registers at once.)
// Every vector-memory operation of the loop is issued from inline asm and waited for by a hand-counted vmcnt (retirement is in issue
// order): see dec_halo_x3.hip for why.  Pixel rows in LDS are C * 4 bytes (b) / 1 KB (y chunk); 16-byte slot s of pixel p sits at slot
// (s + 2 (p & 15)) mod row: conflict-free for the 16-lane groups of ds_read_b128 (64 banks); the epilogue's ds_write_b128 (32 banks, 8-lane
// groups) lands two lanes on a bank -- PMC: 13-17 % of the LDS-busy cycles are conflict cycles, all from those 16 writes per chunk.
#include "internal.h"

namespace sbbseg {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 h4_t;
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

typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
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
__device__ __attribute__((always_inline)) inline void wload2(u4_t& a, u4_t& b, uint32_t voff, u4_t rsrc)
{
    asm volatile("buffer_load_dwordx4 %0, %2, %3, 0 offen\n\t"
                 "buffer_load_dwordx4 %1, %2, %3, 0 offen offset:1024"
                 : "=&v"(a), "=&v"(b) : "v"(voff), "s"(rsrc) : "memory");
}
// a pixel's hi and lo granule (64 bytes apart)
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

}  // namespace

// C = channels of b and a' (128: stage 3, 256: stage 4); y and x have 4 C.  X3: split mode (hi | lo planes, K-steps of 32 channels, three
// MFMAs per product); else plain fp16 (K-steps of 64 channels = two k-halves: the fragment pair (hi, lo) of the text above reads (kk = 0,
// kk = 1), one MFMA each, half the bytes everywhere) -- the same K-steps / accumulation order / epilogue arithmetic as conv_igemm_mfma's
// plain mode: bit-identical there too.
template <int C, bool X3>
__global__ __launch_bounds__(512, 2) void expand_reduce(const ExpRedParams p)
{
    constexpr int EB = X3 ? 4 : 2;                              // stored bytes per channel
    constexpr int KCH = X3 ? 32 : 64;                           // channels per K-step
    constexpr int KS1 = C / KCH;                                // K-steps of GEMM 1
    constexpr int G2S = 256 / KCH;                              // K-steps of GEMM 2 per chunk
    constexpr int NCH = C / 64;                                 // 256-channel chunks of y
    constexpr int MI2 = C / 128;                                // row blocks of a' per wave
    constexpr int LG2 = 2 * MI2;                                // weight loads per K-step of GEMM 2
    constexpr int PB = C * EB, PY = 4 * C * EB, PA = C * EB;    // bytes per stored pixel: b, y / x, a'
    constexpr int YROW = 256 * EB;                              // bytes per pixel of the y chunk's LDS image
    constexpr int SLB = PB / 16, SLY = YROW / 16;               // 16-byte slots per LDS row
    constexpr int kBBytes = 64 * PB, kYBytes = 64 * YROW;
    constexpr int STEPS = NCH * (KS1 + G2S);                    // K-steps per tile, both GEMMs (even)
    constexpr int kEpi1 = X3 ? 16 : 8, kEpi2 = X3 ? 8 : 4;      // vector-memory operations of the epilogues: y stores + x loads; a' stores
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const lds_b = smem;
    char* const lds_y = smem + kBBytes;
    float* const cst = (float*)(smem + kBBytes + kYBytes);      // s3 * wmul3 [4C] | h3 [4C] | s1 * wmul1 [C] | h1 [C]
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(LDS_AS char*)smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int n_tiles = (p.M + 63) >> 6;
    const int G = gridDim.x;
    const int my_tiles = (n_tiles - (int)blockIdx.x + G - 1) / G;
    if (my_tiles <= 0) return;
    auto tile_at = [&](int it) __attribute__((always_inline)) -> int { return (int)blockIdx.x + (it < my_tiles ? it : my_tiles - 1) * G; };

    for (int i = tid; i < 4 * C; i += 512) { cst[i] = p.s3[i] * p.wmul3; cst[4 * C + i] = p.h3[i]; }
    for (int i = tid; i < C; i += 512) { cst[8 * C + i] = p.s1[i] * p.wmul1; cst[9 * C + i] = p.h1[i]; }

    // ---- b tile DMA: instruction i fills bytes [1024 i, 1024 i + 1024) of the LDS image (64 rows of PB bytes); wave w issues i = w + 8 k.
    // (pixel / source granule of a lane are recomputed per tile from `dlane`: 2 x KS1 registers the loop has no room for)
    int dlane = lane;
    auto issue_b = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < KS1; ++k) {
            const int byte = (wave + 8 * k) * 1024 + dlane * 16;
            const int px = byte / PB, slot = (byte - px * PB) >> 4;
            const int go = ((slot - 2 * (px & 15)) & (SLB - 1)) * 16;          // byte offset of the source granule inside the stored pixel
            const int m = tile * 64 + px;
            const uint32_t off = m < p.M ? (uint32_t)m * (uint32_t)PB + (uint32_t)(go + kZeroHeaderBytes) : 0u;
            glds16_hidden(p.b + off, lds0 + (uint32_t)((wave + 8 * k) * 1024));
        }
    };

    // ---- buffers: x (read), y, a' (written); an offset past the end reads zero / is dropped (the last tile of a ragged M)
    const u4_t xrsrc = make_rsrc(p.x + kZeroHeaderBytes, (uint32_t)p.M * (uint32_t)PY);
    const u4_t yrsrc = make_rsrc(p.y + kZeroHeaderBytes, (uint32_t)p.M * (uint32_t)PY);
    const u4_t arsrc = make_rsrc(p.a2 + kZeroHeaderBytes, (uint32_t)p.M * (uint32_t)PA);
    // weights: w3frag = [chunk][K-step][wave][m][hi | lo][64 lanes x 16 B] (4 KB per wave and step), w1frag = [chunk][K-step][wave][mi2][hi | lo][..]
    const u4_t w3rsrc = make_rsrc((const char*)p.w3frag + wave * 4096, (uint32_t)(NCH * KS1 * 8 * 4096));
    const u4_t w1rsrc = make_rsrc((const char*)p.w1frag + wave * (LG2 * 1024), (uint32_t)(NCH * G2S * 8 * LG2 * 1024));
    uint32_t wlane = (uint32_t)lane * 16u;
    int frv = frow, fgv = fg;                                   // (copies the tile loop re-derives its addresses from: see the asm at its top)

    // lane-constant pieces of the addresses
    int rot = fg + 2 * frow;                                    // slot of granule fg of this lane's pixel row, before the K-step's 8 k
    uint32_t xlane = (uint32_t)(frow * PY + wave * (32 * EB) + fg * 16);      // x / y: pixel frow of a 16-pixel block, the wave's 32 channels of a chunk
    uint32_t alane = MI2 == 2 ? (uint32_t)(frow * PA + wave * (32 * EB) + fg * 16)
                              : (uint32_t)(frow * PA + (wave >> 1) * (32 * EB) + fg * 16 + (wave & 1) * 8);

    // step u of a tile: chunk j, GEMM 1 K-step k (u % (KS1 + 8) < KS1) or GEMM 2 K-step k
    u4_t w[2][4];                                               // weight ring: step u in set u & 1
    h8_t bh[2][4], bl[2][4];                                    // pixel fragments of step u in set u & 1
    u4_t xh[4], xl[4];                                          // the residual of the chunk ahead: [pixel block]
    auto issue_w = [&](int u, u4_t (&d)[4]) __attribute__((always_inline)) {             // u in [0, STEPS)
        const int j = u / (KS1 + G2S), r = u % (KS1 + G2S);
        const uint32_t wm = SBBSEG_PROBE(p.dbg & 1) ? 0u : 1u;                 // (timing probe: every request reads the first step's fragments)
        if (r < KS1) wload4(d[0], d[1], d[2], d[3], wlane + wm * (uint32_t)((j * KS1 + r) * 8 * 4096), w3rsrc);
        else if constexpr (MI2 == 2) wload4(d[0], d[1], d[2], d[3], wlane + wm * (uint32_t)((j * G2S + r - KS1) * 8 * 4096), w1rsrc);
        else wload2(d[0], d[1], wlane + wm * (uint32_t)((j * G2S + r - KS1) * 8 * 2048), w1rsrc);
    };
    auto issue_x = [&](int tile, int j) __attribute__((always_inline)) {                  // 8 loads (plain mode: 4)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const uint32_t off = (SBBSEG_PROBE(p.dbg & 4) ? 0xf0000000u : 0u) + xlane + (uint32_t)(tile * 64 + ni * 16) * (uint32_t)PY + (uint32_t)(j * YROW);
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

    // ---- prologue: the first tile's b and its first chunk of x
    issue_b(tile_at(0));
    issue_x(tile_at(0), 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        asm volatile("" : "+v"(xh[ni]));
        if constexpr (X3) asm volatile("" : "+v"(xl[ni]));
    }
    __syncthreads();                                            // b, constants visible
    issue_w(0, w[0]);
    issue_w(1, w[1]);

    for (int it = 0; it < my_tiles; ++it) {
        const int tile = tile_at(it), tile_next = tile_at(it + 1);
        asm volatile("" : "+v"(rot), "+v"(xlane), "+v"(alane), "+v"(wlane), "+v"(dlane), "+v"(frv), "+v"(fgv));
        f4_t acc2[MI2][4];
#pragma unroll
        for (int m = 0; m < MI2; ++m)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc2[m][ni] = (f4_t){0.f, 0.f, 0.f, 0.f};
        f4_t acc1[2][4];
        static_for<0, STEPS>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            constexpr int j = u / (KS1 + G2S), r = u % (KS1 + G2S);
            constexpr bool g1 = r < KS1;
            constexpr int k = g1 ? r : r - KS1;
            constexpr int set = u & 1;
            // loads per weight request of the step after this one (its request is the only one younger than this step's, but for the
            // epilogues in between)
            constexpr int un = (u + 1) % STEPS;
            constexpr int Lnext = (un % (KS1 + G2S)) < KS1 ? 4 : LG2;
            // vector-memory operations issued between this step's weight request (end of step u - 2) and here, beside that one request:
            //   GEMM 2 steps 0, 1: epilogue 1 (+ the b DMA behind the last chunk's)     GEMM 1 steps 0, 1 of chunk 0: the previous tile's epilogue 2
            constexpr int extra = (!g1 && k < 2) ? kEpi1 + (j == NCH - 1 ? KS1 : 0) : 0;
            if constexpr (g1 && k == 0) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc1[m][ni] = (f4_t){0.f, 0.f, 0.f, 0.f};
                load_frags(lds_b, PB, SLB, 0, bh[set], bl[set]);                 // (not requested ahead across the barrier in front of this phase)
            }
            if constexpr (!g1 && k == 0) load_frags(lds_y, YROW, SLY, 0, bh[set], bl[set]);
            // the next step's pixel fragments, inside a phase
            if constexpr (g1 && k + 1 < KS1) load_frags(lds_b, PB, SLB, k + 1, bh[set ^ 1], bl[set ^ 1]);
            if constexpr (!g1 && k + 1 < G2S) load_frags(lds_y, YROW, SLY, k + 1, bh[set ^ 1], bl[set ^ 1]);
            u4_t (&cw)[4] = w[set];
            if constexpr (g1 && j == 0 && k < 2) {
                // (round 6: the previous tile's a' stores are younger than this step's weight request but are NOT allowed for in the count:
                //  a block's FIRST tile has no epilogue in front of it, the count with the stores in was looser than what is in flight there, and
                //  steps 0 / 1 could multiply weights still on their way.  conv3_expand_reduce, which has the same tile top, lost that race once in
                //  ~1e5 tiles when its weight loads missed L2: profiles/r06_experiments.md section 7.  One statement, not an if / else pair of
                //  two asm waits: see there too)
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

            if constexpr (g1 && k == KS1 - 1) {
                // ---- epilogue 1: y = ReLU(s3 * acc + h3 + x) -> hi | lo: to HBM and into the LDS image of the chunk.  x of this chunk was requested
                // an epilogue ago: older than every weight request already waited for (the wait below only ties the registers to that fact)
                if constexpr (X3)
                    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]), "+v"(xl[0]), "+v"(xl[1]), "+v"(xl[2]), "+v"(xl[3])
                                 : "n"(2 * LG2) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]) : "n"(2 * LG2) : "memory");
                const int c0 = j * 256 + wave * 32 + fgv * 8;
                float sc[8], sh[8];
                *(float4*)&sc[0] = *(const float4*)(cst + c0); *(float4*)&sc[4] = *(const float4*)(cst + c0 + 4);
                *(float4*)&sh[0] = *(const float4*)(cst + 4 * C + c0); *(float4*)&sh[4] = *(const float4*)(cst + 4 * C + c0 + 4);
                // the wave's 32 channels = granules wave * 4 + fg of the chunk (split mode: group `wave`, hi granules fg, lo granules 4 + fg)
                // (round 5 timed these writes with a conflict-free rotation -- one slot per pixel, results then wrong -- and found no difference:
                //  profiles/r05_experiments.md section 6; the probe itself made the compiler add a vmcnt wait to the loop and is not kept)
                const int s_hi = (wave * (X3 ? 8 : 4) + rot) & (SLY - 1), s_lo = (wave * 8 + 4 + rot) & (SLY - 1);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    float y[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        y[q] = __builtin_fmaf(acc1[0][ni][q], sc[q], sh[q]);
                        y[4 + q] = __builtin_fmaf(acc1[1][ni][q], sc[4 + q], sh[4 + q]);
                    }
                    const uint32_t off = (SBBSEG_PROBE(p.dbg & 2) ? 0xf0000000u : 0u) + xlane + (uint32_t)(tile * 64 + ni * 16) * (uint32_t)PY + (uint32_t)(j * YROW);
                    char* row = lds_y + (ni * 16 + frv) * YROW;
                    if constexpr (X3) {
                        const h8_t rh = __builtin_bit_cast(h8_t, xh[ni]), rl = __builtin_bit_cast(h8_t, xl[ni]);
#pragma unroll
                        for (int q = 0; q < 8; ++q) y[q] = fmaxf(__fadd_rn(y[q], __fadd_rn((float)rh[q], (float)rl[q])), 0.f);      // (= add_split8 + ReLU, kernels.hip)
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
                        u4_t v;
                        v[0] = pack_h2(y[0], y[1]); v[1] = pack_h2(y[2], y[3]); v[2] = pack_h2(y[4], y[5]); v[3] = pack_h2(y[6], y[7]);
                        asm volatile("buffer_store_dwordx4 %1, %0, %2, 0 offen\n\ts_nop 1" :: "v"(off), "v"(v), "s"(yrsrc) : "memory");
                        *(u4_t*)(row + (s_hi << 4)) = v;
                    }
                }
                // the residual of the chunk after this one (the next tile's first behind the last)
                if constexpr (j + 1 < NCH) issue_x(tile, j + 1);
                else issue_x(tile_next, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                   // the y chunk is complete; (last chunk) every wave is through with b
                asm volatile("" ::: "memory");
                if constexpr (j == NCH - 1) issue_b(tile_next);
            }
            if constexpr (!g1 && k == G2S - 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                   // every wave has taken its last fragment of this y chunk (last chunk: and has waited
                asm volatile("" ::: "memory");                  // for its share of the next tile's b: steps >= 2 of this GEMM)
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
                const uint32_t off = alane + (uint32_t)(tile * 64 + ni * 16) * (uint32_t)PA;
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
                    u4_t v;
                    v[0] = pack_h2(y[0], y[1]); v[1] = pack_h2(y[2], y[3]); v[2] = pack_h2(y[4 % (4 * MI2)], y[5 % (4 * MI2)]); v[3] = pack_h2(y[6 % (4 * MI2)], y[7 % (4 * MI2)]);
                    asm volatile("buffer_store_dwordx4 %1, %0, %2, 0 offen\n\ts_nop 1" :: "v"(off), "v"(v), "s"(arsrc) : "memory");
                } else {
                    u2_t v;
                    v[0] = pack_h2(y[0], y[1]); v[1] = pack_h2(y[2], y[3]);
                    asm volatile("buffer_store_dwordx2 %1, %0, %2, 0 offen\n\ts_nop 1" :: "v"(off), "v"(v), "s"(arsrc) : "memory");
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int C, bool X3> static hipError_t launch_one(const ExpRedParams& p, int num_cus, hipStream_t s)
{
    constexpr int EB = X3 ? 4 : 2;
    constexpr int lds = 64 * C * EB + 64 * 256 * EB + 10 * C * 4;
    static bool attr_done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!attr_done[dev & 63]) {
        e = hipFuncSetAttribute((const void*)expand_reduce<C, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_done[dev & 63] = true;
    }
    const int n_tiles = (p.M + 63) / 64;
    const int grid = n_tiles < num_cus ? n_tiles : num_cus;
    hipLaunchKernelGGL((expand_reduce<C, X3>), dim3(grid), dim3(512), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_expand_reduce_x3(const ExpRedParams& p, int num_cus, hipStream_t s)
{
    if (p.C == 128) return p.x3 ? launch_one<128, true>(p, num_cus, s) : launch_one<128, false>(p, num_cus, s);
    if (p.C == 256) return p.x3 ? launch_one<256, true>(p, num_cus, s) : launch_one<256, false>(p, num_cus, s);
    return hipErrorInvalidValue;
}

}  // namespace sbbseg
