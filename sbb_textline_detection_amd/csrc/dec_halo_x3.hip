// dec_halo_x3.hip -- the decoder conv at 224 x 224 ("dec4": 3x3 conv over [up2(src0: 128 ch @ 112 x 112), skip: 64 ch @ 224 x 224] -> 64 ch,
// BN / ReLU) with its source halos RESIDENT in LDS, split-fp16 mode (kF16X3), round 4.
//
// conv_igemm_mfma runs this layer as a grouped launch of the four output-parity classes on 256 x 64 tiles: every K-step re-stages
// its 256 pixel rows, the nine skip taps and four src0 taps re-read the same lines one to nine K-steps apart, and with 64 blocks per
// XCD streaming ~3 MB per K-step through a 4 MB L2 most of those re-reads miss: PMC FETCH 11.2 GB per 140-patch launch against
// 2.7 GB of unique input, 4.45 TB/s -- HBM-bound on tap re-fetches (profiles/r03_x3_pmc_per_op.txt).
//
// Here a block owns a 16 x 16 output tile = the 8 x 8 grids of all four parity classes:
//   * LDS holds the 10 x 10 src0 halo (as two pieces: channel groups 0-1 and 2-3, 25 KB each) and the 18 x 18 skip halo (81 KB):
//     every source byte is fetched once per tile (x 1.27 halo overlap), all taps and all four classes read it from LDS
//   * eight waves, wave = (parity class, half of the 64 output channels): wave tile 64 pixels x 32 channels
//   * the K loop walks the SAME K-steps in the SAME order as the generic kernel (source, 32-channel group, tap -- the tap order of
//     each class comes from its K-step records) with the same three MFMAs per product (lo*hi, hi*lo, hi*hi) and the same epilogue
//     arithmetic: bit-identical outputs (tests/test_gpu_parity.py)
//   * weights are streamed: per K-step a wave loads its 2 x (hi, lo) A fragments (4 x 16 B per lane, repacked on the host into
//     fragment order: one contiguous KB per wave-instruction) straight into registers, two K-steps ahead of their MFMAs
//   * the three LDS pieces are a ring over the persistent tile loop: as soon as every wave has taken its last fragment from a piece
//     (K-step 8, 16, 34) the piece is refilled with the NEXT tile's halo (LDS-DMA), which then has 3/4 of a tile to land
//   * every vector-memory LOAD of the loop -- weight fragments and halo DMA -- is issued from inline asm and waited for by hand
//     (vmcnt retires in issue order: the counted wait in front of a K-step's MFMAs covers the step's weights AND every older DMA;
//     a piece is first read two barriers after its DMA, one of them behind that wait).  With builtins the compiler drains the queue
//     (vmcnt(0)) after every DMA burst: an LDS-DMA in flight beside ordinary register loads is waited for conservatively at the next
//     LDS access / control-flow join.  Every wave issues the SAME number of loads per step (the DMA pieces are dealt 4 / 4 / 11 per
//     wave, duplicates where 25 / 81 do not divide; the last tile re-issues its own halo), so the counts are constants
//   * pixel rows are 256 B (16 granule slots); granule G of halo pixel (hy, hx) sits at slot (G + 2 hx) & 15 (src0) / (G + hx) & 15
//     (skip, whose fragments take every second pixel): conflict-free for the 16-lane groups of ds_read_b128
#include "internal.h"

namespace sbbseg {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

constexpr int kS0Px = 100, kSkPx = 324;                 // halo pixels: 10 x 10, 18 x 18
constexpr int kS0Instr = 25, kSkInstr = 81;             // wave-instructions of 4 pixels x 256 B
constexpr int kS0Bytes = kS0Instr * 1024;               // one src0 piece (two channel groups of every halo pixel)
constexpr int kSkBytes = kSkInstr * 1024;
constexpr int kDecHaloLdsBytes = 2 * kS0Bytes + kSkBytes;      // 134 144
constexpr int kSteps = 34;                              // 4 groups x 4 taps of src0 + 2 groups x 9 taps of the skip

template <int N> struct IC { static constexpr int value = N; };
// f(IC<B>{}), f(IC<B + 1>{}), ... f(IC<E - 1>{}): the K-step index is a compile-time constant in every copy (register sets are
// selected by t & 1; `#pragma unroll` left the 34-step loop rolled and the sets in scratch memory)
template <int B, int E, class F> __device__ __attribute__((always_inline)) inline void static_for(F&& f)
{
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ inline f4_t mma(h8_t a, h8_t b, f4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

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

typedef __attribute__((ext_vector_type(4))) unsigned u4_t;

// four weight fragments (1 KB apart) of one K-step: 16 bytes per lane each, destination registers valid after wait_w
__device__ __attribute__((always_inline)) inline void wload4(u4_t& a, u4_t& b, u4_t& c, u4_t& d, uint32_t voff, u4_t rsrc)
{
    asm volatile("buffer_load_dwordx4 %0, %4, %5, 0 offen\n\t"
                 "buffer_load_dwordx4 %1, %4, %5, 0 offen offset:1024\n\t"
                 "buffer_load_dwordx4 %2, %4, %5, 0 offen offset:2048\n\t"
                 "buffer_load_dwordx4 %3, %4, %5, 0 offen offset:3072"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(voff), "s"(rsrc) : "memory");
}
// all but the youngest N vector-memory operations of this wave have completed; ties the four registers to the wait
template <int N> __device__ __attribute__((always_inline)) inline void wait_w(u4_t& a, u4_t& b, u4_t& c, u4_t& d)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
// one LDS-DMA wave-instruction hidden from the compiler: lane l's 16 bytes at gsrc(l) land at lds_dst + 16 l (M0 written in the statement that reads it)
__device__ __attribute__((always_inline)) inline void glds16_hidden(const void* gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

}  // namespace

__global__ __launch_bounds__(512, 2) void dec_halo_x3(const DecHaloParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const lds_sk = smem + 2 * kS0Bytes;                  // (the src0 pieces sit at smem + pc * kS0Bytes)
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(LDS_AS char*)smem);      // LDS byte address of smem

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cls = wave >> 1, mh = wave & 1;                  // parity class, half of the output channels
    const int py = cls >> 1, px = cls & 1;
    const int frow = lane & 15, fg = lane >> 4;

    const int H = 2 * p.PH, W = 2 * p.PW;
    const int tiles_x = W / 16, tiles_y = H / 16;
    const int tiles_per_patch = tiles_x * tiles_y;
    // owned-region launch (DecHaloParams::ttab, region.h): the tiles are the table's entries -- (patch, output origin / 2) -- instead of the
    // 14 x 14 grid of every patch; an origin is any even pixel, tiles of one patch may overlap at its far edge (same values twice)
    const int n_tiles = p.ttab ? p.n_tab : p.n * tiles_per_patch;
    const __attribute__((address_space(4))) uint32_t* ttab = (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)p.ttab;
    // XCD-contiguous walk (neighbouring tiles share halo lines in one L2)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, GX = gridDim.x >> 3;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int xcd_lo = xcd * per_xcd, xcd_hi = min(n_tiles, xcd_lo + per_xcd);
    const int my_tiles = xcd_lo + slot < xcd_hi ? (xcd_hi - xcd_lo - slot + GX - 1) / GX : 0;
    if (my_tiles <= 0) return;
    auto tile_at = [&](int it) __attribute__((always_inline)) -> int { return xcd_lo + slot + (it < my_tiles ? it : my_tiles - 1) * GX; };

    // ---- halo DMA: tile-invariant part of this lane's pieces.  Instruction k of a piece fills halo pixels 4 k .. 4 k + 3 (256 B each);
    // lane l writes slot l & 15 of pixel 4 k + (l >> 4), i.e. fetches granule (slot - rot(hx)) & 15 of it.  wave w issues k = w, w + 8, ...
    // -- 4 instructions of a src0 piece, 11 of the skip halo; where 25 / 81 do not divide, the surplus repeats instruction k - 25 /
    // k - 81 (same bytes to the same place): every wave issues the same number of loads
    int s0_hyx[4], sk_hyx[11];                                 // hy | hx << 8 | source granule << 16 | instruction << 24
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        int k = wave + 8 * m;
        k = k < kS0Instr ? k : k - kS0Instr;
        const int hp = 4 * k + (lane >> 4);
        const int hy = hp / 10, hx = hp - hy * 10;
        s0_hyx[m] = hy | (hx << 8) | ((((lane & 15) - 2 * hx) & 15) << 16) | (k << 24);
    }
#pragma unroll
    for (int m = 0; m < 11; ++m) {
        int k = wave + 8 * m;
        k = k < kSkInstr ? k : k - kSkInstr;
        const int hp = 4 * k + (lane >> 4);
        const int hy = hp / 18, hx = hp - hy * 18;
        sk_hyx[m] = hy | (hx << 8) | ((((lane & 15) - hx) & 15) << 16) | (k << 24);
    }
    // tile -> patch and origin (y0, x0) of its 16 x 16 outputs (wave-uniform: the table entry comes through the scalar cache)
    auto tile_coords = [&](int tile, int& n, int& y0, int& x0) __attribute__((always_inline)) {
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
    // src0 piece `pc` (channel groups 2 pc, 2 pc + 1 = bytes 256 pc .. of the 512-byte stored pixel): 4 loads per wave
    auto issue_s0 = [&](int n, int y0, int x0, int pc) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int e = s0_hyx[m];
            const int Y = (y0 >> 1) - 1 + (e & 255), X = (x0 >> 1) - 1 + ((e >> 8) & 255);
            const bool ok = (unsigned)Y < (unsigned)p.PH && (unsigned)X < (unsigned)p.PW;
            const uint32_t off = ok ? (uint32_t)((n * p.PH + Y) * p.PW + X) * 512u + (uint32_t)(pc * 256 + ((e >> 16) & 15) * 16 + kZeroHeaderBytes) : 0u;
            const uint32_t dst = lds0 + (uint32_t)(pc * kS0Bytes) + (uint32_t)__builtin_amdgcn_readfirstlane(e >> 24) * 1024u;
            glds16_hidden(p.src0 + off, dst);
        }
    };
    auto issue_sk = [&](int n, int y0, int x0) __attribute__((always_inline)) {        // 11 loads per wave
#pragma unroll
        for (int m = 0; m < 11; ++m) {
            const int e = sk_hyx[m];
            const int Y = y0 - 1 + (e & 255), X = x0 - 1 + ((e >> 8) & 255);
            const bool ok = (unsigned)Y < (unsigned)H && (unsigned)X < (unsigned)W;
            const uint32_t off = ok ? (uint32_t)((n * H + Y) * W + X) * 256u + (uint32_t)(((e >> 16) & 15) * 16 + kZeroHeaderBytes) : 0u;
            const uint32_t dst = lds0 + (uint32_t)(2 * kS0Bytes) + (uint32_t)__builtin_amdgcn_readfirstlane(e >> 24) * 1024u;
            glds16_hidden(p.skip + off, dst);
        }
    };

    // ---- fragment addressing.  Lane (frow, fg) of pixel block ni holds class-grid pixel (i, j) = (2 ni + (frow >> 3), frow & 7).
    //   src0 K-step (group g, tap (dy, dx)):  halo pixel (i + dy + 1, j + dx + 1), granule 8 (g & 1) + 4 lo + fg, rotation 2 hx
    //   skip K-step (group g, tap (dy, dx)):  halo pixel (2 i + dy + 1, 2 j + dx + 1), granule 8 g + 4 lo + fg, rotation hx
    // The class's 4 + 9 taps come from p.taps (wave-uniform, scalar loads); everything that depends on ni is an immediate offset.
    const int i0 = frow >> 3, j0 = frow & 7;
    int a0 = (i0 + 1) * 10 + j0 + 1, r0 = fg + 2 * (j0 + 1);                       // src0: pixel index / slot without the tap
    int a1 = (2 * i0 + 1) * 18 + 2 * j0 + 1, r1 = fg + 2 * j0 + 1;                 // skip
    int taps[13];                                              // wave-uniform: the class's 4 + 9 taps, (dy & 255) | (dx & 255) << 8
    {
        const __attribute__((address_space(4))) int* tp = (const __attribute__((address_space(4))) int*)(uintptr_t)(p.taps + cls * 16);
#pragma unroll
        for (int k = 0; k < 13; ++k) taps[k] = tp[k];
    }
    // K-step t -> address of the hi / lo fragment of pixel block 0
    auto frag_addr = [&](int t, const char*& ah, const char*& al) __attribute__((always_inline)) {
        if (t < 16) {
            const int g = t >> 2, tp = taps[t & 3];
            const int dy = (tp << 24) >> 24, dx = (tp << 16) >> 24;             // signed bytes
            const int hp = a0 + dy * 10 + dx;
            const int sh = (r0 + 2 * dx + 8 * (g & 1)) & 15;
            const char* base = smem + (g >> 1) * kS0Bytes + hp * 256;
            ah = base + (sh << 4);
            al = base + (((sh + 4) & 15) << 4);
        } else {
            const int u = t - 16, g = u >= 9 ? 1 : 0, tp = taps[4 + (u - 9 * g)];
            const int dy = (tp << 24) >> 24, dx = (tp << 16) >> 24;
            const int hp = a1 + dy * 18 + dx;
            const int sh = (r1 + dx + 8 * g) & 15;
            const char* base = lds_sk + hp * 256;
            ah = base + (sh << 4);
            al = base + (((sh + 4) & 15) << 4);
        }
    };
    // weights: wfrag = [class][K-step][row block mi 0..3][hi | lo][64 lanes x 16 B]; this wave's row blocks are 2 mh, 2 mh + 1: four
    // fragments (m = 0: hi, lo; m = 1: hi, lo) 1 KB apart, 8 KB per K-step
    u4_t wrsrc;
    {
        const uint64_t base = (uint64_t)(uintptr_t)p.wfrag + (uint64_t)(cls * kSteps * 8 + 2 * mh * 2) * 1024u;
        wrsrc[0] = __builtin_amdgcn_readfirstlane((uint32_t)base);
        wrsrc[1] = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32) & 0xffffu);
        wrsrc[2] = (uint32_t)(kSteps * 8 * 1024);
        wrsrc[3] = 0x00020000u;
    }
    uint32_t wlane = (uint32_t)lane * 16u;

    // epilogue constants of this lane's 8 channels (c0 = 32 mh + 8 fg): scale * 2^-s of THIS class's weight pre-scale, shift
    const int c0 = mh * 32 + fg * 8;
    float ksc[8], ksh[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { ksc[q] = p.scale[c0 + q] * p.wmul[cls]; ksh[q] = p.shift[c0 + q]; }
    // (a use the compiler can see: it waits for these loads HERE.  Left to the first use in the epilogue, its wait -- vmcnt(0), it cannot
    // count the asm loads in between -- drained the hand-counted queue once per tile, right behind the skip halo's DMA burst)
#pragma unroll
    for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(ksc[q]), "+v"(ksh[q]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the compiler's own loads above are done before the hand-counted queue starts)

    // ---- prologue: the first tile's halos
    int cn, cy0, cx0, nn, ny0, nx0;                            // this tile, the next one (decoded once per tile)
    tile_coords(tile_at(0), cn, cy0, cx0);
    issue_s0(cn, cy0, cx0, 0);
    issue_s0(cn, cy0, cx0, 1);
    issue_sk(cn, cy0, cx0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    u4_t w[2][4];                                              // weight ring: K-step t in set t & 1 (34 is even: consistent across tiles); [m * 2 + lo]
    h8_t bh[2][4], bl[2][4];                                   // pixel fragments of K-step t in set t & 1
    auto load_b = [&](int t, h8_t (&dh)[4], h8_t (&dl)[4]) __attribute__((always_inline)) {
        const char *ah, *al;
        frag_addr(t, ah, al);
        const int nstride = t < 16 ? 20 * 256 : 72 * 256;      // pixel block ni + 1 = two class rows down
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            dh[ni] = *(const h8_t*)(ah + ni * nstride);
            dl[ni] = *(const h8_t*)(al + ni * nstride);
        }
    };
    wload4(w[0][0], w[0][1], w[0][2], w[0][3], wlane + 0 * 8192u, wrsrc);
    wload4(w[1][0], w[1][1], w[1][2], w[1][3], wlane + 1 * 8192u, wrsrc);
    load_b(0, bh[0], bl[0]);

    for (int it = 0; it < my_tiles; ++it) {
        tile_coords(tile_at(it + 1), nn, ny0, nx0);
        // the fragment addresses of all 34 K-steps are tile-invariant: keep the compiler from hoisting 68 address registers (and as
        // many scalars) out of the tile loop -- they are recomputed per step (~10 VALU / SALU ops beside 24 MFMAs)
        asm volatile("" : "+v"(a0), "+v"(r0), "+v"(a1), "+v"(r1), "+v"(wlane));
#pragma unroll
        for (int k = 0; k < 13; ++k) asm volatile("" : "+s"(taps[k]));
        f4_t acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[m][ni] = (f4_t){0.f, 0.f, 0.f, 0.f};
        static_for<0, kSteps>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            // pixel fragments of the next K-step (the first one of the next tile behind the last; the last tile re-reads its own): requested
            // before this step's MFMAs
            load_b((t + 1) % kSteps, bh[(t + 1) & 1], bl[(t + 1) & 1]);
            // This step's weights (loaded two steps ago) have landed when all but the younger loads of this wave are done.  Younger, in
            // issue order: a DMA burst at the end of step t - 2, the 4 weight loads of step t - 1, a DMA burst at the end of step t - 1,
            // and -- behind the burst of step 33 -- the 8 stores of the epilogue.  Bursts: 4 loads after steps 7 and 15, 11 after step 33.
            // (First tile: the prologue's bursts were drained, nothing but the 4 loads of step t - 1 is younger.)
            u4_t (&cw)[4] = w[t & 1];
            if constexpr (t == 0 || t == 1) {
                if (it == 0) wait_w<4>(cw[0], cw[1], cw[2], cw[3]);
                else wait_w<4 + 11 + 8>(cw[0], cw[1], cw[2], cw[3]);
            } else if constexpr (t == 8 || t == 9 || t == 16 || t == 17) {
                wait_w<4 + 4>(cw[0], cw[1], cw[2], cw[3]);
            } else {
                wait_w<4>(cw[0], cw[1], cw[2], cw[3]);
            }
            const h8_t ch[2] = {__builtin_bit_cast(h8_t, cw[0]), __builtin_bit_cast(h8_t, cw[2])};
            const h8_t cl[2] = {__builtin_bit_cast(h8_t, cw[1]), __builtin_bit_cast(h8_t, cw[3])};
            const h8_t (&xh)[4] = bh[t & 1];
            const h8_t (&xl)[4] = bl[t & 1];
            // three sweeps, small terms first (the order conv_igemm_mfma's split loop uses per accumulator)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[m][ni] = mma(cl[m], xh[ni], acc[m][ni]);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[m][ni] = mma(ch[m], xl[ni], acc[m][ni]);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[m][ni] = mma(ch[m], xh[ni], acc[m][ni]);
            // this set's weights are spent: K-step t + 2 (of the next tile behind the last two; always issued: the counts stay constant)
            wload4(cw[0], cw[1], cw[2], cw[3], wlane + (uint32_t)(((t + 2) % kSteps) * 8192), wrsrc);
            // piece boundaries: every wave has taken its last fragment from the piece (the reads of step t + 1 above belong to the NEXT
            // piece) -> refill it with the next tile's halo (the last tile re-issues its own: constant counts)
            if constexpr (t == 7 || t == 15 || t == kSteps - 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (t == 7) issue_s0(nn, ny0, nx0, 0);
                else if constexpr (t == 15) issue_s0(nn, ny0, nx0, 1);
                else issue_sk(nn, ny0, nx0);
            }
            __builtin_amdgcn_sched_barrier(0);                  // (keeps a step's loads from being scheduled many steps early: register pressure)
        });

        // ---- epilogue: y = ReLU(scale * acc + shift) -> hi | lo, 16 bytes each per pixel and 8-channel group (as conv_igemm_mfma's split
        // path).  EXACTLY 8 store instructions per wave: the wait counts of steps 0 and 1 include them.
        const int n = cn;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int oy = cy0 + 2 * (2 * ni + i0) + py, ox = cx0 + 2 * j0 + px;
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
            h8_t vh, vl;
            split8(y, vh, vl);
            uint16_t* dst = (uint16_t*)p.out + ((size_t)(n * H + oy) * W + ox) * 128 + mh * 64 + fg * 8;      // channel group mh: [32 hi][32 lo]
            asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dwordx4 %0, %2, off offset:64\n\ts_nop 1"
                         :: "v"(dst), "v"(vh), "v"(vl) : "memory");
        }
        cn = nn; cy0 = ny0; cx0 = nx0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the surplus loads of the last tile land before the registers / LDS go away
}

hipError_t launch_dec_halo_x3(const DecHaloParams& p, int num_cus, hipStream_t s)
{
    static bool attr_done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!attr_done[dev & 63]) {
        e = hipFuncSetAttribute((const void*)dec_halo_x3, hipFuncAttributeMaxDynamicSharedMemorySize, kDecHaloLdsBytes);
        if (e != hipSuccess) return e;
        attr_done[dev & 63] = true;
    }
    const int n_tiles = p.ttab ? p.n_tab : p.n * (p.PH / 8) * (p.PW / 8);
    if (n_tiles <= 0) return hipSuccess;
    const int grid = ((n_tiles < num_cus ? n_tiles : num_cus) + 7) & ~7;
    hipLaunchKernelGGL(dec_halo_x3, dim3(grid), dim3(512), kDecHaloLdsBytes, s, p);
    return hipGetLastError();
}

}  // namespace sbbseg
