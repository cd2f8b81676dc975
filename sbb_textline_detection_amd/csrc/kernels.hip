// kernels.hip -- gfx950 (MI355X, CDNA4) device code of libsbbseg.
//
// Hot kernel: conv_igemm_mfma -- im2col-free implicit-GEMM convolution on MFMA.
//   D[channel][pixel] = sum_k W[channel][k] * X[pixel][k]     (v_mfma_f32_16x16x32_{f16,bf16}, fp32 acc)
//   * the contraction axis k walks (source, channel group, tap) in 16-byte granules; nearest x2 upsampling, channel
//     concat of two sources, zero padding and the one_side_pad shift are address arithmetic in the gather -- no tensor
//     is materialised
//   * operands go HBM/L2 -> LDS with 16-byte LDS-DMA loads (lane-linear LDS image, XOR-swizzled through the *source*
//     granule choice), double buffered: `buffer_load ... lds` with the tap displacement in the scalar offset and
//     out-of-bounds taps zero-filled by the hardware (fast gather), or `global_load_lds` with per-lane address
//     arithmetic (plain gather: irregular K-steps, upsampling sources, short-K layers)
//   * MFMA "A" operand = weights, "B" = pixels, so a lane ends up with consecutive channels of one pixel: 16-byte NHWC
//     epilogue stores; BN scale/shift, residual add and ReLU are applied in fp32 registers
//   * X3 = the split-fp16 (label-exact) mode: hi + lo operands, three MFMAs per product.
// Direct kernels on LDS halo tiles where an output tile has few channels: stem_conv_pairs, conv3x3_c64_direct(+_x3),
// dec_tail_fused(+_x3); bottleneck_fused = a whole stage-2 ResNet block per launch.
// Everything else here (ingest, max-pool, head, stitch, resize, Otsu, morphology, components, deskew profiles) is
// HBM-bound byte and integer work.
#include "internal.h"

namespace sbbseg {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = one 16-byte granule
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t; // 8 fp16
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

__host__ __device__ inline uint16_t bf16_bits_rne(float f)
{
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
uint16_t f32_to_bf16_rne(float f) { return bf16_bits_rne(f); }
float bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; return __builtin_bit_cast(float, u); }
uint16_t f32_to_f16_rne(float f)
{
    f = f > 65504.f ? 65504.f : (f < -65504.f ? -65504.f : f);
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
}

__device__ inline float bf16_lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ inline float bf16_hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
__device__ inline uint32_t pack_bf16x2(float a, float b)
{
    uint32_t r;                                   // gfx950 packed RNE convert (no builtin)
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// fp16 twins (SBBSEG_PREC_F16): saturate instead of overflowing to inf, round to nearest even
__device__ inline uint32_t pack_f16x2(float a, float b)
{
    a = fminf(fmaxf(a, -65504.f), 65504.f);
    b = fminf(fmaxf(b, -65504.f), 65504.f);
    f16x2_t v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ inline float f16_lo(uint32_t v) { return (float)__builtin_bit_cast(f16x2_t, v)[0]; }
__device__ inline float f16_hi(uint32_t v) { return (float)__builtin_bit_cast(f16x2_t, v)[1]; }

template <bool F16> __device__ inline uint32_t pack2(float a, float b) { return F16 ? pack_f16x2(a, b) : pack_bf16x2(a, b); }
template <bool F16> __device__ inline float unpack_lo(uint32_t v) { return F16 ? f16_lo(v) : bf16_lo(v); }
template <bool F16> __device__ inline float unpack_hi(uint32_t v) { return F16 ? f16_hi(v) : bf16_hi(v); }
// One LDS-DMA wave-instruction the compiler does not see (lane l's 16 bytes at gsrc(l) land at lds_dst + 16 l; M0 is written in the
// statement that reads it).  After the BUILTIN the compiler waits vmcnt(0) in front of the next LDS access it cannot tell apart from the
// DMA's destination -- in stem_conv_pairs that was the epilogue-constant read in the middle of a tile, i.e. the NEXT tile's halo, just
// requested, had to land there (0.260 -> 0.252 ms per 140 patches; the same change measured nothing on dec_tail_fused, two blocks per CU,
// which keeps the builtin).  A kernel that uses this waits for its DMA by hand (counted s_waitcnt at the top of a tile).
__device__ __attribute__((always_inline)) inline void glds16_hidden(const void* gsrc, const char* lds_dst)
{
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const LDS_AS char*)lds_dst);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

template <bool F16> __device__ inline f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c)
{
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---- split mode (kF16X3) element helpers: v = hi + lo, hi = fp16(v) (saturating), lo = fp16(v - hi)
__device__ inline void split_f32(float v, _Float16& hi, _Float16& lo)
{
    v = fminf(fmaxf(v, -65504.f), 65504.f);
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);               // exact difference (Sterbenz-like: |v - hi| <= ulp(hi)/2), then one rounding
}
// 8 consecutive channels of one pixel: hi halves at dst, lo halves `plane` elements behind
__device__ inline void store_split8(uint16_t* dst, int plane, const float (&y)[8])
{
    f16x8_t h, l;
#pragma unroll
    for (int q = 0; q < 8; ++q) { _Float16 a, b; split_f32(y[q], a, b); h[q] = a; l[q] = b; }
    *(f16x8_t*)dst = h;
    *(f16x8_t*)(dst + plane) = l;
}
__device__ inline void add_split8(const uint16_t* src, int plane, float (&y)[8])
{
    const f16x8_t h = *(const f16x8_t*)src, l = *(const f16x8_t*)(src + plane);
#pragma unroll
    for (int q = 0; q < 8; ++q) y[q] = __fadd_rn(y[q], __fadd_rn((float)h[q], (float)l[q]));      // hi + lo is exact in fp32; one rounding, never fused
}

// ------------------------------------------------------------------------------------------------
// conv_igemm_mfma  (16-bit operands: bf16 or fp16, fp32 accumulate)
// ------------------------------------------------------------------------------------------------
// blocks per CU a tile family is sized for (LDS budget) -> min waves per SIMD for __launch_bounds__
constexpr int conv_blocks_per_cu(int bp, int bc, int wp, int wc, int ns)
{
    const int waves = wp * wc;
    const int wrows = bc > 8 * waves ? bc : 8 * waves;
    const int lds = ns * (bp + wrows) * 128;
    if (waves != 4) return 1;
    return 3 * lds <= 160 * 1024 ? 3 : (2 * lds <= 160 * 1024 ? 2 : 1);
}

template <int BP, int BC, int WP, int WC, int NS, int GS = 8>
struct ConvTile {
    // GS = 16-byte granules of the contraction axis per LDS stage: 8 (a whole 64-element K-step) or 4
    // (half a K-step: twice as many, half as big stages -> deeper prefetch in the same LDS)
    static constexpr int kWaves = WP * WC;
    static constexpr int kThreads = 64 * kWaves;
    static constexpr int kRowBytes = GS * 16;
    static constexpr int kRowsPerInstr = 64 / GS;             // rows one global_load_lds wave-instruction fills
    static constexpr int kStagesPerKStep = 8 / GS;
    static constexpr int kWPX = BP / WP;          // pixels per wave tile
    static constexpr int kWCH = BC / WC;          // channels per wave tile
    static constexpr int kNI = kWPX / 16;
    static constexpr int kMI = kWCH / 16;
    static constexpr int kPLoads = BP / (kRowsPerInstr * kWaves);   // global_load_lds per thread per stage, pixels
    static constexpr int kWRows = BC > kRowsPerInstr * kWaves ? BC : kRowsPerInstr * kWaves;   // weight rows staged (>= one
                                                  // row group per wave, so every wave issues the same number of loads)
    static constexpr int kWLoads = kWRows / (kRowsPerInstr * kWaves);   // ... weights
    static constexpr int kLoads = kPLoads + kWLoads;
    static constexpr int kStageBytes = (BP + kWRows) * kRowBytes;
    static constexpr int kLdsBytes = NS * kStageBytes;
    static constexpr int kBlocksPerCU = (kWaves == 4 && 3 * kLdsBytes <= 160 * 1024) ? 3 : (kWaves == 4 && 2 * kLdsBytes <= 160 * 1024) ? 2 : 1;
    static_assert(GS == 8 || GS == 4, "stage = whole or half K-step");
    static_assert(BP % (kRowsPerInstr * kWaves) == 0 && kWRows % (kRowsPerInstr * kWaves) == 0, "tile rows must split over the waves");
    static_assert(kMI % 2 == 0, "epilogue pairs MFMA row blocks");
};

// blocks per CU a tile family is sized for (LDS budget) -> min waves per SIMD for __launch_bounds__
constexpr int conv_blocks_per_cu(int bp, int bc, int wp, int wc, int ns, int gs)
{
    const int waves = wp * wc;
    const int rpi = 64 / gs;
    const int wrows = bc > rpi * waves ? bc : rpi * waves;
    const int lds = ns * (bp + wrows) * gs * 16;
    if (waves != 4) return 1;
    return 3 * lds <= 160 * 1024 ? 3 : (2 * lds <= 160 * 1024 ? 2 : 1);
}

// value of lane (l ^ 8) inside each row of 16 lanes (DPP row_ror:8)
__device__ inline uint32_t row_ror8(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);
}

// one `buffer_load_dwordx4 ... offen lds`: lane l's 16 bytes at base + voff(l) + soff land at lds + 16 * l; a lane whose
// voff + soff reaches past nrec gets zeros (the resource words are wave-uniform and hoisted out of the loops)
__device__ inline void buffer_load_lds16(const void* base, uint32_t nrec, LDS_AS void* lds, uint32_t voff, uint32_t soff)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)nrec, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, 16, voff, soff, 0, 0);
}

// n / d for n < 2^31 with a host-made (magic, shift) pair (FastDiv, internal.h): one v_mul_hi_u32 + shift instead of the
// ~40-instruction division sequence -- setup_rows divides twice per staged row, which on short-K tiles rivals the MFMA time
__device__ inline int fast_div(int n, uint32_t magic, uint32_t shift)
{
    const uint32_t q = magic ? __umulhi((uint32_t)n, magic) : (uint32_t)n;
    return (int)(q >> shift);
}

template <int N> __device__ inline void wait_vmcnt()
{
    // (the counter field holds 0..63: a larger count cannot be expressed -> drain)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 0 : N) : "memory");
}

// Pipeline: NS LDS stages, stage s+D (D = NS-1) is issued while stage s is multiplied.  Blocks are
// PERSISTENT: block b walks tiles b, b+G, b+2G, ... and the stage stream runs straight across tile
// boundaries, so the first K-step of the next tile is in flight while the current tile finishes its
// MFMAs and runs its epilogue -- short-K layers (1x1 convs, K = 64..512) otherwise spend most of
// their time filling and draining a 1-4 step pipeline.
// K-step descriptors come through the scalar cache (uniform address in the constant address space
// -> s_load, lgkmcnt): no VGPR-destination VMEM load sits in the steady-state loop.
// X3 = the split mode (internal.h, kF16X3): a K-step is 32 channels of one tap staged as slots 0-3 = "hi" halves,
// slots 4-7 = "lo" halves (same LDS image, same staging code: only the source offset of slots 4-7 differs, SrcDesc::lo_off);
// three MFMAs per product; outputs are split again in the epilogue and stored as [C hi][C lo] per pixel.
// FG = the fast gather (ConvParams::fast_gather): every staged row keeps a per-source byte offset and a bit mask of its
// out-of-bounds taps, set up once per tile; a K-step's tap/channel displacement is wave-uniform and rides in the buffer
// load's scalar offset, an out-of-bounds tap sets bit 31 of the lane offset (past num_records -> the load writes zeros
// to LDS, tools/probes/buffer_lds_oob_probe.hip).  Two VALU ops per load instead of ~18: on the long-K tiles the
// address arithmetic of the plain gather costs 20-25 % of the loop (tools/probes/mfma_loop_probe.hip).
// KS = split-K (ConvParams::ks_shift): tile index = tile * 2^ks_shift + split; split s walks K-steps [s * nt, (s + 1) * nt) of the tile and stores
// fp32 partial sums instead of running the epilogue.  A launch of 2-32 tiles of 100-400 K-steps (one patch) leaves most CUs idle for ~1 us per
// K-step; results of a split launch differ from the unsplit one in the last bits (association), so only the whole-image branch uses it.
template <int BP, int BC, int WP, int WC, int NS, bool F16, int GS = 8, bool PH8 = false, bool X3 = false, bool FG = false, bool KS = false>
__global__ __launch_bounds__(64 * WP * WC, (conv_blocks_per_cu(BP, BC, WP, WC, NS, GS) * (WP * WC) / 4))
void conv_igemm_mfma(const ConvParams p)
{
    static_assert(!X3 || (F16 && GS == 8 && !PH8), "split mode: fp16 halves, whole-K-step stages, plain loop");
    static_assert(!KS || (FG && BP == 128 && BC == 128 && NS == 2 && GS == 8 && !PH8), "split-K: the 128 x 128 tile on the fast gather");
    static_assert(!FG || !PH8, "the 8-phase schedule keeps the plain gather");
    constexpr int PL = X3 ? 2 : 1;                       // 16-bit planes per stored activation element
    using T = ConvTile<BP, BC, WP, WC, NS, GS>;
    constexpr int RPI = T::kRowsPerInstr, RB = T::kRowBytes, SPK = T::kStagesPerKStep;
    constexpr int NW = T::kWaves;
    constexpr int D = NS - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave / WC, wc = wave % WC;

    const int n_ct = (p.cout + BC - 1) / BC;
    const int n_pt1 = (p.M + BP - 1) / BP;              // pixel tiles of ONE placement class
    // class-minor order (small weights): the classes of one pixel tile are adjacent and, with the XCD-
    // grouped walk, on the same XCD -- they read the same source pixels, which then come from one L2.
    // Pixel tiles are padded to a multiple of 8 per class there (padding tiles are fully masked).
    const int n_pt1e = p.cls_minor ? (n_pt1 + 7) & ~7 : n_pt1;
    const int n_tiles = (p.n_cls * n_ct * n_pt1e) << (KS ? p.ks_shift : 0);
    const int G = gridDim.x;
    const int nt_full = p.total_ksteps;
    const int nt = KS ? nt_full >> p.ks_shift : nt_full;       // K-steps this block walks per tile
    auto split_of = [&](int tile) __attribute__((always_inline)) -> int { return KS ? tile & ((1 << p.ks_shift) - 1) : 0; };
    // Tile walk of this (persistent) block.  map 0: tiles b, b+G, ... (channel tile fastest): every
    // XCD keeps ONE channel-tile's weight slab hot -- right when the weights dwarf the L2.
    // map 1 (small weight matrices): XCD x = b % 8 owns pixel tiles x, x+8, ...; its blocks walk them
    // channel tile fastest, so the n_ct channel tiles of one pixel tile run on the SAME XCD back to
    // back and the pixel operand is fetched into that L2 once instead of once per XCD.
    const int n_pt = p.n_cls * n_pt1e;                  // "extended" pixel tiles (class x pixel tile)
    auto decode = [&](int tile, int& ctile, int& cls, int& ptile) __attribute__((always_inline)) {
        if constexpr (KS) tile >>= p.ks_shift;
        const int e = fast_div(tile, p.nct_magic, p.nct_shift);
        ctile = tile - e * n_ct;
        if (p.cls_minor) {                              // (n_cls is 1, 2 or 4)
            const int lg = 3 + (p.n_cls >> 1);
            const int grp = e >> lg, r = e & ((1 << lg) - 1);
            cls = r >> 3;
            ptile = grp * 8 + (r & 7);
        } else if (p.n_cls > 1) {
            cls = e / n_pt1;
            ptile = e - cls * n_pt1;
        } else {
            cls = 0;
            ptile = e;
        }
    };
    // map 2 (short-K layers): as map 1, but every block owns a CONTIGUOUS run of its XCD's list, so the
    // channel tiles of one pixel tile run back to back in the SAME block.  Blocks of a short-K layer
    // march in lockstep; under map 1 the sibling blocks miss on the same pixel rows at the same moment
    // and the rows are fetched from HBM once per channel tile (PMC: fetch = n_ct x the input tensor).
    // map 3 (grouped launches): XCD x owns a CONTIGUOUS range of the extended pixel tiles (whole groups of 8 pixel tiles x
    // classes), its blocks walk that range together: the four classes of a pixel tile AND its neighbours above / below
    // meet in one L2, so the halo rows between consecutive pixel tiles are fetched from HBM once, not once per XCD.
    const bool pshare = p.tile_map >= 1 && (G & 7) == 0;
    const bool contig = p.tile_map == 2;
    const bool ranged = p.tile_map == 3 && pshare;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, GX = G >> 3;
    const int unit = 8 * p.n_cls, n_units = n_pt / unit;          // (n_pt is a multiple of 8 * n_cls under cls_minor)
    const int e_lo = ranged ? (int)((long long)xcd * n_units / 8) * unit : 0;
    const int e_hi = ranged ? (int)((long long)(xcd + 1) * n_units / 8) * unit : 0;
    const int xcd_tiles = ranged ? (e_hi - e_lo) * n_ct : pshare ? ((n_pt - xcd + 7) >> 3) * n_ct : 0;
    const int run_lo = contig ? (int)((long long)slot * xcd_tiles / GX) : 0;
    const int run_hi = contig ? (int)((long long)(slot + 1) * xcd_tiles / GX) : 0;
    const int my_tiles = !pshare ? (n_tiles - (int)blockIdx.x + G - 1) / G
                         : contig ? run_hi - run_lo
                                  : (slot < xcd_tiles ? (xcd_tiles - slot + GX - 1) / GX : 0);
    auto tile_at = [&](int q) __attribute__((always_inline)) -> int {
        if (!pshare) return blockIdx.x + q * G;
        const int li = contig ? run_lo + q : slot + q * GX;
        const int lq = fast_div(li, p.nct_magic, p.nct_shift), lr = li - lq * n_ct;
        if (ranged) return (e_lo + lq) * n_ct + lr;
        return (lq * 8 + xcd) * n_ct + lr;
    };
    const int nts = nt * SPK;                           // LDS stages per tile
    const int total = my_tiles * nts;                   // stages this block walks

    // one global_load_lds wave-instruction fills RPI rows x GS granules; the LDS image is lane-linear,
    // the XOR swizzle is applied through the SOURCE granule each lane fetches.  Row r keeps granule g
    // at slot g ^ swz(r): swz = r & 7 for 128-byte rows, {0,2,3,1}[(r >> 2) & 3] for 64-byte rows
    // (both brute-forced conflict-free for the four 16-lane groups of ds_read_b128).
    const int lrow = lane / GS;                         // row inside the instruction's row group
    const int lswz = GS == 8 ? lrow : ((0x78 >> (2 * ((lrow >> 2) & 3))) & 3);
    const int gsrc = (lane % GS) ^ lswz;                // source granule (within the stage) this lane fetches
    const int HoWo = p.Ho * p.Wo;
    // output-grid pixel index inside one patch -> (oy, ox).  Linear: row-major, a pixel tile = a strip of BP consecutive pixels.
    // tile2d (round 4; Ho, Wo multiples of 16): the index space is cut into 16 x 16 BLOCKS (256 consecutive indices = one block, blocks
    // row-major), so a pixel tile is one or two square blocks: the taps of a 3x3 / 2x2 conv re-read an 18 x 18 halo INSIDE the tile's
    // own K loop (L2 hits a few microseconds apart) instead of rows that the tiles above / below fetch at other times on other CUs.
    // Which pixels share a tile changes, what is computed for a pixel does not: results are bit-identical.
    auto decode_yx = [&](int rem, int& oy, int& ox) __attribute__((always_inline)) {
        if (p.tile2d) {
            const int t = rem >> 8, r = rem & 255;
            const int ty = fast_div(t, p.tpr_magic, p.tpr_shift);
            oy = (ty << 4) + (r >> 4);
            ox = ((t - ty * p.tpr) << 4) + (r & 15);
        } else {
            oy = fast_div(rem, p.wo_magic, p.wo_shift);
            ox = rem - oy * p.Wo;
        }
    };
    // pixel index m of class `cls` -> (patch, oy, ox) on the op's output grid.  Owned-region launches (ConvParams::rmap, region.h) look the
    // triple up in the launch's table -- the grid is walked only where the page stitch keeps the result (plus the later levels' halo);
    // the table of a parity class is the one of its placement offset
    // (fast-gather kernels only -- every decoder conv takes them; the plain-gather tiles have no registers for the lookup and the host
    //  never hands them a table: launch_op)
    auto decode_m = [&](int m, int cls, int& n, int& oy, int& ox) __attribute__((always_inline)) {
        if (FG && p.rmap) {
            const int slot = p.n_cls > 1 ? p.ooy_cls[cls] * 2 + p.oox_cls[cls] : 0;
            const uint32_t code = p.rmap[(size_t)slot * (size_t)p.M + (size_t)m];
            n = (int)(code >> 22); oy = (int)((code >> 11) & 2047u); ox = (int)(code & 2047u);
        } else {
            n = fast_div(m, p.howo_magic, p.howo_shift);
            decode_yx(m - n * HoWo, oy, ox);
        }
    };

    // both sources' descriptors live in SGPRs for the whole kernel
    const SrcDesc sd0 = p.src[0];
    const SrcDesc sd1 = p.n_src > 1 ? p.src[1] : p.src[0];
    const int ks0 = p.n_src > 1 ? sd0.ksteps : nt_full;
    const uint32_t img0 = (uint32_t)(sd0.PH * sd0.PW * sd0.pix_bytes), img1 = (uint32_t)(sd1.PH * sd1.PW * sd1.pix_bytes);
    // load-side class state (weights, tap tables), switched in setup_rows
    const char* wbase = (const char*)p.w;
    const __attribute__((address_space(4))) int* kstep_tab =
        (const __attribute__((address_space(4))) int*)(uintptr_t)(FG ? (const void*)p.fgstep_cls[0] : (const void*)p.kstep);
    const KTabEntry* ktab = p.ktab;

    // ---- load side: rows of the tile being STAGED (runs D steps ahead of the compute side)
    // plain gather: output coords (oy, ox, n) of the staged rows.  Fast gather: r_oy = byte offset of the row's centre tap
    // in source 0, r_ox = the same in source 1, r_n = mask of out-of-bounds taps (bit src*16 + (dy-tap_lo_y)*4 + (dx-tap_lo_x)).
    int r_oy[T::kPLoads], r_ox[T::kPLoads], r_n[T::kPLoads];
    uint32_t w_off[T::kWLoads];
    // fast gather: buffer resources.  A source's resource starts fg_bias bytes BEFORE its buffer so that the scalar
    // offset (tap displacement + fg_bias) is never negative.
    const uint32_t fg_bias0 = (uint32_t)(kFgBiasPixels(sd0.PW) * sd0.pix_bytes), fg_bias1 = (uint32_t)(kFgBiasPixels(sd1.PW) * sd1.pix_bytes);
    // bytes from a source row's first granule to the granule this lane fetches
    auto lane_part = [&](const SrcDesc& sd) __attribute__((always_inline)) -> uint32_t {
        return X3 ? (uint32_t)((gsrc & 3) * 16 + (gsrc >> 2) * sd.lo_off) : (uint32_t)(gsrc * 16);
    };
    auto oob_mask = [&](const SrcDesc& sd, int oy, int ox) __attribute__((always_inline)) -> uint32_t {
        const int cy = oy << sd.sy_shift, cx = ox << sd.sx_shift;
        uint32_t xm = 0, m = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) xm |= ((unsigned)(cx + b + sd.tap_lo_x) < (unsigned)sd.lim_x) ? 0u : (1u << b);
#pragma unroll
        for (int a = 0; a < 4; ++a) m |= (((unsigned)(cy + a + sd.tap_lo_y) < (unsigned)sd.lim_y) ? xm : 0xfu) << (4 * a);
        return m;
    };
    auto setup_rows = [&](int tile) __attribute__((always_inline)) {
        int ctile, cls, ptile;
        decode(tile, ctile, cls, ptile);
        if (p.n_cls > 1) {
            wbase = (const char*)p.w_cls[cls];
            kstep_tab = (const __attribute__((address_space(4))) int*)(uintptr_t)(FG ? (const void*)p.fgstep_cls[cls] : (const void*)p.kstep_cls[cls]);
            ktab = p.ktab_cls[cls];
        }
        if constexpr (FG && !X3) {              // (the split-mode tiles have no registers to spare for it: every lane does every row)
            // The 8 lanes of an LDS row (lane & 7 = granule) stage the same kPLoads pixel rows: lane g works out row j = g
            // only, and the group shares the results through ds_bpermute -- instead of every lane repeating all kPLoads
            // rows (~75 VALU ops each; on a 17-K-step decoder tile that was a quarter of the wave's MFMA time).
            static_assert(T::kPLoads <= 8 && RPI == 8, "one row per lane of the 8-lane group");
            const int g8 = lane & 7;
            int my_a = 0, my_b = 0, my_c = -1;              // row past M: every tap out of bounds -> zero rows
            const int mm = ptile * BP + ((g8 < T::kPLoads ? g8 : 0) * NW + wave) * RPI + lrow;
            if (g8 < T::kPLoads && mm < p.M) {
                int n, oy, ox;
                decode_m(mm, cls, n, oy, ox);
                my_a = (int)((uint32_t)n * img0 + (uint32_t)(((oy << sd0.sy_shift) * sd0.PW + (ox << sd0.sx_shift)) * sd0.pix_bytes) +
                             (uint32_t)kZeroHeaderBytes);
                // (fast_gather == 2: every tap is (0, 0) -- pointwise convs -- and in bounds for every real row)
                uint32_t inv = p.fast_gather == 2 ? 0u : oob_mask(sd0, oy, ox);
                if (p.n_src > 1) {
                    my_b = (int)((uint32_t)n * img1 + (uint32_t)(((oy << sd1.sy_shift) * sd1.PW + (ox << sd1.sx_shift)) * sd1.pix_bytes) +
                                 (uint32_t)kZeroHeaderBytes);
                    if (p.fast_gather != 2) inv |= oob_mask(sd1, oy, ox) << 16;
                }
                my_c = (int)inv;
            }
#pragma unroll
            for (int j = 0; j < T::kPLoads; ++j) {
                const int srcl4 = ((lane & ~7) | j) << 2;          // ds_bpermute takes the source lane's byte address
                r_n[j] = __builtin_amdgcn_ds_bpermute(srcl4, my_c);
                r_oy[j] = __builtin_amdgcn_ds_bpermute(srcl4, my_a) + gsrc * 16;      // + lane_part (plain modes: the lane's granule)
                r_ox[j] = __builtin_amdgcn_ds_bpermute(srcl4, my_b) + gsrc * 16;
            }
        }
#pragma unroll
        for (int j = 0; j < T::kPLoads; ++j) {
            const int m = ptile * BP + (j * NW + wave) * RPI + lrow;
            if constexpr (FG && !X3) {
                (void)m;
            } else if constexpr (FG) {
                if (m < p.M) {
                    // (timing probe, variant flag bit 5 / SBBSEG_CONV_PROBE_LOCAL=1: every staged row gathers from the first 1 024 output pixels'
                    //  neighbourhood -- real data, but L2-resident: what the loop does when no pixel load misses.  Results are wrong.)
                    const int mg = SBBSEG_PROBE(p.variant_flags & 32) ? (m & 1023) : m;
                    int n, oy, ox;
                    decode_m(mg, cls, n, oy, ox);
                    r_oy[j] = (int)((uint32_t)n * img0 + (uint32_t)(((oy << sd0.sy_shift) * sd0.PW + (ox << sd0.sx_shift)) * sd0.pix_bytes) +
                                    lane_part(sd0) + (uint32_t)kZeroHeaderBytes);
                    uint32_t inv = p.fast_gather == 2 ? 0u : oob_mask(sd0, oy, ox);
                    if (p.n_src > 1) {
                        r_ox[j] = (int)((uint32_t)n * img1 + (uint32_t)(((oy << sd1.sy_shift) * sd1.PW + (ox << sd1.sx_shift)) * sd1.pix_bytes) +
                                        lane_part(sd1) + (uint32_t)kZeroHeaderBytes);
                        if (p.fast_gather != 2) inv |= oob_mask(sd1, oy, ox) << 16;
                    } else {
                        r_ox[j] = 0;
                    }
                    r_n[j] = (int)inv;
                } else {
                    r_oy[j] = 0;
                    r_ox[j] = 0;
                    r_n[j] = -1;                            // every tap out of bounds -> zero rows
                }
            } else if (m < p.M) {
                int n, oy, ox;
                decode_m(m, cls, n, oy, ox);
                r_oy[j] = oy;
                r_ox[j] = ox;
                r_n[j] = n;
            } else {
                r_oy[j] = -(1 << 20);                   // always out of bounds -> zero granule
                r_ox[j] = 0;
                r_n[j] = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < T::kWLoads; ++j)
            w_off[j] = (uint32_t)((ctile * BC + min((j * NW + wave) * RPI + lrow, BC - 1)) * p.Ktot + gsrc * 8) * 2u;
    };

    int l_t = 0, l_h = 0, l_q = 0, issued = 0;          // load side: K-step, stage inside it, tile
    int l_end = nt;                                     // (split-K: the K-step behind this split's range)
    int rec_yx = kstep_tab[0], rec_coff = kstep_tab[1], rec_irr = kstep_tab[2];   // record of the NEXT stage issued
    auto issue = [&](int buf) __attribute__((always_inline)) {
        const int t = l_t;
        const bool s1 = t >= ks0;
        char* lds_p = smem + buf * T::kStageBytes;
        char* lds_w = lds_p + BP * RB;
        if constexpr (FG) {
            // the fast gather's K-step records (FgStepRec, built by the host) hold the wave-uniform part of the address --
            // tap displacement + channel offset + the resource's bias -- and the tap's bit in the rows' out-of-bounds masks
            const uint32_t soff = (uint32_t)rec_yx + (uint32_t)(X3 ? 0 : l_h * GS * 16);
            const int tapbit = rec_coff;
            auto rows = [&](const char* rbase, uint32_t nrec, const int (&voff)[T::kPLoads]) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < T::kPLoads; ++j) {
                    const uint32_t oob = (uint32_t)__builtin_amdgcn_sbfe(r_n[j], tapbit, 1);      // 0 or ~0
                    const uint32_t off = (oob & 0x80000000u) | (uint32_t)voff[j];
                    buffer_load_lds16(rbase, nrec, (LDS_AS void*)(lds_p + (j * NW + wave) * 1024), off, soff);
                }
            };
            if (s1) rows(sd1.base - fg_bias1, sd1.bytes + fg_bias1, r_ox); else rows(sd0.base - fg_bias0, sd0.bytes + fg_bias0, r_oy);
            // (timing probe, variant flag bit 6 / SBBSEG_CONV_PROBE_WHOT=1: the weight rows of every K-step come from the tile's first sixteen
            //  K-steps -- an L2-resident 32-64 KB per channel tile.  Results are wrong.)
            const uint32_t woff = (uint32_t)((SBBSEG_PROBE(p.variant_flags & 64) ? (t & 15) : t) * (kBK * 2) + l_h * GS * 16);
#pragma unroll
            for (int j = 0; j < T::kWLoads; ++j)
                buffer_load_lds16(wbase, 0x7fffffffu, (LDS_AS void*)(lds_w + (j * NW + wave) * 1024), w_off[j], woff);
        } else {
        const char* base = s1 ? sd1.base : sd0.base;
        const int rowbytes = s1 ? sd1.PW * sd1.pix_bytes : sd0.PW * sd0.pix_bytes;
        const int pixb = s1 ? sd1.pix_bytes : sd0.pix_bytes;
        const uint32_t img = s1 ? img1 : img0;
        const int sh = s1 ? sd1.shift : sd0.shift;
        const int ssy = s1 ? sd1.sy_shift : sd0.sy_shift, ssx = s1 ? sd1.sx_shift : sd0.sx_shift;
        const unsigned lim_y = s1 ? sd1.lim_y : sd0.lim_y, lim_x = s1 ? sd1.lim_x : sd0.lim_x;
        const int gfull = l_h * GS + gsrc;                     // granule inside the 64-element K-step
        int dy = (int)(short)(rec_yx & 0xffff), dx = rec_yx >> 16;
        // slots 4-7 sit lo_off bytes behind slots 0-3: 64 in the plain modes (= gfull * 16), the lo plane in split mode
        int coff = rec_coff + (X3 ? (gfull & 3) * 16 + (gfull >> 2) * (s1 ? sd1.lo_off : sd0.lo_off) : gfull * 16) + kZeroHeaderBytes;
        if (rec_irr) {                                         // granules of this step differ in tap
            const KTabEntry e = ktab[t * kGranulesPerStep + gfull];
            dy = e.dy; dx = e.dx; coff = e.coff + kZeroHeaderBytes;
        }
#pragma unroll
        for (int j = 0; j < T::kPLoads; ++j) {
            const int uy = (r_oy[j] << ssy) + dy;           // dy/dx carry tap - pad - placement offset
            const int ux = (r_ox[j] << ssx) + dx;
            const bool ok = ((unsigned)uy < lim_y) & ((unsigned)ux < lim_x);
            const int yy = uy >> sh, xx = ux >> sh;
            // yy, xx < 2^12 and rowbytes, pixb < 2^24: 24-bit multiplies are exact (full-rate VALU)
            uint32_t off = (uint32_t)r_n[j] * img + __umul24(yy, rowbytes) + __umul24(xx, pixb) + (uint32_t)coff;
            off = ok ? off : 0u;
            // (a non-temporal policy on these loads was measured 5-35 % slower: the tap re-reads live in L2)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(base + off),
                                             (LDS_AS void*)(lds_p + (j * NW + wave) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < T::kWLoads; ++j) {
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(wbase + w_off[j] + (uint32_t)(t * (kBK * 2) + l_h * GS * 16)),
                                             (LDS_AS void*)(lds_w + (j * NW + wave) * 1024), 16, 0, 0);
        }
        }
        // advance the load side; crossing into the next tile re-derives the gather rows
        ++issued;
        if (++l_h == SPK) {
            l_h = 0;
            if (++l_t == (KS ? l_end : nt)) {
                l_t = 0;
                if (++l_q < my_tiles) {
                    setup_rows(tile_at(l_q));
                    if constexpr (KS) l_t = split_of(tile_at(l_q)) * nt;
                }
                if constexpr (KS) l_end = l_t + nt;
            }
            rec_yx = kstep_tab[l_t * 4 + 0]; rec_coff = kstep_tab[l_t * 4 + 1]; rec_irr = kstep_tab[l_t * 4 + 2];
        }
    };

    f32x4_t acc[T::kMI][T::kNI];
#pragma unroll
    for (int mi = 0; mi < T::kMI; ++mi)
#pragma unroll
        for (int ni = 0; ni < T::kNI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // LDS read offsets: row r, granule g lives at r*128 + ((g ^ (r & 7)) * 16)
    const int frow = lane & 15;
    const int fg = lane >> 4;
    const int fswz = GS == 8 ? (frow & 7) : ((0x78 >> (2 * ((frow >> 2) & 3))) & 3);
    const int rd_k0 = frow * RB + (((0 + fg) ^ fswz) << 4);
    const int rd_k1 = frow * RB + (((4 + fg) ^ fswz) << 4);        // second k-half of a whole-K-step stage (GS == 8)
    const int p_rd = (wp * T::kWPX) * RB;
    const int w_rd = BP * RB + (wc * T::kWCH) * RB;

    // ---- epilogue of one finished tile.  Weight rows are packed in the order conv_row_channel()
    // gives, so the two MFMA row blocks (2s, 2s+1) of a lane hold 8 CONSECUTIVE channels of one
    // pixel: 16-byte NHWC stores / residual loads, 64 contiguous bytes per pixel per instruction.
    // linear pixel index inside the output tensor(s) for output-grid pixel m (placement: see ConvParams)
    const bool placed = (p.osy != 1) | (p.osx != 1) | (p.ooy != 0) | (p.oox != 0) | (p.TH != p.Ho) | (p.TW != p.Wo) | (p.n_cls > 1) | (FG && p.rmap != nullptr);
    auto out_pixel = [&](int m, int cls) __attribute__((always_inline)) -> int {
        if (!placed) return m;
        int n, oy, ox;
        decode_m(m, cls, n, oy, ox);
        const int ooy = p.n_cls > 1 ? p.ooy_cls[cls] : p.ooy, oox = p.n_cls > 1 ? p.oox_cls[cls] : p.oox;
        return (n * p.TH + oy * p.osy + ooy) * p.TW + ox * p.osx + oox;
    };

    // residual tile of the tile being finished: requested BEFORE its last K-step's MFMAs so the HBM
    // round trip hides under them
    // (big wave tiles cannot spare the registers; the split mode's hi + lo pairs fit beside the 128x128 tile's 187 registers only)
    constexpr bool kPrefetchRes = (T::kMI / 2) * T::kNI <= 8 && (!X3 || (BP == 128 && BC == 128));
    uint4 res[kPrefetchRes ? T::kMI / 2 : 1][kPrefetchRes ? T::kNI : 1];
    uint4 res_lo[kPrefetchRes && X3 ? T::kMI / 2 : 1][kPrefetchRes && X3 ? T::kNI : 1];      // split mode: the lo halves
    auto prefetch_residual = [&](int tile) __attribute__((always_inline)) {
        int ctile, cls, ptile;
        decode(tile, ctile, cls, ptile);
#pragma unroll
        for (int ni = 0; ni < T::kNI; ++ni) {
            const int m = ptile * BP + wp * T::kWPX + ni * 16 + frow;
            const int opix = m < p.M ? out_pixel(m, cls) : 0;
#pragma unroll
            for (int s2 = 0; s2 < T::kMI / 2; ++s2) {
                const int c0 = ctile * BC + wc * T::kWCH + s2 * 32 + fg * 8;
                if constexpr (kPrefetchRes) {
                    if (c0 < p.cout && m < p.M) {
                        const uint16_t* rp = (const uint16_t*)p.residual + (size_t)opix * (p.cout * PL) + (X3 ? split_hi_elem(p.cout, c0) : c0);
                        res[s2][ni] = *(const uint4*)rp;
                        if constexpr (X3) res_lo[s2][ni] = *(const uint4*)(rp + split_group(p.cout));
                    }
                }
            }
        }
    };

    // epilogue constants of the tile being finished, requested like the residual BEFORE the next
    // stage's loads are issued: vmcnt retires in order, so a load issued in the epilogue itself would
    // only return after the whole next stage has landed -- one extra round trip per tile on short-K layers
    constexpr bool kPrefetchConst = (T::kMI / 2) * T::kNI <= 8;
    float csc[kPrefetchConst ? T::kMI / 2 : 1][8], csh[kPrefetchConst ? T::kMI / 2 : 1][8];
    auto prefetch_consts = [&](int tile) __attribute__((always_inline)) {
        int ctile, cls, ptile;
        decode(tile, ctile, cls, ptile);
#pragma unroll
        for (int s2 = 0; s2 < T::kMI / 2; ++s2) {
            const int c0 = ctile * BC + wc * T::kWCH + s2 * 32 + fg * 8;
            if constexpr (kPrefetchConst) {
                if (c0 < p.cout) {
                    *(float4*)&csc[s2][0] = *(const float4*)(p.scale + c0);
                    *(float4*)&csc[s2][4] = *(const float4*)(p.scale + c0 + 4);
                    *(float4*)&csh[s2][0] = *(const float4*)(p.shift + c0);
                    *(float4*)&csh[s2][4] = *(const float4*)(p.shift + c0 + 4);
                    if constexpr (X3) {                          // undo the class's power-of-two weight pre-scale (exact)
                        const float wm = p.wmul_cls[cls];
#pragma unroll
                        for (int q = 0; q < 8; ++q) csc[s2][q] *= wm;
                    }
                }
            }
        }
    };

    // ---- epilogue of one finished tile.  Weight rows are packed in the order conv_row_channel()
    // gives, so the two MFMA row blocks (2s, 2s+1) of a lane hold 8 CONSECUTIVE channels of one
    // pixel: 16-byte NHWC stores / residual loads, 64 contiguous bytes per pixel per instruction.
    // Returns true when this wave issued EXACTLY kEpiStores (x2 with a raw copy) store instructions:
    // a full interior tile with a plain 16-bit output.  The caller may then leave those stores in
    // flight behind a counted wait (vmcnt retires loads and stores in issue order on gfx9-family
    // parts, and the staging loads were issued before the stores).
    auto epilogue = [&](int tile) __attribute__((always_inline)) -> bool {
        int ctile, cls, ptile;
        decode(tile, ctile, cls, ptile);
        if constexpr (KS) {
            // this split's partial sums, scaled by the class's power-of-two weight multiplier (exact), at the FINAL pixel index of the
            // output tensor: splitk_finish_x3 is then a plain elementwise pass over [pixels][cout]
            float* ws = p.ks_ws + (size_t)split_of(tile) * (size_t)p.ks_split_elems;
            const float wm = p.wmul_cls[cls];
#pragma unroll
            for (int ni = 0; ni < T::kNI; ++ni) {
                const int m = ptile * BP + wp * T::kWPX + ni * 16 + frow;
                if (m < p.M) {
                    const size_t o = (size_t)out_pixel(m, cls) * p.cout;
#pragma unroll
                    for (int s2 = 0; s2 < T::kMI / 2; ++s2) {
                        const int c0 = ctile * BC + wc * T::kWCH + s2 * 32 + fg * 8;
                        if (c0 < p.cout) {
                            const f32x4_t a = acc[2 * s2][ni], b = acc[2 * s2 + 1][ni];
                            *(float4*)(ws + o + c0) = make_float4(a[0] * wm, a[1] * wm, a[2] * wm, a[3] * wm);
                            *(float4*)(ws + o + c0 + 4) = make_float4(b[0] * wm, b[1] * wm, b[2] * wm, b[3] * wm);
                        }
                    }
                }
            }
#pragma unroll
            for (int mi = 0; mi < T::kMI; ++mi)
#pragma unroll
                for (int ni = 0; ni < T::kNI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            return false;
        }
        const bool full = (ptile * BP + (wp + 1) * T::kWPX <= p.M) && (ctile * BC + (wc + 1) * T::kWCH <= p.cout) &&
                          p.out != nullptr && !(BC == 32 && p.head_classes > 0);
        int opix[T::kNI];
#pragma unroll
        for (int ni = 0; ni < T::kNI; ++ni) {
            const int m = ptile * BP + wp * T::kWPX + ni * 16 + frow;
            opix[ni] = m < p.M ? out_pixel(m, cls) : -1;
        }
        if constexpr (T::kMI % 4 == 0 && !X3) {
            // Full interior tile: whole-line stores.  A lane holds 8 channels of pixel `frow` from row-block
            // pair s2 (A) and from pair s2+1 (B); lanes frow and frow^8 swap "B of the low pixel" against
            // "A of the high pixel" (one DPP row rotate), after which each store instruction writes 8
            // pixels x 128 contiguous, line-aligned bytes instead of 16 pixels x 64 (half lines).
            if (full && !p.raw_out && !(p.variant_flags & 2)) {
                const bool hi = (frow & 8) != 0;
#pragma unroll
                for (int sp = 0; sp < T::kMI / 4; ++sp) {
                    const int cA = ctile * BC + wc * T::kWCH + sp * 64 + fg * 8;
                    const int cst = cA + (hi ? 32 : 0);
                    float sc[2][8], sh[2][8];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if constexpr (kPrefetchConst) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) { sc[h][q] = csc[sp * 2 + h][q]; sh[h][q] = csh[sp * 2 + h][q]; }
                        } else {
                            *(float4*)&sc[h][0] = *(const float4*)(p.scale + cA + h * 32);
                            *(float4*)&sc[h][4] = *(const float4*)(p.scale + cA + h * 32 + 4);
                            *(float4*)&sh[h][0] = *(const float4*)(p.shift + cA + h * 32);
                            *(float4*)&sh[h][4] = *(const float4*)(p.shift + cA + h * 32 + 4);
                        }
                    }
#pragma unroll
                    for (int ni = 0; ni < T::kNI; ++ni) {
                        uint4 r[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int s2 = sp * 2 + h;
                            float y[8];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                y[q] = __builtin_fmaf(acc[2 * s2][ni][q], sc[h][q], sh[h][q]);
                                y[4 + q] = __builtin_fmaf(acc[2 * s2 + 1][ni][q], sc[h][4 + q], sh[h][4 + q]);
                            }
                            if (p.residual) {
                                uint4 rr;
                                if constexpr (kPrefetchRes) rr = res[s2][ni];
                                else rr = *(const uint4*)((const uint16_t*)p.residual + (size_t)opix[ni] * p.cout + cA + h * 32);
                                y[0] += unpack_lo<F16>(rr.x); y[1] += unpack_hi<F16>(rr.x);
                                y[2] += unpack_lo<F16>(rr.y); y[3] += unpack_hi<F16>(rr.y);
                                y[4] += unpack_lo<F16>(rr.z); y[5] += unpack_hi<F16>(rr.z);
                                y[6] += unpack_lo<F16>(rr.w); y[7] += unpack_hi<F16>(rr.w);
                            }
                            if (p.relu) {
#pragma unroll
                                for (int q = 0; q < 8; ++q) y[q] = fmaxf(y[q], 0.f);
                            }
                            r[h].x = pack2<F16>(y[0], y[1]); r[h].y = pack2<F16>(y[2], y[3]);
                            r[h].z = pack2<F16>(y[4], y[5]); r[h].w = pack2<F16>(y[6], y[7]);
                        }
                        uint4 give, recv;
                        give.x = hi ? r[0].x : r[1].x; give.y = hi ? r[0].y : r[1].y;
                        give.z = hi ? r[0].z : r[1].z; give.w = hi ? r[0].w : r[1].w;
                        recv.x = row_ror8(give.x); recv.y = row_ror8(give.y);
                        recv.z = row_ror8(give.z); recv.w = row_ror8(give.w);
                        const int o_other = (int)row_ror8((uint32_t)opix[ni]);
                        const int pix0 = hi ? o_other : opix[ni], pix1 = hi ? opix[ni] : o_other;
                        uint4 st0, st1;
                        st0.x = hi ? recv.x : r[0].x; st0.y = hi ? recv.y : r[0].y;
                        st0.z = hi ? recv.z : r[0].z; st0.w = hi ? recv.w : r[0].w;
                        st1.x = hi ? r[1].x : recv.x; st1.y = hi ? r[1].y : recv.y;
                        st1.z = hi ? r[1].z : recv.z; st1.w = hi ? r[1].w : recv.w;
                        *(uint4*)((uint16_t*)p.out + (size_t)pix0 * p.cout + cst) = st0;
                        *(uint4*)((uint16_t*)p.out + (size_t)pix1 * p.cout + cst) = st1;
                    }
                }
#pragma unroll
                for (int mi = 0; mi < T::kMI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < T::kNI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                return true;
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < T::kMI / 2; ++s2) {
            const int c0 = ctile * BC + wc * T::kWCH + s2 * 32 + fg * 8;
            if (c0 < p.cout) {
                float sc[8], sh[8], rsc[8], rsh[8];
                if constexpr (kPrefetchConst) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { sc[q] = csc[s2][q]; sh[q] = csh[s2][q]; }
                } else {
                    *(float4*)&sc[0] = *(const float4*)(p.scale + c0);
                    *(float4*)&sc[4] = *(const float4*)(p.scale + c0 + 4);
                    *(float4*)&sh[0] = *(const float4*)(p.shift + c0);
                    *(float4*)&sh[4] = *(const float4*)(p.shift + c0 + 4);
                    if constexpr (X3) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) sc[q] *= p.wmul_cls[cls];
                    }
                }
                if (p.raw_out) {
                    *(float4*)&rsc[0] = *(const float4*)(p.raw_scale + c0);
                    *(float4*)&rsc[4] = *(const float4*)(p.raw_scale + c0 + 4);
                    *(float4*)&rsh[0] = *(const float4*)(p.raw_shift + c0);
                    *(float4*)&rsh[4] = *(const float4*)(p.raw_shift + c0 + 4);
                    if constexpr (X3) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) rsc[q] *= p.wmul_cls[cls];
                    }
                }
#pragma unroll
                for (int ni = 0; ni < T::kNI; ++ni) {
                    float v[8], y[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] = acc[2 * s2][ni][q]; v[4 + q] = acc[2 * s2 + 1][ni][q]; }
#pragma unroll
                    for (int q = 0; q < 8; ++q) y[q] = __builtin_fmaf(v[q], sc[q], sh[q]);
                    // element offset of the pixel's channel group; the split mode stores [C hi][C lo] per pixel
                    const size_t o = (size_t)(opix[ni] < 0 ? 0 : opix[ni]) * (p.cout * PL) + (X3 ? split_hi_elem(p.cout, c0) : c0);
                    const int lo_d = split_group(p.cout);                  // (split mode: halves from a group's hi to its lo part)
                    if (opix[ni] >= 0) {
                        if (p.raw_out) {
                            float rv[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) rv[q] = v[q] * rsc[q] + rsh[q];
                            if constexpr (X3) store_split8((uint16_t*)p.raw_out + o, lo_d, rv);
                            else {
                                uint4 r;
                                r.x = pack2<F16>(rv[0], rv[1]); r.y = pack2<F16>(rv[2], rv[3]);
                                r.z = pack2<F16>(rv[4], rv[5]); r.w = pack2<F16>(rv[6], rv[7]);
                                *(uint4*)((uint16_t*)p.raw_out + o) = r;
                            }
                        }
                        if (p.residual) {
                            if constexpr (X3 && kPrefetchRes) {
                                const f16x8_t h = __builtin_bit_cast(f16x8_t, res[s2][ni]), l = __builtin_bit_cast(f16x8_t, res_lo[s2][ni]);
#pragma unroll
                                for (int q = 0; q < 8; ++q) y[q] = __fadd_rn(y[q], __fadd_rn((float)h[q], (float)l[q]));      // (= add_split8)
                            } else if constexpr (X3) add_split8((const uint16_t*)p.residual + o, lo_d, y);
                            else {
                                uint4 rr;
                                if constexpr (kPrefetchRes) rr = res[s2][ni];
                                else rr = *(const uint4*)((const uint16_t*)p.residual + o);
                                y[0] += unpack_lo<F16>(rr.x); y[1] += unpack_hi<F16>(rr.x);
                                y[2] += unpack_lo<F16>(rr.y); y[3] += unpack_hi<F16>(rr.y);
                                y[4] += unpack_lo<F16>(rr.z); y[5] += unpack_hi<F16>(rr.z);
                                y[6] += unpack_lo<F16>(rr.w); y[7] += unpack_hi<F16>(rr.w);
                            }
                        }
                    }
                    if (p.relu) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) y[q] = fmaxf(y[q], 0.f);
                    }
                    if (p.out && opix[ni] >= 0) {
                        if constexpr (X3) store_split8((uint16_t*)p.out + o, lo_d, y);
                        else {
                            uint4 r;
                            r.x = pack2<F16>(y[0], y[1]); r.y = pack2<F16>(y[2], y[3]);
                            r.z = pack2<F16>(y[4], y[5]); r.w = pack2<F16>(y[6], y[7]);
                            *(uint4*)((uint16_t*)p.out + o) = r;
                        }
                    }
                    if constexpr (BC == 32) {
                        // fused head: the 32 channels of a pixel sit in the 4 lanes {frow + 16*fg}; each
                        // lane contracts its 8 fp32 channels, two xor-shuffles add the partial logits
                        // (no 16-bit rounding between the last conv and the softmax)
                        if (p.head_classes > 0) {
                            float logit[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                float a = 0.f;
                                if (c < p.head_classes) {
#pragma unroll
                                    for (int q = 0; q < 8; ++q) a = fmaf(y[q], p.head_w[(c0 + q) * p.head_classes + c], a);
                                }
                                a += __shfl_xor(a, 16);
                                a += __shfl_xor(a, 32);
                                logit[c] = a;
                            }
                            if (fg == 0 && opix[ni] >= 0) {
                                float mx = -3.0e38f;
#pragma unroll
                                for (int c = 0; c < 4; ++c)
                                    if (c < p.head_classes) { logit[c] = logit[c] * p.head_scale[c] + p.head_shift[c]; mx = fmaxf(mx, logit[c]); }
                                float pr[4], sum = 0.f;
#pragma unroll
                                for (int c = 0; c < 4; ++c)
                                    if (c < p.head_classes) { pr[c] = expf(logit[c] - mx); sum += pr[c]; }
                                int best = 0;
                                float bestp = -1.f;
#pragma unroll
                                for (int c = 0; c < 4; ++c)
                                    if (c < p.head_classes) {
                                        pr[c] = pr[c] / sum;
                                        if (pr[c] > bestp) { bestp = pr[c]; best = c; }      // first maximum wins (np.argmax)
                                        if (p.probs) p.probs[(size_t)opix[ni] * p.head_classes + c] = pr[c];
                                    }
                                p.labels[opix[ni]] = (uint8_t)best;
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int mi = 0; mi < T::kMI; ++mi)
#pragma unroll
            for (int ni = 0; ni < T::kNI; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        return full;
    };
    constexpr int kEpiStores = (T::kMI / 2) * T::kNI;             // store instructions per wave per output tensor

    if constexpr (PH8) {
        // ============================================================================================
        // 8-phase schedule (256 x 256 tile, 8 waves as 2 x 4, two K-steps = 8 phases per turn of the ring).
        // A K-step is staged as FOUR half-tiles of 16 KB -- W0/W1 = the two 32-channel halves of every
        // wave column, P0/P1 = the two 64-pixel halves of every wave row -- and a half-tile is restaged
        // (with the data of K-step k+2) as soon as both wave groups have taken their fragments from it,
        // not when the whole stage is spent: 5-6 phases of lead time instead of 4 in the same 128 KB.
        // Each phase = [fragment reads + one half-tile issued + counted vmcnt] barrier [16 MFMAs] barrier;
        // waves 4-7 run one barrier behind waves 0-3, so one group's memory half overlaps the other's
        // matrix half on every SIMD.
        //   phase of K-step k :   1            2            3            4
        //   fragments read    :   W0 P0        W1           P1           -
        //   quadrant (W,P)    :   (0,0)        (1,0)        (1,1)        (0,1)
        //   half-tile issued  :   W1(k+1)      P1(k+1)      W0(k+2)      P0(k+2)
        // ============================================================================================
        static_assert(BP == 256 && BC == 256 && WP == 2 && WC == 4 && NS == 2 && GS == 8, "8-phase schedule is built for the 256x256 tile");
        if (my_tiles == 0) return;
        const int total_k = my_tiles * nt;                       // K-steps this block walks
        constexpr int kHalf = 16384;
        // ---- load side
        int q_oy[4], q_ox[4], q_n[4];                            // P rows of this thread: [h*2 + i]
        uint32_t q_w[4];                                         // W rows: [g*2 + i]
        auto setup8 = [&](int tile) __attribute__((always_inline)) {
            int ctile, cls, ptile;
            decode(tile, ctile, cls, ptile);
            if (p.n_cls > 1) {
                wbase = (const char*)p.w_cls[cls];
                kstep_tab = (const __attribute__((address_space(4))) int*)(uintptr_t)p.kstep_cls[cls];
                ktab = p.ktab_cls[cls];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rho = ((j & 1) * 8 + wave) * 8 + lrow;            // row inside the half-tile
                const int m = ptile * BP + (rho >> 6) * 128 + (j >> 1) * 64 + (rho & 63);
                if (m < p.M) {
                    // (no owned-region table on this schedule: ph8_ok() keeps launches with a table off it -- the lookup's registers spill here)
                    const int n = fast_div(m, p.howo_magic, p.howo_shift);
                    const int rem = m - n * HoWo;
                    int oy, ox;
                    decode_yx(rem, oy, ox);
                    q_oy[j] = oy; q_ox[j] = ox; q_n[j] = n;
                } else {
                    q_oy[j] = -(1 << 20); q_ox[j] = 0; q_n[j] = 0;
                }
                const int crow = (rho >> 5) * 64 + (j >> 1) * 32 + (rho & 31);
                q_w[j] = (uint32_t)((ctile * BC + min(crow, BC - 1)) * p.Ktot + gsrc * 8) * 2u;
            }
        };
        int i_t = 0, i_q = 0, i_k = 0;                           // issue side: K-step in tile, tile, K-step overall
        int r8_yx = 0, r8_coff = 0;
        auto load_rec = [&]() __attribute__((always_inline)) {
            r8_yx = kstep_tab[i_t * 4 + 0]; r8_coff = kstep_tab[i_t * 4 + 1];
        };
        // one half-tile of K-step i_k: part 0 = W0, 1 = P0, 2 = W1, 3 = P1 (issued in this order)
        auto issue_part = [&](int part, bool checked = true) __attribute__((always_inline)) {
            if (checked && i_k >= total_k) return;
            char* half = smem + (i_k & 1) * (4 * kHalf) + ((part & 1) ? 2 * kHalf : 0) + (part >> 1) * kHalf;
            if (!(part & 1)) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(wbase + q_w[(part >> 1) * 2 + i] + (uint32_t)(i_t * (kBK * 2))),
                                                     (LDS_AS void*)(half + (i * 8 + wave) * 1024), 16, 0, 0);
            } else {
                const bool s1 = i_t >= ks0;
                const char* base = s1 ? sd1.base : sd0.base;
                const int rowbytes = s1 ? sd1.PW * sd1.pix_bytes : sd0.PW * sd0.pix_bytes;
                const int pixb = s1 ? sd1.pix_bytes : sd0.pix_bytes;
                const uint32_t img = s1 ? img1 : img0;
                const int sh = s1 ? sd1.shift : sd0.shift;
                const int ssy = s1 ? sd1.sy_shift : sd0.sy_shift, ssx = s1 ? sd1.sx_shift : sd0.sx_shift;
                const unsigned lim_y = s1 ? sd1.lim_y : sd0.lim_y, lim_x = s1 ? sd1.lim_x : sd0.lim_x;
                const int dy = (int)(short)(r8_yx & 0xffff), dx = r8_yx >> 16;
                const int coff = r8_coff + gsrc * 16 + kZeroHeaderBytes;
                // (irregular K-steps -- 8-channel sources -- never reach this schedule: ph8_ok())
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int j = (part >> 1) * 2 + i;
                    const int uy = (q_oy[j] << ssy) + dy, ux = (q_ox[j] << ssx) + dx;
                    const bool ok = ((unsigned)uy < lim_y) & ((unsigned)ux < lim_x);
                    const int yy = uy >> sh, xx = ux >> sh;
                    uint32_t off = (uint32_t)q_n[j] * img + __umul24(yy, rowbytes) + __umul24(xx, pixb) + (uint32_t)coff;
                    off = ok ? off : 0u;
                    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(base + off), (LDS_AS void*)(half + (i * 8 + wave) * 1024), 16, 0, 0);
                }
            }
            if (part == 3) {                                     // K-step complete: advance
                ++i_k;
                if (++i_t == nt) {
                    i_t = 0;
                    ++i_q;
                    if (!checked || i_q < my_tiles) setup8(tile_at(i_q));
                }
                if (!checked || i_k < total_k) load_rec();
            }
        };
        // ---- read side
        const int fsw0 = (((0 + fg) ^ (frow & 7)) << 4), fsw1 = (((4 + fg) ^ (frow & 7)) << 4);
        const int a_row = (wc * 32 + frow) * 128, b_row = (wp * 64 + frow) * 128;
        bf16x8_t aw[2][2][2], bp[4][2];                          // [g][m2][kk], [n4][kk]
        auto read_w = [&](const char* buf, int g) __attribute__((always_inline)) {
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                aw[g][m2][0] = *(const bf16x8_t*)(buf + g * kHalf + a_row + m2 * 2048 + fsw0);
                aw[g][m2][1] = *(const bf16x8_t*)(buf + g * kHalf + a_row + m2 * 2048 + fsw1);
            }
        };
        auto read_p = [&](const char* buf, int h) __attribute__((always_inline)) {
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4) {
                bp[n4][0] = *(const bf16x8_t*)(buf + (2 + h) * kHalf + b_row + n4 * 2048 + fsw0);
                bp[n4][1] = *(const bf16x8_t*)(buf + (2 + h) * kHalf + b_row + n4 * 2048 + fsw1);
            }
        };
        auto quad = [&](int g, int h) __attribute__((always_inline)) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int n4 = 0; n4 < 4; ++n4)
                        acc[g * 2 + m2][h * 4 + n4] = mfma16<F16>(aw[g][m2][kk], bp[n4][kk], acc[g * 2 + m2][h * 4 + n4]);
            __builtin_amdgcn_s_setprio(0);
        };
        // after this phase's issue: everything but the 4 youngest half-tiles (8 loads) has landed; once the
        // stream has ended the count no longer says anything -> drain
        auto phase_wait = [&](bool checked = true) __attribute__((always_inline)) {
            if (!checked || i_k < total_k) wait_vmcnt<8>();
            else wait_vmcnt<0>();
        };

        // ---- prologue: K-step 0 complete and W0, P0 of K-step 1 issued; K-step 0's first halves landed
        setup8(tile_at(0));
        load_rec();
        issue_part(0); issue_part(1); issue_part(2); issue_part(3);
        issue_part(0); issue_part(1);
        phase_wait();
        __builtin_amdgcn_s_barrier();
        if (wave >= 4) __builtin_amdgcn_s_barrier();             // second group runs one barrier behind

        int c_t8 = 0, c_q8 = 0;
        // one K-step = four phases; `checked` = the stream may end inside this K-step (last two K-steps only):
        // the steady-state body carries no end-of-stream branches -- the memory half of a phase has to be short
        auto kstep8 = [&](int k, bool checked) __attribute__((always_inline)) {
            const char* buf = smem + (k & 1) * (4 * kHalf);
            // phase 1
            read_w(buf, 0); read_p(buf, 0);
            issue_part(2, checked);
            phase_wait(checked);
            __builtin_amdgcn_s_barrier();
            quad(0, 0);
            __builtin_amdgcn_s_barrier();
            // phase 2
            read_w(buf, 1);
            issue_part(3, checked);
            phase_wait(checked);
            __builtin_amdgcn_s_barrier();
            quad(1, 0);
            __builtin_amdgcn_s_barrier();
            // phase 3
            read_p(buf, 1);
            issue_part(0, checked);
            phase_wait(checked);
            __builtin_amdgcn_s_barrier();
            quad(1, 1);
            __builtin_amdgcn_s_barrier();
            // phase 4
            issue_part(1, checked);
            phase_wait(checked);
            __builtin_amdgcn_s_barrier();
            quad(0, 1);
            if (++c_t8 == nt) {                                  // tile finished
                epilogue(tile_at(c_q8));
                c_t8 = 0;
                ++c_q8;
            }
            __builtin_amdgcn_s_barrier();
        };
        int k8 = 0;
        for (; k8 + 2 < total_k; ++k8) kstep8(k8, false);
        for (; k8 < total_k; ++k8) kstep8(k8, true);
        if (wave < 4) __builtin_amdgcn_s_barrier();              // balance the second group's extra barrier
        return;
    }

    // ---- prologue: D stages in flight, stage 0 landed
    if (total == 0) return;
    setup_rows(tile_at(0));
    if constexpr (KS) { l_t = split_of(tile_at(0)) * nt; l_end = l_t + nt; }
    rec_yx = kstep_tab[l_t * 4 + 0]; rec_coff = kstep_tab[l_t * 4 + 1]; rec_irr = kstep_tab[l_t * 4 + 2];     // (the first tile's class may not be class 0)
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < total) issue(d);
    if (D >= 2 && total >= D) wait_vmcnt<T::kLoads*(D >= 2 ? D - 1 : 0)>();     // stage 0 landed, D-1 stages stay in flight
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    int cur = 0, nxt = D % NS, c_t = 0, c_q = 0;
    for (int s = 0; s < total; ++s) {
        if (c_t == nts - 1) {                               // last stage of a tile: its epilogue inputs go first
            if (kPrefetchConst) prefetch_consts(tile_at(c_q));
            if (kPrefetchRes && p.residual) prefetch_residual(tile_at(c_q));
        }
        // 8-wave tiles of the plain 16-bit modes: waves NW/2.. (the SIMD partners of waves 0..NW/2-1) issue the next stage's loads in
        // the MIDDLE of the K-step, after their first half's MFMAs -- a stage's DMA issue costs a wave several hundred cycles in which
        // it feeds no MFMAs; issued by all eight waves right after the barrier those cycles coincided on every SIMD.  f16: dec3 -8.6 %,
        // 128 -> 128 3x3 -6 %, dec1 / dec2 on 256 x 256 tiles -1.5 % (+1.1 % and +0.75 % throughput).  The split mode's 512 x 128 tile
        // gains the same 8 % per launch but has no register left for the second call site (one VGPR spilled) and its 256 x 256 tile
        // loses 1-3 %: plain modes only.  (variant flag bit 3 = off)
        constexpr bool kSplitIssue = !X3 && NW == 8 && !PH8 && NS == 2;
        const bool late_issue = kSplitIssue && wave >= NW / 2 && !(p.variant_flags & 8);
        if (!late_issue && issued < total) issue(nxt);
        const char* sb = smem + cur * T::kStageBytes;
        if constexpr (X3 && T::kNI >= 8) {
            // issue() has just requested the NEXT stage's K-step record through the scalar cache (s_load: lgkmcnt, may return out of order).
            // With it pending the compiler must wait lgkmcnt(0) -- for ALL sixteen fragment requests below -- in front of the first MFMA; a use
            // it can see makes it wait for the record HERE (a few dozen cycles, scalar cache hit), after which the fragment waits are counted.
            asm volatile("" ::"s"(rec_yx), "s"(rec_coff), "s"(rec_irr));
        }
        if constexpr (X3) {
            // split mode: slots 0-3 hold the hi halves of the stage's 32 channels, slots 4-7 the lo halves (rd_k0 / rd_k1
            // address exactly these).  value = hi + lo on both operands: w*x ~= wl*xh + wh*xl + wh*xh, small terms first.
            // Three sweeps over the wave tile keep MFMAs on the same accumulator 16 instructions apart.
            // Wide wave tiles (8 pixel blocks) take the pixel fragments in two halves: the fragments of a stage do not all fit
            // beside 128 accumulator registers.
            if constexpr (T::kNI >= 8) {
                // Wide wave tiles (8 pixel blocks: the 8-wave 256 x 256 and 512 x 128 tiles): phases of NQ pixel blocks, the pixel fragments of
                // phase i + 1 requested BEFORE the MFMAs of phase i (register double buffer: two sets of NQ hi + NQ lo fragments = the
                // registers one set of four blocks took).  Left to itself hipcc loads every fragment right in front of its first MFMA --
                // a dozen `s_waitcnt lgkmcnt(0)` per K-step, each exposing an LDS round trip to the matrix pipe.
                // Request order at the head of the K-step = the order of use: the first sweep (wl x xh) needs 4 + NQ fragments, not all 16 --
                // LDS returns in order, all eight waves ask at once behind the barrier (16 KB each: ~1 000 cycles of the LDS pipe for the
                // whole burst), and the matrix pipe idles until a wave's first operands are there.
                // (Per accumulator the order of its three MFMAs is unchanged: same bits.)
                constexpr int NQ = 2;
                constexpr int NPH = T::kNI / NQ;
                bf16x8_t ah[T::kMI], al[T::kMI];
                bf16x8_t bh[2][NQ], bl[2][NQ];
#pragma unroll
                for (int mi = 0; mi < T::kMI; ++mi) al[mi] = *(const bf16x8_t*)(sb + w_rd + mi * 16 * RB + rd_k1);
#pragma unroll
                for (int q = 0; q < NQ; ++q) bh[0][q] = *(const bf16x8_t*)(sb + p_rd + q * 16 * RB + rd_k0);
#pragma unroll
                for (int mi = 0; mi < T::kMI; ++mi) ah[mi] = *(const bf16x8_t*)(sb + w_rd + mi * 16 * RB + rd_k0);
#pragma unroll
                for (int q = 0; q < NQ; ++q) bl[0][q] = *(const bf16x8_t*)(sb + p_rd + q * 16 * RB + rd_k1);
                auto load_b = [&](int ph, bf16x8_t (&dh)[NQ], bf16x8_t (&dl)[NQ]) __attribute__((always_inline)) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        dh[q] = *(const bf16x8_t*)(sb + p_rd + (ph * NQ + q) * 16 * RB + rd_k0);
                        dl[q] = *(const bf16x8_t*)(sb + p_rd + (ph * NQ + q) * 16 * RB + rd_k1);
                    }
                };
#pragma unroll
                for (int ph = 0; ph < NPH; ++ph) {
                    if (ph + 1 < NPH) {
                        load_b(ph + 1, bh[(ph + 1) & 1], bl[(ph + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);          // (the requests stay in FRONT of this phase's MFMAs)
                    }
#pragma unroll
                    for (int mi = 0; mi < T::kMI; ++mi)
#pragma unroll
                        for (int q = 0; q < NQ; ++q) acc[mi][ph * NQ + q] = mfma16<true>(al[mi], bh[ph & 1][q], acc[mi][ph * NQ + q]);
#pragma unroll
                    for (int mi = 0; mi < T::kMI; ++mi)
#pragma unroll
                        for (int q = 0; q < NQ; ++q) acc[mi][ph * NQ + q] = mfma16<true>(ah[mi], bl[ph & 1][q], acc[mi][ph * NQ + q]);
#pragma unroll
                    for (int mi = 0; mi < T::kMI; ++mi)
#pragma unroll
                        for (int q = 0; q < NQ; ++q) acc[mi][ph * NQ + q] = mfma16<true>(ah[mi], bh[ph & 1][q], acc[mi][ph * NQ + q]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                bf16x8_t ah[T::kMI], al[T::kMI];
#pragma unroll
                for (int mi = 0; mi < T::kMI; ++mi) {
                    ah[mi] = *(const bf16x8_t*)(sb + w_rd + mi * 16 * RB + rd_k0);
                    al[mi] = *(const bf16x8_t*)(sb + w_rd + mi * 16 * RB + rd_k1);
                }
                bf16x8_t bh[T::kNI], bl[T::kNI];
#pragma unroll
                for (int q = 0; q < T::kNI; ++q) {
                    bh[q] = *(const bf16x8_t*)(sb + p_rd + q * 16 * RB + rd_k0);
                    bl[q] = *(const bf16x8_t*)(sb + p_rd + q * 16 * RB + rd_k1);
                }
#pragma unroll
                for (int mi = 0; mi < T::kMI; ++mi)
#pragma unroll
                    for (int q = 0; q < T::kNI; ++q) acc[mi][q] = mfma16<true>(al[mi], bh[q], acc[mi][q]);
#pragma unroll
                for (int mi = 0; mi < T::kMI; ++mi)
#pragma unroll
                    for (int q = 0; q < T::kNI; ++q) acc[mi][q] = mfma16<true>(ah[mi], bl[q], acc[mi][q]);
#pragma unroll
                for (int mi = 0; mi < T::kMI; ++mi)
#pragma unroll
                    for (int q = 0; q < T::kNI; ++q) acc[mi][q] = mfma16<true>(ah[mi], bh[q], acc[mi][q]);
            }
        } else {
            // The K-step's MFMAs run in phases of (k-half kk, group of <= 4 pixel blocks); the LDS
            // fragments of phase i+1 are requested BEFORE the MFMAs of phase i (register double
            // buffer).  hipcc on its own emits "all ds_reads, lgkmcnt(0), all MFMAs" per k-half, which
            // leaves the matrix pipe idle while all 8 waves of the block read LDS in lockstep.
            constexpr int NIH = T::kNI >= 8 ? T::kNI / 2 : T::kNI;     // pixel blocks per phase
            constexpr int NH = T::kNI / NIH;
            constexpr int NP = (GS / 4) * NH;                      // k-halves per stage x pixel-block groups
            bf16x8_t a[2][T::kMI], b[2][NIH];
            auto load_a = [&](int kk, bf16x8_t (&dst)[T::kMI]) __attribute__((always_inline)) {
                const int rd = kk ? rd_k1 : rd_k0;
#pragma unroll
                for (int mi = 0; mi < T::kMI; ++mi) dst[mi] = *(const bf16x8_t*)(sb + w_rd + mi * 16 * RB + rd);
            };
            auto load_b = [&](int kk, int h, bf16x8_t (&dst)[NIH]) __attribute__((always_inline)) {
                const int rd = kk ? rd_k1 : rd_k0;
#pragma unroll
                for (int q = 0; q < NIH; ++q) dst[q] = *(const bf16x8_t*)(sb + p_rd + (h * NIH + q) * 16 * RB + rd);
            };
            load_a(0, a[0]);
            load_b(0, 0, b[0]);
#pragma unroll
            for (int ph = 0; ph < NP; ++ph) {
                const int kk = ph / NH, h = ph % NH;
                if (ph + 1 < NP) {
                    const int nkk = (ph + 1) / NH, nh = (ph + 1) % NH;
                    if (nh == 0) load_a(nkk, a[nkk & 1]);
                    load_b(nkk, nh, b[(ph + 1) & 1]);
                }
#pragma unroll
                for (int mi = 0; mi < T::kMI; ++mi)
#pragma unroll
                    for (int q = 0; q < NIH; ++q)
                        acc[mi][h * NIH + q] = mfma16<F16>(a[kk & 1][mi], b[ph & 1][q], acc[mi][h * NIH + q]);
                if constexpr (kSplitIssue) {
                    if (ph == NP / 2 - 1 && late_issue && issued < total) issue(nxt);
                }
            }
        }
        bool tile_done = false, counted = false;
        if (++c_t == nts) {                                 // tile finished: its stores overlap the
            counted = epilogue(tile_at(c_q));               // next tile's first stage(s), already in flight
            c_t = 0;
            ++c_q;
            tile_done = true;
        }
        if (s + 1 < total) {
            // stage s+1 must have landed; later issued stages stay in flight across the barrier, and
            // so do the stores of a tile that just finished (they are younger than every staging
            // load): a short-K layer otherwise pays a load AND a store round trip per tile, in series.
            constexpr int kAhead = T::kLoads * (D >= 2 ? D - 1 : 0);
            const bool ring_full = D < 2 || s + D < total;          // D-1 younger stages really are in flight
            if (tile_done) {
                if (counted && ring_full && !(p.variant_flags & 1)) {
                    if (p.raw_out) wait_vmcnt<kAhead + 2 * PL * kEpiStores>();      // (split mode: a hi and a lo store per group)
                    else wait_vmcnt<kAhead + PL * kEpiStores>();
                } else wait_vmcnt<0>();
            } else if (D >= 2 && ring_full) wait_vmcnt<kAhead>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
        cur = cur + 1 == NS ? 0 : cur + 1;
        nxt = nxt + 1 == NS ? 0 : nxt + 1;
    }
}

// ------------------------------------------------------------------------------------------------
// conv_naive_f32 -- SBBSEG_PREC_F32 handles only: same gather table, plain fp32 FMA.  Slow on
// purpose-free grounds: it exists to separate plumbing errors from bf16 rounding in parity tests.
// One thread = one pixel x 4 channels.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_naive_f32(const ConvParams p)
{
    const int cgroups = p.cout / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)p.M * cgroups) return;
    const int m = (int)(idx / cgroups);
    const int c0 = (int)(idx - (long)m * cgroups) * 4;
    const int HoWo = p.Ho * p.Wo;
    const int n = m / HoWo;
    const int rem = m - n * HoWo;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* w = (const float*)p.w;
    int t = 0;
    for (int s = 0; s < p.n_src; ++s) {
        const SrcDesc sd = p.src[s];
        const int iy0 = oy << sd.sy_shift, ix0 = ox << sd.sx_shift;
        for (int ks = 0; ks < sd.ksteps; ++ks, ++t) {
            for (int g = 0; g < kGranulesPerStep; ++g) {
                const KTabEntry e = p.ktab[t * kGranulesPerStep + g];
                const int uy = iy0 + e.dy, ux = ix0 + e.dx;
                if (!(((unsigned)uy < (unsigned)sd.lim_y) & ((unsigned)ux < (unsigned)sd.lim_x))) continue;
                const int yy = uy >> sd.shift, xx = ux >> sd.shift;
                const float* xp = (const float*)(sd.base + kZeroHeaderBytes +
                                                 (size_t)((n * sd.PH + yy) * sd.PW + xx) * sd.pix_bytes + e.coff);
                const int k0 = (t * kGranulesPerStep + g) * 8;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float xv = xp[q];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = fmaf(xv, w[(size_t)(c0 + c) * p.Ktot + k0 + q], acc[c]);
                }
            }
        }
    }
    const size_t opix = (size_t)(n * p.TH + oy * p.osy + p.ooy) * p.TW + ox * p.osx + p.oox;
    const size_t o = opix * p.cout + c0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (p.raw_out) ((float*)p.raw_out)[o + c] = acc[c] * p.raw_scale[c0 + c] + p.raw_shift[c0 + c];
        if (p.out) {
            float y = acc[c] * p.scale[c0 + c] + p.shift[c0 + c];
            if (p.residual) y += ((const float*)p.residual)[o + c];
            if (p.relu) y = fmaxf(y, 0.f);
            ((float*)p.out)[o + c] = y;
        }
    }
}

int conv_tile_bc(int cout) { return cout >= 128 ? 128 : (cout > 32 ? 64 : 32); }

// Channel stored in packed weight row `row` (16-bit modes).  Inside each wave tile of WCH channels
// (64, or 32 when the channel tile is 32) MFMA row block mi, row rho is given channel
// (mi>>1)*32 + (rho>>2)*8 + (mi&1)*4 + (rho&3): see the epilogue of conv_igemm_mfma.
int conv_row_channel(int row, int cout)
{
    const int wch = conv_tile_bc(cout) == 32 ? 32 : 64;
    const int base = (row / wch) * wch, t = row % wch;
    const int mi = t >> 4, rho = t & 15;
    return base + (mi >> 1) * 32 + (rho >> 2) * 8 + (mi & 1) * 4 + (rho & 3);
}

// (magic, shift) with n / d == umulhi(n, magic) >> shift for every 0 <= n < 2^31; magic == 0: d is a power of two, n >> shift
static void make_fast_div(uint32_t d, uint32_t* magic, uint32_t* shift)
{
    if (d == 0) d = 1;
    uint32_t s2 = 0;
    while ((1u << (s2 + 1)) <= d && s2 < 31) ++s2;              // 2^s2 <= d < 2^(s2+1)
    if ((d & (d - 1)) == 0) { *magic = 0; *shift = s2; return; }
    const unsigned long long num = 1ull << (32 + s2);
    *magic = (uint32_t)((num + d - 1) / d);                       // ceil(2^(32+s2) / d), in (2^31, 2^32)
    *shift = s2;
}

template <int BP, int BC, int WP, int WC, int NS, bool F16, int GS, bool PH8, bool X3, bool FG, bool KS = false>
static hipError_t launch_conv_impl(const ConvParams& p, hipStream_t s)
{
    using T = ConvTile<BP, BC, WP, WC, NS, GS>;
    static bool attr_done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!attr_done[dev & 63]) {
        e = hipFuncSetAttribute((const void*)conv_igemm_mfma<BP, BC, WP, WC, NS, F16, GS, PH8, X3, FG, KS>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, T::kLdsBytes);
        if (e != hipSuccess) return e;
        attr_done[dev & 63] = true;
    }
    const int n_ct = (p.cout + BC - 1) / BC;
    const int n_pt = (p.M + BP - 1) / BP;
    const int n_tiles = (p.n_cls * n_ct * (p.cls_minor ? (n_pt + 7) & ~7 : n_pt)) << (KS ? p.ks_shift : 0);
    // persistent grid: as many blocks as are resident at once (2 per CU for the 4-wave tiles, 1 for
    // the 8-wave ones); p.persist_blocks == 0 -> one block per tile (A/B)
    const int resident = p.persist_blocks > 0 ? p.persist_blocks * T::kBlocksPerCU : n_tiles;
    int grid = n_tiles < resident ? n_tiles : resident;
    // Balanced rounds (round 6, SBBSEG_BALANCED_GRID=1, off by default: an experiment): a launch of R = ceil(tiles / resident) rounds runs on
    // ceil(tiles / R) blocks instead of all resident ones -- every block walks R tiles (+- 1), no CU idles through a last partial round
    // while its neighbours finish it, and the CUs the launch does not occupy are free for the other lane's kernel from the start.
    static const bool balanced = getenv("SBBSEG_BALANCED_GRID") && getenv("SBBSEG_BALANCED_GRID")[0] == '1';
    if (balanced && p.persist_blocks > 0 && n_tiles > resident) {
        const int rounds = (n_tiles + resident - 1) / resident;
        grid = (n_tiles + rounds - 1) / rounds;
    }
    if (p.tile_map >= 1) grid = (grid + 7) & ~7;          // the XCD-grouped walk needs a multiple of 8 blocks
    ConvParams q = p;
    make_fast_div((uint32_t)(p.Ho * p.Wo), &q.howo_magic, &q.howo_shift);
    make_fast_div((uint32_t)p.Wo, &q.wo_magic, &q.wo_shift);
    make_fast_div((uint32_t)n_ct, &q.nct_magic, &q.nct_shift);
    // 2D pixel tiles for convs with real taps on the fast gather: an experiment (SBBSEG_TILE2D=1), off by default -- it measured neutral
    // (profiles/r04_experiments.md section 3) and one plan shape (a k = 2 Conv2DTranspose decoder at 32 x 48) came out wrong with it
    static const bool tile2d_on = getenv("SBBSEG_TILE2D") && getenv("SBBSEG_TILE2D")[0] == '1';
    q.tile2d = (tile2d_on && p.fast_gather == 1 && p.Ho % 16 == 0 && p.Wo % 16 == 0) ? 1 : 0;
    q.tpr = p.Wo / 16;
    make_fast_div((uint32_t)(q.tpr > 0 ? q.tpr : 1), &q.tpr_magic, &q.tpr_shift);
    hipLaunchKernelGGL((conv_igemm_mfma<BP, BC, WP, WC, NS, F16, GS, PH8, X3, FG, KS>), dim3(grid), dim3(T::kThreads), T::kLdsBytes, s, q);
    return hipGetLastError();
}

template <int BP, int BC, int WP, int WC, int NS, bool F16, int GS = 8, bool PH8 = false, bool X3 = false>
static hipError_t launch_conv_t(const ConvParams& p, hipStream_t s)
{
    if constexpr (!PH8 && GS == 8 && NS == 2) {        // (the opt-in half-stage, 3-stage and 8-phase forms keep the plain gather)
        if (p.fast_gather) return launch_conv_impl<BP, BC, WP, WC, NS, F16, GS, PH8, X3, true>(p, s);
    }
    if (p.rmap) return hipErrorInvalidValue;           // an owned-region table needs the fast-gather form (launch_op checks before it sets one)
    return launch_conv_impl<BP, BC, WP, WC, NS, F16, GS, PH8, X3, false>(p, s);
}

// Tile choice.  Measured on MI355X (profiles/r01_conv_variants.md): with two 4-wave blocks per CU the
// 2-stage tiles already hide the staging latency (dec1-3 at 925-980 TFLOP/s); the 8-wave / 3-stage /
// counted-vmcnt tiles (1 block per CU) are 5-25 % slower on every layer, so they are opt-in only.
// the 8-phase schedule of the 256x256 tile: regular K-steps only (>= 64-channel sources), at least 2 K-steps
static bool ph8_ok(const ConvParams& p)
{
    if (!(p.variant_flags & 4) || p.total_ksteps < 2 || p.half_stages || p.rmap) return false;      // opt-in: conv variant bit 16; never with an owned-region table
    for (int i = 0; i < p.n_src; ++i)
        if (p.src[i].pix_bytes < 128) return false;
    return true;
}

template <bool F16>
static hipError_t launch_conv_16(const ConvParams& p, hipStream_t s)
{
    const int bc = conv_tile_bc(p.cout);
    const int variant = p.variant;
    const long big_blocks = (long)((p.M + 255) / 256) * ((p.cout + bc - 1) / bc);
    const bool big = variant == 2;
    (void)big_blocks;
    if (p.ks_shift > 0) {                                       // split-K (see launch_conv_x3)
        if (bc != 128 || !p.fast_gather || p.tile_map != 0 || p.cls_minor || (p.total_ksteps & ((1 << p.ks_shift) - 1))) return hipErrorInvalidValue;
        return launch_conv_impl<128, 128, 2, 2, 2, F16, 8, false, false, true, true>(p, s);
    }
    if (p.half_stages) {   // A/B: half-K-step stages, 4-deep ring (3 stages in flight across the barriers)
        const long t256 = (long)p.n_cls * ((p.M + 255) / 256) * (p.cout / 256);
        if (bc == 128 && !p.residual && p.cout % 256 == 0 && p.Ktot >= 512 && t256 >= 200) return launch_conv_t<256, 256, 2, 4, 4, F16, 4>(p, s);
        if (bc == 128) return launch_conv_t<128, 128, 2, 2, 4, F16, 4>(p, s);
        if (bc == 64) return launch_conv_t<256, 64, 4, 1, 4, F16, 4>(p, s);
    }
    if (variant == 3) {   // force: 8 waves, wave tile 128 px x 64 ch (64x64 for cout 64), 2 LDS stages, 1 block per CU
        if (bc == 128 && p.cout % 256 == 0) return ph8_ok(p) ? launch_conv_t<256, 256, 2, 4, 2, F16, 8, true>(p, s) : launch_conv_t<256, 256, 2, 4, 2, F16>(p, s);
        if (bc == 128) return launch_conv_t<512, 128, 4, 2, 2, F16>(p, s);
        if (bc == 64) return launch_conv_t<512, 64, 8, 1, 2, F16>(p, s);
    }
    {   // round 5 (plain modes; profiles/r05_experiments.md section 13): residual layers on the 256 x 256 tile from this K on (0 = never) --
        // the stage-4 / stage-5 expands the fused pair kernel does not take, -3..-4 % each (the K = 128 one of stage 3 loses 1.5 %: left out)
        static const int kres256 = getenv("SBBSEG_F16_RES256_MINK") ? atoi(getenv("SBBSEG_F16_RES256_MINK")) : 256;
        if (variant == 0 && bc == 128 && p.residual && kres256 > 0 && p.cout % 256 == 0 && p.Ktot >= kres256 &&
            (long)p.n_cls * ((p.M + 255) / 256) * (p.cout / 256) >= 200)
            return launch_conv_t<256, 256, 2, 4, 2, F16>(p, s);
    }
    // (384 since round 5: the stage-3 projection merge, K = 384 channels, on the 256 x 256 tile: 0.40 -> 0.31 ms per 160 patches in fp16)
    static const int k256_16 = getenv("SBBSEG_F16_T256_MINK") ? atoi(getenv("SBBSEG_F16_T256_MINK")) : 384;
    if (variant == 0 && bc == 128 && !p.residual) {
        // auto (measured per layer, profiles/r01_conv_variants.md): the 8-wave tiles with 128x64 wave
        // tiles (LDS bytes per MFMA x0.75, L2 bytes per MFMA x0.5) win on long-K layers that still
        // give every CU a block; short-K / residual (HBM-bound) layers and small grids stay on 128x128
        const long t256 = (long)p.n_cls * ((p.M + 255) / 256) * (p.cout / 256);
        const long t512 = (long)p.n_cls * ((p.M + 511) / 512) * ((p.cout + 127) / 128);
        if (p.cout % 256 == 0 && p.Ktot >= k256_16 && t256 >= 200) return ph8_ok(p) ? launch_conv_t<256, 256, 2, 4, 2, F16, 8, true>(p, s) : launch_conv_t<256, 256, 2, 4, 2, F16>(p, s);
        if (p.Ktot >= 1024 && t512 >= 200) return launch_conv_t<512, 128, 4, 2, 2, F16>(p, s);
    }
    if (bc == 128) return big ? launch_conv_t<256, 128, 4, 2, 3, F16>(p, s) : launch_conv_t<128, 128, 2, 2, 2, F16>(p, s);
    if (bc == 64) return big ? launch_conv_t<256, 64, 8, 1, 3, F16>(p, s) : launch_conv_t<256, 64, 4, 1, 2, F16>(p, s);
    return big ? launch_conv_t<256, 32, 8, 1, 3, F16>(p, s) : launch_conv_t<256, 32, 4, 1, 2, F16>(p, s);
}

// split mode: the 4-wave tiles at 2 blocks per CU (3 MFMAs per product for the same LDS bytes: the matrix pipe, not the
// staging path, is what fills first here)
static hipError_t launch_conv_x3(const ConvParams& p, hipStream_t s)
{
    const int bc = conv_tile_bc(p.cout);
    if (p.ks_shift > 0) {                                       // split-K (the caller has checked: 128-channel tiles, fast gather, whole K ranges)
        if (bc != 128 || !p.fast_gather || p.tile_map != 0 || p.cls_minor || (p.total_ksteps & ((1 << p.ks_shift) - 1))) return hipErrorInvalidValue;
        return launch_conv_impl<128, 128, 2, 2, 2, true, 8, false, true, true, true>(p, s);
    }
    // Residual layers (the expand convs that `expand_reduce` does not take: the last block of stages 3 / 4, stage 5) on the 256 x 256 tile when
    // the launch still gives every CU a block: the pixel operand is re-read by half as many channel tiles; the residual is then read in the
    // epilogue (add_split8) instead of being prefetched -- same arithmetic.  Round 5: 0.61 -> 0.56, 0.39 -> 0.35, 0.27 -> 0.25 ms per 160
    // patches (profiles/r05_experiments.md section 12).  SBBSEG_X3_RES256_MINK=0 switches it off.
    static const int kres256 = getenv("SBBSEG_X3_RES256_MINK") ? atoi(getenv("SBBSEG_X3_RES256_MINK")) : 256;
    if (p.variant == 0 && bc == 128 && p.residual && kres256 > 0 && p.cout % 256 == 0 && p.Ktot >= kres256 &&
        (long)p.n_cls * ((p.M + 255) / 256) * (p.cout / 256) >= 200)
        return launch_conv_t<256, 256, 2, 4, 2, true, 8, false, true>(p, s);
    if (p.variant == 0 && bc == 128 && !p.residual) {          // the long-K decoder launches: same 8-wave tiles as the plain modes
        const long t256 = (long)p.n_cls * ((p.M + 255) / 256) * (p.cout / 256);
        const long t512 = (long)p.n_cls * ((p.M + 511) / 512) * ((p.cout + 127) / 128);
        static const int k512 = getenv("SBBSEG_X3_T512_MINK") ? atoi(getenv("SBBSEG_X3_T512_MINK")) : 2048;      // A/B knobs
        // (768 since round 5: the projection-shortcut merge at the head of stage 3 -- 128 + 256 -> 512 channels at 56 x 56, K = 384 channels -- takes
        //  the 256 x 256 tile too: 0.79 -> 0.68 ms per 160 patches; nothing else has a K between 768 and 1024)
        static const int k256 = getenv("SBBSEG_X3_T256_MINK") ? atoi(getenv("SBBSEG_X3_T256_MINK")) : 768;
        if (p.cout % 256 == 0 && p.Ktot >= k256 && t256 >= 200) return launch_conv_t<256, 256, 2, 4, 2, true, 8, false, true>(p, s);
        if (p.Ktot >= k512 && t512 >= 200) return launch_conv_t<512, 128, 4, 2, 2, true, 8, false, true>(p, s);
    }
    // (launches of less than one round of tiles -- one patch, the border model's whole-image forward -- cost the latency of their K-steps'
    // loads, ~1.4 us a step; 3- and 4-stage rings of the 128 x 128 tile were probed there and LOST, 3.57 vs 2.67 ms per one-patch forward:
    // the deeper rings run the plain gather, whose address arithmetic outweighs the extra loads in flight; profiles/r04_experiments.md section 12)
    if (bc == 128) return launch_conv_t<128, 128, 2, 2, 2, true, 8, false, true>(p, s);
    if (bc == 64) return launch_conv_t<256, 64, 4, 1, 2, true, 8, false, true>(p, s);
    return launch_conv_t<256, 32, 4, 1, 2, true, 8, false, true>(p, s);
}

// Second half of a split-K launch: partial sums of the 2^ks_shift splits added in split order, then the conv's epilogue -- scale / shift,
// residual, ReLU, store -- on 8 channels of one pixel per thread.  MODE 0 = split mode ([32 hi][32 lo] channel groups), 1 = fp16, 2 = bf16.
template <int MODE>
__global__ __launch_bounds__(256) void splitk_finish(const float* __restrict__ ws, int splits, long split_elems, long n_groups, int cout,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      const uint16_t* __restrict__ residual, int relu, uint16_t* __restrict__ out)
{
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= n_groups) return;
    const int cg = cout >> 3;
    const long pix = g / cg;
    const int c0 = (int)(g - pix * cg) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* src = ws + pix * cout + c0;
    for (int s = 0; s < splits; ++s) {
        const float4 a = *(const float4*)(src + (size_t)s * split_elems), b = *(const float4*)(src + (size_t)s * split_elems + 4);
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    float y[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) y[q] = __builtin_fmaf(v[q], scale[c0 + q], shift[c0 + q]);
    if constexpr (MODE == 0) {
        const size_t o = (size_t)pix * (cout * 2) + split_hi_elem(cout, c0);
        const int lo_d = split_group(cout);
        if (residual) add_split8(residual + o, lo_d, y);
        if (relu) {
#pragma unroll
            for (int q = 0; q < 8; ++q) y[q] = fmaxf(y[q], 0.f);
        }
        store_split8(out + o, lo_d, y);
    } else {
        constexpr bool F16 = MODE == 1;
        const size_t o = (size_t)pix * cout + c0;
        if (residual) {                                         // (the same additions, in the same order, as conv_igemm_mfma's epilogue)
            const uint4 rr = *(const uint4*)(residual + o);
            y[0] += unpack_lo<F16>(rr.x); y[1] += unpack_hi<F16>(rr.x);
            y[2] += unpack_lo<F16>(rr.y); y[3] += unpack_hi<F16>(rr.y);
            y[4] += unpack_lo<F16>(rr.z); y[5] += unpack_hi<F16>(rr.z);
            y[6] += unpack_lo<F16>(rr.w); y[7] += unpack_hi<F16>(rr.w);
        }
        if (relu) {
#pragma unroll
            for (int q = 0; q < 8; ++q) y[q] = fmaxf(y[q], 0.f);
        }
        uint4 r;
        r.x = pack2<F16>(y[0], y[1]); r.y = pack2<F16>(y[2], y[3]);
        r.z = pack2<F16>(y[4], y[5]); r.w = pack2<F16>(y[6], y[7]);
        *(uint4*)(out + o) = r;
    }
}

hipError_t launch_splitk_finish(const float* ws, int splits, long split_elems, long pixels, int cout, const float* scale, const float* shift,
                                const void* residual, int relu, void* out, int precision, hipStream_t s)
{
    const long n_groups = pixels * (cout / 8);
    const dim3 grid((unsigned)((n_groups + 255) / 256));
    if (precision == kF16X3)
        hipLaunchKernelGGL(splitk_finish<0>, grid, dim3(256), 0, s, ws, splits, split_elems, n_groups, cout, scale, shift, (const uint16_t*)residual, relu, (uint16_t*)out);
    else if (precision == kF16)
        hipLaunchKernelGGL(splitk_finish<1>, grid, dim3(256), 0, s, ws, splits, split_elems, n_groups, cout, scale, shift, (const uint16_t*)residual, relu, (uint16_t*)out);
    else
        hipLaunchKernelGGL(splitk_finish<2>, grid, dim3(256), 0, s, ws, splits, split_elems, n_groups, cout, scale, shift, (const uint16_t*)residual, relu, (uint16_t*)out);
    return hipGetLastError();
}

hipError_t launch_conv(const ConvParams& p0, int precision, hipStream_t s)
{
    if (precision == kF16X3) return launch_conv_x3(p0, s);
    static const bool split_issue = !(getenv("SBBSEG_SPLIT_ISSUE") && getenv("SBBSEG_SPLIT_ISSUE")[0] == '0');      // A/B (plain 16-bit modes)
    ConvParams p = p0;
    if (!split_issue) p.variant_flags |= 8;
    if (precision == kF32) {
        const long total = (long)p.M * (p.cout / 4);
        hipLaunchKernelGGL(conv_naive_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
        return hipGetLastError();
    }
    return precision == kF16 ? launch_conv_16<true>(p, s) : launch_conv_16<false>(p, s);
}

// ------------------------------------------------------------------------------------------------
// dec_tail_fused -- the network's last decoder conv and head in one kernel.
//
//   y   = ReLU(BN(conv3x3([up2(src0: 64 ch @ H/2 x W/2), image: 3 ch @ H x W])))     32 channels, fp32
//   out = argmax(softmax(BN(conv1x1(y))))                                            u8 label per pixel
//
// The generic implicit-GEMM kernel is address-bound here (32 output channels: 16 MFMAs per 256
// gathered rows).  This kernel is a direct conv on LDS-staged tiles instead:
//   * a block owns a 16x16 output tile; its 10x10 src0 halo tile (128 B per pixel) and 18x18
//     image halo tile (16 B per pixel) are copied to LDS once with global_load_lds (double
//     buffered across the persistent tile loop) -- every source pixel is fetched once, not 4-9 x
//   * the four waves are the four output-parity classes (py,px): for a fixed parity the 3x3 taps
//     on the upsampled src0 collapse to 2x2 taps with pre-summed weights (planner.py), so each
//     wave runs 4 K-steps of 64 channels on its 8x8 sub-grid + 2 K-steps for the 9 image taps
//   * the wave's weights (6 K-steps x 32 channels) live in 96 VGPRs in MFMA A-fragment order for
//     the whole kernel; only pixel fragments are read from LDS (ds_read_b128, XOR-swizzled rows)
//   * epilogue in registers: scale/shift/ReLU in fp32, the head's 32-channel contraction with two
//     xor-shuffles, softmax, argmax; labels are assembled in LDS and stored as 16-byte rows.
// ------------------------------------------------------------------------------------------------
constexpr int kTailSrcRowPx = 16;                       // LDS row stride of the src0 halo tile (10 used): stride = 0 mod 8
constexpr int kTailSrcBytes = 10 * kTailSrcRowPx * 128;      // 20 KB
constexpr int kTailImgRowPx = 32;                       // LDS row stride of the image halo tile (18 used)
constexpr int kTailImgBytes = 18 * kTailImgRowPx * 16;       // 9 KB
constexpr int kTailBufBytes = kTailSrcBytes + kTailImgBytes;
constexpr int kTailConstBytes = 32 * 8 * 4;                  // per channel: scale, shift, head_w[NC] (row of 4 or 8 floats)
constexpr int kTailLdsBytes = 2 * kTailBufBytes + 256 + 64 + kTailConstBytes;  // + label tile + zero granule + constants

template <bool F16, int NC>
__global__ __launch_bounds__(256, 2) void dec_tail_fused(const TailParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lbl_tile = smem + 2 * kTailBufBytes;                 // [16][16] u8
    char* zero_gran = lbl_tile + 256;                          // 16 zero bytes (image taps 9..15)
    float* cst = (float*)(zero_gran + 64);                     // [32 channels][8]: scale, shift, head_w[0..3]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int py = wave >> 1, px = wave & 1;
    const int frow = lane & 15, fg = lane >> 4;

    const int H = 2 * p.PH, W = 2 * p.PW;
    const int tiles_x = W / 16, tiles_y = H / 16;
    const int tiles_per_patch = tiles_x * tiles_y;
    // owned-region launch (TailParams::ttab, region.h): the tiles are the table's entries -- (patch, output origin / 2), x origins multiples
    // of 16 (label rows are stored 16 bytes at a time) -- instead of every 16 x 16 tile of every patch
    const int n_tiles = p.ttab ? p.n_tab : p.n * tiles_per_patch;
    const __attribute__((address_space(4))) uint32_t* ttab = (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)p.ttab;
    auto tile_origin = [&](int tile, int& n, int& y0, int& x0) __attribute__((always_inline)) {
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
    // XCD-contiguous walk (grid = a multiple of 8 blocks): XCD x = block % 8 owns tiles [x * per_xcd, (x + 1) * per_xcd), so the
    // halo pixels neighbouring tiles share are fetched into one L2 once instead of once per XCD (a round-robin walk
    // re-fetched them from HBM: 1.8x the input bytes, L2 hit rate 2 %)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, GX = gridDim.x >> 3;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int xcd_lo = xcd * per_xcd, xcd_hi = min(n_tiles, xcd_lo + per_xcd);
    const int my_tiles = xcd_lo + slot < xcd_hi ? (xcd_hi - xcd_lo - slot + GX - 1) / GX : 0;
    if (my_tiles <= 0) return;
    auto tile_at = [&](int it) __attribute__((always_inline)) -> int { return xcd_lo + slot + it * GX; };
    if (tid < 4) ((uint32_t*)zero_gran)[tid] = 0u;
    constexpr int CR = NC <= 2 ? 4 : 8;                        // floats per constant row
    if (tid < 32) {                                            // epilogue constants stay in LDS (VGPRs hold the weights)
        // channel c = fg*8+q lives in row q*4+fg: the four fg lanes groups of one read hit different banks
        float* row = cst + ((tid & 7) * 4 + (tid >> 3)) * CR;
        row[0] = p.scale[tid];
        row[1] = p.shift[tid];
        for (int c = 0; c < CR - 2; ++c) row[2 + c] = c < p.classes ? p.head_w[tid * p.classes + c] : 0.f;
    }

    // ---- this wave's weights, resident in registers
    bf16x8_t wf[kTailKSteps * 4];
    {
        const uint4* src = (const uint4*)p.wfrag + (size_t)(wave * kTailKSteps * 4) * 64 + lane;
#pragma unroll
        for (int f = 0; f < kTailKSteps * 4; ++f) wf[f] = __builtin_bit_cast(bf16x8_t, src[(size_t)f * 64]);
    }
    float hsc[NC], hsh[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { hsc[c] = c < p.classes ? p.head_scale[c] : 0.f; hsh[c] = c < p.classes ? p.head_shift[c] : 0.f; }

    // ---- LDS read offsets (per lane, tile independent): a fragment read then costs no address arithmetic (the kernel is VALU-issue
    // bound at two blocks per CU: 353 of its 858 vector instructions per tile were these).  Pixel block ni of this lane sits at
    //   src0:  hp = hp0 + ni * 32 (+ tap: (ks >> 1) * 16 + (ks & 1)); slot of granule kk * 4 + fg = (kk * 4 + fg) ^ ((hp0 + (ks & 1)) & 7)
    //   image: pixel ib0 + ni * 128 (+ tap offset of this lane's k-group; taps 9..15 carry zero weights: any finite pixel will do)
    const int hp0 = ((frow >> 3) + py) * kTailSrcRowPx + (frow & 7) + px;
    int src_t[2][2];                                           // [kk][ks & 1]
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int e = 0; e < 2; ++e) src_t[kk][e] = hp0 * 128 + (((kk * 4 + fg) ^ ((hp0 + e) & 7)) << 4);
    const int ib0 = (2 * (frow >> 3) + py) * kTailImgRowPx + 2 * (frow & 7) + px;
    int img_t[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int t = s * 8 + kk * 4 + fg;
            img_t[s][kk] = (ib0 + (t < 9 ? (t / 3) * kTailImgRowPx + (t % 3) : 0)) * 16;
        }

    auto issue_tile = [&](int tile, int buf) __attribute__((always_inline)) {
        int n, y0, x0;
        tile_origin(tile, n, y0, x0);
        char* lds_src = smem + buf * kTailBufBytes;
        char* lds_img = lds_src + kTailSrcBytes;
        // src0 halo: 10 rows x 16 px (10 needed) x 8 granules = 20 wave-instructions of 8 px
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int ii = wave + 4 * j;
            const int r = ii >> 1, c = (ii & 1) * 8 + (lane >> 3);
            const int g = (lane & 7) ^ (lane >> 3);                       // (hp & 7) == (c & 7) == lane >> 3
            const int Y = (y0 >> 1) - 1 + r, X = (x0 >> 1) - 1 + c;
            const bool ok = ((unsigned)Y < (unsigned)p.PH) & ((unsigned)X < (unsigned)p.PW) & (c < 10);
            uint32_t off = (uint32_t)((n * p.PH + Y) * p.PW + X) * 128u + (uint32_t)(g * 16 + kZeroHeaderBytes);
            off = ok ? off : 0u;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(p.src0 + off), (LDS_AS void*)(lds_src + ii * 1024), 16, 0, 0);
        }
        // image halo: 18 rows x 32 px (18 needed) x 16 B = 9 wave-instructions of 2 rows
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ii = wave + 4 * j;
            if (ii < 9) {
                const int r = ii * 2 + (lane >> 5), c = lane & 31;
                const int Y = y0 - 1 + r, X = x0 - 1 + c;
                const bool ok = ((unsigned)Y < (unsigned)H) & ((unsigned)X < (unsigned)W) & (c < 18);
                uint32_t off = (uint32_t)((n * H + Y) * W + X) * 16u + (uint32_t)kZeroHeaderBytes;
                off = ok ? off : 0u;
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(p.img + off), (LDS_AS void*)(lds_img + ii * 1024), 16, 0, 0);
            }
        }
    };

    issue_tile(tile_at(0), 0);
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = tile_at(it);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                        // tile `it` landed; everyone is done with tile it-1
        if (it + 1 < my_tiles) issue_tile(tile_at(it + 1), (it + 1) & 1);

        const char* lds_src = smem + (it & 1) * kTailBufBytes;
        const char* lds_img = lds_src + kTailSrcBytes;
        f32x4_t acc[2][4];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        // 12 half-K-steps: 0..7 = src0 (K-step ks = tap (ty,tx) of the parity's 2x2 window, 64 channels,
        // two k-halves), 8..11 = image (two K-steps, one 16-byte granule per tap).  The pixel fragments
        // of step h+1 are requested before the MFMAs of step h (explicit register double buffer: the
        // LDS latency otherwise sits exposed in front of every group of 8 MFMAs).
        auto load_b = [&](int h, bf16x8_t (&b)[4]) __attribute__((always_inline)) {
            if (h < 8) {
                const int ks = h >> 1, kk = h & 1;
                const char* a = lds_src + src_t[kk][ks & 1];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) b[ni] = *(const bf16x8_t*)(a + (ni * 32 + (ks >> 1) * kTailSrcRowPx + (ks & 1)) * 128);
            } else {
                const char* a = lds_img + img_t[(h - 8) >> 1][(h - 8) & 1];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) b[ni] = *(const bf16x8_t*)(a + ni * 128 * 16);
            }
        };
        bf16x8_t b0[4], b1[4];
        load_b(0, b0);
#pragma unroll
        for (int h = 0; h < 12; h += 2) {
            load_b(h + 1, b1);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[mi][ni] = mfma16<F16>(wf[h * 2 + mi], b0[ni], acc[mi][ni]);
            if (h + 2 < 12) load_b(h + 2, b0);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[mi][ni] = mfma16<F16>(wf[(h + 1) * 2 + mi], b1[ni], acc[mi][ni]);
        }

        // ---- epilogue: BN/ReLU, head, softmax, argmax
        int n, ty0, tx0;
        tile_origin(tile, n, ty0, tx0);
        // (channel constants are read once per tile -- q outer, the four pixel blocks inner -- not once per pixel block)
        float lg[4][NC];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int c = 0; c < NC; ++c) lg[ni][c] = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float* row = cst + (q * 4 + fg) * CR;
            const float4 c0 = *(const float4*)row;                                // scale, shift, hw0, hw1
            float2 c1 = make_float2(0.f, 0.f);
            if constexpr (NC > 2) c1 = *(const float2*)(row + 4);                 // hw2, hw3
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const float v = q < 4 ? acc[0][ni][q] : acc[1][ni][q - 4];
                const float yq = fmaxf(v * c0.x + c0.y, 0.f);
                lg[ni][0] = fmaf(yq, c0.z, lg[ni][0]);
                if constexpr (NC > 1) lg[ni][1] = fmaf(yq, c0.w, lg[ni][1]);
                if constexpr (NC > 2) {
                    lg[ni][2] = fmaf(yq, c1.x, lg[ni][2]);
                    lg[ni][3] = fmaf(yq, c1.y, lg[ni][3]);
                }
            }
        }
        // k-group reduction as a two-step butterfly: lane (frow, fg) ends up with the logits of pixel block ni = fg, pixel frow -- the
        // softmax runs once on 64 lanes instead of four times on 16 (the sums associate as before: own + fg^1, then + fg^2)
        {
            const bool o1 = fg & 1, o2 = fg & 2;
            float logit[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float k0 = (o1 ? lg[1][c] : lg[0][c]) + __shfl_xor(o1 ? lg[0][c] : lg[1][c], 16);
                const float k1 = (o1 ? lg[3][c] : lg[2][c]) + __shfl_xor(o1 ? lg[2][c] : lg[3][c], 16);
                const float a = (o2 ? k1 : k0) + __shfl_xor(o2 ? k0 : k1, 32);
                logit[c] = a * hsc[c] + hsh[c];
            }
            float mx = -3.0e38f;
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (c < p.classes) mx = fmaxf(mx, logit[c]);
            float pr[NC], sum = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) { pr[c] = c < p.classes ? expf(logit[c] - mx) : 0.f; sum += pr[c]; }
            int best = 0;
            float bestp = -1.f;
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (c < p.classes) {
                    pr[c] = pr[c] / sum;
                    if (pr[c] > bestp) { bestp = pr[c]; best = c; }              // first maximum wins (np.argmax)
                }
            const int i = fg * 16 + frow;
            const int oy = 2 * (i >> 3) + py, ox = 2 * (i & 7) + px;             // inside the 16x16 tile
            lbl_tile[oy * 16 + ox] = (char)best;
            if (p.probs) {
                float* dst = p.probs + ((size_t)(n * H + ty0 + oy) * W + tx0 + ox) * p.classes;
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (c < p.classes) dst[c] = pr[c];
            }
        }
        __syncthreads();
        if (tid < 16)
            *(uint4*)(p.labels + (size_t)(n * H + ty0 + tid) * W + tx0) = *(const uint4*)(lbl_tile + tid * 16);
    }
}

// ------------------------------------------------------------------------------------------------
// dec_tail_fused_x3ps -- the same tail in the split mode (kF16X3): three MFMAs per product (lo*hi, hi*lo, hi*hi), everything after
// the accumulators as in dec_tail_fused.  The generic kernel needs 5.1 ms per 140 patches for this layer (32 output channels:
// 24 MFMAs per 288 staged rows).
//   * tile = 16 x 16 output pixels, one block of EIGHT waves per CU: wave = (output-parity class, half of the 32 output channels);
//     its weights are hi + lo fragments in registers (80 VGPRs), two waves share a SIMD
//   * src0 halo: 10 x 10 pixels (rows of 16) x 256 B ([32 hi][32 lo][32 hi][32 lo]); pixel hp keeps granule g at slot
//     (g + 2 hp) & 15.  A 16-lane group of ds_read_b128 holds two k-groups (fg = a, a + 1) of eight pixels each whose hp are eight
//     consecutive residues: the rotation sends one k-group to the eight even slots and the other to the eight odd ones (an XOR
//     swizzle collided two-way in every group: PMC SQ_LDS_BANK_CONFLICT 86 %)
//   * image halo: 18 x 18 pixels x 16 B.  In the split mode the C8 input form keeps lo(ch 0..2) a second time in the unused
//     channel slots 4..6 of its hi plane (write_split_input), so the first granule of a pixel is [h0 h1 h2 0 | l0 l1 l2 0]: one
//     16-byte load fetches both planes, a k-group of 8 is TWO taps x 4 channel slots, and the nine image taps take 2 half-K-steps
//     (K = 36 of 64) instead of the 4 of one-tap-per-granule (K = 72 of 128): 120 instead of 144 MFMAs per wave and tile.
//     LDS row y = 32 units of 16 B; pixel x sits at unit ((x >> 1) + 12 (x & 1) + 16 - 4 (y & 3)) & 31 (brute-forced over this
//     family: 1.31 LDS cycles per conflict-free cycle on the image reads, which are 12 of 76 reads per wave and tile)
// ------------------------------------------------------------------------------------------------
constexpr int kT3SrcBytes = 10 * 16 * 256;              // 40 KB: 10 rows x 16 pixels (10 used) x 256 B
constexpr int kT3ImgBytes = 18 * 32 * 16;               // 9 KB: 18 rows x 32 units (18 used) x 16 B
constexpr int kT3BufBytes = kT3SrcBytes + kT3ImgBytes;
constexpr int kT3HalfSteps = 10;                        // 4 taps x 64 channels of src0 = 8 half-K-steps of 32, + 2 for the 9 image taps

// The two waves of every SIMD run half a tile out of phase.  With all eight waves loading, multiplying and running the epilogue at
// the same moments (round 3's first eight-wave form) the three parts simply added up (tools/probes/tail_probe.hip, 140 patches:
// tile loads alone 0.64 ms, MFMAs alone 1.23, epilogue alone 0.83; all of it 2.48).
// Here the group A = waves 0-3 (channel half 0) does   main loop(t) -> tile loads(t+1) -> BN / ReLU / partial logits(t) -> part[t & 1],
// and the group B = waves 4-7 (channel half 1) does    label store(t-2), epilogue(t-1) incl. softmax, main loop(t)
// between two consecutive block barriers.  Wave w and wave w + 4 share a SIMD (waves go to SIMDs round-robin), so while A's wave
// keeps the MFMA pipe busy B's wave issues the address arithmetic, the DMA loads and the epilogue VALU work, and the other way
// round in the second half of the step.  B finishes the pixels (A's partial logits come through LDS, written one step earlier).
// The k-group reduction is a two-step butterfly that leaves ONE pixel per lane (pixel block ni = fg), so the softmax runs once on
// 64 lanes instead of four times on 16.
constexpr int kT3PsPartBytes = 2 * 4 * 64 * 4 * 4;        // [step parity][4 pixel parities][64 lanes][<= 4 classes] partial logits
constexpr int kT3PsLdsBytes = 2 * kT3BufBytes + 2 * 256 + kTailConstBytes + kT3PsPartBytes;
static_assert(kT3PsLdsBytes <= 160 * 1024, "x3 tail: LDS");

template <int NC>
__global__ __launch_bounds__(512, 2) void dec_tail_fused_x3ps(const TailParams p)
{
    constexpr bool F16 = true;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lbl_tile = smem + 2 * kT3BufBytes;                   // [2][16][16] u8
    float* cst = (float*)(lbl_tile + 2 * 256);                 // [32 channels][CR]: scale, shift, head_w
    float* part = (float*)((char*)cst + kTailConstBytes);      // [2][4 parities][64 lanes][NC]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int par = wave & 3, mh = wave >> 2;                  // parity class; channel half = wave group (A = 0, B = 1)
    const int py = par >> 1, px = par & 1;
    const int frow = lane & 15, fg = lane >> 4;

    const int H = 2 * p.PH, W = 2 * p.PW;
    const int tiles_x = W / 16, tiles_y = H / 16;
    const int tiles_per_patch = tiles_x * tiles_y;
    // owned-region launch (TailParams::ttab): see dec_tail_fused
    const int n_tiles = p.ttab ? p.n_tab : p.n * tiles_per_patch;
    const __attribute__((address_space(4))) uint32_t* ttab = (const __attribute__((address_space(4))) uint32_t*)(uintptr_t)p.ttab;
    auto tile_origin = [&](int tile, int& n, int& y0, int& x0) __attribute__((always_inline)) {
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
    // XCD-contiguous walk, as in the other tail kernels
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, GX = gridDim.x >> 3;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int xcd_lo = xcd * per_xcd, xcd_hi = min(n_tiles, xcd_lo + per_xcd);
    const int my_tiles = xcd_lo + slot < xcd_hi ? (xcd_hi - xcd_lo - slot + GX - 1) / GX : 0;
    if (my_tiles <= 0) return;
    auto tile_at = [&](int it) __attribute__((always_inline)) -> int { return xcd_lo + slot + it * GX; };
    constexpr int CR = NC <= 2 ? 4 : 8;
    if (tid < 32) {
        float* row = cst + ((tid & 7) * 4 + (tid >> 3)) * CR;
        row[0] = p.scale[tid];
        row[1] = p.shift[tid];
        for (int c = 0; c < CR - 2; ++c) row[2 + c] = c < p.classes ? p.head_w[tid * p.classes + c] : 0.f;
    }

    // ---- this wave's weights: [plane hi|lo][half-K-step 10] fragments of row block mh (wfrag = per class [hi | lo][10][mi 2])
    constexpr int NH = kT3HalfSteps;
    bf16x8_t whi[NH], wlo[NH];
    {
        const uint4* src = (const uint4*)p.wfrag + (size_t)(par * 2 * NH * 2) * 64 + lane;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            whi[h] = __builtin_bit_cast(bf16x8_t, src[(size_t)(h * 2 + mh) * 64]);
            wlo[h] = __builtin_bit_cast(bf16x8_t, src[(size_t)(NH * 2 + h * 2 + mh) * 64]);
        }
    }
    __syncthreads();                                           // (cst written)
    float4 kc0[4];                                             // this wave's channel constants: [q] = scale, shift, hw0, hw1 of channel fg * 8 + mh * 4 + q
    float2 kc1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float* row = cst + ((mh * 4 + q) * 4 + fg) * CR;
        kc0[q] = *(const float4*)row;
        kc1[q] = make_float2(0.f, 0.f);
        if constexpr (NC > 2) kc1[q] = *(const float2*)(row + 4);
    }
    float hsc[NC], hsh[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { hsc[c] = c < p.classes ? p.head_scale[c] : 0.f; hsh[c] = c < p.classes ? p.head_shift[c] : 0.f; }

    // Tile-invariant LDS read offsets, so that a fragment read costs no address arithmetic.  Pixel block ni of this lane sits at
    //   src0:  hp = hp0 + ni * 32 (+ tap: (ks >> 1) * 16 + (ks & 1)),  slot of granule G = (G + 2 hp) & 15 = (s0 + D) & 15 with
    //          s0 = (fg + 2 hp0) & 15 per lane and D = kk * 4 + 8 * lo + 2 * (ks & 1) known at compile time (even: 8 table entries);
    //   image: row 4 ni + yb (the rotation depends on y & 3 only), tap t = 8 s2 + 2 fg + j of image half-step s2, j = 0 / 1.
    // Everything that depends on ni / ks is a multiple of 256 (2048) bytes and rides in the instruction's immediate offset.
    const int hp0 = ((frow >> 3) + py) * 16 + (frow & 7) + px;
    const int s0 = (fg + 2 * hp0) & 15;
    int src_t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) src_t[e] = hp0 * 256 + (((s0 + 2 * e) & 15) << 4);
    int img_t[3];                                              // [s2 = 0: j = 0, 1][s2 = 1: j = 0]
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int t0 = (e >> 1) * 8 + 2 * fg + (e & 1);
        const int t = t0 < 9 ? t0 : 1;                         // taps 9..15 do not exist: zero weights, any finite pixel will do
        const int yb = 2 * (frow >> 3) + py + t / 3, x = 2 * (frow & 7) + px + t % 3;
        img_t[e] = yb * 512 + ((((x >> 1) + 12 * (x & 1) + 16 - 4 * (yb & 3)) & 31) << 4);
    }

    // Halo loads (group A): wave `par` issues the src0 pieces ii = par + 4 j (j = 0..9: halo row j, halo columns 4 par .. 4 par + 3) and
    // the image pieces q = par + 4 j (< 9).  `buffer_load ... lds` with a per-tile resource (this patch's image, one pixel of
    // bias so that the lane part is never negative): the lane offset of a src0 piece does not depend on j or on the tile, the row
    // rides in the scalar offset, and anything outside the image sets bit 31 of the lane offset (past num_records: the hardware
    // writes zeros).  ~3 vector instructions per piece instead of the ~20 of per-lane global addresses.
    const int c_src = par * 4 + (lane >> 4);                   // halo column of this lane's src0 pixel
    const uint32_t voff_src = (uint32_t)(c_src * 256 + ((((lane & 15) - 2 * c_src) & 15) << 4));      // slot s of pixel hp holds granule (s - 2 hp) & 15
    // image piece q: lane l fills unit l & 31 of halo row 2 q + (l >> 5); (2 q) & 3 = 2 (par & 1) for every piece of this wave
    const int img_v = ((lane & 31) - (16 - 4 * ((2 * par + (lane >> 5)) & 3))) & 31;
    const bool img_xok = img_v < 9 || (img_v >= 12 && img_v < 21);
    const int img_x = img_v < 9 ? 2 * img_v : 2 * (img_v - 12) + 1;
    const uint32_t voff_img = (uint32_t)(((lane >> 5) * W + img_x) * 32);
    const uint32_t src_img_bytes = (uint32_t)(p.PH * p.PW) * 256u, img_img_bytes = (uint32_t)(H * W) * 32u;
    auto issue_src = [&](int n, int y0, int x0, int buf) __attribute__((always_inline)) {           // group A: ten pieces per wave
        char* lds_src = smem + buf * kT3BufBytes;
        const char* sbase = p.src0 + kZeroHeaderBytes - 256 + (size_t)n * src_img_bytes;
        const uint32_t vs = ((unsigned)((x0 >> 1) - 1 + c_src) < (unsigned)p.PW && c_src < 10) ? voff_src : 0x80000000u;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int Y = (y0 >> 1) - 1 + j;
            const bool yok = (unsigned)Y < (unsigned)p.PH;
            const uint32_t soff = yok ? (uint32_t)(Y * p.PW + (x0 >> 1)) * 256u : 0u;
            buffer_load_lds16(sbase, src_img_bytes + 256u, (LDS_AS void*)(lds_src + (par + 4 * j) * 1024), yok ? vs : 0x80000000u, soff);
        }
    };
    auto issue_img = [&](int n, int y0, int x0, int buf) __attribute__((always_inline)) {           // group B: two or three pieces per wave
        char* lds_img = smem + buf * kT3BufBytes + kT3SrcBytes;
        // image: piece q = par + 4 j (< 9) = halo rows 2 q, 2 q + 1 (32 units each); one row + one pixel of bias
        const char* ibase = p.img + kZeroHeaderBytes + (size_t)n * img_img_bytes - (size_t)(W + 1) * 32;
        const uint32_t vi = (img_xok && (unsigned)(x0 - 1 + img_x) < (unsigned)W) ? voff_img : 0x80000000u;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int q = par + 4 * j;
            if (q < 9) {
                const bool yok = (unsigned)(y0 - 1 + 2 * q + (lane >> 5)) < (unsigned)H;
                buffer_load_lds16(ibase, img_img_bytes + (uint32_t)(W + 1) * 32u, (LDS_AS void*)(lds_img + q * 1024), yok ? vi : 0x80000000u,
                                  (uint32_t)((y0 + 2 * q) * W + x0) * 32u);
            }
        }
    };

    f32x4_t acc[4];
    auto main_loop = [&](int it) __attribute__((always_inline)) {
        const char* lds_src = smem + (it & 1) * kT3BufBytes;
        const char* lds_img = lds_src + kT3SrcBytes;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const char* sb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sb[e] = lds_src + src_t[e];
        auto load_b = [&](int h, bf16x8_t (&bh)[4], bf16x8_t (&bl)[4]) __attribute__((always_inline)) {
            if (h < 8) {
                const int ks = h >> 1, kk = h & 1;
                const int dh = (kk * 8 + 2 * (ks & 1)) & 15, dl = (kk * 8 + 4 + 2 * (ks & 1)) & 15;      // src0 pixel: [32 hi][32 lo][32 hi][32 lo]
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int k = (ni * 32 + (ks >> 1) * 16 + (ks & 1)) * 256;      // immediate offset
                    bh[ni] = *(const bf16x8_t*)(sb[dh >> 1] + k);
                    bl[ni] = *(const bf16x8_t*)(sb[dl >> 1] + k);
                }
            } else {
                // image pixel = [h0 h1 h2 0 | l0 l1 l2 0]: the k-group is two taps; half-step 9 has tap 8 only (its second tap
                // multiplies zero weights: the first one's registers do)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const uint4 r0 = *(const uint4*)(lds_img + img_t[(h - 8) * 2] + ni * 2048);
                    const uint4 r1 = h == 8 ? *(const uint4*)(lds_img + img_t[1] + ni * 2048) : r0;
                    bh[ni] = __builtin_bit_cast(bf16x8_t, make_uint4(r0.x, r0.y, r1.x, r1.y));
                    bl[ni] = __builtin_bit_cast(bf16x8_t, make_uint4(r0.z, r0.w, r1.z, r1.w));
                }
            }
        };
        auto mac = [&](int h, const bf16x8_t (&bh)[4], const bf16x8_t (&bl)[4]) __attribute__((always_inline)) {
            // three sweeps over the four accumulators (small terms first)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni] = mfma16<F16>(wlo[h], bh[ni], acc[ni]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni] = mfma16<F16>(whi[h], bl[ni], acc[ni]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni] = mfma16<F16>(whi[h], bh[ni], acc[ni]);
        };
        bf16x8_t b0h[4], b0l[4], b1h[4], b1l[4];
        load_b(0, b0h, b0l);
#pragma unroll
        for (int h = 0; h < NH; h += 2) {
            load_b(h + 1, b1h, b1l);
            mac(h, b0h, b0l);
            if (h + 2 < NH) load_b(h + 2, b0h, b0l);
            mac(h + 1, b1h, b1l);
        }
    };
    // BN / ReLU / head on this wave's 16 channels, then the k-group butterfly: lane (frow, fg) ends up with the partial logits of
    // pixel block ni = fg, pixel frow
    auto partial_logits = [&](float (&tot)[NC]) __attribute__((always_inline)) {
        float lg[4][NC];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int c = 0; c < NC; ++c) lg[ni][c] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 c0 = kc0[q];                                             // scale, shift, hw0, hw1
            const float2 c1 = kc1[q];                                             // hw2, hw3
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const float yq = fmaxf(acc[ni][q] * c0.x + c0.y, 0.f);
                lg[ni][0] = fmaf(yq, c0.z, lg[ni][0]);
                if constexpr (NC > 1) lg[ni][1] = fmaf(yq, c0.w, lg[ni][1]);
                if constexpr (NC > 2) {
                    lg[ni][2] = fmaf(yq, c1.x, lg[ni][2]);
                    lg[ni][3] = fmaf(yq, c1.y, lg[ni][3]);
                }
            }
        }
        const bool o1 = fg & 1, o2 = fg & 2;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            // step 1 (partner fg ^ 1): keep the pixel blocks of this lane's parity, hand over the other two
            const float k0 = (o1 ? lg[1][c] : lg[0][c]) + __shfl_xor(o1 ? lg[0][c] : lg[1][c], 16);      // block 0 + (fg & 1)
            const float k1 = (o1 ? lg[3][c] : lg[2][c]) + __shfl_xor(o1 ? lg[2][c] : lg[3][c], 16);      // block 2 + (fg & 1)
            // step 2 (partner fg ^ 2)
            tot[c] = (o2 ? k1 : k0) + __shfl_xor(o2 ? k0 : k1, 32);
        }
    };
    // group B: finish tile `it` (its accumulators are still in this wave's registers; A's half came through part[it & 1])
    auto finish = [&](int it) __attribute__((always_inline)) {
        float tot[NC];
        partial_logits(tot);
        const float* pa = part + (((it & 1) * 4 + par) * 64 + lane) * NC;
        float logit[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) logit[c] = (pa[c] + tot[c]) * hsc[c] + hsh[c];
        float mx = -3.0e38f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < p.classes) mx = fmaxf(mx, logit[c]);
        float pr[NC], sum = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { pr[c] = c < p.classes ? expf(logit[c] - mx) : 0.f; sum += pr[c]; }
        int best = 0;
        float bestp = -1.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < p.classes) {
                pr[c] = pr[c] / sum;
                if (pr[c] > bestp) { bestp = pr[c]; best = c; }
            }
        const int i = fg * 16 + frow;
        const int oy = 2 * (i >> 3) + py, ox = 2 * (i & 7) + px;
        lbl_tile[(it & 1) * 256 + oy * 16 + ox] = (char)best;
        if (p.probs) {
            int n, ty0, tx0;
            tile_origin(tile_at(it), n, ty0, tx0);
            float* dst = p.probs + ((size_t)(n * H + ty0 + oy) * W + tx0 + ox) * p.classes;
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (c < p.classes) dst[c] = pr[c];
        }
    };
    auto store_labels = [&](int it) __attribute__((always_inline)) {       // (wave 4, after the barrier that follows finish(it))
        if (wave == 4 && lane < 16) {
            int n, ty0, tx0;
            tile_origin(tile_at(it), n, ty0, tx0);
            *(uint4*)(p.labels + (size_t)(n * H + ty0 + lane) * W + tx0) = *(const uint4*)(lbl_tile + (it & 1) * 256 + lane * 16);
        }
    };

    if (mh == 0) {
        int n, y0, x0;
        tile_origin(tile_at(0), n, y0, x0);
        issue_src(n, y0, x0, 0); issue_img(n, y0, x0, 0);
    }
    for (int it = 0; it < my_tiles; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (mh == 1) {
            if (it >= 2) store_labels(it - 2);
            if (it >= 1) finish(it - 1);
        }
        main_loop(it);
        if (mh == 0) {
            // all of the next tile's halo pieces (before the epilogue: more time to land).  Moving the image pieces to group B --
            // before or after its epilogue -- made B the longer group: a piece costs its wave 200-400 cycles there
            if (it + 1 < my_tiles) {
                int n, y0, x0;
                tile_origin(tile_at(it + 1), n, y0, x0);
                issue_src(n, y0, x0, (it + 1) & 1); issue_img(n, y0, x0, (it + 1) & 1);
            }
            float tot[NC];
            partial_logits(tot);
            float* pa = part + (((it & 1) * 4 + par) * 64 + lane) * NC;
#pragma unroll
            for (int c = 0; c < NC; ++c) pa[c] = tot[c];
        }
    }
    __syncthreads();
    if (mh == 1) {
        if (my_tiles >= 2) store_labels(my_tiles - 2);
        finish(my_tiles - 1);
    }
    __syncthreads();
    if (mh == 1) store_labels(my_tiles - 1);
}

hipError_t launch_tail(const TailParams& p, int precision, int num_cus, hipStream_t s)
{
    const int n_tiles = p.ttab ? p.n_tab : p.n * (p.PH / 8) * (p.PW / 8);
    if (n_tiles <= 0) return hipSuccess;
    const int grid = ((n_tiles < 2 * num_cus ? n_tiles : 2 * num_cus) + 7) & ~7;      // (the XCD-contiguous walk: a multiple of 8)
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kTailLdsBytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kTailLdsBytes, s, p);
        return hipSuccess;
    };
    hipError_t e;
    if (precision == kF16X3) {
        const int grid3 = ((n_tiles < num_cus ? n_tiles : num_cus) + 7) & ~7;
        auto gops = [&](auto kern) -> hipError_t {
            hipError_t e8 = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kT3PsLdsBytes);
            if (e8 != hipSuccess) return e8;
            hipLaunchKernelGGL(kern, dim3(grid3), dim3(512), kT3PsLdsBytes, s, p);
            return hipSuccess;
        };
        e = p.classes <= 2 ? gops(dec_tail_fused_x3ps<2>) : gops(dec_tail_fused_x3ps<4>);
        if (e != hipSuccess) return e;
        return hipGetLastError();
    }
    if (precision == kF16) e = p.classes <= 2 ? go(dec_tail_fused<true, 2>) : go(dec_tail_fused<true, 4>);
    else e = p.classes <= 2 ? go(dec_tail_fused<false, 2>) : go(dec_tail_fused<false, 4>);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// stem_conv_pairs -- the network's first conv (7x7, stride 2, 3 -> 64 channels) as a direct conv on an
// LDS halo tile.  The generic kernel gathers 8 differently-placed granules per 128-byte row here (the
// per-granule tap table) and re-fetches every input row ~3.5x across XCDs; this one copies the
// 37-row x 19-granule input halo of a 16x16 output tile to LDS once (12 KB, double buffered over a
// persistent tile loop), keeps all 7 x 64-channel weight fragments in 112 VGPRs, and reads one
// ds_read_b128 per (kernel row, 16-pixel output row): lane (pixel x, granule g) reads granule x + g
// of input row 2y + ky -- consecutive 16-byte slots, conflict-free for any row stride.  Each wave
// owns 4 output rows x 64 channels; epilogue = scale/shift(/ReLU), whole-line 16-bit stores.
// ------------------------------------------------------------------------------------------------
constexpr int kStemRowSlots = 20;                       // granules per LDS row (19 used)
constexpr int kStemRows = 37;                           // 2*16 + 5
constexpr int kStemInstr = (kStemRows * kStemRowSlots + 63) / 64;      // wave-instructions per halo tile (12)
constexpr int kStemBufBytes = kStemInstr * 1024;
constexpr int kStemLdsBytes = 2 * kStemBufBytes + 512;  // + scale[64], shift[64]
constexpr int kStemStores = 8;                          // store instructions per wave per tile

template <bool F16>
__global__ __launch_bounds__(256, 2) void stem_conv_pairs(const StemParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int tiles_x = p.Wo / 16, tiles_y = p.Ho / 16;
    const int tiles_per_patch = tiles_x * tiles_y;
    const int n_tiles = p.n * tiles_per_patch;
    const int G = gridDim.x;
    const int my_tiles = (n_tiles - (int)blockIdx.x + G - 1) / G;
    if (my_tiles <= 0) return;

    // epilogue constants live in LDS: a global load in the epilogue would queue behind the next tile's
    // halo loads (vmcnt retires in order) and stall every tile for one memory round trip
    float* cst = (float*)(smem + 2 * kStemBufBytes);
    if (tid < 64) { cst[tid] = p.scale[tid]; cst[64 + tid] = p.shift[tid]; }
    bf16x8_t wf[7][4];
    {
        const uint4* src = (const uint4*)p.wfrag + lane;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) wf[ky][mi] = __builtin_bit_cast(bf16x8_t, src[(size_t)(ky * 4 + mi) * 64]);
    }

    auto issue_tile = [&](int tile, int buf) __attribute__((always_inline)) {
        const int n = tile / tiles_per_patch;
        const int rem = tile - n * tiles_per_patch;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        char* lds = smem + buf * kStemBufBytes;
#pragma unroll
        for (int j = 0; j < kStemInstr / 4; ++j) {
            const int ii = wave + 4 * j;
            const int slot = ii * 64 + lane;
            const int r = slot / kStemRowSlots, cc = slot - r * kStemRowSlots;
            const int Y = 32 * ty + r, X = 16 * tx + cc;          // (granule 19 of a row is never read: whatever lies there)
            uint32_t off = (uint32_t)((n * p.PHt + Y) * p.PWt + X) * 16u + (uint32_t)kZeroHeaderBytes;
            off = r < kStemRows ? off : 0u;
            glds16_hidden(p.pairs + off, lds + ii * 1024);
        }
    };

    const bool hi = (frow & 8) != 0;
    issue_tile(blockIdx.x, 0);
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = blockIdx.x + it * G;
        // the halo loads of tile `it` are older than the previous tile's stores: leave those in flight
        if (it == 0) wait_vmcnt<0>();
        else wait_vmcnt<kStemStores>();
        __syncthreads();                                        // tile `it` landed; everyone is done with tile it-1
        if (it + 1 < my_tiles) issue_tile(tile + G, (it + 1) & 1);

        const char* lds = smem + (it & 1) * kStemBufBytes;
        f32x4_t acc[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int row = 2 * (wave * 4 + ni) + ky;
                const bf16x8_t b = *(const bf16x8_t*)(lds + (row * kStemRowSlots + frow + fg) * 16);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = mfma16<F16>(wf[ky][mi], b, acc[mi][ni]);
            }
        }

        // ---- epilogue (same lane swap as conv_igemm_mfma's full-tile path: 8 pixels x 128 B per store)
        const int n = tile / tiles_per_patch;
        const int rem = tile - n * tiles_per_patch;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int oy = ty * 16 + wave * 4 + ni;
            const size_t pix0 = ((size_t)n * p.Ho + oy) * p.Wo + tx * 16 + (frow & 7);
            uint4 r[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c0 = h * 32 + fg * 8;
                float sc[8], sh[8], y[8];
                *(float4*)&sc[0] = *(const float4*)(cst + c0);
                *(float4*)&sc[4] = *(const float4*)(cst + c0 + 4);
                *(float4*)&sh[0] = *(const float4*)(cst + 64 + c0);
                *(float4*)&sh[4] = *(const float4*)(cst + 64 + c0 + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    y[q] = acc[2 * h][ni][q] * sc[q] + sh[q];                             // (contracted to one fma; stem_pool<false> states the same arithmetic)
                    y[4 + q] = acc[2 * h + 1][ni][q] * sc[4 + q] + sh[4 + q];
                }
                if (p.relu) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) y[q] = fmaxf(y[q], 0.f);
                }
                r[h].x = pack2<F16>(y[0], y[1]); r[h].y = pack2<F16>(y[2], y[3]);
                r[h].z = pack2<F16>(y[4], y[5]); r[h].w = pack2<F16>(y[6], y[7]);
            }
            uint4 give, recv, st0, st1;
            give.x = hi ? r[0].x : r[1].x; give.y = hi ? r[0].y : r[1].y;
            give.z = hi ? r[0].z : r[1].z; give.w = hi ? r[0].w : r[1].w;
            recv.x = row_ror8(give.x); recv.y = row_ror8(give.y);
            recv.z = row_ror8(give.z); recv.w = row_ror8(give.w);
            st0.x = hi ? recv.x : r[0].x; st0.y = hi ? recv.y : r[0].y;
            st0.z = hi ? recv.z : r[0].z; st0.w = hi ? recv.w : r[0].w;
            st1.x = hi ? r[1].x : recv.x; st1.y = hi ? r[1].y : recv.y;
            st1.z = hi ? r[1].z : recv.z; st1.w = hi ? r[1].w : recv.w;
            const int cst = (hi ? 32 : 0) + fg * 8;
            *(uint4*)((uint16_t*)p.out + pix0 * 64 + cst) = st0;
            *(uint4*)((uint16_t*)p.out + (pix0 + 8) * 64 + cst) = st1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// stem_conv_pairs_x3 -- the stem in the split mode (kF16X3): a PAIRS granule is 32 bytes ([2 px x 4 ch] hi, then lo); the
// halo tile is kept as two LDS images (hi, lo) with the plain kernel's conflict-free slot layout; hi + lo weight fragments
// in 224 VGPRs (one block per CU), three MFMAs per product, outputs split again ([64 hi][64 lo] per pixel).
// ------------------------------------------------------------------------------------------------
constexpr int kStemX3BufBytes = 2 * kStemBufBytes;      // hi image | lo image
constexpr int kStemX3LdsBytes = 2 * kStemX3BufBytes + 512;

__global__ __launch_bounds__(256, 1) void stem_conv_pairs_x3(const StemParams p)
{
    constexpr bool F16 = true;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int tiles_x = p.Wo / 16, tiles_y = p.Ho / 16;
    const int tiles_per_patch = tiles_x * tiles_y;
    const int n_tiles = p.n * tiles_per_patch;
    const int G = gridDim.x;
    const int my_tiles = (n_tiles - (int)blockIdx.x + G - 1) / G;
    if (my_tiles <= 0) return;
    float* cst = (float*)(smem + 2 * kStemX3BufBytes);
    if (tid < 64) { cst[tid] = p.scale[tid] * p.wmul; cst[64 + tid] = p.shift[tid]; }
    bf16x8_t whi[7][4], wlo[7][4];                      // wfrag = [hi | lo][7 ky][4 mi][64 lanes]
    {
        const uint4* src = (const uint4*)p.wfrag + lane;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                whi[ky][mi] = __builtin_bit_cast(bf16x8_t, src[(size_t)(ky * 4 + mi) * 64]);
                wlo[ky][mi] = __builtin_bit_cast(bf16x8_t, src[(size_t)(28 + ky * 4 + mi) * 64]);
            }
    }

    auto issue_tile = [&](int tile, int buf) __attribute__((always_inline)) {
        const int n = tile / tiles_per_patch;
        const int rem = tile - n * tiles_per_patch;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        char* lds = smem + buf * kStemX3BufBytes;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < kStemInstr / 4; ++j) {
                const int ii = wave + 4 * j;
                const int slot = ii * 64 + lane;
                const int r = slot / kStemRowSlots, cc = slot - r * kStemRowSlots;
                const int Y = 32 * ty + r, X = 16 * tx + cc;
                uint32_t off = (uint32_t)((n * p.PHt + Y) * p.PWt + X) * 32u + (uint32_t)(pl * 16 + kZeroHeaderBytes);
                off = r < kStemRows ? off : 0u;
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(p.pairs + off), (LDS_AS void*)(lds + pl * kStemBufBytes + ii * 1024), 16, 0, 0);
            }
    };

    issue_tile(blockIdx.x, 0);
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = blockIdx.x + it * G;
        wait_vmcnt<0>();
        __syncthreads();
        if (it + 1 < my_tiles) issue_tile(tile + G, (it + 1) & 1);

        const char* lds = smem + (it & 1) * kStemX3BufBytes;
        f32x4_t acc[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int row = 2 * (wave * 4 + ni) + ky;
                const int at = (row * kStemRowSlots + frow + fg) * 16;
                const bf16x8_t bh = *(const bf16x8_t*)(lds + at);
                const bf16x8_t bl = *(const bf16x8_t*)(lds + kStemBufBytes + at);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    acc[mi][ni] = mfma16<F16>(wlo[ky][mi], bh, acc[mi][ni]);
                    acc[mi][ni] = mfma16<F16>(whi[ky][mi], bl, acc[mi][ni]);
                    acc[mi][ni] = mfma16<F16>(whi[ky][mi], bh, acc[mi][ni]);
                }
            }
        }

        const int n = tile / tiles_per_patch;
        const int rem = tile - n * tiles_per_patch;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int oy = ty * 16 + wave * 4 + ni;
            const size_t pix = ((size_t)n * p.Ho + oy) * p.Wo + tx * 16 + frow;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c0 = h * 32 + fg * 8;
                float sc[8], sh[8], y[8];
                *(float4*)&sc[0] = *(const float4*)(cst + c0);
                *(float4*)&sc[4] = *(const float4*)(cst + c0 + 4);
                *(float4*)&sh[0] = *(const float4*)(cst + 64 + c0);
                *(float4*)&sh[4] = *(const float4*)(cst + 64 + c0 + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    y[q] = __builtin_fmaf(acc[2 * h][ni][q], sc[q], sh[q]);
                    y[4 + q] = __builtin_fmaf(acc[2 * h + 1][ni][q], sc[4 + q], sh[4 + q]);
                }
                if (p.relu) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) y[q] = fmaxf(y[q], 0.f);
                }
                store_split8((uint16_t*)p.out + pix * 128 + split_hi_elem(64, c0), 32, y);
            }
        }
    }
}

hipError_t launch_stem(const StemParams& p, int precision, int num_cus, hipStream_t s)
{
    const int n_tiles = p.n * (p.Ho / 16) * (p.Wo / 16);
    const int grid = n_tiles < 2 * num_cus ? n_tiles : 2 * num_cus;
    if (precision == kF16X3) {
        hipLaunchKernelGGL(stem_conv_pairs_x3, dim3(n_tiles < num_cus ? n_tiles : num_cus), dim3(256), kStemX3LdsBytes, s, p);
        return hipGetLastError();
    }
    if (precision == kF16) hipLaunchKernelGGL(stem_conv_pairs<true>, dim3(grid), dim3(256), kStemLdsBytes, s, p);
    else hipLaunchKernelGGL(stem_conv_pairs<false>, dim3(grid), dim3(256), kStemLdsBytes, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// conv3x3_c64_direct -- 3x3 / stride 1 / pad 1, 64 -> 64 channels (three layers at 111x111 in the sbb
// nets).  The implicit-GEMM kernel is staging-bound here: 64 output channels amortise a gathered pixel
// row over 4 MFMAs only, and every pixel row is re-staged for each of the 9 taps.  Same recipe as the
// stem: a block owns an 8 x 16 output tile, copies its 10 x 18 pixel halo (128 B per pixel, XOR-swizzled
// granules, 23 KB, double buffered over a persistent tile loop) once, and runs all 9 taps from it.  Wave
// (wp, wc) owns 4 output rows x 32 channels and keeps its 9 x 2 x 2 weight fragments in 144 VGPRs.
// ------------------------------------------------------------------------------------------------
constexpr int kD64HaloW = 18, kD64HaloH = 10;
constexpr int kD64Rows = kD64HaloW * kD64HaloH;                 // 180 pixel rows of 128 B
constexpr int kD64Instr = 24;                                   // wave-instructions of 8 rows (192 >= 180)
constexpr int kD64BufBytes = kD64Instr * 1024;
constexpr int kD64LdsBytes = 2 * kD64BufBytes + 512;            // + scale[64], shift[64]

template <bool F16>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_direct(const Direct64Params p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wc = wave & 1;
    const int frow = lane & 15, fg = lane >> 4;
    const int tiles_x = (p.W + 15) / 16, tiles_y = (p.H + 7) / 8;
    const int tiles_per_patch = tiles_x * tiles_y;
    const int n_tiles = p.n * tiles_per_patch;
    // XCD-contiguous walk (grid = a multiple of 8 blocks), as in the tail kernels
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, GX = gridDim.x >> 3;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int xcd_lo = xcd * per_xcd, xcd_hi = min(n_tiles, xcd_lo + per_xcd);
    const int my_tiles = xcd_lo + slot < xcd_hi ? (xcd_hi - xcd_lo - slot + GX - 1) / GX : 0;
    if (my_tiles <= 0) return;
    auto tile_at = [&](int it) __attribute__((always_inline)) -> int { return xcd_lo + slot + it * GX; };
    float* cst = (float*)(smem + 2 * kD64BufBytes);
    if (tid < 64) { cst[tid] = p.scale[tid]; cst[64 + tid] = p.shift[tid]; }

    bf16x8_t wf[9][2][2];                                       // [tap][kk][mi of this wave]
    {
        const uint4* src = (const uint4*)p.wfrag + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    wf[t][kk][m] = __builtin_bit_cast(bf16x8_t, src[(size_t)((t * 2 + kk) * 4 + wc * 2 + m) * 64]);
    }

    const int lrow = lane >> 3;
    const int gsrc = (lane & 7) ^ lrow;                         // swizzle: halo row j keeps granule g at slot g ^ (j & 7)
    auto issue_tile = [&](int tile, int buf) __attribute__((always_inline)) {
        const int n = tile / tiles_per_patch;
        const int rem = tile - n * tiles_per_patch;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        char* lds = smem + buf * kD64BufBytes;
#pragma unroll
        for (int j = 0; j < kD64Instr / 4; ++j) {
            const int ii = wave + 4 * j;
            const int hr = ii * 8 + lrow;                       // halo row index = hy * 18 + hx
            const int hy = hr / kD64HaloW, hx = hr - hy * kD64HaloW;
            const int Y = ty * 8 - 1 + hy, X = tx * 16 - 1 + hx;
            const bool ok = ((unsigned)Y < (unsigned)p.H) & ((unsigned)X < (unsigned)p.W) & (hr < kD64Rows);
            uint32_t off = (uint32_t)((n * p.H + Y) * p.W + X) * 128u + (uint32_t)(gsrc * 16 + kZeroHeaderBytes);
            off = ok ? off : 0u;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(p.src + off), (LDS_AS void*)(lds + ii * 1024), 16, 0, 0);
        }
    };

    // halo row of output pixel (r, x) = (wp*4 + ni, frow) for tap (0,0); + ky*18 + kx per tap
    int hbase[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) hbase[ni] = (wp * 4 + ni) * kD64HaloW + frow;

    issue_tile(tile_at(0), 0);
    bool prev_full = false;
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = tile_at(it);
        // the halo copies of tile `it` are older than the previous tile's stores (4 per wave when that tile
        // was full): leave those in flight
        if (prev_full) wait_vmcnt<4>();
        else wait_vmcnt<0>();
        __syncthreads();
        if (it + 1 < my_tiles) issue_tile(tile_at(it + 1), (it + 1) & 1);

        const char* lds = smem + (it & 1) * kD64BufBytes;
        f32x4_t acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[m][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // an opaque zero per tile: without it the 72 tile-invariant fragment addresses are hoisted out of the tile
        // loop into registers the weights need (spills)
        int zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = (t / 3) * kD64HaloW + (t % 3) + zero;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int hr = hbase[ni] + toff;
                    const bf16x8_t b = *(const bf16x8_t*)(lds + hr * 128 + (((kk * 4 + fg) ^ (hr & 7)) << 4));
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][ni] = mfma16<F16>(wf[t][kk][m], b, acc[m][ni]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                  // keep the scheduler from hoisting later taps' reads (spills)
        }

        // ---- epilogue: lane holds channels wc*32 + fg*8 .. +7 of pixel (wp*4 + ni, frow)
        const int n = tile / tiles_per_patch;
        const int rem = tile - n * tiles_per_patch;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        prev_full = (ty * 8 + 8 <= p.H) && (tx * 16 + 16 <= p.W);
        const int c0 = wc * 32 + fg * 8;
        float sc[8], sh[8];
        *(float4*)&sc[0] = *(const float4*)(cst + c0);
        *(float4*)&sc[4] = *(const float4*)(cst + c0 + 4);
        *(float4*)&sh[0] = *(const float4*)(cst + 64 + c0);
        *(float4*)&sh[4] = *(const float4*)(cst + 64 + c0 + 4);
        const int ox = tx * 16 + frow;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int oy = ty * 8 + wp * 4 + ni;
            float y[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                y[q] = acc[0][ni][q] * sc[q] + sh[q];
                y[4 + q] = acc[1][ni][q] * sc[4 + q] + sh[4 + q];
            }
            if (p.relu) {
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = fmaxf(y[q], 0.f);
            }
            uint4 r;
            r.x = pack2<F16>(y[0], y[1]); r.y = pack2<F16>(y[2], y[3]);
            r.z = pack2<F16>(y[4], y[5]); r.w = pack2<F16>(y[6], y[7]);
            if (oy < p.H && ox < p.W)
                *(uint4*)((uint16_t*)p.out + (((size_t)n * p.H + oy) * p.W + ox) * 64 + c0) = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv3x3_c64_direct_x3 -- the same direct conv in the split mode (kF16X3): pixels are [32 hi][32 lo][32 hi][32 lo] (256 B, 16 granules,
// slot = (granule + 2 * halo row) & 15: conflict-free for ds_read_b128's 16-lane groups), the wave's weights are hi + lo fragments (288 VGPRs: one block per CU), three MFMAs per
// product, outputs split again.  The generic split kernel needs 0.82 ms per 140 patches for each of these layers.
// ------------------------------------------------------------------------------------------------
constexpr int kD64x3Instr = 48;                                 // wave-instructions of 4 pixels (192 >= 180)
constexpr int kD64x3BufBytes = kD64x3Instr * 1024;
constexpr int kD64x3LdsBytes = 2 * kD64x3BufBytes + 512;

__global__ __launch_bounds__(256, 1) void conv3x3_c64_direct_x3(const Direct64Params p)
{
    constexpr bool F16 = true;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wc = wave & 1;
    const int frow = lane & 15, fg = lane >> 4;
    const int tiles_x = (p.W + 15) / 16, tiles_y = (p.H + 7) / 8;
    const int tiles_per_patch = tiles_x * tiles_y;
    const int n_tiles = p.n * tiles_per_patch;
    // XCD-contiguous walk (grid = a multiple of 8 blocks), as in the tail kernels
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, GX = gridDim.x >> 3;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int xcd_lo = xcd * per_xcd, xcd_hi = min(n_tiles, xcd_lo + per_xcd);
    const int my_tiles = xcd_lo + slot < xcd_hi ? (xcd_hi - xcd_lo - slot + GX - 1) / GX : 0;
    if (my_tiles <= 0) return;
    auto tile_at = [&](int it) __attribute__((always_inline)) -> int { return xcd_lo + slot + it * GX; };
    float* cst = (float*)(smem + 2 * kD64x3BufBytes);
    if (tid < 64) { cst[tid] = p.scale[tid] * p.wmul; cst[64 + tid] = p.shift[tid]; }

    bf16x8_t whi[9][2][2], wlo[9][2][2];                        // [tap][kk][mi of this wave]; wfrag = [hi | lo][9][2][4 mi][64 lanes]
    {
        const uint4* src = (const uint4*)p.wfrag + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const size_t f = (size_t)((t * 2 + kk) * 4 + wc * 2 + m) * 64;
                    whi[t][kk][m] = __builtin_bit_cast(bf16x8_t, src[f]);
                    wlo[t][kk][m] = __builtin_bit_cast(bf16x8_t, src[(size_t)9 * 2 * 4 * 64 + f]);
                }
    }

    auto issue_tile = [&](int tile, int buf) __attribute__((always_inline)) {
        const int n = tile / tiles_per_patch;
        const int rem = tile - n * tiles_per_patch;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        char* lds = smem + buf * kD64x3BufBytes;
#pragma unroll
        for (int j = 0; j < kD64x3Instr / 4; ++j) {
            const int ii = wave + 4 * j;
            const int hr = ii * 4 + (lane >> 4);                // halo row index = hy * 18 + hx
            const int hy = hr / kD64HaloW, hx = hr - hy * kD64HaloW;
            const int g = ((lane & 15) - 2 * hr) & 15;          // slot s of halo pixel hr holds granule (s - 2 hr) & 15 (see dec_tail_fused_x3)
            const int Y = ty * 8 - 1 + hy, X = tx * 16 - 1 + hx;
            const bool ok = ((unsigned)Y < (unsigned)p.H) & ((unsigned)X < (unsigned)p.W) & (hr < kD64Rows);
            uint32_t off = (uint32_t)((n * p.H + Y) * p.W + X) * 256u + (uint32_t)(g * 16 + kZeroHeaderBytes);
            off = ok ? off : 0u;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(p.src + off), (LDS_AS void*)(lds + ii * 1024), 16, 0, 0);
        }
    };

    // tile-invariant read offsets (as dec_tail_fused_x3): halo row hr = wp * 72 + frow + rc with rc = (ni + t / 3) * 18 + t % 3 known at
    // compile time, slot of granule G = (G + 2 hr) & 15 = (s0 + D) & 15, s0 = (fg + 2 frow) & 15, D = kk * 4 + 8 * lo + (2 rc & 15)
    const int s0 = (fg + 2 * frow) & 15;
    int rd_t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) rd_t[e] = (wp * 4 * kD64HaloW + frow) * 256 + (((s0 + 2 * e) & 15) << 4);

    issue_tile(tile_at(0), 0);
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = tile_at(it);
        wait_vmcnt<0>();
        __syncthreads();
        if (it + 1 < my_tiles) issue_tile(tile_at(it + 1), (it + 1) & 1);

        const char* lds = smem + (it & 1) * kD64x3BufBytes;
        const char* rb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) rb[e] = lds + rd_t[e];
        f32x4_t acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[m][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int rc = (ni + t / 3) * kD64HaloW + t % 3;
                    const int dh = (kk * 8 + 2 * rc) & 15, dl = (kk * 8 + 4 + 2 * rc) & 15;      // stored pixel: [32 hi][32 lo][32 hi][32 lo]
                    const bf16x8_t bh = *(const bf16x8_t*)(rb[dh >> 1] + rc * 256);
                    const bf16x8_t bl = *(const bf16x8_t*)(rb[dl >> 1] + rc * 256);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        acc[m][ni] = mfma16<F16>(wlo[t][kk][m], bh, acc[m][ni]);
                        acc[m][ni] = mfma16<F16>(whi[t][kk][m], bl, acc[m][ni]);
                        acc[m][ni] = mfma16<F16>(whi[t][kk][m], bh, acc[m][ni]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue: lane holds channels wc*32 + fg*8 .. +7 of pixel (wp*4 + ni, frow); hi at [c], lo at [64 + c]
        const int n = tile / tiles_per_patch;
        const int rem = tile - n * tiles_per_patch;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int c0 = wc * 32 + fg * 8;
        float sc[8], sh[8];
        *(float4*)&sc[0] = *(const float4*)(cst + c0);
        *(float4*)&sc[4] = *(const float4*)(cst + c0 + 4);
        *(float4*)&sh[0] = *(const float4*)(cst + 64 + c0);
        *(float4*)&sh[4] = *(const float4*)(cst + 64 + c0 + 4);
        const int ox = tx * 16 + frow;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int oy = ty * 8 + wp * 4 + ni;
            float y[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                y[q] = __builtin_fmaf(acc[0][ni][q], sc[q], sh[q]);
                y[4 + q] = __builtin_fmaf(acc[1][ni][q], sc[4 + q], sh[4 + q]);
            }
            if (p.relu) {
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = fmaxf(y[q], 0.f);
            }
            if (oy < p.H && ox < p.W)
                store_split8((uint16_t*)p.out + (((size_t)n * p.H + oy) * p.W + ox) * 128 + split_hi_elem(64, c0), 32, y);
        }
    }
}

hipError_t launch_direct64(const Direct64Params& p, int precision, int num_cus, hipStream_t s)
{
    const int n_tiles = p.n * ((p.H + 7) / 8) * ((p.W + 15) / 16);
    const int grid = ((n_tiles < 2 * num_cus ? n_tiles : 2 * num_cus) + 7) & ~7;      // (XCD-contiguous walk: a multiple of 8)
    if (precision == kF16X3) {
        static bool attr_done[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (!attr_done[dev & 63]) {
            e = hipFuncSetAttribute((const void*)conv3x3_c64_direct_x3, hipFuncAttributeMaxDynamicSharedMemorySize, kD64x3LdsBytes);
            if (e != hipSuccess) return e;
            attr_done[dev & 63] = true;
        }
        hipLaunchKernelGGL(conv3x3_c64_direct_x3, dim3(((n_tiles < num_cus ? n_tiles : num_cus) + 7) & ~7), dim3(256), kD64x3LdsBytes, s, p);
        return hipGetLastError();
    }
    if (precision == kF16) hipLaunchKernelGGL(conv3x3_c64_direct<true>, dim3(grid), dim3(256), kD64LdsBytes, s, p);
    else hipLaunchKernelGGL(conv3x3_c64_direct<false>, dim3(grid), dim3(256), kD64LdsBytes, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// bottleneck_fused -- one whole ResNet stage-2 bottleneck block per launch (three of them at 111x111 in the sbb nets):
//
//   a = ReLU(BN(conv1x1(x, CIN -> 64)))              phase A, on the 10 x 18 halo of an 8 x 16 output tile
//   b = ReLU(BN(conv3x3(a, 64 -> 64)))               phase B, from the halo tile in LDS (as conv3x3_c64_direct)
//   y = ReLU(BN(conv1x1(b, 64 -> 256)) + x)          phase C, identity block (CIN = 256)
//   y = ReLU(BN(conv1x1([b, x], 128 -> 256)))        phase C, projection block (CIN = 64; the planner has folded the
//                                                    shortcut conv into one contraction over [b, x])
//
// Run as three launches these layers are HBM-bound and move the 256-channel tensor four times per block (read for
// the first 1x1, residual read + write in the last) plus the 64-channel tensors four times; fused, x is read once
// (halo overlap from L2) and y written once.  What a CU has to do per tile is small next to that traffic (~300 MFMAs
// per wave), so the kernel is built for memory-level parallelism, not for MFMA rate: ONE block of four waves per CU
// (up to 512 VGPRs per lane), the x fragments of the NEXT tile are requested into registers before the current tile's
// phases run, and the fragments of the inner pixels double as the residual of phase C (the MFMA B-operand layout --
// pixel = lane & 15, 8 channels per lane -- is exactly the layout of the epilogue's 16-byte channel groups).
//   * halo pixel order: n-tiles 0-7 = the 8 inner rows (16 pixels each), n-tiles 8-11 = the 52 border pixels (+12
//     dummies).  Wave w owns inner rows 2w, 2w+1 and border tile 8+w in phases A and C, and the 16 output channels
//     of MFMA row block w in phase B (its 9 x 2 weight fragments live in 72 VGPRs).
//   * W1 and W3 stay in LDS for the whole kernel (A-fragment order, linear 16-byte reads); a (halo, 192 rows x 128 B)
//     and b (128 rows x 128 B) are XOR-swizzled rows; halo pixels outside the image are ZERO (the 3x3 conv pads a, not x).
//   * tiles are walked so that every XCD owns one contiguous range of them: vertical halo neighbours share an L2.
// ------------------------------------------------------------------------------------------------
constexpr int kBlkHaloW = 18;
constexpr int kBlkABytes = 192 * 128;                           // a: 180 halo rows (+12 dummy rows)
constexpr int kBlkBBytes = 128 * 128;                           // b: 8 x 16 pixels
constexpr int kBlkCstBytes = (4 * 64 + 2 * 256) * 4;            // s1, b1, s2, b2 [64]; s3, b3 [256]
constexpr int block_lds_bytes(int cin, bool proj) { return (cin / 32) * 4 * 1024 + (proj ? 4 : 2) * 16 * 1024 + kBlkABytes + kBlkBBytes + kBlkCstBytes; }

template <bool F16, int CIN, bool PROJ>
__global__ __launch_bounds__(256, 1) void bottleneck_fused(const BlockParams p)
{
    static_assert((CIN == 256 && !PROJ) || (CIN == 64 && PROJ), "identity blocks read 256 channels, the projection block 64");
    constexpr int KA = CIN / 32;                                // K-steps (32 channels) of phase A
    constexpr int KC = PROJ ? 4 : 2;                            // K-steps of phase C
    constexpr int PIXB = CIN * 2;                               // bytes per stored x pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_w1 = smem;
    char* lds_w3 = lds_w1 + KA * 4 * 1024;
    char* lds_a = lds_w3 + KC * 16 * 1024;
    char* lds_b = lds_a + kBlkABytes;
    float* cst = (float*)(lds_b + kBlkBBytes);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int tiles_x = (p.W + 15) / 16, tiles_y = (p.H + 7) / 8;
    const int tiles_per_patch = tiles_x * tiles_y;
    const int n_tiles = p.n * tiles_per_patch;
    // XCD-contiguous walk: XCD x = block % 8 owns tiles [x * per_xcd, (x + 1) * per_xcd)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, GX = gridDim.x >> 3;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int xcd_lo = xcd * per_xcd, xcd_hi = min(n_tiles, xcd_lo + per_xcd);
    const int my_tiles = xcd_lo + slot < xcd_hi ? (xcd_hi - xcd_lo - slot + GX - 1) / GX : 0;
    if (my_tiles <= 0) return;

    // ---- one-time: weights and constants to LDS, this wave's 3x3 fragments to registers
    for (int i = tid; i < KA * 4 * 64; i += 256) ((uint4*)lds_w1)[i] = ((const uint4*)p.w1)[i];
    for (int i = tid; i < KC * 16 * 64; i += 256) ((uint4*)lds_w3)[i] = ((const uint4*)p.w3)[i];
    if (tid < 64) {
        cst[tid] = p.s1[tid]; cst[64 + tid] = p.b1[tid]; cst[128 + tid] = p.s2[tid]; cst[192 + tid] = p.b2[tid];
    }
    cst[256 + tid] = p.s3[tid]; cst[512 + tid] = p.b3[tid];
    bf16x8_t wf[9][2];                                          // [tap][kk], MFMA row block `wave`
    {
        const uint4* src = (const uint4*)p.w2 + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) wf[t][kk] = __builtin_bit_cast(bf16x8_t, src[(size_t)((t * 2 + kk) * 4 + wave) * 64]);
    }

    // ---- this lane's three halo pixels (fixed per kernel): (hy, hx) and the LDS row hr = hy * 18 + hx
    int hy[3], hx[3], hr[3];
#pragma unroll
    for (int j = 0; j < 2; ++j) { hy[j] = 2 * wave + j + 1; hx[j] = frow + 1; hr[j] = hy[j] * kBlkHaloW + hx[j]; }
    {
        const int bi = wave * 16 + frow;                       // border pixel index
        int y, x;
        if (bi < 18) { y = 0; x = bi; }
        else if (bi < 36) { y = 9; x = bi - 18; }
        else if (bi < 44) { y = 1 + (bi - 36); x = 0; }
        else if (bi < 52) { y = 1 + (bi - 44); x = 17; }
        else { y = 10; x = bi - 52; }                           // dummies: rows 180..191, never inside the image
        hy[2] = y; hx[2] = x; hr[2] = y * kBlkHaloW + x;
    }

    bool inimg[3];
    uint32_t xoff[3];
    auto locate = [&](int tile, bool (&in)[3], uint32_t (&off)[3]) __attribute__((always_inline)) {
        const int n = tile / tiles_per_patch;
        const int rem = tile - n * tiles_per_patch;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int Y = ty * 8 - 1 + hy[j], X = tx * 16 - 1 + hx[j];
            in[j] = ((unsigned)Y < (unsigned)p.H) & ((unsigned)X < (unsigned)p.W) & (hy[j] < 10);
            // (pixels outside the image read the start of the buffer: finite or not, their column is replaced by zeros)
            off[j] = in[j] ? (uint32_t)((n * p.H + Y) * p.W + X) * (uint32_t)PIXB + (uint32_t)(kZeroHeaderBytes + fg * 16) : 0u;
        }
    };
    bf16x8_t xcur[3][KA], xnext[3][KA];
    auto fetch = [&](const uint32_t (&off)[3], bf16x8_t (&dst)[3][KA]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int kk = 0; kk < KA; ++kk) dst[j][kk] = *(const bf16x8_t*)(p.x + off[j] + kk * 64);
    };

    auto tile_at = [&](int it) __attribute__((always_inline)) -> int { return xcd_lo + slot + it * GX; };
    locate(tile_at(0), inimg, xoff);
    fetch(xoff, xcur);
    __syncthreads();                                            // weights / constants visible

    for (int it = 0; it < my_tiles; ++it) {
        const int tile = tile_at(it);
        bool in_next[3];
        uint32_t off_next[3];
        if (it + 1 < my_tiles) {                                // the next tile's x: in flight across all three phases
            locate(tile_at(it + 1), in_next, off_next);
            fetch(off_next, xnext);
        }

        // ---- phase A: a[halo pixel][64] = ReLU(s1 * (W1 . x) + b1), zero outside the image
        {
            f32x4_t acc[4][3];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[mi][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KA; ++kk) {
                bf16x8_t wa[4];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) wa[mi] = *(const bf16x8_t*)(lds_w1 + (kk * 4 + mi) * 1024 + lane * 16);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[mi][j] = mfma16<F16>(wa[mi], xcur[j][kk], acc[mi][j]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int c0 = s2 * 32 + fg * 8;
                float sc[8], sh[8];
                *(float4*)&sc[0] = *(const float4*)(cst + c0); *(float4*)&sc[4] = *(const float4*)(cst + c0 + 4);
                *(float4*)&sh[0] = *(const float4*)(cst + 64 + c0); *(float4*)&sh[4] = *(const float4*)(cst + 64 + c0 + 4);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float y[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        y[q] = fmaxf(acc[2 * s2][j][q] * sc[q] + sh[q], 0.f);
                        y[4 + q] = fmaxf(acc[2 * s2 + 1][j][q] * sc[4 + q] + sh[4 + q], 0.f);
                    }
                    uint4 r;
                    r.x = pack2<F16>(y[0], y[1]); r.y = pack2<F16>(y[2], y[3]); r.z = pack2<F16>(y[4], y[5]); r.w = pack2<F16>(y[6], y[7]);
                    if (!inimg[j]) r = make_uint4(0u, 0u, 0u, 0u);
                    *(uint4*)(lds_a + hr[j] * 128 + (((s2 * 4 + fg) ^ (hr[j] & 7)) << 4)) = r;
                }
            }
        }
        __syncthreads();

        // ---- phase B: b[pixel][16 channels of row block `wave`] = ReLU(s2 * conv3x3(a) + b2)
        {
            const int sB = wave >> 1, half = wave & 1;
            const int cB = sB * 32 + fg * 8 + half * 4;
            const float4 sc = *(const float4*)(cst + 128 + cB), sh = *(const float4*)(cst + 192 + cB);
#pragma unroll
            for (int g4 = 0; g4 < 2; ++g4) {                   // four output rows at a time: independent accumulator chains
                f32x4_t acc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = (g4 * 4 + i + t / 3) * kBlkHaloW + frow + t % 3;
                            const bf16x8_t bq = *(const bf16x8_t*)(lds_a + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
                            acc[i] = mfma16<F16>(wf[t][kk], bq, acc[i]);
                        }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = (g4 * 4 + i) * 16 + frow;
                    uint2 r;
                    r.x = pack2<F16>(fmaxf(acc[i][0] * sc.x + sh.x, 0.f), fmaxf(acc[i][1] * sc.y + sh.y, 0.f));
                    r.y = pack2<F16>(fmaxf(acc[i][2] * sc.z + sh.z, 0.f), fmaxf(acc[i][3] * sc.w + sh.w, 0.f));
                    *(uint2*)(lds_b + row * 128 + (((sB * 4 + fg) ^ (row & 7)) << 4) + half * 8) = r;
                }
            }
        }
        __syncthreads();

        // ---- phase C: y[inner rows 2w, 2w+1][256] = ReLU(s3 * (W3 . [b, x?]) + b3 (+ x)), 32 channels at a time
        {
            const int n = tile / tiles_per_patch;
            const int rem = tile - n * tiles_per_patch;
            const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
            const int ox = tx * 16 + frow;
            bf16x8_t bf[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int row = (2 * wave + j) * 16 + frow;
                    bf[j][kk] = *(const bf16x8_t*)(lds_b + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
                }
#pragma unroll
            for (int s3 = 0; s3 < 8; ++s3) {
                f32x4_t acc[2][2];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[m][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < KC; ++kk)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const bf16x8_t wa = *(const bf16x8_t*)(lds_w3 + (kk * 16 + 2 * s3 + m) * 1024 + lane * 16);
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            bf16x8_t bq;
                            if constexpr (PROJ) bq = kk < 2 ? bf[j][kk & 1] : xcur[j][kk & 1];
                            else bq = bf[j][kk & 1];
                            acc[m][j] = mfma16<F16>(wa, bq, acc[m][j]);
                        }
                    }
                const int c0 = s3 * 32 + fg * 8;
                float sc[8], sh[8];
                *(float4*)&sc[0] = *(const float4*)(cst + 256 + c0); *(float4*)&sc[4] = *(const float4*)(cst + 256 + c0 + 4);
                *(float4*)&sh[0] = *(const float4*)(cst + 512 + c0); *(float4*)&sh[4] = *(const float4*)(cst + 512 + c0 + 4);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float y[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        y[q] = acc[0][j][q] * sc[q] + sh[q];
                        y[4 + q] = acc[1][j][q] * sc[4 + q] + sh[4 + q];
                    }
                    if constexpr (!PROJ) {                      // the residual: this lane's x fragment of K-step s3 = channels c0 .. c0+7
                        const uint4 rv = __builtin_bit_cast(uint4, xcur[j][s3 % KA]);
                        y[0] += unpack_lo<F16>(rv.x); y[1] += unpack_hi<F16>(rv.x); y[2] += unpack_lo<F16>(rv.y); y[3] += unpack_hi<F16>(rv.y);
                        y[4] += unpack_lo<F16>(rv.z); y[5] += unpack_hi<F16>(rv.z); y[6] += unpack_lo<F16>(rv.w); y[7] += unpack_hi<F16>(rv.w);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) y[q] = fmaxf(y[q], 0.f);
                    uint4 r;
                    r.x = pack2<F16>(y[0], y[1]); r.y = pack2<F16>(y[2], y[3]); r.z = pack2<F16>(y[4], y[5]); r.w = pack2<F16>(y[6], y[7]);
                    const int oy = ty * 8 + 2 * wave + j;
                    if (oy < p.H && ox < p.W)
                        *(uint4*)((uint16_t*)p.out + (((size_t)n * p.H + oy) * p.W + ox) * 256 + c0) = r;
                }
            }
        }

        if (it + 1 < my_tiles) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                inimg[j] = in_next[j];
#pragma unroll
                for (int kk = 0; kk < KA; ++kk) xcur[j][kk] = xnext[j][kk];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bottleneck_fused_pq -- the same block with its phases split over two groups of four waves (one of each per SIMD):
//   P ("producer") waves run phases A and B of tile i:  x halo -> a (LDS) -> b (LDS, double buffered)
//   Q ("consumer") waves run phase C of tile i-1:        b, x -> y
// In the one-group kernel a wave's MFMA, VALU (epilogues) and LDS work serialise (one wave per SIMD); here the MFMA / LDS-
// heavy phases of one tile overlap the VALU / store-heavy phase of the previous one.  Two block-wide barriers per iteration:
// after phase A (a visible to the P waves; Q has done the first half of its channel groups) and at the end (b[i & 1]
// complete, b[(i-1) & 1] and a free).  P requests the next tile's x right after phase A into the registers that phase just
// freed; Q reads its inner-pixel x (the residual; the projection block's second operand) itself, one tile ahead.
// ------------------------------------------------------------------------------------------------
constexpr int block_pq_lds_bytes(int cin, bool proj) { return block_lds_bytes(cin, proj) + kBlkBBytes; }

template <bool F16, int CIN, bool PROJ>
__global__ __launch_bounds__(512, 1) void bottleneck_fused_pq(const BlockParams p)
{
    static_assert((CIN == 256 && !PROJ) || (CIN == 64 && PROJ), "identity blocks read 256 channels, the projection block 64");
    constexpr int KA = CIN / 32;
    constexpr int KC = PROJ ? 4 : 2;
    constexpr int PIXB = CIN * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_w1 = smem;
    char* lds_w3 = lds_w1 + KA * 4 * 1024;
    char* lds_a = lds_w3 + KC * 16 * 1024;
    char* lds_b = lds_a + kBlkABytes;                           // two buffers of kBlkBBytes
    float* cst = (float*)(lds_b + 2 * kBlkBBytes);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_q = wave8 >= 4;
    const int wave = wave8 & 3;                                 // index inside the group
    const int frow = lane & 15, fg = lane >> 4;
    const int tiles_x = (p.W + 15) / 16, tiles_y = (p.H + 7) / 8;
    const int tiles_per_patch = tiles_x * tiles_y;
    const int n_tiles = p.n * tiles_per_patch;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, GX = gridDim.x >> 3;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int xcd_lo = xcd * per_xcd, xcd_hi = min(n_tiles, xcd_lo + per_xcd);
    const int my_tiles = xcd_lo + slot < xcd_hi ? (xcd_hi - xcd_lo - slot + GX - 1) / GX : 0;
    if (my_tiles <= 0) return;
    auto tile_at = [&](int it) __attribute__((always_inline)) -> int { return xcd_lo + slot + it * GX; };

    for (int i = tid; i < KA * 4 * 64; i += 512) ((uint4*)lds_w1)[i] = ((const uint4*)p.w1)[i];
    for (int i = tid; i < KC * 16 * 64; i += 512) ((uint4*)lds_w3)[i] = ((const uint4*)p.w3)[i];
    if (tid < 64) {
        cst[tid] = p.s1[tid]; cst[64 + tid] = p.b1[tid]; cst[128 + tid] = p.s2[tid]; cst[192 + tid] = p.b2[tid];
    }
    if (tid < 256) { cst[256 + tid] = p.s3[tid]; cst[512 + tid] = p.b3[tid]; }

    if (!is_q) {
        // =============================================== P: phases A and B ===============================================
        bf16x8_t wf[9][2];
        {
            const uint4* src = (const uint4*)p.w2 + lane;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) wf[t][kk] = __builtin_bit_cast(bf16x8_t, src[(size_t)((t * 2 + kk) * 4 + wave) * 64]);
        }
        int hy[3], hx[3], hr[3];
#pragma unroll
        for (int j = 0; j < 2; ++j) { hy[j] = 2 * wave + j + 1; hx[j] = frow + 1; hr[j] = hy[j] * kBlkHaloW + hx[j]; }
        {
            const int bi = wave * 16 + frow;
            int y, x;
            if (bi < 18) { y = 0; x = bi; }
            else if (bi < 36) { y = 9; x = bi - 18; }
            else if (bi < 44) { y = 1 + (bi - 36); x = 0; }
            else if (bi < 52) { y = 1 + (bi - 44); x = 17; }
            else { y = 10; x = bi - 52; }
            hy[2] = y; hx[2] = x; hr[2] = y * kBlkHaloW + x;
        }
        bool inimg[3];
        uint32_t xoff[3];
        auto locate = [&](int tile) __attribute__((always_inline)) {
            const int n = tile / tiles_per_patch;
            const int rem = tile - n * tiles_per_patch;
            const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int Y = ty * 8 - 1 + hy[j], X = tx * 16 - 1 + hx[j];
                inimg[j] = ((unsigned)Y < (unsigned)p.H) & ((unsigned)X < (unsigned)p.W) & (hy[j] < 10);
                xoff[j] = inimg[j] ? (uint32_t)((n * p.H + Y) * p.W + X) * (uint32_t)PIXB + (uint32_t)(kZeroHeaderBytes + fg * 16) : 0u;
            }
        };
        bf16x8_t xf[3][KA];
        auto fetch = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int kk = 0; kk < KA; ++kk) xf[j][kk] = *(const bf16x8_t*)(p.x + xoff[j] + kk * 64);
        };
        locate(tile_at(0));
        fetch();
        __syncthreads();                                        // weights / constants visible (all eight waves)

        for (int it = 0; it <= my_tiles; ++it) {
            bool in_cur[3] = {inimg[0], inimg[1], inimg[2]};
            if (it < my_tiles) {
                // ---- phase A of tile `it`, 32 output channels (two MFMA row blocks) at a time: 24 accumulator registers
                // instead of 48 -- with 96 registers of x fragments and 72 of 3x3 weights this wave has no more to give
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    f32x4_t acc[2][3];
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int j = 0; j < 3; ++j) acc[m][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < KA; ++kk) {
                        bf16x8_t wa[2];
#pragma unroll
                        for (int m = 0; m < 2; ++m) wa[m] = *(const bf16x8_t*)(lds_w1 + (kk * 4 + 2 * s2 + m) * 1024 + lane * 16);
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int j = 0; j < 3; ++j) acc[m][j] = mfma16<F16>(wa[m], xf[j][kk], acc[m][j]);
                    }
                    const int c0 = s2 * 32 + fg * 8;
                    float sc[8], sh[8];
                    *(float4*)&sc[0] = *(const float4*)(cst + c0); *(float4*)&sc[4] = *(const float4*)(cst + c0 + 4);
                    *(float4*)&sh[0] = *(const float4*)(cst + 64 + c0); *(float4*)&sh[4] = *(const float4*)(cst + 64 + c0 + 4);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        float y[8];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            y[q] = fmaxf(acc[0][j][q] * sc[q] + sh[q], 0.f);
                            y[4 + q] = fmaxf(acc[1][j][q] * sc[4 + q] + sh[4 + q], 0.f);
                        }
                        uint4 r;
                        r.x = pack2<F16>(y[0], y[1]); r.y = pack2<F16>(y[2], y[3]); r.z = pack2<F16>(y[4], y[5]); r.w = pack2<F16>(y[6], y[7]);
                        if (!in_cur[j]) r = make_uint4(0u, 0u, 0u, 0u);
                        *(uint4*)(lds_a + hr[j] * 128 + (((s2 * 4 + fg) ^ (hr[j] & 7)) << 4)) = r;
                    }
                }
                if (it + 1 < my_tiles) {                        // x of the next tile into the registers phase A just freed
                    locate(tile_at(it + 1));
                    fetch();
                }
            }
            __syncthreads();                                    // B1: a visible
            if (it < my_tiles) {
                // ---- phase B of tile `it` -> b[it & 1]
                char* bb = lds_b + (it & 1) * kBlkBBytes;
                const int sB = wave >> 1, half = wave & 1;
                const int cB = sB * 32 + fg * 8 + half * 4;
                const float4 sc = *(const float4*)(cst + 128 + cB), sh = *(const float4*)(cst + 192 + cB);
#pragma unroll
                for (int g4 = 0; g4 < 2; ++g4) {
                    f32x4_t acc[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int row = (g4 * 4 + i + t / 3) * kBlkHaloW + frow + t % 3;
                                const bf16x8_t bq = *(const bf16x8_t*)(lds_a + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
                                acc[i] = mfma16<F16>(wf[t][kk], bq, acc[i]);
                            }
                        __builtin_amdgcn_sched_barrier(0);      // (keeps later taps' reads from being hoisted: registers)
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = (g4 * 4 + i) * 16 + frow;
                        uint2 r;
                        r.x = pack2<F16>(fmaxf(acc[i][0] * sc.x + sh.x, 0.f), fmaxf(acc[i][1] * sc.y + sh.y, 0.f));
                        r.y = pack2<F16>(fmaxf(acc[i][2] * sc.z + sh.z, 0.f), fmaxf(acc[i][3] * sc.w + sh.w, 0.f));
                        *(uint2*)(bb + row * 128 + (((sB * 4 + fg) ^ (row & 7)) << 4) + half * 8) = r;
                    }
                }
            }
            __syncthreads();                                    // B2: b[it & 1] complete; a and b[(it-1) & 1] free
        }
    } else {
        // =============================================== Q: phase C ===============================================
        // inner-pixel x of this wave's two output rows (2w, 2w+1): residual (identity) / second operand (projection)
        bf16x8_t xq[2][KA], xq_next[2][KA];
        auto fetch_q = [&](int tile, bf16x8_t (&dst)[2][KA]) __attribute__((always_inline)) {
            const int n = tile / tiles_per_patch;
            const int rem = tile - n * tiles_per_patch;
            const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int Y = ty * 8 + 2 * wave + j, X = tx * 16 + frow;
                const bool in = (Y < p.H) & (X < p.W);
                const uint32_t off = in ? (uint32_t)((n * p.H + Y) * p.W + X) * (uint32_t)PIXB + (uint32_t)(kZeroHeaderBytes + fg * 16) : 0u;
#pragma unroll
                for (int kk = 0; kk < KA; ++kk) dst[j][kk] = *(const bf16x8_t*)(p.x + off + kk * 64);
            }
        };
        fetch_q(tile_at(0), xq_next);
        __syncthreads();                                        // weights / constants visible (all eight waves)

        for (int it = 0; it <= my_tiles; ++it) {
            const bool work = it >= 1;
            const int tile = work ? tile_at(it - 1) : 0;
            const int n = tile / tiles_per_patch;
            const int rem = tile - n * tiles_per_patch;
            const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
            const int ox = tx * 16 + frow;
            bf16x8_t bf[2][2];
            if (work) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int kk = 0; kk < KA; ++kk) xq[j][kk] = xq_next[j][kk];
                const char* bb = lds_b + ((it - 1) & 1) * kBlkBBytes;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        const int row = (2 * wave + j) * 16 + frow;
                        bf[j][kk] = *(const bf16x8_t*)(bb + row * 128 + (((kk * 4 + fg) ^ (row & 7)) << 4));
                    }
            }
            if (it < my_tiles) fetch_q(tile_at(it), xq_next);   // one tile ahead
            auto channel_groups = [&](int s_lo, int s_hi) __attribute__((always_inline)) {
#pragma unroll
                for (int s3 = s_lo; s3 < s_hi; ++s3) {
                    f32x4_t acc[2][2];
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[m][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < KC; ++kk)
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const bf16x8_t wa = *(const bf16x8_t*)(lds_w3 + (kk * 16 + 2 * s3 + m) * 1024 + lane * 16);
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                bf16x8_t bq;
                                if constexpr (PROJ) bq = kk < 2 ? bf[j][kk & 1] : xq[j][kk & 1];
                                else bq = bf[j][kk & 1];
                                acc[m][j] = mfma16<F16>(wa, bq, acc[m][j]);
                            }
                        }
                    const int c0 = s3 * 32 + fg * 8;
                    float sc[8], sh[8];
                    *(float4*)&sc[0] = *(const float4*)(cst + 256 + c0); *(float4*)&sc[4] = *(const float4*)(cst + 256 + c0 + 4);
                    *(float4*)&sh[0] = *(const float4*)(cst + 512 + c0); *(float4*)&sh[4] = *(const float4*)(cst + 512 + c0 + 4);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float y[8];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            y[q] = acc[0][j][q] * sc[q] + sh[q];
                            y[4 + q] = acc[1][j][q] * sc[4 + q] + sh[4 + q];
                        }
                        if constexpr (!PROJ) {
                            const uint4 rv = __builtin_bit_cast(uint4, xq[j][s3 % KA]);
                            y[0] += unpack_lo<F16>(rv.x); y[1] += unpack_hi<F16>(rv.x); y[2] += unpack_lo<F16>(rv.y); y[3] += unpack_hi<F16>(rv.y);
                            y[4] += unpack_lo<F16>(rv.z); y[5] += unpack_hi<F16>(rv.z); y[6] += unpack_lo<F16>(rv.w); y[7] += unpack_hi<F16>(rv.w);
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q) y[q] = fmaxf(y[q], 0.f);
                        uint4 r;
                        r.x = pack2<F16>(y[0], y[1]); r.y = pack2<F16>(y[2], y[3]); r.z = pack2<F16>(y[4], y[5]); r.w = pack2<F16>(y[6], y[7]);
                        const int oy = ty * 8 + 2 * wave + j;
                        if (oy < p.H && ox < p.W)
                            *(uint4*)((uint16_t*)p.out + (((size_t)n * p.H + oy) * p.W + ox) * 256 + c0) = r;
                    }
                    __builtin_amdgcn_sched_barrier(0);          // (keeps later groups' weight reads from being hoisted: registers)
                }
            };
            if (work) channel_groups(0, 4);
            __syncthreads();                                    // B1
            if (work) channel_groups(4, 8);
            __syncthreads();                                    // B2
        }
    }
}

hipError_t launch_bottleneck(const BlockParams& p, int precision, int num_cus, hipStream_t s)
{
    const int n_tiles = p.n * ((p.H + 7) / 8) * ((p.W + 15) / 16);
    int grid = n_tiles < num_cus ? n_tiles : num_cus;
    grid = (grid + 7) & ~7;                                     // the XCD-contiguous walk needs a multiple of 8 blocks
    auto go = [&](auto kernel, int lds) -> hipError_t {
        static bool attr_done[4][64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const int slot = (precision == kF16 ? 0 : 1) + (p.proj ? 2 : 0);
        if (!attr_done[slot][dev & 63]) {
            e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return e;
            attr_done[slot][dev & 63] = true;
        }
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, s, p);
        return hipGetLastError();
    };
    if (p.pq) {
        auto go8 = [&](auto kernel, int lds) -> hipError_t {
            static bool attr_done8[4][64] = {};
            int dev = 0;
            hipError_t e = hipGetDevice(&dev);
            if (e != hipSuccess) return e;
            const int slot = (precision == kF16 ? 0 : 1) + (p.proj ? 2 : 0);
            if (!attr_done8[slot][dev & 63]) {
                e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e != hipSuccess) return e;
                attr_done8[slot][dev & 63] = true;
            }
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), lds, s, p);
            return hipGetLastError();
        };
        if (p.proj) return precision == kF16 ? go8(bottleneck_fused_pq<true, 64, true>, block_pq_lds_bytes(64, true)) : go8(bottleneck_fused_pq<false, 64, true>, block_pq_lds_bytes(64, true));
        return precision == kF16 ? go8(bottleneck_fused_pq<true, 256, false>, block_pq_lds_bytes(256, false)) : go8(bottleneck_fused_pq<false, 256, false>, block_pq_lds_bytes(256, false));
    }
    if (p.proj) return precision == kF16 ? go(bottleneck_fused<true, 64, true>, block_lds_bytes(64, true)) : go(bottleneck_fused<false, 64, true>, block_lds_bytes(64, true));
    return precision == kF16 ? go(bottleneck_fused<true, 256, false>, block_lds_bytes(256, false)) : go(bottleneck_fused<false, 256, false>, block_lds_bytes(256, false));
}

// ------------------------------------------------------------------------------------------------
// element helpers for the HBM-bound kernels (E = uint16_t bf16 bits | float)
// ------------------------------------------------------------------------------------------------
template <typename E> __device__ inline E to_elem(float v);
template <> __device__ inline uint16_t to_elem<uint16_t>(float v) { return bf16_bits_rne(v); }
template <> __device__ inline float to_elem<float>(float v) { return v; }
template <> __device__ inline _Float16 to_elem<_Float16>(float v) { return (_Float16)fminf(fmaxf(v, -65504.f), 65504.f); }
template <typename E> __device__ inline float from_elem(E v);
template <> __device__ inline float from_elem<uint16_t>(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
template <> __device__ inline float from_elem<float>(float v) { return v; }
template <> __device__ inline float from_elem<_Float16>(_Float16 v) { return (float)v; }

template <typename E> struct alignas(16) Vec8 { E v[8]; };
template <typename E> struct alignas(sizeof(E) * 4) Vec4 { E v[4]; };

// split-mode writer of one network-input pixel into both input forms (see ingest_u8_kernel)
template <typename E>
__device__ inline void write_split_input(const float (&f)[3], void* c8, void* pairs, long idx, int t, int y, int x,
                                         int H, int pad, int pairs_w)
{
    _Float16 hi[3], lo[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) split_f32(f[i], hi[i], lo[i]);
    const _Float16 z = (_Float16)0.f;
    Vec8<_Float16> oh, ol;
#pragma unroll
    for (int i = 0; i < 8; ++i) { oh.v[i] = i < 3 ? hi[i] : z; ol.v[i] = i < 3 ? lo[i] : z; }
    // The channel slots 4..6 of the hi plane repeat lo(ch 0..2): every kernel that treats the form as an 8-channel tensor multiplies
    // them by zero weights (channels 3..7 do not exist), dec_tail_fused_x3ps reads the first granule as [h0 h1 h2 0 | l0 l1 l2 0]
#pragma unroll
    for (int i = 0; i < 3; ++i) oh.v[4 + i] = lo[i];
    ((Vec8<_Float16>*)c8)[2 * idx] = oh;
    ((Vec8<_Float16>*)c8)[2 * idx + 1] = ol;
    if (pairs) {
        const int PH = H + 2 * pad;
        const int xp = x + pad;
        _Float16* dst = (_Float16*)pairs + (((size_t)t * PH + (y + pad)) * pairs_w + (xp >> 1)) * 16 + (xp & 1) * 4;
        Vec4<_Float16> qh, ql;
        qh.v[0] = hi[0]; qh.v[1] = hi[1]; qh.v[2] = hi[2]; qh.v[3] = z;
        ql.v[0] = lo[0]; ql.v[1] = lo[1]; ql.v[2] = lo[2]; ql.v[3] = z;
        *(Vec4<_Float16>*)dst = qh;
        *(Vec4<_Float16>*)(dst + 8) = ql;
    }
}

// ------------------------------------------------------------------------------------------------
// ingest: u8 page -> normalised network input in both forms (main.py:239 `img / 255.0`, 285 slice)
// one thread per (tile, y, x)
// ------------------------------------------------------------------------------------------------
// SPLIT (kF16X3): every stored element is an fp16 (hi, lo) pair -- C8 pixel = [8 hi][8 lo] (32 bytes; hi slots 4..6 = lo 0..2), PAIRS
// granule = [2 px x 4 hi][2 px x 4 lo] (32 bytes); f32(v / 255.0) is carried to ~22 bits
template <typename E, bool SPLIT = false>
__global__ __launch_bounds__(256) void ingest_u8_kernel(const IngestParams p)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long per = (long)p.H * p.W;
    if (idx >= per * p.n_tiles) return;
    const int t = (int)(idx / per);
    const int rem = (int)(idx - t * per);
    const int y = rem / p.W, x = rem - y * p.W;
    int vy, vx;                                   // position on the (virtual) page
    if (p.whole) { vy = y; vx = x; }
    else if (p.tile_xy) { vx = p.tile_xy[2 * t] + x; vy = p.tile_xy[2 * t + 1] + y; }
    else {
        const int gt = p.grid_first + t;
        const int gi = gt / p.grid_nyf, gj = gt - gi * p.grid_nyf;
        vx = min(gi * p.grid_mid_x, p.Wp - p.W) + x;
        vy = min(gj * p.grid_mid_y, p.Hp - p.H) + y;
    }
    // optional nearest-neighbour rescale (cv2.INTER_NEAREST index tables): the rescaled page of
    // get_image_and_scales (main.py:196-214) / the resize of the whole-image branch (main.py:371)
    // is never materialised, the tiles are gathered straight from the stored page
    const int sy = p.map_y ? p.map_y[vy] : vy, sx = p.map_x ? p.map_x[vx] : vx;
    const uint8_t* px = p.page + ((size_t)sy * p.src_Wp + sx) * 3;
    E v0, v1, v2;
    if (p.bin_thr) {
        // otsu_copy + astype(uint8) + /255 (main.py:178-194, 443-444, 239): channel 0 binarised at the
        // page's Otsu threshold lands in all three channels (reference quirk, lines 191-193): 0.0 or 1.0
        v0 = v1 = v2 = to_elem<E>((int)px[0] > *p.bin_thr ? 1.f : 0.f);
    } else {
        v0 = to_elem<E>(p.lut[px[0]]); v1 = to_elem<E>(p.lut[px[1]]); v2 = to_elem<E>(p.lut[px[2]]);
    }
    const E z = to_elem<E>(0.f);
    if constexpr (SPLIT) {
        float f[3];
        if (p.bin_thr) f[0] = f[1] = f[2] = (int)px[0] > *p.bin_thr ? 1.f : 0.f;
        else { f[0] = p.lut[px[0]]; f[1] = p.lut[px[1]]; f[2] = p.lut[px[2]]; }
        write_split_input<E>(f, p.c8, p.pairs, idx, t, y, x, p.H, p.pad, p.pairs_w);
        return;
    }
    Vec8<E> o;
    o.v[0] = v0; o.v[1] = v1; o.v[2] = v2;
#pragma unroll
    for (int i = 3; i < 8; ++i) o.v[i] = z;
    ((Vec8<E>*)p.c8)[idx] = o;
    if (p.pairs) {
        const int PH = p.H + 2 * p.pad;
        const int xp = x + p.pad;
        E* dst = (E*)p.pairs + (((size_t)t * PH + (y + p.pad)) * p.pairs_w + (xp >> 1)) * 8 + (xp & 1) * 4;
        Vec4<E> q; q.v[0] = v0; q.v[1] = v1; q.v[2] = v2; q.v[3] = z;
        *(Vec4<E>*)dst = q;
    }
}

template <typename E, bool SPLIT = false>
__global__ __launch_bounds__(256) void ingest_f32_kernel(const float* x, int n, int H, int W, void* c8,
                                                         void* pairs, int pad, int pairs_w)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long per = (long)H * W;
    if (idx >= per * n) return;
    const int t = (int)(idx / per);
    const int rem = (int)(idx - t * per);
    const int y = rem / W, xx = rem - y * W;
    const float* px = x + idx * 3;
    if constexpr (SPLIT) {
        const float f[3] = {px[0], px[1], px[2]};
        write_split_input<E>(f, c8, pairs, idx, t, y, xx, H, pad, pairs_w);
        return;
    }
    const E v0 = to_elem<E>(px[0]), v1 = to_elem<E>(px[1]), v2 = to_elem<E>(px[2]);
    const E z = to_elem<E>(0.f);
    Vec8<E> o;
    o.v[0] = v0; o.v[1] = v1; o.v[2] = v2;
#pragma unroll
    for (int i = 3; i < 8; ++i) o.v[i] = z;
    ((Vec8<E>*)c8)[idx] = o;
    if (pairs) {
        const int PH = H + 2 * pad;
        const int xp = xx + pad;
        E* dst = (E*)pairs + (((size_t)t * PH + (y + pad)) * pairs_w + (xp >> 1)) * 8 + (xp & 1) * 4;
        Vec4<E> q; q.v[0] = v0; q.v[1] = v1; q.v[2] = v2; q.v[3] = z;
        *(Vec4<E>*)dst = q;
    }
}

// ------------------------------------------------------------------------------------------------
// Otsu threshold of channel 0 of the (virtually rescaled) page -- cv2.threshold(img[:,:,0], 0, 255,
// THRESH_BINARY + THRESH_OTSU) of otsu_copy (main.py:178-194).  Pass 1: 256-bin histogram, an HBM-bound
// scan (runs of equal bytes are counted in registers first: document pages are mostly one value, and
// same-address LDS atomics serialise).  Pass 2: one thread walks the 256 bins in the order and
// precision OpenCV's getThreshVal_Otsu_8u does [EXT], fp64, no FMA contraction.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hist_u8_kernel(const uint8_t* page, int src_Wp, int Hp, int Wp,
                                                      const int* map_y, const int* map_x, unsigned* hist)
{
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    constexpr int RUN = 16;
    const long total = (long)Hp * Wp;
    for (long base = ((long)blockIdx.x * 256 + threadIdx.x) * RUN; base < total; base += (long)gridDim.x * 256 * RUN) {
        int y = (int)(base / Wp), x = (int)(base - (long)y * Wp);
        const uint8_t* row = page + (size_t)(map_y ? map_y[y] : y) * src_Wp * 3;
        int prev = -1;
        unsigned cnt = 0;
        for (int i = 0; i < RUN && base + i < total; ++i) {
            const int v = row[(size_t)(map_x ? map_x[x] : x) * 3];
            if (v != prev) {
                if (cnt) atomicAdd(&h[prev], cnt);
                prev = v;
                cnt = 0;
            }
            ++cnt;
            if (++x == Wp) {
                x = 0;
                if (++y < Hp) row = page + (size_t)(map_y ? map_y[y] : y) * src_Wp * 3;
            }
        }
        if (cnt) atomicAdd(&h[prev], cnt);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

__global__ void otsu_threshold_kernel(const unsigned* hist, long n_pixels, int* thr)
{
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double scale = 1.0 / (double)n_pixels;
    double mu = 0.0;
    for (int i = 0; i < 256; ++i) mu += (double)i * (double)hist[i];
    mu *= scale;
    const double eps = (double)1.1920928955078125e-07f;      // FLT_EPSILON
    double mu1 = 0.0, q1 = 0.0, max_sigma = 0.0;
    int max_val = 0;
    for (int i = 0; i < 256; ++i) {
        const double p_i = (double)hist[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        const double q2 = 1.0 - q1;
        if (fmin(q1, q2) < eps || fmax(q1, q2) > 1.0 - eps) continue;
        mu1 = (mu1 + (double)i * p_i) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double d = mu1 - mu2;
        const double sigma = q1 * q2 * d * d;
        if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
    }
    *thr = max_val;
}

hipError_t launch_otsu(const uint8_t* page, int src_Wp, int Hp, int Wp, const int* map_y, const int* map_x,
                       unsigned* hist, int* thr, int num_cus, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(hist, 0, 256 * sizeof(unsigned), s);
    if (e != hipSuccess) return e;
    const long total = (long)Hp * Wp;
    long blocks = (total + 256 * 16 - 1) / (256 * 16);
    if (blocks > 8L * num_cus) blocks = 8L * num_cus;
    hipLaunchKernelGGL(hist_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, s, page, src_Wp, Hp, Wp, map_y, map_x, hist);
    hipLaunchKernelGGL(otsu_threshold_kernel, dim3(1), dim3(64), 0, s, hist, total, thr);
    return hipGetLastError();
}

hipError_t launch_ingest_u8(const IngestParams& p, int precision, hipStream_t s)
{
    const long total = (long)p.H * p.W * p.n_tiles;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (precision == kF32) hipLaunchKernelGGL(ingest_u8_kernel<float>, dim3(grid), dim3(256), 0, s, p);
    else if (precision == kF16X3) hipLaunchKernelGGL((ingest_u8_kernel<_Float16, true>), dim3(grid), dim3(256), 0, s, p);
    else if (precision == kF16) hipLaunchKernelGGL(ingest_u8_kernel<_Float16>, dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(ingest_u8_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_ingest_f32(const float* x, int n, int H, int W, void* c8, void* pairs, int pad,
                             int pairs_w, int precision, hipStream_t s)
{
    const long total = (long)H * W * n;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (precision == kF32)
        hipLaunchKernelGGL(ingest_f32_kernel<float>, dim3(grid), dim3(256), 0, s, x, n, H, W, c8, pairs, pad, pairs_w);
    else if (precision == kF16X3)
        hipLaunchKernelGGL((ingest_f32_kernel<_Float16, true>), dim3(grid), dim3(256), 0, s, x, n, H, W, c8, pairs, pad, pairs_w);
    else if (precision == kF16)
        hipLaunchKernelGGL(ingest_f32_kernel<_Float16>, dim3(grid), dim3(256), 0, s, x, n, H, W, c8, pairs, pad, pairs_w);
    else
        hipLaunchKernelGGL(ingest_f32_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, x, n, H, W, c8, pairs, pad, pairs_w);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// max-pool (valid), NHWC, one thread per (pixel, 8-channel granule)
// ------------------------------------------------------------------------------------------------
// optional per-channel affine + ReLU applied to every input element before the max: lets the stem
// write only its pre-BN tensor (the f1 skip) and the pool apply bn_conv1 + relu on the fly
// SPLIT (kF16X3): pixels are channel groups [G hi][G lo] (internal.h); values are re-assembled in fp32 (exact), the maximum is split again
template <typename E, bool SPLIT = false>
__global__ __launch_bounds__(256) void maxpool_kernel(const E* src, E* dst, int n, int H, int W, int C,
                                                      int k, int stride, int Ho, int Wo,
                                                      const float* pre_scale, const float* pre_shift, int pre_relu)
{
    // one thread = 8 channels x up to 4 horizontally adjacent outputs: the windows of neighbours overlap
    // (k - stride shared columns), so the strip is read once -- (3*stride + k) columns instead of 4*k
    constexpr int OX = 4;
    const int cg = C / 8;
    const int wq = (Wo + OX - 1) / OX;
    // Blocks are dealt round-robin to the 8 XCDs; give every XCD one CONTIGUOUS eighth of the output
    // raster, so the input rows shared by vertically adjacent windows (blocks a few indices apart)
    // meet in ONE L2 instead of being fetched by two (PMC: fetch was 1.43x the input tensor).
    const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const unsigned wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const unsigned idx = wg * 256u + threadIdx.x;
    const unsigned total = (unsigned)n * Ho * wq * cg;
    if (idx >= total) return;
    const int g = (int)(idx % cg);
    unsigned pix = idx / cg;
    const int oq = (int)(pix % wq); pix /= wq;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const int ox0 = oq * OX;
    const int nout = min(OX, Wo - ox0);
    float m[OX][8], ps[8], pb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ps[i] = pre_scale ? pre_scale[g * 8 + i] : 1.f;
        pb[i] = pre_scale ? pre_shift[g * 8 + i] : 0.f;
#pragma unroll
        for (int o = 0; o < OX; ++o) m[o][i] = -3.0e38f;
    }
    const int ncol = (nout - 1) * stride + k;                   // input columns of the strip
    const int CS = SPLIT ? 2 * C : C;                           // elements per stored pixel
    if (k == 3 && stride == 2 && nout == OX) {
        // the ResNet stem pool, full strip: all 27 loads are independent -> issue them back to back
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const E* row = src + (((size_t)b * H + oy * 2 + ky) * W + ox0 * 2) * CS + (SPLIT ? split_hi_elem(C, g * 8) : g * 8);
            Vec8<E> v[9], vl[SPLIT ? 9 : 1];
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                v[c] = *(const Vec8<E>*)(row + (size_t)c * CS);
                if constexpr (SPLIT) vl[c] = *(const Vec8<E>*)(row + (size_t)c * CS + split_group(C));
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                float x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float xv = from_elem<E>(v[c].v[i]);
                    if constexpr (SPLIT) xv = __fadd_rn(xv, from_elem<E>(vl[c].v[i]));       // hi + lo: exact in fp32
                    x[i] = __builtin_fmaf(xv, ps[i], pb[i]);                                  // (stem_pool_x3 states the same arithmetic)
                    if (pre_relu) x[i] = fmaxf(x[i], 0.f);
                }
#pragma unroll
                for (int o = 0; o < OX; ++o) {
                    if (c >= 2 * o && c < 2 * o + 3) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) m[o][i] = fmaxf(m[o][i], x[i]);
                    }
                }
            }
        }
    } else
    for (int ky = 0; ky < k; ++ky) {
        const E* row = src + (((size_t)b * H + oy * stride + ky) * W + ox0 * stride) * CS + (SPLIT ? split_hi_elem(C, g * 8) : g * 8);
        for (int c = 0; c < ncol; ++c) {
            const Vec8<E> v = *(const Vec8<E>*)(row + (size_t)c * CS);
            float x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float xv = from_elem<E>(v.v[i]);
                if constexpr (SPLIT) xv = __fadd_rn(xv, from_elem<E>((*(const Vec8<E>*)(row + (size_t)c * CS + split_group(C))).v[i]));
                x[i] = __builtin_fmaf(xv, ps[i], pb[i]);
                if (pre_relu) x[i] = fmaxf(x[i], 0.f);
            }
#pragma unroll
            for (int o = 0; o < OX; ++o) {
                const int kx = c - o * stride;                     // column c inside output o's window?
                if (o < nout && kx >= 0 && kx < k) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) m[o][i] = fmaxf(m[o][i], x[i]);
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < OX; ++o) {
        if (o < nout) {
            E* dp = dst + (((size_t)b * Ho + oy) * Wo + ox0 + o) * CS + (SPLIT ? split_hi_elem(C, g * 8) : g * 8);
            if constexpr (SPLIT) store_split8((uint16_t*)dp, split_group(C), m[o]);
            else {
                Vec8<E> r;
#pragma unroll
                for (int i = 0; i < 8; ++i) r.v[i] = to_elem<E>(m[o][i]);
                *(Vec8<E>*)dp = r;
            }
        }
    }
}

hipError_t launch_maxpool(const void* src, void* dst, int n, int H, int W, int C, int k, int stride,
                          int Ho, int Wo, const float* pre_scale, const float* pre_shift, int pre_relu,
                          int precision, hipStream_t s)
{
    const long total = (long)n * Ho * ((Wo + 3) / 4) * (C / 8);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (precision == kF32)
        hipLaunchKernelGGL(maxpool_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, n, H, W, C, k, stride, Ho, Wo, pre_scale, pre_shift, pre_relu);
    else if (precision == kF16X3)
        hipLaunchKernelGGL((maxpool_kernel<_Float16, true>), dim3(grid), dim3(256), 0, s, (const _Float16*)src, (_Float16*)dst, n, H, W, C, k, stride, Ho, Wo, pre_scale, pre_shift, pre_relu);
    else if (precision == kF16)
        hipLaunchKernelGGL(maxpool_kernel<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16*)src, (_Float16*)dst, n, H, W, C, k, stride, Ho, Wo, pre_scale, pre_shift, pre_relu);
    else
        hipLaunchKernelGGL(maxpool_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)dst, n, H, W, C, k, stride, Ho, Wo, pre_scale, pre_shift, pre_relu);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// head: 1x1 conv + BN + softmax + argmax (main.py:290: np.argmax over the softmax output, first
// maximum wins).  One thread per pixel; weights broadcast from LDS.
// ------------------------------------------------------------------------------------------------
template <typename E, bool SPLIT = false>
__global__ __launch_bounds__(256) void head_kernel(const HeadParams p)
{
    __shared__ float sw[64 * 8];
    __shared__ float ss[16];
    for (int i = threadIdx.x; i < p.cin * p.classes; i += 256) sw[i] = p.w[i];
    if (threadIdx.x < p.classes) { ss[threadIdx.x] = p.scale[threadIdx.x]; ss[8 + threadIdx.x] = p.shift[threadIdx.x]; }
    __syncthreads();
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= p.M) return;
    float logit[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) logit[c] = 0.f;
    const E* src = (const E*)p.src + (size_t)m * p.cin * (SPLIT ? 2 : 1);
    for (int g = 0; g < p.cin / 8; ++g) {
        const int e0 = SPLIT ? split_hi_elem(p.cin, g * 8) : g * 8;
        const Vec8<E> v = *(const Vec8<E>*)(src + e0);
        Vec8<E> vl;
        if constexpr (SPLIT) vl = *(const Vec8<E>*)(src + e0 + split_group(p.cin));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float xv = from_elem<E>(v.v[i]);
            if constexpr (SPLIT) xv += from_elem<E>(vl.v[i]);
            const float* wr = sw + (g * 8 + i) * p.classes;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < p.classes) logit[c] = fmaf(xv, wr[c], logit[c]);
        }
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c < p.classes) { logit[c] = logit[c] * ss[c] + ss[8 + c]; mx = fmaxf(mx, logit[c]); }
    float pr[8], sum = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c < p.classes) { pr[c] = expf(logit[c] - mx); sum += pr[c]; }
    int best = 0;
    float bestp = -1.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c < p.classes) {
            pr[c] = pr[c] / sum;
            if (pr[c] > bestp) { bestp = pr[c]; best = c; }
            if (p.probs) p.probs[(size_t)m * p.classes + c] = pr[c];
        }
    p.labels[m] = (uint8_t)best;
}

hipError_t launch_head(const HeadParams& p, int precision, hipStream_t s)
{
    const unsigned grid = (unsigned)((p.M + 255) / 256);
    if (precision == kF32) hipLaunchKernelGGL(head_kernel<float>, dim3(grid), dim3(256), 0, s, p);
    else if (precision == kF16X3) hipLaunchKernelGGL((head_kernel<_Float16, true>), dim3(grid), dim3(256), 0, s, p);
    else if (precision == kF16) hipLaunchKernelGGL(head_kernel<_Float16>, dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(head_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// stitch: page pixel (y,x) takes the label of its owner tile (closed form of the reference's
// crop-and-overwrite, main.py:294-364).  own_x[x] = (tile column i << 16) | x-inside-tile, own_y alike.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stitch_kernel(const uint8_t* tile_labels, int H, int W, const int* own_x,
                                                     const int* own_y, int nyf, int Hp, int Wp, uint8_t* out)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Hp * Wp) return;
    const int y = (int)(idx / Wp), x = (int)(idx - (long)y * Wp);
    const int ex = own_x[x], ey = own_y[y];
    const int t = (ex >> 16) * nyf + (ey >> 16);
    out[idx] = tile_labels[((size_t)t * H + (ey & 0xffff)) * W + (ex & 0xffff)];
}

hipError_t launch_stitch(const uint8_t* tile_labels, int H, int W, const int* own_x, const int* own_y,
                         int nyf, int Hp, int Wp, uint8_t* out, hipStream_t s)
{
    const long total = (long)Hp * Wp;
    hipLaunchKernelGGL(stitch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, tile_labels, H, W,
                       own_x, own_y, nyf, Hp, Wp, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void resize_labels_kernel(const uint8_t* labels, int H, int W, const int* map_y,
                                                            const int* map_x, int out_h, int out_w, uint8_t* out)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)out_h * out_w) return;
    const int y = (int)(idx / out_w), x = (int)(idx - (long)y * out_w);
    out[idx] = labels[(size_t)map_y[y] * W + map_x[x]];
}

hipError_t launch_resize_labels(const uint8_t* labels, int H, int W, const int* map_y, const int* map_x,
                                int out_h, int out_w, uint8_t* out, hipStream_t s)
{
    const long total = (long)out_h * out_w;
    hipLaunchKernelGGL(resize_labels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, labels, H, W,
                       map_y, map_x, out_h, out_w, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Stage glue around the border / layout models: cv2.erode / cv2.dilate with the reference's
// 5x5 kernel of ones (main.py:57) and the largest connected component of the border mask (main.py:394-404).
// ------------------------------------------------------------------------------------------------
// n iterations of a k x k min (erode) / max (dilate) filter with cv2's default border (the outside never wins:
// BORDER_CONSTANT with +inf / -inf) == ONE (n(k-1)+1)-wide filter over the window clipped to the image, separable.
// pass 0: along x, pass 1: along y.
// binarize: 0 = the plane as it is; 1 = t > 0 ? 255 : 0 (cv2.threshold(gray, 0, 255, THRESH_BINARY), main.py:395); 0x100 | label =
// t == label ? 255 : 0 (the class mask of get_text_region_contours_and_boxes, main.py:457-461)
__device__ __forceinline__ int morph_binarize(int t, int binarize)
{
    if (binarize & 0x100) return t == (binarize & 0xff) ? 255 : 0;
    return binarize ? (t > 0 ? 255 : 0) : t;
}
__global__ __launch_bounds__(256) void morph_pass_kernel(const uint8_t* src, uint8_t* dst, int H, int W, int radius, int is_max,
                                                         int vertical, int binarize)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)H * W) return;
    const int y = (int)(idx / W), x = (int)(idx - (long)y * W);
    int v = is_max ? 0 : 255;
    if (!vertical) {
        const int lo = max(x - radius, 0), hi = min(x + radius, W - 1);
        const uint8_t* row = src + (size_t)y * W;
        for (int q = lo; q <= hi; ++q) {
            int t = row[q];
            t = morph_binarize(t, binarize);
            v = is_max ? max(v, t) : min(v, t);
        }
    } else {
        const int lo = max(y - radius, 0), hi = min(y + radius, H - 1);
        for (int q = lo; q <= hi; ++q) {
            const int t = src[(size_t)q * W + x];
            v = is_max ? max(v, t) : min(v, t);
        }
    }
    dst[idx] = (uint8_t)v;
}

// The same pass, FOUR horizontally adjacent output pixels per thread (W % 4 == 0: every row starts on a 4-byte boundary): the row pass reads
// the 4 + 2 radius window bytes once for its four outputs, the column pass reads one 32-bit word per row.  A thread per pixel issued
// 2 radius + 1 byte loads per output: 147 us per pass on a 4200 x 3000 mask at radius 12 (extract_page's six dilations).
template <int IS_MAX>
__global__ __launch_bounds__(256) void morph_pass4_kernel(const uint8_t* src, uint8_t* dst, int H, int W, int radius, int vertical, int binarize)
{
    const int W4 = W >> 2;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long)H * W4) return;
    const int y = (int)(g / W4), x0 = (int)(g - (long)y * W4) * 4;
    constexpr int ID = IS_MAX ? 0 : 255;
    auto op = [](int a, int b) __attribute__((always_inline)) { return IS_MAX ? max(a, b) : min(a, b); };
    int o0 = ID, o1 = ID, o2 = ID, o3 = ID;
    if (!vertical) {
        const uint8_t* row = src + (size_t)y * W;
        // window of output j = [x0 + j - radius, x0 + j + radius]: bytes x0 - radius + 3 .. x0 + radius are common to all four
        int mid = ID;
        for (int q = x0 - radius + 3; q <= x0 + radius; ++q) {
            if ((unsigned)q < (unsigned)W) { const int t = morph_binarize(row[q], binarize); mid = op(mid, t); }
        }
        int e[6];                                           // the three bytes on either side of the common part
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ql = x0 - radius + j, qr = x0 + radius + 1 + j;
            int tl = ID, tr = ID;
            if ((unsigned)ql < (unsigned)W) tl = morph_binarize(row[ql], binarize);
            if ((unsigned)qr < (unsigned)W) tr = morph_binarize(row[qr], binarize);
            e[j] = tl; e[3 + j] = tr;
        }
        o0 = op(mid, op(e[0], op(e[1], e[2])));
        o1 = op(mid, op(e[1], op(e[2], e[3])));
        o2 = op(mid, op(e[2], op(e[3], e[4])));
        o3 = op(mid, op(e[3], op(e[4], e[5])));
    } else {
        const int lo = max(y - radius, 0), hi = min(y + radius, H - 1);
        for (int q = lo; q <= hi; ++q) {
            const uint32_t t = *(const uint32_t*)(src + (size_t)q * W + x0);
            o0 = op(o0, (int)(t & 255u)); o1 = op(o1, (int)((t >> 8) & 255u)); o2 = op(o2, (int)((t >> 16) & 255u)); o3 = op(o3, (int)(t >> 24));
        }
    }
    *(uint32_t*)(dst + (size_t)y * W + x0) = (uint32_t)o0 | ((uint32_t)o1 << 8) | ((uint32_t)o2 << 16) | ((uint32_t)o3 << 24);
}

hipError_t launch_morph(const uint8_t* src, uint8_t* tmp, uint8_t* dst, int H, int W, int radius, int is_max, int binarize, hipStream_t s)
{
    if ((W & 3) == 0 && radius >= 2 && (((uintptr_t)src | (uintptr_t)tmp | (uintptr_t)dst) & 3) == 0) {
        const unsigned grid4 = (unsigned)(((long)H * (W >> 2) + 255) / 256);
        if (is_max) {
            hipLaunchKernelGGL(morph_pass4_kernel<1>, dim3(grid4), dim3(256), 0, s, src, tmp, H, W, radius, 0, binarize);
            hipLaunchKernelGGL(morph_pass4_kernel<1>, dim3(grid4), dim3(256), 0, s, (const uint8_t*)tmp, dst, H, W, radius, 1, 0);
        } else {
            hipLaunchKernelGGL(morph_pass4_kernel<0>, dim3(grid4), dim3(256), 0, s, src, tmp, H, W, radius, 0, binarize);
            hipLaunchKernelGGL(morph_pass4_kernel<0>, dim3(grid4), dim3(256), 0, s, (const uint8_t*)tmp, dst, H, W, radius, 1, 0);
        }
        return hipGetLastError();
    }
    const unsigned grid = (unsigned)(((long)H * W + 255) / 256);
    hipLaunchKernelGGL(morph_pass_kernel, dim3(grid), dim3(256), 0, s, src, tmp, H, W, radius, is_max, 0, binarize);
    hipLaunchKernelGGL(morph_pass_kernel, dim3(grid), dim3(256), 0, s, (const uint8_t*)tmp, dst, H, W, radius, is_max, 1, 0);
    return hipGetLastError();
}

// 8-connected components of mask > 0 by union-find on pixel indices (roots = smallest index of a component = its first
// pixel in raster order).  parent values only ever decrease and every value ever stored is an ancestor, so a stale read
// (another CU's update not yet visible) costs a retry, never a wrong merge: links are made by atomicMin, whose RETURN
// value is what decides.
__device__ inline int cc_find(int* parent, int i)
{
    int p = parent[i];
    while (p != i) {
        const int g = parent[p];
        if (g != p) parent[i] = g;                          // path halving (any ancestor is a valid parent)
        i = p;
        p = g;
    }
    return i;
}
__device__ inline void cc_union(int* parent, int a, int b)
{
    for (;;) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&parent[b], a);           // hang the larger root under the smaller one
        if (old == b) return;
        b = old;                                            // b had been linked meanwhile: go on from its parent
    }
}
// Wave-aggregated atomics: a page mask is mostly ONE component, so nearly every lane of a wave targets the same root -- 11 M
// single-address atomics took 125 ms before the lanes of a wave were combined (one atomic per wave and distinct root).
__device__ inline void wave_add_by_root(int* dst, int root, int val)
{
    bool pending = root >= 0 && val != 0;
    while (__builtin_amdgcn_ballot_w64(pending)) {
        const unsigned long long live = __builtin_amdgcn_ballot_w64(pending);
        const int leader = __builtin_ctzll(live);
        const int r = __builtin_amdgcn_readlane(root, leader);
        const bool mine = pending && root == r;
        int v = mine ? val : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&dst[r], v);
        pending = pending && !mine;
    }
}
__device__ inline void wave_minmax_by_root(int* dmin, int* dmax, int root, int lo, int hi)
{
    bool pending = root >= 0;
    while (__builtin_amdgcn_ballot_w64(pending)) {
        const unsigned long long live = __builtin_amdgcn_ballot_w64(pending);
        const int leader = __builtin_ctzll(live);
        const int r = __builtin_amdgcn_readlane(root, leader);
        const bool mine = pending && root == r;
        int a = mine ? lo : (1 << 30), b = mine ? hi : -1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { a = min(a, __shfl_xor(a, off)); b = max(b, __shfl_xor(b, off)); }
        if ((int)(threadIdx.x & 63) == leader) { atomicMin(&dmin[r], a); atomicMax(&dmax[r], b); }
        pending = pending && !mine;
    }
}
// parent = first pixel of the horizontal run (keeps the union-find trees flat).  One WAVE per row, 64 pixels per step: the run
// starts of a chunk come from the lanes' mask ballot (highest clear bit below the lane), a run that crosses into the next chunk
// is carried in a scalar.  (Round 3 walked a row per THREAD -- 3 000 dependent, uncoalesced steps: 0.70 ms at 4200 x 3000.)
__global__ __launch_bounds__(64) void cc_rows_kernel(const uint8_t* mask, int* parent, int* count, int H, int W)
{
    const int y = blockIdx.x, lane = threadIdx.x;
    if (y >= H) return;
    const long row = (long)y * W;
    int carry = -1;                                         // start of the run that reaches the left edge of the chunk, or -1
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const bool m = x < W && mask[row + x] != 0;
        const unsigned long long bits = __builtin_amdgcn_ballot_w64(m);
        const unsigned long long below = lane ? (~bits & ((1ull << lane) - 1ull)) : 0ull;      // clear bits under this lane
        int start = below ? (int)(row + x0 + (64 - __builtin_clzll(below))) : (carry >= 0 ? carry : (int)(row + x0));
        if (x < W) {
            parent[row + x] = m ? start : -1;
            count[row + x] = 0;
        }
        const int last = __builtin_amdgcn_readlane(m ? start : -1, 63);
        carry = last;                                       // lane 63 set: its run goes on (x0 + 64 <= W there, or the loop ends)
    }
}
__global__ __launch_bounds__(256) void cc_link_kernel(const uint8_t* mask, int* parent, int H, int W)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)H * W) return;
    const int y = (int)(idx / W), x = (int)(idx - (long)y * W);
    if (y == 0 || !mask[idx]) return;
    const long up = idx - W;
    // Only the LEFT END of a contact between two row runs makes the union: a pixel whose left neighbour is set (same run) and whose
    // upper-left pixel is set too (same upper run as `up`) repeats a union its left neighbour is responsible for -- inside a blob
    // that is every pixel but one per run pair (12.6 M root walks on a page mask: 1.5 ms; now ~ the number of runs).  Same for the
    // diagonal links: through a set left / right neighbour the link exists already (that neighbour sees the pixel as its N).
    const bool left = x > 0 && mask[idx - 1];
    if (mask[up]) {
        if (!(left && mask[up - 1])) cc_union(parent, (int)idx, (int)up);      // N set: NW / NE are joined to it through their row runs
        return;
    }
    if (x > 0 && mask[up - 1] && !left) cc_union(parent, (int)idx, (int)(up - 1));
    if (x + 1 < W && mask[up + 1] && !mask[idx + 1]) cc_union(parent, (int)idx, (int)(up + 1));
}
// flatten + pixel count per root (a lane merges its pixels while their root stays the same, equal roots across the lanes of a wave are
// merged by wave_add_by_root: one atomic per wave and distinct root)
__global__ __launch_bounds__(256) void cc_count_kernel(int* parent, int* count, long n)
{
    // A WAVE per 4 096 consecutive pixels, 64 consecutive pixels per step (coalesced); each lane merges the pixels of its own column of the
    // 64 x 64 block while their root stays the same.  (Up to round 4 a THREAD walked 64 consecutive pixels: every load of the wave touched
    // 64 lines, 0.65 ms at 4200 x 3000; sums do not care how the pixels are dealt to the lanes.)
    const long wave_base = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4096;
    const int lane = threadIdx.x & 63;
    int cur = -1, run = 0;
    for (int k = 0; k < 64; ++k) {
        const long i = wave_base + k * 64 + lane;
        int r = -1;
        if (i < n && parent[i] >= 0) { r = cc_find(parent, (int)i); parent[i] = r; }
        if (__builtin_amdgcn_ballot_w64(r >= 0 && r != cur)) {     // wave-uniform branch: some lane meets another root (background pixels end nothing)
            const bool flush = r >= 0 && r != cur;
            wave_add_by_root(count, flush ? cur : -1, run);
            if (flush) { cur = r; run = 0; }
        }
        run += r >= 0;
    }
    wave_add_by_root(count, cur, run);
}
// cc_count_kernel's `parent[i] = root` races with the path halving of OTHER threads' walks through i (cc_find stores an ancestor it read
// before the root was written): a few pixels per million were left pointing at a non-root ancestor, and the kernels below, which take
// parent[] for the root, credited their cells / extents to that ancestor -- the lower-bound area of a blob came out a little short in
// some runs, so equal-area blobs were ranked at random (round 5: tools/border_repeat_probe.py, three 31 x 33 blobs).  This pass runs
// with no halving writer active: every store is a root, a reader sees an ancestor or the root, the walk ends at the root either way.
__global__ __launch_bounds__(256) void cc_flatten_kernel(int* parent, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int p = parent[i];
    if (p < 0) return;
    int q = parent[p];
    if (q == p) return;                                     // already at its root (nearly every pixel)
    while (q != p) { p = q; q = parent[p]; }                // read-only walk
    parent[i] = p;
}
// ---- ranking by cv2.contourArea (main.py:399-401).  The outer contour cv2.findContours traces runs through the centres of the
// component's boundary pixels (8-connected steps); its polygon area is, for the component with its holes filled, the number of
// 2 x 2 pixel cells that are completely inside plus half the number of cells with exactly three pixels inside (a diagonal
// step cuts such a cell in half).  Counted over the component AS IT IS (holes not filled) that sum is a LOWER bound of the
// contour area, and (w - 1)(h - 1) of the bounding box an UPPER bound: the device picks the component with the largest lower
// bound and reports whether any other component's upper bound could beat it; only then does the host trace contours.
// Areas are kept doubled (integers).  Two set pixels of one 2 x 2 cell are 8-neighbours, i.e. of one component.
__global__ __launch_bounds__(256) void cc_cell_area_kernel(const int* parent, int* area2, int H, int W)
{
    // a WAVE per 64 strips of 64 cells (strips in row-major order of the cell rows): step k = strip k of the wave, a cell per lane
    // (coalesced); a lane sums its cells while their root stays the same and flushes through wave_add_by_root (one atomic per wave and
    // distinct root).  A thread per CELL sent 197 k atomics to the one root of a page mask: 2.2 ms.
    const long strips_per_row = (W - 1 + 63) / 64, n_strips = strips_per_row * (H - 1);
    const long first = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    const int lane = threadIdx.x & 63;
    int cur = -1, sum = 0;
    for (int k = 0; k < 64; ++k) {
        const long sidx = first + k;
        const bool live = sidx < n_strips;
        const int y = live ? (int)(sidx / strips_per_row) : 0;
        const int x = live ? (int)(sidx - (long)y * strips_per_row) * 64 + lane : 0;
        int root = -1, val = 0;
        if (live && x < W - 1) {
            const long i = (long)y * W + x;
            const int a = parent[i], b = parent[i + 1], c2 = parent[i + W], d = parent[i + W + 1];
            const int n = (a >= 0) + (b >= 0) + (c2 >= 0) + (d >= 0);
            if (n >= 3) { root = a >= 0 ? a : b; val = n == 4 ? 2 : 1; }      // (parent[] is flat after cc_count_kernel)
        }
        if (__builtin_amdgcn_ballot_w64(root != cur && val != 0)) {          // some lane's run of one root ends (wave-uniform branch)
            const bool flush = root != cur && val != 0;
            wave_add_by_root(area2, flush ? cur : -1, sum);
            if (flush) { cur = root; sum = 0; }
        }
        sum += val;
    }
    wave_add_by_root(area2, cur, sum);
}
// bounding box per root: {min x, min y, max x, max y} in four arrays indexed by root (initialised by cc_box_init_kernel)
__global__ __launch_bounds__(256) void cc_box_init_kernel(const int* parent, int* area2, int* bx0, int* by0, int* bx1, int* by1, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    area2[i] = 0;
    if (parent[i] == (int)i) { bx0[i] = 1 << 30; by0[i] = 1 << 30; bx1[i] = -1; by1[i] = -1; }
}
__global__ __launch_bounds__(256) void cc_box_kernel(const int* parent, int* bx0, int* by0, int* bx1, int* by1, int H, int W)
{
    // a WAVE per 64 strips of 64 pixels (row-major strips), a pixel per lane per step (coalesced); a lane keeps the x / y extent of its
    // pixels while their root stays the same, equal roots across the lanes are merged by wave_minmax_by_root
    const long strips_per_row = (W + 63) / 64, n_strips = strips_per_row * H;
    const long first = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    const int lane = threadIdx.x & 63;
    int cur = -1, lox = 0, hix = 0, loy = 0, hiy = 0;
    for (int k = 0; k < 64; ++k) {
        const long sidx = first + k;
        const bool live = sidx < n_strips;
        const int y = live ? (int)(sidx / strips_per_row) : 0;
        const int x = live ? (int)(sidx - (long)y * strips_per_row) * 64 + lane : 0;
        const int r = (live && x < W) ? parent[(long)y * W + x] : -1;
        if (__builtin_amdgcn_ballot_w64(r >= 0 && r != cur)) {     // some lane meets another root: flush its extent (wave-uniform branch;
            const bool flush = r >= 0 && r != cur;                 // background pixels and the padding of a row's last strip end nothing)
            wave_minmax_by_root(bx0, bx1, flush ? cur : -1, lox, hix);
            wave_minmax_by_root(by0, by1, flush ? cur : -1, loy, hiy);
            if (flush) { cur = r; lox = x; hix = x; loy = y; hiy = y; }
        }
        if (r >= 0) { lox = min(lox, x); hix = max(hix, x); loy = min(loy, y); hiy = max(hiy, y); }
    }
    wave_minmax_by_root(bx0, bx1, cur, lox, hix);
    wave_minmax_by_root(by0, by1, cur, loy, hiy);
}
// best = max over roots of (area2 lower bound, then LARGEST root index: the reference's np.argmax over OpenCV's contour list, which
// runs in reverse discovery order, keeps the last-discovered of equal areas -- api.hip host_largest_contour); key = area2 << 32 | root + 1
__global__ __launch_bounds__(256) void cc_best_area_kernel(const int* parent, const int* area2, long n, unsigned long long* best)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    unsigned long long key = 0;
    if (i < n && parent[i] == (int)i) key = ((unsigned long long)(unsigned)area2[i] << 32) | ((unsigned)i + 1u);
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(key, off);
        key = o > key ? o : key;
    }
    if ((threadIdx.x & 63) == 0 && key) atomicMax(best, key);
}
// out[0..3] = bounding box of the best root, out[4] = its pixel count, out[5] = number of RIVALS -- other roots whose upper bound
// 2 (w - 1)(h - 1) exceeds the best lower bound (or ties it with a larger index) -- and out[6..] the first kCcMaxRivals of them:
// with rivals the ranking is not decided here
__global__ __launch_bounds__(256) void cc_decide_kernel(const int* parent, const int* count, const int* bx0, const int* by0, const int* bx1,
                                                        const int* by1, long n, const unsigned long long* best, int* out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long key = *best;
    if (!key || i >= n || parent[i] != (int)i) return;
    const int root = (int)((unsigned)(key & 0xffffffffu) - 1u);
    const long long best_lo = (long long)(key >> 32);
    if ((int)i == root) {
        out[0] = bx0[i]; out[1] = by0[i]; out[2] = bx1[i]; out[3] = by1[i]; out[4] = count[i];
        return;
    }
    const long long hi2 = 2ll * (bx1[i] - bx0[i]) * (by1[i] - by0[i]);
    if (hi2 > best_lo || (hi2 == best_lo && (int)i > root)) {
        const int k = atomicAdd(&out[5], 1);                     // out[5] = number of undecided rivals, out[6 + k] = their roots
        if (k < kCcMaxRivals) out[6 + k] = (int)i;
    }
}

// d_out: int[6 + kCcMaxRivals] = {min x, min y, max x, max y, pixels, rivals, rival roots...} of the component with the largest
// contour-area lower bound ({2^30, 2^30, -1, -1, 0, 0} if the mask is empty); d_best: its (area2 << 32 | root + 1) key.  scratch: five int arrays of H * W.
hipError_t launch_largest_contour(const uint8_t* mask, int H, int W, int* parent, int* count, int* area2, int* bx0, int* by0, int* bx1,
                                  int* by1, unsigned long long* d_best, int* d_out, hipStream_t s)
{
    const long n = (long)H * W;
    static const int init_out[6] = {1 << 30, 1 << 30, -1, -1, 0, 0};
    hipError_t e = hipMemsetAsync(d_best, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(d_out, init_out, sizeof(init_out), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cc_rows_kernel, dim3((unsigned)H), dim3(64), 0, s, mask, parent, count, H, W);
    hipLaunchKernelGGL(cc_link_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mask, parent, H, W);
    hipLaunchKernelGGL(cc_count_kernel, dim3((unsigned)((n + 256 * 64 - 1) / (256 * 64))), dim3(256), 0, s, parent, count, n);
    hipLaunchKernelGGL(cc_flatten_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, parent, n);
    hipLaunchKernelGGL(cc_box_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const int*)parent, area2, bx0, by0, bx1, by1, n);
    if (H > 1 && W > 1) {
        const long cell_strips = (long)((W - 1 + 63) / 64) * (H - 1);
        hipLaunchKernelGGL(cc_cell_area_kernel, dim3((unsigned)((cell_strips + 255) / 256)), dim3(256), 0, s, (const int*)parent, area2, H, W);
    }
    const long strips = (long)((W + 63) / 64) * H;
    hipLaunchKernelGGL(cc_box_kernel, dim3((unsigned)((strips + 255) / 256)), dim3(256), 0, s, (const int*)parent, bx0, by0, bx1, by1, H, W);
    hipLaunchKernelGGL(cc_best_area_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const int*)parent, (const int*)area2, n, d_best);
    hipLaunchKernelGGL(cc_decide_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const int*)parent, (const int*)count, (const int*)bx0,
                       (const int*)by0, (const int*)bx1, (const int*)by1, n, (const unsigned long long*)d_best, d_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// deskew_profile_kernel -- the rotate-and-project of the deskew search (main.py:1601-1718): for every angle of a sweep,
// the region mask (centred on a zero square of side S, main.py:1613-1621) is rotated as rotate_image does (main.py:159-163:
// cv2.warpAffine, INTER_CUBIC, BORDER_REPLICATE), binarised (!= 0, main.py:1642) and summed along its rows (main.py:1546).
// One block per (row, angle).  OpenCV's arithmetic [EXT, 4.5.1 imgwarp.cpp]: source coordinates in fixed point with 5
// fractional bits (AB_BITS = 10, round-half-even), 4 x 4 taps with the float bicubic table (A = -0.75), taps accumulated
// one by one in float64.  Floating-point contraction is off: the integer coordinates must come out of the same roundings
// as on the host.  HBM-trivial (the mask is L2-resident); 16 taps are only evaluated where the 4 x 4 window meets the patch.
// ------------------------------------------------------------------------------------------------
struct DeskewParams {
    const uint8_t* mask;      // [H][W] region mask (device)
    int H, W, S, top, left;   // square side, placement of the patch inside the square
    const double* minv;       // [n_angles][6] inverse affine maps (destination -> source), row-major 2 x 3
    const float* cubic;       // [32][4]
    int* counts;              // [n_angles][S]
};

__global__ __launch_bounds__(256) void deskew_profile_kernel(const DeskewParams p)
{
#pragma clang fp contract(off)
    __shared__ float tab[32 * 4];
    __shared__ int total;
    const int y = blockIdx.x, a = blockIdx.y, tid = threadIdx.x;
    if (tid < 128) tab[tid] = p.cubic[tid];
    if (tid == 0) total = 0;
    __syncthreads();
    const double* m = p.minv + (size_t)a * 6;
    const long long X0 = __double2ll_rn((m[1] * (double)y + m[2]) * 1024.0) + 16;
    const long long Y0 = __double2ll_rn((m[4] * (double)y + m[5]) * 1024.0) + 16;
    int cnt = 0;
    for (int x = tid; x < p.S; x += 256) {
        const long long X = (X0 + __double2ll_rn(m[0] * (double)x * 1024.0)) >> 5;
        const long long Y = (Y0 + __double2ll_rn(m[3] * (double)x * 1024.0)) >> 5;
        long long sx = X >> 5, sy = Y >> 5;
        sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
        sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
        const int ax = (int)(X & 31), ay = (int)(Y & 31);
        // window rows sy-1 .. sy+2, columns sx-1 .. sx+2, clamped to the square; non-zero source pixels only inside the patch
        const int x_lo = (int)min(max(sx - 1, 0LL), (long long)p.S - 1), x_hi = (int)min(max(sx + 2, 0LL), (long long)p.S - 1);
        const int y_lo = (int)min(max(sy - 1, 0LL), (long long)p.S - 1), y_hi = (int)min(max(sy + 2, 0LL), (long long)p.S - 1);
        if (x_hi < p.left || x_lo >= p.left + p.W || y_hi < p.top || y_lo >= p.top + p.H) continue;
        double sum = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int yy = (int)min(max(sy - 1 + r, 0LL), (long long)p.S - 1) - p.top;
            const float wy = tab[ay * 4 + r];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int xx = (int)min(max(sx - 1 + cc, 0LL), (long long)p.S - 1) - p.left;
                const float w2 = wy * tab[ax * 4 + cc];                          // the 2-D table entry: a float product
                const bool in = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
                const double v = in ? (double)p.mask[(size_t)yy * p.W + xx] : 0.0;
                sum = sum + v * (double)w2;
            }
        }
        cnt += sum != 0.0;
    }
    // wave reduction, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((tid & 63) == 0 && cnt) atomicAdd(&total, cnt);
    __syncthreads();
    if (tid == 0) p.counts[(size_t)a * p.S + y] = total;
}

hipError_t launch_deskew_profiles(const uint8_t* mask, int H, int W, int S, int top, int left, const double* minv, const float* cubic,
                                  int n_angles, int* counts, hipStream_t s)
{
    DeskewParams p;
    p.mask = mask; p.H = H; p.W = W; p.S = S; p.top = top; p.left = left; p.minv = minv; p.cubic = cubic; p.counts = counts;
    hipLaunchKernelGGL(deskew_profile_kernel, dim3(S, n_angles), dim3(256), 0, s, p);
    return hipGetLastError();
}

template <typename E>
__global__ __launch_bounds__(256) void to_f32_kernel(const E* src, float* dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = from_elem<E>(src[i]);
}

__global__ __launch_bounds__(256) void split_to_f32_kernel(const _Float16* src, float* dst, size_t n, int C)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;         // i = pixel * C + channel
    if (i >= n) return;
    const size_t pix = i / C;
    const int ch = (int)(i - pix * C);
    const size_t e0 = pix * 2 * C + split_hi_elem(C, ch);           // channel groups [G hi][G lo] (internal.h)
    dst[i] = (float)src[e0] + (float)src[e0 + split_group(C)];
}

hipError_t launch_split_to_f32(const void* src, float* dst, size_t npix, int C, hipStream_t s)
{
    const size_t n = npix * C;
    hipLaunchKernelGGL(split_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const _Float16*)src, dst, n, C);
    return hipGetLastError();
}

// u8 label plane -> the reference's return layout: three identical channels (main.py:366, 380)
__global__ __launch_bounds__(256) void replicate3_kernel(const uint8_t* src, uint8_t* dst, size_t n4)
{
    // 4 labels -> 12 bytes per thread
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const uint32_t v = ((const uint32_t*)src)[i];
    const uint32_t a = v & 0xff, b = (v >> 8) & 0xff, c = (v >> 16) & 0xff, d = v >> 24;
    uint32_t* o = (uint32_t*)dst + i * 3;
    o[0] = a | (a << 8) | (a << 16) | (b << 24);
    o[1] = b | (b << 8) | (c << 16) | (c << 24);
    o[2] = c | (d << 8) | (d << 16) | (d << 24);
}

hipError_t launch_replicate3(const uint8_t* src, uint8_t* dst, size_t n, hipStream_t s)
{
    const size_t n4 = (n + 3) / 4;                   // buffers are padded to a multiple of 4 labels by the caller
    hipLaunchKernelGGL(replicate3_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, src, dst, n4);
    return hipGetLastError();
}

hipError_t launch_to_f32(const void* src, float* dst, size_t n, int precision, hipStream_t s)
{
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (precision == kF32) hipLaunchKernelGGL(to_f32_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, dst, n);
    else if (precision == kF16) hipLaunchKernelGGL(to_f32_kernel<_Float16>, dim3(grid), dim3(256), 0, s, (const _Float16*)src, dst, n);
    else hipLaunchKernelGGL(to_f32_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, (const uint16_t*)src, dst, n);
    return hipGetLastError();
}

}  // namespace sbbseg
