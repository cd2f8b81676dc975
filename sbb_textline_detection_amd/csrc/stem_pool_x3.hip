// stem_pool_x3.hip -- the network stem AND its max-pool in one launch, split-fp16 mode (kF16X3), round 4.
//
//   f1   = BN-less 7x7 / stride-2 conv of the image (PAIRS input form: 7 rows x 4 two-pixel granules)      224 x 224 x 64, written once
//   pool = maxpool3x3 / stride 2 / valid over ReLU(bn_conv1(f1))                                            111 x 111 x 64
//
// Unfused (stem_conv_pairs_x3 + maxpool_kernel) the pool re-reads the 1.8 GB (140 patches) the stem wrote 0.6 ms earlier: 0.52 ms
// of pure HBM traffic.  Here the pool is taken from the values the stem's epilogue holds anyway:
//   * a block owns one HALF of the channels (32) of a 16-row strip of the output and walks its 16 x 16 tiles left to right; two
//     blocks per CU (65 KB of LDS, < 256 VGPRs each) -- one block's epilogue / stores run under the other's MFMAs, which the
//     one-wave-per-SIMD, 64-channel kernel could not do (MFMA pipe 42 % busy)
//   * pool windows overlap their neighbours by one row / column: a tile produces the eight windows per row that END inside it
//     (columns 16 tx - 2 + 2 j .. 16 tx + 2 j); the two columns they need from the previous tile are kept in LDS (that is why a
//     block walks a strip), and the one extra row they need below the tile (row 16 ty + 16) is recomputed: one more pixel fragment
//     per tile (+ 6 % MFMAs), taken by wave 3 (the kernel is HBM-bound: the other block of the CU fills the imbalance)
//   * vertical maxima of the four rows a wave owns are taken in registers (rows are register-indexed), the horizontal ones go
//     through a 32 KB LDS stage; every pooled value is the maximum over the SAME fp32 numbers maxpool_kernel would read back
//     (hi + lo of the stored f1, one fma, ReLU): bit-identical to the two-launch form (tests/test_gpu_parity.py)
//   * blocks b and b + 8 (same XCD under the round-robin dispatch) take the two channel halves of the same strip: the input halo
//     is fetched into that XCD's L2 once
#include "internal.h"

namespace sbbseg {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;

constexpr int kSlots = 20;                          // granules per LDS halo row (x + g <= 15 + 3)
constexpr int kHaloRows = 39;                       // 2 * 17 + 5: 16 output rows + the extra row below
constexpr int kHaloInstr = (kHaloRows * kSlots + 63) / 64;      // 13 wave-instructions of 64 granules per plane
constexpr int kPlaneBytes = kHaloInstr * 1024;
constexpr int kColStride = 36;                      // floats per staged (row, column): 32 channels + pad (144 B: b128-aligned, spreads the banks)
constexpr int kStageVBytes = 8 * 18 * kColStride * 4;           // vertical maxima [8 pooled rows][2 saved + 16 columns]
constexpr int kStageRBytes = 5 * 16 * kColStride * 4;           // first row of every wave's band + the extra row [5][16 columns]
constexpr int kSaveBytes = 2 * 8 * 2 * kColStride * 4;          // [tile parity][8 pooled rows][2 columns]
constexpr int kCstBytes = 4 * 64 * 4;               // stem scale * wmul, stem shift, pool scale, pool shift
constexpr int kWloBytes = 7 * 2 * 1024;              // the lo weight fragments of this block's two row blocks [7 ky][2 m][64 lanes x 16 B]
constexpr int kStemPoolLdsBytes = 2 * kPlaneBytes + kStageVBytes + kStageRBytes + kSaveBytes + kCstBytes + kWloBytes;      // 78 848: two blocks per CU
constexpr int kStemPoolF16LdsBytes = kPlaneBytes + kStageVBytes + kStageRBytes + kSaveBytes + kCstBytes;                  // plain fp16 mode: one plane, no lo weights

__device__ inline f4_t mma(h8_t a, h8_t b, f4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// One LDS-DMA wave-instruction (lane l's 16 bytes at gsrc(l) land at lds_dst + 16 l) hidden from the compiler: hipcc drains a
// builtin LDS-DMA (s_waitcnt vmcnt(0)) in front of the next LDS access of the same basic-block chain, whatever it touches -- here
// the epilogue's constant reads right behind the issue -- so the halo of the next tile could never be in flight during the epilogue.
// The wait for these loads is the explicit vmcnt(0) at the top of the tile loop.  (M0 is written in the statement that reads it.)
__device__ inline void glds16_hidden(const void* gsrc, uint32_t lds_dst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ inline void split1(float v, _Float16& hi, _Float16& lo)
{
    v = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);      // = fminf(fmaxf(v, -65504), 65504) for every non-NaN v, one instruction
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

}  // namespace

// X3: split mode (text above).  !X3: the plain fp16 mode, same structure with one plane: a PAIRS granule is 16 bytes, one MFMA per
// product, f1 and the pooled tensor are fp16 ([64] halves per pixel), and what the pool reads back is the ROUNDED f1 value (maxpool_kernel
// reads the stored fp16): bit-identical to stem_conv_pairs + maxpool_kernel there too.
template <bool X3>
__global__ __launch_bounds__(256, 2) void stem_pool(const StemParams p)
{
    constexpr int NPL = X3 ? 2 : 1;                                 // halo planes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;                                              // hi plane | lo plane
    float* stageV = (float*)(smem + NPL * kPlaneBytes);
    float* stageR = (float*)(smem + NPL * kPlaneBytes + kStageVBytes);
    float* save = (float*)(smem + NPL * kPlaneBytes + kStageVBytes + kStageRBytes);
    float* cst = (float*)(smem + NPL * kPlaneBytes + kStageVBytes + kStageRBytes + kSaveBytes);
    char* wlo_lds = smem + NPL * kPlaneBytes + kStageVBytes + kStageRBytes + kSaveBytes + kCstBytes;      // (split mode only)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fg = lane >> 4;
    const int tiles_x = p.Wo / 16, tiles_y = p.Ho / 16;
    const int n_strips = p.n * tiles_y;
    // block = (slot, channel half, XCD): strips slot * 8 + xcd, + 8 * slots, ... ; the two halves of a strip sit 8 blocks apart (same XCD).
    // (Half a grid apart -- most likely the same CU -- measured 4 % slower.)
    const int xcd = blockIdx.x & 7, half = (blockIdx.x >> 3) & 1, slot = blockIdx.x >> 4, n_slots = gridDim.x >> 4;

    if (tid < 64) {
        cst[tid] = p.scale[tid] * p.wmul; cst[64 + tid] = p.shift[tid];
        cst[128 + tid] = p.pool_scale[tid]; cst[192 + tid] = p.pool_shift[tid];
    }
    // wfrag = [hi | lo][7 ky][4 mi][64 lanes]; this block's row blocks mi = 2 half, 2 half + 1.  The hi fragments (two MFMAs per product)
    // live in 56 VGPRs, the lo fragments (one) in LDS -- all 112 in registers left the 256-register budget of two blocks per CU 39 short
    h8_t whi[7][2];
    {
        const uint4* src = (const uint4*)p.wfrag + lane;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int m = 0; m < 2; ++m) whi[ky][m] = __builtin_bit_cast(h8_t, src[(size_t)(ky * 4 + 2 * half + m) * 64]);
        if constexpr (X3)
            for (int i = tid; i < 7 * 2 * 64; i += 256) {
                const int f = i >> 6, l = i & 63;                   // f = ky * 2 + m
                ((uint4*)wlo_lds)[i] = ((const uint4*)p.wfrag)[(size_t)(28 + (f >> 1) * 4 + 2 * half + (f & 1)) * 64 + l];
            }
    }
    const int c0 = half * 32 + fg * 8;              // this lane's 8 channels in the epilogue (MFMA rows fg * 4 + q of row blocks 2 half, 2 half + 1)

    const uint32_t halo_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)halo);      // LDS byte address
    auto issue_halo = [&](int n, int ty, int tx) __attribute__((always_inline)) {
        const int rows = ty == tiles_y - 1 ? kHaloRows - 2 : kHaloRows;      // no row below the last strip
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int j = 0; j < (kHaloInstr + 3) / 4; ++j) {
                const int ii = wave + 4 * j;
                if (ii < kHaloInstr) {
                    const int s = ii * 64 + lane;
                    const int r = s / kSlots, cc = s - r * kSlots;
                    const int Y = 32 * ty + r, X = 16 * tx + cc;
                    uint32_t off = (uint32_t)((n * p.PHt + Y) * p.PWt + X) * (uint32_t)(16 * NPL) + (uint32_t)(pl * 16 + kZeroHeaderBytes);
                    off = r < rows ? off : 0u;
                    glds16_hidden(p.pairs + off, halo_lds + (uint32_t)(pl * kPlaneBytes + ii * 1024));
                }
            }
    };

    int first = 1;
    int par = 0;                                    // parity of the tile inside its strip walk: which save buffer is read
    for (int strip = slot * 8 + xcd; strip < n_strips; strip += n_slots * 8) {
        const int n = strip / tiles_y, ty = strip - n * tiles_y;
        const bool last_ty = ty == tiles_y - 1;
        for (int tx = 0; tx < tiles_x; ++tx) {
            // B1: this tile's halo landed; the previous tile's pool stage is over (stage / save may be rewritten).  vmcnt retires in issue
            // order: behind a tile's halo loads every wave issues its 8 f1 stores (+ 0..2 pooled ones), so "all but the youngest 8" covers
            // the halo and leaves the stores in flight.  Raw barriers: __syncthreads() would drain those stores (vmcnt(0)) at every tile.
            if (first) { issue_halo(n, ty, tx); first = 0; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            else if constexpr (X3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // (plain mode: 4 f1 stores + 0..1 pooled one per wave behind the halo loads)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");

            f4_t acc[2][4], accx[2] = {(f4_t){0.f, 0.f, 0.f, 0.f}, (f4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[m][ni] = (f4_t){0.f, 0.f, 0.f, 0.f};
            // (ky, row) steps with the pixel fragments of step + 1 requested before the MFMAs of step: one block's wave has only the other
            // block's wave on its SIMD to cover an LDS round trip
            {
                h8_t bh[2], bl[2], wlo[2][2];
                auto frag = [&](int step, h8_t& dh, h8_t& dl) __attribute__((always_inline)) {
                    const int ky = step >> 2, ni = step & 3;
                    const int at = ((2 * (wave * 4 + ni) + ky) * kSlots + frow + fg) * 16;
                    dh = *(const h8_t*)(halo + at);
                    if constexpr (X3) dl = *(const h8_t*)(halo + kPlaneBytes + at);
                };
                auto wfrag = [&](int ky, h8_t (&d)[2]) __attribute__((always_inline)) {
                    if constexpr (X3) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) d[m] = *(const h8_t*)(wlo_lds + ((ky * 2 + m) * 64 + lane) * 16);
                    }
                };
                wfrag(0, wlo[0]);
                frag(0, bh[0], bl[0]);
#pragma unroll
                for (int step = 0; step < 28; ++step) {
                    const int ky = step >> 2, ni = step & 3;
                    if (step + 1 < 28) frag(step + 1, bh[(step + 1) & 1], bl[(step + 1) & 1]);
                    if (ni == 0 && ky + 1 < 7) wfrag(ky + 1, wlo[(ky + 1) & 1]);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        if constexpr (X3) {
                            acc[m][ni] = mma(wlo[ky & 1][m], bh[step & 1], acc[m][ni]);
                            acc[m][ni] = mma(whi[ky][m], bl[step & 1], acc[m][ni]);
                        }
                        acc[m][ni] = mma(whi[ky][m], bh[step & 1], acc[m][ni]);
                    }
                    if (ni == 3 && wave == 3 && !last_ty) {       // the extra row 16 (wave-uniform branch): same accumulation order as the tile below uses for its row 0
                        const int at = ((32 + ky) * kSlots + frow + fg) * 16;
                        const h8_t xh = *(const h8_t*)(halo + at);
                        h8_t xl = xh;
                        if constexpr (X3) xl = *(const h8_t*)(halo + kPlaneBytes + at);
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            if constexpr (X3) {
                                accx[m] = mma(wlo[ky & 1][m], xh, accx[m]);
                                accx[m] = mma(whi[ky][m], xl, accx[m]);
                            }
                            accx[m] = mma(whi[ky][m], xh, accx[m]);
                        }
                    }
                }
            }

            __builtin_amdgcn_sched_barrier(0);      // (keeps the epilogue's constant loads out of the MFMA loop's register budget)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // B1': every wave has taken its fragments -- the halo buffer is free
            asm volatile("" ::: "memory");
            {   // next tile's halo: in flight during the epilogue, the pool stage and the stores (single buffer: two blocks per CU)
                int nn = n, nty = ty, ntx = tx + 1, ns = strip;
                if (ntx == tiles_x) { ntx = 0; ns = strip + n_slots * 8; nn = ns / tiles_y; nty = ns - nn * tiles_y; }
                if (ns < n_strips) issue_halo(nn, nty, ntx);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- epilogue, pass 1: f1 = split(acc * scale + shift) stored; the accumulators are overwritten by hi + lo (what the pool reads back)
            {
                float sc[8], sh[8];
                *(float4*)&sc[0] = *(const float4*)(cst + c0); *(float4*)&sc[4] = *(const float4*)(cst + c0 + 4);
                *(float4*)&sh[0] = *(const float4*)(cst + 64 + c0); *(float4*)&sh[4] = *(const float4*)(cst + 64 + c0 + 4);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int oy = ty * 16 + wave * 4 + ni;
                    const size_t pix = ((size_t)n * p.Ho + oy) * p.Wo + tx * 16 + frow;
                    h8_t vh, vl;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float y = __builtin_fmaf(acc[q >> 2][ni][q & 3], sc[q], sh[q]);
                        if (p.relu) y = fmaxf(y, 0.f);
                        if constexpr (X3) {
                            _Float16 a, b;
                            split1(y, a, b);
                            vh[q] = a; vl[q] = b;
                            acc[q >> 2][ni][q & 3] = __fadd_rn((float)a, (float)b);      // exact in fp32
                        } else {
                            vh[q] = (_Float16)__builtin_amdgcn_fmed3f(y, -65504.f, 65504.f);      // (= pack_f16x2: saturate, round to nearest even)
                            acc[q >> 2][ni][q & 3] = (float)vh[q];                       // what maxpool_kernel reads back
                        }
                    }
                    if constexpr (X3) {
                        uint16_t* dst = (uint16_t*)p.out + pix * 128 + half * 64 + fg * 8;      // channel group `half`: [32 hi][32 lo]
                        *(h8_t*)dst = vh;
                        *(h8_t*)(dst + 32) = vl;
                    } else {
                        *(h8_t*)((uint16_t*)p.out + pix * 64 + half * 32 + fg * 8) = vh;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- pass 2: z = ReLU(bn_conv1(hi + lo)) (maxpool_kernel's arithmetic), vertical maxima of the band on the fly:
            // pooled row 2 wave = rows 0..2, pooled row 2 wave + 1 = rows 2, 3 (+ the next band's first row, from stageR)
            {
                float ps[8], pb[8];
                *(float4*)&ps[0] = *(const float4*)(cst + 128 + c0); *(float4*)&ps[4] = *(const float4*)(cst + 128 + c0 + 4);
                *(float4*)&pb[0] = *(const float4*)(cst + 192 + c0); *(float4*)&pb[4] = *(const float4*)(cst + 192 + c0 + 4);
                float v0[8], v1[8];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    float zz[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float v = __builtin_fmaf(acc[q >> 2][ni][q & 3], ps[q], pb[q]);
                        if (p.pool_relu) v = fmaxf(v, 0.f);
                        zz[q] = v;
                    }
                    if (ni == 0) {
                        float* dr = stageR + (wave * 16 + frow) * kColStride + fg * 8;
                        *(float4*)dr = *(const float4*)&zz[0]; *(float4*)(dr + 4) = *(const float4*)&zz[4];
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (ni == 0) v0[q] = zz[q];
                        else if (ni == 1) v0[q] = fmaxf(v0[q], zz[q]);
                        else if (ni == 2) { v0[q] = fmaxf(v0[q], zz[q]); v1[q] = zz[q]; }
                        else v1[q] = fmaxf(v1[q], zz[q]);
                    }
                }
                float* d0 = stageV + ((2 * wave) * 18 + 2 + frow) * kColStride + fg * 8;
                float* d1 = stageV + ((2 * wave + 1) * 18 + 2 + frow) * kColStride + fg * 8;
                *(float4*)d0 = *(const float4*)&v0[0]; *(float4*)(d0 + 4) = *(const float4*)&v0[4];
                *(float4*)d1 = *(const float4*)&v1[0]; *(float4*)(d1 + 4) = *(const float4*)&v1[4];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (wave == 3 && !last_ty) {            // the extra row (never stored: the tile below owns it): only its pool operand
                float sc[8], sh[8], ps[8], pb[8], zx[8];
                *(float4*)&sc[0] = *(const float4*)(cst + c0); *(float4*)&sc[4] = *(const float4*)(cst + c0 + 4);
                *(float4*)&sh[0] = *(const float4*)(cst + 64 + c0); *(float4*)&sh[4] = *(const float4*)(cst + 64 + c0 + 4);
                *(float4*)&ps[0] = *(const float4*)(cst + 128 + c0); *(float4*)&ps[4] = *(const float4*)(cst + 128 + c0 + 4);
                *(float4*)&pb[0] = *(const float4*)(cst + 192 + c0); *(float4*)&pb[4] = *(const float4*)(cst + 192 + c0 + 4);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float y = __builtin_fmaf(accx[q >> 2][q & 3], sc[q], sh[q]);
                    if (p.relu) y = fmaxf(y, 0.f);
                    float back;
                    if constexpr (X3) {
                        _Float16 a, b;
                        split1(y, a, b);
                        back = __fadd_rn((float)a, (float)b);
                    } else {
                        back = (float)(_Float16)__builtin_amdgcn_fmed3f(y, -65504.f, 65504.f);
                    }
                    float v = __builtin_fmaf(back, ps[q], pb[q]);
                    if (p.pool_relu) v = fmaxf(v, 0.f);
                    zx[q] = v;
                }
                float* dr = stageR + (4 * 16 + frow) * kColStride + fg * 8;
                *(float4*)dr = *(const float4*)&zx[0]; *(float4*)(dr + 4) = *(const float4*)&zx[4];
            }
            // B2: stage complete.  A raw barrier behind an LDS-only wait: __syncthreads() would drain the halo DMA in flight (vmcnt(0))
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");


            // ---- pool stage: thread = (pooled row py, window j, channel octet)
            {
                const int oc = tid & 3, j = (tid >> 2) & 7, py = tid >> 5;
                const int PY = 8 * ty + py, PX = 8 * tx - 1 + j;
                const bool odd = py & 1;
                const bool row_ok = PY < p.pool_Ho;
                float m[8];
                auto column = [&](int c, float (&o)[8]) __attribute__((always_inline)) {      // F(py, staged column c >= 2)
                    const float* a = stageV + (py * 18 + c) * kColStride + oc * 8;
                    *(float4*)&o[0] = *(const float4*)a; *(float4*)&o[4] = *(const float4*)(a + 4);
                    if (odd) {
                        const float* r = stageR + (((py + 1) >> 1) * 16 + (c - 2)) * kColStride + oc * 8;
                        float t[8];
                        *(float4*)&t[0] = *(const float4*)r; *(float4*)&t[4] = *(const float4*)(r + 4);
#pragma unroll
                        for (int q = 0; q < 8; ++q) o[q] = fmaxf(o[q], t[q]);
                    }
                };
                auto saved = [&](int c, float (&o)[8]) __attribute__((always_inline)) {       // columns of the previous tile
                    const float* a = save + ((par * 8 + py) * 2 + c) * kColStride + oc * 8;
                    *(float4*)&o[0] = *(const float4*)a; *(float4*)&o[4] = *(const float4*)(a + 4);
                };
                float a0[8], a1[8], a2[8];
                if (j == 0) { saved(0, a0); saved(1, a1); } else { column(2 * j, a0); column(2 * j + 1, a1); }
                column(2 * j + 2, a2);
#pragma unroll
                for (int q = 0; q < 8; ++q) m[q] = fmaxf(fmaxf(a0[q], a1[q]), a2[q]);
                if (j == 7) {                       // columns 14, 15 of this tile = the next tile's saved pair
                    float a3[8];
                    column(17, a3);
                    float* s0 = save + (((par ^ 1) * 8 + py) * 2 + 0) * kColStride + oc * 8;
                    *(float4*)s0 = *(const float4*)&a2[0]; *(float4*)(s0 + 4) = *(const float4*)&a2[4];
                    float* s1 = s0 + kColStride;
                    *(float4*)s1 = *(const float4*)&a3[0]; *(float4*)(s1 + 4) = *(const float4*)&a3[4];
                }
                if (row_ok && PX >= 0 && PX < p.pool_Wo) {
                    h8_t vh, vl;
                    if constexpr (X3) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) { _Float16 a, b; split1(m[q], a, b); vh[q] = a; vl[q] = b; }
                        uint16_t* dst = (uint16_t*)p.pool_out + (((size_t)n * p.pool_Ho + PY) * p.pool_Wo + PX) * 128 + half * 64 + oc * 8;
                        *(h8_t*)dst = vh;
                        *(h8_t*)(dst + 32) = vl;
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) vh[q] = (_Float16)__builtin_amdgcn_fmed3f(m[q], -65504.f, 65504.f);      // (= to_elem<_Float16>)
                        *(h8_t*)((uint16_t*)p.pool_out + (((size_t)n * p.pool_Ho + PY) * p.pool_Wo + PX) * 64 + half * 32 + oc * 8) = vh;
                    }
                }
            }
            par ^= 1;
        }
    }
}

hipError_t launch_stem_pool_x3(const StemParams& p, int num_cus, hipStream_t s)
{
    static bool attr_done[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!attr_done[dev & 63]) {
        e = hipFuncSetAttribute((const void*)stem_pool<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kStemPoolLdsBytes);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)stem_pool<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kStemPoolF16LdsBytes);
        if (e != hipSuccess) return e;
        attr_done[dev & 63] = true;
    }
    const int n_strips = p.n * (p.Ho / 16);
    // two blocks per CU; 16 blocks = 8 strips x 2 channel halves
    int groups = (n_strips + 7) / 8;
    const int max_groups = (2 * num_cus) / 16 > 0 ? (2 * num_cus) / 16 : 1;
    if (groups > max_groups) groups = max_groups;
    if (p.x3) hipLaunchKernelGGL(stem_pool<true>, dim3(groups * 16), dim3(256), kStemPoolLdsBytes, s, p);
    else hipLaunchKernelGGL(stem_pool<false>, dim3(groups * 16), dim3(256), kStemPoolF16LdsBytes, s, p);
    return hipGetLastError();
}

}  // namespace sbbseg
