// internal.h -- structures shared by the host plan (api.hip) and the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sbbseg {

// Every activation buffer starts with a zeroed header; gather lanes that fall into padding (conv
// zero padding, one_side_pad, K padding) read their 16 bytes from offset 0 of the buffer instead
// of branching -- global_load_lds cannot predicate per lane.
constexpr int kZeroHeaderBytes = 256;
constexpr int kBK = 64;               // contraction elements per K-step (8 granules of 8 bf16 = 16 B)
constexpr int kGranulesPerStep = 8;

struct SrcDesc {
    const char* base;     // buffer start (zero header first, data at +kZeroHeaderBytes)
    int PH, PW;           // stored (physical) height / width per patch
    int pix_bytes;        // bytes per stored pixel (channels * elem size)
    int shift;            // nearest upsampling log2 (0|1)
    int lim_y, lim_x;     // PH << shift, PW << shift
    int ksteps;           // K-steps contributed by this source
    int sy_shift, sx_shift; // log2 of this source's input step per output step (stride 1|2); the
                          // source's padding and placement offset are folded into the tap table
    int lo_off;           // byte distance from slot 0-3's granules to slot 4-7's inside a K-step: 64 (the next four
                          // 8-channel granules) in the plain modes; channels * 2 in the split mode (the "lo" plane
                          // of the same 32 channels, see kF16X3)
    uint32_t bytes;       // bytes of the buffer in use (header + the launch's patches): num_records of the fast gather
    int tap_lo_y, tap_lo_x; // fast gather: smallest tap offset of this source over all K-steps / classes (its taps span <= 4 x 4)
};

// one entry per 16-byte granule of the contraction axis
struct KTabEntry {
    int16_t dy, dx;       // tap offset minus the source's placement offset
    int32_t coff;         // byte offset of the granule inside the stored pixel
};

// one record per K-step, read through the scalar cache.  Regular step: all 8 granules share the tap
// (dy,dx) and are channel-consecutive (coff + 16*g).  Irregular: per-granule KTabEntry applies.
struct KStepRec {
    int16_t dy, dx;
    int32_t coff;
    int32_t irregular;
    int32_t pad_;
};

// fast gather (ConvParams::fast_gather): the table handed to the kernel in place of the KStepRec table, same stride.
// A source's buffer resource starts kFgBiasPixels(PW) pixels before the buffer, so `soff` is never negative.
struct FgStepRec {
    uint32_t soff;        // dy * row bytes + dx * pixel bytes + channel byte offset + kFgBiasPixels(PW) * pixel bytes
    int32_t tapbit;       // source * 16 + (dy - tap_lo_y) * 4 + (dx - tap_lo_x): bit of this tap in a row's out-of-bounds mask
    int32_t pad_[2];
};
__host__ __device__ constexpr int kFgBiasPixels(int PW) { return 8 * PW + 8; }   // covers taps down to (-8, -8)

struct ConvParams {
    SrcDesc src[2];
    int n_src;
    const KStepRec* kstep;    // [total_ksteps]
    int variant;              // 0 = auto tile choice, 1 = force 4-wave/2-stage, 2 = force 8-wave/3-stage
    int half_stages;          // A/B: half-K-step LDS stages in a 4-deep ring
    int variant_flags;        // A/B switches: bit 0 = drain the epilogue stores before the next barrier, bit 1 = half-line epilogue stores (old behaviour), bit 2 = 8-phase schedule on the 256x256 tile
    int tile_map;             // 0 = channel tile per XCD (big weights), 1 = pixel tiles grouped per XCD (small weights)
    int persist_blocks;       // CUs of the device (persistent grid = resident blocks); 0 = one block per tile
    const KTabEntry* ktab;    // [total_ksteps * 8]
    const void* w;            // packed [cout_pad][Ktot] (bf16) -- K order = ktab order
    int Ktot;                 // total_ksteps * 64
    int total_ksteps;
    int M;                    // batch * Ho * Wo output pixels of THIS op's output grid
    int Ho, Wo;
    // placement of the op's output grid inside the output tensor(s): tensor[y*osy+ooy][x*osx+oox]
    // (1,1,0,0 = the whole tensor; the parity-split decoder convs write every second pixel)
    int TH, TW, osy, osx, ooy, oox;
    int cout;                 // real output channels (multiple of 8)
    const float* scale;       // [cout_pad]
    const float* shift;
    void* out;                // data pointer (header skipped), [M][cout]; may be null
    const void* residual;     // same shape as out, or null
    void* raw_out;            // or null
    const float* raw_scale;
    const float* raw_shift;
    int relu;
    // output-placement classes sharing all geometry (the four parity classes of a decoder conv run as
    // ONE launch): class q has its own weights / tap tables / placement offset; class 0 = the fields above
    int n_cls;
    int cls_minor;            // tile order: 0 = class-major, 1 = classes of a pixel tile adjacent (needs tile_map 1)
    const void* w_cls[4];
    const KStepRec* kstep_cls[4];
    const KTabEntry* ktab_cls[4];
    int ooy_cls[4], oox_cls[4];
    // fused head (cout == 32 tiles only): 1x1 conv + BN + softmax + argmax on the fp32 epilogue values
    float wmul_cls[4];        // per-class multiplier of `scale` (split mode: 2^-s of the power-of-two weight pre-scale; else 1)
    int head_classes;         // 0 = none
    const float* head_w;      // [cout][classes]
    const float* head_scale;  // [classes]
    const float* head_shift;
    uint8_t* labels;          // [n][TH][TW]
    float* probs;             // [n][TH][TW][classes] or null
    uint32_t howo_magic, howo_shift, wo_magic, wo_shift, nct_magic, nct_shift;   // FastDiv pairs for Ho * Wo, Wo and the number of
                                                           // channel tiles, filled by the launcher (kernels.hip)
    int tile2d, tpr;                  // 1: pixel indices are cut into 16 x 16 blocks (tpr = Wo / 16 blocks per row), see decode_yx in conv_igemm_mfma
    uint32_t tpr_magic, tpr_shift;    //    (filled by the launcher)
    const FgStepRec* fgstep_cls[4];   // fast gather: per-class tables read in place of kstep / kstep_cls (class 0 = entry 0)
    int fast_gather;          // every K-step regular, each source's taps within a 4 x 4 window, no upsampling source, buffers < 2 GiB:
                              // run the FG form of conv_igemm_mfma (kernels.hip)
    // split-K (one-patch launches of the whole-image branch, 16-bit and split modes): 2^ks_shift blocks share a tile's K-steps, each writes its fp32
    // partial sums (x the class's power-of-two weight scale) to ks_ws[split][output pixel][cout]; splitk_finish adds them in split order
    // and runs the epilogue.  0 = off
    int ks_shift;
    float* ks_ws;
    long ks_split_elems;      // floats per split = output pixels of the tensor x cout
    // owned-region launch (region.h): the output grid is not walked whole -- pixel index m of class q is the rmap[q * M + m]-th entry of a
    // per-launch table of (patch, oy, ox) triples (kRegionCode) that lists, patch by patch, only the class-grid pixels the page stitch
    // will keep plus the halo the later decoder levels need; M = entries per class.  null = the whole grid (m = (n * Ho + oy) * Wo + ox)
    const uint32_t* rmap;
};

// fused network tail: 3x3 conv over [nearest-x2-upsampled src0 (64 ch), image C8 (3 ch)] -> 32 ch
// -> BN/ReLU -> 1x1 head -> softmax -> argmax, one launch, nothing but labels (and optional
// probabilities) written.  See dec_tail_fused in kernels.hip.
struct TailParams {
    const char* src0;         // buffer start (zero header), [n][PH][PW][64] 16-bit
    const char* img;          // buffer start (zero header), C8 form [n][2PH][2PW][8] (split mode: hi slots 4..6 repeat lo 0..2)
    int PH, PW;               // src0 size; output is 2PH x 2PW
    int n;                    // patches
    const void* wfrag;        // [4 parities][6 K-steps][2 kk][2 mi][64 lanes] x 16 bytes, MFMA A-fragment order (split mode: per parity
                              // [hi | lo][5 K-steps][2 kk][2 mi][64 lanes], the image taps packed two to a k-group: sbbseg_add_tail)
    const float* scale;       // [32]
    const float* shift;
    int classes;              // <= 4
    const float* head_w;      // [32][classes]
    const float* head_scale;
    const float* head_shift;
    uint8_t* labels;          // [n][2PH][2PW]
    float* probs;             // [n][2PH][2PW][classes] or null
    // owned-region launch (region.h): the launch walks `n_tab` 16 x 16 output tiles listed in ttab (kRegionCode: patch, tile origin / 2)
    // instead of every tile of every patch; null = all tiles
    const uint32_t* ttab = nullptr;
    int n_tab = 0;
};

// Network stem: 7x7 stride-2 conv on the image in the PAIRS input form (7 rows x 4 two-pixel granules
// of 8 values) -> 64 channels, per-channel affine (+ ReLU), 16-bit NHWC.  See stem_conv_pairs.
struct StemParams {
    const char* pairs;        // buffer start (zero header), [n][PHt][PWt][8] 16-bit
    int PHt, PWt;             // padded rows, two-pixel granules per row
    int n;                    // patches
    int Ho, Wo;               // output size, multiples of 16
    const void* wfrag;        // [7 ky][4 mi][64 lanes] x 16 bytes, MFMA A-fragment order, rows = conv_row_channel
                              // (split mode: that block twice -- hi fragments, then lo fragments of the pre-scaled weights)
    const float* scale;       // [64]
    const float* shift;
    int relu;
    float wmul;               // multiplier of `scale` (split mode: 2^-s of the weight pre-scale; else 1)
    void* out;                // data pointer [n][Ho][Wo][64]
    // stem_pool_x3 (stem_pool_x3.hip): the 3x3 / stride-2 / valid max-pool over ReLU(pool_scale * f1 + pool_shift) written by the same launch
    void* pool_out = nullptr; // data pointer [n][pool_Ho][pool_Wo][64], or null (plain stem)
    const float* pool_scale = nullptr;      // [64] the affine maxpool_kernel applies to every tap (bn_conv1)
    const float* pool_shift = nullptr;
    int pool_relu = 0;
    int pool_Ho = 0, pool_Wo = 0;
    int x3 = 1;               // stem_pool: 1 = split mode (hi | lo planes), 0 = plain fp16
};

// 3x3 stride-1 'same' conv, 64 -> 64 channels (the ResNet stage-2 bottleneck convs), as a direct conv on an
// LDS halo tile with the weights resident in registers.  See conv3x3_c64_direct.
struct Direct64Params {
    const char* src;          // buffer start (zero header), [n][H][W][64] 16-bit
    int n, H, W;
    const void* wfrag;        // [9 taps][2 kk][4 mi][64 lanes] x 16 bytes, MFMA A-fragment order, rows = conv_row_channel;
                              // split mode: that block twice (hi fragments, then lo fragments of the pre-scaled weights)
    const float* scale;       // [64]
    const float* shift;
    int relu;
    float wmul;               // multiplier of `scale` (split mode: 2^-s of the power-of-two weight pre-scale; else 1)
    void* out;                // data pointer [n][H][W][64] (split mode: [64 hi][64 lo] per pixel)
};

// Split mode (kF16X3) pixel layout (round 3): a stored pixel of C channels is a sequence of channel GROUPS of G = min(C, 32)
// channels, each group = [G hi halves][G lo halves] -- for C >= 32 one 128-byte line per group, which is exactly what a split K-step
// (32 channels of one tap, hi and lo) reads.  (Round 2 stored [C hi][C lo]: a K-step then touched two half-used lines per pixel,
// doubling the L2 footprint of every gather and fetching every line twice -- PMC: 2-6x the unique bytes on the decoder launches.)
#if defined(__HIPCC__)
#define SBBSEG_HD __host__ __device__
#else
#define SBBSEG_HD
#endif
SBBSEG_HD inline int split_group(int C) { return C < 32 ? C : 32; }                      // channels per group = distance (in halves) hi -> lo
SBBSEG_HD inline int split_hi_elem(int C, int ch)                                         // index (in halves) of channel ch's hi half inside the pixel
{
    const int G = split_group(C);
    return (ch / G) * 2 * G + (ch % G);
}

// One ResNet bottleneck block at 64 internal channels (1x1 -> 3x3 -> 1x1 + shortcut) as ONE launch.  See bottleneck_fused.
struct BlockParams {
    const char* x;            // buffer start (zero header), [n][H][W][CIN] 16-bit; CIN = 256 (identity) | 64 (projection)
    int n, H, W;
    int proj;                 // 0: y = ReLU(BN(W3 b) + x);  1: y = ReLU(BN(W3 [b, x]))  (shortcut conv folded by the planner)
    int pq;                   // 1: bottleneck_fused_pq (producer / consumer wave groups), 0: bottleneck_fused
    const void* w1;           // [CIN/32 kk][4 mi][64 lanes] x 16 bytes, MFMA A-fragment order, rows = conv_row_channel
    const void* w2;           // [9 taps][2 kk][4 mi][64 lanes] x 16 bytes (Direct64Params::wfrag)
    const void* w3;           // [2|4 kk][16 mi][64 lanes] x 16 bytes; projection: kk 0-1 contract b, kk 2-3 contract x
    const float *s1, *b1;     // [64]  scale / shift after the first 1x1
    const float *s2, *b2;     // [64]  ... after the 3x3
    const float *s3, *b3;     // [256] ... after the last 1x1 (before the residual add)
    void* out;                // data pointer [n][H][W][256]
    // split mode (block_x3_identity, block_x3.hip): w1 = [8 kk][4 mi][hi | lo][64 lanes] x 16 B, w3 = [2 kk][16 mi][hi | lo][64 lanes] x 16 B,
    // w2 = [hi | lo][9][2][4 mi][64 lanes] x 16 B, every conv's weights pre-scaled by a power of two; wmulN = 2^-s undoes it in sN
    float wmul1 = 1.f, wmul2 = 1.f, wmul3 = 1.f;
    int dbg = 0;              // timing probes (SBBSEG_BLOCK_DBG; results are WRONG with any bit set): 1 = LDS rows rotated by 1 slot per row on the WRITE side only
};

// The decoder conv at 224 x 224 in the split mode with LDS-resident source halos (dec_halo_x3.hip): the grouped launch of the four
// output-parity classes of  conv3x3([up2(src0: 128 ch), skip: 64 ch]) -> 64 ch  as ONE direct kernel on 16 x 16 output tiles.
struct DecHaloParams {
    const char* src0;         // buffer start (zero header), [n][PH][PW][128] split layout (512 B per pixel)
    const char* skip;         // buffer start (zero header), [n][2 PH][2 PW][64] split layout (256 B per pixel)
    int PH, PW, n;
    const void* wfrag;        // [4 classes][34 K-steps][4 row blocks][hi | lo][64 lanes] x 16 B: MFMA A fragments of the classes' packed weights
    const int* taps;          // [4 classes][16]: the class's 4 src0 taps, then its 9 skip taps, in K-step order: (dy & 255) | (dx & 255) << 8
    const float* scale;       // [64]
    const float* shift;
    float wmul[4];            // per class: 2^-s of its power-of-two weight pre-scale
    int relu;
    void* out;                // data pointer [n][2 PH][2 PW][64] split layout
    const uint32_t* ttab = nullptr;      // owned-region launch: see TailParams
    int n_tab = 0;
};

// Split mode, stages 3 / 4 of the encoder: the last 1x1 conv of an identity bottleneck block (C -> 4C, BN, + residual, ReLU) and the first
// 1x1 conv of the next block (4C -> C, BN, ReLU) in one launch (expand_reduce_x3.hip): y is written once and contracted from LDS.
struct ExpRedParams {
    const char* b;            // buffer start (zero header), [M][C] split layout: the expand's input
    const char* x;            // buffer start (zero header), [M][4C]: the residual
    char* y;                  // buffer start (zero header), [M][4C]: the expand's output
    char* a2;                 // buffer start (zero header), [M][C]: the reduce's output
    int M;                    // pixels (patches x H x W)
    int C;                    // 128 | 256
    int x3;                   // 1: split mode (hi | lo planes); 0: plain fp16 (the fragment pairs are the two k-halves of a 64-channel K-step)
    const void* w3frag;       // [4C / 256 chunks][C / 32 K-steps][8 waves][2 row blocks][hi | lo][64 lanes] x 16 B: A fragments of the expand's packed rows
    const void* w1frag;       // [4C / 256 chunks][8 K-steps][8 waves][C / 128 row blocks][hi | lo][64 lanes] x 16 B: ... of the reduce's
    const float *s3, *h3;     // [4C] scale / shift of the expand
    const float *s1, *h1;     // [C]  ... of the reduce
    float wmul3, wmul1;       // 2^-s of the power-of-two weight pre-scales
    int dbg;                  // timing probes (SBBSEG_ER_DBG; results are WRONG with any bit set): 1 = one weight block, 2 = y stores dropped, 4 = x reads dropped
};

// Encoder stage 3: an identity block's 3x3 conv + its last 1x1 conv + the next block's first 1x1 conv in one launch
// (conv3_expand_reduce.hip): b = ReLU(BN(W2 * a)) lives in LDS only.
struct C3ERParams {
    const char* a;            // buffer start (zero header), [n][H][W][C]: the 3x3 conv's input
    const char* x;            // buffer start (zero header), [n][H][W][4C]: the residual
    char* y;                  // buffer start (zero header), [n][H][W][4C]
    char* a2;                 // buffer start (zero header), [n][H][W][C]: the reduce's output
    int n, H, W;              // H, W multiples of 8
    int C;                    // 128
    int x3;                   // 1: split mode; 0: plain fp16
    const void* w2frag;       // [9 C / KCH K-steps][8 waves][C / 128 row blocks][hi | lo][64 lanes] x 16 B: A fragments of the 3x3 conv's packed rows, its K order
    const void* w3frag;       // as ExpRedParams
    const void* w1frag;
    const float *s2, *h2;     // [C] scale / shift of the 3x3 conv
    const float *s3, *h3;     // [4C]
    const float *s1, *h1;     // [C]
    float wmul2, wmul3, wmul1;
    const int* k0;            // device [9 C / KCH]: the 3x3 conv's K-steps in its own order: (dy & 255) | (dx & 255) << 8 | channel group << 16
};

struct HeadParams {
    const void* src;          // [M][cin] activations (data pointer)
    int cin;                  // <= 64, multiple of 8
    int classes;              // <= 8
    int M;
    const float* w;           // [cin][classes]
    const float* scale;       // [classes]
    const float* shift;
    uint8_t* labels;          // [M]
    float* probs;             // [M][classes] or null
};

struct IngestParams {
    const uint8_t* page;      // [src_Hp][src_Wp][3] u8 (device)
    int Hp, Wp;               // size of the (virtual) page the tile grid lives on
    int src_Hp, src_Wp;       // size of the stored page (== Hp, Wp unless a nearest-resize map is given)
    int whole;                // 1: one patch = the whole page resized to the model input (maps indexed by model coords)
    const int* tile_xy;       // device [n][2] explicit origins, or null = closed-form grid below
    int grid_first, grid_nyf; // grid mode: tile t = grid_first + local index; i = t / nyf, j = t % nyf
    int grid_mid_x, grid_mid_y; //   origin = min(i*mid_x, Wp-W), min(j*mid_y, Hp-H)   (main.py:262-281)
    int n_tiles;
    int H, W;                 // model input size
    const float* lut;         // [256] float32(v / 255.0)
    void* c8;                 // data pointer of the C8 form  [n][H][W][8]
    void* pairs;              // data pointer of the PAIRS form [n][H+2p][PWp][8]; may be null
    int pad, pairs_w;         // p, ceil((W+2p)/2)
    const int* map_y;         // optional nearest-resize gather tables (whole-image branch) or null
    const int* map_x;
    const int* bin_thr;       // device int: binarise channel 0 at this (Otsu) threshold into all 3 channels, or null
};

// kF16X3: error-compensated fp16 ("split") mode.  Every activation and weight v is carried as two fp16 numbers
// hi = fp16(v), lo = fp16(v - hi) (about 22 significant bits together); a product is three MFMAs
// (hi*hi + hi*lo + lo*hi, fp32 accumulate), the dropped lo*lo term is ~2^-22 relative.  Activations are stored
// per pixel as [C hi][C lo] (4 bytes per element), weights per 32-channel K-step as [32 hi][32 lo] after a
// power-of-two pre-scale that keeps their lo parts out of the fp16 subnormal range.
enum Precision { kBF16 = 0, kF32 = 1, kF16 = 2, kF16X3 = 3 };

// Timing probes that make a kernel compute WRONG results on purpose (every pixel gather an L2 hit, one weight block, stores dropped ...)
// exist only in probe builds (`python -m sbb_textline_detection_amd._build --probes`: -DSBBSEG_PROBES, a separate library under
// tools/probes/bin/); in the shipped library the conditions below are the constant `false` and the environment variables that used to
// switch them (SBBSEG_CONV_PROBE_LOCAL / _WHOT, SBBSEG_BLOCK_DBG, SBBSEG_ER_DBG) are not read at all.
#ifdef SBBSEG_PROBES
#define SBBSEG_PROBE(cond) (cond)
#else
#define SBBSEG_PROBE(cond) (false)
#endif
inline bool is_split(int precision) { return precision == kF16X3; }

// launchers implemented in kernels.hip ---------------------------------------------------------
hipError_t launch_conv(const ConvParams& p, int precision, hipStream_t s);
hipError_t launch_splitk_finish(const float* ws, int splits, long split_elems, long pixels, int cout, const float* scale, const float* shift,
                                const void* residual, int relu, void* out, int precision, hipStream_t s);
hipError_t launch_maxpool(const void* src, void* dst, int n, int H, int W, int C, int k, int stride,
                          int Ho, int Wo, const float* pre_scale, const float* pre_shift, int pre_relu,
                          int precision, hipStream_t s);
hipError_t launch_head(const HeadParams& p, int precision, hipStream_t s);
hipError_t launch_tail(const TailParams& p, int precision, int num_cus, hipStream_t s);
hipError_t launch_stem(const StemParams& p, int precision, int num_cus, hipStream_t s);
hipError_t launch_dec_halo_x3(const DecHaloParams& p, int num_cus, hipStream_t s);     // split mode: dec4 with LDS-resident halos (dec_halo_x3.hip)
hipError_t launch_dec_halo_f16(const DecHaloParams& p, int num_cus, hipStream_t s);
hipError_t launch_conv3_expand_reduce(const C3ERParams& p, int num_cus, hipStream_t s);   // 3x3 + expand + next reduce (conv3_expand_reduce.hip)
hipError_t launch_expand_reduce_x3(const ExpRedParams& p, int num_cus, hipStream_t s);   // split mode: expand + next reduce 1x1 (expand_reduce_x3.hip)    // ... plain fp16 mode (dec_halo_f16.hip)
hipError_t launch_stem_pool_x3(const StemParams& p, int num_cus, hipStream_t s);      // split mode: stem + max-pool in one launch (stem_pool_x3.hip)
hipError_t launch_direct64(const Direct64Params& p, int precision, int num_cus, hipStream_t s);
constexpr int kTailKSteps = 6;      // 4 taps x 64 channels of src0 + 2 steps for the 9 image taps
hipError_t launch_ingest_u8(const IngestParams& p, int precision, hipStream_t s);
hipError_t launch_otsu(const uint8_t* page, int src_Wp, int Hp, int Wp, const int* map_y, const int* map_x,
                       unsigned* hist, int* thr, int num_cus, hipStream_t s);
hipError_t launch_ingest_f32(const float* x, int n, int H, int W, void* c8, void* pairs, int pad,
                             int pairs_w, int precision, hipStream_t s);
hipError_t launch_stitch(const uint8_t* tile_labels, int H, int W, const int* own_x, const int* own_y,
                         int nyf, int Hp, int Wp, uint8_t* out, hipStream_t s);
hipError_t launch_resize_labels(const uint8_t* labels, int H, int W, const int* map_y, const int* map_x,
                                int out_h, int out_w, uint8_t* out, hipStream_t s);
// stage glue: iterated 5x5 erode / dilate as one separable clipped min / max filter; largest 8-connected component
hipError_t launch_morph(const uint8_t* src, uint8_t* tmp, uint8_t* dst, int H, int W, int radius, int is_max, int binarize, hipStream_t s);
constexpr int kCcMaxRivals = 250;    // rival roots sbbseg_page_box_dev gets back from the device (more: the host scans the label plane)
hipError_t launch_largest_contour(const uint8_t* mask, int H, int W, int* parent, int* count, int* area2, int* bx0, int* by0, int* bx1,
                                  int* by1, unsigned long long* d_best, int* d_out, hipStream_t s);
hipError_t launch_replicate3(const uint8_t* src, uint8_t* dst, size_t n, hipStream_t s);
hipError_t launch_to_f32(const void* src, float* dst, size_t n, int precision, hipStream_t s);
hipError_t launch_deskew_profiles(const uint8_t* mask, int H, int W, int S, int top, int left, const double* minv, const float* cubic,
                                  int n_angles, int* counts, hipStream_t s);
hipError_t launch_bottleneck(const BlockParams& p, int precision, int num_cus, hipStream_t s);
hipError_t launch_block_x3(const BlockParams& p, int num_cus, hipStream_t s);      // split mode, identity blocks (block_x3.hip)
hipError_t launch_split_to_f32(const void* src, float* dst, size_t npix, int C, hipStream_t s);   // [pix][C hi][C lo] -> [pix][C]

int conv_row_channel(int row, int cout);   // packed weight row -> output channel (16-bit modes)
int conv_tile_bc(int cout);   // channel-tile width the bf16 conv kernel uses for `cout` (weights are padded to it)
int set_error(const char* fmt, ...);   // fills sbbseg_last_error() (thread-local), returns 1
uint16_t f32_to_bf16_rne(float f);
uint16_t f32_to_f16_rne(float f);
float bf16_to_f32(uint16_t h);

}  // namespace sbbseg
