/* sbbseg.h -- C ABI of libsbbseg.so: MI355X (gfx950) patch-wise pixel-segmentation inference.
 *
 * This is the drop-in boundary for ONE path of qurator-spk/sbb_textline_detection:
 * `textline_detector.do_prediction()` and the `model.predict()` it calls per 448x448 patch
 * (reference: qurator/sbb_textline_detector/main.py:225-380).  The reference is Python; its
 * "FFI" for this path is the Keras model duck type.  A maintainer binds these entry points
 * with ctypes (the shipped binding is sbb_textline_detection_amd/_capi.py).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message is in
 *     sbbseg_last_error() (thread-local).  Nothing aborts: the reference's callers rely on
 *     ordinary Python exceptions (main.py:2061-2157), which the Python shim raises from this.
 *   - plain pointers and sizes only; host buffers are caller-owned; the library owns only device
 *     memory behind the opaque handle.  A handle is not thread-safe; distinct handles may be used
 *     from distinct threads / processes (one process per GPU).
 *   - "*_dev" variants take device pointers (HBM-resident inputs/outputs, used by bench.py and
 *     the multi-GPU path); all work is enqueued on the handle's stream (sbbseg_set_stream).
 */
#ifndef SBBSEG_H
#define SBBSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBBSEG_ABI_VERSION 5

typedef struct sbbseg_ctx sbbseg_ctx;

/* arithmetic mode of a handle */
#define SBBSEG_PREC_BF16 0   /* bf16 operands, fp32 MFMA accumulate, fp32 epilogue.  Fast, coarse: 8-bit significands
                                through ~60 layers move near-tie labels (kept for A/B only) */
#define SBBSEG_PREC_F32  1   /* fp32 everything, plain FMA kernels: slow, used to separate plumbing
                                errors from 16-bit rounding in the parity tests */
#define SBBSEG_PREC_F16  2   /* fp16 operands (11-bit significand, saturating stores), same MFMA rate as
                                bf16, fp32 accumulate/epilogue: 8x finer rounding, range +-65504.  The FAST mode:
                                labels can differ from an fp32 run where the top-2 softmax margin is small */
#define SBBSEG_PREC_F16X3 3  /* error-compensated fp16 ("split") mode, the LABEL-EXACT mode and the default of the
                                Python seams: every activation and weight is carried as hi + lo fp16 halves (~22
                                significant bits), each product is three MFMAs (hi*hi + hi*lo + lo*hi) accumulated
                                in fp32.  Same MFMA kernels, ~1/3 of the fp16 mode's arithmetic rate; label maps
                                equal an fp32 evaluation except at exact ties (main.py:290 "argmax bit-exact") */

/* ---- lifecycle: replaces start_new_session_and_model / session.close (main.py:216-223, 428) */
const char* sbbseg_last_error(void);
int sbbseg_abi_version(void);
int sbbseg_device_count(int* count);
int sbbseg_create(int device, int precision, sbbseg_ctx** out);
int sbbseg_destroy(sbbseg_ctx* c);                       /* frees all device memory; NULL ok */
/* ONE-CALL model load: what keras.models.load_model(path, compile=False) + the first predict do in the reference
 * (main.py:216-223).  Takes the .sbbw container (magic "SBBW0001", JSON header = the Keras-2.3 model_config of the .h5 plus a
 * tensor table, little-endian fp32 weights; written offline by tools/h5_to_sbbw.py), reads the layer graph, lowers it to the fused
 * plan (BN folding, parity split of the decoder convs, shortcut merge, fused head / tail -- the planner of planner.py restated in
 * C++, same algebra in the same precision) and returns a finalized handle: no Python needed on the consumer's side.
 * flags: SBBSEG_LOAD_* switches (0 = all lowerings on).  The step-by-step plan API below stays available. */
#define SBBSEG_LOAD_NO_PARITY_SPLIT   1
#define SBBSEG_LOAD_NO_SHORTCUT_MERGE 2
#define SBBSEG_LOAD_NO_FUSED_HEAD     4
#define SBBSEG_LOAD_NO_FUSED_TAIL     8
int sbbseg_model_load(const void* sbbw_bytes, size_t n_bytes, int device, int precision, int max_batch, int flags, sbbseg_ctx** out);
int sbbseg_model_load_file(const char* sbbw_path, int device, int precision, int max_batch, int flags, sbbseg_ctx** out);
/* Test hook (no GPU needed): the plan sbbseg_model_load would build, as text -- one line per tensor / step with the CRC32 of every
 * weight plane, scale and shift vector -- so the library's planner can be compared with planner.py value for value. */
int sbbseg_debug_plan_summary(const void* sbbw_bytes, size_t n_bytes, int precision, int flags, char* out, size_t capacity, size_t* needed);
/* run on the caller's HIP stream (NULL = the legacy default stream, e.g. torch's current stream when
 * no stream context is active); SBBSEG_OWN_STREAM returns to the handle's private non-blocking stream */
#define SBBSEG_OWN_STREAM ((void*)(intptr_t)-1)
int sbbseg_set_stream(sbbseg_ctx* c, void* hip_stream);
int sbbseg_synchronize(sbbseg_ctx* c);
/* Lanes (default 2): the grid-form page entry points split every chunk of >= 16 tiles into two halves
 * that run concurrently -- the second on a private stream with its own activation buffers -- so the
 * launch tails of one half are filled by the other's kernels (measured +6-7 % on a 70-tile page).
 * All work stays ordered on the handle's stream (fork/join events); results do not depend on it.
 * Call before sbbseg_finalize to skip the second set of buffers (lanes = 1), or any time to switch. */
int sbbseg_set_lanes(sbbseg_ctx* c, int lanes);
/* Layout of the HOST label outputs of sbbseg_segment_page / _scaled / _otsu / _whole: 1 (default) = one
 * uint8 plane [H][W]; 3 = the reference's own return layout, uint8 [H][W][3] with three identical channels
 * (main.py:366, 380) -- replicated on the device, the host buffer must hold 3 x H x W bytes. */
int sbbseg_set_label_channels(sbbseg_ctx* c, int channels);
/* Duplicate clamped tiles (default on; SBBSEG_DEDUPE=0 or on = 0 switches it off): when (extent % mid) lies in (0, tile - mid]
 * -- mid = tile - 2 * margin = 360 for the 448-pixel models, so 88 of every 360 page extents -- the inward clamp of main.py:276-281
 * gives the LAST TWO tiles of that axis the same origin: the reference runs the same forward twice and pastes the same labels
 * twice (a 448 x 448 page: four identical forwards).  The fused page entry points (sbbseg_segment_page[_dev / _scaled / _otsu],
 * _segment_pages[_dev], _segment_crop[_dev]) compute each distinct tile once; the label map is the same byte for byte.  The
 * tile-indexed entry points (sbbseg_tile_grid, _segment_tiles_dev, _segment_tile_range[_bin]_dev, _stitch_dev -- the multi-rank
 * protocol) always keep the reference's call list. */
int sbbseg_set_dedupe(sbbseg_ctx* c, int on);
/* Split-K in the whole-image branch (sbbseg_segment_whole[_scaled], sbbseg_extract_page_box[_dev]; every mode but fp32; default on, SBBSEG_KSPLIT=0 or
 * on = 0 switches it off): one patch through a long-K conv occupies 2-32 CUs for 100-400 K-steps, so the K range of such a launch is
 * split over up to 16 blocks per tile and the fp32 partial sums are added in split order by a second launch -- 3.4 -> 1.9 ms per
 * whole-image forward of the 448 model in the split mode, 2.1 -> 1.5 ms in plain fp16.  Deterministic, but the last bits differ from the unsplit launches (summation order): the
 * patch paths and sbbseg_predict never split, their results do not depend on the batch size. */
int sbbseg_set_ksplit(sbbseg_ctx* c, int on);
/* Owned-region launches of the decoder (round 6; default mode 1, SBBSEG_OWNED_REGIONS=0|1|2 overrides the default at creation).
 * do_prediction pastes only part of every tile's label map into the page: the 10 % margin is cropped on every side that is not a page
 * edge (main.py:294-364) and where the inward-clamped last tile of an axis overlaps its neighbour the later tile wins (main.py:276-281
 * + the paste order) -- a 3500 x 2500 page keeps 8.75 of the 14.05 Mpx its 70 tiles produce.  The decoder is local (3x3 convs over
 * nearest-x2 upsamplings, pointwise epilogues), so each decoder level is launched only over the rows / columns the kept pixels depend
 * on: the owned rectangle dilated by one pixel per 3x3 conv above the level and halved per upsampling.  The encoder (receptive field
 * = the tile) runs whole.  Every computed pixel goes through the arithmetic of the full launch: the stitched label map is the same
 * byte for byte.
 *   mode 0: off.  mode 1: the fused page entry points (sbbseg_segment_page[_dev / _scaled / _otsu], _segment_pages[_dev],
 *   _segment_crop[_dev], sbbseg_run_page).  mode 2: also sbbseg_segment_tile_range[_bin]_dev -- the tile labels they return are then
 *   DEFINED ONLY ON EACH TILE'S OWNED REGION (everything sbbseg_stitch_dev reads); a repeated clamped tile owns nothing.
 * sbbseg_predict, sbbseg_segment_tiles_dev and the whole-image branch always compute whole patches. */
int sbbseg_set_owned_regions(sbbseg_ctx* c, int mode);
/* the mode in force, and the decoder levels (the fused tail + the parity-split decoder convs below it) the handle's plan runs as
 * owned-region launches; 0 levels = none (fp32 handles, unfused heads, decoders of another shape: everything is computed whole) */
int sbbseg_owned_region_info(sbbseg_ctx* c, int* mode, int* levels);

/* ---- plan building: the host-side planner (planner.py) lowers the Keras model_config that the
 * reference would have handed to keras.models.load_model (main.py:221) into these calls, in
 * execution order.  Tensors are per-patch NHWC activations; the batch dimension is runtime. */

/* Forms of the network input the kernels consume (both are filled by the ingest kernels):
 *   SBBSEG_INPUT_C8     [H][W][8]            pixel (y,x) channel c at [y][x][c], channels 3..7 zero
 *   SBBSEG_INPUT_PAIRS  [H+2p][ceil((W+2p)/2)][8]  zero-padded by p on every side, two horizontal
 *                       neighbours x 4 channels per 16-byte granule: [y+p][(x+p)>>1][((x+p)&1)*4+c]
 *                       (lets the 7x7 stride-2 stem run as a 7x4 stride-(2,1) conv over 8 channels) */
#define SBBSEG_INPUT_C8    0
#define SBBSEG_INPUT_PAIRS 1
int sbbseg_set_input(sbbseg_ctx* c, int H, int W, int channels);
int sbbseg_input_form(sbbseg_ctx* c, int form, int pad, int* tensor_id);
int sbbseg_add_tensor(sbbseg_ctx* c, int H, int W, int C, int* tensor_id);

typedef struct {
    int32_t tensor;      /* source tensor id */
    int32_t channels;    /* channels taken from it (from channel 0), contraction order */
    int32_t kh, kw;      /* taps of THIS source */
    int32_t stride_y, stride_x;   /* input step per output step, 1 or 2 (in the source's logical coordinates) */
    int32_t pad_top, pad_left;    /* logical input row = oy*stride_y - pad_top + ky  (zero outside) */
    int32_t up_shift;    /* 0, or 1 = nearest-neighbour x2 upsampling (UpSampling2D) fused into the gather */
    int32_t off_y;       /* placement offset of the stored tensor inside the logical one: */
    int32_t off_x;       /*   logical[y][x] = stored[y-off_y][x-off_x] (zero outside); one_side_pad = (1,1) */
} sbbseg_conv_src;

typedef struct {
    int32_t n_src;               /* 1 or 2.  2 = channel concat [src0, src1] (Concatenate fused); the two
                                    sources may have different taps/strides: the planner rewrites a 3x3 conv
                                    over [UpSampling2D(x), skip] as four output-parity classes, each a 2x2 conv
                                    over x at its own resolution (taps that hit the same source pixel are
                                    pre-summed) plus the 3x3 stride-2 conv over the skip */
    sbbseg_conv_src src[2];
    int32_t cout;
    int32_t out_h, out_w;        /* output grid of this op */
    int32_t out_stride_y, out_stride_x, out_off_y, out_off_x;
                                 /* placement in the output tensor(s): tensor[y*stride+off] = result[y] */
    int32_t out_tensor;          /* y = act(scale*conv + shift [+ residual]);  -1 = none */
    int32_t relu;
    int32_t residual_tensor;     /* -1 = none (Add fused into the epilogue) */
    int32_t raw_out_tensor;      /* -1 = none; second output raw_scale*conv + raw_shift, no activation
                                    (the stem's pre-BN skip f1) */
    int32_t head_classes;        /* > 0: fuse the network head (1x1 conv + BN + softmax + argmax, main.py:290)
                                    into this conv's epilogue (needs cout == 32, <= 4 classes, 16-bit mode);
                                    labels/probabilities are the plan's outputs, out_tensor may be -1 */
    double algorithmic_macs;     /* MACs per patch of the reference's formulation of this op (for reporting);
                                    0 = derive from the geometry given here */
} sbbseg_conv_desc;

/* w_src0 / w_src1: float32 [kh][kw][channels][cout] per source (Keras kernel layout); scale/shift:
 * float32 [cout] (BatchNorm and bias folded by the caller); raw_*: only if raw_out_tensor >= 0;
 * head_w [cout][classes], head_scale/head_shift [classes]: only if head_classes > 0. */
int sbbseg_add_conv(sbbseg_ctx* c, const sbbseg_conv_desc* d, const float* w_src0, const float* w_src1,
                    const float* scale, const float* shift,
                    const float* raw_scale, const float* raw_shift,
                    const float* head_w, const float* head_scale, const float* head_shift);
/* k x k max-pool, 'valid'.  pre_scale/pre_shift [C] (or NULL) + pre_relu: per-channel affine and ReLU
 * applied to every input element before the max (a BatchNormalization + relu fused into the pool). */
int sbbseg_add_maxpool(sbbseg_ctx* c, int src_tensor, int dst_tensor, int k, int stride,
                       const float* pre_scale, const float* pre_shift, int pre_relu);
/* Fused network tail (16-bit modes): ReLU(BN(conv3x3 'same' over [UpSampling2D(2)(src0: 64 channels),
 * network input (3 channels, C8 form)])) -> 32 channels -> 1x1 conv + BN + softmax + argmax, in one
 * launch that writes only labels (and probabilities on request).  w_src0 [3][3][64][32], w_img
 * [3][3][3][32] (Keras layout, original 3x3 taps: the library pre-sums them per output parity);
 * scale/shift [32]; head_w [32][classes], head_scale/head_shift [classes], classes <= 4. */
int sbbseg_add_tail(sbbseg_ctx* c, int src0_tensor, int img_c8_tensor, const float* w_src0, const float* w_img,
                    const float* scale, const float* shift, int classes,
                    const float* head_w, const float* head_scale, const float* head_shift,
                    double algorithmic_macs);
/* Final 1x1 conv + BN + softmax + argmax (main.py:290) over src_tensor's channels:
 * w [cin][classes], scale/shift [classes].  Produces u8 labels and (on request) f32 probabilities. */
int sbbseg_add_head(sbbseg_ctx* c, int src_tensor, int cin, int classes,
                    const float* w, const float* scale, const float* shift);
int sbbseg_finalize(sbbseg_ctx* c, int max_batch);

/* ---- queries */
int sbbseg_model_info(sbbseg_ctx* c, int* H, int* W, int* classes, int* max_batch);
int sbbseg_num_ops(sbbseg_ctx* c, int* n);
int sbbseg_op_info(sbbseg_ctx* c, int op, char* name, int name_len, double* flops_per_patch,
                   double* min_bytes_per_patch);
/* MFMA FLOPs op `op` really issues per patch (contraction axis padded to whole K-steps, parity-split decoder convs
 * with pre-summed taps, three MFMAs per product in the split mode): the numerator of bench.py's `frac_issued`;
 * sbbseg_op_info's flops are the ALGORITHMIC ones (reference formulation). */
int sbbseg_op_issued_flops(sbbseg_ctx* c, int op, double* issued_flops_per_patch);
int sbbseg_device_bytes(sbbseg_ctx* c, size_t* bytes);

/* ---- seam 2: model.predict (main.py:287-288, 373-374).
 * x: host float32 [n][H][W][3] already scaled to [0,1]; probs: host float32 [n][H][W][classes]. */
int sbbseg_predict(sbbseg_ctx* c, const float* x_nhwc, int n, float* probs_nhwc);

/* ---- seam 1, patches=True (main.py:231-366), fused: u8/255 LUT normalise, tile with 10 % margin,
 * forward, argmax, margin-crop + last-writer-wins stitch.  page: uint8 [Hp][Wp][3];
 * labels: uint8 [Hp][Wp] (the reference returns this plane replicated x3). */
int sbbseg_segment_page(sbbseg_ctx* c, const uint8_t* page_hwc, int Hp, int Wp, uint8_t* labels_hw);
/* do_prediction(patches=True) for n_pages HOST pages of one size (uint8 [Hp][Wp][3] each; pages_hwc / labels_hw are arrays of
 * n_pages pointers; labels as in sbbseg_segment_page, x3 with sbbseg_set_label_channels(3)).  Pipelined in groups of as many
 * pages as fill a chunk: while one group runs, the next is staged into pinned memory and uploaded on a copy stream and the
 * previous group's label planes come back on another -- page-at-a-time callers (main.py:490-503 per page) pay the PCIe
 * copies serially.  Results equal n_pages sbbseg_segment_page calls. */
int sbbseg_segment_pages(sbbseg_ctx* c, int n_pages, const uint8_t* const* pages_hwc, int Hp, int Wp, uint8_t* const* labels_hw);
/* Many equally sized pages in one call (the reference walks them one by one: ocrd_cli.py:51 -> main.py:2056 per page): their
 * tiles are pooled into chunks of up to max_batch tiles, so launches stay large when a page has few tiles (a 3500x2500 page has
 * 70; the persistent conv grids want a few hundred).  d_pages_hwc / d_labels_hw: HOST arrays of n_pages DEVICE pointers
 * (uint8 [Hp][Wp][3] in, uint8 [Hp][Wp] out).  Result per page == sbbseg_segment_page_dev. */
int sbbseg_segment_pages_dev(sbbseg_ctx* c, int n_pages, const void* const* d_pages_hwc, int Hp, int Wp, void* const* d_labels_hw);
int sbbseg_segment_page_dev(sbbseg_ctx* c, const void* d_page_hwc, int Hp, int Wp, void* d_labels_hw);

/* Same, with the page rescale of get_image_and_scales (main.py:196-214: cv2.resize INTER_NEAREST to Hp x Wp)
 * fused into the tile gather: page is the STORED image [Hs][Ws][3], labels are [Hp][Wp]; the rescaled
 * page is never materialised.  Identical to sbbseg_segment_page on the nearest-resized page. */
int sbbseg_segment_page_scaled(sbbseg_ctx* c, const uint8_t* page_hwc, int Hs, int Ws, int Hp, int Wp,
                               uint8_t* labels_hw);

/* The layout stage's caller fused in: extract_text_regions (main.py:439-447) = otsu_copy (main.py:178-194:
 * cv2.threshold(channel 0, THRESH_BINARY + THRESH_OTSU), the channel-0 result written to all three
 * channels) + astype(uint8) + do_prediction(patches=True), on the page rescaled to Hp x Wp as above
 * (Hs == Hp and Ws == Wp: no rescale).  Histogram and threshold run on the device; the binarised page
 * is never materialised (the tile gather compares channel 0 with the threshold).  *threshold (may be
 * NULL) receives the Otsu threshold. */
int sbbseg_segment_page_otsu(sbbseg_ctx* c, const uint8_t* page_hwc, int Hs, int Ws, int Hp, int Wp,
                             uint8_t* labels_hw, int* threshold);
/* building blocks of the same for sharded runs: threshold of a device page into a device int, and a
 * tile range gathered through that threshold */
int sbbseg_otsu_dev(sbbseg_ctx* c, const void* d_page_hwc, int Hp, int Wp, int* d_threshold);
int sbbseg_segment_tile_range_bin_dev(sbbseg_ctx* c, const void* d_page_hwc, int Hp, int Wp, int first_tile,
                                      int n_tiles, const int* d_threshold, void* d_tile_labels);

/* The two patch stages as textline_detector.run() really calls them (main.py:2061, 2072, 2102): on extract_page's CROPPED page.
 * page = the STORED image [Hs][Ws][3]; Hp x Wp = its size after get_image_and_scales (main.py:196-214); {cx, cy, cw, ch} =
 * extract_page's box on the upscaled page (cv2.boundingRect convention, main.py:404-409); labels = [ch][cw] (x3 with
 * sbbseg_set_label_channels(3)).  binarise != 0: otsu_copy of the CROPPED page first (extract_text_regions, main.py:443: the
 * threshold is the Otsu threshold of the crop's channel 0; *threshold receives it), binarise == 0: textline_contours
 * (main.py:494).  Upscale, crop and binarisation are index arithmetic in the tile gather: neither the upscaled page nor the
 * crop is materialised.  Equal to sbbseg_segment_page[_otsu] on the materialised crop of the nearest-resized page.
 * _dev: page / labels / threshold (may be NULL) are device pointers, work is enqueued on the handle's stream. */
int sbbseg_segment_crop(sbbseg_ctx* c, const uint8_t* page_hwc, int Hs, int Ws, int Hp, int Wp, int cx, int cy, int cw, int ch,
                        int binarise, uint8_t* labels_hw, int* threshold);
int sbbseg_segment_crop_dev(sbbseg_ctx* c, const void* d_page_hwc, int Hs, int Ws, int Hp, int Wp, int cx, int cy, int cw, int ch,
                            int binarise, void* d_labels_hw, int* d_threshold);

/* ---- seam 1, patches=False (main.py:368-380): nearest-resize page to the model size, one forward,
 * argmax, nearest-resize labels to out_h x out_w (cv2.INTER_NEAREST index rule). */
int sbbseg_segment_whole(sbbseg_ctx* c, const uint8_t* page_hwc, int Hp, int Wp,
                         int out_h, int out_w, uint8_t* labels_out);
/* Same for the border stage's actual input: the page do_prediction receives there is the stored image
 * [Hp][Wp] nearest-upscaled to Hs x Ws by get_image_and_scales (main.py:196-214, 387-392).  The two
 * nearest-neighbour index maps are composed, the upscaled page is never built. */
int sbbseg_segment_whole_scaled(sbbseg_ctx* c, const uint8_t* page_hwc, int Hp, int Wp, int Hs, int Ws,
                                int out_h, int out_w, uint8_t* labels_out);

/* ---- stage glue either side of the models, on u8 label planes [H][W]:
 * cv2.erode / cv2.dilate with a ksize x ksize kernel of ones (the reference's self.kernel is 5x5, main.py:57), `iterations` times,
 * OpenCV's default border (outside pixels never win): text_regions erode x 3 / dilate x 4 (main.py:2074-2075), border mask
 * dilate x 6 (main.py:397).  Integer-exact: n iterations of a k x k flat kernel == one (n(k-1)+1)-wide clipped min / max.
 * src may equal dst.  _dev: device pointers, enqueued on the handle's stream. */
#define SBBSEG_MORPH_ERODE  0
#define SBBSEG_MORPH_DILATE 1
int sbbseg_morph_dev(sbbseg_ctx* c, const void* d_src_hw, int H, int W, int op, int ksize, int iterations, void* d_dst_hw);
int sbbseg_morph(sbbseg_ctx* c, const uint8_t* src_hw, int H, int W, int op, int ksize, int iterations, uint8_t* dst_hw);
/* extract_page's box (main.py:394-404): mask > 0 -> dilate 5x5 x 6 -> the component whose OUTER CONTOUR has the largest
 * cv2.contourArea (contours[np.argmax([cv2.contourArea(c) ...])]; a hole's contour lies inside its component's and never wins)
 * -> its bounding box {x, y, w, h} (cv2.boundingRect convention) and pixel count; {0,0,0,0} / 0 when the mask is empty.
 * The contour area of a component = area of the polygon through its boundary pixels' centres = (2x2 pixel cells fully inside)
 * + (cells with three pixels inside) / 2, counted over the component with its holes filled.  The device ranks the components
 * by that sum over the component as it is (a lower bound) and checks the winner against every other component's bounding-box
 * bound; a ring- or frame-shaped blob beside a solid one can leave that undecided, and only then the dilated mask is copied
 * back and the contours are traced on the host (Moore border following + shoelace: exact).  Equal areas: the component whose
 * first pixel comes LAST in raster order wins -- in the reference np.argmax keeps the first maximum of cv2.findContours' list,
 * which runs in reverse discovery order (OpenCV contours.cpp links every new contour in at the head of its parent's child
 * list) [EXT, restated, unpinned: no cv2 build exists here to confirm it on two equal rectangles]. */
int sbbseg_page_box_dev(sbbseg_ctx* c, const void* d_mask_hw, int H, int W, int32_t* box_xywh, int64_t* pixels);
/* extract_page's model + glue in one call: border model on the page as upscaled to Hs x Ws (sbbseg_segment_whole_scaled),
 * then sbbseg_page_box_dev on the label plane while it is still on the device.  mask_out: Hs x Ws labels (x3 with
 * sbbseg_set_label_channels(3)) or NULL. */
int sbbseg_extract_page_box(sbbseg_ctx* c, const uint8_t* page_hwc, int Hp, int Wp, int Hs, int Ws, uint8_t* mask_out,
                            int32_t* box_xywh, int64_t* pixels);
/* the same with the stored page already on the device (run() hands the SAME page to all three stages, main.py:2061-2102: uploaded
 * once, it stays resident for the two patch stages): d_mask_out = device buffer of Hs x Ws labels, or NULL -- the border mask is a
 * local of extract_page (main.py:392-404), only its box leaves the function.  Synchronises the handle's stream (the box is a host
 * result). */
int sbbseg_extract_page_box_dev(sbbseg_ctx* c, const void* d_page_hwc, int Hp, int Wp, int Hs, int Ws, void* d_mask_out,
                                int32_t* box_xywh, int64_t* pixels);

/* get_text_region_contours_and_boxes' existence test (main.py:456-480), the `if len(contours) > 0` that decides whether run() calls
 * the textline model at all (main.py:2083-2096): pixels == label -> 255, cv2.morphologyEx MORPH_OPEN then MORPH_CLOSE with the 5x5
 * kernel, cv2.findContours(RETR_TREE), keep parentless contours whose polygon area >= min_area * H * W (the reference: min_area =
 * 0.00001, max_area = 1).  *present = 1 when at least one contour is kept.  Decided from the largest outer-contour area of the
 * opened / closed plane (the ranking of sbbseg_page_box_dev); the polygons themselves are not produced.  Synchronises the stream. */
int sbbseg_text_regions_present_dev(sbbseg_ctx* c, const void* d_regions_hw, int H, int W, int label, double min_area, int* present);

/* ---- device buffers for callers without a device runtime of their own.  The reference's environment is Keras/TF (requirements.txt),
 * not PyTorch: what run() keeps resident across its three stages (main.py:2056-2107: the stored page, the border mask, the region
 * map, the textline map) lives in buffers the library hands out, and every `_dev` entry point above accepts them.  A buffer belongs
 * to the handle that allocated it (sbbseg_destroy frees what is left); any handle on the same device may use it.  upload / download
 * return when the copy is complete.  sbbseg_download_labels: a u8 label plane [pixels] as 1 channel or as the 3 identical channels
 * do_prediction returns (main.py:366), replicated on the device. */
int sbbseg_device_alloc(sbbseg_ctx* c, size_t bytes, void** d_ptr);
int sbbseg_device_free(sbbseg_ctx* c, void* d_ptr);                       /* NULL ok; a pointer of another handle is an error */
int sbbseg_upload(sbbseg_ctx* c, void* d_dst, const void* src, size_t bytes);
int sbbseg_download(sbbseg_ctx* c, void* dst, const void* d_src, size_t bytes);
int sbbseg_download_labels(sbbseg_ctx* c, uint8_t* dst, const void* d_labels_hw, size_t pixels, int channels);

/* ---- the model-running part of textline_detector.run() (main.py:2056-2107) in ONE call, for callers that want neither Python glue
 * nor device pointers: three finalized handles on one device (border, layout, textline model), the stored page in host memory, the
 * upscaled size Hs x Ws (get_image_and_scales, main.py:196-214).  The page is uploaded once and stays on the device for all three
 * stages.  Error behaviour mirrors run(): extract_page sits outside the reference's try (main.py:2061) -- its failures, an empty
 * border mask included, fail the call; the layout stage and its glue sit in a bare try / except (2069-2091) -- a failure there (a
 * crop smaller than the model input) leaves info->regions_ok = 0 and skips the textline model; the textline model runs only when
 * get_text_region_contours_and_boxes would return a contour (2083, 2096: info->text_present) and a failure there is "no lines"
 * (2152-2157: info->textlines_ok = 0).  Outputs, caller-allocated for the WORST case (the box may be the whole page):
 *   page_mask_out   Hs x Ws x channels bytes or NULL   border labels at the upscaled size
 *   regions_out     Hs x Ws x channels bytes            cleaned layout labels of the box, h x w x channels, densely packed
 *   textlines_out   Hs x Ws bytes                       textline labels of the box, h x w
 * channels = 1 or 3 (the reference's three identical channels, main.py:366). */
typedef struct sbbseg_run_info {
    int32_t box_xywh[4];      /* extract_page's box on the upscaled page (cv2.boundingRect convention) */
    int64_t box_pixels;       /* pixels of the winning component of the dilated border mask */
    int32_t otsu_threshold;   /* otsu_copy's threshold on the crop's channel 0 (main.py:178-194) */
    int32_t regions_ok;       /* 1: regions_out holds the cleaned layout map */
    int32_t text_present;     /* 1: a text-region contour would be kept (the textline model ran) */
    int32_t textlines_ok;     /* 1: textlines_out holds the textline map */
} sbbseg_run_info;
int sbbseg_run_page(sbbseg_ctx* border, sbbseg_ctx* layout, sbbseg_ctx* textline, const uint8_t* page_hwc, int Hp, int Wp, int Hs, int Ws,
                    int channels, uint8_t* page_mask_out, uint8_t* regions_out, uint8_t* textlines_out, sbbseg_run_info* info);

/* ---- stage glue: the rotate-and-project of the deskew search (return_deskew_slope, main.py:1601-1718; per text region,
 * 80 angles in [-25, 25] and 30 more in [-90, -50] -- the reference spreads the regions over cpu_count() processes,
 * main.py:1760-1799).  The H x W u8 region mask is centred on a zero square of side S = (int)(1.4 * max(H, W))
 * (sbbseg_deskew_side; main.py:1613-1621), rotated by every angle exactly as rotate_image does (main.py:159-163:
 * cv2.getRotationMatrix2D((S/2, S/2), angle, 1.0) + cv2.warpAffine INTER_CUBIC / BORDER_REPLICATE), binarised (!= 0,
 * main.py:1642) and summed along its rows (main.py:1546).  counts: host int32 [n_angles][S].  One launch for the sweep.
 * matrices (optional, [n_angles][6] forward 2 x 3 maps) overrides angles_deg; otherwise sbbseg_rotation_matrix is used.
 * The 1-D peak logic on the profiles (scipy gaussian_filter1d / find_peaks, main.py:1545-1599) stays on the host
 * (stages.return_deskew_slope).  [EXT, unpinned]: OpenCV 4.5.1's warpAffine arithmetic is restated (fixed-point source
 * coordinates with 5 fractional bits, float bicubic table with A = -0.75, replicated borders); cv2 is not available to
 * pin it against. */
int sbbseg_deskew_side(int H, int W, int* side);
int sbbseg_rotation_matrix(double cx, double cy, double angle_deg, double* m6);
int sbbseg_deskew_profiles_dev(sbbseg_ctx* c, const void* d_mask_hw, int H, int W, const double* matrices, const double* angles_deg,
                               int n_angles, int32_t* counts);
int sbbseg_deskew_profiles(sbbseg_ctx* c, const uint8_t* mask_hw, int H, int W, const double* matrices, const double* angles_deg,
                           int n_angles, int32_t* counts);

/* ---- multi-GPU: one process per GPU, one handle per process; tiles (sbbseg_segment_tile_range_dev) or whole
 * pages (sbbseg_segment_pages_dev) are sharded by the caller, and the ONE data-path collective -- the all-gather of the u8
 * label maps -- runs on RCCL inside the library, on the handle's stream, so an integrator needs neither PyTorch nor an MPI:
 *   rank 0:     sbbseg_comm_unique_id(id)  -> ship the 128 bytes to every rank by any means (file, socket, env)
 *   every rank: sbbseg_comm_init(ctx, rank, world, id)            (collective: returns when all ranks have joined)
 *               ... sbbseg_segment_tile_range_dev(...) into my slice ...
 *               sbbseg_allgather_labels_dev(ctx, d_my_slice, bytes_per_rank, d_all)      (d_all = world x bytes_per_rank)
 *               sbbseg_stitch_dev(ctx, d_all, Hp, Wp, d_page_mask)
 * librccl is loaded with dlopen on first use (no link-time dependency: the library loads on machines without it).
 * The reference has no counterpart: main.py:259-288 is a serial loop; tiles and pages are independent there. */
int sbbseg_comm_unique_id(char* id128);
int sbbseg_comm_init(sbbseg_ctx* c, int rank, int world, const char* id128);
int sbbseg_comm_info(sbbseg_ctx* c, int* rank, int* world);       /* world = 0: no communicator */
int sbbseg_comm_destroy(sbbseg_ctx* c);
int sbbseg_allgather_labels_dev(sbbseg_ctx* c, const void* d_send, size_t bytes_per_rank, void* d_recv);

/* ---- building blocks (multi-GPU sharding, tests).  tile_xy: host int32 [n][2] = (x0, y0) origins.
 * d_tile_labels: device uint8 [n][H][W]. */
int sbbseg_tile_grid(int Hp, int Wp, int H, int W, int32_t* tile_xy, int capacity, int* nxf, int* nyf);
/* The index rule every nearest-neighbour rescale of this library uses (page upscale of get_image_and_scales, main.py:214;
 * both resizes of the whole-image branch, main.py:371, 378): map[i] = source index of destination index i for
 * cv2.resize(..., interpolation=cv2.INTER_NEAREST) = min(floor(i * (1 / (dst_len / src_len))), src_len - 1).  Host only. */
int sbbseg_nearest_map(int src_len, int dst_len, int32_t* map);
int sbbseg_segment_tiles_dev(sbbseg_ctx* c, const void* d_page_hwc, int Hp, int Wp,
                             const int32_t* tile_xy, int n_tiles, void* d_tile_labels);
/* same, for the contiguous range [first_tile, first_tile+n_tiles) of the page's own tile grid in the
 * reference's call order (x outer, y inner) -- origins come from the closed form, no table upload.
 * This is the unit of work the multi-GPU path shards. */
int sbbseg_segment_tile_range_dev(sbbseg_ctx* c, const void* d_page_hwc, int Hp, int Wp,
                                  int first_tile, int n_tiles, void* d_tile_labels);
int sbbseg_stitch_dev(sbbseg_ctx* c, const void* d_tile_labels, int Hp, int Wp, void* d_labels_hw);
/* ingest only (tests): fills both input forms for n tiles and copies form `form` back as float32 */
int sbbseg_debug_ingest(sbbseg_ctx* c, const uint8_t* page_hwc, int Hp, int Wp, const int32_t* tile_xy,
                        int n_tiles, int form, float* out, size_t out_floats);
/* copy an activation tensor of the last run back as float32 [n][H][W][C] (tests) */
int sbbseg_debug_read_tensor(sbbseg_ctx* c, int tensor_id, int n, float* out, size_t out_floats);

/* A/B knobs of the conv kernel (benchmarking, tests; results never depend on them).  bits 0-1: tile
 * family (0 auto, 1 = 4 waves / 2 LDS stages, 2 = 8 waves / 3 stages, 3 = big 8-wave tiles);
 * bit 2 = one block per tile instead of persistent blocks; bit 3 = no XCD-grouped tile walk;
 * bit 4 = half-K-step stages in a 4-deep ring; bit 5 = XCD-grouped tile walk on single-class layers too;
 * bit 6 = drain the epilogue stores before the next barrier; bit 7 = half-line (64-byte) epilogue stores; bits 8-15 = with bit 5: K limit (units of
 * 64) up to which every block walks a contiguous run of tiles (0 = keep the current limit);
 * bit 16 = 8-phase schedule on the 256x256 tile (half-tile restaging, staggered wave groups);
 * bit 17 = plain gather (per-load address arithmetic) instead of the fast gather on every layer;
 * bit 18 = fused bottleneck blocks run as the three convs they replace; bit 19 = grouped (parity-class) launches walk
 * XCD-contiguous tile ranges; bit 20 = fused bottleneck blocks run bottleneck_fused (one group of four waves per tile)
 * instead of bottleneck_fused_pq (producer / consumer wave groups) -- split mode: block_x3 prefetches the next tile's inner x in
 * place (inside phase C) instead of into its own registers a tile ahead; bit 21 = sbbseg_page_box_dev always ranks on the host;
 * bit 22 = split mode: stem and max-pool as two launches (stem_conv_pairs_x3 + maxpool_kernel) instead of stem_pool_x3;
 * bit 23 = the 224 x 224 decoder conv on conv_igemm_mfma instead of dec_halo_x3 / dec_halo_f16 (LDS-resident halos);
 * bit 24 = split mode: an identity block's last 1x1 conv and the next block's first 1x1 conv (encoder stages 3 / 4) as two
 * conv_igemm_mfma launches instead of expand_reduce_x3; bit 25 = the 3x3 conv of a stage-3 identity block as its own launch in front
 * of expand_reduce instead of inside conv3_expand_reduce */
int sbbseg_debug_set_conv_variant(sbbseg_ctx* c, int variant);
/* the exact host-side contour ranking of sbbseg_page_box_dev on a host mask that is ALREADY dilated (no GPU needed; tests) */
int sbbseg_debug_largest_contour(const uint8_t* mask_hw, int H, int W, int32_t* box_xywh, int64_t* pixels);
/* ... and TWICE the largest outer-contour area itself (host mirror of sbbseg_text_regions_present_dev's ranking; 0 for an empty mask) */
int sbbseg_debug_largest_contour_area2(const uint8_t* mask_hw, int H, int W, int64_t* area2);
/* closed forms of the owned-region launches (no GPU needed; tests): [lo, hi) in tile coordinates of what the stitch keeps of tile t of an
 * axis of n_tiles tiles (origin(t) = min(t * (tile - 2 margin), extent - tile)), and the rows every decoder level below must produce for
 * it: lo_hi[2 k], lo_hi[2 k + 1] for level k = 0 .. levels - 1, level_size[k] = rows of level k (level 0 = the network output) */
int sbbseg_debug_owned_range(int extent, int tile, int margin, int n_tiles, int t, int* lo, int* hi);
int sbbseg_debug_region_rows(int extent, int tile, int margin, int n_tiles, int t, int levels, const int32_t* level_size, int32_t* lo_hi);
/* fill every activation buffer (both lanes) and the tile-label scratch with `byte_value` (tests: a launch that reads what an owned-region
 * launch did not write then shows; 0xFF = NaN in every 16-bit format) */
int sbbseg_debug_poison_activations(sbbseg_ctx* c, int byte_value);
/* counters: which 0 = how often sbbseg_page_box_dev had to fall back to the host ranking on this handle */
int sbbseg_debug_counter(sbbseg_ctx* c, int which, int64_t* value);      /* which: 0 = exact host contour rankings, 1 = patches run through the plan */
/* Test hook for the no-abort guarantee: the nth_check-th next internal host-allocation checkpoint throws
 * std::bad_alloc, which every entry point turns into a non-zero status + sbbseg_last_error() instead of
 * terminating the process (main.py:2061-2157 relies on ordinary exceptions).  0 disarms.  Process-global. */
int sbbseg_debug_inject_alloc_failure(int nth_check);

/* ---- per-op timing with HIP events on the handle's stream (bench.py roofline) */
int sbbseg_profile_enable(sbbseg_ctx* c, int enable);
int sbbseg_profile_reset(sbbseg_ctx* c);
/* accumulated since reset: total ms and number of launches for op `op` */
int sbbseg_profile_get(sbbseg_ctx* c, int op, double* total_ms, int64_t* launches, int64_t* patches);
/* Work op `op` EXECUTED since sbbseg_profile_reset, in whole-patch equivalents: a launch over n patches counts n, an owned-region
 * launch n x (output pixels walked / output pixels of the whole grid).  exec_patches: every launch; timed_exec_patches: the launches
 * the profiling events timed (the denominator's counterpart of sbbseg_profile_get's total_ms).  Executed FLOPs of an op =
 * sbbseg_op_info's flops_per_patch x these. */
int sbbseg_op_executed(sbbseg_ctx* c, int op, double* exec_patches, double* timed_exec_patches);

#ifdef __cplusplus
}
#endif
#endif /* SBBSEG_H */
