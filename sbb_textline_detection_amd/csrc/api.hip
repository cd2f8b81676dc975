// api.hip -- host side of libsbbseg: the C ABI declared in include/sbbseg.h.
//
// Holds the execution plan the Python planner builds (tensors + fused ops), owns all device
// memory, packs weights into the kernels' contraction order, and drives the per-page pipeline
//   ingest (u8 -> LUT -> bf16 tiles)  ->  fused conv plan  ->  head (softmax/argmax)  ->  stitch
// that replaces the reference's per-patch Python loop (main.py:225-380).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <new>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sbbseg.h"
#include <dlfcn.h>

#include <mutex>

#include "internal.h"
#include "region.h"

using namespace sbbseg;

namespace {

thread_local std::string g_err;

int fail(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);   \
    } while (0)

#define REQUIRE(cond, ...)                    \
    do {                                      \
        if (!(cond)) return fail(__VA_ARGS__); \
    } while (0)

// Every extern "C" body runs inside API_BEGIN / API_END: a C++ exception (std::bad_alloc from a std::vector, ...)
// becomes a non-zero status + sbbseg_last_error() instead of terminating the caller's process -- the reference's
// callers rely on ordinary Python exceptions (main.py:2061-2157).
#define API_BEGIN try {
#define API_END                                                                                     \
    }                                                                                               \
    catch (const std::bad_alloc&) { return fail("out of host memory (std::bad_alloc)"); }           \
    catch (const std::exception& e_) { return fail("internal error: %s", e_.what()); }              \
    catch (...) { return fail("unknown internal error"); }

// test hook (sbbseg_debug_inject_alloc_failure): the n-th next alloc_check() throws std::bad_alloc
int g_alloc_fail_countdown = 0;
inline void alloc_check()
{
    if (g_alloc_fail_countdown > 0 && --g_alloc_fail_countdown == 0) throw std::bad_alloc();
}

}  // namespace

namespace sbbseg {
int set_error(const char* fmt, ...)                     // for the other translation units of the library (loader.cpp)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}
}  // namespace sbbseg

namespace {

struct Tensor {
    int H = 0, W = 0, C = 0;
    size_t elems_per_patch = 0;
    char* buf = nullptr;          // zero header + data (of the lane in use)
    char* lane_buf[2] = {nullptr, nullptr};
    bool is_input_form = false;
    int form = -1, pad = 0;
    char* data() const { return buf + kZeroHeaderBytes; }
};

enum OpType { kConv = 0, kPool = 1, kHead = 2, kTail = 3, kBlock = 4 };

struct ConvOp {
    sbbseg_conv_desc d;
    int Ho = 0, Wo = 0, TH = 0, TW = 0;
    int cout_pad = 0, Ktot = 0, total_ksteps = 0, ksteps[2] = {0, 0};
    KTabEntry* d_ktab = nullptr;
    KStepRec* d_kstep = nullptr;
    void* d_w = nullptr;
    float *d_scale = nullptr, *d_shift = nullptr, *d_rscale = nullptr, *d_rshift = nullptr;
    float *d_head_w = nullptr, *d_head_scale = nullptr, *d_head_shift = nullptr;
    // further placement classes merged into this op (parity siblings); class 0 = the fields above
    int n_cls = 1;
    void* d_w_cls[4] = {nullptr, nullptr, nullptr, nullptr};
    KStepRec* d_kstep_cls[4] = {nullptr, nullptr, nullptr, nullptr};
    KTabEntry* d_ktab_cls[4] = {nullptr, nullptr, nullptr, nullptr};
    int ooy_cls[4] = {0, 0, 0, 0}, oox_cls[4] = {0, 0, 0, 0};
    float wmul_cls[4] = {1.f, 1.f, 1.f, 1.f};   // split mode: 2^-s of the class's power-of-two weight pre-scale
    std::vector<float> h_epi;             // host copy of scale | shift | head_w | head_scale | head_shift: parity siblings are
                                          // only merged into one launch when these are identical (they share class 0's)
    uint16_t* d_stem_wfrag = nullptr;     // non-null: the op is the network stem and runs stem_conv_pairs
    int fused_pool = -1;                  // split mode: index of the max-pool op this stem also computes (sbbseg_finalize), or -1
    uint16_t* d_halo_wfrag = nullptr;     // split mode, the 224 x 224 decoder conv: the four classes' weights as MFMA A fragments (dec_halo_x3.hip)
    int* d_halo_taps = nullptr;           //   ... and their taps in K-step order (sbbseg_finalize)
    uint16_t* d_d64_wfrag = nullptr;      // non-null: 3x3 s1 64->64 conv, runs conv3x3_c64_direct
    int fused_reduce = -1;                // split mode: index of the NEXT block's first 1x1 conv, computed by this (expand) conv's launch too
                                          // (expand_reduce_x3.hip; sbbseg_finalize), or -1
    uint16_t *d_er_w3 = nullptr, *d_er_w1 = nullptr;      //   ... the two convs' packed rows as MFMA A fragments
    bool fused_into_expand = false;       // split mode: this op's output is written by the launch of the op before it; it launches nothing
    int fused_conv3 = -1;                 // on an expand conv with fused_reduce: index of the block's 3x3 conv, which the same launch computes too
                                          // (conv3_expand_reduce.hip; sbbseg_finalize), or -1
    uint16_t* d_c3_w2 = nullptr;          //   ... the 3x3 conv's packed rows as MFMA A fragments, and its K-steps (taps / channel groups) in order
    int* d_c3_k0 = nullptr;
    bool fused_into_c3 = false;           // the 3x3 conv of such a block: its output tensor lives in LDS only, the op launches nothing
    std::vector<float> h_w[2];            // host copy of a small 1x1 conv's weights ([cin][cout] per source): bottleneck fusion
                                          // (sbbseg_finalize) repacks them as MFMA A fragments
    bool fg_ok = true;                    // every K-step (of every class) regular: the fast gather of conv_igemm_mfma applies
                                          // (ConvParams::fast_gather) if each source's taps also span at most 4 x 4 offsets
    int tap_lo[2][2] = {{127, 127}, {127, 127}}, tap_hi[2][2] = {{-127, -127}, {-127, -127}};   // [source][y|x] over all classes
    std::vector<KStepRec> h_ksteps_cls[4];   // host copies: sbbseg_finalize builds the fast gather's tables from them
    FgStepRec* d_fgstep_cls[4] = {nullptr, nullptr, nullptr, nullptr};
    bool fg_pointwise = false;            // all taps (0, 0) in bounds: ConvParams::fast_gather = 2 (no masks)
};

struct PoolOp {
    int src, dst, k, stride, Ho, Wo; float *d_pre_scale = nullptr, *d_pre_shift = nullptr; int pre_relu = 0;
    bool fused_into_stem = false;         // split mode: the stem op before it writes this pool's output too (stem_pool_x3); the op then launches nothing
};

struct HeadOp {
    int src, cin, classes;
    float *d_w = nullptr, *d_scale = nullptr, *d_shift = nullptr;
};

struct TailOp {
    int src0 = -1, img = -1, classes = 0;
    void* d_wfrag = nullptr;
    float *d_scale = nullptr, *d_shift = nullptr, *d_head_w = nullptr, *d_head_scale = nullptr, *d_head_shift = nullptr;
};

// a fused ResNet bottleneck block (bottleneck_fused): the three convs it replaces stay alive as `parts` of the op
// (they own the scale / shift arrays and the 3x3 fragments the fused kernel reads, and they are what runs when the
// fusion is switched off at run time, conv variant bit 18)
struct BlockOp {
    int x_tensor = -1, out_tensor = -1, cin = 0, proj = 0, H = 0, W = 0;
    uint16_t *d_w1 = nullptr, *d_w3 = nullptr;
    float wmul[3] = {1.f, 1.f, 1.f};      // split mode: 2^-s of the three convs' weight pre-scales
};

struct Op {
    OpType type;
    std::string name;
    double flops = 0, min_bytes = 0;
    double issued_flops = 0;      // MFMA work the kernel really issues per patch (K padding, pre-summed taps, 3x in split mode)
    ConvOp conv;
    PoolOp pool;
    HeadOp head;
    TailOp tail;
    BlockOp block;
    std::vector<Op> parts;        // kBlock: the convs it fuses
    double prof_ms = 0;
    int64_t prof_launches = 0, prof_patches = 0;
    int region_level = -1;        // >= 0: a decoder level of the owned-region chain (sbbseg_finalize: region_chain; region.h)
    double exec_patches = 0;      // work executed since sbbseg_profile_reset, in whole-patch equivalents: a launch of n patches adds n, an
                                  // owned-region launch n x (pixels walked / pixels of the whole grid)
    double prof_exec_patches = 0; // the same, over the launches the profiling events timed (prof_ms)
};

struct PendingEvent { int op; hipEvent_t a, b; int patches; double exec; };

// owned-region launches (region.h): tables of the chunk a lane is running
struct RegionRun {
    bool on = false;
    int kind[kRegionMaxLevels] = {0};            // 0 = tile table, 1 = pixel map (what the level's kernel takes: dec_halo_* / tail vs conv_igemm_mfma)
    int total[kRegionMaxLevels] = {0};           // entries (per class)
    double frac[kRegionMaxLevels] = {0};         // pixels walked / pixels of the whole grid, over the chunk
    uint32_t* tab[kRegionMaxLevels] = {nullptr};
};

}  // namespace

struct sbbseg_ctx {
    int device = 0;
    int precision = kBF16;
    int elem = 2;                 // bytes per stored half-element (weights, one activation plane)
    int planes = 1;               // 16-bit planes per activation element: 2 in the split mode (hi, lo), else 1
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // second lane: its own activation buffers and stream; a chunk of tiles is split over the two lanes so
    // that one half's launch tails (few tiles left, most CUs idle) are filled by the other half's kernels
    hipStream_t lane_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int lanes = 2, lane1_batch = 0;
    int lane_prio = 0, prio_least = 0, prio_greatest = 0;      // priority class of lane_stream (never the own stream's class: see sbbseg_create)
    // CU-partitioned lanes (round 4, opt-in: SBBSEG_CU_SPLIT=1): two private streams, each confined to HALF of the chip's CUs (hipExtStreamCreateWithCUMask: the
    // mask's bits are dealt round-robin over the eight XCDs, so each half = 16 CUs of every XCD).  The chip is power-capped under
    // dense MFMA (1 350 TFLOP/s at 1.65 GHz on 256 CUs, 910 at 2.2 GHz on 128) and its HBM fabric saturates from half the CUs
    // (4.2 of 4.4 TB/s): an MFMA-bound kernel on one half beside an HBM-bound kernel on the other half get 0.52x + 0.91x of their
    // whole-chip rates AT THE SAME TIME (tools/probes/cu_mask_probe.hip, profiles/r04_cu_mask_probe.txt).  The lanes walk their
    // tile units half a network apart (tile_range_impl), so that one lane's decoder (MFMA) runs beside the other's encoder (HBM).
    hipStream_t half_stream[2] = {nullptr, nullptr};
    hipEvent_t ev_half[2] = {nullptr, nullptr};
    int half_cus[2] = {0, 0};
    bool cu_split = false;           // both masked streams exist
    int stagger_min_tiles = 64;      // tile ranges from this size on take the CU-partitioned, staggered schedule (SBBSEG_STAGGER_MIN_TILES)
    int in_H = 0, in_W = 0, in_C = 0;
    std::vector<Tensor> tensors;
    std::vector<Op> ops;
    int form_tensor[2] = {-1, -1};
    int classes = 0, max_batch = 0;
    bool finalized = false;
    size_t device_bytes = 0;
    // run-time buffers
    float* d_lut = nullptr;
    unsigned* d_hist = nullptr;   // [256] channel-0 histogram + [1] Otsu threshold (int) behind it
    int* d_tile_xy = nullptr;          // [max_batch][2]
    uint8_t* d_batch_labels = nullptr; // [max_batch][H][W] (predict / whole-image path)
    float* d_probs = nullptr;          // [max_batch][H][W][classes], lazily allocated
    float* d_xin = nullptr;            // predict(): staged float input, lazily allocated
    float* d_ks_ws = nullptr; size_t ks_ws_cap = 0;      // split-K partial sums (whole-image branch)
    bool ksplit = true, ksplit_now = false;              // SBBSEG_KSPLIT=0 switches it off; _now: inside the whole-image branch's run_plan
    uint8_t* d_page = nullptr; size_t page_cap = 0;
    uint8_t* d_page_labels = nullptr; size_t page_labels_cap = 0;
    uint8_t* d_page_labels3 = nullptr; size_t page_labels3_cap = 0;   // 3-channel copy for label_channels == 3
    int label_channels = 1;
    uint8_t* d_tile_labels = nullptr; size_t tile_labels_cap = 0;
    int *d_own_x = nullptr, *d_own_y = nullptr; size_t own_cap = 0;
    int own_Hp = -1, own_Wp = -1, own_nyf = 0;
    bool own_dedupe = false;          // the cached owner tables index the deduplicated grid (see fused_grid)
    // Duplicate clamped tiles: when extent % mid lies in (0, tile - mid] the inward clamp (main.py:276-281) gives the LAST
    // TWO tiles of an axis the same origin -- the reference runs the same forward twice and pastes the same labels twice.  The fused
    // page paths skip the repeat (same label map, 1 / n of the forwards of that axis saved); the tile-indexed entry points
    // (sbbseg_tile_grid, _segment_tile_range_dev, _stitch_dev: the multi-rank protocol) keep the reference's call list.
    bool dedupe = true;               // sbbseg_set_dedupe / SBBSEG_DEDUPE=0
    int64_t forwards = 0;             // patches run through the plan so far (sbbseg_debug_counter 1)
    int *d_map = nullptr; size_t map_cap = 0;
    int *d_wmap = nullptr; size_t wmap_cap = 0;      // gather tables of the whole-image branch, cached per geometry
    int wmap_key[6] = {0, 0, 0, 0, 0, 0};            // {Hp, Wp, Hs, Ws, out_h, out_w} (0 = none)
    int map_key[4] = {0, 0, 0, 0};     // {Hs, Ws, Hp, Wp} the nearest maps in d_map were built for (sbbseg_segment_crop_dev; 0 = none)
    // stage glue scratch (morphology planes, union-find arrays, result words)
    uint8_t *d_morph_a = nullptr, *d_morph_b = nullptr; size_t morph_a_cap = 0, morph_b_cap = 0;
    // pipelined multi-page host path (sbbseg_segment_pages): copy streams, two slots of pinned staging + device buffers
    hipStream_t copy_in = nullptr, copy_out = nullptr;
    hipEvent_t pp_in[2] = {nullptr, nullptr}, pp_comp[2] = {nullptr, nullptr}, pp_out[2] = {nullptr, nullptr};
    uint8_t *pp_h_in[2] = {nullptr, nullptr}, *pp_h_out[2] = {nullptr, nullptr}, *pp_d_in[2] = {nullptr, nullptr}, *pp_d_out[2] = {nullptr, nullptr},
            *pp_d_out3[2] = {nullptr, nullptr};
    size_t pp_in_cap = 0, pp_out_cap = 0, pp_out3_cap = 0;        // device buffers (bytes each)
    size_t pp_hin_cap = 0, pp_hout_cap = 0;                       // pinned host staging (bytes each)
    bool pp_ready = false;                                        // streams + events of the page pipeline exist
    // RCCL communicator of the sharded path (sbbseg_comm_init; librccl is dlopen'ed on first use)
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    void* d_deskew = nullptr; size_t deskew_cap = 0;      // inverse maps | bicubic table | row counts of sbbseg_deskew_profiles
    // sbbseg_run_page's resident buffers (owned by the handle passed as `border` / `layout` / `textline` respectively)
    uint8_t *d_run_page = nullptr, *d_run_mask = nullptr, *d_run_a = nullptr, *d_run_b = nullptr;
    size_t run_page_cap = 0, run_mask_cap = 0, run_a_cap = 0, run_b_cap = 0;
    int *d_cc_parent = nullptr, *d_cc_count = nullptr; size_t cc_parent_cap = 0, cc_count_cap = 0;
    bool force_host_contours = false;     // test hook (conv variant bit 21): always take the exact host ranking
    int host_contour_calls = 0;           // how often the exact host ranking ran (sbbseg_debug_counter)
    int* d_cc_list = nullptr;             // [6 + kCcMaxRivals]: launch_largest_contour's result record
    int* d_cc_aux = nullptr; size_t cc_aux_cap = 0;      // five int planes: doubled cell area + bounding boxes per root (sbbseg_page_box_dev)
    unsigned long long* d_cc_small = nullptr;      // [0] best key, [1..2] box (4 ints)
    // profiling
    bool profiling = false;
    int conv_variant = 0;
    bool ph8 = false;            // 8-phase schedule on the 256x256 tile (opt-in, conv variant bit 16)
    int fg_min_ksteps = 9;             // convs with real taps take the fast gather from this many K-steps on (SBBSEG_FG_MIN)
    int fg_pointwise_min_ksteps = 4;   // pointwise convs take the fast gather from this many K-steps on (SBBSEG_FG_POINTWISE_MIN)
    bool ranged_walk = false;    // A/B: grouped launches walk XCD-contiguous tile ranges (conv variant bit 19)
    bool block_pq = true;        // fused bottleneck blocks run the producer / consumer form (conv variant bit 20: the one-group form)
    bool unfuse_blocks = false;  // A/B: run a fused bottleneck block as its three convs (conv variant bit 18)
    bool unfuse_stem_pool = false;     // A/B: stem and max-pool as two launches (conv variant bit 22)
    bool no_dec_halo = false;          // A/B: the 224 x 224 decoder conv on the generic kernel (conv variant bit 23)
    bool no_expand_reduce = false;     // A/B: expand + next reduce 1x1 convs as two launches (conv variant bit 24)
    bool no_c3er = false;              // A/B: the 3x3 conv of a stage-3 identity block as its own launch in front of expand_reduce (conv variant bit 25)
    bool plain_gather = false;   // A/B: per-load address arithmetic instead of the fast gather (conv variant bit 17)
    int contig_max_k = 0;        // short-K layers up to this K walk their tiles in per-block contiguous runs (tile map 2)
    int fused_heads = 0;
    int num_cus = 256;
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> free_events;
    // owned-region launches of the decoder (region.h; sbbseg_set_owned_regions): 0 = off, 1 = the fused page paths (default), 2 = the
    // tile-range entry points of the multi-rank protocol too (their tile labels are then defined on the owned regions only)
    int owned_mode = 1;
    int region_levels = 0;                        // decoder levels of the chain found by sbbseg_finalize (0: the plan has none)
    int region_op[kRegionMaxLevels] = {0};        // op index per level (level 0 = the tail)
    uint32_t* d_rtab[2][kRegionMaxLevels] = {{nullptr}, {nullptr}};     // per lane and level: the chunk's table (allocated on first use)
    size_t rtab_cap[2][kRegionMaxLevels] = {{0}, {0}};
    RegionRun rr;                                 // the chunk run_plan is launching (set by tile_range_impl around run_plan)
    double last_exec_frac = 1.0;                  // share of its output grid the op being launched walks (run_plan's accounting; launch_op resets
                                                  // it to 1 when an A/B knob takes a level off its owned-region form)
    std::vector<std::pair<void*, size_t>> user_bufs;      // sbbseg_device_alloc's buffers still alive (freed by sbbseg_destroy)
};

namespace {

int dmalloc(sbbseg_ctx* c, void** p, size_t bytes)
{
    HIPCHK(hipMalloc(p, bytes));
    c->device_bytes += bytes;
    return 0;
}

template <typename T>
int upload(sbbseg_ctx* c, T** dptr, const T* host, size_t n)
{
    if (dmalloc(c, (void**)dptr, n * sizeof(T))) return 1;
    HIPCHK(hipMemcpy(*dptr, host, n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

int ensure(sbbseg_ctx* c, void** p, size_t* cap, size_t bytes)
{
    if (*cap >= bytes) return 0;
    if (*p) {
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipFree(*p));
        c->device_bytes -= *cap;
        *p = nullptr;
        *cap = 0;
    }
    if (dmalloc(c, p, bytes)) return 1;
    *cap = bytes;
    return 0;
}

int margin_of(int W) { return (int)(0.1 * (double)W); }   // main.py:233  int(0.1 * img_width_model)

// per-axis tiles exactly as main.py:246-281: count = ceil(extent/mid); origin t*mid clamped inward
int axis_tiles(int extent, int tile, int margin, std::vector<int>& origin)
{
    const int mid = tile - 2 * margin;
    if (mid <= 0 || extent < tile) return -1;
    const int n = (extent + mid - 1) / mid;
    origin.resize(n);
    for (int t = 0; t < n; ++t) {
        int d = t * mid;
        if (d + tile > extent) d = extent - tile;
        origin[t] = d;
    }
    return n;
}

// owner table of one axis (closed form of the crop + overwrite order, main.py:294-364)
void axis_owner(int extent, int tile, int margin, const std::vector<int>& origin, std::vector<int>& own)
{
    own.assign(extent, 0);
    const int n = (int)origin.size();
    for (int t = 0; t < n; ++t) {
        const int lo = origin[t] + (t == 0 ? 0 : margin);
        const int hi = origin[t] + (t == n - 1 ? tile : tile - margin);
        for (int q = lo; q < hi; ++q) own[q] = (t << 16) | (q - origin[t]);
    }
}

// cv2.resize(..., INTER_NEAREST) index rule [EXT OpenCV resizeNN]: min(floor(dst * (1/(dst_len/src_len))), src_len-1)
void nearest_map(int src_len, int dst_len, std::vector<int>& m)
{
    m.resize(dst_len);
    const double inv = 1.0 / ((double)dst_len / (double)src_len);
    for (int i = 0; i < dst_len; ++i) {
        int s = (int)std::floor(i * inv);
        m[i] = s < src_len - 1 ? s : src_len - 1;
    }
}

int get_event(sbbseg_ctx* c, hipEvent_t* e)
{
    if (!c->free_events.empty()) {
        *e = c->free_events.back();
        c->free_events.pop_back();
        return 0;
    }
    HIPCHK(hipEventCreate(e));
    return 0;
}

int resolve_pending(sbbseg_ctx* c)
{
    if (c->pending.empty()) return 0;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto& pe : c->pending) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, pe.a, pe.b));
        Op& op = c->ops[pe.op];
        op.prof_ms += ms;
        op.prof_launches += 1;
        op.prof_patches += pe.patches;
        op.prof_exec_patches += pe.exec;
        c->free_events.push_back(pe.a);
        c->free_events.push_back(pe.b);
    }
    c->pending.clear();
    return 0;
}

// the 224 x 224 decoder conv takes the LDS-resident-halo kernel (dec_halo_x3 / dec_halo_f16) -- also decides which table an
// owned-region chunk builds for that level (tile origins vs class-grid pixels)
bool runs_dec_halo(const sbbseg_ctx* c, const ConvOp& co) { return co.d_halo_wfrag && !c->no_dec_halo && !(c->conv_variant & 3); }

// ---- forward pass over n patches whose input forms are already filled -------------------------
int launch_op(sbbseg_ctx* c, Op& op, int n, uint8_t* d_labels, float* d_probs)
{
        if (op.type == kBlock) {
            if (c->unfuse_blocks) {
                for (auto& part : op.parts)
                    if (launch_op(c, part, n, d_labels, d_probs)) return 1;
                return 0;
            }
            const BlockOp& bo = op.block;
            BlockParams bp;
            bp.x = c->tensors[bo.x_tensor].buf; bp.n = n; bp.H = bo.H; bp.W = bo.W; bp.proj = bo.proj; bp.pq = c->block_pq ? 1 : 0;
            bp.w1 = bo.d_w1; bp.w2 = op.parts[1].conv.d_d64_wfrag; bp.w3 = bo.d_w3;
            bp.s1 = op.parts[0].conv.d_scale; bp.b1 = op.parts[0].conv.d_shift;
            bp.s2 = op.parts[1].conv.d_scale; bp.b2 = op.parts[1].conv.d_shift;
            bp.s3 = op.parts[2].conv.d_scale; bp.b3 = op.parts[2].conv.d_shift;
            bp.out = c->tensors[bo.out_tensor].data();
            bp.wmul1 = bo.wmul[0]; bp.wmul2 = bo.wmul[1]; bp.wmul3 = bo.wmul[2];
#ifdef SBBSEG_PROBES
            { static const int blk_dbg = getenv("SBBSEG_BLOCK_DBG") ? atoi(getenv("SBBSEG_BLOCK_DBG")) : 0; bp.dbg = blk_dbg; }
#endif
            if (c->precision == kF16X3) HIPCHK(launch_block_x3(bp, c->num_cus, c->stream));
            else HIPCHK(launch_bottleneck(bp, c->precision, c->num_cus, c->stream));
        } else if (op.type == kConv) {
            const ConvOp& co = op.conv;
            if (co.fused_into_expand && !c->no_expand_reduce && !(c->conv_variant & 3)) return 0;      // written by the expand conv's launch (expand_reduce_x3)
            if (co.fused_into_c3 && !c->no_c3er && !c->no_expand_reduce && !(c->conv_variant & 3)) return 0;     // computed by the expand conv's launch (conv3_expand_reduce)
            ConvParams p;
            p.ks_shift = 0; p.ks_ws = nullptr; p.ks_split_elems = 0;
            memset(&p, 0, sizeof(p));
            p.n_src = co.d.n_src;
            // (short-K layers keep the plain gather: the per-tile mask set-up costs them 1-10 %; from ~9 K-steps on the fast
            // gather wins 2-10 %, profiles/r02_experiments.md)
            bool fg = co.d_fgstep_cls[0] && !c->plain_gather;
            for (int s = 0; s < co.d.n_src; ++s) {
                const Tensor& t = c->tensors[co.d.src[s].tensor];
                SrcDesc& sd = p.src[s];
                sd.base = t.buf;
                sd.PH = t.H; sd.PW = t.W;
                sd.pix_bytes = t.C * c->elem * c->planes;
                sd.lo_off = c->planes == 2 ? split_group(t.C) * c->elem : 64;       // split mode: hi -> lo inside a channel group
                sd.shift = co.d.src[s].up_shift;
                sd.lim_y = t.H << sd.shift; sd.lim_x = t.W << sd.shift;
                sd.ksteps = co.ksteps[s];
                sd.sy_shift = co.d.src[s].stride_y == 2; sd.sx_shift = co.d.src[s].stride_x == 2;
                const size_t bytes = kZeroHeaderBytes + t.elems_per_patch * (size_t)n * c->elem * c->planes;
                sd.bytes = (uint32_t)bytes;
                // fast gather: bit 31 of a lane offset marks an out-of-bounds tap, so every real offset must stay below 2^31
                sd.tap_lo_y = co.tap_lo[s][0]; sd.tap_lo_x = co.tap_lo[s][1];
                if (bytes + (size_t)kFgBiasPixels(t.W) * sd.pix_bytes >= ((size_t)1 << 31)) fg = false;
            }
            p.fast_gather = fg ? (co.fg_pointwise ? 2 : 1) : 0;
            p.ktab = co.d_ktab; p.kstep = co.d_kstep; p.half_stages = (c->conv_variant & 16) ? 1 : 0;
            p.variant = c->conv_variant & 3; p.persist_blocks = (c->conv_variant & 4) ? 0 : c->num_cus;
            p.M = n * co.Ho * co.Wo;
            p.variant_flags = ((c->conv_variant & 64) ? 1 : 0) | ((c->conv_variant & 128) ? 2 : 0) | (c->ph8 ? 4 : 0);
#ifdef SBBSEG_PROBES
            { static const int probe_local = getenv("SBBSEG_CONV_PROBE_LOCAL") ? atoi(getenv("SBBSEG_CONV_PROBE_LOCAL")) : 0; if (probe_local) p.variant_flags |= 32; }
            { static const int probe_whot = getenv("SBBSEG_CONV_PROBE_WHOT") ? atoi(getenv("SBBSEG_CONV_PROBE_WHOT")) : 0; if (probe_whot) p.variant_flags |= 64; }
#endif
            // XCD-grouped walk for single-class layers: measured neutral-to-slower (it removes the n_ct-fold
            // re-fetch of the pixel operand, but those layers are not bound by fetch bytes) -> opt-in, bit 5
            // (split mode: on by default -- twice the pixel bytes; +0.7 % page throughput in two runs, profiles/r03_experiments.md)
            // (SBBSEG_PSHARE_MAX_MB: weight-matrix size up to which the pixel-sharing walk is taken; round 6 experiment -- the K >= 768 merge / reduce
            //  convs of stages 4 / 5 and dec0 re-fetch their pixel operand once per channel tile under map 0: PMC FETCH = n_ct x the input)
            static const size_t pshare_max = (size_t)(getenv("SBBSEG_PSHARE_MAX_MB") ? atoi(getenv("SBBSEG_PSHARE_MAX_MB")) : 2) << 20;
            p.tile_map = (((c->conv_variant & 32) || c->precision == kF16X3) && !(c->conv_variant & 8) && (c->conv_variant & 4) == 0 && co.d.cout > conv_tile_bc(co.d.cout) && p.M >= 256 * 128 &&
                          (size_t)co.cout_pad * co.Ktot * c->elem <= pshare_max) ? 1 : 0;
            if (p.tile_map == 1 && co.n_cls == 1 && co.Ktot <= c->contig_max_k) p.tile_map = 2;
            p.w = co.d_w; p.Ktot = co.Ktot; p.total_ksteps = co.total_ksteps;
            p.Ho = co.Ho; p.Wo = co.Wo; p.M = n * co.Ho * co.Wo;
            p.TH = co.TH; p.TW = co.TW; p.osy = co.d.out_stride_y; p.osx = co.d.out_stride_x;
            p.ooy = co.d.out_off_y; p.oox = co.d.out_off_x;
            p.n_cls = co.n_cls;
            p.cls_minor = 0;
            if (co.n_cls > 1 && (co.n_cls & (co.n_cls - 1)) == 0 && !(c->conv_variant & 8) && !(c->conv_variant & 4) &&
                (size_t)co.cout_pad * co.Ktot * c->elem <= ((size_t)16 << 20)) {     // (dec1/dec2 too: fetch -30 / -53 %, time unchanged)
                p.cls_minor = 1;      // small weights: let the classes share their source pixels in one L2
                p.tile_map = c->ranged_walk ? 3 : 1;
            }
            for (int q = 0; q < 4; ++q) {
                p.w_cls[q] = co.d_w_cls[q]; p.kstep_cls[q] = co.d_kstep_cls[q]; p.ktab_cls[q] = co.d_ktab_cls[q]; p.fgstep_cls[q] = co.d_fgstep_cls[q];
                p.ooy_cls[q] = co.ooy_cls[q]; p.oox_cls[q] = co.oox_cls[q]; p.wmul_cls[q] = co.wmul_cls[q];
            }
            p.head_classes = co.d.head_classes; p.head_w = co.d_head_w; p.head_scale = co.d_head_scale;
            p.head_shift = co.d_head_shift; p.labels = d_labels; p.probs = d_probs;
            p.cout = co.d.cout; p.scale = co.d_scale; p.shift = co.d_shift;
            p.out = co.d.out_tensor >= 0 ? c->tensors[co.d.out_tensor].data() : nullptr;
            p.residual = co.d.residual_tensor >= 0 ? c->tensors[co.d.residual_tensor].data() : nullptr;
            p.raw_out = co.d.raw_out_tensor >= 0 ? c->tensors[co.d.raw_out_tensor].data() : nullptr;
            p.raw_scale = co.d_rscale; p.raw_shift = co.d_rshift; p.relu = co.d.relu;
            // owned-region launch of a decoder level (region.h): the chunk's table replaces the walk over the whole output grid
            const int rlv = c->rr.on ? op.region_level : -1;
            if (rlv >= 0 && c->rr.total[rlv] == 0) return 0;                         // (no patch of the chunk keeps anything)
            if (rlv >= 0 && c->rr.kind[rlv] == 1) {
                // the pixel table is read by the fast-gather form of conv_igemm_mfma, 2-stage whole-K-step tiles only (what every decoder conv
                // runs by default); under an A/B knob that takes the level elsewhere it is launched whole -- a superset, same results
                if (p.fast_gather && !p.half_stages && p.variant != 2 && !(p.variant_flags & 4)) { p.rmap = c->rr.tab[rlv]; p.M = c->rr.total[rlv]; }
                else c->last_exec_frac = 1.0;
            }
            if (co.d_stem_wfrag && !(c->conv_variant & 3)) {
                const Tensor& st = c->tensors[co.d.src[0].tensor];
                StemParams sp;
                sp.pairs = st.buf; sp.PHt = st.H; sp.PWt = st.W; sp.n = n; sp.Ho = co.Ho; sp.Wo = co.Wo;
                sp.wfrag = co.d_stem_wfrag; sp.scale = co.d_scale; sp.shift = co.d_shift; sp.relu = co.d.relu; sp.wmul = co.wmul_cls[0];
                sp.out = c->tensors[co.d.out_tensor].data();
                if (co.fused_pool >= 0 && !c->unfuse_stem_pool) {
                    const PoolOp& po = c->ops[co.fused_pool].pool;
                    sp.pool_out = c->tensors[po.dst].data(); sp.pool_scale = po.d_pre_scale; sp.pool_shift = po.d_pre_shift;
                    sp.pool_relu = po.pre_relu; sp.pool_Ho = po.Ho; sp.pool_Wo = po.Wo;
                    sp.x3 = c->precision == kF16X3;
                    HIPCHK(launch_stem_pool_x3(sp, c->num_cus, c->stream));
                } else
                    HIPCHK(launch_stem(sp, c->precision, c->num_cus, c->stream));
            } else if (co.d_d64_wfrag && !(c->conv_variant & 3)) {
                const Tensor& st = c->tensors[co.d.src[0].tensor];
                Direct64Params dp;
                dp.src = st.buf; dp.n = n; dp.H = st.H; dp.W = st.W; dp.wfrag = co.d_d64_wfrag;
                dp.scale = co.d_scale; dp.shift = co.d_shift; dp.relu = co.d.relu; dp.out = c->tensors[co.d.out_tensor].data();
                dp.wmul = co.wmul_cls[0];
                HIPCHK(launch_direct64(dp, c->precision, c->num_cus, c->stream));
            } else if (co.fused_reduce >= 0 && !c->no_expand_reduce && !(c->conv_variant & 3)) {
                const ConvOp& ro = c->ops[co.fused_reduce].conv;
                if (co.fused_conv3 >= 0 && !c->no_c3er) {
                    const ConvOp& k3 = c->ops[co.fused_conv3].conv;
                    const Tensor& ta = c->tensors[k3.d.src[0].tensor];
                    C3ERParams cp;
                    cp.a = ta.buf; cp.x = c->tensors[co.d.residual_tensor].buf;
                    cp.y = c->tensors[co.d.out_tensor].buf; cp.a2 = c->tensors[ro.d.out_tensor].buf;
                    cp.n = n; cp.H = ta.H; cp.W = ta.W; cp.C = co.d.src[0].channels; cp.x3 = c->precision == kF16X3;
                    cp.w2frag = co.d_c3_w2; cp.w3frag = co.d_er_w3; cp.w1frag = co.d_er_w1;
                    cp.s2 = k3.d_scale; cp.h2 = k3.d_shift; cp.s3 = co.d_scale; cp.h3 = co.d_shift; cp.s1 = ro.d_scale; cp.h1 = ro.d_shift;
                    cp.wmul2 = k3.wmul_cls[0]; cp.wmul3 = co.wmul_cls[0]; cp.wmul1 = ro.wmul_cls[0];
                    cp.k0 = co.d_c3_k0;
                    HIPCHK(launch_conv3_expand_reduce(cp, c->num_cus, c->stream));
                    return 0;
                }
                ExpRedParams ep;
                ep.b = c->tensors[co.d.src[0].tensor].buf; ep.x = c->tensors[co.d.residual_tensor].buf;
                ep.y = c->tensors[co.d.out_tensor].buf; ep.a2 = c->tensors[ro.d.out_tensor].buf;
                ep.M = n * co.Ho * co.Wo; ep.C = co.d.src[0].channels; ep.x3 = c->precision == kF16X3;
                ep.w3frag = co.d_er_w3; ep.w1frag = co.d_er_w1;
                ep.s3 = co.d_scale; ep.h3 = co.d_shift; ep.s1 = ro.d_scale; ep.h1 = ro.d_shift;
                ep.wmul3 = co.wmul_cls[0]; ep.wmul1 = ro.wmul_cls[0];
                ep.dbg = 0;
#ifdef SBBSEG_PROBES
                { static const int er_dbg = getenv("SBBSEG_ER_DBG") ? atoi(getenv("SBBSEG_ER_DBG")) : 0; ep.dbg = er_dbg; }
#endif
                HIPCHK(launch_expand_reduce_x3(ep, c->num_cus, c->stream));
            } else if (runs_dec_halo(c, co)) {
                const Tensor& s0 = c->tensors[co.d.src[0].tensor];
                DecHaloParams hp;
                hp.src0 = s0.buf; hp.skip = c->tensors[co.d.src[1].tensor].buf; hp.PH = s0.H; hp.PW = s0.W; hp.n = n;
                hp.wfrag = co.d_halo_wfrag; hp.taps = co.d_halo_taps; hp.scale = co.d_scale; hp.shift = co.d_shift;
                for (int q = 0; q < 4; ++q) hp.wmul[q] = co.wmul_cls[q];
                hp.relu = co.d.relu; hp.out = c->tensors[co.d.out_tensor].data();
                if (rlv >= 0) { hp.ttab = c->rr.tab[rlv]; hp.n_tab = c->rr.total[rlv]; }
                if (c->precision == kF16X3) HIPCHK(launch_dec_halo_x3(hp, c->num_cus, c->stream));
                else HIPCHK(launch_dec_halo_f16(hp, c->num_cus, c->stream));
            } else {
                // One patch through a long-K conv = 2-32 tiles of 100-400 K-steps at ~1 us a step on as many CUs: the whole-image branch
                // (extract_page's border model, 1 forward per page) splits the K range over the idle CUs -- up to 16 blocks per tile, each
                // at least 4 K-steps, fp32 partial sums added in split order by splitk_finish.  Only there: a split launch differs from
                // the unsplit one in the last bits, and seam 2 / the patch paths promise batch-size-independent results.
                int ks = 0;
                if (c->ksplit_now && n == 1 && c->precision != kF32 && p.fast_gather && conv_tile_bc(p.cout) == 128 && !p.raw_out && !p.head_classes &&
                    p.out && p.cout % 8 == 0 &&
                    ((p.n_cls == 1 && p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 && p.TH == p.Ho && p.TW == p.Wo) ||
                     (p.n_cls == 4 && p.osy == 2 && p.osx == 2 && p.TH == 2 * p.Ho && p.TW == 2 * p.Wo))) {
                    const long tiles = (long)p.n_cls * ((p.M + 127) / 128) * ((p.cout + 127) / 128);
                    while (ks < 4 && (tiles << (ks + 1)) <= 512 && p.total_ksteps % (2 << ks) == 0 && (p.total_ksteps >> (ks + 1)) >= 4) ++ks;
                }
                if (ks > 0) {
                    const size_t split_elems = (size_t)n * p.TH * p.TW * p.cout;
                    if (ensure(c, (void**)&c->d_ks_ws, &c->ks_ws_cap, (split_elems << ks) * sizeof(float))) return 1;
                    p.ks_shift = ks; p.ks_ws = c->d_ks_ws; p.ks_split_elems = (long)split_elems; p.tile_map = 0; p.cls_minor = 0;
                    HIPCHK(launch_conv(p, c->precision, c->stream));
                    HIPCHK(launch_splitk_finish(c->d_ks_ws, 1 << ks, (long)split_elems, (long)n * p.TH * p.TW, p.cout, p.scale, p.shift, p.residual,
                                                p.relu, p.out, c->precision, c->stream));
                } else {
                    HIPCHK(launch_conv(p, c->precision, c->stream));
                }
            }
        } else if (op.type == kPool) {
            const PoolOp& po = op.pool;
            // written by the stem's launch (stem_pool_x3) -- under exactly the condition that sends the stem there (launch_op, kConv)
            if (po.fused_into_stem && !c->unfuse_stem_pool && !(c->conv_variant & 3)) return 0;
            const Tensor& s = c->tensors[po.src];
            HIPCHK(launch_maxpool(s.data(), c->tensors[po.dst].data(), n, s.H, s.W, s.C, po.k, po.stride, po.Ho, po.Wo,
                                  po.d_pre_scale, po.d_pre_shift, po.pre_relu, c->precision, c->stream));
        } else if (op.type == kTail) {
            const TailOp& to = op.tail;
            const Tensor& s0 = c->tensors[to.src0];
            TailParams tp;
            tp.src0 = s0.buf; tp.img = c->tensors[to.img].buf; tp.PH = s0.H; tp.PW = s0.W; tp.n = n;
            tp.wfrag = to.d_wfrag; tp.scale = to.d_scale; tp.shift = to.d_shift; tp.classes = to.classes;
            tp.head_w = to.d_head_w; tp.head_scale = to.d_head_scale; tp.head_shift = to.d_head_shift;
            tp.labels = d_labels; tp.probs = d_probs;
            if (c->rr.on && op.region_level >= 0) {
                if (c->rr.total[op.region_level] == 0) return 0;
                tp.ttab = c->rr.tab[op.region_level]; tp.n_tab = c->rr.total[op.region_level];
            }
            HIPCHK(launch_tail(tp, c->precision, c->num_cus, c->stream));
        } else {
            const HeadOp& ho = op.head;
            const Tensor& s = c->tensors[ho.src];
            HeadParams hp;
            hp.src = s.data(); hp.cin = ho.cin; hp.classes = ho.classes; hp.M = n * s.H * s.W;
            hp.w = ho.d_w; hp.scale = ho.d_scale; hp.shift = ho.d_shift;
            hp.labels = d_labels; hp.probs = d_probs;
            HIPCHK(launch_head(hp, c->precision, c->stream));
        }
    return 0;
}

int run_plan(sbbseg_ctx* c, int n, uint8_t* d_labels, float* d_probs)
{
    c->forwards += n;
    for (size_t i = 0; i < c->ops.size(); ++i) {
        Op& op = c->ops[i];
        hipEvent_t ea = nullptr, eb = nullptr;
        if (c->profiling) {
            if (get_event(c, &ea) || get_event(c, &eb)) return 1;
            HIPCHK(hipEventRecord(ea, c->stream));
        }
        c->last_exec_frac = (c->rr.on && op.region_level >= 0) ? c->rr.frac[op.region_level] : 1.0;
        if (launch_op(c, op, n, d_labels, d_probs)) return 1;
        const double exec = n * c->last_exec_frac;
        op.exec_patches += exec;
        if (c->profiling) {
            HIPCHK(hipEventRecord(eb, c->stream));
            c->pending.push_back({(int)i, ea, eb, n, exec});
        }
    }
    return 0;
}

int fill_ingest(sbbseg_ctx* c, IngestParams& ip)
{
    memset(&ip, 0, sizeof(ip));
    REQUIRE(c->form_tensor[SBBSEG_INPUT_C8] >= 0, "plan has no C8 input form");
    const Tensor& c8 = c->tensors[c->form_tensor[SBBSEG_INPUT_C8]];
    ip.H = c->in_H; ip.W = c->in_W; ip.lut = c->d_lut; ip.c8 = c8.data();
    if (c->form_tensor[SBBSEG_INPUT_PAIRS] >= 0) {
        const Tensor& pr = c->tensors[c->form_tensor[SBBSEG_INPUT_PAIRS]];
        ip.pairs = pr.data(); ip.pad = pr.pad; ip.pairs_w = pr.W;
    }
    return 0;
}

constexpr int kMinLaneTiles = 8;      // a lane gets at least this many tiles, else the chunk runs whole on lane 0

// activation buffers and stream of lane L become the ones run_plan / fill_ingest see
// (halves = true: the lane's CU-masked stream, persistent grids sized for its half of the chip)
struct LaneScope {
    sbbseg_ctx* c; hipStream_t saved; int saved_cus;
    LaneScope(sbbseg_ctx* c_, int lane, bool halves = false) : c(c_), saved(c_->stream), saved_cus(c_->num_cus)
    {
        if (lane == 1)
            for (auto& t : c->tensors) t.buf = t.lane_buf[1];
        if (halves) {
            c->stream = c->half_stream[lane];
            c->num_cus = c->half_cus[lane];
        } else if (lane == 1) {
            c->stream = c->lane_stream;
        }
    }
    ~LaneScope()
    {
        for (auto& t : c->tensors) t.buf = t.lane_buf[0];
        c->stream = saved;
        c->num_cus = saved_cus;
    }
};

// device label plane [pix] -> host buffer, as one plane or as the reference's three identical channels
int labels_to_host(sbbseg_ctx* c, void* host, size_t pix)
{
    if (c->label_channels == 3) {
        if (ensure(c, (void**)&c->d_page_labels3, &c->page_labels3_cap, (pix + 3) / 4 * 12)) return 1;
        HIPCHK(launch_replicate3(c->d_page_labels, c->d_page_labels3, pix, c->stream));
        HIPCHK(hipMemcpyAsync(host, c->d_page_labels3, pix * 3, hipMemcpyDeviceToHost, c->stream));
    } else {
        HIPCHK(hipMemcpyAsync(host, c->d_page_labels, pix, hipMemcpyDeviceToHost, c->stream));
    }
    return 0;
}

// ---- RCCL, loaded at run time: libsbbseg has no link-time dependency on librccl (it must load on a box without it, and
// on the CPU-only build container); the sharded path's one collective is the all-gather of u8 label maps
struct RcclUniqueId { char internal[128]; };                  // = ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;                                      // (handles on distinct devices may be driven from distinct threads)
int rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    REQUIRE(h, "librccl not found (tried librccl.so.1, librccl.so, /opt/rocm/lib/librccl.so.1): %s", dlerror());
    RcclApi a;
    a.GetUniqueId = (int (*)(RcclUniqueId*))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void**, int, RcclUniqueId, int))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    a.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    a.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString) {
        dlclose(h);
        return fail("librccl lacks an expected symbol (ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather)");
    }
    a.lib = h;
    g_rccl = a;
    return 0;
}
#define RCCLCHK(expr)                                                                              \
    do {                                                                                           \
        const int r_ = (expr);                                                                     \
        if (r_ != 0) return fail("%s failed: %s", #expr, g_rccl.GetErrorString(r_));              \
    } while (0)

void comm_release(sbbseg_ctx* c)
{
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
}

// ---- extract_page's ranking, exact, on the host (main.py:398-404): cv2.findContours(RETR_TREE) + cv2.contourArea + np.argmax.
// Only the OUTER contour of a component can win (a hole's contour lies inside it), so: label the 8-connected components, trace
// each one's outer border through its boundary pixels (Moore neighbour tracing from the first pixel in raster order, whose
// west / north neighbours are background; stop when the start pixel is re-entered in the start direction), shoelace area of
// that closed chain (CHAIN_APPROX_SIMPLE drops collinear points only: same area).  Ties: the LAST component in raster order of
// first pixels (round 4) [EXT, restated from OpenCV's contours.cpp, unpinned: the border-following scanner discovers outer borders
// in raster order and cvInsertNodeIntoTree puts each new contour at the HEAD of its parent's child list, so the list findContours
// returns runs in REVERSE discovery order and np.argmax's "first maximum" (main.py:400-401) is the last one discovered].
// Returns {x0, y0, x1, y1, pixels}; false for an empty mask.
// Doubled shoelace area of the outer border of the component `inside` describes, walked from its first pixel in raster order
// (sy, sx) -- whose west / north neighbours are background -- by Moore neighbour tracing, clockwise with y pointing down; stops
// when the start pixel is left again in the first direction.  box = {x0, y0, x1, y1} of the border (= of the component).
template <typename Inside>
long long trace_outer_area2(Inside inside, int sy, int sx, long n_pixels, int (&box)[4])
{
    static const int dx8[8] = {1, 1, 0, -1, -1, -1, 0, 1}, dy8[8] = {0, 1, 1, 1, 0, -1, -1, -1};      // E, SE, S, SW, W, NW, N, NE
    long long area2 = 0;
    int cy = sy, cx = sx, back = 4, first_dir = -1;              // back: direction of the background pixel the search resumes after (W)
    box[0] = box[2] = sx; box[1] = box[3] = sy;
    for (long guard = 0; guard < 4 * n_pixels + 8; ++guard) {
        int d = -1;
        for (int k = 1; k <= 8; ++k) {                           // clockwise from the backtrack direction
            const int dd = (back + k) & 7;
            if (inside(cy + dy8[dd], cx + dx8[dd])) { d = dd; break; }
        }
        if (d < 0) break;                                        // isolated pixel: area 0
        if (cy == sy && cx == sx) {
            if (first_dir < 0) first_dir = d;
            else if (d == first_dir) break;                      // back at the start, leaving the same way: closed
        }
        const int ny = cy + dy8[d], nx = cx + dx8[d];
        area2 += (long long)cx * ny - (long long)nx * cy;
        cy = ny; cx = nx;
        box[0] = cx < box[0] ? cx : box[0]; box[2] = cx > box[2] ? cx : box[2];
        box[1] = cy < box[1] ? cy : box[1]; box[3] = cy > box[3] ? cy : box[3];
        // the neighbour examined just before (direction d - 1 from the old pixel) is background; seen from the new pixel it
        // lies in direction d + 6 (axis step) or d + 5 (diagonal step): the next search resumes right after it
        back = (d + ((d & 1) ? 5 : 6)) & 7;
    }
    return area2 < 0 ? -area2 : area2;
}

bool host_largest_contour(const uint8_t* m, int H, int W, int (&out)[5], long long* area2_out = nullptr)
{
    const long n = (long)H * W;
    std::vector<int> lab(n, -1);
    std::vector<long> stack;
    static const int dx8[8] = {1, 1, 0, -1, -1, -1, 0, 1}, dy8[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    long long best_area2 = -1;
    int n_comp = 0;
    for (long s = 0; s < n; ++s) {
        if (!m[s] || lab[s] >= 0) continue;
        // flood the component; box and pixel count on the way
        const int id = n_comp++;
        int x0 = W, y0 = H, x1 = -1, y1 = -1, cnt = 0;
        stack.clear();
        stack.push_back(s);
        lab[s] = id;
        while (!stack.empty()) {
            const long i = stack.back();
            stack.pop_back();
            const int y = (int)(i / W), x = (int)(i - (long)y * W);
            ++cnt;
            x0 = x < x0 ? x : x0; x1 = x > x1 ? x : x1; y0 = y < y0 ? y : y0; y1 = y > y1 ? y : y1;
            for (int d = 0; d < 8; ++d) {
                const int yy = y + dy8[d], xx = x + dx8[d];
                if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
                const long j = (long)yy * W + xx;
                if (m[j] && lab[j] < 0) { lab[j] = id; stack.push_back(j); }
            }
        }
        auto inside = [&](int y, int x) { return (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && lab[(long)y * W + x] == id; };
        int tb[4];
        const long long area2 = trace_outer_area2(inside, (int)(s / W), (int)(s - (long)(s / W) * W), n, tb);
        if (area2 >= best_area2) { best_area2 = area2; out[0] = x0; out[1] = y0; out[2] = x1; out[3] = y1; out[4] = cnt; }   // ties: the later one
    }
    if (area2_out) *area2_out = best_area2 < 0 ? 0 : best_area2;
    return n_comp > 0;
}

int check_ready(sbbseg_ctx* c)
{
    REQUIRE(c != nullptr, "null handle");
    REQUIRE(c->finalized, "plan not finalized");
    HIPCHK(hipSetDevice(c->device));
    return 0;
}

}  // namespace

// =================================================================================================
extern "C" {

const char* sbbseg_last_error(void) { return g_err.c_str(); }
int sbbseg_abi_version(void) { return SBBSEG_ABI_VERSION; }

int sbbseg_device_count(int* count)
{
    API_BEGIN
    REQUIRE(count, "null count");
    HIPCHK(hipGetDeviceCount(count));
    return 0;
    API_END
}

int sbbseg_create(int device, int precision, sbbseg_ctx** out)
{
    API_BEGIN
    REQUIRE(out, "null out");
    REQUIRE(precision == SBBSEG_PREC_BF16 || precision == SBBSEG_PREC_F32 || precision == SBBSEG_PREC_F16 || precision == SBBSEG_PREC_F16X3,
            "bad precision %d", precision);
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    REQUIRE(device >= 0 && device < ndev, "device %d out of range (have %d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
            "libsbbseg is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
    sbbseg_ctx* c = new sbbseg_ctx();
    c->device = device;
    c->precision = precision;
    c->elem = precision == SBBSEG_PREC_F32 ? 4 : 2;
    c->planes = is_split(precision) ? 2 : 1;
    c->num_cus = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail("hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    c->stream = c->own_stream;
    // The second lane's stream is created in ANOTHER PRIORITY CLASS than the handle's own stream.  HIP multiplexes streams onto a few
    // hardware queues per priority class (GPU_MAX_HW_QUEUES, default 4, dealt round robin): when a handle's two streams land on the
    // same queue its lanes serialise behind each other's barrier packets -- measured 26.4 ms instead of 22.9 ms per 108-tile page on the
    // SECOND handle of a process (the fourth with 8 queues, none of five with 16: tools/handle_order_probe.py, profiles/r04_experiments.md
    // section 10); configs[2]'s layout stage was that handle.  Queues of different priority classes are never shared.
    // SBBSEG_LANE_PRIORITY: -1 (default) = the device's highest priority, 0 = same class as the own stream (the old behaviour), 1 = lowest
    {
        int least = 0, greatest = 0, want = -1;
        if (const char* v = getenv("SBBSEG_LANE_PRIORITY")) want = atoi(v);
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
        const int prio = want < 0 ? greatest : want > 0 ? least : 0;
        e = prio != 0 ? hipStreamCreateWithPriority(&c->lane_stream, hipStreamNonBlocking, prio) : hipStreamCreateWithFlags(&c->lane_stream, hipStreamNonBlocking);
        c->lane_prio = prio; c->prio_least = least; c->prio_greatest = greatest;
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) {
        sbbseg_destroy(c);
        return fail("lane stream/event creation failed: %s", hipGetErrorString(e));
    }
    // CU-partitioned lanes: OPT-IN (SBBSEG_CU_SPLIT=1).  Measured (profiles/r04_experiments.md): the conv kernels of this library need
    // all 256 CUs to fill the HBM fabric -- on half the chip every op takes 1.5-1.8x as long (the streaming probe: 1.06x) -- so the
    // staggered halves come out 2 % (f16x3) to 5 % (f16) BEHIND the shared-chip lanes; kept for A/B, off by default
    const char* split_env = getenv("SBBSEG_CU_SPLIT");
    if (split_env && split_env[0] == '1' && c->num_cus >= 16) {
        const int words = (c->num_cus + 31) / 32, half = c->num_cus / 2;
        std::vector<uint32_t> lo(words, 0u), hi(words, 0u);
        for (int i = 0; i < c->num_cus; ++i) (i < half ? lo : hi)[i / 32] |= 1u << (i % 32);
        hipError_t e0 = hipExtStreamCreateWithCUMask(&c->half_stream[0], (uint32_t)words, lo.data());
        hipError_t e1 = e0 == hipSuccess ? hipExtStreamCreateWithCUMask(&c->half_stream[1], (uint32_t)words, hi.data()) : e0;
        if (e1 == hipSuccess) e1 = hipEventCreateWithFlags(&c->ev_half[0], hipEventDisableTiming);
        if (e1 == hipSuccess) e1 = hipEventCreateWithFlags(&c->ev_half[1], hipEventDisableTiming);
        if (e1 == hipSuccess) {
            c->cu_split = true;
            c->half_cus[0] = half; c->half_cus[1] = c->num_cus - half;
        } else {
            (void)hipGetLastError();
            for (int k = 0; k < 2; ++k) {
                if (c->half_stream[k]) (void)hipStreamDestroy(c->half_stream[k]);
                if (c->ev_half[k]) (void)hipEventDestroy(c->ev_half[k]);
                c->half_stream[k] = nullptr; c->ev_half[k] = nullptr;
            }
        }
    }
    if (const char* v = getenv("SBBSEG_STAGGER_MIN_TILES")) c->stagger_min_tiles = atoi(v);
    if (const char* v = getenv("SBBSEG_DEDUPE")) c->dedupe = v[0] != '0';
    if (const char* v = getenv("SBBSEG_KSPLIT")) c->ksplit = v[0] != '0';
    if (const char* v = getenv("SBBSEG_OWNED_REGIONS")) c->owned_mode = v[0] == '0' ? 0 : (v[0] == '2' ? 2 : 1);
    // experiment (round 6): persistent grids sized for this many CUs instead of the device's -- with two lanes, grids of half the chip let both
    // lanes' kernels run side by side for their whole duration (no CU masks: the dispatcher places the blocks)
    if (const char* v = getenv("SBBSEG_GRID_CUS")) { const int n = atoi(v); if (n >= 8 && n <= c->num_cus) c->num_cus = n & ~7; }
    *out = c;
    return 0;
    API_END
}

int sbbseg_destroy(sbbseg_ctx* c)
{
    API_BEGIN
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->lane_stream) (void)hipStreamSynchronize(c->lane_stream);
    for (int k = 0; k < 2; ++k)
        if (c->half_stream[k]) (void)hipStreamSynchronize(c->half_stream[k]);
    for (auto& t : c->tensors) {
        (void)hipFree(t.lane_buf[0]);
        (void)hipFree(t.lane_buf[1]);
    }
    std::vector<Op*> all_ops;
    for (auto& op : c->ops) {
        all_ops.push_back(&op);
        for (auto& part : op.parts) all_ops.push_back(&part);
    }
    for (Op* opp : all_ops) {
        Op& op = *opp;
        (void)hipFree(op.block.d_w1); (void)hipFree(op.block.d_w3);
        for (int q = 0; q < 4; ++q) (void)hipFree(op.conv.d_fgstep_cls[q]);
        (void)hipFree(op.conv.d_ktab); (void)hipFree(op.conv.d_kstep); (void)hipFree(op.conv.d_w); (void)hipFree(op.conv.d_scale); (void)hipFree(op.conv.d_shift);
        (void)hipFree(op.conv.d_rscale); (void)hipFree(op.conv.d_rshift);
        (void)hipFree(op.conv.d_head_w); (void)hipFree(op.conv.d_head_scale); (void)hipFree(op.conv.d_head_shift); (void)hipFree(op.conv.d_stem_wfrag); (void)hipFree(op.conv.d_d64_wfrag);
        (void)hipFree(op.conv.d_halo_wfrag); (void)hipFree(op.conv.d_halo_taps);
        (void)hipFree(op.conv.d_er_w3); (void)hipFree(op.conv.d_er_w1);
        (void)hipFree(op.conv.d_c3_w2); (void)hipFree(op.conv.d_c3_k0);
        for (int q = 1; q < 4; ++q) { (void)hipFree(op.conv.d_w_cls[q]); (void)hipFree(op.conv.d_kstep_cls[q]); (void)hipFree(op.conv.d_ktab_cls[q]); }
        (void)hipFree(op.head.d_w); (void)hipFree(op.head.d_scale); (void)hipFree(op.head.d_shift);
        (void)hipFree(op.pool.d_pre_scale); (void)hipFree(op.pool.d_pre_shift);
        (void)hipFree(op.tail.d_wfrag); (void)hipFree(op.tail.d_scale); (void)hipFree(op.tail.d_shift); (void)hipFree(op.tail.d_head_w);
        (void)hipFree(op.tail.d_head_scale); (void)hipFree(op.tail.d_head_shift);
    }
    (void)hipFree(c->d_lut); (void)hipFree(c->d_hist); (void)hipFree(c->d_tile_xy); (void)hipFree(c->d_batch_labels); (void)hipFree(c->d_probs); (void)hipFree(c->d_xin); (void)hipFree(c->d_ks_ws);
    (void)hipFree(c->d_page); (void)hipFree(c->d_page_labels); (void)hipFree(c->d_page_labels3); (void)hipFree(c->d_tile_labels);
    (void)hipFree(c->d_own_x); (void)hipFree(c->d_own_y); (void)hipFree(c->d_map); (void)hipFree(c->d_wmap);
    (void)hipFree(c->d_deskew);
    for (int lane = 0; lane < 2; ++lane)
        for (int L = 0; L < kRegionMaxLevels; ++L) (void)hipFree(c->d_rtab[lane][L]);
    (void)hipFree(c->d_run_page); (void)hipFree(c->d_run_mask); (void)hipFree(c->d_run_a); (void)hipFree(c->d_run_b);
    for (auto& ub : c->user_bufs) (void)hipFree(ub.first);
    for (int k = 0; k < 2; ++k) {
        (void)hipHostFree(c->pp_h_in[k]); (void)hipHostFree(c->pp_h_out[k]);
        (void)hipFree(c->pp_d_in[k]); (void)hipFree(c->pp_d_out[k]); (void)hipFree(c->pp_d_out3[k]);
        if (c->pp_in[k]) (void)hipEventDestroy(c->pp_in[k]);
        if (c->pp_comp[k]) (void)hipEventDestroy(c->pp_comp[k]);
        if (c->pp_out[k]) (void)hipEventDestroy(c->pp_out[k]);
    }
    comm_release(c);
    if (c->copy_in) (void)hipStreamDestroy(c->copy_in);
    if (c->copy_out) (void)hipStreamDestroy(c->copy_out);
    (void)hipFree(c->d_morph_a); (void)hipFree(c->d_morph_b); (void)hipFree(c->d_cc_parent); (void)hipFree(c->d_cc_count); (void)hipFree(c->d_cc_aux); (void)hipFree(c->d_cc_list); (void)hipFree(c->d_cc_small);
    for (auto& pe : c->pending) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
    for (auto e : c->free_events) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->lane_stream) (void)hipStreamDestroy(c->lane_stream);
    for (int k = 0; k < 2; ++k) {
        if (c->half_stream[k]) (void)hipStreamDestroy(c->half_stream[k]);
        if (c->ev_half[k]) (void)hipEventDestroy(c->ev_half[k]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    delete c;
    return 0;
    API_END
}

int sbbseg_set_stream(sbbseg_ctx* c, void* hip_stream)
{
    API_BEGIN
    REQUIRE(c, "null handle");
    if (resolve_pending(c)) return 1;
    // NULL is a real stream (the legacy default stream torch uses unless told otherwise)
    c->stream = hip_stream == SBBSEG_OWN_STREAM ? c->own_stream : (hipStream_t)hip_stream;
    // a caller's stream of the lane stream's own priority class could share its hardware queue (see sbbseg_create): move the lane
    // stream to the other end of the range
    int up = 0;
    if (hip_stream != SBBSEG_OWN_STREAM && hip_stream && c->lane_prio != 0 && c->prio_least != c->prio_greatest &&
        hipStreamGetPriority((hipStream_t)hip_stream, &up) == hipSuccess && up == c->lane_prio) {
        const int other = c->lane_prio == c->prio_greatest ? c->prio_least : c->prio_greatest;
        hipStream_t ns = nullptr;
        if (other != 0 && hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, other) == hipSuccess) {
            HIPCHK(hipStreamSynchronize(c->lane_stream));
            HIPCHK(hipStreamDestroy(c->lane_stream));
            c->lane_stream = ns;
            c->lane_prio = other;
        } else (void)hipGetLastError();
    }
    return 0;
    API_END
}

int sbbseg_set_lanes(sbbseg_ctx* c, int lanes)
{
    API_BEGIN
    REQUIRE(c && (lanes == 1 || lanes == 2), "lanes must be 1 or 2");
    REQUIRE(!(c->finalized && lanes == 2 && c->lane1_batch == 0), "the second lane was not allocated at finalize (lanes was 1 or max_batch < 16)");
    c->lanes = lanes;
    return 0;
    API_END
}

int sbbseg_set_label_channels(sbbseg_ctx* c, int channels)
{
    API_BEGIN
    REQUIRE(c && (channels == 1 || channels == 3), "label channels must be 1 or 3");
    c->label_channels = channels;
    return 0;
    API_END
}

int sbbseg_set_dedupe(sbbseg_ctx* c, int on)
{
    API_BEGIN
    REQUIRE(c && (on == 0 || on == 1), "dedupe must be 0 or 1");
    c->dedupe = on != 0;
    return 0;
    API_END
}

int sbbseg_set_owned_regions(sbbseg_ctx* c, int mode)
{
    API_BEGIN
    REQUIRE(c && mode >= 0 && mode <= 2, "mode: 0 = off, 1 = fused page paths, 2 = tile-range entry points too");
    c->owned_mode = mode;
    return 0;
    API_END
}

int sbbseg_owned_region_info(sbbseg_ctx* c, int* mode, int* levels)
{
    API_BEGIN
    REQUIRE(c, "null handle");
    if (mode) *mode = c->owned_mode;
    if (levels) *levels = c->finalized ? c->region_levels : 0;
    return 0;
    API_END
}

int sbbseg_debug_owned_range(int extent, int tile, int margin, int n_tiles, int t, int* lo, int* hi)
{
    API_BEGIN
    REQUIRE(lo && hi && tile > 2 * margin && margin >= 0 && extent >= tile && n_tiles >= 1 && t >= 0 && t < n_tiles, "bad arguments");
    const RegionAxis a = {extent, tile, margin, tile - 2 * margin, n_tiles};
    region_own(a, t, *lo, *hi);
    return 0;
    API_END
}

int sbbseg_debug_region_rows(int extent, int tile, int margin, int n_tiles, int t, int levels, const int32_t* level_size, int32_t* lo_hi)
{
    API_BEGIN
    REQUIRE(lo_hi && level_size && levels >= 1 && levels <= kRegionMaxLevels && tile > 2 * margin && margin >= 0 && extent >= tile && n_tiles >= 1 &&
            t >= 0 && t < n_tiles, "bad arguments");
    const RegionAxis a = {extent, tile, margin, tile - 2 * margin, n_tiles};
    int lo, hi;
    region_own(a, t, lo, hi);
    lo_hi[0] = lo; lo_hi[1] = hi;
    for (int k = 1; k < levels; ++k) {
        region_down(lo, hi, level_size[k]);
        lo_hi[2 * k] = lo; lo_hi[2 * k + 1] = hi;
    }
    return 0;
    API_END
}

int sbbseg_debug_poison_activations(sbbseg_ctx* c, int byte_value)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    HIPCHK(hipDeviceSynchronize());
    for (auto& t : c->tensors) {
        if (t.is_input_form) continue;                     // (their zero borders are part of the form)
        for (int lane = 0; lane < 2; ++lane) {
            if (!t.lane_buf[lane]) continue;
            const size_t n = t.elems_per_patch * (size_t)(lane == 0 ? c->max_batch : c->lane1_batch) * c->elem * c->planes;
            HIPCHK(hipMemset(t.lane_buf[lane] + kZeroHeaderBytes, byte_value & 255, n));
        }
    }
    if (c->d_tile_labels) HIPCHK(hipMemset(c->d_tile_labels, byte_value & 255, c->tile_labels_cap));
    HIPCHK(hipDeviceSynchronize());
    return 0;
    API_END
}

int sbbseg_set_ksplit(sbbseg_ctx* c, int on)
{
    API_BEGIN
    REQUIRE(c && (on == 0 || on == 1), "ksplit must be 0 or 1");
    c->ksplit = on != 0;
    return 0;
    API_END
}

int sbbseg_synchronize(sbbseg_ctx* c)
{
    API_BEGIN
    REQUIRE(c, "null handle");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
    API_END
}

// ---------------------------------------------------------------------------------- plan building
int sbbseg_set_input(sbbseg_ctx* c, int H, int W, int channels)
{
    API_BEGIN
    REQUIRE(c && !c->finalized, "bad handle / already finalized");
    REQUIRE(H > 0 && W > 0 && channels == 3, "input must be HxWx3 (got %dx%dx%d)", H, W, channels);
    c->in_H = H; c->in_W = W; c->in_C = channels;
    return 0;
    API_END
}

int sbbseg_input_form(sbbseg_ctx* c, int form, int pad, int* tensor_id)
{
    API_BEGIN
    REQUIRE(c && !c->finalized && tensor_id, "bad handle / already finalized");
    REQUIRE(c->in_H > 0, "sbbseg_set_input first");
    REQUIRE(form == SBBSEG_INPUT_C8 || form == SBBSEG_INPUT_PAIRS, "unknown input form %d", form);
    if (c->form_tensor[form] >= 0) {
        REQUIRE(c->tensors[c->form_tensor[form]].pad == pad, "input form %d requested with different pad", form);
        *tensor_id = c->form_tensor[form];
        return 0;
    }
    Tensor t;
    t.is_input_form = true; t.form = form; t.pad = pad; t.C = 8;
    if (form == SBBSEG_INPUT_C8) {
        REQUIRE(pad == 0, "C8 form takes pad 0");
        t.H = c->in_H; t.W = c->in_W;
    } else {
        REQUIRE(pad >= 0, "negative pad");
        t.H = c->in_H + 2 * pad; t.W = (c->in_W + 2 * pad + 1) / 2;
    }
    t.elems_per_patch = (size_t)t.H * t.W * t.C;
    c->tensors.push_back(t);
    c->form_tensor[form] = (int)c->tensors.size() - 1;
    *tensor_id = c->form_tensor[form];
    return 0;
    API_END
}

int sbbseg_add_tensor(sbbseg_ctx* c, int H, int W, int C, int* tensor_id)
{
    API_BEGIN
    REQUIRE(c && !c->finalized && tensor_id, "bad handle / already finalized");
    REQUIRE(H > 0 && W > 0 && C > 0 && C % 8 == 0, "tensor %dx%dx%d: channels must be a positive multiple of 8", H, W, C);
    Tensor t;
    t.H = H; t.W = W; t.C = C;
    t.elems_per_patch = (size_t)H * W * C;
    c->tensors.push_back(t);
    *tensor_id = (int)c->tensors.size() - 1;
    return 0;
    API_END
}

int sbbseg_add_conv(sbbseg_ctx* c, const sbbseg_conv_desc* d, const float* w_src0, const float* w_src1,
                    const float* scale, const float* shift, const float* raw_scale, const float* raw_shift,
                    const float* head_w, const float* head_scale, const float* head_shift)
{
    API_BEGIN
    REQUIRE(c && !c->finalized && d && w_src0 && scale && shift, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    REQUIRE(d->n_src == 1 || d->n_src == 2, "n_src must be 1 or 2");
    REQUIRE(d->n_src == 1 || w_src1, "second source needs its weights");
    REQUIRE(d->cout > 0 && d->cout % 8 == 0, "cout %d must be a positive multiple of 8", d->cout);
    REQUIRE(d->out_h > 0 && d->out_w > 0 && d->out_stride_y >= 1 && d->out_stride_x >= 1 && d->out_off_y >= 0 &&
            d->out_off_x >= 0, "bad output grid / placement");
    const int ntens = (int)c->tensors.size();
    for (int s = 0; s < d->n_src; ++s) {
        const sbbseg_conv_src& cs = d->src[s];
        REQUIRE(cs.tensor >= 0 && cs.tensor < ntens, "conv source tensor %d undefined", cs.tensor);
        const Tensor& t = c->tensors[cs.tensor];
        REQUIRE(cs.channels > 0 && ((cs.channels + 7) / 8) * 8 <= t.C, "conv source takes %d channels of a %d-channel tensor", cs.channels, t.C);
        REQUIRE(cs.kh > 0 && cs.kw > 0, "bad kernel size");
        REQUIRE((cs.stride_y == 1 || cs.stride_y == 2) && (cs.stride_x == 1 || cs.stride_x == 2), "strides must be 1 or 2");
        REQUIRE(cs.up_shift == 0 || cs.up_shift == 1, "up_shift must be 0 or 1");
        REQUIRE(!(cs.up_shift && (cs.off_y || cs.off_x)), "offset and upsampling cannot be combined");
        // the last window must start inside the logical input
        const int lh = (t.H << cs.up_shift) + cs.off_y, lw = (t.W << cs.up_shift) + cs.off_x;
        REQUIRE((d->out_h - 1) * cs.stride_y - cs.pad_top < lh && (d->out_w - 1) * cs.stride_x - cs.pad_left < lw,
                "conv geometry: output grid %dx%d does not fit source %d (%dx%d logical)", d->out_h, d->out_w, s, lh, lw);
    }
    REQUIRE(d->out_tensor >= 0 || d->raw_out_tensor >= 0 || d->head_classes > 0, "conv without outputs");
    int TH = -1, TW = -1;
    for (int which = 0; which < 3; ++which) {
        const int id = which == 0 ? d->out_tensor : which == 1 ? d->raw_out_tensor : d->residual_tensor;
        if (id < 0) continue;
        REQUIRE(id < ntens, "conv output/residual tensor %d undefined", id);
        const Tensor& t = c->tensors[id];
        REQUIRE(t.C == d->cout, "conv output tensor has %d channels, cout is %d", t.C, d->cout);
        if (TH < 0) { TH = t.H; TW = t.W; }
        REQUIRE(t.H == TH && t.W == TW, "conv outputs differ in size");
    }
    if (d->head_classes > 0) {
        REQUIRE(c->precision != kF32, "fused head is a 16-bit-mode feature (the fp32 check path runs the head as its own op)");
        REQUIRE(d->cout == 32 && d->head_classes <= 4 && head_w && head_scale && head_shift, "fused head needs cout == 32, <= 4 classes and its weights");
        // several convs may carry the same head (the parity classes of the last decoder conv)
        REQUIRE(c->classes == 0 || (c->fused_heads > 0 && c->classes == d->head_classes), "plan already has a head");
        if (TH < 0) { TH = c->in_H; TW = c->in_W; }
        REQUIRE(TH == c->in_H && TW == c->in_W, "fused head runs at input resolution");
    }
    REQUIRE((d->out_h - 1) * d->out_stride_y + d->out_off_y < TH && (d->out_w - 1) * d->out_stride_x + d->out_off_x < TW,
            "output placement leaves the %dx%d tensor", TH, TW);

    Op op;
    op.type = kConv;
    ConvOp& co = op.conv;
    co.d = *d;
    co.Ho = d->out_h; co.Wo = d->out_w; co.TH = TH; co.TW = TW;
    const int bc = c->precision != kF32 ? 256 : 4;          // pad weight rows for the widest channel tile any variant uses
    co.cout_pad = ((d->cout + bc - 1) / bc) * bc;

    // contraction order: source-major, then 64-channel group, then tap (ky,kx), then the group's
    // 8-channel granules.  Keeping the taps of one channel group ADJACENT makes the shifted re-reads
    // of the same pixel rows hit in L2 (measured with tap-outer order: dec1 fetched 2.5 GB per
    // launch for 70 MB of input).  Each source's segment is padded to whole K-steps (64) with
    // out-of-bounds ("zero") granules.  Tap offsets carry the source's padding and placement offset.
    // Split mode (kF16X3): a K-step is 32 channels (4 granules) of one tap; its slots 0-3 are those granules' "hi"
    // halves, slots 4-7 the "lo" halves of the same channels (lo plane of the stored pixel, lo half of the weight).
    alloc_check();
    const bool split = is_split(c->precision);
    const int gps = split ? 4 : kGranulesPerStep;            // channel granules per K-step
    std::vector<KTabEntry> lin;                              // granule list in contraction order (hi halves in split mode)
    struct KRef { int s, ky, kx, c0; };
    std::vector<KRef> lref;
    double geo_macs = 0;
    for (int s = 0; s < d->n_src; ++s) {
        const sbbseg_conv_src& cs = d->src[s];
        const int g8 = (cs.channels + 7) / 8;
        int granules = 0;
        // Tap order inside a channel group.  A 3x3 stride-2 source (the skip tensor of a parity-split decoder conv) is walked parity
        // set by parity set -- (0,0) (0,2) (2,0) (2,2) | (0,1) (2,1) | (1,0) (1,2) | (1,1): taps of one set read the SAME source pixels
        // (shifted by one output step), so their K-steps, now adjacent, find the lines of the previous step in L2; in row-major
        // order the next touch of a line came 2 or 6 K-steps later, after 4-12 MB of other gathers had passed through the XCD's
        // 4 MB L2 (PMC: dec4 fetched 4.4x its input).  SBBSEG_TAP_ORDER=0: row-major (A/B).  Only the order of the sum changes.
        std::vector<int> tap_order;
        static const bool grouped_taps = !(getenv("SBBSEG_TAP_ORDER") && getenv("SBBSEG_TAP_ORDER")[0] == '0');
        if (grouped_taps && cs.kh == 3 && cs.kw == 3 && cs.stride_y == 2 && cs.stride_x == 2) tap_order = {0, 2, 6, 8, 1, 7, 3, 5, 4};
        else
            for (int t = 0; t < cs.kh * cs.kw; ++t) tap_order.push_back(t);
        for (int cg = 0; cg < g8; cg += gps)
            for (int ti = 0; ti < cs.kh * cs.kw; ++ti)
                {
                    const int ky = tap_order[ti] / cs.kw, kx = tap_order[ti] % cs.kw;
                    for (int g = cg; g < g8 && g < cg + gps; ++g) {
                        KTabEntry e;
                        e.dy = (int16_t)(ky - cs.pad_top - cs.off_y);
                        e.dx = (int16_t)(kx - cs.pad_left - cs.off_x);
                        // byte offset of the granule's 8 channels inside the stored pixel (split mode: of their hi halves, in the
                        // interleaved [group hi | group lo] layout, internal.h)
                        e.coff = split ? split_hi_elem(c->tensors[cs.tensor].C, g * 8) * c->elem : g * 8 * c->elem;
                        lin.push_back(e);
                        lref.push_back({s, ky, kx, g * 8});
                        ++granules;
                    }
                }
        const int ks = (granules + gps - 1) / gps;
        for (int g = granules; g < ks * gps; ++g) {
            KTabEntry e;
            e.dy = 16000; e.dx = 0; e.coff = 0;
            lin.push_back(e);
            lref.push_back({-1, 0, 0, 0});
        }
        co.ksteps[s] = ks;
        co.total_ksteps += ks;
        geo_macs += (double)cs.kh * cs.kw * cs.channels;
    }
    co.Ktot = co.total_ksteps * kBK;
    REQUIRE((size_t)co.cout_pad * co.Ktot * c->elem < (size_t)3 << 30, "weight matrix too large");
    // slot table: 8 entries per K-step (what the kernels index); kpart = 0 plain / hi half, 1 lo half
    std::vector<KTabEntry> ktab((size_t)co.total_ksteps * kGranulesPerStep);
    std::vector<KRef> kref(ktab.size());
    std::vector<int> kpart(ktab.size(), 0);
    for (int t = 0; t < co.total_ksteps; ++t)
        for (int g = 0; g < kGranulesPerStep; ++g) {
            const size_t li = (size_t)t * gps + (split ? (g & 3) : g), ki = (size_t)t * kGranulesPerStep + g;
            ktab[ki] = lin[li];
            kref[ki] = lref[li];
            if (split && g >= 4) {
                kpart[ki] = 1;
                if (lref[li].s >= 0) ktab[ki].coff += split_group(c->tensors[d->src[lref[li].s].tensor].C) * c->elem;      // the group's lo halves
            }
        }

    // pack weights [cout_pad][Ktot]: one source pointer per K element (null = K padding), then row by row in
    // blocks of 64 K elements -- writes are contiguous, the 64 source lines of a block stay in cache across
    // neighbouring output channels (the column-by-column form of this loop took 0.6 s of a model's load time)
    const size_t wn = (size_t)co.cout_pad * co.Ktot;
    std::vector<const float*> ksrc((size_t)co.Ktot, nullptr);
    std::vector<uint8_t> klo((size_t)co.Ktot, 0);
    float wmax = 0.f;
    for (size_t g = 0; g < kref.size(); ++g) {
        const KRef& r = kref[g];
        if (r.s < 0) continue;
        const sbbseg_conv_src& cs = d->src[r.s];
        const float* wsrc_base = r.s == 0 ? w_src0 : w_src1;
        for (int q = 0; q < 8; ++q) {
            const int ch = r.c0 + q;
            if (ch < cs.channels) {
                ksrc[g * 8 + q] = wsrc_base + ((size_t)(r.ky * cs.kw + r.kx) * cs.channels + ch) * d->cout;
                klo[g * 8 + q] = (uint8_t)kpart[g];
            }
        }
    }
    // split mode: one power-of-two pre-scale per conv brings the largest |w| into [256, 512), so that the lo halves of
    // all weights within 2^12 of it are normal fp16 numbers; the epilogue multiplies `scale` by 2^-s (exact)
    float wpre = 1.f;
    if (split) {
        for (int s2 = 0; s2 < d->n_src; ++s2) {
            const sbbseg_conv_src& cs = d->src[s2];
            const float* wp = s2 == 0 ? w_src0 : w_src1;
            const size_t cnt = (size_t)cs.kh * cs.kw * cs.channels * d->cout;
            for (size_t i = 0; i < cnt; ++i) wmax = std::fmax(wmax, std::fabs(wp[i]));
        }
        REQUIRE(std::isfinite(wmax), "%s: non-finite weights", "conv");
        if (wmax > 0.f) {
            int ex = 0;
            (void)std::frexp(wmax, &ex);                     // wmax = m * 2^ex, m in [0.5, 1)
            int sexp = 9 - ex;                               // wmax * 2^sexp in [256, 512)
            sexp = sexp > 60 ? 60 : (sexp < -60 ? -60 : sexp);
            wpre = std::ldexp(1.f, sexp);
        }
        co.wmul_cls[0] = 1.f / wpre;
    }
    std::vector<int> row_ch(co.cout_pad);
    for (int row = 0; row < co.cout_pad; ++row) row_ch[row] = c->precision != kF32 ? conv_row_channel(row, d->cout) : row;
    auto pack = [&](auto* dst, auto conv) {
        for (int kb = 0; kb < co.Ktot; kb += kBK) {
            const float* const* ks = &ksrc[kb];
            const uint8_t* kl = &klo[kb];
            for (int row = 0; row < co.cout_pad; ++row) {
                const int o = row_ch[row];
                auto* out = dst + (size_t)row * co.Ktot + kb;
                if (o >= d->cout) { for (int k = 0; k < kBK; ++k) out[k] = conv(0.f, 0); continue; }
                for (int k = 0; k < kBK; ++k) out[k] = conv(ks[k] ? ks[k][o] : 0.f, (int)kl[k]);
            }
        }
    };
    if (c->precision != kF32) {
        std::vector<uint16_t> wb(wn);
        if (split)
            pack(wb.data(), [wpre](float v, int part) {
                const float sv = v * wpre;                   // exact (power of two)
                const uint16_t hb = f32_to_f16_rne(sv);
                if (!part) return hb;
                const _Float16 h = __builtin_bit_cast(_Float16, hb);
                return f32_to_f16_rne(sv - (float)h);
            });
        else if (c->precision == kF16) pack(wb.data(), [](float v, int) { return f32_to_f16_rne(v); });
        else pack(wb.data(), [](float v, int) { return f32_to_bf16_rne(v); });
        if (upload(c, (uint16_t**)&co.d_w, wb.data(), wn)) return 1;
    } else {
        std::vector<float> wf(wn);
        pack(wf.data(), [](float v, int) { return v; });
        if (upload(c, (float**)&co.d_w, wf.data(), wn)) return 1;
    }
    if (upload(c, &co.d_ktab, ktab.data(), ktab.size())) return 1;
    std::vector<KStepRec> ksteps(co.total_ksteps);
    for (int t = 0; t < co.total_ksteps; ++t) {
        const KTabEntry* e = &ktab[(size_t)t * kGranulesPerStep];
        KStepRec r;
        r.dy = e[0].dy; r.dx = e[0].dx; r.coff = e[0].coff; r.irregular = 0; r.pad_ = 0;
        for (int g = 1; g < kGranulesPerStep; ++g) {
            // regular: one tap, channel-consecutive granules; in split mode slots 4-7 are slots 0-3 moved to the lo plane
            // (same distance for every step of a source: SrcDesc::lo_off)
            int want = e[0].coff + 16 * g;
            if (split) {
                const KRef& r0 = kref[(size_t)t * kGranulesPerStep];
                const int lo_off = r0.s >= 0 ? split_group(c->tensors[d->src[r0.s].tensor].C) * c->elem : 0;
                want = e[0].coff + 16 * (g & 3) + (g >> 2) * lo_off;
                if (r0.s < 0 || kref[(size_t)t * kGranulesPerStep + g].s != r0.s) r.irregular = 1;
            }
            if (e[g].dy != e[0].dy || e[g].dx != e[0].dx || e[g].coff != want) r.irregular = 1;
        }
        if (c->precision == kF32) r.irregular = 1;     // the fp32 check kernel only walks the granule table
        if (r.irregular) co.fg_ok = false;
        {
            const int sidx = t < co.ksteps[0] ? 0 : 1;
            co.tap_lo[sidx][0] = std::min(co.tap_lo[sidx][0], (int)r.dy); co.tap_hi[sidx][0] = std::max(co.tap_hi[sidx][0], (int)r.dy);
            co.tap_lo[sidx][1] = std::min(co.tap_lo[sidx][1], (int)r.dx); co.tap_hi[sidx][1] = std::max(co.tap_hi[sidx][1], (int)r.dx);
        }
        ksteps[t] = r;
    }
    if (upload(c, &co.d_kstep, ksteps.data(), ksteps.size())) return 1;
    co.h_ksteps_cls[0] = ksteps;
    std::vector<float> pad_s(co.cout_pad, 0.f), pad_b(co.cout_pad, 0.f);
    memcpy(pad_s.data(), scale, sizeof(float) * d->cout);
    memcpy(pad_b.data(), shift, sizeof(float) * d->cout);
    if (upload(c, &co.d_scale, pad_s.data(), pad_s.size()) || upload(c, &co.d_shift, pad_b.data(), pad_b.size())) return 1;
    if (d->raw_out_tensor >= 0) {
        REQUIRE(raw_scale && raw_shift, "raw output needs raw_scale/raw_shift");
        memcpy(pad_s.data(), raw_scale, sizeof(float) * d->cout);
        memcpy(pad_b.data(), raw_shift, sizeof(float) * d->cout);
        if (upload(c, &co.d_rscale, pad_s.data(), pad_s.size()) || upload(c, &co.d_rshift, pad_b.data(), pad_b.size())) return 1;
    }
    co.h_epi.assign(scale, scale + d->cout);
    co.h_epi.insert(co.h_epi.end(), shift, shift + d->cout);
    if (d->head_classes > 0) {
        co.h_epi.insert(co.h_epi.end(), head_w, head_w + (size_t)d->cout * d->head_classes);
        co.h_epi.insert(co.h_epi.end(), head_scale, head_scale + d->head_classes);
        co.h_epi.insert(co.h_epi.end(), head_shift, head_shift + d->head_classes);
    }
    if (d->head_classes > 0) {
        if (upload(c, &co.d_head_w, head_w, (size_t)d->cout * d->head_classes) ||
            upload(c, &co.d_head_scale, head_scale, d->head_classes) || upload(c, &co.d_head_shift, head_shift, d->head_classes))
            return 1;
        c->classes = d->head_classes;
        c->fused_heads += 1;
    }
    int cin_total = 0;
    for (int s = 0; s < d->n_src; ++s) cin_total += d->src[s].channels;
    char nm[128];
    snprintf(nm, sizeof(nm), "conv%dx%d_c%dto%d_%dx%d%s%s%s%s", d->src[0].kh, d->src[0].kw, cin_total, d->cout, d->out_h, d->out_w,
             d->n_src == 2 ? "_cat" : "", d->src[0].up_shift ? "_up" : "",
             (d->out_stride_y > 1 || d->out_stride_x > 1) ? (std::string("_par") + char('0' + d->out_off_y) + char('0' + d->out_off_x)).c_str() : "",
             d->head_classes > 0 ? "_head" : "");
    op.name = nm;
    const double macs = d->algorithmic_macs > 0 ? d->algorithmic_macs : (double)d->out_h * d->out_w * d->cout * geo_macs;
    op.flops = 2.0 * macs;
    op.issued_flops = 2.0 * d->out_h * d->out_w * d->cout * (double)co.total_ksteps * (split ? 32 * 3 : kBK);
    double bytes = 0;
    for (int s = 0; s < d->n_src; ++s) {
        const Tensor& t = c->tensors[d->src[s].tensor];
        bytes += (double)t.H * t.W * ((d->src[s].channels + 7) / 8 * 8) * c->elem * c->planes / (d->out_stride_y * d->out_stride_x);
    }
    const double ob = (double)d->out_h * d->out_w * d->cout * c->elem * c->planes;
    bytes += (d->out_tensor >= 0 ? ob : 0) + (d->raw_out_tensor >= 0 ? ob : 0) + (d->residual_tensor >= 0 ? ob : 0);
    op.min_bytes = bytes;
    co.d_w_cls[0] = co.d_w; co.d_kstep_cls[0] = co.d_kstep; co.d_ktab_cls[0] = co.d_ktab;
    co.ooy_cls[0] = d->out_off_y; co.oox_cls[0] = d->out_off_x;

    // the network stem (7 rows x 4 two-pixel granules on the PAIRS form -> 64 channels) has its own kernel
    {
        const sbbseg_conv_src& cs = d->src[0];
        const Tensor& st = c->tensors[cs.tensor];
        const char* env = getenv("SBBSEG_STEM_KERNEL");
        const bool plain16 = c->precision == kF16 || c->precision == kBF16;     // the dedicated kernels read the one-plane layout
        if ((plain16 || split) && !(env && env[0] == '0') && d->n_src == 1 && st.is_input_form && st.form == SBBSEG_INPUT_PAIRS &&
            cs.channels == 8 && cs.kh == 7 && cs.kw == 4 && cs.stride_y == 2 && cs.stride_x == 1 && cs.pad_top == 0 && cs.pad_left == 0 &&
            cs.up_shift == 0 && cs.off_y == 0 && cs.off_x == 0 && d->cout == 64 && d->out_h % 16 == 0 && d->out_w % 16 == 0 &&
            st.H >= 2 * d->out_h + 5 && st.W >= d->out_w + 3 &&
            d->residual_tensor < 0 && d->raw_out_tensor < 0 && d->head_classes == 0 && d->out_tensor >= 0 && d->out_stride_y == 1 &&
            d->out_stride_x == 1 && d->out_off_y == 0 && d->out_off_x == 0 && TH == d->out_h && TW == d->out_w) {
            const size_t ssz = (size_t)7 * 4 * 64 * 8;
            std::vector<uint16_t> frag(ssz * (split ? 2 : 1));
            for (int ky = 0; ky < 7; ++ky)
                for (int mi = 0; mi < 4; ++mi)
                    for (int l = 0; l < 64; ++l) {
                        const int o = conv_row_channel(mi * 16 + (l & 15), 64);
                        const int g = l >> 4;                                   // two-pixel granule = kernel column pair
                        for (int e = 0; e < 8; ++e) {
                            const float v = w_src0[(((size_t)ky * 4 + g) * 8 + e) * 64 + o];
                            const size_t at = ((((size_t)ky * 4 + mi) * 64) + l) * 8 + e;
                            if (split) {                     // hi | lo of the pre-scaled weight (the conv's wpre, see above)
                                const float sv = v * wpre;
                                const uint16_t hb = f32_to_f16_rne(sv);
                                frag[at] = hb;
                                frag[ssz + at] = f32_to_f16_rne(sv - (float)__builtin_bit_cast(_Float16, hb));
                            } else {
                                frag[at] = c->precision == kF16 ? f32_to_f16_rne(v) : f32_to_bf16_rne(v);
                            }
                        }
                    }
            if (upload(c, &co.d_stem_wfrag, frag.data(), frag.size())) return 1;
            op.name = "stem_" + op.name;
        }
        // 3x3 / stride 1 / pad 1, 64 -> 64 channels: direct conv on an LDS halo tile, weights in registers
        const char* env2 = getenv("SBBSEG_DIRECT64_KERNEL");
        if ((plain16 || split) && !(env2 && env2[0] == '0') && d->n_src == 1 && !st.is_input_form && st.C == 64 && cs.channels == 64 &&
            cs.kh == 3 && cs.kw == 3 && cs.stride_y == 1 && cs.stride_x == 1 && cs.pad_top == 1 && cs.pad_left == 1 && cs.up_shift == 0 &&
            cs.off_y == 0 && cs.off_x == 0 && d->cout == 64 && d->out_h == st.H && d->out_w == st.W && d->residual_tensor < 0 &&
            d->raw_out_tensor < 0 && d->head_classes == 0 && d->out_tensor >= 0 && d->out_stride_y == 1 && d->out_stride_x == 1 &&
            d->out_off_y == 0 && d->out_off_x == 0 && TH == d->out_h && TW == d->out_w) {
            const size_t fsz = (size_t)9 * 2 * 4 * 64 * 8;
            std::vector<uint16_t> frag(fsz * (split ? 2 : 1));
            for (int t = 0; t < 9; ++t)
                for (int kk = 0; kk < 2; ++kk)
                    for (int mi = 0; mi < 4; ++mi)
                        for (int l = 0; l < 64; ++l) {
                            const int o = conv_row_channel(mi * 16 + (l & 15), 64);
                            const int chb = (kk * 4 + (l >> 4)) * 8;
                            for (int e = 0; e < 8; ++e) {
                                const float v = w_src0[((size_t)t * 64 + chb + e) * 64 + o];
                                const size_t at = (((((size_t)t * 2 + kk) * 4 + mi) * 64) + l) * 8 + e;
                                if (split) {                     // hi | lo of the pre-scaled weight (the conv's wpre, see above)
                                    const float sv = v * wpre;
                                    const uint16_t hb = f32_to_f16_rne(sv);
                                    frag[at] = hb;
                                    frag[fsz + at] = f32_to_f16_rne(sv - (float)__builtin_bit_cast(_Float16, hb));
                                } else {
                                    frag[at] = c->precision == kF16 ? f32_to_f16_rne(v) : f32_to_bf16_rne(v);
                                }
                            }
                        }
            if (upload(c, &co.d_d64_wfrag, frag.data(), frag.size())) return 1;
            op.name = "direct_" + op.name;
        }
        // small pointwise convs keep their weights on the host until sbbseg_finalize: candidates for bottleneck fusion
        bool pw = (plain16 || split) && d->head_classes == 0 && d->raw_out_tensor < 0 && (d->cout == 64 || d->cout == 256) && d->out_stride_y == 1 &&
                  d->out_stride_x == 1 && d->out_off_y == 0 && d->out_off_x == 0 && TH == d->out_h && TW == d->out_w;
        for (int s2 = 0; pw && s2 < d->n_src; ++s2) {
            const sbbseg_conv_src& q = d->src[s2];
            const Tensor& qt = c->tensors[q.tensor];
            pw = q.kh == 1 && q.kw == 1 && q.stride_y == 1 && q.stride_x == 1 && q.pad_top == 0 && q.pad_left == 0 && q.up_shift == 0 &&
                 q.off_y == 0 && q.off_x == 0 && (q.channels == 64 || q.channels == 256) && q.channels == qt.C && !qt.is_input_form &&
                 qt.H == d->out_h && qt.W == d->out_w;
        }
        if (pw) {
            co.h_w[0].assign(w_src0, w_src0 + (size_t)d->src[0].channels * d->cout);
            if (d->n_src == 2) co.h_w[1].assign(w_src1, w_src1 + (size_t)d->src[1].channels * d->cout);
        }
    }

    // Output-placement siblings (same sources, taps geometry and outputs, only padding / placement
    // offset / weights differ -- the parity classes of one decoder conv) run as ONE launch: bigger
    // grids (the 256x256 tiles become usable on the small-M layers) and 4x fewer launches.
    if (c->precision != kF32 && !c->ops.empty() && c->ops.back().type == kConv && !(c->conv_variant & 16)) {
        Op& prev = c->ops.back();
        ConvOp& pc = prev.conv;
        bool same = pc.n_cls < 4 && pc.d.n_src == d->n_src && pc.d.cout == d->cout && pc.d.out_tensor == d->out_tensor &&
                    pc.d.relu == d->relu && pc.d.residual_tensor < 0 && d->residual_tensor < 0 && pc.d.raw_out_tensor < 0 &&
                    d->raw_out_tensor < 0 && pc.d.out_h == d->out_h && pc.d.out_w == d->out_w &&
                    pc.d.out_stride_y == d->out_stride_y && pc.d.out_stride_x == d->out_stride_x &&
                    (d->out_stride_y > 1 || d->out_stride_x > 1) && pc.d.head_classes == d->head_classes &&
                    pc.total_ksteps == co.total_ksteps && pc.ksteps[0] == co.ksteps[0] &&
                    pc.h_epi == co.h_epi;       // the merged launch applies class 0's BN / head constants to every class
        for (int s = 0; same && s < d->n_src; ++s) {
            const sbbseg_conv_src &x = pc.d.src[s], &y = d->src[s];
            same = x.tensor == y.tensor && x.channels == y.channels && x.kh == y.kh && x.kw == y.kw && x.stride_y == y.stride_y &&
                   x.stride_x == y.stride_x && x.up_shift == y.up_shift && x.off_y == y.off_y && x.off_x == y.off_x;
        }
        if (same) {
            const int q = pc.n_cls++;
            pc.fg_ok = pc.fg_ok && co.fg_ok;
            pc.h_ksteps_cls[q] = co.h_ksteps_cls[0];
            for (int s2 = 0; s2 < 2; ++s2)
                for (int a = 0; a < 2; ++a) {
                    pc.tap_lo[s2][a] = std::min(pc.tap_lo[s2][a], co.tap_lo[s2][a]);
                    pc.tap_hi[s2][a] = std::max(pc.tap_hi[s2][a], co.tap_hi[s2][a]);
                }
            pc.d_w_cls[q] = co.d_w; pc.d_kstep_cls[q] = co.d_kstep; pc.d_ktab_cls[q] = co.d_ktab;
            pc.ooy_cls[q] = d->out_off_y; pc.oox_cls[q] = d->out_off_x; pc.wmul_cls[q] = co.wmul_cls[0];
            (void)hipFree(co.d_scale); (void)hipFree(co.d_shift); (void)hipFree(co.d_head_w); (void)hipFree(co.d_head_scale); (void)hipFree(co.d_head_shift);
            c->device_bytes -= 2 * sizeof(float) * co.cout_pad;
            if (d->head_classes > 0) c->device_bytes -= sizeof(float) * ((size_t)d->cout * d->head_classes + 2 * d->head_classes);
            prev.flops += op.flops;
            prev.issued_flops += op.issued_flops;
            prev.min_bytes += op.min_bytes;
            const size_t pos = prev.name.find("_par");
            if (pos != std::string::npos) prev.name = prev.name.substr(0, pos) + "_par4" + (d->head_classes > 0 ? "_head" : "");
            return 0;
        }
    }
    c->ops.push_back(op);
    return 0;
    API_END
}

int sbbseg_add_maxpool(sbbseg_ctx* c, int src_tensor, int dst_tensor, int k, int stride, const float* pre_scale,
                       const float* pre_shift, int pre_relu)
{
    API_BEGIN
    REQUIRE(c && !c->finalized, "bad handle / already finalized");
    const int ntens = (int)c->tensors.size();
    REQUIRE(src_tensor >= 0 && src_tensor < ntens && dst_tensor >= 0 && dst_tensor < ntens, "maxpool tensors undefined");
    const Tensor& s = c->tensors[src_tensor];
    const Tensor& t = c->tensors[dst_tensor];
    const int Ho = (s.H - k) / stride + 1, Wo = (s.W - k) / stride + 1;
    REQUIRE(t.H == Ho && t.W == Wo && t.C == s.C, "maxpool output should be %dx%dx%d", Ho, Wo, s.C);
    Op op;
    op.type = kPool;
    op.pool.src = src_tensor; op.pool.dst = dst_tensor; op.pool.k = k; op.pool.stride = stride; op.pool.Ho = Ho; op.pool.Wo = Wo;
    if (pre_scale) {
        REQUIRE(pre_shift, "pre_scale needs pre_shift");
        HIPCHK(hipSetDevice(c->device));
        if (upload(c, &op.pool.d_pre_scale, pre_scale, s.C) || upload(c, &op.pool.d_pre_shift, pre_shift, s.C)) return 1;
        op.pool.pre_relu = pre_relu;
    }
    char nm[64];
    snprintf(nm, sizeof(nm), "maxpool%dx%d_s%d_c%d_%dx%d", k, k, stride, s.C, Ho, Wo);
    op.name = nm;
    op.flops = 0;
    op.min_bytes = ((double)s.H * s.W + (double)Ho * Wo) * s.C * c->elem * c->planes;
    c->ops.push_back(op);
    return 0;
    API_END
}

int sbbseg_add_tail(sbbseg_ctx* c, int src0_tensor, int img_c8_tensor, const float* w_src0, const float* w_img,
                    const float* scale, const float* shift, int classes, const float* head_w, const float* head_scale,
                    const float* head_shift, double algorithmic_macs)
{
    API_BEGIN
    REQUIRE(c && !c->finalized && w_src0 && w_img && scale && shift && head_w && head_scale && head_shift, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    REQUIRE(c->precision == kF16 || c->precision == kBF16 || c->precision == kF16X3, "the fused tail is a 16-bit-mode kernel");
    const bool split = c->precision == kF16X3;
    const int ntens = (int)c->tensors.size();
    REQUIRE(src0_tensor >= 0 && src0_tensor < ntens && img_c8_tensor >= 0 && img_c8_tensor < ntens, "tail tensors undefined");
    const Tensor& s0 = c->tensors[src0_tensor];
    const Tensor& im = c->tensors[img_c8_tensor];
    REQUIRE(s0.C == 64, "fused tail needs a 64-channel upsampled source (got %d)", s0.C);
    REQUIRE(im.is_input_form && im.form == SBBSEG_INPUT_C8, "fused tail needs the C8 input form as second source");
    REQUIRE(2 * s0.H == c->in_H && 2 * s0.W == c->in_W && c->in_H % 16 == 0 && c->in_W % 16 == 0, "fused tail: geometry");
    REQUIRE(classes >= 1 && classes <= 4 && c->classes == 0, "fused tail: 1..4 classes, one head per plan");
    static const int taps[2][2][2] = {{{0, 0}, {1, 2}}, {{0, 1}, {2, 2}}};   // [parity][t] -> first,last ky summed
    const int C0 = 64, CO = 32;
    // pre-summed fp32 weights in fragment order first; then 16-bit (plain modes) or hi | lo after the power-of-two pre-scale
    // (split mode: per class [hi fragments][lo fragments], scale divided by the pre-scale, see sbbseg_add_conv)
    // (split mode: the image taps are packed two to a k-group -- [tap 2g: ch 0..3][tap 2g + 1: ch 0..3] -- and take ONE K-step,
    // see dec_tail_fused_x3ps; the plain modes keep one tap per k-group, two K-steps)
    const int KS = split ? 5 : kTailKSteps;
    std::vector<float> fragf((size_t)4 * KS * 4 * 64 * 8, 0.f);
    for (int q = 0; q < 4; ++q) {
        const int py = q >> 1, px = q & 1;
        for (int ks = 0; ks < KS; ++ks)
            for (int kk = 0; kk < 2; ++kk)
                for (int mi = 0; mi < 2; ++mi)
                    for (int l = 0; l < 64; ++l) {
                        const int o = conv_row_channel(mi * 16 + (l & 15), CO);
                        const int gidx = kk * 4 + (l >> 4);
                        float* dst = &fragf[((((size_t)q * KS + ks) * 4 + kk * 2 + mi) * 64 + l) * 8];
                        for (int e = 0; e < 8; ++e) {
                            float v = 0.f;
                            if (ks < 4) {
                                const int ty = ks >> 1, tx = ks & 1, ch = gidx * 8 + e;
                                for (int ky = taps[py][ty][0]; ky <= taps[py][ty][1]; ++ky)
                                    for (int kx = taps[px][tx][0]; kx <= taps[px][tx][1]; ++kx)
                                        v += w_src0[((size_t)(ky * 3 + kx) * C0 + ch) * CO + o];
                            } else if (split) {
                                const int t = kk * 8 + 2 * (l >> 4) + (e >> 2), ch = e & 3;
                                if (t < 9 && ch < 3) v = w_img[((size_t)t * 3 + ch) * CO + o];
                            } else {
                                const int t = (ks - 4) * 8 + gidx;
                                if (t < 9 && e < 3) v = w_img[((size_t)t * 3 + e) * CO + o];
                            }
                            dst[e] = v;
                        }
                    }
    }
    const size_t per_class = (size_t)KS * 4 * 64 * 8;
    std::vector<uint16_t> frag(fragf.size() * (split ? 2 : 1), 0);
    std::vector<float> scale_v(scale, scale + CO);
    if (!split) {
        for (size_t i = 0; i < fragf.size(); ++i) frag[i] = c->precision == kF16 ? f32_to_f16_rne(fragf[i]) : f32_to_bf16_rne(fragf[i]);
    } else {
        float wmax = 0.f;
        for (float v : fragf) wmax = std::fmax(wmax, std::fabs(v));
        REQUIRE(std::isfinite(wmax), "%s: non-finite weights", "tail");
        float wpre = 1.f;
        if (wmax > 0.f) {
            int ex = 0;
            (void)std::frexp(wmax, &ex);
            int sexp = 9 - ex;                               // wmax * 2^sexp in [256, 512)
            sexp = sexp > 60 ? 60 : (sexp < -60 ? -60 : sexp);
            wpre = std::ldexp(1.f, sexp);
        }
        for (int q = 0; q < 4; ++q)
            for (size_t i = 0; i < per_class; ++i) {
                const float sv = fragf[q * per_class + i] * wpre;            // exact (power of two)
                const uint16_t hb = f32_to_f16_rne(sv);
                const _Float16 hh = __builtin_bit_cast(_Float16, hb);
                frag[(size_t)q * 2 * per_class + i] = hb;
                frag[(size_t)q * 2 * per_class + per_class + i] = f32_to_f16_rne(sv - (float)hh);
            }
        for (float& v : scale_v) v /= wpre;
    }
    Op op;
    op.type = kTail;
    op.tail.src0 = src0_tensor; op.tail.img = img_c8_tensor; op.tail.classes = classes;
    if (upload(c, (uint16_t**)&op.tail.d_wfrag, frag.data(), frag.size()) || upload(c, &op.tail.d_scale, scale_v.data(), CO) ||
        upload(c, &op.tail.d_shift, shift, CO) || upload(c, &op.tail.d_head_w, head_w, (size_t)CO * classes) ||
        upload(c, &op.tail.d_head_scale, head_scale, classes) || upload(c, &op.tail.d_head_shift, head_shift, classes))
        return 1;
    char nm[96];
    snprintf(nm, sizeof(nm), "tail_conv3x3_c67to32_up_cat_head%d_%dx%d", classes, c->in_H, c->in_W);
    op.name = nm;
    op.flops = 2.0 * (algorithmic_macs > 0 ? algorithmic_macs : (double)c->in_H * c->in_W * CO * (9.0 * 67 + classes));
    op.issued_flops = 2.0 * c->in_H * c->in_W * CO * (double)(KS * kBK) * (split ? 3 : 1);
    op.min_bytes = (double)s0.H * s0.W * 64 * c->elem * c->planes + (double)c->in_H * c->in_W * (8 * c->elem * c->planes + 1);
    c->classes = classes;
    c->ops.push_back(op);
    return 0;
    API_END
}

int sbbseg_add_head(sbbseg_ctx* c, int src_tensor, int cin, int classes, const float* w, const float* scale,
                    const float* shift)
{
    API_BEGIN
    REQUIRE(c && !c->finalized && w && scale && shift, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    REQUIRE(src_tensor >= 0 && src_tensor < (int)c->tensors.size(), "head source undefined");
    const Tensor& s = c->tensors[src_tensor];
    REQUIRE(cin == s.C && cin <= 64, "head cin %d must equal the source channels (%d) and be <= 64", cin, s.C);
    REQUIRE(classes >= 1 && classes <= 8, "head supports 1..8 classes (got %d)", classes);
    REQUIRE(s.H == c->in_H && s.W == c->in_W, "head runs at input resolution");
    REQUIRE(c->classes == 0, "plan already has a head");
    Op op;
    op.type = kHead;
    op.head.src = src_tensor; op.head.cin = cin; op.head.classes = classes;
    if (upload(c, &op.head.d_w, w, (size_t)cin * classes) || upload(c, &op.head.d_scale, scale, classes) ||
        upload(c, &op.head.d_shift, shift, classes))
        return 1;
    char nm[64];
    snprintf(nm, sizeof(nm), "head1x1_c%dto%d_softmax_argmax", cin, classes);
    op.name = nm;
    op.flops = 2.0 * s.H * s.W * cin * classes;
    op.issued_flops = 0;           // plain FMA kernel, no MFMA
    op.min_bytes = (double)s.H * s.W * (cin * c->elem * c->planes + 1);
    c->classes = classes;
    c->ops.push_back(op);
    return 0;
    API_END
}

// ---- fast gather tables (conv_igemm_mfma<..., FG>): for every conv whose K-steps are all regular, whose sources are not
// upsampled and whose taps span at most 4 x 4 offsets per source (over all merged classes), one FgStepRec per K-step.
// Short-K layers keep the plain gather (the per-tile mask set-up costs them 1-10 %; from ~9 K-steps on the fast gather
// wins 2-10 %, profiles/r02_experiments.md), and so do the split mode's 64-channel tiles (+13 % time there).
static int build_fast_gather_tables(sbbseg_ctx* c)
{
    if (const char* e = getenv("SBBSEG_FG_POINTWISE_MIN")) c->fg_pointwise_min_ksteps = atoi(e);
    if (const char* e = getenv("SBBSEG_FG_MIN")) c->fg_min_ksteps = atoi(e);
    std::vector<ConvOp*> convs;
    for (Op& op : c->ops) {
        if (op.type == kConv) convs.push_back(&op.conv);
        for (Op& part : op.parts)
            if (part.type == kConv) convs.push_back(&part.conv);
    }
    for (ConvOp* cop : convs) {
        ConvOp& co = *cop;
        bool pointwise = true;                             // every tap (0, 0): no bounds masks to set up
        for (int s = 0; s < co.d.n_src; ++s)
            pointwise = pointwise && co.tap_lo[s][0] == 0 && co.tap_hi[s][0] == 0 && co.tap_lo[s][1] == 0 && co.tap_hi[s][1] == 0 &&
                        co.d.src[s].pad_top == 0 && co.d.src[s].pad_left == 0 && co.d.src[s].off_y == 0 && co.d.src[s].off_x == 0;
        co.fg_pointwise = pointwise;
        const int min_ksteps = pointwise ? c->fg_pointwise_min_ksteps : c->fg_min_ksteps;
        // fast gather on the split mode's 64-channel tiles too: +13 % time under round 2's [C hi][C lo] layout, -11 % (dec4 3.25 ->
        // 2.89 ms) with the interleaved groups of round 3; SBBSEG_FG_X3_SMALL=0 turns it off (A/B)
        static const bool x3_small = !(getenv("SBBSEG_FG_X3_SMALL") && getenv("SBBSEG_FG_X3_SMALL")[0] == '0');
        if (!co.fg_ok || c->precision == kF32 || co.total_ksteps < min_ksteps || (c->precision == kF16X3 && co.d.cout < 128 && !x3_small)) continue;
        bool ok = true;
        for (int s = 0; s < co.d.n_src; ++s)
            ok = ok && co.d.src[s].up_shift == 0 && co.tap_hi[s][0] - co.tap_lo[s][0] <= 3 && co.tap_hi[s][1] - co.tap_lo[s][1] <= 3 &&
                 co.tap_lo[s][0] >= -8 && co.tap_lo[s][1] >= -8;
        if (!ok) continue;
        for (int q = 0; q < co.n_cls; ++q) {
            const std::vector<KStepRec>& ks = co.h_ksteps_cls[q];
            std::vector<FgStepRec> fg(ks.size());
            for (size_t t = 0; t < ks.size(); ++t) {
                const int s = (int)t < co.ksteps[0] ? 0 : 1;
                const Tensor& tt = c->tensors[co.d.src[s].tensor];
                const long pixb = (long)tt.C * c->elem * c->planes, rowb = (long)tt.W * pixb;
                const long soff = ks[t].dy * rowb + ks[t].dx * pixb + ks[t].coff + (long)kFgBiasPixels(tt.W) * pixb;
                REQUIRE(soff >= 0 && soff < ((long)1 << 31), "fast gather: scalar offset out of range");
                fg[t].soff = (uint32_t)soff;
                fg[t].tapbit = s * 16 + (ks[t].dy - co.tap_lo[s][0]) * 4 + (ks[t].dx - co.tap_lo[s][1]);
                fg[t].pad_[0] = fg[t].pad_[1] = 0;
            }
            if (upload(c, &co.d_fgstep_cls[q], fg.data(), fg.size())) return 1;
        }
    }
    return 0;
}

// ---- bottleneck fusion: [1x1 CIN->64, ReLU] -> [3x3 64->64 direct, ReLU] -> [1x1 -> 256 (+ shortcut), ReLU] at one resolution,
// the two 64-channel tensors in between read by nobody else  ==>  one kBlock op (bottleneck_fused).  The three convs
// stay alive as its parts.  SBBSEG_FUSE_BLOCKS=0 keeps the plan unfused (per-layer tests read the intermediate tensors).
static int fuse_bottlenecks(sbbseg_ctx* c)
{
    const char* env = getenv("SBBSEG_FUSE_BLOCKS");
    const bool split = c->precision == kF16X3;
    if ((env && env[0] == '0') || !(c->precision == kF16 || c->precision == kBF16 || split)) return 0;
    auto readers = [&](int tensor) {
        int nrd = 0;
        for (const Op& o : c->ops) {
            if (o.type == kConv) {
                for (int s = 0; s < o.conv.d.n_src; ++s) nrd += o.conv.d.src[s].tensor == tensor;
                nrd += o.conv.d.residual_tensor == tensor;
            } else if (o.type == kPool) nrd += o.pool.src == tensor;
            else if (o.type == kHead) nrd += o.head.src == tensor;
            else if (o.type == kTail) nrd += (o.tail.src0 == tensor) + (o.tail.img == tensor);
        }
        return nrd;
    };
    const bool f16 = c->precision == kF16;
    auto half = [&](float v) { return f16 ? f32_to_f16_rne(v) : f32_to_bf16_rne(v); };
    for (size_t i = 0; i + 2 < c->ops.size(); ++i) {
        if (c->ops[i].type != kConv || c->ops[i + 1].type != kConv || c->ops[i + 2].type != kConv) continue;
        const ConvOp &A = c->ops[i].conv, &B = c->ops[i + 1].conv, &C = c->ops[i + 2].conv;
        if (A.h_w[0].empty() || C.h_w[0].empty() || !B.d_d64_wfrag) continue;
        if (A.d.n_src != 1 || A.d.cout != 64 || !A.d.relu || A.d.residual_tensor >= 0 || A.n_cls != 1 || !B.d.relu || !C.d.relu || C.d.cout != 256 ||
            C.n_cls != 1 || A.d.out_tensor < 0 || B.d.out_tensor < 0 || C.d.out_tensor < 0)
            continue;
        const int X = A.d.src[0].tensor, T1 = A.d.out_tensor, T2 = B.d.out_tensor, cin = A.d.src[0].channels;
        if (B.d.src[0].tensor != T1 || readers(T1) != 1 || readers(T2) != 1 || T1 == X || T2 == X || C.d.out_tensor == X) continue;
        int proj = -1, b_src = 0;
        if (C.d.n_src == 1 && C.d.src[0].tensor == T2 && C.d.residual_tensor == X && cin == 256) proj = 0;
        else if (C.d.n_src == 2 && C.d.residual_tensor < 0 && cin == 64 &&
                 ((C.d.src[0].tensor == T2 && C.d.src[1].tensor == X) || (C.d.src[1].tensor == T2 && C.d.src[0].tensor == X))) {
            proj = 1;
            b_src = C.d.src[0].tensor == T2 ? 0 : 1;
        }
        if (proj < 0) continue;
        const Tensor& xt = c->tensors[X];
        Op blk;
        blk.type = kBlock;
        blk.block.x_tensor = X; blk.block.out_tensor = C.d.out_tensor; blk.block.cin = cin; blk.block.proj = proj;
        blk.block.H = xt.H; blk.block.W = xt.W;
        // W1: [cin/32 kk][4 mi][64 lanes][8]; W3: [2|4 kk][16 mi][64 lanes][8]; rows = conv_row_channel, k = kk*32 + (lane>>4)*8 + e
        std::vector<uint16_t> f1((size_t)(cin / 32) * 4 * 64 * 8), f3((size_t)(proj ? 4 : 2) * 16 * 64 * 8);
        for (int kk = 0; kk < cin / 32; ++kk)
            for (int mi = 0; mi < 4; ++mi)
                for (int l = 0; l < 64; ++l) {
                    const int o = conv_row_channel(mi * 16 + (l & 15), 64);
                    for (int e = 0; e < 8; ++e)
                        f1[((((size_t)kk * 4 + mi) * 64) + l) * 8 + e] = half(A.h_w[0][(size_t)(kk * 32 + (l >> 4) * 8 + e) * 64 + o]);
                }
        for (int kk = 0; kk < (proj ? 4 : 2); ++kk)
            for (int mi = 0; mi < 16; ++mi)
                for (int l = 0; l < 64; ++l) {
                    const int o = conv_row_channel(mi * 16 + (l & 15), 256);
                    const std::vector<float>& wsrc = C.h_w[kk < 2 ? b_src : 1 - b_src];      // K-steps 0-1 contract b, 2-3 the block input
                    for (int e = 0; e < 8; ++e)
                        f3[((((size_t)kk * 16 + mi) * 64) + l) * 8 + e] = half(wsrc[(size_t)((kk & 1) * 32 + (l >> 4) * 8 + e) * 256 + o]);
                }
        if (split) {
            // hi | lo fragments of the pre-scaled weights (each conv's own power of two, ConvOp::wmul_cls[0] = 2^-s), interleaved per
            // fragment: W1 [8 kk][4 mi][hi | lo][64 lanes][8], W3 [2 kk][16 mi][hi | lo][64 lanes][8]
            auto put = [&](std::vector<uint16_t>& dst, size_t frag, int l, int e, float v, float wpre) {
                const float sv = v * wpre;                       // exact (power of two)
                const uint16_t hb = f32_to_f16_rne(sv);
                dst[((frag * 2 + 0) * 64 + l) * 8 + e] = hb;
                dst[((frag * 2 + 1) * 64 + l) * 8 + e] = f32_to_f16_rne(sv - (float)__builtin_bit_cast(_Float16, hb));
            };
            const float pre1 = 1.f / A.wmul_cls[0], pre3 = 1.f / C.wmul_cls[0];
            const int ka = cin / 32, kc = proj ? 4 : 2;
            f1.assign((size_t)ka * 4 * 2 * 64 * 8, 0);
            f3.assign((size_t)kc * 16 * 2 * 64 * 8, 0);
            for (int kk = 0; kk < ka; ++kk)
                for (int mi = 0; mi < 4; ++mi)
                    for (int l = 0; l < 64; ++l) {
                        const int o = conv_row_channel(mi * 16 + (l & 15), 64);
                        for (int e = 0; e < 8; ++e) put(f1, (size_t)kk * 4 + mi, l, e, A.h_w[0][(size_t)(kk * 32 + (l >> 4) * 8 + e) * 64 + o], pre1);
                    }
            for (int kk = 0; kk < kc; ++kk)                        // the conv's own K order: source 0, then source 1 (block_x3 adds up the same way)
                for (int mi = 0; mi < 16; ++mi)
                    for (int l = 0; l < 64; ++l) {
                        const int o = conv_row_channel(mi * 16 + (l & 15), 256);
                        const std::vector<float>& wsrc = C.h_w[kk < 2 ? 0 : 1];
                        for (int e = 0; e < 8; ++e) put(f3, (size_t)kk * 16 + mi, l, e, wsrc[(size_t)((kk & 1) * 32 + (l >> 4) * 8 + e) * 256 + o], pre3);
                    }
            blk.block.wmul[0] = A.wmul_cls[0]; blk.block.wmul[1] = B.wmul_cls[0]; blk.block.wmul[2] = C.wmul_cls[0];
            if (proj) blk.block.proj = b_src == 0 ? 1 : 2;         // 1: K order [b, x]; 2: [x, b]
        }
        if (upload(c, &blk.block.d_w1, f1.data(), f1.size()) || upload(c, &blk.block.d_w3, f3.data(), f3.size())) return 1;
        char nm[96];
        snprintf(nm, sizeof(nm), "block%s_c%dto64to256_%dx%d", proj ? "_proj" : "", cin, xt.H, xt.W);
        blk.name = nm;
        for (int k = 0; k < 3; ++k) {
            blk.flops += c->ops[i + k].flops;
            blk.issued_flops += c->ops[i + k].issued_flops;
        }
        blk.min_bytes = (double)xt.H * xt.W * (cin + 256) * c->elem * c->planes;        // x read once, y written once
        blk.parts.assign(c->ops.begin() + i, c->ops.begin() + i + 3);
        c->ops.erase(c->ops.begin() + i, c->ops.begin() + i + 3);
        c->ops.insert(c->ops.begin() + i, std::move(blk));
    }
    return 0;
}

// ---- owned-region chain (region.h): the fused tail and, below it, every decoder conv of the form
//   four output-parity classes of conv3x3([nearest-x2 upsampling of the level below, skip]) -> this level
// whose output is read by the level above only.  Level 0 = the tail (network output), level k = the conv k steps below.  A level's rows
// are the rows above dilated by one and halved (region_down), which is exact when class (py, px) reads rows {py - 1, py} / columns
// {px - 1, px} of the level below -- checked here on the K-step records; anything else (unfused heads, fp32 handles, Conv2DTranspose
// decoders whose classes read other taps) leaves the chain short or empty and those ops run whole.
static void find_region_chain(sbbseg_ctx* c)
{
    c->region_levels = 0;
    for (auto& op : c->ops) op.region_level = -1;
    if (c->precision == kF32 || c->ops.empty() || c->ops.back().type != kTail) return;
    if (c->max_batch > kRegionMaxPatches || c->in_H > 2 * kRegionMaxCoord || c->in_W > 2 * kRegionMaxCoord || (c->in_H & 15) || (c->in_W & 15)) return;
    auto readers = [&](int tensor) {
        int nrd = 0;
        for (const Op& o : c->ops) {
            if (o.type == kConv) {
                for (int s = 0; s < o.conv.d.n_src; ++s) nrd += o.conv.d.src[s].tensor == tensor;
                nrd += o.conv.d.residual_tensor == tensor;
            } else if (o.type == kPool) nrd += o.pool.src == tensor;
            else if (o.type == kHead) nrd += o.head.src == tensor;
            else if (o.type == kTail) nrd += (o.tail.src0 == tensor) + (o.tail.img == tensor);
            else if (o.type == kBlock) nrd += o.block.x_tensor == tensor;
        }
        return nrd;
    };
    int level = 0;
    c->region_op[0] = (int)c->ops.size() - 1;
    c->ops.back().region_level = 0;
    int below = c->ops.back().tail.src0;              // the tensor the level above upsamples
    {
        const Tensor& t = c->tensors[below];
        if (2 * t.H != c->in_H || 2 * t.W != c->in_W) { c->ops.back().region_level = -1; return; }
    }
    c->region_levels = 1;
    while (level + 1 < kRegionMaxLevels) {
        int oi = -1;
        for (size_t i = 0; i < c->ops.size(); ++i)
            if (c->ops[i].type == kConv && c->ops[i].conv.d.out_tensor == below) oi = oi < 0 ? (int)i : -2;
        if (oi < 0 || readers(below) != 1) break;
        const ConvOp& co = c->ops[oi].conv;
        const sbbseg_conv_desc& d = co.d;
        const Tensor& to = c->tensors[below];
        if (co.n_cls != 4 || d.n_src != 2 || d.out_stride_y != 2 || d.out_stride_x != 2 || d.residual_tensor >= 0 || d.raw_out_tensor >= 0 ||
            d.head_classes > 0 || d.src[0].stride_y != 1 || d.src[0].stride_x != 1 || d.src[0].up_shift != 0 || d.src[0].off_y || d.src[0].off_x ||
            to.H != 2 * co.Ho || to.W != 2 * co.Wo || co.TH != to.H || co.TW != to.W || !co.fg_ok)
            break;
        const Tensor& t0 = c->tensors[d.src[0].tensor];
        if (t0.H != co.Ho || t0.W != co.Wo || t0.is_input_form) break;
        bool ok = true;
        int seen = 0;
        for (int q = 0; q < 4 && ok; ++q) {
            const int py = co.ooy_cls[q], px = co.oox_cls[q];
            ok = (py == 0 || py == 1) && (px == 0 || px == 1);
            seen |= 1 << (py * 2 + px);
            const int ks0 = co.ksteps[0];
            ok = ok && (int)co.h_ksteps_cls[q].size() >= ks0;
            for (int t = 0; t < ks0 && ok; ++t) {
                const KStepRec& r = co.h_ksteps_cls[q][t];
                ok = !r.irregular && (r.dy == py - 1 || r.dy == py) && (r.dx == px - 1 || r.dx == px);
            }
        }
        if (!ok || seen != 15) break;
        ++level;
        c->region_op[level] = oi;
        c->ops[oi].region_level = level;
        c->region_levels = level + 1;
        below = d.src[0].tensor;
    }
}

int sbbseg_finalize(sbbseg_ctx* c, int max_batch)
{
    API_BEGIN
    REQUIRE(c && !c->finalized, "bad handle / already finalized");
    HIPCHK(hipSetDevice(c->device));
    if (fuse_bottlenecks(c)) return 1;
    if (build_fast_gather_tables(c)) return 1;
    // split mode: the decoder conv at 224 x 224 -- four merged parity classes of  conv3x3([up2(128 ch @ 112 x 112), 64 ch @ 224 x 224]) -> 64 ch
    // -- runs dec_halo_x3 (source halos resident in LDS, dec_halo_x3.hip) on the classes' OWN packed weights and K-step order:
    // the rows of every class matrix are read back and re-laid as MFMA A fragments.  SBBSEG_DEC_HALO=0 keeps the generic kernel.
    {
        const char* env = getenv("SBBSEG_DEC_HALO");
        // (plain fp16 mode: dec_halo_f16.hip -- K-steps of 64 channels: 2 x 4 + 9 of them, fragments of the two k-halves in place of hi | lo)
        const bool x3 = c->precision == kF16X3;
        const int n0 = x3 ? 16 : 8, n1 = x3 ? 18 : 9, nsteps = n0 + n1;
        for (size_t i = 0; (x3 || c->precision == kF16) && !(env && env[0] == '0') && i < c->ops.size(); ++i) {
            Op& op = c->ops[i];
            if (op.type != kConv) continue;
            ConvOp& co = op.conv;
            const sbbseg_conv_desc& d = co.d;
            if (co.n_cls != 4 || d.n_src != 2 || d.cout != 64 || co.ksteps[0] != n0 || co.ksteps[1] != n1 || !co.fg_ok || d.residual_tensor >= 0 ||
                d.raw_out_tensor >= 0 || d.head_classes > 0 || d.out_tensor < 0 || d.out_stride_y != 2 || d.out_stride_x != 2)
                continue;
            const Tensor &t0 = c->tensors[d.src[0].tensor], &t1 = c->tensors[d.src[1].tensor], &to = c->tensors[d.out_tensor];
            if (t0.C != 128 || d.src[0].channels != 128 || t1.C != 64 || d.src[1].channels != 64 || d.src[0].stride_y != 1 || d.src[0].stride_x != 1 ||
                d.src[0].up_shift != 0 || d.src[1].stride_y != 2 || d.src[1].stride_x != 2 || t0.is_input_form || t1.is_input_form ||
                t1.H != 2 * t0.H || t1.W != 2 * t0.W || to.H != t1.H || to.W != t1.W || d.out_h != t0.H || d.out_w != t0.W || (t0.H & 7) || (t0.W & 7))
                continue;
            // class q must be the output parity (q >> 1, q & 1) -- the kernel's wave <-> class map -- and its taps must stay inside the halos
            bool ok = true;
            std::vector<int> taps(4 * 16, 0);
            for (int q = 0; q < 4 && ok; ++q) {
                ok = co.ooy_cls[q] == (q >> 1) && co.oox_cls[q] == (q & 1) && (int)co.h_ksteps_cls[q].size() == nsteps;
                for (int t = 0; t < nsteps && ok; ++t) {
                    const KStepRec& r = co.h_ksteps_cls[q][t];
                    const int g = t < n0 ? t >> 2 : (t - n0) / 9, ti = t < n0 ? t & 3 : (t - n0) % 9;
                    ok = !r.irregular && r.coff == g * 128;                                   // channel group g of the stored pixel
                    if (t < n0) ok = ok && r.dy >= -1 && r.dy <= 1 && r.dx >= -1 && r.dx <= 1;             // halo row i + dy + 1 in [0, 9]
                    else ok = ok && r.dy >= -1 && r.dy <= 2 && r.dx >= -1 && r.dx <= 2;                    // halo row 2 i + dy + 1 in [0, 17]
                    const int word = (r.dy & 255) | ((r.dx & 255) << 8);
                    const int slot = q * 16 + (t < n0 ? ti : 4 + ti);
                    if (g == 0) taps[slot] = word;
                    else ok = ok && taps[slot] == word;                                       // every group walks the same taps
                }
            }
            if (!ok) continue;
            alloc_check();
            const size_t row_halves = (size_t)co.Ktot, frag_halves = (size_t)4 * nsteps * 4 * 2 * 64 * 8;
            std::vector<uint16_t> mat((size_t)64 * row_halves), frag(frag_halves);
            for (int q = 0; q < 4; ++q) {
                HIPCHK(hipMemcpy(mat.data(), co.d_w_cls[q], mat.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));     // packed rows 0..63
                for (int t = 0; t < nsteps; ++t)
                    for (int mi = 0; mi < 4; ++mi)
                        for (int lo = 0; lo < 2; ++lo)                                        // split mode: hi | lo plane; fp16: k-half
                            for (int l = 0; l < 64; ++l) {
                                const uint16_t* src = &mat[(size_t)(mi * 16 + (l & 15)) * row_halves + (size_t)t * 64 + lo * 32 + (l >> 4) * 8];
                                uint16_t* dst = &frag[((((size_t)(q * nsteps + t) * 4 + mi) * 2 + lo) * 64 + l) * 8];
                                for (int e = 0; e < 8; ++e) dst[e] = src[e];
                            }
            }
            if (upload(c, &co.d_halo_wfrag, frag.data(), frag.size()) || upload(c, &co.d_halo_taps, taps.data(), taps.size())) return 1;
        }
    }
    // split and plain fp16 modes, encoder stages 3 / 4: an identity block's last 1x1 conv (C -> 4C, + residual, ReLU) directly followed by the next block's
    // first 1x1 conv (4C -> C, stride 1, ReLU) -> one launch writes both outputs (expand_reduce_x3.hip: y is contracted from LDS instead
    // of being read back).  Both matrices are read back and re-laid as MFMA A fragments.  SBBSEG_EXPAND_REDUCE=0 keeps two launches.
    {
        const char* env = getenv("SBBSEG_EXPAND_REDUCE");
        const bool x3 = c->precision == kF16X3;
        const int kch = x3 ? 32 : 64;                                  // channels per K-step (plain fp16: two k-halves in place of hi | lo)
        auto pointwise = [&](const ConvOp& co, int cin, int cout) -> bool {
            const sbbseg_conv_desc& d = co.d;
            if (co.n_cls != 1 || d.n_src != 1 || d.cout != cout || d.src[0].channels != cin || d.src[0].kh != 1 || d.src[0].kw != 1 ||
                d.src[0].stride_y != 1 || d.src[0].stride_x != 1 || d.src[0].pad_top || d.src[0].pad_left || d.src[0].up_shift || d.src[0].off_y ||
                d.src[0].off_x || !d.relu || d.raw_out_tensor >= 0 || d.head_classes > 0 || d.out_tensor < 0 || d.out_stride_y != 1 || d.out_stride_x != 1 ||
                d.out_off_y || d.out_off_x || co.d_stem_wfrag || co.d_d64_wfrag || co.d_halo_wfrag || co.total_ksteps != cin / kch ||
                co.Ktot != cin * (x3 ? 2 : 1) || co.cout_pad < cout || (int)co.h_ksteps_cls[0].size() != cin / kch)
                return false;
            const Tensor& t = c->tensors[d.src[0].tensor];
            if (t.C != cin || t.is_input_form || t.H != d.out_h || t.W != d.out_w) return false;
            for (int k = 0; k < cin / kch; ++k) {
                const KStepRec& r = co.h_ksteps_cls[0][k];
                if (r.irregular || r.dy || r.dx || r.coff != k * 128) return false;     // K-step k = channel group k of the stored pixel
            }
            return true;
        };
        for (size_t i = 0; (x3 || c->precision == kF16) && !(env && env[0] == '0') && i + 1 < c->ops.size(); ++i) {
            if (c->ops[i].type != kConv || c->ops[i + 1].type != kConv) continue;
            ConvOp& e = c->ops[i].conv;
            ConvOp& r = c->ops[i + 1].conv;
            const int C = e.d.src[0].channels;
            if ((C != 128 && C != 256) || !pointwise(e, C, 4 * C) || !pointwise(r, 4 * C, C)) continue;
            if (e.d.residual_tensor < 0 || r.d.residual_tensor >= 0 || r.d.src[0].tensor != e.d.out_tensor || e.fused_into_expand) continue;
            const Tensor &tx = c->tensors[e.d.residual_tensor], &ty = c->tensors[e.d.out_tensor], &ta = c->tensors[r.d.out_tensor];
            if (tx.C != 4 * C || tx.H != ty.H || tx.W != ty.W || ty.C != 4 * C || ty.H != e.d.out_h || ty.W != e.d.out_w || ta.C != C ||
                ta.H != ty.H || ta.W != ty.W || tx.is_input_form)
                continue;
            alloc_check();
            const int KS1 = C / kch, G2S = 256 / kch, NCH = C / 64, MI2 = C / 128;
            std::vector<uint16_t> m3((size_t)e.cout_pad * e.Ktot), m1((size_t)r.cout_pad * r.Ktot);
            HIPCHK(hipMemcpy(m3.data(), e.d_w, m3.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(m1.data(), r.d_w, m1.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
            std::vector<uint16_t> f3((size_t)NCH * KS1 * 8 * 2 * 2 * 64 * 8), f1((size_t)NCH * G2S * 8 * MI2 * 2 * 64 * 8);
            for (int j = 0; j < NCH; ++j)
                for (int w = 0; w < 8; ++w)
                    for (int lo = 0; lo < 2; ++lo)
                        for (int l = 0; l < 64; ++l) {
                            for (int k = 0; k < KS1; ++k)
                                for (int m = 0; m < 2; ++m) {
                                    const int row = (j * 16 + w * 2 + m) * 16 + (l & 15);
                                    const uint16_t* src = &m3[(size_t)row * e.Ktot + (size_t)k * 64 + lo * 32 + (l >> 4) * 8];
                                    uint16_t* dst = &f3[((((((size_t)j * KS1 + k) * 8 + w) * 2 + m) * 2 + lo) * 64 + l) * 8];
                                    for (int q = 0; q < 8; ++q) dst[q] = src[q];
                                }
                            for (int k = 0; k < G2S; ++k)
                                for (int m = 0; m < MI2; ++m) {
                                    const int row = (w * MI2 + m) * 16 + (l & 15);
                                    const uint16_t* src = &m1[(size_t)row * r.Ktot + (size_t)(j * G2S + k) * 64 + lo * 32 + (l >> 4) * 8];
                                    uint16_t* dst = &f1[((((((size_t)j * G2S + k) * 8 + w) * MI2 + m) * 2 + lo) * 64 + l) * 8];
                                    for (int q = 0; q < 8; ++q) dst[q] = src[q];
                                }
                        }
            if (upload(c, &e.d_er_w3, f3.data(), f3.size()) || upload(c, &e.d_er_w1, f1.data(), f1.size())) return 1;
            e.fused_reduce = (int)(i + 1);
            r.fused_into_expand = true;
        }
        // Round 6, stage 3 (C = 128; H, W multiples of 8): the identity block's 3x3 conv in front of such a pair joins the launch
        // (conv3_expand_reduce.hip: b stays in LDS).  The conv's packed rows are re-laid as A fragments in ITS K-step order, the taps and
        // channel groups of the K-steps go along as a table.  SBBSEG_C3ER=0 keeps the 3x3 conv's own launch.
        const char* env3 = getenv("SBBSEG_C3ER");
        const char* envfb = getenv("SBBSEG_FUSE_BLOCKS");          // (= 0: "keep every plan tensor materialised" -- the per-layer tests; b would not be)
        if (envfb && envfb[0] == '0') env3 = "0";
        auto readers_of = [&](int tensor) {
            int nrd = 0;
            for (const Op& o : c->ops) {
                if (o.type == kConv) {
                    for (int s = 0; s < o.conv.d.n_src; ++s) nrd += o.conv.d.src[s].tensor == tensor;
                    nrd += o.conv.d.residual_tensor == tensor;
                } else if (o.type == kPool) nrd += o.pool.src == tensor;
                else if (o.type == kHead) nrd += o.head.src == tensor;
                else if (o.type == kTail) nrd += (o.tail.src0 == tensor) + (o.tail.img == tensor);
                else if (o.type == kBlock) nrd += o.block.x_tensor == tensor;
            }
            return nrd;
        };
        for (size_t i = 1; (x3 || c->precision == kF16) && !(env && env[0] == '0') && !(env3 && env3[0] == '0') && i < c->ops.size(); ++i) {
            if (c->ops[i].type != kConv || c->ops[i - 1].type != kConv) continue;
            ConvOp& e = c->ops[i].conv;
            ConvOp& k3 = c->ops[i - 1].conv;
            if (e.fused_reduce < 0) continue;
            const int C = e.d.src[0].channels;
            const sbbseg_conv_desc& d = k3.d;
            const int ks0 = 9 * C / kch;
            if (C != 128 || k3.n_cls != 1 || d.n_src != 1 || d.cout != C || d.src[0].channels != C || d.src[0].kh != 3 || d.src[0].kw != 3 ||
                d.src[0].stride_y != 1 || d.src[0].stride_x != 1 || d.src[0].pad_top != 1 || d.src[0].pad_left != 1 || d.src[0].up_shift || d.src[0].off_y ||
                d.src[0].off_x || !d.relu || d.residual_tensor >= 0 || d.raw_out_tensor >= 0 || d.head_classes > 0 || d.out_tensor != e.d.src[0].tensor ||
                d.out_stride_y != 1 || d.out_stride_x != 1 || d.out_off_y || d.out_off_x || k3.d_stem_wfrag || k3.d_d64_wfrag || k3.d_halo_wfrag ||
                k3.fused_into_expand || k3.fused_reduce >= 0 || k3.total_ksteps != ks0 || k3.Ktot != ks0 * 64 || k3.cout_pad < C ||
                (int)k3.h_ksteps_cls[0].size() != ks0 || readers_of(d.out_tensor) != 1)
                continue;
            const Tensor &ta = c->tensors[d.src[0].tensor], &tb = c->tensors[d.out_tensor];
            if (ta.C != C || ta.is_input_form || ta.H != d.out_h || ta.W != d.out_w || tb.H != ta.H || tb.W != ta.W || (ta.H & 7) || (ta.W & 7)) continue;
            std::vector<int> k0(ks0);
            bool ok = true;
            for (int t = 0; t < ks0 && ok; ++t) {
                const KStepRec& r = k3.h_ksteps_cls[0][t];
                ok = !r.irregular && r.dy >= -1 && r.dy <= 1 && r.dx >= -1 && r.dx <= 1 && r.coff >= 0 && r.coff % 128 == 0 && r.coff / 128 < C / kch;
                k0[t] = (r.dy & 255) | ((r.dx & 255) << 8) | ((r.coff / 128) << 16);
            }
            if (!ok) continue;
            alloc_check();
            const int MI0 = C / 128;
            std::vector<uint16_t> m2((size_t)k3.cout_pad * k3.Ktot);
            HIPCHK(hipMemcpy(m2.data(), k3.d_w, m2.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
            std::vector<uint16_t> f2((size_t)ks0 * 8 * MI0 * 2 * 64 * 8);
            for (int t = 0; t < ks0; ++t)
                for (int w = 0; w < 8; ++w)
                    for (int m = 0; m < MI0; ++m)
                        for (int lo = 0; lo < 2; ++lo)
                            for (int l = 0; l < 64; ++l) {
                                const int row = (w * MI0 + m) * 16 + (l & 15);
                                const uint16_t* src = &m2[(size_t)row * k3.Ktot + (size_t)t * 64 + lo * 32 + (l >> 4) * 8];
                                uint16_t* dst = &f2[(((((size_t)t * 8 + w) * MI0 + m) * 2 + lo) * 64 + l) * 8];
                                for (int q = 0; q < 8; ++q) dst[q] = src[q];
                            }
            if (upload(c, &e.d_c3_w2, f2.data(), f2.size()) || upload(c, &e.d_c3_k0, k0.data(), k0.size())) return 1;
            e.fused_conv3 = (int)(i - 1);
            k3.fused_into_c3 = true;
        }
    }
    // split and plain fp16 modes: the stem (dedicated kernel, raw output only) directly followed by the 3x3 / stride-2 max-pool of that tensor with an
    // affine on every tap (bn_conv1 + ReLU) -> one launch writes both tensors (stem_pool_x3.hip); SBBSEG_STEM_POOL=0 keeps two launches
    {
        const char* env = getenv("SBBSEG_STEM_POOL");
        // (plain fp16 mode: the one-plane form stem_pool<false> is bit-identical too but no faster than the two launches -- 0.53 against
        // 0.26 + 0.29 ms per 140 patches: with one MFMA per product the pool stage is most of the kernel -- so it is opt-in: SBBSEG_STEM_POOL_F16=1)
        const char* env16 = getenv("SBBSEG_STEM_POOL_F16");
        const bool on = c->precision == kF16X3 || (c->precision == kF16 && env16 && env16[0] == '1');
        for (size_t i = 0; on && !(env && env[0] == '0') && i + 1 < c->ops.size(); ++i) {
            Op& a = c->ops[i];
            Op& b = c->ops[i + 1];
            if (a.type != kConv || !a.conv.d_stem_wfrag || b.type != kPool) continue;
            const PoolOp& po = b.pool;
            const Tensor& f1 = c->tensors[a.conv.d.out_tensor];
            if (po.src != a.conv.d.out_tensor || po.k != 3 || po.stride != 2 || !po.d_pre_scale || f1.C != 64 || (f1.H & 15) || (f1.W & 15) ||
                po.Ho != f1.H / 2 - 1 || po.Wo != f1.W / 2 - 1)
                continue;
            a.conv.fused_pool = (int)(i + 1);
            b.pool.fused_into_stem = true;
        }
    }
    REQUIRE(max_batch >= 1, "max_batch must be >= 1");
    REQUIRE(c->classes > 0 && !c->ops.empty(), "plan must contain a head (head op or a conv with a fused head)");
    c->max_batch = max_batch;
    find_region_chain(c);
    for (auto& t : c->tensors) {
        const size_t bytes = kZeroHeaderBytes + t.elems_per_patch * max_batch * c->elem * c->planes + 256;
        REQUIRE(bytes < ((size_t)1 << 32), "tensor %dx%dx%d x batch %d exceeds the 4 GiB gather window", t.H, t.W, t.C, max_batch);
        if (dmalloc(c, (void**)&t.lane_buf[0], bytes)) return 1;
        t.buf = t.lane_buf[0];
        // input forms rely on their zero borders / zero channels; headers must be zero for every tensor
        HIPCHK(hipMemset(t.buf, 0, t.is_input_form ? bytes : (size_t)kZeroHeaderBytes));
    }
    c->lane1_batch = (c->lanes == 2 && max_batch >= 2 * kMinLaneTiles) ? (max_batch + 1) / 2 : 0;
    if (c->lane1_batch)
        for (auto& t : c->tensors) {
            const size_t bytes = kZeroHeaderBytes + t.elems_per_patch * c->lane1_batch * c->elem * c->planes + 256;
            if (dmalloc(c, (void**)&t.lane_buf[1], bytes)) return 1;
            HIPCHK(hipMemset(t.lane_buf[1], 0, t.is_input_form ? bytes : (size_t)kZeroHeaderBytes));
        }
    // a 3x3 conv that conv3_expand_reduce computes never writes its output tensor: give the buffers a defined content (zeros) -- the debug
    // read-back of a plan tensor and the tests that compare whole plans then see the same bytes in every run
    for (const Op& op : c->ops)
        if (op.type == kConv && op.conv.fused_into_c3 && op.conv.d.out_tensor >= 0) {
            Tensor& t = c->tensors[op.conv.d.out_tensor];
            for (int lane = 0; lane < 2; ++lane)
                if (t.lane_buf[lane])
                    HIPCHK(hipMemset(t.lane_buf[lane], 0, kZeroHeaderBytes + t.elems_per_patch * (size_t)(lane == 0 ? max_batch : c->lane1_batch) * c->elem * c->planes));
        }
    float lut[256];
    for (int v = 0; v < 256; ++v) lut[v] = (float)((double)v / 255.0);   // main.py:239 in f64, then Keras' f32 feed
    if (upload(c, &c->d_lut, lut, 256)) return 1;
    if (dmalloc(c, (void**)&c->d_hist, 257 * sizeof(unsigned))) return 1;
    if (dmalloc(c, (void**)&c->d_tile_xy, sizeof(int) * 2 * max_batch)) return 1;
    if (dmalloc(c, (void**)&c->d_batch_labels, (size_t)max_batch * c->in_H * c->in_W)) return 1;
    HIPCHK(hipDeviceSynchronize());
    c->finalized = true;
    return 0;
    API_END
}

// ----------------------------------------------------------------------------------------- queries
int sbbseg_model_info(sbbseg_ctx* c, int* H, int* W, int* classes, int* max_batch)
{
    API_BEGIN
    REQUIRE(c, "null handle");
    if (H) *H = c->in_H;
    if (W) *W = c->in_W;
    if (classes) *classes = c->classes;
    if (max_batch) *max_batch = c->max_batch;
    return 0;
    API_END
}

int sbbseg_num_ops(sbbseg_ctx* c, int* n)
{
    API_BEGIN
    REQUIRE(c && n, "bad arguments");
    *n = (int)c->ops.size();
    return 0;
    API_END
}

int sbbseg_op_info(sbbseg_ctx* c, int op, char* name, int name_len, double* flops_per_patch, double* min_bytes_per_patch)
{
    API_BEGIN
    REQUIRE(c && op >= 0 && op < (int)c->ops.size(), "op index out of range");
    if (name && name_len > 0) snprintf(name, name_len, "%s", c->ops[op].name.c_str());
    if (flops_per_patch) *flops_per_patch = c->ops[op].flops;
    if (min_bytes_per_patch) *min_bytes_per_patch = c->ops[op].min_bytes;
    return 0;
    API_END
}

int sbbseg_op_issued_flops(sbbseg_ctx* c, int op, double* issued_flops_per_patch)
{
    API_BEGIN
    REQUIRE(c && op >= 0 && op < (int)c->ops.size() && issued_flops_per_patch, "op index out of range");
    *issued_flops_per_patch = c->ops[op].issued_flops;
    return 0;
    API_END
}

int sbbseg_device_bytes(sbbseg_ctx* c, size_t* bytes)
{
    API_BEGIN
    REQUIRE(c && bytes, "bad arguments");
    *bytes = c->device_bytes;
    return 0;
    API_END
}

// -------------------------------------------------------------------------------------- seam 2
int sbbseg_predict(sbbseg_ctx* c, const float* x_nhwc, int n, float* probs_nhwc)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(x_nhwc && probs_nhwc && n >= 0, "bad arguments");
    const size_t per_in = (size_t)c->in_H * c->in_W * 3, per_out = (size_t)c->in_H * c->in_W * c->classes;
    if (!c->d_xin && dmalloc(c, (void**)&c->d_xin, per_in * c->max_batch * sizeof(float))) return 1;
    if (!c->d_probs && dmalloc(c, (void**)&c->d_probs, per_out * c->max_batch * sizeof(float))) return 1;
    IngestParams ip;
    if (fill_ingest(c, ip)) return 1;
    for (int done = 0; done < n; done += c->max_batch) {
        const int nb = n - done < c->max_batch ? n - done : c->max_batch;
        HIPCHK(hipMemcpyAsync(c->d_xin, x_nhwc + done * per_in, per_in * nb * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIPCHK(launch_ingest_f32(c->d_xin, nb, c->in_H, c->in_W, ip.c8, ip.pairs, ip.pad, ip.pairs_w, c->precision, c->stream));
        if (run_plan(c, nb, c->d_batch_labels, c->d_probs)) return 1;
        HIPCHK(hipMemcpyAsync(probs_nhwc + done * per_out, c->d_probs, per_out * nb * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return 0;
    API_END
}

// -------------------------------------------------------------------------------------- seam 1
int sbbseg_tile_grid(int Hp, int Wp, int H, int W, int32_t* tile_xy, int capacity, int* nxf, int* nyf)
{
    API_BEGIN
    alloc_check();
    std::vector<int> ox, oy;
    const int margin = margin_of(W);
    const int nx = axis_tiles(Wp, W, margin, ox), ny = axis_tiles(Hp, H, margin, oy);
    REQUIRE(nx > 0 && ny > 0, "page %dx%d is smaller than the model input %dx%d (unsupported by the reference too, main.py:278-281)", Hp, Wp, H, W);
    if (nxf) *nxf = nx;
    if (nyf) *nyf = ny;
    if (tile_xy) {
        REQUIRE(capacity >= nx * ny, "tile_xy capacity %d < %d tiles", capacity, nx * ny);
        for (int i = 0; i < nx; ++i)               // x outer, y inner: main.py:259-260
            for (int j = 0; j < ny; ++j) {
                tile_xy[2 * (i * ny + j)] = ox[i];
                tile_xy[2 * (i * ny + j) + 1] = oy[j];
            }
    }
    return 0;
    API_END
}

int sbbseg_nearest_map(int src_len, int dst_len, int32_t* map)
{
    API_BEGIN
    REQUIRE(src_len > 0 && dst_len > 0 && map, "bad arguments");
    alloc_check();
    std::vector<int> m;
    nearest_map(src_len, dst_len, m);
    for (int i = 0; i < dst_len; ++i) map[i] = m[i];
    return 0;
    API_END
}

int sbbseg_segment_tiles_dev(sbbseg_ctx* c, const void* d_page_hwc, int Hp, int Wp, const int32_t* tile_xy, int n_tiles,
                             void* d_tile_labels)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(d_page_hwc && tile_xy && d_tile_labels && n_tiles >= 0, "bad arguments");
    for (int t = 0; t < n_tiles; ++t)
        REQUIRE(tile_xy[2 * t] >= 0 && tile_xy[2 * t] + c->in_W <= Wp && tile_xy[2 * t + 1] >= 0 && tile_xy[2 * t + 1] + c->in_H <= Hp,
                "tile %d at (%d,%d) leaves the %dx%d page", t, tile_xy[2 * t], tile_xy[2 * t + 1], Hp, Wp);
    IngestParams ip;
    if (fill_ingest(c, ip)) return 1;
    ip.page = (const uint8_t*)d_page_hwc; ip.Hp = Hp; ip.Wp = Wp; ip.src_Hp = Hp; ip.src_Wp = Wp; ip.tile_xy = c->d_tile_xy;
    const size_t per = (size_t)c->in_H * c->in_W;
    for (int done = 0; done < n_tiles; done += c->max_batch) {
        const int nb = n_tiles - done < c->max_batch ? n_tiles - done : c->max_batch;
        // explicit origin lists are the slow, general form: the table is re-used per chunk, so wait
        // for the previous chunk's ingest before overwriting it (the grid form below needs no table)
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpy(c->d_tile_xy, tile_xy + 2 * done, sizeof(int) * 2 * nb, hipMemcpyHostToDevice));
        ip.n_tiles = nb;
        HIPCHK(launch_ingest_u8(ip, c->precision, c->stream));
        if (run_plan(c, nb, (uint8_t*)d_tile_labels + done * per, nullptr)) return 1;
    }
    return 0;
    API_END
}

// Tile grid of the FUSED page paths: the reference's grid (sbbseg_tile_grid) minus the repeated last tile of an axis whose last two
// origins coincide (at most the last two can: three equal origins would need 2 * mid < tile - mid).  Origins are still
// min(t * mid, extent - tile) for t < n', so the ingest kernel's closed form holds on the smaller grid; the dropped tile's pixels
// [origin + margin, origin + tile) are exactly what tile n' - 1 pastes as the new last tile (axis_owner with dedupe).
static int fused_grid(const sbbseg_ctx* c, int Hp, int Wp, bool dedupe, int* nx, int* ny)
{
    if (sbbseg_tile_grid(Hp, Wp, c->in_H, c->in_W, nullptr, 0, nx, ny)) return 1;
    if (!dedupe) return 0;
    const int margin = margin_of(c->in_W);
    std::vector<int> o;
    if (axis_tiles(Wp, c->in_W, margin, o) >= 2 && o[o.size() - 1] == o[o.size() - 2]) --*nx;
    if (axis_tiles(Hp, c->in_H, margin, o) >= 2 && o[o.size() - 1] == o[o.size() - 2]) --*ny;
    return 0;
}

// Tiles [first_tile, first_tile + n_tiles) of the tile list of `n_pages` equally sized pages (page-major: tile g = page g / tpp,
// grid index g % tpp) -> d_tile_labels[g - first_tile].  Chunks of <= max_batch tiles may span pages: big launches fill the chip's
// persistent grids better than one page's 70 tiles (profiles/r02_experiments.md).
// `owned`: the decoder of every tile is launched over the region the page stitch keeps of it (+ the halo the levels above need) only
// (region.h) -- d_tile_labels is then defined on the owned regions, which is all stitch_impl reads.
static int tile_range_impl(sbbseg_ctx* c, const void* const* d_pages, int n_pages, int src_Hp, int src_Wp, const int* d_map_y, const int* d_map_x,
                           int Hp, int Wp, int first_tile, int n_tiles, void* d_tile_labels, const int* d_bin_thr = nullptr, bool dedupe = false,
                           bool owned = false)
{
    REQUIRE(d_pages && n_pages >= 1 && d_tile_labels && first_tile >= 0 && n_tiles >= 0, "bad arguments");
    for (int k = 0; k < n_pages; ++k) REQUIRE(d_pages[k], "null page pointer (page %d)", k);
    int nx = 0, ny = 0;
    if (fused_grid(c, Hp, Wp, dedupe, &nx, &ny)) return 1;
    const int tpp = nx * ny;
    REQUIRE((long)first_tile + n_tiles <= (long)tpp * n_pages, "tile range [%d,%d) exceeds the %d tiles of the %d page(s)", first_tile,
            first_tile + n_tiles, tpp * n_pages, n_pages);
    const int margin = margin_of(c->in_W);
    IngestParams ip;
    if (fill_ingest(c, ip)) return 1;
    ip.page = (const uint8_t*)d_pages[0]; ip.Hp = Hp; ip.Wp = Wp; ip.src_Hp = src_Hp; ip.src_Wp = src_Wp; ip.tile_xy = nullptr;
    ip.map_y = d_map_y; ip.map_x = d_map_x; ip.bin_thr = d_bin_thr;
    ip.grid_nyf = ny; ip.grid_mid_x = c->in_W - 2 * margin; ip.grid_mid_y = c->in_H - 2 * margin;
    const size_t per = (size_t)c->in_H * c->in_W;
    const size_t act = (size_t)c->elem * c->planes;          // bytes per stored element
    // owned-region launches: the page geometry in closed form; the tables of a chunk are built on its lane's stream in front of its forward
    const bool regions = owned && c->region_levels > 0 && c->max_batch <= kRegionMaxPatches;
    RegionGeom rg;
    memset(&rg, 0, sizeof(rg));
    if (regions) {
        rg.ax = {Wp, c->in_W, margin, c->in_W - 2 * margin, nx};
        rg.ay = {Hp, c->in_H, margin, c->in_H - 2 * margin, ny};
        rg.tpp = tpp; rg.ny = ny; rg.n_levels = c->region_levels;
        for (int L = 0; L < c->region_levels; ++L) {
            const Op& lop = c->ops[c->region_op[L]];
            if (L == 0) { rg.kind[L] = 0; rg.Rh[L] = c->in_H; rg.Rw[L] = c->in_W; rg.align_x[L] = 16; }
            else {
                const Tensor& to = c->tensors[lop.conv.d.out_tensor];
                rg.kind[L] = runs_dec_halo(c, lop.conv) ? 0 : 1;
                rg.Rh[L] = to.H; rg.Rw[L] = to.W; rg.align_x[L] = 2;
            }
        }
    }
    auto setup_regions = [&](int lane, int g_first, int nb) -> int {       // (inside the lane's scope: c->stream is the lane's stream)
        RegionBuildParams bp;
        memset(&bp, 0, sizeof(bp));
        bp.g = rg; bp.g0 = g_first; bp.nb = nb;
        RegionRun& rr = c->rr;
        const size_t batch_cap = (size_t)(lane == 0 ? c->max_batch : c->lane1_batch);
        for (int L = 0; L < rg.n_levels; ++L) {
            long total = 0;
            for (int q = 0; q < nb; ++q) {
                const int local = (g_first + q) % tpp, i = local / ny, j = local - i * ny;
                total += region_entries(rg, i, j, L);
            }
            const size_t need = 4 * (rg.kind[L] ? 4 * batch_cap * (size_t)(rg.Rh[L] / 2) * (size_t)(rg.Rw[L] / 2)
                                                 : batch_cap * (size_t)(rg.Rh[L] / 16 + 1) * (size_t)(rg.Rw[L] / 16 + 1));
            if (c->rtab_cap[lane][L] < need) {
                HIPCHK(hipDeviceSynchronize());                  // (first use / a kind switched by an A/B knob: rare)
                if (c->d_rtab[lane][L]) { HIPCHK(hipFree(c->d_rtab[lane][L])); c->device_bytes -= c->rtab_cap[lane][L]; c->d_rtab[lane][L] = nullptr; c->rtab_cap[lane][L] = 0; }
                if (dmalloc(c, (void**)&c->d_rtab[lane][L], need)) return 1;
                c->rtab_cap[lane][L] = need;
            }
            bp.out[L] = c->d_rtab[lane][L]; bp.total[L] = (int)total;
            rr.kind[L] = rg.kind[L]; rr.total[L] = (int)total; rr.tab[L] = c->d_rtab[lane][L];
            rr.frac[L] = (double)total * (rg.kind[L] ? 4.0 : 256.0) / ((double)nb * rg.Rh[L] * rg.Rw[L]);
        }
        HIPCHK(launch_region_build(bp, c->stream));
        rr.on = true;
        return 0;
    };
    struct RegionOff { sbbseg_ctx* c; ~RegionOff() { c->rr.on = false; } };
    auto run_chunk = [&](int lane, int first, int nb, bool halves = false) -> int {
        LaneScope scope(c, lane, halves);
        RegionOff roff{c};
        if (regions && setup_regions(lane, first_tile + first, nb)) return 1;
        IngestParams lp = ip;
        if (fill_ingest(c, lp)) return 1;          // (input-form pointers of this lane)
        char* const c8_base = (char*)lp.c8;
        char* const pairs_base = (char*)lp.pairs;
        const size_t c8_tile = per * 8 * act, pairs_tile = lp.pairs ? (size_t)(c->in_H + 2 * lp.pad) * lp.pairs_w * 8 * act : 0;
        lp.Hp = ip.Hp; lp.Wp = ip.Wp; lp.src_Hp = ip.src_Hp; lp.src_Wp = ip.src_Wp; lp.tile_xy = nullptr;
        lp.map_y = ip.map_y; lp.map_x = ip.map_x; lp.bin_thr = ip.bin_thr;
        lp.grid_nyf = ip.grid_nyf; lp.grid_mid_x = ip.grid_mid_x; lp.grid_mid_y = ip.grid_mid_y;
        for (int done = 0; done < nb;) {            // one ingest launch per page the chunk touches
            const int g = first_tile + first + done, pg = g / tpp, local = g - pg * tpp;
            const int run = nb - done < tpp - local ? nb - done : tpp - local;
            lp.page = (const uint8_t*)d_pages[pg];
            lp.grid_first = local;
            lp.n_tiles = run;
            lp.c8 = c8_base + (size_t)done * c8_tile;
            lp.pairs = pairs_base ? pairs_base + (size_t)done * pairs_tile : nullptr;
            HIPCHK(launch_ingest_u8(lp, c->precision, c->stream));
            done += run;
        }
        return run_plan(c, nb, (uint8_t*)d_tile_labels + first * per, nullptr);
    };
    // CU-partitioned, staggered lanes (see sbbseg_ctx::half_stream): lane 0 takes the first half of the tile range in units of
    // u = max_batch / 2 tiles, lane 1 the second half as [u / 2, u, u, ..., rest] -- its short first unit puts it half a network
    // behind lane 0, so that from then on one lane's decoder (MFMA-bound, power-capped) runs beside the other lane's encoder
    // (HBM-bound) on disjoint halves of the chip; no join before the end of the range.  Results do not depend on the schedule:
    // every tile is computed by the same kernels from the same operands (test_two_lanes_equal_one_lane, the chunking tests).
    if (c->cu_split && c->lane1_batch > 0 && c->lanes == 2 && !c->profiling && n_tiles >= c->stagger_min_tiles && n_tiles >= 4 * kMinLaneTiles) {
        const int u = c->lane1_batch;                              // <= the second lane's buffers (and lane 0's hold max_batch >= u)
        const int n0 = (n_tiles + 1) / 2, n1 = n_tiles - n0;
        std::vector<std::pair<int, int>> units[2];                 // (first tile, count)
        for (int o = 0; o < n0; o += u) units[0].push_back({o, n0 - o < u ? n0 - o : u});
        int o1 = 0;
        static const int head_pct = getenv("SBBSEG_STAGGER_HEAD_PCT") ? atoi(getenv("SBBSEG_STAGGER_HEAD_PCT")) : 50;      // probe knob
        const int head_u = head_pct > 0 ? u * head_pct / 100 : u;
        const int head = n1 > u ? (head_u > kMinLaneTiles ? head_u : kMinLaneTiles) : n1;
        units[1].push_back({n0, head < n1 ? head : n1});
        o1 = units[1][0].second;
        for (; o1 < n1; o1 += u) units[1].push_back({n0 + o1, n1 - o1 < u ? n1 - o1 : u});
        HIPCHK(hipEventRecord(c->ev_fork, c->stream));             // pages / thresholds / earlier ranges are ordered before both lanes
        HIPCHK(hipStreamWaitEvent(c->half_stream[0], c->ev_fork, 0));
        HIPCHK(hipStreamWaitEvent(c->half_stream[1], c->ev_fork, 0));
        for (size_t i = 0; i < units[0].size() || i < units[1].size(); ++i)     // enqueue alternately: neither lane waits for the host
            for (int lane = 0; lane < 2; ++lane)
                if (i < units[lane].size() && run_chunk(lane, units[lane][i].first, units[lane][i].second, true)) return 1;
        for (int lane = 0; lane < 2; ++lane) {
            HIPCHK(hipEventRecord(c->ev_half[lane], c->half_stream[lane]));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_half[lane], 0));
        }
        return 0;
    }
    // chunks of equal size (108 tiles at max_batch 70 -> 54 + 54, not 70 + 38): launches shrink evenly.
    // (Profiling runs every launch alone on one lane: its chunks are capped at the size a LANE's launch has in normal operation -- half a
    // chunk -- so that the per-op times describe the launches the product runs, partial last rounds of the persistent grids included.)
    const int chunk_cap = (c->profiling && c->lanes == 2 && c->lane1_batch > 0 && c->max_batch >= 4 * kMinLaneTiles) ? (c->max_batch + 1) / 2 : c->max_batch;
    const int n_chunks = (n_tiles + chunk_cap - 1) / chunk_cap;
    const int chunk = n_chunks ? (n_tiles + n_chunks - 1) / n_chunks : 0;
    bool forked = false;
    for (int done = 0; done < n_tiles; done += chunk) {
        const int nb = n_tiles - done < chunk ? n_tiles - done : chunk;
        const bool two = c->lane1_batch > 0 && c->lanes == 2 && !c->profiling && nb >= 2 * kMinLaneTiles;
        if (!two) {
            if (forked) {                                       // (a one-lane chunk behind two-lane ones: lane 1 first)
                HIPCHK(hipEventRecord(c->ev_join, c->lane_stream));
                HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
                forked = false;
            }
            // (probe knob SBBSEG_PROFILE_HALF=1: the profiling pass runs on lane 0's HALF of the chip, the other half idle)
            static const bool prof_half = getenv("SBBSEG_PROFILE_HALF") && getenv("SBBSEG_PROFILE_HALF")[0] == '1';
            if (c->profiling && prof_half && c->cu_split) {
                HIPCHK(hipEventRecord(c->ev_fork, c->stream));
                HIPCHK(hipStreamWaitEvent(c->half_stream[0], c->ev_fork, 0));
                if (run_chunk(0, done, nb, true)) return 1;
                HIPCHK(hipEventRecord(c->ev_half[0], c->half_stream[0]));
                HIPCHK(hipStreamWaitEvent(c->stream, c->ev_half[0], 0));
                continue;
            }
            if (run_chunk(0, done, nb)) return 1;               // (not forked here: a fork in front was joined above)
            continue;
        }
        const int na = (nb + 1) / 2, nb2 = nb - na;            // nb2 <= lane1_batch
        // The lanes fork ONCE per tile range and join once behind its last chunk (round 5): a lane's halves of consecutive chunks follow each
        // other on the lane's own stream and buffers, nothing of one lane depends on the other.  (Up to round 4 every chunk forked and joined:
        // the lane that finished its half first waited for the other one -- 7-10 % of the timed region had ONE kernel in flight,
        // profiles/r05_timeline_gaps.txt.  SBBSEG_JOIN_PER_CHUNK=1 restores that for A/B.)
        static const bool join_per_chunk = getenv("SBBSEG_JOIN_PER_CHUNK") && getenv("SBBSEG_JOIN_PER_CHUNK")[0] == '1';
        if (!forked || join_per_chunk) {
            HIPCHK(hipEventRecord(c->ev_fork, c->stream));     // page / threshold / earlier ranges are ordered before
            HIPCHK(hipStreamWaitEvent(c->lane_stream, c->ev_fork, 0));
            forked = true;
        }
        // (a failure from here on must still join the lanes: earlier chunks of this range are in flight on lane_stream, unordered against
        //  whatever the caller does next on the handle's stream -- stitch, reuse of d_tile_labels, destroy)
        auto join_on_error = [&]() -> int {
            if (hipEventRecord(c->ev_join, c->lane_stream) != hipSuccess || hipStreamWaitEvent(c->stream, c->ev_join, 0) != hipSuccess)
                (void)hipStreamSynchronize(c->lane_stream);
            (void)hipGetLastError();
            return 1;
        };
        if (run_chunk(0, done, na)) return join_on_error();
        if (run_chunk(1, done + na, nb2)) return join_on_error();           // (starting lane 1 later -- after lane 0's op k -- measured 3-14 % slower)
        if (join_per_chunk) {
            HIPCHK(hipEventRecord(c->ev_join, c->lane_stream));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
            forked = false;
        }
    }
    if (forked) {
        HIPCHK(hipEventRecord(c->ev_join, c->lane_stream));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    }
    return 0;
}

int sbbseg_segment_tile_range_dev(sbbseg_ctx* c, const void* d_page_hwc, int Hp, int Wp, int first_tile, int n_tiles,
                                  void* d_tile_labels)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    return tile_range_impl(c, &d_page_hwc, 1, Hp, Wp, nullptr, nullptr, Hp, Wp, first_tile, n_tiles, d_tile_labels, nullptr, false, c->owned_mode >= 2);
    API_END
}

static int prepare_owner(sbbseg_ctx* c, int Hp, int Wp, bool dedupe)
{
    if (c->own_Hp == Hp && c->own_Wp == Wp && c->own_dedupe == dedupe) return 0;
    alloc_check();
    std::vector<int> ox, oy, own_x, own_y;
    const int margin = margin_of(c->in_W);
    int nx = axis_tiles(Wp, c->in_W, margin, ox), ny = axis_tiles(Hp, c->in_H, margin, oy);
    REQUIRE(nx > 0 && ny > 0, "page %dx%d is smaller than the model input", Hp, Wp);
    if (dedupe) {                              // (fused_grid: the repeated last tile of an axis is not computed)
        if (nx >= 2 && ox[nx - 1] == ox[nx - 2]) ox.resize(--nx);
        if (ny >= 2 && oy[ny - 1] == oy[ny - 2]) oy.resize(--ny);
    }
    axis_owner(Wp, c->in_W, margin, ox, own_x);
    axis_owner(Hp, c->in_H, margin, oy, own_y);
    const size_t need = sizeof(int) * (size_t)(Hp > Wp ? Hp : Wp);
    if (c->own_cap < need) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->d_own_x) { HIPCHK(hipFree(c->d_own_x)); HIPCHK(hipFree(c->d_own_y)); c->device_bytes -= 2 * c->own_cap; }
        c->d_own_x = c->d_own_y = nullptr;
        if (dmalloc(c, (void**)&c->d_own_x, need) || dmalloc(c, (void**)&c->d_own_y, need)) return 1;
        c->own_cap = need;
    }
    HIPCHK(hipStreamSynchronize(c->stream));   // tables may still be in use by an earlier stitch
    HIPCHK(hipMemcpy(c->d_own_x, own_x.data(), sizeof(int) * Wp, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_own_y, own_y.data(), sizeof(int) * Hp, hipMemcpyHostToDevice));
    c->own_Hp = Hp; c->own_Wp = Wp; c->own_nyf = ny; c->own_dedupe = dedupe;
    return 0;
}

static int stitch_impl(sbbseg_ctx* c, const void* d_tile_labels, int Hp, int Wp, void* d_labels_hw, bool dedupe)
{
    REQUIRE(d_tile_labels && d_labels_hw, "bad arguments");
    if (prepare_owner(c, Hp, Wp, dedupe)) return 1;
    HIPCHK(launch_stitch((const uint8_t*)d_tile_labels, c->in_H, c->in_W, c->d_own_x, c->d_own_y, c->own_nyf, Hp, Wp,
                         (uint8_t*)d_labels_hw, c->stream));
    return 0;
}

int sbbseg_stitch_dev(sbbseg_ctx* c, const void* d_tile_labels, int Hp, int Wp, void* d_labels_hw)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    return stitch_impl(c, d_tile_labels, Hp, Wp, d_labels_hw, false);      // tile labels in the reference's call order
    API_END
}

int sbbseg_segment_page_dev(sbbseg_ctx* c, const void* d_page_hwc, int Hp, int Wp, void* d_labels_hw)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    int nx = 0, ny = 0;
    if (fused_grid(c, Hp, Wp, c->dedupe, &nx, &ny)) return 1;
    if (ensure(c, (void**)&c->d_tile_labels, &c->tile_labels_cap, (size_t)nx * ny * c->in_H * c->in_W)) return 1;
    if (tile_range_impl(c, &d_page_hwc, 1, Hp, Wp, nullptr, nullptr, Hp, Wp, 0, nx * ny, c->d_tile_labels, nullptr, c->dedupe, c->owned_mode >= 1)) return 1;
    return stitch_impl(c, c->d_tile_labels, Hp, Wp, d_labels_hw, c->dedupe);
    API_END
}

int sbbseg_segment_pages_dev(sbbseg_ctx* c, int n_pages, const void* const* d_pages_hwc, int Hp, int Wp, void* const* d_labels_hw)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(n_pages >= 1 && d_pages_hwc && d_labels_hw, "bad arguments");
    for (int k = 0; k < n_pages; ++k) REQUIRE(d_pages_hwc[k] && d_labels_hw[k], "null page / label pointer (page %d)", k);
    int nx = 0, ny = 0;
    if (fused_grid(c, Hp, Wp, c->dedupe, &nx, &ny)) return 1;
    const size_t per = (size_t)c->in_H * c->in_W, tpp = (size_t)nx * ny;
    // Pages are pooled in GROUPS whose tile count is a whole number of chunks where that is possible with few pages
    // (lcm(tiles per page, max_batch)), else about eight chunks: the tile-label scratch stays bounded by the group, not
    // by the caller's page count (64 pages of 4000x3000 would otherwise hold 1.4 GB of tile labels at once), and no
    // chunk is cut short except the very last.  Groups run back to back on the handle's stream (stream order protects the
    // scratch buffer that every group reuses).
    size_t a = tpp, b = (size_t)c->max_batch;
    while (b) { const size_t t = a % b; a = b; b = t; }                 // a = gcd
    size_t G = (size_t)c->max_batch / a;                                 // pages per group with G * tpp = lcm
    if (G > 32) G = (8 * (size_t)c->max_batch + tpp - 1) / tpp;
    if (c->cu_split && c->lanes == 2 && c->lane1_batch > 0) {
        // the staggered lanes run a whole group without a join: groups of about eight chunks (four units per lane and more)
        const size_t want = (8 * (size_t)c->max_batch + tpp - 1) / tpp;
        if (G < want) G = want;
    }
    if (G < 1) G = 1;
    if (G > (size_t)n_pages) G = (size_t)n_pages;
    REQUIRE(tpp * G < (size_t)1 << 30, "page group of %zu tiles is too large", tpp * G);
    if (ensure(c, (void**)&c->d_tile_labels, &c->tile_labels_cap, tpp * G * per)) return 1;
    for (size_t g0 = 0; g0 < (size_t)n_pages; g0 += G) {
        const size_t np = g0 + G <= (size_t)n_pages ? G : (size_t)n_pages - g0;
        if (tile_range_impl(c, d_pages_hwc + g0, (int)np, Hp, Wp, nullptr, nullptr, Hp, Wp, 0, (int)(tpp * np), c->d_tile_labels, nullptr, c->dedupe, c->owned_mode >= 1)) return 1;
        for (size_t k = 0; k < np; ++k)
            if (stitch_impl(c, c->d_tile_labels + k * tpp * per, Hp, Wp, d_labels_hw[g0 + k], c->dedupe)) return 1;
    }
    return 0;
    API_END
}

// do_prediction(patches=True) for several HOST pages of one size, pipelined in groups of as many pages as fill a chunk:
// while group g runs on the handle's stream, the caller's thread stages group g+1 into pinned memory and starts its upload on
// a copy stream, and the label planes of group g-1 come back on another.  Results equal n_pages sbbseg_segment_page calls.
int sbbseg_segment_pages(sbbseg_ctx* c, int n_pages, const uint8_t* const* pages_hwc, int Hp, int Wp, uint8_t* const* labels_hw)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(n_pages >= 1 && pages_hwc && labels_hw, "bad arguments");
    for (int k = 0; k < n_pages; ++k) REQUIRE(pages_hwc[k] && labels_hw[k], "null page / label pointer (page %d)", k);
    REQUIRE(Hp >= c->in_H && Wp >= c->in_W, "page %dx%d is smaller than the model input %dx%d (unsupported by the reference too, main.py:278-281)", Hp, Wp, c->in_H, c->in_W);
    int nx = 0, ny = 0;
    if (fused_grid(c, Hp, Wp, c->dedupe, &nx, &ny)) return 1;
    const size_t pix = (size_t)Hp * Wp, in_b = pix * 3, ch = c->label_channels == 3 ? 3 : 1, out_b = pix * ch;
    const size_t out3_b = (pix + 3) / 4 * 12;                          // launch_replicate3 writes whole 12-byte groups
    int G = c->max_batch / (nx * ny);
    G = G < 1 ? 1 : (G > n_pages ? n_pages : G);
    if (!c->pp_ready) {                                                   // set LAST: a failed creation is retried by the next call
        if (!c->copy_in) HIPCHK(hipStreamCreateWithFlags(&c->copy_in, hipStreamNonBlocking));
        if (!c->copy_out) HIPCHK(hipStreamCreateWithFlags(&c->copy_out, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            if (!c->pp_in[k]) HIPCHK(hipEventCreateWithFlags(&c->pp_in[k], hipEventDisableTiming));
            if (!c->pp_comp[k]) HIPCHK(hipEventCreateWithFlags(&c->pp_comp[k], hipEventDisableTiming));
            if (!c->pp_out[k]) HIPCHK(hipEventCreateWithFlags(&c->pp_out[k], hipEventDisableTiming));
        }
        c->pp_ready = true;
    }
    HIPCHK(hipStreamSynchronize(c->stream));                            // (buffers below may be re-allocated)
    // Every buffer has its own capacity in the units it was allocated in; a capacity is zeroed BEFORE its buffers are
    // freed and set only after all of them exist again, so a failed allocation leaves "nothing allocated", not a stale size.
    auto host_pair = [&](uint8_t* (&h)[2], size_t* cap, size_t bytes) -> int {
        if (*cap >= bytes) return 0;
        *cap = 0;
        for (int k = 0; k < 2; ++k) { (void)hipHostFree(h[k]); h[k] = nullptr; }
        for (int k = 0; k < 2; ++k) HIPCHK(hipHostMalloc((void**)&h[k], bytes, hipHostMallocDefault));
        *cap = bytes;
        return 0;
    };
    auto dev_pair = [&](uint8_t* (&d)[2], size_t* cap, size_t bytes) -> int {
        if (*cap >= bytes) return 0;
        const size_t old = *cap;
        *cap = 0;
        for (int k = 0; k < 2; ++k)
            if (d[k]) { (void)hipFree(d[k]); d[k] = nullptr; c->device_bytes -= old; }
        for (int k = 0; k < 2; ++k)
            if (dmalloc(c, (void**)&d[k], bytes)) return 1;
        *cap = bytes;
        return 0;
    };
    if (host_pair(c->pp_h_in, &c->pp_hin_cap, G * in_b)) return 1;
    if (dev_pair(c->pp_d_in, &c->pp_in_cap, G * in_b)) return 1;
    if (host_pair(c->pp_h_out, &c->pp_hout_cap, G * out_b)) return 1;
    if (dev_pair(c->pp_d_out, &c->pp_out_cap, G * (pix + 4))) return 1;   // one u8 plane (+4 bytes slack) per page, whatever label_channels is
    if (ch == 3 && dev_pair(c->pp_d_out3, &c->pp_out3_cap, G * out3_b)) return 1;
    const int n_groups = (n_pages + G - 1) / G;
    auto group_pages = [&](int g) { return g * G + G <= n_pages ? G : n_pages - g * G; };
    auto drain = [&](int g) -> int {                                      // labels of group g: staging -> caller
        const int slot = g & 1;
        HIPCHK(hipEventSynchronize(c->pp_out[slot]));
        for (int k = 0; k < group_pages(g); ++k) memcpy(labels_hw[g * G + k], c->pp_h_out[slot] + (size_t)k * out_b, out_b);
        return 0;
    };
    alloc_check();
    std::vector<const void*> d_pages(G);
    std::vector<void*> d_labels(G);
    for (int g = 0; g < n_groups; ++g) {
        const int slot = g & 1, np = group_pages(g);
        if (g >= 2 && drain(g - 2)) return 1;                             // frees this slot's staging and device buffers
        for (int k = 0; k < np; ++k) memcpy(c->pp_h_in[slot] + (size_t)k * in_b, pages_hwc[g * G + k], in_b);
        HIPCHK(hipMemcpyAsync(c->pp_d_in[slot], c->pp_h_in[slot], (size_t)np * in_b, hipMemcpyHostToDevice, c->copy_in));
        HIPCHK(hipEventRecord(c->pp_in[slot], c->copy_in));
        HIPCHK(hipStreamWaitEvent(c->stream, c->pp_in[slot], 0));
        for (int k = 0; k < np; ++k) {
            d_pages[k] = c->pp_d_in[slot] + (size_t)k * in_b;
            d_labels[k] = c->pp_d_out[slot] + (size_t)k * (pix + 4);
        }
        if (sbbseg_segment_pages_dev(c, np, d_pages.data(), Hp, Wp, d_labels.data())) return 1;
        const uint8_t* d_src = c->pp_d_out[slot];
        size_t d_stride = pix + 4;
        if (ch == 3) {
            for (int k = 0; k < np; ++k)
                HIPCHK(launch_replicate3(c->pp_d_out[slot] + (size_t)k * (pix + 4), c->pp_d_out3[slot] + (size_t)k * out3_b, pix, c->stream));
            d_src = c->pp_d_out3[slot];
            d_stride = out3_b;
        }
        HIPCHK(hipEventRecord(c->pp_comp[slot], c->stream));
        HIPCHK(hipStreamWaitEvent(c->copy_out, c->pp_comp[slot], 0));
        for (int k = 0; k < np; ++k)
            HIPCHK(hipMemcpyAsync(c->pp_h_out[slot] + (size_t)k * out_b, d_src + (size_t)k * d_stride, out_b, hipMemcpyDeviceToHost, c->copy_out));
        HIPCHK(hipEventRecord(c->pp_out[slot], c->copy_out));
    }
    for (int g = n_groups >= 2 ? n_groups - 2 : 0; g < n_groups; ++g)
        if (drain(g)) return 1;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
    API_END
}

int sbbseg_segment_page(sbbseg_ctx* c, const uint8_t* page_hwc, int Hp, int Wp, uint8_t* labels_hw)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(page_hwc && labels_hw, "bad arguments");
    REQUIRE(Hp >= c->in_H && Wp >= c->in_W, "page %dx%d is smaller than the model input %dx%d (unsupported by the reference too, main.py:278-281)", Hp, Wp, c->in_H, c->in_W);
    const size_t pix = (size_t)Hp * Wp;
    if (ensure(c, (void**)&c->d_page, &c->page_cap, pix * 3)) return 1;
    if (ensure(c, (void**)&c->d_page_labels, &c->page_labels_cap, pix + 4)) return 1;
    HIPCHK(hipMemcpyAsync(c->d_page, page_hwc, pix * 3, hipMemcpyHostToDevice, c->stream));
    if (sbbseg_segment_page_dev(c, c->d_page, Hp, Wp, c->d_page_labels)) return 1;
    if (labels_to_host(c, labels_hw, pix)) return 1;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
    API_END
}

int sbbseg_segment_page_scaled(sbbseg_ctx* c, const uint8_t* page_hwc, int Hs, int Ws, int Hp, int Wp, uint8_t* labels_hw)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(page_hwc && labels_hw && Hs > 0 && Ws > 0, "bad arguments");
    REQUIRE(Hp >= c->in_H && Wp >= c->in_W, "scaled page %dx%d is smaller than the model input %dx%d", Hp, Wp, c->in_H, c->in_W);
    const size_t spix = (size_t)Hs * Ws, pix = (size_t)Hp * Wp;
    if (ensure(c, (void**)&c->d_page, &c->page_cap, spix * 3)) return 1;
    if (ensure(c, (void**)&c->d_page_labels, &c->page_labels_cap, pix + 4)) return 1;
    std::vector<int> my, mx;
    nearest_map(Hs, Hp, my);               // scaled row -> stored row   (main.py:214 -> 112-113)
    nearest_map(Ws, Wp, mx);
    if (ensure(c, (void**)&c->d_map, &c->map_cap, sizeof(int) * (size_t)(Hp + Wp))) return 1;
    c->map_key[0] = 0;                     // (d_map is rewritten below: the crop path's cached maps are gone)
    int* d_my = c->d_map; int* d_mx = d_my + Hp;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(d_my, my.data(), sizeof(int) * Hp, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_mx, mx.data(), sizeof(int) * Wp, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpyAsync(c->d_page, page_hwc, spix * 3, hipMemcpyHostToDevice, c->stream));
    int nx = 0, ny = 0;
    if (fused_grid(c, Hp, Wp, c->dedupe, &nx, &ny)) return 1;
    if (ensure(c, (void**)&c->d_tile_labels, &c->tile_labels_cap, (size_t)nx * ny * c->in_H * c->in_W)) return 1;
    const void* pg_ = c->d_page;
    if (tile_range_impl(c, &pg_, 1, Hs, Ws, d_my, d_mx, Hp, Wp, 0, nx * ny, c->d_tile_labels, nullptr, c->dedupe, c->owned_mode >= 1)) return 1;
    if (stitch_impl(c, c->d_tile_labels, Hp, Wp, c->d_page_labels, c->dedupe)) return 1;
    if (labels_to_host(c, labels_hw, pix)) return 1;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
    API_END
}

int sbbseg_otsu_dev(sbbseg_ctx* c, const void* d_page_hwc, int Hp, int Wp, int* d_threshold)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(d_page_hwc && d_threshold && Hp > 0 && Wp > 0, "bad arguments");
    HIPCHK(launch_otsu((const uint8_t*)d_page_hwc, Wp, Hp, Wp, nullptr, nullptr, c->d_hist, d_threshold, c->num_cus, c->stream));
    return 0;
    API_END
}

int sbbseg_segment_tile_range_bin_dev(sbbseg_ctx* c, const void* d_page_hwc, int Hp, int Wp, int first_tile, int n_tiles,
                                      const int* d_threshold, void* d_tile_labels)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(d_threshold, "bad arguments");
    return tile_range_impl(c, &d_page_hwc, 1, Hp, Wp, nullptr, nullptr, Hp, Wp, first_tile, n_tiles, d_tile_labels, d_threshold, false, c->owned_mode >= 2);
    API_END
}

int sbbseg_segment_page_otsu(sbbseg_ctx* c, const uint8_t* page_hwc, int Hs, int Ws, int Hp, int Wp, uint8_t* labels_hw,
                             int* threshold)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(page_hwc && labels_hw && Hs > 0 && Ws > 0, "bad arguments");
    REQUIRE(Hp >= c->in_H && Wp >= c->in_W, "page %dx%d is smaller than the model input %dx%d", Hp, Wp, c->in_H, c->in_W);
    const size_t spix = (size_t)Hs * Ws, pix = (size_t)Hp * Wp;
    const bool scaled = Hs != Hp || Ws != Wp;
    if (ensure(c, (void**)&c->d_page, &c->page_cap, spix * 3)) return 1;
    if (ensure(c, (void**)&c->d_page_labels, &c->page_labels_cap, pix + 4)) return 1;
    int *d_my = nullptr, *d_mx = nullptr;
    if (scaled) {
        std::vector<int> my, mx;
        nearest_map(Hs, Hp, my);           // scaled row -> stored row   (main.py:214 -> 112-113)
        nearest_map(Ws, Wp, mx);
        if (ensure(c, (void**)&c->d_map, &c->map_cap, sizeof(int) * (size_t)(Hp + Wp))) return 1;
        c->map_key[0] = 0;
        d_my = c->d_map; d_mx = d_my + Hp;
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpy(d_my, my.data(), sizeof(int) * Hp, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_mx, mx.data(), sizeof(int) * Wp, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpyAsync(c->d_page, page_hwc, spix * 3, hipMemcpyHostToDevice, c->stream));
    int* d_thr = (int*)(c->d_hist + 256);
    HIPCHK(launch_otsu(c->d_page, Ws, Hp, Wp, d_my, d_mx, c->d_hist, d_thr, c->num_cus, c->stream));
    int nx = 0, ny = 0;
    if (fused_grid(c, Hp, Wp, c->dedupe, &nx, &ny)) return 1;
    if (ensure(c, (void**)&c->d_tile_labels, &c->tile_labels_cap, (size_t)nx * ny * c->in_H * c->in_W)) return 1;
    const void* pg_ = c->d_page;
    if (tile_range_impl(c, &pg_, 1, Hs, Ws, d_my, d_mx, Hp, Wp, 0, nx * ny, c->d_tile_labels, d_thr, c->dedupe, c->owned_mode >= 1)) return 1;
    if (stitch_impl(c, c->d_tile_labels, Hp, Wp, c->d_page_labels, c->dedupe)) return 1;
    if (labels_to_host(c, labels_hw, pix)) return 1;
    int thr = 0;
    HIPCHK(hipMemcpyAsync(&thr, d_thr, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (threshold) *threshold = thr;
    return 0;
    API_END
}

// The patch stages of run() on the CROPPED page (main.py:2061-2102: extract_page's croped_page goes to extract_text_regions and
// textline_contours): the crop box lives in the coordinates of the page as upscaled to Hp x Wp, the stored image is Hs x Ws.
// Rescale, crop and (binarise != 0) otsu_copy are all index arithmetic in the tile gather: row r / column q of the crop read
// stored row map_y[cy + r] / column map_x[cx + q]; the Otsu histogram is taken over exactly those pixels (channel 0).
int sbbseg_segment_crop_dev(sbbseg_ctx* c, const void* d_page_hwc, int Hs, int Ws, int Hp, int Wp, int cx, int cy, int cw, int ch,
                            int binarise, void* d_labels_hw, int* d_threshold)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(d_page_hwc && d_labels_hw && Hs > 0 && Ws > 0 && Hp > 0 && Wp > 0, "bad arguments");
    REQUIRE(cx >= 0 && cy >= 0 && cw > 0 && ch > 0 && cx + cw <= Wp && cy + ch <= Hp, "crop box {%d,%d,%d,%d} leaves the %dx%d page", cx, cy, cw, ch, Hp, Wp);
    REQUIRE(ch >= c->in_H && cw >= c->in_W, "cropped page %dx%d is smaller than the model input %dx%d (unsupported by the reference too, main.py:278-281)",
            ch, cw, c->in_H, c->in_W);
    if (c->map_key[0] != Hs || c->map_key[1] != Ws || c->map_key[2] != Hp || c->map_key[3] != Wp || !c->d_map) {
        std::vector<int> my, mx;
        nearest_map(Hs, Hp, my);           // scaled row -> stored row   (main.py:214 -> 112-113); identity when Hs == Hp
        nearest_map(Ws, Wp, mx);
        if (ensure(c, (void**)&c->d_map, &c->map_cap, sizeof(int) * (size_t)(Hp + Wp))) return 1;
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpy(c->d_map, my.data(), sizeof(int) * Hp, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_map + Hp, mx.data(), sizeof(int) * Wp, hipMemcpyHostToDevice));
        c->map_key[0] = Hs; c->map_key[1] = Ws; c->map_key[2] = Hp; c->map_key[3] = Wp;
    }
    const int* d_my = c->d_map + cy;
    const int* d_mx = c->d_map + Hp + cx;
    int* d_thr = nullptr;
    if (binarise) {
        d_thr = d_threshold ? d_threshold : (int*)(c->d_hist + 256);
        HIPCHK(launch_otsu((const uint8_t*)d_page_hwc, Ws, ch, cw, d_my, d_mx, c->d_hist, d_thr, c->num_cus, c->stream));
    }
    int nx = 0, ny = 0;
    if (fused_grid(c, ch, cw, c->dedupe, &nx, &ny)) return 1;
    if (ensure(c, (void**)&c->d_tile_labels, &c->tile_labels_cap, (size_t)nx * ny * c->in_H * c->in_W)) return 1;
    if (tile_range_impl(c, &d_page_hwc, 1, Hs, Ws, d_my, d_mx, ch, cw, 0, nx * ny, c->d_tile_labels, d_thr, c->dedupe, c->owned_mode >= 1)) return 1;
    return stitch_impl(c, c->d_tile_labels, ch, cw, d_labels_hw, c->dedupe);
    API_END
}

int sbbseg_segment_crop(sbbseg_ctx* c, const uint8_t* page_hwc, int Hs, int Ws, int Hp, int Wp, int cx, int cy, int cw, int ch,
                        int binarise, uint8_t* labels_hw, int* threshold)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(page_hwc && labels_hw && Hs > 0 && Ws > 0 && cw > 0 && ch > 0, "bad arguments");
    const size_t spix = (size_t)Hs * Ws, pix = (size_t)ch * cw;
    if (ensure(c, (void**)&c->d_page, &c->page_cap, spix * 3)) return 1;
    if (ensure(c, (void**)&c->d_page_labels, &c->page_labels_cap, pix + 4)) return 1;
    HIPCHK(hipMemcpyAsync(c->d_page, page_hwc, spix * 3, hipMemcpyHostToDevice, c->stream));
    if (sbbseg_segment_crop_dev(c, c->d_page, Hs, Ws, Hp, Wp, cx, cy, cw, ch, binarise, c->d_page_labels, nullptr)) return 1;
    if (labels_to_host(c, labels_hw, pix)) return 1;
    int thr = 0;
    if (binarise) HIPCHK(hipMemcpyAsync(&thr, (int*)(c->d_hist + 256), sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (threshold) *threshold = thr;
    return 0;
    API_END
}

int sbbseg_segment_whole(sbbseg_ctx* c, const uint8_t* page_hwc, int Hp, int Wp, int out_h, int out_w, uint8_t* labels_out)
{
    API_BEGIN
    return sbbseg_segment_whole_scaled(c, page_hwc, Hp, Wp, Hp, Wp, out_h, out_w, labels_out);
    API_END
}

// do_prediction(patches=False) on the page as upscaled to Hs x Ws: the page comes from host memory (page_hwc) or is already on the
// device (d_page_in); the label plane lands in c->d_page_labels and, if labels_out is given, in host memory too.
static int whole_scaled_impl(sbbseg_ctx* c, const uint8_t* page_hwc, const void* d_page_in, int Hp, int Wp, int Hs, int Ws, int out_h, int out_w,
                             uint8_t* labels_out)
{
    const size_t pix = (size_t)Hp * Wp, opix = (size_t)out_h * out_w;
    if (!d_page_in && ensure(c, (void**)&c->d_page, &c->page_cap, pix * 3)) return 1;
    if (ensure(c, (void**)&c->d_page_labels, &c->page_labels_cap, opix + 4)) return 1;
    const size_t need = sizeof(int) * (size_t)(c->in_H + c->in_W + out_h + out_w);
    const bool cached = c->d_wmap && c->wmap_cap >= need && c->wmap_key[0] == Hp && c->wmap_key[1] == Wp && c->wmap_key[2] == Hs &&
                        c->wmap_key[3] == Ws && c->wmap_key[4] == out_h && c->wmap_key[5] == out_w;
    if (!cached) {                         // the four gather tables depend on the sizes only: built and uploaded once per geometry
        std::vector<int> my, mx, oy, ox;
        nearest_map(Hs, c->in_H, my);      // model row  -> row of the page do_prediction was handed (main.py:371)
        nearest_map(Ws, c->in_W, mx);
        if (Hs != Hp || Ws != Wp) {        // that page is itself the nearest-upscaled stored image (main.py:214): compose
            std::vector<int> sy, sx;
            nearest_map(Hp, Hs, sy);       // scaled row -> stored row
            nearest_map(Wp, Ws, sx);
            for (auto& v : my) v = sy[v];
            for (auto& v : mx) v = sx[v];
        }
        nearest_map(c->in_H, out_h, oy);   // output row -> model row  (main.py:378)
        nearest_map(c->in_W, out_w, ox);
        if (ensure(c, (void**)&c->d_wmap, &c->wmap_cap, need)) return 1;
        c->wmap_key[0] = 0;
        int* d_my = c->d_wmap; int* d_mx = d_my + c->in_H; int* d_oy = d_mx + c->in_W; int* d_ox = d_oy + out_h;
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpy(d_my, my.data(), sizeof(int) * c->in_H, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_mx, mx.data(), sizeof(int) * c->in_W, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_oy, oy.data(), sizeof(int) * out_h, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_ox, ox.data(), sizeof(int) * out_w, hipMemcpyHostToDevice));
        c->wmap_key[0] = Hp; c->wmap_key[1] = Wp; c->wmap_key[2] = Hs; c->wmap_key[3] = Ws; c->wmap_key[4] = out_h; c->wmap_key[5] = out_w;
    }
    int* d_my = c->d_wmap; int* d_mx = d_my + c->in_H; int* d_oy = d_mx + c->in_W; int* d_ox = d_oy + out_h;
    if (!d_page_in) HIPCHK(hipMemcpyAsync(c->d_page, page_hwc, pix * 3, hipMemcpyHostToDevice, c->stream));
    IngestParams ip;
    if (fill_ingest(c, ip)) return 1;
    ip.page = d_page_in ? (const uint8_t*)d_page_in : c->d_page; ip.Hp = Hp; ip.Wp = Wp; ip.src_Hp = Hp; ip.src_Wp = Wp; ip.tile_xy = nullptr; ip.n_tiles = 1;
    ip.whole = 1; ip.map_y = d_my; ip.map_x = d_mx;
    HIPCHK(launch_ingest_u8(ip, c->precision, c->stream));
    {
        struct KsplitScope {               // reset on every way out (a throwing run_plan included): later n == 1 launches must not split
            sbbseg_ctx* c;
            explicit KsplitScope(sbbseg_ctx* c_) : c(c_) { c->ksplit_now = c->ksplit; }
            ~KsplitScope() { c->ksplit_now = false; }
        } scope(c);
        if (run_plan(c, 1, c->d_batch_labels, nullptr)) return 1;
    }
    HIPCHK(launch_resize_labels(c->d_batch_labels, c->in_H, c->in_W, d_oy, d_ox, out_h, out_w, c->d_page_labels, c->stream));
    if (labels_out) {
        if (labels_to_host(c, labels_out, opix)) return 1;
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int sbbseg_segment_whole_scaled(sbbseg_ctx* c, const uint8_t* page_hwc, int Hp, int Wp, int Hs, int Ws, int out_h, int out_w,
                                uint8_t* labels_out)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(page_hwc && labels_out && Hp > 0 && Wp > 0 && Hs > 0 && Ws > 0 && out_h > 0 && out_w > 0, "bad arguments");
    return whole_scaled_impl(c, page_hwc, nullptr, Hp, Wp, Hs, Ws, out_h, out_w, labels_out);
    API_END
}

// ------------------------------------------------------------------------------ stage glue (8f-3)
int sbbseg_morph_dev(sbbseg_ctx* c, const void* d_src_hw, int H, int W, int op, int ksize, int iterations, void* d_dst_hw)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(d_src_hw && d_dst_hw && H > 0 && W > 0, "bad arguments");
    REQUIRE((op == SBBSEG_MORPH_ERODE || op == SBBSEG_MORPH_DILATE) && ksize >= 1 && (ksize & 1) && iterations >= 1, "morph: op 0|1, odd kernel, iterations >= 1");
    const size_t pix = (size_t)H * W;
    if (ensure(c, (void**)&c->d_morph_a, &c->morph_a_cap, pix)) return 1;
    HIPCHK(launch_morph((const uint8_t*)d_src_hw, c->d_morph_a, (uint8_t*)d_dst_hw, H, W, (ksize - 1) / 2 * iterations, op == SBBSEG_MORPH_DILATE, 0, c->stream));
    return 0;
    API_END
}

int sbbseg_morph(sbbseg_ctx* c, const uint8_t* src_hw, int H, int W, int op, int ksize, int iterations, uint8_t* dst_hw)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(src_hw && dst_hw && H > 0 && W > 0, "bad arguments");
    const size_t pix = (size_t)H * W;
    if (ensure(c, (void**)&c->d_morph_b, &c->morph_b_cap, pix)) return 1;
    HIPCHK(hipMemcpyAsync(c->d_morph_b, src_hw, pix, hipMemcpyHostToDevice, c->stream));
    if (sbbseg_morph_dev(c, c->d_morph_b, H, W, op, ksize, iterations, c->d_morph_b)) return 1;
    HIPCHK(hipMemcpyAsync(dst_hw, c->d_morph_b, pix, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
    API_END
}

// The contour ranking behind sbbseg_page_box_dev and sbbseg_text_regions_present_dev: 8-connected components of the 0 / 255 plane in
// c->d_morph_b, ranked by the area of their outer contour (cv2.contourArea of cv2.findContours' outer borders).  The device ranks by a
// lower bound of that area and checks the winner against every other component's bounding-box bound (launch_largest_contour); when
// that leaves the ranking open -- or when the caller needs the winner's EXACT area (`exact`) -- the host walks the outer borders of
// the candidates on the device's label plane (parent[i] = root = the component's first pixel in raster order): 4 bytes per pixel of
// D2H + the candidates' perimeters.  box = {x0, y0, x1, y1, pixels} of the winner; *any = false for an empty plane; *area2 = twice the
// winner's contour area (exact when traced, else the device's lower bound; *traced says which).
// exact_below2: trace on the host (exact area) also when the device's lower bound of TWICE the winner's area is below this -- the caller's
// threshold: one labelling pass and one copy of the label plane decide "too small", not two
static int rank_contours(sbbseg_ctx* c, int H, int W, bool exact, int (&box)[5], bool* any, long long* area2, bool* traced, double exact_below2 = -1.0)
{
    const size_t pix = (size_t)H * W;
    if (ensure(c, (void**)&c->d_cc_parent, &c->cc_parent_cap, pix * sizeof(int)) || ensure(c, (void**)&c->d_cc_count, &c->cc_count_cap, pix * sizeof(int))) return 1;
    if (ensure(c, (void**)&c->d_cc_aux, &c->cc_aux_cap, 5 * pix * sizeof(int))) return 1;
    if (!c->d_cc_small && dmalloc(c, (void**)&c->d_cc_small, 4 * sizeof(unsigned long long))) return 1;
    if (!c->d_cc_list && dmalloc(c, (void**)&c->d_cc_list, (6 + kCcMaxRivals) * sizeof(int))) return 1;
    int* d_out = c->d_cc_list;
    int* aux = c->d_cc_aux;
    HIPCHK(launch_largest_contour(c->d_morph_b, H, W, c->d_cc_parent, c->d_cc_count, aux, aux + pix, aux + 2 * pix, aux + 3 * pix, aux + 4 * pix,
                                  c->d_cc_small, d_out, c->stream));
    int out[6 + kCcMaxRivals];
    unsigned long long key = 0;
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&key, c->d_cc_small, sizeof(key), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int q = 0; q < 5; ++q) box[q] = out[q];
    *any = out[2] >= 0;
    *area2 = (long long)(key >> 32);
    *traced = false;
    if (*any && (out[5] > 0 || exact || c->force_host_contours || (double)*area2 < exact_below2)) {
        alloc_check();
        std::vector<int> lab(pix);
        HIPCHK(hipMemcpy(lab.data(), c->d_cc_parent, pix * sizeof(int), hipMemcpyDeviceToHost));
        std::vector<int> cand;
        cand.push_back((int)((unsigned)(key & 0xffffffffu) - 1u));
        if (out[5] <= kCcMaxRivals && !c->force_host_contours) cand.insert(cand.end(), out + 6, out + 6 + out[5]);
        else
            for (size_t i = 0; i < pix; ++i)
                if (lab[i] == (int)i && (int)i != cand[0]) cand.push_back((int)i);          // every root
        long long best_area2 = -1;
        int best_root = -1;
        for (int root : cand) {
            auto inside = [&](int y, int x) { return (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && lab[(size_t)y * W + x] == root; };
            int tb[4];
            const long long a2 = trace_outer_area2(inside, root / W, root % W, (long)pix, tb);
            if (a2 > best_area2 || (a2 == best_area2 && root > best_root)) {           // ties: the later root (see host_largest_contour)
                best_area2 = a2; best_root = root;
                box[0] = tb[0]; box[1] = tb[1]; box[2] = tb[2]; box[3] = tb[3];
            }
        }
        HIPCHK(hipMemcpy(&box[4], c->d_cc_count + best_root, sizeof(int), hipMemcpyDeviceToHost));
        c->host_contour_calls += 1;
        *area2 = best_area2;
        *traced = true;
    }
    return 0;
}

int sbbseg_page_box_dev(sbbseg_ctx* c, const void* d_mask_hw, int H, int W, int32_t* box_xywh, int64_t* pixels)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(d_mask_hw && box_xywh && H > 0 && W > 0, "bad arguments");
    REQUIRE((size_t)H * W < ((size_t)1 << 31), "mask too large for 32-bit pixel indices");
    const size_t pix = (size_t)H * W;
    if (ensure(c, (void**)&c->d_morph_a, &c->morph_a_cap, pix) || ensure(c, (void**)&c->d_morph_b, &c->morph_b_cap, pix)) return 1;
    // main.py:394-398: gray > 0 -> 255, dilate with the 5x5 kernel of ones, 6 iterations (= one clipped 25x25 maximum)
    HIPCHK(launch_morph((const uint8_t*)d_mask_hw, c->d_morph_a, c->d_morph_b, H, W, 12, 1, 1, c->stream));
    // main.py:398-404: the contour with the largest cv2.contourArea
    int box[5];
    bool any = false, traced = false;
    long long area2 = 0;
    if (rank_contours(c, H, W, false, box, &any, &area2, &traced)) return 1;
    if (pixels) *pixels = any ? (int64_t)box[4] : 0;
    if (!any) {                                        // empty mask: the reference's np.argmax of an empty list raises (main.py:399-401)
        box_xywh[0] = box_xywh[1] = box_xywh[2] = box_xywh[3] = 0;
        return 0;
    }
    box_xywh[0] = box[0]; box_xywh[1] = box[1]; box_xywh[2] = box[2] - box[0] + 1; box_xywh[3] = box[3] - box[1] + 1;   // cv2.boundingRect
    return 0;
    API_END
}

// get_text_region_contours_and_boxes' EXISTENCE test (main.py:456-480, the `if len(contours) > 0` that gates the textline model,
// main.py:2083-2096): class mask (all channels == label -> 255), MORPH_OPEN, MORPH_CLOSE with the 5x5 kernel, findContours(RETR_TREE),
// keep the contours without a parent whose polygon area lies in [min_area, max_area = 1] x H x W.  After OPEN and CLOSE every
// component and every hole is a union of 5x5 squares, so no contour has fewer than three points (the `jv` bookkeeping of
// filter_contours_area_of_image, main.py:81-91, never drifts), and the largest outer contour of the plane is always a parentless one
// (a component nested in a hole is smaller than the component around it): contours exist <=> the largest outer-contour area
// reaches min_area * H * W.  The contours themselves (polygons, boxes) are out of scope.
int sbbseg_text_regions_present_dev(sbbseg_ctx* c, const void* d_regions_hw, int H, int W, int label, double min_area, int* present)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(d_regions_hw && present && H > 0 && W > 0 && label >= 0 && label <= 255 && min_area >= 0.0, "bad arguments");
    REQUIRE((size_t)H * W < ((size_t)1 << 31), "plane too large for 32-bit pixel indices");
    const size_t pix = (size_t)H * W;
    if (ensure(c, (void**)&c->d_morph_a, &c->morph_a_cap, pix) || ensure(c, (void**)&c->d_morph_b, &c->morph_b_cap, pix)) return 1;
    // OPEN = erode, dilate; CLOSE = dilate, erode (cv2.morphologyEx, one iteration each, default border: outside pixels never win);
    // the two dilations in a row are one clipped 9x9 maximum
    HIPCHK(launch_morph((const uint8_t*)d_regions_hw, c->d_morph_a, c->d_morph_b, H, W, 2, 0, 0x100 | label, c->stream));
    HIPCHK(launch_morph(c->d_morph_b, c->d_morph_a, c->d_morph_b, H, W, 4, 1, 0, c->stream));
    HIPCHK(launch_morph(c->d_morph_b, c->d_morph_a, c->d_morph_b, H, W, 2, 0, 0, c->stream));
    const double need = min_area * (double)((long long)H * W);            // main.py:87: area >= min_area * np.prod(image.shape[:2])
    int box[5];
    bool any = false, traced = false;
    long long area2 = 0;
    // the device's figure is a lower bound (holes not filled): only the exact area can say "too small" -- traced in the same pass
    if (rank_contours(c, H, W, false, box, &any, &area2, &traced, 2.0 * need)) return 1;
    *present = (any && (double)area2 * 0.5 >= need) ? 1 : 0;
    return 0;
    API_END
}

// ---- device buffers for callers that have no device runtime of their own (the reference's environment is Keras/TF, not PyTorch):
// what run() keeps resident across its three stages -- the stored page, the border mask, the region map, the textline map -- lives in
// buffers the library hands out.  They belong to the handle that allocated them (sbbseg_destroy frees what is left) but any handle of
// the same device may read and write them.
int sbbseg_device_alloc(sbbseg_ctx* c, size_t bytes, void** d_ptr)
{
    API_BEGIN
    REQUIRE(c != nullptr && d_ptr != nullptr && bytes > 0, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    alloc_check();
    void* p = nullptr;
    c->user_bufs.reserve(c->user_bufs.size() + 1);
    if (dmalloc(c, &p, bytes)) return 1;
    c->user_bufs.push_back({p, bytes});
    *d_ptr = p;
    return 0;
    API_END
}

int sbbseg_device_free(sbbseg_ctx* c, void* d_ptr)
{
    API_BEGIN
    REQUIRE(c != nullptr, "null handle");
    if (!d_ptr) return 0;
    for (size_t i = 0; i < c->user_bufs.size(); ++i)
        if (c->user_bufs[i].first == d_ptr) {
            HIPCHK(hipSetDevice(c->device));
            HIPCHK(hipDeviceSynchronize());            // DEVICE-wide on purpose: any handle on the device may use the buffer (sbbseg.h), so
                                                       // other handles' streams may still be reading it; a free is not a hot-path call
            HIPCHK(hipFree(d_ptr));
            c->device_bytes -= c->user_bufs[i].second;
            c->user_bufs.erase(c->user_bufs.begin() + (long)i);
            return 0;
        }
    return fail("sbbseg_device_free: %p was not allocated by this handle", d_ptr);
    API_END
}

// host -> device / device -> host on the handle's stream; both return when the copy is complete, so the buffer may be handed to
// another handle (another stream) right away.
int sbbseg_upload(sbbseg_ctx* c, void* d_dst, const void* src, size_t bytes)
{
    API_BEGIN
    REQUIRE(c != nullptr && d_dst && src, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    if (bytes) HIPCHK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
    API_END
}

int sbbseg_download(sbbseg_ctx* c, void* dst, const void* d_src, size_t bytes)
{
    API_BEGIN
    REQUIRE(c != nullptr && dst && d_src, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    if (bytes) HIPCHK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
    API_END
}

// a u8 label plane [pixels] on the device -> host memory, as one plane or as the three identical channels do_prediction returns
// (main.py:366; replicated on the device, sbbseg_set_label_channels is not consulted)
int sbbseg_download_labels(sbbseg_ctx* c, uint8_t* dst, const void* d_labels_hw, size_t pixels, int channels)
{
    API_BEGIN
    REQUIRE(c != nullptr && dst && d_labels_hw && pixels > 0 && (channels == 1 || channels == 3), "bad arguments (channels: 1 or 3)");
    HIPCHK(hipSetDevice(c->device));
    if (channels == 3) {
        if (ensure(c, (void**)&c->d_page_labels3, &c->page_labels3_cap, (pixels + 3) / 4 * 12)) return 1;
        HIPCHK(launch_replicate3((const uint8_t*)d_labels_hw, c->d_page_labels3, pixels, c->stream));
        HIPCHK(hipMemcpyAsync(dst, c->d_page_labels3, pixels * 3, hipMemcpyDeviceToHost, c->stream));
    } else {
        HIPCHK(hipMemcpyAsync(dst, d_labels_hw, pixels, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
    API_END
}

int sbbseg_extract_page_box(sbbseg_ctx* c, const uint8_t* page_hwc, int Hp, int Wp, int Hs, int Ws, uint8_t* mask_out, int32_t* box_xywh,
                            int64_t* pixels)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(page_hwc && box_xywh && Hp > 0 && Wp > 0 && Hs > 0 && Ws > 0, "bad arguments");
    // border model on the (virtually) upscaled page, result at the upscaled size (main.py:384-392) ...  (mask_out == NULL: the mask --
    // a local of extract_page in the reference -- stays on the device)
    if (whole_scaled_impl(c, page_hwc, nullptr, Hp, Wp, Hs, Ws, Hs, Ws, mask_out)) return 1;
    // ... whose label plane is still in d_page_labels: threshold, dilate x 6, largest component, bounding box (main.py:394-404)
    return sbbseg_page_box_dev(c, c->d_page_labels, Hs, Ws, box_xywh, pixels);
    API_END
}

int sbbseg_extract_page_box_dev(sbbseg_ctx* c, const void* d_page_hwc, int Hp, int Wp, int Hs, int Ws, void* d_mask_out, int32_t* box_xywh,
                                int64_t* pixels)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(d_page_hwc && box_xywh && Hp > 0 && Wp > 0 && Hs > 0 && Ws > 0, "bad arguments");
    if (whole_scaled_impl(c, nullptr, d_page_hwc, Hp, Wp, Hs, Ws, Hs, Ws, nullptr)) return 1;
    if (d_mask_out) HIPCHK(hipMemcpyAsync(d_mask_out, c->d_page_labels, (size_t)Hs * Ws, hipMemcpyDeviceToDevice, c->stream));
    return sbbseg_page_box_dev(c, c->d_page_labels, Hs, Ws, box_xywh, pixels);
    API_END
}

// ---- the model-running part of run() (main.py:2056-2107) in ONE call: the stored page is uploaded once and stays in device memory for
// all three stages, the border mask and the region map never leave the device between their model and their glue.
//   extract_page (main.py:2061, 384-437)            border model on the page as upscaled to Hs x Ws, dilate x 6, largest contour, box;
//                                                   outside the reference's try: an error here is the call's error
//   extract_text_regions (2072, 439-454)            layout model on the Otsu'd CROP, erode x 3 / dilate x 4 (2074-2075); a failure --
//                                                   e.g. a crop smaller than the model input, main.py:278-285 -- is "no regions" (2089-2091)
//   get_text_region_contours_and_boxes (2083, 2096) existence only: the textline model runs when a contour would be kept
//   textline_contours (2102, 490-503)               textline model on the crop
int sbbseg_run_page(sbbseg_ctx* border, sbbseg_ctx* layout, sbbseg_ctx* textline, const uint8_t* page_hwc, int Hp, int Wp, int Hs, int Ws,
                    int channels, uint8_t* page_mask_out, uint8_t* regions_out, uint8_t* textlines_out, sbbseg_run_info* info)
{
    API_BEGIN
    if (check_ready(border) || check_ready(layout) || check_ready(textline)) return 1;
    REQUIRE(page_hwc && info && regions_out && textlines_out && Hp > 0 && Wp > 0 && Hs > 0 && Ws > 0 && (channels == 1 || channels == 3), "bad arguments");
    REQUIRE(border->device == layout->device && border->device == textline->device, "the three handles must live on one device");
    memset(info, 0, sizeof(*info));
    const size_t spix = (size_t)Hp * Wp, pix = (size_t)Hs * Ws;
    if (ensure(border, (void**)&border->d_run_page, &border->run_page_cap, spix * 3)) return 1;
    if (page_mask_out && ensure(border, (void**)&border->d_run_mask, &border->run_mask_cap, pix)) return 1;
    HIPCHK(hipMemcpyAsync(border->d_run_page, page_hwc, spix * 3, hipMemcpyHostToDevice, border->stream));      // the one upload of the page
    int64_t pixels = 0;
    if (sbbseg_extract_page_box_dev(border, border->d_run_page, Hp, Wp, Hs, Ws, page_mask_out ? border->d_run_mask : nullptr, info->box_xywh, &pixels)) return 1;
    info->box_pixels = pixels;                                  // (extract_page_box_dev has synchronised the border handle's stream)
    REQUIRE(pixels > 0, "attempt to get argmax of an empty sequence (the border model found no page: main.py:399-401 raises here)");
    if (page_mask_out && sbbseg_download_labels(border, page_mask_out, border->d_run_mask, pix, channels)) return 1;
    const int x = info->box_xywh[0], y = info->box_xywh[1], w = info->box_xywh[2], h = info->box_xywh[3];
    const size_t cpix = (size_t)w * h;
    // A failed stage: wait for whatever it had queued (its kernels read border->d_run_page and the handle's d_run_a / d_run_b, which the next
    // call overwrites / may reallocate), then decide what the failure is.  Only a crop that cannot hold one model input -- what makes the
    // reference's own loop raise inside its bare try / except (main.py:278-281 under 2069-2091 / 2152-2157) -- degrades to "no regions" /
    // "no lines"; a device or allocation error fails the call with its message in sbbseg_last_error(): a fault must not look like an empty page.
    auto stage_failed = [&](sbbseg_ctx* h) -> bool {          // true: a geometry failure (swallowed, as the reference swallows it)
        const std::string msg = g_err;
        (void)hipStreamSynchronize(h->stream);
        if (h->lane_stream) (void)hipStreamSynchronize(h->lane_stream);
        (void)hipGetLastError();
        g_err = msg;
        return msg.find("smaller than the model input") != std::string::npos;
    };
    // layout stage + its post-processing: the reference's bare try / except (main.py:2069-2091)
    int present = 0;
    bool ok = false;
    do {
        if (ensure(layout, (void**)&layout->d_run_a, &layout->run_a_cap, cpix + 4) || ensure(layout, (void**)&layout->d_run_b, &layout->run_b_cap, cpix + 4)) break;
        int* d_thr = (int*)(layout->d_hist + 256);
        if (sbbseg_segment_crop_dev(layout, border->d_run_page, Hp, Wp, Hs, Ws, x, y, w, h, 1, layout->d_run_a, d_thr)) break;
        if (sbbseg_morph_dev(layout, layout->d_run_a, h, w, SBBSEG_MORPH_ERODE, 5, 3, layout->d_run_b)) break;          // main.py:2074
        if (sbbseg_morph_dev(layout, layout->d_run_b, h, w, SBBSEG_MORPH_DILATE, 5, 4, layout->d_run_b)) break;         // main.py:2075
        if (sbbseg_text_regions_present_dev(layout, layout->d_run_b, h, w, 1, 0.00001, &present)) break;                 // main.py:2083, 2096
        if (sbbseg_download(layout, &info->otsu_threshold, d_thr, sizeof(int))) break;
        if (sbbseg_download_labels(layout, regions_out, layout->d_run_b, cpix, channels)) break;
        info->regions_ok = 1;
        ok = true;
    } while (0);
    if (!ok && !stage_failed(layout)) return 1;
    info->text_present = info->regions_ok ? present : 0;
    if (info->text_present) {                                  // main.py:2096-2107; a failure = the outer except (2152-2157): no lines
        ok = false;
        do {
            if (ensure(textline, (void**)&textline->d_run_a, &textline->run_a_cap, cpix + 4)) break;
            if (sbbseg_segment_crop_dev(textline, border->d_run_page, Hp, Wp, Hs, Ws, x, y, w, h, 0, textline->d_run_a, nullptr)) break;
            if (sbbseg_download_labels(textline, textlines_out, textline->d_run_a, cpix, 1)) break;
            info->textlines_ok = 1;
            ok = true;
        } while (0);
        if (!ok && !stage_failed(textline)) return 1;
    }
    return 0;
    API_END
}

// ---- stage glue: the rotate-and-project of the deskew search (main.py:1601-1718) ----------------------------------
int sbbseg_deskew_side(int H, int W, int* side)
{
    API_BEGIN
    REQUIRE(side && H > 0 && W > 0, "bad arguments");
    *side = (int)((double)(H > W ? H : W) * 1.4);              // main.py:1613  int(max_x_y * (1.4))
    return 0;
    API_END
}

// cv2.getRotationMatrix2D(center, angle, 1.0) [EXT OpenCV 4.5.1]: positive angle = counter-clockwise
int sbbseg_rotation_matrix(double cx, double cy, double angle_deg, double* m6)
{
    API_BEGIN
    REQUIRE(m6, "bad arguments");
    const double a = angle_deg * 3.14159265358979323846 / 180.0;
    const double alpha = std::cos(a), beta = std::sin(a);
    m6[0] = alpha; m6[1] = beta; m6[2] = (1 - alpha) * cx - beta * cy;
    m6[3] = -beta; m6[4] = alpha; m6[5] = beta * cx + (1 - alpha) * cy;
    return 0;
    API_END
}

static void invert_affine(const double* M, double* o)
{
#pragma clang fp contract(off)
    // the in-place inversion of cv::warpAffine (no WARP_INVERSE_MAP), same operation order
    double m[6] = {M[0], M[1], M[2], M[3], M[4], M[5]};
    double D = m[0] * m[4] - m[1] * m[3];
    D = D != 0 ? 1.0 / D : 0.0;
    const double A11 = m[4] * D, A22 = m[0] * D;
    m[0] = A11; m[1] *= -D; m[3] *= -D; m[4] = A22;
    const double b1 = -m[0] * m[2] - m[1] * m[5];
    const double b2 = -m[3] * m[2] - m[4] * m[5];
    m[2] = b1; m[5] = b2;
    for (int i = 0; i < 6; ++i) o[i] = m[i];
}

static void cubic_table(float* tab)
{
#pragma clang fp contract(off)
    // interpolateCubic of imgwarp.cpp, A = -0.75, float arithmetic
    const float A = -0.75f;
    for (int i = 0; i < 32; ++i) {
        const float x = (float)i * (1.0f / 32);
        float* c = tab + i * 4;
        c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
        c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
        c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
    }
}

int sbbseg_deskew_profiles_dev(sbbseg_ctx* c, const void* d_mask_hw, int H, int W, const double* matrices, const double* angles_deg,
                               int n_angles, int32_t* counts)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(d_mask_hw && counts && H > 0 && W > 0 && n_angles >= 1 && n_angles <= 4096 && (matrices || angles_deg), "bad arguments");
    const int S = (int)((double)(H > W ? H : W) * 1.4);
    REQUIRE(S >= 1 && S <= 32767, "deskew square side %d out of range", S);
    const int cp = (int)(S / 2.0), top = cp - (int)(H / 2.0), left = cp - (int)(W / 2.0);      // main.py:1615-1619
    alloc_check();
    std::vector<double> minv((size_t)n_angles * 6);
    for (int a = 0; a < n_angles; ++a) {
        double M[6];
        if (matrices) memcpy(M, matrices + (size_t)a * 6, sizeof(M));
        else if (sbbseg_rotation_matrix((double)(S / 2), (double)(S / 2), angles_deg[a], M)) return 1;     // main.py:161  center = (w // 2, h // 2)
        invert_affine(M, &minv[(size_t)a * 6]);
    }
    float tab[128];
    cubic_table(tab);
    const size_t need = (size_t)n_angles * 6 * sizeof(double) + sizeof(tab) + (size_t)n_angles * S * sizeof(int);
    if (ensure(c, (void**)&c->d_deskew, &c->deskew_cap, need)) return 1;
    double* d_minv = (double*)c->d_deskew;
    float* d_tab = (float*)(d_minv + (size_t)n_angles * 6);
    int* d_counts = (int*)(d_tab + 128);
    HIPCHK(hipMemcpyAsync(d_minv, minv.data(), minv.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_tab, tab, sizeof(tab), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));                       // (pageable host staging buffers die with this frame)
    HIPCHK(launch_deskew_profiles((const uint8_t*)d_mask_hw, H, W, S, top, left, d_minv, d_tab, n_angles, d_counts, c->stream));
    HIPCHK(hipMemcpyAsync(counts, d_counts, (size_t)n_angles * S * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
    API_END
}

int sbbseg_deskew_profiles(sbbseg_ctx* c, const uint8_t* mask_hw, int H, int W, const double* matrices, const double* angles_deg, int n_angles,
                           int32_t* counts)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(mask_hw && H > 0 && W > 0, "bad arguments");
    const size_t pix = (size_t)H * W;
    if (ensure(c, (void**)&c->d_morph_b, &c->morph_b_cap, pix)) return 1;
    HIPCHK(hipMemcpyAsync(c->d_morph_b, mask_hw, pix, hipMemcpyHostToDevice, c->stream));
    return sbbseg_deskew_profiles_dev(c, c->d_morph_b, H, W, matrices, angles_deg, n_angles, counts);
    API_END
}

// ----------------------------------------------------------------------------------------- debug
int sbbseg_debug_ingest(sbbseg_ctx* c, const uint8_t* page_hwc, int Hp, int Wp, const int32_t* tile_xy, int n_tiles,
                        int form, float* out, size_t out_floats)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(page_hwc && tile_xy && out && n_tiles >= 1 && n_tiles <= c->max_batch, "bad arguments (n_tiles <= max_batch)");
    REQUIRE(form == SBBSEG_INPUT_C8 || form == SBBSEG_INPUT_PAIRS, "unknown form");
    REQUIRE(c->form_tensor[form] >= 0, "plan does not use input form %d", form);
    const Tensor& t = c->tensors[c->form_tensor[form]];
    const size_t n = t.elems_per_patch * n_tiles;
    REQUIRE(out_floats >= n, "output buffer too small (%zu < %zu)", out_floats, n);
    const size_t pix = (size_t)Hp * Wp;
    if (ensure(c, (void**)&c->d_page, &c->page_cap, pix * 3)) return 1;
    HIPCHK(hipMemcpyAsync(c->d_page, page_hwc, pix * 3, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(c->d_tile_xy, tile_xy, sizeof(int) * 2 * n_tiles, hipMemcpyHostToDevice));
    IngestParams ip;
    if (fill_ingest(c, ip)) return 1;
    ip.page = c->d_page; ip.Hp = Hp; ip.Wp = Wp; ip.src_Hp = Hp; ip.src_Wp = Wp; ip.tile_xy = c->d_tile_xy; ip.n_tiles = n_tiles;
    HIPCHK(launch_ingest_u8(ip, c->precision, c->stream));
    float* d_tmp = nullptr;
    HIPCHK(hipMalloc((void**)&d_tmp, n * sizeof(float)));
    hipError_t e = c->planes == 2 ? launch_split_to_f32(t.data(), d_tmp, n / t.C, t.C, c->stream)
                                  : launch_to_f32(t.data(), d_tmp, n, c->precision, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_tmp, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_tmp);
    HIPCHK(e);
    if (c->planes == 2 && form == SBBSEG_INPUT_C8)      // (the hi plane's slots 4..6 repeat lo(ch 0..2) for the fused tail: not channels)
        for (size_t i = 0; i < n; i += 8)
            for (int ch = 3; ch < 8; ++ch) out[i + ch] = 0.f;
    return 0;
    API_END
}

int sbbseg_debug_read_tensor(sbbseg_ctx* c, int tensor_id, int n, float* out, size_t out_floats)
{
    API_BEGIN
    if (check_ready(c)) return 1;
    REQUIRE(tensor_id >= 0 && tensor_id < (int)c->tensors.size() && out && n >= 1 && n <= c->max_batch, "bad arguments");
    const Tensor& t = c->tensors[tensor_id];
    const size_t cnt = t.elems_per_patch * n;
    REQUIRE(out_floats >= cnt, "output buffer too small (%zu < %zu)", out_floats, cnt);
    float* d_tmp = nullptr;
    HIPCHK(hipMalloc((void**)&d_tmp, cnt * sizeof(float)));
    hipError_t e = c->planes == 2 ? launch_split_to_f32(t.data(), d_tmp, cnt / t.C, t.C, c->stream)
                                  : launch_to_f32(t.data(), d_tmp, cnt, c->precision, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_tmp, cnt * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_tmp);
    HIPCHK(e);
    if (c->planes == 2 && t.is_input_form && t.form == SBBSEG_INPUT_C8)      // (see sbbseg_debug_ingest)
        for (size_t i = 0; i < cnt; i += 8)
            for (int ch = 3; ch < 8; ++ch) out[i + ch] = 0.f;
    return 0;
    API_END
}

int sbbseg_debug_inject_alloc_failure(int nth_check)
{
    API_BEGIN
    REQUIRE(nth_check >= 0, "nth_check must be >= 0 (0 disarms)");
    g_alloc_fail_countdown = nth_check;
    return 0;
    API_END
}

// ------------------------------------------------------------------------------ multi-GPU: the one collective, on RCCL
int sbbseg_comm_unique_id(char* id128)
{
    API_BEGIN
    REQUIRE(id128, "bad arguments");
    if (rccl_load()) return 1;
    RcclUniqueId id;
    RCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, sizeof(id.internal));
    return 0;
    API_END
}

int sbbseg_comm_init(sbbseg_ctx* c, int rank, int world, const char* id128)
{
    API_BEGIN
    REQUIRE(c && id128 && world >= 1 && rank >= 0 && rank < world, "bad arguments (rank %d of %d)", rank, world);
    HIPCHK(hipSetDevice(c->device));
    if (rccl_load()) return 1;
    comm_release(c);
    RcclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    void* comm = nullptr;
    RCCLCHK(g_rccl.CommInitRank(&comm, world, id, rank));
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    return 0;
    API_END
}

int sbbseg_comm_info(sbbseg_ctx* c, int* rank, int* world)
{
    API_BEGIN
    REQUIRE(c && rank && world, "bad arguments");
    *rank = c->comm_rank; *world = c->comm ? c->comm_world : 0;
    return 0;
    API_END
}

int sbbseg_comm_destroy(sbbseg_ctx* c)
{
    API_BEGIN
    REQUIRE(c, "null handle");
    comm_release(c);
    return 0;
    API_END
}

int sbbseg_allgather_labels_dev(sbbseg_ctx* c, const void* d_send, size_t bytes_per_rank, void* d_recv)
{
    API_BEGIN
    REQUIRE(c && d_send && d_recv, "bad arguments");
    REQUIRE(c->comm, "no communicator: call sbbseg_comm_init first");
    HIPCHK(hipSetDevice(c->device));
    if (bytes_per_rank == 0) return 0;
    RCCLCHK(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, /* ncclUint8 */ 1, c->comm, c->stream));
    return 0;
    API_END
}

int sbbseg_debug_largest_contour(const uint8_t* mask_hw, int H, int W, int32_t* box_xywh, int64_t* pixels)
{
    API_BEGIN
    REQUIRE(mask_hw && box_xywh && H > 0 && W > 0 && (size_t)H * W < ((size_t)1 << 31), "bad arguments");
    alloc_check();
    int out[5] = {0, 0, 0, 0, 0};
    const bool any = host_largest_contour(mask_hw, H, W, out);
    if (pixels) *pixels = any ? out[4] : 0;
    box_xywh[0] = any ? out[0] : 0; box_xywh[1] = any ? out[1] : 0;
    box_xywh[2] = any ? out[2] - out[0] + 1 : 0; box_xywh[3] = any ? out[3] - out[1] + 1 : 0;
    return 0;
    API_END
}

int sbbseg_debug_largest_contour_area2(const uint8_t* mask_hw, int H, int W, int64_t* area2)
{
    API_BEGIN
    REQUIRE(mask_hw && area2 && H > 0 && W > 0 && (size_t)H * W < ((size_t)1 << 31), "bad arguments");
    alloc_check();
    int out[5] = {0, 0, 0, 0, 0};
    long long a2 = 0;
    host_largest_contour(mask_hw, H, W, out, &a2);
    *area2 = (int64_t)a2;
    return 0;
    API_END
}

int sbbseg_debug_counter(sbbseg_ctx* c, int which, int64_t* value)
{
    API_BEGIN
    REQUIRE(c && value && (which == 0 || which == 1), "unknown counter %d (0 = exact host contour rankings, 1 = patches run through the plan)", which);
    *value = which == 0 ? (int64_t)c->host_contour_calls : c->forwards;
    return 0;
    API_END
}

int sbbseg_debug_set_conv_variant(sbbseg_ctx* c, int variant)
{
    API_BEGIN
    REQUIRE(c && variant >= 0 && variant <= 0x3ffffff, "variant: bits 0-1 = 0 auto | 1 4-wave/2-stage | 2 8-wave/3-stage; bit 2 = one block per tile (non-persistent); bit 3 = no XCD-grouped tile walk; bit 4 = half-K-step stages; bit 5 = XCD-grouped walk on single-class layers; bit 6 = drain epilogue stores; bit 7 = half-line epilogue stores; bits 8-15 = contiguous-run K limit / 64; bit 16 = 8-phase schedule on the 256x256 tile; bit 17 = plain gather everywhere; bit 18 = fused bottleneck blocks run as their three convs; bit 19 = XCD-contiguous walk for grouped launches; bit 20 = one-group form of the fused block kernel; bit 21 = extract_page ranks contours on the host always; bit 22 = stem and max-pool as two launches; bit 23 = the 224 x 224 decoder conv on the generic kernel; bit 24 = expand + next reduce 1x1 convs as two launches");
    c->conv_variant = variant & 0xff;
    c->ph8 = (variant >> 16) & 1;
    c->plain_gather = (variant >> 17) & 1;
    c->unfuse_blocks = (variant >> 18) & 1;
    c->ranged_walk = (variant >> 19) & 1;
    c->block_pq = !((variant >> 20) & 1);
    c->force_host_contours = (variant >> 21) & 1;
    c->unfuse_stem_pool = (variant >> 22) & 1;
    c->no_dec_halo = (variant >> 23) & 1;
    c->no_expand_reduce = (variant >> 24) & 1;
    c->no_c3er = (variant >> 25) & 1;
    if ((variant >> 8) & 0xff) c->contig_max_k = ((variant >> 8) & 0xff) * 64;
    return 0;
    API_END
}

// ------------------------------------------------------------------------------------- profiling
int sbbseg_profile_enable(sbbseg_ctx* c, int enable)
{
    API_BEGIN
    REQUIRE(c, "null handle");
    if (resolve_pending(c)) return 1;
    c->profiling = enable != 0;
    return 0;
    API_END
}

int sbbseg_profile_reset(sbbseg_ctx* c)
{
    API_BEGIN
    REQUIRE(c, "null handle");
    if (resolve_pending(c)) return 1;
    for (auto& op : c->ops) { op.prof_ms = 0; op.prof_launches = 0; op.prof_patches = 0; op.exec_patches = 0; op.prof_exec_patches = 0; }
    return 0;
    API_END
}

int sbbseg_op_executed(sbbseg_ctx* c, int op, double* exec_patches, double* timed_exec_patches)
{
    API_BEGIN
    REQUIRE(c && op >= 0 && op < (int)c->ops.size(), "op index out of range");
    if (resolve_pending(c)) return 1;
    if (exec_patches) *exec_patches = c->ops[op].exec_patches;
    if (timed_exec_patches) *timed_exec_patches = c->ops[op].prof_exec_patches;
    return 0;
    API_END
}

int sbbseg_profile_get(sbbseg_ctx* c, int op, double* total_ms, int64_t* launches, int64_t* patches)
{
    API_BEGIN
    REQUIRE(c && op >= 0 && op < (int)c->ops.size(), "op index out of range");
    if (resolve_pending(c)) return 1;
    if (total_ms) *total_ms = c->ops[op].prof_ms;
    if (launches) *launches = c->ops[op].prof_launches;
    if (patches) *patches = c->ops[op].prof_patches;
    return 0;
    API_END
}

}  // extern "C"
