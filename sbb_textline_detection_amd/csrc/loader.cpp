// loader.cpp -- one-call model load of libsbbseg: Keras-2.3 model_config JSON + fp32 weights -> fused plan -> device.
//
// Replaces what the reference does in start_new_session_and_model (main.py:216-223: keras.models.load_model on an
// .h5 whose root attribute `model_config` is this JSON) for a C-ABI consumer that has no Python: the JSON reader, the
// layer-graph reader and the planner of sbb_textline_detection_amd/{keras_graph,planner}.py, restated in C++ and
// feeding the same plan-construction entry points (sbbseg_set_input ... sbbseg_finalize).  The Python planner stays
// the mirror the CPU tests interpret (tests/plan_interp.py); tests/test_gpu_parity.py checks that a context loaded
// here computes bit-identical results to a Python-planned one (same algebra in the same precision: fp64 folding of
// BN / bias, fp64 pre-sums of coincident parity taps, one rounding to fp32 at the end).
//
// Container (.sbbw, written by weights.save_sbbw / tools/h5_to_sbbw.py):
//   "SBBW0001" | u64 header_len | header JSON {"model_config": {...}, "tensors": [{"name","shape","offset"}]} | pad to 64 | f32 data
#include <ctype.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sbbseg.h"

namespace sbbseg { int set_error(const char* fmt, ...); }    // api.hip: fills sbbseg_last_error(), returns 1

namespace {

struct PlanError : std::runtime_error { using std::runtime_error::runtime_error; };

[[noreturn]] void fail(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw PlanError(buf);
}

// ------------------------------------------------------------------------------------------------ JSON
struct JVal {
    enum T { Null, Bool, Num, Str, Arr, Obj } t = Null;
    bool b = false;
    double num = 0;
    std::string s;
    std::vector<JVal> a;
    std::vector<std::pair<std::string, JVal>> o;
    const JVal* get(const char* key) const
    {
        for (const auto& kv : o)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const JVal& at(const char* key) const
    {
        const JVal* v = get(key);
        if (!v) fail("JSON: key '%s' missing", key);
        return *v;
    }
    bool is_null() const { return t == Null; }
    long integer() const
    {
        if (t != Num) fail("JSON: number expected");
        if (!(num >= -9.0e15 && num <= 9.0e15)) fail("JSON: integer out of range");      // (also NaN / inf: the cast would be undefined)
        return (long)num;
    }
};

struct JParser {
    const char* p;
    const char* end;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    void expect(char c)
    {
        ws();
        if (p >= end || *p != c) fail("JSON: '%c' expected at offset %ld", c, (long)(end - p));
        ++p;
    }
    std::string str()
    {
        expect('"');
        std::string out;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) break;
                switch (*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {                           // \uXXXX -> UTF-8 (BMP only; enough for layer names)
                        if (end - p < 5) fail("JSON: bad \\u escape");
                        unsigned cp = 0;
                        for (int i = 1; i <= 4; ++i) {
                            const char c = p[i];
                            cp = cp * 16 + (c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : 0);
                        }
                        p += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += *p;
                }
                ++p;
            } else out += *p++;
        }
        if (p >= end) fail("JSON: unterminated string");
        ++p;
        return out;
    }
    int depth = 0;
    struct DepthGuard { int& d; explicit DepthGuard(int& d_) : d(d_) { if (++d > 256) fail("JSON: nesting deeper than 256"); } ~DepthGuard() { --d; } };
    JVal value()
    {
        DepthGuard guard(depth);
        ws();
        if (p >= end) fail("JSON: unexpected end");
        JVal v;
        if (*p == '{') {
            ++p;
            v.t = JVal::Obj;
            ws();
            if (p < end && *p == '}') { ++p; return v; }
            for (;;) {
                ws();
                std::string k = str();
                expect(':');
                v.o.emplace_back(std::move(k), value());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                expect('}');
                return v;
            }
        }
        if (*p == '[') {
            ++p;
            v.t = JVal::Arr;
            ws();
            if (p < end && *p == ']') { ++p; return v; }
            for (;;) {
                v.a.push_back(value());
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                expect(']');
                return v;
            }
        }
        if (*p == '"') { v.t = JVal::Str; v.s = str(); return v; }
        if (end - p >= 4 && !strncmp(p, "true", 4)) { p += 4; v.t = JVal::Bool; v.b = true; return v; }
        if (end - p >= 5 && !strncmp(p, "false", 5)) { p += 5; v.t = JVal::Bool; v.b = false; return v; }
        if (end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; return v; }
        if (end - p >= 3 && !strncmp(p, "NaN", 3)) { p += 3; v.t = JVal::Num; v.num = NAN; return v; }
        // number token: copied to a bounded, NUL-terminated buffer first (the header is not NUL-terminated: strtod on the
        // raw range could run past `end`)
        char tok[64];
        size_t len = 0;
        while (p + len < end && len < sizeof(tok) - 1 && (isdigit((unsigned char)p[len]) || strchr("+-.eEInfity", p[len]))) { tok[len] = p[len]; ++len; }
        tok[len] = 0;
        char* e = nullptr;
        v.num = strtod(tok, &e);
        if (e == tok) fail("JSON: value expected %ld bytes before the end of the header", (long)(end - p));
        p += e - tok;
        v.t = JVal::Num;
        return v;
    }
};

// ------------------------------------------------------------------------------------------------ layer graph
// (the vocabulary of keras_graph.parse_model_config: input|zeropad|conv|convT|bn|act|maxpool|upsample|concat|add|crop_last)
struct GNode {
    std::string name, op;
    std::vector<std::string> inputs;
    int H = 0, W = 0, C = 0;                      // output shape
    int kh = 0, kw = 0, sy = 1, sx = 1, filters = 0;
    bool same = false, use_bias = true, center = true, scale = true;
    std::string activation = "linear", kind;       // conv inline activation; act kind
    int pad[4] = {0, 0, 0, 0};                    // zeropad t, b, l, r
    double eps = 1e-3;
    int ph = 0, pw = 0, fy = 0, fx = 0;
};

void pair_of(const JVal& v, int& a, int& b)
{
    if (v.t == JVal::Num) { a = b = (int)v.integer(); return; }
    if (v.t != JVal::Arr || v.a.size() != 2) fail("pair expected");
    a = (int)v.a[0].integer();
    b = (int)v.a[1].integer();
}

int conv_out(int n, int k, int s, bool same) { return same ? (n + s - 1) / s : (n - k) / s + 1; }

struct Graph {
    std::vector<GNode> nodes;
    std::string input_name, output_name;
    std::map<std::string, int> index;
    const GNode& by(const std::string& n) const
    {
        auto it = index.find(n);
        if (it == index.end()) fail("layer %s undefined", n.c_str());
        return nodes[it->second];
    }
};

Graph read_graph(const JVal& mc)
{
    const JVal* cls = mc.get("class_name");
    if (!cls || (cls->s != "Model" && cls->s != "Functional")) fail("unsupported top-level class (expected a functional Keras Model)");
    const JVal& cfg = mc.at("config");
    Graph g;
    for (const JVal& layer : cfg.at("layers").a) {
        const std::string& lc_cls = layer.at("class_name").s;
        const JVal& lc = layer.at("config");
        GNode n;
        n.name = layer.at("name").s;
        const JVal* inb = layer.get("inbound_nodes");
        if (inb && inb->a.size() > 1) fail("layer %s: shared layers (multiple inbound nodes) unsupported", n.name.c_str());
        if (inb && inb->a.size() == 1)
            for (const JVal& ref : inb->a[0].a) n.inputs.push_back(ref.a.at(0).s);
        for (const auto& i : n.inputs)
            if (!g.index.count(i)) fail("layer %s: input %s not yet defined (config not topologically ordered)", n.name.c_str(), i.c_str());
        const JVal* df = lc.get("data_format");
        if (df && df->t == JVal::Str && df->s != "channels_last") fail("layer %s: only channels_last is supported", n.name.c_str());
        const GNode* in0 = n.inputs.empty() ? nullptr : &g.by(n.inputs[0]);
        auto need_input = [&]() -> const GNode& {
            if (!in0) fail("layer %s: no input", n.name.c_str());
            return *in0;
        };
        if (lc_cls == "InputLayer") {
            const JVal& bis = lc.at("batch_input_shape");
            n.op = "input";
            n.H = (int)bis.a.at(1).integer(); n.W = (int)bis.a.at(2).integer(); n.C = (int)bis.a.at(3).integer();
        } else if (lc_cls == "ZeroPadding2D") {
            const JVal& pd = lc.at("padding");
            int t, b, l, r;
            if (pd.t == JVal::Num) t = b = l = r = (int)pd.integer();
            else {
                const JVal &a = pd.a.at(0), &c = pd.a.at(1);
                if (a.t == JVal::Num) { t = b = (int)a.integer(); l = r = (int)c.integer(); }
                else { t = (int)a.a.at(0).integer(); b = (int)a.a.at(1).integer(); l = (int)c.a.at(0).integer(); r = (int)c.a.at(1).integer(); }
            }
            n.op = "zeropad";
            n.pad[0] = t; n.pad[1] = b; n.pad[2] = l; n.pad[3] = r;
            n.H = need_input().H + t + b; n.W = in0->W + l + r; n.C = in0->C;
        } else if (lc_cls == "Conv2D" || lc_cls == "Conv2DTranspose") {
            pair_of(lc.at("kernel_size"), n.kh, n.kw);
            pair_of(lc.at("strides"), n.sy, n.sx);
            int d0 = 1, d1 = 1;
            if (const JVal* dr = lc.get("dilation_rate")) pair_of(*dr, d0, d1);
            if (d0 != 1 || d1 != 1) fail("layer %s: dilation unsupported", n.name.c_str());
            const std::string& padding = lc.at("padding").s;
            if (padding != "same" && padding != "valid") fail("layer %s: padding '%s' unsupported", n.name.c_str(), padding.c_str());
            n.same = padding == "same";
            n.filters = (int)lc.at("filters").integer();
            if (const JVal* ub = lc.get("use_bias")) n.use_bias = ub->t == JVal::Bool ? ub->b : true;
            if (const JVal* ac = lc.get("activation")) n.activation = ac->t == JVal::Str ? ac->s : "linear";
            need_input();
            if (lc_cls == "Conv2D") {
                n.op = "conv";
                n.H = conv_out(in0->H, n.kh, n.sy, n.same); n.W = conv_out(in0->W, n.kw, n.sx, n.same);
            } else {
                const JVal* op = lc.get("output_padding");
                if (op && !op->is_null()) fail("layer %s: Conv2DTranspose with output_padding unsupported", n.name.c_str());
                n.op = "convT";
                n.H = n.same ? in0->H * n.sy : in0->H * n.sy + std::max(n.kh - n.sy, 0);
                n.W = n.same ? in0->W * n.sx : in0->W * n.sx + std::max(n.kw - n.sx, 0);
            }
            n.C = n.filters;
        } else if (lc_cls == "BatchNormalization") {
            int axis = -1;
            if (const JVal* ax = lc.get("axis")) axis = ax->t == JVal::Arr ? (int)ax->a.at(0).integer() : (int)ax->integer();
            if (axis != 3 && axis != -1) fail("layer %s: BN axis %d unsupported (NHWC only)", n.name.c_str(), axis);
            n.op = "bn";
            if (const JVal* e = lc.get("epsilon")) n.eps = e->num;
            if (const JVal* c = lc.get("center")) n.center = c->t == JVal::Bool ? c->b : true;
            if (const JVal* s = lc.get("scale")) n.scale = s->t == JVal::Bool ? s->b : true;
            n.H = need_input().H; n.W = in0->W; n.C = in0->C;
        } else if (lc_cls == "Activation") {
            n.op = "act";
            n.kind = lc.at("activation").s;
            if (n.kind != "relu" && n.kind != "softmax" && n.kind != "linear") fail("layer %s: activation '%s' unsupported", n.name.c_str(), n.kind.c_str());
            n.H = need_input().H; n.W = in0->W; n.C = in0->C;
        } else if (lc_cls == "MaxPooling2D") {
            pair_of(lc.at("pool_size"), n.ph, n.pw);
            const JVal* st = lc.get("strides");
            if (st && !st->is_null()) pair_of(*st, n.sy, n.sx);
            else { n.sy = n.ph; n.sx = n.pw; }
            const JVal* pd = lc.get("padding");
            if (pd && pd->t == JVal::Str && pd->s != "valid") fail("layer %s: only valid max-pooling supported", n.name.c_str());
            n.op = "maxpool";
            n.H = (need_input().H - n.ph) / n.sy + 1; n.W = (in0->W - n.pw) / n.sx + 1; n.C = in0->C;
        } else if (lc_cls == "UpSampling2D") {
            pair_of(lc.at("size"), n.fy, n.fx);
            const JVal* ip = lc.get("interpolation");
            if (ip && ip->t == JVal::Str && ip->s != "nearest") fail("layer %s: only nearest UpSampling2D supported", n.name.c_str());
            n.op = "upsample";
            n.H = need_input().H * n.fy; n.W = in0->W * n.fx; n.C = in0->C;
        } else if (lc_cls == "Concatenate") {
            const JVal* ax = lc.get("axis");
            if (ax && ax->t == JVal::Num && ax->integer() != 3 && ax->integer() != -1) fail("layer %s: only channel concatenation supported", n.name.c_str());
            n.op = "concat";
            n.H = need_input().H; n.W = in0->W; n.C = 0;
            for (const auto& i : n.inputs) {
                const GNode& q = g.by(i);
                if (q.H != n.H || q.W != n.W) fail("layer %s: concat inputs differ in H,W", n.name.c_str());
                n.C += q.C;
            }
        } else if (lc_cls == "Add") {
            n.op = "add";
            n.H = need_input().H; n.W = in0->W; n.C = in0->C;
            for (const auto& i : n.inputs) {
                const GNode& q = g.by(i);
                if (q.H != n.H || q.W != n.W || q.C != n.C) fail("layer %s: add inputs differ in shape", n.name.c_str());
            }
        } else if (lc_cls == "Lambda") {
            // upstream's only Lambda is one_side_pad's crop x[:, :-1, :-1, :] right after ZeroPadding2D((1,1)); never unmarshalled
            const GNode& prev = need_input();
            if (prev.op != "zeropad" || prev.pad[0] != 1 || prev.pad[1] != 1 || prev.pad[2] != 1 || prev.pad[3] != 1)
                fail("layer %s: Lambda not recognised as one_side_pad crop", n.name.c_str());
            // (the body is opaque marshalled bytecode: what can be checked is -- an anonymous lambda without bound arguments;
            // SBBSEG_STRICT_LAMBDA=1 refuses Lambda layers altogether, as keras_graph.py does)
            const JVal* ft = lc.get("function_type");
            const JVal* args = lc.get("arguments");
            if ((ft && ft->t == JVal::Str && ft->s != "lambda") || (args && ((args->t == JVal::Obj && !args->o.empty()) || (args->t == JVal::Arr && !args->a.empty()))))
                fail("layer %s: Lambda with a named function / bound arguments is not the one_side_pad crop", n.name.c_str());
            if (const char* strict = getenv("SBBSEG_STRICT_LAMBDA"))
                if (strict[0] && strict[0] != '0') fail("layer %s: Lambda layers are refused (SBBSEG_STRICT_LAMBDA); its bytecode cannot be inspected", n.name.c_str());
            n.op = "crop_last";
            n.H = prev.H - 1; n.W = prev.W - 1; n.C = prev.C;
        } else if (lc_cls == "Dropout" || lc_cls == "SpatialDropout2D") {
            n.op = "act";
            n.kind = "linear";
            n.H = need_input().H; n.W = in0->W; n.C = in0->C;
        } else {
            fail("layer %s: unsupported layer class %s", n.name.c_str(), lc_cls.c_str());
        }
        g.index[n.name] = (int)g.nodes.size();
        g.nodes.push_back(std::move(n));
    }
    g.input_name = cfg.at("input_layers").a.at(0).a.at(0).s;
    g.output_name = cfg.at("output_layers").a.at(0).a.at(0).s;
    return g;
}

// ------------------------------------------------------------------------------------------------ weights
struct WTensor { std::vector<long> shape; const float* data = nullptr; size_t n = 0; };
typedef std::map<std::string, WTensor> WeightMap;

const WTensor& weight(const WeightMap& w, const std::string& name)
{
    auto it = w.find(name);
    if (it == w.end()) fail("weight %s missing", name.c_str());
    return it->second;
}

// ------------------------------------------------------------------------------------------------ plan (planner.py)
struct Seg {
    int tensor = -1, channels = 0, shift = 0, off_y = 0, off_x = 0;
    int kh = 0, kw = 0, stride_y = 1, stride_x = 1, pad_top = 0, pad_left = 0;
    std::shared_ptr<std::vector<float>> w;           // [kh][kw][channels][cout]
};
Seg seg(int tensor, int channels, int shift = 0, int off_y = 0, int off_x = 0)
{
    Seg s;
    s.tensor = tensor; s.channels = channels; s.shift = shift; s.off_y = off_y; s.off_x = off_x;
    return s;
}

struct TensorSpec { int H, W, C; std::string kind; int pad; std::string name; };

struct HeadStep { std::string name; int src = -1, cin = 0, classes = 0; std::vector<float> w, scale, shift; };

struct Origin { std::vector<Seg> srcs; int geom[6]; int oh, ow; double macs; std::string name; bool valid = false; };

struct Step {
    std::string kind, name;                        // conv | maxpool | head | tail
    // conv
    std::vector<Seg> srcs;
    int cout = 0, out_h = 0, out_w = 0, out_stride[2] = {1, 1}, out_off[2] = {0, 0};
    std::vector<float> scale, shift, raw_scale, raw_shift;
    int out = -1, residual = -1, raw_out = -1;
    bool relu = false, has_raw = false;
    std::shared_ptr<HeadStep> head;
    double algorithmic_macs = 0;
    Origin origin;
    // maxpool
    int src = -1, dst = -1, k = 0, stride = 0;
    std::vector<float> pre_scale, pre_shift;
    bool pre_relu = false, has_pre = false;
    // tail
    int src0 = -1, img = -1;
    std::shared_ptr<std::vector<float>> w_src0, w_img;
};

struct Pending {
    const GNode* node = nullptr;
    std::vector<Seg> srcs;
    int kh = 0, kw = 0, sy = 1, sx = 1, pt = 0, pl = 0, cout = 0;
    std::shared_ptr<std::vector<float>> w;           // [kh][kw][cin_total][cout]
    int cin_total = 0;
    std::vector<double> scale, shift, raw_scale, raw_shift;
    bool relu = false, raw_needed = false;
    int residual = -1, oh = 0, ow = 0;
    double logical_macs_per_out = 0;
    int emitted_out = -1, emitted_raw = -1;
    std::string stage = "conv";                    // conv -> bn -> (add) -> relu
    bool has_multi = false;
    std::vector<Seg> multi;
    bool is_convT = false;
    int convT[4] = {0, 0, 0, 0}, in_h = 0, in_w = 0;
    std::string name;
};
typedef std::shared_ptr<Pending> PendingP;

struct View {
    int H = 0, W = 0;
    std::vector<Seg> segs;
    int pad[4] = {0, 0, 0, 0};
    PendingP pending, raw_of;
};
typedef std::shared_ptr<View> ViewP;
ViewP view(int H, int W, std::vector<Seg> segs = {}) { auto v = std::make_shared<View>(); v->H = H; v->W = W; v->segs = std::move(segs); return v; }

struct Plan {
    int in_h = 0, in_w = 0, classes = 0;
    std::vector<TensorSpec> tensors;
    std::vector<Step> steps;
};

struct Options { bool parity_split = true, fuse_head = true, fuse_tail = true, merge_shortcut = true; };

std::vector<float> to_f32(const std::vector<double>& v) { std::vector<float> o(v.size()); for (size_t i = 0; i < v.size(); ++i) o[i] = (float)v[i]; return o; }

// taps of a 3x3 window that land on source row a+t (+py-1) of a nearest-x2-upsampled tensor, for output-row parity py
const std::vector<int>& parity_taps(int p, int t)
{
    static const std::vector<int> tab[2][2] = {{{0}, {1, 2}}, {{0, 1}, {2}}};
    return tab[p][t];
}

Plan build_plan(const Graph& graph, const WeightMap& weights, const Options& opt)
{
    std::map<std::string, int> consumers;
    for (const auto& n : graph.nodes)
        for (const auto& i : n.inputs) consumers[i] += 1;
    auto ncons = [&](const std::string& n) { auto it = consumers.find(n); return it == consumers.end() ? 0 : it->second; };
    const GNode& gin = graph.by(graph.input_name);
    const int in_h = gin.H, in_w = gin.W;
    if (gin.C != 3) fail("network input must have 3 channels");
    Plan plan;
    plan.in_h = in_h; plan.in_w = in_w;
    std::map<std::string, ViewP> views;
    std::map<std::pair<int, int>, int> input_forms;

    auto new_tensor = [&](int H, int W, int C, const std::string& name, const std::string& kind = "act", int pad = 0) {
        plan.tensors.push_back({H, W, C, kind, pad, name});
        return (int)plan.tensors.size() - 1;
    };
    auto input_form = [&](int form, int pad) {
        const auto key = std::make_pair(form, pad);
        auto it = input_forms.find(key);
        if (it != input_forms.end()) return it->second;
        int id;
        if (form == SBBSEG_INPUT_C8) id = new_tensor(in_h, in_w, 8, "input_c8", "input_c8", 0);
        else {
            for (const auto& kv : input_forms)
                if (kv.first.first == SBBSEG_INPUT_PAIRS) fail("only one PAIRS input form (one padding) is supported");
            id = new_tensor(in_h + 2 * pad, (in_w + 2 * pad + 1) / 2, 8, "input_pairs", "input_pairs", pad);
        }
        input_forms[key] = id;
        return id;
    };

    auto emit = [&](Pending& p) {
        if (p.emitted_out >= 0 || p.emitted_raw >= 0) return;
        const int oh = p.oh, ow = p.ow;
        p.emitted_out = new_tensor(oh, ow, p.cout, p.name);
        if (p.raw_needed) p.emitted_raw = new_tensor(oh, ow, p.cout, p.node->name + ":raw");
        std::vector<Seg> srcs;
        if (p.has_multi) srcs = p.multi;
        else {
            int cbase = 0;
            for (const Seg& g : p.srcs) {
                Seg s = g;
                s.kh = p.kh; s.kw = p.kw; s.stride_y = p.sy; s.stride_x = p.sx; s.pad_top = p.pt; s.pad_left = p.pl;
                auto w = std::make_shared<std::vector<float>>((size_t)p.kh * p.kw * g.channels * p.cout);
                for (int t = 0; t < p.kh * p.kw; ++t)
                    for (int c = 0; c < g.channels; ++c)
                        memcpy(&(*w)[((size_t)t * g.channels + c) * p.cout], &(*p.w)[((size_t)t * p.cin_total + cbase + c) * p.cout], sizeof(float) * p.cout);
                s.w = w;
                srcs.push_back(std::move(s));
                cbase += g.channels;
            }
        }
        Step common;
        common.kind = "conv";
        common.cout = p.cout;
        common.scale = to_f32(p.scale); common.shift = to_f32(p.shift);
        common.out = p.emitted_out; common.relu = p.relu; common.residual = p.residual; common.raw_out = p.emitted_raw;
        if (p.raw_needed) { common.has_raw = true; common.raw_scale = to_f32(p.raw_scale); common.raw_shift = to_f32(p.raw_shift); }
        if (p.is_convT) {
            // Conv2DTranspose, stride 2 (planner.py emit(): output-parity classes with the sub-kernel of the taps that land on each)
            const int kh = p.convT[0], kw = p.convT[1], pt = p.convT[2], pl = p.convT[3];
            long chsum = 0;
            for (const Seg& g : srcs) chsum += g.channels;
            const double macs_t = (double)p.in_h * p.in_w * kh * kw * chsum * p.cout;
            const size_t first = plan.steps.size();
            auto floordiv2 = [](int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); };
            for (int py = 0; py < 2; ++py) {
                std::set<int> dys;
                for (int ky = 0; ky < kh; ++ky)
                    if (((py + pt - ky) % 2 + 2) % 2 == 0) dys.insert(floordiv2(py + pt - ky));
                for (int px = 0; px < 2; ++px) {
                    std::set<int> dxs;
                    for (int kx = 0; kx < kw; ++kx)
                        if (((px + pl - kx) % 2 + 2) % 2 == 0) dxs.insert(floordiv2(px + pl - kx));
                    const int ch = (oh - py + 1) / 2, cw = (ow - px + 1) / 2;
                    if (ch <= 0 || cw <= 0) continue;
                    if (dys.empty() || dxs.empty()) fail("%s: Conv2DTranspose parity class (%d,%d) receives no taps (kernel smaller than stride)", p.name.c_str(), py, px);
                    const int dy0 = *dys.begin(), dx0 = *dxs.begin();
                    const int kh2 = *dys.rbegin() - dy0 + 1, kw2 = *dxs.rbegin() - dx0 + 1;
                    Step st = common;
                    for (const Seg& g : srcs) {
                        auto w2 = std::make_shared<std::vector<float>>((size_t)kh2 * kw2 * g.channels * p.cout, 0.f);
                        for (int ty = 0; ty < kh2; ++ty) {
                            const int ky = py + pt - 2 * (dy0 + ty);
                            for (int tx = 0; tx < kw2; ++tx) {
                                const int kx = px + pl - 2 * (dx0 + tx);
                                if (ky >= 0 && ky < kh && kx >= 0 && kx < kw)
                                    memcpy(&(*w2)[((size_t)(ty * kw2 + tx) * g.channels) * p.cout], &(*g.w)[((size_t)(ky * kw + kx) * g.channels) * p.cout],
                                           sizeof(float) * g.channels * p.cout);
                            }
                        }
                        Seg s = seg(g.tensor, g.channels, 0, g.off_y, g.off_x);
                        s.kh = kh2; s.kw = kw2; s.stride_y = 1; s.stride_x = 1; s.pad_top = -dy0; s.pad_left = -dx0; s.w = w2;
                        st.srcs.push_back(std::move(s));
                    }
                    char nm[256];
                    snprintf(nm, sizeof(nm), "%s:t%d%d", p.node->name.c_str(), py, px);
                    st.name = nm;
                    st.out_h = ch; st.out_w = cw; st.out_stride[0] = st.out_stride[1] = 2; st.out_off[0] = py; st.out_off[1] = px;
                    plan.steps.push_back(std::move(st));
                }
            }
            const size_t n_cls = plan.steps.size() - first;
            for (size_t k = first; k < plan.steps.size(); ++k) plan.steps[k].algorithmic_macs = macs_t / (double)n_cls;
            return;
        }
        const double macs = (double)((long)oh * ow * p.cout) * p.logical_macs_per_out;
        Origin origin;
        origin.srcs = srcs; origin.oh = oh; origin.ow = ow; origin.macs = macs; origin.name = p.node->name; origin.valid = true;
        const int geom[6] = {p.kh, p.kw, p.sy, p.sx, p.pt, p.pl};
        memcpy(origin.geom, geom, sizeof(geom));
        bool splittable = opt.parity_split && !p.has_multi && srcs[0].shift == 1 && p.kh == 3 && p.kw == 3 && p.sy == 1 && p.sx == 1 &&
                          p.pt == 1 && p.pl == 1 && oh % 2 == 0 && ow % 2 == 0 && p.residual < 0 && !p.raw_needed;
        for (const Seg& g : srcs)
            if (g.shift && (g.off_y || g.off_x)) splittable = false;
        if (!splittable) {
            Step st = common;
            st.name = p.name; st.srcs = srcs; st.out_h = oh; st.out_w = ow; st.algorithmic_macs = macs; st.origin = origin;
            plan.steps.push_back(std::move(st));
            return;
        }
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                Step st = common;
                for (const Seg& g : srcs) {
                    if (g.shift == 1) {
                        // taps that read the same stored pixel are pre-summed in fp64, then rounded once to fp32
                        const size_t plane = (size_t)g.channels * p.cout;
                        std::vector<double> acc(4 * plane, 0.0);
                        for (int ty = 0; ty < 2; ++ty)
                            for (int tx = 0; tx < 2; ++tx)
                                for (int ky : parity_taps(py, ty))
                                    for (int kx : parity_taps(px, tx)) {
                                        const float* src = &(*g.w)[(size_t)(ky * 3 + kx) * plane];
                                        double* dst = &acc[(size_t)(ty * 2 + tx) * plane];
                                        for (size_t i = 0; i < plane; ++i) dst[i] += (double)src[i];
                                    }
                        auto w2 = std::make_shared<std::vector<float>>(4 * plane);
                        for (size_t i = 0; i < acc.size(); ++i) (*w2)[i] = (float)acc[i];
                        Seg s = seg(g.tensor, g.channels, 0, 0, 0);
                        s.kh = 2; s.kw = 2; s.stride_y = 1; s.stride_x = 1; s.pad_top = 1 - py; s.pad_left = 1 - px; s.w = w2;
                        st.srcs.push_back(std::move(s));
                    } else {
                        Seg s = seg(g.tensor, g.channels, 0, g.off_y, g.off_x);
                        s.kh = 3; s.kw = 3; s.stride_y = 2; s.stride_x = 2; s.pad_top = 1 - py; s.pad_left = 1 - px; s.w = g.w;
                        st.srcs.push_back(std::move(s));
                    }
                }
                char nm[256];
                snprintf(nm, sizeof(nm), "%s:p%d%d", p.node->name.c_str(), py, px);
                st.name = nm;
                st.out_h = oh / 2; st.out_w = ow / 2; st.out_stride[0] = st.out_stride[1] = 2; st.out_off[0] = py; st.out_off[1] = px;
                st.algorithmic_macs = macs / 4;
                st.origin = origin;
                plan.steps.push_back(std::move(st));
            }
    };

    auto get_view = [&](const std::string& name) -> ViewP {
        auto it = views.find(name);
        if (it == views.end()) fail("layer %s has no value yet", name.c_str());
        return it->second;
    };
    auto materialize = [&](const std::string& name) -> ViewP {
        ViewP v = get_view(name);
        if (v->pending) {
            emit(*v->pending);
            v->segs = {seg(v->pending->emitted_out, v->pending->cout)};
            v->pending.reset();
        } else if (v->raw_of) {
            Pending& p = *v->raw_of;
            if (p.emitted_out >= 0 && p.emitted_raw < 0) fail("%s: raw conv output requested after the conv was emitted", name.c_str());
            p.raw_needed = true;
            emit(p);
            v->segs = {seg(p.emitted_raw, p.cout)};
            v->raw_of.reset();
        }
        return v;
    };
    auto no_pad = [](const View& v) { return !v.pad[0] && !v.pad[1] && !v.pad[2] && !v.pad[3]; };
    auto plain_tensor = [&](const std::string& name) -> int {
        ViewP v = materialize(name);
        if (v->segs.size() != 1 || v->segs[0].shift || v->segs[0].off_y || v->segs[0].off_x || !no_pad(*v)) fail("%s: needs a plain stored tensor here", name.c_str());
        return v->segs[0].tensor;
    };
    auto gatherable = [&](const std::string& name) -> ViewP {
        ViewP v = materialize(name);
        std::vector<Seg> segs;
        for (const Seg& s : v->segs) segs.push_back(seg(s.tensor, s.channels, s.shift, s.off_y + v->pad[0], s.off_x + v->pad[2]));
        return view(v->H, v->W, std::move(segs));
    };

    for (const GNode& n : graph.nodes) {
        if (n.op == "input") {
            views[n.name] = view(in_h, in_w, {seg(-1, 3)});                 // tensor -1 = the network input
        } else if (n.op == "zeropad") {
            ViewP src = materialize(n.inputs[0]);
            ViewP v = view(src->H + n.pad[0] + n.pad[1], src->W + n.pad[2] + n.pad[3], src->segs);
            for (int k = 0; k < 4; ++k) v->pad[k] = src->pad[k] + n.pad[k];
            views[n.name] = v;
        } else if (n.op == "crop_last") {
            ViewP src = get_view(n.inputs[0]);
            if (src->pad[1] < 1 || src->pad[3] < 1) fail("%s: crop of real data unsupported", n.name.c_str());
            ViewP v = view(src->H - 1, src->W - 1, src->segs);
            v->pad[0] = src->pad[0]; v->pad[1] = src->pad[1] - 1; v->pad[2] = src->pad[2]; v->pad[3] = src->pad[3] - 1;
            views[n.name] = v;
        } else if (n.op == "upsample") {
            if (n.fy != 2 || n.fx != 2) fail("%s: only x2 upsampling supported", n.name.c_str());
            ViewP src = gatherable(n.inputs[0]);
            std::vector<Seg> segs;
            for (const Seg& s : src->segs) {
                if (s.shift || s.off_y || s.off_x) fail("%s: upsampling of an already transformed view", n.name.c_str());
                segs.push_back(seg(s.tensor, s.channels, 1));
            }
            views[n.name] = view(src->H * 2, src->W * 2, std::move(segs));
        } else if (n.op == "concat") {
            std::vector<Seg> segs;
            int H = -1, W = -1;
            for (const auto& i : n.inputs) {
                ViewP p = gatherable(i);
                if (H < 0) { H = p->H; W = p->W; }
                if (p->H != H || p->W != W) fail("%s: concat inputs differ in size", n.name.c_str());
                for (const Seg& s : p->segs) segs.push_back(s);
            }
            views[n.name] = view(H, W, std::move(segs));
        } else if (n.op == "conv") {
            ViewP src = gatherable(n.inputs[0]);
            int kh = n.kh, kw = n.kw, sy = n.sy, sx = n.sx;
            const int oh = n.H, ow = n.W, cout = n.C;
            int pt = 0, pl = 0;
            if (n.same) {
                pt = std::max((oh - 1) * sy + kh - src->H, 0) / 2;
                pl = std::max((ow - 1) * sx + kw - src->W, 0) / 2;
            }
            const WTensor& wk = weight(weights, n.name + "/kernel:0");
            std::vector<Seg> segs = src->segs;
            if (segs.size() > 2) fail("%s: more than two concatenated sources", n.name.c_str());
            int cin_total = 0;
            for (const Seg& s : segs) cin_total += s.channels;
            if (wk.n != (size_t)kh * kw * cin_total * cout) fail("%s: kernel has %zu values, expected %dx%dx%dx%d", n.name.c_str(), wk.n, kh, kw, cin_total, cout);
            auto w = std::make_shared<std::vector<float>>(wk.data, wk.data + wk.n);
            // a placement offset shared by every source is just conv padding
            int my = segs[0].off_y, mx = segs[0].off_x;
            for (const Seg& s : segs) { my = std::min(my, s.off_y); mx = std::min(mx, s.off_x); }
            for (Seg& s : segs) { s.off_y -= my; s.off_x -= mx; }
            pt += my; pl += mx;
            for (const Seg& s : segs)
                if (s.shift && (s.off_y || s.off_x)) fail("%s: upsampled source with a placement offset unsupported", n.name.c_str());
            const double taps = (double)kh * kw * cin_total;
            for (size_t k = 0; k < segs.size(); ++k) {
                Seg& s = segs[k];
                if (s.tensor != -1) continue;                                // a conv that reads the image itself
                if (s.shift) fail("%s: upsampled network input unsupported", n.name.c_str());
                if (sx == 2 && segs.size() == 1) {
                    // stem: fold the horizontal stride into 2-pixel granules of the PAIRS form
                    const int pad = s.off_y + pt;
                    if (pad != s.off_x + pl) fail("%s: asymmetric stem padding unsupported", n.name.c_str());
                    const int t_id = input_form(SBBSEG_INPUT_PAIRS, pad);
                    const int kw2 = (kw + 1) / 2;
                    auto w2 = std::make_shared<std::vector<float>>((size_t)kh * kw2 * 8 * cout, 0.f);
                    for (int ky = 0; ky < kh; ++ky)
                        for (int dx = 0; dx < kw; ++dx)
                            for (int c = 0; c < 3; ++c)
                                memcpy(&(*w2)[(((size_t)ky * kw2 + dx / 2) * 8 + (dx & 1) * 4 + c) * cout], &(*w)[(((size_t)ky * kw + dx) * 3 + c) * cout],
                                       sizeof(float) * cout);
                    s = seg(t_id, 8, 0, 0, 0);
                    w = w2; kw = kw2; sx = 1; pt = 0; pl = 0;
                    cin_total = 8;
                } else if (sx == 1) {
                    s = seg(input_form(SBBSEG_INPUT_C8, 0), 3, 0, s.off_y, s.off_x);
                } else fail("%s: unsupported conv on the network input", n.name.c_str());
            }
            auto p = std::make_shared<Pending>();
            p->node = &n; p->srcs = segs; p->kh = kh; p->kw = kw; p->sy = sy; p->sx = sx; p->pt = pt; p->pl = pl; p->cout = cout;
            p->w = w; p->cin_total = cin_total;
            p->scale.assign(cout, 1.0);
            p->shift.assign(cout, 0.0);
            if (n.use_bias) {
                const WTensor& b = weight(weights, n.name + "/bias:0");
                if (b.n != (size_t)cout) fail("%s: bias size", n.name.c_str());
                for (int c = 0; c < cout; ++c) p->shift[c] = (double)b.data[c];
            }
            p->raw_scale = p->scale; p->raw_shift = p->shift;
            p->oh = oh; p->ow = ow; p->logical_macs_per_out = taps; p->name = n.name;
            if (n.activation == "relu") { p->relu = true; p->stage = "relu"; }
            else if (n.activation != "linear") fail("%s: inline activation %s unsupported", n.name.c_str(), n.activation.c_str());
            ViewP v = view(oh, ow);
            v->pending = p;
            views[n.name] = v;
        } else if (n.op == "convT") {
            ViewP src = gatherable(n.inputs[0]);
            const int kh = n.kh, kw = n.kw, oh = n.H, ow = n.W, cout = n.C;
            if (n.sy != 2 || n.sx != 2 || kh < 2 || kw < 2) fail("%s: only stride-2 Conv2DTranspose with kernel >= 2 is lowered", n.name.c_str());
            if (src->segs.size() > 2) fail("%s: Conv2DTranspose source must be one or two stored tensors", n.name.c_str());
            int cin_total = 0;
            for (const Seg& s : src->segs) {
                if (s.shift || s.tensor < 0) fail("%s: Conv2DTranspose source must be one or two stored tensors", n.name.c_str());
                cin_total += s.channels;
            }
            if (n.activation != "linear" && n.activation != "relu") fail("%s: inline activation %s unsupported", n.name.c_str(), n.activation.c_str());
            const int pt = n.same ? std::max(kh - n.sy, 0) / 2 : 0, pl = n.same ? std::max(kw - n.sx, 0) / 2 : 0;
            const WTensor& wk = weight(weights, n.name + "/kernel:0");             // Keras layout [kh][kw][out][in]
            if (wk.n != (size_t)kh * kw * cout * cin_total) fail("%s: transposed kernel size", n.name.c_str());
            auto wt = std::make_shared<std::vector<float>>(wk.n);
            for (int t = 0; t < kh * kw; ++t)
                for (int o = 0; o < cout; ++o)
                    for (int c = 0; c < cin_total; ++c) (*wt)[((size_t)t * cin_total + c) * cout + o] = wk.data[((size_t)t * cout + o) * cin_total + c];
            auto p = std::make_shared<Pending>();
            p->node = &n; p->srcs = src->segs; p->kh = kh; p->kw = kw; p->cout = cout; p->w = wt; p->cin_total = cin_total;
            p->scale.assign(cout, 1.0);
            p->shift.assign(cout, 0.0);
            if (n.use_bias) {
                const WTensor& b = weight(weights, n.name + "/bias:0");
                if (b.n != (size_t)cout) fail("%s: bias size", n.name.c_str());
                for (int c = 0; c < cout; ++c) p->shift[c] = (double)b.data[c];
            }
            p->raw_scale = p->scale; p->raw_shift = p->shift;
            p->oh = oh; p->ow = ow; p->name = n.name;
            p->is_convT = true;
            p->convT[0] = kh; p->convT[1] = kw; p->convT[2] = pt; p->convT[3] = pl;
            p->in_h = src->H; p->in_w = src->W;
            if (n.activation == "relu") { p->relu = true; p->stage = "relu"; }
            ViewP v = view(oh, ow);
            v->pending = p;
            views[n.name] = v;
        } else if (n.op == "bn") {
            ViewP src = get_view(n.inputs[0]);
            PendingP p = src->pending;
            if (!p || p->stage != "conv") fail("%s: BatchNormalization must follow a Conv2D directly", n.name.c_str());
            if (ncons(n.inputs[0]) > 1) {
                // somebody else wants the un-normalised conv output (the f1 skip): keep both
                p->raw_needed = true;
                ViewP rv = view(src->H, src->W);
                rv->raw_of = p;
                views[n.inputs[0]] = rv;
            }
            const WTensor& mu = weight(weights, n.name + "/moving_mean:0");
            const WTensor& var = weight(weights, n.name + "/moving_variance:0");
            const WTensor* g = n.scale ? &weight(weights, n.name + "/gamma:0") : nullptr;
            const WTensor* be = n.center ? &weight(weights, n.name + "/beta:0") : nullptr;
            if (mu.n != (size_t)p->cout || var.n != (size_t)p->cout) fail("%s: BN statistics size", n.name.c_str());
            if ((g && g->n != (size_t)p->cout) || (be && be->n != (size_t)p->cout)) fail("%s: BN gamma / beta size", n.name.c_str());
            for (int c = 0; c < p->cout; ++c) {
                const double a = (g ? (double)g->data[c] : 1.0) / sqrt((double)var.data[c] + n.eps);
                p->scale[c] = p->scale[c] * a;
                p->shift[c] = (p->shift[c] - (double)mu.data[c]) * a + (be ? (double)be->data[c] : 0.0);
            }
            p->stage = "bn";
            ViewP v = view(src->H, src->W);
            v->pending = p;
            views[n.name] = v;
        } else if (n.op == "add") {
            if (n.inputs.size() != 2) fail("%s: Add of %zu inputs unsupported", n.name.c_str(), n.inputs.size());
            ViewP va = get_view(n.inputs[0]), vb = get_view(n.inputs[1]);
            std::vector<int> cand;
            for (int k = 0; k < 2; ++k) {
                const ViewP& v = k ? vb : va;
                if (v->pending && (v->pending->stage == "conv" || v->pending->stage == "bn") && !v->pending->relu && v->pending->residual < 0 &&
                    ncons(n.inputs[k]) == 1)
                    cand.push_back(k);
            }
            if (cand.empty()) fail("%s: Add needs one input that is a conv(+BN) with a single consumer", n.name.c_str());
            PendingP pa = va->pending, pb = vb->pending;
            if (opt.merge_shortcut && cand.size() == 2 && !pa->has_multi && !pb->has_multi && !pa->is_convT && !pb->is_convT && pa->srcs.size() == 1 &&
                pb->srcs.size() == 1 && !pa->raw_needed && !pb->raw_needed && pa->cout == pb->cout && pa->oh == pb->oh && pa->ow == pb->ow &&
                !pa->srcs[0].shift && !pb->srcs[0].shift && pa->srcs[0].tensor >= 0 && pb->srcs[0].tensor >= 0) {
                // projection-shortcut block: BN_a(conv_a(x)) + BN_b(conv_b(y)) is ONE conv over two sources with the BN scales
                // folded into the weight rows (fp64 product, one rounding to fp32)
                std::vector<Seg> segs;
                for (const PendingP& q : {pa, pb}) {
                    const Seg& g = q->srcs[0];
                    auto wq = std::make_shared<std::vector<float>>(q->w->size());
                    for (size_t i = 0; i < wq->size(); ++i) (*wq)[i] = (float)((double)(*q->w)[i] * q->scale[i % q->cout]);
                    Seg s = seg(g.tensor, g.channels, 0, g.off_y, g.off_x);
                    s.kh = q->kh; s.kw = q->kw; s.stride_y = q->sy; s.stride_x = q->sx; s.pad_top = q->pt; s.pad_left = q->pl; s.w = wq;
                    segs.push_back(std::move(s));
                }
                pa->has_multi = true;
                pa->multi = segs;
                pa->srcs = {seg(segs[0].tensor, segs[0].channels, 0, segs[0].off_y, segs[0].off_x), seg(segs[1].tensor, segs[1].channels, 0, segs[1].off_y, segs[1].off_x)};
                for (int c = 0; c < pa->cout; ++c) { pa->shift[c] = pa->shift[c] + pb->shift[c]; pa->scale[c] = 1.0; }
                pa->logical_macs_per_out += pb->logical_macs_per_out;
                pa->name = pa->node->name + "+" + pb->node->name;
                pa->stage = "add";
                ViewP v = view(va->H, va->W);
                v->pending = pa;
                views[n.name] = v;
                continue;
            }
            const int k = cand[0];
            const int other = plain_tensor(n.inputs[1 - k]);
            PendingP p = get_view(n.inputs[k])->pending;
            const TensorSpec& ot = plan.tensors[other];
            if (ot.H != p->oh || ot.W != p->ow || ot.C != p->cout) fail("%s: residual shape mismatch", n.name.c_str());
            p->residual = other;
            p->stage = "add";
            ViewP v = view(va->H, va->W);
            v->pending = p;
            views[n.name] = v;
        } else if (n.op == "act") {
            ViewP src = get_view(n.inputs[0]);
            if (n.kind == "linear") {
                views[n.name] = src;
            } else if (n.kind == "relu") {
                PendingP p = src->pending;
                if (!p || p->relu || ncons(n.inputs[0]) != 1) fail("%s: relu must follow conv/BN/Add with a single consumer", n.name.c_str());
                p->relu = true;
                p->stage = "relu";
                ViewP v = view(src->H, src->W);
                v->pending = p;
                views[n.name] = v;
            } else {                                                              // softmax
                PendingP p = src->pending;
                if (n.name != graph.output_name || !p || p->relu || p->residual >= 0 || p->kh != 1 || p->kw != 1 || p->srcs.size() != 1 ||
                    p->srcs[0].shift || p->srcs[0].off_y || p->srcs[0].off_x || p->sy != 1 || p->sx != 1 || p->is_convT)
                    fail("%s: softmax is only supported as the final 1x1-conv head", n.name.c_str());
                const int t_id = p->srcs[0].tensor;
                if (t_id < 0) fail("%s: head on the network input", n.name.c_str());
                TensorSpec& ts = plan.tensors[t_id];
                if (ts.C != p->srcs[0].channels || ts.C > 64 || p->cout > 8 || ts.H != in_h || ts.W != in_w)
                    fail("%s: head needs <=64 input channels, <=8 classes, input resolution", n.name.c_str());
                auto head = std::make_shared<HeadStep>();
                head->name = p->node->name; head->src = t_id; head->cin = ts.C; head->classes = p->cout;
                head->w.assign(p->w->begin(), p->w->end());                       // [1][1][cin][classes]
                head->scale = to_f32(p->scale); head->shift = to_f32(p->shift);
                std::vector<size_t> producers;
                for (size_t k = 0; k < plan.steps.size(); ++k)
                    if (plan.steps[k].kind == "conv" && plan.steps[k].out == t_id) producers.push_back(k);
                const std::string& src_layer = p->node->inputs[0];
                bool fusable = opt.fuse_head && !producers.empty() && ts.C == 32 && p->cout <= 4 && ncons(src_layer) == 1;
                for (size_t k : producers)
                    if (plan.steps[k].raw_out >= 0 || plan.steps[k].residual >= 0) fusable = false;
                const Origin* org = producers.empty() ? nullptr : &plan.steps[producers[0]].origin;
                static const int g3311[6] = {3, 3, 1, 1, 1, 1};
                if (fusable && opt.fuse_tail && org && org->valid && !memcmp(org->geom, g3311, sizeof(g3311)) && org->srcs.size() == 2 &&
                    org->srcs[0].shift == 1 && org->srcs[0].channels == 64 && plan.tensors[org->srcs[0].tensor].C == 64 &&
                    plan.tensors[org->srcs[1].tensor].kind == "input_c8" && org->srcs[1].channels == 3 && !org->srcs[1].off_y && !org->srcs[1].off_x &&
                    plan.steps[producers[0]].relu && org->oh == in_h && org->ow == in_w && in_h % 16 == 0 && in_w % 16 == 0) {
                    // dedicated kernel for the network tail: LDS halo tiles, weights in registers
                    Step tail;
                    tail.kind = "tail";
                    tail.name = org->name + "+" + p->node->name;
                    tail.src0 = org->srcs[0].tensor; tail.img = org->srcs[1].tensor;
                    tail.w_src0 = org->srcs[0].w; tail.w_img = org->srcs[1].w;
                    tail.scale = plan.steps[producers[0]].scale; tail.shift = plan.steps[producers[0]].shift;
                    tail.head = head; tail.out_h = in_h; tail.out_w = in_w; tail.algorithmic_macs = org->macs;
                    std::vector<Step> kept;
                    for (size_t k = 0; k < plan.steps.size(); ++k)
                        if (std::find(producers.begin(), producers.end(), k) == producers.end()) kept.push_back(std::move(plan.steps[k]));
                    plan.steps = std::move(kept);
                    plan.steps.push_back(std::move(tail));
                    ts.kind = "unused";
                } else if (fusable) {
                    // the head rides in the epilogue of the conv(s) producing its input; that tensor is never written
                    for (size_t k : producers) { plan.steps[k].head = head; plan.steps[k].out = -1; }
                    ts.kind = "unused";
                } else {
                    Step hs;
                    hs.kind = "head";
                    hs.name = head->name;
                    hs.head = head;
                    plan.steps.push_back(std::move(hs));
                }
                plan.classes = p->cout;
                views[n.name] = view(src->H, src->W);
            }
        } else if (n.op == "maxpool") {
            if (n.ph != n.pw || n.sy != n.sx) fail("%s: non-square pooling unsupported", n.name.c_str());
            ViewP pv = get_view(n.inputs[0]);
            PendingP pp = pv->pending;
            Step ps;
            ps.kind = "maxpool";
            ps.name = n.name;
            int src_t;
            bool all_nonzero = true;
            if (pp)
                for (double v : pp->raw_scale) all_nonzero = all_nonzero && v != 0.0;
            if (pp && pp->raw_needed && pp->residual < 0 && pp->stage == "relu" && ncons(n.inputs[0]) == 1 && pp->emitted_out < 0 && all_nonzero) {
                // the conv's un-normalised output is stored anyway (a skip connection wants it): store ONLY that, and let
                // the pool apply BN + ReLU on the fly (saves one tensor write + read)
                ps.has_pre = true;
                ps.pre_scale.resize(pp->cout);
                ps.pre_shift.resize(pp->cout);
                for (int c = 0; c < pp->cout; ++c) {
                    const double a = pp->scale[c] / pp->raw_scale[c];
                    ps.pre_scale[c] = (float)a;
                    ps.pre_shift[c] = (float)(pp->shift[c] - pp->raw_shift[c] * a);
                }
                pp->scale = pp->raw_scale; pp->shift = pp->raw_shift; pp->relu = false;
                pp->raw_needed = false;
                emit(*pp);                                                        // single output == the raw tensor
                pp->emitted_raw = pp->emitted_out;
                src_t = pp->emitted_out;
                pv->pending.reset();
                plan.tensors[src_t].name += ":raw";
                ps.pre_relu = true;
            } else {
                src_t = plain_tensor(n.inputs[0]);
            }
            const int dst = new_tensor(n.H, n.W, n.C, n.name);
            ps.src = src_t; ps.dst = dst; ps.k = n.ph; ps.stride = n.sy;
            plan.steps.push_back(std::move(ps));
            views[n.name] = view(n.H, n.W, {seg(dst, n.C)});
        } else {
            fail("%s: op %s not lowered", n.name.c_str(), n.op.c_str());
        }
    }
    if (plan.classes == 0) fail("graph does not end in Conv2D 1x1 -> BatchNormalization -> softmax");
    for (const Step& s : plan.steps)
        if (s.kind == "conv")
            for (const Seg& g : s.srcs)
                if (g.tensor < 0) fail("%s: unresolved network-input source", s.name.c_str());
    return plan;
}

// ------------------------------------------------------------------------------------------------ plan -> C ABI (_capi.load_plan)
int load_plan(sbbseg_ctx* h, const Plan& plan, int max_batch)
{
    if (sbbseg_set_input(h, plan.in_h, plan.in_w, 3)) return 1;
    std::vector<int> ids;
    for (const TensorSpec& t : plan.tensors) {
        int tid = -1;
        if (t.kind == "input_c8") { if (sbbseg_input_form(h, SBBSEG_INPUT_C8, 0, &tid)) return 1; }
        else if (t.kind == "input_pairs") { if (sbbseg_input_form(h, SBBSEG_INPUT_PAIRS, t.pad, &tid)) return 1; }
        else if (t.kind == "unused") {}
        else if (sbbseg_add_tensor(h, t.H, t.W, t.C, &tid)) return 1;
        ids.push_back(tid);
    }
    for (const Step& s : plan.steps) {
        if (s.kind == "conv") {
            sbbseg_conv_desc d;
            memset(&d, 0, sizeof(d));
            d.n_src = (int)s.srcs.size();
            for (int k = 0; k < d.n_src; ++k) {
                const Seg& g = s.srcs[k];
                sbbseg_conv_src& cs = d.src[k];
                cs.tensor = ids[g.tensor]; cs.channels = g.channels; cs.kh = g.kh; cs.kw = g.kw; cs.stride_y = g.stride_y; cs.stride_x = g.stride_x;
                cs.pad_top = g.pad_top; cs.pad_left = g.pad_left; cs.up_shift = g.shift; cs.off_y = g.off_y; cs.off_x = g.off_x;
            }
            d.cout = s.cout; d.out_h = s.out_h; d.out_w = s.out_w;
            d.out_stride_y = s.out_stride[0]; d.out_stride_x = s.out_stride[1]; d.out_off_y = s.out_off[0]; d.out_off_x = s.out_off[1];
            d.out_tensor = s.out >= 0 ? ids[s.out] : -1;
            d.relu = s.relu ? 1 : 0;
            d.residual_tensor = s.residual >= 0 ? ids[s.residual] : -1;
            d.raw_out_tensor = s.raw_out >= 0 ? ids[s.raw_out] : -1;
            d.head_classes = s.head ? s.head->classes : 0;
            d.algorithmic_macs = s.algorithmic_macs;
            if (sbbseg_add_conv(h, &d, s.srcs[0].w->data(), d.n_src > 1 ? s.srcs[1].w->data() : nullptr, s.scale.data(), s.shift.data(),
                                s.has_raw ? s.raw_scale.data() : nullptr, s.has_raw ? s.raw_shift.data() : nullptr,
                                s.head ? s.head->w.data() : nullptr, s.head ? s.head->scale.data() : nullptr, s.head ? s.head->shift.data() : nullptr))
                return 1;
        } else if (s.kind == "tail") {
            if (sbbseg_add_tail(h, ids[s.src0], ids[s.img], s.w_src0->data(), s.w_img->data(), s.scale.data(), s.shift.data(), s.head->classes,
                                s.head->w.data(), s.head->scale.data(), s.head->shift.data(), s.algorithmic_macs))
                return 1;
        } else if (s.kind == "maxpool") {
            if (sbbseg_add_maxpool(h, ids[s.src], ids[s.dst], s.k, s.stride, s.has_pre ? s.pre_scale.data() : nullptr,
                                   s.has_pre ? s.pre_shift.data() : nullptr, s.pre_relu ? 1 : 0))
                return 1;
        } else if (s.kind == "head") {
            if (sbbseg_add_head(h, ids[s.head->src], s.head->cin, s.head->classes, s.head->w.data(), s.head->scale.data(), s.head->shift.data())) return 1;
        }
    }
    return sbbseg_finalize(h, max_batch);
}

Options options_for(int precision, int flags);

int load_from(const JVal& model_config, const WeightMap& weights, int device, int precision, int max_batch, int flags, sbbseg_ctx** out)
{
    const Graph graph = read_graph(model_config);
    const Plan plan = build_plan(graph, weights, options_for(precision, flags));
    sbbseg_ctx* h = nullptr;
    if (sbbseg_create(device, precision, &h)) return 1;
    if (load_plan(h, plan, max_batch)) {
        const std::string msg = sbbseg_last_error();                 // (destroy succeeds and would not clear it, but keep it safe)
        sbbseg_destroy(h);
        return sbbseg::set_error("%s", msg.c_str());
    }
    *out = h;
    return 0;
}

// ---- plan summary (test hook): one line per tensor / step with CRC32s of every weight plane, so that the CPU tests can
// compare this planner with the Python one value for value without a GPU
uint32_t crc32_bytes(const void* data, size_t n)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    const uint8_t* b = (const uint8_t*)data;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ b[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
uint32_t crc_f(const std::vector<float>& v) { return crc32_bytes(v.data(), v.size() * sizeof(float)); }

std::string plan_summary(const Plan& plan)
{
    std::string out;
    char buf[1024];
    snprintf(buf, sizeof(buf), "plan %d %d %d\n", plan.in_h, plan.in_w, plan.classes);
    out += buf;
    for (const TensorSpec& t : plan.tensors) {
        snprintf(buf, sizeof(buf), "tensor %d %d %d %s %d %s\n", t.H, t.W, t.C, t.kind.c_str(), t.pad, t.name.c_str());
        out += buf;
    }
    for (const Step& s : plan.steps) {
        if (s.kind == "conv") {
            snprintf(buf, sizeof(buf), "conv %s cout=%d out=%dx%d stride=%d,%d off=%d,%d t=%d relu=%d res=%d raw=%d head=%d macs=%.0f scale=%08x shift=%08x", s.name.c_str(),
                     s.cout, s.out_h, s.out_w, s.out_stride[0], s.out_stride[1], s.out_off[0], s.out_off[1], s.out, (int)s.relu, s.residual, s.raw_out,
                     s.head ? s.head->classes : 0, s.algorithmic_macs, crc_f(s.scale), crc_f(s.shift));
            out += buf;
            if (s.has_raw) { snprintf(buf, sizeof(buf), " rscale=%08x rshift=%08x", crc_f(s.raw_scale), crc_f(s.raw_shift)); out += buf; }
            if (s.head) { snprintf(buf, sizeof(buf), " hw=%08x hs=%08x hb=%08x", crc_f(s.head->w), crc_f(s.head->scale), crc_f(s.head->shift)); out += buf; }
            for (const Seg& g : s.srcs) {
                snprintf(buf, sizeof(buf), " | src t=%d ch=%d k=%dx%d s=%d,%d pad=%d,%d up=%d off=%d,%d w=%08x", g.tensor, g.channels, g.kh, g.kw, g.stride_y,
                         g.stride_x, g.pad_top, g.pad_left, g.shift, g.off_y, g.off_x, crc_f(*g.w));
                out += buf;
            }
            out += "\n";
        } else if (s.kind == "maxpool") {
            snprintf(buf, sizeof(buf), "maxpool %s src=%d dst=%d k=%d stride=%d pre=%d", s.name.c_str(), s.src, s.dst, s.k, s.stride, (int)s.pre_relu);
            out += buf;
            if (s.has_pre) { snprintf(buf, sizeof(buf), " ps=%08x pb=%08x", crc_f(s.pre_scale), crc_f(s.pre_shift)); out += buf; }
            out += "\n";
        } else if (s.kind == "tail") {
            snprintf(buf, sizeof(buf), "tail %s src0=%d img=%d out=%dx%d macs=%.0f w0=%08x wi=%08x scale=%08x shift=%08x head=%d hw=%08x hs=%08x hb=%08x\n", s.name.c_str(),
                     s.src0, s.img, s.out_h, s.out_w, s.algorithmic_macs, crc_f(*s.w_src0), crc_f(*s.w_img), crc_f(s.scale), crc_f(s.shift), s.head->classes,
                     crc_f(s.head->w), crc_f(s.head->scale), crc_f(s.head->shift));
            out += buf;
        } else {
            snprintf(buf, sizeof(buf), "head %s src=%d cin=%d classes=%d hw=%08x hs=%08x hb=%08x\n", s.name.c_str(), s.head->src, s.head->cin, s.head->classes,
                     crc_f(s.head->w), crc_f(s.head->scale), crc_f(s.head->shift));
            out += buf;
        }
    }
    return out;
}

struct Container { JVal header; WeightMap weights; };

// parses the .sbbw container; throws PlanError
Container read_container(const void* sbbw, size_t n_bytes)
{
    const char* b = (const char*)sbbw;
    if (!sbbw || n_bytes < 16 || memcmp(b, "SBBW0001", 8)) fail("not an SBBW0001 container");
    uint64_t hlen = 0;
    memcpy(&hlen, b + 8, 8);
    if (hlen > n_bytes - 16) fail("truncated header");
    JParser jp{b + 16, b + 16 + hlen};
    Container c;
    c.header = jp.value();
    const size_t data_off = 16 + hlen + ((64 - (16 + hlen) % 64) % 64);
    if (data_off > n_bytes || ((uintptr_t)(b + data_off) % 4)) fail("truncated or misaligned container");
    const float* data = (const float*)(b + data_off);
    const size_t n_floats = (n_bytes - data_off) / 4;
    for (const JVal& t : c.header.at("tensors").a) {
        WTensor w;
        size_t n = 1;
        for (const JVal& d : t.at("shape").a) {
            const long long dim = d.integer();
            if (dim <= 0 || (unsigned long long)dim > n_floats || n > n_floats / (size_t)dim) fail("tensor %s: bad shape", t.at("name").s.c_str());
            w.shape.push_back(dim);
            n *= (size_t)dim;
        }
        const long long off_ll = t.at("offset").integer();
        if (off_ll < 0 || (unsigned long long)off_ll > n_floats || n > n_floats - (size_t)off_ll) fail("tensor %s leaves the container", t.at("name").s.c_str());
        const size_t off = (size_t)off_ll;
        w.data = data + off;
        w.n = n;
        c.weights[t.at("name").s] = w;
    }
    return c;
}

Options options_for(int precision, int flags)
{
    Options opt;
    opt.parity_split = !(flags & SBBSEG_LOAD_NO_PARITY_SPLIT);
    opt.merge_shortcut = !(flags & SBBSEG_LOAD_NO_SHORTCUT_MERGE);
    opt.fuse_head = precision != SBBSEG_PREC_F32 && !(flags & SBBSEG_LOAD_NO_FUSED_HEAD);
    opt.fuse_tail = (precision == SBBSEG_PREC_F16 || precision == SBBSEG_PREC_BF16 || precision == SBBSEG_PREC_F16X3) && !(flags & SBBSEG_LOAD_NO_FUSED_TAIL);
    return opt;
}

}  // namespace

extern "C" {

int sbbseg_debug_plan_summary(const void* sbbw, size_t n_bytes, int precision, int flags, char* out, size_t capacity, size_t* needed)
{
    try {
        const Container c = read_container(sbbw, n_bytes);
        const std::string txt = plan_summary(build_plan(read_graph(c.header.at("model_config")), c.weights, options_for(precision, flags)));
        if (needed) *needed = txt.size() + 1;
        if (out && capacity) {
            const size_t n = txt.size() + 1 <= capacity ? txt.size() : capacity - 1;
            memcpy(out, txt.data(), n);
            out[n] = 0;
        }
        return 0;
    } catch (const std::bad_alloc&) {
        return sbbseg::set_error("out of host memory (std::bad_alloc)");
    } catch (const std::exception& e) {
        return sbbseg::set_error("plan summary: %s", e.what());
    } catch (...) {
        return sbbseg::set_error("unknown internal error");
    }
}

int sbbseg_model_load(const void* sbbw, size_t n_bytes, int device, int precision, int max_batch, int flags, sbbseg_ctx** out)
{
    try {
        if (!sbbw || !out || max_batch < 1) return sbbseg::set_error("sbbseg_model_load: bad arguments");
        const Container c = read_container(sbbw, n_bytes);
        const JVal& header = c.header;
        const WeightMap& weights = c.weights;
        return load_from(header.at("model_config"), weights, device, precision, max_batch, flags, out);
    } catch (const std::bad_alloc&) {
        return sbbseg::set_error("out of host memory (std::bad_alloc)");
    } catch (const std::exception& e) {
        return sbbseg::set_error("model load: %s", e.what());
    } catch (...) {
        return sbbseg::set_error("unknown internal error");
    }
}

int sbbseg_model_load_file(const char* path, int device, int precision, int max_batch, int flags, sbbseg_ctx** out)
{
    try {
        if (!path || !out) return sbbseg::set_error("sbbseg_model_load_file: bad arguments");
        struct FileGuard { FILE* f; ~FileGuard() { if (f) fclose(f); } } fg{fopen(path, "rb")};      // closed on every path, incl. a throwing resize
        FILE* f = fg.f;
        if (!f) return sbbseg::set_error("cannot open %s", path);
        std::vector<char> buf;
        if (fseek(f, 0, SEEK_END) == 0) {
            const long n = ftell(f);
            rewind(f);
            if (n > 0) {
                buf.resize((size_t)n + 8);                             // (keeps the float data 4-byte aligned: vector storage is)
                if (fread(buf.data(), 1, (size_t)n, f) != (size_t)n) return sbbseg::set_error("short read on %s", path);
                buf.resize((size_t)n);
            }
        }
        if (buf.empty()) return sbbseg::set_error("%s is empty", path);
        return sbbseg_model_load(buf.data(), buf.size(), device, precision, max_batch, flags, out);
    } catch (const std::bad_alloc&) {
        return sbbseg::set_error("out of host memory (std::bad_alloc)");
    } catch (const std::exception& e) {
        return sbbseg::set_error("model load: %s", e.what());
    } catch (...) {
        return sbbseg::set_error("unknown internal error");
    }
}

}  // extern "C"
