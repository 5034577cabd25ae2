// Generic 2-D layer-graph engine: the model families of SURVEY 8(f) #4 behind the reference's factory --
// ResNet2D8 / ResNet2D8Pool (audio_nets/tc_resnet.py:14-15,23-24,73-99), Res8 / Res15 (/Narrow) (audio_nets/res.py:6-123)
// and the KWSModel architectures (audio_nets/kws.py:15-63).  The topology is described by the host (Python, as in the
// reference) through tcr_g2d_conv / batch_norm / pool / add / dropout; this file lays out the variable arenas under the
// caller's TF names, carves the workspace and sequences the kernels of net2d_kernels.hip + bn.hip for eval forward, train forward
// and backward.  Host-only state; every device buffer belongs to the caller.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.h"
#include "net2d.h"

namespace tcr {
enum { G2D_CONV = 0, G2D_BN = 1, G2D_POOL = 2, G2D_ADD = 3, G2D_DROPOUT = 4, G2D_TFILT = 5, G2D_GSUM = 6 };

struct G2dNode {
    int kind = 0, in0 = -1, in1 = -1;       // inputs: node ids, -1 = the network input
    int c = 0, h = 0, w = 0;                // output shape
    // conv
    int cin = 0, kh = 1, kw = 1, sh = 1, sw = 1, dh = 1, dw = 1, pt = 0, pl = 0;
    bool bias = false, relu = false;
    std::string w_name, b_name, bn_prefix;
    int64_t w_off = -1, b_off = -1;
    // batch norm
    bool center = true, scale = true;
    float decay = 0.997f, eps = 0.001f;
    int64_t gamma_off = -1, beta_off = -1, mean_off = -1, var_off = -1;
    int c_pad = 0;
    // pool
    bool is_max = false;
    // dropout
    float keep = 1.0f;
    // group sum
    int group = 1;
};

static int64_t al64(int64_t v) { return (v + 63) / 64 * 64; }
}  // namespace tcr

using namespace tcr;

struct tcr_g2d {
    std::string scope;
    int in_h = 0, in_w = 0, in_c = 0;
    std::vector<G2dNode> nodes;
    int logits = -1, num_classes = 0;
    bool finalized = false;
    int64_t param_floats = 0, decay_floats = 0, stat_floats = 0, zeros_off = -1;
    int cmax = 0;
    std::vector<tcr_tensor_info> tensors;
};

namespace tcr {

static void shape_of(const tcr_g2d& g, int id, int* c, int* h, int* w) {
    if (id < 0) { *c = g.in_c; *h = g.in_h; *w = g.in_w; }
    else { *c = g.nodes[id].c; *h = g.nodes[id].h; *w = g.nodes[id].w; }
}

static int pp_of(int h, int w) { return h * w + 2 * kHalo; }

// TF output size / leading pad of one spatial dimension
static void out_dim(int len, int k_eff, int stride, bool valid, int* out, int* pad_lo) {
    if (valid) { *out = (len - k_eff) / stride + 1; *pad_lo = 0; }
    else {
        *out = (len + stride - 1) / stride;
        int total = (*out - 1) * stride + k_eff - len;
        if (total < 0) total = 0;
        *pad_lo = total / 2;
    }
}

struct G2dWorkspace {
    std::vector<int64_t> out, grad, mean, invstd, ss, argmax;
    int64_t partial = -1, kcoef = -1, scratch_c = -1, wgrad = -1, loss_utt = -1, dlogits = -1, sums = -1;
    int64_t grad_begin = 0, grad_end = 0;
    int64_t total = 0;
};

static G2dWorkspace carve2d(const tcr_g2d& g, int batch, bool train) {
    G2dWorkspace w;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t at = o; o += al64(n); return at; };
    const size_t nn = g.nodes.size();
    w.out.assign(nn, -1); w.grad.assign(nn, -1); w.mean.assign(nn, -1); w.invstd.assign(nn, -1); w.ss.assign(nn, -1); w.argmax.assign(nn, -1);
    int64_t wgmax = 0;
    for (size_t i = 0; i < nn; ++i) {
        const G2dNode& n = g.nodes[i];
        const int64_t sz = (int64_t)batch * n.c * pp_of(n.h, n.w);
        if (!(n.kind == G2D_DROPOUT && !train)) w.out[i] = take(sz);        // (eval: dropout is the identity -> aliases its input)
        if (n.kind == G2D_BN) {
            w.ss[i] = take(2 * n.c_pad);
            if (train) { w.mean[i] = take(n.c_pad); w.invstd[i] = take(n.c_pad); }
        }
        if (train && n.kind == G2D_POOL && n.is_max) w.argmax[i] = take((int64_t)batch * n.c * n.h * n.w);
        if (train && n.kind == G2D_CONV) {
            const int64_t wg = (int64_t)conv2d_wgrad_partial_floats(n.kh, n.kw, n.cin, n.c, batch);
            wgmax = wg > wgmax ? wg : wgmax;
        }
    }
    if (train) {
        w.grad_begin = o;
        for (size_t i = 0; i < nn; ++i) w.grad[i] = take((int64_t)batch * g.nodes[i].c * pp_of(g.nodes[i].h, g.nodes[i].w));
        w.grad_end = o;
        w.partial = take((int64_t)512 * 2 * g.cmax);
        w.sums = take(2 * 2 * al64(g.cmax));        // 2 x C float64 sums: the cross-replica BN hand-off
        w.kcoef = take(3 * al64(g.cmax));
        w.scratch_c = take(2 * al64(g.cmax));
        w.wgrad = take(wgmax);
        w.loss_utt = take(batch);
        w.dlogits = take((int64_t)batch * g.num_classes * pp_of(1, 1));
    }
    w.total = o;
    return w;
}

static const float* in_ptr(const tcr_g2d& g, const G2dWorkspace& w, const float* base, const float* x, int id, bool train) {
    if (id < 0) return x;
    const G2dNode& n = g.nodes[id];
    if (n.kind == G2D_DROPOUT && !train) return in_ptr(g, w, base, x, n.in0, train);
    return base + w.out[id];
}

// BN nodes in graph order: under cross-replica BN the forward / backward stop behind each one's statistics
static std::vector<int> bn_nodes(const tcr_g2d& g) {
    std::vector<int> v;
    for (size_t i = 0; i < g.nodes.size(); ++i) if (g.nodes[i].kind == G2D_BN) v.push_back((int)i);
    return v;
}

static Conv2dArgs conv_args(const tcr_g2d& g, const G2dNode& n, int batch) {
    Conv2dArgs a;
    std::memset(&a, 0, sizeof(a));
    int c, h, wd;
    shape_of(g, n.in0, &c, &h, &wd);
    a.batch = batch; a.cin = n.cin; a.cout = n.c; a.h = h; a.w = wd; a.oh = n.h; a.ow = n.w;
    a.kh = n.kh; a.kw = n.kw; a.sh = n.sh; a.sw = n.sw; a.dh = n.dh; a.dw = n.dw; a.pt = n.pt; a.pl = n.pl;
    a.ppi = pp_of(h, wd); a.ppo = pp_of(n.h, n.w); a.relu = n.relu ? 1 : 0;
    return a;
}

}  // namespace tcr

// ---- construction -----------------------------------------------------------------------------------------------------------
extern "C" int tcr_g2d_create(const char* scope, int h, int w, int c, tcr_g2d** out) {
    TCR_REQUIRE(scope && out && h > 0 && w > 0 && c > 0, "tcr_g2d_create: bad argument");
    tcr_g2d* g = new tcr_g2d();
    g->scope = scope; g->in_h = h; g->in_w = w; g->in_c = c;
    *out = g;
    return TCR_OK;
}

extern "C" void tcr_g2d_destroy(tcr_g2d* g) { delete g; }

#define G2D_CHECK_IN(g, id, what) TCR_REQUIRE((g) && !(g)->finalized && (id) >= -1 && (id) < (int)(g)->nodes.size(), what ": bad graph / input id")

extern "C" int tcr_g2d_conv(tcr_g2d* g, int in, int kh, int kw, int cout, int sh, int sw, int dh, int dw, int valid_padding, int relu,
                            const char* weights_name, const char* biases_name) {
    G2D_CHECK_IN(g, in, "tcr_g2d_conv");
    TCR_REQUIRE(kh > 0 && kw > 0 && cout > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0 && weights_name, "tcr_g2d_conv: bad argument");
    TCR_REQUIRE((dh == 1 && dw == 1) || (sh == 1 && sw == 1), "tcr_g2d_conv: dilation needs stride 1 (audio_nets/res.py:10)");
    G2dNode n;
    n.kind = G2D_CONV; n.in0 = in;
    int c, h, w;
    shape_of(*g, in, &c, &h, &w);
    n.cin = c; n.kh = kh; n.kw = kw; n.sh = sh; n.sw = sw; n.dh = dh; n.dw = dw; n.c = cout;
    out_dim(h, (kh - 1) * dh + 1, sh, valid_padding != 0, &n.h, &n.pt);
    out_dim(w, (kw - 1) * dw + 1, sw, valid_padding != 0, &n.w, &n.pl);
    TCR_REQUIRE(n.h > 0 && n.w > 0, "tcr_g2d_conv: %dx%d kernel does not fit the %dx%d input", kh, kw, h, w);
    n.relu = relu != 0; n.w_name = weights_name;
    if (biases_name && biases_name[0]) { n.bias = true; n.b_name = biases_name; }
    g->nodes.push_back(n);
    return (int)g->nodes.size() - 1;
}

extern "C" int tcr_g2d_batch_norm(tcr_g2d* g, int in, int center, int scale, int relu, float decay, float eps, const char* prefix) {
    G2D_CHECK_IN(g, in, "tcr_g2d_batch_norm");
    TCR_REQUIRE(in >= 0 && prefix, "tcr_g2d_batch_norm: bad argument");
    G2dNode n;
    n.kind = G2D_BN; n.in0 = in;
    shape_of(*g, in, &n.c, &n.h, &n.w);
    n.center = center != 0; n.scale = scale != 0; n.relu = relu != 0; n.decay = decay; n.eps = eps; n.bn_prefix = prefix;
    n.c_pad = (int)al64(n.c);
    g->nodes.push_back(n);
    return (int)g->nodes.size() - 1;
}

extern "C" int tcr_g2d_pool(tcr_g2d* g, int in, int is_max, int kh, int kw, int sh, int sw, int valid_padding) {
    G2D_CHECK_IN(g, in, "tcr_g2d_pool");
    G2dNode n;
    n.kind = G2D_POOL; n.in0 = in;
    int c, h, w;
    shape_of(*g, in, &c, &h, &w);
    if (kh <= 0 || kw <= 0) { kh = h; kw = w; sh = sw = 1; valid_padding = 1; }         // global pool (kernel = the whole plane)
    TCR_REQUIRE(sh > 0 && sw > 0, "tcr_g2d_pool: bad stride");
    n.c = c; n.kh = kh; n.kw = kw; n.sh = sh; n.sw = sw; n.is_max = is_max != 0;
    out_dim(h, kh, sh, valid_padding != 0, &n.h, &n.pt);
    out_dim(w, kw, sw, valid_padding != 0, &n.w, &n.pl);
    TCR_REQUIRE(n.h > 0 && n.w > 0, "tcr_g2d_pool: %dx%d window does not fit the %dx%d input", kh, kw, h, w);
    g->nodes.push_back(n);
    return (int)g->nodes.size() - 1;
}

extern "C" int tcr_g2d_add(tcr_g2d* g, int a, int b, int relu) {
    G2D_CHECK_IN(g, a, "tcr_g2d_add");
    G2D_CHECK_IN(g, b, "tcr_g2d_add");
    TCR_REQUIRE(a >= 0 && b >= 0, "tcr_g2d_add: the network input cannot be an operand");
    const G2dNode &na = g->nodes[a], &nb = g->nodes[b];
    TCR_REQUIRE(na.c == nb.c && na.h == nb.h && na.w == nb.w, "tcr_g2d_add: shapes differ (%d,%d,%d) vs (%d,%d,%d)", na.c, na.h, na.w, nb.c, nb.h, nb.w);
    G2dNode n;
    n.kind = G2D_ADD; n.in0 = a; n.in1 = b; n.c = na.c; n.h = na.h; n.w = na.w; n.relu = relu != 0;
    g->nodes.push_back(n);
    return (int)g->nodes.size() - 1;
}

extern "C" int tcr_g2d_dropout(tcr_g2d* g, int in, float keep_prob) {
    G2D_CHECK_IN(g, in, "tcr_g2d_dropout");
    TCR_REQUIRE(in >= 0 && keep_prob > 0.f && keep_prob <= 1.f, "tcr_g2d_dropout: bad argument");
    G2dNode n;
    n.kind = G2D_DROPOUT; n.in0 = in; n.keep = keep_prob;
    shape_of(*g, in, &n.c, &n.h, &n.w);
    g->nodes.push_back(n);
    return (int)g->nodes.size() - 1;
}

// SVDF (audio_nets/kws.py:490-680): per-channel filter over the whole input plane -> [C, 1, 1]; variable `weights_name` [C, plane]
extern "C" int tcr_g2d_time_filter(tcr_g2d* g, int in, const char* weights_name) {
    G2D_CHECK_IN(g, in, "tcr_g2d_time_filter");
    TCR_REQUIRE(weights_name && weights_name[0], "tcr_g2d_time_filter: the variable needs a name");
    G2dNode n;
    int c, h, w;
    shape_of(*g, in, &c, &h, &w);
    n.kind = G2D_TFILT; n.in0 = in; n.c = c; n.h = 1; n.w = 1; n.cin = h * w; n.w_name = weights_name;
    g->nodes.push_back(n);
    return (int)g->nodes.size() - 1;
}

// sum of `group` consecutive channels (+ bias) (+ ReLU) of a 1 x 1 node -> [C / group, 1, 1]
extern "C" int tcr_g2d_group_sum(tcr_g2d* g, int in, int group, int relu, const char* biases_name) {
    G2D_CHECK_IN(g, in, "tcr_g2d_group_sum");
    int c, h, w;
    shape_of(*g, in, &c, &h, &w);
    TCR_REQUIRE(h == 1 && w == 1 && group >= 1 && c % group == 0, "tcr_g2d_group_sum: needs a 1 x 1 input whose %d channels divide by the group %d", c, group);
    G2dNode n;
    n.kind = G2D_GSUM; n.in0 = in; n.c = c / group; n.h = 1; n.w = 1; n.group = group; n.relu = relu != 0;
    n.bias = biases_name && biases_name[0];
    if (n.bias) n.b_name = biases_name;
    g->nodes.push_back(n);
    return (int)g->nodes.size() - 1;
}

extern "C" int tcr_g2d_node_shape(const tcr_g2d* g, int node, int* c, int* h, int* w) {
    TCR_REQUIRE(g && c && h && w && node >= -1 && node < (int)g->nodes.size(), "tcr_g2d_node_shape: bad argument");
    shape_of(*g, node, c, h, w);
    return TCR_OK;
}

// Lays out the arenas: every variable whose name lacks "BatchNorm" / "batch_normalization" first (the L2-regularised set of
// factory/audio_nets.py:175-180 -- conv / fc weights AND biases), then the remaining BN gamma / beta, then a zero block that
// stands in for the beta of a BN built with center=False.
extern "C" int tcr_g2d_finalize(tcr_g2d* g, int logits_node) {
    TCR_REQUIRE(g && !g->finalized && logits_node >= 0 && logits_node < (int)g->nodes.size(), "tcr_g2d_finalize: bad argument");
    const G2dNode& ln = g->nodes[logits_node];
    TCR_REQUIRE(ln.h == 1 && ln.w == 1, "tcr_g2d_finalize: the logits node must be 1 x 1 spatially (got %d x %d)", ln.h, ln.w);
    g->logits = logits_node; g->num_classes = ln.c;
    auto info = [&](const std::string& name, int kind, int arena, int64_t off, int64_t size, std::vector<int> shape) {
        tcr_tensor_info ti;
        std::memset(&ti, 0, sizeof(ti));
        std::snprintf(ti.name, sizeof(ti.name), "%s", name.c_str());
        ti.kind = kind; ti.arena = arena; ti.offset = off; ti.size = size; ti.rank = (int)shape.size();
        for (size_t i = 0; i < shape.size() && i < 4; ++i) ti.shape[i] = shape[i];
        g->tensors.push_back(ti);
    };
    auto is_bn_name = [](const std::string& s) { return s.find("BatchNorm") != std::string::npos || s.find("batch_normalization") != std::string::npos; };
    int64_t o = 0, so = 0;
    int cmax = g->in_c;
    for (int pass = 0; pass < 2; ++pass) {          // pass 0: decayed variables, pass 1: BatchNorm-named ones
        for (G2dNode& n : g->nodes) {
            cmax = n.c > cmax ? n.c : cmax;
            if (n.kind == G2D_CONV && pass == 0) {
                const int64_t sz = (int64_t)n.kh * n.kw * n.cin * n.c;
                n.w_off = o;
                info(n.w_name, TCR_WEIGHT, 0, o, sz, {n.kh, n.kw, n.cin, n.c});
                o = al64(o + sz + 64);
                if (n.bias) {
                    n.b_off = o;
                    info(n.b_name, TCR_BETA, 0, o, n.c, {n.c});
                    o = al64(o + n.c + 64);
                }
            } else if (n.kind == G2D_TFILT && pass == 0) {
                const int64_t sz = (int64_t)n.c * n.cin;
                n.w_off = o;
                info(n.w_name, TCR_WEIGHT, 0, o, sz, {n.c, n.cin});
                o = al64(o + sz + 64);
            } else if (n.kind == G2D_GSUM && pass == 0) {
                if (n.bias) {
                    n.b_off = o;
                    info(n.b_name, TCR_BETA, 0, o, n.c, {n.c});
                    o = al64(o + n.c + 64);
                }
            } else if (n.kind == G2D_BN) {
                const bool late = is_bn_name(n.bn_prefix);
                if ((pass == 1) != late) continue;
                if (n.scale) { n.gamma_off = o; info(n.bn_prefix + "/gamma", TCR_GAMMA, 0, o, n.c, {n.c}); o += n.c_pad; }
                if (n.center) { n.beta_off = o; info(n.bn_prefix + "/beta", TCR_BETA, 0, o, n.c, {n.c}); o += n.c_pad; }
            }
        }
        if (pass == 0) g->decay_floats = o;
    }
    g->zeros_off = o;
    o += al64(cmax);
    for (G2dNode& n : g->nodes) {
        if (n.kind != G2D_BN) continue;
        n.mean_off = so; info(n.bn_prefix + "/moving_mean", TCR_MOVING_MEAN, 1, so, n.c, {n.c}); so += n.c_pad;
        n.var_off = so; info(n.bn_prefix + "/moving_variance", TCR_MOVING_VAR, 1, so, n.c, {n.c}); so += n.c_pad;
    }
    g->param_floats = o; g->stat_floats = so > 0 ? so : 64; g->cmax = cmax;
    g->finalized = true;
    return TCR_OK;
}

extern "C" int64_t tcr_g2d_param_floats(const tcr_g2d* g) { return g ? g->param_floats : 0; }
extern "C" int64_t tcr_g2d_decay_floats(const tcr_g2d* g) { return g ? g->decay_floats : 0; }
extern "C" int64_t tcr_g2d_stat_floats(const tcr_g2d* g) { return g ? g->stat_floats : 0; }
extern "C" int tcr_g2d_num_tensors(const tcr_g2d* g) { return g ? (int)g->tensors.size() : 0; }
extern "C" int tcr_g2d_num_classes(const tcr_g2d* g) { return g ? g->num_classes : 0; }
extern "C" int tcr_g2d_tensor_info(const tcr_g2d* g, int index, tcr_tensor_info* out) {
    TCR_REQUIRE(g && out && index >= 0 && index < (int)g->tensors.size(), "tcr_g2d_tensor_info: bad argument");
    *out = g->tensors[index];
    return TCR_OK;
}
extern "C" size_t tcr_g2d_workspace_bytes(const tcr_g2d* g, int batch, int train) {
    if (!g || !g->finalized || batch <= 0) return 0;
    return (size_t)carve2d(*g, batch, train != 0).total * sizeof(float);
}

// Where a node's output lives in the workspace of a forward call of this (batch, train): [batch, C, H*W + 2*halo] floats, `halo`
// floats in before the first element of every plane.  The reference keeps every layer's activation in `endpoints`
// (audio_nets/res.py:66, tc_resnet.py:95 `collect_named_outputs`); this is the same view onto the kernels' buffers.
extern "C" int tcr_g2d_node_output(const tcr_g2d* g, int node, int batch, int train, int64_t* offset_floats, int64_t* plane_floats,
                                   int* halo) {
    TCR_REQUIRE(g && g->finalized && offset_floats && plane_floats && halo && batch > 0 && node >= 0 && node < (int)g->nodes.size(),
                "tcr_g2d_node_output: bad argument");
    const G2dWorkspace w = carve2d(*g, batch, train != 0);
    int id = node;
    while (w.out[id] < 0) {                                  // eval-mode dropout aliases its input
        id = g->nodes[id].in0;
        TCR_REQUIRE(id >= 0, "tcr_g2d_node_output: node %d aliases the network input", node);
    }
    *offset_floats = w.out[id];
    *plane_floats = pp_of(g->nodes[id].h, g->nodes[id].w);
    *halo = kHalo;
    return TCR_OK;
}

extern "C" int tcr_g2d_input_from_features(const float* feat, int batch, int t, int f, float* plane, void* stream) {
    TCR_REQUIRE(feat && plane && batch > 0 && t > 0 && f > 0, "tcr_g2d_input_from_features: bad argument");
    return launch_features_to_plane(feat, plane, batch, t, f, static_cast<hipStream_t>(stream));
}

// ---- forward ------------------------------------------------------------------------------------------------------------------
static int g2d_forward(const tcr_g2d* g, const float* params, float* stats, const float* x, const float* labels, int batch, int global_batch,
                       bool train, uint64_t seed, int64_t sample_offset, float label_smoothing, void* workspace, size_t workspace_bytes,
                       float* logits, float* probs, float* loss_out, void* stream, int stage = -1) {
    // stage >= 0 (cross-replica BN): run from BN node #(stage - 1)'s finalize (stage 0: the input) up to BN node #stage's statistics, which
    // are left as float64 sums for the caller's all-reduce; the last stage finishes with the head
    TCR_REQUIRE(g && g->finalized && params && stats && x && workspace && logits && probs, "tcr_g2d_forward: null argument");
    TCR_REQUIRE(batch > 0 && global_batch >= batch, "tcr_g2d_forward: batch %d / global_batch %d", batch, global_batch);
    const G2dWorkspace w = carve2d(*g, batch, train);
    if ((size_t)w.total * sizeof(float) > workspace_bytes) {
        set_error("tcr_g2d_forward: workspace %zu bytes < required %zu", workspace_bytes, (size_t)w.total * sizeof(float));
        return TCR_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* base = static_cast<float*>(workspace);
    const bool staged = stage >= 0;
    const std::vector<int> bns = staged ? bn_nodes(*g) : std::vector<int>();
    TCR_REQUIRE(!staged || (train && stage <= (int)bns.size()), "tcr_g2d_forward_train_stage: bad stage %d", stage);
    const size_t i0 = staged && stage > 0 ? (size_t)bns[stage - 1] : 0;
    for (size_t i = i0; i < g->nodes.size(); ++i) {
        const G2dNode& n = g->nodes[i];
        const float* in0 = in_ptr(*g, w, base, x, n.in0, train);
        float* out = w.out[i] >= 0 ? base + w.out[i] : nullptr;
        const int pp = pp_of(n.h, n.w);
        switch (n.kind) {
            case G2D_CONV: {
                Conv2dArgs a = conv_args(*g, n, batch);
                a.x = in0; a.wgt = params + n.w_off; a.y = out; a.bias = n.bias ? params + n.b_off : nullptr;
                TCR_TRY(launch_conv2d_fwd(a, s));
                break;
            }
            case G2D_BN: {
                float* ss = base + w.ss[i];
                const float* beta = params + (n.center ? n.beta_off : g->zeros_off);
                if (train) {
                    const bool resume = staged && stage > 0 && (int)i == bns[stage - 1];      // (its statistics were handed over)
                    int nchunk = 0;
                    if (!resume) {
                        ChanReduceArgs r;
                        std::memset(&r, 0, sizeof(r));
                        r.y = in0; r.partial = base + w.partial; r.npos = batch * n.h * n.w; r.c = n.c; r.t = n.h * n.w; r.tp = pp;
                        TCR_TRY(launch_chan_reduce(0, r, &nchunk, s));
                        if (staged) {
                            TCR_TRY(launch_chan_sums(base + w.partial, nchunk, n.c, reinterpret_cast<double*>(base + w.sums), s));
                            return TCR_OK;          // hand-off: the caller all-reduces the sums, then runs the next stage
                        }
                    }
                    BnFinalizeArgs f;
                    f.partial = base + w.partial; f.nchunk = staged ? 0 : nchunk; f.sums = staged ? reinterpret_cast<const double*>(base + w.sums) : nullptr;
                    f.gamma = n.scale ? params + n.gamma_off : nullptr; f.beta = beta;
                    f.moving_mean = stats + n.mean_off; f.moving_var = stats + n.var_off;
                    f.scale = ss; f.shift = ss + n.c_pad; f.mean = base + w.mean[i]; f.invstd = base + w.invstd[i];
                    f.c = n.c; f.count = (double)(staged ? global_batch : batch) * (double)(n.h * n.w); f.decay = n.decay; f.eps = n.eps;
                    TCR_TRY(launch_bn_finalize(f, s));
                } else {
                    BnFoldArgs f;
                    std::memset(&f, 0, sizeof(f));
                    f.params = params; f.stats = stats; f.out = ss; f.eps = n.eps; f.n = 1;
                    f.c[0] = n.c; f.c_pad[0] = n.c_pad; f.gamma_off[0] = n.scale ? n.gamma_off : -1;
                    f.beta_off[0] = n.center ? n.beta_off : g->zeros_off;
                    f.mean_off[0] = n.mean_off; f.var_off[0] = n.var_off; f.out_off[0] = 0; f.bias_off[0] = -1;
                    TCR_TRY(launch_bn_fold(f, s));
                }
                BnApplyArgs a;
                a.y = in0; a.scale = ss; a.shift = ss + n.c_pad; a.res = nullptr; a.out = out;
                a.total = (int64_t)batch * n.c * pp; a.c = n.c; a.t = n.h * n.w; a.tp = pp; a.relu = n.relu ? 1 : 0;
                TCR_TRY(launch_bn_apply(a, s));
                break;
            }
            case G2D_POOL: {
                int c, h, wd;
                shape_of(*g, n.in0, &c, &h, &wd);
                Pool2dArgs a;
                std::memset(&a, 0, sizeof(a));
                a.x = in0; a.y = out; a.argmax = (train && n.is_max) ? reinterpret_cast<int*>(base + w.argmax[i]) : nullptr;
                a.planes = (int64_t)batch * n.c; a.h = h; a.w = wd; a.oh = n.h; a.ow = n.w; a.kh = n.kh; a.kw = n.kw; a.sh = n.sh; a.sw = n.sw;
                a.pt = n.pt; a.pl = n.pl; a.ppi = pp_of(h, wd); a.ppo = pp; a.is_max = n.is_max ? 1 : 0;
                TCR_TRY(launch_pool2d_fwd(a, s));
                break;
            }
            case G2D_ADD: {
                Elt2dArgs a;
                std::memset(&a, 0, sizeof(a));
                a.a = in0; a.b = in_ptr(*g, w, base, x, n.in1, train); a.out = out;
                a.planes = (int64_t)batch * n.c; a.c = n.c; a.plane = n.h * n.w; a.pp = pp; a.relu = n.relu ? 1 : 0;
                TCR_TRY(launch_eltwise2d(0, a, s));
                break;
            }
            case G2D_TFILT: {
                TCR_TRY(launch_tfilt_fwd(in0, params + n.w_off, out, batch, n.c, n.cin, s));
                break;
            }
            case G2D_GSUM: {
                TCR_TRY(launch_gsum_fwd(in0, n.bias ? params + n.b_off : nullptr, out, batch, n.c, n.group, n.relu ? 1 : 0, s));
                break;
            }
            case G2D_DROPOUT: {
                if (!train) break;              // identity: consumers read the input (in_ptr)
                Elt2dArgs a;
                std::memset(&a, 0, sizeof(a));
                a.a = in0; a.out = out; a.planes = (int64_t)batch * n.c; a.c = n.c; a.plane = n.h * n.w; a.pp = pp;
                a.keep_prob = n.keep; a.seed = seed; a.sample_offset = sample_offset; a.node = (int)i;
                TCR_TRY(launch_eltwise2d(3, a, s));
                break;
            }
        }
    }
    const float* z = in_ptr(*g, w, base, x, g->logits, train);
    TCR_TRY(launch_head2d(z, train ? labels : nullptr, logits, probs, train ? base + w.dlogits : nullptr, train ? base + w.loss_utt : nullptr,
                          batch, g->num_classes, pp_of(1, 1), 1.0f / (float)global_batch, label_smoothing, s));
    if (train) TCR_TRY(launch_sum_vector(base + w.loss_utt, batch, loss_out, s));
    return TCR_OK;
}

extern "C" int tcr_g2d_forward_infer(const tcr_g2d* g, const float* params, const float* stats, const float* x, int batch,
                                     void* workspace, size_t workspace_bytes, float* logits, float* probs, void* stream) {
    return g2d_forward(g, params, const_cast<float*>(stats), x, nullptr, batch, batch, false, 0, 0, 0.f, workspace, workspace_bytes, logits, probs,
                       nullptr, stream);
}

extern "C" int tcr_g2d_forward_train(const tcr_g2d* g, const float* params, float* stats, const float* x, const float* labels, int batch,
                                     int global_batch, uint64_t seed, int64_t sample_offset, float label_smoothing, void* workspace,
                                     size_t workspace_bytes, float* logits, float* probs, float* loss_out, void* stream) {
    TCR_REQUIRE(labels && loss_out, "tcr_g2d_forward_train: null argument");
    return g2d_forward(g, params, stats, x, labels, batch, global_batch, true, seed, sample_offset, label_smoothing, workspace, workspace_bytes,
                       logits, probs, loss_out, stream);
}

// ---- backward -----------------------------------------------------------------------------------------------------------------
extern "C" int tcr_g2d_num_stages(const tcr_g2d* g) { return g ? (int)tcr::bn_nodes(*g).size() + 1 : 0; }

extern "C" int tcr_g2d_forward_train_stage(const tcr_g2d* g, const float* params, float* stats, const float* x, const float* labels, int batch,
                                           int global_batch, uint64_t seed, int64_t sample_offset, float label_smoothing, void* workspace,
                                           size_t workspace_bytes, float* logits, float* probs, float* loss_out, int stage, void* stream) {
    TCR_REQUIRE(labels && loss_out && stage >= 0, "tcr_g2d_forward_train_stage: bad argument");
    return g2d_forward(g, params, stats, x, labels, batch, global_batch, true, seed, sample_offset, label_smoothing, workspace, workspace_bytes,
                       logits, probs, loss_out, stream, stage);
}

// float64 sums handed over after stage `stage` (< tcr_g2d_num_stages() - 1): 2 x C of the stage's BN node (backward: reverse graph order)
extern "C" int tcr_g2d_stage_sums(const tcr_g2d* g, int backward, int stage, void* workspace, int batch, double** sums_dev, int64_t* n_doubles) {
    TCR_REQUIRE(g && g->finalized && workspace && sums_dev && n_doubles, "tcr_g2d_stage_sums: bad argument");
    const std::vector<int> bns = tcr::bn_nodes(*g);
    TCR_REQUIRE(stage >= 0 && stage < (int)bns.size(), "tcr_g2d_stage_sums: stage %d has no hand-off", stage);
    const G2dWorkspace w = carve2d(*g, batch, true);
    const int node = backward ? bns[bns.size() - 1 - stage] : bns[stage];
    *sums_dev = reinterpret_cast<double*>(static_cast<float*>(workspace) + w.sums);
    *n_doubles = 2 * (int64_t)g->nodes[node].c;
    return TCR_OK;
}

// stage < 0: the whole backward.  stage >= 0 (cross-replica BN): from BN node #(stage - 1) (in REVERSE graph order)'s finalize -- stage 0:
// the loss gradient -- down to the next BN node's sums.
static int g2d_backward(const tcr_g2d* g, const float* params, const float* x, int batch, int global_batch, uint64_t seed, int64_t sample_offset,
                        void* workspace, size_t workspace_bytes, float* grads, void* stream, int stage) {
    TCR_REQUIRE(g && g->finalized && params && x && workspace && grads && batch > 0, "tcr_g2d_backward: bad argument");
    const G2dWorkspace w = carve2d(*g, batch, true);
    if ((size_t)w.total * sizeof(float) > workspace_bytes) {
        set_error("tcr_g2d_backward: workspace %zu bytes < required %zu", workspace_bytes, (size_t)w.total * sizeof(float));
        return TCR_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* base = static_cast<float*>(workspace);
    const bool staged = stage >= 0;
    std::vector<int> rbn;
    if (staged) { rbn = bn_nodes(*g); std::reverse(rbn.begin(), rbn.end()); }
    TCR_REQUIRE(!staged || stage <= (int)rbn.size(), "tcr_g2d_backward_stage: bad stage %d", stage);
    if (stage <= 0) {
    if (hipMemsetAsync(grads, 0, (size_t)g->param_floats * sizeof(float), s) != hipSuccess ||
        hipMemsetAsync(base + w.grad_begin, 0, (size_t)(w.grad_end - w.grad_begin) * sizeof(float), s) != hipSuccess) {
        set_error("tcr_g2d_backward: hipMemsetAsync failed");
        return TCR_ERR_HIP;
    }
    }
    if (stage <= 0) {   // seed: d loss / d logits
        Elt2dArgs a;
        std::memset(&a, 0, sizeof(a));
        a.a = base + w.dlogits; a.out = base + w.grad[g->logits];
        a.planes = (int64_t)batch * g->num_classes; a.c = g->num_classes; a.plane = 1; a.pp = pp_of(1, 1);
        TCR_TRY(launch_eltwise2d(2, a, s));
    }
    const int istart = staged && stage > 0 ? rbn[stage - 1] : (int)g->nodes.size() - 1;
    for (int i = istart; i >= 0; --i) {
        const G2dNode& n = g->nodes[i];
        float* G = base + w.grad[i];
        const float* out = base + w.out[i];
        const float* in0 = in_ptr(*g, w, base, x, n.in0, true);
        float* gin0 = n.in0 >= 0 ? base + w.grad[n.in0] : nullptr;
        const int pp = pp_of(n.h, n.w);
        auto relu_mask = [&]() -> int {
            Elt2dArgs a;
            std::memset(&a, 0, sizeof(a));
            a.a = out; a.out = G; a.planes = (int64_t)batch * n.c; a.c = n.c; a.plane = n.h * n.w; a.pp = pp;
            return launch_eltwise2d(1, a, s);
        };
        auto fan_out = [&](float* dst) -> int {
            Elt2dArgs a;
            std::memset(&a, 0, sizeof(a));
            a.a = G; a.out = dst; a.planes = (int64_t)batch * n.c; a.c = n.c; a.plane = n.h * n.w; a.pp = pp;
            return launch_eltwise2d(2, a, s);
        };
        switch (n.kind) {
            case G2D_CONV: {
                if (n.relu) TCR_TRY(relu_mask());
                Conv2dArgs a = conv_args(*g, n, batch);
                a.wgt = params + n.w_off;
                if (n.bias) TCR_TRY(launch_chan_sum2d(G, grads + n.b_off, batch, n.c, n.h * n.w, pp, s));
                a.x = in0; a.dy = G;
                TCR_TRY(launch_conv2d_wgrad(a, grads + n.w_off, base + w.wgrad, s));
                if (gin0) {
                    a.x = G; a.y = gin0;
                    TCR_TRY(launch_conv2d_dgrad(a, s));
                }
                break;
            }
            case G2D_BN: {
                ChanReduceArgs r;
                std::memset(&r, 0, sizeof(r));
                r.y = in0; r.da = G; r.m1 = n.relu ? out : nullptr; r.m2 = nullptr;
                r.mean = base + w.mean[i]; r.invstd = base + w.invstd[i]; r.partial = base + w.partial;
                r.npos = batch * n.h * n.w; r.c = n.c; r.t = n.h * n.w; r.tp = pp; r.bcast = 0;
                int nchunk = 0;
                const bool resume = staged && stage > 0 && i == rbn[stage - 1];       // (its sums were handed over)
                if (!resume) {
                    TCR_TRY(launch_chan_reduce(1, r, &nchunk, s));
                    if (staged) {
                        TCR_TRY(launch_chan_sums(base + w.partial, nchunk, n.c, reinterpret_cast<double*>(base + w.sums), s));
                        return TCR_OK;
                    }
                }
                float* kc = base + w.kcoef;
                const int64_t ks = al64(g->cmax);
                BnBwdFinalizeArgs f;
                f.partial = base + w.partial; f.nchunk = staged ? 0 : nchunk; f.sums = staged ? reinterpret_cast<const double*>(base + w.sums) : nullptr;
                f.gamma = n.scale ? params + n.gamma_off : nullptr; f.invstd = base + w.invstd[i];
                f.dgamma = n.scale ? grads + n.gamma_off : nullptr;
                f.dbeta = n.center ? grads + n.beta_off : base + w.scratch_c;
                f.k1 = kc; f.k2 = kc + ks; f.k3 = kc + 2 * ks;
                f.c = n.c; f.count = (double)(staged ? global_batch : batch) * (double)(n.h * n.w);
                f.grad_scale = staged ? (float)((double)batch / (double)global_batch) : 1.0f;
                TCR_TRY(launch_bn_bwd_finalize(f, s));
                if (gin0) {
                    BnBwdApplyArgs a;
                    a.y = in0; a.da = G; a.m1 = r.m1; a.m2 = nullptr; a.mean = base + w.mean[i];
                    a.k1 = f.k1; a.k2 = f.k2; a.k3 = f.k3; a.dy = gin0;
                    a.total = (int64_t)batch * n.c * pp; a.c = n.c; a.t = n.h * n.w; a.tp = pp; a.bcast = 0; a.accumulate = 1;
                    TCR_TRY(launch_bn_bwd_apply(a, s));
                }
                break;
            }
            case G2D_POOL: {
                if (!gin0) break;
                int c, h, wd;
                shape_of(*g, n.in0, &c, &h, &wd);
                Pool2dArgs a;
                std::memset(&a, 0, sizeof(a));
                a.dy = G; a.dx = gin0; a.argmax = n.is_max ? reinterpret_cast<int*>(base + w.argmax[i]) : nullptr;
                a.planes = (int64_t)batch * n.c; a.h = h; a.w = wd; a.oh = n.h; a.ow = n.w; a.kh = n.kh; a.kw = n.kw; a.sh = n.sh; a.sw = n.sw;
                a.pt = n.pt; a.pl = n.pl; a.ppi = pp_of(h, wd); a.ppo = pp; a.is_max = n.is_max ? 1 : 0;
                TCR_TRY(launch_pool2d_bwd(a, s));
                break;
            }
            case G2D_ADD: {
                if (n.relu) TCR_TRY(relu_mask());
                TCR_TRY(fan_out(base + w.grad[n.in0]));
                TCR_TRY(fan_out(base + w.grad[n.in1]));
                break;
            }
            case G2D_TFILT: {
                TCR_TRY(launch_tfilt_bwd(G, in0, params + n.w_off, gin0, grads + n.w_off, batch, n.c, n.cin, s));
                break;
            }
            case G2D_GSUM: {
                if (n.relu) TCR_TRY(relu_mask());
                if (n.bias) TCR_TRY(launch_chan_sum2d(G, grads + n.b_off, batch, n.c, 1, pp, s));
                if (gin0) TCR_TRY(launch_gsum_dx(G, gin0, batch, n.c, n.group, s));
                break;
            }
            case G2D_DROPOUT: {
                Elt2dArgs a;
                std::memset(&a, 0, sizeof(a));
                a.a = G; a.out = gin0; a.planes = (int64_t)batch * n.c; a.c = n.c; a.plane = n.h * n.w; a.pp = pp;
                a.keep_prob = n.keep; a.seed = seed; a.sample_offset = sample_offset; a.node = i;
                TCR_TRY(launch_eltwise2d(4, a, s));
                break;
            }
        }
    }
    return TCR_OK;
}

extern "C" int tcr_g2d_backward(const tcr_g2d* g, const float* params, const float* x, int batch, uint64_t seed, int64_t sample_offset,
                                void* workspace, size_t workspace_bytes, float* grads, void* stream) {
    return g2d_backward(g, params, x, batch, batch, seed, sample_offset, workspace, workspace_bytes, grads, stream, -1);
}

extern "C" int tcr_g2d_backward_stage(const tcr_g2d* g, const float* params, const float* x, int batch, int global_batch, uint64_t seed,
                                      int64_t sample_offset, void* workspace, size_t workspace_bytes, float* grads, int stage, void* stream) {
    TCR_REQUIRE(stage >= 0 && global_batch >= batch, "tcr_g2d_backward_stage: bad argument");
    return g2d_backward(g, params, x, batch, global_batch, seed, sample_offset, workspace, workspace_bytes, grads, stream, stage);
}
