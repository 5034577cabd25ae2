// TC-ResNet engine: turns a tcr_tcresnet_cfg into a layer table (the topology of
// audio_nets/tc_resnet.py:6-54), lays out the flat parameter / moving-stat arenas and the caller's
// workspace, and sequences the gfx950 kernels for eval-mode forward, train-mode forward and
// backward.  Host-only state; every device buffer belongs to the caller.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.h"

namespace tcr {

struct ConvLayer {
    std::string name;           // scope under the model scope, e.g. "block0/conv0_0"
    int k = 1, stride = 1, cin = 0, cout = 0;
    int tin = 0, tout = 0, pad_lo = 0, pad_hi = 0;
    bool bn = true, relu = true;
    int64_t w_off = -1, gamma_off = -1, beta_off = -1;      // trainable arena (floats)
    int64_t mean_off = -1, var_off = -1;                    // moving-stat arena (floats)
    int c_pad = 0;
    int64_t ss_off = -1;        // scale/shift slot inside the workspace "ss" region (floats)
    int in_act = -1;            // index of the activation feeding this conv (-1: the feature input)
};

struct Block {
    int down = -1, a = -1, b = -1;
};

}  // namespace tcr

using namespace tcr;

struct tcr_net {
    tcr_tcresnet_cfg cfg;
    std::vector<ConvLayer> layers;      // conv0, per block [down], a, b; then fc, fc2
    std::vector<Block> blocks;
    int fc = -1, fc2 = -1;
    std::vector<int> units;             // BN'd conv layers in forward execution order
    int64_t param_floats = 0, decay_floats = 0, stat_floats = 0;
    std::vector<tcr_tensor_info> tensors;
    int feat_c = 0, feat_t = 0;
    // Backward runs the weight-gradient kernels on a second stream: they only consume (x, dy) of their own layer and write
    // their own split-K slab, so they overlap the data-gradient / BN-backward chain of the layers below (each kernel of a
    // training step is too short to fill the chip on its own).  Created on first use; joined before the slab reduction.
    mutable hipStream_t side = nullptr;
    mutable hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_down = nullptr, ev_join2 = nullptr, ev_down_dg = nullptr;
    mutable hipStream_t side2 = nullptr;        // the classifier's filter gradient (nothing below depends on it)
    ~tcr_net() {
        if (ev_join2) (void)hipEventDestroy(ev_join2);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (ev_down) (void)hipEventDestroy(ev_down);
        if (ev_down_dg) (void)hipEventDestroy(ev_down_dg);
    }
};

namespace tcr {

static int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// filter gradients that get the finer split-K grid (mfma.hip: wgrad_chunks_for)
// (measured: TCResNet8 step -1.5 %; with TCResNet14-1.5's 24 -> 36 channel layers included that step was +1.3 %: their larger slabs cost
//  more in the reduction than the extra waves win)
// (the first conv too: TCResNet8's 40 -> 16 first conv -1.3 % per step; TCResNet14-1.5's 40 -> 24 one, a six-tile slab: +0.5 %, so the same bound)
static bool wgrad_fine(const ConvLayer& l) { return (l.k == 9 || l.in_act < 0) && ((l.cin + 15) / 16) * ((l.cout + 15) / 16) <= 4 && l.tout >= 20; }

// ---- workspace carving ------------------------------------------------------------------------
struct Workspace {
    // offsets in floats; -1 when absent
    int64_t ss = 0;                         // folded / batch scale+shift per BN layer
    std::vector<int64_t> act;               // activation (post BN/ReLU/residual) per conv layer
    std::vector<int64_t> raw;               // raw conv output per BN layer (train)
    std::vector<int64_t> dyb;               // grad wrt raw conv output (train)
    std::vector<int64_t> gact;              // grad wrt activation per conv layer (train)
    std::vector<int64_t> mean, invstd;      // saved batch statistics (train)
    std::vector<int64_t> wg, wtl;           // per-layer wgrad partial slabs / re-arranged dgrad weights (train)
    std::vector<int64_t> kc;                // per-unit k1 / k2 / k3 rows of the lazy backward (its filter gradients read them on other streams, later)
    int64_t partial = -1, sums = -1, sums_slot = 0, kcoef = -1;    // sums: two slots of sums_slot floats (2 x Cmax doubles): main units / shortcut units
    int64_t partial2 = -1, kcoef2 = -1;     // second set: the shortcut branch's BN backward runs concurrently on the side stream
    int64_t dropped = -1, dscale = -1, dlogits = -1, loss_utt = -1, dpool = -1;
    int64_t wgrad_scratch = -1, wt = -1, fc_partial = -1;
    int64_t total = 0;
};

static Workspace carve(const tcr_net& net, int batch, bool train) {
    Workspace w;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t at = o; o += align_up(n, 64); return at; };
    const size_t nl = net.layers.size();
    w.act.assign(nl, -1); w.raw.assign(nl, -1); w.dyb.assign(nl, -1); w.gact.assign(nl, -1);
    w.mean.assign(nl, -1); w.invstd.assign(nl, -1); w.wg.assign(nl, -1); w.wtl.assign(nl, -1); w.kc.assign(nl, -1);
    int64_t ss = 0;
    for (const ConvLayer& l : net.layers) if (l.bn) ss += 2 * l.c_pad;
    w.ss = take(ss);
    int cmax = 0;
    int64_t wmax = 0, wgmax = 0;
    for (size_t i = 0; i < nl; ++i) {
        const ConvLayer& l = net.layers[i];
        if (!l.bn) continue;
        const int64_t n = (int64_t)batch * l.cout * tcr_padded_len(l.tout);
        w.act[i] = take(n);
        if (train) {
            w.raw[i] = take(n);
            w.dyb[i] = take(n);
            w.gact[i] = take(n);
            w.mean[i] = take(l.c_pad);
            w.invstd[i] = take(l.c_pad);
            w.wg[i] = take((int64_t)wgrad_partial_floats(l.k, l.cin, l.cout, batch, true));        // (sized for the finer grid; which layers use it: wgrad_fine)
            w.wtl[i] = take((int64_t)l.k * l.cin * l.cout);
            w.kc[i] = take(3 * align_up(l.cout, 64) + 8 * (int64_t)l.cout);    // k1, k2, k3 rows + the packed [c][8] table
        }
        cmax = l.cout > cmax ? l.cout : cmax;
        const int64_t wsz = (int64_t)l.k * l.cin * l.cout;
        wmax = wsz > wmax ? wsz : wmax;
        const int64_t wg = (int64_t)wgrad_partial_floats(l.k, l.cin, l.cout, batch);
        wgmax = wg > wgmax ? wg : wgmax;
    }
    if (train) {
        w.partial = take((int64_t)kPhaseMaxRows * 2 * cmax);
        w.sums_slot = align_up(2 * 2 * (int64_t)cmax, 64);        // 2 x Cmax doubles
        w.sums = take(2 * w.sums_slot);     // slot 0: main-chain units, slot 1: the blocks' shortcut (`down`) units -- a dependency level hands both over at once
        w.kcoef = take(3 * (int64_t)align_up(cmax, 64));
        w.partial2 = take((int64_t)kPhaseMaxRows * 2 * cmax);
        w.kcoef2 = take(3 * (int64_t)align_up(cmax, 64));
        const int c = net.feat_c, nc = net.cfg.num_classes;
        w.dropped = take((int64_t)batch * c);
        w.dscale = take((int64_t)batch * c);
        w.dpool = take((int64_t)batch * c);
        w.dlogits = take((int64_t)batch * nc);
        w.loss_utt = take(batch);
        w.wgrad_scratch = take(wgmax);
        w.wt = take(wmax);
        w.fc_partial = take((int64_t)fc_wgrad_chunks(batch) * c * nc);
    }
    w.total = o;
    return w;
}

}  // namespace tcr

// ---- construction -----------------------------------------------------------------------------
extern "C" int tcr_tcresnet_create(const tcr_tcresnet_cfg* cfg, tcr_net** out) {
    TCR_REQUIRE(cfg && out, "tcr_tcresnet_create: null argument");
    TCR_REQUIRE(cfg->n_blocks >= 1 && cfg->n_blocks <= TCR_MAX_BLOCKS, "tcr_tcresnet_create: n_blocks %d out of range", cfg->n_blocks);
    TCR_REQUIRE(cfg->in_channels > 0 && cfg->t_in > 0 && cfg->num_classes > 0, "tcr_tcresnet_create: bad input shape");
    TCR_REQUIRE(cfg->num_classes + 2 <= 48, "tcr_tcresnet_create: num_classes %d exceeds the 46-class head of this build", cfg->num_classes);
    for (int i = 0; i <= cfg->n_blocks; ++i)
        TCR_REQUIRE(cfg->channels[i] > 0, "tcr_tcresnet_create: channels[%d] = %d", i, cfg->channels[i]);
    TCR_REQUIRE(std::strlen(cfg->scope) > 0 && std::strlen(cfg->scope) < sizeof(cfg->scope), "tcr_tcresnet_create: empty scope");

    tcr_net* net = new tcr_net();
    net->cfg = *cfg;
    auto add_conv = [&](const std::string& name, int k, int stride, int cin, int cout, int tin, bool bn, bool relu, int in_act) {
        ConvLayer l;
        l.name = name; l.k = k; l.stride = stride; l.cin = cin; l.cout = cout; l.tin = tin;
        same_pad(tin, k, stride, &l.tout, &l.pad_lo, &l.pad_hi);
        l.bn = bn; l.relu = relu; l.c_pad = (int)align_up(cout, 64); l.in_act = in_act;
        net->layers.push_back(l);
        return (int)net->layers.size() - 1;
    };
    // tc_resnet(): conv0 3x1 s1 (:21); per block: [down 1x1 s2] (:29-32), conv 9x1 s, conv 9x1 s1 no activation (:37-39)
    int cur = add_conv("conv0", 3, 1, cfg->in_channels, cfg->channels[0], cfg->t_in, true, true, -1);
    net->units.push_back(cur);
    int c = cfg->channels[0], t = net->layers[cur].tout;
    for (int i = 0; i < cfg->n_blocks; ++i) {
        const int n = cfg->channels[i + 1];
        Block b;
        int stride = 1;
        const std::string pre = "block" + std::to_string(i) + "/";
        if (n != c) {
            stride = 2;
            b.down = add_conv(pre + "down", 1, 2, c, n, t, true, true, cur);
            net->units.push_back(b.down);
        }
        b.a = add_conv(pre + "conv" + std::to_string(i) + "_0", 9, stride, c, n, t, true, true, cur);
        net->units.push_back(b.a);
        b.b = add_conv(pre + "conv" + std::to_string(i) + "_1", 9, 1, n, n, net->layers[b.a].tout, true, false, b.a);
        net->units.push_back(b.b);
        net->blocks.push_back(b);
        cur = b.b;
        c = n;
        t = net->layers[b.b].tout;
    }
    net->feat_c = c;
    net->feat_t = t;
    net->fc = add_conv("fc", 1, 1, c, cfg->num_classes, 1, false, false, cur);     // :47
    net->fc2 = add_conv("fc2", 1, 1, c, 2, 1, false, false, cur);                  // :50
    for (const ConvLayer& l : net->layers) {
        if (l.pad_lo > TCR_HALO || l.pad_hi > TCR_HALO) {
            set_error("tcr_tcresnet_create: layer %s needs padding (%d,%d) beyond the %d-sample halo", l.name.c_str(), l.pad_lo, l.pad_hi, TCR_HALO);
            delete net;
            return TCR_ERR_ARG;
        }
    }
    if ((int)net->units.size() > kBnMaxLayers) {
        set_error("tcr_tcresnet_create: %d BN layers exceed the limit %d", (int)net->units.size(), kBnMaxLayers);
        delete net;
        return TCR_ERR_ARG;
    }

    // arenas: weights (L2-decayed) first, then gamma/beta; every tensor starts on a 64-float
    // boundary and is followed by >= 64 floats of zero padding (channel-tile over-reads land there).
    const std::string scope = cfg->scope;
    auto info = [&](const std::string& name, int kind, int arena, int64_t off, int64_t size, std::vector<int> shape) {
        tcr_tensor_info ti;
        std::memset(&ti, 0, sizeof(ti));
        std::snprintf(ti.name, sizeof(ti.name), "%s/%s", scope.c_str(), name.c_str());
        ti.kind = kind; ti.arena = arena; ti.offset = off; ti.size = size; ti.rank = (int)shape.size();
        for (size_t i = 0; i < shape.size() && i < 4; ++i) ti.shape[i] = shape[i];
        net->tensors.push_back(ti);
    };
    int64_t o = 0;
    for (ConvLayer& l : net->layers) {
        l.w_off = o;
        const int64_t n = (int64_t)l.k * l.cin * l.cout;
        info(l.name + "/weights", TCR_WEIGHT, 0, o, n, {l.k, 1, l.cin, l.cout});
        o = align_up(o + n + 64, 64);
    }
    net->decay_floats = o;
    int64_t so = 0, ss = 0;
    for (ConvLayer& l : net->layers) {
        if (!l.bn) continue;
        l.gamma_off = o;
        info(l.name + "/BatchNorm/gamma", TCR_GAMMA, 0, o, l.cout, {l.cout});
        o += l.c_pad;
        l.beta_off = o;
        info(l.name + "/BatchNorm/beta", TCR_BETA, 0, o, l.cout, {l.cout});
        o += l.c_pad;
        l.mean_off = so;
        info(l.name + "/BatchNorm/moving_mean", TCR_MOVING_MEAN, 1, so, l.cout, {l.cout});
        so += l.c_pad;
        l.var_off = so;
        info(l.name + "/BatchNorm/moving_variance", TCR_MOVING_VAR, 1, so, l.cout, {l.cout});
        so += l.c_pad;
        l.ss_off = ss;
        ss += 2 * l.c_pad;
    }
    net->param_floats = o;
    net->stat_floats = so;
    *out = net;
    return TCR_OK;
}

extern "C" void tcr_net_destroy(tcr_net* net) { delete net; }
extern "C" int64_t tcr_net_param_floats(const tcr_net* net) { return net ? net->param_floats : 0; }
extern "C" int64_t tcr_net_decay_floats(const tcr_net* net) { return net ? net->decay_floats : 0; }
extern "C" int64_t tcr_net_stat_floats(const tcr_net* net) { return net ? net->stat_floats : 0; }
extern "C" int tcr_net_num_tensors(const tcr_net* net) { return net ? (int)net->tensors.size() : 0; }
extern "C" int tcr_net_out_frames(const tcr_net* net) { return net ? net->feat_t : 0; }
extern "C" int tcr_net_feat_channels(const tcr_net* net) { return net ? net->feat_c : 0; }

extern "C" int tcr_net_tensor_info(const tcr_net* net, int index, tcr_tensor_info* out) {
    TCR_REQUIRE(net && out, "tcr_net_tensor_info: null argument");
    TCR_REQUIRE(index >= 0 && index < (int)net->tensors.size(), "tcr_net_tensor_info: index %d out of range", index);
    *out = net->tensors[index];
    return TCR_OK;
}

extern "C" size_t tcr_net_workspace_bytes(const tcr_net* net, int batch, int train) {
    if (!net || batch <= 0) return 0;
    return (size_t)carve(*net, batch, train != 0).total * sizeof(float);
}

extern "C" int tcr_net_num_stages(const tcr_net* net, int backward) {
    (void)backward;
    return net ? (int)net->units.size() + 1 : 0;
}

// ---- dependency levels of the cross-replica BN hand-off ----------------------------------------------------------------------
// A level = the BN units whose batch statistics become available together.  Forward: conv0 | per block (down, conv_a) | conv_b.
// Backward: per block, last first, (conv_b, down) | conv_a; then conv0 -- a block's shortcut BN backward depends only on the
// block-output gradient, like conv_b's, but its data gradient must be added AFTER conv_a's, so its second half runs one level late.
namespace tcr {
struct LevelPlan {
    std::vector<int> post;      // units finished first (finalize + everything downstream of their statistics) ...
    std::vector<int> pre;       // ... then units whose statistics are produced (one hand-off for all of them)
    bool first = false, last = false;
};
static std::vector<std::vector<int>> level_units(const tcr_net& net, int backward) {
    std::vector<std::vector<int>> lv;
    if (!backward) {
        lv.push_back({0});
        for (const Block& b : net.blocks) {
            if (b.down >= 0) lv.push_back({b.down, b.a}); else lv.push_back({b.a});
            lv.push_back({b.b});
        }
    } else {
        for (int bi = (int)net.blocks.size() - 1; bi >= 0; --bi) {
            const Block& b = net.blocks[bi];
            if (b.down >= 0) lv.push_back({b.b, b.down}); else lv.push_back({b.b});
            lv.push_back({b.a});
        }
        lv.push_back({0});
    }
    return lv;
}
static bool is_down_li(const tcr_net& net, int li) {
    for (const Block& b : net.blocks) if (b.down == li) return true;
    return false;
}
static LevelPlan level_plan(const tcr_net& net, int backward, int level) {
    const std::vector<std::vector<int>> lv = level_units(net, backward);
    const int n = (int)lv.size();
    LevelPlan p;
    p.first = level == 0; p.last = level == n;
    if (level < n) p.pre = lv[level];
    if (!backward) {
        if (level > 0) p.post = lv[level - 1];
    } else {
        if (level > 0) for (int li : lv[level - 1]) if (!is_down_li(net, li)) p.post.push_back(li);
        if (level > 1) for (int li : lv[level - 2]) if (is_down_li(net, li)) p.post.push_back(li);
    }
    return p;
}
}  // namespace tcr

extern "C" int tcr_net_num_levels(const tcr_net* net, int backward) { return net ? (int)tcr::level_units(*net, backward).size() + 1 : 0; }

// ---- eval-mode forward ------------------------------------------------------------------------
namespace tcr {

static const float* layer_input(const tcr_net& net, const Workspace& w, const float* base, const float* feat, const ConvLayer& l) {
    return l.in_act < 0 ? feat : base + w.act[l.in_act];
}

static int conv_forward(const tcr_net& net, const ConvLayer& l, const float* x, const float* params, float* y,
                        const float* scale, const float* shift, const float* res, bool relu, int epi, int batch, hipStream_t s) {
    (void)net;
    if (l.k == 1 && res == nullptr) {
        Conv1x1Args a;
        a.x = x; a.w = params + l.w_off; a.y = y; a.scale = scale; a.shift = shift;
        a.npos = batch * l.tout; a.cin = l.cin; a.cout = l.cout;
        a.tpi = tcr_padded_len(l.tin); a.tout = l.tout; a.tpo = tcr_padded_len(l.tout);
        a.stride = l.stride; a.relu = relu;
        return launch_conv1x1(a, epi == EPI_AFFINE ? MF_AFFINE : MF_RAW, s);
    }
    ConvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = x; a.w = params + l.w_off; a.y = y; a.scale = scale; a.shift = shift; a.res = res;
    a.npos = batch * l.tout; a.cin = l.cin; a.cout = l.cout;
    a.tpi = tcr_padded_len(l.tin); a.tout = l.tout; a.tpo = tcr_padded_len(l.tout);
    a.xoff = TCR_HALO - l.pad_lo; a.relu = relu;
    if (tune_get(TCR_TUNE_CONV_PATH) != 1) {
        const int rc = launch_conv_mfma(l.k, l.stride, a, epi, s);
        if (rc != 1) return rc;             // launched (or failed); 1 == shape not covered -> VALU kernel
    }
    return launch_conv_fwd(l.k, l.stride, a, epi, s);
}

// conv_a (9x1, stride 2) and the block's 1x1 stride-2 `down` shortcut read the same input rows: one launch,
// one LDS image.  Returns 1 when the fused kernel does not apply (caller launches them separately).
static int conv_forward_with_down(const ConvLayer& la, const ConvLayer& ld, const float* x, const float* params,
                                  float* ya, const float* sca, const float* sha, float* yd, const float* scd, const float* shd,
                                  int epi, int batch, hipStream_t s) {
    if (tune_get(TCR_TUNE_CONV_PATH) == 1 || la.k != 9 || la.stride != 2 || ld.k != 1 || ld.stride != 2) return 1;
    ConvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = x; a.w = params + la.w_off; a.y = ya; a.scale = sca; a.shift = sha; a.res = nullptr;
    a.npos = batch * la.tout; a.cin = la.cin; a.cout = la.cout;
    a.tpi = tcr_padded_len(la.tin); a.tout = la.tout; a.tpo = tcr_padded_len(la.tout);
    a.xoff = TCR_HALO - la.pad_lo; a.relu = la.relu;
    return launch_conv_mfma_with_down(a, params + ld.w_off, yd, scd, shd, la.pad_lo, epi, s);
}

// Whole-network fused launch (fused.hip): three rotating LDS buffers per utterance,
//   R0: features, later the `down` outputs;  R1: conv0 / block outputs;  R2: conv_a outputs.
static int forward_infer_fused(const tcr_net& net, const float* params, const float* ss, const float* feat, int batch,
                               float* logits, float* probs, float* ranges, hipStream_t s) {
    FusedArgs a;
    std::memset(&a, 0, sizeof(a));
    if ((int)net.units.size() > kFusedMaxLayers) return 1;
    int sz[3] = {0, 0, 0};
    auto grow = [&](int b, int c, int t) { const int n = (int)align_up((int64_t)c * tcr_padded_len(t), 4); if (n > sz[b]) sz[b] = n; };
    auto add = [&](int li, int in_buf, int out_buf, int res_buf) {
        const ConvLayer& l = net.layers[li];
        FusedLayer& L = a.layer[a.n_layers++];
        L.k = l.k; L.stride = l.stride; L.cin = l.cin; L.cout = l.cout; L.tin = l.tin; L.tout = l.tout; L.pad_lo = l.pad_lo;
        L.relu = l.relu; L.in_buf = in_buf; L.out_buf = out_buf; L.res_buf = res_buf;
        L.w_off = (int)l.w_off; L.ss_off = (int)l.ss_off; L.c_pad = l.c_pad;
        grow(out_buf, l.cout, l.tout);
        return l.cin % 4 == 0;
    };
    bool ok = true;
    const bool in_global = tune_get(TCR_TUNE_NET_FUSED) != 2;
    if (!in_global) grow(0, net.cfg.in_channels, net.cfg.t_in);
    ok &= add(0, 0, 1, -1);
    for (const Block& b : net.blocks) {
        if (b.down >= 0) {
            ok &= add(b.down, 1, 0, -1);
            a.layer[a.n_layers - 1].no_barrier = 1;
        }
        ok &= add(b.a, 1, 2, -1);
        ok &= add(b.b, 2, 1, b.down >= 0 ? 0 : 1);
    }
    if (!ok || net.param_floats >= (int64_t)1 << 31) return 1;
    // Per-utterance stride of every layer's output rows.  One LDS read of a consumer conv serves 2 input channels x 16 columns
    // (fused.hip: fused_layer_s); within an utterance the 16 columns are 2 floats apart (stride 2: consecutive positions; stride 1:
    // every other position), and with out_sz = 2 * T_out(consumer) (stride 2) or T_in (stride 1) modulo 32 the column -> bank
    // pattern runs on unbroken into the next utterance of the group: 32 lanes, 32 banks.  (<= 31 floats of pad per utterance and
    // buffer; knob TCR_TUNE_NET_FUSED = 7: no padding, the A/B arm.)
    const bool pad_strides = tune_get(TCR_TUNE_NET_FUSED) != 7;
    for (int i = 0; i < a.n_layers; ++i) {
        FusedLayer& P = a.layer[i];
        int need = (int)align_up((int64_t)P.cout * tcr_padded_len(P.tout), 4), want = -1;
        for (int j = i + 1; j < a.n_layers; ++j) {
            const FusedLayer& C = a.layer[j];
            if (C.in_buf == P.out_buf && want < 0) want = C.stride == 2 ? 2 * C.tout : C.stride == 1 ? C.tin : -1;
            if (C.out_buf == P.out_buf) break;              // overwritten: no later reader
        }
        if (pad_strides && want >= 0) while ((need - want) % 32 != 0) ++need;
        if (P.res_buf == P.out_buf)                         // identity shortcut, updated in place: the stride of the rows it adds onto
            for (int j = i - 1; j >= 0; --j) if (a.layer[j].out_buf == P.out_buf) { need = a.layer[j].out_sz; break; }
        P.out_sz = need;
        if (need > sz[P.out_buf]) sz[P.out_buf] = need;
    }
    const int feat_row = a.layer[0].cin * tcr_padded_len(a.layer[0].tin);
    for (int i = 0; i < a.n_layers; ++i) {
        FusedLayer& C = a.layer[i];
        C.in_sz = feat_row; C.res_sz = 0;
        for (int j = i - 1; j >= 0; --j) if (a.layer[j].out_buf == C.in_buf) { C.in_sz = a.layer[j].out_sz; break; }
        if (C.res_buf >= 0) for (int j = i - 1; j >= 0; --j) if (a.layer[j].out_buf == C.res_buf) { C.res_sz = a.layer[j].out_sz; break; }
    }
    a.feat_sz = a.layer[a.n_layers - 1].out_sz;
    const int per_utt = sz[0] + sz[1] + sz[2];
    // Policy (scripts/fused_sweep.py, B = 4096, TCResNet8-1.0 / TCResNet14-1.5 at 49 and 98 frames): the largest group of
    // up to 8 utterances whose activations fit the CU's 160 KB of LDS (more positions per layer = fuller MFMA tiles and
    // more jobs per barrier phase), and as many waves per workgroup as keeps ~16 waves on the CU (the kernel needs
    // ~90 VGPRs: 5 waves / SIMD at most).
    auto lds_of = [&](int g) { return ((size_t)g * per_utt + 64) * sizeof(float); };      // + pad: operand prefetch reads one step past the end
    int group = tune_get(TCR_TUNE_FUSED_GROUP);
    if (group <= 0) {
        group = 8;
        while (group > 1 && lds_of(group) > 160 * 1024) --group;
        while (group > 1 && ceil_div(batch, group) < 512) --group;         // small batches: spread over the 256 CUs first
    }
    if (group > 16) group = 16;
    if (group > batch) group = batch;
    if (group < 1) group = 1;
    const size_t lds = lds_of(group);
    if (lds > 160 * 1024) return 1;
    // the head scratch (pooled + logits) lives in a buffer other than the feature buffer
    if ((int64_t)group * (net.feat_c + net.cfg.num_classes + 2) > (int64_t)group * sz[2] && (int64_t)group * (net.feat_c + net.cfg.num_classes + 2) > (int64_t)group * sz[0]) return 1;
    a.params = params; a.ss = ss; a.feat = feat; a.logits = logits; a.probs = probs; a.ranges = ranges;
    a.batch = batch; a.group = group; a.n_groups = ceil_div(batch, group);
    a.buf_off[0] = 0; a.buf_off[1] = group * sz[0]; a.buf_off[2] = group * (sz[0] + sz[1]);
    a.buf_sz[0] = sz[0]; a.buf_sz[1] = sz[1]; a.buf_sz[2] = sz[2];
    a.in_c = net.cfg.in_channels; a.in_tp = tcr_padded_len(net.cfg.t_in);
    a.in_global = in_global ? 1 : 0;
    a.feat_buf = 1; a.feat_c = net.feat_c; a.feat_t = net.feat_t; a.nc = net.cfg.num_classes;
    a.fc_off = (int)net.layers[net.fc].w_off; a.fc2_off = (int)net.layers[net.fc2].w_off;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    int grid = 256 * per_cu;
    if (grid > a.n_groups) grid = a.n_groups;
    if (tune_get(TCR_TUNE_FUSED_GRID) > 0 && grid > tune_get(TCR_TUNE_FUSED_GRID)) grid = tune_get(TCR_TUNE_FUSED_GRID);
    // knob: waves + 100 * ring (0: default)
    const int knob = tune_get(TCR_TUNE_FUSED_WAVES);
    const int waves = knob % 100 ? knob % 100 : (per_cu >= 4 ? 4 : (per_cu >= 2 ? 8 : 16)), ring = knob / 100 ? knob / 100 : 4;
    return launch_net_fused(a, lds, grid, waves, ring, s);
}

}  // namespace tcr

namespace tcr {
// eval-mode BN of every unit folded to per-channel (scale, shift) at `ss` (layout: unit u at ss_off: scale[c_pad], shift[c_pad])
static int fold_bn(const tcr_net& net, const float* params, const float* stats, float* ss, hipStream_t s) {
    BnFoldArgs f;
    std::memset(&f, 0, sizeof(f));
    f.params = params; f.stats = stats; f.out = ss; f.eps = net.cfg.bn_eps;
    for (int li : net.units) {
        const ConvLayer& l = net.layers[li];
        f.c[f.n] = l.cout; f.c_pad[f.n] = l.c_pad;
        f.gamma_off[f.n] = l.gamma_off; f.beta_off[f.n] = l.beta_off;
        f.mean_off[f.n] = l.mean_off; f.var_off[f.n] = l.var_off; f.out_off[f.n] = l.ss_off; f.bias_off[f.n] = -1;
        ++f.n;
    }
    return launch_bn_fold(f, s);
}
static int64_t ss_floats(const tcr_net& net) {
    int64_t n = 0;
    for (const ConvLayer& l : net.layers) if (l.bn) n += 2 * l.c_pad;
    return n;
}
}  // namespace tcr

static int forward_infer_impl(const tcr_net* net, const float* params, const float* stats, const float* frozen_ss, const float* feat,
                              int batch, void* workspace, size_t workspace_bytes, float* logits, float* probs, float* ranges, void* stream);

extern "C" int tcr_net_forward_infer(const tcr_net* net, const float* params, const float* stats, const float* feat,
                                     int batch, void* workspace, size_t workspace_bytes,
                                     float* logits, float* probs, float* ranges, void* stream) {
    TCR_REQUIRE(stats, "tcr_net_forward_infer: null argument");
    return forward_infer_impl(net, params, stats, nullptr, feat, batch, workspace, workspace_bytes, logits, probs, ranges, stream);
}

extern "C" int64_t tcr_net_frozen_floats(const tcr_net* net) { return net ? ss_floats(*net) : 0; }

extern "C" int tcr_net_fold_bn(const tcr_net* net, const float* params, const float* stats, float* frozen_ss, void* stream) {
    TCR_REQUIRE(net && params && stats && frozen_ss, "tcr_net_fold_bn: null argument");
    return fold_bn(*net, params, stats, frozen_ss, static_cast<hipStream_t>(stream));
}

extern "C" int tcr_net_forward_frozen(const tcr_net* net, const float* params, const float* frozen_ss, const float* feat,
                                      int batch, void* workspace, size_t workspace_bytes,
                                      float* logits, float* probs, float* ranges, void* stream) {
    TCR_REQUIRE(frozen_ss, "tcr_net_forward_frozen: null argument");
    return forward_infer_impl(net, params, nullptr, frozen_ss, feat, batch, workspace, workspace_bytes, logits, probs, ranges, stream);
}

extern "C" int tcr_forward_waveform(const tcr_frontend_cfg* cfg, const void* plan_dev, const tcr_net* net, const float* params,
                                    const float* stats, float* frozen_ss, int refold, const float* wav, int batch, float* feat,
                                    void* workspace, size_t workspace_bytes, float* logits, float* probs, float* ranges, void* stream) {
    TCR_REQUIRE(cfg && net && frozen_ss && feat, "tcr_forward_waveform: null argument");
    TCR_REQUIRE(cfg->n_coef == net->cfg.in_channels && cfg->n_frames == net->cfg.t_in,
                "tcr_forward_waveform: the front-end yields %d x %d features, the network expects %d x %d", cfg->n_coef, cfg->n_frames,
                net->cfg.in_channels, net->cfg.t_in);
    TCR_TRY(tcr_frontend_fwd(cfg, plan_dev, wav, batch, feat, stream));
    if (refold) {
        TCR_REQUIRE(params && stats, "tcr_forward_waveform: refold needs params and stats");
        TCR_TRY(fold_bn(*net, params, stats, frozen_ss, static_cast<hipStream_t>(stream)));
    }
    return forward_infer_impl(net, params, nullptr, frozen_ss, feat, batch, workspace, workspace_bytes, logits, probs, ranges, stream);
}

static int forward_infer_impl(const tcr_net* net, const float* params, const float* stats, const float* frozen_ss, const float* feat,
                              int batch, void* workspace, size_t workspace_bytes, float* logits, float* probs, float* ranges, void* stream) {
    TCR_REQUIRE(net && params && feat && workspace && logits && probs, "tcr_net_forward_infer: null argument");
    TCR_REQUIRE(batch > 0, "tcr_net_forward_infer: batch must be positive (got %d)", batch);
    const Workspace w = carve(*net, batch, false);
    if ((size_t)w.total * sizeof(float) > workspace_bytes) {
        set_error("tcr_net_forward_infer: workspace %zu bytes < required %zu", workspace_bytes, (size_t)w.total * sizeof(float));
        return TCR_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* base = static_cast<float*>(workspace);
    const float* ss = frozen_ss;
    if (!ss) {
        TCR_TRY(fold_bn(*net, params, stats, base + w.ss, s));
        ss = base + w.ss;
    }

    if (tune_get(TCR_TUNE_NET_FUSED) != 1) {
        const int rc = forward_infer_fused(*net, params, ss, feat, batch, logits, probs, ranges, s);
        if (rc != 1) return rc;         // launched (or failed); 1 == does not fit -> per-layer kernels below
    }

    auto run = [&](int li, const float* res) -> int {
        const ConvLayer& l = net->layers[li];
        const float* x = layer_input(*net, w, base, feat, l);
        return conv_forward(*net, l, x, params, base + w.act[li], ss + l.ss_off, ss + l.ss_off + l.c_pad, res, l.relu, EPI_AFFINE, batch, s);
    };
    TCR_TRY(run(0, nullptr));
    int cur = 0;
    for (const Block& b : net->blocks) {
        const float* shortcut = base + w.act[cur];
        bool a_done = false;
        if (b.down >= 0) {
            const ConvLayer& la = net->layers[b.a];
            const ConvLayer& ld = net->layers[b.down];
            const int rc = conv_forward_with_down(la, ld, base + w.act[cur], params, base + w.act[b.a], ss + la.ss_off,
                                                  ss + la.ss_off + la.c_pad, base + w.act[b.down], ss + ld.ss_off,
                                                  ss + ld.ss_off + ld.c_pad, EPI_AFFINE, batch, s);
            if (rc == 1) TCR_TRY(run(b.down, nullptr));
            else { TCR_TRY(rc); a_done = true; }
            shortcut = base + w.act[b.down];
        }
        if (!a_done) TCR_TRY(run(b.a, nullptr));
        TCR_TRY(run(b.b, shortcut));
        cur = b.b;
    }
    HeadArgs h;
    std::memset(&h, 0, sizeof(h));
    h.feat = base + w.act[cur];
    h.wfc = params + net->layers[net->fc].w_off;
    h.wfc2 = params + net->layers[net->fc2].w_off;
    h.logits = logits; h.probs = probs; h.ranges = ranges;
    h.batch = batch; h.c = net->feat_c; h.nc = net->cfg.num_classes; h.t = net->feat_t; h.tp = tcr_padded_len(net->feat_t);
    h.keep_prob = 1.0f; h.inv_global_batch = 1.0f;
    return launch_head_fwd(h, false, s);
}

// ---- train-mode forward -----------------------------------------------------------------------
namespace tcr {

// the library's second stream (process-wide, tcr::shared_stream) + the net's own events, on first use
static int side_stream(const tcr_net& net, hipStream_t caller, hipStream_t* out) {
    if (!net.side) {
        net.side2 = shared_stream(1, caller);
        if (!(net.side = shared_stream(0, caller)) || !net.side2 ||
            hipEventCreateWithFlags(&net.ev_fork, internal_event_flags()) != hipSuccess ||
            hipEventCreateWithFlags(&net.ev_join, internal_event_flags()) != hipSuccess ||
            hipEventCreateWithFlags(&net.ev_down, internal_event_flags()) != hipSuccess ||
            hipEventCreateWithFlags(&net.ev_down_dg, internal_event_flags()) != hipSuccess ||
            hipEventCreateWithFlags(&net.ev_join2, internal_event_flags()) != hipSuccess) {
            net.side = nullptr;
            set_error("cannot create the internal side stream");
            return TCR_ERR_HIP;
        }
    }
    *out = net.side;
    return TCR_OK;
}

struct TrainCtx {
    bool sync_bn;           // cross-replica statistics: partial rows are pre-reduced into `sums` for the host all-reduce
    const tcr_net* net;
    Workspace w;
    float* base;
    const float* params;
    const float* feat;
    int batch;
    double bn_batch;        // batch the BN statistics are taken over (global batch under sync BN)
    hipStream_t side;       // backward: stream of the weight-gradient kernels (== s when overlap is off)
    hipStream_t s;
};

// float64 hand-off slot of unit `li` (cross-replica BN): shortcut units use the second slot, so that a block's `down` and the conv
// sharing its dependency level (conv_a in the forward, conv_b in the backward) can be all-reduced in one call
static bool is_down_unit(const tcr_net& net, int li) {
    for (const Block& b : net.blocks) if (b.down == li) return true;
    return false;
}
static double* sums_of(const TrainCtx& c, int li) {
    return reinterpret_cast<double*>(c.base + c.w.sums + (is_down_unit(*c.net, li) ? c.w.sums_slot : 0));
}

// conv + per-channel sums of the raw output (everything before the cross-replica hand-off)
static bool fused_with_down(const tcr_net& net, int li, int* down_of_a, int* a_of_down) {
    for (const Block& b : net.blocks) {
        if (b.down < 0) continue;
        const ConvLayer& la = net.layers[b.a];
        const ConvLayer& ld = net.layers[b.down];
        const bool ok = tune_get(TCR_TUNE_CONV_PATH) != 1 && la.k == 9 && la.stride == 2 && ld.k == 1 && ld.stride == 2 && la.cin % 4 == 0;
        if (!ok) continue;
        if (li == b.a) { if (down_of_a) *down_of_a = b.down; return true; }
        if (li == b.down) { if (a_of_down) *a_of_down = b.a; return true; }
    }
    return false;
}

static int fwd_unit_pre(const TrainCtx& c, int li, hipStream_t rs, float* partial) {     // rs / partial: stream and scratch of the statistics
    const ConvLayer& l = c.net->layers[li];
    const float* x = layer_input(*c.net, c.w, c.base, c.feat, l);
    float* raw = c.base + c.w.raw[li];
    int partner = -1;
    bool conv_done = false;
    if (fused_with_down(*c.net, li, &partner, nullptr) && partner >= 0) {
        conv_done = true;                   // conv_a: its raw output was produced together with the down conv
    } else if (fused_with_down(*c.net, li, nullptr, &partner) && partner >= 0) {
        const ConvLayer& la = c.net->layers[partner];
        const int rc = conv_forward_with_down(la, l, x, c.params, c.base + c.w.raw[partner], nullptr, nullptr, raw, nullptr, nullptr,
                                              EPI_RAW, c.batch, c.s);
        if (rc != 1) { TCR_TRY(rc); conv_done = true; }
        else {
            // (shape not covered after all) fall through to separate launches for both layers
            TCR_TRY(conv_forward(*c.net, la, x, c.params, c.base + c.w.raw[partner], nullptr, nullptr, nullptr, false, EPI_RAW, c.batch, c.s));
        }
    }
    if (!conv_done) TCR_TRY(conv_forward(*c.net, l, x, c.params, raw, nullptr, nullptr, nullptr, false, EPI_RAW, c.batch, c.s));
    ChanReduceArgs r;
    std::memset(&r, 0, sizeof(r));
    r.y = raw; r.partial = partial;
    r.npos = c.batch * l.tout; r.c = l.cout; r.t = l.tout; r.tp = tcr_padded_len(l.tout);
    if (rs != c.s && (hipEventRecord(c.net->ev_fork, c.s) != hipSuccess || hipStreamWaitEvent(rs, c.net->ev_fork, 0) != hipSuccess)) {
        set_error("tcr_net_forward_train: stream fork failed");     // (the conv above ran on the main stream)
        return TCR_ERR_HIP;
    }
    int nchunk = 0;
    TCR_TRY(launch_chan_reduce(0, r, &nchunk, rs));
    if (!c.sync_bn) return TCR_OK;          // the finalize kernel sums the partial rows itself
    return launch_chan_sums(partial, nchunk, l.cout, sums_of(c, li), rs);
}

// statistics -> scale/shift, moving-stat update, normalise (+ReLU / +residual)
static BnFinalizeArgs fwd_finalize_args(const TrainCtx& c, int li, float* stats, float* partial, int rows) {
    const ConvLayer& l = c.net->layers[li];
    float* ss = c.base + c.w.ss + l.ss_off;
    BnFinalizeArgs f;
    f.partial = partial;
    f.nchunk = c.sync_bn ? 0 : (rows >= 0 ? rows : chan_reduce_launch_chunks(c.batch * l.tout, l.tout));
    f.sums = sums_of(c, li);
    f.gamma = c.params + l.gamma_off; f.beta = c.params + l.beta_off;
    f.moving_mean = stats + l.mean_off; f.moving_var = stats + l.var_off;
    f.scale = ss; f.shift = ss + l.c_pad;
    f.mean = c.base + c.w.mean[li]; f.invstd = c.base + c.w.invstd[li];
    f.c = l.cout; f.count = c.bn_batch * (double)l.tout;
    f.decay = c.net->cfg.bn_decay; f.eps = c.net->cfg.bn_eps;
    return f;
}

static int fwd_unit_post(const TrainCtx& c, int li, float* stats, const float* res, hipStream_t rs, float* partial, int rows = -1) {
    // rows >= 0: the partial rows come from a group-resident phase (train_fused.hip), which also normalises on the fly in its
    // consumer: only the finalize runs here
    const ConvLayer& l = c.net->layers[li];
    float* ss = c.base + c.w.ss + l.ss_off;
    const BnFinalizeArgs f = fwd_finalize_args(c, li, stats, partial, rows);
    TCR_TRY(launch_bn_finalize(f, rs));
    if (rows >= 0) return TCR_OK;
    BnApplyArgs a;
    a.y = c.base + c.w.raw[li]; a.scale = ss; a.shift = ss + l.c_pad; a.res = res; a.out = c.base + c.w.act[li];
    a.total = (int64_t)c.batch * l.cout * tcr_padded_len(l.tout);
    a.c = l.cout; a.t = l.tout; a.tp = tcr_padded_len(l.tout); a.relu = l.relu;
    return launch_bn_apply(a, rs);
}

// residual source of a unit: conv_b of a block adds the block's shortcut
static const float* unit_residual(const TrainCtx& c, int li) {
    int cur = 0;
    for (const Block& b : c.net->blocks) {
        if (b.b == li) return c.base + c.w.act[b.down >= 0 ? b.down : cur];
        cur = b.b;
    }
    return nullptr;
}

// ---- group-resident phases (train_fused.hip) ------------------------------------------------------------------------------
static const float* ss_of(const TrainCtx& c, int li) { return c.base + c.w.ss + c.net->layers[li].ss_off; }

// The activation entering block `bi` (bi == number of blocks: the activation the head reads), built from raw tensors while a
// phase stages it: X_0 = relu(bn(raw conv0)); X_i = relu(bn(raw conv_b) + shortcut) of block i-1 (tc_resnet.py:21,40-41).
static PhaseSrc block_input_src(const TrainCtx& c, int bi) {
    const tcr_net& net = *c.net;
    PhaseSrc s;
    std::memset(&s, 0, sizeof(s));
    if (bi == 0) {
        const ConvLayer& l = net.layers[0];
        s.kind = 1; s.a = c.base + c.w.raw[0]; s.ss_a = ss_of(c, 0); s.c_pad_a = l.c_pad; s.out_x = c.base + c.w.act[0];
        s.c = l.cout; s.t = l.tout;
        return s;
    }
    const Block& pb = net.blocks[bi - 1];
    const ConvLayer& lb = net.layers[pb.b];
    s.kind = 2; s.a = c.base + c.w.raw[pb.b]; s.ss_a = ss_of(c, pb.b); s.c_pad_a = lb.c_pad; s.out_x = c.base + c.w.act[pb.b];
    s.c = lb.cout; s.t = lb.tout;
    if (pb.down >= 0) {
        s.s_kind = 1; s.s = c.base + c.w.raw[pb.down]; s.ss_s = ss_of(c, pb.down); s.c_pad_s = net.layers[pb.down].c_pad;
        s.out_s = c.base + c.w.act[pb.down];
    } else {
        s.s_kind = 0; s.s = c.base + c.w.act[bi - 1 == 0 ? 0 : net.blocks[bi - 2].b];      // identity: the previous block output
    }
    return s;
}

static PhaseLayer phase_layer_of(const TrainCtx& c, int li, float* partial) {
    const ConvLayer& l = c.net->layers[li];
    PhaseLayer p;
    p.k = l.k; p.stride = l.stride; p.cin = l.cin; p.cout = l.cout; p.tin = l.tin; p.tout = l.tout; p.pad_lo = l.pad_lo;
    p.w_off = (int)l.w_off; p.raw = c.base + c.w.raw[li]; p.partial = partial;
    return p;
}

// The phase that computes unit `li` (li < 0: the closing, staging-only phase).  *first: the phase is launched when this unit's
// turn comes (false for conv_a of a block with a `down` shortcut: it ran together with `down`).
static TrainPhaseArgs phase_of_unit(const TrainCtx& c, int li, bool* first) {
    const tcr_net& net = *c.net;
    TrainPhaseArgs a;
    std::memset(&a, 0, sizeof(a));
    a.params = c.params; a.batch = c.batch;
    *first = true;
    float* p1 = c.base + c.w.partial;
    float* p2 = c.base + c.w.partial2;
    if (li < 0) { a.src = block_input_src(c, (int)net.blocks.size()); a.n_layers = 0; return a; }
    if (li == 0) {
        a.src.kind = 0; a.src.a = c.feat; a.src.c = net.cfg.in_channels; a.src.t = net.cfg.t_in;
        a.n_layers = 1; a.layer[0] = phase_layer_of(c, 0, p1);
        return a;
    }
    for (size_t bi = 0; bi < net.blocks.size(); ++bi) {
        const Block& b = net.blocks[bi];
        if (li == b.down || li == b.a) {
            a.src = block_input_src(c, (int)bi);
            if (b.down >= 0) {
                a.n_layers = 2; a.layer[0] = phase_layer_of(c, b.down, p2); a.layer[1] = phase_layer_of(c, b.a, p1);
                *first = li == b.down;
            } else {
                a.n_layers = 1; a.layer[0] = phase_layer_of(c, b.a, p1);
            }
            return a;
        }
        if (li == b.b) {
            const ConvLayer& la = net.layers[b.a];
            a.src.kind = 1; a.src.a = c.base + c.w.raw[b.a]; a.src.ss_a = ss_of(c, b.a); a.src.c_pad_a = la.c_pad;
            a.src.out_x = c.base + c.w.act[b.a]; a.src.c = la.cout; a.src.t = la.tout;
            a.n_layers = 1; a.layer[0] = phase_layer_of(c, b.b, p1);
            return a;
        }
    }
    return a;
}

static bool phases_usable(const TrainCtx& c) {
    if (tune_get(TCR_TUNE_TRAIN_FWD) == 1 || tune_get(TCR_TUNE_CONV_PATH) == 1) return false;
    bool first;
    for (int li : c.net->units)
        if (train_phase_rows(phase_of_unit(c, li, &first)) < 0) return false;
    return train_phase_rows(phase_of_unit(c, -1, &first)) >= 0;
}

}  // namespace tcr

static int forward_train_stages(const tcr_net* net, const float* params, float* stats, const float* feat,
                                const float* labels, int batch, int global_batch, int sync_bn, float keep_prob,
                                uint64_t seed, int64_t sample_offset, float label_smoothing,
                                void* workspace, size_t workspace_bytes, float* logits, float* probs, float* loss_out,
                                int stage_begin, int stage_end, void* stream, const LevelPlan* plan = nullptr) {
    TCR_REQUIRE(net && params && stats && feat && labels && workspace && logits && probs && loss_out, "tcr_net_forward_train: null argument");
    TCR_REQUIRE(batch > 0 && global_batch >= batch, "tcr_net_forward_train: batch %d / global_batch %d", batch, global_batch);
    TCR_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "tcr_net_forward_train: keep_prob %g outside (0, 1]", keep_prob);
    TrainCtx c;
    c.net = net; c.w = carve(*net, batch, true);
    if ((size_t)c.w.total * sizeof(float) > workspace_bytes) {
        set_error("tcr_net_forward_train: workspace %zu bytes < required %zu", workspace_bytes, (size_t)c.w.total * sizeof(float));
        return TCR_ERR_WORKSPACE;
    }
    c.base = static_cast<float*>(workspace); c.params = params; c.feat = feat; c.batch = batch;
    c.bn_batch = sync_bn ? (double)global_batch : (double)batch;
    c.sync_bn = sync_bn != 0;
    c.s = static_cast<hipStream_t>(stream);
    c.side = c.s;
    if (tune_get(TCR_TUNE_WGRAD_STREAM) != 1 && !sync_bn) TCR_TRY(side_stream(*net, c.s, &c.side));
    const int nu = (int)net->units.size();
    TCR_REQUIRE(plan || (stage_begin >= 0 && stage_end <= nu + 1 && stage_begin < stage_end), "tcr_net_forward_train: bad stage range [%d, %d)", stage_begin, stage_end);
    const bool phases = phases_usable(c);
    if (plan) {         // one dependency level (cross-replica BN): finish `post`, then produce the statistics of `pre`
        TCR_REQUIRE(c.sync_bn, "tcr_net_forward_train_level: levels exist for the cross-replica hand-off");
        auto partial_of = [&](int li) { return c.base + (is_down_unit(*net, li) ? c.w.partial2 : c.w.partial); };
        bool first = true;
        for (int li : plan->post) {
            if (phases) TCR_TRY(fwd_unit_post(c, li, stats, nullptr, c.s, partial_of(li), train_phase_rows(phase_of_unit(c, li, &first))));
            else TCR_TRY(fwd_unit_post(c, li, stats, unit_residual(c, li), c.s, partial_of(li)));
        }
        for (int li : plan->pre) {
            if (phases) {
                const TrainPhaseArgs pa = phase_of_unit(c, li, &first);
                int rows = train_phase_rows(pa);
                if (first) TCR_TRY(launch_train_phase(pa, &rows, c.s));
                TCR_TRY(launch_chan_sums(partial_of(li), rows, net->layers[li].cout, sums_of(c, li), c.s));
            } else {
                TCR_TRY(fwd_unit_pre(c, li, c.s, partial_of(li)));
            }
        }
        if (!plan->last) return TCR_OK;
        // (round 6: the closing phase also leaves the block output's sums over time -- in the workspace's dpool slot, which the
        //  backward's first launch rewrites later -- and the head starts from them; TCR_TUNE_TRAIN_FWD = 2: the head walks the rows)
        const bool pooled = phases && tune_get(TCR_TUNE_TRAIN_FWD) != 2;
        if (phases) {
            TrainPhaseArgs pc = phase_of_unit(c, -1, &first);
            if (pooled) pc.pool_sum = c.base + c.w.dpool;
            TCR_TRY(launch_train_phase(pc, nullptr, c.s));       // the head's input + the last shortcut
        }
        HeadArgs h;
        std::memset(&h, 0, sizeof(h));
        if (pooled) h.pool_sum = c.base + c.w.dpool;
        h.feat = c.base + c.w.act[net->units[nu - 1]];
        h.wfc = params + net->layers[net->fc].w_off;
        h.wfc2 = params + net->layers[net->fc2].w_off;
        h.labels = labels; h.logits = logits; h.probs = probs; h.ranges = nullptr;
        h.dropped = c.base + c.w.dropped; h.dscale = c.base + c.w.dscale;
        h.dlogits = c.base + c.w.dlogits; h.loss_utt = c.base + c.w.loss_utt;
        h.batch = batch; h.c = net->feat_c; h.nc = net->cfg.num_classes; h.t = net->feat_t; h.tp = tcr_padded_len(net->feat_t);
        h.keep_prob = keep_prob; h.seed = seed; h.sample_offset = sample_offset;
        h.inv_global_batch = 1.0f / (float)global_batch; h.label_smoothing = label_smoothing;
        TCR_TRY(launch_head_fwd(h, true, c.s));
        return launch_sum_vector(c.base + c.w.loss_utt, batch, loss_out, c.s);
    }
    for (int st = stage_begin; phases && st < stage_end; ++st) {
        // Group-resident phases: conv (+ the block's `down`) with the statistics in its epilogue; the previous unit's BN affine,
        // ReLU and residual are applied while the phase stages its input.  Between phases only the tiny finalize kernels run.
        auto is_down = [&](int li) { for (const Block& b : net->blocks) if (b.down == li) return true; return false; };
        auto partial_of = [&](int li) { return c.base + (is_down(li) ? c.w.partial2 : c.w.partial); };
        bool first = true;
        if (st > 0) {
            const int li = net->units[st - 1];
            const int rows = train_phase_rows(phase_of_unit(c, li, &first));
            // a block's shortcut conv and conv_a come out of ONE phase kernel and nothing runs between their finalizes: one launch for
            // the pair (un-staged runs; bitwise the two launches)
            const bool pair_next = !c.sync_bn && is_down(li) && st + 1 < stage_end;                               // this iteration posts the shortcut unit
            const bool pair_prev = !c.sync_bn && st >= 2 && is_down(net->units[st - 2]) && st - 1 >= stage_begin;    // ... and this one its conv_a
            if (pair_next) {
                // (finalized together with conv_a in the next iteration)
            } else if (pair_prev) {
                const int ld = net->units[st - 2];
                bool f2 = true;
                const int rows_d = train_phase_rows(phase_of_unit(c, ld, &f2));
                TCR_TRY(launch_bn_finalize2(fwd_finalize_args(c, ld, stats, partial_of(ld), rows_d), fwd_finalize_args(c, li, stats, partial_of(li), rows), c.s));
            } else {
                TCR_TRY(fwd_unit_post(c, li, stats, nullptr, c.s, partial_of(li), rows));
            }
        }
        if (st < nu) {
            const int li = net->units[st];
            const TrainPhaseArgs pa = phase_of_unit(c, li, &first);
            int rows = train_phase_rows(pa);
            if (first) TCR_TRY(launch_train_phase(pa, &rows, c.s));
            if (c.sync_bn) TCR_TRY(launch_chan_sums(partial_of(li), rows, net->layers[li].cout, sums_of(c, li), c.s));
        } else {
            const bool pooled = tune_get(TCR_TUNE_TRAIN_FWD) != 2;      // (see the level path above)
            TrainPhaseArgs pc = phase_of_unit(c, -1, &first);
            if (pooled) pc.pool_sum = c.base + c.w.dpool;
            TCR_TRY(launch_train_phase(pc, nullptr, c.s));       // the head's input + the last shortcut
            HeadArgs h;
            std::memset(&h, 0, sizeof(h));
            if (pooled) h.pool_sum = c.base + c.w.dpool;
            h.feat = c.base + c.w.act[net->units[nu - 1]];
            h.wfc = params + net->layers[net->fc].w_off;
            h.wfc2 = params + net->layers[net->fc2].w_off;
            h.labels = labels; h.logits = logits; h.probs = probs; h.ranges = nullptr;
            h.dropped = c.base + c.w.dropped; h.dscale = c.base + c.w.dscale;
            h.dlogits = c.base + c.w.dlogits; h.loss_utt = c.base + c.w.loss_utt;
            h.batch = batch; h.c = net->feat_c; h.nc = net->cfg.num_classes; h.t = net->feat_t; h.tp = tcr_padded_len(net->feat_t);
            h.keep_prob = keep_prob; h.seed = seed; h.sample_offset = sample_offset;
            h.inv_global_batch = 1.0f / (float)global_batch; h.label_smoothing = label_smoothing;
            TCR_TRY(launch_head_fwd(h, true, c.s));
            TCR_TRY(launch_sum_vector(c.base + c.w.loss_utt, batch, loss_out, c.s));
        }
    }
    if (phases) return TCR_OK;
    for (int st = stage_begin; st < stage_end; ++st) {
        // Without cross-replica statistics the BN chain (statistics -> finalize -> normalise) of a block's shortcut conv runs
        // on the side stream next to conv_a's: the two raw outputs come from ONE fused launch and nothing reads the shortcut
        // activation before conv_b's residual add.
        const bool early = !c.sync_bn && c.side != c.s;
        auto is_down = [&](int li) { for (const Block& b : net->blocks) if (b.down == li) return true; return false; };
        auto is_block_out_with_down = [&](int li) { for (const Block& b : net->blocks) if (b.b == li && b.down >= 0) return true; return false; };
        float* partial = c.base + c.w.partial;
        if (st > 0) {
            const int li = net->units[st - 1];
            if (early && is_down(li)) {
                TCR_TRY(fwd_unit_post(c, li, stats, unit_residual(c, li), c.side, c.base + c.w.partial2));
                if (hipEventRecord(net->ev_down, c.side) != hipSuccess) { set_error("tcr_net_forward_train: event record failed"); return TCR_ERR_HIP; }
            } else {
                if (early && is_block_out_with_down(li) && hipStreamWaitEvent(c.s, net->ev_down, 0) != hipSuccess) {
                    set_error("tcr_net_forward_train: stream wait failed");
                    return TCR_ERR_HIP;
                }
                TCR_TRY(fwd_unit_post(c, li, stats, unit_residual(c, li), c.s, partial));
            }
        }
        if (st < nu) {
            const int li = net->units[st];
            if (early && is_down(li)) TCR_TRY(fwd_unit_pre(c, li, c.side, c.base + c.w.partial2));
            else TCR_TRY(fwd_unit_pre(c, li, c.s, partial));
        } else {
            HeadArgs h;
            std::memset(&h, 0, sizeof(h));
            h.feat = c.base + c.w.act[net->units[nu - 1]];
            h.wfc = params + net->layers[net->fc].w_off;
            h.wfc2 = params + net->layers[net->fc2].w_off;
            h.labels = labels; h.logits = logits; h.probs = probs; h.ranges = nullptr;
            h.dropped = c.base + c.w.dropped; h.dscale = c.base + c.w.dscale;
            h.dlogits = c.base + c.w.dlogits; h.loss_utt = c.base + c.w.loss_utt;
            h.batch = batch; h.c = net->feat_c; h.nc = net->cfg.num_classes; h.t = net->feat_t; h.tp = tcr_padded_len(net->feat_t);
            h.keep_prob = keep_prob; h.seed = seed; h.sample_offset = sample_offset;
            h.inv_global_batch = 1.0f / (float)global_batch; h.label_smoothing = label_smoothing;
            TCR_TRY(launch_head_fwd(h, true, c.s));
            TCR_TRY(launch_sum_vector(c.base + c.w.loss_utt, batch, loss_out, c.s));
        }
    }
    return TCR_OK;
}

extern "C" int tcr_net_forward_train(const tcr_net* net, const float* params, float* stats, const float* feat,
                                     const float* labels, int batch, int global_batch, float keep_prob,
                                     uint64_t seed, int64_t sample_offset, float label_smoothing,
                                     void* workspace, size_t workspace_bytes,
                                     float* logits, float* probs, float* loss_out, void* stream) {
    const int nu = net ? (int)net->units.size() : 0;
    return forward_train_stages(net, params, stats, feat, labels, batch, global_batch, 0, keep_prob, seed, sample_offset,
                                label_smoothing, workspace, workspace_bytes, logits, probs, loss_out, 0, nu + 1, stream);
}

extern "C" int tcr_net_forward_train_stage(const tcr_net* net, const float* params, float* stats, const float* feat,
                                           const float* labels, int batch, int global_batch, float keep_prob,
                                           uint64_t seed, int64_t sample_offset, float label_smoothing,
                                           void* workspace, size_t workspace_bytes,
                                           float* logits, float* probs, float* loss_out, int stage, void* stream) {
    return forward_train_stages(net, params, stats, feat, labels, batch, global_batch, 1, keep_prob, seed, sample_offset,
                                label_smoothing, workspace, workspace_bytes, logits, probs, loss_out, stage, stage + 1, stream);
}

extern "C" int tcr_net_forward_train_level(const tcr_net* net, const float* params, float* stats, const float* feat,
                                           const float* labels, int batch, int global_batch, float keep_prob, uint64_t seed,
                                           int64_t sample_offset, float label_smoothing, void* workspace, size_t workspace_bytes,
                                           float* logits, float* probs, float* loss_out, int level, void* stream) {
    TCR_REQUIRE(net && level >= 0 && level < tcr_net_num_levels(net, 0), "tcr_net_forward_train_level: bad level %d", level);
    const LevelPlan plan = level_plan(*net, 0, level);
    return forward_train_stages(net, params, stats, feat, labels, batch, global_batch, 1, keep_prob, seed, sample_offset,
                                label_smoothing, workspace, workspace_bytes, logits, probs, loss_out, 0, 1, stream, &plan);
}

// The float64 sums a level hands over: [2 x C] of its main-chain unit at the start of the region, and -- when the level also holds a
// shortcut unit -- [2 x C_down] in the second slot, one contiguous range (the gap behind the first unit's sums rides along).
extern "C" int tcr_net_level_sums(const tcr_net* net, int backward, int level, void* workspace, int batch, double** sums_dev, int64_t* n_doubles) {
    TCR_REQUIRE(net && workspace && sums_dev && n_doubles, "tcr_net_level_sums: null argument");
    const std::vector<std::vector<int>> lv = level_units(*net, backward);
    TCR_REQUIRE(level >= 0 && level < (int)lv.size(), "tcr_net_level_sums: level %d has no hand-off", level);
    const Workspace w = carve(*net, batch, true);
    *sums_dev = reinterpret_cast<double*>(static_cast<float*>(workspace) + w.sums);
    int64_t n = 0;
    for (int li : lv[level]) {
        const int64_t end = (is_down_unit(*net, li) ? w.sums_slot / 2 : 0) + 2 * (int64_t)net->layers[li].cout;
        n = end > n ? end : n;
    }
    *n_doubles = n;
    return TCR_OK;
}

namespace tcr { static std::vector<int> backward_order(const tcr_net& net); }

extern "C" int tcr_net_stage_sums(const tcr_net* net, int backward, int stage, void* workspace, int batch,
                                  double** sums_dev, int64_t* n_doubles) {
    TCR_REQUIRE(net && workspace && sums_dev && n_doubles, "tcr_net_stage_sums: null argument");
    const int nu = (int)net->units.size();
    TCR_REQUIRE(stage >= 0 && stage < nu, "tcr_net_stage_sums: stage %d has no BN hand-off", stage);
    const Workspace w = carve(*net, batch, true);
    const int li = backward ? backward_order(*net)[stage] : net->units[stage];
    *sums_dev = reinterpret_cast<double*>(static_cast<float*>(workspace) + w.sums + (is_down_unit(*net, li) ? w.sums_slot : 0));
    *n_doubles = 2 * (int64_t)net->layers[li].cout;
    return TCR_OK;
}

// ---- backward ---------------------------------------------------------------------------------
namespace tcr {

#if defined(TCR_NET_WHATIF_ENV)
// TCR_WHATIF_BWD: 1 no chan_reduce, 2 no filter gradients (16: nor the first conv's), 4 no BN-backward apply, 8 no data gradients
static int whatif_bwd() { static const int v = getenv("TCR_WHATIF_BWD") ? atoi(getenv("TCR_WHATIF_BWD")) : 0; return v; }
#endif

struct BwdUnit {
    int li;                 // conv layer
    const float* da;        // gradient wrt the unit's (post-activation) output
    int da_bcast;           // da is [B][C] (pooled gradient broadcast over time)
    const float* m1;        // ReLU masks applied to da (activations read back) ...
    const float* m2;
    const float* self_ss;   // ... and / or the unit's own mask recomputed from its raw output: [fmaf(y, scale, shift) > 0]
    int self_cpad;
};

static BwdUnit bwd_unit_of(const TrainCtx& c, int li, const float* dpool, bool masks_from_raw = true) {
    const tcr_net& net = *c.net;
    BwdUnit u;
    u.li = li; u.da = nullptr; u.da_bcast = 0; u.m1 = nullptr; u.m2 = nullptr; u.self_ss = nullptr; u.self_cpad = 0;
    const int last = net.blocks.back().b;
    const bool recompute = masks_from_raw && tune_get(TCR_TUNE_BWD_MASK) != 1;      // (the group-resident phases read the activations)
    auto own = [&](const float** m) {       // the unit's own activation is relu(bn(raw)): its mask needs no second tensor
        if (!recompute) return;
        *m = nullptr;
        u.self_ss = c.base + c.w.ss + net.layers[li].ss_off; u.self_cpad = net.layers[li].c_pad;
    };
    // gradient wrt the OUTPUT activation of block-output layer `b`
    auto grad_of_block_out = [&](int b, const float** g, int* bc) {
        if (b == last) { *g = dpool; *bc = 1; }
        else { *g = c.base + c.w.gact[b]; *bc = 0; }
    };
    if (li == 0) {          // conv0: da = gradient wrt act[conv0], own ReLU
        u.da = c.base + c.w.gact[0]; u.m1 = c.base + c.w.act[0];
        own(&u.m1);
        return u;
    }
    for (const Block& b : net.blocks) {
        if (li == b.b) {            // dz = dOut * [out > 0]
            grad_of_block_out(b.b, &u.da, &u.da_bcast);
            u.m1 = c.base + c.w.act[b.b];
        } else if (li == b.a) {     // da = dgrad of conv_b, own ReLU
            u.da = c.base + c.w.gact[b.a]; u.m1 = c.base + c.w.act[b.a];
            own(&u.m1);
        } else if (li == b.down) {  // shortcut: dOut * [out > 0] * [down > 0]
            grad_of_block_out(b.b, &u.da, &u.da_bcast);
            u.m1 = c.base + c.w.act[b.b]; u.m2 = c.base + c.w.act[b.down];
            own(&u.m2);
        }
    }
    return u;
}

static int bwd_unit_pre(const TrainCtx& c, const BwdUnit& u, hipStream_t st, float* partial) {
    const ConvLayer& l = c.net->layers[u.li];
    ChanReduceArgs r;
    std::memset(&r, 0, sizeof(r));
    r.y = c.base + c.w.raw[u.li]; r.da = u.da; r.m1 = u.m1; r.m2 = u.m2;
    r.mean = c.base + c.w.mean[u.li]; r.invstd = c.base + c.w.invstd[u.li];
    if (u.self_ss) { r.self_scale = u.self_ss; r.self_shift = u.self_ss + u.self_cpad; }
    r.partial = partial;
    r.npos = c.batch * l.tout; r.c = l.cout; r.t = l.tout; r.tp = tcr_padded_len(l.tout); r.bcast = u.da_bcast;
    int nchunk = 0;
#if defined(TCR_NET_WHATIF_ENV)     // (diagnostic builds: TCR_BUILD_EXTRA=-DTCR_NET_WHATIF_ENV; TCR_WHATIF_BWD bits skip launches -- wrong results, timing only)
    if (whatif_bwd() & 1) return TCR_OK;
#endif
    TCR_TRY(launch_chan_reduce(1, r, &nchunk, st));
    if (!c.sync_bn) return TCR_OK;
    return launch_chan_sums(partial, nchunk, l.cout, sums_of(c, u.li), st);
}

// parts: 1 = BN backward (finalize + apply -> dy), 2 = weight gradient, 4 = data gradient.  bn_stream / partial / kc: where the
// BN part runs (the shortcut branch's may run early on the side stream with the second scratch set).
enum { BWD_BN = 1, BWD_WGRAD = 2, BWD_DGRAD = 4, BWD_ALL = 7 };

// down_first: a block's shortcut conv writes the block-input gradient FIRST (its data gradient on `dg_stream`, early), conv_a's
// adds onto the phases the shortcut wrote -- instead of conv_a first and the shortcut's accumulated behind it on the main chain.
static bool down_dgrad_first(const TrainCtx& c, const Block& b) {
    // Measured (batch 4096, scripts/ab_down_dgrad.py): TCResNet8 -0.8 % at 49 frames, -3 % at 98; TCResNet14-1.5 +1 % (its side stream is
    // the longer one already) -- so by width, like the front-end's submit point: nets of <= 48 channels.  Knob 2 forces it on.
    // Round 6 (filter gradients ~25 % faster: the side stream is no longer the longer one): TCResNet14-1.5 at 98 frames 4027 -> 3936 us
    // with it, at 49 frames 2493 -> 2499 -- so for the wide nets from 64 frames up.
    const int knob = tune_get(TCR_TUNE_DOWN_DGRAD);
    if (b.down < 0 || c.sync_bn || c.side == c.s || knob == 1 || (knob == 0 && c.net->feat_c > 48 && c.net->layers[0].tin < 64)) return false;
    const ConvLayer& ld = c.net->layers[b.down];
    const ConvLayer& la = c.net->layers[b.a];
    return ld.in_act >= 0 && conv_dgrad_mfma_covers(ld.k, ld.stride, ld.cout) && conv_dgrad_mfma_covers(la.k, la.stride, la.cout);
}

static int bwd_unit_post(const TrainCtx& c, const BwdUnit& u, float* grads, const float* dpool, int parts, hipStream_t bn_stream,
                         float* partial, float* kc, hipStream_t dg_stream = nullptr) {
    const tcr_net& net = *c.net;
    const ConvLayer& l = net.layers[u.li];
    const int tp = tcr_padded_len(l.tout), tpi = tcr_padded_len(l.tin);
    const int64_t kstride = align_up(l.cout, 64);
    float* dy = c.base + c.w.dyb[u.li];
    // (see BWD_BN below; needs this call to do both halves, the unit's own mask recomputed from raw, and a kernel that takes WgradFly)
    const bool fly_first = l.in_act < 0 && (parts & BWD_BN) && (parts & BWD_WGRAD) && !u.da_bcast && u.m1 == nullptr && u.m2 == nullptr && u.self_ss != nullptr &&
                           conv_wgrad_deferrable(l.k, l.cin, l.cout) && conv_wgrad_fly_covers(l.k, l.stride, false) && tune_get(TCR_TUNE_WGRAD_LDS) != 2;
#if defined(TCR_NET_WHATIF_ENV)
    if (whatif_bwd() & 4) parts &= ~BWD_BN;
    if (whatif_bwd() & 2) parts &= ~BWD_WGRAD;
    if (whatif_bwd() & 8) parts &= ~BWD_DGRAD;
#endif
    if (parts & BWD_BN) {
        BnBwdFinalizeArgs f;
        f.partial = partial;
        f.nchunk = c.sync_bn ? 0 : chan_reduce_launch_chunks(c.batch * l.tout, l.tout);
        f.sums = sums_of(c, u.li); f.gamma = c.params + l.gamma_off; f.invstd = c.base + c.w.invstd[u.li];
        f.dgamma = grads + l.gamma_off; f.dbeta = grads + l.beta_off;
        f.k1 = kc; f.k2 = kc + kstride; f.k3 = kc + 2 * kstride;
        f.c = l.cout; f.count = c.bn_batch * (double)l.tout;
        f.grad_scale = (float)((double)c.batch / c.bn_batch);
        BnBwdApplyArgs a;
        a.accumulate = 0;
        a.y = c.base + c.w.raw[u.li]; a.da = u.da; a.m1 = u.m1; a.m2 = u.m2; a.mean = c.base + c.w.mean[u.li];
        a.k1 = f.k1; a.k2 = f.k2; a.k3 = f.k3; a.dy = dy;
        a.total = (int64_t)c.batch * l.cout * tp; a.c = l.cout; a.t = l.tout; a.tp = tp; a.bcast = u.da_bcast;
        if (u.self_ss) { a.self_scale = u.self_ss; a.self_shift = u.self_ss + u.self_cpad; }
        if (fly_first) {
            // The FIRST conv's dy feeds nothing but its own filter gradient (no gradient flows into the features) and that kernel
            // is the step's tail, alone on the main stream: dy = k1 (dz - k2 - (raw - mean) k3) is built where the filter gradient
            // loads it (WgradFly, as in the lazy chain) instead of by an apply pass that writes it and a kernel that reads it back
            // -- the finalize alone leaves the coefficients (round 6: -25 us at the end of a TCResNet14-1.5 step).
            TCR_TRY(launch_bn_bwd_finalize(f, bn_stream));
        } else {
        const int rc = launch_bn_bwd_apply_fused(f, a, bn_stream);      // finalize inside the apply pass where that applies
        if (rc != TCR_OK && rc != 1) return rc;
        if (rc == 1) {
            TCR_TRY(launch_bn_bwd_finalize(f, bn_stream));
            TCR_TRY(launch_bn_bwd_apply(a, bn_stream));
        }
        }
    }
    // weight gradient
    const float* x = layer_input(net, c.w, c.base, c.feat, l);
    if (!(parts & BWD_WGRAD)) {
        // (not this call)
    } else if (conv_wgrad_deferrable(l.k, l.cin, l.cout)) {     // slabs summed for all layers at once at the end of backward
        hipStream_t ws = u.li == 0 ? c.s : c.side;      // (the first conv's: the step's last, on the main stream -- see reduce_slabs)
        if (ws == c.side && c.side != c.s && tune_get(TCR_TUNE_WGRAD_STREAM) == 4)      // conv_a units on the second internal stream (as the lazy chain)
            for (const Block& b : net.blocks) if (u.li == b.a) ws = net.side2;
        if (bn_stream != c.s) ws = bn_stream;           // dy was written off the main stream (a shortcut unit): its filter gradient follows it there
        else if (ws != c.s) {                           // fork: the side stream waits for dy, the main stream carries on
            if (hipEventRecord(c.net->ev_fork, c.s) != hipSuccess || hipStreamWaitEvent(c.side, c.net->ev_fork, 0) != hipSuccess) {
                set_error("tcr_net_backward: stream fork failed");
                return TCR_ERR_HIP;
            }
        }
        if (fly_first) {
            WgradFly fly;
            fly.raw = c.base + c.w.raw[u.li]; fly.k1 = kc; fly.k2 = kc + kstride; fly.k3 = kc + 2 * kstride; fly.mean = c.base + c.w.mean[u.li];
            fly.self_scale = u.self_ss; fly.self_shift = u.self_ss + u.self_cpad;
            TCR_TRY(launch_conv_wgrad_partial(l.k, l.stride, l.pad_lo, x, u.da, c.base + c.w.wg[u.li], c.batch, l.cin, l.cout, tpi, l.tout, tp,
                                              nullptr, ws, wgrad_fine(l), l.in_act >= 0, &fly));
        } else
        TCR_TRY(launch_conv_wgrad_partial(l.k, l.stride, l.pad_lo, x, dy, c.base + c.w.wg[u.li], c.batch, l.cin, l.cout, tpi, l.tout, tp,
                                          nullptr, ws, wgrad_fine(l), l.in_act >= 0));
    } else {
        // Wide layers (Cout > 80) reduce their slabs at once through the shared scratch, on the main stream.  When this unit's BN
        // backward ran early on the side stream (a block's shortcut), dy is written THERE: the main stream waits for it first.
        if (bn_stream != c.s && hipStreamWaitEvent(c.s, c.net->ev_down, 0) != hipSuccess) {
            set_error("tcr_net_backward: stream wait failed");
            return TCR_ERR_HIP;
        }
        TCR_TRY(launch_conv_wgrad(l.k, l.stride, l.pad_lo, x, dy, grads + l.w_off, c.base + c.w.wgrad_scratch,
                                  c.batch, l.cin, l.cout, tpi, l.tout, tp, c.s, nullptr, nullptr, l.in_act >= 0));
    }
    if (l.in_act < 0 || !(parts & BWD_DGRAD)) return TCR_OK;        // (no gradient flows into the features)
    // data gradient into gact[in_act]; the shortcut branch of the block adds its contribution in the same pass
    float* wt = c.base + c.w.wtl[u.li];       // filled for every layer by the first backward stage
    float* dx = c.base + c.w.gact[l.in_act];
    const float* add = nullptr;
    const float* add_mask = nullptr;
    int add_bcast = 0;
    unsigned add_phases = ~0u;
    hipStream_t ds = dg_stream ? dg_stream : c.s;
    for (size_t bi = 0; bi < net.blocks.size(); ++bi) {
        const Block& b = net.blocks[bi];
        if (u.li == b.down) {               // conv_a's dgrad ran first and already wrote gact[in] -- or this one writes first
            if (!down_dgrad_first(c, b)) add = dx;
        } else if (u.li == b.a && b.down >= 0) {
            if (down_dgrad_first(c, b)) {   // the shortcut's data gradient is in gact[in] (the phases it has taps for): wait for it, add
                const ConvLayer& ld = net.layers[b.down];
                add = dx; add_phases = conv_dgrad_phases(ld.k, ld.stride, ld.pad_lo, ld.tin);
                if (hipStreamWaitEvent(ds, c.net->ev_down_dg, 0) != hipSuccess) { set_error("tcr_net_backward: stream wait failed"); return TCR_ERR_HIP; }
            }
        } else if (u.li == b.a && b.down < 0) {     // identity shortcut: + dOut * [out > 0]
            if (b.b == net.blocks.back().b) { add = dpool; add_bcast = 1; }
            else add = c.base + c.w.gact[b.b];
            add_mask = c.base + c.w.act[b.b];
        }
    }
    {
        const int rc = launch_conv_dgrad_mfma(l.k, l.stride, l.pad_lo, c.params + l.w_off, wt, dy, dx, add, add_mask, add_bcast,
                                              c.batch, l.cin, l.cout, l.tin, l.tout, ds, true, add_phases);
        if (rc != 1) return rc;
    }
    TCR_REQUIRE(ds == c.s && add_phases == ~0u, "tcr_net_backward: layer %d has no matrix-core data gradient", u.li);
    wt = c.base + c.w.wt;
    TCR_TRY(launch_transpose_weights(c.params + l.w_off, wt, l.k, l.cin, l.cout, c.s));
    DgradArgs d;
    std::memset(&d, 0, sizeof(d));
    d.dy = dy; d.wt = wt; d.dx = dx;
    d.ugrp = ceil_div(l.tin, l.stride); d.ngrp = c.batch * d.ugrp;
    d.cin = l.cin; d.cout = l.cout; d.tin = l.tin; d.tpi = tpi; d.tout = l.tout; d.tpo = tp; d.pad_lo = l.pad_lo;
    d.add = add; d.add_mask = add_mask; d.add_bcast = add_bcast;
    return launch_conv_dgrad(l.k, l.stride, d, c.s);
}

// backward execution order of the BN units: reverse blocks; inside a block conv_b, conv_a, down
static std::vector<int> backward_order(const tcr_net& net) {
    std::vector<int> order;
    for (int bi = (int)net.blocks.size() - 1; bi >= 0; --bi) {
        const Block& b = net.blocks[bi];
        order.push_back(b.b);
        order.push_back(b.a);
        if (b.down >= 0) order.push_back(b.down);
    }
    order.push_back(0);
    return order;
}

// ---- group-resident backward phases (train_fused_bwd.hip) ---------------------------------------------------------------------
// dy of unit `li`, built while a phase stages it (kc: the unit's k1 / k2 / k3 rows left by bn_bwd_finalize)
static BwdSrc bwd_src_of_unit(const TrainCtx& c, int li, const float* dpool, const float* kc) {
    const ConvLayer& l = c.net->layers[li];
    const BwdUnit u = bwd_unit_of(c, li, dpool, false);
    const int64_t ks = align_up(l.cout, 64);
    BwdSrc s;
    std::memset(&s, 0, sizeof(s));
    s.kind = 1; s.da = u.da; s.bcast = u.da_bcast; s.m1 = u.m1; s.m2 = u.m2;
    s.raw = c.base + c.w.raw[li]; s.mean = c.base + c.w.mean[li];
    s.k1 = kc; s.k2 = kc + ks; s.k3 = kc + 2 * ks;
    s.out_dy = c.base + c.w.dyb[li]; s.c = l.cout; s.t = l.tout;
    return s;
}

static BwdLayer bwd_layer_of(const TrainCtx& c, int li, int src) {
    const ConvLayer& l = c.net->layers[li];
    BwdLayer b;
    b.src = src; b.k = l.k; b.stride = l.stride; b.pad_lo = l.pad_lo; b.cin = l.cin; b.tin = l.tin; b.cout = l.cout; b.tout = l.tout;
    b.wt = c.base + c.w.wtl[li];
    return b;
}

static BwdStat bwd_stat_of(const TrainCtx& c, int li, const float* dpool, float* partial) {
    const BwdUnit u = bwd_unit_of(c, li, dpool, false);
    BwdStat t;
    t.on = 1; t.m1 = u.m1; t.m2 = u.m2; t.raw = c.base + c.w.raw[li]; t.mean = c.base + c.w.mean[li]; t.invstd = c.base + c.w.invstd[li];
    t.partial = partial;
    return t;
}

// D1 of block bi: dy of conv_b -> its data gradient = the gradient wrt conv_a's activation; statistics of conv_a's BN backward.
static TrainBwdPhaseArgs bwd_phase_d1(const TrainCtx& c, int bi, const float* dpool) {
    const tcr_net& net = *c.net;
    const Block& b = net.blocks[bi];
    TrainBwdPhaseArgs a;
    std::memset(&a, 0, sizeof(a));
    a.batch = c.batch;
    a.src[0] = bwd_src_of_unit(c, b.b, dpool, c.base + c.w.kcoef);
    a.n_layers = 1; a.layer[0] = bwd_layer_of(c, b.b, 0);
    a.out_c = net.layers[b.b].cin; a.out_t = net.layers[b.b].tin; a.out_dx = c.base + c.w.gact[b.a];
    a.stat[0] = bwd_stat_of(c, b.a, dpool, c.base + c.w.partial);
    return a;
}

// D2 of block bi: dy of conv_a (and of the `down` shortcut) -> their data gradients summed (+ the identity shortcut's gradient) =
// the gradient wrt the block input; statistics of the BN backward(s) of the units that produced that input.
static TrainBwdPhaseArgs bwd_phase_d2(const TrainCtx& c, int bi, const float* dpool) {
    const tcr_net& net = *c.net;
    const Block& b = net.blocks[bi];
    const ConvLayer& la = net.layers[b.a];
    TrainBwdPhaseArgs a;
    std::memset(&a, 0, sizeof(a));
    a.batch = c.batch;
    a.src[0] = bwd_src_of_unit(c, b.a, dpool, c.base + c.w.kcoef);
    a.n_layers = 1; a.layer[0] = bwd_layer_of(c, b.a, 0);
    if (b.down >= 0) {
        a.src[1] = bwd_src_of_unit(c, b.down, dpool, c.base + c.w.kcoef2);
        a.n_layers = 2; a.layer[1] = bwd_layer_of(c, b.down, 1);
    } else {                    // identity shortcut: + dOut [out > 0]
        const BwdUnit ub = bwd_unit_of(c, b.b, dpool, false);
        a.add = ub.da; a.add_bcast = ub.da_bcast; a.add_mask = c.base + c.w.act[b.b];
    }
    const int in_act = la.in_act;               // conv0 (bi == 0) or the previous block's conv_b
    a.out_c = la.cin; a.out_t = la.tin; a.out_dx = c.base + c.w.gact[in_act];
    if (bi == 0) {
        a.stat[0] = bwd_stat_of(c, 0, dpool, c.base + c.w.partial);
    } else {
        const Block& pb = net.blocks[bi - 1];
        a.stat[0] = bwd_stat_of(c, pb.b, dpool, c.base + c.w.partial);
        if (pb.down >= 0) a.stat[1] = bwd_stat_of(c, pb.down, dpool, c.base + c.w.partial2);
    }
    return a;
}

// closing phase: dy of conv0 (its weight gradient is all that is left; no gradient flows into the features)
static TrainBwdPhaseArgs bwd_phase_d0(const TrainCtx& c, const float* dpool) {
    TrainBwdPhaseArgs a;
    std::memset(&a, 0, sizeof(a));
    a.batch = c.batch;
    a.src[0] = bwd_src_of_unit(c, 0, dpool, c.base + c.w.kcoef);
    a.n_layers = 0;
    return a;
}

static bool bwd_phases_usable(const TrainCtx& c, const float* dpool) {
    if (tune_get(TCR_TUNE_TRAIN_BWD) != 1 || tune_get(TCR_TUNE_CONV_PATH) == 1) return false;      // (opt-in: measured slower)
    const tcr_net& net = *c.net;
    for (size_t bi = 0; bi < net.blocks.size(); ++bi) {
        for (int li : {net.blocks[bi].down, net.blocks[bi].a, net.blocks[bi].b})
            if (li >= 0 && !conv_dgrad_mfma_covers(net.layers[li].k, net.layers[li].stride, net.layers[li].cout)) return false;
        if (train_bwd_phase_rows(bwd_phase_d1(c, (int)bi, dpool)) < 0 || train_bwd_phase_rows(bwd_phase_d2(c, (int)bi, dpool)) < 0) return false;
    }
    return train_bwd_phase_rows(bwd_phase_d0(c, dpool)) >= 0;
}

// partial rows the statistics of unit `li` arrive in, and their count
static float* bwd_partial_of(const TrainCtx& c, int li) {
    for (const Block& b : c.net->blocks) if (b.down == li) return c.base + c.w.partial2;
    return c.base + c.w.partial;
}
static int bwd_rows_of(const TrainCtx& c, int li, const float* dpool) {
    const tcr_net& net = *c.net;
    const int nb = (int)net.blocks.size();
    if (li == net.blocks[nb - 1].b || li == net.blocks[nb - 1].down) return chan_reduce_launch_chunks(c.batch * net.layers[li].tout, net.layers[li].tout);
    if (li == 0) return train_bwd_phase_rows(bwd_phase_d2(c, 0, dpool));
    for (int bi = 0; bi < nb; ++bi) {
        if (li == net.blocks[bi].a) return train_bwd_phase_rows(bwd_phase_d1(c, bi, dpool));
        if (li == net.blocks[bi].b || li == net.blocks[bi].down) return train_bwd_phase_rows(bwd_phase_d2(c, bi + 1, dpool));
    }
    return -1;
}

// BN backward finalize of unit `li` from `rows` partial rows (or the cross-replica sums): dgamma / dbeta + the k1 / k2 / k3 rows
static int bwd_finalize(const TrainCtx& c, int li, float* grads, float* partial, float* kc, int rows) {
    const ConvLayer& l = c.net->layers[li];
    const int64_t kstride = align_up(l.cout, 64);
    BnBwdFinalizeArgs f;
    f.partial = partial;
    f.nchunk = c.sync_bn ? 0 : rows;
    f.sums = sums_of(c, li); f.gamma = c.params + l.gamma_off; f.invstd = c.base + c.w.invstd[li];
    f.dgamma = grads + l.gamma_off; f.dbeta = grads + l.beta_off;
    f.k1 = kc; f.k2 = kc + kstride; f.k3 = kc + 2 * kstride;
    f.c = l.cout; f.count = c.bn_batch * (double)l.tout;
    f.grad_scale = (float)((double)c.batch / c.bn_batch);
    return launch_bn_bwd_finalize(f, c.s);
}

}  // namespace tcr

// ---- lazy BN backward (bwd_lazy.hip): no reduce / bn_bwd_apply passes, dy never written ----------------------------------------
namespace tcr {

static int block_index_of(const tcr_net& net, int li) {
    for (size_t bi = 0; bi < net.blocks.size(); ++bi) {
        const Block& b = net.blocks[bi];
        if (li == b.b || li == b.a || li == b.down) return (int)bi;
    }
    return -1;
}

// The gradient tensor unit `li` consumes (gz: every mask but, for a shortcut unit, its own already applied by the kernel that wrote
// it) and the coefficients its finalize leaves.
static LazySrc lazy_src_of(const TrainCtx& c, int li) {
    const tcr_net& net = *c.net;
    const ConvLayer& l = net.layers[li];
    const int64_t ks = align_up(l.cout, 64);
    float* kc = c.base + c.w.kc[li];
    LazySrc s;
    std::memset(&s, 0, sizeof(s));
    s.c = l.cout; s.t = l.tout;
    s.raw = c.base + c.w.raw[li]; s.mean = c.base + c.w.mean[li];
    s.k1 = kc; s.k2 = kc + ks; s.k3 = kc + 2 * ks; s.tab = kc + 3 * ks;
    s.gz = c.base + c.w.gact[li];                           // conv0, conv_a, conv_b: the gradient wrt their own activation
    const int bi = block_index_of(net, li);
    if (bi >= 0 && net.blocks[bi].down == li) {             // the shortcut shares the block-output gradient and still applies its own ReLU
        s.gz = c.base + c.w.gact[net.blocks[bi].b];
        s.self_scale = ss_of(c, li); s.self_shift = ss_of(c, li) + l.c_pad;
    }
    return s;
}

static LazyStat lazy_stat_of(const TrainCtx& c, int li, bool self) {
    const ConvLayer& l = c.net->layers[li];
    LazyStat t;
    std::memset(&t, 0, sizeof(t));
    t.on = 1; t.raw = c.base + c.w.raw[li]; t.mean = c.base + c.w.mean[li]; t.invstd = c.base + c.w.invstd[li];
    if (self) { t.self_scale = ss_of(c, li); t.self_shift = ss_of(c, li) + l.c_pad; }
    t.partial = c.base + (is_down_unit(*c.net, li) ? c.w.partial2 : c.w.partial);
    return t;
}

static LazyLayer lazy_layer_of(const TrainCtx& c, int li, int src) {
    const ConvLayer& l = c.net->layers[li];
    LazyLayer L;
    L.src = src; L.k = l.k; L.stride = l.stride; L.pad_lo = l.pad_lo; L.wt = c.base + c.w.wtl[li];
    return L;
}

// conv_b of block bi: writes the gradient wrt conv_a's activation with conv_a's ReLU applied, leaves conv_a's sums
static BwdLazyArgs lazy_args_b(const TrainCtx& c, int bi) {
    const tcr_net& net = *c.net;
    const Block& b = net.blocks[bi];
    const ConvLayer& lb = net.layers[b.b];
    BwdLazyArgs a;
    std::memset(&a, 0, sizeof(a));
    a.batch = c.batch; a.n_layers = 1;
    a.src[0] = lazy_src_of(c, b.b); a.layer[0] = lazy_layer_of(c, b.b, 0);
    a.out_c = lb.cin; a.out_t = lb.tin; a.out_g = c.base + c.w.gact[b.a];
    a.store_self = 1; a.stat[0] = lazy_stat_of(c, b.a, true);
    return a;
}

// conv_a (+ the shortcut conv) of block bi: writes the gradient wrt the block input -- the previous block's output, masked with
// its ReLU, with the sums of that block's conv_b and shortcut unit; or, for the first block, wrt conv0's activation
static BwdLazyArgs lazy_args_a(const TrainCtx& c, int bi) {
    const tcr_net& net = *c.net;
    const Block& b = net.blocks[bi];
    const ConvLayer& la = net.layers[b.a];
    BwdLazyArgs a;
    std::memset(&a, 0, sizeof(a));
    a.batch = c.batch; a.n_layers = 1;
    a.src[0] = lazy_src_of(c, b.a); a.layer[0] = lazy_layer_of(c, b.a, 0);
    if (b.down >= 0) { a.n_layers = 2; a.src[1] = lazy_src_of(c, b.down); a.layer[1] = lazy_layer_of(c, b.down, 1); }
    else a.add = c.base + c.w.gact[b.b];                    // identity shortcut: + the (masked) block-output gradient
    a.out_c = la.cin; a.out_t = la.tin; a.out_g = c.base + c.w.gact[la.in_act];
    if (bi == 0) {
        a.store_self = 1; a.stat[0] = lazy_stat_of(c, 0, true);
    } else {
        const Block& pb = net.blocks[bi - 1];
        a.mask_act = c.base + c.w.act[pb.b];
        a.stat[0] = lazy_stat_of(c, pb.b, false);
        if (pb.down >= 0) a.stat[1] = lazy_stat_of(c, pb.down, true);
    }
    return a;
}

static bool lazy_usable(const TrainCtx& c) {
    const tcr_net& net = *c.net;
    const int knob = tune_get(TCR_TUNE_TRAIN_BWD);
    if ((knob != 0 && knob != 3) || tune_get(TCR_TUNE_CONV_PATH) == 1) return false;
    // Measured (batch 4096, scripts/ab_knob_train.py KNOB=9): TCResNet8-1.0 -7 % per step at 49 frames, -4 % at 98; TCResNet14-1.5
    // (36 / 48 / 72 channels: three to five row tiles per data-gradient job, 150-215 VGPRs) +14 % -- so by width, like the other
    // backward policies: nets of <= 48 channels.  Knob 3 forces it for every net that fits.
    if (knob == 0 && net.feat_c > 48) return false;
    for (int li : net.units) {
        const ConvLayer& l = net.layers[li];
        if (!conv_wgrad_deferrable(l.k, l.cin, l.cout) || !conv_wgrad_fly_covers(l.k, l.stride, l.in_act >= 0)) return false;
        if (l.in_act >= 0 && !conv_dgrad_mfma_covers(l.k, l.stride, l.cout)) return false;
    }
    if ((int)net.units.size() > kMultiMax) return false;
    for (size_t bi = 0; bi < net.blocks.size(); ++bi) {
        const Block& b = net.blocks[bi];
        if (b.down >= 0 && (net.layers[b.down].stride != net.layers[b.a].stride || net.layers[b.down].k != 1)) return false;
        if (bwd_lazy_rows(lazy_args_b(c, (int)bi)) < 0 || bwd_lazy_rows(lazy_args_a(c, (int)bi)) < 0) return false;
    }
    return true;
}

// partial rows holding the backward sums of unit li (who wrote them: the last block's units a reduction pass, all others the
// epilogue of the data-gradient kernel that produced their gradient)
static int lazy_rows_of(const TrainCtx& c, int li) {
    const tcr_net& net = *c.net;
    const int nb = (int)net.blocks.size();
    if (li == 0) return bwd_lazy_rows(lazy_args_a(c, 0));
    const int bi = block_index_of(net, li);
    const Block& b = net.blocks[bi];
    if (li == b.a) return bwd_lazy_rows(lazy_args_b(c, bi));
    if (bi == nb - 1) return chan_reduce_launch_chunks(c.batch * net.layers[li].tout, net.layers[li].tout);
    return bwd_lazy_rows(lazy_args_a(c, bi + 1));
}

static BnBwdFinalizeArgs lazy_finalize_args(const TrainCtx& c, int li, float* grads) {
    const ConvLayer& l = c.net->layers[li];
    const int64_t ks = align_up(l.cout, 64);
    float* kc = c.base + c.w.kc[li];
    BnBwdFinalizeArgs f;
    f.partial = c.base + (is_down_unit(*c.net, li) ? c.w.partial2 : c.w.partial);
    f.nchunk = c.sync_bn ? 0 : lazy_rows_of(c, li);
    f.sums = sums_of(c, li); f.gamma = c.params + l.gamma_off; f.invstd = c.base + c.w.invstd[li];
    f.dgamma = grads + l.gamma_off; f.dbeta = grads + l.beta_off;
    f.k1 = kc; f.k2 = kc + ks; f.k3 = kc + 2 * ks;
    f.c = l.cout; f.count = c.bn_batch * (double)l.tout;
    f.grad_scale = (float)((double)c.batch / c.bn_batch);
    const LazySrc src = lazy_src_of(c, li);
    f.tab = kc + 3 * ks; f.mean = src.mean; f.self_scale = src.self_scale; f.self_shift = src.self_shift;
    return f;
}

// filter gradient of unit li on stream `ws`, dy computed where it is loaded
static int lazy_wgrad(const TrainCtx& c, int li, hipStream_t ws) {
#if defined(TCR_NET_WHATIF) && (TCR_NET_WHATIF & 1)      // timing what-if (scripts/build_whatif_src.sh): no filter-gradient launches; wrong results
    if (li != 0 || (TCR_NET_WHATIF & 2)) return TCR_OK;
#endif
#if defined(TCR_NET_WHATIF_ENV)
    if ((whatif_bwd() & 2) && (li != 0 || (whatif_bwd() & 16))) return TCR_OK;
#endif
    const tcr_net& net = *c.net;
    const ConvLayer& l = net.layers[li];
    const LazySrc src = lazy_src_of(c, li);
    WgradFly fly;
    fly.raw = src.raw; fly.k1 = src.k1; fly.k2 = src.k2; fly.k3 = src.k3; fly.mean = src.mean;
    fly.self_scale = src.self_scale; fly.self_shift = src.self_shift;
    const float* x = layer_input(net, c.w, c.base, c.feat, l);
    return launch_conv_wgrad_partial(l.k, l.stride, l.pad_lo, x, src.gz, c.base + c.w.wg[li], c.batch, l.cin, l.cout, tcr_padded_len(l.tin), l.tout,
                                     tcr_padded_len(l.tout), nullptr, ws, wgrad_fine(l), l.in_act >= 0, &fly);
}

// Sums of the last block's units.  conv_b's reduction reads the head's pooled gradient (broadcast over time) under the block's ReLU
// mask and WRITES that masked tensor on the way (the gz every later reader takes); the shortcut's reads it back with its own mask.
// with_down (round 6): both units in ONE pass (chan_reduce_kernel<2>: the shortcut's sums from the dz in registers instead of from a
// second launch that reads it back) -- bitwise the two launches' rows, one launch less at the head of the backward's chain.
static int lazy_last_block_sums(const TrainCtx& c, int li, hipStream_t st, bool with_down = false) {
    const tcr_net& net = *c.net;
    const ConvLayer& l = net.layers[li];
    const Block& lastb = net.blocks.back();
    const LazySrc src = lazy_src_of(c, li);
    ChanReduceArgs r;
    std::memset(&r, 0, sizeof(r));
    r.y = src.raw; r.mean = src.mean; r.invstd = c.base + c.w.invstd[li];
    if (li == lastb.b) {
        r.da = c.base + c.w.dpool; r.bcast = 1; r.m1 = c.base + c.w.act[lastb.b]; r.g_out = c.base + c.w.gact[lastb.b];
    } else {
        r.da = src.gz; r.self_scale = src.self_scale; r.self_shift = src.self_shift;
    }
    r.partial = c.base + (is_down_unit(net, li) ? c.w.partial2 : c.w.partial);
    r.npos = c.batch * l.tout; r.c = l.cout; r.t = l.tout; r.tp = tcr_padded_len(l.tout);
    int nchunk = 0;
    if (with_down) {
        TCR_REQUIRE(li == lastb.b && lastb.down >= 0, "tcr_net_backward: the two-unit reduction is the last block's");
        const LazySrc sd = lazy_src_of(c, lastb.down);
        r.y2 = sd.raw; r.mean2 = sd.mean; r.invstd2 = c.base + c.w.invstd[lastb.down];
        r.self_scale2 = sd.self_scale; r.self_shift2 = sd.self_shift; r.partial2 = c.base + c.w.partial2;
        return launch_chan_reduce(2, r, &nchunk, st);
    }
    return launch_chan_reduce(1, r, &nchunk, st);
}

static int backward_lazy(const TrainCtx& c, float* grads, const LevelPlan* plan) {
    const tcr_net* net = c.net;
    const int nb = (int)net->blocks.size();
    const Block& lastb = net->blocks[nb - 1];
    const bool multi = !plan && c.side != c.s;              // (a dependency level of the cross-replica hand-off stays on the caller's stream)
    const hipStream_t s_b = multi ? c.side : c.s;           // filter gradients of conv_b + shortcut units
    const hipStream_t s_a = multi ? net->side2 : c.s;       // filter gradients of conv_a units; first the classifier's
    auto fork = [&](hipStream_t to) -> int {
        if (to == c.s) return TCR_OK;
        if (hipEventRecord(net->ev_fork, c.s) != hipSuccess || hipStreamWaitEvent(to, net->ev_fork, 0) != hipSuccess) {
            set_error("tcr_net_backward: stream fork failed");
            return TCR_ERR_HIP;
        }
        return TCR_OK;
    };
    auto launch_d = [&](const BwdLazyArgs& a) -> int {
#if defined(TCR_NET_WHATIF_ENV)
        if (whatif_bwd() & 8) return TCR_OK;
#endif
        const int rc = launch_bwd_lazy(a, nullptr, c.s);
        if (rc == 1) { set_error("tcr_net_backward: lazy data gradient does not cover a layer it was planned for"); return TCR_ERR_ARG; }
        return rc;
    };
    if (!plan || plan->first) {
        // ONE launch: head backward (pooled gradient + the last block's masked output gradient), arena zero fill, re-arranged weights
        const int nc = net->cfg.num_classes;
        DgradWeightsMulti dm;
        dm.n = 0;
        for (int li : net->units) {
            const ConvLayer& l = net->layers[li];
            if (l.in_act < 0) continue;
            dm.e[dm.n++] = {c.params + l.w_off, c.base + c.w.wtl[li], l.k, l.cin, l.cout, l.stride, l.pad_lo};
        }
        BwdPrologueArgs hp;
        hp.dlogits = c.base + c.w.dlogits; hp.wfc = c.params + net->layers[net->fc].w_off; hp.dscale = c.base + c.w.dscale;
        hp.dpool = c.base + c.w.dpool; hp.batch = c.batch; hp.c = net->feat_c; hp.nc = nc; hp.zero = grads; hp.zero_n = net->param_floats;
        TCR_TRY(launch_bwd_prologue(hp, dm, c.s));
        TCR_TRY(fork(s_a));
        TCR_TRY(launch_fc_wgrad(c.base + c.w.dropped, c.base + c.w.dlogits, c.base + c.w.fc_partial,
                                grads + net->layers[net->fc].w_off, c.batch, net->feat_c, nc, s_a));
    }
    // everything downstream of unit li's sums
    auto finish_unit = [&](int li, bool finalize) -> int {
        if (finalize) TCR_TRY(launch_bn_bwd_finalize(lazy_finalize_args(c, li, grads), c.s));
        if (li == 0) return lazy_wgrad(c, 0, c.s);          // the step's last filter gradient: the main stream is idle by then
        const int bi = block_index_of(*net, li);
        const Block& b = net->blocks[bi];
        const hipStream_t ws = li == b.a ? s_a : s_b;
        TCR_TRY(fork(ws));
        TCR_TRY(lazy_wgrad(c, li, ws));
        if (li == b.b) return launch_d(lazy_args_b(c, bi));
        if (li == b.a) return launch_d(lazy_args_a(c, bi));
        return TCR_OK;
    };
    if (plan) {
        TCR_REQUIRE(c.sync_bn, "tcr_net_backward_level: levels exist for the cross-replica hand-off");
        std::vector<int> post;
        for (int li : plan->post) if (is_down_unit(*net, li)) post.push_back(li);       // shortcut units first: conv_a's kernel needs both
        for (int li : plan->post) if (!is_down_unit(*net, li)) post.push_back(li);
        for (int li : post) TCR_TRY(finish_unit(li, true));
        for (int li : plan->pre) {
            if (li == lastb.b || li == lastb.down) TCR_TRY(lazy_last_block_sums(c, li, c.s));
            TCR_TRY(launch_chan_sums(c.base + (is_down_unit(*net, li) ? c.w.partial2 : c.w.partial), lazy_rows_of(c, li), net->layers[li].cout,
                                     sums_of(c, li), c.s));
        }
    } else {
        const ConvLayer& lb = net->layers[lastb.b];
        const bool both = lastb.down >= 0 && net->layers[lastb.down].cout == lb.cout && net->layers[lastb.down].tout == lb.tout && tune_get(TCR_TUNE_BWD_MASK) != 5;
        TCR_TRY(lazy_last_block_sums(c, lastb.b, c.s, both));
        if (lastb.down >= 0 && !both) TCR_TRY(lazy_last_block_sums(c, lastb.down, c.s));
        // A fork (event record on the main chain) per BN unit.  An event record between two kernels costs ~6 us of dispatch gap (the
        // forward, which records none, shows none), so ONE fork per block was tried -- the fork behind (conv_b, shortcut)'s finalize also
        // releasing the previous block's conv_a filter gradient: TCResNet8 975 vs 949 us per step, 98 frames 1492 vs 1443 -- SLOWER: the
        // filter gradients that start ~70 us later end later, and the step is bound by the total work of all streams, not by the main
        // chain's gaps.  Kept behind TCR_TUNE_WGRAD_STREAM = 3.
        const bool fork_per_unit = tune_get(TCR_TUNE_WGRAD_STREAM) != 3 || !multi;
        int pending_a = -1;                                 // conv_a whose filter gradient waits for the next fork
        for (int bi = nb - 1; bi >= 0; --bi) {
            const Block& b = net->blocks[bi];
            if (b.down >= 0) TCR_TRY(launch_bn_bwd_finalize2(lazy_finalize_args(c, b.b, grads), lazy_finalize_args(c, b.down, grads), c.s));
            else TCR_TRY(launch_bn_bwd_finalize(lazy_finalize_args(c, b.b, grads), c.s));
            // the shortcut's filter gradient first: conv_b's is the longer one and nothing waits for either  (A/B: the shortcut's on
            // conv_a's stream instead: 995 vs 950 us per TCResNet8 step, 1517 vs 1448 at 98 frames -- it delays conv_a's)
            TCR_TRY(fork(s_b));
            if (b.down >= 0) TCR_TRY(lazy_wgrad(c, b.down, s_b));
            TCR_TRY(lazy_wgrad(c, b.b, s_b));
            if (pending_a >= 0) {                           // (ordered behind the same record: s_a waits for the event s_b waits for)
                if (s_a != s_b && hipStreamWaitEvent(s_a, net->ev_fork, 0) != hipSuccess) { set_error("tcr_net_backward: stream fork failed"); return TCR_ERR_HIP; }
                TCR_TRY(lazy_wgrad(c, pending_a, s_a));
                pending_a = -1;
            }
            TCR_TRY(launch_d(lazy_args_b(c, bi)));
            TCR_TRY(launch_bn_bwd_finalize(lazy_finalize_args(c, b.a, grads), c.s));
            if (fork_per_unit || bi == 0) { TCR_TRY(fork(s_a)); TCR_TRY(lazy_wgrad(c, b.a, s_a)); }
            else pending_a = b.a;
            TCR_TRY(launch_d(lazy_args_a(c, bi)));
        }
        TCR_TRY(finish_unit(0, true));
    }
    if (plan && !plan->last) return TCR_OK;
    // every layer's split-K slabs -> dW, summed where their producers ran; then the joins
    auto run = [&](int which, hipStream_t st) -> int {      // 0: every layer but the first conv, 1: the first conv, 2: all
        WgradReduceMulti rm;
        rm.n = 0;
        for (int li : net->units) {
            const ConvLayer& l = net->layers[li];
            if ((which == 0 && li == 0) || (which == 1 && li != 0)) continue;
            if (rm.n == kMultiMax) { TCR_TRY(launch_wgrad_reduce_multi(rm, st)); rm.n = 0; }
            rm.e[rm.n++] = conv_wgrad_entry(l.k, l.cin, l.cout, c.batch, c.base + c.w.wg[li], grads + l.w_off, wgrad_fine(l));
        }
        return rm.n ? launch_wgrad_reduce_multi(rm, st) : TCR_OK;
    };
    if (!multi) return run(2, c.s);
    if (hipEventRecord(net->ev_join2, s_a) != hipSuccess || hipStreamWaitEvent(s_b, net->ev_join2, 0) != hipSuccess) {
        set_error("tcr_net_backward: stream join failed");
        return TCR_ERR_HIP;
    }
    TCR_TRY(run(0, s_b));
    TCR_TRY(run(1, c.s));
    if (hipEventRecord(net->ev_join, s_b) != hipSuccess || hipStreamWaitEvent(c.s, net->ev_join, 0) != hipSuccess) {
        set_error("tcr_net_backward: stream join failed");
        return TCR_ERR_HIP;
    }
    return TCR_OK;
}

}  // namespace tcr

static int backward_stages(const tcr_net* net, const float* params, const float* feat, int batch, int global_batch, int sync_bn,
                           void* workspace, size_t workspace_bytes, float* grads, int stage_begin, int stage_end, void* stream,
                           const LevelPlan* plan = nullptr) {
    TCR_REQUIRE(net && params && feat && workspace && grads, "tcr_net_backward: null argument");
    TCR_REQUIRE(batch > 0 && global_batch >= batch, "tcr_net_backward: batch %d / global_batch %d", batch, global_batch);
    TrainCtx c;
    c.net = net; c.w = carve(*net, batch, true);
    if ((size_t)c.w.total * sizeof(float) > workspace_bytes) {
        set_error("tcr_net_backward: workspace %zu bytes < required %zu", workspace_bytes, (size_t)c.w.total * sizeof(float));
        return TCR_ERR_WORKSPACE;
    }
    c.base = static_cast<float*>(workspace); c.params = params; c.feat = feat; c.batch = batch;
    c.bn_batch = sync_bn ? (double)global_batch : (double)batch;
    c.sync_bn = sync_bn != 0;
    c.s = static_cast<hipStream_t>(stream);
    c.side = c.s;
    if (tune_get(TCR_TUNE_WGRAD_STREAM) != 1) TCR_TRY(side_stream(*net, c.s, &c.side));
    const std::vector<int> order = backward_order(*net);
    const int nu = (int)order.size();
    TCR_REQUIRE(plan || (stage_begin >= 0 && stage_end <= nu + 1 && stage_begin < stage_end), "tcr_net_backward: bad stage range [%d, %d)", stage_begin, stage_end);
    // lazy BN backward: whole backward passes and the dependency levels of the cross-replica hand-off (the per-unit stage API keeps the per-layer chain)
    if ((plan || (stage_begin == 0 && stage_end == nu + 1)) && lazy_usable(c)) return backward_lazy(c, grads, plan);
    const float* dpool = c.base + c.w.dpool;
    // where a block's shortcut unit runs (BN backward, data gradient, filter gradient): behind the other units' filter gradients on the
    // side stream, or on the second internal stream (TCR_TUNE_WGRAD_STREAM = 2), where it never queues behind a 50 us filter gradient
    const bool down_on_second = c.side != c.s && tune_get(TCR_TUNE_WGRAD_STREAM) == 2;
    const hipStream_t down_stream = down_on_second ? net->side2 : c.side;
    const bool bwd_phases = !plan && bwd_phases_usable(c, dpool);
    // Every layer's split-K slabs -> dW.  The side stream is the longer one at the end of a step (it still holds block 0's filter
    // gradients when the main chain has finished), so the FIRST conv's filter gradient runs on the main stream (bwd_unit_post) and the
    // slabs are summed where their producers ran: the first conv's on the main stream, all others behind the side stream's last
    // kernel; then the join.
    // Round 6: the slabs of the units of the first half of `order` are summed when that half's filter gradients have been launched (same
    // stream, behind them) instead of with everything else at the step's end, where the 40 us pass over ~200 MB of slabs ran beside the
    // first conv's tail chain on the main stream.  A layer's sum does not depend on when it is taken: bitwise the one-pass result.
    const bool all_on_side = c.side != c.s && !down_on_second && tune_get(TCR_TUNE_WGRAD_STREAM) == 0;
    const int mid = (all_on_side && !bwd_phases && !plan && nu >= 8 && tune_get(TCR_TUNE_WGRAD_LDS) != 3) ? nu / 2 : -1;      // order[0 .. mid]: the early pass
    auto reduce_range = [&](int lo, int hi, bool first_conv, hipStream_t st) -> int {     // units order[lo .. hi] (the first conv only when asked)
        WgradReduceMulti rm;
        rm.n = 0;
        for (int oi = lo; oi <= hi; ++oi) {
            const int li = order[oi];
            const ConvLayer& l = net->layers[li];
            if (!conv_wgrad_deferrable(l.k, l.cin, l.cout) || (li == 0) != first_conv) continue;
            if (rm.n == kMultiMax) { TCR_TRY(launch_wgrad_reduce_multi(rm, st)); rm.n = 0; }
            rm.e[rm.n++] = conv_wgrad_entry(l.k, l.cin, l.cout, batch, c.base + c.w.wg[li], grads + l.w_off, wgrad_fine(l));
        }
        return rm.n ? launch_wgrad_reduce_multi(rm, st) : TCR_OK;
    };
    auto reduce_slabs = [&]() -> int {
        const bool split = c.side != c.s;
        auto run = [&](int which, hipStream_t st) -> int {      // 0: every layer but the first conv, 1: the first conv, 2: all
            if (which == 0 && mid >= 0) return reduce_range(mid + 1, nu - 1, false, st);        // (the early pass took order[0 .. mid])
            WgradReduceMulti rm;
            rm.n = 0;
            for (int li : order) {
                const ConvLayer& l = net->layers[li];
                if (!conv_wgrad_deferrable(l.k, l.cin, l.cout)) continue;
                if ((which == 0 && li == 0) || (which == 1 && li != 0)) continue;
                if (rm.n == kMultiMax) { TCR_TRY(launch_wgrad_reduce_multi(rm, st)); rm.n = 0; }
                rm.e[rm.n++] = conv_wgrad_entry(l.k, l.cin, l.cout, batch, c.base + c.w.wg[li], grads + l.w_off, wgrad_fine(l));
            }
            return rm.n ? launch_wgrad_reduce_multi(rm, st) : TCR_OK;
        };
        if (!split) return run(2, c.s);
        if ((down_stream != c.side || tune_get(TCR_TUNE_WGRAD_STREAM) == 4) &&      // slabs were written on the second internal stream
            (hipEventRecord(net->ev_join2, net->side2) != hipSuccess || hipStreamWaitEvent(c.side, net->ev_join2, 0) != hipSuccess)) {
            set_error("tcr_net_backward: stream join failed");
            return TCR_ERR_HIP;
        }
        TCR_TRY(run(0, c.side));
        TCR_TRY(run(1, c.s));
        if (hipEventRecord(net->ev_join, c.side) != hipSuccess || hipStreamWaitEvent(c.s, net->ev_join, 0) != hipSuccess ||
            hipEventRecord(net->ev_join2, net->side2) != hipSuccess || hipStreamWaitEvent(c.s, net->ev_join2, 0) != hipSuccess) {
            set_error("tcr_net_backward: stream join failed");
            return TCR_ERR_HIP;
        }
        return TCR_OK;
    };
    if (plan) { stage_begin = plan->first ? 0 : 1; stage_end = stage_begin + 1; }
    for (int st = stage_begin; st < stage_end; ++st) {
        if (st == 0) {
            // ONE launch: the head's backward (pooled gradient), the arena's zero fill (padding + fc2, which gets no loss gradient) and the
            // re-arranged (phase-major, transposed) weights of every data-gradient conv
            const int nc = net->cfg.num_classes;
            DgradWeightsMulti dm;
            dm.n = 0;
            for (int li : order) {
                const ConvLayer& l = net->layers[li];
                if (l.in_act < 0 || !conv_dgrad_mfma_covers(l.k, l.stride, l.cout) || dm.n >= kMultiMax) continue;
                dm.e[dm.n++] = {params + l.w_off, c.base + c.w.wtl[li], l.k, l.cin, l.cout, l.stride, l.pad_lo};
            }
            BwdPrologueArgs hp;
            hp.dlogits = c.base + c.w.dlogits; hp.wfc = params + net->layers[net->fc].w_off; hp.dscale = c.base + c.w.dscale;
            hp.dpool = c.base + c.w.dpool; hp.batch = batch; hp.c = net->feat_c; hp.nc = nc; hp.zero = grads; hp.zero_n = net->param_floats;
            TCR_TRY(launch_bwd_prologue(hp, dm, c.s));
            // the classifier's own filter gradient feeds nothing below: on a stream of its own (behind the arena's zero fill)
            hipStream_t fs = c.side != c.s ? (down_on_second ? c.side : net->side2) : c.s;
            if (fs != c.s && (hipEventRecord(net->ev_fork, c.s) != hipSuccess || hipStreamWaitEvent(fs, net->ev_fork, 0) != hipSuccess)) {
                set_error("tcr_net_backward: stream fork failed");
                return TCR_ERR_HIP;
            }
            TCR_TRY(launch_fc_wgrad(c.base + c.w.dropped, c.base + c.w.dlogits, c.base + c.w.fc_partial,
                                    grads + net->layers[net->fc].w_off, batch, net->feat_c, nc, fs));
        }
        if (plan) {     // one dependency level: the per-layer kernels on the caller's stream, shortcut units on the second scratch set
            TCR_REQUIRE(c.sync_bn, "tcr_net_backward_level: levels exist for the cross-replica hand-off");
            for (int li : plan->post) {
                const bool dn = is_down_unit(*net, li);
                TCR_TRY(bwd_unit_post(c, bwd_unit_of(c, li, dpool), grads, dpool, BWD_ALL, c.s, bwd_partial_of(c, li), c.base + (dn ? c.w.kcoef2 : c.w.kcoef)));
            }
            for (int li : plan->pre) TCR_TRY(bwd_unit_pre(c, bwd_unit_of(c, li, dpool), c.s, bwd_partial_of(c, li)));
            if (plan->last) TCR_TRY(reduce_slabs());
            continue;
        }
        if (bwd_phases) {
            // Group-resident phases.  Per unit, in `order` (block: conv_b, conv_a, down):
            //   pre  = the unit's sums exist as partial rows (written by an earlier phase; the last block's by a reduce here)
            //   post = finalize; then conv_b: D1, conv_a of an identity block / down: D2; conv0: the closing phase.
            // The weight gradients follow their dy on the side stream.
            const int nb = (int)net->blocks.size();
            auto block_of = [&](int li) { for (int bi = 0; bi < nb; ++bi) { const Block& b = net->blocks[bi]; if (li == b.b || li == b.a || li == b.down) return bi; } return -1; };
            if (st > 0) {
                const int li = order[st - 1];
                const int bi = block_of(li);
                const bool is_dn = bi >= 0 && net->blocks[bi].down == li;
                TCR_TRY(bwd_finalize(c, li, grads, bwd_partial_of(c, li), c.base + (is_dn ? c.w.kcoef2 : c.w.kcoef), bwd_rows_of(c, li, dpool)));
                if (li == 0) {
                    TCR_TRY(launch_train_bwd_phase(bwd_phase_d0(c, dpool), nullptr, c.s));
                    TCR_TRY(bwd_unit_post(c, bwd_unit_of(c, 0, dpool), grads, dpool, BWD_WGRAD, c.s, nullptr, nullptr));
                } else if (li == net->blocks[bi].b) {
                    TCR_TRY(launch_train_bwd_phase(bwd_phase_d1(c, bi, dpool), nullptr, c.s));
                    TCR_TRY(bwd_unit_post(c, bwd_unit_of(c, li, dpool), grads, dpool, BWD_WGRAD, c.s, nullptr, nullptr));
                } else if (is_dn || (li == net->blocks[bi].a && net->blocks[bi].down < 0)) {
                    TCR_TRY(launch_train_bwd_phase(bwd_phase_d2(c, bi, dpool), nullptr, c.s));
                    TCR_TRY(bwd_unit_post(c, bwd_unit_of(c, net->blocks[bi].a, dpool), grads, dpool, BWD_WGRAD, c.s, nullptr, nullptr));
                    if (is_dn) TCR_TRY(bwd_unit_post(c, bwd_unit_of(c, li, dpool), grads, dpool, BWD_WGRAD, c.s, nullptr, nullptr));
                }
            }
            if (st < nu) {
                const int li = order[st];
                if (li == net->blocks[nb - 1].b || li == net->blocks[nb - 1].down)      // (gradient of the pooled head, broadcast over time)
                    TCR_TRY(bwd_unit_pre(c, bwd_unit_of(c, li, dpool), c.s, bwd_partial_of(c, li)));
                else if (c.sync_bn)
                    TCR_TRY(launch_chan_sums(bwd_partial_of(c, li), bwd_rows_of(c, li, dpool), net->layers[li].cout, sums_of(c, li), c.s));
            }
            if (st == nu) TCR_TRY(reduce_slabs());
            continue;
        }
        // Without cross-replica statistics the shortcut ("down") unit of a block is started EARLY on the side stream: its BN
        // backward and weight gradient depend only on the block-output gradient, which is ready when the block begins; only
        // its data gradient (accumulated onto conv_a's) stays in the main chain.
        const bool early = !c.sync_bn && c.side != c.s;
        auto down_of_block_starting_at = [&](int li) { for (const Block& b : net->blocks) if (b.b == li && b.down >= 0) return b.down; return -1; };
        auto is_down = [&](int li) { for (const Block& b : net->blocks) if (b.down == li) return true; return false; };
        auto down_first_of = [&](int li) { for (const Block& b : net->blocks) if (b.down == li) return early && down_dgrad_first(c, b); return false; };
        float* partial = c.base + c.w.partial;
        float* kc = c.base + c.w.kcoef;
        if (st > 0) {
            const int li = order[st - 1];
            if (early && is_down(li)) {
                if (!down_first_of(li)) {       // (else: ran on the side stream when the block began)
                    if (hipStreamWaitEvent(c.s, net->ev_down, 0) != hipSuccess) { set_error("tcr_net_backward: stream wait failed"); return TCR_ERR_HIP; }
                    TCR_TRY(bwd_unit_post(c, bwd_unit_of(c, li, dpool), grads, dpool, BWD_DGRAD, c.s, partial, kc));
                }
            } else {
                TCR_TRY(bwd_unit_post(c, bwd_unit_of(c, li, dpool), grads, dpool, BWD_ALL, c.s, partial, kc));
            }
            if (st - 1 == mid) TCR_TRY(reduce_range(0, mid, false, c.side));        // (behind the filter gradients launched so far)
        }
        if (st < nu) {
            const int li = order[st];
            const int dn = early ? down_of_block_starting_at(li) : -1;
            if (dn >= 0) {
                if (hipEventRecord(net->ev_fork, c.s) != hipSuccess || hipStreamWaitEvent(down_stream, net->ev_fork, 0) != hipSuccess) {
                    set_error("tcr_net_backward: stream fork failed");
                    return TCR_ERR_HIP;
                }
                const BwdUnit ud = bwd_unit_of(c, dn, dpool);
                TCR_TRY(bwd_unit_pre(c, ud, down_stream, c.base + c.w.partial2));
                TCR_TRY(bwd_unit_post(c, ud, grads, dpool, BWD_BN, down_stream, c.base + c.w.partial2, c.base + c.w.kcoef2));
                if (hipEventRecord(net->ev_down, down_stream) != hipSuccess) { set_error("tcr_net_backward: event record failed"); return TCR_ERR_HIP; }
                if (down_first_of(dn)) {
                    TCR_TRY(bwd_unit_post(c, ud, grads, dpool, BWD_DGRAD, down_stream, c.base + c.w.partial2, c.base + c.w.kcoef2, down_stream));
                    if (hipEventRecord(net->ev_down_dg, down_stream) != hipSuccess) { set_error("tcr_net_backward: event record failed"); return TCR_ERR_HIP; }
                }
                TCR_TRY(bwd_unit_post(c, ud, grads, dpool, BWD_WGRAD, down_stream, c.base + c.w.partial2, c.base + c.w.kcoef2));
            }
            if (!(early && is_down(li))) TCR_TRY(bwd_unit_pre(c, bwd_unit_of(c, li, dpool), c.s, partial));
        }
        if (st == nu) TCR_TRY(reduce_slabs());
    }
    return TCR_OK;
}

extern "C" int tcr_net_backward(const tcr_net* net, const float* params, const float* feat, int batch,
                                void* workspace, size_t workspace_bytes, float* grads, void* stream) {
    const int nu = net ? (int)net->units.size() : 0;
    return backward_stages(net, params, feat, batch, batch, 0, workspace, workspace_bytes, grads, 0, nu + 1, stream);
}

extern "C" int tcr_net_backward_stage(const tcr_net* net, const float* params, const float* feat, int batch, int global_batch,
                                      void* workspace, size_t workspace_bytes, float* grads, int stage, void* stream) {
    return backward_stages(net, params, feat, batch, global_batch, 1, workspace, workspace_bytes, grads, stage, stage + 1, stream);
}

extern "C" int tcr_net_backward_level(const tcr_net* net, const float* params, const float* feat, int batch, int global_batch,
                                      void* workspace, size_t workspace_bytes, float* grads, int level, void* stream) {
    TCR_REQUIRE(net && level >= 0 && level < tcr_net_num_levels(net, 1), "tcr_net_backward_level: bad level %d", level);
    const tcr::LevelPlan plan = tcr::level_plan(*net, 1, level);
    return backward_stages(net, params, feat, batch, global_batch, 1, workspace, workspace_bytes, grads, 0, 1, stream, &plan);
}
