// DS-CNN (S / M / L) eval-mode forward for gfx950: the depthwise-separable baseline of BASELINE.json configs[4].
//
// Replaces DSCNN() + DSCNN_arg_scope (audio_nets/ds_cnn.py:19-118): conv_1 (10x4, +bias) -> BN(no scale)+ReLU,
// N x [depthwise 3x3 (+bias) -> BN+ReLU -> pointwise 1x1 (+bias) -> BN+ReLU], global average pool,
// fully_connected (+bias), softmax (factory/audio_nets.py:154).  Input: the [B, 49, 10, 1] MFCC map, taken
// straight from the front-end's planar buffer feat[b][f][HALO + t] (H = time, W = coefficient).
//
// Kernels: conv_1 is an implicit GEMM on the exact-f32 MFMA (K = 10 kernel rows x 4 kernel columns: one MFMA
// K-step per kernel row); the pointwise convs (88 % of the MACs, 276 x 276 per position) and the classifier reuse
// the matrix-core kernels of mfma.hip / head.hip; the depthwise 3x3 is a small VALU stencil.  Conv bias and
// eval-mode BN are folded into one per-channel scale/shift applied in the producing kernel's epilogue.
// Internal layout: planar [b][c][HALO + h*W + w] (row length tcr_padded_len(H*W)).
#include <cstring>
#include <string>
#include <vector>

#include "kernels.h"

namespace tcr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DsConv1Args {
    const float* feat;      // [B][W_in][Tp_in]  (front-end planar: coefficient-major, time along the row)
    const float* w;         // [kh][4][1][Cout]
    const float* scale;
    const float* shift;
    float* y;               // [B][Cout][Pp]
    int npos;               // B * OH * OW
    int cout, h_in, w_in, tp_in, oh, ow, pp;
    int kh, sh, sw, pad_t, pad_l;
};

// D[row = co][col = (b, oh, ow)] = sum_{i < kh, j < 4} W[i][j][co] * x[oh*sh + i - pad_t][ow*sw + j - pad_l]
template <int MT>
__global__ __launch_bounds__(256) void dscnn_conv1_kernel(const DsConv1Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int pos0 = (blockIdx.x * 4 + wave) * 64;
    const int cot0 = blockIdx.y * MT;
    const int P = a.oh * a.ow;
    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* xb[4];
    int h0[4];
    bool wok[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int p = min(pos0 + nt * 16 + r, a.npos - 1);
        const int n = p / P, rem = p - n * P;
        const int oh = rem / a.ow, ow = rem - oh * a.ow;
        const int wc = ow * a.sw + q - a.pad_l;                 // input column (MFCC coefficient) of this lane's tap j = q
        wok[nt] = wc >= 0 && wc < a.w_in;
        xb[nt] = a.feat + ((size_t)n * a.w_in + (wok[nt] ? wc : 0)) * a.tp_in + kHalo;
        h0[nt] = oh * a.sh - a.pad_t;
    }
    for (int i = 0; i < a.kh; ++i) {
        float af[MT], bf[4];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co = (cot0 + m) * 16 + r;
            af[m] = co < a.cout ? a.w[((size_t)i * 4 + q) * a.cout + co] : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int h = h0[nt] + i;
            bf[nt] = (wok[nt] && h >= 0 && h < a.h_in) ? xb[nt][h] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[nt], acc[m][nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int p = pos0 + nt * 16 + r;
        if (p >= a.npos) continue;
        const int n = p / P, rem = p - n * P;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = (cot0 + m) * 16 + q * 4 + reg;
                if (co >= a.cout) continue;
                const float v = fmaxf(fmaf(acc[m][nt][reg], a.scale[co], a.shift[co]), 0.f);
                a.y[((size_t)n * a.cout + co) * a.pp + kHalo + rem] = v;
            }
    }
}

struct DsDwArgs {
    const float* x;         // [B][C][Ppi]
    const float* w;         // [3][3][C][1]
    const float* scale;
    const float* shift;
    float* y;               // [B][C][Ppo]
    int64_t total;          // B * C * OH * OW
    int c, h_in, w_in, ppi, oh, ow, ppo, sh, sw, pad_t, pad_l;
};

__global__ __launch_bounds__(256) void dscnn_depthwise_kernel(const DsDwArgs a) {
    // one wavefront per (utterance, channel) plane: lanes walk the output map, the 9 taps / scale / shift of the
    // plane's channel are wave-uniform (scalar loads)
    const int P = a.oh * a.ow;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // b * C + c
    if (row * P >= a.total) return;
    const int c = (int)(row % a.c);
    float wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = a.w[(size_t)k * a.c + c];
    const float sc = a.scale[c], sh = a.shift[c];
    const float* xr = a.x + row * a.ppi + kHalo;
    float* yr = a.y + row * a.ppo + kHalo;
    for (int pos = lane; pos < P; pos += 64) {
        const int oh = pos / a.ow, ow = pos - oh * a.ow;
        float s = 0.f;
#pragma unroll
        for (int di = 0; di < 3; ++di) {
            const int h = oh * a.sh + di - a.pad_t;
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                const int w = ow * a.sw + dj - a.pad_l;
                const float xv = (h >= 0 && h < a.h_in && w >= 0 && w < a.w_in) ? xr[h * a.w_in + w] : 0.f;
                s = fmaf(wt[di * 3 + dj], xv, s);
            }
        }
        yr[pos] = fmaxf(fmaf(s, sc, sh), 0.f);
    }
}

// pooled[b][c][HALO] = mean over the P positions of plane (b, c): one wavefront per plane (coalesced row read +
// 64-lane shuffle reduction); feeds head_fwd_kernel with T = 1 when the map is large (DS-CNN: 13 x 5 = 65).
__global__ __launch_bounds__(256) void plane_mean_kernel(const float* __restrict__ x, float* __restrict__ pooled, int64_t rows, int p, int pp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * pp + kHalo;
    float s = 0.f;
    for (int i = lane; i < p; i += 64) s += xr[i];
    s = wave_sum(s);
    if (lane == 0) pooled[row * (1 + 2 * kHalo) + kHalo] = s / (float)p;
}

struct DsLayer {
    std::string scope;
    bool separable;
    int cin, cout, kh, kw, sh, sw, h_in, w_in, oh, ow, pad_t, pad_l;
    // offsets (floats): trainable arena / stat arena / scale-shift workspace slot
    int64_t w_off, b_off, beta_off, mean_off, var_off, ss_off;             // conv or depthwise
    int64_t pw_off, pb_off, pbeta_off, pmean_off, pvar_off, pss_off;       // pointwise (separable only)
};

}  // namespace tcr

using namespace tcr;

struct tcr_dscnn {
    tcr_dscnn_cfg cfg;
    std::vector<DsLayer> layers;
    int64_t fcw_off, fcb_off;
    int64_t param_floats, stat_floats, ss_floats;
    int c_pad;
    std::vector<tcr_tensor_info> tensors;
};

namespace tcr {
static int64_t ds_align(int64_t v) { return (v + 63) / 64 * 64; }
}

extern "C" int tcr_dscnn_create(const tcr_dscnn_cfg* cfg, tcr_dscnn** out) {
    TCR_REQUIRE(cfg && out, "tcr_dscnn_create: null argument");
    TCR_REQUIRE(cfg->depth > 0 && cfg->depth % 4 == 0, "tcr_dscnn_create: depth %d must be a positive multiple of 4", cfg->depth);
    TCR_REQUIRE(cfg->n_separable >= 1 && cfg->n_separable <= 8, "tcr_dscnn_create: n_separable %d out of range", cfg->n_separable);
    TCR_REQUIRE(cfg->h_in > 0 && cfg->w_in > 0 && cfg->num_classes > 0 && cfg->num_classes + 2 <= 48, "tcr_dscnn_create: bad shape");
    TCR_REQUIRE(cfg->conv1_kh >= 1 && cfg->conv1_kh <= 16 && cfg->conv1_kw == 4, "tcr_dscnn_create: conv_1 kernel must be kh x 4 (got %d x %d)", cfg->conv1_kh, cfg->conv1_kw);
    tcr_dscnn* net = new tcr_dscnn();
    net->cfg = *cfg;
    net->c_pad = (int)ds_align(cfg->depth);
    int64_t o = 0, so = 0, ss = 0;
    auto info = [&](const std::string& name, int kind, int arena, int64_t off, int64_t size, std::vector<int> shape) {
        tcr_tensor_info ti;
        std::memset(&ti, 0, sizeof(ti));
        std::snprintf(ti.name, sizeof(ti.name), "DSCNN/%s", name.c_str());
        ti.kind = kind; ti.arena = arena; ti.offset = off; ti.size = size; ti.rank = (int)shape.size();
        for (size_t i = 0; i < shape.size() && i < 4; ++i) ti.shape[i] = shape[i];
        net->tensors.push_back(ti);
    };
    auto take = [&](int64_t n) { int64_t at = o; o = ds_align(o + n + 64); return at; };
    auto take_stat = [&](int64_t n) { int64_t at = so; so += ds_align(n); return at; };
    auto bn = [&](const std::string& prefix, int c, int64_t* beta, int64_t* mean, int64_t* var, int64_t* slot) {
        *beta = take(c); info(prefix + "/beta", TCR_BETA, 0, *beta, c, {c});
        *mean = take_stat(c); info(prefix + "/moving_mean", TCR_MOVING_MEAN, 1, *mean, c, {c});
        *var = take_stat(c); info(prefix + "/moving_variance", TCR_MOVING_VAR, 1, *var, c, {c});
        *slot = ss; ss += 2 * net->c_pad;
    };
    int h = cfg->h_in, w = cfg->w_in, c = 1;
    for (int li = 0; li <= cfg->n_separable; ++li) {
        DsLayer l;
        l.separable = li > 0;
        l.scope = li == 0 ? "conv_1" : "conv_ds_" + std::to_string(li);
        l.cin = c; l.cout = cfg->depth;
        l.kh = li == 0 ? cfg->conv1_kh : 3; l.kw = li == 0 ? cfg->conv1_kw : 3;
        l.sh = li == 0 ? cfg->conv1_sh : (li == 1 ? cfg->ds1_sh : 1);
        l.sw = li == 0 ? cfg->conv1_sw : (li == 1 ? cfg->ds1_sw : 1);
        l.h_in = h; l.w_in = w;
        int ph, pw2;
        same_pad(h, l.kh, l.sh, &l.oh, &l.pad_t, &ph);
        same_pad(w, l.kw, l.sw, &l.ow, &l.pad_l, &pw2);
        if (!l.separable) {
            l.w_off = take((int64_t)l.kh * l.kw * l.cin * l.cout);
            info(l.scope + "/weights", TCR_WEIGHT, 0, l.w_off, (int64_t)l.kh * l.kw * l.cin * l.cout, {l.kh, l.kw, l.cin, l.cout});
            l.b_off = take(l.cout); info(l.scope + "/biases", TCR_BETA, 0, l.b_off, l.cout, {l.cout});
            bn(l.scope + "/batch_norm", l.cout, &l.beta_off, &l.mean_off, &l.var_off, &l.ss_off);
            l.pw_off = l.pb_off = l.pbeta_off = l.pmean_off = l.pvar_off = l.pss_off = -1;
        } else {
            l.w_off = take(9 * (int64_t)l.cin);
            info(l.scope + "/depthwise_conv/depthwise_weights", TCR_WEIGHT, 0, l.w_off, 9 * (int64_t)l.cin, {3, 3, l.cin, 1});
            l.b_off = take(l.cin); info(l.scope + "/depthwise_conv/biases", TCR_BETA, 0, l.b_off, l.cin, {l.cin});
            bn(l.scope + "/dw_batch_norm", l.cin, &l.beta_off, &l.mean_off, &l.var_off, &l.ss_off);
            l.pw_off = take((int64_t)l.cin * l.cout);
            info(l.scope + "/pointwise_conv/weights", TCR_WEIGHT, 0, l.pw_off, (int64_t)l.cin * l.cout, {1, 1, l.cin, l.cout});
            l.pb_off = take(l.cout); info(l.scope + "/pointwise_conv/biases", TCR_BETA, 0, l.pb_off, l.cout, {l.cout});
            bn(l.scope + "/pw_batch_norm", l.cout, &l.pbeta_off, &l.pmean_off, &l.pvar_off, &l.pss_off);
        }
        net->layers.push_back(l);
        h = l.oh; w = l.ow; c = l.cout;
    }
    net->fcw_off = take((int64_t)c * cfg->num_classes);
    info("fc1/weights", TCR_WEIGHT, 0, net->fcw_off, (int64_t)c * cfg->num_classes, {c, cfg->num_classes});
    net->fcb_off = take(cfg->num_classes);
    info("fc1/biases", TCR_BETA, 0, net->fcb_off, cfg->num_classes, {cfg->num_classes});
    net->param_floats = o; net->stat_floats = so; net->ss_floats = ss;
    *out = net;
    return TCR_OK;
}

extern "C" void tcr_dscnn_destroy(tcr_dscnn* net) { delete net; }
extern "C" int64_t tcr_dscnn_param_floats(const tcr_dscnn* net) { return net ? net->param_floats : 0; }
extern "C" int64_t tcr_dscnn_stat_floats(const tcr_dscnn* net) { return net ? net->stat_floats : 0; }
extern "C" int tcr_dscnn_num_tensors(const tcr_dscnn* net) { return net ? (int)net->tensors.size() : 0; }
extern "C" int tcr_dscnn_tensor_info(const tcr_dscnn* net, int index, tcr_tensor_info* out) {
    TCR_REQUIRE(net && out && index >= 0 && index < (int)net->tensors.size(), "tcr_dscnn_tensor_info: bad argument");
    *out = net->tensors[index];
    return TCR_OK;
}

namespace tcr {
static int64_t ds_act_floats(const tcr_dscnn& net, int batch) {
    int64_t m = 0;
    for (const DsLayer& l : net.layers) {
        const int64_t n = (int64_t)batch * l.cout * tcr_padded_len(l.oh * l.ow);
        m = n > m ? n : m;
    }
    return ds_align(m);
}
}  // namespace tcr

extern "C" size_t tcr_dscnn_workspace_bytes(const tcr_dscnn* net, int batch) {
    if (!net || batch <= 0) return 0;
    return (size_t)(ds_align(net->ss_floats) + 2 * ds_act_floats(*net, batch)) * sizeof(float);
}

extern "C" int tcr_dscnn_forward_infer(const tcr_dscnn* net, const float* params, const float* stats, const float* feat,
                                       int batch, void* workspace, size_t workspace_bytes, float* logits, float* probs, void* stream) {
    TCR_REQUIRE(net && params && stats && feat && workspace && logits && probs, "tcr_dscnn_forward_infer: null argument");
    TCR_REQUIRE(batch > 0, "tcr_dscnn_forward_infer: batch must be positive (got %d)", batch);
    if (tcr_dscnn_workspace_bytes(net, batch) > workspace_bytes) {
        set_error("tcr_dscnn_forward_infer: workspace %zu bytes < required %zu", workspace_bytes, tcr_dscnn_workspace_bytes(net, batch));
        return TCR_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* ss = static_cast<float*>(workspace);
    float* buf[2] = {ss + ds_align(net->ss_floats), ss + ds_align(net->ss_floats) + ds_act_floats(*net, batch)};
    const int cp = net->c_pad;

    // conv bias + eval-mode BN (no scale) -> per-channel scale / shift, all layers in one launch
    BnFoldArgs f;
    std::memset(&f, 0, sizeof(f));
    f.params = params; f.stats = stats; f.out = ss; f.eps = net->cfg.bn_eps;
    auto add_fold = [&](int c, int64_t beta, int64_t mean, int64_t var, int64_t bias, int64_t slot) {
        f.c[f.n] = c; f.c_pad[f.n] = cp; f.gamma_off[f.n] = -1; f.beta_off[f.n] = beta; f.mean_off[f.n] = mean;
        f.var_off[f.n] = var; f.bias_off[f.n] = bias; f.out_off[f.n] = slot; ++f.n;
    };
    for (const DsLayer& l : net->layers) {
        add_fold(l.separable ? l.cin : l.cout, l.beta_off, l.mean_off, l.var_off, l.b_off, l.ss_off);
        if (l.separable) add_fold(l.cout, l.pbeta_off, l.pmean_off, l.pvar_off, l.pb_off, l.pss_off);
    }
    TCR_TRY(launch_bn_fold(f, s));

    int cur = 0;
    for (const DsLayer& l : net->layers) {
        const int P = l.oh * l.ow, pp = tcr_padded_len(P);
        if (!l.separable) {
            DsConv1Args a;
            a.feat = feat; a.w = params + l.w_off; a.scale = ss + l.ss_off; a.shift = ss + l.ss_off + cp; a.y = buf[cur];
            a.npos = batch * P; a.cout = l.cout; a.h_in = l.h_in; a.w_in = l.w_in; a.tp_in = tcr_padded_len(l.h_in);
            a.oh = l.oh; a.ow = l.ow; a.pp = pp; a.kh = l.kh; a.sh = l.sh; a.sw = l.sw; a.pad_t = l.pad_t; a.pad_l = l.pad_l;
            const int tiles = ceil_div(l.cout, 16);
            const dim3 grid(ceil_div(a.npos, 256), ceil_div(tiles, 3));
            hipLaunchKernelGGL((dscnn_conv1_kernel<3>), grid, dim3(256), 0, s, a);
            TCR_TRY(check_launch("dscnn_conv1_kernel"));
        } else {
            DsDwArgs d;
            d.x = buf[cur]; d.w = params + l.w_off; d.scale = ss + l.ss_off; d.shift = ss + l.ss_off + cp; d.y = buf[cur ^ 1];
            d.total = (int64_t)batch * l.cin * P; d.c = l.cin; d.h_in = l.h_in; d.w_in = l.w_in;
            d.ppi = tcr_padded_len(l.h_in * l.w_in); d.oh = l.oh; d.ow = l.ow; d.ppo = pp;
            d.sh = l.sh; d.sw = l.sw; d.pad_t = l.pad_t; d.pad_l = l.pad_l;
            const int64_t blocks = ceil_div64((int64_t)batch * l.cin, 4);
            hipLaunchKernelGGL(dscnn_depthwise_kernel, dim3((unsigned)blocks), dim3(256), 0, s, d);
            TCR_TRY(check_launch("dscnn_depthwise_kernel"));
            Conv1x1Args c1;
            c1.x = buf[cur ^ 1]; c1.w = params + l.pw_off; c1.y = buf[cur]; c1.scale = ss + l.pss_off; c1.shift = ss + l.pss_off + cp;
            c1.npos = batch * P; c1.cin = l.cin; c1.cout = l.cout; c1.tpi = pp; c1.tout = P; c1.tpo = pp; c1.stride = 1; c1.relu = 1;
            TCR_TRY(launch_conv1x1(c1, MF_AFFINE, s));
        }
    }
    const DsLayer& last = net->layers.back();
    const int P = last.oh * last.ow;
    const int64_t rows = (int64_t)batch * last.cout;
    hipLaunchKernelGGL(plane_mean_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, s, (const float*)buf[cur], buf[cur ^ 1], rows, P,
                       tcr_padded_len(P));
    TCR_TRY(check_launch("plane_mean_kernel"));
    HeadArgs h;
    std::memset(&h, 0, sizeof(h));
    h.feat = buf[cur ^ 1]; h.wfc = params + net->fcw_off; h.wfc2 = nullptr; h.bias = params + net->fcb_off;
    h.logits = logits; h.probs = probs; h.ranges = nullptr;
    h.batch = batch; h.c = last.cout; h.nc = net->cfg.num_classes; h.t = 1; h.tp = tcr_padded_len(1);
    h.keep_prob = 1.0f; h.inv_global_batch = 1.0f;
    return launch_head_fwd(h, false, s);
}
