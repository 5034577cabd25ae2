// Argument blocks / launchers of the generic 2-D layer-graph kernels (net2d_kernels.hip), used by net2d.cpp.
#pragma once
#include "tcr_common.h"

namespace tcr {

struct Conv2dArgs {
    const float* x;         // gathered operand: forward / wgrad: input [B][Cin][ppi]; dgrad: dy [B][Cout][ppo]
    const float* wgt;       // [kh][kw][Cin][Cout]  (TF HWIO)
    float* y;               // forward: output [B][Cout][ppo] (assigned); dgrad: dx [B][Cin][ppi] (accumulated)
    const float* bias;      // forward: [Cout] or nullptr
    const float* dy;        // wgrad: [B][Cout][ppo]
    int batch, cin, cout;
    int h, w, oh, ow;       // input / output planes
    int kh, kw, sh, sw, dh, dw, pt, pl;
    int ppi, ppo;           // row lengths: h*w + 2*kHalo, oh*ow + 2*kHalo
    int relu;
};

struct Pool2dArgs {
    const float* x;         // [planes][ppi]
    float* y;               // [planes][ppo]
    int* argmax;            // [planes][oh*ow] (max pooling, training)
    const float* dy;        // backward
    float* dx;              // backward (accumulated)
    int64_t planes;         // B * C
    int h, w, oh, ow, kh, kw, sh, sw, pt, pl, ppi, ppo, is_max;
};

struct Elt2dArgs {
    const float* a;
    const float* b;
    float* out;
    int64_t planes;         // B * C
    int c, plane, pp, relu;
    float keep_prob;
    uint64_t seed;
    int64_t sample_offset;
    int node;
};

int launch_conv2d_fwd(const Conv2dArgs& a, hipStream_t s);
int launch_conv2d_dgrad(const Conv2dArgs& a, hipStream_t s);
int conv2d_wgrad_chunks(int batch);
size_t conv2d_wgrad_partial_floats(int kh, int kw, int cin, int cout, int batch);
int launch_conv2d_wgrad(const Conv2dArgs& a, float* dw, float* scratch, hipStream_t s);
int launch_chan_sum2d(const float* g, float* out, int batch, int c, int plane, int pp, hipStream_t s);
int launch_pool2d_fwd(const Pool2dArgs& a, hipStream_t s);
int launch_pool2d_bwd(const Pool2dArgs& a, hipStream_t s);
int launch_eltwise2d(int mode, const Elt2dArgs& a, hipStream_t s);
int launch_head2d(const float* z, const float* labels, float* logits, float* probs, float* dz, float* loss_utt, int batch, int nc, int pp,
                  float inv_global_batch, float label_smoothing, hipStream_t s);
int launch_tfilt_fwd(const float* x, const float* w, float* y, int batch, int c, int plane, hipStream_t s);
int launch_tfilt_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, int batch, int c, int plane, hipStream_t s);
int launch_gsum_fwd(const float* x, const float* bias, float* y, int batch, int units, int group, int relu, hipStream_t s);
int launch_gsum_dx(const float* dy, float* dx, int batch, int units, int group, hipStream_t s);
int launch_features_to_plane(const float* feat, float* out, int batch, int t, int f, hipStream_t s);

}  // namespace tcr
