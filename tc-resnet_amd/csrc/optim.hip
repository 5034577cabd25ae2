// Optimiser steps on the flat parameter arena (one launch per step for the whole model) and the
// L2 regulariser.
//
// Replaces tf.train.MomentumOptimizer / tf.train.AdamOptimizer as built by
// TrainerBase.build_optimizer (helper/trainer.py:171-197) and the
// weight_decay * sum(tf.nn.l2_loss(v)) term of AudioNetModel.build_loss
// (factory/audio_nets.py:175-182): its gradient wd * v is folded into the update for the first
// n_decay floats of the arena (all conv / fc weights; BN gamma/beta follow and are not decayed).
#include <cmath>

#include "tcr_common.h"

namespace tcr {

__global__ __launch_bounds__(256) void sgd_momentum_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                           int64_t n, int64_t n_decay, float lr, float mu, float wd, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float wv = w[i];
        float gv = g[i] * gscale;
        if (i < n_decay) gv = fmaf(wd, wv, gv);
        const float a = fmaf(mu, m[i], gv);         // accum = momentum * accum + grad
        m[i] = a;
        w[i] = fmaf(-lr, a, wv);                    // var -= lr * accum
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, int64_t n_decay, float lr_t, float b1, float b2,
                                                   float eps, float wd, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float wv = w[i];
        float gv = g[i] * gscale;
        if (i < n_decay) gv = fmaf(wd, wv, gv);
        const float mv = m[i] + (1.0f - b1) * (gv - m[i]);
        const float vv = v[i] + (1.0f - b2) * (gv * gv - v[i]);
        m[i] = mv;
        v[i] = vv;
        w[i] = wv - lr_t * mv / (sqrtf(vv) + eps);
    }
}

// tf.train.RMSPropOptimizer (centered=False): ms <- decay*ms + (1-decay)*g^2 ; mom <- momentum*mom + lr*g/sqrt(ms + eps) ;
// var <- var - mom.  (training_ops ApplyRMSProp: epsilon sits INSIDE the square root; the `rms` slot starts at ONE.)
__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ ms,
                                                      float* __restrict__ mom, int64_t n, int64_t n_decay, float lr, float decay,
                                                      float momentum, float eps, float wd, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float wv = w[i];
        float gv = g[i] * gscale;
        if (i < n_decay) gv = fmaf(wd, wv, gv);
        const float msv = ms[i] + (1.0f - decay) * (gv * gv - ms[i]);
        const float mv = momentum * mom[i] + lr * gv / sqrtf(msv + eps);
        ms[i] = msv;
        mom[i] = mv;
        w[i] = wv - mv;
    }
}

// tf.train.ExponentialMovingAverage.apply: shadow -= (1 - decay) * (shadow - var)   (assign_moving_average, zero_debias off)
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ shadow, const float* __restrict__ w, int64_t n, float one_minus_decay) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float sv = shadow[i];
        shadow[i] = sv - one_minus_decay * (sv - w[i]);
    }
}

__global__ __launch_bounds__(256) void l2_loss_kernel(const float* __restrict__ w, int64_t n, float wd, float* __restrict__ out) {
    __shared__ double s_part[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += 0.5 * (double)w[i] * (double)w[i];
    s_part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < 256; ++i) tot += s_part[i];
        out[0] = (float)((double)wd * tot);
    }
}

static int grid_for(int64_t n) {
    int64_t b = ceil_div64(n, 256);
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace tcr

using namespace tcr;

extern "C" int tcr_sgd_momentum_step(float* params, const float* grads, float* momentum, int64_t n, int64_t n_decay,
                                     float lr, float mu, float weight_decay, float grad_scale, void* stream) {
    TCR_REQUIRE(params && grads && momentum && n > 0 && n_decay >= 0 && n_decay <= n, "tcr_sgd_momentum_step: bad argument");
    hipLaunchKernelGGL(sgd_momentum_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), params, grads, momentum,
                       n, n_decay, lr, mu, weight_decay, grad_scale);
    return check_launch("sgd_momentum_kernel");
}

extern "C" int tcr_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, int64_t n_decay, float lr,
                             float beta1, float beta2, float eps, int64_t t, float weight_decay, float grad_scale, void* stream) {
    TCR_REQUIRE(params && grads && m && v && n > 0 && t >= 1 && n_decay >= 0 && n_decay <= n, "tcr_adam_step: bad argument");
    // tf.train.AdamOptimizer: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t); eps is NOT bias-corrected
    const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)t)) / (1.0 - std::pow((double)beta1, (double)t));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), params, grads, m, v, n, n_decay,
                       (float)lr_t, beta1, beta2, eps, weight_decay, grad_scale);
    return check_launch("adam_kernel");
}

extern "C" int tcr_l2_loss(const float* params, int64_t n_decay, float weight_decay, float* out, void* stream) {
    TCR_REQUIRE(params && out && n_decay >= 0, "tcr_l2_loss: bad argument");
    hipLaunchKernelGGL(l2_loss_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), params, n_decay, weight_decay, out);
    return check_launch("l2_loss_kernel");
}

extern "C" int tcr_rmsprop_step(float* params, const float* grads, float* ms, float* mom, int64_t n, int64_t n_decay, float lr,
                                float decay, float momentum, float eps, float weight_decay, float grad_scale, void* stream) {
    TCR_REQUIRE(params && grads && ms && mom && n > 0 && n_decay >= 0 && n_decay <= n, "tcr_rmsprop_step: bad argument");
    hipLaunchKernelGGL(rmsprop_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), params, grads, ms, mom, n,
                       n_decay, lr, decay, momentum, eps, weight_decay, grad_scale);
    return check_launch("rmsprop_kernel");
}

extern "C" int tcr_ema_step(float* shadow, const float* params, int64_t n, float decay, void* stream) {
    TCR_REQUIRE(shadow && params && n > 0 && decay >= 0.f && decay <= 1.f, "tcr_ema_step: bad argument");
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), shadow, params, n, 1.0f - decay);
    return check_launch("ema_kernel");
}
