// Classifier head: global average pool -> (dropout) -> fc / fc2 1x1 convs -> softmax / sigmoid,
// the mean cross-entropy and its gradient.
//
// Replaces slim.avg_pool2d + slim.dropout + the two 1x1 slim.conv2d heads + sigmoid
// (audio_nets/tc_resnet.py:43-52), slim.softmax (factory/audio_nets.py:154) and
// tf.losses.softmax_cross_entropy (factory/audio_nets.py:168-173).
//
// The dense contraction runs on the matrix cores: D[class][utterance] = sum_c Wcat[c][class] *
// pooled[c][utterance] with the exact-f32 16x16x4 MFMA; rows 0..nc-1 are the logits, rows nc, nc+1
// the two "ranges" outputs of fc2.  A wave owns 16 utterances; the softmax over the class rows is a
// register + 2-step cross-lane (xor 16, 32) reduction.
#include "kernels.h"

namespace tcr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MT, bool TRAIN>
__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int u = (blockIdx.x * 4 + wave) * 16 + r;
    const bool uv = u < a.batch;
    const int n = uv ? u : a.batch - 1;

    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Four channel quads per trip: their 4 x 8 feature loads and 4 x MT weight loads are all requested before the first sum (one quad per
    // trip paid a memory round trip per quad -- 12 in a row for 48 channels on 64 workgroups: 26 us at the head of the training step's
    // backward).  Frames are added in frame order and the K-steps run in channel order: bitwise the one-quad loop.
    constexpr int CQ = 4;
    for (int c0 = 0; c0 < a.c; c0 += 4 * CQ) {
        float sum[CQ], afv[CQ][MT];
        const float* row[CQ];
#pragma unroll
        for (int j = 0; j < CQ; ++j) {
            const int cc = min(c0 + 4 * j + q, a.c - 1);
            row[j] = a.feat + ((size_t)n * a.c + cc) * a.tp + kHalo;
            sum[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < CQ; ++j) {
            const int c = c0 + 4 * j + q;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int o = m * 16 + r;                                   // A operand: Wcat[c][o]
                float af = 0.f;
                if (c < a.c) {
                    if (o < a.nc) af = a.wfc[(size_t)c * a.nc + o];
                    else if (o < a.nc + 2 && a.wfc2) af = a.wfc2[(size_t)c * 2 + (o - a.nc)];
                }
                afv[j][m] = af;
            }
        }
        if (a.pool_sum) {           // the sums over time are given (the closing training phase's; frames added in the order of the loop below)
#pragma unroll
            for (int j = 0; j < CQ; ++j) sum[j] = a.pool_sum[(size_t)n * a.c + min(c0 + 4 * j + q, a.c - 1)];
        } else
        for (int t0 = 0; t0 < a.t; t0 += 8) {
            float v[CQ][8];
#pragma unroll
            for (int j = 0; j < CQ; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[j][i] = row[j][min(t0 + i, a.t - 1)];
#pragma unroll
            for (int j = 0; j < CQ; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (t0 + i < a.t) sum[j] += v[j][i];
        }
#pragma unroll
        for (int j = 0; j < CQ; ++j) {
            if (c0 + 4 * j >= a.c) break;                                   // (wave-uniform: the quads past the last channel)
            const int c = c0 + 4 * j + q;
            const bool cv = c < a.c;
            const int cc = cv ? c : a.c - 1;
            float pooled = sum[j] / (float)a.t;                             // tc_resnet.py:43
            if (TRAIN) {
                float ds = a.pool_scale > 0.f ? a.pool_scale : 1.0f / (float)a.t;     // d(pooled)/d(position)
                if (a.keep_prob < 1.0f) {                                   // tf.nn.dropout: div(x, keep_prob) * mask
                    const float rnd = uniform01(a.seed, (uint64_t)(a.sample_offset + n) * (uint64_t)a.c + (uint64_t)cc);
                    const bool keep = rnd < a.keep_prob;
                    pooled = keep ? pooled / a.keep_prob : 0.f;
                    ds = keep ? ds / a.keep_prob : 0.f;
                }
                if (uv && cv) {
                    a.dropped[(size_t)n * a.c + c] = pooled;
                    a.dscale[(size_t)n * a.c + c] = ds;
                }
            }
            const float bf = cv ? pooled : 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(afv[j][m], bf, acc[m], 0, 0, 0);
        }
    }

    if (a.bias) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int o = m * 16 + q * 4 + reg;
                if (o < a.nc) acc[m][reg] += a.bias[o];
            }
    }
    // this lane: utterance r, classes o = 16 m + 4 q + reg
    float mx = -3.0e38f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            if (m * 16 + q * 4 + reg < a.nc) mx = fmaxf(mx, acc[m][reg]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float e[MT][4];
    float se = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const bool cls = m * 16 + q * 4 + reg < a.nc;
            e[m][reg] = cls ? expf(acc[m][reg] - mx) : 0.f;
            se += e[m][reg];
        }
    se += __shfl_xor(se, 16);
    se += __shfl_xor(se, 32);
    const float lse = logf(se);
    float part = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int o = m * 16 + q * 4 + reg;
            const float v = acc[m][reg];
            if (o < a.nc) {
                const float p = e[m][reg] / se;
                if (uv) {
                    a.logits[(size_t)n * a.nc + o] = v;
                    a.probs[(size_t)n * a.nc + o] = p;
                }
                if (TRAIN) {
                    float y = a.labels[(size_t)n * a.nc + o];
                    if (a.label_smoothing > 0.f) y = y * (1.0f - a.label_smoothing) + a.label_smoothing / (float)a.nc;
                    part -= y * (v - mx - lse);
                    if (uv) a.dlogits[(size_t)n * a.nc + o] = (p - y) * a.inv_global_batch;
                }
            } else if (o < a.nc + 2) {
                if (uv && a.ranges) a.ranges[(size_t)n * 2 + (o - a.nc)] = 1.0f / (1.0f + expf(-v));   // tc_resnet.py:52
            }
        }
    if (TRAIN) {
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (uv && q == 0) a.loss_utt[n] = part;
    }
}

int launch_head_fwd(const HeadArgs& a, bool train, hipStream_t s) {
    const int mt = ceil_div(a.nc + 2, 16);
    if (mt > 3) { set_error("head: num_classes %d exceeds the 46-class limit of this build", a.nc); return TCR_ERR_ARG; }
    const dim3 grid(ceil_div(a.batch, 64));
#define TCR_H(MT_)                                                                                  \
    if (train) hipLaunchKernelGGL((head_fwd_kernel<MT_, true>), grid, dim3(256), 0, s, a);          \
    else hipLaunchKernelGGL((head_fwd_kernel<MT_, false>), grid, dim3(256), 0, s, a)
    if (mt == 1) { TCR_H(1); }
    else if (mt == 2) { TCR_H(2); }
    else { TCR_H(3); }
#undef TCR_H
    return check_launch("head_fwd_kernel");
}

// dpool[b][c] = (sum_o dlogits[b][o] * Wfc[c][o]) * dscale[b][c]   (fc dgrad + dropout + avg-pool backward)
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ wfc,
                                                       const float* __restrict__ dscale, float* __restrict__ dpool,
                                                       int batch, int c, int nc) {
    const int64_t total = (int64_t)batch * c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % c);
        const int64_t b = i / c;
        float s = 0.f;
        for (int o = 0; o < nc; ++o) s = fmaf(dlogits[b * nc + o], wfc[(size_t)ch * nc + o], s);
        dpool[i] = s * dscale[i];
    }
}

int launch_head_bwd(const float* dlogits, const float* wfc, const float* dscale, float* dpool, int batch, int c, int nc, hipStream_t s) {
    int64_t blocks = ceil_div64((int64_t)batch * c, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dlogits, wfc, dscale, dpool, batch, c, nc);
    return check_launch("head_bwd_kernel");
}

// partial[chunk][c][o] = sum_{b in chunk} dropped[b][c] * dlogits[b][o]
__global__ __launch_bounds__(256) void fc_wgrad_kernel(const float* __restrict__ dropped, const float* __restrict__ dlogits,
                                                       float* __restrict__ partial, int batch, int c, int nc, int per_block) {
    const int b0 = blockIdx.x * per_block, b1 = min(b0 + per_block, batch);
    for (int i = threadIdx.x; i < c * nc; i += 256) {
        const int ch = i / nc, o = i % nc;
        float s = 0.f;
        int b = b0;
        for (; b + 8 <= b1; b += 8) {       // eight utterances' operands in flight, added in utterance order (one dependent load pair per
            float xv[8], dv[8];             // utterance was 64 L2 round trips per output: 72 us for 2.4 MFLOP)
#pragma unroll
            for (int u = 0; u < 8; ++u) { xv[u] = dropped[(size_t)(b + u) * c + ch]; dv[u] = dlogits[(size_t)(b + u) * nc + o]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) s = fmaf(xv[u], dv[u], s);
        }
        for (; b < b1; ++b) s = fmaf(dropped[(size_t)b * c + ch], dlogits[(size_t)b * nc + o], s);
        partial[(size_t)blockIdx.x * c * nc + i] = s;
    }
}

// out[i] = sum_k partial[k][i]  (double accumulation, fixed order)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int nchunk, int n, float* __restrict__ out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        double s = 0.0;
        int k = 0;
        for (; k + 8 <= nchunk; k += 8) {       // eight rows in flight, added in row order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(k + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; k < nchunk; ++k) s += (double)partial[(size_t)k * n + i];
        out[i] = (float)s;
    }
}

int fc_wgrad_chunks(int batch) {
    int n = ceil_div(batch, 64);
    if (n > 64) n = 64;
    return n < 1 ? 1 : n;
}

int launch_fc_wgrad(const float* dropped, const float* dlogits, float* partial, float* dw, int batch, int c, int nc, hipStream_t s) {
    const int nchunk = fc_wgrad_chunks(batch);
    const int per_block = ceil_div(batch, nchunk);
    const int grid = ceil_div(batch, per_block);
    hipLaunchKernelGGL(fc_wgrad_kernel, dim3(grid), dim3(256), 0, s, dropped, dlogits, partial, batch, c, nc, per_block);
    TCR_TRY(check_launch("fc_wgrad_kernel"));
    hipLaunchKernelGGL(sum_partials_kernel, dim3(ceil_div(c * nc, 256)), dim3(256), 0, s, (const float*)partial, grid, c * nc, dw);
    return check_launch("sum_partials_kernel");
}

// db[o] = sum_b dlogits[b][o]   (bias gradient of slim.fully_connected, ds_cnn.py:99; single workgroup, double
// accumulation: 16 interleaved batch slices per class, added in a fixed order)
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dlogits, int batch, int nc, float* __restrict__ db) {
    __shared__ double s_part[16][64];
    for (int o0 = 0; o0 < nc; o0 += 16) {
        const int o = o0 + (threadIdx.x & 15), part = threadIdx.x >> 4;
        double s = 0.0;
        if (o < nc)
            for (int b = part; b < batch; b += 16) s += (double)dlogits[(size_t)b * nc + o];
        s_part[part][threadIdx.x & 15] = s;
        __syncthreads();
        if (threadIdx.x < 16 && o0 + threadIdx.x < nc) {
            double tot = 0.0;
            for (int k = 0; k < 16; ++k) tot += s_part[k][threadIdx.x];
            db[o0 + threadIdx.x] = (float)tot;
        }
        __syncthreads();
    }
}

int launch_bias_grad(const float* dlogits, int batch, int nc, float* db, hipStream_t s) {
    hipLaunchKernelGGL(bias_grad_kernel, dim3(1), dim3(256), 0, s, dlogits, batch, nc, db);
    return check_launch("bias_grad_kernel");
}

// out[0] = sum_i in[i]   (single workgroup, double accumulation, fixed order)
__global__ __launch_bounds__(256) void sum_vector_kernel(const float* __restrict__ in, int n, float* __restrict__ out) {
    __shared__ double s_part[256];
    double s = 0.0;
    int i = threadIdx.x;
    for (; i + 15 * 256 < n; i += 16 * 256) {           // sixteen loads in flight, added in order (a load per trip: 16 round trips at batch 4096)
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = in[i + u * 256];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += (double)v[u];
    }
    for (; i < n; i += 256) s += (double)in[i];
    s_part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < 256; ++i) tot += s_part[i];
        out[0] = (float)tot;
    }
}

// per-utterance softmax cross-entropy from logits rows [B][NC] (eval builds: tf.losses.softmax_cross_entropy, factory/audio_nets.py:168-173)
__global__ __launch_bounds__(64) void xent_rows_kernel(const float* __restrict__ logits, const float* __restrict__ labels, float* __restrict__ loss_utt,
                                                       int batch, int nc, float label_smoothing) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= batch) return;
    const float* z = logits + (size_t)n * nc;
    float mx = z[0];
    for (int k = 1; k < nc; ++k) mx = fmaxf(mx, z[k]);
    float se = 0.f;
    for (int k = 0; k < nc; ++k) se += expf(z[k] - mx);
    const float lse = logf(se);
    float loss = 0.f;
    for (int k = 0; k < nc; ++k) {
        float y = labels[(size_t)n * nc + k];
        if (label_smoothing > 0.f) y = y * (1.0f - label_smoothing) + label_smoothing / (float)nc;
        loss -= y * ((z[k] - mx) - lse);
    }
    loss_utt[n] = loss;
}

int launch_sum_vector(const float* in, int n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(sum_vector_kernel, dim3(1), dim3(256), 0, s, in, n, out);
    return check_launch("sum_vector_kernel");
}

}  // namespace tcr

// Sum over the batch of the softmax cross-entropy of logits rows against (optionally smoothed) one-hot labels: the evaluation graph's
// model loss (factory/audio_nets.py:161-173) times the batch.  loss_utt: caller-owned scratch of `batch` floats; loss_sum: one float.
extern "C" int tcr_xent_loss_sum(const float* logits, const float* labels, int batch, int num_classes, float label_smoothing,
                                 float* loss_utt, float* loss_sum, void* stream) {
    TCR_REQUIRE(logits && labels && loss_utt && loss_sum && batch > 0 && num_classes > 0, "tcr_xent_loss_sum: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(tcr::xent_rows_kernel, dim3(tcr::ceil_div(batch, 64)), dim3(64), 0, s, logits, labels, loss_utt, batch, num_classes, label_smoothing);
    TCR_TRY(tcr::check_launch("xent_rows_kernel"));
    return tcr::launch_sum_vector(loss_utt, batch, loss_sum, s);
}
