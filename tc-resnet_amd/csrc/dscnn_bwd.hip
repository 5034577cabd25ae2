// Backward kernels of DS-CNN (audio_nets/ds_cnn.py:46-101 under tf.gradients, helper/trainer.py:205-211):
//   * depthwise 3x3: data gradient (transposed stencil) and filter gradient (9 sums per channel over batch x map)
//   * conv_1 (10x4, one input channel): filter gradient as a [taps] x [positions] x [Cout] contraction on the
//     exact-f32 MFMA -- the reduction index (positions) is the MFMA k dimension.
// The pointwise 1x1 convolutions reuse the matrix-core kernels of mfma.hip (data gradient = 1x1 conv with the
// transposed weight, filter gradient = conv_wgrad_mfma_kernel<1, .>); BN reuses bn.hip.
// Every cross-workgroup sum goes through per-workgroup partial slabs added in a FIXED order (bitwise reproducible).
#include "kernels.h"

namespace tcr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// dx[h][w] = sum_{di,dj} W[di][dj] * dz[(h + pad_t - di) / sh][(w + pad_l - dj) / sw]  over taps that divide evenly.
// One wavefront per (utterance, channel) plane; the 9 taps are wave-uniform.
// SH / SW: compile-time strides (1 or 2: the divisions below become shifts), or 0 for the run-time values.
template <int SH, int SW>
__global__ __launch_bounds__(256) void dscnn_dw_dgrad_kernel(const DsDwBwdArgs a) {
    const int sh = SH ? SH : a.sh, sw = SW ? SW : a.sw;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.planes) return;
    const int c = (int)(row % a.c);
    float wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = a.w[(size_t)k * a.c + c];
    const float* dz = a.dz + row * a.ppo + kHalo;
    float* dx = a.dx + row * a.ppi + kHalo;
    const int pin = a.h_in * a.w_in;
    const float inv_w = 1.0f / (float)a.w_in;
    for (int pos = lane; pos < pin; pos += 64) {
        const int h = fast_div(pos, a.w_in, inv_w), w = pos - h * a.w_in;
        float s = 0.f;
#pragma unroll
        for (int di = 0; di < 3; ++di) {
            const int hh = h + a.pad_t - di;
            const int oh = hh / sh;
            const bool hv = hh >= 0 && oh * sh == hh && oh < a.oh;
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                const int ww = w + a.pad_l - dj;
                const int ow = ww / sw;
                const bool v = hv && ww >= 0 && ow * sw == ww && ow < a.ow;
                const float g = v ? dz[oh * a.ow + ow] : 0.f;
                s = fmaf(wt[di * 3 + dj], g, s);
            }
        }
        dx[pos] = s;
    }
}

// LDS-staged form (as dscnn_depthwise_lds_kernel): a workgroup owns 16 consecutive planes and builds, per plane, the
// zero-UPSAMPLED, zero-padded image U[(h_in + 2) x (w_in + 2)] of dz -- U[a][b] = dz[hh / sh][ww / sw] for hh = a - (2 - pad_t),
// ww = b - (2 - pad_l) when both divide evenly and land inside the output map, else 0 -- with batches of independent clamped
// loads; then dx[h][w] = sum_{di,dj} W[di][dj] * U[h + 2 - di][w + 2 - dj] with no predicates.  (The wave-per-plane kernel
// above issues nine predicated global loads per lane: 1.1 TB/s at DS-CNN-L sizes.)
template <int SH, int SW>
__global__ __launch_bounds__(256) void dscnn_dw_dgrad_lds_kernel(const DsDwBwdArgs a) {
    float* img = reinterpret_cast<float*>(dyn_lds());               // [16][h_in + 2][w_in + 2]
    const int sh = SH ? SH : a.sh, sw = SW ? SW : a.sw;
    const int plane = threadIdx.x >> 4, t16 = threadIdx.x & 15;
    const int64_t row_raw = (int64_t)blockIdx.x * 16 + plane;
    const int64_t row = row_raw < a.planes ? row_raw : a.planes - 1;
    const bool live = row_raw < a.planes;
    const int c = (int)(row % a.c);
    const int ir = a.h_in + 2, ic = a.w_in + 2, isz = ir * ic;
    const bool fly = a.fly.da != nullptr;
    const float* dz = (fly ? a.fly.da : a.dz) + row * a.ppo + kHalo;
    const float* rz = fly ? a.fly.raw + row * a.ppo + kHalo : dz;
    const float f_k1 = fly ? a.fly.k1[c] : 0.f, f_k2 = fly ? a.fly.k2[c] : 0.f, f_k3 = fly ? a.fly.k3[c] : 0.f, f_mu = fly ? a.fly.mean[c] : 0.f;
    const float f_sc = fly ? a.fly.self_scale[c] : 0.f, f_sh = fly ? a.fly.self_shift[c] : 0.f;
    float* im = img + plane * isz;
    const float inv_c = 1.0f / (float)ic;
    for (int j0 = t16; j0 < isz; j0 += 16 * 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = min(j0 + 16 * i, isz - 1);
            const int rr = fast_div(j, ic, inv_c), cc = j - rr * ic;
            const int hh = rr - (2 - a.pad_t), ww = cc - (2 - a.pad_l);
            const int oh = hh / sh, ow = ww / sw;
            const bool in = hh >= 0 && ww >= 0 && oh * sh == hh && ow * sw == ww && oh < a.oh && ow < a.ow;
            float g = dz[in ? oh * a.ow + ow : 0];
            if (fly) {          // BN backward of this unit where dy is read (BnBwdFly)
                const float yv = rz[in ? oh * a.ow + ow : 0];
                if (!(fmaf(yv, f_sc, f_sh) > 0.f)) g = 0.f;
                g = f_k1 * (g - f_k2 - (yv - f_mu) * f_k3);
            }
            v[i] = in ? g : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (j0 + 16 * i < isz) im[j0 + 16 * i] = v[i];
    }
    float wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = a.w[(size_t)k * a.c + c];
    __syncthreads();
    if (!live) return;
    const int pin = a.h_in * a.w_in;
    const float inv_w = 1.0f / (float)a.w_in;
    float* dx = a.dx + row * a.ppi + kHalo;
    // EpiSums (backward form): dx is the gradient wrt the activation of the BN unit that feeds this depthwise conv; its sum dz,
    // sum dz * xhat are taken here, from that unit's raw output at the same addresses -- one partial row per utterance
    const bool sums = a.sums.partial != nullptr;
    const float* rawr = sums ? a.sums.raw + row * a.ppi + kHalo : nullptr;
    const float mu = sums ? a.sums.mean[c] : 0.f, is = sums ? a.sums.invstd[c] : 0.f;
    const float ssc = sums ? a.sums.self_scale[c] : 0.f, ssh = sums ? a.sums.self_shift[c] : 0.f;
    float q1 = 0.f, q2 = 0.f;
    for (int pos0 = t16; pos0 < pin; pos0 += 16 * 5) {
        float s[5], rw[5];
        if (sums) {
#pragma unroll
            for (int i = 0; i < 5; ++i) rw[i] = rawr[min(pos0 + 16 * i, pin - 1)];
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int pos = min(pos0 + 16 * i, pin - 1);
            const int h = fast_div(pos, a.w_in, inv_w), w = pos - h * a.w_in;
            const float* p0 = im + (h + 2) * ic + (w + 2);
            s[i] = 0.f;
#pragma unroll
            for (int di = 0; di < 3; ++di)
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) s[i] = fmaf(wt[di * 3 + dj], p0[-di * ic - dj], s[i]);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i)
            if (pos0 + 16 * i < pin) {
                dx[pos0 + 16 * i] = s[i];
                if (sums) {
                    const float dz = fmaf(rw[i], ssc, ssh) > 0.f ? s[i] : 0.f;
                    q1 += dz;
                    q2 = fmaf(dz, (rw[i] - mu) * is, q2);
                }
            }
    }
    if (sums) {                 // (a plane's 16 lanes are one DPP row; `live` is uniform over it)
        q1 = row16_sum(q1);
        q2 = row16_sum(q2);
        if (t16 == 0) {
            const size_t n = (size_t)(row / a.c);
            a.sums.partial[(n * 2 + 0) * a.c + c] = q1;
            a.sums.partial[(n * 2 + 1) * a.c + c] = q2;
        }
    }
}

bool dscnn_dw_dgrad_lds_covers(int h_in, int w_in, int pad_t, int pad_l) {
    return (size_t)16 * (h_in + 2) * (w_in + 2) * sizeof(float) <= 64 * 1024 && pad_t <= 2 && pad_l <= 2;
}

int launch_dscnn_dw_dgrad(const DsDwBwdArgs& a, hipStream_t s) {
    const size_t lds = (size_t)16 * (a.h_in + 2) * (a.w_in + 2) * sizeof(float);
    if (dscnn_dw_dgrad_lds_covers(a.h_in, a.w_in, a.pad_t, a.pad_l)) {
        const dim3 lgrid((unsigned)ceil_div64(a.planes, 16));
        if (a.sh == 1 && a.sw == 1) hipLaunchKernelGGL((dscnn_dw_dgrad_lds_kernel<1, 1>), lgrid, dim3(256), lds, s, a);
        else if (a.sh == 2 && a.sw == 2) hipLaunchKernelGGL((dscnn_dw_dgrad_lds_kernel<2, 2>), lgrid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((dscnn_dw_dgrad_lds_kernel<0, 0>), lgrid, dim3(256), lds, s, a);
        return check_launch("dscnn_dw_dgrad_lds_kernel");
    }
    if (a.sums.partial || a.fly.da) { set_error("dscnn depthwise data gradient: epilogue sums / on-the-fly BN backward need the LDS kernel"); return TCR_ERR_ARG; }
    const dim3 grid((unsigned)ceil_div64(a.planes, 4));
    if (a.sh == 1 && a.sw == 1) hipLaunchKernelGGL((dscnn_dw_dgrad_kernel<1, 1>), grid, dim3(256), 0, s, a);
    else if (a.sh == 2 && a.sw == 2) hipLaunchKernelGGL((dscnn_dw_dgrad_kernel<2, 2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dscnn_dw_dgrad_kernel<0, 0>), grid, dim3(256), 0, s, a);
    return check_launch("dscnn_dw_dgrad_kernel");
}

// partial[chunk][tap][c] = sum_{n in chunk, oh, ow} x[n][c][oh*sh + di - pad_t][ow*sw + dj - pad_l] * dz[n][c][oh][ow]
// One wavefront per (chunk of utterances, channel); lanes walk the flattened (utterance, position) index.
__global__ __launch_bounds__(256) void dscnn_dw_wgrad_kernel(const DsDwWgradArgs a) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= a.c) return;
    const int n0 = blockIdx.x * a.utt_per_block;
    const int cnt = min(a.utt_per_block, a.batch - n0);
    const int P = a.oh * a.ow;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    const float inv_p = 1.0f / (float)P, inv_ow = 1.0f / (float)a.ow;
    const bool aff = a.x_scale != nullptr;
    const float xsc = aff ? a.x_scale[c] : 1.0f, xsf = aff ? a.x_shift[c] : 0.f;
    const bool fly = a.fly.da != nullptr;
    const float* gsrc = fly ? a.fly.da : a.dz;
    const float f_k1 = fly ? a.fly.k1[c] : 0.f, f_k2 = fly ? a.fly.k2[c] : 0.f, f_k3 = fly ? a.fly.k3[c] : 0.f, f_mu = fly ? a.fly.mean[c] : 0.f;
    const float f_sc = fly ? a.fly.self_scale[c] : 0.f, f_sh = fly ? a.fly.self_shift[c] : 0.f;
    for (int idx = lane; idx < cnt * P; idx += 64) {
        const int dn = fast_div(idx, P, inv_p), pos = idx - dn * P;
        const int oh = fast_div(pos, a.ow, inv_ow), ow = pos - oh * a.ow;
        const size_t plane = (size_t)(n0 + dn) * a.c + c;
        float g = gsrc[plane * a.ppo + kHalo + pos];
        if (fly) {              // BN backward of this unit where dy is read (BnBwdFly)
            const float yv = a.fly.raw[plane * a.ppo + kHalo + pos];
            if (!(fmaf(yv, f_sc, f_sh) > 0.f)) g = 0.f;
            g = f_k1 * (g - f_k2 - (yv - f_mu) * f_k3);
        }
        const float* xr = a.x + plane * a.ppi + kHalo;
#pragma unroll
        for (int di = 0; di < 3; ++di) {
            const int h = oh * a.sh + di - a.pad_t;
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                const int w = ow * a.sw + dj - a.pad_l;
                const bool in = h >= 0 && h < a.h_in && w >= 0 && w < a.w_in;
                float xv = in ? xr[h * a.w_in + w] : 0.f;
                if (aff) xv = in ? fmaxf(fmaf(xv, xsc, xsf), 0.f) : 0.f;
                acc[di * 3 + dj] = fmaf(xv, g, acc[di * 3 + dj]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float v = wave_sum(acc[k]);
        if (lane == 0) a.partial[((size_t)blockIdx.x * 9 + k) * a.c + c] = v;
    }
}

static int dw_wgrad_chunks(int batch) {
    int n = ceil_div(batch, 32);
    if (n > 128) n = 128;
    return n < 1 ? 1 : n;
}

size_t dscnn_dw_wgrad_partial_floats(int batch, int c) { return (size_t)dw_wgrad_chunks(batch) * 9 * c; }

int launch_dscnn_dw_wgrad(DsDwWgradArgs a, float* dw, hipStream_t s) {
    a.utt_per_block = ceil_div(a.batch, dw_wgrad_chunks(a.batch));
    const dim3 grid(ceil_div(a.batch, a.utt_per_block), ceil_div(a.c, 4));
    hipLaunchKernelGGL(dscnn_dw_wgrad_kernel, grid, dim3(256), 0, s, a);
    TCR_TRY(check_launch("dscnn_dw_wgrad_kernel"));
    // dw[tap][c] = sum_chunk partial[chunk][tap][c]: the reduction kernel of mfma.hip with k = 9 taps, Cin = 1, Cout = C
    return launch_wgrad_reduce(a.partial, dw, (int)grid.x, 9, 1, a.c, 1, a.c, a.c, 0, s);
}

// conv_1 filter gradient: dW[i][j][co] = sum_{n,oh,ow} feat[n][ow*sw + j - pad_l][oh*sh + i - pad_t] * dz[n][co][oh][ow].
// A = feature patches (tap tile x 4 positions), B = dz (4 positions x 16 output channels), D = [tap][co].  A wave owns all tap
// tiles (3 x 16 >= 10 x 4) and NCO channel tiles, so dz -- the large operand -- is read exactly once; workgroups split the batch
// (split-K).  The MFMA k dimension holds positions 4 q + c of a 16-position block in step c, so a lane's four k-steps take ONE
// 16-byte load per channel row (16 rows x 64 contiguous bytes per instruction; the first version gathered 16 rows x 16 bytes per
// k-step and was bound by the L1's address processing: 1.0 ms for 1.2 GB), and the utterance's small feature map is staged in a
// wave-private LDS copy, so the 12 patch gathers per block are ds_reads.  (A version that staged the channel rows of an utterance
// in LDS as one contiguous block -- two 33 KB blocks, two workgroups per CU, a barrier pair per utterance -- moved the minimum 2.4 GB
// but took 1.49 ms against 1.14: too few waves to cover its load -> barrier -> compute chain, 4-way bank conflicts on rows 258 floats apart.)  BnBwdFly: dz is computed here from the activation
// gradient and the unit's raw output (conv_1's dy has no other reader: no bn_bwd_apply pass, 3.6 GB less per step).
struct __attribute__((packed, aligned(4))) f32x4u { float v[4]; };

template <int NCO>
__global__ __launch_bounds__(256) void dscnn_conv1_wgrad_kernel(const DsConv1WgradArgs a) {
    constexpr int MT = 3;
    __shared__ float s_acc[MT * 16 * NCO * 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int co0 = blockIdx.y * NCO * 16;
    const int fsz = a.w_in * a.tp_in;
    float* sf = reinterpret_cast<float*>(dyn_lds()) + wave * fsz;      // [w_in][tp_in] of the wave's current utterance

    f32x4 acc[MT][NCO];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int m = 0; m < NCO; ++m) acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int ti[MT], tj[MT];
    bool tv[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int tap = t * 16 + r;
        tv[t] = tap < a.taps;
        ti[t] = tap >> 2;               // kernel row  (kw == 4)
        tj[t] = tap & 3;                // kernel column
    }
    const bool fly = a.fly.da != nullptr;
    bool cov[NCO];
    size_t crow[NCO];
    float f_k1[NCO], f_k2[NCO], f_k3[NCO], f_mu[NCO], f_sc[NCO], f_sh[NCO];
#pragma unroll
    for (int m = 0; m < NCO; ++m) {
        const int co = co0 + m * 16 + r;
        cov[m] = co < a.cout;
        const int cc = cov[m] ? co : 0;
        crow[m] = (size_t)cc * a.pp + kHalo;
        f_k1[m] = fly ? a.fly.k1[cc] : 0.f; f_k2[m] = fly ? a.fly.k2[cc] : 0.f; f_k3[m] = fly ? a.fly.k3[cc] : 0.f;
        f_mu[m] = fly ? a.fly.mean[cc] : 0.f; f_sc[m] = fly ? a.fly.self_scale[cc] : 0.f; f_sh[m] = fly ? a.fly.self_shift[cc] : 0.f;
    }
    const int P = a.oh * a.ow;
    const float inv_ow = 1.0f / (float)a.ow;
    const int n_begin = blockIdx.x * a.utt_per_block;
    const int n_end = min(n_begin + a.utt_per_block, a.batch);
    for (int n = n_begin + wave; n < n_end; n += 4) {
        wave_sync();                                                    // (the previous utterance's gathers are done)
        const float* fr = a.feat + (size_t)n * fsz;
        for (int i = lane; i < fsz; i += 64) sf[i] = fr[i];
        wave_sync();
        const size_t nb = (size_t)n * a.cout * a.pp;
        const float* dr = (fly ? a.fly.da : a.dz) + nb;
        const float* rr = fly ? a.fly.raw + nb : dr;
        for (int p0 = 0; p0 < P; p0 += 64) {
            // B operands: FOUR 16-position blocks per trip -- block i holds positions p0 + 16 i + 4 q + c --, each one (unaligned)
            // 16-byte load per channel row, issued back to back: the four loads of a row consume whole 128-byte lines at once.
            // (One block per trip left every line half-used until the next trip; with ~18 waves x 64 rows live per CU the lines were
            // evicted in between: 5.9 GB of HBM traffic for 2.4 GB of tensors.)  Positions past the map are masked below (the loads
            // stay inside the workspace: rows carry halos, the gradient buffers a tail pad).
            f32x4u d4[NCO][4], y4[NCO][4];
#pragma unroll
            for (int m = 0; m < NCO; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pc = min(p0 + 16 * i + 4 * q, P - 1);       // (a whole block past the map: any valid address, masked below)
                    d4[m][i] = *reinterpret_cast<const f32x4u*>(dr + crow[m] + pc);
                    if (fly) y4[m][i] = *reinterpret_cast<const f32x4u*>(rr + crow[m] + pc);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pb = p0 + 16 * i + 4 * q;
                if (p0 + 16 * i >= P) break;                              // (wave-uniform)
                float bv[NCO][4];
#pragma unroll
                for (int m = 0; m < NCO; ++m) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float g = d4[m][i].v[c];
                        if (fly) {
                            const float yv = y4[m][i].v[c];
                            if (!(fmaf(yv, f_sc[m], f_sh[m]) > 0.f)) g = 0.f;
                            g = f_k1[m] * (g - f_k2[m] - (yv - f_mu[m]) * f_k3[m]);
                        }
                        bv[m][c] = (cov[m] && pb + c < P) ? g : 0.f;
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int p = min(pb + c, P - 1);
                    const bool pv = pb + c < P;
                    const int oh = fast_div(p, a.ow, inv_ow), ow = p - oh * a.ow;
                    const int hb = oh * a.sh - a.pad_t, wb = ow * a.sw - a.pad_l;
                    float af[MT];
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        const int h = hb + ti[t], wc = wb + tj[t];
                        const bool v = pv && tv[t] && h >= 0 && h < a.h_in && wc >= 0 && wc < a.w_in;
                        af[t] = v ? sf[wc * a.tp_in + kHalo + h] : 0.f;
                    }
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int m = 0; m < NCO; ++m) acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t], bv[m][c], acc[t][m], 0, 0, 0);
                }
            }
        }
    }
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int m = 0; m < NCO; ++m)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int idx = ((t * 16 + q * 4 + reg) * NCO + m) * 16 + r;
                        if (wv == 0) s_acc[idx] = acc[t][m][reg];
                        else s_acc[idx] += acc[t][m][reg];
                    }
        }
        __syncthreads();
    }
    float* dst = a.partial + (size_t)blockIdx.x * a.taps_pad * a.cout_pad;
    for (int i = threadIdx.x; i < MT * 16 * NCO * 16; i += 256) {
        const int col = i % (NCO * 16), tap = i / (NCO * 16);
        if (tap < a.taps_pad && co0 + col < a.cout_pad) dst[(size_t)tap * a.cout_pad + co0 + col] = s_acc[i];
    }
}

size_t dscnn_conv1_wgrad_partial_floats(int batch, int kh, int cout) {
    return (size_t)wgrad_chunks(batch) * (size_t)(ceil_div(kh * 4, 16) * 16) * (size_t)(ceil_div(cout, 16) * 16);
}

int launch_dscnn_conv1_wgrad(DsConv1WgradArgs a, float* dw, hipStream_t s) {
    a.taps = a.kh * 4;
    a.taps_pad = ceil_div(a.taps, 16) * 16;
    a.cout_pad = ceil_div(a.cout, 16) * 16;
    if (a.taps_pad > 48) { set_error("conv_1 wgrad: %d x 4 kernel exceeds the 48-tap tile", a.kh); return TCR_ERR_ARG; }
    a.utt_per_block = ceil_div(a.batch, wgrad_chunks(a.batch));
    const dim3 grid(ceil_div(a.batch, a.utt_per_block), ceil_div(a.cout_pad / 16, 2));
    const size_t lds = (size_t)4 * a.w_in * a.tp_in * sizeof(float);     // a wave-private copy of its utterance's feature map
    if (lds > 48 * 1024) { set_error("conv_1 wgrad: a %d x %d feature map does not fit the LDS copies", a.w_in, a.tp_in); return TCR_ERR_ARG; }
    hipLaunchKernelGGL((dscnn_conv1_wgrad_kernel<2>), grid, dim3(256), lds, s, a);
    TCR_TRY(check_launch("dscnn_conv1_wgrad_kernel"));
    // dW[tap][0][co]: k = 1, "Cin" = taps
    return launch_wgrad_reduce(a.partial, dw, (int)grid.x, 1, a.taps, a.cout, a.taps_pad, a.cout_pad, a.cout, 0, s);
}

}  // namespace tcr
