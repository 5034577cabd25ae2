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

// Stride-1 layers of the 13 x 5 maps (four of DS-CNN's five depthwise units), every byte through HBM as a contiguous float4: a workgroup
// owns 16 consecutive planes = ONE contiguous block of 16 x Pp floats of dz, of the raw output (EpiSums) and of dx.  dz and raw are copied
// to LDS as they lie (292 float4, no index arithmetic); lane (plane, h) then computes the W outputs of map row h from the three dz rows
// around it (15 LDS reads for 5 outputs; the image kernel above reads 9 per output and spends ~30 VALU instructions per staged element
// on divisions and predicates) and writes them to an LDS copy of the dx block, which leaves as float4.  The backward sums are taken from
// the LDS copies by 16 lanes per plane in the order of the image kernel (positions t16, t16 + 16, ...; row16_sum): bitwise the same
// partial rows, and the same fmaf chain per output (out-of-map taps multiply a zero).
template <int H, int W>
__global__ __launch_bounds__(256) void dscnn_dw_dgrad_rows_kernel(const DsDwBwdArgs a) {
    constexpr int NPL = 16, P = H * W;
    static_assert(NPL * H <= 256, "one lane per (plane, map row)");
    const int pp = a.ppo;                                       // (== ppi)
    const int blk = NPL * pp, n4 = blk / 4;
    float* s_dz = reinterpret_cast<float*>(dyn_lds());
    float* s_raw = s_dz + blk;
    float* s_dx = s_raw + blk;
    const int tid = threadIdx.x;
    const bool sums = a.sums.partial != nullptr;
    const size_t g0 = (size_t)blockIdx.x * blk;
    const f32x4* dz4 = reinterpret_cast<const f32x4*>(a.dz + g0);
    const f32x4* rw4 = reinterpret_cast<const f32x4*>(sums ? a.sums.raw + g0 : a.dz + g0);
    const bool two = tid + 256 < n4;
    const f32x4 z0 = dz4[tid], z1 = dz4[two ? tid + 256 : tid];
    f32x4 r0 = z0, r1 = z1;
    if (sums) { r0 = rw4[tid]; r1 = rw4[two ? tid + 256 : tid]; }
    // channel of the workgroup's first plane (wave-uniform), then per lane
    const int row0 = blockIdx.x * NPL;
    const int n0 = row0 / a.c, c0 = row0 - n0 * a.c;
    const int pl3 = tid / H, h = tid - pl3 * H;                 // the compute phase's (plane, map row)
    const bool act3 = pl3 < NPL;
    int c3 = c0 + (act3 ? pl3 : 0);
    if (c3 >= a.c) c3 -= a.c;
    float wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = a.w[(size_t)k * a.c + c3];
    reinterpret_cast<f32x4*>(s_dz)[tid] = z0;
    if (two) reinterpret_cast<f32x4*>(s_dz)[tid + 256] = z1;
    if (sums) {
        reinterpret_cast<f32x4*>(s_raw)[tid] = r0;
        if (two) reinterpret_cast<f32x4*>(s_raw)[tid + 256] = r1;
    }
    if (tid < NPL * 2 * kHalo) {                                // the halo floats of the dx block leave as zeros
        const int pl = tid / (2 * kHalo), k = tid - pl * (2 * kHalo);
        s_dx[pl * pp + (k < kHalo ? k : P + k)] = 0.f;
    }
    __syncthreads();
    if (act3) {
        float in[3][W + 2];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {                        // rr: map row h + 1 - rr (tap row di = rr)
            const int hh = h + 1 - rr;
            const bool hv = hh >= 0 && hh < H;
            const float* src = s_dz + pl3 * pp + kHalo + (hv ? hh : h) * W;
            in[rr][0] = 0.f; in[rr][W + 1] = 0.f;
#pragma unroll
            for (int w = 0; w < W; ++w) { const float v = src[w]; in[rr][w + 1] = hv ? v : 0.f; }
        }
        float* dst = s_dx + pl3 * pp + kHalo + h * W;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            float sacc = 0.f;
#pragma unroll
            for (int di = 0; di < 3; ++di)
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) sacc = fmaf(wt[di * 3 + dj], in[di][w + 2 - dj], sacc);      // dz[h + 1 - di][w + 1 - dj]
            dst[w] = sacc;
        }
    }
    __syncthreads();
    f32x4* dx4 = reinterpret_cast<f32x4*>(a.dx + g0);
    dx4[tid] = reinterpret_cast<const f32x4*>(s_dx)[tid];
    if (two) dx4[tid + 256] = reinterpret_cast<const f32x4*>(s_dx)[tid + 256];
    if (sums) {
        const int plane = tid >> 4, t16 = tid & 15;
        int c = c0 + plane, n = n0;
        if (c >= a.c) { c -= a.c; ++n; }
        const float mu = a.sums.mean[c], is = a.sums.invstd[c], ssc = a.sums.self_scale[c], ssh = a.sums.self_shift[c];
        const float* sx = s_dx + plane * pp + kHalo;
        const float* sr = s_raw + plane * pp + kHalo;
        float q1 = 0.f, q2 = 0.f;
#pragma unroll
        for (int i = 0; i < (P + 15) / 16; ++i) {
            const int pos = t16 + 16 * i;
            if (pos < P) {
                const float rw = sr[pos];
                const float dzv = fmaf(rw, ssc, ssh) > 0.f ? sx[pos] : 0.f;
                q1 += dzv;
                q2 = fmaf(dzv, (rw - mu) * is, q2);
            }
        }
        q1 = row16_sum(q1);
        q2 = row16_sum(q2);
        if (t16 == 0) {
            a.sums.partial[((size_t)n * 2 + 0) * a.c + c] = q1;
            a.sums.partial[((size_t)n * 2 + 1) * a.c + c] = q2;
        }
    }
}

// The stride-2 unit (dx on the 25 x 10 map from dz on 13 x 5; TF SAME: pad_t 1, pad_l 0), same scheme with 8 planes per workgroup: the dz
// block (8 x 73 floats) and the raw block of the unit below (8 x 258) through LDS, lane (plane, h) computes the ten dx of map row h from
// the one or two dz rows that reach it (taps with (h + 1 - di) even; columns (w - dj) even: the taps that meet a zero of the upsampled
// image in the kernel above are left out -- fmaf(w, 0, s) = s), the dx block leaves as float4; sums in the image kernel's order.
template <int HI, int WI, int HO, int WO>
__global__ __launch_bounds__(256) void dscnn_dw_dgrad_rows_s2_kernel(const DsDwBwdArgs a) {
    constexpr int NPL = 8, PI = HI * WI, PO = HO * WO, PPI = PI + 2 * kHalo, PPO = PO + 2 * kHalo;
    static_assert(NPL * HI <= 256 && (NPL * PPI) % 4 == 0 && (NPL * PPO) % 4 == 0 && WI == 2 * WO && HI == 2 * HO - 1 && NPL * 16 <= 256, "geometry");
    constexpr int NX4 = NPL * PPI / 4, NZ4 = NPL * PPO / 4, XPT = (NX4 + 255) / 256;
    static_assert(NZ4 <= 256, "one dz float4 per thread");
    float* s_dz = reinterpret_cast<float*>(dyn_lds());
    float* s_raw = s_dz + NPL * PPO;
    float* s_dx = s_raw + NPL * PPI;
    const int tid = threadIdx.x;
    const bool sums = a.sums.partial != nullptr;
    const size_t gi = (size_t)blockIdx.x * (NPL * PPI), go = (size_t)blockIdx.x * (NPL * PPO);
    const f32x4 z0 = reinterpret_cast<const f32x4*>(a.dz + go)[min(tid, NZ4 - 1)];
    f32x4 rv[XPT];
    if (sums) {
#pragma unroll
        for (int j = 0; j < XPT; ++j) rv[j] = reinterpret_cast<const f32x4*>(a.sums.raw + gi)[min(tid + 256 * j, NX4 - 1)];
    }
    const int row0 = blockIdx.x * NPL;
    const int n0 = row0 / a.c, c0 = row0 - n0 * a.c;
    const int pl3 = tid / HI, h = tid - pl3 * HI;
    const bool act3 = pl3 < NPL;
    int c3 = c0 + (act3 ? pl3 : 0);
    if (c3 >= a.c) c3 -= a.c;
    float wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = a.w[(size_t)k * a.c + c3];
    if (tid < NZ4) reinterpret_cast<f32x4*>(s_dz)[tid] = z0;
    if (sums) {
#pragma unroll
        for (int j = 0; j < XPT; ++j)
            if (tid + 256 * j < NX4) reinterpret_cast<f32x4*>(s_raw)[tid + 256 * j] = rv[j];
    }
    if (tid < NPL * 2 * kHalo) {
        const int pl = tid / (2 * kHalo), k = tid - pl * (2 * kHalo);
        s_dx[pl * PPI + (k < kHalo ? k : PI + k)] = 0.f;
    }
    __syncthreads();
    if (act3) {
        float g[3][WO];
#pragma unroll
        for (int di = 0; di < 3; ++di) {
            const int hh = h + 1 - di;
            const int oh = hh >> 1;
            const bool hv = hh >= 0 && (hh & 1) == 0 && oh < HO;
            const float* src = s_dz + pl3 * PPO + kHalo + (hv ? oh : 0) * WO;
#pragma unroll
            for (int w = 0; w < WO; ++w) { const float v = src[w]; g[di][w] = hv ? v : 0.f; }
        }
        float* dst = s_dx + pl3 * PPI + kHalo + h * WI;
#pragma unroll
        for (int w = 0; w < WI; ++w) {
            float sacc = 0.f;
#pragma unroll
            for (int di = 0; di < 3; ++di)
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) {
                    const int ww = w - dj;                      // (compile-time)
                    if (ww >= 0 && (ww & 1) == 0 && ww / 2 < WO) sacc = fmaf(wt[di * 3 + dj], g[di][ww / 2], sacc);
                }
            dst[w] = sacc;
        }
    }
    __syncthreads();
    f32x4* dx4 = reinterpret_cast<f32x4*>(a.dx + gi);
#pragma unroll
    for (int j = 0; j < XPT; ++j)
        if (tid + 256 * j < NX4) dx4[tid + 256 * j] = reinterpret_cast<const f32x4*>(s_dx)[tid + 256 * j];
    if (sums && tid < NPL * 16) {
        const int plane = tid >> 4, t16 = tid & 15;
        int c = c0 + plane, n = n0;
        if (c >= a.c) { c -= a.c; ++n; }
        const float mu = a.sums.mean[c], is = a.sums.invstd[c], ssc = a.sums.self_scale[c], ssh = a.sums.self_shift[c];
        const float* sx = s_dx + plane * PPI + kHalo;
        const float* sr = s_raw + plane * PPI + kHalo;
        float q1 = 0.f, q2 = 0.f;
#pragma unroll
        for (int i = 0; i < (PI + 15) / 16; ++i) {
            const int pos = t16 + 16 * i;
            if (pos < PI) {
                const float rw = sr[pos];
                const float dzv = fmaf(rw, ssc, ssh) > 0.f ? sx[pos] : 0.f;
                q1 += dzv;
                q2 = fmaf(dzv, (rw - mu) * is, q2);
            }
        }
        q1 = row16_sum(q1);
        q2 = row16_sum(q2);
        if (t16 == 0) {
            a.sums.partial[((size_t)n * 2 + 0) * a.c + c] = q1;
            a.sums.partial[((size_t)n * 2 + 1) * a.c + c] = q2;
        }
    }
}

static bool dscnn_dw_dgrad_rows_s2_covers(const DsDwBwdArgs& a) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return a.sh == 2 && a.sw == 2 && a.pad_t == 1 && a.pad_l == 0 && a.h_in == 25 && a.w_in == 10 && a.oh == 13 && a.ow == 5 &&
           a.ppi == 25 * 10 + 2 * kHalo && a.ppo == 13 * 5 + 2 * kHalo && a.planes % 8 == 0 && a.planes / 8 < (int64_t)1 << 27 && a.c >= 8 && !a.fly.da &&
           al(a.dz) && al(a.dx) && (!a.sums.partial || al(a.sums.raw)) && tune_get(TCR_TUNE_DW_DGRAD) != 1;
}

static bool dscnn_dw_dgrad_rows_covers(const DsDwBwdArgs& a) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return a.sh == 1 && a.sw == 1 && a.pad_t == 1 && a.pad_l == 1 && a.h_in == 13 && a.w_in == 5 && a.oh == 13 && a.ow == 5 && a.ppi == a.ppo &&
           a.ppo == 13 * 5 + 2 * kHalo && a.planes % 16 == 0 && a.planes / 16 < (int64_t)1 << 27 && a.c >= 16 && !a.fly.da &&
           al(a.dz) && al(a.dx) && (!a.sums.partial || al(a.sums.raw)) && tune_get(TCR_TUNE_DW_DGRAD) != 1;
}

bool dscnn_dw_dgrad_lds_covers(int h_in, int w_in, int pad_t, int pad_l) {
    return (size_t)16 * (h_in + 2) * (w_in + 2) * sizeof(float) <= 64 * 1024 && pad_t <= 2 && pad_l <= 2;
}

int launch_dscnn_dw_dgrad(const DsDwBwdArgs& a, hipStream_t s) {
    const size_t lds = (size_t)16 * (a.h_in + 2) * (a.w_in + 2) * sizeof(float);
    if (dscnn_dw_dgrad_rows_covers(a)) {
        hipLaunchKernelGGL((dscnn_dw_dgrad_rows_kernel<13, 5>), dim3((unsigned)(a.planes / 16)), dim3(256), (size_t)3 * 16 * a.ppo * sizeof(float), s, a);
        return check_launch("dscnn_dw_dgrad_rows_kernel");
    }
    if (dscnn_dw_dgrad_rows_s2_covers(a)) {
        hipLaunchKernelGGL((dscnn_dw_dgrad_rows_s2_kernel<25, 10, 13, 5>), dim3((unsigned)(a.planes / 8)), dim3(256), (size_t)8 * (a.ppo + 2 * a.ppi) * sizeof(float), s, a);
        return check_launch("dscnn_dw_dgrad_rows_s2_kernel");
    }
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

// The stride-1 units on 13 x 5 maps (four of five): a WAVE owns four consecutive channels -- per utterance their x planes and their dz
// planes are two contiguous blocks of 4 x Pp floats, fetched as 73 float4 each (two coalesced loads per lane and block, the next
// utterance's in flight while this one is multiplied) into wave-private LDS; lane (plane, h) then takes map row h of dz against the three
// x rows around it: 20 LDS reads and 45 fmaf per lane and utterance.  (The kernel above issues nine predicated gathers per element and
// ran at 2.1 TB/s.)  The 13 row lanes of a plane meet in LDS in a fixed order; chunk slabs as above.
template <int H, int W>
__global__ __launch_bounds__(256) void dscnn_dw_wgrad_rows_kernel(const DsDwWgradArgs a) {
    constexpr int NPL = 4, P = H * W, PP = P + 2 * kHalo, BLK = NPL * PP, N4 = BLK / 4;
    static_assert(NPL * H <= 64 && BLK % 4 == 0 && N4 > 64 && N4 <= 128 && 64 * 9 <= 2 * BLK, "wave geometry");
    __shared__ __attribute__((aligned(16))) float s_w[4][2 * BLK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (blockIdx.y * 4 + wave) * NPL;
    if (c0 >= a.c) return;                                      // (no workgroup barrier below)
    float* sx = s_w[wave];
    float* sd = sx + BLK;
    const int n0 = blockIdx.x * a.utt_per_block;
    const int cnt = min(a.utt_per_block, a.batch - n0);
    const int pl = lane / H, h = lane - pl * H;
    const bool act = pl < NPL;
    const int c = c0 + (act ? pl : 0);
    const bool aff = a.x_scale != nullptr;
    const float xsc = aff ? a.x_scale[c] : 1.0f, xsf = aff ? a.x_shift[c] : 0.f;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    const bool two = lane + 64 < N4;
    const int l2 = two ? lane + 64 : lane;
    f32x4 x0, x1, d0, d1;
    {
        const size_t off = ((size_t)n0 * a.c + c0) * PP;
        const f32x4* xb = reinterpret_cast<const f32x4*>(a.x + off);
        const f32x4* db = reinterpret_cast<const f32x4*>(a.dz + off);
        x0 = xb[lane]; x1 = xb[l2]; d0 = db[lane]; d1 = db[l2];
    }
    for (int i = 0; i < cnt; ++i) {
        wave_sync();                                            // (the previous utterance's reads are done)
        reinterpret_cast<f32x4*>(sx)[lane] = x0;
        reinterpret_cast<f32x4*>(sd)[lane] = d0;
        if (two) { reinterpret_cast<f32x4*>(sx)[lane + 64] = x1; reinterpret_cast<f32x4*>(sd)[lane + 64] = d1; }
        wave_sync();
        if (i + 1 < cnt) {
            const size_t off = ((size_t)(n0 + i + 1) * a.c + c0) * PP;
            const f32x4* xb = reinterpret_cast<const f32x4*>(a.x + off);
            const f32x4* db = reinterpret_cast<const f32x4*>(a.dz + off);
            x0 = xb[lane]; x1 = xb[l2]; d0 = db[lane]; d1 = db[l2];
        }
        if (act) {
            float xin[3][W + 2], g[W];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {                    // x row h + rr - 1 (tap row di = rr), zero outside the map
                const int hh = h + rr - 1;
                const bool hv = hh >= 0 && hh < H;
                const float* src = sx + pl * PP + kHalo + (hv ? hh : h) * W;
                xin[rr][0] = 0.f; xin[rr][W + 1] = 0.f;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    float v = src[w];
                    if (aff) v = fmaxf(fmaf(v, xsc, xsf), 0.f);
                    xin[rr][w + 1] = hv ? v : 0.f;
                }
            }
#pragma unroll
            for (int w = 0; w < W; ++w) g[w] = sd[pl * PP + kHalo + h * W + w];
#pragma unroll
            for (int di = 0; di < 3; ++di)
#pragma unroll
                for (int dj = 0; dj < 3; ++dj)
#pragma unroll
                    for (int w = 0; w < W; ++w) acc[di * 3 + dj] = fmaf(xin[di][w + dj], g[w], acc[di * 3 + dj]);   // x[h + di - 1][w + dj - 1] dz[h][w]
        }
    }
    wave_sync();
    if (act) {
#pragma unroll
        for (int k = 0; k < 9; ++k) sx[lane * 9 + k] = acc[k];
    }
    wave_sync();
    if (lane < NPL * 9) {
        const int p2 = lane / 9, k = lane - p2 * 9;
        float v = 0.f;
#pragma unroll
        for (int hh = 0; hh < H; ++hh) v += sx[(p2 * H + hh) * 9 + k];
        a.partial[((size_t)blockIdx.x * 9 + k) * a.c + c0 + p2] = v;
    }
}

// The stride-2 unit (x on 25 x 10, dz on 13 x 5; pad_t 1, pad_l 0), same scheme: per utterance the wave's four x planes are 258 float4,
// its four dz planes 73; lane (plane, oh) takes dz row oh against x rows 2 oh - 1 .. 2 oh + 1.
template <int HI, int WI, int HO, int WO>
__global__ __launch_bounds__(256) void dscnn_dw_wgrad_rows_s2_kernel(const DsDwWgradArgs a) {
    constexpr int NPL = 4, PI = HI * WI, PO = HO * WO, PPI = PI + 2 * kHalo, PPO = PO + 2 * kHalo;
    constexpr int NX4 = NPL * PPI / 4, NZ4 = NPL * PPO / 4, XPT = (NX4 + 63) / 64, ZPT = (NZ4 + 63) / 64;
    static_assert(NPL * HO <= 64 && (NPL * PPI) % 4 == 0 && (NPL * PPO) % 4 == 0 && WI == 2 * WO && HI == 2 * HO - 1 && (PPI % 2) == 0 && 64 * 9 <= NPL * PPI, "wave geometry");
    __shared__ __attribute__((aligned(16))) float s_w[4][NPL * (PPI + PPO)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (blockIdx.y * 4 + wave) * NPL;
    if (c0 >= a.c) return;                                      // (no workgroup barrier below)
    float* sx = s_w[wave];
    float* sd = sx + NPL * PPI;
    const int n0 = blockIdx.x * a.utt_per_block;
    const int cnt = min(a.utt_per_block, a.batch - n0);
    const int pl = lane / HO, oh = lane - pl * HO;
    const bool act = pl < NPL;
    const int c = c0 + (act ? pl : 0);
    const bool aff = a.x_scale != nullptr;
    const float xsc = aff ? a.x_scale[c] : 1.0f, xsf = aff ? a.x_shift[c] : 0.f;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    f32x4 xv[XPT], dv[ZPT];
    auto fetch = [&](int n) {
        const f32x4* xb = reinterpret_cast<const f32x4*>(a.x + ((size_t)n * a.c + c0) * PPI);
        const f32x4* db = reinterpret_cast<const f32x4*>(a.dz + ((size_t)n * a.c + c0) * PPO);
#pragma unroll
        for (int j = 0; j < XPT; ++j) xv[j] = xb[min(lane + 64 * j, NX4 - 1)];
#pragma unroll
        for (int j = 0; j < ZPT; ++j) dv[j] = db[min(lane + 64 * j, NZ4 - 1)];
    };
    fetch(n0);
    for (int i = 0; i < cnt; ++i) {
        wave_sync();                                            // (the previous utterance's reads are done)
#pragma unroll
        for (int j = 0; j < XPT; ++j)
            if (lane + 64 * j < NX4) reinterpret_cast<f32x4*>(sx)[lane + 64 * j] = xv[j];
#pragma unroll
        for (int j = 0; j < ZPT; ++j)
            if (lane + 64 * j < NZ4) reinterpret_cast<f32x4*>(sd)[lane + 64 * j] = dv[j];
        wave_sync();
        if (i + 1 < cnt) fetch(n0 + i + 1);
        if (act) {
            float xin[3][WI + 1], g[WO];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {                    // x row 2 oh + rr - 1 (tap row di = rr), zero outside the map
                const int hh = 2 * oh + rr - 1;
                const bool hv = hh >= 0 && hh < HI;
                const float2* src = reinterpret_cast<const float2*>(sx + pl * PPI + kHalo + (hv ? hh : 0) * WI);
                xin[rr][WI] = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < WI / 2; ++w2) {
                    const float2 t = src[w2];
                    float v0 = t.x, v1 = t.y;
                    if (aff) { v0 = fmaxf(fmaf(v0, xsc, xsf), 0.f); v1 = fmaxf(fmaf(v1, xsc, xsf), 0.f); }
                    xin[rr][2 * w2] = hv ? v0 : 0.f;
                    xin[rr][2 * w2 + 1] = hv ? v1 : 0.f;
                }
            }
#pragma unroll
            for (int w = 0; w < WO; ++w) g[w] = sd[pl * PPO + kHalo + oh * WO + w];
#pragma unroll
            for (int di = 0; di < 3; ++di)
#pragma unroll
                for (int dj = 0; dj < 3; ++dj)
#pragma unroll
                    for (int w = 0; w < WO; ++w) acc[di * 3 + dj] = fmaf(xin[di][2 * w + dj], g[w], acc[di * 3 + dj]);   // x[2 oh + di - 1][2 ow + dj] dz[oh][ow]
        }
    }
    wave_sync();
    if (act) {
#pragma unroll
        for (int k = 0; k < 9; ++k) sx[lane * 9 + k] = acc[k];
    }
    wave_sync();
    if (lane < NPL * 9) {
        const int p2 = lane / 9, k = lane - p2 * 9;
        float v = 0.f;
#pragma unroll
        for (int hh = 0; hh < HO; ++hh) v += sx[(p2 * HO + hh) * 9 + k];
        a.partial[((size_t)blockIdx.x * 9 + k) * a.c + c0 + p2] = v;
    }
}

static bool dscnn_dw_wgrad_rows_s2_covers(const DsDwWgradArgs& a) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return a.sh == 2 && a.sw == 2 && a.pad_t == 1 && a.pad_l == 0 && a.h_in == 25 && a.w_in == 10 && a.oh == 13 && a.ow == 5 &&
           a.ppi == 25 * 10 + 2 * kHalo && a.ppo == 13 * 5 + 2 * kHalo && a.c % 4 == 0 && !a.fly.da && al(a.x) && al(a.dz) && tune_get(TCR_TUNE_DW_WGRAD) != 1;
}

static bool dscnn_dw_wgrad_rows_covers(const DsDwWgradArgs& a) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return a.sh == 1 && a.sw == 1 && a.pad_t == 1 && a.pad_l == 1 && a.h_in == 13 && a.w_in == 5 && a.oh == 13 && a.ow == 5 && a.ppi == a.ppo &&
           a.ppo == 13 * 5 + 2 * kHalo && a.c % 4 == 0 && !a.fly.da && al(a.x) && al(a.dz) && tune_get(TCR_TUNE_DW_WGRAD) != 1;
}

static int dw_wgrad_chunks(int batch) {
    int n = ceil_div(batch, 32);
    if (n > 128) n = 128;
    return n < 1 ? 1 : n;
}

size_t dscnn_dw_wgrad_partial_floats(int batch, int c) { return (size_t)dw_wgrad_chunks(batch) * 9 * c; }

int launch_dscnn_dw_wgrad(DsDwWgradArgs a, float* dw, hipStream_t s) {
    a.utt_per_block = ceil_div(a.batch, dw_wgrad_chunks(a.batch));
    dim3 grid(ceil_div(a.batch, a.utt_per_block), ceil_div(a.c, 4));
    if (dscnn_dw_wgrad_rows_covers(a)) {
        grid.y = ceil_div(a.c, 16);
        hipLaunchKernelGGL((dscnn_dw_wgrad_rows_kernel<13, 5>), grid, dim3(256), 0, s, a);
        TCR_TRY(check_launch("dscnn_dw_wgrad_rows_kernel"));
    } else if (dscnn_dw_wgrad_rows_s2_covers(a)) {
        grid.y = ceil_div(a.c, 16);
        hipLaunchKernelGGL((dscnn_dw_wgrad_rows_s2_kernel<25, 10, 13, 5>), grid, dim3(256), 0, s, a);
        TCR_TRY(check_launch("dscnn_dw_wgrad_rows_s2_kernel"));
    } else {
        hipLaunchKernelGGL(dscnn_dw_wgrad_kernel, grid, dim3(256), 0, s, a);
        TCR_TRY(check_launch("dscnn_dw_wgrad_kernel"));
    }
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
