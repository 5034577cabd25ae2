// Temporal (k x 1) convolutions of TC-ResNet as register-blocked direct convolutions on the VALU.
//
// Replaces slim.conv2d -> tf.nn.conv2d (NHWC, SAME, no bias, W == 1) at
// audio_nets/tc_resnet.py:21 (3x1 stem) and :37-38 (9x1 block convs), forward and data-gradient.
//
// Mapping (gfx950): lane == output position (b, t) of the flattened [batch x T_out] list, P positions
// per lane; every lane accumulates CT output channels in registers.  The weights of a (tap, ci)
// step are CT consecutive floats of the [k][Cin][Cout] tensor and are WAVE-UNIFORM, so they arrive
// through scalar loads (s_load_dwordx8/x16) and feed v_pk_fma_f32 straight from SGPRs: no LDS and
// no VGPRs are spent on weights, and the inner loop is FMA-dense.  Activations are planar
// [b][c][HALO + t] with a zero halo, so SAME padding needs no per-tap bounds checks and
// consecutive lanes read consecutive addresses.
#include "kernels.h"

namespace tcr {

// KS > 1 splits the Cin reduction over KS waves of the workgroup (same positions, disjoint ci ranges) and
// combines them through LDS: late layers have few positions (B*T_out shrinks 49 -> 7 while Cin*K grows), and
// a wave's stream of scalar weight loads is latency-bound, so more, shorter waves are what fills the SIMDs.
template <int K, int S, int CT, int P, int EPI, int KS>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const ConvArgs a) {
    __shared__ float s_red[(KS > 1) ? (KS - 1) * (4 / KS) * P * CT * 64 : 1];
    const int co0 = blockIdx.y * CT;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int grp = wave / KS;              // position group inside the workgroup
    const int part = wave % KS;             // slice of the Cin reduction
    const int p0 = ((blockIdx.x * (4 / KS) + grp) * 64) * P + lane;

    float acc[P][CT];
    const float* xb[P];
    const int ci_per = (a.cin + KS - 1) / KS;
    const int ci_begin = part * ci_per;
    const int ci_end = min(a.cin, ci_begin + ci_per);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int p = min(p0 + 64 * i, a.npos - 1);
        const int n = p / a.tout;
        const int t = p - n * a.tout;
        xb[i] = a.x + ((size_t)n * a.cin + ci_begin) * a.tpi + t * S + a.xoff;
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[i][c] = 0.f;
    }
    const float* __restrict__ wr = a.w + (size_t)ci_begin * a.cout + co0;
    const size_t wtap = (size_t)a.cin * a.cout;
    for (int ci = ci_begin; ci < ci_end; ++ci) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            float xv[P];
#pragma unroll
            for (int i = 0; i < P; ++i) xv[i] = xb[i][j];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const float wv = wr[j * wtap + c];
#pragma unroll
                for (int i = 0; i < P; ++i) acc[i][c] = fmaf(wv, xv[i], acc[i][c]);
            }
        }
        wr += a.cout;
#pragma unroll
        for (int i = 0; i < P; ++i) xb[i] += a.tpi;
    }

    if (KS > 1) {
        float* red = s_red + (size_t)grp * (KS - 1) * P * CT * 64;
        if (part > 0) {
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int c = 0; c < CT; ++c) red[((part - 1) * P * CT + i * CT + c) * 64 + lane] = acc[i][c];
        }
        __syncthreads();
        if (part > 0) return;
#pragma unroll
        for (int q = 0; q < KS - 1; ++q)
#pragma unroll
            for (int i = 0; i < P; ++i)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[i][c] += red[(q * P * CT + i * CT + c) * 64 + lane];
    }

#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int p = p0 + 64 * i;
        if (p >= a.npos) continue;
        const int n = p / a.tout;
        const int t = p - n * a.tout;
        const size_t row0 = ((size_t)n * a.cout + co0) * a.tpo + kHalo + t;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (co0 + c >= a.cout) continue;        // (no `break`: keeps acc[] statically indexed)
            float v = acc[i][c];
            const size_t o = row0 + (size_t)c * a.tpo;
            if (EPI == EPI_AFFINE) {
                v = fmaf(v, a.scale[co0 + c], a.shift[co0 + c]);
                if (a.res) v = fmaxf(v + a.res[o], 0.f);     // net += layer_in; relu  (tc_resnet.py:40-41)
                else if (a.relu) v = fmaxf(v, 0.f);
            }
            float* q = a.y + o;
            q[0] = v;
            if (t == 0) { q[-4] = 0.f; q[-3] = 0.f; q[-2] = 0.f; q[-1] = 0.f; }
            if (t == a.tout - 1) { q[1] = 0.f; q[2] = 0.f; q[3] = 0.f; q[4] = 0.f; }
        }
    }
}

// Data gradient: dx[b][ci][tin] = sum_{j, co} dy[b][co][(tin + pad_lo - j) / S] * W[j][ci][co]
// over the taps with (tin + pad_lo - j) divisible by S.  Lane == group u of S consecutive input
// positions (tin = S*u + r), so that every lane runs the same tap list; weights come from the
// transposed copy wt[K][Cout][Cin] (CT consecutive ci, wave-uniform -> scalar loads).
template <int K, int S, int CT>
__global__ __launch_bounds__(256) void conv_dgrad_kernel(const DgradArgs a) {
    const int ci0 = blockIdx.y * CT;
    const int g = min((int)(blockIdx.x * 256 + threadIdx.x), a.ngrp - 1);
    const bool live = (int)(blockIdx.x * 256 + threadIdx.x) < a.ngrp;
    const int n = g / a.ugrp;
    const int u = g - n * a.ugrp;

    float acc[S][CT];
#pragma unroll
    for (int r = 0; r < S; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = 0.f;

    // dy row base for output index t = u + d; d ranges over a small window, halo makes it safe.
    const float* dyb = a.dy + (size_t)n * a.cout * a.tpo + kHalo + u;
    const float* __restrict__ wr = a.wt + ci0;
    const size_t wtap = (size_t)a.cout * a.cin;
    for (int co = 0; co < a.cout; ++co) {
#pragma unroll
        for (int r = 0; r < S; ++r) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                // tin + pad_lo - j = S*u + (r + pad_lo - j) must be a multiple of S
                const int e = r + a.pad_lo - j;
                if ((e % S) != 0) continue;               // wave-uniform
                const int d = e / S;                       // t = u + d  (|d| <= 4 -> inside the halo)
                const float v = dyb[d];
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[r][c] = fmaf(wr[j * wtap + c], v, acc[r][c]);
            }
        }
        wr += a.cin;
        dyb += a.tpo;
    }
    if (!live) return;
#pragma unroll
    for (int r = 0; r < S; ++r) {
        const int tin = S * u + r;
        if (tin >= a.tin) continue;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (ci0 + c >= a.cin) continue;
            const size_t o = ((size_t)n * a.cin + ci0 + c) * a.tpi + kHalo + tin;
            float v = acc[r][c];
            if (a.add) {
                float rv = a.add_bcast ? a.add[(size_t)n * a.cin + ci0 + c] : a.add[o];
                if (a.add_mask && !(a.add_mask[o] > 0.f)) rv = 0.f;
                v += rv;
            }
            float* q = a.dx + o;
            q[0] = v;
            if (tin == 0) { q[-4] = 0.f; q[-3] = 0.f; q[-2] = 0.f; q[-1] = 0.f; }
            if (tin == a.tin - 1) { q[1] = 0.f; q[2] = 0.f; q[3] = 0.f; q[4] = 0.f; }
        }
    }
}

// wt[j][co][ci] = w[j][ci][co]
__global__ __launch_bounds__(256) void transpose_weights_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                                int k, int cin, int cout) {
    const int total = k * cin * cout;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int ci = i % cin;
        const int r = i / cin;
        const int co = r % cout;
        const int j = r / cout;
        wt[i] = w[((size_t)j * cin + ci) * cout + co];
    }
}

// waves launched for a given tiling; the host picks the reduction split that brings this to ~6+ waves/SIMD
static int pick_ksplit(int npos, int cout, int ct, int p, int cin) {
    const long base = (long)ceil_div(npos, 64 * p) * ceil_div(cout, ct);
    int ks = 1;
    while (ks < 4 && base * ks < 6144 && cin / (ks * 2) >= 4) ks *= 2;
    return ks;
}

template <int K, int S, int CT, int P, int KS>
static int launch_fwd_epi(const ConvArgs& a, int epi, hipStream_t s) {
    const dim3 grid(ceil_div(a.npos, 64 * P * (4 / KS)), ceil_div(a.cout, CT));
    if (epi == EPI_RAW) hipLaunchKernelGGL((conv_fwd_kernel<K, S, CT, P, EPI_RAW, KS>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_fwd_kernel<K, S, CT, P, EPI_AFFINE, KS>), grid, dim3(256), 0, s, a);
    return check_launch("conv_fwd_kernel");
}

template <int K, int S, int CT, int P>
static int launch_fwd_ksplit(const ConvArgs& a, int epi, hipStream_t s) {
    switch (pick_ksplit(a.npos, a.cout, CT, P, a.cin)) {
        case 1: return launch_fwd_epi<K, S, CT, P, 1>(a, epi, s);
        case 2: return launch_fwd_epi<K, S, CT, P, 2>(a, epi, s);
        default: return launch_fwd_epi<K, S, CT, P, 4>(a, epi, s);
    }
}

template <int K, int S>
static int launch_fwd_ks(const ConvArgs& a, int ct, int epi, hipStream_t s) {
    switch (ct) {
        case 8: return launch_fwd_ksplit<K, S, 8, 4>(a, epi, s);
        case 12: return launch_fwd_ksplit<K, S, 12, 4>(a, epi, s);
        case 16: return launch_fwd_ksplit<K, S, 16, 2>(a, epi, s);
        default: return launch_fwd_ksplit<K, S, 24, 2>(a, epi, s);
    }
}

// Channel tile: the tile in {24,16,12,8} with the least padded waste (ties -> larger tile).
int pick_channel_tile(int c) {
    const int cand[4] = {24, 16, 12, 8};
    int best = 24, best_waste = 1 << 30;
    for (int i = 0; i < 4; ++i) {
        const int waste = ceil_div(c, cand[i]) * cand[i] - c;
        if (waste < best_waste) { best_waste = waste; best = cand[i]; }
    }
    return best;
}

int launch_conv_fwd(int k, int stride, const ConvArgs& a, int epi, hipStream_t s) {
    const int ct = pick_channel_tile(a.cout);
    if (k == 3 && stride == 1) return launch_fwd_ks<3, 1>(a, ct, epi, s);
    if (k == 9 && stride == 1) return launch_fwd_ks<9, 1>(a, ct, epi, s);
    if (k == 9 && stride == 2) return launch_fwd_ks<9, 2>(a, ct, epi, s);
    if (k == 1 && stride == 1) return launch_fwd_ks<1, 1>(a, ct, epi, s);
    if (k == 1 && stride == 2) return launch_fwd_ks<1, 2>(a, ct, epi, s);
    set_error("conv forward: kernel %dx1 stride %d has no gfx950 instantiation", k, stride);
    return TCR_ERR_ARG;
}

template <int K, int S>
static int launch_dgrad_ks(const DgradArgs& a, int ct, hipStream_t s) {
    const dim3 grid(ceil_div(a.ngrp, 256), ceil_div(a.cin, ct));
    switch (ct) {
        case 8: hipLaunchKernelGGL((conv_dgrad_kernel<K, S, 8>), grid, dim3(256), 0, s, a); break;
        case 12: hipLaunchKernelGGL((conv_dgrad_kernel<K, S, 12>), grid, dim3(256), 0, s, a); break;
        case 16: hipLaunchKernelGGL((conv_dgrad_kernel<K, S, 16>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((conv_dgrad_kernel<K, S, 24>), grid, dim3(256), 0, s, a); break;
    }
    return check_launch("conv_dgrad_kernel");
}

int launch_conv_dgrad(int k, int stride, const DgradArgs& a, hipStream_t s) {
    const int ct = pick_channel_tile(a.cin);
    if (k == 9 && stride == 1) return launch_dgrad_ks<9, 1>(a, ct, s);
    if (k == 9 && stride == 2) return launch_dgrad_ks<9, 2>(a, ct, s);
    if (k == 1 && stride == 2) return launch_dgrad_ks<1, 2>(a, ct, s);
    if (k == 1 && stride == 1) return launch_dgrad_ks<1, 1>(a, ct, s);
    if (k == 3 && stride == 1) return launch_dgrad_ks<3, 1>(a, ct, s);
    set_error("conv dgrad: kernel %dx1 stride %d has no gfx950 instantiation", k, stride);
    return TCR_ERR_ARG;
}

int launch_transpose_weights(const float* w, float* wt, int k, int cin, int cout, hipStream_t s) {
    const int total = k * cin * cout;
    hipLaunchKernelGGL(transpose_weights_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, w, wt, k, cin, cout);
    return check_launch("transpose_weights_kernel");
}

}  // namespace tcr
