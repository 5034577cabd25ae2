// Whole-network eval-mode forward of TC-ResNet in ONE persistent kernel: a workgroup pulls a group of G
// utterances, keeps every activation of the group in LDS (three rotating buffers), and walks the layer table:
//   features -> conv0 -> [down | conv_a -> conv_b (+shortcut, ReLU)] x blocks -> avg-pool -> fc / fc2 -> softmax.
// (audio_nets/tc_resnet.py:6-54 + slim.softmax, factory/audio_nets.py:147-156, with BN folded to scale/shift.)
//
// Why: the per-layer kernels move ~85 KB of activations per utterance through L2/HBM and pay a pipeline
// fill/drain per launch; fused, the only global traffic is the 9 KB feature tile in and 14 floats out, and the
// weights (258 KB for TCResNet8-1.0) stream from L2 with a register lookahead.  Every convolution is an implicit
// GEMM on the exact-f32 16x16x4 MFMA (bitwise an fmaf chain): D[co][position] with positions = G * T_out packed
// across utterances, A = weights straight from L1/L2, B = activations from the LDS rows (taps, stride and SAME
// padding are plain offsets into the zero-halo rows).  A job is (16 output channels) x (32 positions); the four
// waves of the workgroup take jobs round-robin, one s_barrier per layer.
#include "kernels.h"

namespace tcr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void net_fused_kernel(const FusedArgs a) {
    constexpr int CH = 8;                                   // weight lookahead: K-steps (of 4 input channels) per chunk
    float* lds = reinterpret_cast<float*>(dyn_lds());
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;

    for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
        const int n0 = grp * a.group;
        const int ng = min(a.group, a.batch - n0);
        // ---- stage the group's feature rows (contiguous in global memory) ----
        {
            const int row = a.in_c * a.in_tp;
            const float4* src = reinterpret_cast<const float4*>(a.feat + (size_t)n0 * row);
            float4* dst = reinterpret_cast<float4*>(lds + a.buf_off[0]);
            const int nvec = ng * row / 4;
            for (int i = tid; i < nvec; i += 256) dst[i] = src[i];
        }
        __syncthreads();

        for (int li = 0; li < a.n_layers; ++li) {
            const FusedLayer L = a.layer[li];
            const int tpi = L.tin + 2 * kHalo, tpo = L.tout + 2 * kHalo;
            const float* xin = lds + a.buf_off[L.in_buf];
            float* yout = lds + a.buf_off[L.out_buf];
            const float* res = L.res_buf >= 0 ? lds + a.buf_off[L.res_buf] : nullptr;
            const int in_sz = a.buf_sz[L.in_buf], out_sz = a.buf_sz[L.out_buf];
            const int res_sz = L.res_buf >= 0 ? a.buf_sz[L.res_buf] : 0;
            const int npos = ng * L.tout;
            const int ncp = (npos + 31) / 32;               // column pairs (32 positions)
            const int nrt = (L.cout + 15) / 16;             // row tiles (16 output channels)
            const int C4 = L.cin >> 2;
            const int nsteps = L.k * C4;
            const float* w = a.params + L.w_off;
            const float* scale = a.ss + L.ss_off;
            const float* shift = scale + L.c_pad;
            for (int job = wave; job < ncp * nrt; job += 4) {
                const int cp = job / nrt, m = job - cp * nrt;
                const int co_a = m * 16 + r;                // A-fragment column of this lane
                const bool wv = co_a < L.cout;
                const float* wl = w + (size_t)q * L.cout + (wv ? co_a : 0);
                int xo[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int p = min(cp * 32 + nt * 16 + r, npos - 1);
                    const int g = p / L.tout, t = p - g * L.tout;
                    xo[nt] = g * in_sz + q * tpi + t * L.stride + kHalo - L.pad_lo;
                }
                f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                // K-steps are (tap j, 4 input channels c4) pairs, linearised tap-major; weights are fetched one chunk
                // (CH steps) ahead of the MFMAs that consume them
                // K-steps are (tap j, 4 input channels c4) pairs, linearised tap-major; weights are fetched one chunk
                // (CH steps) ahead of the MFMAs that consume them.  (Also prefetching the LDS operands into registers
                // was measured SLOWER -- 338 vs 259 us -- the extra address arithmetic costs more than the latency saved.)
                float afA[CH], afB[CH];
                auto load_chunk = [&](int s0, float (&af)[CH]) {
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        const int s = s0 + i;
                        af[i] = (s < nsteps && wv) ? wl[(size_t)s * 4 * L.cout] : 0.f;   // (j*Cin + 4*c4) * Cout == s * 4 * Cout
                    }
                };
                auto mma_chunk = [&](int s0, int& j, int& c4, const float (&af)[CH]) {
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        if (s0 + i < nsteps) {
                            const int off = 4 * c4 * tpi + j;
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], xin[xo[0] + off], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], xin[xo[1] + off], acc1, 0, 0, 0);
                            if (++c4 == C4) { c4 = 0; ++j; }
                        }
                    }
                };
                int j = 0, c4 = 0;
                load_chunk(0, afA);
                for (int s0 = 0; s0 < nsteps; s0 += 2 * CH) {
                    load_chunk(s0 + CH, afB);
                    mma_chunk(s0, j, c4, afA);
                    load_chunk(s0 + 2 * CH, afA);
                    mma_chunk(s0 + CH, j, c4, afB);
                }
                // ---- epilogue: folded BN (+ shortcut) (+ ReLU) -> LDS rows of the output buffer ----
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int p = cp * 32 + nt * 16 + r;
                    if (p >= npos) continue;
                    const int g = p / L.tout, t = p - g * L.tout;
                    const f32x4 ac = nt == 0 ? acc0 : acc1;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int co = m * 16 + q * 4 + reg;
                        if (co >= L.cout) continue;
                        float v = fmaf(ac[reg], scale[co], shift[co]);
                        if (res) v = fmaxf(v + res[g * res_sz + co * tpo + kHalo + t], 0.f);      // tc_resnet.py:40-41
                        else if (L.relu) v = fmaxf(v, 0.f);
                        float* dst = yout + g * out_sz + co * tpo + kHalo + t;
                        dst[0] = v;
                        if (t == 0) { dst[-4] = 0.f; dst[-3] = 0.f; dst[-2] = 0.f; dst[-1] = 0.f; }
                        if (t == L.tout - 1) { dst[1] = 0.f; dst[2] = 0.f; dst[3] = 0.f; dst[4] = 0.f; }
                    }
                }
            }
            __syncthreads();
        }

        // ---- head: global average pool -> fc / fc2 -> softmax / sigmoid (tc_resnet.py:43-52) ----
        {
            const float* fb = lds + a.buf_off[a.feat_buf];
            const int fsz = a.buf_sz[a.feat_buf], tp = a.feat_t + 2 * kHalo;
            float* pooled = lds + a.buf_off[(a.feat_buf + 1) % 3];         // any buffer other than the feature buffer
            for (int i = tid; i < ng * a.feat_c; i += 256) {
                const int g = i / a.feat_c, c = i - g * a.feat_c;
                const float* row = fb + g * fsz + c * tp + kHalo;
                float s = 0.f;
                for (int t = 0; t < a.feat_t; ++t) s += row[t];
                pooled[i] = s / (float)a.feat_t;
            }
            __syncthreads();
            float* lg = pooled + a.group * a.feat_c;                        // [ng][nc + 2]
            const int no = a.nc + 2;
            for (int i = tid; i < ng * no; i += 256) {
                const int g = i / no, o = i - g * no;
                const float* pv = pooled + g * a.feat_c;
                float s = 0.f;
                if (o < a.nc) {
                    const float* wf = a.params + a.fc_off + o;
                    for (int c = 0; c < a.feat_c; ++c) s = fmaf(pv[c], wf[(size_t)c * a.nc], s);
                } else {
                    const float* wf = a.params + a.fc2_off + (o - a.nc);
                    for (int c = 0; c < a.feat_c; ++c) s = fmaf(pv[c], wf[(size_t)c * 2], s);
                }
                lg[i] = s;
            }
            __syncthreads();
            for (int g = tid; g < ng; g += 256) {
                const float* z = lg + g * no;
                float mx = z[0];
                for (int o = 1; o < a.nc; ++o) mx = fmaxf(mx, z[o]);
                float se = 0.f;
                for (int o = 0; o < a.nc; ++o) se += expf(z[o] - mx);
                const size_t n = (size_t)(n0 + g);
                for (int o = 0; o < a.nc; ++o) {
                    a.logits[n * a.nc + o] = z[o];
                    a.probs[n * a.nc + o] = expf(z[o] - mx) / se;
                }
                if (a.ranges) {
                    a.ranges[n * 2] = 1.0f / (1.0f + expf(-z[a.nc]));
                    a.ranges[n * 2 + 1] = 1.0f / (1.0f + expf(-z[a.nc + 1]));
                }
            }
            __syncthreads();
        }
    }
}

int launch_net_fused(const FusedArgs& a, size_t lds_bytes, int grid, hipStream_t s) {
#if !defined(TCR_HOST_EMULATION)
    static size_t configured = 0;
    if (lds_bytes > 64 * 1024 && lds_bytes > configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(net_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
            (void)hipGetLastError();
            return 1;       // caller falls back to the per-layer kernels
        }
        configured = lds_bytes;
    }
#endif
    hipLaunchKernelGGL(net_fused_kernel, dim3(grid), dim3(256), lds_bytes, s, a);
    return check_launch("net_fused_kernel");
}

}  // namespace tcr
