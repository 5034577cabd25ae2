// Backward of TC-ResNet's BN'd convolutions as GROUP-RESIDENT PHASES -- the mirror image of train_fused.hip
// (tf.gradients of audio_nets/tc_resnet.py:6-54 inside slim.learning.create_train_op, helper/trainer.py:199-222).
//
// The per-layer chain was, for every BN unit:  reduce (sum dz, sum dz*xhat) -> finalize -> bn_bwd_apply (dy) -> data gradient
// (1-2 launches), each a separate pass over HBM.  A phase does, for a group of G utterances resident in LDS:
//   STAGE   dy = k1 * (dz - k2 - (raw - mean) * k3),  dz = dA [act > 0] ([shortcut > 0])   -- BatchNorm backward applied on the
//           fly from the coefficients the finalize kernel left; written once to HBM (the weight-gradient kernels read it);
//   CONV    the data gradient(s) from LDS on the exact-f32 MFMA (per output phase t mod stride a stride-1 convolution over dy
//           with the re-arranged weights of launch_dgrad_weights_multi), accumulated into the group's dx rows in LDS -- conv_a's
//           and the block's `down` shortcut's gradients meet there;
//   FINAL   dx (+ the identity shortcut's gradient) -> HBM, and the statistics of the NEXT BN backward (the unit(s) whose
//           output activation this dx is the gradient of): sum dz', sum dz' * xhat' per channel -> one partial row per workgroup.
// So a unit costs one phase + one finalize instead of four to five launches, and dz / dy are not re-read.  Sums are reduced lane
// group of a channel (shuffle tree) -> the channel's LDS cell -> per-workgroup row, all in a fixed order: bitwise reproducible.
#include "kernels.h"

namespace tcr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// taps / first offset / positions of output phase r of a (k, stride, pad_lo) data gradient (see launch_conv_dgrad_mfma)
__device__ __forceinline__ void dgrad_phase(int k, int stride, int pad_lo, int tin, int r, int* cnt, int* d_min, int* nu) {
    const int res_mod = (r + pad_lo) % stride;
    const int jmax = (k - 1) - (((k - 1) - res_mod + stride) % stride);
    *cnt = jmax < 0 ? 0 : jmax / stride + 1;
    *d_min = jmax < 0 ? 0 : (r + pad_lo - jmax) / stride;
    *nu = (tin - r + stride - 1) / stride;
}

template <int NW, int R>
__device__ __forceinline__ void bwd_layer(const BwdLayer& L, const float* __restrict__ xin, const int in_sz, float* __restrict__ xout,
                                          const int out_sz, const int ng, const int wave, const int r, const int q) {
    const int tps = L.tout + 2 * kHalo, tpo = L.tin + 2 * kHalo;
    const int nrt = (L.cin + 15) / 16;
    const int C4 = L.cout >> 2;
    const int wstep = 4 * L.cin;
    const int xstep = 4 * tps;
    int base = 0;
    for (int ph = 0; ph < L.stride; ++ph) {
        int cnt, d_min, nu;
        dgrad_phase(L.k, L.stride, L.pad_lo, L.tin, ph, &cnt, &d_min, &nu);
        if (cnt == 0 || nu <= 0) { base += cnt; continue; }
        const float* w = L.wt + (size_t)base * L.cout * L.cin;
        const int npos = ng * nu;
        const int ncp = (npos + 31) / 32;
        const int nsteps = cnt * C4;
        const float inv_nu = 1.0f / (float)nu;
        for (int job = wave; job < ncp * nrt; job += NW) {
            const int cp = job / nrt, m = job - cp * nrt;
            const int aidx = q * L.cin + min(m * 16 + r, L.cin - 1);
            const int p0 = min(cp * 32 + r, npos - 1), p1 = min(cp * 32 + 16 + r, npos - 1);
            const int g0 = (int)(((float)p0 + 0.5f) * inv_nu), g1 = (int)(((float)p1 + 0.5f) * inv_nu);
            const int u0 = p0 - g0 * nu, u1 = p1 - g1 * nu;
            const int xo0 = g0 * in_sz + q * tps + u0 + kHalo + d_min;
            const int xo1 = g1 * in_sz + q * tps + u1 + kHalo + d_min;
            f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int last = nsteps - 1;
            float ar[R];
#pragma unroll
            for (int i = 0; i < R; ++i) ar[i] = w[aidx + min(i, last) * wstep];
            int off = xstep, c4 = 1, j = 0;
            if (C4 == 1) { c4 = 0; off = j = 1; }
            float b0 = xin[xo0], b1 = xin[xo1];
#define TCR_BWD_STEP(AREG, RELOAD)                                                                      \
    {                                                                                                   \
const float nb0 = xin[xo0 + off], nb1 = xin[xo1 + off];     /* (one step past the end: inside the pad) */ \
acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(AREG, b0, acc0, 0, 0, 0);                           \
acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(AREG, b1, acc1, 0, 0, 0);                           \
RELOAD                                                                                          \
b0 = nb0;                                                                                       \
b1 = nb1;                                                                                       \
off += xstep;                                                                                   \
if (++c4 == C4) { c4 = 0; off = ++j; }                                                          \
    }
            int s0 = 0;
            for (; s0 + R <= nsteps; s0 += R) {
#pragma unroll
                for (int i = 0; i < R; ++i) TCR_BWD_STEP(ar[i], ar[i] = w[aidx + min(s0 + R + i, last) * wstep];)
            }
#pragma unroll
            for (int i = 0; i < R - 1; ++i)
                if (s0 + i < nsteps) TCR_BWD_STEP(ar[i], )
#undef TCR_BWD_STEP
            // each dx element is owned by exactly one lane of one job of this layer: plain read-modify-write in LDS
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (cp * 32 + nt * 16 + r >= npos) continue;
                const int g = nt == 0 ? g0 : g1, u = nt == 0 ? u0 : u1;
                const f32x4 ac = nt == 0 ? acc0 : acc1;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int co = m * 16 + q * 4 + reg;
                    if (co < L.cin) xout[g * out_sz + co * tpo + kHalo + u * L.stride + ph] += ac[reg];
                }
            }
        }
        base += cnt;
    }
}

template <int NW, int R>
__global__ __launch_bounds__(NW * 64) void train_bwd_phase_kernel(const TrainBwdPhaseArgs a) {
    constexpr int NT = NW * 64;
    float* lds = reinterpret_cast<float*>(dyn_lds());
    float* xs[2] = {lds + a.src_off[0], lds + a.src_off[1]};
    float* xo = lds + a.out_off;
    float* stat = lds + a.stat_off;                        // [2 targets][2][cstat]: one owner thread per channel
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    for (int i = tid; i < 4 * a.cstat; i += NT) stat[i] = 0.f;
    const int otp = a.out_t + 2 * kHalo;
    const int orow = a.out_c * otp;

    for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
        const int n0 = grp * a.group;
        const int ng = min(a.group, a.batch - n0);
        __syncthreads();
        // ---- stage the dy sources; zero the dx rows ----
        for (int si = 0; si < 2; ++si) {
            const BwdSrc& S = a.src[si];
            if (S.kind == 0) continue;
            const int tp = S.t + 2 * kHalo;
            const int row = S.c * tp;
            const float inv_row = 1.0f / (float)row, inv_tp = 1.0f / (float)tp;
            const size_t gbase = (size_t)n0 * row;
            const int total = ng * row;
            constexpr int SU = 4;
            for (int i0 = tid; i0 < total; i0 += NT * SU) {
                float vd[SU], vr[SU], v1[SU], v2[SU];
                int ix[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    ix[u] = min(i0 + u * NT, total - 1);
                    const size_t gi = gbase + ix[u];
                    if (S.kind == 2) { vd[u] = S.da[gi]; vr[u] = 0.f; v1[u] = 1.f; v2[u] = 1.f; continue; }
                    const int g = fast_div(ix[u], row, inv_row);
                    const int ch = fast_div(ix[u] - g * row, tp, inv_tp);
                    vd[u] = S.bcast ? S.da[(size_t)(n0 + g) * S.c + ch] : S.da[gi];
                    vr[u] = S.raw[gi];
                    v1[u] = S.m1 ? S.m1[gi] : 1.f;
                    v2[u] = S.m2 ? S.m2[gi] : 1.f;
                }
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    if (i0 + u * NT >= total) break;
                    const int i = ix[u];
                    const int g = fast_div(i, row, inv_row);
                    const int rem = i - g * row;
                    const int ch = fast_div(rem, tp, inv_tp);
                    const int tt = rem - ch * tp - kHalo;
                    float dy = 0.f;
                    if (S.kind == 2) {
                        dy = vd[u];                        // (materialised dy: its halo is zero)
                    } else if (tt >= 0 && tt < S.t) {
                        float dz = vd[u];
                        if (!(v1[u] > 0.f)) dz = 0.f;
                        if (!(v2[u] > 0.f)) dz = 0.f;
                        dy = S.k1[ch] * (dz - S.k2[ch] - (vr[u] - S.mean[ch]) * S.k3[ch]);     // == bn_bwd_apply_kernel
                    }
                    xs[si][g * a.src_sz[si] + rem] = dy;
                    if (S.out_dy) S.out_dy[gbase + i] = dy;
                }
            }
        }
        if (a.n_layers == 0) continue;
        for (int i = tid; i < ng * a.out_sz; i += NT) xo[i] = 0.f;
        __syncthreads();
        // ---- data gradients into the dx rows ----
        for (int li = 0; li < a.n_layers; ++li) {
            const BwdLayer& L = a.layer[li];
            bwd_layer<NW, R>(L, xs[L.src], a.src_sz[L.src], xo, a.out_sz, ng, wave, r, q);
            __syncthreads();
        }
        // ---- dx (+ identity-shortcut gradient) -> HBM; statistics of the next BN backward ----
        // A channel is owned by a group of `tpc` consecutive lanes (a power of two <= 64): the group walks the channel's
        // ng * T elements EU at a time -- every load of a trip is issued before the first use -- keeps the two sums of each target
        // in registers, folds them with a shuffle tree inside the group and adds them to the channel's LDS cell.  (One owner per
        // channel and workgroup: fixed order, no atomics.)
        {
            constexpr int EU = 4;
            const int tpc = a.tpc;
            const int chn = tid / tpc, sub = tid - chn * tpc;
            const bool live = chn < a.out_c;
            const int ch = live ? chn : 0;
            const int nel = ng * a.out_t;
            const float inv_t = 1.0f / (float)a.out_t;
            const size_t obase = (size_t)n0 * orow;
            float q1[2] = {0.f, 0.f}, q2[2] = {0.f, 0.f};
            float mu[2], is[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) { mu[k] = a.stat[k].on ? a.stat[k].mean[ch] : 0.f; is[k] = a.stat[k].on ? a.stat[k].invstd[ch] : 0.f; }
            if (live) {
                for (int e0 = sub; e0 < nel; e0 += tpc * EU) {
                    float v[EU], av[EU], am[EU], m1[2][EU], m2[2][EU], rw[2][EU];
                    size_t gi[EU];
#pragma unroll
                    for (int u = 0; u < EU; ++u) {
                        const int e = min(e0 + u * tpc, nel - 1);
                        const int g = fast_div(e, a.out_t, inv_t);
                        const int t = e - g * a.out_t;
                        gi[u] = obase + ((size_t)g * a.out_c + ch) * otp + kHalo + t;
                        v[u] = xo[g * a.out_sz + ch * otp + kHalo + t];
                        av[u] = a.add ? (a.add_bcast ? a.add[(size_t)(n0 + g) * a.out_c + ch] : a.add[gi[u]]) : 0.f;
                        am[u] = (a.add && a.add_mask) ? a.add_mask[gi[u]] : 1.f;
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            m1[k][u] = a.stat[k].on ? a.stat[k].m1[gi[u]] : 0.f;
                            m2[k][u] = (a.stat[k].on && a.stat[k].m2) ? a.stat[k].m2[gi[u]] : 1.f;
                            rw[k][u] = a.stat[k].on ? a.stat[k].raw[gi[u]] : 0.f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < EU; ++u) {
                        if (e0 + u * tpc >= nel) break;
                        float val = v[u];
                        if (a.add && am[u] > 0.f) val += av[u];
                        a.out_dx[gi[u]] = val;
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            if (!a.stat[k].on) continue;
                            float dz = val;
                            if (!(m1[k][u] > 0.f)) dz = 0.f;
                            if (!(m2[k][u] > 0.f)) dz = 0.f;
                            q1[k] += dz;
                            q2[k] = fmaf(dz, (rw[k][u] - mu[k]) * is[k], q2[k]);
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!a.stat[k].on) continue;
                for (int msk = 1; msk < tpc; msk <<= 1) {
                    q1[k] += __shfl_xor(q1[k], msk);
                    q2[k] += __shfl_xor(q2[k], msk);
                }
                if (live && sub == 0) {
                    stat[(k * 2) * a.cstat + ch] += q1[k];
                    stat[(k * 2 + 1) * a.cstat + ch] += q2[k];
                }
            }
        }
    }
    if (a.n_layers == 0) return;
    __syncthreads();
    for (int k = 0; k < 2; ++k) {
        if (!a.stat[k].on) continue;
        for (int i = tid; i < 2 * a.out_c; i += NT) {
            const int which = i / a.out_c, co = i - which * a.out_c;
            a.stat[k].partial[((size_t)blockIdx.x * 2 + which) * a.out_c + co] = stat[(k * 2 + which) * a.cstat + co];
        }
    }
}

static bool configure_bwd_phase(TrainBwdPhaseArgs& a, size_t* lds_out, int* grid_out) {
    const int knob = tune_get(TCR_TUNE_PHASE_CFG);
    const int NW = knob / 100 == 4 ? 4 : 8;
    a.nw = NW;
    int per_utt = 0;
    for (int si = 0; si < 2; ++si) {
        a.src_sz[si] = 0;
        if (a.src[si].kind == 0) continue;
        a.src_sz[si] = (a.src[si].c * (a.src[si].t + 2 * kHalo) + 3) / 4 * 4;
        if ((int64_t)a.src_sz[si] * 64 >= (1 << 22)) return false;
        per_utt += a.src_sz[si];
    }
    a.out_sz = a.n_layers ? (a.out_c * (a.out_t + 2 * kHalo) + 3) / 4 * 4 : 0;
    per_utt += a.out_sz;
    for (int i = 0; i < a.n_layers; ++i) {
        const BwdLayer& L = a.layer[i];
        const BwdSrc& S = a.src[L.src];
        if (S.kind == 0 || L.cout % 4 != 0 || L.cout != S.c || L.tout != S.t || L.cin != a.out_c || L.tin != a.out_t) return false;
        if (L.stride < 1 || L.stride > 2 || L.k > 9) return false;
    }
    a.cstat = max(16, (a.out_c + 15) / 16 * 16);
    const size_t stat_bytes = (size_t)4 * a.cstat * sizeof(float);
    a.tpc = 1;
    while (a.tpc * 2 <= 64 && a.tpc * 2 * a.out_c <= NW * 64) a.tpc *= 2;       // lanes per channel in the final pass
    int group = knob % 100 > 0 ? knob % 100 : 8;
    while (group > 1 && ((size_t)group * per_utt + 64) * sizeof(float) + stat_bytes > 80 * 1024) --group;
    while (group > 1 && ceil_div(a.batch, group) < 512) --group;
    const size_t lds = ((size_t)group * per_utt + 64) * sizeof(float) + stat_bytes;
    if (lds > 160 * 1024) return false;
    a.group = group; a.n_groups = ceil_div(a.batch, group);
    a.src_off[0] = 0; a.src_off[1] = group * a.src_sz[0]; a.out_off = group * (a.src_sz[0] + a.src_sz[1]);
    a.stat_off = group * per_utt + 64;
    *lds_out = lds;
    *grid_out = min(a.n_groups, kPhaseMaxRows);
    return true;
}

int train_bwd_phase_rows(const TrainBwdPhaseArgs& a0) {
    TrainBwdPhaseArgs a = a0;
    size_t lds;
    int grid;
    return configure_bwd_phase(a, &lds, &grid) ? grid : -1;
}

int launch_train_bwd_phase(TrainBwdPhaseArgs a, int* rows_out, hipStream_t s) {
    size_t lds;
    int grid;
    if (!configure_bwd_phase(a, &lds, &grid)) return 1;
    if (rows_out) *rows_out = grid;
    void (*kern)(const TrainBwdPhaseArgs) = a.nw == 4 ? train_bwd_phase_kernel<4, 4> : train_bwd_phase_kernel<8, 4>;
    static size_t configured[2] = {0, 0};
    size_t& cfg = configured[a.nw == 4 ? 0 : 1];
    if (lds > 64 * 1024 && lds > cfg) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return 1;
        }
        cfg = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(a.nw * 64), lds, s, a);
    return check_launch("train_bwd_phase_kernel");
}

}  // namespace tcr
