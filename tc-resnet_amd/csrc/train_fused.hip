// Train-mode forward of TC-ResNet as GROUP-RESIDENT PHASES (audio_nets/tc_resnet.py:6-54 with is_training=True;
// BN semantics of TCResNet_arg_scope, :102-123).
//
// Batch-statistics BN puts a grid-wide dependency after every convolution, so the training forward cannot be one kernel like
// the eval forward (fused.hip).  It is cut at exactly those points -- 1 + 2 per block phases instead of the per-layer chain
// conv -> statistics -> finalize -> normalise:
//   P0     features                                -> conv0            -> raw0                     + statistics
//   Pa_i   X_i = block input (built while staging) -> [down_i], conv_a -> raw_down_i, raw_a_i      + statistics of both
//   Pb_i   A_i = relu(bn(raw_a_i))                 -> conv_b           -> raw_b_i                  + statistics
//   Pend   X_last (and its shortcut)               -> (no conv: the activations the head and backward read)
// A workgroup pulls groups of G utterances.  STAGING builds the phase's input activation in LDS straight from the RAW outputs
// of the previous phase -- the batch-norm affine, ReLU and the residual sum are applied on the fly (scale / shift come from the
// tiny finalize kernel that ran in between) -- and writes it once to HBM, because backward reads it (filter gradients, ReLU
// masks).  The convolutions then run from LDS exactly like the eval kernel's layers (implicit GEMM on the exact-f32 16x16x4
// MFMA, 16 channels x 32 positions per job), store the raw output and accumulate sum / sum of squares per channel: lane ->
// 16-lane shuffle tree -> a per-WAVE row in LDS (jobs are dealt to waves statically, so the order is fixed) -> at kernel end
// the waves' rows are added in order into ONE partial row per workgroup, which bn_finalize sums in double.  No float atomics:
// the step stays bitwise reproducible.
//
// What this removes per BN layer: the statistics pass over the raw tensor, the normalise pass (read raw, write activation),
// and the next convolution's re-read of that activation from HBM.
#include "kernels.h"

namespace tcr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One convolution of the phase for the group's `ng` utterances: input rows in LDS, raw output to global, statistics to the
// wave's LDS row `wstat` ([2][cstat]: sums, sums of squares).
template <int NW, int R>
__device__ __forceinline__ void phase_layer(const TrainPhaseArgs& a, const PhaseLayer& L, const float* __restrict__ xin, float* wstat,
                                            const int n0, const int ng, const int wave, const int r, const int q) {
    const int tpi = L.tin + 2 * kHalo, tpo = L.tout + 2 * kHalo;
    const int npos = ng * L.tout;
    const int ncp = (npos + 31) / 32;
    const int nrt = (L.cout + 15) / 16;
    const int C4 = L.cin >> 2;
    const int nsteps = L.k * C4;
    const float* w = a.params + L.w_off;
    const int wstep = 4 * L.cout;
    const int xstep = 4 * tpi;
    const float inv_tout = 1.0f / (float)L.tout;
    // Job -> wave assignment.  When the row tiles divide the waves, a wave keeps ONE channel tile (m = wave % nrt) and walks
    // its position groups: the per-channel sums then stay in the lane's registers across all its jobs and are reduced across
    // lanes once per layer instead of once per job.  Otherwise jobs are dealt round-robin and reduced per job.
    const bool own = (NW % nrt) == 0;
    const int cp_step = own ? NW / nrt : 0;
    float ps1[4] = {0.f, 0.f, 0.f, 0.f}, ps2[4] = {0.f, 0.f, 0.f, 0.f};
    const int m_own = wave % nrt;
    for (int it = 0;; ++it) {
        int cp, m;
        if (own) { cp = wave / nrt + it * cp_step; m = m_own; if (cp >= ncp) break; }
        else { const int job = wave + it * NW; if (job >= ncp * nrt) break; cp = job / nrt; m = job - cp * nrt; }
        const int aidx = q * L.cout + min(m * 16 + r, L.cout - 1);
        const int p0 = min(cp * 32 + r, npos - 1), p1 = min(cp * 32 + 16 + r, npos - 1);
        const int g0 = (int)(((float)p0 + 0.5f) * inv_tout), g1 = (int)(((float)p1 + 0.5f) * inv_tout);
        const int t0 = p0 - g0 * L.tout, t1 = p1 - g1 * L.tout;
        const int xo0 = g0 * a.in_sz + q * tpi + t0 * L.stride + kHalo - L.pad_lo;
        const int xo1 = g1 * a.in_sz + q * tpi + t1 * L.stride + kHalo - L.pad_lo;
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int last = nsteps - 1;
        float ar[R];
#pragma unroll
        for (int i = 0; i < R; ++i) ar[i] = w[aidx + min(i, last) * wstep];
        int off = xstep, c4 = 1, j = 0;
        if (C4 == 1) { c4 = 0; off = j = 1; }
        float b0 = xin[xo0], b1 = xin[xo1];
#define TCR_PHASE_STEP(AREG, RELOAD)                                                                    \
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
            for (int i = 0; i < R; ++i) TCR_PHASE_STEP(ar[i], ar[i] = w[aidx + min(s0 + R + i, last) * wstep];)
        }
#pragma unroll
        for (int i = 0; i < R - 1; ++i)
            if (s0 + i < nsteps) TCR_PHASE_STEP(ar[i], )
#undef TCR_PHASE_STEP
        // ---- epilogue: raw output -> global (interior only), per-channel sums of this job's valid positions ----
        const bool v0 = cp * 32 + r < npos, v1 = cp * 32 + 16 + r < npos;
        float s1[4], s2[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float y0 = v0 ? acc0[reg] : 0.f, y1 = v1 ? acc1[reg] : 0.f;
            s1[reg] = y0 + y1;
            s2[reg] = fmaf(y0, y0, y1 * y1);
            const int co = m * 16 + q * 4 + reg;
            if (co < L.cout) {
                if (v0) L.raw[((size_t)(n0 + g0) * L.cout + co) * tpo + kHalo + t0] = acc0[reg];
                if (v1) L.raw[((size_t)(n0 + g1) * L.cout + co) * tpo + kHalo + t1] = acc1[reg];
            }
        }
        if (own) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { ps1[reg] += s1[reg]; ps2[reg] += s2[reg]; }
            continue;
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {       // (DPP row sums: bitwise the xor-butterfly they replace, without its eight LDS-pipe round trips)
            s1[reg] = row16_sum(s1[reg]);
            s2[reg] = row16_sum(s2[reg]);
        }
        if (r == 0) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = m * 16 + q * 4 + reg;
                if (co < L.cout) {
                    wstat[co] += s1[reg];
                    wstat[a.cstat + co] += s2[reg];
                }
            }
        }
    }
    if (own) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            ps1[reg] = row16_sum(ps1[reg]);
            ps2[reg] = row16_sum(ps2[reg]);
        }
        if (r == 0) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = m_own * 16 + q * 4 + reg;
                if (co < L.cout) {
                    wstat[co] += ps1[reg];
                    wstat[a.cstat + co] += ps2[reg];
                }
            }
        }
    }
}

template <int NW, int R>
__global__ __launch_bounds__(NW * 64) void train_phase_kernel(const TrainPhaseArgs a) {
    constexpr int NT = NW * 64;
    float* lds = reinterpret_cast<float*>(dyn_lds());
    float* stat = lds + a.stat_off;                        // [NW][n_layers][2][cstat]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const PhaseSrc& S = a.src;
    const int tp = S.t + 2 * kHalo;
    const int row = S.c * tp;                              // floats per utterance of the source rows
    const float inv_tp = 1.0f / (float)tp;
    const int nstat = NW * a.n_layers * 2 * a.cstat;
    for (int i = tid; i < nstat; i += NT) stat[i] = 0.f;

    for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
        const int n0 = grp * a.group;
        const int ng = min(a.group, a.batch - n0);
        __syncthreads();                                   // (previous group's convolutions are done with the LDS rows)
        // ---- staging: build the input activation of the group (contiguous rows in global memory) ----
        // SU elements per thread and trip: all their loads are issued before the first dependent use (a one-element loop
        // serialises a global round trip per element).
        const size_t gbase = (size_t)n0 * row;
        const int total = ng * row;
        const float inv_row = 1.0f / (float)row;
        constexpr int SU = 4;
        for (int i0 = tid; i0 < total; i0 += NT * SU) {
            float va[SU], vs[SU];
            int ix[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                ix[u] = min(i0 + u * NT, total - 1);
                va[u] = S.a[gbase + ix[u]];
                vs[u] = S.kind == 2 ? S.s[gbase + ix[u]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                if (i0 + u * NT >= total) break;
                const int i = ix[u];
                const int g = fast_div(i, row, inv_row);
                const int rem = i - g * row;
                const int ch = fast_div(rem, tp, inv_tp);
                const int tt = rem - ch * tp - kHalo;
                const bool inside = tt >= 0 && tt < S.t;
                float x = 0.f, sv = 0.f;
                if (S.kind == 0) {
                    x = va[u];                             // plain rows (their halo is already zero)
                } else if (inside) {
                    x = fmaf(va[u], S.ss_a[ch], S.ss_a[S.c_pad_a + ch]);
                    if (S.kind == 2) {
                        sv = S.s_kind == 1 ? fmaxf(fmaf(vs[u], S.ss_s[ch], S.ss_s[S.c_pad_s + ch]), 0.f) : vs[u];
                        x += sv;
                    }
                    x = fmaxf(x, 0.f);
                }
                lds[g * a.in_sz + rem] = x;
                if (S.out_x) S.out_x[gbase + i] = x;
                if (S.kind == 2 && S.s_kind == 1 && S.out_s) S.out_s[gbase + i] = sv;
            }
        }
        if (a.n_layers == 0) continue;
        __syncthreads();
        for (int li = 0; li < a.n_layers; ++li)
            phase_layer<NW, R>(a, a.layer[li], lds, stat + (wave * a.n_layers + li) * 2 * a.cstat, n0, ng, wave, r, q);
    }
    if (a.n_layers == 0) return;
    __syncthreads();
    // ---- one partial row per workgroup and layer: the waves' rows added in wave order ----
    for (int li = 0; li < a.n_layers; ++li) {
        const PhaseLayer& L = a.layer[li];
        for (int i = tid; i < 2 * L.cout; i += NT) {
            const int which = i / L.cout, co = i - which * L.cout;
            float s = 0.f;
            for (int wv = 0; wv < NW; ++wv) s += stat[((wv * a.n_layers + li) * 2 + which) * a.cstat + co];
            L.partial[((size_t)blockIdx.x * 2 + which) * L.cout + co] = s;
        }
    }
}

// fills the launch geometry; false when the phase cannot be configured (caller falls back to the per-layer kernels)
static bool configure_phase(TrainPhaseArgs& a, size_t* lds_out, int* grid_out) {
    const PhaseSrc& S = a.src;
    const int tp = S.t + 2 * kHalo;
    const int in_sz = (S.c * tp + 3) / 4 * 4;
    int cstat = 16;
    for (int i = 0; i < a.n_layers; ++i) {
        if (a.layer[i].cin % 4 != 0 || a.layer[i].cin != S.c || a.layer[i].tin != S.t) return false;
        cstat = max(cstat, (a.layer[i].cout + 15) / 16 * 16);
    }
    if ((int64_t)S.c * tp * 64 >= (1 << 22)) return false;      // (fast_div range of the staging index)
    const int knob = tune_get(TCR_TUNE_PHASE_CFG);
    const int NW = knob / 100 == 4 ? 4 : 8;
    a.nw = NW;
    // Utterances per group: as many as keep the LDS footprint <= 40 KB (3-4 workgroups per CU), at most 8 -- and few enough
    // that the grid still has >= 512 workgroups (one partial row each).
    const size_t stat_bytes = (size_t)NW * max(a.n_layers, 1) * 2 * cstat * sizeof(float);
    int group = knob % 100 > 0 ? knob % 100 : 8;
    while (group > 1 && ((size_t)group * in_sz + 64) * sizeof(float) + stat_bytes > 40 * 1024) --group;
    while (group > 1 && ceil_div(a.batch, group) < 512) --group;
    const size_t lds = ((size_t)group * in_sz + 64) * sizeof(float) + stat_bytes;
    if (lds > 160 * 1024) return false;
    a.group = group; a.n_groups = ceil_div(a.batch, group); a.in_sz = in_sz; a.cstat = cstat;
    a.stat_off = group * in_sz + 64;
    *lds_out = lds;
    *grid_out = min(a.n_groups, kPhaseMaxRows);
    return true;
}

int train_phase_rows(const TrainPhaseArgs& a0) {
    TrainPhaseArgs a = a0;
    size_t lds;
    int grid;
    return configure_phase(a, &lds, &grid) ? grid : -1;
}

int launch_train_phase(TrainPhaseArgs a, int* rows_out, hipStream_t s) {
    size_t lds;
    int grid;
    if (!configure_phase(a, &lds, &grid)) return 1;
    if (rows_out) *rows_out = grid;
    void (*kern)(const TrainPhaseArgs) = a.nw == 4 ? train_phase_kernel<4, 4> : train_phase_kernel<8, 4>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return 1;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(a.nw * 64), lds, s, a);
    return check_launch("train_phase_kernel");
}

}  // namespace tcr
