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

// Timing what-ifs of the compile-time-shaped phases (scripts/build_whatif_src.sh compiles this file with -DTCR_PHASE_WHATIF=<mask> into
// side libraries; WRONG results, never the product build): 1 no MFMAs, 2 weights of tap 0 only, 4 no staging loads, 8 the staged
// activation is not materialised, 16 no raw-output stores, 32 no statistics, 64 no convolutions, 128 no staging.
#ifndef TCR_PHASE_WHATIF
#define TCR_PHASE_WHATIF 0
#endif
#define TCR_PWHATIF(bit) ((TCR_PHASE_WHATIF & (bit)) != 0)
#ifndef TCR_PHASE_SU
#define TCR_PHASE_SU 4        // staged elements per thread and trip
#endif
#if TCR_PHASE_WHATIF & 1
__device__ __forceinline__ f32x4 phase_nomfma(float a, float b, f32x4 c) { c[0] += a; c[1] += b; return c; }
#define TCR_PMFMA(A, B, C) phase_nomfma((A), (B), (C))
#else
#define TCR_PMFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)
#endif

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
        // which of the job's 32 positions this lane's two columns hold: see phase_layer_s
        const int c0 = L.stride == 1 ? cp * 32 + 2 * r : cp * 32 + r, c1 = L.stride == 1 ? c0 + 1 : c0 + 16;
        const int p0 = min(c0, npos - 1), p1 = min(c1, npos - 1);
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
        const bool v0 = c0 < npos, v1 = c1 < npos;
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

// Staging: build the input activation of the group's `ng` utterances (contiguous rows in global memory) in LDS, and materialise it
// for backward.  SU elements per thread and trip: all their loads are issued before the first dependent use (a one-element loop
// serialises a global round trip per element).  tp / row: padded row length and floats per utterance of the source rows.
template <int NT>
__device__ __forceinline__ void phase_stage(const TrainPhaseArgs& a, float* lds, const int n0, const int ng, const int tid, const int tp, const int row) {
    const PhaseSrc& S = a.src;
    const float inv_tp = 1.0f / (float)tp;
    {
        const size_t gbase = (size_t)n0 * row;
        const int total = ng * row;
        const float inv_row = 1.0f / (float)row;
        constexpr int SU = TCR_PHASE_SU;
        for (int i0 = tid; i0 < total; i0 += NT * SU) {
            float va[SU], vs[SU];
            int ix[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                ix[u] = min(i0 + u * NT, total - 1);
                va[u] = TCR_PWHATIF(4) ? 1.f : S.a[gbase + ix[u]];
                vs[u] = S.kind == 2 && !TCR_PWHATIF(4) ? S.s[gbase + ix[u]] : 0.f;
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
                if (S.out_x && !(TCR_PWHATIF(8) && x != 12345.f)) S.out_x[gbase + i] = x;
                if (S.kind == 2 && S.s_kind == 1 && S.out_s && !(TCR_PWHATIF(8) && x != 12345.f)) S.out_s[gbase + i] = sv;
            }
        }
    }
}

// The same staging with 16-byte accesses (round 6).  The group's source rows are ONE contiguous block of the planar layout and the LDS
// image keeps an utterance's rows contiguous too, so a thread moves four consecutive elements at a time: one float4 load per source
// tensor, one 16-byte LDS store, one float4 store per materialised tensor -- against a load, 2-4 coefficient gathers from global memory
// and 2-3 stores PER ELEMENT in phase_stage (the staging was 148 of TCResNet14-1.5's 858 us of training forward with its loads worth
// 17).  The per-channel coefficients (scale / shift of the source's and of the shortcut's BN) come from an LDS table, two 16-byte rows
// per float4 (its channel and the next: tp >= 9, so four elements touch at most two rows).  Same expression per element: bitwise.
template <int NT>
__device__ __forceinline__ void phase_stage_v4(const TrainPhaseArgs& a, float* lds, const int n0, const int ng, const int tid, const int tp, const int row) {
    const PhaseSrc& S = a.src;
    const float inv_tp = 1.0f / (float)tp, inv_row = 1.0f / (float)row;
    const size_t gbase = (size_t)n0 * row;
    const int tot4 = ng * row / 4;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(S.a + gbase);
    const f32x4* s4 = S.kind == 2 ? reinterpret_cast<const f32x4*>(S.s + gbase) : nullptr;
    f32x4* ox4 = S.out_x ? reinterpret_cast<f32x4*>(S.out_x + gbase) : nullptr;
    f32x4* os4 = (S.kind == 2 && S.s_kind == 1 && S.out_s) ? reinterpret_cast<f32x4*>(S.out_s + gbase) : nullptr;
    const float* tab = lds + a.tab_off;
    const bool lds_v4 = (a.in_sz & 3) == 0;
    constexpr int SV = 2;
    for (int f0 = tid; f0 < tot4; f0 += NT * SV) {
        f32x4 va[SV], vs[SV];
#pragma unroll
        for (int u = 0; u < SV; ++u) {
            const int f = min(f0 + u * NT, tot4 - 1);
            va[u] = TCR_PWHATIF(4) ? (f32x4){1.f, 1.f, 1.f, 1.f} : a4[f];
            vs[u] = (s4 && !TCR_PWHATIF(4)) ? s4[f] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < SV; ++u) {
            const int f = f0 + u * NT;
            if (f >= tot4) break;
            const int i0 = 4 * f;
            const int g = fast_div(i0, row, inv_row);
            const int rem = i0 - g * row;                       // (row % 4 == 0: the four elements stay inside one utterance)
            const int ch = fast_div(rem, tp, inv_tp);
            f32x4 ka = (f32x4){0.f, 0.f, 0.f, 0.f}, kb = ka;
            if (S.kind != 0) {
                ka = *reinterpret_cast<const f32x4*>(tab + ch * 4);
                kb = *reinterpret_cast<const f32x4*>(tab + min(ch + 1, S.c - 1) * 4);
            }
            f32x4 xo, so;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool nx = rem + e >= (ch + 1) * tp;
                const int tt = rem + e - (nx ? ch + 1 : ch) * tp - kHalo;
                const bool inside = tt >= 0 && tt < S.t;
                const float sc_a = nx ? kb[0] : ka[0], sh_a = nx ? kb[1] : ka[1], sc_s = nx ? kb[2] : ka[2], sh_s = nx ? kb[3] : ka[3];
                float x = 0.f, sv = 0.f;
                if (S.kind == 0) {
                    x = va[u][e];                               // plain rows (their halo is already zero)
                } else if (inside) {
                    x = fmaf(va[u][e], sc_a, sh_a);
                    if (S.kind == 2) {
                        sv = S.s_kind == 1 ? fmaxf(fmaf(vs[u][e], sc_s, sh_s), 0.f) : vs[u][e];
                        x += sv;
                    }
                    x = fmaxf(x, 0.f);
                }
                xo[e] = x;
                so[e] = sv;
            }
            if (lds_v4) {
                *reinterpret_cast<f32x4*>(lds + g * a.in_sz + rem) = xo;
            } else {                                            // (an utterance pitch padded to the bank pattern is not always a multiple of 4 floats)
                float* dl = lds + g * a.in_sz + rem;
                dl[0] = xo[0]; dl[1] = xo[1]; dl[2] = xo[2]; dl[3] = xo[3];
            }
            if (ox4 && !(TCR_PWHATIF(8) && xo[0] != 12345.f)) ox4[f] = xo;
            if (os4 && !(TCR_PWHATIF(8) && xo[0] != 12345.f)) os4[f] = so;
        }
    }
}

// the staging table of phase_stage_v4: [c][4] = scale / shift of the source's BN, scale / shift of the shortcut's BN
template <int NT>
__device__ __forceinline__ void phase_fill_table(const TrainPhaseArgs& a, float* lds, const int tid) {
    const PhaseSrc& S = a.src;
    if (!a.vec_stage || S.kind == 0) return;
    float* tab = lds + a.tab_off;
    const bool sbn = S.kind == 2 && S.s_kind == 1;
    for (int i = tid; i < S.c; i += NT) {
        tab[i * 4 + 0] = S.ss_a[i];
        tab[i * 4 + 1] = S.ss_a[S.c_pad_a + i];
        tab[i * 4 + 2] = sbn ? S.ss_s[i] : 0.f;
        tab[i * 4 + 3] = sbn ? S.ss_s[S.c_pad_s + i] : 0.f;
    }
}

// The same layer with its shape known at COMPILE time (the phases of TCResNet8-1.0 and TCResNet14-1.5 at 49 / 98 frames), the K loop
// of the eval kernel's static layer (fused.hip: fused_layer_s): taps rolled, a tap's weight fragments in two half-tap register sets that
// are refilled for the next tap behind the other half's MFMAs, LDS operands at immediate offsets, division by constants.  Same job ->
// wave dealing, same accumulation order, same order of the statistics sums: bitwise the generic layer.
// Columns: one ds_read_b32 serves input channels (q, q + 1) x 16 columns against 32 banks.  With stride 2 consecutive positions are 2
// floats apart and the odd row pitch puts channel q + 1 on the other bank parity; with stride 1 a tile's columns are every OTHER position
// (tile 0 the even, tile 1 the odd ones of the job's 32) for the same picture; in_sz is padded so that the pattern runs on into the
// next utterance of the group (configure_phase).
template <int NW, int K, int S, int CIN, int COUT, int TIN>
__device__ __forceinline__ void phase_layer_s(const TrainPhaseArgs& a, const PhaseLayer& L, const float* __restrict__ xin, float* wstat,
                                              const int n0, const int ng, const int wave, const int r_in, const int q_in) {
    const int oz = opaque_zero();          // (per-lane address arithmetic stays inside the group loop)
    const int r = r_in + oz, q = q_in + oz;
    constexpr int TOUT = (TIN + S - 1) / S;
    constexpr int PADT = ((TOUT - 1) * S + K - TIN) > 0 ? ((TOUT - 1) * S + K - TIN) : 0;
    constexpr int PADLO = PADT / 2;
    constexpr int TPI = TIN + 2 * kHalo, TPO = TOUT + 2 * kHalo;
    constexpr int C4 = CIN / 4, NRT = (COUT + 15) / 16;
    constexpr int WSTEP = 4 * COUT, XSTEP = 4 * TPI;
    constexpr bool IL = S == 1;
    constexpr bool OWN = (NW % NRT) == 0;
    constexpr int CP_STEP = OWN ? NW / NRT : 0;
    static_assert(CIN % 4 == 0, "channel quads");
    const int npos = ng * TOUT;
    const int ncp = (npos + 31) / 32;
    const float* w = a.params + L.w_off;
    const int in_sz = a.in_sz;
    float ps1[4] = {0.f, 0.f, 0.f, 0.f}, ps2[4] = {0.f, 0.f, 0.f, 0.f};
    const int m_own = wave % NRT;
    for (int it = 0;; ++it) {
        int cp, m;
        if (OWN) { cp = wave / NRT + it * CP_STEP; m = m_own; if (cp >= ncp) break; }
        else { const int job = wave + it * NW; if (job >= ncp * NRT) break; cp = job / NRT; m = job - cp * NRT; }
        const int c0 = IL ? cp * 32 + 2 * r : cp * 32 + r, c1 = IL ? c0 + 1 : c0 + 16;
        const int p0 = min(c0, npos - 1), p1 = min(c1, npos - 1);
        const int g0 = p0 / TOUT, g1 = p1 / TOUT;
        const int t0 = p0 - g0 * TOUT, t1 = p1 - g1 * TOUT;
        const float* wp = w + q * COUT + min(m * 16 + r, COUT - 1);
        const float* x0 = xin + g0 * in_sz + q * TPI + t0 * S + kHalo - PADLO;
        const float* x1 = xin + g1 * in_sz + q * TPI + t1 * S + kHalo - PADLO;
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifndef TCR_PHASE_AHEAD
#define TCR_PHASE_AHEAD 8           // input-channel quads up to which a nine-tap layer's taps run with a whole tap of weight lookahead (0: never)
#endif
        // Which layers (measured on the bench's training LEGS, where the next batch's front-end -- 168 registers x 3 waves per SIMD --
        // shares the CUs; scripts/ab_libs_train_leg.py): the nine-tap layers of <= 32 input channels (<= 81 registers: a phase workgroup
        // still fits beside two front-end waves per SIMD): TCResNet8 at 49 frames 830 - 841 -> 812 - 817 us per step of the leg, 98 frames
        // 1283 -> 1266, TCResNet14-1.5 within noise (2415 - 2425; 98 frames 3762 -> 3752 - 3760).  36 / 48 / 72 input channels: 76 - 92
        // registers for no gain in any leg.  The three-tap first conv (40 channels) with the lookahead runs at 90 registers at 49 frames
        // and costs the TCResNet8 leg 40 us there (857) although the step ALONE does not move; at 98 frames (82 registers) it is
        // worth ~10 us to both nets' legs: kept for that shape only.
        constexpr bool kAhead = (K >= 9 && C4 <= TCR_PHASE_AHEAD) || (K == 3 && TIN >= 98 && TCR_PHASE_AHEAD > 0);
        if constexpr (kAhead && !TCR_PWHATIF(2)) {
            // A whole tap of weight lookahead (the eval kernel's round-6 form, fused.hip: fused_job_s AHEAD): in the rolled loop below the
            // compiler sinks a half-tap's refill loads to the END of the tap body and waits for them in front of the next tap's first MFMA,
            // so with 8 - 16 MFMAs per tap every tap waits out an L1 / L2 round trip.  Two full-tap register sets that trade roles, the
            // next tap's fragments requested through a buffer descriptor (uniform base + constant lane offset + uniform tap offset) before
            // this tap's LDS reads and MFMAs.  Same accumulation order (quads ascending, tile 0 then tile 1): bitwise the rolled loop.
            const buf_rsrc wr = make_rsrc(w);
            const unsigned wl = (unsigned)(q * COUT + min(m * 16 + r, COUT - 1)) * 4u;
            float w0[C4], w1[C4];
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4) w0[c4] = buf_load_f32(wr, wl, (unsigned)(c4 * WSTEP) * 4u);
            auto tap = [&](const int j, const float (&wu)[C4], float (&wf)[C4], const bool fill) {
                if (fill) {
#pragma unroll
                    for (int c4 = 0; c4 < C4; ++c4) wf[c4] = buf_load_f32(wr, wl, (unsigned)(((j + 1) * C4 + c4) * WSTEP) * 4u);
                }
                float b0[C4], b1[C4];
#pragma unroll
                for (int c4 = 0; c4 < C4; ++c4) { b0[c4] = x0[c4 * XSTEP + j]; b1[c4] = x1[c4 * XSTEP + j]; }
                __builtin_amdgcn_sched_barrier(0);              // (the requests stay in front of the tap's MFMAs)
#pragma unroll
                for (int c4 = 0; c4 < C4; ++c4) {
                    acc0 = TCR_PMFMA(wu[c4], b0[c4], acc0);
                    acc1 = TCR_PMFMA(wu[c4], b1[c4], acc1);
                }
            };
#pragma unroll 1
            for (int j = 0; j + 2 < K; j += 2) {
                tap(j, w0, w1, true);
                tap(j + 1, w1, w0, true);
            }
            if constexpr (K % 2 == 1) {
                tap(K - 1, w0, w1, false);
            } else {
                tap(K - 2, w0, w1, true);
                tap(K - 1, w1, w0, false);
            }
        } else {
        constexpr int H0 = C4 / 2, H1 = C4 - H0;
        float wa[H0 > 0 ? H0 : 1], wb[H1];
#pragma unroll
        for (int c4 = 0; c4 < H0; ++c4) wa[c4] = wp[c4 * WSTEP];
#pragma unroll
        for (int c4 = 0; c4 < H1; ++c4) wb[c4] = wp[(H0 + c4) * WSTEP];
#pragma unroll 1
        for (int j = 0; j < K; ++j) {
            const int jn = TCR_PWHATIF(2) ? 0 : min(j + 1, K - 1);                           // (the last tap refills with itself: branch-free)
            {
                float b0[H0 > 0 ? H0 : 1], b1[H0 > 0 ? H0 : 1];
#pragma unroll
                for (int c4 = 0; c4 < H0; ++c4) { b0[c4] = x0[c4 * XSTEP + j]; b1[c4] = x1[c4 * XSTEP + j]; }
#pragma unroll
                for (int c4 = 0; c4 < H0; ++c4) {
                    acc0 = TCR_PMFMA(wa[c4], b0[c4], acc0);
                    acc1 = TCR_PMFMA(wa[c4], b1[c4], acc1);
                }
#pragma unroll
                for (int c4 = 0; c4 < H0; ++c4) wa[c4] = wp[(jn * C4 + c4) * WSTEP];
            }
            {
                float b0[H1], b1[H1];
#pragma unroll
                for (int c4 = 0; c4 < H1; ++c4) { b0[c4] = x0[(H0 + c4) * XSTEP + j]; b1[c4] = x1[(H0 + c4) * XSTEP + j]; }
#pragma unroll
                for (int c4 = 0; c4 < H1; ++c4) {
                    acc0 = TCR_PMFMA(wb[c4], b0[c4], acc0);
                    acc1 = TCR_PMFMA(wb[c4], b1[c4], acc1);
                }
#pragma unroll
                for (int c4 = 0; c4 < H1; ++c4) wb[c4] = wp[(jn * C4 + H0 + c4) * WSTEP];
            }
        }
        }
        // ---- epilogue: raw output -> global (interior only), per-channel sums of this job's valid positions ----
        const bool v0 = c0 < npos, v1 = c1 < npos;
        float s1[4], s2[4];
        float* o0 = L.raw + ((size_t)(n0 + g0) * COUT + m * 16 + q * 4) * TPO + kHalo + t0;
        float* o1 = L.raw + ((size_t)(n0 + g1) * COUT + m * 16 + q * 4) * TPO + kHalo + t1;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float y0 = v0 ? acc0[reg] : 0.f, y1 = v1 ? acc1[reg] : 0.f;
            s1[reg] = y0 + y1;
            s2[reg] = fmaf(y0, y0, y1 * y1);
            if (COUT % 16 == 0 || m * 16 + q * 4 + reg < COUT) {
                if (v0 && !(TCR_PWHATIF(16) && acc0[reg] != 12345.f)) o0[reg * TPO] = acc0[reg];
                if (v1 && !(TCR_PWHATIF(16) && acc1[reg] != 12345.f)) o1[reg * TPO] = acc1[reg];
            }
        }
        if (TCR_PWHATIF(32)) continue;
        if (OWN) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) { ps1[reg] += s1[reg]; ps2[reg] += s2[reg]; }
            continue;
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            s1[reg] = row16_sum(s1[reg]);
            s2[reg] = row16_sum(s2[reg]);
        }
        if (r == 0) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = m * 16 + q * 4 + reg;
                if (COUT % 16 == 0 || co < COUT) {
                    wstat[co] += s1[reg];
                    wstat[a.cstat + co] += s2[reg];
                }
            }
        }
    }
    if (OWN) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            ps1[reg] = row16_sum(ps1[reg]);
            ps2[reg] = row16_sum(ps2[reg]);
        }
        if (r == 0) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = m_own * 16 + q * 4 + reg;
                if (COUT % 16 == 0 || co < COUT) {
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
    const int nstat = NW * a.n_layers * 2 * a.cstat;
    for (int i = tid; i < nstat; i += NT) stat[i] = 0.f;
    phase_fill_table<NT>(a, lds, tid);

    for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
        const int n0 = grp * a.group;
        const int ng = min(a.group, a.batch - n0);
        __syncthreads();                                   // (previous group's convolutions are done with the LDS rows; the tables are staged)
        if (a.vec_stage) phase_stage_v4<NT>(a, lds, n0, ng, tid, tp, row);
        else phase_stage<NT>(a, lds, n0, ng, tid, tp, row);
        if (a.n_layers == 0) {
            // Closing phase, round 6: the block output sits in LDS -- its sums over time, frames in order (the head's own order: bitwise its
            // pooling), leave with it, so that head_fwd_kernel starts from [B][C] instead of walking C x T scattered rows per utterance
            // on 64 workgroups (22-29 us at the joint of forward and backward).
            if (a.pool_sum) {
                __syncthreads();
                for (int i = tid; i < ng * S.c; i += NT) {
                    const int g = i / S.c, ch = i - g * S.c;
                    const float* rowp = lds + g * a.in_sz + ch * tp + kHalo;
                    float sum = 0.f;
                    for (int t = 0; t < S.t; ++t) sum += rowp[t];
                    a.pool_sum[(size_t)(n0 + g) * S.c + ch] = sum;
                }
            }
            continue;
        }
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

// A phase with compile-time layer shapes: CIN x TIN input rows, one or two (K1 > 0) convolutions over them (the host lists a block's
// shortcut conv first).  Everything around the layers is the generic kernel's.
template <int NW, int CIN, int TIN, int K0, int S0, int CO0, int K1, int S1, int CO1>
__global__ __launch_bounds__(NW * 64) void train_phase_s_kernel(const TrainPhaseArgs a) {
    constexpr int NT = NW * 64;
    constexpr int NL = K1 > 0 ? 2 : 1;
    constexpr int tp = TIN + 2 * kHalo, row = CIN * tp;
    float* lds = reinterpret_cast<float*>(dyn_lds());
    float* stat = lds + a.stat_off;                        // [NW][n_layers][2][cstat]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int nstat = NW * NL * 2 * a.cstat;
    for (int i = tid; i < nstat; i += NT) stat[i] = 0.f;
    phase_fill_table<NT>(a, lds, tid);
    for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
        const int n0 = grp * a.group;
        const int ng = min(a.group, a.batch - n0);
        __syncthreads();
        if (!TCR_PWHATIF(128)) {
            if (a.vec_stage) phase_stage_v4<NT>(a, lds, n0, ng, tid, tp, row);
            else phase_stage<NT>(a, lds, n0, ng, tid, tp, row);
        }
        __syncthreads();
        if (TCR_PWHATIF(64)) continue;
        phase_layer_s<NW, K0, S0, CIN, CO0, TIN>(a, a.layer[0], lds, stat + (wave * NL + 0) * 2 * a.cstat, n0, ng, wave, r, q);
        if constexpr (K1 > 0) phase_layer_s<NW, K1, S1, CIN, CO1, TIN>(a, a.layer[1], lds, stat + (wave * NL + 1) * 2 * a.cstat, n0, ng, wave, r, q);
    }
    __syncthreads();
    for (int li = 0; li < NL; ++li) {
        const PhaseLayer& L = a.layer[li];
        for (int i = tid; i < 2 * L.cout; i += NT) {
            const int which = i / L.cout, co = i - which * L.cout;
            float s = 0.f;
            for (int wv = 0; wv < NW; ++wv) s += stat[((wv * NL + li) * 2 + which) * a.cstat + co];
            L.partial[((size_t)blockIdx.x * 2 + which) * L.cout + co] = s;
        }
    }
}

typedef void (*PhaseKernel)(const TrainPhaseArgs);
// the compile-time instance for this phase (8 waves), or nullptr
static PhaseKernel static_phase_kernel(const TrainPhaseArgs& a) {
    if (a.nw != 8 || a.n_layers < 1 || (tune_get(TCR_TUNE_PHASE_STATIC) & 1)) return nullptr;
    const PhaseLayer& A = a.layer[0];
    const PhaseLayer& B = a.layer[1];
    const int k1 = a.n_layers == 2 ? B.k : 0, s1 = a.n_layers == 2 ? B.stride : 0, co1 = a.n_layers == 2 ? B.cout : 0;
#define TCR_PS(CIN_, TIN_, K0_, S0_, CO0_, K1_, S1_, CO1_)                                                                         \
    if (a.src.c == CIN_ && a.src.t == TIN_ && A.k == K0_ && A.stride == S0_ && A.cout == CO0_ && k1 == K1_ && s1 == S1_ && co1 == CO1_) \
        return train_phase_s_kernel<8, CIN_, TIN_, K0_, S0_, CO0_, K1_, S1_, CO1_>;
#define TCR_PS_NET(T0_, C0_, C1_, C2_, C3_)                                                                                        \
    TCR_PS(40, T0_, 3, 1, C0_, 0, 0, 0)                                                                                            \
    TCR_PS(C0_, T0_, 1, 2, C1_, 9, 2, C1_) TCR_PS(C1_, (T0_ + 1) / 2, 9, 1, C1_, 0, 0, 0)                                                 \
    TCR_PS(C1_, (T0_ + 1) / 2, 1, 2, C2_, 9, 2, C2_)                                                                               \
    TCR_PS(C2_, ((T0_ + 1) / 2 + 1) / 2, 9, 1, C2_, 0, 0, 0)                                                                        \
    TCR_PS(C2_, ((T0_ + 1) / 2 + 1) / 2, 1, 2, C3_, 9, 2, C3_)                                                                      \
    TCR_PS(C3_, (((T0_ + 1) / 2 + 1) / 2 + 1) / 2, 9, 1, C3_, 0, 0, 0)
    TCR_PS_NET(49, 16, 24, 32, 48)          // TCResNet8-1.0 (BASELINE.json configs[2])
    TCR_PS_NET(98, 16, 24, 32, 48)
    TCR_PS_NET(49, 24, 36, 48, 72)          // TCResNet14-1.5 (configs[3]): the identity blocks' convs have the shapes of the second convs
    TCR_PS_NET(98, 24, 36, 48, 72)
#undef TCR_PS_NET
#undef TCR_PS
    return nullptr;
}

// fills the launch geometry; false when the phase cannot be configured (caller falls back to the per-layer kernels)
static bool configure_phase(TrainPhaseArgs& a, size_t* lds_out, int* grid_out) {
    const PhaseSrc& S = a.src;
    const int tp = S.t + 2 * kHalo;
    // floats per utterance in LDS: 2 * T_out (stride 2) or T_in (stride 1) modulo 32, so that the column -> bank pattern of the
    // layers' operand reads runs on unbroken into the next utterance of the group (phase_layer_s)
    int in_sz = (S.c * tp + 3) / 4 * 4;
    if (a.n_layers > 0 && !(tune_get(TCR_TUNE_PHASE_STATIC) & 2)) {
        const int want = a.layer[0].stride == 2 ? 2 * a.layer[0].tout : a.layer[0].tin;
        while ((in_sz - want) % 32 != 0) ++in_sz;
    }
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
    // (Round 6: 4 instead of 8 for phases of more than 64 output channels -- TCResNet14-1.5's 72-channel blocks: twice the workgroups, four
    //  per CU in different parts of a phase instead of two -- 98 frames: 4027 -> 3924 us per step, 49 frames 2494 -> 2494; 4 for EVERY
    //  phase: 3934 / 2472, i.e. the gain is the wide phases'; 3 / 5 / 6: slower.  TCResNet8: 8 stays -- 4: 806 vs 785 us.)
    int group = knob % 100 > 0 ? knob % 100 : (cstat >= 80 ? 4 : 8);
    while (group > 1 && ((size_t)group * in_sz + 64) * sizeof(float) + stat_bytes > 40 * 1024) --group;
    while (group > 1 && ceil_div(a.batch, group) < 512) --group;
    const size_t tab_bytes = (size_t)(S.c * 4 + 4) * sizeof(float);                 // phase_stage_v4's coefficient table, behind the statistics
    const size_t lds = ((size_t)group * in_sz + 64) * sizeof(float) + stat_bytes + tab_bytes;
    if (lds > 160 * 1024) return false;
    a.group = group; a.n_groups = ceil_div(a.batch, group); a.in_sz = in_sz; a.cstat = cstat;
    a.stat_off = group * in_sz + 64;
    a.tab_off = (a.stat_off + NW * max(a.n_layers, 1) * 2 * cstat + 3) / 4 * 4;
    auto al16 = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    // 16-byte staging: an utterance's rows are whole float4s and every tensor it touches starts on one (TCR_TUNE_PHASE_STATIC bit 2: the
    // one-element loop, A/B arm)
    a.vec_stage = ((S.c * tp) % 4 == 0 && tp >= 9 && al16(S.a) && al16(S.kind == 2 ? S.s : nullptr) && al16(S.out_x) &&
                   al16((S.kind == 2 && S.s_kind == 1) ? S.out_s : nullptr) && !(tune_get(TCR_TUNE_PHASE_STATIC) & 4)) ? 1 : 0;
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
    PhaseKernel kern = static_phase_kernel(a);
    if (!kern) kern = a.nw == 4 ? train_phase_kernel<4, 4> : train_phase_kernel<8, 4>;
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
