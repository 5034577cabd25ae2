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
#include <cstdint>

#include "kernels.h"

namespace tcr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Timing what-ifs of the static-shape kernel (scripts/build_whatif.py compiles this file with -DTCR_FUSED_WHATIF=<mask> into
// side libraries; WRONG results, never part of the product build): 1 no MFMAs, 2 weights of tap 0 only, 4 LDS operands of tap 0
// only, 8 no barriers, 16 no head, 32 no epilogue stores, 64 no first conv, 512 empty kernel, 1024 no halo-zero passes, 2048 phase stamps
// (scripts/fused_ts.py), 4096 / 8192 the lookahead layers without their weight refills / without LDS operand reads after tap 0,
// 16384 / 32768 (right results) s_setprio 2 around a tap's MFMA burst / around its operand requests: 110 / 101 us against 98 -- slower, not built in.
#ifndef TCR_FUSED_WHATIF
#define TCR_FUSED_WHATIF 0
#endif
#define TCR_WHATIF(bit) ((TCR_FUSED_WHATIF & (bit)) != 0)
#if TCR_FUSED_WHATIF & (16384 | 32768)
#define TCR_WHATIF_PRIO(bit, level) do { if (TCR_WHATIF(bit)) __builtin_amdgcn_s_setprio(level); } while (0)
#else
#define TCR_WHATIF_PRIO(bit, level) do { } while (0)       // (the host build of tests/emu has no such builtin)
#endif
#if TCR_FUSED_WHATIF & 1
__device__ __forceinline__ f32x4 whatif_nomfma(float a, float b, f32x4 c) { c[0] += a; c[1] += b; return c; }
#define TCR_MFMA(A, B, C) whatif_nomfma((A), (B), (C))
#else
#define TCR_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)
#endif

// NW: wavefronts per workgroup (jobs of a layer are dealt round-robin to them).  R: depth of the weight register ring.
// One layer of the walk for the group's `ng` utterances.  xin: input rows (LDS buffer, or -- first layer -- the feature
// rows in global memory, which then never occupy LDS), in_sz floats per utterance.
template <int NW, int R>
__device__ __forceinline__ void fused_layer(const FusedArgs& a, const FusedLayer& L, const float* __restrict__ xin, const int in_sz,
                                            float* lds, const int ng, const int wave, const int r, const int q) {
    const int tpi = L.tin + 2 * kHalo, tpo = L.tout + 2 * kHalo;
    float* yout = lds + a.buf_off[L.out_buf];
    const float* res = L.res_buf >= 0 ? lds + a.buf_off[L.res_buf] : nullptr;
    const int out_sz = L.out_sz;
    const int res_sz = L.res_sz;
    const int npos = ng * L.tout;
    const int ncp = (npos + 31) / 32;               // column pairs (32 positions)
    const int nrt = (L.cout + 15) / 16;             // row tiles (16 output channels)
    const int C4 = L.cin >> 2;
    const int nsteps = L.k * C4;
    const float* w = a.params + L.w_off;
    const float* scale = a.ss + L.ss_off;
    const float* shift = scale + L.c_pad;
    const int wstep = 4 * L.cout;                   // weight floats per K-step (4 input channels of one tap)
    const int xstep = 4 * tpi;                      // LDS floats per K-step within a tap
    // position -> (utterance of the group, frame) without an integer division: (p + 0.5) / tout is never within
    // float round-off of an integer for the p < 2^16 that occur here
    const float inv_tout = 1.0f / (float)L.tout;
    for (int job = wave; job < ncp * nrt; job += NW) {
        const int cp = job / nrt, m = job - cp * nrt;
        // A-fragment element of this lane: W[j][4*c4 + q][16*m + r]; rows beyond Cout are clamped (never stored)
        const int aidx = q * L.cout + min(m * 16 + r, L.cout - 1);
        const int p0 = min(cp * 32 + r, npos - 1), p1 = min(cp * 32 + 16 + r, npos - 1);
        const int g0 = (int)(((float)p0 + 0.5f) * inv_tout), g1 = (int)(((float)p1 + 0.5f) * inv_tout);
        const int t0 = p0 - g0 * L.tout, t1 = p1 - g1 * L.tout;
        const int xo0 = g0 * in_sz + q * tpi + t0 * L.stride + kHalo - L.pad_lo;
        const int xo1 = g1 * in_sz + q * tpi + t1 * L.stride + kHalo - L.pad_lo;
        // folded-BN scale / shift of this lane's 4 output channels: fetched now, consumed after the K loop
        float sc[4], sh[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int co = min(m * 16 + q * 4 + reg, L.cout - 1);
            sc[reg] = scale[co];
            sh[reg] = shift[co];
        }
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        // K-steps s = (tap j, channel quad c4), tap-major.  Weights come from L1/L2 through a 4-deep register
        // ring (uniform base + 32-bit lane offset, clamped instead of branched at the tail); activations are read
        // from the LDS rows at xo + off, where off = c4 * xstep + j is advanced with scalar increments.
        const int last = nsteps - 1;
        float ar[R];
#pragma unroll
        for (int i = 0; i < R; ++i) ar[i] = w[aidx + min(i, last) * wstep];
        // LDS operands are read one step ahead of the MFMAs that consume them (b0/b1 = current step,
        // nb0/nb1 = next step), so the matrix pipe never waits on a ds_read it has just issued.
        int off = xstep, c4 = 1, j = 0;
        if (C4 == 1) { c4 = 0; off = j = 1; }
        float b0 = xin[xo0], b1 = xin[xo1];
#define TCR_FUSED_STEP(AREG, RELOAD)                                                                    \
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
            for (int i = 0; i < R; ++i) TCR_FUSED_STEP(ar[i], ar[i] = w[aidx + min(s0 + R + i, last) * wstep];)
        }
#pragma unroll
        for (int i = 0; i < R - 1; ++i)
            if (s0 + i < nsteps) TCR_FUSED_STEP(ar[i], )
#undef TCR_FUSED_STEP
        // ---- epilogue, branch-free (round 3, as in fused_layer_s): lanes past the last position / past Cout store into the pad behind
        // the buffers, the residual reads use clamped addresses, the halo zeros come from fused_zero_halo_g ----
        const int dump = a.buf_off[2] + a.group * a.buf_sz[2] + (q * 16 + r) - a.buf_off[L.out_buf];
        const float lo = L.relu ? 0.f : -3.4e38f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const bool pv = cp * 32 + nt * 16 + r < npos;
            const int g = nt == 0 ? g0 : g1, t = nt == 0 ? t0 : t1;
            const f32x4 ac = nt == 0 ? acc0 : acc1;
            const int base = g * out_sz + kHalo + t;
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (res) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) rv[reg] = res[g * res_sz + min(m * 16 + q * 4 + reg, L.cout - 1) * tpo + kHalo + t];
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = m * 16 + q * 4 + reg;
                float v = fmaf(ac[reg], sc[reg], sh[reg]);
                v = res ? fmaxf(v + rv[reg], 0.f) : fmaxf(v, lo);      // tc_resnet.py:40-41
                yout[(pv && co < L.cout) ? base + co * tpo : dump] = v;
            }
        }
    }
}

// The same layer with its shape known at COMPILE time (the flagship configurations: TCResNet8-1.0 at 49 and 98 frames).
// Every K-step of the generic walk above costs ~14 scalar / vector instructions next to its two MFMAs -- step counters, the
// (tap, channel-quad) -> offset bookkeeping, clamped weight indices -- and with 3-8 jobs per barrier phase a wave cannot keep
// the matrix pipe busy at that ratio.  With K, stride, Cin, Cout and T as template parameters the K loop unrolls, the LDS
// operands are read at immediate offsets (c4 * 4 * Tp + tap), the weights at immediate offsets from one per-tap base, and the
// position -> (utterance, frame) split divides by a constant.  Same accumulation order (tap-major, channel quads inner): the
// result is bitwise the generic layer's.
template <int NW, int K, int S, int CIN, int COUT, int TIN>
__device__ __forceinline__ void fused_layer_t(const FusedArgs& a, const FusedLayer& L, const float* __restrict__ xin, const int in_sz,
                                              float* lds, const int ng, const int wave, const int r, const int q) {
    constexpr int TOUT = (TIN + S - 1) / S;
    constexpr int PADT = ((TOUT - 1) * S + K - TIN) > 0 ? ((TOUT - 1) * S + K - TIN) : 0;
    constexpr int PADLO = PADT / 2;
    constexpr int TPI = TIN + 2 * kHalo, TPO = TOUT + 2 * kHalo;
    constexpr int C4 = CIN / 4, NRT = (COUT + 15) / 16;
    constexpr int WSTEP = 4 * COUT, XSTEP = 4 * TPI;
    static_assert(CIN % 4 == 0, "channel quads");
    float* yout = lds + a.buf_off[L.out_buf];
    const float* res = L.res_buf >= 0 ? lds + a.buf_off[L.res_buf] : nullptr;
    const int out_sz = L.out_sz;
    const int res_sz = L.res_sz;
    const int npos = ng * TOUT;
    const int ncp = (npos + 31) / 32;
    const float* w = a.params + L.w_off;
    const float* scale = a.ss + L.ss_off;
    const float* shift = scale + L.c_pad;
    for (int job = wave; job < ncp * NRT; job += NW) {
        const int cp = job / NRT, m = job - cp * NRT;
        const int p0 = min(cp * 32 + r, npos - 1), p1 = min(cp * 32 + 16 + r, npos - 1);
        const int g0 = p0 / TOUT, g1 = p1 / TOUT;
        const int t0 = p0 - g0 * TOUT, t1 = p1 - g1 * TOUT;
        const float* wp = w + q * COUT + min(m * 16 + r, COUT - 1);
        const float* x0 = xin + g0 * in_sz + q * TPI + t0 * S + kHalo - PADLO;
        const float* x1 = xin + g1 * in_sz + q * TPI + t1 * S + kHalo - PADLO;
        float sc[4], sh[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int co = min(m * 16 + q * 4 + reg, COUT - 1);
            sc[reg] = scale[co];
            sh[reg] = shift[co];
        }
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Taps stay a loop (one pointer increment each; unrolling taps AND channel quads lets the scheduler hoist all K * C4 operand
        // loads: 207 VGPRs, half the occupancy).  The weight fragments of a tap live in two half-tap register sets; a set is refilled
        // for the NEXT tap as soon as its MFMAs are issued, so the (L1 / L2) weight loads fly behind the other half's MFMAs instead of
        // in front of the tap (timing what-if: weights fetched once per job would save 12 % of the kernel).
        constexpr int H0 = C4 / 2, H1 = C4 - H0;
        float wa[H0 > 0 ? H0 : 1], wb[H1];
#pragma unroll
        for (int c4 = 0; c4 < H0; ++c4) wa[c4] = wp[c4 * WSTEP];
#pragma unroll
        for (int c4 = 0; c4 < H1; ++c4) wb[c4] = wp[(H0 + c4) * WSTEP];
#pragma unroll 1
        for (int j = 0; j < K; ++j) {
            const int jn = min(j + 1, K - 1);                           // (the last tap refills with itself: branch-free)
            {
                float b0[H0 > 0 ? H0 : 1], b1[H0 > 0 ? H0 : 1];
#pragma unroll
                for (int c4 = 0; c4 < H0; ++c4) { b0[c4] = x0[c4 * XSTEP + j]; b1[c4] = x1[c4 * XSTEP + j]; }
#pragma unroll
                for (int c4 = 0; c4 < H0; ++c4) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c4], b0[c4], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c4], b1[c4], acc1, 0, 0, 0);
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
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[c4], b0[c4], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[c4], b1[c4], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int c4 = 0; c4 < H1; ++c4) wb[c4] = wp[(jn * C4 + H0 + c4) * WSTEP];
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (cp * 32 + nt * 16 + r >= npos) continue;
            const int g = nt == 0 ? g0 : g1, t = nt == 0 ? t0 : t1;
            const f32x4 ac = nt == 0 ? acc0 : acc1;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = m * 16 + q * 4 + reg;
                if (co >= COUT) continue;
                float v = fmaf(ac[reg], sc[reg], sh[reg]);
                if (res) v = fmaxf(v + res[g * res_sz + co * TPO + kHalo + t], 0.f);
                else if (L.relu) v = fmaxf(v, 0.f);
                float* dst = yout + g * out_sz + co * TPO + kHalo + t;
                dst[0] = v;
                if (t == 0) { dst[-4] = 0.f; dst[-3] = 0.f; dst[-2] = 0.f; dst[-1] = 0.f; }
                if (t == TOUT - 1) { dst[1] = 0.f; dst[2] = 0.f; dst[3] = 0.f; dst[4] = 0.f; }
            }
        }
    }
}

// Round 3: the static-shape layer with a branch-free epilogue.  The round-2 epilogue above spends ~60 instructions and six exec-mask
// branches per output element (bounds tests, the residual read behind a branch, the halo zeros written by whichever lane owns a row's
// first / last frame).  Here lanes past the group's last position / past Cout store into the pad behind the buffers, the residual
// reads are issued together with clamped addresses, and the halo zeros of the output rows are written by `fused_zero_halo` (one pass
// per layer): 17.5 M instead of 21.3 M VALU and 4.9 M instead of 11 M SALU instructions per launch, bitwise the same results.
// (Tried with it and dropped: the taps fully unrolled into a software pipeline -- weight fragments two / three stages ahead in a
// register ring, LDS operands one stage ahead, a scheduling barrier per stage: 131 / 133 us vs 125 us, SQ_WAIT_INST_ANY UP 11 %.)
// One job of the static-shape layer: NTJ 16-position tiles x 16 output channels (row tile m); cc[nt] = the position index of this lane's
// column in tile nt (>= npos: an empty column -- computed on a clamped address, stored into the pad).
// AHEAD: 0 the rolled tap loop with two half-tap weight sets; 1 a whole tap of weight lookahead.  (Measured and removed, round 6: a ring
// of three full-tap sets running on ACROSS the jobs of a wave -- two taps of lookahead, the next job's first two taps requested behind
// this job's last two: 98.9 us against 95.3 for the one-tap form, 128 registers + 124 B of scratch.)
template <int K, int S, int CIN, int COUT, int TIN, bool HAS_RES, int NTJ, bool WLDS, int AHEAD = 0>
__device__ __forceinline__ void fused_job_s(const FusedArgs& a, const FusedLayer& L, const float* __restrict__ xin, const int in_sz,
                                            float* yout, const float* res, const float* w, const int npos, const int m, const int (&cc)[NTJ],
                                            const int r, const int q) {
    constexpr int TOUT = (TIN + S - 1) / S;
    constexpr int PADT = ((TOUT - 1) * S + K - TIN) > 0 ? ((TOUT - 1) * S + K - TIN) : 0;
    constexpr int PADLO = PADT / 2;
    constexpr int TPI = TIN + 2 * kHalo, TPO = TOUT + 2 * kHalo;
    constexpr int C4 = CIN / 4;
    constexpr int WSTEP = 4 * COUT, XSTEP = 4 * TPI;
    const int out_sz = L.out_sz;
    const int res_sz = L.res_sz;
    const float* scale = a.ss + L.ss_off;
    const float* shift = scale + L.c_pad;
    int gg[NTJ], tt[NTJ];
    const float* xp[NTJ];
#pragma unroll
    for (int nt = 0; nt < NTJ; ++nt) {
        const int p = min(cc[nt], npos - 1);
        gg[nt] = p / TOUT;
        tt[nt] = p - gg[nt] * TOUT;
        xp[nt] = xin + gg[nt] * in_sz + q * TPI + tt[nt] * S + kHalo - PADLO;
    }
    const float* wp = w + q * COUT + min(m * 16 + r, COUT - 1);
    float sc[4], sh[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int co = min(m * 16 + q * 4 + reg, COUT - 1);
        sc[reg] = scale[co];
        sh[reg] = shift[co];
    }
    f32x4 acc[NTJ];
#pragma unroll
    for (int nt = 0; nt < NTJ; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (AHEAD == 1) {
        // A whole tap of lookahead for the narrow layers (round 6): in the loop below the compiler sinks a half-tap's refill loads to the
        // END of the tap body (ISA: four global_load_dword behind the last MFMA, `s_waitcnt vmcnt(3)` in front of the next tap's first), so
        // with 8 - 12 MFMAs per tap (16 / 24 input channels) every tap waits out an L1 / L2 round trip.  Two full-tap register sets that
        // trade roles (taps unrolled by two: no copies): the NEXT tap's fragments are requested -- through a buffer descriptor: uniform
        // base + constant lane offset + uniform tap offset, no vector address arithmetic -- before this tap's LDS reads and MFMAs.  Same
        // accumulation order: bitwise.
        const buf_rsrc wr = make_rsrc(w);
        const unsigned wl = (unsigned)(q * COUT + min(m * 16 + r, COUT - 1)) * 4u;
        float w0[C4], w1[C4];
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) w0[c4] = buf_load_f32(wr, wl, (unsigned)(c4 * WSTEP) * 4u);
        auto tap = [&](const int j, const float (&wu)[C4], float (&wf)[C4], const bool fill) {
            TCR_WHATIF_PRIO(32768, 2);
            if (fill && !TCR_WHATIF(4096)) {
#pragma unroll
                for (int c4 = 0; c4 < C4; ++c4) wf[c4] = buf_load_f32(wr, wl, (unsigned)(((j + 1) * C4 + c4) * WSTEP) * 4u);
            }
            float b[NTJ][C4];
            if (TCR_WHATIF(8192) && j > 0) {
#pragma unroll
                for (int c4 = 0; c4 < C4; ++c4)
#pragma unroll
                    for (int nt = 0; nt < NTJ; ++nt) b[nt][c4] = wu[c4] + (float)nt;       // (timing what-if: no LDS operand reads after tap 0)
            } else {
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4)
#pragma unroll
                for (int nt = 0; nt < NTJ; ++nt) b[nt][c4] = xp[nt][c4 * XSTEP + j];
            }
            __builtin_amdgcn_sched_barrier(0);                  // (the requests stay in front of the tap's MFMAs)
            TCR_WHATIF_PRIO(16384, 2);
            TCR_WHATIF_PRIO(32768, 0);
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4)
#pragma unroll
                for (int nt = 0; nt < NTJ; ++nt) acc[nt] = TCR_MFMA(wu[c4], b[nt][c4], acc[nt]);
            TCR_WHATIF_PRIO(16384, 0);
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
    // rolled taps, two half-tap weight sets refilled for the next tap (the round-2 loop)
    constexpr int H0 = C4 / 2, H1 = C4 - H0;
    float wa[H0 > 0 ? H0 : 1], wb[H1];
#pragma unroll
    for (int c4 = 0; c4 < H0; ++c4) wa[c4] = wp[c4 * WSTEP];
#pragma unroll
    for (int c4 = 0; c4 < H1; ++c4) wb[c4] = wp[(H0 + c4) * WSTEP];
    // (throughput kernels: rolled taps -- 128-register budget; small-batch kernel: unrolled, the LDS operand and weight reads of the
    //  following taps move ahead of the MFMAs)
    constexpr int kTapUnroll = WLDS ? K : 1;
#pragma unroll kTapUnroll
    for (int j = 0; j < K; ++j) {
        const int jn = TCR_WHATIF(2) ? 0 : min(j + 1, K - 1);
        const int jx = TCR_WHATIF(4) ? 0 : j;
        {
            float b[NTJ][H0 > 0 ? H0 : 1];
#pragma unroll
            for (int c4 = 0; c4 < H0; ++c4)
#pragma unroll
                for (int nt = 0; nt < NTJ; ++nt) b[nt][c4] = xp[nt][c4 * XSTEP + jx];
#pragma unroll
            for (int c4 = 0; c4 < H0; ++c4)
#pragma unroll
                for (int nt = 0; nt < NTJ; ++nt) acc[nt] = TCR_MFMA(wa[c4], b[nt][c4], acc[nt]);
#pragma unroll
            for (int c4 = 0; c4 < H0; ++c4) wa[c4] = wp[(jn * C4 + c4) * WSTEP];
        }
        {
            float b[NTJ][H1];
#pragma unroll
            for (int c4 = 0; c4 < H1; ++c4)
#pragma unroll
                for (int nt = 0; nt < NTJ; ++nt) b[nt][c4] = xp[nt][(H0 + c4) * XSTEP + jx];
#pragma unroll
            for (int c4 = 0; c4 < H1; ++c4)
#pragma unroll
                for (int nt = 0; nt < NTJ; ++nt) acc[nt] = TCR_MFMA(wb[c4], b[nt][c4], acc[nt]);
#pragma unroll
            for (int c4 = 0; c4 < H1; ++c4) wb[c4] = wp[(jn * C4 + H0 + c4) * WSTEP];
        }
    }
    }
    // ---- epilogue: folded BN (+ shortcut) (+ ReLU) -> LDS rows ----
    const int dump = a.buf_off[2] + a.group * a.buf_sz[2] + (q * 16 + r);
    const float lo = L.relu ? 0.f : -3.4e38f;
    float rv[NTJ][4];
    if (HAS_RES) {
#pragma unroll
        for (int nt = 0; nt < NTJ; ++nt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = min(m * 16 + q * 4 + reg, COUT - 1);
                rv[nt][reg] = res[gg[nt] * res_sz + co * TPO + kHalo + tt[nt]];
            }
    }
#pragma unroll
    for (int nt = 0; nt < NTJ; ++nt) {
        const bool pv = cc[nt] < npos;
        const f32x4 ac = acc[nt];
        const int base = gg[nt] * out_sz + kHalo + tt[nt];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int co = m * 16 + q * 4 + reg;
            float v = fmaf(ac[reg], sc[reg], sh[reg]);
            if (HAS_RES) v = fmaxf(v + rv[nt][reg], 0.f);       // tc_resnet.py:40-41
            else v = fmaxf(v, lo);
            const bool ok = pv && (COUT % 16 == 0 || co < COUT) && !(TCR_WHATIF(32) && v != 12345.f);
            yout[ok ? base + co * TPO : dump - a.buf_off[L.out_buf]] = v;
        }
    }
}

template <int NW, int K, int S, int CIN, int COUT, int TIN, bool HAS_RES, int NTJ = 2, bool WLDS = false>
__device__ __forceinline__ void fused_layer_s(const FusedArgs& a, const FusedLayer& L, const float* __restrict__ xin, const int in_sz,
                                              float* lds, const int ng, const int wave, const int r_in, const int q_in,
                                              const int w_lds = 0) {     // WLDS: this layer's weights are staged in LDS at float offset w_lds (small-batch kernel)
    // lane geometry re-derived from an opaque zero: the per-lane address arithmetic of ten layers must not be hoisted out of the group
    // loop, where it would stay live through every other layer (the kernel runs at a 128-register budget)
    const int oz = opaque_zero();
    const int r = r_in + oz, q = q_in + oz;
    constexpr int TOUT = (TIN + S - 1) / S;
    constexpr int NRT = (COUT + 15) / 16;
    constexpr int JP = 16 * NTJ;            // positions per job
    static_assert(CIN % 4 == 0, "channel quads");
    static_assert(NTJ == 1 || NTJ == 2 || NTJ == 4, "one, two or four 16-position tiles per job");
    float* yout = lds + a.buf_off[L.out_buf];
    const float* res = L.res_buf >= 0 ? lds + a.buf_off[L.res_buf] : nullptr;
    const int npos = ng * TOUT;
    const int ncp = (npos + JP - 1) / JP;
    const float* w = WLDS ? lds + w_lds : a.params + L.w_off;
    // Which of a job's positions a lane's tile columns hold (the columns of the implicit GEMM are independent: any assignment
    // gives the same sums).  One ds_read_b32 serves lanes (q, q + 1) x 16 columns against 32 banks: with stride 2 consecutive columns
    // are 2 floats apart and the odd row pitch puts row q + 1 on the other bank parity; with stride 1 the columns of a tile are every
    // OTHER position (tile 0 the even, tile 1 the odd ones of a run of 32) for the same picture.  The host pads the per-utterance stride so that the
    // pattern continues across the utterances of a group (net.cpp: fused_strides).  PMC, TCResNet8 at 49 frames: see OPTLOG.md.
    // NTJ = 4 (round 5 experiment; the 16 / 24-channel layers at 25 frames): a weight fragment feeds four MFMAs instead of two -- those layers have
    // 4 - 6 channel quads, i.e. 8 - 12 MFMAs between two dependent weight loads of a tap, and 14 jobs for 8 waves; with 64-position jobs a tap's
    // loads sit behind 16 - 24 MFMAs and a layer is one round of <= 8 jobs.  Same sums (columns are independent), bitwise -- and no faster
    // (phase stamps, scripts/fused_ts.py: block 0's first phase 37.5 -> 31.8 k cycles, its second 39.5 -> 43.4 k: the two workgroups of a CU
    // share the matrix pipes, a shorter phase of one lengthens the other's).
    constexpr bool IL = S == 1;
    for (int job = wave; job < ncp * NRT; job += NW) {
        const int cp = job / NRT, m = job - cp * NRT;
        int cc[NTJ];
#pragma unroll
        for (int nt = 0; nt < NTJ; ++nt) cc[nt] = (IL && NTJ > 1) ? cp * JP + 32 * (nt >> 1) + 2 * r + (nt & 1) : cp * JP + 16 * nt + r;
        fused_job_s<K, S, CIN, COUT, TIN, HAS_RES, NTJ, WLDS>(a, L, xin, in_sz, yout, res, w, npos, m, cc, r, q);
    }
}

// The same layer with its work dealt in UNITS of one 16-position tile x 16 output channels (round 6).  fused_layer_s deals whole jobs of
// two tiles round-robin: 14 jobs on 8 waves in block 0 (a SIMD hosts waves w and w + 4 of the workgroup: 4 / 4 / 3 / 3 jobs), 6 jobs in
// block 2 (2 / 2 / 1 / 1: two SIMDs carry twice the matrix work of the other two), and the positions of a group are padded to 32.  Here a
// wave owns a contiguous run of units, as even as units allow -- the first U mod 8 waves one more, so the SIMDs differ by at most one unit
// (block 0: 7 / 7 / 6 / 6 of 26 units instead of 8 / 8 / 6 / 6 of 28; block 1: 14 units instead of 16; block 2: 3 / 3 / 3 / 3 instead of
// 4 / 4 / 2 / 2) --, and walks it two units at a time where both lie in the same row tile (one weight fragment feeds both), one otherwise.
// Columns of the implicit GEMM are independent: bitwise the job form.  Stride-1 layers keep "a tile = every other position of a run of
// 32" (the bank picture above); a last run of <= 16 positions is one tile of consecutive positions.
template <int NW, int K, int S, int CIN, int COUT, int TIN, bool HAS_RES, int AHEAD = 0>
__device__ __forceinline__ void fused_layer_u(const FusedArgs& a, const FusedLayer& L, const float* __restrict__ xin, const int in_sz,
                                              float* lds, const int ng, const int wave, const int r_in, const int q_in) {
    const int oz = opaque_zero();
    const int r = r_in + oz, q = q_in + oz;
    constexpr int TOUT = (TIN + S - 1) / S;
    constexpr int NRT = (COUT + 15) / 16;
    constexpr bool IL = S == 1;
    float* yout = lds + a.buf_off[L.out_buf];
    const float* res = L.res_buf >= 0 ? lds + a.buf_off[L.res_buf] : nullptr;
    const int npos = ng * TOUT;
    const int full = npos >> 5, rem = npos & 31;
    const int nt16 = IL ? 2 * full + (rem == 0 ? 0 : (rem <= 16 ? 1 : 2)) : (npos + 15) >> 4;
    const int il_end = (IL && rem > 0 && rem <= 16) ? 2 * full : nt16;          // tiles from il_end on: consecutive positions
    const int U = NRT * nt16;
    const int per = U / NW, extra = U - per * NW;
    // (measured and removed: the two workgroups of a CU handing their spare units to different SIMDs -- waves rotated by two in every
    //  other workgroup, by blockIdx bit 8 or bit 3: 96.5 / 95.9 us against 95.9; SIMD mates -- waves w and w + 4 -- walking an odd run single
    //  unit first so that their prologues / epilogues do not coincide, also flipped per 256 workgroups: 97.4 / 96.8 against 96.3 - 97.6)
    int u = wave * per + min(wave, extra);
    const int end = u + per + (wave < extra ? 1 : 0);
    const float* w = a.params + L.w_off;
    auto col = [&](int c) { return (IL && c < il_end) ? (c >> 1) * 32 + 2 * r + (c & 1) : (IL ? full * 32 + r : c * 16 + r); };
    while (u < end) {
        const int m = u / nt16, c = u - m * nt16;
        if (c + 1 < nt16 && u + 1 < end) {
            const int cc[2] = {col(c), col(c + 1)};
            fused_job_s<K, S, CIN, COUT, TIN, HAS_RES, 2, false, AHEAD>(a, L, xin, in_sz, yout, res, w, npos, m, cc, r, q);
            u += 2;
        } else {
            const int cc[1] = {col(c)};
            fused_job_s<K, S, CIN, COUT, TIN, HAS_RES, 1, false, AHEAD>(a, L, xin, in_sz, yout, res, w, npos, m, cc, r, q);
            u += 1;
        }
    }
}

// First conv of the static-shape kernel (3 x 1, 40 -> 16, stride 1), activations straight from global memory.  In the generic form
// every K-step's two operand loads were issued ONE MFMA ahead of their use (ISA: `s_waitcnt vmcnt(1)` in front of each MFMA): 30
// exposed L1 / L2 / HBM latencies per job -- the layer took 25 us of the kernel's 125 (timing what-if, scripts/whatif_net.py) for
// 10 % of its MFMAs.  Here a job requests ALL its operands up front (per lane 10 channel quads x 3 taps = three consecutive
// floats each, and the 30 weight fragments; a job is 16 positions so that this fits the register budget), then runs its 30 MFMAs:
// one exposed latency per job.  Same accumulation order.  (Measured and removed, round 6: the next job's activations requested a whole job
// ahead -- 90 operand registers, 128 + 200 B of scratch: 103.7 us against 95.4 -- and half a job ahead in two sets of 15 -- less scratch
// than this form, 96.4 against 95.3: the phase reads the whole feature tensor, 37 MB at batch 4096, with every wave of the chip asking at
// the same moment; it is not a per-job latency.)
template <int NW, int T0, bool WLDS = false>
__device__ __forceinline__ void fused_conv0_s(const FusedArgs& a, const FusedLayer& L, const float* __restrict__ xin, const int in_sz,
                                              float* lds, const int ng, const int wave, const int r_in, const int q_in,
                                              const int w_lds = 0) {
    const int oz = opaque_zero();
    const int r = r_in + oz, q = q_in + oz;
    constexpr int K = 3, CIN = 40, COUT = 16, C4 = CIN / 4;
    constexpr int TPI = T0 + 2 * kHalo, TPO = TPI, TOUT = T0, PADLO = 1;
    constexpr int WSTEP = 4 * COUT, XSTEP = 4 * TPI;
    float* yout = lds + a.buf_off[L.out_buf];
    const int out_sz = L.out_sz;
    const int npos = ng * TOUT;
    const int nct = (npos + 15) / 16;
    const float* w = WLDS ? lds + w_lds : a.params + L.w_off;
    const float* scale = a.ss + L.ss_off;
    const float* shift = scale + L.c_pad;
    for (int job = wave; job < nct; job += NW) {            // job = 16 positions x 16 channels (30 + 30 operand registers)
        const int p0 = min(job * 16 + r, npos - 1);
        const int g0 = p0 / TOUT;
        const int t0 = p0 - g0 * TOUT;
        const float* wp = w + q * COUT + r;
        const float* x0 = xin + g0 * in_sz + q * TPI + t0 + kHalo - PADLO;
        float b0[C4][K], wf[K][C4];
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4)
#pragma unroll
            for (int j = 0; j < K; ++j) b0[c4][j] = x0[c4 * XSTEP + j];
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4) wf[j][c4] = wp[(j * C4 + c4) * WSTEP];
        float sc[4], sh[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            sc[reg] = scale[q * 4 + reg];
            sh[reg] = shift[q * 4 + reg];
        }
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
        TCR_WHATIF_PRIO(16384, 2);
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4) acc0 = TCR_MFMA(wf[j][c4], b0[c4][j], acc0);
        TCR_WHATIF_PRIO(16384, 0);
        const int dump = a.buf_off[2] + a.group * a.buf_sz[2] + (q * 16 + r) - a.buf_off[L.out_buf];
        const float lo = L.relu ? 0.f : -3.4e38f;
        const bool pv = job * 16 + r < npos;
        const int base = g0 * out_sz + kHalo + t0;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float v = fmaxf(fmaf(acc0[reg], sc[reg], sh[reg]), lo);
            yout[pv ? base + (q * 4 + reg) * TPO : dump] = v;
        }
    }
}

// Head of the static-shape kernel: avg-pool by all threads, then ONE wave runs fc / fc2 on the matrix cores (D[output][utterance], the
// arrangement of head.hip: bitwise the fmaf chain of `fused_head`), the softmax in registers with the class sum carried through the
// three 16-lane rows in class order (bitwise the sequential sum of `fused_head`), and the stores.  `fused_head` -- runtime shapes,
// 48 dependent global weight loads per dot product in batches of 8, twelve expf per thread -- cost 20 us of the kernel's 125.
template <int NT, int FC, int FT, int NCLS>
__device__ __forceinline__ void fused_head_s(const FusedArgs& a, float* lds, const int n0, const int ng, const int tid) {
    static_assert(FC % 4 == 0 && NCLS + 2 <= 16, "one 16-row MFMA tile holds the logits and the two range outputs");
    constexpr int TP = FT + 2 * kHalo;
    const float* fb = lds + a.buf_off[a.feat_buf];
    const int fsz = a.feat_sz;
    float* pooled = lds + a.buf_off[(a.feat_buf + 1) % 3];
    const int lane = tid & 63, r = lane & 15, q = lane >> 4;
    // wave 0: the weight fragments are requested before the pooling phase (their latency hides behind it)
    float af[FC / 4];
    if (tid < 64) {
        const int zero = opaque_zero();        // (keeps these loads inside the group loop: hoisted, they would live through every layer)
#pragma unroll
        for (int s = 0; s < FC / 4; ++s) {
            const int c = 4 * s + q;
            const float* src = r < NCLS ? a.params + a.fc_off + c * NCLS + r : a.params + a.fc2_off + c * 2 + min(r - NCLS, 1);
            const float v = src[zero];
            af[s] = r < NCLS + 2 ? v : 0.f;
        }
    }
    for (int i = tid; i < ng * FC; i += NT) {
        const int g = i / FC, c = i - g * FC;
        const float* row = fb + g * fsz + c * TP + kHalo;
        float v[FT];
#pragma unroll
        for (int t = 0; t < FT; ++t) v[t] = row[t];
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < FT; ++t) s += v[t];
        pooled[i] = s / (float)FT;
    }
    __syncthreads();
    if (tid < 64) {
        const int g = min(r, ng - 1);
        float bf[FC / 4];
#pragma unroll
        for (int s = 0; s < FC / 4; ++s) bf[s] = pooled[g * FC + 4 * s + q];
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < FC / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], acc, 0, 0, 0);
        // this lane: utterance r, outputs o = 4 q + reg (classes 0 .. NCLS-1, then the two range outputs)
        float mx = -3.0e38f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) if (4 * q + reg < NCLS) mx = fmaxf(mx, acc[reg]);       // (NCLS % 4 == 0 or not: per-lane test)
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float e[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) e[reg] = expf(acc[reg] - mx);
        // se = ((e_0 + e_1) + e_2) + ... in class order: row q adds its classes to the sum of rows < q
        float se = 0.f;
#pragma unroll
        for (int row = 0; row < (NCLS + 3) / 4; ++row) {
            const float prev = __shfl(se, (row > 0 ? (row - 1) * 16 : 0) + r);
            float cur = row > 0 ? prev : 0.f;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) if (4 * row + reg < NCLS) cur += e[reg];
            if (q == row) se = cur;
        }
        se = __shfl(se, ((NCLS + 3) / 4 - 1) * 16 + r);
        if (r < ng) {
            const size_t n = (size_t)(n0 + r);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int o = 4 * q + reg;
                if (o < NCLS) {
                    a.logits[n * NCLS + o] = acc[reg];
                    a.probs[n * NCLS + o] = e[reg] / se;
                } else if (o < NCLS + 2 && a.ranges) {
                    a.ranges[n * 2 + (o - NCLS)] = 1.0f / (1.0f + expf(-acc[reg]));
                }
            }
        }
    }
    __syncthreads();
}

// Zero halos (4 + 4 floats) of the rows a static-shape layer is about to write: [ng][COUT] rows of TOUT + 8 floats; one thread per
// (row, side), four stores each (one thread per element: three times the index arithmetic, +6 us per launch).
template <int NT, int COUT, int TOUT>
__device__ __forceinline__ void fused_zero_halo(float* yout, const int out_sz, const int ng, const int tid_in) {
    constexpr int TPO = TOUT + 2 * kHalo;
    if (TCR_WHATIF(1024)) return;
    const int tid = tid_in + opaque_zero();
    for (int i = tid; i < ng * COUT * 2; i += NT) {
        const int row = i >> 1;
        const int g = row / COUT, co = row - g * COUT;
        float* p = yout + g * out_sz + co * TPO + ((i & 1) ? kHalo + TOUT : 0);
        p[0] = 0.f; p[1] = 0.f; p[2] = 0.f; p[3] = 0.f;
    }
}

// ---- head: global average pool -> fc / fc2 -> softmax / sigmoid (tc_resnet.py:43-52), shared by the fused kernels ----
template <int NT>
__device__ __forceinline__ void fused_head(const FusedArgs& a, float* lds, const int n0, const int ng, const int tid) {
    {
        const float* fb = lds + a.buf_off[a.feat_buf];
        const int fsz = a.feat_sz, tp = a.feat_t + 2 * kHalo;
        float* pooled = lds + a.buf_off[(a.feat_buf + 1) % 3];         // any buffer other than the feature buffer
        for (int i = tid; i < ng * a.feat_c; i += NT) {
            const int g = i / a.feat_c, c = i - g * a.feat_c;
            const float* row = fb + g * fsz + c * tp + kHalo;
            float s = 0.f;
            for (int t = 0; t < a.feat_t; ++t) s += row[t];
            pooled[i] = s / (float)a.feat_t;
        }
        __syncthreads();
        float* lg = pooled + a.group * a.feat_c;                        // [ng][nc + 2]
        const int no = a.nc + 2;
        for (int i = tid; i < ng * no; i += NT) {
            const int g = i / no, o = i - g * no;
            const float* pv = pooled + g * a.feat_c;
            float s = 0.f;
            if (o < a.nc) {
                const float* wf = a.params + a.fc_off + o;
#pragma unroll 8
                for (int c = 0; c < a.feat_c; ++c) s = fmaf(pv[c], wf[(size_t)c * a.nc], s);
            } else {
                const float* wf = a.params + a.fc2_off + (o - a.nc);
#pragma unroll 8
                for (int c = 0; c < a.feat_c; ++c) s = fmaf(pv[c], wf[(size_t)c * 2], s);
            }
            lg[i] = s;
        }
        __syncthreads();
        // softmax / sigmoid: one thread per (utterance, output); the max and the sum are recomputed per thread
        // from the LDS row in the SAME order as the single-thread form (bitwise identical results)
        for (int i = tid; i < ng * no; i += NT) {
            const int g = i / no, o = i - g * no;
            const float* z = lg + g * no;
            const size_t n = (size_t)(n0 + g);
            if (o < a.nc) {
                float mx = z[0];
                for (int k = 1; k < a.nc; ++k) mx = fmaxf(mx, z[k]);
                float se = 0.f;
                for (int k = 0; k < a.nc; ++k) se += expf(z[k] - mx);
                a.logits[n * a.nc + o] = z[o];
                a.probs[n * a.nc + o] = expf(z[o] - mx) / se;
            } else if (a.ranges) {
                a.ranges[n * 2 + (o - a.nc)] = 1.0f / (1.0f + expf(-z[o]));
            }
        }
        __syncthreads();
    }
}

// TCResNet8-1.0 on 40 coefficients with every layer shape fixed at compile time (T0 = 49 or 98 frames): the flagship
// configurations of BASELINE.json.  Same walk, same LDS plan (FusedArgs), bitwise the generic kernel's results.
#define TCR_WAVES_PER_SIMD_4 TCR_WAVES_PER_SIMD(4)     // two 8-wave workgroups per CU: <= 128 VGPRs
// WD < 0: the round-2 layer (A/B arm, TCR_TUNE_NET_FUSED = 4).  HALO: some consumer convolves this layer's rows (K > 1) and so reads
// their zero halo; the shortcut convs' outputs (only ever a residual term) and the last block output (only pooled) skip the zero pass.
template <int NW, int K, int S, int CIN, int COUT, int TIN, int WD, bool HAS_RES, bool HALO = true, int NTJ = 2, bool WLDS = false>
__device__ __forceinline__ void fused_layer_sel(const FusedArgs& a, const FusedLayer& L, const float* __restrict__ xin, const int in_sz,
                                                float* lds, const int ng, const int wave, const int r, const int q,
                                                const int w_lds = 0) {
    if constexpr (WD < 0) fused_layer_t<NW, K, S, CIN, COUT, TIN>(a, L, xin, in_sz, lds, ng, wave, r, q);
    else {
        if constexpr (HALO) fused_zero_halo<NW * 64, COUT, (TIN + S - 1) / S>(lds + a.buf_off[L.out_buf], L.out_sz, ng, (int)threadIdx.x);
        if constexpr (NTJ == 0) fused_layer_u<NW, K, S, CIN, COUT, TIN, HAS_RES>(a, L, xin, in_sz, lds, ng, wave, r, q);      // (units: see fused_layer_u)
        else if constexpr (NTJ == 3) fused_layer_u<NW, K, S, CIN, COUT, TIN, HAS_RES, (CIN <= 32 ? 1 : 0)>(a, L, xin, in_sz, lds, ng, wave, r, q);   // (+ a whole tap of weight lookahead in the layers of <= 32 input channels)
        else if constexpr (NTJ == 13) fused_layer_u<NW, K, S, CIN, COUT, TIN, HAS_RES, (CIN <= 48 ? 1 : 0)>(a, L, xin, in_sz, lds, ng, wave, r, q);   // (TCResNet14-1.5's kernel: <= 48)
        else fused_layer_s<NW, K, S, CIN, COUT, TIN, HAS_RES, NTJ, WLDS>(a, L, xin, in_sz, lds, ng, wave, r, q, w_lds);
    }
}

template <int NW, int T0, int WD, int NTJ0 = 2>
__global__ __launch_bounds__(NW * 64) TCR_WAVES_PER_SIMD_4 void net_fused_tc8_kernel(const FusedArgs a) {
    constexpr int NT = NW * 64;
    constexpr int T1 = (T0 + 1) / 2, T2 = (T1 + 1) / 2;
    float* lds = reinterpret_cast<float*>(dyn_lds());
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int row = a.in_c * a.in_tp;
#if TCR_FUSED_WHATIF & 2048
    // phase timing (diagnostic side build, WRONG outputs): cycles between the barriers, written over the group's first probabilities
    long long ts[14];
    int nts = 0;
#define TCR_TC8_BARRIER do { __syncthreads(); ts[nts++] = (long long)__builtin_readcyclecounter(); } while (0)
#elif TCR_FUSED_WHATIF & 8
#define TCR_TC8_BARRIER ((void)0)
#else
#define TCR_TC8_BARRIER __syncthreads()
#endif
    // (Measured and removed, round 6: the upper half of the waves running a block's first conv BEFORE its shortcut conv -- both read the same rows --
    //  so that a SIMD's two waves are not in the shortcut's load / store-bound jobs together: 96.5 vs 95.6 us at 49 frames, 173.9 vs 176.1 at 98.)
    // (tiles per job: two; NTJ0 = 4: four for block 0's layers -- 16 / 24 channels at T0 / 2 frames --, the TCR_TUNE_NET_FUSED = 5 arm)
    // (NTJ0 = 0: the nine-tap layers' work dealt in 16-position units, fused_layer_u)
#define TCR_TC8(LI, K_, S_, CI_, CO_, T_) fused_layer_sel<NW, K_, S_, CI_, CO_, T_, WD, (LI == 3 || LI == 6 || LI == 9), (K_ != 1 && LI != 9), ((LI >= 1 && LI <= 3 && NTJ0 == 4) ? 4 : (((NTJ0 == 0 || NTJ0 == 3) && K_ == 9) ? NTJ0 : 2))>(a, a.layer[LI], lds + a.buf_off[a.layer[LI].in_buf], a.layer[LI].in_sz, lds, ng, wave, r, q)
    for (int grp = blockIdx.x; grp < (TCR_WHATIF(512) ? 0 : a.n_groups); grp += gridDim.x) {
        const int n0 = grp * a.group;
        const int ng = min(a.group, a.batch - n0);
#if TCR_FUSED_WHATIF & 2048
        nts = 0;
        ts[nts++] = (long long)__builtin_readcyclecounter();
#endif
        if constexpr (WD < 0) fused_layer_sel<NW, 3, 1, 40, 16, T0, WD, false>(a, a.layer[0], a.feat + (size_t)n0 * row, row, lds, ng, wave, r, q);
        else if (!TCR_WHATIF(64)) {
            fused_zero_halo<NT, 16, T0>(lds + a.buf_off[a.layer[0].out_buf], a.layer[0].out_sz, ng, tid);
            fused_conv0_s<NW, T0>(a, a.layer[0], a.feat + (size_t)n0 * row, row, lds, ng, wave, r, q);
        }
        TCR_TC8_BARRIER;
        TCR_TC8(1, 1, 2, 16, 24, T0);           // block0/down (reads the same rows as conv0_0: no barrier in between)
        TCR_TC8(2, 9, 2, 16, 24, T0);
        TCR_TC8_BARRIER;
        TCR_TC8(3, 9, 1, 24, 24, T1);
        TCR_TC8_BARRIER;
        TCR_TC8(4, 1, 2, 24, 32, T1);
        TCR_TC8(5, 9, 2, 24, 32, T1);
        TCR_TC8_BARRIER;
        TCR_TC8(6, 9, 1, 32, 32, T2);
        TCR_TC8_BARRIER;
        TCR_TC8(7, 1, 2, 32, 48, T2);
        TCR_TC8(8, 9, 2, 32, 48, T2);
        TCR_TC8_BARRIER;
        TCR_TC8(9, 9, 1, 48, 48, (T2 + 1) / 2);
        TCR_TC8_BARRIER;
        if constexpr (WD < 0) fused_head<NT>(a, lds, n0, ng, tid);
        else if (!TCR_WHATIF(16)) {
            if (a.nc == 12) fused_head_s<NT, 48, (T2 + 1) / 2, 12>(a, lds, n0, ng, tid);
            else fused_head<NT>(a, lds, n0, ng, tid);
        }
#if TCR_FUSED_WHATIF & 2048
        __syncthreads();
        ts[nts++] = (long long)__builtin_readcyclecounter();
        if (tid == 0)
            for (int i = 0; i + 1 < nts && i < 12; ++i) a.probs[(size_t)n0 * a.nc + i] = (float)(ts[i + 1] - ts[i]);
#endif
    }
#undef TCR_TC8
#undef TCR_TC8_BARRIER
}

// Small-batch (latency) form of the same network, TCResNet8-1.0 at 49 frames: ONE utterance per 8-wave workgroup, every phase's weights
// copied into LDS by the DMA path (global_load_lds_dwordx4, coalesced, all in flight at once) one phase ahead, A fragments read from
// LDS.  In the throughput kernel a job's nine taps are nine DEPENDENT weight loads from L2 (cold L1 at batch 1): ~4.5 us per layer,
// 35 us for the walk; here a phase costs one L2 round trip, hidden behind the phase before it (two weight buffers: 83 KB + 61 KB, the
// two largest consecutive phases).  One 16-position tile per job: more jobs for the eight waves, half the matrix work of a
// two-tile job whose second tile would be empty at 13 / 7 frames.  Same accumulation order as the throughput kernels: bitwise their
// results, so an utterance's outputs do not depend on the batch it arrives in.  (Round 4's `net_small_kernel` -- every weight fragment of
// a job gathered into registers -- was slower than the throughput kernel: 108 dword gathers per lane and job.)
template <int NW>
__device__ __forceinline__ void small_stage_weights(const float* __restrict__ src, int n_floats, float* dst, int wave, int lane) {
    for (int c0 = wave * 256; c0 < n_floats; c0 += NW * 256) {          // one DMA instruction: 64 lanes x 16 B = 256 floats
        const int c = c0 + 4 * lane;
        if (c < n_floats) glds16(src + c, dst + c0);
    }
}

template <int NW, int T0>
__global__ __launch_bounds__(NW * 64) void net_small_tc8_kernel(const FusedArgs a, const int wa_off, const int wb_off) {
    constexpr int NT = NW * 64;
    constexpr int T1 = (T0 + 1) / 2, T2 = (T1 + 1) / 2, T3 = (T2 + 1) / 2;
    float* lds = reinterpret_cast<float*>(dyn_lds());
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int row = a.in_c * a.in_tp;
    float* WA = lds + wa_off;
    float* WB = lds + wb_off;
    constexpr int W0 = 3 * 40 * 16, WD0 = 16 * 24, W00 = 9 * 16 * 24, W01 = 9 * 24 * 24, WD1 = 24 * 32, W10 = 9 * 24 * 32, W11 = 9 * 32 * 32,
                  WD2 = 32 * 48, W20 = 9 * 32 * 48, W21 = 9 * 48 * 48;
#define TCR_SM_W(LI) (a.params + a.layer[LI].w_off)
#define TCR_SM(LI, K_, S_, CI_, CO_, T_, WP_) fused_layer_sel<NW, K_, S_, CI_, CO_, T_, 0, (LI == 3 || LI == 6 || LI == 9), (K_ != 1 && LI != 9), 1, true>(a, a.layer[LI], lds + a.buf_off[a.layer[LI].in_buf], a.layer[LI].in_sz, lds, 1, wave, r, q, WP_)
#if TCR_FUSED_WHATIF & 2048
    long long ts[16];
    int nts = 0;
#define TCR_SM_TS() ts[nts++] = (long long)__builtin_readcyclecounter()
#else
#define TCR_SM_TS() ((void)0)
#endif
    for (int n0 = blockIdx.x; n0 < a.batch; n0 += gridDim.x) {
        TCR_SM_TS();
        small_stage_weights<NW>(TCR_SM_W(0), W0, WA, wave, lane);               // phase 0 -> A
        small_stage_weights<NW>(TCR_SM_W(1), WD0, WB, wave, lane);              // phase 1 -> B
        small_stage_weights<NW>(TCR_SM_W(2), W00, WB + WD0, wave, lane);
        fused_zero_halo<NT, 16, T0>(lds + a.buf_off[a.layer[0].out_buf], a.layer[0].out_sz, 1, tid);
        wait_dma();
        __syncthreads();
        TCR_SM_TS();
        fused_conv0_s<NW, T0, true>(a, a.layer[0], a.feat + (size_t)n0 * row, row, lds, 1, wave, r, q, wa_off);
        __syncthreads();
        TCR_SM_TS();
        small_stage_weights<NW>(TCR_SM_W(3), W01, WA, wave, lane);              // phase 2 -> A (conv0 is done with it)
        TCR_SM(1, 1, 2, 16, 24, T0, wb_off);
        TCR_SM(2, 9, 2, 16, 24, T0, wb_off + WD0);
        wait_dma();
        __syncthreads();
        TCR_SM_TS();
        small_stage_weights<NW>(TCR_SM_W(4), WD1, WB, wave, lane);              // phase 3 -> B
        small_stage_weights<NW>(TCR_SM_W(5), W10, WB + WD1, wave, lane);
        TCR_SM(3, 9, 1, 24, 24, T1, wa_off);
        wait_dma();
        __syncthreads();
        TCR_SM_TS();
        small_stage_weights<NW>(TCR_SM_W(6), W11, WA, wave, lane);              // phase 4 -> A
        TCR_SM(4, 1, 2, 24, 32, T1, wb_off);
        TCR_SM(5, 9, 2, 24, 32, T1, wb_off + WD1);
        wait_dma();
        __syncthreads();
        TCR_SM_TS();
        small_stage_weights<NW>(TCR_SM_W(7), WD2, WB, wave, lane);              // phase 5 -> B
        small_stage_weights<NW>(TCR_SM_W(8), W20, WB + WD2, wave, lane);
        TCR_SM(6, 9, 1, 32, 32, T2, wa_off);
        wait_dma();
        __syncthreads();
        TCR_SM_TS();
        small_stage_weights<NW>(TCR_SM_W(9), W21, WA, wave, lane);              // phase 6 -> A
        TCR_SM(7, 1, 2, 32, 48, T2, wb_off);
        TCR_SM(8, 9, 2, 32, 48, T2, wb_off + WD2);
        wait_dma();
        __syncthreads();
        TCR_SM_TS();
        TCR_SM(9, 9, 1, 48, 48, T3, wa_off);
        __syncthreads();
        TCR_SM_TS();
        if (a.nc == 12) fused_head_s<NT, 48, T3, 12>(a, lds, n0, 1, tid);
        else fused_head<NT>(a, lds, n0, 1, tid);
#if TCR_FUSED_WHATIF & 2048
        __syncthreads();
        TCR_SM_TS();
        if (tid == 0)
            for (int i = 0; i + 1 < nts && i < 12; ++i) a.probs[(size_t)n0 * a.nc + i] = (float)(ts[i + 1] - ts[i]);
        nts = 0;
#endif
    }
#undef TCR_SM
#undef TCR_SM_W
#undef TCR_SM_TS
}

// TCResNet14-1.5 (channels 24 / 36 / 36 / 48 / 48 / 72 / 72; BASELINE.json configs[3]'s network) with compile-time layer shapes, T0 = 49 or 98
// frames: the same static layer (`fused_layer_s`), first conv (`fused_conv0_g`: 24 output channels = two row tiles) and head as the
// TCResNet8 instance.  Blocks 1 / 3 / 5 have identity shortcuts: their second conv adds the block input, which lives in its own output
// buffer (each lane reads the residual of exactly the elements it then writes).
template <int NW, int T0, int NTJ0 = 13>      // NTJ0: the nine-tap layers as in net_fused_tc8_kernel (13: units + weight lookahead up to 48 input channels, 0: units, 2: jobs of two tiles)
__global__ __launch_bounds__(NW * 64) TCR_WAVES_PER_SIMD_4 void net_fused_tc14w_kernel(const FusedArgs a) {
    constexpr int NT = NW * 64;
    constexpr int T1 = (T0 + 1) / 2, T2 = (T1 + 1) / 2, T3 = (T2 + 1) / 2;
    float* lds = reinterpret_cast<float*>(dyn_lds());
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int row = a.in_c * a.in_tp;
#define TCR_L(LI, K_, S_, CI_, CO_, T_, RES_, HALO_) fused_layer_sel<NW, K_, S_, CI_, CO_, T_, 0, RES_, HALO_, (K_ == 9 ? NTJ0 : 2)>(a, a.layer[LI], lds + a.buf_off[a.layer[LI].in_buf], a.layer[LI].in_sz, lds, ng, wave, r, q)
    for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
        const int n0 = grp * a.group;
        const int ng = min(a.group, a.batch - n0);
        fused_zero_halo<NT, 24, T0>(lds + a.buf_off[a.layer[0].out_buf], a.layer[0].out_sz, ng, tid);
        fused_conv0_g<NW>(a, a.layer[0], a.feat + (size_t)n0 * row, row, lds, ng, wave, r, q);
        __syncthreads();
        TCR_L(1, 1, 2, 24, 36, T0, false, false);       // block0: shortcut conv + first conv (same input rows: one phase)
        TCR_L(2, 9, 2, 24, 36, T0, false, true);
        __syncthreads();
        TCR_L(3, 9, 1, 36, 36, T1, true, true);
        __syncthreads();
        TCR_L(4, 9, 1, 36, 36, T1, false, true);        // block1 (identity shortcut)
        __syncthreads();
        TCR_L(5, 9, 1, 36, 36, T1, true, true);
        __syncthreads();
        TCR_L(6, 1, 2, 36, 48, T1, false, false);       // block2
        TCR_L(7, 9, 2, 36, 48, T1, false, true);
        __syncthreads();
        TCR_L(8, 9, 1, 48, 48, T2, true, true);
        __syncthreads();
        TCR_L(9, 9, 1, 48, 48, T2, false, true);        // block3 (identity)
        __syncthreads();
        TCR_L(10, 9, 1, 48, 48, T2, true, true);
        __syncthreads();
        TCR_L(11, 1, 2, 48, 72, T2, false, false);      // block4
        TCR_L(12, 9, 2, 48, 72, T2, false, true);
        __syncthreads();
        TCR_L(13, 9, 1, 72, 72, T3, true, true);
        __syncthreads();
        TCR_L(14, 9, 1, 72, 72, T3, false, true);       // block5 (identity)
        __syncthreads();
        TCR_L(15, 9, 1, 72, 72, T3, true, false);
        __syncthreads();
        if (a.nc == 12) fused_head_s<NT, 72, T3, 12>(a, lds, n0, ng, tid);
        else fused_head<NT>(a, lds, n0, ng, tid);
    }
#undef TCR_L
}

// 49 / 98 when the plan is exactly TCResNet14-1.5 on 40 coefficients read from global memory, else 0
static int fused_tc14w_frames(const FusedArgs& a) {
    static const int shape[16][4] = {{3, 1, 40, 24}, {1, 2, 24, 36}, {9, 2, 24, 36}, {9, 1, 36, 36}, {9, 1, 36, 36}, {9, 1, 36, 36}, {1, 2, 36, 48}, {9, 2, 36, 48},
                                     {9, 1, 48, 48}, {9, 1, 48, 48}, {9, 1, 48, 48}, {1, 2, 48, 72}, {9, 2, 48, 72}, {9, 1, 72, 72}, {9, 1, 72, 72}, {9, 1, 72, 72}};
    if (a.n_layers != 16 || !a.in_global || a.in_c != 40) return 0;
    const int t0 = a.layer[0].tin;
    if (t0 != 49 && t0 != 98) return 0;
    int t = t0;
    for (int i = 0; i < 16; ++i) {
        const FusedLayer& L = a.layer[i];
        if (L.k != shape[i][0] || L.stride != shape[i][1] || L.cin != shape[i][2] || L.cout != shape[i][3] || L.tin != t) return 0;
        if (i == 2 || i == 7 || i == 12) t = (t + 1) / 2;
    }
    return t0;
}

// 49 / 98 when the plan is exactly TCResNet8-1.0 on 40 coefficients read from global memory, else 0
static int fused_tc8_frames(const FusedArgs& a) {
    static const int shape[10][4] = {{3, 1, 40, 16}, {1, 2, 16, 24}, {9, 2, 16, 24}, {9, 1, 24, 24}, {1, 2, 24, 32},
                                     {9, 2, 24, 32}, {9, 1, 32, 32}, {1, 2, 32, 48}, {9, 2, 32, 48}, {9, 1, 48, 48}};
    if (a.n_layers != 10 || !a.in_global || a.in_c != 40) return 0;
    const int t0 = a.layer[0].tin;
    if (t0 != 49 && t0 != 98) return 0;
    int t = t0;
    for (int i = 0; i < 10; ++i) {
        const FusedLayer& L = a.layer[i];
        if (L.k != shape[i][0] || L.stride != shape[i][1] || L.cin != shape[i][2] || L.cout != shape[i][3] || L.tin != t) return 0;
        if (i == 2 || i == 5 || i == 8) t = (t + 1) / 2;
    }
    return t0;
}

// Runtime-shape twins of fused_zero_halo / fused_conv0_s / fused_head_s for the generic walk (TCResNet14, other widths).
template <int NT>
__device__ __forceinline__ void fused_zero_halo_g(float* yout, const int out_sz, const int ng, const int cout, const int tout, const int tid_in) {
    const int tpo = tout + 2 * kHalo;
    const int tid = tid_in + opaque_zero();
    const float inv = 1.0f / (float)cout;
    for (int i = tid; i < ng * cout * 2; i += NT) {
        const int row = i >> 1;
        const int g = fast_div(row, cout, inv), co = row - g * cout;
        float* p = yout + g * out_sz + co * tpo + ((i & 1) ? kHalo + tout : 0);
        p[0] = 0.f; p[1] = 0.f; p[2] = 0.f; p[3] = 0.f;
    }
}

// first conv (3 x 1, stride 1, 40 input channels) from global memory with every operand of a job requested up front
template <int NW>
__device__ __forceinline__ void fused_conv0_g(const FusedArgs& a, const FusedLayer& L, const float* __restrict__ xin, const int in_sz,
                                              float* lds, const int ng, const int wave, const int r_in, const int q_in) {
    constexpr int K = 3, C4 = 10, PADLO = 1;
    const int oz = opaque_zero();
    const int r = r_in + oz, q = q_in + oz;
    const int cout = L.cout, tout = L.tout, tpi = L.tin + 2 * kHalo, tpo = tout + 2 * kHalo;
    const int wstep = 4 * cout, xstep = 4 * tpi;
    float* yout = lds + a.buf_off[L.out_buf];
    const int out_sz = L.out_sz;
    const int npos = ng * tout;
    const int nct = (npos + 15) / 16, nrt = (cout + 15) / 16;
    const float* w = a.params + L.w_off;
    const float* scale = a.ss + L.ss_off;
    const float* shift = scale + L.c_pad;
    const float inv_tout = 1.0f / (float)tout;
    for (int job = wave; job < nct * nrt; job += NW) {
        const int ct = job / nrt, m = job - ct * nrt;
        const int p0 = min(ct * 16 + r, npos - 1);
        const int g0 = fast_div(p0, tout, inv_tout);
        const int t0 = p0 - g0 * tout;
        const float* wp = w + q * cout + min(m * 16 + r, cout - 1);
        const float* x0 = xin + g0 * in_sz + q * tpi + t0 + kHalo - PADLO;
        float b0[C4][K], wf[K][C4];
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4)
#pragma unroll
            for (int j = 0; j < K; ++j) b0[c4][j] = x0[c4 * xstep + j];
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4) wf[j][c4] = wp[(j * C4 + c4) * wstep];
        float sc[4], sh[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int co = min(m * 16 + q * 4 + reg, cout - 1);
            sc[reg] = scale[co];
            sh[reg] = shift[co];
        }
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][c4], b0[c4][j], acc0, 0, 0, 0);
        const int dump = a.buf_off[2] + a.group * a.buf_sz[2] + (q * 16 + r) - a.buf_off[L.out_buf];
        const float lo = L.relu ? 0.f : -3.4e38f;
        const bool pv = ct * 16 + r < npos;
        const int base = g0 * out_sz + kHalo + t0;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int co = m * 16 + q * 4 + reg;
            const float v = fmaxf(fmaf(acc0[reg], sc[reg], sh[reg]), lo);
            yout[(pv && co < cout) ? base + co * tpo : dump] = v;
        }
    }
}

// head with runtime shapes: pool by all threads, fc / fc2 + softmax by one wave on the matrix cores (feat_c % 4 == 0, nc <= 14)
template <int NT>
__device__ __forceinline__ void fused_head_g(const FusedArgs& a, float* lds, const int n0, const int ng, const int tid_in) {
    const int tid = tid_in + opaque_zero();
    const int fc = a.feat_c, ft = a.feat_t, nc = a.nc;
    const int tp = ft + 2 * kHalo;
    const float* fb = lds + a.buf_off[a.feat_buf];
    const int fsz = a.feat_sz;
    float* pooled = lds + a.buf_off[(a.feat_buf + 1) % 3];
    const int lane = tid & 63, r = lane & 15, q = lane >> 4;
    const float inv_fc = 1.0f / (float)fc;
    for (int i = tid; i < ng * fc; i += NT) {
        const int g = fast_div(i, fc, inv_fc), c = i - g * fc;
        const float* row = fb + g * fsz + c * tp + kHalo;
        float s = 0.f;
        for (int t = 0; t < ft; ++t) s += row[t];
        pooled[i] = s / (float)ft;
    }
    __syncthreads();
    if (tid < 64) {
        const int g = min(r, ng - 1);
        const float* wsrc = r < nc ? a.params + a.fc_off + r : a.params + a.fc2_off + min(r - nc, 1);
        const int wstride = r < nc ? nc : 2;
        const bool wv = r < nc + 2;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int nst = fc >> 2;
        for (int s0 = 0; s0 < nst; s0 += 6) {           // six K-steps per trip: their 6 + 6 operand loads in flight together
            float af[6], bf[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int c = 4 * min(s0 + i, nst - 1) + q;
                af[i] = wsrc[c * wstride];
                bf[i] = pooled[g * fc + c];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (s0 + i < nst) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv ? af[i] : 0.f, bf[i], acc, 0, 0, 0);
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) if (4 * q + reg < nc) mx = fmaxf(mx, acc[reg]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float e[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) e[reg] = expf(acc[reg] - mx);
        float se = 0.f;
#pragma unroll
        for (int row = 0; row < 4; ++row) {             // class sum carried through the 16-lane rows in class order
            const float prev = __shfl(se, (row > 0 ? (row - 1) * 16 : 0) + r);
            float cur = row > 0 ? prev : 0.f;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) if (4 * row + reg < nc) cur += e[reg];
            if (q == row) se = cur;
        }
        se = __shfl(se, 48 + r);
        if (r < ng) {
            const size_t n = (size_t)(n0 + r);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int o = 4 * q + reg;
                if (o < nc) {
                    a.logits[n * nc + o] = acc[reg];
                    a.probs[n * nc + o] = e[reg] / se;
                } else if (o < nc + 2 && a.ranges) {
                    a.ranges[n * 2 + (o - nc)] = 1.0f / (1.0f + expf(-acc[reg]));
                }
            }
        }
    }
    __syncthreads();
}

template <int NW, int R>
__global__ __launch_bounds__(NW * 64) void net_fused_kernel(const FusedArgs a) {
    constexpr int NT = NW * 64;
    float* lds = reinterpret_cast<float*>(dyn_lds());
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // uniform: job bookkeeping runs on the scalar unit
    const int r = lane & 15, q = lane >> 4;

    for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
        const int n0 = grp * a.group;
        const int ng = min(a.group, a.batch - n0);
        const int row = a.in_c * a.in_tp;
        if (!a.in_global) {
            // ---- stage the group's feature rows (contiguous in global memory) ----
            const float4* src = reinterpret_cast<const float4*>(a.feat + (size_t)n0 * row);
            float4* dst = reinterpret_cast<float4*>(lds + a.buf_off[0]);
            const int nvec = ng * row / 4;
            for (int i = tid; i < nvec; i += NT) dst[i] = src[i];
            __syncthreads();
        }

        for (int li = 0; li < a.n_layers; ++li) {
            const FusedLayer L = a.layer[li];
            fused_zero_halo_g<NT>(lds + a.buf_off[L.out_buf], L.out_sz, ng, L.cout, L.tout, tid);
            if (li == 0 && a.in_global) {
                if (L.k == 3 && L.cin == 40 && L.stride == 1 && L.pad_lo == 1) fused_conv0_g<NW>(a, L, a.feat + (size_t)n0 * row, row, lds, ng, wave, r, q);
                else fused_layer<NW, R>(a, L, a.feat + (size_t)n0 * row, row, lds, ng, wave, r, q);
            } else {
                fused_layer<NW, R>(a, L, lds + a.buf_off[L.in_buf], L.in_sz, lds, ng, wave, r, q);
            }
            if (!L.no_barrier) __syncthreads();         // (a block's shortcut conv and its first conv read the same input: one phase)
        }

        if ((a.feat_c & 3) == 0 && a.nc + 2 <= 16) fused_head_g<NT>(a, lds, n0, ng, tid);
        else fused_head<NT>(a, lds, n0, ng, tid);
    }
}

// one utterance per group, TCResNet8-1.0 at 49 frames, few utterances: the small-batch kernel (weights through LDS); 1: not taken
static int launch_net_small(const FusedArgs& a0, hipStream_t s) {
    constexpr int kWA = 9 * 48 * 48, kWB = 32 * 48 + 9 * 32 * 48;       // the two largest consecutive phases: conv2_1 | down2 + conv2_0
    FusedArgs a = a0;
    // the weight copies are 16-byte DMA reads (global_load_lds_dwordx4) at params + w_off: a C-ABI caller's arena aligned to 4 bytes only
    // (or a layer whose weights do not start on a 16-byte boundary) takes the throughput kernel instead (advisor, round 5)
    if (reinterpret_cast<uintptr_t>(a.params) % 16 != 0) return 1;
    for (int li = 0; li < a.n_layers; ++li)
        if (a.layer[li].w_off % 4 != 0) return 1;
    const int act = a.buf_off[2] + a.buf_sz[2];                         // group == 1
    const int wa_off = (act + 64 + 63) / 64 * 64, wb_off = wa_off + kWA + 64;      // (+ pad: the layers' one-step operand lookahead)
    const size_t lds = ((size_t)wb_off + kWB + 64) * sizeof(float);
    if (lds > 160 * 1024) return 1;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(net_small_tc8_kernel<8, 49>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return 1;
        }
        configured = true;
    }
    hipLaunchKernelGGL((net_small_tc8_kernel<8, 49>), dim3(a.batch), dim3(512), lds, s, a, wa_off, wb_off);
    return check_launch("net_small_tc8_kernel");
}

int launch_net_fused(const FusedArgs& a, size_t lds_bytes, int grid, int waves, int ring, hipStream_t s) {
    void (*kern)(const FusedArgs) = nullptr;
    const int tc8 = tune_get(TCR_TUNE_NET_FUSED) == 3 ? 0 : fused_tc8_frames(a);
    if (tc8 == 49 && a.group == 1 && a.batch <= kSmallBatchMax && a.nc + 2 <= 16 && tune_get(TCR_TUNE_NET_FUSED) == 0 && tune_get(TCR_TUNE_NET_SMALL) == 0 &&
        tune_get(TCR_TUNE_FUSED_WAVES) == 0) {
        const int rc = launch_net_small(a, s);
        if (rc != 1) return rc;
    }
    // TCR_TUNE_NET_FUSED: 0 branch-free epilogue (default); 4: the round-2 static-shape kernel (A/B arm)
    const bool r2 = tune_get(TCR_TUNE_NET_FUSED) == 4;
    // Round 6 default: the nine-tap layers' work dealt in 16-position units (fused_layer_u) + a whole tap of weight lookahead (fused_job_s AHEAD) in
    // the layers of <= 24 input channels: 105.1 -> 98.6 -> 95.3 us at 49 frames, 182 -> 178 -> 175 at 98, bitwise; <= 32: 96.4 -> 95.3 / 178.3 -> 173.4;
    // every width: 96.5 (128 registers + 156 B of scratch).  TCResNet14-1.5's kernel: <= 48 input channels (98 frames 837 -> 804 us, 49 frames
    // 423.6 -> 421.7; 224 B of scratch).  A/B arms: 8 the jobs of two tiles dealt round-robin (rounds 3-5), 9 units without the lookahead.
    const bool j2 = tune_get(TCR_TUNE_NET_FUSED) == 8;
    const bool ju = tune_get(TCR_TUNE_NET_FUSED) == 9;
    const bool j4 = tune_get(TCR_TUNE_NET_FUSED) == 5;         // 5: four 16-position tiles per job in block 0's layers (A/B arm, bitwise; measured 103.6 vs 103.2 us at 49 frames, 184.9 vs 181.2 at 98: no gain)
#define TCR_FS(NW_, T_) if (tc8 == T_ && waves == NW_) kern = r2 ? net_fused_tc8_kernel<NW_, T_, -1> : (j4 ? net_fused_tc8_kernel<NW_, T_, 0, 4> : (ju ? net_fused_tc8_kernel<NW_, T_, 0, 0> : (j2 ? net_fused_tc8_kernel<NW_, T_, 0, 2> : net_fused_tc8_kernel<NW_, T_, 0, 3>)));
    TCR_FS(4, 49) TCR_FS(8, 49) TCR_FS(16, 49) TCR_FS(4, 98) TCR_FS(8, 98) TCR_FS(16, 98)
#undef TCR_FS
    const int tc14 = (kern || tune_get(TCR_TUNE_NET_FUSED) == 3 || tune_get(TCR_TUNE_NET_FUSED) == 4) ? 0 : fused_tc14w_frames(a);
#define TCR_F14(NW_, T_) if (tc14 == T_ && waves == NW_) kern = ju ? net_fused_tc14w_kernel<NW_, T_, 0> : (j2 ? net_fused_tc14w_kernel<NW_, T_, 2> : net_fused_tc14w_kernel<NW_, T_, 13>);
    TCR_F14(8, 49) TCR_F14(16, 49) TCR_F14(8, 98) TCR_F14(16, 98)
#undef TCR_F14
    if (kern) ring = 0;
#define TCR_FK(NW_, R_) if (!kern && waves == NW_ && ring == R_) kern = net_fused_kernel<NW_, R_>;
    TCR_FK(4, 4) TCR_FK(4, 8) TCR_FK(4, 16) TCR_FK(8, 4) TCR_FK(8, 8) TCR_FK(8, 16) TCR_FK(16, 4) TCR_FK(16, 8) TCR_FK(16, 16)
#undef TCR_FK
    if (!kern) { set_error("fused kernel: no instantiation for %d waves / ring %d", waves, ring); return TCR_ERR_ARG; }
    if (lds_bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
            (void)hipGetLastError();
            return 1;       // caller falls back to the per-layer kernels
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), lds_bytes, s, a);
    return check_launch("net_fused_kernel");
}

}  // namespace tcr
