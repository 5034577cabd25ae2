// TC-ResNet backward without BatchNorm passes (tf.gradients of audio_nets/tc_resnet.py:21-41 through slim.batch_norm(fused=True),
// as slim.learning.create_train_op builds it: helper/trainer.py:199-222; BN semantics of TCResNet_arg_scope, tc_resnet.py:102-123).
//
// The per-layer chain ran, for every BN unit:  chan_reduce (sum dz, sum dz xhat)  ->  bn_bwd_apply (finalize + write dy)  ->  data
// gradient (one launch per stride phase, + the block's shortcut conv accumulated behind it)  ||  filter gradient.  The first two are
// elementwise passes over HBM (58 % of a TCResNet8 step's bytes) that exist only because BN backward was a kernel of its own.
// Here a unit's dy is never written: see kernels.h (BwdLazyArgs).  One launch of this kernel
//   STAGE    for a group of G whole utterances, dy = k1 (dz - k2 - (raw - mean) k3) of the conv (and of the block's 1x1 shortcut conv)
//            straight into a zero-halo LDS image -- bn_bwd_apply's expression, coefficients from the finalize kernel;
//   CONV     every stride phase of the data gradient as a stride-1 convolution over that image on the exact-f32 16x16x4 MFMA
//            (D[row = input channel][col = position], A = phase-major re-arranged weights from L1/L2, B = LDS); a job is a
//            32-position column pair x ALL row tiles; its K range (taps x channel quads of conv_a, then of the shortcut conv: the two
//            gradients meet in the accumulators) is dealt in chunks to KS waves whose partial tiles are added through LDS in wave
//            order (fixed order: bitwise reproducible);
//   EPILOGUE (+ identity-shortcut gradient), the ReLU mask of the activation this is the gradient of, the store of the next unit's
//            gz, and that unit's (a block output: conv_b's and the shortcut's) backward sums -- a tile column's 16 positions are a
//            DPP row (row16_sum) -> the slot's LDS row -> ONE partial row per workgroup, summed in double by bn_bwd_finalize.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "kernels.h"

#ifndef TCR_LAZY_WHATIF
#define TCR_LAZY_WHATIF 0       // (timing experiments only, wrong results: 1 no weight loads, 2 no staging loads, 4 no epilogue loads, 8 no epilogue stores, 16 no MFMAs)
#endif

namespace tcr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLzNW = 8;            // waves per workgroup
constexpr int kLzCH = 4;            // K-steps (of 4 dy channels) per weight prefetch chunk
constexpr int kLzMaxMT = 6;         // row tiles: gradients of up to 96 channels

// A job is a 32-position column pair x ALL MT row tiles (one B fragment read from LDS feeds MT MFMAs); its K range (taps x channel quads
// of conv_a, then of the shortcut conv) is dealt in chunks of 4 steps to KS waves, weights one chunk ahead.  Measured against a variant
// with one 16-row tile per wave and every weight fragment requested up front (fewer exposed latencies, 3x the LDS reads and waves): this
// form is 10 % faster per TCResNet8 step -- kept.
template <int MT>
__global__ __launch_bounds__(kLzNW * 64) void bwd_lazy_kernel(const BwdLazyArgs a) {
    constexpr int NW = kLzNW, NT = NW * 64, CH = kLzCH;
    float* lds = reinterpret_cast<float*>(dyn_lds());
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int KS = a.ks, SLOTS = NW / KS;
    const int sl = wave / KS, ks = wave - sl * KS;          // (wave-uniform) which job of a round, which share of its K range
    const int KB = KS > 1 ? KS - 1 : 1;                     // tile buffers per slot
    float* red = lds + a.red_off;                           // [SLOTS][KB][MT * 8][64]: partial tiles; [slot][0] then holds the finished tile
    float* wstat = lds + a.stat_off;                        // [SLOTS][2 targets][2][cstat]
    float* coef = lds + a.coef_off;                         // per source: [c][8] = k1, k2, k3, mean, own-mask scale / shift
    float* ecoef = lds + a.ecoef_off;                       // per output channel: mean, invstd, own-mask scale / shift of the two sum targets
    const int n_src = a.n_layers;                           // (layer i reads source i)
#if defined(TCR_LAZY_TS)        // (diagnostic builds only: TCR_BUILD_EXTRA=-DTCR_LAZY_TS python tc-resnet_amd/build.py)
    int nstamp = 0;
    auto stamp = [&]() { if (a.dbg && tid == 0 && nstamp < 16) a.dbg[blockIdx.x * 16 + nstamp++] = clock64(); };
#else
    auto stamp = []() {};
#endif
    stamp();                                                // 0: start

    for (int i = tid; i < SLOTS * 4 * a.cstat; i += NT) wstat[i] = 0.f;
    for (int si = 0; si < n_src; ++si) {                    // the per-channel tables (packed rows of bn_bwd_finalize: two 16-byte halves per channel)
        const LazySrc& S = a.src[si];
        float* cf = coef + (si ? 8 * a.src[0].c : 0);
        for (int i = tid; i < S.c * 2; i += NT) reinterpret_cast<f32x4*>(cf)[i] = reinterpret_cast<const f32x4*>(S.tab)[i];
    }
    for (int i = tid; i < a.out_c; i += NT)
        for (int k = 0; k < 2; ++k) {
            const LazyStat& T = a.stat[k];
            ecoef[i * 8 + k * 4 + 0] = T.on ? T.mean[i] : 0.f;
            ecoef[i * 8 + k * 4 + 1] = T.on ? T.invstd[i] : 0.f;
            ecoef[i * 8 + k * 4 + 2] = (T.on && T.self_scale) ? T.self_scale[i] : 0.f;     // (no own mask: fmaf(raw, 0, 1) > 0 always)
            ecoef[i * 8 + k * 4 + 3] = (T.on && T.self_scale) ? T.self_shift[i] : 1.f;
        }
    const int tpo = a.out_t + 2 * kHalo;
    const int S_out = a.layer[0].stride;
    constexpr int SU = 4;                                   // staged elements per thread and trip: every load issued before the first use

    for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
        const int n0 = grp * a.group;
        const int ng = min(a.group, a.batch - n0);
        stamp();                                            // 1
        __syncthreads();                                    // (tables staged; the previous group is done with the images and tiles)
        stamp();                                            // 2
        // ---- STAGE: dy of the group's utterances -> zero-halo LDS rows (same row pitch as in global memory) ----
        // (Requesting a first trip of the staging list ahead of this barrier and the table fill hides one memory round trip but keeps
        //  16 more registers live across them: 61 -> 97 VGPRs in the MT = 1 instance, half the workgroups per CU, and the TCResNet8 step
        //  got 70 us SLOWER.)
        for (int si = 0; si < n_src; ++si) {
            const LazySrc& S = a.src[si];
            const int tp = S.t + 2 * kHalo;
            float* img = lds + a.img_off[si];
            const float* cf = coef + (si ? 8 * a.src[0].c : 0);
            const size_t gbase = (size_t)n0 * S.c * tp;
            const int nrows = ng * S.c;
            // The group's rows are ONE contiguous block of the planar layout (halos included) and the image has the same pitch: 16-byte
            // loads of gz and raw, all of a trip's in flight before the first use, one 16-byte LDS store per four elements; halo positions
            // are stored as zero by a select (no separate zero pass).  (Round-4 first form: a dword gather per interior element, two
            // trips of four: 2.5x the vector-memory instructions and one more round trip per staging.)
            if (a.vec_stage) {
                const int tot4 = nrows * tp / 4;
                const float inv_tp = 1.0f / (float)tp, inv_c = 1.0f / (float)S.c;
                const f32x4* g4 = reinterpret_cast<const f32x4*>(S.gz + gbase);
                const f32x4* r4 = reinterpret_cast<const f32x4*>(S.raw + gbase);
                constexpr int SV = 2;
                for (int f0 = tid; f0 < tot4; f0 += NT * SV) {
                    f32x4 vg[SV], vr[SV];
#pragma unroll
                    for (int u = 0; u < SV; ++u) {
                        const int f = min(f0 + u * NT, tot4 - 1);
                        vg[u] = (TCR_LAZY_WHATIF & 2) ? (f32x4){1.f, 1.f, 1.f, 1.f} : g4[f];
                        vr[u] = (TCR_LAZY_WHATIF & 2) ? (f32x4){1.f, 1.f, 1.f, 1.f} : r4[f];
                    }
#pragma unroll
                    for (int u = 0; u < SV; ++u) {
                        const int f = f0 + u * NT;
                        if (f >= tot4) break;
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int idx = 4 * f + e;
                            const int row = fast_div(idx, tp, inv_tp);
                            const int tt = idx - row * tp - kHalo;
                            const int g = fast_div(row, S.c, inv_c);
                            const float* cp = cf + (row - g * S.c) * 8;
                            float dz = vg[u][e];
                            const float y = vr[u][e];
                            if (!(fmaf(y, cp[4], cp[5]) > 0.f)) dz = 0.f;
                            const float d = cp[0] * (dz - cp[1] - (y - cp[3]) * cp[2]);      // == bn_bwd_apply_kernel
                            o[e] = (tt >= 0 && tt < S.t) ? d : 0.f;
                        }
                        reinterpret_cast<f32x4*>(img)[f] = o;
                    }
                }
                continue;
            }
            const int total = nrows * S.t;
            const float inv_t = 1.0f / (float)S.t, inv_c = 1.0f / (float)S.c;
            for (int e0 = tid; e0 < total; e0 += NT * SU) {
                float vg[SU], vr[SU];
                int rel[SU], ch[SU];
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int e = min(e0 + u * NT, total - 1);
                    const int row = fast_div(e, S.t, inv_t);
                    const int g = fast_div(row, S.c, inv_c);
                    ch[u] = row - g * S.c;
                    rel[u] = row * tp + kHalo + (e - row * S.t);
                    vg[u] = (TCR_LAZY_WHATIF & 2) ? 1.0f : S.gz[gbase + rel[u]];
                    vr[u] = (TCR_LAZY_WHATIF & 2) ? 1.0f : S.raw[gbase + rel[u]];
                }
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    if (e0 + u * NT >= total) break;
                    const float* cp = cf + ch[u] * 8;
                    float dz = vg[u];
                    if (!(fmaf(vr[u], cp[4], cp[5]) > 0.f)) dz = 0.f;
                    img[rel[u]] = cp[0] * (dz - cp[1] - (vr[u] - cp[3]) * cp[2]);      // == bn_bwd_apply_kernel
                }
            }
            for (int i = tid; i < nrows * 2 * kHalo; i += NT) {
                const int row = i / (2 * kHalo), h = i - row * (2 * kHalo);
                img[row * tp + (h < kHalo ? h : S.t + h)] = 0.f;
            }
        }
        stamp();                                            // 3: staging done (this thread)
        __syncthreads();
        stamp();                                            // 4
        // ---- CONV + EPILOGUE: jobs (phase, 32-position pair), SLOTS at a time ----
        const int ncp0 = (ng * a.nu[0] + 31) >> 5;
        const int ncp1 = S_out > 1 ? (ng * a.nu[1] + 31) >> 5 : 0;
        const int npairs = ncp0 + ncp1;
        for (int rd = 0; rd * SLOTS < npairs; ++rd) {
            const int pi = rd * SLOTS + sl;
            const bool valid = pi < npairs;
            const int ph = (valid && pi >= ncp0) ? 1 : 0;
            const int cp = ph ? pi - ncp0 : pi;
            const int nu = a.nu[ph];
            const int npos = ng * nu;
            f32x4 acc[MT][2];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            int gg[2], uu[2];
            {
                const float inv_nu = 1.0f / (float)nu;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int p = min(cp * 32 + nt * 16 + r, npos - 1);
                    gg[nt] = fast_div(p, nu, inv_nu);
                    uu[nt] = p - gg[nt] * nu;
                }
            }
            if (valid) {
                int chunk_base = 0;
                for (int li = 0; li < a.n_layers; ++li) {
                    const int cnt = a.cnt[li][ph];
                    if (cnt == 0) continue;
                    const LazySrc& S = a.src[li];
                    const int tp = S.t + 2 * kHalo;
                    const int C4 = S.c >> 2;
                    const float* img = lds + a.img_off[li];
                    int xo[2];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) xo[nt] = (gg[nt] * S.c + q) * tp + uu[nt] + kHalo + a.dmin[li][ph];
                    // A[row = ci][k = co]: wt entry [tap][co][ci]; lane (r, q) reads row m * 16 + r (clamped: rows past out_c are never
                    // stored), k = q of the step's channel quad
                    // (round 6: the weights through a buffer descriptor -- uniform base, constant lane offset, uniform running offset: no
                    //  64-bit vector address arithmetic in front of the CH x MT loads of a chunk; gfx950_isa.h: buf_load_f32)
                    const buf_rsrc wl = make_rsrc(a.layer[li].wt + a.wbase[li][ph]);
                    unsigned wq[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) wq[m] = (unsigned)(q * a.out_c + min(m * 16 + r, a.out_c - 1)) * 4u;
                    const int tap_stride = S.c * a.out_c, step_stride = 4 * a.out_c, xq = 4 * tp;
                    const int cpj = (C4 + CH - 1) / CH;
                    const int nch = cnt * cpj;
                    auto load_chunk = [&](int j, int c0, float (&af)[CH][MT]) {
#pragma unroll
                        for (int i = 0; i < CH; ++i) {
                            const int c4 = min(c0 + i, C4 - 1);                 // (tail: re-read the last quad; not multiplied)
#pragma unroll
                            for (int m = 0; m < MT; ++m) af[i][m] = (TCR_LAZY_WHATIF & 1) ? 1.0f : buf_load_f32(wl, wq[m], (unsigned)(j * tap_stride + c4 * step_stride) * 4u);
                        }
                    };
                    auto mma_chunk = [&](int j, int c0, const float (&af)[CH][MT]) {
#pragma unroll
                        for (int i = 0; i < CH; ++i) {
                            if (c0 + i < C4) {
                                const float b0 = img[xo[0] + j + (c0 + i) * xq];
                                const float b1 = img[xo[1] + j + (c0 + i) * xq];
                                if (TCR_LAZY_WHATIF & 16) {
#pragma unroll
                                    for (int m = 0; m < MT; ++m) { acc[m][0][0] += af[i][m] + b0; acc[m][1][0] += af[i][m] + b1; }
                                    continue;
                                }
#pragma unroll
                                for (int m = 0; m < MT; ++m) {
                                    acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][m], b0, acc[m][0], 0, 0, 0);
                                    acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][m], b1, acc[m][1], 0, 0, 0);
                                }
                            }
                        }
                    };
                    // this wave's chunks of the layer: global chunk ids == ks (mod KS)
                    int c = ((ks - chunk_base) % KS + KS) % KS;
                    chunk_base += nch;
                    int j = 0, cq = c;
                    auto norm = [&]() { while (cq >= cpj) { cq -= cpj; ++j; } };
                    norm();
                    float afA[CH][MT], afB[CH][MT];
                    if (c < nch) load_chunk(j, cq * CH, afA);
                    for (; c < nch; c += 2 * KS) {
                        const int ja = j, cqa = cq;
                        cq += KS; norm();
                        const int jb = j, cqb = cq;
                        if (c + KS < nch) load_chunk(jb, cqb * CH, afB);
                        mma_chunk(ja, cqa * CH, afA);
                        cq += KS; norm();
                        if (c + 2 * KS < nch) load_chunk(j, cq * CH, afA);
                        if (c + KS < nch) mma_chunk(jb, cqb * CH, afB);
                    }
                }
            }
            // partial tiles of waves 1 .. KS-1 -> LDS -> added by wave 0 of the slot in wave order; the finished tile goes back to LDS
            // (the slot's first buffer: every lane re-writes exactly the cells it has just read), from where ALL waves run the epilogue
            if (KS > 1) {
                if (valid && ks > 0) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg)
                                red[(((sl * KB + ks - 1) * MT + m) * 8 + nt * 4 + reg) * 64 + lane] = acc[m][nt][reg];
                }
                __syncthreads();
            }
            if (valid && ks == 0) {
                for (int k2 = 1; k2 < KS; ++k2)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg)
                                acc[m][nt][reg] += red[(((sl * KB + k2 - 1) * MT + m) * 8 + nt * 4 + reg) * 64 + lane];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg)
                            red[((sl * KB * MT + m) * 8 + nt * 4 + reg) * 64 + lane] = acc[m][nt][reg];
            }
            stamp();                                        // 5: matrix work + tile
            __syncthreads();
            stamp();                                        // 6
            // ---- EPILOGUE, all waves: item = (slot, channel) -> one 16-lane group (a DPP row) walks the pair's 2 x 16 positions ----
            {
                const int nsv = min(SLOTS, npairs - rd * SLOTS);        // slots that held a pair this round
                const int items = nsv * a.out_c;
                const size_t gofs = (size_t)n0 * a.out_c * tpo;         // every tensor of the epilogue has the output's layout: one
                float* og = a.out_g + gofs;                             // 32-bit offset per element, wave-uniform base pointers
                const float* addp = a.add ? a.add + gofs : nullptr;
                const float* mkp = a.mask_act ? a.mask_act + gofs : nullptr;
                const float* rp0 = a.stat[0].on ? a.stat[0].raw + gofs : nullptr;
                const float* rp1 = a.stat[1].on ? a.stat[1].raw + gofs : nullptr;
                const float inv_oc = 1.0f / (float)a.out_c;
                for (int it = tid >> 4; it < items; it += NT / 16) {
                    const int slot = fast_div(it, a.out_c, inv_oc);
                    const int co = it - slot * a.out_c;
                    const int pj = rd * SLOTS + slot;
                    const int php = pj >= ncp0 ? 1 : 0;
                    const int cpp = php ? pj - ncp0 : pj;
                    const int nup = a.nu[php];
                    const int nposp = ng * nup;
                    const float inv_nup = 1.0f / (float)nup;
                    const float* tile = red + (size_t)(slot * KB * MT + (co >> 4)) * 8 * 64 + (co & 3) * 64 + ((co >> 2) & 3) * 16 + r;
                    const float* ec = ecoef + co * 8;
                    float q1[2] = {0.f, 0.f}, q2[2] = {0.f, 0.f};
                    float v[2], av[2], mk[2], rw0[2], rw1[2];
                    int o[2];
                    bool ok[2];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int p = cpp * 32 + nt * 16 + r;
                        ok[nt] = p < nposp;
                        const int pc = min(p, nposp - 1);
                        const int g = fast_div(pc, nup, inv_nup);
                        o[nt] = (g * a.out_c + co) * tpo + kHalo + (pc - g * nup) * S_out + php;
                        v[nt] = tile[nt * 4 * 64];
                        av[nt] = (addp && !(TCR_LAZY_WHATIF & 4)) ? addp[o[nt]] : 0.f;
                        mk[nt] = (mkp && !(TCR_LAZY_WHATIF & 4)) ? mkp[o[nt]] : 1.f;
                        rw0[nt] = (rp0 && !(TCR_LAZY_WHATIF & 4)) ? rp0[o[nt]] : 0.f;
                        rw1[nt] = (rp1 && !(TCR_LAZY_WHATIF & 4)) ? rp1[o[nt]] : 0.f;
                    }
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        float x = v[nt] + av[nt];
                        if (!(mk[nt] > 0.f)) x = 0.f;
                        const float dz0 = (ok[nt] && fmaf(rw0[nt], ec[2], ec[3]) > 0.f) ? x : 0.f;
                        const float dz1 = (ok[nt] && fmaf(rw1[nt], ec[6], ec[7]) > 0.f) ? x : 0.f;
                        q1[0] += dz0; q2[0] = fmaf(dz0, (rw0[nt] - ec[0]) * ec[1], q2[0]);
                        q1[1] += dz1; q2[1] = fmaf(dz1, (rw1[nt] - ec[4]) * ec[5], q2[1]);
                        if (ok[nt] && !(TCR_LAZY_WHATIF & 8)) og[o[nt]] = a.store_self ? dz0 : x;
                    }
                    float* ws = wstat + slot * 4 * a.cstat;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        if (!a.stat[k].on) continue;
                        const float t1 = row16_sum(q1[k]), t2 = row16_sum(q2[k]);
                        if (r == 0) {                       // (slot, channel) always lands on the same lane group: fixed order
                            ws[(k * 2 + 0) * a.cstat + co] += t1;
                            ws[(k * 2 + 1) * a.cstat + co] += t2;
                        }
                    }
                }
            }
            stamp();                                        // 7: epilogue (this thread)
            __syncthreads();
            stamp();                                        // 8
        }
    }
    __syncthreads();
    // ---- one partial row per workgroup and target: the slots' rows added in slot order ----
    for (int k = 0; k < 2; ++k) {
        if (!a.stat[k].on) continue;
        for (int i = tid; i < 2 * a.out_c; i += NT) {
            const int which = i / a.out_c, co = i - which * a.out_c;
            float s = 0.f;
            for (int sv = 0; sv < SLOTS; ++sv) s += wstat[(sv * 4 + k * 2 + which) * a.cstat + co];
            a.stat[k].partial[((size_t)blockIdx.x * 2 + which) * a.out_c + co] = s;
        }
    }
}

// taps / first dy offset of output phase ph of a (k, stride, pad_lo) data gradient (launch_conv_dgrad_mfma's arithmetic)
static void lazy_phase(int k, int stride, int pad_lo, int ph, int* cnt, int* d_min) {
    const int res_mod = (ph + pad_lo) % stride;
    const int jmax = (k - 1) - (((k - 1) - res_mod + stride) % stride);
    *cnt = jmax < 0 ? 0 : jmax / stride + 1;
    *d_min = jmax < 0 ? 0 : (ph + pad_lo - jmax) / stride;
}

// fills the launch geometry; false when the shape is not covered (the caller keeps the per-layer chain)
static bool configure_lazy(BwdLazyArgs& a, size_t* lds_out, int* grid_out) {
    if (a.n_layers < 1 || a.n_layers > 2 || a.out_c <= 0 || a.out_c > 16 * kLzMaxMT || a.out_t <= 0) return false;
    const int S = a.layer[0].stride;
    if (S < 1 || S > 2) return false;
    for (int li = 0; li < a.n_layers; ++li) {
        const LazyLayer& L = a.layer[li];
        const LazySrc& src = a.src[li];
        if (L.src != li || L.stride != S || L.k < 1 || L.k > 9 || src.c % 4 != 0 || src.c <= 0 || src.t <= 0) return false;
        if ((int64_t)src.c * (src.t + 2 * kHalo) * 16 >= (1 << 22)) return false;      // (fast_div range of the staging index)
    }
    a.mt = ceil_div(a.out_c, 16);
    for (int ph = 0; ph < 2; ++ph) {
        a.nu[ph] = ph < S ? (a.out_t - ph + S - 1) / S : 0;
        for (int li = 0; li < 2; ++li) { a.cnt[li][ph] = 0; a.dmin[li][ph] = 0; a.wbase[li][ph] = 0; }
    }
    if (a.nu[0] <= 0 || (S > 1 && a.nu[1] <= 0)) return false;
    int64_t chunks[2] = {0, 0};                             // K chunks of a job of phase 0 / 1
    for (int li = 0; li < a.n_layers; ++li) {
        const LazyLayer& L = a.layer[li];
        const LazySrc& src = a.src[li];
        int base = 0;
        for (int ph = 0; ph < S; ++ph) {
            int cnt, dmin;
            lazy_phase(L.k, S, L.pad_lo, ph, &cnt, &dmin);
            a.cnt[li][ph] = cnt; a.dmin[li][ph] = dmin; a.wbase[li][ph] = base * src.c * a.out_c;
            base += cnt;
            if (cnt == 0) continue;
            // every dy position a tap reads lies inside the zero-halo row
            if (dmin < -kHalo || (a.nu[ph] - 1) + (cnt - 1) + dmin > src.t - 1 + kHalo) return false;
            chunks[ph] += (int64_t)cnt * ceil_div(src.c >> 2, kLzCH);
        }
    }
    a.cstat = max(16, a.mt * 16);
    // Geometry: G whole utterances per group, KS waves per job.  Cost model (relative): a group costs its staging plus, per round of
    // SLOTS jobs, the longest wave's K chunks and the epilogue; groups run ~two per CU at a time.  TCR_TUNE_BWD_LAZY_CFG = G + 100 KS overrides.
    int per_utt = 0;
    for (int li = 0; li < a.n_layers; ++li) per_utt += a.src[li].c * (a.src[li].t + 2 * kHalo);
    const int coef_floats = 8 * (a.src[0].c + (a.n_layers > 1 ? a.src[1].c : 0));
    auto lds_of = [&](int g, int ks) {
        const int slots = kLzNW / ks;
        return ((size_t)g * per_utt + (size_t)slots * (ks > 1 ? ks - 1 : 1) * a.mt * 8 * 64 + (size_t)slots * 4 * a.cstat + coef_floats + 8 * a.out_c + 64) * sizeof(float);
    };
    int knob = tune_get(TCR_TUNE_BWD_LAZY_CFG);             // G + 100 KS + 10000 (out_c * 10 + n_layers: only that kernel; 0: all)
    if (knob / 10000 > 0 && knob / 10000 != a.out_c * 10 + a.n_layers) knob = 0;
    knob %= 10000;
    int best_g = 0, best_ks = 0;
    double best = 0.0;
    // Measured geometries of the BASELINE.json shapes (TCResNet8-1.0, batch 4096; scripts/sweep_lazy_cfg.py: one kernel varied at a time,
    // whole training step timed): {out_c, out_t, layers, src c, src t} -> {G, KS}.  Every other shape takes the cost model below.
    // (Round 6, re-swept with the software-pipelined filter gradients beside these kernels: {16, 49, 2}: G 3 -> 6, {32, 13, 1}: G 6 / KS 4 ->
    //  8 / 8, {32, 13, 2}: 5 / 2 -> 4 / 4 -- together 839..845 -> 807 us per TCResNet8 step; a second pass over all six: within noise.)
    static const int kMeasured[][7] = {
        {16, 49, 2, 24, 25, 6, 4}, {24, 25, 1, 24, 25, 6, 1}, {24, 25, 2, 32, 13, 6, 2},
        {32, 13, 1, 32, 13, 8, 8}, {32, 13, 2, 48, 7, 4, 4},  {48, 7, 1, 48, 7, 8, 2},
        {16, 98, 2, 24, 49, 4, 1}, {32, 25, 1, 32, 25, 5, 2},     // (98 frames: the two kernels whose best beat the model by > 1 %)
    };
    if (knob == 0 && a.batch >= 1024)
        for (const auto& mrow : kMeasured)
            if (mrow[0] == a.out_c && mrow[1] == a.out_t && mrow[2] == a.n_layers && mrow[3] == a.src[0].c && mrow[4] == a.src[0].t)
                knob = mrow[6] * 100 + mrow[5];
    // The search below is pure in (shape, batch, knob): its result is memoised -- the plan of a net is asked for several times per BN unit
    // and backward (rows of the finalize passes, the coverage test, the launch), always with the same answer.
    struct GeoKey { int v[16]; };
    struct GeoVal { GeoKey key; int g, ks; };
    static std::mutex geo_mu;
    static std::vector<GeoVal> geo_cache;
    GeoKey key = {{a.out_c, a.out_t, a.n_layers, S, a.batch, knob, a.layer[0].k, a.layer[0].pad_lo, a.src[0].c, a.src[0].t,
                   a.n_layers > 1 ? a.layer[1].k : 0, a.n_layers > 1 ? a.layer[1].pad_lo : 0, a.n_layers > 1 ? a.src[1].c : 0, a.n_layers > 1 ? a.src[1].t : 0,
                   device_cus(), 0}};
    bool cached = false;
    {
        std::lock_guard<std::mutex> lock(geo_mu);
        for (const GeoVal& e : geo_cache)
            if (std::memcmp(&e.key, &key, sizeof(key)) == 0) { best_g = e.g; best_ks = e.ks; cached = true; break; }
    }
    for (int g = 1; g <= 16 && !cached; ++g) {
        if (knob % 100 > 0 && g != knob % 100) continue;
        const int g_eff = min(g, a.batch);
        const int pairs0 = ceil_div(g_eff * a.nu[0], 32), pairs1 = S > 1 ? ceil_div(g_eff * a.nu[1], 32) : 0;
        for (int ks = 1; ks <= kLzNW; ks *= 2) {
            if (knob / 100 > 0 && ks != knob / 100) continue;
            const size_t lds = lds_of(g, ks);
            if (lds > 64 * 1024) continue;
            const int slots = kLzNW / ks;
            const int rounds = ceil_div(pairs0 + pairs1, slots);
            const int64_t cmaxk = chunks[0] > chunks[1] ? chunks[0] : chunks[1];
            const double per_round = (double)ceil_div((int)cmaxk, ks) * (1.0 + 0.5 * a.mt) + 6.0 + 1.5 * a.mt + (ks > 1 ? 2.0 + 0.5 * a.mt : 0.0);
            const double stage = (double)g * per_utt / 512.0 * 0.6 + 3.0;
            const double group_cost = stage + rounds * per_round;
            const int n_groups = ceil_div(a.batch, g);
            const int per_cu = (160 * 1024) / lds >= 3 ? 3 : (int)((160 * 1024) / lds);
            const double waves = (double)ceil_div(n_groups, device_cus() * max(per_cu, 1));
            const double cost = group_cost * waves;
            if (best_g == 0 || cost < best) { best = cost; best_g = g; best_ks = ks; }
        }
    }
    if (!cached) {
        std::lock_guard<std::mutex> lock(geo_mu);
        if (geo_cache.size() >= 256) geo_cache.clear();
        geo_cache.push_back(GeoVal{key, best_g, best_ks});
    }
    if (best_g == 0) return false;
    a.group = best_g; a.ks = best_ks; a.nw = kLzNW; a.n_groups = ceil_div(a.batch, a.group);
    const int slots = kLzNW / a.ks;
    a.img_off[0] = 0;
    a.img_off[1] = a.group * a.src[0].c * (a.src[0].t + 2 * kHalo);
    a.vec_stage = tune_get(TCR_TUNE_LAZY_STAGE) != 1;
    for (int i = 0; i < a.n_layers; ++i) {
        const LazySrc& S = a.src[i];
        if ((S.c * (S.t + 2 * kHalo)) % 4 != 0 || (reinterpret_cast<uintptr_t>(S.gz) & 15) || (reinterpret_cast<uintptr_t>(S.raw) & 15)) a.vec_stage = 0;
    }
    a.red_off = a.group * per_utt;
    a.stat_off = a.red_off + slots * (a.ks > 1 ? a.ks - 1 : 1) * a.mt * 8 * 64;
    a.coef_off = a.stat_off + slots * 4 * a.cstat;
    a.ecoef_off = a.coef_off + coef_floats;
    *lds_out = lds_of(a.group, a.ks);
    *grid_out = min(a.n_groups, kPhaseMaxRows);
    static const bool debug_lazy = getenv("TCR_DEBUG_LAZY") != nullptr;
    if (debug_lazy)
        fprintf(stderr, "bwd_lazy: out %dx%d layers %d src %dx%d stride %d -> group %d ks %d mt %d lds %zu grid %d chunks %lld/%lld\n", a.out_c, a.out_t, a.n_layers,
                a.src[0].c, a.src[0].t, S, a.group, a.ks, a.mt, *lds_out, *grid_out, (long long)chunks[0], (long long)chunks[1]);
    return true;
}

int bwd_lazy_rows(const BwdLazyArgs& a0) {
    BwdLazyArgs a = a0;
    size_t lds;
    int grid;
    return configure_lazy(a, &lds, &grid) ? grid : -1;
}

int launch_bwd_lazy(BwdLazyArgs a, int* rows_out, hipStream_t s) {
    size_t lds;
    int grid;
    if (!configure_lazy(a, &lds, &grid)) return 1;
    if (rows_out) *rows_out = grid;
    const dim3 g(grid), b(a.nw * 64);
    a.dbg = nullptr;
#if defined(TCR_LAZY_TS)
    static long long* dbg_dev = nullptr;
    if (getenv("TCR_DEBUG_LAZY_TS")) {
        if (!dbg_dev) (void)hipMalloc(reinterpret_cast<void**>(&dbg_dev), 1024 * 16 * sizeof(long long));
        (void)hipMemsetAsync(dbg_dev, 0, 1024 * 16 * sizeof(long long), s);
        a.dbg = dbg_dev;
    }
#endif
    switch (a.mt) {
        case 1: hipLaunchKernelGGL((bwd_lazy_kernel<1>), g, b, lds, s, a); break;
        case 2: hipLaunchKernelGGL((bwd_lazy_kernel<2>), g, b, lds, s, a); break;
        case 3: hipLaunchKernelGGL((bwd_lazy_kernel<3>), g, b, lds, s, a); break;
        case 4: hipLaunchKernelGGL((bwd_lazy_kernel<4>), g, b, lds, s, a); break;
        case 5: hipLaunchKernelGGL((bwd_lazy_kernel<5>), g, b, lds, s, a); break;
        default: hipLaunchKernelGGL((bwd_lazy_kernel<6>), g, b, lds, s, a); break;
    }
#if defined(TCR_LAZY_TS)
    if (a.dbg) {
        static long long host[1024 * 16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(host, dbg_dev, sizeof(host), hipMemcpyDeviceToHost);
        long long first = host[0];
        for (int w = 0; w < grid; ++w) if (host[w * 16] && host[w * 16] < first) first = host[w * 16];
        for (int w : {0, grid / 3, grid - 1}) {
            fprintf(stderr, "  lazy ts out %dx%d wg %4d: start %+7lld |", a.out_c, a.out_t, w, host[w * 16] - first);
            for (int i = 1; i < 16 && host[w * 16 + i]; ++i) fprintf(stderr, " %lld", host[w * 16 + i] - host[w * 16 + i - 1]);
            fprintf(stderr, "\n");
        }
    }
#endif
    return check_launch("bwd_lazy_kernel");
}

}  // namespace tcr
