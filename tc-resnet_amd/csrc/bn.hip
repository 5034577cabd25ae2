// Batch normalisation for the planar halo layout: eval-mode folding, train-mode batch statistics
// (per-channel reduction over batch x time), normalise + ReLU + residual, and the backward pass.
//
// Replaces slim.batch_norm(fused=True) -> FusedBatchNorm / FusedBatchNormGrad as configured by
// TCResNet_arg_scope (audio_nets/tc_resnet.py:102-123): decay 0.997, epsilon 1e-3, center, scale;
// training normalises with the biased batch variance and moves the running variance towards the
// Bessel-corrected one.  tf.nn.relu and the residual add of tc_resnet.py:40-41 are fused here.
//
// Per-channel reductions: lane == position (coalesced along time), CT channels per lane in
// registers, 64-lane shuffle reduction, LDS across the 4 waves, one partial row per workgroup that
// a single-workgroup kernel sums in double precision in a FIXED order (bitwise reproducible).
#include "kernels.h"

namespace tcr {

// Eval mode: y = gamma * (x - mm) / sqrt(mv + eps) + beta  ==  x * scale + shift
__global__ __launch_bounds__(128) void bn_fold_kernel(const BnFoldArgs a) {
    const int l = blockIdx.x;
    for (int c = threadIdx.x; c < a.c[l]; c += 128) {
        const float g = a.gamma_off[l] >= 0 ? a.params[a.gamma_off[l] + c] : 1.0f, b = a.params[a.beta_off[l] + c];
        const float mm = a.stats[a.mean_off[l] + c], mv = a.stats[a.var_off[l] + c];
        const float cb = a.bias_off[l] >= 0 ? a.params[a.bias_off[l] + c] : 0.0f;
        const float sc = g / sqrtf(mv + a.eps);
        a.out[a.out_off[l] + c] = sc;
        a.out[a.out_off[l] + a.c_pad[l] + c] = fmaf(cb - mm, sc, b);
    }
}

int launch_bn_fold(const BnFoldArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3(a.n), dim3(128), 0, s, a);
    return check_launch("bn_fold_kernel");
}

typedef float bn_f4 __attribute__((ext_vector_type(4)));

// 16-byte (four elements per thread) variants apply: pointers aligned, rows of an utterance a multiple of four floats
// (TCR_TUNE_BWD_MASK = 2: scalar elementwise kernels, bitwise the same; 3: also the scalar reduction kernel, another summation order)
static bool bn_vec4_ok(const void* p0, const void* p1, const void* p2, const void* p3, int per_utt, bool reduction = false) {
    auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const int knob = tune_get(TCR_TUNE_BWD_MASK);
    return per_utt % 4 == 0 && al(p0) && al(p1) && al(p2) && al(p3) && knob != 3 && (reduction || knob != 2);
}

// ---------------------------------------------------------------------------------------------
// Generic per-channel reduction of up to two quantities over [B][C][Tp].
//   MODE 0 (forward stats):   q1 = y, q2 = y*y
//   MODE 1 (backward sums):   dz = dA * [m1 > 0] * [m2 > 0];  q1 = dz, q2 = dz * (y - mean) * invstd
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const ChanReduceArgs a) {
    constexpr int CT = 8;
    __shared__ float s_red[4][2 * CT];
    const int c0 = blockIdx.y * CT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float q1[CT], q2[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) { q1[c] = 0.f; q2[c] = 0.f; }
    float mu[CT], is[CT], ssc[CT], ssh[CT];
    const bool self = MODE >= 1 && a.self_scale != nullptr;
    if (MODE >= 1) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int cc = min(c0 + c, a.c - 1);
            mu[c] = a.mean[cc];
            is[c] = a.invstd[cc];
            ssc[c] = self ? a.self_scale[cc] : 0.f;
            ssh[c] = self ? a.self_shift[cc] : 1.f;
        }
    }
    // MODE 2: the second unit (same dz, its own mask and xhat)
    float r1[MODE == 2 ? CT : 1], r2[MODE == 2 ? CT : 1], mu2[MODE == 2 ? CT : 1], is2[MODE == 2 ? CT : 1], sc2[MODE == 2 ? CT : 1], sh2[MODE == 2 ? CT : 1];
    if (MODE == 2) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int cc = min(c0 + c, a.c - 1);
            r1[MODE == 2 ? c : 0] = 0.f; r2[MODE == 2 ? c : 0] = 0.f;
            mu2[MODE == 2 ? c : 0] = a.mean2[cc];
            is2[MODE == 2 ? c : 0] = a.invstd2[cc];
            sc2[MODE == 2 ? c : 0] = a.self_scale2[cc];
            sh2[MODE == 2 ? c : 0] = a.self_shift2[cc];
        }
    }
    const int blk0 = blockIdx.x * a.pos_per_block;
    const int blk1 = min(blk0 + a.pos_per_block, a.npos);
    const float inv_t = 1.0f / (float)a.t;       // p / t: float multiply + one-step fix-up instead of an integer division
    for (int p = blk0 + threadIdx.x; p < blk1; p += 256) {
        int n = (int)(((float)p + 0.5f) * inv_t);
        n += (n + 1) * a.t <= p ? 1 : (n * a.t > p ? -1 : 0);
        const int t = p - n * a.t;
        const size_t base = ((size_t)n * a.c + c0) * a.tp + kHalo + t;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (c0 + c >= a.c) continue;
            const size_t o = base + (size_t)c * a.tp;
            const float yv = a.y[o];
            if (MODE == 0) {
                q1[c] += yv;
                q2[c] = fmaf(yv, yv, q2[c]);
            } else {
                float dz = a.bcast ? a.da[(size_t)n * a.c + c0 + c] : a.da[o];
                if (a.m1 && !(a.m1[o] > 0.f)) dz = 0.f;
                if (a.m2 && !(a.m2[o] > 0.f)) dz = 0.f;
                if (self && !(fmaf(yv, ssc[c], ssh[c]) > 0.f)) dz = 0.f;
                if (a.g_out) a.g_out[o] = dz;
                q1[c] += dz;
                q2[c] = fmaf(dz, (yv - mu[c]) * is[c], q2[c]);
                if (MODE == 2) {            // (the expressions of a MODE 1 pass over g_out with this unit's self mask)
                    const float y2 = a.y2[o];
                    float dz2 = dz;
                    if (!(fmaf(y2, sc2[MODE == 2 ? c : 0], sh2[MODE == 2 ? c : 0]) > 0.f)) dz2 = 0.f;
                    r1[MODE == 2 ? c : 0] += dz2;
                    r2[MODE == 2 ? c : 0] = fmaf(dz2, (y2 - mu2[MODE == 2 ? c : 0]) * is2[MODE == 2 ? c : 0], r2[MODE == 2 ? c : 0]);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        q1[c] = wave_sum(q1[c]);
        q2[c] = wave_sum(q2[c]);
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) { s_red[wave][c] = q1[c]; s_red[wave][CT + c] = q2[c]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * CT) {
        const int c = threadIdx.x % CT, which = threadIdx.x / CT;
        if (c0 + c < a.c) {
            const float v = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
            a.partial[((size_t)blockIdx.x * 2 + which) * a.c + c0 + c] = v;
        }
    }
    if (MODE == 2) {                        // the second unit's row: the same wave / slice order
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            r1[MODE == 2 ? c : 0] = wave_sum(r1[MODE == 2 ? c : 0]);
            r2[MODE == 2 ? c : 0] = wave_sum(r2[MODE == 2 ? c : 0]);
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < CT; ++c) { s_red[wave][c] = r1[MODE == 2 ? c : 0]; s_red[wave][CT + c] = r2[MODE == 2 ? c : 0]; }
        }
        __syncthreads();
        if (threadIdx.x < 2 * CT) {
            const int c = threadIdx.x % CT, which = threadIdx.x / CT;
            if (c0 + c < a.c) {
                const float v = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
                a.partial2[((size_t)blockIdx.x * 2 + which) * a.c + c0 + c] = v;
            }
        }
    }
}

int chan_reduce_chunks(int npos) {      // upper bound of the partial rows (workspace sizing)
#ifndef TCR_CHAN_REDUCE_POS
#define TCR_CHAN_REDUCE_POS 512
#endif
    int n = ceil_div(npos, TCR_CHAN_REDUCE_POS);        // >= 2 positions per thread; many short workgroups hide the load latency
    if (n > 512) n = 512;               // (round 6, re-measured with the faster apply pass: 128 / 256 / 384 rows are within noise of 512)
    if (n < 1) n = 1;
    return n;
}

// positions per workgroup: whole utterances where an utterance fits (the 16-byte kernel below needs that)
static int chan_reduce_ppb(int npos, int t) {
    int ppb = ceil_div(npos, chan_reduce_chunks(npos));
    if (t > 0 && t <= ppb && npos % t == 0) ppb = ceil_div(ppb, t) * t;
    return ppb;
}

// number of partial rows launch_chan_reduce() writes for npos positions of t-frame utterances (grid.x)
int chan_reduce_launch_chunks(int npos, int t) { return ceil_div(npos, chan_reduce_ppb(npos, t)); }

// Same sums with 16-byte loads: a workgroup owns 8 channels x a run of whole utterances; the 8 channel rows of an utterance are
// one contiguous block of 8 * Tp floats and thread f always takes floats 4 f .. 4 f + 3 of it, so the channel / frame of each of
// its four elements is fixed: four accumulator pairs per thread, no per-element index arithmetic in the loop.  Afterwards the
// threads of a channel (a contiguous range of f) are added in thread order, element order 0..3 (fixed: bitwise reproducible).
template <int MODE>
__global__ __launch_bounds__(512) void chan_reduce4_kernel(const ChanReduceArgs a) {
    constexpr int CT = 8;
    __shared__ float s_q[2][4][512];
    const int c0 = blockIdx.y * CT, gw = min(CT, a.c - c0);
    const int nt = blockDim.x;
    const int e0 = threadIdx.x * 4, e_per = gw * a.tp;
    const bool active = e0 < e_per;
    int ce[4];
    bool ok[4];
    float mu[4], is[4], ssc[4], ssh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = e0 + q;
        int c = e / a.tp;
        const int tt = e - c * a.tp - kHalo;
        ok[q] = active && e < e_per && tt >= 0 && tt < a.t;
        c = min(c, gw - 1);
        ce[q] = c;
        mu[q] = MODE == 1 ? a.mean[c0 + c] : 0.f;
        is[q] = MODE == 1 ? a.invstd[c0 + c] : 0.f;
        ssc[q] = (MODE == 1 && a.self_scale) ? a.self_scale[c0 + c] : 0.f;
        ssh[q] = (MODE == 1 && a.self_scale) ? a.self_shift[c0 + c] : 1.f;
    }
    float q1[4] = {0.f, 0.f, 0.f, 0.f}, q2[4] = {0.f, 0.f, 0.f, 0.f};
    const int upb = a.pos_per_block / a.t;                      // utterances per workgroup
    const int n0 = blockIdx.x * upb, n1 = min(n0 + upb, a.npos / a.t);
    if (active) {
        const size_t ustride = (size_t)a.c * a.tp;
        const size_t base = (size_t)c0 * a.tp + e0;
        for (int n = n0; n < n1; ++n) {
            const size_t i0 = n * ustride + base;
            const bn_f4 y4 = *reinterpret_cast<const bn_f4*>(a.y + i0);
            if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float yv = ok[q] ? y4[q] : 0.f;
                    q1[q] += yv;
                    q2[q] = fmaf(yv, yv, q2[q]);
                }
            } else {
                bn_f4 d4 = (bn_f4){0.f, 0.f, 0.f, 0.f}, m14 = (bn_f4){1.f, 1.f, 1.f, 1.f}, m24 = m14;
                if (!a.bcast) d4 = *reinterpret_cast<const bn_f4*>(a.da + i0);
                if (a.m1) m14 = *reinterpret_cast<const bn_f4*>(a.m1 + i0);
                if (a.m2) m24 = *reinterpret_cast<const bn_f4*>(a.m2 + i0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float yv = ok[q] ? y4[q] : 0.f;       // (halos of raw train-mode tensors are never written: may hold anything)
                    float dz = a.bcast ? a.da[(size_t)n * a.c + c0 + ce[q]] : d4[q];
                    if (!ok[q]) dz = 0.f;
                    if (a.m1 && !(m14[q] > 0.f)) dz = 0.f;
                    if (a.m2 && !(m24[q] > 0.f)) dz = 0.f;
                    if (a.self_scale && !(fmaf(yv, ssc[q], ssh[q]) > 0.f)) dz = 0.f;
                    q1[q] += dz;
                    q2[q] = fmaf(dz, (yv - mu[q]) * is[q], q2[q]);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { s_q[0][q][threadIdx.x] = q1[q]; s_q[1][q][threadIdx.x] = q2[q]; }
    __syncthreads();
    if (threadIdx.x < 2 * CT) {
        const int c = threadIdx.x % CT, which = threadIdx.x / CT;
        if (c < gw) {
            // elements of channel c: floats [c Tp, (c + 1) Tp) of the block = threads c Tp / 4 .. ((c + 1) Tp - 1) / 4
            const int f0 = (c * a.tp) >> 2, f1 = min(((c + 1) * a.tp - 1) >> 2, nt - 1);
            float v = 0.f;
            for (int f = f0; f <= f1; ++f)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = 4 * f + q;
                    if (e >= c * a.tp && e < (c + 1) * a.tp) v += s_q[which][q][f];
                }
            a.partial[((size_t)blockIdx.x * 2 + which) * a.c + c0 + c] = v;
        }
    }
}

int launch_chan_reduce(int mode, ChanReduceArgs a, int* nchunk_out, hipStream_t s) {
    a.pos_per_block = chan_reduce_ppb(a.npos, a.t);
    const dim3 grid(ceil_div(a.npos, a.pos_per_block), ceil_div(a.c, 8));
    *nchunk_out = (int)grid.x;
    // 16-byte kernel: whole utterances per workgroup, every 8-channel block of an utterance 16-byte aligned and a multiple of 4 floats
    const int nt4 = ceil_div(ceil_div(8 * a.tp, 4), 64) * 64;
    // (its workgroups are short -- (8 Tp / 4) threads walking a few utterances -- so it needs many of them: DS-CNN-L, 276 channels:
    //  -3 % on the step; TC-ResNet's 16-48 channels leave it ~1500 waves for 256 CUs: +8 % on the TCResNet8 step -> scalar kernel there)
    const bool vec = mode != 2 && !a.g_out && a.pos_per_block % a.t == 0 && nt4 <= 512 && ((int64_t)grid.x * grid.y * (nt4 / 64) >= 16 * 256 || tune_get(TCR_TUNE_BWD_MASK) == 4) && (8 * a.tp) % 4 == 0 && (a.c * a.tp) % 4 == 0 &&
                     ((a.c % 8) * a.tp) % 4 == 0 && bn_vec4_ok(a.y, (mode == 1 && !a.bcast) ? a.da : nullptr, a.m1, a.m2, a.c * a.tp, true);
    if (vec) {
        if (mode == 0) hipLaunchKernelGGL((chan_reduce4_kernel<0>), grid, dim3(nt4), 0, s, a);
        else hipLaunchKernelGGL((chan_reduce4_kernel<1>), grid, dim3(nt4), 0, s, a);
        return check_launch("chan_reduce4_kernel");
    }
    if (mode == 0) hipLaunchKernelGGL((chan_reduce_kernel<0>), grid, dim3(256), 0, s, a);
    else if (mode == 2) hipLaunchKernelGGL((chan_reduce_kernel<2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((chan_reduce_kernel<1>), grid, dim3(256), 0, s, a);
    return check_launch("chan_reduce_kernel");
}

// Train-mode forward finalize from sums[2][C] over `count` elements.
// A workgroup owns kBnCB consecutive channels.  Sum of the per-workgroup partial rows (double accumulation, FIXED
// order: chunk k goes to slice k % nparts, the slices are then added in order), spread over the whole workgroup; or
// the pre-reduced sums (sync BN).  Returns through s_out[which * cb + (ch - c0)]; ends with a barrier.
// channels per finalize workgroup: 32 columns x 16 slices of the partial rows.  (Round 4: 16 instead of 32 -- twice the workgroups, a thread
// adds 32 rows of <= 512 in two trips of sixteen loads instead of four: TCResNet8 step 889 -> 866 us, TCResNet14-1.5 2737 -> 2730; 8: 867 /
// 2744.  The slicing defines the summation order of every kernel that reduces partial rows: they all take it from here.)
#ifndef TCR_BN_CB
#define TCR_BN_CB 16
#endif
constexpr int kBnCB = TCR_BN_CB;
static_assert(kBnCB <= 64, "the finalize kernels give a channel of the block to one thread");
// Conv epilogues (EpiSums) leave one row per workgroup / per utterance -- thousands, not <= 512: four channels per workgroup
// there (8 columns = one 32-byte sector of a row x 64 slices, eight times the workgroups), so that a thread adds ~65 rows, sixteen
// loads in flight: 17 us per finalize over 4160 rows.  (Measured: the 32-channel geometry 47 us, in the backward's dependency chain;
// 2 channels x 128 slices as fast but every sector fetched four times; 8 channels x 64 slices in 1024-thread workgroups 113 us -- a
// 16-wave workgroup waits for a CU that the side stream's long filter-gradient workgroups have drained.)  The slice order is a
// function of the row count only, the same in every kernel that reduces the rows (finalize, chan_sums).
int bn_finalize_cb(int nchunk) { return nchunk > 1024 ? 4 : kBnCB; }
static int bn_finalize_threads(int) { return 512; }

__device__ __forceinline__ void reduce_partials(const float* partial, int nchunk, const double* sums, int nc, int c0, int cb,
                                                double* s_slices, double* s_out) {
    const int n2 = 2 * cb;
    if (nchunk > 0) {
        const int nparts = (int)blockDim.x / n2;
        const int part = threadIdx.x / n2, col = threadIdx.x - part * n2;     // col = which * cb + (ch - c0)
        if (part < nparts) {
            const int which = col / cb, ch = c0 + col - which * cb;
            // The rows of a slice are added in order, but their loads are independent: eight in flight per trip (a one-load-
            // per-trip loop pays a full L2 round trip per row: 15 us per finalize at 512 rows).
            double acc = 0.0;
            const float* src = partial + (size_t)which * nc + ch;
            const size_t rstride = (size_t)2 * nc;
            int k = part;
            for (; k + 31 * nparts < nchunk; k += 32 * nparts) {       // (round 6: 512 rows in 16 slices = ONE trip of 32 loads instead of two of 16)
                float v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) v[u] = src[(size_t)(k + u * nparts) * rstride];
#pragma unroll
                for (int u = 0; u < 32; ++u) acc += (double)v[u];
            }
            for (; k + 15 * nparts < nchunk; k += 16 * nparts) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(k + u * nparts) * rstride];
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += (double)v[u];
            }
            for (; k + 7 * nparts < nchunk; k += 8 * nparts) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u * nparts) * rstride];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += (double)v[u];
            }
            for (; k < nchunk; k += nparts) acc += (double)src[(size_t)k * rstride];
            s_slices[part * n2 + col] = acc;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n2; i += blockDim.x) {
            double acc = 0.0;
            for (int p = 0; p < nparts; ++p) acc += s_slices[p * n2 + i];
            s_out[i] = acc;
        }
    } else {
        for (int i = threadIdx.x; i < n2; i += blockDim.x) {
            const int which = i / cb;
            s_out[i] = sums[which * nc + c0 + i - which * cb];
        }
    }
    __syncthreads();
}

// sums[2][C] (double) = the per-workgroup partial rows added up exactly as the finalize kernels do it themselves (same
// channel blocks, same slice order), so that "reduce, hand the sums to the host, finalize" is bitwise the unstaged path.
// The host all-reduces these doubles across replicas (sync BN) before the finalize kernels.
__global__ __launch_bounds__(512) void chan_sums_kernel(const float* __restrict__ partial, int nchunk, int c, double* __restrict__ sums, int cbw) {
    __shared__ double s_slices[512];
    __shared__ double s_tot[2 * kBnCB];
    const int c0 = blockIdx.x * cbw, cb = min(cbw, c - c0);
    reduce_partials(partial, nchunk, nullptr, c, c0, cb, s_slices, s_tot);
    for (int i = threadIdx.x; i < 2 * cb; i += blockDim.x) {
        const int which = i / cb;
        sums[which * c + c0 + i - which * cb] = s_tot[i];
    }
}

int launch_chan_sums(const float* partial, int nchunk, int c, double* sums, hipStream_t s) {
    const int cbw = bn_finalize_cb(nchunk);
    hipLaunchKernelGGL(chan_sums_kernel, dim3(ceil_div(c, cbw)), dim3(bn_finalize_threads(nchunk)), 0, s, partial, nchunk, c, sums, cbw);
    return check_launch("chan_sums_kernel");
}

__global__ __launch_bounds__(512) void bn_finalize_kernel(const BnFinalizeArgs a) {
    __shared__ double s_slices[512];
    __shared__ double s_tot[2 * kBnCB];
    const int c0 = blockIdx.x * a.cbw, cb = min(a.cbw, a.c - c0);
    // The channel's inputs are requested BEFORE the reduction (their round trip hides behind the partial rows'), every store comes last:
    // the pointers may alias as far as the compiler knows, and a load behind a store waits for both round trips (round 6: the old order
    // was a chain of four).  One channel per thread (cb <= kBnCB < the workgroup).
    const int i = threadIdx.x, c = c0 + min(i, cb - 1);
    const float gamma_c = a.gamma ? a.gamma[c] : 1.0f, beta_c = a.beta[c], mm_c = a.moving_mean[c], mv_c = a.moving_var[c];
    reduce_partials(a.partial, a.nchunk, a.sums, a.c, c0, cb, s_slices, s_tot);
    if (i < cb) {
        const double s1 = s_tot[i], s2 = s_tot[cb + i];
        const double m = s1 / a.count;
        double var = s2 / a.count - m * m;
        if (var < 0.0) var = 0.0;
        const float meanf = (float)m, varf = (float)var;
        const float inv = 1.0f / sqrtf(varf + a.eps);
        const float sc = a.gamma ? gamma_c * inv : inv;            // DS-CNN: scale=False (ds_cnn.py:104-118)
        a.scale[c] = sc;
        a.shift[c] = fmaf(-meanf, sc, beta_c);
        a.mean[c] = meanf;
        a.invstd[c] = inv;
        // assign_moving_average: v -= (1 - decay) * (v - value); variance is Bessel-corrected
        const float unbiased = (float)(var * (a.count / (a.count > 1.0 ? a.count - 1.0 : 1.0)));
        const float om = 1.0f - a.decay;
        a.moving_mean[c] = mm_c - om * (mm_c - meanf);
        a.moving_var[c] = mv_c - om * (mv_c - unbiased);
    }
}

// two units in one launch (blockIdx.y): a block's shortcut conv and conv_a get their statistics from the same phase kernel
__global__ __launch_bounds__(512) void bn_finalize2_kernel(const BnFinalizeArgs a, const BnFinalizeArgs b) {
    __shared__ double s_slices[512];
    __shared__ double s_tot[2 * kBnCB];
    const BnFinalizeArgs& u = blockIdx.y == 0 ? a : b;
    const int c0 = blockIdx.x * u.cbw, cb = min(u.cbw, u.c - c0);
    if (cb <= 0) return;
    // The channel's inputs are requested BEFORE the reduction (their round trip hides behind the partial rows'), every store comes last:
    // the pointers may alias as far as the compiler knows, and a load behind a store waits for both round trips (round 6: the old order
    // was a chain of four).  One channel per thread (cb <= kBnCB < the workgroup).
    const int i = threadIdx.x, c = c0 + min(i, cb - 1);
    const float gamma_c = u.gamma ? u.gamma[c] : 1.0f, beta_c = u.beta[c], mm_c = u.moving_mean[c], mv_c = u.moving_var[c];
    reduce_partials(u.partial, u.nchunk, u.sums, u.c, c0, cb, s_slices, s_tot);
    if (i < cb) {
        const double s1 = s_tot[i], s2 = s_tot[cb + i];
        const double m = s1 / u.count;
        double var = s2 / u.count - m * m;
        if (var < 0.0) var = 0.0;
        const float meanf = (float)m, varf = (float)var;
        const float inv = 1.0f / sqrtf(varf + u.eps);
        const float sc = u.gamma ? gamma_c * inv : inv;
        u.scale[c] = sc;
        u.shift[c] = fmaf(-meanf, sc, beta_c);
        u.mean[c] = meanf;
        u.invstd[c] = inv;
        const float unbiased = (float)(var * (u.count / (u.count > 1.0 ? u.count - 1.0 : 1.0)));
        const float om = 1.0f - u.decay;
        u.moving_mean[c] = mm_c - om * (mm_c - meanf);
        u.moving_var[c] = mv_c - om * (mv_c - unbiased);
    }
}

int launch_bn_finalize2(const BnFinalizeArgs& a0, const BnFinalizeArgs& b0, hipStream_t s) {
    BnFinalizeArgs a = a0, b = b0;
    a.cbw = bn_finalize_cb(a.nchunk); b.cbw = bn_finalize_cb(b.nchunk);
    const int gx = max(ceil_div(a.c, a.cbw), ceil_div(b.c, b.cbw));
    hipLaunchKernelGGL(bn_finalize2_kernel, dim3(gx, 2), dim3(bn_finalize_threads(a.nchunk)), 0, s, a, b);
    return check_launch("bn_finalize2_kernel");
}

int launch_bn_finalize(const BnFinalizeArgs& a0, hipStream_t s) {
    BnFinalizeArgs a = a0;
    a.cbw = bn_finalize_cb(a.nchunk);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(a.c, a.cbw)), dim3(bn_finalize_threads(a.nchunk)), 0, s, a);
    return check_launch("bn_finalize_kernel");
}

// a = [relu](y * scale + shift)            (res == nullptr)
// a = relu(y * scale + shift + res)        (block output, tc_resnet.py:40-41)
// grid = (ceil(C * Tp / 256), B): one utterance per grid row, so the (channel, frame) split of an index is a 13-bit
// quotient -- a float multiply with a one-step fix-up instead of the
// 64-bit integer division a flat index needs (which made these HBM-streaming kernels instruction-bound).
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApplyArgs a) {
    const int per_utt = a.c * a.tp;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= per_utt) return;
    int c = (int)(((float)j + 0.5f) * a.inv_tp);
    c += (c + 1) * a.tp <= j ? 1 : (c * a.tp > j ? -1 : 0);
    const int tt = j - c * a.tp - kHalo;
    const size_t i = (size_t)blockIdx.y * per_utt + j;
    float v = 0.f;
    if (tt >= 0 && tt < a.t) {
        v = fmaf(a.y[i], a.scale[c], a.shift[c]);
        if (a.res) v = fmaxf(v + a.res[i], 0.f);
        else if (a.relu) v = fmaxf(v, 0.f);
    }
    a.out[i] = v;
}

// Four consecutive elements per thread (one 16-byte load / store per tensor): the elementwise BN passes of the big-activation
// nets (DS-CNN-L: 600 MB tensors) are pure HBM streams.  The four elements may straddle a channel row: channel and frame index
// are per element (same expressions as the scalar kernel -> bitwise the same results).

__global__ __launch_bounds__(256) void bn_apply4_kernel(const BnApplyArgs a) {
    const int per_utt = a.c * a.tp;
    const int j0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (j0 >= per_utt) return;
    const size_t i0 = (size_t)blockIdx.y * per_utt + j0;
    const bn_f4 y4 = *reinterpret_cast<const bn_f4*>(a.y + i0);
    bn_f4 r4 = (bn_f4){0.f, 0.f, 0.f, 0.f};
    if (a.res) r4 = *reinterpret_cast<const bn_f4*>(a.res + i0);
    int c = (int)(((float)j0 + 0.5f) * a.inv_tp);
    c += (c + 1) * a.tp <= j0 ? 1 : (c * a.tp > j0 ? -1 : 0);
    bn_f4 o4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = j0 + e;
        const int ce = c + (j >= (c + 1) * a.tp ? 1 : 0);        // (tp >= 9: at most one row boundary inside the four)
        const int tt = j - ce * a.tp - kHalo;
        float v = 0.f;
        if (tt >= 0 && tt < a.t) {
            v = fmaf(y4[e], a.scale[ce], a.shift[ce]);
            if (a.res) v = fmaxf(v + r4[e], 0.f);
            else if (a.relu) v = fmaxf(v, 0.f);
        }
        o4[e] = v;
    }
    *reinterpret_cast<bn_f4*>(a.out + i0) = o4;
}

int launch_bn_apply(const BnApplyArgs& a0, hipStream_t s) {
    BnApplyArgs a = a0;
    a.inv_tp = 1.0f / (float)a.tp;
    const int per_utt = a.c * a.tp;
    const int batch = (int)(a.total / per_utt);
    if (per_utt >= (1 << 22) || batch > 65535) { set_error("bn_apply: %d x %d exceeds the launch geometry", batch, per_utt); return TCR_ERR_ARG; }
    if (bn_vec4_ok(a.y, a.res, a.out, nullptr, per_utt)) {
        hipLaunchKernelGGL(bn_apply4_kernel, dim3(ceil_div(per_utt / 4, 256), batch), dim3(256), 0, s, a);
        return check_launch("bn_apply4_kernel");
    }
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ceil_div(per_utt, 256), batch), dim3(256), 0, s, a);
    return check_launch("bn_apply_kernel");
}

// Backward finalize: dgamma, dbeta into the gradient arena + the per-channel coefficients of
//   dy = k1 * (dz - k2 - (y - mean) * k3),  k1 = gamma*invstd, k2 = dbeta/n, k3 = invstd*dgamma/n
__global__ __launch_bounds__(512) void bn_bwd_finalize_kernel(const BnBwdFinalizeArgs a) {
    __shared__ double s_slices[512];
    __shared__ double s_tot[2 * kBnCB];
    const int c0 = blockIdx.x * a.cbw, cb = min(a.cbw, a.c - c0);
    // The channel's inputs are requested BEFORE the reduction (their round trip hides behind the partial rows'), every store comes last:
    // the pointers may alias as far as the compiler knows, and a load behind a store waits for both round trips (round 6: the old order
    // -- k1 stored, read back for the table, ... -- was a chain of eight).  One channel per thread (cb <= kBnCB < the workgroup).
    const int i = threadIdx.x, c = c0 + min(i, cb - 1);
    const float invstd_c = a.invstd[c], gamma_c = a.gamma ? a.gamma[c] : 1.0f;
    const float mean_c = a.tab ? a.mean[c] : 0.f;
    const float ssc_c = (a.tab && a.self_scale) ? a.self_scale[c] : 0.f, ssh_c = (a.tab && a.self_scale) ? a.self_shift[c] : 1.f;
    reduce_partials(a.partial, a.nchunk, a.sums, a.c, c0, cb, s_slices, s_tot);
    if (i < cb) {
        const float db = (float)s_tot[i], dg = (float)s_tot[cb + i];
        const float k1 = a.gamma ? gamma_c * invstd_c : invstd_c;
        const float k2 = (float)((double)db / a.count);
        const float k3 = (float)((double)invstd_c * (double)dg / a.count);
        a.dbeta[c] = db * a.grad_scale;
        if (a.dgamma) a.dgamma[c] = dg * a.grad_scale;
        a.k1[c] = k1;
        a.k2[c] = k2;
        a.k3[c] = k3;
        if (a.tab) {
            float* t = a.tab + (size_t)c * 8;
            t[0] = k1; t[1] = k2; t[2] = k3; t[3] = mean_c;
            t[4] = ssc_c; t[5] = ssh_c; t[6] = 0.f; t[7] = 0.f;
        }
    }
}

// two units in one launch (blockIdx.y): a block's conv_b and its shortcut conv get their backward sums from the same kernel
// (the lazy backward's data-gradient epilogue, bwd_lazy.hip), and both must be final before the block's kernels start
__global__ __launch_bounds__(512) void bn_bwd_finalize2_kernel(const BnBwdFinalizeArgs a, const BnBwdFinalizeArgs b) {
    __shared__ double s_slices[512];
    __shared__ double s_tot[2 * kBnCB];
    const BnBwdFinalizeArgs& u = blockIdx.y == 0 ? a : b;
    const int c0 = blockIdx.x * u.cbw, cb = min(u.cbw, u.c - c0);
    if (cb <= 0) return;
    // The channel's inputs are requested BEFORE the reduction (their round trip hides behind the partial rows'), every store comes last:
    // the pointers may alias as far as the compiler knows, and a load behind a store waits for both round trips (round 6: the old order
    // -- k1 stored, read back for the table, ... -- was a chain of eight).  One channel per thread (cb <= kBnCB < the workgroup).
    const int i = threadIdx.x, c = c0 + min(i, cb - 1);
    const float invstd_c = u.invstd[c], gamma_c = u.gamma ? u.gamma[c] : 1.0f;
    const float mean_c = u.tab ? u.mean[c] : 0.f;
    const float ssc_c = (u.tab && u.self_scale) ? u.self_scale[c] : 0.f, ssh_c = (u.tab && u.self_scale) ? u.self_shift[c] : 1.f;
    reduce_partials(u.partial, u.nchunk, u.sums, u.c, c0, cb, s_slices, s_tot);
    if (i < cb) {
        const float db = (float)s_tot[i], dg = (float)s_tot[cb + i];
        const float k1 = u.gamma ? gamma_c * invstd_c : invstd_c;
        const float k2 = (float)((double)db / u.count);
        const float k3 = (float)((double)invstd_c * (double)dg / u.count);
        u.dbeta[c] = db * u.grad_scale;
        if (u.dgamma) u.dgamma[c] = dg * u.grad_scale;
        u.k1[c] = k1;
        u.k2[c] = k2;
        u.k3[c] = k3;
        if (u.tab) {
            float* t = u.tab + (size_t)c * 8;
            t[0] = k1; t[1] = k2; t[2] = k3; t[3] = mean_c;
            t[4] = ssc_c; t[5] = ssh_c; t[6] = 0.f; t[7] = 0.f;
        }
    }
}

int launch_bn_bwd_finalize2(const BnBwdFinalizeArgs& a0, const BnBwdFinalizeArgs& b0, hipStream_t s) {
    BnBwdFinalizeArgs a = a0, b = b0;
    a.cbw = bn_finalize_cb(a.nchunk); b.cbw = bn_finalize_cb(b.nchunk);
    // one launch serves both units with ONE thread count: the slicing of the partial rows (reduce_partials) is a function of the thread
    // count and of cbw, so the units must agree on both; callers whose units differ get two launches (same results, one launch more)
    if (bn_finalize_threads(a.nchunk) != bn_finalize_threads(b.nchunk) || a.cbw != b.cbw) {
        const int rc = launch_bn_bwd_finalize(a0, s);
        return rc != TCR_OK ? rc : launch_bn_bwd_finalize(b0, s);
    }
    const int gx = max(ceil_div(a.c, a.cbw), ceil_div(b.c, b.cbw));
    hipLaunchKernelGGL(bn_bwd_finalize2_kernel, dim3(gx, 2), dim3(bn_finalize_threads(a.nchunk)), 0, s, a, b);
    return check_launch("bn_bwd_finalize2_kernel");
}

int launch_bn_bwd_finalize(const BnBwdFinalizeArgs& a0, hipStream_t s) {
    BnBwdFinalizeArgs a = a0;
    a.cbw = bn_finalize_cb(a.nchunk);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(a.c, a.cbw)), dim3(bn_finalize_threads(a.nchunk)), 0, s, a);
    return check_launch("bn_bwd_finalize_kernel");
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdApplyArgs a) {      // (geometry: see bn_apply_kernel)
    const int per_utt = a.c * a.tp;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= per_utt) return;
    int c = (int)(((float)j + 0.5f) * a.inv_tp);
    c += (c + 1) * a.tp <= j ? 1 : (c * a.tp > j ? -1 : 0);
    const int tt = j - c * a.tp - kHalo;
    const size_t i = (size_t)blockIdx.y * per_utt + j;
    float v = 0.f;
    if (tt >= 0 && tt < a.t) {
        float dz = a.bcast ? a.da[(size_t)blockIdx.y * a.c + c] : a.da[i];
        if (a.m1 && !(a.m1[i] > 0.f)) dz = 0.f;
        if (a.m2 && !(a.m2[i] > 0.f)) dz = 0.f;
        const float yv = a.y[i];
        if (a.self_scale && !(fmaf(yv, a.self_scale[c], a.self_shift[c]) > 0.f)) dz = 0.f;
        v = a.k1[c] * (dz - a.k2[c] - (yv - a.mean[c]) * a.k3[c]);
        if (a.accumulate) v += a.dy[i];
    } else if (a.accumulate) {
        return;
    }
    a.dy[i] = v;
}

__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const BnBwdApplyArgs a) {        // (see bn_apply4_kernel)
    const int per_utt = a.c * a.tp;
    const int j0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (j0 >= per_utt) return;
    const size_t i0 = (size_t)blockIdx.y * per_utt + j0;
    const bn_f4 y4 = *reinterpret_cast<const bn_f4*>(a.y + i0);
    bn_f4 d4 = (bn_f4){0.f, 0.f, 0.f, 0.f}, m14 = (bn_f4){1.f, 1.f, 1.f, 1.f}, m24 = m14;
    if (!a.bcast) d4 = *reinterpret_cast<const bn_f4*>(a.da + i0);
    if (a.m1) m14 = *reinterpret_cast<const bn_f4*>(a.m1 + i0);
    if (a.m2) m24 = *reinterpret_cast<const bn_f4*>(a.m2 + i0);
    int c = (int)(((float)j0 + 0.5f) * a.inv_tp);
    c += (c + 1) * a.tp <= j0 ? 1 : (c * a.tp > j0 ? -1 : 0);
    bn_f4 o4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = j0 + e;
        const int ce = c + (j >= (c + 1) * a.tp ? 1 : 0);
        const int tt = j - ce * a.tp - kHalo;
        float v = 0.f;
        if (tt >= 0 && tt < a.t) {
            float dz = a.bcast ? a.da[(size_t)blockIdx.y * a.c + ce] : d4[e];
            if (a.m1 && !(m14[e] > 0.f)) dz = 0.f;
            if (a.m2 && !(m24[e] > 0.f)) dz = 0.f;
            const float yv = y4[e];
            if (a.self_scale && !(fmaf(yv, a.self_scale[ce], a.self_shift[ce]) > 0.f)) dz = 0.f;
            v = a.k1[ce] * (dz - a.k2[ce] - (yv - a.mean[ce]) * a.k3[ce]);
        }
        o4[e] = v;
    }
    *reinterpret_cast<bn_f4*>(a.dy + i0) = o4;
}

// The same pass for the large tensors (DS-CNN-L: 4096 x 276 x 73 floats, ~1 GB through HBM per launch): kApplyU float4 per thread
// and operand, all loads of a thread issued before anything waits on them (the one-float4 kernel above keeps 32 B per lane in flight
// and ran at 2.8 TB/s; a float4 copy measures 6.3 TB/s on this chip); the six per-channel coefficients of the workgroup's channel
// range are staged in LDS once (8 floats per channel) instead of 24 gathers per float4.  A float4 touches at most two channels (tp >= 4).
// Same expression per element as bn_bwd_apply_kernel: bitwise the same dy.
constexpr int kApplyU = 4;
constexpr int kApplyMaxCh = 264;                // channels a workgroup's 1024 * kApplyU floats can touch at tp >= 16: 4096 / 16 + 2

__global__ __launch_bounds__(256) void bn_bwd_apply4x_kernel(const BnBwdApplyArgs a) {
    __shared__ __attribute__((aligned(16))) float s_k[kApplyMaxCh * 8];
    const int per_utt = a.c * a.tp;
    const int base4 = blockIdx.x * (256 * kApplyU);
    const size_t u0 = (size_t)blockIdx.y * per_utt;
    bn_f4 y4[kApplyU], d4[kApplyU];
    int j0[kApplyU];
#pragma unroll
    for (int u = 0; u < kApplyU; ++u) {         // (past the utterance's end: the last float4 again, not stored)
        j0[u] = min((base4 + u * 256 + (int)threadIdx.x) * 4, per_utt - 4);
        y4[u] = *reinterpret_cast<const bn_f4*>(a.y + u0 + j0[u]);
        d4[u] = (bn_f4){0.f, 0.f, 0.f, 0.f};
        if (!a.bcast) d4[u] = *reinterpret_cast<const bn_f4*>(a.da + u0 + j0[u]);
    }
    const int c_lo = fast_div(min(base4 * 4, per_utt - 4), a.tp, a.inv_tp);
    const int c_hi = min(fast_div(min((base4 + 256 * kApplyU) * 4, per_utt) - 1, a.tp, a.inv_tp) + 1, a.c - 1);
    const bool has_self = a.self_scale != nullptr;
    for (int i = threadIdx.x; i < (c_hi - c_lo + 1) * 8; i += 256) {
        const int ch = c_lo + (i >> 3), k = i & 7;
        float v = 0.f;
        if (k == 0) v = a.k1[ch];
        else if (k == 1) v = a.k2[ch];
        else if (k == 2) v = a.k3[ch];
        else if (k == 3) v = a.mean[ch];
        else if (k == 4) v = has_self ? a.self_scale[ch] : 0.f;
        else if (k == 5) v = has_self ? a.self_shift[ch] : 1.f;
        s_k[i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kApplyU; ++u) {
        if ((base4 + u * 256 + (int)threadIdx.x) * 4 >= per_utt) break;
        const int c = fast_div(j0[u], a.tp, a.inv_tp);
        const int cn = min(c + 1, a.c - 1);
        const bn_f4 ka = *reinterpret_cast<const bn_f4*>(s_k + (c - c_lo) * 8), kb = *reinterpret_cast<const bn_f4*>(s_k + (c - c_lo) * 8 + 4);
        const bn_f4 na = *reinterpret_cast<const bn_f4*>(s_k + (cn - c_lo) * 8), nb = *reinterpret_cast<const bn_f4*>(s_k + (cn - c_lo) * 8 + 4);
        bn_f4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = j0[u] + e;
            const bool nx = j >= (c + 1) * a.tp;
            const int ce = nx ? c + 1 : c;
            const int tt = j - ce * a.tp - kHalo;
            const float k1 = nx ? na[0] : ka[0], k2 = nx ? na[1] : ka[1], k3 = nx ? na[2] : ka[2], mu = nx ? na[3] : ka[3];
            const float ssc = nx ? nb[0] : kb[0], ssh = nx ? nb[1] : kb[1];
            float v = 0.f;
            if (tt >= 0 && tt < a.t) {
                float dz = a.bcast ? a.da[(size_t)blockIdx.y * a.c + ce] : d4[u][e];
                const float yv = y4[u][e];
                if (has_self && !(fmaf(yv, ssc, ssh) > 0.f)) dz = 0.f;
                v = k1 * (dz - k2 - (yv - mu) * k3);
            }
            o4[e] = v;
        }
        *reinterpret_cast<bn_f4*>(a.dy + u0 + j0[u]) = o4;
    }
}

int launch_bn_bwd_apply(const BnBwdApplyArgs& a0, hipStream_t s) {
    BnBwdApplyArgs a = a0;
    a.inv_tp = 1.0f / (float)a.tp;
    const int per_utt = a.c * a.tp;
    const int batch = (int)(a.total / per_utt);
    if (per_utt >= (1 << 22) || batch > 65535) { set_error("bn_bwd_apply: %d x %d exceeds the launch geometry", batch, per_utt); return TCR_ERR_ARG; }
    // the wide kernel where an utterance fills at least two of its workgroups (DS-CNN: 20148 floats; TC-ResNet's 16-72 channels x 57 do not)
    if (!a.accumulate && !a.m1 && !a.m2 && a.tp >= 16 && per_utt >= 2 * 1024 * kApplyU && tune_get(TCR_TUNE_BN_APPLY) != 1 &&
        bn_vec4_ok(a.y, a.bcast ? nullptr : a.da, nullptr, nullptr, per_utt) && (reinterpret_cast<uintptr_t>(a.dy) & 15) == 0) {
        hipLaunchKernelGGL(bn_bwd_apply4x_kernel, dim3(ceil_div(per_utt / 4, 256 * kApplyU), batch), dim3(256), 0, s, a);
        return check_launch("bn_bwd_apply4x_kernel");
    }
    if (!a.accumulate && bn_vec4_ok(a.y, a.bcast ? nullptr : a.da, a.m1, a.m2, per_utt) && (reinterpret_cast<uintptr_t>(a.dy) & 15) == 0) {
        hipLaunchKernelGGL(bn_bwd_apply4_kernel, dim3(ceil_div(per_utt / 4, 256), batch), dim3(256), 0, s, a);
        return check_launch("bn_bwd_apply4_kernel");
    }
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ceil_div(per_utt, 256), batch), dim3(256), 0, s, a);
    return check_launch("bn_bwd_apply_kernel");
}

// BN backward, finalize folded into the apply pass (one launch instead of two in the backward's dependency chain): a workgroup
// owns (kFG consecutive channels, a slab of utterances); it first adds up ITS channels' partial rows -- same channel blocks, same
// slice order as reduce_partials(), so dgamma / dbeta / k1..k3 are bitwise those of bn_bwd_finalize_kernel -- and then streams
// its [slab][kFG][Tp] block (contiguous per utterance).  Slab 0 of every channel group writes dgamma / dbeta.
constexpr int kFG = 8;                  // channels per workgroup (divides kBnCB: a group never straddles a finalize block)
constexpr int kFusedMaxParts = 64;
constexpr int kFusedU = 4;              // float4 per thread, operand and trip of the wide apply loop

__global__ __launch_bounds__(256) void bn_bwd_apply_fused_kernel(const BnBwdFinalizeArgs f, const BnBwdApplyArgs a, int utt_per_slab, int batch, int vec4) {
    __shared__ double s_slices[kFusedMaxParts * 2 * kFG];
    __shared__ float s_k[3][kFG];
    const int g0 = blockIdx.y * kFG, gw = min(kFG, a.c - g0);
    {
        const int c0 = g0 / kBnCB * kBnCB, cb = min(kBnCB, a.c - c0);
        const int nparts = 512 / (2 * cb);                      // reduce_partials' slicing of this channel block (512-thread finalize)
        const int col = threadIdx.x & (2 * kFG - 1);            // which * kFG + channel
        const int which = col / kFG, ch = g0 + col % kFG;
        // (the coefficients' inputs are requested before the reduction, the stores of slab 0 come last: see bn_bwd_finalize_kernel)
        const int cpre = min(g0 + (int)threadIdx.x, a.c - 1);
        const float gamma_c = f.gamma ? f.gamma[cpre] : 1.0f, invstd_c = f.invstd[cpre];
        if (ch < a.c) {
            const float* src = f.partial + (size_t)which * a.c + ch;
            const size_t rstride = (size_t)2 * a.c;
            for (int part = threadIdx.x / (2 * kFG); part < nparts; part += 256 / (2 * kFG)) {
                double acc = 0.0;
                int k = part;
                for (; k + 23 * nparts < f.nchunk; k += 24 * nparts) {      // (round 6: 384 rows in 16 slices = one trip instead of three)
                    float v[24];
#pragma unroll
                    for (int u = 0; u < 24; ++u) v[u] = src[(size_t)(k + u * nparts) * rstride];
#pragma unroll
                    for (int u = 0; u < 24; ++u) acc += (double)v[u];
                }
                for (; k + 7 * nparts < f.nchunk; k += 8 * nparts) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u * nparts) * rstride];
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += (double)v[u];
                }
                for (; k < f.nchunk; k += nparts) acc += (double)src[(size_t)k * rstride];
                s_slices[part * 2 * kFG + col] = acc;
            }
        }
        __syncthreads();
        if (threadIdx.x < gw) {
            const int c = g0 + threadIdx.x;
            double db = 0.0, dg = 0.0;
            for (int p = 0; p < nparts; ++p) { db += s_slices[p * 2 * kFG + threadIdx.x]; dg += s_slices[p * 2 * kFG + kFG + threadIdx.x]; }
            const float dbf = (float)db, dgf = (float)dg;
            const float k1 = f.gamma ? gamma_c * invstd_c : invstd_c;
            const float k2 = (float)((double)dbf / f.count);
            const float k3 = (float)((double)invstd_c * (double)dgf / f.count);
            s_k[0][threadIdx.x] = k1;
            s_k[1][threadIdx.x] = k2;
            s_k[2][threadIdx.x] = k3;
            if (blockIdx.x == 0) {
                f.dbeta[c] = dbf * f.grad_scale;
                if (f.dgamma) f.dgamma[c] = dgf * f.grad_scale;
                f.k1[c] = k1;
                f.k2[c] = k2;
                f.k3[c] = k3;
            }
        }
        __syncthreads();
    }
    const int n0 = blockIdx.x * utt_per_slab, n1 = min(n0 + utt_per_slab, batch);
    const int e_per = gw * a.tp;                                // this group's floats of one utterance (contiguous)
    const float inv_e = 1.0f / (float)e_per;
    const int total = (n1 - n0) * e_per;
    if (vec4 == 2) {
        // Round 6: kFusedU float4 per thread, operand and trip, every load of a trip requested before the first is used, and the
        // group's per-channel coefficients (k1, k2, k3, mean, own-mask scale / shift) read from an LDS table as two 16-byte rows per
        // float4 -- the one-float4 loop below keeps 32 B per lane in flight and gathers three coefficients per ELEMENT from global
        // memory (the apply passes ran at ~3 TB/s where a float4 copy measures 6.3).  Same expression per element: bitwise the same dy.
        constexpr int U = kFusedU;
        __shared__ __attribute__((aligned(16))) float s_tab[(kFG + 1) * 8];
        for (int i = threadIdx.x; i < (kFG + 1) * 8; i += 256) {
            const int cl = min(i >> 3, gw - 1), k = i & 7;
            float v = 0.f;
            if (k < 3) v = s_k[k][cl];
            else if (k == 3) v = a.mean[g0 + cl];
            else if (k == 4) v = a.self_scale ? a.self_scale[g0 + cl] : 0.f;         // (no own mask: fmaf(y, 0, 1) > 0 always)
            else if (k == 5) v = a.self_scale ? a.self_shift[g0 + cl] : 1.f;
            s_tab[i] = v;
        }
        __syncthreads();
        const bool has_m1 = a.m1 != nullptr, has_m2 = a.m2 != nullptr;
        for (int base = threadIdx.x * 4; base < total; base += 1024 * U) {
            bn_f4 y4[U], d4[U], m14[U], m24[U];
            size_t i0[U];
            int e0[U], c0[U], nn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = min(base + u * 1024, total - 4);        // (past the slab's end: its last float4 again, not stored)
                int nl = (int)(((float)idx + 0.5f) * inv_e);
                nl += (nl + 1) * e_per <= idx ? 1 : (nl * e_per > idx ? -1 : 0);
                e0[u] = idx - nl * e_per;                               // (e_per % 4 == 0: the four stay inside one utterance)
                int cc = (int)(((float)e0[u] + 0.5f) * a.inv_tp);
                cc += (cc + 1) * a.tp <= e0[u] ? 1 : (cc * a.tp > e0[u] ? -1 : 0);
                c0[u] = cc;
                nn[u] = n0 + nl;
                i0[u] = ((size_t)nn[u] * a.c + g0) * a.tp + e0[u];
                y4[u] = *reinterpret_cast<const bn_f4*>(a.y + i0[u]);
                d4[u] = (bn_f4){0.f, 0.f, 0.f, 0.f};
                m14[u] = (bn_f4){1.f, 1.f, 1.f, 1.f};
                m24[u] = m14[u];
                if (!a.bcast) d4[u] = *reinterpret_cast<const bn_f4*>(a.da + i0[u]);
                if (has_m1) m14[u] = *reinterpret_cast<const bn_f4*>(a.m1 + i0[u]);
                if (has_m2) m24[u] = *reinterpret_cast<const bn_f4*>(a.m2 + i0[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (base + u * 1024 >= total) break;
                const bn_f4 ka = *reinterpret_cast<const bn_f4*>(s_tab + c0[u] * 8), kb = *reinterpret_cast<const bn_f4*>(s_tab + c0[u] * 8 + 4);
                const bn_f4 na = *reinterpret_cast<const bn_f4*>(s_tab + (c0[u] + 1) * 8), nb = *reinterpret_cast<const bn_f4*>(s_tab + (c0[u] + 1) * 8 + 4);
                bn_f4 o4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = e0[u] + q;
                    const bool nx = e >= (c0[u] + 1) * a.tp;
                    const int cl = nx ? c0[u] + 1 : c0[u];
                    const int tt = e - cl * a.tp - kHalo;
                    const float k1 = nx ? na[0] : ka[0], k2 = nx ? na[1] : ka[1], k3 = nx ? na[2] : ka[2], mu = nx ? na[3] : ka[3];
                    const float ssc = nx ? nb[0] : kb[0], ssh = nx ? nb[1] : kb[1];
                    float v = 0.f;
                    if (tt >= 0 && tt < a.t) {
                        float dz = a.bcast ? a.da[(size_t)nn[u] * a.c + g0 + cl] : d4[u][q];
                        if (has_m1 && !(m14[u][q] > 0.f)) dz = 0.f;
                        if (has_m2 && !(m24[u][q] > 0.f)) dz = 0.f;
                        const float yv = y4[u][q];
                        if (a.self_scale && !(fmaf(yv, ssc, ssh) > 0.f)) dz = 0.f;
                        v = k1 * (dz - k2 - (yv - mu) * k3);
                    }
                    o4[q] = v;
                }
                *reinterpret_cast<bn_f4*>(a.dy + i0[u]) = o4;
            }
        }
        return;
    }
    if (vec4) {                                                 // four consecutive elements per thread and trip (see bn_apply4_kernel)
        for (int idx = threadIdx.x * 4; idx < total; idx += 1024) {
            int nl = (int)(((float)idx + 0.5f) * inv_e);
            nl += (nl + 1) * e_per <= idx ? 1 : (nl * e_per > idx ? -1 : 0);
            const int e0 = idx - nl * e_per;                    // (e_per % 4 == 0: the four stay inside one utterance)
            int c0 = (int)(((float)e0 + 0.5f) * a.inv_tp);
            c0 += (c0 + 1) * a.tp <= e0 ? 1 : (c0 * a.tp > e0 ? -1 : 0);
            const int n = n0 + nl;
            const size_t i0 = ((size_t)n * a.c + g0) * a.tp + e0;
            const bn_f4 y4 = *reinterpret_cast<const bn_f4*>(a.y + i0);
            bn_f4 d4 = (bn_f4){0.f, 0.f, 0.f, 0.f}, m14 = (bn_f4){1.f, 1.f, 1.f, 1.f}, m24 = m14;
            if (!a.bcast) d4 = *reinterpret_cast<const bn_f4*>(a.da + i0);
            if (a.m1) m14 = *reinterpret_cast<const bn_f4*>(a.m1 + i0);
            if (a.m2) m24 = *reinterpret_cast<const bn_f4*>(a.m2 + i0);
            bn_f4 o4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + q;
                const int cl = c0 + (e >= (c0 + 1) * a.tp ? 1 : 0);
                const int tt = e - cl * a.tp - kHalo;
                float v = 0.f;
                if (tt >= 0 && tt < a.t) {
                    float dz = a.bcast ? a.da[(size_t)n * a.c + g0 + cl] : d4[q];
                    if (a.m1 && !(m14[q] > 0.f)) dz = 0.f;
                    if (a.m2 && !(m24[q] > 0.f)) dz = 0.f;
                    const float yv = y4[q];
                    if (a.self_scale && !(fmaf(yv, a.self_scale[g0 + cl], a.self_shift[g0 + cl]) > 0.f)) dz = 0.f;
                    v = s_k[0][cl] * (dz - s_k[1][cl] - (yv - a.mean[g0 + cl]) * s_k[2][cl]);
                }
                o4[q] = v;
            }
            *reinterpret_cast<bn_f4*>(a.dy + i0) = o4;
        }
        return;
    }
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        int nl = (int)(((float)idx + 0.5f) * inv_e);
        nl += (nl + 1) * e_per <= idx ? 1 : (nl * e_per > idx ? -1 : 0);
        const int e = idx - nl * e_per;
        int cl = (int)(((float)e + 0.5f) * a.inv_tp);
        cl += (cl + 1) * a.tp <= e ? 1 : (cl * a.tp > e ? -1 : 0);
        const int tt = e - cl * a.tp - kHalo;
        const int n = n0 + nl;
        const size_t i = ((size_t)n * a.c + g0) * a.tp + e;
        float v = 0.f;
        if (tt >= 0 && tt < a.t) {
            float dz = a.bcast ? a.da[(size_t)n * a.c + g0 + cl] : a.da[i];
            if (a.m1 && !(a.m1[i] > 0.f)) dz = 0.f;
            if (a.m2 && !(a.m2[i] > 0.f)) dz = 0.f;
            const float yv = a.y[i];
            if (a.self_scale && !(fmaf(yv, a.self_scale[g0 + cl], a.self_shift[g0 + cl]) > 0.f)) dz = 0.f;
            v = s_k[0][cl] * (dz - s_k[1][cl] - (yv - a.mean[g0 + cl]) * s_k[2][cl]);
        }
        a.dy[i] = v;
    }
}

// returns 1 (nothing launched) where the pair of kernels is needed: pre-reduced sums (sync BN), accumulate mode, odd channel blocks
int launch_bn_bwd_apply_fused(const BnBwdFinalizeArgs& f, const BnBwdApplyArgs& a0, hipStream_t s) {
    if (f.nchunk <= 0 || a0.accumulate || tune_get(TCR_TUNE_BWD_BN_FUSED) == 1) return 1;
    // (rounds 3-5: folded only for layers of <= 48 channels -- with the one-float4 apply loop the 72-channel layers of TCResNet14-1.5
    //  measured +1.5 % per step folded.  Round 6, wide apply loop: folded everywhere and ~512 workgroups instead of 1024 --
    //  TCResNet14-1.5 step 2519 -> 2480 us at 49 frames (384 / 512 / 768 workgroups alike, 256: 2512), 4096 -> 4048 at 98.)
    for (int c0 = 0; c0 < a0.c; c0 += kBnCB)
        if (512 / (2 * min(kBnCB, a0.c - c0)) > kFusedMaxParts) return 1;
    BnBwdApplyArgs a = a0;
    a.inv_tp = 1.0f / (float)a.tp;
    const int per_utt = a.c * a.tp;
    const int batch = (int)(a.total / per_utt);
    if (kFG * a.tp >= (1 << 20) || batch >= (1 << 20)) return 1;
    const int groups = ceil_div(a.c, kFG);
    int want = tune_get(TCR_TUNE_BWD_BN_FUSED) >= 2 ? tune_get(TCR_TUNE_BWD_BN_FUSED) : 512;        // workgroups aimed at
    int nslab = max(1, min(batch, want / groups));
    const int ups = ceil_div(batch, nslab);
    nslab = ceil_div(batch, ups);
    // 16-byte accesses: every channel group's block of an utterance is a multiple of four floats and starts on one
    const int vec4 = ((kFG * a.tp) % 4 == 0 && (a.c * a.tp) % 4 == 0 && ((a.c % kFG) * a.tp) % 4 == 0 &&
                      bn_vec4_ok(a.y, a.bcast ? nullptr : a.da, a.m1, a.m2, per_utt) && (reinterpret_cast<uintptr_t>(a.dy) & 15) == 0) ? 1 : 0;
    // (2: the wide loop -- TCR_TUNE_BN_APPLY = 1 keeps the one-float4 loop; needs a slab of at least one float4)
    const int vmode = (vec4 && tune_get(TCR_TUNE_BN_APPLY) != 1 && ups * kFG * a.tp >= 4) ? 2 : vec4;
    hipLaunchKernelGGL(bn_bwd_apply_fused_kernel, dim3(nslab, groups), dim3(256), 0, s, f, a, ups, batch, vmode);
    return check_launch("bn_bwd_apply_fused_kernel");
}

}  // namespace tcr
