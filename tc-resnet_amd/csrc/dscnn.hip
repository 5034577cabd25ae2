// DS-CNN (S / M / L) for gfx950: the depthwise-separable baseline of BASELINE.json configs[4] -- eval-mode forward,
// train-mode forward and backward.
//
// Replaces DSCNN() + DSCNN_arg_scope (audio_nets/ds_cnn.py:19-118): conv_1 (10x4, +bias) -> BN(no scale)+ReLU,
// N x [depthwise 3x3 (+bias) -> BN+ReLU -> pointwise 1x1 (+bias) -> BN+ReLU], global average pool,
// fully_connected (+bias), softmax (factory/audio_nets.py:154).  Input: the [B, 49, 10, 1] MFCC map, taken
// straight from the front-end's planar buffer feat[b][f][HALO + t] (H = time, W = coefficient).
//
// Kernels: conv_1 is an implicit GEMM on the exact-f32 MFMA (K = 10 kernel rows x 4 kernel columns: one MFMA
// K-step per kernel row); the pointwise convs (88 % of the MACs, 276 x 276 per position) and the classifier reuse
// the matrix-core kernels of mfma.hip / head.hip; the depthwise 3x3 is a small VALU stencil.  Conv bias and
// eval-mode BN are folded into one per-channel scale/shift applied in the producing kernel's epilogue.
// Internal layout: planar [b][c][HALO + h*W + w] (row length tcr_padded_len(H*W)).
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.h"

namespace tcr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DsConv1Args {
    const float* feat;      // [B][W_in][Tp_in]  (front-end planar: coefficient-major, time along the row)
    const float* w;         // [kh][4][1][Cout]
    const float* scale;
    const float* shift;
    float* y;               // [B][Cout][Pp]
    int npos;               // B * OH * OW
    int cout, h_in, w_in, tp_in, oh, ow, pp;
    int kh, sh, sw, pad_t, pad_l;
    int relu;               // scale == nullptr: y = acc + shift (conv bias, train-mode raw output)
    EpiSums sums;           // (dscnn_conv1_kernel) forward sums of y, y^2 from the epilogue: one partial row per 256 positions
};

// Epilogue of both conv_1 kernels.  One 64-bit row pointer per position tile and one 32-bit channel offset per
// accumulator row: the per-store address is a single add (48 stores with their own 64-bit multiply chains, plus the
// integer divisions of the position split, were most of this kernel's instructions).
template <int MT>
__device__ __forceinline__ void conv1_store(const DsConv1Args& a, const f32x4 (&acc)[MT][4], int pos0, int cot0, int r, int q) {
    const int P = a.oh * a.ow;
    const float inv_p = 1.0f / (float)P;
    float sc[MT][4], sf[MT][4];
    int off[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int co = (cot0 + m) * 16 + q * 4 + reg;
            const int cc = min(co, a.cout - 1);
            sc[m][reg] = a.scale ? a.scale[cc] : 1.0f;
            sf[m][reg] = a.shift[cc];
            off[m][reg] = co < a.cout ? cc * a.pp : -1;
        }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int p = pos0 + nt * 16 + r;
        if (p >= a.npos) continue;
        const int n = a.npos < (1 << 23) ? fast_div(p, P, inv_p) : p / P, rem = p - n * P;    // (float reciprocal: exact below 2^23)
        float* yb = a.y + (size_t)n * a.cout * a.pp + kHalo + rem;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if (off[m][reg] < 0) continue;
                float v = fmaf(acc[m][nt][reg], sc[m][reg], sf[m][reg]);
                if (a.relu) v = fmaxf(v, 0.f);
                yb[off[m][reg]] = v;
            }
    }
}

// D[row = co][col = (b, oh, ow)] = sum_{i < kh, j < 4} W[i][j][co] * x[oh*sh + i - pad_t][ow*sw + j - pad_l]
// (three waves per SIMD: with the epilogue sums the allocator otherwise takes 158 + 48 registers -- two waves -- and the training
//  forward's conv_1 went from 0.54 to 0.69 ms)
template <int MT>
__global__ __launch_bounds__(256) TCR_WAVES_PER_SIMD(3) void dscnn_conv1_kernel(const DsConv1Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int pos0 = (blockIdx.x * 4 + wave) * 64;
    const int cot0 = blockIdx.y * MT;
    const int P = a.oh * a.ow;
    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* xb[4];
    int h0[4];
    bool wok[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int p = min(pos0 + nt * 16 + r, a.npos - 1);
        const int n = p / P, rem = p - n * P;
        const int oh = rem / a.ow, ow = rem - oh * a.ow;
        const int wc = ow * a.sw + q - a.pad_l;                 // input column (MFCC coefficient) of this lane's tap j = q
        wok[nt] = wc >= 0 && wc < a.w_in;
        xb[nt] = a.feat + ((size_t)n * a.w_in + (wok[nt] ? wc : 0)) * a.tp_in + kHalo;
        h0[nt] = oh * a.sh - a.pad_t;
    }
    // operands of kernel row i + 1 are in flight while the 4 * MT MFMAs of row i issue (clamped, unconditional loads;
    // out-of-image taps are zeroed by a select)
    const float* wp[MT];
    bool cok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int co = (cot0 + m) * 16 + r;
        cok[m] = co < a.cout;
        wp[m] = a.w + (size_t)q * a.cout + (cok[m] ? co : 0);
    }
    auto load_row = [&](int i, float (&af)[MT], float (&bf)[4]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float v = wp[m][(size_t)i * 4 * a.cout];
            af[m] = cok[m] ? v : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int h = h0[nt] + i;
            const bool in = wok[nt] && h >= 0 && h < a.h_in;
            const float v = xb[nt][in ? h : 0];
            bf[nt] = in ? v : 0.f;
        }
    };
    float af[MT], bf[4];
    load_row(0, af, bf);
    for (int i = 0; i < a.kh; ++i) {
        float an[MT], bn[4];
        load_row(min(i + 1, a.kh - 1), an, bn);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[nt], acc[m][nt], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = an[m];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = bn[nt];
    }
    conv1_store<MT>(a, acc, pos0, cot0, r, q);
    if (a.sums.partial) {
        // BN batch statistics of the tile while it is in registers (train-mode raw output: y = acc + bias): a channel's 16
        // positions of a column tile are the 16 lanes of a DPP row; the four waves (4 x 64 positions, same channels) meet in LDS
        __shared__ float s_red[4][2][MT * 16];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cc = min((cot0 + m) * 16 + q * 4 + reg, a.cout - 1);
                const float sc = a.scale ? a.scale[cc] : 1.0f, sf = a.shift[cc];
                float q1 = 0.f, q2 = 0.f;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    float v = fmaf(acc[m][nt][reg], sc, sf);
                    if (a.relu) v = fmaxf(v, 0.f);
                    if (pos0 + nt * 16 + r >= a.npos) v = 0.f;
                    q1 += v;
                    q2 = fmaf(v, v, q2);
                }
                q1 = row16_sum(q1);
                q2 = row16_sum(q2);
                if (r == 0) { s_red[wave][0][m * 16 + q * 4 + reg] = q1; s_red[wave][1][m * 16 + q * 4 + reg] = q2; }
            }
        __syncthreads();
        if (threadIdx.x < 2 * MT * 16) {
            const int which = threadIdx.x / (MT * 16), j = threadIdx.x - which * (MT * 16);
            const int co = cot0 * 16 + j;
            if (co < a.cout)
                a.sums.partial[((size_t)blockIdx.x * 2 + which) * a.cout + co] =
                    (s_red[0][which][j] + s_red[1][which][j]) + (s_red[2][which][j] + s_red[3][which][j]);
        }
    }
}

static int dscnn_conv1_sum_rows(int npos) { return ceil_div(npos, 256); }

static int launch_dscnn_conv1(const DsConv1Args& a, hipStream_t s) {
    const int tiles = ceil_div(a.cout, 16);
    const dim3 grid(ceil_div(a.npos, 256), ceil_div(tiles, 3));
    hipLaunchKernelGGL((dscnn_conv1_kernel<3>), grid, dim3(256), 0, s, a);
    return check_launch("dscnn_conv1_kernel");
}

struct DsDwArgs {
    const float* x;         // [B][C][Ppi]
    const float* w;         // [3][3][C][1]
    const float* scale;
    const float* shift;
    float* y;               // [B][C][Ppo]
    int64_t total;          // B * C * OH * OW
    int c, h_in, w_in, ppi, oh, ow, ppo, sh, sw, pad_t, pad_l;
    int relu;
    // LDS kernels (DS-CNN training): x is the producing unit's RAW conv output -- the image is built from
    // relu(x * in_scale[c] + in_shift[c]) (bn_apply's expression; the padding stays zero) --, and the epilogue leaves the sums of
    // y, y^2 per plane: partial[n][2][C], one row per utterance (EpiSums, forward form)
    const float* in_scale = nullptr;
    const float* in_shift = nullptr;
    EpiSums sums;
    int rows_stencil = 0;   // dscnn_conv1_dw_kernel: the row-per-lane stencil (set by its launcher from TCR_TUNE_DW_FWD)
};

__global__ __launch_bounds__(256) void dscnn_depthwise_kernel(const DsDwArgs a) {
    // one wavefront per (utterance, channel) plane: lanes walk the output map, the 9 taps / scale / shift of the
    // plane's channel are wave-uniform (scalar loads)
    const int P = a.oh * a.ow;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // b * C + c
    if (row * P >= a.total) return;
    const int c = (int)(row % a.c);
    float wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = a.w[(size_t)k * a.c + c];
    const float sc = a.scale ? a.scale[c] : 1.0f, sh = a.shift[c];
    const float* xr = a.x + row * a.ppi + kHalo;
    float* yr = a.y + row * a.ppo + kHalo;
    const float inv_ow = 1.0f / (float)a.ow;
    for (int pos = lane; pos < P; pos += 64) {
        const int oh = fast_div(pos, a.ow, inv_ow), ow = pos - oh * a.ow;
        float s = 0.f;
#pragma unroll
        for (int di = 0; di < 3; ++di) {
            const int h = oh * a.sh + di - a.pad_t;
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) {
                const int w = ow * a.sw + dj - a.pad_l;
                const float xv = (h >= 0 && h < a.h_in && w >= 0 && w < a.w_in) ? xr[h * a.w_in + w] : 0.f;
                s = fmaf(wt[di * 3 + dj], xv, s);
            }
        }
        const float v = fmaf(s, sc, sh);
        yr[pos] = a.relu ? fmaxf(v, 0.f) : v;
    }
}

// LDS-staged form: a workgroup owns 16 consecutive planes (rows b*C + c are contiguous in the planar layout), builds their
// zero-padded input images in LDS with coalesced loads, and every 16-lane group then walks one plane's output map with
// its 9 taps in registers and no bounds checks.  (The wave-per-plane kernel above moves 0.5 KB per wave behind nine
// predicated loads per lane: 1.4 TB/s at DS-CNN-L sizes.)
__global__ __launch_bounds__(256) void dscnn_depthwise_lds_kernel(const DsDwArgs a, const int rows, const int img_r, const int img_c) {
    float* img = reinterpret_cast<float*>(dyn_lds());               // [16][img_r][img_c]
    const int plane = threadIdx.x >> 4, t16 = threadIdx.x & 15;
    const int row = min((int)blockIdx.x * 16 + plane, rows - 1);    // (tail workgroup: duplicates of the last plane, not stored)
    const bool live = (int)blockIdx.x * 16 + plane < rows;
    const int c = row % a.c;
    const int isz = img_r * img_c;
    const float* xr = a.x + (size_t)row * a.ppi + kHalo;
    float* im = img + plane * isz;
    const float inv_c = 1.0f / (float)img_c;
    const bool aff = a.in_scale != nullptr;
    const float isc = aff ? a.in_scale[c] : 1.0f, isf = aff ? a.in_shift[c] : 0.f;
    // batches of 8 independent (clamped, unconditional) loads per lane: a rolled loop waits out one global round trip
    // per 16 image elements
    for (int j0 = t16; j0 < isz; j0 += 16 * 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = min(j0 + 16 * i, isz - 1);
            const int rr = fast_div(j, img_c, inv_c), cc = j - rr * img_c;
            const int h = rr - a.pad_t, w = cc - a.pad_l;
            const bool in = h >= 0 && h < a.h_in && w >= 0 && w < a.w_in;
            float xv = xr[in ? h * a.w_in + w : 0];
            if (aff) xv = fmaxf(fmaf(xv, isc, isf), 0.f);
            v[i] = in ? xv : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (j0 + 16 * i < isz) im[j0 + 16 * i] = v[i];
    }
    float wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = a.w[(size_t)k * a.c + c];
    const float sc = a.scale ? a.scale[c] : 1.0f, sh = a.shift[c];
    __syncthreads();
    if (!live) return;
    const int P = a.oh * a.ow;
    const float inv_ow = 1.0f / (float)a.ow;
    float* yr = a.y + (size_t)row * a.ppo + kHalo;
    float q1 = 0.f, q2 = 0.f;
    for (int pos0 = t16; pos0 < P; pos0 += 16 * 5) {        // (13 x 5 maps: one trip, the 45 LDS reads of a lane in flight together)
        float s[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int pos = min(pos0 + 16 * i, P - 1);
            const int oh = fast_div(pos, a.ow, inv_ow), ow = pos - oh * a.ow;
            const float* p0 = im + oh * a.sh * img_c + ow * a.sw;
            s[i] = 0.f;
#pragma unroll
            for (int di = 0; di < 3; ++di)
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) s[i] = fmaf(wt[di * 3 + dj], p0[di * img_c + dj], s[i]);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            float v = fmaf(s[i], sc, sh);
            if (a.relu) v = fmaxf(v, 0.f);
            if (pos0 + 16 * i < P) {
                yr[pos0 + 16 * i] = v;
                q1 += v;
                q2 = fmaf(v, v, q2);
            }
        }
    }
    if (a.sums.partial) {       // (a plane's 16 lanes are one DPP row; `live` is uniform over it)
        q1 = row16_sum(q1);
        q2 = row16_sum(q2);
        if (t16 == 0) {
            const size_t n = (size_t)(row / a.c);
            a.sums.partial[(n * 2 + 0) * a.c + c] = q1;
            a.sums.partial[(n * 2 + 1) * a.c + c] = q2;
        }
    }
}

// Eval-mode fusion of conv_1 (+ folded BN + ReLU) with the first depthwise layer (+ folded BN + ReLU).  conv_1 writes
// B x 276 x 250 activations (1.17 GB at B = 4096) that the stride-2 depthwise layer immediately reduces 4x; here a workgroup
// owns (one utterance, 16 MT channels): the conv_1 tile goes from the MFMA accumulators straight into zero-padded LDS planes
// [48][(oh-1)*sh + 3][(ow-1)*sw + 3], and the 3x3 stencil reads them back -- conv_1's output never reaches HBM.
// (Requires conv_1's map to fit 256 positions: 25 x 10 for every DS-CNN size.)
template <int MT>
__global__ __launch_bounds__(256) void dscnn_conv1_dw_kernel(const DsConv1Args a, const DsDwArgs d, const int ir, const int ic, const int fs_off, const int fs_sz) {
    constexpr int CH = 16 * MT;
    float* planes = reinterpret_cast<float*>(dyn_lds());        // [CH][ir][ic]
    float* s_dw = planes + CH * ir * ic;                        // [9][CH] depthwise taps
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int n = blockIdx.x;
    const int cot0 = blockIdx.y * MT;
    const int P1 = a.oh * a.ow, P2 = d.oh * d.ow;
    const int isz = ir * ic;
    const bool rows5 = d.ow == 5 && d.sw == 2 && d.sh == 2 && ic == 11 && d.rows_stencil;
    for (int j = threadIdx.x; j < CH * isz; j += 256) planes[j] = 0.f;
    for (int j = threadIdx.x; j < 9 * CH; j += 256) {
        const int k = j / CH, cl = j - k * CH;
        s_dw[j] = d.w[(size_t)k * d.c + min(cot0 * 16 + cl, d.c - 1)];
    }

    // ---- conv_1 tile: positions wave * 64 + nt * 16 + r of utterance n (as dscnn_conv1_kernel) ----
    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float inv_ow1 = 1.0f / (float)a.ow;
    const float* xb[4];
    int h0[4], pdst[4];
    bool wok[4], pv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int rem_raw = wave * 64 + nt * 16 + r;
        pv[nt] = rem_raw < P1;
        const int rem = min(rem_raw, P1 - 1);
        const int oh = fast_div(rem, a.ow, inv_ow1), ow = rem - oh * a.ow;
        const int wc = ow * a.sw + q - a.pad_l;
        wok[nt] = wc >= 0 && wc < a.w_in;
        xb[nt] = a.feat + ((size_t)n * a.w_in + (wok[nt] ? wc : 0)) * a.tp_in + kHalo;
        h0[nt] = oh * a.sh - a.pad_t;
        pdst[nt] = (oh + d.pad_t) * ic + ow + d.pad_l;          // this position inside a padded plane
    }
    if (fs_off > 0) {
        // Round 5: both operands from LDS.  The utterance's feature map goes into a zero-padded image s_f[w_in + 3][(oh - 1) sh + kh]
        // (row = coefficient + pad_l, column = frame + pad_t) and the workgroup's weight slice into s_wt[kh * 4][48] (pitch 48: the four
        // tap columns of a fragment on two bank halves), so kernel row i of a lane's fragments is a fixed LDS address + i: ten unrolled
        // steps of 2 + 4 ds_reads at immediate offsets and 8 MFMAs.  (The loop below gathers 6 operands per step from global memory
        // with a 64-bit address, two range tests and a select each: ~480 of the kernel's ~1200 VALU instructions per wave, on a chip
        // where they add to the matrix time.)  Same products in the same order: bitwise.
        constexpr int KH = 10, WP = 48;
        float* s_f = planes + fs_off;
        float* s_wt = s_f + fs_sz;
        const int fpitch = (a.oh - 1) * a.sh + KH;
        const float inv_fp = 1.0f / (float)fpitch;
        const float* fu = a.feat + (size_t)n * a.w_in * a.tp_in + kHalo;
        for (int idx = threadIdx.x; idx < fs_sz; idx += 256) {
            const int row = fast_div(idx, fpitch, inv_fp), col = idx - row * fpitch;
            const int w = row - a.pad_l, h = col - a.pad_t;
            const bool in = w >= 0 && w < a.w_in && h >= 0 && h < a.h_in;
            const float v = fu[in ? w * a.tp_in + h : 0];
            s_f[idx] = in ? v : 0.f;
        }
        for (int idx = threadIdx.x; idx < KH * 4 * CH; idx += 256) {
            const int tap = idx / CH, cl = idx - tap * CH;
            const int co = cot0 * 16 + cl;
            const float v = a.w[(size_t)tap * a.cout + min(co, a.cout - 1)];
            s_wt[tap * WP + cl] = co < a.cout ? v : 0.f;
        }
        __syncthreads();                                        // (also: planes zeroed, taps staged)
        int fo[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int rem = min(wave * 64 + nt * 16 + r, P1 - 1);
            const int oh = fast_div(rem, a.ow, inv_ow1), ow = rem - oh * a.ow;
            fo[nt] = (ow * a.sw + q) * fpitch + oh * a.sh;
        }
        const float* wa = s_wt + q * WP + r;
        float af[MT], bf[4];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = wa[m * 16];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = s_f[fo[nt]];
#pragma unroll
        for (int i = 0; i < KH; ++i) {
            float an[MT], bn[4];
            if (i + 1 < KH) {
#pragma unroll
                for (int m = 0; m < MT; ++m) an[m] = wa[(i + 1) * 4 * WP + m * 16];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bn[nt] = s_f[fo[nt] + i + 1];
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[nt], acc[m][nt], 0, 0, 0);
            if (i + 1 < KH) {
#pragma unroll
                for (int m = 0; m < MT; ++m) af[m] = an[m];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bf[nt] = bn[nt];
            }
        }
    } else {
    const float* wp[MT];
    bool cok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int co = (cot0 + m) * 16 + r;
        cok[m] = co < a.cout;
        wp[m] = a.w + (size_t)q * a.cout + (cok[m] ? co : 0);
    }
    auto load_row = [&](int i, float (&af)[MT], float (&bf)[4]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float v = wp[m][(size_t)i * 4 * a.cout];
            af[m] = cok[m] ? v : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int h = h0[nt] + i;
            const bool in = wok[nt] && h >= 0 && h < a.h_in;
            const float v = xb[nt][in ? h : 0];
            bf[nt] = in ? v : 0.f;
        }
    };
    float af[MT], bf[4];
    load_row(0, af, bf);
    for (int i = 0; i < a.kh; ++i) {
        float an[MT], bn[4];
        load_row(min(i + 1, a.kh - 1), an, bn);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[nt], acc[m][nt], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = an[m];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = bn[nt];
    }
    __syncthreads();                                            // planes zeroed, taps staged
    }
    // ---- conv_1 epilogue (folded BN + ReLU) into the padded planes ----
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int cl = m * 16 + q * 4 + reg;
            const int cc = min(cot0 * 16 + cl, a.cout - 1);
            const float sc = a.scale[cc], sf = a.shift[cc];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                if (pv[nt]) planes[cl * isz + pdst[nt]] = fmaxf(fmaf(acc[m][nt][reg], sc, sf), 0.f);
        }
    __syncthreads();
    // ---- depthwise 3x3 (+ folded BN + ReLU) ----
    if (rows5) {
        // 5-wide stride-2 output rows (every DS-CNN size): one lane per (channel, output row) -- the three padded plane rows it needs (11
        // floats each) and the channel's nine taps are read once for five outputs: ~20 instead of ~60 VALU instructions and 8.4 instead
        // of 18 LDS reads per output (the element loop below pays two divisions and 18 reads for each).  Same fmaf chain per output.
        constexpr int OW = 5, RW = 2 * (OW - 1) + 3;
        const float inv_oh = 1.0f / (float)d.oh;
        for (int task = threadIdx.x; task < CH * d.oh; task += 256) {
            const int cl = fast_div(task, d.oh, inv_oh), oh = task - cl * d.oh;
            const int c = cot0 * 16 + cl;
            if (c >= d.c) break;                                // (channels ascend with task)
            float wt[9], in[3][RW];
#pragma unroll
            for (int k = 0; k < 9; ++k) wt[k] = s_dw[k * CH + cl];
            const float* p0 = planes + cl * isz + oh * 2 * ic;
#pragma unroll
            for (int di = 0; di < 3; ++di)
#pragma unroll
                for (int j = 0; j < RW; ++j) in[di][j] = p0[di * ic + j];
            const float sc = d.scale[c], sf = d.shift[c];
            float* yo = d.y + ((size_t)n * d.c + c) * d.ppo + kHalo + oh * OW;
#pragma unroll
            for (int w = 0; w < OW; ++w) {
                float sacc = 0.f;
#pragma unroll
                for (int di = 0; di < 3; ++di)
#pragma unroll
                    for (int dj = 0; dj < 3; ++dj) sacc = fmaf(wt[di * 3 + dj], in[di][2 * w + dj], sacc);
                yo[w] = fmaxf(fmaf(sacc, sc, sf), 0.f);
            }
        }
        return;
    }
    const float inv_p2 = 1.0f / (float)P2, inv_ow2 = 1.0f / (float)d.ow;
    for (int idx = threadIdx.x; idx < CH * P2; idx += 256) {
        const int cl = fast_div(idx, P2, inv_p2), pos = idx - cl * P2;
        const int c = cot0 * 16 + cl;
        if (c >= d.c) break;                                    // (channels ascend with idx)
        const int oh = fast_div(pos, d.ow, inv_ow2), ow = pos - oh * d.ow;
        const float* p0 = planes + cl * isz + oh * d.sh * ic + ow * d.sw;
        float sacc = 0.f;
#pragma unroll
        for (int di = 0; di < 3; ++di)
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) sacc = fmaf(s_dw[(di * 3 + dj) * CH + cl], p0[di * ic + dj], sacc);
        const float v = fmaf(sacc, d.scale[c], d.shift[c]);
        d.y[((size_t)n * d.c + c) * d.ppo + kHalo + pos] = fmaxf(v, 0.f);
    }
}

// The same fusion with a workgroup walking SEVERAL utterances (round 5): the depthwise taps, the workgroup's conv_1 weight slice, both
// layers' folded scale / shift and the planes' zero border are set up ONCE; per utterance the 2.3 KB feature map is the only thing fetched
// (into registers while the previous utterance's MFMAs run, then into the zero-padded LDS image), conv_1 runs out of LDS (ten unrolled
// steps), its epilogue overwrites the plane interiors and the stencil runs one lane per (channel, output row).  The one-utterance kernel
// above pays the weight / tap / scale staging and three exposed global round trips per 80 MFMAs of a wave.  Shapes: kh = 10, 5-wide
// stride-2 depthwise rows (every DS-CNN size); same products and sums in the same order: bitwise.
template <int MT>
__global__ __launch_bounds__(256) void dscnn_conv1_dw_loop_kernel(const DsConv1Args a, const DsDwArgs d, const int ir, const int ic, const int fs_off,
                                                                  const int fs_sz, const int upw, const int batch) {
    constexpr int CH = 16 * MT, KH = 10, WP = 48, OW = 5, RW = 2 * (OW - 1) + 3, FPT = 4;
    float* planes = reinterpret_cast<float*>(dyn_lds());        // [CH][ir][ic]
    float* s_dw = planes + CH * ir * ic;                        // [9][CH]
    float* s_f = planes + fs_off;                               // [w_in + 3][fpitch]
    float* s_wt = s_f + fs_sz;                                  // [KH * 4][WP]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int cot0 = blockIdx.y * MT;
    const int P1 = a.oh * a.ow;
    const int isz = ir * ic;
    const int n_begin = blockIdx.x * upw, n_end = min(n_begin + upw, batch);
    for (int j = tid; j < CH * isz; j += 256) planes[j] = 0.f;
    for (int j = tid; j < 9 * CH; j += 256) {
        const int k = j / CH, cl = j - k * CH;
        s_dw[j] = d.w[(size_t)k * d.c + min(cot0 * 16 + cl, d.c - 1)];
    }
    for (int idx = tid; idx < KH * 4 * CH; idx += 256) {
        const int tap = idx / CH, cl = idx - tap * CH;
        const int co = cot0 * 16 + cl;
        const float v = a.w[(size_t)tap * a.cout + min(co, a.cout - 1)];
        s_wt[tap * WP + cl] = co < a.cout ? v : 0.f;
    }
    // feature image roles of this thread (fixed for the kernel): element idx = tid + 256 k of s_f
    const int fpitch = (a.oh - 1) * a.sh + KH;
    const float inv_fp = 1.0f / (float)fpitch;
    int fsrc[FPT];
    bool fin[FPT];
#pragma unroll
    for (int k = 0; k < FPT; ++k) {
        const int idx = min(tid + 256 * k, fs_sz - 1);
        const int row = fast_div(idx, fpitch, inv_fp), col = idx - row * fpitch;
        const int w = row - a.pad_l, h = col - a.pad_t;
        fin[k] = w >= 0 && w < a.w_in && h >= 0 && h < a.h_in;
        fsrc[k] = fin[k] ? w * a.tp_in + kHalo + h : 0;
    }
    float fr[FPT];
    auto fetch = [&](int n) {
        const float* fu = a.feat + (size_t)n * a.w_in * a.tp_in;
#pragma unroll
        for (int k = 0; k < FPT; ++k) fr[k] = fu[fsrc[k]];
    };
    // conv_1 fragments / epilogue roles
    const float inv_ow1 = 1.0f / (float)a.ow;
    int fo[4], pdst[4];
    bool pv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int rem_raw = wave * 64 + nt * 16 + r;
        pv[nt] = rem_raw < P1;
        const int rem = min(rem_raw, P1 - 1);
        const int oh = fast_div(rem, a.ow, inv_ow1), ow = rem - oh * a.ow;
        fo[nt] = (ow * a.sw + q) * fpitch + oh * a.sh;
        pdst[nt] = (oh + d.pad_t) * ic + ow + d.pad_l;
    }
    float esc[MT][4], esf[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int cc = min(cot0 * 16 + m * 16 + q * 4 + reg, a.cout - 1);
            esc[m][reg] = a.scale[cc]; esf[m][reg] = a.shift[cc];
        }
    // stencil roles: tasks tid and tid + 256 of CH * d.oh (channel, output row) pairs
    const float inv_oh = 1.0f / (float)d.oh;
    int tcl[2], toh[2], tc[2];
    bool tok[2];
    float tsc[2], tsf[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int task = tid + 256 * k;
        tcl[k] = min(fast_div(task, d.oh, inv_oh), CH - 1);
        toh[k] = task - fast_div(task, d.oh, inv_oh) * d.oh;
        tc[k] = cot0 * 16 + tcl[k];
        tok[k] = task < CH * d.oh && tc[k] < d.c;
        const int cc = min(tc[k], d.c - 1);
        tsc[k] = d.scale[cc]; tsf[k] = d.shift[cc];
    }
    const float* wa = s_wt + q * WP + r;
    if (n_begin < n_end) fetch(n_begin);
    for (int n = n_begin; n < n_end; ++n) {
#pragma unroll
        for (int k = 0; k < FPT; ++k)
            if (tid + 256 * k < fs_sz) s_f[tid + 256 * k] = fin[k] ? fr[k] : 0.f;
        __syncthreads();                                        // feature image (first trip: planes, taps, weights too); the previous stencil is done
        if (n + 1 < n_end) fetch(n + 1);
        f32x4 acc[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float af[MT], bf[4];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = wa[m * 16];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = s_f[fo[nt]];
#pragma unroll
        for (int i = 0; i < KH; ++i) {
            float an[MT], bn[4];
            if (i + 1 < KH) {
#pragma unroll
                for (int m = 0; m < MT; ++m) an[m] = wa[(i + 1) * 4 * WP + m * 16];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bn[nt] = s_f[fo[nt] + i + 1];
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[nt], acc[m][nt], 0, 0, 0);
            if (i + 1 < KH) {
#pragma unroll
                for (int m = 0; m < MT; ++m) af[m] = an[m];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bf[nt] = bn[nt];
            }
        }
        // conv_1 epilogue (folded BN + ReLU) into the padded planes
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cl = m * 16 + q * 4 + reg;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    if (pv[nt]) planes[cl * isz + pdst[nt]] = fmaxf(fmaf(acc[m][nt][reg], esc[m][reg], esf[m][reg]), 0.f);
            }
        __syncthreads();
        // depthwise 3x3 stride 2 (+ folded BN + ReLU), one lane per (channel, output row)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!tok[k]) continue;
            float wt[9], in[3][RW];
#pragma unroll
            for (int t = 0; t < 9; ++t) wt[t] = s_dw[t * CH + tcl[k]];
            const float* p0 = planes + tcl[k] * isz + toh[k] * 2 * ic;
#pragma unroll
            for (int di = 0; di < 3; ++di)
#pragma unroll
                for (int j = 0; j < RW; ++j) in[di][j] = p0[di * ic + j];
            float* yo = d.y + ((size_t)n * d.c + tc[k]) * d.ppo + kHalo + toh[k] * OW;
#pragma unroll
            for (int w = 0; w < OW; ++w) {
                float sacc = 0.f;
#pragma unroll
                for (int di = 0; di < 3; ++di)
#pragma unroll
                    for (int dj = 0; dj < 3; ++dj) sacc = fmaf(wt[di * 3 + dj], in[di][2 * w + dj], sacc);
                yo[w] = fmaxf(fmaf(sacc, tsc[k], tsf[k]), 0.f);
            }
        }
    }
}

// returns 1 (nothing launched) when the shapes do not fit the fused kernel
static int launch_dscnn_conv1_dw(const DsConv1Args& a, const DsDwArgs& d, int batch, hipStream_t s) {
    const int ir = (d.oh - 1) * d.sh + 3, ic = (d.ow - 1) * d.sw + 3;
    const size_t lds = ((size_t)32 * ir * ic + 9 * 32) * sizeof(float);
    if (a.oh * a.ow > 256 || lds > 64 * 1024 || d.c != a.cout || d.h_in != a.oh || d.w_in != a.ow || !a.scale || !d.scale || !a.relu || !d.relu ||
        d.pad_t + a.oh > ir || d.pad_l + a.ow > ic || batch > 65535 * 16) return 1;
    // 32 channels per workgroup (38 KB of LDS, 4 workgroups per CU): 4.15 ms; 48 channels: 4.28 ms; 16: 4.41 ms (DS-CNN-L eval)
    const dim3 grid(batch, ceil_div(ceil_div(a.cout, 16), 2));
    DsDwArgs d2 = d;
    d2.rows_stencil = tune_get(TCR_TUNE_DW_FWD) != 1 ? 1 : 0;
    // both conv_1 operands from LDS (kh = 10 kernels: every DS-CNN size): feature image + weight slice behind the planes and the depthwise taps
    int fs_off = 0, fs_sz = 0;
    size_t lds_all = lds;
    if (d2.rows_stencil && a.kh == 10 && a.oh * a.ow <= 256) {
        fs_sz = (a.w_in + 3) * ((a.oh - 1) * a.sh + 10);
        fs_off = 32 * ir * ic + 9 * 32;
        lds_all = ((size_t)fs_off + fs_sz + 40 * 48) * sizeof(float);
        if (lds_all > 64 * 1024) { fs_off = 0; fs_sz = 0; lds_all = lds; }
    }
    if (fs_off > 0 && d.ow == 5 && d.sw == 2 && d.sh == 2 && ic == 11 && fs_sz <= 4 * 256 && 32 * d.oh <= 512) {
        // utterances per workgroup: 8 at throughput sizes (4096 / 8 x 9 channel groups = 4608 workgroups, six rounds of the ~768 slots
        // -- ~50 KB of LDS: three workgroups per CU); fewer while the grid would not fill those slots even once, so that small and mid
        // batches keep one workgroup per (utterance, channel group) instead of walking 8 utterances serially on a mostly idle chip
        // (advisor, round 5).  An utterance's result does not depend on upw: the loop handles them one after the other.
        const int slots = 3 * device_cus();
        const int upw = max(1, min(8, (batch * (int)grid.y) / slots));
        const dim3 lgrid(ceil_div(batch, upw), grid.y);
        hipLaunchKernelGGL((dscnn_conv1_dw_loop_kernel<2>), lgrid, dim3(256), lds_all, s, a, d2, ir, ic, fs_off, fs_sz, upw, batch);
        return check_launch("dscnn_conv1_dw_loop_kernel");
    }
    hipLaunchKernelGGL((dscnn_conv1_dw_kernel<2>), grid, dim3(256), lds_all, s, a, d2, ir, ic, fs_off, fs_sz);
    return check_launch("dscnn_conv1_dw_kernel");
}

// Small maps (image <= 128 elements: the 13 x 5 layers): every 16-lane group carries TWO planes (g and g + 16 of the
// workgroup's 32), so twice the bytes are in flight per wave at the same occupancy -- these layers are pure HBM streams
// (0.27 ms per 688 MB with one plane per group).
__global__ __launch_bounds__(256) void dscnn_depthwise_lds2_kernel(const DsDwArgs a, const int rows, const int img_r, const int img_c) {
    float* img = reinterpret_cast<float*>(dyn_lds());               // [32][img_r][img_c]
    const int g = threadIdx.x >> 4, t16 = threadIdx.x & 15;
    const int isz = img_r * img_c;
    const float inv_c = 1.0f / (float)img_c;
    int row[2];
    bool live[2];
    float v[2][8];
    const bool aff = a.in_scale != nullptr;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const int rr0 = (int)blockIdx.x * 32 + g + 16 * pl;
        live[pl] = rr0 < rows;
        row[pl] = min(rr0, rows - 1);
        const float* xr = a.x + (size_t)row[pl] * a.ppi + kHalo;
        const float isc = aff ? a.in_scale[row[pl] % a.c] : 1.0f, isf = aff ? a.in_shift[row[pl] % a.c] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = min(t16 + 16 * i, isz - 1);
            const int rr = fast_div(j, img_c, inv_c), cc = j - rr * img_c;
            const int h = rr - a.pad_t, w = cc - a.pad_l;
            const bool in = h >= 0 && h < a.h_in && w >= 0 && w < a.w_in;
            float xv = xr[in ? h * a.w_in + w : 0];
            if (aff) xv = fmaxf(fmaf(xv, isc, isf), 0.f);
            v[pl][i] = in ? xv : 0.f;
        }
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (t16 + 16 * i < isz) img[(g + 16 * pl) * isz + t16 + 16 * i] = v[pl][i];
    __syncthreads();
    const int P = a.oh * a.ow;
    const float inv_ow = 1.0f / (float)a.ow;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        if (!live[pl]) continue;
        const int c = row[pl] % a.c;
        float wt[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wt[k] = a.w[(size_t)k * a.c + c];
        const float sc = a.scale ? a.scale[c] : 1.0f, sh = a.shift[c];
        const float* im = img + (g + 16 * pl) * isz;
        float* yr = a.y + (size_t)row[pl] * a.ppo + kHalo;
        float q1 = 0.f, q2 = 0.f;
        for (int pos0 = t16; pos0 < P; pos0 += 16 * 5) {
            float s[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int pos = min(pos0 + 16 * i, P - 1);
                const int oh = fast_div(pos, a.ow, inv_ow), ow = pos - oh * a.ow;
                const float* p0 = im + oh * a.sh * img_c + ow * a.sw;
                s[i] = 0.f;
#pragma unroll
                for (int di = 0; di < 3; ++di)
#pragma unroll
                    for (int dj = 0; dj < 3; ++dj) s[i] = fmaf(wt[di * 3 + dj], p0[di * img_c + dj], s[i]);
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                float o = fmaf(s[i], sc, sh);
                if (a.relu) o = fmaxf(o, 0.f);
                if (pos0 + 16 * i < P) {
                    yr[pos0 + 16 * i] = o;
                    q1 += o;
                    q2 = fmaf(o, o, q2);
                }
            }
        }
        if (a.sums.partial) {       // (a plane's 16 lanes are one DPP row; live[pl] is uniform over it)
            q1 = row16_sum(q1);
            q2 = row16_sum(q2);
            if (t16 == 0) {
                const size_t n = (size_t)(row[pl] / a.c);
                a.sums.partial[(n * 2 + 0) * a.c + c] = q1;
                a.sums.partial[(n * 2 + 1) * a.c + c] = q2;
            }
        }
    }
}

// The stride-1 layers on 13 x 5 maps, every byte through HBM as a contiguous float4 (see dscnn_dw_dgrad_rows_kernel): a workgroup owns 16
// consecutive planes = one contiguous block of 16 x Pp floats of x and of y.  x is copied to LDS as it lies; lane (plane, h) computes the
// W outputs of map row h from the three x rows around it (in-affine + ReLU where they leave LDS, zeros outside the map; the same fmaf
// chain per output as the image kernels) into an LDS copy of the y block, which leaves as float4 (halo floats zero).  Epilogue sums from
// the LDS copy by 16 lanes per plane in the image kernels' order: bitwise the same y and partial rows.
template <int H, int W>
__global__ __launch_bounds__(256) void dscnn_depthwise_rows_kernel(const DsDwArgs a) {
    constexpr int NPL = 16, P = H * W;
    static_assert(NPL * H <= 256, "one lane per (plane, map row)");
    const int pp = a.ppo;                                       // (== ppi)
    const int blk = NPL * pp, n4 = blk / 4;
    float* s_x = reinterpret_cast<float*>(dyn_lds());
    float* s_y = s_x + blk;
    const int tid = threadIdx.x;
    const size_t g0 = (size_t)blockIdx.x * blk;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x + g0);
    const bool two = tid + 256 < n4;
    const f32x4 z0 = x4[tid], z1 = x4[two ? tid + 256 : tid];
    const int row0 = blockIdx.x * NPL;
    const int n0 = row0 / a.c, c0 = row0 - n0 * a.c;
    const int pl3 = tid / H, h = tid - pl3 * H;                 // the compute phase's (plane, map row)
    const bool act3 = pl3 < NPL;
    int c3 = c0 + (act3 ? pl3 : 0);
    if (c3 >= a.c) c3 -= a.c;
    float wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = a.w[(size_t)k * a.c + c3];
    const bool aff = a.in_scale != nullptr;
    const float isc = aff ? a.in_scale[c3] : 1.0f, isf = aff ? a.in_shift[c3] : 0.f;
    const float sc = a.scale ? a.scale[c3] : 1.0f, sh = a.shift[c3];
    reinterpret_cast<f32x4*>(s_x)[tid] = z0;
    if (two) reinterpret_cast<f32x4*>(s_x)[tid + 256] = z1;
    if (tid < NPL * 2 * kHalo) {
        const int pl = tid / (2 * kHalo), k = tid - pl * (2 * kHalo);
        s_y[pl * pp + (k < kHalo ? k : P + k)] = 0.f;
    }
    __syncthreads();
    if (act3) {
        float in[3][W + 2];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {                        // x row h + rr - 1 (tap row di = rr)
            const int hh = h + rr - 1;
            const bool hv = hh >= 0 && hh < H;
            const float* src = s_x + pl3 * pp + kHalo + (hv ? hh : h) * W;
            in[rr][0] = 0.f; in[rr][W + 1] = 0.f;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                float v = src[w];
                if (aff) v = fmaxf(fmaf(v, isc, isf), 0.f);
                in[rr][w + 1] = hv ? v : 0.f;
            }
        }
        float* dst = s_y + pl3 * pp + kHalo + h * W;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            float sacc = 0.f;
#pragma unroll
            for (int di = 0; di < 3; ++di)
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) sacc = fmaf(wt[di * 3 + dj], in[di][w + dj], sacc);          // x[h + di - 1][w + dj - 1]
            float o = fmaf(sacc, sc, sh);
            if (a.relu) o = fmaxf(o, 0.f);
            dst[w] = o;
        }
    }
    __syncthreads();
    f32x4* y4 = reinterpret_cast<f32x4*>(a.y + g0);
    y4[tid] = reinterpret_cast<const f32x4*>(s_y)[tid];
    if (two) y4[tid + 256] = reinterpret_cast<const f32x4*>(s_y)[tid + 256];
    if (a.sums.partial) {
        const int plane = tid >> 4, t16 = tid & 15;
        int c = c0 + plane, n = n0;
        if (c >= a.c) { c -= a.c; ++n; }
        const float* sy = s_y + plane * pp + kHalo;
        float q1 = 0.f, q2 = 0.f;
#pragma unroll
        for (int i = 0; i < (P + 15) / 16; ++i) {
            const int pos = t16 + 16 * i;
            if (pos < P) {
                const float o = sy[pos];
                q1 += o;
                q2 = fmaf(o, o, q2);
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

// The stride-2 layer (25 x 10 -> 13 x 5, TF SAME: one zero row above, one zero column right), same scheme: the x block of 16 planes
// (16 x 258 floats, contiguous) through LDS, lane (plane, oh) computes output row oh from x rows 2 oh - 1 .. 2 oh + 1 (8-byte LDS reads),
// the y block leaves as float4.  Bitwise the image kernel's outputs and statistics.
template <int HI, int WI, int HO, int WO>
__global__ __launch_bounds__(256) void dscnn_depthwise_rows_s2_kernel(const DsDwArgs a) {
    constexpr int NPL = 16, PI = HI * WI, PO = HO * WO, PPI = PI + 2 * kHalo, PPO = PO + 2 * kHalo;
    static_assert(NPL * HO <= 256 && WI % 2 == 0 && (PPI % 2) == 0 && (NPL * PPI) % 4 == 0 && (NPL * PPO) % 4 == 0 && WI == 2 * WO && HI == 2 * HO - 1, "geometry");
    constexpr int NX4 = NPL * PPI / 4, NY4 = NPL * PPO / 4;
    float* s_x = reinterpret_cast<float*>(dyn_lds());
    float* s_y = s_x + NPL * PPI;
    const int tid = threadIdx.x;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x + (size_t)blockIdx.x * (NPL * PPI));
    constexpr int XPT = (NX4 + 255) / 256;
    f32x4 xv[XPT];
#pragma unroll
    for (int j = 0; j < XPT; ++j) xv[j] = x4[min(tid + 256 * j, NX4 - 1)];
    const int row0 = blockIdx.x * NPL;
    const int n0 = row0 / a.c, c0 = row0 - n0 * a.c;
    const int pl3 = tid / HO, oh = tid - pl3 * HO;
    const bool act3 = pl3 < NPL;
    int c3 = c0 + (act3 ? pl3 : 0);
    if (c3 >= a.c) c3 -= a.c;
    float wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = a.w[(size_t)k * a.c + c3];
    const bool aff = a.in_scale != nullptr;
    const float isc = aff ? a.in_scale[c3] : 1.0f, isf = aff ? a.in_shift[c3] : 0.f;
    const float sc = a.scale ? a.scale[c3] : 1.0f, sh = a.shift[c3];
#pragma unroll
    for (int j = 0; j < XPT; ++j)
        if (tid + 256 * j < NX4) reinterpret_cast<f32x4*>(s_x)[tid + 256 * j] = xv[j];
    if (tid < NPL * 2 * kHalo) {
        const int pl = tid / (2 * kHalo), k = tid - pl * (2 * kHalo);
        s_y[pl * PPO + (k < kHalo ? k : PO + k)] = 0.f;
    }
    __syncthreads();
    if (act3) {
        float in[3][WI + 1];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {                        // x row 2 oh + rr - 1 (tap row di = rr)
            const int hh = 2 * oh + rr - 1;
            const bool hv = hh >= 0 && hh < HI;
            const float2* src = reinterpret_cast<const float2*>(s_x + pl3 * PPI + kHalo + (hv ? hh : 0) * WI);
            in[rr][WI] = 0.f;                                   // (the zero column right of the map)
#pragma unroll
            for (int w2 = 0; w2 < WI / 2; ++w2) {
                const float2 t = src[w2];
                float v0 = t.x, v1 = t.y;
                if (aff) { v0 = fmaxf(fmaf(v0, isc, isf), 0.f); v1 = fmaxf(fmaf(v1, isc, isf), 0.f); }
                in[rr][2 * w2] = hv ? v0 : 0.f;
                in[rr][2 * w2 + 1] = hv ? v1 : 0.f;
            }
        }
        float* dst = s_y + pl3 * PPO + kHalo + oh * WO;
#pragma unroll
        for (int w = 0; w < WO; ++w) {
            float sacc = 0.f;
#pragma unroll
            for (int di = 0; di < 3; ++di)
#pragma unroll
                for (int dj = 0; dj < 3; ++dj) sacc = fmaf(wt[di * 3 + dj], in[di][2 * w + dj], sacc);      // x[2 oh + di - 1][2 ow + dj]
            float o = fmaf(sacc, sc, sh);
            if (a.relu) o = fmaxf(o, 0.f);
            dst[w] = o;
        }
    }
    __syncthreads();
    f32x4* y4 = reinterpret_cast<f32x4*>(a.y + (size_t)blockIdx.x * (NPL * PPO));
    y4[tid] = reinterpret_cast<const f32x4*>(s_y)[tid];
    if (tid + 256 < NY4) y4[tid + 256] = reinterpret_cast<const f32x4*>(s_y)[tid + 256];
    if (a.sums.partial) {
        const int plane = tid >> 4, t16 = tid & 15;
        int c = c0 + plane, n = n0;
        if (c >= a.c) { c -= a.c; ++n; }
        const float* sy = s_y + plane * PPO + kHalo;
        float q1 = 0.f, q2 = 0.f;
#pragma unroll
        for (int i = 0; i < (PO + 15) / 16; ++i) {
            const int pos = t16 + 16 * i;
            if (pos < PO) {
                const float o = sy[pos];
                q1 += o;
                q2 = fmaf(o, o, q2);
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

static bool dscnn_depthwise_rows_s2_covers(const DsDwArgs& d, int batch) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const int64_t planes = (int64_t)batch * d.c;
    return d.sh == 2 && d.sw == 2 && d.pad_t == 1 && d.pad_l == 0 && d.h_in == 25 && d.w_in == 10 && d.oh == 13 && d.ow == 5 &&
           d.ppi == 25 * 10 + 2 * kHalo && d.ppo == 13 * 5 + 2 * kHalo && planes % 16 == 0 && planes / 16 < (int64_t)1 << 27 && d.c >= 16 &&
           al(d.x) && al(d.y) && tune_get(TCR_TUNE_DW_FWD) != 1;
}

static bool dscnn_depthwise_rows_covers(const DsDwArgs& d, int batch) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const int64_t planes = (int64_t)batch * d.c;
    return d.sh == 1 && d.sw == 1 && d.pad_t == 1 && d.pad_l == 1 && d.h_in == 13 && d.w_in == 5 && d.oh == 13 && d.ow == 5 && d.ppi == d.ppo &&
           d.ppo == 13 * 5 + 2 * kHalo && planes % 16 == 0 && planes / 16 < (int64_t)1 << 27 && d.c >= 16 && al(d.x) && al(d.y) &&
           tune_get(TCR_TUNE_DW_FWD) != 1;
}

static int launch_dscnn_depthwise(const DsDwArgs& d, int batch, hipStream_t s) {
    if (dscnn_depthwise_rows_covers(d, batch)) {
        hipLaunchKernelGGL((dscnn_depthwise_rows_kernel<13, 5>), dim3((unsigned)((int64_t)batch * d.c / 16)), dim3(256), (size_t)2 * 16 * d.ppo * sizeof(float), s, d);
        return check_launch("dscnn_depthwise_rows_kernel");
    }
    if (dscnn_depthwise_rows_s2_covers(d, batch)) {
        hipLaunchKernelGGL((dscnn_depthwise_rows_s2_kernel<25, 10, 13, 5>), dim3((unsigned)((int64_t)batch * d.c / 16)), dim3(256),
                           (size_t)16 * (d.ppi + d.ppo) * sizeof(float), s, d);
        return check_launch("dscnn_depthwise_rows_s2_kernel");
    }
    const int rows = batch * d.c;
    const int img_r = (d.oh - 1) * d.sh + 3, img_c = (d.ow - 1) * d.sw + 3;
    const size_t lds = (size_t)16 * img_r * img_c * sizeof(float);
    if (img_r * img_c <= 128 && (int64_t)batch * d.c < ((int64_t)1 << 31) && tune_get(TCR_TUNE_CONV_B) != 3) {
        hipLaunchKernelGGL(dscnn_depthwise_lds2_kernel, dim3(ceil_div(rows, 32)), dim3(256), 2 * lds, s, d, rows, img_r, img_c);
        return check_launch("dscnn_depthwise_lds2_kernel");
    }
    if (lds <= 64 * 1024 && (int64_t)batch * d.c < ((int64_t)1 << 31)) {
        hipLaunchKernelGGL(dscnn_depthwise_lds_kernel, dim3(ceil_div(rows, 16)), dim3(256), lds, s, d, rows, img_r, img_c);
        return check_launch("dscnn_depthwise_lds_kernel");
    }
    if (d.in_scale || d.sums.partial) { set_error("dscnn depthwise: in-affine / epilogue sums need the LDS kernels"); return TCR_ERR_ARG; }
    hipLaunchKernelGGL(dscnn_depthwise_kernel, dim3((unsigned)ceil_div64((int64_t)batch * d.c, 4)), dim3(256), 0, s, d);
    return check_launch("dscnn_depthwise_kernel");
}

static bool dscnn_depthwise_lds_covers(int oh, int ow, int sh, int sw) {
    return (size_t)16 * ((oh - 1) * sh + 3) * ((ow - 1) * sw + 3) * sizeof(float) <= 64 * 1024;
}

// pooled[b][c][HALO] = mean over the P positions of plane (b, c); feeds head_fwd_kernel with T = 1 when the map is large
// (DS-CNN: 13 x 5 = 65).
__global__ __launch_bounds__(256) void plane_mean_kernel(const float* __restrict__ x, float* __restrict__ pooled, int64_t rows, int p, int pp,
                                                         const float* __restrict__ in_scale = nullptr, const float* __restrict__ in_shift = nullptr, int c = 1) {
    // 16 lanes per plane (16 planes per workgroup: more rows in flight than one wavefront per 65-element plane); the 16
    // partial sums are combined with a fixed xor tree
    const int t16 = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const float* xr = x + (row < rows ? row : rows - 1) * pp + kHalo;
    float s = 0.f;
    if (in_scale) {             // x is a raw train-mode conv output: pool relu(x * in_scale[ch] + in_shift[ch])
        const int ch = (int)((row < rows ? row : rows - 1) % c);
        const float sc = in_scale[ch], sf = in_shift[ch];
        for (int i = t16; i < p; i += 16) s += fmaxf(fmaf(xr[i], sc, sf), 0.f);
    } else {
        for (int i = t16; i < p; i += 16) s += xr[i];
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (t16 == 0 && row < rows) pooled[row * (1 + 2 * kHalo) + kHalo] = s / (float)p;
}

// The same pooling with the workgroup's 16 planes fetched as ONE contiguous block of float4 (16 x Pp floats) through LDS; every 16-lane group
// then adds its plane in the order of the kernel above: bitwise.  (Rows x Pp a multiple of 4, whole blocks of 16 planes: the launcher checks.)
__global__ __launch_bounds__(256) void plane_mean_block_kernel(const float* __restrict__ x, float* __restrict__ pooled, int p, int pp,
                                                               const float* __restrict__ in_scale, const float* __restrict__ in_shift, int c) {
    float* s_x = reinterpret_cast<float*>(dyn_lds());
    const int tid = threadIdx.x;
    const int n4 = 16 * pp / 4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x + (size_t)blockIdx.x * 16 * pp);
    for (int i = tid; i < n4; i += 256) reinterpret_cast<f32x4*>(s_x)[i] = x4[i];
    __syncthreads();
    const int t16 = tid & 15, plane = tid >> 4;
    const int64_t row = (int64_t)blockIdx.x * 16 + plane;
    const float* xr = s_x + plane * pp + kHalo;
    float s = 0.f;
    if (in_scale) {
        const int ch = (int)(row % c);
        const float sc = in_scale[ch], sf = in_shift[ch];
        for (int i = t16; i < p; i += 16) s += fmaxf(fmaf(xr[i], sc, sf), 0.f);
    } else {
        for (int i = t16; i < p; i += 16) s += xr[i];
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (t16 == 0) pooled[row * (1 + 2 * kHalo) + kHalo] = s / (float)p;
}

static int launch_plane_mean(const float* x, float* pooled, int64_t rows, int p, int pp, const float* in_scale, const float* in_shift, int c, hipStream_t s) {
    if (rows % 16 == 0 && (16 * pp) % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (size_t)16 * pp * sizeof(float) <= 64 * 1024 &&
        tune_get(TCR_TUNE_DW_FWD) != 1) {
        hipLaunchKernelGGL(plane_mean_block_kernel, dim3((unsigned)(rows / 16)), dim3(256), (size_t)16 * pp * sizeof(float), s, x, pooled, p, pp, in_scale, in_shift, c);
        return check_launch("plane_mean_block_kernel");
    }
    hipLaunchKernelGGL(plane_mean_kernel, dim3((unsigned)ceil_div64(rows, 16)), dim3(256), 0, s, x, pooled, rows, p, pp, in_scale, in_shift, c);
    return check_launch("plane_mean_kernel");
}

struct DsLayer {
    std::string scope;
    bool separable;
    int cin, cout, kh, kw, sh, sw, h_in, w_in, oh, ow, pad_t, pad_l;
    // offsets (floats): trainable arena / stat arena / scale-shift workspace slot
    int64_t w_off, b_off, beta_off, mean_off, var_off, ss_off;             // conv or depthwise
    int64_t pw_off, pb_off, pbeta_off, pmean_off, pvar_off, pss_off;       // pointwise (separable only)
};

}  // namespace tcr

using namespace tcr;

struct tcr_dscnn {
    tcr_dscnn_cfg cfg;
    std::vector<DsLayer> layers;
    int64_t fcw_off, fcb_off;
    int64_t param_floats, stat_floats, ss_floats;
    int c_pad;
    std::vector<tcr_tensor_info> tensors;
    // backward: filter-gradient kernels on a second stream (see tcr_net in net.cpp); the gradient buffers rotate through a pool of four for it
    mutable hipStream_t side = nullptr;
    mutable hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_done[4] = {nullptr, nullptr, nullptr, nullptr};
    mutable bool ev_rec[4] = {false, false, false, false};      // ev_done[i] was recorded in the current backward
    ~tcr_dscnn() {
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        for (hipEvent_t e : ev_done) if (e) (void)hipEventDestroy(e);
    }
};

namespace tcr {
static int64_t ds_align(int64_t v) { return (v + 63) / 64 * 64; }
}

extern "C" int tcr_dscnn_create(const tcr_dscnn_cfg* cfg, tcr_dscnn** out) {
    TCR_REQUIRE(cfg && out, "tcr_dscnn_create: null argument");
    TCR_REQUIRE(cfg->depth > 0 && cfg->depth % 4 == 0, "tcr_dscnn_create: depth %d must be a positive multiple of 4", cfg->depth);
    TCR_REQUIRE(cfg->n_separable >= 1 && cfg->n_separable <= 8, "tcr_dscnn_create: n_separable %d out of range", cfg->n_separable);
    TCR_REQUIRE(cfg->h_in > 0 && cfg->w_in > 0 && cfg->num_classes > 0 && cfg->num_classes + 2 <= 48, "tcr_dscnn_create: bad shape");
    TCR_REQUIRE(cfg->conv1_kh >= 1 && cfg->conv1_kh <= 16 && cfg->conv1_kw == 4, "tcr_dscnn_create: conv_1 kernel must be kh x 4 (got %d x %d)", cfg->conv1_kh, cfg->conv1_kw);
    tcr_dscnn* net = new tcr_dscnn();
    net->cfg = *cfg;
    net->c_pad = (int)ds_align(cfg->depth);
    int64_t o = 0, so = 0, ss = 0;
    auto info = [&](const std::string& name, int kind, int arena, int64_t off, int64_t size, std::vector<int> shape) {
        tcr_tensor_info ti;
        std::memset(&ti, 0, sizeof(ti));
        std::snprintf(ti.name, sizeof(ti.name), "DSCNN/%s", name.c_str());
        ti.kind = kind; ti.arena = arena; ti.offset = off; ti.size = size; ti.rank = (int)shape.size();
        for (size_t i = 0; i < shape.size() && i < 4; ++i) ti.shape[i] = shape[i];
        net->tensors.push_back(ti);
    };
    auto take = [&](int64_t n) { int64_t at = o; o = ds_align(o + n + 64); return at; };
    auto take_stat = [&](int64_t n) { int64_t at = so; so += ds_align(n); return at; };
    auto bn = [&](const std::string& prefix, int c, int64_t* beta, int64_t* mean, int64_t* var, int64_t* slot) {
        *beta = take(c); info(prefix + "/beta", TCR_BETA, 0, *beta, c, {c});
        *mean = take_stat(c); info(prefix + "/moving_mean", TCR_MOVING_MEAN, 1, *mean, c, {c});
        *var = take_stat(c); info(prefix + "/moving_variance", TCR_MOVING_VAR, 1, *var, c, {c});
        *slot = ss; ss += 2 * net->c_pad;
    };
    int h = cfg->h_in, w = cfg->w_in, c = 1;
    for (int li = 0; li <= cfg->n_separable; ++li) {
        DsLayer l;
        l.separable = li > 0;
        l.scope = li == 0 ? "conv_1" : "conv_ds_" + std::to_string(li);
        l.cin = c; l.cout = cfg->depth;
        l.kh = li == 0 ? cfg->conv1_kh : 3; l.kw = li == 0 ? cfg->conv1_kw : 3;
        l.sh = li == 0 ? cfg->conv1_sh : (li == 1 ? cfg->ds1_sh : 1);
        l.sw = li == 0 ? cfg->conv1_sw : (li == 1 ? cfg->ds1_sw : 1);
        l.h_in = h; l.w_in = w;
        int ph, pw2;
        same_pad(h, l.kh, l.sh, &l.oh, &l.pad_t, &ph);
        same_pad(w, l.kw, l.sw, &l.ow, &l.pad_l, &pw2);
        if (!l.separable) {
            l.w_off = take((int64_t)l.kh * l.kw * l.cin * l.cout);
            info(l.scope + "/weights", TCR_WEIGHT, 0, l.w_off, (int64_t)l.kh * l.kw * l.cin * l.cout, {l.kh, l.kw, l.cin, l.cout});
            l.b_off = take(l.cout); info(l.scope + "/biases", TCR_BETA, 0, l.b_off, l.cout, {l.cout});
            bn(l.scope + "/batch_norm", l.cout, &l.beta_off, &l.mean_off, &l.var_off, &l.ss_off);
            l.pw_off = l.pb_off = l.pbeta_off = l.pmean_off = l.pvar_off = l.pss_off = -1;
        } else {
            l.w_off = take(9 * (int64_t)l.cin);
            info(l.scope + "/depthwise_conv/depthwise_weights", TCR_WEIGHT, 0, l.w_off, 9 * (int64_t)l.cin, {3, 3, l.cin, 1});
            l.b_off = take(l.cin); info(l.scope + "/depthwise_conv/biases", TCR_BETA, 0, l.b_off, l.cin, {l.cin});
            bn(l.scope + "/dw_batch_norm", l.cin, &l.beta_off, &l.mean_off, &l.var_off, &l.ss_off);
            l.pw_off = take((int64_t)l.cin * l.cout);
            info(l.scope + "/pointwise_conv/weights", TCR_WEIGHT, 0, l.pw_off, (int64_t)l.cin * l.cout, {1, 1, l.cin, l.cout});
            l.pb_off = take(l.cout); info(l.scope + "/pointwise_conv/biases", TCR_BETA, 0, l.pb_off, l.cout, {l.cout});
            bn(l.scope + "/pw_batch_norm", l.cout, &l.pbeta_off, &l.pmean_off, &l.pvar_off, &l.pss_off);
        }
        net->layers.push_back(l);
        h = l.oh; w = l.ow; c = l.cout;
    }
    net->fcw_off = take((int64_t)c * cfg->num_classes);
    info("fc1/weights", TCR_WEIGHT, 0, net->fcw_off, (int64_t)c * cfg->num_classes, {c, cfg->num_classes});
    net->fcb_off = take(cfg->num_classes);
    info("fc1/biases", TCR_BETA, 0, net->fcb_off, cfg->num_classes, {cfg->num_classes});
    net->param_floats = o; net->stat_floats = so; net->ss_floats = ss;
    *out = net;
    return TCR_OK;
}

extern "C" void tcr_dscnn_destroy(tcr_dscnn* net) { delete net; }
extern "C" int64_t tcr_dscnn_param_floats(const tcr_dscnn* net) { return net ? net->param_floats : 0; }
extern "C" int64_t tcr_dscnn_stat_floats(const tcr_dscnn* net) { return net ? net->stat_floats : 0; }
extern "C" int tcr_dscnn_num_tensors(const tcr_dscnn* net) { return net ? (int)net->tensors.size() : 0; }
extern "C" int tcr_dscnn_tensor_info(const tcr_dscnn* net, int index, tcr_tensor_info* out) {
    TCR_REQUIRE(net && out && index >= 0 && index < (int)net->tensors.size(), "tcr_dscnn_tensor_info: bad argument");
    *out = net->tensors[index];
    return TCR_OK;
}

namespace tcr {
static int64_t ds_act_floats(const tcr_dscnn& net, int batch) {
    int64_t m = 0;
    for (const DsLayer& l : net.layers) {
        const int64_t n = (int64_t)batch * l.cout * tcr_padded_len(l.oh * l.ow);
        m = n > m ? n : m;
    }
    return ds_align(m);
}
}  // namespace tcr

extern "C" size_t tcr_dscnn_workspace_bytes(const tcr_dscnn* net, int batch) {
    if (!net || batch <= 0) return 0;
    return (size_t)(ds_align(net->ss_floats) + 2 * ds_act_floats(*net, batch)) * sizeof(float);
}

extern "C" int tcr_dscnn_forward_infer(const tcr_dscnn* net, const float* params, const float* stats, const float* feat,
                                       int batch, void* workspace, size_t workspace_bytes, float* logits, float* probs, void* stream) {
    TCR_REQUIRE(net && params && stats && feat && workspace && logits && probs, "tcr_dscnn_forward_infer: null argument");
    TCR_REQUIRE(batch > 0, "tcr_dscnn_forward_infer: batch must be positive (got %d)", batch);
    if (tcr_dscnn_workspace_bytes(net, batch) > workspace_bytes) {
        set_error("tcr_dscnn_forward_infer: workspace %zu bytes < required %zu", workspace_bytes, tcr_dscnn_workspace_bytes(net, batch));
        return TCR_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* ss = static_cast<float*>(workspace);
    float* buf[2] = {ss + ds_align(net->ss_floats), ss + ds_align(net->ss_floats) + ds_act_floats(*net, batch)};
    const int cp = net->c_pad;

    // conv bias + eval-mode BN (no scale) -> per-channel scale / shift, all layers in one launch
    BnFoldArgs f;
    std::memset(&f, 0, sizeof(f));
    f.params = params; f.stats = stats; f.out = ss; f.eps = net->cfg.bn_eps;
    auto add_fold = [&](int c, int64_t beta, int64_t mean, int64_t var, int64_t bias, int64_t slot) {
        f.c[f.n] = c; f.c_pad[f.n] = cp; f.gamma_off[f.n] = -1; f.beta_off[f.n] = beta; f.mean_off[f.n] = mean;
        f.var_off[f.n] = var; f.bias_off[f.n] = bias; f.out_off[f.n] = slot; ++f.n;
    };
    for (const DsLayer& l : net->layers) {
        add_fold(l.separable ? l.cin : l.cout, l.beta_off, l.mean_off, l.var_off, l.b_off, l.ss_off);
        if (l.separable) add_fold(l.cout, l.pbeta_off, l.pmean_off, l.pvar_off, l.pb_off, l.pss_off);
    }
    TCR_TRY(launch_bn_fold(f, s));

    int cur = 0;
    bool dw_done = false;                   // the first depthwise layer ran inside the fused conv_1 kernel
    for (size_t li = 0; li < net->layers.size(); ++li) {
        const DsLayer& l = net->layers[li];
        const int P = l.oh * l.ow, pp = tcr_padded_len(P);
        auto dw_args = [&](const DsLayer& dl, const float* x, float* y) {
            DsDwArgs d;
            d.x = x; d.w = params + dl.w_off; d.scale = ss + dl.ss_off; d.shift = ss + dl.ss_off + cp; d.y = y;
            d.total = (int64_t)batch * dl.cin * dl.oh * dl.ow; d.c = dl.cin; d.h_in = dl.h_in; d.w_in = dl.w_in;
            d.ppi = tcr_padded_len(dl.h_in * dl.w_in); d.oh = dl.oh; d.ow = dl.ow; d.ppo = tcr_padded_len(dl.oh * dl.ow);
            d.sh = dl.sh; d.sw = dl.sw; d.pad_t = dl.pad_t; d.pad_l = dl.pad_l; d.relu = 1;
            return d;
        };
        if (!l.separable) {
            DsConv1Args a;
            a.feat = feat; a.w = params + l.w_off; a.scale = ss + l.ss_off; a.shift = ss + l.ss_off + cp; a.y = buf[cur];
            a.npos = batch * P; a.cout = l.cout; a.h_in = l.h_in; a.w_in = l.w_in; a.tp_in = tcr_padded_len(l.h_in);
            a.oh = l.oh; a.ow = l.ow; a.pp = pp; a.kh = l.kh; a.sh = l.sh; a.sw = l.sw; a.pad_t = l.pad_t; a.pad_l = l.pad_l; a.relu = 1;
            int fused = 1;
            if (li + 1 < net->layers.size() && net->layers[li + 1].separable && tune_get(TCR_TUNE_CONV_B) != 3) {
                fused = launch_dscnn_conv1_dw(a, dw_args(net->layers[li + 1], nullptr, buf[cur ^ 1]), batch, s);
                if (fused < 0) return fused;
            }
            if (fused == 1) TCR_TRY(launch_dscnn_conv1(a, s));
            else dw_done = true;
        } else {
            if (!dw_done) TCR_TRY(launch_dscnn_depthwise(dw_args(l, buf[cur], buf[cur ^ 1]), batch, s));
            dw_done = false;
            Conv1x1Args c1;
            c1.x = buf[cur ^ 1]; c1.w = params + l.pw_off; c1.y = buf[cur]; c1.scale = ss + l.pss_off; c1.shift = ss + l.pss_off + cp;
            c1.npos = batch * P; c1.cin = l.cin; c1.cout = l.cout; c1.tpi = pp; c1.tout = P; c1.tpo = pp; c1.stride = 1; c1.relu = 1;
            TCR_TRY(launch_conv1x1(c1, MF_AFFINE, s));
        }
    }
    const DsLayer& last = net->layers.back();
    const int P = last.oh * last.ow;
    const int64_t rows = (int64_t)batch * last.cout;
    TCR_TRY(launch_plane_mean(buf[cur], buf[cur ^ 1], rows, P, tcr_padded_len(P), nullptr, nullptr, 1, s));
    HeadArgs h;
    std::memset(&h, 0, sizeof(h));
    h.feat = buf[cur ^ 1]; h.wfc = params + net->fcw_off; h.wfc2 = nullptr; h.bias = params + net->fcb_off;
    h.logits = logits; h.probs = probs; h.ranges = nullptr;
    h.batch = batch; h.c = last.cout; h.nc = net->cfg.num_classes; h.t = 1; h.tp = tcr_padded_len(1);
    h.keep_prob = 1.0f; h.inv_global_batch = 1.0f;
    return launch_head_fwd(h, false, s);
}

// ---------------------------------------------------------------------------------------------------------------
// Training (is_training=True graph of DSCNN() + tf.gradients, helper/trainer.py:199-222).
// A "unit" is one convolution followed by its BN(no scale)+ReLU: conv_1, then (depthwise, pointwise) per block.
// Train-mode forward keeps, per unit, the raw conv output (+bias) and the activation; backward walks the units in
// reverse: BN backward (bn.hip) -> filter gradient -> data gradient.
// The conv biases feed a train-mode BN, which subtracts the batch mean: their gradient is analytically zero (TF
// computes round-off noise there) and is written as exactly 0.
// ---------------------------------------------------------------------------------------------------------------
namespace tcr {

enum DsUnitKind { DS_CONV1 = 0, DS_DW = 1, DS_PW = 2 };

struct DsUnit {
    int kind, layer;
    int c, P;                   // channels / output positions of this unit
    int64_t w_off, b_off, beta_off, mean_off, var_off, ss_off;
};

static std::vector<DsUnit> ds_units(const tcr_dscnn& net) {
    std::vector<DsUnit> u;
    for (size_t li = 0; li < net.layers.size(); ++li) {
        const DsLayer& l = net.layers[li];
        const int P = l.oh * l.ow;
        if (!l.separable) {
            u.push_back({DS_CONV1, (int)li, l.cout, P, l.w_off, l.b_off, l.beta_off, l.mean_off, l.var_off, l.ss_off});
        } else {
            u.push_back({DS_DW, (int)li, l.cin, P, l.w_off, l.b_off, l.beta_off, l.mean_off, l.var_off, l.ss_off});
            u.push_back({DS_PW, (int)li, l.cout, P, l.pw_off, l.pb_off, l.pbeta_off, l.pmean_off, l.pvar_off, l.pss_off});
        }
    }
    return u;
}

// "Lazy" training (the default wherever every consumer has the form): a unit's normalised activation relu(raw * scale + shift) is
// never written -- its consumers (next conv, both filter-gradient kernels, the pooling) apply the affine + ReLU to the RAW conv output
// as they stage / read it --, BN batch statistics come from the conv epilogues (EpiSums) and the backward sums of a unit from the
// epilogue of the data-gradient kernel that produces its activation gradient: 5 of the ~14 tensor passes per unit disappear
// (statistics reduce, normalise read + write, backward reduce's two reads).  TCR_TUNE_DS_TRAIN = 1: the materialising path.
static bool ds_lazy(const tcr_dscnn& net) {
    if (tune_get(TCR_TUNE_DS_TRAIN) == 1) return false;
    for (const DsLayer& l : net.layers) {
        if (!l.separable) continue;
        const int pp = tcr_padded_len(l.oh * l.ow);
        if (!conv1x1_lds_covers(l.cin, l.cout) || !conv1x1_lds_covers(l.cout, l.cin) || !pw_wgrad_lds_covers(l.cin, l.cout, pp)) return false;
        if (!dscnn_depthwise_lds_covers(l.oh, l.ow, l.sh, l.sw) || !dscnn_dw_dgrad_lds_covers(l.h_in, l.w_in, l.pad_t, l.pad_l)) return false;
    }
    return true;
}

// partial rows of unit u's forward statistics (lazy: from its conv's epilogue)
static int ds_fwd_rows(const DsUnit& u, int batch, bool lazy) {
    if (!lazy) return chan_reduce_launch_chunks(batch * u.P, u.P);
    return u.kind == DS_PW ? conv1x1_sum_rows(batch * u.P) : (u.kind == DS_DW ? batch : dscnn_conv1_sum_rows(batch * u.P));
}

// partial rows of unit ui's backward sums (lazy: from the epilogue of the data gradient of unit ui + 1; the last unit's come from
// the reduction kernel -- its activation gradient is the broadcast pooled one)
static int ds_bwd_rows(const std::vector<DsUnit>& units, int ui, int batch, bool lazy) {
    const DsUnit& u = units[ui];
    if (!lazy || ui + 1 == (int)units.size()) return chan_reduce_launch_chunks(batch * u.P, u.P);
    return units[ui + 1].kind == DS_PW ? conv1x1_sum_rows(batch * u.P) : batch;
}

struct DsTrainWs {
    int64_t ss, kcoef, partial, sums, pooled, dropped, dscale, dlogits, loss_utt, dpool, fc_partial, scratch, wt, gbuf[4], total;
    std::vector<int64_t> raw, act, mean, invstd;
};

static DsTrainWs ds_carve(const tcr_dscnn& net, int batch) {
    DsTrainWs w;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t at = o; o += ds_align(n); return at; };
    const std::vector<DsUnit> units = ds_units(net);
    const int cp = net.c_pad, nc = net.cfg.num_classes;
    const int cl = net.layers.back().cout;
    w.ss = take(net.ss_floats);
    w.kcoef = take((int64_t)units.size() * 3 * cp);     // k1..k3 per unit (read by the side stream's kernels after the main stream has moved on)
    w.sums = take(2 * 2 * (int64_t)cp);         // doubles: cross-replica BN hand-off
    int maxpos = 0;
    int64_t max_act = 0, scratch = 0;
    for (const DsUnit& u : units) {
        const int64_t n = (int64_t)batch * u.c * tcr_padded_len(u.P);
        w.raw.push_back(take(n));
        w.act.push_back(take(n));
        w.mean.push_back(take(cp));
        w.invstd.push_back(take(cp));
        max_act = n > max_act ? n : max_act;
        maxpos = batch * u.P > maxpos ? batch * u.P : maxpos;
        const DsLayer& l = net.layers[u.layer];
        int64_t sc = 0;
        if (u.kind == DS_CONV1) sc = (int64_t)dscnn_conv1_wgrad_partial_floats(batch, l.kh, l.cout);
        else if (u.kind == DS_DW) sc = (int64_t)dscnn_dw_wgrad_partial_floats(batch, u.c);
        else sc = (int64_t)wgrad_partial_floats(1, l.cin, l.cout, batch);
        scratch = sc > scratch ? sc : scratch;
    }
    int64_t rows = chan_reduce_chunks(maxpos);          // (also covers the epilogue sums: a row per utterance / per 64 or 256 positions)
    rows = std::max<int64_t>(rows, std::max<int64_t>(batch, conv1x1_sum_rows(maxpos)));
    w.partial = take(rows * 2 * cp);
    w.pooled = take((int64_t)batch * cl * tcr_padded_len(1));
    w.dropped = take((int64_t)batch * cl);
    w.dscale = take((int64_t)batch * cl);
    w.dlogits = take((int64_t)batch * nc);
    w.loss_utt = take(batch);
    w.dpool = take((int64_t)batch * cl);
    w.fc_partial = take((int64_t)fc_wgrad_chunks(batch) * cl * nc);
    w.scratch = take(scratch);
    w.wt = take((int64_t)net.cfg.depth * net.cfg.depth);
    for (int i = 0; i < 4; ++i) w.gbuf[i] = take(max_act + 64);       // (+ pad: 16-byte loads of a row's last positions)
    w.total = o;
    return w;
}

}  // namespace tcr

extern "C" size_t tcr_dscnn_train_workspace_bytes(const tcr_dscnn* net, int batch) {
    if (!net || batch <= 0) return 0;
    return (size_t)ds_carve(*net, batch).total * sizeof(float);
}

// Stages [stage_begin, stage_end) of the train-mode forward.  Stage u (u < units): finalize + normalise unit u-1, then conv + statistics
// of unit u; the last stage finalizes the last unit and runs the head.  sync != 0 (cross-replica BN): a stage ends with the unit's
// 2*C float64 sums in the hand-off buffer, which the host all-reduces before the next stage, and statistics span global_batch.
static int ds_forward_train_stages(const tcr_dscnn* net, const float* params, float* stats, const float* feat, const float* labels,
                                   int batch, int global_batch, int sync, float label_smoothing, void* workspace, size_t workspace_bytes,
                                   float* logits, float* probs, float* loss_out, int stage_begin, int stage_end, void* stream) {
    TCR_REQUIRE(net && params && stats && feat && labels && workspace && logits && probs && loss_out, "tcr_dscnn_forward_train: null argument");
    TCR_REQUIRE(batch > 0 && global_batch >= batch, "tcr_dscnn_forward_train: batch %d / global_batch %d", batch, global_batch);
    const DsTrainWs w = ds_carve(*net, batch);
    if ((size_t)w.total * sizeof(float) > workspace_bytes) {
        set_error("tcr_dscnn_forward_train: workspace %zu bytes < required %zu", workspace_bytes, (size_t)w.total * sizeof(float));
        return TCR_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* base = static_cast<float*>(workspace);
    const std::vector<DsUnit> units = ds_units(*net);
    const int cp = net->c_pad;
    const int nu = (int)units.size();
    TCR_REQUIRE(stage_begin >= 0 && stage_end <= nu + 1 && stage_begin < stage_end, "tcr_dscnn_forward_train: bad stage range [%d, %d)", stage_begin, stage_end);
    const double bn_batch = sync ? (double)global_batch : (double)batch;
    const bool lazy = ds_lazy(*net);
    // finalize + normalise of unit ui (its partial rows / cross-replica sums are ready)
    auto post = [&](int ui) -> int {
        const DsUnit& u = units[ui];
        const int pp = tcr_padded_len(u.P);
        float* ss = base + w.ss + u.ss_off;
        BnFinalizeArgs f;
        f.partial = base + w.partial; f.nchunk = sync ? 0 : ds_fwd_rows(u, batch, lazy);
        f.sums = reinterpret_cast<const double*>(base + w.sums);
        f.gamma = nullptr; f.beta = params + u.beta_off;
        f.moving_mean = stats + u.mean_off; f.moving_var = stats + u.var_off;
        f.scale = ss; f.shift = ss + cp; f.mean = base + w.mean[ui]; f.invstd = base + w.invstd[ui];
        f.c = u.c; f.count = bn_batch * (double)u.P; f.decay = net->cfg.bn_decay; f.eps = net->cfg.bn_eps;
        TCR_TRY(launch_bn_finalize(f, s));
        if (lazy) return TCR_OK;                    // (consumers apply scale / shift + ReLU to the raw output themselves)
        BnApplyArgs ap;
        ap.y = base + w.raw[ui]; ap.scale = ss; ap.shift = ss + cp; ap.res = nullptr; ap.out = base + w.act[ui];
        ap.total = (int64_t)batch * u.c * pp; ap.c = u.c; ap.t = u.P; ap.tp = pp; ap.relu = 1;
        return launch_bn_apply(ap, s);
    };
    for (int ui = stage_begin; ui < stage_end && ui < nu; ++ui) {
        if (ui > 0) TCR_TRY(post(ui - 1));
        const float* x = ui > 0 ? base + (lazy ? w.raw[ui - 1] : w.act[ui - 1]) : nullptr;      // activation feeding the unit (nullptr: the features)
        const float* in_scale = (lazy && ui > 0) ? base + w.ss + units[ui - 1].ss_off : nullptr;    // lazy: the producing unit's raw output + its BN affine
        const float* in_shift = in_scale ? in_scale + cp : nullptr;
        const DsUnit& u = units[ui];
        const DsLayer& l = net->layers[u.layer];
        const int pp = tcr_padded_len(u.P);
        float* raw = base + w.raw[ui];
        EpiSums es;
        if (lazy) es.partial = base + w.partial;
        if (u.kind == DS_CONV1) {
            DsConv1Args a;
            a.feat = feat; a.w = params + u.w_off; a.scale = nullptr; a.shift = params + u.b_off; a.y = raw;
            a.npos = batch * u.P; a.cout = l.cout; a.h_in = l.h_in; a.w_in = l.w_in; a.tp_in = tcr_padded_len(l.h_in);
            a.oh = l.oh; a.ow = l.ow; a.pp = pp; a.kh = l.kh; a.sh = l.sh; a.sw = l.sw; a.pad_t = l.pad_t; a.pad_l = l.pad_l; a.relu = 0;
            a.sums = es;
            TCR_TRY(launch_dscnn_conv1(a, s));
        } else if (u.kind == DS_DW) {
            DsDwArgs d;
            d.x = x; d.w = params + u.w_off; d.scale = nullptr; d.shift = params + u.b_off; d.y = raw;
            d.total = (int64_t)batch * l.cin * u.P; d.c = l.cin; d.h_in = l.h_in; d.w_in = l.w_in;
            d.ppi = tcr_padded_len(l.h_in * l.w_in); d.oh = l.oh; d.ow = l.ow; d.ppo = pp;
            d.sh = l.sh; d.sw = l.sw; d.pad_t = l.pad_t; d.pad_l = l.pad_l; d.relu = 0;
            d.in_scale = in_scale; d.in_shift = in_shift; d.sums = es;
            TCR_TRY(launch_dscnn_depthwise(d, batch, s));
        } else {
            Conv1x1Args c1;
            c1.x = x; c1.w = params + u.w_off; c1.y = raw; c1.scale = nullptr; c1.shift = params + u.b_off;
            c1.npos = batch * u.P; c1.cin = l.cin; c1.cout = l.cout; c1.tpi = pp; c1.tout = u.P; c1.tpo = pp; c1.stride = 1; c1.relu = 0;
            c1.in_scale = in_scale; c1.in_shift = in_shift; c1.sums = es;
            TCR_TRY(launch_conv1x1(c1, MF_AFFINE, s));
        }
        // batch statistics of the raw output (the finalize + normalise run at the start of the next stage)
        int nchunk = ds_fwd_rows(u, batch, lazy);
        if (!lazy) {
            ChanReduceArgs r;
            std::memset(&r, 0, sizeof(r));
            r.y = raw; r.partial = base + w.partial; r.npos = batch * u.P; r.c = u.c; r.t = u.P; r.tp = pp;
            TCR_TRY(launch_chan_reduce(0, r, &nchunk, s));
        }
        if (sync) TCR_TRY(launch_chan_sums(base + w.partial, nchunk, u.c, reinterpret_cast<double*>(base + w.sums), s));
    }
    if (stage_end <= nu) return TCR_OK;
    TCR_TRY(post(nu - 1));
    const float* x = base + (lazy ? w.raw[nu - 1] : w.act[nu - 1]);
    const float* pool_scale = lazy ? base + w.ss + units[nu - 1].ss_off : nullptr;
    const DsLayer& last = net->layers.back();
    const int P = last.oh * last.ow;
    const int64_t rows = (int64_t)batch * last.cout;
    TCR_TRY(launch_plane_mean(x, base + w.pooled, rows, P, tcr_padded_len(P), pool_scale, pool_scale ? pool_scale + cp : (const float*)nullptr, last.cout, s));
    HeadArgs h;
    std::memset(&h, 0, sizeof(h));
    h.feat = base + w.pooled; h.wfc = params + net->fcw_off; h.wfc2 = nullptr; h.bias = params + net->fcb_off;
    h.labels = labels; h.logits = logits; h.probs = probs; h.ranges = nullptr;
    h.dropped = base + w.dropped; h.dscale = base + w.dscale; h.dlogits = base + w.dlogits; h.loss_utt = base + w.loss_utt;
    h.batch = batch; h.c = last.cout; h.nc = net->cfg.num_classes; h.t = 1; h.tp = tcr_padded_len(1);
    h.keep_prob = 1.0f;                                     // slim.dropout is arg-scoped but never called (ds_cnn.py:89-101)
    h.inv_global_batch = 1.0f / (float)global_batch; h.label_smoothing = label_smoothing;
    h.pool_scale = 1.0f / (float)P;
    TCR_TRY(launch_head_fwd(h, true, s));
    return launch_sum_vector(base + w.loss_utt, batch, loss_out, s);
}

extern "C" int tcr_dscnn_forward_train(const tcr_dscnn* net, const float* params, float* stats, const float* feat, const float* labels,
                                       int batch, int global_batch, float label_smoothing, void* workspace, size_t workspace_bytes,
                                       float* logits, float* probs, float* loss_out, void* stream) {
    const int nu = net ? (int)ds_units(*net).size() : 0;
    return ds_forward_train_stages(net, params, stats, feat, labels, batch, global_batch, 0, label_smoothing, workspace, workspace_bytes,
                                   logits, probs, loss_out, 0, nu + 1, stream);
}

extern "C" int tcr_dscnn_num_stages(const tcr_dscnn* net) { return net ? (int)ds_units(*net).size() + 1 : 0; }

extern "C" int tcr_dscnn_forward_train_stage(const tcr_dscnn* net, const float* params, float* stats, const float* feat, const float* labels,
                                             int batch, int global_batch, float label_smoothing, void* workspace, size_t workspace_bytes,
                                             float* logits, float* probs, float* loss_out, int stage, void* stream) {
    return ds_forward_train_stages(net, params, stats, feat, labels, batch, global_batch, 1, label_smoothing, workspace, workspace_bytes,
                                   logits, probs, loss_out, stage, stage + 1, stream);
}

extern "C" int tcr_dscnn_num_units(const tcr_dscnn* net) { return net ? (int)ds_units(*net).size() : 0; }

extern "C" int tcr_dscnn_unit_output(const tcr_dscnn* net, int unit, int batch, int64_t* offset, int* channels, int* positions, int* padded) {
    TCR_REQUIRE(net && offset && channels && positions && padded && batch > 0, "tcr_dscnn_unit_output: bad argument");
    const std::vector<DsUnit> units = ds_units(*net);
    TCR_REQUIRE(unit >= 0 && unit < (int)units.size(), "tcr_dscnn_unit_output: unit %d of %d", unit, (int)units.size());
    const DsTrainWs w = ds_carve(*net, batch);
    *offset = w.act[unit]; *channels = units[unit].c; *positions = units[unit].P; *padded = tcr_padded_len(units[unit].P);
    return TCR_OK;
}

extern "C" int tcr_dscnn_materialize_unit(const tcr_dscnn* net, int unit, int batch, void* workspace, size_t workspace_bytes, void* stream) {
    TCR_REQUIRE(net && workspace && batch > 0, "tcr_dscnn_materialize_unit: bad argument");
    const std::vector<DsUnit> units = ds_units(*net);
    TCR_REQUIRE(unit >= 0 && unit < (int)units.size(), "tcr_dscnn_materialize_unit: unit %d of %d", unit, (int)units.size());
    const DsTrainWs w = ds_carve(*net, batch);
    if ((size_t)w.total * sizeof(float) > workspace_bytes) { set_error("tcr_dscnn_materialize_unit: workspace too small"); return TCR_ERR_WORKSPACE; }
    float* base = static_cast<float*>(workspace);
    const DsUnit& u = units[unit];
    const int pp = tcr_padded_len(u.P);
    const float* ss = base + w.ss + u.ss_off;
    BnApplyArgs ap;
    ap.y = base + w.raw[unit]; ap.scale = ss; ap.shift = ss + net->c_pad; ap.res = nullptr; ap.out = base + w.act[unit];
    ap.total = (int64_t)batch * u.c * pp; ap.c = u.c; ap.t = u.P; ap.tp = pp; ap.relu = 1;
    return launch_bn_apply(ap, static_cast<hipStream_t>(stream));
}

extern "C" int tcr_dscnn_stage_sums(const tcr_dscnn* net, int backward, int stage, void* workspace, int batch, double** sums_dev,
                                    int64_t* n_doubles) {
    TCR_REQUIRE(net && workspace && sums_dev && n_doubles, "tcr_dscnn_stage_sums: null argument");
    const std::vector<DsUnit> units = ds_units(*net);
    const int nu = (int)units.size();
    TCR_REQUIRE(stage >= 0 && stage < nu, "tcr_dscnn_stage_sums: stage %d has no BN hand-off", stage);
    const DsTrainWs w = ds_carve(*net, batch);
    *sums_dev = reinterpret_cast<double*>(static_cast<float*>(workspace) + w.sums);
    *n_doubles = 2 * (int64_t)units[backward ? nu - 1 - stage : stage].c;
    return TCR_OK;
}

// Stages of backward: stage k handles unit nu-1-k: stage 0 = head + the last unit's sums; stage k > 0 = BN backward / filter + data
// gradients of unit nu-k, then the sums of unit nu-1-k; the last stage (k = nu) finishes unit 0.
static int ds_backward_stages(const tcr_dscnn* net, const float* params, const float* feat, int batch, int global_batch, int sync,
                              void* workspace, size_t workspace_bytes, float* grads, int stage_begin, int stage_end, void* stream) {
    TCR_REQUIRE(net && params && feat && workspace && grads, "tcr_dscnn_backward: null argument");
    TCR_REQUIRE(batch > 0 && global_batch >= batch, "tcr_dscnn_backward: batch %d / global_batch %d", batch, global_batch);
    const DsTrainWs w = ds_carve(*net, batch);
    if ((size_t)w.total * sizeof(float) > workspace_bytes) {
        set_error("tcr_dscnn_backward: workspace %zu bytes < required %zu", workspace_bytes, (size_t)w.total * sizeof(float));
        return TCR_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* base = static_cast<float*>(workspace);
    const std::vector<DsUnit> units = ds_units(*net);
    const int cp = net->c_pad, nc = net->cfg.num_classes;
    const int cl = net->layers.back().cout;
    const int nu = (int)units.size();
    TCR_REQUIRE(stage_begin >= 0 && stage_end <= nu + 1 && stage_begin < stage_end, "tcr_dscnn_backward: bad stage range [%d, %d)", stage_begin, stage_end);
    const double bn_batch = sync ? (double)global_batch : (double)batch;
    const bool lazy = ds_lazy(*net);
    hipStream_t side = s;
    if (tune_get(TCR_TUNE_WGRAD_STREAM) != 1) {
        if (!net->side) {
            bool ok = (net->side = shared_stream(0, s)) != nullptr &&         // (process-wide: see tcr::shared_stream)
                      hipEventCreateWithFlags(&net->ev_fork, internal_event_flags()) == hipSuccess &&
                      hipEventCreateWithFlags(&net->ev_join, internal_event_flags()) == hipSuccess;
            for (int i = 0; i < 4 && ok; ++i) ok = hipEventCreateWithFlags(&net->ev_done[i], internal_event_flags()) == hipSuccess;
            if (!ok) { net->side = nullptr; set_error("tcr_dscnn_backward: cannot create the filter-gradient stream"); return TCR_ERR_HIP; }
        }
        side = net->side;
    }
    if (stage_begin == 0) {
        // zero the arena: padding and the conv biases (exactly-zero gradient, see above)
        if (hipMemsetAsync(grads, 0, (size_t)net->param_floats * sizeof(float), s) != hipSuccess) {
            set_error("tcr_dscnn_backward: hipMemsetAsync failed");
            return TCR_ERR_HIP;
        }
        // the classifier's own gradients feed nothing below: on the side stream (0.2 ms of small kernels out of the main chain)
        if (side != s && (hipEventRecord(net->ev_fork, s) != hipSuccess || hipStreamWaitEvent(side, net->ev_fork, 0) != hipSuccess)) {
            set_error("tcr_dscnn_backward: stream fork failed");
            return TCR_ERR_HIP;
        }
        TCR_TRY(launch_fc_wgrad(base + w.dropped, base + w.dlogits, base + w.fc_partial, grads + net->fcw_off, batch, cl, nc, side));
        TCR_TRY(launch_bias_grad(base + w.dlogits, batch, nc, grads + net->fcb_off, side));
        TCR_TRY(launch_head_bwd(base + w.dlogits, params + net->fcw_off, base + w.dscale, base + w.dpool, batch, cl, nc, s));
    }

    // Filter gradients run on a second stream, overlapped with the BN-backward / data-gradient chain of the units below.
    // Gradient buffers come from a pool of four, handed out round-robin: a unit's materialised dy (`D`), the activation gradient
    // its data-gradient kernel writes for the unit below (`gout`).  The side stream reads D -- or, for units whose BN backward is
    // applied on the fly (BnBwdFly: conv_1 in lazy mode; the depthwise units behind knob 3), the incoming activation gradient `gin` --, so the main
    // stream waits for the buffer's last side-stream reader (ev_done) before it writes a buffer again.  The hand-out order is a
    // function of the unit list only, so a staged run (one host call per stage) recomputes it.
    if (stage_begin == 0) for (bool& r : net->ev_rec) r = false;
    // on the fly: conv_1 (its dy has one reader, and the step ends with it: 0.69 ms of bn_bwd_apply gone from the tail).  The
    // depthwise units measured SLOWER that way (their bn_bwd_apply pass runs in the shadow of the pointwise filter gradient on the
    // side stream; computing dy in the data-gradient kernel lengthens the main chain: 2.03 vs 1.90 ms per block) -- knob 3 turns it on
    const int fly_knob = tune_get(TCR_TUNE_DS_TRAIN);
    auto unit_fly = [&](int ui) {
        if (!lazy || fly_knob == 2 || ui == nu - 1) return false;
        return units[ui].kind == DS_CONV1 || (fly_knob == 3 && units[ui].kind == DS_DW);
    };
    // knob 4 -- pointwise units (all but the last, whose activation gradient is the broadcast pooled one): the data-gradient kernel applies
    // the BN backward while it stages dy and writes dy for the filter gradient (conv1x1_lds_kernel MODE 3), no bn_bwd_apply pass.  Measured
    // SLOWER (DS-CNN-L step 16.33 vs 16.10 ms, M 7.72 vs 7.64): the filter gradient then starts behind the data gradient instead of
    // beside it, and the two matrix-pipe kernels overlapping was worth more than the 0.2 ms pass.  Not the default.
    auto unit_fused_apply = [&](int ui) { return lazy && fly_knob == 4 && units[ui].kind == DS_PW && ui != nu - 1; };
    // buffer plan: stage st (unit nu - st) reads gin[st], materialises into dbuf[st] (-1: on the fly), writes gout[st] (-1: conv_1)
    std::vector<int> gin(nu + 1, -1), dbuf(nu + 1, -1), gout(nu + 1, -1);
    {
        int next = 0, cur = -1;
        for (int st = 1; st <= nu; ++st) {
            const int ui = nu - st;
            gin[st] = cur;
            if (!unit_fly(ui)) dbuf[st] = next++ % 4;
            if (ui > 0) { gout[st] = next++ % 4; cur = gout[st]; }
        }
    }
    // the main stream is about to WRITE pool buffer i: its last side-stream reader must be done
    auto claim = [&](int i) -> int {
        if (side != s && net->ev_rec[i] && hipStreamWaitEvent(s, net->ev_done[i], 0) != hipSuccess) { set_error("tcr_dscnn_backward: stream wait failed"); return TCR_ERR_HIP; }
        return TCR_OK;
    };
    // per-channel sums of unit ui's BN backward by the reduction kernel (materialising path; lazy: only the last unit, whose
    // activation gradient is the broadcast pooled one)
    auto pre = [&](int ui, const float* da) -> int {
        const DsUnit& u = units[ui];
        const int pp = tcr_padded_len(u.P);
        ChanReduceArgs r;
        std::memset(&r, 0, sizeof(r));
        r.y = base + w.raw[ui]; r.da = da; r.m1 = base + w.act[ui]; r.m2 = nullptr;
        if (lazy || tune_get(TCR_TUNE_BWD_MASK) != 1) {     // the unit's own ReLU mask from its raw output (one tensor read less; lazy: there is no stored activation)
            r.m1 = nullptr; r.self_scale = base + w.ss + u.ss_off; r.self_shift = r.self_scale + cp;
        }
        r.mean = base + w.mean[ui]; r.invstd = base + w.invstd[ui];
        r.partial = base + w.partial; r.npos = batch * u.P; r.c = u.c; r.t = u.P; r.tp = pp; r.bcast = ui == nu - 1 ? 1 : 0;
        int nchunk = 0;
        TCR_TRY(launch_chan_reduce(1, r, &nchunk, s));
        if (sync) TCR_TRY(launch_chan_sums(base + w.partial, nchunk, u.c, reinterpret_cast<double*>(base + w.sums), s));
        return TCR_OK;
    };
    if (stage_begin == 0) TCR_TRY(pre(nu - 1, base + w.dpool));
    for (int st = stage_begin > 0 ? stage_begin : 1; st < stage_end; ++st) {
        const int ui = nu - st;                     // the unit whose sums the previous stage produced
        const DsUnit& u = units[ui];
        const DsLayer& l = net->layers[u.layer];
        const int pp = tcr_padded_len(u.P);
        const float* raw = base + w.raw[ui];
        const float* act = base + w.act[ui];
        const float* da = ui == nu - 1 ? base + w.dpool : base + w.gbuf[gin[st]];
        const int bcast = ui == nu - 1 ? 1 : 0;
        const bool fly = unit_fly(ui);
        float* kc = base + w.kcoef + (int64_t)ui * 3 * cp;
        BnBwdFinalizeArgs f;
        f.partial = base + w.partial; f.nchunk = sync ? 0 : ds_bwd_rows(units, ui, batch, lazy);
        f.sums = reinterpret_cast<const double*>(base + w.sums); f.gamma = nullptr; f.invstd = base + w.invstd[ui];
        f.dgamma = nullptr; f.dbeta = grads + u.beta_off;
        f.k1 = kc; f.k2 = kc + cp; f.k3 = kc + 2 * cp;
        f.c = u.c; f.count = bn_batch * (double)u.P; f.grad_scale = (float)((double)batch / bn_batch);
        TCR_TRY(launch_bn_bwd_finalize(f, s));
        const float* self_scale = base + w.ss + u.ss_off;
        const float* dz = nullptr;              // the unit's dy, materialised ...
        BnBwdFly bf;                            // ... or computed by its readers
        const bool fused_apply = unit_fused_apply(ui);
        if (fused_apply) {                      // (dy is written by the data-gradient kernel below)
            TCR_TRY(claim(dbuf[st]));
            dz = base + w.gbuf[dbuf[st]];
            bf.raw = raw; bf.mean = base + w.mean[ui]; bf.k1 = f.k1; bf.k2 = f.k2; bf.k3 = f.k3;
            bf.self_scale = self_scale; bf.self_shift = self_scale + cp;
        } else if (fly) {
            bf.da = da; bf.raw = raw; bf.mean = base + w.mean[ui]; bf.k1 = f.k1; bf.k2 = f.k2; bf.k3 = f.k3;
            bf.self_scale = self_scale; bf.self_shift = self_scale + cp;
        } else {
            TCR_TRY(claim(dbuf[st]));
            float* dzw = base + w.gbuf[dbuf[st]];
            BnBwdApplyArgs ap;
            ap.accumulate = 0;
            ap.y = raw; ap.da = da; ap.m1 = act; ap.m2 = nullptr; ap.mean = base + w.mean[ui];
            if (lazy || tune_get(TCR_TUNE_BWD_MASK) != 1) { ap.m1 = nullptr; ap.self_scale = self_scale; ap.self_shift = self_scale + cp; }
            ap.k1 = f.k1; ap.k2 = f.k2; ap.k3 = f.k3; ap.dy = dzw;
            ap.total = (int64_t)batch * u.c * pp; ap.c = u.c; ap.t = u.P; ap.tp = pp; ap.bcast = bcast;
            TCR_TRY(launch_bn_bwd_apply(ap, s));
            dz = dzw;
        }
        auto fork = [&]() -> int {
            if (side != s && (hipEventRecord(net->ev_fork, s) != hipSuccess || hipStreamWaitEvent(side, net->ev_fork, 0) != hipSuccess)) {
                set_error("tcr_dscnn_backward: stream fork failed");
                return TCR_ERR_HIP;
            }
            return TCR_OK;
        };
        if (!fused_apply) TCR_TRY(fork());
        float* ga = nullptr;
        if (ui > 0) { TCR_TRY(claim(gout[st])); ga = base + w.gbuf[gout[st]]; }
        const float* xin = ui > 0 ? base + (lazy ? w.raw[ui - 1] : w.act[ui - 1]) : nullptr;
        const float* x_scale = (lazy && ui > 0) ? base + w.ss + units[ui - 1].ss_off : nullptr;     // lazy: xin is a raw output + its BN affine
        const float* x_shift = x_scale ? x_scale + cp : nullptr;
        EpiSums es;             // lazy: the data gradient's epilogue takes the backward sums of unit ui - 1
        if (lazy && ui > 0) {
            es.partial = base + w.partial; es.raw = base + w.raw[ui - 1]; es.mean = base + w.mean[ui - 1]; es.invstd = base + w.invstd[ui - 1];
            es.self_scale = x_scale; es.self_shift = x_shift;
        }
        if (u.kind == DS_PW) {
            if (!fused_apply) TCR_TRY(launch_conv_wgrad(1, 1, 0, xin, dz, grads + u.w_off, base + w.scratch, batch, l.cin, l.cout, pp, u.P, pp, side, x_scale, x_shift));
            TCR_TRY(launch_transpose_weights(params + u.w_off, base + w.wt, 1, l.cin, l.cout, s));
            Conv1x1Args c1;
            c1.x = fused_apply ? da : dz; c1.w = base + w.wt; c1.y = ga; c1.scale = nullptr; c1.shift = nullptr;
            c1.npos = batch * u.P; c1.cin = l.cout; c1.cout = l.cin; c1.tpi = pp; c1.tout = u.P; c1.tpo = pp; c1.stride = 1; c1.relu = 0;
            c1.sums = es;
            if (fused_apply) { c1.fly = bf; c1.dy_out = const_cast<float*>(dz); }
            TCR_TRY(launch_conv1x1(c1, MF_RAW, s));
            if (fused_apply) {              // the filter gradient reads the dy the data gradient has just written
                TCR_TRY(fork());
                TCR_TRY(launch_conv_wgrad(1, 1, 0, xin, dz, grads + u.w_off, base + w.scratch, batch, l.cin, l.cout, pp, u.P, pp, side, x_scale, x_shift));
            }
        } else if (u.kind == DS_DW) {
            const int ppi = tcr_padded_len(l.h_in * l.w_in);
            DsDwWgradArgs g;
            g.x = xin; g.dz = dz; g.partial = base + w.scratch; g.batch = batch; g.c = u.c; g.h_in = l.h_in; g.w_in = l.w_in; g.ppi = ppi;
            g.oh = l.oh; g.ow = l.ow; g.ppo = pp; g.sh = l.sh; g.sw = l.sw; g.pad_t = l.pad_t; g.pad_l = l.pad_l; g.utt_per_block = 0;
            g.x_scale = x_scale; g.x_shift = x_shift; g.fly = bf;
            TCR_TRY(launch_dscnn_dw_wgrad(g, grads + u.w_off, side));
            DsDwBwdArgs d;
            d.dz = dz; d.w = params + u.w_off; d.dx = ga; d.planes = (int64_t)batch * u.c; d.c = u.c; d.h_in = l.h_in; d.w_in = l.w_in;
            d.ppi = ppi; d.oh = l.oh; d.ow = l.ow; d.ppo = pp; d.sh = l.sh; d.sw = l.sw; d.pad_t = l.pad_t; d.pad_l = l.pad_l;
            d.sums = es; d.fly = bf;
            TCR_TRY(launch_dscnn_dw_dgrad(d, s));
        } else {
            DsConv1WgradArgs g;
            g.feat = feat; g.dz = dz; g.partial = base + w.scratch; g.batch = batch; g.cout = l.cout;
            g.cout_pad = 0; g.taps = 0; g.taps_pad = 0; g.utt_per_block = 0;
            g.h_in = l.h_in; g.w_in = l.w_in; g.tp_in = tcr_padded_len(l.h_in); g.oh = l.oh; g.ow = l.ow; g.pp = pp;
            g.kh = l.kh; g.sh = l.sh; g.sw = l.sw; g.pad_t = l.pad_t; g.pad_l = l.pad_l; g.fly = bf;
            TCR_TRY(launch_dscnn_conv1_wgrad(g, grads + u.w_off, side));
        }
        // the buffer the side stream's kernel of this unit reads: its dy, or the incoming activation gradient
        const int side_reads = fly ? gin[st] : dbuf[st];
        if (side != s && side_reads >= 0) {
            if (hipEventRecord(net->ev_done[side_reads], side) != hipSuccess) { set_error("tcr_dscnn_backward: event record failed"); return TCR_ERR_HIP; }
            net->ev_rec[side_reads] = true;
        }
        if (ui > 0) {
            if (!lazy) TCR_TRY(pre(ui - 1, ga));
            else if (sync) TCR_TRY(launch_chan_sums(base + w.partial, ds_bwd_rows(units, ui - 1, batch, lazy), units[ui - 1].c, reinterpret_cast<double*>(base + w.sums), s));
        }
    }
    if (stage_end == nu + 1 && side != s && (hipEventRecord(net->ev_join, side) != hipSuccess || hipStreamWaitEvent(s, net->ev_join, 0) != hipSuccess)) {
        set_error("tcr_dscnn_backward: stream join failed");
        return TCR_ERR_HIP;
    }
    return TCR_OK;
}

extern "C" int tcr_dscnn_backward(const tcr_dscnn* net, const float* params, const float* feat, int batch,
                                  void* workspace, size_t workspace_bytes, float* grads, void* stream) {
    const int nu = net ? (int)ds_units(*net).size() : 0;
    return ds_backward_stages(net, params, feat, batch, batch, 0, workspace, workspace_bytes, grads, 0, nu + 1, stream);
}

extern "C" int tcr_dscnn_backward_stage(const tcr_dscnn* net, const float* params, const float* feat, int batch, int global_batch,
                                        void* workspace, size_t workspace_bytes, float* grads, int stage, void* stream) {
    return ds_backward_stages(net, params, feat, batch, global_batch, 1, workspace, workspace_bytes, grads, stage, stage + 1, stream);
}
