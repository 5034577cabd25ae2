// Dense contractions on the gfx950 matrix cores with the exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate, bitwise an fmaf chain):
//   * 1x1 "down" shortcut convolutions                    (audio_nets/tc_resnet.py:30-32)
//   * weight gradients of every convolution -- a [K*Cin] x [positions] x [Cout] contraction over the
//     whole batch (tf.gradients of tf.nn.conv2d wrt the filter, helper/trainer.py:205-211).
//
// Fragment layout of the 16x16x4 f32 MFMA (wave64): A[i = lane & 15][k = lane >> 4],
// B[k = lane >> 4][j = lane & 15], D[row = 4 * (lane >> 4) + reg][col = lane & 15].
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kernels.h"

namespace tcr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// D[row = co][col = position] = sum_ci W[ci][co] * x[b][ci][t * stride]
template <int MT, int EPI>
__global__ __launch_bounds__(256) void conv1x1_mfma_kernel(const Conv1x1Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int pos0 = (blockIdx.x * 4 + wave) * 64;
    const int cot0 = blockIdx.y * MT;

    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* xb[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int p = min(pos0 + nt * 16 + r, a.npos - 1);
        const int n = p / a.tout, t = p - n * a.tout;
        xb[nt] = a.x + (size_t)n * a.cin * a.tpi + kHalo + t * a.stride;
    }
    if ((a.cin & 3) == 0) {
        // Lean loop (every shape in use: Cin % 4 == 0): per-lane operand pointers advanced by constant strides, no per-load
        // predicates (output channels past Cout are clamped to a valid row that is never stored), next step's operands
        // in flight while the current 4 * MT MFMAs issue.
        const float* wp[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) wp[m] = a.w + (size_t)q * a.cout + min((cot0 + m) * 16 + r, a.cout - 1);
        const float* xq[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) xq[nt] = xb[nt] + (size_t)q * a.tpi;
        const int wstep = 4 * a.cout, xstep = 4 * a.tpi;
        const int nsteps = a.cin >> 2;
        float af[MT], bf[4];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = wp[m][0];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = xq[nt][0];
        for (int st = 0; st < nsteps; ++st) {
            const int nx = min(st + 1, nsteps - 1);         // (last step re-reads itself)
            float an[MT], bn[4];
#pragma unroll
            for (int m = 0; m < MT; ++m) an[m] = wp[m][nx * wstep];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bn[nt] = xq[nt][nx * xstep];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[nt], acc[m][nt], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m] = an[m];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bf[nt] = bn[nt];
        }
    } else
    for (int ci0 = 0; ci0 < a.cin; ci0 += 4) {
        const int ci = ci0 + q;
        const bool civ = ci < a.cin;
        float af[MT], bf[4];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co = (cot0 + m) * 16 + r;
            af[m] = (civ && co < a.cout) ? a.w[(size_t)ci * a.cout + co] : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = civ ? xb[nt][(size_t)ci * a.tpi] : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[nt], acc[m][nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int p = pos0 + nt * 16 + r;
        if (p >= a.npos) continue;
        const int n = p / a.tout, t = p - n * a.tout;
        float* yb = a.y + (size_t)n * a.cout * a.tpo + kHalo + t;       // (lean addressing: see conv_mfma_store)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = (cot0 + m) * 16 + q * 4 + reg;
                if (co >= a.cout) continue;
                float v = acc[m][nt][reg];
                if (EPI == MF_AFFINE) {
                    v = fmaf(v, a.scale ? a.scale[co] : 1.0f, a.shift[co]);    // (scale == nullptr: conv bias only)
                    if (a.relu) v = fmaxf(v, 0.f);
                }
                float* o = yb + co * a.tpo;
                o[0] = v;
                if (EPI == MF_AFFINE) {     // (raw outputs: see conv_mfma_store)
                    if (t == 0) { o[-4] = 0.f; o[-3] = 0.f; o[-2] = 0.f; o[-1] = 0.f; }
                    if (t == a.tout - 1) { o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f; }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-tiled 1x1 convolution for the wide pointwise layers (DS-CNN: 276 -> 276 over B x 65 positions, 88 % of that
// network's flops; audio_nets/ds_cnn.py:49-57).  The register-fed kernel above issues 10 gather loads (4 x 64 B
// segments each) per 24 MFMAs and re-reads the activations once per 96-channel tile from HBM; here a workgroup of
// FOUR waves (one per SIMD: the matrix pipes stay balanced) owns 32 NT flat positions x ALL output channels, stages
// 12 input channels of x (coalesced dwords) and of W (float4) per chunk through double-buffered LDS with one barrier
// per chunk, and wave (wm, wn) keeps an (16 MT) x (16 NT) accumulator block: MT + NT conflict-free ds_read_b32 per
// MT x NT MFMAs, x read once.  LDS rows are padded so that the four k-rows of a fragment read land in disjoint bank
// quarters.  NT (positions per workgroup) is picked by the launcher so that the grid fills whole dispatch waves.
// MODE (DS-CNN training, so that no BN pass streams the tensors again): 1 = x is the producing unit's RAW conv output, staged as
// relu(x * in_scale[ci] + in_shift[ci]) (bitwise bn_apply's expression), and the epilogue leaves the per-channel sums of y, y^2
// (EpiSums, forward form); 2 = data gradient whose epilogue leaves the backward sums of the unit it writes the gradient of
// (EpiSums, backward form: that unit's raw output is read at the tile's own addresses); 3 = as 2, and x is the gradient wrt the
// unit's ACTIVATION: its BN backward is applied while x is staged (BnBwdFly; rows are wave-uniform: the six coefficients by scalar
// loads) and the staged dy is also written to dy_out for the filter gradient -- every element is staged exactly once across the
// grid --, so the bn_bwd_apply pass (3 tensor passes, 0.2 ms per layer in the backward's main chain) disappears.  Sums: the 16 positions of a tile column
// block live in the 16 lanes of a DPP row -> row16_sum; the two position halves (wn) meet in LDS; one partial row per workgroup.
// Timing what-ifs (scripts/build_whatif_src.sh mfma TCR_PW_WHATIF <mask>; wrong results, never the product build): 1 no global loads
// after the first chunk, 2 no LDS stores after the first chunk, 4 no barrier in the chunk loop, 8 no MFMAs, 16 no output stores (register-path epilogue),
// 32 / 64 no epilogue affine / no halo zeroing (register-path epilogue), 128 the lean epilogue's stores as fully coalesced 256-byte runs (round 5:
// -15 us per 460 us launch -- an LDS transpose of the output tile would buy less than that).
#ifndef TCR_PW_WHATIF
#define TCR_PW_WHATIF 0
#endif
#define TCR_PWW(bit) ((TCR_PW_WHATIF & (bit)) != 0)
// NWN: wave columns per workgroup (2: four waves, 32 NT positions).  Round 5 measured NWN = 3 (six waves, 96 positions: a staged weight chunk
// serves half again as many positions -- every workgroup walks ALL of W, 1.3 GB of L2 -> CU traffic per launch at 276 channels): DS-CNN-L
// eval 4.48 vs 4.06 ms, training step 17.5 vs 15.7 -- slower, not instantiated.  MINW = 4 (default for the nine-tile instances since round 5):
// the compiler keeps the 72 accumulator registers in VGPRs and fits 128 -- four waves per SIMD instead of three (92 VGPRs + 72 AGPRs):
// DS-CNN-L eval 4.06 -> 3.93 ms, training step 15.72 -> 15.47 ms; TCR_TUNE_PW_POS = 1 selects the unconstrained build (bitwise the same).
// WDMA (round 5, the default of the nine-tile instances): the weight chunk goes global -> LDS by the DMA path (global_load_lds_dwordx4: an
// instruction writes 64 x 16 contiguous bytes of LDS from per-lane global addresses; the chunk's LDS image [12][304] is 14.25 such
// kilobytes, four instructions per wave, their per-lane source offsets fixed for the whole kernel) -- no staging registers, no LDS store
// pass, no per-load predicates or address arithmetic (the register path spends ~100 VALU instructions per chunk and wave beside its 54
// MFMAs, and on this chip they add to the matrix time); the 16 registers it frees hold the NEXT 4-position step's fragments, so the LDS
// reads of a step are in flight behind the previous step's MFMAs (the <= 128-register build otherwise reads, waits, multiplies).
// Columns past Cout carry other weights instead of zeros: their output rows are never stored.
template <int MT, int NT, int EPI, int MODE, int NWN = 2, int MINW = 1, bool WDMA = false>
__global__ __launch_bounds__(128 * NWN, MINW) void conv1x1_lds_kernel(const Conv1x1Args a) {     // MINW = 4: <= 128 registers (four waves per SIMD)
    constexpr int NTHR = 128 * NWN;
#ifndef TCR_PW_KC
#define TCR_PW_KC 12
#endif
    constexpr int KC = WDMA ? TCR_PW_KC : 12;       // input channels per chunk = 3 MFMA k-steps  (side builds: -DTCR_PW_KC=16, round 5: see OPTLOG)
    constexpr int MW = 32 * MT;             // output channels covered (2 wave rows)
    constexpr int XN = 16 * NT * NWN;       // positions per workgroup (NWN wave columns)
    constexpr int WLD = MW + 16;            // (32 MT + 16) % 64 in {16, 48} for MT = 6, 9
    constexpr int XLD = XN + 16;            // (XN + 16) % 64 in {16, 48} for XN = 64, 96
    constexpr int RS = NTHR / XN;           // x rows staged per pass (threads >= RS * XN idle in the x stage)
    constexpr int XPT = KC / RS;            // x dwords per thread and chunk
    constexpr int W4 = MW / 4;              // float4 per weight row
    constexpr int WPT = (KC * W4 + NTHR - 1) / NTHR;
    static_assert(NTHR % XN == 0 && (XLD % 64 == 16 || XLD % 64 == 48), "x staging geometry / bank pattern");
    static_assert(KC % RS == 0, "x staging");
    constexpr int WIMG = KC * WLD;                              // floats of a weight chunk's LDS image
    constexpr int ND = (WIMG + 255) / 256;                      // WDMA: 1 KB copies per chunk
    constexpr int WBUF = WDMA ? ND * 256 : WIMG;                // (the last copy's tail lands in the buffer's own padding)
    constexpr int DPW = (ND + NTHR / 64 - 1) / (NTHR / 64);     // copies per wave and chunk
    __shared__ __attribute__((aligned(16))) float s_w[2][WBUF];
    __shared__ __attribute__((aligned(16))) float s_x[2][KC * XLD];      // (the lean epilogue reads its constant table from here with 16-byte LDS loads)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    const int pos0 = blockIdx.x * XN;

    // ---- staging roles ----
    const bool xuse = tid < RS * XN;
    const int xpos = tid % XN, xrow0 = XN == 64 ? wave : min(tid / XN, RS - 1);   // rows xrow0 + RS * j  (XN = 64: wave-uniform -> scalar loads of in_scale / in_shift)
    static_assert(XN != 64 || NTHR == 256, "wave-uniform x rows");
    const float* xsrc;
    bool xvalid, xfirst, xlast;
    {
        const int p = min(pos0 + xpos, a.npos - 1);
        const int n = p / a.tout, t = p - n * a.tout;
        xsrc = a.x + (size_t)n * a.cin * a.tpi + kHalo + t * a.stride;
        xvalid = pos0 + xpos < a.npos; xfirst = t == 0; xlast = t == a.tout - 1;
    }
    const ptrdiff_t xrel = xsrc - a.x;                          // (MODE 3: the same element of the unit's raw output / of dy_out)
    int wrow[WPT], wcol[WPT];
    bool wuse[WPT], wval[WPT];
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        const int idx = tid + NTHR * j;
        wuse[j] = idx < KC * W4;
        wrow[j] = min(idx / W4, KC - 1);
        wcol[j] = 4 * (idx % W4);
        wval[j] = wcol[j] < a.cout;                         // (Cout % 4 == 0: launcher)
    }
    // WDMA: copy i = wave + (NTHR / 64) j of a chunk writes image floats [256 i, 256 i + 256); this lane's four start at f
    unsigned dsrc[DPW];                                         // byte offset of the lane's source float4 from the chunk's first weight row
    auto dma_src = [&](int j, int rows) -> unsigned {           // rows: weight rows the chunk really has (clamped past them)
        const int f = (wave + (NTHR / 64) * j) * 256 + lane * 4;
        const int row = min(f / WLD, rows - 1), col = min(f % WLD, a.cout - 4);
        return (unsigned)(row * a.cout + col) * 4u;
    };
    if (WDMA) {
#pragma unroll
        for (int j = 0; j < DPW; ++j) dsrc[j] = dma_src(j, KC);
    }
    auto dma_chunk = [&](int c0, int buf) {
        const char* wsrc = reinterpret_cast<const char*>(a.w + (size_t)c0 * a.cout);
        const bool part = c0 + KC > a.cin;                      // (the last chunk of a Cin that is not a multiple of 12)
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            const int i = wave + (NTHR / 64) * j;
            if (i < ND) glds16(wsrc + (part ? dma_src(j, a.cin - c0) : dsrc[j]), &s_w[buf][i * 256]);
        }
    };
    float xr[XPT], xsc[XPT], xsf[XPT];
    float yr[XPT], fk1[XPT], fk2[XPT], fk3[XPT], fmu[XPT];     // MODE 3 (xsc / xsf hold the unit's own scale / shift there)
    int xrw[XPT];
    f32x4 wr[WPT];
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int j = 0; j < XPT; ++j) {
            const int row = min(c0 + xrow0 + RS * j, a.cin - 1);
            xr[j] = xsrc[(size_t)row * a.tpi];
            if (MODE == 1) { xsc[j] = a.in_scale[row]; xsf[j] = a.in_shift[row]; }
            if (MODE == 3) {
                yr[j] = a.fly.raw[xrel + (ptrdiff_t)row * a.tpi];
                xsc[j] = a.fly.self_scale[row]; xsf[j] = a.fly.self_shift[row];
                fk1[j] = a.fly.k1[row]; fk2[j] = a.fly.k2[row]; fk3[j] = a.fly.k3[row]; fmu[j] = a.fly.mean[row];
                xrw[j] = row;
            }
        }
        if (WDMA) return;
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
            const float* src = a.w + (size_t)min(c0 + wrow[j], a.cin - 1) * a.cout + min(wcol[j], a.cout - 4);
            const f32x4 v = *reinterpret_cast<const f32x4*>(src);
            wr[j] = wval[j] ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_chunk = [&](int buf) {
        if (xuse) {
#pragma unroll
            for (int j = 0; j < XPT; ++j) {
                float v = MODE == 1 ? fmaxf(fmaf(xr[j], xsc[j], xsf[j]), 0.f) : xr[j];
                if (MODE == 3) {        // dy = k1 (dz - k2 - (raw - mean) k3), dz = dA [fmaf(raw, scale, shift) > 0]  (bn_bwd_apply's expression)
                    float g = xr[j];
                    if (!(fmaf(yr[j], xsc[j], xsf[j]) > 0.f)) g = 0.f;
                    v = fk1[j] * (g - fk2[j] - (yr[j] - fmu[j]) * fk3[j]);
                    if (xvalid) {
                        float* o = a.dy_out + xrel + (ptrdiff_t)xrw[j] * a.tpi;
                        o[0] = v;
                        if (xfirst) { o[-4] = 0.f; o[-3] = 0.f; o[-2] = 0.f; o[-1] = 0.f; }
                        if (xlast) { o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f; }
                    }
                }
                s_x[buf][(xrow0 + RS * j) * XLD + xpos] = v;
            }
        }
        if (WDMA) return;
#pragma unroll
        for (int j = 0; j < WPT; ++j)
            if (wuse[j]) *reinterpret_cast<f32x4*>(&s_w[buf][wrow[j] * WLD + wcol[j]]) = wr[j];
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // WDMA: the epilogue's per-channel constants (AFFINE: scale / shift; backward sums: mean, invstd, the mask's scale / shift) are fetched
    // here, parked in a few registers through the chunk loop and laid out in the x buffers after it: the epilogue reads them with 16-byte
    // LDS loads (the register-path epilogue gathers them from global memory per output element: 41 us of a 460 us launch, what-if 32)
    constexpr int NA = MODE >= 2 ? 4 : 2;
    constexpr bool TAB = WDMA && (EPI == MF_AFFINE || MODE >= 2);
    constexpr int TPT = (NA * MW + NTHR - 1) / NTHR;
    static_assert(!WDMA || (NA * MW <= 2 * KC * XLD && NWN == 2), "the table fits the x buffers; two wave columns");
    float tabr[TPT];
    if (TAB) {
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            const int i = min(tid + NTHR * j, NA * MW - 1);
            const int which = i / MW, c = min(i - which * MW, a.cout - 1);
            float v;
            if (MODE >= 2) v = which == 0 ? a.sums.mean[c] : which == 1 ? a.sums.invstd[c] : which == 2 ? a.sums.self_scale[c] : a.sums.self_shift[c];
            else v = which == 0 ? (a.scale ? a.scale[c] : 1.0f) : a.shift[c];
            tabr[j] = v;
        }
    }
    const int nchunks = (a.cin + KC - 1) / KC;
    if (WDMA) dma_chunk(0, 0);
    load_chunk(0);
    store_chunk(0);
    if (WDMA) wait_dma();
    __syncthreads();
    const int aoff = q * WLD + wm * (16 * MT) + r, boff = q * XLD + wn * (16 * NT) + r;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < nchunks;
        if (WDMA && more && !TCR_PWW(1)) dma_chunk((ch + 1) * KC, buf ^ 1);
        if (more && !TCR_PWW(1)) load_chunk((ch + 1) * KC);
        const int steps = min(KC / 4, (a.cin - ch * KC) >> 2);      // (last chunk of a Cin that is not a multiple of 12)
        const float* sw = s_w[buf] + aoff;
        const float* sx = s_x[buf] + boff;
        if (WDMA) {             // fragments one step ahead of the MFMAs that use them
            float af[MT], bf[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m] = sw[m * 16];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt] = sx[nt * 16];
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                if (ks < steps) {
                    float an[MT], bn[NT];
                    if (ks + 1 < KC / 4) {      // (a step past `steps` reads stale rows of the buffer and is not multiplied)
#pragma unroll
                        for (int m = 0; m < MT; ++m) an[m] = sw[(ks + 1) * 4 * WLD + m * 16];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) bn[nt] = sx[(ks + 1) * 4 * XLD + nt * 16];
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            if (TCR_PWW(8)) { acc[m][nt][0] += af[m] + bf[nt]; continue; }
                            acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[nt], acc[m][nt], 0, 0, 0);
                        }
                    if (ks + 1 < KC / 4) {
#pragma unroll
                        for (int m = 0; m < MT; ++m) af[m] = an[m];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) bf[nt] = bn[nt];
                    }
                }
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            if (ks < steps) {
                float af[MT], bf[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) af[m] = sw[ks * 4 * WLD + m * 16];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bf[nt] = sx[ks * 4 * XLD + nt * 16];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (TCR_PWW(8)) { acc[m][nt][0] += af[m] + bf[nt]; continue; }
                        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[nt], acc[m][nt], 0, 0, 0);
                    }
            }
        }
        }
        if (more && !TCR_PWW(2)) store_chunk(buf ^ 1);
        if (WDMA) wait_dma();
        if (!TCR_PWW(4)) __syncthreads();
    }

    // (lean addressing: see conv_mfma_store)
    const float inv_tout = 1.0f / (float)a.tout;
    if (WDMA) {
        // Lean epilogue: constants from the LDS table, the common case (whole tile inside the tensor, whole 16-channel tile below Cout) without
        // per-store predicates, halo zeroing after the stores and only for the lanes that sit on an utterance edge.  Same arithmetic per
        // element and the same order of the sums as the register-path epilogue below: bitwise.
        float* s_tab = &s_x[0][0];                              // [NA][MW]  (the loop's last barrier is behind every read of the x buffers)
        if (TAB) {
#pragma unroll
            for (int j = 0; j < TPT; ++j)
                if (tid + NTHR * j < NA * MW) s_tab[tid + NTHR * j] = tabr[j];
            __syncthreads();
        }
        size_t yo[NT];
        bool pv[NT], first[NT], last[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int p = pos0 + (wn * NT + nt) * 16 + r;
            pv[nt] = p < a.npos;
            const int pc = min(p, a.npos - 1);
            const int n = a.npos < (1 << 23) ? fast_div(pc, a.tout, inv_tout) : pc / a.tout, t = pc - n * a.tout;
            yo[nt] = (size_t)n * a.cout * a.tpo + kHalo + t;
            first[nt] = t == 0; last[nt] = t == a.tout - 1;
        }
        const bool full = pos0 + XN <= a.npos;
        float* s_sum = &s_w[0][0];
        auto tile = [&](int m, auto fast_tag) {
            constexpr bool FAST = decltype(fast_tag)::value;
            const int co0 = (wm * MT + m) * 16;
            f32x4 t0 = {1.f, 1.f, 1.f, 1.f}, t1 = {0.f, 0.f, 0.f, 0.f}, t2 = t1, t3 = t1;
            if (TAB) {
                t0 = *reinterpret_cast<const f32x4*>(s_tab + co0 + q * 4);
                t1 = *reinterpret_cast<const f32x4*>(s_tab + MW + co0 + q * 4);
                if (MODE >= 2) {
                    t2 = *reinterpret_cast<const f32x4*>(s_tab + 2 * MW + co0 + q * 4);
                    t3 = *reinterpret_cast<const f32x4*>(s_tab + 3 * MW + co0 + q * 4);
                }
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = co0 + q * 4 + reg;
                const bool cok = FAST || co < a.cout;
                const int cc = FAST ? co : min(co, a.cout - 1);
                float q1 = 0.f, q2 = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bool ok = FAST || (cok && pv[nt]);
                    float v = acc[m][nt][reg];
                    if (EPI == MF_AFFINE) {
                        v = fmaf(v, t0[reg], t1[reg]);
                        if (a.relu) v = fmaxf(v, 0.f);
                    }
                    const size_t off = yo[nt] + (size_t)(cc * a.tpo);
                    if (TCR_PWW(128)) {     // timing what-if: the same bytes as fully coalesced 256-byte runs (wrong addresses)
                        if (ok) a.y[(size_t)blockIdx.x * (4 * MT * 4 * NT * 64) + (size_t)wave * (MT * 4 * NT * 64) + (size_t)((m * 4 + reg) * NT + nt) * 64 + lane] = v;
                    } else
                    if (ok && !(TCR_PWW(16) && v != 12345.f)) a.y[off] = v;
                    if (MODE == 1) {
                        const float yv = ok ? v : 0.f;
                        q1 += yv;
                        q2 = fmaf(yv, yv, q2);
                    } else if (MODE >= 2) {
                        const float rawv = a.sums.raw[off];     // (clamped address: always valid)
                        const float dz = (ok && fmaf(rawv, t2[reg], t3[reg]) > 0.f) ? v : 0.f;
                        q1 += dz;
                        q2 = fmaf(dz, (rawv - t0[reg]) * t1[reg], q2);
                    }
                }
                if (MODE >= 1) {
                    q1 = row16_sum(q1);
                    q2 = row16_sum(q2);
                    if (r == 0 && cok) { s_sum[(wn * 2 + 0) * MW + co] = q1; s_sum[(wn * 2 + 1) * MW + co] = q2; }
                }
            }
        };
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int co0 = (wm * MT + m) * 16;
            if (co0 >= a.cout) break;
            if (full && co0 + 16 <= a.cout) tile(m, std::true_type());
            else tile(m, std::false_type());
        }
        if (EPI == MF_AFFINE) {         // zero halo of the rows this tile starts / ends (the lanes on an utterance's first / last position)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (pv[nt] && (first[nt] || last[nt])) {
                    for (int cq = wm * MT * 16 + q * 4; cq < min((wm + 1) * MT * 16, a.cout); cq += 16)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            if (cq + reg >= a.cout) continue;
                            float* o = a.y + yo[nt] + (size_t)((cq + reg) * a.tpo);
                            if (first[nt]) { o[-4] = 0.f; o[-3] = 0.f; o[-2] = 0.f; o[-1] = 0.f; }
                            if (last[nt]) { o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f; }
                        }
                }
            }
        }
        if (MODE >= 1) {
            __syncthreads();
            for (int i = tid; i < 2 * a.cout; i += NTHR) {
                const int which = i >= a.cout ? 1 : 0, c = i - which * a.cout;
                const float v = s_sum[which * MW + c] + s_sum[(2 + which) * MW + c];
                a.sums.partial[((size_t)blockIdx.x * 2 + which) * a.cout + c] = v;
            }
        }
        return;
    }
    if (MODE == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int p = pos0 + (wn * NT + nt) * 16 + r;
            if (p >= a.npos) continue;
            const int n = a.npos < (1 << 23) ? fast_div(p, a.tout, inv_tout) : p / a.tout, t = p - n * a.tout;
            float* yb = a.y + (size_t)n * a.cout * a.tpo + kHalo + t;
            const bool first = t == 0, last = t == a.tout - 1;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co0 = (wm * MT + m) * 16;
                if (co0 >= a.cout) break;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int co = co0 + q * 4 + reg;
                    if (co >= a.cout) continue;
                    float v = acc[m][nt][reg];
                    if (EPI == MF_AFFINE && !TCR_PWW(32)) {
                        v = fmaf(v, a.scale ? a.scale[co] : 1.0f, a.shift[co]);
                        if (a.relu) v = fmaxf(v, 0.f);
                    }
                    float* o = yb + co * a.tpo;
                    if (TCR_PWW(16) && v != 12345.f) continue;
                    o[0] = v;
                    if (EPI == MF_AFFINE && !TCR_PWW(64)) {
                        if (first) { o[-4] = 0.f; o[-3] = 0.f; o[-2] = 0.f; o[-1] = 0.f; }
                        if (last) { o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f; }
                    }
                }
            }
        }
        return;
    }
    // MODE 1 / 2: channel-outer order, so that a channel's contributions of both column tiles meet in one pair of registers
    size_t yo[NT];
    bool pv[NT], first[NT], last[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int p = pos0 + (wn * NT + nt) * 16 + r;
        pv[nt] = p < a.npos;
        const int pc = min(p, a.npos - 1);
        const int n = a.npos < (1 << 23) ? fast_div(pc, a.tout, inv_tout) : pc / a.tout, t = pc - n * a.tout;
        yo[nt] = (size_t)n * a.cout * a.tpo + kHalo + t;
        first[nt] = t == 0; last[nt] = t == a.tout - 1;
    }
    float* s_sum = &s_w[0][0];              // [wn][which][MW] (the loop's last barrier is behind every LDS read of the tiles)
    static_assert(NWN * 2 * MW <= KC * WLD, "sums fit the first weight buffer");
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int co0 = (wm * MT + m) * 16;
        if (co0 >= a.cout) break;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int co = co0 + q * 4 + reg;
            const bool cok = co < a.cout;                       // (uniform over a DPP row: its lanes share q)
            const int cc = min(co, a.cout - 1);
            float sc1 = 1.0f, sf1 = 0.f, mu = 0.f, is = 0.f, ssc = 0.f, ssh = 0.f;
            if (EPI == MF_AFFINE) { sc1 = a.scale ? a.scale[cc] : 1.0f; sf1 = a.shift[cc]; }
            if (MODE >= 2) { mu = a.sums.mean[cc]; is = a.sums.invstd[cc]; ssc = a.sums.self_scale[cc]; ssh = a.sums.self_shift[cc]; }
            float q1 = 0.f, q2 = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool ok = cok && pv[nt];
                float v = acc[m][nt][reg];
                if (EPI == MF_AFFINE) {
                    v = fmaf(v, sc1, sf1);
                    if (a.relu) v = fmaxf(v, 0.f);
                }
                const size_t off = yo[nt] + (size_t)(cc * a.tpo);
                if (ok && !(TCR_PWW(16) && v != 12345.f)) {
                    float* o = a.y + off;
                    o[0] = v;
                    if (EPI == MF_AFFINE) {
                        if (first[nt]) { o[-4] = 0.f; o[-3] = 0.f; o[-2] = 0.f; o[-1] = 0.f; }
                        if (last[nt]) { o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; o[4] = 0.f; }
                    }
                }
                if (MODE == 1) {
                    const float yv = ok ? v : 0.f;
                    q1 += yv;
                    q2 = fmaf(yv, yv, q2);
                } else {
                    const float rawv = a.sums.raw[off];         // (clamped address: always valid)
                    const float dz = (ok && fmaf(rawv, ssc, ssh) > 0.f) ? v : 0.f;
                    q1 += dz;
                    q2 = fmaf(dz, (rawv - mu) * is, q2);
                }
            }
            q1 = row16_sum(q1);
            q2 = row16_sum(q2);
            if (r == 0 && cok) { s_sum[(wn * 2 + 0) * MW + co] = q1; s_sum[(wn * 2 + 1) * MW + co] = q2; }
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * a.cout; i += NTHR) {
        const int which = i >= a.cout ? 1 : 0, c = i - which * a.cout;
        float v = s_sum[which * MW + c] + s_sum[(2 + which) * MW + c];
        if (NWN > 2) v += s_sum[(4 + which) * MW + c];
        a.sums.partial[((size_t)blockIdx.x * 2 + which) * a.cout + c] = v;
    }
}

bool conv1x1_lds_covers(int cin, int cout) {
    const int tiles = ceil_div(cout, 16);
    return tiles >= 7 && tiles <= 18 && (cin & 3) == 0 && (cout & 3) == 0 && tune_get(TCR_TUNE_CONV_B) != 3;
}

int conv1x1_sum_rows(int npos) { return ceil_div(npos, 64); }

int launch_conv1x1(const Conv1x1Args& a, int epi, hipStream_t s) {
    const int tiles = ceil_div(a.cout, 16);
    const int knob = tune_get(TCR_TUNE_CONV_B);
    const bool extras = a.in_scale || a.sums.partial || a.dy_out;
    if (conv1x1_lds_covers(a.cin, a.cout)) {
        const int mt = tiles > 12 ? 9 : 6;
        // 2 column tiles per wave = 64 positions per workgroup (3 tiles: 236 VGPRs, 2 waves per SIMD, slower; 4: slower still)
        const dim3 lgrid(ceil_div(a.npos, 64));
        const dim3 lblk(256);
        const bool cap128 = tune_get(TCR_TUNE_PW_POS) != 1;     // nine-tile instances at <= 128 registers (four waves per SIMD)
        const bool wdma = tune_get(TCR_TUNE_PW_POS) == 0 && a.cout % 4 == 0 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0;     // ... with the weight chunks by DMA
        if (extras) {
            // training forms: (in-affine + forward sums) with the bias epilogue, or (backward sums) on the raw data gradient
            const bool fwd = epi == MF_AFFINE && a.in_scale && a.in_shift && a.sums.partial && !a.sums.raw;
            const bool bwd = epi == MF_RAW && !a.in_scale && a.sums.partial && a.sums.raw && a.sums.mean && a.sums.invstd && a.sums.self_scale && a.sums.self_shift && a.stride == 1;
            const bool bfly = bwd && a.dy_out && a.fly.raw && a.fly.mean && a.fly.k1 && a.fly.k2 && a.fly.k3 && a.fly.self_scale && a.fly.self_shift && a.tpi == a.tpo;
            if (!fwd && !bwd) { set_error("conv1x1: unsupported combination of in-affine / epilogue sums"); return TCR_ERR_ARG; }
            if (a.dy_out && !bfly) { set_error("conv1x1: on-the-fly BN backward needs the data-gradient form with sums"); return TCR_ERR_ARG; }
#define TCR_LT(MT_, NWN_)                                                                                               \
    if (fwd) hipLaunchKernelGGL((conv1x1_lds_kernel<MT_, 2, MF_AFFINE, 1, NWN_>), lgrid, lblk, 0, s, a);               \
    else if (bfly) hipLaunchKernelGGL((conv1x1_lds_kernel<MT_, 2, MF_RAW, 3, NWN_>), lgrid, lblk, 0, s, a);            \
    else hipLaunchKernelGGL((conv1x1_lds_kernel<MT_, 2, MF_RAW, 2, NWN_>), lgrid, lblk, 0, s, a)
            if (wdma && mt == 9 && !bfly) {
                if (fwd) hipLaunchKernelGGL((conv1x1_lds_kernel<9, 2, MF_AFFINE, 1, 2, 4, true>), lgrid, lblk, 0, s, a);
                else hipLaunchKernelGGL((conv1x1_lds_kernel<9, 2, MF_RAW, 2, 2, 4, true>), lgrid, lblk, 0, s, a);
            }
            else if (wdma && mt == 6 && !bfly) {        // (DS-CNN-M, 172 channels)
                if (fwd) hipLaunchKernelGGL((conv1x1_lds_kernel<6, 2, MF_AFFINE, 1, 2, 4, true>), lgrid, lblk, 0, s, a);
                else hipLaunchKernelGGL((conv1x1_lds_kernel<6, 2, MF_RAW, 2, 2, 4, true>), lgrid, lblk, 0, s, a);
            }
            else if (cap128 && mt == 9) {
                if (fwd) hipLaunchKernelGGL((conv1x1_lds_kernel<9, 2, MF_AFFINE, 1, 2, 4>), lgrid, lblk, 0, s, a);
                else if (bfly) hipLaunchKernelGGL((conv1x1_lds_kernel<9, 2, MF_RAW, 3, 2, 1>), lgrid, lblk, 0, s, a);     // (this form spills at 128)
                else hipLaunchKernelGGL((conv1x1_lds_kernel<9, 2, MF_RAW, 2, 2, 4>), lgrid, lblk, 0, s, a);
            }
            else if (mt == 9) { TCR_LT(9, 2); } else { TCR_LT(6, 2); }
#undef TCR_LT
            return check_launch("conv1x1_lds_kernel");
        }
#define TCR_LL(MT_, NWN_)                                                                                               \
    if (epi == MF_RAW) hipLaunchKernelGGL((conv1x1_lds_kernel<MT_, 2, MF_RAW, 0, NWN_>), lgrid, lblk, 0, s, a);        \
    else hipLaunchKernelGGL((conv1x1_lds_kernel<MT_, 2, MF_AFFINE, 0, NWN_>), lgrid, lblk, 0, s, a)
        if (wdma && mt == 9) {
            if (epi == MF_RAW) hipLaunchKernelGGL((conv1x1_lds_kernel<9, 2, MF_RAW, 0, 2, 4, true>), lgrid, lblk, 0, s, a);
            else hipLaunchKernelGGL((conv1x1_lds_kernel<9, 2, MF_AFFINE, 0, 2, 4, true>), lgrid, lblk, 0, s, a);
        }
        else if (wdma && mt == 6) {
            if (epi == MF_RAW) hipLaunchKernelGGL((conv1x1_lds_kernel<6, 2, MF_RAW, 0, 2, 4, true>), lgrid, lblk, 0, s, a);
            else hipLaunchKernelGGL((conv1x1_lds_kernel<6, 2, MF_AFFINE, 0, 2, 4, true>), lgrid, lblk, 0, s, a);
        }
        else if (cap128 && mt == 9) {
            if (epi == MF_RAW) hipLaunchKernelGGL((conv1x1_lds_kernel<9, 2, MF_RAW, 0, 2, 4>), lgrid, lblk, 0, s, a);
            else hipLaunchKernelGGL((conv1x1_lds_kernel<9, 2, MF_AFFINE, 0, 2, 4>), lgrid, lblk, 0, s, a);
        }
        else if (mt == 9) { TCR_LL(9, 2); } else { TCR_LL(6, 2); }
#undef TCR_LL
        return check_launch("conv1x1_lds_kernel");
    }
    if (extras) { set_error("conv1x1: in-affine / epilogue sums need the LDS-tiled kernel (%d -> %d channels)", a.cin, a.cout); return TCR_ERR_ARG; }
    const int mt = tiles >= 12 ? 6 : (tiles >= 3 ? 3 : tiles);       // (9 tiles per wave spill: 1.5x slower)       // wide layers: 96 channels per wave halve the re-reads of x
    const dim3 grid(ceil_div(a.npos, 256), ceil_div(tiles, mt));
#define TCR_L1(MT_)                                                                                         \
    if (epi == MF_RAW) hipLaunchKernelGGL((conv1x1_mfma_kernel<MT_, MF_RAW>), grid, dim3(256), 0, s, a);    \
    else hipLaunchKernelGGL((conv1x1_mfma_kernel<MT_, MF_AFFINE>), grid, dim3(256), 0, s, a)
    if (mt == 1) { TCR_L1(1); }
    else if (mt == 2) { TCR_L1(2); }
    else if (mt == 6) { TCR_L1(6); }
    else { TCR_L1(3); }
#undef TCR_L1
    return check_launch("conv1x1_mfma_kernel");
}

// ---------------------------------------------------------------------------------------------
// k x 1 convolution as an implicit GEMM on the matrix cores:
//   D[row = co][col = position] = sum_{j, ci} W[j][ci][co] * x[b][ci][t*S + j - pad_lo]
// A = W (16 output channels x 4 input channels of one tap per instruction, straight from L1/L2 -- the
// weight tensor is shared by every workgroup), B = activations read from an LDS image of the
// workgroup's utterances.  Because activations are planar per utterance, the rows a workgroup needs
// ([n0..n1][Cin][Tp]) are ONE contiguous block of global memory: the LDS image is a straight float4
// memcpy (fully coalesced) and taps / strides / SAME padding become plain LDS offsets into the halo'd rows.
// The f32 MFMA is bitwise an fmaf chain, so this changes scheduling, not numerics.
// Why not the scalar-fed VALU kernel of conv.hip everywhere: once the weight tensor outgrows the 16 KB
// scalar cache (every block conv; 20-83 KB) its waves stall on scalar-cache misses (measured 5-8 % of the
// FP32 peak, 75 % of wave time in s_waitcnt; profiles/r01_baseline_pmc.csv).
// ---------------------------------------------------------------------------------------------
// DOWN: additionally produce the block's 1x1 stride-2 "down" shortcut (tc_resnet.py:30-32) from the SAME
// LDS image -- it is the centre-tap column of the stride-2 9x1 conv with its own weights.
struct ConvDownArgs {
    const float* w;         // [Cin][Cout]
    float* y;               // [B][Cout][Tpo]
    const float* scale;
    const float* shift;
    int tap;                // pad_lo of the main conv: x[t*S + tap - pad_lo] == x[t*S]
    int relu;
};

struct ConvStoreExtra {
    int ostride, ooff;
    const float* add;
    const float* add_mask;
    int add_bcast;
};

template <int MT, int EPI>
__device__ __forceinline__ void conv_mfma_store(const f32x4 (&acc)[MT][2], float* y, const float* scale, const float* shift,
                                                const float* res, int relu, int cot0, int cout, int tout, int tpo,
                                                int p_base, int wg_p1, int r, int q, const ConvStoreExtra ex) {
    // One 64-bit row offset per position tile and one 32-bit channel offset per accumulator row: the per-store address is
    // a single add (own 64-bit multiply chains per store and the integer division of the position split were a large part
    // of the short train-mode launches' instruction count).
    const float inv_tout = 1.0f / (float)tout;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int p = p_base + nt * 16 + r;
        if (p >= wg_p1) continue;
        const int n = wg_p1 < (1 << 23) ? fast_div(p, tout, inv_tout) : p / tout, t = p - n * tout;
        const size_t ob = (size_t)n * cout * tpo + kHalo + t * ex.ostride + ex.ooff;
        // The data gradient's addend (an identity shortcut's gradient under its mask, or the phases another conv wrote first): ALL of a
        // lane's 4 MT addend / mask loads are requested before the first is used.  (Round 6: as part of the store loop below every
        // element was load -> s_waitcnt vmcnt(0) -> add -> store, 4 MT dependent memory round trips per position tile in a ~50 us kernel.)
        float addv[MT][4];
        if (EPI != EPI_AFFINE && ex.add) {
            float mkv[MT][4];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int co = min((cot0 + m) * 16 + q * 4 + reg, cout - 1);       // (clamped: rows past Cout are never stored)
                    const size_t o = ob + (size_t)(co * tpo);
                    addv[m][reg] = ex.add_bcast ? ex.add[(size_t)n * cout + co] : ex.add[o];
                    mkv[m][reg] = ex.add_mask ? ex.add_mask[o] : 1.f;
                }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (!(mkv[m][reg] > 0.f)) addv[m][reg] = 0.f;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int co = (cot0 + m) * 16 + q * 4 + reg;
                if (co >= cout) continue;
                float v = acc[m][nt][reg];
                const size_t o = ob + (size_t)(co * tpo);
                if (EPI == EPI_AFFINE) {
                    v = fmaf(v, scale[co], shift[co]);
                    if (res) v = fmaxf(v + res[o], 0.f);            // net += layer_in; relu  (tc_resnet.py:40-41)
                    else if (relu) v = fmaxf(v, 0.f);
                } else if (ex.add) {
                    v += addv[m][reg];
                }
                float* dst = y + o;
                dst[0] = v;
                // Only the eval-mode activations are read through their zero halo by the next conv.  Raw train-mode
                // outputs and data gradients go through bn_apply / bn_bwd_apply, which rewrite whole rows (halo = 0)
                // and read interiors only -- skipping the halo stores there also keeps the epilogue code short.
                if (EPI == EPI_AFFINE) {
                    if (t == 0) { dst[-4] = 0.f; dst[-3] = 0.f; dst[-2] = 0.f; dst[-1] = 0.f; }
                    if (t == tout - 1) { dst[1] = 0.f; dst[2] = 0.f; dst[3] = 0.f; dst[4] = 0.f; }
                }
            }
    }
}

// LDSB = true: activations via the LDS image; false: B fragments straight from global/L1 (each fragment load is
// 4 input-channel rows x 16 consecutive positions = four 64-byte segments), prefetched together with the
// weights -- no staging pass, no barrier, no LDS-limited occupancy, no over-fetch of whole utterances.
template <int K, int S, int MT, int EPI, bool DOWN, bool LDSB>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a, const ConvDownArgs d, const int ppw) {
    constexpr int CH = 4;                                   // K-steps (of 4 input channels) per weight prefetch chunk
    float* xt = reinterpret_cast<float*>(dyn_lds());
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int wg_p0 = blockIdx.x * ppw;
    const int wg_p1 = min(wg_p0 + ppw, a.npos);
    const int n0 = wg_p0 / a.tout, n1 = (wg_p1 - 1) / a.tout;
    const int row = a.cin * a.tpi;                          // floats per utterance (multiple of 4)
    if (LDSB) {
        const float4* src = reinterpret_cast<const float4*>(a.x + (size_t)n0 * row);
        float4* dst = reinterpret_cast<float4*>(xt);
        const int nvec = (n1 - n0 + 1) * (row / 4);
        for (int i = tid; i < nvec; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const int p_base = wg_p0 + wave * 32;                   // 2 column tiles of 16 positions per wave
    if (p_base >= wg_p1) return;

    const float* xb = LDSB ? xt : a.x + (size_t)n0 * row;   // same indexing for the LDS image and for global
    int xo[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int p = min(p_base + nt * 16 + r, wg_p1 - 1);
        const int n = p / a.tout, t = p - n * a.tout;
        xo[nt] = (n - n0) * row + t * S + a.xoff + q * a.tpi;
    }
    const int cot0 = blockIdx.y * MT;
    f32x4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int wofs[MT];                                           // q * Cout + co of this lane's A-fragment column
    bool wv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int co = (cot0 + m) * 16 + r;
        wv[m] = co < a.cout;
        wofs[m] = q * a.cout + (wv[m] ? co : 0);
    }
    const int C4 = a.cin >> 2;
    const int tap_stride = a.cin * a.cout;
    const int step_stride = 4 * a.cout;

    // Weights stream from L1/L2 with a one-chunk (4 K-steps = 4*MT loads) lookahead so that their latency
    // hides behind the previous chunk's 8*MT MFMAs; activations come from the LDS image.
    // (round 6: buffer-descriptor loads as in conv_mfma_ksplit_kernel -- uniform base, constant lane offset, uniform running offset;
    //  channel quads past C4 and output channels past Cout read a clamped address and are zeroed by the selects, as before)
    const buf_rsrc wr = make_rsrc(a.w), wdr = make_rsrc(DOWN ? d.w : a.w), xrs = make_rsrc(a.x + (size_t)n0 * row);
    auto load_chunk = [&](const buf_rsrc w, int j, int c0, float (&af)[CH][MT + 2]) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c4 = c0 + i;
            const int c4c = min(c4, C4 - 1);
            const unsigned ws = (unsigned)(j * tap_stride + c4c * step_stride) * 4u, xs = (unsigned)(4 * c4c * a.tpi + j) * 4u;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float v = buf_load_f32(w, (unsigned)wofs[m] * 4u, ws);
                af[i][m] = (c4 < C4 && wv[m]) ? v : 0.f;
            }
            if (!LDSB) {
                const float v0 = buf_load_f32(xrs, (unsigned)xo[0] * 4u, xs), v1 = buf_load_f32(xrs, (unsigned)xo[1] * 4u, xs);
                af[i][MT] = (c4 < C4) ? v0 : 0.f;
                af[i][MT + 1] = (c4 < C4) ? v1 : 0.f;
            }
        }
    };
    auto mma_chunk = [&](int j, int c0, const float (&af)[CH][MT + 2], f32x4 (&ac)[MT][2]) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c4 = c0 + i;
            if (c4 < C4) {
                const float b0 = LDSB ? xb[xo[0] + 4 * c4 * a.tpi + j] : af[i][MT];
                const float b1 = LDSB ? xb[xo[1] + 4 * c4 * a.tpi + j] : af[i][MT + 1];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    ac[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][m], b0, ac[m][0], 0, 0, 0);
                    ac[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][m], b1, ac[m][1], 0, 0, 0);
                }
            }
        }
    };
    const int cpj = (C4 + CH - 1) / CH;
    const int nchunks = K * cpj;
    float afA[CH][MT + 2], afB[CH][MT + 2];
    int j = 0, c0 = 0;
    load_chunk(wr, 0, 0, afA);
    for (int ch = 0; ch < nchunks; ch += 2) {
        int j1 = j, c1 = c0 + CH;
        if (c1 >= C4) { c1 = 0; ++j1; }
        if (ch + 1 < nchunks) load_chunk(wr, j1, c1, afB);
        mma_chunk(j, c0, afA, acc);
        int j2 = j1, c2 = c1 + CH;
        if (c2 >= C4) { c2 = 0; ++j2; }
        if (ch + 2 < nchunks) load_chunk(wr, j2, c2, afA);
        if (ch + 1 < nchunks) mma_chunk(j1, c1, afB, acc);
        j = j2;
        c0 = c2;
    }
    const ConvStoreExtra ex = {a.ostride > 0 ? a.ostride : 1, a.ooff, a.add, a.add_mask, a.add_bcast};
    conv_mfma_store<MT, EPI>(acc, a.y, a.scale, a.shift, a.res, a.relu, cot0, a.cout, a.tout, a.tpo, p_base, wg_p1, r, q, ex);

    if (DOWN) {
        f32x4 acc2[MT][2];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc2[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int cc = 0; cc < C4; cc += CH) {
            load_chunk(wdr, 0, cc, afA);
            if (!LDSB) {
#pragma unroll
                for (int i = 0; i < CH; ++i) {      // B of the centre tap (load_chunk fetched tap 0)
                    afA[i][MT] = (cc + i < C4) ? xb[xo[0] + 4 * (cc + i) * a.tpi + d.tap] : 0.f;
                    afA[i][MT + 1] = (cc + i < C4) ? xb[xo[1] + 4 * (cc + i) * a.tpi + d.tap] : 0.f;
                }
            }
            mma_chunk(d.tap, cc, afA, acc2);
        }
        const ConvStoreExtra ex2 = {1, 0, nullptr, nullptr, 0};
        conv_mfma_store<MT, EPI>(acc2, d.y, d.scale, d.shift, nullptr, d.relu, cot0, a.cout, a.tout, a.tpo, p_base, wg_p1, r, q, ex2);
    }
}

// K-split form for launches that cannot fill the chip (training at the late layers: a few hundred to a few thousand
// 32-position waves on 1024 SIMDs, each walking K * Cin / 4 dependent-latency steps): a workgroup is ONE 32-position
// group and KS waves that take every KS-th chunk of the reduction; the partial accumulators are added through LDS in a
// fixed order (wave 0 + wave 1 + ...), so the result is reproducible -- though not bitwise the same sum as KS = 1.
// Raw epilogue only (train-mode forward and the data gradient), activations straight from global memory.
template <int K, int S, int MT, bool DOWN, int KS>
__global__ __launch_bounds__(64 * KS) void conv_mfma_ksplit_kernel(const ConvArgs a, const ConvDownArgs d) {
    constexpr int CH = 4;
    float* red = reinterpret_cast<float*>(dyn_lds());       // [KS - 1][MT * 8][64]
    const int lane = threadIdx.x & 63;
    // (readfirstlane: tells the compiler the wave index is uniform, so the chunk bookkeeping below runs on the scalar unit)
    const int ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int p_base = blockIdx.x * 32;
    const int wg_p1 = min(p_base + 32, a.npos);
    const int n0 = p_base / a.tout;
    const int row = a.cin * a.tpi;
    const int C4 = a.cin >> 2;
    // Per-lane operand pointers at (tap 0, channel quad 0); a K-step adds a wave-uniform offset.  Out-of-range output
    // channels / channel quads are CLAMPED, not predicated: the products land in rows that are never stored or in steps
    // that are never multiplied, and the loop stays free of per-load masks.
    // Round 6: operands through buffer descriptors (gfx950_isa.h: buf_load_f32) -- a wave-uniform base, a per-lane 32-bit byte offset
    // that never changes, a wave-uniform running offset.  As per-lane 64-bit pointers + a uniform offset (rounds 2-5) every load was
    // preceded by a 64-bit vector add (v_lshl_add_u64) and half a dozen scalar instructions: ~4 VALU + ~4 SALU instructions per MFMA in
    // the PMC counts of this kernel (profiles/r06_train14_pmc.csv), on a chip where VALU issue adds to the matrix pipe's time.  (Uniform
    // base + zero-extended lane offset as plain global loads did NOT get there: the compiler hoists the zero-extension out of the loop
    // and adds the 64-bit pair to the scalar base with the same vector instruction.)
    const buf_rsrc xr = make_rsrc(a.x + (size_t)n0 * row), wr = make_rsrc(a.w), wdr = make_rsrc(DOWN ? d.w : a.w);
    unsigned xo[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int p = min(p_base + nt * 16 + r, wg_p1 - 1);
        const int n = p / a.tout, t = p - n * a.tout;
        xo[nt] = (unsigned)((n - n0) * row + t * S + a.xoff + q * a.tpi) * 4u;
    }
    const int cot0 = blockIdx.y * MT;
    unsigned wo[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) wo[m] = (unsigned)(q * a.cout + min((cot0 + m) * 16 + r, a.cout - 1)) * 4u;
    f32x4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tap_stride = a.cin * a.cout;
    const int step_stride = 4 * a.cout;
    const int xq = 4 * a.tpi;
    // chunk = CH consecutive channel quads of one tap; woff / xoff: uniform float offsets of its first K-step
    auto load_chunk = [&](const buf_rsrc wu, int woff, int xoff, int c0, float (&af)[CH][MT + 2]) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int ci = min(c0 + i, C4 - 1) - c0;                // (tail: re-read the last quad; not multiplied)
            const unsigned ws = (unsigned)(woff + ci * step_stride) * 4u, xs = (unsigned)(xoff + ci * xq) * 4u;     // (uniform)
#pragma unroll
            for (int m = 0; m < MT; ++m) af[i][m] = buf_load_f32(wu, wo[m], ws);
            af[i][MT] = buf_load_f32(xr, xo[0], xs);
            af[i][MT + 1] = buf_load_f32(xr, xo[1], xs);
        }
    };
    auto mma_chunk = [&](int c0, const float (&af)[CH][MT + 2], f32x4 (&ac)[MT][2]) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (c0 + i < C4) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    ac[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][m], af[i][MT], ac[m][0], 0, 0, 0);
                    ac[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][m], af[i][MT + 1], ac[m][1], 0, 0, 0);
                }
            }
        }
    };
    // partial accumulators of waves 1 .. KS-1 -> LDS -> added by wave 0 in wave order
    auto reduce = [&](f32x4 (&ac)[MT][2]) {
        if (KS == 1) return;
        if (ks > 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) red[(((ks - 1) * MT + m) * 8 + nt * 4 + reg) * 64 + lane] = ac[m][nt][reg];
        }
        __syncthreads();
        if (ks == 0) {
            for (int k2 = 1; k2 < KS; ++k2)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) ac[m][nt][reg] += red[(((k2 - 1) * MT + m) * 8 + nt * 4 + reg) * 64 + lane];
        }
        __syncthreads();
    };
    const int cpj = (C4 + CH - 1) / CH;
    const int nchunks = K * cpj;
    // this wave's chunks: ks, ks + KS, ...; (tap j, chunk-in-tap cq) advance with scalar adds
    int j = 0, cq = ks;
    auto norm = [&]() { while (cq >= cpj) { cq -= cpj; ++j; } };
    norm();
    float afA[CH][MT + 2], afB[CH][MT + 2];
    if (ks < nchunks) load_chunk(wr, j * tap_stride + cq * CH * step_stride, j + cq * CH * xq, cq * CH, afA);
    for (int ch = ks; ch < nchunks; ch += 2 * KS) {
        const int cqa = cq;
        cq += KS; norm();
        const int cqb = cq;
        if (ch + KS < nchunks) load_chunk(wr, j * tap_stride + cq * CH * step_stride, j + cq * CH * xq, cq * CH, afB);
        mma_chunk(cqa * CH, afA, acc);
        cq += KS; norm();
        if (ch + 2 * KS < nchunks) load_chunk(wr, j * tap_stride + cq * CH * step_stride, j + cq * CH * xq, cq * CH, afA);
        if (ch + KS < nchunks) mma_chunk(cqb * CH, afB, acc);
    }
    reduce(acc);
    if (ks == 0) {
        const ConvStoreExtra ex = {a.ostride > 0 ? a.ostride : 1, a.ooff, a.add, a.add_mask, a.add_bcast};
        conv_mfma_store<MT, EPI_RAW>(acc, a.y, a.scale, a.shift, a.res, a.relu, cot0, a.cout, a.tout, a.tpo, p_base, wg_p1, r, q, ex);
    }
    if (DOWN) {
        f32x4 acc2[MT][2];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc2[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int cc = ks * CH; cc < C4; cc += KS * CH) {
            load_chunk(wdr, cc * step_stride, d.tap + cc * xq, cc, afA);
            mma_chunk(cc, afA, acc2);
        }
        reduce(acc2);
        if (ks == 0) {
            const ConvStoreExtra ex2 = {1, 0, nullptr, nullptr, 0};
            conv_mfma_store<MT, EPI_RAW>(acc2, d.y, d.scale, d.shift, nullptr, d.relu, cot0, a.cout, a.tout, a.tpo, p_base, wg_p1, r, q, ex2);
        }
    }
}

template <int K, int S, int MT, bool DOWN>
static void launch_conv_ksplit(int ks, dim3 grid, const ConvArgs& a, const ConvDownArgs& d, hipStream_t s) {
    const size_t lds = (size_t)(ks - 1) * MT * 8 * 64 * sizeof(float);
    if (ks == 4) hipLaunchKernelGGL((conv_mfma_ksplit_kernel<K, S, MT, DOWN, 4>), grid, dim3(256), lds, s, a, d);
    else hipLaunchKernelGGL((conv_mfma_ksplit_kernel<K, S, MT, DOWN, 2>), grid, dim3(128), lds, s, a, d);
}

// Returns TCR_OK after launching, or 1 when the shape does not fit this kernel (caller falls back).
template <int K, int S>
static int launch_conv_mfma_ks(const ConvArgs& a, const ConvDownArgs* down, int epi, hipStream_t s) {
    if (a.cin % 4 != 0) return 1;
    const int row = a.cin * a.tpi;
    // positions per workgroup (32 per wave): shrink until the grid has >= ~768 workgroups and the LDS image
    // of the utterances a workgroup touches fits 64 KB
    int ppw = 128;
    while (ppw > 32 && ceil_div(a.npos, ppw) < 768) ppw /= 2;
    size_t lds = 0;
    for (; ppw >= 32; ppw /= 2) {
        const int span = (ppw + a.tout - 2) / a.tout + 1;       // utterances a workgroup can touch
        lds = (size_t)span * row * sizeof(float);
        if (lds <= 64 * 1024) break;
    }
    const bool ldsb = tune_get(TCR_TUNE_CONV_B) == 1;
    if (ldsb && ppw < 32) return 1;
    if (!ldsb) { ppw = 128; lds = 0; while (ppw > 32 && ceil_div(a.npos, ppw) < 1024) ppw /= 2; }
    const int tiles = ceil_div(a.cout, 16);
    const int mt = tiles <= 3 ? tiles : (tiles == 4 ? 2 : (tiles == 5 ? 5 : 3));
    ConvDownArgs d;
    if (down) d = *down; else { d.w = nullptr; d.y = nullptr; d.scale = d.shift = nullptr; d.tap = 0; d.relu = 0; }
    // K-split (raw epilogue = training only; the eval path keeps the single-chain sum that is bitwise the fused kernel's)
    if (epi == EPI_RAW && !ldsb) {
        const int groups = ceil_div(a.npos, 32) * ceil_div(tiles, mt);          // 32-position waves of the plain form
        const int nchunks = K * ceil_div(a.cin >> 2, 4);
        int ks = tune_get(TCR_TUNE_CONV_KSPLIT);
        if (ks == 0) ks = groups >= 8192 ? 1 : (groups >= 4096 ? 2 : 4);
        while (ks > 1 && nchunks < 2 * ks) ks /= 2;
        if (ks == 2 || ks == 4) {
            const dim3 kgrid(ceil_div(a.npos, 32), ceil_div(tiles, mt));
#define TCR_CK(MT_) if (down) launch_conv_ksplit<K, S, MT_, true>(ks, kgrid, a, d, s); else launch_conv_ksplit<K, S, MT_, false>(ks, kgrid, a, d, s)
            switch (mt) {
                case 1: TCR_CK(1); break;
                case 2: TCR_CK(2); break;
                case 3: TCR_CK(3); break;
                default: TCR_CK(5); break;
            }
#undef TCR_CK
            return check_launch("conv_mfma_ksplit_kernel");
        }
    }
    const dim3 grid(ceil_div(a.npos, ppw), ceil_div(tiles, mt));
    const dim3 block(ppw * 2);
#define TCR_CM3(MT_, EPI_, LB_)                                                                                         \
    if (down) hipLaunchKernelGGL((conv_mfma_kernel<K, S, MT_, EPI_, true, LB_>), grid, block, lds, s, a, d, ppw);       \
    else hipLaunchKernelGGL((conv_mfma_kernel<K, S, MT_, EPI_, false, LB_>), grid, block, lds, s, a, d, ppw)
#define TCR_CM2(MT_, EPI_) if (ldsb) { TCR_CM3(MT_, EPI_, true); } else { TCR_CM3(MT_, EPI_, false); }
#define TCR_CM(MT_) if (epi == EPI_RAW) { TCR_CM2(MT_, EPI_RAW); } else { TCR_CM2(MT_, EPI_AFFINE); }
    switch (mt) {
        case 1: TCR_CM(1); break;
        case 2: TCR_CM(2); break;
        case 3: TCR_CM(3); break;
        default: TCR_CM(5); break;
    }
#undef TCR_CM
#undef TCR_CM2
#undef TCR_CM3
    return check_launch("conv_mfma_kernel");
}

int launch_conv_mfma(int k, int stride, const ConvArgs& a, int epi, hipStream_t s) {
    if (k == 3 && stride == 1) return launch_conv_mfma_ks<3, 1>(a, nullptr, epi, s);
    if (k == 9 && stride == 1) return launch_conv_mfma_ks<9, 1>(a, nullptr, epi, s);
    if (k == 9 && stride == 2) return launch_conv_mfma_ks<9, 2>(a, nullptr, epi, s);
    return 1;
}

// ---- data gradient on the matrix cores -------------------------------------------------------------------------
// dx[ci][tin] = sum_{j,co} dy[co][(tin + pad_lo - j) / S] * W[j][ci][co] over taps with S | (tin + pad_lo - j).
// For each output phase r = tin mod S this is a stride-1 convolution over dy:
//   dx[ci][S*u + r] = sum_{i'} sum_co dy[co][u + i' + d_min] * Wr[i'][co][ci],   Wr[i'] = W[r + pad_lo - S*(i' + d_min)]^T
// so the forward implicit-GEMM kernel is reused with re-arranged weights, an output stride of S and offset r.
__device__ __forceinline__ void dgrad_weights_body(const float* __restrict__ w, float* __restrict__ wt, int k, int cin, int cout,
                                                   int stride, int pad_lo, int first, int step) {
    // wt holds the phases back to back: phase r has taps j = j0_r, j0_r + S, ... ; entry [i'][co][ci]
    const int total = k * cin * cout;
    for (int idx = first; idx < total; idx += step) {
        const int ci = idx % cin;
        const int rest = idx / cin;
        const int co = rest % cout;
        const int slot = rest / cout;               // 0 .. k-1 over all phases
        // enumerate phases in order, taps of a phase by increasing i' (i.e. decreasing j)
        int base = 0, j = -1;
        for (int r = 0; r < stride && j < 0; ++r) {
            // taps of phase r: j == (r + pad_lo) mod S; jmax = the largest such j <= k-1 (may be < 0: empty phase)
            const int jmax = (k - 1) - (((k - 1) - ((r + pad_lo) % stride) + stride) % stride);
            const int cnt = jmax < 0 ? 0 : jmax / stride + 1;      // taps jmax, jmax - S, ..., >= 0
            if (slot < base + cnt) j = jmax - (slot - base) * stride;
            base += cnt;
        }
        wt[idx] = w[((size_t)j * cin + ci) * cout + co];
    }
}

__global__ __launch_bounds__(256) void dgrad_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int k, int cin, int cout,
                                                            int stride, int pad_lo) {
    dgrad_weights_body(w, wt, k, cin, cout, stride, pad_lo, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// every layer of a network in one launch: blockIdx.y = layer
__global__ __launch_bounds__(256) void dgrad_weights_multi_kernel(const DgradWeightsMulti m) {
    const DgradWeightsEntry e = m.e[blockIdx.y];
    dgrad_weights_body(e.w, e.wt, e.k, e.cin, e.cout, e.stride, e.pad_lo, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// The backward's first launch (TC-ResNet): blockIdx.y == 0 -- the head's backward (dpool[b][c] = (sum_o dlogits[b][o] Wfc[c][o]) dscale[b][c]:
// head_bwd_kernel's arithmetic) and the zero fill of the gradient arena; blockIdx.y = 1 + layer -- that layer's re-arranged data-gradient
// weights.  As three launches (a hipMemsetAsync, head_bwd_kernel, dgrad_weights_multi_kernel) they were ~18 us of kernels plus two ~7 us
// dispatch gaps between the forward's last kernel and the backward's first reduction.
__global__ __launch_bounds__(256) void bwd_prologue_kernel(const BwdPrologueArgs h, const DgradWeightsMulti m) {
    if (blockIdx.y > 0) {
        const DgradWeightsEntry e = m.e[blockIdx.y - 1];
        dgrad_weights_body(e.w, e.wt, e.k, e.cin, e.cout, e.stride, e.pad_lo, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
        return;
    }
    const int64_t total = (int64_t)h.batch * h.c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % h.c);
        const int64_t b = i / h.c;
        float s = 0.f;
        for (int o = 0; o < h.nc; ++o) s = fmaf(h.dlogits[b * h.nc + o], h.wfc[(size_t)ch * h.nc + o], s);
        h.dpool[i] = s * h.dscale[i];
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < h.zero_n; i += (int64_t)gridDim.x * 256) h.zero[i] = 0.f;
}

int launch_bwd_prologue(const BwdPrologueArgs& h, const DgradWeightsMulti& m, hipStream_t s) {
    int most = 0;
    for (int i = 0; i < m.n; ++i) most = max(most, m.e[i].k * m.e[i].cin * m.e[i].cout);
    int64_t blocks = ceil_div64((int64_t)h.batch * h.c, 256);
    if (blocks < ceil_div(most, 256)) blocks = ceil_div(most, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(bwd_prologue_kernel, dim3((unsigned)blocks, 1 + m.n), dim3(256), 0, s, h, m);
    return check_launch("bwd_prologue_kernel");
}

int launch_dgrad_weights_multi(const DgradWeightsMulti& m, hipStream_t s) {
    if (m.n <= 0) return TCR_OK;
    int most = 0;
    for (int i = 0; i < m.n; ++i) most = max(most, m.e[i].k * m.e[i].cin * m.e[i].cout);
    hipLaunchKernelGGL(dgrad_weights_multi_kernel, dim3(min(ceil_div(most, 256), 64), m.n), dim3(256), 0, s, m);
    return check_launch("dgrad_weights_multi_kernel");
}

bool conv_dgrad_mfma_covers(int k, int stride, int cout) {
    return !(cout % 4 != 0 || stride < 1 || stride > 2 || k > 9) && tune_get(TCR_TUNE_CONV_PATH) != 1;
}

// bit r set: output phase r (positions S*u + r of dx) receives taps of this conv
unsigned conv_dgrad_phases(int k, int stride, int pad_lo, int tin) {
    unsigned m = 0;
    for (int r = 0; r < stride; ++r) {
        const int jmax = (k - 1) - (((k - 1) - ((r + pad_lo) % stride) + stride) % stride);
        if (jmax >= 0 && tin - r > 0) m |= 1u << r;
    }
    return m;
}

int launch_conv_dgrad_mfma(int k, int stride, int pad_lo, const float* w, float* wt, const float* dy, float* dx, const float* add,
                           const float* add_mask, int add_bcast, int batch, int cin, int cout, int tin, int tout, hipStream_t s,
                           bool wt_ready, unsigned add_phases) {
    if (!conv_dgrad_mfma_covers(k, stride, cout)) return 1;
    if (!wt_ready) {
        hipLaunchKernelGGL(dgrad_weights_kernel, dim3(ceil_div(k * cin * cout, 256)), dim3(256), 0, s, w, wt, k, cin, cout, stride, pad_lo);
        TCR_TRY(check_launch("dgrad_weights_kernel"));
    }
    int base = 0;
    for (int r = 0; r < stride; ++r) {
        const int res_mod = (r + pad_lo) % stride;
        const int jmax = (k - 1) - (((k - 1) - res_mod + stride) % stride);
        if (jmax < 0) continue;
        const int cnt = jmax / stride + 1;
        const int d_min = (r + pad_lo - jmax) / stride;      // (exact division: same residue)
        const int nu = (tin - r + stride - 1) / stride;      // positions tin = S*u + r < Tin
        if (nu <= 0) { base += cnt; continue; }
        ConvArgs a;
        std::memset(&a, 0, sizeof(a));
        a.x = dy; a.w = wt + (size_t)base * cout * cin; a.y = dx;
        a.npos = batch * nu; a.cin = cout; a.cout = cin;
        a.tpi = tcr_padded_len(tout); a.tout = nu; a.tpo = tcr_padded_len(tin);
        a.xoff = kHalo + d_min; a.relu = 0;
        a.ostride = stride; a.ooff = r;
        if (add_phases >> r & 1u) { a.add = add; a.add_mask = add_mask; a.add_bcast = add_bcast; }
        int rc;
        switch (cnt) {
            case 1: rc = launch_conv_mfma_ks<1, 1>(a, nullptr, EPI_RAW, s); break;
            case 3: rc = launch_conv_mfma_ks<3, 1>(a, nullptr, EPI_RAW, s); break;
            case 4: rc = launch_conv_mfma_ks<4, 1>(a, nullptr, EPI_RAW, s); break;
            case 5: rc = launch_conv_mfma_ks<5, 1>(a, nullptr, EPI_RAW, s); break;
            case 9: rc = launch_conv_mfma_ks<9, 1>(a, nullptr, EPI_RAW, s); break;
            default: set_error("dgrad: %d-tap phase has no instantiation", cnt); return TCR_ERR_ARG;
        }
        if (rc != TCR_OK) return rc == 1 ? TCR_ERR_ARG : rc;
        base += cnt;
    }
    return TCR_OK;
}

// 9x1 stride-2 conv + the block's 1x1 stride-2 "down" conv in one launch (shared LDS image).
int launch_conv_mfma_with_down(const ConvArgs& a, const float* w_down, float* y_down, const float* scale_down,
                               const float* shift_down, int pad_lo, int epi, hipStream_t s) {
    ConvDownArgs d;
    d.w = w_down; d.y = y_down; d.scale = scale_down; d.shift = shift_down; d.tap = pad_lo; d.relu = 1;
    return launch_conv_mfma_ks<9, 2>(a, &d, epi, s);
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: dW[j][ci][co] = sum_{b,t} x[b][ci][t*S + j - pad_lo] * dy[b][co][t].
// Per wave: one 16-channel ci tile, all K taps, NCO co tiles; the reduction index (positions) is the
// MFMA k dimension, 4 positions per instruction.  Workgroups split the position range (split-K);
// the 4 waves of a workgroup are combined through LDS and every workgroup writes one partial
// [K][16][Cout] slab that a second kernel sums in a fixed order (bitwise reproducible).
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x;         // [B][Cin][Tpi]   (input activation of the conv, zero halo)
    const float* dy;        // [B][Cout][Tpo]  (gradient wrt the conv output)
    float* partial;         // [nchunk][K][Cin_pad][Cout_pad]
    int batch;
    int cin, cout, cin_pad, cout_pad;   // cout = width of this output-channel slice
    int cout_all, co_base;              // full channel count of dy / first channel of the slice
    int tpi, tout, tpo, stride;
    int xoff;               // HALO - pad_lo
    int utt_per_block;
    int pcol = 0;           // first column of this launch inside the [.][Cout_pad] slab rows (a layer split into channel slices that share one slab)
    WgradFly fly;           // (16-byte-load kernel) dy computed where it is loaded: `dy` is the unit's gz, fly.raw its raw conv output
};

template <int K, int NCO>
__global__ __launch_bounds__(256) void conv_wgrad_mfma_kernel(const WgradArgs a) {
    __shared__ float s_acc[K * 16 * NCO * 16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int ci = blockIdx.y * 16 + r;
    const bool civ = ci < a.cin;
    const int cic = civ ? ci : a.cin - 1;

    f32x4 acc[K][NCO];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int m = 0; m < NCO; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bool cov[NCO];
    int coc[NCO];
#pragma unroll
    for (int m = 0; m < NCO; ++m) {
        cov[m] = m * 16 + r < a.cout;
        coc[m] = a.co_base + (cov[m] ? m * 16 + r : 0);
    }

    // The reduction index is (utterance, t): the MFMA k dimension holds 4 consecutive t of one utterance
    // (lane group q), so operand addresses advance by plain increments -- no per-step division.
    // Lean addressing: the utterance base is wave-uniform (scalar registers), the per-lane part is a 32-bit offset that
    // never changes; no per-load predicates -- steps past the row end multiply dy's ZERO halo (bn_bwd_apply writes it)
    // with a clamped, finite x.
    const int n_begin = blockIdx.x * a.utt_per_block;
    const int n_end = min(n_begin + a.utt_per_block, a.batch);
    int doff[NCO];
#pragma unroll
    for (int m = 0; m < NCO; ++m) doff[m] = coc[m] * a.tpo + kHalo + q;
    const int xlane = cic * a.tpi + a.xoff;
    const int tlast = a.tout - 1;
    for (int n = n_begin + wave; n < n_end; n += 4) {
        const float* xr = a.x + (size_t)n * a.cin * a.tpi;
        const float* dr = a.dy + (size_t)n * a.cout_all * a.tpo;
#pragma unroll 2
        for (int t0 = 0; t0 < a.tout; t0 += 4) {
            float bf[NCO], af[K];
#pragma unroll
            for (int m = 0; m < NCO; ++m) bf[m] = dr[doff[m] + t0];
            const float* xt = xr + xlane + min(t0 + q, tlast) * a.stride;
#pragma unroll
            for (int j = 0; j < K; ++j) af[j] = xt[j];
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int m = 0; m < NCO; ++m) acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[m], acc[j][m], 0, 0, 0);
        }
    }
    // combine the 4 waves in LDS (fixed order), then write the slab
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int m = 0; m < NCO; ++m)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int row = q * 4 + reg;        // ci within the tile
                        const int idx = ((j * 16 + row) * NCO + m) * 16 + r;
                        if (wv == 0) s_acc[idx] = acc[j][m][reg];
                        else s_acc[idx] += acc[j][m][reg];
                    }
        }
        __syncthreads();
    }
    float* dst = a.partial + (size_t)blockIdx.x * K * a.cin_pad * a.cout_pad;
    for (int i = threadIdx.x; i < K * 16 * NCO * 16; i += 256) {
        const int col = i % (NCO * 16);
        const int row = (i / (NCO * 16)) % 16;
        const int j = i / (NCO * 16 * 16);
        const int cig = blockIdx.y * 16 + row;
        if (cig < a.cin_pad && a.pcol + col < a.cout_pad) dst[((size_t)j * a.cin_pad + cig) * a.cout_pad + a.pcol + col] = s_acc[i];
    }
}

// The same contraction with 16-byte operand loads.  The kernel above issues K + NCO gather loads (16 rows x 16 bytes each) per MFMA
// k-step and its waves -- one or two per SIMD: K x NCO accumulator tiles -- wait out every one of them: the TCResNet14-1.5 layers ran at
// a third of their matrix-pipe bound and the side stream that carries them became the step's critical path.  Here a trip covers 16
// positions: lane (r, q) takes positions t0 + 4 q + c (c = 0..3) of its row -- for dy ONE 16-byte load per channel tile, for x the
// 3 S + K floats that the K taps of its four positions touch ((3 S + K + 3) / 4 loads; tap j of position c is window[c S + j], a
// compile-time register) -- and feeds four k-steps (k-step c: lane group q supplies position 4 q + c in both operands); a remainder
// of at most 12 positions goes through 8-position trips (two positions per lane, two k-steps) so that short rows cost no extra
// k-steps.  Positions
// past the row are zeroed by selects in both operands (the loads run past the row into the next row / buffer: the caller guarantees
// x and dy are followed by readable memory -- workspace tensors; the first conv, whose x is the caller's feature buffer, keeps the
// kernel above).  Another summation order than the kernel above: results agree to rounding, not bitwise.
struct __attribute__((packed, aligned(4))) wg_f4u { float v[4]; };

// One trip over BS = 16 (8) positions starting at t0: lane (r, q) holds NPL = BS / 4 consecutive positions t0 + NPL q + c of its row.
// Rows whose length leaves 1..8 positions behind the 16-position trips finish with an 8-position trip (two k-steps): a 7-frame layer
// costs 2 k-steps, as with the 4-position steps of the kernel above, not 4.
template <int K, int S, int NCO, int BS, bool SAFE, bool FLY>
__device__ __forceinline__ void wgrad4_trip(f32x4 (&acc)[K][NCO], const float* xr, const float* dr, const int (&doff)[NCO], const bool (&cov)[NCO],
                                            bool civ, int q, int t0, int tout, bool safe, int xmax, const float* rr, const float (&kk)[NCO][6]) {
    constexpr int NPL = BS / 4, W = (NPL - 1) * S + K, NW4 = (W + 3) / 4;
    wg_f4u d4[NCO], w4[NW4];
#pragma unroll
    for (int m = 0; m < NCO; ++m) d4[m] = *reinterpret_cast<const wg_f4u*>(dr + doff[m] + t0 + NPL * q);      // (NPL = 2: the upper half is not used)
    if (FLY) {          // dy = k1 (dz - k2 - (raw - mean) k3), dz = gz [fmaf(raw, scale, shift) > 0]: bn_bwd_apply's expression, the lane's channels fixed
        wg_f4u r4[NCO];
#pragma unroll
        for (int m = 0; m < NCO; ++m) r4[m] = *reinterpret_cast<const wg_f4u*>(rr + doff[m] + t0 + NPL * q);
#pragma unroll
        for (int m = 0; m < NCO; ++m)
#pragma unroll
            for (int c = 0; c < NPL; ++c) {
                float g = d4[m].v[c];
                const float y = r4[m].v[c];
                if (!(fmaf(y, kk[m][4], kk[m][5]) > 0.f)) g = 0.f;
                d4[m].v[c] = kk[m][0] * (g - kk[m][1] - (y - kk[m][3]) * kk[m][2]);
            }
    }
    if (!SAFE || !safe) {
#pragma unroll
        for (int i = 0; i < NW4; ++i) w4[i] = *reinterpret_cast<const wg_f4u*>(xr + (t0 + NPL * q) * S + 4 * i);
    } else {            // (wave-uniform) the last rows of a buffer with nothing behind it: element loads clamped to the row
#pragma unroll
        for (int i = 0; i < NW4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) w4[i].v[e] = xr[min((t0 + NPL * q) * S + 4 * i + e, xmax)];
    }
#pragma unroll
    for (int c = 0; c < NPL; ++c) {
        if (t0 + c >= tout) break;                              // (wave-uniform: no lane has a position in this k-step)
        const bool pv = t0 + NPL * q + c < tout;
        float bf[NCO], af[K];
#pragma unroll
        for (int m = 0; m < NCO; ++m) bf[m] = (pv && cov[m]) ? d4[m].v[c] : 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) af[j] = (pv && civ) ? w4[(c * S + j) / 4].v[(c * S + j) % 4] : 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int m = 0; m < NCO; ++m) acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[m], acc[j][m], 0, 0, 0);
    }
}

// The same trip in two halves -- the operand loads and the arithmetic -- for the software-pipelined loop below (round 6).  The ISA of the
// one-piece trip is  loads -> s_waitcnt 0 -> K x NCO x NPL MFMAs  per trip: with 1 .. 1.5 waves on a SIMD (128 split-K chunks x 1 .. 3
// input tiles) every trip's L2 / HBM round trip stands in front of its MFMAs.  Same loads, same arithmetic, same order: bitwise the same.
// npl: positions per lane of the trip (4: a 16-position trip, 2: an 8-position trip) -- a run-time, wave-uniform value here, so that
// ONE code path serves both trip lengths (two paths selected by a branch made the compiler keep two copies of the K x NCO accumulator
// tiles and move all of them between the paths).  An 8-position trip loads the x window of a 16-position trip (up to one 16-byte load
// more than it needs, inside the readable slack behind x) and uses the same elements wgrad4_trip<.., 8, ..> does.
template <int K, int S, int NCO, bool FLY, int NW4M>
__device__ __forceinline__ void wgrad4_load(wg_f4u (&d4)[NCO], wg_f4u (&r4)[NCO], wg_f4u (&w4)[NW4M], const float* xr, const float* dr, const float* rr,
                                            const int (&doff)[NCO], int lq, int t0) {      // lq = npl * q
#pragma unroll
    for (int m = 0; m < NCO; ++m) d4[m] = *reinterpret_cast<const wg_f4u*>(dr + doff[m] + t0 + lq);
    if (FLY) {
#pragma unroll
        for (int m = 0; m < NCO; ++m) r4[m] = *reinterpret_cast<const wg_f4u*>(rr + doff[m] + t0 + lq);
    }
#pragma unroll
    for (int i = 0; i < NW4M; ++i) w4[i] = *reinterpret_cast<const wg_f4u*>(xr + (t0 + lq) * S + 4 * i);
}

template <int K, int S, int NCO, bool FLY, int NW4M>
__device__ __forceinline__ void wgrad4_mma(f32x4 (&acc)[K][NCO], wg_f4u (&d4)[NCO], const wg_f4u (&r4)[NCO], const wg_f4u (&w4)[NW4M], const bool (&cov)[NCO],
                                           bool civ, int lq, int npl, int t0, int tout, const float (&kk)[NCO][6]) {
    if (FLY) {          // (wgrad4_trip's expression; elements past the trip's positions are computed and never used)
#pragma unroll
        for (int m = 0; m < NCO; ++m)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float g = d4[m].v[c];
                const float y = r4[m].v[c];
                if (!(fmaf(y, kk[m][4], kk[m][5]) > 0.f)) g = 0.f;
                d4[m].v[c] = kk[m][0] * (g - kk[m][1] - (y - kk[m][3]) * kk[m][2]);
            }
    }
    const int nsteps = min(npl, tout - t0);                     // (wave-uniform: k-steps in which some lane has a position)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= nsteps) break;
        const bool pv = t0 + lq + c < tout;
        float bf[NCO], af[K];
#pragma unroll
        for (int m = 0; m < NCO; ++m) bf[m] = (pv && cov[m]) ? d4[m].v[c] : 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) af[j] = (pv && civ) ? w4[(c * S + j) / 4].v[(c * S + j) % 4] : 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int m = 0; m < NCO; ++m) acc[j][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[m], acc[j][m], 0, 0, 0);
    }
}

// NWV waves per workgroup (4, 8, 12 or 16): a wave takes every NWV-th utterance of the workgroup's chunk.  With four waves the 9-tap
// layers of TCResNet8 put 1 .. 1.5 waves on a SIMD (128 split-K chunks x 1 .. 3 input-channel tiles), and every trip's operand loads --
// an L2 / HBM round trip -- are waited out in front of its MFMAs; more waves per workgroup hide them without more slabs to reduce.  The
// waves are combined in LDS in a fixed order: wave w adds onto slab w / 4 in round w % 4, the slabs are added in order on the way out.
template <int K, int S, int NCO, bool SAFE, bool FLY, int NWV = 4, bool PIPE = false>
__global__ __launch_bounds__(NWV * 64) void conv_wgrad_mfma4_kernel(const WgradArgs a) {
    static_assert(!(PIPE && SAFE), "the clamped-load instantiation keeps the one-piece trips");
    constexpr int SLAB = K * 16 * NCO * 16, NG = NWV / 4;
    __shared__ float s_acc[NG * SLAB];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int ci = blockIdx.y * 16 + r;
    const bool civ = ci < a.cin;
    const int cic = civ ? ci : a.cin - 1;

    f32x4 acc[K][NCO];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int m = 0; m < NCO; ++m) acc[j][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bool cov[NCO];
    int doff[NCO];
#pragma unroll
    for (int m = 0; m < NCO; ++m) {
        cov[m] = m * 16 + r < a.cout;
        doff[m] = (a.co_base + (cov[m] ? m * 16 + r : 0)) * a.tpo + kHalo;
    }
    float kk[NCO][6];                               // FLY: k1, k2, k3, mean, own-mask scale / shift of the lane's dy channels
#pragma unroll
    for (int m = 0; m < NCO; ++m) {
        const int ch = a.co_base + (cov[m] ? m * 16 + r : 0);
        kk[m][0] = FLY ? a.fly.k1[ch] : 0.f; kk[m][1] = FLY ? a.fly.k2[ch] : 0.f; kk[m][2] = FLY ? a.fly.k3[ch] : 0.f;
        kk[m][3] = FLY ? a.fly.mean[ch] : 0.f;
        kk[m][4] = (FLY && a.fly.self_scale) ? a.fly.self_scale[ch] : 0.f;      // (no own mask: fmaf(raw, 0, 1) > 0 always)
        kk[m][5] = (FLY && a.fly.self_scale) ? a.fly.self_shift[ch] : 1.f;
    }
    const int n_begin = blockIdx.x * a.utt_per_block;
    const int n_end = min(n_begin + a.utt_per_block, a.batch);
    const int xlane = cic * a.tpi + a.xoff;
    const int xmax = a.tpi - 1 - a.xoff;            // last element of a row, relative to xr
    if constexpr (PIPE) {
        constexpr int NW4M = (3 * S + K + 3) / 4;
        wg_f4u dA[NCO], rA[NCO], wA[NW4M], dB[NCO], rB[NCO], wB[NW4M];
        const int tout = a.tout;
        // a wave's trips in order: utterance n_begin + wave (+ NWV ...), positions 0, 16, ... while more than 12 remain, then 8 at a time;
        // the operands of trip i + 1 are requested before the MFMAs of trip i (two register sets, the loop unrolled by two)
#define TCR_WG4_ISSUE(D_, R_, W_, N_, T0_)                                                                            \
        {                                                                                                             \
            const float* xr = a.x + (size_t)(N_) * a.cin * a.tpi + xlane;                                             \
            const float* dr = a.dy + (size_t)(N_) * a.cout_all * a.tpo;                                               \
            const float* rr = FLY ? a.fly.raw + (size_t)(N_) * a.cout_all * a.tpo : nullptr;                          \
            wgrad4_load<K, S, NCO, FLY, NW4M>(D_, R_, W_, xr, dr, rr, doff, (tout - (T0_) > 12 ? 4 : 2) * q, T0_);    \
        }
#define TCR_WG4_RUN(D_, R_, W_, T0_)                                                                                  \
        {                                                                                                             \
            const int npl = tout - (T0_) > 12 ? 4 : 2;                                                                \
            wgrad4_mma<K, S, NCO, FLY, NW4M>(acc, D_, R_, W_, cov, civ, npl * q, npl, T0_, tout, kk);                 \
        }
#define TCR_WG4_ADVANCE(N_, T0_)                                                                                      \
        {                                                                                                             \
            T0_ += tout - (T0_) > 12 ? 16 : 8;                                                                        \
            if (T0_ >= tout) { T0_ = 0; N_ += NWV; }                                                                  \
        }
        // (the request for "the trip after the last" repeats the last one instead of being skipped: a conditional request would
        //  leave the compiler's s_waitcnt counting to the worst case -- vmcnt(0) in front of every trip's MFMAs, i.e. no overlap)
        int n = n_begin + wave, t0 = 0;
        if (n < n_end) {
            TCR_WG4_ISSUE(dA, rA, wA, n, t0)
            for (;;) {
                int n1 = n, t1 = t0;
                TCR_WG4_ADVANCE(n1, t1)
                const bool h1 = n1 < n_end;
                { const int nl = h1 ? n1 : n, tl = h1 ? t1 : t0; TCR_WG4_ISSUE(dB, rB, wB, nl, tl) }
                TCR_WG4_RUN(dA, rA, wA, t0)
                if (!h1) break;
                n = n1; t0 = t1;
                TCR_WG4_ADVANCE(n, t0)
                const bool h2 = n < n_end;
                { const int nl = h2 ? n : n1, tl = h2 ? t0 : t1; TCR_WG4_ISSUE(dA, rA, wA, nl, tl) }
                TCR_WG4_RUN(dB, rB, wB, t1)
                if (!h2) break;
            }
        }
#undef TCR_WG4_ISSUE
#undef TCR_WG4_RUN
#undef TCR_WG4_ADVANCE
    } else {
        for (int n = n_begin + wave; n < n_end; n += NWV) {
            const float* xr = a.x + (size_t)n * a.cin * a.tpi + xlane;
            const float* dr = a.dy + (size_t)n * a.cout_all * a.tpo;
            const float* rr = FLY ? a.fly.raw + (size_t)n * a.cout_all * a.tpo : nullptr;
            const bool safe = SAFE && n == a.batch - 1;
            int t0 = 0;
            for (; a.tout - t0 > 12; t0 += 16) wgrad4_trip<K, S, NCO, 16, SAFE, FLY>(acc, xr, dr, doff, cov, civ, q, t0, a.tout, safe, xmax, rr, kk);
            for (; t0 < a.tout; t0 += 8) wgrad4_trip<K, S, NCO, 8, SAFE, FLY>(acc, xr, dr, doff, cov, civ, q, t0, a.tout, safe, xmax, rr, kk);
        }
    }
    // combine the waves in LDS (fixed order), then write the slab
    float* sa = s_acc + (wave >> 2) * SLAB;
    for (int wv = 0; wv < 4; ++wv) {
        if ((wave & 3) == wv) {
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int m = 0; m < NCO; ++m)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int row = q * 4 + reg;        // ci within the tile
                        const int idx = ((j * 16 + row) * NCO + m) * 16 + r;
                        if (wv == 0) sa[idx] = acc[j][m][reg];
                        else sa[idx] += acc[j][m][reg];
                    }
        }
        __syncthreads();
    }
    float* dst = a.partial + (size_t)blockIdx.x * K * a.cin_pad * a.cout_pad;
    for (int i = threadIdx.x; i < SLAB; i += NWV * 64) {
        const int col = i % (NCO * 16);
        const int row = (i / (NCO * 16)) % 16;
        const int j = i / (NCO * 16 * 16);
        const int cig = blockIdx.y * 16 + row;
        float v = s_acc[i];
#pragma unroll
        for (int g = 1; g < NG; ++g) v += s_acc[g * SLAB + i];
        if (cig < a.cin_pad && a.pcol + col < a.cout_pad) dst[((size_t)j * a.cin_pad + cig) * a.cout_pad + a.pcol + col] = v;
    }
}

// dw[j][ci][co] = sum_chunk partial[chunk][j][ci][co].  Four lanes per output walk interleaved chunk subsets and are
// combined with a fixed two-step shuffle tree, so the result is bitwise reproducible.
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ partial, float* __restrict__ dw,
                                                  int nchunk, int k, int cin, int cout, int cin_pad, int cout_pad,
                                                  int cout_all, int co_base) {
    const int total = k * cin * cout;
    if ((int)(blockIdx.x * 256) >= total * 4) return;
    const size_t slab = (size_t)k * cin_pad * cout_pad;
    const int part = threadIdx.x & 3;
    const int i = min((int)((blockIdx.x * 256 + threadIdx.x) >> 2), total - 1);
    const bool live = (int)((blockIdx.x * 256 + threadIdx.x) >> 2) < total;
    const int co = i % cout;
    const int r = i / cout;
    const int ci = r % cin;
    const int j = r / cin;
    const size_t off = ((size_t)j * cin_pad + ci) * cout_pad + co;
    // a lane's slabs are added in order, but their loads are independent: eight in flight per trip (one load per trip serialises a
    // memory round trip per slab: 32 of them for 128 slabs -- the step's LAST kernels, 20 us on an otherwise idle chip)
    float s = 0.f;
    int c = part;
    for (; c + 28 < nchunk; c += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(c + 4 * u) * slab + off];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < nchunk; c += 4) s += partial[(size_t)c * slab + off];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (live && part == 0) dw[((size_t)j * cin + ci) * cout_all + co_base + co] = s;
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                           int nchunk, int k, int cin, int cout, int cin_pad, int cout_pad,
                                                           int cout_all, int co_base) {
    wgrad_reduce_body(partial, dw, nchunk, k, cin, cout, cin_pad, cout_pad, cout_all, co_base);
}

// the slabs of every layer of a network in one launch: blockIdx.y = layer
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const WgradReduceMulti m) {
    const WgradReduceEntry e = m.e[blockIdx.y];
    wgrad_reduce_body(e.partial, e.dw, e.nchunk, e.k, e.cin, e.cout, e.cin_pad, e.cout_pad, e.cout, 0);
}

int launch_wgrad_reduce_multi(const WgradReduceMulti& m, hipStream_t s) {
    if (m.n <= 0) return TCR_OK;
    int most = 0;
    for (int i = 0; i < m.n; ++i) most = max(most, m.e[i].k * m.e[i].cin * m.e[i].cout);
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(ceil_div(most * 4, 256), m.n), dim3(256), 0, s, m);
    return check_launch("wgrad_reduce_multi_kernel");
}

int launch_wgrad_reduce(const float* partial, float* dw, int nchunk, int k, int cin, int cout, int cin_pad, int cout_pad,
                        int cout_all, int co_base, hipStream_t s) {
    const int total = k * cin * cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(total * 4, 256)), dim3(256), 0, s, partial, dw, nchunk, k, cin, cout, cin_pad,
                       cout_pad, cout_all, co_base);
    return check_launch("wgrad_reduce_kernel");
}

// ---------------------------------------------------------------------------------------------
// Pointwise (1x1, stride 1) weight gradient for wide layers (DS-CNN: 172 / 276 channels):
//   dW[ci][co] = sum_{n,p} x[n][ci][p] * dz[n][co][p]
// The slab kernel above re-reads dz once per 16-row ci tile and x once per 80-column slice -- 8 GB of L2/HBM traffic
// for a 0.33 GB operand pair at 276 channels (measured 2.9 ms, 14 TFLOP/s).  Here a workgroup owns a 96 x 96 block of
// dW and walks its utterances: the 96 x-rows and 96 dz-rows of an utterance are two CONTIGUOUS blocks of the planar
// layout, copied to LDS with coalesced loads and shared by the four waves (3 x 3 tiles each), so every operand
// element is fetched three times instead of 18 / 4.  The zero halo of the rows pads the position loop to a multiple of 4.
// ---------------------------------------------------------------------------------------------
struct PwWgradArgs {
    const float* x;         // [B][Cin][Pp]
    const float* dz;        // [B][Cout][Pp]
    float* partial;         // [nchunk][Cin_pad][Cout_pad]
    int batch, cin, cout, cin_pad, cout_pad, p, pp, utt_per_block;
    int nchunk, nby, nbz;   // launch geometry (see the workgroup -> (chunk, block) map in the kernel)
    // x is a RAW train-mode conv output (halo unwritten): the A operand is relu(x * x_scale[ci] + x_shift[ci]) for positions < p,
    // 0 past them -- applied to the fragment as it leaves LDS (a lane's three rows are fixed: six registers)
    const float* x_scale;
    const float* x_shift;
};

__global__ __launch_bounds__(256) void pw_wgrad_lds_kernel(const PwWgradArgs a) {
    constexpr int BT = 96;                                  // block edge: 6 MFMA tiles
    constexpr int NV = 14;                                  // float4 per thread per utterance: 2 * 96 * pp / 4 <= 14 * 256
    float* xs = reinterpret_cast<float*>(dyn_lds());
    float* ds = xs + BT * a.pp;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    // XCD-aware workgroup -> tile map.  The nby x nbz blocks of dW that share a chunk of utterances read the same x / dz rows, and
    // workgroups go to the eight XCDs (each with its own L2) round-robin by linear id: a 3-D grid put the nine blocks of a chunk on
    // different XCDs, far apart in time -- every operand came from HBM three times (10.1 GB per step for 3.3 GB of tensors).  Here
    // the workgroups of XCD k (ids k, k + 8, ...) take chunk 8 (j / nb) + k, block j % nb: a chunk's blocks are dispatched back to
    // back onto ONE XCD and walk the same utterances together, so all but the first read of a row hit that XCD's L2.
    const int nb = a.nby * a.nbz;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int chunk = (jj / nb) * 8 + xcd, blk = jj % nb;
    if (chunk >= a.nchunk) return;
    const int ci0 = (blk % a.nby) * BT, co0 = (blk / a.nby) * BT;
    const int xrows = min(BT, a.cin - ci0), drows = min(BT, a.cout - co0);
    const int xv = xrows * a.pp / 4, dv = drows * a.pp / 4;             // float4 counts (host checks divisibility)
    for (int i = tid; i < 2 * BT * a.pp; i += 256) xs[i] = 0.f;          // rows past the channel count stay zero
    const int wm = (wave >> 1) * 3, wn = (wave & 1) * 3;
    f32x4 acc[3][3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int ao[3], bo[3];
    float xsc[3], xsf[3];
    const bool xaff = a.x_scale != nullptr;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        ao[m] = ((wm + m) * 16 + r) * a.pp + kHalo + q;
        bo[m] = ((wn + m) * 16 + r) * a.pp + kHalo + q;
        const int ci = ci0 + (wm + m) * 16 + r;
        xsc[m] = (xaff && ci < a.cin) ? a.x_scale[ci] : 0.f;        // (rows past Cin: relu(0 * 0 + 0) = 0)
        xsf[m] = (xaff && ci < a.cin) ? a.x_shift[ci] : 0.f;
    }
    const int n_begin = chunk * a.utt_per_block;
    const int n_end = min(n_begin + a.utt_per_block, a.batch);
    // The next utterance's rows travel global -> registers while the current one is multiplied out of LDS.
    // (Every lane always loads from a valid address -- clamped past the end; 14 named registers rather than an array,
    // which the compiler kept in scratch memory.)
    float4 p0, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10, p11, p12, p13;
    const float4 *xg4, *dg4;
#define TCR_PW_LD(I_, P_) { const int v = tid + (I_) * 256; P_ = *(v < xv ? xg4 + v : dg4 + min(v - xv, dv - 1)); }
#define TCR_PW_ST(I_, P_) { const int v = tid + (I_) * 256; if (v < xv) xs4[v] = P_; else if (v - xv < dv) ds4[v - xv] = P_; }
#define TCR_PW_ALL(OP_) OP_(0, p0) OP_(1, p1) OP_(2, p2) OP_(3, p3) OP_(4, p4) OP_(5, p5) OP_(6, p6) OP_(7, p7) OP_(8, p8) OP_(9, p9) \
                        OP_(10, p10) OP_(11, p11) OP_(12, p12) OP_(13, p13)
#define TCR_PW_PREFETCH(N_)                                                                         \
    {                                                                                               \
        xg4 = reinterpret_cast<const float4*>(a.x + ((size_t)(N_) * a.cin + ci0) * a.pp);           \
        dg4 = reinterpret_cast<const float4*>(a.dz + ((size_t)(N_) * a.cout + co0) * a.pp);         \
        TCR_PW_ALL(TCR_PW_LD)                                                                       \
    }
    static_assert(NV == 14, "TCR_PW_ALL lists 14 registers");
    if (n_begin < n_end) TCR_PW_PREFETCH(n_begin)
    float4* xs4 = reinterpret_cast<float4*>(xs);
    float4* ds4 = reinterpret_cast<float4*>(ds);
    for (int n = n_begin; n < n_end; ++n) {
        __syncthreads();
        TCR_PW_ALL(TCR_PW_ST)
        __syncthreads();
        if (n + 1 < n_end) TCR_PW_PREFETCH(n + 1)
        float af[3], bf[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) { af[m] = xs[ao[m]]; bf[m] = ds[bo[m]]; }
        for (int k0 = 0; k0 < a.p; k0 += 4) {
            float an[3], bn[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) { an[m] = xs[ao[m] + k0 + 4]; bn[m] = ds[bo[m] + k0 + 4]; }   // (past the end: next row's halo / pad)
            if (xaff) {
                const bool in = k0 + q < a.p;
#pragma unroll
                for (int m = 0; m < 3; ++m) af[m] = in ? fmaxf(fmaf(af[m], xsc[m], xsf[m]), 0.f) : 0.f;
            }
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int nn = 0; nn < 3; ++nn) acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[nn], acc[m][nn], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 3; ++m) { af[m] = an[m]; bf[m] = bn[m]; }
        }
    }
#undef TCR_PW_PREFETCH
#undef TCR_PW_ALL
#undef TCR_PW_ST
#undef TCR_PW_LD
    float* dst = a.partial + (size_t)chunk * a.cin_pad * a.cout_pad;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int nn = 0; nn < 3; ++nn)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int ci = ci0 + (wm + m) * 16 + q * 4 + reg, co = co0 + (wn + nn) * 16 + r;
                if (ci < a.cin_pad && co < a.cout_pad) dst[(size_t)ci * a.cout_pad + co] = acc[m][nn][reg];
            }
}

// The same kernel for a compile-time map size (DS-CNN: P = 65 positions, rows of 73 floats), round 5.  The run-time kernel above spends 4.1
// VALU instructions per MFMA -- six address adds and six register moves per 4-position step (a rolled loop with a one-step lookahead), the
// x-affine's compare / select in every step, and per staged float4 a pointer select, a clamp and a branch pair around the LDS store --
// and on this chip the VALU and matrix instructions of a SIMD's waves share its issue: the kernel's time fits MFMA cycles + 4 x VALU
// instructions, not their maximum.  Here the 17 steps are unrolled (LDS reads at immediate offsets, no moves, the tail mask in the last
// step only) and the staging roles are fixed per thread: float4 i < NI of a thread belong to the x block, the rest to the dz block (each
// block owns NI x 256 float4 of LDS; a thread's clamped duplicates of a block's last float4 are stored where that float4 goes anyway),
// so a staged float4 costs its load and its store.  Same MFMA order per accumulator: bitwise the kernel above.
template <int P, bool XAFF>
__global__ __launch_bounds__(256) void pw_wgrad_lds_p_kernel(const PwWgradArgs a) {
    constexpr int BT = 96, PP = P + 2 * kHalo, KS = (P + 3) / 4;
    constexpr int B4 = BT * PP / 4, NI = (B4 + 255) / 256, SLOT = NI * 256;         // float4 per block / per thread and block / LDS float4 per block
    static_assert((BT * PP) % 4 == 0 && KS * 4 <= P + kHalo, "rows end on a float4; the last step reads into the halo only");
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4* xs4 = reinterpret_cast<f4*>(dyn_lds());
    f4* ds4 = xs4 + SLOT;
    const float* xs = reinterpret_cast<const float*>(xs4);
    const float* ds = reinterpret_cast<const float*>(ds4);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int nb = a.nby * a.nbz;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int chunk = (jj / nb) * 8 + xcd, blk = jj % nb;       // (XCD-aware map: see pw_wgrad_lds_kernel)
    if (chunk >= a.nchunk) return;
    const int ci0 = (blk % a.nby) * BT, co0 = (blk / a.nby) * BT;
    const int xrows = min(BT, a.cin - ci0), drows = min(BT, a.cout - co0);
    const int xv = xrows * PP / 4, dv = drows * PP / 4;         // (host checks divisibility)
#pragma unroll
    for (int i = 0; i < 2 * NI; ++i) xs4[tid + 256 * i] = (f4){0.f, 0.f, 0.f, 0.f};     // rows past the channel count stay zero
    const int wm = (wave >> 1) * 3, wn = (wave & 1) * 3;
    f32x4 acc[3][3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int ao[3], bo[3];
    float xsc[3], xsf[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        ao[m] = ((wm + m) * 16 + r) * PP + kHalo + q;
        bo[m] = ((wn + m) * 16 + r) * PP + kHalo + q;
        const int ci = ci0 + (wm + m) * 16 + r;
        xsc[m] = (XAFF && ci < a.cin) ? a.x_scale[ci] : 0.f;    // (rows past Cin: relu(0 * 0 + 0) = 0)
        xsf[m] = (XAFF && ci < a.cin) ? a.x_shift[ci] : 0.f;
    }
    int xo[NI], dof[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) { xo[i] = min(tid + 256 * i, xv - 1); dof[i] = min(tid + 256 * i, dv - 1); }
    const int n_begin = chunk * a.utt_per_block;
    const int n_end = min(n_begin + a.utt_per_block, a.batch);
    f4 px[NI], pd[NI];
    auto prefetch = [&](int n) {
        const f4* xg4 = reinterpret_cast<const f4*>(a.x + ((size_t)n * a.cin + ci0) * PP);
        const f4* dg4 = reinterpret_cast<const f4*>(a.dz + ((size_t)n * a.cout + co0) * PP);
#pragma unroll
        for (int i = 0; i < NI; ++i) { px[i] = xg4[xo[i]]; pd[i] = dg4[dof[i]]; }
    };
    if (n_begin < n_end) prefetch(n_begin);
    for (int n = n_begin; n < n_end; ++n) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i) { xs4[xo[i]] = px[i]; ds4[dof[i]] = pd[i]; }
        __syncthreads();
        if (n + 1 < n_end) prefetch(n + 1);
        float af[3], bf[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) { af[m] = xs[ao[m]]; bf[m] = ds[bo[m]]; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float an[3], bn[3];
            if (ks + 1 < KS) {
#pragma unroll
                for (int m = 0; m < 3; ++m) { an[m] = xs[ao[m] + 4 * (ks + 1)]; bn[m] = ds[bo[m] + 4 * (ks + 1)]; }
            }
            if (XAFF) {
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    af[m] = fmaxf(fmaf(af[m], xsc[m], xsf[m]), 0.f);
                    if (4 * ks + 3 >= P) af[m] = 4 * ks + q < P ? af[m] : 0.f;     // (the last step's lanes past the map)
                }
            }
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int nn = 0; nn < 3; ++nn) acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[nn], acc[m][nn], 0, 0, 0);
            if (ks + 1 < KS) {
#pragma unroll
                for (int m = 0; m < 3; ++m) { af[m] = an[m]; bf[m] = bn[m]; }
            }
        }
    }
    float* dst = a.partial + (size_t)chunk * a.cin_pad * a.cout_pad;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int nn = 0; nn < 3; ++nn)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int ci = ci0 + (wm + m) * 16 + q * 4 + reg, co = co0 + (wn + nn) * 16 + r;
                if (ci < a.cin_pad && co < a.cout_pad) dst[(size_t)ci * a.cout_pad + co] = acc[m][nn][reg];
            }
}

// ---------------------------------------------------------------------------------------------
// The same filter gradient with the operand blocks copied global -> LDS by the DMA path (global_load_lds_dwordx4: no staging registers,
// no ds_write pass) into TWO buffers: while the twelve waves of a workgroup multiply utterance n out of one buffer, utterance n + 1
// lands in the other; one barrier per utterance.  One workgroup per CU (112 KB of LDS), three waves per SIMD: three groups of four
// waves, every group the 2 x 2 arrangement of 3 x 3-tile waves of the kernel above (six LDS reads feed nine MFMAs) on every third
// 4-position step of the utterance (rotating with the utterance: 17 steps = 6 + 6 + 5); the groups' accumulators meet in LDS at the end.
// Round 4's kernel (above: register-staged, one buffer, two 4-wave workgroups per CU, two barriers and a 57 KB ds_write pass per
// utterance) spent ~45 % of its time outside the MFMA loop.
// The LDS image of a stage is the x block followed DIRECTLY by the dz block (a DMA instruction writes base + lane * 16: the image is
// one contiguous run of float4); rows past the block's channel count: x rows are masked where the fragment leaves LDS, dz rows read the
// zeroed space behind the image.
// ---------------------------------------------------------------------------------------------
template <bool XAFF>
__global__ __launch_bounds__(768) void pw_wgrad_glds_kernel(const PwWgradArgs a) {
    constexpr int BT = 96, NW = 12, NG = 3;
    float* lds = reinterpret_cast<float*>(dyn_lds());
    const int stage = 2 * BT * a.pp;                        // floats per buffer
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int grp = wave >> 2, w4 = wave & 3;
    // XCD-aware workgroup -> (chunk, block) map of pw_wgrad_lds_kernel: a chunk's blocks run on ONE XCD and share its L2
    const int nb = a.nby * a.nbz;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int chunk = (jj / nb) * 8 + xcd, blk = jj % nb;
    if (chunk >= a.nchunk) return;
    const int ci0 = (blk % a.nby) * BT, co0 = (blk / a.nby) * BT;
    const int xrows = min(BT, a.cin - ci0), drows = min(BT, a.cout - co0);
    const int xv = xrows * a.pp / 4, tv = xv + drows * a.pp / 4;        // float4 counts: x block, whole image (host checks divisibility)
    for (int i = tid; i < 2 * stage; i += 64 * NW) lds[i] = 0.f;        // (the space behind an image stays zero: dz rows past Cout)
    const int wm = (w4 >> 1) * 3, wn = (w4 & 1) * 3;
    f32x4 acc[3][3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int ao[3], bo[3];
    float xsc[3], xsf[3];
    bool aok[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int arow = (wm + m) * 16 + r;
        aok[m] = arow < xrows;
        ao[m] = arow * a.pp + kHalo + q;
        bo[m] = xrows * a.pp + ((wn + m) * 16 + r) * a.pp + kHalo + q;
        xsc[m] = (XAFF && aok[m]) ? a.x_scale[ci0 + arow] : 0.f;
        xsf[m] = (XAFF && aok[m]) ? a.x_shift[ci0 + arow] : 0.f;
    }
    const int n_begin = chunk * a.utt_per_block;
    const int n_end = min(n_begin + a.utt_per_block, a.batch);
    const int nks = (a.p + 3) >> 2;                         // 4-position steps per utterance
    // one stage: DMA instruction i of this wave covers float4 [(i * NW + wave) * 64, + 64) of the image
    auto issue = [&](int n, int buf) {
        const float4* xg4 = reinterpret_cast<const float4*>(a.x + ((size_t)n * a.cin + ci0) * a.pp);
        const float4* dg4 = reinterpret_cast<const float4*>(a.dz + ((size_t)n * a.cout + co0) * a.pp);
        float* dst = lds + buf * stage;
        for (int v0 = wave * 64; v0 < tv; v0 += 64 * NW) {
            const int v = v0 + lane;
            if (v < tv) glds16(v < xv ? xg4 + v : dg4 + (v - xv), dst + v0 * 4);
        }
    };
    __syncthreads();                                        // the zeroed buffers
    if (n_begin < n_end) issue(n_begin, 0);
    int rot = grp;                                          // this group's first step of the current utterance: (grp - i) mod 3
    for (int n = n_begin; n < n_end; ++n) {
        const int cur = (n - n_begin) & 1;
        wait_dma();                                         // this wave's share of stage n has landed ...
        __syncthreads();                                    // ... everybody's has; nobody still reads the other buffer
        if (n + 1 < n_end) issue(n + 1, cur ^ 1);
        const float* xs = lds + cur * stage;
        if (rot < nks) {
            float af[3], bf[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) { af[m] = xs[ao[m] + 4 * rot]; bf[m] = xs[bo[m] + 4 * rot]; }
            for (int ks = rot; ks < nks; ks += NG) {
                const int k0 = 4 * ks;
                float an[3], bn[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) { an[m] = xs[ao[m] + k0 + 4 * NG]; bn[m] = xs[bo[m] + k0 + 4 * NG]; }   // (past the end: rows / zeroed space behind; never used)
                const bool in = k0 + q < a.p;
#pragma unroll
                for (int m = 0; m < 3; ++m) af[m] = XAFF ? ((in && aok[m]) ? fmaxf(fmaf(af[m], xsc[m], xsf[m]), 0.f) : 0.f) : (aok[m] ? af[m] : 0.f);
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int nn = 0; nn < 3; ++nn) acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[nn], acc[m][nn], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 3; ++m) { af[m] = an[m]; bf[m] = bn[m]; }
            }
        }
        rot = rot == 0 ? NG - 1 : rot - 1;
    }
    // the three groups' partial tiles: groups 1, 2 through LDS (fixed order: group 0 + group 1 + group 2)
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(lds);
    if (grp > 0) {
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int nn = 0; nn < 3; ++nn) red[(((grp - 1) * 4 + w4) * 9 + m * 3 + nn) * 64 + lane] = acc[m][nn];
    }
    __syncthreads();
    if (grp > 0) return;
    float* dst = a.partial + (size_t)chunk * a.cin_pad * a.cout_pad;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int nn = 0; nn < 3; ++nn) {
            f32x4 t = acc[m][nn];
#pragma unroll
            for (int g = 0; g < NG - 1; ++g) {
                const f32x4 o = red[((g * 4 + w4) * 9 + m * 3 + nn) * 64 + lane];
                t[0] += o[0]; t[1] += o[1]; t[2] += o[2]; t[3] += o[3];
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int ci = ci0 + (wm + m) * 16 + q * 4 + reg, co = co0 + (wn + nn) * 16 + r;
                if (ci < a.cin_pad && co < a.cout_pad) dst[(size_t)ci * a.cout_pad + co] = t[reg];
            }
        }
}

// Chunks of utterances (split-K slabs).  The workgroups of a chunk -- one per 96 x 96 block of dW -- land on ONE XCD (see the kernel's
// map), two per CU: a count that fills a whole number of rounds of an XCD's slots.  (Round 4 used ceil(batch / 36): at batch 4096 and
// nine blocks that is 1026 workgroups on 1024 slots -- XCDs 0 and 1 ran a third round for one extra chunk each: +50 % on the kernel.)
static int pw_wgrad_chunks(int batch, int cin, int cout, int wg_per_cu) {
    const int nb = ceil_div(cin, 96) * ceil_div(cout, 96);
    const int slots_per_xcd = max(1, wg_per_cu * device_cus() / 8);
    const int one = 8 * max(1, slots_per_xcd / nb), two = 8 * max(1, 2 * slots_per_xcd / nb);      // chunks that fill one / two rounds
    int n = ceil_div(batch, one) > 48 ? two : one;         // (two rounds once a chunk would hold more than 48 utterances: shorter tail)
    if (n > 128) n = 128;
    if (n > batch) n = batch;
    return n < 1 ? 1 : n;
}
// round 4's register-staged kernel unless the knob asks for the DMA-staged one (measured: 573 vs 561 us per launch at 276 channels, 258 vs
// 249 at 172 -- with one stage of lookahead both sit at ~3.8 us per utterance and CU, the operand blocks' L2 / HBM round trip, not the
// 2.0 us of matrix work; a third LDS buffer does not fit: OPTLOG round 5)
static bool pw_wgrad_use_glds() { return tune_get(TCR_TUNE_PW_WGRAD) == 1; }
static int pw_wgrad_chunks(int batch, int cin, int cout) {      // (scratch sizing: the larger of the two kernels' slab counts)
    return max(pw_wgrad_chunks(batch, cin, cout, 1), pw_wgrad_chunks(batch, cin, cout, 2));
}

static bool pw_wgrad_fits(int k, int stride, int cin, int cout, int tpi, int tpo) {
    // rows are copied as float4: every 96-row block of an utterance must start and end on a 16-byte boundary, and the
    // two blocks must fit the kernel's 14 float4 per thread
    return k == 1 && stride == 1 && tpi == tpo && cin > 80 && cout > 80 && cin % 4 == 0 && cout % 4 == 0 && 2 * 96 * tpi <= 14 * 256 * 4;
}

bool pw_wgrad_lds_covers(int cin, int cout, int tp) { return pw_wgrad_fits(1, 1, cin, cout, tp, tp); }

static int launch_pw_wgrad_lds(const float* x, const float* dy, float* dw, float* scratch, int batch, int cin, int cout, int tpi, int tout,
                               hipStream_t s, const float* x_scale, const float* x_shift) {
    PwWgradArgs a;
    a.x_scale = x_scale; a.x_shift = x_shift;
    a.x = x; a.dz = dy; a.partial = scratch; a.batch = batch; a.cin = cin; a.cout = cout;
    a.cin_pad = ceil_div(cin, 16) * 16; a.cout_pad = ceil_div(cout, 16) * 16; a.p = tout; a.pp = tpi;
    const bool glds = pw_wgrad_use_glds();
    a.utt_per_block = ceil_div(batch, pw_wgrad_chunks(batch, cin, cout, glds ? 1 : 2));
    if (glds) {
        const size_t lds2 = ((size_t)4 * 96 * tpi + 16) * sizeof(float);  // two buffers (+ pad: the one-step operand lookahead of the last row)
        static size_t configured2 = 0;
        if (lds2 > configured2) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(pw_wgrad_glds_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(pw_wgrad_glds_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess) {
                set_error("pw_wgrad_glds_kernel: cannot reserve %zu bytes of LDS", lds2);
                return TCR_ERR_HIP;
            }
            configured2 = lds2;
        }
        a.nchunk = ceil_div(batch, a.utt_per_block); a.nby = ceil_div(cin, 96); a.nbz = ceil_div(cout, 96);
        const dim3 grid2(ceil_div(a.nchunk, 8) * 8 * a.nby * a.nbz);
        if (x_scale) hipLaunchKernelGGL(pw_wgrad_glds_kernel<true>, grid2, dim3(768), lds2, s, a);
        else hipLaunchKernelGGL(pw_wgrad_glds_kernel<false>, grid2, dim3(768), lds2, s, a);
        TCR_TRY(check_launch("pw_wgrad_glds_kernel"));
        return launch_wgrad_reduce(scratch, dw, a.nchunk, 1, cin, cout, a.cin_pad, a.cout_pad, cout, 0, s);
    }
    const size_t lds = ((size_t)2 * 96 * tpi + 16) * sizeof(float);      // (+ pad: the one-step operand lookahead)
    static size_t configured = 0;
    if (lds > 64 * 1024 && lds > configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(pw_wgrad_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error("pw_wgrad_lds_kernel: cannot reserve %zu bytes of LDS", lds);
            return TCR_ERR_HIP;
        }
        configured = lds;
    }
    a.nchunk = ceil_div(batch, a.utt_per_block); a.nby = ceil_div(cin, 96); a.nbz = ceil_div(cout, 96);
    const dim3 grid(ceil_div(a.nchunk, 8) * 8 * a.nby * a.nbz);
    if (tout == 65 && tpi == 65 + 2 * kHalo && tune_get(TCR_TUNE_PW_WGRAD) != 2) {      // DS-CNN's 13 x 5 maps: the unrolled kernel
        const size_t ldsp = (size_t)2 * 7 * 256 * 16;
        static bool configured_p = false;
        if (!configured_p) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(pw_wgrad_lds_p_kernel<65, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(pw_wgrad_lds_p_kernel<65, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp) != hipSuccess) {
                set_error("pw_wgrad_lds_p_kernel: cannot reserve %zu bytes of LDS", ldsp);
                return TCR_ERR_HIP;
            }
            configured_p = true;
        }
        if (x_scale) hipLaunchKernelGGL((pw_wgrad_lds_p_kernel<65, true>), grid, dim3(256), ldsp, s, a);
        else hipLaunchKernelGGL((pw_wgrad_lds_p_kernel<65, false>), grid, dim3(256), ldsp, s, a);
        TCR_TRY(check_launch("pw_wgrad_lds_p_kernel"));
        return launch_wgrad_reduce(scratch, dw, a.nchunk, 1, cin, cout, a.cin_pad, a.cout_pad, cout, 0, s);
    }
    hipLaunchKernelGGL(pw_wgrad_lds_kernel, grid, dim3(256), lds, s, a);
    TCR_TRY(check_launch("pw_wgrad_lds_kernel"));
    return launch_wgrad_reduce(scratch, dw, a.nchunk, 1, cin, cout, a.cin_pad, a.cout_pad, cout, 0, s);
}

// ---------------------------------------------------------------------------------------------
// LDS-staged filter gradient (round 4).  The 16-byte-load kernel above gives every wave whole utterances: each trip is a global
// round trip in front of its MFMAs, a wave accumulates ALL K x NCO tiles (72 .. 108 accumulator registers: one or two waves per SIMD),
// and x / dy are fetched once per 16-row input tile.  Here a workgroup of NINE waves walks its chunk of utterances in stages of `ub`:
//   STAGE   the stage's x rows (Cin x Tp) and dy rows (Cout x Tp) are CONTIGUOUS blocks of the planar layout: coalesced 16-byte loads
//           into registers one stage ahead, then into the other LDS buffer (one barrier per stage).  dy is built while it is staged when
//           the unit's BN backward is applied on the fly (WgradFly: k1 (dz - k2 - (raw - mean) k3), bn_bwd_apply's expression; the
//           coefficients sit in an LDS table), and every position outside [0, T) is stored as zero.
//   MFMA    wave w owns tap j = w of a 9-tap filter with all its (input tile, output tile) pairs -- for the 3-tap first conv (tap, input
//           tile) = (w % 3, w / 3) -- : per 4 positions CPW A fragments (x at the tap's offset) + NCO B fragments (dy) from LDS feed
//           CPW x NCO MFMAs.  No wave shares an accumulator with another: the slab is written straight from the registers.
// D[row = ci][col = co] as in the kernels above, same slab layout, same deferred reduction.  Another summation order than the kernels
// above (utterances in order within a workgroup): results agree to rounding.  Built for the first conv only (wgrad_lds_instance).
// ---------------------------------------------------------------------------------------------
struct WgradLdsArgs {
    const float* x;         // [B][Cin][Tpi]
    const float* dy;        // [B][Cout][Tpo] (fly.raw: the unit's gz)
    float* partial;         // [nchunk][K][Cin_pad][Cout_pad]
    int batch, cin, cout, cin_pad, cout_pad, tpi, tout, tpo, stride, xoff, utt_per_block, ub;
    WgradFly fly;
};

template <int K, int NCI, int NCO, bool FLY>
__global__ __launch_bounds__(576) void conv_wgrad_lds_kernel(const WgradLdsArgs a) {
    constexpr int NW = 9, NT = NW * 64, XI = 2;
    constexpr int CPW = K == 9 ? NCI : 1;               // input-channel tiles per wave
    static_assert(K == 9 || (K == 3 && NCI == 3), "wave <-> (tap, input tile) maps");
    float* lds = reinterpret_cast<float*>(dyn_lds());
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int xsz = a.cin * a.tpi, dsz = a.cout * a.tpo;            // floats per utterance (multiples of 4: launcher)
    const int bufsz = a.ub * (xsz + dsz);
    float* tab = lds + 2 * bufsz;                                   // FLY: [cout][8] = k1, k2, k3, mean, own-mask scale / shift
    if (FLY) {
        for (int i = tid; i < a.cout; i += NT) {
            tab[i * 8 + 0] = a.fly.k1[i]; tab[i * 8 + 1] = a.fly.k2[i]; tab[i * 8 + 2] = a.fly.k3[i]; tab[i * 8 + 3] = a.fly.mean[i];
            tab[i * 8 + 4] = a.fly.self_scale ? a.fly.self_scale[i] : 0.f;          // (no own mask: fmaf(raw, 0, 1) > 0 always)
            tab[i * 8 + 5] = a.fly.self_scale ? a.fly.self_shift[i] : 1.f;
            tab[i * 8 + 6] = 0.f; tab[i * 8 + 7] = 0.f;
        }
    }
    const int n_begin = blockIdx.x * a.utt_per_block;
    const int n_end = min(n_begin + a.utt_per_block, a.batch);
    const int nstages = (n_end - n_begin + a.ub - 1) / a.ub;
    const float inv_dsz = 1.0f / (float)dsz, inv_tpo = 1.0f / (float)a.tpo;

    // Staging registers in TWO sets (round 6, LA2 below): stage st + 2 is requested before stage st's MFMAs, so a stage's HBM round trip has
    // two MFMA phases to hide behind (one workgroup of nine waves per CU: nothing else covers it).  The stage loop is unrolled by two so
    // that the sets are compile-time indices.
    f32x4 xr2[2][XI], gr2[2][XI], rr2[2][XI];
    auto load_stage = [&](int st, f32x4 (&xr)[XI], f32x4 (&gr)[XI], f32x4 (&rr)[XI]) {
        const int n0 = n_begin + st * a.ub;
        const int nu = min(a.ub, n_end - n0);
        const f32x4* xs = reinterpret_cast<const f32x4*>(a.x + (size_t)n0 * xsz);
        const f32x4* gs = reinterpret_cast<const f32x4*>(a.dy + (size_t)n0 * dsz);
        const f32x4* rs = FLY ? reinterpret_cast<const f32x4*>(a.fly.raw + (size_t)n0 * dsz) : nullptr;
        const int nx4 = nu * xsz / 4, nd4 = nu * dsz / 4;
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int ix = tid + i * NT;
            xr[i] = xs[min(ix, nx4 - 1)];
            gr[i] = gs[min(ix, nd4 - 1)];
            if (FLY) rr[i] = rs[min(ix, nd4 - 1)];
        }
    };
    auto store_stage = [&](int st, int buf, const f32x4 (&xr)[XI], const f32x4 (&gr)[XI], const f32x4 (&rr)[XI]) {
        const int n0 = n_begin + st * a.ub;
        const int nu = min(a.ub, n_end - n0);
        const int nx4 = nu * xsz / 4, nd4 = nu * dsz / 4;
        float* xb = lds + buf * bufsz;
        float* db = xb + a.ub * xsz;
        // (a short last stage: the tap offsets of the last row's last positions read up to 16 floats past the staged rows -- times a zero
        //  dy, but the LDS there may never have been written)
        if (nu < a.ub && tid < 4) reinterpret_cast<f32x4*>(xb + nu * xsz)[tid] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int ix = tid + i * NT;
            if (ix < nx4) reinterpret_cast<f32x4*>(xb)[ix] = xr[i];
            if (ix < nd4) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f = 4 * ix + e;
                    const int u = fast_div(f, dsz, inv_dsz);
                    const int rem = f - u * dsz;
                    const int ch = fast_div(rem, a.tpo, inv_tpo);
                    const int tt = rem - ch * a.tpo - kHalo;
                    float d = gr[i][e];
                    if (FLY) {
                        const f32x4 c0 = *reinterpret_cast<const f32x4*>(tab + ch * 8);
                        const float msc = tab[ch * 8 + 4], msh = tab[ch * 8 + 5];
                        const float y = rr[i][e];
                        if (!(fmaf(y, msc, msh) > 0.f)) d = 0.f;
                        d = c0[0] * (d - c0[1] - (y - c0[3]) * c0[2]);
                    }
                    v[e] = (tt >= 0 && tt < a.tout) ? d : 0.f;
                }
                reinterpret_cast<f32x4*>(db)[ix] = v;
            }
        }
    };

    f32x4 acc[CPW][NCO];
#pragma unroll
    for (int i = 0; i < CPW; ++i)
#pragma unroll
        for (int m = 0; m < NCO; ++m) acc[i][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int j = K == 9 ? wave : wave % 3;                     // this wave's tap
    const int cit0 = K == 9 ? 0 : wave / 3;                     // ... and first input-channel tile
    int xrow[CPW], drow[NCO];
#pragma unroll
    for (int i = 0; i < CPW; ++i) xrow[i] = min((cit0 + i) * 16 + r, a.cin - 1) * a.tpi + q * a.stride + j + a.xoff;   // (rows past Cin: clamped, never read back)
#pragma unroll
    for (int m = 0; m < NCO; ++m) drow[m] = min(m * 16 + r, a.cout - 1) * a.tpo + kHalo + q;

    if (nstages > 0) load_stage(0, xr2[0], gr2[0], rr2[0]);
    // (LA2: the 16-channel first conv of TCResNet8-1.0, which runs alone at the step's tail -- 49 frames 739 -> 726 us with two utterances per
    //  stage and this lookahead, 98 frames 1156 -> 1143; the 24-channel instance of TCResNet14-1.5 runs beside the data-gradient chain, and
    //  there the deeper lookahead costs: 98 frames 3611 -> 3656 us per step, 49 frames neutral -- it keeps one stage)
    constexpr bool LA2 = NCO == 1;
    if (LA2 && nstages > 1) load_stage(1, xr2[1], gr2[1], rr2[1]);
    __syncthreads();                                            // (the coefficient table)
    if (nstages > 0) store_stage(0, 0, xr2[0], gr2[0], rr2[0]);
    __syncthreads();
    auto stage = [&](const int st, f32x4 (&xa)[XI], f32x4 (&ga)[XI], f32x4 (&ra)[XI], const f32x4 (&xb_)[XI], const f32x4 (&gb_)[XI], const f32x4 (&rb_)[XI]) {
        // (xa / ga / ra: this stage's set -- already in LDS, free for stage st + 2; xb_ / gb_ / rb_: stage st + 1's, stored behind the MFMAs)
        const int buf = st & 1;
        if (LA2) { if (st + 2 < nstages) load_stage(st + 2, xa, ga, ra); }
        else if (st + 1 < nstages) load_stage(st + 1, const_cast<f32x4 (&)[XI]>(xb_), const_cast<f32x4 (&)[XI]>(gb_), const_cast<f32x4 (&)[XI]>(rb_));
        const int nu = min(a.ub, n_end - (n_begin + st * a.ub));
        const float* xb = lds + buf * bufsz;
        const float* db = xb + a.ub * xsz;
        for (int u = 0; u < nu; ++u) {
            const float* xu = xb + u * xsz;
            const float* du = db + u * dsz;
            for (int t0 = 0; t0 < a.tout; t0 += 4) {
                float af[CPW], bf[NCO];
#pragma unroll
                for (int i = 0; i < CPW; ++i) af[i] = xu[xrow[i] + t0 * a.stride];
#pragma unroll
                for (int m = 0; m < NCO; ++m) bf[m] = du[drow[m] + t0];
#pragma unroll
                for (int i = 0; i < CPW; ++i)
#pragma unroll
                    for (int m = 0; m < NCO; ++m) acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[m], acc[i][m], 0, 0, 0);
            }
        }
        if (st + 1 < nstages) store_stage(st + 1, buf ^ 1, xb_, gb_, rb_);
        __syncthreads();
    };
    if constexpr (LA2) {
        for (int st = 0; st < nstages; st += 2) {
            stage(st, xr2[0], gr2[0], rr2[0], xr2[1], gr2[1], rr2[1]);
            if (st + 1 < nstages) stage(st + 1, xr2[1], gr2[1], rr2[1], xr2[0], gr2[0], rr2[0]);
        }
    } else {
        for (int st = 0; st < nstages; ++st) stage(st, xr2[0], gr2[0], rr2[0], xr2[0], gr2[0], rr2[0]);      // (one set: the round-4 loop)
    }
    float* dst = a.partial + (size_t)blockIdx.x * K * a.cin_pad * a.cout_pad + (size_t)j * a.cin_pad * a.cout_pad;
#pragma unroll
    for (int i = 0; i < CPW; ++i)
#pragma unroll
        for (int m = 0; m < NCO; ++m)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int cig = (cit0 + i) * 16 + q * 4 + reg, col = m * 16 + r;
                if (cig < a.cin_pad && col < a.cout_pad) dst[(size_t)cig * a.cout_pad + col] = acc[i][m][reg];
            }
}

// Shapes with an instance of the LDS-staged kernel: the 3-tap first conv (40 coefficients = three input tiles, one or two output tiles).
// Measured with instances for the 9-tap layers too (one wave per tap), whole training step at batch 4096, 16-byte-load kernel -> LDS-staged
// for the first conv only / for every layer: TCResNet8 909 -> 894 / 899 us, 98 frames 1361 -> 1294 / 1344, TCResNet14-1.5 2753 -> 2712 / 2819,
// 98 frames 4391 -> 4322 / 4491.  On ONE stream the 9-tap instances save 50 (TCResNet8) / 220 us (TCResNet14-1.5) of kernel time, next to
// the data-gradient chain they take it back: they are the heavier neighbours.  Only the first conv's instance is built.
static bool wgrad_lds_instance(int k, int cin, int cout) {
    const int nci = ceil_div(cin, 16), nco = ceil_div(cout, 16);
    return k == 3 && nci == 3 && (nco == 1 || nco == 2);
}
static bool wgrad_lds_shape(int k, int cin, int cout) { return tune_get(TCR_TUNE_WGRAD_LDS) != 1 && wgrad_lds_instance(k, cin, cout); }
// Split-K workgroups of a layer -- decided by the SHAPE alone, so that the slab buffer, the launch and the deferred reduction agree
// whichever kernel runs: the LDS-staged kernel is one workgroup per chunk (no input-tile dimension in its grid), 256 of them (512: +35 us
// per TCResNet8 step, 1024: +110: slab traffic and the reduction).
static int wgrad_nchunk(int k, int cin, int cout, int batch, bool fine) {
    // (round 6, software-pipelined kernel, chunk counts by input-channel tiles: whole rounds of SIMD slots -- 170 chunks for three tiles,
    //  102 for five -- measured +90 us per TCResNet14-1.5 step, half as many chunks +-0: the filter gradients are the main chain's
    //  neighbours, and more of their waves on a CU cost the data-gradient chain more than they gain)
    if (!wgrad_lds_shape(k, cin, cout)) return wgrad_chunks_for(batch, fine);
#ifndef TCR_WGRAD_LDS_CHUNKS
#define TCR_WGRAD_LDS_CHUNKS 256
#endif
    // (TCR_TUNE_WGRAD_LDS = 4, test arm: 16 utterances per workgroup whatever the batch -- the geometry of batch 4096, i.e. eight stages of two
    //  utterances per workgroup and the two-stage lookahead in its steady state, at the batch sizes the emulator tests can afford)
    int n = ceil_div(batch, tune_get(TCR_TUNE_WGRAD_LDS) == 4 ? 16 : 4);
    if (n > TCR_WGRAD_LDS_CHUNKS) n = TCR_WGRAD_LDS_CHUNKS;
    return n < 1 ? 1 : n;
}

// launches the LDS-staged kernel when the shape has an instance and the operands are 16-byte aligned; 1: not covered (nothing launched)
static int launch_wgrad_lds(int k, int stride, int pad_lo, const float* x, const float* dy, float* scratch, int batch, int cin, int cout,
                            int tpi, int tout, int tpo, int nchunk, const WgradFly* fly, hipStream_t s) {
    if (!wgrad_lds_shape(k, cin, cout)) return 1;
    const int nci = ceil_div(cin, 16), nco = ceil_div(cout, 16);
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool f = fly && fly->raw;
    if ((cin * tpi) % 4 || (cout * tpo) % 4 || !al16(x) || !al16(dy) || (f && !al16(fly->raw))) return 1;
    WgradLdsArgs a;
    if (f) a.fly = *fly;
    a.x = x; a.dy = dy; a.partial = scratch; a.batch = batch; a.cin = cin; a.cout = cout; a.cin_pad = nci * 16; a.cout_pad = nco * 16;
    a.tpi = tpi; a.tout = tout; a.tpo = tpo; a.stride = stride; a.xoff = kHalo - pad_lo;
    a.utt_per_block = ceil_div(batch, nchunk);
    // utterances per stage: enough 4-position steps per barrier (>= 8), within two 16-byte loads per thread and tensor
    const int steps = ceil_div(tout, 4);
#ifndef TCR_WGRAD_LDS_STEPS
#define TCR_WGRAD_LDS_STEPS 16     // (round 6: two utterances per stage at 49 frames -- half the barriers, every staging thread loads real data: TCResNet8 step 739 -> 731 us, bitwise)
#endif
    int ub = ceil_div(TCR_WGRAD_LDS_STEPS, steps);
    if (ub > a.utt_per_block) ub = a.utt_per_block;
    while (ub > 1 && (ub * cin * tpi > 8 * 576 || ub * cout * tpo > 8 * 576)) --ub;
    if (ub < 1 || ub * cin * tpi > 8 * 576 || ub * cout * tpo > 8 * 576) return 1;
    a.ub = ub;
    const size_t lds = ((size_t)2 * ub * (cin * tpi + cout * tpo) + (size_t)cout * 8 + 64) * sizeof(float);
    if (lds > 160 * 1024) return 1;
    const dim3 grid(ceil_div(batch, a.utt_per_block));
    void (*kern)(const WgradLdsArgs) = nullptr;
#define TCR_WL(K_, NCI_, NCO_) if (k == K_ && nci == NCI_ && nco == NCO_) kern = f ? conv_wgrad_lds_kernel<K_, NCI_, NCO_, true> : conv_wgrad_lds_kernel<K_, NCI_, NCO_, false>;
    TCR_WL(3, 3, 1) TCR_WL(3, 3, 2)
#undef TCR_WL
    if (!kern) return 1;
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return 1;
    }
    hipLaunchKernelGGL(kern, grid, dim3(576), lds, s, a);
    return check_launch("conv_wgrad_lds_kernel");
}

int wgrad_chunks(int batch) {
    int n = ceil_div(batch, 16);        // >= 16 utterances (4 per wave) per workgroup
    if (n > 128) n = 128;           // (measured at batch 4096: 64 / 256 / 512 split-K workgroups are 3-15 % slower per training step)
    if (n < 1) n = 1;
    return n;
}

// `fine`: twice the split-K workgroups (>= 8 utterances each) -- for the 9-tap layers of one or two input-channel tiles and >= 20 frames
// (TCResNet8's first blocks): 128 x (1..2) workgroups put at most one wave on a SIMD, and these are the filter gradients still running
// when the backward's main chain has ended.  (Four times the workgroups on the old kernel, for the step's last filter gradient only:
// 49 -> 66 us, and its 512-slab reduction 40 us -- the reduction's four lanes per output walk the slabs serially.)
int wgrad_chunks_for(int batch, bool fine) {
    if (!fine) return wgrad_chunks(batch);
    int n = ceil_div(batch, 8);
    if (n > 256) n = 256;
    return n < 1 ? 1 : n;
}

size_t wgrad_partial_floats(int k, int cin, int cout, int batch, bool fine) {
    const int cin_pad = ceil_div(cin, 16) * 16;
    const int cs = cout > 80 ? 80 : cout;
    const int cout_pad = ceil_div(cs, 16) * 16;
    const size_t slab = (size_t)max(wgrad_chunks_for(batch, fine), wgrad_lds_instance(k, cin, cout) ? min(ceil_div(batch, 4), 256) : 0) * k * cin_pad * cout_pad;       // (whatever the knob says later)
    const size_t pw = (k == 1 && cin > 80 && cout > 80) ? (size_t)pw_wgrad_chunks(batch, cin, cout) * cin_pad * (ceil_div(cout, 16) * 16) : 0;   // pw_wgrad_lds_kernel
    return slab > pw ? slab : pw;
}

// waves per workgroup of the on-the-fly 9-tap filter gradient (TCR_TUNE_WGRAD_WAVES: 0 policy, else 4 / 8 / 12 / 16).  (The first conv's
// 3-tap gradient -- the step's last one, alone on the chip -- with 16 waves: 916 vs 906 us per step: two utterances per wave do not pay
// for the sixteen-wave combine.)
static int wgrad4_waves(int nco) {
    const int k = tune_get(TCR_TUNE_WGRAD_WAVES);
    if (k == 4 || k == 8 || k == 12 || k == 16) return nco >= 3 && k > 8 ? 8 : k;      // (three tiles: 12 / 16 waves would spill)
    return 4;
}

template <int K, int S, bool SAFE>
static int launch_wgrad4_k(const WgradArgs& a, int nco, dim3 grid, hipStream_t s) {
    // software-pipelined trips (round 6; TCR_TUNE_WGRAD_PIPE = 1: the one-piece trips): the four-wave instantiations of the 9-tap and
    // 1-tap layers whose K x NCO accumulator tiles leave room for two operand sets (<= 27 tiles: every TC-ResNet launch)
    if constexpr (!SAFE && (K == 9 || K == 1)) {
        const bool four = !(K == 9 && a.fly.raw && wgrad4_waves(nco) != 4);
        if (tune_get(TCR_TUNE_WGRAD_PIPE) != 1 && four && K * nco <= 27) {
#define TCR_W4P(NCO_)                                                                                                           \
            if (nco == NCO_) {                                                                                                  \
                if (a.fly.raw) hipLaunchKernelGGL((conv_wgrad_mfma4_kernel<K, S, NCO_, false, true, 4, true>), grid, dim3(256), 0, s, a);   \
                else hipLaunchKernelGGL((conv_wgrad_mfma4_kernel<K, S, NCO_, false, false, 4, true>), grid, dim3(256), 0, s, a);            \
                return check_launch("conv_wgrad_mfma4_kernel");                                                                 \
            }
            TCR_W4P(1) TCR_W4P(2) TCR_W4P(3)
            if constexpr (K == 1) { TCR_W4P(4) TCR_W4P(5) }
#undef TCR_W4P
        }
    }
#define TCR_W4(NCO_)                                                                                                            \
    if (a.fly.raw) hipLaunchKernelGGL((conv_wgrad_mfma4_kernel<K, S, NCO_, SAFE, true>), grid, dim3(256), 0, s, a);            \
    else hipLaunchKernelGGL((conv_wgrad_mfma4_kernel<K, S, NCO_, SAFE, false>), grid, dim3(256), 0, s, a)
    // more waves per workgroup: the on-the-fly 9-tap layers of up to three output tiles (every conv of TCResNet8's lazy backward)
    if constexpr (K == 9 && !SAFE) {
        const int nwv = a.fly.raw ? wgrad4_waves(nco) : 4;
#define TCR_W4W(NCO_, NWV_) if (nco == NCO_ && nwv == NWV_) { hipLaunchKernelGGL((conv_wgrad_mfma4_kernel<K, S, NCO_, SAFE, true, NWV_>), grid, dim3(NWV_ * 64), 0, s, a); return check_launch("conv_wgrad_mfma4_kernel"); }
        TCR_W4W(1, 8) TCR_W4W(1, 12) TCR_W4W(1, 16) TCR_W4W(2, 8) TCR_W4W(2, 12) TCR_W4W(2, 16) TCR_W4W(3, 8) TCR_W4W(3, 12) TCR_W4W(3, 16)
#undef TCR_W4W
    }
    switch (nco) {
        case 1: TCR_W4(1); break;
        case 2: TCR_W4(2); break;
        case 3: TCR_W4(3); break;
        case 4: TCR_W4(4); break;
        default: TCR_W4(5); break;
    }
#undef TCR_W4
    return check_launch("conv_wgrad_mfma4_kernel");
}

// x_slack: x is followed by readable memory (a workspace tensor).  The first conv (3 x 1, stride 1) reads the caller's feature buffer:
// its instantiation sends the batch's last utterance through element loads clamped to the row (dy is always a workspace tensor).
static bool wgrad4_covers(int k, int stride, bool x_slack) {
    return ((k == 9 || k == 1) ? (x_slack && (stride == 1 || stride == 2)) : (k == 3 && stride == 1)) && tune_get(TCR_TUNE_CONV_B) != 3;
}

template <int K>
static int launch_wgrad_k(const WgradArgs& a0, int nco, dim3 grid, hipStream_t s, bool x_slack = false) {
    if (wgrad4_covers(K, a0.stride, x_slack)) {
        if (K == 3) return x_slack ? launch_wgrad4_k<3, 1, false>(a0, nco, grid, s) : launch_wgrad4_k<3, 1, true>(a0, nco, grid, s);
        constexpr int KK = K == 3 ? 9 : K;
        return a0.stride == 1 ? launch_wgrad4_k<KK, 1, false>(a0, nco, grid, s) : launch_wgrad4_k<KK, 2, false>(a0, nco, grid, s);
    }
    const WgradArgs& a = a0;
    if (a.fly.raw) { set_error("conv wgrad: on-the-fly BN backward needs the 16-byte-load kernel (%dx1, stride %d)", K, a.stride); return TCR_ERR_ARG; }
    switch (nco) {
        case 1: hipLaunchKernelGGL((conv_wgrad_mfma_kernel<K, 1>), grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((conv_wgrad_mfma_kernel<K, 2>), grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((conv_wgrad_mfma_kernel<K, 3>), grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((conv_wgrad_mfma_kernel<K, 4>), grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((conv_wgrad_mfma_kernel<K, 5>), grid, dim3(256), 0, s, a); break;
    }
    return check_launch("conv_wgrad_mfma_kernel");
}

bool conv_wgrad_deferrable(int k, int cin, int cout) { return (k == 9 || k == 3 || k == 1) && cout <= 80 && !(k == 1 && cin > 80 && cout > 80); }

WgradReduceEntry conv_wgrad_entry(int k, int cin, int cout, int batch, const float* scratch, float* dw, bool fine) {
    WgradReduceEntry e;
    e.partial = scratch; e.dw = dw; e.k = k; e.cin = cin; e.cout = cout;
    e.cin_pad = ceil_div(cin, 16) * 16; e.cout_pad = ceil_div(cout, 16) * 16;
    e.nchunk = ceil_div(batch, ceil_div(batch, wgrad_nchunk(k, cin, cout, batch, fine)));
    return e;
}

bool conv_wgrad_fly_covers(int k, int stride, bool x_slack) { return wgrad4_covers(k, stride, x_slack); }

int launch_conv_wgrad_partial(int k, int stride, int pad_lo, const float* x, const float* dy, float* scratch, int batch, int cin, int cout,
                              int tpi, int tout, int tpo, WgradReduceEntry* entry, hipStream_t s, bool fine, bool x_slack, const WgradFly* fly) {
    if (!conv_wgrad_deferrable(k, cin, cout)) { set_error("conv wgrad: shape %dx1 %d->%d cannot defer its reduction", k, cin, cout); return TCR_ERR_ARG; }
    WgradArgs a;
    if (fly) a.fly = *fly;
    a.x = x; a.dy = dy; a.partial = scratch;
    a.batch = batch; a.cin = cin; a.cout = cout; a.cout_all = cout; a.co_base = 0;
    a.cin_pad = ceil_div(cin, 16) * 16;
    a.cout_pad = ceil_div(cout, 16) * 16;
    a.tpi = tpi; a.tout = tout; a.tpo = tpo; a.stride = stride;
    a.xoff = kHalo - pad_lo;
    const int nchunk = wgrad_nchunk(k, cin, cout, batch, fine);
    a.utt_per_block = ceil_div(batch, nchunk);
    {
        const int rc = launch_wgrad_lds(k, stride, pad_lo, x, dy, scratch, batch, cin, cout, tpi, tout, tpo, nchunk, fly, s);
        if (rc != 1) {
            if (rc == TCR_OK && entry) *entry = conv_wgrad_entry(k, cin, cout, batch, scratch, nullptr, fine);
            return rc;
        }
    }
    const dim3 grid(ceil_div(batch, a.utt_per_block), a.cin_pad / 16);
    const int nco = a.cout_pad / 16;
    // 9-tap layers of 4-5 channel tiles (TCResNet14-1.5's 72 channels): 36-45 accumulator tiles leave ONE wave per SIMD, and the side
    // stream that carries these kernels was the critical path of that step.  Launches of two tiles each into the same slab run three
    // waves per SIMD (x is read once per launch: small against the latency hidden).  TCResNet14-1.5 step at batch 4096 by tiles per
    // launch: 5 -> 3255 us, 3 -> 3012, 2 -> 2960, 1 -> 3116; TCResNet8 (<= 3 tiles per layer): 1048 / 1048 / 1054 / 1118 -- so only
    // layers of more than three tiles are split (TCR_TUNE_WGRAD_TILES overrides).
    const int tk = tune_get(TCR_TUNE_WGRAD_TILES);
    // Round 6 (software-pipelined kernel): three-tile layers split 2 + 1 as well -- 27 accumulator tiles + two operand sets are two waves per
    // SIMD, 18 are three: TCResNet8 step 810 -> 781 us (98 frames 1252 -> 1229), TCResNet14-1.5 2484 -> 2483 / 3857 -> 3875.
    const int tiles_per_launch = (k == 9 && wgrad4_covers(k, stride, x_slack)) ? (tk > 0 ? min(tk, nco) : (nco > 2 ? 2 : nco)) : nco;
    for (int t0 = 0; t0 < nco; t0 += tiles_per_launch) {
        WgradArgs b = a;
        const int nt = min(tiles_per_launch, nco - t0);
        b.co_base = t0 * 16; b.pcol = t0 * 16; b.cout = min(cout - t0 * 16, nt * 16);
        int rc;
        if (k == 9) rc = launch_wgrad_k<9>(b, nt, grid, s, x_slack);
        else if (k == 3) rc = launch_wgrad_k<3>(b, nt, grid, s, x_slack);
        else rc = launch_wgrad_k<1>(b, nt, grid, s, x_slack);
        TCR_TRY(rc);
    }
    if (entry) *entry = conv_wgrad_entry(k, cin, cout, batch, scratch, nullptr, fine);
    return TCR_OK;
}

// dw: [K][Cin][Cout]; scratch: wgrad_partial_floats(...) floats.  Output channels are processed in
// slices of at most 80 (5 MFMA column tiles per wave).
int launch_conv_wgrad(int k, int stride, int pad_lo, const float* x, const float* dy, float* dw, float* scratch,
                      int batch, int cin, int cout, int tpi, int tout, int tpo, hipStream_t s, const float* x_scale, const float* x_shift, bool x_slack) {
    if (k != 9 && k != 3 && k != 1) { set_error("conv wgrad: kernel %dx1 has no gfx950 instantiation", k); return TCR_ERR_ARG; }
    if (pw_wgrad_fits(k, stride, cin, cout, tpi, tpo)) return launch_pw_wgrad_lds(x, dy, dw, scratch, batch, cin, cout, tpi, tout, s, x_scale, x_shift);
    if (x_scale) { set_error("conv wgrad: an in-affine operand needs the pointwise LDS kernel"); return TCR_ERR_ARG; }
    for (int co_base = 0; co_base < cout; co_base += 80) {
        WgradArgs a;
        a.x = x; a.dy = dy; a.partial = scratch;
        a.batch = batch;
        a.cin = cin;
        a.cout = (cout - co_base) > 80 ? 80 : (cout - co_base);
        a.cout_all = cout; a.co_base = co_base;
        a.cin_pad = ceil_div(cin, 16) * 16;
        a.cout_pad = ceil_div(a.cout, 16) * 16;
        a.tpi = tpi; a.tout = tout; a.tpo = tpo; a.stride = stride;
        a.xoff = kHalo - pad_lo;
        const int nchunk = wgrad_chunks(batch);
        a.utt_per_block = ceil_div(batch, nchunk);
        const dim3 grid(ceil_div(batch, a.utt_per_block), a.cin_pad / 16);
        const int nco = a.cout_pad / 16;
        int rc;
        if (k == 9) rc = launch_wgrad_k<9>(a, nco, grid, s, x_slack);
        else if (k == 3) rc = launch_wgrad_k<3>(a, nco, grid, s, x_slack);
        else rc = launch_wgrad_k<1>(a, nco, grid, s, x_slack);
        TCR_TRY(rc);
        TCR_TRY(launch_wgrad_reduce(scratch, dw, (int)grid.x, k, cin, a.cout, a.cin_pad, a.cout_pad, cout, co_base, s));
    }
    return TCR_OK;
}

}  // namespace tcr
