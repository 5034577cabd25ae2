// Fused MFCC / log-mel front-end for gfx950: framing + periodic Hann + real FFT + power +
// sparse HTK mel + log + DCT-II in ONE kernel; the waveform is read once from HBM (overlapping
// frames hit L2) and only the [n_coef][T] feature tile is written back.
//
// Replaces the TF ops behind datasets/preprocessors.py:68-94 (stft, |.|^2, mel tensordot, log)
// and :191-193 (mfccs_from_log_mel_spectrograms, [..., :num_mfccs]).
//
// Work decomposition (256-thread workgroup = 64 consecutive frames of the flattened
// [batch x n_frames] frame list, processed in rounds):
//   * FFT: a real FFT of length nfft = 2*NC is a complex FFT of length NC on z[m] = x[2m] + i x[2m+1].
//     The complex FFT is built from 256-point units: 16 lanes x 16 points per lane, two in-register
//     radix-16 passes with ONE 16x16 transpose through LDS in between.  NC == 512 uses two units
//     (even / odd decimation) that are recombined in the post-processing step.
//   * post-processing: bins k and nfft/2-k share one butterfly; the power of both is produced from
//     one (Z[k], Z[NC-k]) pair and written to LDS.
//   * mel: every spectrogram bin lies in one mel-edge segment and feeds <= 2 filters; LPF lanes
//     per frame walk disjoint segment sets (boustrophedon assignment balances the ragged lengths).
//   * DCT-II: after 64 frames of log-mel are in LDS, lane == frame and the DCT coefficients are
//     wave-uniform (scalar loads feeding v_fmac), folded 64 -> 32 terms by the even/odd symmetry.
#include "frontend_plan.h"
#include "frontend_args.h"

namespace tcr {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
// a * conj(b)
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}

// 4-point forward DFT (W4 = -i), in place on (a, b, c, d) -> (y0, y1, y2, y3).
__device__ __forceinline__ void dft4(float2& a, float2& b, float2& c, float2& d) {
    const float2 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
    a = cadd(t0, t2);
    c = csub(t0, t2);
    b = make_float2(t1.x + t3.y, t1.y - t3.x);      // t1 - i t3
    d = make_float2(t1.x - t3.y, t1.y + t3.x);      // t1 + i t3
}

// 16-point forward DFT in registers, natural order in and out (4 x 4 Cooley-Tukey).
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128673848f;   // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f;   // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;   // sqrt(1/2)
    // step A: over n1 for each n0 (x[n0 + 4 n1]) -> T[n0][k1] stored at v[n0 + 4 k1]
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) dft4(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);
    // step B: T[n0][k1] *= W16^(n0 k1)
    v[1 + 4] = cmul(v[1 + 4], make_float2(C1, -S1));      // W^1
    v[1 + 8] = cmul(v[1 + 8], make_float2(R2, -R2));      // W^2
    v[1 + 12] = cmul(v[1 + 12], make_float2(S1, -C1));    // W^3
    v[2 + 4] = cmul(v[2 + 4], make_float2(R2, -R2));      // W^2
    v[2 + 8] = make_float2(v[2 + 8].y, -v[2 + 8].x);      // W^4 = -i
    v[2 + 12] = cmul(v[2 + 12], make_float2(-R2, -R2));   // W^6
    v[3 + 4] = cmul(v[3 + 4], make_float2(S1, -C1));      // W^3
    v[3 + 8] = cmul(v[3 + 8], make_float2(-R2, -R2));     // W^6
    v[3 + 12] = cmul(v[3 + 12], make_float2(-C1, S1));    // W^9
    // step C: over n0 for each k1 -> X[k1 + 4 k0] lands at v[4 k1 + k0]; un-shuffle below
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
    // v[4 k1 + k0] holds X[k1 + 4 k0]: transpose the 4x4 index grid to natural order
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i + 1; j < 4; ++j) {
            const float2 t = v[4 * i + j];
            v[4 * i + j] = v[4 * j + i];
            v[4 * j + i] = t;
        }
}

// One (Z[k], Z[N-k]) pair of the real-FFT post-processing: powers of X[k] and X[N-k].
__device__ __forceinline__ void real_pair_power(float2 zk, float2 zn, float2 w, float& p_lo, float& p_hi) {
    const float ax = 0.5f * (zk.x + zn.x), ay = 0.5f * (zk.y - zn.y);       // A = (Z[k] + conj Z[N-k]) / 2
    const float bx = 0.5f * (zk.y + zn.y), by = -0.5f * (zk.x - zn.x);      // B = -i (Z[k] - conj Z[N-k]) / 2
    const float2 c = cmul(make_float2(bx, by), w);                          // W^k B
    const float xr = ax + c.x, xi = ay + c.y, yr = ax - c.x, yi = ay - c.y;
    p_lo = fmaf(xr, xr, xi * xi);       // |X[k]|^2
    p_hi = fmaf(yr, yr, yi * yi);       // |X[N-k]|^2
}

// VAR bit 0: the phases of a round are ordered by wave-local points (a frame's lanes never straddle a
//            wavefront) instead of s_barrier;  bit 1: the next round's samples are prefetched into registers
//            while the current round computes (the kernel is latency-bound at its 2 waves/SIMD).
template <int NC, int VAR>
__global__ __launch_bounds__(256, 2) void frontend_kernel(const FrontendArgs a) {
    constexpr bool WSYNC = (VAR & 1) != 0;
    constexpr bool PREFETCH = (VAR & 2) != 0;
    constexpr int LPF = NC / 16;            // lanes per frame
    constexpr int FPR = 256 / LPF;          // frames per round
    constexpr int ROUNDS = 64 / FPR;
    constexpr int SUB = NC / 256;           // 256-point units per frame
    constexpr int NBINS = NC + 1;
    constexpr int NMEL = 64, NSEG = NMEL + 1;
    constexpr int XLD = 17;                 // padded row of the 16x16 transpose tile (float2)
    constexpr int UNIT = 16 * XLD;          // float2 per unit (>= 256 for the E/O image)
    constexpr int PLD = NBINS + 3;

    __shared__ float2 s_x[16 * UNIT];               // transpose tiles, then the FFT output of each unit
    __shared__ float s_p[FPR * PLD];                // power (or magnitude) spectrum
    __shared__ float s_ud[FPR * 2 * (NSEG + 1)];    // per-segment up / down partial sums
    __shared__ float s_lm[NMEL * 65];               // log-mel [mel][frame], 64 frames
    __shared__ float2 s_wud[NBINS];                 // mel slopes per bin (LDS copy: the mel loop is latency-bound on them)
    __shared__ int s_seg[NSEG + 1];

    const int tid = threadIdx.x;
    const int f = tid / LPF;                // frame slot in the round
    const int lf = tid % LPF;               // lane within the frame
    const int u = lf >> 4;                  // unit within the frame (0: even, 1: odd decimation)
    const int l = tid & 15;                 // lane within the unit
    const int unit = tid >> 4;

    // Round-invariant per-lane constants: window taps and twiddles of this lane's 16 points.
    float2 wnd[16], tw[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int idx = 2 * (SUB * (l + 16 * q) + u);
        wnd[q] = (idx < a.win) ? make_float2(a.window[idx], a.window[idx + 1]) : make_float2(0.f, 0.f);
        tw[q] = a.tw256[l * 16 + q];
    }
    float2 twr[8], twc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = lf + LPF * i;
        twr[i] = a.tw_real[k];
        twc[i] = (SUB == 2) ? a.tw_combine[k] : make_float2(1.f, 0.f);
    }

#define TCR_SYNC() do { if (WSYNC) wave_sync(); else __syncthreads(); } while (0)
    for (int i = threadIdx.x; i < NBINS; i += 256) s_wud[i] = a.wud[i];
    for (int i = threadIdx.x; i <= NSEG; i += 256) s_seg[i] = a.seg_start[i];
    const float2 twmid = a.tw_real[NC / 2];
    __syncthreads();
    float2 xa[16];
    auto load_frame = [&](int rr, float2 (&dst)[16]) {
        int g = blockIdx.x * 64 + rr * FPR + f;
        g = min(g, a.total_frames - 1);
        const int n = g / a.n_frames;
        const int t = g - n * a.n_frames;
        const float* src = a.wav + (size_t)n * a.n_samples + (size_t)t * a.hop;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int idx = 2 * (SUB * (l + 16 * q) + u);
            float2 x = make_float2(0.f, 0.f);
            if (idx < a.win) {
                if (a.aligned) x = *reinterpret_cast<const float2*>(src + idx);
                else x = make_float2(src[idx], src[idx + 1]);
            }
            dst[q] = x;
        }
    };
    for (int r = 0; r < ROUNDS; ++r) {
        // ---------------- load (+ prefetch of the next round) + window + first radix-16 pass ----------------
        float2 v[16];
        if (!PREFETCH || r == 0) load_frame(r, xa);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = make_float2(xa[q].x * wnd[q].x, xa[q].y * wnd[q].y);
        if (PREFETCH && r + 1 < ROUNDS) load_frame(r + 1, xa);
        dft16(v);
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) s_x[unit * UNIT + k2 * XLD + l] = cmul(v[k2], tw[k2]);
        TCR_SYNC();
        // ---------------- transpose + second radix-16 pass ----------------
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = s_x[unit * UNIT + l * XLD + n1];
        dft16(v);
        TCR_SYNC();
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) s_x[unit * UNIT + 16 * k1 + l] = v[k1];     // bin 16 k1 + l
        TCR_SYNC();
        // ---------------- real-FFT post-processing -> power spectrum ----------------
        {
            const float2* E = s_x + (f * SUB) * UNIT;
            const float2* O = s_x + (f * SUB + SUB - 1) * UNIT;
            float* P = s_p + f * PLD;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = lf + LPF * i;                 // 0 .. NC/2-1
                const int kn = (SUB == 2) ? ((256 - k) & 255) : ((NC - k) & (NC - 1));
                float2 zk, zn;
                if (SUB == 2) {
                    zk = cadd(E[k], cmul(O[k], twc[i]));        // Z[k]      = E[k] + W512^k O[k]
                    zn = cadd(E[kn], cmulc(O[kn], twc[i]));     // Z[512-k]  = E[256-k] + conj(W512^k) O[256-k]
                } else {
                    zk = E[k];
                    zn = E[kn];
                }
                float plo, phi;
                real_pair_power(zk, zn, twr[i], plo, phi);
                if (a.magnitude) { plo = sqrtf(plo); phi = sqrtf(phi); }
                P[k] = plo;
                P[NC - k] = phi;
            }
            if (lf == 0) {                                  // the self-paired middle bin k = NC/2
                float2 z;
                if (SUB == 2) z = csub(E[0], O[0]);         // Z[256] = E[0] - O[0]
                else z = E[NC / 2];
                float plo, phi;
                real_pair_power(z, z, twmid, plo, phi);
                if (a.magnitude) plo = sqrtf(plo);
                P[NC / 2] = plo;
            }
        }
        TCR_SYNC();
        // ---------------- sparse mel: per-segment up/down sums ----------------
        {
            const float* P = s_p + f * PLD;
            float* UD = s_ud + f * 2 * (NSEG + 1);
            for (int i = 0;; ++i) {
                const int j = (i & 1) ? (i + 1) * LPF - 1 - lf : i * LPF + lf;
                if (i * LPF >= NSEG) break;
                if (j < NSEG) {
                    const int k0 = s_seg[j], k1 = s_seg[j + 1];
                    float up = 0.f, dn = 0.f;
                    int k = k0;
                    for (; k + 4 <= k1; k += 4) {           // 4 bins per trip: the 8 LDS reads are independent
                        const float p0 = P[k], p1 = P[k + 1], p2 = P[k + 2], p3 = P[k + 3];
                        const float2 w0 = s_wud[k], w1 = s_wud[k + 1], w2 = s_wud[k + 2], w3 = s_wud[k + 3];
                        up = fmaf(w0.x, p0, up); dn = fmaf(w0.y, p0, dn);
                        up = fmaf(w1.x, p1, up); dn = fmaf(w1.y, p1, dn);
                        up = fmaf(w2.x, p2, up); dn = fmaf(w2.y, p2, dn);
                        up = fmaf(w3.x, p3, up); dn = fmaf(w3.y, p3, dn);
                    }
                    for (; k < k1; ++k) {
                        const float p = P[k];
                        const float2 w = s_wud[k];
                        up = fmaf(w.x, p, up);
                        dn = fmaf(w.y, p, dn);
                    }
                    UD[j] = up;
                    UD[NSEG + 1 + j] = dn;
                }
            }
        }
        TCR_SYNC();
        // ---------------- log(mel + 1e-6) -> [mel][frame] ----------------
        {
            const float* UD = s_ud + f * 2 * (NSEG + 1);
#pragma unroll
            for (int i = 0; i < NMEL / LPF; ++i) {
                const int m = lf + LPF * i;
                const float mel = UD[m] + UD[NSEG + 1 + m + 1];     // up-slope of segment m + down-slope of segment m+1
                s_lm[m * 65 + r * FPR + f] = a.log_floor ? logf(fmaxf(mel, 1e-12f)) : logf(mel + 1e-6f);
            }
        }
    }
    __syncthreads();

#undef TCR_SYNC
    // ---------------- DCT-II (lane == frame, wave-uniform coefficients) + store ----------------
    const int fr = tid & 63;
    const int w = tid >> 6;
    const int g = blockIdx.x * 64 + fr;
    const bool valid = g < a.total_frames;
    const int gg = valid ? g : a.total_frames - 1;
    const int n = gg / a.n_frames;
    const int t = gg - n * a.n_frames;
    float* dst = a.out + (size_t)n * a.n_coef * a.tp + kHalo + t;
    if (a.no_dct) {
        for (int m = w; m < a.n_coef; m += 4) {
            if (valid) {
                float* row = dst + (size_t)m * a.tp;
                row[0] = s_lm[m * 65 + fr];
                if (t == 0) { row[-4] = 0.f; row[-3] = 0.f; row[-2] = 0.f; row[-1] = 0.f; }
                if (t == a.n_frames - 1) { row[1] = 0.f; row[2] = 0.f; row[3] = 0.f; row[4] = 0.f; }
            }
        }
        return;
    }
    float h[NMEL / 2];
    const float sgn = (w & 1) ? -1.f : 1.f;     // odd coefficients use l[n] - l[N-1-n]
#pragma unroll
    for (int i = 0; i < NMEL / 2; ++i) h[i] = fmaf(sgn, s_lm[(NMEL - 1 - i) * 65 + fr], s_lm[i * 65 + fr]);
    for (int c = w; c < a.n_coef; c += 4) {
        const float* d = a.dcth + c * (NMEL / 2);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NMEL / 2; ++i) acc = fmaf(d[i], h[i], acc);
        if (valid) {
            float* row = dst + (size_t)c * a.tp;
            row[0] = acc;
            if (t == 0) { row[-4] = 0.f; row[-3] = 0.f; row[-2] = 0.f; row[-1] = 0.f; }
            if (t == a.n_frames - 1) { row[1] = 0.f; row[2] = 0.f; row[3] = 0.f; row[4] = 0.f; }
        }
    }
}

// [B][T][F] <-> [B][F][Tp] re-layouts for the "no_preprocessing" path.
// ---------------------------------------------------------------------------------------------
// Deploy-path MFCC in float64 (method 2: contrib_audio.audio_spectrogram + contrib_audio.mfcc, datasets/preprocessors.py:98-124,
// 196-203).  TF's two C++ ops compute in double, and the path has NO log offset (log(max(x, 1e-12))): on clean tones the empty
// bands hold nothing but the transform's round-off, which a float32 FFT puts five orders of magnitude above a float64 one -- the
// float32 kernels were up to 0.5 off on the pure-tone fixture rows.  The reference runs this path one utterance at a time (freeze.py,
// on-device comparison), so rate does not matter: one workgroup per frame, iterative radix-2 FFT on doubles in LDS with twiddles from
// cos / sin in double, magnitude, the op's filterbank (the plan's slopes), log, DCT-II in double; float32 only on the way out.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frontend_deploy_f64_kernel(const FrontendArgs a, int nfft, int log_nfft) {
    constexpr double kPi = 3.14159265358979323846;
    __shared__ double s_re[1024], s_im[1024];
    __shared__ double s_mag[513];
    __shared__ double s_lm[64];
    // Tables filled once per (persistent) workgroup -- round 6, advisor: as cos / sin calls inside the frame loop they were 1024 window
    // + 5120 butterfly + 64 n_coef evaluations in double PER FRAME, which made a large-batch call through this method orders of
    // magnitude slower than it has to be.  Window and twiddles are bitwise the per-butterfly expressions (-pi pos / half and
    // -pi j / (nfft / 2) with j = pos (nfft / 2) / half differ by exact power-of-two scalings of numerator and denominator).
    __shared__ double s_win[1024];
    __shared__ double s_twr[512], s_twi[512];
    __shared__ double s_dct[256];           // cos(pi j / 128): the DCT-II's cos(pi c (2 m + 1) / 128) at j = c (2 m + 1) mod 256
    const int tid = threadIdx.x;
    const float2* wud = reinterpret_cast<const float2*>(a.wud);
    for (int i = tid; i < nfft; i += 256) s_win[i] = i < a.win ? 0.5 - 0.5 * cos(2.0 * kPi * (double)i / (double)a.win) : 0.0;
    for (int j = tid; j < nfft / 2; j += 256) {
        const double ang = -kPi * (double)j / (double)(nfft / 2);
        s_twr[j] = cos(ang);
        s_twi[j] = sin(ang);
    }
    s_dct[tid] = cos(kPi * (double)tid / 128.0);
    __syncthreads();
    for (int g = blockIdx.x; g < a.total_frames; g += gridDim.x) {
        const int n = g / a.n_frames, t = g - n * a.n_frames;
        const float* src = a.wav + (size_t)n * a.n_samples + (size_t)t * a.hop;
        for (int i = tid; i < nfft; i += 256) {
            const double x = i < a.win ? (double)src[i] * s_win[i] : 0.0;
            const int j = (int)(__brev((unsigned)i) >> (32 - log_nfft));
            s_re[j] = x;
            s_im[j] = 0.0;
        }
        __syncthreads();
        int tstep = nfft / 2;               // table stride of the stage: (nfft / 2) / half
        for (int half = 1; half < nfft; half <<= 1, tstep >>= 1) {
            for (int b = tid; b < nfft / 2; b += 256) {
                const int pos = b & (half - 1), i0 = ((b - pos) << 1) + pos, i1 = i0 + half;
                const double wr = s_twr[pos * tstep], wi = s_twi[pos * tstep];
                const double xr = s_re[i1], xi = s_im[i1];
                const double tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
                const double ur = s_re[i0], ui = s_im[i0];
                s_re[i1] = ur - tr; s_im[i1] = ui - ti;
                s_re[i0] = ur + tr; s_im[i0] = ui + ti;
            }
            __syncthreads();
        }
        for (int k = tid; k <= nfft / 2; k += 256) s_mag[k] = sqrt(s_re[k] * s_re[k] + s_im[k] * s_im[k]);
        __syncthreads();
        if (tid < 64) {            // filter m: the up-slopes of segment m, then the down-slopes of segment m + 1 (ascending bins, as the op adds them)
            double sum = 0.0;
            for (int k = a.seg_start[tid]; k < a.seg_start[tid + 1]; ++k) sum += (double)wud[k].x * s_mag[k];
            for (int k = a.seg_start[tid + 1]; k < a.seg_start[tid + 2]; ++k) sum += (double)wud[k].y * s_mag[k];
            s_lm[tid] = log(sum > 1e-12 ? sum : 1e-12);
        }
        __syncthreads();
        if (tid < a.n_coef) {
            double v = 0.0;
            for (int m = 0; m < 64; ++m) v += s_lm[m] * s_dct[(tid * (2 * m + 1)) & 255];
            float* row = a.out + ((size_t)n * a.n_coef + tid) * a.tp + kHalo + t;
            row[0] = (float)(v * sqrt(2.0 / 64.0));
            if (t == 0) { row[-4] = 0.f; row[-3] = 0.f; row[-2] = 0.f; row[-1] = 0.f; }
            if (t == a.n_frames - 1) { row[1] = 0.f; row[2] = 0.f; row[3] = 0.f; row[4] = 0.f; }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void to_planar_kernel(const float* __restrict__ ntf, float* __restrict__ planar,
                                                        int batch, int t_len, int f_len) {
    const int tp = t_len + 2 * kHalo;
    const int64_t total = (int64_t)batch * f_len * tp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int tt = (int)(i % tp) - kHalo;
        const int64_t r = i / tp;
        const int c = (int)(r % f_len);
        const int64_t b = r / f_len;
        planar[i] = (tt >= 0 && tt < t_len) ? ntf[(b * t_len + tt) * f_len + c] : 0.f;
    }
}

__global__ __launch_bounds__(256) void from_planar_kernel(const float* __restrict__ planar, float* __restrict__ ntf,
                                                          int batch, int t_len, int f_len) {
    const int tp = t_len + 2 * kHalo;
    const int64_t total = (int64_t)batch * t_len * f_len;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % f_len);
        const int64_t r = i / f_len;
        const int tt = (int)(r % t_len);
        const int64_t b = r / t_len;
        ntf[i] = planar[(b * f_len + c) * tp + kHalo + tt];
    }
}

}  // namespace tcr

using namespace tcr;

extern "C" int tcr_frontend_fwd(const tcr_frontend_cfg* cfg, const void* plan_dev, const float* wav, int batch,
                                float* feat, void* stream) {
    return tcr_frontend_fwd_rounds(cfg, plan_dev, wav, batch, feat, 0, stream);
}

extern "C" int tcr_frontend_fwd_rounds(const tcr_frontend_cfg* cfg, const void* plan_dev, const float* wav, int batch,
                                       float* feat, int rounds, void* stream) {
    TCR_REQUIRE(cfg && plan_dev && wav && feat, "tcr_frontend_fwd: null argument");
    TCR_REQUIRE(batch > 0, "tcr_frontend_fwd: batch must be positive (got %d)", batch);
    TCR_REQUIRE(cfg->nfft == 512 || cfg->nfft == 1024, "tcr_frontend_fwd: unresolved or unsupported configuration (nfft=%d)", cfg->nfft);
    TCR_REQUIRE((int64_t)batch * cfg->n_frames < (int64_t)1 << 31, "tcr_frontend_fwd: batch too large");
    const FrontendPlanLayout L = frontend_plan_layout(*cfg);
    const float* p = static_cast<const float*>(plan_dev);
    FrontendArgs a;
    a.wav = wav;
    a.out = feat;
    a.window = p + L.window;
    a.window_sgn = p + L.window_sgn;
    a.tw256 = reinterpret_cast<const float2*>(p + L.tw256);
    a.tw_combine = reinterpret_cast<const float2*>(p + L.tw_combine);
    a.tw_real = reinterpret_cast<const float2*>(p + L.tw_real);
    a.seg_start = reinterpret_cast<const int*>(p + L.seg_start);
    a.wud = reinterpret_cast<const float2*>(p + L.wud);
    a.dcth = p + L.dcth;
    a.mel_items = reinterpret_cast<const int*>(p + L.mel_items);
    a.mel_ifirst = reinterpret_cast<const int*>(p + L.mel_ifirst);
    a.mel_wit = reinterpret_cast<const float2*>(p + L.mel_wit);
    a.dct_tab = p + L.dct_tab;
    a.n_samples = cfg->n_samples;
    a.win = cfg->win;
    a.hop = cfg->hop;
    a.n_frames = cfg->n_frames;
    a.n_coef = cfg->n_coef;
    a.tp = tcr_padded_len(cfg->n_frames);
    a.total_frames = batch * cfg->n_frames;
    a.magnitude = cfg->method != 0;
    a.no_dct = cfg->method == 1;
    a.log_floor = cfg->method == 2;
    a.rounds = rounds > 0 ? rounds : 0;        // (> 0: the caller's choice; the packed launchers clamp it, 0: their cost model)
    a.stagger = 0;
    a.stagger_div = 1;
    a.aligned = ((cfg->n_samples | cfg->hop) & 1) == 0 && (reinterpret_cast<uintptr_t>(wav) & 7) == 0;
    const int grid = ceil_div(a.total_frames, 64);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int knob = tune_get(TCR_TUNE_FRONTEND);
    if (cfg->method == 2 && tune_get(TCR_TUNE_DEPLOY_F32) == 0) {      // deploy path: float64 (the ops it restates compute in double)
        TCR_REQUIRE(cfg->n_mel == 64 && cfg->nfft <= 1024 && cfg->n_coef <= 64,
                    "tcr_frontend_fwd: the float64 deploy kernel is built for 64 mel bands, <= 64 coefficients and nfft <= 1024 (got %d / %d / %d)", cfg->n_mel, cfg->n_coef, cfg->nfft);
        int lg = 0;
        while ((1 << lg) < cfg->nfft) ++lg;
        hipLaunchKernelGGL(frontend_deploy_f64_kernel, dim3(min(a.total_frames, 16 * device_cus())), dim3(256), 0, s, a, cfg->nfft, lg);
        return check_launch("frontend_deploy_f64_kernel");
    }
    if (knob == 0 || knob == 5 || knob >= 10) {           // default: packed-FP32 kernels where their window specialisation applies
        if (tune_get(TCR_TUNE_FE_KERNEL) == 0) {          // three waves per SIMD (frontend_pk3.hip) for filterbanks its unrolled trips cover
            const int rc = launch_frontend_pk3(cfg->nfft / 2, a, frontend_mel_item_count(*cfg), s);
            if (rc != 1) return rc;
        }
        const int rc = launch_frontend_pk(cfg->nfft / 2, a, s);     // two waves per SIMD (frontend_pk.hip)
        if (rc != 1) return rc;
    }
    const int var = knob == 0 || knob == 5 || knob >= 10 ? 3 : (knob - 1) & 3;
#define TCR_FE(NC_)                                                                                     \
    switch (var) {                                                                                      \
        case 0: hipLaunchKernelGGL((frontend_kernel<NC_, 0>), dim3(grid), dim3(256), 0, s, a); break;   \
        case 1: hipLaunchKernelGGL((frontend_kernel<NC_, 1>), dim3(grid), dim3(256), 0, s, a); break;   \
        case 2: hipLaunchKernelGGL((frontend_kernel<NC_, 2>), dim3(grid), dim3(256), 0, s, a); break;   \
        default: hipLaunchKernelGGL((frontend_kernel<NC_, 3>), dim3(grid), dim3(256), 0, s, a); break;  \
    }
    if (cfg->nfft == 512) { TCR_FE(256); } else { TCR_FE(512); }
#undef TCR_FE
    return check_launch("frontend_kernel");
}

extern "C" int tcr_features_to_planar(const float* ntf, int batch, int t, int f, float* planar, void* stream) {
    TCR_REQUIRE(ntf && planar && batch > 0 && t > 0 && f > 0, "tcr_features_to_planar: bad argument");
    const int64_t total = (int64_t)batch * f * tcr_padded_len(t);
    const int grid = (int)(ceil_div64(total, 256) < 4096 ? ceil_div64(total, 256) : 4096);
    hipLaunchKernelGGL(to_planar_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), ntf, planar, batch, t, f);
    return check_launch("to_planar_kernel");
}

extern "C" int tcr_features_from_planar(const float* planar, int batch, int t, int f, float* ntf, void* stream) {
    TCR_REQUIRE(ntf && planar && batch > 0 && t > 0 && f > 0, "tcr_features_from_planar: bad argument");
    const int64_t total = (int64_t)batch * f * t;
    const int grid = (int)(ceil_div64(total, 256) < 4096 ? ceil_div64(total, 256) : 4096);
    hipLaunchKernelGGL(from_planar_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), planar, ntf, batch, t, f);
    return check_launch("from_planar_kernel");
}
