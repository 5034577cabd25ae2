// Packed-FP32 build of the fused MFCC / log-mel front-end (same decomposition as frontend.hip; see there).
//
// Why: at 2 waves / SIMD the scalar-FP32 kernel keeps the VALU ~65 % busy but retires < 1 flop per lane-instruction
// -- the FFT butterflies and the real-FFT post-processing are complex adds / multiplies issued one component at a
// time (profiles/r01_final_pmc.csv: 125 M VALU instructions for 7 GFLOP).  gfx950 has two-wide FP32 VOP3P forms
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) whose op_sel / neg modifiers express exactly the complex idioms:
//   a +- b            1 instruction            a -+ i b        1 (swap + negate one half of b)
//   a * b             2 (pk_mul + pk_fma)      a * conj(b)     2
//   a +- conj(b)      1
// The 0.5 factors of the real-FFT split are not applied: the spectrum is 4x (power) / 2x (magnitude) too large and
// the mel slopes are pre-scaled by the inverse power of two when they are staged in LDS (exact).
#include "frontend_plan.h"
#include "frontend_args.h"

namespace tcr {

typedef float v2 __attribute__((ext_vector_type(2)));

#if defined(TCR_HOST_EMULATION)
#define TCR_PK_ASM 0
#else
#define TCR_PK_ASM 1
#endif

// a - i b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ v2 c_submi(v2 a, v2 b) {
#if TCR_PK_ASM
    v2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
#else
    return (v2){a.x + b.y, a.y - b.x};
#endif
}
// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ v2 c_addmi(v2 a, v2 b) {
#if TCR_PK_ASM
    v2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
#else
    return (v2){a.x - b.y, a.y + b.x};
#endif
}
// a + conj(b), a - conj(b)
__device__ __forceinline__ v2 c_addc(v2 a, v2 b) {
#if TCR_PK_ASM
    v2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
#else
    return (v2){a.x + b.x, a.y - b.y};
#endif
}
__device__ __forceinline__ v2 c_subc(v2 a, v2 b) {
#if TCR_PK_ASM
    v2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
#else
    return (v2){a.x - b.x, a.y + b.y};
#endif
}
// a * b:  t = a.y * (b.y, b.x);  r = (fma(a.x, b.x, -t.x), fma(a.x, b.y, t.y))
__device__ __forceinline__ v2 c_mul(v2 a, v2 b) {
#if TCR_PK_ASM
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
#else
    const float tx = a.y * b.y, ty = a.y * b.x;
    return (v2){fmaf(a.x, b.x, -tx), fmaf(a.x, b.y, ty)};
#endif
}
// a * b with a wave-uniform constant b (scalar register pair)
__device__ __forceinline__ v2 c_mulk(v2 a, v2 b) {
#if TCR_PK_ASM
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "s"(b), "v"(t));
    return r;
#else
    return c_mul(a, b);
#endif
}
// a * conj(b):  r = (fma(a.x, b.x, t.x), fma(-a.x, b.y, t.y))
__device__ __forceinline__ v2 c_mulc(v2 a, v2 b) {
#if TCR_PK_ASM
    v2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
#else
    const float tx = a.y * b.y, ty = a.y * b.x;
    return (v2){fmaf(a.x, b.x, tx), fmaf(-a.x, b.y, ty)};
#endif
}

// 4-point forward DFT (W4 = -i), in place: 8 packed instructions.
__device__ __forceinline__ void pk_dft4(v2& a, v2& b, v2& c, v2& d) {
    const v2 t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
    a = t0 + t2;
    c = t0 - t2;
    b = c_submi(t1, t3);
    d = c_addmi(t1, t3);
}

// 16-point forward DFT in registers, natural order in and out (4 x 4 Cooley-Tukey).
__device__ __forceinline__ void pk_dft16(v2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128673848f;   // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f;   // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;   // sqrt(1/2)
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) pk_dft4(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);
    v[1 + 4] = c_mulk(v[1 + 4], (v2){C1, -S1});         // W^1
    v[1 + 8] = c_mulk(v[1 + 8], (v2){R2, -R2});         // W^2
    v[1 + 12] = c_mulk(v[1 + 12], (v2){S1, -C1});       // W^3
    v[2 + 4] = c_mulk(v[2 + 4], (v2){R2, -R2});         // W^2
    v[2 + 8] = c_submi((v2){0.f, 0.f}, v[2 + 8]);       // W^4 = -i
    v[2 + 12] = c_mulk(v[2 + 12], (v2){-R2, -R2});      // W^6
    v[3 + 4] = c_mulk(v[3 + 4], (v2){S1, -C1});         // W^3
    v[3 + 8] = c_mulk(v[3 + 8], (v2){-R2, -R2});        // W^6
    v[3 + 12] = c_mulk(v[3 + 12], (v2){-C1, S1});       // W^9
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) pk_dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i + 1; j < 4; ++j) {
            const v2 t = v[4 * i + j];
            v[4 * i + j] = v[4 * j + i];
            v[4 * j + i] = t;
        }
}

// One (Z[k], Z[N-k]) pair of the real-FFT split, WITHOUT the 1/2 factors: returns 4 |X[k]|^2 and 4 |X[N-k]|^2.
// wmi = -i W^k.   2X[k] = A + C, 2X[N-k] = conj(A - C) with A = Z[k] + conj Z[N-k], C = W^k (-i)(Z[k] - conj Z[N-k]).
__device__ __forceinline__ void pk_real_pair_power(v2 zk, v2 zn, v2 wmi, float& p_lo, float& p_hi) {
    const v2 A = c_addc(zk, zn), D = c_subc(zk, zn);
    const v2 C = c_mul(D, wmi);
    const v2 X = A + C, Y = A - C;
    p_lo = fmaf(X.x, X.x, X.y * X.y);
    p_hi = fmaf(Y.x, Y.x, Y.y * Y.y);
}

// QV: number of leading radix-16 inputs per lane that fall inside the analysis window -- identical for every lane when
// win / 2 is a multiple of 16 * SUB (640 -> 10 of 16, 480 -> 15 of 16).  The rest of the zero-padded FFT input is never
// loaded, windowed or butterflied (the compiler folds the zeros through the first radix-16 pass), and no load is masked.
template <int NC, int QV, bool MAG>
__global__ __launch_bounds__(256, 2) void frontend_pk_kernel(const FrontendArgs a) {
    constexpr int LPF = NC / 16;            // lanes per frame
    constexpr int FPR = 256 / LPF;          // frames per round
    constexpr int ROUNDS = 64 / FPR;
    constexpr int SUB = NC / 256;           // 256-point units per frame
    constexpr int NBINS = NC + 1;
    constexpr int NMEL = 64, NSEG = NMEL + 1;
    constexpr int XLD = 17;                 // padded row of the 16x16 transpose tile
    constexpr int UNIT = 16 * XLD;
    constexpr int PLD = NBINS + 31;         // row stride = 32 (mod 64) banks: the two frames of a wave never collide

    __shared__ v2 s_x[16 * UNIT];                   // transpose tiles, then the FFT output of each unit
    __shared__ float s_p[FPR * PLD];                // 4 x power (or 2 x magnitude) spectrum
    constexpr int ULD = NSEG + 1;                   // (a bank-friendlier 80 pushes NC = 256 past 80 KB of LDS: 1 workgroup / CU)
    __shared__ v2 s_ud[FPR * ULD];                  // per-segment (up, down) partial sums
    __shared__ float s_lm[NMEL * 65];               // log-mel [mel][frame], 64 frames
    __shared__ v2 s_wud[NBINS];                     // mel slopes per bin, pre-scaled by 1/4 (1/2)
    __shared__ int s_seg[NSEG + 1];

    const int tid = threadIdx.x;
    const int f = tid / LPF;                // frame slot in the round
    const int lf = tid % LPF;               // lane within the frame
    const int u = lf >> 4;                  // unit within the frame (0: even, 1: odd decimation)
    const int l = tid & 15;                 // lane within the unit
    const int unit = tid >> 4;
    const v2* tw256 = reinterpret_cast<const v2*>(a.tw256);
    const v2* tw_real = reinterpret_cast<const v2*>(a.tw_real);
    const v2* tw_combine = reinterpret_cast<const v2*>(a.tw_combine);

    v2 wnd[QV], tw[16];
#pragma unroll
    for (int q = 0; q < QV; ++q) {
        const int idx = 2 * (SUB * (l + 16 * q) + u);
        wnd[q] = (v2){a.window[idx], a.window[idx + 1]};
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) tw[q] = tw256[l * 16 + q];
    v2 twr[8], twc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = lf + LPF * i;
        const v2 w = tw_real[k];
        twr[i] = (v2){w.y, -w.x};           // -i W^k
        twc[i] = (SUB == 2) ? tw_combine[k] : (v2){1.f, 0.f};
    }
    {
        const float fold = MAG ? 0.5f : 0.25f;
        const v2* wud = reinterpret_cast<const v2*>(a.wud);
        for (int i = threadIdx.x; i < NBINS; i += 256) s_wud[i] = wud[i] * fold;
        for (int i = threadIdx.x; i <= NSEG; i += 256) s_seg[i] = a.seg_start[i];
    }
    const v2 wm = tw_real[NC / 2];
    const v2 twmid = (v2){wm.y, -wm.x};
    __syncthreads();

    const float inv_frames = 1.0f / (float)a.n_frames;
    const int rounds = a.rounds;            // <= ROUNDS; chosen by the launcher so that the grid fills whole dispatch waves
    const int fpw = rounds * FPR;           // frames per workgroup
    v2 xa[QV];
    auto load_frame = [&](int rr, v2 (&dst)[QV]) {
        int g = blockIdx.x * fpw + rr * FPR + f;
        g = min(g, a.total_frames - 1);
        int n = (int)(((float)g + 0.5f) * inv_frames);          // g / n_frames: float multiply + one-step fix-up
        n += (n + 1) * a.n_frames <= g ? 1 : (n * a.n_frames > g ? -1 : 0);
        const int t = g - n * a.n_frames;
        const float* src = a.wav + (size_t)n * a.n_samples + (size_t)t * a.hop;
#pragma unroll
        for (int q = 0; q < QV; ++q) dst[q] = *reinterpret_cast<const v2*>(src + 2 * (SUB * (l + 16 * q) + u));     // (8-byte aligned: launcher)
    };
    // (A frame's lanes never straddle a wavefront: the phases of a round are ordered by wave-local sync points.)
    for (int r = 0; r < rounds; ++r) {
        // ---------------- load (+ prefetch of the next round) + window + first radix-16 pass ----------------
        v2 v[16];
        if (r == 0) load_frame(r, xa);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = q < QV ? xa[q] * wnd[q] : (v2){0.f, 0.f};
        if (r + 1 < rounds) load_frame(r + 1, xa);
        pk_dft16(v);
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) s_x[unit * UNIT + k2 * XLD + l] = c_mul(v[k2], tw[k2]);
        wave_sync();
        // ---------------- transpose + second radix-16 pass ----------------
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = s_x[unit * UNIT + l * XLD + n1];
        pk_dft16(v);
        wave_sync();
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) s_x[unit * UNIT + 16 * k1 + l] = v[k1];     // bin 16 k1 + l
        wave_sync();
        // ---------------- real-FFT split -> 4 x power spectrum ----------------
        {
            const v2* E = s_x + (f * SUB) * UNIT;
            const v2* O = s_x + (f * SUB + SUB - 1) * UNIT;
            float* P = s_p + f * PLD;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = lf + LPF * i;                 // 0 .. NC/2-1
                const int kn = (SUB == 2) ? ((256 - k) & 255) : ((NC - k) & (NC - 1));
                v2 zk, zn;
                if (SUB == 2) {
                    zk = E[k] + c_mul(O[k], twc[i]);        // Z[k]      = E[k] + W512^k O[k]
                    zn = E[kn] + c_mulc(O[kn], twc[i]);     // Z[512-k]  = E[256-k] + conj(W512^k) O[256-k]
                } else {
                    zk = E[k];
                    zn = E[kn];
                }
                float plo, phi;
                pk_real_pair_power(zk, zn, twr[i], plo, phi);
                if (MAG) { plo = sqrtf(plo); phi = sqrtf(phi); }
                P[k] = plo;
                P[NC - k] = phi;
            }
            if (lf == 0) {                                  // the self-paired middle bin k = NC/2
                v2 z;
                if (SUB == 2) z = E[0] - O[0];              // Z[256] = E[0] - O[0]
                else z = E[NC / 2];
                float plo, phi;
                pk_real_pair_power(z, z, twmid, plo, phi);
                if (MAG) plo = sqrtf(plo);
                P[NC / 2] = plo;
            }
        }
        wave_sync();
        // ---------------- sparse mel: per-segment (up, down) sums, one packed FMA per bin ----------------
        {
            const float* P = s_p + f * PLD;
            v2* UD = s_ud + f * ULD;
            for (int i = 0;; ++i) {
                const int j = (i & 1) ? (i + 1) * LPF - 1 - lf : i * LPF + lf;
                if (i * LPF >= NSEG) break;
                if (j < NSEG) {
                    const int k0 = s_seg[j], k1 = s_seg[j + 1];
                    v2 ud = (v2){0.f, 0.f};
                    int k = k0;
                    for (; k + 4 <= k1; k += 4) {           // 4 bins per trip: the 8 LDS reads are independent
                        const float p0 = P[k], p1 = P[k + 1], p2 = P[k + 2], p3 = P[k + 3];
                        const v2 w0 = s_wud[k], w1 = s_wud[k + 1], w2 = s_wud[k + 2], w3 = s_wud[k + 3];
                        ud = __builtin_elementwise_fma(w0, (v2){p0, p0}, ud);
                        ud = __builtin_elementwise_fma(w1, (v2){p1, p1}, ud);
                        ud = __builtin_elementwise_fma(w2, (v2){p2, p2}, ud);
                        ud = __builtin_elementwise_fma(w3, (v2){p3, p3}, ud);
                    }
                    for (; k < k1; ++k) {
                        const float p = P[k];
                        ud = __builtin_elementwise_fma(s_wud[k], (v2){p, p}, ud);
                    }
                    UD[j] = ud;
                }
            }
        }
        wave_sync();
        // ---------------- log(mel + 1e-6) -> [mel][frame] ----------------
        {
            const v2* UD = s_ud + f * ULD;
#pragma unroll
            for (int i = 0; i < NMEL / LPF; ++i) {
                const int m = lf + LPF * i;
                const float mel = UD[m].x + UD[m + 1].y;    // up-slope of segment m + down-slope of segment m+1
                s_lm[m * 65 + r * FPR + f] = a.log_floor ? logf(fmaxf(mel, 1e-12f)) : logf(mel + 1e-6f);
            }
        }
    }
    __syncthreads();

    // ---------------- DCT-II (lane == frame, wave-uniform coefficients) + store ----------------
    const int fr = tid & 63;
    const int w = tid >> 6;
    const int g = blockIdx.x * fpw + fr;
    const bool valid = fr < fpw && g < a.total_frames;
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

// returns 1 (nothing launched) when the configuration needs the general kernel: unaligned frames, or a window whose
// valid radix-16 inputs differ from lane to lane
int launch_frontend_pk(int nc, const FrontendArgs& a0, hipStream_t s) {
    const FrontendArgs& a_in = a0;
    int grid = 0;
    const int sub = nc / 256;
    if (!a_in.aligned || (a_in.win & 1) || (a_in.win / 2) % (16 * sub) != 0) return 1;
    if (a_in.total_frames >= (1 << 23)) return 1;              // (the frame -> utterance split uses a float reciprocal)
    const int qv = a_in.win / (32 * sub);
    // Frames per workgroup: rounds x (4096 / nc) frames, at most 64.  All workgroups cost the same (rounds + ~0.85 of a
    // round for setup and the DCT), so the grid drains in dispatch waves of (2 workgroups x CUs); a last wave that leaves
    // every CU with one workgroup runs ~1.6x faster.  Model fitted on MI355X at B = 1024 .. 16384 (scripts/fe_rounds.py,
    // within 4 %); the round count that minimises it wins 4 % at B = 4096 and 8 % at B = 1024 over always using 64 frames.
    FrontendArgs a = a0;
    {
        const int fpr = 4096 / nc, max_rounds = 64 / fpr, slots = 2 * device_cus();
        int best = max_rounds;
        float best_cost = 3.4e38f;
        for (int r = max_rounds; r >= (max_rounds + 1) / 2; --r) {
            const int wgs = ceil_div(a.total_frames, r * fpr);
            const int full = wgs / slots, rest = wgs % slots;
            const float waves = (float)full + (rest == 0 ? 0.f : (2 * rest <= slots ? 0.6f : 1.f));
            const float cost = waves * ((float)r + 0.85f);
            if (cost < best_cost * 0.995f) { best_cost = cost; best = r; }
        }
        const int knob = tune_get(TCR_TUNE_FRONTEND);
        if (knob >= 10) best = min(max(knob - 10, 1), max_rounds);
        a.rounds = best;
        grid = ceil_div(a.total_frames, best * fpr);
    }
#define TCR_FPK(NC_, QV_)                                                                                           \
    if (nc == NC_ && qv == QV_) {                                                                                   \
        if (a.magnitude) hipLaunchKernelGGL((frontend_pk_kernel<NC_, QV_, true>), dim3(grid), dim3(256), 0, s, a);  \
        else hipLaunchKernelGGL((frontend_pk_kernel<NC_, QV_, false>), dim3(grid), dim3(256), 0, s, a);             \
        return check_launch("frontend_pk_kernel");                                                                  \
    }
    TCR_FPK(512, 10)    // 40 ms window @ 16 kHz, FFT 1024
    TCR_FPK(256, 15)    // 30 ms window, FFT 512
    TCR_FPK(512, 16)
    TCR_FPK(256, 16)
    TCR_FPK(256, 10)    // 20 ms window, FFT 512
    TCR_FPK(512, 15)
#undef TCR_FPK
    return 1;
}

}  // namespace tcr
