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
//
// Mel filterbank: sparse (two slopes per bin), load-balanced -- the plan cuts every mel-edge segment into items of <= 8 bins and a
// lane takes one item per trip with all 16 LDS reads of the trip in flight (one lane per SEGMENT meant data-dependent loops of
// 2 .. 20 bins with the wave waiting for the longest: 20 % of the kernel by a timing what-if).  DCT-II: on the matrix cores
// (exact-f32 v_mfma_f32_16x16x4_f32, otherwise idle here), A fragments from a coalesced plan table (the VALU form cost 7 %).
// Tried and dropped: the filterbank as banded MFMA work shared by the four waves (127 k-steps per round) -- needs two workgroup
// barriers per round and partial-tile sums through LDS: 325 us vs 261 us.
#include "frontend_plan.h"
#include "frontend_args.h"

namespace tcr {

// (v2 / v4, the packed-FP32 complex idioms c_*, row_swap, lane_gather, fast_log, pk_sq_pair: gfx950_isa.h)

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
    pk_sq_pair(A, C, p_lo, p_hi);
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
    constexpr int PLD = NBINS + (LPF >= 32 ? 31 : 15);      // row stride: 16 lanes per frame put TWO frames into a 32-lane ds_read_b32 / ds_write_b32 group
                                            // (banks are mod 32 for these): their rows sit 16 banks apart (stride = 16 mod 32) and a trip's items
                                            // start in distinct bins mod 16; at 32 lanes per frame a group is one frame (distinct mod 32)
    constexpr int kMelItemBins = mel_item_bins(NC);
    constexpr bool kItemsLds = QV >= 15;    // mel item descriptors read from LDS per trip instead of held in registers (see kDctPre)
    constexpr int LMS = 80;                 // log-mel row stride, = 16 (mod 32): the DCT's MFMA B fragment reads 32 distinct banks

    __shared__ v2 s_x[16 * UNIT];                   // transpose tiles, then the FFT output of each unit; after the real-FFT split the
                                                    // frame's first unit holds its mel items' (up, down) partial sums (wave-local reuse)
    __shared__ float s_p[FPR * PLD];                // 4 x power (or 2 x magnitude) spectrum
    __shared__ float s_lm[NMEL * LMS];              // log-mel [mel][frame], 64 frames
    constexpr int NIT = mel_items_fast(NC);         // items of the unrolled trips
    __shared__ v2 s_wit[kMelItemBins * NIT];        // mel slopes [bin of the item][item], pre-scaled by 1/4 (1/2), zero past an item's end
    __shared__ int s_items[kItemsLds ? kMelItemsMax : 1];
    static_assert(kMelItemsMax < UNIT && NIT < UNIT, "the item sums of a frame (and the empty slots' cell, mel_dummy_item) live in one transpose unit");

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
        const v2* wit = reinterpret_cast<const v2*>(a.mel_wit);
        for (int i = threadIdx.x; i < kMelItemBins * NIT; i += 256) s_wit[i] = wit[i] * fold;
        for (int i = threadIdx.x; i < FPR * (PLD - NBINS); i += 256)       // row pads: read (times a zero slope) past an item's end
            s_p[(i / (PLD - NBINS)) * PLD + NBINS + i % (PLD - NBINS)] = 0.f;
        (void)wud;
    }
    const int nitems = a.mel_ifirst[NSEG];
    // round-invariant: the items this lane takes (one per trip) and the item ranges of the bands it finishes
    constexpr int TRIPS = mel_trips(NC);
    int item_d[kItemsLds ? 1 : TRIPS], band_i[NMEL / LPF];
    if (kItemsLds) {
        for (int i = threadIdx.x; i < kMelItemsMax; i += 256) s_items[i] = a.mel_items[i];
    } else {
#pragma unroll
        for (int tr = 0; tr < TRIPS; ++tr) item_d[kItemsLds ? 0 : tr] = a.mel_items[min(lf + LPF * tr, kMelItemsMax - 1)];
    }
#pragma unroll
    for (int i = 0; i < NMEL / LPF; ++i) {
        const int m = lf + LPF * i;
        band_i[i] = a.mel_ifirst[m] | (a.mel_ifirst[m + 1] << 8) | (a.mel_ifirst[m + 2] << 16);
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const v2 wm = tw_real[NC / 2];
    const v2 twmid = (v2){wm.y, -wm.x};
    // (no barrier here: the LDS tables staged above are first read by round 0's mel phase -- the barrier sits there, behind the FFT)

    const float inv_frames = 1.0f / (float)a.n_frames;
    const int rounds = a.rounds;            // <= ROUNDS; chosen by the launcher so that the grid fills whole dispatch waves
    const int fpw = rounds * FPR;           // frames per workgroup
    v2 xa[QV];
    auto load_frame = [&](int chunk, int rr, v2 (&dst)[QV]) {
        int g = chunk * fpw + rr * FPR + f;
        g = min(g, a.total_frames - 1);
        int n = (int)(((float)g + 0.5f) * inv_frames);          // g / n_frames: float multiply + one-step fix-up
        n += (n + 1) * a.n_frames <= g ? 1 : (n * a.n_frames > g ? -1 : 0);
        const int t = g - n * a.n_frames;
        const float* src = a.wav + (size_t)n * a.n_samples + (size_t)t * a.hop;
#pragma unroll
        for (int q = 0; q < QV; ++q) dst[q] = *reinterpret_cast<const v2*>(src + 2 * (SUB * (l + 16 * q) + u));     // (8-byte aligned: launcher)
    };
    // DCT A fragments of the first kDctPre coefficient tiles: round-invariant, loaded once (their latency would otherwise be paid
    // at the end of every chunk)
    // 30 / 20 ms windows (QV 15 / 16): 10 - 12 more window + prefetch registers than the 40 ms window.  Preloading three DCT tiles
    // spilled 20 - 72 B per lane there; one tile is spill-free but exposes the other tiles' loads at every chunk end (250 vs 234 us);
    // two tiles + the mel item descriptors read from LDS instead of held in registers: spill-free AND as fast.
    constexpr int kDctPre = QV >= 15 ? (MAG ? 1 : 2) : 3;      // (the magnitude variants -- deploy path, log-mel -- carry the square roots: one tile)
    float dcta[kDctPre][NMEL / 4];
    if (!a.no_dct) {
#pragma unroll
        for (int ct = 0; ct < kDctPre; ++ct)
#pragma unroll
            for (int st = 0; st < NMEL / 4; ++st) dcta[ct][st] = a.dct_tab[(ct * (NMEL / 4) + st) * 64 + (tid & 63)];
    }
    // Persistent workgroups: a workgroup walks chunks of `fpw` frames (tables, twiddles and DCT fragments are set up once; as one
    // workgroup per chunk the set-up, DCT and drain cost 4.6 us per workgroup against 4.0 us per round).
    const int nchunks = (a.total_frames + fpw - 1) / fpw;
    load_frame(blockIdx.x, 0, xa);
#pragma unroll 1
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    // lane geometry re-derived per chunk from an opaque zero: the address arithmetic of the unrolled round body must not be hoisted
    // out of the chunk loop (it would stay live through the DCT section)
    const int tid = (int)threadIdx.x + opaque_zero();
    const int f = tid / LPF, lf = tid % LPF, u = lf >> 4, l = tid & 15, unit = tid >> 4;
    // (A frame's lanes never straddle a wavefront: the phases of a round are ordered by wave-local sync points.)
    for (int r = 0; r < rounds; ++r) {
        // ---------------- load (+ prefetch of the next round) + window + first radix-16 pass ----------------
        v2 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = q < QV ? xa[q] * wnd[q] : (v2){0.f, 0.f};
        if (r + 1 < rounds) load_frame(chunk, r + 1, xa);
        else if (chunk + (int)gridDim.x < nchunks) load_frame(chunk + gridDim.x, 0, xa);
        pk_dft16(v);
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) s_x[unit * UNIT + k2 * XLD + l] = c_mul(v[k2], tw[k2]);
        wave_sync();
        // ---------------- transpose + second radix-16 pass ----------------
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) v[n1] = s_x[unit * UNIT + l * XLD + n1];
        pk_dft16(v);                                                                // v[k1] = bin 16 k1 + l of this lane's unit
        // ---------------- real-FFT split -> 4 x power spectrum, operands moved between lanes in registers ----------------
        // Lane (h, l) of a frame (h: its 256-point unit, l: lane in the unit) finishes bins k_i = lf + LPF i, i < 8:
        //   nfft 1024: k_i = 16 (2 i + h) + l -- E[k_i], O[k_i] are registers 2 i + h of lanes (0, l), (1, l): one row swap per pair
        //              hands both to the lane; Z[k] = E + W^k O and Z[k + 256] = E - W^k O (the butterfly's other output);
        //              Z[512 - k_i] = Z[k' + 256] with k' = 256 - k_i, finished by lane (1 - h, 16 - l) as its bin 7 - i;
        //   nfft  512: k_i = 16 i + l is register i; Z[256 - k_i] is register 15 - i of lane 16 - l.
        // Lanes with l = 0 pair with themselves (a gather from self), except lane 0 of the frame, whose partner registers are
        // one further along (bins 16 * even) and whose bin 0 pairs with itself.
        {
            float* P = s_p + f * PLD;
            const int lane = tid & 63;
            const int partner = (lane & ~(LPF - 1)) | (SUB == 2 ? (l ? (16 * (1 - u) + 16 - l) : 16 * u) : ((16 - l) & 15));
            if (SUB == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float ex = v[2 * i].x, ey = v[2 * i].y, ox = v[2 * i + 1].x, oy = v[2 * i + 1].y;
                    row_swap(ex, ox, lane);
                    row_swap(ey, oy, lane);
                    const v2 t = c_mul((v2){ox, oy}, twc[i]);
                    const v2 e = (v2){ex, ey};
                    v[2 * i] = e + t;                       // Z[k_i]
                    v[2 * i + 1] = e - t;                   // Z[k_i + 256]
                }
            }
            v2 prev = v[0];                                 // lane 0 of the frame: bin 0 pairs with itself, and its partner register
#pragma unroll                                              // of step i is the one its self-gather of step i - 1 returned
            for (int i = 0; i < 8; ++i) {
                const int k = lf + LPF * i;                 // 0 .. NC/2-1
                const v2 zk = SUB == 2 ? v[2 * i] : v[i];
                const v2 src = SUB == 2 ? v[15 - 2 * i] : v[15 - i];
                const v2 g = (v2){lane_gather(src.x, partner), lane_gather(src.y, partner)};
                const v2 zn = lf == 0 ? prev : g;
                prev = g;
                float plo, phi;
                pk_real_pair_power(zk, zn, twr[i], plo, phi);
                if (MAG) { plo = sqrtf(plo); phi = sqrtf(phi); }
                P[k] = plo;
                P[NC - k] = phi;
            }
            if (lf == 0) {                                  // the self-paired middle bin k = NC/2
                const v2 z = SUB == 2 ? v[1] : v[8];        // Z[256] = E[0] - O[0]; nfft 512: bin 128 = register 8
                float plo, phi;
                pk_real_pair_power(z, z, twmid, plo, phi);
                if (MAG) plo = sqrtf(plo);
                P[NC / 2] = plo;
            }
        }
        wave_sync();
        if (r == 0) __syncthreads();                        // the slope table / zeroed row pads staged in the prologue; s_lm free again
        // ---------------- sparse mel: one item (<= 8 bins of one segment) per lane and trip, one packed FMA per bin ----------------
        v2* UD = s_x + (f * SUB) * UNIT;                    // (this frame's first unit: dead until the next round's first pass)
        {
            const float* P = s_p + f * PLD;
#pragma unroll 1
            for (int tr = 0; tr < TRIPS; ++tr) {            // (rolled: one trip's 16 reads in flight, not all trips' -- register budget)
                const int it = lf + LPF * tr;
                int d;
                if (kItemsLds) {
                    d = s_items[min(it, kMelItemsMax - 1)];
                } else {
                    d = item_d[0];
#pragma unroll
                    for (int q = 1; q < TRIPS; ++q) d = tr == q ? item_d[kItemsLds ? 0 : q] : d;
                }
                const float* pk = P + (d & 1023);            // (items past the filterbank's last: bin 0 with zero slopes)
                const v2* wk = s_wit + it;
                float p[kMelItemBins];
                v2 wv[kMelItemBins];
#pragma unroll
                for (int b = 0; b < kMelItemBins; ++b) {    // constant offsets from two per-lane bases: no per-bin address arithmetic,
                    p[b] = pk[b];                           // no masks (slopes past the item's end are zero, the row pad is zero)
                    wv[b] = wk[b * NIT];
                }
                v2 ud = (v2){0.f, 0.f};
#pragma unroll
                for (int b = 0; b < kMelItemBins; ++b) ud = __builtin_elementwise_fma(wv[b], (v2){p[b], p[b]}, ud);
                UD[(d >> 21) & 255] = ud;                    // (the slot's LOGICAL item: the log phase adds a band's items in logical order)
            }
            if (nitems > NIT) {                             // (filterbanks with more items than the trips cover: slow path)
                const float fold = MAG ? 0.5f : 0.25f;
                const v2* wud = reinterpret_cast<const v2*>(a.wud);
                for (int it = lf + NIT; it < nitems; it += LPF) {
                    const int d = a.mel_items[it];
                    const int k0 = d & 1023, nb = (d >> 10) & 15;
                    v2 ud = (v2){0.f, 0.f};
                    for (int b = 0; b < nb; ++b) ud = __builtin_elementwise_fma(wud[k0 + b] * fold, (v2){P[k0 + b], P[k0 + b]}, ud);
                    UD[it] = ud;
                }
            }
        }
        wave_sync();
        // ---------------- log(mel + 1e-6) -> [mel][frame]: up-slope items of segment m + down-slope items of segment m + 1 ----------------
        {
#pragma unroll
            for (int i = 0; i < NMEL / LPF; ++i) {
                const int m = lf + LPF * i;
                const int i0 = band_i[i] & 255, i1 = (band_i[i] >> 8) & 255, i2 = band_i[i] >> 16;
                constexpr int MAXC = NC == 512 ? 3 : 2;     // items per segment read unconditionally (segments: <= 20 bins / 8, <= 10 bins / 4 mostly <= 8)
                float up[MAXC], dn[MAXC];
#pragma unroll
                for (int c = 0; c < MAXC; ++c) {            // all reads in flight
                    up[c] = UD[min(i0 + c, UNIT - 1)].x;
                    dn[c] = UD[min(i1 + c, UNIT - 1)].y;
                }
                float mel = 0.f;
#pragma unroll
                for (int c = 0; c < MAXC; ++c) mel += i0 + c < i1 ? up[c] : 0.f;
                for (int it = i0 + MAXC; it < i1; ++it) mel += UD[it].x;
#pragma unroll
                for (int c = 0; c < MAXC; ++c) mel += i1 + c < i2 ? dn[c] : 0.f;
                for (int it = i1 + MAXC; it < i2; ++it) mel += UD[it].y;
                s_lm[m * LMS + r * FPR + f] = fast_log(a.log_floor ? fmaxf(mel, 1e-12f) : mel + 1e-6f);
            }
        }
        wave_sync();            // the item sums live in the unit the next round's first pass overwrites
    }
    __syncthreads();

    // ---------------- DCT-II on the matrix cores: wave w owns frames 16 w .. 16 w + 15, all coefficient tiles ----------------
    if (a.no_dct) {
        const int fr = tid & 63;
        const int w = tid >> 6;
        const int g = chunk * fpw + fr;
        const bool valid = fr < fpw && g < a.total_frames;
        const int gg = valid ? g : a.total_frames - 1;
        const int n = gg / a.n_frames;
        const int t = gg - n * a.n_frames;
        float* dst = a.out + (size_t)n * a.n_coef * a.tp + kHalo + t;
        for (int m = w; m < a.n_coef; m += 4) {
            if (valid) {
                float* row = dst + (size_t)m * a.tp;
                row[0] = s_lm[m * LMS + fr];
                if (t == 0) { row[-4] = 0.f; row[-3] = 0.f; row[-2] = 0.f; row[-1] = 0.f; }
                if (t == a.n_frames - 1) { row[1] = 0.f; row[2] = 0.f; row[3] = 0.f; row[4] = 0.f; }
            }
        }
    } else {
        const int lane = tid & 63;
        const int kq = lane >> 4, col = lane & 15;
        const int fr = 16 * wave + col;
        const int g = chunk * fpw + fr;
        const bool valid = fr < fpw && g < a.total_frames;
        const int gg = valid ? g : a.total_frames - 1;
        const int n = gg / a.n_frames;
        const int t = gg - n * a.n_frames;
        float* dst = a.out + (size_t)n * a.n_coef * a.tp + kHalo + t;
        constexpr int CT = NMEL / 16;                                   // coefficient tiles (n_coef <= NMEL)
        v4 acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = (v4){0.f, 0.f, 0.f, 0.f};
        const int ntile = (a.n_coef + 15) >> 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if (ct < ntile ) {                        // (wave-uniform)
                float av[NMEL / 4], bv[NMEL / 4];
#pragma unroll
                for (int st = 0; st < NMEL / 4; ++st) {                 // all fragments of the tile in flight, then the MFMA chain
                    av[st] = ct < kDctPre ? dcta[ct < kDctPre ? ct : 0][st] : a.dct_tab[(ct * (NMEL / 4) + st) * 64 + lane];
                    bv[st] = s_lm[(4 * st + kq) * LMS + fr];
                }
#pragma unroll
                for (int st = 0; st < NMEL / 4; ++st) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st], bv[st], acc[ct], 0, 0, 0);
            }
        }
        if (valid) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int c = 16 * ct + 4 * kq + rr;
                    if (c < a.n_coef) {
                        float* row = dst + (size_t)c * a.tp;
                        row[0] = acc[ct][rr];
                        if (t == 0) { row[-4] = 0.f; row[-3] = 0.f; row[-2] = 0.f; row[-1] = 0.f; }
                        if (t == a.n_frames - 1) { row[1] = 0.f; row[2] = 0.f; row[3] = 0.f; row[4] = 0.f; }
                    }
                }
            }
        }
    }
    // (no barrier here: the next chunk's first write to s_lm sits behind round 0's barrier below, which every wave reaches only
    //  after its DCT of this chunk)
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
    // Persistent workgroups (2 per CU) walk chunks of rounds x (4096 / nc) frames, at most 64: tables, twiddles and DCT fragments are
    // set up once per workgroup.  A chunk costs rounds + ~1.45 rounds (DCT, stores, barriers, the sample-prefetch bubble); the
    // workgroups take ceil(chunks / slots) of them, a last generation that leaves every CU one workgroup runs ~1.6x faster.  The round
    // count that minimises that model matched the measured best at B = 1024 (5), 4096 (7), 16384 (8) (scripts/fe_rounds_fit.py).
    FrontendArgs a = a0;
    {
        const int fpr = 4096 / nc, max_rounds = 64 / fpr, slots = 2 * device_cus();
        int best = max_rounds;
        float best_cost = 3.4e38f;
        // small batches (fewer chunks than half the slots even at one round per chunk... the latency regime): down to ONE round per
        // chunk, so that an utterance's frames spread over several workgroups instead of queueing in one (batch 1: 7 workgroups x 1
        // round instead of 1 workgroup x 7 rounds)
        const int r_min = ceil_div(a.total_frames, fpr) <= slots / 2 ? 1 : (max_rounds + 1) / 2;
        for (int r = max_rounds; r >= r_min; --r) {
            const int chunks = ceil_div(a.total_frames, r * fpr);
            const int full = chunks / slots, rest = chunks % slots;
            const float gens = (float)full + (rest == 0 ? 0.f : (2 * rest <= slots ? 0.6f : 1.f));
            const float cost = gens * ((float)r + 1.45f);
            if (cost < best_cost * 0.995f) { best_cost = cost; best = r; }
        }
        const int knob = tune_get(TCR_TUNE_FRONTEND);
        if (knob >= 10) best = min(max(knob - 10, 1), max_rounds);
        if (a0.rounds > 0) best = min(a0.rounds, max_rounds);      // (tcr_frontend_fwd_rounds: the caller's per-call choice)
        a.rounds = best;
        grid = min(ceil_div(a.total_frames, best * fpr), slots);
        const int cap = tune_get(TCR_TUNE_FE_GRID);
        if (cap > 0) grid = min(grid, cap);
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
