// Packed-FP32 fused MFCC / log-mel front-end at THREE waves per SIMD (the arithmetic of frontend_pk.hip, its decomposition in
// frontend.hip; reference semantics: datasets/preprocessors.py:64-96,183-194).
//
// frontend_pk.hip holds 245 VGPRs and 78.8 KB of LDS per 4-wave workgroup: two waves per SIMD, VALU 47 % and LDS 46 % busy, a quarter
// of the wave-cycles waiting (profiles/r04_final_pmc.csv).  This build fits three workgroups per CU -- <= 168 VGPRs, <= 53 KB:
//   * every LDS region is WAVE-LOCAL (a wave owns its frames from the samples to the DCT): one workgroup barrier in the kernel's life
//     (the slope table), the waves of a workgroup drift apart and cover each other's LDS / memory round trips;
//   * the 16 x 16 transpose between the two radix-16 passes goes through HALF a tile (16 rows x 8 columns) in two passes, in place
//     in the registers and without selects: lanes 8..15 of a unit run pass one on (-1)^n x[n] (the sign lives in their window
//     table), so their register j holds bin j ^ 8 (shift theorem) and "registers 0..7" are the diagonal blocks of the transpose for
//     every lane; what they receive sits in register n ^ 8, i.e. pass two sees its input rotated by 8 and its odd outputs negated
//     -- put right by eight packed multiplies with a per-lane +-1;
//   * the power spectrum reuses the tile's space, the item sums are compact, the log-mel block is [mel][16 frames] per wave with an
//     XOR swizzle (conflict-free for the log phase's column writes and the DCT's B-fragment reads without padding);
//   * the next round's samples are requested AFTER the real-FFT split (the FFT registers are dead by then), the DCT's A fragments
//     are fetched per chunk (L1 / L2 hits) instead of living in 48 registers.
// Results are bitwise those of frontend_pk.hip (same operations in the same order on every value).
#include "frontend_plan.h"
#include "frontend_args.h"

namespace tcr {

// (v2 / v4, the packed-FP32 complex idioms c_*, row_swap, lane_gather, fast_log, pk_sq_pair: gfx950_isa.h)

namespace {

// 4-point forward DFT (W4 = -i), in place: 8 packed instructions.
__device__ __forceinline__ void pk3_dft4(v2& a, v2& b, v2& c, v2& d) {
    const v2 t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
    a = t0 + t2;
    c = t0 - t2;
    b = c_submi(t1, t3);
    d = c_addmi(t1, t3);
}

// 16-point forward DFT in registers, natural order in and out (4 x 4 Cooley-Tukey) -- the operation order of frontend_pk.hip's.
__device__ __forceinline__ void pk3_dft16(v2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128673848f;   // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f;   // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;   // sqrt(1/2)
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) pk3_dft4(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);
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
    for (int k1 = 0; k1 < 4; ++k1) pk3_dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i + 1; j < 4; ++j) {
            const v2 t = v[4 * i + j];
            v[4 * i + j] = v[4 * j + i];
            v[4 * j + i] = t;
        }
}

__device__ __forceinline__ void pk3_real_pair_power(v2 zk, v2 zn, v2 wmi, float& p_lo, float& p_hi) {
    const v2 A = c_addc(zk, zn), D = c_subc(zk, zn);
    const v2 C = c_mul(D, wmi);
    pk_sq_pair(A, C, p_lo, p_hi);
}

}  // namespace

// LDS of one workgroup (bytes): 4 waves x (tile / power 4608 + item sums FPWV x 832 + log-mel 4096) + slopes + items
//   nfft 1024: 18432 + 6656 + 16384 + 6144       = 47616      nfft 512: 18432 + 13312 + 16384 + 3072 + 768 = 51968
// three workgroups per CU: <= 54613.
template <int NC, int QV, bool MAG>
__global__ __launch_bounds__(256, 3) void frontend_pk3_kernel(const FrontendArgs a) {
    constexpr int LPF = NC / 16;            // lanes per frame
    constexpr int FPR = 256 / LPF;          // frames per round (workgroup)
    constexpr int FPWV = FPR / 4;           // frames per round and wave
    constexpr int SUB = NC / 256;           // 256-point units per frame
    constexpr int NBINS = NC + 1;
    constexpr int NMEL = 64, NSEG = NMEL + 1;
    constexpr int RS = 9;                   // row of the 16 x 8 half tile (v2 units): odd -> the 16 lanes of a ds_write_b64 group hit 32 banks
    constexpr int UNITSZ = 16 * RS;         // half tile of one 256-point unit
    constexpr int TILEW = 4 * UNITSZ;       // v2 per wave (4 units)
    constexpr int PLD = NBINS + (LPF >= 32 ? 31 : 15);      // power-spectrum row stride (frontend_pk.hip)
    static_assert(FPWV * PLD <= 2 * TILEW, "the power spectrum of a wave's frames lives in its tile space");
    constexpr int kMelItemBins = mel_item_bins(NC);
    constexpr bool kItemsLds = true;        // the trips' item descriptors: one ds_read_b32 per trip instead of TRIPS registers
    constexpr int NIT = mel_items_fast(NC);         // items of the unrolled trips; the launcher sends filterbanks with more to frontend_pk.hip
    constexpr int USZ = NIT + 8; static_assert(NIT + 3 <= USZ, "the log phase reads up to three cells from a band's first item");
                       // item sums of one frame (+ the cell of the empty slots, mel_dummy_item, + the log phase's read-ahead)
    static_assert(mel_dummy_item(NC, NIT) < USZ, "the empty slots' cell lies inside the item-sum row");
    constexpr int TRIPS = mel_trips(NC);

    __shared__ v2 s_t[4 * TILEW];
    __shared__ v2 s_ud[4 * FPWV * USZ];
    __shared__ float s_lm[4 * NMEL * 16];
    __shared__ v2 s_wit[kMelItemBins * NIT];
    __shared__ int s_items[kItemsLds ? NIT : 1];
    constexpr bool kBandsLds = true;                // the bands' item ranges from LDS (one ds_read_b32 per band and round instead of 2 - 4 registers)
    __shared__ int s_band[kBandsLds ? NMEL : 1];
    __shared__ v2 s_wnd[QV * LPF];          // signed window [q][lane of the frame]: samples 2 (SUB (l + 16 q) + u), + 1

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const float fold = MAG ? 0.5f : 0.25f;
        const v2* wit = reinterpret_cast<const v2*>(a.mel_wit);
        for (int i = tid; i < kMelItemBins * NIT; i += 256) s_wit[i] = wit[i] * fold;
        if (kItemsLds)
            for (int i = tid; i < NIT; i += 256) s_items[i] = a.mel_items[i];
        if (kBandsLds && tid < NMEL) s_band[tid] = a.mel_ifirst[tid] | (a.mel_ifirst[tid + 1] << 8) | (a.mel_ifirst[tid + 2] << 16);
        for (int i = tid; i < QV * LPF; i += 256) {
            const int q = i / LPF, lfi = i % LPF;
            s_wnd[i] = *reinterpret_cast<const v2*>(a.window_sgn + 2 * (SUB * ((lfi & 15) + 16 * q) + (lfi >> 4)));
        }
    }
    const v2* tw256 = reinterpret_cast<const v2*>(a.tw256);
    const v2* tw_real = reinterpret_cast<const v2*>(a.tw_real);
    const v2* tw_combine = reinterpret_cast<const v2*>(a.tw_combine);
    // round-invariant per-lane tables that stay in registers: the real-FFT split's and the even / odd recombination's twiddles.  The
    // window (20 registers) and the inter-pass twiddles (32) are re-read every round from L1 -- the vector-memory path is otherwise
    // idle (ten sample loads per round), the LDS pipe is not -- so that 3 waves per SIMD (<= 168 registers) hold without scratch.
    v2 tw[16], twr[8], twc[8];
    float sgn;
    int item_d[kItemsLds ? 1 : TRIPS], band_i[kBandsLds ? 1 : NMEL / LPF];
    {
        const int lf = tid % LPF, l = tid & 15, h = l >> 3;
        sgn = h ? -1.f : 1.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) tw[j] = tw256[l * 16 + (j ^ (8 * h))];    // lanes 8..15: register j holds bin j ^ 8 after the first pass
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = lf + LPF * i;
            const v2 w = tw_real[k];
            twr[i] = (v2){w.y, -w.x};           // -i W^k
            twc[i] = (SUB == 2) ? tw_combine[k] : (v2){1.f, 0.f};
        }
        if (!kItemsLds) {
#pragma unroll
            for (int tr = 0; tr < TRIPS; ++tr) item_d[kItemsLds ? 0 : tr] = a.mel_items[lf + LPF * tr];
        }
        if (!kBandsLds) {
#pragma unroll
            for (int i = 0; i < NMEL / LPF; ++i) {
                const int m = lf + LPF * i;
                band_i[kBandsLds ? 0 : i] = a.mel_ifirst[m] | (a.mel_ifirst[m + 1] << 8) | (a.mel_ifirst[m + 2] << 16);
            }
        }
    }
    const v2 wm = tw_real[NC / 2];
    const v2 twmid = (v2){wm.y, -wm.x};

    const float inv_frames = 1.0f / (float)a.n_frames;
    const int rounds = a.rounds;            // <= 16 / FPWV; chosen by the launcher
    const int fpwv = rounds * FPWV;         // frames per chunk and wave (the columns of its DCT tile)
    const int fpw = 4 * fpwv;               // frames per chunk
    v2 xa[QV];
    auto load_frame = [&](int chunk, int rr, v2 (&dst)[QV]) {
        const int tid = (int)threadIdx.x;
        const int lf = tid % LPF, u = lf >> 4, l = tid & 15, fw = (tid & 63) / LPF;
        int g = chunk * fpw + wave * fpwv + rr * FPWV + fw;
        g = min(g, a.total_frames - 1);
        int n = (int)(((float)g + 0.5f) * inv_frames);          // g / n_frames: float multiply + one-step fix-up
        n += (n + 1) * a.n_frames <= g ? 1 : (n * a.n_frames > g ? -1 : 0);
        const int t = g - n * a.n_frames;
        const float* src = a.wav + (size_t)n * a.n_samples + (size_t)t * a.hop;
#pragma unroll
        for (int q = 0; q < QV; ++q) dst[q] = *reinterpret_cast<const v2*>(src + 2 * (SUB * (l + 16 * q) + u));     // (8-byte aligned: launcher)
    };
    const int nchunks = (a.total_frames + fpw - 1) / fpw;
    load_frame(blockIdx.x, 0, xa);
    __syncthreads();                        // the slope table / item list staged above (the only workgroup barrier)
    // De-phasing: the twelve waves of a CU start together and do identical work, so they would stay in lock step -- all in their VALU
    // phases, then all queueing at the LDS.  A one-off delay of (workgroup generation * 4 + wave) * stagger * 64 cycles spreads them over a round.
    if (a.stagger > 0) {
        const int steps = (((int)blockIdx.x / a.stagger_div) * 4 + wave) * a.stagger;
        for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(1);
    }

#pragma unroll 1
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    // lane geometry re-derived per chunk from an opaque zero: the address arithmetic of the unrolled round body must not be hoisted
    // out of the chunk loop (it would stay live through the DCT section)
    const int tid = (int)threadIdx.x + opaque_zero();
    const int lane = tid & 63;
    const int lf = tid % LPF, u = lf >> 4, l = tid & 15, h = l >> 3;
    const int fw = lane / LPF;              // frame of the wave in this round
    v2* T = s_t + wave * TILEW + (lane >> 4) * UNITSZ;
    float* Pw = reinterpret_cast<float*>(s_t + wave * TILEW);
    float* LM = s_lm + wave * (NMEL * 16);
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {
        // ---------------- window + first radix-16 pass ----------------
        // The window comes from LDS every round (20 registers for a moment instead of for the kernel's life; lanes 8..15 of a unit:
        // x[n] (-1)^n -- a.window_sgn carries the sign -- so that their register j holds bin j ^ 8 after the pass).
        v2 v[16];
        {
            const v2* wsrc = s_wnd + lf;
            v2 wnd[QV];
#pragma unroll
            for (int q = 0; q < QV; ++q) wnd[q] = wsrc[q * LPF];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = q < QV ? xa[q] * wnd[q] : (v2){0.f, 0.f};
        }
        pk3_dft16(v);                                           // register j: bin j ^ 8h of this lane's 16 samples
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = c_mul(v[j], tw[j]);
        // ---------------- transpose through half a tile, two passes (diagonal blocks, then off-diagonal blocks) ----------------
        {
            v2* wa = T + (8 * h) * RS + (l & 7);
            v2* wb = T + (8 * (1 - h)) * RS + (l & 7);
            const v2* rd = T + l * RS;
#pragma unroll
            for (int j = 0; j < 8; ++j) wa[j * RS] = v[j];
            wave_sync();
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = rd[c];
            wave_sync();
#pragma unroll
            for (int j = 0; j < 8; ++j) wb[j * RS] = v[8 + j];
            wave_sync();
#pragma unroll
            for (int c = 0; c < 8; ++c) v[8 + c] = rd[c];
        }
        // ---------------- second radix-16 pass: register n holds input n ^ 8h -> odd outputs carry (-1)^h ----------------
        pk3_dft16(v);
        {
            const v2 s2 = (v2){sgn, sgn};
#pragma unroll
            for (int k1 = 1; k1 < 16; k1 += 2) v[k1] = v[k1] * s2;
        }
        wave_sync();                                            // (the tile space becomes the power spectrum)
        // ---------------- real-FFT split -> 4 x power spectrum (frontend_pk.hip) ----------------
        {
            float* P = Pw + fw * PLD;
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
            v2 prev = v[0];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = lf + LPF * i;                 // 0 .. NC/2-1
                const v2 zk = SUB == 2 ? v[2 * i] : v[i];
                const v2 src = SUB == 2 ? v[15 - 2 * i] : v[15 - i];
                const v2 g = (v2){lane_gather(src.x, partner), lane_gather(src.y, partner)};
                const v2 zn = lf == 0 ? prev : g;
                prev = g;
                float plo, phi;
                pk3_real_pair_power(zk, zn, twr[i], plo, phi);
                if (MAG) { plo = sqrtf(plo); phi = sqrtf(phi); }
                P[k] = plo;
                P[NC - k] = phi;
            }
            if (lf == 0) {                                  // the self-paired middle bin k = NC/2
                const v2 z = SUB == 2 ? v[1] : v[8];
                float plo, phi;
                pk3_real_pair_power(z, z, twmid, plo, phi);
                if (MAG) plo = sqrtf(plo);
                P[NC / 2] = plo;
            }
            if (lf < PLD - NBINS) P[NBINS + lf] = 0.f;      // row pad: read (times a zero slope) past an item's end; the space held tile data
        }
        // ---------------- the next round's samples (the FFT registers are free) ----------------
        if (r + 1 < rounds) load_frame(chunk, r + 1, xa);
        else if (chunk + (int)gridDim.x < nchunks) load_frame(chunk + gridDim.x, 0, xa);
        wave_sync();
        // ---------------- sparse mel: one item (<= 8 bins of one segment) per lane and trip, one packed FMA per bin ----------------
        v2* UD = s_ud + (wave * FPWV + fw) * USZ;
        int bi_pre[NMEL / LPF];                             // the bands' item ranges: requested with the trips' operands, used by the log phase
        if (kBandsLds) {
#pragma unroll
            for (int i = 0; i < NMEL / LPF; ++i) bi_pre[i] = s_band[lf + LPF * i];
        }
        {
            const float* P = Pw + fw * PLD;
#pragma unroll
            for (int tr = 0; tr < TRIPS; ++tr) {            // (unrolled: the trips' descriptor and operand reads overlap instead of TRIPS dependent LDS round trips in a row)
                const int it = lf + LPF * tr;
                int d;
                if (kItemsLds) {
                    d = s_items[it];
                } else {
                    d = item_d[0];
#pragma unroll
                    for (int q = 1; q < TRIPS; ++q) d = tr == q ? item_d[kItemsLds ? 0 : q] : d;
                }
                const float* pk = P + (d & 1023);
                const v2* wk = s_wit + it;
                float p[kMelItemBins];
                v2 wv[kMelItemBins];
#pragma unroll
                for (int b = 0; b < kMelItemBins; ++b) {
                    p[b] = pk[b];
                    wv[b] = wk[b * NIT];
                }
                v2 ud = (v2){0.f, 0.f};
#pragma unroll
                for (int b = 0; b < kMelItemBins; ++b) ud = __builtin_elementwise_fma(wv[b], (v2){p[b], p[b]}, ud);
                UD[(d >> 21) & 255] = ud;                    // (the slot's LOGICAL item; empty slots: mel_dummy_item)
            }
        }
        wave_sync();
        // ---------------- log(mel + 1e-6) -> [mel][frame ^ swizzle] ----------------
        {
            const int c = r * FPWV + fw;                    // column of the wave's DCT tile
#pragma unroll
            for (int i = 0; i < NMEL / LPF; ++i) {
                const int m = lf + LPF * i;
                const int bi = kBandsLds ? bi_pre[i] : band_i[kBandsLds ? 0 : i];
                const int i0 = bi & 255, i1 = (bi >> 8) & 255, i2 = bi >> 16;
                constexpr int MAXC = NC == 512 ? 3 : 2;
                float up[MAXC], dn[MAXC];
#pragma unroll
                for (int cc = 0; cc < MAXC; ++cc) {
                    up[cc] = UD[i0 + cc].x;                 // (i0, i1 <= NIT: inside the row; cells past the band's items are read and dropped)
                    dn[cc] = UD[i1 + cc].y;
                }
                // (nfft 1024: a segment has <= MAXC items -- the launcher sends other filterbanks to frontend_pk.hip --, so no loop behind
                //  the unconditional reads; nfft 512: the reference filterbank has a few 3-item segments, the loops stay)
                float mel = 0.f;
#pragma unroll
                for (int cc = 0; cc < MAXC; ++cc) mel += i0 + cc < i1 ? up[cc] : 0.f;
                if (NC != 512) for (int it = i0 + MAXC; it < i1; ++it) mel += UD[it].x;
#pragma unroll
                for (int cc = 0; cc < MAXC; ++cc) mel += i1 + cc < i2 ? dn[cc] : 0.f;
                if (NC != 512) for (int it = i1 + MAXC; it < i2; ++it) mel += UD[it].y;
                LM[m * 16 + (c ^ ((m >> 1) & 15))] = fast_log(a.log_floor ? fmaxf(mel, 1e-12f) : mel + 1e-6f);
            }
        }
        // (no sync: the next LDS phase that touches the item sums or the power spectrum sits behind the next round's syncs)
    }
    wave_sync();

    // ---------------- DCT-II on the matrix cores: a wave's own <= 16 frames, all coefficient tiles ----------------
    {
        const int kq = lane >> 4, col = lane & 15;
        const int g = chunk * fpw + wave * fpwv + col;
        const bool valid = col < fpwv && g < a.total_frames;
        const int gg = valid ? g : a.total_frames - 1;
        const int n = gg / a.n_frames;
        const int t = gg - n * a.n_frames;
        float* dst = a.out + (size_t)n * a.n_coef * a.tp + kHalo + t;
        if (a.no_dct) {
            for (int m = kq; m < a.n_coef; m += 4) {
                if (valid) {
                    float* row = dst + (size_t)m * a.tp;
                    row[0] = LM[m * 16 + (col ^ ((m >> 1) & 15))];
                    if (t == 0) { row[-4] = 0.f; row[-3] = 0.f; row[-2] = 0.f; row[-1] = 0.f; }
                    if (t == a.n_frames - 1) { row[1] = 0.f; row[2] = 0.f; row[3] = 0.f; row[4] = 0.f; }
                }
            }
        } else {
            constexpr int CT = NMEL / 16;                                   // coefficient tiles (n_coef <= NMEL)
            const int ntile = (a.n_coef + 15) >> 4;
            const int c0 = col ^ (kq >> 1);                                 // m = 4 st + kq: (m >> 1) & 15 = (2 st & 14) | (kq >> 1)
            float bv[NMEL / 4];
#pragma unroll
            for (int st = 0; st < NMEL / 4; ++st) bv[st] = LM[(4 * st + kq) * 16 + (c0 ^ ((2 * st) & 14))];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (ct < ntile) {                                           // (wave-uniform)
                    float av[NMEL / 4];
#pragma unroll
                    for (int st = 0; st < NMEL / 4; ++st) av[st] = a.dct_tab[(ct * (NMEL / 4) + st) * 64 + lane];
                    v4 acc = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int st = 0; st < NMEL / 4; ++st) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st], bv[st], acc, 0, 0, 0);
                    if (valid) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int c = 16 * ct + 4 * kq + rr;
                            if (c < a.n_coef) {
                                float* row = dst + (size_t)c * a.tp;
                                row[0] = acc[rr];
                                if (t == 0) { row[-4] = 0.f; row[-3] = 0.f; row[-2] = 0.f; row[-1] = 0.f; }
                                if (t == a.n_frames - 1) { row[1] = 0.f; row[2] = 0.f; row[3] = 0.f; row[4] = 0.f; }
                            }
                        }
                    }
                }
            }
        }
    }
    // (no barrier: the next chunk's first write to the log-mel block sits behind that chunk's wave-local syncs)
    }
}

// returns 1 (nothing launched) when the configuration needs another kernel: unaligned frames, a window whose valid radix-16 inputs
// differ from lane to lane, or a filterbank with more items than the unrolled trips take / with a segment of more items than the log
// phase reads (n_items < 0: frontend_mel_item_count)
int launch_frontend_pk3(int nc, const FrontendArgs& a0, int n_items, hipStream_t s) {
    const int sub = nc / 256;
    if (!a0.aligned || (a0.win & 1) || (a0.win / 2) % (16 * sub) != 0) return 1;
    if (a0.total_frames >= (1 << 23)) return 1;                // (the frame -> utterance split uses a float reciprocal)
    if (n_items < 0 || n_items > mel_items_fast(nc)) return 1;
    const int qv = a0.win / (32 * sub);
    // Persistent workgroups (3 per CU) walk chunks of rounds x (4096 / nc) frames: at most 16 frames per wave (one DCT tile).  The cost
    // model of frontend_pk.hip's launcher with three slots per CU (a chunk costs rounds + ~1.45 rounds).
    FrontendArgs a = a0;
    int grid = 0;
    {
        const int fpr = 4096 / nc, max_rounds = 64 / fpr, slots = 3 * device_cus();
        int best = max_rounds;
        float best_cost = 3.4e38f;
        const int r_min = ceil_div(a.total_frames, fpr) <= slots / 2 ? 1 : (max_rounds + 1) / 2;
        for (int r = max_rounds; r >= r_min; --r) {
            const int chunks = ceil_div(a.total_frames, r * fpr);
            const int full = chunks / slots, rest = chunks % slots;
            const float gens = (float)full + (rest == 0 ? 0.f : (3 * rest <= slots ? 0.45f : (3 * rest <= 2 * slots ? 0.75f : 1.f)));
            const float cost = gens * ((float)r + 1.45f);
            if (cost < best_cost * 0.995f) { best_cost = cost; best = r; }
        }
        const int knob = tune_get(TCR_TUNE_FRONTEND);
        if (knob >= 10 && knob < 30) best = min(max(knob - 10, 1), max_rounds);
        if (a0.rounds > 0) best = min(a0.rounds, max_rounds);      // (tcr_frontend_fwd_rounds: the caller's per-call choice)
        a.rounds = best;
        grid = min(ceil_div(a.total_frames, best * fpr), slots);
        const int cap = tune_get(TCR_TUNE_FE_GRID);
        if (cap > 0) grid = min(grid, cap);
        a.stagger = tune_get(TCR_TUNE_FE_STAGGER);
        a.stagger_div = device_cus();
    }
#define TCR_FPK3(NC_, QV_)                                                                                          \
    if (nc == NC_ && qv == QV_) {                                                                                   \
        if (a.magnitude) hipLaunchKernelGGL((frontend_pk3_kernel<NC_, QV_, true>), dim3(grid), dim3(256), 0, s, a); \
        else hipLaunchKernelGGL((frontend_pk3_kernel<NC_, QV_, false>), dim3(grid), dim3(256), 0, s, a);            \
        return check_launch("frontend_pk3_kernel");                                                                 \
    }
    TCR_FPK3(512, 10)   // 40 ms window @ 16 kHz, FFT 1024
    TCR_FPK3(256, 15)   // 30 ms window, FFT 512
    TCR_FPK3(512, 16)
    TCR_FPK3(256, 16)
    TCR_FPK3(256, 10)   // 20 ms window, FFT 512
    TCR_FPK3(512, 15)
#undef TCR_FPK3
    return 1;
}

}  // namespace tcr
