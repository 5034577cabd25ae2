// Host-side construction of the front-end constant tables (float64 math, float32 storage).
//
// Restates the constants TF builds inside the graph for datasets/preprocessors.py:64-96,183-194:
//   hann_window(periodic=True), the rfft twiddles, tf.contrib.signal.linear_to_mel_weight_matrix
//   (HTK mel scale, DC bin zeroed, triangular, un-normalised) and the DCT-II of
//   mfccs_from_log_mel_spectrograms (x 1/sqrt(2N)).
// The mel matrix is stored in its sparse form: every spectrogram bin lies in exactly one
// mel-edge segment j and contributes to at most two filters (up-slope of j, down-slope of j-1).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#include "frontend_plan.h"

namespace tcr {
namespace {

constexpr double kPi = 3.14159265358979323846;

double hertz_to_mel(double f) { return 1127.0 * std::log1p(f / 700.0); }

// numpy.linspace(start, stop, num)[i]
double linspace_at(double start, double stop, int num, int i) {
    if (i == num - 1) return stop;
    double step = (stop - start) / (double)(num - 1);
    return (double)i * step + start;
}

// Dense mel matrix exactly as TF builds it: [nbins][n_mel], row 0 (DC) zero.
std::vector<double> dense_mel(const tcr_frontend_cfg& c, int nbins) {
    const int nm = c.n_mel;
    std::vector<double> m((size_t)nbins * nm, 0.0);
    const double nyquist = c.sample_rate / 2.0;
    std::vector<double> edges(nm + 2);
    const double mlo = hertz_to_mel(c.lower_hz), mhi = hertz_to_mel(c.upper_hz);
    for (int i = 0; i < nm + 2; ++i) edges[i] = linspace_at(mlo, mhi, nm + 2, i);
    for (int k = 1; k < nbins; ++k) {
        const double mel = hertz_to_mel(linspace_at(0.0, nyquist, nbins, k));
        for (int j = 0; j < nm; ++j) {
            const double lower = edges[j], center = edges[j + 1], upper = edges[j + 2];
            const double ls = (mel - lower) / (center - lower);
            const double us = (upper - mel) / (upper - center);
            const double v = std::fmax(0.0, std::fmin(ls, us));
            m[(size_t)k * nm + j] = v;
        }
    }
    return m;
}

}  // namespace
}  // namespace tcr

using namespace tcr;

extern "C" int tcr_frontend_resolve(tcr_frontend_cfg* cfg) {
    TCR_REQUIRE(cfg != nullptr, "tcr_frontend_resolve: null cfg");
    TCR_REQUIRE(cfg->sample_rate > 0 && cfg->n_samples > 0, "front-end: sample_rate/n_samples must be positive");
    TCR_REQUIRE(cfg->win > 0 && cfg->hop > 0 && cfg->win <= cfg->n_samples, "front-end: bad window (%d) / stride (%d)", cfg->win, cfg->hop);
    int nfft = 1;
    while (nfft < cfg->win) nfft <<= 1;
    TCR_REQUIRE(nfft == 512 || nfft == 1024,
                "front-end: fft_length %d (window %d samples) unsupported; this build has the 512- and 1024-point kernels", nfft, cfg->win);
    TCR_REQUIRE((cfg->win & 1) == 0, "front-end: window_size_samples must be even (got %d)", cfg->win);
    TCR_REQUIRE(cfg->n_mel == 64, "front-end: num_mel_bins must be 64 (got %d)", cfg->n_mel);
    TCR_REQUIRE(cfg->method >= 0 && cfg->method <= 2, "front-end: method must be 0 (mfcc), 1 (log_mel_spectrogram) or 2 (deploy-path mfcc)");
    if (cfg->method == 1) cfg->n_coef = cfg->n_mel;
    TCR_REQUIRE(cfg->n_coef >= 1 && cfg->n_coef <= cfg->n_mel, "front-end: num_mfccs must be in [1, %d] (got %d)", cfg->n_mel, cfg->n_coef);
    TCR_REQUIRE(cfg->lower_hz >= 0.f && cfg->lower_hz < cfg->upper_hz && cfg->upper_hz <= cfg->sample_rate / 2.0f,
                "front-end: mel edges [%g, %g] Hz invalid for sample rate %d", cfg->lower_hz, cfg->upper_hz, cfg->sample_rate);
    cfg->nfft = nfft;
    cfg->n_frames = 1 + (cfg->n_samples - cfg->win) / cfg->hop;
    return TCR_OK;
}

extern "C" size_t tcr_frontend_plan_bytes(const tcr_frontend_cfg* cfg) {
    if (!cfg || cfg->nfft <= 0) return 0;
    return frontend_plan_layout(*cfg).words * 4;
}

extern "C" int tcr_frontend_plan_init(const tcr_frontend_cfg* cfg, void* host_plan) {
    TCR_REQUIRE(cfg && host_plan, "tcr_frontend_plan_init: null argument");
    TCR_REQUIRE(cfg->nfft == 512 || cfg->nfft == 1024, "tcr_frontend_plan_init: call tcr_frontend_resolve first");
    const FrontendPlanLayout L = frontend_plan_layout(*cfg);
    float* w = static_cast<float*>(host_plan);
    std::memset(w, 0, L.words * 4);

    for (int i = 0; i < cfg->win; ++i) w[L.window + i] = (float)(0.5 - 0.5 * std::cos(2.0 * kPi * i / cfg->win));
    for (int i = 0; i < cfg->win; ++i) {                    // (frontend_pk3.hip: lanes 8..15 of a 256-point unit transform x[n] (-1)^n)
        const int np = i / (cfg->nfft / 256);               // index within the frame's even / odd (or only) complex sequence: l + 16 q
        const bool neg = ((np >> 3) & 1) && ((np >> 4) & 1);
        w[L.window_sgn + i] = neg ? -w[L.window + i] : w[L.window + i];
    }
    for (int n1 = 0; n1 < 16; ++n1)
        for (int k2 = 0; k2 < 16; ++k2) {
            const double a = -2.0 * kPi * (double)(n1 * k2) / 256.0;
            w[L.tw256 + 2 * (n1 * 16 + k2)] = (float)std::cos(a);
            w[L.tw256 + 2 * (n1 * 16 + k2) + 1] = (float)std::sin(a);
        }
    for (int k = 0; k < 256; ++k) {
        const double a = -2.0 * kPi * (double)k / 512.0;
        w[L.tw_combine + 2 * k] = (float)std::cos(a);
        w[L.tw_combine + 2 * k + 1] = (float)std::sin(a);
    }
    for (int k = 0; k <= L.nc / 2; ++k) {
        const double a = -2.0 * kPi * (double)k / (double)cfg->nfft;
        w[L.tw_real + 2 * k] = (float)std::cos(a);
        w[L.tw_real + 2 * k + 1] = (float)std::sin(a);
    }

    const int nm = cfg->n_mel;
    if (cfg->method == 2) {
        // Deploy path: the filterbank of the contrib_audio.mfcc op (TF core/kernels/mfcc_mel_filterbank.cc, restated):
        // centre frequencies equally spaced on the mel scale, bins [start_index, end_index], bin i with band b = band_mapper[i]
        // adds weight[i] * |X| to channel b and (1 - weight[i]) * |X| to channel b + 1.  In the kernel's segment form bin i
        // lies in segment b + 1 with up-slope (-> filter b + 1) 1 - weight and down-slope (-> filter b) weight.
        const int nbins = L.nbins;
        const double mel_low = hertz_to_mel(cfg->lower_hz), mel_hi = hertz_to_mel(cfg->upper_hz);
        const double mel_spacing = (mel_hi - mel_low) / (double)(nm + 1);
        std::vector<double> center(nm + 1);
        for (int i = 0; i < nm + 1; ++i) center[i] = mel_low + mel_spacing * (i + 1);
        const double hz_per_sbin = 0.5 * cfg->sample_rate / (double)(nbins - 1);
        const int start_index = (int)(1.5 + cfg->lower_hz / hz_per_sbin);
        const int end_index = (int)(cfg->upper_hz / hz_per_sbin);
        int32_t* segd = reinterpret_cast<int32_t*>(w + L.seg_start);
        std::vector<int> band(nbins, -2);
        int channel = 0;
        for (int i = 0; i < nbins; ++i) {
            const double melf = hertz_to_mel(i * hz_per_sbin);
            if (i < start_index || i > end_index) continue;
            while (center[channel] < melf && channel < nm) ++channel;
            band[i] = channel - 1;
        }
        for (int j = 0; j <= L.nseg; ++j) {
            int first = end_index + 1;
            for (int i = end_index; i >= start_index; --i)
                if (band[i] + 1 >= j) first = i;
            segd[j] = j == 0 ? start_index : first;
        }
        for (int i = start_index; i <= end_index && i < nbins; ++i) {
            const double melf = hertz_to_mel(i * hz_per_sbin);
            const int b = band[i];
            const double wgt = b >= 0 ? (center[b + 1] - melf) / (center[b + 1] - center[b]) : (center[0] - melf) / (center[0] - mel_low);
            w[L.wud + 2 * i] = (float)(1.0 - wgt);
            w[L.wud + 2 * i + 1] = (float)wgt;
        }
    } else {
    // Sparse mel: segment boundaries in bins + two slopes per bin, extracted from the dense matrix.
    const std::vector<double> M = dense_mel(*cfg, L.nbins);
    const double nyquist = cfg->sample_rate / 2.0;
    const double mlo = hertz_to_mel(cfg->lower_hz), mhi = hertz_to_mel(cfg->upper_hz);
    int32_t* seg = reinterpret_cast<int32_t*>(w + L.seg_start);
    // seg_of[k] = j when edges[j] <= mel(k) < edges[j+1]; -1 below the first edge, nm+1 at/after the
    // last one (both weightless).  mel(k) is increasing, so segments are contiguous bin ranges and
    // seg[j] = first bin whose segment index is >= j.
    std::vector<int> seg_of(L.nbins, -1);
    for (int k = 1; k < L.nbins; ++k) {
        const double mel = hertz_to_mel(linspace_at(0.0, nyquist, L.nbins, k));
        int j = -1;
        for (int e = 0; e < nm + 2; ++e)
            if (mel >= linspace_at(mlo, mhi, nm + 2, e)) j = e;
        seg_of[k] = j;
    }
    for (int j = 0; j <= L.nseg; ++j) {
        int first = L.nbins;
        for (int k = L.nbins - 1; k >= 1; --k)
            if (seg_of[k] >= j) first = k;
        seg[j] = first;
    }
    for (int k = 1; k < L.nbins; ++k) {
        const int j = seg_of[k];
        float up = 0.f, down = 0.f;
        if (j >= 0 && j < nm) up = (float)M[(size_t)k * nm + j];
        if (j >= 1 && j <= nm) down = (float)M[(size_t)k * nm + j - 1];
        w[L.wud + 2 * k] = up;
        w[L.wud + 2 * k + 1] = down;
        for (int m = 0; m < nm; ++m) {       // every other entry of the row must be exactly zero
            if (m == j || m == j - 1) continue;
            if (M[(size_t)k * nm + m] != 0.0) {
                set_error("front-end: mel matrix row %d has an unexpected non-zero at filter %d (segment %d)", k, m, j);
                return TCR_ERR_ARG;
            }
        }
    }
    }

    // DCT-II rows, folded: dcth[c][n] = 2 cos(pi c (2n+1) / (2 N)) / sqrt(2 N), n < N/2
    // (the n' = N-1-n half is (-1)^c times the same value).
    const int half = nm / 2;
    for (int c = 0; c < nm; ++c)
        for (int n = 0; n < half; ++n)
            w[L.dcth + (size_t)c * half + n] = (float)(2.0 * std::cos(kPi * c * (2.0 * n + 1.0) / (2.0 * nm)) / std::sqrt(2.0 * nm));

    // The segments cut into items of <= mel_item_bins() bins (frontend_pk.hip: one lane per item)
    {
        const int kMelItemBins = mel_item_bins(L.nc);
        const int32_t* seg = reinterpret_cast<const int32_t*>(w + L.seg_start);
        int32_t* items = reinterpret_cast<int32_t*>(w + L.mel_items);
        int32_t* ifirst = reinterpret_cast<int32_t*>(w + L.mel_ifirst);
        int n = 0;
        for (int j = 0; j < L.nseg; ++j) {
            ifirst[j] = n;
            for (int k = seg[j]; k < seg[j + 1]; k += kMelItemBins) {
                TCR_REQUIRE(n < kMelItemsMax, "front-end: mel filterbank needs more than %d work items", kMelItemsMax);
                items[n++] = k | (std::min(kMelItemBins, seg[j + 1] - k) << 10) | (j << 14);
            }
        }
        ifirst[L.nseg] = n; ifirst[L.nseg + 1] = n;
        // PHYSICAL order and read bases of the items the unrolled trips take (slot = lane + lanes-per-frame * trip).  In logical
        // (ascending-bin) order the 32 lanes of a ds_read_b32 group start their item 2-8 bins apart, i.e. three to five of them on every
        // LDS bank (banks are mod 32 for these reads): the sparse-mel reads were 3-way conflicted (31 % of the kernel's LDS cycles,
        // profiles/r03_final_pmc.csv).  Here a trip holds at most one item per start class (read base mod 32; mod 16 at 16 lanes per
        // frame, where a lane group holds two frames whose rows sit 16 banks apart), so its 8 (4) reads -- constant offsets from the
        // base -- are conflict-free.  The freedom that makes this a perfect matching for the reference filterbanks (91 / 91 and 89 / 89
        // items): an item shorter than the trip's 8 (4) bins may start its reads up to (8 - bins) bins EARLY, the leading bins meeting
        // zero slopes.  Maximum bipartite matching items -> (class, trip) by augmenting paths; what stays unmatched (other filterbanks)
        // takes a free slot at its natural base.  A slot's descriptor carries the item's LOGICAL index (bits 21+): the item sums are
        // stored, and read by the log phase, in logical order.
        const int nfast = mel_items_fast(L.nc), lpf = L.nc / 16, trips = mel_trips(L.nc);
        const int classes = lpf >= 32 ? 32 : 16;
        const int nf = std::min(n, nfast);
        std::vector<int32_t> logical(items, items + n);
        std::vector<int> owner(classes * trips, -1);            // (class, trip) -> item
        std::vector<int> slot_of(nf, -1);                       // item -> class * trips + trip
        auto allowed = [&](int i, int c) {                      // can item i read from a base of class c?  (-1: no; else the base)
            const int k0 = logical[i] & 1023, nb = (logical[i] >> 10) & 15;
            for (int d = 0; d <= kMelItemBins - nb && k0 - d >= 0; ++d)
                if ((k0 - d) % classes == c) return k0 - d;
            return -1;
        };
        std::vector<char> seen;
        std::function<bool(int)> place = [&](int i) -> bool {
            for (int c = 0; c < classes; ++c) {
                if (allowed(i, c) < 0) continue;
                for (int t = 0; t < trips; ++t) {
                    const int sidx = c * trips + t;
                    if (seen[sidx]) continue;
                    seen[sidx] = 1;
                    if (owner[sidx] < 0 || place(owner[sidx])) { owner[sidx] = i; slot_of[i] = sidx; return true; }
                }
            }
            return false;
        };
        for (int i = 0; i < nf; ++i) { seen.assign(classes * trips, 0); (void)place(i); }
        // lanes of a trip: the 16-lane halves of the sum store (ds_write_b64: 16-lane groups) prefer distinct logical indices mod 16
        std::vector<int> phys(nfast, -1), base_of(nf, 0);
        std::vector<int> fill(trips, 0);
        std::vector<std::vector<int>> trip_items(trips);
        for (int i = 0; i < nf; ++i)
            if (slot_of[i] >= 0 && owner[slot_of[i]] == i) { trip_items[slot_of[i] % trips].push_back(i); base_of[i] = allowed(i, slot_of[i] / trips); }
        std::vector<int> leftovers;
        for (int i = 0; i < nf; ++i)
            if (!(slot_of[i] >= 0 && owner[slot_of[i]] == i)) { leftovers.push_back(i); base_of[i] = logical[i] & 1023; }
        for (int t = 0; t < trips; ++t) {
            std::vector<int>& v = trip_items[t];
            while ((int)v.size() < lpf && !leftovers.empty()) { v.push_back(leftovers.back()); leftovers.pop_back(); }
            TCR_REQUIRE((int)v.size() <= lpf, "front-end: mel item matching overfilled a trip");
            std::vector<int> lanes(lpf, -1);
            std::vector<char> used16(lpf, 0);                   // [half * 16 + residue]
            std::vector<int> later;
            for (int i : v) {                                   // first pass: a half whose residue class is still free
                bool done = false;
                for (int h = 0; h < lpf / 16 && !done; ++h) {
                    if (used16[h * 16 + i % 16]) continue;
                    for (int l = h * 16; l < h * 16 + 16; ++l)
                        if (lanes[l] < 0) { lanes[l] = i; used16[h * 16 + i % 16] = 1; done = true; break; }
                }
                if (!done) later.push_back(i);
            }
            for (int i : later)
                for (int l = 0; l < lpf; ++l)
                    if (lanes[l] < 0) { lanes[l] = i; break; }
            for (int l = 0; l < lpf; ++l) phys[t * lpf + l] = lanes[l];
        }
        TCR_REQUIRE(leftovers.empty(), "front-end: mel item slots exhausted");
        const int kDummyItem = mel_dummy_item(L.nc, n);    // logical index of an empty slot: a cell of the item-sum rows nobody reads
        static_assert(127 < (1 << 7) && kMelItemsMax < 256, "descriptor fields: segment 7 bits (14..20), logical index 8 bits (21..28)");
        TCR_REQUIRE(L.nseg <= 127, "front-end: %d mel-edge segments do not fit the item descriptor's 7-bit field", L.nseg);
        for (int slot = 0; slot < nfast; ++slot) {
            const int i = phys[slot];
            if (i < 0) { items[slot] = kDummyItem << 21; continue; }
            const int k0 = logical[i] & 1023, nb = (logical[i] >> 10) & 15, kb = base_of[i];
            items[slot] = kb | (nb << 10) | (logical[i] & (127 << 14)) | (i << 21);        // (bin field = the READ BASE, <= the item's first bin)
            for (int b = 0; b < kMelItemBins; ++b) {
                const bool in = kb + b >= k0 && kb + b < k0 + nb;
                w[L.mel_wit + 2 * ((size_t)b * nfast + slot)] = in ? w[L.wud + 2 * (kb + b)] : 0.f;
                w[L.mel_wit + 2 * ((size_t)b * nfast + slot) + 1] = in ? w[L.wud + 2 * (kb + b) + 1] : 0.f;
            }
        }
        for (int i = nfast; i < n; ++i) items[i] = (logical[i] & ((1 << 21) - 1)) | (i << 21);      // (slow path: logical == physical, true first bin)
    }
    // DCT-II as MFMA A fragments (coefficient tile ct, k-step s over the mel index)
    for (int ct = 0; ct < nm / 16; ++ct)
        for (int s = 0; s < nm / 4; ++s)
            for (int l = 0; l < 64; ++l) {
                const int c = 16 * ct + (l & 15), n = 4 * s + (l >> 4);
                float v = 0.f;
                if (c < cfg->n_coef) {
                    v = w[L.dcth + (size_t)c * half + (n < half ? n : nm - 1 - n)];
                    if (n >= half && (c & 1)) v = -v;
                }
                w[L.dct_tab + ((size_t)ct * (nm / 4) + s) * 64 + l] = v;
            }
    return TCR_OK;
}

int tcr::frontend_mel_item_count(const tcr_frontend_cfg& c) {
    // (-1 as well when, at nfft 1024, a segment has more than the three items the three-waves kernel's log phase reads)
    struct Entry { tcr_frontend_cfg cfg; int n; };
    static std::mutex mu;
    static std::vector<Entry> cache;
    std::lock_guard<std::mutex> lock(mu);
    for (const Entry& e : cache)
        if (std::memcmp(&e.cfg, &c, sizeof(c)) == 0) return e.n;
    int n = -1;
    if (c.nfft == 512 || c.nfft == 1024) {
        const FrontendPlanLayout L = frontend_plan_layout(c);
        std::vector<float> plan(L.words);
        if (tcr_frontend_plan_init(&c, plan.data()) == TCR_OK) {
            const int32_t* ifirst = reinterpret_cast<const int32_t*>(plan.data() + L.mel_ifirst);
            n = ifirst[L.nseg];
            if (L.nc == 512)
                for (int j = 0; j < L.nseg; ++j)
                    if (ifirst[j + 1] - ifirst[j] > 3) n = -1;
        }
    }
    if (cache.size() >= 64) cache.clear();
    cache.push_back(Entry{c, n});
    return n;
}

extern "C" int tcr_frontend_plan_mel_matrix(const tcr_frontend_cfg* cfg, const void* host_plan, float* out) {
    TCR_REQUIRE(cfg && host_plan && out, "tcr_frontend_plan_mel_matrix: null argument");
    const FrontendPlanLayout L = frontend_plan_layout(*cfg);
    const float* w = static_cast<const float*>(host_plan);
    const int32_t* seg = reinterpret_cast<const int32_t*>(w + L.seg_start);
    const int nm = cfg->n_mel;
    std::memset(out, 0, sizeof(float) * (size_t)L.nbins * nm);
    for (int j = 0; j < L.nseg; ++j)
        for (int k = seg[j]; k < seg[j + 1]; ++k) {
            if (j < nm) out[(size_t)k * nm + j] = w[L.wud + 2 * k];
            if (j >= 1) out[(size_t)k * nm + j - 1] = w[L.wud + 2 * k + 1];
        }
    return TCR_OK;
}

extern "C" int tcr_frontend_plan_dct_matrix(const tcr_frontend_cfg* cfg, const void* host_plan, float* out) {
    TCR_REQUIRE(cfg && host_plan && out, "tcr_frontend_plan_dct_matrix: null argument");
    const FrontendPlanLayout L = frontend_plan_layout(*cfg);
    const float* w = static_cast<const float*>(host_plan);
    const int nm = cfg->n_mel, half = nm / 2;
    for (int n = 0; n < nm; ++n)
        for (int c = 0; c < cfg->n_coef; ++c) {
            const float v = w[L.dcth + (size_t)c * half + (n < half ? n : nm - 1 - n)];
            out[(size_t)n * cfg->n_coef + c] = (n < half || (c & 1) == 0) ? v : -v;
        }
    return TCR_OK;
}
