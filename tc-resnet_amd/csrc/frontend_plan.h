// Layout of the front-end constant tables ("plan").  A pure function of the configuration, so
// host (table builder) and device launcher agree without a header inside the device buffer.
#pragma once
#include "tcr_common.h"

namespace tcr {

struct FrontendPlanLayout {
    int nc;        // complex FFT length = nfft / 2 (256 or 512)
    int nbins;     // nfft / 2 + 1
    int nseg;      // n_mel + 1 mel-edge segments
    // offsets in 4-byte words from the start of the plan
    size_t window;     // float  [win]
    size_t tw256;      // float2 [16][16]        W_256^(n1*k2)       (radix-16 x radix-16 inter-stage twiddle)
    size_t tw_combine; // float2 [256]           W_512^k             (only nc == 512: even/odd recombination)
    size_t tw_real;    // float2 [nc/2 + 1]      W_nfft^k            (real-FFT post-processing)
    size_t seg_start;  // int32  [nseg + 1]      first spectrogram bin of each mel-edge segment
    size_t wud;        // float2 [nbins]         (up-slope weight into filter j, down-slope weight into filter j-1)
    size_t dcth;       // float  [n_coef][n_mel/2]   DCT-II rows folded by the even/odd symmetry
    // Sparse mel for the packed kernel, load-balanced: every mel-edge segment is cut into items of <= mel_item_bins(nc) bins, one
    // lane per item (a segment's bins used to be one lane's loop: 2 .. 20 bins, the wave waited for the longest).
    size_t mel_items;  // int32  [kMelItemsMax]   first bin | bins << 10 | segment << 14 | logical item index << 21; slots < mel_items_fast() in the
                       //        bank-conflict-free PHYSICAL order of the unrolled trips (frontend_plan.cpp), the rest in logical order
    size_t mel_ifirst; // int32  [nseg + 2]       first item of segment j; [nseg] = [nseg + 1] = number of items
    size_t mel_wit;    // float2 [mel_item_bins][mel_items_fast]  (up, down) slopes of bin b of item i, zero past the item's end and for
                       //                        items the filterbank does not have: the kernel's LDS copy (bin-major: a trip's lanes read
                       //                        consecutive items -> consecutive addresses), pre-scaled like `wud` is NOT (the kernel folds)
    size_t window_sgn; // float  [win]           the window with the sign (-1)^q of frontend_pk3.hip's lanes 8..15: sample i of a frame is
                       //                        radix-16 input q = n' >> 4 of lane l = n' & 15, n' = i / (nfft / 256); negated where l >= 8 and q is odd
    size_t dct_tab;    // float  [n_mel/16][n_mel/4][64]  DCT A fragments: tile ct, step s, lane l -> D[16 ct + (l & 15)][4 s + (l >> 4)] (0 past n_coef)
    size_t words;      // total size in words
};

constexpr int kMelItemsMax = 192;
// bins per item: segments are 2 .. 20 bins long at nfft 1024, 1 .. 10 at nfft 512
constexpr int mel_item_bins(int nc) { return nc == 512 ? 8 : 4; }
// items the packed kernel handles with its unrolled trips (trips x lanes per frame); a filterbank with more items runs the rest in a slow loop
constexpr int mel_trips(int nc) { return nc == 512 ? 3 : 6; }      // (the reference filterbank: 91 items of 8 bins at nfft 1024, 89 of 4 at nfft 512)
constexpr int mel_items_fast(int nc) { return mel_trips(nc) * (nc / 16); }
// Logical index the plan gives the EMPTY slots of the unrolled trips: a cell of the item-sum row no band ever reads.  A filterbank
// whose items all fit the unrolled trips (n <= mel_items_fast) puts it right behind them, so the three-waves kernel's compact item-sum
// rows (frontend_pk3.hip) hold it; a larger one (slow path, frontend_pk.hip only) behind the last possible item.
constexpr int mel_dummy_item(int nc, int n_items) { return n_items <= mel_items_fast(nc) ? mel_items_fast(nc) : kMelItemsMax; }
static_assert(kMelItemsMax < 255 && mel_items_fast(512) <= kMelItemsMax && mel_items_fast(256) <= kMelItemsMax,
              "an item descriptor carries the logical index in 8 bits (bits 21..28), the empty slots' cell included");

// Number of work items of the configuration's filterbank (what tcr_frontend_plan_init writes to mel_ifirst[nseg]); cached per
// configuration.  -1: the configuration does not resolve.
int frontend_mel_item_count(const tcr_frontend_cfg& c);

inline FrontendPlanLayout frontend_plan_layout(const tcr_frontend_cfg& c) {
    FrontendPlanLayout l{};
    l.nc = c.nfft / 2;
    l.nbins = c.nfft / 2 + 1;
    l.nseg = c.n_mel + 1;
    size_t o = 0;
    auto take = [&](size_t words) { size_t at = o; o += (words + 15) / 16 * 16; return at; };
    l.window = take((size_t)c.win);
    l.tw256 = take(2 * 256);
    l.tw_combine = take(2 * 256);
    l.tw_real = take(2 * (size_t)(l.nc / 2 + 1));
    l.seg_start = take((size_t)l.nseg + 1);
    l.wud = take(2 * (size_t)l.nbins);
    l.dcth = take((size_t)c.n_mel * (size_t)(c.n_mel / 2));   // sized for n_coef == n_mel
    l.mel_items = take(kMelItemsMax);
    l.mel_ifirst = take((size_t)l.nseg + 2);
    l.mel_wit = take(2 * (size_t)mel_item_bins(l.nc) * (size_t)mel_items_fast(l.nc));
    l.window_sgn = take((size_t)c.win);
    l.dct_tab = take((size_t)(c.n_mel / 16) * (size_t)(c.n_mel / 4) * 64);
    l.words = o;
    return l;
}

}  // namespace tcr
