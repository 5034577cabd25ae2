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
    size_t words;      // total size in words
};

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
    l.words = o;
    return l;
}

}  // namespace tcr
