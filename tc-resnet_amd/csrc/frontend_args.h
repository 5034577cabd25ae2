// Argument block of the fused front-end kernels (frontend.hip: scalar FP32; frontend_pk.hip: packed FP32).
#pragma once
#include "tcr_common.h"

namespace tcr {

struct FrontendArgs {
    const float* wav;
    float* out;
    const float* window;
    const float* window_sgn;    // frontend_pk3.hip (frontend_plan.h)
    const float2* tw256;
    const float2* tw_combine;
    const float2* tw_real;
    const int* seg_start;
    const float2* wud;
    const float* dcth;
    const int* mel_items;   // load-balanced sparse mel (frontend_plan.h)
    const int* mel_ifirst;
    const float2* mel_wit;
    const float* dct_tab;
    int n_samples, win, hop, n_frames, n_coef, tp;
    int total_frames;
    int magnitude;      // 1: the mel filterbank takes |S| instead of |S|^2
    int no_dct;         // 1: log-mel preprocessor (output = log-mel, no DCT)
    int log_floor;      // 0: log(x + 1e-6) (tf.contrib.signal path); 1: log(max(x, 1e-12)) (contrib_audio.mfcc op)
    int aligned;        // frame starts are 8-byte aligned -> float2 loads
    int stagger, stagger_div;   // frontend_pk3.hip: start-up delay unit (x 64 cycles) per (workgroup generation, wave), workgroups per generation
    int rounds;         // packed kernel: rounds of (4096 / nc) frames per workgroup (set by launch_frontend_pk)
};

// packed-FP32 kernel; nc = nfft / 2 (256 or 512); returns 1 when the general kernel must be used
int launch_frontend_pk(int nc, const FrontendArgs& a, hipStream_t s);
// the same at three waves per SIMD (frontend_pk3.hip); n_items: frontend_mel_item_count() of the configuration
int launch_frontend_pk3(int nc, const FrontendArgs& a, int n_items, hipStream_t s);

}  // namespace tcr
