// Input stage ahead of the MFCC front-end, on the device: 16-bit PCM -> float, crop / zero-pad to the clip length,
// random time shift with zero fill, background-noise mix, clip to [-1, 1].
//
// Replaces, per utterance, contrib_audio.decode_wav(desired_samples) + _shift_audio + _mix_background of
// datasets/augmentation_factory.py:30-211 (anchored_slice_or_pad / anchored_slice_or_pad_with_shift /
// no_augmentation_audio) as mapped over the batch by AudioDataWrapper._parse_function
// (datasets/audio_data_wrapper.py:37-58).  The reference draws its random numbers inside the TF graph; here the draws
// (shift, background window, background volume) are INPUTS, so the arithmetic is a pure function of them:
//   fg[i] = pcm[clip_off + i - shift] / 32768   if 0 <= i - shift < min(clip_len, desired)   else 0     (silent: clip_len = 0)
//   out[i] = clip(bg[bg_off + i] / 32768 * bg_vol + fg[i], -1, 1)                                        (no mix: bg_vol = 0)
// with one IEEE multiply and one IEEE add (no contraction), exactly as tf.multiply / tf.add / tf.clip_by_value do.
//
// HBM-bound byte work: 2 B (+2 B when mixing) in, 4 B out per sample; 2-byte loads because shift and pool offsets are
// arbitrary.
#include "tcr_common.h"

namespace tcr {

struct AugmentArgs {
    const int16_t* pcm;
    const int64_t* clip_off;
    const int32_t* clip_len;
    const int32_t* shift;
    const int16_t* bg;
    const int64_t* bg_off;
    const float* bg_vol;
    float* out;
    int batch, desired, quads;      // quads = ceil(desired / 4)
};

// grid = (ceil(desired / 2048), batch): a workgroup covers 2048 consecutive samples of ONE utterance, so the utterance's
// parameters are wave-uniform (scalar loads) and there is no index division; a thread produces two float4 of output
// from 8 + 8 independent 2-byte loads.  (Measured alternatives at batch 4096: one flat grid-stride loop with an index
// division 2.6 TB/s; lane-contiguous 2-byte loads with 4-byte stores 2.8 TB/s; this form 3.3 TB/s.)
__global__ __launch_bounds__(256) void augment_kernel(const AugmentArgs a) {
    const int b = blockIdx.y;
    const int len = min(a.clip_len ? a.clip_len[b] : a.desired, a.desired);
    const int sh = a.shift ? a.shift[b] : 0;
    const float vol = a.bg_vol ? a.bg_vol[b] : 0.f;
    const int16_t* src = a.pcm + a.clip_off[b];
    const int16_t* bsrc = vol != 0.f ? a.bg + a.bg_off[b] : nullptr;
    float* row = a.out + (size_t)b * a.desired;
    const bool vec = (a.desired & 3) == 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i0 = blockIdx.x * 2048 + h * 1024 + threadIdx.x * 4;
        if (i0 >= a.desired) continue;
        float fg[4], bgv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k, j = i - sh;
            fg[k] = (i < a.desired && j >= 0 && j < len) ? (float)src[j] * (1.0f / 32768.0f) : 0.f;
        }
        if (bsrc) {
#pragma unroll
            for (int k = 0; k < 4; ++k) bgv[k] = (i0 + k < a.desired) ? (float)bsrc[i0 + k] * (1.0f / 32768.0f) : 0.f;
        }
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float o = fg[k];
            if (bsrc) {
#pragma clang fp contract(off)      // tf.multiply then tf.add: two roundings, never an fma
                const float scaled = bgv[k] * vol;
                o = scaled + fg[k];
            }
            v[k] = fminf(fmaxf(o, -1.0f), 1.0f);
        }
        if (vec) {
            *reinterpret_cast<float4*>(row + i0) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k < a.desired) row[i0 + k] = v[k];
        }
    }
}

}  // namespace tcr

using namespace tcr;

extern "C" int tcr_augment_fwd(const int16_t* pcm, const int64_t* clip_off, const int32_t* clip_len, const int32_t* shift,
                               const int16_t* background, const int64_t* bg_off, const float* bg_vol, int batch,
                               int desired_samples, float* out, void* stream) {
    TCR_REQUIRE(pcm && clip_off && out, "tcr_augment_fwd: null argument");
    TCR_REQUIRE(batch > 0 && desired_samples > 0, "tcr_augment_fwd: batch %d / desired_samples %d must be positive", batch, desired_samples);
    TCR_REQUIRE(!bg_vol || (background && bg_off), "tcr_augment_fwd: bg_vol given without a background pool / offsets");
    AugmentArgs a;
    a.pcm = pcm; a.clip_off = clip_off; a.clip_len = clip_len; a.shift = shift; a.bg = background; a.bg_off = bg_off; a.bg_vol = bg_vol;
    a.out = out; a.batch = batch; a.desired = desired_samples; a.quads = ceil_div(desired_samples, 4);
    TCR_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "tcr_augment_fwd: out must be 16-byte aligned");
    TCR_REQUIRE(batch <= 65535, "tcr_augment_fwd: batch %d exceeds the 65535-utterance launch limit", batch);
    hipLaunchKernelGGL(augment_kernel, dim3(ceil_div(desired_samples, 2048), batch), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("augment_kernel");
}
