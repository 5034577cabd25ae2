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
// HBM-bound byte work: 2 B (+2 B when mixing) in, 4 B out per sample; one thread per 4 consecutive samples of an
// utterance (16-byte store), 2-byte loads because shift and pool offsets are arbitrary.
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

__global__ __launch_bounds__(256) void augment_kernel(const AugmentArgs a) {
    const int64_t total = (int64_t)a.batch * a.quads;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int b = (int)(idx / a.quads);
        const int i0 = (int)(idx - (int64_t)b * a.quads) * 4;
        const int len = min(a.clip_len ? a.clip_len[b] : a.desired, a.desired);
        const int sh = a.shift ? a.shift[b] : 0;
        const float vol = a.bg_vol ? a.bg_vol[b] : 0.f;
        const int16_t* src = a.pcm + a.clip_off[b];
        const int16_t* bsrc = vol != 0.f ? a.bg + a.bg_off[b] : nullptr;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + k;
            const int j = i - sh;
            float fg = 0.f;
            if (i < a.desired && j >= 0 && j < len) fg = (float)src[j] * (1.0f / 32768.0f);
            float o = fg;
            if (bsrc && i < a.desired) {
                const float bgv = (float)bsrc[i] * (1.0f / 32768.0f);
                o = __fadd_rn(__fmul_rn(bgv, vol), fg);
            }
            v[k] = fminf(fmaxf(o, -1.0f), 1.0f);
        }
        float* dst = a.out + (size_t)b * a.desired + i0;
        if ((a.desired & 3) == 0) {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k < a.desired) dst[k] = v[k];
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
    TCR_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "tcr_augment_fwd: out must be 16-byte aligned");
    AugmentArgs a;
    a.pcm = pcm; a.clip_off = clip_off; a.clip_len = clip_len; a.shift = shift; a.bg = background; a.bg_off = bg_off; a.bg_vol = bg_vol;
    a.out = out; a.batch = batch; a.desired = desired_samples; a.quads = ceil_div(desired_samples, 4);
    int64_t blocks = ceil_div64((int64_t)batch * a.quads, 256);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("augment_kernel");
}
