"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's audio input stage
(datasets/augmentation_factory.py:30-211 as mapped per element by datasets/audio_data_wrapper.py:37-58).

PARITY UNPINNED (see oracle/__init__.py): the reference runs these ops inside a TF graph with TF's own random
streams; here every random draw is an explicit argument, so the functions are deterministic and an implementation can
be compared BIT-EXACTLY (all arithmetic is float32, one multiply and one add per sample).
"""
from __future__ import annotations

import numpy as np


def decode_wav(pcm16: np.ndarray, desired_samples: int) -> np.ndarray:
    """contrib_audio.decode_wav(desired_channels=1, desired_samples=N) (:143-155): int16 -> float32 / 32768, cropped or
    zero-padded to N samples.  pcm16: 1-D int16 (empty for the '' filename of a silent sample, :177-183)."""
    out = np.zeros(desired_samples, np.float32)
    n = min(len(pcm16), desired_samples)
    out[:n] = pcm16[:n].astype(np.float32) * np.float32(1.0 / 32768.0)
    return out


def shift_audio(audio: np.ndarray, time_shift_amount: int) -> np.ndarray:
    """_shift_audio (:104-141): tf.pad at the front (amount >= 0) or the back (amount < 0), then slice desired_samples."""
    n = len(audio)
    out = np.zeros_like(audio)
    s = int(time_shift_amount)
    if s >= 0:
        out[s:] = audio[:n - s] if s < n else 0
    else:
        out[:n + s] = audio[-s:] if -s < n else 0
    return out


def mix_background(foreground: np.ndarray, background_crop: np.ndarray, background_volume: float) -> np.ndarray:
    """_mix_background (:92-97): clip_by_value(background * volume + foreground, -1, 1) in float32."""
    bg = background_crop.astype(np.float32) * np.float32(background_volume)
    return np.clip(bg + foreground.astype(np.float32), np.float32(-1.0), np.float32(1.0)).astype(np.float32)


def draw(rng: np.random.RandomState, desired_samples: int, n_background, background_lengths, is_training: bool,
         background_frequency: float, background_max_volume: float, with_shift: bool, shift_ratio: float = 0.1):
    """One element's random draws in the order of the reference graph (naive_version=True, :69-79): shift amount in
    [-desired*ratio, desired*ratio); background index; crop offset; mix decision; volume."""
    time_shift = int(desired_samples * shift_ratio)
    shift = int(rng.randint(-time_shift, time_shift)) if with_shift and time_shift > 0 else 0
    bg_idx, bg_crop, vol = 0, 0, 0.0
    if n_background:
        bg_idx = int(rng.randint(0, n_background))
        bg_crop = int(rng.randint(0, background_lengths[bg_idx] - desired_samples + 1))
        if is_training and rng.uniform() < background_frequency:
            vol = float(np.float32(rng.uniform(0.0, background_max_volume)))
    return shift, bg_idx, bg_crop, vol


def augment_batch(pcm_pool: np.ndarray, clip_off, clip_len, shift, bg_pool, bg_off, bg_vol, desired_samples: int) -> np.ndarray:
    """The whole stage for a batch, element by element (anchored_slice_or_pad_with_shift, :190-211; shift = 0 gives
    anchored_slice_or_pad, :165-187; bg_vol = 0 and shift = 0 give no_augmentation_audio, :158-162)."""
    b = len(clip_off)
    out = np.zeros((b, desired_samples), np.float32)
    for i in range(b):
        clip = pcm_pool[clip_off[i]:clip_off[i] + clip_len[i]]
        audio = shift_audio(decode_wav(clip, desired_samples), shift[i])
        if bg_pool is not None and bg_vol[i] != 0.0:
            crop = bg_pool[bg_off[i]:bg_off[i] + desired_samples].astype(np.float32) * np.float32(1.0 / 32768.0)
            out[i] = mix_background(audio, crop, bg_vol[i])
        else:
            out[i] = np.clip(audio, -1.0, 1.0)
    return out
