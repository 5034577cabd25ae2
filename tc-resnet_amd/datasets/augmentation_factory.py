"""Audio augmentation with the reference's names (datasets/augmentation_factory.py:6-211), batched on the device.

The reference maps `anchored_slice_or_pad_with_shift(filename, ...)` over a tf.data pipeline on 8 CPU threads
(WAV decode, crop/pad, random shift, background mix).  Here the decoded 16-bit clips live in one device pool
(`PcmPool`) and a batch is produced by ONE kernel launch (`tcr_augment_fwd`); the random draws the reference makes
inside its graph come from a seeded host generator, in the same per-element order, and are passed to the kernel.
"""
from __future__ import annotations

import struct
from typing import List, Optional, Sequence

import numpy as np
import torch

from .. import runtime
from ..engine import _resolve

_available_audio_augmentation_methods = [
    "anchored_slice_or_pad",
    "anchored_slice_or_pad_with_shift",
    "no_augmentation_audio",
]

_available_augmentation_methods = _available_audio_augmentation_methods + ["no_augmentation"]


def read_wav_pcm16(path: str) -> np.ndarray:
    """Mono 16-bit PCM samples of a RIFF/WAVE file (what contrib_audio.decode_wav accepts, :143-155); multi-channel files
    keep channel 0 (desired_channels=1)."""
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, channels, bits, pcm = 12, 1, 16, None
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt, channels, _rate, _br, _align, bits = struct.unpack("<HHIIHH", body[:16])
            if fmt != 1 or bits != 16:
                raise ValueError(f"{path}: only 16-bit PCM is supported (format {fmt}, {bits} bits)")
        elif tag == b"data":
            pcm = np.frombuffer(body[:len(body) // 2 * 2], dtype="<i2")
        pos += 8 + size + (size & 1)
    if pcm is None:
        raise ValueError(f"{path}: no data chunk")
    return np.ascontiguousarray(pcm.reshape(-1, channels)[:, 0])


class PcmPool:
    """All clips of a split (or all background recordings) as one int16 device tensor + offsets."""

    def __init__(self, clips: Sequence[np.ndarray], device=None, pad: int = 0):
        self.lib, self.device = _resolve(runtime.default_lib(), device if device is not None else runtime.default_device())
        self.lengths = np.array([len(c) for c in clips], dtype=np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(self.lengths)[:-1]]).astype(np.int64) if len(clips) else np.zeros(0, np.int64)
        flat = np.concatenate([np.asarray(c, dtype=np.int16) for c in clips] + [np.zeros(pad, np.int16)]) if len(clips) else np.zeros(max(pad, 1), np.int16)
        self.data = torch.from_numpy(flat).to(self.device)

    def __len__(self):
        return len(self.lengths)

    @classmethod
    def from_files(cls, paths: Sequence[str], device=None):
        return cls([read_wav_pcm16(p) if p else np.zeros(0, np.int16) for p in paths], device=device)


def no_augmentation(x):
    return x


class _Augmenter:
    with_shift = False
    mix = True

    def __call__(self, pool: PcmPool, indices: Sequence[int], desired_samples: int, file_format: str = "wav", sample_rate: int = 16000,
                 background_data: Optional[PcmPool] = None, is_training: bool = True, background_frequency: float = 0.8,
                 background_max_volume: float = 0.1, rng: Optional[np.random.RandomState] = None, shift_ratio: float = 0.1,
                 out: Optional[torch.Tensor] = None, draws=None, **_kwargs) -> torch.Tensor:
        """indices: clip ids of the batch (an id whose clip is empty is a "silent" sample).  Returns [B, desired, 1] f32.
        draws: optional precomputed (shift, bg_index, bg_crop, bg_vol) arrays (tests); otherwise drawn from `rng`."""
        if file_format != "wav":
            raise ValueError(f"Unsupported file format: {file_format}")
        lib, dev = pool.lib, pool.device
        b = len(indices)
        idx = np.asarray(indices, dtype=np.int64)
        nbg = len(background_data) if (background_data is not None and self.mix) else 0
        if draws is None:
            rng = rng or np.random.RandomState(0)
            shift, bg_idx, bg_crop, vol = np.zeros(b, np.int32), np.zeros(b, np.int64), np.zeros(b, np.int64), np.zeros(b, np.float32)
            time_shift = int(desired_samples * shift_ratio)
            if b > 64:              # large batches: the same distributions drawn array-wise (a Python loop would cost ~3 us x 4 per element)
                if self.with_shift and time_shift > 0:
                    shift = rng.randint(-time_shift, time_shift, b).astype(np.int32)
                if nbg:
                    bg_idx = rng.randint(0, nbg, b).astype(np.int64)
                    span = background_data.lengths[bg_idx] - desired_samples + 1
                    bg_crop = np.minimum((rng.random_sample(b) * span).astype(np.int64), span - 1)
                    if is_training:
                        vol = (rng.uniform(0.0, background_max_volume, b) * (rng.uniform(size=b) < background_frequency)).astype(np.float32)
            for i in range(b if b <= 64 else 0):      # the reference's per-element order of draws (augmentation_factory.py:104-110, 58-79)
                if self.with_shift and time_shift > 0:
                    shift[i] = rng.randint(-time_shift, time_shift)
                if nbg:
                    bg_idx[i] = rng.randint(0, nbg)
                    bg_crop[i] = rng.randint(0, background_data.lengths[bg_idx[i]] - desired_samples + 1)
                    if is_training and rng.uniform() < background_frequency:
                        vol[i] = np.float32(rng.uniform(0.0, background_max_volume))
        else:
            shift, bg_idx, bg_crop, vol = (np.asarray(x) for x in draws)
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        clip_off, clip_len = t(pool.offsets[idx], np.int64), t(pool.lengths[idx], np.int32)
        d_shift = t(shift, np.int32) if self.with_shift else None
        d_bgoff = t(background_data.offsets[bg_idx] + bg_crop, np.int64) if nbg else None
        d_vol = t(vol, np.float32) if nbg else None
        if out is None:
            out = torch.empty((b, desired_samples, 1), dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
        p = lambda x: None if x is None else x.data_ptr()
        lib.check(lib.tcr_augment_fwd(pool.data.data_ptr(), clip_off.data_ptr(), clip_len.data_ptr(), p(d_shift),
                                      background_data.data.data_ptr() if nbg else None, p(d_bgoff), p(d_vol), b, int(desired_samples),
                                      out.data_ptr(), stream), "tcr_augment_fwd")
        self.last_draws = (shift, bg_idx, bg_crop, vol)
        return out


class _NoAug(_Augmenter):
    mix = False


class _Anchored(_Augmenter):
    pass


class _AnchoredShift(_Augmenter):
    with_shift = True


no_augmentation_audio = _NoAug()
anchored_slice_or_pad = _Anchored()
anchored_slice_or_pad_with_shift = _AnchoredShift()


def get_audio_augmentation_fn(name):
    if name not in _available_audio_augmentation_methods:
        raise ValueError(f"Augmentation name [{name}] was not recognized")
    return {"anchored_slice_or_pad": anchored_slice_or_pad, "anchored_slice_or_pad_with_shift": anchored_slice_or_pad_with_shift,
            "no_augmentation_audio": no_augmentation_audio}[name]
