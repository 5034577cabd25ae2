"""Synthetic 1 s @ 16 kHz utterances generated on the device (SURVEY 8(d) protocol: uniform noise * 0.5 plus a
per-utterance sine at 0.25).  Stands in for SingleLabelAudioDataWrapper (datasets/audio_data_wrapper.py:113-175):
the real tf.data / WAV-decode / augmentation pipeline is outside the hot path."""
from __future__ import annotations

import torch

from .. import runtime


class SyntheticAudioDataWrapper:
    def __init__(self, args, session=None, dataset_split_name="train", is_training=True, device=None, num_samples=22246):
        self.args = args
        self.batch_size = int(args.batch_size)
        self.num_classes = int(args.num_classes or 12)
        self.num_samples = int(num_samples)             # size of the reference's training split
        self.device = torch.device(device if device is not None else (runtime.default_device() or "cuda"))
        self.n_samples = int(args.sample_rate * args.clip_duration_ms / 1000)
        self.is_training = is_training
        self._cursor = 0
        self._gen = torch.Generator(device=self.device).manual_seed(1234)
        self.label_names = ["__null__", "down", "go", "left", "no", "off", "on", "right", "stop", "unknown", "up", "yes"][:self.num_classes]

    def next_batch(self, rank: int = 0, world: int = 1):
        b = self.batch_size
        start = self._cursor + rank * b
        self._cursor += b * world
        idx = torch.arange(b, device=self.device) + start
        noise = torch.rand((b, self.n_samples), generator=self._gen, device=self.device) - 0.5
        t = torch.arange(self.n_samples, device=self.device, dtype=torch.float32) / float(self.args.sample_rate)
        f = 440.0 * (1 + idx % 8).to(torch.float32)
        wav = (noise + 0.25 * torch.sin(2.0 * torch.pi * f[:, None] * t[None, :])).unsqueeze(-1).contiguous()
        labels = torch.zeros((b, self.num_classes), device=self.device)
        labels[torch.arange(b, device=self.device), idx % self.num_classes] = 1.0      # one-hot float32
        return wav, labels

    def setup_iterator(self, *_args):
        """Re-initialise the iterator: the next batch is the first of the split again (data_wrapper_base.py setup_iterator)."""
        self._cursor = 0
        self._gen.manual_seed(1234)

    def get_input_and_output_op(self):
        """Reference name: the (wavs [B, n, 1], labels [B, num_classes]) pair the graph is built on.  A peek: the iterator is
        where it was afterwards, like building the reference's graph does not consume the dataset."""
        state, cursor = self._gen.get_state(), self._cursor
        out = self.next_batch()
        self._gen.set_state(state)
        self._cursor = cursor
        return out
