"""SingleLabelAudioDataWrapper over a directory of WAV files (datasets/audio_data_wrapper.py:113-175 +
datasets/data_wrapper_base.py:107-191 of the reference), with the per-element work on the device.

Layout, as in the reference: <dataset_path>/<split>/<label>/*.wav; directories whose name starts with "_" are not
labels; "_background_noise_" holds the recordings mixed in by `_mix_background`; `--add_null_class` puts "__null__" first
and `--num_silent` samples with an empty filename are appended to it (-1: num_samples // num_labels).

The reference decodes / augments each element in a tf.data map on `--num_threads` CPU threads.  Here every clip is decoded
once at start-up into an int16 device pool (`PcmPool`) and a batch is ONE launch of `tcr_augment_fwd` through the
reference-named augmentation function (`--augmentation_method`).
"""
from __future__ import annotations

import logging
import os
import random
from pathlib import Path
from typing import List

import numpy as np
import torch

from .augmentation_factory import PcmPool, get_audio_augmentation_fn, read_wav_pcm16

NULL_CLASS_LABEL = "__null__"                       # const.py:8
BACKGROUND_NOISE_DIR_NAME = "_background_noise_"    # const.py:9


class SingleLabelAudioDataWrapper:
    def __init__(self, args, session=None, dataset_split_name="train", is_training=True, name="SingleLabelAudioDataWrapper", device=None):
        self.args, self.is_training, self.dataset_split_name = args, bool(is_training), dataset_split_name
        self.log = logging.getLogger(name)
        self.batch_size = int(args.batch_size)
        self.shuffle = bool(getattr(args, "shuffle", True))
        root = Path(args.dataset_path) / dataset_split_name
        if not root.is_dir():
            raise FileNotFoundError(f"{root}: no such dataset split directory")
        # labels: sorted directory names, "_"-prefixed ones skipped, the null class first (data_wrapper_base.py:118-146)
        self.label_names: List[str] = ([NULL_CLASS_LABEL] if getattr(args, "add_null_class", True) else []) + \
            sorted(c.name for c in root.iterdir() if c.is_dir() and c.name[0] != "_")
        self.num_labels = len(self.label_names)
        if NULL_CLASS_LABEL not in self.label_names or int(args.num_classes) != self.num_labels:
            raise ValueError(f"--num_classes {args.num_classes} != {self.num_labels} labels found ({self.label_names}); "
                             f"'{NULL_CLASS_LABEL}' must be among them (audio_data_wrapper.py:124-125)")
        filenames, labels = [], []
        for idx, cname in enumerate(self.label_names):          # data_wrapper_base.py:171-176
            d = root / cname
            if d.is_dir():
                for f in sorted(d.glob("*")):
                    filenames.append(str(f))
                    labels.append(idx)
        if not filenames:
            raise ValueError(f"{root}: no input files")
        # silent samples: empty filename, null label (audio_data_wrapper.py:163-174)
        num_silent = int(getattr(args, "num_silent", -1))
        if num_silent < 0:
            num_silent = len(filenames) // self.num_labels
        null_idx = self.label_names.index(NULL_CLASS_LABEL)
        filenames += [""] * num_silent
        labels += [null_idx] * num_silent
        self.log.info("%d silent samples will be added.", num_silent)
        self.filenames, self.labels = filenames, np.asarray(labels, dtype=np.int64)
        self.num_samples = len(filenames)
        self.pool = PcmPool.from_files(filenames, device=device)
        bg_dir = root / BACKGROUND_NOISE_DIR_NAME               # audio_data_wrapper.py:146-160
        bg_files = sorted(str(p) for p in bg_dir.glob("*.wav")) if bg_dir.is_dir() else []
        self.desired_samples = int(args.sample_rate * args.clip_duration_ms / 1000)
        clips = [read_wav_pcm16(p) for p in bg_files]
        clips = [c for c in clips if len(c) >= self.desired_samples]       # tf.random_crop needs at least one full window
        self.background = PcmPool(clips, device=device) if clips else None
        self.log.info("%d background files are loaded.", len(clips))
        self._aug = get_audio_augmentation_fn(args.augmentation_method)
        # two generators: the epoch reshuffle must be IDENTICAL on every data-parallel rank (shards of one global order),
        # the augmentation draws must DIFFER per rank (else every shard gets the same shifts / background crops)
        self._seed = int(getattr(args, "seed", 0) or 0)
        self._shuffle_rng = np.random.RandomState(self._seed)
        self._aug_rngs = {}
        self._order = np.arange(self.num_samples)
        self._cursor = 0
        if self.shuffle:
            random.Random(0).shuffle(self._order)
        self._order0 = self._order.copy()

    def _aug_rng(self, rank: int) -> np.random.RandomState:
        if rank not in self._aug_rngs:
            self._aug_rngs[rank] = np.random.RandomState((self._seed + 7919 * (rank + 1)) % (2 ** 31))
        return self._aug_rngs[rank]

    def setup_iterator(self, *_args):
        """Re-initialise the iterator (data_wrapper_base.py setup_iterator): evaluation starts from the first sample."""
        self._cursor = 0
        self._order = self._order0.copy()
        self._shuffle_rng = np.random.RandomState(self._seed)
        self._aug_rngs = {}

    def next_batch(self, rank: int = 0, world: int = 1):
        """(wavs [B, desired_samples, 1] f32, labels_onehot [B, num_classes] f32) for this rank's shard of the next global batch."""
        b = self.batch_size
        idx = np.empty(b, dtype=np.int64)
        for i in range(b):
            pos = self._cursor + rank * b + i
            if pos >= self.num_samples:                         # .repeat(-1) for training; evaluation wraps too (callers bound the loop)
                pos %= self.num_samples
            idx[i] = self._order[pos]
        self._cursor += b * world
        if self._cursor >= self.num_samples:
            self._cursor %= self.num_samples
            if self.is_training and self.shuffle:
                self._shuffle_rng.shuffle(self._order)
        wavs = self._aug(self.pool, idx, self.desired_samples, "wav", int(self.args.sample_rate), background_data=self.background,
                         is_training=self.is_training, background_frequency=float(getattr(self.args, "background_frequency", 0.8)),
                         background_max_volume=float(getattr(self.args, "background_max_volume", 0.1)), rng=self._aug_rng(rank))
        labels = torch.zeros((b, self.num_labels), dtype=torch.float32, device=wavs.device)
        labels[torch.arange(b, device=wavs.device), torch.from_numpy(self.labels[idx]).to(wavs.device)] = 1.0      # parse_label :114-118
        return wavs, labels

    def get_input_and_output_op(self):
        """The (wavs, labels) pair the model is first built on.  A peek: cursor, order and generators are restored, like
        building the reference's graph does not consume the dataset."""
        import copy
        saved = (self._cursor, self._order.copy(), copy.deepcopy(self._shuffle_rng), copy.deepcopy(self._aug_rngs))
        out = self.next_batch()
        self._cursor, self._order, self._shuffle_rng, self._aug_rngs = saved
        return out
