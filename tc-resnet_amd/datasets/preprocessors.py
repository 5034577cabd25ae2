"""Audio preprocessors with the reference's names and call signature (datasets/preprocessors.py:8-203).

The reference builds TF graph ops; here `preprocess()` runs the fused HIP front-end immediately and returns the
feature tensor in the reference's shape [B, T, F, 1] (a view of the planar halo buffer the network kernels
consume, kept on `.planar`)."""
from __future__ import annotations

from abc import ABC, abstractmethod

import torch

from .. import runtime
from ..engine import Frontend, features_to_planar


class PreprocessorBase(ABC):
    def __init__(self, scope: str, preprocessed_node_name: str):
        self._scope = scope
        self._input_node = None
        self._preprocessed_node = None
        self._preprocessed_node_name = preprocessed_node_name
        self.planar = None          # [B, F, T + 2*HALO] buffer fed to the network kernels

    @abstractmethod
    def preprocess(self, inputs, reuse=False):
        raise NotImplementedError

    @property
    def input_node(self):
        return self._input_node

    @property
    def preprocessed_node(self):
        return self._preprocessed_node


class NoOpPreprocessor(PreprocessorBase):
    """--preprocess_method no_preprocessing: features arrive as [B, T, F, 1] (reference :45-49)."""

    def preprocess(self, inputs, reuse=False, **kwargs):
        self._input_node = inputs
        self.planar = features_to_planar(inputs, lib=runtime.default_lib())
        self._preprocessed_node = inputs if inputs.dim() == 4 else inputs.unsqueeze(-1)
        return self._preprocessed_node


class AudioPreprocessorBase(PreprocessorBase):
    method = None

    def __init__(self, scope, preprocessed_node_name):
        super().__init__(scope, preprocessed_node_name)
        self._frontend = None
        self._key = None

    def preprocess(self, inputs, window_size_samples, window_stride_samples, for_deploy, **kwargs):
        """kwargs read (as in the reference, :80-86,192): num_mel_bins, sample_rate, lower_edge_hertz,
        upper_edge_hertz, num_mfccs."""
        method = self.method
        if for_deploy:
            # the contrib_audio.audio_spectrogram + mfcc deploy variant (:98-124,196-203) is a different MFCC definition
            # (SURVEY F7); the reference only takes batch 1 there, this kernel takes any batch.  The log-mel deploy path of
            # the reference cannot run (undefined const.MEL_WEIGHT_...).
            if method != "mfcc":
                raise NotImplementedError("for_deploy=True exists for the mfcc preprocessor only (the reference's log-mel deploy path is broken)")
            method = "mfcc_deploy"
        self._input_node = inputs
        sr = int(kwargs.get("sample_rate", 16000))
        n = inputs.shape[1]
        key = (sr, n, int(window_size_samples), int(window_stride_samples), int(kwargs.get("num_mel_bins", 64)),
               int(kwargs.get("num_mfccs", 40)), float(kwargs.get("lower_edge_hertz", 80.0)),
               float(kwargs.get("upper_edge_hertz", 7600.0)), method)
        if self._frontend is None or key != self._key:
            self._frontend = Frontend(sample_rate=sr, clip_duration_ms=n * 1000 // sr, window_size_samples=key[2],
                                      window_stride_samples=key[3], num_mel_bins=key[4], num_mfccs=key[5],
                                      lower_edge_hertz=key[6], upper_edge_hertz=key[7], method=method,
                                      lib=runtime.default_lib(), device=runtime.default_device() or inputs.device)
            self._key = key
        wav = inputs if inputs.is_contiguous() else inputs.contiguous()
        self.planar = self._frontend(wav)
        self._preprocessed_node = self._frontend.reference_view(self.planar)
        return self._preprocessed_node


class LogMelSpectrogramPreprocessor(AudioPreprocessorBase):
    method = "log_mel_spectrogram"       # magnitude spectrum, no DCT (reference :161-169)


class MFCCPreprocessor(AudioPreprocessorBase):
    method = "mfcc"                      # power spectrum + DCT-II, first num_mfccs (reference :183-194)
