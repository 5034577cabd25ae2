"""Host-side objects over the C ABI: the MFCC front-end and the TC-ResNet engine.

PyTorch is used here for device memory, streams and (in `parallel.py`) torch.distributed only --
every number on the path is produced by the HIP kernels behind `include/tcresnet_hip.h`.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import HALO, DSCNNCfg, FrontendCfg, TCResNetCfg, TensorInfo, TcrError, padded_len


DP_TAIL = 16        # floats behind the gradient arena that ride along with its all-reduce (the replicas' loss sum): one collective per step


def _resolve(lib: Optional[_lib.Library], device) -> Tuple[_lib.Library, torch.device]:
    if lib is None:
        lib = _lib.get()            # raises when the HIP extension is missing
    if lib.kind == "hip":
        if not torch.cuda.is_available():
            raise TcrError("the gfx950 library needs a visible AMD GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
        dev = torch.device(device if device is not None else "cuda")
        if dev.type != "cuda":
            raise TcrError(f"device {dev} is not a GPU; the HIP path has no CPU fallback")
    else:                            # emulator build: tests only
        dev = torch.device("cpu")
    return lib, dev


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class _Base:
    _kver = 0            # bumped by every method that lets a kernel write the variables / moving statistics (TCResNet caches its BN fold on it)

    def _stream(self):
        if self.device.type == "cuda":
            return torch.cuda.current_stream(self.device).cuda_stream
        return None

    def tf_shape(self, name: str) -> Tuple[int, ...]:
        """Shape of variable `name` as the reference's TF graph declares it (what a TF-written checkpoint holds)."""
        over = getattr(self, "tf_shapes", None)
        if over and name in over:
            return tuple(over[name])
        ti = self.tensors[name]
        return tuple(ti.shape[i] for i in range(ti.rank))

    def _check_tensor(self, t: torch.Tensor, what: str):
        if t.device.type != self.device.type or t.dtype != torch.float32 or not t.is_contiguous():
            raise TcrError(f"{what}: expected a contiguous float32 tensor on {self.device}, got {t.dtype} on {t.device}"
                           f"{'' if t.is_contiguous() else ' (non-contiguous)'}")


class Frontend(_Base):
    """MFCC / log-mel front-end (datasets/preprocessors.py:54-96,183-194 of the reference)."""

    METHODS = {"mfcc": 0, "log_mel_spectrogram": 1, "mfcc_deploy": 2}

    def __init__(self, sample_rate: int = 16000, clip_duration_ms: int = 1000, window_size_samples: int = 480,
                 window_stride_samples: int = 160, num_mel_bins: int = 64, num_mfccs: int = 40,
                 lower_edge_hertz: float = 80.0, upper_edge_hertz: float = 7600.0, method: str = "mfcc",
                 lib: Optional[_lib.Library] = None, device=None):
        self.lib, self.device = _resolve(lib, device)
        if method not in self.METHODS:
            raise NotImplementedError(method)
        cfg = FrontendCfg(int(sample_rate), int(sample_rate * clip_duration_ms / 1000), int(window_size_samples),
                          int(window_stride_samples), 0, 0, int(num_mel_bins), int(num_mfccs),
                          float(lower_edge_hertz), float(upper_edge_hertz), self.METHODS[method])
        self.lib.check(self.lib.tcr_frontend_resolve(C.byref(cfg)), "tcr_frontend_resolve")
        self.cfg = cfg
        nbytes = self.lib.tcr_frontend_plan_bytes(C.byref(cfg))
        host = torch.zeros(nbytes // 4, dtype=torch.float32)
        self.lib.check(self.lib.tcr_frontend_plan_init(C.byref(cfg), host.data_ptr()), "tcr_frontend_plan_init")
        self._plan_host = host
        self.plan = host.to(self.device)

    # shapes ------------------------------------------------------------------------------------
    @property
    def n_frames(self) -> int:
        return self.cfg.n_frames

    @property
    def n_coef(self) -> int:
        return self.cfg.n_coef

    @property
    def n_samples(self) -> int:
        return self.cfg.n_samples

    def mel_matrix(self) -> np.ndarray:
        out = np.zeros((self.cfg.nfft // 2 + 1, self.cfg.n_mel), np.float32)
        self.lib.check(self.lib.tcr_frontend_plan_mel_matrix(C.byref(self.cfg), self._plan_host.data_ptr(), out.ctypes.data),
                       "tcr_frontend_plan_mel_matrix")
        return out

    def dct_matrix(self) -> np.ndarray:
        out = np.zeros((self.cfg.n_mel, self.cfg.n_coef), np.float32)
        self.lib.check(self.lib.tcr_frontend_plan_dct_matrix(C.byref(self.cfg), self._plan_host.data_ptr(), out.ctypes.data),
                       "tcr_frontend_plan_dct_matrix")
        return out

    # compute -----------------------------------------------------------------------------------
    def __call__(self, wav: torch.Tensor, out: Optional[torch.Tensor] = None, rounds: int = 0) -> torch.Tensor:
        """wav [B, n_samples] or [B, n_samples, 1] -> planar features [B, n_coef, T + 2*HALO].  rounds > 0: a per-call launch hint
        (rounds of frames per persistent-workgroup chunk, tcr_frontend_fwd_rounds); the features do not depend on it."""
        if wav.dim() == 3:
            if wav.shape[-1] != 1:
                raise TcrError("front-end accepts single-channel audio only")     # tf.squeeze(audio, -1)
            wav = wav[..., 0]
        if wav.dim() != 2 or wav.shape[1] != self.cfg.n_samples:
            raise TcrError(f"front-end expects [B, {self.cfg.n_samples}] waveforms, got {tuple(wav.shape)}")
        self._check_tensor(wav, "front-end input")
        b = wav.shape[0]
        if out is None:
            out = torch.empty((b, self.cfg.n_coef, padded_len(self.cfg.n_frames)), dtype=torch.float32, device=self.device)
        self.lib.check(self.lib.tcr_frontend_fwd_rounds(C.byref(self.cfg), self.plan.data_ptr(), wav.data_ptr(), b, out.data_ptr(),
                                                        int(rounds), self._stream()), "tcr_frontend_fwd")
        return out

    def reference_view(self, feat: torch.Tensor) -> torch.Tensor:
        """Planar [B, F, Tp] -> the reference's [B, T, F, 1] (a view, no copy)."""
        return feat[:, :, HALO:HALO + self.cfg.n_frames].permute(0, 2, 1).unsqueeze(-1)


def features_to_planar(x: torch.Tensor, lib: Optional[_lib.Library] = None) -> torch.Tensor:
    """Reference-layout features [B, T, F] or [B, T, F, 1] -> planar halo layout [B, F, T + 2*HALO]."""
    if x.dim() == 4:
        x = x[..., 0]
    x = x.contiguous()
    lib2, dev = _resolve(lib, x.device)
    b, t, f = x.shape
    out = torch.empty((b, f, padded_len(t)), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else None
    lib2.check(lib2.tcr_features_to_planar(x.data_ptr(), b, t, f, out.data_ptr(), stream), "tcr_features_to_planar")
    return out


class TCResNet(_Base):
    """TC-ResNet parameters + kernels (audio_nets/tc_resnet.py:6-70 of the reference)."""

    KIND = {0: "weight", 1: "gamma", 2: "beta", 3: "moving_mean", 4: "moving_variance"}

    def __init__(self, scope: str, channels: Sequence[int], in_channels: int, t_in: int, num_classes: int,
                 bn_decay: float = 0.997, bn_eps: float = 0.001, lib: Optional[_lib.Library] = None, device=None):
        self.lib, self.device = _resolve(lib, device)
        cfg = TCResNetCfg()
        cfg.scope = scope.encode()
        cfg.in_channels, cfg.t_in, cfg.num_classes = int(in_channels), int(t_in), int(num_classes)
        cfg.n_blocks = len(channels) - 1
        if cfg.n_blocks > _lib.MAX_BLOCKS:
            raise TcrError(f"at most {_lib.MAX_BLOCKS} blocks")
        for i, c in enumerate(channels):
            cfg.channels[i] = int(c)
        cfg.bn_decay, cfg.bn_eps = float(bn_decay), float(bn_eps)
        handle = C.c_void_p()
        self.lib.check(self.lib.tcr_tcresnet_create(C.byref(cfg), C.byref(handle)), "tcr_tcresnet_create")
        self.cfg, self._h = cfg, handle
        self.scope, self.channels = scope, list(channels)
        self.num_classes, self.in_channels, self.t_in = int(num_classes), int(in_channels), int(t_in)
        self.n_param = self.lib.tcr_net_param_floats(handle)
        self.n_decay = self.lib.tcr_net_decay_floats(handle)
        self.n_stat = self.lib.tcr_net_stat_floats(handle)
        self.tensors: Dict[str, TensorInfo] = {}
        for i in range(self.lib.tcr_net_num_tensors(handle)):
            ti = TensorInfo()
            self.lib.check(self.lib.tcr_net_tensor_info(handle, i, C.byref(ti)), "tcr_net_tensor_info")
            self.tensors[ti.name.decode()] = ti
        self.params = torch.zeros(self.n_param, dtype=torch.float32, device=self.device)
        self.stats = torch.zeros(self.n_stat, dtype=torch.float32, device=self.device)
        self._grads_buf = torch.zeros(self.n_param + DP_TAIL, dtype=torch.float32, device=self.device)     # arena + the data-parallel tail (parallel.py)
        self.grads = self._grads_buf[:self.n_param]
        self.slots: Dict[str, torch.Tensor] = {}      # optimiser slots (arena-shaped)
        self._ws: Dict[Tuple[int, int], torch.Tensor] = {}
        self._kver, self._fold_key, self._fold_ss, self._fold_event, self._fold_stream, self._fold_readers = 0, None, None, None, None, {}
        self._fold_waited = set()
        self._wave_feat: Dict[Tuple[int, int, int], torch.Tensor] = {}
        self.handoff = "level"         # cross-replica BN hand-off granularity: "level" (default) or "unit" (tcr_net_*_stage)
        self.reset_bn()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.tcr_net_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- parameters ---------------------------------------------------------------------------
    def _view(self, name: str) -> torch.Tensor:
        ti = self.tensors[name]
        arena = self.params if ti.arena == 0 else self.stats
        return arena[ti.offset:ti.offset + ti.size].view(*[ti.shape[i] for i in range(ti.rank)])

    def grad_view(self, name: str) -> torch.Tensor:
        ti = self.tensors[name]
        assert ti.arena == 0
        return self.grads[ti.offset:ti.offset + ti.size].view(*[ti.shape[i] for i in range(ti.rank)])

    def trainable_names(self) -> List[str]:
        return [n for n, ti in self.tensors.items() if ti.arena == 0]

    def total_params(self) -> int:
        return sum(int(ti.size) for ti in self.tensors.values() if ti.arena == 0)

    def reset_bn(self):
        for n, ti in self.tensors.items():
            if ti.kind in (1, 4):       # gamma, moving_variance -> 1
                self._view(n).fill_(1.0)
            elif ti.kind in (2, 3):
                self._view(n).zero_()

    def init_xavier(self, seed: int = 0):
        """slim.initializers.xavier_initializer() (uniform) for every conv/fc weight (tc_resnet.py:112)."""
        gen = torch.Generator().manual_seed(int(seed))
        for n, ti in self.tensors.items():
            if ti.kind != 0:
                continue
            k, _, cin, cout = (ti.shape[i] for i in range(4))
            lim = math.sqrt(6.0 / (k * cin + k * cout))
            w = (torch.rand((k, 1, cin, cout), generator=gen, dtype=torch.float32) * 2.0 - 1.0) * lim
            self._view(n).copy_(w.to(self.device))
        self.reset_bn()

    def state_dict(self) -> Dict[str, np.ndarray]:
        """TF variable name -> array in the reference checkpoint shape."""
        return {n: self._view(n).detach().cpu().numpy().copy() for n in self.tensors}

    def load_state_dict(self, sd: Dict[str, np.ndarray], strict: bool = True):
        for n, ti in self.tensors.items():
            if n not in sd:
                if strict:
                    raise KeyError(n)
                continue
            v = torch.as_tensor(np.asarray(sd[n], dtype=np.float32))
            shape = tuple(ti.shape[i] for i in range(ti.rank))
            if ti.rank == 4 and v.dim() == 3:       # [k, Cin, Cout] accepted for [k, 1, Cin, Cout]
                v = v.unsqueeze(1)
            if tuple(v.shape) != shape:
                raise TcrError(f"{n}: shape {tuple(v.shape)} != {shape}")
            self._view(n).copy_(v.to(self.device))

    # ---- workspaces ---------------------------------------------------------------------------
    def workspace(self, batch: int, train: bool) -> torch.Tensor:
        key = (int(batch), int(train))
        ws = self._ws.get(key)
        if ws is None:
            nbytes = self.lib.tcr_net_workspace_bytes(self._h, batch, int(train))
            # (training: zeroed once -- the cross-replica hand-off all-reduces one contiguous range over both float64 sum slots, gap included)
            ws = (torch.zeros if train else torch.empty)(nbytes // 4, dtype=torch.float32, device=self.device)
            self._ws[key] = ws
        return ws

    def new_workspace(self, batch: int, train: bool) -> torch.Tensor:
        """A fresh scratch tensor of the size `workspace(batch, train)` has (not cached, not shared)."""
        return torch.empty(self.lib.tcr_net_workspace_bytes(self._h, batch, int(train)) // 4, dtype=torch.float32, device=self.device)

    def _check_feat(self, feat: torch.Tensor):
        self._check_tensor(feat, "features")
        want = (self.in_channels, padded_len(self.t_in))
        if feat.dim() != 3 or tuple(feat.shape[1:]) != want:
            raise TcrError(f"features must be planar [B, {want[0]}, {want[1]}], got {tuple(feat.shape)}")

    # ---- compute ------------------------------------------------------------------------------
    def _weights_version(self):
        """Changes whenever the variables or the moving statistics may have changed: torch-side writes bump the tensors' version
        counters, kernel-side writes (train-mode forward, optimiser steps) bump `_kver` in the methods that launch them."""
        return (self.params._version, self.stats._version, self._kver, self.params.data_ptr(), self.stats.data_ptr())

    def invalidate_folded(self):
        """Call after writing `params` / `stats` through a path torch does not see (a raw pointer handed to another library)."""
        self._kver += 1

    def _folded_table(self) -> torch.Tensor:
        """Eval-mode BN folded to (scale, shift) ONCE per weight version (the fold used to be a launch in front of every eval
        forward although eval weights never change between steps); refolded after load_state_dict / an optimiser step / a
        train-mode forward.  Other streams order themselves behind the fold with an event."""
        key = self._weights_version()
        cur = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        if self._fold_key != key or self._fold_ss is None:
            if self._fold_ss is None:
                self._fold_ss = torch.zeros(self.lib.tcr_net_frozen_floats(self._h), dtype=torch.float32, device=self.device)
            elif cur is not None and self._fold_readers:
                for h, st in self._fold_readers.items():    # forwards that may still read the old table on other streams (the previous fold
                    if h != cur.cuda_stream:                # stream included): everything issued on them so far
                        cur.wait_stream(st)
            self.lib.check(self.lib.tcr_net_fold_bn(self._h, self.params.data_ptr(), self.stats.data_ptr(), self._fold_ss.data_ptr(),
                                                    self._stream()), "tcr_net_fold_bn")
            self._fold_key = key
            self._fold_readers = {}
            self._fold_waited = set()
            if cur is not None:
                self._fold_event = torch.cuda.Event()
                self._fold_event.record(cur)
                self._fold_stream = cur.cuda_stream
        elif cur is not None and cur.cuda_stream != self._fold_stream and cur.cuda_stream not in self._fold_waited:
            cur.wait_event(self._fold_event)                # once per stream and fold: stream order covers the stream's later forwards
            self._fold_waited.add(cur.cuda_stream)
        return self._fold_ss

    def forward_infer(self, feat: torch.Tensor, want_ranges: bool = False, out=None, workspace: Optional[torch.Tensor] = None):
        """Eval-mode forward.  `out=(logits, probs)` reuses caller-owned output tensors (stream pipelines); `workspace`: a caller-owned
        scratch tensor of `new_workspace(batch, False)`'s size instead of the net's own -- forwards in flight on two streams at once need
        one each (layer activations live there whenever the shape has no fused kernel)."""
        self._check_feat(feat)
        b = feat.shape[0]
        ws = self.workspace(b, False) if workspace is None else workspace
        if out is not None:
            logits, probs = out
        else:
            logits = torch.empty((b, self.num_classes), dtype=torch.float32, device=self.device)
            probs = torch.empty_like(logits)
        ranges = torch.empty((b, 2), dtype=torch.float32, device=self.device) if want_ranges else None
        ss = self._folded_table()
        self.lib.check(self.lib.tcr_net_forward_frozen(self._h, self.params.data_ptr(), ss.data_ptr(), feat.data_ptr(), b,
                                                       ws.data_ptr(), ws.numel() * 4, logits.data_ptr(), probs.data_ptr(),
                                                       _ptr(ranges), self._stream()), "tcr_net_forward_frozen")
        self._note_fold_reader()
        return (logits, probs, ranges) if want_ranges else (logits, probs)

    def _note_fold_reader(self):
        """A forward on the current stream has read the folded table: a refold issued from ANOTHER stream must wait for it (readers on
        the fold stream itself are noted too -- the next fold may come from a different stream).  Only the stream is noted -- the refold
        waits for everything issued on it so far (`wait_stream`): no event record behind every forward (an event record costs the
        stream a few microseconds of dispatch gap: per batch of the inference pipeline, per call of the batch-1 path)."""
        if self.device.type == "cuda":
            cur = torch.cuda.current_stream(self.device)
            self._fold_readers[cur.cuda_stream] = cur

    def forward_waveform(self, frontend: "Frontend", wav: torch.Tensor, want_ranges: bool = False, out=None, feat: Optional[torch.Tensor] = None):
        """Waveforms [B, n_samples] -> (logits, probs): front-end, BN fold when the weights changed since the last fold, and the network
        in ONE host call (`tcr_forward_waveform`).  Bitwise `forward_infer(frontend(wav))`; built for small batches, where the three
        calls of that form cost more than their kernels."""
        if wav.dim() == 3:
            wav = wav[..., 0]
        if wav.dim() != 2 or wav.shape[1] != frontend.cfg.n_samples:
            raise TcrError(f"front-end expects [B, {frontend.cfg.n_samples}] waveforms, got {tuple(wav.shape)}")
        self._check_tensor(wav, "front-end input")
        b = wav.shape[0]
        key = (b, frontend.cfg.n_coef, frontend.cfg.n_frames)
        if feat is None:
            feat = self._wave_feat.get(key)
            if feat is None:
                feat = self._wave_feat[key] = torch.empty((b, frontend.cfg.n_coef, padded_len(frontend.cfg.n_frames)), dtype=torch.float32,
                                                          device=self.device)
        ws = self.workspace(b, False)
        if out is not None:
            logits, probs = out
        else:
            logits = torch.empty((b, self.num_classes), dtype=torch.float32, device=self.device)
            probs = torch.empty_like(logits)
        ranges = torch.empty((b, 2), dtype=torch.float32, device=self.device) if want_ranges else None
        ver = self._weights_version()
        cur = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None
        refold = self._fold_key != ver or self._fold_ss is None or (cur is not None and cur != self._fold_stream)
        if refold and (self._fold_ss is None or any(st != cur for st in self._fold_readers) or (cur is not None and cur != self._fold_stream)):
            ss = self._folded_table()           # allocation / cross-stream ordering: the general path
            refold = False
        else:
            ss = self._fold_ss
        self.lib.check(self.lib.tcr_forward_waveform(C.byref(frontend.cfg), frontend.plan.data_ptr(), self._h, self.params.data_ptr(),
                                                     self.stats.data_ptr(), ss.data_ptr(), int(refold), wav.data_ptr(), b, feat.data_ptr(),
                                                     ws.data_ptr(), ws.numel() * 4, logits.data_ptr(), probs.data_ptr(), _ptr(ranges),
                                                     self._stream()), "tcr_forward_waveform")
        if refold:          # the call refolded in line (on the fold stream, no other readers): other streams order themselves behind THIS fold
            self._fold_key = ver
            self._fold_readers = {}
            self._fold_waited = set()
            if self.device.type == "cuda":
                st = torch.cuda.current_stream(self.device)
                self._fold_event = torch.cuda.Event()
                self._fold_event.record(st)
                self._fold_stream = st.cuda_stream
        self._note_fold_reader()
        return (logits, probs, ranges) if want_ranges else (logits, probs)

    def waveform_call(self, frontend: "Frontend", wav: torch.Tensor, out, feat: Optional[torch.Tensor] = None):
        """A PREPARED `tcr_forward_waveform` call for a fixed set of buffers (the latency regime: one utterance at a time into the same
        tensors): validates, folds BN and allocates once -- through one ordinary `forward_waveform` -- and returns a zero-argument callable
        that issues the single C-ABI call with the pointers bound, on the stream that was current when it was prepared.  What a C / C++
        host of the ABI does per utterance; `forward_waveform` itself spends more time in Python than the GPU spends on the utterance.
        The callable refuses to run once the variables or moving statistics have changed (prepare again) -- the folded table it bound
        belongs to the weights it was prepared for."""
        logits, probs = out
        self.forward_waveform(frontend, wav, out=out, feat=feat)
        if wav.dim() == 3:
            wav = wav[..., 0]
        b = wav.shape[0]
        if feat is None:
            feat = self._wave_feat[(b, frontend.cfg.n_coef, frontend.cfg.n_frames)]
        ws = self.workspace(b, False)
        ss = self._fold_ss
        keep = (frontend, wav, feat, ws, ss, logits, probs, self.params, self.stats)        # (the bound pointers stay alive with the callable)
        args = (C.byref(frontend.cfg), frontend.plan.data_ptr(), self._h, self.params.data_ptr(), self.stats.data_ptr(), ss.data_ptr(), 0,
                wav.data_ptr(), b, feat.data_ptr(), ws.data_ptr(), ws.numel() * 4, logits.data_ptr(), probs.data_ptr(), None, self._stream())
        fn, check = self.lib.tcr_forward_waveform, self.lib.check
        params, stats, kver = self.params, self.stats, (self.params._version, self.stats._version, self._kver)

        def call(_keep=keep):
            if (params._version, stats._version, self._kver) != kver:
                raise TcrError("waveform_call: the variables / moving statistics changed since the call was prepared; prepare it again")
            rc = fn(*args)
            if rc:
                check(rc, "tcr_forward_waveform")
        return call

    def fold_bn(self) -> torch.Tensor:
        """Eval-mode BN of every layer folded to per-channel (scale, shift) constants: the table a frozen export stores."""
        ss = torch.zeros(self.lib.tcr_net_frozen_floats(self._h), dtype=torch.float32, device=self.device)
        self.lib.check(self.lib.tcr_net_fold_bn(self._h, self.params.data_ptr(), self.stats.data_ptr(), ss.data_ptr(), self._stream()),
                       "tcr_net_fold_bn")
        return ss

    def forward_frozen(self, feat: torch.Tensor, frozen_ss: torch.Tensor, want_ranges: bool = False):
        """Eval forward from constants only (conv / fc weights of the arena + the folded table); bitwise forward_infer."""
        self._check_feat(feat)
        self._check_tensor(frozen_ss, "frozen table")
        b = feat.shape[0]
        ws = self.workspace(b, False)
        logits = torch.empty((b, self.num_classes), dtype=torch.float32, device=self.device)
        probs = torch.empty_like(logits)
        ranges = torch.empty((b, 2), dtype=torch.float32, device=self.device) if want_ranges else None
        self.lib.check(self.lib.tcr_net_forward_frozen(self._h, self.params.data_ptr(), frozen_ss.data_ptr(), feat.data_ptr(), b,
                                                       ws.data_ptr(), ws.numel() * 4, logits.data_ptr(), probs.data_ptr(),
                                                       _ptr(ranges), self._stream()), "tcr_net_forward_frozen")
        return (logits, probs, ranges) if want_ranges else (logits, probs)

    def forward_train(self, feat: torch.Tensor, labels: torch.Tensor, keep_prob: float = 1.0, seed: int = 0,
                      sample_offset: int = 0, global_batch: Optional[int] = None, label_smoothing: float = 0.0,
                      sync_hook=None):
        """Train-mode forward.  Returns (logits, probs, loss_sum) where loss_sum is a device scalar holding
        the SUM over this replica's utterances of the per-utterance cross-entropy.
        sync_hook(sums: Tensor) -- optional in-place cross-replica reduction of BN sums (sync BN)."""
        self._check_feat(feat)
        self._check_tensor(labels, "labels")
        b = feat.shape[0]
        gb = int(global_batch or b)
        ws = self.workspace(b, True)
        logits = torch.empty((b, self.num_classes), dtype=torch.float32, device=self.device)
        probs = torch.empty_like(logits)
        loss = torch.empty(2, dtype=torch.float32, device=self.device)      # ([0]: written by the library; no fill kernel in front of the step)
        common = (self._h, self.params.data_ptr(), self.stats.data_ptr(), feat.data_ptr(), labels.data_ptr(), b, gb,
                  float(keep_prob), int(seed), int(sample_offset), float(label_smoothing), ws.data_ptr(), ws.numel() * 4,
                  logits.data_ptr(), probs.data_ptr(), loss.data_ptr())
        if sync_hook is None:
            self.lib.check(self.lib.tcr_net_forward_train(*common, self._stream()), "tcr_net_forward_train")
        elif self.handoff == "unit":      # one hand-off per BN unit (tcr_net_*_stage)
            ns = self.lib.tcr_net_num_stages(self._h, 0)
            for st in range(ns):
                self.lib.check(self.lib.tcr_net_forward_train_stage(*common, st, self._stream()), "tcr_net_forward_train_stage")
                if st < ns - 1:
                    sync_hook(self._stage_sums(0, st, ws, b))
        else:           # cross-replica BN: one hand-off per dependency level (a block's shortcut conv rides with its first conv)
            nl = self.lib.tcr_net_num_levels(self._h, 0)
            for lv in range(nl):
                self.lib.check(self.lib.tcr_net_forward_train_level(*common, lv, self._stream()), "tcr_net_forward_train_level")
                if lv < nl - 1:
                    sync_hook(self._level_sums(0, lv, ws, b))
        self._kver += 1          # the moving statistics changed
        self._last = (feat, b, gb, sync_hook)
        return logits, probs, loss[0]

    def _stage_sums(self, backward: int, stage: int, ws: torch.Tensor, batch: int) -> torch.Tensor:
        ptr, n = C.c_void_p(), C.c_int64()
        self.lib.check(self.lib.tcr_net_stage_sums(self._h, backward, stage, ws.data_ptr(), batch, C.byref(ptr), C.byref(n)),
                       "tcr_net_stage_sums")
        off = (ptr.value - ws.data_ptr()) // 4
        return ws[off:off + 2 * n.value].view(torch.float64)        # 2*C float64 sums living inside the f32 workspace

    def _level_sums(self, backward: int, level: int, ws: torch.Tensor, batch: int) -> torch.Tensor:
        ptr, n = C.c_void_p(), C.c_int64()
        self.lib.check(self.lib.tcr_net_level_sums(self._h, backward, level, ws.data_ptr(), batch, C.byref(ptr), C.byref(n)),
                       "tcr_net_level_sums")
        off = (ptr.value - ws.data_ptr()) // 4
        return ws[off:off + 2 * n.value].view(torch.float64)        # the level's float64 sums (one contiguous range)

    def backward(self) -> torch.Tensor:
        """Gradient of the mean cross-entropy wrt every trainable, into self.grads (L2 excluded)."""
        feat, b, gb, sync_hook = self._last
        ws = self.workspace(b, True)
        if sync_hook is None:
            self.lib.check(self.lib.tcr_net_backward(self._h, self.params.data_ptr(), feat.data_ptr(), b, ws.data_ptr(),
                                                     ws.numel() * 4, self.grads.data_ptr(), self._stream()), "tcr_net_backward")
        elif self.handoff == "unit":
            ns = self.lib.tcr_net_num_stages(self._h, 1)
            for st in range(ns):
                self.lib.check(self.lib.tcr_net_backward_stage(self._h, self.params.data_ptr(), feat.data_ptr(), b, gb,
                                                               ws.data_ptr(), ws.numel() * 4, self.grads.data_ptr(), st,
                                                               self._stream()), "tcr_net_backward_stage")
                if st < ns - 1:
                    sync_hook(self._stage_sums(1, st, ws, b))
        else:
            nl = self.lib.tcr_net_num_levels(self._h, 1)
            for lv in range(nl):
                self.lib.check(self.lib.tcr_net_backward_level(self._h, self.params.data_ptr(), feat.data_ptr(), b, gb,
                                                               ws.data_ptr(), ws.numel() * 4, self.grads.data_ptr(), lv,
                                                               self._stream()), "tcr_net_backward_level")
                if lv < nl - 1:
                    sync_hook(self._level_sums(1, lv, ws, b))
        return self.grads

    def _slot(self, name: str) -> torch.Tensor:
        if name not in self.slots:
            self.slots[name] = torch.zeros_like(self.params)
        return self.slots[name]

    def slot_arena(self, name: str) -> torch.Tensor:
        """Arena-shaped optimiser slot under its TF name, created with TF's initial value: `RMSProp` (the rms accumulator)
        starts at one, `ExponentialMovingAverage` as a copy of the variables, every other slot at zero."""
        if name not in self.slots:
            if name == "RMSProp":
                self.slots[name] = torch.ones_like(self.params)
            elif name == "ExponentialMovingAverage":
                self.slots[name] = self.params.clone()
            else:
                self.slots[name] = torch.zeros_like(self.params)
        return self.slots[name]

    def ema_init(self):
        self.slots["ExponentialMovingAverage"] = self.params.clone()

    def sgd_momentum_step(self, lr: float, momentum: float = 0.9, weight_decay: float = 0.0, grad_scale: float = 1.0):
        m = self._slot("Momentum")
        self.lib.check(self.lib.tcr_sgd_momentum_step(self.params.data_ptr(), self.grads.data_ptr(), m.data_ptr(), self.n_param,
                                                      self.n_decay, float(lr), float(momentum), float(weight_decay),
                                                      float(grad_scale), self._stream()), "tcr_sgd_momentum_step")
        self._kver += 1

    def adam_step(self, lr: float, step: int, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                  weight_decay: float = 0.0, grad_scale: float = 1.0):
        m, v = self._slot("Adam"), self._slot("Adam_1")
        self.lib.check(self.lib.tcr_adam_step(self.params.data_ptr(), self.grads.data_ptr(), m.data_ptr(), v.data_ptr(),
                                              self.n_param, self.n_decay, float(lr), float(beta1), float(beta2), float(eps),
                                              int(step), float(weight_decay), float(grad_scale), self._stream()), "tcr_adam_step")
        self._kver += 1

    def rmsprop_step(self, lr: float, decay: float = 0.9, momentum: float = 0.0, eps: float = 1e-10, weight_decay: float = 0.0,
                     grad_scale: float = 1.0):
        """tf.train.RMSPropOptimizer; slot names as in TF (`RMSProp` = rms, initialised to one; `RMSProp_1` = momentum)."""
        ms, mom = self.slot_arena("RMSProp"), self.slot_arena("RMSProp_1")
        self.lib.check(self.lib.tcr_rmsprop_step(self.params.data_ptr(), self.grads.data_ptr(), ms.data_ptr(), mom.data_ptr(),
                                                 self.n_param, self.n_decay, float(lr), float(decay), float(momentum), float(eps),
                                                 float(weight_decay), float(grad_scale), self._stream()), "tcr_rmsprop_step")
        self._kver += 1

    def ema_step(self, decay: float):
        """tf.train.ExponentialMovingAverage(decay).apply(trainables): the shadow arena starts as a copy of the variables'
        initial values (`ema_init()`, called when the train op is built)."""
        sh = self.slot_arena("ExponentialMovingAverage")
        self.lib.check(self.lib.tcr_ema_step(sh.data_ptr(), self.params.data_ptr(), self.n_param, float(decay), self._stream()),
                       "tcr_ema_step")

    def l2_loss(self, weight_decay: float) -> torch.Tensor:
        out = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.lib.check(self.lib.tcr_l2_loss(self.params.data_ptr(), self.n_decay, float(weight_decay), out.data_ptr(),
                                            self._stream()), "tcr_l2_loss")
        return out[0]


class DSCNN(_Base):
    """DS-CNN S / M / L: eval forward, train forward / backward, Adam (audio_nets/ds_cnn.py:19-118 of the reference)."""

    NET_DEFS = {            # depth, separable blocks, conv_1 stride, conv_ds_1 stride  (ds_cnn.py:19-43)
        "S": (64, 4, (2, 2), (1, 1)),
        "M": (172, 4, (2, 1), (2, 2)),
        "L": (276, 5, (2, 1), (2, 2)),
    }

    def __init__(self, size: str, h_in: int, w_in: int, num_classes: int, bn_decay: float = 0.96, bn_eps: float = 0.001,
                 lib: Optional[_lib.Library] = None, device=None):
        self.lib, self.device = _resolve(lib, device)
        depth, nsep, s1, sds = self.NET_DEFS[size]
        cfg = DSCNNCfg(int(h_in), int(w_in), int(num_classes), depth, nsep, 10, 4, s1[0], s1[1], sds[0], sds[1],
                       float(bn_decay), float(bn_eps))
        handle = C.c_void_p()
        self.lib.check(self.lib.tcr_dscnn_create(C.byref(cfg), C.byref(handle)), "tcr_dscnn_create")
        self.cfg, self._h, self.size = cfg, handle, size
        self.num_classes, self.h_in, self.w_in = int(num_classes), int(h_in), int(w_in)
        self.n_param = self.lib.tcr_dscnn_param_floats(handle)
        self.n_decay = self.n_param     # the reference's name filter excludes only "BatchNorm"; these scopes are "batch_norm"
        self.n_stat = self.lib.tcr_dscnn_stat_floats(handle)
        self.tensors: Dict[str, TensorInfo] = {}
        for i in range(self.lib.tcr_dscnn_num_tensors(handle)):
            ti = TensorInfo()
            self.lib.check(self.lib.tcr_dscnn_tensor_info(handle, i, C.byref(ti)), "tcr_dscnn_tensor_info")
            self.tensors[ti.name.decode()] = ti
        self.params = torch.zeros(self.n_param, dtype=torch.float32, device=self.device)
        self.stats = torch.zeros(self.n_stat, dtype=torch.float32, device=self.device)
        self._grads_buf = torch.zeros(self.n_param + DP_TAIL, dtype=torch.float32, device=self.device)     # arena + the data-parallel tail (parallel.py)
        self.grads = self._grads_buf[:self.n_param]
        self.slots: Dict[str, torch.Tensor] = {}
        self._ws: Dict[int, torch.Tensor] = {}
        self._train_ws: Dict[int, torch.Tensor] = {}
        for n, ti in self.tensors.items():
            if ti.kind == 4:
                self._view(n).fill_(1.0)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.tcr_dscnn_destroy(self._h)
                self._h = None
        except Exception:
            pass

    _view = TCResNet._view
    grad_view = TCResNet.grad_view
    trainable_names = TCResNet.trainable_names
    state_dict = TCResNet.state_dict
    load_state_dict = TCResNet.load_state_dict
    _slot = TCResNet._slot
    adam_step = TCResNet.adam_step
    sgd_momentum_step = TCResNet.sgd_momentum_step
    rmsprop_step = TCResNet.rmsprop_step
    ema_step = TCResNet.ema_step
    slot_arena = TCResNet.slot_arena
    ema_init = TCResNet.ema_init

    def total_params(self) -> int:
        return sum(int(ti.size) for ti in self.tensors.values() if ti.arena == 0)

    def init_xavier(self, seed: int = 0):
        gen = torch.Generator().manual_seed(int(seed))
        for n, ti in self.tensors.items():
            if ti.kind != 0:
                continue
            shape = tuple(ti.shape[i] for i in range(ti.rank))
            if ti.rank == 4:
                kh, kw, cin, cout = shape
                fan_in, fan_out = kh * kw * cin, kh * kw * cout
            else:
                fan_in, fan_out = shape
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            self._view(n).copy_(((torch.rand(shape, generator=gen) * 2.0 - 1.0) * lim).to(self.device))

    def _check_feat(self, feat: torch.Tensor):
        self._check_tensor(feat, "features")
        want = (self.w_in, padded_len(self.h_in))
        if feat.dim() != 3 or tuple(feat.shape[1:]) != want:
            raise TcrError(f"features must be planar [B, {want[0]}, {want[1]}], got {tuple(feat.shape)}")

    def train_workspace(self, batch: int) -> torch.Tensor:
        ws = self._train_ws.get(batch)
        if ws is None:
            nbytes = self.lib.tcr_dscnn_train_workspace_bytes(self._h, batch)
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=self.device)
            self._train_ws[batch] = ws
        return ws

    def forward_train(self, feat: torch.Tensor, labels: torch.Tensor, global_batch: Optional[int] = None,
                      label_smoothing: float = 0.0, **_unused):
        """Train-mode forward (batch statistics, moving averages updated).  Returns (logits, probs, loss_sum) like
        TCResNet.forward_train; the graph applies no dropout (ds_cnn.py:89-101), so keep_prob / seed are ignored."""
        sync_hook = _unused.get("sync_hook")
        self._check_feat(feat)
        self._check_tensor(labels, "labels")
        b = feat.shape[0]
        gb = int(global_batch or b)
        ws = self.train_workspace(b)
        logits = torch.empty((b, self.num_classes), dtype=torch.float32, device=self.device)
        probs = torch.empty_like(logits)
        loss = torch.empty(2, dtype=torch.float32, device=self.device)      # ([0]: written by the library; no fill kernel in front of the step)
        common = (self._h, self.params.data_ptr(), self.stats.data_ptr(), feat.data_ptr(), labels.data_ptr(), b, gb, float(label_smoothing),
                  ws.data_ptr(), ws.numel() * 4, logits.data_ptr(), probs.data_ptr(), loss.data_ptr())
        if sync_hook is None:
            self.lib.check(self.lib.tcr_dscnn_forward_train(*common, self._stream()), "tcr_dscnn_forward_train")
        else:               # cross-replica BN: the host all-reduces each unit's float64 sums between stages
            ns = self.lib.tcr_dscnn_num_stages(self._h)
            for st in range(ns):
                self.lib.check(self.lib.tcr_dscnn_forward_train_stage(*common, st, self._stream()), "tcr_dscnn_forward_train_stage")
                if st < ns - 1:
                    sync_hook(self._stage_sums(0, st, ws, b))
        self._last = (feat, b, gb, sync_hook)
        return logits, probs, loss[0]

    def unit_output(self, unit: int, batch: int) -> torch.Tensor:
        """[B, C, P] view of BN unit `unit`'s post-ReLU activation left in the training workspace by the last forward_train at `batch`."""
        off, c, pos, pad = C.c_int64(), C.c_int(), C.c_int(), C.c_int()
        self.lib.check(self.lib.tcr_dscnn_unit_output(self._h, int(unit), int(batch), C.byref(off), C.byref(c), C.byref(pos), C.byref(pad)),
                       "tcr_dscnn_unit_output")
        ws = self.train_workspace(batch)
        # the default training path keeps only the unit's raw conv output (TCR_TUNE_DS_TRAIN): normalise it into the activation slot
        self.lib.check(self.lib.tcr_dscnn_materialize_unit(self._h, int(unit), int(batch), ws.data_ptr(), ws.numel() * 4, self._stream()),
                       "tcr_dscnn_materialize_unit")
        return ws[off.value: off.value + batch * c.value * pad.value].view(batch, c.value, pad.value)[:, :, HALO:HALO + pos.value]

    def _stage_sums(self, backward: int, stage: int, ws: torch.Tensor, batch: int) -> torch.Tensor:
        ptr, n = C.c_void_p(), C.c_int64()
        self.lib.check(self.lib.tcr_dscnn_stage_sums(self._h, backward, stage, ws.data_ptr(), batch, C.byref(ptr), C.byref(n)),
                       "tcr_dscnn_stage_sums")
        off = (ptr.value - ws.data_ptr()) // 4
        return ws[off:off + 2 * n.value].view(torch.float64)

    def backward(self) -> torch.Tensor:
        """Gradient of the mean cross-entropy wrt every trainable, into self.grads."""
        feat, b, gb, sync_hook = self._last
        ws = self.train_workspace(b)
        if sync_hook is None:
            self.lib.check(self.lib.tcr_dscnn_backward(self._h, self.params.data_ptr(), feat.data_ptr(), b, ws.data_ptr(),
                                                       ws.numel() * 4, self.grads.data_ptr(), self._stream()), "tcr_dscnn_backward")
        else:
            ns = self.lib.tcr_dscnn_num_stages(self._h)
            for st in range(ns):
                self.lib.check(self.lib.tcr_dscnn_backward_stage(self._h, self.params.data_ptr(), feat.data_ptr(), b, gb, ws.data_ptr(),
                                                                 ws.numel() * 4, self.grads.data_ptr(), st, self._stream()),
                               "tcr_dscnn_backward_stage")
                if st < ns - 1:
                    sync_hook(self._stage_sums(1, st, ws, b))
        return self.grads

    def forward_infer(self, feat: torch.Tensor):
        self._check_feat(feat)
        b = feat.shape[0]
        ws = self._ws.get(b)
        if ws is None:
            ws = torch.empty(self.lib.tcr_dscnn_workspace_bytes(self._h, b) // 4, dtype=torch.float32, device=self.device)
            self._ws[b] = ws
        logits = torch.empty((b, self.num_classes), dtype=torch.float32, device=self.device)
        probs = torch.empty_like(logits)
        self.lib.check(self.lib.tcr_dscnn_forward_infer(self._h, self.params.data_ptr(), self.stats.data_ptr(), feat.data_ptr(), b,
                                                        ws.data_ptr(), ws.numel() * 4, logits.data_ptr(), probs.data_ptr(),
                                                        self._stream()), "tcr_dscnn_forward_infer")
        return logits, probs

    def l2_loss(self, weight_decay: float) -> torch.Tensor:
        out = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.lib.check(self.lib.tcr_l2_loss(self.params.data_ptr(), self.n_decay, float(weight_decay), out.data_ptr(),
                                            self._stream()), "tcr_l2_loss")
        return out[0]


class Graph2D(_Base):
    """Generic 2-D layer graph on the HIP kernels (C ABI tcr_g2d_*): the remaining model families of the reference's factory
    -- ResNet2D8(/Pool), Res8/15(/Narrow), KWSModel (audio_nets/tc_resnet.py:73-99, res.py:6-123, kws.py:15-63).  The topology
    is described node by node by the builders in tcresnet_amd/audio_nets, the way the reference's Python builds its TF graph;
    variables are created under the TF names given there.  Input: the [T x F] single-channel feature plane ([N, T, F, 1])."""

    def __init__(self, scope: str, h: int, w: int, c: int = 1, lib: Optional[_lib.Library] = None, device=None):
        self.lib, self.device = _resolve(lib, device)
        handle = C.c_void_p()
        self.lib.check(self.lib.tcr_g2d_create(scope.encode(), int(h), int(w), int(c), C.byref(handle)), "tcr_g2d_create")
        self._h, self.scope = handle, scope
        self.h_in, self.w_in, self.c_in = int(h), int(w), int(c)
        self.initializers: Dict[str, object] = {}           # variable name -> "xavier" | ("truncated_normal", stddev) | "zeros"
        self.relu_nodes: List[int] = []                      # nodes whose output went through a ReLU, in build order
        self.dropout_nodes: List[int] = []                  # node ids of the dropout layers (each mask is keyed by its node id)
        self.constants: Dict[str, np.ndarray] = {}          # non-trainable variables of the reference graph that stay constant (checkpoint completeness)
        self.tf_shapes: Dict[str, Tuple[int, ...]] = {}      # variables whose TF shape differs from the kernel's [kh, kw, cin, cout] (matmul weights: [K, N])
        self.finalized = False
        self._ws: Dict[Tuple[int, int], torch.Tensor] = {}
        self.slots: Dict[str, torch.Tensor] = {}

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.tcr_g2d_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- topology ---------------------------------------------------------------------------------------------------
    def _node(self, rc: int, what: str) -> int:
        if rc < 0:
            self.lib.check(rc, what)
        return rc

    def conv(self, inp: int, kernel, cout: int, weights_name: str, stride=(1, 1), rate=(1, 1), padding: str = "SAME", relu: bool = False,
             biases_name: Optional[str] = None, init="xavier", tf_shape: Optional[Sequence[int]] = None) -> int:
        if tf_shape is not None:
            self.tf_shapes[weights_name] = tuple(int(d) for d in tf_shape)
        kh, kw = (kernel, kernel) if isinstance(kernel, int) else kernel
        sh, sw = (stride, stride) if isinstance(stride, int) else stride
        dh, dw = (rate, rate) if isinstance(rate, int) else rate
        if padding not in ("SAME", "VALID"):
            raise ValueError(padding)
        self.initializers[weights_name] = init
        if biases_name:
            self.initializers[biases_name] = "zeros"
        return self._relu(relu, self._node(self.lib.tcr_g2d_conv(self._h, inp, kh, kw, int(cout), sh, sw, dh, dw, int(padding == "VALID"),
                                                                 int(relu), weights_name.encode(), (biases_name or "").encode()), "tcr_g2d_conv"))

    def batch_norm(self, inp: int, prefix: str, center: bool = True, scale: bool = True, relu: bool = False, decay: float = 0.997,
                   eps: float = 0.001) -> int:
        return self._relu(relu, self._node(self.lib.tcr_g2d_batch_norm(self._h, inp, int(center), int(scale), int(relu), float(decay),
                                                                       float(eps), prefix.encode()), "tcr_g2d_batch_norm"))

    def time_filter(self, inp: int, weights_name: str, init="xavier") -> int:
        """One filter of the plane's length per channel -> [C, 1, 1] (SVDF time filters, kws.py:604-612); variable [C, plane]."""
        self.initializers[weights_name] = init
        return self._node(self.lib.tcr_g2d_time_filter(self._h, inp, weights_name.encode()), "tcr_g2d_time_filter")

    def group_sum(self, inp: int, group: int, relu: bool = False, biases_name: Optional[str] = None) -> int:
        """Sum of `group` consecutive channels of a 1 x 1 node (+ bias) (+ ReLU) (SVDF rank sum, kws.py:613-628)."""
        if biases_name:
            self.initializers[biases_name] = "zeros"
        return self._relu(relu, self._node(self.lib.tcr_g2d_group_sum(self._h, inp, int(group), int(relu), (biases_name or "").encode()),
                                           "tcr_g2d_group_sum"))

    def pool(self, inp: int, kind: str, kernel=None, stride=(1, 1), padding: str = "VALID") -> int:
        kh, kw = (0, 0) if kernel is None else ((kernel, kernel) if isinstance(kernel, int) else kernel)
        sh, sw = (stride, stride) if isinstance(stride, int) else stride
        return self._node(self.lib.tcr_g2d_pool(self._h, inp, int(kind == "max"), kh, kw, sh, sw, int(padding == "VALID")), "tcr_g2d_pool")

    def add(self, a: int, b: int, relu: bool = False) -> int:
        return self._relu(relu, self._node(self.lib.tcr_g2d_add(self._h, a, b, int(relu)), "tcr_g2d_add"))

    def dropout(self, inp: int, keep_prob: float) -> int:
        node = self._node(self.lib.tcr_g2d_dropout(self._h, inp, float(keep_prob)), "tcr_g2d_dropout")
        self.dropout_nodes.append(node)
        return node

    def _relu(self, relu: bool, node: int) -> int:
        if relu:
            self.relu_nodes.append(node)
        return node

    def node_output(self, node: int, batch: int, train: bool) -> torch.Tensor:
        """[B, C, H, W] view of a node's activation after a forward call at (batch, train) -- the reference's `endpoints`."""
        off, plane, halo = C.c_int64(), C.c_int64(), C.c_int()
        self.lib.check(self.lib.tcr_g2d_node_output(self._h, int(node), int(batch), int(train), C.byref(off), C.byref(plane), C.byref(halo)),
                       "tcr_g2d_node_output")
        c, h, w = self.shape(node)
        buf = self.workspace(batch, train)[off.value: off.value + batch * c * plane.value].view(batch, c, plane.value)
        return buf[:, :, halo.value: halo.value + h * w].reshape(batch, c, h, w)

    def shape(self, node: int) -> Tuple[int, int, int]:
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        self.lib.check(self.lib.tcr_g2d_node_shape(self._h, node, C.byref(c), C.byref(h), C.byref(w)), "tcr_g2d_node_shape")
        return c.value, h.value, w.value

    def finalize(self, logits: int):
        self.lib.check(self.lib.tcr_g2d_finalize(self._h, logits), "tcr_g2d_finalize")
        self.finalized = True
        self.num_classes = self.lib.tcr_g2d_num_classes(self._h)
        self.n_param = self.lib.tcr_g2d_param_floats(self._h)
        self.n_decay = self.lib.tcr_g2d_decay_floats(self._h)
        self.n_stat = self.lib.tcr_g2d_stat_floats(self._h)
        self.tensors: Dict[str, TensorInfo] = {}
        for i in range(self.lib.tcr_g2d_num_tensors(self._h)):
            ti = TensorInfo()
            self.lib.check(self.lib.tcr_g2d_tensor_info(self._h, i, C.byref(ti)), "tcr_g2d_tensor_info")
            self.tensors[ti.name.decode()] = ti
        self.params = torch.zeros(self.n_param, dtype=torch.float32, device=self.device)
        self.stats = torch.zeros(self.n_stat, dtype=torch.float32, device=self.device)
        self._grads_buf = torch.zeros(self.n_param + DP_TAIL, dtype=torch.float32, device=self.device)     # arena + the data-parallel tail (parallel.py)
        self.grads = self._grads_buf[:self.n_param]
        self.init_variables(0)

    # ---- variables ---------------------------------------------------------------------------------------------------
    _view = TCResNet._view
    grad_view = TCResNet.grad_view
    trainable_names = TCResNet.trainable_names
    total_params = TCResNet.total_params
    _slot = TCResNet._slot
    slot_arena = TCResNet.slot_arena
    ema_init = TCResNet.ema_init
    ema_step = TCResNet.ema_step
    adam_step = TCResNet.adam_step
    sgd_momentum_step = TCResNet.sgd_momentum_step
    rmsprop_step = TCResNet.rmsprop_step
    l2_loss = TCResNet.l2_loss

    def state_dict(self) -> Dict[str, np.ndarray]:
        """TF variable name -> array in the shape the reference's graph declares (matmul weights [K, N], not [h, w, c, N])."""
        sd = {n: self._view(n).detach().cpu().numpy().copy().reshape(self.tf_shape(n)) for n in self.tensors}
        sd.update({n: v.copy() for n, v in getattr(self, "constants", {}).items()})     # non-trainable graph variables that never change here
        return sd

    def load_state_dict(self, sd: Dict[str, np.ndarray], strict: bool = True):
        for n, ti in self.tensors.items():
            if n not in sd:
                if strict:
                    raise KeyError(n)
                continue
            v = torch.as_tensor(np.asarray(sd[n], dtype=np.float32))
            shape = tuple(ti.shape[i] for i in range(ti.rank))
            if v.numel() != int(ti.size):
                raise TcrError(f"{n}: shape {tuple(v.shape)} != {shape}")
            self._view(n).copy_(v.reshape(shape).to(self.device))       # (a [K, N] matmul weight IS the [h, w, c, N] conv weight)

    def init_variables(self, seed: int = 0):
        """Weights by the initializer the reference gives each variable (xavier uniform under the slim arg scopes,
        truncated normal in kws.py), biases / beta / moving_mean 0, gamma / moving_variance 1."""
        gen = torch.Generator().manual_seed(int(seed))
        for n, ti in self.tensors.items():
            shape = tuple(ti.shape[i] for i in range(ti.rank))
            if ti.kind == 0:
                init = self.initializers.get(n, "xavier")
                if init == "xavier":
                    kh, kw, cin, cout = shape if len(shape) == 4 else (1, 1) + tuple(shape)
                    lim = math.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
                    w = (torch.rand(shape, generator=gen, dtype=torch.float32) * 2.0 - 1.0) * lim
                else:                       # ("truncated_normal", stddev): redraw beyond two standard deviations
                    std = float(init[1])
                    w = torch.randn(shape, generator=gen, dtype=torch.float32)
                    bad = w.abs() > 2.0
                    while bool(bad.any()):
                        w[bad] = torch.randn(int(bad.sum()), generator=gen, dtype=torch.float32)
                        bad = w.abs() > 2.0
                    w = w * std
                self._view(n).copy_(w.to(self.device))
            elif ti.kind in (1, 4):
                self._view(n).fill_(1.0)
            else:
                self._view(n).zero_()

    # ---- compute ------------------------------------------------------------------------------------------------------
    def workspace(self, batch: int, train: bool) -> torch.Tensor:
        key = (int(batch), int(train))
        ws = self._ws.get(key)
        if ws is None:
            ws = torch.empty(self.lib.tcr_g2d_workspace_bytes(self._h, batch, int(train)) // 4, dtype=torch.float32, device=self.device)
            self._ws[key] = ws
        return ws

    def input_from_features(self, feat: torch.Tensor) -> torch.Tensor:
        """Front-end planar features [B, F, T + 2*HALO] -> the network's input plane [B, 1, T*F + 2*HALO] ([N, T, F, 1])."""
        self._check_tensor(feat, "features")
        b = feat.shape[0]
        if self.c_in != 1 or tuple(feat.shape[1:]) != (self.w_in, padded_len(self.h_in)):
            raise TcrError(f"features must be planar [B, {self.w_in}, {padded_len(self.h_in)}], got {tuple(feat.shape)}")
        x = torch.empty((b, 1, self.h_in * self.w_in + 2 * HALO), dtype=torch.float32, device=self.device)
        self.lib.check(self.lib.tcr_g2d_input_from_features(feat.data_ptr(), b, self.h_in, self.w_in, x.data_ptr(), self._stream()),
                       "tcr_g2d_input_from_features")
        return x

    def forward_infer(self, feat: torch.Tensor):
        x = self.input_from_features(feat)
        b = x.shape[0]
        ws = self.workspace(b, False)
        logits = torch.empty((b, self.num_classes), dtype=torch.float32, device=self.device)
        probs = torch.empty_like(logits)
        self.lib.check(self.lib.tcr_g2d_forward_infer(self._h, self.params.data_ptr(), self.stats.data_ptr(), x.data_ptr(), b, ws.data_ptr(),
                                                      ws.numel() * 4, logits.data_ptr(), probs.data_ptr(), self._stream()), "tcr_g2d_forward_infer")
        return logits, probs

    def forward_train(self, feat: torch.Tensor, labels: torch.Tensor, keep_prob: float = 1.0, seed: int = 0, sample_offset: int = 0,
                      global_batch: Optional[int] = None, label_smoothing: float = 0.0, sync_hook=None):
        """Train-mode forward; returns (logits, probs, loss_sum) like TCResNet.forward_train.  Dropout probabilities belong to the
        graph (they are fixed where the reference builds it), so `keep_prob` is not read here."""
        self._check_tensor(labels, "labels")
        x = self.input_from_features(feat)
        b = x.shape[0]
        gb = int(global_batch or b)
        ws = self.workspace(b, True)
        logits = torch.empty((b, self.num_classes), dtype=torch.float32, device=self.device)
        probs = torch.empty_like(logits)
        loss = torch.empty(2, dtype=torch.float32, device=self.device)      # ([0]: written by the library; no fill kernel in front of the step)
        common = (self._h, self.params.data_ptr(), self.stats.data_ptr(), x.data_ptr(), labels.data_ptr(), b, gb, int(seed), int(sample_offset),
                  float(label_smoothing), ws.data_ptr(), ws.numel() * 4, logits.data_ptr(), probs.data_ptr(), loss.data_ptr())
        if sync_hook is None:
            self.lib.check(self.lib.tcr_g2d_forward_train(*common, self._stream()), "tcr_g2d_forward_train")
        else:           # cross-replica BN: the forward stops behind every BN node's statistics (tcr_g2d_*_stage)
            ns = self.lib.tcr_g2d_num_stages(self._h)
            for st in range(ns):
                self.lib.check(self.lib.tcr_g2d_forward_train_stage(*common, st, self._stream()), "tcr_g2d_forward_train_stage")
                if st < ns - 1:
                    sync_hook(self._stage_sums(0, st, ws, b))
        self._last = (x, b, int(seed), int(sample_offset), gb, sync_hook)
        return logits, probs, loss[0]

    def _stage_sums(self, backward: int, stage: int, ws: torch.Tensor, batch: int) -> torch.Tensor:
        ptr, n = C.c_void_p(), C.c_int64()
        self.lib.check(self.lib.tcr_g2d_stage_sums(self._h, backward, stage, ws.data_ptr(), batch, C.byref(ptr), C.byref(n)), "tcr_g2d_stage_sums")
        off = (ptr.value - ws.data_ptr()) // 4
        return ws[off:off + 2 * n.value].view(torch.float64)

    def backward(self) -> torch.Tensor:
        x, b, seed, off, gb, sync_hook = self._last
        ws = self.workspace(b, True)
        if sync_hook is None:
            self.lib.check(self.lib.tcr_g2d_backward(self._h, self.params.data_ptr(), x.data_ptr(), b, seed, off, ws.data_ptr(), ws.numel() * 4,
                                                     self.grads.data_ptr(), self._stream()), "tcr_g2d_backward")
        else:
            ns = self.lib.tcr_g2d_num_stages(self._h)
            for st in range(ns):
                self.lib.check(self.lib.tcr_g2d_backward_stage(self._h, self.params.data_ptr(), x.data_ptr(), b, gb, seed, off, ws.data_ptr(),
                                                               ws.numel() * 4, self.grads.data_ptr(), st, self._stream()), "tcr_g2d_backward_stage")
                if st < ns - 1:
                    sync_hook(self._stage_sums(1, st, ws, b))
        return self.grads
