"""TEST INFRASTRUCTURE ONLY -- independent PyTorch-CPU restatement of the reference hot path.

PARITY UNPINNED (see oracle/__init__.py).  This is the *second* implementation used to
cross-check oracle/numpy_ref.py (different code path: torch.fft, F.conv1d with explicit
asymmetric F.pad, F.batch_norm, autograd for every gradient) and the CPU baseline that
bench.py times on the GPU box's host cores ("kind": "port").  It never runs on the GPU and
is never imported by the product package.

Reference call sites: datasets/preprocessors.py:64-96,183-194; audio_nets/tc_resnet.py:6-54,
102-123; factory/audio_nets.py:147-183; helper/trainer.py:188-190.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import numpy_ref as R


def mfcc(wav: torch.Tensor, cfg: R.FrontendCfg) -> torch.Tensor:
    """[B, n] -> [B, T, num_mfccs] (dtype of wav)."""
    dt = wav.dtype
    window = torch.from_numpy(R.hann_periodic(cfg.win, np.float64)).to(dt)
    # torch.stft with center=False frames exactly like signal.frame(pad_end=False); the window is
    # zero-padded to n_fft *centred* by torch, so frame + rfft explicitly instead.
    frames = wav.unfold(-1, cfg.win, cfg.hop) * window
    spec = torch.fft.rfft(frames, n=cfg.nfft, dim=-1)
    power = spec.real ** 2 + spec.imag ** 2
    mel_w = torch.from_numpy(R.linear_to_mel_weight_matrix(cfg.num_mel_bins, cfg.n_bins, cfg.sample_rate,
                                                            cfg.lower_edge_hertz, cfg.upper_edge_hertz)).to(dt)
    logmel = torch.log(power @ mel_w + R.LOG_OFFSET)
    dct = torch.from_numpy(R.dct2_matrix(cfg.num_mel_bins, cfg.num_mfccs)).to(dt)
    return logmel @ dct


def _conv(x: torch.Tensor, w: torch.Tensor, stride: int) -> torch.Tensor:
    """x [N, C, L]; w TF layout [k, Cin, Cout]; TF SAME padding (extra on the high side)."""
    k = w.shape[0]
    _, lo, hi = R.same_pad(x.shape[-1], k, stride)
    return F.conv1d(F.pad(x, (lo, hi)), w.permute(2, 1, 0).contiguous(), stride=stride)


def forward(arch: R.TCResNetArch, params: Dict[str, torch.Tensor], stats: Dict[str, torch.Tensor],
            x: torch.Tensor, is_training: bool = False, keep_prob: float = 1.0,
            dropout_mask: Optional[torch.Tensor] = None, decay: float = R.BN_DECAY_TCRESNET):
    """x: [N, T, F].  Returns dict(logits, probs, ranges, new_stats).  Autograd-friendly."""
    new_stats = dict(stats)

    def cba(c: R.ConvSpec, h):
        base = f"{arch.scope}/{c.name}"
        y = _conv(h, params[base + "/weights"], c.stride)
        if c.bn:
            g, b = params[base + "/BatchNorm/gamma"], params[base + "/BatchNorm/beta"]
            mm, mv = stats[base + "/BatchNorm/moving_mean"], stats[base + "/BatchNorm/moving_variance"]
            if is_training:
                rm, rv = mm.clone(), mv.clone()
                # F.batch_norm updates running stats as (1-m)*r + m*stat with the unbiased variance.
                y = F.batch_norm(y, rm, rv, g, b, True, 1.0 - decay, R.BN_EPS)
                new_stats[base + "/BatchNorm/moving_mean"] = rm
                new_stats[base + "/BatchNorm/moving_variance"] = rv
            else:
                y = F.batch_norm(y, mm, mv, g, b, False, 0.0, R.BN_EPS)
        return F.relu(y) if c.relu else y

    net = cba(arch.conv0, x.transpose(1, 2))
    for b in arch.blocks:
        shortcut = cba(b.down, net) if b.down is not None else net
        h = cba(b.conv_b, cba(b.conv_a, net))
        net = F.relu(h + shortcut)
    pooled = net.mean(dim=-1)
    if is_training and dropout_mask is not None:
        pooled = (pooled / keep_prob) * dropout_mask
    logits = pooled @ params[f"{arch.scope}/fc/weights"][0]
    ranges = torch.sigmoid(pooled @ params[f"{arch.scope}/fc2/weights"][0])
    return {"logits": logits, "probs": F.softmax(logits, dim=-1), "ranges": ranges, "new_stats": new_stats}


def total_loss(logits, labels_onehot, params, weight_decay: float):
    model = torch.mean(-(labels_onehot * F.log_softmax(logits, dim=-1)).sum(dim=-1))
    l2 = weight_decay * sum(0.5 * (v ** 2).sum() for k, v in params.items() if R.is_l2_param(k))
    return model + l2, model


def grads(arch, params_np: Dict[str, np.ndarray], stats_np, x_np, labels_np, weight_decay, keep_prob=1.0,
          mask_np=None, dtype=torch.float64):
    """Autograd gradients of the total loss wrt every trainable (independent of numpy_ref.backward)."""
    params = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in params_np.items()}
    stats = {k: torch.tensor(v, dtype=dtype) for k, v in stats_np.items()}
    mask = None if mask_np is None else torch.tensor(mask_np, dtype=dtype)
    out = forward(arch, params, stats, torch.tensor(x_np, dtype=dtype), True, keep_prob, mask)
    tot, model = total_loss(out["logits"], torch.tensor(labels_np, dtype=dtype), params, weight_decay)
    tot.backward()
    g = {k: (v.grad.detach().numpy() if v.grad is not None else np.zeros_like(params_np[k])) for k, v in params.items()}
    return g, float(tot.detach()), float(model.detach()), {k: v.detach().numpy() for k, v in out["new_stats"].items()}


class CpuBaseline:
    """Waveform -> softmax on the host cores in float32 (the bench's cpu_baseline leg)."""

    def __init__(self, arch: R.TCResNetArch, cfg: R.FrontendCfg, params_np, stats_np, threads: Optional[int] = None):
        if threads:
            torch.set_num_threads(threads)
        self.arch, self.cfg = arch, cfg
        self.params = {k: torch.tensor(v, dtype=torch.float32) for k, v in params_np.items()}
        self.stats = {k: torch.tensor(v, dtype=torch.float32) for k, v in stats_np.items()}

    @torch.no_grad()
    def infer(self, wav: torch.Tensor) -> torch.Tensor:
        return forward(self.arch, self.params, self.stats, mfcc(wav, self.cfg), False)["probs"]

    def train_step(self, wav: torch.Tensor, labels: torch.Tensor, lr: float, mu: float, weight_decay: float) -> float:
        """One float32 training step on the host cores: MFCC, train-mode forward, autograd backward, momentum update, moving
        statistics (the bench's CPU train-step leg)."""
        if not hasattr(self, "_mom"):
            self._mom = {k: torch.zeros_like(v) for k, v in self.params.items()}
        params = {k: v.detach().requires_grad_(True) for k, v in self.params.items()}
        with torch.no_grad():
            x = mfcc(wav, self.cfg)
        out = forward(self.arch, params, self.stats, x, True)
        tot, _ = total_loss(out["logits"], labels, params, weight_decay)
        tot.backward()
        with torch.no_grad():
            for k, v in params.items():
                g = v.grad if v.grad is not None else torch.zeros_like(v)
                self._mom[k].mul_(mu).add_(g)
                self.params[k] = v.detach() - lr * self._mom[k]
            self.stats = {k: v.detach() for k, v in out["new_stats"].items()}
        return float(tot.detach())
