"""TEST INFRASTRUCTURE ONLY -- PyTorch-CPU float64 restatement of the reference's 2-D model families.

PARITY UNPINNED (oracle/__init__.py): TensorFlow cannot run here and the reference ships no vectors; the semantics below are the
TF 1.13 / slim ones already restated for the TC-ResNet path (SAME padding with the extra element on the high side,
FusedBatchNorm with biased batch variance and Bessel-corrected moving variance, conv -> [BN] -> activation order), applied to

  * ResNet2D8 / ResNet2D8Pool   audio_nets/tc_resnet.py:6-54 (debug_2d branch :14-15, avg_pool_0 :23-24), :73-99, arg scope :102-123
  * Res8 / Res15 (/Narrow)      audio_nets/res.py:6-123
  * KWSModel architectures       audio_nets/kws.py:65-757 (all but low_latency_svdf)

written directly from those files (F.conv2d / F.batch_norm / F.max_pool2d / autograd), independently of the product's graph builders.
Tensors are NCHW; weights keep the TF HWIO layout under the TF variable names.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import numpy_ref as R

DT = torch.float64

# ReLU-kink bookkeeping for the tests: with `KINK_LOG["on"]` every ReLU input is scanned for elements within `tau` of zero -- an
# f32 implementation cannot be expected to land on the same side of such an input, and each flipped mask moves the gradients.
# `KINK_LOG["decide"]`: optional list, one entry per ReLU in call order, of boolean tensors "the implementation under test kept
# this element" -- consulted ONLY for the elements within `tau` of zero, so that the float64 gradient is taken on the same side of
# those kinks as the implementation's (every other element keeps its own sign; a wrong mask elsewhere is still caught).
KINK_LOG = {"on": False, "tau": 1e-5, "near": 0, "total": 0, "decide": None, "idx": 0, "followed": 0}


def _relu(x):
    if KINK_LOG["on"]:
        near = x.detach().abs() < KINK_LOG["tau"]
        n_near = int(near.sum())
        KINK_LOG["near"] += n_near
        KINK_LOG["total"] += x.numel()
        dec = KINK_LOG["decide"]
        if dec is not None:
            kept = dec[KINK_LOG["idx"]]
            KINK_LOG["idx"] += 1
            if kept.numel() != x.numel():
                raise ValueError(f"ReLU #{KINK_LOG['idx'] - 1}: mask of {tuple(kept.shape)} for an input of {tuple(x.shape)}")
            if n_near:
                mask = torch.where(near, kept.reshape(x.shape), x.detach() > 0)
                KINK_LOG["followed"] += int((mask != (x.detach() > 0)).sum())
                return x * mask.to(x.dtype)
    return F.relu(x)


def _same_pads(length: int, k_eff: int, stride: int):
    out = -(-length // stride)
    total = max((out - 1) * stride + k_eff - length, 0)
    return total // 2, total - total // 2


def conv2d(x, w_hwio, stride=(1, 1), rate=(1, 1), padding="SAME", bias=None):
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    if padding == "SAME":
        pt, pb = _same_pads(x.shape[2], (kh - 1) * rate[0] + 1, stride[0])
        pl, pr = _same_pads(x.shape[3], (kw - 1) * rate[1] + 1, stride[1])
        x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w_hwio.permute(3, 2, 0, 1), bias, stride=stride, dilation=rate)


def batch_norm(x, name, params, stats, new_stats, is_training, decay, eps, center, scale):
    g = params[name + "/gamma"] if scale else None
    b = params[name + "/beta"] if center else None
    mm, mv = stats[name + "/moving_mean"], stats[name + "/moving_variance"]
    if is_training:
        rm, rv = mm.clone(), mv.clone()
        y = F.batch_norm(x, rm, rv, g, b, True, 1.0 - decay, eps)       # running stats move towards the UNBIASED batch variance
        new_stats[name + "/moving_mean"], new_stats[name + "/moving_variance"] = rm, rv
        return y
    return F.batch_norm(x, mm, mv, g, b, False, 0.0, eps)


def max_pool(x, kernel, stride, padding):
    if padding == "SAME":
        pt, pb = _same_pads(x.shape[2], kernel[0], stride[0])
        pl, pr = _same_pads(x.shape[3], kernel[1], stride[1])
        x = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
    return F.max_pool2d(x, kernel, stride)


class _Masks:
    """Dropout masks in order of appearance (the product draws them from its counter-based generator; tests pass them in)."""

    def __init__(self, masks: Optional[List[np.ndarray]]):
        self.masks, self.i = masks, 0

    def apply(self, x, keep_prob, is_training):
        if not is_training:
            return x
        if self.masks is None:
            assert keep_prob == 1.0, "training with dropout needs explicit masks"
            return x
        m = torch.as_tensor(self.masks[self.i], dtype=x.dtype).reshape(x.shape)
        self.i += 1
        return (x / keep_prob) * m              # tf.nn.dropout: div(x, keep_prob) * floor(keep_prob + uniform)


# ---- audio_nets/tc_resnet.py, debug_2d branch -------------------------------------------------------------------------------
def resnet2d_channels(f: int, width_multiplier: float) -> List[int]:
    n = [int(x * width_multiplier) for x in [16, 24, 32, 48]]          # :74-76
    c1, c2 = n[0:2]
    n[0] = int((3 * f * c1 + 10 * c1 * c2) / (9 + 10 * c2))             # :79-82
    return n


def resnet2d_forward(params, stats, x, scope="ResNet2D8", width_multiplier=1.0, pool=None, is_training=False, keep_prob=1.0, masks=None):
    """x [N, T, F] -> dict(logits, probs, new_stats).  tc_resnet(debug_2d=True): 3 x 3 convs on the [N, L, F, 1] input (:14-15,21)."""
    new_stats = dict(stats)
    drop = _Masks(masks)
    n_channels = resnet2d_channels(x.shape[2], width_multiplier)

    def conv_bn(h, name, stride, relu):
        y = conv2d(h, params[f"{scope}/{name}/weights"], (stride, stride))
        y = batch_norm(y, f"{scope}/{name}/BatchNorm", params, stats, new_stats, is_training, 0.997, 0.001, True, True)
        return _relu(y) if relu else y

    net = conv_bn(x.unsqueeze(1), "conv0", 1, True)
    if pool is not None:
        net = F.avg_pool2d(net, pool[0], pool[1])                       # :23-24 (slim default padding VALID)
    c = n_channels[0]
    for i, n in enumerate(n_channels[1:]):
        if n != c:
            stride, layer_in = 2, conv_bn(net, f"block{i}/down", 2, True)        # :29-32 (1x1; default activation ReLU)
        else:
            stride, layer_in = 1, net
        h = conv_bn(net, f"block{i}/conv{i}_0", stride, True)
        h = conv_bn(h, f"block{i}/conv{i}_1", 1, False)
        net = _relu(h + layer_in)
        c = n
    net = net.mean(dim=(2, 3), keepdim=True)                            # :43
    net = drop.apply(net, keep_prob, is_training)                       # :45
    logits = conv2d(net, params[f"{scope}/fc/weights"]).flatten(1)
    return {"logits": logits, "probs": F.softmax(logits, dim=-1), "new_stats": new_stats}


# ---- audio_nets/res.py -----------------------------------------------------------------------------------------------------
RES_VARIANTS = {"Res8": (6, 45, (4, 3), False), "Res8Narrow": (6, 19, (4, 3), False), "Res15": (13, 45, None, True),
                "Res15Narrow": (13, 19, None, True)}


def res_forward(params, stats, x, variant="Res8", is_training=False, scope="Res"):
    num_layers, _ch, pool_size, use_dilation = RES_VARIANTS[variant]
    new_stats = dict(stats)

    def bn(h, name):
        return batch_norm(h, f"{scope}/{name}", params, stats, new_stats, is_training, 0.997, 0.001, False, False)

    def conv_relu_bn(h, idx, with_bn):                                  # :6-26
        rate = int(2 ** (idx // 3)) if use_dilation else 1
        h = _relu(conv2d(h, params[f"{scope}/conv{idx}/weights"], rate=(rate, rate)))
        return bn(h, f"conv{idx}_bn") if with_bn else h

    net = _relu(conv2d(x.unsqueeze(1), params[f"{scope}/f_conv/weights"]))
    if pool_size:
        net = F.avg_pool2d(net, pool_size, 1)
    idx = 0
    for i in range(num_layers // 2):
        layer_in = net
        net = conv_relu_bn(net, idx, True)
        idx += 1
        net = conv_relu_bn(net, 2 * i + 1, False)
        idx += 1
        net = bn(net + layer_in, f"conv{2 * i + 1}_bn")
    if num_layers % 2 != 0:
        net = conv_relu_bn(net, idx, True)
    net = net.mean(dim=(2, 3), keepdim=True)
    logits = conv2d(net, params[f"{scope}/fc/weights"]).flatten(1)
    return {"logits": logits, "probs": F.softmax(logits, dim=-1), "new_stats": new_stats}


# ---- audio_nets/kws.py -----------------------------------------------------------------------------------------------------
def _flat(x):
    """tf.reshape(NHWC activation, [-1, H*W*C])."""
    return x.permute(0, 2, 3, 1).flatten(1)


def kws_forward(params, x, architecture, is_training=False, masks=None):
    drop = _Masks(masks)
    d = lambda h: drop.apply(h, 0.5, is_training)
    p = params
    x4 = x.unsqueeze(1)
    t = x.shape[1]
    if architecture == "single_fc":
        logits = x.flatten(1) @ p["weights"].reshape(-1, p["weights"].shape[-1]) + p["bias"]
    elif architecture == "conv":
        h = d(_relu(conv2d(x4, p["first_weights"], padding="SAME", bias=p["first_bias"])))
        h = max_pool(h, (2, 2), (2, 2), "SAME")
        h = d(_relu(conv2d(h, p["second_weights"], padding="SAME", bias=p["second_bias"])))
        logits = _flat(h) @ p["final_fc_weights"].reshape(-1, p["final_fc_weights"].shape[-1]) + p["final_fc_bias"]
    elif architecture == "trad_fpool3":
        h = d(_relu(conv2d(x4, p["first_weights"], padding="VALID")))
        h = max_pool(h, (1, 3), (1, 3), "VALID")
        h = d(_relu(conv2d(h, p["second_weights"], padding="VALID")))
        h = _flat(h) @ p["linear_weights"].reshape(-1, 32)
        h = h @ p["first_fc_weights"].reshape(32, 128)
        logits = h @ p["final_fc_weights"].reshape(128, -1)
    elif architecture == "low_latency_conv":
        h = d(_relu(conv2d(x4, p["first_weights"], padding="VALID", bias=p["first_bias"])))
        h = d(_flat(h) @ p["first_fc_weights"].reshape(-1, 128) + p["first_fc_bias"])
        h = d(h @ p["second_fc_weights"].reshape(128, 128) + p["second_fc_bias"])
        logits = h @ p["final_fc_weights"].reshape(128, -1) + p["final_fc_bias"]
    elif architecture == "one_fstride4":
        h = d(_relu(conv2d(x4, p["first_weights"], stride=(1, 4), padding="VALID")))
        h = _flat(h) @ p["first_linear_weights"].reshape(-1, 32)
        h = d(h @ p["first_fc_weights"].reshape(32, 128))
        h = d(h @ p["second_fc_weights"].reshape(128, 128))
        logits = h @ p["final_fc_weights"].reshape(128, -1)
    elif architecture == "tiny_conv":
        h = d(_relu(conv2d(x4, p["first_weights"], stride=(2, 2), padding="SAME", bias=p["first_bias"])))
        logits = _flat(h) @ p["final_fc_weights"].reshape(-1, p["final_fc_weights"].shape[-1]) + p["final_fc_bias"]
    elif architecture == "low_latency_svdf":                                      # kws.py:490-680, training graph (all frames, no runtime memory)
        rank = 2
        wf = p["weights_frequency"].reshape(x.shape[2], -1)                       # [F, filters]
        act = torch.einsum("btf,fk->kbt", x, wf)                                  # conv1d over time, transposed to [filters, B, T] (:586-600)
        out = torch.einsum("kbt,kt->kb", act, p["weights_time"].reshape(act.shape[0], -1))     # tf.matmul(activations_time, weights_time) (:604-612)
        units = out.reshape(-1, rank, x.shape[0]).sum(dim=1).t()                  # [B, units] (:613-621)
        h = d(_relu(units + p["bias"]))
        h = d(h @ p["first_fc_weights"].reshape(-1, 256) + p["first_fc_bias"])
        h = d(h @ p["second_fc_weights"].reshape(256, 256) + p["second_fc_bias"])
        logits = h @ p["final_fc_weights"].reshape(256, -1) + p["final_fc_bias"]
    else:
        raise ValueError(architecture)
    return {"logits": logits, "probs": F.softmax(logits, dim=-1), "new_stats": {}}


# ---- loss / gradients ------------------------------------------------------------------------------------------------------
def loss_and_grads(forward_fn, params_np: Dict[str, np.ndarray], labels_np: np.ndarray, weight_decay: float = 0.0):
    """(out, model_loss, total_loss, grads of the MODEL loss) through autograd.  forward_fn(params as float64 tensors) -> dict."""
    params = {k: torch.tensor(np.asarray(v), dtype=DT, requires_grad=True) for k, v in params_np.items()}
    out = forward_fn(params)
    y = torch.tensor(labels_np, dtype=DT)
    model = torch.mean(-(y * F.log_softmax(out["logits"], dim=-1)).sum(dim=-1))          # factory/audio_nets.py:168-173
    l2 = weight_decay * sum(0.5 * (v ** 2).sum() for k, v in params.items() if R.is_l2_param(k))    # :175-180
    model.backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else np.zeros_like(params_np[k])) for k, v in params.items()}
    return ({k: (v.detach().numpy() if torch.is_tensor(v) else {a: b.detach().numpy() for a, b in v.items()}) for k, v in out.items()},
            float(model.detach()), float((model + l2).detach()), grads)


def dropout_mask(seed: int, node: int, sample_offset: int, batch: int, per_sample: int, keep_prob: float) -> np.ndarray:
    """Bit-exact mirror of the kernels' mask (csrc/net2d_kernels.hip eltwise2d_kernel): element e of sample n draws
    uniform01(seed + 0x9e3779b97f4a7c15 * (node + 1), (sample_offset + n) * per_sample + e) < keep_prob."""
    from .make_golden import dropout_uniform
    s = (int(seed) + 0x9e3779b97f4a7c15 * (int(node) + 1)) & ((1 << 64) - 1)
    idx = (np.uint64(sample_offset) + np.arange(batch, dtype=np.uint64)[:, None]) * np.uint64(per_sample) + np.arange(per_sample, dtype=np.uint64)[None, :]
    return (dropout_uniform(s, idx) < np.float32(keep_prob)).astype(np.float64)
