"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of DS-CNN (audio_nets/ds_cnn.py:19-118 of the reference).

PARITY UNPINNED (see oracle/__init__.py).  Layout follows TF NHWC: activations [N, H, W, C] with H = time
(49 frames), W = MFCC coefficient (10), C = 1 at the input (factory/audio_nets.py feeds [B, T, F, 1]).

DSCNN_arg_scope (:104-118): conv2d / separable_conv2d have activation_fn=None, Xavier weights, ZERO-initialised
biases (present); slim.batch_norm with decay 0.96, epsilon 1e-3 (slim default), center=True, scale=False (slim
default -> no gamma), activation_fn=relu applied inside the BN layer.  SAME padding everywhere (slim default).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

from .numpy_ref import BN_EPS, same_pad, softmax

BN_DECAY = 0.96


@dataclass
class DsBlock:
    type: str            # "conv" | "separable"
    depth: int
    kernel: Tuple[int, int]
    stride: Tuple[int, int]
    scope: str


def net_def(size: str) -> List[DsBlock]:
    """S_NET_DEF / M_NET_DEF / L_NET_DEF (audio_nets/ds_cnn.py:19-43)."""
    depth = {"S": 64, "M": 172, "L": 276}[size]
    first = (2, 2) if size == "S" else (2, 1)
    ds1 = (1, 1) if size == "S" else (2, 2)
    n_ds = 4 if size in ("S", "M") else 5
    out = [DsBlock("conv", depth, (10, 4), first, "conv_1")]
    for i in range(1, n_ds + 1):
        out.append(DsBlock("separable", depth, (3, 3), ds1 if i == 1 else (1, 1), f"conv_ds_{i}"))
    return out


def init_params(blocks: List[DsBlock], num_classes: int = 12, in_channels: int = 1, seed: int = 0, dtype=np.float64,
                randomize: bool = True):
    """TF variable names: DSCNN/conv_1/{weights,biases}, DSCNN/conv_1/batch_norm/{beta,moving_*},
    DSCNN/conv_ds_k/depthwise_conv/{depthwise_weights,biases}, .../dw_batch_norm/*, .../pointwise_conv/{weights,biases},
    .../pw_batch_norm/*, DSCNN/fc1/{weights,biases}.  `randomize` fills biases/beta/moving stats with non-trivial
    values (the reference initialises them to 0 / 0 / 0,1) so that parity exercises every term."""
    rng = np.random.RandomState(seed)
    p: Dict[str, np.ndarray] = {}
    s: Dict[str, np.ndarray] = {}

    def xavier(shape, fan_in, fan_out):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-lim, lim, size=shape).astype(dtype)

    def bn(prefix, c):
        p[prefix + "/beta"] = (rng.uniform(-0.5, 0.5, c) if randomize else np.zeros(c)).astype(dtype)
        s[prefix + "/moving_mean"] = (rng.uniform(-0.5, 0.5, c) if randomize else np.zeros(c)).astype(dtype)
        s[prefix + "/moving_variance"] = (rng.uniform(0.5, 2.0, c) if randomize else np.ones(c)).astype(dtype)

    def bias(c):
        return (rng.uniform(-0.2, 0.2, c) if randomize else np.zeros(c)).astype(dtype)

    c = in_channels
    for b in blocks:
        kh, kw = b.kernel
        if b.type == "conv":
            p[f"DSCNN/{b.scope}/weights"] = xavier((kh, kw, c, b.depth), kh * kw * c, kh * kw * b.depth)
            p[f"DSCNN/{b.scope}/biases"] = bias(b.depth)
            bn(f"DSCNN/{b.scope}/batch_norm", b.depth)
        else:
            # slim.separable_conv2d(num_outputs=None, depth_multiplier=1): depthwise only; xavier fans as for a conv
            p[f"DSCNN/{b.scope}/depthwise_conv/depthwise_weights"] = xavier((kh, kw, c, 1), kh * kw * c, kh * kw * 1)
            p[f"DSCNN/{b.scope}/depthwise_conv/biases"] = bias(c)
            bn(f"DSCNN/{b.scope}/dw_batch_norm", c)
            p[f"DSCNN/{b.scope}/pointwise_conv/weights"] = xavier((1, 1, c, b.depth), c, b.depth)
            p[f"DSCNN/{b.scope}/pointwise_conv/biases"] = bias(b.depth)
            bn(f"DSCNN/{b.scope}/pw_batch_norm", b.depth)
        c = b.depth
    p["DSCNN/fc1/weights"] = xavier((c, num_classes), c, num_classes)
    p["DSCNN/fc1/biases"] = bias(num_classes)
    return p, s


def _pad2d(x, kh, kw, sh, sw):
    n, h, w, c = x.shape
    oh, pt, pb = same_pad(h, kh, sh)
    ow, pl, pr = same_pad(w, kw, sw)
    return np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0))), oh, ow


def conv2d(x, w, stride):
    """tf.nn.conv2d NHWC / HWIO, SAME."""
    kh, kw, ci, co = w.shape
    sh, sw = stride
    xp, oh, ow = _pad2d(x, kh, kw, sh, sw)
    hi = np.arange(oh)[:, None] * sh + np.arange(kh)[None, :]
    wi = np.arange(ow)[:, None] * sw + np.arange(kw)[None, :]
    cols = xp[:, hi[:, None, :, None], wi[None, :, None, :], :]          # [N, oh, ow, kh, kw, C]
    return np.einsum("nhwijc,ijco->nhwo", cols, w, optimize=True)


def depthwise_conv2d(x, w, stride):
    """tf.nn.depthwise_conv2d, depth_multiplier 1: w [kh, kw, C, 1]."""
    kh, kw, c, _ = w.shape
    sh, sw = stride
    xp, oh, ow = _pad2d(x, kh, kw, sh, sw)
    hi = np.arange(oh)[:, None] * sh + np.arange(kh)[None, :]
    wi = np.arange(ow)[:, None] * sw + np.arange(kw)[None, :]
    cols = xp[:, hi[:, None, :, None], wi[None, :, None, :], :]
    return np.einsum("nhwijc,ijc->nhwc", cols, w[..., 0], optimize=True)


def _bn_relu(y, p, s, prefix, is_training, new_stats, cache=None):
    beta = p[prefix + "/beta"]
    if is_training:
        mean = y.mean(axis=(0, 1, 2))
        var = ((y - mean) ** 2).mean(axis=(0, 1, 2))
        if cache is not None:
            inv = 1.0 / np.sqrt(var + BN_EPS)
            cache[prefix] = {"xhat": (y - mean) * inv, "inv": inv}
        n = y.shape[0] * y.shape[1] * y.shape[2]
        mm, mv = s[prefix + "/moving_mean"], s[prefix + "/moving_variance"]
        new_stats[prefix + "/moving_mean"] = mm - (1.0 - BN_DECAY) * (mm - mean)
        new_stats[prefix + "/moving_variance"] = mv - (1.0 - BN_DECAY) * (mv - var * n / max(n - 1, 1))
    else:
        mean, var = s[prefix + "/moving_mean"], s[prefix + "/moving_variance"]
    return np.maximum((y - mean) / np.sqrt(var + BN_EPS) + beta, 0.0)       # scale=False: no gamma; relu inside BN


def forward(blocks: List[DsBlock], p, s, x, is_training: bool = False):
    """DSCNN() (audio_nets/ds_cnn.py:89-101).  x: [N, H, W] or [N, H, W, 1].  Returns dict(logits, probs, new_stats)."""
    if x.ndim == 3:
        x = x[..., None]
    new_stats = dict(s)
    cache: Dict[str, dict] = {}
    net = x
    for b in blocks:
        if b.type == "conv":                                                     # parse_block :66-74
            cache[b.scope + "/in"] = net
            net = conv2d(net, p[f"DSCNN/{b.scope}/weights"], b.stride) + p[f"DSCNN/{b.scope}/biases"]
            net = _bn_relu(net, p, s, f"DSCNN/{b.scope}/batch_norm", is_training, new_stats, cache)
        else:                                                                    # _depthwise_separable_conv :46-62
            pre = f"DSCNN/{b.scope}"
            cache[b.scope + "/in"] = net
            net = depthwise_conv2d(net, p[pre + "/depthwise_conv/depthwise_weights"], b.stride) + p[pre + "/depthwise_conv/biases"]
            net = _bn_relu(net, p, s, pre + "/dw_batch_norm", is_training, new_stats, cache)
            cache[b.scope + "/mid"] = net
            net = conv2d(net, p[pre + "/pointwise_conv/weights"], (1, 1)) + p[pre + "/pointwise_conv/biases"]
            net = _bn_relu(net, p, s, pre + "/pw_batch_norm", is_training, new_stats, cache)
        cache[b.scope + "/out"] = net
    pooled = net.mean(axis=(1, 2))                                               # avg_pool over the full map (:96)
    logits = pooled @ p["DSCNN/fc1/weights"] + p["DSCNN/fc1/biases"]             # slim.fully_connected (:99)
    return {"logits": logits, "probs": softmax(logits), "new_stats": new_stats, "feat": net, "pooled": pooled, "cache": cache}


# --------------------------------------------------------------------------- #
# Training: tf.gradients of the graph above + tf.train.AdamOptimizer (helper/trainer.py:171-222;
# scripts/commands/DSCNN*Model*.sh: --optimizer adam, lr 5e-4 -> 1e-4 at step 10000)
# --------------------------------------------------------------------------- #
def _cols_index(h, w, kh, kw, sh, sw):
    oh, pt, pb = same_pad(h, kh, sh)
    ow, pl, pr = same_pad(w, kw, sw)
    hi = (np.arange(oh)[:, None] * sh + np.arange(kh)[None, :])[:, None, :, None]
    wi = (np.arange(ow)[:, None] * sw + np.arange(kw)[None, :])[None, :, None, :]
    return (pt, pb, pl, pr), hi, wi


def conv2d_bwd(x, w, dy, stride, need_dx=True):
    """Gradients of tf.nn.conv2d (NHWC / HWIO, SAME) wrt input and filter."""
    kh, kw, ci, co = w.shape
    (pt, pb, pl, pr), hi, wi = _cols_index(x.shape[1], x.shape[2], kh, kw, *stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    cols = xp[:, hi, wi, :]
    dw = np.einsum("nhwijc,nhwo->ijco", cols, dy, optimize=True)
    dx = None
    if need_dx:
        dxp = np.zeros_like(xp)
        np.add.at(dxp, (slice(None), hi, wi, slice(None)), np.einsum("nhwo,ijco->nhwijc", dy, w, optimize=True))
        dx = dxp[:, pt:pt + x.shape[1], pl:pl + x.shape[2], :]
    return dx, dw


def depthwise_conv2d_bwd(x, w, dy, stride):
    kh, kw, c, _ = w.shape
    (pt, pb, pl, pr), hi, wi = _cols_index(x.shape[1], x.shape[2], kh, kw, *stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    cols = xp[:, hi, wi, :]
    dw = np.einsum("nhwijc,nhwc->ijc", cols, dy, optimize=True)[..., None]
    dxp = np.zeros_like(xp)
    np.add.at(dxp, (slice(None), hi, wi, slice(None)), dy[:, :, :, None, None, :] * w[None, None, None, :, :, :, 0])
    return dxp[:, pt:pt + x.shape[1], pl:pl + x.shape[2], :], dw


def _bn_relu_bwd(da, out, c, positive=None):
    """Backward of relu(xhat + beta) with train-mode statistics (FusedBatchNormGrad, scale=False): returns (dz, dbeta).
    positive: optional boolean mask to use instead of (out > 0) (tests: the implementation's side of inputs within round-off of 0)."""
    dy = da * ((out > 0) if positive is None else positive)
    n = dy.shape[0] * dy.shape[1] * dy.shape[2]
    dbeta = dy.sum(axis=(0, 1, 2))
    dgx = (dy * c["xhat"]).sum(axis=(0, 1, 2))
    return c["inv"] * (dy - dbeta / n - c["xhat"] * dgx / n), dbeta


def loss(logits, labels_onehot):
    from .numpy_ref import log_softmax
    return float(-(labels_onehot * log_softmax(logits)).sum(axis=1).mean())


def backward(blocks: List[DsBlock], p, fwd, labels_onehot, masks=None):
    """Gradient of the mean softmax cross-entropy wrt every trainable.  The conv biases feed a train-mode BN: their
    gradient (sum of dz) is zero up to round-off; it is computed honestly here."""
    g: Dict[str, np.ndarray] = {}
    cache = fwd["cache"]
    masks = masks or {}             # cache key ("conv_ds_2/mid", ...) -> ReLU mask replacing (activation > 0)
    b = labels_onehot.shape[0]
    dlogits = (fwd["probs"] - labels_onehot) / b
    g["DSCNN/fc1/weights"] = fwd["pooled"].T @ dlogits
    g["DSCNN/fc1/biases"] = dlogits.sum(axis=0)
    dpooled = dlogits @ p["DSCNN/fc1/weights"].T
    feat = fwd["feat"]
    da = np.broadcast_to(dpooled[:, None, None, :], feat.shape) / (feat.shape[1] * feat.shape[2])
    for blk in reversed(blocks):
        pre = f"DSCNN/{blk.scope}"
        if blk.type == "separable":
            dz, g[pre + "/pw_batch_norm/beta"] = _bn_relu_bwd(da, cache[blk.scope + "/out"], cache[pre + "/pw_batch_norm"], masks.get(blk.scope + "/out"))
            g[pre + "/pointwise_conv/biases"] = dz.sum(axis=(0, 1, 2))
            da, g[pre + "/pointwise_conv/weights"] = conv2d_bwd(cache[blk.scope + "/mid"], p[pre + "/pointwise_conv/weights"], dz, (1, 1))
            dz, g[pre + "/dw_batch_norm/beta"] = _bn_relu_bwd(da, cache[blk.scope + "/mid"], cache[pre + "/dw_batch_norm"], masks.get(blk.scope + "/mid"))
            g[pre + "/depthwise_conv/biases"] = dz.sum(axis=(0, 1, 2))
            da, g[pre + "/depthwise_conv/depthwise_weights"] = depthwise_conv2d_bwd(
                cache[blk.scope + "/in"], p[pre + "/depthwise_conv/depthwise_weights"], dz, blk.stride)
        else:
            dz, g[pre + "/batch_norm/beta"] = _bn_relu_bwd(da, cache[blk.scope + "/out"], cache[pre + "/batch_norm"], masks.get(blk.scope + "/out"))
            g[pre + "/biases"] = dz.sum(axis=(0, 1, 2))
            _, g[pre + "/weights"] = conv2d_bwd(cache[blk.scope + "/in"], p[pre + "/weights"], dz, blk.stride, need_dx=False)
    return g


def adam_step(params, m, v, grads, lr: float, t: int, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
    """tf.train.AdamOptimizer: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t); m, v exponential averages; w -= lr_t * m / (sqrt(v) + eps)."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    np_, nm, nv = {}, {}, {}
    for k, w in params.items():
        gk = grads[k]
        nm[k] = m[k] + (1.0 - beta1) * (gk - m[k])
        nv[k] = v[k] + (1.0 - beta2) * (gk * gk - v[k])
        np_[k] = w - lr_t * nm[k] / (np.sqrt(nv[k]) + eps)
    return np_, nm, nv


def train_step(blocks, p, s, m, v, x, labels_onehot, lr: float, t: int):
    fwd = forward(blocks, p, s, x, True)
    g = backward(blocks, p, fwd, labels_onehot)
    np_, nm, nv = adam_step(p, m, v, g, lr, t)
    return np_, fwd["new_stats"], nm, nv, {"logits": fwd["logits"], "model_loss": loss(fwd["logits"], labels_onehot), "grads": g}
