"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference hot path.

PARITY UNPINNED (see oracle/__init__.py): TensorFlow 1.13.1 cannot run here and
the reference ships no golden vectors; every function below restates the TF op
semantics at the cited reference call site (paths relative to /root/reference).

Layout follows the reference (TF NHWC with W == 1): activations are [N, L, C],
conv weights are [k, Cin, Cout] (= TF HWIO [k, 1, Cin, Cout] squeezed).
Default dtype is float64 (the checker); pass dtype=np.float32 for an f32 run.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

BN_EPS = 0.001          # slim.batch_norm default epsilon (audio_nets/tc_resnet.py:103-109 sets no epsilon)
BN_DECAY_TCRESNET = 0.997   # audio_nets/tc_resnet.py:107
BN_DECAY_DSCNN = 0.96       # audio_nets/ds_cnn.py:107
LOG_OFFSET = 1e-6       # datasets/preprocessors.py:93


# --------------------------------------------------------------------------- #
# Front-end: datasets/preprocessors.py:54-96,183-194
# --------------------------------------------------------------------------- #
@dataclass
class FrontendCfg:
    """Flags of datasets/audio_data_wrapper.py:61-110 that the front-end reads."""
    sample_rate: int = 16000
    clip_duration_ms: int = 1000
    window_size_ms: float = 30.0
    window_stride_ms: float = 10.0
    num_mel_bins: int = 64
    lower_edge_hertz: float = 80.0
    upper_edge_hertz: float = 7600.0
    num_mfccs: int = 40

    @property
    def n_samples(self) -> int:
        return int(self.sample_rate * self.clip_duration_ms / 1000)

    @property
    def win(self) -> int:      # factory/audio_nets.py:63
        return int(self.sample_rate * self.window_size_ms / 1000)

    @property
    def hop(self) -> int:      # factory/audio_nets.py:64
        return int(self.sample_rate * self.window_stride_ms / 1000)

    @property
    def nfft(self) -> int:     # tf.contrib.signal.stft default fft_length: enclosing power of two
        return 1 << int(math.ceil(math.log2(self.win)))

    @property
    def n_frames(self) -> int:  # signal.frame(pad_end=False)
        return 1 + (self.n_samples - self.win) // self.hop

    @property
    def n_bins(self) -> int:
        return self.nfft // 2 + 1


FRONTEND_3010 = FrontendCfg(window_size_ms=30.0, window_stride_ms=10.0)   # scripts/commands/TCResNet8Model-1.0_mfcc_40_3010_0.001_mom_l1.sh:3
FRONTEND_4020 = FrontendCfg(window_size_ms=40.0, window_stride_ms=20.0)   # README.md:53 (--height 49 --width 40)


def hann_periodic(length: int, dtype=np.float64) -> np.ndarray:
    """tf.contrib.signal.hann_window(periodic=True): 0.5 - 0.5 cos(2 pi i / L)."""
    i = np.arange(length, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * i / length)).astype(dtype)


def hertz_to_mel(f):
    return 1127.0 * np.log1p(np.asarray(f, dtype=np.float64) / 700.0)


def linear_to_mel_weight_matrix(num_mel_bins: int, num_spectrogram_bins: int, sample_rate: float,
                                lower_edge_hertz: float, upper_edge_hertz: float,
                                dtype=np.float64) -> np.ndarray:
    """tf.contrib.signal.linear_to_mel_weight_matrix (datasets/preprocessors.py:80-86).

    HTK mel scale, DC bin zeroed, triangular filters, no area normalisation.
    Built in float64 and cast."""
    nyquist = sample_rate / 2.0
    lin = np.linspace(0.0, nyquist, num_spectrogram_bins)[1:]
    bins_mel = hertz_to_mel(lin)[:, None]
    edges = np.linspace(hertz_to_mel(lower_edge_hertz), hertz_to_mel(upper_edge_hertz), num_mel_bins + 2)
    lower, center, upper = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    lower_slopes = (bins_mel - lower) / (center - lower)
    upper_slopes = (upper - bins_mel) / (upper - center)
    m = np.maximum(0.0, np.minimum(lower_slopes, upper_slopes))
    m = np.concatenate([np.zeros((1, num_mel_bins)), m], axis=0)
    return m.astype(dtype)


def frame_signal(wav: np.ndarray, win: int, hop: int) -> np.ndarray:
    """tf.contrib.signal.frame(pad_end=False): [B, n] -> [B, T, win]."""
    n = wav.shape[-1]
    t = 1 + (n - win) // hop
    idx = np.arange(t)[:, None] * hop + np.arange(win)[None, :]
    return wav[..., idx]


def log_mel_spectrogram(wav: np.ndarray, cfg: FrontendCfg, magnitude_squared: bool = True,
                        dtype=np.float64) -> np.ndarray:
    """AudioPreprocessorBase._log_mel_spectrogram (datasets/preprocessors.py:64-96).

    wav: [B, n_samples] or [B, n_samples, 1] -> [B, T, num_mel_bins]."""
    wav = np.asarray(wav)
    if wav.ndim == 3:
        wav = wav[..., 0]                       # tf.squeeze(audio, -1)  (:67)
    wav = wav.astype(dtype)
    frames = frame_signal(wav, cfg.win, cfg.hop) * hann_periodic(cfg.win, dtype)   # stft (:68-70)
    spec = np.fft.rfft(frames.astype(np.float64) if dtype == np.float64 else frames, n=cfg.nfft, axis=-1)
    if dtype != np.float64:
        spec = spec.astype(np.complex64)
    if magnitude_squared:
        s = (spec.real * spec.real + spec.imag * spec.imag)     # tf.real(stfts * conj(stfts)) (:74-75)
    else:
        s = np.abs(spec)                                         # (:76-77)
    s = s.astype(dtype)
    mel_w = linear_to_mel_weight_matrix(cfg.num_mel_bins, cfg.n_bins, cfg.sample_rate,
                                        cfg.lower_edge_hertz, cfg.upper_edge_hertz, dtype)
    mel = s @ mel_w                                              # tf.tensordot(..., 1) (:88)
    return np.log(mel + dtype(LOG_OFFSET)).astype(dtype)         # (:93-94)


def dct2_matrix(n_in: int, n_out: int, dtype=np.float64) -> np.ndarray:
    """mfccs_from_log_mel_spectrograms: unnormalised DCT-II times rsqrt(2 N)  -> [n_in, n_out]."""
    n = np.arange(n_in, dtype=np.float64)[:, None]
    k = np.arange(n_out, dtype=np.float64)[None, :]
    d = 2.0 * np.cos(np.pi * k * (2.0 * n + 1.0) / (2.0 * n_in)) / math.sqrt(2.0 * n_in)
    return d.astype(dtype)


def mfcc(wav: np.ndarray, cfg: FrontendCfg, dtype=np.float64) -> np.ndarray:
    """MFCCPreprocessor._preprocess (datasets/preprocessors.py:183-194) without the trailing
    expand_dims: [B, n_samples(,1)] -> [B, T, num_mfccs]."""
    lm = log_mel_spectrogram(wav, cfg, True, dtype)
    d = dct2_matrix(cfg.num_mel_bins, cfg.num_mel_bins, dtype)
    return (lm @ d)[..., :cfg.num_mfccs].astype(dtype)           # mfccs[..., :num_mfccs] (:192)


# --------------------------------------------------------------------------- #
# Network description: audio_nets/tc_resnet.py:6-70
# --------------------------------------------------------------------------- #
def mfcc_deploy(wav: np.ndarray, cfg: FrontendCfg, dtype=np.float64) -> np.ndarray:
    """Deploy-path MFCC (datasets/preprocessors.py:98-124,196-203): contrib_audio.audio_spectrogram(magnitude_squared=True)
    followed by contrib_audio.mfcc(spectrogram, sample_rate, upper/lower_frequency_limit, filterbank_channel_count,
    dct_coefficient_count).  The two ops are TF C++ kernels (core/kernels/spectrogram.cc, mfcc.cc, mfcc_mel_filterbank.cc,
    mfcc_dct.cc), restated here from their published algorithm (PARITY UNPINNED):
      * spectrogram: periodic Hann 0.5 - 0.5 cos(2 pi i / L), zero-padded to the next power of two, |rfft|^2,
        frames = 1 + (N - L) // stride;
      * filterbank: C + 1 centre frequencies equally spaced in mel(f) = 1127 ln(1 + f / 700) between the two limits,
        hz_per_bin = (sr / 2) / (bins - 1), bins int(1.5 + lower / hz_per_bin) .. int(upper / hz_per_bin); bin i in band b
        (largest b with centre[b] < mel_i; -1 below the first centre) adds w_i sqrt(P_i) to channel b and (1 - w_i) sqrt(P_i)
        to channel b + 1, w_i = (centre[b+1] - mel_i) / (centre[b+1] - centre[b]) (for b = -1: (centre[0] - mel_i) / (centre[0] - mel_low));
      * log(max(x, 1e-12)); DCT-II with sqrt(2 / C) cos(pi k (n + 0.5) / C), first dct_coefficient_count outputs."""
    wav = np.asarray(wav, dtype=dtype)
    frames = frame_signal(wav, cfg.win, cfg.hop) * hann_periodic(cfg.win, dtype)
    spec = np.fft.rfft(frames, n=cfg.nfft, axis=-1)
    power = spec.real ** 2 + spec.imag ** 2
    nbins, nch = cfg.n_bins, cfg.num_mel_bins
    mel = lambda f: 1127.0 * np.log1p(np.asarray(f, dtype=np.float64) / 700.0)
    mel_low, mel_hi = mel(cfg.lower_edge_hertz), mel(cfg.upper_edge_hertz)
    center = mel_low + (mel_hi - mel_low) / (nch + 1) * (np.arange(nch + 1) + 1)
    hz_per_sbin = 0.5 * cfg.sample_rate / (nbins - 1)
    start, end = int(1.5 + cfg.lower_edge_hertz / hz_per_sbin), int(cfg.upper_edge_hertz / hz_per_sbin)
    out = np.zeros(power.shape[:-1] + (nch,), dtype=dtype)
    channel = 0
    for i in range(start, min(end, nbins - 1) + 1):
        melf = mel(i * hz_per_sbin)
        while channel < nch and center[channel] < melf:
            channel += 1
        b = channel - 1
        w = (center[b + 1] - melf) / (center[b + 1] - center[b]) if b >= 0 else (center[0] - melf) / (center[0] - mel_low)
        v = np.sqrt(power[..., i])
        if b >= 0:
            out[..., b] += w * v
        if b + 1 < nch:
            out[..., b + 1] += v - w * v
    logmel = np.log(np.maximum(out, 1e-12))
    n = np.arange(nch)
    dct = np.sqrt(2.0 / nch) * np.cos(np.pi / nch * np.outer(np.arange(cfg.num_mfccs), n + 0.5))
    return logmel @ dct.T


@dataclass
class ConvSpec:
    name: str            # TF variable scope under the model scope, e.g. "block0/conv0_0"
    k: int
    stride: int
    cin: int
    cout: int
    bn: bool = True
    relu: bool = True


@dataclass
class BlockSpec:
    index: int
    down: Optional[ConvSpec]
    conv_a: ConvSpec
    conv_b: ConvSpec


@dataclass
class TCResNetArch:
    scope: str
    in_channels: int
    num_classes: int
    conv0: ConvSpec
    blocks: List[BlockSpec]
    fc: ConvSpec = None
    fc2: ConvSpec = None

    def convs(self) -> List[ConvSpec]:
        out = [self.conv0]
        for b in self.blocks:
            if b.down is not None:
                out.append(b.down)
            out += [b.conv_a, b.conv_b]
        return out + [self.fc, self.fc2]


def tcresnet_channels(name: str, width_multiplier: float) -> List[int]:
    base = {"TCResNet8": [16, 24, 32, 48],                       # audio_nets/tc_resnet.py:59
            "TCResNet14": [16, 24, 24, 32, 32, 48, 48]}[name]    # :67
    return [int(x * width_multiplier) for x in base]              # :60,68


def make_tcresnet(name: str = "TCResNet8", width_multiplier: float = 1.0, in_channels: int = 40,
                  num_classes: int = 12) -> TCResNetArch:
    """Mirror of tc_resnet() (audio_nets/tc_resnet.py:6-54), temporal-conv branch."""
    ch = tcresnet_channels(name, width_multiplier)
    conv0 = ConvSpec("conv0", 3, 1, in_channels, ch[0])          # :18,21
    blocks = []
    c = ch[0]
    for i, n in enumerate(ch[1:]):                                # :28
        if n != c:                                                # :29-32
            stride = 2
            down = ConvSpec(f"block{i}/down", 1, 2, c, n)
        else:
            stride = 1
            down = None
        a = ConvSpec(f"block{i}/conv{i}_0", 9, stride, c, n)      # :37
        b = ConvSpec(f"block{i}/conv{i}_1", 9, 1, n, n, relu=False)   # :38-39
        blocks.append(BlockSpec(i, down, a, b))
        c = n
    fc = ConvSpec("fc", 1, 1, c, num_classes, bn=False, relu=False)   # :47
    fc2 = ConvSpec("fc2", 1, 1, c, 2, bn=False, relu=False)           # :50
    return TCResNetArch(name, in_channels, num_classes, conv0, blocks, fc, fc2)


def init_params(arch: TCResNetArch, seed: int = 0, dtype=np.float64) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """Xavier-uniform weights (audio_nets/tc_resnet.py:112), gamma=1, beta=0, moving mean 0 / var 1.

    Returns (trainables, moving_stats) keyed by TF variable name."""
    rng = np.random.RandomState(seed)
    p, s = {}, {}
    for c in arch.convs():
        fan_in, fan_out = c.k * c.cin, c.k * c.cout
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        p[f"{arch.scope}/{c.name}/weights"] = rng.uniform(-lim, lim, size=(c.k, c.cin, c.cout)).astype(dtype)
        if c.bn:
            p[f"{arch.scope}/{c.name}/BatchNorm/gamma"] = np.ones(c.cout, dtype)
            p[f"{arch.scope}/{c.name}/BatchNorm/beta"] = np.zeros(c.cout, dtype)
            s[f"{arch.scope}/{c.name}/BatchNorm/moving_mean"] = np.zeros(c.cout, dtype)
            s[f"{arch.scope}/{c.name}/BatchNorm/moving_variance"] = np.ones(c.cout, dtype)
    return p, s


def randomize_bn(arch: TCResNetArch, params, stats, seed: int = 1):
    """Non-trivial gamma/beta/moving stats so eval-mode parity exercises every BN term."""
    rng = np.random.RandomState(seed)
    for c in arch.convs():
        if not c.bn:
            continue
        base = f"{arch.scope}/{c.name}/BatchNorm/"
        dt = params[base + "gamma"].dtype
        params[base + "gamma"] = rng.uniform(0.5, 1.5, c.cout).astype(dt)
        params[base + "beta"] = rng.uniform(-0.5, 0.5, c.cout).astype(dt)
        stats[base + "moving_mean"] = rng.uniform(-1.0, 1.0, c.cout).astype(dt)
        stats[base + "moving_variance"] = rng.uniform(0.5, 2.0, c.cout).astype(dt)


# --------------------------------------------------------------------------- #
# Layer arithmetic
# --------------------------------------------------------------------------- #
def same_pad(length: int, k: int, stride: int) -> Tuple[int, int, int]:
    """TF 'SAME' rule: out = ceil(L/s); extra padding goes on the high side."""
    out = -(-length // stride)
    total = max((out - 1) * stride + k - length, 0)
    lo = total // 2
    return out, lo, total - lo


def _im2col(x: np.ndarray, k: int, stride: int) -> Tuple[np.ndarray, int, int]:
    n, length, c = x.shape
    out, lo, hi = same_pad(length, k, stride)
    xp = np.pad(x, ((0, 0), (lo, hi), (0, 0)))
    idx = np.arange(out)[:, None] * stride + np.arange(k)[None, :]
    return xp[:, idx, :], lo, hi      # [N, out, k, C]


def conv1d(x: np.ndarray, w: np.ndarray, stride: int) -> np.ndarray:
    """slim.conv2d -> tf.nn.conv2d NHWC, SAME, no bias, W == 1 (audio_nets/tc_resnet.py:21,32,37,38,47,50)."""
    cols, _, _ = _im2col(x, w.shape[0], stride)
    return np.einsum("ntjc,jco->nto", cols, w, optimize=True)


def conv1d_bwd(x: np.ndarray, w: np.ndarray, dy: np.ndarray, stride: int) -> Tuple[np.ndarray, np.ndarray]:
    k = w.shape[0]
    cols, lo, hi = _im2col(x, k, stride)
    dw = np.einsum("ntjc,nto->jco", cols, dy, optimize=True)
    dcols = np.einsum("nto,jco->ntjc", dy, w, optimize=True)
    n, length, c = x.shape
    dxp = np.zeros((n, length + lo + hi, c), dtype=x.dtype)
    out = dy.shape[1]
    for j in range(k):
        dxp[:, j:j + (out - 1) * stride + 1:stride, :] += dcols[:, :, j, :]
    return dxp[:, lo:lo + length, :], dw


def bn_train(y, gamma, beta, eps=BN_EPS):
    """FusedBatchNorm, is_training=True: biased batch variance for normalisation."""
    mean = y.mean(axis=(0, 1))
    var = ((y - mean) ** 2).mean(axis=(0, 1))
    inv = 1.0 / np.sqrt(var + eps)
    xhat = (y - mean) * inv
    return xhat * gamma + beta, (xhat, inv, mean, var)


def bn_eval(y, gamma, beta, mm, mv, eps=BN_EPS):
    return (y - mm) / np.sqrt(mv + eps) * gamma + beta


def bn_bwd(dz, xhat, inv, gamma):
    n = dz.shape[0] * dz.shape[1]
    dbeta = dz.sum(axis=(0, 1))
    dgamma = (dz * xhat).sum(axis=(0, 1))
    dy = gamma * inv * (dz - dbeta / n - xhat * dgamma / n)
    return dy, dgamma, dbeta


def softmax(z):
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=-1, keepdims=True)


def log_softmax(z):
    z = z - z.max(axis=-1, keepdims=True)
    return z - np.log(np.exp(z).sum(axis=-1, keepdims=True))


# --------------------------------------------------------------------------- #
# Forward / loss / backward / optimiser
# --------------------------------------------------------------------------- #
def _conv_bn_act(arch, c: ConvSpec, x, params, stats, is_training, cache, new_stats, decay):
    base = f"{arch.scope}/{c.name}"
    y = conv1d(x, params[base + "/weights"], c.stride)
    rec = {"x": x, "y": y}
    if c.bn:
        g, b = params[base + "/BatchNorm/gamma"], params[base + "/BatchNorm/beta"]
        if is_training:
            z, (xhat, inv, mean, var) = bn_train(y, g, b)
            n = y.shape[0] * y.shape[1]
            # moving stats: slim.batch_norm(fused) -> assign_moving_average with the *unbiased*
            # batch variance returned by FusedBatchNorm (flagged there).
            mm, mv = stats[base + "/BatchNorm/moving_mean"], stats[base + "/BatchNorm/moving_variance"]
            new_stats[base + "/BatchNorm/moving_mean"] = mm - (1.0 - decay) * (mm - mean)
            new_stats[base + "/BatchNorm/moving_variance"] = mv - (1.0 - decay) * (mv - var * n / max(n - 1, 1))
            rec.update(xhat=xhat, inv=inv, mean=mean, var=var)
        else:
            z = bn_eval(y, g, b, stats[base + "/BatchNorm/moving_mean"], stats[base + "/BatchNorm/moving_variance"])
    else:
        z = y
    rec["z"] = z
    a = np.maximum(z, 0.0) if c.relu else z
    rec["a"] = a
    cache[c.name] = rec
    return a


def forward(arch: TCResNetArch, params, stats, x, is_training: bool = False, keep_prob: float = 1.0,
            dropout_mask: Optional[np.ndarray] = None, bn_decay: float = BN_DECAY_TCRESNET):
    """tc_resnet() forward (audio_nets/tc_resnet.py:6-54) + softmax (factory/audio_nets.py:147-156).

    x: [N, T, F] MFCC features (the [N,T,F,1] -> [N,T,1,F] reshape of :17 is a reinterpretation).
    dropout_mask: [N, C] of {0,1} (train only); None with keep_prob==1 means identity.
    Returns dict(logits, probs, ranges, cache, new_stats)."""
    cache: Dict[str, dict] = {}
    new_stats = dict(stats)
    net = _conv_bn_act(arch, arch.conv0, x, params, stats, is_training, cache, new_stats, bn_decay)
    for b in arch.blocks:
        block_in = net
        if b.down is not None:
            shortcut = _conv_bn_act(arch, b.down, block_in, params, stats, is_training, cache, new_stats, bn_decay)
        else:
            shortcut = block_in
        h = _conv_bn_act(arch, b.conv_a, block_in, params, stats, is_training, cache, new_stats, bn_decay)
        h = _conv_bn_act(arch, b.conv_b, h, params, stats, is_training, cache, new_stats, bn_decay)
        pre = h + shortcut                                   # :40
        net = np.maximum(pre, 0.0)                           # :41
        cache[f"block{b.index}/out"] = {"pre": pre, "a": net}
    pooled = net.mean(axis=1)                                # :43 avg_pool over full L
    if is_training and (keep_prob < 1.0 or dropout_mask is not None):
        assert dropout_mask is not None, "training with keep_prob<1 needs an explicit mask"
        dropped = (pooled / keep_prob) * dropout_mask        # tf.nn.dropout: div(x, keep_prob) * mask  (:45)
    else:
        dropped = pooled
    wfc = params[f"{arch.scope}/fc/weights"][0]
    wfc2 = params[f"{arch.scope}/fc2/weights"][0]
    logits = dropped @ wfc                                   # :47-48
    ranges = 1.0 / (1.0 + np.exp(-(dropped @ wfc2)))         # :50-52
    cache["head"] = {"feat": net, "pooled": pooled, "dropped": dropped, "mask": dropout_mask, "keep_prob": keep_prob}
    return {"logits": logits, "probs": softmax(logits), "ranges": ranges, "cache": cache, "new_stats": new_stats}


def is_l2_param(name: str) -> bool:
    """factory/audio_nets.py:175-176: every trainable whose name lacks BatchNorm/batch_normalization."""
    return ("batch_normalization" not in name) and ("BatchNorm" not in name)


def loss(logits, labels_onehot, params, weight_decay: float, label_smoothing: float = 0.0):
    """AudioNetModel.build_loss (factory/audio_nets.py:161-183). Returns (total, model, l2)."""
    y = labels_onehot
    if label_smoothing > 0:
        y = y * (1.0 - label_smoothing) + label_smoothing / y.shape[-1]
    model = float(np.mean(-(y * log_softmax(logits)).sum(axis=-1)))
    l2 = weight_decay * sum(0.5 * float((v.astype(np.float64) ** 2).sum()) for k, v in params.items() if is_l2_param(k))
    return model + l2, model, l2


def backward(arch: TCResNetArch, params, fwd, labels_onehot, weight_decay: float, label_smoothing: float = 0.0):
    """d(total_loss)/d(trainables): tf.gradients inside slim.learning.create_train_op
    (helper/trainer.py:205-211).  Standard formulas, SURVEY App. A.3."""
    cache = fwd["cache"]
    grads: Dict[str, np.ndarray] = {}
    y = labels_onehot
    if label_smoothing > 0:
        y = y * (1.0 - label_smoothing) + label_smoothing / y.shape[-1]
    nb = y.shape[0]
    dlogits = (fwd["probs"] - y) / nb
    head = cache["head"]
    wfc = params[f"{arch.scope}/fc/weights"][0]
    grads[f"{arch.scope}/fc/weights"] = (head["dropped"].T @ dlogits)[None]
    grads[f"{arch.scope}/fc2/weights"] = np.zeros_like(params[f"{arch.scope}/fc2/weights"])
    ddropped = dlogits @ wfc.T
    if head["mask"] is not None:
        dpooled = (ddropped * head["mask"]) / head["keep_prob"]
    else:
        dpooled = ddropped
    feat = head["feat"]
    dnet = np.repeat(dpooled[:, None, :], feat.shape[1], axis=1) / feat.shape[1]

    def conv_bn_act_bwd(c: ConvSpec, da):
        rec = cache[c.name]
        base = f"{arch.scope}/{c.name}"
        dz = da * (rec["a"] > 0) if c.relu else da
        if c.bn:
            dy, dg, db = bn_bwd(dz, rec["xhat"], rec["inv"], params[base + "/BatchNorm/gamma"])
            grads[base + "/BatchNorm/gamma"] = dg
            grads[base + "/BatchNorm/beta"] = db
        else:
            dy = dz
        dx, dw = conv1d_bwd(rec["x"], params[base + "/weights"], dy, c.stride)
        grads[base + "/weights"] = dw
        return dx

    for b in reversed(arch.blocks):
        out = cache[f"block{b.index}/out"]
        dpre = dnet * (out["a"] > 0)
        dh = conv_bn_act_bwd(b.conv_b, dpre)
        din = conv_bn_act_bwd(b.conv_a, dh)
        if b.down is not None:
            din = din + conv_bn_act_bwd(b.down, dpre)
        else:
            din = din + dpre
        dnet = din
    conv_bn_act_bwd(arch.conv0, dnet)
    for k in grads:
        if is_l2_param(k):
            grads[k] = grads[k] + weight_decay * params[k]     # d/dv [wd * 0.5 * |v|^2]
    return grads


def momentum_step(params, moms, grads, lr: float, mu: float = 0.9):
    """tf.train.MomentumOptimizer, use_nesterov=False (helper/trainer.py:188-190):
    a <- mu*a + g ; w <- w - lr*a."""
    new_p, new_m = {}, {}
    for k in params:
        a = mu * moms[k] + grads[k]
        new_m[k] = a
        new_p[k] = params[k] - lr * a
    return new_p, new_m


def piecewise_constant_lr(step: int, boundaries: List[int], values: List[float]) -> float:
    """tf.train.piecewise_constant (helper/trainer.py:135-137): x <= b0 -> v0; b0 < x <= b1 -> v1; ..."""
    for b, v in zip(boundaries, values):
        if step <= b:
            return v
    return values[-1]


def train_step(arch, params, stats, moms, x, labels_onehot, lr, weight_decay, mu=0.9, keep_prob=1.0,
               dropout_mask=None, label_smoothing=0.0):
    """One slim train_op: fwd (train-mode BN) + loss + grads + momentum update + moving-stat update."""
    fwd = forward(arch, params, stats, x, True, keep_prob, dropout_mask)
    total, model, l2 = loss(fwd["logits"], labels_onehot, params, weight_decay, label_smoothing)
    grads = backward(arch, params, fwd, labels_onehot, weight_decay, label_smoothing)
    new_p, new_m = momentum_step(params, moms, grads, lr, mu)
    return new_p, fwd["new_stats"], new_m, {"total_loss": total, "model_loss": model, "l2_loss": l2,
                                             "logits": fwd["logits"], "probs": fwd["probs"], "grads": grads}


# --------------------------------------------------------------------------- #
# Synthetic inputs of SURVEY 8(d)
# --------------------------------------------------------------------------- #
def synth_waveforms(batch: int, n_samples: int = 16000, seed: int = 1234, start: int = 0) -> np.ndarray:
    """uniform(-1,1)*0.5 noise + per-utterance sine 440*(1 + i mod 8) Hz at 0.25 amplitude, float32.
    Utterance i depends only on (seed, start+i) so shards of a global batch are reproducible."""
    out = np.empty((batch, n_samples), np.float32)
    t = np.arange(n_samples, dtype=np.float64) / 16000.0
    for i in range(batch):
        g = start + i
        rng = np.random.RandomState((seed * 1000003 + g) % (2 ** 31 - 1))
        noise = rng.uniform(-1.0, 1.0, n_samples) * 0.5
        sine = 0.25 * np.sin(2.0 * np.pi * 440.0 * (1 + g % 8) * t)
        out[i] = (noise + sine).astype(np.float32)
    return out


def synth_labels(batch: int, num_classes: int = 12, start: int = 0) -> np.ndarray:
    lab = np.zeros((batch, num_classes), np.float32)
    lab[np.arange(batch), (start + np.arange(batch)) % num_classes] = 1.0
    return lab
