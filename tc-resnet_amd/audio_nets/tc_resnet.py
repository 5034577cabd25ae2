"""TC-ResNet builders with the reference's names and arguments (audio_nets/tc_resnet.py:57-70,102-123).

`TCResNet8(inputs, num_classes, width_multiplier, scope)` returns (logits, endpoints) like the reference; the
layer topology itself lives in the C++ engine (csrc/net.cpp), which the call instantiates (or reuses: engines
are cached per variable scope, the eager analogue of TF variable reuse).  `TCResNet_arg_scope(...)` returns the
settings the reference passes through slim.arg_scope; use it with `arg_scope(...)` below."""
from __future__ import annotations

import contextlib
from typing import Dict, Optional, Tuple

import torch

from .. import runtime
from ..engine import TCResNet, features_to_planar
from .._lib import HALO

_scope_stack = []
_engines: Dict[Tuple, TCResNet] = {}


def TCResNet_arg_scope(is_training, weight_decay=0.001, keep_prob=0.5):
    """Reference :102-123: BN decay 0.997 (epsilon: slim default 1e-3), center+scale, Xavier init, no conv bias,
    SAME padding, ReLU; dropout keep_prob.  (slim's own l2 regulariser collection is never added to the loss
    by the reference, so weight_decay is only carried along.)"""
    return {"is_training": bool(is_training), "weight_decay": float(weight_decay), "keep_prob": float(keep_prob),
            "bn_decay": 0.997, "bn_eps": 0.001}


@contextlib.contextmanager
def arg_scope(scope: dict):
    _scope_stack.append(scope)
    try:
        yield scope
    finally:
        _scope_stack.pop()


def current_scope() -> dict:
    return _scope_stack[-1] if _scope_stack else TCResNet_arg_scope(False)


def reset_engines():
    _engines.clear()


def get_engine(scope: str, channels, in_channels: int, t_in: int, num_classes: int, bn_decay=0.997, bn_eps=0.001) -> TCResNet:
    key = (scope, tuple(channels), in_channels, t_in, num_classes, id(runtime.default_lib()))
    eng = _engines.get(key)
    if eng is None:
        eng = TCResNet(scope, list(channels), in_channels, t_in, num_classes, bn_decay=bn_decay, bn_eps=bn_eps,
                       lib=runtime.default_lib(), device=runtime.default_device())
        eng.init_xavier(0)
        _engines[key] = eng
    return eng


def _planar_of(inputs) -> torch.Tensor:
    planar = getattr(inputs, "_tcr_planar", None)
    return planar if planar is not None else features_to_planar(inputs, lib=runtime.default_lib())


def tc_resnet(inputs, num_classes, n_blocks, n_channels, scope, planar: Optional[torch.Tensor] = None):
    """inputs: [N, L, C, 1] features (reference :6-54, temporal-conv branch).  Eval-mode forward when the active
    arg scope has is_training False; training goes through AudioNetModel.train_step (factory/audio_nets.py)."""
    assert n_blocks == len(n_channels) - 1
    sc = current_scope()
    L, C = int(inputs.shape[1]), int(inputs.shape[2])
    eng = get_engine(scope, n_channels, C, L, num_classes, sc["bn_decay"], sc["bn_eps"])
    if planar is None:
        planar = _planar_of(inputs)
    endpoints = {}
    if sc["is_training"]:
        raise RuntimeError("train-mode graphs are driven by AudioNetModel.build/train_step (they need labels)")
    logits, probs, ranges = eng.forward_infer(planar, want_ranges=True)
    endpoints["ranges"] = ranges
    endpoints["softmax"] = probs        # what slim.softmax(logits) gives the caller (factory/audio_nets.py:154)
    endpoints["engine"] = eng
    return logits, endpoints


def tcresnet_channels(base, width_multiplier):
    return [int(x * width_multiplier) for x in base]


def TCResNet8(inputs, num_classes, width_multiplier=1.0, scope="TCResNet8", planar=None):
    return tc_resnet(inputs, num_classes, 3, tcresnet_channels([16, 24, 32, 48], width_multiplier), scope, planar)


def TCResNet14(inputs, num_classes, width_multiplier=1.0, scope="TCResNet14", planar=None):
    return tc_resnet(inputs, num_classes, 6, tcresnet_channels([16, 24, 24, 32, 32, 48, 48], width_multiplier), scope, planar)


# ---- the 2-D ablations (reference :14-15,23-24,73-99): 3 x 3 convolutions over the [L, F] plane -------------------------------
def build_resnet2d(g, num_classes, n_channels, scope, keep_prob, pool=None):
    """tc_resnet(..., debug_2d=True[, pool]) as Graph2D nodes; returns the logits node."""
    def conv_bn(inp, cout, kernel, stride, name, relu):
        net = g.conv(inp, kernel, cout, f"{scope}/{name}/weights", stride=stride)
        return g.batch_norm(net, f"{scope}/{name}/BatchNorm", center=True, scale=True, relu=relu, decay=0.997, eps=0.001)

    net = conv_bn(-1, n_channels[0], 3, 1, "conv0", True)
    if pool is not None:
        net = g.pool(net, "avg", tuple(pool[0]), stride=pool[1], padding="VALID")       # scope avg_pool_0
    c = n_channels[0]
    for i, n in enumerate(n_channels[1:]):
        if n != c:
            stride = 2
            layer_in = conv_bn(net, n, 1, 2, f"block{i}/down", True)
        else:
            stride, layer_in = 1, net
        h = conv_bn(net, n, 3, stride, f"block{i}/conv{i}_0", True)
        h = conv_bn(h, n, 3, 1, f"block{i}/conv{i}_1", False)
        net = g.add(h, layer_in, relu=True)
        c = n
    net = g.pool(net, "avg", None)
    net = g.dropout(net, keep_prob)
    logits = g.conv(net, 1, num_classes, f"{scope}/fc/weights")
    g.conv(net, 1, 2, f"{scope}/fc2/weights")           # endpoints["ranges"]: a variable of the graph, outside the loss
    return logits


def _resnet2d_channels(f, width_multiplier):
    n = tcresnet_channels([16, 24, 32, 48], width_multiplier)
    c1, c2 = n[0:2]
    n[0] = int((3 * f * c1 + 10 * c1 * c2) / (9 + 10 * c2))       # reference :79-82: same MACs as the temporal first block
    return n


def get_engine_2d(scope, h, w, num_classes, width_multiplier, keep_prob, pool):
    from ..engine import Graph2D
    key = (scope, h, w, num_classes, float(width_multiplier), float(keep_prob), id(runtime.default_lib()))
    eng = _engines.get(key)
    if eng is None:
        eng = Graph2D(scope, h, w, 1, lib=runtime.default_lib(), device=runtime.default_device())
        eng.finalize(build_resnet2d(eng, num_classes, _resnet2d_channels(w, width_multiplier), scope, keep_prob, pool))
        _engines[key] = eng
    return eng


def _run_2d(scope, inputs, num_classes, width_multiplier, planar, pool):
    sc = current_scope()
    eng = get_engine_2d(scope, int(inputs.shape[1]), int(inputs.shape[2]), num_classes, width_multiplier, sc["keep_prob"], pool)
    if sc["is_training"]:
        raise RuntimeError("train-mode graphs are driven by AudioNetModel.build/train_step (they need labels)")
    logits, probs = eng.forward_infer(planar if planar is not None else _planar_of(inputs))
    return logits, {"softmax": probs, "engine": eng}


def ResNet2D8(inputs, num_classes, width_multiplier=1.0, scope="ResNet2D8", planar=None):
    return _run_2d(scope, inputs, num_classes, width_multiplier, planar, None)


def ResNet2D8Pool(inputs, num_classes, width_multiplier=1.0, scope="ResNet2D8Pool", planar=None):
    return _run_2d(scope, inputs, num_classes, width_multiplier, planar, ([4, 4], 4))
