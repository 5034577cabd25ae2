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
