"""Res8 / Res8Narrow / Res15 / Res15Narrow with the reference's names and arguments (audio_nets/res.py:6-123: the honk
"Deep Residual Learning for Small-Footprint Keyword Spotting" re-implementation), on the generic 2-D graph engine.

`Res8(inputs, num_classes)` returns (logits, endpoints) like the reference; the topology below is built once per shape
(cached like a TF variable scope) through Graph2D's node builders, under the reference's variable names
(`Res/f_conv/weights`, `Res/conv{i}/weights`, `Res/conv{i}_bn/moving_mean|moving_variance`, `Res/fc/weights`)."""
from __future__ import annotations

from typing import Optional

from .. import runtime
from ..engine import Graph2D
from . import tc_resnet

BN_DECAY, BN_EPS = 0.997, 0.001            # Res_arg_scope (:105-112); epsilon = slim.batch_norm default


def Res_arg_scope(is_training, weight_decay=0.00001):
    """Reference :105-123: BN without center / scale, decay 0.997; conv: Xavier, ReLU, no bias, no normalizer, SAME."""
    return {"is_training": bool(is_training), "weight_decay": float(weight_decay), "keep_prob": 1.0, "bn_decay": BN_DECAY, "bn_eps": BN_EPS}


def build_resnet(g: Graph2D, num_classes: int, num_layers: int, num_channels: int, pool_size, use_dilation: bool, scope: str = "Res") -> int:
    """resnet() (:29-66) as graph nodes; returns the logits node."""
    def conv_relu_bn(net, idx, bn):                                     # conv_relu_bn (:6-26)
        rate = int(2 ** (idx // 3)) if use_dilation else 1
        net = g.conv(net, 3, num_channels, f"{scope}/conv{idx}/weights", stride=1, rate=rate, relu=True)
        if bn:
            net = g.batch_norm(net, f"{scope}/conv{idx}_bn", center=False, scale=False, decay=BN_DECAY, eps=BN_EPS)
        return net

    net = g.conv(-1, 3, num_channels, f"{scope}/f_conv/weights", stride=1, relu=True)
    if pool_size:
        net = g.pool(net, "avg", tuple(pool_size), stride=1, padding="VALID")       # slim.avg_pool2d default padding
    idx = 0
    for i in range(num_layers // 2):
        layer_in = net
        net = conv_relu_bn(net, idx, True)
        idx += 1
        net = conv_relu_bn(net, 2 * i + 1, False)
        idx += 1
        net = g.add(net, layer_in)
        net = g.batch_norm(net, f"{scope}/conv{2 * i + 1}_bn", center=False, scale=False, decay=BN_DECAY, eps=BN_EPS)
    if num_layers % 2 != 0:
        net = conv_relu_bn(net, idx, True)
    net = g.pool(net, "avg", None)                                       # kernel_size = net.shape[1:3]
    return g.conv(net, 1, num_classes, f"{scope}/fc/weights")            # activation_fn=None, no bias


_VARIANTS = {           # name -> (num_layers, num_channels, pool_size, use_dilation)   (:69-102)
    "Res8": (6, 45, [4, 3], False),
    "Res8Narrow": (6, 19, [4, 3], False),
    "Res15": (13, 45, None, True),
    "Res15Narrow": (13, 19, None, True),
}


def get_engine(variant: str, h: int, w: int, num_classes: int) -> Graph2D:
    key = ("Res", variant, h, w, num_classes, id(runtime.default_lib()))
    eng = tc_resnet._engines.get(key)
    if eng is None:
        eng = Graph2D("Res", h, w, 1, lib=runtime.default_lib(), device=runtime.default_device())
        layers, channels, pool, dil = _VARIANTS[variant]
        eng.finalize(build_resnet(eng, num_classes, layers, channels, pool, dil))
        tc_resnet._engines[key] = eng
    return eng


def _run(variant: str, inputs, num_classes: int, planar=None):
    eng = get_engine(variant, int(inputs.shape[1]), int(inputs.shape[2]), num_classes)
    if tc_resnet.current_scope()["is_training"]:
        raise RuntimeError("train-mode graphs are driven by AudioNetModel.build/train_step (they need labels)")
    logits, probs = eng.forward_infer(planar if planar is not None else tc_resnet._planar_of(inputs))
    return logits, {"softmax": probs, "engine": eng}


def Res8(inputs, num_classes, planar=None):
    return _run("Res8", inputs, num_classes, planar)


def Res8Narrow(inputs, num_classes, planar=None):
    return _run("Res8Narrow", inputs, num_classes, planar)


def Res15(inputs, num_classes, planar=None):
    return _run("Res15", inputs, num_classes, planar)


def Res15Narrow(inputs, num_classes, planar=None):
    return _run("Res15Narrow", inputs, num_classes, planar)
