"""The KWSModel architectures with the reference's names (audio_nets/kws.py:15-757; "Convolutional Neural Networks for
Small-footprint Keyword Spotting" models as in tensorflow/examples/speech_commands): `create_model(fingerprint_input,
model_settings, model_architecture, is_training)` -> logits.  Built on the generic 2-D graph engine under the reference's
variable names (first_weights, first_bias, second_weights, ..., final_fc_weights, final_fc_bias; truncated-normal init).

A TF `reshape(x, [-1, H*W*C]) @ W[H*W*C, N]` is the VALID convolution of the [H, W, C] activation with W viewed as
[H, W, C, N] -- the flatten order of NHWC and the HWIO weight layout coincide -- so every matmul is a conv node.

`low_latency_svdf` (kws.py:490-680) is built as the reference's TRAINING graph computes it -- every frame of the window, no runtime memory --:
frequency filters = a 1 x F VALID conv, `Graph2D.time_filter` = the per-filter time filters, `Graph2D.group_sum` = rank sum + bias + ReLU, three
fully-connected layers.  The reference's evaluation branch streams through a batch-1 `runtime-memory` variable and cannot even be built there
(`runtime_settings` is None at :576); here evaluation runs the same all-frames graph without dropout, and `runtime-memory` is carried as a
constant zero variable so that checkpoints hold every global variable of the reference graph."""
from __future__ import annotations

import numpy as np

import math

from .. import runtime
from ..engine import Graph2D
from . import tc_resnet

ARCHITECTURES = ["single_fc", "conv", "low_latency_conv", "low_latency_svdf", "tiny_conv", "one_fstride4", "trad_fpool3"]
TN = ("truncated_normal", 0.01)


def _fc(g: Graph2D, net: int, units: int, name: str, bias: bool, init=TN) -> int:
    """tf.matmul(flattened, weights) [+ bias] over whatever [c, h, w] the input node has."""
    c, h, w = g.shape(net)
    # (the variable is the reference's 2-D [K, N] matmul weight -- kws.py:79-81, 149-153 ...: NHWC flatten order == the [h, w, c, N] kernel layout)
    return g.conv(net, (h, w), units, name, padding="VALID", biases_name=(name.replace("weights", "bias") if bias else None), init=init,
                  tf_shape=(h * w * c, units))


def build_model(g: Graph2D, model_settings, model_architecture: str) -> int:
    t, f, nc = model_settings["spectrogram_length"], model_settings["fingerprint_width"], model_settings["label_count"]
    drop = lambda n: g.dropout(n, 0.5)                                   # tf.nn.dropout(x, 0.5) when is_training
    if model_architecture == "single_fc":                                 # :65-95
        return g.conv(-1, (t, f), nc, "weights", padding="VALID", biases_name="bias", init=("truncated_normal", 0.001),
                      tf_shape=(t * f, nc))                               # [fingerprint_size, label_count] (:79-81)
    if model_architecture == "conv":                                      # :98-201
        net = drop(g.conv(-1, (20, 8), 64, "first_weights", padding="SAME", relu=True, biases_name="first_bias", init=TN))
        net = g.pool(net, "max", (2, 2), stride=(2, 2), padding="SAME")
        net = drop(g.conv(net, (10, 4), 64, "second_weights", padding="SAME", relu=True, biases_name="second_bias", init=TN))
        return _fc(g, net, nc, "final_fc_weights", True)
    if model_architecture == "trad_fpool3":                               # :204-290
        net = drop(g.conv(-1, (20, 8), 64, "first_weights", padding="VALID", relu=True, init=TN))
        net = g.pool(net, "max", (1, 3), stride=(1, 3), padding="VALID")
        net = drop(g.conv(net, (10, 4), 64, "second_weights", padding="VALID", relu=True, init=TN))
        net = _fc(g, net, 32, "linear_weights", False)
        net = _fc(g, net, 128, "first_fc_weights", False)
        return _fc(g, net, nc, "final_fc_weights", False)
    if model_architecture == "low_latency_conv":                          # :293-405
        net = drop(g.conv(-1, (t, 8), 186, "first_weights", padding="VALID", relu=True, biases_name="first_bias", init=TN))
        net = drop(_fc(g, net, 128, "first_fc_weights", True))
        net = drop(_fc(g, net, 128, "second_fc_weights", True))
        return _fc(g, net, nc, "final_fc_weights", True)
    if model_architecture == "one_fstride4":                              # :408-487
        net = drop(g.conv(-1, (t, 8), 186, "first_weights", stride=(1, 4), padding="VALID", relu=True, init=TN))
        net = _fc(g, net, 32, "first_linear_weights", False)
        net = drop(_fc(g, net, 128, "first_fc_weights", False))
        net = drop(_fc(g, net, 128, "second_fc_weights", False))
        return _fc(g, net, nc, "final_fc_weights", False)
    if model_architecture == "tiny_conv":                                 # :681-757
        net = drop(g.conv(-1, (10, 8), 8, "first_weights", stride=(2, 2), padding="SAME", relu=True, biases_name="first_bias", init=TN))
        return _fc(g, net, nc, "final_fc_weights", True)
    if model_architecture == "low_latency_svdf":                          # :490-680, the training graph (every frame, no runtime memory)
        rank, num_units = 2, 1280
        nf = rank * num_units
        # frequency filters: tf.nn.conv1d over [B, T * F, 1] with stride F == a 1 x F VALID conv of the [T x F] plane (:586-596)
        net = g.conv(-1, (1, f), nf, "weights_frequency", padding="VALID", init=TN, tf_shape=(f, nf))
        net = g.time_filter(net, "weights_time", init=TN)                 # [filters, T] time filters (:604-612)
        net = drop(g.group_sum(net, rank, relu=True, biases_name="bias"))  # rank sum + bias + relu (+ dropout) (:613-634)
        net = drop(_fc(g, net, 256, "first_fc_weights", True))
        net = drop(_fc(g, net, 256, "second_fc_weights", True))
        # `runtime-memory` [filters, 1, T]: the streaming-inference state (trainable=False, zeros; only the deploy-time branch writes it)
        g.constants["runtime-memory"] = np.zeros((nf, 1, t), np.float32)
        return _fc(g, net, nc, "final_fc_weights", True)
    raise Exception('model_architecture argument "' + model_architecture + '" not recognized, should be one of "single_fc", "conv",'
                    ' "low_latency_conv, "one_fstride4", "trad_fpool3", "low_latency_svdf" or "tiny_conv"')


def get_engine(model_settings, model_architecture: str) -> Graph2D:
    t, f, nc = model_settings["spectrogram_length"], model_settings["fingerprint_width"], model_settings["label_count"]
    key = ("KWS", model_architecture, t, f, nc, id(runtime.default_lib()))
    eng = tc_resnet._engines.get(key)
    if eng is None:
        eng = Graph2D("", t, f, 1, lib=runtime.default_lib(), device=runtime.default_device())
        eng.finalize(build_model(eng, model_settings, model_architecture))
        tc_resnet._engines[key] = eng
    return eng


def create_model(fingerprint_input, model_settings, model_architecture, is_training, runtime_settings=None, planar=None):
    """Reference :15-63.  Eval-mode forward; training goes through AudioNetModel.build / train_step."""
    eng = get_engine(model_settings, model_architecture)
    if is_training:
        raise RuntimeError("train-mode graphs are driven by AudioNetModel.build/train_step (they need labels)")
    logits, probs = eng.forward_infer(planar if planar is not None else tc_resnet._planar_of(fingerprint_input))
    create_model.last = {"softmax": probs, "engine": eng}
    return logits
