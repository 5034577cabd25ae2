"""The KWSModel architectures with the reference's names (audio_nets/kws.py:15-757; "Convolutional Neural Networks for
Small-footprint Keyword Spotting" models as in tensorflow/examples/speech_commands): `create_model(fingerprint_input,
model_settings, model_architecture, is_training)` -> logits.  Built on the generic 2-D graph engine under the reference's
variable names (first_weights, first_bias, second_weights, ..., final_fc_weights, final_fc_bias; truncated-normal init).

A TF `reshape(x, [-1, H*W*C]) @ W[H*W*C, N]` is the VALID convolution of the [H, W, C] activation with W viewed as
[H, W, C, N] -- the flatten order of NHWC and the HWIO weight layout coincide -- so every matmul is a conv node.

`low_latency_svdf` is not built: its graph keeps a [num_filters, batch = 1, time] runtime-memory variable and only runs at
batch 1 (kws.py:490-680; `runtime_settings` is None at :576, so the reference cannot build it for evaluation); none of the reference's scripts uses it."""
from __future__ import annotations

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
    if model_architecture == "low_latency_svdf":
        raise NotImplementedError("low_latency_svdf keeps a batch-1 runtime-memory variable (kws.py:490-680; `runtime_settings` is None at :576, so the reference cannot build it for evaluation) and is not built")
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
