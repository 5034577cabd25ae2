"""Model classes: the drop-in boundary of the reference (factory/audio_nets.py:19-187,341-409).

Same class names, constructor `(args, dataset=None)`, `add_arguments`, `build(wavs, labels, is_training)` and the
attributes the trainer/evaluator read afterwards (`audio`, `audio_original`, `inputs`, `logits`, `outputs`,
`labels`, `endpoints`, `total_loss`, `model_loss`, `endpoints_loss`, `total_params`,
`input_preprocessors_for_tflite`, `is_training`).  The reference builds a TF graph that a session runs later;
here `build()` executes the HIP kernels at once, and `train_step()` is what the trainer's `session.run(train_op)`
becomes (fwd with batch statistics + bwd + optimiser + moving-average update in one call).
"""
from __future__ import annotations

import logging
from typing import Dict, Tuple

import torch

from .. import runtime
from ..audio_nets import kws, res, tc_resnet
from ..datasets import preprocessor_factory
from ..common import tf_utils
from ..parallel import DataParallel
from .base import TFModel

_available_nets = [
    "KWSModel", "Res8Model", "Res8NarrowModel", "Res15Model", "Res15NarrowModel", "DSCNNSModel", "DSCNNMModel",
    "DSCNNLModel", "TCResNet8Model", "TCResNet14Model", "ResNet2D8Model", "ResNet2D8PoolModel",
]


class AudioNetModel(TFModel):
    def __init__(self, args, dataset=None):
        self.log = logging.getLogger("AudioNetModel")
        self.dataset = dataset
        self.args = args
        self.engine = None
        self._dp = None
        self._step = 0

    # ---- reference surface ---------------------------------------------------------------------
    def build(self, wavs, labels, is_training):
        self._audio_original = wavs
        self.is_training = is_training
        self.labels = labels
        self.preprocess_input()
        self.inputs, self.logits, self._outputs, self.endpoints = self.build_output(self.audio, self.is_training, self.args.output_name)
        self._total_loss, self._model_loss, self.endpoints_loss = self.build_loss(self.logits, self.outputs, self.labels)
        self.total_params = tf_utils.show_models(self.log, self.engine)

    def preprocess_input(self, for_deploy=False):
        window_size_samples = int(self.args.sample_rate * self.args.window_size_ms / 1000)
        window_stride_samples = int(self.args.sample_rate * self.args.window_stride_ms / 1000)
        if not hasattr(self, "_preprocessor"):
            self._preprocessor = preprocessor_factory.factory(
                preprocess_method=self.args.preprocess_method, scope="input/audio/preprocessing",
                preprocessed_node_name="input/audio/preprocessed")
        self._audio = self._preprocessor.preprocess(
            self._audio_original, window_size_samples=window_size_samples, window_stride_samples=window_stride_samples,
            for_deploy=for_deploy, **vars(self.args))
        self.args.height, self.args.width, self.args.channels = [int(x) for x in self._audio.shape[1:4]]
        self.input_preprocessors_for_tflite = [self._preprocessor]

    def build_deployable_model(self, include_preprocess=True):
        """Reference :87-125.  Returns (input_tensors, output_tensor): specs of the deploy graph's placeholder(s) and of the
        node named --output_name.  include_preprocess=True: waveform `input/audio/before_preprocessing`
        [input_batch_size, samples, 1] through the DEPLOY-path MFCC (for_deploy=True); False: features `input`
        [1, height, width, channels].  The graph is built in eval mode on a zero input (this is what creates the variables);
        `freeze()` then converts the current variables to constants (deploy.FrozenModel)."""
        from ..deploy import TensorSpec
        dev = runtime.default_device() or ("cuda" if runtime.default_lib() is None or runtime.default_lib().kind == "hip" else "cpu")
        self.is_training = False
        self.labels = None
        if include_preprocess:
            desired_samples = int(self.args.sample_rate * self.args.clip_duration_ms / 1000)
            spec = TensorSpec("input/audio/before_preprocessing", (int(self.args.input_batch_size), desired_samples, 1))
            self._audio_original = torch.zeros(spec.shape, dtype=torch.float32, device=dev)
            self.preprocess_input(for_deploy=True)
            inputs = self._audio
        else:
            self.log.info("Build graph which excludes preprocessing for freezing!")
            assert self.args.height > 0 and self.args.width > 0 and self.args.channels > 0
            spec = TensorSpec("input", (1, int(self.args.height), int(self.args.width), int(self.args.channels)))
            self._preprocessor = preprocessor_factory.factory("no_preprocessing", "input/audio/preprocessing", "input/audio/preprocessed")
            inputs = self._preprocessor.preprocess(torch.zeros(spec.shape, dtype=torch.float32, device=dev))
            self._audio = inputs
        _, _, out, _ = self.build_output(inputs, False, self.args.output_name)
        self._deploy = (bool(include_preprocess), [spec], TensorSpec(self.args.output_name, tuple(int(d) for d in out.shape)))
        return self._deploy[1], self._deploy[2]

    def freeze(self):
        """graph_util.convert_variables_to_constants (freeze.py:36-40) for the deployable model built last."""
        from ..deploy import export_frozen
        include_preprocess, inputs, output = self._deploy
        return export_frozen(self, include_preprocess, inputs, output)

    @property
    def model_loss(self):
        return self._model_loss

    @property
    def total_loss(self):
        return self._total_loss

    @property
    def audio_original(self):
        return self._audio_original

    @property
    def audio(self):
        return self._audio

    @property
    def outputs(self):
        return self._outputs

    def build_output(self, inputs, is_training, output_name):
        logits, endpoints = self.build_inference(inputs, is_training=is_training)
        return inputs, logits, self._probs, endpoints          # self._probs: slim.softmax(logits), the node named output_name

    def build_inference(self, inputs, is_training=True):
        raise NotImplementedError

    def build_loss(self, logits, scores, labels) -> Tuple[torch.Tensor, torch.Tensor, Dict]:
        """model_loss = mean softmax cross-entropy; total = model + weight_decay * sum l2_loss(non-BN trainables)
        (reference :161-183).  In eval builds the cross-entropy is evaluated from the softmax outputs."""
        if self._loss_sum is not None:
            model_loss = self._mean_loss
        else:
            eng, b = self.engine, int(logits.shape[0])
            scratch = torch.empty(b + 1, dtype=torch.float32, device=logits.device)
            lg, lb = logits.contiguous(), labels.to(torch.float32).contiguous()
            eng.lib.check(eng.lib.tcr_xent_loss_sum(lg.data_ptr(), lb.data_ptr(), b, int(logits.shape[1]), float(getattr(self.args, "label_smoothing", 0.0)),
                                                    scratch.data_ptr(), scratch[b:].data_ptr(), eng._stream()), "tcr_xent_loss_sum")
            model_loss = scratch[b] / float(b)
        total = model_loss + self.engine.l2_loss(self.args.weight_decay)
        return total, model_loss, {}

    @staticmethod
    def add_arguments(parser):
        parser.add_argument("--label_smoothing", default=0.0, type=float)

    # ---- what session.run(train_op) becomes ------------------------------------------------------
    def data_parallel(self, sync_bn: bool = False) -> DataParallel:
        if self._dp is None or self._dp.sync_bn != sync_bn:
            self._dp = DataParallel(self.engine, sync_bn=sync_bn)
        return self._dp

    def set_step(self, step: int):
        """Continue the step counter of a restored run: Adam's bias-correction power t and the dropout stream are functions of it."""
        self._step = int(step)

    def step_count(self) -> int:
        return self._step

    def set_trainable_scopes(self, trainable_scopes: str, logger=None):
        """`--trainable_scopes a,b`: tf_utils.get_variables_to_train (common/tf_utils.py:19-37 of the reference) -- the trainable
        variables whose name re.match-es one of the comma-separated scopes (tf.get_collection's filter); '' trains everything.
        Every other variable is left out of the optimiser's var_list (helper/trainer.py:199-222): no update, no slot update, no
        weight decay step (its L2 term still counts in total_loss); BN moving statistics keep updating (slim's update ops)."""
        import re
        scopes = [sc.strip() for sc in trainable_scopes.split(",")] if trainable_scopes else []
        names = [n for n, ti in self.engine.tensors.items() if ti.arena == 0]
        train = [n for n in names if any(re.match(sc, n) for sc in scopes)] if scopes else names
        self._frozen = [(int(ti.offset), int(ti.size)) for n, ti in self.engine.tensors.items() if ti.arena == 0 and n not in set(train)]
        self._trained_names = set(train)                # (optimiser slots / EMA shadows exist for these only: tf.train.Optimizer.minimize(var_list=...))
        self._train_nothing = bool(scopes) and not train
        for n in (train if scopes else []):
            if logger is not None:
                logger.info("vars to train > %s", n)
        return train

    _OPT_SLOTS = {"mom": ("Momentum",), "gd": ("Momentum",), "adam": ("Adam", "Adam_1"), "rmsprop": ("RMSProp", "RMSProp_1")}

    def _frozen_arenas(self, optimizer, ema):
        eng = self.engine
        return [eng.params] + [eng.slot_arena(n) for n in self._OPT_SLOTS[optimizer] + (("ExponentialMovingAverage",) if ema else ())]

    def train_step(self, wavs, labels, learning_rate, optimizer="mom", momentum=None, decay=None, epsilon=None, sync_bn=False,
                   ema_decay=None):
        """One optimisation step on a (local shard of a) batch; returns (total_loss, model_loss) as device scalars.
        momentum / decay / epsilon: the tf.train optimiser constructor overrides (None = TF's default:
        Momentum needs `momentum`; Adam beta1 .9, beta2 .999, epsilon 1e-8; RMSProp decay .9, momentum 0, epsilon 1e-10)."""
        self._audio_original, self.labels, self.is_training = wavs, labels, True
        self.preprocess_input()
        if getattr(self, "_train_nothing", False):
            # Empty variables_to_train: the reference's train op is tf.no_op() (helper/trainer.py:220-222) -- no optimiser, no UPDATE_OPS
            # (BN moving averages untouched), no slot, only the losses are fetched: the training-graph forward with everything it
            # would have written put back.
            saved = self.engine.stats.clone()
            dp0 = self.data_parallel(sync_bn)
            # (fresh dropout per session.run, as in the reference: a counter of its own -- self._step only moves with optimiser steps --
            #  offset so that it never shares a seed with an optimisation step of the same run)
            self._noop_step = getattr(self, "_noop_step", 0) + 1
            self.logits, self._outputs, loss_sum = dp0.forward_train(
                self._preprocessor.planar, labels, keep_prob=self._keep_prob(), seed=self._step + (1 << 24) + self._noop_step,
                label_smoothing=float(getattr(self.args, "label_smoothing", 0.0)))
            self.engine.stats.copy_(saved)
            self._model_loss = dp0.mean_loss(loss_sum, wavs.shape[0])
            self._total_loss = self._model_loss + self.engine.l2_loss(self.args.weight_decay)
            return self._total_loss, self._model_loss
        self._step += 1
        dp = self.data_parallel(sync_bn)
        b = wavs.shape[0]
        self.logits, self._outputs, loss_sum = dp.forward_train(
            self._preprocessor.planar, labels, keep_prob=self._keep_prob(), seed=self._step,
            label_smoothing=float(getattr(self.args, "label_smoothing", 0.0)))
        dp.backward()
        l2 = self.engine.l2_loss(self.args.weight_decay)
        wd = float(self.args.weight_decay)
        frozen = getattr(self, "_frozen", None)
        stash = None
        if frozen:      # variables outside --trainable_scopes: the arena-wide optimiser kernels run, then their slices are put back
            stash = [[a[o:o + n].clone() for o, n in frozen] for a in self._frozen_arenas(optimizer, ema_decay is not None)]
        if optimizer == "mom":
            self.engine.sgd_momentum_step(learning_rate, 0.9 if momentum is None else momentum, wd)
        elif optimizer == "gd":
            self.engine.sgd_momentum_step(learning_rate, 0.0, wd)
        elif optimizer == "adam":
            self.engine.adam_step(learning_rate, self._step, 0.9, 0.999, 1e-8 if epsilon is None else epsilon, wd)
        elif optimizer == "rmsprop":
            self.engine.rmsprop_step(learning_rate, 0.9 if decay is None else decay, 0.0 if momentum is None else momentum,
                                     1e-10 if epsilon is None else epsilon, wd)
        else:
            raise NotImplementedError(f"optimizer {optimizer}")
        if ema_decay is not None:
            self.engine.ema_step(ema_decay)
        if stash is not None:
            for a, saved in zip(self._frozen_arenas(optimizer, ema_decay is not None), stash):
                for (o, n), v in zip(frozen, saved):
                    a[o:o + n].copy_(v)
        self._model_loss = dp.mean_loss(loss_sum, b)
        self._total_loss = self._model_loss + l2
        return self._total_loss, self._model_loss

    def _keep_prob(self) -> float:
        return 1.0


class _TCResNetModel(AudioNetModel):
    scope = None
    base_channels = None

    @staticmethod
    def add_arguments(parser):
        parser.add_argument("--weight_decay", default=0.0001, type=float)
        parser.add_argument("--dropout_keep_prob", default=0.5, type=float)
        parser.add_argument("--width_multiplier", default=1.0, type=float)

    def _keep_prob(self) -> float:
        return float(self.args.dropout_keep_prob)

    def build_inference(self, inputs, is_training):
        scope = tc_resnet.TCResNet_arg_scope(is_training=is_training, weight_decay=self.args.weight_decay,
                                             keep_prob=self.args.dropout_keep_prob)
        channels = tc_resnet.tcresnet_channels(self.base_channels, self.args.width_multiplier)
        self.engine = tc_resnet.get_engine(self.scope, channels, int(inputs.shape[2]), int(inputs.shape[1]),
                                           self.args.num_classes, scope["bn_decay"], scope["bn_eps"])
        planar = self._preprocessor.planar
        self._loss_sum = None
        if is_training:
            # the reference's training graph: batch-statistics BN, dropout, CE (moving averages are only
            # assigned by the train op, so this read-only build restores them afterwards)
            saved = self.engine.stats.clone()
            logits, probs, loss_sum = self.engine.forward_train(planar, self.labels, keep_prob=self._keep_prob(), seed=0,
                                                                label_smoothing=float(getattr(self.args, "label_smoothing", 0.0)))
            self.engine.stats.copy_(saved)
            self._loss_sum, self._mean_loss = loss_sum, loss_sum / float(planar.shape[0])
            self._probs = probs
            return logits, {"engine": self.engine}
        with tc_resnet.arg_scope(scope):
            builder = tc_resnet.TCResNet8 if self.scope == "TCResNet8" else tc_resnet.TCResNet14
            logits, endpoints = builder(inputs, self.args.num_classes, width_multiplier=self.args.width_multiplier, planar=planar)
        self._probs = endpoints["softmax"]
        return logits, endpoints


class TCResNet8Model(_TCResNetModel):
    scope = "TCResNet8"
    base_channels = [16, 24, 32, 48]


class TCResNet14Model(_TCResNetModel):
    scope = "TCResNet14"
    base_channels = [16, 24, 24, 32, 32, 48, 48]


class _DSCNNModel(AudioNetModel):
    """DS-CNN S / M / L (factory/audio_nets.py:299-359 of the reference) on the HIP kernels: eval-mode and training
    builds (BN without scale, conv biases; the reference's scripts train this family with --optimizer adam)."""
    size = None

    @staticmethod
    def add_arguments(parser):
        parser.add_argument("--weight_decay", default=0.0, type=float)

    def build_inference(self, inputs, is_training):
        from ..engine import DSCNN
        key = ("DSCNN", self.size, int(inputs.shape[1]), int(inputs.shape[2]), self.args.num_classes, id(runtime.default_lib()))
        eng = tc_resnet._engines.get(key)
        if eng is None:
            eng = DSCNN(self.size, int(inputs.shape[1]), int(inputs.shape[2]), self.args.num_classes,
                        lib=runtime.default_lib(), device=runtime.default_device())
            eng.init_xavier(0)
            tc_resnet._engines[key] = eng
        self.engine = eng
        self._loss_sum = None
        if is_training:     # read-only build of the training graph (see _TCResNetModel.build_inference)
            planar = self._preprocessor.planar
            saved = eng.stats.clone()
            logits, probs, loss_sum = eng.forward_train(planar, self.labels, label_smoothing=float(getattr(self.args, "label_smoothing", 0.0)))
            eng.stats.copy_(saved)
            self._loss_sum, self._mean_loss = loss_sum, loss_sum / float(planar.shape[0])
            self._probs = probs
            return logits, {"engine": eng}
        logits, probs = eng.forward_infer(self._preprocessor.planar)
        self._probs = probs
        return logits, {"engine": eng}


class DSCNNSModel(_DSCNNModel):
    size = "S"


class DSCNNMModel(_DSCNNModel):
    size = "M"


class DSCNNLModel(_DSCNNModel):
    size = "L"


class _GraphModel(AudioNetModel):
    """Model classes whose network runs on the generic 2-D graph engine (engine.Graph2D)."""

    def _engine_for(self, inputs):
        raise NotImplementedError

    def build_inference(self, inputs, is_training=True):
        eng = self._engine_for(inputs)
        self.engine = eng
        planar = self._preprocessor.planar
        self._loss_sum = None
        if is_training:     # read-only build of the training graph (see _TCResNetModel.build_inference)
            saved = eng.stats.clone()
            logits, probs, loss_sum = eng.forward_train(planar, self.labels, seed=0, label_smoothing=float(getattr(self.args, "label_smoothing", 0.0)))
            eng.stats.copy_(saved)
            self._loss_sum, self._mean_loss = loss_sum, loss_sum / float(planar.shape[0])
        else:
            logits, probs = eng.forward_infer(planar)
        self._probs = probs
        return logits, {"engine": eng}


class GoogleKWS:
    def __init__(self, args):
        self.args = args

    def build_model_settings(self, inputs):
        """Reference :194-202."""
        t, f = int(inputs.shape[1]), int(inputs.shape[2])
        return {"fingerprint_width": f, "spectrogram_length": t, "fingerprint_size": t * f, "label_count": self.args.num_classes,
                "sample_rate": self.args.sample_rate, "window_stride_samples": int(self.args.sample_rate * self.args.window_stride_ms / 1000)}


class KWSModel(_GraphModel):
    """factory/audio_nets.py:205-225 -> audio_nets/kws.py; `--architecture` as in the reference (all seven)."""

    def __init__(self, args, dataset=None):
        super().__init__(args, dataset)
        self.google_kws = GoogleKWS(args)

    @staticmethod
    def add_arguments(parser):
        parser.add_argument("--architecture", default="conv", choices=kws.ARCHITECTURES)

    def _engine_for(self, inputs):
        return kws.get_engine(self.google_kws.build_model_settings(inputs), self.args.architecture)


class _ResModel(_GraphModel):
    variant = None

    @staticmethod
    def add_arguments(parser):
        parser.add_argument("--weight_decay", default=0.00001, type=float)

    def _engine_for(self, inputs):
        res.Res_arg_scope(is_training=self.is_training, weight_decay=self.args.weight_decay)
        return res.get_engine(self.variant, int(inputs.shape[1]), int(inputs.shape[2]), self.args.num_classes)


class Res8Model(_ResModel):
    variant = "Res8"


class Res8NarrowModel(_ResModel):
    variant = "Res8Narrow"


class Res15Model(_ResModel):
    variant = "Res15"


class Res15NarrowModel(_ResModel):
    variant = "Res15Narrow"


class _ResNet2D8Model(_GraphModel):
    scope = None
    pool = None

    @staticmethod
    def add_arguments(parser):
        parser.add_argument("--weight_decay", default=0.0001, type=float)
        parser.add_argument("--dropout_keep_prob", default=0.5, type=float)
        parser.add_argument("--width_multiplier", default=1.0, type=float)

    def _engine_for(self, inputs):
        return tc_resnet.get_engine_2d(self.scope, int(inputs.shape[1]), int(inputs.shape[2]), self.args.num_classes, self.args.width_multiplier,
                                       self.args.dropout_keep_prob, self.pool)


class ResNet2D8Model(_ResNet2D8Model):
    scope = "ResNet2D8"


class ResNet2D8PoolModel(_ResNet2D8Model):
    scope = "ResNet2D8Pool"
    pool = ([4, 4], 4)
