"""Training loop with the reference trainer's behaviour (helper/trainer.py:24-599):
piecewise-constant LR over global_step (absolute / relative to the restored step, boundaries in steps or epochs),
gd / adam / mom / rmsprop with the optional --momentum / --optimizer_decay / --optimizer_epsilon overrides, optional
ExponentialMovingAverage of the trainables, one fused step per iteration (what session.run(train_op) was), step timing,
TF-format checkpoints `<train_dir>/<ModelName>-<global_step>.{index,data-00000-of-00001}` every --step_save_checkpoint
steps (+ the `checkpoint` state file, --max_to_keep), in-training evaluation on the training graph every
--step_evaluation steps, and the --max_step_from_restore / --max_epoch_from_restore stop rules."""
from __future__ import annotations

import logging
import os
import sys
import time
from pathlib import Path
from typing import Callable, Dict

import numpy as np
import torch
import torch.distributed as dist

from ..common import tf_bundle, tf_utils
from ..common.model_loader import EMA_SUFFIX, Ckpt
from .base import AudioBase

# TF slot names per optimiser (tf.train.*Optimizer): engine slot arena -> suffix of the per-variable checkpoint entry
_SLOTS = {"mom": ["Momentum"], "gd": [], "adam": ["Adam", "Adam_1"], "rmsprop": ["RMSProp", "RMSProp_1"]}


def piecewise_constant(step: int, boundaries, values) -> float:
    """tf.train.piecewise_constant: x <= b0 -> v0; b0 < x <= b1 -> v1; ... (helper/trainer.py:135-137)."""
    for b, v in zip(boundaries, values):
        if step <= b:
            return v
    return values[-1]


def positive_int(v):
    iv = int(v)
    if iv <= 0:
        raise ValueError(f"{v} is not a positive integer")      # argparse reports it as an invalid value (common/utils.py positive_int)
    return iv


class TrainerBase(AudioBase):
    def __init__(self, model, session, args, dataset, dataset_name, name="TrainerBase"):
        self.model, self.session, self.args, self.dataset, self.dataset_name = model, session, args, dataset, dataset_name
        self.log = logging.getLogger(name)
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.input_shape = list(model.audio.shape)
        self.last_loss: Dict[str, float] = {}

    # ---- setup ---------------------------------------------------------------------------------------------------
    def setup_essentials(self, max_to_keep=5):
        a = self.args
        a.checkpoint_path = tf_utils.resolve_checkpoint_path(getattr(a, "checkpoint_path", ""), self.log, is_training=True)
        self.train_dir_name = (Path.cwd() / Path(a.train_dir)).resolve()
        self.global_step_from_checkpoint = tf_utils.get_global_step_from_checkpoint(a.checkpoint_path)
        self.global_step = self.global_step_from_checkpoint
        if a.boundaries_epoch:
            # (an optimisation step consumes batch_size samples on EVERY data-parallel rank: epochs count the global batch)
            boundaries = [b * self.dataset.num_samples // (self.dataset.batch_size * self.world) for b in a.boundaries]
        else:
            boundaries = list(a.boundaries)
        if a.relative:
            self.boundaries = [self.global_step_from_checkpoint + b for b in boundaries]
            self.log.info("global_step starts with %d, so, update boundaries %s to %s", self.global_step_from_checkpoint, boundaries,
                          self.boundaries)
        else:
            self.boundaries = boundaries
        if len(a.lr_list) != len(self.boundaries) + 1:
            raise ValueError(f"--lr_list needs {len(self.boundaries) + 1} values for {len(self.boundaries)} boundaries "
                             f"(tf.train.piecewise_constant), got {len(a.lr_list)}")
        self._kept = []

    def build_optimizer(self, optimizer, momentum=None, decay=None, epsilon=None) -> Dict[str, float]:
        """The keyword overrides the reference forwards to the tf.train optimiser constructor: only the truthy ones
        (helper/trainer.py:171-197); everything else keeps TF's default."""
        if optimizer not in _SLOTS:
            self.log.error("Unknown optimizer: %s", optimizer)
            raise NotImplementedError(optimizer)
        kw = {}
        if momentum:
            kw["momentum"] = float(momentum)
        if decay:
            kw["decay"] = float(decay)
        if epsilon:
            kw["epsilon"] = float(epsilon)
        allowed = {"gd": set(), "adam": {"epsilon"}, "mom": {"momentum"}, "rmsprop": {"momentum", "decay", "epsilon"}}[optimizer]
        bad = set(kw) - allowed
        if bad or (optimizer == "mom" and "momentum" not in kw):
            # tf.train.<Optimizer>(**kwargs) raises TypeError on an unexpected / missing constructor argument
            raise TypeError(f"optimizer {optimizer}: unexpected or missing arguments {sorted(bad) or ['momentum']}")
        self.log.info("Use %s optimizer %s", optimizer, kw)
        return kw

    def setup_trainer(self):
        a = self.args
        self.setup_essentials(a.max_to_keep)
        self.optimizer_kwargs = self.build_optimizer(a.optimizer, momentum=a.momentum, decay=a.optimizer_decay, epsilon=a.optimizer_epsilon)
        if getattr(a, "trainable_scopes", ""):
            trained = self.model.set_trainable_scopes(a.trainable_scopes, self.log)
            if not trained:
                # the reference's train op is then tf.no_op() (:220-222): nothing is ever updated -- and its global_step never advances, so
                # its loop never ends.  Here the steps are forward-only as well (no variable, slot or moving statistic changes) but the
                # step counter still advances, so that --max_step_from_restore ends the run: the one deliberate deviation.
                self.log.info("Empty variables_to_train")
        if a.use_ema:
            self.model.engine.ema_init()
        self.routine_restore_and_initialize()

    # ---- checkpoints (TF bundle; variable names of SURVEY App. C) ---------------------------------------------------
    def _slot_views(self) -> Dict[str, torch.Tensor]:
        """`<var>/<slot>` -> view into the arena-shaped optimiser slot, for every trainable variable."""
        eng = self.model.engine
        out = {}
        names = list(_SLOTS[self.args.optimizer]) + (["ExponentialMovingAverage"] if self.args.use_ema else [])
        for slot in names:
            arena = eng.slot_arena(slot)
            trained = getattr(self.model, "_trained_names", None)       # --trainable_scopes: slots / EMA shadows exist for the trained variables only
            for name, ti in eng.tensors.items():
                if ti.arena == 0 and (trained is None or name in trained):
                    out[f"{name}/{slot}"] = arena[ti.offset:ti.offset + ti.size].view(*eng.tf_shape(name))    # slots have their variable's TF shape
        return out

    def _extra_variables(self) -> Dict[str, Callable[[np.ndarray], None]]:
        """Non-model variables of the training graph: global_step, optimiser slots, Adam's beta powers."""
        extra: Dict[str, Callable[[np.ndarray], None]] = {}

        def set_step(v):
            self.global_step = int(v)
        extra["global_step"] = set_step
        for key, view in self._slot_views().items():
            extra[key] = (lambda v, t=view: t.copy_(torch.as_tensor(np.asarray(v, np.float32)).reshape(t.shape).to(t.device)))
        if self.args.optimizer == "adam":
            extra["beta1_power"] = lambda v: None       # beta^t is recomputed from global_step (the arena step counter below)
            extra["beta2_power"] = lambda v: None
        return extra

    def routine_restore_and_initialize(self, checkpoint_path=None):
        a = self.args
        if checkpoint_path is None:
            checkpoint_path = a.checkpoint_path
        if not checkpoint_path:
            self.log.info("Initialize global / local variables")
            return
        loader = Ckpt(self.model.engine, include_scopes=a.checkpoint_include_scopes, exclude_scopes=a.checkpoint_exclude_scopes,
                      ignore_missing_vars=a.ignore_missing_vars, logger=self.log)
        loader.load(checkpoint_path, extra=self._extra_variables())
        # Adam's bias correction and the dropout stream continue from the restored step (TF restores beta1_power / beta2_power)
        self.model.set_step(self.global_step)

    def save(self) -> str:
        """saver.save(session, train_dir/<model>, global_step): every global variable of the training graph."""
        a, eng = self.args, self.model.engine
        os.makedirs(a.train_dir, exist_ok=True)
        name = f"{a.model}-{self.global_step}"
        prefix = os.path.join(a.train_dir, name)
        sd = eng.state_dict()
        for key, view in self._slot_views().items():
            sd[key] = view.detach().cpu().numpy()
        sd["global_step"] = np.array(self.global_step, np.int64)
        if a.optimizer == "adam":
            t = self.model.step_count()
            sd["beta1_power"] = np.array(0.9 ** (t + 1), np.float32)      # AdamOptimizer: accumulator starts at beta, x beta per step
            sd["beta2_power"] = np.array(0.999 ** (t + 1), np.float32)
        tf_bundle.write_checkpoint(prefix, sd)
        if name in self._kept:
            self._kept.remove(name)
        self._kept.append(name)
        while len(self._kept) > int(a.max_to_keep):
            old = os.path.join(a.train_dir, self._kept.pop(0))
            for suffix in (".index", ".data-00000-of-00001"):
                if os.path.exists(old + suffix):
                    os.remove(old + suffix)
        tf_bundle.update_checkpoint_state(a.train_dir, name, list(self._kept))
        return prefix

    def _sync_replica_state(self):
        """Data parallel without --sync_bn: every replica moved its BN moving statistics towards its own shard's; average
        them so that the checkpoint (and every replica from here on) holds the statistics of the whole data."""
        if self.world > 1 and not bool(getattr(self.args, "sync_bn", False)):
            st = self.model.engine.stats
            dist.all_reduce(st, op=dist.ReduceOp.SUM)
            st.div_(float(self.world))

    # ---- loop ---------------------------------------------------------------------------------------------------------
    def build_epoch(self, step):
        """helper/trainer.py:104-105 of the reference, with the data-parallel global batch (world = 1: the reference's formula)."""
        return (step * self.dataset.batch_size * self.world) / self.dataset.num_samples

    def run_single_step(self):
        lr = piecewise_constant(self.global_step, self.boundaries, self.args.lr_list)
        wavs, labels = self.dataset.next_batch(self.rank, self.world)
        st = time.time()
        total, model = self.model.train_step(wavs, labels, lr, optimizer=self.args.optimizer, sync_bn=bool(getattr(self.args, "sync_bn", False)),
                                             ema_decay=self.args.ema_decay if self.args.use_ema else None, **self.optimizer_kwargs)
        self.global_step += 1
        return total, model, lr, st

    def train(self, name: str = "Training"):
        a = self.args
        self.log.info("%s started", name)
        while True:
            total, model, lr, st = self.run_single_step()
            global_step = self.global_step
            step_from_restore = global_step - self.global_step_from_checkpoint
            epoch_from_restore = self.build_epoch(step_from_restore)
            stop = epoch_from_restore >= a.max_epoch_from_restore or step_from_restore >= a.max_step_from_restore
            if step_from_restore % int(a.step_save_summaries) == 0 or stop:
                tl, ml = float(total), float(model)             # one host sync per logging interval
                dt = (time.time() - st) * 1e3
                if self.rank == 0:
                    self.log.info("[%s] GlobalStep / StepFromRestore / EpochFromRestore: %8d / %8d / %3.3f\tSingleStepPerInstance(ms) / "
                                  "SingleStep(ms): %.4f / %.3f (%s;%d)\tTotalLoss / ModelLoss: %7.4f / %7.4f | lr %.4g | %d params",
                                  self.dataset_name, global_step, step_from_restore, epoch_from_restore,
                                  dt / (self.dataset.batch_size * self.world), dt, self.input_shape, self.dataset.batch_size, tl, ml, lr,
                                  self.model.total_params)
                if not np.isfinite(tl):
                    raise FloatingPointError("total_loss is not finite")       # tf.check_numerics in create_train_op
            if step_from_restore % int(a.step_save_checkpoint) == 0 or stop:
                self._sync_replica_state()
                if self.rank == 0:
                    self.log.info("save checkpoint: %s", self.save())
            if step_from_restore % int(a.step_evaluation) == 0:
                self.evaluate(epoch_from_restore, step_from_restore, global_step, self.dataset_name)
            if epoch_from_restore >= a.max_epoch_from_restore:
                self.log.info("Reached %s epochs from restore.", a.max_epoch_from_restore)
                break
            if step_from_restore >= a.max_step_from_restore:
                self.log.info("Reached %s steps from restore.", a.max_step_from_restore)
                break
        self.log.info("%s finished", name)

    def evaluate(self, epoch_from_restore: float, step_from_restore: int, global_step: int, dataset_name: str, iters: int = None):
        """In-training evaluation ON THE TRAINING GRAPH (helper/trainer.py:436-460): `iters` more batches of the training
        input run forward with is_training=True (batch-statistics BN, dropout), no update; the metrics are logged."""
        iters = self.build_evaluate_iterations(iters)
        t0 = time.time()
        self._sync_replica_state()          # (replicas drift apart in their BN moving statistics between checkpoints)
        eval_dict = self.run_inference(global_step, iters=iters, is_training=True)
        self.last_eval = self.evaluate_metrics(eval_dict)
        if self.rank == 0:
            self.log.info("[%s-Evaluation] global_step / step_from_restore / epoch_from_restore: %8d / %5d / %3.3f (run_evaluation, "
                          "iterations: %d, %.3f s)", dataset_name, global_step, step_from_restore, epoch_from_restore, iters, time.time() - t0)
            self.log_metrics(global_step, self.last_eval)
        return self.last_eval

    @staticmethod
    def add_arguments(parser, name: str = "TrainerBase"):
        g = parser.add_argument_group(f"({name}) Optimizer Arguments")
        g.add_argument("--optimizer", default="adam", type=str, choices=["gd", "adam", "mom", "rmsprop"])
        g.add_argument("--momentum", default=None, type=float)
        g.add_argument("--optimizer_decay", default=None, type=float)
        g.add_argument("--optimizer_epsilon", default=None, type=float)
        g = parser.add_argument_group(f"({name}) Saver(Restore) Arguments")
        g.add_argument("--trainable_scopes", default="", type=str)
        g = parser.add_argument_group(f"({name}) Training options(step, batch_size, path) Arguments")
        g.add_argument("--train_dir", required=True, type=str)
        g.add_argument("--step_save_summaries", default=10, type=int)
        g.add_argument("--step_save_verbose_summaries", default=2000, type=int)
        g.add_argument("--step_save_first_n_summaries", default=30, type=int)
        g.add_argument("--step_save_checkpoint", default=500, type=int)
        g.add_argument("--step_evaluation", default=500, type=positive_int)
        g.add_argument("--no-write_pbtxt", dest="write_pbtxt", action="store_false")
        g.add_argument("--write_pbtxt", dest="write_pbtxt", action="store_true")
        g.set_defaults(write_pbtxt=True)
        g.add_argument("--max_to_keep", default=5, type=positive_int)
        g.add_argument("--max_outputs", default=5, type=positive_int)
        g.add_argument("--max_epoch_from_restore", default=50000, type=float)
        g.add_argument("--step_min_summaries", default=0, type=int)
        g.add_argument("--max_step_from_restore", default=sys.maxsize, type=int)
        g.add_argument("--class_sampling_factor", default=20, type=int)
        g.add_argument("--maximum_num_labels_for_metric", default=10, type=int)
        g = parser.add_argument_group("Learning Rate Scheduling Arguments")
        g.add_argument("--learning_rate", default=1e-4, type=float)
        g.add_argument("--boundaries", default=[100000, 200000], type=int, nargs="*")
        g.add_argument("--boundaries_epoch", dest="boundaries_epoch", action="store_true")
        g.add_argument("--no-boundaries_epoch", dest="boundaries_epoch", action="store_false")
        g.add_argument("--lr_list", default=[1e-3, 1e-4, 1e-5], type=float, nargs="*")
        g.add_argument("--relative_schedule", dest="relative", action="store_true")
        g.add_argument("--absolute_schedule", dest="relative", action="store_false")
        g.set_defaults(relative=True, boundaries_epoch=True)
        g = parser.add_argument_group("(data parallel) Arguments")     # new capability: the reference is single-device
        g.add_argument("--sync_bn", dest="sync_bn", action="store_true", help="cross-replica BN statistics (data parallel)")
        g.set_defaults(sync_bn=False)


class SingleLabelAudioTrainer(TrainerBase):
    def __init__(self, model, session, args, dataset, dataset_name, name="AudioTrainer"):
        super().__init__(model, session, args, dataset, dataset_name, name)
        self.setup_dataset_related_attr()
        self.setup_trainer()
        self.log.info("Watch validation with: --checkpoint_path %s", self.train_dir_name)

    def setup_dataset_related_attr(self):
        self.label_names = list(self.dataset.label_names)
        assert len(self.label_names) == self.args.num_classes
        self.use_class_metrics = len(self.label_names) < self.args.maximum_num_labels_for_metric

    def build_evaluate_iterations(self, iters):
        if iters is not None:
            return iters
        if self.args.evaluation_iterations is not None:
            return self.args.evaluation_iterations
        return max((self.args.class_sampling_factor * self.args.num_classes) // self.args.batch_size, 1)

    @staticmethod
    def add_arguments(parser):
        pass
