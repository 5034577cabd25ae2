"""Evaluation with the reference evaluator's behaviour (helper/evaluator.py:20-212): resolve a checkpoint directory to its
latest checkpoint, restore through the scope-filtering loader (optionally the EMA shadows), run the forward-only loop over
num_samples // batch_size batches (remainder dropped), compute the metric set, keep the best checkpoint per metric
(`BestKeeper`: <watch_path>/<split>/<metric>/ + scores.tsv) and stop once a checkpoint at or beyond
--max_step_from_restore has been evaluated."""
from __future__ import annotations

import logging
from pathlib import Path

from ..common import tf_bundle, tf_utils
from ..common.model_loader import Ckpt
from .base import AudioBase


class Evaluator(AudioBase):
    def __init__(self, model, session, args, dataset, dataset_name, name):
        self.log = logging.getLogger(name)
        self.model, self.session, self.args, self.dataset, self.dataset_name = model, session, args, dataset, dataset_name
        if Path(self.args.checkpoint_path).is_dir():
            latest = tf_bundle.latest_checkpoint(self.args.checkpoint_path)
            if latest is not None:
                self.args.checkpoint_path = latest
            self.log.info("Get latest checkpoint and update to it: %s", self.args.checkpoint_path)
        self.watch_path = self._build_watch_path()
        self.ckpt_loader = Ckpt(model.engine, include_scopes=args.checkpoint_include_scopes, exclude_scopes=args.checkpoint_exclude_scopes,
                                ignore_missing_vars=args.ignore_missing_vars, use_ema=args.use_ema, ema_decay=args.ema_decay, logger=self.log)
        self.finished = False       # set once a checkpoint with step >= --max_step_from_restore was evaluated (the reference exits there)

    def _build_watch_path(self) -> Path:
        p = Path(self.args.checkpoint_path)
        return p if p.is_dir() else p.parent

    @staticmethod
    def build_evaluation_step(checkpoint_path: str) -> int:
        tail = str(checkpoint_path).split("-")[-1]
        return int(tail) if "-" in str(checkpoint_path) and tail.isdigit() else 0

    @staticmethod
    def build_checkpoint_paths(checkpoint_path: str):
        return Path(str(checkpoint_path) + "*"), Path(checkpoint_path)

    def build_miscellaneous_path(self, name: str) -> Path:
        target = self.watch_path / "miscellaneous" / self.dataset_name / name
        target.mkdir(parents=True, exist_ok=True)
        return target

    def setup_best_keeper(self):
        self.best_keeper = tf_utils.BestKeeper(self.best_keep_metrics(), self.dataset_name, self.watch_path, self.log)

    def evaluate_once(self, checkpoint_path):
        self.log.info("Evaluation started")
        self.setup_dataset_iterator()
        self.ckpt_loader.load(checkpoint_path)
        step = self.build_evaluation_step(checkpoint_path)
        checkpoint_glob, checkpoint_path = self.build_checkpoint_paths(str(checkpoint_path))
        metrics = self.run_evaluation(step, iters=self.args.evaluation_iterations, is_training=False)
        best_keep = {k: metrics[k] for k in self.best_keeper.metric_best if k in metrics}
        is_keep, metrics_keep = self.best_keeper.monitor(self.dataset_name, best_keep)
        if self.args.save_best_keeper:
            meta_info = {"step": step, "model_size": self.model.total_params}
            self.best_keeper.remove_old_best(self.dataset_name, metrics_keep)
            self.best_keeper.save_best(self.dataset_name, metrics_keep, checkpoint_glob)
            self.best_keeper.remove_temp_dir()
            self.best_keeper.save_scores(self.dataset_name, metrics_keep, best_keep, meta_info)
        self.log_metrics(step, metrics)
        self.log.info("Evaluation finished")
        metrics["step"] = step
        metrics["accuracy"] = metrics[f"accuracy/{self.dataset_name}"]
        metrics["total_loss"] = metrics[f"metric_loss/{self.dataset_name}/total_loss"]
        if step >= self.args.max_step_from_restore:
            self.log.info("Evaluation stopped")
            self.finished = True
        return metrics

    def build_train_directory(self) -> str:
        p = Path(self.args.checkpoint_path)
        return str(p) if p.is_dir() else str(p.parent)

    @staticmethod
    def add_arguments(parser):
        g = parser.add_argument_group("(Evaluator) arguments")
        g.add_argument("--valid_type", default="loop", type=str, choices=["loop", "once"])
        g.add_argument("--max_outputs", default=5, type=int)
        g.add_argument("--maximum_num_labels_for_metric", default=10, type=int)
        g.add_argument("--no-save_best_keeper", dest="save_best_keeper", action="store_false")
        g.add_argument("--save_best_keeper", dest="save_best_keeper", action="store_true")
        g.set_defaults(save_best_keeper=True)
        g.add_argument("--no-flatten_output", dest="flatten_output", action="store_false")
        g.add_argument("--flatten_output", dest="flatten_output", action="store_true")
        g.set_defaults(flatten_output=False)
        g.add_argument("--max_step_from_restore", default=1e20, type=float)
        g.add_argument("--loop_timeout_secs", default=None, type=float,
                       help="(not in the reference) --valid_type loop gives up after this long without a new checkpoint; default: wait forever")


class SingleLabelAudioEvaluator(Evaluator):
    def __init__(self, model, session, args, dataset, dataset_name):
        super().__init__(model, session, args, dataset, dataset_name, "SingleLabelAudioEvaluator")
        self.setup_dataset_related_attr()
        self.setup_best_keeper()

    def setup_dataset_related_attr(self):
        assert len(self.dataset.label_names) == self.args.num_classes
        self.use_class_metrics = len(self.dataset.label_names) < self.args.maximum_num_labels_for_metric

    def setup_dataset_iterator(self):
        self.dataset.setup_iterator()           # every evaluation starts from the first sample of the split
