"""Checkpoint bookkeeping with the reference's names (common/tf_utils.py:64-247): BestKeeper + scores.tsv, the
checkpoint-directory watcher behind `--valid_type loop`, checkpoint-path resolution.  Host-side file handling only."""
from __future__ import annotations

import logging
import math
import shutil
import time
from pathlib import Path
from typing import Dict, Iterator, Optional

from . import tf_bundle


def show_models(logger, engine) -> int:
    """Lists every variable of the model and returns the parameter total the reference logs (:40-61: trainable and
    non-trainable variables of the graph, i.e. including the BN moving statistics)."""
    total = trainable = 0
    for name, ti in engine.tensors.items():
        shape = [ti.shape[i] for i in range(ti.rank)]
        total += int(ti.size)
        trainable += int(ti.size) if ti.arena == 0 else 0
        logger.debug(">>    %s float32 : %s, %d ... %d (is_trainable: %s)", name, shape, int(ti.size), total, ti.arena == 0)
    logger.info(">> Number of variables: %d, total prod + sum of shape: %d (%d trainable)", len(engine.tensors), total, trainable)
    return total


def wait_for_new_checkpoint(checkpoint_dir, last_checkpoint: Optional[str], seconds_to_sleep: float = 1.0,
                            timeout: Optional[float] = None) -> Optional[str]:
    """tf.contrib.training.wait_for_new_checkpoint: blocks until latest_checkpoint(dir) names a checkpoint other than
    `last_checkpoint`; None when `timeout` seconds pass first."""
    stop = None if timeout is None else time.time() + timeout
    while True:
        path = tf_bundle.latest_checkpoint(str(checkpoint_dir))
        if path is not None and path != last_checkpoint:
            return path
        if stop is not None and time.time() + seconds_to_sleep > stop:
            return None
        time.sleep(seconds_to_sleep)


def ckpt_iterator(checkpoint_dir, min_interval_secs: float = 0, timeout: Optional[float] = None, timeout_fn=None,
                  logger=None, seconds_to_sleep: float = 1.0) -> Iterator[str]:
    """Yields every new checkpoint that appears in `checkpoint_dir` (:64-66 -> tf.contrib.training.checkpoints_iterator).
    With `timeout` None it waits forever, like the reference's evaluate loop; otherwise iteration ends when no new
    checkpoint shows up within `timeout` seconds (and `timeout_fn`, if given, returns True)."""
    last = None
    while True:
        new = wait_for_new_checkpoint(checkpoint_dir, last, seconds_to_sleep=seconds_to_sleep, timeout=timeout)
        if new is None:
            if timeout_fn is None or timeout_fn():
                return
            continue
        start = time.time()
        last = new
        yield new
        wait = start + min_interval_secs - time.time()
        if wait > 0:
            time.sleep(wait)


class BestKeeper:
    """Keeps, per metric, a copy of the checkpoint that scored best on it under <directory>/<dataset>/<metric>/ together
    with a one-row scores.tsv (:70-216).  `mode` "max": larger is better, "min": smaller; a new optimum has to beat the
    old one by more than `epsilon`."""

    def __init__(self, metric_with_modes: Dict[str, str], dataset_name: str, directory, logger=None, epsilon: float = 0.00005,
                 score_file: str = "scores.tsv", metric_best: Optional[Dict[str, float]] = None):
        self.log = logger or logging.getLogger("BestKeeper")
        self.score_file = score_file
        self.metric_best: Dict[str, float] = {} if metric_best is None else metric_best
        self.modes = dict(metric_with_modes)
        self.epsilon = float(epsilon)
        self.directory = Path(directory)
        self.output_temp_dir = self.directory / f"{dataset_name}_best_keeper_temp"
        self.log.info("Initialize BestKeeper: Monitor %s & Save to %s", dataset_name, directory)
        for metric_name, mode in self.modes.items():
            if mode not in ("min", "max"):
                raise ValueError(f"Unsupported mode : {mode}")
            default = math.inf if mode == "min" else -math.inf
            self.metric_best[metric_name] = self.load_metric_from_scores_tsv(
                self.directory / dataset_name / metric_name / score_file, metric_name, default)

    def x_better_than_y(self, metric_name: str, x: float, y: float) -> bool:
        return x < y - self.epsilon if self.modes[metric_name] == "min" else x > y + self.epsilon

    @staticmethod
    def load_metric_from_scores_tsv(full_path: Path, metric_name: str, default_value: float) -> float:
        full_path = Path(full_path)
        if not full_path.exists():
            return default_value
        with open(full_path) as fh:
            header = fh.readline().rstrip("\n").split("\t")
            values = fh.readline().rstrip("\n").split("\t")
        cell = values[header.index(metric_name)]
        return float(cell) if cell else default_value

    def monitor(self, dataset_name: str, eval_scores: Dict[str, float]):
        metrics_keep, is_keep = {}, False
        for metric_name in self.metric_best:
            score = eval_scores[metric_name]
            if self.x_better_than_y(metric_name, score, self.metric_best[metric_name]):
                self.log.info("[KeepBest] %s %.6f -> %.6f, so keep it!", metric_name, self.metric_best[metric_name], score)
                self.metric_best[metric_name] = score
                metrics_keep[metric_name] = is_keep = True
            else:
                metrics_keep[metric_name] = False
        return is_keep, metrics_keep

    def _keep_path(self, dataset_name: str, metric_name: str) -> Path:
        return self.directory / Path(dataset_name) / Path(metric_name)

    def remove_old_best(self, dataset_name: str, metrics_keep: Dict[str, bool]):
        for metric_name, keep in metrics_keep.items():
            if keep:
                p = self._keep_path(dataset_name, metric_name)
                if p.exists():
                    shutil.rmtree(str(p))
                p.mkdir(parents=True)

    def save_best(self, dataset_name: str, metrics_keep: Dict[str, bool], ckpt_glob: Path):
        for metric_name, keep in metrics_keep.items():
            if keep:
                p = self._keep_path(dataset_name, metric_name)
                self.keep_checkpoint(p, ckpt_glob)
                self.keep_converted_files(p)

    def save_scores(self, dataset_name: str, metrics_keep: Dict[str, bool], eval_scores: Dict[str, float], meta_info=None):
        row = dict(eval_scores)
        row.update(meta_info or {})
        cols = sorted(row)                                          # pandas .sort_index() of the one-row frame
        def cell(v):
            return f"{v:.5f}" if isinstance(v, float) else str(v)   # float_format="%.5f"; integers print as they are
        for metric_name, keep in metrics_keep.items():
            if keep:
                p = self._keep_path(dataset_name, metric_name)
                p.mkdir(parents=True, exist_ok=True)
                with open(p / self.score_file, "w") as fh:
                    fh.write("\t".join(cols) + "\n" + "\t".join(cell(row[c]) for c in cols) + "\n")

    def keep_checkpoint(self, keep_dir, ckpt_glob: Path):
        keep_dir, ckpt_glob = Path(keep_dir), Path(ckpt_glob)
        stem = None
        for path in sorted(ckpt_glob.parent.glob(ckpt_glob.name)):  # .index, .data-00000-of-00001 (and .meta when present)
            shutil.copy(str(path), str(keep_dir))
            stem = Path(path.name).stem
        if stem is not None:
            with open(keep_dir / "checkpoint", "w") as fh:
                fh.write(f'model_checkpoint_path: "{stem}"')

    def keep_converted_files(self, keep_path):
        keep_path = Path(keep_path)
        if not self.output_temp_dir.exists():
            return
        for path in self.output_temp_dir.glob("*"):
            if path.is_dir():
                shutil.copytree(str(path), str(keep_path / path.name))
            else:
                shutil.copy(str(path), str(keep_path / path.name))

    def remove_temp_dir(self):
        if self.output_temp_dir.exists():
            shutil.rmtree(str(self.output_temp_dir))


def resolve_checkpoint_path(checkpoint_path, log, is_training: bool, poll_secs: float = 1.0, timeout: Optional[float] = None):
    """A directory becomes its latest checkpoint; an evaluator waits until one exists (:219-235)."""
    if checkpoint_path and Path(checkpoint_path).is_dir():
        old = checkpoint_path
        checkpoint_path = tf_bundle.latest_checkpoint(old)
        if not is_training and checkpoint_path is None:
            log.info("There are no checkpoint file yet")
            checkpoint_path = wait_for_new_checkpoint(old, None, seconds_to_sleep=poll_secs, timeout=timeout)
        log.info("self.args.checkpoint_path updated: %s -> %s", old, checkpoint_path)
    else:
        log.info("checkpoint_path is %s", checkpoint_path)
    return checkpoint_path


def get_global_step_from_checkpoint(checkpoint_path) -> int:
    """`.../Model-1234` -> 1234; no path or no numeric suffix -> 0 (:238-247)."""
    if not checkpoint_path:
        return 0
    stem = Path(checkpoint_path).stem
    if "-" in stem and stem.split("-")[-1].isdigit():
        return int(stem.split("-")[-1])
    return 0
