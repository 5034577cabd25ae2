"""What the reference's trainer and evaluator share (helper/base.py:15-182): the forward-only inference loop that
stacks labels / softmax outputs / loss over batches, metric evaluation on the stacked arrays, and the `--use_ema`,
`--ema_decay`, `--evaluation_iterations` flags."""
from __future__ import annotations

import logging
import time
from typing import Dict, Optional

import numpy as np

from ..metrics import audio_metrics, best_keep_metric_with_modes


class Base:
    def build_iters_from_batch_size(self, num_samples: int, batch_size: int) -> int:
        iters = num_samples // batch_size
        ignored = num_samples % batch_size
        if ignored > 0:
            self.log.warning("Number of samples cannot be divided by batch_size, so it ignores some data examples in evaluation: "
                             "%d %% %d = %d", num_samples, batch_size, ignored)
        return iters

    def run_inference(self, global_step: int, iters: Optional[int] = None, is_training: bool = False, do_eval: bool = True):
        """One forward pass per batch (`session.run(fetch_ops)` in the reference, :52-125); returns the per-batch values
        stacked with np.vstack: labels_onehot, predictions_onehot, total_loss, model_loss, batch / unit inference time (ms)."""
        rank, world = getattr(self, "rank", 0), getattr(self, "world", 1)
        if iters is None:
            # one iteration consumes batch_size utterances PER RANK: the default covers the split once, not `world` times
            iters = self.build_iters_from_batch_size(self.dataset.num_samples, self.args.batch_size * world)
            if world > 1 and self.dataset.num_samples % (self.args.batch_size * world):
                self.log.warning("evaluation over %d ranks drops the last %d of %d samples (batches of %d per rank)", world,
                                 self.dataset.num_samples % (self.args.batch_size * world), self.dataset.num_samples, self.args.batch_size)
        agg: Dict[str, list] = {k: [] for k in ("labels_onehot", "predictions_onehot", "total_loss", "model_loss",
                                                "batch_infer_time", "unit_infer_time")}
        for _ in range(int(iters)):
            wavs, labels = self.dataset.next_batch(rank, world)
            st = time.time()
            self.model.build(wavs, labels, is_training=is_training)
            probs = self.model.outputs.detach().cpu().numpy()              # (the copy is the synchronisation point)
            ms = (time.time() - st) * 1000.0
            agg["predictions_onehot"].append(probs)
            if do_eval:
                agg["labels_onehot"].append(labels.detach().cpu().numpy())
                agg["total_loss"].append(np.float32(float(self.model.total_loss)))
                agg["model_loss"].append(np.float32(float(self.model.model_loss)))
            agg["batch_infer_time"].append(ms)
            agg["unit_infer_time"].append(ms / self.args.batch_size)
        out = {k: np.vstack(v) for k, v in agg.items() if v}
        if world > 1:
            # data parallel: every rank evaluated its shard of each batch; the metrics are over all of them (rank order = sample order
            # within a batch does not matter to any metric)
            import torch
            import torch.distributed as dist
            dev = self.model.outputs.device if dist.get_backend() == "nccl" else torch.device("cpu")
            gathered = {}
            for k in sorted(out):           # fixed-shape tensors (every rank ran the same iterations), no pickling
                t = torch.from_numpy(np.ascontiguousarray(out[k], dtype=np.float32)).to(dev)
                parts = [torch.empty_like(t) for _ in range(world)]
                dist.all_gather(parts, t)
                gathered[k] = np.vstack([p.cpu().numpy() for p in parts])
            out = gathered
        return out

    def run_evaluation(self, global_step: int, iters: Optional[int] = None, is_training: bool = False) -> Dict[str, object]:
        eval_dict = self.run_inference(global_step, iters, is_training, do_eval=True)
        return self.evaluate_metrics(eval_dict)

    def evaluate_metrics(self, eval_dict) -> Dict[str, object]:
        split = self.dataset_name
        names = list(getattr(self.dataset, "label_names", None) or [str(i) for i in range(eval_dict["predictions_onehot"].shape[1])])
        out = audio_metrics(eval_dict["predictions_onehot"], eval_dict["labels_onehot"], names, split,
                            use_class_metrics=self.use_class_metrics)
        out[f"metric_loss/{split}/total_loss"] = float(np.mean(eval_dict["total_loss"]))        # LossesMetricOp.expectation_of
        out[f"metric_loss/{split}/model_loss"] = float(np.mean(eval_dict["model_loss"]))
        out["batch_infer_time"] = float(np.mean(eval_dict["batch_infer_time"]))
        out["unit_infer_time"] = float(np.mean(eval_dict["unit_infer_time"]))
        out["num_evaluated"] = int(eval_dict["labels_onehot"].shape[0])
        return out

    def log_metrics(self, step: int, metrics: Dict[str, object]):
        for k in sorted(metrics):
            v = metrics[k]
            if isinstance(v, float):
                self.log.info("[%s] step %d %s: %.5f", self.dataset_name, step, k, v)
            elif isinstance(v, str) and k.startswith("classification_report/"):
                self.log.info("%s", v)

    def best_keep_metrics(self) -> Dict[str, str]:
        names = list(getattr(self.dataset, "label_names", None) or [])
        return best_keep_metric_with_modes(self.dataset_name, names, self.use_class_metrics)

    @staticmethod
    def add_arguments(parser):
        g = parser.add_argument_group("Base")
        g.add_argument("--no-use_ema", dest="use_ema", action="store_false")
        g.add_argument("--use_ema", dest="use_ema", action="store_true", help="Exponential Moving Average. It may take more memory.")
        g.set_defaults(use_ema=False)
        g.add_argument("--ema_decay", default=0.999, type=float)
        g.add_argument("--evaluation_iterations", type=int, default=None)


class AudioBase(Base):
    """(the reference's AudioBase only names the fetch ops; the stacked arrays above carry the same keys)"""
    log = logging.getLogger("AudioBase")
