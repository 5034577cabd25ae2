"""Evaluation with the reference evaluator's behaviour (helper/base.py:52-125, helper/evaluator.py:102-133,
metrics/parser.py:136-138): iterations = num_samples // batch_size (remainder dropped), fetch softmax + labels +
loss per batch, class id = argmax, accuracy + mean loss + per-batch / per-unit inference time."""
from __future__ import annotations

import logging
import time

import numpy as np
import torch


class SingleLabelAudioEvaluator:
    def __init__(self, model, session, args, dataset, dataset_name):
        self.model, self.args, self.dataset, self.dataset_name = model, args, dataset, dataset_name
        self.log = logging.getLogger("Evaluator")

    @staticmethod
    def add_arguments(parser):
        g = parser.add_argument_group("(Evaluator) Arguments")
        g.add_argument("--valid_type", default="loop", type=str, choices=["loop", "once"])
        g.add_argument("--evaluation_iterations", default=None, type=int)
        g.add_argument("--max_step_from_restore", default=1e20, type=float)
        g.add_argument("--save_best_keeper", dest="save_best_keeper", action="store_true")
        g.add_argument("--no-save_best_keeper", dest="save_best_keeper", action="store_false")
        g.set_defaults(save_best_keeper=False)

    def evaluate_once(self, checkpoint_path=None):
        if checkpoint_path:
            import os
            if os.path.isdir(checkpoint_path):          # directory -> latest checkpoint (common/tf_utils.py:219-247)
                cands = sorted((f for f in os.listdir(checkpoint_path) if f.endswith(".npz")), key=lambda f: int(f.rsplit("-", 1)[1][:-4]))
                checkpoint_path = os.path.join(checkpoint_path, cands[-1])
            ck = dict(np.load(checkpoint_path))
            self.model.engine.load_state_dict({k: v for k, v in ck.items() if not k.startswith("__slot__/") and k != "global_step"})
        iters = self.args.evaluation_iterations or max(self.dataset.num_samples // self.dataset.batch_size, 1)
        correct = total = 0
        losses, times = [], []
        all_probs, all_labels = [], []
        for _ in range(int(iters)):
            wavs, labels = self.dataset.next_batch()
            t0 = time.perf_counter()
            self.model.build(wavs, labels, is_training=False)
            pred = self.model.outputs.argmax(dim=1)            # predictions_onehot argmax
            truth = labels.argmax(dim=1)
            correct += int((pred == truth).sum())
            total += int(labels.shape[0])
            losses.append(float(self.model.total_loss))
            all_probs.append(self.model.outputs.detach().cpu().numpy())         # np.vstack of the fetched batches (helper/base.py:86-104)
            all_labels.append(labels.detach().cpu().numpy())
            times.append(time.perf_counter() - t0)
        out = {"accuracy": correct / max(total, 1), "total_loss": float(np.mean(losses)), "batch_infer_time": float(np.mean(times)),
               "unit_infer_time": float(np.mean(times)) / self.dataset.batch_size, "num_evaluated": total}
        from ..metrics import audio_metrics
        names = getattr(self.dataset, "label_names", None) or [str(i) for i in range(all_probs[0].shape[1])]
        out.update(audio_metrics(np.vstack(all_probs), np.vstack(all_labels), names, self.dataset_name))
        self.log.info("%s", {k: v for k, v in out.items() if not k.startswith(("precision/", "recall/", "f1score/", "ap/", "classification_report/"))})
        return out
