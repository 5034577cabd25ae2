"""Training loop with the reference trainer's knobs (helper/trainer.py:107-137,171-222,312-434,462-536):
piecewise-constant LR over global_step, gd/adam/mom optimisers, one fused step per iteration
(what session.run(train_op) was), step timing, periodic checkpoints under train_dir as
`<ModelName>-<global_step>.npz` keyed by TF variable name."""
from __future__ import annotations

import logging
import os
import time

import numpy as np
import torch
import torch.distributed as dist


def piecewise_constant(step: int, boundaries, values) -> float:
    """tf.train.piecewise_constant: x <= b0 -> v0; b0 < x <= b1 -> v1; ... (helper/trainer.py:135-137)."""
    for b, v in zip(boundaries, values):
        if step <= b:
            return v
    return values[-1]


class SingleLabelAudioTrainer:
    def __init__(self, model, session, args, dataset, dataset_name):
        self.model, self.args, self.dataset, self.dataset_name = model, args, dataset, dataset_name
        self.log = logging.getLogger("Trainer")
        self.global_step = 0
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if args.boundaries_epoch:
            self.boundaries = [b * dataset.num_samples // dataset.batch_size for b in args.boundaries]
        else:
            self.boundaries = list(args.boundaries)
        if getattr(args, "checkpoint_path", ""):
            self.global_step = self.restore(args.checkpoint_path)
            if getattr(args, "relative", False):
                self.boundaries = [self.global_step + b for b in self.boundaries]

    @staticmethod
    def add_arguments(parser):
        g = parser.add_argument_group("(Trainer) Arguments")
        g.add_argument("--optimizer", default="adam", type=str, choices=["gd", "adam", "mom", "rmsprop"])
        g.add_argument("--momentum", default=None, type=float)
        g.add_argument("--optimizer_decay", default=None, type=float)
        g.add_argument("--optimizer_epsilon", default=None, type=float)
        g.add_argument("--train_dir", required=True, type=str)
        g.add_argument("--step_save_checkpoint", default=500, type=int)
        g.add_argument("--step_evaluation", default=500, type=int)
        g.add_argument("--step_save_summaries", default=10, type=int)
        g.add_argument("--max_to_keep", default=5, type=int)
        g.add_argument("--max_step_from_restore", default=1e20, type=float)
        g.add_argument("--max_epoch_from_restore", default=50000, type=int)
        g.add_argument("--learning_rate", default=1e-4, type=float)
        g.add_argument("--boundaries", default=[1000000], type=int, nargs="*")
        g.add_argument("--boundaries_epoch", dest="boundaries_epoch", action="store_true")
        g.add_argument("--no-boundaries_epoch", dest="boundaries_epoch", action="store_false")
        g.set_defaults(boundaries_epoch=False)
        g.add_argument("--lr_list", default=[1e-4], type=float, nargs="*")
        g.add_argument("--relative_schedule", dest="relative", action="store_true")
        g.add_argument("--absolute_schedule", dest="relative", action="store_false")
        g.set_defaults(relative=False)
        g.add_argument("--sync_bn", dest="sync_bn", action="store_true", help="cross-replica BN statistics (data parallel)")
        g.set_defaults(sync_bn=False)

    # ---- checkpoints (TF variable names as keys) -----------------------------------
    def save(self) -> str:
        os.makedirs(self.args.train_dir, exist_ok=True)
        path = os.path.join(self.args.train_dir, f"{self.args.model}-{self.global_step}.npz")
        sd = self.model.engine.state_dict()
        for slot, arena in self.model.engine.slots.items():
            sd[f"__slot__/{slot}"] = arena.detach().cpu().numpy()
        sd["global_step"] = np.array(self.global_step, np.int64)
        np.savez(path, **sd)
        kept = sorted((f for f in os.listdir(self.args.train_dir) if f.startswith(f"{self.args.model}-") and f.endswith(".npz")),
                      key=lambda f: int(f[len(self.args.model) + 1:-4]))
        for old in kept[:-int(self.args.max_to_keep)]:
            os.remove(os.path.join(self.args.train_dir, old))
        return path

    def restore(self, path: str) -> int:
        if os.path.isdir(path):
            cands = sorted((f for f in os.listdir(path) if f.endswith(".npz")), key=lambda f: int(f.rsplit("-", 1)[1][:-4]))
            path = os.path.join(path, cands[-1])
        ck = dict(np.load(path))
        self.model.engine.load_state_dict({k: v for k, v in ck.items() if not k.startswith("__slot__/") and k != "global_step"})
        for k, v in ck.items():
            if k.startswith("__slot__/"):
                self.model.engine._slot(k[len("__slot__/"):]).copy_(torch.as_tensor(v).to(self.model.engine.device))
        return int(ck["global_step"])

    # ---- loop ---------------------------------------------------------------------------------------
    def run_single_step(self):
        lr = piecewise_constant(self.global_step, self.boundaries, self.args.lr_list)
        wavs, labels = self.dataset.next_batch(self.rank, self.world)
        t0 = time.perf_counter()
        total, model = self.model.train_step(wavs, labels, lr, optimizer=self.args.optimizer,
                                             momentum=self.args.momentum if self.args.momentum is not None else 0.9,
                                             sync_bn=bool(getattr(self.args, "sync_bn", False)))
        self.global_step += 1
        return total, model, lr, t0

    def train(self):
        max_step = int(min(self.args.max_step_from_restore, 1e18))
        while self.global_step < max_step:
            total, model, lr, t0 = self.run_single_step()
            if self.global_step % int(self.args.step_save_summaries) == 0 or self.global_step == max_step:
                tl, ml = float(total), float(model)          # one host sync per logging interval
                dt = (time.perf_counter() - t0) * 1e3
                if self.rank == 0:
                    self.log.info("step %d lr %.4g total_loss %.5f model_loss %.5f SingleStep(ms) %.3f SingleStepPerInstance(ms) %.5f",
                                  self.global_step, lr, tl, ml, dt, dt / (self.dataset.batch_size * self.world))
                if not np.isfinite(tl):
                    raise FloatingPointError("total_loss is not finite")       # tf.check_numerics in create_train_op
            if self.rank == 0 and self.global_step % int(self.args.step_save_checkpoint) == 0:
                self.save()
        if self.rank == 0:
            self.save()
