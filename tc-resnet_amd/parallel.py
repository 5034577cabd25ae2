"""Single-node data parallelism: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

The reference is single-device (const.py:4-7, no collective anywhere), so this is new capability.  The path
shards by utterance:
  * eval forward: independent replicas, no collective;
  * training: ONE all-reduce per step of the flat gradient arena (65 264 floats for TCResNet8-1.0, 303 144 for
    TCResNet14-1.5: a latency-bound message, so never per-tensor calls), optionally preceded by the tiny
    per-BN-layer statistic all-reduces of cross-replica ("sync") BN, which reproduces the reference's
    global-batch statistics exactly.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def ranks_share_gpu() -> bool:
    """True when more local ranks than GPUs are running (several processes time-sharing one device: a CPU-side rehearsal
    of the multi-GPU path, never a production layout).  Cross-stream event waits of one process can then sit out whole
    scheduler time slices of the other (measured: seconds per step), so the stream-level overlaps are switched off."""
    import os
    return torch.cuda.is_available() and int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > torch.cuda.device_count()


class DataParallel:
    def __init__(self, net, sync_bn: bool = False, group: Optional[dist.ProcessGroup] = None):
        self.net = net
        self.group = group
        self.sync_bn = bool(sync_bn)
        # (TCR_DP_FORCE=1: collectives also in a group of ONE rank -- the only way a 1-GPU box can send this code through RCCL: tests)
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("TCR_DP_FORCE") == "1")
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0
        if self.enabled and ranks_share_gpu() and hasattr(net, "lib"):
            net.lib.tcr_tune(7, 1)          # TCR_TUNE_WGRAD_STREAM: everything on the caller's stream (see ranks_share_gpu)
        self.collectives = 0                # all-reduces issued so far (tests / bench: collectives per step)
        self._loss = None                   # this step's local loss sum (forward_train) ...
        self._loss_reduced = None           # ... and its all-reduced value, carried in the tail of the gradient arena

    def _sum(self, t: torch.Tensor):
        if self.enabled:
            self.collectives += 1
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def forward_train(self, feat, labels, keep_prob=1.0, seed=0, label_smoothing=0.0):
        """Local shard forward.  dlogits are scaled by 1/global_batch so that the SUM of the replicas'
        gradients is the gradient of the global mean loss; dropout masks are indexed by global sample id."""
        b = feat.shape[0]
        hook = self._sum if (self.sync_bn and self.enabled) else None
        out = self.net.forward_train(feat, labels, keep_prob=keep_prob, seed=seed, sample_offset=self.rank * b,
                                     global_batch=self.world * b, label_smoothing=label_smoothing, sync_hook=hook)
        self._loss, self._loss_reduced = out[2], None
        return out

    def backward(self):
        """Local backward, then ONE all-reduce of the gradient arena; the replica's loss sum rides in the arena's tail
        (engine.DP_TAIL), so that logging the global mean loss costs no second collective."""
        g = self.net.backward()
        buf = getattr(self.net, "_grads_buf", None)           # the arena followed by engine.DP_TAIL floats
        n = g.numel()
        has_tail = self.enabled and self._loss is not None and buf is not None and buf.numel() > n and buf.data_ptr() == g.data_ptr()
        if has_tail:
            buf[n:n + 1].copy_(self._loss.detach().reshape(1))
            self._sum(buf[:n + 1])
            self._loss_reduced = (self._loss, buf[n].clone())
            buf[n:n + 1].zero_()
        else:
            self._sum(g)
        return g

    def mean_loss(self, loss_sum: torch.Tensor, local_batch: int) -> torch.Tensor:
        if self._loss_reduced is not None and self._loss_reduced[0] is loss_sum:
            return self._loss_reduced[1] / float(self.world * local_batch)
        t = loss_sum.detach().clone().reshape(1)
        self._sum(t)
        return t[0] / float(self.world * local_batch)
