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
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0
        if self.enabled and ranks_share_gpu() and hasattr(net, "lib"):
            net.lib.tcr_tune(7, 1)          # TCR_TUNE_WGRAD_STREAM: everything on the caller's stream (see ranks_share_gpu)

    def _sum(self, t: torch.Tensor):
        if self.enabled:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def forward_train(self, feat, labels, keep_prob=1.0, seed=0, label_smoothing=0.0):
        """Local shard forward.  dlogits are scaled by 1/global_batch so that the SUM of the replicas'
        gradients is the gradient of the global mean loss; dropout masks are indexed by global sample id."""
        b = feat.shape[0]
        hook = self._sum if (self.sync_bn and self.enabled) else None
        return self.net.forward_train(feat, labels, keep_prob=keep_prob, seed=seed, sample_offset=self.rank * b,
                                      global_batch=self.world * b, label_smoothing=label_smoothing, sync_hook=hook)

    def backward(self):
        g = self.net.backward()
        self._sum(g)                    # one RCCL all-reduce of the whole arena
        return g

    def mean_loss(self, loss_sum: torch.Tensor, local_batch: int) -> torch.Tensor:
        t = loss_sum.detach().clone().reshape(1)
        self._sum(t)
        return t[0] / float(self.world * local_batch)
