"""Data-parallel scene sharding: one process per GPU, one scene per rank, ONE all-reduce (RCCL over xGMI on the
MI355X node; gloo in the CPU tests) of the flat fp32 gradient bucket per optimizer step.

The reference has no distributed code; this is the north star's multi-GPU path.  Parameters and gradients are views
into two flat buffers, so the collective and the SGD update each touch one contiguous tensor (12.5 M params = 50 MB at
32->256 ch: a single bucket; xGMI is point-to-point so one large message per peer beats many small ones).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class FlatParams:
    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.empty(total, dtype=ref.dtype, device=ref.device)
        self.flat_grad = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p)
            p.grad = self.flat_grad[off:off + n].view_as(p)
            off += n

    def zero_grad(self):
        self.flat_grad.zero_()
        off = 0
        for p in self.params:                       # autograd may have replaced .grad; re-point at the bucket
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_grad[off:off + n].data_ptr():
                p.grad = self.flat_grad[off:off + n].view_as(p)
            off += n

    def all_reduce_mean(self, weight: float = 1.0, total_weight: float | None = None):
        """Sum gradients over ranks.  `weight` lets ranks with different active-voxel counts contribute in proportion
        (the reference normalises its losses by batch-level counts, loss.py:401-431)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if weight != 1.0:
                self.flat_grad.mul_(weight)
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)
            denom = total_weight if total_weight is not None else float(dist.get_world_size())
            self.flat_grad.div_(denom)

    def sgd_step(self, lr: float):
        self.flat.add_(self.flat_grad, alpha=-lr)


def broadcast_params(fp: FlatParams, src: int = 0):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(fp.flat, src)
