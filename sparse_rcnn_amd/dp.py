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
    """Parameters are views into one flat buffer.  Gradients are produced by autograd as separate tensors (`.grad` is
    reset to None each step, so AccumulateGrad adopts the kernel's output instead of launching one add per parameter);
    `gather_grads` packs them into the flat bucket with ONE multi-tensor copy."""

    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.empty(total, dtype=ref.dtype, device=ref.device)
        self.flat_grad = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.grad_views = []
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p)
            self.grad_views.append(self.flat_grad[off:off + n].view_as(p))
            off += n

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def gather_grads(self):
        """Pack the per-parameter gradients into the flat bucket (zeros for parameters that received none)."""
        have = [(v, p.grad) for v, p in zip(self.grad_views, self.params) if p.grad is not None]
        if len(have) != len(self.params):
            self.flat_grad.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])

    def all_reduce_mean(self, weight: float = 1.0, total_weight: float | None = None):
        """Sum gradients over ranks.  `weight` lets ranks with different active-voxel counts contribute in proportion
        (the reference normalises its losses by batch-level counts, loss.py:401-431)."""
        self.gather_grads()
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
