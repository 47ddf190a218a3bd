"""Data-parallel scene sharding: one process per GPU, one scene per rank, all-reduce (RCCL over xGMI on the MI355X
node; gloo in the CPU tests) of the flat fp32 gradient buffer once per optimizer step.

The reference has no distributed code; this is the north star's multi-GPU path.  Parameters and gradients are views
into two flat buffers, so the SGD update touches one contiguous tensor (12.5 M params = 50 MB at 32->256 ch) and the
collective a few large slices of it: xGMI is point-to-point (ring collectives are per-link bound), so messages are
kept large -- `n_buckets` contiguous slices of ~12 MB, cut in REVERSE parameter order.  A slice is reduced as soon as
backward has produced all of its gradients (post-accumulate hooks), i.e. while the rest of backward still runs; the
bottom-level weights (60 % of the bytes) are complete halfway through backward.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class FlatParams:
    """Parameters are views into one flat buffer.  Gradients are produced by autograd as separate tensors (`.grad` is
    reset to None each step, so AccumulateGrad adopts the kernel's output instead of launching one add per parameter);
    `gather_grads` packs them into the flat bucket with ONE multi-tensor copy."""

    TAIL_FRACTION = 0.05          # share of the gradient bytes left to the last (exposed) bucket

    def __init__(self, module: torch.nn.Module, n_buckets: int = 0):
        """n_buckets > 0: overlap the gradient all-reduce with backward (only used when a process group with more than
        one rank exists); 0: one all-reduce after backward."""
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
        self._setup_buckets(n_buckets)

    # ---- overlapped, bucketed all-reduce -----------------------------------------------------------------------
    def _setup_buckets(self, n_buckets):
        self.buckets = []                 # (param indices, flat slice), in the order backward completes them
        self._bucket_of, self._pending, self._works, self._launched = {}, [], [], []
        # (SCN_DP_FORCE_BUCKETS: take the overlapped path with a single rank too -- a 1-GPU rehearsal of the RCCL calls)
        min_world = 1 if os.environ.get("SCN_DP_FORCE_BUCKETS") else 2
        if n_buckets <= 0 or not (dist.is_available() and dist.is_initialized() and dist.get_world_size() >= min_world):
            return
        total = self.flat.numel()
        # The LAST bucket is complete only when backward ends, so its all-reduce is the one nothing hides: it gets the
        # parameters backward reaches last (the first encoder levels: a few per cent of the bytes) and the other buckets
        # share the rest equally.  (Equal cuts left 12 MB of the 50 MB at 32->256 channels exposed after backward.)
        tail = int(total * self.TAIL_FRACTION) if n_buckets > 1 else 0
        target = -(-(total - tail) // max(1, n_buckets - 1)) if n_buckets > 1 else total
        acc, idx = 0, []
        off = total
        for i in range(len(self.params) - 1, -1, -1):          # reverse parameter order ~ order of backward
            n = self.params[i].numel()
            idx.append(i); acc += n; off -= n
            last_bucket = len(self.buckets) == n_buckets - 1   # everything that is left goes into the last one
            if ((acc >= target or 0 < off <= tail) and not last_bucket) or i == 0:
                self.buckets.append((idx, self.flat_grad[off:off + acc]))
                idx, acc = [], 0
        for b, (ids, _) in enumerate(self.buckets):
            for i in ids:
                self._bucket_of[i] = b
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))
        self._reset_buckets()

    def _reset_buckets(self):
        self._pending = [len(ids) for ids, _ in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []

    def _make_hook(self, i):
        def hook(_param):
            b = self._bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch_bucket(b)
        return hook

    def _launch_bucket(self, b):
        ids, flat_slice = self.buckets[b]
        have = [(self.grad_views[i], self.params[i].grad) for i in ids if self.params[i].grad is not None]
        if len(have) != len(ids):
            flat_slice.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        self._launched[b] = True
        self._works.append(dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, async_op=True))

    def zero_grad(self):
        for p in self.params:
            p.grad = None
        if self.buckets:
            self._reset_buckets()

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
        if self.buckets:
            if weight != 1.0:
                raise ValueError("weighted ranks need the single-bucket path (n_buckets=0)")
            for b in range(len(self.buckets)):              # slices whose parameters received no (or not all) gradients
                if not self._launched[b]:
                    self._launch_bucket(b)
            for w in self._works:
                w.wait()
            denom = total_weight if total_weight is not None else float(dist.get_world_size())
            self.flat_grad.div_(denom)
            return
        self.gather_grads()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if weight != 1.0:
                self.flat_grad.mul_(weight)
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)
            denom = total_weight if total_weight is not None else float(dist.get_world_size())
            self.flat_grad.div_(denom)

    def sgd_step(self, lr: float):
        self.flat.add_(self.flat_grad, alpha=-lr)

    def step_single_rank(self, lr: float):
        """all_reduce_mean + sgd_step for ONE rank without a process group: there is nothing to reduce, so the gradients
        need not be packed into the flat bucket first (a 50 MB copy at 32->256 channels) -- the update reads them where
        autograd left them, one multi-tensor launch.  Same arithmetic per element as sgd_step."""
        if (self.buckets or os.environ.get("SCN_STEP_PACKED") or
                (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)):
            self.all_reduce_mean()
            self.sgd_step(lr)
            return
        have = [(p, p.grad) for p in self.params if p.grad is not None]
        if have:
            torch._foreach_add_([p.data for p, _ in have], [g for _, g in have], alpha=-lr)


def broadcast_params(fp: FlatParams, src: int = 0):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(fp.flat, src)
