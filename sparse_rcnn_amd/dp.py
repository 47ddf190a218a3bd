"""Data-parallel scene sharding: one process per GPU, one scene per rank, all-reduce (RCCL over xGMI on the MI355X
node; gloo in the CPU tests) of the flat fp32 gradient buffer once per optimizer step.

The reference has no distributed code; this is the north star's multi-GPU path.  Parameters and gradients are views
into two flat buffers, so the SGD update touches one contiguous tensor (12.5 M params = 50 MB at 32->256 ch) and the
collective a few large slices of it: xGMI is point-to-point (ring collectives are per-link bound), so messages are
kept large -- `n_buckets` contiguous slices, cut in REVERSE parameter order.  A slice is reduced as soon as backward has
produced all of its gradients (post-accumulate hooks) AND every slice before it has been launched, i.e. while the rest
of backward still runs; the bottom-level weights (60 % of the bytes) are complete halfway through backward.

Collectives are issued in ONE order on every rank (bucket 0, 1, 2, ...), whatever order the local graph completes the
buckets in: a rank whose ROI crop is empty gives its mask-branch parameters no gradient at all, and ranks that launched
"whatever is ready" would issue all-reduces of different sizes in different orders (a hang, or mixed buffers).

`flat_grad` holds the SUM over ranks of `rank_weight x gradient`; the mean is `flat_grad x grad_scale`.  The scale is
folded into the SGD step (`alpha = -lr x grad_scale`): no pass over the 50 MB buffer after the last all-reduce.

Gradient accumulation (the reference's only batch-scaling mechanism: `(loss / batches_per_step).backward()` N times, then
one `optimizer.step()` -- ndsis/training/training.py:436,458-460; `batches_per_step` = 2 or 6 with the mask head,
scannet_config/run.py:377-396): every micro-batch but the last runs its backward under `with fp.accumulate():` -- the
bucket hooks count nothing and pack nothing, autograd accumulates into `.grad` -- and the LAST backward, outside the block,
packs the accumulated `.grad` of a bucket when that bucket's last hook fires: ONE all-reduce per bucket and optimizer step.
A gradient hook that fires a second time after its bucket has been counted (a second backward with the hooks armed and no
`zero_grad()` in between) raises: the slice it belongs to may already be on the wire.
"""
from __future__ import annotations

import contextlib
import os

import torch
import torch.distributed as dist


def _dist_on(min_world=2):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() >= min_world


class FlatParams:
    """Parameters are views into one flat buffer.  Gradients are produced by autograd as separate tensors (`.grad` is
    reset to None each step, so AccumulateGrad adopts the kernel's output instead of launching one add per parameter);
    `gather_grads` packs them into the flat bucket with ONE multi-tensor copy."""

    TAIL_FRACTION = 0.05          # share of the gradient bytes left to the last (exposed) bucket
    EVENT_RING = 32               # steps whose exposed all-reduce time is kept (HIP events around the final waits)

    def __init__(self, module: torch.nn.Module, n_buckets: int = 0):
        """n_buckets > 0: overlap the gradient all-reduce with backward (only used when a process group with more than
        one rank exists); 0: one all-reduce after backward."""
        self.params = [p for p in module.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.empty(total, dtype=ref.dtype, device=ref.device)
        self.flat_grad = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.grad_views = []
        # this rank's share of a count-weighted mean (the reference normalises its losses by batch-level counts,
        # loss.py:401-431: a scene contributes in proportion to its rows).  The bucketed path applies it when a slice is
        # packed, i.e. DURING backward: set it before backward starts.
        self.rank_weight = 1.0
        self.grad_scale = 1.0         # mean gradient = flat_grad * grad_scale (valid after all_reduce_mean)
        self.flat_grad_valid = False  # False after a step that never packed the flat buffer (step_single_rank fast path)
        self.sync = True              # False inside `accumulate()`: a backward only accumulates into `.grad`
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p)
            self.grad_views.append(self.flat_grad[off:off + n].view_as(p))
            off += n
        self._datas = [p.data for p in self.params]
        self._events, self._ev_used = [], 0
        self._setup_buckets(n_buckets)

    # ---- overlapped, bucketed all-reduce -----------------------------------------------------------------------
    def _setup_buckets(self, n_buckets):
        self.buckets = []                 # (param indices, flat slice), in the order backward completes them
        self._bucket_of, self._pending, self._works = {}, [], []
        self._ready, self._next = [], 0
        # (SCN_DP_FORCE_BUCKETS: take the overlapped path with a single rank too -- a 1-GPU rehearsal of the RCCL calls)
        min_world = 1 if os.environ.get("SCN_DP_FORCE_BUCKETS") else 2
        if n_buckets <= 0 or not _dist_on(min_world):
            return
        total = self.flat.numel()
        # The LAST bucket is complete only when backward ends, so its all-reduce is the one nothing hides: it gets the
        # parameters backward reaches last (the first encoder levels: a few per cent of the bytes) and the other buckets
        # share the rest equally.  (Equal cuts left 12 MB of the 50 MB at 32->256 channels exposed after backward.)
        tail = int(total * self.TAIL_FRACTION) if n_buckets > 1 else 0
        target = -(-(total - tail) // max(1, n_buckets - 1)) if n_buckets > 1 else total
        acc, idx = 0, []
        off = total
        tail_started = False

        def close():
            nonlocal acc, idx
            self.buckets.append((idx, self.flat_grad[off:off + acc]))
            idx, acc = [], 0
        for i in range(len(self.params) - 1, -1, -1):          # reverse parameter order ~ order of backward
            n = self.params[i].numel()
            idx.append(i); acc += n; off -= n
            if i == 0:
                close()
            elif tail_started or len(self.buckets) >= n_buckets - 1:
                continue                                       # everything that is left goes into the last bucket
            elif off <= tail:                                  # crossing into the tail: the body buckets end here, once
                close()
                tail_started = True
            elif acc >= target:
                close()
        for b, (ids, _) in enumerate(self.buckets):
            for i in ids:
                self._bucket_of[i] = b
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))
        self._reset_buckets()

    def _reset_buckets(self):
        self._pending = [len(ids) for ids, _ in self.buckets]
        self._ready = [False] * len(self.buckets)
        self._fired = [False] * len(self.params)
        self._next = 0
        self._works = []

    @contextlib.contextmanager
    def accumulate(self):
        """Micro-batches of one optimizer step (training.py:436,458-460): inside the block a backward accumulates into
        `.grad` and no slice is packed or reduced; run the LAST micro-batch's backward outside it.  (torch DDP's `no_sync`.)"""
        prev, self.sync = self.sync, False
        try:
            yield self
        finally:
            self.sync = prev

    def _make_hook(self, i):
        def hook(_param):
            if not self.sync:                    # an accumulating micro-batch: autograd has added into .grad, nothing else
                return
            if self._fired[i]:
                raise RuntimeError(
                    "FlatParams: the gradient hook of parameter %d fired a second time since zero_grad() -- its slice has "
                    "been counted (and may be on the wire) with the FIRST backward's gradient only.  Accumulate micro-"
                    "batches under `with fp.accumulate():` and run only the last backward outside it "
                    "(training.py:436,458-460), or call zero_grad() between steps." % i)
            self._fired[i] = True
            b = self._bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._ready[b] = True
                self._drain()
        return hook

    def _drain(self, force=False):
        """Launch buckets strictly in index order (the same order on every rank): bucket b goes out when it is complete and
        every bucket before it has gone out; force: the rest, complete or not (parameters without a gradient: zeros)."""
        while self._next < len(self.buckets) and (force or self._ready[self._next]):
            self._launch_bucket(self._next)
            self._next += 1

    def _launch_bucket(self, b):
        ids, flat_slice = self.buckets[b]
        have = [(self.grad_views[i], self.params[i].grad) for i in ids if self.params[i].grad is not None]
        if len(have) != len(ids):
            flat_slice.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if self.rank_weight != 1.0:
            flat_slice.mul_(self.rank_weight)
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
        self.flat_grad_valid = True

    # ---- exposed all-reduce time (HIP events on the compute stream around the final waits) ---------------------
    def _event_pair(self):
        if not self.flat.is_cuda:
            return None
        if len(self._events) < self.EVENT_RING:
            self._events.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
        pair = self._events[self._ev_used % self.EVENT_RING]
        self._ev_used += 1
        return pair

    def exposed_allreduce_ms(self):
        """Mean time the compute stream spent waiting for gradient all-reduces after backward had ended, over the last
        <= EVENT_RING steps (None without a GPU or before the first bucketed step).  Call after a synchronize."""
        n = min(self._ev_used, len(self._events))
        if n == 0:
            return None
        return sum(a.elapsed_time(b) for a, b in self._events[:n]) / n

    def all_reduce_mean(self, weight: float | None = None, total_weight: float | None = None):
        """Sum `rank_weight x gradient` over ranks; afterwards the mean gradient is `flat_grad x grad_scale` with
        grad_scale = 1 / total_weight (default: 1 / world size), which `sgd_step` folds into its alpha.

        weight: this rank's share (None: `self.rank_weight`).  On the bucketed path the slices were packed and scaled
        during backward, so a weight given here must equal the `rank_weight` that was set before backward."""
        if not self.sync:
            raise RuntimeError("all_reduce_mean inside `accumulate()`: the block is for the micro-batches BEFORE the last one")
        w = self.rank_weight if weight is None else float(weight)
        world = float(dist.get_world_size()) if _dist_on(1) else 1.0
        if self.buckets:
            if w != self.rank_weight:
                raise ValueError("bucketed all-reduce: set FlatParams.rank_weight BEFORE backward (slices are packed and "
                                 "scaled from the gradient hooks); got weight=%r, rank_weight=%r" % (w, self.rank_weight))
            ev = self._event_pair()
            self._drain(force=True)          # slices whose parameters received no (or not all) gradients, in order
            if ev is not None:
                ev[0].record()
            for wk in self._works:
                wk.wait()
            if ev is not None:
                ev[1].record()
            self.grad_scale = 1.0 / (total_weight if total_weight is not None else world)
            self.flat_grad_valid = True
            return
        self.gather_grads()
        if w != 1.0:
            self.flat_grad.mul_(w)
        if _dist_on(2):
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)
        self.grad_scale = 1.0 / (total_weight if total_weight is not None else world)

    def mean_grad(self) -> torch.Tensor:
        """The all-reduced MEAN gradient as one flat tensor (a copy when grad_scale != 1).  Only valid after a step that
        packed the flat buffer (all_reduce_mean / gather_grads): the single-rank fast path of step_single_rank does not."""
        if not self.flat_grad_valid:
            raise RuntimeError("flat_grad was not packed by the last step (step_single_rank fast path): call "
                               "gather_grads() / all_reduce_mean(), or set SCN_STEP_PACKED=1")
        return self.flat_grad if self.grad_scale == 1.0 else self.flat_grad * self.grad_scale

    def mean_grad_views(self):
        """Per-parameter views of mean_grad(), in `self.params` order."""
        flat, out, off = self.mean_grad(), [], 0
        for p in self.params:
            out.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return out

    def sgd_step(self, lr: float):
        self.flat.add_(self.flat_grad, alpha=-lr * self.grad_scale)

    def step_single_rank(self, lr: float):
        """all_reduce_mean + sgd_step for ONE rank without a process group: there is nothing to reduce, so the gradients
        need not be packed into the flat bucket first (a 50 MB copy at 32->256 channels) -- the update reads them where
        autograd left them, one multi-tensor launch.  Same arithmetic per element as sgd_step.  The flat gradient buffer
        is NOT filled on this path (`flat_grad_valid` False; `mean_grad()` raises)."""
        if self.buckets or os.environ.get("SCN_STEP_PACKED") or _dist_on(2) or self.rank_weight != 1.0:
            self.all_reduce_mean()
            self.sgd_step(lr)
            return
        self.flat_grad_valid = False
        if self._datas[0].data_ptr() != self.params[0].data_ptr() or self._datas[-1].data_ptr() != self.params[-1].data_ptr():
            self._datas = [p.data for p in self.params]        # (a parameter was re-pointed after construction: module.to(...))
        grads = [p.grad for p in self.params]
        # (`None in grads` would call Tensor.__eq__(None) on every entry: 11 us each, 1.7 ms for the 156 tensors of cfg 3)
        if any(g is None for g in grads):
            have = [(d, g) for d, g in zip(self._datas, grads) if g is not None]
            if have:
                torch._foreach_add_([d for d, _ in have], [g for _, g in have], alpha=-lr)
        else:                                   # (the views into `flat` are made once: `p.data` builds a tensor per access)
            torch._foreach_add_(self._datas, grads, alpha=-lr)


def broadcast_params(fp: FlatParams, src: int = 0):
    if _dist_on(2):
        dist.broadcast(fp.flat, src)
