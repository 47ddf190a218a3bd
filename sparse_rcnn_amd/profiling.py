"""Per-launch HIP-event timing of the hot kernels, recorded on the stream the kernels are launched on (torch's
current stream: every libscn_mi355x call receives ``torch.cuda.current_stream().cuda_stream``).

Creating and recording two timing events per launch costs the host ~5-10 us; at ~230 launches per step that turned a
GPU-bound step into a host-bound one (measured: 8.9 ms/step untimed vs 9.8-12.5 ms with every launch timed).  The
timer therefore (a) takes its events from a pool created BEFORE the timed region and (b) samples: only every
``every``-th step is instrumented.  The averages it reports are over the sampled launches of the timed region."""
from __future__ import annotations

import torch


class KernelTimer:
    def __init__(self, every: int = 1, names=None):
        self.every = max(1, int(every))
        self.names = set(names) if names else None      # only these launch names are timed (None = all)
        self.step = -1
        self.active = True
        self.count_only = False
        self.count = 0
        self.pool = []
        self.used = 0
        self.records = []          # (name, flops, bytes, start, end)

    def reserve(self, n_events: int):
        while len(self.pool) < n_events:
            self.pool.append(torch.cuda.Event(enable_timing=True))

    def begin_step(self):
        self.step += 1
        self.active = self.step % self.every == 0

    def _event(self):
        if self.used == len(self.pool):
            self.pool.append(torch.cuda.Event(enable_timing=True))
        e = self.pool[self.used]
        self.used += 1
        return e

    def launch(self, name, flops, nbytes, fn):
        if self.count_only:
            self.count += self.names is None or name in self.names
            return fn()
        if not self.active or (self.names is not None and name not in self.names):
            return fn()
        s, e = self._event(), self._event()
        s.record()
        out = fn()
        e.record()
        self.records.append((name, flops, nbytes, s, e))
        return out

    def summary(self):
        """name -> dict(launches, ms, flops, bytes) over the sampled launches.  Call after torch.cuda.synchronize()."""
        out = {}
        for name, flops, nbytes, s, e in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += flops() if callable(flops) else flops
            d["bytes"] += nbytes() if callable(nbytes) else nbytes
        return out

    @property
    def sampled_steps(self):
        return (self.step // self.every) + 1 if self.step >= 0 else 0


TIMER: KernelTimer | None = None


def timed(name, flops, nbytes, fn):
    """flops / nbytes may be callables: they are evaluated in summary(), not on the launch path (rule counts come back
    from the device asynchronously -- metadata.Rules)."""
    if TIMER is None:
        return fn()
    return TIMER.launch(name, flops, nbytes, fn)
