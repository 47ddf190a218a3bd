"""Per-launch HIP-event timing of the hot kernels, recorded on the stream the kernels are launched on (torch's
current stream: every libscn_mi355x call receives ``torch.cuda.current_stream().cuda_stream``)."""
from __future__ import annotations

import torch


class KernelTimer:
    def __init__(self):
        self.records = []          # (name, flops, bytes, start, end)

    def launch(self, name, flops, nbytes, fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.records.append((name, flops, nbytes, s, e))
        return out

    def summary(self):
        """name -> dict(launches, ms, flops, bytes).  Call after torch.cuda.synchronize()."""
        out = {}
        for name, flops, nbytes, s, e in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


TIMER: KernelTimer | None = None


def timed(name, flops, nbytes, fn):
    if TIMER is None:
        return fn()
    return TIMER.launch(name, flops, nbytes, fn)
