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
        self.exec_records = []     # (name, flops, bytes, ms): launches the step executor timed inside its C calls
        self.exec_shapes = {}      # (op, bf16, cin, cout, rows out) -> [launches, ms, flops]: the same launches by layer shape

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

    # ---- the step executor (executor.py): a level's launches sit inside one C call, which brackets its tile-convolution
    # launches itself while `exec_timing()` is true (scn_exec_timing_enable); the sampled steps stay on the production path
    def exec_timing(self):
        return self.active and not self.count_only and (self.names is None or ("k_conv_ts" in self.names or "k_conv_tb" in self.names))

    def collect_exec(self):
        """Fetch what the executor recorded since the last call (waits for those events)."""
        import ctypes as C
        from . import _lib as L
        cap = 4096
        ms = (C.c_float * cap)()
        info = (C.c_int64 * (7 * cap))()
        while True:
            n = L.lib().scn_exec_timing_collect(ms, info, cap)
            for k in range(n):
                op, bf16, cin, cout, n_in, n_out, rules = (int(info[7 * k + j]) for j in range(7))
                if op not in (2, 3):                                  # (scn_exec_timing_enable(2) times every op: a tool's mode)
                    continue
                n_off = 8 if op == 3 else 27                          # SCN_OP_CONV_CHILD = 3
                es = 2.0 if bf16 else 4.0
                sh = self.exec_shapes.setdefault((op, bf16, cin, cout, n_out), [0, 0.0, 0.0])
                sh[0] += 1
                sh[1] += float(ms[k])
                sh[2] += 2.0 * rules * cin * cout
                self.exec_records.append(("k_conv_tb" if bf16 else "k_conv_ts", 2.0 * rules * cin * cout,
                                          es * (n_in * cin + n_out * cout + (n_off * cin * cout if not bf16 else 0))
                                          + (2.0 * n_off * cin * cout if bf16 else 0.0) + 8.0 * rules, float(ms[k])))
            if n < cap:
                break

    def summary(self):
        """name -> dict(launches, ms, flops, bytes) over the sampled launches.  Call after torch.cuda.synchronize()."""
        self.collect_exec()
        out = {}
        for name, flops, nbytes, t in self.exec_records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += t
            d["flops"] += flops
            d["bytes"] += nbytes
        for name, flops, nbytes, s, e in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += flops() if callable(flops) else flops
            d["bytes"] += nbytes() if callable(nbytes) else nbytes
        return out

    def by_shape(self, steps):
        """The executor-timed tile-convolution launches by layer shape, largest level first: launches and us per step, us per
        launch and useful TFLOP/s (HIP events inside the C calls; bench.py reports it as roofline.by_shape)."""
        out = []
        for (op, bf16, cin, cout, n_out), (c, ms, fl) in sorted(self.exec_shapes.items(), key=lambda kv: (-kv[0][4], kv[0])):
            out.append(dict(table="child 2^3" if op == 3 else "subm 3^3", bf16=bool(bf16), cin=cin, cout=cout, rows_out=n_out,
                            launches_per_step=c / steps, us_per_launch=ms * 1e3 / c, us_per_step=ms * 1e3 / steps,
                            useful_tflops=fl / (ms * 1e-3) / 1e12 if ms else None))
        return out

    @property
    def sampled_steps(self):
        return (self.step // self.every) + 1 if self.step >= 0 else 0


TIMER: KernelTimer | None = None


def exec_ok():
    """May a forward take the step executor?  Yes without a timer, and with one that only samples the dominant tile kernel
    (the executor's C calls bracket those launches themselves); a timer that counts launches or times every kernel needs
    the layer-by-layer path."""
    t = TIMER
    return t is None or (not t.count_only and t.names is not None and t.names <= {"k_conv_ts", "k_conv_tb"})


def timed(name, flops, nbytes, fn):
    """flops / nbytes may be callables: they are evaluated in summary(), not on the launch path (rule counts come back
    from the device asynchronously -- metadata.Rules)."""
    if TIMER is None:
        return fn()
    return TIMER.launch(name, flops, nbytes, fn)
