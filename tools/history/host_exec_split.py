"""Developer tool (GPU box): host time of the step executor per step -- inside the C calls (scn_exec_run: the kernel launches)
vs the Python around them (tables, layouts, allocations, gradient views), forward and backward.
    python tools/host_exec_split.py [cfg2|cfg3] [f32|bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd import executor as EX
from sparse_rcnn_amd.trainstep import SceneStep
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
acc = {"c": 0.0, "fwd": 0.0, "bwd": 0.0, "n": 0}
_run0 = EX._run
def _run(*a, **k):
    t = time.perf_counter(); r = _run0(*a, **k); acc["c"] += time.perf_counter() - t; return r
EX._run = _run
f0, b0 = EX.StageFunction.forward, EX.StageFunction.backward
def fwd(ctx, *a):
    t = time.perf_counter(); r = f0(ctx, *a); acc["fwd"] += time.perf_counter() - t; return r
def bwd(ctx, *a):
    t = time.perf_counter(); r = b0(ctx, *a); acc["bwd"] += time.perf_counter() - t; return r
EX.StageFunction.forward = staticmethod(fwd); EX.StageFunction.backward = staticmethod(bwd)
job = SceneStep(wl, torch.device("cuda", 0), dtype=dt)
for _ in range(10): job.step()
torch.cuda.synchronize()
for k in ("c", "fwd", "bwd"): acc[k] = 0.0
N = 30
t0 = time.perf_counter()
for _ in range(N): job.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
job.finish()
print(f"{wl} {dt}: host enqueue {1e3 * (t1 - t0) / N:.2f} ms/step, wall {1e3 * (t2 - t0) / N:.2f}; stage nodes fwd {1e3 * acc['fwd'] / N:.2f} + bwd {1e3 * acc['bwd'] / N:.2f} "
      f"of which inside scn_exec_run {1e3 * acc['c'] / N:.2f} ms/step")
