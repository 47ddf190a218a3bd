"""Round 5 probe: which torch operators (not this library's C calls) a step still issues -- name, calls per step, device time per
step (torch.profiler over a few steps).  python tools/r5_torch_ops.py <workload> <dtype> [steps]"""
import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from sparse_rcnn_amd.trainstep import SceneStep
wl, dt = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
job = SceneStep(wl, torch.device("cuda", 0), dtype=dt, prefetch=True, seed=1)
for _ in range(15):
    job.step()
job.finish(); torch.cuda.synchronize(); gc.collect()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(n):
        job.step()
    job.finish(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=6):
    dev = getattr(e, "self_device_time_total", None)
    if dev is None:
        dev = getattr(e, "self_cuda_time_total", 0)
    if dev > 0 and e.key.startswith("aten::"):
        stack = [s for s in e.stack if "sparse_rcnn_amd" in s or "tools/" in s][:3]
        rows.append((dev / n, e.count / n, e.key, " <- ".join(s.split("sparse_rcnn_amd/")[-1] for s in stack)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{wl} {dt}: {tot:.0f} us of device time per step in torch operators")
for dev, cnt, key, stack in rows[:60]:
    print(f"{dev:8.1f} us/step {cnt:6.1f} calls/step  {key:28s} {stack[:170]}")

# where they come from: a dispatch mode sees every aten call (also the autograd engine's) and walks the Python frames
import collections
from torch.utils._python_dispatch import TorchDispatchMode
sites = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        if name.split("::")[-1].rstrip("_") in ("copy", "_to_copy", "add", "fill", "zero", "cat", "mul", "index", "nonzero", "clone", "zeros", "sum", "arange", "stack", "exp", "div", "sub"):
            f, fr = sys._getframe(1), []
            while f is not None and len(fr) < 3:
                fn = f.f_code.co_filename
                if "sparse_rcnn_amd/" in fn:
                    fr.append(f"{fn.split('sparse_rcnn_amd/')[-1]}:{f.f_lineno}")
                f = f.f_back
            big = max([a.numel() for a in args if isinstance(a, torch.Tensor)] + [0])
            sites[(name, " <- ".join(fr) or "(no package frame)", "big" if big > 4096 else "small")] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    for _ in range(n):
        job.step()
    job.finish(); torch.cuda.synchronize()
for (name, site, size), c in sorted(sites.items(), key=lambda kv: (-kv[1]))[:80]:
    print(f"{c / n:6.1f} calls/step  {name:18s} {size:5s} {site[:200]}")
