"""Round 5 A/B inside ONE process: blocks of steps alternate between settings of trainstep's module switches, so box and
time drift cancel (separate bench.py runs of a host-bound step scatter by +-0.8 ms on one box).
python tools/r5_ab_inproc.py <workload> <dtype> name=FLAG:val,FLAG:val name2=... [--blocks 6] [--steps 25]"""
import sys, os, time, gc, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd import trainstep as TS
wl, dt = sys.argv[1], sys.argv[2]
modes, blocks, steps = [], 6, 25
args = sys.argv[3:]
while args:
    a = args.pop(0)
    if a == "--blocks": blocks = int(args.pop(0))
    elif a == "--steps": steps = int(args.pop(0))
    else:
        name, spec = a.split("=")
        modes.append((name, [(kv.split(":")[0], (int(kv.split(":")[1]) if kv.split(":")[1] not in "01" else kv.split(":")[1] == "1"))
                             for kv in spec.split(",") if kv]))
job = TS.SceneStep(wl, torch.device("cuda", 0), dtype=dt, prefetch=True, seed=1)
def apply(flags):
    from sparse_rcnn_amd._lib import switches
    for k, v in flags:
        if k.startswith("SCN_"):                 # a library switch (scn_debug_set): SCN_PYRAMID_V1:1 / :0 (0 = unset)
            if v:
                switches[k] = "1"
            else:
                del switches[k]
        elif "." in k:                           # a switch of another module of the package: proposals.TORCH_TOPK:1
            import importlib
            mod, attr = k.rsplit(".", 1)
            setattr(importlib.import_module("sparse_rcnn_amd." + mod), attr, v)
        else:
            setattr(TS, k, v)
for _ in range(20):
    job.step()
torch.cuda.synchronize(); gc.collect(); gc.freeze()
res = {n: [] for n, _ in modes}
for b in range(blocks):
    order = modes if b % 2 == 0 else modes[::-1]
    for name, flags in order:
        apply(flags)
        for _ in range(3):
            job.step()
        job.finish(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            job.step()
        job.finish(); torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / steps * 1e3)
for name, v in res.items():
    print(f"{wl} {dt} {name:40s} mean {statistics.mean(v):.3f} ms  min {min(v):.3f}  max {max(v):.3f}   blocks {[round(x, 2) for x in v]}", flush=True)
