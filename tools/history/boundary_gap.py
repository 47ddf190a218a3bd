"""Developer tool (GPU box): GPU-timeline gap at the step boundary of bench.py's loop, without a profiler: an event
after the SGD kernel of step i and one after the gradient-buffer fill of step i+1."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd.dp import FlatParams
from sparse_rcnn_amd.synthetic import make_batch
from sparse_rcnn_amd.unet import Backbone
dev = torch.device("cuda")
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, dup=1.15, seed=1)
cd, fd = coords.to(dev), feats.to(dev)
torch.manual_seed(0)
model = Backbone(7, (32, 64, 128, 256)).to(dev)
flat = FlatParams(model, n_buckets=4)
gy = None; md_next = None
ev = []
host = []
def step(record):
    global gy, md_next
    t0 = time.perf_counter()
    flat.zero_grad()
    if record:
        e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(("start", e))
    fin = fd.detach().requires_grad_()
    t1 = time.perf_counter()
    md = md_next.result() if md_next is not None else None
    t2 = time.perf_counter()
    md_next = model.prefetch_in_thread(cd, size, 1)
    if record:
        e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(("go", e))
    t3 = time.perf_counter()
    out = model(cd, fin, size, 1, metadata=md)
    t4 = time.perf_counter()
    if gy is None: gy = torch.randn(out.features.shape).to(dev)
    out.features.backward(gy)
    t5 = time.perf_counter()
    flat.all_reduce_mean(); flat.sgd_step(1e-6)
    if record:
        e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(("end", e))
    t6 = time.perf_counter()
    host.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5))
for _ in range(5): step(False)
gc.collect(); gc.freeze()
torch.cuda.synchronize(); host.clear()
T0 = time.perf_counter()
for _ in range(20): step(True)
md_next.result(); torch.cuda.synchronize()
print(f"ms/step {(time.perf_counter() - T0) / 20 * 1e3:.3f}")
ends = [e for k, e in ev if k == "end"]; starts = [e for k, e in ev if k == "start"]
gaps = [ends[i].elapsed_time(starts[i + 1]) * 1e3 for i in range(len(ends) - 1)]
body = [starts[i].elapsed_time(ends[i]) * 1e3 for i in range(len(ends))]
print("boundary (SGD end -> fill done) us:", [round(g) for g in gaps])
gos = [e for k, e in ev if k == "go"]
idle = [starts[i].elapsed_time(gos[i]) * 1e3 for i in range(len(gos))]
print("GPU timeline: fill done -> helper joined and next helper started (no kernels in between; = GPU idle) us:", [round(g) for g in idle])
print("step body (fill done -> SGD end) us: mean", round(sum(body) / len(body)))
import numpy as np
h = np.array(host) * 1e6
print("host us per phase [zero_grad, wait helper, start helper, forward, backward, sgd]:", h.mean(0).round().tolist(), "total", round(h.sum(1).mean()))
