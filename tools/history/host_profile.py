"""Developer tool (GPU box): cProfile of the host thread over a few bench steps (the GPU runs asynchronously, so the
profile shows where the enqueueing thread spends its time)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_rcnn_amd  # noqa
from sparse_rcnn_amd.dp import FlatParams
from sparse_rcnn_amd.synthetic import make_batch
from sparse_rcnn_amd.unet import Backbone

dev = torch.device("cuda", 0)
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, dup=1.15, seed=1)
coords_d, feats_d = coords.to(dev), feats.to(dev)
torch.manual_seed(0)
model = Backbone(7, (32, 64, 128, 256), bf16_blocks="all" if "--bf16" in sys.argv else False).to(dev)
flat = FlatParams(model)
gy = None
md_next = None
PREFETCH = "--no-prefetch" not in sys.argv
def step():
    global gy, md_next
    flat.zero_grad()
    fin = feats_d.detach().requires_grad_()
    md = md_next.result() if md_next is not None else None
    md_next = model.prefetch_in_thread(coords_d, size, 1) if PREFETCH else None
    out = model(coords_d, fin, size, 1, metadata=md)
    if gy is None:
        gy = torch.randn_like(out.features)
    out.features.backward(gy)
    flat.all_reduce_mean()
    flat.sgd_step(1e-6)
for _ in range(5): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/10:.2f} ms/step, wall {1e3*(t2-t0)/10:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(10): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:6000])
