"""Developer tool (GPU box): wall time per group of 5 steps over a long run (warm-up effects)."""
import os, sys, time, gc
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
model = Backbone(7, (32, 64, 128, 256)).to(dev)
flat = FlatParams(model)
gy = None
def step():
    global gy
    flat.zero_grad()
    fin = feats_d.detach().requires_grad_()
    out = model(coords_d, fin, size, 1)
    if gy is None:
        gy = torch.randn_like(out.features)
    out.features.backward(gy)
    flat.all_reduce_mean()
    flat.sgd_step(1e-6)
if "freeze" in sys.argv:
    step(); gc.collect(); gc.freeze()
if "events" in sys.argv:
    pool = [torch.cuda.Event(enable_timing=True) for _ in range(1280)]
if "recorded" in sys.argv:
    pool = [torch.cuda.Event(enable_timing=True) for _ in range(1280)]
    for e in pool: e.record()
for grp in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"steps {5*grp:3d}-{5*grp+4:3d}: {(t1-t0)*200:.2f} ms/step  alloc {torch.cuda.memory_reserved()/2**20:.0f} MiB", flush=True)
