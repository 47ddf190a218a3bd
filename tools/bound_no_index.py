import os, sys, time, gc
sys.path.insert(0, "/root/repo")
import torch
import sparse_rcnn_amd  # noqa
from sparse_rcnn_amd.dp import FlatParams
from sparse_rcnn_amd.synthetic import make_batch
from sparse_rcnn_amd.unet import Backbone
from sparse_rcnn_amd.metadata import Metadata
dev = torch.device("cuda", 0)
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, dup=1.15, seed=1)
coords_d, feats_d = coords.to(dev), feats.to(dev)
torch.manual_seed(0)
model = Backbone(7, (32, 64, 128, 256)).to(dev)
flat = FlatParams(model)
gy = None
def step(md=None):
    global gy
    flat.zero_grad()
    fin = feats_d.detach().requires_grad_()
    out = model(coords_d, fin, size, 1, metadata=md)
    if gy is None: gy = torch.randn_like(out.features)
    out.features.backward(gy)
    flat.all_reduce_mean(); flat.sgd_step(1e-6)
for _ in range(5): step()
gc.collect(); gc.freeze()
def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("full step             %.2f ms" % timeit(step))
# reuse one prepared Metadata (index structures built once): lower bound of a perfectly hidden index build
md = Metadata(3).prepare_async(size, coords_d, 1, 4, 4, 3)
torch.cuda.synchronize()
class Keep:
    pass
def step_cached():
    m = Metadata(3)
    m.__dict__.update(md.__dict__)
    m.ready_event = None
    step(m)
for _ in range(3): step_cached()
print("index build excluded  %.2f ms" % timeit(step_cached))
