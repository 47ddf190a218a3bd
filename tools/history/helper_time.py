import os, sys, time, gc
sys.path.insert(0, "/root/repo")
import torch
import sparse_rcnn_amd  # noqa
from sparse_rcnn_amd.dp import FlatParams
from sparse_rcnn_amd.synthetic import make_batch
from sparse_rcnn_amd.unet import Backbone
from sparse_rcnn_amd import metadata as MD
dev = torch.device("cuda", 0)
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, dup=1.15, seed=1)
coords_d, feats_d = coords.to(dev), feats.to(dev)
model = Backbone(7, (32, 64, 128, 256)).to(dev)
flat = FlatParams(model)
gy = None; md_next = None
durs = []
orig = MD.Metadata.prepare_async
def timed_prepare(self, *a, **k):
    t = time.perf_counter(); r = orig(self, *a, **k); durs.append(time.perf_counter() - t); return r
MD.Metadata.prepare_async = timed_prepare
waits = []
def step():
    global gy, md_next
    flat.zero_grad()
    fin = feats_d.detach().requires_grad_()
    t = time.perf_counter()
    md = md_next.result() if md_next is not None else None
    waits.append(time.perf_counter() - t)
    md_next = model.prefetch_in_thread(coords_d, size, 1)
    out = model(coords_d, fin, size, 1, metadata=md)
    if gy is None: gy = torch.randn_like(out.features)
    out.features.backward(gy)
    flat.all_reduce_mean(); flat.sgd_step(1e-6)
for _ in range(5): step()
gc.collect(); gc.freeze(); durs.clear(); waits.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print(f"step {dt*1e3:.2f} ms; helper build mean {1e3*sum(durs)/len(durs):.2f} ms max {1e3*max(durs):.2f}; main waits for it mean {1e3*sum(waits)/len(waits):.3f} ms")
