"""Developer tool (GPU box): wall time of the phases of one bench step (index build / forward / backward)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd.synthetic import make_batch
from sparse_rcnn_amd.unet import Backbone
from sparse_rcnn_amd.ioLayers import InputLayer
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, seed=1)
dev = torch.device("cuda")
coords_d, feats_d = coords.to(dev), feats.to(dev)
model = Backbone(7, (32, 64, 128, 256)).to(dev)
gy = None
def sync(): torch.cuda.synchronize(); return time.perf_counter()
T = {"input": 0, "pyramid": 0, "fwd": 0, "bwd": 0}
for it in range(8):
    for p in model.parameters(): p.grad = None
    fin = feats_d.detach().requires_grad_()
    t0 = sync()
    x = InputLayer(3, size, mode=4)((coords_d, fin, 1))
    t1 = sync()
    x.metadata.build_pyramid(size, 4, 3)
    t2 = sync()
    out = model.unet(x)
    t3 = sync()
    if gy is None: gy = torch.randn_like(out.features)
    out.features.backward(gy)
    t4 = sync()
    if it >= 3:
        T["input"] += t1 - t0; T["pyramid"] += t2 - t1; T["fwd"] += t3 - t2; T["bwd"] += t4 - t3
print({k: round(v / 5 * 1e3, 3) for k, v in T.items()}, "ms")
