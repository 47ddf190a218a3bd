"""Developer tool (GPU box): fresh pageable host coordinates per batch -- direct .to(device) against a persistent pinned
staging buffer.    python tools/diag_host_coords3.py direct|staged"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd.synthetic import make_batch
from sparse_rcnn_amd.metadata import Metadata
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, dup=1.15, seed=1)
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev); torch.cuda.synchronize()
mode = sys.argv[1]
stage = torch.empty(coords.shape, dtype=coords.dtype).pin_memory()
ts = []
for i in range(60):
    t0 = time.perf_counter()
    x = coords.clone()
    t1 = time.perf_counter()
    if mode == "direct":
        y = x.to(dev)
    else:
        stage.copy_(x)
        y = stage.to(dev, non_blocking=True)
    t2 = time.perf_counter()
    md = Metadata(3).build_native(size, y, 1, 4, 4, 3)
    t3 = time.perf_counter()
    del x
    t4 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
import numpy as np
a = np.array(ts[5:])
for k, name in enumerate(("clone", "to device", "build", "free")):
    print(f"{mode:7s} {name:10s} median {np.median(a[:, k]):6.2f}  mean {a[:, k].mean():6.2f}  max {a[:, k].max():6.2f} ms")
print(f"{mode:7s} per iteration mean {a.sum(1).mean():.2f} ms")
