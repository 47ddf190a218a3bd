"""Developer tool (GPU box): bench.py's pipelined step over a long run with a DIFFERENT scene every step (four scenes of
different sizes in rotation): per-50-step wall time, allocator high-water marks, finite outputs -- drift and leak check."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd.dp import FlatParams
from sparse_rcnn_amd.synthetic import make_batch
from sparse_rcnn_amd.unet import Backbone
dev = torch.device("cuda", 0)
scenes = []
for seed, target in ((1, 150000), (2, 120000), (3, 165000), (4, 90000)):
    c, f, size, bs, _ = make_batch(1, (512, 512, 256), target, dup=1.15, seed=seed)
    scenes.append((c.to(dev), f.to(dev), size))
torch.manual_seed(0)
BF16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"        # bf16 storage (XCD-local tile order, bf16 tile kernels)
model = Backbone(7, (32, 64, 128, 256), bf16_blocks="all" if BF16 else False).to(dev)
flat = FlatParams(model, n_buckets=4)
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
pending = model.prefetch_in_thread(scenes[0][0], scenes[0][2], 1)
t0 = time.perf_counter(); vox = 0
for it in range(n_steps):
    c, f, size = scenes[it % 4]
    md = pending.result()
    nc, _, nsize = scenes[(it + 1) % 4]
    pending = model.prefetch_in_thread(nc, nsize, 1)
    flat.zero_grad()
    out = model(c, f.detach().requires_grad_(), size, 1, metadata=md)
    out.features.backward(torch.ones_like(out.features))
    flat.step_single_rank(1e-12)        # (upstream gradient = ones on every row: keep the weights where they are)
    vox += out.features.shape[0]
    if it == 4:
        gc.collect(); gc.freeze()
    if (it + 1) % 50 == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ok = bool(torch.isfinite(out.features).all())
        print(f"steps {it - 48:4d}-{it + 1:4d}: {(t1 - t0) / 50 * 1e3:6.2f} ms/step  {vox / (t1 - t0) / 1e6:5.1f} M voxels/s  "
              f"reserved {torch.cuda.memory_reserved() / 2**20:6.0f} MiB  allocated {torch.cuda.memory_allocated() / 2**20:6.0f} MiB  finite {ok}", flush=True)
        t0 = time.perf_counter(); vox = 0
pending.result()
