"""us per scn_conv_tiles_bf16_pack_many call over the weight images of the cfg-2 backbone (forward + backward-data images of
every tile-kernel layer: what a bf16 step packs once), against the bytes it moves."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd import functional as F
from sparse_rcnn_amd.unet import Backbone

dev = torch.device("cuda:0")
net = Backbone(7, (32, 64, 128, 256), bf16_blocks="all").to(dev)
plan = F.PackPlan(net.unet._pack_jobs())
read = sum(j[0].numel() * 4 for j in plan.jobs)
from sparse_rcnn_amd import _lib as L
buf = torch.empty(plan.total, dtype=torch.uint8, device=dev)
for i, o in enumerate(plan.offs):
    plan.Ip[i] = buf.data_ptr() + o
lib = L.lib()


def call():
    L.check(lib.scn_conv_tiles_bf16_pack_many(plan.n, plan.Wp, plan.ci, plan.co, plan.no, plan.fl, plan.Ip, L.stream()))


for _ in range(5):
    call()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
a.record()
for _ in range(n):
    call()
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) / n * 1e3
print(f"pack_many: {plan.n} images, {read / 1e6:.1f} MB read + {plan.total / 1e6:.1f} MB written, {us:.1f} us per back-to-back "
      f"call = {(read + plan.total) / us / 1e6:.2f} TB/s")
