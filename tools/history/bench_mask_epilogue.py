"""Developer tool (GPU box): the N2 mask-head epilogue kernels at BASELINE config-3 size (150 k points, 64 boxes, 18
classes): time per call and bytes moved against the HBM roofline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import sparse_rcnn_amd  # noqa
from sparse_rcnn_amd import roi
from sparse_rcnn_amd.synthetic import make_batch, make_boxes

K = 18
coords, feats, size, bs, splits = make_batch(1, (512, 512, 256), 150000, seed=1)
boxes = make_boxes(coords, 64)
b32, counts, assoc = roi.transform_boxes(boxes, size, False)
_, _, sel = roi.roi_cut_device(coords, feats.cuda(), b32)
m = sel.src_row.shape[0]
scores = torch.randn(m, K, device="cuda", requires_grad=True)
classes = torch.randint(0, K, (64,))
keep = [torch.ones(64, dtype=torch.bool)]
gassoc = [torch.randint(0, 20, (64,))]
glabels = [torch.randint(0, K, (20,))]
gmasks = [(torch.rand(20, splits[0]) < 0.3).cuda()]      # ground-truth masks live on the device, as in the reference

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / n

t1 = timeit(lambda: roi.mask_predict(scores, sel, counts, splits, classes))
t2 = timeit(lambda: roi.mask_loss_select(scores, sel, counts, splits, keep, gassoc, glabels, gmasks))
out_bytes = 64 * splits[0] * 4
print(f"crop rows M = {m}, boxes 64, points {splits[0]}, classes {K}")
print(f"mask_predict     {t1:8.1f} us   (kernel traffic ~{(m * (4 + 4 + 4) + out_bytes) / 1e6:.1f} MB incl. zero-fill of the dense [64, N] output)")
print(f"mask_loss_select {t2:8.1f} us   (kernel traffic ~{m * (4 + 4 + 4 + 4 + 4 + 1) / 1e6:.1f} MB + host-side list handling)")
