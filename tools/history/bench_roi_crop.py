"""Developer tool (GPU box): the sparse ROI crop at BASELINE config 3 size -- 64 boxes x ~172 k points -- timed, with the
bytes it has to move (coords read + selection written) against the [boxes, points] objects the reference builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd import roi
from sparse_rcnn_amd.synthetic import make_batch, make_boxes

coords, feats, size, bs, splits = make_batch(1, (512, 512, 256), 150000, dup=1.15, seed=1)
boxes = make_boxes(coords, 64, seed=3)
b32, counts, assoc = roi.transform_boxes(boxes, size, False)
c32 = roi._coords_to_device(coords)
f23 = torch.randn(len(coords), 23, device="cuda")


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


sel = roi.roi_select(c32, b32)
n, m, bb = len(coords), sel.src_row.shape[0], 64
us_sel = timed(lambda: roi.roi_select(c32, b32))
us_feat = timed(lambda: roi.select_features(f23, sel))
us_dense = timed(lambda: roi.roi_select(c32, b32).is_inside_u8())
alg = 2 * 16.0 * n + m * (4 + 4 + 32)
print(f"{bb} boxes x {n} points -> {m} selected rows")
print(f"roi_select (count + scan + fill, one host wait for the row counts): {us_sel:7.1f} us; algorithmic bytes "
      f"{alg / 1e6:.1f} MB (coords twice + src_row / box_of / int64 coords written) -> {alg / us_sel / 1e3:.1f} GB/s")
print(f"select_features, 23 channels ({m} rows gathered): {us_feat:7.1f} us = {2.0 * m * 23 * 4 / us_feat / 1e3:.0f} GB/s")
print(f"with the reference's dense indicator on request (+ {bb * n / 1e6:.1f} MB u8 zero-fill + scatter): {us_dense:7.1f} us")
print(f"the reference's own objects at this size: bool [boxes, points] {bb * n / 1e6:.1f} MB, expanded-view gather over "
      f"{bb * n * 23 * 4 / 1e9:.2f} GB of (virtual) features, int64 [boxes, points, 3] coordinate view")
