"""Developer tool (GPU box): scn_wgrad_rules_bf16 against scn_wgrad_rules on the levels of the cfg-2 scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L, functional as F
from sparse_rcnn_amd.synthetic import make_batch
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata
md.build_pyramid(size, 4, 3)
sz = tuple(int(s) for s in size)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
for level, C in enumerate([32, 64, 128, 256]):
    rb = md.subm_rulebook(sz, 3)
    n, r = rb.n, rb.rules
    P = r.total
    X = torch.randn(n, C, device="cuda"); G = torch.randn(n, C, device="cuda")
    Xb, Gb = X.to(torch.bfloat16), G.to(torch.bfloat16)
    us32 = timed(lambda: F.wgrad_rules(X, G, r.in_rows, r.out_rows, r.prefix_host, 27, L.F_RELU_IN))
    us16 = timed(lambda: F.wgrad_rules_bf16(Xb, Gb, r.in_rows, r.out_rows, r.prefix_host, 27, L.F_RELU_IN))
    fl = 2.0 * P * C * C
    print(f"L{level} C={C:3d} P={P:7d}  fp32 {us32:6.1f} us {fl / us32 / 1e6:6.1f} TF   bf16 storage {us16:6.1f} us {fl / us16 / 1e6:6.1f} TF   x{us32 / us16:.2f}")
    sz = tuple(s // 2 for s in sz)
