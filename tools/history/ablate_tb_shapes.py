"""Developer tool (GPU box): what holds a bf16 tile-convolution launch at the deeper levels?  One level of a scene, the
same tiles, channel shapes that change the K split (cin) and the column chunks (cout) independently, each as the fused
launch, as the tile kernel alone (partials to the slabs, no sum) and as the two-launch form.
    python tools/ablate_tb_shapes.py [voxels=600000] [grid=1024] [level=2]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import functional as F, _lib as L
from sparse_rcnn_amd.synthetic import make_batch

vox = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
g = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
level = int(sys.argv[3]) if len(sys.argv) > 3 else 2
coords, feats, size, bs, _ = make_batch(1, (g, g, g // 2), vox, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata
sz = tuple(int(s) for s in size)
for l in range(level):
    md.strided_rulebook(sz); sz = tuple(s // 2 for s in sz)
rb = md.subm_rulebook(sz, 3)
n, P, t = rb.n, rb.rules.total, rb.tiles
print(f"level {level} N={n} P={P}")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / reps
for cin, cout in ((32, 64), (32, 128), (64, 64), (64, 128), (128, 64), (128, 128), (256, 64), (256, 128), (256, 256), (32, 256)):
    X = torch.randn(n, cin, device="cuda").bfloat16(); W = torch.randn(27, cin, cout, device="cuda") * 0.05
    img = F.pack_weights_bf16(W, cin, cout, 27, 0)
    out = []
    for name, fl, fused in (("fused", 0, True), ("tile kernel only", L.F_SPLIT_SUM, False), ("two launches", 0, False)):
        F.FUSED_K = fused
        us = timeit(lambda: F.conv_rules_bf16(X, t, n, W, None, cout, fl, image=img))
        out.append(f"{name} {us:6.1f} us")
    F.FUSED_K = True
    print(f"  cin {cin:3d} cout {cout:3d}: " + "  ".join(out) + f"   ({2.0 * P * cin * cout / timeit(lambda: F.conv_rules_bf16(X, t, n, W, None, cout, 0, image=img)) / 1e6:6.1f} TF fused)", flush=True)
