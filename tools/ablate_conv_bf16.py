"""Developer tool (GPU box): scn_conv_tiles_bf16 against scn_conv_tiles (fp32) on the levels of the cfg-2 scene.
    python tools/ablate_conv_bf16.py"""
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
    n, P, t = rb.n, rb.rules.total, rb.tiles
    X = torch.randn(n, C, device="cuda"); W = torch.randn(27, C, C, device="cuda") * 0.05
    Xb = X.to(torch.bfloat16)
    us32 = timed(lambda: F.conv_rules(X, t, n, W, None, C, L.F_RELU_IN, n_rules=P))
    img = F.pack_weights_bf16(W, C, C, 27, 0)
    us16 = timed(lambda: F.conv_rules_bf16(Xb, t, n, W, None, C, L.F_RELU_IN, image=img))
    uspk = timed(lambda: F.pack_weights_bf16(W, C, C, 27, 0))
    F.FUSED_K = False
    us16_2 = timed(lambda: F.conv_rules_bf16(Xb, t, n, W, None, C, L.F_RELU_IN, image=img))
    F.FUSED_K = True
    y32 = F.conv_rules(Xb.float(), t, n, W.to(torch.bfloat16).float(), None, C, L.F_RELU_IN, n_rules=P)
    y16 = F.conv_rules_bf16(Xb, t, n, W, None, C, L.F_RELU_IN).float()
    err = ((y16 - y32).abs().max() / y32.abs().max()).item()
    fl = 2.0 * P * C * C
    print(f"L{level} C={C:3d} n={n:6d} P={P:7d}  fp32 {us32:6.1f} us {fl / us32 / 1e6:6.1f} TF   bf16 {us16:6.1f} us {fl / us16 / 1e6:6.1f} TF"
          f"   x{us32 / us16:.2f}  (2-launch K sum {us16_2:6.1f} us, weight pack {uspk:5.1f} us; compulsory bytes "
          f"{(4.0 * n * C + 2 * 27 * C * C + 8.0 * P) / 1e6:5.1f} MB -> {(4.0 * n * C + 2 * 27 * C * C + 8.0 * P) / us16 / 1e6:5.2f} TB/s)   max|bf16 - fp32 on the same rounded operands| / scale {err:.1e}")
    sz = tuple(s // 2 for s in sz)
