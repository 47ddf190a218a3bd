"""Developer tool (GPU box): how much of a bf16 tile-convolution launch is L2-miss time?  Times scn_conv_tiles_bf16 on the
levels of a scene with the real rulebook table and with every gather folded onto the first F rows (F = 1024: L1/L2
resident; F = 16384: L2 resident) -- same instruction stream, same steps, only the addresses differ.
    python tools/ablate_tb_locality.py [voxels=150000] [grid=512] [f32]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import functional as F
from sparse_rcnn_amd.synthetic import make_batch

vox = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
g = int(sys.argv[2]) if len(sys.argv) > 2 else 512
FP32 = len(sys.argv) > 3 and sys.argv[3] == "f32"          # the fp32 tile kernel (k_conv_ts) instead
coords, feats, size, bs, _ = make_batch(1, (g, g, g // 2), vox, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata
sz = tuple(int(s) for s in size)
chans = [32, 64, 128, 256, 512]
for level in range(5 if vox > 300000 else 4):
    C = chans[level]
    rb = md.subm_rulebook(sz, 3)
    n, P, t = rb.n, rb.rules.total, rb.tiles
    X = torch.randn(n, C, device="cuda"); W = torch.randn(27, C, C, device="cuda") * 0.05
    if not FP32:
        X = X.bfloat16()
        img = F.pack_weights_bf16(W, C, C, 27, 0)
    row = [f"level {level} N={n} P={P} C={C}"]
    for fold in (0, 1024, 16384):
        tt = types.SimpleNamespace(tstab=t.tstab, tile_mask=t.tile_mask, perm=t.perm, tile_order=t.tile_order, n_off=t.n_off)
        if fold:
            ts = t.tstab.clone(); ts[ts >= 0] %= fold; tt.tstab = ts
        fn = (lambda: F.conv_rules(X, tt, n, W, None, C, 0)) if FP32 else (lambda: F.conv_rules_bf16(X, tt, n, W, None, C, 0, image=img))
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 50
        row.append(f"fold {fold:6d}: {us:7.1f} us ({2.0 * P * C * C / us / 1e6:6.1f} TF)")
    print("  ".join(row), flush=True)
    md.strided_rulebook(sz); sz = tuple(s // 2 for s in sz)
