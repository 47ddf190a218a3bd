"""Round 6 (GPU box), VERDICT r5 item 2: tiles whose rows come from ONE region of the scene -- the sort key's top bits are the
row bin (scn_tiles_build_x, bits 8-10 of with_x) -- against the plain mask sort, bf16 (and fp32) tile kernels per level:
executed / useful steps and time per launch; outputs must be identical (row order is internal to the tile tables).
    python tools/r6_bin_tiles_probe.py [voxels=600000] [grid=1024] [f32]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import functional as F, metadata as MD, _lib as L
from sparse_rcnn_amd.synthetic import make_batch

vox = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
g = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
FP32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
coords, feats, size, bs, _ = make_batch(1, (g, g, g // 2), vox, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata
md.xcd_order = True
sz = tuple(int(s) for s in size)
chans = [32, 64, 128, 256, 512]
for level in range(4 if vox > 300000 else 3):
    C = chans[level]
    rb = md.subm_rulebook(sz, 3)
    n, P = rb.n, rb.rules.total
    X = torch.randn(n, C, device="cuda"); W = torch.randn(27, C, C, device="cuda") * 0.05
    img = None
    if not FP32:
        X = X.bfloat16(); img = F.pack_weights_bf16(W, C, C, 27, 0)
    row = [f"level {level} N={n} C={C}"]
    ref = None
    for lb in (0, 2, 3, 4, 5):
        t = MD.build_tiles(rb.table, 27, n, with_x=True, log2_bins=lb)
        tm = t.tile_mask.cpu().numpy().view(np.uint32)
        execd = int(sum(bin(int(v)).count("1") for v in tm)) * 16
        for xorder in ((True, False) if not FP32 else (False,)):
            tt = types.SimpleNamespace(tstab=t.tstab, tile_mask=t.tile_mask, perm=t.perm, tile_order=t.tile_order, n_off=t.n_off,
                                       has_x=xorder, n=n)
            if FP32:
                fn = lambda: F.conv_rules(X, tt, n, W, None, C, 0)
            else:
                fn = lambda: F.conv_rules_bf16(X, tt, n, W, None, C, 0, image=img)
            y = fn()
            if ref is None: ref = y
            else: assert torch.equal(ref, y), "row order inside the tile tables must not change a result"
            for _ in range(3): fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): fn()
            e.record(); torch.cuda.synchronize()
            row.append(f"bins {1 << lb:2d}{' x' if xorder else '  '}: {execd / P:.3f} {s.elapsed_time(e) * 50:6.1f} us")
    print("  |  ".join(row), flush=True)
    md.strided_rulebook(sz); sz = tuple(s // 2 for s in sz)
