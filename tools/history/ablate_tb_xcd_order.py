"""Developer tool (GPU box): would an XCD-local tile hand-out help the bf16 tile convolution at levels 0-1?  Host-side
experiment, no kernel change: the tile list is re-dealt so that the workgroups of XCD x (blockIdx % 8 == x, one slice per
launch at these levels) own the tiles of spatial bin x (bins by the mean row number of a tile's 16 rows -- rows are numbered
in mesh order), most expensive first inside a workgroup's list.  Same tiles, same results, different hand-out.
    python tools/ablate_tb_xcd_order.py [voxels=600000] [grid=1024] [f32]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import functional as F
from sparse_rcnn_amd.synthetic import make_batch

vox = int(sys.argv[1]) if len(sys.argv) > 1 else 600000
g = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
FP32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
coords, feats, size, bs, _ = make_batch(1, (g, g, g // 2), vox, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata
sz = tuple(int(s) for s in size)

def redeal(t, n, n_slices, bins=8, wg_per_cu=1):
    """tile_order' with list position tg + k * n_tg = k-th tile of workgroup tg; tg's XCD = (tg * n_slices) % 8 for one
    slice per tile group ... only exact for n_slices == 1 (level 0) and approximately otherwise."""
    nt = (n + 15) // 16
    order = t.tile_order.cpu().numpy().astype(np.int64)[:nt]
    perm = t.perm.cpu().numpy()[:nt * 16].reshape(nt, 16).astype(np.float64)
    perm[perm < 0] = np.nan
    pos = np.nanmean(perm, axis=1)                                   # spatial key of a tile
    cost = np.array([bin(int(v)).count("1") for v in t.tile_mask.cpu().numpy().view(np.uint32)[:nt]])
    n_tg = max(1, min(256 * wg_per_cu // n_slices, (nt + 15) // 16))
    n_tiles = [(nt - tg + n_tg - 1) // n_tg for tg in range(n_tg)]
    wg_of_bin = [[tg for tg in range(n_tg) if (tg * n_slices) % bins == b] for b in range(bins)]
    by_pos = np.argsort(pos, kind="stable")
    out = np.empty(nt, dtype=np.int32)
    cur = 0
    for b in range(bins):
        wgs = wg_of_bin[b]
        if not wgs: continue
        cnt = sum(n_tiles[tg] for tg in wgs)
        mine = by_pos[cur:cur + cnt]; cur += cnt
        mine = mine[np.argsort(-cost[mine], kind="stable")]          # LPT inside the bin
        k = [0] * len(wgs)
        j = 0
        for tile in mine:                                            # deal round-robin to the bin's workgroups
            while k[j % len(wgs)] >= n_tiles[wgs[j % len(wgs)]]: j += 1
            w = j % len(wgs)
            out[wgs[w] + k[w] * n_tg] = tile
            k[w] += 1; j += 1
    assert cur == nt and sorted(out.tolist()) == list(range(nt))
    return torch.from_numpy(out).to(t.tile_order.device)

chans = [32, 64, 128, 256]
for level in range(3):
    C = chans[level]
    rb = md.subm_rulebook(sz, 3)
    n, P, t = rb.n, rb.rules.total, rb.tiles
    X = torch.randn(n, C, device="cuda"); W = torch.randn(27, C, C, device="cuda") * 0.05
    img = None
    if not FP32:
        X = X.bfloat16(); img = F.pack_weights_bf16(W, C, C, 27, 0)
    n_slices = (1 if C <= 64 else C // 64) * (C // 32) if not FP32 else (C // 32) * (C // 32)
    if not FP32 and C == 32: n_slices = 1
    row = [f"level {level} N={n} C={C}"]
    ref = None
    for name in ("LPT order", "XCD bins"):
        tt = types.SimpleNamespace(tstab=t.tstab, tile_mask=t.tile_mask, perm=t.perm, tile_order=t.tile_order, n_off=t.n_off)
        if name == "XCD bins":
            tt.tile_order = redeal(t, n, n_slices, wg_per_cu=2 if (not FP32 and C == 32) else 1)
        fn = (lambda: F.conv_rules(X, tt, n, W, None, C, 0)) if FP32 else (lambda: F.conv_rules_bf16(X, tt, n, W, None, C, 0, image=img))
        y = fn()
        if ref is None: ref = y
        else: assert torch.equal(ref, y)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        row.append(f"{name}: {s.elapsed_time(e) * 50:7.1f} us")
    print("  ".join(row), flush=True)
    md.strided_rulebook(sz); sz = tuple(s // 2 for s in sz)
