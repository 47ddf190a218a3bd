"""Developer tool (GPU box): scn_conv_tiles on a DENSE cube (every interior voxel has all 27 neighbours): the kernel's
ceiling without tile-cost imbalance, short tiles or wasted rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L
for C, dims in ((32, (64, 64, 36)), (64, (40, 40, 32)), (128, (24, 24, 22)), (256, (16, 16, 12))):
    g = np.stack(np.meshgrid(*[np.arange(d) for d in dims], indexing="ij"), -1).reshape(-1, 3) + 8
    coords = torch.from_numpy(np.concatenate([g, np.zeros((len(g), 1), np.int64)], 1))
    feats = torch.randn(len(g), 7)
    size = torch.tensor([128, 128, 64])
    x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
    md = x.metadata
    rb = md.subm_rulebook(tuple(int(s) for s in size), 3)
    n, P, t = rb.n, rb.rules.total, rb.tiles
    X = torch.randn(n, C, device="cuda"); W = torch.randn(27, C, C, device="cuda") * 0.05
    Y = torch.empty(n, C, device="cuda")
    lib = L.lib()
    SCR = torch.empty(max(1, lib.scn_conv_tiles_scratch_bytes(C, n, C)), dtype=torch.uint8, device="cuda")
    tm = t.tile_mask.cpu().numpy().view("uint32")
    execd = sum(bin(int(v)).count("1") for v in tm) * 16
    def run():
        L.check(lib.scn_conv_tiles(L.ptr(X), n, C, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), L.ptr(t.tile_order), 27, n, L.ptr(W), 0, 0, 0,
                                   L.ptr(Y), C, 0, L.ptr(SCR), 0, L.stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): run()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    print(f"dense C={C} N={n} P={P} tiles={len(tm)} exec/useful={execd/P:.3f}: {us:7.1f} us  {2.0*P*C*C/us/1e6:6.1f} TF useful {2.0*execd*C*C/us/1e6:6.1f} TF executed", flush=True)
