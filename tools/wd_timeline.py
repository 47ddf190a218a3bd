"""Developer tool (GPU box): per-wave timeline of k_wgrad_direct on the SubM rule lists of the cfg-2 scene.  Needs a library
built with -DWD_TIMELINE=1 for scn_wgrad.hip (tools/build_variant.sh tools/ab/libscn_wdtl.so scn_wgrad.hip "-DWD_TIMELINE=1"),
selected with SCN_MI355X_LIB."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.synthetic import make_batch
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata; sz = tuple(int(s) for s in size)
lib = L.lib()
raw = ctypes.CDLL(L.LIB_PATH)
buf = np.zeros(65536 * 4, np.int64)
for level, C in enumerate([32, 64, 128, 256]):
    rb = md.subm_rulebook(sz, 3); n, r = rb.n, rb.rules
    X = torch.randn(n, C, device="cuda"); dY = torch.randn(n, C, device="cuda")
    dW = torch.empty(27, C, C, device="cuda")
    scratch = torch.empty(lib.scn_wgrad_scratch_bytes(C, C, r.prefix_host, 27), dtype=torch.uint8, device="cuda")
    for relu in (0,):
        for _ in range(200):
            L.check(lib.scn_wgrad_rules(L.ptr(X), C, L.ptr(dY), C, L.ptr(r.in_rows), L.ptr(r.out_rows), r.prefix_host, 27,
                                        L.ptr(dW), L.ptr(scratch), relu, L.stream()))
        torch.cuda.synchronize()
        assert raw.scn_debug_wd_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        d = buf.reshape(-1, 4); d = d[d[:, 2] > 0]
        # keep the stamps of the LAST launch only: its t0 values cluster at the maximum
        t0max = d[:, 0].max(); d = d[d[:, 0] > t0max - 20000]          # 200 us window (100 MHz clock)
        t0 = d[:, 0].min(); us = lambda v: v / 100.0
        span = d[:, 2].max() - t0
        life = (d[:, 2] - d[:, 0])
        print(f"L{level} C={C} P={r.total}: waves {len(d)} span {us(span):.1f} us; start first..last {us(d[:,0].min()-t0):.1f}..{us(d[:,0].max()-t0):.1f}; "
              f"wave life mean {us(life.mean()):.1f} min {us(life.min()):.1f} max {us(life.max()):.1f}; loop mean {us((d[:,1]-d[:,0]).mean()):.1f}; "
              f"epilogue mean {us((d[:,2]-d[:,1]).mean()):.1f}; rules/wave mean {d[:,3].mean():.0f} max {d[:,3].max()}")
        tot = span * len(d)
        print(f"   wave-time split: before start {(d[:,0]-t0).sum()/tot:.3f} loop {(d[:,1]-d[:,0]).sum()/tot:.3f} epilogue {(d[:,2]-d[:,1]).sum()/tot:.3f} idle after end {(d[:,2].max()-d[:,2]).sum()/tot:.3f}")
        # rate vs start time
        late = d[:, 0] - t0 > 0.2 * span
        print(f"   waves starting after 20 % of the span: {late.mean():.2f}; us per 64 rules: early {us((life[~late] / np.maximum(1, d[~late, 3]) * 64).mean()):.2f}" +
              (f" late {us((life[late] / np.maximum(1, d[late, 3]) * 64).mean()):.2f}" if late.any() else ""))
    if level < 3:
        md.strided_rulebook(sz); sz = tuple(s // 2 for s in sz)
