"""Developer tool (GPU box): k_conv_ts on channel counts that are not multiples of 32 (the reference's own plan
32-48-64-80-96-112, scannet_config/run.py:539-549) -- the TAIL variants (dead K halves / column blocks skipped, slices
weighted) against the padded kernel (SCN_TS_NO_TAIL=1), useful TFLOP/s next to the neighbouring multiples of 32.
    python tools/ablate_conv_tail.py [level ...]          SCN_TS_W_HALF / SCN_TS_W_BOTH: slice weights (read at first use)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd._lib import switches as _SW      # library switches: scn_debug_set (the environment is read once at load)
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.synthetic import make_batch

levels = [int(a) for a in sys.argv[1:]] or [0, 1, 2]
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata
md.build_pyramid(size, max(levels) + 1, 3)
lib = L.lib()


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


sz = tuple(int(s) for s in size)
for level in range(max(levels) + 1):
    rb = md.subm_rulebook(sz, 3)
    if level in levels:
        n, P, t = rb.n, rb.rules.total, rb.tiles
        print(f"level {level}: N={n} P={P}  (w_half={os.environ.get('SCN_TS_W_HALF', '0.70')} w_both={os.environ.get('SCN_TS_W_BOTH', '0.50')})")
        for C in (32, 48, 64, 80, 96, 112, 128):
            X = torch.randn(n, C, device="cuda"); W = torch.randn(27, C, C, device="cuda") * 0.05
            Y = torch.empty(n, C, device="cuda")
            SCR = torch.empty(max(1, lib.scn_conv_tiles_scratch_bytes(C, n, C)), dtype=torch.uint8, device="cuda")
            ARR = torch.zeros(max(1, lib.scn_conv_tiles_arrival_counters(C, n, C)), dtype=torch.int32, device="cuda")

            def run(fl=1):
                L.check(lib.scn_conv_tiles(L.ptr(X), n, C, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), L.ptr(t.tile_order), 27, n,
                                           L.ptr(W), 0, 0, 0, L.ptr(Y), C, fl, L.ptr(SCR), L.ptr(ARR), L.stream()))
            us = timed(run)
            _SW["SCN_TS_NO_TAIL"] = "1"
            us0 = timed(run)
            del _SW["SCN_TS_NO_TAIL"]
            fl = 2.0 * P * C * C
            print(f"  C={C:3d}  tail {us:6.1f} us {fl / us / 1e6:6.1f} TF useful   padded {us0:6.1f} us {fl / us0 / 1e6:6.1f} TF   x{us0 / us:.2f}")
    if level < max(levels):
        sz = md.strided_rulebook(sz).coarse_size
