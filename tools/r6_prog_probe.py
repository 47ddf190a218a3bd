"""Round 6 (GPU box), VERDICT r5 item 1a: progressive staging of the forward weight slice of k_conv_ts (SCN_TS_PROG = offsets
staged before the workgroup barrier; the rest lands by LDS-DMA while the first tile runs) against the classic staging --
bit-equality and time per level of the cfg-2 scene.
    python tools/r6_prog_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.synthetic import make_batch

coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata
md.build_pyramid(size, 4, 3)
lib = L.lib()


def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def setp(v):
    L.check(lib.scn_debug_set(b"SCN_TS_PROG", str(v).encode() if v else None))


sz = tuple(int(s) for s in size)
for level, Cc in enumerate([32, 64, 128, 256]):
    rb = md.subm_rulebook(sz, 3)
    n, P, t = rb.n, rb.rules.total, rb.tiles
    g = torch.Generator(device="cuda").manual_seed(level)
    X = torch.randn(n, Cc, device="cuda", generator=g)
    W = torch.randn(27, Cc, Cc, device="cuda", generator=g) * (0.3 / Cc ** 0.5)
    B = torch.randn(Cc, device="cuda", generator=g) * 0.1
    R = torch.randn(n, Cc, device="cuda", generator=g)
    SCR = torch.empty(max(1, lib.scn_conv_tiles_scratch_bytes(Cc, n, Cc)), dtype=torch.uint8, device="cuda")
    ARR = torch.zeros(max(1, lib.scn_conv_tiles_arrival_counters(Cc, n, Cc)), dtype=torch.int32, device="cuda")
    arr = L.ptr(ARR) if Cc > 32 else 0

    def run(Y, fl=1, res=True):
        L.check(lib.scn_conv_tiles(L.ptr(X), n, Cc, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), L.ptr(t.tile_order), 27, n,
                                   L.ptr(W), L.ptr(B), L.ptr(R) if res else 0, 0, L.ptr(Y), Cc, fl, L.ptr(SCR), arr, L.stream()))
    Y0, Y1 = torch.empty(n, Cc, device="cuda"), torch.empty(n, Cc, device="cuda")
    setp(0); run(Y0); torch.cuda.synchronize()
    line = f"L{level} C={Cc:3d} n={n:6d}  classic {timed(lambda: run(Y0)):7.1f} us"
    for g0 in (4, 8, 12):
        setp(g0)
        Y1.fill_(float("nan")); run(Y1); torch.cuda.synchronize()
        same = bool(torch.equal(Y0, Y1))
        for _ in range(25): run(Y1)
        torch.cuda.synchronize()
        same2 = bool(torch.equal(Y0, Y1))
        line += f"   g0={g0:2d} {timed(lambda: run(Y1)):7.1f} us bit-equal {same}/{same2}"
    setp(0)
    line += f"   classic again {timed(lambda: run(Y0)):7.1f}"
    print(line, flush=True)
    sz = tuple(s // 2 for s in sz)
