"""Round 6 (GPU box): the chained launch of a level's four SubM convolutions (scn_conv_tiles_chain) against four plain
scn_conv_tiles launches -- bit-equality and time per level of the cfg-2 scene.
    python tools/r6_chain_probe.py [bwd]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.synthetic import make_batch

bwd = len(sys.argv) > 1 and sys.argv[1] == "bwd"
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata
md.build_pyramid(size, 4, 3)
sz = tuple(int(s) for s in size)
lib = L.lib()


class Role(C.Structure):
    _fields_ = [("X", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p), ("relu_mask", C.c_void_p),
                ("Y", C.c_void_p), ("flags", C.c_int32), ("reserved", C.c_int32)]


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


for level, Cc in enumerate([32, 64, 128, 256]):
    rb = md.subm_rulebook(sz, 3)
    n, P, t = rb.n, rb.rules.total, rb.tiles
    g = torch.Generator(device="cuda").manual_seed(level)
    X = torch.randn(n, Cc, device="cuda", generator=g)
    Ws = [torch.randn(27, Cc, Cc, device="cuda", generator=g) * (0.3 / Cc ** 0.5) for _ in range(4)]
    Bs = [torch.randn(Cc, device="cuda", generator=g) * 0.1 for _ in range(4)]
    M = [(torch.rand(n, Cc, device="cuda", generator=g) > 0.5).float() for _ in range(4)]
    SCR = torch.empty(max(1, lib.scn_conv_tiles_scratch_bytes(Cc, n, Cc)), dtype=torch.uint8, device="cuda")
    ARR = torch.zeros(max(1, lib.scn_conv_tiles_arrival_counters(Cc, n, Cc)), dtype=torch.int32, device="cuda")

    def bufs():
        return [torch.empty(n, Cc, device="cuda") for _ in range(4)]
    if not bwd:      # two residual units, forward: y1 = conv(relu x); y = x + conv(relu y1); ...
        fl = 1
        def plan(Y):
            return [(X, Ws[0], Bs[0], None, None, Y[0], fl), (Y[0], Ws[1], Bs[1], X, None, Y[1], fl),
                    (Y[1], Ws[2], Bs[2], None, None, Y[2], fl), (Y[2], Ws[3], Bs[3], Y[1], None, Y[3], fl)]
    else:            # backward-data of two units: dy1 = mask . conv^T(g); dx = g + mask . conv^T(dy1) (residual last)
        BACK = 2 | 4
        def plan(Y):
            return [(X, Ws[0], None, None, M[0], Y[0], BACK), (Y[0], Ws[1], None, X, M[1], Y[1], BACK | 8),
                    (Y[1], Ws[2], None, None, M[2], Y[2], BACK), (Y[2], Ws[3], None, Y[1], M[3], Y[3], BACK | 8)]
    arr = L.ptr(ARR) if Cc > 32 else 0
    Ya, Yb = bufs(), bufs()
    pa, pb = plan(Ya), plan(Yb)

    def run_plain():
        for (xi, w, b, r, m, y, f) in pa:
            L.check(lib.scn_conv_tiles(L.ptr(xi), n, Cc, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), L.ptr(t.tile_order), 27, n,
                                       L.ptr(w), L.ptr(b) if b is not None else 0, L.ptr(r) if r is not None else 0,
                                       L.ptr(m) if m is not None else 0, L.ptr(y), Cc, f, L.ptr(SCR), arr, L.stream()))
    roles = (Role * 4)(*[Role(L.ptr(xi), L.ptr(w), L.ptr(b) if b is not None else None, L.ptr(r) if r is not None else None,
                              L.ptr(m) if m is not None else None, L.ptr(y), f, 0) for (xi, w, b, r, m, y, f) in pb])

    def run_chain(k=4):
        L.check(lib.scn_conv_tiles_chain(k, roles, n, Cc, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), L.ptr(t.tile_order), 27, n,
                                         Cc, L.ptr(SCR), arr, L.stream()))
    run_plain(); run_chain(); torch.cuda.synchronize()
    same = [bool(torch.equal(a, b)) for a, b in zip(Ya, Yb)]
    cnt = (C.c_int64 * 2)(); lib.scn_conv_tiles_chain_counts(cnt, 1)
    tp, tc = timed(run_plain), timed(run_chain)
    extra = ""
    for e in (os.environ.get("EXPS", "").split(",") if os.environ.get("EXPS") else []):
        L.check(lib.scn_debug_set(b"SCN_EXP_A", e.encode()))
        extra += f"  exp{e} {timed(run_chain):7.1f}"
    L.check(lib.scn_debug_set(b"SCN_EXP_A", None))
    run_chain(); torch.cuda.synchronize()
    # repeated launches: the words must be back at zero each time; results stay equal
    for _ in range(5): run_chain()
    torch.cuda.synchronize()
    same2 = [bool(torch.equal(a, b)) for a, b in zip(Ya, Yb)]
    print(f"L{level} C={Cc:3d} n={n:6d}  {'bwd' if bwd else 'fwd'}: 4 plain launches {tp:7.1f} us   chained {tc:7.1f} us   "
          f"saves {(tp - tc) / 3:5.1f} us per link   bit-equal {same} / after 25 more {same2}   chained launches {cnt[0]} roles {cnt[1]}{extra}",
          flush=True)
    sz = tuple(s // 2 for s in sz)
