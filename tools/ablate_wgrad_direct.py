"""Developer tool (GPU box): scn_wgrad_rules timing per U-Net layer shape; run with and without SCN_WGRAD_DIRECT=1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.synthetic import make_batch

coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata; sz = tuple(int(s) for s in size)
lib = L.lib()

def timeit(run, n=10):
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / n

def case(name, X, dY, ir, orr, ph, n_off, P):
    cin, cout = X.shape[1], dY.shape[1]
    dW = torch.empty(n_off, cin, cout, device="cuda")
    scratch = torch.empty(lib.scn_wgrad_scratch_bytes(cin, cout, ph, n_off), dtype=torch.uint8, device="cuda")
    def run():
        L.check(lib.scn_wgrad_rules(L.ptr(X), cin, L.ptr(dY), cout, L.ptr(ir), L.ptr(orr), ph, n_off,
                                    L.ptr(dW), L.ptr(scratch), 0, L.stream()))
    t = timeit(run)
    # reference: fp64 on the device through index_add
    ref = torch.zeros(n_off, cin, cout, device="cuda", dtype=torch.float64)
    for o in range(n_off):
        a, b = int(ph[o]), int(ph[o + 1])
        if b > a:
            xi = X[ir[a:b].long()] if ir is not None else X[a:b]
            yo = dY[orr[a:b].long()] if orr is not None else dY[a:b]
            ref[o] = xi.double().t() @ yo.double()
    err = (dW.double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"{name:28s} P={P:8d} {cin:3d}->{cout:3d}: {t:7.1f} us {2.0*P*cin*cout/t/1e6:6.1f} TF  relerr {err:.1e}", flush=True)

for level, C in enumerate([32, 64, 128, 256]):
    rb = md.subm_rulebook(sz, 3); n, r = rb.n, rb.rules
    X = torch.randn(n, C, device="cuda"); dY = torch.randn(n, C, device="cuda")
    case(f"L{level} subm", X, dY, r.in_rows, r.out_rows, r.prefix_host, 27, r.total)
    if level == 0:
        X7 = torch.randn(n, 7, device="cuda")
        h = L.host_i64(2); h[0], h[1] = 0, n
        case("L0 input 1x1 (identity)", X7, dY, None, None, h, 1, n)
    if level < 3:
        srb = md.strided_rulebook(sz); sr = srb.rules
        Xc = torch.randn(srb.n_coarse, 2 * C, device="cuda")
        case(f"L{level} conv s2", X, Xc, sr.in_rows, sr.out_rows, sr.prefix_host, 8, sr.total)
        case(f"L{level} deconv s2", Xc, X, sr.out_rows, sr.in_rows, sr.prefix_host, 8, sr.total)
        X2 = torch.randn(n, 2 * C, device="cuda")
        h = L.host_i64(2); h[0], h[1] = 0, n
        case(f"L{level} NiN 2C->C (identity)", X2, dY, None, None, h, 1, n)
        sz = tuple(s // 2 for s in sz)
