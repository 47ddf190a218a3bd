"""Developer tool (GPU box): time the 1x1 (scn_gemm_table without a table) and strided rule-list (scn_gemm_rules) GEMMs
at the shapes of the cfg-2 U-Net.    python tools/ablate_gemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.synthetic import make_batch

CH = [32, 64, 128, 256]
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata
md.build_pyramid(size, 4, 3)
lib = L.lib()


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


total = total_v1 = total_join = 0.0
V1 = L.F_GEMM_V1
sz = tuple(int(s) for s in size)
for l in range(3):
    c, cu = CH[l], CH[l + 1]
    sb = md.strided_rulebook(sz)
    n, nc = sb.n_fine, sb.n_coarse
    A = torch.randn(n, 2 * c, device="cuda"); Wn = torch.randn(1, 2 * c, c, device="cuda") * 0.05
    Y = torch.empty(n, c, device="cuda"); G = torch.randn(n, c, device="cuda"); dA = torch.empty(n, 2 * c, device="cuda")
    f1 = lambda: L.check(lib.scn_gemm_table(L.ptr(A), n, 2 * c, 0, 1, n, L.ptr(Wn), 0, 0, 0, L.ptr(Y), c, 0, L.stream()))
    f2 = lambda: L.check(lib.scn_gemm_table(L.ptr(G), n, c, 0, 1, n, L.ptr(Wn), 0, 0, 0, L.ptr(dA), 2 * c, L.F_W_TRANSPOSED, L.stream()))
    Xc = torch.randn(nc, cu, device="cuda"); Wu = torch.randn(8, cu, c, device="cuda") * 0.05; Yf = torch.empty(n, c, device="cuda")
    r = sb.rules
    # Deconvolution fwd: coarse rows in, fine rows out (roles of the encoder rulebook swapped)
    f3 = lambda: L.check(lib.scn_gemm_rules(L.ptr(Xc), cu, L.ptr(r.out_rows), L.ptr(r.in_rows), r.prefix_host, 8, L.ptr(Wu), 0, 0, L.ptr(Yf), c, 0, L.stream()))
    # Convolution backward-data: dY coarse -> dX fine with W^T
    Wd = torch.randn(8, c, cu, device="cuda") * 0.05
    f4 = lambda: L.check(lib.scn_gemm_rules(L.ptr(Xc), cu, L.ptr(r.out_rows), L.ptr(r.in_rows), r.prefix_host, 8, L.ptr(Wd), 0, 0, L.ptr(Yf), c, L.F_W_TRANSPOSED, L.stream()))
    # the same four through the register-only kernels (SCN_F_GEMM_V1), and the NiN over the two JoinTable parts in one launch
    v1 = lambda: L.check(lib.scn_gemm_table(L.ptr(A), n, 2 * c, 0, 1, n, L.ptr(Wn), 0, 0, 0, L.ptr(Y), c, V1, L.stream()))
    v2 = lambda: L.check(lib.scn_gemm_table(L.ptr(G), n, c, 0, 1, n, L.ptr(Wn), 0, 0, 0, L.ptr(dA), 2 * c, L.F_W_TRANSPOSED | V1, L.stream()))
    v3 = lambda: L.check(lib.scn_gemm_rules(L.ptr(Xc), cu, L.ptr(r.out_rows), L.ptr(r.in_rows), r.prefix_host, 8, L.ptr(Wu), 0, 0, L.ptr(Yf), c, V1, L.stream()))
    v4 = lambda: L.check(lib.scn_gemm_rules(L.ptr(Xc), cu, L.ptr(r.out_rows), L.ptr(r.in_rows), r.prefix_host, 8, L.ptr(Wd), 0, 0, L.ptr(Yf), c, L.F_W_TRANSPOSED | V1, L.stream()))
    A0, A1 = torch.randn(n, c, device="cuda"), torch.randn(n, c, device="cuda")
    d0, d1 = torch.empty(n, c, device="cuda"), torch.empty(n, c, device="cuda")
    j1 = lambda: L.check(lib.scn_gemm_rows2(L.ptr(A0), c, L.ptr(A1), c, n, L.ptr(Wn), 0, 0, 0, L.ptr(Y), c, 0, 0, 0, 0, L.stream()))
    j2 = lambda: L.check(lib.scn_gemm_rows2(L.ptr(G), c, 0, 0, n, L.ptr(Wn), 0, 0, 0, L.ptr(d0), c, L.ptr(d1), c, L.F_W_TRANSPOSED, 0, L.stream()))
    # round 2's per-part form: two launches, the second reads the first's output as residual
    Wh = Wn[0, :c].contiguous()
    p1 = lambda: (L.check(lib.scn_gemm_table(L.ptr(A0), n, c, 0, 1, n, L.ptr(Wh), 0, 0, 0, L.ptr(Y), c, V1, L.stream())),
                  L.check(lib.scn_gemm_table(L.ptr(A1), n, c, 0, 1, n, L.ptr(Wh), 0, L.ptr(Y), 0, L.ptr(Y), c, V1, L.stream())))
    for name, fn in (("NiN fwd", v1), ("NiN bwd-data", v2), ("deconv fwd", v3), ("conv bwd-data", v4)):
        us = timed(fn); total_v1 += us
        print(f"L{l} {name:14s} n={n:6d} {us:7.1f} us  register-only kernel (SCN_F_GEMM_V1)")
    for name, fn in (("NiN fwd 2 src", j1), ("NiN bwd 2 dst", j2)):
        us = timed(fn); total_join += us
        print(f"L{l} {name:14s} n={n:6d} {us:7.1f} us  {2.0 * n * 2 * c * c / us / 1e6:6.1f} TF  {4.0 * n * 3 * c / us / 1e3:6.0f} GB/s (compulsory)")
    print(f"L{l} NiN fwd, one register-kernel launch per part (round 2a): {timed(p1):7.1f} us")
    for name, fn, flop, byts in (("NiN fwd", f1, 2.0 * n * 2 * c * c, 4.0 * n * 3 * c), ("NiN bwd-data", f2, 2.0 * n * 2 * c * c, 4.0 * n * 3 * c),
                                 ("deconv fwd", f3, 2.0 * n * cu * c, 4.0 * (nc * cu + n * c)), ("conv bwd-data", f4, 2.0 * n * cu * c, 4.0 * (nc * cu + n * c))):
        us = timed(fn); total += us
        print(f"L{l} {name:14s} n={n:6d} {us:7.1f} us  {flop / us / 1e6:6.1f} TF  {byts / us / 1e3:6.0f} GB/s (compulsory)")
    sz = tuple(s // 2 for s in sz)
print(f"total {total:.1f} us (LDS-tiled kernels; the same twelve launches through the register-only kernels: {total_v1:.1f} us; "
      f"the six NiN launches over two sources / destinations: {total_join:.1f} us)")
