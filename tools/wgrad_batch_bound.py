"""Developer tool (GPU box): upper bound for batching the two weight gradients of a residual unit into one launch -- the rule
list of every offset repeated twice (one launch, twice the units of the same size) against two launches of the plain list."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.synthetic import make_batch
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata; sz = tuple(int(s) for s in size); lib = L.lib()
def timeit(run, n=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / n
for level, C in enumerate([32, 64, 128, 256]):
    rb = md.subm_rulebook(sz, 3); n, r = rb.n, rb.rules
    X = torch.randn(n, C, device="cuda"); dY = torch.randn(n, C, device="cuda")
    dW = torch.empty(27, C, C, device="cuda")
    ph = [int(r.prefix_host[o]) for o in range(28)]
    ir, orr = r.in_rows, r.out_rows
    ir2 = torch.cat([torch.cat([ir[ph[o]:ph[o + 1]]] * 2) for o in range(27)]); or2 = torch.cat([torch.cat([orr[ph[o]:ph[o + 1]]] * 2) for o in range(27)])
    ph2 = L.host_i64(28)
    for o in range(28): ph2[o] = 2 * ph[o]
    def mk(irx, orx, phx):
        scratch = torch.empty(lib.scn_wgrad_scratch_bytes(C, C, phx, 27), dtype=torch.uint8, device="cuda")
        return lambda: L.check(lib.scn_wgrad_rules(L.ptr(X), C, L.ptr(dY), C, L.ptr(irx), L.ptr(orx), phx, 27, L.ptr(dW), L.ptr(scratch), 1, L.stream()))
    one, two = mk(ir, orr, r.prefix_host), mk(ir2, or2, ph2)
    t1, t2 = timeit(one), timeit(two)
    print(f"L{level} C={C}: one list {t1:6.1f} us; doubled list in one call {t2:6.1f} us = {t2 / (2 * t1):.3f} of two calls", flush=True)
    if level < 3:
        md.strided_rulebook(sz); sz = tuple(s // 2 for s in sz)
