"""Developer tool (GPU box): how much of scn_wgrad_rules' time is memory latency?  Same rule counts, but all rules
folded into the first M rows (M small => every gather hits in L2)."""
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

for level, C in enumerate([32, 64]):
    rb = md.subm_rulebook(sz, 3); n, r = rb.n, rb.rules
    X = torch.randn(n, C, device="cuda"); dY = torch.randn(n, C, device="cuda")
    dW = torch.empty(27, C, C, device="cuda")
    scratch = torch.empty(lib.scn_wgrad_scratch_bytes(C, C, r.prefix_host, 27), dtype=torch.uint8, device="cuda")
    for M in (0, 65536, 16384, 4096, 1024):
        ir = r.in_rows if M == 0 else (r.in_rows % M).contiguous()
        orr = r.out_rows if M == 0 else (r.out_rows % M).contiguous()
        def run():
            L.check(lib.scn_wgrad_rules(L.ptr(X), C, L.ptr(dY), C, L.ptr(ir), L.ptr(orr), r.prefix_host, 27,
                                        L.ptr(dW), L.ptr(scratch), 0, L.stream()))
        t = timeit(run)
        print(f"L{level} C={C} fold={M:6d}: {t:7.1f} us {2.0*r.total*C*C/t/1e6:6.1f} TF", flush=True)
    md.strided_rulebook(sz); sz = tuple(s // 2 for s in sz)
