"""Developer tool (GPU box): time scn_wgrad_rules on one level of the cfg-2 scene for several block sizes / K-splits.
    python tools/ablate_wgrad.py [level]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.synthetic import make_batch

level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
C = [32, 64, 128, 256][level]
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata; sz = tuple(int(s) for s in size)
for l in range(level):
    md.strided_rulebook(sz); sz = tuple(s // 2 for s in sz)
rb = md.subm_rulebook(sz, 3); n, r = rb.n, rb.rules
X = torch.randn(n, C, device="cuda"); dY = torch.randn(n, C, device="cuda")
dW = torch.empty(27, C, C, device="cuda")
lib = L.lib()
ref = None
print(f"level {level} N={n} P={r.total} C={C}")
cfgs = [(None, None)] + [(cb, sp) for cb in (32, 64, 128) if cb <= max(32, C) for sp in (128, 256, 512, 1024, 2048)]
for cb, sp in cfgs:
    for k, v in (("SCN_WGRAD_CB", cb), ("SCN_WGRAD_SPLITS", sp)):
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    nbytes = lib.scn_wgrad_scratch_bytes(C, C, r.prefix_host, 27)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    def run():
        L.check(lib.scn_wgrad_rules(L.ptr(X), C, L.ptr(dY), C, L.ptr(r.in_rows), L.ptr(r.out_rows), r.prefix_host, 27,
                                    L.ptr(dW), L.ptr(scratch), 0, L.stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    if ref is None: ref = dW.clone()
    err = (dW - ref).abs().max().item() / ref.abs().max().item()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): run()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    print(f"cb={cb} splits={sp}: {us:8.1f} us  {2.0 * r.total * C * C / us / 1e6:6.1f} TF  scratch {nbytes/1e6:6.1f} MB  relerr {err:.1e}")
