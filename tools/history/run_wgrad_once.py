"""GPU-box helper for counter collection: runs scn_wgrad_rules a few times on one level."""
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
X = torch.randn(n, C, device="cuda"); dY = torch.randn(n, C, device="cuda"); dW = torch.empty(27, C, C, device="cuda")
lib = L.lib()
scratch = torch.empty(lib.scn_wgrad_scratch_bytes(C, C, r.prefix_host, 27), dtype=torch.uint8, device="cuda")
for _ in range(5):
    L.check(lib.scn_wgrad_rules(L.ptr(X), C, L.ptr(dY), C, L.ptr(r.in_rows), L.ptr(r.out_rows), r.prefix_host, 27,
                                L.ptr(dW), L.ptr(scratch), 0, L.stream()))
torch.cuda.synchronize()
