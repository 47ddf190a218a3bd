"""GPU-box helper for counter collection: runs conv_tiles a few times on one level (see tools/ablate_conv.py)."""
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
rb = md.subm_rulebook(sz, 3); n, t = rb.n, rb.tiles
X = torch.randn(n, C, device="cuda"); W = torch.randn(27, C, C, device="cuda") * 0.05; Y = torch.empty(n, C, device="cuda")
lib = L.lib()
SCR = torch.empty(max(1, lib.scn_conv_tiles_scratch_bytes(C, n, C)), dtype=torch.uint8, device='cuda')
for _ in range(5):
    L.check(lib.scn_conv_tiles(L.ptr(X), n, C, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), L.ptr(t.tile_order), 27, n, L.ptr(W), 0, 0, 0, L.ptr(Y), C, 0, L.ptr(SCR), 0, L.stream()))
torch.cuda.synchronize()
