import sys, os
sys.path.insert(0, '/root/repo')
import torch, numpy as np
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.synthetic import make_batch
coords, feats, size, bs, _ = make_batch(1, (32, 32, 32), 3000, seed=1, uniform=False)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
rb = x.metadata.subm_rulebook(size, 3); n, t = rb.n, rb.tiles
C = 32
torch.manual_seed(0)
X = torch.randn(n, C, device="cuda"); W = torch.randn(27, C, C, device="cuda") * 0.05
Y = torch.empty(n, C, device="cuda"); Y2 = torch.empty(n, C, device="cuda")
lib = L.lib()
L.check(lib.scn_conv_tiles(L.ptr(X), C, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), L.ptr(t.tile_order), 27, n, L.ptr(W), 0, 0, 0, L.ptr(Y), C, 0, 0, L.stream()))
L.check(lib.scn_gemm_table(L.ptr(X), n, C, L.ptr(rb.table), 27, n, L.ptr(W), 0, 0, 0, L.ptr(Y2), C, 0, L.stream()))
torch.cuda.synchronize()
d = (Y - Y2).abs().cpu().numpy()
print("n", n, "max", d.max(), "bad rows", (d.max(1) > 1e-3).sum(), "bad cols", np.nonzero(d.max(0) > 1e-3)[0])
bad = np.nonzero(d.max(1) > 1e-3)[0]
perm = t.perm.cpu().numpy()
pos = {r: i for i, r in enumerate(perm)}
print("bad row sorted positions mod 16:", sorted(set(pos[r] % 16 for r in bad[:200])))
print("bad tiles:", sorted(set(pos[r] // 16 for r in bad))[:20], "of", len(perm)//16)
# single-offset test: W only offset 13
for o in (0, 13, 26):
    W1 = torch.zeros_like(W); W1[o] = W[o]
    L.check(lib.scn_conv_tiles(L.ptr(X), C, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), L.ptr(t.tile_order), 27, n, L.ptr(W1), 0, 0, 0, L.ptr(Y), C, 0, 0, L.stream()))
    L.check(lib.scn_gemm_table(L.ptr(X), n, C, L.ptr(rb.table), 27, n, L.ptr(W1), 0, 0, 0, L.ptr(Y2), C, 0, L.stream()))
    torch.cuda.synchronize()
    print("offset", o, "max diff", (Y - Y2).abs().max().item())
