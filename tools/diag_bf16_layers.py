"""Developer tool (GPU box): where does the bf16-storage backbone leave the oracle evaluated with the same roundings?
Relative L2 of the encoder outputs level by level (Backbone.unet.interims vs the oracle's skips) on a 40k-voxel scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import scn_oracle as O
from sparse_rcnn_amd.synthetic import make_batch
from sparse_rcnn_amd.unet import Backbone

coords, feats, size, bs, _ = make_batch(1, (256, 256, 128), 40000, dup=1.15, seed=1)
ch = (32, 64, 128, 256)
params = O.init_unet_params(7, list(ch), seed=0)
scene = O.OracleScene(coords.numpy())
only27 = lambda w: O.bf16_storage(w) if w.shape[0] == 27 else w
only8 = lambda w: O.bf16_storage(w) if w.shape[0] == 8 else w
for mode, kw in (("mirrored", dict(storage=O.bf16_storage, tile_weights=O.bf16_storage)),
                 ("round SubM3 weights only", dict(storage=O.bf16_storage, tile_weights=only27)),
                 ("round strided weights only", dict(storage=O.bf16_storage, tile_weights=only8)),
                 ("storage only", dict(storage=O.bf16_storage)), ("fp32 oracle", dict())):
    rec = []
    exp = O.unet_forward(scene, feats, params, list(ch), record=rec, **kw)
    net = Backbone(7, ch, bf16_blocks="all").cuda()
    net.unet.load_oracle_params(params)
    out = net(coords, feats.cuda(), size, 1)
    got = [("enc%d" % l, t.features.detach().float().cpu()) for l, t in enumerate(net.unet.interims)]
    rl2 = lambda a, b: ((a - b).norm() / b.norm()).item()
    print(mode, " ".join(f"{n}:{rl2(g, dict(rec)[n]):.2e}" for n, g in got), f"final:{rl2(out.features.detach().cpu(), exp):.2e}")

# ---- level 1 piece by piece, each HIP module fed the ORACLE's own (mirrored) input ------------------------------------
rec = []
O.unet_forward(scene, feats, params, list(ch), record=rec, storage=O.bf16_storage, tile_weights=O.bf16_storage)
rec = dict(rec)
net = Backbone(7, ch, bf16_blocks="all").cuda()
net.unet.load_oracle_params(params)
out = net(coords, feats.cuda(), size, 1)
md = out.metadata
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import functional as F
lv_size = [torch.as_tensor([int(s) >> l for s in size]) for l in range(4)]
rl2 = lambda a, b: ((a - b).norm() / b.norm()).item()
for l in (1, 2, 3):
    t_in = scn.SparseConvNetTensor(features=rec[f"enc{l-1}"].to(torch.bfloat16).cuda(), metadata=md, spatial_size=lv_size[l - 1])
    with F.packed_weights(net.unet._pack_jobs()):
        h = net.unet.encoder[l][0](t_in)
        u_in = scn.SparseConvNetTensor(features=rec[f"enc{l}.head"].to(torch.bfloat16).cuda(), metadata=md, spatial_size=lv_size[l])
        u = net.unet.encoder[l][1](u_in)
    print(f"level {l}: HIP strided conv on the oracle's input vs oracle {rl2(h.features.float().cpu(), rec[f'enc{l}.head']):.2e};"
          f"  HIP residual units on the oracle's input vs oracle {rl2(u.features.float().cpu(), rec[f'enc{l}']):.2e}")

# ---- single layer: how often does the bf16 result round the other way than the oracle's fp32 / fp64 accumulation? -------
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import functional as F
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
sz = tuple(int(s) for s in size)
rb = x.metadata.subm_rulebook(sz, 3)
rules = scene.subm_rules(0, 3)
n = rb.n
q = lambda t: t.to(torch.bfloat16).float()
for C in (32, 64, 128, 256):
    g = torch.Generator().manual_seed(C)
    X = q(torch.randn(n, C, generator=g)); W = q(torch.randn(27, C, C, generator=g) * (2.0 / (27 * C)) ** 0.5)
    y = F.conv_rules_bf16(X.to(torch.bfloat16).cuda(), rb.tiles, n, W.cuda(), None, C).float().cpu()
    y32 = O.conv_fwd(X, rules, W, None, n)
    y64 = O.conv_fwd(X.double(), rules, W.double(), None, n)
    for nm, ref in (("fp32-accumulated oracle", q(y32)), ("fp64-accumulated oracle", q(y64.float()))):
        diff = (y != ref)
        print(f"C={C:3d} vs {nm}: {diff.float().mean().item() * 100:.3f} % of the bf16 outputs differ, rel L2 "
              f"{((y - ref).norm() / ref.norm()).item():.2e}")
    print(f"      oracle32 vs oracle64 after rounding: {(q(y32) != q(y64.float())).float().mean().item() * 100:.3f} % differ")
