"""-m gpu: BASELINE-size (150k and 600k voxels) checks through size-independent properties: structural invariants of the
index structures, agreement of two independent kernels, linearity, adjointness of forward / backward-data / weight
gradient, bitwise reproducibility, the in-launch K reduction against its two-launch form.  (Oracle parity at these sizes:
tests/test_gpu_atsize.py.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(gpu):
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd.synthetic import make_batch
    coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150_000, seed=1)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), 1))
    x.metadata.build_pyramid(size, 4, 3)
    return scn, coords, feats, size, x


def test_index_invariants_full_size(scene):
    scn, coords, feats, size, x = scene
    md = x.metadata
    n0 = x.features.shape[0]
    assert n0 == 150_000
    assert int(md.row_count.sum().item()) == len(coords)                      # every point lands in exactly one voxel
    loc = x.get_spatial_locations().numpy()
    assert len(np.unique(loc, axis=0)) == n0                                   # rows are distinct sites
    first = md.row_first.cpu().numpy()
    assert (np.diff(first) > 0).all()                                          # first-occurrence numbering
    sz = tuple(int(s) for s in size)
    for level in range(4):
        rb = md.subm_rulebook(sz, 3)
        t = rb.table.cpu().numpy()
        n = rb.n
        assert (t[13] == np.arange(n)).all()                                   # centre offset = identity
        # symmetry R_o^T = R_{26-o}: (i -> r) under o  <=>  (r -> i) under 26-o
        for o in (0, 5, 12):
            r = np.nonzero(t[o] >= 0)[0]
            assert (t[26 - o][t[o][r]] == r).all()
        pl = rb.rules.prefix_list()
        assert pl[-1] == int((t >= 0).sum())
        outr = rb.rules.out_rows.cpu().numpy()
        for o in (0, 13, 26):
            assert (np.diff(outr[pl[o]:pl[o + 1]]) > 0).all()                   # canonical order
        # tiles: perm is a permutation, tile masks cover exactly the table's entries
        perm = rb.tiles.perm.cpu().numpy()
        assert sorted(perm[perm >= 0].tolist()) == list(range(n))
        tm = rb.tiles.tile_mask.cpu().numpy().view(np.uint32)
        order = rb.tiles.tile_order.cpu().numpy()
        assert sorted(order.tolist()) == list(range(len(tm)))
        pc = np.array([bin(int(v)).count("1") for v in tm])[order]
        assert (np.diff(pc) <= 0).all()                                        # cost-descending hand-out order
        if level < 3:
            sb = md.strided_rulebook(sz)
            parent = sb.parent.cpu().numpy()
            child = sb.child.cpu().numpy()
            assert int((child >= 0).sum()) == sb.n_fine                        # each fine row has exactly one (parent, offset)
            f = np.nonzero(child >= 0)
            assert (parent[child[f]] == f[1]).all()
            fc = np.full(sb.n_coarse, sb.n_fine); np.minimum.at(fc, parent, np.arange(sb.n_fine))
            assert (np.diff(fc) > 0).all()                                     # coarse rows by first touching fine row
            sz = tuple(s // 2 for s in sz)


@pytest.mark.parametrize("level,C", [(0, 32), (1, 64), (3, 256)])
def test_conv_kernels_agree_and_are_linear_and_adjoint(scene, gpu, level, C):
    from sparse_rcnn_amd import functional as F, _lib as L
    scn, coords, feats, size, x = scene
    sz = tuple(int(s) >> level for s in size)
    rb = x.metadata.subm_rulebook(sz, 3)
    n, P = rb.n, rb.rules.total
    g = torch.Generator(device="cpu").manual_seed(level)
    X1, X2 = (torch.randn(n, C, generator=g).to(gpu) for _ in range(2))
    G = torch.randn(n, C, generator=g).to(gpu)
    W = (torch.randn(27, C, C, generator=g) * (2.0 / (27 * C)) ** 0.5).to(gpu)
    b = torch.randn(C, generator=g).to(gpu)
    y_ts = F.conv_rules(X1, rb.tiles, n, W, b, C, n_rules=P)                   # mask-sorted tiles, LDS weights
    y_tb = F.gemm_table(X1, rb.table, 27, n, W, b, C)                          # natural-order table, global operands
    scale = y_tb.abs().max().item()
    assert (y_ts - y_tb).abs().max().item() <= 1e-4 * max(1.0, scale)         # two independent kernels agree
    # linearity (bias off)
    ya = F.conv_rules(X1, rb.tiles, n, W, None, C, n_rules=P)
    yb = F.conv_rules(X2, rb.tiles, n, W, None, C, n_rules=P)
    yc = F.conv_rules(2.5 * X1 - 0.5 * X2, rb.tiles, n, W, None, C, n_rules=P)
    assert (yc - (2.5 * ya - 0.5 * yb)).abs().max().item() <= 1e-4 * max(1.0, yc.abs().max().item())
    # adjointness: <conv(X), G> = <X, conv^T(G)> = <W, dW(X, G)>
    dX = F.conv_rules(G, rb.tiles, n, W, None, C, L.F_W_TRANSPOSED | L.F_OFF_REVERSE, n_rules=P)
    dW = F.wgrad_rules(X1, G, rb.rules.in_rows, rb.rules.out_rows, rb.rules.prefix_host, 27)
    lhs = (ya.double() * G.double()).sum().item()
    mid = (X1.double() * dX.double()).sum().item()
    rhs = (W.double() * dW.double()).sum().item()
    ref = max(1.0, abs(lhs))
    assert abs(lhs - mid) / ref < 1e-4 and abs(lhs - rhs) / ref < 1e-4
    # fused bias gradient == column sum
    dW2, db = F.wgrad_bias_rules(X1, G, rb.rules.in_rows, rb.rules.out_rows, rb.rules.prefix_host, 27, 1 << 13)
    assert torch.equal(dW2, dW)
    assert (db.double() - G.double().sum(0)).abs().max().item() <= 1e-3 * max(1.0, G.double().sum(0).abs().max().item())


def test_backbone_step_with_in_launch_k_reduction_equals_two_launch_form_at_150k(scene, gpu):
    """The whole 150k-voxel backbone step (62 tile-convolution launches, 44 of them split over K) with the K-chunk partial
    sums added inside the launch vs by the second launch: every output and gradient word equal.  Run as a whole step so
    that the hand-off happens under the real, uneven load (waves of all levels' tiles, index kernels beside them)."""
    from sparse_rcnn_amd import functional as F
    from sparse_rcnn_amd.unet import Backbone
    scn, coords, feats, size, x = scene
    torch.manual_seed(0)
    net = Backbone(7, (32, 64, 128, 256)).to(gpu)
    cd, fd = coords.to(gpu), feats.to(gpu)
    res = []
    for fused in (False, True, True):
        F.FUSED_K = fused
        try:
            for p in net.parameters():
                p.grad = None
            fin = fd.clone().requires_grad_()
            out = net(cd, fin, size, 1).features
            out.backward(torch.ones_like(out))
            res.append([out.detach().clone(), fin.grad.clone()] + [p.grad.clone() for p in net.parameters()])
        finally:
            F.FUSED_K = True
    for a, b, c in zip(*res):
        assert torch.equal(a, b) and torch.equal(b, c)


def test_backbone_step_full_size_is_finite_and_reproducible(scene, gpu):
    from sparse_rcnn_amd.unet import Backbone
    scn, coords, feats, size, x = scene
    torch.manual_seed(0)
    net = Backbone(7, (32, 64, 128, 256)).to(gpu)
    outs = []
    for _ in range(2):
        for p in net.parameters():
            p.grad = None
        out = net(coords.to(gpu), feats.to(gpu), size, 1).features
        out.backward(torch.ones_like(out))
        outs.append((out.detach().clone(), [p.grad.clone() for p in net.parameters()]))
    assert torch.isfinite(outs[0][0]).all() and outs[0][0].shape == (150_000, 32)
    assert torch.equal(outs[0][0], outs[1][0])                                 # no atomics on the feature path: bitwise
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


def test_config5_shape_600k_voxels_five_levels_to_512_channels(gpu):
    """BASELINE config 5's shape on one GPU, in fp32 (its bf16-storage form: tests/test_gpu_atsize.py::test_cfg5_shape_bf16_properties): grid 1024x1024x512, 600k active
    voxels, U-Net 32..512.  The largest case the configs name: 4.6 M SubM pairs at level 0 (> 10 M over the
    five levels' SubM and strided rulebooks), 512-channel bottom level, coordinates beyond 512.  Checked through size-independent properties; the native index build must agree bit for bit
    with the per-call one at this size too."""
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd.synthetic import make_batch
    from sparse_rcnn_amd.unet import Backbone
    from sparse_rcnn_amd import functional as F
    coords, feats, size, bs, _ = make_batch(1, (1024, 1024, 512), 600_000, seed=5)
    assert int(coords[:, :3].max()) > 512
    cd, fd = coords.to(gpu), feats.to(gpu)
    x = scn.InputLayer(3, size, mode=4)((coords, fd, 1))
    assert x.features.shape[0] == 600_000
    x.metadata.build_pyramid(size, 5, 3)
    sz = tuple(int(s) for s in size)
    rb = x.metadata.subm_rulebook(sz, 3)
    P = rb.rules.total
    assert 4_000_000 < P < 27 * 600_000                                        # surfaces: 7-10 neighbours per voxel
    t = rb.table.cpu().numpy()
    assert (t[13] == np.arange(600_000)).all()
    assert int((t >= 0).sum()) == P                                            # scan total == table population
    r = np.nonzero(t[3] >= 0)[0]
    assert (t[23][t[3][r]] == r).all()                                         # R_o^T = R_{26-o}

    torch.manual_seed(0)
    net = Backbone(7, (32, 64, 128, 256, 512)).to(gpu)
    md_native = net.prefetch_in_thread(cd, size, 1).result()                   # scn_pyramid_build, helper thread
    outs = []
    for md in (None, md_native):
        for p in net.parameters():
            p.grad = None
        out = net(cd, fd, size, 1, metadata=md).features
        out.backward(torch.ones_like(out))
        outs.append((out.detach().clone(), [p.grad.clone() for p in net.parameters()]))
    assert outs[0][0].shape == (600_000, 32) and torch.isfinite(outs[0][0]).all()
    assert torch.equal(outs[0][0], outs[1][0])                                 # both index builds: same rows, same tiles
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.isfinite(a).all() and torch.equal(a, b)

    # bottom level (C = 512): the tile kernel against the natural-order table kernel
    lsz = tuple(s >> 4 for s in sz)
    rb4 = x.metadata.subm_rulebook(lsz, 3)
    n4 = rb4.n
    g = torch.Generator(device="cpu").manual_seed(9)
    X = torch.randn(n4, 512, generator=g).to(gpu)
    W = (torch.randn(27, 512, 512, generator=g) * (2.0 / (27 * 512)) ** 0.5).to(gpu)
    y_ts = F.conv_rules(X, rb4.tiles, n4, W, None, 512, n_rules=rb4.rules.total)
    y_tb = F.gemm_table(X, rb4.table, 27, n4, W, None, 512)
    assert (y_ts - y_tb).abs().max().item() <= 1e-4 * max(1.0, y_tb.abs().max().item())
