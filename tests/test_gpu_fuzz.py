"""-m gpu: seeded random sweep of layer shapes against the CPU oracle -- channel counts that are not multiples of 4, 16
or 32 (the kernels' vector width, MFMA tile and K-chunk), row counts around the tile sizes, thin grids, empty samples,
with and without bias / input ReLU.  The cases are fixed by their seeds; a failure names the drawn configuration."""
import os

import numpy as np
import pytest
import torch

from oracle import scn_oracle as O

pytestmark = pytest.mark.gpu

FEAT_TOL = 1e-4          # BASELINE.json north_star: "features within 1e-4 fp32" (relative to the output scale)
EXTRA = int(os.environ.get("SCN_FUZZ_EXTRA", "0"))      # developer switch: that many more seeds per test


def _seeds(start, n):
    return list(range(start, start + n)) + list(range(10_000 + start * 100, 10_000 + start * 100 + EXTRA))


def _draw(seed):
    torch.manual_seed(seed)                                                    # layer initialisation: fixed per case
    rng = np.random.default_rng(1000 + seed)
    grid = tuple(int(2 * rng.integers(2, 13)) for _ in range(3))             # even extents 4..24 (strided conv needs even)
    batch = int(rng.integers(1, 4))
    cells = grid[0] * grid[1] * grid[2]
    cs = []
    for b in range(batch):
        n = int(min(cells, rng.choice([0, 1, 2, 15, 16, 17, 31, 33, 64, 200, 900, 2500])))
        if b == 0 and n == 0:
            n = 5
        lin = rng.choice(cells, size=n, replace=False)
        p = np.stack(np.unravel_index(lin, grid), 1).reshape(n, 3)
        if n and rng.random() < 0.5:
            p = np.concatenate([p, p[rng.integers(0, n, size=int(rng.integers(1, n + 1)))]])
            rng.shuffle(p)
        cs.append(np.concatenate([p, np.full((len(p), 1), b)], 1))
    coords = torch.from_numpy(np.concatenate(cs).astype(np.int64))
    pick = lambda: int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 16, 17, 23, 31, 32, 33, 40, 48, 63, 64, 65, 80, 96, 112, 130]))
    return rng, coords, torch.tensor(grid), batch, pick(), pick()


def _close(a, b, what, cfg):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, cfg, a.shape, b.shape)
    if a.numel() == 0:
        return
    scale = max(1.0, b.abs().max().item())
    err = (a - b).abs().max().item() / scale
    assert err <= FEAT_TOL, f"{what} {cfg}: max err {err:.3e} (scale {scale:.3g})"


@pytest.mark.parametrize("seed", _seeds(0, 24))
def test_fuzz_submanifold_conv(gpu, seed):
    import sparse_rcnn_amd as scn
    rng, coords, size, batch, cin, cout = _draw(seed)
    k = int(rng.choice([1, 3, 3, 3]))
    relu_in, bias = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    cfg = dict(seed=seed, grid=size.tolist(), batch=batch, points=len(coords), cin=cin, cout=cout, k=k, relu=relu_in,
               bias=bias)
    feats = torch.randn(len(coords), cin, generator=torch.Generator().manual_seed(seed))
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu).requires_grad_(), batch))
    conv = scn.SubmanifoldConvolution(3, cin, cout, k, bias).to(gpu)
    if bias:
        with torch.no_grad():
            conv.bias.normal_(0, 0.5)
    y = (scn.Sequential(scn.ReLU(), conv) if relu_in else conv)(x).features
    scene = O.OracleScene(coords.numpy())
    n = scene.n(0)
    rules = scene.subm_rules(0, k)
    if k == 3:                                                                 # (k = 1 is the identity: no rulebook)
        rb = x.metadata.subm_rulebook(tuple(int(s) for s in size), k)
        pairs, prefix = O.rules_concat(rules)
        assert rb.rules.prefix_list() == prefix.tolist(), cfg                 # rulebook: bit-exact
        assert np.array_equal(rb.rules.in_rows.cpu().numpy(), pairs[:, 0]), cfg
        assert np.array_equal(rb.rules.out_rows.cpu().numpy(), pairs[:, 1]), cfg
    Xo = O.input_layer_fwd(feats, scene.prow, n, 4).requires_grad_()
    W = conv.weight.detach().cpu().requires_grad_()
    b = conv.bias.detach().cpu().requires_grad_() if bias else None
    yo = O.conv(torch.relu(Xo) if relu_in else Xo, W, b, rules, n)
    _close(y, yo, "fwd", cfg)
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(seed + 7))
    ps = (x.features, conv.weight) + ((conv.bias,) if bias else ())
    po = (Xo, W) + ((b,) if bias else ())
    for a, e, name in zip(torch.autograd.grad(y, ps, g.to(gpu)), torch.autograd.grad(yo, po, g), ("dX", "dW", "db")):
        _close(a, e, name, cfg)


@pytest.mark.parametrize("seed", _seeds(100, 12))
def test_fuzz_strided_conv_deconv_and_residual_block(gpu, seed):
    """Convolution 2^3/2 -> pre-activation residual block at the coarse level -> Deconvolution back, as one graph."""
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd.unet import residual_block
    rng, coords, size, batch, cin, cout = _draw(seed)
    cfg = dict(seed=seed, grid=size.tolist(), batch=batch, points=len(coords), cin=cin, cout=cout)
    feats = torch.randn(len(coords), cin, generator=torch.Generator().manual_seed(seed))
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu).requires_grad_(), batch))
    down = scn.Convolution(3, cin, cout, (2, 2, 2), (2, 2, 2), True).to(gpu)
    block = residual_block(cout).to(gpu)
    up = scn.Deconvolution(3, cout, cin, (2, 2, 2), (2, 2, 2), False).to(gpu)
    convs = [m for m in block.modules() if isinstance(m, scn.SubmanifoldConvolution)]
    assert len(convs) == 2
    with torch.no_grad():
        for m in (down, convs[0], convs[1]):
            m.bias.normal_(0, 0.5)
    y = up(scn.Sequential(scn.ReLU())(block(down(x)))).features
    scene = O.OracleScene(coords.numpy())
    st = scene.strided_rules(0)
    n, nc = scene.n(0), scene.n(1)
    sub = scene.subm_rules(1, 3)
    cpu = lambda p: p.detach().cpu().requires_grad_()
    Xo = O.input_layer_fwd(feats, scene.prow, n, 4).requires_grad_()
    Wd, bd, W1, b1, W2, b2, Wu = (cpu(down.weight), cpu(down.bias), cpu(convs[0].weight), cpu(convs[0].bias),
                                   cpu(convs[1].weight), cpu(convs[1].bias), cpu(up.weight))
    d = O.conv(Xo, Wd, bd, st, nc)
    h = O.conv(torch.relu(d), W1, b1, sub, nc)
    r = d + O.conv(torch.relu(h), W2, b2, sub, nc)
    yo = O.conv(torch.relu(r), Wu, None, O.swap_rules(st), n)
    _close(y, yo, "fwd", cfg)
    near0 = min(t.detach().abs().min().item() for t in (d, h, r) if t.numel())
    if near0 < 4e-6:        # a ReLU input within rounding of zero: the two sides may pick different masks, legitimately
        pytest.skip(f"ReLU input {near0:.1e} from zero: backward mask not comparable {cfg}")
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(seed + 7))
    got = torch.autograd.grad(y, (x.features, down.weight, down.bias, convs[0].weight, convs[0].bias, convs[1].weight,
                                  convs[1].bias, up.weight), g.to(gpu))
    exp = torch.autograd.grad(yo, (Xo, Wd, bd, W1, b1, W2, b2, Wu), g)
    for a, e, name in zip(got, exp, ("dX", "dWd", "dbd", "dW1", "db1", "dW2", "db2", "dWu")):
        _close(a, e, name, cfg)


@pytest.mark.parametrize("seed", _seeds(200, 12))
def test_fuzz_input_output_layers(gpu, seed):
    """A3 / A10 on random clouds: row numbering, duplicate maps and counts bit-exact, every mode's features bit-exact."""
    import sparse_rcnn_amd as scn
    rng, coords, size, batch, cin, _ = _draw(seed)
    mode = int(rng.integers(0, 5))
    if mode == 0:                                                              # mode 0 promises unique coordinates
        coords = torch.from_numpy(np.unique(coords.numpy(), axis=0)[rng.permutation(len(np.unique(coords.numpy(), axis=0)))])
    cfg = dict(seed=seed, grid=size.tolist(), batch=batch, points=len(coords), cin=cin, mode=mode)
    feats = torch.randn(len(coords), cin, generator=torch.Generator().manual_seed(seed))
    fg = feats.to(gpu).requires_grad_()
    x = scn.InputLayer(3, size, mode=mode)((coords, fg, batch))
    scene = O.OracleScene(coords.numpy())
    assert np.array_equal(x.get_spatial_locations().numpy(), scene.coords0), cfg
    assert np.array_equal(x.metadata.item_row.cpu().numpy(), scene.prow), cfg
    assert np.array_equal(x.metadata.row_count.cpu().numpy(), scene.counts), cfg
    assert x.batch_size() == batch
    exp = O.input_layer_fwd(feats, scene.prow, scene.n(0), mode)
    assert torch.equal(x.features.detach().cpu(), exp), cfg
    g = torch.randn(exp.shape, generator=torch.Generator().manual_seed(seed + 1))
    x.features.backward(g.to(gpu))
    _close(fg.grad, O.input_layer_bwd(g, scene.prow, mode), "input bwd", cfg)
    xf = x.features.detach().clone().requires_grad_()
    y = scn.ioLayers.OutputLayerFunction.apply(3, x.metadata, xf)
    assert torch.equal(y.detach().cpu(), O.output_layer_fwd(exp, scene.prow)), cfg
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(seed + 2))
    y.backward(gy.to(gpu))
    _close(xf.grad, O.output_layer_bwd(gy, scene.prow, scene.n(0)), "output bwd", cfg)


@pytest.mark.parametrize("seed", _seeds(300, 10))
def test_fuzz_roi_crop(gpu, seed):
    """A11 on random clouds and boxes (fractional corners, boxes that catch nothing, samples without boxes)."""
    from sparse_rcnn_amd import roi
    rng, coords, size, batch, cin, _ = _draw(seed)
    bbox_batch = []
    for b in range(batch):
        nb = int(rng.choice([0, 1, 2, 7, 20]))
        lo = rng.uniform(-3, np.array(size.tolist()), size=(nb, 3))
        ext = rng.uniform(0.2, 14, size=(nb, 3))
        bbox_batch.append(torch.from_numpy(np.stack([lo, lo + ext], 1).astype(np.float32)).reshape(nb, 2, 3))
    cfg = dict(seed=seed, grid=size.tolist(), batch=batch, points=len(coords), cin=cin,
               boxes=[len(b) for b in bbox_batch])
    feats = torch.randn(len(coords), cin, generator=torch.Generator().manual_seed(seed))
    fg = feats.to(gpu).requires_grad_()
    cut = roi.SparseRoiCut(roi.RawToTensorFeatureExtractorCombiner())
    out, (is_inside, counts, splits) = cut((coords, fg, size + 32, batch, [0]), bbox_batch)
    bi, cnt, assoc = O.transform_boxes([b.numpy() for b in bbox_batch])
    src, box_of, inside = O.roi_crop(coords.numpy(), bi, assoc)
    assert np.array_equal(is_inside.numpy(), inside) and list(counts) == list(cnt), cfg
    n_boxes = sum(len(b) for b in bbox_batch)
    if out is None:                                                            # custom_operations.py:71,85-86: no rows -> None
        assert len(src) == 0, cfg
        return
    new_coords = np.concatenate([coords.numpy()[src][:, :3], box_of[:, None]], 1)
    ac, prow, _ = O.input_layer_rules(new_coords)
    assert np.array_equal(out.get_spatial_locations().numpy(), ac), cfg
    assert out.batch_size() == n_boxes, cfg
    exp = O.input_layer_fwd(feats[torch.from_numpy(src)], prow, len(ac), 4)
    assert torch.equal(out.features.detach().cpu(), exp), cfg
    g = torch.randn(exp.shape, generator=torch.Generator().manual_seed(seed + 3))
    out.features.backward(g.to(gpu))
    gsel = O.input_layer_bwd(g, prow, 4)
    gexp = torch.zeros(len(coords), cin, dtype=torch.float64).index_add_(0, torch.from_numpy(src), gsel.double())
    _close(fg.grad, gexp.float(), "roi crop backward", cfg)


@pytest.mark.parametrize("seed", _seeds(400, 10))
def test_fuzz_nms(gpu, seed):
    """N3: greedy NMS keep masks bit-exact on random box sets around the kernel's block sizes."""
    from sparse_rcnn_amd.proposals import non_maximum_suppression
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 257, 1000, 1025, 3000]))
    batch = int(rng.integers(1, 4))
    thr = float(rng.choice([0.0, 0.1, 0.25, 0.5, 0.9]))
    spread = float(rng.choice([4, 20, 60]))
    c = rng.uniform(0, spread, size=(batch, n, 3)); sz = rng.uniform(0.5, 6, size=(batch, n, 3))
    boxes = np.stack([c - sz, c + sz], 2).astype(np.float32)
    if n > 3:
        boxes[:, 1] = boxes[:, 0]                                              # an exact duplicate: IoU 1
    got = non_maximum_suppression(torch.from_numpy(boxes).to(gpu), thr).cpu().numpy()
    for b in range(batch):
        assert np.array_equal(got[b], O.nms(boxes[b], thr)), dict(seed=seed, n=n, batch=batch, thr=thr, sample=b)


@pytest.mark.parametrize("seed", _seeds(500, 8))
def test_fuzz_batchnorm_relu(gpu, seed):
    import sparse_rcnn_amd as scn
    rng, coords, size, batch, c, _ = _draw(seed)
    leak = float(rng.choice([0.0, 0.0, 0.1, 0.333]))
    training = bool(rng.integers(0, 2))
    cfg = dict(seed=seed, points=len(coords), c=c, leak=leak, training=training)
    feats = torch.randn(len(coords), c, generator=torch.Generator().manual_seed(seed)) * 2 + 0.5
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu).requires_grad_(), batch))
    scene = O.OracleScene(coords.numpy())
    n = scene.n(0)
    if training and n < 2:
        return
    bn = (scn.BatchNormLeakyReLU(c, 1e-4, 0.9, leak) if leak else scn.BatchNormReLU(c, 1e-4, 0.9)).to(gpu)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
    bn.train(training)
    rm0, rv0 = bn.running_mean.detach().cpu().clone(), bn.running_var.detach().cpu().clone()
    y = bn(x).features
    Xo = O.input_layer_fwd(feats, scene.prow, n, 4).requires_grad_()
    ga, be = bn.weight.detach().cpu().requires_grad_(), bn.bias.detach().cpu().requires_grad_()
    yo = O.batchnorm_relu_fwd(Xo, ga, be, rm0, rv0, 1e-4, 0.9, leak, training)
    _close(y, yo, "fwd", cfg)
    _close(bn.running_mean, rm0, "running_mean", cfg); _close(bn.running_var, rv0, "running_var", cfg)
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(seed + 8))
    for a, e, name in zip(torch.autograd.grad(y, (x.features, bn.weight, bn.bias), g.to(gpu)),
                          torch.autograd.grad(yo, (Xo, ga, be), g), ("dX", "dgamma", "dbeta")):
        _close(a, e, name, cfg)


UNET_PLANS = [(8, 16), (16, 24, 32), (32, 48, 64, 80), (32, 64, 128), (12, 20, 28, 36), (32, 48, 64, 80, 96)]


@pytest.mark.parametrize("seed", _seeds(600, 6))
def test_fuzz_unet(gpu, seed):
    """A12 with several channel plans -- the reference's own 32,48,64,80(,96) = arange*16+32 (run.py:539-549) among them,
    i.e. K-chunks that are not multiples of 32 -- and 2 to 5 levels, forward and every parameter gradient."""
    from sparse_rcnn_amd.unet import Backbone
    rng = np.random.default_rng(seed)
    channels = UNET_PLANS[seed % len(UNET_PLANS)]
    L = len(channels)
    grid = tuple(int(rng.integers(1, 3)) << L for _ in range(3))              # multiples of 2^levels (A6)
    cells = grid[0] * grid[1] * grid[2]
    batch = int(rng.integers(1, 3))
    cs = []
    for b in range(batch):
        n = int(min(cells // 2, rng.choice([300, 1200, 3000])))
        p = np.stack(np.unravel_index(rng.choice(cells, size=n, replace=False), grid), 1)
        p = np.concatenate([p, p[rng.integers(0, n, size=n // 8)]]); rng.shuffle(p)
        cs.append(np.concatenate([p, np.full((len(p), 1), b)], 1))
    coords = torch.from_numpy(np.concatenate(cs).astype(np.int64))
    cfg = dict(seed=seed, channels=channels, grid=grid, batch=batch, points=len(coords))
    feats = torch.randn(len(coords), 7, generator=torch.Generator().manual_seed(seed))
    params = O.init_unet_params(7, channels, seed=seed)
    net = Backbone(7, channels).to(gpu)
    net.unet.load_oracle_params(params)
    from sparse_rcnn_amd import functional as F
    F.RELU_RECORD = []                       # the sign mask of every slab a ReLU is applied to, in call order (as test_gpu_atsize)
    try:
        out = net(coords, feats.to(gpu), torch.tensor(grid), batch)
        masks = F.RELU_RECORD
    finally:
        F.RELU_RECORD = None
    scene = O.OracleScene(coords.numpy())
    po = {k: v.clone().requires_grad_() for k, v in params.items()}
    exp = O.unet_forward(scene, feats, po, channels)
    _close(out.features, exp, "unet fwd", cfg)
    g = torch.randn(exp.shape, generator=torch.Generator().manual_seed(seed + 1))
    out.features.backward(g.to(gpu))
    exp.backward(g)
    # (1) the oracle with the ReLU masks the HIP forward took (O.FrozenReLU): both sides differentiate the same piecewise-linear
    # function, what is left is summation order -- every parameter gradient within 2e-5 relative L2 (test_gpu_atsize's bound)
    pf = {k: v.clone().requires_grad_() for k, v in params.items()}
    fr = O.FrozenReLU(masks)
    O.unet_forward(scene, feats, pf, channels, relu=fr).backward(g)
    assert fr.k == len(masks), (fr.k, len(masks), cfg)
    for k, p in net.unet.named_oracle_params().items():
        a, e = p.grad.detach().cpu().double(), pf[k].grad.view_as(p).double()
        l2 = ((a - e).norm() / e.norm().clamp_min(1e-12)).item()
        assert l2 <= 2e-5, f"grad {k} {cfg}: relative L2 {l2:.2e} with the device's ReLU masks"
    # (2) the oracle with its OWN ReLU decisions:
    # Gradients of a deep ReLU network are only piecewise continuous: among ~10^6 activations a few lie within fp32
    # rounding of zero, and the two sides may then take different ReLU masks -- the fp64 oracle's own gradient moves by
    # 2e-2 (relative to max |grad|) in one weight slice when its input is scaled by 1 - 5e-7.  Such a flip changes one
    # output channel of one layer (by ~1/rows) and spreads, decaying, towards the input: observed relative L2 errors of
    # the affected parameters are 2e-4 .. 1e-3, against 1e-6 without a flip and >= 1e-1 for one wrong channel -- on the
    # 10^5-row scenes; the drawn scenes here have a few hundred rows per level, where ONE flipped row moves a weight slice by
    # ~1 / rows (round 5: seed 70275 of an extended sweep, 2.3e-2 on `enc1.res1.conv0.weight` with its own masks, 1e-6 with
    # the shared ones).  So: strict with shared masks (above), on the forward pass and on single layers and blocks; here the
    # relative L2 error of every parameter's gradient below 5e-2 (one wrong channel of a layer is >= 1e-1).
    for k, p in net.unet.named_oracle_params().items():
        a, e = p.grad.detach().cpu().double(), po[k].grad.view_as(p).double()
        l2 = ((a - e).norm() / e.norm().clamp_min(1e-12)).item()
        assert l2 <= 5e-2, f"grad {k} {cfg}: relative L2 {l2:.2e}"


@pytest.mark.parametrize("seed", _seeds(700, 12))
def test_fuzz_native_index_build(gpu, seed):
    """scn_pyramid_build (the build bench.py runs on the helper thread) on random scenes and 1-5 levels: every index
    tensor equal to the step-by-step build, and the level-0 and coarse rulebooks equal to the oracle's."""
    from sparse_rcnn_amd.metadata import Metadata
    rng = np.random.default_rng(seed)
    levels = int(rng.integers(1, 6))
    grid = tuple(int(rng.integers(1, 4)) << (levels - 1 + int(rng.integers(0, 2))) for _ in range(3))
    cells = grid[0] * grid[1] * grid[2]
    batch = int(rng.integers(1, 4))
    cs = []
    for b in range(batch):
        n = int(min(cells, rng.choice([0, 1, 17, 300, 2000, 9000])))
        if b == 0 and n == 0:
            n = min(3, cells)
        p = np.stack(np.unravel_index(rng.choice(cells, size=n, replace=False), grid), 1).reshape(n, 3)
        if n and rng.random() < 0.6:
            p = np.concatenate([p, p[rng.integers(0, n, size=max(1, n // 5))]]); rng.shuffle(p)
        cs.append(np.concatenate([p, np.full((len(p), 1), b)], 1))
    coords = torch.from_numpy(np.concatenate(cs).astype(np.int64))
    size = torch.tensor(grid)
    cfg = dict(seed=seed, grid=grid, levels=levels, batch=batch, points=len(coords))
    cg = coords.to(gpu)
    Metadata.LEVELS_HINT.pop(tuple(int(g_) for g_ in grid), None)      # an earlier case on this grid may have gone deeper
    a = Metadata(3); a.set_input(size, cg, batch, 4); a.build_pyramid(size, levels, 3)
    b = Metadata(3).build_native(size, cg, batch, 4, levels, 3)
    eq = torch.equal
    assert eq(a.item_row, b.item_row) and eq(a.row_count, b.row_count) and eq(a.row_first, b.row_first), cfg
    assert a.n_samples == b.n_samples and list(a.grids) == list(b.grids), cfg
    for s in a.grids:
        assert a.grids[s].n == b.grids[s].n and eq(a.grids[s].coords, b.grids[s].coords), (cfg, s)
    assert set(a.subm) == set(b.subm) and set(a.strided) == set(b.strided), cfg
    for key in a.subm:
        ra, rb = a.subm[key], b.subm[key]
        assert eq(ra.table, rb.table) and ra.rules.prefix_list() == rb.rules.prefix_list(), (cfg, key)
        assert eq(ra.rules.in_rows, rb.rules.in_rows) and eq(ra.rules.out_rows, rb.rules.out_rows), (cfg, key)
        for f in ("perm", "tstab", "tile_mask", "tile_order"):
            assert eq(getattr(ra.tiles, f), getattr(rb.tiles, f)), (cfg, key, f)
    for key in a.strided:
        ra, rb = a.strided[key], b.strided[key]
        assert eq(ra.parent, rb.parent) and eq(ra.fine_off, rb.fine_off) and eq(ra.child, rb.child), (cfg, key)
        assert ra.rules.prefix_list() == rb.rules.prefix_list() and eq(ra.rules.in_rows, rb.rules.in_rows), (cfg, key)
        for f in ("perm", "tstab", "tile_mask", "tile_order"):
            assert eq(getattr(ra.tiles, f), getattr(rb.tiles, f)), (cfg, key, f)
    # against the oracle: rows, SubM rules of every level, strided rules between levels
    scene = O.OracleScene(coords.numpy())
    assert np.array_equal(b.item_row.cpu().numpy(), scene.prow), cfg
    sz = grid
    for l in range(levels):
        rb = b.subm_rulebook(sz, 3)
        pairs, prefix = O.rules_concat(scene.subm_rules(l, 3))
        assert rb.rules.prefix_list() == prefix.tolist(), (cfg, l)
        assert np.array_equal(rb.rules.in_rows.cpu().numpy(), pairs[:, 0]), (cfg, l)
        assert np.array_equal(rb.rules.out_rows.cpu().numpy(), pairs[:, 1]), (cfg, l)
        if l + 1 < levels:
            sb = b.strided_rulebook(sz)
            pairs, prefix = O.rules_concat(scene.strided_rules(l))
            assert sb.rules.prefix_list() == prefix.tolist(), (cfg, l)
            assert np.array_equal(sb.rules.in_rows.cpu().numpy(), pairs[:, 0]), (cfg, l)
            assert np.array_equal(sb.rules.out_rows.cpu().numpy(), pairs[:, 1]), (cfg, l)
            sz = tuple(s // 2 for s in sz)


@pytest.mark.parametrize("seed", _seeds(800, 10))
def test_fuzz_voxelisation(gpu, seed):
    """N4: augment_coords with random rotations / scales / offsets, with and without a spatial size, shift and cut-out."""
    from sparse_rcnn_amd import voxelize
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 50, 1000, 20000]))
    pts = (rng.normal(size=(n, 3)) * rng.uniform(0.5, 4, size=3)).astype(np.float32)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    rot = (q * rng.uniform(10, 60)).astype(np.float32)
    off = rng.uniform(0, 1, size=3).astype(np.float32)
    mode = int(rng.choice([0, 2, 3]))            # (a size without shift or start positions is the RNG-driven cut-out)
    size = None if mode == 0 else tuple(int(v) for v in rng.integers(8, 200, size=3))
    shift = int(rng.integers(0, 40)) if mode == 2 else None
    start = tuple(int(v) for v in rng.integers(-20, 60, size=3)) if mode == 3 else None
    cfg = dict(seed=seed, n=n, mode=mode, size=size, shift=shift, start=start)
    kw = dict(rot_and_scale=rot, sub_pixel_offset=off, spatial_size=size)
    if mode == 2:
        kw["shift"] = shift
    if mode == 3:
        kw["start_positions"] = start
    rows, inside, out_size, cshift = voxelize.augment_coords(torch.from_numpy(pts).to(gpu), **kw)
    res, oin, osize, oshift = O.augment_coords(pts, rot, off, size, shift, start)
    assert np.array_equal(rows[:, :3].cpu().numpy(), res), cfg
    assert np.array_equal(inside.cpu().numpy(), oin), cfg
    assert np.array_equal(out_size.numpy(), osize) and np.array_equal(cshift.numpy(), oshift), cfg


@pytest.mark.parametrize("seed", _seeds(900, 10))
def test_fuzz_mask_epilogue(gpu, seed):
    """N2: SparseMaskPredictor / SparseMaskLossSelector from the CSR selection against the oracle's dense-indicator form."""
    from sparse_rcnn_amd import roi
    rng, coords, size, batch, _, _ = _draw(seed)
    cnp = coords.numpy()
    order = np.argsort(cnp[:, 3], kind="stable")                               # batch column non-decreasing (A4)
    coords = torch.from_numpy(cnp[order])
    splits = [int((coords[:, 3] == b).sum()) for b in range(batch)]
    K = int(rng.integers(1, 6))
    bbox_batch = []
    for b in range(batch):
        nb = int(rng.choice([0, 1, 3, 8]))
        lo = rng.uniform(-2, np.array(size.tolist()) - 2, size=(nb, 3))
        bbox_batch.append(torch.from_numpy(np.stack([lo, lo + rng.uniform(1, 12, size=(nb, 3))], 1).astype(np.float32)).reshape(nb, 2, 3))
    boxes, counts, _ = roi.transform_boxes(bbox_batch, size, False)
    cfg = dict(seed=seed, batch=batch, points=len(coords), boxes=list(counts), K=K)
    _, _, sel = roi.roi_cut_device(coords, torch.zeros(len(coords), 1).to(gpu), boxes)
    inside = sel.is_inside().numpy()
    m = int(inside.sum())
    scores = torch.randn(m, K, generator=torch.Generator().manual_seed(seed)) * 3
    bb = sum(counts)
    num_valid = int(rng.choice([0, K]))
    classes = rng.integers(-1, K + 2 if num_valid else K, size=bb)             # -1 and >= num_valid: invalid -> zeros
    pred = roi.mask_predict(scores.to(gpu), sel, counts, splits, torch.from_numpy(classes), num_valid)
    exp = O.mask_predict(scores.numpy(), inside, counts, splits, classes, num_valid)
    assert [tuple(p.shape) for p in pred] == [e.shape for e in exp], cfg
    for p, e in zip(pred, exp):
        assert np.allclose(p.cpu().numpy(), e, rtol=0, atol=1e-6) and np.array_equal(p.cpu().numpy() == 0, e == 0), cfg
    keep_list, assoc_list, labels_list, masks_list = [], [], [], []
    for s, (nb, npts) in enumerate(zip(counts, splits)):
        g = int(rng.integers(1, 4))
        keep = rng.random(nb) < 0.6
        keep_list.append(keep)
        assoc_list.append(rng.integers(0, g, size=int(keep.sum())))
        labels_list.append(rng.integers(0, K, size=g))
        masks_list.append(rng.random((g, npts)) < 0.5)
    sg = scores.to(gpu).requires_grad_()
    p, gt, rows, labels = roi.mask_loss_select(sg, sel, counts, splits, keep_list, assoc_list, labels_list, masks_list)
    ep, eg, er, el = O.mask_loss_select(scores.numpy(), inside, counts, splits, keep_list, assoc_list, labels_list,
                                        masks_list)
    assert np.array_equal(p.detach().cpu().numpy(), ep) and np.array_equal(gt.cpu().numpy(), eg), cfg
    assert list(rows) == list(er) and np.array_equal(labels.numpy(), el), cfg
    if p.numel():
        w = torch.randn(p.shape, generator=torch.Generator().manual_seed(seed + 1))
        (ds,) = torch.autograd.grad(p, sg, w.to(gpu))
        assert abs(ds.sum().item() - w.sum().item()) <= 1e-3 * max(1.0, w.abs().sum().item()), cfg   # one-hot routing


def test_pipelined_index_build_over_changing_scenes(gpu):
    """bench.py's pipeline as a training loop would use it: a DIFFERENT scene every step (sizes grow and shrink, so the
    workspaces are re-used and re-allocated), the next scene's index structures built by the helper thread while the
    current step's forward and backward run.  Every step must give the bits of the same step run without the pipeline."""
    from sparse_rcnn_amd.unet import Backbone
    rng = np.random.default_rng(77)
    scenes = []
    for step, n in enumerate([2500, 400, 6000, 1, 3000, 9000, 50, 2500]):
        grid = (32, 32, 16) if step % 3 else (64, 32, 32)
        cells = grid[0] * grid[1] * grid[2]
        p = np.stack(np.unravel_index(rng.choice(cells, size=n, replace=False), grid), 1).reshape(n, 3)
        p = np.concatenate([p, p[rng.integers(0, n, size=max(1, n // 6))]]); rng.shuffle(p)
        b = (np.arange(len(p)) * 2 // len(p))[:, None]                         # two samples
        coords = torch.from_numpy(np.concatenate([p, b], 1).astype(np.int64)).to(gpu)
        feats = torch.randn(len(p), 7, generator=torch.Generator().manual_seed(step)).to(gpu)
        scenes.append((coords, feats, torch.tensor(grid)))
    torch.manual_seed(5)
    net = Backbone(7, (16, 24, 32)).to(gpu)

    def run(coords, feats, size, md):
        for q in net.parameters():
            q.grad = None
        f = feats.clone().requires_grad_()
        out = net(coords, f, size, 2, metadata=md).features
        out.backward(torch.ones_like(out))
        return [out.detach().clone(), f.grad.clone()] + [q.grad.clone() for q in net.parameters()]

    plain = [run(c, f, s, None) for c, f, s in scenes]
    pending = net.prefetch_in_thread(scenes[0][0], scenes[0][2], 2)
    for i, (c, f, s) in enumerate(scenes):
        md = pending.result()
        pending = net.prefetch_in_thread(*[scenes[i + 1][j] for j in (0, 2)], 2) if i + 1 < len(scenes) else None
        got = run(c, f, s, md)
        for a, e in zip(got, plain[i]):
            assert torch.equal(a, e), f"step {i}"


@pytest.mark.parametrize("seed", _seeds(1000, 10))
def test_fuzz_nin_join_pool_dense(gpu, seed):
    """A9 / A13 / N1 on random shapes: NetworkInNetwork over a JoinTable, Max/AveragePooling, SparseToDense."""
    import sparse_rcnn_amd as scn
    rng, coords, size, batch, cin, cout = _draw(seed)
    cfg = dict(seed=seed, grid=size.tolist(), batch=batch, points=len(coords), cin=cin, cout=cout)
    feats = torch.randn(len(coords), cin, generator=torch.Generator().manual_seed(seed))
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu).requires_grad_(), batch))
    scene = O.OracleScene(coords.numpy())
    n = scene.n(0)
    Xo = O.input_layer_fwd(feats, scene.prow, n, 4).requires_grad_()
    # join (x, relu(x)) -> 1x1
    nin = scn.NetworkInNetwork(2 * cin, cout, True).to(gpu)
    with torch.no_grad():
        nin.bias.normal_(0, 0.5)
    j = scn.JoinTable()([x, scn.Sequential(scn.ReLU())(x)])
    y = nin(j).features
    W, b = nin.weight.detach().cpu().requires_grad_(), nin.bias.detach().cpu().requires_grad_()
    yo = torch.cat([Xo, torch.relu(Xo)], 1) @ W.reshape(2 * cin, cout) + b
    _close(y, yo, "nin fwd", cfg)
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(seed + 1))
    for a, e, name in zip(torch.autograd.grad(y, (x.features, nin.weight, nin.bias), g.to(gpu)),
                          torch.autograd.grad(yo, (Xo, W, b), g), ("dX", "dW", "db")):
        _close(a, e.view_as(a.cpu()), "nin " + name, cfg)
    # pooling 2^3 / 2
    average = bool(rng.integers(0, 2))
    pool = (scn.AveragePooling if average else scn.MaxPooling)(3, (2, 2, 2), (2, 2, 2))
    yp = pool(x)
    scene.strided_rules(0)
    po = O.pool_fwd(Xo, scene.strided[0]["child"], average)
    _close(yp.features, po, "pool fwd", cfg)
    assert np.array_equal(yp.get_spatial_locations().numpy(), scene.strided[0]["coords"]), cfg
    gp = torch.randn(po.shape, generator=torch.Generator().manual_seed(seed + 2))
    (gx,) = torch.autograd.grad(yp.features, x.features, gp.to(gpu))
    (ox,) = torch.autograd.grad(po, Xo, gp)
    _close(gx, ox, "pool bwd", cfg)
    # dense
    d = scn.SparseToDense(3, cin)(x)
    exp = O.sparse_to_dense(Xo.detach(), scene.coords0, size.tolist(), batch)
    assert torch.equal(d.detach().cpu(), exp), cfg


@pytest.mark.parametrize("seed", _seeds(1100, 12))
def test_fuzz_conv_tiles_bf16(gpu, seed):
    """scn_conv_tiles_bf16 (bf16 storage, fp32 accumulation) on random scenes: SubM 3^3 tables and the 2^3 / 2 child
    tables, channel counts that are multiples of 8 (the kernel's requirement) but not of the 32-channel K-chunk, odd
    output widths, K split -- against the oracle on the same bf16-rounded operands, within 2^-7 of the output scale."""
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import functional as F, _lib as L
    rng, coords, size, batch, _, cout = _draw(seed)
    cin = int(rng.choice([8, 16, 24, 32, 40, 64, 72, 96, 128, 136]))
    strided = bool(rng.integers(0, 2))
    relu_in = bool(rng.integers(0, 2))
    cfg = dict(seed=seed, grid=size.tolist(), batch=batch, points=len(coords), cin=cin, cout=cout, strided=strided,
               relu=relu_in)
    x = scn.InputLayer(3, size, mode=4)((coords, torch.zeros(len(coords), 1).to(gpu), batch))
    scene = O.OracleScene(coords.numpy())
    sz = tuple(int(s) for s in size)
    if strided:
        sb = x.metadata.strided_rulebook(sz)
        tiles, n_in, n_out, n_off = sb.tiles, sb.n_fine, sb.n_coarse, 8
        rules = scene.strided_rules(0)
    else:
        rb = x.metadata.subm_rulebook(sz, 3)
        tiles, n_in, n_out, n_off = rb.tiles, rb.n, rb.n, 27
        rules = scene.subm_rules(0, 3)
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(n_in, cin, generator=g).to(torch.bfloat16)
    W = torch.randn(n_off, cin, cout, generator=g) * (2.0 / (n_off * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.5
    y = F.conv_rules_bf16(X.to(gpu), tiles, n_out, W.to(gpu), b.to(gpu), cout, L.F_RELU_IN if relu_in else 0)
    xin = torch.relu(X.float()) if relu_in else X.float()
    yo = O.conv_fwd(xin, rules, W.to(torch.bfloat16).float(), b, n_out)
    a, e = y.float().cpu().double(), yo.double()
    assert a.shape == e.shape, cfg
    if a.numel():
        err = (a - e).abs().max().item() / max(1.0, e.abs().max().item())
        assert err <= 2.0 ** -7, f"{cfg}: {err:.3e}"


@pytest.mark.parametrize("seed", _seeds(1200, 6))
def test_fuzz_backbone_bf16_storage_modes(gpu, seed):
    """Backbone with bf16-stored features (residual units only / everything after the first layer) against the fp32
    backbone with the same parameters, over channel plans (multiples of 8), 2-4 levels and random scenes: outputs within
    3 % and parameter gradients within 15 % over all parameters (50 % for any single one) in relative L2 -- a bound against gross errors (a wrong kernel gives ~100 %),
    not a precision claim: bf16 keeps 8 significant bits per stored activation AND gradient value, and an activation
    within 0.4 % of zero may take the other ReLU branch; the kernels themselves are pinned by the exact tests in
    test_gpu_parity.py (bit-equality with the fp32 kernels on widened operands, oracle with the same roundings)."""
    from sparse_rcnn_amd.unet import Backbone
    rng = np.random.default_rng(seed)
    channels = [(16, 24), (32, 48, 64), (8, 16, 24, 32), (32, 64, 128), (40, 56, 72)][seed % 5]
    L = len(channels)
    grid = tuple(int(rng.integers(1, 3)) << L for _ in range(3))
    cells = grid[0] * grid[1] * grid[2]
    n = int(min(cells // 2, rng.choice([500, 2000, 4000])))
    p = np.stack(np.unravel_index(rng.choice(cells, size=n, replace=False), grid), 1)
    p = np.concatenate([p, p[rng.integers(0, n, size=n // 8)]]); rng.shuffle(p)
    coords = torch.from_numpy(np.concatenate([p, np.zeros((len(p), 1), np.int64)], 1).astype(np.int64))
    feats = torch.randn(len(coords), 7, generator=torch.Generator().manual_seed(seed)).to(gpu)
    mode = "all" if seed % 2 else True
    cfg = dict(seed=seed, channels=channels, grid=grid, points=len(coords), mode=mode)
    torch.manual_seed(seed)
    ref = Backbone(7, channels).to(gpu)
    mix = Backbone(7, channels, bf16_blocks=mode).to(gpu)
    mix.load_state_dict(ref.state_dict())
    outs = []
    for net in (ref, mix):
        out = net(coords, feats, torch.tensor(grid), 1).features
        out.backward(torch.ones_like(out))
        outs.append(out.detach())
    l2 = ((outs[1] - outs[0]).norm() / outs[0].norm().clamp_min(1e-12)).item()
    assert torch.isfinite(outs[1]).all() and l2 < 3e-2, (cfg, l2)
    num = den = 0.0
    for (k, a), b in zip(ref.named_parameters(), mix.parameters()):
        rel = ((b.grad - a.grad).norm() / a.grad.norm().clamp_min(1e-12)).item()
        assert torch.isfinite(b.grad).all() and rel < 0.5, (cfg, k, rel)       # (deep levels hold a few dozen rows)
        num += (b.grad - a.grad).double().pow(2).sum().item(); den += a.grad.double().pow(2).sum().item()
    assert (num / max(den, 1e-30)) ** 0.5 < 0.15, (cfg, (num / max(den, 1e-30)) ** 0.5)


@pytest.mark.parametrize("seed", _seeds(700, 40))
def test_fuzz_radix_select_topk(gpu, seed):
    """`scn_topk_boxes` on drawn fields: sizes around the pass boundaries (2048-element tiles, 256 workgroups), k from 1 to
    2048 (and k = n), value distributions that put the k-th score in a crowded or an empty bucket (few distinct values, one
    dominant constant, heavy tails, exact duplicates of the threshold), batches -- against the first k entries of a stable
    descending sort on the CPU (values bit for bit, indices, gathered boxes)."""
    from sparse_rcnn_amd import proposals as PR
    rng = np.random.default_rng(9000 + seed)
    b = int(rng.integers(1, 4))
    n = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 2047, 2048, 2049, 4097, 30000, 65537, 300001, 524288 + 3]))
    k = int(min(n, rng.choice([1, 2, 7, 64, 100, 1000, 1024, 2047, 2048])))
    if rng.random() < 0.15:
        k = min(n, 2048)
    g = torch.Generator().manual_seed(seed)
    kind = int(rng.integers(0, 6))
    if kind == 0:
        s = torch.randn(b, n, generator=g)
    elif kind == 1:
        s = torch.sigmoid(torch.randn(b, n, generator=g) * 3 - 4)
    elif kind == 2:                                                     # few distinct values
        s = torch.randint(0, int(rng.integers(1, 40)), (b, n), generator=g).float() * 0.125 - 1
    elif kind == 3:                                                     # one dominant constant around a thin set of others
        s = torch.full((b, n), float(rng.random()))
        m = max(1, n // int(rng.choice([3, 50, 1000])))
        s[:, torch.randperm(n, generator=g)[:m]] = torch.rand(m, generator=g)
    elif kind == 4:                                                     # heavy tail over many binades
        s = torch.randn(b, n, generator=g) * torch.exp(torch.randn(b, n, generator=g) * 6)
    else:                                                               # exact duplicates of whatever the threshold will be
        s = torch.randn(b, n, generator=g)
        thr = torch.sort(s, dim=1, descending=True)[0][:, k - 1:k]
        dup = torch.rand(b, n, generator=g) < 0.3
        s = torch.where(dup, thr.expand(b, n), s)
    boxes = torch.randn(b, n, 2, 3, generator=g)
    exp_v, exp_i = torch.sort(s, dim=1, descending=True, stable=True)
    exp_i = exp_i[:, :k]
    v, i, bx = PR.topk_boxes(s.to(gpu), boxes.to(gpu), k)
    what = f"b={b} n={n} k={k} kind={kind}"
    assert i.dtype == torch.int64 and torch.equal(i.cpu(), exp_i), what
    assert torch.equal(v.cpu().view(torch.int32), s.gather(1, exp_i).view(torch.int32)), what
    assert torch.equal(bx.cpu(), boxes[torch.arange(b).unsqueeze(1), exp_i]), what
