"""-m gpu parity tests: the HIP path (through the C ABI / Python operator mirror) against the CPU oracle on the same
seeded inputs.  Index work must be BIT-EXACT; fp32 features within the north star's 1e-4 (relative to the output
scale, stated per test)."""
import glob
import os

import numpy as np
import pytest
import torch

from sparse_rcnn_amd._lib import switches as _SW      # developer switches of the library: scn_debug_set, not the environment

from oracle import scn_oracle as O

pytestmark = pytest.mark.gpu

FEAT_TOL = 1e-4          # BASELINE.json north_star: "features within 1e-4 fp32"


def _scn():
    import sparse_rcnn_amd as scn
    return scn


def _cloud(seed, grid=(24, 20, 16), n=1500, batch=2, dup=200):
    rng = np.random.default_rng(seed)
    cs = []
    for b in range(batch):
        lin = rng.choice(grid[0] * grid[1] * grid[2], size=n, replace=False)
        p = np.stack(np.unravel_index(lin, grid), 1)
        p = np.concatenate([p, p[rng.integers(0, n, size=dup)]]) if dup else p
        rng.shuffle(p)
        cs.append(np.concatenate([p, np.full((len(p), 1), b)], 1))
    return torch.from_numpy(np.concatenate(cs).astype(np.int64)), torch.tensor(grid), batch


def _close(a, b, tol=FEAT_TOL, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = b.abs().max().item()          # relative to the oracle's own scale (no floor at 1: small outputs are held
    scale = scale if scale > 0 else 1.0   # to a relative bound too; an all-zero expectation is held to tol absolutely)
    err = (a - b).abs().max().item() / scale
    assert err <= tol, f"{what}: max err {err:.3e} (scale {scale:.3g}) > {tol}"


def _grads(y, params, g):
    return torch.autograd.grad(y, params, g, allow_unused=False)


def _input(gpu, seed=0, cin=5, mode=4, **kw):
    scn = _scn()
    coords, size, batch = _cloud(seed, **kw)
    feats = torch.randn(len(coords), cin, generator=torch.Generator().manual_seed(seed + 100))
    fg = feats.to(gpu).requires_grad_()
    x = scn.InputLayer(3, size, mode=mode)((coords, fg, batch))
    scene = O.OracleScene(coords.numpy())
    return scn, coords, feats, fg, x, scene, size


# ---------------------------------------------------------------------------------------- index path
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_input_output_layer_modes(gpu, mode):
    dup = 0 if mode == 0 else 300
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=mode, cin=7, mode=mode, dup=dup)
    assert np.array_equal(x.get_spatial_locations().numpy(), scene.coords0)            # first-occurrence order
    assert np.array_equal(x.metadata.item_row.cpu().numpy(), scene.prow)
    assert np.array_equal(x.metadata.row_count.cpu().numpy(), scene.counts)
    assert x.batch_size() == 2
    exp = O.input_layer_fwd(feats, scene.prow, scene.n(0), mode)
    assert torch.equal(x.features.detach().cpu(), exp), "InputLayer features must be bit-exact (fp64 accumulate)"
    g = torch.randn(exp.shape, generator=torch.Generator().manual_seed(9))
    x.features.backward(g.to(gpu))
    _close(fg.grad, O.input_layer_bwd(g, scene.prow, mode), 1e-6, "input bwd")
    # OutputLayer: one row per original point, backward = segment sum
    xf = x.features.detach().clone().requires_grad_()
    y = scn.ioLayers.OutputLayerFunction.apply(3, x.metadata, xf)
    assert torch.equal(y.detach().cpu(), O.output_layer_fwd(exp, scene.prow))
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(10))
    y.backward(gy.to(gpu))
    _close(xf.grad, O.output_layer_bwd(gy, scene.prow, scene.n(0)), 1e-6, "output bwd")


def test_stale_or_mismatched_index_announcements_are_not_adopted(gpu):
    """ADVICE r2 (metadata.py): a batch announced with scn.prefetch_index and never run must not be adopted by a later
    same-shaped tensor with other coordinates (the entry holds the announced tensor: its identity cannot be recycled); an
    in-place torch modification after the announcement, another mode or batch_size, or a generator closed early drop the
    announcement; the InputLayer then builds its own structures -- checked against the oracle of the coordinates it got."""
    scn = _scn()
    from sparse_rcnn_amd import metadata as MD
    MD.drop_prefetched()
    ca, size, batch = _cloud(31)
    cb, _, _ = _cloud(32)
    assert ca.shape == cb.shape and not torch.equal(ca, cb)
    feats = torch.randn(len(ca), 4, generator=torch.Generator().manual_seed(1)).to(gpu)

    def rows(x):
        return x.get_spatial_locations().numpy()
    # A announced, dropped (never run); B arrives: same shape, other values
    scn.prefetch_index(ca, size, batch)
    xb = scn.InputLayer(3, size, mode=4)((cb, feats, batch))
    assert np.array_equal(rows(xb), O.OracleScene(cb.numpy()).coords0) and xb.metadata._prepared_for is None
    assert len(MD._prefetched) == 1                                  # A is still parked (and keeps its tensor alive)
    xa = scn.InputLayer(3, size, mode=4)((ca, feats, batch))          # ... and is adopted by A itself
    assert xa.metadata._prepared_for is not None and np.array_equal(rows(xa), O.OracleScene(ca.numpy()).coords0)
    assert not MD._prefetched
    # modified in place after the announcement: version differs -> not adopted
    cc = ca.clone()
    scn.prefetch_index(cc, size, batch)
    cc.copy_(cb)
    xc = scn.InputLayer(3, size, mode=4)((cc, feats, batch))
    assert xc.metadata._prepared_for is None and np.array_equal(rows(xc), O.OracleScene(cb.numpy()).coords0)
    # announced for mode 4; a mode-0 layer keeps its own check (duplicates -> error), a batch_size-3 layer its own count
    cd = ca.clone()
    scn.prefetch_index(cd, size, batch)
    with pytest.raises(scn.ScnError, match="unique"):
        scn.InputLayer(3, size, mode=0)((cd, feats, batch))
    ce = ca.clone()
    scn.prefetch_index(ce, size, batch)
    xe = scn.InputLayer(3, size, mode=4)((ce, feats, 3))
    assert xe.batch_size() == 3 and xe.metadata._prepared_for is None
    # a loop that ends early forgets what it announced
    MD.drop_prefetched()
    batches = [(ca.clone(), size, batch) for _ in range(3)]
    for i, bt in enumerate(scn.index_prefetching(batches, lambda t: t)):
        if i == 0:
            break
    assert not MD._prefetched


def test_empty_and_zero_row_samples(gpu):
    scn = _scn()
    # batch_size fixes the sample count even when a sample has no rows (ROI path, roi_select_sparse.py:49-50,74-81)
    coords = torch.tensor([[1, 2, 3, 0], [1, 2, 3, 0], [4, 4, 4, 3]])
    x = scn.InputLayer(3, torch.tensor([8, 8, 8]), mode=4)((coords, torch.ones(3, 2).to(gpu), 5))
    assert x.batch_size() == 5 and x.features.shape == (2, 2)
    e = scn.InputLayer(3, torch.tensor([8, 8, 8]), mode=4)((torch.zeros(0, 4, dtype=torch.long), torch.zeros(0, 2).to(gpu), 2))
    assert e.features.shape == (0, 2) and e.batch_size() == 2


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, dict(grid=(64, 64, 32), n=20000, batch=1, dup=2000))])
def test_rulebooks_bit_exact(gpu, seed, kw):
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=seed, **kw)
    md = x.metadata
    for k in (3,):
        nbr, rules = O.subm_rulebook(scene.coords0, k)
        rb = md.subm_rulebook(size, k)
        assert np.array_equal(rb.table.cpu().numpy(), nbr)
        pairs, prefix = O.rules_concat(rules)
        assert rb.rules.prefix_list() == prefix.tolist()
        assert np.array_equal(rb.rules.in_rows.cpu().numpy(), pairs[:, 0])
        assert np.array_equal(rb.rules.out_rows.cpu().numpy(), pairs[:, 1])
    level_size, level_coords = tuple(int(s) for s in size), scene.coords0
    for level in range(2):                                                   # two strided transitions
        ref = O.strided_rulebook(level_coords, 2)
        rb = md.strided_rulebook(level_size)
        coarse = tuple(s // 2 for s in level_size)
        assert np.array_equal(md.get_spatial_locations(coarse).numpy(), ref["coords"])   # canonical coarse row order
        assert np.array_equal(rb.parent.cpu().numpy(), ref["parent"])
        assert np.array_equal(rb.fine_off.cpu().numpy(), ref["off"])
        assert np.array_equal(rb.child.cpu().numpy(), ref["child"])
        pairs, prefix = O.rules_concat(ref["rules"])
        assert rb.rules.prefix_list() == prefix.tolist()
        assert np.array_equal(rb.rules.in_rows.cpu().numpy(), pairs[:, 0])
        assert np.array_equal(rb.rules.out_rows.cpu().numpy(), pairs[:, 1])
        nbr, _ = O.subm_rulebook(ref["coords"], 3)                           # coarse grid's own hash works
        assert np.array_equal(md.subm_rulebook(coarse, 3).table.cpu().numpy(), nbr)
        level_size, level_coords = coarse, ref["coords"]


def test_odd_size_strided_is_rejected(gpu):
    scn = _scn()
    x = scn.InputLayer(3, torch.tensor([9, 8, 8]), mode=4)((torch.tensor([[1, 2, 3, 0]]), torch.ones(1, 2).to(gpu), 1))
    with pytest.raises(scn.ScnError):
        scn.Convolution(3, 2, 4, (2, 2, 2), (2, 2, 2), True).to(gpu)(x)
    with pytest.raises(NotImplementedError):
        scn.Convolution(3, 2, 4, (3, 3, 3), (2, 2, 2), True)


def test_coordinate_range_limits(gpu):
    """16 bits per coordinate field: 65535 is accepted (and neighbours across the limit simply do not exist), 65536 and
    negative values are rejected with ScnError -- never a silent wrap."""
    scn = _scn()
    size = torch.tensor([65536, 65536, 65536])
    coords = torch.tensor([[65535, 65535, 65535, 0], [65534, 65535, 65535, 0], [0, 0, 0, 0]])
    x = scn.InputLayer(3, size, mode=4)((coords, torch.ones(3, 4).to(gpu), 1))
    rb = x.metadata.subm_rulebook(tuple(int(v) for v in size), 3)
    nbr, _ = O.subm_rulebook(coords.numpy(), 3)
    assert np.array_equal(rb.table.cpu().numpy(), nbr)
    for bad in ([65536, 0, 0, 0], [0, -1, 0, 0]):
        with pytest.raises(scn.ScnError):
            scn.InputLayer(3, size, mode=4)((torch.tensor([bad]), torch.ones(1, 4).to(gpu), 1))


# ---------------------------------------------------------------------------------------- feature path
CHANNELS = [(3, 16), (7, 32), (32, 32), (16, 23), (23, 20), (64, 48), (40, 8)]


@pytest.mark.parametrize("n", [1, 15, 16, 17, 33])
@pytest.mark.parametrize("cin,cout", [(32, 32), (5, 7)])
def test_tiny_scenes_and_tile_boundaries(gpu, n, cin, cout):
    """Row counts around the 16-row tile size, down to a single voxel, and a batch whose middle sample is empty."""
    scn = _scn()
    rng = np.random.default_rng(n)
    lin = rng.choice(6 * 6 * 6, size=n, replace=False)
    p = np.stack(np.unravel_index(lin, (6, 6, 6)), 1)
    b = np.where(np.arange(n) < (n + 1) // 2, 0, 2)[:, None]                 # samples 0 and 2 of 3; sample 1 is empty
    coords = torch.from_numpy(np.concatenate([p, b], 1).astype(np.int64))
    feats = torch.randn(n, cin, generator=torch.Generator().manual_seed(n))
    fg = feats.to(gpu).requires_grad_()
    x = scn.InputLayer(3, torch.tensor([8, 8, 8]), mode=4)((coords, fg, 3))
    assert x.batch_size() == 3
    conv = scn.SubmanifoldConvolution(3, cin, cout, 3, True).to(gpu)
    y = scn.Sequential(scn.ReLU(), conv)(x).features
    scene = O.OracleScene(coords.numpy())
    Xo = O.input_layer_fwd(feats, scene.prow, scene.n(0), 4).requires_grad_()
    W, bb = conv.weight.detach().cpu().requires_grad_(), conv.bias.detach().cpu().requires_grad_()
    yo = O.conv(torch.relu(Xo), W, bb, scene.subm_rules(0, 3), scene.n(0))
    _close(y, yo, what="fwd")
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(1))
    gx, gw, gb = _grads(y, (x.features, conv.weight, conv.bias), g.to(gpu))
    ox, ow, ob = _grads(yo, (Xo, W, bb), g)
    _close(gx, ox, what="dX"); _close(gw, ow, what="dW"); _close(gb, ob, what="db")


@pytest.mark.parametrize("cin,cout", [(128, 128), (256, 128), (64, 128), (128, 32)])
def test_wide_layers_take_the_quad_and_k_paths(gpu, cin, cout):
    """Channel counts of the deep U-Net levels: K-split slabs in scn_conv_tiles, the 128x128 (QUAD) and 64x64 (K mode)
    wave blocks of the weight gradient, rectangular blocks -- against the oracle, not only against each other."""
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=11, cin=cin, n=900, dup=100)
    conv = scn.SubmanifoldConvolution(3, cin, cout, 3, True).to(gpu)
    y = conv(x).features
    n = scene.n(0)
    Xo = O.input_layer_fwd(feats, scene.prow, n, 4).requires_grad_()
    W, b = conv.weight.detach().cpu().requires_grad_(), conv.bias.detach().cpu().requires_grad_()
    yo = O.conv(Xo, W, b, scene.subm_rules(0, 3), n)
    _close(y, yo, what="fwd")
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(5))
    gx, gw, gb = _grads(y, (x.features, conv.weight, conv.bias), g.to(gpu))
    ox, ow, ob = _grads(yo, (Xo, W, b), g)
    _close(gx, ox, what="dX"); _close(gw, ow, what="dW"); _close(gb, ob, what="db")


@pytest.mark.parametrize("cin,cout", CHANNELS)
@pytest.mark.parametrize("k", [1, 3])
@pytest.mark.parametrize("relu_in", [False, True])
def test_submanifold_conv(gpu, cin, cout, k, relu_in):
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=3, cin=cin)
    conv = scn.SubmanifoldConvolution(3, cin, cout, k, True).to(gpu)
    with torch.no_grad():
        conv.bias.normal_(0, 0.5)
    net = scn.Sequential(scn.ReLU(), conv) if relu_in else conv
    y = net(x).features
    rules = scene.subm_rules(0, k)
    n = scene.n(0)
    Xo = O.input_layer_fwd(feats, scene.prow, n, 4).requires_grad_()
    W, b = conv.weight.detach().cpu().requires_grad_(), conv.bias.detach().cpu().requires_grad_()
    yo = O.conv(torch.relu(Xo) if relu_in else Xo, W, b, rules, n)
    _close(y, yo, what="fwd")
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(5))
    gx, gw, gb = _grads(y, (x.features, conv.weight, conv.bias), g.to(gpu))
    ox, ow, ob = _grads(yo, (Xo, W, b), g)
    _close(gx, ox, what="dX"); _close(gw, ow, what="dW"); _close(gb, ob, what="db")


def test_submanifold_conv_5_cubed(gpu):
    """`scn.SubmanifoldConvolution(..., filter_size=5)`: 125 offsets -- beyond the tile kernels' 27-bit masks and the rule-list
    entry points' 32 offsets per call: the table-walk GEMM forward and backward-data, the weight gradient in chunks of 32
    offsets.  Forward and every gradient against the oracle."""
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=9, cin=8)
    conv = scn.SubmanifoldConvolution(3, 8, 16, 5, True).to(gpu)
    with torch.no_grad():
        conv.bias.normal_(0, 0.5)
    y = scn.Sequential(scn.ReLU(), conv)(x).features
    rules, n = scene.subm_rules(0, 5), scene.n(0)
    Xo = O.input_layer_fwd(feats, scene.prow, n, 4).requires_grad_()
    W, b = conv.weight.detach().cpu().requires_grad_(), conv.bias.detach().cpu().requires_grad_()
    yo = O.conv(torch.relu(Xo), W, b, rules, n)
    _close(y, yo, what="fwd")
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(5))
    gx, gw, gb = _grads(y, (x.features, conv.weight, conv.bias), g.to(gpu))
    ox, ow, ob = _grads(yo, (Xo, W, b), g)
    _close(gx, ox, what="dX"); _close(gw, ow, what="dW"); _close(gb, ob, what="db")


@pytest.mark.parametrize("cin,cout,k,groups", [(16, 32, 3, 2), (32, 32, 3, 4), (24, 24, 1, 24), (6, 9, 3, 3)])
def test_grouped_submanifold_conv(gpu, cin, cout, k, groups):
    """`scn.SubmanifoldConvolution(..., groups=G)` (module_factory.py:398-406 passes the argument through; :596 asks for a
    depthwise 1^3 layer, groups = channels): parameter in SparseConvNet's grouped layout [k^3, G, nIn/G, nOut/G], group g maps
    input channels [g nIn/G, (g+1) nIn/G) to output channels [g nOut/G, (g+1) nOut/G).  Against the oracle convolution run
    once per group on its channel slice: forward, dX, the grouped dW and db."""
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=6, cin=cin)
    conv = scn.SubmanifoldConvolution(3, cin, cout, k, True, groups=groups).to(gpu)
    assert tuple(conv.weight.shape) == (k ** 3, groups, cin // groups, cout // groups)
    assert tuple(conv.state_dict()["weight"].shape) == tuple(conv.weight.shape)
    with torch.no_grad():
        conv.bias.normal_(0, 0.5)
    y = scn.Sequential(scn.ReLU(), conv)(x).features
    rules, n = scene.subm_rules(0, k), scene.n(0)
    Xo = O.input_layer_fwd(feats, scene.prow, n, 4).requires_grad_()
    W, b = conv.weight.detach().cpu().requires_grad_(), conv.bias.detach().cpu().requires_grad_()
    ci, co = cin // groups, cout // groups
    A = torch.relu(Xo)
    yo = torch.cat([O.conv(A[:, g * ci:(g + 1) * ci], W[:, g], b[g * co:(g + 1) * co], rules, n) for g in range(groups)], 1)
    _close(y, yo, what="fwd")
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(5))
    gx, gw, gb = _grads(y, (x.features, conv.weight, conv.bias), g.to(gpu))
    ox, ow, ob = _grads(yo, (Xo, W, b), g)
    _close(gx, ox, what="dX"); _close(gw, ow, what="dW"); _close(gb, ob, what="db")
    with pytest.raises(ValueError):
        scn.SubmanifoldConvolution(3, 10, 16, 3, True, groups=4)


@pytest.mark.parametrize("cin,cout", [(4, 6), (32, 64), (23, 32), (48, 64)])
@pytest.mark.parametrize("relu_in", [False, True])
def test_strided_conv_and_deconv(gpu, cin, cout, relu_in):
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=4, cin=cin)
    down = scn.Convolution(3, cin, cout, (2, 2, 2), (2, 2, 2), True).to(gpu)
    up = scn.Deconvolution(3, cout, cin, (2, 2, 2), (2, 2, 2), True).to(gpu)
    with torch.no_grad():
        down.bias.normal_(0, 0.5); up.bias.normal_(0, 0.5)
    d = (scn.Sequential(scn.ReLU(), down) if relu_in else down)(x)
    u = (scn.Sequential(scn.ReLU(), up) if relu_in else up)(d)
    assert tuple(int(s) for s in d.spatial_size) == tuple(int(s) // 2 for s in size)
    assert tuple(int(s) for s in u.spatial_size) == tuple(int(s) for s in size)
    rules = scene.strided_rules(0)
    n, nc = scene.n(0), scene.n(1)
    act = torch.relu if relu_in else (lambda t: t)
    Xo = O.input_layer_fwd(feats, scene.prow, n, 4).requires_grad_()
    Wd, bd = down.weight.detach().cpu().requires_grad_(), down.bias.detach().cpu().requires_grad_()
    Wu, bu = up.weight.detach().cpu().requires_grad_(), up.bias.detach().cpu().requires_grad_()
    do = O.conv(act(Xo), Wd, bd, rules, nc)
    uo = O.conv(act(do), Wu, bu, O.swap_rules(rules), n)
    _close(d.features, do, what="conv fwd"); _close(u.features, uo, what="deconv fwd")
    g = torch.randn(uo.shape, generator=torch.Generator().manual_seed(6))
    got = _grads(u.features, (x.features, down.weight, down.bias, up.weight, up.bias), g.to(gpu))
    exp = _grads(uo, (Xo, Wd, bd, Wu, bu), g)
    for a, e, name in zip(got, exp, ("dX", "dWd", "dbd", "dWu", "dbu")):
        _close(a, e, what=name)


@pytest.mark.parametrize("stride", [(3, 3, 3), (2, 2, 1), (1, 2, 3), (4, 4, 4), (3, 3, 3, "bf16")])
def test_general_stride_conv_deconv_and_pooling(gpu, stride):
    """VERDICT r3 missing 5: `get_downsampler(stride=...)` / `get_upsampler` hand ANY int or per-axis stride to
    scn.Convolution / scn.Deconvolution as filter_size = filter_stride (module_factory.py:221-258; the pooling factories
    :315-354 likewise); every shipped configuration uses 2.  Coarse sites by per-axis division, a child table of sx sy sz
    offsets (27 and fewer: the tile kernels; 4^3 = 64: the table-walk GEMM), Deconvolution back to the cached fine level:
    forward and every gradient of a Convolution -> Deconvolution pair and both poolings against the oracle, the Convolution
    also against the reference's own dense twin (module_factory.py:236-239: torch conv3d with kernel = stride on the
    zero-filled grid), a second request served from the cache, an indivisible size refused."""
    import torch.nn.functional as Fn
    bf16 = len(stride) == 4
    stride = tuple(stride[:3])
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=12, cin=8, grid=(24, 24, 12))
    n_off = stride[0] * stride[1] * stride[2]
    cout = 16
    down = scn.Convolution(3, 8, cout, stride, stride, True).to(gpu)
    up = scn.Deconvolution(3, cout, 8, stride, stride, True).to(gpu)
    assert tuple(down.weight.shape) == (n_off, 8, cout) and tuple(up.weight.shape) == (n_off, cout, 8)
    with torch.no_grad():
        down.bias.normal_(0, 0.5); up.bias.normal_(0, 0.5)
    xin = scn.CastFeatures(torch.bfloat16)(x) if bf16 else x
    d = scn.Sequential(scn.ReLU(), down)(xin)
    u = scn.Sequential(scn.ReLU(), up)(d)
    csize = tuple(int(s) // st for s, st in zip(size, stride))
    assert tuple(int(s) for s in d.spatial_size) == csize and tuple(int(s) for s in u.spatial_size) == tuple(int(s) for s in size)
    rb = O.strided_rulebook(scene.level_coords[0], stride)
    rules, n, nc = rb["rules"], scene.n(0), len(rb["coords"])
    assert d.features.shape == (nc, cout) and np.array_equal(d.get_spatial_locations().numpy()[:, :3], rb["coords"][:, :3])
    md = x.metadata
    srb = md.strided_rulebook(tuple(int(s) for s in size), stride)
    assert srb is md.strided_rulebook(tuple(int(s) for s in size), stride)                       # cached
    assert np.array_equal(srb.child.cpu().numpy(), rb["child"]) and np.array_equal(srb.parent.cpu().numpy(), rb["parent"])
    assert (srb.tiles is None) == (n_off > 27)
    q = O.bf16_storage if bf16 else (lambda t: t)
    Xo = O.input_layer_fwd(feats, scene.prow, n, 4).requires_grad_()
    Wd, bd = down.weight.detach().cpu().requires_grad_(), down.bias.detach().cpu().requires_grad_()
    Wu, bu = up.weight.detach().cpu().requires_grad_(), up.bias.detach().cpu().requires_grad_()
    wq = q if n_off <= 27 else (lambda t: t)                                 # (the tile kernel's weight image is bf16)
    do = q(O.conv(torch.relu(q(Xo)), wq(Wd), bd, rules, nc))
    uo = q(O.conv(torch.relu(do), Wu, bu, O.swap_rules(rules), n))
    tol = 2.0 ** -6 if bf16 else FEAT_TOL
    _close(d.features.float(), do, tol, "conv fwd"); _close(u.features.float(), uo, tol, "deconv fwd")
    if not bf16:
        # the reference's dense twin of the down-sampler: conv3d(kernel = stride = s) on the zero-filled grid, weight [o][i][c]
        dense = O.sparse_to_dense(torch.relu(Xo.detach()), scene.coords0, size.tolist(), 2)
        wt = Wd.detach().view(stride[0], stride[1], stride[2], 8, cout).permute(4, 3, 0, 1, 2).contiguous()
        dd = Fn.conv3d(dense, wt, bd.detach(), stride=stride)
        c = torch.from_numpy(rb["coords"])
        _close(d.features, dd[c[:, 3], :, c[:, 0], c[:, 1], c[:, 2]], what="dense twin of the convolution")
        g = torch.randn(uo.shape, generator=torch.Generator().manual_seed(6))
        got = _grads(u.features, (x.features, down.weight, down.bias, up.weight, up.bias), g.to(gpu))
        exp = _grads(uo, (Xo, Wd, bd, Wu, bu), g)
        for a, e, name in zip(got, exp, ("dX", "dWd", "dbd", "dWu", "dbu")):
            _close(a, e, what=name)
        for average in (False, True):
            pool = (scn.AveragePooling if average else scn.MaxPooling)(3, stride, stride)
            y = pool(x)
            Xp = O.input_layer_fwd(feats, scene.prow, n, 4).requires_grad_()
            yo = O.pool_fwd(Xp, rb["child"], average)
            _close(y.features, yo, 1e-6, "pool fwd")
            gp = torch.randn(yo.shape, generator=torch.Generator().manual_seed(4))
            (gx,) = torch.autograd.grad(y.features, x.features, gp.to(gpu), retain_graph=True)
            (ox,) = torch.autograd.grad(yo, Xp, gp)
            _close(gx, ox, 1e-6, "pool bwd")
    with pytest.raises(scn.ScnError):
        scn.Convolution(3, 8, 8, (5, 5, 5), (5, 5, 5), True).to(gpu)(x)                   # 24 is no multiple of 5
    with pytest.raises(NotImplementedError):
        scn.Convolution(3, 8, 8, (3, 3, 3), (2, 2, 2), True)                               # overlapping filters: not the reference's


@pytest.mark.parametrize("order", ["two_then_four", "four_then_two"])
def test_a_second_path_to_the_same_spatial_size_lands_in_the_existing_grid(gpu, order):
    """SparseConvNet keys a Metadata's grids by spatial size, so 24 -> 6 by ONE stride-4 Convolution after
    24 -> 12 -> 6 by two stride-2 layers (or the other way round) lands in the SAME grid and takes ITS row numbering
    (`scn_parent_lookup_div` + the child table against that numbering); round 4 raised "already holds a grid".  Checked
    against the oracle's rulebook built into the existing grid (forward, every gradient, the Deconvolution back), the two
    results live on one row order (their sum is a plain AddTable), and a grid that lacks a needed site is refused."""
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=13, cin=8, grid=(24, 24, 16))
    s0 = tuple(int(v) for v in size)
    c4 = scn.Convolution(3, 8, 16, (4, 4, 4), (4, 4, 4), True).to(gpu)
    c2a = scn.Convolution(3, 8, 8, (2, 2, 2), (2, 2, 2), True).to(gpu)
    c2b = scn.Convolution(3, 8, 16, (2, 2, 2), (2, 2, 2), True).to(gpu)
    up4 = scn.Deconvolution(3, 16, 8, (4, 4, 4), (4, 4, 4), True).to(gpu)
    with torch.no_grad():
        for m in (c4, c2a, c2b, up4):
            m.bias.normal_(0, 0.5)
    if order == "two_then_four":
        y2 = c2b(c2a(x)); y4 = c4(x)
    else:
        y4 = c4(x); y2 = c2b(c2a(x))
    assert y2.metadata is y4.metadata and tuple(int(v) for v in y4.spatial_size) == tuple(v // 4 for v in s0)
    # the oracle: whichever path came first numbers the 6 x 6 x 4 grid; the other is built INTO it
    r1 = O.strided_rulebook(scene.coords0, 2)
    if order == "two_then_four":
        r2 = O.strided_rulebook(r1["coords"], 2)
        r4 = O.strided_rulebook(scene.coords0, 4, existing=r2["coords"])
        grid = r2["coords"]
    else:
        r4 = O.strided_rulebook(scene.coords0, 4)
        r2 = O.strided_rulebook(r1["coords"], 2, existing=r4["coords"])
        grid = r4["coords"]
    nc = len(grid)
    assert np.array_equal(y4.get_spatial_locations().numpy(), grid) and np.array_equal(y2.get_spatial_locations().numpy(), grid)
    md = x.metadata
    assert np.array_equal(md.strided_rulebook(s0, (4, 4, 4)).child.cpu().numpy(), r4["child"])
    assert np.array_equal(md.strided_rulebook(tuple(v // 2 for v in s0)).child.cpu().numpy(), r2["child"])
    Xo = O.input_layer_fwd(feats, scene.prow, scene.n(0), 4).requires_grad_()
    P = {n: (m.weight.detach().cpu().requires_grad_(), m.bias.detach().cpu().requires_grad_())
         for n, m in (("c4", c4), ("c2a", c2a), ("c2b", c2b), ("up4", up4))}
    o4 = O.conv(Xo, *P["c4"], r4["rules"], nc)
    o2 = O.conv(O.conv(Xo, *P["c2a"], r1["rules"], len(r1["coords"])), *P["c2b"], r2["rules"], nc)
    _close(y4.features, o4, what="stride-4 path"); _close(y2.features, o2, what="two stride-2 layers")
    total = scn.AddTable()([y4, y2])                                   # one grid, one row order
    back = up4(total)
    ob = O.conv(o4 + o2, *P["up4"], O.swap_rules(r4["rules"]), scene.n(0))
    _close(back.features, ob, what="deconvolution back through the stride-4 rulebook")
    g = torch.randn(ob.shape, generator=torch.Generator().manual_seed(3))
    mods = (c4, c2a, c2b, up4)
    got = _grads(back.features, (x.features,) + tuple(m.weight for m in mods) + tuple(m.bias for m in mods), g.to(gpu))
    names = ("c4", "c2a", "c2b", "up4")
    exp = _grads(ob, (Xo,) + tuple(P[n][0] for n in names) + tuple(P[n][1] for n in names), g)
    for a, e, name in zip(got, exp, ("dX",) + tuple("dW " + n for n in names) + tuple("db " + n for n in names)):
        _close(a, e, what=name)
    # first-occurrence numbering composes (both paths number the sites alike), so the lookup is also pinned on a grid that is
    # numbered some OTHER way: the same sites planted in a permuted order
    from sparse_rcnn_amd.metadata import dedup
    scn3, coords3, feats3, fg3, x3, scene3, size3 = _input(gpu, seed=13, cin=8, grid=(24, 24, 16))
    shuffled = np.ascontiguousarray(grid[np.random.default_rng(1).permutation(nc)])
    x3.metadata.grids[tuple(v // 4 for v in s0)] = dedup(torch.from_numpy(shuffled).to(torch.int32).to(gpu).contiguous(), 0,
                                                         False, False)[0]
    y3 = c4(x3)
    r3 = O.strided_rulebook(scene3.coords0, 4, existing=shuffled)
    assert np.array_equal(y3.get_spatial_locations().numpy(), shuffled)
    assert np.array_equal(x3.metadata.strided_rulebook(s0, (4, 4, 4)).child.cpu().numpy(), r3["child"])
    assert not np.array_equal(r3["child"], r4["child"])
    _close(y3.features, O.conv(Xo.detach(), P["c4"][0].detach(), P["c4"][1].detach(), r3["rules"], nc), what="planted grid")
    # a grid of that size which lacks sites: refused, not grown (planted here: the first five coarse sites only)
    scn2, coords2, feats2, fg2, x2, scene2, size2 = _input(gpu, seed=14, cin=8, grid=(24, 24, 16))
    few = torch.from_numpy(np.ascontiguousarray(O.strided_rulebook(scene2.coords0, 4)["coords"][:5])).to(torch.int32).to(gpu)
    x2.metadata.grids[tuple(v // 4 for v in s0)] = dedup(few.contiguous(), 0, False, False)[0]
    with pytest.raises(scn.ScnError, match="lacks"):
        c4(x2)


def _topk_case(name):
    g = torch.Generator().manual_seed(len(name))
    if name == "rpn_field":                       # the detection step's shape: 1 x 524 288 sigmoid scores, 1024 kept
        return torch.sigmoid(torch.randn(1, 524288, generator=g) * 2 - 3), 1024
    if name == "constant_background":             # most anchors carry ONE score; the threshold lies above it
        s = torch.full((2, 200000), 0.0474)
        s[:, torch.randperm(200000, generator=g)[:5000]] = torch.rand(5000, generator=g)
        return s, 1024
    if name == "threshold_is_the_constant":       # ... and below it: > 6144 ties in the threshold's bucket (the slow path)
        s = torch.full((2, 150000), 0.25)
        s[:, torch.randperm(150000, generator=g)[:700]] = 0.5 + torch.rand(700, generator=g) / 2
        s[1, 77:140000:3] = 0.1
        return s, 2048
    if name == "ragged_batch":
        return torch.randn(3, 10007, generator=g), 100
    if name == "everything":
        return torch.randn(2, 1500, generator=g), 1500
    if name == "one":
        return torch.randn(1, 1, generator=g), 1
    if name == "few_values":                      # 50 distinct values: every bucket is a tie
        return torch.randint(0, 50, (2, 5000), generator=g).float() - 20, 300
    if name == "specials":
        s = torch.randn(2, 4096, generator=g)
        s[0, 5] = s[0, 700] = s[1, 9] = float("nan")
        s[0, 6] = s[1, 10] = float("inf"); s[0, 7] = float("-inf")
        s[1, 100:140] = 0.0; s[1, 110:120] = -0.0
        return s, 2048
    if name == "all_equal":
        return torch.zeros(2, 30000), 257
    raise KeyError(name)


@pytest.mark.parametrize("case", ["rpn_field", "constant_background", "threshold_is_the_constant", "ragged_batch", "everything",
                                  "one", "few_values", "specials", "all_equal"])
def test_radix_select_topk_equals_a_stable_descending_sort(gpu, case):
    """`scn_topk_boxes` (ProposalSelector's torch.topk + rpn_bbox[batch, indices], proposal_selector.py:60-75) against the
    first k entries of torch's STABLE descending sort on the CPU: same values bit for bit, same indices (equal scores by
    ascending index -- torch.topk itself leaves that order open; NaN on top and -0 == +0 as in torch), the boxes of those
    indices; against torch.topk on the device the values are equal too.  Twice in a row (the state goes back to zero), and
    the gradients of score and boxes equal those of the torch calls."""
    from sparse_rcnn_amd import proposals as PR
    score, k = _topk_case(case)
    b, n = score.shape
    boxes = torch.randn(b, n, 2, 3, generator=torch.Generator().manual_seed(3))
    sd, bd = score.to(gpu).requires_grad_(), boxes.to(gpu).requires_grad_()
    exp_v, exp_i = torch.sort(score, dim=1, descending=True, stable=True)
    exp_v, exp_i = exp_v[:, :k], exp_i[:, :k]
    for _ in range(2):
        v, i, bx = PR.topk_boxes(sd, bd, k)
        assert i.dtype == torch.int64 and tuple(bx.shape) == (b, k, 2, 3)
        assert torch.equal(i.cpu(), exp_i), case
        assert torch.equal(v.detach().cpu().view(torch.int32), score.gather(1, exp_i).view(torch.int32))
        assert torch.equal(bx.detach().cpu(), boxes[torch.arange(b).unsqueeze(1), exp_i])
    tv, _ = torch.topk(sd.detach(), k, dim=1, sorted=True)
    assert torch.equal(torch.nan_to_num(tv, nan=7.0), torch.nan_to_num(v.detach(), nan=7.0))
    gv = torch.randn(b, k, generator=torch.Generator().manual_seed(4)).to(gpu)
    gb = torch.randn(b, k, 2, 3, generator=torch.Generator().manual_seed(5)).to(gpu)
    ds, db = torch.autograd.grad([v, bx], [sd, bd], [gv, gb])
    with PR_switch(PR, "TORCH_TOPK", True):
        v2, i2, bx2 = PR.topk_boxes(sd, bd, k)
        assert not isinstance(v2.grad_fn, type(v.grad_fn))
        # (torch's own tie order may differ: scatter the gradients through OUR indices for the comparison)
    ds2 = torch.zeros_like(sd).scatter_(1, i, gv)
    db2 = torch.zeros_like(bd)
    db2[torch.arange(b, device=gpu).unsqueeze(1), i] = gb
    assert torch.equal(ds, ds2) and torch.equal(db, db2)


class PR_switch:
    def __init__(self, mod, name, value):
        self.mod, self.name, self.value = mod, name, value

    def __enter__(self):
        self.old = getattr(self.mod, self.name); setattr(self.mod, self.name, self.value)

    def __exit__(self, *a):
        setattr(self.mod, self.name, self.old)


def test_cell_map_of_a_level(gpu):
    """`scn_cell_map`: the cell of every active row in the channels-last volume [B X Y Z] and the inverse map, against the
    integer formula; a row outside the volume is counted, gets cell 0 and no map entry."""
    from sparse_rcnn_amd import _lib as L
    rng = np.random.default_rng(5)
    B, X, Y, Z = 3, 7, 5, 9
    lin = rng.choice(B * X * Y * Z, size=400, replace=False)
    b, x, y, z = np.unravel_index(lin, (B, X, Y, Z))
    coords = np.stack([x, y, z, b], 1).astype(np.int32)
    coords[17] = (X, 0, 0, 0)                                            # outside
    cd = torch.from_numpy(coords).to(gpu)
    ridx = torch.empty(400, dtype=torch.int64, device=gpu)
    cmap = torch.empty(B * X * Y * Z, dtype=torch.int32, device=gpu)
    flag = torch.empty(1, dtype=torch.int32, device=gpu)
    hs = L.host_i64(3)
    hs[0], hs[1], hs[2] = X, Y, Z
    L.check(L.lib().scn_cell_map(L.ptr(cd), 400, B, hs, L.ptr(ridx), L.ptr(cmap), L.ptr(flag), L.stream()))
    exp = lin.astype(np.int64).copy(); exp[17] = 0
    assert int(flag.item()) == 1 and np.array_equal(ridx.cpu().numpy(), exp)
    em = np.full(B * X * Y * Z, -1, np.int32)
    keep = np.arange(400) != 17
    em[lin[keep]] = np.arange(400, dtype=np.int32)[keep]
    assert np.array_equal(cmap.cpu().numpy(), em)


def test_deconvolution_needs_cached_level(gpu):
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=4, cin=4)
    with pytest.raises(scn.ScnError):
        scn.Deconvolution(3, 4, 4, (2, 2, 2), (2, 2, 2), True).to(gpu)(x)


@pytest.mark.parametrize("cin,cout", [(64, 32), (46, 23), (7, 5)])
def test_nin_relu_add_join(gpu, cin, cout):
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=5, cin=cin)
    nin = scn.NetworkInNetwork(cin, cout, True).to(gpu)
    net = scn.Sequential(scn.ConcatTable(scn.Identity(), scn.ReLU()), scn.AddTable())      # x + relu(x)
    y = nin(net(x)).features
    j = scn.JoinTable()([x, x]).features
    n = scene.n(0)
    Xo = O.input_layer_fwd(feats, scene.prow, n, 4).requires_grad_()
    W, b = nin.weight.detach().cpu().requires_grad_(), nin.bias.detach().cpu().requires_grad_()
    yo = (Xo + torch.relu(Xo)) @ W + b
    _close(y, yo, what="fwd")
    assert torch.equal(j.detach().cpu(), torch.cat([Xo, Xo], 1).detach())
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(7))
    for a, e, name in zip(_grads(y, (x.features, nin.weight, nin.bias), g.to(gpu)), _grads(yo, (Xo, W, b), g),
                          ("dX", "dW", "db")):
        _close(a, e, what=name)


@pytest.mark.parametrize("leak", [0.0, 0.2])
@pytest.mark.parametrize("training", [True, False])
def test_batchnorm_relu(gpu, leak, training):
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=6, cin=24)
    bn = (scn.BatchNormLeakyReLU(24, 1e-4, 0.9, leak) if leak else scn.BatchNormReLU(24, 1e-4, 0.9)).to(gpu)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
    bn.train(training)
    rm0, rv0 = bn.running_mean.detach().cpu().clone(), bn.running_var.detach().cpu().clone()
    y = bn(x).features
    n = scene.n(0)
    Xo = O.input_layer_fwd(feats, scene.prow, n, 4).requires_grad_()
    ga, be = bn.weight.detach().cpu().requires_grad_(), bn.bias.detach().cpu().requires_grad_()
    yo = O.batchnorm_relu_fwd(Xo, ga, be, rm0, rv0, 1e-4, 0.9, leak, training)
    _close(y, yo, what="fwd")
    _close(bn.running_mean, rm0, 1e-6, "running_mean (retain-fraction momentum)")
    _close(bn.running_var, rv0, 1e-6, "running_var")
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(8))
    for a, e, name in zip(_grads(y, (x.features, bn.weight, bn.bias), g.to(gpu)), _grads(yo, (Xo, ga, be), g),
                          ("dX", "dgamma", "dbeta")):
        _close(a, e, what=name)


def test_sparse_to_dense(gpu):
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=7, cin=6)
    d = scn.SparseToDense(3, 6)(x)
    Xo = O.input_layer_fwd(feats, scene.prow, scene.n(0), 4)
    exp = O.sparse_to_dense(Xo, scene.coords0, size.tolist(), 2)
    assert torch.equal(d.detach().cpu(), exp)
    g = torch.randn(exp.shape, generator=torch.Generator().manual_seed(2))
    (gx,) = torch.autograd.grad(d, x.features, g.to(gpu))
    c = torch.from_numpy(scene.coords0)
    assert torch.equal(gx.cpu(), g[c[:, 3], :, c[:, 0], c[:, 1], c[:, 2]])


# ---------------------------------------------------------------------------------------- the U-Net (A12)
def test_unet_forward_backward_matches_oracle(gpu):
    from sparse_rcnn_amd.unet import Backbone
    channels = (16, 24, 32)
    coords, size, batch = _cloud(11, grid=(32, 32, 16), n=3000, batch=2, dup=300)
    feats = torch.randn(len(coords), 7, generator=torch.Generator().manual_seed(3))
    params = O.init_unet_params(7, channels, seed=4)
    net = Backbone(7, channels).to(gpu)
    net.unet.load_oracle_params(params)
    out = net(coords, feats.to(gpu), size, batch)
    scene = O.OracleScene(coords.numpy())
    po = {k: v.clone().requires_grad_() for k, v in params.items()}
    exp = O.unet_forward(scene, feats, po, channels)
    _close(out.features, exp, what="unet fwd")
    g = torch.randn(exp.shape, generator=torch.Generator().manual_seed(5))
    out.features.backward(g.to(gpu))
    exp.backward(g)
    for k, p in net.unet.named_oracle_params().items():
        _close(p.grad, po[k].grad.view_as(p), 2e-4, f"grad {k}")


# ---------------------------------------------------------------------------------------- ROI crop (A11)
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "roi_crop_*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_roi_crop_matches_reference_golden(gpu, path):
    from sparse_rcnn_amd import roi
    z = np.load(path)
    n_pts = int(z["n_pts"])
    inside = np.unpackbits(z["is_inside"], axis=1)[:, :n_pts].astype(bool)
    counts = z["box_counts"].tolist()
    bbox_batch = [torch.from_numpy(b) for b in np.split(z["boxes"], np.cumsum(counts)[:-1])]
    resize = z["resize"].tolist() if len(z["resize"]) else None
    boxes, got_counts, assoc = roi.transform_boxes(bbox_batch, z["spatial_size"], bool(z["clip"]), resize)
    exp_boxes = np.concatenate([z["bbox_tensor"][:, 0], z["assoc"][:, None], z["bbox_tensor"][:, 1],
                                z["assoc"][:, None] + 1], 1)
    assert np.array_equal(boxes.cpu().numpy(), exp_boxes) and got_counts == counts
    # reference-signature entry point
    new_coords, new_feats, is_inside = roi.roi_cut(torch.from_numpy(z["coords"]), torch.from_numpy(z["feats"]).to(gpu),
                                                   torch.from_numpy(z["bbox_tensor"]), torch.from_numpy(z["assoc"]))
    assert new_coords.dtype == torch.int64 and not new_coords.is_cuda and is_inside.dtype == torch.bool
    assert np.array_equal(new_coords.numpy(), z["out_coords"])
    assert np.array_equal(new_feats.cpu().numpy(), z["out_feats"])
    assert np.array_equal(is_inside.numpy(), inside)
    # module level, constructed exactly as the reference does (roi_select_sparse.py:29-52, model.py:577-580,625): raw
    # combiner, then SparseRoiExtraCut of a second feature map with (a) the list selection, (b) the reference's bool matrix
    splits = z["splits"].tolist()
    scene = (torch.from_numpy(z["coords"]), torch.from_numpy(z["feats"]).to(gpu), torch.from_numpy(z["spatial_size"]),
             len(splits), splits)
    kw = dict(clip_boxes=bool(z["clip"]), resize_boxes=resize)
    for dense in (False, True):
        cut = roi.SparseRoiCut(roi.RawToRawFeatureExtractorCombiner(), **kw, dense_inside=dense)
        (m_coords, m_feats, m_size, m_bs), selection = cut(scene, bbox_batch)
        assert np.array_equal(m_coords.cpu().numpy(), z["out_coords"]) and m_bs == len(inside)
        assert np.array_equal(m_feats.cpu().numpy(), z["out_feats"])
        assert selection[1] == counts and selection[2] == splits
        if dense:
            assert np.array_equal(selection[0].numpy(), inside)
        else:
            assert isinstance(selection[0], roi.RoiSelection) and selection[0].prefix[-1] == len(z["out_coords"])
        extra = roi.SparseRoiExtraCut(roi.RawToFeaturesSceneFeatureExtractorCombiner())(
            (scene[0], torch.from_numpy(z["extra_in"]).to(gpu)) + scene[2:], selection)
        assert np.array_equal(extra.cpu().numpy(), z["extra_out"])


MASK_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mask_epilogue_*.npz")))


@pytest.mark.parametrize("path", MASK_GOLDEN, ids=[os.path.basename(p) for p in MASK_GOLDEN])
def test_mask_epilogue_matches_reference_golden(gpu, path):
    """SparseMaskPredictor / SparseMaskLossSelector from the CSR selection on the device, against the
    outputs of the reference's own classes (tests/golden/make_mask_golden.py) and the oracle."""
    from sparse_rcnn_amd import roi
    from test_oracle_golden import load_mask
    d = load_mask(path)
    z = np.load(os.path.join(os.path.dirname(path), f"roi_crop_{str(d['roi_case'])}.npz"))
    counts, splits = d["box_counts"].tolist(), d["batch_splits"].tolist()
    bt = torch.from_numpy(z["bbox_tensor"]).to(gpu).to(torch.int32)
    sa = torch.from_numpy(z["assoc"]).to(gpu).to(torch.int32).reshape(-1, 1)
    boxes = torch.cat([bt[:, 0], sa, bt[:, 1], sa + 1], 1).contiguous()
    _, _, sel = roi.roi_cut_device(torch.from_numpy(z["coords"]), torch.from_numpy(z["feats"]).to(gpu), boxes)
    scores = torch.from_numpy(d["scores"]).to(gpu).requires_grad_()
    pred = roi.mask_predict(scores, sel, counts, splits, torch.from_numpy(d["classes"]), int(d["num_valid"]))
    got = torch.cat([p.reshape(-1) for p in pred]).cpu().numpy()
    assert [tuple(p.shape) for p in pred] == [(c, s) for c, s in zip(counts, splits)]
    assert np.allclose(got, d["pred_masks"], rtol=0, atol=1e-6)
    assert np.array_equal(got == 0, d["pred_masks"] == 0)                     # zeros exactly where the reference has them
    p, g, rows, labels = roi.mask_loss_select(scores, sel, counts, splits, d["keep_list"], d["assoc_list"],
                                              d["labels_list"], d["masks_list"])
    assert np.array_equal(p.detach().cpu().numpy(), d["loss_pred"]) and np.array_equal(g.cpu().numpy(), d["loss_gt"])
    assert rows == d["loss_rows"].tolist() and np.array_equal(labels.numpy(), d["loss_labels"])
    # gradient of the selected scores: one-hot of the box label on the rows of kept boxes
    w = torch.randn(p.shape, generator=torch.Generator().manual_seed(2)).to(gpu)
    (ds,) = torch.autograd.grad(p, scores, w)
    exp = np.zeros(d["scores"].shape, np.float32)
    kept_boxes = np.nonzero(np.concatenate(d["keep_list"]))[0]
    starts = np.asarray(sel.prefix)
    r = 0
    for b, lab in zip(kept_boxes, d["loss_labels"]):
        n = starts[b + 1] - starts[b]
        exp[starts[b]:starts[b + 1], lab] = w.cpu().numpy()[r:r + n]
        r += n
    assert np.array_equal(ds.cpu().numpy(), exp)


NMS_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "nms_*.npz")))


@pytest.mark.parametrize("path", NMS_GOLDEN, ids=[os.path.basename(p) for p in NMS_GOLDEN])
def test_nms_and_proposal_selector_match_reference_golden(gpu, path):
    """one-launch greedy NMS (scn_nms) and the ProposalSelector mirror against the outputs of the
    reference's own non_maximum_supression / ProposalSelector -- keep decisions bit-exact."""
    from sparse_rcnn_amd.proposals import ProposalSelector, non_maximum_suppression
    z = np.load(path)
    thr = float(z["thr"])
    keep = non_maximum_suppression(torch.from_numpy(z["sorted_boxes"]).to(gpu), thr)
    assert keep.dtype == torch.bool and np.array_equal(keep.cpu().numpy(), z["keep"])
    s, b, i = ProposalSelector(int(z["pre"]), int(z["post"]), thr)(torch.from_numpy(z["score"]).to(gpu),
                                                                  torch.from_numpy(z["boxes"]).to(gpu))
    assert [len(x) for x in s] == z["out_counts"].tolist()
    assert np.array_equal(torch.cat(s).cpu().numpy(), z["out_scores"])
    assert np.array_equal(torch.cat(b).cpu().numpy(), z["out_boxes"])
    assert np.array_equal(torch.cat(i).numpy(), z["out_index"])


def test_rpn_boundary_chain_sparse_to_dense_to_proposals(gpu):
    """SURVEY §8f N3 assembled as ONE boundary (module_factory.py:581-611 dilation network on a SparseToDense'd level;
    anchor_network.py:127-219 1x1 heads; proposal_selector.py:60-89): sparse level features -> scn.SparseToDense ->
    dense 3^3 conv stack + ReLU (torch / MIOpen: outside the hot path) -> per-anchor score and box heads ->
    ProposalSelector (top-k + one-launch NMS) -> gradient of the kept scores back into the sparse features.
    SparseToDense must equal the oracle bit for bit; the dense stack agrees with the same layers on the CPU; the
    selection made on the device's own scores equals the oracle's top-k + greedy NMS on those scores (keep decisions and
    indices bit-exact); the gradient that reaches the sparse rows equals the CPU chain's."""
    from sparse_rcnn_amd.proposals import ProposalSelector
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=51, cin=12, grid=(24, 24, 16), n=2500, batch=2, dup=200)
    n0 = scene.n(0)
    C, A = 12, 3
    torch.manual_seed(5)
    stack = torch.nn.Sequential(torch.nn.Conv3d(C, 16, 3, padding=1), torch.nn.ReLU(),
                                torch.nn.Conv3d(16, 16, 3, padding=1), torch.nn.ReLU())
    score_head, box_head = torch.nn.Conv3d(16, A, 1), torch.nn.Conv3d(16, A * 6, 1)
    sel = ProposalSelector(128, 32, 0.3)

    import copy
    nets = {"cpu": (stack, score_head, box_head),
            "cuda": tuple(copy.deepcopy(m).to(gpu) for m in (stack, score_head, box_head))}

    def chain(xfeat, dense, dev):
        st, sh, bh = nets[torch.device(dev).type]
        h = st(dense)
        score = sh(h).flatten(1)                                                   # [B, A X Y Z]
        delta = bh(h).view(2, A, 6, -1).permute(0, 1, 3, 2).reshape(2, -1, 6)
        g = torch.stack(torch.meshgrid(*[torch.arange(int(s), dtype=torch.float32) for s in size], indexing="ij"), -1)
        ctr = g.reshape(1, 1, -1, 3).expand(2, A, -1, 3).reshape(2, -1, 3).to(dev)
        half = 2.0 + torch.nn.functional.softplus(delta[..., 3:])
        box = torch.stack([ctr + delta[..., :3] - half, ctr + delta[..., :3] + half], -2)          # [B, N, 2, 3]
        return score, box

    # device
    Xg = x.features.detach().clone().requires_grad_()
    xt = scn.SparseConvNetTensor(features=Xg, metadata=x.metadata, spatial_size=x.spatial_size)
    dense_g = scn.SparseToDense(3, C)(xt)
    score_g, box_g = chain(Xg, dense_g, gpu)
    s_list, b_list, i_list = sel(score_g, box_g)
    # oracle pieces
    Xo = x.features.detach().cpu().clone().requires_grad_()
    dense_o = O.sparse_to_dense(Xo, scene.coords0, size.tolist(), 2)
    assert torch.equal(dense_g.detach().cpu(), dense_o.detach())                   # A13 bit-exact
    score_o, box_o = chain(Xo, dense_o, torch.device("cpu"))
    _close(score_g, score_o, 1e-4, "dense stack scores (MIOpen vs CPU)")
    _close(box_g, box_o, 1e-4, "dense stack boxes")
    sg, bg = score_g.detach().cpu(), box_g.detach().cpu()
    for b in range(2):                                                             # selection on the device's own scores
        top, idx = torch.topk(sg[b], 128, sorted=True)
        keep = torch.from_numpy(O.nms(bg[b][idx].numpy(), 0.3))
        assert torch.equal(i_list[b], idx[keep][:32]) and torch.equal(s_list[b].detach().cpu(), top[keep][:32])
        assert torch.equal(b_list[b].detach().cpu(), bg[b][idx][keep][:32])
    # gradient of the kept scores back to the sparse rows: the same selection applied to the CPU chain
    torch.cat(s_list).sum().backward()
    sum(score_o[b][i_list[b]].sum() for b in range(2)).backward()
    _close(Xg.grad, Xo.grad, 1e-4, "d kept scores / d sparse features")
    assert Xg.grad.shape == (n0, C)


def test_nms_edge_cases(gpu):
    from sparse_rcnn_amd.proposals import non_maximum_suppression
    e = non_maximum_suppression(torch.zeros(2, 0, 2, 3, device=gpu), 0.5)
    assert e.shape == (2, 0)
    one = non_maximum_suppression(torch.tensor([[[[0., 0, 0], [1, 1, 1]]]], device=gpu), 0.5)
    assert one.tolist() == [[True]]
    rng = np.random.default_rng(0)
    n = 2500                                                          # more boxes than threads: several per thread
    c = rng.uniform(0, 30, size=(n, 3)); sz = rng.uniform(1, 6, size=(n, 3))
    boxes = np.stack([c - sz, c + sz], 1).astype(np.float32)
    got = non_maximum_suppression(torch.from_numpy(boxes).to(gpu)[None], 0.25)[0].cpu().numpy()
    assert np.array_equal(got, O.nms(boxes, 0.25))


@pytest.mark.parametrize("n,batch,thr", [(1, 1, 0.5), (63, 2, 0.3), (64, 1, 0.5), (65, 3, 0.5), (1000, 2, 0.25), (1024, 8, 0.5),
                                          (1025, 1, 0.7), (2500, 2, 0.25), (4096, 1, 0.4)])
def test_bit_matrix_nms_equals_the_serial_kernel_and_the_oracle(gpu, n, batch, thr):
    """scn_nms_bits (round 5: suppression bit matrix over the chip + one serial walk out of LDS) against scn_nms (one workgroup
    walking the boxes) and the oracle's greedy NMS: keep decisions bit for bit, also with degenerate (zero-volume, NaN IoU)
    and duplicate boxes, sizes around the 64-box word and 256-row staging boundaries."""
    from sparse_rcnn_amd import proposals as P
    rng = np.random.default_rng(n * 7 + batch)
    c = rng.uniform(0, 40, size=(batch, n, 3)); sz = rng.uniform(1, 8, size=(batch, n, 3))
    boxes = np.stack([c - sz, c + sz], 2).astype(np.float32)
    if n >= 64:
        boxes[:, 5] = boxes[:, 3]                               # a duplicate
        boxes[:, 7, 1] = boxes[:, 7, 0]                         # zero volume
        boxes[:, 9] = 0.0                                       # zero box: 0 / 0 against its copies
        boxes[:, 11] = 0.0
    b = torch.from_numpy(boxes).to(gpu)
    got = P.non_maximum_suppression(b, thr).cpu().numpy()
    P.SERIAL_NMS = True
    try:
        ref = P.non_maximum_suppression(b, thr).cpu().numpy()
    finally:
        P.SERIAL_NMS = False
    assert np.array_equal(got, ref)
    for s_ in range(batch):
        assert np.array_equal(got[s_], O.nms(boxes[s_], thr)), s_


VOX_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "voxelize_*.npz")))


@pytest.mark.parametrize("path", VOX_GOLDEN, ids=[os.path.basename(p) for p in VOX_GOLDEN])
def test_voxelisation_matches_reference_golden(gpu, path):
    """augment_coords on the device against the outputs of the reference's own augment_coords."""
    from sparse_rcnn_amd import voxelize
    from test_oracle_golden import vox_args
    z = np.load(path)
    size, shift = vox_args(z)
    rows, inside, out_size, cshift = voxelize.augment_coords(
        torch.from_numpy(z["coords"]).to(gpu), rot_and_scale=z["rot_and_scale"], sub_pixel_offset=z["offset"],
        spatial_size=size, shift=shift, batch_index=3)
    assert rows.dtype == torch.int64 and rows.is_cuda
    assert np.array_equal(rows[:, :3].cpu().numpy(), z["out_coords"]) and (rows[:, 3] == 3).all()
    assert np.array_equal(inside.cpu().numpy(), z["is_inside"])
    assert np.array_equal(out_size.numpy(), z["out_size"]) and np.array_equal(cshift.numpy(), z["out_shift"])


def test_voxelisation_cut_out_with_drawn_start_positions(gpu):
    """The cut-out of random_cut_out for given start positions (the reference's own function raises on torch 2.10, so
    this mode is pinned by the oracle only); the rows feed straight into InputLayer."""
    from sparse_rcnn_amd import voxelize
    scn = _scn()
    z = np.load(VOX_GOLDEN[0])
    start, size = (17, -4, 9), (96, 128, 48)
    rows, inside, out_size, cshift = voxelize.augment_coords(
        torch.from_numpy(z["coords"]).to(gpu), rot_and_scale=z["rot_and_scale"], sub_pixel_offset=z["offset"],
        spatial_size=size, start_positions=start)
    res, oin, osize, oshift = O.augment_coords(z["coords"], z["rot_and_scale"], z["offset"], size, None, start)
    assert np.array_equal(rows[:, :3].cpu().numpy(), res) and np.array_equal(inside.cpu().numpy(), oin)
    assert np.array_equal(cshift.numpy(), oshift) and out_size.tolist() == list(size)
    batch = voxelize.collate_coords([rows, rows.clone()])
    x = scn.InputLayer(3, out_size, mode=4)((batch, torch.ones(len(batch), 2, device=gpu), 1))
    assert x.features.shape[0] == len(np.unique(res, axis=0))


def test_roi_cut_module_revoxelises_like_oracle(gpu):
    from sparse_rcnn_amd import roi
    from sparse_rcnn_amd.synthetic import make_boxes
    coords, size, batch = _cloud(21, grid=(48, 40, 24), n=4000, batch=2, dup=500)
    feats = torch.randn(len(coords), 23, generator=torch.Generator().manual_seed(1))
    bbox_batch = make_boxes(coords, n_boxes=9, seed=2, lo=3, hi=30)
    fg = feats.to(gpu).requires_grad_()
    cut = roi.SparseRoiCut(roi.RawToTensorFeatureExtractorCombiner())         # as model.py:577 constructs it
    out, (is_inside, counts, splits) = cut((coords, fg, size + 32, batch, [0]), bbox_batch)   # model.py:585
    bi, cnt, assoc = O.transform_boxes([b.numpy() for b in bbox_batch])
    src, box_of, inside = O.roi_crop(coords.numpy(), bi, assoc)
    assert np.array_equal(is_inside.numpy(), inside) and counts == cnt
    new_coords = np.concatenate([coords.numpy()[src][:, :3], box_of[:, None]], 1)
    ac, prow, _ = O.input_layer_rules(new_coords)
    assert np.array_equal(out.get_spatial_locations().numpy(), ac)
    assert out.batch_size() == 18 and tuple(out.spatial_size.tolist()) == tuple(s + 32 for s in size.tolist())
    exp = O.input_layer_fwd(feats[torch.from_numpy(src)], prow, len(ac), 4)
    assert torch.equal(out.features.detach().cpu(), exp)
    g = torch.randn(exp.shape, generator=torch.Generator().manual_seed(3))
    out.features.backward(g.to(gpu))
    gsel = O.input_layer_bwd(g, prow, 4)
    gexp = torch.zeros(len(coords), 23, dtype=torch.float64).index_add_(0, torch.from_numpy(src), gsel.double())
    _close(fg.grad, gexp.float(), 1e-6, "roi crop backward")


def test_roi_no_boxes(gpu):
    from sparse_rcnn_amd import roi
    coords, size, batch = _cloud(22, n=200, dup=0)
    nc, nf, inside = roi.roi_cut(coords, torch.ones(len(coords), 3).to(gpu), torch.zeros(0, 2, 3, dtype=torch.long),
                                 torch.zeros(0, dtype=torch.long))
    assert nc.shape == (0, 4) and nf.shape == (0, 3) and inside.shape == (0, len(coords))


def test_index_prefetch_on_side_stream_is_equivalent(gpu):
    """Metadata.prepare_async (index structures built one batch ahead on the index stream) gives the same tensors."""
    from sparse_rcnn_amd.unet import Backbone
    coords, size, batch = _cloud(31, grid=(32, 32, 16), n=2500, batch=2, dup=200)
    feats = torch.randn(len(coords), 7, generator=torch.Generator().manual_seed(3)).to(gpu)
    net = Backbone(7, (16, 24, 32)).to(gpu)
    ref = net(coords, feats, size, batch).features
    g = torch.randn_like(ref)
    gr = torch.autograd.grad(ref, list(net.parameters()), g)
    for _ in range(2):
        md = net.prefetch(coords.to(gpu), size, batch)
        out = net(coords.to(gpu), feats, size, batch, metadata=md).features
        gp = torch.autograd.grad(out, list(net.parameters()), g)
        assert torch.equal(out, ref)
        for a, b in zip(gp, gr):
            assert torch.equal(a, b)
    with pytest.raises(_scn().ScnError):
        net(coords[:-1].to(gpu), feats[:-1], size, batch, metadata=net.prefetch(coords.to(gpu), size, batch))
    # the same build driven by a helper thread (bench.py --prefetch)
    pending = net.prefetch_in_thread(coords.to(gpu), size, batch)
    out = net(coords.to(gpu), feats, size, batch, metadata=pending.result()).features
    assert torch.equal(out, ref)


def test_native_pyramid_build_is_bit_identical(gpu):
    """scn_pyramid_build (one C call, one workspace) against the step-by-step build driven from Python: every index
    tensor of every level equal, rule lists included; and a network run on either gives the same bits."""
    from sparse_rcnn_amd.metadata import Metadata
    from sparse_rcnn_amd.unet import Backbone
    coords, size, batch = _cloud(41, grid=(32, 32, 16), n=2600, batch=2, dup=300)
    cg = coords.to(gpu)
    a = Metadata(3); a.set_input(size, cg, batch, 4); a.build_pyramid(size, 3, 3)
    b = Metadata(3).build_native(size, cg, batch, 4, 3, 3)
    eq = lambda x, y: torch.equal(x, y)
    assert eq(a.item_row, b.item_row) and eq(a.row_count, b.row_count) and eq(a.row_first, b.row_first)
    assert a.n_samples == b.n_samples and list(a.grids) == list(b.grids)
    for s in a.grids:
        assert a.grids[s].n == b.grids[s].n and eq(a.grids[s].coords, b.grids[s].coords)
    for key in a.subm:
        ra, rb = a.subm[key], b.subm[key]
        assert eq(ra.table, rb.table) and ra.rules.prefix_list() == rb.rules.prefix_list()
        assert eq(ra.rules.in_rows, rb.rules.in_rows) and eq(ra.rules.out_rows, rb.rules.out_rows)
        for f in ("perm", "tstab", "tile_mask", "tile_order"):
            assert eq(getattr(ra.tiles, f), getattr(rb.tiles, f)), f
    for key in a.strided:
        ra, rb = a.strided[key], b.strided[key]
        assert eq(ra.parent, rb.parent) and eq(ra.fine_off, rb.fine_off) and eq(ra.child, rb.child)
        assert ra.rules.prefix_list() == rb.rules.prefix_list() and eq(ra.rules.in_rows, rb.rules.in_rows)
        for f in ("perm", "tstab", "tile_mask", "tile_order"):
            assert eq(getattr(ra.tiles, f), getattr(rb.tiles, f)), f
    feats = torch.randn(len(coords), 7, generator=torch.Generator().manual_seed(3)).to(gpu)
    net = Backbone(7, (16, 24, 32)).to(gpu)
    net.NATIVE_INDEX = False
    ref = net(cg, feats, size, batch).features
    net.NATIVE_INDEX = True
    assert torch.equal(net(cg, feats, size, batch).features, ref)
    with pytest.raises(_scn().ScnError):
        Metadata(3).build_native(torch.tensor([30, 32, 16]), cg, batch, 4, 3, 3)      # 30/2 = 15 is odd


def _index_tensors(md, with_x=False):
    """Every deterministic index structure of a Metadata (the hash tables' slot layout is not: insertion races, capacity)."""
    out = [("item_row", md.item_row), ("row_count", md.row_count), ("row_first", md.row_first), ("point_coords", md.point_coords)]
    for s, g in md.grids.items():
        out.append((f"coords{s}", g.coords))
    for key, rb in md.subm.items():
        if rb.table is None:
            continue
        nt = rb.tiles.perm.numel() // 16
        order = rb.tiles.tile_order
        if with_x:                       # the XCD-local order and its nine bin starts sit behind the first order
            base = order.untyped_storage()
            order = torch.empty(0, dtype=torch.int32, device=order.device).set_(base, order.storage_offset(), (2 * nt + 9,))
        out += [(f"table{key}", rb.table), (f"in{key}", rb.rules.in_rows), (f"out{key}", rb.rules.out_rows),
                (f"prefix{key}", rb.rules.prefix_dev), (f"perm{key}", rb.tiles.perm), (f"tstab{key}", rb.tiles.tstab),
                (f"tmask{key}", rb.tiles.tile_mask), (f"torder{key}", order),
                (f"prefix_host{key}", torch.tensor(rb.rules.prefix_list()))]
    for key, sb in md.strided.items():
        out += [(f"parent{key}", sb.parent), (f"fine_off{key}", sb.fine_off), (f"child{key}", sb.child),
                (f"cin{key}", sb.rules.in_rows), (f"cout{key}", sb.rules.out_rows), (f"cprefix{key}", sb.rules.prefix_dev),
                (f"cperm{key}", sb.tiles.perm), (f"ctstab{key}", sb.tiles.tstab), (f"ctmask{key}", sb.tiles.tile_mask),
                (f"ctorder{key}", sb.tiles.tile_order), (f"cprefix_host{key}", torch.tensor(sb.rules.prefix_list()))]
    return out


@pytest.mark.parametrize("case", ["150k-4", "12crops-6", "sparse-overflow", "edges"])
def test_brick_tables_of_the_fused_build_change_no_structure(gpu, case):
    """Round 5: the fused build looks neighbours / siblings / children / parents up in 4^3 bricks (scn_pyramid2.hip) instead of
    probing a voxel hash table 27 times per row.  With the bricks off (SCN_PYRAMID_NO_BRICKS) every index structure is the
    same bit for bit; `sparse-overflow`: 5 000 random points in 256^3 -- nearly one brick per point, the directory (a quarter
    of the voxel table) overflows and the build falls back to the voxel tables by itself, same structures; `edges`: sites on
    the faces of the 16-bit coordinate range (neighbours outside it do not exist)."""
    from sparse_rcnn_amd.metadata import Metadata
    from sparse_rcnn_amd.synthetic import make_batch
    if case == "sparse-overflow":
        g = torch.Generator().manual_seed(5)
        coords = torch.cat([torch.randint(0, 256, (5000, 3), generator=g), torch.zeros(5000, 1, dtype=torch.long)], 1)
        size, bs, levels = torch.tensor([256, 256, 256]), 1, 3
    elif case == "edges":
        g = torch.Generator().manual_seed(6)
        c = torch.randint(0, 6, (3000, 3), generator=g)
        c = torch.where(torch.rand(3000, 3, generator=g) < 0.5, c, 65535 - c)          # both ends of every axis
        coords = torch.cat([c, torch.randint(0, 2, (3000, 1), generator=g)], 1)
        coords = coords[coords[:, 3].argsort(stable=True)]
        size, bs, levels = torch.tensor([65536, 65536, 65536]), 2, 3
    else:
        n_s, grid, target, levels = {"150k-4": (1, (512, 512, 256), 150_000, 4), "12crops-6": (12, (128, 128, 64), 12_500, 6)}[case]
        coords, _, size, bs, _ = make_batch(n_s, grid, target, dup=1.15, seed=4)
    cg = coords.to(gpu)
    b = Metadata(3).build_native(size, cg, bs, 4, levels, 3)
    _SW["SCN_PYRAMID_NO_BRICKS"] = "1"
    try:
        a = Metadata(3).build_native(size, cg, bs, 4, levels, 3)
    finally:
        del _SW["SCN_PYRAMID_NO_BRICKS"]
    torch.cuda.synchronize()
    assert list(a.grids) == list(b.grids) and [g.n for g in a.grids.values()] == [g.n for g in b.grids.values()]
    ta, tb = _index_tensors(a), _index_tensors(b)
    assert [n for n, _ in ta] == [n for n, _ in tb]
    for (name, x), (_, y) in zip(ta, tb):
        assert x.shape == y.shape and torch.equal(x, y), (case, name)


@pytest.mark.parametrize("case", ["150k-4", "150k-6-x", "12crops-6", "small-3-x", "tiny-2", "one-level"])
def test_fused_pyramid_build_equals_the_round3_builder(gpu, case):
    """scn_pyramid_build_ex(SCN_PYRAMID_FUSED) -- level sizes device-resident, the levels side by side inside each launch,
    one look-back numbering pass, ONE host wait (scn_pyramid2.hip) -- against the round-3 builder (SCN_PYRAMID_V1=1: ~136
    launches, a host wait per level): every index structure bit for bit, incl. both tile hand-out orders and the bin starts
    of the XCD-local one, at the BASELINE scene (four and six levels), on the reference's training batch of 12 crops, and on
    scenes whose deep levels shrink to a handful of rows."""
    import os
    from sparse_rcnn_amd.metadata import Metadata
    from sparse_rcnn_amd.synthetic import make_batch
    from sparse_rcnn_amd import metadata as MD
    assert MD.FUSED_INDEX
    n_s, grid, target, levels, with_x = {"150k-4": (1, (512, 512, 256), 150_000, 4, False),
                                         "150k-6-x": (1, (512, 512, 256), 150_000, 6, True),
                                         "12crops-6": (12, (128, 128, 64), 12_500, 6, False),
                                         "small-3-x": (2, (64, 64, 32), 3_000, 3, True),
                                         "tiny-2": (1, (8, 8, 4), 20, 2, False),
                                         "one-level": (1, (64, 64, 32), 5_000, 1, True)}[case]
    coords, feats, size, bs, _ = make_batch(n_s, grid, target, dup=1.15, seed=3)
    cg = coords.to(gpu)
    b = Metadata(3).build_native(size, cg, bs, 4, levels, 3, xcd_order=with_x)
    _SW["SCN_PYRAMID_V1"] = "1"
    try:
        a = Metadata(3).build_native(size, cg, bs, 4, levels, 3, xcd_order=with_x)
    finally:
        del _SW["SCN_PYRAMID_V1"]
    torch.cuda.synchronize()
    assert list(a.grids) == list(b.grids) and [g.n for g in a.grids.values()] == [g.n for g in b.grids.values()]
    assert set(a.subm) == set(b.subm) and set(a.strided) == set(b.strided) and a.n_samples == b.n_samples
    ta, tb = _index_tensors(a, with_x), _index_tensors(b, with_x)
    assert len(ta) == len(tb) and len(ta) >= 4 + levels * 10
    for (na, x), (nb_, y) in zip(ta, tb):
        assert na == nb_ and x.shape == y.shape and torch.equal(x, y), (case, na)
    # the hash tables differ in capacity and slot layout but answer the same questions: every row is found under its key
    for s, g in b.grids.items():
        if g.n:
            rb = b.subm_rulebook(s, 3)
            assert torch.equal(rb.table[13], torch.arange(g.n, dtype=torch.int32, device=gpu))


def test_fused_pyramid_build_reports_out_of_range_coordinates(gpu):
    """Coordinates outside the 16-bit key fields: the fused build returns the error of the step-by-step path (the count
    travels back with the sizes in the one device -> host copy)."""
    from sparse_rcnn_amd.metadata import Metadata
    coords, size, batch = _cloud(11, grid=(32, 32, 16), n=500, batch=1, dup=0)
    bad = coords.clone()
    bad[7, 1] = 70_000
    with pytest.raises(_scn().ScnError, match="outside"):
        Metadata(3).build_native(torch.tensor([32, 32, 16]), bad.to(gpu), 1, 4, 2, 3)
    neg = coords.clone()
    neg[3, 0] = -1
    with pytest.raises(_scn().ScnError, match="outside"):
        Metadata(3).build_native(torch.tensor([32, 32, 16]), neg.to(gpu), 1, 4, 2, 3)


def test_async_row_count_equals_synchronous_dedup(gpu):
    """scn_dedup_launch (row count left on the device, no host sync) numbers rows exactly like scn_dedup_build."""
    from sparse_rcnn_amd import metadata as M
    coords, size, batch = _cloud(5, grid=(32, 32, 16), n=3000, batch=2, dup=400)
    c32 = coords.to(gpu).to(torch.int32).contiguous()
    for shift in (0, 1):
        pend = M._Dedup(c32, shift, True, True)
        # unrelated work may be queued before the count is awaited
        _ = torch.ones(1024, device=gpu).sum()
        grid, item_row, cnt, first = pend.finish()
        keys = O.pack_keys(np.concatenate([coords[:, :3].numpy() >> shift, coords[:, 3:].numpy()], 1))
        _, first_idx, inv = np.unique(keys, return_index=True, return_inverse=True)
        order = np.argsort(first_idx)                       # rows numbered by first occurrence
        rank = np.empty_like(order); rank[order] = np.arange(len(order))
        assert grid.n == len(first_idx)
        assert np.array_equal(item_row.cpu().numpy(), rank[inv])
        assert np.array_equal(first.cpu().numpy(), np.sort(first_idx))
        assert np.array_equal(cnt.cpu().numpy(), np.bincount(rank[inv]))


@pytest.mark.parametrize("average", [False, True])
def test_pooling(gpu, average):
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=8, cin=6)
    pool = (scn.AveragePooling if average else scn.MaxPooling)(3, (2, 2, 2), (2, 2, 2))
    y = pool(x)
    assert tuple(int(s) for s in y.spatial_size) == tuple(int(s) // 2 for s in size)
    scene.strided_rules(0)
    child = scene.strided[0]["child"]
    Xo = O.input_layer_fwd(feats, scene.prow, scene.n(0), 4).requires_grad_()
    yo = O.pool_fwd(Xo, child, average)
    _close(y.features, yo, 1e-6, "pool fwd")
    # dense twin (module_factory.py:330-332,351-353) where it is defined the same way: average pooling of the zero-filled grid
    if average:
        import torch.nn.functional as Fn
        dense = O.sparse_to_dense(Xo.detach(), scene.coords0, size.tolist(), 2)
        dd = Fn.avg_pool3d(dense, 2, 2)
        c = torch.from_numpy(scene.strided[0]["coords"])
        _close(y.features, dd[c[:, 3], :, c[:, 0], c[:, 1], c[:, 2]], 1e-6, "dense avg twin")
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(4))
    (gx,) = torch.autograd.grad(y.features, x.features, g.to(gpu))
    (ox,) = torch.autograd.grad(yo, Xo, g)
    _close(gx, ox, 1e-6, "pool bwd")


def test_mask_head_path_matches_oracle(gpu):
    """BASELINE config 3 shape in fp32: backbone features -> per-point features (OutputLayer) ++ raw point features ->
    sparse ROI crop (mode-4 re-voxelisation, batch_size = #boxes, spatial size + 32) -> internal U-Net on the ROI batch
    -> per-point logits (model.py:583-596,758-782; roi_select_sparse.py:38-52,74-84)."""
    from sparse_rcnn_amd import roi
    from sparse_rcnn_amd.synthetic import make_boxes
    from sparse_rcnn_amd.unet import Backbone, SparseUNet
    scn = _scn()
    coords, size, batch = _cloud(41, grid=(32, 32, 16), n=2500, batch=2, dup=300)
    raw = torch.randn(len(coords), 7, generator=torch.Generator().manual_seed(1))
    bparams = O.init_unet_params(7, (16, 24), seed=2)
    mparams = O.init_unet_params(16 + 7, (24, 32), seed=3)
    backbone = Backbone(7, (16, 24)).to(gpu); backbone.unet.load_oracle_params(bparams)
    mask_unet = SparseUNet(23, (24, 32)).to(gpu); mask_unet.load_oracle_params(mparams)
    head = torch.nn.Linear(24, 5).to(gpu)
    bbox_batch = make_boxes(coords, n_boxes=6, seed=5, lo=4, hi=24)
    # --- HIP path
    raw_g = raw.to(gpu).requires_grad_()
    fmap = backbone(coords, raw_g, size, batch)
    per_point = scn.OutputLayer(3)(fmap)                                     # [Npts, 16]
    cat = torch.cat([per_point, raw_g], 1)                                   # [Npts, 23]
    cut = roi.SparseRoiCut(roi.RawToTensorFeatureExtractorCombiner())
    roi_tensor, (is_inside, counts, _) = cut((coords, cat, size + 32, batch, [0]), bbox_batch)
    assert roi_tensor.batch_size() == 12
    logits = head(scn.OutputLayer(3)(mask_unet(roi_tensor)))                 # one row per cropped point
    # --- oracle
    scene = O.OracleScene(coords.numpy())
    raw_o = raw.clone().requires_grad_()
    bp = {k: v.clone() for k, v in bparams.items()}
    f_o = O.unet_forward(scene, raw_o, bp, (16, 24))
    cat_o = torch.cat([O.output_layer_fwd(f_o, scene.prow), raw_o], 1)
    boxes, cnt, assoc = O.transform_boxes([b.numpy() for b in bbox_batch])
    src, box_of, inside = O.roi_crop(coords.numpy(), boxes, assoc)
    assert np.array_equal(is_inside.numpy(), inside) and counts == cnt
    new_coords = np.concatenate([coords.numpy()[src][:, :3], box_of[:, None]], 1)
    rscene = O.OracleScene(new_coords)
    mp = {k: v.clone() for k, v in mparams.items()}
    m_o = O.unet_forward(rscene, cat_o[torch.from_numpy(src)], mp, (24, 32))
    logits_o = O.output_layer_fwd(m_o, rscene.prow) @ head.weight.detach().cpu().t() + head.bias.detach().cpu()
    _close(logits, logits_o, 2e-4, "mask logits")
    g = torch.randn(logits_o.shape, generator=torch.Generator().manual_seed(7))
    logits.backward(g.to(gpu))
    logits_o.backward(g)
    _close(raw_g.grad, raw_o.grad, 5e-4, "d raw features through ROI crop + both U-Nets")


# ---------------------------------------------------------------------------------------- bf16 storage (first piece)
BF16_TOL = 2.0 ** -7     # outputs are rounded to bf16 (8 significant bits: half an ulp = 2^-9 relative) after an fp32
                         # accumulation of bf16-rounded operands; stated relative to the output scale


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64), (8, 16), (48, 80), (128, 64), (256, 256), (24, 24), (512, 40)])
@pytest.mark.parametrize("relu_in", [False, True])
def test_conv_tiles_bf16_forward_and_backward_data(gpu, cin, cout, relu_in):
    """scn_conv_tiles_bf16 against the oracle convolution evaluated on the SAME bf16-rounded operands (features bf16,
    weights rounded to bf16, fp32 accumulation): only the final rounding of the result to bf16 and the summation order
    separate the two.  Forward with bias + residual, backward-data with the ReLU mask."""
    from sparse_rcnn_amd import functional as F, _lib as L
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=13, cin=8, n=1200, dup=100)
    sz = tuple(int(s) for s in size)
    rb = x.metadata.subm_rulebook(sz, 3)
    n = rb.n
    rules = scene.subm_rules(0, 3)
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    X = torch.randn(n, cin, generator=g).to(torch.bfloat16)
    R = torch.randn(n, cout, generator=g).to(torch.bfloat16)
    W = torch.randn(27, cin, cout, generator=g) * (2.0 / (27 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.5
    Wb = W.to(torch.bfloat16).float()                                         # what the kernel's LDS image holds
    flags = L.F_RELU_IN if relu_in else 0
    y = F.conv_rules_bf16(X.to(gpu), rb.tiles, n, W.to(gpu), b.to(gpu), cout, flags, residual=R.to(gpu))
    assert y.dtype == torch.bfloat16
    xin = torch.relu(X.float()) if relu_in else X.float()
    yo = O.conv_fwd(xin, rules, Wb, b, n) + R.float()
    _close(y.float(), yo, BF16_TOL, "bf16 fwd")
    # backward-data: dX = mask(X) * sum_o G[nbr_{26-o}] . W[o]^T
    G = torch.randn(n, cout, generator=g).to(torch.bfloat16)
    M = torch.randn(n, cin, generator=g).to(torch.bfloat16)                   # ReLU input saved by the forward
    dx = F.conv_rules_bf16(G.to(gpu), rb.tiles, n, W.to(gpu), None, cin, L.F_W_TRANSPOSED | L.F_OFF_REVERSE,
                           relu_mask=M.to(gpu))
    dxo, _, _ = O.conv_bwd(torch.zeros(n, cin), G.float(), rules, Wb, has_bias=False)
    dxo = dxo * (M.float() > 0)
    _close(dx.float(), dxo, BF16_TOL, "bf16 bwd-data")
    # bitwise reproducible, and the in-launch K reduction gives the bits of the two-launch form
    assert torch.equal(y, F.conv_rules_bf16(X.to(gpu), rb.tiles, n, W.to(gpu), b.to(gpu), cout, flags, residual=R.to(gpu)))
    F.FUSED_K = False
    try:
        y2 = F.conv_rules_bf16(X.to(gpu), rb.tiles, n, W.to(gpu), b.to(gpu), cout, flags, residual=R.to(gpu))
        dx2 = F.conv_rules_bf16(G.to(gpu), rb.tiles, n, W.to(gpu), None, cin, L.F_W_TRANSPOSED | L.F_OFF_REVERSE,
                                relu_mask=M.to(gpu))
    finally:
        F.FUSED_K = True
    assert torch.equal(y, y2) and torch.equal(dx, dx2)
    # a packed image may be shared by calls as long as the weights do not change
    img = F.pack_weights_bf16(W.to(gpu), cin, cout, 27, flags)
    assert torch.equal(y, F.conv_rules_bf16(X.to(gpu), rb.tiles, n, W.to(gpu), b.to(gpu), cout, flags, residual=R.to(gpu),
                                            image=img))


def _bf16_image_restated(W, cin, cout, n_off, wt, rev):
    """The documented layout of the bf16 weight image (scn_conv_ts_bf16.hip: image[chunk * n_kc + kci][o][n = 16 nb + i][kk] =
    bf16(W[o'][k = 32 kci + kk][col = ct chunk + NB i + nb]), zero outside the layer; NB = 4 / ct = 64 above 32 output columns,
    else 2 / 32) built with torch indexing -> int16 bit patterns."""
    nb = 4 if cout > 32 else 2
    ct, n_chunks, n_kc = 16 * nb, -(-cout // (16 * nb)), -(-cin // 32)
    Wd = W.flip(0) if rev else W
    Wd = Wd.transpose(1, 2) if wt else Wd                              # -> [o][k][col]
    full = torch.zeros(n_off, n_kc * 32, n_chunks * ct)
    full[:, :cin, :cout] = Wd
    n = torch.arange(ct)
    col_of_n = nb * (n % 16) + n // 16                                 # n = 16 nb_idx + i  ->  local column NB i + nb_idx
    img = full.view(n_off, n_kc, 32, n_chunks, ct)[..., col_of_n]      # [o][kci][kk][chunk][n]
    img = img.permute(3, 1, 0, 4, 2).contiguous()                      # [chunk][kci][o][n][kk]
    return img.to(torch.bfloat16).view(torch.int16).reshape(-1)


@pytest.mark.parametrize("cin,cout,n_off", [(64, 64, 27), (32, 32, 27), (24, 48, 27), (128, 96, 8), (40, 24, 8), (256, 256, 27)])
def test_bf16_weight_image_layout_bit_for_bit(gpu, cin, cout, n_off):
    """scn_conv_tiles_bf16_pack and _pack_many (round 4: one wave per (slice, offset) unit, lane = column) against the
    documented image layout restated with torch indexing: forward image, backward-data image (stored [o][col][k], offsets
    reversed), edge slices (24, 40, 48, 96 channels), both entry points -- equal bit patterns."""
    from sparse_rcnn_amd import functional as F, _lib as L
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    W = torch.randn(n_off, cin, cout, generator=g)
    Wg = W.to(gpu)
    back = L.F_W_TRANSPOSED | L.F_OFF_REVERSE
    # the backward-data image of a layer [o][cin][cout]: the kernel sees a (cout -> cin) layer whose weights are stored
    # [o][col][k] -- the same tensor
    cases = [(0, _bf16_image_restated(W, cin, cout, n_off, False, False), (cin, cout)),
             (back, _bf16_image_restated(W, cout, cin, n_off, True, True), (cout, cin))]
    for flags, want, (ci, co) in cases:
        got = F.pack_weights_bf16(Wg, ci, co, n_off, flags).view(torch.int16).cpu()
        assert got.numel() == want.numel() and torch.equal(got, want), (flags, int((got != want).sum()))
    with F.packed_weights([(Wg, cin, cout, n_off, 0), (Wg, cout, cin, n_off, back)]):
        for flags, want, (ci, co) in cases:
            got = F.packed_image(Wg, ci, co, n_off, flags).view(torch.int16).cpu()
            assert torch.equal(got, want), flags


def test_conv_tiles_bf16_rejects_unsupported_inputs(gpu):
    from sparse_rcnn_amd import functional as F
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=13, cin=8, n=300, dup=0)
    rb = x.metadata.subm_rulebook(tuple(int(s) for s in size), 3)
    W = torch.zeros(27, 12, 8, device=gpu)
    with pytest.raises(scn.ScnError):
        F.conv_rules_bf16(torch.zeros(rb.n, 12, device=gpu), rb.tiles, rb.n, W, None, 8)             # fp32 features
    with pytest.raises(scn.ScnError):
        F.conv_rules_bf16(torch.zeros(rb.n, 12, device=gpu, dtype=torch.bfloat16), rb.tiles, rb.n, W, None, 8)   # cin % 8


def test_conv_tiles_in_launch_k_reduction_gives_the_bits_of_the_two_launch_form(gpu):
    """Layers with more than 32 input channels split K over workgroups.  scn_conv_tiles adds the K-chunk partial sums
    inside the launch (last arriver per (tile, column chunk), write-through slab stores, agent-scope ticket, ascending
    K-chunk order); the two-launch form (SCN_F_SPLIT_SUM + scn_conv_tiles_finish, or arrival == NULL) is the cross-check:
    same association, so every output word must be equal -- over repeated launches that re-use the same slabs and
    counters with fresh data (a stale line or a lost arrival shows up as a mismatch), forward and backward-data."""
    from sparse_rcnn_amd import functional as F, profiling, _lib as L
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=17, cin=8, n=1500, dup=100)
    rb = x.metadata.subm_rulebook(tuple(int(s) for s in size), 3)
    n = rb.n
    back = L.F_W_TRANSPOSED | L.F_OFF_REVERSE
    for cin, cout in ((32, 32), (64, 48), (128, 128), (48, 80), (256, 64), (512, 32)):
        g = torch.Generator().manual_seed(cin)
        W = (torch.randn(27, cin, cout, generator=g) * 0.05).to(gpu); b = torch.randn(cout, generator=g).to(gpu)
        Wt = (torch.randn(27, cout, cin, generator=g) * 0.05).to(gpu)
        for rep in range(6):
            X = torch.randn(n, cin, generator=g).to(gpu); R = torch.randn(n, cout, generator=g).to(gpu)
            M = torch.randn(n, cout, generator=g).to(gpu)
            outs = []
            for fused in (False, True):
                F.FUSED_K = fused
                try:
                    y = F.conv_rules(X, rb.tiles, n, W, b, cout, L.F_RELU_IN, residual=R, n_rules=rb.rules.total)
                    d = F.conv_rules(X, rb.tiles, n, Wt, None, cout, back | L.F_RESIDUAL_LAST, residual=R, relu_mask=M,
                                     n_rules=rb.rules.total)
                finally:
                    F.FUSED_K = True
                outs.append((y, d))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (cin, cout, rep)
    # the two-launch form under the kernel timer still reports both kernels; the fused form is one launch
    X = torch.randn(n, 64, generator=g).to(gpu); W = (torch.randn(27, 64, 32, generator=g) * 0.05).to(gpu)
    for fused, names in ((False, {"k_conv_ts", "k_conv_ts_sum"}), (True, {"k_conv_ts"})):
        timer = profiling.KernelTimer(every=1, names=None)
        timer.begin_step()
        profiling.TIMER, F.FUSED_K = timer, fused
        try:
            F.conv_rules(X, rb.tiles, n, W, None, 32, n_rules=rb.rules.total)
        finally:
            profiling.TIMER, F.FUSED_K = None, True
        torch.cuda.synchronize()
        assert set(timer.summary()) == names


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 64), (8, 16), (48, 80), (128, 64), (256, 256), (23, 7), (130, 33),
                                      (24, 24), (32, 64), (64, 32), (512, 256), (16, 136)])
@pytest.mark.parametrize("relu_in", [False, True])
def test_wgrad_bf16_storage(gpu, cin, cout, relu_in):
    """scn_wgrad_rules_bf16: operands stored in bf16, products exact (a bf16 x bf16 product fits fp32), fp32 sums --
    against the oracle's weight gradient of the same (bf16-representable) operands at the fp32 tolerance.  Channel
    counts that are multiples of 8 run on the bf16 MFMA kernel (LDS-transposed operands, its own summation order: equal
    to the fp32 kernel fed the widened operands within fp32 rounding); the others widen in registers and run the fp32
    kernel's arithmetic in its order (bit-identical)."""
    from sparse_rcnn_amd import functional as F, _lib as L
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=19, cin=8, n=1300, dup=100)
    rb = x.metadata.subm_rulebook(tuple(int(s) for s in size), 3)
    n = rb.n
    g = torch.Generator().manual_seed(cin * 77 + cout)
    X = torch.randn(n, cin, generator=g).to(torch.bfloat16)
    G = torch.randn(n, cout, generator=g).to(torch.bfloat16)
    flags = L.F_RELU_IN if relu_in else 0
    r = rb.rules
    dW = F.wgrad_rules_bf16(X.to(gpu), G.to(gpu), r.in_rows, r.out_rows, r.prefix_host, 27, flags)
    xin = torch.relu(X.float()) if relu_in else X.float()
    _, dWo, _ = O.conv_bwd(xin, G.float(), scene.subm_rules(0, 3), torch.zeros(27, cin, cout), has_bias=False)
    _close(dW, dWo, 1e-4, "bf16-storage dW")
    ref = F.wgrad_rules(X.float().to(gpu), G.float().to(gpu), r.in_rows, r.out_rows, r.prefix_host, 27, flags)
    if cin % 8 or cout % 8:
        assert torch.equal(dW, ref)
    else:
        _close(dW, ref, 1e-5, "bf16-MFMA dW vs fp32-MFMA dW of the widened operands")
        assert torch.equal(dW, F.wgrad_rules_bf16(X.to(gpu), G.to(gpu), r.in_rows, r.out_rows, r.prefix_host, 27, flags))
    # bias gradient from the same pass (centre offset): column sums of dY
    dW2, db = F.wgrad_bias_rules(X.to(gpu), G.to(gpu), r.in_rows, r.out_rows, r.prefix_host, 27, 1 << 13, flags)
    assert torch.equal(dW2, dW)
    _close(db, G.float().sum(0), 1e-5, "db from the bf16 weight-gradient pass")
    _close(F.colsum(G.to(gpu)), G.float().sum(0), 1e-5, "bf16 colsum")


def _bf16r(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("c", [32, 64, 24])
def test_residual_block_bf16_storage(gpu, c):
    """functional.ResidualBlockFunctionBF16 (bf16-stored x, h, y and gradients; fp32 parameters) against the oracle
    evaluated with the same storage roundings: h = bf16(conv1(relu x)), y = bf16(x + conv2(relu h)), weights rounded to
    bf16 for the feature path.  Forward within 2^-6 of the output scale (two roundings), gradients by relative L2."""
    scn = _scn()
    from sparse_rcnn_amd.unet import residual_block
    _, coords, feats, fg, x, scene, size = _input(gpu, seed=23, cin=8, n=1400, dup=120)
    n = scene.n(0)
    rules = scene.subm_rules(0, 3)
    g = torch.Generator().manual_seed(c)
    X = torch.randn(n, c, generator=g).to(torch.bfloat16)
    block = residual_block(c).to(gpu)
    convs = [m for m in block.modules() if isinstance(m, scn.SubmanifoldConvolution)]
    with torch.no_grad():
        for m in convs:
            m.bias.normal_(0, 0.3)
    xg = X.to(gpu).requires_grad_()
    xin = scn.SparseConvNetTensor(features=xg, metadata=x.metadata, spatial_size=x.spatial_size)
    y = block(xin).features
    assert y.dtype == torch.bfloat16
    W1, b1, W2, b2 = (convs[0].weight.detach().cpu(), convs[0].bias.detach().cpu(), convs[1].weight.detach().cpu(),
                      convs[1].bias.detach().cpu())
    Xo = X.float().requires_grad_()
    W1o, W2o = _bf16r(W1).view(27, c, c).requires_grad_(), _bf16r(W2).view(27, c, c).requires_grad_()
    b1o, b2o = b1.clone().requires_grad_(), b2.clone().requires_grad_()
    h = O.conv(torch.relu(Xo), W1o, b1o, rules, n)
    hq = h + (_bf16r(h.detach()) - h.detach())                                  # storage rounding, straight-through
    yo = Xo + O.conv(torch.relu(hq), W2o, b2o, rules, n)
    _close(y.float(), _bf16r(yo.detach()), 2.0 ** -6, "bf16 block fwd")
    G = torch.randn(n, c, generator=g).to(torch.bfloat16)
    got = torch.autograd.grad(y, (xg, convs[0].weight, convs[0].bias, convs[1].weight, convs[1].bias), G.to(gpu))
    exp = torch.autograd.grad(yo, (Xo, W1o, b1o, W2o, b2o), G.float())
    for a, e, name in zip(got, exp, ("dX", "dW1", "db1", "dW2", "db2")):
        a, e = a.detach().float().cpu().double().reshape(-1), e.double().reshape(-1)
        l2 = ((a - e).norm() / e.norm().clamp_min(1e-12)).item()
        assert l2 <= 2e-2, f"{name}: relative L2 {l2:.2e}"                      # bf16 gradients: 2^-8 per stored value


def test_backbone_with_bf16_blocks_tracks_the_fp32_backbone(gpu):
    """Backbone(bf16_blocks=True): residual units on the bf16 storage path, the rest fp32.  Same parameters as an fp32
    backbone: outputs and parameter gradients agree to bf16 accuracy (relative L2), everything finite."""
    from sparse_rcnn_amd.unet import Backbone
    coords, size, batch = _cloud(29, grid=(32, 32, 16), n=3000, batch=2, dup=300)
    feats = torch.randn(len(coords), 7, generator=torch.Generator().manual_seed(3)).to(gpu)
    torch.manual_seed(1)
    ref = Backbone(7, (16, 24, 32)).to(gpu)
    mix = Backbone(7, (16, 24, 32), bf16_blocks=True).to(gpu)
    mix.load_state_dict(ref.state_dict())
    outs = []
    for net in (ref, mix):
        out = net(coords, feats, size, batch).features
        assert out.dtype == torch.float32
        out.backward(torch.ones_like(out))
        outs.append(out.detach())
    l2 = ((outs[1] - outs[0]).norm() / outs[0].norm()).item()
    assert torch.isfinite(outs[1]).all() and l2 < 3e-2, l2
    for (k, p), q in zip(ref.named_parameters(), mix.parameters()):
        assert torch.isfinite(q.grad).all(), k
        rel = ((q.grad - p.grad).norm() / p.grad.norm().clamp_min(1e-12)).item()
        assert rel < 0.1, (k, rel)


@pytest.mark.parametrize("cin,cout", [(64, 32), (32, 64), (48, 23), (7, 16)])
def test_small_gemms_bf16_storage_equal_the_fp32_kernels_rounded(gpu, cin, cout):
    """scn_gemm_table_bf16 / scn_gemm_rules_bf16: bf16-stored features widened exactly, the fp32 kernels' arithmetic,
    one rounding of the result -- so they equal the fp32 kernels on the widened operands, rounded to bf16, bit for bit
    (1x1 layer forward / backward-data, Deconvolution forward, Convolution backward-data with the ReLU mask)."""
    from sparse_rcnn_amd import functional as F, _lib as L
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=31, cin=8, n=1500, dup=100)
    sz = tuple(int(s) for s in size)
    sb = x.metadata.strided_rulebook(sz)
    n, nc, r = sb.n_fine, sb.n_coarse, sb.rules
    g = torch.Generator().manual_seed(cin + cout)
    bf = lambda t: t.to(torch.bfloat16)
    # 1x1
    X = bf(torch.randn(n, cin, generator=g)).to(gpu); R = bf(torch.randn(n, cout, generator=g)).to(gpu)
    W = (torch.randn(1, cin, cout, generator=g) * 0.2).to(gpu); b = torch.randn(cout, generator=g).to(gpu)
    y = F.gemm_table(X, None, 1, n, W, b, cout, residual=R)
    assert y.dtype == torch.bfloat16
    assert torch.equal(y, bf(F.gemm_table(X.float(), None, 1, n, W, b, cout, residual=R.float())))
    G = bf(torch.randn(n, cout, generator=g)).to(gpu)
    dx = F.gemm_table(G, None, 1, n, W, None, cin, L.F_W_TRANSPOSED)
    assert torch.equal(dx, bf(F.gemm_table(G.float(), None, 1, n, W, None, cin, L.F_W_TRANSPOSED)))
    # strided rule lists: Deconvolution forward (coarse -> fine) and Convolution backward-data with the ReLU mask
    Xc = bf(torch.randn(nc, cin, generator=g)).to(gpu); Wu = (torch.randn(8, cin, cout, generator=g) * 0.2).to(gpu)
    u = F.gemm_rules(Xc, r.out_rows, r.in_rows, r.prefix_host, 8, n, Wu, b, cout, L.F_RELU_IN)
    assert torch.equal(u, bf(F.gemm_rules(Xc.float(), r.out_rows, r.in_rows, r.prefix_host, 8, n, Wu, b, cout, L.F_RELU_IN)))
    Wd = (torch.randn(8, cout, cin, generator=g) * 0.2).to(gpu); M = bf(torch.randn(n, cout, generator=g)).to(gpu)
    d = F.gemm_rules(Xc, r.out_rows, r.in_rows, r.prefix_host, 8, n, Wd, None, cout, L.F_W_TRANSPOSED, relu_mask=M)
    assert torch.equal(d, bf(F.gemm_rules(Xc.float(), r.out_rows, r.in_rows, r.prefix_host, 8, n, Wd, None, cout,
                                         L.F_W_TRANSPOSED, relu_mask=M.float())))


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 32), (128, 64), (256, 128), (24, 24), (40, 18), (256, 200)])
@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_lds_tiled_row_gemms_reproduce_the_register_kernels_bit_for_bit(gpu, cin, cout, storage):
    """scn_gemm_lt.hip behind scn_gemm_table (identity table) and scn_gemm_rules: the A tile staged through LDS, the
    summation order of scn_conv.hip's kernels -- SCN_F_GEMM_V1 runs those, and the results are equal bit for bit, with
    every epilogue operand (bias, residual, ReLU-backward mask, input ReLU), both weight orientations, ragged tiles."""
    from sparse_rcnn_amd import functional as F, _lib as L
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=37, cin=8, n=1700, dup=100)
    sz = tuple(int(s) for s in size)
    sb = x.metadata.strided_rulebook(sz)
    n, nc, r = sb.n_fine, sb.n_coarse, sb.rules
    g = torch.Generator().manual_seed(3 * cin + cout)
    st = (lambda t: t.to(torch.bfloat16)) if storage == "bf16" else (lambda t: t)
    X = st(torch.randn(n, cin, generator=g)).to(gpu); R = st(torch.randn(n, cout, generator=g)).to(gpu)
    M = st(torch.randn(n, cout, generator=g)).to(gpu)
    W = (torch.randn(1, cin, cout, generator=g) * 0.2).to(gpu); b = torch.randn(cout, generator=g).to(gpu)
    for fl, kw in ((0, dict(residual=R)), (L.F_RELU_IN, dict(relu_mask=M)), (0, {})):
        y = F.gemm_table(X, None, 1, n, W, b, cout, fl, **kw)
        assert torch.equal(y, F.gemm_table(X, None, 1, n, W, b, cout, fl | L.F_GEMM_V1, **kw)), (fl, list(kw))
    G = st(torch.randn(n, cout, generator=g)).to(gpu); MX = st(torch.randn(n, cin, generator=g)).to(gpu)
    dx = F.gemm_table(G, None, 1, n, W, None, cin, L.F_W_TRANSPOSED, relu_mask=MX)
    assert torch.equal(dx, F.gemm_table(G, None, 1, n, W, None, cin, L.F_W_TRANSPOSED | L.F_GEMM_V1, relu_mask=MX))
    ref = (G.double() @ W[0].double().t()) * (MX.double() > 0)
    _close(dx.float(), ref, 2e-2 if storage == "bf16" else FEAT_TOL, "dX vs fp64")
    # rule lists (8 offsets, ragged 128/64/32-rule tiles): Deconvolution forward, Convolution backward-data
    Xc = st(torch.randn(nc, cin, generator=g)).to(gpu); Wu = (torch.randn(8, cin, cout, generator=g) * 0.2).to(gpu)
    u = F.gemm_rules(Xc, r.out_rows, r.in_rows, r.prefix_host, 8, n, Wu, b, cout, L.F_RELU_IN)
    assert torch.equal(u, F.gemm_rules(Xc, r.out_rows, r.in_rows, r.prefix_host, 8, n, Wu, b, cout, L.F_RELU_IN | L.F_GEMM_V1))
    Wd = (torch.randn(8, cout, cin, generator=g) * 0.2).to(gpu)
    d = F.gemm_rules(Xc, r.out_rows, r.in_rows, r.prefix_host, 8, n, Wd, None, cout, L.F_W_TRANSPOSED, relu_mask=M)
    assert torch.equal(d, F.gemm_rules(Xc, r.out_rows, r.in_rows, r.prefix_host, 8, n, Wd, None, cout,
                                       L.F_W_TRANSPOSED | L.F_GEMM_V1, relu_mask=M))


@pytest.mark.parametrize("c0,c1,cout", [(32, 32, 32), (64, 64, 64), (128, 128, 128), (24, 24, 24), (16, 40, 48), (512, 512, 64)])
@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_row_gemm_over_join_parts_equals_the_gemm_on_the_concatenated_slab(gpu, c0, c1, cout, storage):
    """scn_gemm_rows2: NetworkInNetwork over JoinTable([up, skip]) read from the two slabs, and its backward-data written to
    the two gradient slabs -- bit-equal to the one-source kernel on torch.cat of the parts / to the column split of its
    one-destination result, and within fp32 tolerance of the fp64 product."""
    from sparse_rcnn_amd import functional as F, _lib as L
    n = 3001
    g = torch.Generator().manual_seed(c0 + 7 * c1 + cout)
    st = (lambda t: t.to(torch.bfloat16)) if storage == "bf16" else (lambda t: t)
    X0 = st(torch.randn(n, c0, generator=g)).to(gpu); X1 = st(torch.randn(n, c1, generator=g)).to(gpu)
    W = (torch.randn(c0 + c1, cout, generator=g) * 0.2).to(gpu); b = torch.randn(cout, generator=g).to(gpu)
    y = F.gemm_rows2(X0, X1, W, b, cout)
    cat = torch.cat([X0, X1], 1).contiguous()
    assert torch.equal(y, F.gemm_table(cat, None, 1, n, W, b, cout))
    _close(y.float(), cat.double() @ W.double() + b.double(), 2e-2 if storage == "bf16" else FEAT_TOL, "rows2 fwd vs fp64")
    G = st(torch.randn(n, cout, generator=g)).to(gpu)
    d0, d1 = F.gemm_rows2_bwd(G, W, c0, c1)
    both = F.gemm_table(G, None, 1, n, W, None, c0 + c1, L.F_W_TRANSPOSED)
    assert torch.equal(d0, both[:, :c0]) and torch.equal(d1, both[:, c0:])
    # the autograd node on top of it, against the part-by-part form
    Wp = W.clone().requires_grad_(); bp = b.clone().requires_grad_()
    parts = [X0.clone().requires_grad_(), X1.clone().requires_grad_()]
    out = F.JoinedNetworkInNetworkFunction.apply(Wp, bp, *parts)
    assert torch.equal(out, y)
    out.backward(G)
    assert torch.equal(parts[0].grad, d0) and torch.equal(parts[1].grad, d1)
    F.FUSED_JOIN = False
    try:
        Wq = W.clone().requires_grad_(); bq = b.clone().requires_grad_()
        parts2 = [X0.clone().requires_grad_(), X1.clone().requires_grad_()]
        out2 = F.JoinedNetworkInNetworkFunction.apply(Wq, bq, *parts2)
        out2.backward(G)
    finally:
        F.FUSED_JOIN = True
    _close(out.float(), out2.float(), 2e-2 if storage == "bf16" else 1e-5, "fused vs chained forward")
    assert torch.equal(Wp.grad, Wq.grad) and torch.equal(bp.grad, bq.grad)
    assert torch.equal(parts[0].grad, parts2[0].grad) and torch.equal(parts[1].grad, parts2[1].grad)


def test_backbone_with_bf16_features_everywhere_tracks_the_fp32_backbone(gpu):
    """Backbone(bf16_blocks="all"): every layer after the first 1x1 convolution on bf16-stored features."""
    from sparse_rcnn_amd.unet import Backbone
    coords, size, batch = _cloud(37, grid=(32, 32, 16), n=3000, batch=2, dup=300)
    feats = torch.randn(len(coords), 7, generator=torch.Generator().manual_seed(3)).to(gpu)
    torch.manual_seed(1)
    ref = Backbone(7, (16, 24, 32)).to(gpu)
    mix = Backbone(7, (16, 24, 32), bf16_blocks="all").to(gpu)
    mix.load_state_dict(ref.state_dict())
    outs = []
    for net in (ref, mix):
        fin = feats.clone().requires_grad_()
        out = net(coords, fin, size, batch).features
        assert out.dtype == torch.float32
        out.backward(torch.ones_like(out))
        outs.append((out.detach(), fin.grad))
    l2 = ((outs[1][0] - outs[0][0]).norm() / outs[0][0].norm()).item()
    assert torch.isfinite(outs[1][0]).all() and l2 < 3e-2, l2
    assert ((outs[1][1] - outs[0][1]).norm() / outs[0][1].norm()).item() < 0.1
    for (k, p), q in zip(ref.named_parameters(), mix.parameters()):
        assert q.grad.dtype == torch.float32 and torch.isfinite(q.grad).all(), k
        rel = ((q.grad - p.grad).norm() / p.grad.norm().clamp_min(1e-12)).item()
        assert rel < 0.1, (k, rel)


def test_config3_path_with_bf16_storage_tracks_fp32(gpu):
    """BASELINE config 3's path -- backbone -> per-point features ++ raw features -> sparse ROI crop -> mask-head U-Net
    -> per-point logits -- with both U-Nets on bf16-stored features (bf16_blocks="all"; InputLayer, OutputLayer, the ROI
    crop and the first layer of each U-Net stay fp32), against the same path in fp32: logits within 3 % and the gradient
    of the raw point features within 15 % in relative L2 (gross-error bounds, see test_gpu_fuzz.py)."""
    from sparse_rcnn_amd import roi
    from sparse_rcnn_amd.synthetic import make_boxes
    from sparse_rcnn_amd.unet import Backbone, SparseUNet
    scn = _scn()
    coords, size, batch = _cloud(43, grid=(32, 32, 16), n=2500, batch=2, dup=300)
    raw = torch.randn(len(coords), 7, generator=torch.Generator().manual_seed(1))
    bparams = O.init_unet_params(7, (16, 24), seed=2)
    mparams = O.init_unet_params(16 + 7, (24, 32), seed=3)
    head = torch.nn.Linear(24, 5).to(gpu)
    bbox_batch = make_boxes(coords, n_boxes=6, seed=5, lo=4, hi=24)
    g = torch.randn(1, generator=torch.Generator().manual_seed(7))
    res = []
    for mode in (False, "all"):
        backbone = Backbone(7, (16, 24), bf16_blocks=mode).to(gpu); backbone.unet.load_oracle_params(bparams)
        mask_unet = SparseUNet(23, (24, 32), bf16_blocks=mode).to(gpu); mask_unet.load_oracle_params(mparams)
        raw_g = raw.to(gpu).requires_grad_()
        fmap = backbone(coords, raw_g, size, batch)
        cat = torch.cat([scn.OutputLayer(3)(fmap), raw_g], 1)
        cut = roi.SparseRoiCut(roi.RawToTensorFeatureExtractorCombiner(), dense_inside=False)
        roi_tensor, _ = cut((coords, cat, size + 32, batch, [0]), bbox_batch)
        logits = head(scn.OutputLayer(3)(mask_unet(roi_tensor)))
        assert logits.dtype == torch.float32
        logits.backward(torch.ones_like(logits))
        res.append((logits.detach(), raw_g.grad.clone()))
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    assert torch.isfinite(res[1][0]).all() and rel(res[1][0], res[0][0]) < 3e-2, rel(res[1][0], res[0][0])
    assert torch.isfinite(res[1][1]).all() and rel(res[1][1], res[0][1]) < 0.15, rel(res[1][1], res[0][1])


# ---------------------------------------------------------------------------------------- paired weight gradient
@pytest.mark.parametrize("C", [8, 32, 48, 64, 128])
@pytest.mark.parametrize("bias", [True, False])
def test_paired_weight_gradient_of_a_residual_unit(gpu, C, bias):
    """scn_wgrad_bias_rules2 (both weight gradients of a residual unit in one launch) against the oracle, and against the
    one-call-per-layer form: the same block, the same inputs, every gradient."""
    from sparse_rcnn_amd import functional as F
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=11, cin=C, n=2500, dup=100)
    rb = x.metadata.subm_rulebook(size, 3)
    g = torch.Generator().manual_seed(3)
    w1 = (torch.randn(27, C, C, generator=g) * (2.0 / (27 * C)) ** 0.5)
    w2 = (torch.randn(27, C, C, generator=g) * (2.0 / (27 * C)) ** 0.5)
    b1 = torch.randn(C, generator=g) * 0.1 if bias else None
    b2 = torch.randn(C, generator=g) * 0.1 if bias else None
    gy = torch.randn(rb.n, C, generator=g)

    def run(pair):
        F.WGRAD_PAIR = pair
        try:
            X = x.features.detach().clone().requires_grad_()
            ps = [t.to(gpu).requires_grad_() if t is not None else None for t in (w1, b1, w2, b2)]
            y = F.ResidualBlockFunction.apply(X, ps[0], ps[1], ps[2], ps[3], x.metadata, size)
            gr = torch.autograd.grad(y, [X] + [p_ for p_ in ps if p_ is not None], gy.to(gpu))
            return y.detach(), gr
        finally:
            F.WGRAD_PAIR = True
    y_p, g_p = run(True)
    y_s, g_s = run(False)
    assert torch.equal(y_p, y_s)
    for a, b in zip(g_p, g_s):              # same kernels on the same operands; the plans (units) differ -> rounding only
        _close(a, b, 2e-6, "paired vs single weight gradients")
    # oracle
    nbr, rules = O.subm_rulebook(scene.coords0, 3)
    Xo = x.features.detach().cpu().clone().requires_grad_()
    po = [t.clone().requires_grad_() if t is not None else None for t in (w1, b1, w2, b2)]
    h = O.conv(torch.relu(Xo), po[0], po[1], rules, scene.n(0))
    yo = Xo + O.conv(torch.relu(h), po[2], po[3], rules, scene.n(0))
    go = torch.autograd.grad(yo, [Xo] + [p_ for p_ in po if p_ is not None], gy)
    _close(y_p, yo, FEAT_TOL, "residual unit forward")
    for a, b in zip(g_p, go):
        _close(a, b, 2e-4, "residual unit gradients (paired weight gradient)")


@pytest.mark.parametrize("C", [16, 32, 64, 128])
def test_paired_weight_gradient_bf16_storage(gpu, C):
    """The bf16-stored residual unit: both weight gradients in one launch (scn_wgrad_bias_rules2_bf16, bf16 MFMA where the
    shape allows) against the one-call-per-layer form on the same operands -- products of bf16 values are exact in fp32,
    so only the order of the fp32 sums differs."""
    from sparse_rcnn_amd import functional as F
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=12, cin=C, n=2500, dup=100)
    rb = x.metadata.subm_rulebook(size, 3)
    g = torch.Generator().manual_seed(4)
    w1 = (torch.randn(27, C, C, generator=g) * (2.0 / (27 * C)) ** 0.5)
    w2 = (torch.randn(27, C, C, generator=g) * (2.0 / (27 * C)) ** 0.5)
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    gy = torch.randn(rb.n, C, generator=g).to(torch.bfloat16)
    X0 = x.features.detach().to(torch.bfloat16)

    def run(pair):
        F.WGRAD_PAIR = pair
        try:
            X = X0.clone().requires_grad_()
            ps = [t.to(gpu).requires_grad_() for t in (w1, b1, w2, b2)]
            y = F.ResidualBlockFunctionBF16.apply(X, ps[0], ps[1], ps[2], ps[3], x.metadata, size)
            gr = torch.autograd.grad(y, [X] + ps, gy.to(gpu))
            return y.detach(), gr
        finally:
            F.WGRAD_PAIR = True
    y_p, g_p = run(True)
    y_s, g_s = run(False)
    assert torch.equal(y_p, y_s) and torch.equal(g_p[0], g_s[0])
    for a, b in zip(g_p[1:], g_s[1:]):
        _close(a.float(), b.float(), 3e-6, "paired vs single weight gradients, bf16 storage")


@pytest.mark.parametrize("C,bf16", [(32, False), (64, False), (32, True), (128, True)])
def test_four_weight_gradients_in_one_launch(gpu, C, bf16):
    """scn_wgrad_bias_rules_n with four operand pairs on one rule list (what two stacked residual units would hand it)
    against four scn_wgrad_bias_rules calls: equal up to the order of the fp32 unit sums.  (The four-problem launch is NOT
    used by the module path: measured without gain over two two-problem launches)"""
    from sparse_rcnn_amd import functional as F
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=14, cin=C, n=2500, dup=100)
    r = x.metadata.subm_rulebook(size, 3).rules
    g = torch.Generator().manual_seed(8)
    dt = torch.bfloat16 if bf16 else torch.float32
    n = x.features.shape[0]
    Xs = [torch.randn(n, C, generator=g).to(gpu).to(dt) for _ in range(4)]
    Gs = [torch.randn(n, C, generator=g).to(gpu).to(dt) for _ in range(4)]
    dW, db = F.wgrad_bias_rules_n(Xs, Gs, r.in_rows, r.out_rows, r.prefix_host, 27, 1 << 13, L_F_RELU_IN())
    for q in range(4):
        dW1, db1 = F.wgrad_bias_rules(Xs[q], Gs[q], r.in_rows, r.out_rows, r.prefix_host, 27, 1 << 13, L_F_RELU_IN())
        _close(dW[q], dW1, 3e-6, f"problem {q}: dW")
        _close(db[q], db1, 3e-6, f"problem {q}: db")
    dW0, db0 = F.wgrad_bias_rules_n(Xs[:3], Gs[:3], r.in_rows, r.out_rows, r.prefix_host, 27, 0, 0)     # three, no bias, no ReLU
    assert db0 is None
    _close(dW0[2], F.wgrad_rules(Xs[2], Gs[2], r.in_rows, r.out_rows, r.prefix_host, 27, 0), 3e-6, "three problems")


def L_F_RELU_IN():
    from sparse_rcnn_amd import _lib as L
    return L.F_RELU_IN


@pytest.mark.parametrize("n", [1, 3, 20, 70])
def test_paired_weight_gradient_on_tiny_scenes(gpu, n):
    """Residual unit on scenes of a few voxels (most offsets have no rule; some problems' units are empty): the paired
    weight gradient equals the one-call-per-layer form and the oracle."""
    from sparse_rcnn_amd import functional as F
    C = 32
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=15, cin=C, n=n, dup=0, batch=1, grid=(6, 5, 4))
    g = torch.Generator().manual_seed(9)
    ws = [torch.randn(27, C, C, generator=g) * 0.05 for _ in range(2)]
    bs = [torch.randn(C, generator=g) * 0.1 for _ in range(2)]
    gy = torch.randn(x.features.shape[0], C, generator=g)

    def run(pair):
        F.WGRAD_PAIR = pair
        try:
            X = x.features.detach().clone().requires_grad_()
            ps = [ws[0].to(gpu).requires_grad_(), bs[0].to(gpu).requires_grad_(), ws[1].to(gpu).requires_grad_(), bs[1].to(gpu).requires_grad_()]
            y = F.ResidualBlockFunction.apply(X, *ps, x.metadata, size)
            return y.detach(), torch.autograd.grad(y, [X] + ps, gy.to(gpu))
        finally:
            F.WGRAD_PAIR = True
    y_p, g_p = run(True)
    y_s, g_s = run(False)
    assert torch.equal(y_p, y_s)
    for a, b in zip(g_p, g_s):
        _close(a, b, 2e-6, "paired vs single on a tiny scene")
    nbr, rules = O.subm_rulebook(scene.coords0, 3)
    Xo = x.features.detach().cpu().clone().requires_grad_()
    po = [ws[0].clone().requires_grad_(), bs[0].clone().requires_grad_(), ws[1].clone().requires_grad_(), bs[1].clone().requires_grad_()]
    h = O.conv(torch.relu(Xo), po[0], po[1], rules, scene.n(0))
    yo = Xo + O.conv(torch.relu(h), po[2], po[3], rules, scene.n(0))
    go = torch.autograd.grad(yo, [Xo] + po, gy)
    for a, b in zip(g_p, go):
        _close(a, b, 2e-4, "tiny scene gradients vs oracle")


# ---------------------------------------------------------------------------------------- N1: SparseGlobalPool / split_batch
POOL_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "globalpool_*.npz")))


@pytest.mark.parametrize("path", POOL_GOLDEN, ids=[os.path.basename(p) for p in POOL_GOLDEN])
def test_global_pool_and_split_batch_match_reference_golden(gpu, path):
    """sparse_rcnn_amd.SparseGlobalPool / split_batch (scn_segment_pool_*, scn_sample_counts) against fixtures produced by the
    reference's own SparseGlobalPool / split_batch (tests/golden/make_globalpool_golden.py) and against the oracle: rows
    grouped by sample and shuffled, an empty sample, a batch of zero samples; mean / sum / amax with tied maxima; forward
    within fp32 rounding of the reference's own summation (1e-6 of the scale), backward likewise, the split bit-exact."""
    scn = _scn()
    d = np.load(path)
    fn = getattr(torch, str(d["fn"]))
    bs = int(d["batch_size"])
    coords, feats = torch.from_numpy(d["coords"]), torch.from_numpy(d["feats"])
    fg = feats.to(gpu).requires_grad_()
    if bs == 0:
        md = scn.Metadata(3)
        md.n_samples = 0
        t = scn.SparseConvNetTensor(features=fg, metadata=md, spatial_size=torch.tensor([48, 48, 48]))
        assert scn.SparseGlobalPool(fn)(t).shape == (0, feats.shape[1]) and scn.split_batch(t) == []
        return
    t = scn.InputLayer(3, torch.tensor([48, 48, 48]), mode=0)((coords, fg, bs))
    assert torch.equal(t.features.detach().cpu(), feats)                   # mode 0: the fixture's rows are the tensor's rows
    y = scn.SparseGlobalPool(fn)(t)
    _close(y, torch.from_numpy(d["out"]), 1e-6, "pooled vs reference")
    _close(y, O.global_pool(feats, d["coords"], bs, fn), 1e-6, "pooled vs oracle")
    if str(d["fn"]) == "amax":
        assert torch.equal(y.detach().cpu(), torch.from_numpy(d["out"]))   # a maximum is exact
    y.backward(torch.from_numpy(d["gy"]).to(gpu))
    _close(fg.grad, torch.from_numpy(d["dfeats"]), 1e-6, "pool backward vs reference")
    parts = scn.split_batch(t.detach())
    assert [len(p) for p in parts] == d["split_rows"].tolist()
    assert torch.equal(torch.cat(parts).cpu(), torch.from_numpy(d["split_cat"]))
    # a pooling function the device pass does not know still works (the reference's formulation over the split)
    med = scn.SparseGlobalPool(lambda f, dim: torch.median(f, dim=dim).values)(t.detach())
    exp = torch.stack([torch.median(p, dim=0).values if len(p) else p.new_zeros(p.shape[1]) for p in
                       O.split_batch(feats, d["coords"], bs)])
    assert torch.equal(med.cpu(), exp)


def test_dense_rpn_stack_on_the_tile_kernels_equals_the_miopen_engine(gpu):
    """rpn.DenseRpn: a dense same-convolution is a submanifold convolution on a fully active grid.  The "tiles" engine (the
    volume as the channels-last slab of a fully active Metadata, 3^3 layers on k_conv_ts / k_wgrad, head on the row GEMM)
    against the "miopen" engine (scn.SparseToDense -> torch conv3d on NCDHW) with the same nn.Conv3d parameters: rpn_bbox /
    rpn_score and every gradient -- the sparse level's features, both dense convolutions, the head -- within 1e-4 of the
    scale (MIOpen's own accuracy), two samples, fp32; and against torch's CPU conv3d."""
    import copy
    from sparse_rcnn_amd.rpn import DenseRpn
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=77, cin=16, grid=(16, 12, 8), n=500, batch=2, dup=50)
    torch.manual_seed(9)
    net = DenseRpn(16, stride=8, width=8, num_dilations=2, keep_inside=False).to(gpu)       # (every anchor: the engines are the subject)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(torch.randn_like(p) * 0.1)
    res = {}
    for engine in ("tiles", "tiles-dense-first", "miopen"):
        net.engine = "tiles" if engine.startswith("tiles") else engine
        net.SPARSE_FIRST = engine == "tiles"          # (the first layer as one row GEMM over the active rows + a dilation gather)
        net.zero_grad()
        X = x.features.detach().clone().requires_grad_()
        t = scn.SparseConvNetTensor(features=X, metadata=x.metadata, spatial_size=x.spatial_size)
        bb, sc, an = net(t)
        g = torch.Generator().manual_seed(3)
        gb, gs = torch.randn(bb.shape, generator=g).to(gpu), torch.randn(sc.shape, generator=g).to(gpu)
        torch.autograd.backward([bb, sc], [gb, gs])
        res[engine] = [bb.detach(), sc.detach(), X.grad] + [p.grad.clone() for p in net.parameters()]
        assert an.shape == (16 * 12 * 8 * net.n_anchors, 2, 3) and bb.shape == (2, an.shape[0], 2, 3)
    # keep_inside (the default): the same outputs compacted to the anchors inside the scene, in anchor order (anchor.py:103-113,177-197)
    net.engine, net.SPARSE_FIRST, net.keep_inside = "tiles", True, True
    t = scn.SparseConvNetTensor(features=x.features.detach(), metadata=x.metadata, spatial_size=x.spatial_size)
    bb_i, sc_i, an_i = net(t)
    idx, _ = net.inside_for((16, 12, 8), gpu)
    assert 0 < len(idx) < an.shape[0] and torch.equal(an_i, an[idx])
    assert torch.equal(bb_i, res["tiles"][0][:, idx]) and torch.equal(sc_i, res["tiles"][1][:, idx])
    torch.cuda.synchronize()
    assert [int(f[0]) for f in sc_i.cell_flags] == [0]               # (scn_cell_map: no row outside the volume)
    net.keep_inside = False
    # torch on the CPU: the oracle of the dense layers
    cpu = copy.deepcopy(net).cpu()
    Xo = x.features.detach().cpu().clone().requires_grad_()
    dense = O.sparse_to_dense(Xo, scene.coords0, size.tolist(), 2)
    raw = cpu.head(cpu.stack(dense))
    raw = raw.view(2, cpu.n_anchors, 7, -1).permute(0, 3, 1, 2).reshape(2, -1, 7)
    ob, os_ = raw[..., :6].reshape(2, -1, 2, 3), raw[..., 6]
    cpu.zero_grad()
    torch.autograd.backward([ob, os_], [gb.cpu(), gs.cpu()])
    ref = [ob.detach(), os_.detach(), Xo.grad] + [p.grad for p in cpu.parameters()]
    names = ["rpn_bbox", "rpn_score", "d level features"] + ["d " + n for n, _ in net.named_parameters()]
    for n, a, a2, b, r in zip(names, res["tiles"], res["tiles-dense-first"], res["miopen"], ref):
        _close(a, r, 1e-4, f"tiles engine (sparse first layer) vs CPU: {n}")
        _close(a2, r, 1e-4, f"tiles engine (every layer on the volume) vs CPU: {n}")
        _close(b, r, 1e-4, f"miopen engine vs CPU: {n}")
    # bf16-stored level features: the two forms of the first layer against each other (different roundings: both round the
    # layer's output to bf16 once, the sparse form also its GEMM result)
    outs = []
    for sparse_first in (True, False):
        net.engine, net.SPARSE_FIRST = "tiles", sparse_first
        Xb = x.features.detach().to(torch.bfloat16)
        t = scn.SparseConvNetTensor(features=Xb, metadata=x.metadata, spatial_size=x.spatial_size)
        bb, sc, _ = net(t)
        outs.append((bb.float(), sc.float()))
    for u, v in zip(*outs):
        assert float((u - v).abs().max()) <= 3e-2 * max(float(v.abs().max()), 1e-6)


def test_roi_selector_takes_the_references_anchor_description_callable(gpu):
    """`RoiSelector.forward(rpn_bbox, rpn_score, anchor_description)` (proposal_selector.py:34-50): the reference hands over a
    callable that decodes and clips (`AnchorDescriptionMultiLevel.forward`, anchor.py:218-227).  Both forms -- the callable
    (decode every anchor, then select) and the anchors tensor + scene shape (select, then decode the selected) -- give the same
    proposals bit for bit."""
    from sparse_rcnn_amd import rpn as R
    g = torch.Generator().manual_seed(5)
    n, scene = 6000, (96.0, 80.0, 48.0)
    ctr = torch.rand(n, 3, generator=g) * torch.tensor(scene)
    size = torch.rand(n, 3, generator=g) * 20 + 4
    anchors = torch.stack([ctr, size], 1).to(gpu)
    rpn_bbox = (torch.randn(2, n, 2, 3, generator=g) * 0.2).to(gpu)
    rpn_score = torch.randn(2, n, generator=g).to(gpu)
    sel = R.RoiSelector(512, 40, 0.4)

    def anchor_description(deltas):                       # what the reference's module computes, as a plain callable
        return R.decode_boxes(anchors, deltas, scene)
    a = sel(rpn_bbox, rpn_score, anchors, scene)
    b = sel(rpn_bbox, rpn_score, anchor_description)
    for x, y in zip(a, b):
        assert len(x) == len(y) == 2
        for u, v in zip(x, y):
            assert torch.equal(u.cpu(), v.cpu())
    assert all(1 <= len(s) <= 40 for s in a[0])
    assert float(a[1][0].min()) >= 0.0 and float((a[1][0] - torch.tensor(scene, device=a[1][0].device)).max()) <= 0.0    # clipped


def test_tile_major_weight_gradient_agrees_with_the_rule_major_kernel(gpu):
    """scn_wgrad_tiles32 (round 6, experiment (c): one row gather per rule, dY tiles through LDS, offsets dealt to waves;
    profiles/r6_wgrad_one_gather.txt) computes the weight gradient of a 32 -> 32 SubM 3^3 layer (module_factory.py:396-414) from
    the forward kernel's tile tables: equal to an fp64 evaluation of the rules within 2e-6 of the scale, as the product kernel
    (scn_wgrad_rules) is; deterministic; with and without the fused input ReLU; a level whose last tile is ragged."""
    from sparse_rcnn_amd import _lib as L
    lib = L.lib()
    scn, coords, feats, fg, x, scene, size = _input(gpu, seed=31, cin=4, grid=(40, 36, 20), n=9003, batch=2, dup=200)
    rb = x.metadata.subm_rulebook(tuple(int(s) for s in size), 3)
    n, r, t = rb.n, rb.rules, rb.tiles
    assert n % 16 != 0
    g = torch.Generator(device="cuda").manual_seed(2)
    X = torch.randn(n, 32, device=gpu, generator=g)
    dY = torch.randn(n, 32, device=gpu, generator=g)
    ph = r.prefix_host
    s0 = torch.empty(lib.scn_wgrad_scratch_bytes(32, 32, ph, 27), dtype=torch.uint8, device=gpu)
    s1 = torch.empty(lib.scn_wgrad_tiles32_scratch_bytes(), dtype=torch.uint8, device=gpu)
    for relu in (0, 1):
        dW0 = torch.empty(27, 32, 32, device=gpu)
        dW1 = torch.full((27, 32, 32), float("nan"), device=gpu)
        L.check(lib.scn_wgrad_rules(L.ptr(X), 32, L.ptr(dY), 32, L.ptr(r.in_rows), L.ptr(r.out_rows), ph, 27, L.ptr(dW0), L.ptr(s0),
                                    relu, L.stream()))
        L.check(lib.scn_wgrad_tiles32(L.ptr(X), n, L.ptr(dY), n, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), ph, L.ptr(dW1),
                                      L.ptr(s1), relu, L.stream()))
        again = torch.empty_like(dW1)
        L.check(lib.scn_wgrad_tiles32(L.ptr(X), n, L.ptr(dY), n, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), ph, L.ptr(again),
                                      L.ptr(s1), relu, L.stream()))
        Xr = X.clamp_min(0) if relu else X
        ref = torch.zeros(27, 32, 32, device=gpu, dtype=torch.float64)
        for o in range(27):
            a, b = int(ph[o]), int(ph[o + 1])
            if b > a:
                ref[o] = Xr[r.in_rows[a:b].long()].double().t() @ dY[r.out_rows[a:b].long()].double()
        sc = float(ref.abs().max())
        assert torch.equal(dW1, again)
        assert float((dW1.double() - ref).abs().max()) <= 2e-6 * sc and float((dW0.double() - ref).abs().max()) <= 2e-6 * sc
