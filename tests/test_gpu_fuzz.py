"""-m gpu: seeded random sweep of layer shapes against the CPU oracle -- channel counts that are not multiples of 4, 16
or 32 (the kernels' vector width, MFMA tile and K-chunk), row counts around the tile sizes, thin grids, empty samples,
with and without bias / input ReLU.  The cases are fixed by their seeds; a failure names the drawn configuration."""
import numpy as np
import pytest
import torch

from oracle import scn_oracle as O

pytestmark = pytest.mark.gpu

FEAT_TOL = 1e-4          # BASELINE.json north_star: "features within 1e-4 fp32" (relative to the output scale)


def _draw(seed):
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


@pytest.mark.parametrize("seed", range(24))
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


@pytest.mark.parametrize("seed", range(100, 112))
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
    g = torch.randn(yo.shape, generator=torch.Generator().manual_seed(seed + 7))
    got = torch.autograd.grad(y, (x.features, down.weight, down.bias, convs[0].weight, convs[0].bias, convs[1].weight,
                                  convs[1].bias, up.weight), g.to(gpu))
    exp = torch.autograd.grad(yo, (Xo, Wd, bd, W1, b1, W2, b2, Wu), g)
    for a, e, name in zip(got, exp, ("dX", "dWd", "dbd", "dW1", "db1", "dW2", "db2", "dWu")):
        _close(a, e, name, cfg)
