"""The C++/OpenMP CPU restatement (oracle/scn_cpu_baseline.cpp -- bench.py's timed `cpu_baseline`) against the Python
oracle: same rulebook sizes, forward output and every gradient of the A12 U-Net.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import cpu_baseline as CB
from oracle import scn_oracle as O


def _cloud(seed, grid, n, batch, dup):
    rng = np.random.default_rng(seed)
    cs = []
    for b in range(batch):
        lin = rng.choice(grid[0] * grid[1] * grid[2], size=n, replace=False)
        p = np.stack(np.unravel_index(lin, grid), 1)
        p = np.concatenate([p, p[rng.integers(0, n, size=dup)]])
        rng.shuffle(p)
        cs.append(np.concatenate([p, np.full((len(p), 1), b)], 1))
    return np.concatenate(cs).astype(np.int64)


@pytest.mark.parametrize("channels,threads", [((8, 16, 24), 1), ((16, 32), 4), ((32, 64, 128), 3)])
def test_cpp_restatement_matches_python_oracle(channels, threads):
    coords = _cloud(3, (16, 16, 8), 500, 2, 60)
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(len(coords), 7, generator=g)
    params = {k: v.requires_grad_() for k, v in O.init_unet_params(7, list(channels), seed=4).items()}
    scene = O.OracleScene(coords)
    fo = feats.clone().requires_grad_()
    out = O.unet_forward(scene, fo, params, list(channels))
    dY = torch.randn(out.shape, generator=g)
    out.backward(dY)
    res = CB.unet_step(coords, feats.numpy(), channels, CB.flat_params(params, 7, channels), dY.numpy(), threads=threads)
    assert res["n_active"] == scene.n(0)
    assert res["n_rules0"] == sum(len(r[0]) for r in scene.subm_rules(0, 3))
    scale = max(1.0, out.abs().max().item())
    assert np.abs(res["out"] - out.detach().numpy()).max() / scale < 1e-5
    gd = CB.unflatten(res["grads"], 7, channels)
    for k, p in params.items():
        e = p.grad.numpy()
        err = np.abs(gd[k] - e).max() / max(1.0, np.abs(e).max())
        assert err < 2e-5, (k, err)
    e = fo.grad.numpy()
    assert np.abs(res["dfeats"] - e).max() / max(1.0, np.abs(e).max()) < 2e-5


def test_rejects_out_of_range_coordinates():
    coords = np.array([[0, 0, 70000, 0]], dtype=np.int64)
    with pytest.raises(RuntimeError):
        CB.unet_step(coords, np.ones((1, 7), np.float32), (8,), CB.flat_params(O.init_unet_params(7, [8]), 7, (8,)))
