"""Generates tests/golden/dense_twin_*.npz by RUNNING the dense-mode layers the reference's own factories pair with every
scn layer (ndsis/modules/module_factory.py, `sparse=False` branches: get_same_convolution :396-414, get_channel_changer
:377-393, get_channel_changer_or_identity :357-374, get_downsampler :221-241, get_upsampler :244-271,
get_batchnorm_leaky_relu :92-113) ONCE on a densified seeded scene: inputs on the active sites, the layer's parameters in torch's
layout, the output and -- for a seeded upstream gradient that is non-zero on the active sites only -- the gradients of the input
(sampled on the active sites) and of the parameters.

What this pins: the reference's own statement of what each sparse layer computes.  What it does NOT pin:
SparseConvNet's conventions (offset enumeration and weight layout of `scn.*Convolution.weight`, the retain-fraction meaning of
the BatchNorm momentum, rule order) -- that library is absent; the mapping torch layout -> scn layout used by the tests is this
repository's.  Run in the build container only (needs /root/reference):

    python tests/golden/make_dense_twin_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, "/root/reference")
import sparse_rcnn_amd                                         # noqa: E402
sys.modules["sparseconvnet"] = sparse_rcnn_amd                 # (module_factory imports it; the sparse branches are never taken)
from ndsis.modules import module_factory as MF                 # noqa: E402


def cloud(seed, grid, n, batch):
    """Unique active sites per sample, shuffled: int64 [N, 4] = (x, y, z, batch), batch-major."""
    rng = np.random.default_rng(seed)
    cs = []
    for b in range(batch):
        lin = rng.choice(grid[0] * grid[1] * grid[2], size=n, replace=False)
        p = np.stack(np.unravel_index(lin, grid), 1)
        rng.shuffle(p)
        cs.append(np.concatenate([p, np.full((len(p), 1), b)], 1))
    return np.concatenate(cs).astype(np.int64)


def dense_of(X, coords, grid, batch):
    out = torch.zeros(batch, X.shape[1], *grid, dtype=X.dtype)
    c = torch.from_numpy(coords)
    out[c[:, 3], :, c[:, 0], c[:, 1], c[:, 2]] = X
    return out


def sample(D, coords):
    c = torch.from_numpy(coords)
    return D[c[:, 3], :, c[:, 0], c[:, 1], c[:, 2]]


def coarse_sites(coords, stride):
    """The cells of the strided grid with at least one active child, in order of first occurrence (this repository's canonical
    numbering; any order would do for the fixture: the rows carry their coordinates)."""
    cc = coords.copy()
    cc[:, :3] //= stride
    _, first = np.unique(cc, axis=0, return_index=True)
    return cc[np.sort(first)]


def run(name, layer, coords_in, grid_in, coords_out, grid_out, cin, seed, batch, extra=None):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in layer.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    X = torch.randn(len(coords_in), cin, generator=g).requires_grad_()
    D = dense_of(X, coords_in, grid_in, batch)
    Yd = layer(D)
    assert tuple(Yd.shape[2:]) == tuple(grid_out), (Yd.shape, grid_out)
    Y = sample(Yd, coords_out)
    dY = torch.randn(Y.shape, generator=g)
    params = list(layer.parameters())
    grads = torch.autograd.grad(Y, [X] + params, dY)
    out = dict(coords_in=coords_in, coords_out=coords_out, grid_in=np.array(grid_in, np.int64),
               grid_out=np.array(grid_out, np.int64), batch=np.array(batch), X=X.detach().numpy(), Y=Y.detach().numpy(),
               dY=dY.numpy(), dX=grads[0].numpy())
    for (pn, p), gp in zip(layer.named_parameters(), grads[1:]):
        out["param_" + pn] = p.detach().numpy()
        out["grad_" + pn] = gp.numpy()
    out.update(extra or {})
    np.savez_compressed(os.path.join(HERE, f"dense_twin_{name}.npz"), **out)
    print(name, type(layer).__name__, "in", X.shape, "out", Y.shape, [k for k in out if k.startswith("param_")])


if __name__ == "__main__":
    grid, batch = (12, 10, 8), 2
    pts = cloud(0, grid, 140, batch)
    # SubmanifoldConvolution 3^3 <-> Conv3d(k 3, padding 1)
    sp, stride, ch, layer = MF.get_same_convolution(3, False, 5, 7, kernel_size=3)
    assert not sp and ch == 7
    run("same_conv3", layer, pts, grid, pts, grid, 5, 1, batch)
    # SubmanifoldConvolution 1^3 (channel changer) <-> Conv3d(k 1)
    sp, stride, ch, layer = MF.get_channel_changer(3, False, 5, 6, kernel_size=1)
    run("channel_changer1", layer, pts, grid, pts, grid, 5, 2, batch)
    # NetworkInNetwork <-> Conv3d(k 1)
    sp, stride, ch, layer = MF.get_channel_changer_or_identity(3, False, 6, 4)
    run("network_in_network", layer, pts, grid, pts, grid, 6, 3, batch)
    # Convolution 2^3 / 2 <-> Conv3d(k 2, stride 2): sampled on the coarse cells with an active child
    cgrid = tuple(s // 2 for s in grid)
    cpts = coarse_sites(pts, 2)
    sp, stride, ch, layer = MF.get_downsampler(3, False, 5, 8, stride=2)
    run("downsampler2", layer, pts, grid, cpts, cgrid, 5, 4, batch)
    # Deconvolution 2^3 / 2 <-> ConvTranspose3d(k 2, stride 2): from the coarse sites back to the fine active set
    sp, stride, ch, layer = MF.get_upsampler(3, False, 8, 5, stride=2)
    run("upsampler2", layer, cpts, cgrid, pts, grid, 8, 5, batch)
    # BatchNorm(Leaky)ReLU <-> BatchNormNd + (Leaky)ReLU, training mode.  Dense batch statistics run over EVERY cell, the sparse
    # layer's over the active rows: the twins agree on a fully active grid, which is what the fixture uses.
    fgrid = (4, 3, 2)
    full = np.stack(np.meshgrid(np.arange(fgrid[0]), np.arange(fgrid[1]), np.arange(fgrid[2]), np.arange(batch), indexing="ij"),
                    -1).reshape(-1, 4).astype(np.int64)
    full = full[np.lexsort((full[:, 2], full[:, 1], full[:, 0], full[:, 3]))]
    for leak in (0, 0.2):
        sp, stride, ch, layer = MF.get_batchnorm_leaky_relu(3, False, 6, eps=1e-4, momentum=0.9, leakiness=leak)
        layer.train()
        run(f"batchnorm_leaky{str(leak).replace('.', 'p')}", layer, full, fgrid, full, fgrid, 6, 6, batch,
            extra=dict(eps=np.array(1e-4), leakiness=np.array(float(leak))))
