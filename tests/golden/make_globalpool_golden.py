"""Generates tests/golden/globalpool_*.npz by RUNNING the reference's own SparseGlobalPool / split_batch
(ndsis/modules/custom_operations.py:24-59; pure torch once a tensor object with .features / .batch_size() /
.get_spatial_locations() is supplied) on seeded inputs, forward and -- through torch autograd over the reference's code --
backward.  Run in the build container only (needs /root/reference):

    python tests/golden/make_globalpool_golden.py

`import sparseconvnet` is satisfied by this repository's package.  Only inputs and outputs are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, "/root/reference")
import sparse_rcnn_amd                                         # noqa: E402
sys.modules["sparseconvnet"] = sparse_rcnn_amd
from ndsis.modules.custom_operations import SparseGlobalPool, split_batch      # noqa: E402


class _Tensor:
    """What the reference's helpers touch of a SparseConvNetTensor."""

    def __init__(self, features, coords, batch_size):
        self.features, self._coords, self._bs = features, coords, batch_size

    def batch_size(self):
        return self._bs

    def get_spatial_locations(self):
        return self._coords


FUNCS = {"mean": torch.mean, "sum": torch.sum, "amax": torch.amax}


def case(name, seed, counts, c, fn, shuffle=False, ties=False):
    rng = np.random.default_rng(seed)
    b = np.concatenate([np.full(k, i) for i, k in enumerate(counts)]).astype(np.int64) if sum(counts) else np.zeros(0, np.int64)
    if shuffle:
        rng.shuffle(b)
    n = len(b)
    # unique sites per sample (an InputLayer in mode 0 keeps the rows as given, so the fixture's rows ARE the tensor's rows)
    xyz = np.zeros((n, 3), np.int64)
    for i in range(len(counts)):
        sel = np.nonzero(b == i)[0]
        lin = rng.choice(48 * 48 * 48, size=len(sel), replace=False)
        xyz[sel] = np.stack(np.unravel_index(lin, (48, 48, 48)), 1)
    coords = torch.from_numpy(np.concatenate([xyz, b[:, None]], 1).astype(np.int64))
    feats = rng.normal(size=(n, c)).astype(np.float32)
    if ties:                                     # repeated maxima inside a sample: torch.amax splits the gradient evenly
        feats = np.round(feats * 2) / 2
    x = torch.from_numpy(feats).requires_grad_()
    t = _Tensor(x, coords, len(counts))
    y = SparseGlobalPool(FUNCS[fn])(t)
    gy = torch.from_numpy(rng.normal(size=tuple(y.shape)).astype(np.float32))
    if n:
        y.backward(gy)
    parts = split_batch(_Tensor(x.detach(), coords, len(counts)))
    np.savez_compressed(os.path.join(HERE, f"globalpool_{name}.npz"), fn=np.array(fn), coords=coords.numpy(), feats=feats,
                        batch_size=np.array(len(counts)), out=y.detach().numpy(), gy=gy.numpy(),
                        dfeats=x.grad.numpy() if x.grad is not None else np.zeros_like(feats),
                        split_rows=np.array([len(p) for p in parts], np.int64),
                        split_cat=torch.cat(parts).numpy() if parts else np.zeros((0, c), np.float32))
    print(name, fn, "rows", n, "samples", len(counts), "out", tuple(y.shape))


if __name__ == "__main__":
    case("mean3", 1, [700, 450, 901], 32, "mean")
    case("sum_emptysample", 2, [300, 0, 520, 77], 23, "sum")
    case("amax_ties_unsorted", 3, [400, 380, 120], 16, "amax", shuffle=True, ties=True)
    case("mean_unsorted", 4, [1000, 1300], 7, "mean", shuffle=True)
    case("mean_nosamples", 5, [], 8, "mean")
