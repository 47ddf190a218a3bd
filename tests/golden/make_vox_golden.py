"""Generates tests/golden/voxelize_*.npz by RUNNING the reference's own augment_coords
(ndsis/data/sparse_augmentation.py) on seeded point clouds.  Run in the build container only:

    python tests/golden/make_vox_golden.py

The random objects the reference draws (distortion matrix, sub-pixel offset, random cut-out start) are stored as
inputs next to the outputs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from ndsis.data import sparse_augmentation as SA                               # noqa: E402


def case(name, seed, n, scale, spatial_size=None, shift=None, random_cut=False):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    # a few planes in metres, like a ScanNet mesh
    pts = np.concatenate([rng.uniform(0, 6, size=(n // 2, 3)) * np.array([1, 1, 0.02]),
                          rng.uniform(0, 6, size=(n - n // 2, 3)) * np.array([1, 0.02, 0.5])]).astype(np.float32)
    coords = torch.from_numpy(pts)
    offset = torch.rand(3)
    kw = dict(coord_noise_sigma=0.05, theta=None, mirror=None)
    if random_cut:
        # the reference path: random_cut_out draws from torch's RNG; replay it to learn the start positions
        state = torch.get_rng_state()
        res, inside, size, augm, ortho = SA.augment_coords(coords, scale=scale, spatial_size=spatial_size,
                                                           max_empty_border_size_divisor=8, shift=None,
                                                           sub_pixel_offset=offset, **kw)
        torch.set_rng_state(state)
        ortho2 = SA.get_coord_distortion_matrix(coords.dtype, **kw)
        assert torch.equal(ortho, ortho2)
        aug = coords @ (ortho * scale)
        cs = -aug.min(0).values + offset
        start = (cs - augm["coords_shift"]).round().long()
    else:
        res, inside, size, augm, ortho = SA.augment_coords(coords, scale=scale, spatial_size=spatial_size,
                                                           max_empty_border_size_divisor=None, shift=shift,
                                                           sub_pixel_offset=offset, **kw)
        start = None
    np.savez_compressed(
        os.path.join(HERE, f"voxelize_{name}.npz"), coords=pts, rot_and_scale=(ortho * scale).numpy(), offset=offset.numpy(),
        spatial_size=np.array(-1 if spatial_size is None else spatial_size), shift=np.array(-1 if shift is None else shift),
        has_shift=np.array(shift is not None), start=np.array(-1) if start is None else start.numpy(),
        out_coords=res.numpy(), is_inside=inside.numpy(), out_size=torch.as_tensor(size).numpy(),
        out_shift=augm["coords_shift"].numpy())
    print(name, tuple(res.shape), int(inside.sum()), torch.as_tensor(size).tolist())


if __name__ == "__main__":
    case("nocut", 0, 6000, 50.0)
    case("nocut_shift", 1, 5000, 50.0, shift=3)
    case("fixcut", 2, 8000, 25.0, spatial_size=(128, 128, 64), shift=0)
    case("fixcut_shift", 3, 8000, 20.0, spatial_size=(96, 128, 64), shift=5)
    # random_cut_out (sparse_augmentation.py:50-78) raises on torch 2.10 ("is_inside[is_inside] = ..." aliasing), so the
    # start_positions mode of the device path is pinned by the oracle only
