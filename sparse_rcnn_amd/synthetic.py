"""Deterministic synthetic ScanNet-shaped voxel clouds (SURVEY.md §8d "Synthetic inputs").

Produces the reference's input contract (ndsis/data/data.py:88-115 ``collate_fn``):
``coords`` int64 CPU ``[Npts,4]`` = (x,y,z,batch), sample-major, duplicates allowed;
``feats`` fp32 ``[Npts,7]`` (3 colour in [-1,1], 1 ones, 3 unit normal); ``spatial_size`` int64 [3].
"""
from __future__ import annotations

import numpy as np
import torch


def _surface_voxels(rng, grid, n_cuboids=20, n_planes=3, scale=0.9):
    """Room shell + furniture shells + tilted planes, 1 voxel thick with +-1 jitter.  `scale` = room extent as a
    fraction of the grid (the room sits in the grid centre), chosen by the caller so that the raw surface has about
    the target number of voxels -- surfaces stay dense (ScanNet-like ~9-10 rules per voxel), not thinned noise."""
    gx, gy, gz = grid
    pts = []
    x0, x1 = int(gx * (0.5 - scale / 2)), int(gx * (0.5 + scale / 2))
    y0, y1 = int(gy * (0.5 - scale / 2)), int(gy * (0.5 + scale / 2))
    z0, z1 = int(gz * (0.5 - scale / 2)), int(gz * (0.5 + scale / 2))
    gx, gy, gz = max(x1 - x0, 8), max(y1 - y0, 8), max(z1 - z0, 8)        # furniture sizes relative to the room
    xs, ys = np.meshgrid(np.arange(x0, x1), np.arange(y0, y1), indexing="ij")
    pts.append(np.stack([xs.ravel(), ys.ravel(), np.full(xs.size, z0)], 1))
    xs, zs = np.meshgrid(np.arange(x0, x1), np.arange(z0, z1), indexing="ij")
    for y in (y0, y1 - 1):
        pts.append(np.stack([xs.ravel(), np.full(xs.size, y), zs.ravel()], 1))
    ys, zs = np.meshgrid(np.arange(y0, y1), np.arange(z0, z1), indexing="ij")
    for x in (x0, x1 - 1):
        pts.append(np.stack([np.full(ys.size, x), ys.ravel(), zs.ravel()], 1))
    # furniture: cuboid shells standing on the floor
    for _ in range(n_cuboids):
        ex = rng.integers(max(4, gx // 32), max(6, gx // 6), size=3)
        ex[2] = min(ex[2], (z1 - z0) // 2)
        cx = rng.integers(x0 + 1, max(x0 + 2, x1 - ex[0] - 1))
        cy = rng.integers(y0 + 1, max(y0 + 2, y1 - ex[1] - 1))
        lo = np.array([cx, cy, z0 + 1]); hi = lo + ex
        a, b = np.meshgrid(np.arange(lo[0], hi[0]), np.arange(lo[1], hi[1]), indexing="ij")
        pts.append(np.stack([a.ravel(), b.ravel(), np.full(a.size, hi[2])], 1))            # top
        a, c = np.meshgrid(np.arange(lo[0], hi[0]), np.arange(lo[2], hi[2]), indexing="ij")
        for y in (lo[1], hi[1] - 1):
            pts.append(np.stack([a.ravel(), np.full(a.size, y), c.ravel()], 1))
        b, c = np.meshgrid(np.arange(lo[1], hi[1]), np.arange(lo[2], hi[2]), indexing="ij")
        for x in (lo[0], hi[0] - 1):
            pts.append(np.stack([np.full(b.size, x), b.ravel(), c.ravel()], 1))
    # a few tilted planes
    for _ in range(n_planes):
        w = rng.integers(gx // 8, gx // 3); h = rng.integers(gy // 8, gy // 3)
        ox = rng.integers(x0, max(x0 + 1, x1 - w)); oy = rng.integers(y0, max(y0 + 1, y1 - h))
        sx, sy = rng.uniform(-0.4, 0.4, 2)
        a, b = np.meshgrid(np.arange(w), np.arange(h), indexing="ij")
        z = (z0 + (z1 - z0) * 0.4 + sx * a + sy * b).astype(np.int64)
        pts.append(np.stack([ox + a.ravel(), oy + b.ravel(), z.ravel()], 1))
    p = np.concatenate(pts).astype(np.int64)
    p += rng.integers(-1, 2, size=p.shape) * (rng.random(p.shape) < 0.15)                    # +-1 voxel jitter
    p = np.clip(p, 0, np.array(grid) - 1)
    p = np.unique(p, axis=0)
    return p[(p[:, 0] >= x0) & (p[:, 0] < x1) & (p[:, 1] >= y0) & (p[:, 1] < y1)]


def make_scene(grid=(512, 512, 256), target_active=150_000, dup=1.15, seed=1, uniform=False):
    """One sample.  Returns (coords int64 [Npts,3], feats fp32 [Npts,7], n_active)."""
    rng = np.random.default_rng(seed)
    grid = tuple(int(g) for g in grid)
    if uniform:
        lin = rng.choice(grid[0] * grid[1] * grid[2], size=target_active, replace=False)
        vox = np.stack(np.unravel_index(lin, grid), 1).astype(np.int64)
    else:
        lo, hi = 0.05, 0.98                                          # bisect the room scale for ~1.04 x target voxels
        state = rng.bit_generator.state
        for _ in range(14):
            mid = 0.5 * (lo + hi)
            rng.bit_generator.state = state
            vox = _surface_voxels(rng, grid, scale=mid)
            if len(vox) < 1.04 * target_active:
                lo = mid
            else:
                hi = mid
        rng.bit_generator.state = state
        vox = _surface_voxels(rng, grid, scale=hi)
        tries = 0
        while len(vox) < target_active and tries < 8:                                       # grid too small: densify
            extra = _surface_voxels(rng, grid, n_cuboids=40, n_planes=6, scale=0.98)
            vox = np.unique(np.concatenate([vox, extra]), axis=0)
            tries += 1
        if len(vox) > target_active:
            vox = vox[np.sort(rng.choice(len(vox), size=target_active, replace=False))]
    n_active = len(vox)
    n_dup = int(round(n_active * (dup - 1.0)))
    pts = np.concatenate([vox, vox[rng.integers(0, n_active, size=n_dup)]]) if n_dup else vox
    # mesh-vertex-like order: spatially coherent blocks, shuffled inside a block
    key = (pts[:, 0] // 16) * 1_000_000 + (pts[:, 1] // 16) * 1000 + pts[:, 2] // 16
    pts = pts[np.lexsort((rng.random(len(pts)), key))]
    colour = rng.uniform(-1, 1, size=(len(pts), 3))
    normal = rng.normal(size=(len(pts), 3)); normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    feats = np.concatenate([colour, np.ones((len(pts), 1)), normal], 1).astype(np.float32)
    return pts, feats, n_active


def make_batch(n_scenes=1, grid=(512, 512, 256), target_active=150_000, dup=1.15, seed=1, uniform=False,
               cin=7):
    """collate_fn-shaped batch: (coords [Npts,4] int64 CPU, feats [Npts,cin] fp32, spatial_size [3] int64,
    batch_size, batch_splits)."""
    cs, fs, splits = [], [], []
    for b in range(n_scenes):
        p, f, _ = make_scene(grid, target_active, dup, seed + b, uniform)
        cs.append(np.concatenate([p, np.full((len(p), 1), b, np.int64)], 1))
        fs.append(f[:, :cin] if cin <= 7 else np.tile(f, (1, (cin + 6) // 7))[:, :cin])
        splits.append(len(p))
    coords = torch.from_numpy(np.concatenate(cs))
    feats = torch.from_numpy(np.ascontiguousarray(np.concatenate(fs)))
    return coords, feats, torch.tensor(grid, dtype=torch.long), n_scenes, splits


def make_boxes(coords, n_boxes=64, seed=3, lo=8.0, hi=96.0):
    """cfg-3 boxes: centres on random points, edge lengths log-uniform lo..hi voxels, fp32 non-integer corners.
    Returns a list (one per sample) of fp32 [n,2,3] tensors (start, stop)."""
    rng = np.random.default_rng(seed)
    c = coords.numpy() if isinstance(coords, torch.Tensor) else np.asarray(coords)
    out = []
    for b in range(int(c[:, 3].max()) + 1 if len(c) else 0):
        pts = c[c[:, 3] == b][:, :3]
        ctr = pts[rng.integers(0, len(pts), size=n_boxes)].astype(np.float64)
        edge = np.exp(rng.uniform(np.log(lo), np.log(hi), size=(n_boxes, 3)))
        start = ctr - edge / 2 + rng.uniform(0, 1, size=(n_boxes, 3))
        stop = start + edge
        out.append(torch.from_numpy(np.stack([start, stop], 1).astype(np.float32)))
    return out
