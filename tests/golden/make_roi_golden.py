"""Generates tests/golden/roi_crop_*.npz by RUNNING the reference's own sparse ROI crop
(ndsis/modules/roi_select_sparse.py roi_cut, roi_select_bbox_transform.py BBoxTransformerSlice)
on seeded synthetic inputs.  Run in the build container only (needs /root/reference):

    python tests/golden/make_roi_golden.py

The reference's top-level ``import sparseconvnet`` is satisfied by an empty placeholder module
(the crop functions themselves are pure torch).  Only inputs/outputs are stored.
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
sys.modules.setdefault("sparseconvnet", types.ModuleType("sparseconvnet"))
from ndsis.modules.roi_select_sparse import (roi_cut, SparseRoiCut, SparseRoiExtraCut,              # noqa: E402
                                             RawToRawFeatureExtractorCombiner,
                                             RawToFeaturesSceneFeatureExtractorCombiner)
from ndsis.modules.roi_select_bbox_transform import BBoxTransformerSlice   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def case(name, seed, n_pts, grid, batch, boxes_per_sample, c, empty_box=False, clip=False, resize=None):
    rng = np.random.default_rng(seed)
    coords = []
    for b in range(batch):
        p = rng.integers(0, grid, size=(n_pts, 3))
        coords.append(np.concatenate([p, np.full((n_pts, 1), b)], 1))
    coords = torch.from_numpy(np.concatenate(coords).astype(np.int64))
    feats = torch.from_numpy(rng.normal(size=(len(coords), c)).astype(np.float32))
    bbox_batch = []
    for b in range(batch):
        nb = boxes_per_sample[b]
        ctr = rng.uniform(0, grid, size=(nb, 3))
        edge = np.exp(rng.uniform(np.log(2), np.log(max(grid) * 0.8), size=(nb, 3)))
        start = ctr - edge / 2
        stop = ctr + edge / 2
        if empty_box and nb:
            start[0] = np.array(grid) + 5.25; stop[0] = start[0] + 3.5            # a box with no points
        bbox_batch.append(torch.from_numpy(np.stack([start, stop], 1).astype(np.float32)))
    tr = BBoxTransformerSlice(clip=clip, resize=resize)
    bbox_tensor, counts, assoc = tr(bbox_batch, torch.tensor(grid))
    new_coords, new_feats, is_inside = roi_cut(coords, feats, bbox_tensor, assoc)
    # the module level (roi_select_sparse.py:8-52): SparseRoiCut with the raw combiner, then SparseRoiExtraCut of a second
    # feature map with the same selection (what SparseFeaturemapSelectorBoth does, model.py:573-596)
    splits = [n_pts] * batch
    cut = SparseRoiCut(RawToRawFeatureExtractorCombiner(), clip_boxes=clip, resize_boxes=resize)
    (m_coords, m_feats, m_size, m_bs), selection = cut((coords, feats, torch.tensor(grid), batch, splits), bbox_batch)
    assert torch.equal(m_coords, new_coords) and torch.equal(m_feats, new_feats) and m_bs == len(is_inside)
    extra_in = torch.from_numpy(rng.normal(size=(len(coords), 4)).astype(np.float32))
    extra_out = SparseRoiExtraCut(RawToFeaturesSceneFeatureExtractorCombiner())(
        (coords, extra_in, torch.tensor(grid), batch, splits), selection)
    np.savez_compressed(
        os.path.join(HERE, f"roi_crop_{name}.npz"),
        coords=coords.numpy(), feats=feats.numpy(),
        boxes=torch.cat(bbox_batch).numpy() if bbox_batch else np.zeros((0, 2, 3), np.float32),
        box_counts=np.array(counts, np.int64), spatial_size=np.array(grid, np.int64), clip=np.array(clip),
        bbox_tensor=bbox_tensor.numpy(), assoc=assoc.numpy(),
        out_coords=new_coords.numpy(), out_feats=new_feats.numpy(),
        is_inside=np.packbits(is_inside.numpy(), axis=1), n_pts=np.array(len(coords)),
        resize=np.zeros(0, np.float32) if resize is None else np.atleast_1d(np.asarray(resize, np.float32)),
        extra_in=extra_in.numpy(), extra_out=extra_out.numpy(), splits=np.array(splits, np.int64))
    print(name, tuple(new_coords.shape), tuple(new_feats.shape), tuple(is_inside.shape))


if __name__ == "__main__":
    case("small", 0, 500, (24, 20, 16), 2, [3, 4], 5)
    case("emptybox", 1, 400, (16, 16, 16), 3, [2, 0, 3], 7, empty_box=True)
    case("clip", 2, 600, (32, 24, 8), 1, [6], 3, clip=True)
    case("c23", 3, 2000, (48, 48, 24), 2, [8, 8], 23)
    case("resize2", 4, 700, (24, 24, 12), 2, [5, 3], 4, resize=2.0)                  # Divider (roi_select_bbox_transform.py:15-21)
    case("resize_axes_clip", 5, 700, (20, 12, 6), 2, [4, 6], 4, clip=True, resize=(1.5, 2.0, 4.0))
