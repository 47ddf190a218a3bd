"""Pins the oracle's ROI crop against golden vectors produced by the reference's own
roi_cut / BBoxTransformerSlice (tests/golden/make_roi_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import scn_oracle as O

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "roi_crop_*.npz")))


def load(path):
    z = np.load(path)
    d = {k: z[k] for k in z.files}
    d["is_inside"] = np.unpackbits(d["is_inside"], axis=1)[:, :int(d["n_pts"])].astype(bool)
    counts = d["box_counts"].tolist()
    d["bbox_batch"] = np.split(d["boxes"], np.cumsum(counts)[:-1]) if counts else []
    return d


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_roi_crop_matches_reference(path):
    d = load(path)
    boxes, counts, assoc = O.transform_boxes(d["bbox_batch"], d["spatial_size"], bool(d["clip"]))
    assert np.array_equal(boxes, d["bbox_tensor"]) and np.array_equal(assoc, d["assoc"])
    assert counts == d["box_counts"].tolist()
    src, box_of, inside = O.roi_crop(d["coords"], boxes, assoc)
    assert np.array_equal(inside, d["is_inside"])
    out_coords = np.concatenate([d["coords"][src][:, :3], box_of[:, None]], 1)
    assert np.array_equal(out_coords, d["out_coords"])
    assert np.array_equal(d["feats"][src], d["out_feats"])


def test_golden_present():
    assert len(GOLDEN) >= 4
