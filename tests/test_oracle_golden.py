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
    resize = d["resize"] if len(d["resize"]) else None
    boxes, counts, assoc = O.transform_boxes(d["bbox_batch"], d["spatial_size"], bool(d["clip"]), resize)
    assert np.array_equal(boxes, d["bbox_tensor"]) and np.array_equal(assoc, d["assoc"])
    assert counts == d["box_counts"].tolist()
    src, box_of, inside = O.roi_crop(d["coords"], boxes, assoc)
    assert np.array_equal(inside, d["is_inside"])
    out_coords = np.concatenate([d["coords"][src][:, :3], box_of[:, None]], 1)
    assert np.array_equal(out_coords, d["out_coords"])
    assert np.array_equal(d["feats"][src], d["out_feats"])
    assert np.array_equal(d["extra_in"][src], d["extra_out"])                 # SparseRoiExtraCut (roi_select_sparse.py:8-26)


def test_golden_present():
    assert len(GOLDEN) >= 6


# ---- N2: mask-head epilogue, pinned by the reference's SparseMaskPredictor / SparseMaskLossSelector ------------------
MASK_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mask_epilogue_*.npz")))


def load_mask(path):
    z = np.load(path)
    d = {k: z[k] for k in z.files}
    roi = load(os.path.join(os.path.dirname(path), f"roi_crop_{str(d['roi_case'])}.npz"))
    d["is_inside"] = roi["is_inside"]
    counts, splits = d["box_counts"].tolist(), d["batch_splits"].tolist()
    thr = float(d["thr"])
    mo = np.split(d["max_overlap"], np.cumsum(counts)[:-1])
    ao = np.split(d["argmax_overlap"], np.cumsum(counts)[:-1])
    d["keep_list"] = [m >= thr for m in mo]                                   # LossFilter (model.py:1017-1032)
    d["assoc_list"] = [a[k] for a, k in zip(ao, d["keep_list"])]
    g = d["gt_counts"].tolist()
    d["labels_list"] = np.split(d["gt_labels"], np.cumsum(g)[:-1])
    sizes = [gi * s for gi, s in zip(g, splits)]
    d["masks_list"] = [m.reshape(gi, s) for m, gi, s in zip(np.split(d["gt_masks"], np.cumsum(sizes)[:-1]), g, splits)]
    return d


@pytest.mark.parametrize("path", MASK_GOLDEN, ids=[os.path.basename(p) for p in MASK_GOLDEN])
def test_oracle_mask_epilogue_matches_reference(path):
    d = load_mask(path)
    counts, splits = d["box_counts"].tolist(), d["batch_splits"].tolist()
    pred = O.mask_predict(d["scores"], d["is_inside"], counts, splits, d["classes"], int(d["num_valid"]))
    assert np.allclose(np.concatenate([p.reshape(-1) for p in pred]), d["pred_masks"], rtol=0, atol=1e-6)
    p, g, rows, labels = O.mask_loss_select(d["scores"], d["is_inside"], counts, splits, d["keep_list"], d["assoc_list"],
                                            d["labels_list"], d["masks_list"])
    assert np.array_equal(p, d["loss_pred"]) and np.array_equal(g, d["loss_gt"])
    assert rows == d["loss_rows"].tolist() and np.array_equal(labels, d["loss_labels"])


def test_mask_golden_present():
    assert len(MASK_GOLDEN) >= 3


# ---- N3: greedy NMS, pinned by the reference's non_maximum_supression / ProposalSelector ------------------------------
NMS_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "nms_*.npz")))


@pytest.mark.parametrize("path", NMS_GOLDEN, ids=[os.path.basename(p) for p in NMS_GOLDEN])
def test_oracle_nms_matches_reference(path):
    z = np.load(path)
    for boxes, keep in zip(z["sorted_boxes"], z["keep"]):
        assert np.array_equal(O.nms(boxes, float(z["thr"])), keep)


def test_nms_golden_present():
    assert len(NMS_GOLDEN) >= 3


# ---- N4: voxelisation, pinned by the reference's augment_coords ------------------------------------------------------
VOX_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "voxelize_*.npz")))


def vox_args(z):
    size = None if z["spatial_size"].ndim == 0 else z["spatial_size"]
    shift = int(z["shift"]) if bool(z["has_shift"]) else None
    return size, shift


@pytest.mark.parametrize("path", VOX_GOLDEN, ids=[os.path.basename(p) for p in VOX_GOLDEN])
def test_oracle_voxelisation_matches_reference(path):
    z = np.load(path)
    size, shift = vox_args(z)
    res, inside, out_size, cshift = O.augment_coords(z["coords"], z["rot_and_scale"], z["offset"], size, shift)
    assert np.array_equal(res, z["out_coords"]) and np.array_equal(inside, z["is_inside"])
    assert np.array_equal(out_size, z["out_size"]) and np.array_equal(cshift, z["out_shift"])


def test_vox_golden_present():
    assert len(VOX_GOLDEN) >= 4


# ---- N1: SparseGlobalPool / split_batch, pinned by the reference's own classes (custom_operations.py:24-59) -------------
POOL_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "globalpool_*.npz")))
POOL_FN = {"mean": "mean", "sum": "sum", "amax": "amax"}


@pytest.mark.parametrize("path", POOL_GOLDEN, ids=[os.path.basename(p) for p in POOL_GOLDEN])
def test_oracle_global_pool_matches_reference(path):
    import torch
    d = np.load(path)
    fn = getattr(torch, POOL_FN[str(d["fn"])])
    x = torch.from_numpy(d["feats"]).requires_grad_()
    bs = int(d["batch_size"])
    y = O.global_pool(x, d["coords"], bs, fn)
    assert tuple(y.shape) == d["out"].shape and np.array_equal(y.detach().numpy(), d["out"])
    if len(d["feats"]):
        y.backward(torch.from_numpy(d["gy"]))
        assert np.array_equal(x.grad.numpy(), d["dfeats"])
    parts = O.split_batch(x.detach(), d["coords"], bs)
    assert [len(p) for p in parts] == d["split_rows"].tolist()
    if parts:
        assert np.array_equal(torch.cat(parts).numpy(), d["split_cat"])


def test_global_pool_golden_present():
    assert len(POOL_GOLDEN) >= 5
