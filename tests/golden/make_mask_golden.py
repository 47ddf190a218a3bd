"""Generates tests/golden/mask_epilogue_*.npz by RUNNING the reference's own mask-head epilogue
(ndsis/modules/model.py SparseMaskPredictor, SparseMaskLossSelector) on the crops of
tests/golden/roi_crop_*.npz.  Run in the build container only (needs /root/reference):

    python tests/golden/make_mask_golden.py

`import sparseconvnet` is satisfied by this repository's package (the two classes are pure torch).
Only inputs and outputs are stored.
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
from ndsis.modules.model import SparseMaskPredictor, SparseMaskLossSelector      # noqa: E402


def case(name, seed, k=5, num_valid=0, thr=0.5):
    d = np.load(os.path.join(HERE, f"roi_crop_{name}.npz"))
    n_pts = int(d["n_pts"])
    is_inside = torch.from_numpy(np.unpackbits(d["is_inside"], axis=1)[:, :n_pts].astype(bool))
    counts = [int(c) for c in d["box_counts"]]
    coords = d["coords"]
    splits = [int((coords[:, 3] == b).sum()) for b in range(len(counts))]
    rng = np.random.default_rng(seed)
    m, bb = int(is_inside.sum()), is_inside.shape[0]
    scores = torch.from_numpy(rng.normal(size=(m, k)).astype(np.float32))
    classes = torch.from_numpy(rng.integers(-1, k, size=bb).astype(np.int64))
    pred = SparseMaskPredictor(num_valid)(scores, (is_inside, counts, splits), classes)
    # loss selector: per sample G_s ground-truth instances, overlaps decide which boxes are kept
    tuples, labels, gmasks = [], [], []
    for s, (nb, npts) in enumerate(zip(counts, splits)):
        g = 3 + s
        max_ov = torch.from_numpy(rng.uniform(0, 1, size=nb).astype(np.float32))
        arg_ov = torch.from_numpy(rng.integers(0, g, size=nb).astype(np.int64))
        tuples.append((None, None, max_ov, arg_ov))
        labels.append(torch.from_numpy(rng.integers(0, k, size=g).astype(np.int64)))
        gmasks.append(torch.from_numpy(rng.uniform(0, 1, size=(g, npts)) < 0.4))
    pm, gm, sl = SparseMaskLossSelector(thr)(scores, (is_inside, counts, splits), None, tuples, labels, gmasks)
    cat = lambda nested: torch.cat([t.reshape(-1).float() for sample in nested for t in sample]) \
        if any(len(sample) for sample in nested) else torch.zeros(0)
    out = dict(
        roi_case=np.array(name), k=np.array(k), num_valid=np.array(num_valid), thr=np.array(thr),
        box_counts=np.array(counts, np.int64), batch_splits=np.array(splits, np.int64),
        scores=scores.numpy(), classes=classes.numpy(),
        pred_masks=np.concatenate([p.numpy().reshape(-1) for p in pred]) if pred else np.zeros(0, np.float32),
        max_overlap=np.concatenate([t[2].numpy() for t in tuples]), argmax_overlap=np.concatenate([t[3].numpy() for t in tuples]),
        gt_counts=np.array([len(l) for l in labels], np.int64), gt_labels=np.concatenate([l.numpy() for l in labels]),
        gt_masks=np.concatenate([g.numpy().reshape(-1) for g in gmasks]),
        loss_pred=cat(pm).numpy(), loss_gt=cat(gm).numpy(),
        loss_rows=np.array([len(t) for sample in pm for t in sample], np.int64),
        loss_labels=torch.cat(list(sl)).numpy())
    np.savez_compressed(os.path.join(HERE, f"mask_epilogue_{name}.npz"), **out)
    print(name, "M", m, "BB", bb, "kept boxes", len(out["loss_rows"]), "pred", out["pred_masks"].shape)


if __name__ == "__main__":
    case("small", 10, k=5)
    case("emptybox", 11, k=4, num_valid=3)
    case("c23", 12, k=6, thr=0.3)
