"""Generates tests/golden/nms_*.npz by RUNNING the reference's own proposal selection
(ndsis/modules/proposal_selector.py ProposalSelector, ndsis/utils/bbox.py non_maximum_supression)
on seeded boxes.  Run in the build container only (needs /root/reference):

    python tests/golden/make_nms_golden.py
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
from ndsis.modules.proposal_selector import ProposalSelector                   # noqa: E402
from ndsis.utils.bbox import non_maximum_supression                            # noqa: E402


def case(name, seed, batch, n, pre, post, thr, grid=64.0, degenerate=False):
    rng = np.random.default_rng(seed)
    ctr = rng.uniform(0, grid, size=(batch, n, 3))
    # clustered centres so that many boxes overlap
    ctr = ctr[:, rng.integers(0, max(4, n // 12), size=n)] + rng.normal(0, 1.5, size=(batch, n, 3))
    edge = np.exp(rng.uniform(np.log(2), np.log(16), size=(batch, n, 3)))
    boxes = np.stack([ctr - edge / 2, ctr + edge / 2], 2).astype(np.float32)
    if degenerate:
        boxes[:, 3, 1] = boxes[:, 3, 0]                       # a zero-volume box
        boxes[:, 5] = boxes[:, 4]                              # an exact duplicate
    score = rng.normal(size=(batch, n)).astype(np.float32)
    sel = ProposalSelector(pre, post, thr)
    s, b, i = sel(torch.from_numpy(score), torch.from_numpy(boxes))
    order = torch.argsort(torch.from_numpy(score), dim=1, descending=True)
    sorted_boxes = torch.from_numpy(boxes)[torch.arange(batch)[:, None], order]
    keep = non_maximum_supression(sorted_boxes, thr)
    np.savez_compressed(os.path.join(HERE, f"nms_{name}.npz"), score=score, boxes=boxes, pre=np.array(pre), post=np.array(post),
                        thr=np.array(thr, np.float32), sorted_boxes=sorted_boxes.numpy(), keep=keep.numpy(),
                        out_counts=np.array([len(x) for x in s], np.int64),
                        out_scores=np.concatenate([x.numpy() for x in s]), out_boxes=np.concatenate([x.numpy() for x in b]),
                        out_index=np.concatenate([x.numpy() for x in i]))
    print(name, "kept", [int(k.sum()) for k in keep], "selected", [len(x) for x in s])


if __name__ == "__main__":
    case("small", 0, 2, 64, 32, 8, 0.3)
    case("pre1024", 1, 2, 3000, 1024, 200, 0.35)
    case("degenerate", 2, 1, 100, 0, 50, 0.2, degenerate=True)
