"""Generates tests/golden/anchor_*.npz by RUNNING the reference's own anchor bookkeeping
(ndsis/modules/anchor.py AnchorDescriptionMultiLevel: pixel-wise anchors, `inside_indicator`, `rpn_permuter` +
`rpn_bbox_score_splitter`, and `forward` = bbox_transform_inv + clip_boxes) on seeded head outputs.  Run in the build container
only (needs /root/reference):

    python tests/golden/make_anchor_golden.py
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
from ndsis.modules.anchor import AnchorDescriptionMultiLevel   # noqa: E402
from sparse_rcnn_amd import rpn as R                           # noqa: E402  (anchor tables only)


def case(name, seed, scene_shape, conv_shapes, strides, anchor_levels, batch=2, border=0):
    g = torch.Generator().manual_seed(seed)
    raw_levels = [torch.tensor(a, dtype=torch.float32) for a in anchor_levels]
    stride_levels = torch.tensor([[float(s)] * 3 for s in strides])
    desc = AnchorDescriptionMultiLevel(tuple(scene_shape), [tuple(c) for c in conv_shapes], raw_levels, stride_levels,
                                       allowed_border=border)
    # head outputs as the reference's conv heads deliver them: [batch, A * 7, X, Y, Z] per level
    heads = [torch.randn((batch, len(a) * 7) + tuple(c), generator=g) * 0.3 for a, c in zip(anchor_levels, conv_shapes)]
    bbox, score = desc.rpn_bbox_score_splitter(desc.rpn_permuter(heads))
    boxes = desc(bbox)
    out = dict(scene_shape=np.array(scene_shape, np.int64), strides=np.array(strides, np.int64), border=np.array(border),
               inside_indicator=desc.inside_indicator.numpy(), inside_anchors=desc.inside_anchors.numpy(),
               rpn_bbox=bbox.numpy(), rpn_score=score.numpy(), boxes=boxes.numpy(), n_levels=np.array(len(heads)))
    for l, (h, a, c) in enumerate(zip(heads, anchor_levels, conv_shapes)):
        out[f"head{l}"] = h.numpy()
        out[f"anchors{l}"] = np.array(a, np.float32)
        out[f"conv_shape{l}"] = np.array(c, np.int64)
    np.savez_compressed(os.path.join(HERE, f"anchor_{name}.npz"), **out)
    print(name, "anchors", len(desc.inside_indicator), "inside", int(desc.inside_indicator.sum()),
          "clipped", int((boxes != R.decode_boxes(desc.inside_anchors, bbox)).sum()))


if __name__ == "__main__":
    case("one_level", 0, (64, 48, 32), [(8, 6, 4)], [8], [R.DEFAULT_ANCHORS])
    case("two_levels", 1, (32, 32, 16), [(8, 8, 4), (4, 4, 2)], [4, 8], R.REF_ANCHOR_LEVELS_VOXELS, batch=3)
    case("border", 2, (24, 24, 24), [(3, 3, 3)], [8], [((6.0, 6.0, 6.0), (20.0, 12.0, 9.0))], batch=1, border=2)
