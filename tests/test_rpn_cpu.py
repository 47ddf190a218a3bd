"""CPU: the anchor bookkeeping of the RPN boundary (rpn.py) against fixtures produced by the reference's own
AnchorDescriptionMultiLevel (ndsis/modules/anchor.py:57-227; tests/golden/make_anchor_golden.py): pixel-wise anchors in
spatial-major / anchor-minor order, the inside-the-scene indicator (`allowed_border`), the compaction of head outputs
(`rpn_permuter` + `rpn_bbox_score_splitter`) and the decode + clip of `forward` -- bit for bit."""
import os

import numpy as np
import pytest
import torch

from sparse_rcnn_amd import rpn as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, f"anchor_{name}.npz"))


def _raw_of_head(head, n_anchors):
    """[B, A * 7, X, Y, Z] -> [B, X Y Z A, 7]: what DenseRpn.forward hands to `_finish` (spatial-major, anchor-minor)."""
    B = head.shape[0]
    return head.view(B, n_anchors, 7, -1).permute(0, 3, 1, 2).reshape(B, -1, 7)


@pytest.mark.parametrize("name", ["one_level", "border"])
def test_single_level_anchors_inside_mask_and_compaction_equal_the_reference(name):
    g = _load(name)
    stride, border = int(g["strides"][0]), float(g["border"])
    anchors = [tuple(map(float, a)) for a in g["anchors0"]]
    shape = tuple(int(v) for v in g["conv_shape0"])
    rpn = R.DenseRpn(8, stride=stride, width=8, num_dilations=1, anchors=anchors, allowed_border=border)
    a = rpn.anchors_for(shape, "cpu")
    ind = R.inside_indicator(a, torch.from_numpy(g["scene_shape"]).float(), border)
    assert np.array_equal(ind.numpy(), g["inside_indicator"])
    idx, inside = rpn.inside_for(shape, "cpu")
    assert torch.equal(idx, ind.nonzero().squeeze(1)) and np.array_equal(inside.numpy(), g["inside_anchors"])
    assert 0 < len(idx) < len(a)
    bbox, score, anch = rpn._finish(_raw_of_head(torch.from_numpy(g["head0"]), len(anchors)), shape)
    assert np.array_equal(bbox.numpy(), g["rpn_bbox"]) and np.array_equal(score.numpy(), g["rpn_score"])
    assert np.array_equal(anch.numpy(), g["inside_anchors"])
    boxes = R.decode_boxes(anch, bbox, tuple(float(v) for v in g["scene_shape"]))
    assert np.array_equal(boxes.numpy(), g["boxes"])
    unclipped = R.decode_boxes(anch, bbox)
    assert (unclipped != boxes).any()                       # the fixture does clip something


def test_two_anchor_levels_concatenate_and_compact_like_the_reference():
    g = _load("two_levels")
    levels = []
    for l in range(int(g["n_levels"])):
        levels.append((8, int(g["strides"][l]), 8, [tuple(map(float, a)) for a in g[f"anchors{l}"]]))
    m = R.MultiLevelRpn(levels, num_dilations=1)
    outs = []
    for l, rpn in enumerate(m.levels):
        shape = tuple(int(v) for v in g[f"conv_shape{l}"])
        assert not rpn.keep_inside
        outs.append(rpn._finish(_raw_of_head(torch.from_numpy(g[f"head{l}"]), rpn.n_anchors), shape))
    scene = tuple(int(v) for v in g["scene_shape"])
    bbox, score, anch = m.combine(outs, scene)
    assert np.array_equal(anch.numpy(), g["inside_anchors"])
    assert np.array_equal(bbox.numpy(), g["rpn_bbox"]) and np.array_equal(score.numpy(), g["rpn_score"])
    assert np.array_equal(R.decode_boxes(anch, bbox, tuple(float(v) for v in scene)).numpy(), g["boxes"])
    # the reference's anchor table in voxels (scannet_config/network.py:7-23 at 0.0375 m): 3 small + 11 large
    assert [len(a) for a in R.REF_ANCHOR_LEVELS_VOXELS] == [3, 11]
    assert np.allclose(R.REF_ANCHOR_LEVELS_VOXELS[0][0], np.array([0.3752, 0.3752, 0.4221]) / 0.0375)


def test_roi_selector_raises_on_rows_outside_the_volume():
    sel = R.RoiSelector(4, 2, 0.5)
    flag = torch.ones(1, dtype=torch.int32)

    class _PS(torch.nn.Module):     # the selection itself needs the GPU library; the flag check is host logic
        def finish(self, st):
            return ([], [], [])
    sel.proposal_selector = _PS()
    with pytest.raises(Exception, match="outside its spatial_size"):
        sel.finish((None, None, None, None, [flag]))
    assert sel.finish((None, None, None, None, [torch.zeros(1, dtype=torch.int32)])) == ([], [], [])


def test_get_roi_selector_follows_training_mode_like_the_references_conditional_stage():
    """proposal_selector.py:6-20 / custom_container.py:102-116: one selector, or a train / eval pair chosen by `.training`
    (scannet_config/run.py:847-853: 1024 / 256 / 0.5 against 1024 / 32 / 0.3)."""
    one = R.get_roi_selector(1024, 256, 0.5)
    assert isinstance(one, R.RoiSelector) and one.proposal_selector.num_keep_post_nms == 256
    pair = R.get_roi_selector(1024, 256, 0.5, val_num_keep_pre_nms=1024, val_num_keep_post_nms=32, val_thresh_nms=0.3)
    assert pair._member().proposal_selector.num_keep_post_nms == 256 and pair._member().proposal_selector.thresh_nms == 0.5
    pair.eval()
    assert pair._member().proposal_selector.num_keep_post_nms == 32 and pair._member().proposal_selector.thresh_nms == 0.3
    pair.train()
    assert pair._member() is pair.train_module
