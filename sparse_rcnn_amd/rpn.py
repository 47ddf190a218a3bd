"""RPN boundary of the detection + mask step (BASELINE configs[2] "Backbone + RPN + sparse ROI-crop mask
head"): the proposals of a step come out of the SAME forward that feeds the mask head.

    encoder level (sparse, stride 8)  --scn.SparseToDense-->  dense [B, C, X/8, Y/8, Z/8]
        --dilation stack (Conv3d 3^3 + ReLU, torch / MIOpen: SURVEY §2 row 8 leaves the dense RPN to PyTorch)-->
        --1x1x1 head--> per anchor 6 box deltas + 1 score          (anchor_network.py:73-124 `AnchorNetworkConv`)
        --RoiSelector: detach, decode against the anchors, sigmoid, top-k, greedy NMS (ONE launch: scn_nms), keep `post`-->
        list of fp32 boxes [n_i, 2, 3] per sample                    (proposal_selector.py:23-89; bbox.py:139-165,367-398)

What this package contributes to it: SparseToDense (A13, scn_elem.hip), the one-launch NMS (proposals.py) and the consumer
of the boxes (roi.SparseRoiCut).  The dense layers are plain torch modules with the reference's structure
(module_factory.py:581-611 `get_dilation_network`: `num_dilations` x [same convolution + ReLU] behind a SparseToDense).
"""
from __future__ import annotations

import torch
from torch import nn

from . import modules as M
from .proposals import ProposalSelector

# anchor edge lengths in voxels at the one anchor level (stride 8); the synthetic boxes of cfg 3 have edges 8-96
DEFAULT_ANCHORS = ((12.0, 12.0, 12.0), (24.0, 24.0, 24.0), (48.0, 48.0, 32.0), (96.0, 96.0, 48.0))


class DenseRpn(nn.Module):
    """`DenseRpn(channels, stride)`: SparseToDense -> dilation stack -> AnchorNetworkConv-style head.
    forward(level_tensor) -> (rpn_bbox [B, N, 2, 3] = (position delta, log-size delta), rpn_score [B, N] raw,
    anchors [N, 2, 3] = (centre, size)), N = X' Y' Z' x A, spatial-major / anchor-minor."""

    def __init__(self, channels, stride=8, width=32, num_dilations=2, anchors=DEFAULT_ANCHORS, autocast_bf16=False):
        super().__init__()
        self.channels, self.stride, self.width = int(channels), int(stride), int(width)
        self.to_dense = M.SparseToDense(3, self.channels)
        layers, cin = [], self.channels
        for d in range(num_dilations):                       # get_dilation_network: same convolution + ReLU, dilation 1, 2, ...
            layers += [nn.Conv3d(cin, self.width, 3, padding=d + 1, dilation=d + 1), nn.ReLU(inplace=True)]
            cin = self.width
        self.stack = nn.Sequential(*layers)
        self.register_buffer("anchor_sizes", torch.tensor(anchors, dtype=torch.float32))
        self.n_anchors = len(anchors)
        self.head = nn.Conv3d(cin, self.n_anchors * 7, 1)    # anchor_network.py:88-92: num_anchors x (2 x num_dims + 1)
        self.autocast_bf16 = bool(autocast_bf16)
        self._anchor_cache = {}

    def anchors_for(self, shape, device):
        key = (tuple(shape), str(device))
        a = self._anchor_cache.get(key)
        if a is None:
            g = torch.stack(torch.meshgrid(*[(torch.arange(int(s), dtype=torch.float32) + 0.5) * self.stride for s in shape],
                                           indexing="ij"), -1).reshape(-1, 1, 3)                   # cell centres, voxels
            sz = self.anchor_sizes.detach().cpu().reshape(1, -1, 3)
            a = torch.stack([g.expand(-1, sz.shape[1], -1), sz.expand(g.shape[0], -1, -1)], 2).reshape(-1, 2, 3)
            a = self._anchor_cache[key] = a.to(device)
        return a

    def forward(self, level_tensor):
        dense = self.to_dense(level_tensor)                  # [B, C, X', Y', Z'] (bf16 when the slab is bf16-stored)
        if self.autocast_bf16:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                raw = self.head(self.stack(dense))
            raw = raw.float()
        else:
            raw = self.head(self.stack(dense.float()))
        B = raw.shape[0]
        shape = raw.shape[2:]
        raw = raw.view(B, self.n_anchors, 7, -1).permute(0, 3, 1, 2).reshape(B, -1, 7)
        rpn_bbox = raw[..., :6].reshape(B, -1, 2, 3)
        rpn_score = raw[..., 6]
        return rpn_bbox, rpn_score, self.anchors_for(shape, raw.device)


def decode_boxes(anchors, deltas):
    """bbox_transform_inv (ndsis/utils/bbox.py:139-165,267-287,367-398): anchors (centre, size), deltas (position, log size)
    -> boxes (start, stop).  fp32, same operation order as the reference."""
    pos = deltas[..., 0, :] * anchors[..., 1, :] + anchors[..., 0, :]
    size = torch.exp(deltas[..., 1, :]) * anchors[..., 1, :]
    half = size / 2
    return torch.stack((pos - half, pos + half), dim=-2)


class RoiSelector(nn.Module):
    """`RoiSelector(num_keep_pre_nms, num_keep_post_nms, thresh_nms, detach=True)` (proposal_selector.py:23-50):
    forward(rpn_bbox, rpn_score, anchors) -> (roi_score, roi_bbox, roi_index) lists per sample.  Where the reference calls
    `anchor_description(rpn_bbox)`, this takes the anchors tensor and decodes with `decode_boxes`."""

    def __init__(self, num_keep_pre_nms=1024, num_keep_post_nms=64, thresh_nms=0.5, detach=True):
        super().__init__()
        self.proposal_selector = ProposalSelector(num_keep_pre_nms, num_keep_post_nms, thresh_nms)
        self.detach = detach

    def forward(self, rpn_bbox, rpn_score, anchors):
        if self.detach:
            rpn_bbox, rpn_score = rpn_bbox.detach(), rpn_score.detach()
        return self.proposal_selector(torch.sigmoid(rpn_score), decode_boxes(anchors, rpn_bbox))
