"""RPN boundary on the device: proposal selection = top-k + greedy NMS.

Mirrors ndsis/modules/proposal_selector.py ``ProposalSelector`` (:52-89) and ndsis/utils/bbox.py
``non_maximum_supression`` (:713-759).  The reference's NMS sweeps an N x N threshold matrix with one tiny kernel per
box (N = num_keep_pre_nms launches per step); here it is one launch, one workgroup per scene (scn_nms).
"""
from __future__ import annotations

import torch

from . import _lib as L


SERIAL_NMS = False          # True: the round-1 kernel (scn_nms: one workgroup walks the boxes); the tests' cross-check


def non_maximum_suppression(proposed_boxes: torch.Tensor, overlap_threshold: float) -> torch.Tensor:
    """proposed_boxes fp32 [*, N, 2, D=3] sorted by descending confidence -> bool [*, N] (True = kept).
    N <= 4096: scn_nms_bits (suppression bit matrix over the chip + one serial walk, ~80 us for 1024 boxes against 600); above: scn_nms."""
    if proposed_boxes.shape[-2:] != (2, 3):
        raise NotImplementedError("scn_nms handles 3-D boxes [*, N, 2, 3] (the reference's ScanNet path)")
    lead, n = proposed_boxes.shape[:-3], proposed_boxes.shape[-3]
    if n == 0 or proposed_boxes.numel() == 0:
        return torch.ones(*lead, n, dtype=torch.bool, device=proposed_boxes.device)
    b = proposed_boxes.detach().to(torch.float32).reshape(-1, n, 6).contiguous()
    if not b.is_cuda:
        raise L.ScnError("boxes must live on the MI355X (no CPU fallback)")
    keep = torch.empty((b.shape[0], n), dtype=torch.uint8, device=b.device)
    lib = L.lib()
    if n <= 4096 and not SERIAL_NMS:
        scratch = L.scratch(lib.scn_nms_scratch_bytes(b.shape[0], n), b.device)
        L.check(lib.scn_nms_bits(L.ptr(b), b.shape[0], n, float(overlap_threshold), L.ptr(keep), scratch.data_ptr(), L.stream()))
    else:
        L.check(lib.scn_nms(L.ptr(b), b.shape[0], n, float(overlap_threshold), L.ptr(keep), L.stream()))
    return keep.bool().reshape(*lead, n)


class ProposalSelector(torch.nn.Module):
    """``ProposalSelector(num_keep_pre_nms, num_keep_post_nms, thresh_nms)`` (proposal_selector.py:52-89): same
    arguments and return values (lists of per-sample score / box / index tensors)."""

    def __init__(self, num_keep_pre_nms, num_keep_post_nms, thresh_nms):
        super().__init__()
        self.num_keep_pre_nms, self.num_keep_post_nms, self.thresh_nms = num_keep_pre_nms, num_keep_post_nms, thresh_nms

    def forward(self, rpn_score, rpn_bbox):
        return self.finish(self.start(rpn_score, rpn_bbox))

    def start(self, rpn_score, rpn_bbox):
        """top-k + gather + NMS queued, nothing awaited (-> state for `finish`).  The reference's forward is start + finish;
        split so that a caller can queue other work between the NMS launch and the data-dependent selection."""
        if self.num_keep_pre_nms > 0:
            rpn_score, indices = torch.topk(rpn_score, self.num_keep_pre_nms, dim=1, sorted=True)
        else:
            rpn_score, indices = torch.sort(rpn_score, dim=1, descending=True)
        batch_index = torch.arange(len(rpn_bbox), device=rpn_bbox.device).unsqueeze(1)
        rpn_bbox = rpn_bbox[batch_index, indices]
        keep = non_maximum_suppression(rpn_bbox, self.thresh_nms)
        return rpn_score, rpn_bbox, indices, keep

    def finish(self, state):
        """The data-dependent part: boolean selection of the kept proposals (one host wait), first `num_keep_post_nms`."""
        rpn_score, rpn_bbox, indices, keep = state
        post = self.num_keep_post_nms
        scores = [s[k][:post] for s, k in zip(rpn_score, keep)]
        boxes = [b[k][:post] for b, k in zip(rpn_bbox, keep)]
        index = [i[k][:post].cpu() for i, k in zip(indices, keep)]        # the reference returns CPU indices (:66)
        return scores, boxes, index
