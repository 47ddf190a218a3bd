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
    N <= 4096: scn_nms_bits (suppression bit matrix over the chip + one serial walk, 12 + 42 us for 1024 boxes against 600); above: scn_nms."""
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


TORCH_TOPK = False          # True: torch.topk + advanced indexing (the reference's calls); the tests' cross-check and the A/B
DECODE_FIRST = False        # True: RoiSelector decodes every anchor's box before the selection (the reference's order of calls)
_TOPK_MAX_K = 2048          # scn_topk_boxes: k <= 2048
_topk_scratch = {}          # (device, batch, STREAM) -> the zeroed state scn_topk_boxes keeps between calls: the kernels'
                            # histogram / cursor atomics of two calls must never interleave, and calls on one stream do not


class _TopkBoxes(torch.autograd.Function):
    """torch.topk(score, k, dim=1, sorted=True) and boxes[batch, indices] as one call (scn_topk_boxes: radix select, 4 launches
    against torch.topk's 19 on a [1, 524 288] field).  Equal scores come out by ascending index (torch leaves that open)."""

    @staticmethod
    def forward(ctx, score, boxes, k):
        b, n = score.shape
        s = score.detach().contiguous()
        bx = boxes.detach().reshape(b, n, 6).contiguous()
        vals = torch.empty((b, k), dtype=torch.float32, device=s.device)
        idx = torch.empty((b, k), dtype=torch.int64, device=s.device)
        out = torch.empty((b, k, 2, 3), dtype=torch.float32, device=s.device)
        lib = L.lib()
        stream = L.stream()
        key = (s.device, b, stream)
        scratch = _topk_scratch.get(key)
        if scratch is None:
            scratch = _topk_scratch[key] = torch.zeros(lib.scn_topk_scratch_bytes(b), dtype=torch.uint8, device=s.device)
        try:
            L.check(lib.scn_topk_boxes(L.ptr(s), L.ptr(bx), b, n, k, L.ptr(vals), L.ptr(idx), L.ptr(out), L.ptr(scratch),
                                       stream))
        except Exception:
            scratch.zero_()                     # (the state is only zero again after a COMPLETE call)
            raise
        ctx.save_for_backward(idx)
        ctx.shape = (b, n, tuple(boxes.shape))
        ctx.mark_non_differentiable(idx)
        return vals, idx, out

    @staticmethod
    def backward(ctx, gv, _gi, gb):
        (idx,) = ctx.saved_tensors
        b, n, bshape = ctx.shape
        ds = db = None
        if ctx.needs_input_grad[0] and gv is not None:
            ds = torch.zeros((b, n), dtype=gv.dtype, device=gv.device).scatter_(1, idx, gv)
        if ctx.needs_input_grad[1] and gb is not None:
            db = torch.zeros((b, n, 6), dtype=gb.dtype, device=gb.device)
            db.scatter_(1, idx.unsqueeze(-1).expand(b, idx.shape[1], 6), gb.reshape(b, -1, 6))
            db = db.reshape(bshape)
        return ds, db, None


def topk_boxes(score: torch.Tensor, boxes: torch.Tensor, k: int):
    """-> (score [B, k] descending, indices int64 [B, k], boxes[batch, indices] [B, k, 2, 3]); differentiable in score and boxes."""
    b, n = score.shape
    if (TORCH_TOPK or not score.is_cuda or score.dtype != torch.float32 or boxes.dtype != torch.float32 or k > _TOPK_MAX_K
            or k > n or tuple(boxes.shape[-2:]) != (2, 3) or b == 0):
        vals, idx = torch.topk(score, k, dim=1, sorted=True)
        return vals, idx, boxes[torch.arange(len(boxes), device=boxes.device).unsqueeze(1), idx]
    return _TopkBoxes.apply(score, boxes, int(k))


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
            rpn_score, indices, rpn_bbox = topk_boxes(rpn_score, rpn_bbox, self.num_keep_pre_nms)
        else:
            rpn_score, indices = torch.sort(rpn_score, dim=1, descending=True)
            rpn_bbox = rpn_bbox[torch.arange(len(rpn_bbox), device=rpn_bbox.device).unsqueeze(1), indices]
        keep = non_maximum_suppression(rpn_bbox, self.thresh_nms)
        return rpn_score, rpn_bbox, indices, keep

    def start_from_deltas(self, rpn_score, deltas, anchors, decode):
        """`start(rpn_score, decode(anchors, deltas))` with the decode AFTER the selection: the box arithmetic is per anchor, so
        decoding the k selected anchors gives the bits decoding all of them and gathering would (8 elementwise launches over
        524 k anchors become 8 over 1024).  anchors [N, 2, 3], deltas [B, N, 2, 3]."""
        if self.num_keep_pre_nms <= 0 or DECODE_FIRST:
            return self.start(rpn_score, decode(anchors, deltas))
        rpn_score, indices, deltas_k = topk_boxes(rpn_score, deltas, self.num_keep_pre_nms)
        rpn_bbox = decode(anchors.reshape(-1, 2, 3)[indices], deltas_k)
        keep = non_maximum_suppression(rpn_bbox, self.thresh_nms)
        return rpn_score, rpn_bbox, indices, keep

    def finish(self, state):
        """The data-dependent part: boolean selection of the kept proposals (one host wait), first `num_keep_post_nms`."""
        rpn_score, rpn_bbox, indices, keep = state
        post = self.num_keep_post_nms
        # (x[k][:post] three times is three `nonzero` passes and three host waits: ONE list of kept positions, three gathers)
        sel = [k.nonzero().squeeze(1)[:post] for k in keep]
        scores = [s[j] for s, j in zip(rpn_score, sel)]
        boxes = [b[j] for b, j in zip(rpn_bbox, sel)]
        index = [i[j].cpu() for i, j in zip(indices, sel)]                # the reference returns CPU indices (:66)
        return scores, boxes, index
