"""Sparse ROI crop on the device.

Mirrors ndsis/modules/roi_select_sparse.py (``roi_cut`` :170-180, ``select_features`` :125-133, ``select_coords``
:136-149, ``get_inside_indicator`` :157-167, ``SparseRoiCut`` :29-52, ``SparseRoiExtraCut`` :8-26,
``RawToTensorFeatureExtractorCombiner`` :67-84) and roi_select_bbox_transform.py ``BBoxTransformerSlice`` (:56-70,87-97).

Differences in mechanism, not in results: the box test writes an int32 [BB,N] rule table that the same wave-ballot
compaction as the rulebooks turns into the (box-major, ascending point row) selection; no BB x N x C expanded view is
materialised, features are gathered once by row index; nothing is copied back to the host unless the caller asks for
the reference's return types (``roi_cut`` returns CPU coords and a CPU bool matrix, roi_select_sparse.py:180).
"""
from __future__ import annotations

import torch

from . import _lib as L
from .functional import _f32
from .ioLayers import InputLayerFunction
from .metadata import Metadata, compact_rules
from .tensor import SparseConvNetTensor


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, rows, n_src):
        X = _f32(features)
        m, c = rows.shape[0], X.shape[1]
        Y = torch.empty((m, c), dtype=torch.float32, device=X.device)
        L.check(L.lib().scn_gather_rows(L.ptr(X), L.ptr(rows), m, c, L.ptr(Y), L.stream()))
        ctx.rows, ctx.n_src = rows, n_src
        return Y

    @staticmethod
    def backward(ctx, dY):
        dY = _f32(dY)
        c = dY.shape[1]
        dX = torch.empty((ctx.n_src, c), dtype=torch.float32, device=dY.device)
        acc = torch.empty((ctx.n_src, c), dtype=torch.float64, device=dY.device)
        L.check(L.lib().scn_segment_sum(L.ptr(dY), L.ptr(ctx.rows), ctx.rows.shape[0], ctx.n_src, c, L.ptr(dX),
                                        L.ptr(acc), L.stream()))
        return dX, None, None


def transform_boxes(bbox_batch, spatial_size=None, clip=False):
    """BBoxTransformerSlice.forward: list of fp32 [n_i,2,3] -> (int32 device [BB,8] start|stop incl. sample interval,
    per-sample counts, per-box sample index)."""
    lib = L.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    counts = [len(b) for b in bbox_batch]
    bb = sum(counts)
    assoc = torch.repeat_interleave(torch.arange(len(counts)), torch.tensor(counts, dtype=torch.long)) \
        if counts else torch.zeros(0, dtype=torch.long)
    out = torch.empty((bb, 8), dtype=torch.int32, device=dev)
    if bb:
        raw = torch.cat([b.reshape(-1, 2, 3) for b in bbox_batch]).to(device=dev, dtype=torch.float32).contiguous()
        size = torch.as_tensor([int(s) for s in spatial_size], dtype=torch.int32, device=dev) if clip else None
        sample = assoc.to(device=dev, dtype=torch.int32)
        L.check(lib.scn_roi_boxes(L.ptr(raw), L.ptr(sample), bb, L.ptr(size), L.ptr(out), L.stream()))
    return out, counts, assoc


class RoiSelection:
    """Device-side result of the crop: CSR over boxes instead of the reference's dense bool matrix."""

    def __init__(self, src_row, box_of, prefix, n_points, n_boxes, inside_u8=None):
        self.src_row, self.box_of, self.prefix = src_row, box_of, prefix      # int32 [M], int32 [M], list[BB+1]
        self.n_points, self.n_boxes = n_points, n_boxes
        self._inside = inside_u8

    def is_inside(self):
        """bool CPU [BB, N] as roi_cut returns it."""
        return self._inside.bool().cpu()


def roi_select(coords_i32: torch.Tensor, boxes_i32: torch.Tensor, want_inside=True) -> RoiSelection:
    lib = L.lib()
    n, bb = coords_i32.shape[0], boxes_i32.shape[0]
    dev = coords_i32.device
    if bb == 0 or n == 0:
        e = torch.zeros(0, dtype=torch.int32, device=dev)
        inside = torch.zeros((bb, n), dtype=torch.uint8, device=dev)
        return RoiSelection(e, e, [0] * (bb + 1), n, bb, inside)
    table = torch.empty((bb, n), dtype=torch.int32, device=dev)
    inside = torch.empty((bb, n), dtype=torch.uint8, device=dev) if want_inside else None
    L.check(lib.scn_roi_table(L.ptr(coords_i32), n, L.ptr(boxes_i32), bb, L.ptr(table), L.ptr(inside), L.stream()))
    rules, seg = compact_rules(table, bb, n, want_seg=True)
    return RoiSelection(rules.in_rows, seg, rules.prefix_list(), n, bb, inside)


def _coords_to_device(coords):
    lib = L.lib()
    import ctypes as C
    dev = torch.device("cuda", torch.cuda.current_device())
    c64 = coords.to(device=dev, dtype=torch.int64).contiguous()
    c32 = torch.empty((c64.shape[0], 4), dtype=torch.int32, device=dev)
    bad, flag = C.c_int64(0), torch.empty(1, dtype=torch.int32, device=dev)
    L.check(lib.scn_coords_to_i32(L.ptr(c64), c64.shape[0], L.ptr(c32), L.ptr(flag), C.byref(bad), L.stream()))
    return c32


def roi_cut_device(coords, features, boxes_i32):
    """-> (new_coords int64 device [M,4] = (x,y,z,box), new_features [M,C], RoiSelection)."""
    lib = L.lib()
    c32 = coords if coords.dtype == torch.int32 and coords.is_cuda else _coords_to_device(coords)
    sel = roi_select(c32, boxes_i32)
    m = sel.src_row.shape[0]
    new_coords = torch.empty((m, 4), dtype=torch.int64, device=c32.device)
    L.check(lib.scn_roi_coords(L.ptr(c32), L.ptr(sel.src_row), L.ptr(sel.box_of), m, L.ptr(new_coords), L.stream()))
    new_features = _GatherRows.apply(features.to(c32.device), sel.src_row, features.shape[0])
    return new_coords, new_features, sel


def roi_cut(coords, features, bbox_tensor, bbox_sample_association):
    """Same signature and return types as the reference's ``roi_cut`` (roi_select_sparse.py:170-180):
    bbox_tensor int64 [BB,2,3] (already rounded), association int64 [BB] ->
    (extended_coordinates int64 CPU [M,4], selected_features [M,C] on the device, is_inside bool CPU [BB,N])."""
    dev = torch.device("cuda", torch.cuda.current_device())
    bt = bbox_tensor.to(device=dev, dtype=torch.int32).reshape(-1, 2, 3)
    sa = bbox_sample_association.to(device=dev, dtype=torch.int32).reshape(-1, 1)
    boxes = torch.cat([bt[:, 0], sa, bt[:, 1], sa + 1], 1).contiguous()
    new_coords, new_features, sel = roi_cut_device(coords, features, boxes)
    return new_coords.cpu(), new_features, sel.is_inside()


class SparseRoiCut(torch.nn.Module):
    """``SparseRoiCut(RawToTensorFeatureExtractorCombiner)`` (roi_select_sparse.py:29-52,67-84): feature_map is the
    raw tuple (coords, features, spatial_size, ..., batch_splits); returns (SparseConvNetTensor over the ROI batch,
    (is_inside, bbox_sample_count, batch_splits))."""

    def __init__(self, clip_boxes=False, spatial_size_offset=0, mode=4, dense_inside=True):
        super().__init__()
        self.clip, self.offset, self.mode, self.dense_inside = clip_boxes, spatial_size_offset, mode, dense_inside

    def forward(self, feature_map, bbox_batch):
        coords, features, spatial_size, *_, batch_splits = feature_map
        boxes, counts, _ = transform_boxes(bbox_batch, spatial_size, self.clip)
        new_coords, new_features, sel = roi_cut_device(coords, features, boxes)
        size = torch.as_tensor([int(s) + self.offset for s in spatial_size], dtype=torch.long)
        if new_coords.shape[0] == 0:        # no point in any box: CustomInputLayer returns None (custom_operations.py:71,85-86)
            out = None
        else:
            md = Metadata(3)
            feats = InputLayerFunction.apply(3, md, size, new_coords, new_features, boxes.shape[0], self.mode)
            out = SparseConvNetTensor(features=feats, metadata=md, spatial_size=size)
        return out, (sel.is_inside() if self.dense_inside else sel, counts, batch_splits)


# ------------------------------------------------------------------------------------------------------
# Mask-head epilogue on the device: what the reference does with the dense BB x N `is_inside` matrix
# after the mask network -- SparseMaskPredictor (model.py:824-882) and SparseMaskLossSelector (model.py:1150-1227) --
# from the CSR selection, without bringing the indicator to the host.
# ------------------------------------------------------------------------------------------------------
def _box_layout(sel: RoiSelection, box_sample_count, batch_splits):
    """Per-box (sample, first box of the sample, first point row of the sample, points of the sample) on the host."""
    import numpy as np
    counts = np.asarray([int(c) for c in box_sample_count], dtype=np.int64)
    splits = np.asarray([int(c) for c in batch_splits], dtype=np.int64)
    if counts.sum() != sel.n_boxes or splits.sum() != sel.n_points or len(counts) != len(splits):
        raise L.ScnError("box_sample_count / batch_splits do not match the selection")
    sample = np.repeat(np.arange(len(counts)), counts)
    box_start = np.concatenate([[0], np.cumsum(counts)])[:-1]
    point_start = np.concatenate([[0], np.cumsum(splits)])[:-1]
    return counts, splits, sample, box_start, point_start


def mask_predict(mask_output, sel: RoiSelection, box_sample_count, batch_splits, class_indices, num_valid=0):
    """SparseMaskPredictor.forward: list (one per sample) of fp32 device tensors [boxes_s, points_s] holding
    sigmoid(score of the box's class) at the points inside each box and 0 elsewhere / for invalid classes."""
    import numpy as np
    lib = L.lib()
    S = _f32(mask_output.detach())
    dev = S.device
    counts, splits, sample, box_start, point_start = _box_layout(sel, box_sample_count, batch_splits)
    sizes = counts * splits
    out_off = np.concatenate([[0], np.cumsum(sizes)])
    out = torch.zeros(int(out_off[-1]), dtype=torch.float32, device=dev)
    box = np.arange(sel.n_boxes)
    row_base = out_off[sample] + (box - box_start[sample]) * splits[sample] - point_start[sample]
    m = sel.src_row.shape[0]
    if m:
        if S.shape[0] != m:
            raise L.ScnError(f"mask_output has {S.shape[0]} rows, the selection {m}")
        cls = torch.as_tensor(class_indices, dtype=torch.int64).to(dev).contiguous()
        rb = torch.from_numpy(row_base.astype(np.int64)).to(dev)
        L.check(lib.scn_mask_scatter(L.ptr(S), m, S.shape[1], L.ptr(sel.src_row), L.ptr(sel.box_of), L.ptr(cls),
                                     int(num_valid), L.ptr(rb), L.ptr(out), L.stream()))
    return [out[out_off[s]:out_off[s + 1]].view(int(counts[s]), int(splits[s])) for s in range(len(counts))]


class _MaskGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, src_point, box_of, label_of_box, gt_base, gt_flat):
        S = _f32(scores)
        m, k = S.shape
        pred = torch.empty(m, dtype=torch.float32, device=S.device)
        gt = torch.empty(m, dtype=torch.float32, device=S.device)
        keep = torch.empty(m, dtype=torch.uint8, device=S.device)
        L.check(L.lib().scn_mask_gather(L.ptr(S), m, k, L.ptr(src_point), L.ptr(box_of), L.ptr(label_of_box),
                                        L.ptr(gt_base), L.ptr(gt_flat), L.ptr(pred), L.ptr(gt), L.ptr(keep),
                                        L.stream()))
        ctx.k, ctx.idx = k, (box_of, label_of_box, gt_base)
        ctx.mark_non_differentiable(gt, keep)
        return pred, gt, keep

    @staticmethod
    def backward(ctx, dpred, _dgt, _dkeep):
        box_of, label_of_box, gt_base = ctx.idx
        dpred = _f32(dpred)
        m = dpred.shape[0]
        dS = torch.empty((m, ctx.k), dtype=torch.float32, device=dpred.device)
        L.check(L.lib().scn_mask_gather_bwd(L.ptr(dpred), m, ctx.k, L.ptr(box_of), L.ptr(label_of_box), L.ptr(gt_base),
                                            L.ptr(dS), L.stream()))
        return dS, None, None, None, None, None


def mask_loss_select(mask_scores, sel: RoiSelection, box_sample_count, batch_splits, keep_list, gt_associations_list,
                     gt_labels_list, gt_masks_list):
    """SparseMaskLossSelector.forward on the device.

    keep_list[s]: bool [boxes_s] (LossFilter keep); gt_associations_list[s]: int64 [kept boxes of s] (ground-truth index
    of every kept box); gt_labels_list[s]: int64 [G_s]; gt_masks_list[s]: [G_s, points_s] (bool or float).
    Returns (pred, gt, rows_per_kept_box, labels): pred / gt are flat fp32 device tensors over the rows of the KEPT
    boxes in crop order (pred differentiable w.r.t. mask_scores), i.e. the concatenation of the reference's nested
    lists `pred_masks` / `gt_masks`; labels = torch.cat(selected_labels)."""
    import numpy as np
    dev = mask_scores.device
    counts, splits, sample, box_start, point_start = _box_layout(sel, box_sample_count, batch_splits)
    n_s = len(counts)
    label = np.full(sel.n_boxes, -1, dtype=np.int64)
    gt_base = np.full(sel.n_boxes, -1, dtype=np.int64)
    gt_sizes = np.asarray([int(g.shape[0]) * int(splits[s]) for s, g in enumerate(gt_masks_list)], dtype=np.int64)
    gt_off = np.concatenate([[0], np.cumsum(gt_sizes)])
    labels_out = []
    for s in range(n_s):
        keep = np.asarray(torch.as_tensor(keep_list[s]).cpu().numpy(), dtype=bool)
        assoc = torch.as_tensor(gt_associations_list[s]).cpu().numpy().astype(np.int64)
        boxes = box_start[s] + np.nonzero(keep)[0]
        if len(boxes) != len(assoc):
            raise L.ScnError("gt_associations_list must have one entry per kept box")
        lab = torch.as_tensor(gt_labels_list[s]).cpu().numpy().astype(np.int64)[assoc]
        label[boxes] = lab
        gt_base[boxes] = gt_off[s] + assoc * splits[s] - point_start[s]
        labels_out.append(torch.from_numpy(lab))
    gt_flat = torch.cat([torch.as_tensor(g).reshape(-1).to(torch.float32) for g in gt_masks_list]).to(dev) \
        if gt_off[-1] else torch.zeros(1, dtype=torch.float32, device=dev)
    lab_d = torch.from_numpy(label).to(dev)
    base_d = torch.from_numpy(gt_base).to(dev)
    m = sel.src_row.shape[0]
    if m == 0:
        e = torch.zeros(0, dtype=torch.float32, device=dev)
        return e, e, [0] * int((label >= 0).sum()), torch.cat(labels_out) if labels_out else torch.zeros(0, dtype=torch.long)
    pred, gt, keep_row = _MaskGather.apply(mask_scores, sel.src_row, sel.box_of, lab_d, base_d, gt_flat)
    rows = np.diff(np.asarray(sel.prefix, dtype=np.int64))
    kept = label >= 0
    if kept.all():
        return pred, gt, rows.tolist(), torch.cat(labels_out)
    kr = keep_row.bool()
    return pred[kr], gt[kr], rows[kept].tolist(), torch.cat(labels_out)
