"""Sparse ROI crop on the device.

Mirrors ndsis/modules/roi_select_sparse.py -- ``roi_cut`` :170-180, ``select_features`` :125-133, ``select_coords``
:136-149, ``get_inside_indicator`` :157-167, ``SparseRoiCut`` :29-52, ``SparseRoiExtraCut`` :8-26 and the four
extractor/combiner classes :55-122 -- and roi_select_bbox_transform.py ``BBoxTransformerSlice`` (:56-70,87-97), with the
same class names, constructor signatures and return structure.

Differences in mechanism, not in results: the crop is count -> scan -> scatter over 256-point units (scn_roi_count /
scn_roi_fill, csrc/scn_roi.hip).  No [boxes, points] object and no BB x N x C expanded view exist; the selection is a CSR
list (`RoiSelection`: point row and box of every selected row, box-major, ascending point row -- the order the
reference's boolean-mask gather produces).  The dense bool matrix ``is_inside`` that the reference hands on to its mask
predictor / loss selector is rebuilt from the list only when asked for (``RoiSelection.is_inside()``,
``dense_inside=True``); the device consumers below (``mask_predict``, ``mask_loss_select``, ``SparseRoiExtraCut``) take
the list.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .functional import _f32
from .ioLayers import InputLayerFunction
from .metadata import Metadata, _Readback
from .tensor import SparseConvNetTensor


class _GatherRows(torch.autograd.Function):
    """select_features (roi_select_sparse.py:125-133): rows of `features` by the selection's point rows; backward is the
    segment sum over the boxes a point fell into."""

    @staticmethod
    def forward(ctx, features, rows, n_src):
        hb = features.dtype == torch.bfloat16          # bf16 storage: the same row copy on half the bytes
        X = features.contiguous() if hb else _f32(features)
        m, c = rows.shape[0], X.shape[1]
        Y = torch.empty((m, c), dtype=X.dtype, device=X.device)
        gather = L.lib().scn_gather_rows_bf16 if hb else L.lib().scn_gather_rows
        L.check(gather(L.ptr(X), L.ptr(rows), m, c, L.ptr(Y), L.stream()))
        ctx.rows, ctx.n_src, ctx.hb = rows, n_src, hb
        return Y

    @staticmethod
    def backward(ctx, dY):
        dY = dY.to(torch.bfloat16).contiguous() if ctx.hb else _f32(dY)
        c = dY.shape[1]
        dX = torch.empty((ctx.n_src, c), dtype=dY.dtype, device=dY.device)
        acc = torch.empty((ctx.n_src, c), dtype=torch.float64, device=dY.device)
        seg = L.lib().scn_segment_sum_bf16 if ctx.hb else L.lib().scn_segment_sum
        L.check(seg(L.ptr(dY), L.ptr(ctx.rows), ctx.rows.shape[0], ctx.n_src, c, L.ptr(dX), L.ptr(acc), L.stream()))
        return dX, None, None


def transform_boxes(bbox_batch, spatial_size=None, clip=False, resize=None):
    """BBoxTransformerSlice.forward (roi_select_bbox_transform.py:56-70,87-97): list of fp32 [n_i,2,3] -> (int32 device
    [BB,8] start|stop incl. the sample interval, per-sample counts, per-box sample index).  resize: the Divider's value
    (scalar or one per axis, :15-21), applied in fp32 before floor / ceil."""
    lib = L.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    counts = [len(b) for b in bbox_batch]
    bb = sum(counts)
    # (a Python list, not torch.repeat_interleave: that CPU operator enters torch's intra-op thread pool, which on a box whose
    #  cpu_count() exceeds the job's share stalls the step by tens of milliseconds every few calls -- profiles/r6_ref_crop_rpn.txt)
    assoc = torch.tensor([s for s, c in enumerate(counts) for _ in range(c)], dtype=torch.long)
    out = torch.empty((bb, 8), dtype=torch.int32, device=dev)
    if bb:
        raw = torch.cat([b.reshape(-1, 2, 3) for b in bbox_batch]).to(device=dev, dtype=torch.float32).contiguous()
        size = torch.as_tensor([int(s) for s in spatial_size], dtype=torch.int32, device=dev) if clip else None
        div = None
        if resize is not None:
            div = torch.as_tensor(resize, dtype=torch.float32).reshape(-1)
            div = (div.repeat(3) if div.numel() == 1 else div).to(dev).contiguous()
            if div.numel() != 3:
                raise ValueError("resize_boxes must be a scalar or have one entry per axis")
        sample = assoc.to(device=dev, dtype=torch.int32)
        L.check(lib.scn_roi_boxes(L.ptr(raw), L.ptr(sample), bb, L.ptr(size), L.ptr(div), L.ptr(out), L.stream()))
    return out, counts, assoc


class RoiSelection:
    """Device-side result of the crop: the list form (CSR over boxes) of the reference's dense bool matrix.

      src_row int32 [M]   point row of every selected row        box_of int32 [M]   its box
      prefix  list[BB+1]  first selected row of every box (host) new_coords int64 [M,4] = (x, y, z, box) on the device
    """

    def __init__(self, src_row, box_of, prefix, n_points, n_boxes, new_coords=None):
        self.src_row, self.box_of, self.prefix = src_row, box_of, prefix
        self.n_points, self.n_boxes = n_points, n_boxes
        self.new_coords = new_coords
        self._inside = None

    def is_inside_u8(self):
        """uint8 device [BB, N] (scn_roi_inside), built on first request."""
        if self._inside is None:
            dev = self.src_row.device
            inside = torch.empty((self.n_boxes, self.n_points), dtype=torch.uint8, device=dev)
            L.check(L.lib().scn_roi_inside(L.ptr(self.src_row), L.ptr(self.box_of), self.src_row.shape[0], self.n_points,
                                           self.n_boxes, L.ptr(inside), L.stream()))
            self._inside = inside
        return self._inside

    def is_inside(self):
        """bool CPU [BB, N] as roi_cut returns it (roi_select_sparse.py:180)."""
        return self.is_inside_u8().bool().cpu()

    # the reference's consumers use `len(is_inside)` for the number of boxes (roi_select_sparse.py:23,50)
    def __len__(self):
        return self.n_boxes


class _SelectStarted:
    """roi_select after its count pass has been queued: `finish()` waits for the per-box counts and queues the fill."""

    def __init__(self, coords_i32, boxes_i32, want_coords=True):
        lib = L.lib()
        self.coords, self.boxes, self.want_coords = coords_i32, boxes_i32, want_coords
        n, bb = coords_i32.shape[0], boxes_i32.shape[0]
        self.readback = None
        if bb == 0 or n == 0:
            return
        dev = coords_i32.device
        self.offsets = torch.empty(bb * lib.scn_roi_units(n) + bb, dtype=torch.int32, device=dev)
        self.prefix_dev = torch.empty(bb + 1, dtype=torch.int64, device=dev)
        L.check(lib.scn_roi_count(L.ptr(coords_i32), n, L.ptr(boxes_i32), bb, L.ptr(self.offsets), L.ptr(self.prefix_dev), None,
                                  L.stream()))
        self.readback = _Readback(self.prefix_dev)

    def finish(self) -> "RoiSelection":
        lib = L.lib()
        coords_i32, boxes_i32 = self.coords, self.boxes
        n, bb = coords_i32.shape[0], boxes_i32.shape[0]
        dev = coords_i32.device
        if self.readback is None:
            e = torch.zeros(0, dtype=torch.int32, device=dev)
            return RoiSelection(e, e, [0] * (bb + 1), n, bb, torch.zeros((0, 4), dtype=torch.int64, device=dev))
        prefix = self.readback.get()[0].tolist()
        m = int(prefix[bb])
        if m >= 2 ** 31 - 1:
            raise L.ScnError("ROI selection exceeds int32 rows")
        src_row = torch.empty(m, dtype=torch.int32, device=dev)
        box_of = torch.empty(m, dtype=torch.int32, device=dev)
        new_coords = torch.empty((m, 4), dtype=torch.int64, device=dev) if self.want_coords else None
        if m:
            L.check(lib.scn_roi_fill(L.ptr(coords_i32), n, L.ptr(boxes_i32), bb, L.ptr(self.offsets), L.ptr(self.prefix_dev),
                                     L.ptr(src_row), L.ptr(box_of), L.ptr(new_coords), L.stream()))
        return RoiSelection(src_row, box_of, prefix, n, bb, new_coords)


def roi_select(coords_i32: torch.Tensor, boxes_i32: torch.Tensor, want_coords=True) -> RoiSelection:
    """count -> scan -> scatter.  One host wait (for the per-box row counts, M = prefix[BB])."""
    return _SelectStarted(coords_i32, boxes_i32, want_coords).finish()


def _coords_to_device(coords):
    lib = L.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    if coords.dtype == torch.int32 and coords.is_cuda:
        return coords.contiguous()
    c64 = coords.to(device=dev, dtype=torch.int64).contiguous()
    c32 = torch.empty((c64.shape[0], 4), dtype=torch.int32, device=dev)
    bad, flag = C.c_int64(0), torch.empty(1, dtype=torch.int32, device=dev)
    L.check(lib.scn_coords_to_i32(L.ptr(c64), c64.shape[0], L.ptr(c32), L.ptr(flag), C.byref(bad), L.stream()))
    return c32


def select_features(features, selection: RoiSelection):
    """select_features (roi_select_sparse.py:125-133) from the list: features[src_row], differentiable."""
    dev = selection.src_row.device
    return _GatherRows.apply(features.to(dev), selection.src_row, features.shape[0])


def select_coords(coords, selection: RoiSelection):
    """select_coords (roi_select_sparse.py:136-149): int64 device [M,4] = (x, y, z, box)."""
    if selection.new_coords is not None:
        return selection.new_coords
    c32 = _coords_to_device(coords)
    m = selection.src_row.shape[0]
    out = torch.empty((m, 4), dtype=torch.int64, device=c32.device)
    L.check(L.lib().scn_roi_coords(L.ptr(c32), L.ptr(selection.src_row), L.ptr(selection.box_of), m, L.ptr(out),
                                   L.stream()))
    return out


def roi_cut_device(coords, features, boxes_i32):
    """-> (new_coords int64 device [M,4] = (x,y,z,box), new_features [M,C], RoiSelection)."""
    sel = roi_select(_coords_to_device(coords), boxes_i32)
    return sel.new_coords, select_features(features, sel), sel


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


# ---- the reference's extractor / combiner classes (roi_select_sparse.py:55-122) -----------------------------------
class RawToFeaturesSceneFeatureExtractorCombiner:
    """:55-64 -- raw scene tuple in, the selected feature rows out."""
    NEED_COORDS = False

    @staticmethod
    def extract(feature_map):
        new_coords, new_features, spatial_size, *_, batch_splits = feature_map
        return new_coords, new_features, spatial_size, batch_splits

    @staticmethod
    def combine(new_coords, new_features, spatial_size, batch_size=0):
        return new_features


class RawToTensorFeatureExtractorCombiner:
    """:67-84 -- raw scene tuple in, a SparseConvNetTensor over the ROI batch out (InputLayer mode 4, batch_size = BB so
    that empty boxes stay addressable)."""
    NEED_COORDS = True
    MODE = 4
    extract = staticmethod(RawToFeaturesSceneFeatureExtractorCombiner.extract)

    @classmethod
    def combine(cls, new_coords, new_features, spatial_size, batch_size=0, metadata=None):
        """metadata: a Metadata already prepared for new_coords (prepare_cut_in_thread); None: built here."""
        if new_coords.shape[0] == 0:     # CustomInputLayer's contract for an empty crop (custom_operations.py:71,85-86)
            return None
        md = metadata if metadata is not None else Metadata(len(spatial_size))
        size = torch.as_tensor([int(s) for s in spatial_size], dtype=torch.long)
        feats = InputLayerFunction.apply(len(spatial_size), md, size, new_coords, new_features, batch_size, cls.MODE)
        return SparseConvNetTensor(features=feats, metadata=md, spatial_size=size)


class RawToRawFeatureExtractorCombiner:
    """:87-96 -- raw scene tuple in, raw (coords, features, spatial_size, batch_size) out."""
    NEED_COORDS = True
    extract = staticmethod(RawToFeaturesSceneFeatureExtractorCombiner.extract)

    @staticmethod
    def combine(new_coords, new_features, spatial_size, batch_size=0):
        return new_coords, new_features, spatial_size, batch_size


class TensorToTensorFeatureExtractorCombiner(RawToTensorFeatureExtractorCombiner):
    """:99-122 -- a SparseConvNetTensor in (its active sites are the points: unique per sample, so the re-voxelisation
    is InputLayer mode 0), a SparseConvNetTensor over the ROI batch out.  The sites stay on the device (the int32 grid
    of the tensor's Metadata); batch_splits = rows per sample."""
    MODE = 0

    @staticmethod
    def extract(feature_map):
        size = tuple(int(s) for s in feature_map.spatial_size)
        grid = feature_map.metadata.grid(size)
        batch = grid.coords[:, 3]
        if grid.n and bool((batch[1:] < batch[:-1]).any()):      # the reference asserts is_sorted (:106-107)
            raise L.ScnError("TensorToTensor ROI cut needs rows grouped by ascending sample")
        batch_splits = torch.bincount(batch.long(), minlength=feature_map.batch_size()).cpu()
        return grid.coords, feature_map.features, feature_map.spatial_size, batch_splits


class SparseRoiCut(torch.nn.Module):
    """``SparseRoiCut(feature_extractor_combiner, clip_boxes=False, resize_boxes=None)`` (roi_select_sparse.py:29-52).

    forward(feature_map, bbox_batch) -> (combiner output, (is_inside, bbox_sample_count, batch_splits)).  ``is_inside``
    is the reference's bool CPU [BB, N] matrix when ``dense_inside`` (keyword-only, default True: the reference's return
    type) and the `RoiSelection` list otherwise -- what this package's own consumers (SparseRoiExtraCut, mask_predict,
    mask_loss_select) take; the matrix is never built on that path."""

    def __init__(self, feature_extractor_combiner, clip_boxes=False, resize_boxes=None, *, dense_inside=True):
        super().__init__()
        if not (hasattr(feature_extractor_combiner, "extract") and hasattr(feature_extractor_combiner, "combine")):
            raise TypeError("SparseRoiCut: the first argument is the feature extractor/combiner "
                            "(roi_select_sparse.py:30-36), got " + repr(feature_extractor_combiner))
        if not isinstance(clip_boxes, bool):
            raise TypeError("clip_boxes must be a bool")
        self.feature_extractor_combiner = feature_extractor_combiner
        self.clip_boxes, self.resize_boxes, self.dense_inside = clip_boxes, resize_boxes, dense_inside

    def forward(self, feature_map, bbox_batch, prepared=None):
        """prepared: the result of `prepare_cut_in_thread` for the same coordinates, boxes and spatial size -- the
        selection and the ROI batch's index structures depend on those only, so a caller that knows the boxes before the
        features are ready (the mask branch: boxes come from the RPN, features from layers that still run) has them built
        meanwhile."""
        fec = self.feature_extractor_combiner
        old_coords, old_features, spatial_size, batch_splits = fec.extract(feature_map)
        if prepared is not None:
            boxes, counts, sel, md = prepared.result()
            if sel.n_points != old_coords.shape[0]:
                raise L.ScnError("SparseRoiCut: `prepared` was built for other coordinates")
            new_coords, new_features = sel.new_coords, select_features(old_features, sel)
            if md is not None:
                md.handover()
                box_features = fec.combine(new_coords, new_features, spatial_size, sel.n_boxes, metadata=md)
            else:
                box_features = fec.combine(new_coords, new_features, spatial_size, sel.n_boxes)
        else:
            boxes, counts, _ = transform_boxes(bbox_batch, spatial_size, self.clip_boxes, self.resize_boxes)
            new_coords, new_features, sel = roi_cut_device(old_coords, old_features, boxes)
            box_features = fec.combine(new_coords, new_features, spatial_size, sel.n_boxes)
        return box_features, (sel.is_inside() if self.dense_inside else sel, counts, batch_splits)

    def prepare_cut_in_thread(self, coords, spatial_size, bbox_batch, n_levels=0, in_thread=True, split=False, xcd_order=None):
        """Start building what the cut needs from coordinates and boxes alone -- the selection list and, for the
        RawToTensor combiner, the InputLayer rules + rulebook pyramid of the ROI batch (n_levels; 0: the depth the last
        network over this spatial size used, Metadata.LEVELS_HINT) -- on its own high-priority stream, so that the host
        waits of the build (row counts) wait for index kernels only and not for whatever the caller has queued before.
        in_thread: on a helper thread as well (False: on the caller's thread, which then blocks for the build's own
        ~1 ms while the GPU works through the caller's queue).
        split (caller's thread): only the part WITHOUT host waits runs now -- boxes to the device, the selection's count pass
        queued behind what the caller's stream holds at this moment; the two waits (selected rows, level sizes) and the
        launches between them happen inside `result()`.  A caller that queues independent work in between (the mask
        branch: its scene-level input stage) keeps the GPU busy through both waits.
        Returns a handle for `forward(..., prepared=)`."""
        from concurrent.futures import ThreadPoolExecutor
        global _roi_pool, _roi_stream
        dev = torch.device("cuda", torch.cuda.current_device())
        if _roi_pool is None and in_thread and not split:
            _roi_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="scn-roi-index")
        if _roi_stream.get(dev) is None:
            _roi_stream[dev] = torch.cuda.Stream(device=dev, priority=-1)
        side, main = _roi_stream[dev], torch.cuda.current_stream()
        size = tuple(int(s) for s in spatial_size)
        want_md = isinstance(self.feature_extractor_combiner, RawToTensorFeatureExtractorCombiner) or \
            (isinstance(self.feature_extractor_combiner, type) and
             issubclass(self.feature_extractor_combiner, RawToTensorFeatureExtractorCombiner))
        mode = getattr(self.feature_extractor_combiner, "MODE", 4)
        clip, resize = self.clip_boxes, self.resize_boxes

        def start():
            torch.cuda.set_device(dev)
            with torch.cuda.stream(side):
                boxes, counts, _ = transform_boxes(bbox_batch, size, clip, resize)     # (host boxes: nothing of `main` needed)
                side.wait_stream(main)                   # the coordinates may have been produced on the caller's stream
                started = _SelectStarted(_coords_to_device(coords), boxes)
            return boxes, counts, started

        def finish(boxes, counts, started):
            with torch.cuda.stream(side):
                sel = started.finish()
                md = None
                if want_md and sel.src_row.shape[0]:
                    levels = n_levels or Metadata.LEVELS_HINT.get(size, 1)
                    lv, ok = size, 1
                    while ok < levels and all(v % 2 == 0 for v in lv):
                        lv, ok = tuple(v // 2 for v in lv), ok + 1
                    md = Metadata(3)
                    md._prepared_for = (sel.new_coords, sel.new_coords._version)
                    md.build_native(size, sel.new_coords, sel.n_boxes, mode, ok, 3, xcd_order=xcd_order)
                    md._unrequested = set(md.strided)
                    md.ready_event = torch.cuda.Event()
                    md.ready_event.record(side)
                for t in (sel.src_row, sel.box_of, sel.new_coords):
                    if t is not None:
                        t.record_stream(main)
                ev = torch.cuda.Event()
                ev.record(side)
            return boxes, counts, sel, md, ev

        if split:
            begun = start()
            done = []

            def get():
                if not done:
                    done.append(finish(*begun))
                return done[0]
        elif in_thread:
            fut = _roi_pool.submit(lambda: finish(*start()))
            get = fut.result
        else:
            done = finish(*start())
            get = lambda: done

        class _Pending:
            def result(self_inner):
                boxes, counts, sel, md, ev = get()
                torch.cuda.current_stream().wait_event(ev)
                return boxes, counts, sel, md
        return _Pending()


_roi_pool = None
_roi_stream = {}


class SparseRoiExtraCut(torch.nn.Module):
    """``SparseRoiExtraCut(feature_extractor_combiner)`` (roi_select_sparse.py:8-26): a second feature map cut with the
    selection of an earlier SparseRoiCut.  `selection[0]` is the `RoiSelection` list (re-used as is) or the reference's
    bool matrix (converted to the list once)."""

    def __init__(self, feature_extractor_combiner):
        super().__init__()
        self.feature_extractor_combiner = feature_extractor_combiner

    def forward(self, feature_map, selection, new_coords=None):
        fec = self.feature_extractor_combiner
        inside, bbox_sample_count, batch_splits = selection
        old_coords, old_features, spatial_size, batch_splits = fec.extract(feature_map)
        sel = inside if isinstance(inside, RoiSelection) else selection_from_matrix(inside)
        new_features = select_features(old_features, sel)
        if fec.NEED_COORDS and new_coords is None:
            new_coords = select_coords(old_coords, sel)
        return fec.combine(new_coords, new_features, spatial_size, sel.n_boxes)


def selection_from_matrix(is_inside) -> RoiSelection:
    """The list form of a reference-style bool [BB, N] matrix (row-major nonzero = box-major, ascending point row)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    m = torch.as_tensor(is_inside).to(dev)
    bb, n = m.shape
    nz = m.nonzero()
    prefix = [0] + torch.cumsum(m.sum(1), 0).tolist()
    return RoiSelection(nz[:, 1].to(torch.int32).contiguous(), nz[:, 0].to(torch.int32).contiguous(), prefix, n, bb)


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
