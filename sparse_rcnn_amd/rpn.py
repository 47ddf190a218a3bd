"""RPN boundary of the detection + mask step (BASELINE configs[2] "Backbone + RPN + sparse ROI-crop mask
head"): the proposals of a step come out of the SAME forward that feeds the mask head.

    encoder level (sparse, stride 8)  --scn.SparseToDense-->  dense [B, C, X/8, Y/8, Z/8]
        --dilation stack (Conv3d 3^3 + ReLU; engine "tiles", the default: this library's tile kernels on a fully active grid;
          engine "miopen": torch.nn.functional.conv3d -- SURVEY §2 row 8 leaves the dense RPN to PyTorch)-->
        --1x1x1 head--> per anchor 6 box deltas + 1 score          (anchor_network.py:73-124 `AnchorNetworkConv`)
        --anchors that leave the scene dropped: deltas / scores / anchors compacted (anchor.py:103-113, :177-197)-->
        --RoiSelector: detach, sigmoid, top-k (scn_topk_boxes), decode + clip the selected anchors, greedy NMS (scn_nms_bits), keep `post`-->
        list of fp32 boxes [n_i, 2, 3] per sample                    (proposal_selector.py:23-89; bbox.py:139-165,367-398)

What this package contributes to it: SparseToDense (A13, scn_elem.hip), the one-launch NMS (proposals.py) and the consumer
of the boxes (roi.SparseRoiCut).  The dense layers are plain torch modules with the reference's structure
(module_factory.py:581-611 `get_dilation_network`: `num_dilations` x [same convolution + ReLU] behind a SparseToDense;
`get_same_convolution` :396-414: kernel 3, dilation 1, padding 1) -- nn.Conv3d parameters with the reference's names / shapes.

Two engines evaluate those layers (same parameters, same mathematics):
  "miopen"  torch.nn.functional.conv3d on the NCDHW volume.  On this stack MIOpen takes its GEMM fallback solvers for the 3-D
            backward passes (im2col, 450 MB of workspace): 2.4 ms forward but 43 ms forward + backward in fp32, 22 ms under
            bf16 autocast, for a 64 x 64 x 32 volume -- five times the whole sparse step (profiles/r5_dense_rpn_probe.txt).
  "tiles"   (default) a dense same-convolution IS a submanifold convolution on a grid whose every site is active: the volume
            is kept channels-last as the feature slab [B X Y Z, C] of a fully active Metadata (built once per volume shape and
            kept), and the 3^3 layers run on this package's tile kernels (k_conv_ts / k_conv_tb, k_wgrad_*), the 1x1 head on
            the row GEMM.  W_scn[(a*3+b)*3+c, ci, co] = W_torch[co, ci, a, b, c] (SURVEY Appendix B dense identity).
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib as L
from . import modules as M
from .proposals import ProposalSelector

class _DilateGather(torch.autograd.Function):
    """h[cell] = bias + sum_o P[map[cell + d_o]][o]  (scn_dilate_gather_fwd / _bwd): the scatter half of a dense 3^3
    same-convolution whose input volume is non-zero on the active rows only; P = X @ [W[0] | ... | W[26]] is one row GEMM."""

    @staticmethod
    def forward(ctx, P, bias, cmap, cell_of_row, batch, size):
        from . import functional as F
        P = F._feat(P)
        hb = P.dtype == torch.bfloat16
        n, c = P.shape[0], P.shape[1] // 27
        cells = batch * size[0] * size[1] * size[2]
        out = torch.empty((cells, c), dtype=P.dtype, device=P.device)
        hs = L.host_i64(3)
        hs[0], hs[1], hs[2] = size
        b = None if bias is None else F._f32(bias)
        L.check(L.lib().scn_dilate_gather_fwd(L.ptr(P), L.ptr(cmap), batch, hs, c, 1 if hb else 0, L.ptr(b), L.ptr(out), L.stream()))
        ctx.save_for_backward(cell_of_row)
        ctx.meta = (n, c, tuple(size), hb, bias is not None)
        return out

    @staticmethod
    def backward(ctx, dOut):
        from . import functional as F
        (cell_of_row,) = ctx.saved_tensors
        n, c, size, hb, has_bias = ctx.meta
        dOut = dOut.to(torch.bfloat16).contiguous() if hb else F._f32(dOut)
        dP = torch.empty((n, 27 * c), dtype=dOut.dtype, device=dOut.device)
        hs = L.host_i64(3)
        hs[0], hs[1], hs[2] = size
        L.check(L.lib().scn_dilate_gather_bwd(L.ptr(dOut), L.ptr(cell_of_row), n, hs, c, 1 if hb else 0, L.ptr(dP), L.stream()))
        db = F.colsum(dOut) if has_bias else None
        return dP, db, None, None, None, None


# the reference's anchor table (scannet_config/network.py:7-23, metres) at its input voxel size with the up-convoluted anchor
# network (run.py:361: 0.0375 m): the three small anchors on the first anchor level, the eleven large ones on the second
REF_RAW_ANCHORS_M = ((0.3752, 0.3752, 0.4221), (0.6566, 0.6566, 0.5159), (0.6566, 0.6566, 0.9380), (0.4221, 0.4221, 1.6415),
                     (0.1876, 1.3132, 1.0318), (0.3283, 0.9849, 1.8291), (0.7035, 1.5008, 0.8442), (1.3132, 0.1876, 1.0318),
                     (0.9849, 0.3283, 1.7822), (1.5008, 0.7035, 0.8442), (0.8442, 2.1574, 0.3752), (2.1574, 0.8442, 0.3752),
                     (2.4857, 1.1256, 1.0318), (1.1256, 2.4857, 1.0318))
REF_VOXEL_M = 0.0375
REF_ANCHOR_LEVELS_VOXELS = tuple(tuple(tuple(v / REF_VOXEL_M for v in a) for a in lv)
                                 for lv in (REF_RAW_ANCHORS_M[:3], REF_RAW_ANCHORS_M[3:]))

# anchor edge lengths in voxels at the one anchor level (stride 8); the synthetic boxes of cfg 3 have edges 8-96
DEFAULT_ANCHORS = ((12.0, 12.0, 12.0), (24.0, 24.0, 24.0), (48.0, 48.0, 32.0), (96.0, 96.0, 48.0))


class DenseRpn(nn.Module):
    """`DenseRpn(channels, stride)`: SparseToDense -> dilation stack -> AnchorNetworkConv-style head.
    forward(level_tensor) -> (rpn_bbox [B, N, 2, 3] = (position delta, log-size delta), rpn_score [B, N] raw,
    anchors [N, 2, 3] = (centre, size)), N = X' Y' Z' x A, spatial-major / anchor-minor."""

    ENGINE = "tiles"            # class-wide default: "tiles" | "miopen"
    # "tiles": the FIRST layer of the stack sees a volume that is non-zero on the active rows only (2.3 % of the cells at the
    # stride-8 level of the cfg-2 scene): it runs as ONE row GEMM over the active rows, P = X @ [W[0] | ... | W[26]], + a
    # dilation gather (scn_dilate_gather_*) -- 1.3 GFLOP instead of the volume's 58; False: on the tile kernels like the rest
    SPARSE_FIRST = True

    def __init__(self, channels, stride=8, width=32, num_dilations=2, anchors=DEFAULT_ANCHORS, autocast_bf16=False, engine=None,
                 keep_inside=True, allowed_border=0):
        """keep_inside: return only the anchors that lie inside the scene (+ allowed_border), as the reference's
        `rpn_bbox_score_splitter` does (anchor.py:103-113,177-197; `allowed_border=0`, scannet_config/run.py:841) -- deltas,
        scores and anchors compacted in anchor order.  The reference's own shape is width 128 / 256 and num_dilations 5
        (run.py:525-536,609): the defaults here are the light stand-in of `--workload cfg3-rpn` (trainstep.py)."""
        super().__init__()
        self.engine = engine
        self.keep_inside, self.allowed_border = bool(keep_inside), float(allowed_border)
        self.channels, self.stride, self.width = int(channels), int(stride), int(width)
        self.to_dense = M.SparseToDense(3, self.channels)
        layers, cin = [], self.channels
        for d in range(num_dilations):                       # get_dilation_network: [same convolution (3^3, dilation 1) + ReLU] x n
            layers += [nn.Conv3d(cin, self.width, 3, padding=1), nn.ReLU(inplace=True)]
            cin = self.width
        self.stack = nn.Sequential(*layers)
        self.register_buffer("anchor_sizes", torch.tensor(anchors, dtype=torch.float32))
        self.n_anchors = len(anchors)
        self.head = nn.Conv3d(cin, self.n_anchors * 7, 1)    # anchor_network.py:88-92: num_anchors x (2 x num_dims + 1)
        self.autocast_bf16 = bool(autocast_bf16)
        self._anchor_cache = {}
        self._dense_md = {}
        self._flag_host = {}

    def __getstate__(self):
        d = self.__dict__.copy()                 # (the fully active Metadata objects: device index structures, rebuilt on use)
        d["_dense_md"], d["_anchor_cache"], d["_flag_host"] = {}, {}, {}
        return d

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

    def inside_for(self, shape, device):
        """-> (int64 indices of the anchors of `anchors_for(shape)` that lie inside the scene, those anchors); cached per shape."""
        key = ("inside", tuple(shape), str(device))
        got = self._anchor_cache.get(key)
        if got is None:
            a = self.anchors_for(shape, device)
            scene = torch.tensor([int(v) * self.stride for v in shape], dtype=torch.float32, device=a.device)
            idx = inside_indicator(a, scene, self.allowed_border).nonzero().squeeze(1)
            got = self._anchor_cache[key] = (idx, a[idx])
        return got

    def _finish(self, raw, shape):
        """raw [B, N, 7] in anchor order -> (rpn_bbox, rpn_score, anchors), inside-the-scene anchors only when asked."""
        if self.keep_inside:
            idx, anchors = self.inside_for(shape, raw.device)
            raw = raw.index_select(1, idx)
        else:
            anchors = self.anchors_for(shape, raw.device)
        return raw[..., :6].reshape(raw.shape[0], -1, 2, 3), raw[..., 6], anchors

    # ---- "tiles": the dense volume as the slab of a fully active grid ------------------------------------------------------
    def dense_metadata(self, size, batch, device):
        """Metadata of the grid `size` with EVERY site of every sample active, rows in (b, x, y, z) order -- row r is cell
        r of the channels-last volume [B, X, Y, Z, C].  Depends on the shape only: built once, kept."""
        from .metadata import Metadata
        key = (tuple(size), int(batch), str(device))
        md = self._dense_md.get(key)
        if md is None:
            X, Y, Z = size
            g = torch.stack(torch.meshgrid(torch.arange(batch), torch.arange(X), torch.arange(Y), torch.arange(Z), indexing="ij"),
                            -1).reshape(-1, 4)
            coords = g[:, [1, 2, 3, 0]].contiguous().to(torch.int64)          # (x, y, z, batch), batch-major rows
            md = self._dense_md[key] = Metadata(3).build_native(size, coords.to(device), batch, 4, 1, 3)
        return md

    def _forward_tiles(self, level_tensor):
        from . import functional as F
        feats = level_tensor.features
        md = level_tensor.metadata
        size = tuple(int(v) for v in level_tensor.spatial_size)
        B = md.n_samples
        # cell of every active row, ((b X + x) Y + y) Z + z, and the inverse map (-1 on empty cells): scn_cell_map, one launch
        n_cells = B * size[0] * size[1] * size[2]
        n_rows = feats.shape[0]
        ridx = torch.empty(n_rows, dtype=torch.int64, device=feats.device)
        cmap = torch.empty(n_cells, dtype=torch.int32, device=feats.device)
        flag = torch.empty(1, dtype=torch.int32, device=feats.device)      # (rows outside the volume: none by construction)
        hs = L.host_i64(3)
        hs[0], hs[1], hs[2] = size
        L.check(L.lib().scn_cell_map(L.ptr(md.grid(size).coords), n_rows, B, hs, L.ptr(ridx), L.ptr(cmap), L.ptr(flag), L.stream()))
        dmd = self.dense_metadata(size, B, feats.device)
        ssz = torch.as_tensor(size, dtype=torch.long)
        layers = list(self.stack)
        if self.SPARSE_FIRST and isinstance(layers[0], nn.Conv3d) and layers[0].in_channels % 8 == 0 and layers[0].out_channels % 8 == 0:
            # SparseToDense + the first same-convolution without ever building the (mostly zero) input volume
            conv = layers.pop(0)
            Wall = conv.weight.permute(1, 2, 3, 4, 0).reshape(conv.in_channels, 27 * conv.out_channels)     # [Cin][o][Cout]
            P = F.NetworkInNetworkFunction.apply(feats, Wall, None)
            x = _DilateGather.apply(P, conv.bias, cmap, ridx, B, size)
        else:
            x = feats.new_zeros((B * size[0] * size[1] * size[2], feats.shape[1])).index_copy(0, ridx, feats)   # SparseToDense, channels-last
        relu_in = False
        for layer in layers:
            if isinstance(layer, nn.ReLU):
                relu_in = True                                                  # fused into the next layer's gather
                continue
            W = layer.weight.permute(2, 3, 4, 1, 0).reshape(27, layer.in_channels, layer.out_channels)
            x = M._conv_input(x, layer.in_channels, layer.out_channels, True)
            x = F.SubmanifoldConvolutionFunction.apply(x, W, layer.bias, dmd, ssz, 3, relu_in, None)
            relu_in = False
        x = x.float()
        if relu_in:
            x = F.ReLUFunction.apply(x)
        Wh = self.head.weight.reshape(self.head.out_channels, self.head.in_channels).t()
        raw = F.NetworkInNetworkFunction.apply(x, Wh, self.head.bias)           # [B X Y Z, A * 7]: spatial-major, anchor-minor
        out = self._finish(raw.view(B, -1, 7), size)
        # scn_cell_map's count of rows outside the volume (a level tensor whose spatial_size / batch disagrees with its
        # coordinates): copied to pinned memory behind the launch, carried on rpn_score, read by RoiSelector.finish after the
        # host wait it makes anyway
        fh = self._flag_host.get(str(feats.device))
        if fh is None:
            fh = self._flag_host[str(feats.device)] = torch.zeros(1, dtype=torch.int32).pin_memory()
        fh.copy_(flag, non_blocking=True)
        out[1].cell_flags = [fh]
        return out

    def forward(self, level_tensor):
        if (self.engine or self.ENGINE) == "tiles" and level_tensor.features.is_cuda:
            return self._forward_tiles(level_tensor)
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
        return self._finish(raw, shape)


def inside_indicator(anchors, scene_shape, allowed_border=0.0):
    """AnchorDescriptionMultiLevel.inside_indicator (ndsis/modules/anchor.py:103-113): anchors [N, 2, 3] = (centre, size),
    True where start >= -border and end <= scene_shape + border on every axis (calc_start_end, utils/bbox.py:267-287)."""
    half = anchors[..., 1, :] / 2
    start, end = anchors[..., 0, :] - half, anchors[..., 0, :] + half
    scene_shape = torch.as_tensor(scene_shape, dtype=anchors.dtype, device=anchors.device)
    return torch.all(start >= -allowed_border, dim=-1) & torch.all(end <= scene_shape + allowed_border, dim=-1)


def decode_boxes(anchors, deltas, scene_shape=None):
    """bbox_transform_inv (ndsis/utils/bbox.py:139-165,267-287,367-398): anchors (centre, size), deltas (position, log size)
    -> boxes (start, stop).  fp32, same operation order as the reference.  scene_shape: clip the boxes to [0, scene_shape]
    as `AnchorDescriptionMultiLevel.forward` does (anchor.py:218-227, `clip_boxes` utils/bbox.py:16-34)."""
    pos = deltas[..., 0, :] * anchors[..., 1, :] + anchors[..., 0, :]
    size = torch.exp(deltas[..., 1, :]) * anchors[..., 1, :]
    half = size / 2
    boxes = torch.stack((pos - half, pos + half), dim=-2)
    if scene_shape is not None:
        boxes = torch.min(boxes, torch.as_tensor(scene_shape, dtype=boxes.dtype, device=boxes.device)).clamp(min=0)
    return boxes


class MultiLevelRpn(nn.Module):
    """The reference's RPN shape with several anchor levels (scannet_config/run.py:525-536: `upconvoluted_anchornetwork`,
    one dilation network per anchor level, `anchor_output_channels = [128, 256]`, `num_dilations = 5` :609): one DenseRpn per
    level -- SparseToDense of THAT encoder level, its own dilation stack, a 1x1 head for its own anchors -- the levels'
    outputs concatenated in level order and the anchors that leave the scene dropped (`rpn_bbox_score_splitter`,
    anchor.py:177-197).  The reference's heads are transposed convolutions that refine the anchor grid
    (`AnchorNetworkUpsample`, anchor_network.py:127-219): dense torch modules, out of scope; the 1x1 heads
    here keep every level's anchors on the level's own grid.
    levels: [(channels, stride, width, anchors [A, 3] in voxels)];  forward(level_tensors) -> (rpn_bbox [B, N, 2, 3],
    rpn_score [B, N], anchors [N, 2, 3]) over the inside anchors of all levels."""

    def __init__(self, levels, num_dilations=5, autocast_bf16=False, engine=None, allowed_border=0):
        super().__init__()
        self.levels = nn.ModuleList(DenseRpn(c, stride, width, num_dilations, tuple(map(tuple, a)), autocast_bf16, engine,
                                             keep_inside=False) for (c, stride, width, a) in levels)
        self.allowed_border = float(allowed_border)
        self._cache = {}

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_cache"] = {}
        return d

    def forward(self, level_tensors):
        outs = [rpn(t) for rpn, t in zip(self.levels, level_tensors)]
        scene = tuple(int(v) * self.levels[0].stride for v in level_tensors[0].spatial_size)
        return self.combine(outs, scene)

    def combine(self, outs, scene_shape):
        """Per-level (rpn_bbox, rpn_score, anchors) over ALL anchors of each level -> the reference's concatenation in level
        order with the anchors that leave `scene_shape` dropped (`rpn_bbox_score_splitter`, anchor.py:177-197)."""
        key = (tuple(scene_shape), tuple(o[2].shape[0] for o in outs), str(outs[0][0].device))
        got = self._cache.get(key)
        if got is None:
            anchors = torch.cat([o[2] for o in outs], 0)
            idx = inside_indicator(anchors, torch.tensor(scene_shape, dtype=torch.float32), self.allowed_border).nonzero().squeeze(1)
            got = self._cache[key] = (idx, anchors[idx])
        idx, anchors = got
        rpn_bbox = torch.cat([o[0] for o in outs], 1).index_select(1, idx)
        rpn_score = torch.cat([o[1] for o in outs], 1).index_select(1, idx)
        rpn_score.cell_flags = [f for o in outs for f in getattr(o[1], "cell_flags", [])]
        return rpn_bbox, rpn_score, anchors


class RoiSelector(nn.Module):
    """`RoiSelector(num_keep_pre_nms, num_keep_post_nms, thresh_nms, detach=True)` (proposal_selector.py:23-50):
    forward(rpn_bbox, rpn_score, anchors[, scene_shape]) -> (roi_score, roi_bbox, roi_index) lists per sample.  `anchors` is
    either the anchors tensor [N, 2, 3] (decoded with `decode_boxes` AFTER the top-k, clipped to `scene_shape` when given) or --
    the reference's own third argument -- a callable `anchor_description` that maps rpn_bbox to boxes (decoded before the top-k)."""

    def __init__(self, num_keep_pre_nms=1024, num_keep_post_nms=64, thresh_nms=0.5, detach=True):
        super().__init__()
        self.proposal_selector = ProposalSelector(num_keep_pre_nms, num_keep_post_nms, thresh_nms)
        self.detach = detach
        self._scene_cache = {}                   # (scene shape, device) -> device tensor: no host-to-device copy per step

    def forward(self, rpn_bbox, rpn_score, anchors, scene_shape=None):
        return self.finish(self.start(rpn_bbox, rpn_score, anchors, scene_shape))

    def start(self, rpn_bbox, rpn_score, anchors, scene_shape=None):
        """Everything up to and including the NMS launch, nothing awaited (ProposalSelector.start).  scene_shape: the
        proposals are clipped to the scene, as `anchor_description(rpn_bbox)` does in the reference (anchor.py:218-227)."""
        flags = getattr(rpn_score, "cell_flags", [])
        if self.detach:
            rpn_bbox, rpn_score = rpn_bbox.detach(), rpn_score.detach()
        if callable(anchors) and not torch.is_tensor(anchors):
            # the reference's own call: `anchor_description(rpn_bbox)` -- an AnchorDescriptionMultiLevel (anchor.py:218-227) or
            # any callable that turns the deltas into clipped (start, stop) boxes -- decodes every anchor before the selection
            return self.proposal_selector.start(torch.sigmoid(rpn_score), anchors(rpn_bbox)) + (flags,)
        if scene_shape is not None and not torch.is_tensor(scene_shape):
            key = (tuple(float(v) for v in scene_shape), str(rpn_bbox.device))
            t = self._scene_cache.get(key)
            if t is None:
                t = self._scene_cache[key] = torch.tensor(key[0], dtype=torch.float32, device=rpn_bbox.device)
            scene_shape = t
        decode = decode_boxes if scene_shape is None else (lambda a, d: decode_boxes(a, d, scene_shape))
        return self.proposal_selector.start_from_deltas(torch.sigmoid(rpn_score), rpn_bbox, anchors, decode) + (flags,)

    def finish(self, state):
        out = self.proposal_selector.finish(state[:4])          # (waits on the host: the pinned cell flags are complete)
        if any(int(f[0]) != 0 for f in state[4]):
            raise L.ScnError("DenseRpn: rows of a level tensor lie outside its spatial_size / batch (scn_cell_map counted "
                             f"{[int(f[0]) for f in state[4]]}): its Metadata and its coordinates disagree")
        return out


class _TrainValSelector(nn.Module):
    """Training-mode / evaluation-mode pair of selectors (the reference wraps its two RoiSelectors in a `ConditionalStage`,
    custom_container.py:102-116: every call and attribute goes to the member that matches `self.training`)."""

    def __init__(self, train_module, val_module):
        super().__init__()
        self.train_module, self.val_module = train_module, val_module

    def _member(self):
        return self.train_module if self.training else self.val_module

    def forward(self, *a, **kw):
        return self._member()(*a, **kw)

    def start(self, *a, **kw):
        return self._member().start(*a, **kw)

    def finish(self, state):
        return self._member().finish(state)


def get_roi_selector(num_keep_pre_nms=1000, num_keep_post_nms=500, thresh_nms=0.5, val_num_keep_pre_nms=None,
                     val_num_keep_post_nms=None, val_thresh_nms=None):
    """`get_roi_selector` (proposal_selector.py:6-20) with the reference's defaults: one RoiSelector, or -- when evaluation has
    its own numbers (scannet_config/run.py:847-853: 1024 / 256 / 0.5 in training, 1024 / 32 or 256 / 0.3 in evaluation) -- a pair
    that follows `module.training`."""
    sel = RoiSelector(num_keep_pre_nms, num_keep_post_nms, thresh_nms)
    if val_num_keep_pre_nms:
        sel = _TrainValSelector(sel, RoiSelector(val_num_keep_pre_nms, val_num_keep_post_nms, val_thresh_nms))
    return sel
