"""RPN boundary of the detection + mask step (BASELINE configs[2] "Backbone + RPN + sparse ROI-crop mask
head"): the proposals of a step come out of the SAME forward that feeds the mask head.

    encoder level (sparse, stride 8)  --scn.SparseToDense-->  dense [B, C, X/8, Y/8, Z/8]
        --dilation stack (Conv3d 3^3 + ReLU, torch / MIOpen: SURVEY §2 row 8 leaves the dense RPN to PyTorch)-->
        --1x1x1 head--> per anchor 6 box deltas + 1 score          (anchor_network.py:73-124 `AnchorNetworkConv`)
        --RoiSelector: detach, sigmoid, top-k (scn_topk_boxes), decode the selected anchors, greedy NMS (scn_nms_bits), keep `post`-->
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

    def __init__(self, channels, stride=8, width=32, num_dilations=2, anchors=DEFAULT_ANCHORS, autocast_bf16=False, engine=None):
        super().__init__()
        self.engine = engine
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

    def __getstate__(self):
        d = self.__dict__.copy()                 # (the fully active Metadata objects: device index structures, rebuilt on use)
        d["_dense_md"], d["_anchor_cache"] = {}, {}
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
        raw = raw.view(B, -1, 7)
        return raw[..., :6].reshape(B, -1, 2, 3), raw[..., 6], self.anchors_for(size, raw.device)

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
        return self.finish(self.start(rpn_bbox, rpn_score, anchors))

    def start(self, rpn_bbox, rpn_score, anchors):
        """Everything up to and including the NMS launch, nothing awaited (ProposalSelector.start)."""
        if self.detach:
            rpn_bbox, rpn_score = rpn_bbox.detach(), rpn_score.detach()
        return self.proposal_selector.start_from_deltas(torch.sigmoid(rpn_score), rpn_bbox, anchors, decode_boxes)

    def finish(self, state):
        return self.proposal_selector.finish(state)
