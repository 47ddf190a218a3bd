"""Graph builder for the per-ROI mask branch: the topology ``SparseMaskNetwork`` has under the reference's
configuration (ndsis/modules/model.py:572-782 with scannet_config/run.py:741-810), expressed with this package's scn-API
modules and its device ROI crop.  Together with ``unet.Backbone`` it is BASELINE config 3's sparse path:

  backbone features (C0 ch, one row per active voxel)
    -> input_conv_layer   : 'B' level, 16 ch, stride 1: SubM 1^3 C0->16 + 2 residual units             (run.py:749-755)
    -> SparseFeaturemapSelectorBoth (model.py:573-596):
         OutputLayer -> one row per POINT (16 ch) ++ the raw point features (7 ch) = 23 ch
         SparseRoiCut(RawToTensor) over the selected boxes, spatial size + 32, InputLayer mode 4, batch_size = #boxes
         SparseRoiExtraCut(RawToFeaturesScene) of the raw scene with the same selection (skip features)
    -> output_conv_layer  : internal U-Net  I(23) -> B32/2 -> B48/2 -> B64/2 and back up to 23 ch      (run.py:756-775)
    -> SparseFeaturemapFirst: OutputLayer -> one row per CROPPED point                                 (model.py:647-653)
    -> Linear(23->32) -> ReLU -> Linear(32->num_classes)                                               (run.py:806; module_factory.py:700-716)

The proposal source (RPN + TrainSelector) is dense PyTorch outside the hot path; `forward` takes the
selected boxes as the reference's `selected_bbox` list.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from . import modules as M
from . import roi
from .ioLayers import OutputLayer
from .unet import SparseUNet, units


def reference_key_map(n_unet_levels=4, num_units=2, n_linear=2, with_input=True):
    """state_dict key of the reference's SparseMaskNetwork (model.py:572-782 under scannet_config/run.py:741-810:
    `input_conv_layer` = Sequential of one 'B' level, `output_conv_layer` = UnetContainer(downsampling_layer = SequentialInterims
    whose level 0 is the FLD('I') Identity, upsampling_layer.module_list = SkipConnectionReuniter per decoder level),
    `linear_layer` = Linear / ReLU stack) -> this package's parameter name (`MaskBranch.named_oracle_params`).  Checked against
    the key list of a reference SparseMaskNetwork built on this package (tests/golden/dropin_mask_network.json).
    with_input=False: `use_unet_features=False` -- no input_conv_layer (model.py:723-729)."""
    out = {}
    for t in ("weight", "bias"):
        for u in range(num_units if with_input else 0):
            out[f"input_conv_layer.0.0.0.{t}"] = f"in.{t}"
            for v, idx in enumerate((1, 3)):                     # Sequential(ReLU, SubM, ReLU, SubM) inside ConcatTable[1]
                out[f"input_conv_layer.0.1.{u}.0.1.{idx}.{t}"] = f"in.res{u}.conv{v}.{t}"
        for l in range(1, n_unet_levels):
            out[f"output_conv_layer.downsampling_layer.{l}.0.0.{t}"] = f"unet.enc{l}.in.{t}"
            for u in range(num_units):
                for v, idx in enumerate((1, 3)):
                    out[f"output_conv_layer.downsampling_layer.{l}.1.{u}.0.1.{idx}.{t}"] = f"unet.enc{l}.res{u}.conv{v}.{t}"
        for i in range(n_unet_levels - 1):
            l = n_unet_levels - 2 - i
            pre = f"output_conv_layer.upsampling_layer.module_list.{i}"
            out[f"{pre}.input_stage.1.{t}"] = f"unet.dec{l}.up.{t}"
            out[f"{pre}.channel_changer.{t}"] = f"unet.dec{l}.nin.{t}"
            for u in range(num_units):
                for v, idx in enumerate((1, 3)):
                    out[f"{pre}.output_stage.{u}.0.1.{idx}.{t}"] = f"unet.dec{l}.res{u}.conv{v}.{t}"
        for i in range(n_linear):
            out[f"linear_layer.{2 * i}.{t}"] = f"lin{i}.{t}"          # Linear, ReLU, Linear: indices 0, 2
    return out


class MaskBranch(nn.Module):
    # Where the ROI batch's selection + rulebooks are built: "0" inline on the caller's stream | "thread": helper thread +
    # its own stream | "stream": its own stream, caller's thread.  Measured on one box each (tools/ab_roi_prefetch.sh,
    # profiles/r2_ab_roi_prefetch.log): fp32 12.1-12.9 ms whichever way; bf16 9.6-10.2 ms inline, 9.7-12.8 with the helper
    # thread (two Python threads issuing launches share one interpreter lock), 10.0-11.6 on the side stream -> inline.
    # Round 4, "split": the selection's count pass is queued on the index stream when the forward starts, the scene-level input
    # stage is queued on the caller's stream, and only then the caller waits (selected rows; level sizes of the ROI batch) --
    # the GPU runs the input stage through both waits (tools/ab_env.py SCN_ROI_PREFETCH).
    PREFETCH_ROI_INDEX = os.environ.get("SCN_ROI_PREFETCH", "0")       # "0" inline | "thread" | "stream" | "split"

    def __init__(self, backbone_channels=32, raw_channels=7, input_channels=16, unet_channels=(32, 48, 64),
                 linear_channels=(32, 18), bf16_blocks=False, *, use_unet_features=True, use_raw_features=True,
                 use_skip_features=False, min_channels=16):
        """use_unet_features / use_raw_features / use_skip_features: the reference's switches (model.py:702-751) --
          unet + raw  `SparseFeaturemapSelectorBoth` (:573-596, the configured one, run.py:799-801): the ROI batch is cut from
                      [per-point output of input_conv_layer ++ raw point features];
          unet only   `SparseFeaturemapSelector` (:597-619): from the per-point output alone;
          raw only    `SparseFeaturemapSelectorRaw` (:621-637): from the raw point features, no input_conv_layer at all;
          skip        `SparseFeaturemapCombiner` (:639-651) instead of `SparseFeaturemapFirst`: the cropped raw features are
                      joined to the internal U-Net's per-point output in front of the Linear stack.
        min_channels: `unet_params['min_channels']` of the internal U-Net (run.py:786; unet.SparseUNet)."""
        super().__init__()
        if not (use_unet_features or use_raw_features):
            raise ValueError("MaskBranch: use_unet_features=False needs use_raw_features (model.py:724)")
        self.bf16 = bool(bf16_blocks)           # bf16 STORAGE of the branch's feature slabs (BASELINE configs 3-5)
        self.use_unet_features, self.use_raw_features = bool(use_unet_features), bool(use_raw_features)
        self.use_skip_features, self.raw_channels = bool(use_skip_features), int(raw_channels)
        c = (input_channels if use_unet_features else 0) + (raw_channels if use_raw_features else 0)
        self.input_conv_layer = None
        if use_unet_features:
            self.input_conv_layer = M.Sequential(M.SubmanifoldConvolution(3, backbone_channels, input_channels, 1, True),
                                                 units(input_channels, 2))
        self.output_layer = OutputLayer(3)
        # dense_inside=False: the selection stays the CSR list (no [boxes, points] matrix); every consumer here takes it
        self.output_roi_cut = roi.SparseRoiCut(roi.RawToTensorFeatureExtractorCombiner(), dense_inside=False)
        self.scene_roi_extra_cut = roi.SparseRoiExtraCut(roi.RawToFeaturesSceneFeatureExtractorCombiner())
        self.spatial_size_extention = 32                                        # model.py:701
        self.output_conv_layer = SparseUNet(c, (c,) + tuple(unet_channels), identity_first=True,
                                            bf16_blocks=bf16_blocks, min_channels=min_channels)
        c = self.output_conv_layer.out_channels                             # (min_channels: 7 in -> 16 out, model.py:789-804)
        self.roi_output_layer = OutputLayer(3)
        layers, cin = [], c + (raw_channels if use_skip_features else 0)
        for i, co in enumerate(linear_channels):
            if i:
                layers.append(nn.ReLU(inplace=True))
            layers.append(nn.Linear(cin, co))
            cin = co
        self.linear_layer = nn.Sequential(*layers)
        self.classes = cin

    def prepare_cut(self, coords, spatial_size, selected_bbox):
        """Start the crop's selection and the ROI batch's index build (they depend on coordinates and boxes only) on the
        helper thread + its own stream BEFORE the backbone runs; pass the handle to forward(..., prepared_cut=).  coords:
        int64 [N, 4] (host or device) or the int32 device copy a Metadata holds (`point_coords`)."""
        size = torch.as_tensor([int(s) for s in spatial_size], dtype=torch.long) + self.spatial_size_extention
        return self.output_roi_cut.prepare_cut_in_thread(coords, size, selected_bbox, in_thread=True)

    def forward(self, raw_scene, backbone_features, selected_bbox, prepared_cut=None):
        """raw_scene: the collate tuple (coords, features, spatial_size, batch_size, batch_splits) with `features` on
        the device; backbone_features: SparseConvNetTensor (unet_feature_maps[-1]; unused -- may be None -- without
        use_unet_features); selected_bbox: list (one per sample) of fp32 [n, 2, 3] boxes.
        -> (per-point-per-class mask logits [M, classes], selection)."""
        coords, features, spatial_size, *other, batch_splits = raw_scene
        # the InputLayer of this scene left an int32 device copy of the point coordinates in its Metadata: the crop reads
        # that instead of converting (and range-checking, one host wait) the int64 coordinates a second time
        pc = getattr(getattr(backbone_features, "metadata", None), "point_coords", None)
        if pc is not None and pc.shape[0] == coords.shape[0]:
            coords = pc
            raw_scene = (pc,) + tuple(raw_scene[1:])
        size = torch.as_tensor([int(s) for s in spatial_size], dtype=torch.long) + self.spatial_size_extention
        # the crop's selection and the ROI batch's index structures depend on coordinates and boxes only: a helper thread
        # builds them (its own high-priority stream) while the scene-level layers below are queued and run
        how = self.PREFETCH_ROI_INDEX
        how = {True: "thread", False: "0", "1": "thread"}.get(how, how)
        pending = prepared_cut
        if pending is None and how != "0":
            from . import metadata as MD
            pending = self.output_roi_cut.prepare_cut_in_thread(
                coords, size, selected_bbox, in_thread=how == "thread", split=how == "split",
                xcd_order=True if (self.bf16 and MD.XCD_ORDER_BF16) else None)
        # bf16 storage: the scene-level units, the per-point gather (OutputLayer), the per-point slab and the crop's feature
        # gather all run on bf16 rows (round 3: no fp32 island between backbone and internal U-Net); only the InputLayer's
        # mean over the cropped points accumulates in fp32 / fp64
        parts = ()
        if self.use_unet_features:
            converted = self._input_stage_exec(backbone_features)        # one C call each way (executor.py) where it applies
            if converted is None and self.bf16:
                converted = self.input_conv_layer(M.CastFeatures(torch.bfloat16)(backbone_features))
            elif converted is None:
                converted = self.input_conv_layer(backbone_features)
            parts += (self.output_layer(converted),)
        if self.use_raw_features:
            parts += (features.to(parts[0].dtype if parts else (torch.bfloat16 if self.bf16 else features.dtype)),)
        unet = self.output_conv_layer
        if unet.phys0 != unet.channels[0]:      # 23 -> 24 columns: zero column appended where the slab is assembled anyway
            parts += (parts[0].new_zeros((parts[0].shape[0], unet.phys0 - unet.channels[0])),)
        combined = torch.cat(parts, dim=-1) if len(parts) > 1 else parts[0]
        roi_tensor, selection = self.output_roi_cut((coords, combined, size, *other, batch_splits), selected_bbox,
                                                    prepared=pending)
        skip_features = self.scene_roi_extra_cut(raw_scene, selection)
        if len(skip_features) == 0 or roi_tensor is None:
            return skip_features.new_zeros((0, self.classes)), selection
        if self.bf16:           # the ROI batch's SubM tiles (built on first use below) with the XCD-local hand-out order
            from . import metadata as MD
            if MD.XCD_ORDER_BF16 and not roi_tensor.metadata.subm:
                roi_tensor.metadata.xcd_order = True
        out = self.roi_output_layer(unet(roi_tensor))            # [cropped points, phys0]; the pad column is zero
        if self.use_skip_features:              # SparseFeaturemapCombiner (model.py:639-651): [U-Net output ++ cropped raw features]
            out = torch.cat((out[:, :unet.out_channels], skip_features.to(out.dtype)), dim=-1)
        return self._linear(out), selection

    def __getstate__(self):
        d = self.__dict__.copy()                 # (the compiled input stage: ctypes tables, rebuilt on first use by a copy)
        d.pop("_input_stage", None)
        return d

    def _input_stage_exec(self, fmap):
        """input_conv_layer (SubM 1^3 + residual units on the scene's level 0; bf16 storage: cast in, cast out) through the
        step executor, or None when it does not apply."""
        from . import executor as EX, functional as F, profiling
        from .tensor import SparseConvNetTensor
        f = fmap.features
        if (not (EX.ENABLED and SparseUNet.EXEC) or not profiling.exec_ok()
                or not (f.is_cuda and f.dtype == torch.float32 and f.shape[0] > 0)):
            return None
        st = self.__dict__.get("_input_stage")
        if st is None:
            blocks = EX._plain_blocks(self.input_conv_layer[1])
            head = self.input_conv_layer[0]
            st = False
            if blocks is not None and head.nOut % 8 == 0 and head.nIn % 8 == 0 and EX._require_bias(head):
                st = EX.compile_encoder_stage(0, head, blocks, head.nIn, self.bf16, cast_first=True, cast_last=False)
            object.__setattr__(self, "_input_stage", st)
        if not st:
            return None
        lv = EX.build_levels(fmap.metadata, fmap.spatial_size, 1)
        if lv is None:
            return None
        return SparseConvNetTensor(features=EX.run_stage(st, lv, [f]), metadata=fmap.metadata, spatial_size=fmap.spatial_size)

    # parameter naming shared with the checker (tests map the oracle's mask-branch parameters by these names)
    def named_oracle_params(self):
        out = {}
        ic = self.input_conv_layer
        if ic is not None:
            out["in.weight"], out["in.bias"] = ic[0].weight, ic[0].bias
        for u, block in enumerate(ic[1] if ic is not None else ()):
            convs = [m for m in block[0][1] if isinstance(m, M.SubmanifoldConvolution)]
            for v, cv in enumerate(convs):
                out[f"in.res{u}.conv{v}.weight"], out[f"in.res{u}.conv{v}.bias"] = cv.weight, cv.bias
        for k, p in self.output_conv_layer.named_oracle_params().items():
            out["unet." + k] = p
        lin = [m for m in self.linear_layer if isinstance(m, nn.Linear)]
        for i, m in enumerate(lin):
            out[f"lin{i}.weight"], out[f"lin{i}.bias"] = m.weight, m.bias
        return out

    def load_reference_state_dict(self, state_dict, prefix=None, strict=True):
        """Load the mask-network part of a checkpoint written by the REFERENCE (training.py:386-391 saves
        model.state_dict(); the SparseMaskNetwork sits under `mask_network.` in InstanceSegmentationNetwork, model.py:31-114):
        keys are mapped onto this branch's parameters by `reference_key_map`; SparseConvNet's grouped weight layout
        [fv, 1, nIn, nOut] is accepted.  prefix=None: detected from the first key ending in 'input_conv_layer.0.0.0.weight'
        ('output_conv_layer.downsampling_layer.1.0.0.weight' without use_unet_features).
        -> (missing reference keys, unused checkpoint keys under the prefix)."""
        if prefix is None:
            tail = ("input_conv_layer.0.0.0.weight" if self.input_conv_layer is not None
                    else "output_conv_layer.downsampling_layer.1.0.0.weight")
            prefix = next((k[:-len(tail)] for k in state_dict if k.endswith(tail)), "")
        own = self.named_oracle_params()
        n_lin = len([m for m in self.linear_layer if isinstance(m, nn.Linear)])
        kmap = reference_key_map(len(self.output_conv_layer.channels), n_linear=n_lin,
                                 with_input=self.input_conv_layer is not None)
        missing, used = [], set()
        with torch.no_grad():
            for rk, name in kmap.items():
                t = state_dict.get(prefix + rk)
                if t is None:
                    missing.append(prefix + rk)
                    continue
                if t.dim() == 4 and t.shape[1] == 1:
                    t = t.squeeze(1)
                p = own[name]
                if tuple(t.shape) != tuple(p.shape):
                    raise ValueError(f"{prefix + rk}: checkpoint shape {tuple(t.shape)} != {tuple(p.shape)} ({name})")
                p.copy_(t)
                used.add(prefix + rk)
        mine = ("input_conv_layer.", "output_conv_layer.", "linear_layer.")
        unused = [k for k in state_dict if k.startswith(prefix) and k[len(prefix):].startswith(mine) and k not in used]
        if strict and (missing or unused):
            raise KeyError(f"reference checkpoint mismatch: missing {missing[:4]}... unused {unused[:4]}...")
        return missing, unused

    def _linear(self, x):
        """The Linear / ReLU stack (module_factory.py:700-716) on the library's row GEMM: nn.Linear parameters (state_dict
        names and shapes kept), y = x W^T + b through NetworkInNetworkFunction with the transposed weight view; a padded
        input column meets a zero weight row."""
        from . import functional as F
        for m in self.linear_layer:
            if isinstance(m, nn.Linear):
                W = m.weight.t()
                if x.shape[1] != W.shape[0]:
                    W = torch.nn.functional.pad(W, (0, 0, 0, x.shape[1] - W.shape[0]))
                x = F.NetworkInNetworkFunction.apply(x, W, m.bias)
            else:
                x = F.ReLUFunction.apply(x)
        return x
