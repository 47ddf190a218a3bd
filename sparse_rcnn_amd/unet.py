"""Graph builder for the benchmark backbone: the topology the reference's factory code produces for its sparse
U-Net, expressed with this package's scn-API modules.

  encoder level l : {l = 0: SubM 1^3 Cin->C0 | l > 0: Convolution 2^3/2 C(l-1)->Cl}            (module_factory.py:438-486)
                    + num_units x  [ x + SubM3(ReLU(SubM3(ReLU(x)))) ]                            (:127-183, relu_first)
  decoder level l : ReLU -> Deconvolution 2^3/2 -> JoinTable([up, skip]) -> NetworkInNetwork(2C->C)
                    -> num_units x residual                                                       (:533-578; custom_container.py:70-83)

Flags reproduced: relu_first=True, main_path_relu=False, bottleneck_divisor=0, drop_input_relu=True, num_units=2,
use_residuals=True, concat=True, batchnorm=False (scannet_config/run.py:516-519,609-623).
"""
from __future__ import annotations

import torch
from torch import nn

from . import modules as M
from .ioLayers import InputLayer


def residual_block(c, batchnorm=False):
    """module_factory.py:51-57 sparse_residual(inner_block, Identity, None) with relu_first inner block."""
    def act():
        return M.BatchNormReLU(c) if batchnorm else M.ReLU()
    inner = M.Sequential(act(), M.SubmanifoldConvolution(3, c, c, 3, not batchnorm),
                         act(), M.SubmanifoldConvolution(3, c, c, 3, not batchnorm))
    return M.Sequential(M.ConcatTable(M.Identity(), inner), M.AddTable())


def units(c, num_units=2, batchnorm=False):
    return M.Sequential(*[residual_block(c, batchnorm) for _ in range(num_units)])


def reference_key_map(n_levels, num_units=2):
    """state_dict key of the reference's FeatureExtractor (sparse + U-Net: `main_network` = SequentialInterims of encoder
    levels, `unet.module_list` = SkipConnectionReuniter per decoder level; module_factory.py:438-578, model.py:364-391)
    -> this package's parameter name (`SparseUNet.named_oracle_params`).  Checked against the key list of a reference
    FeatureExtractor built on this package (tests/golden/dropin_feature_extractor.json)."""
    out = {}
    for l in range(n_levels):
        for t in ("weight", "bias"):
            out[f"main_network.{l}.0.0.{t}"] = f"enc{l}.in.{t}"
            for u in range(num_units):
                for v, idx in enumerate((1, 3)):                 # Sequential(ReLU, SubM, ReLU, SubM) inside ConcatTable[1]
                    out[f"main_network.{l}.1.{u}.0.1.{idx}.{t}"] = f"enc{l}.res{u}.conv{v}.{t}"
    for i in range(n_levels - 1):
        l = n_levels - 2 - i
        for t in ("weight", "bias"):
            out[f"unet.module_list.{i}.input_stage.1.{t}"] = f"dec{l}.up.{t}"
            out[f"unet.module_list.{i}.channel_changer.{t}"] = f"dec{l}.nin.{t}"
            for u in range(num_units):
                for v, idx in enumerate((1, 3)):
                    out[f"unet.module_list.{i}.output_stage.{u}.0.1.{idx}.{t}"] = f"dec{l}.res{u}.conv{v}.{t}"
    return out


class SparseUNet(nn.Module):
    """forward(SparseConvNetTensor with Cin channels) -> SparseConvNetTensor with channels[0] channels at full
    resolution; ``.interims`` holds the encoder outputs (the reference's SequentialInterims, custom_container.py:5-12)."""

    def __init__(self, cin=7, channels=(32, 64, 128, 256), num_units=2, batchnorm=False, bf16_blocks=False,
                 identity_first=False, min_channels=0):
        """identity_first: level 0 of the encoder is the reference's FLD('I') -- no layer at all, channels[0] == cin
        (the mask head's internal U-Net: scannet_config/run.py:756-775, I -> B32/2 -> B48/2 -> B64/2; the decoder's
        level 0 still has its NiN(2 cin -> cin) + residual units, module_factory.py:533-578).
        min_channels: a decoder level is at least this wide (module_factory.py:789-804, `unet_params['min_channels'] = 16` in
        run.py:594,786): level l comes up as d_l = max(channels[l], min_channels) channels -- Deconvolution(-> d_l),
        JoinTable with the channels[l]-wide skip, NetworkInNetwork(d_l + channels[l] -> d_l), units(d_l).  It only bites below
        16 channels: the mask network cut from the raw point features alone (7 channels in, 16 out)."""
        super().__init__()
        self.channels = tuple(channels)
        self.dec_channels = tuple(max(int(c), int(min_channels)) for c in self.channels[:-1])
        self.out_channels = self.dec_channels[0] if self.dec_channels else self.channels[0]
        self.identity_first = bool(identity_first)
        if self.identity_first and self.channels[0] != cin:
            raise ValueError("identity_first needs channels[0] == cin")
        # bf16_blocks: the residual units (28 of the 44 convolutions) keep their features, intermediates and gradients
        # in bf16 (functional.ResidualBlockFunctionBF16); the strided / 1x1 layers between them stay fp32, with one
        # cast on either side of a run of units.  Not the reference's arithmetic: BASELINE configs 3-5.
        # bf16_blocks="all": every layer after the first 1x1 convolution keeps its features in bf16 (the strided
        # convolutions, deconvolutions and 1x1 layers through the bf16 entry points of the fp32-arithmetic kernels),
        # one cast after the first layer and one at the end.
        self.bf16_all = bf16_blocks == "all"
        self.bf16_blocks = bool(bf16_blocks) and not self.bf16_all
        if bf16_blocks and (batchnorm or any(c % 8 for c in self.channels[1 if identity_first else 0:])):
            raise ValueError("bf16_blocks needs channel counts that are multiples of 8 and no batch norm")
        enc = []
        for l, c in enumerate(self.channels):
            if l == 0 and self.identity_first:
                enc.append(M.Sequential(M.Identity(), M.Identity()))
                continue
            head = (M.SubmanifoldConvolution(3, cin, c, 1, True) if l == 0
                    else M.Convolution(3, self.channels[l - 1], c, (2, 2, 2), (2, 2, 2), True))
            enc.append(M.Sequential(head, units(c, num_units, batchnorm)))
        self.encoder = nn.ModuleList(enc)
        dec = []
        for l in range(len(self.channels) - 2, -1, -1):
            c, d = self.channels[l], self.dec_channels[l]
            cup = self.channels[l + 1] if l == len(self.channels) - 2 else self.dec_channels[l + 1]
            dec.append(nn.ModuleDict(dict(
                up=M.Sequential(M.ReLU(), M.Deconvolution(3, cup, d, (2, 2, 2), (2, 2, 2), True)),
                join=M.JoinTable(),
                nin=M.NetworkInNetwork(d + c, d, True),
                units=units(d, num_units, batchnorm))))
        self.decoder = nn.ModuleList(dec)
        # identity_first with a level-0 width that is no multiple of 8 (the mask head's 23): the level runs on slabs
        # zero-padded to `phys0` columns (16-byte rows: vector kernels in fp32, bf16 storage possible at all); the caller
        # hands in the padded slab and gets the padded result (modules._ConvBase.pad_out_to)
        c0 = self.channels[0]
        self.phys0 = self.out_phys = c0
        if self.identity_first and c0 % 8:
            self.phys0 = (c0 + 7) // 8 * 8
            d0 = self.decoder[-1]
            if self.out_channels == c0:                      # the whole level is padded
                self.out_phys = self.phys0
                d0["up"][1].pad_out_to = self.phys0
                d0["nin"].pad_out_to = self.phys0
                d0["nin"].in_groups = (c0, c0)
                for m in d0["units"].modules():
                    if isinstance(m, M.SubmanifoldConvolution):
                        m.pad_out_to = self.phys0
            else:                                            # min_channels widened the decoder: only the skip slab is padded
                self.out_phys = self.out_channels
                d0["nin"].in_groups = (self.out_channels, c0)
                d0["nin"].in_phys = (self.out_channels, self.phys0)

    def _pack_jobs(self, pad_requests=(), forward_only=False):
        """(W, cin, cout, n_off, flags) of every bf16 weight image this network's forward + backward will stage: forward and
        backward-data images of the SubM 3^3 layers, forward of the strided convolutions, backward-data of the
        deconvolutions (the other directions of the strided layers run on the rule-list GEMMs).
        pad_requests: the (layer, physical input width) pairs of the channel-padded layers (functional.PAD_RECORD).  The image
        of a zero-padded weight IS the image of the logical weight -- the pack zero-fills outside the layer and padding to the
        next multiple of 8 never crosses a slice boundary -- so those layers join the one pack launch with their logical
        parameters, and `aliases` = [(pad key, (cin_phys, cout_phys, n_off, flags), job index)] lets the forward register
        the image under the padded tensor it hands the kernels.  -> jobs (and the aliases as the attribute `_pack_aliases`).
        forward_only (a forward under torch.no_grad(): evaluation, training.py:244-304): only the images a forward reads --
        no backward-data image is packed (half the bytes of the pack); aliases go to `_pack_aliases_fwd`."""
        from . import functional as F
        from . import _lib as L
        lib = L.lib()
        jobs, aliases = [], []
        back = L.F_W_TRANSPOSED | L.F_OFF_REVERSE
        padded = {}
        for m, cin_phys in pad_requests:
            padded.setdefault(id(m), []).append(int(cin_phys))

        def add(m, W, cin, cout, n_off, fl, phys=None):
            if forward_only and fl:                          # a backward-data image (W^T): nobody reads it without a backward
                return
            if phys is not None:
                pc, po, key = phys
                if (pc % 8 or po % 8 or (cout > 32) != (po > 32) or -(-cin // 32) != -(-pc // 32)
                        or lib.scn_conv_tiles_bf16_image_bytes(cin, cout, n_off) != lib.scn_conv_tiles_bf16_image_bytes(pc, po, n_off)):
                    return                                   # (not the same image: packed per call as before)
                aliases.append((key, (pc, po, n_off, fl & back), len(jobs)))
            jobs.append((W, cin, cout, n_off, fl))

        for m in self.modules():
            if getattr(m, "groups", 1) != 1:
                continue                                     # grouped layers hand over a fresh block-diagonal weight per call
            if not isinstance(m, M._ConvBase) or getattr(m, "stride", (2, 2, 2)) != (2, 2, 2):
                continue                                     # (other strides: images packed per call)
            is_padded = bool(getattr(m, "pad_out_to", None)) or id(m) in padded
            if is_padded:
                pout = m.pad_out_to or m.nOut
                for pin in padded.get(id(m), []):
                    key = (id(m), pin, "w")
                    if isinstance(m, M.SubmanifoldConvolution) and m.filter_size == 3:
                        add(m, m.weight, m.nIn, m.nOut, 27, 0, (pin, pout, key))
                        add(m, m.weight, m.nOut, m.nIn, 27, back, (pout, pin, key))
                    elif isinstance(m, M.Convolution):
                        add(m, m.weight, m.nIn, m.nOut, 8, 0, (pin, pout, key))
                    elif isinstance(m, M.Deconvolution):
                        add(m, m.weight, m.nOut, m.nIn, 8, L.F_W_TRANSPOSED, (pout, pin, key))
                continue
            if m.nIn % 8 or m.nOut % 8:
                continue                                     # (rows that are no 16-byte multiples: not on the bf16 tile kernels)
            if isinstance(m, M.SubmanifoldConvolution) and m.filter_size == 3:
                add(m, m.weight, m.nIn, m.nOut, 27, 0)
                add(m, m.weight, m.nOut, m.nIn, 27, back)
            elif isinstance(m, M.Convolution):
                add(m, m.weight, m.nIn, m.nOut, 8, 0)
            elif isinstance(m, M.Deconvolution):
                add(m, m.weight, m.nOut, m.nIn, 8, L.F_W_TRANSPOSED)
        object.__setattr__(self, "_pack_aliases_fwd" if forward_only else "_pack_aliases", aliases)
        return jobs

    _PLAN_ATTRS = ("_pack_aliases", "_pack_aliases_fwd", "_pad_plan", "_pad_requests", "_pack_plan", "_pack_plan_fwd",
                   "_exec_plan_cache")

    def __getstate__(self):
        """copy.deepcopy (EMA copies) / torch.save after a forward: the compiled launch plans hold ctypes pointer arrays and
        name THIS object's modules; a copy compiles its own on first use."""
        d = self.__dict__.copy()
        for k in self._PLAN_ATTRS:
            d.pop(k, None)
        return d

    def forward(self, x, prebuild=True):
        """prebuild=False: no up-front index build -- every rulebook is requested by the first layer that needs it, the
        way the reference's module tree drives the scn surface (DropinBackbone)."""
        from . import functional as F
        # channel-padded layers (phys0): their zero-padded parameter copies come from ONE launch (functional.padded_params);
        # the first forward records which (layer, physical input width) pairs ask for one
        pads = self.__dict__.get("_pad_plan") if F.PAD_MANY else False
        if pads is None and F.PAD_RECORD is None:
            F.PAD_RECORD = rec = []
            try:
                y = self._forward_packed(x, prebuild)
            finally:
                F.PAD_RECORD = None
            object.__setattr__(self, "_pad_plan", F.PadPlan.from_requests(rec))
            object.__setattr__(self, "_pad_requests", [(m, c) for m, c in rec if isinstance(m, M._ConvBase)])
            object.__setattr__(self, "_pack_plan", None)     # (rebuilt with the padded layers' images in it)
            object.__setattr__(self, "_pack_plan_fwd", None)
            return y
        with F.padded_params(pads):
            return self._forward_packed(x, prebuild)

    def _forward_packed(self, x, prebuild):
        if self.bf16_all or self.bf16_blocks:
            from . import functional as F
            fwd_only = not torch.is_grad_enabled()           # evaluation: no backward-data images
            attr, al = ("_pack_plan_fwd", "_pack_aliases_fwd") if fwd_only else ("_pack_plan", "_pack_aliases")
            plan = self.__dict__.get(attr)
            if plan is None:                                 # host-side tables of the pack call: built once
                use_pads = F.PAD_MANY and bool(self.__dict__.get("_pad_plan"))
                plan = F.PackPlan(self._pack_jobs(self.__dict__.get("_pad_requests", ()) if use_pads else (), fwd_only))
                object.__setattr__(self, attr, plan)
            with F.packed_weights(plan) as pw:               # one pack launch for the whole network, gone after the forward
                for pad_key, dims, j in self.__dict__.get(al, ()):
                    Wp = F.PADDED.get(pad_key)               # this forward's padded tensor -> the image of its logical weight
                    if Wp is not None:
                        key = (Wp.data_ptr(),) + dims
                        F.PACKED[key] = F.PACKED[plan.keys[j]]
                        pw.keys = pw.keys + [key]
                return self._forward(x, prebuild)
        return self._forward(x, prebuild)

    # ---- step executor (executor.py): one autograd node and two C calls per LEVEL instead of one per layer -------------
    EXEC = True                 # class-wide switch (tests compare the two ways of driving the same kernels)

    def _exec_plan(self):
        """Launch plans of the levels, compiled once from the module tree; False when this network is not covered (batch
        norm, bf16_blocks=True, a block that is not the plain pre-activation unit, channel counts the two-source
        NetworkInNetwork kernel does not take)."""
        plan = self.__dict__.get("_exec_plan_cache")
        if plan is not None:
            return plan
        from . import executor as EX
        plan = False
        ch, L = self.channels, len(self.channels)
        ok = not self.bf16_blocks and all(c % 8 == 0 for c in ch[1:]) and (self.phys0 % 8 == 0)
        ok = ok and self.dec_channels == tuple(ch[:-1])      # (min_channels in effect: the layer-by-layer path)
        enc_blocks = [EX._plain_blocks(lvl[1]) if not (l == 0 and self.identity_first) else [] for l, lvl in enumerate(self.encoder)]
        dec_blocks = [EX._plain_blocks(d["units"]) for d in self.decoder]
        heads = [lvl[0] for l, lvl in enumerate(self.encoder) if not (l == 0 and self.identity_first)]
        ok = ok and EX._require_bias(*heads, *[d["up"][1] for d in self.decoder], *[d["nin"] for d in self.decoder])
        if ok and all(b is not None for b in enc_blocks + dec_blocks):
            bf16 = self.bf16_all
            phys = [self.phys0] + list(ch[1:])
            enc = []
            for l in range(L):
                if l == 0 and self.identity_first:
                    enc.append(None)
                    continue
                head = self.encoder[l][0]
                cin_phys = head.nIn if l == 0 else phys[l - 1]
                enc.append(EX.compile_encoder_stage(l, head, enc_blocks[l], cin_phys, bf16))
            dec = []
            for i, d in enumerate(self.decoder):
                l = L - 2 - i
                dec.append(EX.compile_decoder_stage(l, d["up"][1], d["nin"], dec_blocks[i], phys[l + 1], bf16))
            plan = dict(enc=enc, dec=dec)
        object.__setattr__(self, "_exec_plan_cache", plan)
        return plan

    def _forward_exec(self, x):
        """The forward through the step executor, or None when it does not apply to this call."""
        from . import executor as EX, functional as F, profiling
        from .tensor import SparseConvNetTensor
        if not (EX.ENABLED and SparseUNet.EXEC) or not profiling.exec_ok():
            return None
        f = x.features
        if not (f.is_cuda and f.dtype == torch.float32 and f.dim() == 2 and f.shape[0] > 0):
            return None
        plan = self._exec_plan()
        if not plan:
            return None
        L = len(self.channels)
        lv = EX.build_levels(x.metadata, x.spatial_size, L)
        if lv is None:                                        # an empty level: the layer-by-layer path handles it
            return None
        size = [int(s) for s in x.spatial_size]
        sizes = [torch.as_tensor([s >> l for s in size], dtype=torch.long) for l in range(L)]
        interims = []
        for l in range(L):
            if plan["enc"][l] is None:                       # identity_first: the level's slab is the input (stored: cast)
                f = f.to(torch.bfloat16) if self.bf16_all else f
            else:
                f = EX.run_stage(plan["enc"][l], lv, [f])
            interims.append(SparseConvNetTensor(features=f, metadata=x.metadata, spatial_size=sizes[l]))
        object.__setattr__(self, "interims", interims)
        self._after_encoder()
        for i, st in enumerate(plan["dec"]):
            l = L - 2 - i
            f = EX.run_stage(st, lv, [f, interims[l].features])
        if self.bf16_all:
            f = f.to(torch.float32)
        return SparseConvNetTensor(features=f, metadata=x.metadata, spatial_size=sizes[0])

    def _forward(self, x, prebuild):
        if prebuild:
            x.metadata.build_pyramid(x.spatial_size, len(self.channels), 3)
            y = self._forward_exec(x)
            if y is not None:
                return y
        interims = []
        for l, level in enumerate(self.encoder):
            x = level[0](x)
            if self.bf16_all and l == 0:
                x = M.CastFeatures(torch.bfloat16)(x)
            x = self._units(level[1], x)
            interims.append(x)
        object.__setattr__(self, "interims", interims)      # (nn.Module.__setattr__ costs ~60 us per call on the hot path)
        self._after_encoder()
        for i, d in enumerate(self.decoder):
            skip = interims[len(self.channels) - 2 - i]
            x = self._units(d["units"], d["nin"](d["join"]([d["up"](x), skip])))
        return M.CastFeatures(torch.float32)(x) if self.bf16_all else x

    def _after_encoder(self):
        """A consumer of the encoder outputs that wants its kernels queued BEFORE the decoder's (Backbone.forward(...,
        after_encoder=): the RPN heads of a detection step -- the decoder then runs while the host waits for the proposals)."""
        hook = self.__dict__.get("_encoder_hook")
        if hook is not None:
            object.__setattr__(self, "_encoder_hook", None)
            hook(self.interims)

    def _units(self, seq, x):
        if not self.bf16_blocks:
            return seq(x)
        return M.CastFeatures(torch.float32)(seq(M.CastFeatures(torch.bfloat16)(x)))

    # parameter naming shared with oracle.scn_oracle.unet_param_shapes (test infrastructure maps by these names)
    def named_oracle_params(self):
        out = {}
        for l, level in enumerate(self.encoder):
            if l == 0 and self.identity_first:
                continue
            out[f"enc{l}.in.weight"], out[f"enc{l}.in.bias"] = level[0].weight, level[0].bias
            self._res(out, f"enc{l}", level[1])
        for i, d in enumerate(self.decoder):
            l = len(self.channels) - 2 - i
            out[f"dec{l}.up.weight"], out[f"dec{l}.up.bias"] = d["up"][1].weight, d["up"][1].bias
            out[f"dec{l}.nin.weight"], out[f"dec{l}.nin.bias"] = d["nin"].weight, d["nin"].bias
            self._res(out, f"dec{l}", d["units"])
        return out

    @staticmethod
    def _res(out, prefix, unit_seq):
        for u, block in enumerate(unit_seq):
            inner = block[0][1]
            convs = [m for m in inner if isinstance(m, M.SubmanifoldConvolution)]
            for v, cv in enumerate(convs):
                out[f"{prefix}.res{u}.conv{v}.weight"], out[f"{prefix}.res{u}.conv{v}.bias"] = cv.weight, cv.bias

    def load_reference_state_dict(self, state_dict, prefix=None, strict=True):
        """Load a checkpoint written by the REFERENCE's FeatureExtractor (training.py:386-391 saves model.state_dict()):
        keys `<prefix>main_network...` / `<prefix>unet.module_list...` are mapped onto this network's parameters by
        `reference_key_map`; SparseConvNet's grouped weight layout [fv, 1, nIn, nOut] is accepted.  prefix=None: detected
        from the first key that ends in 'main_network.0.0.0.weight'.  -> (missing reference keys, unused checkpoint keys)."""
        if self.identity_first:
            raise NotImplementedError("the mask head's internal U-Net is not a FeatureExtractor checkpoint")
        if prefix is None:
            tail = "main_network.0.0.0.weight"
            prefix = next((k[:-len(tail)] for k in state_dict if k.endswith(tail)), "")
        own = self.named_oracle_params()
        kmap = reference_key_map(len(self.channels))
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
        unused = [k for k in state_dict if k.startswith(prefix + "main_network.") or k.startswith(prefix + "unet.")]
        unused = [k for k in unused if k not in used]
        if strict and (missing or unused):
            raise KeyError(f"reference checkpoint mismatch: missing {missing[:4]}... unused {unused[:4]}...")
        return missing, unused

    def load_oracle_params(self, params):
        with torch.no_grad():
            for k, p in self.named_oracle_params().items():
                p.copy_(params[k].view_as(p))


class Backbone(nn.Module):
    """InputLayer(mode 4) + SparseUNet: what model.py:414-431 runs for sparse + include_unet."""

    def __init__(self, cin=7, channels=(32, 64, 128, 256), num_units=2, batchnorm=False, bf16_blocks=False):
        super().__init__()
        self.unet = SparseUNet(cin, channels, num_units, batchnorm, bf16_blocks)

    # How a forward without prepared metadata builds its index structures: False = step by step from Python (row-count
    # waits overlapped with kernel queueing: faster when nothing else can run meanwhile), True = one scn_pyramid_build
    # call.  The prefetch paths always use the native call: it holds no interpreter lock, which is what lets a helper
    # thread overlap it with the main thread (measured 7.3 -> 6.6 ms/step; the Python-driven prefetch gained nothing).
    NATIVE_INDEX = False

    def forward(self, coords, feats, spatial_size, batch_size=0, metadata=None, after_encoder=None):
        """metadata: optional Metadata prepared for the same coords (`prefetch` / `prefetch_in_thread`).
        after_encoder: callable(interims) run between the encoder and the decoder -- whoever consumes the encoder outputs
        (the RPN of model.py:141-160 reads `box_feature_map_levels`) queues its kernels ahead of the decoder's; results do not
        depend on it."""
        object.__setattr__(self.unet, "_encoder_hook", after_encoder)
        if metadata is None and self.NATIVE_INDEX and coords.shape[0] > 0:
            from .metadata import Metadata
            metadata = Metadata(3).build_native(spatial_size, coords, batch_size, 4, len(self.unet.channels), 3,
                                                xcd_order=self._xcd_order())
        x = InputLayer(3, spatial_size, mode=4)((coords, feats, batch_size), metadata)
        try:
            return self.unet(x)
        finally:
            object.__setattr__(self.unet, "_encoder_hook", None)

    def _xcd_order(self):
        """bf16 storage: the SubM tiles of the pyramid also get the XCD-local hand-out order (scn_tiles_build_x)."""
        from . import metadata as MD
        return bool((self.unet.bf16_all or self.unet.bf16_blocks) and MD.XCD_ORDER_BF16) or MD.XCD_ORDER_DEFAULT

    def load_reference_state_dict(self, state_dict, prefix=None, strict=True):
        """A reference FeatureExtractor checkpoint -> this backbone (SparseUNet.load_reference_state_dict)."""
        return self.unet.load_reference_state_dict(state_dict, prefix, strict)

    def prefetch_in_thread(self, coords, spatial_size, batch_size=0):
        """As `prefetch`, on a helper thread: returns a PendingMetadata whose `.result()` is passed as `metadata=`."""
        from .metadata import Metadata
        return Metadata(3).prepare_in_thread(spatial_size, coords, batch_size, 4, len(self.unet.channels), 3,
                                             xcd_order=self._xcd_order())

    def prefetch(self, coords, spatial_size, batch_size=0):
        """Build the index structures of a coming batch on the index stream (overlaps the current batch's kernels)."""
        from .metadata import Metadata
        return Metadata(3).prepare_async(spatial_size, coords, batch_size, 4, len(self.unet.channels), 3, native=True,
                                         xcd_order=self._xcd_order())


class DropinBackbone(nn.Module):
    """The backbone driven the way the reference's own module tree drives the scn surface (the drop-in path proper):
    `CustomInputLayer` creates the Metadata inside the forward from HOST coordinates (custom_operations.py:67-83), no
    layer announces the depth of the network, rulebooks are requested one by one by the layers that need them, there is no
    helper thread, and the control flow is the reference containers': every encoder level is CALLED AS ONE scn.Sequential
    and its outputs are collected (SequentialInterims, custom_container.py:5-12), a decoder level is four calls -- input
    stage, JoinTable, NetworkInNetwork, output stage (SkipConnectionReuniter, :70-83) -- over the levels in reverse
    (ReuniteSequentialInterims, :42-55).  Shares the parameters of a `Backbone`; used by `bench.py --dropin` and the tests to
    compare the two ways of driving the same kernels.  Storage type: `scn.set_feature_storage` (the reference's factory
    knows no storage types), not the Backbone's `bf16_blocks`."""

    def __init__(self, backbone: Backbone):
        super().__init__()
        self.unet = backbone.unet

    def forward(self, coords, feats, spatial_size, batch_size=0):
        x = InputLayer(3, spatial_size, mode=4)((coords, feats, batch_size))
        u = self.unet
        interims = []
        for level in u.encoder:
            x = level(x)
            interims.append(x)
        object.__setattr__(u, "interims", interims)
        *skips, x = interims
        for d, skip in zip(u.decoder, skips[::-1]):
            x = d["units"](d["nin"](d["join"]([d["up"](x), skip])))
        return x
