"""Module classes of the ``scn`` operator API the reference's ndsis.modules code calls.

Constructor signatures, positional/keyword use and parameter attribute names (``weight``, ``bias``,
``running_mean``, ``running_var``) follow the call sites in module_factory.py / model.py so that
``sys.modules['sparseconvnet'] = sparse_rcnn_amd`` is a drop-in for that path.
"""
from __future__ import annotations

import math

import torch
from torch.nn import Module, Parameter

from . import functional as F
from .tensor import DeferredTensor, JoinedTensor, SparseConvNetTensor


def _triple(v, what):
    if isinstance(v, (int,)) or (hasattr(v, "ndim") and getattr(v, "ndim") == 0):
        return (int(v),) * 3
    t = tuple(int(x) for x in v)
    if len(t) != 3:
        raise ValueError(f"{what} must have 3 entries")
    return t


def _out(input, features, spatial_size=None):
    return SparseConvNetTensor(features=features, metadata=input.metadata,
                               spatial_size=input.spatial_size if spatial_size is None else spatial_size)


# ----------------------------------------------------------------------------------------------------------------------
# Feature storage of a module tree somebody else built (round 4).  The reference's factory code (module_factory.py:127-183,
# 513-578) knows nothing about storage types; `scn.set_feature_storage(torch.bfloat16)` switches every tree built on this
# package to bf16-STORED feature slabs (BASELINE configs 3-5; parameters, their gradients and all accumulation stay fp32):
# a conv-type layer whose output width is a multiple of 8 stores its result in the storage type, and a layer the bf16 entry
# points do not serve (input width no multiple of 8 on the tile kernels, batch norm, a stand-alone ReLU) widens its input.
# With the default (torch.float32) nothing changes.
# ----------------------------------------------------------------------------------------------------------------------
FEATURE_STORAGE = torch.float32


def set_feature_storage(dtype):
    """torch.float32 (default) | torch.bfloat16.  -> the previous setting."""
    global FEATURE_STORAGE
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("feature storage: torch.float32 or torch.bfloat16")
    prev, FEATURE_STORAGE = FEATURE_STORAGE, dtype
    return prev


def _stored(y, n_out):
    """The output slab of a conv-type layer in the tree's storage type."""
    if FEATURE_STORAGE is torch.bfloat16 and y.dtype == torch.float32 and n_out % 8 == 0:
        return y.to(torch.bfloat16)
    return y


def _conv_input(x, c_in, c_out, tile_kernel):
    """bf16-stored rows reach a layer only when its bf16 entry point takes them; otherwise the layer runs on widened rows."""
    if x.dtype == torch.bfloat16 and (c_in % 8 or (tile_kernel and c_out % 8)):
        return x.float()
    return x


# ----------------------------------------------------------------------------------------------------------------------
# Step-executor stages for module trees somebody else built (executor.py).  The reference's tree is
#   encoder level   scn.Sequential(Sequential(SubM 1^3 | Convolution 2^3/2), Sequential(residual units))   module_factory.py:513-530
#   decoder level   SkipConnectionReuniter(input_stage = scn.Sequential(ReLU, Deconvolution), combiner = scn.JoinTable,
#                   channel_changer = scn.NetworkInNetwork, output_stage = scn.Sequential(residual units))  :533-578
# An encoder level IS one of this package's Sequential objects: its forward recognises the shape and runs the level as one
# executor stage (one autograd node, one C call each way).  A decoder level is four calls from a container this package
# never sees (custom_container.py:70-83), so the first and the third return a DeferredTensor and the fourth -- a Sequential
# of residual units that is handed a pending NetworkInNetwork -- runs the whole level.  Anything that does not fit computes
# layer by layer as before; both ways launch the same kernels with the same arguments (bit-identical, tests/test_gpu_exec.py).
# ----------------------------------------------------------------------------------------------------------------------
TREE_STAGES = __import__('os').environ.get("SCN_TREE_STAGES", "1") != "0"
STAGE_STATS = {"enc": 0, "dec": 0, "layerwise_units": 0}       # how often a forward took which way (tests, bench.py)


def _usable(f):
    from . import executor as EX, profiling, unet
    return (TREE_STAGES and EX.ENABLED and unet.SparseUNet.EXEC and profiling.exec_ok() and f.is_cuda and f.dim() == 2
            and f.shape[0] > 0 and f.dtype in (torch.float32, torch.bfloat16))


def _head_of(m):
    """The conv-type head of an encoder level: the layer itself or a Sequential holding just it."""
    if type(m) is Sequential and len(m._modules) == 1:
        m = next(iter(m._modules.values()))
    if type(m) is SubmanifoldConvolution and m.filter_size == 1 and m.groups == 1:
        return m
    return m if (type(m) is Convolution and m.stride == (2, 2, 2)) else None


# A cached shape (`_stage_kind`) / compiled plan (`_stages`) of a Sequential captures its WHOLE subtree: the head, the inner
# Sequential's residual units, their bias / padding flags (editing the INNER Sequential, or replacing a child
# through __setitem__ / __delitem__ / insert / extend / add_module, left the outer plan running the old layer list).  Every
# structural edit of any scn.Sequential and every `bias` / `pad_out_to` assignment of a conv-type layer advances the tree
# epoch; a cache from an older epoch is kept only if the subtree's signature (identity + type of every module below, the
# conv layers' shape / bias / padding) is still the one it was compiled for.  Per call: one integer comparison.
_TREE_EPOCH = [0]
_PLAN_CACHES = ("_stage_kind", "_stages", "_stage_sig", "_stage_epoch")


def _bump_tree_epoch():
    _TREE_EPOCH[0] += 1


def _signature(seq):
    sig = []

    def walk(m):
        for c in m._modules.values():
            sig.append((id(c), type(c).__name__))
            if isinstance(c, _ConvBase):
                sig.append((c.nIn, c.nOut, c.bias is None, c.pad_out_to))
            if c is not None:
                walk(c)
    walk(seq)
    return tuple(sig)


def _validate_plan_caches(seq):
    d = seq.__dict__
    if d.get("_stage_epoch") == _TREE_EPOCH[0]:
        return
    sig = _signature(seq)
    if d.get("_stage_sig") != sig:
        d.pop("_stage_kind", None)
        d.pop("_stages", None)
        d["_stage_sig"] = sig
    d["_stage_epoch"] = _TREE_EPOCH[0]


def _kind(seq):
    """'enc': (head, units) of an encoder level | 'units': residual units only | 'up': (ReLU, Deconvolution) | None."""
    _validate_plan_caches(seq)
    kind = seq.__dict__.get("_stage_kind")
    if kind is not None:
        return kind or None
    from . import executor as EX
    mods = list(seq._modules.values())
    kind = False
    if (len(mods) == 2 and type(mods[0]) is ReLU and type(mods[1]) is Deconvolution and mods[1].bias is not None
            and mods[1].stride == (2, 2, 2)):
        kind = "up"
    elif len(mods) == 2 and _head_of(mods[0]) is not None and type(mods[1]) is Sequential:
        head, blocks = _head_of(mods[0]), EX._plain_blocks(mods[1])
        if (blocks and head.bias is not None and head.pad_out_to is None and all(c1.nIn == head.nOut for c1, _ in blocks)
                and head.nOut % 8 == 0 and (type(head) is SubmanifoldConvolution or head.nIn % 8 == 0)):
            kind = "enc"
    elif mods and EX._plain_blocks(seq):
        kind = "units"
    object.__setattr__(seq, "_stage_kind", kind)
    return kind or None


def _enc_stage(seq, input):
    """Encoder level through the executor, or NotImplemented."""
    from . import executor as EX
    f = input.features
    if not _usable(f):
        return NotImplemented
    mods = list(seq._modules.values())
    head = _head_of(mods[0])
    level = 0 if type(head) is SubmanifoldConvolution else 1
    if f.shape[1] != head.nIn:
        return NotImplemented
    in16 = f.dtype == torch.bfloat16
    if level == 0:
        bf16 = in16 or FEATURE_STORAGE is torch.bfloat16
        if in16 and head.nIn % 8:
            return NotImplemented
    else:
        bf16 = in16
        if FEATURE_STORAGE is torch.bfloat16 and not in16:
            # an fp32 slab reaching a strided head under bf16 storage: layer by layer the head's result is stored in bf16 and
            # the units run on bf16 rows; the stage would keep the whole level in fp32 -> the layer-by-layer path
            return NotImplemented
    key = (bf16, in16)
    cache = seq.__dict__.setdefault("_stages", {})
    st = cache.get(key)
    if st is None:
        st = cache[key] = EX.compile_encoder_stage(level, head, EX._plain_blocks(mods[1]), head.nIn, bf16, in_bf16=in16)
    md = input.metadata
    lv = EX.build_levels(md, input.spatial_size, level + 1)
    if lv is None:
        return NotImplemented
    y = EX.run_stage(st, lv, [f], pack=True)
    STAGE_STATS["enc"] += 1
    size = input.spatial_size if level == 0 else torch.as_tensor([int(s) // 2 for s in input.spatial_size], dtype=torch.long)
    return SparseConvNetTensor(features=y, metadata=md, spatial_size=size)


def _dec_stage(seq, input):
    """Residual units handed a pending NetworkInNetwork(JoinTable([Deconvolution(ReLU(x)), skip])): the decoder level through
    the executor, or NotImplemented (the DeferredTensor then computes layer by layer)."""
    from . import executor as EX
    tag = input.tag
    if tag is None or tag[0] != "nin":
        return NotImplemented
    _, up_t, skip_t, nin = tag
    if not (isinstance(up_t, DeferredTensor) and up_t.pending and up_t.tag[0] == "up"):
        return NotImplemented
    _, coarse_t, deconv = up_t.tag
    xc, xs = coarse_t.features, skip_t.features
    c = deconv.nOut
    blocks = EX._plain_blocks(seq)
    if not (_usable(xc) and _usable(xs) and xc.dtype == xs.dtype and xs.shape[1] == c and nin.nIn == 2 * c and nin.nOut == c
            and nin.bias is not None and c % 8 == 0 and deconv.nIn % 8 == 0 and xc.shape[1] == deconv.nIn
            and deconv.pad_out_to is None and nin.pad_out_to is None and all(c1.nIn == c for c1, _ in blocks)):
        return NotImplemented
    md = input.metadata
    fine = tuple(int(s) for s in input.spatial_size)
    if md.cached_strided_rulebook(fine) is None or skip_t.metadata is not md:
        return NotImplemented                            # (the layer-by-layer path raises the reference's error)
    bf16 = xc.dtype == torch.bfloat16
    cache = seq.__dict__.setdefault("_stages", {})
    key = (id(deconv), id(nin), bf16)
    st = cache.get(key)
    if st is None:
        st = EX.compile_decoder_stage(0, deconv, nin, blocks, deconv.nIn, bf16)
        cache[key] = st
        st._keep = (deconv, nin)                         # (the ids in the key stay valid while the plan lives)
    lv = EX.build_levels(md, fine, 2)
    if lv is None:
        return NotImplemented
    y = EX.run_stage(st, lv, [xc, xs], pack=True)
    STAGE_STATS["dec"] += 1
    return SparseConvNetTensor(features=y, metadata=md, spatial_size=input.spatial_size)


class Sequential(torch.nn.Sequential):
    """``scn.Sequential(*modules)`` with ``.append`` / ``.add`` (module_factory.py:52-56,421-424).
    Peephole: ``ReLU`` directly followed by a conv-type layer runs as one kernel (ReLU fused into the gather)."""

    def append(self, module):
        self.add_module(str(len(self._modules)), module)
        return self

    def add(self, module):
        return self.append(module)

    # ---- every structural edit advances the tree epoch (cached stage plans of ANY enclosing Sequential are re-validated)
    def add_module(self, name, module):
        _bump_tree_epoch()
        return super().add_module(name, module)

    register_module = add_module

    def __setattr__(self, name, value):                  # (__setitem__ of torch.nn.Sequential lands here)
        _bump_tree_epoch()
        return super().__setattr__(name, value)

    def __delattr__(self, name):                         # (__delitem__ / pop)
        _bump_tree_epoch()
        return super().__delattr__(name)

    def insert(self, index, module):
        _bump_tree_epoch()
        return super().insert(index, module)

    def __delitem__(self, idx):
        _bump_tree_epoch()
        return super().__delitem__(idx)

    def __getstate__(self):
        """copy.deepcopy / torch.save of a tree that has run: the compiled plans hold ctypes pointer arrays (not picklable)
        and belong to THIS object's modules -- a copy compiles its own on first use."""
        d = self.__dict__.copy()
        for k in _PLAN_CACHES:
            d.pop(k, None)
        return d

    def forward(self, input, residual=None):
        """residual: features to add to the output of the LAST module when that is a SubmanifoldConvolution (used by
        the residual-block peephole below)."""
        if residual is None and TREE_STAGES and isinstance(input, SparseConvNetTensor):
            kind = _kind(self)
            if kind == "enc":
                y = _enc_stage(self, input)
                if y is not NotImplemented:
                    return y
            elif kind == "units" and isinstance(input, DeferredTensor) and input.pending:
                y = _dec_stage(self, input)
                if y is not NotImplemented:
                    return y
                STAGE_STATS["layerwise_units"] += 1
            elif kind == "up" and _usable(input.features):
                deconv = self._modules["1"]
                out_size = torch.as_tensor([int(s) * 2 for s in input.spatial_size], dtype=torch.long)
                return DeferredTensor(lambda: deconv(input, relu_in=True).features, input.metadata, out_size,
                                      ("up", input, deconv))
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            # residual block  Sequential(ConcatTable(Identity, inner), AddTable): the add runs in inner's last conv
            if (FUSE_ADD and type(m) is ConcatTable and i + 1 < len(mods) and type(mods[i + 1]) is AddTable
                    and len(m._modules) == 2):
                a, inner = list(m._modules.values())
                if type(a) is Identity and type(inner) is Sequential and inner._is_plain_residual_branch(input):
                    c1, c2 = inner._modules["1"], inner._modules["3"]          # ReLU, SubM3, ReLU, SubM3
                    fn = (F.ResidualBlockFunctionBF16 if input.features.dtype == torch.bfloat16
                          else F.ResidualBlockFunction)                        # bf16-stored features: CastFeatures
                    w1, b1 = c1._wb(input.features.shape[1])
                    w2, b2 = c2._wb(w1.shape[-1])
                    y = fn.apply(input.features, w1, b1, w2, b2, input.metadata, input.spatial_size)
                    input = _out(input, y)
                    i += 2
                    continue
                if type(a) is Identity and type(inner) is Sequential and inner._ends_with_subm():
                    input = inner(input, residual=input.features)
                    i += 2
                    continue
            last = residual is not None and (i == len(mods) - 1 or (i == len(mods) - 2 and type(m) is ReLU))
            if (FUSE_RELU and type(m) is ReLU and i + 1 < len(mods)
                    and isinstance(mods[i + 1], (SubmanifoldConvolution, Convolution, Deconvolution))):
                nxt = mods[i + 1]
                if last and isinstance(nxt, SubmanifoldConvolution):
                    input = nxt(input, relu_in=True, residual=residual)
                else:
                    input = nxt(input, relu_in=True)
                i += 2
            else:
                if last and isinstance(m, SubmanifoldConvolution):
                    input = m(input, residual=residual)
                else:
                    input = m(input)
                i += 1
        return input

    def _is_plain_residual_branch(self, input):
        """ReLU, SubM 3^3, ReLU, SubM 3^3 with the block's channel count kept (the reference's residual unit,
        module_factory.py:127-183 with relu_first=True, two convolutions): runs as one fused autograd node."""
        mods = list(self._modules.values())
        return (FUSE_BLOCK and len(mods) == 4 and list(self._modules.keys()) == ["0", "1", "2", "3"]
                and type(mods[0]) is ReLU and type(mods[2]) is ReLU
                and type(mods[1]) is SubmanifoldConvolution and type(mods[3]) is SubmanifoldConvolution
                and mods[1].filter_size == 3 and mods[3].filter_size == 3
                and mods[1].nIn == mods[3].nOut and mods[1].nOut == mods[3].nIn
                and (mods[3].pad_out_to or mods[3].nOut) == input.features.shape[1])

    def _ends_with_subm(self):
        mods = list(self._modules.values())
        return bool(mods) and isinstance(mods[-1], SubmanifoldConvolution)


import os as _os

FUSE_RELU = True
FUSE_ADD = _os.environ.get("SCN_FUSE_ADD", "1") != "0"     # developer switch (tools/ab_bench.py)
FUSE_BLOCK = _os.environ.get("SCN_FUSE_BLOCK", "1") != "0"


class CastFeatures(Module):
    """Storage dtype of the feature slab (not a reference module): `CastFeatures(torch.bfloat16)` in front of a run of
    residual units and `CastFeatures(torch.float32)` behind it puts those units on the bf16 storage path
    (ResidualBlockFunctionBF16); every other layer takes fp32 features."""

    def __init__(self, dtype):
        super().__init__()
        self.dtype = dtype

    def forward(self, input):
        return _out(input, input.features.to(self.dtype))


class ConcatTable(Sequential):
    """Applies every child to the same input -> list (module_factory.py:52-54)."""

    def forward(self, input):
        return [m(input) for m in self._modules.values()]


class AddTable(Module):
    def forward(self, input):
        feats = _wide(input[0].features)
        for t in input[1:]:
            feats = F.AddFunction.apply(feats, _wide(t.features))
        return _out(input[0], feats)


class JoinTable(Module):
    """Channel concat of tensors sharing Metadata / row order (module_factory.py:301)."""

    def forward(self, input):
        return JoinedTensor(None, input[0].metadata, input[0].spatial_size, sources=input)


class Identity(Module):
    def forward(self, input):
        return input


def _wide(x):
    """Layers without a bf16 entry point (stand-alone ReLU, AddTable outside a residual unit, batch norm) widen bf16-stored
    rows; the next conv-type layer stores its result in the tree's storage type again."""
    return x.float() if x.dtype == torch.bfloat16 else x


class ReLU(Module):
    def forward(self, input):
        return _out(input, F.ReLUFunction.apply(_wide(input.features)))


class _BatchNorm(Module):
    # SYNC (class-wide switch, default off): in a data-parallel step with one scene per rank, take the batch statistics
    # over the scenes of all ranks -- what the reference's single-process batch does (module_factory.py:92-102); costs
    # one small all-reduce per layer and direction.  Without it each rank normalises with its own scene's statistics.
    SYNC = False

    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, affine=True, leakiness=0.0):
        super().__init__()
        self.nPlanes, self.eps, self.momentum, self.leakiness = nPlanes, eps, momentum, leakiness
        self.register_buffer("running_mean", torch.zeros(nPlanes))
        self.register_buffer("running_var", torch.ones(nPlanes))
        self.weight = Parameter(torch.ones(nPlanes))
        self.bias = Parameter(torch.zeros(nPlanes))

    def forward(self, input):
        y = F.BatchNormReLUFunction.apply(_wide(input.features), self.weight, self.bias, self.running_mean, self.running_var,
                                          float(self.eps), float(self.momentum), float(self.leakiness), self.training,
                                          _BatchNorm.SYNC)
        return _out(input, y)

    def extra_repr(self):
        return f"{self.nPlanes}, eps={self.eps}, momentum={self.momentum}, leakiness={self.leakiness}"


class BatchNormReLU(_BatchNorm):
    """``scn.BatchNormReLU(nPlanes, eps, momentum)`` (module_factory.py:101-102)."""

    def __init__(self, nPlanes, eps=1e-4, momentum=0.9):
        super().__init__(nPlanes, eps, momentum, True, 0.0)


class BatchNormLeakyReLU(_BatchNorm):
    """``scn.BatchNormLeakyReLU(nPlanes, eps, momentum, leakiness)`` (module_factory.py:98-99)."""

    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, leakiness=0.333):
        super().__init__(nPlanes, eps, momentum, True, leakiness)


class _ConvBase(Module):
    # Channel padding (not a reference feature): a feature slab may be PHYSICALLY wider than the layer's nIn / nOut, the
    # extra columns being zero (sparse_rcnn_amd.unet pads the mask head's 23-channel level to 24 so that its rows are
    # 16-byte aligned and its layers take the vector kernels).  The parameters keep their logical shape (state_dict
    # compatibility); `_wb` hands the kernels a zero-padded view through torch's differentiable pad.
    pad_out_to = None

    def __setattr__(self, name, value):
        if name == "bias" or name == "pad_out_to":       # what a compiled stage plan of an enclosing Sequential depends on
            _bump_tree_epoch()
        return super().__setattr__(name, value)

    def _wb(self, cin_phys):
        """(weight, bias) as the kernels see them for an input slab of cin_phys columns."""
        W, b = self.weight, self.bias
        pin = int(cin_phys) - self.nIn
        pout = (self.pad_out_to - self.nOut) if self.pad_out_to else 0
        if pin < 0 or pout < 0:
            raise ValueError(f"{type(self).__name__}: {cin_phys} input columns for nIn={self.nIn}")
        if pin or pout:
            Wp = F.PADDED.get((id(self), int(cin_phys), "w"))        # (functional.padded_params: one launch for a network)
            if Wp is not None:
                return Wp, F.PADDED.get((id(self), int(cin_phys), "b"), b)
            if F.PAD_RECORD is not None:
                F.PAD_RECORD.append((self, int(cin_phys)))
            W = torch.nn.functional.pad(W, (0, pout, 0, pin))
            if b is not None and pout:
                b = torch.nn.functional.pad(b, (0, pout))
        return W, b

    def _pad_jobs(self, cin_phys):
        """functional.PadPlan jobs for the tensors `_wb(cin_phys)` pads."""
        pout = (self.pad_out_to - self.nOut) if self.pad_out_to else 0
        fv = self.weight.shape[0]
        jobs = [((id(self), int(cin_phys), "w"), self, "weight", (fv, int(cin_phys), self.nOut + pout), None)]
        if self.bias is not None and pout:
            jobs.append(((id(self), int(cin_phys), "b"), self, "bias", (self.nOut + pout,), None))
        return jobs

    def _init(self, filter_volume, nIn, nOut, bias):
        self.nIn, self.nOut = int(nIn), int(nOut)
        std = math.sqrt(2.0 / self.nIn / filter_volume)
        self.weight = Parameter(torch.empty(filter_volume, self.nIn, self.nOut).normal_(0, std))
        if bias:
            self.bias = Parameter(torch.zeros(self.nOut))
        else:
            self.register_parameter("bias", None)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # checkpoints written with SparseConvNet's grouped layout [fv, groups=1, nIn, nOut] load into [fv, nIn, nOut]
        k = prefix + "weight"
        if k in state_dict and state_dict[k].dim() == 4 and state_dict[k].shape[1] == 1 and self.weight.dim() == 3:
            state_dict[k] = state_dict[k].squeeze(1)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class SubmanifoldConvolution(_ConvBase):
    """``scn.SubmanifoldConvolution(dimension, nIn, nOut, filter_size, bias, groups=1)`` (module_factory.py:383-385,404-406)."""

    def __init__(self, dimension, nIn, nOut, filter_size, bias, groups=1):
        super().__init__()
        if int(dimension) != 3:
            raise NotImplementedError("only dimension 3")
        fs = _triple(filter_size, "filter_size")
        if len(set(fs)) != 1 or fs[0] % 2 == 0:
            raise ValueError("SubmanifoldConvolution needs an odd cubic filter")
        self.dimension, self.filter_size = 3, fs[0]
        self.groups = int(groups)
        if self.groups < 1 or int(nIn) % self.groups or int(nOut) % self.groups:
            raise ValueError(f"SubmanifoldConvolution: groups={groups} must divide nIn={nIn} and nOut={nOut}")
        self._init(self.filter_size ** 3, nIn, nOut, bias)
        if self.groups > 1:
            # SparseConvNet's grouped layout [filter volume, groups, nIn / groups, nOut / groups]: group g maps input channels
            # [g nIn/G, (g+1) nIn/G) to output channels [g nOut/G, (g+1) nOut/G).  No reference configuration reaches it
            # (module_factory.py:145,152 pin groups = 1 in the residual blocks; :596 only in the dense dilation network), so the
            # kernels see the block-diagonal dense weight -- the products with its zero blocks add exact zeros.
            fv, G = self.filter_size ** 3, self.groups
            std = math.sqrt(2.0 / (self.nIn // G) / fv)
            self.weight = Parameter(torch.empty(fv, G, self.nIn // G, self.nOut // G).normal_(0, std))

    def _wb(self, cin_phys):
        if self.groups == 1:
            return super()._wb(cin_phys)
        if int(cin_phys) != self.nIn or self.pad_out_to:
            raise ValueError("SubmanifoldConvolution: channel padding is not combined with groups")
        G = self.groups
        eye = torch.eye(G, dtype=self.weight.dtype, device=self.weight.device)
        W = torch.einsum("ogij,gh->ogihj", self.weight, eye).reshape(self.weight.shape[0], self.nIn, self.nOut)
        return W, self.bias

    def forward(self, input, relu_in=False, residual=None):
        W, b = self._wb(input.features.shape[1])          # (physical widths: a channel-padded slab is wider than nIn)
        x = _conv_input(input.features, W.shape[1], W.shape[2], self.filter_size == 3)
        if residual is not None and residual.dtype != x.dtype:
            residual = residual.to(x.dtype)
        y = F.SubmanifoldConvolutionFunction.apply(x, W, b, input.metadata, input.spatial_size, self.filter_size, relu_in,
                                                   residual)
        return _out(input, _stored(y, W.shape[-1]))

    def extra_repr(self):
        return f"{self.nIn}->{self.nOut} C{self.filter_size}" + (f" groups={self.groups}" if self.groups > 1 else "")


def _size_stride(filter_size, filter_stride):
    """filter_size == filter_stride, one positive entry per axis (`get_downsampler` / `get_upsampler`, module_factory.py:221-258,
    hand `stride_tuple` to both arguments; every shipped configuration uses 2)."""
    fs, st = _triple(filter_size, "filter_size"), _triple(filter_stride, "filter_stride")
    if fs != st or any(v < 1 for v in st):
        raise NotImplementedError(
            f"only filter_size == filter_stride (the reference's down/up-samplers, module_factory.py:221-258); "
            f"got size {fs} stride {st}")
    return st


class Convolution(_ConvBase):
    """``scn.Convolution(dimension, nIn, nOut, filter_size, filter_stride, bias)`` (module_factory.py:232-234)."""

    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias):
        super().__init__()
        if int(dimension) != 3:
            raise NotImplementedError("only dimension 3")
        self.stride = _size_stride(filter_size, filter_stride)
        self._init(self.stride[0] * self.stride[1] * self.stride[2], nIn, nOut, bias)

    def forward(self, input, relu_in=False):
        in_size = tuple(int(s) for s in input.spatial_size)
        W, b = self._wb(input.features.shape[1])
        x = _conv_input(input.features, W.shape[1], W.shape[2], W.shape[0] <= 27)
        y = F.ConvolutionFunction.apply(x, W, b, input.metadata, in_size, relu_in, self.stride)
        out_size = torch.as_tensor([s // st for s, st in zip(in_size, self.stride)], dtype=torch.long)
        return _out(input, _stored(y, W.shape[-1]), out_size)

    def input_spatial_size(self, out_size):
        return out_size * torch.as_tensor(self.stride, dtype=torch.long)

    def extra_repr(self):
        st = self.stride
        return f"{self.nIn}->{self.nOut} C{st[0]}/{st[0]}" if len(set(st)) == 1 else f"{self.nIn}->{self.nOut} C{st}/{st}"


class Deconvolution(_ConvBase):
    """``scn.Deconvolution(dimension, nIn, nOut, filter_size, filter_stride, bias)`` (module_factory.py:256-258)."""

    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias):
        super().__init__()
        if int(dimension) != 3:
            raise NotImplementedError("only dimension 3")
        self.stride = _size_stride(filter_size, filter_stride)
        self._init(self.stride[0] * self.stride[1] * self.stride[2], nIn, nOut, bias)

    def forward(self, input, relu_in=False):
        out_size = tuple(int(s) * st for s, st in zip(input.spatial_size, self.stride))
        W, b = self._wb(input.features.shape[1])
        x = _conv_input(input.features, W.shape[1], W.shape[2], W.shape[0] <= 27)
        y = F.DeconvolutionFunction.apply(x, W, b, input.metadata, out_size, relu_in, self.stride)
        return _out(input, _stored(y, W.shape[-1]), torch.as_tensor(out_size, dtype=torch.long))

    def extra_repr(self):
        st = self.stride
        return f"{self.nIn}->{self.nOut} D{st[0]}/{st[0]}" if len(set(st)) == 1 else f"{self.nIn}->{self.nOut} D{st}/{st}"


class NetworkInNetwork(Module):
    """``scn.NetworkInNetwork(nIn, nOut, bias)`` (module_factory.py:366-367)."""

    def __init__(self, nIn, nOut, bias):
        super().__init__()
        self.nIn, self.nOut = int(nIn), int(nOut)
        self.weight = Parameter(torch.empty(self.nIn, self.nOut).normal_(0, math.sqrt(2.0 / self.nIn)))
        if bias:
            self.bias = Parameter(torch.zeros(self.nOut))
        else:
            self.register_parameter("bias", None)

    # channel padding (see _ConvBase): in_groups = logical widths of the joined inputs, each physically padded to the same
    # width (JoinTable of two padded slabs) or to in_phys[i]; pad_out_to = physical output width
    pad_out_to = None
    in_groups = None
    in_phys = None

    def _layout(self, cin_phys):
        groups = self.in_groups or (self.nIn,)
        phys = self.in_phys or (cin_phys // len(groups),) * len(groups)
        if sum(phys) != cin_phys or sum(groups) != self.nIn or any(p < g for p, g in zip(phys, groups)):
            raise ValueError(f"NetworkInNetwork: {cin_phys} input columns for nIn={self.nIn}")
        return groups, phys

    def _pad_jobs(self, cin_phys):
        """functional.PadPlan jobs for the tensors `_wb(cin_phys)` pads (at most two joined parts)."""
        groups, phys = self._layout(cin_phys)
        if len(groups) > 2:
            return []
        pout = (self.pad_out_to - self.nOut) if self.pad_out_to else 0
        segs, r0, d0 = [], 0, 0
        for gsz, gp in zip(groups, phys):
            segs.append((r0, gsz, d0))
            r0, d0 = r0 + gsz, d0 + gp
        jobs = [((id(self), int(cin_phys), "w"), self, "weight", (int(cin_phys), self.nOut + pout), segs)]
        if self.bias is not None and pout:
            jobs.append(((id(self), int(cin_phys), "b"), self, "bias", (self.nOut + pout,), None))
        return jobs

    def _wb(self, cin_phys):
        W, b = self.weight, self.bias
        pad = torch.nn.functional.pad
        if cin_phys != self.nIn or self.pad_out_to:
            Wp = F.PADDED.get((id(self), int(cin_phys), "w"))
            if Wp is not None:
                return Wp, F.PADDED.get((id(self), int(cin_phys), "b"), b)
            if F.PAD_RECORD is not None and (cin_phys != self.nIn or self.pad_out_to != self.nOut):
                F.PAD_RECORD.append((self, int(cin_phys)))
        if cin_phys != self.nIn:
            groups, phys = self._layout(cin_phys)
            parts, r0 = [], 0
            for gsz, gp in zip(groups, phys):
                parts.append(pad(W[r0:r0 + gsz], (0, 0, 0, gp - gsz)))
                r0 += gsz
            W = torch.cat(parts, 0)
        pout = (self.pad_out_to - self.nOut) if self.pad_out_to else 0
        if pout:
            W = pad(W, (0, pout))
            b = pad(b, (0, pout)) if b is not None else None
        return W, b

    def forward(self, input):
        if (TREE_STAGES and isinstance(input, JoinedTensor) and not input.materialized and input.sources is not None
                and len(input.sources) == 2 and isinstance(input.sources[0], DeferredTensor) and input.sources[0].pending
                and input.sources[0].tag[0] == "up"):
            # JoinTable([Deconvolution(ReLU(x)), skip]) -> NetworkInNetwork with the deconvolution still pending: stay pending;
            # the residual units behind this layer run the decoder level as one executor stage (_dec_stage), anybody else
            # gets the layer-by-layer result
            joined = input
            return DeferredTensor(lambda: self.forward(_materialized_join(joined)).features, input.metadata, input.spatial_size,
                                  ("nin", input.sources[0], input.sources[1], self))
        if isinstance(input, JoinedTensor) and not input.materialized and input.n_parts > 1:
            # JoinTable -> NetworkInNetwork (module_factory.py:557-563): one row GEMM per joined part against its rows of
            # the weight, accumulated through the kernel's residual operand -- the concatenated slab is never built
            parts = input.parts
            if len({p.dtype for p in parts}) > 1 or any(p.dtype == torch.bfloat16 and p.shape[1] % 8 for p in parts):
                parts = [p.float() for p in parts]
            W, b = self._wb(sum(p.shape[1] for p in parts))
            y = F.JoinedNetworkInNetworkFunction.apply(W, b, *parts)
            return SparseConvNetTensor(features=_stored(y, W.shape[-1]), metadata=input.metadata, spatial_size=input.spatial_size)
        W, b = self._wb(input.features.shape[1])
        x = _conv_input(input.features, W.shape[0], W.shape[1], False)
        return _out(input, _stored(F.NetworkInNetworkFunction.apply(x, W, b), W.shape[-1]))

    def extra_repr(self):
        return f"{self.nIn}->{self.nOut}"


def _materialized_join(joined):
    """The JoinedTensor with its sources computed (a pending Deconvolution runs now) -- for the layer-by-layer path."""
    joined.parts = [t.features for t in joined.sources]
    joined.sources = None
    return joined


class SparseToDense(Module):
    """``scn.SparseToDense(dimension, nPlanes)`` (module_factory.py:429-435) -> dense [B, C, X, Y, Z]."""

    def __init__(self, dimension, nPlanes):
        super().__init__()
        self.dimension, self.nPlanes = dimension, nPlanes

    def forward(self, input):
        return F.SparseToDenseFunction.apply(input.features, input.metadata, input.spatial_size)


class _Pooling(Module):
    AVERAGE = False

    def __init__(self, dimension, pool_size, pool_stride, nFeaturesToDrop=0):
        super().__init__()
        if int(dimension) != 3:
            raise NotImplementedError("only dimension 3")
        ps, st = _triple(pool_size, "pool_size"), _triple(pool_stride, "pool_stride")
        if ps != st or any(v < 1 for v in st) or nFeaturesToDrop:
            raise NotImplementedError("only pool_size == pool_stride (the reference's down-poolings, "
                                      "module_factory.py:315-354) and nFeaturesToDrop = 0")
        self.stride = st

    def forward(self, input):
        in_size = tuple(int(s) for s in input.spatial_size)
        y = F.PoolingFunction.apply(input.features, input.metadata, in_size, self.AVERAGE, self.stride)
        return _out(input, y, torch.as_tensor([s // v for s, v in zip(in_size, self.stride)], dtype=torch.long))


class MaxPooling(_Pooling):
    """``scn.MaxPooling(dimension, pool_size, pool_stride)`` (module_factory.py:326-327): max(0, active children)."""


class AveragePooling(_Pooling):
    """``scn.AveragePooling(dimension, pool_size, pool_stride)`` (module_factory.py:347-348): sum(active children) / 8."""
    AVERAGE = True
