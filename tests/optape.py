"""TEST INFRASTRUCTURE (never imported by the product): the op tape of a forward pass over the scn operator surface.

VERDICT r3 item 5b: `unet.DropinBackbone` / `maskhead.MaskBranch` are this repository's imitation of how the reference's
`FeatureExtractor.forward` (ndsis/modules/model.py:414-446) and `SparseMaskNetwork.forward` (:758-782) drive the surface.
The reference may not travel to the GPU box and this package has no CPU path, so the two never ran side by side.  What CAN
be pinned is the call pattern:

  * `record()` notes every LEAF operator call of a forward pass -- the conv-type modules, ReLU, AddTable, JoinTable,
    SparseToDense and the two ioLayers functions -- with its layer signature, where its operands came from (index of the
    producing tape entry), the spatial size and the row / channel counts of its result;
  * `shape_stub()` replaces the arithmetic of the surface by SHAPES ONLY (zero tensors on the CPU; row counts from a numpy
    dedup of the coordinates), so that the REFERENCE's forward can run on this package in the build container
    (tests/golden/make_optape_golden.py -> tests/golden/optape_*.json);
  * on the GPU, tests/test_gpu_exec.py records the tape of DropinBackbone / MaskBranch with every fusion switched off (one
    leaf call per layer: modules.FUSE_*, TREE_STAGES, SparseUNet.EXEC) on the same seeded scene and compares the two tapes
    entry by entry.  The fused paths are pinned to the unfused one bit for bit by the other tests of that file.
"""
from __future__ import annotations

import contextlib

import numpy as np
import torch

import sparse_rcnn_amd as scn
from sparse_rcnn_amd import ioLayers, modules as M, roi
from sparse_rcnn_amd.tensor import JoinedTensor, SparseConvNetTensor

LEAVES = (M.SubmanifoldConvolution, M.Convolution, M.Deconvolution, M.NetworkInNetwork, M.ReLU, M.AddTable, M.JoinTable,
          M._BatchNorm, M.SparseToDense, M._Pooling)


def _sig(m):
    return f"{type(m).__name__}({m.extra_repr()})" if hasattr(m, "extra_repr") else type(m).__name__


class Tape:
    def __init__(self):
        self.entries = []
        self._producer = {}            # id(object) -> tape index
        self._keep = []                # ids stay unique while the objects live

    # ---- producers -----------------------------------------------------------------------------------------------
    def _ids(self, v):
        out = [id(v)]
        if type(v) is SparseConvNetTensor and v.features is not None:
            out.append(id(v.features))
        return out

    def source(self, v):
        """Tape index of the entry that produced value v | "ext" (came from outside the surface) | list for a list."""
        if isinstance(v, (list, tuple)):
            return [self.source(x) for x in v]
        if isinstance(v, JoinedTensor) and v.sources is not None and id(v) not in self._producer:
            return [self.source(x) for x in v.sources]
        for i in self._ids(v):
            if i in self._producer:
                return self._producer[i]
        return "ext"

    def produced(self, v, index):
        if isinstance(v, (list, tuple)):
            return
        self._keep.append(v)
        for i in self._ids(v):
            self._producer.setdefault(i, index)          # (Identity hands its input on: the first producer stays)
        if type(v) is SparseConvNetTensor and v.features is not None:
            self._keep.append(v.features)

    def add(self, **e):
        self.entries.append(e)
        return len(self.entries) - 1


def _shape_of(v):
    if isinstance(v, SparseConvNetTensor):
        if isinstance(v, JoinedTensor):
            parts = v.parts if v.sources is None else [s.features for s in v.sources]
            return int(parts[0].shape[0]), int(sum(p.shape[1] for p in parts)), [int(s) for s in v.spatial_size]
        f = v.features
        return int(f.shape[0]), int(f.shape[1]), [int(s) for s in v.spatial_size]
    if torch.is_tensor(v):
        return int(v.shape[0]), int(v.shape[1]), None
    return None, None, None


@contextlib.contextmanager
def record():
    """with record() as tape: run a forward.  tape.entries afterwards: one dict per leaf operator call, in call order."""
    tape = Tape()
    hooks = []

    def post(module, args, out):
        if not isinstance(module, LEAVES):
            return
        rows, ch, size = _shape_of(out)
        if isinstance(module, M._ConvBase) or isinstance(module, M.NetworkInNetwork):
            ch = module.nOut                             # the LOGICAL width (a channel-padded slab is wider)
        idx = tape.add(op=type(module).__name__, sig=_sig(module), src=tape.source(args[0]), rows=rows, ch=ch, size=size)
        tape.produced(out, idx)
    hooks.append(torch.nn.modules.module.register_module_forward_hook(post))

    real_in, real_out = ioLayers.InputLayerFunction, ioLayers.OutputLayerFunction

    class InRec:
        @staticmethod
        def apply(dimension, metadata, spatial_size, coords, features, batch_size, mode):
            y = real_in.apply(dimension, metadata, spatial_size, coords, features, batch_size, mode)
            idx = tape.add(op="InputLayerFunction", mode=int(mode), batch_size=int(batch_size), src=tape.source(features),
                           size=[int(s) for s in spatial_size], n_in=int(coords.shape[0]), rows=int(y.shape[0]),
                           ch=int(y.shape[1]))
            tape.produced(y, idx)
            return y

    class OutRec:
        @staticmethod
        def apply(dimension, metadata, features):
            y = real_out.apply(dimension, metadata, features)
            idx = tape.add(op="OutputLayerFunction", src=tape.source(features), rows=int(y.shape[0]), ch=int(y.shape[1]))
            tape.produced(y, idx)
            return y

    ioLayers.InputLayerFunction, ioLayers.OutputLayerFunction, roi.InputLayerFunction = InRec, OutRec, InRec
    try:
        yield tape
    finally:
        ioLayers.InputLayerFunction, ioLayers.OutputLayerFunction, roi.InputLayerFunction = real_in, real_out, real_in
        for h in hooks:
            h.remove()


# ----------------------------------------------------------------------------------------------------------------------
# shapes-only stand-in of the surface (build container, CPU)
# ----------------------------------------------------------------------------------------------------------------------
class StubMetadata:
    """Row counts of every level from a numpy dedup of the coordinates; nothing else."""

    def __init__(self, dimension=3):
        self.dimension = dimension
        self.input_size = None
        self.levels = {}
        self.n_samples = 0
        self.n_items = 0
        self._convolved = set()

    def set_points(self, spatial_size, coords, batch_size):
        c = np.asarray(coords, dtype=np.int64)
        self.input_size = tuple(int(s) for s in spatial_size)
        self.n_items = len(c)
        self.n_samples = int(batch_size) if batch_size else (int(c[:, 3].max()) + 1 if len(c) else 0)
        size, l = self.input_size, 0
        while True:
            cl = np.concatenate([c[:, :3] >> l, c[:, 3:]], 1)
            self.levels[size] = len(np.unique(cl, axis=0)) if len(cl) else 0
            if any(s % 2 for s in size) or min(size) < 2:
                break
            size, l = tuple(s // 2 for s in size), l + 1

    def rows(self, size):
        return self.levels[tuple(int(s) for s in size)]


def _zeros(n, c):
    return torch.zeros((int(n), int(c)))


@contextlib.contextmanager
def shape_stub():
    """Inside the block the scn surface computes shapes only, on the CPU."""
    saved = []

    def patch(obj, name, value):
        saved.append((obj, name, getattr(obj, name)))
        setattr(obj, name, value)

    def out(input, feats, size=None):
        return SparseConvNetTensor(features=feats, metadata=input.metadata,
                                   spatial_size=input.spatial_size if size is None else torch.as_tensor(size, dtype=torch.long))

    def subm(self, input, relu_in=False, residual=None):
        return out(input, _zeros(input.features.shape[0], self.nOut))

    def conv(self, input, relu_in=False):
        size = [int(s) // st for s, st in zip(input.spatial_size, self.stride)]
        input.metadata._convolved.add(tuple(int(s) for s in input.spatial_size))
        return out(input, _zeros(input.metadata.rows(size), self.nOut), size)

    def deconv(self, input, relu_in=False):
        size = [int(s) * st for s, st in zip(input.spatial_size, self.stride)]
        if tuple(size) not in input.metadata._convolved:
            raise scn.ScnError("Deconvolution: no cached Convolution rulebook")
        return out(input, _zeros(input.metadata.rows(size), self.nOut), size)

    def nin(self, input):
        return out(input, _zeros(input.features.shape[0], self.nOut))

    def same(self, input):
        return out(input, torch.zeros_like(input.features))

    def add(self, input):
        return out(input[0], torch.zeros_like(input[0].features))

    def join(self, input):
        return JoinedTensor([t.features for t in input], input[0].metadata, input[0].spatial_size)

    def plain_sequential(self, input, residual=None):
        for m in self._modules.values():
            input = m(input)
        return input

    class InStub:
        @staticmethod
        def apply(dimension, metadata, spatial_size, coords, features, batch_size, mode):
            metadata.set_points(spatial_size, coords.numpy(), batch_size)
            return _zeros(metadata.rows(metadata.input_size), features.shape[1])

    class OutStub:
        @staticmethod
        def apply(dimension, metadata, features):
            return _zeros(metadata.n_items, features.shape[1])

    patch(M.SubmanifoldConvolution, "forward", subm)
    patch(M.Convolution, "forward", conv)
    patch(M.Deconvolution, "forward", deconv)
    patch(M.NetworkInNetwork, "forward", nin)
    patch(M.ReLU, "forward", same)
    patch(M._BatchNorm, "forward", same)
    patch(M.AddTable, "forward", add)
    patch(M.JoinTable, "forward", join)
    patch(M.Sequential, "forward", plain_sequential)
    patch(scn, "Metadata", StubMetadata)
    patch(ioLayers, "Metadata", StubMetadata)                # (scn.InputLayer creates its Metadata through this name)
    patch(ioLayers, "InputLayerFunction", InStub)
    patch(ioLayers, "OutputLayerFunction", OutStub)
    patch(SparseConvNetTensor, "batch_size", lambda self: self.metadata.n_samples)
    try:
        yield
    finally:
        for obj, name, value in reversed(saved):
            setattr(obj, name, value)


def normalised(entries, pad=None, join_pad=None, at_size=None):
    """Tape entries without what legitimately differs between the two sides: `pad` maps a physical channel count of the
    channel-padded mask U-Net level onto the logical one ({24: 23}); `join_pad` the same for the JoinTable entries at spatial
    size `at_size` (two padded slabs side by side: {48: 46})."""
    pad, join_pad = pad or {}, join_pad or {}
    out = []
    for e in entries:
        e = dict(e)
        if e["op"] == "JoinTable" and e.get("size") == at_size and e.get("ch") in join_pad:
            e["ch"] = join_pad[e["ch"]]
        elif e.get("ch") in pad:
            e["ch"] = pad[e["ch"]]
        out.append(e)
    return out
