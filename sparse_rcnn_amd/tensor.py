"""``scn.SparseConvNetTensor`` (custom_operations.py:20-21,81-83; roi_select_sparse.py:83-84,103,109)."""
from __future__ import annotations

import torch


class SparseConvNetTensor:
    def __init__(self, features=None, metadata=None, spatial_size=None):
        self.features = features
        self.metadata = metadata
        self.spatial_size = spatial_size

    def _size(self):
        return tuple(int(s) for s in self.spatial_size)

    def get_spatial_locations(self, spatial_size=None):
        """int64 CPU [N, 4] = (x, y, z, batch) in row order; batch column non-decreasing when the input was
        sample-major (roi_select_sparse.py:103-107)."""
        size = self._size() if spatial_size is None else tuple(int(s) for s in spatial_size)
        return self.metadata.get_spatial_locations(size)

    def batch_size(self):
        return self.metadata.n_samples

    def getSpatialLocations(self, spatial_size=None):      # upstream alias
        return self.get_spatial_locations(spatial_size)

    def to(self, device):
        self.features = self.features.to(device)
        return self

    def cuda(self):
        self.features = self.features.cuda()
        return self

    def type(self, t=None):
        if t is None:
            return self.features.type()
        self.features = self.features.type(t)
        return self

    def detach(self):
        return SparseConvNetTensor(self.features.detach(), self.metadata, self.spatial_size)

    def __repr__(self):
        n = None if self.features is None else tuple(self.features.shape)
        return f"SparseConvNetTensor<features={n}, spatial_size={None if self.spatial_size is None else self._size()}>"


class DeferredTensor(SparseConvNetTensor):
    """A SparseConvNetTensor whose features have NOT been computed yet (round 4: the fast path for module trees somebody else
    built).  The reference's decoder level is four calls into this package from a container of its own
    (custom_container.py:70-83: input_stage -> combiner -> channel_changer -> output_stage); the step executor wants them as
    ONE launch plan.  So `Sequential(ReLU, Deconvolution)` and `NetworkInNetwork` over a JoinTable return a DeferredTensor
    that remembers what to compute (`tag`), and the `Sequential` of residual units that receives it runs the whole level as
    one executor stage.  Anybody else who touches `.features` gets them computed layer by layer, exactly as before
    (`thunk`): deferral never changes a result, only who launches the kernels."""

    def __init__(self, thunk, metadata, spatial_size, tag):
        self._thunk, self._feat = thunk, None
        self.metadata = metadata
        self.spatial_size = spatial_size
        self.tag = tag

    @property
    def pending(self):
        return self._feat is None

    @property
    def features(self):
        if self._feat is None:
            self._feat = self._thunk()
            self._thunk = self.tag = None
        return self._feat

    @features.setter
    def features(self, value):
        self._feat, self._thunk, self.tag = value, None, None

    def __repr__(self):
        if self._feat is None:
            return f"DeferredTensor<pending {self.tag[0]}, spatial_size={None if self.spatial_size is None else self._size()}>"
        return super().__repr__()


class JoinedTensor(SparseConvNetTensor):
    """What ``scn.JoinTable`` returns: the channel concatenation of tensors that share Metadata and row order
    (module_factory.py:298-301), kept as its PARTS.  ``.features`` is the concatenated slab, built on first access -- a
    consumer that can read the parts as separate row sources (NetworkInNetwork: one GEMM per part against the matching
    rows of its weight) never asks for it, and the copy (two slabs written and read again per decoder level) does not
    happen.  `sources`: the joined SparseConvNetTensor objects themselves (a DeferredTensor among them stays pending until
    somebody asks for `parts` / `features`)."""

    def __init__(self, parts=None, metadata=None, spatial_size=None, sources=None):
        self.sources = list(sources) if sources is not None else None
        self._parts = list(parts) if parts is not None else None
        self._cat = None
        self.metadata = metadata
        self.spatial_size = spatial_size

    @property
    def parts(self):
        if self._parts is None:
            self._parts = [t.features for t in self.sources]
        return self._parts

    @parts.setter
    def parts(self, value):
        self._parts = list(value)

    @property
    def n_parts(self):
        return len(self._parts) if self._parts is not None else len(self.sources)

    @property
    def features(self):
        if self._cat is None:
            self._cat = torch.cat(self.parts, 1)
        return self._cat

    @features.setter
    def features(self, value):
        self._cat = value
        self._parts = [value]
        self.sources = None

    @property
    def materialized(self):
        return self._cat is not None
