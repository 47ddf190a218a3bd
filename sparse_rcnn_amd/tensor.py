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


class JoinedTensor(SparseConvNetTensor):
    """What ``scn.JoinTable`` returns: the channel concatenation of tensors that share Metadata and row order
    (module_factory.py:298-301), kept as its PARTS.  ``.features`` is the concatenated slab, built on first access -- a
    consumer that can read the parts as separate row sources (NetworkInNetwork: one GEMM per part against the matching
    rows of its weight) never asks for it, and the copy (two slabs written and read again per decoder level) does not
    happen."""

    def __init__(self, parts, metadata=None, spatial_size=None):
        self.parts = list(parts)
        self._cat = None
        self.metadata = metadata
        self.spatial_size = spatial_size

    @property
    def features(self):
        if self._cat is None:
            self._cat = torch.cat(self.parts, 1)
        return self._cat

    @features.setter
    def features(self, value):
        self._cat = value
        self.parts = [value]

    @property
    def materialized(self):
        return self._cat is not None
