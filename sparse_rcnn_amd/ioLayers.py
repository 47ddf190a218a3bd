"""``scn.ioLayers`` -- the submodule attribute the reference reaches into
(custom_operations.py:8,17,72; roi_select_sparse.py:79,117)."""
import torch
from torch.nn import Module

from .functional import InputLayerFunction, OutputLayerFunction
from .metadata import Metadata, take_prefetched
from .tensor import SparseConvNetTensor


class InputLayer(Module):
    """``scn.InputLayer(dimension, spatial_size, mode=3)``: input = (coords LongTensor [N, dim+1], features[, batch_size])."""

    def __init__(self, dimension, spatial_size, mode=3):
        super().__init__()
        self.dimension = dimension
        self.spatial_size = torch.as_tensor(spatial_size, dtype=torch.long).reshape(-1)
        if self.spatial_size.numel() == 1:
            self.spatial_size = self.spatial_size.repeat(dimension)
        self.mode = mode

    def forward(self, input, metadata=None):
        coords, features = input[0], input[1]
        batch_size = input[2] if len(input) == 3 else 0
        coords = coords.long()
        md = metadata
        if md is None:
            # announced by scn.prefetch_index for this very tensor object AND for this layer's spatial size, mode and
            # batch_size (a mode-0 layer must keep its "unique coordinates" check, n_samples must be the layer's)
            md = take_prefetched(coords, self.spatial_size.tolist(), batch_size, self.mode)
        if md is None:
            md = Metadata(self.dimension)
        feats = InputLayerFunction.apply(self.dimension, md, self.spatial_size, coords, features, batch_size, self.mode)
        return SparseConvNetTensor(features=feats, metadata=md, spatial_size=self.spatial_size)


class OutputLayer(Module):
    """``scn.OutputLayer(dimension)`` (model.py:461,576,600,643,658)."""

    def __init__(self, dimension):
        super().__init__()
        self.dimension = dimension

    def forward(self, input):
        return OutputLayerFunction.apply(self.dimension, input.metadata, input.features)

    def __repr__(self):
        return "OutputLayer()"
