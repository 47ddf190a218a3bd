"""sparse_rcnn_amd -- MI355X-native drop-in for the ``sparseconvnet`` (scn) operator API that the reference's
ScanNet instance-segmentation code calls (ndsis/modules/{module_factory,model,custom_operations,roi_select_sparse}.py).

    import sys, sparse_rcnn_amd
    sys.modules["sparseconvnet"] = sparse_rcnn_amd      # then `import sparseconvnet as scn` resolves here

Host code is Python/PyTorch-ROCm (device memory, streams, autograd, torch.distributed); all arithmetic runs in
hand-written gfx950 HIP kernels behind the C ABI of include/scn_mi355x.h (libscn_mi355x.so).  No CPU fallback.
"""
from . import ioLayers                                                     # noqa: F401  (scn.ioLayers.*Function)
from ._lib import ScnError, EXPORTS, LIB_PATH, load as load_library          # noqa: F401
from .ioLayers import InputLayer, OutputLayer                              # noqa: F401
from .metadata import Metadata, index_prefetching, prefetch_index           # noqa: F401
from .modules import (AddTable, AveragePooling, BatchNormLeakyReLU, BatchNormReLU, CastFeatures, ConcatTable,  # noqa: F401
                      Convolution, Deconvolution, Identity, JoinTable, MaxPooling, NetworkInNetwork, ReLU,
                      Sequential, SparseToDense, SubmanifoldConvolution, set_feature_storage)
from .tensor import SparseConvNetTensor                                    # noqa: F401
from .custom_operations import SparseGlobalPool, split_batch               # noqa: F401  (device forms of the reference's own helpers)

__all__ = [
    "Metadata", "SparseConvNetTensor", "ioLayers", "InputLayer", "OutputLayer", "Sequential", "ConcatTable",
    "AddTable", "JoinTable", "Identity", "ReLU", "BatchNormReLU", "BatchNormLeakyReLU", "Convolution",
    "Deconvolution", "SubmanifoldConvolution", "NetworkInNetwork", "MaxPooling", "AveragePooling", "SparseToDense",
    "prefetch_index", "index_prefetching", "SparseGlobalPool", "split_batch", "set_feature_storage",
]
