"""Device forms of the reference's own helpers around the scn surface (ndsis/modules/custom_operations.py:24-59):
``split_batch`` and ``SparseGlobalPool``.  The reference builds a ``[samples, rows]`` bool mask on the host from
``get_spatial_locations()`` (a D2H copy of the coordinates per call) and boolean-indexes the slab once per sample; here the
sample of a row is read from the grid's int32 coordinates in HBM by one streaming kernel (csrc/scn_segpool.hip).

    from sparse_rcnn_amd.custom_operations import SparseGlobalPool, split_batch     # same names, same signatures
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib as L

_OPS = {torch.mean: 0, torch.sum: 1, torch.amax: 2}


def _grid_of(sparse_tensor):
    md = sparse_tensor.metadata
    return md.grid(tuple(int(s) for s in sparse_tensor.spatial_size))


class _SegmentPoolFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, coords_i32, n_samples, op):
        lib = L.lib()
        if features.dtype != torch.float32 or not features.is_cuda:
            raise L.ScnError("SparseGlobalPool pools fp32 features on the MI355X (no CPU fallback)")
        X = features.contiguous()
        n, c = X.shape
        dev = X.device
        Y = torch.empty((n_samples, c), dtype=torch.float32, device=dev)
        cnt = torch.empty(max(n_samples, 1), dtype=torch.int32, device=dev)
        flag = torch.empty(1, dtype=torch.int32, device=dev)
        scratch = torch.empty(lib.scn_segment_pool_scratch_bytes(n_samples, c), dtype=torch.uint8, device=dev)
        L.check(lib.scn_segment_pool_fwd(L.ptr(X), L.ptr(coords_i32), n, c, n_samples, op, L.ptr(Y), L.ptr(cnt), L.ptr(flag),
                                         L.ptr(scratch), L.stream()))
        ctx.save_for_backward(X, Y, coords_i32, cnt)
        ctx.cfg = (n_samples, op)
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = L.lib()
        X, Y, coords_i32, cnt = ctx.saved_tensors
        n_samples, op = ctx.cfg
        n, c = X.shape
        dY = dY.contiguous().float()
        dX = torch.empty_like(X)
        scratch = torch.empty(lib.scn_segment_pool_scratch_bytes(n_samples, c), dtype=torch.uint8, device=X.device)
        L.check(lib.scn_segment_pool_bwd(L.ptr(X), L.ptr(Y), L.ptr(dY), L.ptr(coords_i32), n, c, n_samples, op, L.ptr(cnt),
                                         L.ptr(dX), L.ptr(scratch), L.stream()))
        return dX, None, None, None


def split_batch(sparse_tensor):
    """custom_operations.py:24-39: list (one entry per sample) of the feature rows of that sample, in row order.  Rows
    grouped by sample (the InputLayer's order for the reference's sample-major batches, data.py:95-98) come back as row
    RANGES of the slab -- VIEWS, no copy (the reference's boolean indexing returns copies: a caller that edits a piece in
    place edits the slab here; every use in the reference only reads them, model.py:507-512); one D2H of the per-sample row
    counts.  Any other order: one device-side row selection per sample (copies; still no host coordinates)."""
    lib = L.lib()
    n_samples = int(sparse_tensor.batch_size())
    feats = sparse_tensor.features
    if n_samples == 0:
        return []
    g = _grid_of(sparse_tensor)
    dev = feats.device
    cnt = torch.empty(n_samples + 1, dtype=torch.int32, device=dev)       # [counts..., unsorted flag]
    L.check(lib.scn_sample_counts(L.ptr(g.coords), g.n, n_samples, L.ptr(cnt), cnt.data_ptr() + 4 * n_samples, L.stream()))
    host = cnt.cpu().tolist()
    counts, unsorted = host[:n_samples], host[n_samples]
    if not unsorted:
        out, r0 = [], 0
        for k in counts:
            out.append(feats[r0:r0 + k])
            r0 += k
        return out
    b = g.coords[:, 3]
    return [feats[(b == i).nonzero(as_tuple=True)[0]] for i in range(n_samples)]


class SparseGlobalPool(nn.Module):
    """``SparseGlobalPool(pooling_function=torch.mean)`` (custom_operations.py:42-59): [samples, C] = pooling_function over
    the rows of each sample, zeros for a sample without rows.  torch.mean / torch.sum / torch.amax run as one device pass
    with a matching backward; any other callable falls back to the reference's formulation over `split_batch` (whose
    pieces are views of the slab), so every function the reference accepts is still accepted."""

    def __init__(self, pooling_function=torch.mean):
        super().__init__()
        self.pooling_function = pooling_function

    def forward(self, sparse_tensor):
        n_samples = int(sparse_tensor.batch_size())
        feats = sparse_tensor.features
        if n_samples == 0:
            return feats[:0]
        op = _OPS.get(self.pooling_function)
        if op is None:
            parts = split_batch(sparse_tensor)
            return torch.stack([self.pooling_function(f, dim=0) if len(f) else f.new_zeros((f.shape[1])) for f in parts])
        if feats.dtype == torch.bfloat16:
            # bf16-STORED features (scn.set_feature_storage / bf16 networks): pooled in fp32 on exactly widened values -- the
            # reference's formulation accepts any floating dtype; the pooled [samples, C] result is fp32
            feats = feats.float()
        return _SegmentPoolFunction.apply(feats, _grid_of(sparse_tensor).coords, n_samples, op)
