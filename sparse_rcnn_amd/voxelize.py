"""Voxelisation front-end on the device.

Mirrors the deterministic core of ndsis/data/sparse_augmentation.py ``augment_coords`` (:81-126) with ``fix_cut_out``
(:42-47) / ``random_cut_out`` (:50-78) and the batch column of ``collate_fn`` (ndsis/data/data.py:95-98).  The reference
draws three random objects -- the distortion matrix (``get_coord_distortion_matrix``), the sub-pixel offset and, for the
random cut-out, the start positions; they are inputs here (draw them with torch exactly as the reference does), the rest
runs on the MI355X: no host coordinates, no H2D copy per step.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .metadata import compact_rules


def _i32x3(v):
    a = (C.c_int32 * 3)()
    for k in range(3):
        a[k] = int(v[k])
    return a


def _f32xn(v, n):
    a = (C.c_float * n)()
    flat = [float(x) for x in torch.as_tensor(v, dtype=torch.float32).reshape(-1).tolist()]
    for k in range(n):
        a[k] = flat[k]
    return a


def augment_coords(coords, *, rot_and_scale, sub_pixel_offset, spatial_size=None, shift=None, start_positions=None,
                   batch_index=0):
    """coords fp32 [N,3] (device).  rot_and_scale = almost_orthonormal * scale (sparse_augmentation.py:93).

    spatial_size and shift given      -> fix_cut_out (rows whose UNMOVED voxel lies in [0, size) are kept, moved by shift);
    spatial_size and start_positions  -> the cut-out of random_cut_out for the drawn start positions;
    spatial_size None                 -> no cut-out; spatial_size = max voxel (+ 2 shift), coords moved by shift.

    Returns (coords_batch_rows int64 device [M,4] = (x, y, z, batch_index), is_inside bool device [N],
    spatial_size int64 CPU [3], complete_shift fp32 CPU [3]) -- the first three columns / the other values are what the
    reference's augment_coords returns (:121-126)."""
    lib = L.lib()
    P = coords.to(torch.float32).contiguous()
    if not P.is_cuda or P.dim() != 2 or P.shape[1] != 3:
        raise L.ScnError("coords must be a device tensor [N, 3]")
    n, dev = P.shape[0], P.device
    if n == 0:
        raise L.ScnError("augment_coords needs at least one point (the reference takes min over the points)")
    aug = torch.empty((n, 3), dtype=torch.float32, device=dev)
    shift_max = torch.empty(6, dtype=torch.float32, device=dev)
    scratch = L.scratch(lib.scn_vox_scratch_bytes(n), dev)
    L.check(lib.scn_vox_project(L.ptr(P), n, _f32xn(rot_and_scale, 9), _f32xn(sub_pixel_offset, 3), L.ptr(aug),
                                L.ptr(shift_max), L.ptr(scratch), L.stream()))
    discrete = torch.empty((n, 3), dtype=torch.int32, device=dev)
    table = torch.empty((1, n), dtype=torch.int32, device=dev)
    if spatial_size is not None:
        size = [int(s) for s in torch.as_tensor(spatial_size).expand(3).tolist()]
        if shift is not None:                         # fix_cut_out: start = -shift, the test uses the unmoved voxel
            sh = [int(s) for s in torch.as_tensor(shift).expand(3).tolist()]
            start, test = [-s for s in sh], [0, 0, 0]
        elif start_positions is not None:
            start = [int(s) for s in torch.as_tensor(start_positions).expand(3).tolist()]
            test = start
        else:
            raise NotImplementedError("random_cut_out draws its start positions from torch's RNG: pass start_positions")
        L.check(lib.scn_vox_discretize(L.ptr(aug), n, L.ptr(shift_max), _i32x3(test), _i32x3(size), L.ptr(discrete),
                                       L.ptr(table), L.stream()))
    else:
        sh = [int(s) for s in torch.as_tensor(0 if shift is None else shift).expand(3).tolist()]
        start = [-s for s in sh]
        L.check(lib.scn_vox_discretize(L.ptr(aug), n, L.ptr(shift_max), None, None, L.ptr(discrete), L.ptr(table),
                                       L.stream()))
    rules = compact_rules(table, 1, n)
    rows = rules.in_rows                               # ascending kept rows (one host wait for their number)
    m = rows.shape[0]
    out = torch.empty((m, 4), dtype=torch.int64, device=dev)
    L.check(lib.scn_vox_gather(L.ptr(discrete), L.ptr(rows), m, _i32x3(start), int(batch_index), L.ptr(out), L.stream()))
    host = shift_max.cpu()
    complete_shift = host[:3] - torch.tensor([float(s) for s in start])        # complete_shift -= start (:112)
    if spatial_size is not None:
        size_out = torch.tensor(size, dtype=torch.int64)
    else:                                              # discrete.max(0) = trunc(max(aug) + shift): trunc is monotonic
        size_out = (host[3:] + host[:3]).to(torch.int64) + 2 * torch.tensor(sh, dtype=torch.int64)
    return out, table[0] >= 0, size_out, complete_shift


def collate_coords(rows_list):
    """coords_batch of collate_fn (data.py:95-98) from per-sample `augment_coords(..., batch_index=i)` outputs; stays on
    the device (Metadata.set_input takes device coordinates)."""
    return torch.cat(list(rows_list)) if rows_list else torch.zeros((0, 4), dtype=torch.int64)
