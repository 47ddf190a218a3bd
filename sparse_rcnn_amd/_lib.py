"""ctypes binding of libscn_mi355x.so (C ABI declared in include/scn_mi355x.h).

The product path has no CPU fallback: if the shared library is missing, or no MI355X is visible, every
operator raises.  PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCN_MI355X_LIB") or os.path.join(_HERE, "libscn_mi355x.so")   # env: developer builds

_lib = None

p = C.c_void_p
i32, i64, f32 = C.c_int, C.c_int64, C.c_float

_SIGS = {
    "scn_abi_version": (C.c_int, []),
    "scn_last_error_string": (C.c_char_p, []),
    "scn_debug_set": (C.c_int, [C.c_char_p, C.c_char_p]),
    "scn_debug_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(i64)]),
    "scn_hash_capacity": (i64, [i64]),
    "scn_coords_to_i32": (C.c_int, [p, i64, p, p, C.POINTER(i64), p]),
    "scn_dedup_scratch_bytes": (i64, [i64]),
    "scn_dedup_build": (C.c_int, [p, i64, i32, p, p, i64, p, p, p, p, p, C.POINTER(i64), p]),
    "scn_dedup_launch": (C.c_int, [p, i64, i32, p, p, i64, p, p, p, p, p, p, p]),
    "scn_dedup_launch_div": (C.c_int, [p, i64, i32, i32, i32, p, p, i64, p, p, p, p, p, p, p]),
    "scn_child_table_div": (C.c_int, [p, p, i64, i64, i32, i32, i32, p, p, p]),
    "scn_parent_lookup_div": (C.c_int, [p, i64, i32, i32, i32, p, p, i64, p, p, p]),
    "scn_subm_table": (C.c_int, [p, i64, p, p, i64, i32, p, p]),
    "scn_child_table": (C.c_int, [p, p, i64, i64, p, p, p]),
    "scn_rules_blocks": (i64, [i32, i64]),
    "scn_rules_scan": (C.c_int, [p, i32, i64, p, p, C.POINTER(i64), p]),
    "scn_rules_fill": (C.c_int, [p, i32, i64, p, p, p, p, p]),
    "scn_pyramid_workspace_bytes": (i64, [i64, i32, i32]),
    "scn_pyramid_build": (C.c_int, [p, i64, i32, i32, p, i64, C.POINTER(i64), p]),
    "scn_pyramid_build_ex": (C.c_int, [p, i64, i32, i32, p, i64, C.POINTER(i64), i32, p]),
    "scn_roi_units": (i64, [i64]),
    "scn_roi_count": (C.c_int, [p, i64, p, i32, p, p, p, p]),
    "scn_roi_fill": (C.c_int, [p, i64, p, i32, p, p, p, p, p, p]),
    "scn_roi_inside": (C.c_int, [p, p, i64, i64, i32, p, p]),
    "scn_roi_coords": (C.c_int, [p, p, p, i64, p, p]),
    "scn_roi_boxes": (C.c_int, [p, p, i32, p, p, p, p]),
    "scn_gemm_table": (C.c_int, [p, i64, i32, p, i32, i64, p, p, p, p, p, i32, i32, p]),
    "scn_tiles_scratch_bytes": (i64, [i32, i64]),
    "scn_tiles_build": (C.c_int, [p, i32, i64, p, p, p, p, p, p]),
    "scn_tiles_build_x": (C.c_int, [p, i32, i64, p, p, p, p, i32, p, p]),
    "scn_tiles_order_ints": (i64, [i64, i32]),
    "scn_sort_pairs_scratch_bytes": (i64, [i64]),
    "scn_sort_pairs": (C.c_int, [p, p, i64, i32, p, p, p, p]),
    "scn_conv_tiles_scratch_bytes": (i64, [i32, i64, i32]),
    "scn_conv_tiles_arrival_counters": (i64, [i32, i64, i32]),
    "scn_conv_tiles": (C.c_int, [p, i64, i32, p, p, p, p, i32, i64, p, p, p, p, p, i32, i32, p, p, p]),
    "scn_conv_tiles_chain": (C.c_int, [i32, p, i64, i32, p, p, p, p, i32, i64, i32, p, p, p]),
    "scn_conv_tiles_chain_counts": (None, [C.POINTER(i64), i32]),
    "scn_wgrad_tiles32_scratch_bytes": (i64, []),
    "scn_wgrad_tiles32": (C.c_int, [p, i64, p, i64, p, p, p, p, p, p, i32, p]),
    "scn_conv_tiles_path_counts": (None, [C.POINTER(i64), i32]),
    "scn_conv_tiles_split_count": (i64, [i32]),
    "scn_conv_tiles_finish": (C.c_int, [i32, i64, p, p, p, p, i32, i32, p, p]),
    "scn_conv_tiles_bf16_image_bytes": (i64, [i32, i32, i32]),
    "scn_conv_tiles_bf16_pack": (C.c_int, [p, i32, i32, i32, i32, p, p]),
    "scn_conv_tiles_bf16_pack_many": (C.c_int, [i32, p, p, p, p, p, p, p]),
    "scn_conv_tiles_bf16_scratch_bytes": (i64, [i32, i64, i32]),
    "scn_conv_tiles_bf16_arrival_counters": (i64, [i32, i64, i32]),
    "scn_conv_tiles_bf16": (C.c_int, [p, i64, i32, p, p, p, p, i32, i64, p, p, p, p, p, i32, i32, p, p, p]),
    "scn_gemm_rules": (C.c_int, [p, i32, p, p, C.POINTER(i64), i32, p, p, p, p, i32, i32, p]),
    "scn_wgrad_scratch_bytes": (i64, [i32, i32, C.POINTER(i64), i32]),
    "scn_wgrad_rules": (C.c_int, [p, i32, p, i32, p, p, C.POINTER(i64), i32, p, p, i32, p]),
    "scn_gemm_table_bf16": (C.c_int, [p, i64, i32, p, i32, i64, p, p, p, p, p, i32, i32, p]),
    "scn_gemm_rules_bf16": (C.c_int, [p, i32, p, p, C.POINTER(i64), i32, p, p, p, p, i32, i32, p]),
    "scn_gemm_rows2": (C.c_int, [p, i32, p, i32, i64, p, p, p, p, p, i32, p, i32, i32, i32, p]),
    "scn_wgrad_rules_bf16": (C.c_int, [p, i32, p, i32, p, p, C.POINTER(i64), i32, p, p, i32, p]),
    "scn_wgrad_bias_rules": (C.c_int, [p, i32, p, i32, p, p, C.POINTER(i64), i32, p, p, C.c_uint32, p, i32, p]),
    "scn_wgrad_bias_rules_bf16": (C.c_int, [p, i32, p, i32, p, p, C.POINTER(i64), i32, p, p, C.c_uint32, p, i32, p]),
    "scn_wgrad_scratch_bytes2": (i64, [i32, i32, C.POINTER(i64), i32]),
    "scn_wgrad_defer_begin": (C.c_int, []),
    "scn_wgrad_defer_flush": (C.c_int, [p]),
    "scn_wgrad_bias_rules2": (C.c_int, [p, p, p, p, i32, i32, p, p, C.POINTER(i64), i32, p, p, C.c_uint32, p, i32, p]),
    "scn_wgrad_bias_rules2_bf16": (C.c_int, [p, p, p, p, i32, i32, p, p, C.POINTER(i64), i32, p, p, C.c_uint32, p, i32, p]),
    "scn_wgrad_scratch_bytes_n": (i64, [i32, i32, C.POINTER(i64), i32, i32]),
    "scn_wgrad_bias_rules_n": (C.c_int, [C.POINTER(p), C.POINTER(p), i32, i32, i32, p, p, C.POINTER(i64), i32, p, p, C.c_uint32, p, i32, p]),
    "scn_wgrad_bias_rules_n_bf16": (C.c_int, [C.POINTER(p), C.POINTER(p), i32, i32, i32, p, p, C.POINTER(i64), i32, p, p, C.c_uint32, p, i32, p]),
    "scn_colsum": (C.c_int, [p, i64, i32, p, p, p]),
    "scn_colsum_bf16": (C.c_int, [p, i64, i32, p, p, p]),
    "scn_relu_fwd": (C.c_int, [p, i64, p, p]),
    "scn_relu_bwd": (C.c_int, [p, p, i64, p, p]),
    "scn_add": (C.c_int, [p, p, i64, p, p]),
    "scn_bn_scratch_bytes": (i64, [i32]),
    "scn_bn_stats": (C.c_int, [p, i64, i32, p, p, p, p]),
    "scn_bn_sums": (C.c_int, [p, i64, i32, p, p, p]),
    "scn_bn_bwd_reduce": (C.c_int, [p, p, i64, i32, p, p, f32, p, p, f32, p, p, p, p, p]),
    "scn_bn_bwd_apply": (C.c_int, [p, p, i64, i32, p, p, f32, p, p, f32, p, i64, p, p]),
    "scn_bn_fwd": (C.c_int, [p, i64, i32, p, p, f32, p, p, f32, p, p]),
    "scn_bn_bwd": (C.c_int, [p, p, i64, i32, p, p, f32, p, p, f32, i32, p, p, p, p, p]),
    "scn_input_fwd": (C.c_int, [p, p, p, p, i64, i64, i32, i32, p, p, p, p]),
    "scn_input_bwd": (C.c_int, [p, p, p, p, p, i64, i32, i32, p, p]),
    "scn_gather_rows": (C.c_int, [p, p, i64, i32, p, p]),
    "scn_segment_sum": (C.c_int, [p, p, i64, i64, i32, p, p, p]),
    "scn_mask_scatter": (C.c_int, [p, i64, i32, p, p, p, i32, p, p, p]),
    "scn_mask_gather": (C.c_int, [p, i64, i32, p, p, p, p, p, p, p, p, p]),
    "scn_mask_gather_bwd": (C.c_int, [p, i64, i32, p, p, p, p, p]),
    "scn_vox_scratch_bytes": (i64, [i64]),
    "scn_vox_project": (C.c_int, [p, i64, p, p, p, p, p, p]),
    "scn_vox_discretize": (C.c_int, [p, i64, p, p, p, p, p, p]),
    "scn_vox_gather": (C.c_int, [p, p, i64, p, i64, p, p]),
    "scn_nms": (C.c_int, [p, i32, i32, f32, p, p]),
    "scn_dilate_gather_fwd": (C.c_int, [p, p, i32, C.POINTER(i64), i32, i32, p, p, p]),
    "scn_dilate_gather_bwd": (C.c_int, [p, p, i64, C.POINTER(i64), i32, i32, p, p]),
    "scn_cell_map": (C.c_int, [p, i64, i32, C.POINTER(i64), p, p, p, p]),
    "scn_nms_scratch_bytes": (i64, [i32, i32]),
    "scn_nms_bits": (C.c_int, [p, i32, i32, f32, p, p, p]),
    "scn_topk_scratch_bytes": (C.c_int64, [i32]),
    "scn_topk_boxes": (C.c_int, [p, p, i32, i64, i32, p, p, p, p, p]),
    "scn_pool_fwd": (C.c_int, [p, p, i64, i32, i32, p, p]),
    "scn_pool_bwd": (C.c_int, [p, p, p, p, i64, i32, i32, p, p]),
    "scn_segment_pool_scratch_bytes": (i64, [i32, i32]),
    "scn_sample_counts": (C.c_int, [p, i64, i32, p, p, p]),
    "scn_segment_pool_fwd": (C.c_int, [p, p, i64, i32, i32, i32, p, p, p, p, p]),
    "scn_segment_pool_bwd": (C.c_int, [p, p, p, p, i64, i32, i32, i32, p, p, p, p]),
    "scn_pad_params_many": (C.c_int, [i32, p, p, p, i32, p]),
    "scn_cast_f32_to_bf16": (C.c_int, [p, i64, p, p]),
    "scn_cast_bf16_to_f32": (C.c_int, [p, i64, p, p]),
    "scn_add_bf16": (C.c_int, [p, p, i64, p, p]),
    "scn_gather_rows_bf16": (C.c_int, [p, p, i64, i32, p, p]),
    "scn_segment_sum_bf16": (C.c_int, [p, p, i64, i64, i32, p, p, p]),
    "scn_pool_fwd_bf16": (C.c_int, [p, p, i64, i32, i32, p, p]),
    "scn_pool_bwd_bf16": (C.c_int, [p, p, p, p, i64, i32, i32, p, p]),
    "scn_sparse_to_dense_fwd_bf16": (C.c_int, [p, p, i64, i32, C.POINTER(i64), p, p]),
    "scn_sparse_to_dense_bwd_bf16": (C.c_int, [p, p, i64, i32, C.POINTER(i64), p, p]),
    "scn_exec_struct_bytes": (i64, [i32]),
    "scn_exec_requirements": (C.c_int, [p, i32, p, i32, C.POINTER(i64), C.POINTER(i64)]),
    "scn_exec_run": (C.c_int, [p, i32, p, i32, p, p, p, p, i64, p, p]),
    "scn_exec_run_streams": (C.c_int, [p, i32, p, i32, p, p, p, p, i64, p, p, p, p, i64]),
    "scn_exec_timing_enable": (C.c_int, [i32]),
    "scn_exec_timing_collect": (i64, [p, p, i64]),
    "scn_sparse_to_dense_fwd": (C.c_int, [p, p, i64, i32, C.POINTER(i64), p, p]),
    "scn_sparse_to_dense_bwd": (C.c_int, [p, p, i64, i32, C.POINTER(i64), p, p]),
}

EXPORTS = tuple(_SIGS)

F_RELU_IN, F_W_TRANSPOSED, F_OFF_REVERSE, F_RESIDUAL_LAST, F_SPLIT_SUM, F_GEMM_V1 = 1, 2, 4, 8, 16, 32
F_TILE_ORDER_X = 64
OK, EINVAL, ESIZE, EHASH, EHIP = 0, 1, 2, 3, 4
ABI_VERSION = 5                 # include/scn_mi355x.h SCN_ABI_VERSION this host layer was written against
PYRAMID_MAX_LEVELS, PYRAMID_LEVEL_STRIDE = 8, 72
PYRAMID_TWO_QUEUES = 1
PYRAMID_XCD_ORDER = 2
PYRAMID_FUSED = 4
PYRAMID_DESC_LEN = 8 + PYRAMID_MAX_LEVELS * PYRAMID_LEVEL_STRIDE
COLSUM_BLOCKS = 512


class ScnError(RuntimeError):
    pass


def load():
    """Load the shared library (no GPU needed).  Raises ScnError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ScnError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.scn_abi_version() != ABI_VERSION:
            raise ScnError("libscn_mi355x.so ABI version mismatch")
        _lib = lib
    return _lib


_gpu_ok = False


def lib():
    """Library handle for compute calls: additionally requires a visible GPU (checked once; the check costs ~2 us and
    this function runs ~150 times per step)."""
    global _gpu_ok
    if not _gpu_ok:
        l = load()
        if not torch.cuda.is_available():
            raise ScnError("sparse_rcnn_amd needs an MI355X (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
        _gpu_ok = True
        return l
    return _lib


import contextlib


@contextlib.contextmanager
def debug_switch(name: str, value):
    """A developer switch of the library (scn_debug.hip's table) set for the duration of the block -- the tests' A/B runs of
    two kernel variants inside one process.  The library reads the environment once, at the first use of any switch; after
    that only scn_debug_set changes one (no getenv per launch).  value None: unset."""
    l = load()
    was_set, was = C.c_int(0), i64(0)
    check(l.scn_debug_get(name.encode(), C.byref(was_set), C.byref(was)))
    check(l.scn_debug_set(name.encode(), None if value is None else str(value).encode()))
    try:
        yield
    finally:
        check(l.scn_debug_set(name.encode(), str(was.value).encode() if was_set.value else None))


class _Switches:
    """`switches["SCN_TS_NO_TAIL"] = "1"` / `del switches["SCN_TS_NO_TAIL"]`: scn_debug_set in mapping clothes."""

    def __setitem__(self, name, value):
        check(load().scn_debug_set(name.encode(), str(value).encode()))

    def __delitem__(self, name):
        check(load().scn_debug_set(name.encode(), None))


switches = _Switches()


def check(rc: int):
    if rc != 0:
        raise ScnError(f"libscn_mi355x error {rc}: {load().scn_last_error_string().decode()}")


_raw_stream, _get_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice


def stream() -> int:
    """Raw hipStream_t of torch's current stream.  torch.cuda.current_stream() costs ~9 us per call in Python and
    torch.cuda.current_device() ~1 us (lazy-init checks); every C call needs the handle (560 calls per config-3 step), so
    this goes to the two C bindings directly -- lib() has already made sure the GPU context exists."""
    return _raw_stream(_get_device())


_scratch = {}


def scratch(nbytes: int, device):
    """Scratch buffer for a kernel call on the CURRENT stream: one growing buffer per (device, stream).  Calls on a
    stream are ordered, and every scratch user has finished with it when its call's last kernel ends, so the next
    call on the same stream may overwrite it."""
    key = (device.index if device.index is not None else _get_device(), stream())
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _scratch[key] = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
    return buf


_arrival = {}


def arrival(n: int, device):
    """Zeroed int32 arrival counters for the in-launch K reduction of scn_conv_tiles on the CURRENT stream: one growing
    buffer per (device, stream), zeroed when it is (re)allocated -- the kernels leave it zero (include/scn_mi355x.h)."""
    key = (device.index if device.index is not None else _get_device(), stream())
    buf = _arrival.get(key)
    if buf is None or buf.numel() < n:
        buf = _arrival[key] = torch.zeros(max(int(n), 1 << 16), dtype=torch.int32, device=device)
    return buf


def ptr(t):
    return 0 if t is None else t.data_ptr()


def host_i64(n):
    return (i64 * n)()
