"""``scn.Metadata`` -- per-forward container of voxel grids and rulebooks, resident in HBM.

Mirrors the object the reference creates once per InputLayer call and shares between every tensor and layer
derived from it (ndsis/modules/custom_operations.py:15,70; roi_select_sparse.py:77,115).
Upstream keeps host hash maps; here everything lives on the device:

  grid of a spatial size  : int32 coords [N,4] + open-addressing hash (uint64 keys, int32 rows)
  SubM rulebook (size, k) : neighbour table int32 [k^3, N]   (hot kernels read this, output-stationary)
                            + compacted rules (in_rows, out_rows, prefix) in canonical order (weight grads, parity)
  strided rulebook (size) : parent / child table / compacted rules, shared by Convolution and Deconvolution

Buffers are torch tensors (caching allocator, stream ordered); the C library only fills them.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch

from . import _lib as L


def _empty(n, dtype, dev):
    return torch.empty(int(n), dtype=dtype, device=dev)


_index_streams = {}
_INDEX_PRIORITY = int(__import__("os").environ.get("SCN_INDEX_PRIORITY", "-1"))      # developer switch (A/B): -1 high, 0 normal


def index_stream(device) -> "torch.cuda.Stream":
    st = _index_streams.get(device)
    if st is None:
        # high priority: the index kernels are tiny and latency-bound; next to the matrix kernels of another batch they
        # should be dispatched as soon as a slot frees up
        st = _index_streams[device] = torch.cuda.Stream(device=device, priority=_INDEX_PRIORITY)
    return st


_index_pool = None


class PendingMetadata:
    """A Metadata being built by the index helper thread (Metadata.prepare_in_thread).  One persistent worker: builds
    are short (~1.5 ms) and at most one is in flight per training loop."""

    def __init__(self, fn):
        global _index_pool
        if _index_pool is None:
            from concurrent.futures import ThreadPoolExecutor
            _index_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="scn-index")
        self._future = _index_pool.submit(fn)

    def result(self) -> "Metadata":
        return self._future.result()             # re-raises what the build raised


# ---- index prefetch for the drop-in path -----------------------------------------------------------------------------
# The reference's CustomInputLayer (custom_operations.py:67-83) builds `scn.InputLayer(...)([coords, features, batch_size])`
# inside the forward: the layer has no say in when the index structures are built.  They depend on the coordinates only, so
# whoever holds the NEXT batch can have them built while the current one runs: `prefetch_index(coords, ...)` starts the
# build on the helper thread / index stream and parks it under the identity of the coords tensor; the InputLayer that later
# receives that very tensor adopts it (ioLayers.InputLayer.forward).  `index_prefetching(loader, extract)` is the one-line
# form for a training loop.
# Entries are keyed by the identity of the coords tensor OBJECT and hold a strong reference to it: an announced batch that
# is never run (an early `break`, an exception in the loop) cannot have its storage freed and recycled for another
# same-shaped tensor that would then match a stale entry.  The contract is "the same tensor object, unmodified": an in-place
# refill that does not go through torch (numpy views of a recycled loader buffer) is invisible to `_version` and is the
# caller's to avoid.
class _Announced:
    __slots__ = ("coords", "version", "size", "batch_size", "mode", "pending")

    def __init__(self, coords, size, batch_size, mode, pending):
        self.coords, self.version, self.size = coords, coords._version, size
        self.batch_size, self.mode, self.pending = int(batch_size), int(mode), pending


_prefetched: "Dict[int, _Announced]" = {}
_PREFETCH_KEEP = 4


def prefetch_index(coords: torch.Tensor, spatial_size, batch_size: int = 0, mode: int = 4, n_levels: int = 0, k: int = 3):
    """Start building the index structures of a COMING batch (InputLayer rules + the rulebook pyramid as deep as the last
    network over this spatial size went, Metadata.LEVELS_HINT).  coords: the int64 [N, 4] tensor the forward will hand to
    scn.InputLayer -- the same tensor object, unmodified; mode / batch_size: those of the InputLayer that will consume it
    (the reference's backbone: mode 4, custom_operations.py:67-83) -- an InputLayer with other settings builds its own."""
    size = tuple(int(s) for s in torch.as_tensor(spatial_size).reshape(-1).tolist())
    levels = n_levels or Metadata.LEVELS_HINT.get(size, 0)
    while len(_prefetched) >= _PREFETCH_KEEP:                 # batches that were announced and never run
        _prefetched.pop(next(iter(_prefetched)))
    pending = Metadata(len(size)).prepare_in_thread(spatial_size, coords, int(batch_size), mode, levels, k)
    _prefetched[id(coords)] = _Announced(coords, size, batch_size, mode, pending)
    return pending


def take_prefetched(coords: torch.Tensor, spatial_size=None, batch_size=None, mode=None):
    """The Metadata prefetched for exactly this coords tensor object (unmodified since the announcement) and -- when given --
    for this spatial size / batch_size / mode, or None (the caller then builds its own)."""
    if not _prefetched:
        return None
    e = _prefetched.get(id(coords))
    if e is None or e.coords is not coords:
        return None
    del _prefetched[id(coords)]
    if e.version != coords._version:
        return None                                           # modified in place since it was announced
    if spatial_size is not None and e.size != tuple(int(s) for s in spatial_size):
        return None
    if (batch_size is not None and int(batch_size) != e.batch_size) or (mode is not None and int(mode) != e.mode):
        return None
    return e.pending.result()


def drop_prefetched(coords=None):
    """Forget announced batches (all, or the one announced for `coords`)."""
    if coords is None:
        _prefetched.clear()
    else:
        _prefetched.pop(id(coords), None)


def index_prefetching(batches, extract):
    """Wrap a data loader: yields its batches unchanged, and before yielding batch i announces batch i+1's coordinates
    (extract(batch) -> (coords, spatial_size, batch_size)) with prefetch_index, so that their index build overlaps batch
    i's kernels.      for batch in scn.index_prefetching(loader, lambda b: (b[0], b[2], b[3])): ...
    When the loop ends early (break, exception) the batch that was announced and never run is forgotten."""
    it = iter(batches)
    try:
        cur = next(it)
    except StopIteration:
        return
    announced = None
    try:
        for nxt in it:
            args = extract(nxt)
            announced = args[0]
            prefetch_index(*args)
            yield cur
            cur = nxt
        yield cur
    finally:
        if announced is not None:
            drop_prefetched(announced)


def _ws_view(ws, off, shape, dtype):
    """A view of `shape` / `dtype` at byte offset `off` of the uint8 workspace tensor `ws` (one tensor op)."""
    es = _ELEM[dtype]
    return torch.empty(0, dtype=dtype, device=ws.device).set_(ws.untyped_storage(), (ws.storage_offset() + off) // es, shape)


_ELEM = {torch.int32: 4, torch.int64: 8, torch.uint8: 1, torch.float32: 4}


class _Views:
    """Base of the index-structure records below.  Their tensors are either given (step-by-step build) or views of ONE build
    workspace (scn_pyramid_build) that are only CREATED WHEN SOMEBODY ASKS for them: a four-level build has ~80 such views and
    creating them eagerly cost ~0.3 ms of host time per build with the device idle behind the build's size read-back; the
    step executor never needs the tensors at all -- `ptr(name)` gives the address from the workspace layout."""

    def _lazy(self, ws, ws_ptr, specs):
        """specs: attribute name -> (byte offset into ws, shape, dtype)."""
        d = self.__dict__
        d["_ws"], d["_ws_ptr"], d["_specs"] = ws, ws_ptr, specs
        return self

    def __getattr__(self, name):                       # (only reached when the attribute does not exist yet)
        d = self.__dict__
        specs = d.get("_specs")
        if specs is None or name not in specs:
            raise AttributeError(f"{type(self).__name__} has no attribute {name!r}")
        off, shape, dtype = specs[name]
        t = d[name] = _ws_view(d["_ws"], off, shape, dtype)
        return t

    def ptr(self, name) -> int:
        """Device address of tensor attribute `name` (0 for None) without creating a view."""
        d = self.__dict__
        t = d.get(name)
        if t is None:
            specs = d.get("_specs")
            if specs is not None and name in specs:
                return d["_ws_ptr"] + specs[name][0]
            t = getattr(self, name)
        return 0 if t is None else t.data_ptr()


@dataclass
class Grid(_Views):
    coords: torch.Tensor            # int32 [N,4] device
    table_keys: torch.Tensor        # int64 view of uint64 keys [cap]
    table_rows: torch.Tensor        # int32 [cap]
    cap: int
    n: int


class _Readback:
    """A small device -> pinned-host copy queued on the current stream; `get()` waits for it (and only for it)."""

    def __init__(self, *dev_tensors):
        self.host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in dev_tensors]
        for h, t in zip(self.host, dev_tensors):
            h.copy_(t, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def get(self):
        self.event.synchronize()
        return self.host


class RuleCount:
    """Total rule count of a Rules, readable without keeping its device buffers alive (profiling closures)."""

    def __init__(self, readback, n_off):
        self._rb, self._n_off, self._total = readback, n_off, None

    @property
    def total(self):
        if self._total is None:
            self._total = int(self._rb.get()[0][self._n_off])
        return self._total


class _KnownCount:
    def __init__(self, total):
        self.total = total


class Rules:
    """Compacted rule list, offset-major, output row ascending inside an offset (canonical order).

    Built in two phases (scn_rules_scan -> sizes -> scn_rules_fill).  Only the weight-gradient kernels and the
    rule-list GEMMs (Deconvolution forward, Convolution backward-data) read compacted rules; the forward hot kernel
    walks tiles of the table.  So the scan is queued when the rulebook is built, its sizes travel back through a
    pinned buffer behind an event, and the fill runs on first access -- by then (decoder / backward pass) the event
    has long fired and the host never waits for the device."""

    def __init__(self, table, n_off, n_out, want_seg=False):
        lib = L.lib()
        dev = table.device
        self.n_off = int(n_off)
        self._table, self._n_out, self._want_seg = table, int(n_out), want_seg
        self._block_sums = _empty(lib.scn_rules_blocks(n_off, n_out), torch.int32, dev)
        self.prefix_dev = _empty(n_off + 1, torch.int64, dev)
        L.check(lib.scn_rules_scan(L.ptr(table), n_off, n_out, L.ptr(self._block_sums), L.ptr(self.prefix_dev), None,
                                   L.stream()))
        self._rb = _Readback(self.prefix_dev)
        self.count = RuleCount(self._rb, self.n_off)
        self._prefix_host = None
        self._in = self._out = self._seg = None

    @classmethod
    def from_scan(cls, table, n_off, n_out, block_sums, prefix_dev, prefix_values, in_rows=None, out_rows=None,
                  want_seg=False):
        """A Rules whose scan (and fill) has already run (scn_pyramid_build)."""
        self = cls.__new__(cls)
        self.n_off = int(n_off)
        self._table, self._n_out, self._want_seg = table, int(n_out), want_seg
        self._block_sums, self.prefix_dev = block_sums, prefix_dev
        self._rb = None
        arr = (C.c_int64 * (self.n_off + 1))(*[int(v) for v in prefix_values])
        self._prefix_host = arr
        self.count = _KnownCount(int(arr[self.n_off]))
        self._in, self._out, self._seg = in_rows, out_rows, None
        return self

    @classmethod
    def from_workspace(cls, ws, ws_ptr, n_off, n_out, prefix_values, specs):
        """As from_scan, the tensors being views of the build workspace `ws` that are created on first access (specs:
        `_table`, `_block_sums`, `prefix_dev`, `_in`, `_out` -> (byte offset, shape, dtype); _Views)."""
        self = cls.__new__(cls)
        d = self.__dict__
        d["n_off"], d["_n_out"], d["_want_seg"], d["_rb"], d["_seg"] = int(n_off), int(n_out), False, None, None
        arr = (C.c_int64 * (int(n_off) + 1))(*prefix_values)
        d["_prefix_host"] = arr
        d["count"] = _KnownCount(int(arr[int(n_off)]))
        d["_ws"], d["_ws_ptr"], d["_specs"] = ws, ws_ptr, specs
        return self

    __getattr__ = _Views.__getattr__

    def ptr(self, name) -> int:
        """Device address of `in_rows` / `out_rows` (filled on first use) or of a raw field, without creating a view."""
        if name in ("in_rows", "out_rows"):
            raw = "_in" if name == "in_rows" else "_out"
            if self.__dict__.get(raw) is None and "_specs" not in self.__dict__:
                self._fill()
            name = raw
        return _Views.ptr(self, name)

    @property
    def prefix_host(self):
        """int64[n_off+1] on the host, usable as the `prefix_host` argument of the C calls."""
        if self._prefix_host is None:
            self._pinned = self._rb.get()[0]
            self._prefix_host = C.cast(self._pinned.data_ptr(), C.POINTER(C.c_int64))
        return self._prefix_host

    @property
    def total(self):
        return int(self.prefix_host[self.n_off])

    def prefix_list(self):
        ph = self.prefix_host
        return [int(ph[i]) for i in range(self.n_off + 1)]

    def _fill(self):
        if self._in is None:
            lib = L.lib()
            total = self.total
            dev = self._table.device
            self._in = _empty(total, torch.int32, dev)
            self._out = _empty(total, torch.int32, dev)
            self._seg = _empty(total, torch.int32, dev) if self._want_seg else None
            if total:                                   # (no rule at all, e.g. ROI boxes that catch nothing: empty lists)
                L.check(lib.scn_rules_fill(L.ptr(self._table), self.n_off, self._n_out, L.ptr(self._block_sums),
                                           L.ptr(self._in), L.ptr(self._out), L.ptr(self._seg), L.stream()))

    @property
    def in_rows(self):
        self._fill()
        return self._in

    @property
    def out_rows(self):
        self._fill()
        return self._out

    @property
    def seg(self):
        self._fill()
        return self._seg

    def tensors(self):
        return [t for t in (self._table, self._block_sums, self.prefix_dev, self._in, self._out, self._seg)
                if t is not None]


# scn_pyramid_build_ex(SCN_PYRAMID_FUSED): the one-call index build without host round trips (level sizes stay on the device,
# the levels run side by side inside each launch: 17 launches and one host wait for four levels instead of ~136 and five).
# SCN_PYRAMID_FUSED=0 (or SCN_PYRAMID_V1=1 inside the library) keeps the round-3 builder -- same structures, bit for bit.
FUSED_INDEX = os.environ.get("SCN_PYRAMID_FUSED", "1") != "0"

# default of Metadata.xcd_order (SCN_XCD_ORDER=1: every Metadata builds the second order; 0: nobody does)
XCD_ORDER_DEFAULT = os.environ.get("SCN_XCD_ORDER", "") == "1"
XCD_ORDER_BF16 = os.environ.get("SCN_XCD_ORDER", "") != "0"     # bf16-storage networks ask for it
# with the XCD-local order: the LEVEL-0 SubM tiles are sorted by (row bin, mask) -- 8 row ranges = regions of the scene -- instead
# of by mask alone (round 6, profiles/r6_bin_tiles.txt; the fused build does the same: scn_pyramid2.hip `lbins`)
BIN_TILES = os.environ.get("SCN_TB_NO_BINS", "0") in ("", "0")


@dataclass
class Tiles(_Views):
    """Mask-sorted row tiles of a rule table (scn_tiles_build): what the hot kernel scn_conv_tiles walks."""
    perm: torch.Tensor              # int32 [nt*16]
    tstab: torch.Tensor             # int32 [nt, n_off, 16]
    tile_mask: torch.Tensor         # int32 view of uint32 [nt]
    n_off: int
    n: int
    tile_order: torch.Tensor        # int32 [nt], tiles by offset count descending
    has_x: bool = False             # tile_order's buffer continues with the XCD-local order and its bin starts (scn_tiles_build_x)


def build_tiles(table: torch.Tensor, n_off: int, n: int, with_x: bool = False, log2_bins: int = 0) -> Tiles:
    """with_x: also the XCD-local hand-out order of the bf16 tile kernel (behind the first order, in the same buffer).
    log2_bins: the rows are sorted by (row bin, offset mask) instead of by mask alone -- 2^log2_bins row ranges (regions of the
    scene: rows are numbered in mesh order), each mask-sorted on its own (27-offset tables; the bf16 tile kernels, round 6)."""
    lib = L.lib()
    dev = table.device
    nt = (n + 15) // 16
    perm = _empty(nt * 16, torch.int32, dev)
    tstab = torch.empty((nt, n_off, 16), dtype=torch.int32, device=dev)
    tile_mask = _empty(nt, torch.int32, dev)
    order_buf = _empty(lib.scn_tiles_order_ints(n, 1 if with_x else 0), torch.int32, dev)
    scratch = _empty(lib.scn_tiles_scratch_bytes(n_off, n), torch.uint8, dev)
    L.check(lib.scn_tiles_build_x(L.ptr(table), n_off, n, L.ptr(perm), L.ptr(tstab), L.ptr(tile_mask),
                                  L.ptr(order_buf), (1 if with_x else 0) | (int(log2_bins) << 8), L.ptr(scratch), L.stream()))
    return Tiles(perm, tstab, tile_mask, n_off, n, order_buf[:nt] if with_x else order_buf, bool(with_x))


@dataclass
class SubmRulebook(_Views):
    table: Optional[torch.Tensor]   # int32 [k^3, N]; None for k == 1 (identity)
    rules: Optional[Rules]
    k: int
    n: int
    tiles: Optional[Tiles] = None


@dataclass
class StridedRulebook(_Views):
    parent: torch.Tensor            # int32 [Nf]  fine row -> coarse row
    fine_off: torch.Tensor          # int32 [Nf]
    child: torch.Tensor             # int32 [n_off, Nc]
    rules: Rules                    # in = fine rows, out = coarse rows
    n_fine: int
    n_coarse: int
    coarse_size: Tuple[int, ...]
    tiles: Optional[Tiles] = None   # None above 27 offsets (the tile kernels' 32-bit masks): the table-walk GEMM serves those
    n_off: int = 8                  # filter volume = sx * sy * sz
    stride: Tuple[int, ...] = (2, 2, 2)


def compact_rules(table: torch.Tensor, n_off: int, n_out: int, want_seg=False):
    """Rule table -> Rules (lazy: the scan is queued now, sizes and fill on first access)."""
    rules = Rules(table, n_off, n_out, want_seg)
    return (rules, rules.seg) if want_seg else rules


class _Dedup:
    """A queued scn_dedup_launch; `finish()` waits for the row count and returns what `dedup` returns.
    shift: an int (coarse site = coordinate >> shift) or a 3-tuple of divisors (coarse site = coordinate // divisor per axis:
    scn_dedup_launch_div)."""

    def __init__(self, coords_i32, shift, want_counts, want_first, extra=()):
        lib = L.lib()
        dev = coords_i32.device
        n = coords_i32.shape[0]
        self.cap = lib.scn_hash_capacity(n)
        self.keys = _empty(self.cap, torch.int64, dev)
        self.rows = _empty(self.cap, torch.int32, dev)
        self.item_row = _empty(n, torch.int32, dev)
        self.row_count = _empty(n, torch.int32, dev) if want_counts else None
        self.row_first = _empty(n, torch.int32, dev) if want_first else None
        self.row_coords = torch.empty((n, 4), dtype=torch.int32, device=dev)
        scratch = _empty(lib.scn_dedup_scratch_bytes(n), torch.uint8, dev)
        n_rows = _empty(1, torch.int64, dev)
        if isinstance(shift, tuple):
            L.check(lib.scn_dedup_launch_div(L.ptr(coords_i32), n, shift[0], shift[1], shift[2], L.ptr(self.keys),
                                             L.ptr(self.rows), self.cap, L.ptr(self.item_row), L.ptr(self.row_count),
                                             L.ptr(self.row_first), L.ptr(self.row_coords), L.ptr(scratch), L.ptr(n_rows),
                                             L.stream()))
        else:
            L.check(lib.scn_dedup_launch(L.ptr(coords_i32), n, shift, L.ptr(self.keys), L.ptr(self.rows), self.cap,
                                         L.ptr(self.item_row), L.ptr(self.row_count), L.ptr(self.row_first),
                                         L.ptr(self.row_coords), L.ptr(scratch), L.ptr(n_rows), L.stream()))
        self._rb = _Readback(n_rows, *extra)

    def finish(self):
        host = self._rb.get()
        nr = int(host[0][0])
        self.extra = host[1:]
        grid = Grid(self.row_coords[:nr], self.keys, self.rows, self.cap, nr)
        return (grid, self.item_row, self.row_count[:nr] if self.row_count is not None else None,
                self.row_first[:nr] if self.row_first is not None else None)


def dedup(coords_i32: torch.Tensor, shift: int, want_counts: bool, want_first: bool):
    """First-occurrence row numbering of (b, x>>shift, y>>shift, z>>shift).  One host wait (for the row count)."""
    return _Dedup(coords_i32, shift, want_counts, want_first).finish()


class Metadata:
    """``scn.Metadata(dimension)``.  Only dimension 3 is on the reference's path (model.py:31-114: 3D scenes)."""

    # Drop-in path: the reference's module tree never announces how deep a network is -- every layer asks its Metadata
    # for the rulebook it needs.  The first forward over an input spatial size therefore builds rulebooks one by one; the
    # deepest level it reached is remembered here (input size -> number of levels), and later Metadata objects of that
    # input size build the whole pyramid with ONE scn_pyramid_build call when their InputLayer rules are set.  A wrong
    # hint only costs unused structures (or falls back to lazy building); results never depend on it.
    LEVELS_HINT: Dict[Tuple[int, ...], int] = {}
    AUTO_NATIVE = True

    def __init__(self, dimension=3):
        if int(dimension) != 3:
            raise NotImplementedError("sparse_rcnn_amd.Metadata: only dimension 3 (the reference's ScanNet path)")
        self.dimension = 3
        self.grids: Dict[Tuple[int, ...], Grid] = {}
        self.subm: Dict[Tuple[Tuple[int, ...], int], SubmRulebook] = {}
        self.strided: Dict[Tuple[int, ...], StridedRulebook] = {}
        self.strided_general = {}       # (fine size, stride) -> StridedRulebook for strides other than (2, 2, 2)
        # InputLayer bookkeeping (kept for OutputLayer: custom_operations.py:7-10)
        self.input_size: Optional[Tuple[int, ...]] = None
        self.item_row: Optional[torch.Tensor] = None
        self.row_count: Optional[torch.Tensor] = None
        self.row_first: Optional[torch.Tensor] = None
        self.row_last: Optional[torch.Tensor] = None
        self.n_items = 0
        self.n_samples = 0
        self.device = None
        self.ready_event = None         # set by prepare_async: index structures were built on a side stream
        self.point_coords: Optional[torch.Tensor] = None     # int32 [Npts,4] device copy of the InputLayer coordinates
        self._depth: Dict[Tuple[int, ...], int] = {}
        self._unrequested = set()       # strided rulebooks built ahead on a depth hint that no layer has asked for yet
        # SubM tiles also get the XCD-local hand-out order of the bf16 tile kernel (scn_tiles_build_x): set by the owner of a
        # bf16-storage network BEFORE the structures are built (unet.Backbone / maskhead.MaskBranch do)
        self.xcd_order = XCD_ORDER_DEFAULT
        self._prepared_for = None       # identity of the coords tensor a prefetch was built for

    def _note_levels(self, size, extra):
        """A layer asked for a rulebook that needs depth(size) + extra levels: remember the deepest request per input size."""
        if self.input_size is not None:
            need = self._depth.get(tuple(size), 0) + extra
            if need > Metadata.LEVELS_HINT.get(self.input_size, 0):
                Metadata.LEVELS_HINT[self.input_size] = need

    # ---- InputLayer rules -------------------------------------------------------------------------
    def set_input(self, spatial_size, coords: torch.Tensor, batch_size: int, mode: int, auto_native: bool = True):
        """coords: int64 [Npts, 4] (x,y,z,batch), CPU (the reference's contract, data.py:95-98,207-210) or device."""
        lib = L.lib()
        size = tuple(int(s) for s in spatial_size)
        if len(size) != 3:
            raise ValueError("spatial_size must have 3 entries")
        if coords.dim() != 2 or coords.shape[1] != 4:
            raise ValueError("coords must be [N, 4] = (x, y, z, batch); single-sample [N,3] input is not used by the reference")
        levels = Metadata.LEVELS_HINT.get(size, 0) if (auto_native and Metadata.AUTO_NATIVE) else 0
        if levels >= 1 and coords.shape[0] > 0:
            lv, ok = size, 1
            while ok < levels and all(v % 2 == 0 for v in lv):
                lv, ok = tuple(v // 2 for v in lv), ok + 1
            self.build_native(spatial_size, coords, batch_size, mode, ok, 3)
            # built on a hint, not on a layer's request: a Deconvolution may only use a strided rulebook once a
            # Convolution / pooling layer of this forward has asked for it (custom_container.py:70-83)
            self._unrequested = set(self.strided)
            return self.grids[size]
        dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        c64 = coords.to(device=dev, dtype=torch.int64).contiguous()
        n = c64.shape[0]
        c32 = torch.empty((n, 4), dtype=torch.int32, device=dev)
        flag = _empty(1, torch.int32, dev)
        L.check(lib.scn_coords_to_i32(L.ptr(c64), n, L.ptr(c32), L.ptr(flag), None, L.stream()))
        dd = _Dedup(c32, 0, True, True, extra=(flag,))           # range flag and row count come back together
        grid, item_row, row_count, row_first = dd.finish()
        if int(dd.extra[0][0]):
            raise L.ScnError(f"libscn_mi355x error {L.EHASH}: coordinates outside [0,65535] "
                             f"in {int(dd.extra[0][0])} wave(s)")
        self.grids[size] = grid
        self.input_size = size
        self._depth[size] = 0
        self.point_coords = c32
        self.item_row, self.row_count, self.row_first = item_row, row_count, row_first
        self.n_items = n
        if mode == 0 and grid.n != n:
            raise L.ScnError("InputLayer mode 0 requires unique coordinates")
        if batch_size and batch_size > 0:
            self.n_samples = int(batch_size)
        else:
            self.n_samples = int(c64[:, 3].max().item()) + 1 if n else 0
        return grid

    def build_native(self, spatial_size, coords: torch.Tensor, batch_size: int, mode: int, n_levels: int, k: int = 3,
                     two_queues: bool = True, xcd_order: Optional[bool] = None):
        """set_input + build_pyramid through ONE C call (scn_pyramid_build_ex): same structures, bit-identical, carved out of
        one workspace tensor; the call holds no interpreter state, so a helper thread can run it next to the main
        thread's kernel queueing (prepare_in_thread).  two_queues: the SubM work of the levels on the library's side stream
        (shorter when the build has the GPU to itself: the inline builds; the pipelined prefetch passes False).
        xcd_order: the SubM tiles also get the XCD-local hand-out order of the bf16 tile kernel (None: `self.xcd_order`)."""
        lib = L.lib()
        if xcd_order is not None:
            self.xcd_order = bool(xcd_order)
        size = tuple(int(s) for s in spatial_size)
        if len(size) != 3:
            raise ValueError("spatial_size must have 3 entries")
        if coords.dim() != 2 or coords.shape[1] != 4:
            raise ValueError("coords must be [N, 4] = (x, y, z, batch)")
        if not 1 <= n_levels <= L.PYRAMID_MAX_LEVELS or k not in (1, 3):
            raise ValueError("n_levels in 1..8 and k in (1, 3)")
        lv_size = size
        for _ in range(n_levels - 1):
            if any(v % 2 for v in lv_size):
                raise L.ScnError(f"Convolution size=stride=2 needs even spatial size, got {lv_size} "
                                 "((out-1)*stride+filter != in)")
            lv_size = tuple(v // 2 for v in lv_size)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        c64 = coords.to(device=dev, dtype=torch.int64).contiguous()
        n = c64.shape[0]
        if n == 0:                       # nothing to build natively; the step-by-step path handles the empty batch
            self.set_input(spatial_size, coords, batch_size, mode, auto_native=False)
            return self
        ws = torch.empty(lib.scn_pyramid_workspace_bytes(n, n_levels, k), dtype=torch.uint8, device=dev)
        desc = (C.c_int64 * L.PYRAMID_DESC_LEN)()
        L.check(lib.scn_pyramid_build_ex(L.ptr(c64), n, n_levels, k, L.ptr(ws), ws.numel(), desc,
                                         (L.PYRAMID_TWO_QUEUES if two_queues else 0) |
                                         (L.PYRAMID_XCD_ORDER if self.xcd_order else 0) |
                                         (L.PYRAMID_FUSED if FUSED_INDEX else 0), L.stream()))
        self._workspace = ws

        wsp = ws.data_ptr()
        i32, i64 = torch.int32, torch.int64
        n_off = k ** 3
        n0 = int(desc[8])
        if int(desc[3]):
            raise L.ScnError(f"libscn_mi355x error {L.EHASH}: coordinates outside the key range in {int(desc[3])} wave(s)")
        self.input_size = size
        # (the records below hold byte offsets into the workspace; a tensor view exists only once somebody reads the attribute)
        self.point_coords = _ws_view(ws, int(desc[7]), (n, 4), i32)
        self.item_row = _ws_view(ws, int(desc[4]), (n,), i32)
        self.row_count = _ws_view(ws, int(desc[5]), (n0,), i32)
        self.row_first = _ws_view(ws, int(desc[6]), (n0,), i32)
        self.n_items = n
        if mode == 0 and n0 != n:
            raise L.ScnError("InputLayer mode 0 requires unique coordinates")
        self.n_samples = int(batch_size) if batch_size and batch_size > 0 else int(c64[:, 3].max().item()) + 1
        lv_size = size
        new = object.__new__
        stride = L.PYRAMID_LEVEL_STRIDE
        per_level = [desc[8 + l * stride:8 + (l + 1) * stride] for l in range(n_levels)]      # (ctypes slices: plain int lists)
        # (executor.build_levels fills its level table straight from these descriptors: no attribute chains, no views)
        self._native = (size, n_levels, k, wsp, per_level, bool(self.xcd_order))
        for l in range(n_levels):
            D = per_level[l]
            Dn = per_level[l + 1] if l + 1 < n_levels else None
            nl, cap = D[0], D[1]
            grid = new(Grid)._lazy(ws, wsp, {"coords": (D[2], (nl, 4), i32), "table_keys": (D[3], (cap,), i64),
                                             "table_rows": (D[4], (cap,), i32)})
            grid.__dict__["cap"], grid.__dict__["n"] = cap, nl
            self.grids[lv_size] = grid
            self._depth[lv_size] = l
            if k == 1:
                self.subm[(lv_size, 1)] = SubmRulebook(None, None, 1, nl)
            elif nl > 0:
                nt = D[13]
                P = D[25 + n_off]
                rules = Rules.from_workspace(ws, wsp, n_off, nl, D[25:25 + n_off + 1],
                                             {"_table": (D[5], (n_off, nl), i32), "_block_sums": (D[6], (D[7],), i32),
                                              "prefix_dev": (D[8], (n_off + 1,), i64), "_in": (D[64], (P,), i32),
                                              "_out": (D[65], (P,), i32)})
                tiles = new(Tiles)._lazy(ws, wsp, {"perm": (D[9], (nt * 16,), i32), "tstab": (D[10], (nt, n_off, 16), i32),
                                                   "tile_mask": (D[11], (nt,), i32), "tile_order": (D[12], (nt,), i32)})
                td = tiles.__dict__
                td["n_off"], td["n"], td["has_x"] = n_off, nl, bool(self.xcd_order)
                rb = new(SubmRulebook)._lazy(ws, wsp, {"table": (D[5], (n_off, nl), i32)})
                rd = rb.__dict__
                rd["rules"], rd["k"], rd["n"], rd["tiles"] = rules, k, nl, tiles
                self.subm[(lv_size, k)] = rb
            if l + 1 < n_levels and nl > 0:
                nc = Dn[0]
                ntc = D[24]
                Pc = D[61]
                rules = Rules.from_workspace(ws, wsp, 8, nc, D[53:62],
                                             {"_table": (D[16], (8, nc), i32), "_block_sums": (D[17], (D[18],), i32),
                                              "prefix_dev": (D[19], (9,), i64), "_in": (D[66], (Pc,), i32),
                                              "_out": (D[67], (Pc,), i32)})
                tiles = new(Tiles)._lazy(ws, wsp, {"perm": (D[20], (ntc * 16,), i32), "tstab": (D[21], (ntc, 8, 16), i32),
                                                   "tile_mask": (D[22], (ntc,), i32), "tile_order": (D[23], (ntc,), i32)})
                td = tiles.__dict__
                td["n_off"], td["n"], td["has_x"] = 8, nc, False
                coarse = tuple(v // 2 for v in lv_size)
                sb = new(StridedRulebook)._lazy(ws, wsp, {"parent": (D[14], (nl,), i32), "fine_off": (D[15], (nl,), i32),
                                                          "child": (D[16], (8, nc), i32)})
                sd = sb.__dict__
                sd["rules"], sd["n_fine"], sd["n_coarse"], sd["coarse_size"], sd["tiles"] = rules, nl, nc, coarse, tiles
                self.strided[lv_size] = sb
                lv_size = coarse
            elif l + 1 < n_levels:
                break
        return self

    # ---- rulebooks ----------------------------------------------------------------------------------
    def _is_input_size(self, size) -> bool:
        return self.input_size is not None and tuple(int(v) for v in size) == tuple(int(v) for v in self.input_size)

    def grid(self, size) -> Grid:
        size = tuple(int(s) for s in size)
        if size not in self.grids:
            raise L.ScnError(f"Metadata holds no grid of spatial size {size}")
        return self.grids[size]

    def subm_rulebook(self, size, k: int) -> SubmRulebook:
        size = tuple(int(s) for s in size)
        key = (size, int(k))
        rb = self.subm.get(key)
        if rb is None:
            g = self.grid(size)
            if k == 1:
                rb = SubmRulebook(None, None, 1, g.n)
            else:
                lib = L.lib()
                table = torch.empty((k ** 3, g.n), dtype=torch.int32, device=g.coords.device)
                L.check(lib.scn_subm_table(L.ptr(g.coords), g.n, L.ptr(g.table_keys), L.ptr(g.table_rows), g.cap, k,
                                           L.ptr(table), L.stream()))
                # (filters above 3^3 -- 125 offsets -- have no mask tiles: the tile kernels hold a tile's offsets in 27 bits)
                rb = SubmRulebook(table, compact_rules(table, k ** 3, g.n), k, g.n,
                                  build_tiles(table, k ** 3, g.n, with_x=self.xcd_order,
                                              log2_bins=3 if (self.xcd_order and k == 3 and self._is_input_size(size)
                                                              and BIN_TILES) else 0) if k ** 3 <= 27 else None)
            self.subm[key] = rb
        if k == 3:
            self._note_levels(size, 1)
        return rb

    def _strided_launch(self, size):
        """Queue the coarse-site numbering of a size=stride=2 Convolution; `_strided_finish` needs its row count."""
        if any(s % 2 for s in size):
            raise L.ScnError(f"Convolution size=stride=2 needs even spatial size, got {size} "
                             "((out-1)*stride+filter != in)")
        coarse_size = tuple(s // 2 for s in size)
        if coarse_size in self.grids:
            raise L.ScnError(f"Metadata already holds a grid of size {coarse_size}")
        return _Dedup(self.grid(size).coords, 1, False, False)

    def _strided_finish(self, size, pending) -> StridedRulebook:
        coarse_size = tuple(s // 2 for s in size)
        g = self.grid(size)
        lib = L.lib()
        dev = g.coords.device
        cg, parent, _, _ = pending.finish()
        self.grids[coarse_size] = cg
        self._depth[coarse_size] = self._depth.get(tuple(size), 0) + 1
        child = torch.empty((8, cg.n), dtype=torch.int32, device=dev)
        fine_off = _empty(g.n, torch.int32, dev)
        L.check(lib.scn_child_table(L.ptr(g.coords), L.ptr(parent), g.n, cg.n, L.ptr(child), L.ptr(fine_off),
                                    L.stream()))
        rules = compact_rules(child, 8, cg.n)
        rb = StridedRulebook(parent, fine_off, child, rules, g.n, cg.n, coarse_size, build_tiles(child, 8, cg.n))
        self.strided[size] = rb
        return rb

    def strided_rulebook(self, size, stride=(2, 2, 2)) -> StridedRulebook:
        """size = stride Convolution from `size` to size / stride; creates the coarse grid on first use.  stride (2, 2, 2) is the
        reference's (and the only one the fused index build, the step executor and the level hints know); any other
        per-axis stride (module_factory.py:221-241) builds its coarse grid and child table on request
        (`_general_strided_rulebook`)."""
        size = tuple(int(s) for s in size)
        stride = tuple(int(v) for v in stride)
        if stride != (2, 2, 2):
            return self._general_strided_rulebook(size, stride)
        rb = self.strided.get(size)
        if rb is None and tuple(s // 2 for s in size) in self.grids and not any(s % 2 for s in size):
            rb = self.strided[size] = self._general_strided_rulebook(size, stride)      # into the grid that exists
        if rb is None:
            rb = self._strided_finish(size, self._strided_launch(size))
        self._unrequested.discard(size)
        self._note_levels(size, 2)
        return rb

    def _general_strided_rulebook(self, size, stride) -> StridedRulebook:
        rb = self.strided_general.get((size, stride))
        if rb is not None:
            return rb
        if any(st < 1 for st in stride) or any(s % st for s, st in zip(size, stride)):
            raise L.ScnError(f"Convolution size=stride={stride} needs a spatial size that is a multiple of it, got {size} "
                             "((out-1)*stride+filter != in)")
        coarse_size = tuple(s // st for s, st in zip(size, stride))
        g = self.grid(size)
        lib = L.lib()
        dev = g.coords.device
        n_off = stride[0] * stride[1] * stride[2]
        if stride == (1, 1, 1):                       # a 1^3 / 1 "downsampler" keeps the grid
            cg = g
            parent = torch.arange(g.n, dtype=torch.int32, device=dev)
        elif coarse_size in self.grids:
            # Another path of the network has reached this spatial size on this Metadata (64 -> 16 by stride 4 after 64 -> 32 ->
            # 16 by two stride-2 layers): SparseConvNet keys its grids by spatial size, so the layer lands in THAT grid and its
            # row numbering.  Upstream would also add sites the grid lacks -- under the tensors that already live on it; every
            # site is there whenever both paths start from the same input, anything else is refused.
            cg = self.grids[coarse_size]
            parent = _empty(g.n, torch.int32, dev)
            missing = torch.empty(1, dtype=torch.int64, device=dev)
            L.check(lib.scn_parent_lookup_div(L.ptr(g.coords), g.n, stride[0], stride[1], stride[2], L.ptr(cg.table_keys),
                                              L.ptr(cg.table_rows), cg.cap, L.ptr(parent), L.ptr(missing), L.stream()))
            n_missing = int(missing.item())
            if n_missing:
                raise L.ScnError(f"Metadata already holds a grid of size {coarse_size} that lacks {n_missing} of the sites a "
                                 f"size=stride={stride} Convolution from {size} needs (SparseConvNet would grow that grid "
                                 "under the tensors living on it)")
        else:
            cg, parent, _, _ = _Dedup(g.coords, stride, False, False).finish()
            self.grids[coarse_size] = cg
            self._depth[coarse_size] = self._depth.get(size, 0) + 1
        child = torch.empty((n_off, cg.n), dtype=torch.int32, device=dev)
        fine_off = _empty(g.n, torch.int32, dev)
        L.check(lib.scn_child_table_div(L.ptr(g.coords), L.ptr(parent), g.n, cg.n, stride[0], stride[1], stride[2],
                                        L.ptr(child), L.ptr(fine_off), L.stream()))
        rules = compact_rules(child, n_off, cg.n)
        tiles = build_tiles(child, n_off, cg.n) if n_off <= 27 else None
        rb = StridedRulebook(parent, fine_off, child, rules, g.n, cg.n, coarse_size, tiles, n_off, stride)
        self.strided_general[(size, stride)] = rb
        return rb

    def cached_strided_rulebook(self, size, stride=(2, 2, 2)):
        """The rulebook a Deconvolution back to `size` re-uses, or None when no layer of this forward has built it."""
        size = tuple(int(s) for s in size)
        stride = tuple(int(v) for v in stride)
        if stride != (2, 2, 2):
            return self.strided_general.get((size, stride))
        return None if size in self._unrequested else self.strided.get(size)

    def prepare_async(self, spatial_size, coords, batch_size: int = 0, mode: int = 4, n_levels: int = 0, k: int = 3,
                      native: bool = False, caller_stream=None, xcd_order: Optional[bool] = None):
        """Build the InputLayer rules (and optionally the rulebook pyramid of an n_levels U-Net) on the index stream.

        All index structures depend only on the coordinates, so a training loop can build those of batch i+1 while the
        matrix kernels of batch i still run: the index kernels are short and latency-bound and fit beside them, and
        the host syncs of the size queries then wait on the index stream only.  `InputLayerFunction` adopts a prepared
        Metadata (same coords) instead of rebuilding it; `handover()` orders the consumer stream behind the build."""
        if xcd_order is not None:
            self.xcd_order = bool(xcd_order)
        side = index_stream(torch.device("cuda", torch.cuda.current_device()))
        # coords may have been produced on the CALLER's stream (prepare_in_thread captures it: on the helper thread
        # torch.cuda.current_stream() is that thread's default stream, not the caller's)
        side.wait_stream(caller_stream if caller_stream is not None else torch.cuda.current_stream())
        self._prepared_for = (coords, coords._version)         # strong reference: the identity cannot be recycled
        with torch.cuda.stream(side):
            if native and n_levels:        # beside another batch's matrix kernels: one queue (include/scn_mi355x.h)
                self.build_native(spatial_size, coords, batch_size, mode, n_levels, k, two_queues=False)
            else:
                self.set_input(spatial_size, coords, batch_size, mode, auto_native=False)
                if n_levels:
                    self.build_pyramid(spatial_size, n_levels, k)
            self.ready_event = torch.cuda.Event()
            self.ready_event.record(side)
        return self

    def prepare_in_thread(self, spatial_size, coords, batch_size: int = 0, mode: int = 4, n_levels: int = 0,
                          k: int = 3, native: bool = True, xcd_order: Optional[bool] = None) -> "PendingMetadata":
        """prepare_async on a helper thread.  The build waits four times for a row count; on the caller's thread those
        waits would keep it from queueing the matrix kernels of the current batch (measured: slower than no prefetch).
        The helper spends its time inside C calls and event waits, which release the GIL."""
        dev = torch.cuda.current_device()
        caller_stream = torch.cuda.current_stream() if coords.is_cuda else None

        def fn():
            torch.cuda.set_device(dev)
            return self.prepare_async(spatial_size, coords, batch_size, mode, n_levels, k, native=native,
                                      caller_stream=caller_stream, xcd_order=xcd_order)
        return PendingMetadata(fn)

    def prepared_for(self, coords: torch.Tensor) -> bool:
        """Was this (prefetched) Metadata built for `coords`?  The same tensor object and version is accepted at once;
        anything else is compared value by value with the stored int32 copy (one host wait: a rare path)."""
        pf = self._prepared_for
        if pf is not None and pf[0] is coords and pf[1] == coords._version:
            return True
        if self.point_coords is None or tuple(coords.shape) != tuple(self.point_coords.shape):
            return False
        return bool((coords.to(self.point_coords.device) == self.point_coords).all().item())

    def _all_tensors(self):
        out = [self.item_row, self.row_count, self.row_first, self.row_last, self.point_coords,
               getattr(self, "_workspace", None)]
        for g in self.grids.values():
            out += [g.coords, g.table_keys, g.table_rows]
        for rb in list(self.subm.values()) + list(self.strided.values()) + list(self.strided_general.values()):
            for obj in (rb, getattr(rb, "tiles", None)):
                if obj is not None:
                    out += [v for v in vars(obj).values() if isinstance(v, torch.Tensor)]
            if getattr(rb, "rules", None) is not None:
                out += rb.rules.tensors()
        return [t for t in out if t is not None]

    def handover(self):
        """Make the current stream wait for a side-stream build and tell the allocator the buffers are used here."""
        if self.ready_event is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(self.ready_event)
            ws = getattr(self, "_workspace", None)
            if ws is not None:                   # native build: every index tensor is a view of one allocation
                ws.record_stream(cur)
            else:
                for t in self._all_tensors():
                    t.record_stream(cur)
            self.ready_event = None

    def build_pyramid(self, size, n_levels: int, k: int = 3):
        """Build the rulebooks of an n_levels U-Net up front (SubM k^3 at every level, 2^3/2 between levels).
        Same work as building them lazily, ordered so that the host never idles the device: the coarse-site numbering
        of level l+1 is queued first, then the SubM table / scan / tiles of level l (which do not need its row
        count), and only then the count is awaited."""
        size = tuple(int(s) for s in size)
        for level in range(n_levels):
            pending = None
            if level + 1 < n_levels and size not in self.strided and tuple(s // 2 for s in size) not in self.grids:
                pending = self._strided_launch(size)
            self.subm_rulebook(size, k)
            if level + 1 < n_levels:
                if pending is not None:
                    self._strided_finish(size, pending)
                elif size not in self.strided:
                    self.strided_rulebook(size)                # the coarse grid exists (another path built it): reuse it
                size = tuple(s // 2 for s in size)

    # ---- parity helpers (tests) -------------------------------------------------------------------
    def get_spatial_locations(self, size) -> torch.Tensor:
        """int64 CPU [N,4] in row order (roi_select_sparse.py:103)."""
        return self.grid(size).coords.to(torch.int64).cpu()
