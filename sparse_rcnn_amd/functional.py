"""Autograd functions of the sparse operator set, each a thin host wrapper over the C ABI.

Forward / backward definitions follow SURVEY.md Appendix B; each function names the reference call site it serves.
No function here falls back to PyTorch arithmetic: feature math runs in libscn_mi355x kernels only.
"""
from __future__ import annotations

import os

import ctypes as C

import torch

from . import _lib as L
from . import profiling
from .metadata import Metadata, Rules


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"sparse_rcnn_amd operators compute in fp32; got {t.dtype}")
    if not t.is_cuda:
        raise L.ScnError("feature tensors must live on the MI355X (no CPU fallback)")
    return t.contiguous()


def _new(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


def _feat(t: torch.Tensor) -> torch.Tensor:
    """Feature slab of a conv-type layer: fp32, or bf16 STORAGE (BASELINE configs 3-5: the bf16 entry points of the
    library; parameters and their gradients stay fp32)."""
    if t.dtype == torch.bfloat16:
        if not t.is_cuda:
            raise L.ScnError("feature tensors must live on the MI355X (no CPU fallback)")
        return t.contiguous()
    return _f32(t)


def _is_bf16(t):
    return t.dtype == torch.bfloat16


# Debug hook for the parity tests: when a list, every FORWARD that applies a ReLU to a stored slab
# (the fused input ReLU of a convolution, scn.ReLU) appends that slab's sign mask (x > 0, bool, on the host) in call order.
# An oracle that takes these masks instead of its own `relu` decisions differentiates the same piecewise-linear function,
# so ReLU inputs within rounding of zero no longer separate the two gradients.  None (default): nothing is recorded.
RELU_RECORD = None


def _rec_relu(X):
    if RELU_RECORD is not None:
        RELU_RECORD.append((X.detach() > 0).cpu())


# ------------------------------------------------------------------------------------------------------
# raw kernels (no autograd)
# ------------------------------------------------------------------------------------------------------

def _conv_bytes(n_in, cin, n_out, cout, n_off, n_rules):
    """Compulsory HBM bytes of one conv-type launch: X once, Y once, W, int32 rule pairs."""
    return 4.0 * (n_in * cin + n_out * cout + n_off * cin * cout) + 8.0 * n_rules


def _count(n_rules):
    """n_rules: an int or a metadata.RuleCount (whose total is read lazily; it holds no device buffers)."""
    return n_rules if isinstance(n_rules, int) else n_rules.total


def gemm_table(X, table, n_off, n_out, W, bias, cout, flags=0, residual=None, relu_mask=None, n_rules=None):
    """n_rules: number of (in,out) rules the table holds (int or RuleCount) -- only used for the algorithmic-FLOP
    accounting, evaluated when a profile is summarised."""
    lib = L.lib()
    cin = X.shape[1]
    Y = _new((n_out, cout), X, X.dtype)
    n_in = X.shape[0]
    P = (lambda: n_out) if n_rules is None else (lambda: _count(n_rules))
    entry = lib.scn_gemm_table_bf16 if _is_bf16(X) else lib.scn_gemm_table       # bf16 storage: same arithmetic

    def run():
        L.check(entry(L.ptr(X), X.shape[0], cin, L.ptr(table), n_off, n_out, L.ptr(W), L.ptr(bias),
                      L.ptr(residual), L.ptr(relu_mask), L.ptr(Y), cout, flags, L.stream()))
    profiling.timed("k_gemm_table", lambda: 2.0 * P() * cin * cout,
                    lambda: _conv_bytes(n_in, cin, n_out, cout, n_off, P()), run)
    return Y


_ts_scratch_cache = {}
_tb_sizes = {}
FUSED_K = True      # scn_conv_tiles adds its K-chunk partial sums inside the launch (False: second launch k_conv_ts_sum)


def _conv_tiles_scratch_bytes(cin, n_out, cout):
    key = (cin, n_out, cout)
    v = _ts_scratch_cache.get(key)
    if v is None:
        if len(_ts_scratch_cache) > 4096:
            _ts_scratch_cache.clear()
        v = _ts_scratch_cache[key] = L.lib().scn_conv_tiles_scratch_bytes(cin, n_out, cout)
    return v


def conv_rules(X, tiles, n_out, W, bias, cout, flags=0, residual=None, relu_mask=None, n_rules=0, image=None):
    """The hot kernel: output-stationary convolution over mask-sorted tiles (scn_conv_tiles).  image: bf16 storage only,
    a packed weight image for these flags."""
    if _is_bf16(X):
        return conv_rules_bf16(X, tiles, n_out, W, bias, cout, flags, residual, relu_mask, image=image, n_rules=n_rules)
    lib = L.lib()
    cin = X.shape[1]
    Y = _new((n_out, cout), X)
    n_in, n_off = X.shape[0], tiles.n_off      # the profiling closures below must not keep device buffers alive
    P = lambda: _count(n_rules)
    scratch = L.scratch(_conv_tiles_scratch_bytes(cin, n_out, cout), X.device)
    n_kc = (cin + 31) // 32
    # layers with more than 32 input channels: the K-chunk partial sums are added inside the launch (zeroed arrival
    # counters, left zero by the kernel); FUSED_K = False is the two-launch form (cross-check in the tests)
    arr = L.arrival(((n_out + 15) // 16) * ((cout + 31) // 32), X.device) if (n_kc > 1 and FUSED_K) else None

    def run(fl=flags):
        L.check(lib.scn_conv_tiles(L.ptr(X), n_in, cin, L.ptr(tiles.tstab), L.ptr(tiles.tile_mask), L.ptr(tiles.perm),
                                   L.ptr(tiles.tile_order), tiles.n_off, n_out, L.ptr(W), L.ptr(bias), L.ptr(residual), L.ptr(relu_mask),
                                   L.ptr(Y), cout, fl, L.ptr(scratch), L.ptr(arr), L.stream()))
    if profiling.TIMER is None:
        run()
        return Y
    if arr is not None or n_kc == 1:
        # one launch per convolution: the timing events bracket exactly the production kernel
        profiling.timed("k_conv_ts", lambda: 2.0 * P() * cin * cout,
                        lambda: _conv_bytes(n_in, cin, n_out, cout, n_off, P()), run)
        return Y
    # two-launch form, timed: the events bracket the tile kernel alone; the K-chunk slab sum is launched -- and timed -- apart
    profiling.timed("k_conv_ts", lambda: 2.0 * P() * cin * cout,
                    lambda: _conv_bytes(n_in, cin, n_out, cout, n_off, P()), lambda: run(flags | L.F_SPLIT_SUM))
    profiling.timed("k_conv_ts_sum", 0.0, 4.0 * n_out * cout * (n_kc + 1) if n_kc > 1 else 0.0,
                    lambda: L.check(lib.scn_conv_tiles_finish(cin, n_out, L.ptr(bias), L.ptr(residual), L.ptr(relu_mask),
                                                              L.ptr(Y), cout, flags, L.ptr(scratch), L.stream())))
    return Y


def pack_weights_bf16(W, cin, cout, n_off, flags=0):
    """scn_conv_tiles_bf16_pack: the layer's fp32 master weights -> the bf16 image the tile kernel stages (one rounding per
    weight).  flags: F_W_TRANSPOSED | F_OFF_REVERSE select the backward-data image.  Valid until W changes: callers pack
    per use (forward / backward of one step) and never cache across steps."""
    lib = L.lib()
    img = torch.empty(lib.scn_conv_tiles_bf16_image_bytes(cin, cout, n_off), dtype=torch.uint8, device=W.device)
    L.check(lib.scn_conv_tiles_bf16_pack(L.ptr(W), cin, cout, n_off, flags & (L.F_W_TRANSPOSED | L.F_OFF_REVERSE),
                                         L.ptr(img), L.stream()))
    return img


# ---- channel-padded parameters of a network, one launch each way -----------------------------------------------------------
# modules._ConvBase / NetworkInNetwork hand the kernels zero-padded copies of their logical parameters where a level runs on
# padded slabs (unet.SparseUNet.phys0: the mask network's 23 -> 24 columns).  Through torch.nn.functional.pad that is a fill +
# a copy per tensor forward and a slice copy backward -- ~50 launches of ~4.5 us per detection + mask step.  `padded_params`
# pads every tensor of a fixed list with ONE scn_pad_params_many launch (its backward: one launch slices all gradients);
# `_wb` of the modules picks its tensors up from PADDED while the block is open.
PADDED = {}
PAD_MANY = os.environ.get("SCN_PAD_MANY", "1") != "0"      # 0: tensor-by-tensor torch pads (A/B, cross-check in the tests)
PAD_RECORD = None        # a list while a network records which (module, cin_phys) its forward asks `_wb` for


class PadPlan:
    """jobs: [(key, module, attribute, padded shape, segments)] -- key: what `_wb` asks PADDED for ((id(module), cin_phys,
    "w" | "b")); getattr(module, attribute): the logical parameter, [fv, rows, cols] / [rows, cols] / [cols] (fetched at every
    use: module.to(...) replaces the tensors); padded shape: same rank; segments: up to two (src row, count, dst row) triples
    (None: all rows at row 0)."""

    @classmethod
    def from_requests(cls, requests):
        """requests: the (module, cin_phys) pairs a forward recorded (PAD_RECORD) -> plan, or False when there are none."""
        jobs, seen = [], set()
        for m, cin_phys in requests:
            if (id(m), cin_phys) not in seen:
                seen.add((id(m), cin_phys))
                jobs += m._pad_jobs(cin_phys)
        return cls(jobs) if jobs else False

    def fetch(self):
        return [getattr(m, a) for _, m, a, _, _ in self.jobs]

    def __init__(self, jobs):
        self.jobs = list(jobs)
        self.n = len(self.jobs)
        self.params = self.fetch()
        desc, self.out_shapes, self.out_offs, self.in_offs = [], [], [], []
        tot_out = tot_in = 0
        for (key, _, _, shape, segs), t in zip(self.jobs, self.params):
            s3 = (1,) * (3 - t.dim()) + tuple(t.shape)
            d3 = (1,) * (3 - len(shape)) + tuple(int(v) for v in shape)
            segs = list(segs) if segs else [(0, s3[1], 0)]
            segs += [(0, 0, 0)] * (2 - len(segs))
            desc += [s3[0], s3[1], s3[2], d3[1], d3[2], *segs[0], *segs[1]]
            if d3[0] != s3[0]:
                raise ValueError("PadPlan: the leading dimension is never padded")
            self.out_shapes.append(tuple(int(v) for v in shape))
            self.out_offs.append(tot_out); self.in_offs.append(tot_in)
            tot_out += (d3[0] * d3[1] * d3[2] + 3) & ~3                       # 16-byte aligned tensors
            tot_in += (t.numel() + 3) & ~3
        self.total_out, self.total_in = tot_out, tot_in
        self.desc = (C.c_int32 * len(desc))(*desc)
        self.src = (C.c_void_p * self.n)()
        self.dst = (C.c_void_p * self.n)()
        self.keys = [j[0] for j in self.jobs]


class _PadMany(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, *params):
        buf = torch.empty(plan.total_out, dtype=torch.float32, device=params[0].device)
        base = buf.data_ptr()
        for i, p in enumerate(params):
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError("padded_params: contiguous fp32 parameters")
            plan.src[i] = p.data_ptr()
            plan.dst[i] = base + 4 * plan.out_offs[i]
        L.check(L.lib().scn_pad_params_many(plan.n, plan.src, plan.dst, plan.desc, 0, L.stream()))
        ctx.plan = plan
        outs = []
        for off, shape in zip(plan.out_offs, plan.out_shapes):
            n = 1
            for v in shape:
                n *= v
            outs.append(buf[off:off + n].view(shape))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        plan = ctx.plan
        dev = next(g.device for g in grads if g is not None)
        buf = torch.empty(plan.total_in, dtype=torch.float32, device=dev)
        base, keep = buf.data_ptr(), []
        src, dst = (C.c_void_p * plan.n)(), (C.c_void_p * plan.n)()      # (own tables: backward runs on autograd's thread)
        for i, g in enumerate(grads):
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                g = g.float().contiguous()
            keep.append(g)
            src[i] = g.data_ptr() if g is not None else None
            dst[i] = base + 4 * plan.in_offs[i]
        L.check(L.lib().scn_pad_params_many(plan.n, src, dst, plan.desc, 1, L.stream()))
        return (None,) + tuple(buf[off:off + p.numel()].view(p.shape) for off, p in zip(plan.in_offs, plan.params))


class padded_params:
    """Context manager: the padded tensors of `plan` are in PADDED until the block ends (nested blocks of other networks keep
    theirs).  Outside a CUDA device or with no jobs it does nothing -- `_wb` then pads tensor by tensor as before."""

    def __init__(self, plan):
        self.plan, self.keys = plan, []

    def __enter__(self):
        pl = self.plan
        if not pl or not pl.n:
            return self
        pl.params = pl.fetch()
        if not pl.params[0].is_cuda:
            return self
        outs = _PadMany.apply(pl, *pl.params)
        for key, t in zip(pl.keys, outs):
            PADDED[key] = t
        self.keys = pl.keys
        return self

    def __exit__(self, *exc):
        for k in self.keys:
            PADDED.pop(k, None)
        return False


# Weight images packed for the forward pass that is running (packed_weights below): (data_ptr, cin, cout, n_off, flags & 6)
# -> image.  Filled by ONE launch for a whole network, emptied when its forward ends; a backward-data image is handed to the
# autograd node during the forward (ctx) -- nothing here outlives a step, so an optimizer update can never meet a stale image.
PACKED = {}
_BACK = L.F_W_TRANSPOSED | L.F_OFF_REVERSE


def packed_image(W, cin, cout, n_off, flags):
    """The packed bf16 image of a layer's weight as a tensor (a view of the pack buffer), or None."""
    e = PACKED.get((W.data_ptr(), cin, cout, n_off, flags & _BACK))
    if e is None or torch.is_tensor(e):
        return e
    buf, o, b, _ = e
    return buf[o:o + b]


def packed_entry(W, cin, cout, n_off, flags):
    """(pack buffer, device address of the image) or None -- what the step executor needs (an address for its table and ONE
    tensor to keep alive); slicing a view per image cost the host ~2.5 us each, 124 per step."""
    e = PACKED.get((W.data_ptr(), cin, cout, n_off, flags & _BACK))
    if e is None:
        return None
    if torch.is_tensor(e):
        return e, e.data_ptr()
    return e[0], e[3]


class PackPlan:
    """Host-side argument tables of one scn_conv_tiles_bf16_pack_many call for a fixed list of layers."""

    def __init__(self, jobs):
        self.jobs = list(jobs)
        self.n = len(self.jobs)
        self.weights = [j[0] for j in self.jobs]
        if self.n:
            self.rebuild()

    def rebuild(self):
        lib = L.lib()
        jobs, n = self.jobs, self.n
        self.ptrs = [W.data_ptr() for W in self.weights]
        self.sizes = [lib.scn_conv_tiles_bf16_image_bytes(ci, co, no) for (_, ci, co, no, _) in jobs]
        self.offs, tot = [], 0
        for b in self.sizes:
            self.offs.append(tot)
            tot += (b + 255) & ~255
        self.total = tot
        self.Wp = (C.c_void_p * n)(*self.ptrs)
        self.Ip = (C.c_void_p * n)()
        self.ci = (C.c_int32 * n)(*[j[1] for j in jobs]); self.co = (C.c_int32 * n)(*[j[2] for j in jobs])
        self.no = (C.c_int32 * n)(*[j[3] for j in jobs]); self.fl = (C.c_int32 * n)(*[j[4] & _BACK for j in jobs])
        self.keys = [(W.data_ptr(), cin, cout, n_off, flags & _BACK) for (W, cin, cout, n_off, flags) in jobs]


class packed_weights:
    """Context manager: pack the bf16 weight images (forward and backward-data) of `jobs` = [(W, cin, cout, n_off, flags)]
    with one scn_conv_tiles_bf16_pack_many launch; they are visible through packed_image() until the block ends."""

    def __init__(self, jobs):
        """jobs: a list, or a `PackPlan` (the host-side tables of a fixed list, built once and re-used every step)."""
        self.plan = jobs if isinstance(jobs, PackPlan) else PackPlan(jobs)
        self.keys = []

    def __enter__(self):
        pl = self.plan
        if not pl.n:
            return self
        if pl.ptrs != [W.data_ptr() for W in pl.weights]:          # a parameter moved (module.to(...), load): rebuild
            pl.rebuild()
        buf = torch.empty(pl.total, dtype=torch.uint8, device=pl.weights[0].device)
        base = buf.data_ptr()
        for i, o in enumerate(pl.offs):
            pl.Ip[i] = base + o
        L.check(L.lib().scn_conv_tiles_bf16_pack_many(pl.n, pl.Wp, pl.ci, pl.co, pl.no, pl.fl, pl.Ip, L.stream()))
        for key, o, b in zip(pl.keys, pl.offs, pl.sizes):
            PACKED[key] = (buf, o, b, base + o)           # (views are cut on demand: packed_image)
        self.keys = pl.keys
        return self

    def __exit__(self, *exc):
        for k in self.keys:
            PACKED.pop(k, None)
        return False


def conv_rules_bf16(X, tiles, n_out, W, bias, cout, flags=0, residual=None, relu_mask=None, image=None, n_rules=0):
    """scn_conv_tiles_bf16: the tile convolution for bf16-stored features (X, residual, relu_mask, result: torch.bfloat16;
    W, bias: the layer's fp32 parameters).  Forward and, with F_W_TRANSPOSED | F_OFF_REVERSE, backward-data.
    image: a packed weight image of W for these flags (pack_weights_bf16); None: packed here."""
    lib = L.lib()
    for t in (X, residual, relu_mask):
        if t is not None and (t.dtype != torch.bfloat16 or not t.is_contiguous()):
            raise L.ScnError("conv_rules_bf16 takes contiguous torch.bfloat16 features")
    if W.dtype != torch.float32 or (bias is not None and bias.dtype != torch.float32):
        raise L.ScnError("conv_rules_bf16 takes the fp32 master weights")
    cin = X.shape[1]
    if cin % 8:
        raise L.ScnError("bf16 storage needs input channel counts that are multiples of 8 (16-byte row pieces)")
    n_in, n_off = X.shape[0], tiles.n_off
    if image is None:
        image = packed_image(W, cin, cout, n_off, flags)
    if image is None:
        image = pack_weights_bf16(W, cin, cout, n_off, flags)
    Y = torch.empty((n_out, cout), dtype=torch.bfloat16, device=X.device)
    if getattr(tiles, "has_x", False):
        flags |= L.F_TILE_ORDER_X               # tile_order continues with the XCD-local hand-out order (scn_tiles_build_x)
    key = (cin, n_out, cout)
    sz = _tb_sizes.get(key)
    if sz is None:
        if len(_tb_sizes) > 4096:
            _tb_sizes.clear()
        sz = _tb_sizes[key] = (lib.scn_conv_tiles_bf16_scratch_bytes(cin, n_out, cout),
                               lib.scn_conv_tiles_bf16_arrival_counters(cin, n_out, cout))
    scratch = L.scratch(sz[0], X.device)
    arr = L.arrival(sz[1], X.device) if (sz[1] and FUSED_K) else None

    def run():
        L.check(lib.scn_conv_tiles_bf16(
            L.ptr(X), n_in, cin, L.ptr(tiles.tstab), L.ptr(tiles.tile_mask), L.ptr(tiles.perm),
            L.ptr(tiles.tile_order), n_off, n_out, L.ptr(image), L.ptr(bias), L.ptr(residual), L.ptr(relu_mask),
            L.ptr(Y), cout, flags, L.ptr(scratch), L.ptr(arr), L.stream()))
    if profiling.TIMER is None:
        run()
        return Y
    P = lambda: _count(n_rules)
    profiling.timed("k_conv_tb", lambda: 2.0 * P() * cin * cout,
                    lambda: 2.0 * (n_in * cin + n_out * cout + n_off * cin * cout) + 8.0 * P(), run)
    return Y


# The 3^3 / 2^3 table convolutions run on the tile kernel (scn_conv_tiles); USE_TILES = False routes them through the
# plain table GEMM (scn_gemm_table: no mask sorting, 2-3x wasted matrix work) -- kept as a cross-check of the two kernels.
USE_TILES = True


def gemm_rules(X, in_rows, out_rows, prefix_host, n_off, n_out, W, bias, cout, flags=0, relu_mask=None):
    lib = L.lib()
    Y = _new((n_out, cout), X, X.dtype)
    cin = X.shape[1]
    P = int(prefix_host[n_off] - prefix_host[0])
    entry = lib.scn_gemm_rules_bf16 if _is_bf16(X) else lib.scn_gemm_rules

    def run():
        # the entry point takes up to 32 offsets per call; every output row occurs in exactly one rule (the callers: Deconvolution
        # forward, Convolution backward-data), so the chunks of a larger filter (a 4^3 stride: 64) write disjoint rows of one Y
        for o0 in range(0, n_off, 32):
            k = min(32, n_off - o0)
            ph = _offset_chunk(prefix_host, o0)
            Wp = L.ptr(W) + 4 * o0 * W.shape[1] * W.shape[2] if o0 else L.ptr(W)
            L.check(entry(L.ptr(X), cin, L.ptr(in_rows), L.ptr(out_rows), ph, k, Wp,
                          L.ptr(bias), L.ptr(relu_mask), L.ptr(Y), cout, flags, L.stream()))
    profiling.timed("k_gemm_rules", 2.0 * P * cin * cout, _conv_bytes(X.shape[0], cin, n_out, cout, n_off, P), run)
    return Y


def _offset_chunk(prefix_host, o0):
    """`prefix_host` advanced by o0 offsets (the rule-list entry points take up to 32 offsets per call; rule rows are addressed
    through the absolute prefix, so a chunk is the same arrays with a later prefix pointer)."""
    return prefix_host if o0 == 0 else C.cast(C.addressof(prefix_host.contents) + 8 * o0, C.POINTER(C.c_int64))


def wgrad_rules_bf16(X, dY, in_rows, out_rows, prefix_host, n_off, flags=0):
    """scn_wgrad_rules_bf16: dW (fp32) from bf16-stored X and dY -- the weight gradient of the bf16 storage path."""
    if n_off > 32:          # (a 4^3 stride, a 5^3 filter: chunks of 32 offsets)
        return torch.cat([wgrad_rules_bf16(X, dY, in_rows, out_rows, _offset_chunk(prefix_host, o0), min(32, n_off - o0), flags)
                          for o0 in range(0, n_off, 32)], 0)
    lib = L.lib()
    for t in (X, dY):
        if t.dtype != torch.bfloat16 or not t.is_contiguous():
            raise L.ScnError("wgrad_rules_bf16 takes contiguous torch.bfloat16 operands")
    cin, cout = X.shape[1], dY.shape[1]
    nbytes = lib.scn_wgrad_scratch_bytes(cin, cout, prefix_host, n_off)
    if nbytes < 0:
        raise L.ScnError("scn_wgrad_scratch_bytes: bad arguments")
    scratch = L.scratch(nbytes, X.device)
    dW = torch.empty((n_off, cin, cout), dtype=torch.float32, device=X.device)
    profiling.timed("k_wgrad_rules_bf16", 0.0, 0.0, lambda: L.check(lib.scn_wgrad_rules_bf16(
        L.ptr(X), cin, L.ptr(dY), cout, L.ptr(in_rows), L.ptr(out_rows), prefix_host, n_off, L.ptr(dW), L.ptr(scratch),
        flags, L.stream())))
    return dW


def wgrad_rules(X, dY, in_rows, out_rows, prefix_host, n_off, flags=0):
    if _is_bf16(X):
        return wgrad_rules_bf16(X, dY, in_rows, out_rows, prefix_host, n_off, flags)
    if n_off > 32:
        return torch.cat([wgrad_rules(X, dY, in_rows, out_rows, _offset_chunk(prefix_host, o0), min(32, n_off - o0), flags)
                          for o0 in range(0, n_off, 32)], 0)
    lib = L.lib()
    cin, cout = X.shape[1], dY.shape[1]
    nbytes = lib.scn_wgrad_scratch_bytes(cin, cout, prefix_host, n_off)
    if nbytes < 0:
        raise L.ScnError("scn_wgrad_scratch_bytes: bad arguments")
    scratch = L.scratch(nbytes, X.device)
    dW = _new((n_off, cin, cout), X)
    P = int(prefix_host[n_off] - prefix_host[0])

    def run():
        L.check(lib.scn_wgrad_rules(L.ptr(X), cin, L.ptr(dY), cout, L.ptr(in_rows), L.ptr(out_rows), prefix_host,
                                    n_off, L.ptr(dW), L.ptr(scratch), flags, L.stream()))
    profiling.timed("k_wgrad_rules", 2.0 * P * cin * cout,
                    4.0 * (X.shape[0] * cin + dY.shape[0] * cout + n_off * cin * cout) + 8.0 * P, run)
    return dW


def wgrad_bias_rules(X, dY, in_rows, out_rows, prefix_host, n_off, db_offsets, flags=0):
    """dW and db in one pass (scn_wgrad_bias_rules; bf16-stored operands: scn_wgrad_bias_rules_bf16, fp32 results)."""
    cin, cout = X.shape[1], dY.shape[1]
    lib = L.lib()
    hb = _is_bf16(X)
    if hb and (dY.dtype != torch.bfloat16 or not (X.is_contiguous() and dY.is_contiguous())):
        raise L.ScnError("wgrad_bias_rules takes contiguous operands of one storage type")
    nbytes = lib.scn_wgrad_scratch_bytes(cin, cout, prefix_host, n_off)
    scratch = L.scratch(nbytes, X.device)
    dW = torch.empty((n_off, cin, cout), dtype=torch.float32, device=X.device)
    db = torch.empty((cout,), dtype=torch.float32, device=X.device)
    P = int(prefix_host[n_off] - prefix_host[0])
    entry = lib.scn_wgrad_bias_rules_bf16 if hb else lib.scn_wgrad_bias_rules
    es = 2.0 if hb else 4.0

    def run():
        L.check(entry(L.ptr(X), cin, L.ptr(dY), cout, L.ptr(in_rows), L.ptr(out_rows), prefix_host,
                      n_off, L.ptr(dW), L.ptr(db), db_offsets, L.ptr(scratch), flags, L.stream()))
    profiling.timed("k_wgrad_rules_bf16" if hb else "k_wgrad_rules", 2.0 * P * cin * cout,
                    es * (X.shape[0] * cin + dY.shape[0] * cout) + 4.0 * n_off * cin * cout + 8.0 * P, run)
    return dW, db


# residual units: both weight gradients in one launch (False / SCN_WGRAD_PAIR=0: one call each -- cross-check, A/B)
WGRAD_PAIR = os.environ.get("SCN_WGRAD_PAIR", "1") != "0"


def wgrad_bias_rules_n(Xs, dYs, in_rows, out_rows, prefix_host, n_off, db_offsets, flags=0):
    """The weight (and, db_offsets != 0, bias) gradients of 2 ... 4 convolutions that share a rule list and their channel
    counts -- the two of a residual unit, the four of two stacked units -- in one launch + one sum
    (scn_wgrad_bias_rules_n / _bf16): operand pairs (Xs[p], dYs[p]) of one storage type.
    -> fp32 dW [n_prob, n_off, cin, cout], db [n_prob, cout] or None."""
    lib = L.lib()
    n_prob = len(Xs)
    cin, cout = Xs[0].shape[1], dYs[0].shape[1]
    hb = _is_bf16(Xs[0])
    if any(_is_bf16(t) != hb or not t.is_contiguous() for t in tuple(Xs) + tuple(dYs)):
        raise L.ScnError("wgrad_bias_rules_n takes contiguous operands of one storage type")
    entry = lib.scn_wgrad_bias_rules_n_bf16 if hb else lib.scn_wgrad_bias_rules_n
    es = 2.0 if hb else 4.0
    nbytes = lib.scn_wgrad_scratch_bytes_n(cin, cout, prefix_host, n_off, n_prob)
    if nbytes < 0:
        raise L.ScnError("scn_wgrad_scratch_bytes_n: bad arguments")
    scratch = L.scratch(nbytes, Xs[0].device)
    dW = torch.empty((n_prob, n_off, cin, cout), dtype=torch.float32, device=Xs[0].device)
    db = torch.empty((n_prob, cout), dtype=torch.float32, device=Xs[0].device) if db_offsets else None
    P = int(prefix_host[n_off] - prefix_host[0])
    xp = (C.c_void_p * n_prob)(*[t.data_ptr() for t in Xs])
    yp = (C.c_void_p * n_prob)(*[t.data_ptr() for t in dYs])

    def run():
        L.check(entry(xp, yp, n_prob, cin, cout, L.ptr(in_rows), L.ptr(out_rows), prefix_host, n_off, L.ptr(dW), L.ptr(db),
                      db_offsets, L.ptr(scratch), flags, L.stream()))
    profiling.timed("k_wgrad_rules_bf16" if hb else "k_wgrad_rules", 2.0 * n_prob * P * cin * cout,
                    n_prob * (es * (Xs[0].shape[0] * cin + dYs[0].shape[0] * cout) + 4.0 * n_off * cin * cout + 8.0 * P), run)
    return dW, db


def wgrad_bias_rules2(X0, dY0, X1, dY1, in_rows, out_rows, prefix_host, n_off, db_offsets, flags=0):
    return wgrad_bias_rules_n((X0, X1), (dY0, dY1), in_rows, out_rows, prefix_host, n_off, db_offsets, flags)


def colsum(dY):
    """Bias gradient db[c] = sum_r dY[r][c] (fp32 result; dY fp32 or bf16-stored), two-stage, fixed order."""
    lib = L.lib()
    c = dY.shape[1]
    scratch = torch.empty(L.COLSUM_BLOCKS * c, dtype=torch.float32, device=dY.device)
    db = torch.empty(c, dtype=torch.float32, device=dY.device)
    entry = lib.scn_colsum_bf16 if _is_bf16(dY) else lib.scn_colsum
    L.check(entry(L.ptr(dY), dY.shape[0], c, L.ptr(db), L.ptr(scratch), L.stream()))
    return db


def _on_leaf_stream(dY, fn):
    """Weight / bias gradients are leaves of the backward graph.  Queuing them on a second HIP stream (so that their
    workgroups fill the CUs the backward-data chain leaves idle) was measured in round 1: no gain, and the bucketed
    all-reduce hooks of dp.py would have needed an extra stream join -- removed; leaves run on the main stream."""
    return fn()


def _identity_prefix(n):
    h = L.host_i64(2)
    h[0], h[1] = 0, n
    return h


# ------------------------------------------------------------------------------------------------------
# A5 SubmanifoldConvolution  (module_factory.py:383-385 k=1, 404-406 k=3)
# ------------------------------------------------------------------------------------------------------
class SubmanifoldConvolutionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, weight, bias, metadata: Metadata, spatial_size, k, relu_in=False, residual=None):
        """residual (optional, [N, Cout]): added in the kernel epilogue -- the AddTable of a residual block fused
        (module_factory.py:51-57: Sequential(ConcatTable(Identity, inner), AddTable))."""
        X, W = _feat(features), _f32(weight)
        rb = metadata.subm_rulebook(spatial_size, k)
        n_off = k ** 3
        cout = W.shape[-1]
        b = _f32(bias) if bias is not None else None
        R = _feat(residual) if residual is not None else None
        if relu_in:
            _rec_relu(X)
        if USE_TILES and rb.rules is not None and rb.tiles is not None:
            Y = conv_rules(X, rb.tiles, rb.n, W, b, cout, L.F_RELU_IN if relu_in else 0, residual=R,
                           n_rules=rb.rules.count)
        else:
            Y = gemm_table(X, rb.table, n_off, rb.n, W, b, cout, L.F_RELU_IN if relu_in else 0, residual=R,
                           n_rules=rb.rules.count if rb.rules is not None else rb.n)
        ctx.save_for_backward(X, W)
        ctx.rb, ctx.has_bias, ctx.relu_in = rb, bias is not None, relu_in
        ctx.bwd_image = packed_image(W, cout, X.shape[1], n_off, _BACK) if _is_bf16(X) else None
        return Y

    @staticmethod
    def backward(ctx, dY):
        X, W = ctx.saved_tensors
        rb = ctx.rb
        dY = _feat(dY)
        n_off = rb.k ** 3
        cin = X.shape[1]
        fl = L.F_RELU_IN if ctx.relu_in else 0
        dX = dW = db = None
        if ctx.needs_input_grad[0]:
            if USE_TILES and rb.rules is not None and rb.tiles is not None:
                dX = conv_rules(dY, rb.tiles, rb.n, W, None, cin, L.F_W_TRANSPOSED | L.F_OFF_REVERSE,
                                relu_mask=X if ctx.relu_in else None, n_rules=rb.rules.count, image=ctx.bwd_image)
            else:
                dX = gemm_table(dY, rb.table, n_off, rb.n, W, None, cin, L.F_W_TRANSPOSED | L.F_OFF_REVERSE,
                                relu_mask=X if ctx.relu_in else None,
                                n_rules=rb.rules.count if rb.rules is not None else rb.n)
        def leaves():
            dW = db = None
            want_b = ctx.has_bias and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                ir, orr = (None, None) if rb.k == 1 else (rb.rules.in_rows, rb.rules.out_rows)
                ph = _identity_prefix(rb.n) if rb.k == 1 else rb.rules.prefix_host
                if want_b and n_off <= 32:      # the centre offset lists every row once: bias gradient from the same pass
                    dW, db = wgrad_bias_rules(X, dY, ir, orr, ph, n_off, 1 << (n_off // 2), fl)
                else:
                    dW = wgrad_rules(X, dY, ir, orr, ph, n_off, fl)
                    if want_b:
                        db = colsum(dY)
                dW = dW.view_as(W)
            elif want_b:
                db = colsum(dY)
            return dW, db
        dW, db = _on_leaf_stream(dY, leaves)
        dR = dY if (len(ctx.needs_input_grad) > 7 and ctx.needs_input_grad[7]) else None
        return dX, dW, db, None, None, None, None, dR


# ------------------------------------------------------------------------------------------------------
# Residual block  x + SubM3(ReLU(SubM3(ReLU(x))))  as ONE autograd node  (module_factory.py:127-183 get_residual_block
# with relu_first, two units; custom_container.py: Sequential(ConcatTable(Identity, inner), AddTable))
# ------------------------------------------------------------------------------------------------------
class ResidualBlockFunction(torch.autograd.Function):
    """Same kernels as the layer-by-layer path; what the fusion removes is the glue: the AddTable runs in the epilogue
    of the second convolution (forward) and the sum of the two gradient paths of x in the epilogue of the first
    convolution's backward-data kernel (SCN_F_RESIDUAL_LAST: dX = mask(conv1^T dY1) + dY) -- no elementwise launches, one
    autograd node instead of two."""

    @staticmethod
    def forward(ctx, features, w1, b1, w2, b2, metadata: Metadata, spatial_size):
        X, W1, W2 = _f32(features), _f32(w1), _f32(w2)
        rb = metadata.subm_rulebook(spatial_size, 3)
        B1 = _f32(b1) if b1 is not None else None
        B2 = _f32(b2) if b2 is not None else None
        Y1 = conv_rules(X, rb.tiles, rb.n, W1, B1, W1.shape[-1], L.F_RELU_IN, n_rules=rb.rules.count)
        Y = conv_rules(Y1, rb.tiles, rb.n, W2, B2, W2.shape[-1], L.F_RELU_IN, residual=X, n_rules=rb.rules.count)
        _rec_relu(X); _rec_relu(Y1)
        ctx.save_for_backward(X, Y1, W1, W2)
        ctx.rb, ctx.has_b1, ctx.has_b2 = rb, b1 is not None, b2 is not None
        return Y

    @staticmethod
    def backward(ctx, dY):
        X, Y1, W1, W2 = ctx.saved_tensors
        rb, r = ctx.rb, ctx.rb.rules
        dY = _f32(dY)
        need = ctx.needs_input_grad
        back = L.F_W_TRANSPOSED | L.F_OFF_REVERSE
        dY1 = conv_rules(dY, rb.tiles, rb.n, W2, None, W2.shape[1], back, relu_mask=Y1, n_rules=r.count)
        centre = 1 << 13

        def wgrad(Xin, G, W, want_w, want_b):
            if want_w and want_b:
                dW, db = wgrad_bias_rules(Xin, G, r.in_rows, r.out_rows, r.prefix_host, 27, centre, L.F_RELU_IN)
                return dW.view_as(W), db
            if want_w:
                return wgrad_rules(Xin, G, r.in_rows, r.out_rows, r.prefix_host, 27, L.F_RELU_IN).view_as(W), None
            return None, (colsum(G) if want_b else None)
        want_b1, want_b2 = ctx.has_b1 and need[2], ctx.has_b2 and need[4]
        # both weight gradients in one launch (same rule list, same channel counts; bias gradients both or neither)
        pair = (WGRAD_PAIR and need[1] and need[3] and want_b1 == want_b2 and W1.shape == W2.shape
                and W1.shape[1] == W1.shape[2] and r.prefix_host[0] == 0 and X.is_contiguous() and Y1.is_contiguous()
                and dY.is_contiguous() and dY1.is_contiguous())
        if not pair:
            dW2, db2 = _on_leaf_stream(dY, lambda: wgrad(Y1, dY, W2, need[3], want_b2))
        dX = None
        if need[0]:
            dX = conv_rules(dY1, rb.tiles, rb.n, W1, None, W1.shape[1], back | L.F_RESIDUAL_LAST, relu_mask=X,
                            residual=dY, n_rules=r.count)
        if pair:
            dWp, dbp = wgrad_bias_rules2(X, dY1, Y1, dY, r.in_rows, r.out_rows, r.prefix_host, 27,
                                         centre if want_b1 else 0, L.F_RELU_IN)
            dW1, dW2 = dWp[0].view_as(W1), dWp[1].view_as(W2)
            db1, db2 = (dbp[0], dbp[1]) if dbp is not None else (None, None)
        else:
            dW1, db1 = _on_leaf_stream(dY1, lambda: wgrad(X, dY1, W1, need[1], want_b1))
        return dX, dW1, db1, dW2, db2, None, None


class ResidualBlockFunctionBF16(torch.autograd.Function):
    """ResidualBlockFunction for bf16-STORED features (BASELINE configs 3-5): x, the intermediate h, the
    result and every gradient tensor of the block are torch.bfloat16; the parameters stay fp32 (rounded to bf16 as
    they are staged into LDS) and their gradients come back fp32.  Six launches: scn_conv_tiles_bf16 x 4 (two forward,
    two backward-data with the ReLU mask / the skip gradient in the epilogue) and scn_wgrad_rules_bf16 x 2."""

    @staticmethod
    def forward(ctx, features, w1, b1, w2, b2, metadata: Metadata, spatial_size):
        if features.dtype != torch.bfloat16:
            raise L.ScnError("ResidualBlockFunctionBF16 takes torch.bfloat16 features")
        if features.shape[1] % 8 != 0 or w1.shape[-1] % 8 != 0:
            raise L.ScnError("bf16 storage needs channel counts that are multiples of 8 (16-byte row pieces)")
        X = features.contiguous()
        W1, W2 = _f32(w1), _f32(w2)
        rb = metadata.subm_rulebook(spatial_size, 3)
        B1 = _f32(b1) if b1 is not None else None
        B2 = _f32(b2) if b2 is not None else None
        Y1 = conv_rules_bf16(X, rb.tiles, rb.n, W1, B1, W1.shape[-1], L.F_RELU_IN, n_rules=rb.rules.count)
        Y = conv_rules_bf16(Y1, rb.tiles, rb.n, W2, B2, W2.shape[-1], L.F_RELU_IN, residual=X, n_rules=rb.rules.count)
        _rec_relu(X); _rec_relu(Y1)
        ctx.save_for_backward(X, Y1, W1, W2)
        ctx.rb, ctx.has_b1, ctx.has_b2 = rb, b1 is not None, b2 is not None
        ctx.bwd_images = (packed_image(W1, W1.shape[-1], W1.shape[1], 27, _BACK),
                          packed_image(W2, W2.shape[-1], W2.shape[1], 27, _BACK))
        return Y

    @staticmethod
    def backward(ctx, dY):
        X, Y1, W1, W2 = ctx.saved_tensors
        rb, r = ctx.rb, ctx.rb.rules
        dY = dY.to(torch.bfloat16).contiguous()
        need = ctx.needs_input_grad
        back = L.F_W_TRANSPOSED | L.F_OFF_REVERSE
        dY1 = conv_rules_bf16(dY, rb.tiles, rb.n, W2, None, W2.shape[1], back, relu_mask=Y1, image=ctx.bwd_images[1],
                              n_rules=r.count)

        def wgrad(Xin, G, W, want_w, want_b):
            if want_w and want_b:        # the centre offset lists every row once: bias gradient from the same pass
                dW, db = wgrad_bias_rules(Xin, G, r.in_rows, r.out_rows, r.prefix_host, 27, 1 << 13, L.F_RELU_IN)
                return dW.view_as(W), db
            dW = wgrad_rules_bf16(Xin, G, r.in_rows, r.out_rows, r.prefix_host, 27, L.F_RELU_IN).view_as(W) \
                if want_w else None
            return dW, (colsum(G) if want_b else None)
        want_b1, want_b2 = ctx.has_b1 and need[2], ctx.has_b2 and need[4]
        pair = (WGRAD_PAIR and need[1] and need[3] and want_b1 == want_b2 and W1.shape == W2.shape
                and W1.shape[1] == W1.shape[2] and r.prefix_host[0] == 0 and dY1.is_contiguous())
        if not pair:
            dW2, db2 = wgrad(Y1, dY, W2, need[3], want_b2)
        dX = None
        if need[0]:
            dX = conv_rules_bf16(dY1, rb.tiles, rb.n, W1, None, W1.shape[1], back | L.F_RESIDUAL_LAST, relu_mask=X,
                                 residual=dY, image=ctx.bwd_images[0], n_rules=r.count)
        if pair:            # both weight gradients in one launch (see ResidualBlockFunction)
            dWp, dbp = wgrad_bias_rules2(X, dY1, Y1, dY, r.in_rows, r.out_rows, r.prefix_host, 27,
                                         (1 << 13) if want_b1 else 0, L.F_RELU_IN)
            dW1, dW2 = dWp[0].view_as(W1), dWp[1].view_as(W2)
            db1, db2 = (dbp[0], dbp[1]) if dbp is not None else (None, None)
        else:
            dW1, db1 = wgrad(X, dY1, W1, need[1], want_b1)
        return dX, dW1, db1, dW2, db2, None, None


# ------------------------------------------------------------------------------------------------------
# A6 Convolution size=stride=2  (module_factory.py:232-234)
# ------------------------------------------------------------------------------------------------------
class ConvolutionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, weight, bias, metadata: Metadata, in_size, relu_in=False, stride=(2, 2, 2)):
        X, W = _feat(features), _f32(weight)
        rb = metadata.strided_rulebook(in_size, stride)
        b = _f32(bias) if bias is not None else None
        if relu_in:
            _rec_relu(X)
        if USE_TILES and rb.tiles is not None:
            Y = conv_rules(X, rb.tiles, rb.n_coarse, W, b, W.shape[-1], L.F_RELU_IN if relu_in else 0,
                           n_rules=rb.n_fine)
        else:                   # (more than 27 offsets -- a 4^3 filter -- : the table-walk GEMM; fp32 rows)
            Y = gemm_table(X, rb.child, rb.n_off, rb.n_coarse, W, b, W.shape[-1], L.F_RELU_IN if relu_in else 0,
                           n_rules=rb.n_fine)
        ctx.save_for_backward(X, W)
        ctx.rb, ctx.has_bias, ctx.relu_in = rb, bias is not None, relu_in
        return Y

    @staticmethod
    def backward(ctx, dY):
        X, W = ctx.saved_tensors
        rb, r = ctx.rb, ctx.rb.rules
        dY = _feat(dY)
        fl = L.F_RELU_IN if ctx.relu_in else 0
        dX = dW = db = None
        if ctx.needs_input_grad[0]:      # dX[f] = dY[parent f] . W[off f]^T : rule list with roles swapped
            dX = gemm_rules(dY, r.out_rows, r.in_rows, r.prefix_host, rb.n_off, rb.n_fine, W, None, X.shape[1],
                            L.F_W_TRANSPOSED, relu_mask=X if ctx.relu_in else None)
        def leaves():
            dW = db = None
            if ctx.needs_input_grad[1]:
                dW = wgrad_rules(X, dY, r.in_rows, r.out_rows, r.prefix_host, rb.n_off, fl).view_as(W)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = colsum(dY)
            return dW, db
        dW, db = _on_leaf_stream(dY, leaves)
        return dX, dW, db, None, None, None, None


# ------------------------------------------------------------------------------------------------------
# A7 Deconvolution size=stride back to the cached fine level  (module_factory.py:256-258)
# ------------------------------------------------------------------------------------------------------
class DeconvolutionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, weight, bias, metadata: Metadata, out_size, relu_in=False, stride=(2, 2, 2)):
        X, W = _feat(features), _f32(weight)
        out_size = tuple(int(s) for s in out_size)
        rb = metadata.cached_strided_rulebook(out_size, stride)
        if rb is None:
            raise L.ScnError(f"Deconvolution: no cached Convolution rulebook from spatial size {out_size}; the "
                             "reference only deconvolves back to an encoder level (custom_container.py:70-83)")
        r = rb.rules
        b = _f32(bias) if bias is not None else None
        if relu_in:
            _rec_relu(X)
        Y = gemm_rules(X, r.out_rows, r.in_rows, r.prefix_host, rb.n_off, rb.n_fine, W, b, W.shape[-1],
                       L.F_RELU_IN if relu_in else 0)
        ctx.save_for_backward(X, W)
        ctx.rb, ctx.has_bias, ctx.relu_in = rb, bias is not None, relu_in
        ctx.bwd_image = packed_image(W, W.shape[-1], X.shape[1], rb.n_off, L.F_W_TRANSPOSED) if _is_bf16(X) else None
        return Y

    @staticmethod
    def backward(ctx, dY):
        X, W = ctx.saved_tensors
        rb, r = ctx.rb, ctx.rb.rules
        dY = _feat(dY)
        fl = L.F_RELU_IN if ctx.relu_in else 0
        dX = dW = db = None
        if ctx.needs_input_grad[0]:      # dX[c] = sum_o dY[child[o][c]] . W[o]^T
            if USE_TILES and rb.tiles is not None:
                dX = conv_rules(dY, rb.tiles, rb.n_coarse, W, None, X.shape[1], L.F_W_TRANSPOSED,
                                relu_mask=X if ctx.relu_in else None, n_rules=rb.n_fine, image=ctx.bwd_image)
            else:
                dX = gemm_table(dY, rb.child, rb.n_off, rb.n_coarse, W, None, X.shape[1], L.F_W_TRANSPOSED,
                                relu_mask=X if ctx.relu_in else None, n_rules=rb.n_fine)
        def leaves():
            dW = db = None
            want_b = ctx.has_bias and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                if want_b and rb.n_off <= 27:      # every fine (output) row occurs in exactly one of the rule lists
                    dW, db = wgrad_bias_rules(X, dY, r.out_rows, r.in_rows, r.prefix_host, rb.n_off, (1 << rb.n_off) - 1, fl)
                else:
                    dW = wgrad_rules(X, dY, r.out_rows, r.in_rows, r.prefix_host, rb.n_off, fl)
                    if want_b:
                        db = colsum(dY)
                dW = dW.view_as(W)
            elif want_b:
                db = colsum(dY)
            return dW, db
        dW, db = _on_leaf_stream(dY, leaves)
        return dX, dW, db, None, None, None, None


# ------------------------------------------------------------------------------------------------------
# A9 NetworkInNetwork (module_factory.py:366-367), ReLU (:88), AddTable (:52-54,308)
# ------------------------------------------------------------------------------------------------------
class NetworkInNetworkFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, weight, bias):
        X, W = _feat(features), _f32(weight)
        b = _f32(bias) if bias is not None else None
        n = X.shape[0]
        Y = gemm_table(X, None, 1, n, W, b, W.shape[-1])
        ctx.save_for_backward(X, W)
        ctx.has_bias = bias is not None
        return Y

    @staticmethod
    def backward(ctx, dY):
        X, W = ctx.saved_tensors
        dY = _feat(dY)
        n = X.shape[0]
        dX = dW = db = None
        if ctx.needs_input_grad[0]:
            dX = gemm_table(dY, None, 1, n, W, None, X.shape[1], L.F_W_TRANSPOSED)
        def leaves():
            dW = db = None
            want_b = ctx.has_bias and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1]:
                if want_b:
                    dW, db = wgrad_bias_rules(X, dY, None, None, _identity_prefix(n), 1, 1)
                else:
                    dW = wgrad_rules(X, dY, None, None, _identity_prefix(n), 1)
                dW = dW.view_as(W)
            elif want_b:
                db = colsum(dY)
            return dW, db
        dW, db = _on_leaf_stream(dY, leaves)
        return dX, dW, db


FUSED_JOIN = True   # NetworkInNetwork over two JoinTable parts: one two-source launch (False: one GEMM per part, chained)


def gemm_rows2(X0, X1, W, bias, cout, flags=0):
    """scn_gemm_rows2 with two sources: Y = bias + [X0 | X1] . W (W is the layer's [c0 + c1][cout] weight)."""
    n, c0, c1 = X0.shape[0], X0.shape[1], X1.shape[1]
    Y = _new((n, cout), X0, X0.dtype)

    def run():
        L.check(L.lib().scn_gemm_rows2(L.ptr(X0), c0, L.ptr(X1), c1, n, L.ptr(W), L.ptr(bias), 0, 0, L.ptr(Y), cout, 0, 0,
                                       flags, int(_is_bf16(X0)), L.stream()))
    eb = X0.element_size()
    profiling.timed("k_gemm_table", 2.0 * n * (c0 + c1) * cout, eb * n * (c0 + c1 + cout) + 4.0 * (c0 + c1) * cout, run)
    return Y


def gemm_rows2_bwd(dY, W, c0, c1):
    """scn_gemm_rows2 with two destinations: dX0 = dY . W[:c0]^T, dX1 = dY . W[c0:]^T from one read of dY."""
    n, cout = dY.shape
    d0, d1 = _new((n, c0), dY, dY.dtype), _new((n, c1), dY, dY.dtype)

    def run():
        L.check(L.lib().scn_gemm_rows2(L.ptr(dY), cout, 0, 0, n, L.ptr(W), 0, 0, 0, L.ptr(d0), c0, L.ptr(d1), c1,
                                       L.F_W_TRANSPOSED, int(_is_bf16(dY)), L.stream()))
    eb = dY.element_size()
    profiling.timed("k_gemm_table", 2.0 * n * (c0 + c1) * cout, eb * n * (c0 + c1 + cout) + 4.0 * (c0 + c1) * cout, run)
    return d0, d1


class JoinedNetworkInNetworkFunction(torch.autograd.Function):
    """NetworkInNetwork over a JoinTable without the concatenated slab: Y = b + sum_k X_k . W[rows of part k].
    Two parts in 8-channel groups (every decoder level): ONE launch reads both slabs (scn_gemm_rows2) and one launch writes
    both gradients.  Otherwise one identity-table GEMM per part, the running sum carried through the kernel's residual
    operand, and dX_k = dY . W[rows k]^T per part.  dW[rows k] = X_k^T dY (db with the first)."""

    @staticmethod
    def forward(ctx, weight, bias, *parts):
        W = _f32(weight)
        b = _f32(bias) if bias is not None else None
        Xs = [_feat(p) for p in parts]
        n, cout = Xs[0].shape[0], W.shape[-1]
        if sum(X.shape[1] for X in Xs) != W.shape[0]:
            raise L.ScnError("JoinTable parts do not add up to the NetworkInNetwork's input width")
        c0, c1 = (Xs[0].shape[1], Xs[1].shape[1]) if len(Xs) == 2 else (0, 0)
        ctx.fused = (FUSED_JOIN and len(Xs) == 2 and Xs[0].dtype == Xs[1].dtype and c0 >= 8 and c1 >= 8 and c0 % 8 == 0
                     and c1 % 8 == 0 and n > 0)
        if ctx.fused:
            y = gemm_rows2(Xs[0], Xs[1], W, b, cout)
        else:
            y, r0 = None, 0
            for k, X in enumerate(Xs):
                c = X.shape[1]
                y = gemm_table(X, None, 1, n, W[r0:r0 + c], b if k == 0 else None, cout, residual=y)
                r0 += c
        ctx.save_for_backward(W, *Xs)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dY):
        W, *Xs = ctx.saved_tensors
        dY = _feat(dY)
        n = dY.shape[0]
        need = ctx.needs_input_grad
        dW = torch.empty_like(W) if need[0] else None
        db, dXs, r0 = None, [], 0
        if ctx.fused and need[2] and need[3] and dY.shape[1] % 8 == 0:       # (the kernel stages dY in 8-channel groups)
            dXs = list(gemm_rows2_bwd(dY, W, Xs[0].shape[1], Xs[1].shape[1]))
        for k, X in enumerate(Xs):
            c = X.shape[1]
            if len(dXs) <= k:
                dXs.append(gemm_table(dY, None, 1, n, W[r0:r0 + c], None, c, L.F_W_TRANSPOSED) if need[2 + k] else None)
            if need[0]:
                if k == 0 and ctx.has_bias and need[1]:
                    dWk, db = wgrad_bias_rules(X, dY, None, None, _identity_prefix(n), 1, 1)
                else:
                    dWk = wgrad_rules(X, dY, None, None, _identity_prefix(n), 1)
                dW[r0:r0 + c] = dWk.view(c, -1)
            r0 += c
        if db is None and ctx.has_bias and need[1]:
            db = colsum(dY)
        return (dW, db, *dXs)


class ReLUFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features):
        X = _f32(features)
        _rec_relu(X)
        Y = torch.empty_like(X)
        L.check(L.lib().scn_relu_fwd(L.ptr(X), X.numel(), L.ptr(Y), L.stream()))
        ctx.save_for_backward(X)
        return Y

    @staticmethod
    def backward(ctx, dY):
        (X,) = ctx.saved_tensors
        dY = _f32(dY)
        dX = torch.empty_like(X)
        L.check(L.lib().scn_relu_bwd(L.ptr(X), L.ptr(dY), X.numel(), L.ptr(dX), L.stream()))
        return dX


class AddFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        A, B = _f32(a), _f32(b)
        if A.shape != B.shape:
            raise ValueError("AddTable: shape mismatch")
        Y = torch.empty_like(A)
        L.check(L.lib().scn_add(L.ptr(A), L.ptr(B), A.numel(), L.ptr(Y), L.stream()))
        return Y

    @staticmethod
    def backward(ctx, dY):
        return dY, dY


# ------------------------------------------------------------------------------------------------------
# A8 BatchNorm(Leaky)ReLU (module_factory.py:92-102)
# ------------------------------------------------------------------------------------------------------
def _sync_group(sync):
    """The process group SyncBN reduces over, or None (single process / not requested)."""
    import torch.distributed as dist
    if not sync or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return None
    return dist.group.WORLD


class BatchNormReLUFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, weight, bias, running_mean, running_var, eps, momentum, leak, training, sync=False):
        lib = L.lib()
        X = _f32(features)
        n, c = X.shape
        g, b = _f32(weight), _f32(bias)
        scratch = torch.empty(lib.scn_bn_scratch_bytes(c), dtype=torch.uint8, device=X.device)
        group = _sync_group(sync) if training else None
        n_stat = n
        if group is not None:
            # batch statistics over the scenes of ALL ranks (module_factory.py:92-102: one feature matrix per batch):
            # one all-reduce of (sum x, sum x^2, rows) per layer, float64
            import torch.distributed as dist
            pack = torch.empty(2 * c + 1, dtype=torch.float64, device=X.device)
            L.check(lib.scn_bn_sums(L.ptr(X), n, c, L.ptr(pack), L.ptr(scratch), L.stream()))
            pack[2 * c] = float(n)
            dist.all_reduce(pack, group=group)
            n_stat = int(round(pack[2 * c].item()))
            mu = pack[:c] / max(n_stat, 1)
            mean = mu.float()
            var = (pack[c:2 * c] / max(n_stat, 1) - mu * mu).clamp_(min=0).float()
            running_mean.mul_(momentum).add_(mean, alpha=1 - momentum)
            running_var.mul_(momentum).add_(var, alpha=(1 - momentum) * (n_stat / (n_stat - 1) if n_stat > 1 else 1.0))
        elif training:
            mean, var = _new((c,), X), _new((c,), X)
            L.check(lib.scn_bn_stats(L.ptr(X), n, c, L.ptr(mean), L.ptr(var), L.ptr(scratch), L.stream()))
            # momentum is the RETAIN fraction (SparseConvNet convention; SURVEY.md §4.1 caveat).  The running variance
            # takes the UNBIASED batch estimate (sum of squared differences / (n - 1)), the normalisation the biased
            # one -- SparseConvNet's BatchNormalization forward as recalled [UPSTREAM-SCN], and torch.nn.BatchNorm's
            # convention too; n = 1 keeps the (zero) biased value instead of dividing by zero
            running_mean.mul_(momentum).add_(mean, alpha=1 - momentum)
            running_var.mul_(momentum).add_(var, alpha=(1 - momentum) * (n / (n - 1) if n > 1 else 1.0))
        else:
            mean, var = _f32(running_mean), _f32(running_var)
        Y = torch.empty_like(X)
        L.check(lib.scn_bn_fwd(L.ptr(X), n, c, L.ptr(mean), L.ptr(var), eps, L.ptr(g), L.ptr(b), leak, L.ptr(Y),
                               L.stream()))
        ctx.save_for_backward(X, g, b, mean, var)
        ctx.cfg = (eps, leak, training)
        ctx.sync = (group, n_stat)
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = L.lib()
        X, g, b, mean, var = ctx.saved_tensors
        eps, leak, training = ctx.cfg
        dY = _f32(dY)
        n, c = X.shape
        scratch = torch.empty(lib.scn_bn_scratch_bytes(c), dtype=torch.uint8, device=X.device)
        dX, dg, db = torch.empty_like(X), _new((c,), X), _new((c,), X)
        group, n_stat = ctx.sync
        if group is not None:
            # (sum g, sum g x^) over all ranks; dgamma / dbeta stay this rank's share (the gradient all-reduce adds them)
            import torch.distributed as dist
            sums = torch.empty(2 * c, dtype=torch.float64, device=X.device)
            L.check(lib.scn_bn_bwd_reduce(L.ptr(X), L.ptr(dY), n, c, L.ptr(mean), L.ptr(var), eps, L.ptr(g), L.ptr(b), leak,
                                          L.ptr(dg), L.ptr(db), L.ptr(sums), L.ptr(scratch), L.stream()))
            dist.all_reduce(sums, group=group)
            L.check(lib.scn_bn_bwd_apply(L.ptr(X), L.ptr(dY), n, c, L.ptr(mean), L.ptr(var), eps, L.ptr(g), L.ptr(b), leak,
                                         L.ptr(sums), n_stat, L.ptr(dX), L.stream()))
            return dX, dg, db, None, None, None, None, None, None, None
        L.check(lib.scn_bn_bwd(L.ptr(X), L.ptr(dY), n, c, L.ptr(mean), L.ptr(var), eps, L.ptr(g), L.ptr(b), leak,
                               1 if training else 0, L.ptr(dX), L.ptr(dg), L.ptr(db), L.ptr(scratch), L.stream()))
        return dX, dg, db, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------------
# A3 / A10 ioLayers  (custom_operations.py:7-10, 67-86; roi_select_sparse.py:79-81,117-119)
# ------------------------------------------------------------------------------------------------------
class InputLayerFunction(torch.autograd.Function):
    """``scn.ioLayers.InputLayerFunction.apply(dimension, metadata, spatial_size, coords, features, batch_size, mode)``"""

    @staticmethod
    def forward(ctx, dimension, metadata, spatial_size, coords, input_features, batch_size, mode):
        lib = L.lib()
        if int(dimension) != 3:
            raise NotImplementedError("only dimension 3")
        mode = int(mode)
        dev = torch.device("cuda", torch.cuda.current_device())
        F = input_features.to(dev)
        if F.dtype == torch.bfloat16:           # bf16-stored point features (the ROI crop of a bf16 slab): widened exactly;
            F = F.float()                       # the mean over a voxel's points accumulates in fp32 / fp64
        F = _f32(F)
        if coords.shape[0] != F.shape[0]:
            raise ValueError("coords / features row mismatch")
        if metadata.input_size is None:
            grid = metadata.set_input(spatial_size, coords, int(batch_size), mode)
        else:                                   # prepared ahead of time (Metadata.prepare_async) for these coords
            if metadata.n_items != coords.shape[0] or metadata.input_size != tuple(int(s) for s in spatial_size):
                raise L.ScnError("InputLayer: this Metadata was prepared for different coordinates")
            metadata.handover()
            if not metadata.prepared_for(coords):        # same point count, other coordinates (fixed-size sampling)
                raise L.ScnError("InputLayer: this Metadata was prepared for different coordinates")
            grid = metadata.grid(metadata.input_size)
        n_items, c = F.shape
        Y = _new((grid.n, c), F)
        acc = torch.empty((grid.n, c), dtype=torch.float64, device=dev) if mode in (3, 4) else None
        if mode == 1:
            metadata.row_last = torch.empty(grid.n, dtype=torch.int32, device=dev)
        L.check(lib.scn_input_fwd(L.ptr(F), L.ptr(metadata.item_row), L.ptr(metadata.row_count),
                                  L.ptr(metadata.row_first), n_items, grid.n, c, mode, L.ptr(Y), L.ptr(acc),
                                  L.ptr(metadata.row_last), L.stream()))
        ctx.md, ctx.mode, ctx.shape, ctx.in_device, ctx.in_dtype = metadata, mode, (n_items, c), input_features.device, \
            input_features.dtype
        return Y

    @staticmethod
    def backward(ctx, dY):
        md = ctx.md
        dY = _f32(dY)
        n_items, c = ctx.shape
        dF = _new((n_items, c), dY)
        L.check(L.lib().scn_input_bwd(L.ptr(dY), L.ptr(md.item_row), L.ptr(md.row_count), L.ptr(md.row_first),
                                      L.ptr(md.row_last), n_items, c, ctx.mode, L.ptr(dF), L.stream()))
        return None, None, None, None, dF.to(device=ctx.in_device, dtype=ctx.in_dtype), None, None


class OutputLayerFunction(torch.autograd.Function):
    """``scn.ioLayers.OutputLayerFunction.apply(dimension, metadata, features)`` -> one row per ORIGINAL input point."""

    @staticmethod
    def forward(ctx, dimension, metadata, input_features):
        X = _feat(input_features)               # fp32, or bf16 storage (a row copy either way)
        if metadata.item_row is None:
            raise L.ScnError("OutputLayer: this Metadata has no InputLayer rules")
        c = X.shape[1]
        Y = _new((metadata.n_items, c), X, X.dtype)
        gather = L.lib().scn_gather_rows_bf16 if _is_bf16(X) else L.lib().scn_gather_rows
        L.check(gather(L.ptr(X), L.ptr(metadata.item_row), metadata.n_items, c, L.ptr(Y), L.stream()))
        ctx.md, ctx.n_rows, ctx.hb = metadata, X.shape[0], _is_bf16(X)
        return Y

    @staticmethod
    def backward(ctx, dY):
        md = ctx.md
        dY = dY.to(torch.bfloat16).contiguous() if ctx.hb else _f32(dY)
        c = dY.shape[1]
        dX = _new((ctx.n_rows, c), dY, dY.dtype)
        acc = torch.empty((ctx.n_rows, c), dtype=torch.float64, device=dY.device)
        seg = L.lib().scn_segment_sum_bf16 if ctx.hb else L.lib().scn_segment_sum
        L.check(seg(L.ptr(dY), L.ptr(md.item_row), md.n_items, ctx.n_rows, c, L.ptr(dX), L.ptr(acc), L.stream()))
        return None, None, dX


# ------------------------------------------------------------------------------------------------------
# A13 SparseToDense (module_factory.py:429-435)
# ------------------------------------------------------------------------------------------------------
class SparseToDenseFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, metadata, spatial_size):
        X = _feat(features)                     # fp32, or bf16 storage (the dense volume then is bf16 too)
        size = tuple(int(s) for s in spatial_size)
        g = metadata.grid(size)
        n, c = X.shape
        out = torch.zeros((metadata.n_samples, c) + size, dtype=X.dtype, device=X.device)
        hs = L.host_i64(3)
        hs[0], hs[1], hs[2] = size
        fwd = L.lib().scn_sparse_to_dense_fwd_bf16 if _is_bf16(X) else L.lib().scn_sparse_to_dense_fwd
        L.check(fwd(L.ptr(X), L.ptr(g.coords), n, c, hs, L.ptr(out), L.stream()))
        ctx.g, ctx.size, ctx.shape, ctx.hb = g, size, (n, c), _is_bf16(X)
        return out

    @staticmethod
    def backward(ctx, dOut):
        dOut = dOut.to(torch.bfloat16).contiguous() if ctx.hb else _f32(dOut)
        n, c = ctx.shape
        dX = _new((n, c), dOut, dOut.dtype)
        hs = L.host_i64(3)
        hs[0], hs[1], hs[2] = ctx.size
        bwd = L.lib().scn_sparse_to_dense_bwd_bf16 if ctx.hb else L.lib().scn_sparse_to_dense_bwd
        L.check(bwd(L.ptr(dOut), L.ptr(ctx.g.coords), n, c, hs, L.ptr(dX), L.stream()))
        return dX, None, None


# ------------------------------------------------------------------------------------------------------
# MaxPooling / AveragePooling 2^3 stride 2 (module_factory.py:315-354)
# ------------------------------------------------------------------------------------------------------
class PoolingFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, metadata: Metadata, in_size, average, stride=(2, 2, 2)):
        X = _feat(features)                     # fp32, or bf16 storage
        rb = metadata.strided_rulebook(in_size, stride)
        c = X.shape[1]
        Y = _new((rb.n_coarse, c), X, X.dtype)
        mode = int(bool(average)) | ((rb.n_off << 8) if rb.n_off != 8 else 0)     # pool volume above bit 8 (0: the 2^3 default)
        fwd = L.lib().scn_pool_fwd_bf16 if _is_bf16(X) else L.lib().scn_pool_fwd
        L.check(fwd(L.ptr(X), L.ptr(rb.child), rb.n_coarse, c, mode, L.ptr(Y), L.stream()))
        ctx.save_for_backward(X, Y)
        ctx.rb, ctx.average = rb, mode
        return Y

    @staticmethod
    def backward(ctx, dY):
        X, Y = ctx.saved_tensors
        hb = _is_bf16(X)
        dY = dY.to(torch.bfloat16).contiguous() if hb else _f32(dY)
        dX = torch.empty_like(X)
        bwd = L.lib().scn_pool_bwd_bf16 if hb else L.lib().scn_pool_bwd
        L.check(bwd(L.ptr(X), L.ptr(Y), L.ptr(dY), L.ptr(ctx.rb.parent), X.shape[0], X.shape[1], ctx.average, L.ptr(dX),
                    L.stream()))
        return dX, None, None, None, None
