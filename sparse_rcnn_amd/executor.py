"""Host side of the step executor (csrc/scn_exec.hip, include/scn_mi355x.h "Step executor").

The reference drives the scn surface one layer at a time from Python (module_factory.py builds nested scn.Sequential trees,
model.py:414-431 / 758-782 run them), and so did this package's autograd functions: ~330 launches per backbone step, ~760
with the mask branch, ~13 us of host time each.  Here a U-Net LEVEL runs as one autograd node whose forward and backward are
one C call each: the node's launch plan (a flat list of `scn_exec_op`) is compiled ONCE per network from the module tree,
and a step only fills three pointer tables (feature slabs carved out of one workspace tensor, parameters / bf16 weight
images, parameter gradients).  Every op is one of the library's own entry points with the arguments the layer-by-layer
path passes, so both paths produce the same bits (tests/test_gpu_exec.py).

One node per LEVEL rather than per network: the parameter gradients of a level leave its node when that level's backward
ends, so the bucketed gradient all-reduce of dp.py still overlaps with the rest of backward, skip connections stay ordinary
autograd edges, and the encoder outputs (`SparseUNet.interims`, the RPN's inputs in the reference) stay differentiable.

Not covered (the layer-by-layer path runs instead): batch norm, `bf16_blocks=True` (casts around every run of units),
channel plans the two-source NetworkInNetwork kernel does not take, empty levels, and any step in which
`profiling.TIMER` brackets single launches with events.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from . import functional as F
from . import modules as M

ENABLED = os.environ.get("SCN_EXEC", "1") != "0"
# parameter-gradient ops of a backward pass on a second stream beside the backward-data chain (scn_exec_run_streams)
SIDE_LEAVES = os.environ.get("SCN_EXEC_SIDE", "0") != "0"
_side_streams, _side_scratch = {}, {}

i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p

OP_GEMM_IDENT, OP_CONV_SUBM, OP_CONV_CHILD, OP_RULES_CHILD, OP_ROWS2 = 1, 2, 3, 4, 5
OP_WGRAD_SUBM, OP_WGRAD2_SUBM, OP_WGRAD_DOWN, OP_WGRAD_UP, OP_WGRAD_IDENT, OP_COLSUM, OP_ADD, OP_CAST = 6, 7, 8, 9, 10, 11, 12, 13
XF_BF16, XF_COARSE_ROWS = 1 << 16, 1 << 17
BACK = L.F_W_TRANSPOSED | L.F_OFF_REVERSE


class ExecOp(C.Structure):
    _fields_ = [(n, i32) for n in ("op", "flags", "level", "cin", "cout", "x", "y", "r", "m", "x1", "y1", "c1", "w", "b", "aux",
                                   "reserved")]


class ExecLevel(C.Structure):
    _fields_ = [("n", i64), ("tstab", vp), ("tile_mask", vp), ("perm", vp), ("tile_order", vp), ("in_rows", vp),
                ("out_rows", vp), ("prefix_host", vp), ("n_coarse", i64), ("c_tstab", vp), ("c_tile_mask", vp),
                ("c_perm", vp), ("c_tile_order", vp), ("c_in_rows", vp), ("c_out_rows", vp), ("c_prefix_host", vp),
                ("flags", i64)]


def _addr(prefix_host):
    return C.cast(prefix_host, vp).value


def build_levels(md, size, n_levels):
    """The index structures of `n_levels` U-Net levels below spatial size `size`, as the executor's level table.
    -> (ctypes array, objects kept alive, rows per level), cached on the Metadata; None when a level is empty."""
    size = tuple(int(s) for s in size)
    cache = md.__dict__.setdefault("_exec_levels", {})
    hit = cache.get((size, n_levels))
    if hit is not None:
        return hit or None
    nat = md.__dict__.get("_native")
    if nat is not None and nat[0] == size and nat[1] >= n_levels and nat[2] == 3:
        out = _levels_from_native(md, nat, n_levels)
        if out is not None:
            cache[(size, n_levels)] = out
            return out or None
    arr = (ExecLevel * n_levels)()
    keep, ns = [], []
    sz = size
    for l in range(n_levels):
        rb = md.subm_rulebook(sz, 3)
        if rb.n == 0 or rb.rules is None or rb.tiles is None:
            cache[(size, n_levels)] = False
            return None
        e, t, r = arr[l], rb.tiles, rb.rules
        e.n = rb.n
        e.flags = 1 if getattr(t, "has_x", False) else 0           # SCN_XL_TILE_ORDER_X
        # (`ptr`: addresses straight from the build workspace's layout -- no tensor view is created for the executor)
        e.tstab, e.tile_mask, e.perm, e.tile_order = t.ptr("tstab"), t.ptr("tile_mask"), t.ptr("perm"), t.ptr("tile_order")
        ph = r.prefix_host
        e.in_rows, e.out_rows, e.prefix_host = r.ptr("in_rows"), r.ptr("out_rows"), _addr(ph)
        keep += [rb, ph]
        ns.append(rb.n)
        if l + 1 < n_levels:
            sb = md.strided_rulebook(sz)
            if sb.n_coarse == 0:
                cache[(size, n_levels)] = False
                return None
            ct, cr = sb.tiles, sb.rules
            cph = cr.prefix_host
            e.n_coarse = sb.n_coarse
            e.c_tstab, e.c_tile_mask, e.c_perm, e.c_tile_order = ct.ptr("tstab"), ct.ptr("tile_mask"), ct.ptr("perm"), ct.ptr("tile_order")
            e.c_in_rows, e.c_out_rows, e.c_prefix_host = cr.ptr("in_rows"), cr.ptr("out_rows"), _addr(cph)
            keep += [sb, cph]
            sz = sb.coarse_size
    out = cache[(size, n_levels)] = (arr, keep, ns)
    return out


def _levels_from_native(md, nat, n_levels):
    """build_levels for a Metadata that ONE native call built (Metadata.build_native): every address is workspace base +
    descriptor offset (include/scn_mi355x.h: scn_pyramid_build descriptor), the rule prefixes are the host arrays the rule
    objects already own -- ~15 integer operations per level instead of ~40 attribute chains / view lookups (round 5: the
    detection + mask step follows the host; two Metadata objects per step).  False: a level is empty (layer-by-layer path);
    None: something is not as a native build leaves it (the general path decides)."""
    size, _, _, wsp, per_level, has_x = nat
    arr = (ExecLevel * n_levels)()
    keep, ns = [getattr(md, "_workspace", None)], []
    sz = size
    for l in range(n_levels):
        D = per_level[l]
        rb = md.subm.get((sz, 3))
        if D[0] == 0 or rb is None:
            return False
        rules = rb.__dict__.get("rules")
        ph = None if rules is None else rules.__dict__.get("_prefix_host")
        if ph is None or "_specs" not in rules.__dict__:
            return None
        e = arr[l]
        e.n = D[0]
        e.flags = 1 if has_x else 0
        e.tstab, e.tile_mask, e.perm, e.tile_order = wsp + D[10], wsp + D[11], wsp + D[9], wsp + D[12]
        e.in_rows, e.out_rows, e.prefix_host = wsp + D[64], wsp + D[65], _addr(ph)
        keep.append(ph)
        ns.append(D[0])
        md._note_levels(sz, 1)                           # (the depth hint of the drop-in path, as Metadata.subm_rulebook)
        if l + 1 < n_levels:
            sb = md.strided.get(sz)
            if sb is None or per_level[l + 1][0] == 0:
                return False
            md._unrequested.discard(sz)                 # (as Metadata.strided_rulebook: a layer of this forward asked for it)
            md._note_levels(sz, 2)
            crules = sb.__dict__.get("rules")
            cph = None if crules is None else crules.__dict__.get("_prefix_host")
            if cph is None or "_specs" not in crules.__dict__:
                return None
            e.n_coarse = per_level[l + 1][0]
            e.c_tstab, e.c_tile_mask, e.c_perm, e.c_tile_order = wsp + D[21], wsp + D[22], wsp + D[20], wsp + D[23]
            e.c_in_rows, e.c_out_rows, e.c_prefix_host = wsp + D[66], wsp + D[67], _addr(cph)
            keep.append(cph)
            sz = sb.coarse_size
    return arr, keep, ns


# ----------------------------------------------------------------------------------------------------------------------
# plan compiler
# ----------------------------------------------------------------------------------------------------------------------
class Stage:
    """The launch plan of one autograd node: forward and backward op lists over symbolic buffer / parameter / gradient ids."""

    def __init__(self, bf16, n_inputs):
        self.bf16 = bool(bf16)
        self.hb = XF_BF16 if bf16 else 0
        self.es = 2 if bf16 else 4
        self.n_inputs = n_inputs
        self.fwd, self.bwd = [], []
        self.bufs = []                  # id -> (level, channels, elem bytes, kind)   kind: "f" fwd ws | "b" bwd ws | "x" external
        self.mods = []                  # modules whose (W, b) the node takes: (module, cin_phys)
        self.params = []                # params-table descriptors: ("w", mod idx) | ("b", mod idx) | ("img", mod idx, cin, cout, n_off, flags)
        self.gregions = []              # gradient regions: list of [(mod idx, "w"|"b"), ...] laid out back to back
        self.in_ids, self.din_ids = [], []
        self.out_id = self.dout_id = None

    # ---- symbols ----------------------------------------------------------------------------------------------------
    def buf(self, level, ch, es=None, kind="f"):
        self.bufs.append((level, int(ch), self.es if es is None else es, kind))
        return len(self.bufs) - 1

    def mod(self, module, cin_phys):
        for i, (m, c) in enumerate(self.mods):
            if m is module:
                return i
        self.mods.append((module, int(cin_phys)))
        return len(self.mods) - 1

    def param(self, *desc):
        if desc not in self.params:
            self.params.append(desc)
        return self.params.index(desc)

    def weight(self, mi, cin, cout, n_off, flags, tile_kernel):
        """params id of the weight operand of a data op: the fp32 weight, or (bf16 storage, tile kernel) its packed image."""
        if self.bf16 and tile_kernel:
            return self.param("img", mi, cin, cout, n_off, flags & BACK)
        return self.param("w", mi)

    def gregion(self, *members):
        self.gregions.append(list(members))
        return len(self.gregions) - 1

    def op(self, lst, op, level, cin, cout, x=-1, y=-1, r=-1, m=-1, x1=-1, y1=-1, c1=0, w=-1, b=-1, aux=0, flags=0):
        lst.append(ExecOp(op, flags, level, cin, cout, x, y, r, m, x1, y1, c1, w, b, aux, 0))

    # ---- residual units ----------------------------------------------------------------------------------------------
    def units_fwd(self, level, c, x, blocks, out_id=None):
        """blocks: [(conv1, conv2)] of x + SubM3(ReLU(SubM3(ReLU(x)))) at `c` physical channels.  -> (output id, saved)"""
        saved = []
        for k, (c1, c2) in enumerate(blocks):
            m1, m2 = self.mod(c1, c), self.mod(c2, c)
            y1 = self.buf(level, c)
            y = out_id if (out_id is not None and k == len(blocks) - 1) else self.buf(level, c)
            fl = L.F_RELU_IN | self.hb
            self.op(self.fwd, OP_CONV_SUBM, level, c, c, x=x, y=y1, w=self.weight(m1, c, c, 27, 0, True), b=self.param("b", m1), flags=fl)
            self.op(self.fwd, OP_CONV_SUBM, level, c, c, x=y1, y=y, r=x, w=self.weight(m2, c, c, 27, 0, True), b=self.param("b", m2), flags=fl)
            saved.append((x, y1, m1, m2))
            x = y
        return x, saved

    def units_bwd(self, level, c, g, saved, din_id=None):
        """Backward of units_fwd: g = gradient of the last block's output -> id of the gradient of the first block's input."""
        for k, (x, y1, m1, m2) in enumerate(reversed(saved)):
            dy1 = self.buf(level, c, kind="b")
            dx = din_id if (din_id is not None and k == len(saved) - 1) else self.buf(level, c, kind="b")
            self.op(self.bwd, OP_CONV_SUBM, level, c, c, x=g, y=dy1, m=y1, w=self.weight(m2, c, c, 27, BACK, True), flags=BACK | self.hb)
            self.op(self.bwd, OP_CONV_SUBM, level, c, c, x=dy1, y=dx, m=x, r=g, w=self.weight(m1, c, c, 27, BACK, True),
                    flags=BACK | L.F_RESIDUAL_LAST | self.hb)
            gw, gb = self.gregion((m1, "w"), (m2, "w")), self.gregion((m1, "b"), (m2, "b"))
            self.op(self.bwd, OP_WGRAD2_SUBM, level, c, c, x=x, y=dy1, x1=y1, y1=g, w=gw, b=gb, flags=L.F_RELU_IN | self.hb)
            g = dx
        return g

    # ---- finish -----------------------------------------------------------------------------------------------------
    def freeze(self):
        self.fwd_arr = (ExecOp * len(self.fwd))(*self.fwd)
        self.bwd_arr = (ExecOp * len(self.bwd))(*self.bwd)
        self.fwd_ws = [i for i, b in enumerate(self.bufs) if b[3] == "f"]
        self.bwd_ws = [i for i, b in enumerate(self.bufs) if b[3] == "b"]
        # host-side tables of a call, built once (the per-call Python is what a host-bound step feels: DESIGN.md section 5)
        self.fwd_specs = [(self.bufs[i][0], self.bufs[i][1] * self.bufs[i][2]) for i in self.fwd_ws]     # (level, row bytes)
        self.bwd_specs = [(self.bufs[i][0], self.bufs[i][1] * self.bufs[i][2]) for i in self.bwd_ws]
        self.w_entries = [(k, 2 * d[1]) for k, d in enumerate(self.params) if d[0] == "w"]
        self.b_entries = [(k, 2 * d[1] + 1) for k, d in enumerate(self.params) if d[0] == "b"]
        self.img_entries = [(k, 2 * d[1]) + tuple(d[2:]) for k, d in enumerate(self.params) if d[0] not in ("w", "b")]
        self.grad_views = {}                              # tuple of parameter shapes -> (region offsets, total, views)
        self._lean_slots()
        covered = sorted(m for reg in self.gregions for m in reg)
        want = sorted((i, k) for i in range(len(self.mods)) for k in ("w", "b"))
        if covered != want:
            raise RuntimeError("executor plan: every parameter must sit in exactly one gradient region")
        return self


def _lean_slots(self):
    """Forward-only plan (evaluation: `eval_model`, ndsis/training/training.py:244-304, and `SparseMaskPredictor`,
    model.py:826-882, run the forward under torch.no_grad()): nothing is kept for a backward pass, so a forward slab's
    storage is handed on as soon as its last reader has been queued.  Slots per (level, row bytes) class by liveness over the
    forward op list; an op's output never shares a slot with one of its own operands (a slot is released AFTER the op)."""
    last = {}
    for t, op in enumerate(self.fwd):
        for v in (op.x, op.r, op.m, op.x1, op.y, op.y1):
            if v >= 0:
                last[v] = t
    ws = set(i for i, b in enumerate(self.bufs) if b[3] == "f")
    free, n_slots, slot_of = {}, {}, {}
    for t, op in enumerate(self.fwd):
        for v in (op.y, op.y1):
            if v in ws and v not in slot_of:
                cls = (self.bufs[v][0], self.bufs[v][1] * self.bufs[v][2])
                if free.get(cls):
                    slot_of[v] = (cls, free[cls].pop())
                else:
                    slot_of[v] = (cls, n_slots.get(cls, 0))
                    n_slots[cls] = n_slots.get(cls, 0) + 1
        for v in (op.x, op.r, op.m, op.x1, op.y, op.y1):
            if v in slot_of and last[v] == t and slot_of[v] is not None:
                cls, k = slot_of[v]
                if k not in free.setdefault(cls, []):
                    free[cls].append(k)
    for v in ws:                                          # (a declared slab no forward op writes: give it a slot of its own)
        if v not in slot_of:
            cls = (self.bufs[v][0], self.bufs[v][1] * self.bufs[v][2])
            slot_of[v] = (cls, n_slots.get(cls, 0))
            n_slots[cls] = n_slots.get(cls, 0) + 1
    self.lean_classes = sorted(n_slots.items())            # [((level, row bytes), slots)]
    self.lean_slot_of = [slot_of[i] for i in self.fwd_ws]


Stage._lean_slots = _lean_slots


def _lean_layout(stage, ns):
    """-> (offset per forward workspace buffer, total bytes) of the forward-only plan."""
    base, tot = {}, 0
    for (lv, row_bytes), k in stage.lean_classes:
        size = (ns[lv] * row_bytes + 255) & ~255
        base[(lv, row_bytes)] = (tot, size)
        tot += size * k
    return [base[cls][0] + k * base[cls][1] for cls, k in stage.lean_slot_of], tot


def _plain_blocks(unit_seq):
    """[(conv1, conv2)] of a `units(...)` Sequential, or None when a block is not the plain pre-activation unit."""
    out = []
    for block in unit_seq:
        if not (type(block) is M.Sequential and len(block) == 2 and type(block[0]) is M.ConcatTable and type(block[1]) is M.AddTable):
            return None
        a, inner = list(block[0]._modules.values()) if len(block[0]._modules) == 2 else (None, None)
        if type(a) is not M.Identity or type(inner) is not M.Sequential or len(inner) != 4:
            return None
        r0, c1, r1, c2 = list(inner._modules.values())
        if not (type(r0) is M.ReLU and type(r1) is M.ReLU and type(c1) is M.SubmanifoldConvolution
                and type(c2) is M.SubmanifoldConvolution and c1.filter_size == 3 and c2.filter_size == 3
                and c1.bias is not None and c2.bias is not None and c1.nIn == c2.nOut and c1.nOut == c2.nIn == c1.nIn
                and c1.groups == 1 and c2.groups == 1):
            return None
        out.append((c1, c2))
    return out


def _phys(m):
    return int(getattr(m, "pad_out_to", None) or m.nOut)


def compile_encoder_stage(level, head, blocks, cin_phys, bf16, cast_first=False, cast_last=False, in_bf16=False):
    """Encoder level: head (SubM 1^3 at level 0 | Convolution 2^3/2 from level-1) + residual units.
    level 0 in bf16 storage: the head runs in fp32 on the fp32 input and its result is cast (SparseUNet), or -- cast_first,
    the mask branch's input stage -- the input is cast first and the head runs on bf16 rows, or -- in_bf16 -- the input
    slab is bf16-stored already (a SubM 1^3 stage behind a bf16 network in a module tree); cast_last: fp32 output."""
    st = Stage(bf16, 1)
    in_bf16 = bool(bf16 and in_bf16 and level == 0)
    es_in = (2 if in_bf16 else 4) if (level == 0) else st.es
    c = _phys(head)
    IN = st.buf(level if level == 0 else level - 1, cin_phys, es_in, "x")
    st.in_ids = [IN]
    mh = st.mod(head, cin_phys)
    out_es = 4 if (cast_last or not bf16) else 2
    OUT = st.buf(level, c, out_es, "x")
    st.out_id = OUT
    if level == 0:
        if bf16 and (cast_first or in_bf16):
            xin = IN
            if not in_bf16:
                xin = st.buf(0, cin_phys)
                st.op(st.fwd, OP_CAST, 0, cin_phys, 0, x=IN, y=xin, flags=XF_BF16)
            h = st.buf(0, c)
            st.op(st.fwd, OP_GEMM_IDENT, 0, cin_phys, c, x=xin, y=h, w=st.param("w", mh), b=st.param("b", mh), flags=XF_BF16)
            x, head_in, head_hb = h, xin, XF_BF16
        else:
            h = st.buf(0, c, 4)
            st.op(st.fwd, OP_GEMM_IDENT, 0, cin_phys, c, x=IN, y=h, w=st.param("w", mh), b=st.param("b", mh), flags=0)
            x, head_in, head_hb = h, IN, 0
            if bf16:
                x = st.buf(0, c)
                st.op(st.fwd, OP_CAST, 0, c, 0, x=h, y=x, flags=XF_BF16)
    else:
        x = st.buf(level, c)
        st.op(st.fwd, OP_CONV_CHILD, level - 1, cin_phys, c, x=IN, y=x, w=st.weight(mh, cin_phys, c, 8, 0, True),
              b=st.param("b", mh), flags=st.hb)
    if bf16 and cast_last:
        last, saved = st.units_fwd(level, c, x, blocks)
        st.op(st.fwd, OP_CAST, level, c, 0, x=last, y=OUT, flags=0)
    else:
        last, saved = st.units_fwd(level, c, x, blocks, out_id=OUT)
    # ---- backward
    DOUT = st.buf(level, c, out_es, "x")
    st.dout_id = DOUT
    g = DOUT
    if bf16 and cast_last:
        g = st.buf(level, c, kind="b")
        st.op(st.bwd, OP_CAST, level, c, 0, x=DOUT, y=g, flags=XF_BF16)
    g = st.units_bwd(level, c, g, saved)
    DIN = st.buf(level if level == 0 else level - 1, cin_phys, es_in, "x")
    st.din_ids = [DIN]
    gw, gb = st.gregion((mh, "w")), st.gregion((mh, "b"))
    if level == 0:
        if head_hb:                                           # head ran on bf16 rows: its input gradient is cast back at the end
            dinb = DIN if in_bf16 else st.buf(0, cin_phys, kind="b")
            st.op(st.bwd, OP_GEMM_IDENT, 0, c, cin_phys, x=g, y=dinb, w=st.param("w", mh), flags=BACK | XF_BF16)
            st.op(st.bwd, OP_WGRAD_IDENT, 0, cin_phys, c, x=head_in, y=g, w=gw, b=gb, flags=XF_BF16)
            if not in_bf16:
                st.op(st.bwd, OP_CAST, 0, cin_phys, 0, x=dinb, y=DIN, flags=0)
        else:
            gf = g
            if bf16:
                gf = st.buf(0, c, 4, "b")
                st.op(st.bwd, OP_CAST, 0, c, 0, x=g, y=gf, flags=0)
            st.op(st.bwd, OP_GEMM_IDENT, 0, c, cin_phys, x=gf, y=DIN, w=st.param("w", mh), flags=BACK)
            st.op(st.bwd, OP_WGRAD_IDENT, 0, cin_phys, c, x=IN, y=gf, w=gw, b=gb, flags=0)
    else:
        st.op(st.bwd, OP_RULES_CHILD, level - 1, c, cin_phys, x=g, y=DIN, w=st.param("w", mh), flags=L.F_W_TRANSPOSED | st.hb)
        st.op(st.bwd, OP_WGRAD_DOWN, level - 1, cin_phys, c, x=IN, y=g, w=gw, flags=st.hb)
        st.op(st.bwd, OP_COLSUM, level, c, 0, x=g, b=gb, flags=st.hb)
    return st.freeze()


def compile_decoder_stage(level, up, nin, blocks, c_coarse, bf16):
    """Decoder level: ReLU -> Deconvolution 2^3/2 (coarse level+1 -> level) -> JoinTable([up, skip]) -> NetworkInNetwork ->
    residual units.  Inputs: (coarse slab, skip slab)."""
    st = Stage(bf16, 2)
    c = _phys(up)
    IN0 = st.buf(level + 1, c_coarse, None, "x")
    IN1 = st.buf(level, c, None, "x")
    st.in_ids = [IN0, IN1]
    OUT = st.buf(level, c, None, "x")
    st.out_id = OUT
    mu, mn = st.mod(up, c_coarse), st.mod(nin, 2 * c)
    upb = st.buf(level, c)
    st.op(st.fwd, OP_RULES_CHILD, level, c_coarse, c, x=IN0, y=upb, w=st.param("w", mu), b=st.param("b", mu),
          flags=L.F_RELU_IN | st.hb)
    h = st.buf(level, c)
    st.op(st.fwd, OP_ROWS2, level, c, c, x=upb, x1=IN1, c1=c, y=h, w=st.param("w", mn), b=st.param("b", mn), flags=st.hb)
    last, saved = st.units_fwd(level, c, h, blocks, out_id=OUT)
    # ---- backward
    DOUT = st.buf(level, c, None, "x")
    st.dout_id = DOUT
    g = st.units_bwd(level, c, DOUT, saved)
    DIN0 = st.buf(level + 1, c_coarse, None, "x")
    DIN1 = st.buf(level, c, None, "x")
    st.din_ids = [DIN0, DIN1]
    dup = st.buf(level, c, kind="b")
    st.op(st.bwd, OP_ROWS2, level, c, c, x=g, y=dup, y1=DIN1, c1=c, w=st.param("w", mn), flags=L.F_W_TRANSPOSED | st.hb)
    gwn, gbn = st.gregion((mn, "w")), st.gregion((mn, "b"))
    st.op(st.bwd, OP_WGRAD_IDENT, level, c, c, x=upb, y=g, w=gwn, b=gbn, aux=0, flags=st.hb)
    st.op(st.bwd, OP_WGRAD_IDENT, level, c, c, x=IN1, y=g, w=gwn, b=-1, aux=c * c, flags=st.hb)
    st.op(st.bwd, OP_CONV_CHILD, level, c, c_coarse, x=dup, y=DIN0, m=IN0,
          w=st.weight(mu, c, c_coarse, 8, L.F_W_TRANSPOSED, True), flags=L.F_W_TRANSPOSED | st.hb)
    gwu, gbu = st.gregion((mu, "w")), st.gregion((mu, "b"))
    st.op(st.bwd, OP_WGRAD_UP, level, c_coarse, c, x=IN0, y=dup, w=gwu, b=gbu, flags=L.F_RELU_IN | st.hb)
    return st.freeze()


# ----------------------------------------------------------------------------------------------------------------------
# run time
# ----------------------------------------------------------------------------------------------------------------------
def _layout(specs, ns):
    offs, tot = [], 0
    for lv, row_bytes in specs:
        offs.append(tot)
        tot += (ns[lv] * row_bytes + 255) & ~255
    return offs, tot


def _run(stage, ops, levels, table, ptab, gtab, dev, side=False):
    from . import profiling
    t = profiling.TIMER
    if t is not None and t.exec_timing():      # a sampled step of bench.py: the C call brackets its tile-convolution launches
        lib = L.lib()
        lib.scn_exec_timing_enable(1)
        try:
            return _run_plain(stage, ops, levels, table, ptab, gtab, dev, side)
        finally:
            lib.scn_exec_timing_enable(0)
    return _run_plain(stage, ops, levels, table, ptab, gtab, dev, side)


def _run_plain(stage, ops, levels, table, ptab, gtab, dev, side=False):
    lib = L.lib()
    arr, _, ns = levels
    sb, ac = i64(0), i64(0)
    L.check(lib.scn_exec_requirements(ops, len(ops), arr, len(ns), C.byref(sb), C.byref(ac)))
    scratch = L.scratch(sb.value, dev)
    arrival = L.arrival(max(ac.value, 1), dev)
    if side and SIDE_LEAVES:
        key = (dev.index, L.stream())
        st = _side_streams.get(key)
        if st is None:
            st = _side_streams[key] = torch.cuda.Stream(device=dev)
        ss = _side_scratch.get(key)
        if ss is None or ss.numel() < scratch.numel():
            ss = _side_scratch[key] = torch.empty(scratch.numel(), dtype=torch.uint8, device=dev)
        L.check(lib.scn_exec_run_streams(ops, len(ops), arr, len(ns), table, ptab, gtab, scratch.data_ptr(), scratch.numel(),
                                         arrival.data_ptr(), L.stream(), st.cuda_stream, ss.data_ptr(), ss.numel()))
        return
    L.check(lib.scn_exec_run(ops, len(ops), arr, len(ns), table, ptab, gtab, scratch.data_ptr(), scratch.numel(),
                             arrival.data_ptr(), L.stream()))


LAST_FORWARD = {"lean": False, "ws_bytes": 0}      # what the last stage forward did (tests)


def _lean(inputs, phys):
    return not torch.is_grad_enabled() or not any(t is not None and t.requires_grad for t in (*inputs, *phys))


class StageFunction(torch.autograd.Function):
    """forward(stage, levels, lean, *inputs, *(W, b per module of the stage)) -> the stage's output slab.
    lean: nothing will ask for a backward pass (torch.no_grad(), or no operand requires a gradient): the forward-only slab
    plan (`_lean_layout`: storage re-used along the op list) and nothing saved."""

    @staticmethod
    def forward(ctx, stage, levels, lean, *ts):
        n_in = stage.n_inputs
        inputs = [F._feat(t) for t in ts[:n_in]]
        phys = ts[n_in:]
        dev = inputs[0].device
        ns = levels[2]
        lean = bool(lean) and F.RELU_RECORD is None
        offs, total = _lean_layout(stage, ns) if lean else _layout(stage.fwd_specs, ns)
        LAST_FORWARD["lean"], LAST_FORWARD["ws_bytes"] = lean, total
        ws = torch.empty(max(total, 256), dtype=torch.uint8, device=dev)
        lv, ch, es, _ = stage.bufs[stage.out_id]
        out = torch.empty((ns[lv], ch), dtype=torch.float32 if es == 4 else torch.bfloat16, device=dev)
        table = (vp * len(stage.bufs))()
        base = ws.data_ptr()
        for i, o in zip(stage.fwd_ws, offs):
            table[i] = base + o
        for i, t in zip(stage.in_ids, inputs):
            blv, bch, bes, _ = stage.bufs[i]
            if t.shape != (ns[blv], bch) or t.element_size() != bes:
                raise L.ScnError(f"executor: input slab {tuple(t.shape)} {t.dtype} where the plan expects {(ns[blv], bch)} x {bes} B")
            table[i] = t.data_ptr()
        table[stage.out_id] = out.data_ptr()
        ptab = (vp * max(len(stage.params), 1))()
        images = []
        f32 = torch.float32
        for k, j in stage.w_entries:
            t = phys[j]
            ptab[k] = (t if (t.dtype is f32 and t.is_contiguous()) else F._f32(t)).data_ptr()
        for k, j in stage.b_entries:
            t = phys[j]
            ptab[k] = 0 if t is None else (t if (t.dtype is f32 and t.is_contiguous()) else F._f32(t)).data_ptr()
        for k, j, cin, cout, n_off, fl in stage.img_entries:
            if lean and fl:                      # a backward-data image: only a backward pass reads it
                continue
            W = phys[j]
            ent = F.packed_entry(W, cin, cout, n_off, fl)
            if ent is None:
                img = F.pack_weights_bf16(W.detach(), cin, cout, n_off, fl)
                ent = (img, img.data_ptr())
            if not images or images[-1] is not ent[0]:
                images.append(ent[0])                # (the pack buffer of a network: one tensor for all its images)
            ptab[k] = ent[1]
        _run(stage, stage.fwd_arr, levels, table, ptab, None, dev)
        if F.RELU_RECORD is not None:
            _record_relu_masks(stage, ns, ws, offs, inputs, out)
        if lean:
            return out
        ctx.stage, ctx.levels, ctx.table, ctx.ptab = stage, levels, table, ptab
        ctx.phys_shapes = tuple(None if t is None else tuple(t.shape) for t in phys)
        # Everything the pointer tables name travels through save_for_backward: the parameter tensors, the forward workspace,
        # the packed weight images and the input slabs.  They stay alive exactly as long as autograd keeps the graph
        # (retain_graph=True: a second backward reads the same slabs; otherwise they are released when backward ends and a
        # second backward raises autograd's own "backward through the graph a second time" error).
        ctx.save_for_backward(*[t for t in phys if t is not None], ws, *images, *inputs)
        return out

    @staticmethod
    def backward(ctx, dout):
        kept = ctx.saved_tensors                 # (raises when the graph's buffers have been freed)
        stage, levels, ptab = ctx.stage, ctx.levels, ctx.ptab
        table = type(ctx.table).from_buffer_copy(ctx.table)      # a fresh table per pass: the forward's entries stay as they were
        ns = levels[2]
        dev = dout.device
        lv, ch, es, _ = stage.bufs[stage.dout_id]
        dout = dout.contiguous()
        if dout.element_size() != es:
            dout = dout.to(torch.float32 if es == 4 else torch.bfloat16)
        offs, total = _layout(stage.bwd_specs, ns)
        ws = torch.empty(max(total, 256), dtype=torch.uint8, device=dev)
        base = ws.data_ptr()
        for i, o in zip(stage.bwd_ws, offs):
            table[i] = base + o
        table[stage.dout_id] = dout.data_ptr()
        dins = []
        for i in stage.din_ids:
            blv, bch, bes, _ = stage.bufs[i]
            t = torch.empty((ns[blv], bch), dtype=torch.float32 if bes == 4 else torch.bfloat16, device=dev)
            dins.append(t)
            table[i] = t.data_ptr()
        # gradient regions back to back in one fp32 buffer; the returned gradients are views of it
        shapes = ctx.phys_shapes
        cached = stage.grad_views.get(shapes)
        if cached is None:
            goffs, gtot, views = [], 0, [None] * len(shapes)
            for reg in stage.gregions:
                goffs.append(gtot)
                start = gtot
                for mi, kind in reg:
                    sh = shapes[2 * mi + (kind == "b")]
                    if sh is not None:
                        n = 1
                        for v in sh:
                            n *= v
                        views[2 * mi + (kind == "b")] = (gtot, n, sh)
                        gtot += n
                if gtot == start:                    # (plans are only compiled for layers WITH bias: _require_bias)
                    raise L.ScnError("executor: a gradient region without a tensor (bias=None layer in a compiled stage)")
                gtot = (gtot + 63) & ~63
            cached = stage.grad_views[shapes] = (goffs, gtot, views)
        goffs, gtot, views = cached
        flat = torch.empty(max(gtot, 1), dtype=torch.float32, device=dev)
        gtab = (vp * max(len(stage.gregions), 1))()
        gb = flat.data_ptr()
        for k, o in enumerate(goffs):
            gtab[k] = gb + 4 * o
        _run(stage, stage.bwd_arr, levels, table, ptab, gtab, dev, side=True)
        grads = [None if v is None else flat[v[0]:v[0] + v[1]].view(v[2]) for v in views]
        del kept
        return (None, None, None, *dins, *grads)


def _record_relu_masks(stage, ns, ws, offs, inputs, out):
    """functional.RELU_RECORD for a stage (the parity tests' frozen ReLU masks): the sign mask of every slab a forward op of
    the plan applies its fused input ReLU to, in op order -- the order the layer-by-layer path records them in (a residual
    unit: x, then y1; a decoder level: the coarse input of its deconvolution first).  Read from the slabs the executor
    keeps for backward, after the forward launches have been queued."""
    where = dict(zip(stage.fwd_ws, offs))
    ext = dict(zip(stage.in_ids, inputs))
    ext[stage.out_id] = out
    for op in stage.fwd:
        if not (op.flags & L.F_RELU_IN):
            continue
        lv, ch, es, kind = stage.bufs[op.x]
        if op.x in ext:
            t = ext[op.x]
        else:
            nb = ns[lv] * ch * es
            t = ws[where[op.x]:where[op.x] + nb].view(torch.float32 if es == 4 else torch.bfloat16).view(ns[lv], ch)
        F._rec_relu(t)


def _require_bias(*mods):
    """Plans write a bias gradient for every conv-type layer they cover (COLSUM / the b region of the WGRAD ops): a layer
    built with bias=False keeps the layer-by-layer path."""
    return all(getattr(m, "bias", None) is not None for m in mods)


def run_stage(stage, levels, inputs, pack=False):
    """Apply a compiled stage: the physical (W, b) of its modules are fetched through `_wb` (zero-padded views of the logical
    parameters where a layer is channel-padded -- autograd carries their gradients back through the pad).
    pack: bf16 storage outside a network-wide `packed_weights` block (a stage of a module tree somebody else built): the
    stage's weight images are packed by ONE launch of its own instead of one per image."""
    phys = []
    for m, cin_phys in stage.mods:
        W, b = m._wb(cin_phys)
        phys += [W, b]
    if pack and stage.bf16:
        lean = _lean(inputs, phys)
        attr = "_pack_plan_fwd" if lean else "_pack_plan"
        plan = stage.__dict__.get(attr)
        if plan is None:
            jobs = []
            for d in stage.params:
                if d[0] == "img":
                    _, mi, cin, cout, n_off, fl = d
                    if lean and fl:
                        continue
                    m, cin_phys = stage.mods[mi]
                    if getattr(m, "pad_out_to", None) or cin_phys != m.nIn:
                        jobs = None                      # padded layers see fresh padded weight tensors: packed per call
                        break
                    jobs.append((m.weight, cin, cout, n_off, fl))
            plan = F.PackPlan(jobs) if jobs else False
            setattr(stage, attr, plan)
        if plan:
            with F.packed_weights(plan):
                return StageFunction.apply(stage, levels, lean, *inputs, *phys)
    return StageFunction.apply(stage, levels, _lean(inputs, phys), *inputs, *phys)
