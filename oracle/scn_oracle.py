"""CPU oracle for the sparse-convolution hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``sparse_rcnn_amd``) never does.

PARITY UNPINNED for the sparse-convolution arithmetic: that arithmetic lives in
the third-party package ``facebookresearch/SparseConvNet`` (imported by the
reference as ``sparseconvnet``; no version pinned, README.md:23-31), which is
neither under /root/reference nor installed here, and the reference holds no
tests or golden vectors for it.  What pins this restatement
instead: (i) every operator below is checked against the dense ``torch.nn.functional``
twin the reference itself pairs it with (module_factory.py:96-112, 231-239,
255-264, 365-372, 402-412) in ``tests/test_oracle_dense.py``; (ii) the ROI crop
(the one piece of hot-path arithmetic the reference owns) IS pinned by golden
vectors generated from the reference's own ``roi_cut`` (``tests/golden/``).

Index work is numpy (integer exact); feature work is torch-CPU fp32, organised
the way the SparseConvNet CPU path is: per kernel offset gather rows ->
``sgemm`` -> scatter-add, bias first, offsets ascending.

Canonical order (DESIGN.md §"Canonical order"): the reference leaves the row
numbering of strided-conv output sites and the order of rules inside an offset
implementation-defined (hash-map iteration upstream).  Here, and in the HIP
path: active rows are numbered by FIRST OCCURRENCE scanning input rows
ascending; rules inside an offset are sorted by OUTPUT row ascending.
"""
from __future__ import annotations

import numpy as np
import torch

# --------------------------------------------------------------------------
# keys
# --------------------------------------------------------------------------

def pack_keys(coords: np.ndarray) -> np.ndarray:
    """(x,y,z,b) int64 rows -> one uint64 key, b most significant.

    Same packing as the device hash (csrc/scn_index.hip pack_key): 16 bits per
    field.  Sorting keys sorts by (b, x, y, z).
    """
    c = np.asarray(coords, dtype=np.int64)
    assert c.ndim == 2 and c.shape[1] == 4
    assert (c >= 0).all() and (c < 65536).all(), "coordinate out of 16-bit range"
    c = c.astype(np.uint64)
    return (c[:, 3] << np.uint64(48)) | (c[:, 0] << np.uint64(32)) | (c[:, 1] << np.uint64(16)) | c[:, 2]


def _first_occurrence_rows(keys: np.ndarray):
    """Number distinct keys by first occurrence.  Returns (row_of_item, first_item_of_row)."""
    if len(keys) == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    uniq, first_idx, inverse = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first_idx, kind="stable")          # unique-id sorted by first appearance
    rank = np.empty(len(uniq), np.int64)
    rank[order] = np.arange(len(uniq))
    return rank[inverse.reshape(-1)], first_idx[order]


# --------------------------------------------------------------------------
# A3 InputLayer / A10 OutputLayer     (custom_operations.py:62-86, 7-10)
# --------------------------------------------------------------------------

def input_layer_rules(coords: np.ndarray):
    """coords int64 [Npts,4] (x,y,z,b) -> (active coords [N,4] int64, point_row [Npts], counts [N]).

    Rows are numbered by first occurrence in input-row order across the whole
    batch (InputLayer).
    """
    coords = np.asarray(coords, dtype=np.int64).reshape(-1, 4)
    prow, first = _first_occurrence_rows(pack_keys(coords))
    n = len(first)
    counts = np.bincount(prow, minlength=n).astype(np.int64)
    return coords[first], prow, counts


def input_layer_fwd(feats: torch.Tensor, prow: np.ndarray, n_active: int, mode: int) -> torch.Tensor:
    """mode 0 copy (unique guaranteed) / 1 last / 2 first / 3 sum / 4 mean."""
    feats = feats.detach()
    idx = torch.from_numpy(np.asarray(prow, dtype=np.int64))
    C = feats.shape[1]
    if mode in (3, 4):
        acc = torch.zeros(n_active, C, dtype=torch.float64)
        acc.index_add_(0, idx, feats.double())
        if mode == 4:
            cnt = torch.bincount(idx, minlength=n_active).clamp_(min=1).double()
            acc /= cnt[:, None]
        return acc.to(feats.dtype)
    out = torch.zeros(n_active, C, dtype=feats.dtype)
    if mode == 0:
        assert len(np.unique(prow)) == len(prow), "mode 0 requires unique coordinates"
        out[idx] = feats
    elif mode == 1:                                   # last writer wins
        last = np.full(n_active, -1, np.int64)
        np.maximum.at(last, prow, np.arange(len(prow)))
        out = feats[torch.from_numpy(last)]
    elif mode == 2:                                   # first writer wins
        first = np.full(n_active, len(prow), np.int64)
        np.minimum.at(first, prow, np.arange(len(prow)))
        out = feats[torch.from_numpy(first)]
    else:
        raise ValueError(mode)
    return out


def input_layer_bwd(dY: torch.Tensor, prow: np.ndarray, mode: int) -> torch.Tensor:
    idx = torch.from_numpy(np.asarray(prow, dtype=np.int64))
    n_active = dY.shape[0]
    g = dY[idx]
    if mode == 4:
        cnt = torch.bincount(idx, minlength=n_active).clamp_(min=1).to(dY.dtype)
        g = g / cnt[idx][:, None]
    elif mode in (1, 2):
        ar = np.arange(len(prow))
        if mode == 1:
            win = np.full(n_active, -1, np.int64); np.maximum.at(win, prow, ar)
        else:
            win = np.full(n_active, len(prow), np.int64); np.minimum.at(win, prow, ar)
        keep = torch.from_numpy(win[prow] == ar)
        g = g * keep[:, None].to(g.dtype)
    return g


def output_layer_fwd(X: torch.Tensor, prow: np.ndarray) -> torch.Tensor:
    return X[torch.from_numpy(np.asarray(prow, dtype=np.int64))]


def output_layer_bwd(dY: torch.Tensor, prow: np.ndarray, n_active: int) -> torch.Tensor:
    acc = torch.zeros(n_active, dY.shape[1], dtype=torch.float64)
    acc.index_add_(0, torch.from_numpy(np.asarray(prow, dtype=np.int64)), dY.double())
    return acc.to(dY.dtype)


# --------------------------------------------------------------------------
# A5 submanifold rulebook, A6/A7 strided rulebook
# --------------------------------------------------------------------------

def _lookup(sorted_keys, sorted_rows, q):
    pos = np.searchsorted(sorted_keys, q)
    pos[pos >= len(sorted_keys)] = 0
    hit = sorted_keys[pos] == q
    return np.where(hit, sorted_rows[pos], -1)


def subm_rulebook(coords: np.ndarray, k: int = 3):
    """Active coords [N,4] -> (nbr [k^3, N] int32, rules list of (in,out) int32 arrays per offset).

    Offset o = ((dx+h)*k + (dy+h))*k + (dz+h), x slowest (Appendix B).  nbr[o, r]
    is the row of site coords[r] + delta_o or -1.  rules[o] holds the pairs with
    nbr >= 0, out ascending.
    """
    coords = np.asarray(coords, dtype=np.int64).reshape(-1, 4)
    n = len(coords)
    h = k // 2
    nbr = np.full((k ** 3, n), -1, np.int32)
    if n:
        keys = pack_keys(coords)
        order = np.argsort(keys, kind="stable")
        sk, sr = keys[order], order.astype(np.int64)
        o = 0
        for dx in range(-h, h + 1):
            for dy in range(-h, h + 1):
                for dz in range(-h, h + 1):
                    q = coords + np.array([dx, dy, dz, 0])
                    ok = ((q[:, :3] >= 0) & (q[:, :3] < 65536)).all(1)
                    qk = pack_keys(np.where(ok[:, None], q, 0))
                    r = _lookup(sk, sr, qk)
                    nbr[o] = np.where(ok, r, -1)
                    o += 1
    rules = []
    for o in range(k ** 3):
        out = np.nonzero(nbr[o] >= 0)[0].astype(np.int32)
        rules.append((nbr[o][out].astype(np.int32), out))
    return nbr, rules


def strided_rulebook(coords: np.ndarray, s=2, existing=None):
    """Fine coords [Nf,4] -> dict(coarse coords, parent, off, child table, rules).
    `existing` = the coords [Nc,4] of a coarse grid the Metadata already holds (grids are keyed by spatial size upstream: a
    second path to the same size lands in the same grid): rows are numbered as THAT grid numbers them.

    size = stride = s (an int, or one entry per axis: `get_downsampler(stride=...)`, module_factory.py:221-241).
    Coarse site = floor(p/s); offset o = ((x%sx)*sy + y%sy)*sz + z%sz.
    Coarse rows numbered by first occurrence scanning fine rows ascending.
    rules[o] = (fine rows, coarse rows), coarse ascending.
    """
    st = np.asarray((s, s, s) if np.isscalar(s) else tuple(s), dtype=np.int64)
    coords = np.asarray(coords, dtype=np.int64).reshape(-1, 4)
    cc = coords.copy()
    cc[:, :3] //= st
    if existing is None:
        parent, first = _first_occurrence_rows(pack_keys(cc))
        nc, coarse = len(first), cc[first]
    else:
        coarse = np.asarray(existing, dtype=np.int64).reshape(-1, 4)
        ek = pack_keys(coarse)
        order = np.argsort(ek, kind="stable")
        parent = _lookup(ek[order], order.astype(np.int64), pack_keys(cc))
        assert (parent >= 0).all(), "the existing grid lacks sites this layer needs"
        nc = len(coarse)
    r = coords[:, :3] % st
    off = ((r[:, 0] * st[1] + r[:, 1]) * st[2] + r[:, 2]).astype(np.int32)
    n_off = int(st.prod())
    child = np.full((n_off, nc), -1, np.int32)
    child[off, parent] = np.arange(len(coords), dtype=np.int32)
    rules = []
    for o in range(n_off):
        out = np.nonzero(child[o] >= 0)[0].astype(np.int32)
        rules.append((child[o][out].astype(np.int32), out))
    return dict(coords=coarse, parent=parent.astype(np.int32), off=off, child=child, rules=rules)


def rules_concat(rules):
    """list of (in,out) -> (pairs [P,2] int32 offset-major, prefix [n_off+1] int64)."""
    prefix = np.zeros(len(rules) + 1, np.int64)
    for o, (i, _) in enumerate(rules):
        prefix[o + 1] = prefix[o] + len(i)
    if prefix[-1] == 0:
        return np.zeros((0, 2), np.int32), prefix
    pairs = np.concatenate([np.stack([i, j], 1) for i, j in rules]).astype(np.int32)
    return pairs, prefix


# --------------------------------------------------------------------------
# A5-A7, A9 feature arithmetic: gather -> sgemm -> scatter-add
# --------------------------------------------------------------------------

def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a).astype(np.int64))


def conv_fwd(X, rules, W, b, n_out):
    """Y = 1 b^T; for o ascending: Y[out] += X[in] W[o].   W [n_off, Cin, Cout]."""
    Y = torch.zeros(n_out, W.shape[2], dtype=X.dtype)
    if b is not None:
        Y += b
    for o, (i, j) in enumerate(rules):
        if len(i):
            Y.index_add_(0, _t(j), X[_t(i)] @ W[o])
    return Y


def conv_bwd(X, dY, rules, W, has_bias=True):
    """Returns dX [N_in,Cin], dW like W, db [Cout] or None."""
    dX = torch.zeros_like(X)
    dW = torch.zeros_like(W)
    for o, (i, j) in enumerate(rules):
        if len(i):
            ti, tj = _t(i), _t(j)
            g = dY[tj]
            dX.index_add_(0, ti, g @ W[o].t())
            dW[o] = X[ti].t() @ g
    db = dY.sum(0) if has_bias else None
    return dX, dW, db


def swap_rules(rules):
    """Deconvolution uses the encoder's rulebook with roles swapped (Appendix B)."""
    return [(j, i) for (i, j) in rules]


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, W, b, rules, n_out):
        ctx.rules = rules
        ctx.has_bias = b is not None
        ctx.save_for_backward(X, W)
        return conv_fwd(X, rules, W, b, n_out)

    @staticmethod
    def backward(ctx, dY):
        X, W = ctx.saved_tensors
        dX, dW, db = conv_bwd(X, dY.contiguous(), ctx.rules, W, ctx.has_bias)
        return dX, dW, db, None, None


def conv(X, W, b, rules, n_out):
    return _ConvFn.apply(X, W, b, rules, n_out)


def batchnorm_relu_fwd(X, gamma, beta, running_mean, running_var, eps=1e-4, momentum=0.9,
                       leak=0.0, training=True):
    """A8.  momentum is the RETAIN fraction (SparseConvNet convention, SURVEY §4.1 caveat).
    The normalisation uses the biased batch variance (divide by N); the running_var update takes the UNBIASED
    estimate (divide by N - 1) -- [UPSTREAM-SCN] BatchNormalization forward as recalled, also torch.nn.BatchNorm's
    convention (the dense twin the reference pairs this layer with, module_factory.py:101-112)."""
    if training:
        mean = X.mean(0)
        var = X.var(0, unbiased=False)
        n = X.shape[0]
        running_mean.mul_(momentum).add_(mean.detach() * (1 - momentum))
        running_var.mul_(momentum).add_(var.detach() * ((1 - momentum) * (n / (n - 1) if n > 1 else 1.0)))
    else:
        mean, var = running_mean, running_var
    y = (X - mean) / torch.sqrt(var + eps) * gamma + beta
    return torch.where(y > 0, y, y * leak)


def pool_fwd(X, child, average):
    """MaxPooling / AveragePooling, pool size = stride, on the child table [n_off, Nc] ([UPSTREAM-SCN]; the
    reference's poolings are 2^3 / 2: n_off = 8):
    max: zero-initialised output, max over existing children;  avg: sum over existing children / n_off."""
    nc = child.shape[1]
    n_off = child.shape[0]
    Y = torch.zeros(nc, X.shape[1], dtype=X.dtype)
    for o in range(n_off):
        rows = np.nonzero(child[o] >= 0)[0]
        if len(rows):
            src = X[_t(child[o][rows])]
            if average:
                Y.index_add_(0, _t(rows), src / float(n_off))
            else:
                Y[_t(rows)] = torch.maximum(Y[_t(rows)], src)
    return Y


def split_batch(X, coords, batch_size):
    """custom_operations.py:24-39: per sample b the rows whose batch column equals b, in row order (a [B, N] bool mask
    `arange(B)[:, None] == coords[:, -1]` and one boolean index per sample)."""
    b = torch.from_numpy(np.asarray(coords)[:, -1].astype(np.int64))
    mask = torch.arange(batch_size).unsqueeze(1) == b
    return [X[m] for m in mask]


def global_pool(X, coords, batch_size, pooling_function=torch.mean):
    """custom_operations.py:42-59 SparseGlobalPool.forward: stack of pooling_function(rows of sample b, dim=0), zeros for
    a sample without rows, `features[:0]` for a batch of zero samples."""
    parts = split_batch(X, coords, batch_size)
    if parts:
        return torch.stack([pooling_function(f, dim=0) if len(f) else f.new_zeros((f.shape[1])) for f in parts])
    return X[:0]


def sparse_to_dense(X, coords, spatial_size, batch_size):
    """A13: zeros [B,C,X,Y,Z]; out[b,:,x,y,z] = X[row]."""
    sx, sy, sz = (int(v) for v in spatial_size)
    out = torch.zeros(batch_size, X.shape[1], sx, sy, sz, dtype=X.dtype)
    c = torch.from_numpy(np.asarray(coords, dtype=np.int64))
    out[c[:, 3], :, c[:, 0], c[:, 1], c[:, 2]] = X
    return out


# --------------------------------------------------------------------------
# A11 sparse ROI crop   (roi_select_sparse.py:125-180, roi_select_bbox_transform.py:56-70,
#                        ndsis/utils/bbox.py:87-106)
# --------------------------------------------------------------------------

def round_boxes(boxes: np.ndarray) -> np.ndarray:
    """fp boxes [BB,2,3] -> int64 (floor start, ceil stop)   (bbox.py:87-106)."""
    boxes = np.asarray(boxes)
    return np.stack([np.floor(boxes[:, 0]), np.ceil(boxes[:, 1])], 1).astype(np.int64)


def transform_boxes(bbox_batch, spatial_size=None, clip=False, resize=None):
    """BBoxTransformerSlice.forward (roi_select_bbox_transform.py:56-70,87-97): list of fp [n_i,2,3] ->
    (int64 [BB,2,3] floor/ceil (optionally clipped: start to [0,S-1], stop to [1,S], bbox.py:62-84),
    per-sample counts, per-box sample index).  resize: the Divider's value (:15-21), fp32 division before rounding."""
    counts = [len(b) for b in bbox_batch]
    raw = np.concatenate([np.asarray(b, dtype=np.float32).reshape(-1, 2, 3) for b in bbox_batch]) \
        if bbox_batch else np.zeros((0, 2, 3), np.float32)
    if resize is not None:
        raw = (raw / np.asarray(resize, dtype=np.float32)).astype(np.float32)
    boxes = round_boxes(raw)
    if clip:
        s = np.asarray(spatial_size, dtype=np.int64)
        boxes[:, 0] = np.clip(boxes[:, 0], 0, s - 1)
        boxes[:, 1] = np.clip(boxes[:, 1], 1, s)
    assoc = np.repeat(np.arange(len(counts), dtype=np.int64), counts)
    return boxes, counts, assoc


def roi_crop(coords: np.ndarray, boxes_int: np.ndarray, box_sample: np.ndarray):
    """coords int64 [N,4], boxes_int int64 [BB,2,3], box_sample int64 [BB].

    Returns (src_row [M] int64 box-major then ascending point row, box_of [M], is_inside [BB,N] bool).
    Output coords = (x,y,z,box index); output feats = feats[src_row]
    (roi_select_sparse.py:125-149,157-180).
    """
    coords = np.asarray(coords, dtype=np.int64).reshape(-1, 4)
    bb = len(boxes_int)
    if bb == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros((0, len(coords)), bool)
    start = np.concatenate([boxes_int[:, 0], box_sample[:, None]], 1)          # [BB,4]
    stop = np.concatenate([boxes_int[:, 1], box_sample[:, None] + 1], 1)
    inside = ((start[:, None, :] <= coords[None]) & (coords[None] < stop[:, None, :])).all(-1)
    box_of, src_row = np.nonzero(inside)                                       # row-major = box-major
    return src_row.astype(np.int64), box_of.astype(np.int64), inside


# --------------------------------------------------------------------------
# A12 the benchmark topology, on the oracle ops (used as checker and CPU baseline)
# --------------------------------------------------------------------------

class OracleScene:
    """Index state for one batch: levels of active coords + cached rulebooks."""

    def __init__(self, coords_pts: np.ndarray):
        self.coords0, self.prow, self.counts = input_layer_rules(coords_pts)
        self.level_coords = [self.coords0]
        self.subm = {}
        self.strided = {}

    def subm_rules(self, level, k=3):
        key = (level, k)
        if key not in self.subm:
            self.subm[key] = subm_rulebook(self.level_coords[level], k)[1]
        return self.subm[key]

    def strided_rules(self, level):
        if level not in self.strided:
            rb = strided_rulebook(self.level_coords[level], 2)
            self.strided[level] = rb
            if len(self.level_coords) == level + 1:
                self.level_coords.append(rb["coords"])
        return self.strided[level]["rules"]

    def n(self, level):
        return len(self.level_coords[level])


def unet_param_shapes(cin, channels, identity_first=False, min_channels=0):
    """Ordered (name, shape) list for the A12 U-Net (SURVEY Appendix A.1 layer list).  identity_first: encoder level 0
    is the reference's FLD('I') (no layer; the mask head's internal U-Net, scannet_config/run.py:756-775).
    min_channels: decoder level l comes up max(channels[l], min_channels) wide (module_factory.py:789-804, 533-578):
    Deconvolution(-> d), NetworkInNetwork(d + channels[l] -> d), units(d)."""
    shapes = []
    L = len(channels)
    for l, c in enumerate(channels):
        if l == 0 and identity_first:
            continue
        if l == 0:
            shapes += [(f"enc{l}.in.weight", (1, cin, c)), (f"enc{l}.in.bias", (c,))]
        else:
            shapes += [(f"enc{l}.in.weight", (8, channels[l - 1], c)), (f"enc{l}.in.bias", (c,))]
        for u in range(2):
            for v in range(2):
                shapes += [(f"enc{l}.res{u}.conv{v}.weight", (27, c, c)), (f"enc{l}.res{u}.conv{v}.bias", (c,))]
    cup = channels[-1]
    for l in range(L - 2, -1, -1):
        c, d = channels[l], max(channels[l], min_channels)
        shapes += [(f"dec{l}.up.weight", (8, cup, d)), (f"dec{l}.up.bias", (d,))]
        shapes += [(f"dec{l}.nin.weight", (d + c, d)), (f"dec{l}.nin.bias", (d,))]
        for u in range(2):
            for v in range(2):
                shapes += [(f"dec{l}.res{u}.conv{v}.weight", (27, d, d)), (f"dec{l}.res{u}.conv{v}.bias", (d,))]
        cup = d
    return shapes


def init_unet_params(cin, channels, seed=0, identity_first=False):
    """N(0, sqrt(2/(Cin*k^3))) weights, small random biases so bias paths are exercised."""
    g = torch.Generator().manual_seed(seed)
    params = {}
    for name, shape in unet_param_shapes(cin, channels, identity_first):
        if name.endswith("weight"):
            fan = shape[-2] * (shape[0] if len(shape) == 3 else 1)
            params[name] = torch.randn(shape, generator=g) * (2.0 / fan) ** 0.5
        else:
            params[name] = torch.randn(shape, generator=g) * 0.01
    return params


class FrozenReLU:
    """ReLU with PRESCRIBED sign masks (tests: the masks the HIP forward recorded, sparse_rcnn_amd.functional.RELU_RECORD):
    call k multiplies its argument by mask k instead of deciding `x > 0` itself.  Two evaluations of a ReLU network that
    share their masks compute the same piecewise-linear function, so inputs within rounding of zero cannot flip a row's
    contribution in one of them only -- gradients then differ by summation order alone.  A mask may be wider than the
    argument (slabs padded with zero columns on the device)."""

    def __init__(self, masks):
        self.masks, self.k = list(masks), 0

    def __call__(self, x):
        m = self.masks[self.k]
        self.k += 1
        if m.shape[0] != x.shape[0] or m.shape[1] < x.shape[1]:
            raise ValueError(f"FrozenReLU: mask {self.k - 1} has shape {tuple(m.shape)}, argument {tuple(x.shape)}")
        return x * m[:, :x.shape[1]].to(x.dtype)


def unet_forward(scene: OracleScene, feats_pts: torch.Tensor, params: dict, channels, identity_first=False,
                 storage=None, tile_weights=None, split_nin=False, record=None, relu=None, interims=None):
    """A12: encoder level = {SubM1 | Conv2s2} + 2x[x + SubM3(ReLU(SubM3(ReLU(x))))];
    decoder level = ReLU -> Deconv2s2 -> Join(up, skip) -> NiN -> 2x residual
    (module_factory.py:127-183, 513-578; custom_container.py:70-83: cat((upsampled, skip))).
    identity_first: encoder level 0 has no layer (FLD('I')).  storage: optional rounding applied to every stored
    feature slab after the first layer (straight-through in backward) -- the bf16 STORAGE mode of the HIP path restated
    on the CPU: storage=bf16_storage.  tile_weights: rounding applied to the weights of the layers that run
    on the bf16 tile kernel in that mode (SubM 3^3, Convolution: their LDS image is bf16; the 1x1 / deconvolution GEMMs
    keep fp32 weights).  split_nin: the NetworkInNetwork over a JoinTable as the HIP path evaluates it -- one GEMM per
    joined part, the first partial result stored (rounded) before the second is added.
    relu: replaces torch.relu (FrozenReLU: prescribed sign masks, consumed in the order the network applies its ReLUs).
    interims: a list that receives the encoder outputs (differentiable: the RPN's inputs, model.py:293-298)."""
    P = params
    relu = torch.relu if relu is None else relu        # relu: a FrozenReLU (prescribed masks), default the real one
    q = storage if storage is not None else (lambda t: t)
    wq = tile_weights if tile_weights is not None else (lambda t: t)
    x = _InputFn.apply(feats_pts, scene)
    skips = []
    L = len(channels)

    def residual(x, prefix, level):
        rules = scene.subm_rules(level, 3)
        n = scene.n(level)
        for u in range(2):
            y = q(conv(relu(x), wq(P[f"{prefix}.res{u}.conv0.weight"]), P[f"{prefix}.res{u}.conv0.bias"], rules, n))
            y = conv(relu(y), wq(P[f"{prefix}.res{u}.conv1.weight"]), P[f"{prefix}.res{u}.conv1.bias"], rules, n)
            x = q(x + y)
        return x

    for l in range(L):
        if l == 0 and identity_first:
            x = q(x)                           # the level's slab is stored (bf16 storage: cast) before anything reads it
            skips.append(x)
            continue
        if l == 0:
            ident = [(np.arange(scene.n(0), dtype=np.int32),) * 2]
            x = q(conv(x, P["enc0.in.weight"], P["enc0.in.bias"], ident, scene.n(0)))
        else:
            rules = scene.strided_rules(l - 1)
            x = q(conv(x, wq(P[f"enc{l}.in.weight"]), P[f"enc{l}.in.bias"], rules, scene.n(l)))
        if record is not None:
            record.append((f"enc{l}.head", x.detach()))
        x = residual(x, f"enc{l}", l)
        skips.append(x)
        if record is not None:
            record.append((f"enc{l}", x.detach()))
    if interims is not None:
        interims.extend(skips)
    for l in range(L - 2, -1, -1):
        rules = swap_rules(scene.strided_rules(l))
        up = q(conv(relu(x), P[f"dec{l}.up.weight"], P[f"dec{l}.up.bias"], rules, scene.n(l)))
        Wn = P[f"dec{l}.nin.weight"]
        if split_nin:
            c_up = up.shape[1]
            x = q(skips[l] @ Wn[c_up:] + q(up @ Wn[:c_up] + P[f"dec{l}.nin.bias"]))
        else:
            x = q(torch.cat([up, skips[l]], 1) @ Wn + P[f"dec{l}.nin.bias"])
        x = residual(x, f"dec{l}", l)
        if record is not None:
            record.append((f"dec{l}", x.detach()))
    return x


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def bf16_storage(t):
    """Round to bf16 and widen back (value as stored by the bf16 storage path); gradient passes straight through."""
    return _RoundBF16.apply(t)



class _InputFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, scene):
        ctx.scene = scene
        return input_layer_fwd(feats, scene.prow, scene.n(0), 4)

    @staticmethod
    def backward(ctx, dY):
        return input_layer_bwd(dY, ctx.scene.prow, 4), None


# --------------------------------------------------------------------------
# N2: mask-head epilogue over the ROI selection
# (ndsis/modules/model.py:824-882 SparseMaskPredictor, :1150-1227 SparseMaskLossSelector;
#  ndsis/utils/basic_functions.py:177-216 split_select_nd)
# --------------------------------------------------------------------------

def mask_predict(mask_output, is_inside, box_sample_count, batch_splits, class_indices, num_valid=0):
    """is_inside bool [BB, N]; mask_output fp32 [M, K] with rows box-major / ascending point.
    -> list per sample of fp32 [boxes_s, points_s] (model.py:859-882)."""
    mo = np.asarray(mask_output, dtype=np.float32)
    ins = np.asarray(is_inside, dtype=bool)
    out, r, b0, p0 = [], 0, 0, 0
    for nb, npts in zip(box_sample_count, batch_splits):
        blk = np.zeros((nb, npts), np.float32)
        for j in range(nb):
            pts = np.nonzero(ins[b0 + j])[0]
            c = int(class_indices[b0 + j])
            valid = c >= 0 and (num_valid == 0 or c < num_valid)
            if valid:
                x = mo[r:r + len(pts), c].astype(np.float64)
                blk[j, pts - p0] = (1.0 / (1.0 + np.exp(-x))).astype(np.float32)
            r += len(pts)
        out.append(blk)
        b0 += nb
        p0 += npts
    return out


def mask_loss_select(mask_scores, is_inside, box_sample_count, batch_splits, keep_list, gt_associations_list,
                     gt_labels_list, gt_masks_list):
    """-> (pred flat, gt flat, rows per kept box, labels) in crop order over the kept boxes (model.py:1157-1227)."""
    ms = np.asarray(mask_scores, dtype=np.float32)
    ins = np.asarray(is_inside, dtype=bool)
    pred, gt, rows, labels = [], [], [], []
    r, b0, p0 = 0, 0, 0
    for s, (nb, npts) in enumerate(zip(box_sample_count, batch_splits)):
        keep = np.asarray(keep_list[s], dtype=bool)
        assoc = np.asarray(gt_associations_list[s], dtype=np.int64)
        a = 0
        for j in range(nb):
            pts = np.nonzero(ins[b0 + j])[0]
            if keep[j]:
                g = int(assoc[a]); a += 1
                lab = int(np.asarray(gt_labels_list[s])[g])
                pred.append(ms[r:r + len(pts), lab])
                gt.append(np.asarray(gt_masks_list[s], dtype=np.float32)[g, pts - p0])
                rows.append(len(pts)); labels.append(lab)
            r += len(pts)
        b0 += nb
        p0 += npts
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.float32)
    return cat(pred), cat(gt), rows, np.asarray(labels, np.int64)


# --------------------------------------------------------------------------
# N3: greedy NMS of score-sorted boxes (ndsis/utils/bbox.py:713-759, IoU :205-242, :598-620)
# --------------------------------------------------------------------------

def nms(boxes, thr):
    """boxes fp32 [N, 2, 3] sorted by descending confidence -> bool [N].  fp32 arithmetic in the reference's order."""
    b = np.asarray(boxes, dtype=np.float32).reshape(-1, 2, 3)
    n = len(b)
    size = b[:, 1] - b[:, 0]
    vol = (size[:, 0] * size[:, 1]) * size[:, 2]
    keep = np.ones(n, dtype=bool)
    thr = np.float32(thr)
    for j in range(n):
        if not keep[j]:
            continue
        lo = np.maximum(b[j, 0], b[j + 1:, 0])
        hi = np.minimum(b[j, 1], b[j + 1:, 1])
        e = np.maximum(hi - lo, np.float32(0))
        inter = (e[:, 0] * e[:, 1]) * e[:, 2]
        union = (vol[j] + vol[j + 1:]) - inter
        with np.errstate(divide="ignore", invalid="ignore"):
            ov = inter / union
        keep[j + 1:] &= ~(ov > thr)
    return keep


# --------------------------------------------------------------------------
# N4: voxelisation (ndsis/data/sparse_augmentation.py:81-126 augment_coords, :42-47 fix_cut_out,
#     :50-78 random_cut_out given its drawn start positions)
# --------------------------------------------------------------------------

def augment_coords(coords, rot_and_scale, sub_pixel_offset, spatial_size=None, shift=None, start_positions=None):
    """-> (resulting int64 [M,3], is_inside bool [N], spatial_size int64 [3], complete_shift fp32 [3]).
    points @ R in fp32 as fma(z, R2j, fma(y, R1j, x*R0j)) (torch's CPU association for K = 3; the fused multiply-adds are
    emulated in float64: the product of two fp32 values is exact there)."""
    p = np.asarray(coords, dtype=np.float32)
    r = np.asarray(rot_and_scale, dtype=np.float32).reshape(3, 3)
    aug = np.empty_like(p)
    for j in range(3):
        acc = (p[:, 0] * r[0, j]).astype(np.float32)
        for k in (1, 2):
            acc = (p[:, k].astype(np.float64) * np.float64(r[k, j]) + acc.astype(np.float64)).astype(np.float32)
        aug[:, j] = acc
    complete_shift = (-aug.min(0) + np.asarray(sub_pixel_offset, dtype=np.float32)).astype(np.float32)
    discrete = (aug + complete_shift).astype(np.float32).astype(np.int64)          # .long(): truncation
    if spatial_size is not None:
        size = np.broadcast_to(np.asarray(spatial_size, dtype=np.int64), (3,))
        if shift is not None:
            start = np.broadcast_to(-np.asarray(shift, dtype=np.int64), (3,))
            inside = ((0 <= discrete) & (discrete < size)).all(1)
        else:
            start = np.broadcast_to(np.asarray(start_positions, dtype=np.int64), (3,))
            moved = discrete - start
            inside = ((0 <= moved) & (moved < size)).all(1)
        res = (discrete - start)[inside]
        complete_shift = complete_shift - start.astype(np.float32)
        return res, inside, size.copy(), complete_shift
    size = discrete.max(0)
    res = discrete
    if shift is not None:
        sh = np.broadcast_to(np.asarray(shift, dtype=np.int64), (3,))
        res = discrete + sh
        size = size + 2 * sh
        complete_shift = complete_shift + sh.astype(np.float32)
    return res, np.ones(len(p), bool), size, complete_shift
