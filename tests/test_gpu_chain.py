"""-m gpu: scn_conv_tiles_chain (round 6) -- the dependent SubM 3^3 convolutions of a level's residual units
(ndsis/modules/module_factory.py:127-183, two units per level :513-530) as ONE launch -- must give the bits of the same
convolutions as separate scn_conv_tiles launches (which tests/test_gpu_parity.py and test_gpu_atsize.py hold to the oracle)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


class Role(C.Structure):
    _fields_ = [("X", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p), ("relu_mask", C.c_void_p),
                ("Y", C.c_void_p), ("flags", C.c_int32), ("reserved", C.c_int32)]


def _level(n_target, levels_down):
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd.synthetic import make_batch
    coords, feats, size, bs, _ = make_batch(1, (256, 256, 128), n_target, seed=3)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
    md = x.metadata
    sz = tuple(int(s) for s in size)
    for _ in range(levels_down):
        md.strided_rulebook(sz)
        sz = tuple(s // 2 for s in sz)
    return md.subm_rulebook(sz, 3)


def _plans(n, c, backward, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    X = torch.randn(n, c, device="cuda", generator=g)
    Ws = [torch.randn(27, c, c, device="cuda", generator=g) * (0.3 / c ** 0.5) for _ in range(4)]
    Bs = [torch.randn(c, device="cuda", generator=g) * 0.1 for _ in range(4)]
    Ms = [(torch.rand(n, c, device="cuda", generator=g) > 0.5).float() for _ in range(4)]

    def plan(Y):
        if not backward:        # y1 = conv(relu x); y = x + conv(relu y1); the same again on y
            return [(X, Ws[0], Bs[0], None, None, Y[0], 1), (Y[0], Ws[1], Bs[1], X, None, Y[1], 1),
                    (Y[1], Ws[2], Bs[2], None, None, Y[2], 1), (Y[2], Ws[3], Bs[3], Y[1], None, Y[3], 1)]
        back = 2 | 4            # SCN_F_W_TRANSPOSED | SCN_F_OFF_REVERSE; 8 = SCN_F_RESIDUAL_LAST
        return [(X, Ws[0], None, None, Ms[0], Y[0], back), (Y[0], Ws[1], None, X, Ms[1], Y[1], back | 8),
                (Y[1], Ws[2], None, None, Ms[2], Y[2], back), (Y[2], Ws[3], None, Y[1], Ms[3], Y[3], back | 8)]
    keep = (X, Ws, Bs, Ms)
    return plan, keep


def _run_both(rb, c, backward, n_roles, seed):
    from sparse_rcnn_amd import _lib as L
    lib = L.lib()
    n, t = rb.n, rb.tiles
    plan, keep = _plans(n, c, backward, seed)
    Ya = [torch.full((n, c), float("nan"), device="cuda") for _ in range(4)]
    Yb = [torch.full((n, c), float("nan"), device="cuda") for _ in range(4)]
    scr = torch.empty(max(1, lib.scn_conv_tiles_scratch_bytes(c, n, c)), dtype=torch.uint8, device="cuda")
    arr_t = torch.zeros(max(1, lib.scn_conv_tiles_arrival_counters(c, n, c)), dtype=torch.int32, device="cuda")
    arr = L.ptr(arr_t) if c > 32 else 0
    o = lambda v: L.ptr(v) if v is not None else None
    for (xi, w, b, r, m, y, f) in plan(Ya)[:n_roles]:
        L.check(lib.scn_conv_tiles(L.ptr(xi), n, c, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), L.ptr(t.tile_order), 27, n,
                                   L.ptr(w), o(b) or 0, o(r) or 0, o(m) or 0, L.ptr(y), c, f, L.ptr(scr), arr, L.stream()))
    roles = (Role * n_roles)(*[Role(L.ptr(xi), L.ptr(w), o(b), o(r), o(m), L.ptr(y), f, 0)
                               for (xi, w, b, r, m, y, f) in plan(Yb)[:n_roles]])
    cnt = (C.c_int64 * 2)()
    lib.scn_conv_tiles_chain_counts(cnt, 1)
    reps = 6                                   # repeated launches: the sync words are back at zero every time
    for _ in range(reps):
        L.check(lib.scn_conv_tiles_chain(n_roles, roles, n, c, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm),
                                         L.ptr(t.tile_order), 27, n, c, L.ptr(scr), arr, L.stream()))
    torch.cuda.synchronize()
    lib.scn_conv_tiles_chain_counts(cnt, 1)
    assert int(arr_t.abs().max()) == 0, "arrival counters must be left at zero"
    return Ya[:n_roles], Yb[:n_roles], (cnt[0], cnt[1]), reps, keep


@pytest.mark.parametrize("backward", [False, True])
@pytest.mark.parametrize("c,down,n_roles", [(32, 0, 4), (32, 0, 2), (64, 1, 4), (128, 1, 3)])
def test_chained_launch_reproduces_separate_launches_bit_for_bit(c, down, n_roles, backward):
    rb = _level(60000, down)
    plain, chained, (launches, roles), reps, _ = _run_both(rb, c, backward, n_roles, seed=c + down)
    assert launches == reps and roles == reps * n_roles, "the shapes of this test must take the chained kernel"
    for k, (a, b) in enumerate(zip(plain, chained)):
        assert not torch.isnan(b).any()
        assert torch.equal(a, b), f"role {k}: chained launch differs from the plain launch"


def test_chain_entry_falls_back_to_plain_launches_outside_its_shapes():
    # a small level (the four-waves-per-tile loop) and a channel count that is no multiple of 32: plain launches, same bits
    rb = _level(6000, 1)
    for c in (32, 48):
        plain, chained, (launches, roles), _, _ = _run_both(rb, c, False, 4, seed=7)
        assert launches == 0 and roles == 0
        for a, b in zip(plain, chained):
            assert torch.equal(a, b)


def test_chain_switch_off_runs_plain_launches():
    from sparse_rcnn_amd import _lib as L
    rb = _level(60000, 0)
    with L.debug_switch("SCN_TS_NO_CHAIN", 1):
        plain, chained, (launches, _), _, _ = _run_both(rb, 32, False, 4, seed=1)
    assert launches == 0
    for a, b in zip(plain, chained):
        assert torch.equal(a, b)


@pytest.mark.parametrize("g0", [4, 8, 12])
@pytest.mark.parametrize("c,down", [(32, 0), (64, 1), (128, 2)])
def test_progressive_staging_is_bit_equal(c, down, g0):
    """SCN_TS_PROG (round 6, profiles/r6_prog_staging.txt): the forward weight slice staged by LDS-DMA in offset order, the first
    tile started behind the first g0 offsets -- a different way into LDS, the same arithmetic: same bits, every launch."""
    from sparse_rcnn_amd import _lib as L
    lib = L.lib()
    rb = _level(60000, down)
    n, t = rb.n, rb.tiles
    g = torch.Generator(device="cuda").manual_seed(11 * c + g0)
    X = torch.randn(n, c, device="cuda", generator=g)
    W = torch.randn(27, c, c, device="cuda", generator=g) * (0.3 / c ** 0.5)
    B = torch.randn(c, device="cuda", generator=g) * 0.1
    R = torch.randn(n, c, device="cuda", generator=g)
    scr = torch.empty(max(1, lib.scn_conv_tiles_scratch_bytes(c, n, c)), dtype=torch.uint8, device="cuda")
    arr_t = torch.zeros(max(1, lib.scn_conv_tiles_arrival_counters(c, n, c)), dtype=torch.int32, device="cuda")
    arr = L.ptr(arr_t) if c > 32 else 0

    def run(Y, flags):
        L.check(lib.scn_conv_tiles(L.ptr(X), n, c, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), L.ptr(t.tile_order), 27, n,
                                   L.ptr(W), L.ptr(B), L.ptr(R), 0, L.ptr(Y), c, flags, L.ptr(scr), arr, L.stream()))
    for flags in (0, 1, 4):                    # plain, fused input ReLU, reversed offsets (forward-layout weights)
        Ya = torch.full((n, c), float("nan"), device="cuda")
        Yb = torch.full((n, c), float("nan"), device="cuda")
        run(Ya, flags)
        with L.debug_switch("SCN_TS_PROG", g0):
            for _ in range(5):
                run(Yb, flags)
        torch.cuda.synchronize()
        assert not torch.isnan(Yb).any()
        assert torch.equal(Ya, Yb), f"flags {flags}: progressive staging differs from the classic staging"
