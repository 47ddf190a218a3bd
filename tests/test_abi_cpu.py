"""CPU: the C-ABI shared library loads without a GPU and exports every symbol include/scn_mi355x.h declares; the
product path refuses to compute without its extension / without a GPU (no CPU fallback)."""
import ctypes
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "scn_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(scn_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    fns = declared_functions()
    for must in ("scn_dedup_build", "scn_subm_table", "scn_child_table", "scn_rules_scan", "scn_rules_fill",
                 "scn_tiles_build", "scn_conv_tiles", "scn_gemm_table", "scn_gemm_rules", "scn_wgrad_rules",
                 "scn_bn_fwd", "scn_bn_bwd", "scn_input_fwd", "scn_input_bwd", "scn_gather_rows", "scn_segment_sum",
                 "scn_sparse_to_dense_fwd", "scn_roi_count", "scn_roi_fill", "scn_roi_inside", "scn_roi_boxes", "scn_roi_coords"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()                      # hipcc cross-compiles gfx950 without a GPU; no-op when up to date
    import sparse_rcnn_amd as scn
    lib = ctypes.CDLL(scn.LIB_PATH)
    missing = [f for f in declared_functions() if not hasattr(lib, f)]
    assert not missing, missing
    # and the Python binding table covers the header
    assert sorted(scn.EXPORTS) == declared_functions()
    loaded = scn.load_library()
    assert loaded.scn_abi_version() == 5          # include/scn_mi355x.h: history of the ABI
    assert loaded.scn_hash_capacity(1000) == 2048 and loaded.scn_hash_capacity(0) == 1024
    assert loaded.scn_rules_blocks(27, 5000) == 27 * 5
    # the step executor's plan records: the ctypes structures of executor.py have the C layout
    from sparse_rcnn_amd import executor as EX
    assert loaded.scn_exec_struct_bytes(0) == ctypes.sizeof(EX.ExecOp) == 64
    assert loaded.scn_exec_struct_bytes(1) == ctypes.sizeof(EX.ExecLevel) == 136


def test_executor_plans_compile_without_a_gpu():
    """The launch plans of the step executor are compiled from the module tree alone (no GPU): per level one stage whose op
    lists name only buffers / parameters / gradient regions the stage declares, and every parameter of the network sits
    in exactly one gradient region of exactly one stage."""
    from sparse_rcnn_amd import executor as EX
    from sparse_rcnn_amd.maskhead import MaskBranch
    from sparse_rcnn_amd.unet import SparseUNet
    for kw in (dict(), dict(bf16_blocks="all")):
        net = SparseUNet(7, (32, 48, 64, 80), **kw)
        plan = net._exec_plan()
        stages = [s for s in plan["enc"] + plan["dec"] if s is not None]
        assert len(stages) == 4 + 3
        seen = []
        for st in stages:
            for op in st.fwd + st.bwd:
                for field in ("x", "y", "r", "m", "x1", "y1"):
                    v = getattr(op, field)
                    assert -1 <= v < len(st.bufs), (field, v)
            for op in st.fwd:
                assert -1 <= op.w < len(st.params) and -1 <= op.b < len(st.params)
                assert op.op in (EX.OP_GEMM_IDENT, EX.OP_CONV_SUBM, EX.OP_CONV_CHILD, EX.OP_RULES_CHILD, EX.OP_ROWS2, EX.OP_CAST)
            for op in st.bwd:
                if op.op in (EX.OP_WGRAD_SUBM, EX.OP_WGRAD2_SUBM, EX.OP_WGRAD_DOWN, EX.OP_WGRAD_UP, EX.OP_WGRAD_IDENT, EX.OP_COLSUM):
                    assert -1 <= op.w < len(st.gregions) and -1 <= op.b < len(st.gregions)
            seen += [m for m, _ in st.mods]
        convs = [m for m in net.modules() if hasattr(m, "weight") and hasattr(m, "nIn")]
        assert len(seen) == len(set(map(id, seen))) == len(convs)
    assert MaskBranch(32, 7).output_conv_layer._exec_plan()["enc"][0] is None       # identity_first: level 0 is no stage
    bn = SparseUNet(7, (16, 32), batchnorm=True)
    assert bn._exec_plan() is False                                                  # batch norm: layer-by-layer path


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_cpu_fallback():
    import sparse_rcnn_amd as scn
    layer = scn.InputLayer(3, torch.tensor([8, 8, 8]), mode=4)
    with pytest.raises(scn.ScnError, match="no CPU fallback"):
        layer((torch.zeros(1, 4, dtype=torch.long), torch.ones(1, 2), 1))
    conv = scn.SubmanifoldConvolution(3, 2, 4, 3, True)
    x = scn.SparseConvNetTensor(torch.ones(1, 2), scn.Metadata(3), torch.tensor([8, 8, 8]))
    with pytest.raises((scn.ScnError, RuntimeError)):
        conv(x)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from sparse_rcnn_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libscn_mi355x.so"))
    with pytest.raises(_lib.ScnError, match="missing"):
        _lib.load()


def test_header_is_plain_c_and_library_links_from_c():
    """include/scn_mi355x.h compiles as C11 and the shared library links into a host with no Python in it
    (tests/c_host/scn_c_host.c; it runs in tests/test_gpu_c_host.py)."""
    import subprocess
    import __graft_entry__ as g
    g.build_c_host(force=True)
    assert os.path.exists(g.C_HOST_BIN)
    out = subprocess.run(["ldd", g.C_HOST_BIN], capture_output=True, text=True).stdout
    assert "libscn_mi355x.so" in out and "not found" not in out, out
    assert "python" not in out.lower() and "torch" not in out.lower(), out


def test_developer_switches_are_read_once_and_change_only_through_the_abi():
    """the library's developer switches no longer follow the ambient environment per launch.  The
    environment is read ONCE (first use of any switch) -- SCN_TB_KH=2 given at process start is seen, a variable exported
    afterwards is not; scn_debug_set / the tests' `_lib.debug_switch` change one; an unknown name is SCN_EINVAL.  And no
    product source calls getenv outside scn_debug.hip."""
    import subprocess
    code = r'''
import ctypes, os, sys
sys.path.insert(0, %r)
from sparse_rcnn_amd import _lib as L
l = L.load()
s, v = ctypes.c_int(0), ctypes.c_int64(0)
assert l.scn_debug_get(b"SCN_TB_KH", ctypes.byref(s), ctypes.byref(v)) == 0 and (s.value, v.value) == (1, 2)
os.environ["SCN_TS_NO_TAIL"] = "1"                                   # after the one read: not seen
assert l.scn_debug_get(b"SCN_TS_NO_TAIL", ctypes.byref(s), ctypes.byref(v)) == 0 and s.value == 0
with L.debug_switch("SCN_TS_NO_TAIL", 1):
    l.scn_debug_get(b"SCN_TS_NO_TAIL", ctypes.byref(s), ctypes.byref(v)); assert (s.value, v.value) == (1, 1)
    with L.debug_switch("SCN_TS_NO_TAIL", None):
        l.scn_debug_get(b"SCN_TS_NO_TAIL", ctypes.byref(s), ctypes.byref(v)); assert s.value == 0
    l.scn_debug_get(b"SCN_TS_NO_TAIL", ctypes.byref(s), ctypes.byref(v)); assert (s.value, v.value) == (1, 1)
l.scn_debug_get(b"SCN_TS_NO_TAIL", ctypes.byref(s), ctypes.byref(v)); assert s.value == 0
L.switches["SCN_TS_SPLIT"] = "0"
l.scn_debug_get(b"SCN_TS_SPLIT", ctypes.byref(s), ctypes.byref(v)); assert (s.value, v.value) == (1, 0)
del L.switches["SCN_TS_SPLIT"]
assert l.scn_debug_set(b"SCN_NO_SUCH_SWITCH", b"1") == L.EINVAL and b"SCN_NO_SUCH_SWITCH" in l.scn_last_error_string()
print("OK")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, SCN_TB_KH="2"))
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout, r.stderr[-2000:])
    csrc = os.path.join(ROOT, "sparse_rcnn_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".inc", ".h")) and f != "scn_debug.hip":
            with open(os.path.join(csrc, f)) as fh:
                assert "getenv(" not in fh.read(), f
