"""Developer tool (GPU box): time the conv kernels on one level of the cfg-2 scene.
    python tools/ablate_conv.py [level]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.synthetic import make_batch

level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
C = [32, 64, 128, 256][level]
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata
sz = tuple(int(s) for s in size)
for l in range(level):
    md.strided_rulebook(sz); sz = tuple(s // 2 for s in sz)
rb = md.subm_rulebook(sz, 3)
n, P, t = rb.n, rb.rules.total, rb.tiles
X = torch.randn(n, C, device="cuda"); W = torch.randn(27, C, C, device="cuda") * 0.05
Y = torch.empty(n, C, device="cuda"); Y2 = torch.empty(n, C, device="cuda")
lib = L.lib()
FOLD = int(os.environ.get("FOLD", "0"))          # FOLD=1024: every gather lands on the first 1024 rows (cache-resident)
if FOLD:
    ts = t.tstab.clone(); ts[ts >= 0] %= FOLD
    t.tstab = ts
    print("gathers folded onto", FOLD, "rows (results differ from the table kernel by construction)")
SCR = torch.empty(max(1, lib.scn_conv_tiles_scratch_bytes(C, n, C)), dtype=torch.uint8, device='cuda')
tm = t.tile_mask.cpu().numpy().view("uint32")
import numpy as np
execd = sum(bin(int(v)).count("1") for v in tm) * 16
print(f"level {level} N={n} P={P} C={C} tiles={len(tm)} executed/useful={execd / P:.3f}")
ARR = torch.zeros(max(1, lib.scn_conv_tiles_arrival_counters(C, n, C)), dtype=torch.int32, device="cuda")
def run_ts(fl=0, fused=True):
    L.check(lib.scn_conv_tiles(L.ptr(X), n, C, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), L.ptr(t.tile_order), 27, n, L.ptr(W), 0, 0, 0,
                               L.ptr(Y), C, fl, L.ptr(SCR), L.ptr(ARR) if fused else 0, L.stream()))
def run_tab(fl=0):
    L.check(lib.scn_gemm_table(L.ptr(X), n, C, L.ptr(rb.table), 27, n, L.ptr(W), 0, 0, 0, L.ptr(Y2), C, fl, L.stream()))
for name, fn in (("conv_tiles", run_ts), ("conv_tiles 2-launch K sum", lambda: run_ts(0, False)), ("conv_tiles tile kernel only", lambda: run_ts(16, False)),
                 ("conv_tiles relu_in", lambda: run_ts(1)), ("conv_tiles W^T", lambda: run_ts(6)), ("gemm_table (v1)", run_tab)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    print(f"{name:20s} {us:8.1f} us   {2.0 * P * C * C / us / 1e6:6.1f} TF useful   {2.0 * execd * C * C / us / 1e6:6.1f} TF executed")
run_ts(); run_tab(); torch.cuda.synchronize()
print("max |ts - table| =", (Y - Y2).abs().max().item())
