"""Per-op table of a step's executor launches (HIP events inside the C calls, scn_exec_timing_enable(2)): every op kind -- tile
convolutions, weight gradients, row GEMMs, casts -- by (op, Cin -> Cout, level rows): launches per step, us per launch, us per step.
    python tools/exec_launch_table.py [cfg2|cfg3|ref|...] [f32|bf16]"""
import collections
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.trainstep import SceneStep

NAMES = {1: "gemm_ident", 2: "conv_subm", 3: "conv_child", 4: "rules_child", 5: "rows2", 6: "wgrad_subm", 7: "wgrad2_subm",
         8: "wgrad_down", 9: "wgrad_up", 10: "wgrad_ident", 11: "colsum", 12: "add", 13: "cast"}
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
job = SceneStep(wl, torch.device("cuda", 0), dtype=dt)
for _ in range(5):
    job.step()
torch.cuda.synchronize()
lib = L.lib()
N = 5
lib.scn_exec_timing_enable(2)
for _ in range(N):
    job.step()
torch.cuda.synchronize()
lib.scn_exec_timing_enable(0)
cap = 16384
ms = (C.c_float * cap)()
info = (C.c_int64 * (7 * cap))()
n = lib.scn_exec_timing_collect(ms, info, cap)
rows = collections.defaultdict(lambda: [0, 0.0])
for k in range(n):
    op, bf16, cin, cout, n_in, n_out, rules = (int(info[7 * k + j]) for j in range(7))
    r = rows[(op, cin, cout, n_in)]
    r[0] += 1
    r[1] += float(ms[k]) * 1e3
job.finish()
tot = sum(r[1] for r in rows.values()) / N
print(f"{wl} {dt}: {n / N:.0f} executor ops per step, {tot / 1e3:.3f} ms per step inside them (events; the deferred unit sums not included)")
by_op = collections.defaultdict(float)
for (op, *_), (c, us) in rows.items():
    by_op[op] += us / N
print("  per op kind, us per step: " + ", ".join(f"{NAMES.get(o, o)} {v:.0f}" for o, v in sorted(by_op.items(), key=lambda kv: -kv[1])))
print("  op            cin->cout  level rows  launches/step  us/launch  us/step")
for (op, cin, cout, n_in), (c, us) in sorted(rows.items(), key=lambda kv: (-kv[0][3], kv[0][0])):
    print(f"  {NAMES.get(op, op):12s} {cin:4d}->{cout:<4d} {n_in:9d}  {c / N:5.1f}  {us / c:8.1f}  {us / N:8.1f}")
