"""Per-shape table of the tile-convolution launches of a step (HIP events inside the executor's C calls, scn_exec_timing_*):
    python tools/conv_launch_table.py [cfg2|cfg3|ref|...] [f32|bf16]
rows: (op, Cin -> Cout, output rows): launches per step, us per launch, useful TFLOP/s."""
import collections
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.trainstep import SceneStep

wl = sys.argv[1] if len(sys.argv) > 1 else "ref"
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
job = SceneStep(wl, torch.device("cuda", 0), dtype=dt)
for _ in range(5):
    job.step()
torch.cuda.synchronize()
lib = L.lib()
N = 5
lib.scn_exec_timing_enable(1)
for _ in range(N):
    job.step()
torch.cuda.synchronize()
lib.scn_exec_timing_enable(0)
cap = 8192
ms = (C.c_float * cap)()
info = (C.c_int64 * (7 * cap))()
n = lib.scn_exec_timing_collect(ms, info, cap)
rows = collections.defaultdict(lambda: [0, 0.0, 0.0])
for k in range(n):
    op, bf16, cin, cout, n_in, n_out, rules = (int(info[7 * k + j]) for j in range(7))
    r = rows[(op, cin, cout, n_out)]
    r[0] += 1
    r[1] += float(ms[k]) * 1e3
    r[2] += 2.0 * rules * cin * cout
job.finish()
tot = sum(r[1] for r in rows.values()) / N
print(f"{wl} {dt}: {n / N:.0f} tile-convolution launches per step, {tot / 1e3:.3f} ms per step inside them")
print("  op(2 subm / 3 child)  cin->cout  rows_out  launches/step  us/launch  us/step  useful TFLOP/s")
for (op, cin, cout, n_out), (c, us, fl) in sorted(rows.items(), key=lambda kv: -kv[0][3]):
    print(f"  {op}  {cin:4d}->{cout:<4d} {n_out:8d}  {c / N:5.1f}  {us / c:8.1f}  {us / N:8.1f}  {fl / us / 1e6:7.1f}")
