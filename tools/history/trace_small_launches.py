"""Developer tool (GPU box): which host call sites of a cfg-3 step issue the small launches (memcpy, fill, elementwise)?
torch.profiler over a few steps, device-side events grouped by name, host ops grouped by (op, python frame)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from sparse_rcnn_amd.trainstep import SceneStep
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
st = SceneStep(wl, dtype=dt)
for _ in range(5): st.step()
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    for _ in range(N): st.step()
    torch.cuda.synchronize()
st.finish()
ops = collections.Counter(); where = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.device_type.name != "CPU":
        continue
    n = ev.name
    if n.startswith("aten::") and n.split("::")[1] in ("copy_", "fill_", "zero_", "add", "add_", "mul", "mul_", "clone", "to", "_to_copy", "contiguous", "empty", "zeros", "cat", "index_select", "sum", "item", "_local_scalar_dense", "randn", "neg", "sub", "div", "where", "eq", "gt", "lt"):
        ops[n] += 1
        frames = [f for f in (ev.stack or []) if ("sparse_rcnn_amd" in f or "bench.py" in f) and "_lib.py" not in f]
        where[n][" <- ".join(fr.split("/")[-1] for fr in frames[:3]) if frames else "?"] += 1
print("host ops per step (4 steps profiled):")
for n, c in ops.most_common(24):
    print(f"  {n:32s} {c / N:7.1f}")
    for f, k in where[n].most_common(6):
        print(f"        {k / N:6.1f}  {f[:150]}")
dev = collections.Counter()
for ev in prof.events():
    if ev.device_type.name == "CUDA":
        dev[ev.name[:70]] += 1
print("device events per step:")
for n, c in dev.most_common(30):
    print(f"  {c / N:7.1f}  {n}")
