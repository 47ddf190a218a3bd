"""Round 5 probe: where the host spends a detection + mask step (cProfile over N steps after warm-up; wall per step with and
without the profiler).  python tools/r5_host_profile.py cfg3 bf16 [steps]"""
import cProfile, pstats, sys, os, time, io, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd.trainstep import SceneStep
wl, dt = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
job = SceneStep(wl, torch.device("cuda", 0), dtype=dt, prefetch=True, seed=1)
for _ in range(15):
    job.step()
torch.cuda.synchronize(); gc.collect(); gc.freeze()
t0 = time.perf_counter()
for _ in range(n):
    job.step()
job.finish(); torch.cuda.synchronize()
print(f"{wl} {dt}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step unprofiled")
# host-only cost: the same loop with the GPU never waited for except where the step itself waits
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    job.step()
job.finish(); torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
print(s.getvalue()[:8000])
