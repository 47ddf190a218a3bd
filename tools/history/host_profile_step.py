"""Developer tool (GPU box): cProfile of the enqueueing thread over steps of trainstep.SceneStep.
    python tools/host_profile_step.py [cfg2|cfg3|cfg5] [f32|bf16]"""
import cProfile, pstats, os, sys, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd.trainstep import SceneStep

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
job = SceneStep(wl, torch.device("cuda", 0), dtype=dt, prefetch="--no-prefetch" not in sys.argv)
for _ in range(5): job.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): job.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{wl} {dt}: host enqueue {1e3*(t1-t0)/10:.2f} ms/step, wall {1e3*(t2-t0)/10:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(10): job.step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats("sparse_rcnn_amd|bench", 40)
print(s.getvalue()[:9000])
