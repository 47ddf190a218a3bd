"""Round 6 (GPU box): cProfile of N iterations of a step (host time by function, C calls included)."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd.trainstep import SceneStep
wl, dt = sys.argv[1], sys.argv[2]
what = sys.argv[3] if len(sys.argv) > 3 else "forward_only"
job = SceneStep(wl, torch.device("cuda", 0), dtype=dt, prefetch=False, seed=1)
fn = getattr(job, what)
for _ in range(6): job.step()
for _ in range(4): fn()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(24):
    fn(); torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:6000])
