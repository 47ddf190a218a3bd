"""Round 5 (GPU box): wall time per block of 25 steps of a bench workload over a long run, with the caching allocator's
counters beside it -- which steps of a run are still warming up, and whether the allocator is what warms.
python tools/r5_step_trend.py <workload> <dtype> [blocks] [--sleep-ms M]"""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd import trainstep as TS
wl, dt = sys.argv[1], sys.argv[2]
blocks = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 24
job = TS.SceneStep(wl, torch.device("cuda", 0), dtype=dt, prefetch=True, seed=1)
for _ in range(5):
    job.step()
job.finish(); torch.cuda.synchronize(); gc.collect(); gc.freeze()
def stats():
    s = torch.cuda.memory_stats()
    return (s.get("num_device_alloc", 0), s.get("num_device_free", 0), s.get("num_alloc_retries", 0),
            s.get("allocation.all.allocated", 0), s.get("segment.all.allocated", 0), s.get("reserved_bytes.all.current", 0) >> 20)
prev = stats()
for b in range(blocks):
    t0 = time.perf_counter()
    for _ in range(25):
        job.step()
    job.finish(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 25 * 1e3
    cur = stats()
    print(f"{wl} {dt} steps {5 + 25 * b:4d}-{29 + 25 * b:4d}: {ms:6.2f} ms/step   device allocs +{cur[0] - prev[0]} frees +{cur[1] - prev[1]} "
          f"retries +{cur[2] - prev[2]}  block allocs/step {(cur[3] - prev[3]) / 25:.0f}  segments +{cur[4] - prev[4]}  reserved {cur[5]} MiB",
          flush=True)
    prev = cur
