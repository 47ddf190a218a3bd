"""Round 5 probe: host cost of the bucketed gradient all-reduce on ONE GPU (1-rank NCCL process group: the collective moves nothing,
hooks / packs / launches remain).  cProfile over N steps; backward runs on the calling thread, so the hooks are visible."""
import cProfile, pstats, sys, os, time, io, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1", SCN_DP_FORCE_BUCKETS="1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from sparse_rcnn_amd.trainstep import SceneStep
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
job = SceneStep("cfg2", torch.device("cuda", 0), dtype="f32", prefetch=True, seed=1, n_buckets=nb)
print("buckets", len(job.flat.buckets), [len(ids) for ids, _ in job.flat.buckets], [s.numel() for _, s in job.flat.buckets])
for _ in range(15):
    job.step()
torch.cuda.synchronize(); gc.collect(); gc.freeze()
n = 40
t0 = time.perf_counter()
for _ in range(n):
    job.step()
job.finish(); torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / n * 1e3:.3f} ms/step unprofiled")
pr = cProfile.Profile(); pr.enable()
for _ in range(n):
    job.step()
job.finish(); torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats("dp.py"); print(s.getvalue()[:3000])
dist.destroy_process_group()
