"""Developer tool (GPU box): where the enqueueing thread spends a step of trainstep.SceneStep (wall-clock sections on the
host; the GPU runs behind).  python tools/host_step_split.py [cfg2|cfg3] [f32|bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd.trainstep import SceneStep
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
job = SceneStep(wl, torch.device("cuda", 0), dtype=dt)
for _ in range(10): job.step()
torch.cuda.synchronize()
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + t - t0; return t
N = 30
m = job.model
DEPTH = int(os.environ.get("PREFETCH_DEPTH", "1"))       # index structures built this many batches ahead
from collections import deque
queue = deque()
if job._md_next is not None:
    queue.append(job._md_next); job._md_next = None
while len(queue) < DEPTH:
    queue.append(m.backbone.prefetch_in_thread(job.coords, job.size, job.batch_size))
T0 = time.perf_counter()
for _ in range(N):
    t = time.perf_counter()
    job.flat.zero_grad()
    fin = job.feats.detach().requires_grad_()
    md = queue.popleft().result()
    t = tick("wait for the prefetched index", t)
    queue.append(m.backbone.prefetch_in_thread(job.coords, job.size, job.batch_size))
    t = tick("start the next prefetch", t)
    out = m.backbone(job.coords, fin, job.size, job.batch_size, metadata=md)
    t = tick("backbone forward", t)
    if m.mask is None:
        out.features.backward(job.upstream_grads()[0])
    else:
        scene = (job.coords, fin, job.size, job.batch_size, job.splits)
        logits, selection = m.mask(scene, out, job.boxes)
        t = tick("mask branch forward", t)
        torch.autograd.backward([out.features, logits], [job.upstream_grads()[0], job.upstream_grads()[1]])
    t = tick("backward", t)
    job.flat.step_single_rank(job.lr)
    t = tick("optimizer", t)
T1 = time.perf_counter()
torch.cuda.synchronize()
T2 = time.perf_counter()
for q in queue: q.result()
job.finish()
print(f"{wl} {dt} depth {DEPTH}: host {1e3 * (T1 - T0) / N:.2f} ms/step, wall {1e3 * (T2 - T0) / N:.2f}")
for k, v in acc.items():
    print(f"   {k:32s} {1e3 * v / N:6.2f} ms/step")
