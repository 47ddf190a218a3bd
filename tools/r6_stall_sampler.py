"""Round 6 (GPU box): which Python line the main thread sits on during the slow iterations of a step (a 1 ms stack sampler).
    python tools/r6_stall_sampler.py <workload> <dtype> [forward_only|step]"""
import os, sys, time, threading, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd.trainstep import SceneStep
wl, dt = sys.argv[1], sys.argv[2]
what = sys.argv[3] if len(sys.argv) > 3 else "forward_only"
job = SceneStep(wl, torch.device("cuda", 0), dtype=dt, prefetch=(os.environ.get("PREFETCH", "1") == "1"), seed=1)
fn = getattr(job, what)
for _ in range(6): job.step()
for _ in range(4): fn()
torch.cuda.synchronize()
import gc
if os.environ.get('GC') == '0': gc.disable()
if os.environ.get('GC') == 'freeze': gc.collect(); gc.freeze()
main_id = threading.main_thread().ident
samples, cur, stop = [], [0], [False]
def sampler():
    while not stop[0]:
        fr = sys._current_frames().get(main_id)
        if fr is not None:
            st = traceback.extract_stack(fr)
            samples.append((cur[0], time.perf_counter(), tuple((os.path.basename(f.filename), f.lineno, f.name) for f in st[-6:])))
        time.sleep(0.001)
th = threading.Thread(target=sampler, daemon=True); th.start()
ts = []
for it in range(24):
    cur[0] = it
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
stop[0] = True; th.join()
print("per-iteration ms:", [round(t, 1) for t in ts])
med = sorted(ts)[len(ts) // 2]
slow = {i for i, t in enumerate(ts) if t > 1.8 * med}
print("median", round(med, 1), "slow iterations", sorted(slow))
cnt = collections.Counter()
for it, t, st in samples:
    if it in slow:
        cnt[st[-3:]] += 1
for st, c in cnt.most_common(8):
    print(c, " <- ".join(f"{f}:{l}:{n}" for f, l, n in reversed(st)))
