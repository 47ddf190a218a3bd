"""Developer tool (GPU box): where the host time of a step goes.  A tiny scene (2 000 voxels) makes every kernel short, so
the step's wall time IS the enqueue cost of the same launch sequence the 150k-voxel step issues.
    python tools/host_cost.py [cfg2|cfg3] [f32|bf16]"""
import cProfile, pstats, os, sys, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.trainstep import SceneStep

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda", 0)
lib = L.lib()


def per_call(fn, n=20000):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return 1e6 * t


x = torch.zeros(1024, device=dev)
print(f"ctypes call, no argument, no launch (scn_abi_version)      {per_call(lib.scn_abi_version):6.2f} us")
s = L.stream()
print(f"L.stream()                                                  {per_call(L.stream):6.2f} us")
print(f"L.ptr(x)                                                    {per_call(lambda: L.ptr(x)):6.2f} us")
print(f"torch.empty((1000, 32), device)                             {per_call(lambda: torch.empty((1000, 32), device=dev)):6.2f} us")
print(f"torch x.add_(1) (one elementwise launch)                    {per_call(lambda: x.add_(1), 5000):6.2f} us")
print(f"torch x.new_empty + copy_                                   {per_call(lambda: x.new_empty(x.shape).copy_(x), 5000):6.2f} us")

job = SceneStep(wl, dev, dtype=dt, prefetch=False, target=2000, grid=(64, 64, 32))
for _ in range(5):
    job.step()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    job.step()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"{wl} {dt} tiny scene ({job.n_active} voxels), no prefetch: {1e3 * (t1 - t0) / n:.2f} ms/step = host cost of a step")
from sparse_rcnn_amd import profiling
m = job.model
fin = job.feats
with torch.no_grad():
    for _ in range(3):
        m.backbone(job.coords, fin, job.size, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m.backbone(job.coords, fin, job.size, 1)
    torch.cuda.synchronize()
    print(f"backbone forward only (no_grad, incl. index build): {1e3 * (time.perf_counter() - t0) / n:.2f} ms")
    md = m.backbone(job.coords, fin, job.size, 1).metadata
    t0 = time.perf_counter()
    for _ in range(n):
        m.backbone(job.coords, fin, job.size, 1, metadata=md)
    torch.cuda.synchronize()
    print(f"backbone forward only (no_grad, index structures given): {1e3 * (time.perf_counter() - t0) / n:.2f} ms")
# backward on the calling thread, so that the profile below sees the Python side of the backward functions too
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    job.forward_backward()
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(70)
print(st.getvalue()[:16000])
