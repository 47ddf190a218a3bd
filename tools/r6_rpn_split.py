"""Round 6 (GPU box): where a forward of --workload ref-crop-rpn goes (synchronised pieces, forward only and training forward)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd.trainstep import SceneStep
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
wl = sys.argv[2] if len(sys.argv) > 2 else "ref-crop-rpn"
job = SceneStep(wl, torch.device("cuda", 0), dtype=dt, prefetch=False, seed=1)
m = job.model
for _ in range(3): job.step()
torch.cuda.synchronize()
def T(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
for grad in (False, True):
    ctx = torch.enable_grad() if grad else torch.no_grad()
    with ctx:
        feats = job.feats.detach().requires_grad_(grad)
        t_bb, out = T(lambda: m.backbone(job.coords, feats, job.size, job.batch_size))
        inter = m.backbone.unet.interims
        t_rpn, (bb, sc, an) = T(lambda: m.run_rpn(inter))
        per = []
        for r, l in zip(m.rpn.levels if hasattr(m.rpn, "levels") else [m.rpn], m.rpn_levels):
            per.append(round(T(lambda: r(inter[l]))[0], 2))
        t_sel, (rs, boxes, ri) = T(lambda: m.roi_selector(bb, sc, an, job._scene_shape()))
        if job.mask_boxes: boxes = [b[:job.mask_boxes] for b in boxes]
        scene = (job.coords, feats, job.size, job.batch_size, job.splits)
        t_mask, _ = T(lambda: m.mask(scene, out, boxes))
    print(f"{wl} {dt} grad={grad}: backbone {t_bb:.2f} ms  rpn {t_rpn:.2f} (levels {per})  selection {t_sel:.2f}  crop+mask {t_mask:.2f}  "
          f"anchors {sc.shape[1]} kept {[len(b) for b in boxes][:4]}...", flush=True)
t, _ = T(lambda: job.forward_only(), 5); print(f"forward_only {t:.2f} ms")
t, _ = T(lambda: job.step(), 5); print(f"step {t:.2f} ms")
# per-iteration times + allocator counters: is the spread a stall (host allocator / hipMalloc) or steady?
import gc
gc.collect(); gc.freeze()
for name, fn in (("forward_only", job.forward_only), ("step", job.step)):
    ts = []
    s0 = torch.cuda.memory_stats()
    for _ in range(24):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    s1 = torch.cuda.memory_stats()
    print(name, "per-iteration ms:", [round(t, 1) for t in ts], " hipMalloc calls", s1["num_device_alloc"] - s0["num_device_alloc"],
          "hipFree", s1["num_device_free"] - s0["num_device_free"], "retries", s1["num_alloc_retries"] - s0["num_alloc_retries"],
          "reserved GB", round(s1["reserved_bytes.all.current"] / 1e9, 2), flush=True)
