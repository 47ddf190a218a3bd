"""Round 5 A/B inside ONE process: blocks of steps alternate between settings of trainstep's module switches, so box and
time drift cancel (separate bench.py runs of a host-bound step scatter by +-0.8 ms on one box).
python tools/r5_ab_inproc.py <workload> <dtype> name=FLAG:val,FLAG:val name2=... [--blocks 6] [--steps 25]"""
import sys, os, time, gc, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd import trainstep as TS
wl, dt = sys.argv[1], sys.argv[2]
modes, blocks, steps = [], 6, 25
args = sys.argv[3:]
while args:
    a = args.pop(0)
    if a == "--blocks": blocks = int(args.pop(0))
    elif a == "--steps": steps = int(args.pop(0))
    else:
        name, spec = a.split("=")
        modes.append((name, [(kv.split(":")[0], (int(kv.split(":")[1]) if kv.split(":")[1] not in "01" else kv.split(":")[1] == "1"))
                             for kv in spec.split(",") if kv]))
# measurement only (profiles/r5_ab_index_interference.txt): REUSE_INDEX:1 = every step re-uses the index structures of the first
# one (the bound of a perfectly hidden index build; one scene repeated: the structures are the same anyway); :2 = the helper thread
# still builds -- a TINY scene (host side of a build only); :3-7 = what of a helper-thread build costs the step when the build
# itself is taken away (3 an empty job on the helper thread, 4 + the workspace allocation on the index stream, 5 + one small
# kernel and a device->host wait there, 6 / 7 = 17 / 68 one-workgroup kernels on the index stream).  Lives HERE, as a subclass
# of the product step: SceneStep.forward_backward contains only the step.
REUSE_INDEX = False
_reused_md = {}


class ReuseIndexStep(TS.SceneStep):
    def _take_index(self, k):
        md = super()._take_index(k)
        if not REUSE_INDEX:
            return md
        from sparse_rcnn_amd.metadata import Metadata, PendingMetadata, index_stream
        m = self.model
        if REUSE_INDEX in (3, 4, 5, 6, 7):
            pend = _reused_md.pop("tiny_pending", None)
            if pend is not None:
                pend.result()
            dev, mode = self.device, REUSE_INDEX

            def job():
                torch.cuda.set_device(dev)
                if mode >= 4:
                    side = index_stream(dev)
                    with torch.cuda.stream(side):
                        ws = torch.empty(40 << 20, dtype=torch.uint8, device=dev)
                        if mode == 5:
                            ws[:1024].zero_()
                            int(ws[:8].sum().item())
                        if mode >= 6:
                            for _ in range(17 if mode == 6 else 68):
                                ws[:256].zero_()
                            side.synchronize()
                return None
            _reused_md["tiny_pending"] = PendingMetadata(job)
        if REUSE_INDEX == 2:
            tiny = _reused_md.get("tiny")
            if tiny is None:
                from sparse_rcnn_amd.synthetic import make_batch
                c, _, sz, bs_, _ = make_batch(1, (64, 64, 32), 2000, dup=1.15, seed=5)
                tiny = _reused_md["tiny"] = (c.to(self.device), sz, bs_)
            pend = _reused_md.pop("tiny_pending", None)
            if pend is not None:
                pend.result()
            _reused_md["tiny_pending"] = m.backbone.prefetch_in_thread(*tiny)
        base = _reused_md.get((id(self), k))
        if base is None and md is not None:
            base = _reused_md[(id(self), k)] = md
        if base is not None:
            md = Metadata(3)
            md.__dict__.update(base.__dict__)
            md.ready_event = None
        return md

    def _start_prefetch(self, k):
        if REUSE_INDEX and (id(self), k) in _reused_md:
            return
        super()._start_prefetch(k)


job = ReuseIndexStep(wl, torch.device("cuda", 0), dtype=dt, prefetch=True, seed=1)
def apply(flags):
    from sparse_rcnn_amd._lib import switches
    for k, v in flags:
        if k == "REUSE_INDEX":
            global REUSE_INDEX
            REUSE_INDEX = v
        elif k.startswith("SCN_"):               # a library switch (scn_debug_set): SCN_PYRAMID_V1:1 / SCN_TS_PROG:8 / :0 (0 = unset)
            if v:
                switches[k] = "1" if v is True else str(v)
            else:
                del switches[k]
        elif "." in k:                           # a switch of another module of the package: proposals.TORCH_TOPK:1
            import importlib
            mod, attr = k.rsplit(".", 1)
            setattr(importlib.import_module("sparse_rcnn_amd." + mod), attr, v)
        else:
            setattr(TS, k, v)
for _ in range(20):
    job.step()
torch.cuda.synchronize(); gc.collect(); gc.freeze()
res = {n: [] for n, _ in modes}
for b in range(blocks):
    order = modes if b % 2 == 0 else modes[::-1]
    for name, flags in order:
        apply(flags)
        for _ in range(3):
            job.step()
        job.finish(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            job.step()
        job.finish(); torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / steps * 1e3)
for name, v in res.items():
    print(f"{wl} {dt} {name:40s} mean {statistics.mean(v):.3f} ms  min {min(v):.3f}  max {max(v):.3f}   blocks {[round(x, 2) for x in v]}", flush=True)
