"""Which Python lines of a step call torch's own small device ops (copies, fills, elementwise adds, cat)?  A TorchDispatchMode
logs every aten op that touches a CUDA tensor with the innermost frame inside this repository; printed per step with the sizes.
(torch.profiler's with_stack gives no Python frames on this build.)"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from torch.utils._pytree import tree_flatten

from sparse_rcnn_amd.trainstep import SceneStep

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("aten.view", "aten.detach", "aten.alias", "aten.slice", "aten.select", "aten.as_strided", "aten.t.", "aten.transpose",
        "aten._unsafe_view", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.permute", "aten.reshape", "aten.split",
        "aten.unbind", "aten.narrow", "aten.empty", "aten.lift_fresh", "aten._local_scalar_dense", "aten.is_pinned",
        "aten.record_stream", "aten.set_", "aten.resize_")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.defaultdict(lambda: [0, 0])

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name.startswith(SKIP):
            return out
        flat = tree_flatten((args, kwargs, out))[0]
        cuda = [t for t in flat if isinstance(t, torch.Tensor) and t.is_cuda]
        if not cuda:
            return out
        frame = "(outside the repository)"
        for f in reversed(traceback.extract_stack()[:-1]):
            if f.filename.startswith(ROOT) and "tools/find_torch_kernels" not in f.filename:
                frame = f"{f.filename[len(ROOT) + 1:]}:{f.lineno} {f.line}"
                break
        r = self.rows[(name, frame)]
        r[0] += 1
        r[1] += max(t.numel() * t.element_size() for t in cuda)
        return out


wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
job = SceneStep(wl, torch.device("cuda:0"), dtype=dtype, prefetch=False)
for _ in range(3):
    job.step()
torch.cuda.synchronize()
N = 2
log = Log()
with log:
    for _ in range(N):
        job.step()
torch.cuda.synchronize()
print(f"{wl} {dtype}: aten ops on device tensors per step (count, largest operand bytes per call), innermost repository frame")
for (name, frame), (c, b) in sorted(log.rows.items(), key=lambda kv: (-kv[1][0], kv[0])):
    print(f"{c / N:6.1f} {b / c / 1e3:10.1f} KB  {name:34s} {frame[:150]}")
