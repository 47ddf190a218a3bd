"""Developer tool (GPU): the index build of the cfg-2 scene alone, N times -- run under rocprofv3 --kernel-trace --stats to
see what its ~60 launches cost when nothing else shares the chip."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from sparse_rcnn_amd.metadata import Metadata
from sparse_rcnn_amd.synthetic import make_batch
n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 30
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, dup=1.15, seed=1)
coords = coords.cuda()
for i in range(n_rep + 3):
    if i == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    md = Metadata(3).build_native(size, coords, 1, 4, 4, 3)
torch.cuda.synchronize()
print(f"index build: {(time.perf_counter() - t0) / n_rep * 1e3:.3f} ms per build")
