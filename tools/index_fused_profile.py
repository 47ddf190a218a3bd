"""Developer tool (GPU box): N fused index builds of one workload, for `rocprofv3 --kernel-trace --stats`.
    cd /tmp && rocprofv3 --kernel-trace --stats -d out -- python3 /root/repo/tools/index_fused_profile.py [cfg2|crops|roi|cfg5] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd.metadata import Metadata
from sparse_rcnn_amd.synthetic import make_batch

which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n_s, grid, target, levels = {"cfg2": (1, (512, 512, 256), 150_000, 4), "crops": (12, (128, 128, 64), 12_500, 6),
                             "roi": (64, (544, 544, 288), 1_900, 4), "cfg5": (1, (1024, 1024, 512), 600_000, 5)}[which]
coords, feats, size, bs, _ = make_batch(n_s, grid, target, dup=1.15, seed=1)
cd = coords.cuda()
for _ in range(reps):
    Metadata(3).build_native(size, cd, bs, 4, levels, 3)
torch.cuda.synchronize()
