"""Developer tool (GPU box): the fused index build (scn_pyramid2.hip, SCN_PYRAMID_FUSED) against the round-3 builder
(SCN_PYRAMID_V1=1), alone on an idle GPU, alternating: cfg-2 scene with 4 and 6 levels, the reference's batch of 12 crops,
a 600k-voxel scene with 5 levels.      python tools/index_fused_ab.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd._lib import switches as _SW      # library switches: scn_debug_set (the environment is read once at load)
from sparse_rcnn_amd.metadata import Metadata
from sparse_rcnn_amd.synthetic import make_batch

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def timed(build):
    for _ in range(3):
        build()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        build()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name, n_s, grid, target, levels in (("cfg2 150k, 4 levels", 1, (512, 512, 256), 150_000, 4),
                                         ("cfg2 150k, 6 levels", 1, (512, 512, 256), 150_000, 6),
                                         ("12 crops 128x128x64, 6 levels", 12, (128, 128, 64), 12_500, 6),
                                         ("ROI-batch-like 120k pts, 4 levels", 64, (544, 544, 288), 1_900, 4),
                                         ("600k, 5 levels", 1, (1024, 1024, 512), 600_000, 5)):
    coords, feats, size, bs, _ = make_batch(n_s, grid, target, dup=1.15, seed=1)
    cd = coords.cuda()
    build = lambda: Metadata(3).build_native(size, cd, bs, 4, levels, 3)
    res = []
    for rep in range(3):
        fused = timed(build)
        _SW["SCN_PYRAMID_V1"] = "1"
        old = timed(build)
        del _SW["SCN_PYRAMID_V1"]
        res.append((fused, old))
    print(f"{name}: {len(coords)} points; ms per build fused / round-3: " + "  ".join(f"{f:.3f}/{o:.3f}" for f, o in res), flush=True)
