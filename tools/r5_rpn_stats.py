"""Round 5 probe: the synthetic RPN of `--workload cfg3-rpn` as bench.py builds it (zero scn biases): score / box statistics,
proposals kept, cropped points."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd.trainstep import SceneStep
from sparse_rcnn_amd import rpn as R
for dtype in ("f32", "bf16"):
    job = SceneStep("cfg3-rpn", torch.device("cuda", 0), dtype=dtype, prefetch=False, seed=1)
    for it in range(3):
        job.step()
        torch.cuda.synchronize()
        rb, rs, an, sc, bx, ix = job.rpn_out
        lvl = job.model.backbone.unet.interims[-1].features
        dec = R.decode_boxes(an, rb.detach())
        print(dtype, it, "level-3 |x| max", float(lvl.float().abs().max()), "score", float(rs.min()), float(rs.max()), "finite", bool(torch.isfinite(rs).all()),
              "delta max", float(rb.abs().max()), "boxes finite", bool(torch.isfinite(dec).all()), "kept", len(bx[0]),
              "box sizes", (bx[0][:, 1] - bx[0][:, 0]).mean(0).tolist() if len(bx[0]) else None, "roi rows", job.logits.shape[0], flush=True)
