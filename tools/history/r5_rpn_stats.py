"""Round 5 probe: the synthetic RPN of `--workload cfg3-rpn` as bench.py builds it: score / box statistics, proposals kept,
cropped points, per step, with and without the index prefetch thread."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd.trainstep import SceneStep
from sparse_rcnn_amd import rpn as R
for dtype, prefetch in (("f32", True), ("f32", False), ("bf16", True)):
    job = SceneStep("cfg3-rpn", torch.device("cuda", 0), dtype=dtype, prefetch=prefetch, seed=1)
    for it in range(8):
        job.step()
        torch.cuda.synchronize()
        rb, rs, an, sc, bx, ix = job.rpn_out
        dec = R.decode_boxes(an, rb.detach())
        print(dtype, "prefetch", prefetch, it, "score", round(float(rs.min()), 3), round(float(rs.max()), 3), "nonconst", int((rs.detach() != rs.detach().flatten()[0]).sum()),
              "delta max", round(float(rb.detach().abs().max()), 4), "kept", len(bx[0]), "top score", float(sc[0][0]) if len(sc[0]) else None,
              "box0", bx[0][0].flatten().tolist() if len(bx[0]) else None, "roi rows", job.logits.shape[0], flush=True)
    job.finish()
