"""Helper of tests/test_gpu_atsize.py::test_two_rank_dp_step_matches_oracle -- ONE rank of a data-parallel step of the
real Backbone (cfg2) or of Backbone + sparse ROI crop + mask branch (cfg3), run as a fresh process (RANK / WORLD_SIZE /
MASTER_* in the environment, gloo collective so that two ranks may share one GPU).  Writes the parameters and the
all-reduced mean gradient as seen by this rank.

argv: out.npz target[/target_rank1] grid_x,grid_y,grid_z [workload [dtype [n_boxes [empty_rank [weighting [batches_per_step]]]]]]
empty_rank: that rank's boxes are moved outside the scene (its ROI crop is empty: no mask-branch gradients there)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def main():
    out_path, grid = sys.argv[1], tuple(int(v) for v in sys.argv[3].split(","))
    targets = [int(t) for t in sys.argv[2].split("/")]             # one target per rank ("12000/8000") or one for all
    workload = sys.argv[4] if len(sys.argv) > 4 else "cfg2"
    dtype = sys.argv[5] if len(sys.argv) > 5 else "f32"
    n_boxes = int(sys.argv[6]) if len(sys.argv) > 6 else None
    empty_rank = int(sys.argv[7]) if len(sys.argv) > 7 else -1
    weighting = sys.argv[8] if len(sys.argv) > 8 else "equal"
    bps = int(sys.argv[9]) if len(sys.argv) > 9 else 1            # micro-batches accumulated before the one all-reduce
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    target = targets[rank % len(targets)]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from sparse_rcnn_amd.trainstep import SceneStep
    job = SceneStep(workload, torch.device("cuda", 0), dtype=dtype, prefetch=False, seed=10 + rank, grad_seed=100 + rank,
                    n_buckets=4, target=target, grid=grid, lr=0.0, n_boxes=n_boxes, weighting=weighting,
                    batches_per_step=bps)
    assert job.flat.buckets, "the bucketed, overlapped all-reduce must be active with 2 ranks"
    if rank == empty_rank:
        job.boxes = [b + 10_000.0 for b in job.boxes]
    from sparse_rcnn_amd import functional as F
    F.RELU_RECORD = []                      # the sign masks of this rank's forward, for the oracle's FrozenReLU
    job.step()
    torch.cuda.synchronize()
    masks, F.RELU_RECORD = F.RELU_RECORD, None
    named = {k: p for k, p in job.model.backbone.unet.named_oracle_params().items()}
    if job.model.mask is not None:
        named.update({"m:" + k: p for k, p in job.model.mask.named_oracle_params().items()})
    params = {k: p.detach().cpu().numpy() for k, p in named.items()}
    views = dict(zip([id(p) for p in job.flat.params], job.flat.mean_grad_views()))
    assert len(views) == len(named), (len(views), len(named))       # every parameter of the step is named
    grads = {"g:" + k: views[id(p)].detach().cpu().numpy() for k, p in named.items()}
    mk = {f"mask{i}": np.packbits(m.numpy().reshape(-1)) for i, m in enumerate(masks)}
    mk["mask_shapes"] = np.array([tuple(m.shape) for m in masks], np.int64).reshape(-1, 2)
    per = np.array([job.upstream_grads(k)[0].shape[0] for k in range(bps)], np.int64)      # active voxels per micro-batch
    np.savez(out_path, n_active=job.n_active, n_active_per=per, n_roi_rows=job.n_roi_rows, **params, **grads, **mk)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
