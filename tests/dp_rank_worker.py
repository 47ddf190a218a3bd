"""Helper of tests/test_gpu_atsize.py::test_two_rank_dp_step_matches_oracle -- ONE rank of a data-parallel step of the
real Backbone, run as a fresh process (RANK / WORLD_SIZE / MASTER_* in the environment, gloo collective so that two
ranks may share one GPU).  Writes the parameters and the all-reduced mean gradient as seen by this rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def main():
    out_path, target, grid = sys.argv[1], int(sys.argv[2]), tuple(int(v) for v in sys.argv[3].split(","))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from sparse_rcnn_amd.trainstep import SceneStep
    job = SceneStep("cfg2", torch.device("cuda", 0), dtype="f32", prefetch=False, seed=10 + rank, grad_seed=100 + rank,
                    n_buckets=4, target=target, grid=grid, lr=0.0)
    assert job.flat.buckets, "the bucketed, overlapped all-reduce must be active with 2 ranks"
    job.step()
    torch.cuda.synchronize()
    names = list(job.model.backbone.unet.named_oracle_params())
    params = {k: p.detach().cpu().numpy() for k, p in job.model.backbone.unet.named_oracle_params().items()}
    grads = {}
    views = dict(zip([id(p) for p in job.flat.params], job.flat.grad_views))
    for k, p in job.model.backbone.unet.named_oracle_params().items():
        grads["g:" + k] = views[id(p)].detach().cpu().numpy()
    np.savez(out_path, n_active=job.n_active, **params, **grads)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
