"""Helper of tests/test_gpu_atsize.py::test_syncbn_two_ranks_equals_single_process -- ONE rank (fresh process, gloo, both
ranks on cuda:0): BatchNormLeakyReLU with statistics over the rows of all ranks on this rank's share of a seeded matrix."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def main():
    out_path = sys.argv[1]
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import modules as M
    g = torch.Generator().manual_seed(7)
    X = torch.randn(5000, 24, generator=g) * 2 + 0.5
    G = torch.randn(5000, 24, generator=g)
    lo, hi = (0, 1800) if rank == 0 else (1800, 5000)            # uneven shares
    M._BatchNorm.SYNC = True
    bn = scn.BatchNormLeakyReLU(24, leakiness=0.2).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, 24)); bn.bias.copy_(torch.linspace(-0.3, 0.3, 24))
    x = X[lo:hi].cuda().requires_grad_()
    t = scn.SparseConvNetTensor(features=x, metadata=None, spatial_size=None)
    y = bn(t).features
    y.backward(G[lo:hi].cuda())
    torch.cuda.synchronize()
    np.savez(out_path, y=y.detach().cpu().numpy(), dx=x.grad.cpu().numpy(), dg=bn.weight.grad.cpu().numpy(),
             db=bn.bias.grad.cpu().numpy(), rm=bn.running_mean.cpu().numpy(), rv=bn.running_var.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
