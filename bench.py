#!/usr/bin/env python3
"""Headline benchmark: active-voxels/sec, forward + backward of the ScanNet U-Net backbone (BASELINE.json configs[1]):
~150k active voxels per scene, 32->64->128->256 channels, 3^3 submanifold + 2^3/2 conv/deconv, fp32.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per GPU)

A step = InputLayer (voxel hash + first-occurrence rows + mean of duplicate points) + every rulebook build (they are
rebuilt per forward, as in the reference) + U-Net forward + backward to all parameters and the input features + the
all-reduce of the flat gradient buffer over RCCL (N > 1) + SGD update.  The index structures depend on the coordinates
only, so they are pipelined like a data loader's output: those of batch i+1 are built by a helper thread (one
scn_pyramid_build call on the high-priority index stream) while batch i runs; every timed step contains exactly one
complete index build (--no-prefetch builds them inside the forward pass instead).  One scene per GPU (weak scaling); inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

WORKLOADS = {      # name -> (channels, grid, active voxels, BASELINE.json entry)
    "cfg2": ((32, 64, 128, 256), (512, 512, 256), 150_000, "configs[1]"),
    "cfg5": ((32, 64, 128, 256, 512), (1024, 1024, 512), 600_000, "configs[4] shape, fp32 storage"),
}
CHANNELS, GRID, TARGET, _ = WORKLOADS["cfg2"]
PEAK_FP32_MATRIX_TFLOPS = 157.3          # MI355X_MICROARCH.md "Peak FP32 (matrix)"


def _host_threads():
    """Threads for the CPU baseline: the cores this process may use, capped at the 16-core share of a 1-GPU box."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def cpu_baseline(coords, feats, CHANNELS=CHANNELS):
    """The CPU restatement (oracle) of the same step -- SparseConvNet's CPU algorithm (hash -> rulebook; per offset
    gather -> sgemm -> scatter-add) -- timed on this box's host cores.  It is NOT the SparseConvNet binary (unavailable:
    SURVEY.md §8c).  Bounded sample: three full steps (rulebooks + fwd + bwd) of the same 150k-voxel scene, ~10 s."""
    from oracle import scn_oracle as O
    params = {k: v.requires_grad_() for k, v in O.init_unet_params(7, CHANNELS, seed=0).items()}
    c_np, f = coords.cpu().numpy(), feats.cpu()
    torch.set_num_threads(_host_threads())
    reps = 3                                  # ~10 s of CPU work on a 1-GPU box's 16-core share
    t0 = time.perf_counter()
    for _ in range(reps):
        for v in params.values():
            v.grad = None
        scene = O.OracleScene(c_np)           # rulebooks are rebuilt every step, as on the GPU side
        out = O.unet_forward(scene, f, params, CHANNELS)
        out.backward(torch.ones_like(out))
    dt = time.perf_counter() - t0
    return dict(value=reps * scene.n(0) / dt, unit="active-voxels/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{reps} full steps (rulebooks+fwd+bwd) of the same {scene.n(0)}-voxel scene, torch-CPU fp32 "
                       f"oracle port of the SparseConvNet CPU algorithm, {dt:.2f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--target", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2",
                    help="cfg2 = the configuration the metric is quoted on (default); cfg5 = BASELINE configs[4]'s "
                         "shape (600k voxels, 5 levels to 512 channels) in fp32, a size check, not the headline")
    ap.add_argument("--profile-all", action="store_true",
                    help="time every GEMM kernel launch of the sampled steps, not only the dominant kernel")
    ap.add_argument("--bf16-all", action="store_true",
                    help="NOT the headline configuration: as --bf16-blocks, and every other layer after the first 1x1 "
                         "convolution keeps bf16 features too (fp32 arithmetic in the strided / 1x1 GEMMs)")
    ap.add_argument("--bf16-blocks", action="store_true",
                    help="NOT the headline configuration: the residual units keep features, intermediates and gradients "
                         "in bf16 (fp32 accumulation, fp32 parameters); strided / 1x1 layers and everything else fp32")
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false",
                    help="build the index structures inside the forward pass instead of one batch ahead on a helper "
                         "thread (scn_pyramid_build on the index stream)")
    args = ap.parse_args()
    CHANNELS, GRID, target, cfg_name = WORKLOADS[args.workload]
    target = args.target or target

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    # one rank per GPU; SCN_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 code path
    backend = os.environ.get("SCN_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)       # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import sparse_rcnn_amd  # noqa: F401
    from sparse_rcnn_amd import profiling
    from sparse_rcnn_amd.dp import FlatParams, broadcast_params
    from sparse_rcnn_amd.synthetic import make_batch
    from sparse_rcnn_amd.unet import Backbone

    # one balanced scene per rank (cfg 2: seed 1; cfg 4 style: seeds 10+rank)
    seed = 1 if world == 1 else 10 + rank
    coords, feats, size, bs, _ = make_batch(1, GRID, target, dup=1.15, seed=seed)
    coords_d, feats_d = coords.to(dev), feats.to(dev)

    torch.manual_seed(0)
    model = Backbone(7, CHANNELS, bf16_blocks="all" if args.bf16_all else args.bf16_blocks).to(dev)
    args.bf16_blocks = args.bf16_blocks or args.bf16_all
    flat = FlatParams(model, n_buckets=4)      # N > 1: gradient slices are all-reduced while backward still runs
    broadcast_params(flat)
    gen = torch.Generator(device="cpu").manual_seed(100 + rank)
    gy = None
    n_active = 0
    md_next = None

    def step():
        nonlocal gy, n_active, md_next
        flat.zero_grad()
        fin = feats_d.detach().requires_grad_()
        md = md_next.result() if md_next is not None else None
        md_next = None
        # the index structures of the NEXT batch (they depend on its coordinates only, like a data loader's output) are
        # built by a helper thread on the high-priority index stream while this batch's forward and backward run: every
        # timed step still contains one complete index build
        if args.prefetch:
            md_next = model.prefetch_in_thread(coords_d, size, 1)
        out = model(coords_d, fin, size, 1, metadata=md)
        if gy is None or gy.shape != out.features.shape:
            gy = torch.randn(out.features.shape, generator=gen).to(dev)          # upstream grad dY ~ N(0,1)
            n_active = out.features.shape[0]
        out.features.backward(gy)
        flat.all_reduce_mean()
        flat.sgd_step(1e-6)

    # Kernel timing: HIP events around the launches of the dominant kernel (k_conv_ts: 62 launches per step; in the sampled
    # steps scn_conv_tiles runs with SCN_F_SPLIT_SUM so that its slab-sum kernel is launched, and timed, apart) on 3-4
    # steps spread over the timed region, from a pool of events created before it.  Timing events are not free: every
    # launch of every step timed cost 1-3.5 ms/step (host-bound, and each event pair fences the queue), all four GEMM
    # kernels on every 5th step still ~0.5 ms/step.  --profile-all times all GEMM kernels (the "kernels" table).
    every = max(5, (args.steps + 2) // 3)
    timer = profiling.KernelTimer(every=every, names=None if args.profile_all else {"k_conv_ts", "k_conv_tb"})
    for w in range(args.warmup):
        if w == args.warmup - 1:                    # count the launches of one step to size the event pool
            timer.count_only = True
            profiling.TIMER = timer
        step()
    profiling.TIMER = None
    timer.count_only = False
    timer.reserve(2 * max(timer.count, 128) * ((args.steps + every - 1) // every))
    use_timer = not os.environ.get("SCN_BENCH_NO_TIMER")
    # a generation-2 pass of Python's cyclic GC over the (large, static) torch heap costs 60-90 ms when it lands in
    # the timed region; collect now and move the survivors to the permanent generation, as a training loop would
    gc.collect()
    gc.freeze()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if use_timer:
            timer.begin_step()
            profiling.TIMER = timer if timer.active else None
        step()
    if md_next is not None:                  # the index build started in the last timed step ends inside the timed region
        md_next.result()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    profiling.TIMER = None
    sampled = max(1, timer.sampled_steps)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    vox = torch.tensor([float(n_active)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(vox, op=dist.ReduceOp.SUM)
    dt, total_vox = tmax.item(), vox.item()

    if rank == 0:
        ks = timer.summary()
        if not ks:                                   # SCN_BENCH_NO_TIMER (developer switch): wall clock only
            print(json.dumps({"ms_per_step": dt / args.steps * 1e3, "value": total_vox * args.steps / dt}), flush=True)
            if world > 1:
                dist.destroy_process_group()
            return
        dom = max(ks, key=lambda k: ks[k]["ms"])
        d = ks[dom]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        # HBM-side bytes per launch of the dominant kernel: rocprofv3 PMC passes cannot run inside this process; the
        # figure comes from the committed separate passes (tools/collect_traffic.py -> profiles/r1_traffic.json)
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r1_traffic.json")) as f:
                traffic = json.load(f)["kernels"][dom]["hbm_bytes_per_launch"] if args.workload == "cfg2" else None
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "active-voxels/sec fwd+bwd, ScanNet U-Net backbone",
            "value": total_vox * args.steps / dt,
            "unit": "active-voxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16 storage after the first layer, fp32 accumulate" if args.bf16_all else
                      "f32 (residual units: bf16 storage, fp32 accumulate)") if args.bf16_blocks else "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE {cfg_name}: one synthetic ScanNet-shaped scene per GPU, "
                                   f"{n_active} active voxels (grid {GRID[0]}x{GRID[1]}x{GRID[2]}, 1.15 points/voxel), "
                                   "U-Net " + "-".join(map(str, CHANNELS)) + ", 2 pre-act residual blocks/level, 2^3/2 conv+deconv, "
                                   "step = rulebooks + fwd + bwd (+ grad all-reduce + SGD)"
                                   + ("; rulebooks of batch i+1 built on a helper thread during batch i" if args.prefetch else "")
                                   + ("; NOT the fp32 configuration: residual units on the bf16 storage path" if args.bf16_blocks else ""),
                       "parallelism": f"dp{world} (1 scene/GPU, flat-bucket all-reduce)"},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": traffic,
                         "traffic_unit": "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r1_traffic.json)",
                         "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                         "launches_per_step": d["launches"] / sampled, "sampled_steps": sampled,
                         "avg_launch_us": d["ms"] * 1e3 / d["launches"],
                         "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9},
            "kernels": {k: {"ms_per_step": v["ms"] / sampled, "launches_per_step": v["launches"] / sampled,
                            "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else None}
                        for k, v in ks.items()},
        }
        if args.bf16_blocks:         # side measurement: the roofline object of the contract belongs to the fp32 run
            out["roofline"] = {"note": "mixed-storage side measurement; kernel times in `kernels` (use --profile-all), "
                                       "roofline of the fp32 configuration: run without --bf16-blocks"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(coords, feats, CHANNELS)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
