#!/usr/bin/env python3
"""Headline benchmark: active-voxels/sec, forward + backward of the ScanNet U-Net backbone (BASELINE.json configs[1]):
~150k active voxels per scene, 32->64->128->256 channels, 3^3 submanifold + 2^3/2 conv/deconv, fp32.

  python bench.py --gpus N --steps K --warmup W [--workload cfg2|cfg3|cfg5] [--dtype f32|bf16]

N > 1, either way:
  * `python bench.py --gpus N ...` with no torchrun environment: this process makes NO GPU call and starts N fresh child
    processes of itself (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, 127.0.0.1 rendezvous), relays
    rank 0's JSON line and exits with the worst child's code;
  * `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`: the ranks are already there.

A step (sparse_rcnn_amd/trainstep.py) = InputLayer (voxel hash + first-occurrence rows + mean of duplicate points) + every
rulebook build (rebuilt per batch, as in the reference) + forward + backward to all parameters and the input features +
all-reduce of the flat gradient buffer over RCCL (N > 1) + SGD update.  The index structures depend on the coordinates
only, so they are pipelined like a data loader's output: those of batch i+1 are built by a helper thread while batch i
runs; every timed step contains exactly one complete index build (`ms_per_step_no_prefetch` in the JSON is the same loop
with the build inside the forward pass).  One scene per GPU: N ranks process N scenes per step (the batch of N scenes of
BASELINE configs[3] sharded one per GPU); per-GPU work is fixed as N grows, so the line says "weak" -- a batch of 8 on
8 GPUs against the same batch on 1 GPU is the same curve read as strong scaling (`scaling_note`).  Inputs are resident
in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3          # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBPS = 8000.0                   # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)       # >= 10 warm-up, >= 50 timed iterations
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--target", type=int, default=None, help="active voxels per scene (default: the workload's)")
    ap.add_argument("--workload", choices=("cfg2", "cfg2-bn", "cfg3", "cfg3-rpn", "cfg5", "ref", "ref-crop", "ref-crop-rpn"), default="cfg2",
                    help="cfg2 = the configuration the metric is quoted on (default); cfg2-bn = cfg2 with BatchNormReLU in the "
                         "residual units (the reference ships batch norm off); cfg3 = backbone + OutputLayer + "
                         "sparse ROI crop (64 synthetic boxes) + mask-branch U-Net, fwd+bwd (crop + mask branch only); cfg3-rpn = "
                         "configs[2] with the RPN boundary inside the step: the boxes come out of the same forward (SparseToDense "
                         "-> a STAND-IN dense stack, 2 x 32 on one anchor level, on this library's tile kernels -> inside anchors "
                         "-> top-k + NMS, 64 kept); cfg5 = 600k voxels, 5 levels to 512; "
                         "ref = the reference's own 6-level plan 32-48-64-80-96-112 on the 150k scene; ref-crop = the same "
                         "plan on the reference's training batch (12 crops of 128x128x64); ref-crop-rpn = that batch with the "
                         "reference's RPN shape (two anchor levels, 5 x 128 / 5 x 256 dilation stacks, top-1024 / NMS 0.5 / 256 "
                         "kept, scannet_config/run.py:525-536,609,847-853) + crop + mask branch on the 24 best per sample")
    ap.add_argument("--dtype", choices=("f32", "bf16", "bf16-blocks"), default="f32",
                    help="feature STORAGE type: f32 (the headline, the reference's arithmetic) or bf16 (BASELINE configs "
                         "3-5: bf16-stored features, fp32 accumulation, fp32 parameters)")
    ap.add_argument("--bf16-all", action="store_true", help="alias of --dtype bf16")
    ap.add_argument("--bf16-blocks", action="store_true", help="alias of --dtype bf16-blocks (residual units only)")
    ap.add_argument("--profile-all", action="store_true",
                    help="time every GEMM kernel launch of the sampled steps, not only the dominant kernel")
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false",
                    help="build the index structures inside the forward pass instead of one batch ahead")
    ap.add_argument("--dropin", dest="dropin", action="store_true", default=True,
                    help="(default) also time the layer-by-layer module path the reference's module_factory builds (lazy "
                         "Metadata, no helper thread) and report it as `dropin` next to the headline")
    ap.add_argument("--no-dropin", dest="dropin", action="store_false")
    ap.add_argument("--no-bf16-leg", dest="bf16_leg", action="store_false",
                    help="skip the short bf16-storage side leg (cfg2 in bf16: ms/step + k_conv_tb roofline) of the default run")
    ap.add_argument("--no-extras", action="store_true", help="skip the index-build / no-prefetch side measurements")
    ap.add_argument("--buckets", type=int, default=4,
                    help="slices of the flat gradient buffer all-reduced from the gradient hooks while backward runs (N > 1); "
                         "0: one all-reduce after backward")
    ap.add_argument("--batches-per-step", type=int, default=1,
                    help="micro-batches (scenes) accumulated per optimizer step and rank -- the reference's batch scaling, "
                         "training.py:436,458-460 (2 or 6 with the mask head); ONE gradient all-reduce per step")
    ap.add_argument("--forward-only-child", action="store_true",
                    help="(internal) evaluation only, in this fresh process: forward_only steps of the workload, nothing else ever "
                         "allocated -- prints one JSON line with its time and peak HBM (the parent's `forward_only.fresh_process`)")
    a = ap.parse_args(argv)
    if a.bf16_all:
        a.dtype = "bf16"
    elif a.bf16_blocks:
        a.dtype = "bf16-blocks"
    return a


# ----------------------------------------------------------------------------------------------------------------------
# self-launch: the parent makes no GPU call
# ----------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tail(path, nbytes=4000):
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            size = f.tell()
            f.seek(max(0, size - nbytes))
            return f.read().decode(errors="replace")
    except OSError:
        return ""


def self_launch(args, script=None, argv=None, poll_s=0.05, grace_s=5.0):
    """Start --gpus fresh ranks of this script.  Nothing here touches torch.cuda: a forked/exec'd child of a process that
    initialised the GPU is not allowed on this pool, and RCCL wants one fresh process per device.

    All ranks are polled: when ANY rank exits non-zero (out of memory, RCCL initialisation, a crash) the others -- which
    would sit in a collective until the driver's time limit -- are terminated, the dead rank's stderr tail is printed, and
    the launch returns that rank's code.  Every rank writes stdout / stderr to its own file (no pipe can fill up); rank 0's
    stdout (the ONE JSON line) is relayed on success, rank 0's stderr always."""
    import signal
    import tempfile
    port = _free_port()
    tmp = tempfile.mkdtemp(prefix="scn_bench_")
    procs, files = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SCN_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out_p, err_p = os.path.join(tmp, f"rank{r}.out"), os.path.join(tmp, f"rank{r}.err")
        fo, fe = open(out_p, "wb"), open(err_p, "wb")
        files.append((out_p, err_p, fo, fe))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] +
                                      (sys.argv[1:] if argv is None else list(argv)), env=env, stdout=fo, stderr=fe))
    dead = None
    while dead is None:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            dead = bad[0]
        elif all(c == 0 for c in codes):
            break
        else:
            time.sleep(poll_s)
    rc = 0
    if dead is not None:
        for p in procs:                                  # the survivors wait in a collective that will never complete
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        t_end = time.time() + grace_s
        for p in procs:
            try:
                p.wait(timeout=max(0.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        rc = abs(dead[1]) or 1
    for _, _, fo, fe in files:
        fo.close(); fe.close()
    sys.stderr.write(_tail(files[0][1], 20000))
    if dead is not None:
        sys.stderr.write(f"\n[bench.py] rank {dead[0]} exited with code {dead[1]}; the other ranks were terminated.  "
                         f"Its stderr tail:\n{_tail(files[dead[0]][1])}\n")
    else:
        with open(files[0][0], "rb") as f:
            sys.stdout.write(f.read().decode(errors="replace"))
    sys.stdout.flush(); sys.stderr.flush()
    for out_p, err_p, _, _ in files:
        for q in (out_p, err_p):
            try:
                os.remove(q)
            except OSError:
                pass
    try:
        os.rmdir(tmp)
    except OSError:
        pass
    return rc


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline leg (test infrastructure: oracle/)
# ----------------------------------------------------------------------------------------------------------------------
def _host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))          # the 16-core share of a 1-GPU box


def cpu_baseline(coords, feats, channels):
    """SparseConvNet's CPU algorithm restated in C++17 / OpenMP (oracle/scn_cpu_baseline.cpp: per-sample hash -> rulebooks;
    per offset gather -> sgemm -> scatter-add; same layer list, fwd + bwd), compiled -O3 -march=native on THIS box, timed
    single-threaded and on the host cores this process may use.  It is NOT the SparseConvNet binary (unavailable: SURVEY.md
    §8c).  Bounded sample: whole steps of the same scene, ~10-20 s of CPU work in total."""
    from oracle import cpu_baseline as CB
    return CB.timed_baseline(coords.cpu().numpy(), feats.cpu().numpy(), channels, _host_threads())


# ----------------------------------------------------------------------------------------------------------------------
def run(args):
    import torch
    import torch.distributed as dist

    # the contract is ONE JSON line on stdout: collective libraries print banners there ("[Gloo] Rank 0 is connected ..."),
    # so everything written to fd 1 before the result goes to stderr instead
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one rank per GPU; SCN_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 code path
    backend = os.environ.get("SCN_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()                         # does not initialise the GPU
    if n_dev == 0:
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    local = local % n_dev if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ranks_seen = 1
    # SCN_BENCH_FORCE_DIST=1: create the process group even for one rank (a 1-GPU box rehearses the RCCL code path:
    # communicator setup, bucketed all-reduce from the gradient hooks, barriers)
    force_dist = bool(os.environ.get("SCN_BENCH_FORCE_DIST")) and world == 1
    if force_dist:
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    dist_on = world > 1 or force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)       # RCCL over xGMI
        else:
            dist.init_process_group(backend)
        one = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    import sparse_rcnn_amd  # noqa: F401
    from sparse_rcnn_amd import profiling
    from sparse_rcnn_amd.trainstep import SceneStep
    # torch's intra-op CPU pool follows os.cpu_count() (256 on a GPU box whose job owns 16 cores): a CPU operator that enters it
    # stalls the step by tens of milliseconds every few calls (round 6: profiles/r6_ref_crop_rpn.txt) -- size it to the share
    torch.set_num_threads(_host_threads())

    # one balanced scene per rank (cfg 2: seed 1; cfg 4 style: seeds 10+rank)
    seed = 1 if world == 1 else 10 + rank
    job = SceneStep(args.workload, dev, dtype=args.dtype, prefetch=args.prefetch, seed=seed, grad_seed=100 + rank,
                    target=args.target, batches_per_step=args.batches_per_step, n_buckets=args.buckets)

    # Kernel timing: HIP events on the launch stream around the launches of the dominant kernel on 3-4 steps spread over the
    # timed region.  Round 4: those steps stay on the production path -- the step executor's C calls bracket their
    # tile-convolution launches themselves (scn_exec_timing_enable; rounds 1-3 sent the sampled steps through the
    # layer-by-layer path, whose host cost the bf16 steps felt).  Timing every launch of every step makes the step host-bound,
    # so the other kernels are only timed with --profile-all (layer-by-layer path).
    dom_names = {"k_conv_ts", "k_conv_tb"}
    every = max(5, (args.steps + 2) // 3)
    timer = profiling.KernelTimer(every=every, names=None if args.profile_all else dom_names)
    for w in range(args.warmup):
        if w == args.warmup - 1:                    # count the launches of one step to size the event pool
            timer.count_only = True
            profiling.TIMER = timer
        job.step()
    profiling.TIMER = None
    timer.count_only = False
    # test hook (tests/test_gpu_atsize.py: the self-launch must notice a rank that dies in a running job): this rank leaves
    # after its warm-up steps, while its peers head for the barrier / the first all-reduce of the timed region
    if os.environ.get("SCN_BENCH_DIE_RANK") == str(rank) and world > 1:
        sys.stderr.write(f"rank {rank}: SCN_BENCH_DIE_RANK -- leaving with code 3 after warm-up\n")
        sys.stderr.flush()
        os._exit(3)
    timer.reserve(2 * max(timer.count, 128) * ((args.steps + every - 1) // every))
    use_timer = not os.environ.get("SCN_BENCH_NO_TIMER")
    # a generation-2 pass of Python's cyclic GC over the (large, static) torch heap costs 60-90 ms when it lands in
    # the timed region; collect now and move the survivors to the permanent generation, as a training loop would
    gc.collect()
    gc.freeze()
    import ctypes
    from sparse_rcnn_amd import _lib as L
    paths = (ctypes.c_int64 * 4)()
    L.lib().scn_conv_tiles_path_counts(paths, 1)              # count the kernel variants of the timed region only
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()                      # peak HBM of the timed region (memory audit)
    base_alloc = torch.cuda.memory_allocated()                # resident before the step: parameters, inputs, the prefetched index
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if use_timer:
            timer.begin_step()
            profiling.TIMER = timer if timer.active else None
        job.step()
    job.finish()                             # the index build started in the last timed step ends inside the timed region
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    peak_alloc, peak_reserved = torch.cuda.max_memory_allocated(), torch.cuda.max_memory_reserved()
    profiling.TIMER = None
    timer.collect_exec()            # the launch times the step executor recorded on the sampled steps (before any other leg runs)
    sampled = max(1, timer.sampled_steps)
    L.lib().scn_conv_tiles_path_counts(paths, 0)
    dt_local = dt

    red_dev = dev if (world == 1 or backend == "nccl") else "cpu"
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    vox = torch.tensor([float(job.n_active)], dtype=torch.float64, device=red_dev)
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(vox, op=dist.ReduceOp.SUM)
    dt, total_vox = tmax.item(), vox.item()
    # per-rank figures (a bad scaling curve must be diagnosable from the one line): step time as each rank saw it, the time
    # its compute stream waited for gradient all-reduces after backward had ended (HIP events, mean of the last steps), rows
    exposed = job.flat.exposed_allreduce_ms()
    mine = torch.tensor([dt_local / args.steps * 1e3, float(job.n_active), -1.0 if exposed is None else exposed,
                         float(job.n_roi_rows)], dtype=torch.float64, device=red_dev)
    per_rank = [mine]
    if dist_on:
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
    per_rank = [dict(rank=r, ms_per_step=v[0], n_active=int(v[1]), allreduce_ms_exposed=None if v[2] < 0 else v[2],
                     n_roi_rows=int(v[3])) for r, v in enumerate(t.tolist() for t in per_rank)]

    # ---- side measurements outside the timed region (rank 0's numbers; every rank runs them so collectives stay matched)
    extras = {}
    if not args.no_extras:
        extras = side_measurements(job, args, world, dist, torch)

    if rank == 0:
        ks = timer.summary()
        out = {
            "metric": "active-voxels/sec fwd+bwd, ScanNet U-Net backbone",
            "value": total_vox * args.steps / dt,
            "unit": "active-voxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "scaling_note": "1 scene per GPU: N ranks process a batch of N scenes per step; against the same batch on one "
                            "GPU (N x ms_per_step at n_gpus=1) this is the strong-scaling curve of BASELINE configs[3]",
            "dtype": {"f32": "f32", "bf16": "bf16", "bf16-blocks": "f32 (residual units: bf16 storage)"}[args.dtype],
            "dtype_note": None if args.dtype == "f32" else "bf16 STORAGE of features and feature gradients after the "
                          "first layer, fp32 accumulation, fp32 parameters / parameter gradients / optimizer",
            "data": "synthetic",
            "config": {"workload": job.describe(),
                       "parallelism": f"dp{world} (1 scene/GPU, flat-bucket all-reduce)",
                       "batches_per_step": job.batches_per_step,
                       "optimizer": "plain SGD on the flat fp32 parameter buffer (dp.FlatParams); the reference trains with Adam",
                       "lr": job.lr},
            "n_ranks_seen_by_rccl": ranks_seen if backend == "nccl" else None,
            "n_ranks_seen": ranks_seen, "collective_backend": backend if dist_on else None,
            "per_rank": per_rank,
            "peak_hbm_bytes": int(peak_alloc),
            "peak_hbm": {"allocated_bytes": int(peak_alloc), "reserved_bytes": int(peak_reserved),
                         "resident_before_step_bytes": int(base_alloc), "step_working_set_bytes": int(peak_alloc - base_alloc),
                         "note": "torch.cuda.max_memory_allocated / max_memory_reserved over the timed region of rank 0: feature "
                                 "slabs, index structures of two batches (this one + the prefetched one), workspaces, "
                                 "parameters, gradients, weight images; of 288 GB"},
            "inputs": "coordinates (int64 [N,4]) and features resident in HBM before the timed region (the contract's "
                      "`value`); the reference hands coordinates over on the HOST (ndsis/data/data.py:207-210): the `dropin` "
                      "leg times that contract (host coords, H2D + range check inside the step)",
            "fast_path": {"conv_tiles_fast": int(paths[0]), "conv_tiles_general": int(paths[1]),
                          "k_reduction_in_launch": int(paths[2]), "k_reduction_second_launch": int(paths[3]),
                          "all_fast": int(paths[1]) == 0 and int(paths[3]) == 0,
                          "note": "scn_conv_tiles launches of the timed region by kernel variant (rank 0): `general` = the "
                                  "64-bit-addressing fallback above 2^23 rows / 4 GB slabs"} if args.dtype == "f32" else None,
        }
        out.update(extras)
        if ks:
            dom = max((k for k in ks if k in dom_names), key=lambda k: ks[k]["ms"], default=max(ks, key=lambda k: ks[k]["ms"]))
            d = ks[dom]
            out["roofline"] = roofline(dom, d, sampled, args)
            shapes = [r for r in timer.by_shape(sampled) if r["bf16"] == (dom == "k_conv_tb")]
            if shapes:                      # the dominant kernel's launches by layer shape (the level that sets the average)
                out["roofline"]["by_shape"] = shapes
            out["kernels"] = {k: {"ms_per_step": v["ms"] / sampled, "launches_per_step": v["launches"] / sampled,
                                  "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else None,
                                  "algorithmic_GBps": v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] else None}
                              for k, v in ks.items()}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(job.coords_cpu, job.feats_cpu, job.channels)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def roofline(dom, d, sampled, args):
    """`roofline` object of the contract for the dominant kernel (HIP events around its launches, live)."""
    us = d["ms"] * 1e3 / d["launches"]
    traffic = None
    # separate rocprofv3 --pmc passes (tools/collect_traffic.py, tools/collect_r6.sh); since round 5: one file, an entry per
    # (workload, storage type) -- the bf16 lines have counter bytes too
    tfile = next((t for t in (os.path.join(ROOT, "profiles", f"r{r}_traffic.json") for r in (6, 5, 4)) if os.path.exists(t)),
                 os.path.join(ROOT, "profiles", "r6_traffic.json"))
    try:
        with open(tfile) as f:
            t = json.load(f)
        for ent in t.get("entries", [t]):
            if ent.get("workload") == args.workload and ent.get("dtype", "f32") == args.dtype:
                traffic = ent["kernels"][dom]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    common = {"kernel": dom, "traffic": traffic,
              "traffic_unit": "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
              "traffic_source": "RECORDED by separate rocprofv3 --pmc passes of an earlier run of this command "
                                "(profiles/%s), not measured by this run" % os.path.basename(tfile),
              "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
              "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
              "launches_per_step": d["launches"] / sampled, "sampled_steps": sampled, "avg_launch_us": us}
    if dom == "k_conv_tb":        # bf16 storage: the bf16 MFMA peak is 16x the fp32 one; the tile kernel is HBM-bound
        achieved = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        return dict(bound="hbm", achieved=achieved, peak=PEAK_HBM_GBPS, unit="GB/s", frac=achieved / PEAK_HBM_GBPS,
                    **common)
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    return dict(bound="mfma", achieved=achieved, peak=PEAK_FP32_MATRIX_TFLOPS, unit="TFLOP/s",
                frac=achieved / PEAK_FP32_MATRIX_TFLOPS, **common)


def side_measurements(job, args, world, dist, torch):
    """What SURVEY §8d asks to report next to the headline: the index build alone (ms, algorithmic bytes -> GB/s against
    the HBM roofline) and the step without the pipelined index build."""
    from sparse_rcnn_amd.metadata import Metadata
    ex = {}
    n_levels = len(job.channels)
    torch.cuda.synchronize()
    reps = 10
    md = None
    for i in range(reps + 2):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        md = Metadata(3).build_native(job.size, job.coords, job.batch_size, 4, n_levels, 3)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    # algorithmic bytes of the build: per level 16 N (coords) + 8 P (pairs) + 16 N (hash insert + init);
    # strided rulebooks 16 N_fine + 8 N_fine + 16 N_coarse; InputLayer rules 16 Npts + 16 N0
    nbytes = 32.0 * job.coords.shape[0]
    size = tuple(int(s) for s in job.size)
    for l in range(n_levels):
        rb = md.subm.get((size, 3))
        if rb is None:
            break
        nbytes += 32.0 * rb.n + 8.0 * rb.rules.total
        if l + 1 < n_levels and size in md.strided:
            sb = md.strided[size]
            nbytes += 24.0 * sb.n_fine + 16.0 * sb.n_coarse
            size = sb.coarse_size
    ex["index_build_ms"] = ms
    ex["index_build"] = {"ms": ms, "algorithmic_bytes": nbytes, "GBps": nbytes / (ms * 1e-3) / 1e9,
                         "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                         "note": "one scn_pyramid_build call alone on an idle GPU, incl. its host waits for the level sizes"}
    del md
    if job.prefetch:
        job.finish()
        job.prefetch = False
        n = max(5, min(10, args.steps))
        for _ in range(2):
            job.step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            job.step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ex["ms_per_step_no_prefetch"] = (time.perf_counter() - t0) / n * 1e3
        job.prefetch = True
    # ---- forward only (evaluation: eval_model, training.py:244-304; SparseMaskPredictor, model.py:826-882): the same scene
    # under torch.no_grad() -- forward-only slab plan, no backward-data weight images -- with its own peak HBM
    job.finish()
    job.out = job.logits = job.fin = job.rpn_out = None       # (what the last training step left behind)
    job.flat.zero_grad()
    n = max(5, min(20, args.steps))
    for _ in range(3):
        job.forward_only()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    fo_base = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    for _ in range(n):
        job.forward_only()
    job.finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    fo_ms = (time.perf_counter() - t0) / n * 1e3
    ex["forward_only_ms"] = fo_ms
    ex["forward_only"] = {"ms_per_step": fo_ms, "value": job.n_active / job.batches_per_step / (fo_ms * 1e-3),
                          "unit": "active-voxels/s, forward only", "steps": n,
                          "peak_hbm_bytes": int(torch.cuda.max_memory_allocated()),
                          "resident_before_bytes": int(fo_base),
                          "working_set_bytes": int(torch.cuda.max_memory_allocated() - fo_base),
                          "note": "index build (pipelined as in the step) + forward under torch.no_grad(): the executor's "
                                  "forward-only slab plan, no backward-data weight images; bit-equal to the training forward "
                                  "(tests/test_gpu_exec.py::test_forward_only_is_bit_equal...); measured AFTER the training steps "
                                  "in this process: peak_hbm_bytes carries their leftovers (resident_before_bytes) -- "
                                  "`fresh_process` is the same evaluation in a process that never trained"}
    if world == 1 and _FRESH is not None:
        ex["forward_only"]["fresh_process"] = _FRESH
    if args.workload == "cfg2" and world == 1 and args.target is None and args.batches_per_step == 1:
        ex["changing_scenes"] = changing_scenes_leg(job, args, torch)
    if args.dropin and args.workload == "cfg2" and world == 1:
        ex["dropin"] = dropin_measurement(job, args, torch)
    if args.bf16_leg and args.workload == "cfg2" and args.dtype == "f32" and world == 1 and args.target is None:
        ex["bf16"] = bf16_side_leg(args, job.device, torch)
    return ex


def changing_scenes_leg(job, args, torch):
    """The headline times ONE scene repeated (BASELINE configs[1] names a scene); a data loader hands over another scene every
    step.  The same pipelined step over four scenes of 90-165 k voxels in rotation (other seeds than the headline's): shapes,
    rulebooks and tile counts change every step, the caching allocator sees four size classes."""
    from sparse_rcnn_amd.synthetic import make_batch
    m, flat, dev = job.model.backbone, job.flat, job.device
    scenes = []
    for seed, target in ((11, 150_000), (12, 120_000), (13, 165_000), (14, 90_000)):
        c, f, size, bs, _ = make_batch(1, job.grid, target, dup=1.15, seed=seed)
        scenes.append((c.to(dev), f.to(dev), size, bs))
    gys = {}
    n, warm = max(8, min(40, args.steps)), 8
    pending = m.prefetch_in_thread(scenes[0][0], scenes[0][2], scenes[0][3]) if job.prefetch else None
    vox, t0 = 0, None
    for it in range(warm + n):
        if it == warm:
            torch.cuda.synchronize()
            vox, t0 = 0, time.perf_counter()
        c, f, size, bs = scenes[it % 4]
        md = pending.result() if pending is not None else None
        if job.prefetch:
            nc, _, nsize, nbs = scenes[(it + 1) % 4]
            pending = m.prefetch_in_thread(nc, nsize, nbs)
        flat.zero_grad()
        out = m(c, f.detach().requires_grad_(), size, bs, metadata=md)
        gy = gys.get(it % 4)
        if gy is None:
            gy = gys[it % 4] = torch.randn_like(out.features)
        with torch.autograd.set_multithreading_enabled(False):
            out.features.backward(gy)
        flat.step_single_rank(job.lr)
        vox += out.features.shape[0]
    if pending is not None:
        pending.result()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"ms_per_step": dt / n * 1e3, "value": vox / dt, "steps": n, "scenes": "4 scenes of 150 / 120 / 165 / 90 k target voxels "
            "(seeds 11-14) in rotation, a different one every step; same pipelined step"}


def bf16_side_leg(args, dev, torch):
    """BASELINE configs[2..4] store features in bf16: the same cfg-2 step in bf16 STORAGE (fp32 accumulation, fp32
    parameters), a short run outside the timed region, with the roofline of ITS dominant kernel (k_conv_tb: HBM-bound)."""
    from sparse_rcnn_amd import profiling
    from sparse_rcnn_amd.trainstep import SceneStep
    job = SceneStep("cfg2", dev, dtype="bf16", prefetch=args.prefetch, seed=1, grad_seed=100)
    n, warm = max(10, min(40, args.steps)), 10
    timer = profiling.KernelTimer(every=max(5, n // 2), names={"k_conv_tb"})
    for _ in range(warm):
        job.step()
    timer.reserve(1024)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        timer.begin_step()
        profiling.TIMER = timer if timer.active else None
        job.step()
    job.finish()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    profiling.TIMER = None
    ks = timer.summary()
    out = {"ms_per_step": ms, "value": job.n_active / (ms * 1e-3), "steps": n, "warmup": warm,
           "workload": "cfg2 in bf16 storage (bench.py --dtype bf16 times it as the main leg)"}
    for _ in range(3):
        job.forward_only()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        job.forward_only()
    job.finish()
    torch.cuda.synchronize()
    out["forward_only_ms"] = (time.perf_counter() - t0) / n * 1e3
    if "k_conv_tb" in ks:
        out["roofline"] = roofline("k_conv_tb", ks["k_conv_tb"], max(1, timer.sampled_steps),
                                   argparse.Namespace(workload="cfg2", dtype="bf16"))
        shapes = [r for r in timer.by_shape(max(1, timer.sampled_steps)) if r["bf16"]]
        if shapes:
            out["roofline"]["by_shape"] = shapes
    del job
    torch.cuda.empty_cache()
    return out


def dropin_measurement(job, args, torch):
    """The same backbone driven the way the reference's module tree drives the scn surface: CustomInputLayer creates the
    Metadata inside the forward (custom_operations.py:67-83), rulebooks are built lazily by the first layer that needs
    them, no helper thread.  Timed twice: as is, and with the training loop's batches wrapped in scn.index_prefetching
    (one line around the data loader) so that the coming batch's index build overlaps this one's kernels."""
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd.unet import DropinBackbone
    net = DropinBackbone(job.model.backbone)
    n = max(5, min(10, args.steps))

    # what a DataLoader yields: another host coords tensor per batch (same scene here).  Three recycled buffers, as a
    # loader's shared-memory blocks are: a FRESH 5.5 MB host allocation per step costs this pool's boxes a ~90 ms stall
    # every few steps (mmap / page-fault work under the process's mmap lock, which every HIP call of every thread needs;
    # tools/diag_host_coords3.py) -- an effect of the host allocator, not of the path measured here
    ring = [job.coords_cpu.clone() for _ in range(3)]

    def loader(count):
        for i in range(count):
            yield (ring[i % 3], job.size, 1)

    def timed(batches):
        gy, t0 = None, None
        for i, (coords, size, bs) in enumerate(batches):
            if i == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            job.flat.zero_grad()
            fin = job.feats.detach().requires_grad_()
            out = net(coords, fin, size, bs)
            if gy is None:
                gy = torch.randn_like(out.features)
            out.features.backward(gy)
            job.flat.all_reduce_mean()
            job.flat.sgd_step(job.lr)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    from sparse_rcnn_amd import modules as M
    M.STAGE_STATS.update(enc=0, dec=0, layerwise_units=0)
    ms = timed(loader(n + 3))
    stats = dict(M.STAGE_STATS)
    n_fwd = n + 3
    ms_pf = timed(scn.index_prefetching(loader(n + 3), lambda b: b))
    out = {"ms_per_step": ms, "value": job.n_active / (ms * 1e-3),
           "ms_per_step_index_prefetching": ms_pf, "value_index_prefetching": job.n_active / (ms_pf * 1e-3),
           # the module tree's levels ran as step-executor stages (modules._enc_stage / _dec_stage), counted per forward
           "executor": stats["enc"] > 0 and stats["dec"] > 0 and stats["layerwise_units"] == 0,
           "stage_nodes_per_forward": {"encoder": stats["enc"] / n_fwd, "decoder": stats["dec"] / n_fwd},
           "note": "the scn module tree driven as the reference's containers drive it (every encoder level called as one "
                   "scn.Sequential, a decoder level as input stage -> JoinTable -> NetworkInNetwork -> output stage), the "
                   "Metadata created inside the forward from HOST coords (the reference's CustomInputLayer contract), no "
                   "helper thread; round 4: the levels run as step-executor stages (deferred tensors), the index build "
                   "is the fused one; `index_prefetching`: the same with the loop's batches wrapped in "
                   "scn.index_prefetching (the coming batch's index build runs on the helper thread during this batch)"}
    # the same tree with bf16-STORED feature slabs: one package-level switch, nothing in the tree changes
    prev = scn.set_feature_storage(torch.bfloat16)
    try:
        b_ms = timed(loader(n + 3))
        b_pf = timed(scn.index_prefetching(loader(n + 3), lambda b: b))
    finally:
        scn.set_feature_storage(prev)
    out["bf16"] = {"ms_per_step": b_ms, "value": job.n_active / (b_ms * 1e-3), "ms_per_step_index_prefetching": b_pf,
                   "value_index_prefetching": job.n_active / (b_pf * 1e-3),
                   "note": "scn.set_feature_storage(torch.bfloat16) around the same module tree (INTEGRATION.md)"}
    return out


def forward_only_child(args):
    """Evaluation in a process that has never trained (the in-process forward-only leg runs after the training
    steps and its absolute peak carries their leftovers -- gradients, the backward workspaces' cached blocks): the scene, the
    parameters, `forward_only()` steps.  One JSON line."""
    import torch
    import sparse_rcnn_amd  # noqa: F401
    from sparse_rcnn_amd.trainstep import SceneStep
    torch.set_num_threads(_host_threads())
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    job = SceneStep(args.workload, dev, dtype=args.dtype, prefetch=args.prefetch, seed=1, target=args.target)
    for p in job.model.parameters():
        p.requires_grad_(False)
    n = max(5, min(20, args.steps))
    t_w = time.perf_counter()
    k = 0
    while k < 8 or time.perf_counter() - t_w < 2.0:        # a fresh process starts on an idle chip: let the clocks and the
        job.forward_only()                                 # caching allocator settle before anything is timed
        k += 1
    job.finish()
    torch.cuda.synchronize()
    gc.collect()
    gc.freeze()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    st0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    for _ in range(n):
        job.forward_only()
    job.finish()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    st1 = torch.cuda.memory_stats()
    out_rows = int(job.forward_only()[0].features.shape[0])
    print(json.dumps({"ms_per_step": ms, "steps": n, "active_voxels": out_rows, "value": out_rows / (ms * 1e-3),
                      "unit": "active-voxels/s, forward only",
                      "peak_hbm_bytes": int(torch.cuda.max_memory_allocated()),
                      "peak_reserved_bytes": int(torch.cuda.max_memory_reserved()),
                      "resident_before_bytes": int(base),
                      "working_set_bytes": int(torch.cuda.max_memory_allocated() - base),
                      "device_allocs_in_timed_region": int(st1["num_device_alloc"] - st0["num_device_alloc"]),
                      "device_frees_in_timed_region": int(st1["num_device_free"] - st0["num_device_free"]),
                      "note": "a process that only evaluates: parameters (no gradients), the scene, the index structures of two "
                              "batches (this one + the prefetched one), the forward-only slab plan"}), flush=True)
    return 0


def forward_only_fresh(args):
    """Run `forward_only_child` as a child process (its own HIP context; this process keeps its GPU state) -> dict | error."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--forward-only-child", "--workload", args.workload, "--dtype", args.dtype,
           "--steps", str(args.steps)]
    if args.target is not None:
        cmd += ["--target", str(args.target)]
    if not args.prefetch:
        cmd += ["--no-prefetch"]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            return {"error": f"child exit {p.returncode}: {p.stderr.decode()[-300:]}"}
        return json.loads(line[-1])
    except Exception as e:                       # noqa: BLE001  (a side figure must not take the headline down)
        return {"error": repr(e)}


_FRESH = None          # `forward_only.fresh_process` of this run: measured by a child BEFORE this process touches the GPU


def main():
    global _FRESH
    args = parse_args()
    if args.forward_only_child:
        sys.exit(forward_only_child(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.no_extras:
        # the evaluation-only process runs first and alone: as a child of a process that holds a HIP context of its own the
        # same forward took 3 ms longer per step at 600 k voxels (two contexts time-sliced on one GPU)
        _FRESH = forward_only_fresh(args)
    sys.exit(run(args))


if __name__ == "__main__":
    main()
