"""CPU: host-side logic -- synthetic generator, flat-bucket data parallelism over gloo (world_size 2)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import scn_oracle as O
from sparse_rcnn_amd.dp import FlatParams, broadcast_params
from sparse_rcnn_amd.synthetic import make_batch, make_boxes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synthetic_scene_is_deterministic_and_surface_like():
    a = make_batch(1, (128, 128, 64), 12000, seed=5)
    b = make_batch(1, (128, 128, 64), 12000, seed=5)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    coords, feats, size, bs, splits = a
    assert coords.dtype == torch.int64 and coords.shape[1] == 4 and feats.shape[1] == 7 and feats.dtype == torch.float32
    assert (coords[:, :3] >= 0).all() and (coords[:, :3] < size).all() and splits == [len(coords)]
    ac, prow, cnt = O.input_layer_rules(coords.numpy())
    assert len(ac) == 12000 and 1.1 < len(coords) / len(ac) < 1.2            # ~1.15 points per voxel
    nbr, rules = O.subm_rulebook(ac, 3)
    per_voxel = sum(len(i) for i, _ in rules) / len(ac)
    assert 5.0 < per_voxel < 14.0                                             # 2-D surfaces, not noise (27) nor dust (1)
    boxes = make_boxes(coords, 5)
    assert len(boxes) == 1 and boxes[0].shape == (5, 2, 3) and (boxes[0][:, 1] > boxes[0][:, 0]).all()


def test_batch_rows_are_sample_major():
    coords, feats, size, bs, splits = make_batch(3, (64, 64, 32), 2000, seed=1)
    b = coords[:, 3].numpy()
    assert (np.diff(b) >= 0).all() and bs == 3 and sum(splits) == len(coords)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _dp_worker(rank, world, port, out, n_buckets=0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(rank)                                   # different init per rank: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    if n_buckets:                                             # a parameter that never receives a gradient
        net.unused = torch.nn.Parameter(torch.ones(7))
    fp = FlatParams(net, n_buckets=n_buckets)
    broadcast_params(fp)
    x = torch.full((4, 6), float(rank + 1))                   # rank-dependent "scene"
    fp.zero_grad()
    net(x).sum().backward()
    if n_buckets:                                             # overlapped path: slices were reduced during backward
        assert len(fp.buckets) >= 2
        local = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in fp.params])
    else:
        fp.gather_grads()
        local = fp.flat_grad.clone()
    fp.all_reduce_mean()
    fp.sgd_step(0.1)
    # by value (numpy): a tensor would travel as a shared-memory handle that dies with this process
    out.put((rank, fp.flat.detach().numpy().copy(), local.detach().numpy().copy(), fp.mean_grad().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


class _TwoHeads(torch.nn.Module):
    """A 'backbone' and a 'mask branch' whose forward may be skipped (a rank whose ROI crop is empty)."""

    def __init__(self):
        super().__init__()
        self.backbone = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
        self.mask = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.Linear(8, 2))


def _dp_worker_uneven(rank, world, port, out, weighted):
    """Rank 1 never runs its mask branch (no gradients for those parameters): every rank must still issue the bucket
    all-reduces in the same order.  weighted: ranks contribute in proportion to a per-rank count (set before backward)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = _TwoHeads()
    fp = FlatParams(net, n_buckets=3)
    broadcast_params(fp)
    assert len(fp.buckets) >= 2
    res = []
    for step in range(2):
        x = torch.full((4, 6), float(rank + 1 + step))
        fp.zero_grad()
        w = float(rank + 1) if weighted else 1.0
        fp.rank_weight = w
        h = net.backbone(x)
        loss = h.sum()
        if rank == 0:                                  # rank 1: empty crop, the mask head never runs
            loss = loss + net.mask(h).square().sum()
        order = []
        orig = fp._launch_bucket
        fp._launch_bucket = lambda b, orig=orig: (order.append(b), orig(b))[1]
        loss.backward()
        local = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in fp.params])
        fp.all_reduce_mean(total_weight=3.0 if weighted else None)
        fp._launch_bucket = orig
        assert order == list(range(len(fp.buckets))), order       # strictly in index order, on every rank
        res.append((local.numpy().copy(), fp.mean_grad().numpy().copy(), w))
        fp.sgd_step(0.1)
    out.put((rank, fp.flat.detach().numpy().copy(), res))
    dist.barrier()
    dist.destroy_process_group()


def _dp_worker_accum(rank, world, port, out, n_buckets, bps):
    """`bps` micro-batches per optimizer step (training.py:436,458-460): all but the last under fp.accumulate().  The mask
    head runs on rank 0 in the FIRST micro-batch only: its parameters hold an accumulated gradient whose hook never fires
    in the last backward (the forced drain must still pack it)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = _TwoHeads()
    fp = FlatParams(net, n_buckets=n_buckets)
    broadcast_params(fp)
    assert bool(fp.buckets) == bool(n_buckets)
    res = []
    for step in range(2):
        fp.zero_grad()
        launched = []
        orig = fp._launch_bucket
        fp._launch_bucket = lambda b, orig=orig: (launched.append(b), orig(b))[1]
        parts = []
        for k in range(bps):
            x = torch.full((4, 6), float(rank + 1 + step)) + 0.25 * k + torch.arange(6.0) * 0.1
            h = net.backbone(x)
            loss = h.sum()
            if rank == 0 and k == 0:
                loss = loss + net.mask(h).square().sum()
            gs = torch.autograd.grad(loss / bps, fp.params, allow_unused=True, retain_graph=True)
            parts.append(torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(gs, fp.params)]))
            if k < bps - 1:
                with fp.accumulate():
                    (loss / bps).backward()
                assert launched == [], launched            # nothing goes out before the last micro-batch
            else:
                (loss / bps).backward()
        fp.all_reduce_mean()
        fp._launch_bucket = orig
        if n_buckets:
            assert launched == list(range(len(fp.buckets))), launched        # every slice exactly once, in order
        res.append((torch.stack(parts).numpy().copy(), fp.mean_grad().numpy().copy()))
        fp.sgd_step(0.05)
    out.put((rank, fp.flat.detach().numpy().copy(), res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bps", [2, 6])
def test_gradient_accumulation_world2_bucketed_equals_unbucketed(bps):
    """`batches_per_step` micro-batches, ONE all-reduce per optimizer step.  The reduced gradient == the
    single-process accumulation over all micro-batches of both ranks, and the bucketed path is bit-equal to the unbucketed
    one (two optimizer steps, so that zero_grad re-arms the hooks)."""
    ctx = mp.get_context("spawn")
    got = {}
    for nb in (0, 3):
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_dp_worker_accum, args=(r, 2, port, q, nb, bps)) for r in range(2)]
        for p in ps: p.start()
        res = sorted([q.get(timeout=180) for _ in ps], key=lambda t: t[0])
        for p in ps: p.join(60)
        (_, w0, r0), (_, w1, r1) = res
        assert np.array_equal(w0, w1)                                   # ranks stay in lock-step
        for (p0, m0), (p1, m1) in zip(r0, r1):
            assert np.array_equal(m0, m1)
            want = (p0.astype(np.float64).sum(0) + p1.astype(np.float64).sum(0)) / 2     # all micro-batches of both ranks
            assert np.allclose(m0, want, rtol=2e-6, atol=1e-6), np.abs(m0 - want).max()
            assert np.abs(p0[1:]).sum() > 0                             # the later micro-batches do contribute
            first_only = (p0[0] + p1[0]) / 2
            assert np.abs(m0 - first_only).max() > 1e-2                 # (what round 4's hooks silently produced)
        got[nb] = (w0, [m for _, m in r0])
    assert np.array_equal(got[0][0], got[3][0])
    for a, b in zip(got[0][1], got[3][1]):
        assert np.array_equal(a, b)                                     # bucketed == unbucketed, bit for bit


def test_second_backward_with_armed_hooks_raises():
    """A second backward with the bucket hooks armed and no zero_grad() (accumulation WITHOUT fp.accumulate()) must not be
    dropped silently: the hook raises, naming the remedy.  Inside fp.accumulate() the same sequence is fine."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", WORLD_SIZE="1", SCN_DP_FORCE_BUCKETS="1")
dist.init_process_group("gloo")
from sparse_rcnn_amd.dp import FlatParams
net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2))
fp = FlatParams(net, n_buckets=2)
assert fp.buckets
x = torch.ones(1, 3)
fp.zero_grad(); net(x).sum().backward()
try:
    net(2 * x).sum().backward()
    print("NOT RAISED")
except RuntimeError as e:
    assert "accumulate" in str(e), e
    print("RAISED")
fp.all_reduce_mean()
fp.zero_grad()
with fp.accumulate():
    net(x).sum().backward()
net(2 * x).sum().backward()
fp.all_reduce_mean()
g1 = fp.mean_grad().clone()
fp.zero_grad(); net(x).sum().backward(); fp.all_reduce_mean(); a = fp.mean_grad().clone()
fp.zero_grad(); net(2 * x).sum().backward(); fp.all_reduce_mean(); b = fp.mean_grad().clone()
assert torch.allclose(g1, a + b), (g1, a + b)
try:
    with fp.accumulate():
        fp.all_reduce_mean()
    print("NO GUARD")
except RuntimeError:
    print("GUARDED")
''' % (ROOT, str(_free_port()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RAISED" in r.stdout and "NOT RAISED" not in r.stdout and "GUARDED" in r.stdout, \
        (r.stdout, r.stderr[-2000:])


def test_flat_bucket_all_reduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps: p.join(60)
    (_, w0, g0, m0), (_, w1, g1, m1) = [(r, *map(torch.from_numpy, t)) for r, *t in res]
    assert torch.equal(w0, w1)                                 # ranks stay in lock-step
    assert torch.allclose(m0, (g0 + g1) / 2) and torch.equal(m0, m1)


def test_bucketed_overlapped_all_reduce_world2():
    """n_buckets > 0: slices of the flat gradient are all-reduced from post-accumulate hooks while backward runs; same
    result as the single all-reduce, also with a parameter that receives no gradient, and over two steps."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, 3)) for r in range(2)]
    for p in ps: p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps: p.join(60)
    (_, w0, g0, m0), (_, w1, g1, m1) = [(r, *map(torch.from_numpy, t)) for r, *t in res]
    assert torch.equal(w0, w1)
    assert torch.allclose(m0, (g0 + g1) / 2) and torch.equal(m0, m1)
    assert torch.equal(m0[:7], torch.zeros(7))                # the unused parameter (first in parameters()) stays zero


@pytest.mark.parametrize("weighted", [False, True])
def test_bucket_order_with_a_rank_whose_mask_branch_gets_no_gradient(weighted):
    """a rank with an empty ROI crop gives its mask-branch parameters no gradient; buckets still go out in
    index order on every rank, the result is the (count-weighted) mean with zeros for the missing gradients, over two steps."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_dp_worker_uneven, args=(r, 2, port, q, weighted)) for r in range(2)]
    for p in ps: p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps: p.join(60)
    (_, w0, r0), (_, w1, r1) = res
    assert np.array_equal(w0, w1)
    for (l0, m0, a0), (l1, m1, a1) in zip(r0, r1):
        assert np.array_equal(m0, m1)
        denom = 3.0 if weighted else 2.0
        assert np.allclose(m0, (a0 * l0 + a1 * l1) / denom, rtol=1e-6, atol=1e-7)
        assert np.abs(l1).sum() > 0 and np.abs(l0 - l1).sum() > 0


def test_weight_given_after_backward_is_rejected_on_the_bucketed_path():
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", WORLD_SIZE="1", SCN_DP_FORCE_BUCKETS="1")
dist.init_process_group("gloo")
from sparse_rcnn_amd.dp import FlatParams
net = torch.nn.Linear(3, 2)
fp = FlatParams(net, n_buckets=2)
fp.zero_grad(); net(torch.ones(1, 3)).sum().backward()
try:
    fp.all_reduce_mean(weight=2.0)
except ValueError as e:
    print("OK", e)
''' % (ROOT, str(_free_port()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_flat_params_views_survive_backward():
    net = torch.nn.Linear(4, 3)
    fp = FlatParams(net)
    for _ in range(2):
        fp.zero_grad()
        net(torch.ones(2, 4)).sum().backward()
        fp.gather_grads()
        assert torch.allclose(fp.flat_grad[:12], torch.full((12,), 2.0))        # no accumulation across steps
    w = fp.flat.clone()
    fp.sgd_step(0.5)
    assert torch.allclose(fp.flat, w - 0.5 * fp.flat_grad) and net.weight.data_ptr() == fp.flat.data_ptr()


# ---- bench.py --gpus N without torchrun: the parent starts the ranks itself and never touches the GPU ------------------
def test_bench_self_launch_starts_fresh_ranks(tmp_path, capsys):
    import bench
    script = tmp_path / "rank.py"
    script.write_text(
        "import os, sys, json\n"
        "assert 'torch' not in sys.modules\n"
        "if os.environ['RANK'] == '0':\n"
        "    print(json.dumps({k: os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}))\n"
        "sys.exit(0 if sys.argv[1:] == ['--gpus', '3'] else 7)\n")
    args = bench.parse_args(["--gpus", "3"])
    before = "torch.cuda" in sys.modules and sys.modules["torch"].cuda.is_initialized()
    rc = bench.self_launch(args, script=str(script), argv=["--gpus", "3"])
    import json
    env = json.loads(capsys.readouterr().out.strip())
    assert rc == 0 and env["WORLD_SIZE"] == "3" and env["RANK"] == "0" and env["MASTER_ADDR"] == "127.0.0.1"
    import torch
    assert torch.cuda.is_initialized() == bool(before)              # launching made no GPU call in this process
    assert bench.self_launch(args, script=str(script), argv=["--other"]) == 7     # a failing rank fails the launch


def test_bench_self_launch_returns_promptly_when_a_rank_dies(tmp_path, capsys):
    """rank 1 exits with code 3 while rank 0 would wait (here: sleep) for minutes -- the launcher
    terminates the survivor, reports the dead rank's stderr tail and returns its code within seconds."""
    import time
    import bench
    script = tmp_path / "rank.py"
    script.write_text(
        "import os, sys, time\n"
        "if os.environ['RANK'] == '1':\n"
        "    sys.stderr.write('rank 1: simulated RCCL init failure\\n'); sys.exit(3)\n"
        "time.sleep(120)\n"
        "print('never printed')\n")
    args = bench.parse_args(["--gpus", "2"])
    t0 = time.time()
    rc = bench.self_launch(args, script=str(script), argv=[])
    took = time.time() - t0
    cap = capsys.readouterr()
    assert rc == 3 and took < 20, (rc, took)
    assert "never printed" not in cap.out and "simulated RCCL init failure" in cap.err and "rank 1 exited with code 3" in cap.err


def test_bench_dtype_aliases():
    import bench
    assert bench.parse_args(["--bf16-all"]).dtype == "bf16" and bench.parse_args(["--bf16-blocks"]).dtype == "bf16-blocks"
    a = bench.parse_args([])
    assert a.gpus == 1 and a.workload == "cfg2" and a.dtype == "f32" and a.prefetch


# ---- ROI classes are constructed the way the reference constructs them (model.py:577-580,625; roi_select_sparse.py:29-36)
def test_roi_cut_signature_is_the_reference_one():
    from sparse_rcnn_amd import roi
    cut = roi.SparseRoiCut(roi.RawToTensorFeatureExtractorCombiner())
    assert cut.clip_boxes is False and cut.resize_boxes is None
    cut = roi.SparseRoiCut(roi.RawToRawFeatureExtractorCombiner(), True, (2, 2, 2))
    assert cut.clip_boxes is True and cut.resize_boxes == (2, 2, 2)
    with pytest.raises(TypeError):
        roi.SparseRoiCut(True)                                        # a flag where the combiner belongs
    with pytest.raises(TypeError):
        roi.SparseRoiCut(roi.RawToRawFeatureExtractorCombiner(), roi.RawToRawFeatureExtractorCombiner())
    roi.SparseRoiExtraCut(roi.RawToFeaturesSceneFeatureExtractorCombiner())
    scene = (torch.zeros(3, 4, dtype=torch.long), torch.ones(3, 2), torch.tensor([8, 8, 8]), 1, [3])
    for comb in (roi.RawToTensorFeatureExtractorCombiner(), roi.RawToRawFeatureExtractorCombiner(),
                 roi.RawToFeaturesSceneFeatureExtractorCombiner()):
        c, f, s, splits = comb.extract(scene)                           # instances, as the reference passes them
        assert splits == [3] and f.shape == (3, 2)
    assert roi.RawToRawFeatureExtractorCombiner().combine(1, 2, 3, 4) == (1, 2, 3, 4)
    assert roi.RawToTensorFeatureExtractorCombiner().combine(torch.zeros(0, 4, dtype=torch.long), torch.zeros(0, 2),
                                                            torch.tensor([8, 8, 8]), 2) is None


def test_last_gradient_bucket_is_small():
    """dp.FlatParams cuts its buckets by when backward completes them: the last one (whose all-reduce nothing hides)
    holds at most ~5 % of the bytes, every parameter sits in exactly one bucket, and the slices tile the flat buffer in
    reverse parameter order.  Runs in a child process (it creates a one-rank gloo group)."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", WORLD_SIZE="1", SCN_DP_FORCE_BUCKETS="1")
dist.init_process_group("gloo")
from sparse_rcnn_amd.dp import FlatParams
sizes = [7 * 32, 32] + [27 * 32 * 32, 32] * 4 + [8 * 32 * 64, 64] + [27 * 64 * 64, 64] * 4 + [8 * 64 * 128, 128] + [27 * 128 * 128, 128] * 4
net = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(n)) for n in sizes])
fp = FlatParams(net, n_buckets=4)
total = fp.flat.numel()
assert len(fp.buckets) == 4, len(fp.buckets)
ids = [i for b, _ in fp.buckets for i in b]
assert ids == list(range(len(sizes) - 1, -1, -1)), ids                 # reverse parameter order, each exactly once
assert sum(s.numel() for _, s in fp.buckets) == total
off = total
for b, s in fp.buckets:                                                # contiguous slices, back to front
    off -= s.numel()
    assert s.data_ptr() == fp.flat_grad.data_ptr() + 4 * off
assert 0 < fp.buckets[-1][1].numel() <= 0.06 * total, fp.buckets[-1][1].numel() / total
one = FlatParams(torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(5))]), n_buckets=4)
assert len(one.buckets) == 1 and one.buckets[0][1].numel() == 5
print("OK")
''' % (ROOT, str(_free_port()))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_single_rank_step_equals_packed_step():
    """FlatParams.step_single_rank (no process group: update straight from the per-parameter gradients) == all_reduce_mean +
    sgd_step through the flat bucket, bit for bit."""
    torch.manual_seed(0)
    def make():
        torch.manual_seed(1)
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
        return net, FlatParams(net)
    x = torch.randn(11, 5)
    outs = []
    for single in (True, False):
        net, fp = make()
        for _ in range(3):
            fp.zero_grad()
            net(x).square().sum().backward()
            if single:
                fp.step_single_rank(1e-2)
            else:
                fp.all_reduce_mean()
                fp.sgd_step(1e-2)
        outs.append(fp.flat.clone())
    assert torch.equal(outs[0], outs[1])
