"""-m gpu: the step executor (sparse_rcnn_amd/executor.py + csrc/scn_exec.hip: one autograd node and two C calls per U-Net
level) against the layer-by-layer module path.  Both drive the same entry points with the same arguments, so EVERYTHING must
be bit-identical: output features, encoder outputs, the input-feature gradient and every parameter gradient -- fp32 and bf16
storage, the benchmark plan, the reference's own plan 32-48-64-80-96-112, the mask branch (channel-padded internal U-Net,
cast-in / cast-out input stage), gradients arriving at the encoder outputs (the RPN's inputs in the reference)."""
import os
import sys

import numpy as np
import pytest
import torch

from sparse_rcnn_amd._lib import switches as _SW      # developer switches of the library: scn_debug_set, not the environment

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))          # tests/optape.py

pytestmark = pytest.mark.gpu


def _scene(target, grid, seed):
    from sparse_rcnn_amd.synthetic import make_batch
    return make_batch(1, grid, target, dup=1.15, seed=seed)


def _run_backbone(net, coords, feats, size, gy_seed, gpu, use_exec, with_interims=False):
    from sparse_rcnn_amd.unet import SparseUNet
    SparseUNet.EXEC = use_exec
    try:
        for p in net.parameters():
            p.grad = None
        fin = feats.to(gpu).requires_grad_()
        out = net(coords, fin, size, 1)
        g = torch.Generator().manual_seed(gy_seed)
        gy = torch.randn(out.features.shape, generator=g).to(gpu)
        outs, grads = [out.features], [gy]
        if with_interims:                       # gradients arriving at encoder outputs too (SparseToDense -> RPN in the reference)
            for t in net.unet.interims[1:]:
                outs.append(t.features)
                grads.append(torch.randn(t.features.shape, generator=g).to(gpu).to(t.features.dtype))
        torch.autograd.backward(outs, grads)
        torch.cuda.synchronize()
        return (out.features.detach().clone(), [t.features.detach().clone() for t in net.unet.interims], fin.grad.clone(),
                [p.grad.clone() for p in net.parameters()])
    finally:
        SparseUNet.EXEC = True


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("plan", ["bench", "reference", "two-level"])
def test_executor_backbone_is_bit_identical_to_the_module_path(gpu, dtype, plan):
    from sparse_rcnn_amd.unet import Backbone
    ch, grid, target = {"bench": ((32, 64, 128, 256), (256, 256, 128), 30_000),
                        "reference": ((32, 48, 64, 80, 96, 112), (256, 256, 128), 20_000),
                        "two-level": ((16, 40), (64, 64, 32), 3_000)}[plan]
    coords, feats, size, bs, _ = _scene(target, grid, seed=7)
    torch.manual_seed(3)
    net = Backbone(7, ch, bf16_blocks="all" if dtype == "bf16" else False).to(gpu)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.05)
    assert net.unet._exec_plan(), "this network must be covered by the executor"
    for with_interims in (False, True):
        a = _run_backbone(net, coords, feats, size, 11, gpu, True, with_interims)
        b = _run_backbone(net, coords, feats, size, 11, gpu, False, with_interims)
        assert torch.equal(a[0], b[0]), "output features"
        for l, (x, y) in enumerate(zip(a[1], b[1])):
            assert x.dtype == y.dtype and torch.equal(x, y), f"encoder output {l}"
        assert torch.equal(a[2], b[2]), "input-feature gradient"
        names = [n for n, _ in net.named_parameters()]
        for n, x, y in zip(names, a[3], b[3]):
            assert torch.equal(x, y), f"gradient of {n} (interims={with_interims})"


def test_executor_is_actually_used_and_cuts_the_autograd_graph_to_one_node_per_level(gpu):
    from sparse_rcnn_amd.unet import Backbone
    from sparse_rcnn_amd import executor as EX
    coords, feats, size, bs, _ = _scene(3000, (64, 64, 32), seed=2)
    net = Backbone(7, (16, 32, 64)).to(gpu)
    out = net(coords, feats.to(gpu).requires_grad_(), size, 1)
    seen, todo, n_stage = set(), [out.features.grad_fn], 0
    while todo:
        f = todo.pop()
        if f is None or f in seen:
            continue
        seen.add(f)
        n_stage += type(f).__name__.startswith("StageFunction")
        todo += [g for g, _ in f.next_functions]
    assert n_stage == 3 + 2, n_stage                          # 3 encoder levels + 2 decoder levels


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_executor_mask_branch_is_bit_identical_to_the_module_path(gpu, dtype):
    """trainstep's cfg3 step (backbone -> input stage -> OutputLayer -> crop -> channel-padded internal U-Net -> Linear) with
    the executor on and off: logits, backbone output and every gradient bit for bit."""
    from sparse_rcnn_amd.trainstep import SceneStep
    from sparse_rcnn_amd.unet import SparseUNet
    res = []
    for use in (True, False):
        SparseUNet.EXEC = use
        try:
            job = SceneStep("cfg3", gpu, dtype=dtype, prefetch=False, seed=5, grad_seed=9, target=20_000, grid=(256, 256, 128),
                            n_boxes=12, lr=0.0)
            with torch.no_grad():
                g = torch.Generator().manual_seed(1)
                for p in job.model.parameters():
                    if p.dim() == 1:
                        p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            job.forward_backward()
            torch.cuda.synchronize()
            res.append((job.out.features.detach().clone(), job.logits.detach().clone(), job.fin.grad.clone(),
                        [(n, p.grad.clone()) for n, p in job.model.named_parameters()]))
        finally:
            SparseUNet.EXEC = True
    a, b = res
    assert a[1].shape[0] > 1000
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for (n, x), (_, y) in zip(a[3], b[3]):
        assert torch.equal(x, y), n


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("workload", ["cfg2", "cfg3"])
def test_executor_is_bit_identical_to_the_module_path_at_baseline_size(gpu, workload, dtype):
    """the path bench.py TIMES (step executor: scn_exec_run, deferred weight-gradient sums, one autograd
    node per level) against the layer-by-layer module path at BASELINE size -- the seed-1 150k-voxel scene of configs[1] and
    the configs[2] step (150k voxels + 64 boxes -> crop -> mask branch), fp32 and bf16 storage.  Kernel plans depend on the
    size (fast-path thresholds, the weight-gradient unit plan, XCD-local hand-out), so the small-scene identity tests above do
    not cover this point.  Every output and every gradient: torch.equal."""
    from sparse_rcnn_amd.trainstep import SceneStep
    from sparse_rcnn_amd.unet import SparseUNet
    res = []
    for use in (True, False):
        SparseUNet.EXEC = use
        try:
            job = SceneStep(workload, gpu, dtype=dtype, prefetch=False, seed=1, grad_seed=100, lr=0.0)
            with torch.no_grad():
                g = torch.Generator().manual_seed(21)
                for p in job.model.parameters():
                    if p.dim() == 1:
                        p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            job.forward_backward()
            torch.cuda.synchronize()
            assert job.out.features.shape[0] == 150_000
            n_stage, seen, todo = 0, set(), [job.out.features.grad_fn]
            while todo:
                f = todo.pop()
                if f is None or f in seen:
                    continue
                seen.add(f)
                n_stage += type(f).__name__.startswith("StageFunction")
                todo += [h for h, _ in f.next_functions]
            assert (n_stage == 7) if use else (n_stage == 0), (use, n_stage)      # the executor really ran / really did not
            res.append((job.out.features.detach().clone(), None if job.logits is None else job.logits.detach().clone(),
                        job.fin.grad.clone(), [(n, p.grad.clone()) for n, p in job.model.named_parameters()]))
            del job
        finally:
            SparseUNet.EXEC = True
    a, b = res
    assert torch.equal(a[0], b[0]), "backbone output"
    if workload == "cfg3":
        assert a[1].shape[0] > 100_000 and torch.equal(a[1], b[1]), "mask logits"
    assert torch.equal(a[2], b[2]), "input-feature gradient"
    assert len(a[3]) == (76 + 80 if workload == "cfg3" else 76)
    for (n, x), (_, y) in zip(a[3], b[3]):
        assert torch.isfinite(x).all() and torch.equal(x, y), n


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_executor_backward_twice_with_retain_graph_and_error_without(gpu, dtype):
    """ADVICE r3 (executor.py): a stage node keeps its forward workspace, weight images and inputs through
    save_for_backward -- a second backward with retain_graph=True reads the same slabs and gives the same bits as the
    first (and as the module path); without retain_graph the second backward raises autograd's own error instead of
    launching kernels on freed memory."""
    from sparse_rcnn_amd.unet import Backbone, SparseUNet
    coords, feats, size, bs, _ = _scene(8_000, (128, 128, 64), seed=3)
    torch.manual_seed(5)
    net = Backbone(7, (16, 32, 64), bf16_blocks="all" if dtype == "bf16" else False).to(gpu)
    gy = None
    got = {}
    for use in (True, False):
        SparseUNet.EXEC = use
        try:
            fin = feats.to(gpu).requires_grad_()
            out = net(coords, fin, size, 1).features
            if gy is None:
                gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(2)).to(gpu)
            passes = []
            for k in range(2):
                for p in net.parameters():
                    p.grad = None
                fin.grad = None
                out.backward(gy, retain_graph=(k == 0))
                torch.cuda.synchronize()
                # (something else takes the freed blocks between the passes: a stale pointer would show)
                junk = [torch.full((1 << 20,), float("nan"), device=gpu) for _ in range(8)]
                passes.append([fin.grad.clone()] + [p.grad.clone() for p in net.parameters()])
                del junk
            for x, y in zip(*passes):
                assert torch.equal(x, y)
            got[use] = passes[0]
            with pytest.raises(RuntimeError, match="second time|already been freed"):
                out.backward(gy)
        finally:
            SparseUNet.EXEC = True
    for x, y in zip(got[True], got[False]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_executor_records_the_relu_masks_of_the_module_path(gpu, dtype):
    """functional.RELU_RECORD with the executor ON (round 4: the at-size oracle tests now run the path bench.py times): the
    sign masks a stage records from the slabs it keeps for backward == the masks the layer-by-layer path records, in the
    same order -- backbone and the mask branch's input stage / channel-padded internal U-Net."""
    from sparse_rcnn_amd import functional as F
    from sparse_rcnn_amd.trainstep import SceneStep
    from sparse_rcnn_amd.unet import SparseUNet
    rec = []
    for use in (True, False):
        SparseUNet.EXEC = use
        try:
            job = SceneStep("cfg3", gpu, dtype=dtype, prefetch=False, seed=5, grad_seed=9, target=10_000, grid=(128, 128, 64),
                            n_boxes=8, lr=0.0)
            F.RELU_RECORD = []
            job.forward_backward()
            torch.cuda.synchronize()
            rec.append(F.RELU_RECORD)
        finally:
            F.RELU_RECORD = None
            SparseUNet.EXEC = True
    assert len(rec[0]) == len(rec[1]) == 31 + 32
    for k, (x, y) in enumerate(zip(*rec)):
        assert x.shape == y.shape and torch.equal(x, y), k


class _Interims(torch.nn.Sequential):
    """Control flow of the reference's SequentialInterims (custom_container.py:5-12): apply the children in turn, keep every
    output.  Test-side stand-in: the reference's containers do not travel to the GPU box."""

    def forward(self, x):
        outs = []
        for m in self:
            x = m(x)
            outs.append(x)
        return outs


class _Reuniter(torch.nn.Module):
    """Control flow of the reference's SkipConnectionReuniter (custom_container.py:70-83)."""

    def __init__(self, input_stage, combiner, channel_changer, output_stage):
        super().__init__()
        self.input_stage, self.combiner, self.channel_changer, self.output_stage = input_stage, combiner, channel_changer, output_stage

    def forward(self, x, skip):
        return self.output_stage(self.channel_changer(self.combiner([self.input_stage(x), skip])))


class _ReferenceShapedTree(torch.nn.Module):
    """The module tree the reference's factory builds for sparse + U-Net (tests/golden/dropin_feature_extractor.json `repr`:
    main_network = SequentialInterims of scn.Sequential(scn.Sequential(head), scn.Sequential(units)); unet.module_list =
    SkipConnectionReuniter per decoder level), assembled from the modules (= the parameters) of a Backbone."""

    def __init__(self, backbone):
        super().__init__()
        import sparse_rcnn_amd as scn
        u = backbone.unet
        self.main_network = _Interims(*[scn.Sequential(scn.Sequential(lvl[0]), lvl[1]) for lvl in u.encoder])
        self.module_list = torch.nn.ModuleList([_Reuniter(d["up"], d["join"], d["nin"], d["units"]) for d in u.decoder])

    def forward(self, coords, feats, size, bs):
        import sparse_rcnn_amd as scn
        x = scn.InputLayer(3, size, mode=4)((coords, feats, bs))
        *skips, x = self.main_network(x)
        self.interims = skips + [x]
        for m, skip in zip(self.module_list, skips[::-1]):
            x = m(x, skip)
        return x


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("plan", ["bench", "reference"])
def test_module_tree_stages_are_bit_identical_to_layer_by_layer(gpu, dtype, plan):
    """a module tree somebody else built -- the reference's shape, driven by the reference's container
    control flow -- gets the step executor: encoder levels recognised by scn.Sequential, decoder levels through deferred
    tensors (modules._enc_stage / _dec_stage).  Stages on vs off (SCN_TREE_STAGES): output, every encoder output, input
    gradient and every parameter gradient bit for bit, fp32 and bf16 storage (scn.set_feature_storage), the benchmark plan
    and the reference's 32-48-64-80-96-112.  bf16: the tree's result also equals the Backbone's own bf16 mode (the one the
    oracle tests check), whose final cast to fp32 is the only difference."""
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import modules as M
    from sparse_rcnn_amd.unet import Backbone
    ch, target = {"bench": ((32, 64, 128, 256), 30_000), "reference": ((32, 48, 64, 80, 96, 112), 20_000)}[plan]
    coords, feats, size, bs, _ = _scene(target, (256, 256, 128), seed=7)
    torch.manual_seed(3)
    net = Backbone(7, ch, bf16_blocks="all" if dtype == "bf16" else False).to(gpu)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.05)
    tree = _ReferenceShapedTree(net)
    prev = scn.set_feature_storage(torch.bfloat16 if dtype == "bf16" else torch.float32)
    res = []
    try:
        for stages in (True, False):
            M.TREE_STAGES = stages
            M.STAGE_STATS.update(enc=0, dec=0, layerwise_units=0)
            for p in net.parameters():
                p.grad = None
            fin = feats.to(gpu).requires_grad_()
            out = tree(coords, fin, size, 1)
            assert out.features.dtype == (torch.bfloat16 if dtype == "bf16" else torch.float32)
            g = torch.Generator().manual_seed(11)
            outs = [out.features] + [t.features for t in tree.interims[1:]]
            grads = [torch.randn(o.shape, generator=g).to(gpu).to(o.dtype) for o in outs]
            torch.autograd.backward(outs, grads)
            torch.cuda.synchronize()
            L = len(ch)
            if stages:
                assert M.STAGE_STATS == dict(enc=L, dec=L - 1, layerwise_units=0), M.STAGE_STATS
            else:
                assert M.STAGE_STATS["enc"] == 0 and M.STAGE_STATS["dec"] == 0
            res.append((out.features.detach().clone(), [t.features.detach().clone() for t in tree.interims], fin.grad.clone(),
                        [p.grad.clone() for p in net.parameters()]))
    finally:
        M.TREE_STAGES = True
        scn.set_feature_storage(prev)
    a, b = res
    assert torch.equal(a[0], b[0]), "output features"
    for l, (x, y) in enumerate(zip(a[1], b[1])):
        assert x.dtype == y.dtype and torch.equal(x, y), f"encoder output {l}"
    assert torch.equal(a[2], b[2]), "input-feature gradient"
    for (n, _), x, y in zip(net.named_parameters(), a[3], b[3]):
        assert torch.equal(x, y), f"gradient of {n}"
    # the Backbone's own forward on the same parameters (its bf16 mode ends with a cast to fp32)
    ref = net(coords, feats.to(gpu), size, 1).features
    assert torch.equal(ref, a[0].float())


def test_stage_plans_follow_edits_of_the_inner_module_tree(gpu):
    """ADVICE r4 (modules.py): the compiled plan of an encoder level captures the INNER Sequential's residual units.  After one
    forward: a unit appended to the inner Sequential, a unit replaced through __setitem__, a unit deleted, a bias removed --
    the next forward must run the EDITED tree (== the same tree with stages off, bit for bit), not the cached plan; and a
    deep copy of a tree that has run (ctypes tables in its caches) works and computes the same."""
    import copy
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import modules as M
    coords, feats, size, bs, _ = _scene(6_000, (64, 64, 32), seed=5)
    torch.manual_seed(2)

    def unit(c):
        return scn.Sequential(scn.ConcatTable(scn.Identity(), scn.Sequential(
            scn.ReLU(), scn.SubmanifoldConvolution(3, c, c, 3, True), scn.ReLU(), scn.SubmanifoldConvolution(3, c, c, 3, True))),
            scn.AddTable())
    inner = scn.Sequential(unit(16), unit(16))
    level = scn.Sequential(scn.Sequential(scn.SubmanifoldConvolution(3, 7, 16, 1, True)), inner).to(gpu)
    inp = scn.InputLayer(3, size, mode=4)

    def run(stages):
        M.TREE_STAGES = stages
        M.STAGE_STATS.update(enc=0, dec=0, layerwise_units=0)
        try:
            for p in level.parameters():
                p.grad = None
            x = inp((coords, feats.to(gpu), 1))
            y = level(x).features
            y.backward(torch.ones_like(y))
            return y.detach().clone(), [p.grad.clone() for p in level.parameters()], M.STAGE_STATS["enc"]
        finally:
            M.TREE_STAGES = True

    def both(expect_stage=1):
        a, b = run(True), run(False)
        assert a[2] == expect_stage and b[2] == 0
        assert torch.equal(a[0], b[0])
        assert len(a[1]) == len(b[1]) and all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
        return a[0]
    y0 = both()
    twin = copy.deepcopy(level)                                    # a tree that has run: plans are not copied, ctypes and all
    assert "_stages" in level.__dict__ and "_stages" not in twin.__dict__
    inner.append(unit(16).to(gpu))                                 # three units now
    y1 = both()
    assert not torch.equal(y0, y1)
    inner[1] = unit(16).to(gpu)                                    # __setitem__: another unit in the middle
    y2 = both()
    assert not torch.equal(y1, y2)
    del inner[0]                                                   # two units again, different ones
    y3 = both()
    assert not torch.equal(y2, y3)
    inner[0][0][1][1].bias = None                                  # no bias: plans need one -> the level runs layer by layer
    both(expect_stage=0)
    M.TREE_STAGES = True
    x = inp((coords, feats.to(gpu), 1))
    assert torch.equal(twin(x).features, y0)                       # the copy still is the tree as it was when copied


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_forward_only_is_bit_equal_to_the_training_forward_and_allocates_less(gpu, dtype):
    """evaluation (`eval_model`, ndsis/training/training.py:244-304; `SparseMaskPredictor`,
    model.py:826-882) runs the forward under torch.no_grad().  There the executor plans no backward workspace -- a stage's
    slabs share storage along its op list -- and packs no backward-data weight image; the results are the training forward's
    bit for bit (backbone output, every encoder output, mask logits) and the forward's peak allocation is smaller."""
    from sparse_rcnn_amd import executor as EX
    from sparse_rcnn_amd.trainstep import SceneStep
    job = SceneStep("cfg3", gpu, dtype=dtype, prefetch=False, seed=3, target=30_000, grid=(256, 256, 128), n_boxes=16, n_buckets=0)
    m = job.model

    def train_forward():
        fin = job.feats.detach().requires_grad_()
        out = m.backbone(job.coords, fin, job.size, job.batch_size)
        inter = [t.features.detach().clone() for t in m.backbone.unet.interims]
        logits, _ = m.mask((job.coords, fin, job.size, job.batch_size, job.splits), out, job.boxes)
        return out, logits, inter
    out, logits, inter = train_forward()                   # (first call: plans, caches)
    torch.cuda.synchronize()
    assert EX.LAST_FORWARD["lean"] is False
    ref = (out.features.detach().clone(), logits.detach().clone())
    del out, logits
    o2, l2 = job.forward_only()
    assert EX.LAST_FORWARD["lean"] is True
    assert not o2.features.requires_grad and not l2.requires_grad
    assert torch.equal(o2.features, ref[0]) and torch.equal(l2, ref[1])
    for a, t in zip(inter, m.backbone.unet.interims):
        assert torch.equal(a, t.features)
    del o2, l2
    # ---- the PLANNED saving, from the stage plans themselves (a bound that follows from the plan, not from
    # a measurement): a stage's training workspace keeps one slab per intermediate (head, and per residual unit y1 and x + y),
    # its forward-only workspace three slots per level (`_lean_slots`: a slab's slot is free again after its last reader).
    # Every stage of a forward_only pass reports both totals for ITS shapes.
    plan = {"lean": 0, "plain": 0, "stages": 0}
    orig_lean = EX._lean_layout

    def counting_lean(stage, ns):
        r = orig_lean(stage, ns)
        plan["lean"] += r[1]
        plan["plain"] += EX._layout(stage.fwd_specs, ns)[1]
        plan["stages"] += 1
        return r
    EX._lean_layout = counting_lean
    try:
        keep = job.forward_only()
        torch.cuda.synchronize()
        del keep
    finally:
        EX._lean_layout = orig_lean
    assert plan["stages"] >= 7 + 7                        # backbone 4 + 3 levels, mask branch input stage + 3 + 3
    planned_saving = plan["plain"] - plan["lean"]
    # three slots against at least four slabs per residual level (five or more where a decoder level joins and projects first)
    assert plan["lean"] <= 0.75 * plan["plain"], plan
    # ---- the MEASURED peak allocation of one forward, both ways (graph alive at the end of the training forward, as before a
    # backward): the allocator must see at least 90 % of the planned slab saving (both peaks include the index structures a
    # forward builds -- scene pyramid, ROI batch --, which evaluation needs as well; bf16 storage also drops the backward-data
    # weight images from the pack, on top of the slabs)
    import gc

    def peak(fn):
        # nothing of an earlier forward may be alive when the measurement starts: the network keeps its last encoder outputs
        # (`.interims`, and through them a training forward's whole graph and workspaces) until the next forward replaces them
        object.__setattr__(m.backbone.unet, "interims", [])
        object.__setattr__(m.mask.output_conv_layer, "interims", [])
        job.out = job.logits = job.fin = None
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        keep = fn()
        torch.cuda.synchronize()
        p = torch.cuda.max_memory_allocated() - base
        del keep
        return p
    p_train = peak(train_forward)
    p_eval = peak(job.forward_only)
    assert p_train - p_eval >= 0.9 * planned_saving, (p_train, p_eval, plan)
    # a frozen network under grad mode (no operand requires a gradient) takes the forward-only plan as well
    for p in m.parameters():
        p.requires_grad_(False)
    out3 = m.backbone(job.coords, job.feats, job.size, job.batch_size)
    assert EX.LAST_FORWARD["lean"] is True and torch.equal(out3.features, ref[0])


def test_deferred_tensors_compute_layer_by_layer_when_nobody_fuses_them(gpu):
    """A pending Deconvolution / NetworkInNetwork whose consumer is NOT a run of residual units (the features are read
    directly; a JoinTable is materialised) gives the layer-by-layer result, and a Deconvolution to a level no Convolution of
    this forward built still raises the reference's error."""
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import modules as M
    from sparse_rcnn_amd.tensor import DeferredTensor
    from sparse_rcnn_amd.unet import Backbone
    coords, feats, size, bs, _ = _scene(5_000, (64, 64, 32), seed=9)
    torch.manual_seed(1)
    net = Backbone(7, (16, 32)).to(gpu)
    u = net.unet
    d = u.decoder[0]
    outs = []
    for stages in (True, False):
        M.TREE_STAGES = stages
        try:
            x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), 1))
            e0 = u.encoder[0](x)
            e1 = u.encoder[1](e0)
            up = d["up"](e1)
            assert isinstance(up, DeferredTensor) == stages
            nin = d["nin"](d["join"]([up, e0]))
            assert isinstance(nin, DeferredTensor) == stages and (not stages or (up.pending and nin.pending))
            outs.append((nin.features.clone(), up.features.clone(), d["join"]([d["up"](e1), e0]).features.clone()))
            assert not stages or not (up.pending or nin.pending)
        finally:
            M.TREE_STAGES = True
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), 1))
    lone = scn.Sequential(scn.ReLU(), scn.Deconvolution(3, 16, 16, (2, 2, 2), (2, 2, 2), True)).to(gpu)
    half = scn.SparseConvNetTensor(features=u.encoder[0](x).features, metadata=x.metadata,
                                   spatial_size=torch.as_tensor([32, 32, 16]))
    with pytest.raises(scn.ScnError, match="no cached Convolution rulebook"):
        lone(half).features


class _unfused:
    """One leaf operator call per layer: every fusion of the host layer off (the fused paths equal this one bit for bit)."""

    def __enter__(self):
        from sparse_rcnn_amd import modules as M
        from sparse_rcnn_amd.unet import SparseUNet
        self.saved = (M.FUSE_RELU, M.FUSE_ADD, M.FUSE_BLOCK, M.TREE_STAGES, SparseUNet.EXEC)
        M.FUSE_RELU = M.FUSE_ADD = M.FUSE_BLOCK = M.TREE_STAGES = False
        SparseUNet.EXEC = False

    def __exit__(self, *exc):
        from sparse_rcnn_amd import modules as M
        from sparse_rcnn_amd.unet import SparseUNet
        M.FUSE_RELU, M.FUSE_ADD, M.FUSE_BLOCK, M.TREE_STAGES, SparseUNet.EXEC = self.saved
        return False


def _tapes():
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    return json.load(open(os.path.join(here, "golden", "optape_reference_forward.json")))


@pytest.mark.parametrize("name", ["cfg2_32_256", "ref_32_112"])
def test_dropin_backbone_issues_the_reference_forwards_op_tape_on_the_gpu(gpu, name):
    """VERDICT r3 item 5b with REAL kernels: the leaf-operator tape of unet.DropinBackbone on the GPU (layer signatures,
    operand wiring, spatial sizes, row counts from the device index build, channels) equals the tape the reference's
    FeatureExtractor.forward produced on this package in the build container (tests/golden/optape_reference_forward.json;
    tests/optape.py), and the unfused forward that issued it gives the bits of the fused one."""
    import optape
    from sparse_rcnn_amd.synthetic import make_batch
    from sparse_rcnn_amd.unet import Backbone, DropinBackbone
    T = _tapes()
    sc = T["scene"]
    coords, feats, size, bs, _ = make_batch(sc["n_samples"], tuple(sc["grid"]), sc["target"], dup=sc["dup"], seed=sc["seed"])
    ch = [32, 64, 128, 256] if name == "cfg2_32_256" else [32, 48, 64, 80, 96, 112]
    torch.manual_seed(2)
    net = DropinBackbone(Backbone(7, ch).to(gpu))
    with _unfused(), optape.record() as tape:
        out = net(coords, feats.to(gpu), size, bs)
    ref = T["feature_extractor_" + name]
    assert len(tape.entries) == len(ref)
    for i, (a, b) in enumerate(zip(tape.entries, ref)):
        assert a == b, (i, a, b)
    fused = net(coords, feats.to(gpu), size, bs)
    assert torch.equal(fused.features, out.features)


def test_mask_branch_issues_the_reference_mask_networks_op_tape_on_the_gpu(gpu):
    """The same for maskhead.MaskBranch against the tape of the reference's SparseMaskNetwork.forward (model.py:758-782, eval
    mode: the given boxes are the selected ones; scannet_config/run.py:741-810): SubM 1^3 + units on the scene -> OutputLayer
    -> [crop] -> InputLayer(mode 4, batch_size = boxes, spatial size + 32) -> internal U-Net -> OutputLayer, every leaf call
    with its wiring, sizes and row counts -- the crop's row counts come from this package's device crop (scn_roi.hip), the
    tape's from the reference's own roi_cut.  Differences by design, normalised: the 23-channel level is physically padded to
    24 columns here."""
    import optape
    from sparse_rcnn_amd.maskhead import MaskBranch
    from sparse_rcnn_amd.synthetic import make_batch, make_boxes
    from sparse_rcnn_amd.unet import Backbone
    T = _tapes()
    sc = T["scene"]
    coords, feats, size, bs, splits = make_batch(sc["n_samples"], tuple(sc["grid"]), sc["target"], dup=sc["dup"], seed=sc["seed"])
    boxes = make_boxes(coords, sc["n_boxes"], seed=sc["box_seed"])
    torch.manual_seed(2)
    bb = Backbone(7, (32, 64, 128, 256)).to(gpu)
    mb = MaskBranch(32, 7).to(gpu)
    fd = feats.to(gpu)
    scene = (coords.to(gpu), fd, size, bs, splits)
    with _unfused():
        fmap = bb(coords.to(gpu), fd, size, bs)
        with optape.record() as tape:
            logits, selection = mb(scene, fmap, boxes)
    ref = T["mask_network"]
    got = optape.normalised(tape.entries, pad={24: 23}, join_pad={48: 46}, at_size=[96, 96, 64])
    assert list(logits.shape) == T["mask_logits_shape"] and len(got) == len(ref) == 89
    for i, (a, b) in enumerate(zip(got, ref)):
        assert a == b, (i, a, b)
    fused, _ = mb(scene, bb(coords.to(gpu), fd, size, bs), boxes)
    assert torch.equal(fused, logits)


@pytest.mark.parametrize("target", [40_000, 2_500])
@pytest.mark.parametrize("c", [64, 128, 256])
def test_streaming_bf16_tile_kernel_agrees_with_the_k_split_kernel(gpu, c, target):
    """k_conv_tbs (round 4: offsets outside, the 64-column weight slice of one offset streamed through LDS, full K per
    workgroup -- no K split, no partial tiles) against k_conv_tb (SCN_TB_STREAM=0: weights of all offsets resident, K split over
    workgroups, in-launch combine) on the same packed weight image: SubM 3^3 forward with input ReLU + residual, backward-data
    with ReLU mask + residual-last, and the 2^3 child table of a strided convolution, at a tile count that takes the 8-wave
    workgroups and one that takes the 4-wave ones.  The two kernels sum the same products in a different order (K-chunks inside
    an offset vs offsets inside a K-chunk), so they agree to fp32 rounding BEFORE the one bf16 rounding of the result: apart by
    at most one bf16 ulp of the value plus the fp32 noise of a 1728-term sum (1e-5 of the output scale: what a result that
    cancels to nearly zero can move by), on a small fraction of the elements -- and each is bitwise reproducible."""
    import os
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import functional as F, _lib as L
    coords, feats, size, bs, _ = _scene(target, (256, 256, 128), seed=4)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), 1))
    rb = x.metadata.subm_rulebook(size, 3)
    sb = x.metadata.strided_rulebook(size)
    g = torch.Generator().manual_seed(c)
    bf = torch.bfloat16
    X = torch.randn(rb.n, c, generator=g).to(gpu).to(bf)
    W = (torch.randn(27, c, c, generator=g) * (2.0 / (27 * c)) ** 0.5).to(gpu)
    b = torch.randn(c, generator=g).to(gpu)
    R = torch.randn(rb.n, c, generator=g).to(gpu).to(bf)
    Mk = torch.randn(rb.n, c, generator=g).to(gpu).to(bf)
    W8 = (torch.randn(8, c, c, generator=g) * (2.0 / (8 * c)) ** 0.5).to(gpu)
    back = L.F_W_TRANSPOSED | L.F_OFF_REVERSE

    def run():
        out = [F.conv_rules(X, rb.tiles, rb.n, W, b, c, L.F_RELU_IN, residual=R),
               F.conv_rules(X, rb.tiles, rb.n, W, None, c, back | L.F_RESIDUAL_LAST, residual=R, relu_mask=Mk),
               F.conv_rules(X, sb.tiles, sb.n_coarse, W8, b, c, 0),
               F.conv_rules(X[:sb.n_coarse].contiguous(), sb.tiles, sb.n_coarse, W8, None, c, L.F_W_TRANSPOSED)]
        torch.cuda.synchronize()
        return out
    _SW["SCN_TB_STREAM"] = "1"                  # every eligible layer (default: only Cin = 64, 3^3)
    try:
        a, a2 = run(), run()
        _SW["SCN_TB_STREAM"] = "0"
        ref = run()
    finally:
        del _SW["SCN_TB_STREAM"]
    for k, (u, v, r) in enumerate(zip(a, a2, ref)):
        assert u.dtype == bf and torch.equal(u, v), f"op {k}: not reproducible"
        uf, rf = u.float(), r.float()
        diff = (uf - rf).abs()
        bound = rf.abs() * (2.0 ** -7 * 1.001) + 1e-5 * float(rf.abs().max())    # one bf16 ulp <= 2^-7 of the value
        assert bool((diff <= bound).all()), (k, float((diff / bound).max()))
        assert float((diff > 0).float().mean()) < 0.02, (k, float((diff > 0).float().mean()))
        assert float(rf.abs().max()) > 0


def test_executor_side_stream_leaves_give_the_same_bits(gpu):
    """scn_exec_run_streams with the parameter-gradient ops on a second stream (SCN_EXEC_SIDE=1: each leaf behind an
    event of the main stream, joined before the call returns) -- shipped, measured slower, off by default -- gives the bits of
    the one-stream pass."""
    from sparse_rcnn_amd import executor as EX
    from sparse_rcnn_amd.unet import Backbone
    coords, feats, size, bs, _ = _scene(12_000, (128, 128, 64), seed=8)
    torch.manual_seed(4)
    net = Backbone(7, (16, 32, 64)).to(gpu)
    res = []
    for side in (False, True):
        EX.SIDE_LEAVES = side
        try:
            res.append(_run_backbone(net, coords, feats, size, 13, gpu, True))
        finally:
            EX.SIDE_LEAVES = False
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])
    for x, y in zip(a[3], b[3]):
        assert torch.equal(x, y)


def test_executor_times_its_tile_launches_for_a_sampling_timer(gpu):
    """ADVICE r3 / bench.py's `roofline`: with a profiling.KernelTimer that samples only the dominant tile kernel, a forward +
    backward stays on the step executor (the production path) and the executor's C calls bracket their tile-convolution
    launches with HIP events themselves: 62 records for the cfg-2 plan (28 SubM 3^3 x 2 + 3 strided forward + 3 deconvolution
    backward-data), positive times, the algorithmic FLOPs of the layer-by-layer timer; and the same bits as an untimed step."""
    from sparse_rcnn_amd import profiling
    from sparse_rcnn_amd.unet import Backbone
    coords, feats, size, bs, _ = _scene(30_000, (256, 256, 128), seed=7)
    torch.manual_seed(3)
    net = Backbone(7, (32, 64, 128, 256)).to(gpu)
    ref = _run_backbone(net, coords, feats, size, 11, gpu, True)
    sums = {}
    for use_exec in (True, False):
        t = profiling.KernelTimer(every=1, names={"k_conv_ts", "k_conv_tb"})
        t.begin_step()
        profiling.TIMER = t
        try:
            got = _run_backbone(net, coords, feats, size, 11, gpu, use_exec)
        finally:
            profiling.TIMER = None
        torch.cuda.synchronize()
        sums[use_exec] = t.summary()["k_conv_ts"]
        assert len(t.exec_records) == (62 if use_exec else 0) and len(t.records) == (0 if use_exec else 62)
        assert torch.equal(got[0], ref[0]) and all(torch.equal(x, y) for x, y in zip(got[3], ref[3]))
    a, b = sums[True], sums[False]
    assert a["launches"] == b["launches"] == 62 and a["ms"] > 0 and b["ms"] > 0
    assert abs(a["flops"] - b["flops"]) <= 1e-9 * b["flops"] and abs(a["bytes"] - b["bytes"]) <= 1e-9 * b["bytes"]


@pytest.mark.parametrize("target", [40_000, 2_500])
@pytest.mark.parametrize("c", [64, 128])
def test_streaming_fp32_tile_kernel_agrees_with_the_k_split_kernel(gpu, c, target):
    """k_conv_tss (round 4: offsets outside, the 64-column fp32 weight slice of one offset streamed through LDS, full K per
    workgroup, no K split) against k_conv_ts (SCN_TS_STREAM=0) on the same operands: SubM 3^3 forward with input ReLU +
    residual, backward-data (transposed, offset-reversed weights) with ReLU mask + residual-last, the 2^3 child table forward
    and transposed -- at a tile count that takes the 16-wave workgroups and one that takes the 8-wave ones.  Same products,
    different summation order (K-chunks inside an offset vs offsets inside a K-chunk): agreement to fp32 rounding, 2e-6 of
    the output scale; each kernel bitwise reproducible; both within 1e-5 of an fp64 evaluation of the forward."""
    import os
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import functional as F, _lib as L
    coords, feats, size, bs, _ = _scene(target, (256, 256, 128), seed=4)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), 1))
    rb = x.metadata.subm_rulebook(size, 3)
    sb = x.metadata.strided_rulebook(size)
    g = torch.Generator().manual_seed(c)
    X = torch.randn(rb.n, c, generator=g).to(gpu)
    W = (torch.randn(27, c, c, generator=g) * (2.0 / (27 * c)) ** 0.5).to(gpu)
    b = torch.randn(c, generator=g).to(gpu)
    R = torch.randn(rb.n, c, generator=g).to(gpu)
    Mk = torch.randn(rb.n, c, generator=g).to(gpu)
    W8 = (torch.randn(8, c, c, generator=g) * (2.0 / (8 * c)) ** 0.5).to(gpu)
    back = L.F_W_TRANSPOSED | L.F_OFF_REVERSE
    paths = (__import__("ctypes").c_int64 * 4)()

    def run():
        out = [F.conv_rules(X, rb.tiles, rb.n, W, b, c, L.F_RELU_IN, residual=R),
               F.conv_rules(X, rb.tiles, rb.n, W, None, c, back | L.F_RESIDUAL_LAST, residual=R, relu_mask=Mk),
               F.conv_rules(X, sb.tiles, sb.n_coarse, W8, b, c, 0),
               F.conv_rules(X[:sb.n_coarse].contiguous(), sb.tiles, sb.n_coarse, W8, None, c, L.F_W_TRANSPOSED)]
        torch.cuda.synchronize()
        return out
    _SW["SCN_TS_STREAM"] = "1"                  # (opt-in: measured slower than k_conv_ts)
    try:
        a, a2 = run(), run()
    finally:
        del _SW["SCN_TS_STREAM"]
    ref = run()
    for k, (u, v, r) in enumerate(zip(a, a2, ref)):
        assert torch.equal(u, v), f"op {k}: not reproducible"
        scale = float(r.abs().max())
        assert scale > 0 and float((u - r).abs().max()) <= 2e-6 * scale, (k, float((u - r).abs().max()) / scale)
    assert not torch.equal(a[0], ref[0])                              # (the streaming kernel really ran: another summation order)
    # fp64 forward of op 0 on a sample of rows through the neighbour table
    rows = torch.arange(0, rb.n, max(1, rb.n // 512), device=gpu)
    tab = rb.table[:, rows].long()                                   # [27, rows]
    acc = b.double().unsqueeze(0).repeat(len(rows), 1) + R[rows].double()
    for o in range(27):
        ok = tab[o] >= 0
        acc[ok] += torch.relu(X[tab[o][ok]]).double() @ W[o].double()
    for y in (a[0], ref[0]):
        assert float((y[rows].double() - acc).abs().max()) <= 1e-5 * float(acc.abs().max())


def test_bf16_elementwise_forms_match_torch(gpu):
    """scn_cast_* / scn_add_bf16 / scn_gather_rows_bf16 / scn_segment_sum_bf16 / pooling / SparseToDense in bf16 storage against
    torch on the same bits (casts and gathers bit-exact; sums within one bf16 rounding of the fp64 sum)."""
    from sparse_rcnn_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(5000, 24, generator=g) * 3).to(gpu)
    xb = torch.empty(x.shape, dtype=torch.bfloat16, device=gpu)
    L.check(lib.scn_cast_f32_to_bf16(x.data_ptr(), x.numel(), xb.data_ptr(), L.stream()))
    assert torch.equal(xb, x.to(torch.bfloat16))
    xf = torch.empty_like(x)
    L.check(lib.scn_cast_bf16_to_f32(xb.data_ptr(), xb.numel(), xf.data_ptr(), L.stream()))
    assert torch.equal(xf, xb.float())
    odd = x[:777, :7].contiguous()                                            # unaligned tail path
    ob = torch.empty(odd.shape, dtype=torch.bfloat16, device=gpu)
    L.check(lib.scn_cast_f32_to_bf16(odd.data_ptr(), odd.numel(), ob.data_ptr(), L.stream()))
    assert torch.equal(ob, odd.to(torch.bfloat16))
    yb = (torch.randn(5000, 24, generator=g)).to(gpu).to(torch.bfloat16)
    zb = torch.empty_like(xb)
    L.check(lib.scn_add_bf16(xb.data_ptr(), yb.data_ptr(), xb.numel(), zb.data_ptr(), L.stream()))
    assert torch.equal(zb, xb + yb)
    rows = torch.randint(0, 5000, (12345,), generator=g).to(torch.int32).to(gpu)
    out = torch.empty((len(rows), 24), dtype=torch.bfloat16, device=gpu)
    L.check(lib.scn_gather_rows_bf16(xb.data_ptr(), rows.data_ptr(), len(rows), 24, out.data_ptr(), L.stream()))
    assert torch.equal(out, xb[rows.long()])
    dy = torch.randn(len(rows), 24, generator=g).to(gpu).to(torch.bfloat16)
    dx = torch.empty((5000, 24), dtype=torch.bfloat16, device=gpu)
    acc = torch.empty((5000, 24), dtype=torch.float64, device=gpu)
    L.check(lib.scn_segment_sum_bf16(dy.data_ptr(), rows.data_ptr(), len(rows), 5000, 24, dx.data_ptr(), acc.data_ptr(), L.stream()))
    exp = torch.zeros(5000, 24, dtype=torch.float64, device=gpu).index_add_(0, rows.long(), dy.double())
    assert torch.equal(dx, exp.float().to(torch.bfloat16))


@pytest.mark.parametrize("cin,cout", [(48, 48), (80, 80), (112, 112), (16, 16), (48, 32), (32, 48), (96, 80), (24, 40), (144, 48)])
def test_conv_tiles_tail_slices_reproduce_the_padded_kernel_bit_for_bit(gpu, cin, cout):
    """k_conv_ts TAIL variants (round 3: the reference's own plan 32-48-64-80-96-112, scannet_config/run.py:539-549): a
    K-chunk with <= 16 valid channels / a column chunk with <= 16 valid columns skips the MFMAs of its dead half, and the
    slices get workgroups in proportion to their cost.  The skipped MFMAs multiplied zeros: forward, backward-data (with
    ReLU mask and residual) and the in-launch K reduction give the bits of the padded kernel (SCN_TS_NO_TAIL=1)."""
    import os
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import functional as F, _lib as L
    coords, feats, size, bs, _ = _scene(20_000, (256, 256, 128), seed=4)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), 1))
    rb = x.metadata.subm_rulebook(size, 3)
    sb = x.metadata.strided_rulebook(size)
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    X = torch.randn(rb.n, cin, generator=g).to(gpu)
    W = (torch.randn(27, cin, cout, generator=g) * 0.1).to(gpu)
    b = torch.randn(cout, generator=g).to(gpu)
    R = torch.randn(rb.n, cout, generator=g).to(gpu)
    Mk = torch.randn(rb.n, cout, generator=g).to(gpu)
    W8 = (torch.randn(8, cin, cout, generator=g) * 0.1).to(gpu)

    def run():
        out = [F.conv_rules(X, rb.tiles, rb.n, W, b, cout, L.F_RELU_IN, residual=R),
               F.conv_rules(X, rb.tiles, rb.n, W.transpose(1, 2).contiguous(), None, cout, L.F_W_TRANSPOSED | L.F_OFF_REVERSE | L.F_RESIDUAL_LAST,
                            residual=R, relu_mask=Mk),
               F.conv_rules(X, sb.tiles, sb.n_coarse, W8, b, cout, 0)]
        torch.cuda.synchronize()
        return out
    import ctypes
    paths = (ctypes.c_int64 * 4)()
    a = run()
    _SW["SCN_TS_NO_TAIL"] = "1"
    try:
        ref = run()
    finally:
        del _SW["SCN_TS_NO_TAIL"]
    for u, v in zip(a, ref):
        assert torch.equal(u, v)
    assert float(a[0].abs().max()) > 0


@pytest.mark.parametrize("cin,cout", [(48, 48), (80, 80), (112, 112), (48, 80), (96, 48)])
def test_wgrad_whole_fragment_edge_blocks_keep_vector_loads_and_the_bits(gpu, cin, cout):
    """k_wgrad_direct on the reference's 48 / 80 / 112-channel layers: the channel blocks are 64 wide, so the last block is an
    EDGE block -- but a lane's 4 channels are all inside the layer or all outside, so its row pieces stay single vector
    loads (round 3; element loads before).  Same arithmetic: the bits of the element-wise path (SCN_WD_NO_EVEC=1), single
    and paired problems, with the bias gradient."""
    import os
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import functional as F, _lib as L
    coords, feats, size, bs, _ = _scene(20_000, (256, 256, 128), seed=6)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), 1))
    rb = x.metadata.subm_rulebook(size, 3)
    r = rb.rules
    g = torch.Generator().manual_seed(cin * 100 + cout)
    X = torch.randn(rb.n, cin, generator=g).to(gpu)
    dY = torch.randn(rb.n, cout, generator=g).to(gpu)

    def run():
        a = F.wgrad_bias_rules(X, dY, r.in_rows, r.out_rows, r.prefix_host, 27, 1 << 13, L.F_RELU_IN)
        out = [a[0], a[1]]
        if cin == cout:
            X2, dY2 = X * 0.5 + 1.0, dY * 2.0
            b = F.wgrad_bias_rules2(X, dY, X2, dY2, r.in_rows, r.out_rows, r.prefix_host, 27, 1 << 13, L.F_RELU_IN)
            out += [b[0], b[1]]
        torch.cuda.synchronize()
        return out
    a = run()
    _SW["SCN_WD_NO_EVEC"] = "1"
    try:
        ref = run()
    finally:
        del _SW["SCN_WD_NO_EVEC"]
    for u, v in zip(a, ref):
        assert torch.equal(u, v)
    assert float(a[0].abs().max()) > 0


@pytest.mark.parametrize("c", [48, 96])
def test_wgrad_48_wide_blocks_match_the_padded_blocks(gpu, c):
    """k_wgrad_direct<3,3> (round 3): the reference's 48- and 96-channel layers are whole multiples of a 48-wide wave block
    (three-dword row pieces) instead of padded 64- / 128-wide ones (1.78x the MFMAs).  Same rules, same per-rule products;
    the unit plan differs, so the fp32 unit sums associate differently: <= 2e-6 of the scale against the padded blocks
    (SCN_WD_NO_T3=1), single and paired problems, weight and bias gradients, the identity list (NetworkInNetwork)."""
    import os
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import functional as F, _lib as L
    coords, feats, size, bs, _ = _scene(20_000, (256, 256, 128), seed=6)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), 1))
    rb = x.metadata.subm_rulebook(size, 3)
    r = rb.rules
    g = torch.Generator().manual_seed(c)
    X = torch.randn(rb.n, c, generator=g).to(gpu)
    dY = torch.randn(rb.n, c, generator=g).to(gpu)
    X2, dY2 = (X * 0.5 + 1.0).contiguous(), (dY * 2.0).contiguous()

    def run():
        a = F.wgrad_bias_rules(X, dY, r.in_rows, r.out_rows, r.prefix_host, 27, 1 << 13, L.F_RELU_IN)
        b = F.wgrad_bias_rules2(X, dY, X2, dY2, r.in_rows, r.out_rows, r.prefix_host, 27, 1 << 13, L.F_RELU_IN)
        n = F.wgrad_bias_rules(X, dY, None, None, F._identity_prefix(rb.n), 1, 1)
        torch.cuda.synchronize()
        return [a[0], a[1], b[0], b[1], n[0], n[1]]
    a = run()
    a2 = run()
    _SW["SCN_WD_NO_T3"] = "1"
    try:
        ref = run()
    finally:
        del _SW["SCN_WD_NO_T3"]
    for u, v, w in zip(a, ref, a2):
        assert torch.equal(u, w)                                              # reproducible
        scale = float(v.abs().max())
        assert float((u - v).abs().max()) <= 2e-6 * scale, (float((u - v).abs().max()), scale)


@pytest.mark.parametrize("bf16", [False, True])
def test_deferred_weight_gradient_sums_equal_the_immediate_ones_bit_for_bit(gpu, bf16):
    """scn_wgrad_defer_begin / _flush (round 3d): eight weight-gradient launches of different channel shapes record their
    unit sums, the flush adds them in two batched launches (six + two).  Per layer the sum's arithmetic and order are the
    single launch's: same bits as eight immediate calls (dW and db), fp32 and bf16-stored operands."""
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import functional as F, _lib as L
    coords, feats, size, bs, _ = _scene(6_000, (128, 128, 64), seed=8)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), 1))
    rb = x.metadata.subm_rulebook(size, 3)
    r = rb.rules
    lib = L.lib()
    shapes = [(32, 32), (64, 64), (32, 64), (128, 128), (64, 32), (48, 48), (16, 32), (256, 64)]
    g = torch.Generator().manual_seed(5)
    ops = []
    for cin, cout in shapes:
        X = torch.randn(rb.n, cin, generator=g).to(gpu)
        dY = torch.randn(rb.n, cout, generator=g).to(gpu)
        if bf16:
            X, dY = X.bfloat16(), dY.bfloat16()
        ops.append((X, dY))
    ref = [F.wgrad_bias_rules(X, dY, r.in_rows, r.out_rows, r.prefix_host, 27, 1 << 13, L.F_RELU_IN) for X, dY in ops]
    torch.cuda.synchronize()
    entry = lib.scn_wgrad_bias_rules_bf16 if bf16 else lib.scn_wgrad_bias_rules
    outs, keep = [], []
    L.check(lib.scn_wgrad_defer_begin())
    for X, dY in ops:
        cin, cout = X.shape[1], dY.shape[1]
        scratch = torch.empty(lib.scn_wgrad_scratch_bytes(cin, cout, r.prefix_host, 27), dtype=torch.uint8, device=gpu)
        dW = torch.full((27, cin, cout), float("nan"), device=gpu)
        db = torch.full((cout,), float("nan"), device=gpu)
        L.check(entry(L.ptr(X), cin, L.ptr(dY), cout, L.ptr(r.in_rows), L.ptr(r.out_rows), r.prefix_host, 27, L.ptr(dW),
                      L.ptr(db), 1 << 13, L.ptr(scratch), L.F_RELU_IN, L.stream()))
        outs.append((dW, db)); keep.append(scratch)
    torch.cuda.synchronize()
    assert all(bool(torch.isnan(dW).all()) for dW, _ in outs)            # nothing is summed before the flush
    L.check(lib.scn_wgrad_defer_flush(L.stream()))
    torch.cuda.synchronize()
    for (dW, db), (rW, rb_) in zip(outs, ref):
        assert torch.equal(dW, rW) and torch.equal(db, rb_)
    assert float(outs[0][0].abs().max()) > 0


def test_xcd_local_tile_order_is_a_valid_second_order_and_changes_no_bit(gpu):
    """scn_tiles_build_x (round 3d): behind the LPT order sits the list of the tiles by (spatial bin of the first row, cost
    descending) and its nine bin starts; scn_conv_tiles_bf16 with SCN_F_TILE_ORDER_X hands bin x to the workgroups of XCD x.
    The list is a permutation of the tiles, bins are row ranges, costs descend inside a bin; the first order is unchanged;
    the convolution gives the SAME bits with either hand-out (forward and backward-data, with and without the K split)."""
    import os
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import functional as F, _lib as L, metadata as MD
    coords, feats, size, bs, _ = _scene(40_000, (256, 256, 128), seed=9)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), 1))
    md = x.metadata
    rb_plain = md.subm_rulebook(size, 3)
    t0 = rb_plain.tiles
    assert not t0.has_x
    tx = MD.build_tiles(rb_plain.table, 27, rb_plain.n, with_x=True)
    assert tx.has_x
    nt = (rb_plain.n + 15) // 16
    assert torch.equal(tx.tile_order, t0.tile_order) and torch.equal(tx.perm, t0.perm) and torch.equal(tx.tstab, t0.tstab)
    full = torch.as_strided(tx.tile_order, (2 * nt + 9,), (1,)).cpu().numpy()          # the buffer behind the first order
    order_x, bs9 = full[nt:2 * nt], full[2 * nt:2 * nt + 9]
    assert sorted(order_x.tolist()) == list(range(nt))
    assert bs9[0] == 0 and bs9[8] == nt and all(bs9[i] <= bs9[i + 1] for i in range(8))
    first_row = tx.perm.cpu().numpy().reshape(nt, 16)[:, 0]
    cost = np.array([bin(int(v)).count("1") for v in tx.tile_mask.cpu().numpy().view(np.uint32)])
    for b in range(8):
        ids = order_x[bs9[b]:bs9[b + 1]]
        if len(ids) == 0:
            continue
        assert all((np.maximum(first_row[ids], 0).astype(np.int64) * 8) // rb_plain.n == b)
        c = cost[ids]
        assert all(c[i] >= c[i + 1] for i in range(len(c) - 1))
    sizes = np.diff(bs9)
    assert sizes.min() > 0.5 * nt / 8 and sizes.max() < 1.5 * nt / 8, sizes           # row ranges hold similar tile counts
    for cin, cout in ((32, 32), (64, 64), (32, 64), (128, 128)):
        X = torch.randn(rb_plain.n, cin, device=gpu).bfloat16()
        W = torch.randn(27, cin, cout, device=gpu) * 0.1
        for fl in (0, L.F_W_TRANSPOSED | L.F_OFF_REVERSE if cin == cout else 0):
            Wc = W if not fl else W.transpose(1, 2).contiguous()
            a = F.conv_rules_bf16(X, tx, rb_plain.n, Wc, None, cout, fl | L.F_RELU_IN)
            b = F.conv_rules_bf16(X, t0, rb_plain.n, Wc, None, cout, fl | L.F_RELU_IN)
            _SW["SCN_TB_NO_XORDER"] = "1"
            try:
                c = F.conv_rules_bf16(X, tx, rb_plain.n, Wc, None, cout, fl | L.F_RELU_IN)
            finally:
                del _SW["SCN_TB_NO_XORDER"]
            assert torch.equal(a, b) and torch.equal(a, c)
            assert float(a.float().abs().max()) > 0


def test_padded_parameters_in_one_launch_equal_the_tensor_by_tensor_pads(gpu):
    """functional.padded_params (round 4): the zero-padded parameter copies of the mask network's 23 -> 24-column level (Deconvolution
    out, NetworkInNetwork over two padded parts, four SubM 3^3, the Convolution reading the padded slab) from ONE
    scn_pad_params_many launch and their gradients sliced back by one more -- against torch.nn.functional.pad tensor by tensor
    (SCN_PAD_MANY=0): logits, every parameter gradient and both input gradients bit for bit, for the configured network and for
    the raw-only variant (7 -> 8 columns under a 16-wide decoder level: unequal joined parts), and in bf16 storage, where the
    padded layers' weight images are packed from the LOGICAL weights by the network's one pack launch.  Plus the entry point alone on a
    two-segment job against torch indexing."""
    import ctypes as C
    from sparse_rcnn_amd import functional as F, _lib as L, tensor as T
    from sparse_rcnn_amd.maskhead import MaskBranch
    from sparse_rcnn_amd.synthetic import make_batch, make_boxes
    import sparse_rcnn_amd as scn
    # -- the entry point alone: [2][5][3] -> [2][9][4], rows 0-1 -> 0-1 and 2-4 -> 5-7
    src = torch.randn(2, 5, 3, device=gpu)
    dst = torch.full((2, 9, 4), 7.0, device=gpu)
    desc = (C.c_int32 * 11)(2, 5, 3, 9, 4, 0, 2, 0, 2, 3, 5)
    a, b = (C.c_void_p * 1)(src.data_ptr()), (C.c_void_p * 1)(dst.data_ptr())
    L.check(L.lib().scn_pad_params_many(1, a, b, desc, 0, L.stream()))
    want = torch.zeros(2, 9, 4, device=gpu)
    want[:, 0:2, :3], want[:, 5:8, :3] = src[:, 0:2], src[:, 2:5]
    assert torch.equal(dst, want)
    g = torch.randn(2, 9, 4, device=gpu)
    back = torch.full((2, 5, 3), 7.0, device=gpu)
    a, b = (C.c_void_p * 1)(g.data_ptr()), (C.c_void_p * 1)(back.data_ptr())
    L.check(L.lib().scn_pad_params_many(1, a, b, desc, 1, L.stream()))
    assert torch.equal(back, torch.cat([g[:, 0:2, :3], g[:, 5:8, :3]], 1))
    a[0] = None
    L.check(L.lib().scn_pad_params_many(1, a, b, desc, 1, L.stream()))
    assert not back.any()
    # -- the mask network
    coords, feats, size, bs, splits = make_batch(2, (128, 128, 64), 8_000, seed=5)
    boxes = make_boxes(coords, 12, seed=6)
    for flags in (dict(), dict(use_unet_features=False), dict(bf16_blocks="all")):
        torch.manual_seed(4)
        mb = MaskBranch(32, 7, **flags).to(gpu)
        x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), bs))
        X = torch.randn(x.features.shape[0], 32, generator=torch.Generator().manual_seed(1)).to(gpu)
        gm = None
        res = []
        for many in (True, False):
            F.PAD_MANY = many
            object.__setattr__(mb.output_conv_layer, "_pad_plan", None)
            try:
                for rep in range(2):                       # (first forward of a network records, the second uses the plan)
                    mb.zero_grad()
                    Xd, fd = X.clone().requires_grad_(), feats.to(gpu).requires_grad_()
                    fmap = T.SparseConvNetTensor(features=Xd, metadata=x.metadata, spatial_size=x.spatial_size)
                    logits, _ = mb((coords.to(gpu), fd, size, bs, splits), fmap if mb.use_unet_features else None, boxes)
                    gm = torch.randn(logits.shape, generator=torch.Generator().manual_seed(2)).to(gpu) if gm is None else gm
                    logits.backward(gm)
            finally:
                F.PAD_MANY = True
            plan = mb.output_conv_layer.__dict__.get("_pad_plan")
            assert bool(plan) == many and (not many or plan.n == (13 if mb.use_unet_features else 2))
            res.append([logits.detach()] + [p.grad.clone() for p in mb.parameters()] + [fd.grad.clone()]
                       + ([Xd.grad.clone()] if mb.use_unet_features else []))
        assert not F.PADDED
        for u, v in zip(*res):
            assert torch.equal(u, v)


@pytest.mark.parametrize("bf16", [False, True])
def test_roi_cut_preparation_modes_give_the_inline_result(gpu, bf16):
    """MaskBranch.PREFETCH_ROI_INDEX: "0" (the crop's selection and the ROI batch's index build inline, on the caller's stream),
    "split" (round 4: count pass queued on the index stream first, the scene-level input stage queued on the caller's stream,
    THEN the two host waits), "stream" and "thread" (everything on the index stream, from the caller's / a helper thread) and
    MaskBranch.prepare_cut + forward(prepared_cut=) -- the same logits, selection and gradients bit for bit (bf16: the prepared
    index carries the XCD-local tile order the inline path asks for, so the same kernels run)."""
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import tensor as T
    from sparse_rcnn_amd.maskhead import MaskBranch
    from sparse_rcnn_amd.synthetic import make_batch, make_boxes
    coords, feats, size, bs, splits = make_batch(2, (128, 128, 64), 10_000, seed=8)
    boxes = make_boxes(coords, 10, seed=9)
    torch.manual_seed(5)
    mb = MaskBranch(32, 7, bf16_blocks="all" if bf16 else False).to(gpu)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), bs))
    X = torch.randn(x.features.shape[0], 32, generator=torch.Generator().manual_seed(1)).to(gpu)
    gm = None
    res = {}
    saved = MaskBranch.PREFETCH_ROI_INDEX
    try:
        for how in ("0", "split", "stream", "thread", "prepared"):
            MaskBranch.PREFETCH_ROI_INDEX = "0" if how == "prepared" else how
            for rep in range(2):
                mb.zero_grad()
                Xd, fd = X.clone().requires_grad_(), feats.to(gpu).requires_grad_()
                fmap = T.SparseConvNetTensor(features=Xd, metadata=x.metadata, spatial_size=x.spatial_size)
                scene = (coords.to(gpu), fd, size, bs, splits)
                cut = mb.prepare_cut(coords.to(gpu), size, boxes) if how == "prepared" else None
                logits, sel = mb(scene, fmap, boxes, prepared_cut=cut)
                gm = torch.randn(logits.shape, generator=torch.Generator().manual_seed(2)).to(gpu) if gm is None else gm
                logits.backward(gm)
            torch.cuda.synchronize()
            res[how] = [logits.detach(), sel[0].src_row, sel[0].box_of, fd.grad, Xd.grad] + [p.grad.clone() for p in mb.parameters()]
    finally:
        MaskBranch.PREFETCH_ROI_INDEX = saved
    for how, r in res.items():
        if bf16 and how in ("stream", "thread", "prepared"):
            # (these hand over an index without the XCD-local order: another hand-out order of the same tiles -- same sums per
            # tile, so still the same bits)
            pass
        for u, v in zip(res["0"], r):
            assert torch.equal(u, v), how


@pytest.mark.parametrize("c,target", [(80, 2_900), (96, 700), (112, 180), (64, 1_500), (128, 400), (32, 300)])
def test_four_waves_per_tile_loop_agrees_with_the_plain_loop(gpu, c, target):
    """scn_conv_ts_small.inc (round 4): where a level has fewer (tile, slice) pairs than half the chip's waves -- the coarse
    levels of the reference's plan, 80 / 96 / 112 channels on ~2 900 / 700 / 180 rows -- four waves share a tile (offsets dealt
    by rank among the mask's set bits, all of a wave's gathers in flight at once, partial tiles added in LDS in ascending share
    order).  Against the plain loop (SCN_TS_SPLIT=0) on the same operands: SubM 3^3 forward with input ReLU + residual,
    backward-data with ReLU mask + residual-last, the 2^3 child table forward and transposed, the K reduction inside the launch
    and as a second launch (functional.FUSED_K = False; bit-equal to the in-launch form in BOTH loops), TAIL slices (80, 112) and the
    padded kernel (SCN_TS_NO_TAIL=1: bit-equal in both loops).  The two loops associate an element's sum differently: 2e-6 of
    the output scale; each bitwise reproducible; both within 1e-5 of an fp64 forward; and the split loop really ran."""
    import os
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import functional as F, _lib as L
    coords, feats, size, bs, _ = _scene(target, (96, 96, 48), seed=7)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), 1))
    rb = x.metadata.subm_rulebook(size, 3)
    sb = x.metadata.strided_rulebook(size)
    g = torch.Generator().manual_seed(c)
    X = torch.randn(rb.n, c, generator=g).to(gpu)
    W = (torch.randn(27, c, c, generator=g) * (2.0 / (27 * c)) ** 0.5).to(gpu)
    b = torch.randn(c, generator=g).to(gpu)
    R = torch.randn(rb.n, c, generator=g).to(gpu)
    Mk = torch.randn(rb.n, c, generator=g).to(gpu)
    W8 = (torch.randn(8, c, c, generator=g) * (2.0 / (8 * c)) ** 0.5).to(gpu)
    back = L.F_W_TRANSPOSED | L.F_OFF_REVERSE
    lib = L.lib()

    def run(fused=True):
        F.FUSED_K = fused                                                 # False: the K-chunk slabs are added by a second launch
        try:
            out = [F.conv_rules(X, rb.tiles, rb.n, W, b, c, L.F_RELU_IN, residual=R),
                   F.conv_rules(X, rb.tiles, rb.n, W, None, c, back | L.F_RESIDUAL_LAST, residual=R, relu_mask=Mk),
                   F.conv_rules(X, sb.tiles, sb.n_coarse, W8, b, c, 0),
                   F.conv_rules(X[:sb.n_coarse].contiguous(), sb.tiles, sb.n_coarse, W8, None, c, L.F_W_TRANSPOSED)]
        finally:
            F.FUSED_K = True
        torch.cuda.synchronize()
        return out
    lib.scn_conv_tiles_split_count(1)
    a, a2 = run(), run()
    assert lib.scn_conv_tiles_split_count(1) == 8                         # every launch took the four-waves-per-tile loop
    two = run(False)                                                      # (cin <= 32: nothing to add, same launch)
    _SW["SCN_TS_NO_TAIL"] = "1"
    try:
        padded = run()
    finally:
        del _SW["SCN_TS_NO_TAIL"]
    lib.scn_conv_tiles_split_count(1)
    _SW["SCN_TS_SPLIT"] = "0"
    try:
        ref = run()
        ref_two = run(False)
    finally:
        del _SW["SCN_TS_SPLIT"]
    assert lib.scn_conv_tiles_split_count(1) == 0
    for k in range(4):
        assert torch.equal(a[k], a2[k]) and torch.equal(a[k], two[k]) and torch.equal(a[k], padded[k]), k
        assert torch.equal(ref[k], ref_two[k]), k
        scale = float(ref[k].abs().max())
        assert scale > 0 and float((a[k] - ref[k]).abs().max()) <= 2e-6 * scale, (k, float((a[k] - ref[k]).abs().max()) / scale)
    rows = torch.arange(0, rb.n, max(1, rb.n // 256), device=gpu)
    tab = rb.table[:, rows].long()
    acc = b.double().unsqueeze(0).repeat(len(rows), 1) + R[rows].double()
    for o in range(27):
        ok = tab[o] >= 0
        acc[ok] += torch.relu(X[tab[o][ok]]).double() @ W[o].double()
    for y in (a[0], ref[0]):
        assert float((y[rows].double() - acc).abs().max()) <= 1e-5 * float(acc.abs().max())
