"""-m gpu: ORACLE parity at BASELINE sizes.  The CPU oracle finishes a full 150k-voxel forward +
backward in a few seconds, so it IS the checker here: index structures bit-exact, features and every gradient against
oracle/scn_oracle.py on the same seeded inputs.  Every test records the errors it achieved (max abs, max abs relative
to the oracle's max, relative L2) in gpurun_out/parity_r4.jsonl; the committed copy is profiles/r4_parity_errors.jsonl.

Round 4: the sign masks are recorded by the step executor from the slabs it keeps for backward, so the cfg-2 / cfg-3 tests
below run the path bench.py TIMES (one autograd node per level, scn_exec_run, deferred weight-gradient sums); the
reference-plan test keeps the layer-by-layer module path (SparseUNet.EXEC = False) as the second variant.

Bounds: the north star's "features within 1e-4 fp32" is read relative to the output scale (max |oracle|); gradients are
bounded by relative L2 per tensor with the ReLU masks of the HIP forward prescribed to the oracle."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import scn_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, "gpurun_out", "parity_r4.jsonl")

FEAT_TOL = 1e-4          # BASELINE.json north_star, relative to max |oracle|
# Gradients of the 60-layer nets are ill-conditioned in fp32 when each side makes its OWN ReLU decisions: the fp32 oracle
# itself sits 2e-4 .. 2.5e-3 (relative L2) from the same oracle evaluated in fp64 -- a ReLU input within rounding of zero
# flips a row's whole contribution, and which side flips is chance (round 2 bounded gradients by 3e-3 with an fp64 arbiter;
# profiles/r2_parity_errors.jsonl).  Round 3 removes the cause instead of widening the bound: the HIP forward records the
# sign mask of every slab a ReLU is applied to and the oracle is evaluated with THOSE masks (O.FrozenReLU), so both sides
# differentiate the same piecewise-linear function -- see FROZEN_L2_F32 / FROZEN_L2_BF16 below.  Forward features are
# always compared against the oracle's own ReLU decisions.


def _err(a, b):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    d = (a - b).abs()
    scale = max(b.abs().max().item(), 1e-30)
    nz = b.abs() > 1e-3 * scale
    p99 = d.kthvalue(max(1, int(0.99 * d.numel()))).values.item() / scale
    return dict(max_abs=d.max().item(), scale=scale, rel_to_scale=d.max().item() / scale, p99_to_scale=p99,
                max_elem_rel=(d[nz] / b[nz].abs()).max().item() if nz.any() else 0.0,
                rel_l2=(d.norm() / b.norm().clamp_min(1e-30)).item())


def _record(test, what, e, bound):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    row = dict(test=test, what=what, bound=bound, **e)
    with open(LOG, "a") as f:
        f.write(json.dumps(row) + "\n")
    print(f"[parity] {test} {what}: max|d|={e['max_abs']:.3e} rel_to_scale={e['rel_to_scale']:.3e} "
          f"rel_l2={e['rel_l2']:.3e} (scale {e['scale']:.3g})")


@pytest.fixture(scope="module")
def scene150k():
    from sparse_rcnn_amd.synthetic import make_batch
    coords, feats, size, bs, splits = make_batch(1, (512, 512, 256), 150_000, dup=1.15, seed=1)
    return coords, feats, size, bs, splits, O.OracleScene(coords.numpy())


# ------------------------------------------------------------------------------------------------ cfg 2
def test_cfg2_rulebooks_bit_exact_at_150k(gpu, scene150k):
    import sparse_rcnn_amd as scn
    coords, feats, size, bs, splits, scene = scene150k
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), 1))
    md = x.metadata
    md.build_pyramid(size, 4, 3)
    assert np.array_equal(md.item_row.cpu().numpy(), scene.prow)
    assert np.array_equal(md.row_count.cpu().numpy(), scene.counts)
    sz = tuple(int(s) for s in size)
    for level in range(4):
        rules = scene.subm_rules(level, 3)
        if level < 3:
            srules = scene.strided_rules(level)
        assert np.array_equal(md.grid(sz).coords.cpu().numpy().astype(np.int64), scene.level_coords[level]), level
        rb = md.subm_rulebook(sz, 3)
        pairs, prefix = O.rules_concat(rules)
        assert rb.rules.prefix_list() == prefix.tolist(), level
        assert np.array_equal(rb.rules.in_rows.cpu().numpy(), pairs[:, 0]), level
        assert np.array_equal(rb.rules.out_rows.cpu().numpy(), pairs[:, 1]), level
        if level < 3:
            sb = md.strided_rulebook(sz)
            spairs, sprefix = O.rules_concat(srules)
            assert sb.rules.prefix_list() == sprefix.tolist(), level
            assert np.array_equal(sb.rules.in_rows.cpu().numpy(), spairs[:, 0]), level
            assert np.array_equal(sb.rules.out_rows.cpu().numpy(), spairs[:, 1]), level
            sz = tuple(v // 2 for v in sz)
    # the one-call native build gives the same structures
    md2 = scn.Metadata(3).build_native(size, coords, 1, 4, 4, 3)
    sz = tuple(int(s) for s in size)
    for level in range(4):
        assert torch.equal(md2.subm[(sz, 3)].table, md.subm[(sz, 3)].table)
        assert torch.equal(md2.subm[(sz, 3)].rules.in_rows, md.subm[(sz, 3)].rules.in_rows)
        sz = tuple(v // 2 for v in sz)


def test_cfg2_full_unet_forward_and_every_gradient_vs_oracle_at_150k(gpu, scene150k):
    """BASELINE configs[1] exactly: Backbone(7, (32, 64, 128, 256)) on the seed-1 150k-voxel scene, forward, every
    parameter gradient and the input-feature gradient against O.unet_forward.  Round 3: the oracle takes the ReLU sign
    masks the HIP forward recorded (O.FrozenReLU), so a ReLU input within rounding of zero no longer flips a row's whole
    contribution on one side only: every gradient within 2e-5 relative L2 (round 2, own masks: 3e-3 with an fp64 arbiter)."""
    from sparse_rcnn_amd.unet import Backbone
    coords, feats, size, bs, splits, scene = scene150k
    ch = (32, 64, 128, 256)
    params = O.init_unet_params(7, list(ch), seed=0)
    net = Backbone(7, ch).to(gpu)
    net.unet.load_oracle_params(params)
    fin = feats.to(gpu).requires_grad_()
    with _record_relu_masks() as masks:
        out = net(coords, fin, size, 1)
    assert _n_stage_nodes(out.features) == 7 and len(masks) == 31       # the step executor ran (the path bench.py times)
    gy = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(5))
    out.features.backward(gy.to(gpu))
    torch.cuda.synchronize()
    po = {k: v.clone().requires_grad_() for k, v in params.items()}
    fo = feats.clone().requires_grad_()
    # forward against the oracle's OWN relu decisions (the north star's feature bound), gradients with the shared masks
    with torch.no_grad():
        plain = O.unet_forward(scene, feats, params, list(ch))
    exp = O.unet_forward(scene, fo, po, list(ch), relu=O.FrozenReLU(masks))
    exp.backward(gy)
    name = "cfg2_full_unet_150k"
    e = _err(out.features, plain)
    _record(name, "forward features", e, FEAT_TOL)
    assert out.features.shape[0] == 150_000 and e["rel_to_scale"] <= FEAT_TOL, e
    for k, p in list(net.unet.named_oracle_params().items()) + [("input features", fin)]:
        _check_grad_frozen(name, "grad " + k, p.grad, fo.grad if p is fin else po[k].grad.view_as(p), FROZEN_L2_F32)


def _n_stage_nodes(t):
    """Number of step-executor nodes (executor.StageFunction) in the autograd graph below tensor t."""
    n, seen, todo = 0, set(), [t.grad_fn]
    while todo:
        f = todo.pop()
        if f is None or f in seen:
            continue
        seen.add(f)
        n += type(f).__name__.startswith("StageFunction")
        todo += [g for g, _ in f.next_functions]
    return n


class _record_relu_masks:
    """with _record_relu_masks() as masks: ... -- the sign mask (x > 0) of every slab a ReLU is applied to in the HIP
    FORWARD passes run inside the block, in call order (sparse_rcnn_amd.functional.RELU_RECORD).  Fed to the oracle as
    O.FrozenReLU(masks), both sides differentiate the same piecewise-linear function."""

    def __enter__(self):
        from sparse_rcnn_amd import functional as F
        F.RELU_RECORD = []
        return F.RELU_RECORD

    def __exit__(self, *exc):
        from sparse_rcnn_amd import functional as F
        F.RELU_RECORD = None
        return False


# With the ReLU masks of the HIP forward prescribed to the oracle, what is left between the two gradients is summation
# order (fp32) and, in bf16 storage, the rounding of the stored gradient slabs (the oracle differentiates the rounded
# forward straight-through in fp32): bounds on the relative L2 of EVERY gradient tensor.
# Achieved in round 3 (profiles/r3_parity_errors.jsonl; this round's values: profiles/r4_parity_errors.jsonl): fp32 4e-6 worst
# over 157 tensors of the cfg-3 step (median 8e-7) -- round 2's bound without shared masks was 3e-3; bf16 storage 6-9e-3 for
# the backbone (worst 8.9e-3, deepest encoder level), 1.2-1.45e-2 for the deepest levels of the mask branch's internal U-Net
# (`m:unet.enc3.res1.conv1.weight` 1.32e-2 / 1.43e-2, `m:unet.dec2.res1.conv1.weight` 1.29e-2 / 1.45e-2 on the 60k chain / the
# 12k two-rank scenes: the longest chain of bf16-rounded gradient slabs) -- round 2: 1.5 x a measured 4-16 % yardstick, 25 % for
# the mask branch.  THE BF16 BOUND IS 2e-2: round 3 first tried 1e-2 and was red on the mask branch's tensors (first failures
# `m:in.res0.conv0.bias` 1.19e-2 and `m:in.bias` 1.23e-2, gpurun_out/r3c_tests.log); the backbone alone stays below 1e-2.
FROZEN_L2_F32 = 2e-5
FROZEN_L2_BF16 = 2e-2


RPN_L2_F32 = 2e-5          # dense RPN parameters (the stack runs on this package's tile kernels, its ReLU masks frozen into the oracle)


def _check_grad_frozen(name, what, got, ref, bound):
    e = _err(got, ref)
    _record(name, what, e, f"frozen ReLU masks: rel_l2 <= {bound}")
    assert bool(torch.isfinite(got).all()) and e["rel_l2"] <= bound, (what, e)


def test_cfg2_bf16_storage_vs_oracle_with_the_same_roundings_at_150k(gpu, scene150k):
    """The bf16 STORAGE mode (BASELINE configs 3-5) of the full backbone at 150k voxels against the ORACLE evaluated with
    the same storage roundings -- every stored slab after the first layer rounded to bf16, the tile-kernel layers' weights
    rounded to bf16 (the NetworkInNetwork over the JoinTable is one two-source launch: no rounding between its parts) -- not
    against the HIP path's own fp32 run.  Forward: what is left between the two is the summation order inside a layer and,
    through it, single bf16 roundings that fall the other way (0.01 % of a layer's outputs, tools/diag_bf16_layers.py); a
    ReLU network amplifies those until they sit at the noise floor of bf16 storage itself, so the mirrored roundings are
    sharp where the path is short (first encoder level <= 5e-4 relative L2) and the final features are held to 2^-6 of the
    scale / 1e-2 relative L2.
    Gradients (round 3): the ORACLE takes the ReLU sign masks the HIP forward recorded (O.FrozenReLU) -- round 2 measured
    that mask flips, not gradient rounding, were what separated the two (two ulp-nudged realisations of the oracle itself
    sat 4-16 % apart), and bounded each tensor by 1.5 x that distance.  With the masks shared, every parameter gradient and
    the input gradient is held to FROZEN_L2_BF16 = 2e-2 relative L2 (the HIP path stores its gradient slabs in bf16, the
    oracle differentiates the rounded forward straight-through in fp32; achieved: worst 8.9e-3 here)."""
    from sparse_rcnn_amd.unet import Backbone
    coords, feats, size, bs, splits, scene = scene150k
    ch = (32, 64, 128, 256)
    params = O.init_unet_params(7, list(ch), seed=0)
    net = Backbone(7, ch, bf16_blocks="all").to(gpu)
    net.unet.load_oracle_params(params)
    fin = feats.to(gpu).requires_grad_()
    with _record_relu_masks() as masks:
        out = net(coords, fin, size, 1)
    assert out.features.dtype == torch.float32 and len(masks) == 31          # 2 per residual unit + 1 per deconvolution
    assert _n_stage_nodes(out.features) == 7                                 # through the step executor
    gy = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(5))
    out.features.backward(gy.to(gpu))
    torch.cuda.synchronize()
    po = {k: v.clone().requires_grad_() for k, v in params.items()}
    fo = feats.clone().requires_grad_()
    rec = []
    fr = O.FrozenReLU(masks)
    exp = O.unet_forward(scene, fo, po, list(ch), storage=O.bf16_storage, tile_weights=O.bf16_storage, record=rec, relu=fr)
    assert fr.k == len(masks)
    exp.backward(gy)
    name = "cfg2_bf16_storage_150k"
    e = _err(net.unet.interims[0].features.float(), dict(rec)["enc0"])
    _record(name, "encoder level 0 output vs oracle with the same roundings", e, "rel_l2 <= 5e-4")
    assert e["rel_l2"] <= 5e-4, e
    e = _err(out.features, exp)
    _record(name, "forward features vs oracle with the same roundings", e, "rel_to_scale <= 2^-6, rel_l2 <= 1e-2")
    assert e["rel_to_scale"] <= 2.0 ** -6 and e["rel_l2"] <= 1e-2, e
    for k, p in list(net.unet.named_oracle_params().items()) + [("input features", fin)]:
        ref = fo.grad if p is fin else po[k].grad.view_as(p)
        _check_grad_frozen(name, "grad " + k, p.grad, ref, FROZEN_L2_BF16)


def test_reference_plan_32_112_full_unet_vs_oracle_at_150k(gpu, scene150k):
    """The network the reference actually trains (scannet_config/run.py:539-549,587-591: six levels 32-48-64-80-96-112) on
    the 150k-voxel scene: rulebooks of the two extra levels bit-exact, forward within 1e-4 of the scale, every one of the
    120 parameter gradients and the input gradient within FROZEN_L2_F32 = 2e-5 relative L2 of the oracle (ReLU masks of the
    HIP forward prescribed to the oracle).  The 48 / 80 / 112-channel layers run the TAIL variants of k_conv_ts (dead K
    halves and column blocks skipped).  This test keeps the LAYER-BY-LAYER module path (SparseUNet.EXEC = False: one autograd
    node per residual unit / layer, immediate weight-gradient sums) -- the way the reference's own module tree drives the
    surface; the cfg-2 / cfg-3 tests run the step executor."""
    from sparse_rcnn_amd.unet import Backbone, SparseUNet
    from sparse_rcnn_amd.trainstep import REF_PLAN
    coords, feats, size, bs, splits, scene = scene150k
    ch = REF_PLAN
    params = O.init_unet_params(7, list(ch), seed=4)
    net = Backbone(7, ch).to(gpu)
    net.unet.load_oracle_params(params)
    fin = feats.to(gpu).requires_grad_()
    SparseUNet.EXEC = False
    try:
        with _record_relu_masks() as masks:
            out = net(coords, fin, size, 1)
        assert len(masks) == 6 * 4 + 5 * 5 and _n_stage_nodes(out.features) == 0
        gy = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(6))
        out.features.backward(gy.to(gpu))
        torch.cuda.synchronize()
    finally:
        SparseUNet.EXEC = True
    po = {k: v.clone().requires_grad_() for k, v in params.items()}
    fo = feats.clone().requires_grad_()
    fr = O.FrozenReLU(masks)
    exp = O.unet_forward(scene, fo, po, list(ch), relu=fr)
    assert fr.k == len(masks)
    exp.backward(gy)
    md = out.metadata
    sz = tuple(int(s) // 16 for s in size)
    for level in (4, 5):                                        # the levels the 4-level benchmark plan never builds
        rules = scene.subm_rules(level, 3)
        rb = md.subm_rulebook(sz, 3)
        pairs, prefix = O.rules_concat(rules)
        assert rb.rules.prefix_list() == prefix.tolist(), level
        assert np.array_equal(rb.rules.in_rows.cpu().numpy(), pairs[:, 0]) and \
            np.array_equal(rb.rules.out_rows.cpu().numpy(), pairs[:, 1]), level
        sz = tuple(v // 2 for v in sz)
    name = "ref_plan_32_112_150k"
    e = _err(out.features, exp)
    _record(name, "forward features", e, FEAT_TOL)
    assert out.features.shape == (150_000, 32) and e["rel_to_scale"] <= FEAT_TOL, e
    named = net.unet.named_oracle_params()
    assert len(named) == 120
    for k, p in list(named.items()) + [("input features", fin)]:
        _check_grad_frozen(name, "grad " + k, p.grad, fo.grad if p is fin else po[k].grad.view_as(p), FROZEN_L2_F32)


def test_dropin_path_equals_backbone_path_at_150k(gpu, scene150k):
    """The layer-by-layer module path (Metadata created inside the forward from HOST coordinates, rulebooks requested
    lazily -- then, from the second forward on, built by one native call on the remembered depth) gives the bits of the
    Backbone path."""
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd.unet import Backbone, DropinBackbone
    coords, feats, size, bs, splits, scene = scene150k
    net = Backbone(7, (32, 64, 128, 256)).to(gpu)
    ref = net(coords, feats.to(gpu), size, 1).features
    drop = DropinBackbone(net)
    scn.Metadata.LEVELS_HINT.pop(tuple(int(s) for s in size), None)
    lazy = drop(coords, feats.to(gpu), size, 1).features                    # first forward: rulebooks one by one
    assert scn.Metadata.LEVELS_HINT[tuple(int(s) for s in size)] == 4
    hinted = drop(coords, feats.to(gpu), size, 1)                           # second: one scn_pyramid_build call
    assert getattr(hinted.metadata, "_workspace", None) is not None
    assert torch.equal(lazy, ref) and torch.equal(hinted.features, ref)
    # the training loop's batches wrapped in scn.index_prefetching: batch i+1's index structures are built on the helper
    # thread while batch i runs, and the InputLayer inside the forward adopts them (same tensor object) -- same bits
    from sparse_rcnn_amd import metadata as MD
    batches = [(coords.clone(), size, 1) for _ in range(3)]
    outs = [drop(c, feats.to(gpu), s_, b) for c, s_, b in scn.index_prefetching(batches, lambda t: t)]
    assert all(torch.equal(o.features, ref) for o in outs)
    assert outs[0].metadata._prepared_for is None                       # the first batch was never announced
    assert outs[1].metadata._prepared_for is not None and outs[2].metadata._prepared_for is not None
    assert not MD._prefetched


# ------------------------------------------------------------------------------------------------ cfg 3
def _oracle_mask_branch(coords_np, raw, bb_feats, mp, boxes_np, assoc, scene, bf16=False, relu=None, use_unet=True,
                        use_raw=True, use_skip=False):
    """SparseMaskNetwork forward (model.py:758-782) on the oracle ops, reference configuration (run.py:741-810);
    use_unet / use_raw / use_skip: the other selectors / the combiner (model.py:597-651).
    bf16: with the storage roundings of the HIP path's bf16 mode (stored slabs -- incl. the per-point slab the crop reads --
    and the tile-kernel layers' weights rounded; InputLayer mean and the Linear stack stay fp32)."""
    relu = torch.relu if relu is None else relu                 # O.FrozenReLU: the masks the HIP forward recorded
    q = O.bf16_storage if bf16 else (lambda t: t)               # stored slabs (gradient passes straight through)
    wq = q                                                      # tile-kernel weights
    kw = dict(storage=O.bf16_storage, tile_weights=O.bf16_storage) if bf16 else {}
    n0 = scene.n(0)
    parts = []
    if use_unet:
        ident = [(np.arange(n0, dtype=np.int32),) * 2]
        x = q(O.conv(q(bb_feats), mp["in.weight"], mp["in.bias"], ident, n0))
        rules = scene.subm_rules(0, 3)
        for u in range(2):
            y = q(O.conv(relu(x), wq(mp[f"in.res{u}.conv0.weight"]), mp[f"in.res{u}.conv0.bias"], rules, n0))
            y = O.conv(relu(y), wq(mp[f"in.res{u}.conv1.weight"]), mp[f"in.res{u}.conv1.bias"], rules, n0)
            x = q(x + y)
        parts.append(x[torch.from_numpy(scene.prow)])                        # OutputLayer
    if use_raw:
        parts.append(q(raw))                                                 # (bf16 storage: the per-point slab is bf16)
    cat = torch.cat(parts, 1)
    c0 = cat.shape[1]
    src, box_of, inside = O.roi_crop(coords_np, boxes_np, assoc)
    if len(src) == 0:                                    # no box caught a point: the branch ends here (model.py:768-770)
        return None, src, box_of, None
    new_coords = np.concatenate([coords_np[src][:, :3], box_of[:, None]], 1)
    rscene = O.OracleScene(new_coords)
    unet_p = {k[5:]: v for k, v in mp.items() if k.startswith("unet.")}
    m = O.unet_forward(rscene, cat[torch.from_numpy(src)], unet_p, [c0, 32, 48, 64], identity_first=True, relu=relu, **kw)
    pts = m[torch.from_numpy(rscene.prow)]                                   # OutputLayer over the ROI batch
    if use_skip:                                                             # SparseFeaturemapCombiner (model.py:639-651)
        pts = torch.cat([pts, raw[torch.from_numpy(src)]], 1)
    h = relu(pts @ mp["lin0.weight"].t() + mp["lin0.bias"])
    return h @ mp["lin1.weight"].t() + mp["lin1.bias"], src, box_of, rscene


def _mask_oracle_params(named, dt=torch.float32, c0=23):
    """MaskBranch.named_oracle_params() tensors (torch or numpy) -> oracle-shaped leaf tensors of dtype dt."""
    shapes = dict(O.unet_param_shapes(c0, [c0, 32, 48, 64], identity_first=True, min_channels=16))
    mo = {}
    for k, p in named.items():
        t = (torch.from_numpy(np.asarray(p)) if not torch.is_tensor(p) else p.detach().cpu()).clone()
        if k.startswith("unet."):
            t = t.view(shapes[k[5:]])
        elif k.endswith("conv0.weight") or k.endswith("conv1.weight"):
            t = t.view(27, 16, 16)
        elif k == "in.weight":
            t = t.view(1, 32, 16)
        mo[k] = t.to(dt).requires_grad_()
    return mo


def _oracle_cfg3_step(coords, feats, boxes, pb, pm, ch, grad_seed, dt=torch.float32, bf16=False, masks=None):
    """One cfg-3 step on the oracle: backbone -> (its OUTPUT feeds the mask branch) OutputLayer -> crop -> mask branch; the
    upstream gradients are drawn as trainstep.SceneStep draws them (one generator: backbone output first, then the logits).
    pb / pm: backbone / mask-branch parameters by oracle name (any float dtype; module or oracle shape).
    -> (out, logits or None, {name: gradient}) with the mask names prefixed 'm:' and the input gradient 'input features'."""
    scene = O.OracleScene(coords.numpy())
    bshapes = dict(O.unet_param_shapes(7, list(ch)))
    po = {k: (torch.from_numpy(np.asarray(v)) if not torch.is_tensor(v) else v.detach().cpu()).clone().view(bshapes[k])
          .to(dt).requires_grad_() for k, v in pb.items()}
    mo = _mask_oracle_params(pm, dt)
    fo = feats.to(dt).clone().requires_grad_()
    kw = dict(storage=O.bf16_storage, tile_weights=O.bf16_storage) if bf16 else {}
    fr = O.FrozenReLU(masks) if masks is not None else None       # the masks of the HIP forward: backbone first, then branch
    out = O.unet_forward(scene, fo, po, list(ch), relu=fr, **kw)
    gen = torch.Generator().manual_seed(grad_seed)
    gy = torch.randn(out.shape, generator=gen)
    boxes_np, cnt, assoc = O.transform_boxes([b.numpy() for b in boxes])
    logits, src, box_of, rscene = _oracle_mask_branch(coords.numpy(), fo, out, mo, boxes_np, assoc, scene, bf16=bf16, relu=fr)
    assert fr is None or fr.k == len(masks) or logits is None, (fr.k, len(masks))
    if logits is None:
        out.backward(gy.to(dt))
    else:
        gm = torch.randn(logits.shape, generator=gen)
        torch.autograd.backward([out, logits], [gy.to(dt), gm.to(dt)])
    grads = {k: v.grad for k, v in po.items()}
    grads.update({"m:" + k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in mo.items()})
    grads["input features"] = fo.grad
    return out, logits, grads, len(src)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_cfg3_end_to_end_backbone_crop_mask_vs_oracle_at_150k(gpu, dtype):
    """BASELINE configs[2] as ONE chain at size: trainstep.SceneStep("cfg3") -- the step bench.py times --
    on the 150k-voxel scene with 64 boxes: Backbone 32-64-128-256 -> its output through SubM1 + units -> OutputLayer -> sparse
    ROI crop -> internal U-Net -> Linear stack, backward from BOTH heads (dY on the backbone output, dM on the logits).  The
    mask branch consumes the backbone's own output (not random features): forward of both heads, every one of the 76 + 80
    parameter gradients and the input-feature gradient against the oracle, which takes the ReLU sign masks the HIP forward
    recorded (O.FrozenReLU: 63 masks) -- fp32 within FROZEN_L2_F32 = 2e-5 relative L2 per tensor, bf16 storage (oracle with
    the same roundings) within FROZEN_L2_BF16 = 2e-2.  Round 4: both legs at 150k voxels (round 3 ran the bf16 leg at 60k),
    both through the step executor (the masks come from the slabs its nodes keep for backward)."""
    from sparse_rcnn_amd import roi
    from sparse_rcnn_amd.trainstep import SceneStep
    bf16 = dtype == "bf16"
    job = SceneStep("cfg3", gpu, dtype=dtype, prefetch=False, seed=1, grad_seed=100, lr=0.0)
    grabbed = []
    hook = job.model.mask.output_roi_cut.register_forward_hook(lambda m, a, out: grabbed.append(out[1]))
    with torch.no_grad():                    # biases away from zero (they are initialised to zero)
        g = torch.Generator().manual_seed(21)
        for p in job.model.parameters():
            if p.dim() == 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    with _record_relu_masks() as masks:
        job.forward_backward()
    torch.cuda.synchronize()
    assert len(masks) == 31 + 32
    assert _n_stage_nodes(job.out.features) == 7 and _n_stage_nodes(job.logits) == 7 + 1 + 6     # backbone; + input stage + 3 + 3
    pb = dict(job.model.backbone.unet.named_oracle_params())
    pm = dict(job.model.mask.named_oracle_params())
    ch = job.channels
    out, logits, grads, n_sel = _oracle_cfg3_step(job.coords_cpu, job.feats_cpu, job.boxes, pb, pm, ch, 100, bf16=bf16,
                                                  masks=masks)
    hook.remove()
    name = "cfg3_end_to_end_" + ("150k_bf16_storage" if bf16 else "150k")
    assert job.out.features.shape[0] == 150_000 and job.logits.shape == logits.shape and n_sel == job.n_roi_rows
    # the sparse ROI crop at size: 64 boxes x ~172k points, the selection bit for bit against the oracle, no [boxes, points] object
    sel, counts = grabbed[0][0], grabbed[0][1]
    boxes_np, cnt, assoc = O.transform_boxes([b.numpy() for b in job.boxes])
    src, box_of, _ = O.roi_crop(job.coords_cpu.numpy(), boxes_np, assoc)
    assert isinstance(sel, roi.RoiSelection) and sel._inside is None and counts == cnt
    assert np.array_equal(sel.src_row.cpu().numpy(), src) and np.array_equal(sel.box_of.cpu().numpy(), box_of)
    assert np.array_equal(sel.new_coords.cpu().numpy(), np.concatenate([job.coords_cpu.numpy()[src][:, :3], box_of[:, None]], 1))
    print(f"[parity] {name}: {n_sel} cropped points")
    e_out, e_log = _err(job.out.features, out), _err(job.logits, logits)
    got = {k: p.grad for k, p in pb.items()}
    got.update({"m:" + k: p.grad for k, p in pm.items()})
    got["input features"] = job.fin.grad
    assert set(got) == set(grads)
    if bf16:
        _record(name, "backbone features vs oracle with the same roundings", e_out, "rel_to_scale <= 2^-6, rel_l2 <= 1e-2")
        _record(name, "mask logits vs oracle with the same roundings", e_log, "rel_to_scale <= 2^-4, rel_l2 <= 4e-2")
        assert e_out["rel_to_scale"] <= 2.0 ** -6 and e_out["rel_l2"] <= 1e-2, e_out
        assert e_log["rel_to_scale"] <= 2.0 ** -4 and e_log["rel_l2"] <= 4e-2, e_log
        for k in grads:
            _check_grad_frozen(name, "grad " + k, got[k], grads[k].view_as(got[k]), FROZEN_L2_BF16)
        return
    _record(name, "backbone features", e_out, FEAT_TOL)
    _record(name, "mask logits", e_log, FEAT_TOL)
    assert e_out["rel_to_scale"] <= FEAT_TOL and e_log["rel_to_scale"] <= FEAT_TOL, (e_out, e_log)
    for k in grads:
        _check_grad_frozen(name, "grad " + k, got[k], grads[k].view_as(got[k]), FROZEN_L2_F32)


def _rpn_chain_vs_oracle(gpu, workload, name):
    """The detection + mask step with the RPN boundary inside (`trainstep.SceneStep(workload)`, the step bench.py times) against
    the oracle.  The boxes do not exist before the forward: backbone -> SparseToDense of the anchor level(s) -> dense dilation
    stack(s) + 1x1 head(s) (on this package's tile kernels: rpn.DenseRpn engine "tiles") -> anchors that leave the scene dropped
    -> RoiSelector (sigmoid, top-1024, decode + clip, one-launch NMS, <= n kept) -> sparse ROI crop with THOSE boxes -> mask
    branch; backward from the backbone output, rpn_bbox, rpn_score and the mask logits (model.py:116-240,
    anchor_network.py:73-124, anchor.py:177-227, proposal_selector.py:23-89).
    Oracle side: SparseToDense of the oracle's encoder slabs, the same dense layers as torch CPU conv3d (their ReLUs frozen to
    the device's sign masks like every other ReLU of the chain), rpn outputs within 1e-4 of their scale; the selection -- made
    on the DEVICE's scores: two evaluations of a large score field differ in their last bits and in top-k ties -- equals the
    oracle's greedy NMS on the device's sorted boxes bit for bit; crop, logits and ALL gradients (backbone + dense RPN + mask
    tensors + the input features) with the ReLU masks of the HIP forward frozen into the oracle."""
    import copy
    from sparse_rcnn_amd import rpn as R
    from sparse_rcnn_amd import trainstep as TS
    from sparse_rcnn_amd.trainstep import SceneStep
    job = SceneStep(workload, gpu, dtype="f32", prefetch=False, seed=1, grad_seed=100, lr=0.0)
    m = job.model
    with torch.no_grad():
        g = torch.Generator().manual_seed(21)
        for n_, p in m.named_parameters():
            if p.dim() == 1 and ".head." not in n_ and not n_.startswith("rpn.head"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    with _record_relu_masks() as masks:
        job.forward_backward()
    torch.cuda.synchronize()
    ch = job.channels
    L_ = len(ch)
    rpns = list(m.rpn.levels) if hasattr(m.rpn, "levels") else [m.rpn]
    n_dense = sum(len(r.stack) // 2 for r in rpns)
    n_enc, n_dec = 4 * L_, 5 * (L_ - 1)             # encoder: levels x two units x two ReLUs; decoder level: ReLU + two units
    assert rpns[0].ENGINE == "tiles" and len(masks) == n_enc + n_dec + n_dense + 32
    # (the RPN's kernels are queued between encoder and decoder -- trainstep.RPN_BEFORE_DECODER -- so its masks follow the encoder's)
    at = n_enc if TS.RPN_BEFORE_DECODER else n_enc + n_dec
    rpn_masks, masks = masks[at:at + n_dense], masks[:at] + masks[at + n_dense:]
    rpn_bbox, rpn_score, anchors, roi_score, roi_bbox, roi_index = job.rpn_out
    B = job.batch_size
    size = [int(v) for v in job.size]
    # anchors: every level's cells x anchors, those inside the scene kept, in level order (tests/test_rpn_cpu.py pins the
    # bookkeeping against the reference's AnchorDescriptionMultiLevel)
    all_anchors = torch.cat([r.anchors_for([v // r.stride for v in size], "cpu") for r in rpns], 0)
    inside = R.inside_indicator(all_anchors, torch.tensor(size, dtype=torch.float32), 0).nonzero().squeeze(1)
    n_anch = rpn_score.shape[1]
    assert rpn_bbox.shape == (B, n_anch, 2, 3) and n_anch == len(inside) and 0 < n_anch < len(all_anchors)
    assert torch.equal(anchors.cpu(), all_anchors[inside])
    post = job.n_boxes
    assert all(1 <= len(b) <= post for b in roi_bbox) and job.n_roi_rows > 0
    print(f"[parity] {name}: {n_anch} of {len(all_anchors)} anchors inside, {[len(b) for b in roi_bbox]} proposals kept, "
          f"{job.n_roi_rows} cropped points, score range {float(rpn_score.min()):.2f} .. {float(rpn_score.max()):.2f}")
    # ---- selection: the device's own decoded + clipped boxes in its own top-k order -> the oracle's greedy NMS
    sc = torch.sigmoid(rpn_score.detach())
    top, idx = torch.topk(sc, 1024, dim=1, sorted=True)
    cpu_top = torch.topk(sc.cpu(), 1024, dim=1, sorted=True)[0]
    assert torch.equal(top.cpu(), cpu_top)                              # (values: ties may pick other indices)
    dec_all = R.decode_boxes(anchors, rpn_bbox.detach(), job._scene_shape())
    boxes = []
    for b in range(B):
        dec = dec_all[b][idx[b]].cpu()
        keep = torch.from_numpy(O.nms(dec.numpy(), 0.5))
        assert torch.equal(roi_index[b], idx[b].cpu()[keep][:post])
        assert torch.equal(roi_bbox[b].cpu(), dec[keep][:post]) and torch.equal(roi_score[b].cpu(), top[b].cpu()[keep][:post])
        kept = roi_bbox[b].detach().cpu()
        boxes.append(kept if job.mask_boxes is None else kept[:job.mask_boxes])
    # ---- the oracle chain
    pb = dict(m.backbone.unet.named_oracle_params())
    pm = dict(m.mask.named_oracle_params())
    scene = O.OracleScene(job.coords_cpu.numpy())
    bshapes = dict(O.unet_param_shapes(7, list(ch)))
    po = {k: v.detach().cpu().clone().view(bshapes[k]).requires_grad_() for k, v in pb.items()}
    mo = _mask_oracle_params(pm)
    fo = job.feats_cpu.clone().requires_grad_()
    fr = O.FrozenReLU(masks)
    inter = []
    out = O.unet_forward(scene, fo, po, list(ch), relu=fr, interims=inter)
    gen = torch.Generator().manual_seed(100)
    gy = torch.randn(out.shape, generator=gen)
    raws, dense_o, mk_at = [], {}, 0
    for li, (r, lvl) in enumerate(zip(rpns, m.rpn_levels)):
        size_l = [v // r.stride for v in size]
        dense = O.sparse_to_dense(inter[lvl], scene.level_coords[lvl], size_l, B)
        got_dense = r.to_dense(m.backbone.unet.interims[lvl]).detach().cpu()
        assert got_dense.shape == dense.shape and torch.equal(got_dense != 0, dense.detach() != 0)       # same cells
        _record(name, f"SparseToDense of the stride-{r.stride} level", _err(got_dense, dense.detach()), FEAT_TOL)
        stack, head = copy.deepcopy(r.stack).cpu(), copy.deepcopy(r.head).cpu()
        h = dense
        for layer in stack:
            if not isinstance(layer, torch.nn.Conv3d):
                continue
            h = layer(h)                                                  # conv3d -> ReLU with the device's sign decisions
            mk = rpn_masks[mk_at].view(B, *size_l, -1).permute(0, 4, 1, 2, 3)      # the slab is the volume channels-last
            mk_at += 1
            assert mk.shape == h.shape
            h = h * mk.to(h.dtype)
        raw = head(h)
        raws.append(raw.view(B, r.n_anchors, 7, -1).permute(0, 3, 1, 2).reshape(B, -1, 7))
        pre = "rpn." if len(rpns) == 1 else f"rpn.levels.{li}."
        dense_o.update(list(stack.named_parameters(prefix=pre + "stack")) + list(head.named_parameters(prefix=pre + "head")))
    assert mk_at == n_dense
    raw = torch.cat(raws, 1)[:, inside]
    ob, os_ = raw[..., :6].reshape(B, -1, 2, 3), raw[..., 6]
    e_b, e_s = _err(rpn_bbox, ob), _err(rpn_score, os_)
    _record(name, "rpn_bbox (dense stack: this library's tile kernels vs torch CPU conv3d)", e_b, 1e-4)
    _record(name, "rpn_score", e_s, 1e-4)
    assert e_b["rel_to_scale"] <= 1e-4 and e_s["rel_to_scale"] <= 1e-4, (e_b, e_s)
    gr = [g.cpu() for g in job._grs[0]]                 # the gradients SceneStep drew for rpn_bbox / rpn_score
    torch.randn(ob.shape, generator=gen), torch.randn(os_.shape, generator=gen)        # (advance the generator as it did)
    boxes_np, cnt, assoc = O.transform_boxes([b.numpy() for b in boxes])
    logits, src, box_of, rscene = _oracle_mask_branch(job.coords_cpu.numpy(), fo, out, mo, boxes_np, assoc, scene, relu=fr)
    assert fr.k == len(masks) and len(src) == job.n_roi_rows and logits.shape == job.logits.shape
    gm = job.upstream_grads(0)[1].cpu()
    assert gm.shape == logits.shape
    torch.autograd.backward([out, ob, os_, logits], [gy, gr[0], gr[1], gm])
    e_out, e_log = _err(job.out.features, out), _err(job.logits, logits)
    _record(name, "backbone features", e_out, FEAT_TOL)
    _record(name, "mask logits", e_log, FEAT_TOL)
    assert e_out["rel_to_scale"] <= FEAT_TOL and e_log["rel_to_scale"] <= FEAT_TOL, (e_out, e_log)
    grads = {k: v.grad for k, v in po.items()}
    grads.update({"m:" + k: v.grad for k, v in mo.items()})
    grads["input features"] = fo.grad
    got = {k: p.grad for k, p in pb.items()}
    got.update({"m:" + k: p.grad for k, p in pm.items()})
    got["input features"] = job.fin.grad
    dense_g = {k: p for k, p in m.named_parameters() if k.startswith("rpn.")}
    assert set(dense_o) == set(dense_g) and len(dense_o) == 2 * (n_dense + len(rpns))
    for k in dense_o:
        grads["d:" + k], got["d:" + k] = dense_o[k].grad, dense_g[k].grad
    assert set(got) == set(grads) and len(grads) == len(pb) + 80 + len(dense_o) + 1
    for k in grads:
        # the dense layers' own gradients: same kernels, same bound
        _check_grad_frozen(name, "grad " + k, got[k], grads[k].view_as(got[k]), RPN_L2_F32 if k.startswith("d:") else FROZEN_L2_F32)


def test_cfg3_rpn_chain_vs_oracle_at_150k(gpu):
    """BASELINE configs[2] with the RPN boundary inside the step (`bench.py --workload cfg3-rpn`) on the
    150k-voxel scene: a stand-in RPN (ONE anchor level, 2 x 32 stack, 64 kept) -- see `_rpn_chain_vs_oracle`; 76 backbone + 6
    dense RPN + 80 mask gradient tensors + the input features."""
    _rpn_chain_vs_oracle(gpu, "cfg3-rpn", "cfg3_rpn_chain_150k")


def test_ref_crop_rpn_chain_vs_oracle_at_size(gpu):
    """the REFERENCE's detection shape as far as this path reaches (`bench.py --workload ref-crop-rpn`):
    plan 32-48-64-80-96-112 on its training batch (12 crops of 128 x 128 x 64, run.py:364,485-488), SparseToDense of BOTH
    anchor levels (64 ch at stride 4, 80 ch at stride 8), a 5 x 128 and a 5 x 256 dilation stack (run.py:525-536,609), 3 + 11
    anchors per cell, top-1024 / NMS 0.5 / 256 kept (run.py:847-853), the 24 best per sample -> crop -> mask branch: see
    `_rpn_chain_vs_oracle`; 120 backbone + 24 dense RPN + 80 mask gradient tensors + the input features."""
    _rpn_chain_vs_oracle(gpu, "ref-crop-rpn", "ref_crop_rpn_chain")


@pytest.mark.parametrize("variant", ["unet_only", "both_skip", "raw_skip"])
def test_mask_branch_variants_vs_oracle(gpu, variant):
    """VERDICT r3 missing 4: the other feature-map selectors and the combiner of the reference's SparseMaskNetwork
    (model.py:597-651) as MaskBranch switches -- `unet_only` SparseFeaturemapSelector (crop of the 16-channel per-point output),
    `raw_only` SparseFeaturemapSelectorRaw (crop of the 7 raw channels: no input_conv_layer, the internal U-Net's level 0 runs on
    a 7->8 padded slab and comes up 16 wide, unet_params['min_channels']), `*_skip` SparseFeaturemapCombiner (cropped raw
    features joined in front of the Linear stack; the raw-only selector without it is the `raw_skip` case minus the join and is
    covered on the CPU side by its fixture).  Two 12k-voxel samples, 16 boxes each, fp32: logits against the oracle within FEAT_TOL
    of its scale; every parameter gradient and the gradients of the backbone features / raw features against the oracle with
    the HIP forward's ReLU masks, FROZEN_L2_F32."""
    from sparse_rcnn_amd import tensor as T
    from sparse_rcnn_amd.maskhead import MaskBranch
    from sparse_rcnn_amd.synthetic import make_batch, make_boxes
    import sparse_rcnn_amd as scn
    flags = dict(unet_only=dict(use_raw_features=False), raw_only=dict(use_unet_features=False),
                 both_skip=dict(use_skip_features=True), raw_skip=dict(use_unet_features=False, use_skip_features=True))[variant]
    use_unet, use_raw = flags.get("use_unet_features", True), flags.get("use_raw_features", True)
    use_skip = flags.get("use_skip_features", False)
    coords, feats, size, bs, splits = make_batch(2, (160, 160, 96), 12_000, seed=11)
    boxes = make_boxes(coords, 16, seed=5)
    scene = O.OracleScene(coords.numpy())
    torch.manual_seed(3)
    mb = MaskBranch(32, 7, **flags).to(gpu)
    with torch.no_grad():
        g = torch.Generator().manual_seed(22)
        for p in mb.parameters():
            if p.dim() == 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), bs))
    X = torch.randn(scene.n(0), 32, generator=torch.Generator().manual_seed(9))
    Xd, fd = X.to(gpu).requires_grad_(), feats.to(gpu).requires_grad_()
    fmap = T.SparseConvNetTensor(features=Xd, metadata=x.metadata, spatial_size=x.spatial_size) if use_unet else None
    with _record_relu_masks() as masks:
        logits, selection = mb((coords.to(gpu), fd, size, bs, splits), fmap, boxes)
    gm = torch.randn(logits.shape, generator=torch.Generator().manual_seed(10))
    logits.backward(gm.to(gpu))
    torch.cuda.synchronize()
    pm = dict(mb.named_oracle_params())
    c0 = (16 if use_unet else 0) + (7 if use_raw else 0)
    mo = _mask_oracle_params(pm, c0=c0)
    Xo, fo = X.clone().requires_grad_(), feats.clone().requires_grad_()
    boxes_np, cnt, assoc = O.transform_boxes([b.numpy() for b in boxes])
    fr = O.FrozenReLU(masks)
    ref, src, box_of, rscene = _oracle_mask_branch(coords.numpy(), fo, Xo, mo, boxes_np, assoc, scene, relu=fr,
                                                   use_unet=use_unet, use_raw=use_raw, use_skip=use_skip)
    assert fr.k == len(masks) and len(src) > 2_000 and logits.shape == ref.shape
    own, _, _, _ = _oracle_mask_branch(coords.numpy(), fo.detach(), Xo.detach(), {k: v.detach() for k, v in mo.items()},
                                       boxes_np, assoc, scene, use_unet=use_unet, use_raw=use_raw, use_skip=use_skip)
    name = "mask_branch_" + variant
    e = _err(logits, own)                                  # forward: against the oracle's OWN ReLU decisions
    _record(name, "mask logits", e, FEAT_TOL)
    assert e["rel_to_scale"] <= FEAT_TOL, e
    ref.backward(gm)
    for k, p in pm.items():
        _check_grad_frozen(name, "grad " + k, p.grad, mo[k].grad.view_as(p.grad), FROZEN_L2_F32)
    if use_raw or use_skip:
        _check_grad_frozen(name, "grad raw features", fd.grad, fo.grad, FROZEN_L2_F32)
    else:
        assert fd.grad is None
    if use_unet:
        _check_grad_frozen(name, "grad backbone features", Xd.grad, Xo.grad, FROZEN_L2_F32)


def test_tensor_to_tensor_roi_cut_vs_oracle_at_size(gpu, scene150k):
    """TensorToTensorFeatureExtractorCombiner (roi_select_sparse.py:99-122): the active sites of a SparseConvNetTensor
    are cut (mode-0 re-voxelisation); rows, coordinates and features against the oracle crop of the voxel list."""
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import roi
    from sparse_rcnn_amd.synthetic import make_boxes
    coords, feats, size, bs, splits, scene = scene150k
    bbox_batch = make_boxes(coords, 64, seed=4)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.to(gpu), 1))
    cut = roi.SparseRoiCut(roi.TensorToTensorFeatureExtractorCombiner(), dense_inside=False)
    out, (sel, cnt, bsplits) = cut(x, bbox_batch)
    boxes_np, ocnt, assoc = O.transform_boxes([b.numpy() for b in bbox_batch])
    src, box_of, inside = O.roi_crop(scene.coords0, boxes_np, assoc)
    assert np.array_equal(sel.src_row.cpu().numpy(), src) and cnt == ocnt and bsplits.tolist() == [scene.n(0)]
    new_coords = np.concatenate([scene.coords0[src][:, :3], box_of[:, None]], 1)
    assert np.array_equal(out.get_spatial_locations().numpy(), new_coords)    # mode 0: rows stay as selected
    assert torch.equal(out.features.cpu(), x.features.cpu()[torch.from_numpy(src)])
    assert out.batch_size() == 64


# ------------------------------------------------------------------------------------------------ cfg 4 (DP)
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _load_masks(z):
    """ReLU sign masks a dp_rank_worker saved (bit-packed), in forward order."""
    out = []
    for i, (r, c) in enumerate(z["mask_shapes"].tolist()):
        out.append(torch.from_numpy(np.unpackbits(z[f"mask{i}"])[:r * c].astype(bool).reshape(r, c)))
    return out


def _both_ranks(fn):
    """fn(0), fn(1) on two host threads (the oracle is torch-on-CPU: its kernels release the GIL); a failure in either
    thread re-raises here."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as ex:
        futs = [ex.submit(fn, r) for r in range(2)]
        return [f.result() for f in futs]


@pytest.mark.parametrize("weighting", ["equal", "count"])
def test_two_rank_dp_step_matches_oracle(gpu, tmp_path, weighting):
    """BASELINE configs[3]'s mechanism on one GPU: two fresh processes (gloo, both on cuda:0), one scene each, one step of
    the real Backbone 32-64-128-256 through trainstep.SceneStep with the bucketed all-reduce overlapped with backward.
    The averaged gradient every rank ends up with == the mean of the two scenes' ORACLE gradients (each oracle run with the
    ReLU masks its rank recorded).  "count": scenes of 12k and 8k voxels, every rank weighted by its active voxels -- what a
    loss normalised by batch-level counts gives when the batch is sharded one scene per rank (loss.py:401-431)."""
    targets, grid = ((12_000, 12_000) if weighting == "equal" else (12_000, 8_000)), (128, 128, 64)
    port = _free_port()
    procs, outs = [], []
    for r in range(2):
        out = str(tmp_path / f"rank{r}.npz")
        outs.append(out)
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_rank_worker.py"), out,
                                       "/".join(map(str, targets)), ",".join(map(str, grid)), "cfg2", "f32", "0", "-1", weighting],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    z = [np.load(o) for o in outs]
    from sparse_rcnn_amd.synthetic import make_batch
    ch = [32, 64, 128, 256]
    names = [n for n, _ in O.unet_param_shapes(7, ch)]
    shapes = dict(O.unet_param_shapes(7, ch))
    for k in names:
        assert np.array_equal(z[0][k], z[1][k]), k                            # broadcast: ranks hold the same parameters
        assert np.array_equal(z[0]["g:" + k], z[1]["g:" + k]), k               # ... and the same reduced gradient
    n_act = [int(z[r]["n_active"]) for r in range(2)]
    w = [0.5, 0.5] if weighting == "equal" else [n / sum(n_act) for n in n_act]
    assert weighting == "equal" or n_act[0] != n_act[1]
    def oracle_rank(r):
        coords, feats, size, bs, _ = make_batch(1, grid, targets[r], dup=1.15, seed=10 + r)
        scene = O.OracleScene(coords.numpy())
        assert scene.n(0) == n_act[r]
        po = {k: torch.from_numpy(z[0][k]).view(shapes[k]).requires_grad_() for k in names}
        out = O.unet_forward(scene, feats, po, ch, relu=O.FrozenReLU(_load_masks(z[r])))     # this rank's recorded masks
        gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(100 + r))
        out.backward(gy)
        return {k: po[k].grad * w[r] for k in names}
    grs = _both_ranks(oracle_rank)                       # the two scenes' oracle passes side by side (host threads)
    mean = {k: grs[0][k] + grs[1][k] for k in names}
    for k in names:
        _check_grad_frozen("cfg4_two_rank_dp_step_" + weighting, "mean grad " + k, torch.from_numpy(z[0]["g:" + k]).reshape(-1),
                           mean[k].reshape(-1), FROZEN_L2_F32)


def test_two_rank_dp_gradient_accumulation_matches_oracle(gpu, tmp_path):
    """the reference's batch scaling is `(loss / batches_per_step).backward()` N times, then ONE
    optimizer step (ndsis/training/training.py:436,458-460).  Two fresh gloo ranks on cuda:0, TWO micro-batches (two scenes)
    per rank through the step executor with the bucketed all-reduce: the first under FlatParams.accumulate() (hooks pack
    nothing), the second packs the accumulated `.grad`.  The reduced gradient == the mean over ranks of the oracle's
    gradients SUMMED over both micro-batches (each scaled by 1/2, each with the ReLU masks its forward recorded)."""
    target, grid, bps = 8_000, (128, 128, 64), 2
    port = _free_port()
    procs, outs = [], []
    for r in range(2):
        out = str(tmp_path / f"rank{r}.npz")
        outs.append(out)
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_rank_worker.py"), out, str(target),
                                       ",".join(map(str, grid)), "cfg2", "f32", "0", "-1", "equal", str(bps)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    z = [np.load(o) for o in outs]
    from sparse_rcnn_amd.synthetic import make_batch
    ch = [32, 64, 128, 256]
    names = [n for n, _ in O.unet_param_shapes(7, ch)]
    shapes = dict(O.unet_param_shapes(7, ch))
    for k in names:
        assert np.array_equal(z[0][k], z[1][k]), k
        assert np.array_equal(z[0]["g:" + k], z[1]["g:" + k]), k
    def oracle_rank(r):
        masks = _load_masks(z[r])
        assert len(masks) == 31 * bps
        gen = torch.Generator().manual_seed(100 + r)             # SceneStep draws dY of micro-batch 0, then of micro-batch 1
        tot = None
        for k in range(bps):
            coords, feats, size, bs, _ = make_batch(1, grid, target, dup=1.15, seed=10 + r + 1000 * k)
            scene = O.OracleScene(coords.numpy())
            assert scene.n(0) == int(z[r]["n_active_per"][k])
            po = {n: torch.from_numpy(z[0][n]).view(shapes[n]).requires_grad_() for n in names}
            out = O.unet_forward(scene, feats, po, ch, relu=O.FrozenReLU(masks[31 * k:31 * (k + 1)]))
            gy = torch.randn(out.shape, generator=gen)
            out.backward(gy / bps)
            tot = {n: po[n].grad if tot is None else tot[n] + po[n].grad for n in names}
        return {n: tot[n] / 2 for n in names}, {n: po[n].grad / 2 for n in names}
    grs = _both_ranks(oracle_rank)
    for k in names:
        got = torch.from_numpy(z[0]["g:" + k]).reshape(-1)
        mean = (grs[0][0][k] + grs[1][0][k]).reshape(-1)
        _check_grad_frozen("cfg4_two_rank_dp_accumulation", "mean grad " + k, got, mean, FROZEN_L2_F32)
    # ... and it is NOT one micro-batch's gradient (what the bucket hooks of round 4 reduced without a word)
    k = "enc3.res1.conv1.weight"
    last_only = (grs[0][1][k] + grs[1][1][k]).reshape(-1)
    got = torch.from_numpy(z[0]["g:" + k]).reshape(-1)
    assert (got - last_only).norm() > 0.1 * got.norm()


@pytest.mark.parametrize("case", ["bf16", "f32-empty-rank"])
def test_two_rank_dp_cfg3_step_matches_oracle(gpu, tmp_path, case):
    """BASELINE configs[3] (data-parallel detection + mask step) on what one GPU can run: two fresh
    gloo ranks on cuda:0, one scene and its 16 boxes each, ONE flat buffer over backbone + mask branch, bucketed all-reduce
    from the gradient hooks.  The averaged flat gradient every rank ends up with == the mean of the two scenes' ORACLE
    gradients (backbone and mask parameters), each oracle run taking the ReLU sign masks its rank's forward recorded
    (fp32 <= FROZEN_L2_F32 = 2e-5, bf16 storage against the oracle with the same roundings <= FROZEN_L2_BF16 = 2e-2 relative
    L2 per tensor).  Round 4: the ranks run the step executor (stage nodes hand a level's gradients to the bucket hooks
    together) -- the production path -- and record their masks from it.
    "f32-empty-rank": rank 1's boxes catch no point -- its mask branch produces no gradient and its buckets go out
    (zero-filled) in the same order as rank 0's."""
    dtype = "bf16" if case == "bf16" else "f32"
    empty = 1 if case.endswith("empty-rank") else -1
    target, grid, n_boxes = 12_000, (128, 128, 64), 8
    port = _free_port()
    procs, outs = [], []
    for r in range(2):
        out = str(tmp_path / f"rank{r}.npz")
        outs.append(out)
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_rank_worker.py"), out, str(target),
                                       ",".join(map(str, grid)), "cfg3", dtype, str(n_boxes), str(empty), "equal"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    z = [np.load(o) for o in outs]
    from sparse_rcnn_amd.synthetic import make_batch, make_boxes
    ch = [32, 64, 128, 256]
    bnames = [n for n, _ in O.unet_param_shapes(7, ch)]
    mnames = [k[2:] for k in z[0].files if k.startswith("m:")]
    assert len(mnames) == 80 and len(bnames) == 76
    keys = bnames + ["m:" + k for k in mnames]
    for k in keys:
        assert np.array_equal(z[0][k], z[1][k]), k                            # broadcast: ranks hold the same parameters
        assert np.array_equal(z[0]["g:" + k], z[1]["g:" + k]), k               # ... and the same reduced gradient
    bf16 = dtype == "bf16"
    def oracle_rank(r):
        coords, feats, size, bs, _ = make_batch(1, grid, target, dup=1.15, seed=10 + r)
        boxes = make_boxes(coords, n_boxes, seed=10 + r + 2)
        if r == empty:
            boxes = [b + 10_000.0 for b in boxes]
        pb = {k: z[0][k] for k in bnames}
        pm = {k: z[0]["m:" + k] for k in mnames}
        masks = _load_masks(z[r])                        # the ReLU sign masks this rank's HIP forward recorded
        assert len(masks) == (31 + 4 if r == empty else 63)
        out, logits, gr, n_sel = _oracle_cfg3_step(coords, feats, boxes, pb, pm, ch, 100 + r, bf16=bf16, masks=masks)
        assert out.shape[0] == int(z[r]["n_active"]) and n_sel == int(z[r]["n_roi_rows"]), (r, n_sel)
        assert (n_sel == 0) == (r == empty)
        return {k: gr[k] / 2 for k in keys}
    grs = _both_ranks(oracle_rank)                       # the two scenes' oracle passes side by side (host threads)
    mean = {k: grs[0][k] + grs[1][k] for k in keys}
    name = "cfg3_two_rank_dp_step_" + case
    for k in keys:
        _check_grad_frozen(name, "mean grad " + k, torch.from_numpy(z[0]["g:" + k]).reshape(-1), mean[k].reshape(-1),
                           FROZEN_L2_BF16 if bf16 else FROZEN_L2_F32)


def test_syncbn_two_ranks_equals_single_process(gpu, tmp_path):
    """SyncBN (modules._BatchNorm.SYNC): two ranks hold uneven shares of one feature matrix; outputs, input gradients,
    the summed parameter gradients and the running statistics equal the single-process layer on the whole matrix -- what
    the reference's single-process batch computes (module_factory.py:92-102)."""
    import sparse_rcnn_amd as scn
    port = _free_port()
    procs, outs = [], []
    for r in range(2):
        out = str(tmp_path / f"bn{r}.npz")
        outs.append(out)
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "syncbn_rank_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), logs
    z = [np.load(o) for o in outs]
    g = torch.Generator().manual_seed(7)
    X = torch.randn(5000, 24, generator=g) * 2 + 0.5
    G = torch.randn(5000, 24, generator=g)
    bn = scn.BatchNormLeakyReLU(24, leakiness=0.2).to(gpu)
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, 24)); bn.bias.copy_(torch.linspace(-0.3, 0.3, 24))
    x = X.to(gpu).requires_grad_()
    y = bn(scn.SparseConvNetTensor(features=x, metadata=None, spatial_size=None)).features
    y.backward(G.to(gpu))
    cat = lambda k: torch.from_numpy(np.concatenate([z[0][k], z[1][k]]))
    for what, a, b, tol in (("forward", cat("y"), y, 1e-6), ("dX", cat("dx"), x.grad, 1e-5),
                            ("dgamma (sum over ranks)", torch.from_numpy(z[0]["dg"] + z[1]["dg"]), bn.weight.grad, 1e-5),
                            ("dbeta (sum over ranks)", torch.from_numpy(z[0]["db"] + z[1]["db"]), bn.bias.grad, 1e-5),
                            ("running_mean", torch.from_numpy(z[0]["rm"]), bn.running_mean, 1e-6),
                            ("running_var", torch.from_numpy(z[1]["rv"]), bn.running_var, 1e-6)):
        e = _err(a, b)
        _record("syncbn_two_ranks", what, e, tol)
        assert e["rel_to_scale"] <= tol, (what, e)


# ------------------------------------------------------------------------------------------------ cfg 5 shape
def test_cfg5_shape_bf16_properties(gpu):
    """BASELINE configs[4]'s per-GPU shape (600k voxels, 5 levels to 512 channels) in bf16 storage: the oracle is not the
    checker at this size in bf16 (its fp32 result differs by the storage roundings); size-independent properties instead:
    finite, bitwise reproducible, the bf16 run tracks the fp32 run of the same parameters, index structures consistent."""
    from sparse_rcnn_amd.synthetic import make_batch
    from sparse_rcnn_amd.unet import Backbone
    coords, feats, size, bs, _ = make_batch(1, (1024, 1024, 512), 600_000, dup=1.15, seed=2)
    ch = (32, 64, 128, 256, 512)
    torch.manual_seed(0)
    ref = Backbone(7, ch).to(gpu)
    low = Backbone(7, ch, bf16_blocks="all").to(gpu)
    low.load_state_dict(ref.state_dict())
    cd, fd = coords.to(gpu), feats.to(gpu)
    outs = []
    for net in (ref, low, low):
        fin = fd.clone().requires_grad_()
        o = net(cd, fin, size, 1).features
        o.backward(torch.ones_like(o))
        outs.append((o.detach(), fin.grad.clone(), [p.grad.clone() for p in net.parameters()]))
        for p in net.parameters():
            p.grad = None
    assert outs[0][0].shape == (600_000, 32)
    assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])      # bitwise reproducible
    for a, b in zip(outs[1][2], outs[2][2]):
        assert torch.equal(a, b)
    e = _err(outs[1][0], outs[0][0])
    _record("cfg5_shape_bf16", "bf16-storage forward vs fp32 forward (same parameters)", e, 3e-2)
    assert torch.isfinite(outs[1][0]).all() and e["rel_l2"] < 3e-2, e
    e = _err(outs[1][1], outs[0][1])
    _record("cfg5_shape_bf16", "bf16-storage input gradient vs fp32", e, 0.1)
    assert torch.isfinite(outs[1][1]).all() and e["rel_l2"] < 0.1, e


def test_reference_plan_on_its_training_batch_of_12_crops_vs_oracle(gpu):
    """`--workload ref-crop`: the network the reference trains (six levels 32-48-64-80-96-112) on the batch it trains on -- 12
    random crops of 128 x 128 x 64 voxels (scannet_config/run.py:364,485-488), ~12 500 active voxels each, ONE batch of twelve
    samples: through the step executor (the coarse levels take the four-waves-per-tile loop, the 48 / 80 / 112-channel layers the
    TAIL slices, the hash holds twelve batch indices), forward within 1e-4 of the scale and every one of the 120 parameter
    gradients + the input gradient within FROZEN_L2_F32 of the oracle with the HIP forward's ReLU masks."""
    from sparse_rcnn_amd.synthetic import make_batch
    from sparse_rcnn_amd.trainstep import REF_PLAN
    from sparse_rcnn_amd.unet import Backbone
    from sparse_rcnn_amd import _lib as L
    coords, feats, size, bs, splits = make_batch(12, (128, 128, 64), 12_500, dup=1.15, seed=21)
    ch = REF_PLAN
    scene = O.OracleScene(coords.numpy())
    params = O.init_unet_params(7, list(ch), seed=5)
    net = Backbone(7, ch).to(gpu)
    net.unet.load_oracle_params(params)
    fin = feats.to(gpu).requires_grad_()
    L.lib().scn_conv_tiles_split_count(1)
    with _record_relu_masks() as masks:
        out = net(coords, fin, size, bs)
    assert len(masks) == 6 * 4 + 5 * 5 and _n_stage_nodes(out.features) == 11
    gy = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(8))
    out.features.backward(gy.to(gpu))
    torch.cuda.synchronize()
    assert L.lib().scn_conv_tiles_split_count(1) > 0                    # coarse levels: four waves per tile
    po = {k: v.clone().requires_grad_() for k, v in params.items()}
    fo = feats.clone().requires_grad_()
    fr = O.FrozenReLU(masks)
    exp = O.unet_forward(scene, fo, po, list(ch), relu=fr)
    assert fr.k == len(masks)
    exp.backward(gy)
    own = O.unet_forward(scene, fo.detach(), {k: v.detach() for k, v in po.items()}, list(ch))
    name = "ref_plan_12_crops"
    e = _err(out.features, own)
    _record(name, "forward features", e, FEAT_TOL)
    assert out.features.shape[0] == scene.n(0) and out.batch_size() == 12 and e["rel_to_scale"] <= FEAT_TOL, e
    for k, p in list(net.unet.named_oracle_params().items()) + [("input features", fin)]:
        ref = fo.grad if p is fin else po[k].grad.view_as(p)
        _check_grad_frozen(name, "grad " + k, p.grad, ref, FROZEN_L2_F32)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_cfg5_shape_vs_oracle_at_600k(gpu, dtype):
    """BASELINE configs[4]'s per-GPU shape against the ORACLE (round 4: the suite's oracle passes run on the job's real CPU
    threads, so the largest configuration fits too): 600k voxels, five levels 32-64-128-256-512, through the step executor.
    fp32: forward within 1e-4 of the scale, every one of the 96 parameter gradients and the input gradient within
    FROZEN_L2_F32 of the oracle with the HIP forward's ReLU masks.  bf16 storage: the oracle with the same roundings (stored
    slabs and tile-kernel weights rounded to bf16), forward <= 2^-6 of the scale / 1e-2 relative L2, gradients FROZEN_L2_BF16.
    (n_in >= 2^23 rows is where the raw-buffer fast path ends; 600k rows stay far below: `fast_path` of the bench line.)"""
    from sparse_rcnn_amd.synthetic import make_batch
    from sparse_rcnn_amd.unet import Backbone
    bf16 = dtype == "bf16"
    coords, feats, size, bs, _ = make_batch(1, (1024, 1024, 512), 600_000, dup=1.15, seed=2)
    ch = (32, 64, 128, 256, 512)
    scene = O.OracleScene(coords.numpy())
    params = O.init_unet_params(7, list(ch), seed=3)
    net = Backbone(7, ch, bf16_blocks="all" if bf16 else False).to(gpu)
    net.unet.load_oracle_params(params)
    fin = feats.to(gpu).requires_grad_()
    with _record_relu_masks() as masks:
        out = net(coords, fin, size, 1)
    assert out.features.shape == (600_000, 32) and len(masks) == 9 * 4 + 4 and _n_stage_nodes(out.features) == 9
    gy = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(7))
    out.features.backward(gy.to(gpu))
    torch.cuda.synchronize()
    po = {k: v.clone().requires_grad_() for k, v in params.items()}
    fo = feats.clone().requires_grad_()
    fr = O.FrozenReLU(masks)
    kw = dict(storage=O.bf16_storage, tile_weights=O.bf16_storage) if bf16 else {}
    exp = O.unet_forward(scene, fo, po, list(ch), relu=fr, **kw)
    assert fr.k == len(masks)
    exp.backward(gy)
    own = O.unet_forward(scene, fo.detach(), {k: v.detach() for k, v in po.items()}, list(ch), **kw)
    name = "cfg5_shape_600k" + ("_bf16_storage" if bf16 else "")
    e = _err(out.features, own)                       # forward: against the oracle's OWN ReLU decisions
    if bf16:
        _record(name, "forward features vs oracle with the same roundings", e, "rel_to_scale <= 2^-6, rel_l2 <= 1e-2")
        assert e["rel_to_scale"] <= 2.0 ** -6 and e["rel_l2"] <= 1e-2, e
    else:
        _record(name, "forward features", e, FEAT_TOL)
        assert e["rel_to_scale"] <= FEAT_TOL, e
    bound = FROZEN_L2_BF16 if bf16 else FROZEN_L2_F32
    for k, p in list(net.unet.named_oracle_params().items()) + [("input features", fin)]:
        ref = fo.grad if p is fin else po[k].grad.view_as(p)
        _check_grad_frozen(name, "grad " + k, p.grad, ref, bound)


def test_bench_self_launch_two_ranks_on_one_gpu(gpu):
    """`python bench.py --gpus 2` with no torchrun environment starts its own two ranks (gloo here, so that both may share
    cuda:0) and prints ONE JSON line on stdout with n_gpus = 2 -- the launch path of the driver's scaling run."""
    env = dict(os.environ, SCN_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--target", "30000", "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["roofline"]["kernel"] == "k_conv_ts" and 0 < d["roofline"]["frac"] < 1


def test_bench_self_launch_five_ranks_detection_and_mask_step_on_one_gpu(gpu):
    """VERDICT r3 item 8, within this pool's limit of SIX processes with the card open (this test process is one of them; the
    8-process form may not be started here -- a 6-rank attempt was killed by the box's process guard): five fresh ranks of
    `bench.py --gpus 5 --workload cfg3 --dtype bf16` from a parent that makes no GPU call -- each with its index helper thread,
    the stage nodes' bucket hooks over backbone + mask branch, gloo so that they may share cuda:0.  One JSON line, five ranks
    seen, five per-rank entries.  Then the same launch with a rank that leaves after warm-up: the launcher terminates the four
    survivors (who would sit in a collective) and returns the dead rank's code within seconds of it."""
    import time
    env = dict(os.environ, SCN_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "5", "--steps", "5", "--warmup", "2", "--target", "20000",
           "--workload", "cfg3", "--dtype", "bf16", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 5 and d["n_ranks_seen"] == 5 and len(d["per_rank"]) == 5 and d["value"] > 0
    assert sorted(p["rank"] for p in d["per_rank"]) == list(range(5))
    assert all(p["n_active"] > 15_000 and p["n_roi_rows"] > 0 and p["ms_per_step"] > 0 for p in d["per_rank"])
    assert d["dtype"] == "bf16" and "configs[2]" in d["config"]["workload"]
    t0 = time.time()
    r = subprocess.run(cmd, env=dict(env, SCN_BENCH_DIE_RANK="3"), capture_output=True, text=True, timeout=600)
    took = time.time() - t0
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert "rank 3 exited with code 3" in r.stderr and not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert took < 240, took                      # (start-up of five processes dominates; the survivors do not wait for a time limit)
