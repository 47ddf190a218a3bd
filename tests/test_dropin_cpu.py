"""CPU: the scn operator surface the reference calls and the drop-in fixture (§8c iii)."""
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

import sparse_rcnn_amd as scn
from sparse_rcnn_amd.unet import SparseUNet

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "dropin_feature_extractor.json")))


def test_surface_matches_the_reference_call_sites():
    # every scn.* name at the 36 call-site lines
    for name in ("Metadata", "SparseConvNetTensor", "ioLayers", "Sequential", "ConcatTable", "AddTable", "JoinTable",
                 "Identity", "ReLU", "BatchNormReLU", "BatchNormLeakyReLU", "Convolution", "Deconvolution",
                 "SubmanifoldConvolution", "NetworkInNetwork", "MaxPooling", "AveragePooling", "SparseToDense",
                 "OutputLayer", "InputLayer"):
        assert hasattr(scn, name), name
    assert hasattr(scn.ioLayers, "InputLayerFunction") and hasattr(scn.ioLayers, "OutputLayerFunction")
    # constructor call shapes used by module_factory.py (numpy-int tuples for size/stride: :225-234)
    st = tuple(np.full(3, 2))
    c = scn.Convolution(3, 32, 64, filter_size=st, filter_stride=st, bias=True)
    d = scn.Deconvolution(3, 64, 32, filter_size=st, filter_stride=st, bias=True)
    s = scn.SubmanifoldConvolution(3, 32, 32, filter_size=3, bias=True, groups=1)
    s1 = scn.SubmanifoldConvolution(3, 7, 32, filter_size=1, bias=True)
    n = scn.NetworkInNetwork(64, 32, True)
    assert c.weight.shape == (8, 32, 64) and d.weight.shape == (8, 64, 32)
    assert s.weight.shape == (27, 32, 32) and s1.weight.shape == (1, 7, 32) and n.weight.shape == (64, 32)
    bn = scn.BatchNormLeakyReLU(16, 1e-4, 0.9, 0.2)
    assert sorted(bn.state_dict()) == ["bias", "running_mean", "running_var", "weight"]
    seq = scn.Sequential(scn.ConcatTable(scn.Identity(), s), scn.AddTable())
    seq.append(scn.ReLU())
    assert len(seq) == 3
    with pytest.raises(NotImplementedError):
        scn.Convolution(3, 4, 4, (3, 3, 3), (2, 2, 2), True)
    scn.MaxPooling(3, pool_size=st, pool_stride=st); scn.AveragePooling(3, pool_size=st, pool_stride=st)
    with pytest.raises(NotImplementedError):
        scn.MaxPooling(3, 3, 2)
    g2 = scn.SubmanifoldConvolution(3, 4, 6, 3, True, groups=2)                # SparseConvNet's grouped layout (round 4)
    assert g2.weight.shape == (27, 2, 2, 3) and g2._wb(4)[0].shape == (27, 4, 6)
    Wd = g2._wb(4)[0]
    assert torch.equal(Wd[:, :2, :3], g2.weight[:, 0]) and torch.equal(Wd[:, 2:, 3:], g2.weight[:, 1])
    assert not Wd[:, :2, 3:].any() and not Wd[:, 2:, :3].any()
    with pytest.raises(ValueError):
        scn.SubmanifoldConvolution(3, 4, 4, 3, True, groups=3)


def test_grouped_checkpoint_layout_loads():
    s = scn.SubmanifoldConvolution(3, 4, 6, 3, True)
    sd = {"weight": torch.randn(27, 1, 4, 6), "bias": torch.zeros(6)}      # SparseConvNet's [fv, groups, nIn, nOut]
    s.load_state_dict(sd)
    assert torch.equal(s.weight.data, sd["weight"].squeeze(1))


@pytest.mark.parametrize("name", sorted(FIX))
def test_own_graph_builder_reproduces_the_reference_feature_extractor(name):
    """The fixture was produced by building the REFERENCE's FeatureExtractor on top of this package
    (tests/golden/make_dropin_golden.py).  This package's own builder must hold the same parameters."""
    fx = FIX[name]
    net = SparseUNet(7, fx["channels"])
    own = sorted(tuple(v.shape) for v in net.state_dict().values())
    ref = sorted(tuple(v) for v in fx["keys"].values())
    assert own == ref
    assert sum(v.numel() for v in net.state_dict().values()) == fx["n_params"]
    census = {}
    for m in net.modules():
        if type(m).__module__.startswith("sparse_rcnn_amd") and type(m).__name__ != "SparseUNet":
            census[type(m).__name__] = census.get(type(m).__name__, 0) + 1
    for k in ("SubmanifoldConvolution", "Convolution", "Deconvolution", "NetworkInNetwork", "ReLU", "AddTable",
              "JoinTable", "ConcatTable"):
        assert census[k] == fx["census"][k], (k, census[k], fx["census"][k])


_LEAF = ("SubmanifoldConvolution(", "Convolution(", "Deconvolution(", "NetworkInNetwork(", "ReLU()", "AddTable()",
         "JoinTable()", "Identity()")


def _leaf_lines(tree_repr):
    """The scn layer lines of a module-tree repr, in tree order, without their container indices."""
    out = []
    for line in tree_repr.splitlines():
        body = line.strip().split(": ", 1)[-1]
        if body.startswith(_LEAF):
            out.append(body)
    return out


@pytest.mark.parametrize("name", sorted(FIX))
def test_reference_checkpoint_names_and_layer_order(name):
    """SURVEY §8c(iii) / VERDICT r2 item 8: the fixture holds the state_dict key NAMES and the module-tree repr of the
    reference's FeatureExtractor built on this package.  (a) unet.reference_key_map covers exactly those key names and maps
    each onto a parameter of this package's own builder with the fixture's shape; (b) a state dict with those names loads
    through Backbone.load_reference_state_dict (grouped [fv, 1, nIn, nOut] weights included) and every parameter arrives;
    (c) the scn layers appear in the same order, with the same channel signatures, in both module trees."""
    from sparse_rcnn_amd.unet import Backbone, reference_key_map
    fx = FIX[name]
    ch = fx["channels"]
    kmap = reference_key_map(len(ch))
    assert set(kmap) == set(fx["keys"]), sorted(set(kmap) ^ set(fx["keys"]))[:6]
    net = Backbone(7, ch)
    own = net.unet.named_oracle_params()
    assert sorted(kmap.values()) == sorted(own)
    for rk, shape in fx["keys"].items():
        assert list(own[kmap[rk]].shape) == shape, (rk, kmap[rk])
    g = torch.Generator().manual_seed(0)
    sd = {"feature_extractor." + rk: torch.randn(shape, generator=g) for rk, shape in fx["keys"].items()}
    k27 = next(k for k, v in sd.items() if v.dim() == 3 and v.shape[0] == 27)
    sd[k27] = sd[k27].unsqueeze(1)                                    # SparseConvNet's grouped layout
    sd["rpn.something.weight"] = torch.zeros(3)                       # other parts of the model's checkpoint are ignored
    missing, unused = net.load_reference_state_dict(sd)
    assert not missing and not unused
    for rk in fx["keys"]:
        t = sd["feature_extractor." + rk]
        assert torch.equal(own[kmap[rk]].detach(), t.squeeze(1) if t.dim() == 4 else t), rk
    with pytest.raises(KeyError):
        net.load_reference_state_dict({k: v for k, v in sd.items() if not k.endswith("unet.module_list.0.channel_changer.bias")})
    ref_leaves = _leaf_lines(fx["repr"])
    own_leaves = _leaf_lines(repr(net.unet))
    assert ref_leaves == own_leaves, [(i, a, b) for i, (a, b) in enumerate(zip(ref_leaves, own_leaves)) if a != b][:4]
    assert "SkipConnectionReuniter" in fx["repr"] and "SequentialInterims" in fx["repr"]      # the reference's containers


TAPES = json.load(open(os.path.join(HERE, "golden", "optape_reference_forward.json")))


@pytest.mark.parametrize("name", ["cfg2_32_256", "ref_32_112"])
def test_dropin_backbone_issues_the_op_tape_of_the_reference_forward(name):
    """tests/golden/optape_reference_forward.json holds the leaf-operator tape of the REFERENCE's
    FeatureExtractor.forward (model.py:414-446) run on this package with the arithmetic replaced by shapes
    (tests/optape.py, generator tests/golden/make_optape_golden.py): per call the layer signature, the tape index of the
    entry that produced each operand, spatial size, rows and channels.  unet.DropinBackbone -- this repository's imitation
    of that call pattern, the thing bench.py's `dropin` leg and the GPU tests drive -- issues exactly that tape on the same
    seeded scene (here under the same shape stub; on the GPU with real kernels: tests/test_gpu_exec.py)."""
    sys.path.insert(0, HERE)
    import optape
    from sparse_rcnn_amd.synthetic import make_batch
    from sparse_rcnn_amd.unet import Backbone, DropinBackbone
    sc = TAPES["scene"]
    coords, feats, size, bs, splits = make_batch(sc["n_samples"], tuple(sc["grid"]), sc["target"], dup=sc["dup"], seed=sc["seed"])
    ch = [32, 64, 128, 256] if name == "cfg2_32_256" else [32, 48, 64, 80, 96, 112]
    net = DropinBackbone(Backbone(7, ch))
    with optape.shape_stub(), optape.record() as tape:
        out = net(coords, feats, size, bs)
    ref = TAPES["feature_extractor_" + name]
    assert len(tape.entries) == len(ref) == (87 if name == "cfg2_32_256" else 137)
    for i, (a, b) in enumerate(zip(tape.entries, ref)):
        assert a == b, (i, a, b)
    assert tuple(out.features.shape) == (ref[0]["rows"], 32)


MASK_FIXTURES = json.load(open(os.path.join(HERE, "golden", "dropin_mask_network.json")))
MASK_FIX = MASK_FIXTURES["run_config"]


def test_mask_branch_reproduces_the_reference_mask_network():
    """tests/golden/dropin_mask_network.json holds the state_dict key NAMES + shapes, the scn layer census
    and the module-tree repr of the reference's SparseMaskNetwork (model.py:572-782, configured as scannet_config/run.py:
    741-810) built on this package.  (a) maskhead.reference_key_map covers exactly those keys and maps each onto a parameter
    of MaskBranch with the fixture's shape; (b) a state dict with those names -- under the `mask_network.` prefix of the
    whole model's checkpoint, one weight in SparseConvNet's grouped layout -- loads through
    MaskBranch.load_reference_state_dict and every parameter arrives; (c) the scn layers appear in the same order with the
    same channel signatures in both module trees (the FLD('I') level contributes the reference's lone Identity)."""
    from sparse_rcnn_amd.maskhead import MaskBranch, reference_key_map
    fx = MASK_FIX
    mb = MaskBranch(32, 7, linear_channels=(32, fx["classes"]))
    kmap = reference_key_map(4)
    assert set(kmap) == set(fx["keys"]), sorted(set(kmap) ^ set(fx["keys"]))[:6]
    own = mb.named_oracle_params()
    assert sorted(kmap.values()) == sorted(own) and len(own) == 80
    for rk, shape in fx["keys"].items():
        assert list(own[kmap[rk]].shape) == shape, (rk, kmap[rk])
    assert sum(p.numel() for p in mb.parameters()) == fx["n_params"] == 1_342_506
    g = torch.Generator().manual_seed(0)
    sd = {"mask_network." + rk: torch.randn(shape, generator=g) for rk, shape in fx["keys"].items()}
    k27 = next(k for k, v in sd.items() if v.dim() == 3 and v.shape[0] == 27)
    sd[k27] = sd[k27].unsqueeze(1)                                    # SparseConvNet's grouped layout
    sd["feature_extractor.main_network.0.0.0.weight"] = torch.zeros(1, 7, 32)          # other parts of the checkpoint are ignored
    missing, unused = mb.load_reference_state_dict(sd)
    assert not missing and not unused
    for rk in fx["keys"]:
        t = sd["mask_network." + rk]
        assert torch.equal(own[kmap[rk]].detach(), t.squeeze(1) if t.dim() == 4 else t), rk
    with pytest.raises(KeyError):
        mb.load_reference_state_dict({k: v for k, v in sd.items() if not k.endswith("linear_layer.2.bias")})
    # layer order: input_conv_layer, then the internal U-Net (its Identity level first), in tree order
    ref_leaves = _leaf_lines(fx["repr"])
    own_leaves = _leaf_lines(repr(mb.input_conv_layer)) + _leaf_lines(repr(mb.output_conv_layer))
    # MaskBranch's builder spells the FLD('I') level as Sequential(Identity, Identity) (unet.SparseUNet, identity_first); the
    # reference as one Identity
    first_id = own_leaves.index("Identity()", len(_leaf_lines(repr(mb.input_conv_layer))))
    assert own_leaves[first_id + 1] == "Identity()"
    del own_leaves[first_id + 1]
    assert ref_leaves == own_leaves, [(i, a, b) for i, (a, b) in enumerate(zip(ref_leaves, own_leaves)) if a != b][:4]
    for cls in ("UnetContainer", "SkipConnectionReuniter", "SequentialInterims", "SparseFeaturemapSelectorBoth",
                "SparseFeaturemapFirst", "TrainSelector"):
        assert cls in fx["repr"], cls                                 # the reference's containers around this package's layers
    for k in ("SubmanifoldConvolution", "Convolution", "Deconvolution", "NetworkInNetwork", "ReLU", "AddTable", "JoinTable",
              "ConcatTable"):
        got = sum(1 for m in list(mb.input_conv_layer.modules()) + list(mb.output_conv_layer.modules()) if type(m).__name__ == k)
        assert got == fx["census"][k], (k, got, fx["census"][k])


@pytest.mark.parametrize("variant", ["unet_only", "raw_only", "both_skip", "raw_skip"])
def test_mask_branch_variants_reproduce_the_reference_mask_network(variant):
    """VERDICT r3 missing 4: the other feature-map selectors / the combiner of SparseMaskNetwork (model.py:597-651:
    `SparseFeaturemapSelector` = use_raw_features False, `SparseFeaturemapSelectorRaw` = use_unet_features False,
    `SparseFeaturemapCombiner` = use_skip_features True) as MaskBranch keyword switches.  Per variant the fixture holds keys +
    shapes + repr of the reference's network built with that switch: same key set through reference_key_map, same shapes (the
    Linear stack's input width moves with the switches: 16 / 16 / 23+7 / 16+7 -- a 7-channel internal U-Net comes up 16 wide,
    `min_channels`), same parameter count, same selector class."""
    from sparse_rcnn_amd.maskhead import MaskBranch, reference_key_map
    fx = MASK_FIXTURES[variant]
    mb = MaskBranch(32, 7, linear_channels=(32, fx["classes"]), **fx["flags"])
    kmap = reference_key_map(4, with_input=mb.input_conv_layer is not None)
    assert set(kmap) == set(fx["keys"]), sorted(set(kmap) ^ set(fx["keys"]))[:6]
    own = mb.named_oracle_params()
    assert sorted(kmap.values()) == sorted(own)
    for rk, shape in fx["keys"].items():
        assert list(own[kmap[rk]].shape) == shape, (rk, kmap[rk])
    assert sum(p.numel() for p in mb.parameters()) == fx["n_params"]
    g = torch.Generator().manual_seed(1)
    sd = {"mask_network." + rk: torch.randn(shape, generator=g) for rk, shape in fx["keys"].items()}
    missing, unused = mb.load_reference_state_dict(sd)
    assert not missing and not unused
    assert all(torch.equal(own[kmap[rk]].detach(), sd["mask_network." + rk]) for rk in fx["keys"])
    selector = {"unet_only": "SparseFeaturemapSelector(", "raw_only": "SparseFeaturemapSelectorRaw(",
                "both_skip": "SparseFeaturemapSelectorBoth(", "raw_skip": "SparseFeaturemapSelectorRaw("}[variant]
    assert selector in fx["repr"]
    assert ("SparseFeaturemapCombiner(" in fx["repr"]) == variant.endswith("skip")
    c0 = {"unet_only": 16, "raw_only": 7, "both_skip": 23, "raw_skip": 7}[variant]
    assert mb.output_conv_layer.channels[0] == c0 and mb.output_conv_layer.phys0 == (c0 + 7) // 8 * 8
    assert mb.output_conv_layer.out_channels == max(c0, 16)           # unet_params['min_channels'] = 16 (run.py:786)
    assert mb.linear_layer[0].in_features == max(c0, 16) + (7 if variant.endswith("skip") else 0)
    with pytest.raises(ValueError):
        MaskBranch(32, 7, use_unet_features=False, use_raw_features=False)


@pytest.mark.skipif(not os.path.isdir("/root/reference/ndsis"), reason="reference checkout only exists in the build container")
def test_reference_feature_extractor_constructs_on_this_package():
    saved = sys.modules.get("sparseconvnet")
    sys.modules["sparseconvnet"] = scn
    sys.path.insert(0, "/root/reference")
    try:
        sys.path.insert(0, os.path.join(HERE, "golden"))
        import make_dropin_golden as g
        fe = g.build([32, 64, 128, 256])
        assert sum(p.numel() for p in fe.parameters()) == 12_457_856
        assert fe.calc_unet and list(fe.unet_channels) == [128, 64, 32]
    finally:
        sys.path.remove("/root/reference")
        if saved is None:
            sys.modules.pop("sparseconvnet", None)
        else:
            sys.modules["sparseconvnet"] = saved


@pytest.mark.skipif(not os.path.isdir("/root/reference/ndsis"), reason="reference checkout only exists in the build container")
def test_reference_factories_construct_strided_layers_on_this_package():
    """`get_downsampler` / `get_upsampler` / `get_down_maxpooling` / `get_down_avgpooling` (module_factory.py:221-258,315-354)
    with the strides they accept -- an int or one entry per axis (numpy array, as the reference passes them) -- build this
    package's Convolution / Deconvolution / pooling layers with the filter volume sx sy sz (round 4; the arithmetic:
    tests/test_gpu_parity.py::test_general_stride_conv_deconv_and_pooling)."""
    import numpy as np
    saved = sys.modules.get("sparseconvnet")
    sys.modules["sparseconvnet"] = scn
    sys.path.insert(0, "/root/reference")
    try:
        from ndsis.modules import module_factory as mf
        for st, vol in ((3, 27), (np.array([2, 2, 1]), 4), (np.array([1, 2, 3]), 6), (4, 64), (2, 8)):
            want = tuple(int(v) for v in (np.full(3, st) if np.isscalar(st) else st))
            _, _, _, d = mf.get_downsampler(3, True, 8, 16, stride=st)
            _, _, _, u = mf.get_upsampler(3, True, 16, 8, stride=st)
            _, _, _, mp = mf.get_down_maxpooling(3, True, 8, stride=st)
            _, _, _, ap = mf.get_down_avgpooling(3, True, 8, stride=st)
            assert type(d) is scn.Convolution and tuple(d.weight.shape) == (vol, 8, 16) and d.stride == want
            assert type(u) is scn.Deconvolution and tuple(u.weight.shape) == (vol, 16, 8) and u.stride == want
            assert type(mp) is scn.MaxPooling and type(ap) is scn.AveragePooling and mp.stride == ap.stride == want
    finally:
        sys.path.remove("/root/reference")
        for k in [k for k in sys.modules if k == "ndsis" or k.startswith("ndsis.")]:
            del sys.modules[k]
        if saved is None:
            sys.modules.pop("sparseconvnet", None)
        else:
            sys.modules["sparseconvnet"] = saved


def test_stage_plan_caches_are_revalidated_after_tree_edits_and_left_out_of_copies():
    """the cached shape / compiled plans of an scn.Sequential (modules._kind, `_stages`) are dropped when anything
    BELOW it changes -- the inner Sequential's units, a replaced / deleted child, a layer's bias -- and never travel with
    copy.deepcopy / pickle (they hold ctypes pointer arrays and name the original's modules)."""
    import copy
    import ctypes
    import io
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import modules as M
    from sparse_rcnn_amd.unet import Backbone

    def unit(c):
        return scn.Sequential(scn.ConcatTable(scn.Identity(), scn.Sequential(
            scn.ReLU(), scn.SubmanifoldConvolution(3, c, c, 3, True), scn.ReLU(), scn.SubmanifoldConvolution(3, c, c, 3, True))),
            scn.AddTable())
    inner = scn.Sequential(unit(16), unit(16))
    level = scn.Sequential(scn.Sequential(scn.SubmanifoldConvolution(3, 8, 16, 1, True)), inner)
    assert M._kind(level) == "enc" and M._kind(inner) == "units"
    level.__dict__["_stages"] = {"planted": (ctypes.c_void_p * 2)()}
    assert M._kind(level) == "enc" and "planted" in level.__dict__["_stages"]          # nothing changed: the cache stays
    inner.append(unit(16))
    assert M._kind(level) == "enc" and "_stages" not in level.__dict__                  # inner edit: outer plans dropped
    level.__dict__["_stages"] = {"planted": 1}
    inner[1] = scn.ReLU()                                                               # no longer plain residual units
    assert M._kind(level) is None and M._kind(inner) is None and "_stages" not in level.__dict__
    inner[1] = unit(16)
    assert M._kind(level) == "enc"
    level.__dict__["_stages"] = {"planted": 1}
    inner[0][0][1][3].bias = None
    assert M._kind(level) is None and "_stages" not in level.__dict__
    inner[0][0][1][3].bias = torch.nn.Parameter(torch.zeros(16))
    assert M._kind(level) == "enc"
    level.__dict__["_stages"] = {"planted": (ctypes.c_void_p * 2)()}
    del inner[2]
    assert "_stages" not in level.__dict__ or M._kind(level) == "enc" and "_stages" not in level.__dict__
    # copies: plans stay behind
    level.__dict__["_stages"] = {"planted": (ctypes.c_void_p * 2)()}
    M._kind(level)
    twin = copy.deepcopy(level)
    assert "_stages" not in twin.__dict__ and "_stage_sig" not in twin.__dict__ and M._kind(twin) == "enc"
    torch.save(level, io.BytesIO())
    net = Backbone(7, (16, 32))
    for k in net.unet._PLAN_ATTRS:
        object.__setattr__(net.unet, k, (ctypes.c_void_p * 2)())
    c = copy.deepcopy(net)
    assert not any(k in c.unet.__dict__ for k in net.unet._PLAN_ATTRS)
    torch.save(net, io.BytesIO())
