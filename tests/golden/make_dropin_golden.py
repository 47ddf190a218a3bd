"""Drop-in boundary fixture (SURVEY.md §8c iii): in the build container only, register this package as
``sparseconvnet``, import the REFERENCE's ndsis.modules.model and build its FeatureExtractor (sparse + U-Net) with the
parameter dict restated from scannet_config/run.py:581-627.  Stores the state_dict key list + shapes, the scn layer
census and the module-tree `repr` (the reference's container class names around this package's layer lines: output of
the run, not source text); tests/test_dropin_cpu.py checks this package's own graph builder against it -- key NAMES
through unet.reference_key_map, the order of the scn layers through the repr -- without the reference present.

    python tests/golden/make_dropin_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import sparse_rcnn_amd                                     # noqa: E402
sys.modules["sparseconvnet"] = sparse_rcnn_amd
from ndsis.modules.model import FeatureExtractor, SparseMaskNetwork, FeatureLevelDescriptor as FLD   # noqa: E402


def build(channels, cin=7):
    layer_descriptions = [FLD('B', channels[0], dict(stride=1, drop_input_relu=True)),
                          *[FLD('B', c) for c in channels[1:]]]
    unet_params = dict(use_residuals=True, num_units=2, bottleneck_divisor=0, groups=1, main_path_relu=False,
                       relu_first=True, batchnorm=False, concat=True, min_channels=16)
    params = dict(num_dims=3, sparse=True, input_channels=cin, network_description=layer_descriptions,
                  class_output_anchor=False, class_output_upsampled=False, class_output_index=-1, num_dilations=5,
                  num_units=2, bottleneck_divisor=0, stride=2, maxpool=False, relu_first=True, main_path_relu=False,
                  bottleneck_groups=1, batchnorm=False, use_residuals=True, drop_input_relu=True, include_unet=True,
                  unet_params=unet_params)
    return FeatureExtractor(**params)


def build_mask_network(backbone_channels=(32, 64, 128, 256), cin=7, num_instance_classes=18, use_unet_features=True,
                       use_raw_features=True, use_skip_features=False):
    """The reference's SparseMaskNetwork under its own configuration (scannet_config/run.py:741-810: input network B16 x 2
    units at stride 1, internal U-Net I -> B32/2 -> B48/2 -> B64/2, Linear 23-32-classes, use_raw_features, no skip
    features), fed by the last decoder level of a sparse U-Net backbone (unet_channels = the decoder's channels).
    The three switches select the other feature-map selectors / the combiner of model.py:597-651."""
    common = dict(use_residuals=True, main_path_relu=False, relu_first=True, bottleneck_divisor=0, drop_input_relu=True,
                  make_dense=False)
    inp = [FLD(type='B', channels=16, params={**common, 'num_units': 2}, anchor_path=None)]
    outd = [FLD('I'), *[FLD(type='B', channels=c, params={**common, 'num_units': 2, 'stride': 2}, anchor_path=None)
                        for c in (32, 48, 64)]]
    unet_params = dict(use_residuals=True, num_units=2, bottleneck_divisor=0, groups=1, main_path_relu=False, relu_first=True,
                       batchnorm=False, concat=True, min_channels=16)
    return SparseMaskNetwork(3, True, cin, None, None, list(backbone_channels[-2::-1]), None, use_unet_features=use_unet_features,
                             use_raw_features=use_raw_features, use_skip_features=use_skip_features, internal_unet=True, unet_params=unet_params,
                             input_network_description=inp, output_network_description=outd,
                             channel_list=[32, num_instance_classes], selection_tuple=(24, 0, True), positive_threshold=0.2)


def _census(net):
    census = {}
    for m in net.modules():
        if type(m).__module__.startswith("sparse_rcnn_amd"):
            census[type(m).__name__] = census.get(type(m).__name__, 0) + 1
    return census


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    mn = build_mask_network()
    sd = mn.state_dict()
    mask = dict(backbone_channels=[32, 64, 128, 256], n_params=int(sum(v.numel() for v in sd.values())),
                keys={k: list(v.shape) for k, v in sd.items()}, census=_census(mn), repr=repr(mn), classes=int(mn.classes))
    print("mask_network", mask["n_params"], mask["census"])
    fixtures = dict(run_config=mask)
    for name, flags in (("unet_only", dict(use_raw_features=False)), ("raw_only", dict(use_unet_features=False)),
                        ("both_skip", dict(use_skip_features=True)), ("raw_skip", dict(use_unet_features=False,
                                                                                       use_skip_features=True))):
        mn = build_mask_network(**flags)
        sd = mn.state_dict()
        fixtures[name] = dict(flags=flags, n_params=int(sum(v.numel() for v in sd.values())),
                              keys={k: list(v.shape) for k, v in sd.items()}, census=_census(mn), repr=repr(mn),
                              classes=int(mn.classes))
        print(name, fixtures[name]["n_params"], fixtures[name]["census"])
    with open(os.path.join(here, "dropin_mask_network.json"), "w") as f:
        json.dump(fixtures, f, indent=0, sort_keys=True)
    out = {}
    for name, ch in (("cfg2_32_256", [32, 64, 128, 256]), ("ref_32_112", [32, 48, 64, 80, 96, 112])):
        fe = build(ch)
        sd = fe.state_dict()
        census = {}
        for m in fe.modules():
            if type(m).__module__.startswith("sparse_rcnn_amd"):
                census[type(m).__name__] = census.get(type(m).__name__, 0) + 1
        out[name] = dict(channels=ch, n_params=int(sum(v.numel() for v in sd.values())),
                         keys={k: list(v.shape) for k, v in sd.items()}, census=census, repr=repr(fe))
        print(name, out[name]["n_params"], census)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin_feature_extractor.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
