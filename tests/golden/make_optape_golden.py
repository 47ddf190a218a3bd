"""Op-tape fixtures (tests/optape.py): in the build container only, the REFERENCE's
`FeatureExtractor.forward` (ndsis/modules/model.py:414-446) and `SparseMaskNetwork.forward` (:758-782, eval mode: the given
boxes are the selected ones) run on this package with the surface's arithmetic replaced by shapes (optape.shape_stub), and
every leaf operator call is recorded: layer signature, operand wiring, spatial size, row and channel counts.  The tapes are
data -- a list of small dicts per forward -- not source text; tests/test_gpu_exec.py replays the same seeded scene through
unet.DropinBackbone / maskhead.MaskBranch on the GPU and compares.

    python tests/golden/make_optape_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
import torch                                              # noqa: E402
import make_dropin_golden as G                            # noqa: E402  (registers this package as `sparseconvnet`)
import optape                                             # noqa: E402
from sparse_rcnn_amd.synthetic import make_batch, make_boxes   # noqa: E402

SCENE = dict(n_samples=2, grid=(64, 64, 32), target=3000, dup=1.15, seed=3, n_boxes=6, box_seed=5)


def scene():
    coords, feats, size, bs, splits = make_batch(SCENE["n_samples"], SCENE["grid"], SCENE["target"], dup=SCENE["dup"],
                                                 seed=SCENE["seed"])
    boxes = make_boxes(coords, SCENE["n_boxes"], seed=SCENE["box_seed"])
    return (coords, feats, size, bs, splits), boxes


if __name__ == "__main__":
    data, boxes = scene()
    out = dict(scene=dict(SCENE, grid=list(SCENE["grid"])))
    with optape.shape_stub():
        for name, ch in (("cfg2_32_256", [32, 64, 128, 256]), ("ref_32_112", [32, 48, 64, 80, 96, 112])):
            fe = G.build(ch)
            with optape.record() as tape:
                scene_size, batch_size, anchor_outputs, class_output, mask_outputs, unet_output = fe(data)
            out["feature_extractor_" + name] = tape.entries
            print(name, len(tape.entries), "leaf calls; unet output", tuple(unet_output[-1].features.shape))
        # the mask network on the (stub) output of the 32-256 backbone's decoder
        fe = G.build([32, 64, 128, 256])
        *_, unet_output = fe(data)
        mn = G.build_mask_network().eval()
        with optape.record() as tape:
            logits, selection, _ = mn(data, None, unet_output, boxes, None)
        out["mask_network"] = tape.entries
        out["mask_logits_shape"] = list(logits.shape)
        print("mask_network", len(tape.entries), "leaf calls; logits", tuple(logits.shape))
    with open(os.path.join(HERE, "optape_reference_forward.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
