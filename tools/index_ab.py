"""Developer tool (GPU box): scn_pyramid_build alone on an idle GPU -- two queues (default) against one (SCN_PYRAMID_ONE_STREAM=1),
alternating, cfg-2 scene (4 levels) and the reference plan's 6 levels; bit-equality of every structure of the two forms.
    python tools/index_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse_rcnn_amd._lib import switches as _SW      # library switches: scn_debug_set (the environment is read once at load)
from sparse_rcnn_amd.metadata import Metadata
from sparse_rcnn_amd.synthetic import make_batch

coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, dup=1.15, seed=1)
cd = coords.cuda()


def build(levels):
    return Metadata(3).build_native(size, cd, 1, 4, levels, 3)


def timed(levels, reps=20):
    for _ in range(3):
        build(levels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        build(levels)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for levels in (4, 6):
    a = build(levels)
    _SW["SCN_PYRAMID_ONE_STREAM"] = "1"
    b = build(levels)
    del _SW["SCN_PYRAMID_ONE_STREAM"]
    torch.cuda.synchronize()
    def det(md):        # the deterministic structures (the hash tables' slot layout depends on the insertion race)
        out = [md.item_row, md.row_count, md.row_first, md.point_coords] + [g.coords for g in md.grids.values()]
        for rb in md.subm.values():
            out += [rb.table, rb.rules.in_rows, rb.rules.out_rows, rb.rules.prefix_dev, rb.tiles.perm, rb.tiles.tstab,
                    rb.tiles.tile_mask, rb.tiles.tile_order]
        for sb in md.strided.values():
            out += [sb.parent, sb.fine_off, sb.child, sb.rules.in_rows, sb.rules.out_rows, sb.rules.prefix_dev, sb.tiles.perm,
                    sb.tiles.tstab, sb.tiles.tile_mask, sb.tiles.tile_order]
        return out
    same = all(torch.equal(x, y) for x, y in zip(det(a), det(b)))
    n_t = len(det(a))
    res = []
    for rep in range(3):
        two = timed(levels)
        _SW["SCN_PYRAMID_ONE_STREAM"] = "1"
        one = timed(levels)
        del _SW["SCN_PYRAMID_ONE_STREAM"]
        res.append((two, one))
    print(f"levels={levels}: {n_t} index tensors identical: {same};  ms per build (two queues / one): " +
          "  ".join(f"{t:.3f}/{o:.3f}" for t, o in res))
