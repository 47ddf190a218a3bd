"""Developer tool (CPU): list-scheduling simulation of k_conv_ts's tile queue on the cfg-2 scene.
Cost model: a tile costs popcount(tile_mask) + C0 offset steps; a workgroup owns every n_tg-th entry of tile_order."""
import sys, os, heapq
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import scn_oracle as O
from sparse_rcnn_amd.synthetic import make_batch

C0 = float(os.environ.get("C0", 3))
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150_000, dup=1.15, seed=1)
scene = O.OracleScene(coords.numpy())

def keybits():
    pos = np.zeros(27, dtype=np.int64); nxt = 0
    for cls in range(4):
        for o in range(27):
            dx, dy, dz = o // 9 - 1, (o // 3) % 3 - 1, o % 3 - 1
            if (dx != 0) + (dy != 0) + (dz != 0) == cls:
                pos[o] = nxt; nxt += 1
    return pos

def simulate(cost, n_tg, waves, split=None):
    order = np.argsort(-cost, kind="stable")
    worst = 0.0; tot = 0.0
    for tg in range(n_tg):
        mine = cost[order[tg::n_tg]]
        items = []
        for c in mine:
            if split and c - C0 > split:
                h = (c - C0) / 2
                items += [np.ceil(h) + C0, np.floor(h) + C0]
            else:
                items.append(c)
        heap = [0.0] * waves
        for c in items:
            t = heapq.heappop(heap); heapq.heappush(heap, t + c)
        worst = max(worst, max(heap)); tot += sum(items)
    return worst, tot / (n_tg * waves)

pos = keybits()
for lvl, ch in enumerate((32, 64, 128, 256)):
    if lvl:
        scene.strided_rules(lvl - 1)
    nbr, _ = O.subm_rulebook(scene.level_coords[lvl], 3)
    n = nbr.shape[1]
    key = np.zeros(n, dtype=np.int64); mask = np.zeros(n, dtype=np.int64)
    for o in range(27):
        have = (nbr[o] >= 0).astype(np.int64)
        key |= have << pos[o]; mask |= have << o
    order = np.argsort(key, kind="stable")
    nt = (n + 15) // 16
    m = np.zeros(nt * 16, dtype=np.int64); m[:n] = mask[order]
    tm = np.bitwise_or.reduce(m.reshape(nt, 16), axis=1)
    pc = np.array([bin(int(x)).count("1") for x in tm], dtype=np.float64)
    useful = (nbr >= 0).sum() / 16.0
    n_chunks, n_kc = ch // 32, ch // 32
    n_tg = max(1, min(256 // (n_chunks * n_kc), (nt + 15) // 16))
    cost = pc + C0
    line = f"L{lvl} N={n} nt={nt} exec/useful={pc.sum()/useful:.3f} n_tg={n_tg} max_tile={pc.max():.0f}"
    for waves, split in ((16, None), (8, None), (16, 14), (16, 9), (16, 7)):
        worst, avg = simulate(cost, n_tg, waves, split)
        ideal = cost.sum() / (n_tg * waves)
        line += f" | w{waves} s{split}: eff {ideal/worst:.2f}"
    print(line, flush=True)
