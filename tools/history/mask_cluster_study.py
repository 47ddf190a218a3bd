"""Developer tool (CPU, oracle index structures): what a better grouping of rows into 16-row tiles could buy over the
mask-key radix sort of scn_tiles.hip -- executed/useful matrix work of the cfg-2 scene under (a) the current key order, (b) a
top-down partition on the most uncertain bit, (c) Morton order, (d) a greedy union-growth clustering of the leftover rows
(O(n^2): an upper bound on clustering quality, not a candidate).  DESIGN.md 4.1.    python tools/mask_cluster_study.py"""
import sys, numpy as np, time
sys.path.insert(0,'/root/repo')
from sparse_rcnn_amd.synthetic import make_batch
from oracle import scn_oracle as O
coords, feats, size, bs, _ = make_batch(1, (512,512,256), 150000, dup=1.15, seed=1)
scene = O.OracleScene(coords.numpy())
def pos_perm():
    pos = np.zeros(27, int); nxt = 0
    for cls in range(4):
        for o in range(27):
            dx, dy, dz = o//9-1, (o//3)%3-1, o%3-1
            if (dx!=0)+(dy!=0)+(dz!=0) == cls: pos[o] = nxt; nxt += 1
    return pos
PC = np.array([bin(i).count('1') for i in range(1<<16)])
def popc(a):
    a = a.astype(np.uint64); return PC[a & 0xffff] + PC[(a>>16) & 0xffff]
def waste_of_order(masks_sorted):
    n = len(masks_sorted); nt = (n + 15)//16
    pad = np.zeros(nt*16, np.uint32); pad[:n] = masks_sorted
    tm = np.bitwise_or.reduce(pad.reshape(nt,16), axis=1)
    return popc(tm).sum()*16
def adaptive(masks, idx, out, leaf=16, used=0):
    # recursive partition: split on the most uncertain unused bit
    if len(idx) <= leaf:
        out.append(idx); return
    m = masks[idx]
    best, bb = None, None
    for b in range(27):
        if used >> b & 1: continue
        f = ((m >> b) & 1).mean()
        s = abs(f - 0.5)
        if f == 0 or f == 1: continue
        if best is None or s < best: best, bb = s, b
    if bb is None:
        out.append(idx); return
    bit = ((m >> bb) & 1).astype(bool)
    adaptive(masks, idx[bit], out, leaf, used | 1 << bb)
    adaptive(masks, idx[~bit], out, leaf, used | 1 << bb)
sys.setrecursionlimit(10000)
for level in (0, 1, 2, 3):
    if level: scene.strided_rules(level-1)
    nbr, rules = O.subm_rulebook(scene.level_coords[level], 3)
    n = nbr.shape[1]
    have = (nbr >= 0)
    P = have.sum()
    nat = (have * (1 << np.arange(27))[:, None]).sum(0).astype(np.uint32)
    pos = pos_perm()
    key = (have * (1 << pos)[:, None]).sum(0).astype(np.uint32)
    ex = waste_of_order(nat[np.argsort(key, kind='stable')])
    print(f"L{level} n={n} P={P} current key: {ex/P:.3f}", flush=True)
    t=time.time()
    out=[]; adaptive(nat, np.arange(n), out)
    order = np.concatenate(out)
    ex = waste_of_order(nat[order])
    print(f"   adaptive tree order (concatenated leaves): {ex/P:.3f}  ({time.time()-t:.1f}s, {len(out)} leaves)", flush=True)
    # morton order
    c = scene.level_coords[level][:, :3].astype(np.int64)
    def part(v):
        r = np.zeros_like(v)
        for b in range(10): r |= ((v >> b) & 1) << (3*b)
        return r
    mort = part(c[:,0]) | part(c[:,1])<<1 | part(c[:,2])<<2
    ex = waste_of_order(nat[np.argsort(mort, kind='stable')])
    print(f"   morton order: {ex/P:.3f}")
print("---- greedy union-growth clustering (upper bound on what clustering can do)")
scene = O.OracleScene(coords.numpy())
for level in (0, 1, 2, 3):
    if level: scene.strided_rules(level-1)
    nbr, rules = O.subm_rulebook(scene.level_coords[level], 3)
    have = (nbr >= 0); P = have.sum(); n = nbr.shape[1]
    nat = (have * (1 << np.arange(27))[:, None]).sum(0).astype(np.uint32)
    u, cnt = np.unique(nat, return_counts=True)
    upc = popc(u)
    executed = ((cnt // 16) * 16 * upc).sum()
    left = (cnt % 16).astype(np.int64)
    t = time.time()
    ntiles = 0
    while left.sum() > 0:
        # seed: the remaining mask with the most bits
        alive = left > 0
        cand = np.where(alive)[0]
        s = cand[np.argmax(upc[cand])]
        union = u[s]; room = 16
        take = min(room, left[s]); left[s] -= take; room -= take
        while room > 0 and left.sum() > 0:
            alive = left > 0
            cand = np.where(alive)[0]
            add = popc(u[cand] & ~union)
            # fewest new bits; tie: most bits in common
            j = cand[np.lexsort((-upc[cand], add))[0]]
            union |= u[j]
            take = min(room, left[j]); left[j] -= take; room -= take
        executed += 16 * popc(np.array([union]))[0]
        ntiles += 1
    print(f"L{level}: greedy clustering executed/useful {executed/P:.3f} ({time.time()-t:.0f}s)", flush=True)
print("---- sorting only the upper key bits (a cheaper sort): executed/useful")
scene = O.OracleScene(coords.numpy())
for level in (0, 1, 2, 3):
    if level: scene.strided_rules(level-1)
    nbr, rules = O.subm_rulebook(scene.level_coords[level], 3)
    have = (nbr >= 0); P = have.sum()
    nat = (have * (1 << np.arange(27))[:, None]).sum(0).astype(np.uint32)
    key = (have * (1 << pos_perm())[:, None]).sum(0).astype(np.uint32)
    for lo in (0, 3, 7, 11, 15, 19):
        ex = waste_of_order(nat[np.argsort(key >> lo, kind='stable')])
        print(f"L{level} sort bits [{lo},27): {ex/P:.3f}")
