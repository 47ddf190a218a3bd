"""Developer tool (CPU): executed/useful matrix work of 16-row tiles under alternative row orderings, from the oracle's
neighbour table of the cfg-2 scene."""
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
def waste_of_order(masks_sorted):
    n = len(masks_sorted); nt = (n + 15)//16
    pad = np.zeros(nt*16, np.uint32); pad[:n] = masks_sorted
    tm = np.bitwise_or.reduce(pad.reshape(nt,16), axis=1)
    pc = lambda a: np.array([bin(int(v)).count('1') for v in a])
    return pc(tm).sum()*16, nt
for level in (0, 1):
    if level: scene.strided_rules(level-1)
    nbr, rules = O.subm_rulebook(scene.level_coords[level], 3)
    n = nbr.shape[1]
    have = (nbr >= 0)
    P = have.sum()
    nat = (have * (1 << np.arange(27))[:, None]).sum(0).astype(np.uint32)
    pos = pos_perm()
    key = (have * (1 << pos)[:, None]).sum(0).astype(np.uint32)
    popc = have.sum(0)
    ex, nt = waste_of_order(nat[np.argsort(key, kind='stable')])
    print(f"L{level} n={n} P={P} current key: executed/useful {ex/P:.3f}")
    # distinct masks
    u, inv, cnt = np.unique(nat, return_inverse=True, return_counts=True)
    print(f"   distinct masks {len(u)}; rows in groups >=16: {cnt[cnt>=16].sum()/n:.3f}; groups<16: {np.sum(cnt<16)} with {cnt[cnt<16].sum()} rows")
    # pure-group tiling: full tiles from pure groups + leftovers (size%16) sorted by key into mixed tiles
    upc = np.array([bin(int(v)).count('1') for v in u])
    full_exec = ((cnt//16)*16*upc).sum()
    left_masks = np.repeat(u, cnt % 16); left_keys = np.repeat((np.array([sum(((int(m)>>o)&1)<<pos[o] for o in range(27)) for m in u])), cnt % 16)
    exl, ntl = waste_of_order(left_masks[np.argsort(left_keys, kind='stable')])
    print(f"   hybrid (pure full tiles + leftovers in key order): executed/useful {(full_exec+exl)/P:.3f}  (leftover rows {len(left_masks)}, {ntl} tiles, their ratio {exl/max(1,(np.repeat(upc,cnt%16)).sum()):.3f})")
    # alternative keys: (popcount, key)
    k2 = popc.astype(np.int64)*(1<<27) + key
    ex2,_ = waste_of_order(nat[np.argsort(k2, kind='stable')]); print(f"   key (popcount, mask): {ex2/P:.3f}")
