"""Round 6 (GPU box), experiment (c): the TILE-major, one-gather-per-rule weight gradient (scn_wgrad_tiles32) against the
rule-major product kernel (scn_wgrad_rules) on the C = 32 level of the cfg-2 scene (and of other scene sizes): agreement and time.
    python tools/r6_wgrad_tiles_probe.py [voxels ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.synthetic import make_batch

lib = L.lib()


def timeit(run, n=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / n


for vox in ([int(v) for v in sys.argv[1:]] or [150000, 600000]):
    g = 512 if vox <= 200000 else 1024
    coords, feats, size, bs, _ = make_batch(1, (g, g, g // 2), vox, seed=1)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
    md = x.metadata
    rb = md.subm_rulebook(tuple(int(s) for s in size), 3)
    n, r, t = rb.n, rb.rules, rb.tiles
    gen = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn(n, 32, device="cuda", generator=gen); dY = torch.randn(n, 32, device="cuda", generator=gen)
    ph = r.prefix_host
    for relu in (0, 1):
        dW0 = torch.empty(27, 32, 32, device="cuda"); dW1 = torch.full((27, 32, 32), float("nan"), device="cuda")
        s0 = torch.empty(lib.scn_wgrad_scratch_bytes(32, 32, ph, 27), dtype=torch.uint8, device="cuda")
        s1 = torch.empty(lib.scn_wgrad_tiles32_scratch_bytes(), dtype=torch.uint8, device="cuda")
        def run0():
            L.check(lib.scn_wgrad_rules(L.ptr(X), 32, L.ptr(dY), 32, L.ptr(r.in_rows), L.ptr(r.out_rows), ph, 27, L.ptr(dW0), L.ptr(s0),
                                        relu, L.stream()))
        def run1():
            L.check(lib.scn_wgrad_tiles32(L.ptr(X), n, L.ptr(dY), n, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), ph, L.ptr(dW1),
                                          L.ptr(s1), relu, L.stream()))
        run0(); run1(); torch.cuda.synchronize()
        Xr = X.clamp_min(0) if relu else X
        ref = torch.zeros(27, 32, 32, device="cuda", dtype=torch.float64)
        for o in range(27):
            a, b = int(ph[o]), int(ph[o + 1])
            if b > a:
                ref[o] = Xr[r.in_rows[a:b].long()].double().t() @ dY[r.out_rows[a:b].long()].double()
        sc = ref.abs().max().item()
        e0, e1 = (dW0.double() - ref).abs().max().item() / sc, (dW1.double() - ref).abs().max().item() / sc
        dW1b = dW1.clone(); run1(); torch.cuda.synchronize()
        t0, t1 = timeit(run0), timeit(run1)
        P = int(ph[27])
        print(f"N={n:7d} P={P:8d} relu_in={relu}: rule-major {t0:6.1f} us ({2.0 * P * 1024 / t0 / 1e6:5.1f} TF)  err {e0:.1e}   "
              f"tile-major {t1:6.1f} us ({2.0 * P * 1024 / t1 / 1e6:5.1f} TF)  err {e1:.1e}  repeatable {bool(torch.equal(dW1, dW1b))}", flush=True)

# ---- where the tile-major form loses: the lockstep imbalance, from the tile masks (last scene above) ---------------------------
import numpy as np
tm = t.tile_mask.cpu().numpy().view(np.uint32)
nt = len(tm)
w = np.array([float(ph[o + 1] - ph[o]) for o in range(27)] + [0.0]); w[27] = w[13] / 2; w[13] /= 2
order = sorted(range(28), key=lambda v: -w[v])
own, load = [[] for _ in range(16)], [0.0] * 16
for v in order:                                                   # the host's deal: heaviest first onto the lightest wave with a free set
    k = min((i for i in range(16) if len(own[i]) < 2), key=lambda i: load[i])
    own[k].append(v); load[k] += w[v]
n_wg, R = 256, 8
tot = lock = worst_wave = 0.0
for b in range(n_wg):
    mine = tm[b::n_wg]
    per_wave_total = np.zeros(16)
    for r0 in range(0, len(mine), R):
        rnd = mine[r0:r0 + R]
        items = np.zeros(16)
        for k in range(16):
            for v in own[k]:
                o = 13 if v == 27 else v
                has = (rnd >> o) & 1
                if v == 13: has = has * (np.arange(len(rnd)) % 2 == 0)
                if v == 27: has = has * (np.arange(len(rnd)) % 2 == 1)
                items[k] += has.sum()
        tot += items.sum(); lock += items.max() * 16; per_wave_total += items
    worst_wave += per_wave_total.max() * 16
print(f"items (tile, offset) {tot:.0f}; lockstep rounds of {R}: sum over rounds of the slowest wave x 16 = {lock:.0f} ({lock / tot:.2f} x);"
      f" without rounds, slowest wave of a workgroup x 16 = {worst_wave:.0f} ({worst_wave / tot:.2f} x)")
