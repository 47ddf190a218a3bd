"""Developer tool (GPU box): per-wave timeline of k_conv_ts (wall_clock64 stamps, 100 MHz).  Needs a library built with
-DTS_TIMELINE=1 for scn_conv_ts.hip, e.g.
    cd sparse_rcnn_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTS_TIMELINE=1 -c scn_conv_ts.hip -o /tmp/ts_tl.o \\
      && hipcc --offload-arch=gfx950 -fPIC -shared -o ../../tools/libscn_timeline.so build/scn_index.hip.o build/scn_conv.hip.o \\
         /tmp/ts_tl.o build/scn_conv_ts_bf16.hip.o build/scn_tiles.hip.o build/scn_pyramid.hip.o build/scn_wgrad.hip.o build/scn_elem.hip.o
    SCN_MI355X_LIB=$PWD/tools/libscn_timeline.so python tools/ts_timeline.py 0"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L
from sparse_rcnn_amd.synthetic import make_batch
level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
C = [32, 64, 128, 256][level]
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata; sz = tuple(int(s) for s in size)
for l in range(level):
    md.strided_rulebook(sz); sz = tuple(s // 2 for s in sz)
rb = md.subm_rulebook(sz, 3); n, t = rb.n, rb.tiles
X = torch.randn(n, C, device="cuda"); W = torch.randn(27, C, C, device="cuda") * 0.05
Y = torch.empty(n, C, device="cuda")
lib = L.lib()
SCR = torch.zeros(lib.scn_conv_tiles_scratch_bytes(C, n, C) + (4 << 20), dtype=torch.uint8, device="cuda")
def run():
    L.check(lib.scn_conv_tiles(L.ptr(X), n, C, L.ptr(t.tstab), L.ptr(t.tile_mask), L.ptr(t.perm), L.ptr(t.tile_order), 27, n, L.ptr(W), 0, 0, 0,
                               L.ptr(Y), C, 0, L.ptr(SCR), 0, L.stream()))
for _ in range(int(os.environ.get('TL_LAUNCHES', '4000'))): run()      # sustained load: the clock settles (DVFS)
torch.cuda.synchronize()
base = lib.scn_conv_tiles_scratch_bytes(C, n, C)
d = SCR[base:base + 8192 * 64].view(torch.int64).cpu().numpy().reshape(-1, 8)
d = d[d[:, 2] > 0]
t0 = d[:, 0].min()
us = lambda v: v / 100.0
print(f"waves {len(d)}  kernel span {us(d[:,2].max() - t0):.1f} us")
print(f"start  (first..last wave)     : {us(d[:,0].min()-t0):.1f} .. {us(d[:,0].max()-t0):.1f} us")
print(f"staged (t1 - t0) mean / max    : {us((d[:,1]-d[:,0]).mean()):.1f} / {us((d[:,1]-d[:,0]).max()):.1f} us")
work = d[:, 2] - d[:, 1]
print(f"tile phase (end - staged) mean / min / max : {us(work.mean()):.1f} / {us(work.min()):.1f} / {us(work.max()):.1f} us")
print(f"wave end (first..last)        : {us(d[:,2].min()-t0):.1f} .. {us(d[:,2].max()-t0):.1f} us")
print(f"tiles per wave mean {d[:,3].mean():.2f}  steps per wave mean {d[:,4].mean():.1f} max {d[:,4].max()}  epilogue per wave mean {us(d[:,5].mean()):.2f} us")
ghz = (d[:, 7] - d[:, 6]) / np.maximum(1, d[:, 2] - d[:, 1]) * 0.1
print(f"in-kernel shader clock over the tile phase (clock64 / wall_clock64): median {np.median(ghz):.2f} GHz "
      f"(p10 {np.percentile(ghz, 10):.2f}, p90 {np.percentile(ghz, 90):.2f})")
steps = d[:, 4].astype(float)
ok = steps > 0
print(f"us per step (tile phase / steps): mean {us((work[ok]/steps[ok]).mean()):.2f}   corr(steps, time) = {np.corrcoef(steps[ok], work[ok])[0,1]:.2f}")
# least squares: time = a * steps + b * tiles + c
A = np.stack([steps[ok], d[ok, 3].astype(float), np.ones(ok.sum())], 1)
coef, *_ = np.linalg.lstsq(A, work[ok].astype(float), rcond=None)
print(f"fit: time = {us(coef[0]):.3f} us/step + {us(coef[1]):.2f} us/tile + {us(coef[2]):.2f} us")
# per-workgroup view: d rows are ordered (blockIdx.x, wave); waves of a workgroup pull from one queue and end together
full = SCR[base:base + 8192 * 64].view(torch.int64).cpu().numpy().reshape(-1, 8)
nwg = len(full) // 16
wg = full[:nwg * 16].reshape(nwg, 16, 8)
live = wg[:, :, 2].max(1) > 0
wg = wg[live]
wg_end = us(wg[:, :, 2].max(1) - t0); wg_steps = wg[:, :, 4].sum(1); wg_tiles = wg[:, :, 3].sum(1)
print(f"workgroups {len(wg)}: end {wg_end.min():.1f} .. {wg_end.max():.1f} us (mean {wg_end.mean():.1f}); steps per WG "
      f"{wg_steps.min()} .. {wg_steps.max()} (mean {wg_steps.mean():.0f}); tiles per WG {wg_tiles.min()} .. {wg_tiles.max()}")
print(f"corr(WG steps, WG end) = {np.corrcoef(wg_steps, wg_end)[0, 1]:.2f}")
ids = np.nonzero(live)[0]
for x in range(8):
    sel = ids % 8 == x
    print(f"  blockIdx % 8 == {x}: end mean {wg_end[sel].mean():.1f} us  min {wg_end[sel].min():.1f}  max {wg_end[sel].max():.1f}  steps mean {wg_steps[sel].mean():.0f}")
# where the wave-time of the launch goes (fractions of waves x kernel span)
span = float(d[:, 2].max() - t0)
tot = span * len(d)
print(f"wave-time split: before start {(d[:,0]-t0).sum()/tot:.3f}  staging {(d[:,1]-d[:,0]).sum()/tot:.3f}  "
      f"tile phase {(d[:,2]-d[:,1]).sum()/tot:.3f}  idle after own end {(d[:,2].max()-d[:,2]).sum()/tot:.3f}")
wend = wg[:, :, 2].astype(float)
print(f"  idle after own end, split: inside the workgroup (wave end -> WG end) {((wend.max(1, keepdims=True)-wend).sum())/tot:.3f}  "
      f"across workgroups (WG end -> kernel end) {((wend.max()-wend.max(1)).sum()*16)/tot:.3f}")
