"""Developer tool (GPU box): where a k_conv_tbs launch goes -- the library built with -DTBS_EXP=1..4 (tools/build_variant.sh:
1 no weight streaming behind slice 0, 2 no row gathers behind the first D, 3 no barriers in the offset loop, 4 no MFMAs;
results invalid) against the production build and against k_conv_tb (SCN_TB_STREAM=0), levels 1-3 of the cfg-2 scene.
    python tools/ablate_tbs_exp.py  (each variant runs in a fresh process: SCN_MI355X_LIB)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys
sys.path.insert(0, %r)
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import _lib as L, functional as F
from sparse_rcnn_amd.synthetic import make_batch
coords, feats, size, bs, _ = make_batch(1, (512, 512, 256), 150000, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata
md.build_pyramid(size, 4, 3)
sz = tuple(int(s) for s in size)
def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
out = []
for level, C in enumerate([32, 64, 128, 256]):
    rb = md.subm_rulebook(sz, 3)
    if level:
        n, t = rb.n, rb.tiles
        Xb = torch.randn(n, C, device="cuda").to(torch.bfloat16); W = torch.randn(27, C, C, device="cuda") * 0.05
        img = F.pack_weights_bf16(W, C, C, 27, 0)
        out.append("%%6.1f" %% timed(lambda: F.conv_rules_bf16(Xb, t, n, W, None, C, L.F_RELU_IN, image=img)))
    sz = tuple(s // 2 for s in sz)
print(" ".join(out))
''' % ROOT
rows = [("production k_conv_tbs", {}), ("k_conv_tb (SCN_TB_STREAM=0)", {"SCN_TB_STREAM": "0"})]
for e, what in ((2, "no gathers behind the first D"), (4, "no MFMAs")):
    rows.append((f"TBS_EXP={e} {what}", {"SCN_MI355X_LIB": os.path.join(ROOT, "tools", "ab", f"tbs_exp{e}.so")}))
print("us per launch at levels 1 / 2 / 3 (C = 64 / 128 / 256)")
for name, env in rows:
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, **env), capture_output=True, text=True)
    print(f"{name:40s} {r.stdout.strip().splitlines()[-1] if r.returncode == 0 else 'failed: ' + r.stderr[-300:]}", flush=True)
