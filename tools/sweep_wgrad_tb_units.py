"""Developer tool (GPU box): unit count of the bf16-MFMA weight gradient (k_wgrad_tb; SWEEP_F32=1: k_wgrad_direct on fp32 rows), paired launch (two problems), per level:
SCN_WGRAD_SPLITS sweeps the target number of units (each unit writes a cin x cout fp32 block to the slabs).
    python tools/sweep_wgrad_tb_units.py [voxels=150000] [grid=512]"""
import os, sys, subprocess
if len(sys.argv) > 3 and sys.argv[3] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import sparse_rcnn_amd as scn
    from sparse_rcnn_amd import functional as F, _lib as L
    from sparse_rcnn_amd.synthetic import make_batch
    vox, g = int(sys.argv[1]), int(sys.argv[2])
    coords, feats, size, bs, _ = make_batch(1, (g, g, g // 2), vox, seed=1)
    x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
    md = x.metadata
    sz = tuple(int(s) for s in size)
    out = []
    for level, C in enumerate((32, 64, 128, 256)):
        rb = md.subm_rulebook(sz, 3); r = rb.rules
        mk = (lambda: torch.randn(rb.n, C, device="cuda")) if os.environ.get("SWEEP_F32") else (lambda: torch.randn(rb.n, C, device="cuda").bfloat16())
        Xs, dYs = [mk() for _ in range(2)], [mk() for _ in range(2)]
        fn = lambda: F.wgrad_bias_rules_n(Xs, dYs, r.in_rows, r.out_rows, r.prefix_host, 27, 1 << 13, L.F_RELU_IN)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        out.append(f"{s.elapsed_time(e) * 50:6.1f}")
        md.strided_rulebook(sz); sz = tuple(s // 2 for s in sz)
    print(" ".join(out))
    sys.exit(0)
vox = sys.argv[1] if len(sys.argv) > 1 else "150000"
g = sys.argv[2] if len(sys.argv) > 2 else "512"
print("units target : us per paired launch + sum at levels 0 1 2 3")
for t in (os.environ.get("SWEEP", "default,96,160,224,320,448,640,970,1500,2000").split(",")):
    env = dict(os.environ)
    if t != "default": env["SCN_WGRAD_SPLITS"] = t
    r = subprocess.run([sys.executable, __file__, vox, g, "child"], env=env, capture_output=True, text=True)
    print(f"{t:>8s} : {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]}", flush=True)
