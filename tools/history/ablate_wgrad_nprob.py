"""Developer tool (GPU box): the weight gradients of 2 x 2 vs 1 x 4 (and 1 + 3) problems of a level in bf16 / fp32 storage --
would two residual units' weight gradients as ONE launch pay where the launches are short (bf16)?
    python tools/ablate_wgrad_nprob.py [voxels=150000] [grid=512] [f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_rcnn_amd as scn
from sparse_rcnn_amd import functional as F, _lib as L
from sparse_rcnn_amd.synthetic import make_batch
vox = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
g = int(sys.argv[2]) if len(sys.argv) > 2 else 512
FP32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
coords, feats, size, bs, _ = make_batch(1, (g, g, g // 2), vox, seed=1)
x = scn.InputLayer(3, size, mode=4)((coords, feats.cuda(), 1))
md = x.metadata
sz = tuple(int(s) for s in size)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000 / reps
for level, C in enumerate((32, 64, 128, 256)):
    rb = md.subm_rulebook(sz, 3)
    r = rb.rules
    mk = lambda: (torch.randn(rb.n, C, device="cuda") if FP32 else torch.randn(rb.n, C, device="cuda").bfloat16())
    Xs, dYs = [mk() for _ in range(4)], [mk() for _ in range(4)]
    db = 1 << 13
    t22 = timeit(lambda: (F.wgrad_bias_rules_n(Xs[:2], dYs[:2], r.in_rows, r.out_rows, r.prefix_host, 27, db, L.F_RELU_IN),
                          F.wgrad_bias_rules_n(Xs[2:], dYs[2:], r.in_rows, r.out_rows, r.prefix_host, 27, db, L.F_RELU_IN)))
    t4 = timeit(lambda: F.wgrad_bias_rules_n(Xs, dYs, r.in_rows, r.out_rows, r.prefix_host, 27, db, L.F_RELU_IN))
    t3 = timeit(lambda: F.wgrad_bias_rules_n(Xs[:3], dYs[:3], r.in_rows, r.out_rows, r.prefix_host, 27, db, L.F_RELU_IN))
    t1 = timeit(lambda: F.wgrad_bias_rules(Xs[0], dYs[0], r.in_rows, r.out_rows, r.prefix_host, 27, db, L.F_RELU_IN))
    print(f"level {level} N={rb.n} C={C}: 2+2 problems {t22:6.1f} us   4 in one {t4:6.1f} us   3 in one {t3:6.1f}   1: {t1:6.1f}  (launch + sum each)", flush=True)
    md.strided_rulebook(sz); sz = tuple(s // 2 for s in sz)
