"""Round 5 probe: scn_nms_bits on 1024 / 2048 / 4096 score-sorted boxes of one scene (HIP events, 100 calls)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_rcnn_amd import proposals as PR
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for n in (1024, 2048, 4096):
    c = torch.rand(1, n, 3, generator=g) * 200
    s = torch.rand(1, n, 3, generator=g) * 40 + 8
    boxes = torch.stack((c - s / 2, c + s / 2), dim=2).to(dev)
    for _ in range(5): k = PR.non_maximum_suppression(boxes, 0.5)
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(100): PR.non_maximum_suppression(boxes, 0.5)
    ev[1].record(); torch.cuda.synchronize()
    print(f"n={n}: {ev[0].elapsed_time(ev[1]) * 10:.1f} us per call, kept {int(k.sum())}", flush=True)
