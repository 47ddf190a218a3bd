"""Round 5 probe: torch.topk of 524 288 scores, k = 1024 (the RPN's pre-NMS selection) -- one call vs two stages."""
import torch, time
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.sigmoid(torch.randn(1, 524288, generator=g) * 2).to(dev)
x[0, ::3] = 0.5                                   # a big tie class below the top, like the empty cells of the dense volume
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6, r
def one(): return torch.topk(x, 1024, dim=1, sorted=True)
def two(G):
    def f():
        v, i = torch.topk(x.view(1, G, -1), min(1024, x.shape[1] // G), dim=2, sorted=False)
        i = i + (torch.arange(G, device=dev) * (x.shape[1] // G)).view(1, G, 1)
        v2, j = torch.topk(v.reshape(1, -1), 1024, dim=1, sorted=True)
        return v2, torch.gather(i.reshape(1, -1), 1, j)
    return f
us, (v, i) = t(one); print(f"one call            {us:7.1f} us")
for G in (16, 64, 256, 512):
    us, (v2, i2) = t(two(G)); print(f"two stages, G={G:4d}  {us:7.1f} us  values equal {bool(torch.equal(v, v2))} indices equal {bool(torch.equal(i, i2))}")
us, _ = t(lambda: torch.sort(x, dim=1, descending=True)); print(f"full sort           {us:7.1f} us")
# round 5, later: the radix select of this library (scn_topk_boxes) against torch.topk + the advanced-indexing gather
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_rcnn_amd import proposals as PR
boxes = torch.randn(1, 524288, 2, 3, generator=g).to(dev)
def ours(): return PR.topk_boxes(x, boxes, 1024)
def theirs():
    v, i = torch.topk(x, 1024, dim=1, sorted=True)
    return v, i, boxes[torch.arange(1, device=dev).unsqueeze(1), i]
us, r = t(theirs); print(f"torch.topk + gather {us:7.1f} us")
us, r2 = t(ours); print(f"scn_topk_boxes      {us:7.1f} us   values equal {bool(torch.equal(r[0], r2[0]))}")
y = torch.sigmoid(torch.randn(1, 524288, generator=g) * 2 - 3).to(dev)          # no tie class
x = y
us, r = t(theirs); print(f"no tie class: torch {us:7.1f} us")
us, r2 = t(ours); print(f"no tie class: ours  {us:7.1f} us   values equal {bool(torch.equal(r[0], r2[0]))} indices equal {bool(torch.equal(r[1], r2[1]))}")
