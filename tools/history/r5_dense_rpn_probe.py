"""Round 5 probe: what the dense RPN stack costs through torch / MIOpen on the stride-8 grid of the cfg-2 scene (64x64x32 cells,
256 channels, 2.3 % of the cells active), fp32 / bf16, cudnn.benchmark on / off, channels_last_3d, and a 1x1 bottleneck first."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
dev = torch.device("cuda", 0)

def stack(kind, cin=256, w=32):
    if kind == "wide":
        return nn.Sequential(nn.Conv3d(cin, w, 3, padding=1), nn.ReLU(True), nn.Conv3d(w, w, 3, padding=2, dilation=2), nn.ReLU(True), nn.Conv3d(w, 28, 1))
    return nn.Sequential(nn.Conv3d(cin, w, 1), nn.ReLU(True), nn.Conv3d(w, w, 3, padding=1), nn.ReLU(True), nn.Conv3d(w, w, 3, padding=2, dilation=2), nn.ReLU(True), nn.Conv3d(w, 28, 1))

def run(kind, dtype, bench, cl):
    torch.backends.cudnn.benchmark = bench
    torch.manual_seed(0)
    net = stack(kind).to(dev)
    x = torch.zeros(1, 256, 64, 64, 32, device=dev)
    idx = torch.randperm(64 * 64 * 32, device=dev)[:2983]
    x.view(1, 256, -1)[0][:, idx] = torch.randn(256, 2983, device=dev)
    if cl:
        net = net.to(memory_format=torch.channels_last_3d)
        x = x.contiguous(memory_format=torch.channels_last_3d)
    x.requires_grad_()
    def step():
        if dtype == "bf16":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = net(x)
        else:
            y = net(x)
        return y
    for _ in range(3):
        y = step(); y.float().sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        y = step()
    torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / 10 * 1e3
    t0 = time.perf_counter()
    for _ in range(10):
        y = step(); y.float().sum().backward()
    torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{kind:10s} {dtype:5s} benchmark={bench!s:5s} channels_last={cl!s:5s}  fwd {tf:7.2f} ms   fwd+bwd {tb:7.2f} ms", flush=True)

for kind in ("wide", "bottleneck"):
    for dtype in ("f32", "bf16"):
        for bench in (False, True):
            for cl in (False, True):
                try:
                    run(kind, dtype, bench, cl)
                except Exception as e:
                    print(kind, dtype, bench, cl, "FAILED", repr(e)[:200], flush=True)
