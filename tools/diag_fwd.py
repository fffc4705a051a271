import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from unflow_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
def ref(f0, f1):
    B, C, H, W = f0.shape
    p = torch.nn.functional.pad(f1, (20, 20, 20, 20))
    out = torch.empty((B, 441, H, W), device=dev)
    for iy in range(21):
        for ix in range(21):
            dy, dx = (iy - 10) * 2, (ix - 10) * 2
            out[:, iy * 21 + ix] = (f0 * p[:, :, 20 + dy:20 + dy + H, 20 + dx:20 + dx + W]).mean(1)
    return out
for C in (8, 16, 24, 32, 64, 256):
    f0 = torch.randn((1, C, 48, 160), device=dev, generator=g)
    f1 = torch.randn((1, C, 48, 160), device=dev, generator=g)
    for rep in range(3):
        got = ops.correlation(f0, f1)
        want = ref(f0, f1)
        bad = ((got - want).abs() > 1e-4)
        print("C=%3d rep %d bad %.4f%%" % (C, rep, 100 * bad.float().mean().item()), end="")
        if bad.any():
            yy = bad.any(1)[0]  # [H,W]
            rows = yy.any(1).nonzero().flatten().tolist()
            cols = yy.any(0).nonzero().flatten().tolist()
            print("  rows", rows[:12], "... cols", cols[:24], "planes", bad.any(3).any(2)[0].nonzero().flatten().tolist()[:10])
        else:
            print()
