"""Runs only the correlation forward (and backward if asked) a few times: profiling target."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unflow_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
what = sys.argv[2] if len(sys.argv) > 2 else "fwd"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
fa = torch.randn((B, 256, 48, 160), device=dev, generator=g)
fb = torch.randn((B, 256, 48, 160), device=dev, generator=g)
top = torch.randn((B, 441, 48, 160), device=dev, generator=g)
for _ in range(n):
    if "fwd" in what:
        out = ops.correlation(fa, fb)
    if "bwd" in what:
        g0, g1 = ops.correlation_grad(top, fa, fb)
torch.cuda.synchronize()
