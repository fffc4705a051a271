"""Repeat the correlation forward / backward on fixed inputs and compare the results bitwise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unflow_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(2)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f0 = torch.randn((B, 256, 48, 160), device=dev, generator=g)
f1 = torch.randn((B, 256, 48, 160), device=dev, generator=g)
top = torch.randn((B, 441, 48, 160), device=dev, generator=g)
c0 = ops.correlation(f0, f1)
g0, g1 = ops.correlation_grad(top, f0, f1)
bad = [0, 0, 0]
for i in range(n):
    # churn the allocator / caches between repetitions
    junk = torch.full((32 * 1024 * 1024,), float(i) + 1e30, device=dev); del junk
    c = ops.correlation(f0, f1)
    a, b = ops.correlation_grad(top, f0, f1)
    for k, (x, y) in enumerate(((c, c0), (a, g0), (b, g1))):
        if not torch.equal(x, y):
            bad[k] += 1
            d = (x - y).abs()
            idx = torch.nonzero(d > 0)
            print("rep %d tensor %d: %d elements differ, max %.3e, first idx %s" % (i, k, idx.shape[0], float(d.max()), idx[:4].tolist()))
print("mismatching repetitions (fwd, g0, g1):", bad, "of", n)
