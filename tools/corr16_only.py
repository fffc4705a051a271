"""Runs only the f16 matrix-core correlation (forward and / or backward) a few times: profiling target.
usage: corr16_only.py [cfg5|cfg3] [fwd|bwd|fwdbwd] [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unflow_amd import ops
shape = (32, 256, 48, 160) if (len(sys.argv) > 1 and sys.argv[1] == "cfg3") else (16, 256, 96, 160)
what = sys.argv[2] if len(sys.argv) > 2 else "fwd"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
fa = torch.randn(shape, device=dev, generator=g).half()
fb = torch.randn(shape, device=dev, generator=g).half()
top = torch.randn((shape[0], 441, shape[2], shape[3]), device=dev, generator=g).half()
for _ in range(n):
    if "fwd" in what:
        out = ops.correlation_fp16(fa, fb)
    if "bwd" in what:
        g0, g1 = ops.correlation_fp16_grad(top, fa, fb)
torch.cuda.synchronize()
