"""Times the f16 matrix-core correlation against the float32 kernels and the fp16-storage (exact) form at
BASELINE config 5's feature shape [16,256,96,160] (default) or cfg3's [32,256,48,160] (--cfg3)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unflow_amd import ops

shape = (32, 256, 48, 160) if "--cfg3" in sys.argv else (16, 256, 96, 160)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
f0 = torch.nn.functional.leaky_relu(torch.randn(shape, device=dev, generator=g), 0.1)
f1 = torch.nn.functional.leaky_relu(torch.randn(shape, device=dev, generator=g), 0.1)
top = torch.randn((shape[0], 441, shape[2], shape[3]), device=dev, generator=g)
h0, h1, th = f0.half(), f1.half(), top.half()


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


rows = [("f32 fwd", lambda: ops.correlation(f0, f1)), ("f32 bwd", lambda: ops.correlation_grad(top, f0, f1)),
        ("f16 storage (exact) fwd", lambda: ops.correlation_fp16_exact(h0, h1)),
        ("f16 storage (exact) bwd", lambda: ops.correlation_fp16_grad(th, h0, h1, exact=True)),
        ("f16 matrix cores fwd", lambda: ops.correlation_fp16(h0, h1)),
        ("f16 matrix cores bwd", lambda: ops.correlation_fp16_grad(th, h0, h1))]
if "--only16" in sys.argv:      # A/B runs of variant builds (UNFLOW_HIP_LIB=unflow_amd/variants/<name>.so)
    rows = rows[4:]
    print("build:", os.environ.get("UNFLOW_HIP_LIB", "default"))
with torch.no_grad():
    for name, fn in rows:
        print("%-28s %8.3f ms" % (name, timeit(fn)), flush=True)
    ref = ops.correlation(h0.float(), h1.float())
    got = ops.correlation_fp16(h0, h1).float()
    print("fwd max abs err vs f32 kernel on widened inputs: %.3g (max |ref| %.3g)" % (float((got - ref).abs().max()), float(ref.abs().max())))
    r0, r1 = ops.correlation_grad(th.float(), h0.float(), h1.float())
    g0, g1 = ops.correlation_fp16_grad(th, h0, h1)
    print("bwd max abs err: g0 %.3g g1 %.3g (max |ref| %.3g)" % (float((g0.float() - r0).abs().max()), float((g1.float() - r1).abs().max()), float(r0.abs().max())))
