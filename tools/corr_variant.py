"""Times the correlation forward / backward of ONE build of the library (UNFLOW_HIP_LIB) with HIP
events at B=32 [32,256,48,160] and checks the results against the default build's, loaded beside it
through its own ctypes handle.  usage: python tools/corr_variant.py <name> [path.so]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1]
if len(sys.argv) > 2:
    os.environ["UNFLOW_HIP_LIB"] = os.path.abspath(sys.argv[2])
import torch  # noqa: E402
from unflow_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "32"))
g = torch.Generator(device=dev)
g.manual_seed(0)
lrelu = torch.nn.functional.leaky_relu
fa = lrelu(torch.randn((B, 256, 48, 160), device=dev, generator=g), 0.1)
fb = lrelu(torch.randn((B, 256, 48, 160), device=dev, generator=g), 0.1)
top = torch.randn((B, 441, 48, 160), device=dev, generator=g)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


out = ops.correlation(fa, fb)
g0, g1 = ops.correlation_grad(top, fa, fb)
msg = ""
if len(sys.argv) > 2:       # reference results from the default build
    ref = ctypes.CDLL(os.path.join(ROOT, "unflow_amd", "libunflow_hip.so"))
    P = ctypes.c_void_p
    o2, h0, h1 = torch.empty_like(out), torch.empty_like(g0), torch.empty_like(g1)
    st = P(torch.cuda.current_stream().cuda_stream)
    rc1 = ref.unflow_correlation_fwd(P(fa.data_ptr()), P(fb.data_ptr()), P(o2.data_ptr()), None, None, B, 256, 48, 160, 1, 20, 20, 1, 2, st)
    rc2 = ref.unflow_correlation_bwd(P(top.data_ptr()), P(fa.data_ptr()), P(fb.data_ptr()), None, None, P(h0.data_ptr()),
                                     P(h1.data_ptr()), B, 256, 48, 160, 1, 20, 20, 1, 2, st)
    torch.cuda.synchronize()
    assert rc1 == 0 and rc2 == 0

    def err(a, b):
        return float((a - b).abs().max() / b.abs().max())
    msg = " | vs default build: fwd %s (%.2g), g0 %s (%.2g), g1 %s (%.2g)" % (
        "bit-equal" if torch.equal(out, o2) else "DIFFERS", err(out, o2), "bit-equal" if torch.equal(g0, h0) else "DIFFERS",
        err(g0, h0), "bit-equal" if torch.equal(g1, h1) else "DIFFERS", err(g1, h1))
fwd = timeit(lambda: ops.correlation(fa, fb))
bwd = timeit(lambda: ops.correlation_grad(top, fa, fb))
print("%-28s fwd %.4f ms (%.1f TF)  bwd %.4f ms (%.1f TF, %.3f of peak)%s" % (
    name, fwd, 2 * 441 * 48 * 160 * 256 * B / fwd / 1e9, bwd, 4 * 441 * 48 * 160 * 256 * B / bwd / 1e9,
    4 * 441 * 48 * 160 * 256 * B / bwd / 1e9 / 157.3, msg))
