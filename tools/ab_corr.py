"""A/B timing of the correlation forward / backward of several builds of the library IN ONE PROCESS, rounds interleaved
(boxes differ by a few percent and so do separate processes; deltas below that need this).  Every build is loaded
through its own ctypes handle; results are compared with the first build's.
usage: python tools/ab_corr.py name=path.so [name=path.so ...]     (B, ROUNDS, N in the environment)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "32"))
ROUNDS, N = int(os.environ.get("ROUNDS", "9")), int(os.environ.get("N", "10"))
g = torch.Generator(device=dev)
g.manual_seed(0)
lrelu = torch.nn.functional.leaky_relu
fa = lrelu(torch.randn((B, 256, 48, 160), device=dev, generator=g), 0.1)
fb = lrelu(torch.randn((B, 256, 48, 160), device=dev, generator=g), 0.1)
top = torch.randn((B, 441, 48, 160), device=dev, generator=g)
P = ctypes.c_void_p
st = P(torch.cuda.current_stream().cuda_stream)
builds = []
for a in sys.argv[1:]:
    name, path = a.split("=", 1)
    lib = ctypes.CDLL(os.path.abspath(path))
    out, g0, g1 = torch.empty((B, 441, 48, 160), device=dev), torch.empty_like(fa), torch.empty_like(fb)

    def fwd(lib=lib, out=out):
        assert lib.unflow_correlation_fwd(P(fa.data_ptr()), P(fb.data_ptr()), P(out.data_ptr()), None, None, B, 256, 48, 160, 1, 20, 20, 1, 2, st) == 0

    def bwd(lib=lib, g0=g0, g1=g1):
        assert lib.unflow_correlation_bwd(P(top.data_ptr()), P(fa.data_ptr()), P(fb.data_ptr()), None, None, P(g0.data_ptr()), P(g1.data_ptr()),
                                          B, 256, 48, 160, 1, 20, 20, 1, 2, st) == 0
    builds.append(dict(name=name, fwd=fwd, bwd=bwd, out=out, g0=g0, g1=g1, t={"fwd": [], "bwd": []}))


def timeit(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(N):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / N


for b in builds:
    b["fwd"](); b["bwd"]()
torch.cuda.synchronize()
for r in range(ROUNDS):
    for kind in ("fwd", "bwd"):
        for b in (builds if r % 2 == 0 else builds[::-1]):
            b["t"][kind].append(timeit(b[kind]))
ref = builds[0]
for b in builds:
    same = "" if b is ref else "  vs %s: fwd %s g0 %s g1 %s" % (ref["name"], *("bit-equal" if torch.equal(b[k], ref[k]) else "DIFFERS" for k in ("out", "g0", "g1")))
    f, w = np.array(b["t"]["fwd"][1:]), np.array(b["t"]["bwd"][1:])
    print("%-16s fwd median %.4f min %.4f ms (%.3f of peak) | bwd median %.4f min %.4f ms (%.3f of peak)%s" % (
        b["name"], np.median(f), f.min(), 2 * 441 * 48 * 160 * 256 * B / np.median(f) / 1e9 / 157.3, np.median(w), w.min(),
        4 * 441 * 48 * 160 * 256 * B / np.median(w) / 1e9 / 157.3, same))
