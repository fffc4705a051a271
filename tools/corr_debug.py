"""Where does the correlation backward of the default build differ from another build's?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from unflow_amd import ops
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "2"))
g = torch.Generator(device=dev); g.manual_seed(0)
fa = torch.randn((B, 256, 48, 160), device=dev, generator=g)
fb = torch.randn((B, 256, 48, 160), device=dev, generator=g)
top = torch.randn((B, 441, 48, 160), device=dev, generator=g)
g0, g1 = ops.correlation_grad(top, fa, fb)
ref = ctypes.CDLL(os.path.abspath(sys.argv[1]))
P = ctypes.c_void_p
h0, h1 = torch.empty_like(g0), torch.empty_like(g1)
rc = ref.unflow_correlation_bwd(P(top.data_ptr()), P(fa.data_ptr()), P(fb.data_ptr()), None, None, P(h0.data_ptr()), P(h1.data_ptr()),
                                B, 256, 48, 160, 1, 20, 20, 1, 2, P(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
assert rc == 0
for name, a, b in (("g0", g0, h0), ("g1", g1, h1)):
    d = (a - b).abs()
    bad = d > 1e-5 * b.abs().max()
    print(name, "max err", float(d.max()), "ref max", float(b.abs().max()), "bad fraction", float(bad.float().mean()), "nan", int(torch.isnan(a).sum()))
    if bad.any():
        idx = bad.nonzero()
        for dim, nm in enumerate(("b", "c", "y", "x")):
            vals = idx[:, dim].unique()
            print("   bad", nm, "count", len(vals), "values", vals[:40].tolist())
        print("   sample", [(a[tuple(i)].item(), b[tuple(i)].item()) for i in idx[:6]])
        byc = bad.float().mean(dim=(0, 2, 3)); print("   bad frac by channel[0:16]", [round(float(v), 2) for v in byc[:16]], "...[128:136]", [round(float(v), 2) for v in byc[128:136]])
        byy = bad.float().mean(dim=(0, 1, 3)); print("   bad frac by y", [round(float(v), 2) for v in byy])
        byx = bad.float().mean(dim=(0, 1, 2)); print("   bad frac by x[0:40]", [round(float(v), 2) for v in byx[:40]])
