import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from unflow_amd import ops, _lib
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
B = 32
fa = torch.randn((B, 256, 48, 160), device=dev, generator=g)
fb = torch.randn((B, 256, 48, 160), device=dev, generator=g)
top = torch.randn((B, 441, 48, 160), device=dev, generator=g)
buf = torch.zeros(2 * 8 * 4096, dtype=torch.int64, device=dev)
L = _lib.load()
L.unflow_debug_set_buffer.argtypes = [ctypes.c_void_p]
for _ in range(2):
    ops.correlation_grad(top, fa, fb)
L.unflow_debug_set_buffer(ctypes.c_void_p(buf.data_ptr()))
ops.correlation_grad(top, fa, fb)
torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(2, 4096, 8)[:, :1280]
for k, name in enumerate(("wrt0", "wrt1")):
    x = a[k].astype(np.float64)
    print(name, "cycles per workgroup (mean): prologue %.0f  ksteps %.0f  barrier %.0f  tile-epilogues %.0f  total %.0f" % tuple(x[:, i].mean() for i in range(5)))
    span = (x[:, 6].max() - x[:, 5].min())
    print("   kernel span %.0f cycles; sum of per-WG totals / 256 CUs = %.0f; min/max WG total %.0f / %.0f" % (span, x[:, 4].sum() / 256, x[:, 4].min(), x[:, 4].max()))
