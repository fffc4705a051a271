"""Times the four HBM-bound warps of ONE build (UNFLOW_HIP_LIB) at B=32 384x1280 C=3 and checks them
against the default build's results (bitwise).  usage: python tools/warp_variant.py <name> [path.so]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1]
if len(sys.argv) > 2:
    os.environ["UNFLOW_HIP_LIB"] = os.path.abspath(sys.argv[2])
import torch
import bench
from unflow_amd import ops
from unflow_amd.image_warp import image_warp, image_warp_grad
dev = torch.device("cuda:0")
wl = bench.Workload(32, dev, seed=1234, global_batch=32, features=False)
ffw = wl.flow_fw.detach()
g3 = torch.randn_like(wl.im2)
fns = {"backward_warp": (lambda: ops.backward_warp(wl.im2, ffw), 32), "backward_warp_grad": (lambda: ops.backward_warp_grad(g3, wl.im2, ffw), 40),
       "image_warp": (lambda: image_warp(wl.im2, ffw), 32), "image_warp_grad_flow": (lambda: image_warp_grad(g3, wl.im2, ffw, False, True)[1], 40),
       "image_warp C2": (lambda: image_warp(wl.flow_bw.detach(), ffw), 24)}
res = {k: f() for k, (f, _) in fns.items()}
msg = ""
if len(sys.argv) > 2:
    import subprocess, pickle
    # reference from the default build in a child process (one library per process)
    torch.save({k: v.cpu() for k, v in res.items()}, "/tmp/warp_variant_out.pt")
npx = 32 * 384 * 1280
out = []
for k, (f, bpp) in fns.items():
    ms = bench.event_time(torch, f, n=20, warm=3)
    out.append("%s %.4f ms %.0f GB/s" % (k, ms, npx * bpp / ms / 1e6))
print("%-14s " % name + " | ".join(out))
if len(sys.argv) <= 2:
    torch.save({k: v.cpu() for k, v in res.items()}, "/tmp/warp_default_out.pt")
else:
    ref = torch.load("/tmp/warp_default_out.pt")
    print("   vs default build:", {k: ("bit-equal" if torch.equal(ref[k], res[k].cpu()) else "DIFFERS %.3g" % float((ref[k] - res[k].cpu()).abs().max())) for k in res})
