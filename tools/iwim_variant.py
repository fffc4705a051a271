"""Times the image gradient of image_warp (the LDS-privatised scatter, csrc/warp.hip image_warp_bwd_im_tiled_kernel) of ONE build
(UNFLOW_HIP_LIB) at B=32 384x1280 for C = 3 and C = 2 and at the pyramid's H/4 level, and compares the results with the default
build's.  usage: python tools/iwim_variant.py <name> [path.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1]
if len(sys.argv) > 2:
    os.environ["UNFLOW_HIP_LIB"] = os.path.abspath(sys.argv[2])
import torch
import bench
from unflow_amd.image_warp import image_warp_grad
dev = torch.device("cuda:0")
wl = bench.Workload(32, dev, seed=1234, global_batch=32, features=False)
ffw = wl.flow_fw.detach()
torch.manual_seed(7)
g3 = torch.randn_like(wl.im2)
g2 = torch.randn_like(wl.flow_bw.detach())
fbw = wl.flow_bw.detach()
# the pyramid's finest default level: 96 x 320
g3s, ims, fls = g3[:, ::4, ::4].contiguous(), wl.im2[:, ::4, ::4].contiguous(), (ffw[:, ::4, ::4] / 4).contiguous()
fns = {"im C3": (lambda: image_warp_grad(g3, wl.im2, ffw, True, False)[0], 32 * 384 * 1280 * 32),
       "im C2": (lambda: image_warp_grad(g2, fbw, ffw, True, False)[0], 32 * 384 * 1280 * 24),
       "im C3 96x320": (lambda: image_warp_grad(g3s, ims, fls, True, False)[0], 32 * 96 * 320 * 32)}
res = {k: f() for k, (f, _) in fns.items()}
out = []
for k, (f, nbytes) in fns.items():
    ms = bench.event_time(torch, f, n=20, warm=3)
    out.append("%s %.4f ms %.0f GB/s" % (k, ms, nbytes / ms / 1e6))
print("%-14s " % name + " | ".join(out))
if len(sys.argv) <= 2:
    torch.save({k: v.cpu() for k, v in res.items()}, "/tmp/iwim_default_out.pt")
else:
    ref = torch.load("/tmp/iwim_default_out.pt")
    print("   vs default build:", {k: ("bit-equal" if torch.equal(ref[k], res[k].cpu()) else "differs by %.3g of %.3g" % (float((ref[k] - res[k].cpu()).abs().max()), float(ref[k].abs().max()))) for k in res})
