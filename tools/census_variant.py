"""Times the fused census-of-a-warp loss (forward, and forward + backward) of ONE build
(UNFLOW_HIP_LIB) at B=32 384x1280 and checks it against the default build's results (bitwise
for the forward scalar, allclose for the scatter-added gradients).
usage: python tools/census_variant.py <name> [path.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1]
if len(sys.argv) > 2:
    os.environ["UNFLOW_HIP_LIB"] = os.path.abspath(sys.argv[2])
import torch
import bench
from unflow_amd import losses
dev = torch.device("cuda:0")
wl = bench.Workload(32, dev, seed=1234, global_batch=32, features=False)
torch.manual_seed(7)
mask = (torch.rand(32, 384, 1280, 1, device=dev) > 0.2).float()
flow = wl.flow_fw.detach().clone().requires_grad_(True)


def fwd():
    with torch.no_grad():
        return losses.ternary_loss_warped(wl.im1, wl.im2, flow, mask, max_distance=3)


def fwd_bwd():
    flow.grad = None
    losses.ternary_loss_warped(wl.im1, wl.im2, flow, mask, max_distance=3).backward()
    return flow.grad


res = {"loss": fwd().reshape(1), "dflow": fwd_bwd().clone()}
t_f = bench.event_time(torch, fwd, n=20, warm=3)
t_fb = bench.event_time(torch, fwd_bwd, n=20, warm=3)
print("%-10s fused census+warp: forward %.4f ms | forward+backward %.4f ms" % (name, t_f, t_fb))
if len(sys.argv) <= 2:
    torch.save({k: v.cpu() for k, v in res.items()}, "/tmp/census_default_out.pt")
else:
    ref = torch.load("/tmp/census_default_out.pt")
    print("   vs default build:", {k: ("bit-equal" if torch.equal(ref[k], res[k].cpu()) else "max diff %.3g" % float((ref[k] - res[k].cpu()).abs().max())) for k in res})
