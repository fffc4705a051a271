"""Times forward_warp (the Gaussian splat of the occlusion masks) of ONE build (UNFLOW_HIP_LIB) at
B=32 384x1280 on the bench's coherent flow and on i.i.d. +-30 px flows, and compares the result
with the default build's (saved by the run without a path).
usage: python tools/fwarp_variant.py <name> [path.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1]
if len(sys.argv) > 2:
    os.environ["UNFLOW_HIP_LIB"] = os.path.abspath(sys.argv[2])
import torch
import bench
from unflow_amd import ops
dev = torch.device("cuda:0")
wl = bench.Workload(32, dev, seed=1234, global_batch=32, features=False)
g = torch.Generator(device="cpu").manual_seed(5)
flows = {"coherent": wl.flow_fw.detach(), "iid30": ((torch.rand(32, 384, 1280, 2, generator=g) - 0.5) * 60).to(dev)}
out, res = [], {}
for k, f in flows.items():
    res[k] = ops.forward_warp(f)
    out.append("%s %.4f ms" % (k, bench.event_time(torch, lambda: ops.forward_warp(f), n=20, warm=3)))
print("%-18s " % name + " | ".join(out))
if len(sys.argv) <= 2:
    torch.save({k: v.cpu() for k, v in res.items()}, "/tmp/fwarp_default_out.pt")
else:
    ref = torch.load("/tmp/fwarp_default_out.pt")
    print("   vs default build:", {k: "max |diff| %.3g (max value %.3g)" % (float((ref[k] - res[k].cpu()).abs().max()), float(ref[k].max())) for k in res})
