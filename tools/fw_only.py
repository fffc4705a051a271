import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unflow_amd import ops
import bench
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
B = 32
flow = bench.smooth_flow(g, B)
rnd = (torch.rand((B, 384, 1280, 2), device=dev, generator=g) - 0.5) * 60
for name, f in (("smooth", flow), ("random U(-30,30)", rnd)):
    ops.forward_warp(f)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        ops.forward_warp(f)
    e.record(); torch.cuda.synchronize()
    print("dbg=%s %-18s forward_warp %.3f ms" % (os.environ.get("UNFLOW_FW_DBG", "0"), name, s.elapsed_time(e) / 5))
