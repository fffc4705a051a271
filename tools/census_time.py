"""HIP-event times of the step's census kernels (fused-warp forward / backward) for one build of the library.
usage: python tools/census_time.py <label> [variant.so]"""
import os, sys
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 2:
    os.environ["UNFLOW_HIP_LIB"] = os.path.abspath(sys.argv[2]); os.environ["UNFLOW_ALLOW_DIAGNOSTIC_BUILD"] = "1"
import torch, bench
wl = bench.Workload(32, torch.device("cuda:0"), seed=1234, global_batch=32, features=False)
copy = bench.stream_copy_gbs(torch, torch.device("cuda:0"))
best = None
for _ in range(3):
    r = bench.loss_kernels_roofline(wl, copy)
    ms = r["census_fwd_fused_warp"]["ms"], r["census_bwd_fused_warp"]["ms"], r["forward_warp"]["ms"]
    best = ms if best is None else tuple(min(a, b) for a, b in zip(best, ms))
print("%-10s census fwd %.4f ms  bwd %.4f ms  forward_warp %.4f ms" % (sys.argv[1], *best))
