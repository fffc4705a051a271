"""graph / eager time of bench.py's kitti_config_step for ONE build (UNFLOW_HIP_LIB): usage: python tools/kitti_step_time.py name [path.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2:
    os.environ["UNFLOW_HIP_LIB"] = os.path.abspath(sys.argv[2]); os.environ["UNFLOW_ALLOW_DIAGNOSTIC_BUILD"] = "1"
import torch, bench
wl = bench.Workload(32, torch.device("cuda:0"), seed=1234, global_batch=32, features=False)
best = None
for _ in range(3):
    r = bench.kitti_config_step(wl)
    best = r if best is None or r["graph_ms"] < best["graph_ms"] else best
print("%-10s graph %.4f ms  eager %.3f ms  loss %.6f" % (sys.argv[1], best["graph_ms"], best["eager_ms"], best["loss"]))
