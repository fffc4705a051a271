import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from unflow_amd import ops, losses, util, flownet, image_warp
wl = bench.Workload(32, torch.device("cuda:0"), seed=1234, global_batch=32, features=False)
def run(tag):
    best = None
    for _ in range(3):
        r = bench.kitti_config_step(wl)
        best = r if best is None or r["eager_ms"] < best["eager_ms"] else best
    print("%-28s eager %.3f ms  graph %.4f ms" % (tag, best["eager_ms"], best["graph_ms"]))
run("fast host path")
fast_on, fast_raw = ops._on, ops._raw_stream
slow_on = lambda d: torch.cuda.device(d)
for m in (ops, losses, util, flownet, image_warp):
    m._on = slow_on
ops._raw_stream = None
run("round-3 host path")
for m in (ops, losses, util, flownet, image_warp):
    m._on = fast_on
ops._raw_stream = fast_raw
run("fast host path (again)")
