"""The eager KITTI-config pyramid step timed in successive blocks of ten steps, from the first step of the process: shows the one
slow block early in a process (bench.py times three blocks and reports their minimum).  usage: python tools/eager_blocks.py [nogc]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "nogc":
    import gc
    gc.disable()
import torch, bench, time
wl = bench.Workload(32, torch.device("cuda:0"), seed=1234, global_batch=32, features=False)
from unflow_amd import unsupervised
pf, pb = bench.kitti_pyramid_flows(torch, wl.im1.device, 32)
def step():
    loss, _ = unsupervised.pyramid_loss(wl.im1, wl.im2, wl.border, pf, pb, bench.KITTI)
    return (loss,) + torch.autograd.grad(loss, pf + pb)
for blk in range(6):
    t0 = time.perf_counter()
    ms = bench.event_time(torch, step, n=10, warm=0)
    print("block %d: %.3f ms per step by events, %.3f by the host clock" % (blk, ms, (time.perf_counter() - t0) * 100))
