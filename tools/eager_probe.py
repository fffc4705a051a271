"""Host timeline of the eager KITTI-config pyramid step: forward (Python), backward (autograd engine), wait for the
GPU -- and the same with ops._dev replaced by the identity (what its checks cost in the step)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
from unflow_amd import ops, losses, util, flownet, image_warp, unsupervised
wl = bench.Workload(32, torch.device("cuda:0"), seed=1234, global_batch=32, features=False)
pf, pb = bench.kitti_pyramid_flows(torch, wl.im1.device, 32)


def step():
    t0 = time.perf_counter()
    loss, _ = unsupervised.pyramid_loss(wl.im1, wl.im2, wl.border, pf, pb, bench.KITTI)
    t1 = time.perf_counter()
    g = torch.autograd.grad(loss, pf + pb)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2


def run(tag):
    for _ in range(5):
        step()
    rows = [step() for _ in range(30)]
    med = [sorted(r[i] for r in rows)[len(rows) // 2] * 1e3 for i in range(3)]
    print("%-22s forward %.3f ms  backward %.3f ms  wait %.3f ms  total %.3f ms" % (tag, med[0], med[1], med[2], sum(med)))


run("as shipped")
keep = ops._dev
for m in (ops, losses, util, flownet, image_warp):
    m._dev = lambda t, name: t
run("_dev = identity")
for m in (ops, losses, util, flownet, image_warp):
    m._dev = keep
run("as shipped (again)")
