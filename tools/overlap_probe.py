"""Do forward_warp (LDS-atomic bound) and the census forward (VALU bound) overlap when they run on two HIP streams?
cfg3 shapes; HIP-event time of {2 forward_warp + 2 ternary_loss_warped forward} in one stream and forked."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from unflow_amd import ops, losses
wl = bench.Workload(32, torch.device("cuda:0"), seed=1234, global_batch=32, features=False)
dev = wl.im1.device
ffw, fbw = wl.flow_fw.detach(), wl.flow_bw.detach()
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream()


def tern():
    with torch.no_grad():
        a = losses.ternary_loss_warped(wl.im1, wl.im2, ffw, wl.border, max_distance=3)
        b = losses.ternary_loss_warped(wl.im2, wl.im1, fbw, wl.border, max_distance=3)
    return a, b


def fwarp():
    return ops.forward_warp(ffw), ops.forward_warp(fbw)


def serial():
    fwarp(); tern()


def forked():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        fwarp()
    tern()
    main.wait_stream(side)


def forked_rev():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        tern()
    fwarp()
    main.wait_stream(side)


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for name, f in (("forward_warp x2 alone", fwarp), ("census forward x2 alone", tern), ("one stream", serial), ("two streams (splat on the side)", forked),
                ("two streams (census on the side)", forked_rev), ("one stream (again)", serial)):
    print("%-36s %.4f ms" % (name, timed(f)))
