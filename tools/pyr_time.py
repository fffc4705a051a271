"""HIP-event times of ops.downsample_pyramid against the chain of ops.downsample calls it replaces (B = 32, 384 x 1280), and the
KITTI-config pyramid step eager / graph (best of 5).  usage: python tools/pyr_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from unflow_amd import ops
dev = torch.device("cuda:0")
wl = bench.Workload(32, dev, seed=1234, global_batch=32, features=False)
mask = torch.ones((32, 384, 1280, 1), device=dev)


def chain():
    out = []
    for t in (wl.im1, wl.im2, mask):
        x = ops.downsample(t, 4)
        for _ in range(4):
            x = ops.downsample(x, 2)
        out.append(x)
    return out


def fused():
    return ops.downsample_pyramid(wl.im1, wl.im2, 5, both_orders=True), ops.downsample_pyramid(mask, mask, 5)


nbytes = 2 * wl.im1.numel() * 4 * (1 + 1 / 16 * 4 / 3) + 2 * mask.numel() * 4
for name, f in (("chain of downsample calls (15 launches)", chain), ("downsample_pyramid (2 launches, both stacked orders)", fused),
                ("downsample_pyramid images only", lambda: ops.downsample_pyramid(wl.im1, wl.im2, 5, both_orders=True))):
    ms = bench.event_time(torch, f, n=20, warm=3)
    print("%-55s %.4f ms" % (name, ms))
print("images only: %.0f GB/s of reads" % (2 * wl.im1.numel() * 4 / bench.event_time(torch, lambda: ops.downsample_pyramid(wl.im1, wl.im2, 5, both_orders=True), n=20, warm=3) / 1e6))
best = None
for _ in range(5):
    r = bench.kitti_config_step(wl)
    if best is None or r["graph_ms"] < best["graph_ms"]:
        best = dict(r)
    best["eager_ms"] = min(best["eager_ms"], r["eager_ms"])
print("kitti_config_step: graph %.4f ms  eager %.3f ms  loss %.6f" % (best["graph_ms"], best["eager_ms"], best["loss"]))
