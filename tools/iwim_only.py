"""Profiling target: the image gradient of image_warp (C = 3 and C = 2, bench flows, B = 32 at 384 x 1280) and
ops.downsample_pyramid, a few launches each.  usage: python tools/iwim_only.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unflow_amd import ops
from unflow_amd.image_warp import image_warp_grad
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
wl = bench.Workload(32, dev, seed=1234, global_batch=32, features=False)
ffw, fbw = wl.flow_fw.detach(), wl.flow_bw.detach()
torch.manual_seed(7)
g3, g2 = torch.randn_like(wl.im2), torch.randn_like(fbw)
for _ in range(n):
    image_warp_grad(g3, wl.im2, ffw, True, False)
    image_warp_grad(g2, fbw, ffw, True, False)
    ops.downsample_pyramid(wl.im1, wl.im2, 5, both_orders=True)
torch.cuda.synchronize()
