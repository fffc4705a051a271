"""Profiling target: the loss leg of the bench step (fused census + warp kernels, fb-occlusion tile kernel,
forward_warp + sym) and the image gradient of image_warp (C = 3 and C = 2: the fb term's scatter), at the bench's
batch and on the bench's synthetic fields.  usage: loss_step_only.py [B] [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unflow_amd.image_warp import image_warp_grad
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
wl = bench.Workload(B, dev, seed=1234, global_batch=B, features=False)
g3 = torch.randn_like(wl.im2)
g2 = torch.randn_like(wl.flow_fw.detach())
ffw, fbw = wl.flow_fw.detach(), wl.flow_bw.detach()
for _ in range(n):
    wl.loss_leg()
    image_warp_grad(g3, wl.im2, ffw, True, False)
    image_warp_grad(g2, fbw, ffw, True, False)
torch.cuda.synchronize()
