"""Runs the level-0 loss kernels (image_warp, forward_warp, census fwd+bwd) a few times: profiling target."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unflow_amd import losses, ops
from unflow_amd.image_warp import image_warp
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
H, W = 384, 1280
x = torch.rand((B, 3, H // 8, W // 8), device=dev, generator=g)
up = lambda t: torch.nn.functional.interpolate(t, size=(H, W), mode="bicubic", align_corners=False).clamp(0, 1).permute(0, 2, 3, 1).contiguous()
im1, im2 = up(x), up(torch.rand((B, 3, H // 8, W // 8), device=dev, generator=g))
f = torch.randn((B, 2, H // 16, W // 16), device=dev, generator=g) * 8
flow = torch.nn.functional.interpolate(f, size=(H, W), mode="bicubic", align_corners=False).permute(0, 2, 3, 1).contiguous().requires_grad_(True)
mask = losses.create_border_mask(im1, 0.1)
for _ in range(n):
    flow.grad = None
    w = image_warp(im2, flow)
    loss = losses.ternary_loss(im1, w, mask, max_distance=3)
    loss.backward()
    ops.forward_warp(flow.detach())
torch.cuda.synchronize()
