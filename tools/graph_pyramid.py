"""Feasibility: capture the 5-level pyramid loss fwd+bwd in a HIP graph (torch.cuda.CUDAGraph)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unflow_amd import unsupervised, losses

dev = torch.device("cuda:0")
B, H, W = 32, 384, 1280
g = torch.Generator(device=dev); g.manual_seed(0)
im1 = torch.rand((B, H, W, 3), device=dev, generator=g)
im2 = torch.rand((B, H, W, 3), device=dev, generator=g)
border = losses.create_border_mask(im1, 0.1)
kitti = dict(pyramid_loss=True, border_mask=True, mask_occlusion="fb", ternary_weight=1.0,
             smooth_2nd_weight=3.0, fb_weight=0.2, occ_weight=12.4)
pf = [(torch.randn((B, H >> (2 + i), W >> (2 + i), 2), device=dev, generator=g) * 0.3).requires_grad_(True) for i in range(5)]
pb = [(-f.detach() + 0.02 * torch.randn(f.shape, device=dev, generator=g)).requires_grad_(True) for f in pf]

def step():
    loss, _ = unsupervised.pyramid_loss(im1, im2, border, pf, pb, kitti)
    grads = torch.autograd.grad(loss, pf + pb)
    return loss, grads

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print("eager  %.3f ms" % timeit(step))
_, ga = step()
_, gb = step()
print("eager vs eager max grad diff", max(float((a - b).abs().max()) for a, b in zip(ga, gb)), "max |grad|", max(float(a.abs().max()) for a in ga))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    loss_g, grads_g = step()
loss_e, grads_e = step()
graph.replay(); torch.cuda.synchronize()
print("loss eager %.6f graph %.6f" % (float(loss_e), float(loss_g)))
print("max grad diff", max(float((a - b).abs().max()) for a, b in zip(grads_e, grads_g)))
print("graph  %.3f ms" % timeit(graph.replay))
