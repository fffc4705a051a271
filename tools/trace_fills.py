"""Which host-side calls of the pyramid step launch fill / copy kernels: torch profiler with Python stacks over one
eager step of the KITTI-config pyramid loss (the workload of bench.py's kitti_config_step)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unflow_amd import unsupervised

wl = bench.Workload(32, torch.device("cuda:0"), 1237, 32, features=False)
dev = wl.im1.device
g = torch.Generator(device=dev); g.manual_seed(7)
H, W, B = wl.im1.shape[1], wl.im1.shape[2], wl.im1.shape[0]
pf = [(8.0 / 2 ** i * torch.randn((B, H // (4 << i), W // (4 << i), 2), device=dev, generator=g)).requires_grad_(True) for i in range(5)]
pb = [(-f.detach() + 0.02 * torch.randn(f.shape, device=dev, generator=g)).requires_grad_(True) for f in pf]


def step():
    loss, _ = unsupervised.pyramid_loss(wl.im1, wl.im2, wl.border, pf, pb, bench.KITTI)
    return (loss,) + torch.autograd.grad(loss, pf + pb)


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
want = ("aten::zeros", "aten::zero_", "aten::fill_", "aten::clone", "aten::copy_", "aten::ones", "aten::zeros_like", "aten::ones_like", "aten::add", "aten::add_", "aten::mul", "aten::neg", "aten::sum", "aten::_foreach_mul", "aten::contiguous")
cnt = collections.Counter()
for e in prof.events():
    if e.name in want:
        st = [s for s in (e.stack or []) if "unflow_amd" in s or "bench.py" in s or "autograd" in s]
        cnt[(e.name, tuple(e.input_shapes[0]) if e.input_shapes else (), st[0] if st else "(engine)")] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(v, k)
