"""cProfile of the eager KITTI-config pyramid step (host side): where the Python time of its ~155 launches goes."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from unflow_amd import unsupervised
wl = bench.Workload(32, torch.device("cuda:0"), seed=1234, global_batch=32, features=False)
pf, pb = bench.kitti_pyramid_flows(torch, wl.im1.device, 32)


def step():
    loss, _ = unsupervised.pyramid_loss(wl.im1, wl.im2, wl.border, pf, pb, bench.KITTI)
    return (loss,) + torch.autograd.grad(loss, pf + pb)


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
