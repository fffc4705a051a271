import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from unflow_amd import ops
from unflow_amd.image_warp import image_warp
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(7)
for B in (8, 32):
    im1, im2 = bench.smooth_images(torch, gen, B), bench.smooth_images(torch, gen, B)
    ffw = bench.smooth_flow(torch, gen, B); fbw = -ffw + 0.05 * bench.smooth_flow(torch, gen, B)
    fns = {"backward_warp x2": lambda: (ops.backward_warp(im2, ffw), ops.backward_warp(im1, fbw)),
           "image_warp x2": lambda: (image_warp(im2, ffw), image_warp(im1, fbw)),
           "backward_warp x1": lambda: ops.backward_warp(im2, ffw), "image_warp x1": lambda: image_warp(im2, ffw)}
    with torch.no_grad():
        for rep in range(2):
            for k in ("backward_warp x2", "image_warp x2", "image_warp x1", "backward_warp x1", "backward_warp x2"):
                ms = bench.event_time(torch, fns[k], n=10, warm=2)
                n = 2 if "x2" in k else 1
                print("B=%d %-18s %.4f ms  %.0f GB/s" % (B, k, ms, n * B * 384 * 1280 * 32 / ms / 1e6))
