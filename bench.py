#!/usr/bin/env python
"""bench.py -- the hot path on synthetic KITTI-shaped pairs (BASELINE.json metric).

One "step" = one pass of the whole path over a batch of image pairs
(BASELINE cfg3 shapes):

  * FlowNetC correlation, both directions (core/flownet.py:35-43 runs it fw and
    bw): 2 forward calls on [B,256,H/8,W/8] and their 2 backward calls;
  * one level-0 ``compute_losses`` at full resolution, max_distance=3
    (core/unsupervised.py:87-95): the 4 image_warp (the 2 of the images are
    evaluated inside the census kernels, the 2 of the flows inside the
    fb-occlusion mask kernel: nothing else reads them), fb-occlusion masks, 2
    forward_warp disocclusion densities + the 'sym' term, 2 soft-census ternary
    losses 7x7 with the charbonnier reductions -- forward, then backward to
    dL/dflow_fw, dL/dflow_bw;
  * with N>1 ranks: RCCL all-reduce of the loss scalar and of a 156.7 MB
    gradient buffer standing in for the FlowNetC weights.

Inputs are resident in HBM before the timed region.  Prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from unflow_amd import losses, ops  # noqa: E402

H, W, FEAT_C = 384, 1280, 256
GRAD_BUFFER_FLOATS = 39_170_000  # FlowNetC weights, core/flownet.py:179-233: 156.7 MB

# algorithmic bytes / flops per image pair, fp32 (DESIGN.md "Kernels")
H8, W8, D = H // 8, W // 8, 441
CORR_FWD_BYTES = 4 * H8 * W8 * (2 * FEAT_C + D)          # 29.28 MB
CORR_FWD_FLOP = 2 * D * H8 * W8 * FEAT_C                 # 1.734 GFLOP
CORR_BWD_BYTES = 4 * H8 * W8 * (D + 4 * FEAT_C)          # 45.0 MB
CORR_BWD_FLOP = 2 * CORR_FWD_FLOP                        # 3.47 GFLOP
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_MATRIX_PEAK_TF = 157.3  # MI355X_MICROARCH.md: dense f32 MFMA peak


def smooth_images(gen, B):
    """Low-pass filtered uniform noise in [0,1] built on the device."""
    x = torch.rand((B, 3, H // 8, W // 8), device=gen.device, generator=gen)
    x = torch.nn.functional.interpolate(x, size=(H, W), mode="bicubic", align_corners=False)
    x = x + 0.05 * torch.rand((B, 3, H, W), device=gen.device, generator=gen)
    x = (x - x.amin()) / (x.amax() - x.amin())
    return x.permute(0, 2, 3, 1).contiguous()


def smooth_flow(gen, B):
    """Smooth field ~N(0, 8 px) plus a per-sample translation U(-20,20)."""
    x = torch.randn((B, 2, H // 32, W // 32), device=gen.device, generator=gen) * 8.0
    x = torch.nn.functional.interpolate(x, size=(H, W), mode="bicubic", align_corners=False)
    t = (torch.rand((B, 2, 1, 1), device=gen.device, generator=gen) - 0.5) * 40.0
    return (x + t).permute(0, 2, 3, 1).contiguous()


class Workload:
    def __init__(self, B, dev, seed, global_batch):
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        self.B, self.global_batch = B, global_batch
        lrelu = torch.nn.functional.leaky_relu
        self.fa = lrelu(torch.randn((B, FEAT_C, H8, W8), device=dev, generator=gen), 0.1)
        self.fb = lrelu(torch.randn((B, FEAT_C, H8, W8), device=dev, generator=gen), 0.1)
        self.top_fw = torch.randn((B, D, H8, W8), device=dev, generator=gen)
        self.top_bw = torch.randn((B, D, H8, W8), device=dev, generator=gen)
        self.im1 = smooth_images(gen, B)
        self.im2 = smooth_images(gen, B)
        self.flow_fw = smooth_flow(gen, B).requires_grad_(True)
        self.flow_bw = (-self.flow_fw.detach() + 0.5 * smooth_flow(gen, B) * 0.1).requires_grad_(True)
        self.border = losses.create_border_mask(self.im1, 0.1)
        self.grad_buffer = torch.zeros(GRAD_BUFFER_FLOATS, device=dev)
        self.corr_events = []   # (start, end, kind) HIP events on the launch stream

    def _timed(self, kind, fn, *a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn(*a, **k)
        e.record()
        self.corr_events.append((s, e, kind))
        return r

    def step(self, world, bidirectional=False):
        with torch.no_grad():
            if bidirectional:
                # extension: corr(b, a) is a bit-exact re-indexing of corr(a, b), and one
                # backward call on top_fw + flip(top_bw) returns the summed gradients
                c_fw = ops.correlation(self.fa, self.fb)
                c_bw = ops.correlation_flip(c_fw)
                g_fw = ops.correlation_grad(ops.correlation_flip(self.top_bw, self.top_fw), self.fa, self.fb)
                g_bw = None
            else:
                c_fw = self._timed("corr_fwd", ops.correlation, self.fa, self.fb)
                c_bw = self._timed("corr_fwd", ops.correlation, self.fb, self.fa)
                g_fw = self._timed("corr_bwd", ops.correlation_grad, self.top_fw, self.fa, self.fb)
                g_bw = self._timed("corr_bwd", ops.correlation_grad, self.top_bw, self.fb, self.fa)
        self.flow_fw.grad = None
        self.flow_bw.grad = None
        ls = losses.compute_losses(self.im1, self.im2, self.flow_fw, self.flow_bw,
                                   border_mask=self.border, mask_occlusion="fb",
                                   data_max_distance=3, norm_batch=self.global_batch,
                                   terms=("ternary", "sym"))
        loss = ls["ternary"]
        loss.backward()
        if world > 1:
            dist.all_reduce(loss.detach(), op=dist.ReduceOp.SUM)
            dist.all_reduce(self.grad_buffer, op=dist.ReduceOp.SUM)
        return loss.detach(), (c_fw, c_bw, g_fw, g_bw)


def cpu_baseline(n_pairs=2, threads=0):
    """The CPU oracle (a restatement: kind "port") timed on this host over a bounded
    sample of the same step: n_pairs 384x1280 pairs, all host threads (threads=0) or
    the given number."""
    import oracle
    oracle.build()
    all_threads = oracle.num_threads()
    if threads:
        oracle.set_num_threads(threads)
    rng = np.random.default_rng(1234)
    f0 = rng.standard_normal((n_pairs, FEAT_C, H8, W8)).astype(np.float32)
    f1 = rng.standard_normal((n_pairs, FEAT_C, H8, W8)).astype(np.float32)
    top = rng.standard_normal((n_pairs, D, H8, W8)).astype(np.float32)
    im1 = rng.uniform(size=(n_pairs, H, W, 3)).astype(np.float32)
    im2 = rng.uniform(size=(n_pairs, H, W, 3)).astype(np.float32)
    ffw = (rng.standard_normal((n_pairs, H, W, 2)) * 4).astype(np.float32)
    fbw = -ffw
    border = oracle.create_border_mask((n_pairs, H, W))
    t0 = time.perf_counter()
    for a, b in ((f0, f1), (f1, f0)):
        c, p0, p1 = oracle.correlation(a, b)
        oracle.correlation_grad(top, a, b, p0, p1)
    _, aux = oracle.compute_losses(im1, im2, ffw, fbw, border, "fb", 3, terms=("ternary", "sym"))
    g2 = oracle.ternary_loss_grad(im1, aux["im2_warped"], aux["mask_fw"], 3)
    oracle.image_warp_grad(g2, im2, ffw, want_im=False)
    g1 = oracle.ternary_loss_grad(im2, aux["im1_warped"], aux["mask_bw"], 3)
    oracle.image_warp_grad(g1, im1, fbw, want_im=False)
    dt = time.perf_counter() - t0
    cores = oracle.num_threads()
    oracle.set_num_threads(all_threads)
    return {"value": n_pairs / dt, "unit": "image-pairs/s", "cores": cores, "kind": "port",
            "sample": "%d pairs of 384x1280 (same step: 2 corr fwd+bwd, level-0 loss fwd+bwd), %.1f s, "
                      "OpenMP C restatement oracle/unflow_oracle.c" % (n_pairs, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="image pairs per GPU (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-single-thread", action="store_true",
                    help="also time the CPU oracle on ONE thread (one pair, about a minute; SURVEY 8d)")
    ap.add_argument("--breakdown", action="store_true", help="also print per-op timings to stderr")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    wl = Workload(args.batch, dev, seed=1234 + rank, global_batch=args.batch * world)
    for _ in range(args.warmup):
        wl.step(world)
    wl.corr_events.clear()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = wl.step(world)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # for information only (never `value`): the same step with the bidirectional-correlation extension
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(world, bidirectional=True)
    fence()
    dt_bidir = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([dt_bidir], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_bidir = float(t.item())

    # dominant kernel: correlation backward (3.47 GFLOP/pair, f32 MFMA); averaged over the timed region
    per_kind = {}
    for s, e, kind in wl.corr_events:
        per_kind.setdefault(kind, []).append(s.elapsed_time(e))
    bwd_ms = float(np.mean(per_kind["corr_bwd"]))
    fwd_ms = float(np.mean(per_kind["corr_fwd"]))

    if args.breakdown and rank == 0:
        breakdown(wl, fwd_ms, bwd_ms)

    if rank == 0:
        pairs = args.batch * world * args.steps
        achieved_tf = CORR_BWD_FLOP * args.batch / (bwd_ms * 1e-3) / 1e12
        # HBM bytes of that launch from PMC counters are collected offline (bench.py cannot
        # run under rocprofv3 --pmc and time itself): tools/traffic_pmc.sh -> profiles/r01_traffic.json
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath) and args.batch == 32:
            with open(tpath) as fh:
                traffic = json.load(fh)["corr_bwd_both_launches"]["hbm_bytes"]
        # measured copy roofline of this GPU (1 GiB device-to-device)
        a = torch.empty(256 * 1024 * 1024, device=dev)
        b = torch.empty_like(a)
        b.copy_(a)
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(3):
            b.copy_(a)
        e0.record()
        torch.cuda.synchronize()
        copy_gbs = 3 * 2 * a.numel() * 4 / (s0.elapsed_time(e0) * 1e-3) / 1e9
        del a, b
        out = {
            "metric": "image-pairs/sec (corr+warp+census fwd+bwd) at 384x1280 b32",
            "value": pairs / dt, "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg3: 384x1280 pairs, batch %d per GPU: 2x FlowNetC correlation fwd+bwd on "
                                   "[B,256,48,160] + level-0 compute_losses (4 image_warp evaluated inside the census and fb-occlusion "
                                   "kernels, 2 forward_warp + sym term, 2 ternary 7x7) fwd+bwd" % args.batch,
                       "global_batch": args.batch * world, "height": H, "width": W,
                       "parallelism": "batch-sharded x%d, all-reduce of loss + 156.7 MB grad buffer" % world},
            "roofline": {"bound": "mfma",
                         "kernel": "correlation backward: corr_mfma_bwd_kernel, both gradient launches of one call",
                         "achieved": achieved_tf, "peak": FP32_MATRIX_PEAK_TF, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP32_MATRIX_PEAK_TF, "traffic": traffic,
                         "measured_copy_gbs": copy_gbs,
                         "avg_launch_ms": bwd_ms,
                         "hbm_gbs": CORR_BWD_BYTES * args.batch / (bwd_ms * 1e-3) / 1e9,
                         "hbm_frac_of_8TBs": CORR_BWD_BYTES * args.batch / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "corr_fwd": {"avg_launch_ms": fwd_ms,
                                      "tflops": CORR_FWD_FLOP * args.batch / (fwd_ms * 1e-3) / 1e12,
                                      "hbm_gbs": CORR_FWD_BYTES * args.batch / (fwd_ms * 1e-3) / 1e9}},
            "loss": float(loss),
        }
        out["extension_bidirectional_correlation"] = {
            "note": "NOT the headline: the same step when the caller obtains the (b,a) cost volume as a bit-exact "
                    "re-indexing of the (a,b) one and runs one backward call for both directions "
                    "(ops.correlation_bidirectional, DESIGN.md 4.1)",
            "ms_per_step": dt_bidir / args.steps * 1e3, "value": pairs / dt_bidir, "unit": "image-pairs/s"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            if args.cpu_single_thread:
                out["cpu_baseline"]["single_thread"] = cpu_baseline(n_pairs=1, threads=1)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def breakdown(wl, fwd_ms, bwd_ms):
    """Per-op device time (HIP events on the launch stream), to stderr."""
    from unflow_amd.image_warp import image_warp, image_warp_grad

    def timeit(fn, n=5):
        fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n

    B = wl.B
    npx = B * H * W
    ffw, fbw = wl.flow_fw.detach(), wl.flow_bw.detach()
    im2w = image_warp(wl.im2, ffw)
    g3 = torch.randn_like(im2w)
    rows = [("corr_fwd", fwd_ms, CORR_FWD_BYTES * B), ("corr_bwd", bwd_ms, CORR_BWD_BYTES * B)]
    # extension: both directions of a bidirectional FlowNetC from one correlation (bit-exact re-indexing)
    rows.append(("corr_fwd x2 via correlation_bidirectional (extension)",
                 timeit(lambda: ops.correlation_bidirectional(wl.fa, wl.fb)), 2 * CORR_FWD_BYTES * B))
    fa_g, fb_g = wl.fa.clone().requires_grad_(True), wl.fb.clone().requires_grad_(True)
    cfw, cbw = ops.correlation_bidirectional(fa_g, fb_g)
    rows.append(("corr_bwd x2 via correlation_bidirectional (extension)",
                 timeit(lambda: torch.autograd.grad([cfw, cbw], [fa_g, fb_g], [wl.top_fw, wl.top_bw], retain_graph=True)),
                 2 * CORR_BWD_BYTES * B))
    del cfw, cbw, fa_g, fb_g
    fah, fbh = wl.fa.half(), wl.fb.half()
    rows.append(("corr_fwd, fp16 storage / f32 accumulate (extension)", timeit(lambda: ops.correlation_fp16(fah, fbh)),
                 CORR_FWD_BYTES * B // 2))
    toph = wl.top_fw.half()
    rows.append(("corr_bwd, fp16 storage / f32 accumulate (extension)",
                 timeit(lambda: ops.correlation_fp16_grad(toph, fah, fbh)), CORR_BWD_BYTES * B // 2))
    del toph, fah, fbh
    rows.append(("image_warp C3", timeit(lambda: image_warp(wl.im2, ffw)), npx * 32))
    rows.append(("image_warp C2", timeit(lambda: image_warp(fbw, ffw)), npx * 24))
    rows.append(("image_warp_grad flow C3", timeit(lambda: image_warp_grad(g3, wl.im2, ffw, False, True)), npx * 40))
    rows.append(("image_warp_grad im C3", timeit(lambda: image_warp_grad(g3, wl.im2, ffw, True, False)), npx * 32))
    rows.append(("backward_warp C3", timeit(lambda: ops.backward_warp(wl.im2, ffw)), npx * 32))
    rows.append(("backward_warp_grad C3", timeit(lambda: ops.backward_warp_grad(g3, wl.im2, ffw)), npx * 40))
    rows.append(("forward_warp", timeit(lambda: ops.forward_warp(ffw)), npx * 12))
    rows.append(("forward_warp_grad", timeit(lambda: ops.forward_warp_grad(wl.border, ffw)), npx * 16))
    # SURVEY 8d "adversarial" set: i.i.d. U(-30,30) flows (worst-case gather / splat locality)
    adv = (torch.rand((B, H, W, 2), device=ffw.device) - 0.5) * 60.0
    rows.append(("image_warp C3, i.i.d. +-30 px flows", timeit(lambda: image_warp(wl.im2, adv)), npx * 32))
    rows.append(("backward_warp C3, i.i.d. +-30 px flows", timeit(lambda: ops.backward_warp(wl.im2, adv)), npx * 32))
    rows.append(("forward_warp, i.i.d. +-30 px flows", timeit(lambda: ops.forward_warp(adv)), npx * 12))
    x = im2w.clone().requires_grad_(True)

    def tern():
        return losses.ternary_loss(wl.im1, x, wl.border, max_distance=3)
    rows.append(("ternary fwd 7x7", timeit(tern), npx * 28))
    loss = tern()
    rows.append(("ternary bwd 7x7", timeit(lambda: torch.autograd.grad(loss, x, retain_graph=True)), npx * 40))
    for name, fn in (("smoothness_loss", losses.smoothness_loss), ("second_order_loss", losses.second_order_loss)):
        f = ffw.clone().requires_grad_(True)
        rows.append((name + " fwd", timeit(lambda: fn(f)), npx * 8))
        ls = fn(f)
        rows.append((name + " bwd", timeit(lambda: torch.autograd.grad(ls, f, retain_graph=True)), npx * 16))
    gl = losses.gradient_loss(wl.im1, x, wl.border)
    rows.append(("gradient_loss fwd", timeit(lambda: losses.gradient_loss(wl.im1, x, wl.border)), npx * 28))
    rows.append(("gradient_loss bwd", timeit(lambda: torch.autograd.grad(gl, x, retain_graph=True)), npx * 40))
    # the 5-level pyramid variant with the reference's [train_kitti] weights
    from unflow_amd import unsupervised
    kitti = dict(pyramid_loss=True, border_mask=True, mask_occlusion="fb", ternary_weight=1.0,
                 smooth_2nd_weight=3.0, fb_weight=0.2, occ_weight=12.4)
    g = torch.Generator(device=ffw.device)
    g.manual_seed(7)
    pf = [(torch.randn((B, H >> (2 + i), W >> (2 + i), 2), device=ffw.device, generator=g) * 0.3).requires_grad_(True)
          for i in range(5)]
    pb = [(-f.detach() + 0.02 * torch.randn(f.shape, device=ffw.device, generator=g)).requires_grad_(True) for f in pf]

    def pyr():
        loss, _ = unsupervised.pyramid_loss(wl.im1, wl.im2, wl.border, pf, pb, kitti)
        loss.backward()
    rows.append(("pyramid loss, 5 levels H/4..H/64, kitti weights, fwd+bwd", timeit(pyr), 0))
    from unflow_amd import graph as ugraph

    def pyr_step():
        loss, _ = unsupervised.pyramid_loss(wl.im1, wl.im2, wl.border, pf, pb, kitti)
        return (loss,) + torch.autograd.grad(loss, pf + pb)
    gs = ugraph.GraphedStep(pyr_step)
    rows.append(("  the same as one HIP graph (unflow_amd.graph.GraphedStep)", timeit(gs.replay), 0))
    from unflow_amd import flownet
    flr = torch.randn((B, H // 4, W // 4, 2), device=ffw.device, generator=g) * 0.3
    rows.append(("stacked_inputs (resize+warp+abs+concat, 14 ch)", timeit(lambda: flownet.stacked_inputs(wl.im1, wl.im2, flr)), npx * 80))
    rows.append(("fb_occlusion", timeit(lambda: losses.fb_occlusion(ffw, fbw)), npx * 28))
    rows.append(("downsample x2 C3", timeit(lambda: ops.downsample(wl.im1, 2)), npx * 15))
    a = torch.empty(256 * 1024 * 1024, device=ffw.device)
    b = torch.empty_like(a)
    cp = timeit(lambda: b.copy_(a))
    rows.append(("d2d copy 1 GiB (measured HBM roofline)", cp, 2 * a.numel() * 4))
    for name, ms, nbytes in rows:
        print("%-42s %9.3f ms  %8.1f GB/s algorithmic" % (name, ms, nbytes / ms / 1e6), file=sys.stderr)


if __name__ == "__main__":
    main()
