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

``--gpus N`` with N > 1 and no launcher environment (WORLD_SIZE unset) makes this
process a pure launcher: it starts N fresh child processes (one rank per GPU, the
reference splits its batch the same way, src/run.py:48) BEFORE it touches the GPU
itself, waits for them and exits with their status.  Under ``torch.distributed.run``
the ranks come from the environment.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
F16_MATRIX_PEAK_TF = 2516.6  # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak (16x the f32 matrix rate, "~2.5 PF")
KITTI = dict(pyramid_loss=True, border_mask=True, mask_occlusion="fb", ternary_weight=1.0,
             smooth_2nd_weight=3.0, fb_weight=0.2, occ_weight=12.4)   # config_template/config.ini:79,98,123,166-174


# ============================================================================= launcher
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="image pairs per GPU (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-runs", type=int, default=5, help="timed repetitions of the CPU baseline (median)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the informational fields (warps roofline, kitti_config_step, extensions)")
    ap.add_argument("--breakdown", action="store_true", help="also print per-op timings to stderr")
    ap.add_argument("--backend", default=None, choices=("nccl", "gloo"),
                    help="process-group backend (default: nccl = RCCL)")
    ap.add_argument("--dry", action="store_true",
                    help="launch and rendezvous only: no GPU work (CPU rehearsal of the N-rank launch path)")
    ap.add_argument("--config", default="cfg3", choices=("cfg2", "cfg3", "cfg5"),
                    help="BASELINE.json config: cfg3 (default, the headline: full loss path, batch 32, fp32); cfg2 "
                         "(batch 8: correlation + bidirectional backward warp, fp32) and cfg5 (768x1280 batch 16, fp16 "
                         "on the f16 matrix cores) print informational lines, single GPU")
    return ap.parse_args(argv)


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpu_count():
    """GPUs this process tree may use, counted WITHOUT any HIP / torch.cuda call (the launcher parent must not
    initialise the runtime before it forks the ranks): the visibility variables if set, else the KFD topology
    (/sys/class/kfd/kfd/topology/nodes/*/properties: a node with simd_count > 0 is a GPU).  None if neither
    source is readable -- the children then find out themselves (torch.cuda.set_device fails with a status
    the launcher propagates)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(base)
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(base, node, "properties")) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    return n


def spawn_ranks(args):
    """Start args.gpus children, one rank each, and return the worst exit status.  The
    parent never initialises the GPU (no HIP call and no torch import at all) and never
    re-execs: children are fresh interpreters."""
    n = args.gpus
    if not args.dry:
        have = visible_gpu_count()
        if have is not None and have < n and not os.environ.get("UNFLOW_BENCH_SHARE_GPU"):
            print("bench.py: --gpus %d but only %d visible" % (n, have), file=sys.stderr)
            return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    worst = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0:
                worst = worst or rc
                print("bench.py: rank %d exited with status %d; stopping the others" % (r, rc), file=sys.stderr)
                for o in alive:
                    procs[o].terminate()
        time.sleep(0.05)
    return worst


# ============================================================================= workload
def smooth_images(torch, gen, B):
    """Low-pass filtered uniform noise in [0,1] built on the device."""
    x = torch.rand((B, 3, H // 8, W // 8), device=gen.device, generator=gen)
    x = torch.nn.functional.interpolate(x, size=(H, W), mode="bicubic", align_corners=False)
    x = x + 0.05 * torch.rand((B, 3, H, W), device=gen.device, generator=gen)
    x = (x - x.amin()) / (x.amax() - x.amin())
    return x.permute(0, 2, 3, 1).contiguous()


def smooth_flow(torch, gen, B):
    """Smooth field ~N(0, 8 px) plus a per-sample translation U(-20,20)."""
    x = torch.randn((B, 2, H // 32, W // 32), device=gen.device, generator=gen) * 8.0
    x = torch.nn.functional.interpolate(x, size=(H, W), mode="bicubic", align_corners=False)
    t = (torch.rand((B, 2, 1, 1), device=gen.device, generator=gen) - 0.5) * 40.0
    return (x + t).permute(0, 2, 3, 1).contiguous()


class Workload:
    """The synthetic batch of one rank and the step over it."""

    def __init__(self, B, dev, seed, global_batch, features=True):
        import torch
        from unflow_amd import dist as udist
        from unflow_amd import losses
        self.torch = torch
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        self.B, self.global_batch = B, global_batch
        lrelu = torch.nn.functional.leaky_relu
        # the generator is consumed in this order whatever `features` is, so the images and flows
        # of a rank are the same tensors with and without the correlation inputs
        fa = lrelu(torch.randn((B, FEAT_C, H8, W8), device=dev, generator=gen), 0.1)
        fb = lrelu(torch.randn((B, FEAT_C, H8, W8), device=dev, generator=gen), 0.1)
        top_fw = torch.randn((B, D, H8, W8), device=dev, generator=gen)
        top_bw = torch.randn((B, D, H8, W8), device=dev, generator=gen)
        if features:
            self.fa, self.fb, self.top_fw, self.top_bw = fa, fb, top_fw, top_bw
        del fa, fb, top_fw, top_bw
        self.im1 = smooth_images(torch, gen, B)
        self.im2 = smooth_images(torch, gen, B)
        self.flow_fw = smooth_flow(torch, gen, B).requires_grad_(True)
        self.flow_bw = (-self.flow_fw.detach() + 0.5 * smooth_flow(torch, gen, B) * 0.1).requires_grad_(True)
        self.border = losses.create_border_mask(self.im1, 0.1)
        self.grad_buffer = udist.FlatGradBuffer(GRAD_BUFFER_FLOATS, dev) if features else None
        self.corr_events = []   # (start, end, kind, calls between them): HIP events on the launch stream

    def _mark(self):
        """One HIP event on the launch stream (the stream the ABI is given).  An event record costs the stream a
        marker packet and ~5-10 us between two kernels (rocprofv3 trace of the step), so the step brackets each PAIR of
        correlation calls with ONE event on either side -- three per step -- instead of two per call."""
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def loss_leg(self):
        """The loss half of the step: level-0 compute_losses (fused census/warp and occlusion
        kernels) forward, backward to both flows.  tests/test_gpu_fullsize.py checks exactly
        this call at this size against the composed HIP path and the oracle."""
        from unflow_amd import losses
        self.flow_fw.grad = None
        self.flow_bw.grad = None
        ls = losses.compute_losses(self.im1, self.im2, self.flow_fw, self.flow_bw,
                                   border_mask=self.border, mask_occlusion="fb",
                                   data_max_distance=3, norm_batch=self.global_batch,
                                   terms=("ternary", "sym"))
        ls["ternary"].backward()
        return ls

    def step(self, world, bidirectional=False, serialized=False):
        from unflow_amd import dist as udist
        from unflow_amd import ops
        torch = self.torch
        order = self.step_order = []
        with torch.no_grad():
            if bidirectional:
                # extension: corr(b, a) is a bit-exact re-indexing of corr(a, b), and one
                # backward call on top_fw + flip(top_bw) returns the summed gradients
                c_fw = ops.correlation(self.fa, self.fb)
                c_bw = ops.correlation_flip(c_fw)
                g_fw = ops.correlation_grad(ops.correlation_flip(self.top_bw, self.top_fw), self.fa, self.fb)
                g_bw = None
            else:
                e0 = self._mark()
                c_fw = ops.correlation(self.fa, self.fb)
                c_bw = ops.correlation(self.fb, self.fa)
                e1 = self._mark()
                g_fw = ops.correlation_grad(self.top_fw, self.fa, self.fb)
                g_bw = ops.correlation_grad(self.top_bw, self.fb, self.fa)
                e2 = self._mark()
                self.corr_events += [(e0, e1, "corr_fwd", 2), (e1, e2, "corr_bwd", 2)]    # (start, end, kind, calls)
            order += ["corr_fwd", "corr_fwd", "corr_bwd", "corr_bwd"]
        # N > 1: the gradient buffer stands for the FlowNetC weights' gradient, which a trainer has at the
        # earliest when the correlation backward is through (core/train.py:446-480 averages after the whole
        # backward pass).  So its all-reduce is started HERE: an event recorded behind the second
        # correlation_grad, the collective's stream waits for it, and what is left of the step -- the loss
        # leg -- is all that can overlap the exchange; the step ends with the launch stream waiting (no
        # host synchronisation) for the collective.  serialized=True waits at once (no overlap at all).
        pending = None
        if world > 1:
            pending = self.grad_buffer.all_reduce_async(after=udist.record_event())
            order.append("allreduce_start")
            if serialized:
                pending.wait()
                pending = None
                order.append("allreduce_wait")
        ls = self.loss_leg()
        order.append("loss_leg")
        loss = ls["ternary"].detach()
        if world > 1:    # the reference's average_gradients (core/train.py:446-480) as two RCCL all-reduces
            loss = udist.reduce_losses({"ternary": loss})["ternary"]
            order.append("loss_allreduce")
            if pending is not None:
                pending.wait()
                order.append("allreduce_wait")
        return loss, (c_fw, c_bw, g_fw, g_bw)

    def composed_loss(self):
        """The same loss through the UNFUSED HIP ops (image_warp -> fb_occlusion -> ternary_loss):
        the in-run check of the fused kernels the step times."""
        from unflow_amd import losses
        from unflow_amd.image_warp import image_warp
        with self.torch.no_grad():
            npx = self.global_batch * H * W
            m_fw, m_bw = self.border.clone(), self.border.clone()
            losses.fb_occlusion(self.flow_fw, image_warp(self.flow_bw, self.flow_fw), m_fw)
            losses.fb_occlusion(self.flow_bw, image_warp(self.flow_fw, self.flow_bw), m_bw)
            return (losses.ternary_loss(self.im1, image_warp(self.im2, self.flow_fw), m_fw, 3, norm_count=npx) +
                    losses.ternary_loss(self.im2, image_warp(self.im1, self.flow_bw), m_bw, 3, norm_count=npx))


# ============================================================================= CPU baseline (oracle = checker + timed sample)
def cpu_baseline(wl, n_pairs=2, runs=5):
    """The CPU oracle (a restatement: kind "port") on the FIRST n_pairs pairs of the GPU
    workload (same smooth fields), one warm-up + the median of `runs` timed repetitions,
    with all host threads and with one.  The same oracle outputs double as a full-size parity
    check of the HIP path on those pairs (loss and cost volume)."""
    import numpy as np
    import oracle
    from unflow_amd import losses, ops
    torch = wl.torch
    oracle.build()
    all_threads = oracle.num_threads()

    def host(t, n):
        return t[:n].detach().cpu().numpy()

    def one_pass(n):
        f0, f1, top = host(wl.fa, n), host(wl.fb, n), host(wl.top_fw, n)
        im1, im2, ffw, fbw, border = (host(t, n) for t in (wl.im1, wl.im2, wl.flow_fw, wl.flow_bw, wl.border))
        t0 = time.perf_counter()
        c_first = None
        for a, b in ((f0, f1), (f1, f0)):
            c, p0, p1 = oracle.correlation(a, b)
            oracle.correlation_grad(top, a, b, p0, p1)
            c_first = c if c_first is None else c_first
        ls, aux = oracle.compute_losses(im1, im2, ffw, fbw, border, "fb", 3, terms=("ternary", "sym"))
        g2 = oracle.ternary_loss_grad(im1, aux["im2_warped"], aux["mask_fw"], 3)
        oracle.image_warp_grad(g2, im2, ffw, want_im=False)
        g1 = oracle.ternary_loss_grad(im2, aux["im1_warped"], aux["mask_bw"], 3)
        oracle.image_warp_grad(g1, im1, fbw, want_im=False)
        return time.perf_counter() - t0, ls, c_first

    def median_time(n, threads):
        oracle.set_num_threads(threads)
        one_pass(n)                               # warm-up (pages in the library and the thread pool)
        times, last = [], None
        for _ in range(runs):
            dt, ls, c = one_pass(n)
            times.append(dt)
            last = (ls, c)
        return float(np.median(times)), times, last

    t_all, times_all, (ls, c_ref) = median_time(n_pairs, all_threads)
    t_one, times_one, _ = median_time(1, 1)
    oracle.set_num_threads(all_threads)

    # parity of the HIP path on the same pairs, at full size
    with torch.no_grad():
        c_gpu = ops.correlation(wl.fa[:n_pairs], wl.fb[:n_pairs]).cpu().numpy()
    corr_err = float(np.abs(c_gpu - c_ref).max())
    ffw = wl.flow_fw[:n_pairs].detach().clone().requires_grad_(True)
    fbw = wl.flow_bw[:n_pairs].detach().clone().requires_grad_(True)
    got = losses.compute_losses(wl.im1[:n_pairs], wl.im2[:n_pairs], ffw, fbw, border_mask=wl.border[:n_pairs],
                                mask_occlusion="fb", data_max_distance=3, terms=("ternary", "sym"))
    rel = {k: abs(float(got[k].detach()) - ls[k]) / abs(ls[k]) for k in ("ternary", "sym")}
    if corr_err > 1e-5 or rel["ternary"] > 2e-4 or rel["sym"] > 2e-4:
        raise AssertionError("bench: HIP path and CPU oracle disagree at full size: corr max abs err %.3g, "
                             "loss rel err %s" % (corr_err, rel))
    return {"value": n_pairs / t_all, "unit": "image-pairs/s", "cores": all_threads, "kind": "port",
            "sample": "the first %d pairs of the GPU workload (384x1280, same smooth fields; same step: 2 corr fwd+bwd, "
                      "level-0 loss fwd+bwd), 1 warm-up + median of %d runs (%.2f s each), OpenMP C restatement "
                      "oracle/unflow_oracle.c" % (n_pairs, runs, t_all),
            "runs_s": [round(t, 3) for t in times_all],
            "single_thread": {"value": 1.0 / t_one, "unit": "image-pairs/s", "cores": 1,
                              "sample": "the first pair, 1 warm-up + median of %d runs" % runs,
                              "runs_s": [round(t, 3) for t in times_one]},
            "parity_check": {"pairs": n_pairs, "corr_fwd_max_abs_err": corr_err, "loss_rel_err": rel,
                             "loss_oracle": {k: float(ls[k]) for k in ("ternary", "sym")}}}


# ============================================================================= measurements beside the headline
def event_time(torch, fn, n=5, warm=1):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def stream_copy_gbs(torch, dev, gib=1):
    """Rate of the library's hand-written float4 streaming copy over `gib` GiB (read + write bytes)."""
    import ctypes
    from unflow_amd import _lib
    n = gib * 256 * 1024 * 1024
    a = torch.empty(n, device=dev)
    b = torch.empty_like(a)
    a.normal_()
    lib = _lib.load()

    def go():
        _lib.check(lib.unflow_stream_copy(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                          ctypes.c_size_t(n * 4), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    ms = event_time(torch, go, n=5, warm=2)
    assert torch.equal(a[:4096], b[:4096]) and torch.equal(a[-4096:], b[-4096:])
    return 2 * n * 4 / (ms * 1e-3) / 1e9


def warps_roofline(wl, copy_gbs):
    """In-run HIP-event timings of the HBM-bound warps at the step's batch (north_star names
    backward-warp): algorithmic bytes / time, as fractions of the measured copy and of 8 TB/s."""
    from unflow_amd import ops
    from unflow_amd.image_warp import image_warp, image_warp_grad
    torch = wl.torch
    npx = wl.B * H * W
    ffw = wl.flow_fw.detach()
    g3 = torch.randn_like(wl.im2)
    rows = {}
    for name, fn, bpp in (("backward_warp", lambda: ops.backward_warp(wl.im2, ffw), 32),
                          ("backward_warp_grad", lambda: ops.backward_warp_grad(g3, wl.im2, ffw), 40),
                          ("image_warp", lambda: image_warp(wl.im2, ffw), 32),
                          ("image_warp_grad_flow", lambda: image_warp_grad(g3, wl.im2, ffw, False, True), 40)):
        ms = event_time(torch, fn, n=10, warm=2)
        gbs = npx * bpp / (ms * 1e-3) / 1e9
        rows[name] = {"ms": ms, "bytes_per_px": bpp, "gbs": gbs, "frac_of_copy": gbs / copy_gbs,
                      "frac_of_8TBs": gbs / HBM_PEAK_GBS}
    return rows


def loss_kernels_roofline(wl, copy_gbs):
    """The loss kernels of the step, each against the ceiling that bounds it (in-run HIP-event times at the step's batch):
      * soft census forward / backward, fused with the warp (ternary_*_pair_kernel<3,16,true>): VALU ISSUE.  A wave walks
        19 rows x 24 pair taps x 3 transcendentals (8 issue cycles each) among `valu_insts_per_wave` vector instructions
        (4 cycles each; the count is the PMC's SQ_INSTS_VALU / SQ_WAVES of profiles/r04_pmc_loss.json); floor = that
        issue time over 1024 SIMDs at 2.4 GHz.  `pmc_valu_issue_frac` is the measured SQ_ACTIVE_INST_VALU share.
      * forward_warp: LDS integer atomics.  floor = 81 taps per source pixel as wave instructions x 5.3 cycles (the
        conflict-free ds_add_u32 rate per CU, profiles/ubench_lds_atomic.txt) over 256 CUs.
      * fb-occlusion (+ warp of the other flow) and the sym reduction: HBM, algorithmic bytes / time."""
    from unflow_amd import losses, ops
    torch = wl.torch
    B = wl.B
    npx = B * H * W
    ffw, fbw = wl.flow_fw.detach(), wl.flow_bw.detach()
    pmc = {}
    path = os.path.join(ROOT, "profiles", "r04_pmc_loss.json")      # (collected at the end of round 4; round 3's as a fallback)
    if not os.path.exists(path):
        path = os.path.join(ROOT, "profiles", "r03_pmc_loss.json")
    if os.path.exists(path):
        with open(path) as fh:
            pmc = json.load(fh)
    clock = 2.4e9
    rows = {}
    f = ffw.clone().requires_grad_(True)

    def census_fwd():
        return losses.ternary_loss_warped(wl.im1, wl.im2, f, wl.border, max_distance=3)
    ms_f = event_time(torch, census_fwd, n=10, warm=2)
    loss = census_fwd()
    ms_b = event_time(torch, lambda: torch.autograd.grad(loss, f, retain_graph=True), n=10, warm=2)
    for name, ms, key in (("census_fwd_fused_warp", ms_f, "ternary_fwd_pair_kernel<3, 16, true>"),
                          ("census_bwd_fused_warp", ms_b, "ternary_bwd_pair_kernel<3, 16, true>")):
        rec = pmc.get(key, {})
        ipw = rec.get("valu_insts_per_wave")
        row = {"ms": ms, "bound": "valu issue", "pmc_valu_issue_frac": rec.get("valu_issue_frac"),
               "pmc_source": os.path.relpath(path, ROOT) if rec else None}
        if ipw:
            trans = 19 * 24 * 3
            floor_ms = rec["waves"] * ((ipw - trans) * 4 + trans * 8) / 1024 / clock * 1e3 * (B / 32.0)
            row.update(valu_insts_per_wave=ipw, transcendentals_per_wave=trans, floor_ms=floor_ms, frac=floor_ms / ms)
        rows[name] = row
    ms = event_time(torch, lambda: ops.forward_warp(ffw), n=10, warm=2)
    floor_ms = (npx / 64) * 81 * 5.3 / 256 / clock * 1e3
    rows["forward_warp"] = {"ms": ms, "bound": "lds integer atomics", "floor_ms": floor_ms, "frac": floor_ms / ms,
                            "model": "81 taps/pixel as ds_add_u32 wave instructions x 5.3 cycles per CU (conflict free), 256 CUs, 2.4 GHz"}
    m = wl.border.clone()
    ms = event_time(torch, lambda: losses.fb_occlusion_warp(ffw, fbw, m, mask_in=wl.border), n=10, warm=2)
    gbs = npx * 24 / (ms * 1e-3) / 1e9
    rows["fb_occlusion_warp"] = {"ms": ms, "bound": "hbm", "bytes_per_px": 24, "gbs": gbs, "frac_of_copy": gbs / copy_gbs,
                                 "frac": gbs / HBM_PEAK_GBS}
    dens = ops.forward_warp(fbw)
    ms = event_time(torch, lambda: losses.sym_loss(m, dens, norm_count=npx), n=10, warm=2)
    gbs = npx * 8 / (ms * 1e-3) / 1e9
    rows["sym_loss"] = {"ms": ms, "bound": "hbm", "bytes_per_px": 8, "gbs": gbs, "frac_of_copy": gbs / copy_gbs, "frac": gbs / HBM_PEAK_GBS}
    return rows


def kitti_config_step(wl):
    """The reference's default KITTI loss (config_template/config.ini:79,98,123,166-174): 5-level
    pyramid H/4..H/64, ternary + 3 smooth_2nd + 0.2 fb + 12.4 occ, fb occlusion masks, fwd+bwd to
    every level's flows -- eager and as one HIP graph (informational, never `value`)."""
    from unflow_amd import graph as ugraph
    from unflow_amd import unsupervised
    torch = wl.torch
    B = wl.B
    pf, pb = kitti_pyramid_flows(torch, wl.im1.device, B)

    def step():
        loss, _ = unsupervised.pyramid_loss(wl.im1, wl.im2, wl.border, pf, pb, KITTI)
        return (loss,) + torch.autograd.grad(loss, pf + pb)
    # (the first eager steps also fill the caching allocator with the pyramid's temporaries: 5 warm-up steps.  One of the
    # first blocks of ten steps of a process takes 5-9 ms per step on the host side -- tools/eager_blocks.py: 1.9, 8.8, 1.4,
    # 1.4, 1.4, 1.4 ms; it is a full collection of Python's cyclic garbage collector: with gc.disable() every block takes
    # 1.3 ms -- so three blocks are timed, all of them reported, and the steady state is their minimum.)
    eager_blocks = [event_time(torch, step, n=10, warm=5 if k == 0 else 0) for k in range(3)]
    eager_ms = min(eager_blocks)
    eager = [t.detach().clone() for t in step()]
    gs = ugraph.GraphedStep(step)
    graph_ms = event_time(torch, gs.replay, n=10, warm=2)
    # the replayed graph computes what the eager step computes: the loss to 1e-6, every level's flow gradients to 1e-6 of
    # their scale (the same launches in the same order; the scatter-adds flush with float atomics, so not bit for bit).
    # tests/test_gpu_fullsize.py checks one pair of this step against the oracle.
    replayed = gs.replay()
    torch.cuda.synchronize()
    for a, b in zip(eager, replayed):
        if not float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()):
            raise AssertionError("bench: the graph replay of the KITTI pyramid step differs from the eager step")
    return {"note": "NOT the headline: 5-level pyramid loss of the reference's [train_kitti] config "
                    "(ternary + 3*smooth_2nd + 0.2*fb + 12.4*occ, mask_occlusion=fb), fwd+bwd, batch %d" % B,
            "eager_ms": eager_ms, "eager_ms_blocks_of_10": eager_blocks, "graph_ms": graph_ms, "pairs_per_s_graph": B / (graph_ms * 1e-3),
            "loss": float(eager[0]), "graph_matches_eager_to": 1e-6}


def kitti_pyramid_flows(torch, dev, B):
    """The five flow levels (H/4 .. H/64, both directions) of kitti_config_step, seeded."""
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    pf = [(torch.randn((B, H >> (2 + i), W >> (2 + i), 2), device=dev, generator=g) * 0.3).requires_grad_(True)
          for i in range(5)]
    pb = [(-f.detach() + 0.02 * torch.randn(f.shape, device=dev, generator=g)).requires_grad_(True) for f in pf]
    return pf, pb


TRAFFIC_FILE = "profiles/r04_traffic.json"
KERNEL_SOURCES = ("correlation_bwd4.hip", "corr_bwd4_asm.inc", "correlation_fwd2.hip", "corr_fwd2_asm.inc", "corr_mfma_common.h")


def correlation_kernel_source_sha16():
    """Hash of the sources of the shipped float32 correlation kernels (forward and backward, with their generated
    instruction blocks): a traffic record is only valid for the source it was collected on."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        path = os.path.join(ROOT, "unflow_amd", "csrc", name)
        h.update(open(path, "rb").read() if os.path.exists(path) else b"?")
    return h.hexdigest()[:16]


def traffic_record():
    """HBM bytes of the dominant kernel from PMC counters.  bench.py cannot sit under
    ``rocprofv3 --pmc`` and time itself, so the counters come from a separate pass
    (tools/profile_round.sh -> profiles/r04_traffic.json); the record carries the hash of the
    kernel source it was collected on and is only used when that is the source of this build."""
    path = os.path.join(ROOT, TRAFFIC_FILE)
    if not os.path.exists(path):
        return None, None
    with open(path) as fh:
        rec = json.load(fh)
    sha = correlation_kernel_source_sha16()
    if rec.get("kernel_source_sha16") != sha:
        return None, {"file": TRAFFIC_FILE, "stale": True,
                      "note": "collected on another version of the correlation kernels' source: not reported"}
    return rec["corr_bwd_both_launches"]["hbm_bytes"], {
        "file": TRAFFIC_FILE, "kernel_source_sha16": sha,
        "how": "offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/profile_round.sh), FETCH_SIZE doubled "
               "(gfx950), per call = both gradient launches; not measured in this run"}


# ============================================================================= the other single-GPU configs (informational)
def side_config(config, batch, steps, warmup):
    """BASELINE.json configs[1] (cfg2) and configs[4] (cfg5) as bench records of their own: same contract, single
    GPU, never the headline (`--config cfg2|cfg5` prints one as a line; the default run attaches both to the headline
    line as `side_configs`).  cfg2 = SURVEY 8d: B=8, 2 FlowNetC correlation calls (fw and bw direction) on
    [8,256,48,160] + bidirectional backward_warp (2 calls on [8,384,1280,3]); image_warp (what the loss calls)
    is timed beside it.  cfg5 = B=16, 768x1280, fp16 tensors: 2 correlation calls on [16,256,96,160] on the f16
    matrix cores with their 2 backward calls, 2 forward_warp of half flows, 4 full-resolution image_warp of half
    images (core/flownet.py:27-70: the stacked nets warp at full resolution)."""
    import torch
    from unflow_amd import ops
    from unflow_amd.image_warp import image_warp
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    gen = torch.Generator(device=dev)
    lrelu = torch.nn.functional.leaky_relu
    global H, W
    if config == "cfg2":
        B = 8 if batch == 32 else batch
        gen.manual_seed(1234 + 2)
        fa = lrelu(torch.randn((B, FEAT_C, H8, W8), device=dev, generator=gen), 0.1)
        fb = lrelu(torch.randn((B, FEAT_C, H8, W8), device=dev, generator=gen), 0.1)
        im1, im2 = smooth_images(torch, gen, B), smooth_images(torch, gen, B)
        ffw = smooth_flow(torch, gen, B)
        fbw = -ffw + 0.05 * smooth_flow(torch, gen, B)
        parts = {"corr_fwd": lambda: (ops.correlation(fa, fb), ops.correlation(fb, fa)),
                 "backward_warp": lambda: (ops.backward_warp(im2, ffw), ops.backward_warp(im1, fbw)),
                 "image_warp": lambda: (image_warp(im2, ffw), image_warp(im1, fbw))}

        def step():
            with torch.no_grad():
                parts["corr_fwd"]()
                parts["backward_warp"]()
        dtype, hh, ww = "f32", H, W
        work = ("cfg2: 384x1280 pairs, batch %d: 2x FlowNetC correlation forward on [B,256,48,160] (fw and bw direction) + "
                "bidirectional backward_warp (2 calls on [B,384,1280,3])" % B)
        alg = {"corr_fwd": (2 * CORR_FWD_FLOP * B, 2 * CORR_FWD_BYTES * B), "backward_warp": (0, 2 * B * H * W * 32),
               "image_warp": (0, 2 * B * H * W * 32)}
        peak_tf = FP32_MATRIX_PEAK_TF
    else:
        B = 16 if batch == 32 else batch
        hh, ww = 768, 1280
        h8, w8 = hh // 8, ww // 8
        gen.manual_seed(1234 + 5)
        fa = lrelu(torch.randn((B, FEAT_C, h8, w8), device=dev, generator=gen), 0.1).half()
        fb = lrelu(torch.randn((B, FEAT_C, h8, w8), device=dev, generator=gen), 0.1).half()
        top_fw = torch.randn((B, D, h8, w8), device=dev, generator=gen).half()
        top_bw = torch.randn((B, D, h8, w8), device=dev, generator=gen).half()
        H, W = hh, ww                                   # smooth_images / smooth_flow build at the module's extents
        im1, im2 = smooth_images(torch, gen, B).half(), smooth_images(torch, gen, B).half()
        ffw32 = smooth_flow(torch, gen, B)
        ffw, fbw = ffw32.half(), (-ffw32 + 0.05 * smooth_flow(torch, gen, B)).half()
        del ffw32
        H, W = 384, 1280
        parts = {"corr_fwd": lambda: (ops.correlation_fp16(fa, fb), ops.correlation_fp16(fb, fa)),
                 "corr_bwd": lambda: (ops.correlation_fp16_grad(top_fw, fa, fb), ops.correlation_fp16_grad(top_bw, fb, fa)),
                 "forward_warp": lambda: (ops.forward_warp_fp16(ffw), ops.forward_warp_fp16(fbw)),
                 "image_warp": lambda: (ops.image_warp_fp16(im2, ffw), ops.image_warp_fp16(im1, fbw),
                                        ops.image_warp_fp16(im2, fbw), ops.image_warp_fp16(im1, ffw))}

        def step():
            with torch.no_grad():
                for k in ("corr_fwd", "corr_bwd", "forward_warp", "image_warp"):
                    parts[k]()
        dtype = "f16 (f32 accumulate)"
        work = ("cfg5: 768x1280 pairs, batch %d, fp16 tensors: 2x FlowNetC correlation fwd+bwd on [B,256,96,160] on the f16 "
                "matrix cores, 2x forward_warp of half flows, 4x full-resolution image_warp of half images" % B)
        fl = 2 * D * h8 * w8 * FEAT_C
        alg = {"corr_fwd": (2 * fl * B, 2 * 2 * h8 * w8 * (2 * FEAT_C + D) * B),
               "corr_bwd": (4 * fl * B, 2 * 2 * h8 * w8 * (D + 4 * FEAT_C) * B),
               "forward_warp": (0, 2 * B * hh * ww * 8), "image_warp": (0, 4 * B * hh * ww * 16)}
        peak_tf = F16_MATRIX_PEAK_TF
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = {}
    with torch.no_grad():
        for k, fn in parts.items():
            ms = event_time(torch, fn, n=10, warm=2)
            fl, by = alg[k]
            per[k] = {"ms": ms, "hbm_gbs": by / (ms * 1e-3) / 1e9, "frac_of_8TBs": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            if fl:
                per[k].update(tflops=fl / (ms * 1e-3) / 1e12, frac_of_matrix_peak=fl / (ms * 1e-3) / 1e12 / peak_tf)
    dom = "corr_bwd" if "corr_bwd" in per else "corr_fwd"
    # these kernels are priced against BOTH ceilings: on the f16 matrix cores the correlation's algorithmic
    # traffic at 8 TB/s takes longer than its flops at the dense f16 peak, so cfg5 is HBM-bound
    hbm_bound = per[dom]["frac_of_8TBs"] > per[dom].get("frac_of_matrix_peak", 0.0)
    out = {"metric": "image-pairs/sec, BASELINE %s (informational; the headline is cfg3)" % config,
           "value": B * steps / dt, "unit": "image-pairs/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": dtype, "data": "synthetic", "informational": True,
           "config": {"workload": work, "global_batch": B, "height": hh, "width": ww, "parallelism": "single GPU"},
           "roofline": {"bound": "hbm" if hbm_bound else "mfma", "kernel": dom + " (both calls of the step)",
                        "achieved": per[dom]["hbm_gbs"] if hbm_bound else per[dom]["tflops"],
                        "peak": HBM_PEAK_GBS if hbm_bound else peak_tf, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                        "frac": per[dom]["frac_of_8TBs"] if hbm_bound else per[dom]["frac_of_matrix_peak"],
                        "traffic": None, "frac_basis": "HIP events around the calls on the launch stream; algorithmic bytes / flops of SURVEY 8d",
                        "parts": per}}
    return out


def run_side_config(args):
    print(json.dumps(side_config(args.config, args.batch, args.steps, args.warmup)))
    return 0


# ============================================================================= main
def run_dry(args):
    """Launch-path rehearsal without a GPU: rendezvous, one all-reduce, one JSON line."""
    import torch
    from unflow_amd import dist as udist
    if os.environ.get("UNFLOW_BENCH_FAIL_RANK") == os.environ.get("RANK", "0"):   # test hook: a rank that dies
        sys.exit(17)
    rank, world = udist.init_process_group(args.backend or "gloo")
    t = torch.tensor([float(rank + 1)])
    udist.all_reduce_sum_(t)
    assert float(t) == world * (world + 1) / 2
    start, stop = udist.shard_range(args.batch * world, rank, world)
    assert stop - start == args.batch
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "value": 0.0, "n_gpus": world, "dry": True,
                          "gpus_requested": args.gpus, "backend": args.backend or "gloo",
                          "allreduce_check": float(t)}))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.dry:
        sys.exit(run_dry(args))
    if args.config != "cfg3":
        sys.exit(run_side_config(args))

    import numpy as np
    import torch
    import torch.distributed as dist
    from unflow_amd import dist as udist

    rank, world = udist.init_process_group(args.backend)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1) and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d ranks; reporting n_gpus=%d"
              % (args.gpus, world, world), file=sys.stderr)
    # UNFLOW_BENCH_SHARE_GPU=1: rehearsal of the N-rank path on a one-GPU box (all ranks on cuda:0, use
    # --backend gloo: RCCL refuses two ranks on one device); the line then says so and is no measurement
    shared = bool(os.environ.get("UNFLOW_BENCH_SHARE_GPU"))
    dev = torch.device("cuda", 0 if shared else local_rank)
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

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # Python's cyclic garbage collector stays out of the timed loops, as in timeit: a full collection is a 40-70 ms pause of
    # the launching thread (tools/eager_blocks.py), and when it falls into the K timed steps the launch queue drains -- one
    # run of the round printed 6.08 ms per step with every kernel at its usual duration where the others print 5.36-5.58.
    # Nothing of the step is skipped: the collector runs right before the region and is switched on again after the loops.
    import gc
    gc.collect()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = wl.step(world)
    fence()
    dt = max_over_ranks(time.perf_counter() - t0)

    # dominant kernel: correlation backward (3.47 GFLOP/pair, f32 MFMA); averaged over the timed region
    per_kind = {}
    for s, e, kind, calls in wl.corr_events:
        per_kind.setdefault(kind, []).append(s.elapsed_time(e) / calls)      # per call
    bwd_ms = float(np.mean(per_kind["corr_bwd"]))
    fwd_ms = float(np.mean(per_kind["corr_fwd"]))

    # N > 1: the same step with the exchange waited for at once (no overlap) and the bare all-reduce, so that
    # whatever the scaling run prints is bounded on both sides
    dt_serial = allreduce_ms = None
    step_order = list(wl.step_order)
    if world > 1:
        fence()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            wl.step(world, serialized=True)
        fence()
        dt_serial = max_over_ranks(time.perf_counter() - t2)
        fence()
        t3 = time.perf_counter()
        for _ in range(5):
            wl.grad_buffer.all_reduce()
        fence()
        allreduce_ms = max_over_ranks(time.perf_counter() - t3) / 5 * 1e3

    dt_bidir = None
    if not args.no_extras:
        # for information only (never `value`): the same step with the bidirectional-correlation extension
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            wl.step(world, bidirectional=True)
        fence()
        dt_bidir = max_over_ranks(time.perf_counter() - t1)

    if gc_was_enabled:
        gc.enable()

    # in-run check of the fused loss kernels the step times (every rank checks its own shard)
    fused = float(wl.loss_leg()["ternary"].detach())
    composed = float(wl.composed_loss())
    if not abs(fused - composed) <= 1e-6 * abs(composed):
        raise AssertionError("bench: fused loss %.9g != composed HIP loss %.9g" % (fused, composed))

    if args.breakdown and rank == 0:
        breakdown(wl, fwd_ms, bwd_ms)

    if rank == 0:
        pairs = args.batch * world * args.steps
        achieved_tf = CORR_BWD_FLOP * args.batch / (bwd_ms * 1e-3) / 1e12
        traffic, traffic_src = traffic_record() if args.batch == 32 else (None, None)
        copy_gbs = stream_copy_gbs(torch, dev)
        out = {
            "metric": "image-pairs/sec (corr+warp+census fwd+bwd) at 384x1280 b32",
            "value": pairs / dt, "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg3: 384x1280 pairs, batch %d per GPU: 2x FlowNetC correlation fwd+bwd on "
                                   "[B,256,48,160] + level-0 compute_losses (4 image_warp evaluated inside the census and fb-occlusion "
                                   "kernels, 2 forward_warp + sym term, 2 ternary 7x7) fwd+bwd" % args.batch,
                       "global_batch": args.batch * world, "height": H, "width": W,
                       "parallelism": "batch-sharded x%d (one process per GPU), all-reduce of loss + 156.7 MB grad buffer "
                                      "(started behind the second correlation_grad, overlapping the loss leg only)" % world,
                       "world_size": world, "backend": dist.get_backend() if world > 1 else None,
                       "timing": "K steps between fences (synchronize + barrier); Python's garbage collector runs right before "
                                 "the region and is off inside it (as in timeit)",
                       **({"step_order": step_order} if world > 1 else {}),
                       **({"rehearsal_all_ranks_on_one_gpu": True} if shared else {})},
            "roofline": {"bound": "mfma",
                         "kernel": "correlation backward: corr_mfma_bwd4_kernel, both gradient launches of one call",
                         "achieved": achieved_tf, "peak": FP32_MATRIX_PEAK_TF, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP32_MATRIX_PEAK_TF,
                         "frac_basis": "HIP events on the launch stream around the step's two correlation_grad calls (one event before, one "
                                       "behind the pair: per call = half), mean over the timed region "
                                       "(rocprofv3 kernel averages of the same command: profiles/r04_bench_kernel_stats.csv)",
                         "traffic": traffic, "traffic_source": traffic_src,
                         "stream_copy_gbs": copy_gbs,
                         "avg_launch_ms": bwd_ms,
                         "hbm_gbs": CORR_BWD_BYTES * args.batch / (bwd_ms * 1e-3) / 1e9,
                         "hbm_frac_of_8TBs": CORR_BWD_BYTES * args.batch / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "corr_fwd": {"avg_launch_ms": fwd_ms,
                                      "tflops": CORR_FWD_FLOP * args.batch / (fwd_ms * 1e-3) / 1e12,
                                      "frac": CORR_FWD_FLOP * args.batch / (fwd_ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TF,
                                      "hbm_gbs": CORR_FWD_BYTES * args.batch / (fwd_ms * 1e-3) / 1e9}},
            **({"ms_per_step_serialized": dt_serial / args.steps * 1e3, "allreduce_ms": allreduce_ms,
                "allreduce_note": "ms_per_step: the 156.7 MB all-reduce starts when the correlation backward is through and "
                                  "overlaps the loss leg; ms_per_step_serialized: the same step waiting for it at once; "
                                  "allreduce_ms: the bare collective"} if world > 1 else {}),
            "loss": float(loss),
            "loss_check": {"fused": fused, "composed_hip": composed,
                           "note": "rank 0's shard through the fused kernels vs the unfused HIP ops, asserted to 1e-6; "
                                   "the oracle check on the first pairs is in cpu_baseline.parity_check"},
        }
        if not args.no_extras:
            out["roofline"]["warps"] = dict(
                warps_roofline(wl, copy_gbs), bound="hbm",
                note="algorithmic bytes / HIP-event time at B=%d; frac_of_copy is against stream_copy_gbs "
                     "(hand-written float4 copy, 1 GiB), frac_of_8TBs against the nominal HBM peak" % args.batch)
            out["roofline"]["loss_kernels"] = loss_kernels_roofline(wl, copy_gbs)
            out["extension_bidirectional_correlation"] = {
                "note": "NOT the headline: the same step when the caller obtains the (b,a) cost volume as a bit-exact "
                        "re-indexing of the (a,b) one and runs one backward call for both directions "
                        "(ops.correlation_bidirectional, DESIGN.md 4.1)",
                "ms_per_step": dt_bidir / args.steps * 1e3, "value": pairs / dt_bidir, "unit": "image-pairs/s"}
            out["kitti_config_step"] = kitti_config_step(wl)
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only: the other ranks would idle at the barrier
            out["cpu_baseline"] = cpu_baseline(wl, runs=args.cpu_runs)
        if not args.no_extras and world == 1 and args.batch == 32:
            # BASELINE configs[1] and configs[4], timed after the headline's region (a few seconds): the driver's record
            # carries them; `value` above stays cfg3
            del wl
            torch.cuda.empty_cache()
            out["side_configs"] = {}
            for cfg in ("cfg2", "cfg5"):
                sc = side_config(cfg, 32, args.steps, args.warmup)
                out["side_configs"][cfg] = {"workload": sc["config"]["workload"], "value": sc["value"], "unit": sc["unit"],
                                            "ms_per_step": sc["ms_per_step"], "dtype": sc["dtype"], "roofline": sc["roofline"]}
                torch.cuda.empty_cache()
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def breakdown(wl, fwd_ms, bwd_ms):
    """Per-op device time (HIP events on the launch stream), to stderr."""
    from unflow_amd import losses, ops
    from unflow_amd.image_warp import image_warp, image_warp_grad
    torch = wl.torch

    def timeit(fn, n=5):
        return event_time(torch, fn, n=n, warm=1)

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
    rows.append(("corr_fwd, fp16 on the f16 matrix cores (extension)", timeit(lambda: ops.correlation_fp16(fah, fbh)),
                 CORR_FWD_BYTES * B // 2))
    toph = wl.top_fw.half()
    rows.append(("corr_bwd, fp16 on the f16 matrix cores (extension)",
                 timeit(lambda: ops.correlation_fp16_grad(toph, fah, fbh)), CORR_BWD_BYTES * B // 2))
    rows.append(("corr_fwd, fp16 storage / f32 MFMA (exact form)", timeit(lambda: ops.correlation_fp16_exact(fah, fbh)),
                 CORR_FWD_BYTES * B // 2))
    rows.append(("corr_bwd, fp16 storage / f32 MFMA (exact form)",
                 timeit(lambda: ops.correlation_fp16_grad(toph, fah, fbh, exact=True)), CORR_BWD_BYTES * B // 2))
    del toph, fah, fbh
    rows.append(("image_warp C3", timeit(lambda: image_warp(wl.im2, ffw)), npx * 32))
    rows.append(("image_warp C2", timeit(lambda: image_warp(fbw, ffw)), npx * 24))
    rows.append(("image_warp_grad flow C3", timeit(lambda: image_warp_grad(g3, wl.im2, ffw, False, True)), npx * 40))
    rows.append(("image_warp_grad im C3", timeit(lambda: image_warp_grad(g3, wl.im2, ffw, True, False)), npx * 32))
    g2 = torch.randn_like(ffw)
    rows.append(("image_warp_grad im C2", timeit(lambda: image_warp_grad(g2, fbw, ffw, True, False)), npx * 24))
    rows.append(("backward_warp C3", timeit(lambda: ops.backward_warp(wl.im2, ffw)), npx * 32))
    rows.append(("backward_warp_grad C3", timeit(lambda: ops.backward_warp_grad(g3, wl.im2, ffw)), npx * 40))
    rows.append(("forward_warp", timeit(lambda: ops.forward_warp(ffw)), npx * 12))
    # fp16 storage of the warps (extension, BASELINE config 5): half the bytes per pixel
    im2h, ffwh, g3h = wl.im2.half(), ffw.half(), g3.half()
    rows.append(("image_warp C3, fp16 storage (extension)", timeit(lambda: ops.image_warp_fp16(im2h, ffwh)), npx * 16))
    f16 = ffwh.clone().requires_grad_(True)
    o16 = ops.image_warp_fp16(im2h, f16)
    rows.append(("image_warp_grad flow C3, fp16 storage (extension)",
                 timeit(lambda: torch.autograd.grad(o16, f16, g3h, retain_graph=True)), npx * 20))
    rows.append(("backward_warp C3, fp16 storage (extension)", timeit(lambda: ops.backward_warp_fp16(im2h, ffwh)), npx * 16))
    rows.append(("forward_warp, fp16 flows (extension)", timeit(lambda: ops.forward_warp_fp16(ffwh)), npx * 8))
    del im2h, ffwh, g3h, f16, o16
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
    rows.append(("loss leg of the step (compute_losses ternary+sym fwd+bwd)", timeit(wl.loss_leg), 0))
    k = kitti_config_step(wl)
    rows.append(("pyramid loss, 5 levels H/4..H/64, kitti weights, fwd+bwd", k["eager_ms"], 0))
    rows.append(("  the same as one HIP graph (unflow_amd.graph.GraphedStep)", k["graph_ms"], 0))
    from unflow_amd import flownet
    g = torch.Generator(device=ffw.device)
    g.manual_seed(7)
    flr = torch.randn((B, H // 4, W // 4, 2), device=ffw.device, generator=g) * 0.3
    rows.append(("stacked_inputs (resize+warp+abs+concat, 14 ch)", timeit(lambda: flownet.stacked_inputs(wl.im1, wl.im2, flr)), npx * 80))
    rows.append(("fb_occlusion", timeit(lambda: losses.fb_occlusion(ffw, fbw)), npx * 28))
    rows.append(("downsample x2 C3", timeit(lambda: ops.downsample(wl.im1, 2)), npx * 15))
    a = torch.empty(256 * 1024 * 1024, device=ffw.device)
    b = torch.empty_like(a)
    rows.append(("d2d torch copy_ 1 GiB", timeit(lambda: b.copy_(a)), 2 * a.numel() * 4))
    del a, b
    gbs = stream_copy_gbs(torch, ffw.device)
    rows.append(("unflow_stream_copy 1 GiB (measured HBM roofline)", 2 * 2 ** 30 / gbs / 1e6, 2 * 2 ** 30))
    for name, ms, nbytes in rows:
        print("%-42s %9.3f ms  %8.1f GB/s algorithmic" % (name, ms, nbytes / ms / 1e6), file=sys.stderr)


if __name__ == "__main__":
    main()
