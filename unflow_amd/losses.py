"""``src/e2eflow/core/losses.py`` on the HIP kernels.

Same function names and signatures as the reference: ``ternary_loss``
(losses.py:90), ``charbonnier_loss`` (:298), ``create_mask`` (:325),
``create_border_mask`` (:338), ``create_outgoing_mask`` (:347), ``length_sq``
(:12), ``photometric_loss`` (:198), ``gradient_loss`` (:244), ``smoothness_loss``
(:250), ``second_order_loss`` (:290), their ``_*_deltas`` helpers, and
``compute_losses`` (:16-87) with all eight entries of its dict, and ``occlusion``
(:125).  ``divergence`` (:147), ``norm`` (:165) and ``diffusion_loss`` (:173) have
no caller anywhere in the reference and are not mirrored.
"""
import ctypes
import math
import threading

import torch

from . import _lib
from .image_warp import image_warp, image_warp_grad
from .ops import _check_flow, _dev, _on, _ptr, _require_rank4, _stream, forward_warp

DISOCC_THRESH = 0.8  # losses.py:9


def _workspace(B, H, W, device):
    """Per-workgroup partial sums of one loss call (a few thousand floats)."""
    n = _lib.load().unflow_reduce_workspace_floats(int(B), int(H), int(W))
    return torch.empty(int(n), device=device, dtype=torch.float32)


# --------------------------------------------------------------------------- deferred reductions
class DeferredReductions:
    """``with DeferredReductions(device) as d:`` -- every loss forward inside leaves its per-workgroup partial
    sums in its workspace and returns ``d.results[slot]`` (not yet written) instead of launching its own one-block
    final reduction; ``d.fold(...)`` then reduces all of them in ONE launch and folds the weighted sum in a second.
    The loss pyramid of the reference (core/unsupervised.py:114-141) evaluates ~40 scalars per step: ~40 launches
    become 2.  The C side is ``unflow_reduce_plan_*`` (include/unflow_hip.h); the plan is per thread."""

    _tls = threading.local()        # the C plan is per thread (thread_local in census.hip): so is the open context

    @classmethod
    def current(cls):
        return getattr(cls._tls, "active", None)

    def __init__(self, device):
        self.plan = _lib.ReducePlan()
        self.results = torch.empty(_lib.PLAN_CAPACITY, device=device, dtype=torch.float32)
        self.keep = []              # the workspaces must live until fold()

    def __enter__(self):
        if DeferredReductions.current() is not None:
            raise RuntimeError("DeferredReductions does not nest")
        _lib.check(_lib.load().unflow_reduce_plan_begin(ctypes.byref(self.plan)))
        DeferredReductions._tls.active = self
        return self

    def __exit__(self, *exc):
        DeferredReductions._tls.active = None
        _lib.load().unflow_reduce_plan_end()
        return False

    def last(self, ws, back=0):
        """The scalar of the plan's last entry (``back`` = 1: the one before it), keeping its workspace alive."""
        self.keep.append(ws)
        return self.results[self.plan.n - 1 - back]

    def fold(self, coeff, term, term_weight, total, per_term):
        """results[0..n) <- the deferred reductions; total <- sum coeff[i] results[i]; per_term[t] <- sum over the
        segments of term t of term_weight[i] results[i] (host sequences of plan.n entries)."""
        n = self.plan.n
        c = (ctypes.c_float * max(n, 1))(*coeff)
        t = (ctypes.c_int * max(n, 1))(*term)
        w = (ctypes.c_float * max(n, 1))(*term_weight)
        with _on(self.results.device):
            _lib.check(_lib.load().unflow_reduce_plan_run(ctypes.byref(self.plan), c, t, w, _ptr(self.results), _ptr(total),
                                                          _ptr(per_term), _stream()))


def _loss_out(device):
    """The scalar a loss forward writes -- None (NULL: not written) while reductions are deferred."""
    return None if DeferredReductions.current() is not None else torch.empty((), device=device, dtype=torch.float32)


def _loss_result(loss, ws):
    d = DeferredReductions.current()
    return loss if d is None else d.last(ws)


class _FanOut(torch.autograd.Function):
    """n aliases of one tensor, one per consumer: the backward pass then sums the consumers' gradients in ONE launch
    (``unflow_sum_tensors``) where the autograd engine would issue n - 1 adds."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)      # an alias nobody used has no gradient: no zero-fill launch for it
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        live = [g.contiguous() for g in grads if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        out = torch.empty_like(live[0])
        ptrs = (ctypes.c_void_p * len(live))(*[g.data_ptr() for g in live])
        with _on(out.device):
            _lib.check(_lib.load().unflow_sum_tensors(ptrs, len(live), _ptr(out), out.numel(), _stream()))
        return out, None


def _fan_out(x, n):
    if torch.is_grad_enabled() and x.requires_grad and x.is_cuda and x.dtype == torch.float32:
        return _FanOut.apply(x, n)
    return (x,) * n


def length_sq(x):
    return torch.sum(torch.square(x), 3, keepdim=True)


# --------------------------------------------------------------------------- masks
def create_mask(tensor, paddings):
    """Ones inside, zeros on a border of paddings[0] rows / paddings[1] columns
    (losses.py:325-335).  No gradient (tf.stop_gradient)."""
    B, H, W = tensor.shape[0], tensor.shape[1], tensor.shape[2]
    (pt, pb), (pl, pr) = paddings
    mask = torch.empty((B, H, W, 1), device=tensor.device, dtype=torch.float32)
    if not mask.is_cuda:
        raise RuntimeError("create_mask: tensor must live on the GPU")
    with _on(mask.device):
        _lib.check(_lib.load().unflow_create_mask(_ptr(mask), B, H, W, pt, pb, pl, pr, _stream()))
    return mask


def border_size(height, width, border_ratio=0.1):
    """sz of create_border_mask (losses.py:341-342), computed in float32 like TF."""
    import numpy as np
    return int(np.ceil(np.float32(min(height, width)) * np.float32(border_ratio)))


def create_border_mask(tensor, border_ratio=0.1):
    sz = border_size(tensor.shape[1], tensor.shape[2], border_ratio)
    return create_mask(tensor, [[sz, sz], [sz, sz]])


def create_outgoing_mask(flow):
    """Zero where the flow carries a pixel over the image boundary (losses.py:347-366)."""
    flow = _dev(flow.detach(), "flow")
    _check_flow(None, flow)
    B, H, W, _ = flow.shape
    mask = torch.empty((B, H, W, 1), device=flow.device, dtype=torch.float32)
    with _on(flow.device):
        _lib.check(_lib.load().unflow_outgoing_mask(_ptr(flow), _ptr(mask), B, H, W, _stream()))
    return mask


def fb_occlusion(flow, flow_other_warped, mask_inout=None, mask_in=None):
    """(flow_diff, occ) of losses.py:38-51; multiplies mask_inout by (1-occ) in place
    when given (losses.py:53-55).  With ``mask_in`` the product ``mask_in * (1 - occ)`` is written
    to ``mask_inout`` instead (no copy of a shared border mask).  occ and the mask carry no gradient;
    flow_diff is returned detached (use ``flow + flow_other_warped`` for a differentiable one)."""
    flow = _dev(flow.detach(), "flow")
    other = _dev(flow_other_warped.detach(), "flow_other_warped")
    _check_flow(None, flow)
    B, H, W, _ = flow.shape
    diff = torch.empty_like(flow)
    occ = torch.empty((B, H, W, 1), device=flow.device, dtype=torch.float32)
    with _on(flow.device):
        if mask_in is not None:
            if mask_inout is None:
                raise ValueError("fb_occlusion: mask_in needs mask_inout to write the product to")
            _lib.check(_lib.load().unflow_fb_occlusion_masked(_ptr(flow), _ptr(other), _ptr(diff), _ptr(occ),
                                                              _ptr(_dev(mask_in, "mask_in")), _ptr(mask_inout), B, H, W, _stream()))
        else:
            _lib.check(_lib.load().unflow_fb_occlusion(_ptr(flow), _ptr(other), _ptr(diff), _ptr(occ),
                                                       _ptr(mask_inout), B, H, W, _stream()))
    return diff, occ


def fb_occlusion_losses(flow, flow_other_warped, mask_in, mask_out, norm_fb=0, norm_occ=0, warp_other=False):
    """``fb_occlusion`` with ``mask_out = mask_in * (1 - occ)`` and, from the same pass, the two scalars that need
    nothing else (losses.py:57-58, 62-67): ``charbonnier_loss(flow_diff, mask_out)`` and ``charbonnier_loss(1 - mask_out)``
    (extension: one launch instead of three).  Returns (flow_diff, occ, loss_fb, loss_occ), all detached -- wrap
    ``loss_fb`` with ``_CharbonnierGiven`` for its gradient.  ``warp_other``: the second argument is the other
    direction's flow itself and ``image_warp(flow_other, flow)`` is taken on the fly (one launch instead of four;
    ``_FlowDiffWarp`` carries the warp's gradient)."""
    flow = _dev(flow.detach(), "flow")
    other = _dev(flow_other_warped.detach(), "flow_other_warped")
    _check_flow(None, flow)
    B, H, W, _ = flow.shape
    diff = torch.empty_like(flow)
    occ = torch.empty((B, H, W, 1), device=flow.device, dtype=torch.float32)
    l_fb, l_occ = _loss_out(flow.device), _loss_out(flow.device)
    ws_fb, ws_occ = _workspace(B, H, W, flow.device), _workspace(B, H, W, flow.device)
    with _on(flow.device):
        fn = _lib.load().unflow_fb_occlusion_warp_losses if warp_other else _lib.load().unflow_fb_occlusion_losses
        _lib.check(fn(
            _ptr(flow), _ptr(other), _ptr(diff), _ptr(occ), _ptr(_dev(mask_in, "mask_in")), _ptr(mask_out),
            _ptr(l_fb), _ptr(ws_fb), int(norm_fb), _ptr(l_occ), _ptr(ws_occ), int(norm_occ), B, H, W, _stream()))
    d = DeferredReductions.current()
    if d is not None:          # the plan's last two entries: fb, occ
        l_fb, l_occ = d.last(ws_fb, back=1), d.last(ws_occ)
    return diff, occ, l_fb, l_occ


class _FlowDiffWarp(torch.autograd.Function):
    """``flow + image_warp(flow_other, coords)`` (losses.py:38-41) whose value the occlusion kernel has computed on the
    fly: the gradient goes to ``flow`` unchanged and through ``image_warp``'s backward to ``flow_other`` and ``coords``."""

    @staticmethod
    def forward(ctx, flow, flow_other, coords, diff):
        ctx.save_for_backward(flow_other, coords)
        return diff.view_as(diff)

    @staticmethod
    def backward(ctx, g):
        flow_other, coords = ctx.saved_tensors
        d_other, d_coords = image_warp_grad(g, flow_other, coords, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return g, d_other, d_coords, None


class _CharbonnierGiven(torch.autograd.Function):
    """``charbonnier_loss(x, mask)`` (default alpha, beta, epsilon) whose VALUE another kernel has already computed
    (``fb_occlusion_losses``); the gradient is the ordinary one."""

    @staticmethod
    def forward(ctx, x, mask, value, norm_count):
        ctx.save_for_backward(x, mask)
        ctx.norm_count = norm_count
        return value.view_as(value)          # (a view: the slot of a deferred reduction is found by its storage offset)

    @staticmethod
    def backward(ctx, gout):
        x, mask = ctx.saved_tensors
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        with _on(x.device):
            _lib.check(_lib.load().unflow_charbonnier_bwd(
                _ptr(x), _ptr(mask), mask.shape[3], _ptr(gout.contiguous()), _ptr(dx), B, H, W, C, 0.45, 1.0, 0.001,
                0.0, ctx.norm_count, _stream()))
        return dx, None, None, None


def fb_occlusion_warp(flow, flow_other, mask_inout=None, want_warped=False, unwarped_magnitude=False,
                      mask_in=None):
    """``fb_occlusion(flow, image_warp(flow_other, flow), mask_inout)`` in one pass: the warped
    flow is computed on the fly and only written when ``want_warped``.  Returns (warped or
    None, occ); no gradient (the occlusion map is a thresholded quantity).  With ``mask_in``
    the product ``mask_in * (1 - occ)`` is written to ``mask_inout`` instead of in place."""
    flow = _dev(flow.detach(), "flow")
    other = _dev(flow_other.detach(), "flow_other")
    _check_flow(None, flow)
    if other.shape != flow.shape:
        raise ValueError("flow %s and flow_other %s must have the same shape" % (tuple(flow.shape), tuple(other.shape)))
    B, H, W, _ = flow.shape
    warped = torch.empty_like(flow) if want_warped else None
    occ = torch.empty((B, H, W, 1), device=flow.device, dtype=torch.float32)
    with _on(flow.device):
        _lib.check(_lib.load().unflow_fb_occlusion_warp(
            _ptr(flow), _ptr(other), _ptr(warped), _ptr(occ),
            _ptr(mask_in if mask_in is not None else mask_inout), _ptr(mask_inout), B, H, W,
            int(bool(unwarped_magnitude)), _stream()))
    return warped, occ


def occlusion(flow_fw, flow_bw):
    """(occ_fw, occ_bw) of ``occlusion`` (losses.py:125-134), the evaluation-time variant
    (core/train.py:350) whose threshold uses the magnitudes of the UNWARPED flows."""
    _, occ_fw = fb_occlusion_warp(flow_fw, flow_bw, unwarped_magnitude=True)
    _, occ_bw = fb_occlusion_warp(flow_bw, flow_fw, unwarped_magnitude=True)
    return occ_fw, occ_bw


def sym_loss(mask, density_other, norm_count=0, thresh=DISOCC_THRESH):
    """One direction of ``losses['sym']`` (losses.py:60-61):
    ``charbonnier_loss((1 - mask) - cast(density_other < thresh))`` in one pass."""
    mask, density_other = _dev(mask.detach(), "mask"), _dev(density_other.detach(), "density_other")
    _require_rank4(mask, "mask")
    B, H, W, C = mask.shape
    if C != 1 or density_other.shape != mask.shape:
        raise ValueError("sym_loss expects two [B,H,W,1] maps")
    loss = _loss_out(mask.device)
    ws = _workspace(B, H, W, mask.device)
    with _on(mask.device):
        _lib.check(_lib.load().unflow_sym_loss_fwd(_ptr(mask), _ptr(density_other), float(thresh), _ptr(loss),
                                                   _ptr(ws), B, H, W, norm_count, _stream()))
    return _loss_result(loss, ws)


def occ_loss(mask, norm_count=0):
    """One direction of ``losses['occ']`` (losses.py:62-63): ``charbonnier_loss(1 - mask)`` in one
    pass (the sym kernel with a comparison that is never true); no gradient, as in the reference
    (the masks are stop-gradient'd)."""
    return sym_loss(mask, mask, norm_count=norm_count, thresh=-1.0)


class _FlowDiff(torch.autograd.Function):
    """``flow + flow_other_warped`` (losses.py:40-41) taken from the occlusion kernel, which computes it
    anyway; the gradient passes to both summands unchanged."""

    @staticmethod
    def forward(ctx, flow, flow_other_warped, diff):
        return diff.view_as(diff)

    @staticmethod
    def backward(ctx, g):
        return g, g, None


# --------------------------------------------------------------------------- charbonnier
class _Charbonnier(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, truncate, alpha, beta, epsilon, norm_count):
        x = _dev(x, "x")
        _require_rank4(x, "x")
        B, H, W, C = x.shape
        mc = 0
        if mask is not None:
            mask = _dev(mask, "mask")
            mc = mask.shape[3]
            if mask.shape[:3] != x.shape[:3] or mc not in (1, C):
                raise ValueError("mask channels must be 1 or match x")
        loss = _loss_out(x.device)
        ws = _workspace(B, H, W, x.device)
        with _on(x.device):
            _lib.check(_lib.load().unflow_charbonnier_fwd(
                _ptr(x), _ptr(mask), mc, _ptr(loss), _ptr(ws), B, H, W, C, alpha, beta, epsilon,
                truncate or 0.0, norm_count, _stream()))
        ctx.save_for_backward(x, mask)
        ctx.cfg = (mc, truncate or 0.0, alpha, beta, epsilon, norm_count)
        return _loss_result(loss, ws)

    @staticmethod
    def backward(ctx, gout):
        x, mask = ctx.saved_tensors
        mc, truncate, alpha, beta, epsilon, norm_count = ctx.cfg
        B, H, W, C = x.shape
        gout = gout.contiguous()
        dx = torch.empty_like(x)
        with _on(x.device):
            _lib.check(_lib.load().unflow_charbonnier_bwd(
                _ptr(x), _ptr(mask), mc, _ptr(gout), _ptr(dx), B, H, W, C, alpha, beta, epsilon,
                truncate, norm_count, _stream()))
        return dx, None, None, None, None, None, None


def charbonnier_loss(x, mask=None, truncate=None, alpha=0.45, beta=1.0, epsilon=0.001,
                     norm_count=0):
    """Generalized charbonnier loss of the difference tensor x (losses.py:298-322).
    ``norm_count`` (extension): the normalisation; 0 = batch*height*width*channels
    of x, a batch shard passes the global element count."""
    return _Charbonnier.apply(x, mask, truncate, alpha, beta, epsilon, norm_count)


def photometric_loss(im_diff, mask):
    return charbonnier_loss(im_diff, mask, beta=255)  # losses.py:198-199


# --------------------------------------------------------------------------- 3x3 stencil terms
def _deltas(flow, order):
    flow = _dev(flow, "flow")
    _require_rank4(flow, "flow")
    _check_flow(None, flow)
    B, H, W, _ = flow.shape
    K = 2 if order == 1 else 4
    du, dv, m = (torch.empty((B, H, W, K), device=flow.device, dtype=torch.float32) for _ in range(3))
    with _on(flow.device):
        _lib.check(_lib.load().unflow_smoothness_deltas(_ptr(flow), _ptr(du), _ptr(dv), _ptr(m), B, H, W,
                                                        order, _stream()))
    return du, dv, m


def _smoothness_deltas(flow):
    """(delta_u, delta_v, mask), each [B,H,W,2] (losses.py:206-225).  Values only (test hook);
    the differentiable path is ``smoothness_loss``."""
    return _deltas(flow, 1)


def _second_order_deltas(flow):
    """(delta_u, delta_v, mask), each [B,H,W,4] (losses.py:258-287)."""
    return _deltas(flow, 2)


class _SmoothnessLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow, order, norm_count):
        flow = _dev(flow, "flow")
        _require_rank4(flow, "flow")
        _check_flow(None, flow)
        B, H, W, _ = flow.shape
        loss = _loss_out(flow.device)
        ws = _workspace(B, H, W, flow.device)
        with _on(flow.device):
            _lib.check(_lib.load().unflow_smoothness_loss_fwd(_ptr(flow), _ptr(loss), _ptr(ws), B, H, W,
                                                              order, norm_count, _stream()))
        ctx.save_for_backward(flow)
        ctx.cfg = (order, norm_count)
        return _loss_result(loss, ws)

    @staticmethod
    def backward(ctx, gout):
        flow, = ctx.saved_tensors
        order, norm_count = ctx.cfg
        B, H, W, _ = flow.shape
        gout = gout.contiguous()
        d = torch.empty_like(flow)
        with _on(flow.device):
            _lib.check(_lib.load().unflow_smoothness_loss_bwd(_ptr(flow), _ptr(gout), _ptr(d), B, H, W,
                                                              order, norm_count, _stream()))
        return d, None, None


def smoothness_loss(flow, norm_count=0):
    """First-order smoothness (losses.py:250-255): loss_u + loss_v of the masked
    forward differences.  ``norm_count`` (extension): 0 = B*H*W*2."""
    return _SmoothnessLoss.apply(flow, 1, norm_count)


def second_order_loss(flow, norm_count=0):
    """Second-order smoothness (losses.py:290-295).  ``norm_count``: 0 = B*H*W*4."""
    return _SmoothnessLoss.apply(flow, 2, norm_count)


def _gradient_delta(im1, im2_warped):
    """conv2d(im1, sobel) - conv2d(im2_warped, sobel), [B,H,W,6] (losses.py:228-241).  Values only."""
    im1, im2_warped = _dev(im1, "im1"), _dev(im2_warped, "im2_warped")
    _require_rank4(im1, "im1")
    B, H, W, C = im1.shape
    if C != 3 or im2_warped.shape != im1.shape:
        raise ValueError("_gradient_delta expects two [B,H,W,3] images")
    diff = torch.empty((B, H, W, 6), device=im1.device, dtype=torch.float32)
    with _on(im1.device):
        _lib.check(_lib.load().unflow_gradient_delta(_ptr(im1), _ptr(im2_warped), _ptr(diff), B, H, W,
                                                     _stream()))
    return diff


class _GradientLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im1, im2_warped, mask, norm_count):
        im1, im2_warped, mask = _dev(im1, "im1"), _dev(im2_warped, "im2_warped"), _dev(mask, "mask")
        _require_rank4(im1, "im1")
        B, H, W, C = im1.shape
        if C != 3 or im2_warped.shape != im1.shape:
            raise ValueError("gradient_loss expects two [B,H,W,3] images, got %s and %s"
                             % (tuple(im1.shape), tuple(im2_warped.shape)))
        if tuple(mask.shape) != (B, H, W, 1):
            raise ValueError("mask must have shape %s, got %s" % ((B, H, W, 1), tuple(mask.shape)))
        loss = _loss_out(im1.device)
        ws = _workspace(B, H, W, im1.device)
        with _on(im1.device):
            _lib.check(_lib.load().unflow_gradient_loss_fwd(_ptr(im1), _ptr(im2_warped), _ptr(mask), _ptr(loss),
                                                            _ptr(ws), B, H, W, norm_count, _stream()))
        ctx.save_for_backward(im1, im2_warped, mask)
        ctx.norm_count = norm_count
        return _loss_result(loss, ws)

    @staticmethod
    def backward(ctx, gout):
        im1, im2_warped, mask = ctx.saved_tensors
        B, H, W, _ = im1.shape
        gout = gout.contiguous()
        d = torch.empty_like(im2_warped)
        with _on(im1.device):
            _lib.check(_lib.load().unflow_gradient_loss_bwd(_ptr(im1), _ptr(im2_warped), _ptr(mask), _ptr(gout),
                                                            _ptr(d), B, H, W, ctx.norm_count, _stream()))
        return None, d, None, None


def gradient_loss(im1, im2_warped, mask, norm_count=0):
    """Charbonnier of the masked Sobel-gradient difference (losses.py:244-247).
    Gradient flows into ``im2_warped`` only.  ``norm_count``: 0 = B*H*W*6."""
    return _GradientLoss.apply(im1, im2_warped, mask, norm_count)


# --------------------------------------------------------------------------- census
class _TernaryLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im1, im2_warped, mask, max_distance, norm_count):
        im1, im2_warped = _dev(im1, "im1"), _dev(im2_warped, "im2_warped")
        mask = _dev(mask, "mask")
        _require_rank4(im1, "im1")
        B, H, W, C = im1.shape
        if C != 3 or im2_warped.shape != im1.shape:
            raise ValueError("ternary_loss expects two [B,H,W,3] images (rgb_to_grayscale), got %s and %s"
                             % (tuple(im1.shape), tuple(im2_warped.shape)))
        if tuple(mask.shape) != (B, H, W, 1):
            raise ValueError("mask must have shape %s, got %s" % ((B, H, W, 1), tuple(mask.shape)))
        loss = _loss_out(im1.device)
        dldd = torch.empty((B, H, W), device=im1.device, dtype=torch.float32)
        ws = _workspace(B, H, W, im1.device)
        with _on(im1.device):
            _lib.check(_lib.load().unflow_ternary_loss_fwd(
                _ptr(im1), _ptr(im2_warped), _ptr(mask), _ptr(loss), _ptr(dldd), _ptr(None), _ptr(ws),
                B, H, W, max_distance, norm_count, _stream()))
        ctx.save_for_backward(im1, im2_warped, dldd)
        ctx.max_distance = max_distance
        return _loss_result(loss, ws)

    @staticmethod
    def backward(ctx, gout):
        im1, im2_warped, dldd = ctx.saved_tensors
        B, H, W, _ = im1.shape
        gout = gout.contiguous()
        d = torch.empty_like(im2_warped)
        with _on(im1.device):
            _lib.check(_lib.load().unflow_ternary_loss_bwd(
                _ptr(im1), _ptr(im2_warped), _ptr(dldd), _ptr(gout), _ptr(d), B, H, W,
                ctx.max_distance, _stream()))
        # im1 is stop-gradient'd upstream in the reference (core/augment.py:53,108)
        return None, d, None, None, None


class _TernaryWarped(torch.autograd.Function):
    """ternary_loss(im1, image_warp(im2, flow), mask): warp fused into the census forward,
    backward fused down to the flow; neither the warped image nor its gradient exists."""

    @staticmethod
    def forward(ctx, im1, im2, flow, mask, max_distance, norm_count):
        im1, im2, mask = _dev(im1, "im1"), _dev(im2, "im2"), _dev(mask, "mask")
        flow_d = _dev(flow.detach(), "flow")
        _check_flow(im1, flow_d)
        B, H, W, C = im1.shape
        if C != 3 or im2.shape != im1.shape or tuple(mask.shape) != (B, H, W, 1):
            raise ValueError("ternary_loss_warped expects [B,H,W,3] images and a [B,H,W,1] mask")
        loss = _loss_out(im1.device)
        dldd, gray1, gray2w = (torch.empty((B, H, W), device=im1.device, dtype=torch.float32) for _ in range(3))
        ws = _workspace(B, H, W, im1.device)
        with _on(im1.device):
            _lib.check(_lib.load().unflow_ternary_loss_warp_fwd(
                _ptr(im1), _ptr(im2), _ptr(flow_d), _ptr(mask), _ptr(loss), _ptr(dldd), _ptr(gray1), _ptr(gray2w),
                _ptr(ws), B, H, W, max_distance, norm_count, _stream()))
        ctx.save_for_backward(im2, flow_d, gray1, gray2w, dldd)
        ctx.max_distance = max_distance
        return _loss_result(loss, ws)

    @staticmethod
    def backward(ctx, gout):
        im2, flow, gray1, gray2w, dldd = ctx.saved_tensors
        B, H, W, _ = im2.shape
        gout = gout.contiguous()
        d_flow = torch.empty_like(flow)
        with _on(im2.device):
            _lib.check(_lib.load().unflow_ternary_loss_warp_bwd(
                _ptr(gray1), _ptr(gray2w), _ptr(dldd), _ptr(gout), _ptr(im2), _ptr(flow), _ptr(d_flow),
                B, H, W, ctx.max_distance, _stream()))
        return None, None, d_flow, None, None, None


def ternary_loss_warped(im1, im2, flow, mask, max_distance=1, norm_count=0):
    """``ternary_loss(im1, image_warp(im2, flow), mask, max_distance)`` (losses.py:22,82-83) whose
    forward evaluates the warp while staging the census tiles and whose backward goes straight
    to ``flow`` (max_distance 2 and 3): two kernels instead of four, and neither the warped
    image nor its [B,H,W,3] gradient touches HBM.  Other radii compose the two ops, and so
    does an ``im2`` that requires a gradient: the fused backward only differentiates down to
    ``flow``, the composed one also reaches ``im2`` through ``image_warp`` (``im1`` gets none
    on either path: ``ternary_loss`` treats it as a constant, core/augment.py:53,108)."""
    fused = max_distance in (2, 3) and im1.numel() // 3 < (1 << 29)
    if fused and not (torch.is_grad_enabled() and im2.requires_grad):
        return _TernaryWarped.apply(im1, im2, flow, mask, max_distance, norm_count)
    return ternary_loss(im1, image_warp(im2, flow), mask, max_distance, norm_count)


def ternary_loss(im1, im2_warped, mask, max_distance=1, norm_count=0):
    """Soft ternary census + "Hamming" distance + charbonnier (losses.py:90-122),
    one fused kernel.  Gradient flows into ``im2_warped`` only."""
    return _TernaryLoss.apply(im1, im2_warped, mask, max_distance, norm_count)


def ternary_dist(im1, im2_warped, max_distance=1):
    """Per-pixel distance map ``dist_sum`` [B,H,W,1] of losses.py:110-114 (test hook)."""
    im1, im2_warped = _dev(im1, "im1"), _dev(im2_warped, "im2_warped")
    B, H, W, _ = im1.shape
    mask = torch.ones((B, H, W, 1), device=im1.device, dtype=torch.float32)
    loss = torch.empty((), device=im1.device, dtype=torch.float32)
    dist = torch.empty((B, H, W, 1), device=im1.device, dtype=torch.float32)
    ws = _workspace(B, H, W, im1.device)
    with _on(im1.device):
        _lib.check(_lib.load().unflow_ternary_loss_fwd(
            _ptr(im1), _ptr(im2_warped), _ptr(mask), _ptr(loss), _ptr(None), _ptr(dist), _ptr(ws), B,
            H, W, max_distance, 0, _stream()))
    return dist


# --------------------------------------------------------------------------- integer census (extension)
def census_transform(im, max_distance=3, eps=0.9):
    """Integer ternary census signature [B,H,W,2] (int64 bit patterns) of ``im`` [B,H,W,3].
    Not part of the reference (whose census is the soft ``ternary_loss``): neighbour k is
    'brighter' / 'darker' if 255*gray differs from the centre by more than ``eps``."""
    im = _dev(im, "im")
    _require_rank4(im, "im")
    B, H, W, C = im.shape
    if C != 3:
        raise ValueError("census_transform expects a [B,H,W,3] image, got %s" % (tuple(im.shape),))
    sig = torch.empty((B, H, W, 2), device=im.device, dtype=torch.int64)
    with _on(im.device):
        _lib.check(_lib.load().unflow_census_transform(_ptr(im), ctypes.c_void_p(sig.data_ptr()), B, H, W,
                                                       max_distance, eps, _stream()))
    return sig


def census_hamming(sig1, sig2, mask=None):
    """(dist [B,H,W] int32, total int64 scalar tensor): per-pixel number of differing ternary
    digits (64-bit popcounts) and its sum over the pixels where ``mask`` is non-zero."""
    if sig1.shape != sig2.shape or sig1.dtype != torch.int64 or sig2.dtype != torch.int64 or not sig1.is_cuda:
        raise ValueError("census_hamming expects two int64 [B,H,W,2] signatures on the GPU")
    B, H, W, _ = sig1.shape
    sig1, sig2 = sig1.contiguous(), sig2.contiguous()
    dist = torch.empty((B, H, W), device=sig1.device, dtype=torch.int32)
    total = torch.empty((), device=sig1.device, dtype=torch.int64)
    if mask is not None:
        mask = _dev(mask, "mask")
    with _on(sig1.device):
        _lib.check(_lib.load().unflow_census_hamming(
            ctypes.c_void_p(sig1.data_ptr()), ctypes.c_void_p(sig2.data_ptr()), _ptr(mask),
            ctypes.c_void_p(dist.data_ptr()), ctypes.c_void_p(total.data_ptr()), B, H, W, _stream()))
    return dist, total


# --------------------------------------------------------------------------- composite
def compute_losses(im1, im2, flow_fw, flow_bw, border_mask=None, mask_occlusion='',
                   data_max_distance=1, norm_batch=0, terms=None, return_aux=False):
    """``compute_losses`` (losses.py:16-87): 'sym', 'occ', 'photo', 'grad',
    'smooth_1st', 'smooth_2nd', 'fb' and 'ternary'.  ``terms`` (extension) restricts the dict to the named entries --
    TF1 only executes the graph nodes a fetched loss needs, so a config with just
    ``ternary_weight`` never runs the others; None computes all of them.  The
    disocclusion maps (two forward_warp calls) are computed when 'sym' is asked
    for, when ``mask_occlusion == 'disocc'`` or when ``return_aux`` is set.
    ``norm_batch`` (extension) is the GLOBAL batch size when this call sees a
    shard of the batch; 0 = the local batch.  ``return_aux`` also returns the intermediate
    tensors (warped images, masks, occlusion maps)."""
    parts, aux = compute_losses_parts(im1, im2, flow_fw, flow_bw, border_mask, mask_occlusion, data_max_distance,
                                      norm_batch, terms, return_aux)
    losses = {k: v[0] + v[1] for k, v in parts.items()}      # losses.py: each entry is forward + backward direction
    return (losses, aux) if return_aux else losses


def compute_losses_parts(im1, im2, flow_fw, flow_bw, border_mask=None, mask_occlusion='',
                         data_max_distance=1, norm_batch=0, terms=None, return_aux=False):
    """``compute_losses`` with every entry left as its (forward, backward) pair of scalars: the pyramid
    driver folds all of them into its weighted sum with ONE fused op instead of an add and a scale
    per term and level.  Returns (parts, aux)."""
    assert mask_occlusion in ['fb', 'disocc', '']
    B, H, W, _ = im1.shape
    npx = (norm_batch or B) * H * W
    # one alias of each flow per consumer (image warp, the two smoothness terms, flow and image of the flow warps,
    # the flow difference): their gradients are summed in one launch instead of five chained adds
    fw, bw = _fan_out(flow_fw, 6), _fan_out(flow_bw, 6)
    # losses.py:16-87 writes every step once per direction; the two directions only meet in the other flow's role
    # as the warped "image" of the forward-backward check (alias 4) and in the splat density of the other flow
    l_fw, a_fw = _direction_losses(im1, im2, fw, bw[4], flow_fw, flow_bw, border_mask, mask_occlusion,
                                   data_max_distance, npx, terms, return_aux)
    l_bw, a_bw = _direction_losses(im2, im1, bw, fw[4], flow_bw, flow_fw, border_mask, mask_occlusion,
                                   data_max_distance, npx, terms, return_aux)
    losses = {k: (l_fw[k], l_bw[k]) for k in l_fw}
    aux = dict(im2_warped=a_fw['warped'], im1_warped=a_bw['warped'], mask_fw=a_fw['mask'], mask_bw=a_bw['mask'],
               disocc_fw=a_bw['disocc_other'], disocc_bw=a_fw['disocc_other'], flow_bw_warped=a_fw['flow_other_warped'],
               flow_fw_warped=a_bw['flow_other_warped'], fb_occ_fw=a_fw['fb_occ'], fb_occ_bw=a_bw['fb_occ'])
    return losses, aux


def compute_losses_stacked(ims, ims_other, flows, border_mask=None, mask_occlusion='', data_max_distance=1,
                           norm_batch=0, terms=None, flows_other=None):
    """Both directions of ``compute_losses`` (losses.py:16-87) as ONE batch of 2B (extension).  The reference evaluates
    every step of the loss twice, once per direction, with the roles of the images and flows exchanged; nothing couples
    the items of a batch, so the backward direction is the forward direction of B more pairs:

        ims = [im1; im2], ims_other = [im2; im1], flows = [flow_fw; flow_bw]   (``[a; b]`` = concatenated along the batch)

    and the other direction's flow of item b is item (b + B) mod 2B of ``flows``.  Every entry of the result is the
    1-tuple (forward + backward sum), half the launches of ``compute_losses_parts`` -- what a loss pyramid's coarse
    levels, whose launches do not fill the chip, are bound by.  ``border_mask`` is [2B,H,W,1] (the mask twice);
    ``norm_batch`` the number of PAIRS the means are taken over (default B = half the stacked batch).  ``flows_other``:
    ``[flow_bw; flow_fw]`` if the caller has it (the pyramid driver makes both orders in one launch); else it is
    rolled from ``flows`` here."""
    assert mask_occlusion in ['fb', 'disocc', '']
    B2, H, W, _ = ims.shape
    if B2 % 2 or flows.shape[0] != B2 or ims_other.shape != ims.shape:
        raise ValueError("compute_losses_stacked expects [2B,H,W,C] images and [2B,H,W,2] flows")
    B = B2 // 2
    npx = (norm_batch or B) * H * W
    fa = _fan_out(flows, 6)
    want_fb = terms is None or 'fb' in terms
    # the other direction's flow: the halves exchanged.  With 'fb' it is the image of a differentiable warp (alias 4)
    if flows_other is not None:
        if flows_other.shape != flows.shape:
            raise ValueError("flows_other must have the shape of flows")
        other_img, other_raw = (flows_other if want_fb else None), flows_other.detach()
    else:
        other_img = torch.roll(fa[4], B, 0) if want_fb else None
        need_raw = (terms is None or 'sym' in terms) or mask_occlusion in ('disocc', 'fb')
        other_raw = other_img.detach() if other_img is not None else (torch.roll(flows.detach(), B, 0) if need_raw else None)
    losses, _ = _direction_losses(ims, ims_other, fa, other_img, flows, other_raw, border_mask, mask_occlusion,
                                  data_max_distance, npx, terms, False)
    return {k: (v,) for k, v in losses.items()}


def _direction_losses(im_a, im_b, fa, other_img, flow, flow_other, border_mask, mask_occlusion, data_max_distance,
                      npx, terms, return_aux):
    """One direction of losses.py:16-87: ``im_b`` is warped towards ``im_a`` by ``flow``; ``fa`` = the six aliases of
    ``flow`` (``_fan_out``), ``other_img`` the other direction's flow in its role as the warped image of the
    forward-backward check, ``flow_other`` the other direction's flow itself (density, occlusion test: no gradient)."""
    losses = {}
    want = (lambda k: True) if terms is None else (lambda k: k in terms)
    b_warped = None
    # the warped image as a graph node, unless the census term is its only consumer: then
    # ternary_loss_warped warps inside and differentiates straight down to the flow
    if want('photo') or want('grad') or return_aux:
        b_warped = image_warp(im_b, fa[0])
    # Intermediates are only produced when something asks for them (TF1 runs just the nodes a
    # fetched tensor needs): the splat density for 'sym' / 'disocc' / aux, the warped flow
    # for 'fb' / aux -- otherwise the occlusion test warps the other flow on the fly.
    dens_other = disocc_other = None
    if want('sym') or mask_occlusion == 'disocc' or return_aux:
        dens_other = forward_warp(flow_other.detach())
    if mask_occlusion == 'disocc' or return_aux:
        # tf.cast(forward_warp(flow) < DISOCC_THRESH): no gradient through the comparison
        disocc_other = (dens_other < DISOCC_THRESH).float()

    if border_mask is None:
        mask = create_outgoing_mask(flow)
    elif mask_occlusion == 'fb':       # the occlusion kernels write border_mask * (1 - occ) into a fresh buffer: no copy
        mask = torch.empty_like(border_mask)
    else:
        mask = border_mask.clone()

    flow_other_warped = fb_occ = diff = fb_value = occ_value = None
    fdiff = None
    if want('fb') and want('occ') and mask_occlusion == 'fb' and not return_aux:
        # the occlusion pass warps the other flow on the fly and has the flow difference and the final mask in
        # registers: warp, test, mask and both scalars are one launch
        diff, fb_occ, fb_value, occ_value = fb_occlusion_losses(
            flow, other_img, border_mask if border_mask is not None else mask, mask, norm_fb=npx * 2, norm_occ=npx,
            warp_other=True)
        fdiff = _FlowDiffWarp.apply(fa[5], other_img, fa[3], diff)
    elif want('fb') or return_aux:
        flow_other_warped = image_warp(other_img, fa[3])
        fb = mask_occlusion == 'fb'
        diff, fb_occ = fb_occlusion(flow, flow_other_warped, mask if fb else None, mask_in=border_mask if fb else None)
    elif mask_occlusion == 'fb':
        _, fb_occ = fb_occlusion_warp(flow, flow_other, mask, mask_in=border_mask)
    if mask_occlusion == 'disocc':
        mask = mask * (1 - disocc_other)

    if want('sym'):
        losses['sym'] = sym_loss(mask, dens_other, norm_count=npx)
    if want('occ'):      # charbonnier_loss(1 - mask), the complement taken inside the kernel
        losses['occ'] = occ_value if occ_value is not None else occ_loss(mask, norm_count=npx)
    if want('photo'):
        losses['photo'] = charbonnier_loss(im_a - b_warped, mask, beta=255, norm_count=npx * 3)
    if want('grad'):
        losses['grad'] = gradient_loss(im_a, b_warped, mask, norm_count=npx * 6)
    if want('smooth_1st'):
        losses['smooth_1st'] = smoothness_loss(fa[1], norm_count=npx * 2)
    if want('smooth_2nd'):
        losses['smooth_2nd'] = second_order_loss(fa[2], norm_count=npx * 4)
    if want('fb'):       # flow + warped other flow: the sum the occlusion kernel already wrote (same rounding)
        if fdiff is None:
            fdiff = _FlowDiff.apply(fa[5], flow_other_warped, diff)
        losses['fb'] = (_CharbonnierGiven.apply(fdiff, mask, fb_value, npx * 2) if fb_value is not None
                        else charbonnier_loss(fdiff, mask, norm_count=npx * 2))
    if want('ternary') and b_warped is None:
        losses['ternary'] = ternary_loss_warped(im_a, im_b, fa[0], mask, max_distance=data_max_distance, norm_count=npx)
    elif want('ternary'):
        losses['ternary'] = ternary_loss(im_a, b_warped, mask, max_distance=data_max_distance, norm_count=npx)
    aux = dict(warped=b_warped, mask=mask, disocc_other=disocc_other, flow_other_warped=flow_other_warped, fb_occ=fb_occ)
    return losses, aux
