"""The full-resolution warp block of the stacked networks -- mirror of the
``flow is not None`` branch of ``_flownet_s`` (``src/e2eflow/core/flownet.py:46-57``).
The networks themselves (``flownet_s`` / ``flownet_c``, :79-233) are outside this path.
"""
import torch

from . import _lib
from .image_warp import image_warp
from .ops import _dev, _on, _ptr, _require_rank4, _stream

FLOW_SCALE = 5.0  # flownet.py:11


class _ResizeBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, height, width, mult):
        x = _dev(x, "images")
        _require_rank4(x, "images")
        B, h, w, C = x.shape
        out = torch.empty((B, height, width, C), device=x.device, dtype=torch.float32)
        with _on(x.device):
            _lib.check(_lib.load().unflow_resize_bilinear_fwd(_ptr(x), _ptr(out), B, h, w, height, width, C,
                                                              mult, _stream()))
        ctx.cfg = (h, w, mult)
        return out

    @staticmethod
    def backward(ctx, gout):
        h, w, mult = ctx.cfg
        gout = gout.contiguous()
        B, H, W, C = gout.shape
        gin = torch.empty((B, h, w, C), device=gout.device, dtype=torch.float32)
        with _on(gout.device):
            _lib.check(_lib.load().unflow_resize_bilinear_bwd(_ptr(gout), _ptr(gin), B, h, w, H, W, C, mult,
                                                              _stream()))
        return gin, None, None, None


def resize_bilinear(images, size, mult=1.0):
    """``tf.image.resize_bilinear(images, size)`` (TF 1.x, align_corners=False), times ``mult``."""
    return _ResizeBilinear.apply(images, int(size[0]), int(size[1]), float(mult))


def stacked_inputs(im1, im2, flow, train_all=False):
    """The 14-channel input of a refinement network (flownet.py:46-57):
    ``concat([im1, im2, flow, warp, diff], axis=3)`` with
    ``flow = resize_bilinear(flow, [H, W]) * 4 * FLOW_SCALE``, ``warp = image_warp(im2, flow)``,
    ``diff = |warp - im1|``.  Without ``train_all`` flow, warp and diff are
    stop-gradient'd (:51-54) and the whole block is one fused kernel."""
    im1, im2, flow = _dev(im1, "im1"), _dev(im2, "im2"), _dev(flow, "flow")
    _require_rank4(im1, "im1")
    B, H, W, C = im1.shape
    if C != 3 or im2.shape != im1.shape:
        raise ValueError("stacked_inputs expects two [B,H,W,3] images, got %s and %s"
                         % (tuple(im1.shape), tuple(im2.shape)))
    if flow.dim() != 4 or flow.shape[0] != B or flow.shape[3] != 2:
        raise ValueError("flow must be [B,h,w,2], got %s" % (tuple(flow.shape),))
    mult = 4 * FLOW_SCALE
    if train_all:
        full = resize_bilinear(flow, (H, W), mult)
        warp = image_warp(im2, full)
        return torch.cat([im1, im2, full, warp, torch.abs(warp - im1)], dim=3)
    out = torch.empty((B, H, W, 14), device=im1.device, dtype=torch.float32)
    with _on(im1.device):
        _lib.check(_lib.load().unflow_stacked_inputs_fwd(
            _ptr(im1), _ptr(im2), _ptr(flow.detach()), _ptr(out), B, H, W, flow.shape[1], flow.shape[2],
            mult, _stream()))
    return out
