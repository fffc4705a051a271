"""``image_warp`` -- mirror of ``src/e2eflow/core/image_warp.py:4-76``.

The bilinear backward warp the loss really calls (losses.py:22-23,38-39; the
stacked nets, core/flownet.py:48-49): the FLOW is floored and the four taps are
clamped to the image.  One fused HIP kernel replaces the reference's
tile/range/clip/gather x4/add_n graph; the gradients TF autodiff derives
(scatter-add into the image, weight derivatives into the flow) are one kernel.
"""
import torch

from . import _lib
from .ops import _check_flow, _dev, _on, _ptr, _stream


def image_warp_raw(im, flow):
    im, flow = _dev(im, "im"), _dev(flow, "flow")
    _check_flow(im, flow)
    B, H, W, C = im.shape
    out = torch.empty_like(im)
    with _on(im.device):
        _lib.check(_lib.load().unflow_image_warp_fwd(_ptr(im), _ptr(flow), _ptr(out), B, H, W, C,
                                                     _stream()))
    return out


def image_warp_grad(grad, im, flow, want_im=True, want_flow=True):
    """(d_im, d_flow); each None when not requested."""
    grad, im, flow = _dev(grad, "grad"), _dev(im, "im"), _dev(flow, "flow")
    _check_flow(im, flow)
    B, H, W, C = im.shape
    d_im = torch.empty_like(im) if want_im else None
    d_flow = torch.empty_like(flow) if want_flow else None
    with _on(im.device):
        _lib.check(_lib.load().unflow_image_warp_bwd(_ptr(grad), _ptr(im), _ptr(flow), _ptr(d_im),
                                                     _ptr(d_flow), B, H, W, C, _stream()))
    return d_im, d_flow


class _ImageWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, flow):
        ctx.save_for_backward(im, flow)
        return image_warp_raw(im, flow)

    @staticmethod
    def backward(ctx, grad):
        im, flow = ctx.saved_tensors
        return image_warp_grad(grad, im, flow, ctx.needs_input_grad[0], ctx.needs_input_grad[1])


def image_warp(im, flow):
    """Mirror of ``image_warp`` (src/e2eflow/core/image_warp.py:4-76): samples ``im``
    [B,H,W,C] bilinearly at ``(x + flow[...,0], y + flow[...,1])`` with the tap indices
    clamped to the image (``floor`` is taken of the flow, as the reference does); returns
    a tensor shaped like ``im``.  Differentiable in both arguments."""
    return _ImageWarp.apply(im, flow)
