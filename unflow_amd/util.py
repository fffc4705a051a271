"""``downsample`` dispatcher -- mirror of ``src/e2eflow/core/util.py:23-28``."""
import torch

from . import _lib
from .ops import _dev, _on, _ptr, _require_rank4, _stream
from .ops import downsample as downsample_ops


def resize_area(images, size):
    """``tf.image.resize_area(images, size)`` (TensorFlow 1.x, align_corners=False): box
    average with fractional coverage weights, [B,H,W,C] -> [B,size[0],size[1],C].
    Third-party arithmetic restated from its published algorithm (assumption A3 of
    DESIGN.md; parity unpinned: the reference holds no fixture for it).  Not differentiable
    (the reference only resizes images and masks with it)."""
    oh, ow = int(size[0]), int(size[1])
    if oh <= 0 or ow <= 0:
        raise ValueError("output dimensions must be positive")        # resize op's InvalidArgument
    images = _dev(images.detach(), "images")
    _require_rank4(images, "images")
    B, H, W, C = images.shape
    if B * C > 0 and (H == 0 or W == 0):
        raise ValueError("input image must be of non-zero size")
    out = torch.empty((B, oh, ow, C), device=images.device, dtype=torch.float32)
    with _on(images.device):
        _lib.check(_lib.load().unflow_resize_area_fwd(_ptr(images), _ptr(out), B, H, W, C, oh, ow, _stream()))
    return out


def downsample(tensor, num):
    """The reference tests ``% 2`` whatever ``num`` is (util.py:25): an input whose
    extents are even but not divisible by ``num`` reaches the op and raises
    "Input height and width must be divisible by scale".  Odd extents take
    ``tf.image.resize_area`` to ``(int(height / num), int(width / num))`` (util.py:28)."""
    _, height, width, _ = tensor.shape
    if height % 2 == 0 and width % 2 == 0:
        return downsample_ops(tensor, num)
    return resize_area(tensor, (int(height / num), int(width / num)))
