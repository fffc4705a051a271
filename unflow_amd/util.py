"""``downsample`` dispatcher -- mirror of ``src/e2eflow/core/util.py:23-28``."""
from .ops import downsample as downsample_ops


def downsample(tensor, num):
    """The reference tests ``% 2`` whatever ``num`` is (util.py:25): an input whose
    extents are even but not divisible by ``num`` reaches the op and raises
    "Input height and width must be divisible by scale".  Odd extents take the
    reference's ``tf.image.resize_area`` branch, which is outside this path."""
    _, height, width, _ = tensor.shape
    if height % 2 == 0 and width % 2 == 0:
        return downsample_ops(tensor, num)
    raise NotImplementedError(
        "odd extents %dx%d: the reference falls back to tf.image.resize_area (core/util.py:28), "
        "which is not part of the hot path" % (height, width))
