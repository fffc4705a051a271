"""unflow_amd -- MI355X-native implementation of UnFlow's per-pair dense hot path.

FlowNetC correlation cost volume, backward / forward warping and the soft
ternary-census loss (forward and backward) as hand-written HIP kernels for
gfx950 behind a C ABI (``include/unflow_hip.h``), exposed under the reference's
own op names and signatures:

    unflow_amd.ops         <->  src/e2eflow/ops.py
    unflow_amd.image_warp  <->  src/e2eflow/core/image_warp.py
    unflow_amd.losses      <->  src/e2eflow/core/losses.py
    unflow_amd.util        <->  src/e2eflow/core/util.py     (downsample dispatch)
    unflow_amd.unsupervised <-> src/e2eflow/core/unsupervised.py:83-145 (pyramid loss)
    unflow_amd.flownet     <->  src/e2eflow/core/flownet.py:46-57 (stacked-net warp block)

There is no CPU fallback: every op raises if ``libunflow_hip.so`` is missing or
a tensor does not live on the GPU.
"""
from . import _lib  # noqa: F401

__all__ = ["ops", "image_warp", "losses", "util", "unsupervised", "flownet", "graph", "dist"]
