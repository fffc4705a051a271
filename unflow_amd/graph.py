"""HIP-graph capture of a launch-bound sequence of ops.

Every entry point of ``libunflow_hip.so`` takes the stream explicitly, allocates
nothing and never synchronises, so a whole loss pyramid (about 350 small
launches at the coarse levels) can be recorded once on a capture stream and
replayed as one graph launch: 5.0 ms -> 2.1 ms for the 5-level KITTI pyramid at
B=32 (tools/graph_pyramid.py).  The reference gets the same effect from the TF1
executor running a static graph; this is its MI355X counterpart.
"""
import torch

# torch exposes a setter but no getter for this flag (default: on); GraphedStep switches the warning off only while it
# captures and puts back what the process last set THROUGH set_accumulate_grad_stream_warning() below (True if never).
_warn_stream_mismatch = True


def set_accumulate_grad_stream_warning(on):
    """Process-wide: autograd's warning about gradient accumulation across streams (what GraphedStep restores)."""
    global _warn_stream_mismatch
    _warn_stream_mismatch = bool(on)
    setter = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
    if setter is not None:
        setter(_warn_stream_mismatch)


def _detached(x):
    if isinstance(x, torch.Tensor):
        return x.detach()
    if isinstance(x, (tuple, list)):
        return type(x)(_detached(v) for v in x)
    if isinstance(x, dict):
        return {k: _detached(v) for k, v in x.items()}
    return x


class GraphedStep:
    """Capture ``fn()`` (which reads fixed input tensors and returns tensors) into a
    HIP graph.  ``replay()`` re-runs it on the current contents of those inputs and
    returns the same output tensors, overwritten in place."""

    def __init__(self, fn, warmup=2):
        # The leaves of fn's autograd graph were created on the caller's stream while warm-up and capture run on
        # side streams (capture requires it): the gradient accumulation then crosses streams BY DESIGN -- the
        # streams are joined explicitly below -- so autograd's warning about it is switched off for this scope.
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # warm-up off the default stream, as capture requires
                for _ in range(warmup):
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                out = fn()
            # Replays need the output BUFFERS, not the autograd graph behind them: dropping it here also frees the
            # gradient accumulators that were created on the capture stream, so that a later eager backward of the
            # same leaves (on the caller's stream) does not meet them.
            self.outputs = _detached(out)
        finally:
            if quiet is not None:
                quiet(_warn_stream_mismatch)

    def replay(self):
        self.graph.replay()
        return self.outputs
