"""Custom-op surface of the hot path -- mirror of ``src/e2eflow/ops.py``.

Same names, keyword arguments, defaults, tensor layouts and error messages as
the reference's ``ops.correlation / backward_warp / forward_warp / downsample``
(ops.py:69-75) and their registered gradients (ops.py:80-107).  Tensors are
float32 ``torch`` CUDA tensors used purely as device-memory containers; every
function forwards raw pointers to the C ABI of ``libunflow_hip.so`` on torch's
current HIP stream.  The differentiable entry points are ``torch.autograd``
functions wired exactly like the reference's ``RegisterGradient`` table.
"""
import ctypes

import torch

from . import _lib

OP_NAMES = ['backward_warp', 'downsample', 'correlation', 'forward_warp']  # ops.py:11


# --------------------------------------------------------------------------- plumbing
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """torch's current HIP stream of the current device as the ABI's stream argument (the raw handle: building a
    torch.cuda.Stream object per launch is a few microseconds of the ~15 a launch costs from Python)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Here:
    """Stands in for ``torch.cuda.device(d)`` when d is the current device already."""
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_HERE = _Here()


def _on(device):
    """``with _on(t.device):`` -- make t's device current for the launch; a no-op context in the usual case that it is
    already (entering torch.cuda.device costs a get / set device pair per launch)."""
    return _HERE if device.index == torch.cuda.current_device() else torch.cuda.device(device)


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _dev(t, name):
    """float32, contiguous, on the GPU -- there is no CPU kernel (the reference
    registers DEVICE_GPU only, ops/correlation_op.cc:189-194)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor, got %s" % (name, type(t).__name__))
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA/HIP tensor: unflow_amd has no CPU kernels" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    return t.contiguous()


def _require_rank4(t, name):
    if t.dim() != 4:
        raise ValueError("%s must have rank 4, got shape %s" % (name, tuple(t.shape)))


def correlation_output_shape(in_shape, kernel_size=1, max_displacement=20, pad=20, stride_1=1,
                             stride_2=2):
    """CorrelationState (ops/correlation_op.h:28-51) for an NCHW input shape.
    Host-only; raises ValueError with the reference's messages."""
    B, C, H, W = in_shape
    dims = (ctypes.c_int * 5)()
    _lib.check(_lib.load().unflow_correlation_geometry(H, W, kernel_size, max_displacement, pad,
                                                       stride_1, stride_2, dims))
    return dict(correlation=(B, dims[0], dims[1], dims[2]), padded=(B, dims[3], dims[4], C))


# --------------------------------------------------------------------------- raw ops
def _correlation_raw(first, second, materialize_padded, kernel_size, max_displacement, pad,
                     stride_1, stride_2):
    first, second = _dev(first, "input_0"), _dev(second, "input_1")
    _require_rank4(first, "input_0")
    if first.shape != second.shape:
        raise ValueError("Input shapes have to be the same")  # ops/correlation_op.cc:47-48
    B, C, H, W = first.shape
    shapes = correlation_output_shape(first.shape, kernel_size, max_displacement, pad, stride_1,
                                      stride_2)
    out = torch.empty(shapes["correlation"], device=first.device, dtype=torch.float32)
    p0 = p1 = None
    if materialize_padded:
        p0 = torch.empty(shapes["padded"], device=first.device, dtype=torch.float32)
        p1 = torch.empty(shapes["padded"], device=first.device, dtype=torch.float32)
    with _on(first.device):
        _lib.check(_lib.load().unflow_correlation_fwd(
            _ptr(first), _ptr(second), _ptr(out), _ptr(p0), _ptr(p1), B, C, H, W, kernel_size,
            max_displacement, pad, stride_1, stride_2, _stream()))
    return out, p0, p1


def correlation_grad(in_grad, original_input_0, original_input_1, padded_0=None, padded_1=None,
                     kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2):
    """``_correlation_module.correlation_grad`` (ops/correlation_op.cc:87-129).
    padded_0/padded_1 are accepted for signature parity and not read."""
    in_grad = _dev(in_grad, "input_grad")
    in0, in1 = _dev(original_input_0, "original_input_0"), _dev(original_input_1, "original_input_1")
    B, C, H, W = in0.shape
    shapes = correlation_output_shape(in0.shape, kernel_size, max_displacement, pad, stride_1,
                                      stride_2)
    if tuple(in_grad.shape) != shapes["correlation"]:
        raise ValueError("input_grad has shape %s, expected %s" %
                         (tuple(in_grad.shape), shapes["correlation"]))
    g0, g1 = torch.empty_like(in0), torch.empty_like(in0)
    with _on(in0.device):
        _lib.check(_lib.load().unflow_correlation_bwd(
            _ptr(in_grad), _ptr(in0), _ptr(in1), _ptr(padded_0), _ptr(padded_1), _ptr(g0), _ptr(g1),
            B, C, H, W, kernel_size, max_displacement, pad, stride_1, stride_2, _stream()))
    return g0, g1


def _check_flow(images, flows):
    _require_rank4(flows, "flows")
    if flows.shape[3] != 2:
        raise ValueError("flows must have 2 channels, got shape %s" % (tuple(flows.shape),))
    if images is not None:
        _require_rank4(images, "images")
        if images.shape[:3] != flows.shape[:3]:
            raise ValueError("images %s and flows %s must agree in batch, height and width" %
                             (tuple(images.shape), tuple(flows.shape)))


def _backward_warp_raw(images, flows):
    images, flows = _dev(images, "images"), _dev(flows, "flows")
    _check_flow(images, flows)
    B, H, W, C = images.shape
    out = torch.empty_like(images)
    with _on(images.device):
        _lib.check(_lib.load().unflow_backward_warp_fwd(_ptr(images), _ptr(flows), _ptr(out), B, H,
                                                        W, C, _stream()))
    return out


def backward_warp_grad(grads, original_images, original_flows):
    """``_backward_warp_module.backward_warp_grad`` (ops/backward_warp_op.cc:54-75)."""
    grads, images = _dev(grads, "grads"), _dev(original_images, "original_images")
    flows = _dev(original_flows, "original_flows")
    _check_flow(images, flows)
    B, H, W, C = images.shape
    out = torch.empty_like(flows)
    with _on(images.device):
        _lib.check(_lib.load().unflow_backward_warp_bwd(_ptr(grads), _ptr(images), _ptr(flows),
                                                        _ptr(out), B, H, W, C, _stream()))
    return out


def _forward_warp_raw(flows):
    flows = _dev(flows, "flows")
    _check_flow(None, flows)
    B, H, W, _ = flows.shape
    out = torch.empty((B, H, W, 1), device=flows.device, dtype=torch.float32)
    with _on(flows.device):
        _lib.check(_lib.load().unflow_forward_warp_fwd(_ptr(flows), _ptr(out), B, H, W, _stream()))
    return out


def forward_warp_grad(grads, original_flows):
    """``_forward_warp_module.forward_warp_grad`` (ops/forward_warp_op.cc:57-76)."""
    grads, flows = _dev(grads, "grads"), _dev(original_flows, "original_flows")
    _check_flow(None, flows)
    B, H, W, _ = flows.shape
    out = torch.empty_like(flows)
    with _on(flows.device):
        _lib.check(_lib.load().unflow_forward_warp_bwd(_ptr(grads), _ptr(flows), _ptr(out), B, H, W,
                                                       _stream()))
    return out


def downsample(images, scale=2):
    """``ops.downsample`` (ops.py:74; attr default ops/downsample_op.cc:66).
    Not differentiable (ops.py:107): the result is detached."""
    images = _dev(images, "images")
    _require_rank4(images, "images")
    B, H, W, C = images.shape
    if scale <= 0 or H % scale != 0 or W % scale != 0:
        raise ValueError("Input height and width must be divisible by scale")
    out = torch.empty((B, H // scale, W // scale, C), device=images.device, dtype=torch.float32)
    with _on(images.device):
        _lib.check(_lib.load().unflow_downsample_fwd(_ptr(images), _ptr(out), B, H, W, C, scale,
                                                     _stream()))
    return out


def downsample_pyramid(im_a, im_b=None, levels=5, both_orders=False):
    """The image chain of the loss pyramid (core/unsupervised.py:97-103,143-145) as ONE launch (extension):
    ``downsample(x, 4)`` followed by ``levels - 1`` times ``downsample(., 2)``, every level bit-identical to the
    chain of ``downsample`` calls.  With ``im_b`` the levels are the stacked batches ``[a; b]`` (and, with
    ``both_orders``, also ``[b; a]``) along the batch.  Returns the list of levels (or the two lists).  One or
    three channels, at most five levels, extents divisible by ``4 << (levels - 1)``; not differentiable."""
    im_a = _dev(im_a.detach(), "im_a")
    _require_rank4(im_a, "im_a")
    if im_b is not None:
        im_b = _dev(im_b.detach(), "im_b")
        if im_b.shape != im_a.shape:
            raise ValueError("downsample_pyramid: im_a %s and im_b %s must have the same shape" % (tuple(im_a.shape), tuple(im_b.shape)))
    elif both_orders:
        raise ValueError("downsample_pyramid: both_orders needs two tensors")
    B, H, W, C = im_a.shape
    if not 1 <= levels <= 5:
        raise ValueError("downsample_pyramid: 1 to 5 levels")
    step = 4 << (levels - 1)
    if H % step != 0 or W % step != 0:
        raise ValueError("Input height and width must be divisible by scale")
    nb = B * (2 if im_b is not None else 1)
    mk = lambda: [torch.empty((nb, H // (4 << l), W // (4 << l), C), device=im_a.device, dtype=torch.float32) for l in range(levels)]
    ab = mk()
    ba = mk() if both_orders else None
    arr = lambda ts: (ctypes.c_void_p * levels)(*[t.data_ptr() for t in ts])
    with _on(im_a.device):
        _lib.check(_lib.load().unflow_downsample_pyramid(_ptr(im_a), _ptr(im_b), arr(ab), arr(ba) if ba else None,
                                                         levels, B, H, W, C, _stream()))
    return (ab, ba) if both_orders else ab


# --------------------------------------------------------------------------- gradients (ops.py:80-104)
class _Correlation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, first, second, kernel_size, max_displacement, pad, stride_1, stride_2):
        out, _, _ = _correlation_raw(first, second, False, kernel_size, max_displacement, pad,
                                     stride_1, stride_2)
        ctx.save_for_backward(first, second)
        ctx.attrs = (kernel_size, max_displacement, pad, stride_1, stride_2)
        return out

    @staticmethod
    def backward(ctx, in_grad):
        first, second = ctx.saved_tensors
        k, md, pad, s1, s2 = ctx.attrs
        g0, g1 = correlation_grad(in_grad, first, second, None, None, kernel_size=k,
                                  max_displacement=md, pad=pad, stride_1=s1, stride_2=s2)
        return g0, g1, None, None, None, None, None


class _BackwardWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, flows):
        ctx.save_for_backward(images, flows)
        return _backward_warp_raw(images, flows)

    @staticmethod
    def backward(ctx, grad):
        images, flows = ctx.saved_tensors
        return None, backward_warp_grad(grad, images, flows)  # ops.py:84: [None, grad0]


class _ForwardWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flows):
        ctx.save_for_backward(flows)
        return _forward_warp_raw(flows)

    @staticmethod
    def backward(ctx, grad):
        (flows,) = ctx.saved_tensors
        return forward_warp_grad(grad, flows)


# --------------------------------------------------------------------------- public surface (ops.py:69-75)
def correlation(first, second, kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2):
    """``ops.correlation`` (ops.py:69-70): output 0 of the three-output op.
    NCHW in, NCHW ``[B, D, Ho, Wo]`` out; attr defaults ops/correlation_op.cc:136-140."""
    return _Correlation.apply(first, second, kernel_size, max_displacement, pad, stride_1, stride_2)


def correlation_flip(volume, add=None, kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2):
    """``add + flip(volume)``: flip maps the cost volume of (a, b) onto that of (b, a),
    ``flip(c)[d, p] = c[-d, p + stride_2*d]`` (0 outside the image); it is its own inverse."""
    volume = _dev(volume, "volume")
    _require_rank4(volume, "volume")
    if add is not None:
        add = _dev(add, "add")
        if add.shape != volume.shape:
            raise ValueError("add must have the shape of volume")
    B, D, H, W = volume.shape
    want = correlation_output_shape((B, 1, H, W), kernel_size, max_displacement, pad, stride_1, stride_2)["correlation"]
    if (D, H, W) != tuple(want[1:]):
        raise ValueError("volume has shape %s, the attrs imply [B,%d,%d,%d]" % ((tuple(volume.shape),) + tuple(want[1:])))
    out = torch.empty_like(volume)
    with _on(volume.device):
        _lib.check(_lib.load().unflow_correlation_flip(_ptr(volume), _ptr(add), _ptr(out), B, D, H, W, kernel_size,
                                                       max_displacement, pad, stride_1, stride_2, _stream()))
    return out


class _CorrelationBidirectional(torch.autograd.Function):
    @staticmethod
    def forward(ctx, first, second, attrs):
        c_fw, _, _ = _correlation_raw(first, second, False, *attrs)
        c_bw = correlation_flip(c_fw, None, *attrs)
        ctx.save_for_backward(first, second)
        ctx.attrs = attrs
        return c_fw, c_bw

    @staticmethod
    def backward(ctx, g_fw, g_bw):
        first, second = ctx.saved_tensors
        top = correlation_flip(g_bw.contiguous(), g_fw.contiguous(), *ctx.attrs)      # g_fw + flip(g_bw)
        k, md, pad, s1, s2 = ctx.attrs
        g0, g1 = correlation_grad(top, first, second, None, None, kernel_size=k, max_displacement=md, pad=pad,
                                  stride_1=s1, stride_2=s2)
        return g0, g1, None


def correlation_bidirectional(first, second, kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2):
    """Extension for the bidirectional FlowNetC (core/flownet.py:33-43 calls ``correlation(a, b)``
    and ``correlation(b, a)``): returns both cost volumes from ONE correlation -- the second is
    a bit-exact re-indexing of the first -- and its backward folds both upstream gradients into
    one ``correlation_grad`` call that returns the summed gradients.  Needs kernel_size 1,
    stride_1 1, pad == max_displacement."""
    return _CorrelationBidirectional.apply(first, second, (kernel_size, max_displacement, pad, stride_1, stride_2))


def _half(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float16:
        raise TypeError("%s must be a float16 CUDA/HIP tensor" % name)
    return t.contiguous()


def _hp(t):
    return ctypes.c_void_p(t.data_ptr())


class _CorrelationFP16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, first, second, attrs, exact):
        first, second = _half(first, "first"), _half(second, "second")
        _require_rank4(first, "first")
        if first.shape != second.shape:
            raise ValueError("Input shapes have to be the same")
        B, C, H, W = first.shape
        shapes = correlation_output_shape(first.shape, *attrs)
        out = torch.empty(shapes["correlation"], device=first.device, dtype=torch.float16)
        lib = _lib.load()
        fn = lib.unflow_correlation_fwd_f16_exact if exact else lib.unflow_correlation_fwd_f16
        with _on(first.device):
            _lib.check(fn(_hp(first), _hp(second), _hp(out), B, C, H, W, *attrs, _stream()))
        ctx.save_for_backward(first, second)
        ctx.attrs, ctx.exact = attrs, exact
        return out

    @staticmethod
    def backward(ctx, in_grad):
        first, second = ctx.saved_tensors
        g0, g1 = correlation_fp16_grad(in_grad, first, second, *ctx.attrs, exact=ctx.exact)
        return g0, g1, None, None


def correlation_fp16_grad(in_grad, original_input_0, original_input_1, kernel_size=1, max_displacement=20,
                          pad=20, stride_1=1, stride_2=2, exact=False):
    """Gradients of ``correlation_fp16``: all tensors float16, f32 sums (``exact``: of ``correlation_fp16_exact``)."""
    in_grad = _half(in_grad, "input_grad")
    in0, in1 = _half(original_input_0, "original_input_0"), _half(original_input_1, "original_input_1")
    B, C, H, W = in0.shape
    attrs = (kernel_size, max_displacement, pad, stride_1, stride_2)
    shapes = correlation_output_shape(in0.shape, *attrs)
    if tuple(in_grad.shape) != shapes["correlation"]:
        raise ValueError("input_grad has shape %s, expected %s" % (tuple(in_grad.shape), shapes["correlation"]))
    g0, g1 = torch.empty_like(in0), torch.empty_like(in0)
    lib = _lib.load()
    fn = lib.unflow_correlation_bwd_f16_exact if exact else lib.unflow_correlation_bwd_f16
    with _on(in0.device):
        _lib.check(fn(_hp(in_grad), _hp(in0), _hp(in1), _hp(g0), _hp(g1), B, C, H, W, *attrs, _stream()))
    return g0, g1


def correlation_fp16(first, second, kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2):
    """Extension (not in the reference, whose ops are float32 only; BASELINE config 5): ``correlation`` on
    half-precision tensors on the f16 matrix cores -- halves in HBM, exact f32 products, f32
    accumulation inside ``v_mfma_f32_16x16x16_f16``, one rounding to half on output; differentiable.
    FlowNetC attributes only."""
    return _CorrelationFP16.apply(first, second, (kernel_size, max_displacement, pad, stride_1, stride_2), False)


def correlation_fp16_exact(first, second, kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2):
    """``correlation_fp16`` with fp16 STORAGE only: the halves are widened and run through the float32
    op's f32 MFMA loop (its k-ordered fmaf chain on the widened tensors, rounded to half once).
    Slower than the float32 op; kept as the exact form."""
    return _CorrelationFP16.apply(first, second, (kernel_size, max_displacement, pad, stride_1, stride_2), True)


class _WarpFP16(torch.autograd.Function):
    """backward_warp / image_warp with fp16 storage: flow gradient only (the reference registers no
    image gradient for backward_warp either; image_warp's is an f32 scatter-add)."""

    @staticmethod
    def forward(ctx, images, flows, clamp):
        images, flows = _half(images, "images"), _half(flows, "flows")
        _require_rank4(images, "images")
        _check_flow(images, flows)
        B, H, W, C = images.shape
        out = torch.empty_like(images)
        fn = _lib.load().unflow_image_warp_fwd_f16 if clamp else _lib.load().unflow_backward_warp_fwd_f16
        with _on(images.device):
            _lib.check(fn(_hp(images), _hp(flows), _hp(out), B, H, W, C, _stream()))
        ctx.save_for_backward(images, flows)
        ctx.clamp = clamp
        return out

    @staticmethod
    def backward(ctx, grad):
        images, flows = ctx.saved_tensors
        grad = _half(grad, "grad")
        B, H, W, C = images.shape
        d_flow = torch.empty_like(flows)
        fn = _lib.load().unflow_image_warp_bwd_flow_f16 if ctx.clamp else _lib.load().unflow_backward_warp_bwd_f16
        with _on(images.device):
            _lib.check(fn(_hp(grad), _hp(images), _hp(flows), _hp(d_flow), B, H, W, C, _stream()))
        return None, d_flow, None


def backward_warp_fp16(images, flows):
    """Extension (BASELINE config 5): ``backward_warp`` on float16 tensors -- fp16 in HBM, the f32
    kernel's arithmetic, one rounding to half per output; differentiable in the flows.  1 <= C <= 3."""
    return _WarpFP16.apply(images, flows, False)


def image_warp_fp16(im, flow):
    """Extension (BASELINE config 5): ``image_warp`` (core/image_warp.py:4) on float16 tensors;
    differentiable in the flow (the image gradient needs the float32 op).  1 <= C <= 3."""
    return _WarpFP16.apply(im, flow, True)


def forward_warp_fp16(flows):
    """Extension (BASELINE config 5): ``forward_warp`` of float16 flows; the density comes back in
    float32, the precision it is accumulated in.  Not differentiable."""
    flows = _half(flows, "flows")
    _require_rank4(flows, "flows")
    if flows.shape[3] != 2:
        raise ValueError("flows must have 2 channels")
    B, H, W, _ = flows.shape
    out = torch.empty((B, H, W, 1), device=flows.device, dtype=torch.float32)
    with _on(flows.device):
        _lib.check(_lib.load().unflow_forward_warp_fwd_f16(_hp(flows), ctypes.c_void_p(out.data_ptr()), B, H, W, _stream()))
    return out


def correlation_with_padded(first, second, kernel_size=1, max_displacement=20, pad=20, stride_1=1,
                            stride_2=2):
    """All three outputs of the reference op (``_correlation_module.correlation``):
    (correlation, padded_0, padded_1).  Not differentiable."""
    return _correlation_raw(first, second, True, kernel_size, max_displacement, pad, stride_1,
                            stride_2)


def backward_warp(images, flows):
    """``ops.backward_warp`` (ops.py:73): NHWC images, flows [B,H,W,2]; zero outside."""
    return _BackwardWarp.apply(images, flows)


def forward_warp(flows):
    """``ops.forward_warp`` (ops.py:75): flows [B,H,W,2] -> density [B,H,W,1]."""
    return _ForwardWarp.apply(flows)
