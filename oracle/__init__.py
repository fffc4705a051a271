"""ctypes front end of the CPU oracle (``oracle/unflow_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg.  ``unflow_amd`` never imports it.

All functions take and return C-contiguous float32 numpy arrays in the
reference's layouts (correlation NCHW, everything else NHWC).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libunflow_oracle.so")

_ERRORS = {
    1: "kernel_size must be odd",                                   # ops/correlation_op.h:16-17
    2: "Input shapes have to be the same",                          # ops/correlation_op.cc:47-48
    3: "Invalid correlation settings",                              # ops/correlation_op.cc:60-61
    4: "Input height and width must be divisible by scale",         # ops/downsample_op.cc:37-40
    5: "invalid argument",
}


def build(force=False):
    """Compile the oracle with gcc (``make -C oracle``)."""
    src = os.path.join(_HERE, "unflow_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libunflow_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.oracle_charbonnier_fwd.restype = ctypes.c_double
        L.oracle_ternary_loss_fwd.restype = ctypes.c_double
        L.oracle_smoothness_loss_fwd.restype = ctypes.c_double
        L.oracle_gradient_loss_fwd.restype = ctypes.c_double
        L.oracle_census_hamming.restype = ctypes.c_uint64
        L.oracle_census_hamming.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long]
        L.oracle_num_threads.restype = ctypes.c_int
        L.oracle_border_size.restype = ctypes.c_int
        L.oracle_border_size.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_float]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    if a is None:
        return ctypes.c_void_p(0)
    return a.ctypes.data_as(ctypes.c_void_p)


def _check(rc):
    if rc != 0:
        raise ValueError(_ERRORS.get(rc, "oracle error %d" % rc))


def num_threads():
    return lib().oracle_num_threads()


def set_num_threads(n):
    lib().oracle_set_num_threads(ctypes.c_int(int(n)))


# --------------------------------------------------------------------------- correlation
def correlation_geometry(H, W, kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2):
    dims = (ctypes.c_int * 5)()
    _check(lib().oracle_correlation_geometry(H, W, kernel_size, max_displacement, pad, stride_1,
                                             stride_2, dims))
    return dict(out_channels=dims[0], out_height=dims[1], out_width=dims[2],
                padded_height=dims[3], padded_width=dims[4])


def correlation(first, second, kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2):
    """Returns (correlation, padded_0, padded_1) like the 3-output reference op."""
    first, second = _f32(first), _f32(second)
    if first.shape != second.shape:
        raise ValueError(_ERRORS[2])
    B, C, H, W = first.shape
    g = correlation_geometry(H, W, kernel_size, max_displacement, pad, stride_1, stride_2)
    out = np.empty((B, g["out_channels"], g["out_height"], g["out_width"]), np.float32)
    p0 = np.empty((B, g["padded_height"], g["padded_width"], C), np.float32)
    p1 = np.empty_like(p0)
    _check(lib().oracle_correlation_fwd(_p(first), _p(second), _p(out), _p(p0), _p(p1), B, C, H, W,
                                        kernel_size, max_displacement, pad, stride_1, stride_2))
    return out, p0, p1


def correlation_grad(in_grad, in0, in1, padded0, padded1, kernel_size=1, max_displacement=20,
                     pad=20, stride_1=1, stride_2=2):
    in_grad, padded0, padded1 = _f32(in_grad), _f32(padded0), _f32(padded1)
    B, C, H, W = in0.shape
    g0 = np.empty((B, C, H, W), np.float32)
    g1 = np.empty((B, C, H, W), np.float32)
    _check(lib().oracle_correlation_bwd(_p(in_grad), _p(padded0), _p(padded1), _p(g0), _p(g1), B, C,
                                        H, W, kernel_size, max_displacement, pad, stride_1, stride_2))
    return g0, g1


# --------------------------------------------------------------------------- warps
def backward_warp(images, flows):
    images, flows = _f32(images), _f32(flows)
    B, H, W, C = images.shape
    out = np.empty_like(images)
    lib().oracle_backward_warp_fwd(_p(images), _p(flows), _p(out), B, H, W, C)
    return out


def backward_warp_grad(grads, images, flows):
    grads, images, flows = _f32(grads), _f32(images), _f32(flows)
    B, H, W, C = images.shape
    out = np.empty((B, H, W, 2), np.float32)
    lib().oracle_backward_warp_bwd(_p(grads), _p(images), _p(flows), _p(out), B, H, W, C)
    return out


def forward_warp(flows):
    flows = _f32(flows)
    B, H, W, _ = flows.shape
    out = np.empty((B, H, W, 1), np.float32)
    lib().oracle_forward_warp_fwd(_p(flows), _p(out), B, H, W)
    return out


def forward_warp_grad(grads, flows):
    grads, flows = _f32(grads), _f32(flows)
    B, H, W, _ = flows.shape
    out = np.empty((B, H, W, 2), np.float32)
    lib().oracle_forward_warp_bwd(_p(grads), _p(flows), _p(out), B, H, W)
    return out


def downsample(images, scale):
    images = _f32(images)
    B, H, W, C = images.shape
    if scale <= 0 or H % scale or W % scale:
        raise ValueError(_ERRORS[4])
    out = np.empty((B, H // scale, W // scale, C), np.float32)
    _check(lib().oracle_downsample(_p(images), _p(out), B, H, W, C, scale))
    return out


def resize_area(images, height, width):
    """tf.image.resize_area (TF 1.x, align_corners=False); assumption A3, parity unpinned."""
    images = _f32(images)
    B, H, W, C = images.shape
    out = np.empty((B, height, width, C), np.float32)
    lib().oracle_resize_area_fwd(_p(images), _p(out), B, H, W, C, height, width)
    return out


def util_downsample(tensor, num):
    """core/util.downsample (src/e2eflow/core/util.py:23-28): the op when both extents are
    even (it tests % 2 whatever num is), tf.image.resize_area otherwise."""
    _, height, width, _ = np.shape(tensor)
    if height % 2 == 0 and width % 2 == 0:
        return downsample(tensor, num)
    return resize_area(tensor, int(height / num), int(width / num))


def image_warp(im, flow):
    im, flow = _f32(im), _f32(flow)
    B, H, W, C = im.shape
    out = np.empty_like(im)
    lib().oracle_image_warp_fwd(_p(im), _p(flow), _p(out), B, H, W, C)
    return out


def image_warp_grad(grad, im, flow, want_im=True, want_flow=True):
    grad, im, flow = _f32(grad), _f32(im), _f32(flow)
    B, H, W, C = im.shape
    d_im = np.empty_like(im) if want_im else None
    d_flow = np.empty((B, H, W, 2), np.float32) if want_flow else None
    lib().oracle_image_warp_bwd(_p(grad), _p(im), _p(flow), _p(d_im), _p(d_flow), B, H, W, C)
    return d_im, d_flow


# --------------------------------------------------------------------------- masks / losses
def create_mask(shape_bhw, paddings):
    B, H, W = shape_bhw
    (pt, pb), (pl, pr) = paddings
    m = np.empty((B, H, W, 1), np.float32)
    lib().oracle_create_mask(_p(m), B, H, W, pt, pb, pl, pr)
    return m


def create_border_mask(shape_bhw, border_ratio=0.1):
    B, H, W = shape_bhw
    sz = lib().oracle_border_size(H, W, border_ratio)
    return create_mask(shape_bhw, [[sz, sz], [sz, sz]])


def create_outgoing_mask(flow):
    flow = _f32(flow)
    B, H, W, _ = flow.shape
    m = np.empty((B, H, W, 1), np.float32)
    lib().oracle_create_outgoing_mask(_p(flow), _p(m), B, H, W)
    return m


def fb_occlusion(flow, flow_other_warped):
    flow, fw = _f32(flow), _f32(flow_other_warped)
    B, H, W, _ = flow.shape
    diff = np.empty_like(flow)
    occ = np.empty((B, H, W, 1), np.float32)
    lib().oracle_fb_occlusion(_p(flow), _p(fw), _p(diff), _p(occ), B, H, W)
    return diff, occ


def charbonnier_loss(x, mask=None, truncate=None, alpha=0.45, beta=1.0, epsilon=0.001):
    x = _f32(x)
    B, H, W, C = x.shape
    mc = 0
    if mask is not None:
        mask = _f32(mask)
        mc = mask.shape[3]
        assert mc in (1, C)
    return lib().oracle_charbonnier_fwd(_p(x), _p(mask), mc, B, H, W, C, ctypes.c_float(alpha),
                                        ctypes.c_float(beta), ctypes.c_float(epsilon),
                                        ctypes.c_float(truncate if truncate else 0.0))


def charbonnier_loss_grad(x, mask=None, gout=1.0, truncate=None, alpha=0.45, beta=1.0, epsilon=0.001):
    x = _f32(x)
    B, H, W, C = x.shape
    mc = 0
    if mask is not None:
        mask = _f32(mask)
        mc = mask.shape[3]
    dx = np.empty_like(x)
    lib().oracle_charbonnier_bwd(_p(x), _p(mask), mc, ctypes.c_float(gout), _p(dx), B, H, W, C,
                                 ctypes.c_float(alpha), ctypes.c_float(beta), ctypes.c_float(epsilon),
                                 ctypes.c_float(truncate if truncate else 0.0))
    return dx


def ternary_loss(im1, im2_warped, mask, max_distance=1, return_dist=False):
    im1, im2_warped, mask = _f32(im1), _f32(im2_warped), _f32(mask)
    B, H, W, C = im1.shape
    assert C == 3
    dist = np.empty((B, H, W, 1), np.float32) if return_dist else None
    loss = lib().oracle_ternary_loss_fwd(_p(im1), _p(im2_warped), _p(mask), _p(dist), B, H, W,
                                         max_distance)
    return (loss, dist) if return_dist else loss


def ternary_loss_grad(im1, im2_warped, mask, max_distance=1, gout=1.0):
    im1, im2_warped, mask = _f32(im1), _f32(im2_warped), _f32(mask)
    B, H, W, C = im1.shape
    d = np.empty_like(im2_warped)
    lib().oracle_ternary_loss_bwd(_p(im1), _p(im2_warped), _p(mask), ctypes.c_float(gout), _p(d), B,
                                  H, W, max_distance)
    return d


def occlusion(flow_fw, flow_bw):
    """occlusion() (losses.py:125-134): thresholds from the unwarped magnitudes."""
    flow_fw, flow_bw = _f32(flow_fw), _f32(flow_bw)

    def lsq(a):
        return (a[..., 0:1] * a[..., 0:1] + a[..., 1:2] * a[..., 1:2]).astype(np.float32)
    mag_sq = lsq(flow_fw) + lsq(flow_bw)
    thresh = np.float32(0.01) * mag_sq + np.float32(0.5)
    diff_fw = flow_fw + image_warp(flow_bw, flow_fw)
    diff_bw = flow_bw + image_warp(flow_fw, flow_bw)
    return (lsq(diff_fw) > thresh).astype(np.float32), (lsq(diff_bw) > thresh).astype(np.float32)


# --------------------------------------------------------------------------- 3x3 stencil terms
def smoothness_deltas(flow, order=1):
    """_smoothness_deltas (order 1, losses.py:206-225) / _second_order_deltas (order 2, :258-287):
    (delta_u, delta_v, mask), each [B,H,W,K], K = 2 or 4."""
    flow = _f32(flow)
    B, H, W, C = flow.shape
    assert C == 2
    K = 2 if order == 1 else 4
    du, dv, m = (np.empty((B, H, W, K), np.float32) for _ in range(3))
    rc = lib().oracle_smoothness_deltas(_p(flow), _p(du), _p(dv), _p(m), B, H, W, order)
    if rc:
        raise ValueError("order must be 1 or 2")
    return du, dv, m


def smoothness_loss(flow, order=1, norm_count=0):
    """smoothness_loss (order 1, losses.py:250-255) / second_order_loss (order 2, :290-295)."""
    flow = _f32(flow)
    B, H, W, _ = flow.shape
    return lib().oracle_smoothness_loss_fwd(_p(flow), B, H, W, order, ctypes.c_double(norm_count))


def smoothness_loss_grad(flow, order=1, gout=1.0, norm_count=0):
    flow = _f32(flow)
    B, H, W, _ = flow.shape
    d = np.empty_like(flow)
    lib().oracle_smoothness_loss_bwd(_p(flow), ctypes.c_float(gout), _p(d), B, H, W, order,
                                     ctypes.c_double(norm_count))
    return d


def second_order_loss(flow, norm_count=0):
    return smoothness_loss(flow, 2, norm_count)


def gradient_delta(im1, im2_warped):
    """_gradient_delta (losses.py:228-241): [B,H,W,6]."""
    im1, im2_warped = _f32(im1), _f32(im2_warped)
    B, H, W, C = im1.shape
    assert C == 3
    diff = np.empty((B, H, W, 6), np.float32)
    lib().oracle_gradient_delta(_p(im1), _p(im2_warped), _p(diff), B, H, W)
    return diff


def gradient_loss(im1, im2_warped, mask, norm_count=0):
    """gradient_loss (losses.py:244-247)."""
    im1, im2_warped, mask = _f32(im1), _f32(im2_warped), _f32(mask)
    B, H, W, _ = im1.shape
    return lib().oracle_gradient_loss_fwd(_p(im1), _p(im2_warped), _p(mask), B, H, W,
                                          ctypes.c_double(norm_count))


def gradient_loss_grad(im1, im2_warped, mask, gout=1.0, norm_count=0):
    im1, im2_warped, mask = _f32(im1), _f32(im2_warped), _f32(mask)
    B, H, W, _ = im1.shape
    d = np.empty_like(im2_warped)
    lib().oracle_gradient_loss_bwd(_p(im1), _p(im2_warped), _p(mask), ctypes.c_float(gout), _p(d), B, H, W,
                                   ctypes.c_double(norm_count))
    return d


# --------------------------------------------------------------------------- composite
DISOCC_THRESH = 0.8  # losses.py:9


def compute_losses(im1, im2, flow_fw, flow_bw, border_mask=None, mask_occlusion="",
                   data_max_distance=1, terms=None):
    """compute_losses (losses.py:16-87), all eight entries (or only ``terms``, as the TF1
    executor only runs what a fetched scalar needs).  Returns (losses, aux)."""
    im2_warped = image_warp(im2, flow_fw)
    im1_warped = image_warp(im1, flow_bw)
    disocc_fw = (forward_warp(flow_fw) < DISOCC_THRESH).astype(np.float32)
    disocc_bw = (forward_warp(flow_bw) < DISOCC_THRESH).astype(np.float32)
    if border_mask is None:
        mask_fw = create_outgoing_mask(flow_fw)
        mask_bw = create_outgoing_mask(flow_bw)
    else:
        mask_fw = _f32(border_mask).copy()
        mask_bw = _f32(border_mask).copy()
    flow_bw_warped = image_warp(flow_bw, flow_fw)
    flow_fw_warped = image_warp(flow_fw, flow_bw)
    flow_diff_fw, fb_occ_fw = fb_occlusion(flow_fw, flow_bw_warped)
    flow_diff_bw, fb_occ_bw = fb_occlusion(flow_bw, flow_fw_warped)
    if mask_occlusion == "fb":
        mask_fw = mask_fw * (1 - fb_occ_fw)
        mask_bw = mask_bw * (1 - fb_occ_bw)
    elif mask_occlusion == "disocc":
        mask_fw = mask_fw * (1 - disocc_bw)
        mask_bw = mask_bw * (1 - disocc_fw)
    occ_fw, occ_bw = 1 - mask_fw, 1 - mask_bw
    table = {
        "sym": lambda: charbonnier_loss(occ_fw - disocc_bw) + charbonnier_loss(occ_bw - disocc_fw),
        "occ": lambda: charbonnier_loss(occ_fw) + charbonnier_loss(occ_bw),
        "photo": lambda: charbonnier_loss(_f32(im1) - im2_warped, mask_fw, beta=255) +
        charbonnier_loss(_f32(im2) - im1_warped, mask_bw, beta=255),
        "grad": lambda: gradient_loss(im1, im2_warped, mask_fw) + gradient_loss(im2, im1_warped, mask_bw),
        "smooth_1st": lambda: smoothness_loss(flow_fw, 1) + smoothness_loss(flow_bw, 1),
        "smooth_2nd": lambda: smoothness_loss(flow_fw, 2) + smoothness_loss(flow_bw, 2),
        "fb": lambda: charbonnier_loss(flow_diff_fw, mask_fw) + charbonnier_loss(flow_diff_bw, mask_bw),
        "ternary": lambda: ternary_loss(im1, im2_warped, mask_fw, data_max_distance) +
        ternary_loss(im2, im1_warped, mask_bw, data_max_distance),
    }
    losses = {k: fn() for k, fn in table.items() if terms is None or k in terms}
    aux = dict(im2_warped=im2_warped, im1_warped=im1_warped, mask_fw=mask_fw, mask_bw=mask_bw,
               disocc_fw=disocc_fw, disocc_bw=disocc_bw, flow_bw_warped=flow_bw_warped,
               flow_fw_warped=flow_fw_warped, flow_diff_fw=flow_diff_fw, flow_diff_bw=flow_diff_bw,
               fb_occ_fw=fb_occ_fw, fb_occ_bw=fb_occ_bw)
    return losses, aux


def compute_losses_grad(im1, im2, flow_fw, flow_bw, border_mask=None, mask_occlusion="",
                        data_max_distance=1, weights=None):
    """d(sum_k weights[k] * losses[k]) / d(flow_fw, flow_bw) of compute_losses (losses.py:16-87),
    composed the way TF's autodiff walks the graph:

    * ``ternary``, ``photo``, ``grad`` reach a flow through ``image_warp``'s flow gradient of the
      warped image (im1 / im2 are inputs, not variables);
    * ``fb``: ``flow_diff_fw = flow_fw + image_warp(flow_bw, flow_fw)`` (losses.py:38-41), so the
      charbonnier gradient g goes to flow_fw directly AND through the warp's flow gradient, and to
      flow_bw through the warp's image gradient (the scatter-add of tf.gather); mirrored for bw;
    * ``smooth_1st`` / ``smooth_2nd`` act on the flows themselves;
    * ``occ`` and ``sym`` are built from ``tf.cast(... > ...)`` maps only (losses.py:28-29,48-49,
      51-65): no gradient; the masks are constants for every other term for the same reason.

    Returns (d_flow_fw, d_flow_bw, losses)."""
    weights = dict(weights or {})
    im1, im2, flow_fw, flow_bw = _f32(im1), _f32(im2), _f32(flow_fw), _f32(flow_bw)
    losses, aux = compute_losses(im1, im2, flow_fw, flow_bw, border_mask, mask_occlusion, data_max_distance)
    d_fw = np.zeros_like(flow_fw, dtype=np.float64)
    d_bw = np.zeros_like(flow_bw, dtype=np.float64)

    def through_warp(g_warped, im, flow):
        return image_warp_grad(g_warped, im, flow, want_im=False)[1]

    for k, wgt in weights.items():
        if not wgt:
            continue
        w = np.float32(wgt)
        if k == "ternary":
            g2 = ternary_loss_grad(im1, aux["im2_warped"], aux["mask_fw"], data_max_distance)
            g1 = ternary_loss_grad(im2, aux["im1_warped"], aux["mask_bw"], data_max_distance)
            d_fw += w * through_warp(g2, im2, flow_fw)
            d_bw += w * through_warp(g1, im1, flow_bw)
        elif k == "photo":
            # photometric_loss(im1 - im2_warped, mask): d/d im2_warped = -dcharb (losses.py:24,198-199)
            g2 = -charbonnier_loss_grad(im1 - aux["im2_warped"], aux["mask_fw"], beta=255)
            g1 = -charbonnier_loss_grad(im2 - aux["im1_warped"], aux["mask_bw"], beta=255)
            d_fw += w * through_warp(g2, im2, flow_fw)
            d_bw += w * through_warp(g1, im1, flow_bw)
        elif k == "grad":
            g2 = gradient_loss_grad(im1, aux["im2_warped"], aux["mask_fw"])
            g1 = gradient_loss_grad(im2, aux["im1_warped"], aux["mask_bw"])
            d_fw += w * through_warp(g2, im2, flow_fw)
            d_bw += w * through_warp(g1, im1, flow_bw)
        elif k in ("smooth_1st", "smooth_2nd"):
            order = 1 if k == "smooth_1st" else 2
            d_fw += w * smoothness_loss_grad(flow_fw, order)
            d_bw += w * smoothness_loss_grad(flow_bw, order)
        elif k == "fb":
            g_fw = charbonnier_loss_grad(aux["flow_diff_fw"], aux["mask_fw"])
            g_bw = charbonnier_loss_grad(aux["flow_diff_bw"], aux["mask_bw"])
            to_bw, via_fw = image_warp_grad(g_fw, flow_bw, flow_fw)      # flow_bw is the warped "image"
            to_fw, via_bw = image_warp_grad(g_bw, flow_fw, flow_bw)
            d_fw += w * (g_fw.astype(np.float64) + via_fw + to_fw)
            d_bw += w * (g_bw.astype(np.float64) + via_bw + to_bw)
        elif k in ("occ", "sym"):
            pass
        else:
            raise KeyError(k)
    return d_fw.astype(np.float32), d_bw.astype(np.float32), losses


def pyramid_loss_grad(im1_norm, im2_norm, border_mask, flows_fw, flows_bw, params):
    """Gradient of pyramid_loss's combined scalar to every flow of the pyramid
    (core/unsupervised.py:114-141: level i sees flow * final_flow_scale / 2**i and weighs its
    layer loss by layer_weights[i]).  Returns (grads_fw, grads_bw), None where no loss sees the level."""
    layer_weights = [12.7, 4.35, 3.9, 3.4, 1.1]
    layer_patch_distances = [3, 2, 2, 1, 1]
    if params.get('full_res'):
        layer_weights = [12.7, 5.5, 5.0, 4.35, 3.9, 3.4, 1.1]
        layer_patch_distances = [3, 3] + layer_patch_distances
        im1_s, im2_s, mask_s = _f32(im1_norm), _f32(im2_norm), _f32(border_mask)
        final_flow_scale = 5.0 * 4
    else:
        im1_s, im2_s, mask_s = (util_downsample(t, 4) for t in (im1_norm, im2_norm, border_mask))
        final_flow_scale = 5.0
    weights = {k: params[k + '_weight'] for k in LOSSES if params.get(k + '_weight')}
    n_levels = len(flows_fw) if params.get('pyramid_loss') else 1
    gfw, gbw = [None] * len(flows_fw), [None] * len(flows_bw)
    for i in range(n_levels):
        flow_scale = np.float32(final_flow_scale / (2 ** i))
        d_fw, d_bw, _ = compute_losses_grad(im1_s, im2_s, _f32(flows_fw[i]) * flow_scale, _f32(flows_bw[i]) * flow_scale,
                                            mask_s if params.get('border_mask') else None,
                                            params.get('mask_occlusion', ''), layer_patch_distances[i], weights)
        gfw[i] = (np.float32(layer_weights[i]) * flow_scale * d_fw).astype(np.float32)
        gbw[i] = (np.float32(layer_weights[i]) * flow_scale * d_bw).astype(np.float32)
        if i + 1 < n_levels:
            im1_s, im2_s, mask_s = (util_downsample(t, 2) for t in (im1_s, im2_s, mask_s))
    return gfw, gbw


# --------------------------------------------------------------------------- integer census (extension)
def census_transform(im, max_distance=3, eps=0.9):
    """[B,H,W,2] uint64 ternary census signature (definition: include/unflow_hip.h)."""
    im = _f32(im)
    B, H, W, C = im.shape
    assert C == 3 and 1 <= max_distance <= 3
    sig = np.empty((B, H, W, 2), np.uint64)
    lib().oracle_census_transform(_p(im), sig.ctypes.data_as(ctypes.c_void_p), B, H, W, max_distance,
                                  ctypes.c_float(eps))
    return sig


def census_hamming(sig1, sig2, mask=None):
    sig1, sig2 = np.ascontiguousarray(sig1, np.uint64), np.ascontiguousarray(sig2, np.uint64)
    B, H, W, _ = sig1.shape
    dist = np.empty((B, H, W), np.int32)
    m = None if mask is None else _f32(mask)
    total = lib().oracle_census_hamming(sig1.ctypes.data_as(ctypes.c_void_p), sig2.ctypes.data_as(ctypes.c_void_p),
                                        _p(m), dist.ctypes.data_as(ctypes.c_void_p), B * H * W)
    return dist, int(total)


# --------------------------------------------------------------------------- stacked-net warp block
def resize_bilinear(x, height, width, mult=1.0):
    """tf.image.resize_bilinear (TF 1.x, align_corners=False) times ``mult``."""
    x = _f32(x)
    B, h, w, C = x.shape
    out = np.empty((B, height, width, C), np.float32)
    lib().oracle_resize_bilinear_fwd(_p(x), _p(out), B, h, w, height, width, C, ctypes.c_float(mult))
    return out


def resize_bilinear_grad(gout, h, w, mult=1.0):
    gout = _f32(gout)
    B, H, W, C = gout.shape
    gin = np.empty((B, h, w, C), np.float32)
    lib().oracle_resize_bilinear_bwd(_p(gout), _p(gin), B, h, w, H, W, C, ctypes.c_float(mult))
    return gin


def stacked_inputs(im1, im2, flow_lr, mult=20.0):
    """flownet.py:46-57: concat([im1, im2, flow, warp, |warp - im1|]) with
    flow = resize_bilinear(flow_lr) * 4 * FLOW_SCALE."""
    im1, im2, flow_lr = _f32(im1), _f32(im2), _f32(flow_lr)
    B, H, W, _ = im1.shape
    _, h, w, _ = flow_lr.shape
    out = np.empty((B, H, W, 14), np.float32)
    lib().oracle_stacked_inputs(_p(im1), _p(im2), _p(flow_lr), _p(out), B, H, W, h, w, ctypes.c_float(mult))
    return out


LOSSES = ['occ', 'sym', 'fb', 'grad', 'ternary', 'photo', 'smooth_1st', 'smooth_2nd']  # unsupervised.py:10


def pyramid_loss(im1_norm, im2_norm, border_mask, flows_fw, flows_bw, params):
    """Loss section of unsupervised_loss (core/unsupervised.py:83-145), flow pyramid given.
    Returns (combined_loss, combined_losses)."""
    layer_weights = [12.7, 4.35, 3.9, 3.4, 1.1]
    layer_patch_distances = [3, 2, 2, 1, 1]
    if params.get('full_res'):
        layer_weights = [12.7, 5.5, 5.0, 4.35, 3.9, 3.4, 1.1]
        layer_patch_distances = [3, 3] + layer_patch_distances
        im1_s, im2_s, mask_s = _f32(im1_norm), _f32(im2_norm), _f32(border_mask)
        final_flow_scale = 5.0 * 4
    else:
        im1_s, im2_s, mask_s = (util_downsample(t, 4) for t in (im1_norm, im2_norm, border_mask))
        final_flow_scale = 5.0
    combined_losses = {k: 0.0 for k in LOSSES}
    combined_loss = 0.0
    pairs = list(enumerate(zip(flows_fw, flows_bw))) if params.get('pyramid_loss') \
        else [(0, (flows_fw[0], flows_bw[0]))]
    for n, (i, (ffw, fbw)) in enumerate(pairs):
        flow_scale = np.float32(final_flow_scale / (2 ** i))
        losses, _ = compute_losses(im1_s, im2_s, _f32(ffw) * flow_scale, _f32(fbw) * flow_scale,
                                   mask_s if params.get('border_mask') else None,
                                   params.get('mask_occlusion', ''), layer_patch_distances[i])
        layer_loss = 0.0
        for k in LOSSES:
            if params.get(k + '_weight'):
                layer_loss += params[k + '_weight'] * losses[k]
                combined_losses[k] += layer_weights[i] * losses[k]
        combined_loss += layer_weights[i] * layer_loss
        if n + 1 < len(pairs):
            im1_s, im2_s, mask_s = (util_downsample(t, 2) for t in (im1_s, im2_s, mask_s))
    return combined_loss, combined_losses
