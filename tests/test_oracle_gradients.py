"""Numeric-gradient checks of the oracle's backward functions, mirroring the
reference's gradient_checker.compute_gradient tests at rtol=atol=1e-3
(test/ops/correlation.py:21-28, backward_warp.py:21-25, forward_warp.py:9-19).
The numeric side differentiates the independent float64 numpy restatement."""
import numpy as np
import pytest

from oracle import np_ref

TOL = dict(rtol=1e-3, atol=1e-3)


def A(x):
    return np.asarray(x, np.float32)


def away_from_integers(rng, shape, scale):
    """Flows whose fractional part stays in [0.15, 0.85]: bilinear kinks and the
    splat window's floor() jumps sit at integers."""
    base = np.floor(rng.standard_normal(shape) * scale)
    return (base + rng.uniform(0.15, 0.85, shape)).astype(np.float32)


@pytest.mark.parametrize("name", ["correlation_trivial", "correlation_batch", "correlation_3x3_inputs"])
def test_correlation_grad_reference_cases(kat, oracle_mod, name):
    case = kat[name]
    f0, f1, attrs = A(case["first"]), A(case["second"]), case["attrs"]
    out, p0, p1 = oracle_mod.correlation(f0, f1, **attrs)
    rng = np.random.default_rng(0)
    top = rng.standard_normal(out.shape).astype(np.float32)
    g0, g1 = oracle_mod.correlation_grad(top, f0, f1, p0, p1, **attrs)
    n0 = np_ref.numeric_grad(lambda x: (np_ref.correlation(x, f1, **attrs) * top).sum(), f0)
    n1 = np_ref.numeric_grad(lambda x: (np_ref.correlation(f0, x, **attrs) * top).sum(), f1)
    np.testing.assert_allclose(g0, n0, **TOL)
    np.testing.assert_allclose(g1, n1, **TOL)


def test_correlation_grad_flownetc_settings(oracle_mod):
    rng = np.random.default_rng(1)
    f0 = rng.standard_normal((1, 3, 5, 6)).astype(np.float32)
    f1 = rng.standard_normal((1, 3, 5, 6)).astype(np.float32)
    out, p0, p1 = oracle_mod.correlation(f0, f1)
    top = rng.standard_normal(out.shape).astype(np.float32)
    g0, g1 = oracle_mod.correlation_grad(top, f0, f1, p0, p1)
    n0 = np_ref.numeric_grad(lambda x: (np_ref.correlation(x, f1) * top).sum(), f0)
    n1 = np_ref.numeric_grad(lambda x: (np_ref.correlation(f0, x) * top).sum(), f1)
    np.testing.assert_allclose(g0, n0, **TOL)
    np.testing.assert_allclose(g1, n1, **TOL)


def test_backward_warp_grad(oracle_mod):
    rng = np.random.default_rng(2)
    im = rng.uniform(size=(2, 6, 7, 3)).astype(np.float32)
    flow = away_from_integers(rng, (2, 6, 7, 2), 2.0)
    g = rng.standard_normal(im.shape).astype(np.float32)
    got = oracle_mod.backward_warp_grad(g, im, flow)
    num = np_ref.numeric_grad(lambda f: (np_ref.backward_warp(im, f.astype(np.float32)) * g).sum(), flow, 1e-3)
    np.testing.assert_allclose(got, num, rtol=5e-3, atol=5e-3)  # float32 flow perturbation


def test_image_warp_grads(oracle_mod):
    rng = np.random.default_rng(3)
    im = rng.uniform(size=(2, 6, 7, 3)).astype(np.float32)
    flow = away_from_integers(rng, (2, 6, 7, 2), 2.0)
    g = rng.standard_normal(im.shape).astype(np.float32)
    d_im, d_flow = oracle_mod.image_warp_grad(g, im, flow)
    n_im = np_ref.numeric_grad(lambda x: (np_ref.image_warp(x, flow) * g).sum(), im)
    np.testing.assert_allclose(d_im, n_im, **TOL)
    n_fl = np_ref.numeric_grad(lambda f: (np_ref.image_warp(im, f.astype(np.float32)) * g).sum(), flow, 1e-3)
    np.testing.assert_allclose(d_flow, n_fl, rtol=5e-3, atol=5e-3)
    # NULL outputs are honoured
    only_flow = oracle_mod.image_warp_grad(g, im, flow, want_im=False)
    assert only_flow[0] is None
    np.testing.assert_array_equal(only_flow[1], d_flow)


def test_forward_warp_grad(kat, oracle_mod):
    shape = kat["forward_warp_gradcheck"]["flow_shape"]
    rng = np.random.default_rng(4)
    flow = away_from_integers(rng, shape, 1.5)
    g = rng.standard_normal(shape[:3] + [1]).astype(np.float32)
    got = oracle_mod.forward_warp_grad(g, flow)
    num = np_ref.numeric_grad(lambda f: (np_ref.forward_warp(f.astype(np.float32)) * g).sum(), flow, 1e-3)
    np.testing.assert_allclose(got, num, rtol=5e-3, atol=5e-3)


@pytest.mark.parametrize("r", [1, 2, 3])
def test_ternary_grad(oracle_mod, r):
    rng = np.random.default_rng(5)
    im1 = rng.uniform(size=(1, 9, 10, 3)).astype(np.float32)
    im2 = rng.uniform(size=(1, 9, 10, 3)).astype(np.float32)
    mask = (rng.uniform(size=(1, 9, 10, 1)) > 0.2).astype(np.float32)
    got = oracle_mod.ternary_loss_grad(im1, im2, mask, r)
    num = np_ref.numeric_grad(lambda x: np_ref.ternary_loss(im1, x, mask, r), im2, 1e-5)
    np.testing.assert_allclose(got, num, rtol=1e-3, atol=1e-6)


def test_charbonnier_grad(oracle_mod):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((2, 4, 5, 3)).astype(np.float32)
    for mask in (None, (rng.uniform(size=(2, 4, 5, 1)) > 0.3).astype(np.float32),
                 (rng.uniform(size=(2, 4, 5, 3)) > 0.3).astype(np.float32)):
        for beta in (1.0, 255.0):
            loss = oracle_mod.charbonnier_loss(x, mask, beta=beta)
            np.testing.assert_allclose(loss, np_ref.charbonnier(x, mask, beta=beta), rtol=1e-5)
            got = oracle_mod.charbonnier_loss_grad(x, mask, gout=1.0, beta=beta)
            num = np_ref.numeric_grad(lambda v: np_ref.charbonnier(v, mask, beta=beta), x, 1e-6)
            np.testing.assert_allclose(got, num, rtol=2e-3, atol=1e-5)


@pytest.mark.parametrize("order", [1, 2])
def test_smoothness_grad(oracle_mod, order):
    """smoothness_loss / second_order_loss gradient vs float64 central differences."""
    rng = np.random.default_rng(12 + order)
    flow = (rng.standard_normal((2, 6, 7, 2)) * 3).astype(np.float32)
    g = oracle_mod.smoothness_loss_grad(flow, order)
    num = np_ref.numeric_grad(lambda x: np_ref.smoothness_loss(x, order), flow, eps=1e-6)
    np.testing.assert_allclose(g, num, rtol=1e-3, atol=1e-6)


def test_gradient_loss_grad(oracle_mod):
    rng = np.random.default_rng(14)
    a = rng.uniform(size=(2, 6, 7, 3)).astype(np.float32)
    b = rng.uniform(size=(2, 6, 7, 3)).astype(np.float32)
    mask = (rng.uniform(size=(2, 6, 7, 1)) > 0.2).astype(np.float32)
    g = oracle_mod.gradient_loss_grad(a, b, mask)
    num = np_ref.numeric_grad(lambda x: np_ref.gradient_loss(a, x, mask), b, eps=1e-6)
    np.testing.assert_allclose(g, num, rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("mask_occlusion", ["fb", ""])
def test_compute_losses_grad_composition(oracle_mod, mask_occlusion):
    """oracle.compute_losses_grad (the composition TF's autodiff makes of losses.py:16-87) against
    central differences of the float64 restatement of the same weighted sum, the reference's
    default KITTI mix plus photo and grad (config_template/config.ini:123,166-174).  The 0/1 masks
    are piecewise constant in the flows (tf.cast of a comparison): held fixed on the numeric side."""
    rng = np.random.default_rng(40)
    B, H, W = 1, 9, 10
    im1 = rng.uniform(size=(B, H, W, 3)).astype(np.float32)
    im2 = rng.uniform(size=(B, H, W, 3)).astype(np.float32)
    ffw = away_from_integers(rng, (B, H, W, 2), 1.0)
    fbw = away_from_integers(rng, (B, H, W, 2), 1.0)
    border = oracle_mod.create_border_mask((B, H, W)) if mask_occlusion else None
    wts = dict(ternary=1.0, smooth_2nd=3.0, fb=0.2, occ=12.4, photo=0.7, grad=0.5, smooth_1st=0.3, sym=0.1)
    r = 2
    d_fw, d_bw, losses = oracle_mod.compute_losses_grad(im1, im2, ffw, fbw, border, mask_occlusion, r, wts)
    _, aux = oracle_mod.compute_losses(im1, im2, ffw, fbw, border, mask_occlusion, r)
    mfw, mbw = aux["mask_fw"], aux["mask_bw"]
    assert 0 < mfw.sum() < mfw.size or not mask_occlusion

    def warped_terms(f_fw, f_bw):
        f_fw32, f_bw32 = f_fw.astype(np.float32), f_bw.astype(np.float32)
        i2w, i1w = np_ref.image_warp(im2, f_fw32), np_ref.image_warp(im1, f_bw32)
        s = wts["ternary"] * (np_ref.ternary_loss(im1, i2w, mfw, r) + np_ref.ternary_loss(im2, i1w, mbw, r))
        s += wts["photo"] * (np_ref.charbonnier(im1 - i2w, mfw, beta=255) + np_ref.charbonnier(im2 - i1w, mbw, beta=255))
        s += wts["grad"] * (np_ref.gradient_loss(im1, i2w, mfw) + np_ref.gradient_loss(im2, i1w, mbw))
        s += wts["fb"] * (np_ref.charbonnier(f_fw + np_ref.image_warp(f_bw, f_fw32), mfw) +
                          np_ref.charbonnier(f_bw + np_ref.image_warp(f_fw, f_bw32), mbw))
        return s                       # occ and sym: functions of the masks alone

    def smooth_terms(f_fw, f_bw):
        return (wts["smooth_1st"] * (np_ref.smoothness_loss(f_fw, 1) + np_ref.smoothness_loss(f_bw, 1)) +
                wts["smooth_2nd"] * (np_ref.smoothness_loss(f_fw, 2) + np_ref.smoothness_loss(f_bw, 2)))

    # two step sizes: the warp casts the flow to float32 (1e-3, as test_image_warp_grads); the
    # charbonnier of a flow difference bends at |delta| ~ epsilon = 1e-3 (1e-6, as test_smoothness_grad)
    f64 = np.float64
    n_fw = (np_ref.numeric_grad(lambda x: warped_terms(x, fbw.astype(f64)), ffw, 1e-3) +
            np_ref.numeric_grad(lambda x: smooth_terms(x, fbw.astype(f64)), ffw, 1e-6))
    n_bw = (np_ref.numeric_grad(lambda x: warped_terms(ffw.astype(f64), x), fbw, 1e-3) +
            np_ref.numeric_grad(lambda x: smooth_terms(ffw.astype(f64), x), fbw, 1e-6))
    # float32 flow perturbation inside the warp: the reference's gradient-check tolerance scaled
    # to the gradient's size (test/ops/backward_warp.py:21-25 uses 1e-3 on O(1) gradients)
    for got, num in ((d_fw, n_fw), (d_bw, n_bw)):
        np.testing.assert_allclose(got, num, rtol=5e-3, atol=5e-3 * np.abs(num).max())
    # the forward scalars it returns are compute_losses' own
    assert set(losses) == {"sym", "occ", "photo", "grad", "smooth_1st", "smooth_2nd", "fb", "ternary"}
