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
