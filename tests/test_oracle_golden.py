"""The CPU oracle against every known-answer vector the reference's own tests
hold for this path (tests/golden/reference_kat.json), against the independent
numpy restatement, and against analytic spot values."""
import numpy as np
import pytest

from oracle import np_ref


def A(x):
    return np.asarray(x, np.float32)


@pytest.mark.parametrize("name", ["warp_move", "backward_warp_batches", "image_warp_batches",
                                  "warp_interpolate"])
def test_warp_known_answers(kat, oracle_mod, name):
    case = kat[name]
    for op in case["applies_to"]:
        got = getattr(oracle_mod, op)(A(case["image"]), A(case["flow"]))
        # assertAllClose defaults of tf.test.TestCase: rtol=atol=1e-6
        np.testing.assert_allclose(got, A(case["expected"]), rtol=1e-6, atol=1e-6, err_msg=op)


@pytest.mark.parametrize("name", ["correlation_trivial", "correlation_batch"])
def test_correlation_known_answers(kat, oracle_mod, name):
    case = kat[name]
    out, p0, p1 = oracle_mod.correlation(A(case["first"]), A(case["second"]), **case["attrs"])
    np.testing.assert_allclose(out, A(case["expected"]), rtol=1e-6, atol=1e-6)
    assert p0.shape == (out.shape[0], 4, 4, 1)


def test_downsample_known_answer(kat, oracle_mod):
    case = kat["downsample"]
    got = oracle_mod.downsample(A(case["image"]), case["scale"])
    np.testing.assert_allclose(got, A(case["expected"]), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", ["outgoing_mask_all_directions", "outgoing_mask_large_movement"])
def test_outgoing_mask_known_answers(kat, oracle_mod, name):
    case = kat[name]
    got = oracle_mod.create_outgoing_mask(A(case["flow"]))
    np.testing.assert_array_equal(got[0, :, :, 0], A(case["expected"]))


def test_correlation_geometry_flownetc(oracle_mod):
    g = oracle_mod.correlation_geometry(48, 160)
    assert g == dict(out_channels=441, out_height=48, out_width=160, padded_height=88, padded_width=200)


def test_correlation_errors(oracle_mod):
    f = np.zeros((1, 1, 4, 4), np.float32)
    with pytest.raises(ValueError, match="kernel_size must be odd"):
        oracle_mod.correlation(f, f, kernel_size=2)
    with pytest.raises(ValueError, match="Input shapes have to be the same"):
        oracle_mod.correlation(f, np.zeros((1, 1, 4, 5), np.float32))
    with pytest.raises(ValueError, match="Invalid correlation settings"):
        oracle_mod.correlation(f, f, max_displacement=20, pad=0)
    with pytest.raises(ValueError, match="divisible by scale"):
        oracle_mod.downsample(np.zeros((1, 5, 4, 1), np.float32), 2)


def test_correlation_self_zero_displacement_is_mean_square(oracle_mod):
    rng = np.random.default_rng(0)
    f = rng.standard_normal((1, 8, 6, 7)).astype(np.float32)
    out, _, _ = oracle_mod.correlation(f, f, max_displacement=4, pad=4, stride_2=2)
    centre = out.shape[1] // 2
    np.testing.assert_allclose(out[:, centre], (f.astype(np.float64) ** 2).mean(1), rtol=1e-6)


@pytest.mark.parametrize("attrs", [
    dict(),  # FlowNetC defaults
    dict(kernel_size=3, max_displacement=1, pad=2, stride_1=1, stride_2=1),  # the disabled 3x3 test's settings
    dict(kernel_size=3, max_displacement=4, pad=3, stride_1=2, stride_2=2),
    dict(kernel_size=1, max_displacement=5, pad=2, stride_1=1, stride_2=2),  # pad < max_displacement
    dict(kernel_size=5, max_displacement=3, pad=6, stride_1=3, stride_2=1),
])
def test_correlation_vs_closed_form(oracle_mod, attrs):
    rng = np.random.default_rng(1)
    f0 = rng.standard_normal((2, 6, 9, 11)).astype(np.float32)
    f1 = rng.standard_normal((2, 6, 9, 11)).astype(np.float32)
    out, p0, p1 = oracle_mod.correlation(f0, f1, **attrs)
    np.testing.assert_allclose(out, np_ref.correlation(f0, f1, **attrs), rtol=1e-5, atol=1e-6)
    top = rng.standard_normal(out.shape).astype(np.float32)
    g0, g1 = oracle_mod.correlation_grad(top, f0, f1, p0, p1, **attrs)
    r0, r1 = np_ref.correlation_grad(top, f0, f1, **attrs)
    np.testing.assert_allclose(g0, r0, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(g1, r1, rtol=1e-4, atol=1e-5)


def test_forward_warp_analytic_values(oracle_mod):
    # zero flow: interior = (sum_{d=-4..4} exp(-d^2/2))^2, corner = (sum_{d=0..4})^2
    out = oracle_mod.forward_warp(np.zeros((1, 12, 12, 2), np.float32))
    s_full = sum(np.exp(-d * d / 2.0) for d in range(-4, 5))
    s_half = sum(np.exp(-d * d / 2.0) for d in range(0, 5))
    np.testing.assert_allclose(out[0, 6, 6, 0], s_full ** 2, rtol=1e-6)
    np.testing.assert_allclose(out[0, 0, 0, 0], s_half ** 2, rtol=1e-6)
    assert abs(s_full ** 2 - 6.2831479) < 1e-5 and abs(s_half ** 2 - 3.0740974) < 1e-5


def test_forward_warp_vs_gather_form(oracle_mod):
    rng = np.random.default_rng(2)
    flow = (rng.standard_normal((2, 9, 11, 2)) * 4).astype(np.float32)
    flow[0, 0, 0] = [-30, 2]     # window entirely outside
    flow[0, 1, 1] = [-5.5, -5.2]  # window partially outside
    np.testing.assert_allclose(oracle_mod.forward_warp(flow), np_ref.forward_warp(flow), rtol=1e-5, atol=1e-6)


def test_image_warp_vs_backward_warp_semantics(oracle_mod):
    """D2: clamp-to-edge vs zero-outside differ exactly where taps leave the image."""
    im = np.ones((1, 4, 4, 1), np.float32)
    flow = np.zeros((1, 4, 4, 2), np.float32)
    flow[0, 0, 0] = [-2.5, 0]
    assert oracle_mod.image_warp(im, flow)[0, 0, 0, 0] == 1.0
    assert oracle_mod.backward_warp(im, flow)[0, 0, 0, 0] == 0.0


def test_ternary_matches_numpy(oracle_mod):
    rng = np.random.default_rng(3)
    im1 = rng.uniform(size=(2, 10, 12, 3)).astype(np.float32)
    im2 = rng.uniform(size=(2, 10, 12, 3)).astype(np.float32)
    mask = (rng.uniform(size=(2, 10, 12, 1)) > 0.2).astype(np.float32)
    for r in (1, 2, 3):
        loss, dist = oracle_mod.ternary_loss(im1, im2, mask, r, return_dist=True)
        np.testing.assert_allclose(dist[..., 0], np_ref.ternary_dist(im1, im2, r), rtol=2e-5, atol=1e-5)
        np.testing.assert_allclose(loss, np_ref.ternary_loss(im1, im2, mask, r), rtol=1e-5)
    # identical images: every t1 == t2, distance 0, loss = eps^(2*alpha) * mean(mask')
    loss0 = oracle_mod.ternary_loss(im1, im1, np.ones_like(mask), 1)
    inner = (10 - 2) * (12 - 2) / (10 * 12)
    np.testing.assert_allclose(loss0, (0.001 ** 2) ** 0.45 * inner, rtol=1e-5)


def test_masks(oracle_mod):
    m = oracle_mod.create_mask((2, 5, 7), [[1, 2], [0, 3]])
    assert m.shape == (2, 5, 7, 1)
    assert m[0, :, :, 0].sum() == (5 - 3) * (7 - 3) and m[0, 1, 0, 0] == 1 and m[0, 0, 0, 0] == 0
    b = oracle_mod.create_border_mask((1, 384, 1280), 0.1)  # ceil(38.4)=39
    assert b[0, 38, 39, 0] == 0 and b[0, 39, 39, 0] == 1 and b[0, 384 - 40, 1280 - 40, 0] == 1


def test_fb_occlusion(oracle_mod):
    f = np.zeros((1, 2, 2, 2), np.float32)
    w = np.zeros((1, 2, 2, 2), np.float32)
    f[0, 0, 0] = [3, 0]
    w[0, 0, 0] = [-3, 0]       # consistent: diff 0
    f[0, 0, 1] = [3, 0]
    w[0, 0, 1] = [0, 0]        # |diff|^2 = 9 > 0.01*9+0.5
    diff, occ = oracle_mod.fb_occlusion(f, w)
    assert occ[0, 0, 0, 0] == 0 and occ[0, 0, 1, 0] == 1 and occ[0, 1, 1, 0] == 0
    np.testing.assert_array_equal(diff, f + w)


def test_smoothness_deltas_known_answer(kat, oracle_mod):
    """test/test_losses.py:11-41 of the reference, values and masks exact."""
    k = kat["smoothness_deltas"]
    du, dv, m = oracle_mod.smoothness_deltas(np.asarray(k["flow"], np.float32), 1)
    np.testing.assert_array_equal(m[0, :, :, 0], k["mask0"])
    np.testing.assert_array_equal(m[0, :, :, 1], k["mask1"])
    np.testing.assert_array_equal((du * m)[0, :, :, 0], k["delta_u0"])
    np.testing.assert_array_equal((du * m)[0, :, :, 1], k["delta_u1"])
    np.testing.assert_array_equal((dv * m)[0, :, :, 0], k["delta_v0"])
    np.testing.assert_array_equal((dv * m)[0, :, :, 1], k["delta_v1"])


def test_gradient_loss_known_answer(kat, oracle_mod):
    """test/test_losses.py:97-121: an image against itself + 1 has (almost) no gradient loss."""
    k = kat["gradient_loss"]
    loss = oracle_mod.gradient_loss(k["im1"], k["im2"], np.ones((1, 3, 3, 1), np.float32))
    assert abs(loss - k["expected"]) <= k["atol"]
    # exactly: 3 colours x (column 1 for d/dx + row 1 for d/dy) = 18 unmasked entries, all with diff 0
    np.testing.assert_allclose(loss, (0.001 ** 2) ** 0.45 * 18 / 54, rtol=1e-5)


def test_stencil_terms_match_numpy(oracle_mod):
    rng = np.random.default_rng(8)
    flow = (rng.standard_normal((2, 9, 11, 2)) * 3).astype(np.float32)
    for order in (1, 2):
        np.testing.assert_allclose(oracle_mod.smoothness_loss(flow, order), np_ref.smoothness_loss(flow, order),
                                   rtol=1e-6)
    im1 = rng.uniform(size=(2, 9, 11, 3)).astype(np.float32)
    im2 = rng.uniform(size=(2, 9, 11, 3)).astype(np.float32)
    mask = (rng.uniform(size=(2, 9, 11, 1)) > 0.2).astype(np.float32)
    np.testing.assert_allclose(oracle_mod.gradient_loss(im1, im2, mask), np_ref.gradient_loss(im1, im2, mask),
                               rtol=1e-6)
    # second-order deltas of an affine flow vanish inside the masks
    yy, xx = np.meshgrid(np.arange(9.0), np.arange(11.0), indexing="ij")
    aff = np.stack([2 * xx - yy + 1, 0.5 * xx + 3 * yy], -1)[None].astype(np.float32)
    du, dv, m = oracle_mod.smoothness_deltas(aff, 2)
    assert np.abs(du * m).max() == 0 and np.abs(dv * m).max() == 0
    assert m[0, :, :, 2].sum() == 7 * 9 and m[0, :, :, 0].sum() == 9 * 9 and m[0, :, :, 1].sum() == 7 * 11


def test_resize_bilinear_and_stacked_inputs(oracle_mod):
    """tf.image.resize_bilinear (TF 1.x legacy semantics, assumption A2) restated; flownet.py:46-57."""
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 5, 7, 2)).astype(np.float32)
    np.testing.assert_array_equal(oracle_mod.resize_bilinear(x, 5, 7), x)            # identity at scale 1
    up = oracle_mod.resize_bilinear(x, 20, 28)
    np.testing.assert_array_equal(up[:, ::4, ::4], x)                                 # integer positions hit the samples
    np.testing.assert_allclose(up[:, 2, 0], 0.5 * (x[:, 0, 0] + x[:, 1, 0]), rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(up[:, 17:, :], np.broadcast_to(up[:, 16:17, :], up[:, 17:, :].shape))  # clamped last row
    const = np.full((1, 3, 4, 2), 1.25, np.float32)
    np.testing.assert_array_equal(oracle_mod.resize_bilinear(const, 12, 16, mult=20.0), np.full((1, 12, 16, 2), 25.0))
    # adjoint identity of the gradient
    g = rng.standard_normal(up.shape).astype(np.float32)
    gi = oracle_mod.resize_bilinear_grad(g, 5, 7)
    np.testing.assert_allclose((up.astype(np.float64) * g).sum(), (x.astype(np.float64) * gi).sum(), rtol=1e-5)
    im1 = rng.uniform(size=(2, 20, 28, 3)).astype(np.float32)
    im2 = rng.uniform(size=(2, 20, 28, 3)).astype(np.float32)
    out = oracle_mod.stacked_inputs(im1, im2, x * 0.1)
    flow = oracle_mod.resize_bilinear(x * 0.1, 20, 28, mult=20.0)
    warp = oracle_mod.image_warp(im2, flow)
    np.testing.assert_array_equal(out, np.concatenate([im1, im2, flow, warp, np.abs(warp - im1)], 3))


def test_integer_census_definition(oracle_mod):
    """Integer ternary census (extension, not in the reference): hand-checked bits."""
    im = np.zeros((1, 3, 3, 3), np.float32)
    im[0, 1, 1] = 0.5                      # centre: gray 127.4...
    im[0, 0, 0] = 1.0                      # neighbour 0 brighter
    im[0, 1, 2] = 0.5                      # neighbour 4 equal
    sig = oracle_mod.census_transform(im, 1, eps=0.9)
    assert sig[0, 1, 1, 0] == 0b00000001                  # only neighbour 0 is brighter
    assert sig[0, 1, 1, 1] == 0b11101110                  # all others darker except the equal one (bit 4) and bit 0
    # corner pixel (0,0): its three in-image neighbours (bits 4, 6, 7) and the zero padding (bits 0-3, 5) are all darker
    assert sig[0, 0, 0, 0] == 0 and sig[0, 0, 0, 1] == 0xFF
    d, total = oracle_mod.census_hamming(sig, sig)
    assert total == 0 and not d.any()
    rng = np.random.default_rng(4)
    a = rng.integers(0, 4, size=(2, 9, 11, 3)).astype(np.float32) / 3.0     # many exact ties
    b = rng.integers(0, 4, size=(2, 9, 11, 3)).astype(np.float32) / 3.0
    sa, sb = oracle_mod.census_transform(a, 3, 0.0), oracle_mod.census_transform(b, 3, 0.0)
    assert not (sa[..., 0] & sa[..., 1]).any()                              # a neighbour is never both
    d, total = oracle_mod.census_hamming(sa, sb)
    x = (sa[..., 0] ^ sb[..., 0]) | (sa[..., 1] ^ sb[..., 1])
    ref = np.array([bin(int(v)).count("1") for v in x.ravel()]).reshape(x.shape)
    np.testing.assert_array_equal(d, ref)
    assert total == ref.sum() and d.max() <= 48


def test_seeded_fixture_is_reproducible(seeded, oracle_mod):
    """The committed oracle outputs are what the oracle computes today."""
    c, p0, p1 = oracle_mod.correlation(seeded["corr_f0"], seeded["corr_f1"])
    np.testing.assert_array_equal(c, seeded["corr_out"])
    np.testing.assert_array_equal(oracle_mod.image_warp(seeded["im2"], seeded["flow"]), seeded["image_warp"])
    np.testing.assert_allclose(oracle_mod.forward_warp(seeded["flow"]), seeded["forward_warp"], rtol=1e-6)
    np.testing.assert_allclose(oracle_mod.smoothness_loss(seeded["flow"], 2), float(seeded["smooth_loss_o2"]), rtol=1e-7)
    np.testing.assert_array_equal(oracle_mod.gradient_loss_grad(seeded["im1"], seeded["image_warp"], seeded["border_mask"]),
                                  seeded["gradient_loss_grad"])
    np.testing.assert_array_equal(oracle_mod.census_transform(seeded["im1"], 3, 0.9), seeded["census_sig1"])
    np.testing.assert_array_equal(oracle_mod.stacked_inputs(seeded["im1"], seeded["im2"], seeded["stacked_flow_lr"]),
                                  seeded["stacked_inputs"])


# ----------------------------------------------------------------------------- resize_area (util.downsample, odd extents)
@pytest.mark.parametrize("H,W,oh,ow", [(5, 18, 2, 9), (7, 9, 3, 4), (12, 40, 3, 10), (3, 3, 1, 1), (9, 31, 4, 15), (5, 5, 5, 5)])
def test_resize_area_vs_overlap_integral(oracle_mod, H, W, oh, ow):
    """PARITY UNPINNED (tf.image.resize_area is third-party, assumption A3): the C restatement of
    TF's loop against an independent float64 overlap integral; on divisible extents it is the
    Downsample op's box mean (ops/downsample_op.cu.cc:38-46); a constant image stays constant."""
    rng = np.random.default_rng(H * 100 + W)
    x = rng.uniform(size=(2, H, W, 3)).astype(np.float32)
    got = oracle_mod.resize_area(x, oh, ow)
    assert got.shape == (2, oh, ow, 3)
    np.testing.assert_allclose(got, np_ref.resize_area(x, oh, ow), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(oracle_mod.resize_area(np.full((1, H, W, 1), 0.75, np.float32), oh, ow), 0.75, rtol=1e-6)
    if H % oh == 0 and W % ow == 0 and H // oh == W // ow:
        np.testing.assert_allclose(got, oracle_mod.downsample(x, H // oh), rtol=1e-6, atol=1e-7)


def test_util_downsample_dispatch(oracle_mod):
    """core/util.py:23-28: even extents -> the op (which raises when they do not divide by num),
    an odd extent -> resize_area to (int(H/num), int(W/num))."""
    rng = np.random.default_rng(8)
    x = rng.uniform(size=(1, 8, 12, 2)).astype(np.float32)
    np.testing.assert_array_equal(oracle_mod.util_downsample(x, 2), oracle_mod.downsample(x, 2))
    with pytest.raises(ValueError, match="divisible by scale"):
        oracle_mod.util_downsample(np.zeros((1, 6, 8, 1), np.float32), 4)
    y = rng.uniform(size=(1, 5, 18, 2)).astype(np.float32)
    np.testing.assert_array_equal(oracle_mod.util_downsample(y, 2), oracle_mod.resize_area(y, 2, 9))
    assert oracle_mod.util_downsample(rng.uniform(size=(1, 9, 10, 1)).astype(np.float32), 4).shape == (1, 2, 2, 1)
