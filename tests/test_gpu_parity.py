"""Parity of the HIP path (through the C ABI of libunflow_hip.so) with the CPU
oracle: the reference's known-answer vectors, the committed seeded fixtures and
fresh seeded inputs.  Tolerances: 1e-5 (north_star) on warp/correlation
tensors; census 1e-5 relative on dist_sum and the loss."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def A(x):
    return np.asarray(x, np.float32)


def close(got, want, rtol=1e-5, atol=1e-5, msg=""):
    np.testing.assert_allclose(N(got) if hasattr(got, "detach") else got, want, rtol=rtol, atol=atol, err_msg=msg)


# ----------------------------------------------------------------------------- known answers
@pytest.mark.parametrize("name", ["warp_move", "backward_warp_batches", "image_warp_batches", "warp_interpolate"])
def test_warp_known_answers(kat, dev, name):
    from unflow_amd import ops
    from unflow_amd.image_warp import image_warp
    case = kat[name]
    fns = dict(backward_warp=ops.backward_warp, image_warp=image_warp)
    for op in case["applies_to"]:
        got = fns[op](T(case["image"], dev), T(case["flow"], dev))
        close(got, A(case["expected"]), 1e-6, 1e-6, op)


@pytest.mark.parametrize("name", ["correlation_trivial", "correlation_batch"])
def test_correlation_known_answers(kat, dev, name):
    from unflow_amd import ops
    case = kat[name]
    out = ops.correlation(T(case["first"], dev), T(case["second"], dev), **case["attrs"])
    close(out, A(case["expected"]), 1e-6, 1e-6)
    out3 = ops.correlation_with_padded(T(case["first"], dev), T(case["second"], dev), **case["attrs"])
    close(out3[1], np.transpose(A(case["first"]), (0, 2, 3, 1)), 0, 0)


def test_downsample_known_answer(kat, dev):
    from unflow_amd import ops
    case = kat["downsample"]
    close(ops.downsample(T(case["image"], dev), case["scale"]), A(case["expected"]), 1e-6, 1e-6)


@pytest.mark.parametrize("name", ["outgoing_mask_all_directions", "outgoing_mask_large_movement"])
def test_outgoing_mask_known_answers(kat, dev, name):
    from unflow_amd import losses
    case = kat[name]
    got = N(losses.create_outgoing_mask(T(case["flow"], dev)))
    np.testing.assert_array_equal(got[0, :, :, 0], A(case["expected"]))


# ----------------------------------------------------------------------------- seeded fixtures
def test_seeded_correlation(seeded, dev):
    from unflow_amd import ops
    f0, f1 = T(seeded["corr_f0"], dev), T(seeded["corr_f1"], dev)
    close(ops.correlation(f0, f1), seeded["corr_out"])
    g0, g1 = ops.correlation_grad(T(seeded["corr_top"], dev), f0, f1)
    close(g0, seeded["corr_g0"])
    close(g1, seeded["corr_g1"])
    attrs = dict(zip(("kernel_size", "max_displacement", "pad", "stride_1", "stride_2"),
                     [int(v) for v in seeded["corrg_attrs"]]))
    a0, a1 = T(seeded["corrg_f0"], dev), T(seeded["corrg_f1"], dev)
    close(ops.correlation(a0, a1, **attrs), seeded["corrg_out"])
    h0, h1 = ops.correlation_grad(T(seeded["corrg_top"], dev), a0, a1, **attrs)
    close(h0, seeded["corrg_g0"])
    close(h1, seeded["corrg_g1"])


def test_seeded_warps(seeded, dev):
    from unflow_amd import ops
    from unflow_amd.image_warp import image_warp, image_warp_grad
    im2, flow, grad = T(seeded["im2"], dev), T(seeded["flow"], dev), T(seeded["grad"], dev)
    close(ops.backward_warp(im2, flow), seeded["backward_warp"])
    close(ops.backward_warp_grad(grad, im2, flow), seeded["backward_warp_grad"], 1e-5, 1e-5)
    close(image_warp(im2, flow), seeded["image_warp"])
    d_im, d_flow = image_warp_grad(grad, im2, flow)
    close(d_im, seeded["image_warp_dim"], 1e-5, 1e-5)
    close(d_flow, seeded["image_warp_dflow"], 1e-5, 1e-5)
    close(ops.forward_warp(flow), seeded["forward_warp"], 1e-5, 1e-5)
    close(ops.forward_warp_grad(T(seeded["forward_warp_gin"], dev), flow), seeded["forward_warp_grad"], 1e-5, 1e-5)
    close(ops.downsample(T(seeded["im1"], dev), 2), seeded["downsample2"], 1e-6, 1e-6)
    close(ops.downsample(T(seeded["im1"], dev), 4), seeded["downsample4"], 1e-6, 1e-6)


@pytest.mark.parametrize("r", [1, 2, 3])
def test_seeded_census(seeded, dev, r):
    from unflow_amd import losses
    im1, im2w = T(seeded["im1"], dev), T(seeded["image_warp"], dev).requires_grad_(True)
    mask = T(seeded["border_mask"], dev)
    assert np.abs(N(losses.ternary_dist(im1, im2w, r)) - seeded["ternary_dist_r%d" % r]).max() <= 1e-5 * (2 * r + 1) ** 2
    loss = losses.ternary_loss(im1, im2w, mask, max_distance=r)
    np.testing.assert_allclose(float(loss.detach()), float(seeded["ternary_loss_r%d" % r]), rtol=1e-5)
    loss.backward()
    want = seeded["ternary_grad_r%d" % r]
    close(im2w.grad, want, 1e-5, 1e-5 * np.abs(want).max())  # 1e-5 of the gradient's scale (the reference's own checks use 1e-3)


def test_smoothness_deltas_known_answer(kat, dev):
    """test/test_losses.py:11-41 of the reference through the HIP path, exact."""
    from unflow_amd import losses
    k = kat["smoothness_deltas"]
    du, dv, m = (N(t) for t in losses._smoothness_deltas(T(k["flow"], dev)))
    np.testing.assert_array_equal(m[0, :, :, 0], k["mask0"])
    np.testing.assert_array_equal(m[0, :, :, 1], k["mask1"])
    for name, d in (("delta_u", du), ("delta_v", dv)):
        np.testing.assert_array_equal((d * m)[0, :, :, 0], k[name + "0"])
        np.testing.assert_array_equal((d * m)[0, :, :, 1], k[name + "1"])


def test_gradient_loss_known_answer(kat, dev):
    """test/test_losses.py:97-121."""
    from unflow_amd import losses
    k = kat["gradient_loss"]
    loss = float(losses.gradient_loss(T(k["im1"], dev), T(k["im2"], dev), torch.ones((1, 3, 3, 1), device=dev)))
    assert abs(loss - k["expected"]) <= k["atol"]
    np.testing.assert_allclose(loss, (0.001 ** 2) ** 0.45 * 18 / 54, rtol=1e-5)


def test_seeded_stencil_terms(seeded, dev):
    from unflow_amd import losses
    for order, fn in ((1, losses.smoothness_loss), (2, losses.second_order_loss)):
        flow = T(seeded["flow"], dev).requires_grad_(True)
        loss = fn(flow)
        np.testing.assert_allclose(float(loss.detach()), float(seeded["smooth_loss_o%d" % order]), rtol=1e-5)
        loss.backward()
        want = seeded["smooth_grad_o%d" % order]
        close(flow.grad, want, 1e-5, 1e-5 * np.abs(want).max())
    im1, im2w = T(seeded["im1"], dev), T(seeded["image_warp"], dev).requires_grad_(True)
    loss = losses.gradient_loss(im1, im2w, T(seeded["border_mask"], dev))
    np.testing.assert_allclose(float(loss.detach()), float(seeded["gradient_loss"]), rtol=1e-5)
    loss.backward()
    want = seeded["gradient_loss_grad"]
    close(im2w.grad, want, 1e-5, 1e-5 * np.abs(want).max())


def test_seeded_extensions(seeded, dev):
    """Committed vectors of the extensions: integer census (bit-exact), stacked-net block, fp16 correlation."""
    from unflow_amd import flownet, losses, ops
    im1, im2w = T(seeded["im1"], dev), T(seeded["image_warp"], dev)
    s1, s2 = losses.census_transform(im1, 3, 0.9), losses.census_transform(im2w, 3, 0.9)
    np.testing.assert_array_equal(N(s1).view(np.uint64), seeded["census_sig1"])
    dist, total = losses.census_hamming(s1, s2, T(seeded["border_mask"], dev))
    np.testing.assert_array_equal(N(dist), seeded["census_dist"])
    assert int(total) == int(seeded["census_total"])
    close(flownet.stacked_inputs(im1, T(seeded["im2"], dev), T(seeded["stacked_flow_lr"], dev)), seeded["stacked_inputs"])
    h0, h1 = torch.from_numpy(seeded["corr16_f0"]).to(dev), torch.from_numpy(seeded["corr16_f1"]).to(dev)
    w = seeded["corr16_out"]
    for fn, identical in ((ops.correlation_fp16, False), (ops.correlation_fp16_exact, True)):
        g = fn(h0, h1).cpu().numpy()
        np.testing.assert_allclose(g.astype(np.float32), w.astype(np.float32), rtol=2e-3, atol=1e-6)
        if identical:       # f32 MFMA on the widened halves: the oracle's products, rounded once
            assert (g == w).mean() > 0.99


# ----------------------------------------------------------------------------- fresh inputs vs oracle
@pytest.mark.parametrize("shape,attrs", [
    ((2, 32, 16, 24), dict()),                                   # FlowNetC attrs
    ((1, 256, 16, 16), dict()),                                  # BASELINE cfg1 feature shape
    ((1, 7, 9, 13), dict()),                                     # ragged extents and channels
    # C % 128 == 0 and W % 4 == 0: the persistent row-sweep backward (corr_mfma_bwd4_kernel) --
    # odd heights (the two row classes differ in length), one-row classes, strips that leave the
    # image, more tiles than the accumulator ring holds, item counts that do not divide by 8
    ((1, 128, 9, 12), dict()),
    ((3, 128, 7, 8), dict()),
    ((1, 128, 2, 4), dict()),
    ((1, 256, 100, 36), dict()),
    ((2, 128, 30, 20), dict()),
    # more items than CUs with two- and three-row classes: a workgroup walks on to its next item
    # while the DMA already runs a row ahead (the row barrier sits inside the row)
    ((130, 128, 4, 8), dict()),
    ((66, 256, 5, 8), dict()),
    # ... and with TALL row classes (14 rows: the row barrier is the arrival counter in LDS on 13 of them, the epilogue
    # image is recycled in-loop) on workgroups that walk two items: 280 items on 256 CUs
    ((14, 128, 28, 160), dict()),
    # C % 128 == 0 but not a power of two: the tile stores divide by C instead of multiplying by 1/C
    ((2, 384, 13, 20), dict()),
    # the forward's ring of four 4-channel stages (corr_fwd2_kernel, C % 16 == 0 and W % 4 == 0): one group of four K-steps
    # (prologue and tail only), three groups, a channel count that is not a power of two, regions that leave the image
    ((1, 16, 8, 8), dict()),
    ((2, 48, 10, 12), dict()),
    ((1, 80, 21, 36), dict()),
    ((2, 6, 10, 12), dict(kernel_size=3, max_displacement=4, pad=3, stride_1=2, stride_2=2)),
    ((1, 5, 9, 11), dict(kernel_size=1, max_displacement=5, pad=2, stride_1=1, stride_2=2)),
    ((1, 4, 12, 9), dict(kernel_size=5, max_displacement=3, pad=6, stride_1=3, stride_2=1)),
    ((1, 8, 6, 7), dict(kernel_size=1, max_displacement=0, pad=0, stride_1=1, stride_2=1)),
])
def test_correlation_vs_oracle(oracle_mod, dev, shape, attrs):
    from unflow_amd import ops
    rng = np.random.default_rng(hash(shape) % 1000)
    f0 = rng.standard_normal(shape).astype(np.float32)
    f1 = rng.standard_normal(shape).astype(np.float32)
    want, p0, p1 = oracle_mod.correlation(f0, f1, **attrs)
    a = T(f0, dev).requires_grad_(True)
    b = T(f1, dev).requires_grad_(True)
    got = ops.correlation(a, b, **attrs)
    close(got, want)
    top = rng.standard_normal(want.shape).astype(np.float32)
    got.backward(T(top, dev))
    g0, g1 = oracle_mod.correlation_grad(top, f0, f1, p0, p1, **attrs)
    close(a.grad, g0)
    close(b.grad, g1)
    o3 = ops.correlation_with_padded(T(f0, dev), T(f1, dev), **attrs)
    close(o3[1], p0, 0, 0)
    close(o3[2], p1, 0, 0)


def test_correlation_adversarial_cancellation(oracle_mod, dev):
    """Large-magnitude cancelling inputs: error is judged relative to sum |a*b| / C."""
    from unflow_amd import ops
    rng = np.random.default_rng(7)
    f0 = (rng.standard_normal((1, 64, 8, 8)) * 1e3).astype(np.float32)
    f1 = f0 * np.where(np.arange(64)[None, :, None, None] % 2 == 0, 1, -1).astype(np.float32)
    want, _, _ = oracle_mod.correlation(f0, f1)
    got = N(ops.correlation(T(f0, dev), T(f1, dev)))
    scale = (np.abs(f0) ** 2).mean()
    assert np.abs(got - want).max() <= 1e-5 * scale


def _between_nans(a, dev):
    """The array as a tensor in the middle of a larger NaN-filled allocation (8-byte aligned): a kernel that
    reads outside its tensor -- a channel tail that relies on a range check, a halo row -- shows up as NaN."""
    t = torch.from_numpy(a)
    pad = 4096
    buf = torch.full((t.numel() + 2 * pad,), float("nan"), dtype=t.dtype, device=dev)
    buf[pad:pad + t.numel()] = t.reshape(-1).to(dev)
    return buf[pad:pad + t.numel()].view(t.shape)


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("shape", [(2, 32, 16, 24), (1, 256, 16, 16), (1, 7, 12, 20), (2, 20, 50, 36), (1, 136, 70, 44),
                                   (1, 16, 2, 4), (1, 3, 33, 8)])
def test_correlation_fp16(oracle_mod, dev, shape, exact):
    """Extension (BASELINE config 5 wording "fp16 with fp32 accumulate"; not in the reference): half
    tensors in HBM, exact f32 products, f32 sums.  ``correlation_fp16`` runs v_mfma_f32_16x16x16_f16,
    ``correlation_fp16_exact`` widens the halves and runs the float32 op's f32 MFMA loop.  Against the
    fp32 oracle on the SAME half-rounded inputs the only differences are the f32 summation order and
    the one rounding of the result to half: SURVEY 8d's rtol 2e-3; the exact form is in addition
    bit-identical to the rounded oracle except on rounding boundaries.
    Shapes: several row blocks / column strips / channel groups with tails, odd class extents."""
    from unflow_amd import ops
    fwd = ops.correlation_fp16_exact if exact else ops.correlation_fp16
    rng = np.random.default_rng(sum(shape))
    f0 = rng.standard_normal(shape).astype(np.float16)
    f1 = rng.standard_normal(shape).astype(np.float16)
    want32, _, _ = oracle_mod.correlation(f0.astype(np.float32), f1.astype(np.float32))
    want = want32.astype(np.float16)
    got = fwd(_between_nans(f0, dev), _between_nans(f1, dev))
    assert got.dtype == torch.float16 and tuple(got.shape) == want.shape
    g = got.cpu().numpy()
    np.testing.assert_allclose(g.astype(np.float32), want.astype(np.float32), rtol=2e-3, atol=1e-6)
    # measured: the matrix-core form also lands on the oracle's half almost everywhere (f32 sums of
    # exact products in another order); only the exact form is REQUIRED to
    assert (g == want).mean() > (0.99 if exact else 0.97)
    if exact:   # and against the fp32 kernel on the widened inputs, rounded the same way
        ref = ops.correlation(torch.from_numpy(f0.astype(np.float32)).to(dev), torch.from_numpy(f1.astype(np.float32)).to(dev))
        assert (got == ref.half()).float().mean() > 0.99
    # gradients: oracle on the half-rounded top / inputs, rounded to half
    top = rng.standard_normal(want.shape).astype(np.float16)
    _, p0, p1 = oracle_mod.correlation(f0.astype(np.float32), f1.astype(np.float32))
    w0, w1 = oracle_mod.correlation_grad(top.astype(np.float32), f0.astype(np.float32), f1.astype(np.float32), p0, p1)
    a = _between_nans(f0, dev).requires_grad_(True)
    b = _between_nans(f1, dev).requires_grad_(True)
    fwd(a, b).backward(_between_nans(top, dev))
    for gg, ww in ((a.grad, w0), (b.grad, w1)):
        assert gg.dtype == torch.float16
        gnp, wh = gg.cpu().numpy(), ww.astype(np.float16)
        np.testing.assert_allclose(gnp.astype(np.float32), wh.astype(np.float32), rtol=2e-3, atol=2e-4 * np.abs(ww).max())
        assert (gnp == wh).mean() > (0.98 if exact else 0.95)
    with pytest.raises(ValueError, match="FlowNetC"):
        fwd(torch.from_numpy(f0).to(dev), torch.from_numpy(f1).to(dev), max_displacement=4, pad=4)
    with pytest.raises(TypeError):
        fwd(torch.from_numpy(f0.astype(np.float32)).to(dev), torch.from_numpy(f1).to(dev))


def test_correlation_fp16_known_answers(dev):
    """Structure of the f16 matrix-core kernels on integer data, where every sum is exact: a one-hot
    feature map shifted by a displacement puts 1/C on exactly that displacement plane, and the
    gradients of a one-hot upstream gradient are the other input shifted back."""
    from unflow_amd import ops
    B, C, H, W = 1, 16, 40, 48
    f0 = torch.zeros((B, C, H, W), device=dev, dtype=torch.float16)
    f1 = torch.zeros_like(f0)
    f0[0, 3, 17, 22] = 2.0
    f1[0, 3, 17 + 2 * 4, 22 - 2 * 7] = 3.0          # displacement (dy', dx') = (10 + 4, 10 - 7)
    out = ops.correlation_fp16(f0, f1)
    want = torch.zeros_like(out)
    want[0, (10 + 4) * 21 + (10 - 7), 17, 22] = 6.0 / C
    assert torch.equal(out, want)
    top = torch.zeros_like(out)
    top[0, (10 - 2) * 21 + (10 + 5), 20, 30] = 4.0  # d = (-2, +5) at position (20, 30)
    a, b = torch.rand_like(f0).round_().requires_grad_(True), torch.rand_like(f0).mul_(4).round_().requires_grad_(True)
    ops.correlation_fp16(a, b).backward(top)
    g0, g1 = torch.zeros_like(f0), torch.zeros_like(f0)
    g0[0, :, 20, 30] = 4.0 / C * b.detach()[0, :, 20 - 4, 30 + 10]
    g1[0, :, 20 - 4, 30 + 10] = 4.0 / C * a.detach()[0, :, 20, 30]
    assert torch.equal(a.grad, g0) and torch.equal(b.grad, g1)


@pytest.mark.parametrize("B,H,W,C", [(2, 37, 70, 3), (3, 5, 7, 3), (1, 16, 64, 2), (3, 9, 11, 1)])
def test_warps_fp16_storage(oracle_mod, dev, B, H, W, C):
    """Extension (BASELINE config 5; not in the reference): backward_warp / image_warp with half images,
    flows and gradients in memory.  The arithmetic is the f32 kernels' on the widened values, so the result
    equals the f32 op on the same half-rounded inputs, rounded to half (bit for bit), and the oracle's
    within the final rounding (rtol 2e-3)."""
    from unflow_amd import ops
    from unflow_amd.image_warp import image_warp
    rng = np.random.default_rng(B * 100 + H + C)
    im = rng.uniform(0, 1, (B, H, W, C)).astype(np.float16)
    flow = (rng.standard_normal((B, H, W, 2)) * 6).astype(np.float16)
    flow[0, 0, 0] = [-0.0001, 1e-5]             # image_warp floors the FLOW: tiny negatives matter (D2b)
    gout = rng.standard_normal((B, H, W, C)).astype(np.float16)
    im32, fl32, g32 = im.astype(np.float32), flow.astype(np.float32), gout.astype(np.float32)
    for name, op16, op32, fwd_o, grad_o in (
            ("backward_warp", ops.backward_warp_fp16, ops.backward_warp, oracle_mod.backward_warp, oracle_mod.backward_warp_grad),
            ("image_warp", ops.image_warp_fp16, image_warp, oracle_mod.image_warp,
             lambda g, i, f: oracle_mod.image_warp_grad(g, i, f, want_im=False)[1])):
        f = torch.from_numpy(flow).to(dev).requires_grad_(True)
        got = op16(torch.from_numpy(im).to(dev), f)
        assert got.dtype == torch.float16 and tuple(got.shape) == im.shape, name
        f32 = torch.from_numpy(fl32).to(dev).requires_grad_(True)
        ref = op32(torch.from_numpy(im32).to(dev), f32)
        assert torch.equal(got, ref.half()), name
        np.testing.assert_allclose(got.detach().float().cpu().numpy(), fwd_o(im32, fl32), rtol=2e-3, atol=1e-6, err_msg=name)
        got.backward(torch.from_numpy(gout).to(dev))
        ref.backward(torch.from_numpy(g32).to(dev))
        assert f.grad.dtype == torch.float16
        assert torch.equal(f.grad, f32.grad.half()), name
        if grad_o is not None:
            want = grad_o(g32, im32, fl32)
            np.testing.assert_allclose(f.grad.float().cpu().numpy(), want, rtol=2e-3, atol=2e-4 * np.abs(want).max(), err_msg=name)
    with pytest.raises(TypeError):
        ops.image_warp_fp16(torch.from_numpy(im32).to(dev), torch.from_numpy(flow).to(dev))
    with pytest.raises(ValueError, match="channels"):
        ops.image_warp_fp16(torch.zeros((1, 4, 4, 5), dtype=torch.float16, device=dev), torch.zeros((1, 4, 4, 2), dtype=torch.float16, device=dev))
    with pytest.raises(ValueError, match="single-pixel"):      # its 8-byte tap would reach in front of the tensor
        ops.image_warp_fp16(torch.zeros((1, 1, 1, 3), dtype=torch.float16, device=dev), torch.zeros((1, 1, 1, 2), dtype=torch.float16, device=dev))
    one = ops.image_warp_fp16(torch.ones((1, 1, 2, 3), dtype=torch.float16, device=dev), torch.zeros((1, 1, 2, 2), dtype=torch.float16, device=dev))
    assert torch.equal(one, torch.ones_like(one))
    # a half tensor at an odd half (2-byte aligned only) is refused, not read with 4-byte accesses
    big = torch.zeros(B * H * W * 2 + 1, dtype=torch.float16, device=dev)
    odd = big[1:].view(B, H, W, 2)
    assert odd.data_ptr() % 4 == 2
    with pytest.raises(ValueError, match="4-byte aligned"):
        ops.image_warp_fp16(torch.from_numpy(im).to(dev), odd)


@pytest.mark.parametrize("B,H,W", [(2, 40, 70), (1, 7, 130)])
def test_forward_warp_fp16_storage(oracle_mod, dev, B, H, W):
    """forward_warp of half flows: the density is accumulated and returned in f32 and equals the f32 op on
    the widened flows up to the order of the float atomics of the flush."""
    from unflow_amd import ops
    rng = np.random.default_rng(B + H + W)
    flow = (rng.standard_normal((B, H, W, 2)) * 5).astype(np.float16)
    got = ops.forward_warp_fp16(torch.from_numpy(flow).to(dev))
    assert got.dtype == torch.float32 and tuple(got.shape) == (B, H, W, 1)
    want = oracle_mod.forward_warp(flow.astype(np.float32))
    close(got, want, 1e-5, 1e-5)
    ref = ops.forward_warp(torch.from_numpy(flow.astype(np.float32)).to(dev))
    close(got, N(ref), 1e-6, 1e-6)
    with pytest.raises(TypeError):
        ops.forward_warp_fp16(torch.from_numpy(flow.astype(np.float32)).to(dev))


@pytest.mark.parametrize("shape", [(2, 32, 16, 24), (1, 7, 9, 13), (2, 16, 48, 160)])
def test_correlation_bidirectional(oracle_mod, dev, shape):
    """Extension for the bidirectional FlowNetC (core/flownet.py:33-43): the cost volume of (b, a) is a
    bit-exact re-indexing of that of (a, b), and one backward call yields the summed gradients."""
    from unflow_amd import ops
    rng = np.random.default_rng(sum(shape) + 1)
    f0 = rng.standard_normal(shape).astype(np.float32)
    f1 = rng.standard_normal(shape).astype(np.float32)
    a, b = T(f0, dev).requires_grad_(True), T(f1, dev).requires_grad_(True)
    c_fw, c_bw = ops.correlation_bidirectional(a, b)
    assert torch.equal(c_fw, ops.correlation(a.detach(), b.detach()))
    assert torch.equal(c_bw, ops.correlation(b.detach(), a.detach()))          # bit-identical
    assert torch.equal(ops.correlation_flip(c_bw.detach()), c_fw.detach())       # the map is its own inverse
    want_bw, _, _ = oracle_mod.correlation(f1, f0)
    close(c_bw, want_bw)
    t_fw = T(rng.standard_normal(c_fw.shape).astype(np.float32), dev)
    t_bw = T(rng.standard_normal(c_fw.shape).astype(np.float32), dev)
    ((c_fw * t_fw).sum() + (c_bw * t_bw).sum()).backward()
    g0a, g1a = ops.correlation_grad(t_fw, a.detach(), b.detach())                # direction (a, b)
    g0b, g1b = ops.correlation_grad(t_bw, b.detach(), a.detach())                # direction (b, a)
    wa, wb = N(g0a + g1b), N(g1a + g0b)
    close(a.grad, wa, 1e-5, 1e-5 * np.abs(wa).max())
    close(b.grad, wb, 1e-5, 1e-5 * np.abs(wb).max())


def test_correlation_errors(dev):
    from unflow_amd import ops
    f = torch.zeros((1, 1, 4, 4), device=dev)
    with pytest.raises(ValueError, match="kernel_size must be odd"):
        ops.correlation(f, f, kernel_size=2)
    with pytest.raises(ValueError, match="Input shapes have to be the same"):
        ops.correlation(f, torch.zeros((1, 1, 4, 5), device=dev))
    with pytest.raises(ValueError, match="Invalid correlation settings"):
        ops.correlation(f, f, max_displacement=20, pad=0)
    with pytest.raises(ValueError, match="divisible by scale"):
        ops.downsample(torch.zeros((1, 5, 4, 1), device=dev), 2)


@pytest.mark.parametrize("B,H,W,C,levels", [(2, 128, 192, 3, 5), (3, 64, 64, 1, 5), (1, 96, 160, 3, 3), (2, 40, 72, 1, 2),
                                            (1, 384, 1280, 3, 5), (2, 12, 20, 3, 1)])
def test_downsample_pyramid_is_the_chain_bit_for_bit(dev, B, H, W, C, levels):
    """ops.downsample_pyramid (one launch for the image chain of core/unsupervised.py:97-103,143-145) against the chain of
    ops.downsample calls it replaces: every level bit-identical, for one tensor and for the stacked orders [a; b], [b; a];
    extents that are not multiples of the kernel's 64 x 64 block; the chain's own error for indivisible extents."""
    from unflow_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(B * 100 + H)
    a = torch.rand((B, H, W, C), device=dev, generator=g)
    b = torch.rand((B, H, W, C), device=dev, generator=g)

    def chain(x):
        out = [ops.downsample(x, 4)]
        for _ in range(levels - 1):
            out.append(ops.downsample(out[-1], 2))
        return out

    ca, cb = chain(a), chain(b)
    one = ops.downsample_pyramid(a, None, levels)
    ab = ops.downsample_pyramid(a, b, levels)
    ab2, ba2 = ops.downsample_pyramid(a, b, levels, both_orders=True)
    for l in range(levels):
        assert one[l].shape == ca[l].shape and torch.equal(one[l], ca[l]), l
        assert torch.equal(ab[l], torch.cat((ca[l], cb[l]))), l
        assert torch.equal(ab2[l], ab[l]) and torch.equal(ba2[l], torch.cat((cb[l], ca[l]))), l
    aa = ops.downsample_pyramid(a, a, levels)              # one tensor twice: read once, both halves written
    for l in range(levels):
        assert torch.equal(aa[l], torch.cat((ca[l], ca[l]))), l
    with pytest.raises(ValueError, match="divisible by scale"):
        ops.downsample_pyramid(torch.zeros((1, 8 << levels, (4 << levels) - 4, C), device=dev), None, levels + 1 if levels < 5 else 5)
    with pytest.raises(ValueError, match="both_orders"):
        ops.downsample_pyramid(a, None, levels, both_orders=True)


def _pair(rng, B, H, W):
    from tests.golden.make_golden import smooth_field
    im1 = smooth_field(rng, (B, H, W, 3), 2.0)
    im2 = smooth_field(rng, (B, H, W, 3), 2.0)
    flow = (rng.standard_normal((B, H, W, 2)) * 4 + rng.uniform(-10, 10, (B, 1, 1, 2))).astype(np.float32)
    return im1, im2, flow


@pytest.mark.parametrize("B,H,W,C", [(2, 37, 53, 3), (1, 64, 96, 2), (3, 8, 8, 1), (1, 40, 33, 5)])
def test_warps_vs_oracle(oracle_mod, dev, B, H, W, C):
    from unflow_amd import ops
    from unflow_amd.image_warp import image_warp, image_warp_grad
    rng = np.random.default_rng(B * 1000 + H)
    im = rng.uniform(size=(B, H, W, C)).astype(np.float32)
    flow = (rng.standard_normal((B, H, W, 2)) * 5).astype(np.float32)
    flow[0, 0, 0] = [-1000.0, 1000.0]   # far outside
    flow[0, 1, 1] = [0.0, 0.0]          # exactly on the grid
    flow[0, 2, 2] = [-1e-7, -1e-7]      # D2b: tiny negative flow
    grad = rng.standard_normal(im.shape).astype(np.float32)
    close(ops.backward_warp(T(im, dev), T(flow, dev)), oracle_mod.backward_warp(im, flow))
    close(ops.backward_warp_grad(T(grad, dev), T(im, dev), T(flow, dev)), oracle_mod.backward_warp_grad(grad, im, flow))
    close(image_warp(T(im, dev), T(flow, dev)), oracle_mod.image_warp(im, flow))
    d_im, d_flow = image_warp_grad(T(grad, dev), T(im, dev), T(flow, dev))
    w_im, w_flow = oracle_mod.image_warp_grad(grad, im, flow)
    close(d_im, w_im, 1e-5, 1e-5)
    close(d_flow, w_flow, 1e-5, 1e-5)


@pytest.mark.parametrize("B,H,W,C,shift", [(2, 37, 53, 3, (-9.5, 4.25)), (1, 50, 75, 2, (20.0, -13.0)), (2, 33, 40, 1, (0.0, 0.0)),
                                           (1, 48, 96, 3, (-60.0, 30.0))])
def test_image_warp_image_gradient_coherent_flow(oracle_mod, dev, B, H, W, C, shift):
    """The LDS-window path of the image gradient (csrc/warp.hip image_warp_bwd_im_tiled_kernel: 32 x 16 pixel tiles,
    fixed-point window, a pixel's right-hand taps merged into its right neighbour's through DPP) on affine flows --
    stretched, compressed and sheared, so that floor(flow) changes inside the 16-lane rows, that neighbouring lanes
    share cells, and that whole rows and columns of taps are clamped at the image border (a translation that pushes
    part of the image outside).  Ragged extents: lanes beyond the image take no part in the merge."""
    from unflow_amd.image_warp import image_warp_grad
    rng = np.random.default_rng(H * 100 + W)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    flow = np.empty((B, H, W, 2), np.float32)
    for b in range(B):
        ax, ay = (0.23, -0.31) if b == 0 else (-0.4, 0.12)          # px / px: stretch in one item, compression in the other
        flow[b, ..., 0] = shift[0] + ax * (xx - W / 2) + 0.11 * (yy - H / 2)
        flow[b, ..., 1] = shift[1] + ay * (yy - H / 2) - 0.07 * (xx - W / 2)
    flow += rng.uniform(-0.01, 0.01, flow.shape).astype(np.float32)
    im = rng.uniform(size=(B, H, W, C)).astype(np.float32)
    grad = rng.standard_normal(im.shape).astype(np.float32)
    grad[0, 3:5, 7:20] = 0.0
    d_im, d_flow = image_warp_grad(T(grad, dev), T(im, dev), T(flow, dev))
    w_im, w_flow = oracle_mod.image_warp_grad(grad, im, flow)
    close(d_im, w_im, 1e-5, 1e-5 * float(np.abs(w_im).max()))
    close(d_flow, w_flow, 1e-5, 1e-5 * float(np.abs(w_flow).max()))
    # the image gradient alone takes the memset + scatter route instead of the zeroing flow kernel
    only_im, _ = image_warp_grad(T(grad, dev), T(im, dev), T(flow, dev), True, False)
    close(only_im, w_im, 1e-5, 1e-5 * float(np.abs(w_im).max()))
    # total mass: every tap weight quadruple sums to one, so the gradient's sum is conserved per channel
    np.testing.assert_allclose(only_im.cpu().numpy().sum(axis=(0, 1, 2)), grad.sum(axis=(0, 1, 2)), rtol=1e-4, atol=1e-3)


def test_image_warp_autograd(oracle_mod, dev):
    from unflow_amd.image_warp import image_warp
    rng = np.random.default_rng(11)
    im = rng.uniform(size=(2, 16, 20, 2)).astype(np.float32)
    flow = (rng.standard_normal((2, 16, 20, 2)) * 3).astype(np.float32)
    grad = rng.standard_normal(im.shape).astype(np.float32)
    a, f = T(im, dev).requires_grad_(True), T(flow, dev).requires_grad_(True)
    image_warp(a, f).backward(T(grad, dev))
    w_im, w_flow = oracle_mod.image_warp_grad(grad, im, flow)
    close(a.grad, w_im)
    close(f.grad, w_flow)


@pytest.mark.parametrize("B,H,W", [(2, 37, 53), (1, 10, 10), (1, 96, 128)])
def test_forward_warp_vs_oracle(oracle_mod, dev, B, H, W):
    from unflow_amd import ops
    rng = np.random.default_rng(H)
    flow = (rng.standard_normal((B, H, W, 2)) * 6).astype(np.float32)
    flow[0, 0, 0] = [-40.0, 3.0]
    flow[0, 1, 1] = [float(W), float(H)]
    want = oracle_mod.forward_warp(flow)
    f = T(flow, dev).requires_grad_(True)
    got = ops.forward_warp(f)
    # fp32 atomics in arbitrary order: tolerance relative to the value range (~2*pi*overlap)
    close(got, want, 1e-5, 1e-5 * max(1.0, want.max()))
    g = rng.standard_normal(want.shape).astype(np.float32)
    got.backward(T(g, dev))
    close(f.grad, oracle_mod.forward_warp_grad(g, flow), 1e-5, 1e-5)


def test_forward_warp_everything_lands_on_one_pixel(oracle_mod, dev):
    """Worst case of the splat's 32-bit fixed-point LDS window (2^-21 granule, at most 1024 source pixels per
    window): every pixel of a 128x32 image -- four full 64x16 tiles -- flows to ONE target pixel with weight
    exp(0) = 1, i.e. each tile adds 1024 * 2^21 = 2^31 to one cell.  The density there is the pixel count."""
    from unflow_amd import ops
    H, W = 32, 128
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    flow = np.stack([70.0 - xs, 9.0 - ys], axis=-1)[None]          # x + u = 70, y + v = 9 for every pixel
    got = N(ops.forward_warp(T(flow, dev)))[0, :, :, 0]
    # exact: 4096 identical 9x9 Gaussians around (9, 70), clipped to the image
    exact = np.zeros((H, W))
    for dy in range(-4, 5):
        for dx in range(-4, 5):
            if 0 <= 9 + dy < H and 0 <= 70 + dx < W:
                exact[9 + dy, 70 + dx] = H * W * np.exp(-(dx * dx + dy * dy) / 2.0)
    assert got[9, 70] == float(H * W)                               # 4 x 2^31 window units, no overflow, no rounding
    # a contribution is rounded to 2^-21 (4.8e-7): at most 2.4e-7 * 4096 = 1e-3 off per cell
    np.testing.assert_allclose(got, exact, rtol=1e-6, atol=1.1e-3)
    # the oracle adds 4096 floats one after the other like the reference's float atomicAdd: its own rounding
    # (~sqrt(N) ulp(4096) = 0.03) is what separates it from the exact value, not the kernel
    want = oracle_mod.forward_warp(flow.astype(np.float32))[0, :, :, 0]
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1.1e-3)
    assert np.abs(got - exact).max() <= np.abs(want - exact).max() + 1.1e-3


def test_forward_warp_zero_flow_analytic(dev):
    from unflow_amd import ops
    out = N(ops.forward_warp(torch.zeros((1, 12, 12, 2), device=dev)))
    np.testing.assert_allclose(out[0, 6, 6, 0], 6.2831479, rtol=1e-6)
    np.testing.assert_allclose(out[0, 0, 0, 0], 3.0740974, rtol=1e-6)


# r = 2, 3 run the pair-symmetric kernels (58- or 60-column x 64/32-row workgroup tiles): sizes
# below, at and across those tile borders, single rows / columns, an image smaller than the window
@pytest.mark.parametrize("B,H,W,r", [(2, 37, 53, 3), (1, 16, 16, 1), (2, 24, 40, 2), (1, 9, 70, 3), (1, 20, 20, 4),
                                     (1, 70, 130, 3), (2, 64, 58, 3), (1, 65, 59, 3), (1, 33, 121, 2),
                                     (1, 1, 200, 3), (1, 100, 1, 3), (1, 2, 2, 3), (3, 5, 4, 2)])
def test_census_vs_oracle(oracle_mod, dev, B, H, W, r):
    from unflow_amd import losses
    rng = np.random.default_rng(B * 100 + W)
    im1, im2, _ = _pair(rng, B, H, W)
    mask = (rng.uniform(size=(B, H, W, 1)) > 0.15).astype(np.float32)
    want_loss, want_dist = oracle_mod.ternary_loss(im1, im2, mask, r, return_dist=True)
    # dist_sum lies in [0, P*P]; 1e-5 of that range.  Per-tap 1-ulp differences
    # (v_rsq/v_rcp vs IEEE sqrt/div) are amplified by the t1-t2 cancellation on smooth images.
    P2 = (2 * r + 1) ** 2
    got_dist = N(losses.ternary_dist(T(im1, dev), T(im2, dev), r))
    assert np.abs(got_dist - want_dist).max() <= 1e-5 * P2
    assert np.abs(got_dist - want_dist).mean() <= 1e-6 * P2
    x = T(im2, dev).requires_grad_(True)
    loss = losses.ternary_loss(T(im1, dev), x, T(mask, dev), max_distance=r)
    np.testing.assert_allclose(float(loss.detach()), want_loss, rtol=1e-5)
    (loss * 2.5).backward()
    want_grad = oracle_mod.ternary_loss_grad(im1, im2, mask, r, gout=2.5)
    close(x.grad, want_grad, 1e-5, 1e-5 * np.abs(want_grad).max())      # measured 4e-7 of the scale


@pytest.mark.parametrize("B,H,W", [(16, 256, 640), (16, 384, 640), (24, 384, 640)])   # strips of 8, 12 and 16 rows on 256 CUs
def test_census_strip_choice_does_not_change_results(dev, B, H, W):
    """The pair kernels pick their rows-per-wave strip (4, 8, 12 or 16) from the launch size (census.hip pair_rows): a
    launch that does not fill the chip takes short strips.  The per-pixel arithmetic does not depend on the strip: the
    whole batch (long strips) must give, pixel for pixel, the gradients of its items run one at a time (one item is a
    small launch: strips of 4), and the same loss up to the rounding of the block-partial sums."""
    from unflow_amd import losses
    g = torch.Generator(device=dev)
    g.manual_seed(B + H)
    im1 = torch.rand((B, H, W, 3), device=dev, generator=g)
    im2 = torch.rand((B, H, W, 3), device=dev, generator=g)
    mask = (torch.rand((B, H, W, 1), device=dev, generator=g) > 0.2).float()
    for r in (2, 3):
        x = im2.clone().requires_grad_(True)
        whole = losses.ternary_loss(im1, x, mask, max_distance=r)
        whole.backward()
        total = 0.0
        for b in range(0, B, max(B // 4, 1)):           # four items are enough
            xb = im2[b:b + 1].clone().requires_grad_(True)
            lb = losses.ternary_loss(im1[b:b + 1], xb, mask[b:b + 1], max_distance=r, norm_count=B * H * W)
            lb.backward()
            assert torch.equal(xb.grad, x.grad[b:b + 1]), (r, b)
        # the loss itself: every item one at a time
        with torch.no_grad():
            for b in range(B):
                total += float(losses.ternary_loss(im1[b:b + 1], im2[b:b + 1], mask[b:b + 1], max_distance=r, norm_count=B * H * W))
        np.testing.assert_allclose(total, float(whole.detach()), rtol=1e-6)


def test_census_backward_ignores_memory_past_its_buffers(oracle_mod, dev):
    """The pair-symmetric backward reads dldd with linear addressing and relies on the buffer
    descriptor's range check past the end: poison the allocator's free memory first."""
    from unflow_amd import losses
    rng = np.random.default_rng(91)
    B, H, W = 2, 40, 70
    im1, im2, _ = _pair(rng, B, H, W)
    mask = np.ones((B, H, W, 1), np.float32)
    want = oracle_mod.ternary_loss_grad(im1, im2, mask, 3)
    for _ in range(3):
        poison = torch.full((64 * 1024 * 1024 // 4,), 1e30, device=dev)   # 64 MB of 1e30 ...
        del poison                                                       # ... back to the caching allocator
        x = T(im2, dev).requires_grad_(True)
        losses.ternary_loss(T(im1, dev), x, T(mask, dev), max_distance=3).backward()
        close(x.grad, want, 1e-5, 1e-5 * np.abs(want).max())


def test_charbonnier_and_masks_vs_oracle(oracle_mod, dev):
    from unflow_amd import losses
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 9, 11, 3)).astype(np.float32)
    for mask in (None, (rng.uniform(size=(2, 9, 11, 1)) > 0.3).astype(np.float32),
                 (rng.uniform(size=(2, 9, 11, 3)) > 0.3).astype(np.float32)):
        for beta in (1.0, 255.0):
            xt = T(x, dev).requires_grad_(True)
            m = None if mask is None else T(mask, dev)
            loss = losses.charbonnier_loss(xt, m, beta=beta)
            np.testing.assert_allclose(float(loss.detach()), oracle_mod.charbonnier_loss(x, mask, beta=beta), rtol=1e-5)
            loss.backward()
            want = oracle_mod.charbonnier_loss_grad(x, mask, beta=beta)
            close(xt.grad, want, 1e-5, 1e-6 * max(1.0, np.abs(want).max()))
    z = torch.zeros((2, 5, 7, 1), device=dev)
    np.testing.assert_array_equal(N(losses.create_mask(z, [[1, 2], [0, 3]])), oracle_mod.create_mask((2, 5, 7), [[1, 2], [0, 3]]))
    big = torch.zeros((1, 384, 1280, 1), device=dev)
    np.testing.assert_array_equal(N(losses.create_border_mask(big)), oracle_mod.create_border_mask((1, 384, 1280)))
    f = (rng.standard_normal((2, 9, 11, 2)) * 3).astype(np.float32)
    w = (rng.standard_normal((2, 9, 11, 2)) * 3).astype(np.float32)
    diff, occ = losses.fb_occlusion(T(f, dev), T(w, dev))
    wd, wo = oracle_mod.fb_occlusion(f, w)
    close(diff, wd, 0, 0)
    np.testing.assert_array_equal(N(occ), wo)
    # the mask product of compute_losses (losses.py:53-55): in place, and with a separate input (unflow_fb_occlusion_masked)
    m = rng.random((2, 9, 11, 1)).astype(np.float32)
    m_inplace = T(m, dev)
    losses.fb_occlusion(T(f, dev), T(w, dev), m_inplace)
    np.testing.assert_array_equal(N(m_inplace), m * (1 - wo))
    m_src, m_out = T(m, dev), torch.full((2, 9, 11, 1), float("nan"), device=dev)
    losses.fb_occlusion(T(f, dev), T(w, dev), m_out, mask_in=m_src)
    np.testing.assert_array_equal(N(m_out), m * (1 - wo))
    np.testing.assert_array_equal(N(m_src), m)
    with pytest.raises(ValueError):
        losses.fb_occlusion(T(f, dev), T(w, dev), None, mask_in=m_src)


@pytest.mark.parametrize("B,H,W", [(2, 40, 56), (1, 9, 33), (3, 8, 32), (1, 1, 1), (1, 2, 70), (2, 17, 3)])
def test_stencil_terms_vs_oracle(oracle_mod, dev, B, H, W):
    """smoothness / second-order / gradient loss, deltas bit-exact, losses 1e-5, gradients 1e-5 of their scale
    (tile 32x8: sizes below, at and across tile borders; degenerate 1x1)."""
    from unflow_amd import losses
    rng = np.random.default_rng(100 + H * W)
    flow = (rng.standard_normal((B, H, W, 2)) * 3).astype(np.float32)
    flow[0, 0, 0] = 0      # a delta of exactly zero
    for order, dfn, lfn in ((1, losses._smoothness_deltas, losses.smoothness_loss),
                            (2, losses._second_order_deltas, losses.second_order_loss)):
        want = oracle_mod.smoothness_deltas(flow, order)
        got = dfn(T(flow, dev))
        for g, w in zip(got, want):
            np.testing.assert_array_equal(N(g), w)
        f = T(flow, dev).requires_grad_(True)
        loss = lfn(f)
        np.testing.assert_allclose(float(loss.detach()), oracle_mod.smoothness_loss(flow, order), rtol=1e-5)
        (loss * 3.0).backward()
        wg = oracle_mod.smoothness_loss_grad(flow, order, gout=3.0)
        close(f.grad, wg, 1e-5, 1e-5 * max(np.abs(wg).max(), 1e-30))
        # a shard normalises by the global count
        np.testing.assert_allclose(float(lfn(T(flow, dev), norm_count=7 * B * H * W)),
                                   oracle_mod.smoothness_loss(flow, order, norm_count=7 * B * H * W), rtol=1e-5)
    im1 = rng.uniform(size=(B, H, W, 3)).astype(np.float32)
    im2 = rng.uniform(size=(B, H, W, 3)).astype(np.float32)
    mask = (rng.uniform(size=(B, H, W, 1)) > 0.2).astype(np.float32)
    np.testing.assert_array_equal(N(losses._gradient_delta(T(im1, dev), T(im2, dev))),
                                  oracle_mod.gradient_delta(im1, im2))
    x = T(im2, dev).requires_grad_(True)
    loss = losses.gradient_loss(T(im1, dev), x, T(mask, dev))
    np.testing.assert_allclose(float(loss.detach()), oracle_mod.gradient_loss(im1, im2, mask), rtol=1e-5)
    loss.backward()
    wg = oracle_mod.gradient_loss_grad(im1, im2, mask)
    close(x.grad, wg, 1e-5, 1e-5 * max(np.abs(wg).max(), 1e-30))


@pytest.mark.parametrize("mask_occlusion,use_border", [("fb", True), ("", False), ("disocc", True)])
def test_compute_losses_vs_oracle(oracle_mod, dev, mask_occlusion, use_border):
    from unflow_amd import losses
    rng = np.random.default_rng(21)
    B, H, W = 2, 40, 56
    im1, im2, flow_fw = _pair(rng, B, H, W)
    flow_bw = (-flow_fw + rng.standard_normal(flow_fw.shape) * 0.5).astype(np.float32)
    border = oracle_mod.create_border_mask((B, H, W)) if use_border else None
    want, waux = oracle_mod.compute_losses(im1, im2, flow_fw, flow_bw, border, mask_occlusion, 3)
    ffw, fbw = T(flow_fw, dev).requires_grad_(True), T(flow_bw, dev).requires_grad_(True)
    got, aux = losses.compute_losses(T(im1, dev), T(im2, dev), ffw, fbw,
                                     None if border is None else T(border, dev), mask_occlusion, 3,
                                     return_aux=True)
    # Thresholded maps (occlusion, disocclusion): a pixel within fp32 noise of its threshold may
    # fall on the other side, and at 4480 pixels ONE flipped pixel moves a scalar by 2e-4.  So the
    # maps are compared for equality here (measured: no pixel differs on these inputs,
    # tools/tolerance_probe.py), which is what lets the scalars hold north_star's 1e-5.
    for k in ("mask_fw", "mask_bw", "disocc_fw", "disocc_bw", "fb_occ_fw", "fb_occ_bw"):
        np.testing.assert_array_equal(N(aux[k]), waux[k], err_msg=k)
    close(aux["im2_warped"], waux["im2_warped"])
    assert set(got) == {"sym", "occ", "photo", "grad", "smooth_1st", "smooth_2nd", "fb", "ternary"}  # losses.py:60-85
    for k in ("ternary", "occ", "fb", "photo", "sym", "grad", "smooth_1st", "smooth_2nd"):
        np.testing.assert_allclose(float(got[k].detach()), want[k], rtol=1e-5, err_msg=k)
    # backward to the flows through census -> image_warp
    got["ternary"].backward()
    gim2 = oracle_mod.ternary_loss_grad(im1, waux["im2_warped"], waux["mask_fw"], 3)
    _, want_dfw = oracle_mod.image_warp_grad(gim2, im2, flow_fw, want_im=False)
    gim1 = oracle_mod.ternary_loss_grad(im2, waux["im1_warped"], waux["mask_bw"], 3)
    _, want_dbw = oracle_mod.image_warp_grad(gim1, im1, flow_bw, want_im=False)
    # gradients: 1e-5 of the gradient's scale (measured 1.7e-6: per-tap rsq/rcp ulps through the
    # t1 - t2 cancellation, then the warp's finite differences)
    close(ffw.grad, want_dfw, 1e-5, 1e-5 * np.abs(want_dfw).max())
    close(fbw.grad, want_dbw, 1e-5, 1e-5 * np.abs(want_dbw).max())


KITTI_MIX = dict(ternary=1.0, smooth_2nd=3.0, fb=0.2, occ=12.4)       # config_template/config.ini:123,166-174


@pytest.mark.parametrize("mask_occlusion,use_border,weights", [
    ("fb", True, KITTI_MIX),
    ("fb", True, dict(KITTI_MIX, photo=0.7, grad=0.5, smooth_1st=0.3, sym=0.1)),   # every entry of losses.py:60-85
    ("", False, dict(KITTI_MIX, photo=0.7, grad=0.5)),
    ("disocc", True, dict(ternary=1.0, fb=0.2, grad=0.5)),
])
@pytest.mark.parametrize("restrict_terms", [False, True])
def test_compute_losses_full_mix_gradient_vs_oracle(oracle_mod, dev, mask_occlusion, use_border, weights, restrict_terms):
    """The gradient the trainer differentiates (losses.py:16-87 under the reference's default loss mix):
    sum_k w_k * losses[k] back to BOTH flows -- ternary / photo / grad through image_warp's flow
    gradient, fb through _FlowDiff into image_warp's flow gradient AND its image-gradient scatter
    into the other flow, the stencil terms directly -- against oracle.compute_losses_grad, the
    composition checked by central differences in tests/test_oracle_gradients.py.
    restrict_terms: only the weighted entries are evaluated (the census then runs fused with its
    warp); otherwise all eight, through the materialised warped images."""
    from unflow_amd import losses
    rng = np.random.default_rng(77)
    B, H, W = 2, 40, 56
    im1, im2, flow_fw = _pair(rng, B, H, W)
    flow_bw = (-flow_fw + rng.standard_normal(flow_fw.shape) * 0.5).astype(np.float32)
    border = oracle_mod.create_border_mask((B, H, W)) if use_border else None
    want_dfw, want_dbw, want = oracle_mod.compute_losses_grad(im1, im2, flow_fw, flow_bw, border, mask_occlusion, 3,
                                                              weights)
    ffw, fbw = T(flow_fw, dev).requires_grad_(True), T(flow_bw, dev).requires_grad_(True)
    got = losses.compute_losses(T(im1, dev), T(im2, dev), ffw, fbw, None if border is None else T(border, dev),
                                mask_occlusion, 3, terms=set(weights) if restrict_terms else None)
    total = sum(w * got[k] for k, w in weights.items())
    np.testing.assert_allclose(float(total.detach()), sum(w * want[k] for k, w in weights.items()), rtol=1e-5)
    total.backward()
    # 1e-5 of the gradient's scale (the bar of the single-term tests above)
    close(ffw.grad, want_dfw, 1e-5, 1e-5 * np.abs(want_dfw).max())
    close(fbw.grad, want_dbw, 1e-5, 1e-5 * np.abs(want_dbw).max())


@pytest.mark.parametrize("full_res", [False, True])
def test_pyramid_loss_kitti_mix_gradient_vs_oracle(oracle_mod, dev, full_res):
    """core/unsupervised.py:83-145 with the reference's default KITTI weights (+ grad): the gradient of
    the combined scalar to EVERY flow of the pyramid, both directions, against the oracle's
    composition (layer weight x flow scale x compute_losses_grad per level)."""
    from unflow_amd import unsupervised
    rng = np.random.default_rng(34)
    B, H, W = 2, 128, 192
    im1, im2, _ = _pair(rng, B, H, W)
    border = oracle_mod.create_border_mask((B, H, W))
    levels = 7 if full_res else 5
    h0, w0 = (H, W) if full_res else (H // 4, W // 4)
    fws = [(rng.standard_normal((B, h0 >> i, w0 >> i, 2)) * 0.3).astype(np.float32) for i in range(levels)]
    bws = [(-f + rng.standard_normal(f.shape) * 0.05).astype(np.float32) for f in fws]
    params = dict(full_res=full_res, pyramid_loss=True, mask_occlusion="fb", border_mask=True,
                  ternary_weight=1.0, smooth_2nd_weight=3.0, fb_weight=0.2, occ_weight=12.4, grad_weight=0.5)
    want_fw, want_bw = oracle_mod.pyramid_loss_grad(im1, im2, border, fws, bws, params)
    tf_ = [T(f, dev).requires_grad_(True) for f in fws]
    tb_ = [T(f, dev).requires_grad_(True) for f in bws]
    got, _ = unsupervised.pyramid_loss(T(im1, dev), T(im2, dev), T(border, dev), tf_, tb_, params)
    got.backward()
    for i in range(levels):
        for t, w in ((tf_[i], want_fw[i]), (tb_[i], want_bw[i])):
            close(t.grad, w, 1e-5, 1e-5 * np.abs(w).max())


@pytest.mark.parametrize("mask_occlusion,with_border,terms", [
    ("fb", True, None), ("fb", False, None), ("disocc", True, None), ("", False, None),
    ("fb", True, ("ternary", "smooth_2nd", "occ")), ("fb", False, ("ternary",)), ("", True, ("sym", "photo", "fb"))])
def test_compute_losses_stacked_equals_the_two_directions(oracle_mod, dev, mask_occlusion, with_border, terms):
    """losses.compute_losses_stacked (both directions of losses.py:16-87 as one batch of 2B: [im1; im2], [im2; im1],
    [flow_fw; flow_bw]) against compute_losses (one pass per direction) and against the oracle: every term, and the
    gradient of a weighted sum of all of them to both flows."""
    from unflow_amd import losses
    rng = np.random.default_rng(91)
    B, H, W = 3, 40, 72
    im1, im2, _ = _pair(rng, B, H, W)
    ffw = (rng.standard_normal((B, H, W, 2)) * 1.5 + rng.uniform(-4, 4, (B, 1, 1, 2))).astype(np.float32)
    fbw = (-ffw + rng.standard_normal(ffw.shape) * 0.4).astype(np.float32)
    border = oracle_mod.create_border_mask((B, H, W)) if with_border else None
    names = terms or ("occ", "sym", "fb", "grad", "ternary", "photo", "smooth_1st", "smooth_2nd")
    wts = {k: 0.3 + 0.2 * i for i, k in enumerate(names)}

    def run(stacked):
        a, b = T(ffw, dev).requires_grad_(True), T(fbw, dev).requires_grad_(True)
        i1, i2 = T(im1, dev), T(im2, dev)
        m = T(border, dev) if with_border else None
        if stacked:
            parts = losses.compute_losses_stacked(torch.cat((i1, i2)), torch.cat((i2, i1)), torch.cat((a, b)),
                                                  border_mask=torch.cat((m, m)) if with_border else None,
                                                  mask_occlusion=mask_occlusion, data_max_distance=2, terms=terms)
            assert all(len(v) == 1 for v in parts.values())
            vals = {k: v[0] for k, v in parts.items()}
        else:
            vals = losses.compute_losses(i1, i2, a, b, border_mask=m, mask_occlusion=mask_occlusion,
                                         data_max_distance=2, terms=terms)
        assert set(vals) == set(names)
        sum(wts[k] * vals[k] for k in names).backward()
        return {k: float(v.detach()) for k, v in vals.items()}, a.grad, b.grad

    v2, ga2, gb2 = run(False)
    v1, ga1, gb1 = run(True)
    want, _ = oracle_mod.compute_losses(im1, im2, ffw, fbw, border, mask_occlusion, 2)
    for k in names:
        np.testing.assert_allclose(v1[k], v2[k], rtol=5e-6, err_msg=k)
        np.testing.assert_allclose(v1[k], want[k], rtol=1e-5, err_msg=k)
    for g1, g2 in ((ga1, ga2), (gb1, gb2)):
        if g2 is None:
            assert g1 is None or float(g1.abs().max()) == 0.0
            continue
        scale = float(g2.abs().max())
        assert float((g1 - g2).abs().max()) <= 5e-6 * scale


def test_fb_occlusion_losses_is_the_three_passes(oracle_mod, dev):
    """losses.fb_occlusion_losses (occlusion test + the 'fb' and 'occ' scalars from one pass) against fb_occlusion,
    charbonnier_loss and occ_loss: the flow difference, the occlusion map and the mask bit for bit, the scalars to 5e-6
    (their partial sums are grouped differently); immediate and inside DeferredReductions; the gradient of the given
    value is charbonnier's."""
    from unflow_amd import losses
    rng = np.random.default_rng(5)
    B, H, W = 2, 33, 50
    f = T((rng.standard_normal((B, H, W, 2)) * 2).astype(np.float32), dev)
    w = T((-N(f) + rng.standard_normal((B, H, W, 2)) * 0.8).astype(np.float32), dev)
    m_in = T((rng.uniform(size=(B, H, W, 1)) > 0.2).astype(np.float32), dev)
    npx = B * H * W
    m_ref = torch.empty_like(m_in)
    d_ref, o_ref = losses.fb_occlusion(f, w, m_ref, mask_in=m_in)
    x = d_ref.clone().requires_grad_(True)
    l_fb_ref = losses.charbonnier_loss(x, m_ref, norm_count=npx * 2)
    l_occ_ref = losses.occ_loss(m_ref, norm_count=npx)
    l_fb_ref.backward()
    m_out = torch.empty_like(m_in)
    d, o, l_fb, l_occ = losses.fb_occlusion_losses(f, w, m_in, m_out, norm_fb=npx * 2, norm_occ=npx)
    assert torch.equal(d, d_ref) and torch.equal(o, o_ref) and torch.equal(m_out, m_ref)
    assert 0.0 < float(o.mean()) < 1.0
    np.testing.assert_allclose(float(l_fb), float(l_fb_ref.detach()), rtol=5e-6)
    np.testing.assert_allclose(float(l_occ), float(l_occ_ref), rtol=5e-6)
    x2 = d.clone().requires_grad_(True)
    losses._CharbonnierGiven.apply(x2, m_out, l_fb, npx * 2).backward()
    assert torch.equal(x2.grad, x.grad)
    # the other flow warped on the fly (image_warp(flow_other, flow) never written): same difference, map and mask as the
    # two passes, and _FlowDiffWarp's gradient is image_warp's
    from unflow_amd.image_warp import image_warp
    fo = T((-N(f) + rng.standard_normal((B, H, W, 2)) * 0.8).astype(np.float32), dev)
    fo_a, f_a = fo.clone().requires_grad_(True), f.clone().requires_grad_(True)
    w_a = image_warp(fo_a, f_a)
    m_w_ref = torch.empty_like(m_in)
    d_w_ref, o_w_ref = losses.fb_occlusion(f, w_a, m_w_ref, mask_in=m_in)
    m_w = torch.empty_like(m_in)
    d_w, o_w, l_fb_w, l_occ_w = losses.fb_occlusion_losses(f, fo, m_in, m_w, norm_fb=npx * 2, norm_occ=npx, warp_other=True)
    assert torch.equal(d_w, d_w_ref) and torch.equal(o_w, o_w_ref) and torch.equal(m_w, m_w_ref)
    np.testing.assert_allclose(float(l_occ_w), float(losses.occ_loss(m_w_ref, norm_count=npx)), rtol=5e-6)
    up = torch.randn_like(f)
    (losses._FlowDiff.apply(f_a, w_a, d_w_ref) * up).sum().backward()
    fo_b, f_b, f_c = fo.clone().requires_grad_(True), f.clone().requires_grad_(True), f.clone().requires_grad_(True)
    (losses._FlowDiffWarp.apply(f_c, fo_b, f_b, d_w) * up).sum().backward()
    assert float((fo_b.grad - fo_a.grad).abs().max()) <= 5e-6 * float(fo_a.grad.abs().max())      # (a float-atomic scatter)
    assert torch.equal(f_b.grad + f_c.grad, f_a.grad) or float((f_b.grad + f_c.grad - f_a.grad).abs().max()) <= 5e-6 * float(f_a.grad.abs().max())
    # in place (mask_in is mask_out) and deferred: the two scalars are the plan's last two entries
    m_io = m_in.clone()
    with losses.DeferredReductions(dev) as plan:
        before = losses.occ_loss(m_ref, norm_count=npx)                       # an entry in front of the pair
        _, _, l_fb_d, l_occ_d = losses.fb_occlusion_losses(f, w, m_io, m_io, norm_fb=npx * 2, norm_occ=npx)
        n = plan.plan.n
        assert n == 3 and l_fb_d.storage_offset() == 1 and l_occ_d.storage_offset() == 2
        total = torch.empty((), device=dev)
        sums = torch.empty(8, device=dev)
        plan.fold([0.0, 1.0, 10.0], [-1, -1, -1], [0.0, 0.0, 0.0], total, sums)
    assert torch.equal(m_io, m_ref)
    np.testing.assert_allclose(float(l_fb_d), float(l_fb), rtol=1e-6)
    np.testing.assert_allclose(float(l_occ_d), float(l_occ), rtol=1e-6)
    np.testing.assert_allclose(float(before), float(l_occ_ref), rtol=1e-6)
    np.testing.assert_allclose(float(total), float(l_fb) + 10.0 * float(l_occ), rtol=1e-6)


def test_pyramid_loss_stacked_levels_equal_the_unstacked(dev):
    """unsupervised.pyramid_loss evaluates the levels below STACK_BELOW_PIXELS with both directions as one batch:
    same scalar, same per-term sums, same gradients at every level as one pass per direction; and a pyramid whose
    finest levels stay unstacked while the coarse ones stack (full_res with a threshold in between)."""
    from unflow_amd import unsupervised
    rng = np.random.default_rng(34)
    B, H, W = 2, 128, 192
    im1, im2, _ = _pair(rng, B, H, W)
    border = np.ones((B, H, W, 1), np.float32)
    border[:, :12] = 0
    border[:, :, -20:] = 0
    params = dict(full_res=True, pyramid_loss=True, mask_occlusion="fb", border_mask=True,
                  ternary_weight=1.0, smooth_2nd_weight=3.0, fb_weight=0.2, occ_weight=12.4, grad_weight=0.5)
    fws = [(rng.standard_normal((B, H >> i, W >> i, 2)) * 0.3).astype(np.float32) for i in range(7)]
    bws = [(-f + rng.standard_normal(f.shape) * 0.05).astype(np.float32) for f in fws]

    def run(stack, below):
        old = unsupervised.STACK_DIRECTIONS, unsupervised.STACK_BELOW_PIXELS
        unsupervised.STACK_DIRECTIONS, unsupervised.STACK_BELOW_PIXELS = stack, below
        try:
            tf_ = [T(f, dev).requires_grad_(True) for f in fws]
            tb_ = [T(f, dev).requires_grad_(True) for f in bws]
            got, terms = unsupervised.pyramid_loss(T(im1, dev), T(im2, dev), T(border, dev), tf_, tb_, params)
            got.backward()
            return float(got.detach()), {k: float(v) for k, v in terms.items()}, [t.grad for t in tf_ + tb_]
        finally:
            unsupervised.STACK_DIRECTIONS, unsupervised.STACK_BELOW_PIXELS = old

    base = run(False, 0)
    for below in (1 << 30, B * (H // 4) * (W // 4)):      # every level stacked; the two finest unstacked
        got = run(True, below)
        np.testing.assert_allclose(got[0], base[0], rtol=5e-6)
        for k in base[1]:
            np.testing.assert_allclose(got[1][k], base[1][k], rtol=5e-6, atol=1e-12, err_msg=k)
        for g, w in zip(got[2], base[2]):
            assert float((g - w).abs().max()) <= 5e-6 * float(w.abs().max())


@pytest.mark.parametrize("full_res,pyramid", [(False, True), (True, True), (False, False)])
def test_pyramid_loss_vs_oracle(oracle_mod, dev, full_res, pyramid):
    """core/unsupervised.py:83-145: layer weights, patch distances, flow scaling and the
    downsample chain around compute_losses; gradient of the combined scalar to the finest flow."""
    from unflow_amd import unsupervised
    rng = np.random.default_rng(33)
    B, H, W = 2, 128, 192
    im1, im2, _ = _pair(rng, B, H, W)
    border = oracle_mod.create_border_mask((B, H, W))
    levels = 7 if full_res else 5
    h0, w0 = (H, W) if full_res else (H // 4, W // 4)
    fws = [(rng.standard_normal((B, h0 >> i, w0 >> i, 2)) * 0.3).astype(np.float32) for i in range(levels)]
    bws = [(-f + rng.standard_normal(f.shape) * 0.05).astype(np.float32) for f in fws]
    params = dict(full_res=full_res, pyramid_loss=pyramid, mask_occlusion="fb", border_mask=True,
                  ternary_weight=1.0, smooth_2nd_weight=3.0, fb_weight=0.2, occ_weight=12.4, grad_weight=0.5)
    want, want_terms = oracle_mod.pyramid_loss(im1, im2, border, fws, bws, params)
    tf_ = [T(f, dev).requires_grad_(True) for f in fws]
    tb_ = [T(f, dev).requires_grad_(True) for f in bws]
    got, got_terms = unsupervised.pyramid_loss(T(im1, dev), T(im2, dev), T(border, dev), tf_, tb_, params)
    np.testing.assert_allclose(float(got.detach()), want, rtol=1e-5)
    for k in unsupervised.LOSSES:
        np.testing.assert_allclose(float(got_terms[k]), want_terms[k], rtol=1e-5, atol=1e-12, err_msg=k)
    assert float(got_terms["photo"]) == 0.0 and float(got_terms["sym"]) == 0.0      # no weight: not evaluated
    got.backward()
    assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in tf_[:levels if pyramid else 1])
    if not pyramid:
        assert tf_[1].grad is None
        # d/d flows_fw[0] of  w0 * (ternary + 3 smooth_2nd)  composed from the oracle's pieces
        p2 = dict(params, fb_weight=0, occ_weight=0, grad_weight=0)
        f0 = T(fws[0], dev).requires_grad_(True)
        l2, _ = unsupervised.pyramid_loss(T(im1, dev), T(im2, dev), T(border, dev), [f0], [T(bws[0], dev)], p2)
        l2.backward()
        scale = np.float32(5.0)
        i1, i2, m = oracle_mod.downsample(im1, 4), oracle_mod.downsample(im2, 4), oracle_mod.downsample(border, 4)
        _, aux = oracle_mod.compute_losses(i1, i2, fws[0] * scale, bws[0] * scale, m, "fb", 3)
        g_im = oracle_mod.ternary_loss_grad(i1, aux["im2_warped"], aux["mask_fw"], 3)
        _, g_t = oracle_mod.image_warp_grad(g_im, i2, fws[0] * scale, want_im=False)
        g_s = oracle_mod.smoothness_loss_grad(fws[0] * scale, 2)
        wantg = 12.7 * scale * (g_t + 3.0 * g_s)
        close(f0.grad, wantg, 1e-5, 1e-5 * np.abs(wantg).max())


@pytest.mark.parametrize("B,H,W,h,w", [(2, 64, 96, 16, 24), (1, 37, 53, 10, 14), (3, 8, 8, 2, 2), (1, 5, 300, 5, 75)])
def test_stacked_inputs_vs_oracle(oracle_mod, dev, B, H, W, h, w):
    """flownet.py:46-57: fused block == composed block (bitwise) == oracle (1e-5)."""
    from unflow_amd import flownet
    rng = np.random.default_rng(B * 100 + H)
    im1 = rng.uniform(size=(B, H, W, 3)).astype(np.float32)
    im2 = rng.uniform(size=(B, H, W, 3)).astype(np.float32)
    flow = (rng.standard_normal((B, h, w, 2)) * 0.3).astype(np.float32)
    want = oracle_mod.stacked_inputs(im1, im2, flow)
    fused = flownet.stacked_inputs(T(im1, dev), T(im2, dev), T(flow, dev))
    assert fused.shape == (B, H, W, 14)
    close(fused, want, 1e-5, 1e-5)
    np.testing.assert_array_equal(N(fused[..., 6:8]), want[..., 6:8])        # the resized flow is bit-exact
    f = T(flow, dev).requires_grad_(True)
    comp = flownet.stacked_inputs(T(im1, dev), T(im2, dev), f, train_all=True)
    np.testing.assert_array_equal(N(comp), N(fused))
    # train_all: gradient reaches the low-resolution flow through warp and diff
    g = rng.standard_normal(want.shape).astype(np.float32)
    (comp * T(g, dev)).sum().backward()
    full = want[..., 6:8]
    warp = want[..., 8:11]
    g_warp = g[..., 8:11] + g[..., 11:14] * np.sign(warp - im1)
    _, g_full = oracle_mod.image_warp_grad(g_warp, im2, full, want_im=False)
    wantg = oracle_mod.resize_bilinear_grad(g_full + g[..., 6:8], h, w, mult=20.0)
    close(f.grad, wantg, 1e-5, 1e-5 * np.abs(wantg).max())


def test_resize_bilinear_vs_oracle(oracle_mod, dev):
    from unflow_amd import flownet
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 6, 9, 3)).astype(np.float32)
    for size in ((6, 9), (24, 36), (13, 20), (3, 4)):
        np.testing.assert_array_equal(N(flownet.resize_bilinear(T(x, dev), size)), oracle_mod.resize_bilinear(x, *size))


def test_graph_replay_matches_eager(dev):
    """The ABI is capturable (explicit stream, no allocation, no sync): a HIP graph of
    compute_losses fwd+bwd replays to the eager result on new input contents."""
    from unflow_amd import graph, losses
    rng = np.random.default_rng(77)
    B, H, W = 2, 32, 48
    im1, im2, flow = _pair(rng, B, H, W)
    tim1, tim2 = T(im1, dev), T(im2, dev)
    ffw = T(flow * 0.3, dev).requires_grad_(True)
    fbw = T(-flow * 0.3, dev).requires_grad_(True)

    def step():
        l = losses.compute_losses(tim1, tim2, ffw, fbw, None, "fb", 2, terms=["ternary", "smooth_2nd", "grad"])
        total = l["ternary"] + 3.0 * l["smooth_2nd"] + l["grad"]
        return (total,) + torch.autograd.grad(total, [ffw, fbw])
    g = graph.GraphedStep(step)
    with torch.no_grad():
        ffw.copy_(T(flow * 0.25 + 0.1, dev))       # new contents in the captured buffers
    want = [N(t) for t in step()]
    got = [N(t) for t in g.replay()]
    for a, b in zip(got, want):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("B,H,W,r,eps", [(2, 37, 53, 3, 0.9), (1, 8, 32, 1, 0.0), (1, 9, 70, 2, 2.5), (1, 1, 1, 3, 0.9)])
def test_integer_census_bit_exact(oracle_mod, dev, B, H, W, r, eps):
    """Integer ternary census + popcount Hamming distance (extension): bit-exact, ties included."""
    from unflow_amd import losses
    rng = np.random.default_rng(H * 7 + W)
    im1 = (rng.integers(0, 64, size=(B, H, W, 3)) / 63.0).astype(np.float32)   # quantised: exact ties occur
    im2 = np.clip(im1 + rng.integers(-2, 3, size=im1.shape) / 63.0, 0, 1).astype(np.float32)
    mask = (rng.uniform(size=(B, H, W, 1)) > 0.3).astype(np.float32)
    s1, s2 = losses.census_transform(T(im1, dev), r, eps), losses.census_transform(T(im2, dev), r, eps)
    w1, w2 = oracle_mod.census_transform(im1, r, eps), oracle_mod.census_transform(im2, r, eps)
    np.testing.assert_array_equal(N(s1).view(np.uint64), w1)
    np.testing.assert_array_equal(N(s2).view(np.uint64), w2)
    dist, total = losses.census_hamming(s1, s2, T(mask, dev))
    wd, wt = oracle_mod.census_hamming(w1, w2, mask)
    np.testing.assert_array_equal(N(dist), wd)
    assert int(total) == wt
    with pytest.raises(ValueError, match="max_distance"):
        losses.census_transform(T(im1, dev), 4)


@pytest.mark.parametrize("r", [1, 2, 3])
def test_ternary_loss_warped_gradient(oracle_mod, dev, r):
    """ternary_loss_warped == ternary_loss(image_warp(.)) in value and in d/d flow (fused backward for
    r = 2, 3), and the flow gradient matches the oracle's census gradient pushed through its warp gradient."""
    from unflow_amd import losses
    from unflow_amd.image_warp import image_warp
    rng = np.random.default_rng(60 + r)
    B, H, W = 2, 70, 61
    im1, im2, flow = _pair(rng, B, H, W)
    mask = (rng.uniform(size=(B, H, W, 1)) > 0.1).astype(np.float32)
    f1 = T(flow, dev).requires_grad_(True)
    l1 = losses.ternary_loss_warped(T(im1, dev), T(im2, dev), f1, T(mask, dev), max_distance=r)
    (l1 * 1.5).backward()
    f2 = T(flow, dev).requires_grad_(True)
    l2 = losses.ternary_loss(T(im1, dev), image_warp(T(im2, dev), f2), T(mask, dev), max_distance=r)
    (l2 * 1.5).backward()
    assert float(l1) == float(l2)
    close(f1.grad, N(f2.grad), 1e-5, 1e-6 * float(f2.grad.abs().max()))
    w = oracle_mod.image_warp(im2, flow)
    g_im = oracle_mod.ternary_loss_grad(im1, w, mask, r, gout=1.5)
    _, want = oracle_mod.image_warp_grad(g_im, im2, flow, want_im=False)
    close(f1.grad, want, 1e-5, 1e-5 * np.abs(want).max())


def test_fused_loss_glue_matches_composition(oracle_mod, dev):
    """fb_occlusion_warp == fb_occlusion(image_warp(...)) bitwise; sym_loss == the charbonnier
    composition; compute_losses without aux (fused path) == with aux (composed path) == oracle."""
    from unflow_amd import losses
    from unflow_amd.image_warp import image_warp
    from unflow_amd.ops import forward_warp
    rng = np.random.default_rng(55)
    B, H, W = 2, 40, 56
    im1, im2, flow_fw = _pair(rng, B, H, W)
    flow_bw = (-flow_fw + rng.standard_normal(flow_fw.shape) * 0.7).astype(np.float32)
    ffw, fbw = T(flow_fw, dev), T(flow_bw, dev)
    border = T(oracle_mod.create_border_mask((B, H, W)), dev)
    m1, m2 = border.clone(), border.clone()
    warped, occ1 = losses.fb_occlusion_warp(ffw, fbw, m1, want_warped=True)
    ref_w = image_warp(fbw, ffw)
    _, occ2 = losses.fb_occlusion(ffw, ref_w, m2)
    assert torch.equal(warped, ref_w) and torch.equal(occ1, occ2) and torch.equal(m1, m2)
    assert 0 < float(occ1.mean()) < 1
    dens = forward_warp(fbw)
    got = float(losses.sym_loss(m1, dens))
    want = float(losses.charbonnier_loss((1 - m1) - (dens < losses.DISOCC_THRESH).float()))
    np.testing.assert_allclose(got, want, rtol=1e-6)
    # 16 bytes per lane when the pixel count allows it, one float otherwise: the same sum either way
    for shape in ((1, 5, 7, 1), (3, 9, 13, 1), (1, 4, 9, 1)):
        m = torch.from_numpy((rng.random(shape) < 0.6).astype(np.float32)).to(dev)
        d = torch.from_numpy((rng.random(shape) * 0.5).astype(np.float32)).to(dev)
        want = float(losses.charbonnier_loss((1 - m) - (d < losses.DISOCC_THRESH).float()))
        np.testing.assert_allclose(float(losses.sym_loss(m, d)), want, rtol=1e-6, err_msg=str(shape))
        tail = m.reshape(-1)[1:].reshape(1, -1, 1, 1), d.reshape(-1)[1:].reshape(1, -1, 1, 1)   # 4-byte aligned views
        want = float(losses.charbonnier_loss((1 - tail[0]) - (tail[1] < losses.DISOCC_THRESH).float()))
        np.testing.assert_allclose(float(losses.sym_loss(*tail)), want, rtol=1e-6, err_msg="view of %s" % (shape,))
    for mo in ("fb", "disocc", ""):
        fused = losses.compute_losses(T(im1, dev), T(im2, dev), ffw, fbw, border, mo, 2,
                                      terms=("ternary", "sym", "occ", "smooth_2nd"))
        full, _ = losses.compute_losses(T(im1, dev), T(im2, dev), ffw, fbw, border, mo, 2, return_aux=True)
        want, _ = oracle_mod.compute_losses(im1, im2, flow_fw, flow_bw, N(border), mo, 2)
        assert set(fused) == {"ternary", "sym", "occ", "smooth_2nd"}
        for k in fused:
            np.testing.assert_allclose(float(fused[k]), float(full[k]), rtol=1e-6, err_msg=k)
            np.testing.assert_allclose(float(fused[k]), want[k], rtol=1e-5, err_msg=k)
    # occlusion() (losses.py:125-134): unwarped magnitudes in the threshold
    got_fw, got_bw = losses.occlusion(ffw, fbw)
    want_fw, want_bw = oracle_mod.occlusion(flow_fw, flow_bw)
    np.testing.assert_array_equal(N(got_fw), want_fw)
    np.testing.assert_array_equal(N(got_bw), want_bw)
    assert 0 < want_fw.mean() < 1
    only_smooth = losses.compute_losses(T(im1, dev), T(im2, dev), ffw, fbw, None, "", 1, terms=("smooth_1st",))
    assert set(only_smooth) == {"smooth_1st"}


def test_empty_inputs(dev):
    from unflow_amd import ops
    from unflow_amd.image_warp import image_warp
    assert ops.backward_warp(torch.zeros((0, 4, 4, 3), device=dev), torch.zeros((0, 4, 4, 2), device=dev)).shape == (0, 4, 4, 3)
    assert image_warp(torch.zeros((2, 0, 4, 3), device=dev), torch.zeros((2, 0, 4, 2), device=dev)).shape == (2, 0, 4, 3)
    assert ops.forward_warp(torch.zeros((0, 4, 4, 2), device=dev)).shape == (0, 4, 4, 1)
    assert ops.correlation(torch.zeros((0, 8, 4, 4), device=dev), torch.zeros((0, 8, 4, 4), device=dev)).shape == (0, 441, 4, 4)
    assert ops.downsample(torch.zeros((0, 4, 4, 3), device=dev), 2).shape == (0, 2, 2, 3)
    from unflow_amd import losses
    assert float(losses.smoothness_loss(torch.zeros((0, 4, 4, 2), device=dev))) == 0.0
    assert float(losses.gradient_loss(torch.zeros((0, 4, 4, 3), device=dev), torch.zeros((0, 4, 4, 3), device=dev),
                                      torch.zeros((0, 4, 4, 1), device=dev))) == 0.0


# ----------------------------------------------------------------------------- round-2 additions
@pytest.mark.parametrize("B,H,W,C,oh,ow", [(2, 5, 18, 3, 2, 9), (1, 7, 9, 1, 3, 4), (2, 25, 39, 3, 12, 19), (1, 3, 3, 2, 1, 1),
                                           (1, 96, 320, 3, 24, 80), (1, 9, 31, 1, 4, 15)])
def test_resize_area_vs_oracle(oracle_mod, dev, B, H, W, C, oh, ow):
    """util.downsample's odd-extent branch (core/util.py:28, tf.image.resize_area; assumption A3,
    PARITY UNPINNED by the reference).  Same operation order as the oracle: 1e-6."""
    from unflow_amd import util
    rng = np.random.default_rng(H * 10 + W)
    x = rng.uniform(size=(B, H, W, C)).astype(np.float32)
    got = util.resize_area(T(x, dev), (oh, ow))
    assert got.shape == (B, oh, ow, C)
    close(got, oracle_mod.resize_area(x, oh, ow), 1e-6, 1e-7)


def test_util_downsample_dispatch_on_gpu(oracle_mod, dev):
    from unflow_amd import ops, util
    rng = np.random.default_rng(12)
    even = rng.uniform(size=(2, 8, 12, 3)).astype(np.float32)
    assert torch.equal(util.downsample(T(even, dev), 2), ops.downsample(T(even, dev), 2))
    with pytest.raises(ValueError, match="divisible by scale"):
        util.downsample(T(np.zeros((1, 6, 8, 1), np.float32), dev), 4)        # even, not divisible: reaches the op
    odd = rng.uniform(size=(2, 5, 18, 3)).astype(np.float32)
    close(util.downsample(T(odd, dev), 2), oracle_mod.util_downsample(odd, 2), 1e-6, 1e-7)
    assert util.downsample(T(rng.uniform(size=(1, 9, 10, 1)).astype(np.float32), dev), 4).shape == (1, 2, 2, 1)
    assert util.resize_area(torch.zeros((0, 5, 7, 3), device=dev), (2, 3)).shape == (0, 2, 3, 3)


def test_pyramid_loss_odd_extents_vs_oracle(oracle_mod, dev):
    """A resolution whose pyramid levels turn odd (200x312 -> 50x78 -> 25x39 -> 12x19 -> 6x9 -> 3x4):
    levels below 25x39 go through resize_area on both sides."""
    from unflow_amd import unsupervised
    rng = np.random.default_rng(44)
    B, H, W = 1, 200, 312
    im1, im2, _ = _pair(rng, B, H, W)
    border = oracle_mod.create_border_mask((B, H, W))
    shapes = [(50, 78), (25, 39), (12, 19), (6, 9), (3, 4)]
    fws = [(rng.standard_normal((B, h, w, 2)) * 0.3).astype(np.float32) for h, w in shapes]
    bws = [(-f + rng.standard_normal(f.shape) * 0.05).astype(np.float32) for f in fws]
    params = dict(pyramid_loss=True, mask_occlusion="fb", border_mask=True, ternary_weight=1.0, smooth_2nd_weight=3.0,
                  fb_weight=0.2, occ_weight=12.4)
    want, want_terms = oracle_mod.pyramid_loss(im1, im2, border, fws, bws, params)
    tf_ = [T(f, dev).requires_grad_(True) for f in fws]
    tb_ = [T(f, dev).requires_grad_(True) for f in bws]
    got, got_terms = unsupervised.pyramid_loss(T(im1, dev), T(im2, dev), T(border, dev), tf_, tb_, params)
    np.testing.assert_allclose(float(got.detach()), want, rtol=1e-5)
    for k in ("ternary", "smooth_2nd", "fb", "occ"):
        np.testing.assert_allclose(float(got_terms[k]), want_terms[k], rtol=1e-5, err_msg=k)
    got.backward()
    assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in tf_ + tb_)


def test_correlation_flip_rejects_a_volume_of_other_attrs(dev):
    """A volume built with non-default attrs flipped with the defaults would be overrun (the
    launch is sized from the attrs): the shape is checked on the host AND in the C entry point."""
    import ctypes
    from unflow_amd import _lib, ops
    vol = torch.zeros((1, 81, 8, 8), device=dev)              # 9x9 displacements: max_displacement 8, stride_2 2
    with pytest.raises(ValueError, match="attrs imply"):
        ops.correlation_flip(vol)
    assert ops.correlation_flip(vol, max_displacement=8, pad=8).shape == vol.shape
    out = torch.empty_like(vol)
    rc = _lib.load().unflow_correlation_flip(ctypes.c_void_p(vol.data_ptr()), None, ctypes.c_void_p(out.data_ptr()),
                                             1, 81, 8, 8, 1, 20, 20, 1, 2, None)
    assert rc == _lib.UNFLOW_ERR_INVALID_ARGUMENT and "441" in _lib.last_error()


def test_ternary_warped_gives_the_composed_set_of_gradients(dev):
    """Which gradients exist must not depend on whether compute_losses takes the fused census path:
    an im2 that requires grad gets one (through image_warp) on both paths; im1 never does."""
    from unflow_amd import losses
    from unflow_amd.image_warp import image_warp
    rng = np.random.default_rng(73)
    B, H, W = 1, 48, 64
    im1, im2, flow = _pair(rng, B, H, W)
    mask = torch.ones((B, H, W, 1), device=dev)
    grads = []
    for fused_entry in (True, False):
        a, b = T(im1, dev).requires_grad_(True), T(im2, dev).requires_grad_(True)
        f = T(flow, dev).requires_grad_(True)
        if fused_entry:
            loss = losses.ternary_loss_warped(a, b, f, mask, max_distance=3)
        else:
            loss = losses.ternary_loss(a, image_warp(b, f), mask, max_distance=3)
        loss.backward()
        assert a.grad is None and b.grad is not None and f.grad is not None
        grads.append((b.grad, f.grad))
    # the image gradient is a scatter-add (float atomics, order varies): equal up to rounding
    close(grads[0][0], N(grads[1][0]), 1e-6, 1e-6 * float(grads[1][0].abs().max()))
    assert torch.equal(grads[0][1], grads[1][1])
    # without a gradient request on im2 the fused kernels run and agree with the composition
    f1 = T(flow, dev).requires_grad_(True)
    l1 = losses.ternary_loss_warped(T(im1, dev), T(im2, dev), f1, mask, max_distance=3)
    l1.backward()
    assert float(l1) == float(loss.detach())
    close(f1.grad, N(grads[1][1]), 1e-5, 1e-6 * float(grads[1][1].abs().max()))


def test_stream_copy_probe(dev):
    import ctypes
    from unflow_amd import _lib
    a = torch.randn(1 << 20, device=dev)
    b = torch.zeros_like(a)
    _lib.check(_lib.load().unflow_stream_copy(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                              ctypes.c_size_t(a.numel() * 4 - 16), None))
    torch.cuda.synchronize()
    assert torch.equal(a[:-4], b[:-4]) and float(b[-4:].abs().max()) == 0.0


def _offset_by_one_float(t):
    """The same values at an address that is 4-byte but not 8- or 16-byte aligned (a view one float into a
    larger allocation; torch allocations themselves are 512-byte aligned)."""
    big = torch.empty(t.numel() + 1, device=t.device, dtype=t.dtype)
    view = big[1:].view(t.shape)
    view.copy_(t)
    assert view.data_ptr() % 8 == 4 and view.is_contiguous()
    return view


def test_unaligned_tensors_take_the_same_results(dev):
    """Inputs that are only 4-byte aligned (a slice of a larger buffer): the vectorised paths -- LDS-DMA and
    float4 stores of the correlation, the vector row loads of downsample, the float4 output of the stacked
    block -- are gated on alignment and fall back; results are the same as on aligned tensors."""
    from unflow_amd import flownet, losses, ops
    from unflow_amd.image_warp import image_warp
    g = torch.Generator(device=dev).manual_seed(3)
    f0 = torch.randn((2, 128, 12, 16), device=dev, generator=g)
    f1 = torch.randn((2, 128, 12, 16), device=dev, generator=g)
    top = torch.randn((2, 441, 12, 16), device=dev, generator=g)
    c = ops.correlation(f0, f1)
    g0, g1 = ops.correlation_grad(top, f0, f1)
    u0, u1, utop = _offset_by_one_float(f0), _offset_by_one_float(f1), _offset_by_one_float(top)
    torch.testing.assert_close(ops.correlation(u0, u1), c, rtol=1e-5, atol=1e-6)      # generic kernels: another summation order
    h0, h1 = ops.correlation_grad(utop, u0, u1)
    torch.testing.assert_close(h0, g0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(h1, g1, rtol=1e-5, atol=1e-6)
    im = torch.rand((2, 20, 64, 3), device=dev, generator=g)
    flow = torch.randn((2, 20, 64, 2), device=dev, generator=g) * 4
    uim, uflow = _offset_by_one_float(im), _offset_by_one_float(flow)
    assert torch.equal(image_warp(uim, uflow), image_warp(im, flow))
    assert torch.equal(ops.backward_warp(uim, uflow), ops.backward_warp(im, flow))
    assert torch.equal(ops.downsample(uim, 2), ops.downsample(im, 2))
    torch.testing.assert_close(ops.forward_warp(uflow), ops.forward_warp(flow), rtol=1e-6, atol=1e-6)
    mask = (torch.rand((2, 20, 64, 1), device=dev, generator=g) > 0.2).float()
    im2 = torch.rand((2, 20, 64, 3), device=dev, generator=g)
    a = losses.ternary_loss(uim, _offset_by_one_float(im2), _offset_by_one_float(mask), max_distance=3)
    b = losses.ternary_loss(im, im2, mask, max_distance=3)
    torch.testing.assert_close(a, b, rtol=1e-6, atol=0)
    lr = torch.randn((2, 5, 16, 2), device=dev, generator=g)
    assert torch.equal(flownet.stacked_inputs(uim, _offset_by_one_float(im2), _offset_by_one_float(lr)),
                       flownet.stacked_inputs(im, im2, lr))


def test_deferred_reductions_match_immediate(dev):
    """unflow_reduce_plan_*: the loss forwards of a deferred block leave their partial sums behind and ONE launch
    reduces them -- the same fixed-order double accumulation as the immediate one-block reduction, so every scalar
    is bit-identical; the fold gives the weighted total and the per-term sums."""
    from unflow_amd import losses
    rng = np.random.default_rng(91)
    B, H, W = 2, 40, 56
    im1, im2, flow = _pair(rng, B, H, W)
    tim1, tim2, tflow = T(im1, dev), T(im2, dev), T(flow, dev)
    mask = (torch.rand((B, H, W, 1), device=dev) > 0.2).float()
    dens = torch.rand((B, H, W, 1), device=dev) * 2

    def run():
        return [losses.ternary_loss(tim1, tim2, mask, 1), losses.ternary_loss(tim1, tim2, mask, 3),
                losses.ternary_loss_warped(tim1, tim2, tflow, mask, 2), losses.charbonnier_loss(tim1 - tim2, mask, beta=255),
                losses.smoothness_loss(tflow), losses.second_order_loss(tflow), losses.gradient_loss(tim1, tim2, mask),
                losses.sym_loss(mask, dens), losses.occ_loss(mask),
                losses.charbonnier_loss(torch.zeros((0, 4, 4, 2), device=dev))]          # an empty tensor: an empty segment
    want = [float(v) for v in run()]
    with losses.DeferredReductions(dev) as d:
        got = run()
        assert d.plan.n == len(want)
    coeff = [0.5 + i for i in range(len(want))]
    term = [i % 8 for i in range(len(want))]
    total = torch.empty((), device=dev)
    per_term = torch.empty(8, device=dev)
    d.fold(coeff, term, [2.0] * len(want), total, per_term)
    assert [float(v) for v in got] == want                      # bit-identical scalars
    np.testing.assert_allclose(float(total), sum(c * w for c, w in zip(coeff, want)), rtol=1e-6)
    for t in range(8):
        np.testing.assert_allclose(float(per_term[t]), sum(2.0 * w for i, w in enumerate(want) if i % 8 == t), rtol=1e-6)
    assert losses.DeferredReductions.current() is None
    assert float(losses.charbonnier_loss(tim1 - tim2, mask, beta=255)) == want[3]        # the immediate form is back
    with pytest.raises(RuntimeError, match="nest"):
        with losses.DeferredReductions(dev):
            with losses.DeferredReductions(dev):
                pass
    assert losses.DeferredReductions.current() is None


def test_deferred_reductions_are_per_thread_and_a_null_scalar_is_an_error(dev):
    """The C plan is thread_local and so is the Python context: a loss forward on ANOTHER thread while a deferred block
    is open runs in immediate mode (its own scalar, right away), and the C entry points refuse a NULL output scalar when
    no plan is open on the calling thread instead of writing through it."""
    import threading
    from unflow_amd import _lib, losses
    from unflow_amd.losses import _ptr, _stream, _workspace
    x = torch.rand((1, 8, 8, 2), device=dev)
    want = float(losses.charbonnier_loss(x))
    seen = {}

    def other():
        with torch.cuda.device(dev):
            seen["current"] = losses.DeferredReductions.current()
            seen["value"] = float(losses.charbonnier_loss(x))
    with losses.DeferredReductions(dev) as d:
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert d.plan.n == 0                                   # nothing of the other thread landed in this plan
    assert seen["current"] is None and seen["value"] == want
    ws = _workspace(1, 8, 8, dev)
    with torch.cuda.device(dev):
        rc = _lib.load().unflow_charbonnier_fwd(_ptr(x), None, 0, None, _ptr(ws), 1, 8, 8, 2, 0.45, 1.0, 0.001, 0.0, 0, _stream())
    assert rc == _lib.UNFLOW_ERR_INVALID_ARGUMENT
    assert b"NULL" in _lib.load().unflow_last_error()
    torch.cuda.synchronize()


def test_fan_out_sums_the_consumers_gradients_in_one_launch(dev):
    """losses._fan_out: n aliases of a tensor whose gradients are summed by unflow_sum_tensors (same order as a chain
    of adds); unused aliases contribute nothing."""
    from unflow_amd import losses
    x = torch.randn((3, 5, 7, 2), device=dev, requires_grad=True)
    a = losses._fan_out(x, 6)
    ws = [torch.randn_like(x) for _ in range(4)]
    (a[0] * ws[0] + a[2] * ws[1] + a[3] * ws[2] + a[5] * ws[3]).sum().backward()
    want = ((ws[0] + ws[1]) + ws[2]) + ws[3]
    assert torch.equal(x.grad, want)
    y = torch.randn((2, 3), device=dev, requires_grad=True)
    b = losses._fan_out(y, 3)
    (b[1] * 2.0).sum().backward()
    assert torch.equal(y.grad, torch.full_like(y, 2.0))
    assert losses._fan_out(y.detach(), 4)[2] is not None and losses._fan_out(y.detach(), 4)[0].data_ptr() == y.data_ptr()
