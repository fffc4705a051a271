"""Parity at BASELINE.json's full sizes (384x1280 pairs, FlowNetC features
[B,256,48,160]) through size-independent properties -- the CPU oracle is too slow
to replay these sizes in a test -- plus an oracle spot check on one full-size
sample.  All through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

H, W, C8, H8, W8 = 384, 1280, 256, 48, 160


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    return torch.device("cuda:0")


def gen(dev, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return g


def test_correlation_properties_cfg2(dev):
    """cfg2: B=8, [8,256,48,160].  Linearity in each argument, the zero-displacement
    plane equals mean_c(f0*f1), and a shifted copy peaks on the matching plane."""
    from unflow_amd import ops
    g = gen(dev, 1)
    f0 = torch.randn((8, C8, H8, W8), device=dev, generator=g)
    f1 = torch.randn((8, C8, H8, W8), device=dev, generator=g)
    c = ops.correlation(f0, f1)
    assert c.shape == (8, 441, H8, W8)
    # bilinear: exact for power-of-two scales
    assert torch.equal(ops.correlation(2 * f0, f1), 2 * c)
    assert torch.equal(ops.correlation(f0, 0.5 * f1), 0.5 * c)
    f2 = torch.randn(f1.shape, device=dev, generator=g)
    torch.testing.assert_close(ops.correlation(f0, f1 + f2), c + ops.correlation(f0, f2), rtol=1e-5, atol=1e-5)
    # centre plane (dy=dx=0) is the channel mean of the product
    torch.testing.assert_close(c[:, 220], (f0.double() * f1.double()).mean(1).float(), rtol=1e-5, atol=1e-5)
    # f1 = f0 shifted by (dy, dx) = (+4, -6): plane of displacement (4,-6) holds mean(f0^2) where defined
    dy, dx = 4, -6
    shifted = torch.zeros_like(f0)
    shifted[:, :, dy:, :W8 + dx] = f0[:, :, :H8 - dy, -dx:]
    cs = ops.correlation(f0, shifted)
    plane = (dy // 2 + 10) * 21 + (dx // 2 + 10)
    want = (f0.double() ** 2).mean(1).float()
    torch.testing.assert_close(cs[:, plane, :H8 - dy, -dx:], want[:, :H8 - dy, -dx:], rtol=1e-5, atol=1e-5)


def test_correlation_gradient_adjoint_cfg2(dev):
    """<corr(f0,f1), top> differentiates to the kernels' gradients: the adjoint
    identity <top, corr(df0, f1)> == <g0, df0> holds at full size."""
    from unflow_amd import ops
    g = gen(dev, 2)
    f0 = torch.randn((4, C8, H8, W8), device=dev, generator=g)
    f1 = torch.randn((4, C8, H8, W8), device=dev, generator=g)
    top = torch.randn((4, 441, H8, W8), device=dev, generator=g)
    g0, g1 = ops.correlation_grad(top, f0, f1)
    d0 = torch.randn(f0.shape, device=dev, generator=g)      # seeded: the check must not depend on the run
    d1 = torch.randn(f1.shape, device=dev, generator=g)
    lhs0 = (ops.correlation(d0, f1).double() * top.double()).sum()
    lhs1 = (ops.correlation(f0, d1).double() * top.double()).sum()
    # both sides are sums of ~8e6 signed terms: compare on the scale |g|*|d| (a relative
    # tolerance on the sum itself fails whenever the random sum happens to land near zero)
    for lhs, gg, dd in ((lhs0, g0, d0), (lhs1, g1, d1)):
        rhs = (gg.double() * dd.double()).sum()
        scale = float(gg.double().norm() * dd.double().norm())
        assert abs(float(rhs) - float(lhs)) <= 1e-6 * scale, (float(rhs), float(lhs), scale)
        assert abs(float(rhs)) > 0


def test_correlation_is_deterministic_cfg2(dev):
    """The MFMA kernels stage their operands by LDS-DMA behind hand-written barriers: a barrier that does not
    wait for the DMA is a race that shows as run-to-run differences at full size (one such was found in
    round 2).  Forward and both gradients, repeated with allocator churn in between, must agree bitwise."""
    from unflow_amd import ops
    g = gen(dev, 21)
    f0 = torch.randn((8, C8, H8, W8), device=dev, generator=g)
    f1 = torch.randn((8, C8, H8, W8), device=dev, generator=g)
    top = torch.randn((8, 441, H8, W8), device=dev, generator=g)
    c0 = ops.correlation(f0, f1)
    g0, g1 = ops.correlation_grad(top, f0, f1)
    for i in range(4):
        junk = torch.full((16 * 1024 * 1024,), float(i), device=dev)
        del junk
        assert torch.equal(ops.correlation(f0, f1), c0)
        a, b = ops.correlation_grad(top, f0, f1)
        assert torch.equal(a, g0) and torch.equal(b, g1)


def test_correlation_gradients_do_not_depend_on_the_item_schedule(dev):
    """A persistent workgroup of the backward walks several items at B = 32 and hands the last tiles of one item to the
    first rows of the next one's sweep (deferred end tiles, correlation_bwd4.hip); at B = 2 every workgroup has one item
    and ends it synchronously.  Both paths add the same products in the same order: the gradients of two pairs must
    not change by a bit when the batch holds sixteen copies of them -- nor may the forward."""
    from unflow_amd import ops
    g = gen(dev, 77)
    f0 = torch.randn((2, C8, H8, W8), device=dev, generator=g)
    f1 = torch.randn((2, C8, H8, W8), device=dev, generator=g)
    top = torch.randn((2, 441, H8, W8), device=dev, generator=g)
    c_small = ops.correlation(f0, f1)
    g0_small, g1_small = ops.correlation_grad(top, f0, f1)
    rep = lambda t: t.repeat(16, 1, 1, 1).contiguous()
    c_big = ops.correlation(rep(f0), rep(f1))
    g0_big, g1_big = ops.correlation_grad(rep(top), rep(f0), rep(f1))
    for small, big, name in ((c_small, c_big, "forward"), (g0_small, g0_big, "gradient 0"), (g1_small, g1_big, "gradient 1")):
        assert torch.equal(big, rep(small)), name
    assert float(g0_small.abs().max()) > 0 and float(g1_small.abs().max()) > 0


def test_correlation_full_size_sample_vs_oracle(oracle_mod, dev):
    """One full-size pair against the CPU oracle (about 3 s of CPU)."""
    from unflow_amd import ops
    rng = np.random.default_rng(3)
    f0 = rng.standard_normal((1, C8, H8, W8)).astype(np.float32)
    f1 = rng.standard_normal((1, C8, H8, W8)).astype(np.float32)
    want, p0, p1 = oracle_mod.correlation(f0, f1)
    a, b = torch.from_numpy(f0).to(dev), torch.from_numpy(f1).to(dev)
    np.testing.assert_allclose(ops.correlation(a, b).cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    top = rng.standard_normal(want.shape).astype(np.float32)
    g0, g1 = ops.correlation_grad(torch.from_numpy(top).to(dev), a, b)
    w0, w1 = oracle_mod.correlation_grad(top, f0, f1, p0, p1)
    np.testing.assert_allclose(g0.cpu().numpy(), w0, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(g1.cpu().numpy(), w1, rtol=1e-5, atol=1e-5)


def test_warps_integer_shift_cfg2(dev):
    """An integer flow is a pure shift: exact for both warp semantics; the two differ
    only where taps leave the image (zero vs clamp)."""
    from unflow_amd import ops
    from unflow_amd.image_warp import image_warp
    g = gen(dev, 4)
    im = torch.rand((8, H, W, 3), device=dev, generator=g)
    flow = torch.zeros((8, H, W, 2), device=dev)
    flow[..., 0], flow[..., 1] = 5.0, -3.0      # sample at (x+5, y-3)
    bw, iw = ops.backward_warp(im, flow), image_warp(im, flow)
    assert torch.equal(bw[:, 3:, :W - 5], im[:, :H - 3, 5:])
    assert torch.equal(iw[:, 3:, :W - 5], im[:, :H - 3, 5:])
    assert float(bw[:, :3].abs().max()) == 0.0 and float(bw[:, :, W - 5:].abs().max()) == 0.0
    assert torch.equal(iw[:, :3, :W - 5], im[:, :1, 5:].expand(-1, 3, -1, -1))      # clamped rows repeat row 0
    assert torch.equal(image_warp(im, torch.zeros_like(flow)), im)                   # idempotent at zero flow


def test_forward_warp_mass_cfg3(dev):
    """Every source pixel deposits (sum_d exp(-d^2/2))^2 over its window when it is
    away from the border; a global translation keeps the interior density at 2*pi."""
    from unflow_amd import ops
    flow = torch.zeros((32, H, W, 2), device=dev)
    flow[..., 0], flow[..., 1] = 7.25, -4.5
    out = ops.forward_warp(flow)
    s = sum(np.exp(-(d - 0.25) ** 2 / 2.0) for d in range(-4, 5)) * sum(np.exp(-(d - 0.5) ** 2 / 2.0) for d in range(-4, 5))
    interior = out[:, 20:H - 20, 20:W - 20, 0]
    np.testing.assert_allclose(float(interior.min()), s, rtol=2e-6)
    np.testing.assert_allclose(float(interior.max()), s, rtol=2e-6)
    # the in-window sums are integer (order independent); only cells fed by several tiles see
    # float atomics in varying order: two launches agree to the last bits
    torch.testing.assert_close(ops.forward_warp(flow), out, rtol=1e-6, atol=1e-6)


def test_census_properties_cfg3(dev):
    """Identical images: distance 0 everywhere, loss = eps^(2*alpha) * mean(mask');
    the loss is additive over a split of the batch when normalised globally."""
    from unflow_amd import losses
    g = gen(dev, 5)
    im1 = torch.rand((32, H, W, 3), device=dev, generator=g)
    im2 = torch.rand((32, H, W, 3), device=dev, generator=g)
    mask = losses.create_border_mask(im1, 0.1)
    l0 = losses.ternary_loss(im1, im1, mask, max_distance=3)
    inner = float(mask.mean())        # border (39) > census border (3)
    np.testing.assert_allclose(float(l0), (0.001 ** 2) ** 0.45 * inner, rtol=1e-5)
    full = losses.ternary_loss(im1, im2, mask, max_distance=3)
    npx = 32 * H * W
    parts = [losses.ternary_loss(im1[i:i + 8], im2[i:i + 8], mask[i:i + 8], max_distance=3, norm_count=npx)
             for i in range(0, 32, 8)]
    np.testing.assert_allclose(float(sum(parts)), float(full), rtol=1e-5)
    # dist_sum bounds: 0 <= D <= P*P
    d = losses.ternary_dist(im1[:4], im2[:4], 3)
    assert float(d.min()) >= 0.0 and float(d.max()) <= 49.0


def test_cfg5_shapes_fp32(dev):
    """BASELINE config 5's shapes (768x1280, batch 16, features [16,256,96,160]) in fp32 and,
    for the correlation, with fp16 storage (extension; the reference is float32 only).
    Correlation: adjoint identity and zero-displacement plane; forward_warp: splat mass;
    stacked-net block: zero flow reproduces [im1, im2, 0, im2, |im2 - im1|]."""
    from unflow_amd import flownet, ops
    g = gen(dev, 9)
    f0 = torch.randn((16, C8, 96, 160), device=dev, generator=g)
    f1 = torch.randn((16, C8, 96, 160), device=dev, generator=g)
    c = ops.correlation(f0, f1)
    assert c.shape == (16, 441, 96, 160)
    torch.testing.assert_close(c[:, 220], (f0.double() * f1.double()).mean(1).float(), rtol=1e-5, atol=1e-5)
    top = torch.randn(c.shape, device=dev, generator=g)
    g0, g1 = ops.correlation_grad(top, f0, f1)
    df = torch.randn(f0.shape, device=dev, generator=g)
    for gg, corr in ((g0, lambda: ops.correlation(df, f1)), (g1, lambda: ops.correlation(f0, df))):
        lhs = (top.double() * corr().double()).sum()
        rhs = (gg.double() * df.double()).sum()
        scale = float(gg.double().norm() * df.double().norm())
        assert abs(float(lhs) - float(rhs)) <= 1e-6 * scale, (float(lhs), float(rhs), scale)
    # fp16 storage at the same shape: equals the f32 kernel on the widened inputs up to the final rounding
    h0, h1 = f0.half(), f1.half()
    ref = ops.correlation(h0.float(), h1.float()).half()
    ch = ops.correlation_fp16_exact(h0, h1)
    assert (ch == ref).float().mean() > 0.99
    torch.testing.assert_close(ch.float(), ref.float(), rtol=2e-3, atol=1e-6)
    # the f16 matrix-core form: same products, f32 sums in another order (rtol 2e-3), forward and
    # both gradients against the float32 kernels on the widened tensors
    ch = ops.correlation_fp16(h0, h1)
    torch.testing.assert_close(ch.float(), ref.float(), rtol=2e-3, atol=1e-6)
    th = top.half()
    gh0, gh1 = ops.correlation_fp16_grad(th, h0, h1)
    r0, r1 = ops.correlation_grad(th.float(), h0.float(), h1.float())
    for gg, rr in ((gh0, r0), (gh1, r1)):
        torch.testing.assert_close(gg.float(), rr.half().float(), rtol=2e-3, atol=2e-4 * float(rr.abs().max()))
    del c, top, g0, g1, df, f0, f1, h0, h1, ch, ref, th, gh0, gh1, r0, r1
    flow = torch.zeros((16, 768, 1280, 2), device=dev)
    out = ops.forward_warp(flow)
    np.testing.assert_allclose(float(out[:, 10:-10, 10:-10].min()), 6.2831479, rtol=1e-6)
    np.testing.assert_allclose(float(out[:, 10:-10, 10:-10].max()), 6.2831479, rtol=1e-6)
    im1 = torch.rand((16, 768, 1280, 3), device=dev, generator=g)
    im2 = torch.rand((16, 768, 1280, 3), device=dev, generator=g)
    blk = flownet.stacked_inputs(im1, im2, torch.zeros((16, 192, 320, 2), device=dev))
    assert torch.equal(blk, torch.cat([im1, im2, torch.zeros_like(flow), im2, (im2 - im1).abs()], 3))
    del blk


def test_cfg5_warps_fp16_storage(dev):
    """BASELINE config 5's full-resolution warps with fp16 storage (extension): image_warp of half images
    with half flows equals the f32 op on the widened tensors rounded to half, bit for bit, forward and flow
    gradient; zero flow is the identity; forward_warp of half flows equals the f32 op on the widened flows."""
    from unflow_amd import ops
    from unflow_amd.image_warp import image_warp
    g = gen(dev, 10)
    B, H, W = 16, 768, 1280
    im = torch.rand((B, H, W, 3), device=dev, generator=g).half()
    flow = (torch.randn((B, H, W, 2), device=dev, generator=g) * 8).half()
    assert torch.equal(ops.image_warp_fp16(im, torch.zeros_like(flow)), im)
    f16 = flow.clone().requires_grad_(True)
    out = ops.image_warp_fp16(im, f16)
    f32 = flow.float().requires_grad_(True)
    ref = image_warp(im.float(), f32)
    assert torch.equal(out, ref.half())
    gout = torch.randn(out.shape, device=dev, generator=g).half()
    out.backward(gout)
    ref.backward(gout.float())
    assert torch.equal(f16.grad, f32.grad.half())
    del out, ref, gout, f16, f32, im
    d16 = ops.forward_warp_fp16(flow)
    d32 = ops.forward_warp(flow.float())
    torch.testing.assert_close(d16, d32, rtol=1e-5, atol=1e-5)
    total = float(d32.double().sum())
    assert abs(float(d16.double().sum()) - total) <= 1e-6 * total


def test_bench_loss_leg_cfg3(oracle_mod, dev):
    """cfg3, the path bench.py times: exactly ``bench.Workload.loss_leg`` at 32x384x1280 --
    ``compute_losses(..., 'fb', 3, terms=('ternary','sym'))`` through the warp-fused pair-symmetric
    census kernels and the warp-fused occlusion kernel, forward and backward to both flows.
    (i) against the composed HIP path (image_warp -> fb_occlusion -> ternary_loss / forward_warp ->
    charbonnier) on the whole batch: loss 1e-6, flow gradients 1e-5 of their scale;
    (ii) one pair against the CPU oracle: scalars 1e-5, flow gradients 1e-5 of their scale."""
    import bench
    from unflow_amd import losses
    from unflow_amd.image_warp import image_warp
    from unflow_amd.ops import forward_warp
    B = 32
    wl = bench.Workload(B, dev, seed=1234, global_batch=B, features=False)
    ls = wl.loss_leg()
    fused = float(ls["ternary"].detach())
    g_fw, g_bw = wl.flow_fw.grad.clone(), wl.flow_bw.grad.clone()
    assert np.isfinite(fused) and torch.isfinite(g_fw).all() and torch.isfinite(g_bw).all()

    # (i) the same quantities through the unfused ops
    npx = B * H * W
    ffw = wl.flow_fw.detach().clone().requires_grad_(True)
    fbw = wl.flow_bw.detach().clone().requires_grad_(True)
    m_fw, m_bw = wl.border.clone(), wl.border.clone()
    losses.fb_occlusion(ffw, image_warp(fbw.detach(), ffw.detach()), m_fw)
    losses.fb_occlusion(fbw, image_warp(ffw.detach(), fbw.detach()), m_bw)
    composed = (losses.ternary_loss(wl.im1, image_warp(wl.im2, ffw), m_fw, 3, norm_count=npx) +
                losses.ternary_loss(wl.im2, image_warp(wl.im1, fbw), m_bw, 3, norm_count=npx))
    composed.backward()
    np.testing.assert_allclose(fused, float(composed.detach()), rtol=1e-6)
    assert float(wl.composed_loss()) == float(composed.detach())           # what bench.py asserts in every run
    for got, want in ((g_fw, ffw.grad), (g_bw, fbw.grad)):
        scale = float(want.abs().max())
        assert scale > 0
        assert float((got - want).abs().max()) <= 1e-5 * scale
    sym = (losses.charbonnier_loss((1 - m_fw) - (forward_warp(fbw.detach()) < losses.DISOCC_THRESH).float(), norm_count=npx) +
           losses.charbonnier_loss((1 - m_bw) - (forward_warp(ffw.detach()) < losses.DISOCC_THRESH).float(), norm_count=npx))
    np.testing.assert_allclose(float(ls["sym"]), float(sym), rtol=1e-6)
    del ffw, fbw, composed, m_fw, m_bw

    # (ii) pair 5 alone against the oracle (about a second of CPU)
    i = 5
    host = lambda t: t[i:i + 1].detach().cpu().numpy()
    im1, im2, f_fw, f_bw, border = (host(t) for t in (wl.im1, wl.im2, wl.flow_fw, wl.flow_bw, wl.border))
    want, waux = oracle_mod.compute_losses(im1, im2, f_fw, f_bw, border, "fb", 3, terms=("ternary", "sym"))
    tfw = wl.flow_fw[i:i + 1].detach().clone().requires_grad_(True)
    tbw = wl.flow_bw[i:i + 1].detach().clone().requires_grad_(True)
    got = losses.compute_losses(wl.im1[i:i + 1], wl.im2[i:i + 1], tfw, tbw, border_mask=wl.border[i:i + 1],
                                mask_occlusion="fb", data_max_distance=3, terms=("ternary", "sym"))
    # 491 520 pixels: a threshold pixel that flips moves the scalar by 2e-6, so the scalars hold 1e-5
    np.testing.assert_allclose(float(got["ternary"].detach()), want["ternary"], rtol=1e-5)
    np.testing.assert_allclose(float(got["sym"].detach()), want["sym"], rtol=1e-5)
    got["ternary"].backward()
    gim2 = oracle_mod.ternary_loss_grad(im1, waux["im2_warped"], waux["mask_fw"], 3)
    _, want_dfw = oracle_mod.image_warp_grad(gim2, im2, f_fw, want_im=False)
    gim1 = oracle_mod.ternary_loss_grad(im2, waux["im1_warped"], waux["mask_bw"], 3)
    _, want_dbw = oracle_mod.image_warp_grad(gim1, im1, f_bw, want_im=False)
    for got_g, want_g in ((tfw.grad, want_dfw), (tbw.grad, want_dbw)):
        np.testing.assert_allclose(got_g.cpu().numpy(), want_g, rtol=1e-5, atol=1e-5 * np.abs(want_g).max())


def test_bench_kitti_pyramid_step_full_size_vs_oracle(oracle_mod, dev):
    """bench.py's `kitti_config_step` at the size it is timed at (384x1280; core/unsupervised.py:85-145 with
    config_template/config.ini:123,166-174's mix): one pair of that step -- the same images, border mask and
    five-level flow pyramid -- against oracle.pyramid_loss / pyramid_loss_grad: combined loss and every term 1e-5,
    every level's flow gradients (both directions) 1e-5 of their scale.  A few seconds of CPU."""
    import bench
    from unflow_amd import unsupervised
    B, pair = 2, 1
    wl = bench.Workload(B, dev, seed=1234, global_batch=B, features=False)
    pf, pb = bench.kitti_pyramid_flows(torch, dev, B)
    sl = slice(pair, pair + 1)
    tf_ = [f.detach()[sl].clone().requires_grad_(True) for f in pf]
    tb_ = [f.detach()[sl].clone().requires_grad_(True) for f in pb]
    got, got_terms = unsupervised.pyramid_loss(wl.im1[sl], wl.im2[sl], wl.border[sl], tf_, tb_, bench.KITTI)
    got.backward()
    host = lambda t: t.detach().cpu().numpy()
    args = (host(wl.im1[sl]), host(wl.im2[sl]), host(wl.border[sl]), [host(f) for f in tf_], [host(f) for f in tb_], dict(bench.KITTI))
    want, want_terms = oracle_mod.pyramid_loss(*args)
    np.testing.assert_allclose(float(got.detach()), want, rtol=1e-5)
    for k in ("ternary", "smooth_2nd", "fb", "occ"):
        np.testing.assert_allclose(float(got_terms[k]), want_terms[k], rtol=1e-5, atol=1e-12, err_msg=k)
    want_fw, want_bw = oracle_mod.pyramid_loss_grad(*args)
    for i in range(5):
        for t, w in ((tf_[i], want_fw[i]), (tb_[i], want_bw[i])):
            assert np.abs(host(t.grad) - w).max() <= 1e-5 * np.abs(w).max(), "level %d" % i


def test_batch_shards_sum_to_whole_batch_cfg3(dev):
    """The N-GPU partition on one device: compute_losses on contiguous batch shards with
    norm_batch = global batch sums to the whole-batch losses, and each shard's flow gradients are
    the corresponding slices of the whole-batch gradients (pairs are independent; only the
    normalisation is global)."""
    import bench
    from unflow_amd import dist as udist
    from unflow_amd import losses
    B, world = 8, 3                                  # ragged: shards of 3, 3, 2 pairs
    wl = bench.Workload(B, dev, seed=77, global_batch=B, features=False)
    terms = ("ternary", "sym", "occ", "fb", "smooth_2nd")

    def run(sl, norm_batch):
        ffw = wl.flow_fw[sl].detach().clone().requires_grad_(True)
        fbw = wl.flow_bw[sl].detach().clone().requires_grad_(True)
        ls = losses.compute_losses(wl.im1[sl], wl.im2[sl], ffw, fbw, border_mask=wl.border[sl], mask_occlusion="fb",
                                   data_max_distance=3, norm_batch=norm_batch, terms=terms)
        (ls["ternary"] + 3.0 * ls["smooth_2nd"] + 0.2 * ls["fb"] + 12.4 * ls["occ"]).backward()
        return {k: float(v.detach()) for k, v in ls.items()}, ffw.grad, fbw.grad
    whole, g_fw, g_bw = run(slice(0, B), 0)
    total = {k: 0.0 for k in terms}
    for r in range(world):
        s, e = udist.shard_range(B, r, world)
        part, p_fw, p_bw = run(slice(s, e), B)
        for k in terms:
            total[k] += part[k]
        for got, want in ((p_fw, g_fw[s:e]), (p_bw, g_bw[s:e])):
            assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())
    for k in terms:
        np.testing.assert_allclose(total[k], whole[k], rtol=1e-6, err_msg=k)


@pytest.mark.parametrize("C", [3, 2])
def test_image_warp_image_gradient_fixed_point_full_size(oracle_mod, dev, C):
    """The scatter-add of image_warp's image gradient accumulates in two-word fixed point with a
    per-tile scale.  One full-size item with the bench's smooth flow and an upstream
    gradient whose magnitude spans nine decades (and a few exact zeros), against the oracle's float
    adds: 1e-5 of the gradient's scale overall, and -- the point of the second word -- 1e-5 of the LOCAL
    scale inside every 64 x 64 block, so small gradients next to large ones keep their digits."""
    import bench
    from unflow_amd.image_warp import image_warp_grad
    wl = bench.Workload(1, dev, seed=4321, global_batch=1, features=False)
    rng = np.random.default_rng(C)
    mag = 10.0 ** rng.uniform(-6, 3, size=(1, H // 64, W // 64, 1)).astype(np.float32)       # one decade per 64x64 block
    g = (rng.standard_normal((1, H, W, C)).astype(np.float32) * np.repeat(np.repeat(mag, 64, 1), 64, 2))
    g[0, 100:110, 200:260] = 0.0
    im = wl.im2[..., :C].contiguous()
    flow = wl.flow_fw.detach()
    got, _ = image_warp_grad(torch.from_numpy(g).to(dev), im, flow, True, False)
    want, _ = oracle_mod.image_warp_grad(g, im.cpu().numpy(), flow.cpu().numpy(), want_flow=False)
    got = got.cpu().numpy()
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    err = np.abs(got - want).reshape(H // 64, 64, W // 64, 64, C).max(axis=(1, 3, 4))
    ref = np.abs(want).reshape(H // 64, 64, W // 64, 64, C).max(axis=(1, 3, 4))
    # a block's taps reach a few pixels into its neighbours (flow up to ~30 px): compare with the
    # largest scale among the block and its 8 neighbours
    pad = np.pad(ref, 1, mode="edge")
    local = np.max([pad[1 + dy:1 + dy + ref.shape[0], 1 + dx:1 + dx + ref.shape[1]] for dy in (-1, 0, 1) for dx in (-1, 0, 1)], axis=0)
    assert (err <= 1e-5 * local).all(), float((err / local).max())
