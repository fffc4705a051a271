"""Writes the golden fixtures used by the parity tests.

Two files:

``reference_kat.json``  -- the known-answer vectors the reference's OWN tests hold
    for this path (inputs and expected outputs, transcribed as data; citations
    are relative to the reference checkout).  These pin the oracle.

``oracle_seeded.npz``   -- seeded inputs with outputs produced by THIS repo's CPU
    oracle (oracle/unflow_oracle.c), cross-checked against oracle/np_ref.py when
    generated.  They are a regression net for the cases the reference's tests
    do not cover (FlowNetC correlation settings, forward_warp values, soft
    census): "parity unpinned" by the reference, pinned to the restatement.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))


def reference_kat():
    zero = [0, 0]
    kat = {}

    # src/e2eflow/test/ops/backward_warp.py:27-48 and test/test_image_warp.py:23-44
    move = dict(
        first=[[0, 0, 0, 0], [0, 1, 0.5, 0], [0, 0.3, 0.4, 0], [0, 0, 0, 0]],
        second=[[0, 1, 0, 0], [0, 0, 0, 0.5], [0.3, 0, 0, 0], [0, 0, 0.4, 0]],
        flow=[[zero, [-1, 0], zero, zero],
              [zero, [0, -1], [1, 0], [0, -1]],
              [[0, -1], [-1, 0], [0, 1], zero],
              [zero, zero, [0, -1], zero]])
    kat["warp_move"] = dict(
        cite="test/ops/backward_warp.py:27-48; test/test_image_warp.py:23-44",
        applies_to=["backward_warp", "image_warp"],
        expected=np.reshape(move["first"], [1, 4, 4, 1]).tolist(),
        image=np.reshape(move["second"], [1, 4, 4, 1]).tolist(),
        flow=np.reshape(move["flow"], [1, 4, 4, 2]).tolist())

    # test/ops/backward_warp.py:50-78 (flow_2 = zeros)
    z4 = np.zeros([4, 4])
    first = np.concatenate([move["first"], z4, move["first"]])
    second = np.concatenate([move["second"], z4, move["second"]])
    flow = np.concatenate([move["flow"], np.zeros([4, 4, 2]), move["flow"]])
    kat["backward_warp_batches"] = dict(
        cite="test/ops/backward_warp.py:50-78", applies_to=["backward_warp", "image_warp"],
        expected=np.reshape(first, [3, 4, 4, 1]).tolist(),
        image=np.reshape(second, [3, 4, 4, 1]).tolist(),
        flow=np.reshape(flow, [3, 4, 4, 2]).tolist())

    # test/test_image_warp.py:46-74 (flow_2 = ones on a zero image)
    flow = np.concatenate([move["flow"], np.ones([4, 4, 2]), move["flow"]])
    kat["image_warp_batches"] = dict(
        cite="test/test_image_warp.py:46-74", applies_to=["image_warp", "backward_warp"],
        expected=np.reshape(first, [3, 4, 4, 1]).tolist(),
        image=np.reshape(second, [3, 4, 4, 1]).tolist(),
        flow=np.reshape(flow, [3, 4, 4, 2]).tolist())

    # test/ops/backward_warp.py:80-101 and test/test_image_warp.py:76-97
    kat["warp_interpolate"] = dict(
        cite="test/ops/backward_warp.py:80-101; test/test_image_warp.py:76-97",
        applies_to=["backward_warp", "image_warp"],
        expected=np.reshape([[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 2.1]],
                            [1, 4, 4, 1]).tolist(),
        image=np.reshape([[0, 0, 0, 0], [0, 1, 2, 0], [0, 3, 4, 0], [0, 0, 0, 0]],
                         [1, 4, 4, 1]).tolist(),
        flow=np.reshape([[zero, zero, zero, zero],
                         [zero, [-2, -2], [-2, -2], zero],
                         [zero, [-2, -2], [-2, -2], zero],
                         [zero, zero, zero, [-1.7, -1.6]]], [1, 4, 4, 2]).tolist())

    # test/ops/correlation.py:30-47 (trivial) and :49-69 (batch of 2)
    img = [[1, 1, 2, 2], [0, 0, 2, 2], [3, 3, 4, 4], [3, 3, 2, 2]]
    f = np.reshape(img, [1, 1, 4, 4]).astype(float)
    attrs = dict(kernel_size=1, stride_2=1, max_displacement=0, pad=0)
    kat["correlation_trivial"] = dict(
        cite="test/ops/correlation.py:30-47", first=f.tolist(), second=f.tolist(),
        expected=np.square(f).tolist(), attrs=attrs)
    f2 = np.concatenate([f, f], 0)
    kat["correlation_batch"] = dict(
        cite="test/ops/correlation.py:49-69", first=f2.tolist(), second=f2.tolist(),
        expected=np.square(f2).tolist(), attrs=attrs)
    # test/ops/correlation.py:74-89 -- disabled in the reference (early return), no
    # expected output there; kept as a gradient-consistency input only.
    kat["correlation_3x3_inputs"] = dict(
        cite="test/ops/correlation.py:74-89 (disabled test; inputs only)",
        first=np.reshape([[1, 1, 3], [0, 0, 1], [2, 2, 0.2]], [1, 1, 3, 3]).tolist(),
        second=np.reshape([[1, 2, 0.1], [3, 4, 2.2], [4, 5, 1.6]], [1, 1, 3, 3]).tolist(),
        attrs=dict(kernel_size=3, stride_2=1, max_displacement=1, pad=2))

    # test/ops/downsample.py:8-22
    kat["downsample"] = dict(
        cite="test/ops/downsample.py:8-22", scale=2,
        image=np.reshape(img, [1, 4, 4, 1]).tolist(),
        expected=np.reshape([[0.5, 2], [3, 3]], [1, 2, 2, 1]).tolist())

    # test/test_losses.py:43-55 and :56-68
    def flow2(u, v):
        fl = np.ones([1, 3, 3, 2], np.float32)
        fl[0, :, :, 0] = u
        fl[0, :, :, 1] = v
        return fl.tolist()
    kat["outgoing_mask_all_directions"] = dict(
        cite="test/test_losses.py:43-55",
        flow=flow2([[0, 0, 1], [-1, 3, 0], [0, 1, 0]], [[-1, 0, 0], [0, 0, 0], [1, -1, 0]]),
        expected=[[0, 1, 0], [0, 0, 1], [0, 1, 1]])
    kat["outgoing_mask_large_movement"] = dict(
        cite="test/test_losses.py:56-68",
        flow=flow2([[3, 2, 1], [2, 1, 0], [0, -2, -1]], [[0, 0, 0], [0, 0, 0], [0, 0, 0]]),
        expected=[[0, 0, 0], [1, 1, 1], [1, 0, 1]])

    # test/test_losses.py:11-41: _smoothness_deltas * mask
    d0 = [[0, 0, 0], [-8, 5, 0], [-1, 1, 0]]
    d1 = [[0, -8, -3], [0, 7, 3], [0, 0, 0]]
    kat["smoothness_deltas"] = dict(
        cite="test/test_losses.py:11-41",
        flow=flow2([[0, 0, 0], [0, 8, 3], [0, 1, 0]], [[0, 0, 0], [0, 8, 3], [0, 1, 0]]),
        mask0=[[1, 1, 0], [1, 1, 0], [1, 1, 0]], mask1=[[1, 1, 1], [1, 1, 1], [0, 0, 0]],
        delta_u0=d0, delta_u1=d1, delta_v0=d0, delta_v1=d1)

    # test/test_losses.py:97-121: gradient_loss of an image and the same image + 1 is ~0
    def im3(rows):
        im = np.ones([1, 3, 3, 3], np.float32)
        for c in range(3):
            im[0, :, :, c] = rows
        return im.tolist()
    kat["gradient_loss"] = dict(
        cite="test/test_losses.py:97-121",
        im1=im3([[0, 1, 0], [0, 2, 0], [0, 3, 4]]), im2=im3([[1, 2, 1], [1, 3, 1], [1, 4, 5]]),
        expected=0.0, atol=1e-2)

    # test/ops/forward_warp.py:9-19 holds no values, only the shape of its
    # gradient check ([1,10,10,2] flow, rtol=atol=1e-3).
    kat["forward_warp_gradcheck"] = dict(cite="test/ops/forward_warp.py:9-19",
                                         flow_shape=[1, 10, 10, 2], tol=1e-3)
    return kat


def smooth_field(rng, shape, sigma):
    """Low-pass filtered uniform noise rescaled to [0,1]."""
    from scipy.ndimage import gaussian_filter
    a = rng.uniform(size=shape)
    a = gaussian_filter(a, sigma=[0, sigma, sigma, 0])
    a -= a.min()
    a /= a.max()
    return a.astype(np.float32)


def oracle_seeded():
    import oracle
    from oracle import np_ref
    rng = np.random.default_rng(1234)
    out = {}

    # FlowNetC settings on a small map (C=16 keeps the file small)
    f0 = rng.standard_normal((2, 16, 12, 20)).astype(np.float32)
    f1 = rng.standard_normal((2, 16, 12, 20)).astype(np.float32)
    f0 = np.where(f0 > 0, f0, 0.1 * f0)
    f1 = np.where(f1 > 0, f1, 0.1 * f1)
    c, p0, p1 = oracle.correlation(f0, f1)
    assert np.allclose(c, np_ref.correlation(f0, f1), rtol=1e-5, atol=1e-6)
    top = rng.standard_normal(c.shape).astype(np.float32)
    g0, g1 = oracle.correlation_grad(top, f0, f1, p0, p1)
    r0, r1 = np_ref.correlation_grad(top, f0, f1)
    assert np.allclose(g0, r0, rtol=1e-4, atol=1e-5) and np.allclose(g1, r1, rtol=1e-4, atol=1e-5)
    out.update(corr_f0=f0, corr_f1=f1, corr_out=c, corr_top=top, corr_g0=g0, corr_g1=g1)

    # generic attrs
    attrs = dict(kernel_size=3, max_displacement=4, pad=3, stride_1=2, stride_2=2)
    a0 = rng.standard_normal((1, 5, 11, 13)).astype(np.float32)
    a1 = rng.standard_normal((1, 5, 11, 13)).astype(np.float32)
    c2, q0, q1 = oracle.correlation(a0, a1, **attrs)
    assert np.allclose(c2, np_ref.correlation(a0, a1, **attrs), rtol=1e-5, atol=1e-6)
    top2 = rng.standard_normal(c2.shape).astype(np.float32)
    h0, h1 = oracle.correlation_grad(top2, a0, a1, q0, q1, **attrs)
    out.update(corrg_f0=a0, corrg_f1=a1, corrg_out=c2, corrg_top=top2, corrg_g0=h0, corrg_g1=h1,
               corrg_attrs=np.array([attrs[k] for k in
                                     ("kernel_size", "max_displacement", "pad", "stride_1", "stride_2")]))

    # warps / splat / census on a 2x24x32 pair
    im1 = smooth_field(rng, (2, 24, 32, 3), 1.5)
    im2 = smooth_field(rng, (2, 24, 32, 3), 1.5)
    flow = (rng.standard_normal((2, 24, 32, 2)) * 3 + rng.uniform(-6, 6, (2, 1, 1, 2))).astype(np.float32)
    grad = rng.standard_normal((2, 24, 32, 3)).astype(np.float32)
    out.update(im1=im1, im2=im2, flow=flow, grad=grad)
    out["backward_warp"] = oracle.backward_warp(im2, flow)
    assert np.allclose(out["backward_warp"], np_ref.backward_warp(im2, flow), atol=1e-5)
    out["backward_warp_grad"] = oracle.backward_warp_grad(grad, im2, flow)
    out["image_warp"] = oracle.image_warp(im2, flow)
    assert np.allclose(out["image_warp"], np_ref.image_warp(im2, flow), atol=1e-5)
    d_im, d_flow = oracle.image_warp_grad(grad, im2, flow)
    out.update(image_warp_dim=d_im, image_warp_dflow=d_flow)
    out["forward_warp"] = oracle.forward_warp(flow)
    assert np.allclose(out["forward_warp"], np_ref.forward_warp(flow), rtol=1e-5, atol=1e-5)
    g1c = rng.standard_normal((2, 24, 32, 1)).astype(np.float32)
    out.update(forward_warp_gin=g1c, forward_warp_grad=oracle.forward_warp_grad(g1c, flow))
    out["downsample2"] = oracle.downsample(im1, 2)
    out["downsample4"] = oracle.downsample(im1, 4)
    mask = oracle.create_border_mask((2, 24, 32), 0.1)
    out["border_mask"] = mask
    for r in (1, 2, 3):
        loss, dist = oracle.ternary_loss(im1, out["image_warp"], mask, r, return_dist=True)
        ref = np_ref.ternary_loss(im1, out["image_warp"], mask, r)
        assert abs(loss - ref) <= 1e-5 * abs(ref), (loss, ref)
        out["ternary_loss_r%d" % r] = np.float64(loss)
        out["ternary_dist_r%d" % r] = dist
        out["ternary_grad_r%d" % r] = oracle.ternary_loss_grad(im1, out["image_warp"], mask, r)

    # 3x3 stencil terms (smoothness, second order, gradient loss) on the same pair
    for order in (1, 2):
        l = oracle.smoothness_loss(flow, order)
        ref = np_ref.smoothness_loss(flow, order)
        assert abs(l - ref) <= 1e-6 * abs(ref), (l, ref)
        out["smooth_loss_o%d" % order] = np.float64(l)
        out["smooth_grad_o%d" % order] = oracle.smoothness_loss_grad(flow, order)
    l = oracle.gradient_loss(im1, out["image_warp"], mask)
    ref = np_ref.gradient_loss(im1, out["image_warp"], mask)
    assert abs(l - ref) <= 1e-6 * abs(ref), (l, ref)
    out["gradient_loss"] = np.float64(l)
    out["gradient_loss_grad"] = oracle.gradient_loss_grad(im1, out["image_warp"], mask)

    # extensions (defined by include/unflow_hip.h, not by the reference): integer ternary census,
    # stacked-net warp block, fp16-storage correlation (fp32 oracle on half-rounded inputs)
    s1, s2 = oracle.census_transform(im1, 3, 0.9), oracle.census_transform(out["image_warp"], 3, 0.9)
    d, total = oracle.census_hamming(s1, s2, mask)
    out.update(census_sig1=s1, census_dist=d, census_total=np.uint64(total))
    flow_lr = (rng.standard_normal((2, 6, 8, 2)) * 0.2).astype(np.float32)
    out.update(stacked_flow_lr=flow_lr, stacked_inputs=oracle.stacked_inputs(im1, im2, flow_lr))
    h0, h1 = f0.astype(np.float16), f1.astype(np.float16)
    ch, hp0, hp1 = oracle.correlation(h0.astype(np.float32), h1.astype(np.float32))
    out.update(corr16_f0=h0, corr16_f1=h1, corr16_out=ch.astype(np.float16))
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "reference_kat.json"), "w") as fh:
        json.dump(reference_kat(), fh, indent=1)
    np.savez_compressed(os.path.join(HERE, "oracle_seeded.npz"), **oracle_seeded())
    print("wrote", os.listdir(HERE))
