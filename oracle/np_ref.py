"""Independent float64 numpy restatements (closed forms) of the same path.

TEST INFRASTRUCTURE ONLY.  These are written from the *mathematical*
definition of each op, not from the loop nests, so that the
literal C restatement in ``unflow_oracle.c`` and these agree only if both read
the reference the same way.  Small inputs only (pure numpy, some Python loops).
"""
import numpy as np


def correlation(f0, f1, kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2):
    """ops/correlation_op.cu.cc:51-117 as a closed form.

    out[b,(p+r)*gw+(o+r),y,x] = 1/(k*k*C) * sum_{j,i,c}
        P0[b,c, y*s1+md+j, x*s1+md+i] * P1[b,c, y*s1+md+j+p*s2, x*s1+md+i+o*s2]
    with P = zero padding of the input by `pad`.
    """
    f0 = np.asarray(f0, np.float64)
    f1 = np.asarray(f1, np.float64)
    B, C, H, W = f0.shape
    k, md, s1, s2 = kernel_size, max_displacement, stride_1, stride_2
    kr = (k - 1) // 2
    border = md + kr
    r = md // s2
    gw = 2 * r + 1
    ph, pw = H + 2 * pad, W + 2 * pad
    oh = int(np.ceil(np.float32(ph - 2 * border) / np.float32(s1)))
    ow = int(np.ceil(np.float32(pw - 2 * border) / np.float32(s1)))
    P0 = np.pad(f0, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    P1 = np.pad(f1, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    out = np.zeros((B, gw * gw, oh, ow))
    ys = np.arange(oh) * s1 + md
    xs = np.arange(ow) * s1 + md
    for p in range(-r, r + 1):
        for o in range(-r, r + 1):
            acc = np.zeros((B, oh, ow))
            for j in range(k):
                for i in range(k):
                    a = P0[:, :, ys[:, None] + j, xs[None, :] + i]
                    b = P1[:, :, ys[:, None] + j + p * s2, xs[None, :] + i + o * s2]
                    acc += (a * b).sum(1)
            out[:, (p + r) * gw + (o + r)] = acc / (k * k * C)
    return out


def correlation_grad(top, f0, f1, kernel_size=1, max_displacement=20, pad=20, stride_1=1, stride_2=2):
    """Exact adjoint of ``correlation`` above (independent of Backward0/1's
    range arithmetic): scatter of top * other-input through the same index map."""
    f0 = np.asarray(f0, np.float64)
    f1 = np.asarray(f1, np.float64)
    top = np.asarray(top, np.float64)
    B, C, H, W = f0.shape
    k, md, s1, s2 = kernel_size, max_displacement, stride_1, stride_2
    r = md // s2
    gw = 2 * r + 1
    oh, ow = top.shape[2], top.shape[3]
    P0 = np.pad(f0, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    P1 = np.pad(f1, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    G0 = np.zeros_like(P0)
    G1 = np.zeros_like(P1)
    ys = np.arange(oh) * s1 + md
    xs = np.arange(ow) * s1 + md
    norm = k * k * C
    for p in range(-r, r + 1):
        for o in range(-r, r + 1):
            t = top[:, (p + r) * gw + (o + r)][:, None] / norm  # [B,1,oh,ow]
            for j in range(k):
                for i in range(k):
                    y0, x0 = ys[:, None] + j, xs[None, :] + i
                    y1, x1 = y0 + p * s2, x0 + o * s2
                    # strided positions never collide inside one (p,o,j,i) term
                    G0[:, :, y0, x0] += t * P1[:, :, y1, x1]
                    G1[:, :, y1, x1] += t * P0[:, :, y0, x0]
    sl = (slice(None), slice(None), slice(pad, pad + H), slice(pad, pad + W))
    return G0[sl], G1[sl]


def backward_warp(images, flows):
    """ops/backward_warp_op.cu.cc:14-68: bilinear, zero outside."""
    images = np.asarray(images, np.float64)
    B, H, W, C = images.shape
    f = np.asarray(flows, np.float32)
    gx = np.arange(W, dtype=np.float32)[None, None, :]
    gy = np.arange(H, dtype=np.float32)[None, :, None]
    x = (gx + f[..., 0]).astype(np.float32)
    y = (gy + f[..., 1]).astype(np.float32)
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    out = np.zeros_like(images)
    bidx = np.arange(B)[:, None, None]
    for dy in (0, 1):
        for dx in (0, 1):
            xi, yi = x0 + dx, y0 + dy
            wx = (x - x0) if dx else (x0 + 1 - x)
            wy = (y - y0) if dy else (y0 + 1 - y)
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            v = images[bidx, np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)]
            out += (ok * wx.astype(np.float64) * wy.astype(np.float64))[..., None] * v
    return out


def image_warp(im, flow):
    """core/image_warp.py:4-76: floor the flow, clamp the taps."""
    im = np.asarray(im, np.float64)
    B, H, W, C = im.shape
    f = np.asarray(flow, np.float32)
    fl = np.floor(f)
    w = (f - fl).astype(np.float64)
    xw, yw = w[..., 0], w[..., 1]
    x0 = np.arange(W)[None, None, :] + fl[..., 0].astype(np.int64)
    y0 = np.arange(H)[None, :, None] + fl[..., 1].astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    x0, x1 = np.clip(x0, 0, W - 1), np.clip(x1, 0, W - 1)
    y0, y1 = np.clip(y0, 0, H - 1), np.clip(y1, 0, H - 1)
    b = np.arange(B)[:, None, None]
    Ia, Ib, Ic, Id = im[b, y0, x0], im[b, y1, x0], im[b, y0, x1], im[b, y1, x1]
    wa, wb = (1 - xw) * (1 - yw), (1 - xw) * yw
    wc, wd = xw * (1 - yw), xw * yw
    return wa[..., None] * Ia + wb[..., None] * Ib + wc[..., None] * Ic + wd[..., None] * Id


def forward_warp(flows):
    """ops/forward_warp_op.cu.cc:16-65 as a *gather*: every output pixel sums
    exp(-|n - t|^2/2) over the source pixels whose clamped window contains it."""
    f = np.asarray(flows, np.float32)
    B, H, W, _ = f.shape
    out = np.zeros((B, H, W, 1))
    k = np.float32(4.0)
    for b in range(B):
        for sy in range(H):
            for sx in range(W):
                tx = np.float32(sx) + f[b, sy, sx, 0]
                ty = np.float32(sy) + f[b, sy, sx, 1]
                if not (np.floor(tx - k) < W and np.floor(tx + k) >= 0 and
                        np.floor(ty - k) < H and np.floor(ty + k) >= 0):
                    continue
                x_lo = int(np.floor(tx - k)) if tx - k > 0 else 0
                y_lo = int(np.floor(ty - k)) if ty - k > 0 else 0
                x_hi = int(np.floor(tx + k)) if tx + k < W else W - 1
                y_hi = int(np.floor(ty + k)) if ty + k < H else H - 1
                if x_hi < x_lo or y_hi < y_lo:
                    continue
                nx = np.arange(x_lo, x_hi + 1, dtype=np.float64)[None, :]
                ny = np.arange(y_lo, y_hi + 1, dtype=np.float64)[:, None]
                out[b, y_lo:y_hi + 1, x_lo:x_hi + 1, 0] += np.exp(
                    -((nx - float(tx)) ** 2 + (ny - float(ty)) ** 2) / 2.0)
    return out


def downsample(images, scale):
    images = np.asarray(images, np.float64)
    B, H, W, C = images.shape
    return images.reshape(B, H // scale, scale, W // scale, scale, C).mean(axis=(2, 4))


def gray255(im):
    im = np.asarray(im, np.float64)
    w = np.array([0.2989, 0.5870, 0.1140], np.float32).astype(np.float64)
    return (im * w).sum(-1) * 255.0


def ternary_dist(im1, im2_warped, max_distance):
    """losses.py:93-114: per-pixel soft Hamming distance [B,H,W]."""
    r = max_distance
    g1 = np.pad(gray255(im1), ((0, 0), (r, r), (r, r)))
    g2 = np.pad(gray255(im2_warped), ((0, 0), (r, r), (r, r)))
    B, Hp, Wp = g1.shape
    H, W = Hp - 2 * r, Wp - 2 * r
    c1, c2 = g1[:, r:r + H, r:r + W], g2[:, r:r + H, r:r + W]
    dist = np.zeros((B, H, W))
    for i in range(2 * r + 1):
        for j in range(2 * r + 1):
            d1 = g1[:, i:i + H, j:j + W] - c1
            d2 = g2[:, i:i + H, j:j + W] - c2
            t1 = d1 / np.sqrt(0.81 + d1 * d1)
            t2 = d2 / np.sqrt(0.81 + d2 * d2)
            e = (t1 - t2) ** 2
            dist += e / (0.1 + e)
    return dist


def charbonnier(x, mask=None, alpha=0.45, beta=1.0, epsilon=0.001):
    x = np.asarray(x, np.float64)
    err = ((x * beta) ** 2 + epsilon ** 2) ** alpha
    if mask is not None:
        err = err * np.asarray(mask, np.float64)
    return err.sum() / x.size


def ternary_loss(im1, im2_warped, mask, max_distance=1):
    r = max_distance
    dist = ternary_dist(im1, im2_warped, r)[..., None]
    B, H, W, _ = dist.shape
    tm = np.zeros((B, H, W, 1))
    tm[:, r:H - r, r:W - r] = 1.0
    return charbonnier(dist, np.asarray(mask, np.float64) * tm)


def _shift(a, dy, dx):
    """a[y+dy, x+dx] with zeros outside (NHWC, conv2d SAME)."""
    B, H, W, C = a.shape
    p = np.zeros((B, H + 2, W + 2, C), a.dtype)
    p[:, 1:-1, 1:-1] = a
    return p[:, 1 + dy:1 + dy + H, 1 + dx:1 + dx + W]


def smoothness_loss(flow, order=1):
    """losses.py:206-225,250-255 (order 1) and :258-295 (order 2), float64, by array shifts."""
    f = np.asarray(flow, np.float64)
    B, H, W, _ = f.shape
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    if order == 1:
        deltas = [f - _shift(f, 0, 1), f - _shift(f, 1, 0)]
        masks = [xx < W - 1, yy < H - 1]
    else:
        mx, my = (xx >= 1) & (xx < W - 1), (yy >= 1) & (yy < H - 1)
        deltas = [_shift(f, 0, -1) - 2 * f + _shift(f, 0, 1), _shift(f, -1, 0) - 2 * f + _shift(f, 1, 0),
                  _shift(f, -1, -1) - 2 * f + _shift(f, 1, 1), _shift(f, -1, 1) - 2 * f + _shift(f, 1, -1)]
        masks = [mx, my, mx & my, mx & my]
    K = len(deltas)
    total = 0.0
    for d, m in zip(deltas, masks):
        total += ((d ** 2 + 1e-6) ** 0.45 * m[None, :, :, None]).sum()
    return total / (B * H * W * K)      # loss_u + loss_v, each normalised by B*H*W*K


def gradient_loss(im1, im2_warped, mask):
    """losses.py:228-247, float64."""
    a, b = np.asarray(im1, np.float64), np.asarray(im2_warped, np.float64)
    B, H, W, _ = a.shape
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    mx, my = (xx >= 1) & (xx < W - 1), (yy >= 1) & (yy < H - 1)

    def sobel(z):
        gx = sum(w * (_shift(z, dy, 1) - _shift(z, dy, -1)) for dy, w in ((-1, 1), (0, 2), (1, 1)))
        gy = sum(w * (_shift(z, 1, dx) - _shift(z, -1, dx)) for dx, w in ((-1, 1), (0, 2), (1, 1)))
        return gx, gy
    ax, ay = sobel(a)
    bx, by = sobel(b)
    m = np.asarray(mask, np.float64)
    ex = ((ax - bx) ** 2 + 1e-6) ** 0.45 * m * mx[None, :, :, None]
    ey = ((ay - by) ** 2 + 1e-6) ** 0.45 * m * my[None, :, :, None]
    return (ex.sum() + ey.sum()) / (B * H * W * 6)


def resize_area(images, out_h, out_w):
    """tf.image.resize_area (TF 1.x, align_corners=False) as an exact overlap integral in
    float64: out[y, x] = mean of the input over the box [y*sy, (y+1)*sy) x [x*sx, (x+1)*sx),
    the input being piecewise constant on unit pixels and clamped at the border --
    independent of the loop/weight formulation the C oracle restates."""
    a = np.asarray(images, np.float64)
    B, H, W, C = a.shape

    def overlap(n_in, n_out):
        s = n_in / n_out
        m = np.zeros((n_out, n_in))
        for o in range(n_out):
            lo, hi = o * s, (o + 1) * s
            for i in range(int(np.floor(lo)), int(np.ceil(hi))):
                m[o, min(max(i, 0), n_in - 1)] += max(0.0, min(hi, i + 1) - max(lo, i))
        return m / s
    my, mx = overlap(H, out_h), overlap(W, out_w)
    return np.einsum("yh,bhwc,xw->byxc", my, a, mx)


def numeric_grad(fn, x, eps=1e-3):
    """Central-difference gradient of scalar fn wrt array x (float64)."""
    x = np.array(x, np.float64)
    g = np.zeros_like(x)
    it = np.nditer(x, flags=["multi_index"])
    while not it.finished:
        idx = it.multi_index
        old = x[idx]
        x[idx] = old + eps
        fp = fn(x)
        x[idx] = old - eps
        fm = fn(x)
        x[idx] = old
        g[idx] = (fp - fm) / (2 * eps)
        it.iternext()
    return g
