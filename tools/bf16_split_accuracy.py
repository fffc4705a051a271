"""Numerical side of the split-precision experiment: correlation dot products over C
channels with every f32 operand split into three bf16 terms (hi, mid, lo) and the six largest
cross products accumulated in f32 -- emulated in numpy -- against float64, next to plain f32
accumulation.  Cases: N(0,1) FlowNetC features (C=256) and the adversarial cancelling inputs of
tests/test_gpu_parity.py::test_correlation_adversarial_cancellation."""
import numpy as np


def bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split3(x):
    hi = bf16(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16(r1)
    lo = bf16((r1 - mid).astype(np.float32))
    return hi, mid, lo


def dot_split(a, b):
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    acc = np.zeros(a.shape[1:], np.float32)
    for c0 in range(0, a.shape[0], 32):          # one 16x16x32 MFMA per term and 32 channels, f32 accumulate
        s = slice(c0, c0 + 32)
        for x, y in ((ah, bh), (ah, bm), (am, bh), (ah, bl), (al, bh), (am, bm)):
            acc = (acc + (x[s].astype(np.float64) * y[s]).sum(0).astype(np.float32)).astype(np.float32)
    return acc


def dot_f32(a, b):
    acc = np.zeros(a.shape[1:], np.float32)
    for c in range(a.shape[0]):
        acc = np.float32(a[c] * b[c] + acc)      # fmaf chain (MFMA f32 is exact f32, k-ordered)
    return acc


rng = np.random.default_rng(0)
for name, a, b in (("N(0,1) features, C=256", rng.standard_normal((256, 4096)).astype(np.float32), rng.standard_normal((256, 4096)).astype(np.float32)),
                   ("adversarial: 1e3-magnitude cancelling pairs, C=64", None, None)):
    if a is None:
        a = (rng.standard_normal((64, 4096)) * 1e3).astype(np.float32)
        b = a * np.where(np.arange(64)[:, None] % 2 == 0, 1, -1).astype(np.float32)
    C = a.shape[0]
    ref = (a.astype(np.float64) * b).sum(0) / C
    scale = (np.abs(a.astype(np.float64) * b)).sum(0).mean() / C       # the test's yardstick: mean sum|a b| / C
    for tag, got in (("f32 fma chain", dot_f32(a, b) / np.float32(C)), ("bf16 x3, 6 products", dot_split(a, b) / np.float32(C))):
        err = np.abs(got - ref).max()
        print("%-52s %-22s max abs err %.3e  = %.2e of mean sum|ab|/C  (bar 1e-5)" % (name, tag, err, err / scale))
