"""Cycle stamps of a -DUNFLOW_TUNE_FWD2_STAMPS build of the second-generation correlation forward (tools/fwd2_variants.sh
... "FWD2_STAMPS=1" "-DUNFLOW_TUNE_FWD2_STAMPS"): per wave the cycles of the channel sweep, of the epilogue, the cycles spent
waiting for its own LDS-DMA and at the chunk barriers, and the shader clock the chip held.
usage: python tools/stamps_fwd2.py path.so"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["UNFLOW_HIP_LIB"] = os.path.abspath(sys.argv[1])
os.environ["UNFLOW_ALLOW_DIAGNOSTIC_BUILD"] = "1"
import torch
from unflow_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
fa = torch.randn((32, 256, 48, 160), device=dev, generator=g)
fb = torch.randn((32, 256, 48, 160), device=dev, generator=g)
for _ in range(20):
    ops.correlation(fa, fb)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["UNFLOW_HIP_LIB"])
SLOTS, WORDS = 64, 8
buf = np.zeros(SLOTS * 8 * WORDS, np.uint64)
assert lib.unflow_debug_read_fwd2_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
st = buf.reshape(SLOTS, 8, WORDS).astype(np.int64)
st = st[st[:, 0, 0] != 0]
loop = st[:, :, 1] - st[:, :, 0]
epi = st[:, :, 2] - st[:, :, 1]
clk = (st[:, :, 2] - st[:, :, 0]) / np.maximum(st[:, :, 6] - st[:, :, 5], 1) * 100.0     # MHz
mask = st[:, :, 7] & 7
live = np.array([bin(m).count("1") for m in mask.ravel()]).reshape(mask.shape)
print("variant %s: %d workgroups stamped" % (os.path.basename(sys.argv[1]), len(st)))
print("  clock %.0f MHz (median over waves)" % np.median(clk))
for nl in (3, 2, 1, 0):
    sel = live == nl
    if sel.any():
        print("  waves with %d live N-tile rows (%4d): sweep %6.0f cycles (per chunk %5.0f), DMA wait %5.0f (%4.1f %%), barrier wait %5.0f (%4.1f %%), epilogue %5.0f" % (
            nl, sel.sum(), loop[sel].mean(), loop[sel].mean() / 32, st[:, :, 3][sel].mean(), 100 * st[:, :, 3][sel].mean() / loop[sel].mean(),
            st[:, :, 4][sel].mean(), 100 * st[:, :, 4][sel].mean() / loop[sel].mean(), epi[sel].mean()))
wg = loop.max(axis=1)
print("  workgroup sweep (slowest wave): mean %.0f, min %.0f, max %.0f cycles; epilogue mean %.0f" % (wg.mean(), wg.min(), wg.max(), epi.max(axis=1).mean()))
