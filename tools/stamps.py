"""Reads the shader-clock stamps of a UNFLOW_TUNE_STAMPS build of the correlation backward
(diagnostic): per window row, cycles spent in setup / K-steps / barrier / after the barrier.
usage: python tools/stamps.py <variant.so>"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["UNFLOW_HIP_LIB"] = os.path.abspath(sys.argv[1])
os.environ["UNFLOW_ALLOW_DIAGNOSTIC_BUILD"] = "1"   # a stamps build is a diagnostic build
import torch  # noqa: E402
from unflow_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(0)
B = 32
fa = torch.randn((B, 256, 48, 160), device=dev, generator=g)
fb = torch.randn((B, 256, 48, 160), device=dev, generator=g)
top = torch.randn((B, 441, 48, 160), device=dev, generator=g)
for _ in range(5):
    ops.correlation_grad(top, fa, fb)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["UNFLOW_HIP_LIB"])
BLOCKS, ROWS, PER = 8, 128, 4
buf = np.zeros(BLOCKS * 2 * (ROWS + 1) * PER, np.uint64)
rc = lib.unflow_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
st = buf.reshape(BLOCKS, 2, ROWS + 1, PER).astype(np.int64)
hs = int(os.environ.get('ROWS', '24'))
print("variant", os.path.basename(sys.argv[1]))
for blk in (0, 1, 4, 5):
    for wv in (0, 1):
        s = st[blk, wv]
        if s[0, 0] == 0:
            continue
        rows = s[:hs]
        setup = rows[:, 1] - rows[:, 0]
        kst = rows[:, 2] - rows[:, 1]
        bar = rows[:, 3] - rows[:, 2]
        after = np.append(rows[1:, 0] - rows[:-1, 3], 0)
        total = rows[-1, 3] - rows[0, 0]
        print("slot block %d wave %d: rows total %d cycles; prologue end -> row 0: %d" % (blk, wv * 4, total, rows[0, 0] - s[ROWS, 0]))
        print("   row :" + "".join("%6d" % (w % 24) for w in range(hs)))
        print("  setup:" + "".join("%6d" % v for v in setup))
        print("  kstep:" + "".join("%6d" % v for v in kst))
        print("  barr :" + "".join("%6d" % v for v in bar))
        print("  after:" + "".join("%6d" % v for v in after))
        print("  sums : setup %d  ksteps %d  barrier %d  after-barrier %d" % (setup.sum(), kst.sum(), bar.sum(), after.sum()))
