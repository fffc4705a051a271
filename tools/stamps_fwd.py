"""Shader-clock stamps of a -DUNFLOW_TUNE_STAMPS build of the correlation FORWARD (half-window kernel):
per 8-channel chunk the cycles in the two K-steps and in the chunk barrier, then the epilogue."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["UNFLOW_HIP_LIB"] = os.path.abspath(sys.argv[1])
import torch
from unflow_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
fa = torch.randn((32, 256, 48, 160), device=dev, generator=g)
fb = torch.randn((32, 256, 48, 160), device=dev, generator=g)
for _ in range(5):
    ops.correlation(fa, fb)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["UNFLOW_HIP_LIB"])
BLOCKS, ROWS, PER = 8, 128, 4
buf = np.zeros(BLOCKS * 2 * (ROWS + 1) * PER, np.uint64)
assert lib.unflow_debug_read_fwd_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
st = buf.reshape(BLOCKS, 2, ROWS + 1, PER).astype(np.int64)
print("variant", os.path.basename(sys.argv[1]))
for blk in (0, 1, 4, 5):
    for wv in (0, 1):
        s = st[blk, wv]
        if s[0, 0] == 0:
            continue
        c = s[:32]
        kst, bar = c[:, 2] - c[:, 0], c[:, 3] - c[:, 2]
        print("block slot %d wave %d: prologue->chunk0 %d | chunks: ksteps mean %d (min %d max %d), barrier mean %d (min %d max %d), chunk period mean %d | "
              "main loop %d cycles, epilogue %d cycles" % (blk, wv * 4, c[0, 0] - s[ROWS, 0], kst.mean(), kst.min(), kst.max(), bar.mean(), bar.min(), bar.max(),
                                                         np.diff(c[:, 0]).mean(), s[ROWS, 1] - c[0, 0], s[ROWS, 2] - s[ROWS, 1]))
