"""Reads the shader-clock stamps of a -DUNFLOW_TUNE_FW_STAMPS build of warp.hip (diagnostic): the phases of
forward_warp_fwd_kernel for 64 interior workgroups (waves 0 and 15), bench flow, B=32 384x1280.
usage: python tools/stamps_fwarp.py <variant.so>"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["UNFLOW_HIP_LIB"] = os.path.abspath(sys.argv[1])
os.environ["UNFLOW_ALLOW_DIAGNOSTIC_BUILD"] = "1"   # a stamps build is a diagnostic build
import torch  # noqa: E402
import bench  # noqa: E402
from unflow_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
wl = bench.Workload(32, dev, seed=1234, global_batch=32, features=False)
flow = wl.flow_fw.detach()
for _ in range(3):
    ops.forward_warp(flow)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["UNFLOW_HIP_LIB"])
buf = np.zeros(64 * 2 * 8, np.uint64)
lib.unflow_debug_read_fw_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.unflow_debug_read_fw_stamps(buf.ctypes.data, buf.nbytes) == 0
st = buf.reshape(64, 2, 8).astype(np.int64)
names = ["load flows + guess", "barrier-or (window miss?)", "zero window + barrier", "taps (81 LDS atomics / px)", "barrier", "flush"]
for wv, label in ((0, "wave 0"), (1, "wave 15")):
    d = np.diff(st[:, wv, :7], axis=1)
    print(label, "phase cycles: mean / min / max over 64 workgroups")
    for i, n in enumerate(names):
        print("  %-28s %7.0f %7d %7d" % (n, d[:, i].mean(), d[:, i].min(), d[:, i].max()))
    print("  %-28s %7.0f" % ("total", (st[:, wv, 6] - st[:, wv, 0]).mean()))
print("wave 15 starts %.0f cycles after wave 0 (mean); wave 15 ends %.0f cycles after wave 0" % (
    (st[:, 1, 0] - st[:, 0, 0]).mean(), (st[:, 1, 6] - st[:, 0, 6]).mean()))
t0 = np.sort(st[:, 0, 0])
print("start-to-start of the 64 stamped workgroups: span %d cycles" % (t0[-1] - t0[0]))
rt = st[:, :, 7].astype(np.float64)
print("shader clock from s_memrealtime (100 MHz): %.0f MHz" % ((st[:, :, 6] - st[:, :, 0]) / rt * 100).mean())
order = st[:, 0, 0].reshape(4, 16)          # [tile row 8..11][tile column 4..19] = dispatch order within a row
base = order.min()
print("start of wave 0 relative to the earliest of the 64 (cycles), by tile row / column:")
for r in range(4):
    print("  row %d:" % (8 + r), " ".join("%7d" % (v - base) for v in order[r]))
print("end (flush issued), same layout:")
for r in range(4):
    print("  row %d:" % (8 + r), " ".join("%7d" % (v - base) for v in st[:, 0, 6].reshape(4, 16)[r]))
