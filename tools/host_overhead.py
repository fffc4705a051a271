import os, sys, time, ctypes
sys.path.insert(0, os.getcwd())
import torch
from unflow_amd import ops, _lib
t = torch.empty(1000, device="cuda:0")
g = torch.empty(1000, device="cuda:0", requires_grad=True)
for name, fn in (("data_ptr", lambda: t.data_ptr()), ("_ptr", lambda: ops._ptr(t)), ("_ptr(None)", lambda: ops._ptr(None)), ("_ptr(requires_grad)", lambda: ops._ptr(g)),
                 ("_stream", ops._stream), ("_on", lambda: ops._on(t.device)), ("torch.empty", lambda: torch.empty((32, 96, 320, 2), device=t.device)),
                 ("_dev", lambda: ops._dev(t, "x")), ("load", _lib.load)):
    n = 20000
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    print("%-22s %.2f us" % (name, (time.perf_counter() - t0) / n * 1e6))
