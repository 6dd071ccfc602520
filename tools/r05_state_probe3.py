"""Developer probe (GPU box; must be the FIRST process on a fresh box): the first process after boot runs the frame-buffer kernel ~7 % slower
(0.727 against 0.675 ms, tools/r05_state_probe2.sh).  Does the state belong to the process, to its first allocation, or to time?"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from respmon_amd import _capi, device, synth  # noqa: E402
from respmon_amd.base import _Backend  # noqa: E402

T, H, W = 256, 1080, 1920
v8 = synth.synth_breathing(T, H, W, seed=1234)
lib = _capi.load()
be = _Backend()
ctx = device.ctx()
ms = (ctypes.c_double * 8)()
nc = ctypes.c_int()


def make():
    b = torch.empty((T, H, W), dtype=torch.float64, device="cuda")
    for t0 in range(0, T, 16):
        b[t0:t0 + 16] = torch.from_numpy(v8[t0:t0 + 16]).cuda().to(torch.float64) * (1.0 / 255)
    torch.cuda.synchronize()
    return b


def measure(b, tag, steps=80):
    for _ in range(10):
        be.locate(b, 10, 0.1, 1.0, 500, 9, 4, 0.7, 20, 0)
    _capi.check(lib, lib.rm_profile_enable(ctx, 1), "prof")
    t0 = time.perf_counter()
    for _ in range(steps):
        be.locate(b, 10, 0.1, 1.0, 500, 9, 4, 0.7, 20, 0)
    dt = (time.perf_counter() - t0) / steps * 1e3
    _capi.check(lib, lib.rm_profile_read(ctx, ms, ctypes.byref(nc)), "prof")
    _capi.check(lib, lib.rm_profile_enable(ctx, 0), "prof")
    print("%-46s step %.4f ms  kernel %.4f ms  ptr %#x" % (tag, dt, ms[0] / max(nc.value, 1), b.data_ptr()), flush=True)


a = make()
measure(a, "first buffer")
time.sleep(4)
measure(a, "first buffer, 4 s later")
b = make()
measure(b, "second buffer (first alive)")
measure(a, "first buffer again")
del a
torch.cuda.empty_cache()
c = make()
measure(c, "third buffer (first freed to the driver)")
del b, c
torch.cuda.empty_cache()
d = make()
measure(d, "fourth buffer (everything freed before)")
for i in range(3):
    time.sleep(3)
    measure(d, "fourth buffer, later")
