"""Developer tool (GPU box): does the frame-buffer kernel's time depend on where the [T,H,W] buffer sits?
Times locate() (frame-buffer kernel, HIP events) on the same video copied to (a) separately allocated buffers, (b) views at
different byte offsets inside one large allocation."""
import ctypes, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from respmon_amd import _capi, device, synth
from respmon_amd.base import _Backend

T, H, W = 256, 1080, 1920
v8 = synth.synth_breathing(T, H, W, seed=1234)
lib = _capi.load(); be = _Backend(); ctx = device.ctx()

def fill(buf):
    for t0 in range(0, T, 16):
        buf[t0:t0 + 16] = torch.from_numpy(v8[t0:t0 + 16]).cuda().to(torch.float64) * (1.0 / 255)

def measure(buf, steps=120):
    for _ in range(10):
        be.locate(buf, 10, 0.1, 1.0, 500, 9, 4, 0.7, 20, 0)
    torch.cuda.synchronize()
    lib.rm_profile_enable(ctx, 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        roi = be.locate(buf, 10, 0.1, 1.0, 500, 9, 4, 0.7, 20, 0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    pm = (ctypes.c_double * 4)(); n = ctypes.c_int()
    lib.rm_profile_read(ctx, pm, ctypes.byref(n)); lib.rm_profile_enable(ctx, 0)
    return ms, pm[0] / max(n.value, 1), roi

n = T * H * W
b0 = torch.empty((T, H, W), dtype=torch.float64, device="cuda"); fill(b0)
for _ in range(400):
    be.locate(b0, 10, 0.1, 1.0, 500, 9, 4, 0.7, 20, 0)
print("separate allocations:")
bufs = [b0]
for i in range(3):
    b = torch.empty((T, H, W), dtype=torch.float64, device="cuda"); fill(b); bufs.append(b)
for rnd in range(2):
    for i, b in enumerate(bufs):
        ms, k, roi = measure(b)
        print("  buffer %d ptr %#x (mod 2MiB %#x): step %.4f kernel %.4f" % (i, b.data_ptr(), b.data_ptr() % (2 << 20), ms, k))
del bufs[1:]
torch.cuda.empty_cache()
pool = torch.empty(n + (8 << 20) // 8, dtype=torch.float64, device="cuda")
print("views into one allocation, ptr %#x:" % pool.data_ptr())
for off in (0, 512, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 3 << 20):
    v = pool[off // 8: off // 8 + n].view(T, H, W)
    v.copy_(b0)
    ms, k, roi = measure(v)
    print("  offset %8d: step %.4f kernel %.4f roi %s" % (off, ms, k, roi))
