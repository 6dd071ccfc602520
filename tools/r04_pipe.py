"""Experiment: K rm_locate steps with 1, 2 or 3 calibration buffers in flight (one host thread + library context + HIP stream each).
Run on the GPU box: python tools/r04_pipe.py [steps]"""
import ctypes, os, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from respmon_amd import _capi, device, synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib = _capi.load()
T, H, W = 256, 1080, 1920
vid = synth.synth_breathing(T, H, W, seed=1234)
NMAX = 3
bufs = [torch.from_numpy(vid).cuda().to(torch.float64).mul_(1.0 / 255) for _ in range(NMAX)]
ctxs, streams = [], []
for i in range(NMAX):
    h = ctypes.c_void_p()
    _capi.check(lib, lib.rm_ctx_create(0, ctypes.byref(h)), "ctx")
    ctxs.append(h); streams.append(torch.cuda.Stream())
rois = [None] * NMAX

def worker(i, n):
    xywh = (ctypes.c_int32 * 4)()
    sp = ctypes.c_void_p(streams[i].cuda_stream)
    for _ in range(n):
        rc = lib.rm_locate(ctxs[i], device.ptr(bufs[i]), device.dtype_code(bufs[i]), T, H, W, 10.0, 0.1, 1.0, 500.0, 9, 4, 0.7, 20, 0, xywh, sp)
        assert rc == 0, rc
    rois[i] = tuple(xywh)

for nfl in (1, 2, 3, 1, 2):
    for i in range(nfl): worker(i, 5)
    torch.cuda.synchronize()
    per = K // nfl
    th = [threading.Thread(target=worker, args=(i, per)) for i in range(nfl)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("in flight %d: %.4f ms per step (%d steps)  roi %s" % (nfl, dt * 1e3 / (per * nfl), per * nfl, rois[:nfl]))
