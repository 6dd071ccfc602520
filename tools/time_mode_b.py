"""Time the rank-local part of a Mode B step (calibrate -> sparse pack -> merge -> ROI) against plain rm_locate on one GPU:
what a multi-rank step costs besides the all-gather itself."""
import sys, time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from respmon_amd import synth, dist as rdist
from respmon_amd.base import _Backend

T, H, W = 256, 1080, 1920
vid = synth.synth_breathing(T, H, W, seed=1234)
d8 = torch.from_numpy(vid).cuda()
buf = torch.empty((T, H, W), dtype=torch.float64, device="cuda")
for t0 in range(0, T, 16):
    buf[t0:t0 + 16] = d8[t0:t0 + 16].to(torch.float64) * (1.0 / 255)
del d8
be = _Backend()
def plain():
    return be.locate(buf, 10, 0.1, 1.0, 500, 9, 4, 0.7, 20, 0)
def modeb():
    return rdist.locate_streams(buf, 10, threshold=20, sparse=True, pyramid_levels=9, skip_levels_at_top=4)
def modeb_dense():
    return rdist.locate_streams(buf, 10, threshold=20, sparse=False, pyramid_levels=9, skip_levels_at_top=4)
for _ in range(600):
    plain()
for name, fn in [("plain", plain), ("modeB_sparse", modeb), ("modeB_dense", modeb_dense), ("plain", plain), ("modeB_sparse", modeb)]:
    for _ in range(20):
        r = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        r = fn()
    torch.cuda.synchronize()
    print("%-14s %.4f ms/step roi=%s" % (name, (time.perf_counter() - t0) / 200 * 1e3, r), flush=True)
