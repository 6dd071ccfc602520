"""Timing + sanity of the other BASELINE.json configs on one MI355X (not the driver's bench line; see bench.py).

  Q  config 2: 128 x 720p, 4-level pyramid, skip 2 (module defaults of eulerian_magnification_bandpass)
  R  config 5: 512 x 4K, 6-level pyramid, skip 2, float16 frame buffer
  F  config 3: pyramidal LK on a 256x256 ROI, 1000 corners, 30 fps stream
  P8 config 4 at uint8 / float32 frame buffers (the float64 line is bench.py's)

    python tools/bench_configs.py [Q R F P]      -> one JSON line per config
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _time(fn, steps, warmup):
    import torch
    for _ in range(warmup):
        r = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, r


def calib(name, T, H, W, L, S, dtype, steps=10, warmup=2, seed=1234):
    import ctypes
    import torch
    from respmon_amd import _capi, device, synth
    from respmon_amd.base import _Backend
    lib = _capi.load()
    ctx = device.ctx()
    tdt = {"f64": torch.float64, "f32": torch.float32, "f16": torch.float16, "u8": torch.uint8}[dtype]
    esz = {"f64": 8, "f32": 4, "f16": 2, "u8": 1}[dtype]
    buf = torch.empty((T, H, W), dtype=tdt, device="cuda")
    vid8 = synth.synth_breathing(T, H, W, seed=seed)     # uint8 on the host (4.2 GB at 4K x 512), converted on the device
    chunk = 16
    for t0 in range(0, T, chunk):
        v8 = torch.from_numpy(vid8[t0:t0 + chunk]).cuda()
        buf[t0:t0 + chunk] = v8 if dtype == "u8" else (v8.to(torch.float64) * (1.0 / 255)).to(tdt)   # uint8_to_float, then storage dtype
    del vid8
    be = _Backend()
    fn = lambda: be.locate(buf, 10, 0.1, 1.0, 500, L, S, 0.7, 20, 0)
    _capi.check(lib, lib.rm_profile_enable(ctx, 0), "prof")
    dt, roi = _time(fn, steps, warmup)
    _capi.check(lib, lib.rm_profile_enable(ctx, 2), "prof")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ms = (ctypes.c_double * 4)()
    n = ctypes.c_int()
    _capi.check(lib, lib.rm_profile_read(ctx, ms, ctypes.byref(n)), "prof")
    _capi.check(lib, lib.rm_profile_enable(ctx, 0), "prof")
    b_alg = T * H * W * esz + H * W * 8
    k_ms = ms[0] / max(n.value, 1)
    out = {"config": name, "frames": T, "height": H, "width": W, "levels": L, "skip": S, "frame_buffer_dtype": dtype,
           "ms_per_step": dt * 1e3, "frames_per_s": T / dt, "roi": roi,
           "frame_buffer_kernel_ms": k_ms, "algorithmic_GB": b_alg / 1e9,
           "achieved_GBps_frame_buffer_kernel": b_alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None,
           "phases_ms": {"frame_buffer_kernel": k_ms, "pyramid_rest_and_temporal": ms[1] / max(n.value, 1),
                         "collapse_passes": ms[2] / max(n.value, 1), "heatmap_to_roi": ms[3] / max(n.value, 1)},
           "workspace_GB": lib.rm_ctx_workspace_bytes(ctx) / 1e9}
    print(json.dumps(out), flush=True)
    del buf
    torch.cuda.empty_cache()


def flow(steps=60):
    import torch
    from respmon_amd import synth
    from respmon_amd.base import _Backend
    be = _Backend()
    render = synth.synth_texture(256, 256, seed=4321)
    frames = [render(1.5 * np.sin(2 * np.pi * 0.4 * t / 30), 0.5 * np.sin(2 * np.pi * 0.4 * t / 30 + np.pi / 3)) for t in range(steps + 1)]
    dev = [torch.from_numpy(f).cuda() for f in frames]
    pts = be.good_features_to_track(dev[0], 1000, 0.01, 7, 7)
    n0 = 0 if pts is None else len(pts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p = pts
    lost = 0
    for i in range(steps):
        p1, st = be.calc_optical_flow_pyr_lk(dev[i], dev[i + 1], p, (15, 15), 2, (3, 10, 0.03))
        good = st.reshape(-1) == 1
        lost += int((~good).sum())
        p = p1[good].reshape(-1, 1, 2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"config": "F: pyramidal LK, 256x256 ROI, %d corners (maxCorners=1000), winSize 15, maxLevel 2" % n0,
                      "ms_per_frame": dt * 1e3, "frames_per_s": 1 / dt, "budget_ms_at_30fps": 33.3, "points_lost": lost}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["Q", "R", "F", "P"]
    if "Q" in which:
        calib("Q: 128x720p L=4 S=2", 128, 720, 1280, 4, 2, "f64")
        calib("Q: 128x720p L=4 S=2", 128, 720, 1280, 4, 2, "f32")
    if "P" in which:
        calib("P: 256x1080p L=9 S=4", 256, 1080, 1920, 9, 4, "f32")
        calib("P: 256x1080p L=9 S=4", 256, 1080, 1920, 9, 4, "u8")
    if "R" in which:
        calib("R: 512x4K L=6 S=2", 512, 2160, 3840, 6, 2, "f16", steps=3, warmup=3)   # the value store sizes itself over the first calls
    if "F" in which:
        flow()
