"""Developer tool (GPU box): A/B two builds of the library on the SAME box in one process-per-variant run.
    python tools/ab_lib.py respmon_amd/csrc/librespmon_hip.so /tmp/variant.so [--config P] [--steps 200]
Prints ms per locate() and the frame-buffer kernel's ms (HIP events, every 8th call) for each library, alternating A B A B."""
import argparse
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(lib_path, config, steps):
    import torch
    from respmon_amd import _capi, device, synth
    _capi.LIB_PATH = os.path.abspath(lib_path)
    lib = _capi.load()
    import bench
    T, H, W, L, S, dt = bench.CONFIGS[config]
    v8 = synth.synth_breathing(T, H, W, seed=1234)
    tdt = {"f64": torch.float64, "f32": torch.float32, "f16": torch.float16, "u8": torch.uint8}[dt]
    buf = torch.empty((T, H, W), dtype=tdt, device="cuda")
    for t0 in range(0, T, 16):
        buf[t0:t0 + 16] = (torch.from_numpy(v8[t0:t0 + 16]).cuda().to(torch.float64) * (1.0 / 255)).to(tdt)
    from respmon_amd.base import _Backend
    be = _Backend()
    f = lambda: be.locate(buf, 10, 0.1, 1.0, 500, L, S, 0.7, 20, 0)
    for _ in range(600):
        f()
    torch.cuda.synchronize()
    out = []
    for rep in range(3):
        _capi.check(lib, lib.rm_profile_enable(device.ctx(), 1), "enable")
        t0 = time.perf_counter()
        for _ in range(steps):
            f()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps * 1e3
        ms = (ctypes.c_double * 4)(); n = ctypes.c_int()
        _capi.check(lib, lib.rm_profile_read(device.ctx(), ms, ctypes.byref(n)), "read")
        out.append((el, ms[0] / max(n.value, 1)))
    print("%-50s step %s  kernel %s" % (os.path.basename(lib_path), " ".join("%.4f" % a for a, _ in out), " ".join("%.4f" % b for _, b in out)), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--config", default="P")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--child", default=None)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    if a.child:
        child(a.child, a.config, a.steps)
    else:
        for rep in range(a.reps):
            for lp in a.libs:
                subprocess.call([sys.executable, os.path.abspath(__file__), "--child", lp, "--config", a.config, "--steps", str(a.steps), lp])
