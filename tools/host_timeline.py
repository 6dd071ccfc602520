"""Developer tool: where the host time of a synchronous rm_locate goes (include/respmon_hip_debug.h rm_debug_host_timeline).
    python tools/host_timeline.py [steps] [P|Q|R]
Prints the medians over `steps` back-to-back locate() calls on the headline workload (1080p x 256 float64):
  between calls (caller), entry -> first launch issued, -> all launches issued, -> device done (wait), -> contour stage done."""
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


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    cfg = sys.argv[2] if len(sys.argv) > 2 else "P"
    import bench
    T, H, W, L, S, dt = bench.CONFIGS[cfg]
    v8 = (synth.synth_breathing_blocks if T * H * W > 1 << 30 else synth.synth_breathing)(T, H, W, seed=1234)
    tdt = {"f64": torch.float64, "f32": torch.float32, "f16": torch.float16, "u8": torch.uint8}[dt]
    buf = torch.empty((T, H, W), dtype=tdt, device="cuda")
    for t0 in range(0, T, 16):
        buf[t0:t0 + 16] = (torch.from_numpy(v8[t0:t0 + 16]).cuda().to(torch.float64) * (1.0 / 255)).to(tdt)
    torch.cuda.synchronize()
    lib = _capi.load()
    be = _Backend()
    ctx = device.ctx()
    for _ in range(300 if cfg == "P" else 30):
        roi = be.locate(buf, 10, 0.1, 1.0, 500, L, S, 0.7, 20, 0)
    marks = np.zeros((steps, 5))
    out = (ctypes.c_double * 5)()
    t0 = time.perf_counter()
    for i in range(steps):
        roi = be.locate(buf, 10, 0.1, 1.0, 500, L, S, 0.7, 20, 0)
        lib.rm_debug_host_timeline(ctx, out)
        marks[i] = list(out)
    dt = (time.perf_counter() - t0) / steps * 1e3
    m = np.median(marks[5:], axis=0)
    from respmon_amd import dist
    print("roi", roi, "ms_per_step (incl. the timeline read) %.4f" % dt, "| host contour stage path (RM_ROI_PATH_*)", dist.roi_path(),
          "| components, labelled", dist.contour_stats())
    print("between calls %.1f us | entry -> first launch issued %.1f | -> all launches issued %.1f | -> device done %.1f | -> contour stage done %.1f"
          % (m[0], m[1], m[2], m[3], m[4]))
    print("host contour stage %.1f us; host work after the device finished + before the next first launch: %.1f us (+ wake-up latency of the wait)"
          % (m[4] - m[3], (m[4] - m[3]) + m[0] + m[1]))


if __name__ == "__main__":
    main()
