"""Developer tool (GPU box): A/B of rm_debug_set combinations INSIDE one process, on ONE frame buffer.
    python tools/ab_inproc.py [--config P|Q|R] [--in-dtype f64] [--rounds 4] [--steps 150] "dc_prio=0" "dc_prio=2" "dc_prio=2,dc_prio_shift=5" ...
Separate processes allocate the frame buffer on different physical pages, and the frame-buffer kernel's time moves by ~6 % with
the placement alone (0.70 / 0.745 ms for one binary on one box): only alternating the settings on the same buffer separates a
2 % effect from that.  Prints, per setting, the step and frame-buffer-kernel times of every round and their medians."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from respmon_amd import _capi, device, synth  # noqa: E402
from respmon_amd.base import _Backend  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="P")
    ap.add_argument("--in-dtype", default=None)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--video", default="breathing")
    ap.add_argument("combos", nargs="+")
    a = ap.parse_args()
    T, H, W, L, S, dt = bench.CONFIGS[a.config]
    dt = a.in_dtype or dt
    gen = {"breathing": synth.synth_breathing_blocks if T * H * W > 1 << 30 else synth.synth_breathing, "noise": synth.synth_noise_only,
           "dense": synth.synth_breathing_dense, "blobs16": synth.synth_breathing_16}[a.video]
    v8 = gen(T, H, W, seed=1234) if a.video == "breathing" else gen(T, H, W)
    tdt = {"f64": torch.float64, "f32": torch.float32, "f16": torch.float16, "u8": torch.uint8, "bgr8": torch.uint8}[dt]
    buf = torch.empty((T, H, W, 3) if dt == "bgr8" else (T, H, W), dtype=tdt, device="cuda")
    for t0 in range(0, T, 16):
        chunk = torch.from_numpy(v8[t0:t0 + 16]).cuda()
        if dt == "bgr8":     # three equal planes: gray(x, x, x) == x
            buf[t0:t0 + 16] = chunk.unsqueeze(-1).expand(-1, -1, -1, 3)
        else:
            buf[t0:t0 + 16] = chunk if dt == "u8" else (chunk.to(torch.float64) * (1.0 / 255)).to(tdt)
    del v8
    torch.cuda.synchronize()
    lib = _capi.load()
    be = _Backend()
    ctx = device.ctx()
    defaults = {}

    def apply(combo):
        for k, v in defaults.items():
            _capi.check(lib, lib.rm_debug_set(ctx, k.encode(), v), "rm_debug_set")
        if combo in ("", "default"):
            return
        for kv in combo.split(","):
            k, v = kv.split("=")
            _capi.check(lib, lib.rm_debug_set(ctx, k.encode(), int(v)), "rm_debug_set")

    # the values the knobs have by default (to put them back between settings): what the combos mention, read from the first token "k=v@d"
    combos = []
    for c in a.combos:
        parts = []
        for kv in c.split(","):
            if "@" in kv:
                kv, d = kv.split("@")
                defaults[kv.split("=")[0]] = int(d)
            parts.append(kv)
        combos.append(",".join(parts))
    for c in combos:
        for kv in c.split(","):
            if "=" in kv and kv.split("=")[0] not in defaults:
                defaults[kv.split("=")[0]] = {"dc_prio": 2, "dc_prio_shift": 0, "heat_rows": 1, "ff_parts": 0, "dc_split": 0, "dc_segs": 0}.get(kv.split("=")[0], 0)

    def step():
        return be.locate(buf, 10, 0.1, 1.0, 500, L, S, 0.7, 20, 0)
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); one = time.perf_counter() - t0
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); one = time.perf_counter() - t0
    for _ in range(int(min(600, max(10, 0.7 / max(one, 1e-5))))):
        step()
    res = {c: [] for c in combos}
    rois = {}
    for r in range(a.rounds):
        for c in combos:
            apply(c)
            for _ in range(5):
                rois[c] = step()
            torch.cuda.synchronize()
            _capi.check(lib, lib.rm_profile_enable(ctx, 1), "profile")
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            pm = (ctypes.c_double * 4)(); n = ctypes.c_int()
            _capi.check(lib, lib.rm_profile_read(ctx, pm, ctypes.byref(n)), "profile_read")
            _capi.check(lib, lib.rm_profile_enable(ctx, 0), "profile")
            res[c].append((ms, pm[0] / max(n.value, 1)))
    for c in combos:
        ms = np.array(res[c])
        print("%-44s step ms %s  median %.4f | kernel ms median %.4f | roi %s" % (c, " ".join("%.4f" % v for v in ms[:, 0]), np.median(ms[:, 0]),
                                                                                 np.median(ms[:, 1]), rois[c]))


if __name__ == "__main__":
    main()
