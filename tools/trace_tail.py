"""Developer tool (GPU box): workgroup timelines of one locate() step, per kernel.

    make -C respmon_amd/csrc librespmon_hip_trace.so && python tools/trace_tail.py [--config P|Q|R] [--out gpurun_out/trace]

Loads the TRACING build of the library (never the product .so), runs a few warm steps, then one traced step, and prints
per kernel: first workgroup start and last workgroup end relative to the step's first record, the spread of workgroup
start times (launch ramp), workgroup durations (min / median / max) and how many distinct CUs were used.  The raw
records go to <out>/trace.npz."""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NAMES = {0: "k_down_chain", 1: "k_small_pyramid", 2: "k_temporal_mfma", 3: "k_small_filter_first (or k_small_collapse_bounds)", 4: "k_select_pairs",
         5: "k_eval_pairs | k_eval_c", 6: "k_masked_sum_tiles | k_tile_sum", 7: "k_heat_to_u8", 8: "k_extra8", 9: "k_extra9"}
KERNELS, BLOCKS = 16, 20480


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="P")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trace"))
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--debug-set", action="append", default=[], metavar="KEY=VALUE", help="rm_debug_set switches of the traced context")
    a = ap.parse_args()
    import torch
    from respmon_amd import _capi, synth
    _capi.LIB_PATH = os.path.join(ROOT, "respmon_amd", "csrc", os.environ.get("RM_TRACE_LIB", "librespmon_hip_trace.so"))
    lib = _capi.load()
    lib.rm_trace_start.restype = ctypes.c_int
    lib.rm_trace_read.restype = ctypes.c_int
    lib.rm_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    import bench
    T, H, W, L, S, dt = bench.CONFIGS[a.config]
    gen = synth.synth_breathing_dense if a.dense else (synth.synth_breathing_blocks if T * H * W > 1 << 30 else synth.synth_breathing)
    v8 = gen(T, H, W, seed=4321 if a.dense else 1234)
    tdt = {"f64": torch.float64, "f32": torch.float32, "f16": torch.float16, "u8": torch.uint8}[dt]
    buf = torch.empty((T, H, W), dtype=tdt, device="cuda")
    for t0 in range(0, T, 16):
        buf[t0:t0 + 16] = (torch.from_numpy(v8[t0:t0 + 16]).cuda().to(torch.float64) * (1.0 / 255)).to(tdt)
    from respmon_amd.base import _Backend
    be = _Backend()
    from respmon_amd import device
    for kv in a.debug_set:
        k, v = kv.split("=", 1)
        lib.rm_debug_set(device.ctx(), k.encode(), int(v))

    def step():
        return be.locate(buf, 10, 0.1, 1.0, 500, L, S, 0.7, 20, 0)
    for _ in range(a.steps):
        roi = step()
    torch.cuda.synchronize()
    assert lib.rm_trace_start() == 0
    roi = step()
    torch.cuda.synchronize()
    rec = np.zeros((KERNELS, BLOCKS), dtype=[("t0", "<u8"), ("t1", "<u8"), ("c0", "<u8"), ("c1", "<u8"), ("hwid", "<u4"), ("xcc", "<u4")])
    assert lib.rm_trace_read(rec.ctypes.data, rec.nbytes) == 0
    marks = rec[KERNELS - 1]["t0"][:KERNELS * 16].reshape(KERNELS, 16).copy()
    rec[KERNELS - 1]["t1"][:] = 0
    os.makedirs(a.out, exist_ok=True)
    np.savez_compressed(os.path.join(a.out, "trace.npz"), rec=rec)
    print("roi", roi)
    base = None
    rows = []
    for k in range(KERNELS - 1):
        r = rec[k]
        m = r["t1"] != 0
        if not m.any():
            continue
        t0 = r["t0"][m].astype(np.int64); t1 = r["t1"][m].astype(np.int64)
        if base is None:
            base = t0.min()
        cu = (r["xcc"][m] & 0xf).astype(np.int64) * 4096 + (r["hwid"][m] >> 8 & 0xff)   # xcc | se/sh/cu bits 8..15
        d = (t1 - t0) * 0.01   # 100 MHz ticks -> us
        long_ = (t1 - t0) >= 200   # >= 2 us: enough ticks for a clock estimate
        ghz = float(np.median((r["c1"][m].astype(np.int64) - r["c0"][m].astype(np.int64))[long_] / ((t1 - t0)[long_] * 10.0))) if long_.any() else 0.0
        rows.append((NAMES.get(k, str(k)), int(m.sum()), (t0.min() - base) * 0.01, (t1.max() - base) * 0.01, (t1.max() - t0.min()) * 0.01,
                     (t0.max() - t0.min()) * 0.01, d.min(), float(np.median(d)), d.max(), len(np.unique(cu)), ghz))
    print("%-26s %6s %9s %9s %8s %9s %8s %8s %8s %5s %5s" % ("kernel", "wgs", "start_us", "end_us", "span_us", "ramp_us", "wg_min", "wg_med", "wg_max", "CUs", "GHz"))
    for r in rows:
        print("%-26s %6d %9.2f %9.2f %8.2f %9.2f %8.2f %8.2f %8.2f %5d %5.2f" % r)
    for k in range(KERNELS):   # phase marks of workgroup 5 (us since its first mark)
        mk = marks[k]
        if mk.any():
            i0 = int(np.nonzero(mk)[0][0])
            print("marks %-24s %s" % (NAMES.get(k, str(k)), " ".join("%d:%.2f" % (i, (int(mk[i]) - int(mk[i0])) * 0.01) for i in range(16) if mk[i])))
    # gaps between consecutive kernels
    rows.sort(key=lambda r: r[2])
    for a_, b_ in zip(rows, rows[1:]):
        print("gap %-24s -> %-24s %7.2f us" % (a_[0], b_[0], b_[2] - a_[3]))


if __name__ == "__main__":
    main()
