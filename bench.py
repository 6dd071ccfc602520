"""bench.py -- Eulerian-calibration frames/sec on a 1080p x 256 buffer (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one locate() (calibration + ROI) over one [T,H,W] frame buffer already resident in HBM.
N>1: one process per GPU, each with its own stream (weak scaling, Mode B of respmon_amd/dist.py) and one RCCL
exchange of the [H,W] heatmaps per step (an all-gather of sparse packets; dense all-reduce(sum) as the fallback);
--mode sharded: ONE buffer split by frame index over the GPUs (Mode A, strong scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
DT_BYTES = {"f64": 8, "f32": 4, "f16": 2, "u8": 1}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--in-dtype", default="f64", choices=list(DT_BYTES), help="device frame-buffer element type; "
                    "f64 is the reference's calibration_buffer dtype (base.py:119)")
    ap.add_argument("--levels", type=int, default=9)
    ap.add_argument("--skip", type=int, default=4)
    ap.add_argument("--prewarm-steps", type=int, default=600, help="untimed steps run BEFORE the W warmup steps (a fixed count so that "
                    "all ranks issue the same collectives): a fresh box starts in a low power state and its clocks take a few hundred ms "
                    "of load to settle -- measured 1.18 -> 1.14 ms/step (reported as pre_warm_steps; 0 disables)")
    ap.add_argument("--mode", default="streams", choices=["streams", "sharded"], help="N>1 only.  streams (default, BASELINE config 4): one "
                    "independent [T,H,W] stream per GPU + one heatmap all-reduce, weak scaling.  sharded: ONE [T,H,W] buffer split by frame "
                    "index over the GPUs (respmon_amd.dist.locate_sharded), strong scaling")
    ap.add_argument("--no-prune", action="store_true")
    ap.add_argument("--no-u8-alt", action="store_true", help="skip the extra measurement with a uint8 frame buffer")
    ap.add_argument("--no-roi-flow", action="store_true", help="skip the per-frame ROI optical-flow measurement")
    ap.add_argument("--cpu-frames", type=int, default=-1, help="frames of the same workload timed on the CPU oracle (0 = skip; -1 = all T frames if host memory allows, else 64)")
    return ap.parse_args()


def cpu_baseline(vid_u8, n_frames, levels, skip):
    """The oracle (a port: the reference's materialising algorithm with the build's C restatement of the
    cv2 calls) timed on this host, single thread like the reference, on the first n_frames frames."""
    from oracle import respmon_oracle as oracle
    oracle.build()
    frames = oracle.uint8_to_float(vid_u8[:n_frames])
    t0 = time.perf_counter()
    roi = oracle.locate(frames, 10, pyramid_levels=levels, skip_levels_at_top=skip)
    dt = time.perf_counter() - t0
    return {"value": n_frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d of the %d frames of the same %dx%d video, oracle.locate (L=%d,S=%d), %.1f s, host has %d cores"
                      % (n_frames, vid_u8.shape[0], vid_u8.shape[1], vid_u8.shape[2], levels, skip, dt, os.cpu_count()),
            "roi": roi}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from respmon_amd import _capi, device, synth
    from respmon_amd import dist as rdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one rank per GPU; the modulo only matters for the single-GPU dry run of this path
        # (RESPMON_BENCH_BACKEND=gloo, several ranks on one device), never for the 8-GPU node
        dev_index = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        backend_name = os.environ.get("RESPMON_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend_name == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend_name)
    else:
        torch.cuda.set_device(0)
    assert a.gpus == world, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (a.gpus, world)

    T, H, W = a.frames, a.height, a.width
    sharded = world > 1 and a.mode == "sharded"
    # config 4: an independent stream per GPU; sharded mode: the same buffer everywhere, each rank keeps its frame shard
    vid_u8 = synth.synth_breathing(T, H, W, seed=1234 + (0 if sharded else rank))
    if sharded:
        t_lo, t_hi = rdist.shard_frames(T, rank, world)
        vid_u8 = vid_u8[t_lo:t_hi]
    tdt = {"f64": torch.float64, "f32": torch.float32, "f16": torch.float16, "u8": torch.uint8}[a.in_dtype]
    dev_u8 = torch.from_numpy(vid_u8).cuda()
    if a.in_dtype == "u8":
        buf = dev_u8
    else:
        buf = torch.empty(tuple(dev_u8.shape), dtype=tdt, device="cuda")
        for t0 in range(0, buf.shape[0], 16):  # uint8_to_float in float64 (base.py:231), then the storage dtype
            buf[t0:t0 + 16] = (dev_u8[t0:t0 + 16].to(torch.float64) * (1.0 / 255)).to(tdt)
        del dev_u8
    torch.cuda.synchronize()

    lib = _capi.load()
    ctx = device.ctx()
    flags = _capi.RM_FLAG_NO_PRUNE if a.no_prune else 0
    kw = dict(pyramid_levels=a.levels, skip_levels_at_top=a.skip, flags=flags)

    from respmon_amd.base import _Backend
    backend = _Backend()

    def step():
        if world == 1:   # exactly RespiratoryMonitor.locate: one rm_locate call
            return backend.locate(buf, 10, 0.1, 1.0, 500, a.levels, a.skip, 0.7, 20, flags)
        if sharded:
            return rdist.locate_sharded(buf, T, 10, threshold=20, **kw)
        return rdist.locate_streams(buf, 10, threshold=20, **kw)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    roi = None
    pre_steps = max(0, a.prewarm_steps)
    for _ in range(pre_steps):   # clock ramp of a fresh box; not part of W or K
        roi = step()
    for _ in range(a.warmup):
        roi = step()
    barrier()
    _capi.check(lib, lib.rm_profile_enable(ctx, 1), "rm_profile_enable")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        roi = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ms = (ctypes.c_double * 4)()
    ncalls = ctypes.c_int()
    _capi.check(lib, lib.rm_profile_read(ctx, ms, ctypes.byref(ncalls)), "rm_profile_read")
    k_ms_total, k_calls = ms[0], ncalls.value     # frame-buffer kernel, HIP events inside the timed region
    tn = ctypes.c_int(0)
    _capi.check(lib, lib.rm_heat_sparse_tiles_needed(ctx, ctypes.byref(tn)), "rm_heat_sparse_tiles_needed")
    tiles_needed = tn.value                       # largest per-rank tile count of the last sparse exchange (world > 1)
    # phase breakdown: separate untimed pass (bracketing every phase costs ~10 us of stream idle time each)
    _capi.check(lib, lib.rm_profile_enable(ctx, 2), "rm_profile_enable")
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _capi.check(lib, lib.rm_profile_read(ctx, ms, ctypes.byref(ncalls)), "rm_profile_read")
    _capi.check(lib, lib.rm_profile_enable(ctx, 0), "rm_profile_enable")
    # same video kept as uint8 in HBM (what a camera delivers; kernels apply uint8_to_float on the fly):
    # bit-identical ROI at 1/8 of the frame-buffer bytes.  Reported beside the headline, never as `value`.
    alt = None
    if world == 1 and a.in_dtype != "u8" and not a.no_u8_alt:
        buf8 = torch.from_numpy(vid_u8).cuda()
        roi8 = None
        for _ in range(a.warmup):
            roi8 = backend.locate(buf8, 10, 0.1, 1.0, 500, a.levels, a.skip, 0.7, 20, flags)
        torch.cuda.synchronize()
        t8 = time.perf_counter()
        for _ in range(a.steps):
            roi8 = backend.locate(buf8, 10, 0.1, 1.0, 500, a.levels, a.skip, 0.7, 20, flags)
        torch.cuda.synchronize()
        e8 = time.perf_counter() - t8
        alt = {"frame_buffer_dtype": "u8", "value": T * a.steps / e8, "unit": "frames/s", "ms_per_step": e8 / a.steps * 1e3,
               "roi": roi8, "roi_equals_headline": list(roi8 or []) == list(roi or [])}
        del buf8
    # BASELINE config 4 is "calibration + ROI flow": the per-frame motion extraction (base.py:354-407, 'flow' method) on the
    # ROI just found -- Shi-Tomasi corners once, then pyramidal LK + mean flow + PCA per frame.  Latency bound
    # (no roofline fraction is meaningful); reported beside the headline, never part of `value`.
    roi_flow = None
    if world == 1 and roi is not None and not a.no_roi_flow:
        x, y, w, h = roi
        n_fl = min(T - 1, 60)
        crops = [backend.roi_to_uint8(torch.from_numpy(vid_u8[i]).cuda(), x, y, w, h) for i in range(n_fl + 1)]
        pts = backend.good_features_to_track(crops[0], 100, 0.3, 7, 7)      # base.py:91-94
        if pts is not None:
            motion = []
            torch.cuda.synchronize()
            tf = time.perf_counter()
            p = pts
            for i in range(n_fl):
                if len(p) == 0:
                    break
                p1, st = backend.calc_optical_flow_pyr_lk(crops[i], crops[i + 1], p, (15, 15), 2, (3, 10, 0.03))   # base.py:96-98
                mean, n_good = backend.mean_flow(p, p1, st)
                p = p1[st == 1].reshape(-1, 1, 2)
                if n_good:
                    motion.append([mean[0], mean[1]])
                if len(motion) >= 2:
                    backend.pca_reduce(np.array(motion, dtype=np.float32))
            torch.cuda.synchronize()
            dtf = (time.perf_counter() - tf) / max(n_fl, 1)
            roi_flow = {"roi": [x, y, w, h], "corners": int(len(pts)), "frames": n_fl, "ms_per_frame": dtf * 1e3,
                        "frames_per_s": 1.0 / dtf if dtf > 0 else None, "budget_ms_at_30fps": 33.3}
        else:
            roi_flow = {"roi": [x, y, w, h], "corners": 0}
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])

    dbg = (ctypes.c_longlong * 4)()
    _capi.check(lib, lib.rm_debug_counters(ctx, dbg, device.stream_ptr()), "rm_debug_counters")
    if rank == 0:
        frames_total = (1 if sharded else world) * T * a.steps
        # one read of the (rank-local) frame buffer + the heatmap
        b_alg = int(buf.shape[0]) * H * W * DT_BYTES[a.in_dtype] + H * W * 8
        k_ms = k_ms_total / max(k_calls, 1)
        achieved = b_alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "%s_%dx%dx%d" % (a.in_dtype, int(buf.shape[0]), H, W)
                traffic = tj.get(key, {}).get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Eulerian-calibration frames/sec on 1080p x 256 buffer; achieved HBM GB/s",
            "value": frames_total / elapsed, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "pre_warm_steps": pre_steps, "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Eulerian calibration + ROI (locate) on a %dx%dx%d frame buffer, %d-level Laplacian pyramid, "
                                   "skip %d, temporal FFT band-pass 0.1-1.0 Hz @10 fps; %s" % (T, H, W, a.levels, a.skip,
                                   "ONE buffer sharded by frame index over the GPUs: all-gather of the small pyramid, min/max all-reduce, "
                                   "one RCCL heatmap all-reduce" if sharded else "one independent stream per GPU + one RCCL exchange of the heatmaps"),
                       "frame_buffer_dtype": a.in_dtype, "frames": T, "height": H, "width": W, "prune": not a.no_prune},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": "frame-buffer pyrDown kernel (reads [T,H,W] once)", "kernel_ms": k_ms,
                         "algorithmic_bytes": b_alg},
            "phases_ms_per_step": {"frame_buffer_kernel": ms[0] / max(ncalls.value, 1), "pyramid_rest_and_temporal": ms[1] / max(ncalls.value, 1),
                                   "collapse_passes": ms[2] / max(ncalls.value, 1), "heatmap_to_roi": ms[3] / max(ncalls.value, 1)},
            "roi": roi,
            "heatmap_exchange": (rdist.LAST_EXCHANGE and {"sparse": "one all-gather of sparse packets (%d-tile cap, %.2f MB per rank)"
                                                          % (rdist.SPARSE_CAP_TILES, 8e-6 * (4 + rdist.SPARSE_CAP_TILES * 1025)),
                                                          "dense": "all-reduce(sum) of the [H,W] float64 heatmap"}[rdist.LAST_EXCHANGE])
                                if world > 1 else None,
            "heatmap_exchange_tiles_needed": tiles_needed if world > 1 else None,
            "alt_uint8_buffer": alt,
            "roi_flow": roi_flow,
            "collapse_pairs": {"total": dbg[0], "evaluated": dbg[1], "kept_for_sum": dbg[2], "store_capacity": dbg[3]},
        }
        if world == 1 and a.cpu_frames != 0:
            n_cpu = a.cpu_frames
            if n_cpu < 0:
                import psutil
                need = 6.0 * T * H * W * 8   # the materialising algorithm holds ~5-6 float64 [T,H,W] arrays
                n_cpu = T if psutil.virtual_memory().available > 1.3 * need else 64
            out["cpu_baseline"] = cpu_baseline(vid_u8, min(n_cpu, T), a.levels, a.skip)
            if min(n_cpu, T) == T:
                out["cpu_baseline"]["roi_equals_gpu"] = list(out["cpu_baseline"]["roi"] or []) == list(roi or [])
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
