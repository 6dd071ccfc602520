"""Multi-GPU orchestration: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

Mode B (BASELINE config 4, the weak-scaling figure): every rank calibrates its OWN [T,H,W] stream
(no data-path collective -- streams are independent units), then ONE all-reduce(sum) of the
float64 [H,W] heatmap gives the fused multi-stream heatmap every rank turns into the same ROI.

Mode A (one buffer sharded by frame index) is documented in DESIGN.md; its building blocks are
`all_reduce_heatmap` and `all_reduce_minmax`.

The functions take a `calibrate_fn` / `roi_fn` pair so that the collective sequencing can be tested
on CPU with gloo and a test double (tests/test_dist_gloo.py); on the GPU they default to the HIP path."""
import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


def all_reduce_heatmap(heat, group=None):
    """In-place sum of the [H,W] float64 heatmap over ranks (the single data collective of Mode B)."""
    dist = _dist()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(heat, op=dist.ReduceOp.SUM, group=group)
    return heat


def all_reduce_minmax(mn, mx, like, group=None):
    """Global (min, max) of raw over frame shards (Mode A)."""
    import torch
    dist = _dist()
    t = torch.tensor([-mn, mx], dtype=torch.float64, device=like.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return -float(t[0]), float(t[1])


def hip_calibrate(buf, fps, freq_min=0.1, freq_max=1.0, amplification=500, pyramid_levels=9, skip_levels_at_top=4,
                  temporal_threshold=0.7, flags=0):
    """rm_calibrate on the current device -> float64 [H,W] heatmap tensor (asynchronous)."""
    from . import _capi, device
    t = device.require_gpu()
    lib = _capi.load()
    T, H, W = buf.shape
    heat = t.empty((H, W), dtype=t.float64, device=buf.device)
    _capi.check(lib, lib.rm_calibrate(device.ctx(), device.ptr(buf), device.dtype_code(buf), T, H, W, float(fps),
                                      float(freq_min), float(freq_max), float(amplification), int(pyramid_levels),
                                      int(skip_levels_at_top), float(temporal_threshold), int(flags), device.ptr(heat), None,
                                      device.stream_ptr()), "rm_calibrate")
    return heat


def hip_heatmap_to_roi(heat, threshold=20):
    import ctypes
    from . import _capi, device
    lib = _capi.load()
    H, W = heat.shape
    xywh = (ctypes.c_int32 * 4)()
    rc = _capi.check(lib, lib.rm_heatmap_to_roi(device.ctx(), device.ptr(heat), H, W, int(threshold), xywh, None, None,
                                                device.stream_ptr()), "rm_heatmap_to_roi")
    return None if rc == _capi.RM_NO_CONTOUR else (int(xywh[0]), int(xywh[1]), int(xywh[2]), int(xywh[3]))


def locate_streams(buf, fps, threshold=20, group=None, calibrate_fn=hip_calibrate, roi_fn=hip_heatmap_to_roi, **kw):
    """Mode B step: local calibration -> all-reduce(sum) of the heatmap -> ROI of the fused heatmap.
    With one rank this is exactly RespiratoryMonitor.locate."""
    heat = calibrate_fn(buf, fps, **kw)
    all_reduce_heatmap(heat, group)
    return roi_fn(heat, threshold)


def shard_frames(T, rank, world):
    """Mode A frame-index shard [t0, t1) of rank `rank` (contiguous, sizes differ by at most one)."""
    base, rem = divmod(T, world)
    t0 = rank * base + min(rank, rem)
    return t0, t0 + base + (1 if rank < rem else 0)
