"""Multi-GPU orchestration: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

Mode B (BASELINE config 4, the weak-scaling figure): every rank calibrates its OWN [T,H,W] stream
(no data-path collective -- streams are independent units), then the float64 [H,W] heatmaps are summed over the
ranks -- as ONE all-gather of sparse packets (1 MB per rank; xGMI ring collectives are per-link bound, so the
bytes are what to cut), or a dense all-reduce(sum) as the fallback -- and every rank turns the fused heatmap into
the same ROI.

Mode A (`locate_sharded`, BASELINE north_star): ONE [T,H,W] buffer split by frame index; rank r holds frames
shard_frames(T, r, world).  Three collectives, at the three points where frames meet:
  all-gather of the small Laplacian pyramid ([T, NP] float64, <= 22 MB at 1080p x 256),
  all-reduce(MAX) of {-min, max} (16 bytes), all-reduce(SUM) of the [H,W] heat sum (the heatmap collective).

The functions take a `calibrate_fn` / `roi_fn` pair so that the collective sequencing can be tested
on CPU with gloo and a test double (tests/test_dist_gloo.py); on the GPU they default to the HIP path."""
import numpy as np


def _dist():
    import torch.distributed as dist
    return dist


# Test / smoke switch: issue every collective of this module even when the group has ONE rank (a sum / max / gather over one
# rank is the identity, so results must not change).  It lets the un-staged RCCL branches below run on a single-GPU box
# (tests/test_gpu_rccl.py); never set in production -- at world 1 the collectives are skipped.
COLLECTIVES_AT_WORLD_1 = False


def _collective(group=None):
    """True when this step's collectives are to be issued: an initialised group of more than one rank (or the switch above)."""
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or COLLECTIVES_AT_WORLD_1


def all_reduce_heatmap(heat, group=None):
    """In-place sum of the [H,W] float64 heatmap over ranks (the single data collective of Mode B)."""
    dist = _dist()
    if _collective(group):
        _all_reduce(heat, dist.ReduceOp.SUM, group)
    return heat


def all_reduce_minmax(mn, mx, like, group=None):
    """Global (min, max) of raw over frame shards (Mode A)."""
    import torch
    dist = _dist()
    t = torch.tensor([-mn, mx], dtype=torch.float64, device=like.device)
    if _collective(group):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return -float(t[0]), float(t[1])


_SCRATCH = {}


def _scratch(name, shape, dtype, device):
    """Per-step temporaries of the multi-GPU step that nobody outside it sees (packets, the gathered packets, a heatmap the caller
    did not ask for): allocated once per (name, shape, dtype, device) instead of once per step."""
    from . import device as _device
    t = _device.require_gpu()
    key = (name, tuple(shape), dtype, str(device))
    b = _SCRATCH.get(key)
    if b is None:
        b = _SCRATCH[key] = t.empty(tuple(shape), dtype=dtype, device=device)
    return b


def hip_calibrate(buf, fps, freq_min=0.1, freq_max=1.0, amplification=500, pyramid_levels=9, skip_levels_at_top=4,
                  temporal_threshold=0.7, flags=0, return_minmax=False, out=None):
    """rm_calibrate on the current device -> float64 [H,W] heatmap tensor (asynchronous).
    return_minmax=True also returns (raw.min(), raw.max()) of transforms.py:185-187 (synchronises the stream)."""
    import ctypes
    from . import _capi, device
    t = device.require_gpu()
    lib = _capi.load()
    T, H, W = device.buffer_shape(buf)
    heat = out if out is not None else t.empty((H, W), dtype=t.float64, device=buf.device)
    mm = (ctypes.c_double * 2)() if return_minmax else None
    _capi.check(lib, lib.rm_calibrate(device.ctx(), device.ptr(buf), device.buffer_dtype_code(buf), T, H, W, float(fps),
                                      float(freq_min), float(freq_max), float(amplification), int(pyramid_levels),
                                      int(skip_levels_at_top), float(temporal_threshold), int(flags), device.ptr(heat), mm,
                                      device.stream_ptr()), "rm_calibrate")
    if return_minmax:
        return heat, (mm[0], mm[1])
    return heat


def hip_heatmap_to_roi(heat, threshold=20, clip_frame=False, labelling=None):
    """base.py:563-575 on a device heatmap.  clip_frame=True: cv2.findContours as OpenCV <= 3.1 did it (rm_set_contour_clip_frame).
    labelling: None = the library's rule, False / True = follow every border on the host / label the components on the device
    first (rm_set_contour_labelling); the ROI is the same."""
    import ctypes
    from . import _capi, device
    lib = _capi.load()
    H, W = heat.shape
    xywh = (ctypes.c_int32 * 4)()
    # per-call overrides: what the context had (rm_set_contour_clip_frame / rm_set_contour_labelling) is read first and put back
    ctx = device.ctx()
    had_clip, had_mode = ctypes.c_int(0), ctypes.c_int(-1)
    if clip_frame:
        _capi.check(lib, lib.rm_get_contour_clip_frame(ctx, ctypes.byref(had_clip)), "rm_get_contour_clip_frame")
        _capi.check(lib, lib.rm_set_contour_clip_frame(ctx, 1), "rm_set_contour_clip_frame")
    if labelling is not None:
        _capi.check(lib, lib.rm_get_contour_labelling(ctx, ctypes.byref(had_mode)), "rm_get_contour_labelling")
        _capi.check(lib, lib.rm_set_contour_labelling(ctx, 1 if labelling else 0), "rm_set_contour_labelling")
    try:
        rc = _capi.check(lib, lib.rm_heatmap_to_roi(ctx, device.ptr(heat), H, W, int(threshold), xywh, None, None,
                                                    device.stream_ptr()), "rm_heatmap_to_roi")
    finally:
        if clip_frame:
            lib.rm_set_contour_clip_frame(ctx, had_clip.value)
        if labelling is not None:
            lib.rm_set_contour_labelling(ctx, had_mode.value)
    return None if rc == _capi.RM_NO_CONTOUR else (int(xywh[0]), int(xywh[1]), int(xywh[2]), int(xywh[3]))


def contour_stats():
    """(components / contours the last ROI extraction of this process's context met, whether it ran on the labelled path)."""
    import ctypes
    from . import _capi, device
    lib = _capi.load()
    n, lab = ctypes.c_int(), ctypes.c_int()
    _capi.check(lib, lib.rm_contour_stats(device.ctx(), ctypes.byref(n), ctypes.byref(lab)), "rm_contour_stats")
    return n.value, bool(lab.value)


def roi_path():
    """RM_ROI_PATH_* of the last host contour stage of this process's context (include/respmon_hip_debug.h rm_debug_roi_path): 1 the
    one-blob rule, 2 every border followed, 3 labelled components + the borders that can win, 4 labelled + area bound, no border."""
    import ctypes
    from . import _capi, device
    v = ctypes.c_int(-1)
    _capi.check(_capi.load(), _capi.load().rm_debug_roi_path(device.ctx(), ctypes.byref(v)), "rm_debug_roi_path")
    return v.value


LAST_EXCHANGE = None      # "sparse" / "dense": how the last locate_streams summed the heatmaps (bench.py reports it)
SPARSE_CAP_TILES = 128   # tiles (64x16 px) a packet can carry: 1 MB per rank; the synthetic 1080p x 256 stream needs ~80


def hip_sparse_exchange_roi(heat, threshold, group=None, cap_tiles=SPARSE_CAP_TILES, avg_T=0, keep_fused=True):
    """All ranks' heatmaps summed in rank order through ONE all-gather of sparse packets (include/respmon_hip.h
    rm_heat_sparse_*), then the ROI stage.  Returns (status, roi, fused): status False = a packet overflowed on some
    rank (every rank sees that) and the caller must use the dense all-reduce.
    avg_T > 0: `heat` is this rank's partial time SUM of a frame-sharded buffer (rm_shard_heat); fused = sum / avg_T.
    keep_fused=False: the caller only wants the ROI; `fused` then lives in a scratch buffer the next call overwrites."""
    import ctypes
    from . import _capi, device
    t = device.require_gpu()
    dist = _dist()
    lib = _capi.load()
    rank, world = _world(group)
    H, W = heat.shape
    pd = int(lib.rm_heat_sparse_packet_doubles(cap_tiles))
    packet = _scratch("packet", (pd,), t.float64, heat.device)
    _capi.check(lib, lib.rm_heat_sparse_pack(device.ctx(), device.ptr(heat), H, W, cap_tiles, device.ptr(packet), device.stream_ptr()),
                "rm_heat_sparse_pack")
    if _collective(group):
        if _staged(packet, group):
            host = packet.cpu()
            allp = host.new_empty(world * pd)
            dist.all_gather_into_tensor(allp, host, group=group)
            allp = allp.to(heat.device)
        else:
            allp = _scratch("packets", (world * pd,), t.float64, heat.device)
            dist.all_gather_into_tensor(allp, packet, group=group)
    else:
        allp = packet
    fused = t.empty((H, W), dtype=t.float64, device=heat.device) if keep_fused else _scratch("fused", (H, W), t.float64, heat.device)
    xywh = (ctypes.c_int32 * 4)()
    rc = _capi.check(lib, lib.rm_heat_sparse_merge_roi(device.ctx(), device.ptr(allp), world, H, W, cap_tiles, int(threshold),
                                                       int(avg_T), device.ptr(fused), xywh, device.stream_ptr()),
                     "rm_heat_sparse_merge_roi")
    if rc == _capi.RM_SPARSE_FALLBACK:
        return False, None, None
    roi = None if rc == _capi.RM_NO_CONTOUR else (int(xywh[0]), int(xywh[1]), int(xywh[2]), int(xywh[3]))
    return True, roi, fused


def sparse_tiles_needed():
    """Largest per-rank tile count the last sparse merge on this context saw (identical on every rank: each one reads all the
    packet headers), whether or not the packets overflowed."""
    import ctypes
    from . import _capi, device
    lib = _capi.load()
    n = ctypes.c_int(0)
    _capi.check(lib, lib.rm_heat_sparse_tiles_needed(device.ctx(), ctypes.byref(n)), "rm_heat_sparse_tiles_needed")
    return n.value


class ExchangePolicy:
    """What a heatmap exchange remembers from one step to the next (per heatmap geometry).  A packet that overflowed tells every
    rank how many tiles the fullest rank needed: the cap grows to hold that (packets stay <= SPARSE_MAX_TILES * 8 KB = 4 MB), and a
    stream that needs more than that goes straight to the dense all-reduce for the next DENSE_HOLD steps instead of paying
    pack + all-gather + merge for a refusal on every step.  Every rank sees the same figures, so all of them switch together."""

    def __init__(self):
        self.cap = SPARSE_CAP_TILES
        self.dense_left = 0

    def use_sparse(self):
        """True: try the sparse form this step.  False: a dense step of the hold is due -- the caller reports it done with
        dense_step_done() AFTER its collective returned (a step that raised in between is not counted on this rank only)."""
        return self.dense_left <= 0

    def dense_step_done(self):
        if self.dense_left > 0:
            self.dense_left -= 1

    def overflowed(self, needed):
        want = (int(needed * 1.25) + 63) // 64 * 64
        if needed > 0 and want <= SPARSE_MAX_TILES:
            self.cap = max(self.cap, want)
        else:
            self.dense_left = DENSE_HOLD


SPARSE_MAX_TILES = 512    # 4 MB packets at most
DENSE_HOLD = 64           # steps a stream that does not fit stays on the dense all-reduce before the sparse form is tried again
_POLICY = {}


def exchange_policy(H, W, mode, group=None):
    """The policy of ONE process group (None: the default group) for one heatmap geometry and mode: ranks of a group share a call
    history, ranks of different groups do not, so the state is never shared between groups.  A group object that is created anew
    starts from the defaults (the entry keeps a reference to its group: no id is reused while the entry lives);
    reset_exchange_policy() forgets everything, e.g. after destroy_process_group()."""
    key = (id(group) if group is not None else 0, int(H), int(W), mode)
    ent = _POLICY.get(key)
    if ent is None or ent[0] is not group:
        ent = _POLICY[key] = (group, ExchangePolicy())
    return ent[1]


def reset_exchange_policy():
    _POLICY.clear()


def hip_sparse_tiles(heat, cap_tiles=SPARSE_CAP_TILES):
    """Tiles a sparse packet of the heatmap the LAST hip_calibrate on this context produced would have to carry (may
    exceed `cap_tiles`: that is the overflow that sends every rank to the dense all-reduce); None when the heatmap has no
    pruning bookkeeping (dense exchange only).  Diagnostic for bench.py; synchronises the stream."""
    from . import _capi, device
    t = device.require_gpu()
    lib = _capi.load()
    H, W = heat.shape
    pd = int(lib.rm_heat_sparse_packet_doubles(cap_tiles))
    packet = t.empty(pd, dtype=t.float64, device=heat.device)
    _capi.check(lib, lib.rm_heat_sparse_pack(device.ctx(), device.ptr(heat), H, W, cap_tiles, device.ptr(packet), device.stream_ptr()),
                "rm_heat_sparse_pack")
    count = int(packet[:1].view(t.int32)[0].item()) & 0xffffffff
    return None if count == 0xffffffff else count


def locate_streams(buf, fps, threshold=20, group=None, calibrate_fn=hip_calibrate, roi_fn=hip_heatmap_to_roi, sparse=None,
                   return_heatmap=False, **kw):
    """Mode B step: local calibration -> sum of the per-stream heatmaps over the ranks -> ROI of the fused heatmap.
    With one rank this is exactly RespiratoryMonitor.locate.
    The sum travels as ONE all-gather of sparse packets (a stream's heatmap is a single constant outside the few
    tiles that survive the pruning: 1 MB per rank instead of a 16.6 MB all-reduce at 1080p, summed in rank order);
    `sparse=False`, a test double for the calibration, or a packet overflow uses the dense all-reduce(sum)."""
    global LAST_EXCHANGE
    if calibrate_fn is hip_calibrate and roi_fn is hip_heatmap_to_roi and sparse is None and cabi_comm_active(group, _device_index_of(buf)):
        return cabi_locate_streams(buf, fps, threshold=threshold, return_heatmap=return_heatmap, **kw)
    if calibrate_fn is hip_calibrate and not return_heatmap:   # nobody sees this rank's own heatmap: no allocation per step
        from . import device as _device
        heat = hip_calibrate(buf, fps, out=_scratch("heat", buf.shape[1:3], _device.torch().float64, buf.device), **kw)
    else:
        heat = calibrate_fn(buf, fps, **kw)
    if sparse is None:
        sparse = calibrate_fn is hip_calibrate and roi_fn is hip_heatmap_to_roi and _collective(group)
    if sparse:
        pol = exchange_policy(heat.shape[0], heat.shape[1], "streams", group)
        if pol.use_sparse():
            ok, roi, fused = hip_sparse_exchange_roi(heat, threshold, group, cap_tiles=pol.cap, keep_fused=return_heatmap)
            if ok:
                LAST_EXCHANGE = "sparse"
                return (roi, fused) if return_heatmap else roi
            pol.overflowed(sparse_tiles_needed())
            pol = None      # (this step's dense all-reduce is the refused attempt's, not one of the hold)
    else:
        pol = None
    LAST_EXCHANGE = "dense"
    all_reduce_heatmap(heat, group)
    if pol is not None:
        pol.dense_step_done()
    roi = roi_fn(heat, threshold)
    return (roi, heat) if return_heatmap else roi


def shard_frames(T, rank, world):
    """Mode A frame-index shard [t0, t1) of rank `rank` (contiguous, sizes differ by at most one)."""
    base, rem = divmod(T, world)
    t0 = rank * base + min(rank, rem)
    return t0, t0 + base + (1 if rank < rem else 0)


# ---------------------------------------------------------------------------------------------------
# Mode A: one buffer sharded by frame index
# ---------------------------------------------------------------------------------------------------
def _world(group=None):
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _staged(tensor, group):
    """gloo moves host memory: stage device tensors through the host for it (single-GPU dry runs and tests);
    RCCL ("nccl") works on the device tensor itself."""
    dist = _dist()
    return tensor.is_cuda and dist.get_backend(group) == "gloo"


def _all_reduce(tensor, op, group=None):
    dist = _dist()
    if not _collective(group):
        return tensor
    if _staged(tensor, group):
        host = tensor.cpu()
        dist.all_reduce(host, op=op, group=group)
        tensor.copy_(host)
    else:
        dist.all_reduce(tensor, op=op, group=group)
    return tensor


def all_gather_frames(local, T, group=None):
    """[T_local, NP] rows of every rank -> [T, NP] in frame order (shards are contiguous and ordered by rank)."""
    import torch
    dist = _dist()
    rank, world = _world(group)
    if not _collective(group):
        return local
    spans = [shard_frames(T, r, world) for r in range(world)]
    counts = [b - a for a, b in spans]
    width = local.shape[1]
    cmax = max(counts)
    work = local
    if _staged(local, group):
        work = local.cpu()
    if cmax != counts[rank]:   # uneven shards: pad to the longest, compact afterwards
        pad = work.new_zeros((cmax, width))
        pad[:counts[rank]] = work
        work = pad
    out = work.new_empty((world * cmax, width))
    dist.all_gather_into_tensor(out, work.contiguous(), group=group)
    if any(c != cmax for c in counts):
        out = torch.cat([out[r * cmax:r * cmax + counts[r]] for r in range(world)], dim=0)
    return out.to(local.device) if out.device != local.device else out


class HipShardStages:
    """The per-rank stages of the frame-sharded calibration on the current HIP device (include/respmon_hip.h
    rm_shard_*).  tests/test_dist_gloo.py substitutes an oracle-based double to check the collective sequencing
    on CPU."""

    def __init__(self):
        from . import _capi, device
        self._capi, self._device = _capi, device
        self.t = device.require_gpu()
        self.lib = _capi.load()

    def layout(self, H, W, levels, skip, flags=0):
        import ctypes
        n = ctypes.c_size_t()
        self._capi.check(self.lib, self.lib.rm_shard_layout_flags(H, W, levels, skip, int(flags), ctypes.byref(n)), "rm_shard_layout_flags")
        return int(n.value)

    def pyramid(self, buf_local, levels, skip, flags, NP):
        d, t = self._device, self.t
        Tl, H, W = d.buffer_shape(buf_local)
        lap = t.empty((Tl, NP), dtype=t.float64, device=buf_local.device)
        if NP:
            self._capi.check(self.lib, self.lib.rm_shard_pyramid(d.ctx(), d.ptr(buf_local), d.buffer_dtype_code(buf_local), Tl, H, W, levels,
                                                                 skip, int(flags), d.ptr(lap), d.stream_ptr()), "rm_shard_pyramid")
        return lap

    def collapse(self, lap_all, T, t0, t1, H, W, fps, fmin, fmax, amp, levels, skip, thr, flags):
        d, t = self._device, self.t
        mm = t.empty(2, dtype=t.float64, device=lap_all.device)
        self._capi.check(self.lib, self.lib.rm_shard_collapse(d.ctx(), d.ptr(lap_all) if lap_all.numel() else None, T, t0, t1, H, W,
                                                              float(fps), float(fmin), float(fmax), float(amp), levels, skip, float(thr),
                                                              int(flags), d.ptr(mm), d.stream_ptr()), "rm_shard_collapse")
        return mm

    def heat(self, negmin_max, thr, H, W):
        d, t = self._device, self.t
        hs = t.empty((H, W), dtype=t.float64, device=negmin_max.device)
        self._capi.check(self.lib, self.lib.rm_shard_heat(d.ctx(), d.ptr(negmin_max), float(thr), d.ptr(hs), d.stream_ptr()),
                         "rm_shard_heat")
        return hs

    def finish(self, heat_sum, T, threshold):
        import ctypes
        d, t = self._device, self.t
        H, W = heat_sum.shape
        heat = t.empty((H, W), dtype=t.float64, device=heat_sum.device)
        xywh = (ctypes.c_int32 * 4)()
        rc = self._capi.check(self.lib, self.lib.rm_shard_finish(d.ctx(), d.ptr(heat_sum), T, H, W, int(threshold), d.ptr(heat), xywh,
                                                                 d.stream_ptr()), "rm_shard_finish")
        roi = None if rc == self._capi.RM_NO_CONTOUR else (int(xywh[0]), int(xywh[1]), int(xywh[2]), int(xywh[3]))
        return roi, heat


def locate_sharded(buf_local, T, fps, freq_min=0.1, freq_max=1.0, amplification=500, pyramid_levels=9, skip_levels_at_top=4,
                   temporal_threshold=0.7, threshold=20, flags=0, group=None, stages=None, return_heatmap=False, sparse=None):
    """Mode A step: RespiratoryMonitor.locate (base.py:547-601) of ONE [T,H,W] calibration buffer whose frames
    shard_frames(T, rank, world) live on this rank as `buf_local`.  Every rank returns the same ROI."""
    dist = _dist()
    rank, world = _world(group)
    t0, t1 = shard_frames(T, rank, world)
    if buf_local.shape[0] != t1 - t0:
        raise ValueError("rank %d holds %d frames, shard_frames(%d, %d, %d) says %d" % (rank, buf_local.shape[0], T, rank, world, t1 - t0))
    if t1 - t0 < 1:
        raise ValueError("every rank needs at least one frame (T=%d, world=%d)" % (T, world))
    if stages is None and sparse is None and cabi_comm_active(group, _device_index_of(buf_local)):
        return cabi_locate_sharded(buf_local, T, fps, freq_min, freq_max, amplification, pyramid_levels, skip_levels_at_top,
                                   temporal_threshold, threshold, flags, return_heatmap)
    st = stages if stages is not None else HipShardStages()
    H, W = int(buf_local.shape[1]), int(buf_local.shape[2])
    L, S = int(pyramid_levels), int(skip_levels_at_top)
    NP = st.layout(H, W, L, S, flags)   # doubles per frame the ranks exchange: G_S (default) or the Laplacian levels S .. L-2
    lap_local = st.pyramid(buf_local, L, S, flags, NP)
    lap_all = all_gather_frames(lap_local, T, group) if NP else lap_local
    mm = st.collapse(lap_all, T, t0, t1, H, W, fps, freq_min, freq_max, amplification, L, S, temporal_threshold, flags)
    _all_reduce(mm, dist.ReduceOp.MAX, group)
    heat_sum = st.heat(mm, temporal_threshold, H, W)
    if sparse is None:
        sparse = stages is None and _collective(group)
    global LAST_EXCHANGE
    if sparse:   # the partial heat sums are one constant outside a few tiles too: sparse all-gather instead of a 16.6 MB all-reduce
        pol = exchange_policy(H, W, "sharded", group)
        if pol.use_sparse():
            ok, roi, heat = hip_sparse_exchange_roi(heat_sum, threshold, group, cap_tiles=pol.cap, avg_T=T, keep_fused=return_heatmap)
            if ok:
                LAST_EXCHANGE = "sparse"
                return (roi, heat) if return_heatmap else roi
            pol.overflowed(sparse_tiles_needed())
            pol = None
    else:
        pol = None
    LAST_EXCHANGE = "dense"
    _all_reduce(heat_sum, dist.ReduceOp.SUM, group)
    if pol is not None:
        pol.dense_step_done()
    roi, heat = st.finish(heat_sum, T, threshold)
    return (roi, heat) if return_heatmap else roi


# ---------------------------------------------------------------------------------------------------
# RCCL behind the C-ABI (include/respmon_hip.h rm_comm_* / rm_locate_streams / rm_locate_sharded): one call per step, one host
# synchronisation, the collectives issued by the library on the caller's stream.  torch.distributed only ships the 128-byte
# communicator id once.  locate_streams / locate_sharded take this path when the context of the current device holds a communicator
# of the group's size (cabi_comm_init); gloo dry runs and test doubles keep the torch.distributed path above.
# ---------------------------------------------------------------------------------------------------
_CABI_COMM = {}   # device index -> (rank, world, the torch.distributed group the communicator was made from)


def _device_index_of(buf):
    """Device index of a frame buffer tensor (None for anything else: the current device is then looked at)."""
    dev = getattr(buf, "device", None)
    return dev.index if (dev is not None and getattr(dev, "type", "") == "cuda") else None


def _group_key(group):
    """None and torch.distributed.group.WORLD name the same ranks: one key for both, so that a communicator made with group=None
    serves locate_streams(group=dist.group.WORLD) and the reverse (the call used to drop to the torch.distributed
    path silently).  Any other group is its own key (object identity: a subgroup of the same size and rank is another group)."""
    if group is None:
        return None
    try:
        world_group = _dist().group.WORLD
    except Exception:   # noqa: BLE001 -- no torch.distributed in this process
        world_group = None
    return None if (world_group is not None and group is world_group) else group


def cabi_comm_init(group=None):
    """Create the library's RCCL communicator for the current device from an initialised torch.distributed group (backend nccl):
    rank 0 makes the id (rm_comm_unique_id), the group broadcasts it, every rank calls rm_comm_init.  Returns (rank, world,
    rccl_ranks_seen) -- rccl_ranks_seen is ncclCommCount of the new communicator."""
    import ctypes
    from . import _capi, device
    t = device.require_gpu()
    dist = _dist()
    lib = _capi.load()
    rank, world = _world(group)
    ctx = device.ctx()
    idbuf = (ctypes.c_ubyte * _capi.RM_COMM_ID_BYTES)()
    if rank == 0:
        _capi.check(lib, lib.rm_comm_unique_id(idbuf), "rm_comm_unique_id")
    if world > 1:
        dev = t.device("cuda", t.cuda.current_device()) if dist.get_backend(group) != "gloo" else t.device("cpu")
        tid = t.tensor(list(idbuf), dtype=t.uint8, device=dev)
        dist.broadcast(tid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(tid.cpu().tolist())
        ctypes.memmove(idbuf, raw, _capi.RM_COMM_ID_BYTES)
    _capi.check(lib, lib.rm_comm_init(ctx, rank, world, idbuf), "rm_comm_init")
    r, w, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _capi.check(lib, lib.rm_comm_info(ctx, ctypes.byref(r), ctypes.byref(w), ctypes.byref(n)), "rm_comm_info")
    _CABI_COMM[t.cuda.current_device()] = (r.value, w.value, _group_key(group))
    return r.value, w.value, n.value


def cabi_comm_destroy():
    from . import _capi, device
    t = device.require_gpu()
    _capi.load().rm_comm_destroy(device.ctx())
    _CABI_COMM.pop(t.cuda.current_device(), None)


def cabi_comm_active(group=None, device_index=None):
    """Does the library context of `device_index` (default: the current device) hold a communicator made from `group`?"""
    from . import device
    t = device.torch()
    if not t.cuda.is_available():
        return False
    have = _CABI_COMM.get(t.cuda.current_device() if device_index is None else device_index)
    # the communicator serves the group it was made from, and only that one: another group of the same size and rank (a subgroup, a
    # group created after destroy_process_group) would put the wrong ranks into the collective
    return have is not None and have[2] is _group_key(group) and have[:2] == _world(group)


def _cabi_step(fn_name, buf, T, fps, freq_min, freq_max, amplification, pyramid_levels, skip_levels_at_top, temporal_threshold, threshold,
               flags, return_heatmap):
    import ctypes
    from . import _capi, device
    global LAST_EXCHANGE
    t = device.require_gpu()
    lib = _capi.load()
    _, H, W = device.buffer_shape(buf)
    heat = t.empty((H, W), dtype=t.float64, device=buf.device) if return_heatmap else None
    xywh = (ctypes.c_int32 * 4)()
    how = ctypes.c_int(0)
    rc = _capi.check(lib, getattr(lib, fn_name)(device.ctx(buf.device.index), device.ptr(buf), device.buffer_dtype_code(buf), int(T), H, W, float(fps), float(freq_min),
                                               float(freq_max), float(amplification), int(pyramid_levels), int(skip_levels_at_top),
                                               float(temporal_threshold), int(threshold), int(flags), device.ptr(heat), xywh, ctypes.byref(how),
                                               device.stream_ptr()), fn_name)
    LAST_EXCHANGE = {_capi.RM_EXCHANGE_SPARSE: "sparse", _capi.RM_EXCHANGE_DENSE: "dense"}.get(how.value)
    roi = None if rc == _capi.RM_NO_CONTOUR else (int(xywh[0]), int(xywh[1]), int(xywh[2]), int(xywh[3]))
    return (roi, heat) if return_heatmap else roi


def cabi_locate_streams(buf, fps, freq_min=0.1, freq_max=1.0, amplification=500, pyramid_levels=9, skip_levels_at_top=4, temporal_threshold=0.7,
                        threshold=20, flags=0, return_heatmap=False):
    """Mode B step in ONE C-ABI call (rm_locate_streams)."""
    return _cabi_step("rm_locate_streams", buf, buf.shape[0], fps, freq_min, freq_max, amplification, pyramid_levels, skip_levels_at_top,
                      temporal_threshold, threshold, flags, return_heatmap)


def cabi_locate_sharded(buf_local, T, fps, freq_min=0.1, freq_max=1.0, amplification=500, pyramid_levels=9, skip_levels_at_top=4,
                        temporal_threshold=0.7, threshold=20, flags=0, return_heatmap=False):
    """Mode A step in ONE C-ABI call (rm_locate_sharded); buf_local holds the frames rm_shard_frames(T, rank, world)."""
    return _cabi_step("rm_locate_sharded", buf_local, T, fps, freq_min, freq_max, amplification, pyramid_levels, skip_levels_at_top,
                      temporal_threshold, threshold, flags, return_heatmap)
