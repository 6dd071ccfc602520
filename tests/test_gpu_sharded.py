"""Frame-sharded calibration (Mode A, include/respmon_hip.h rm_shard_*) on the MI355X:
  * one rank: the staged path equals rm_locate / rm_calibrate bit for bit;
  * several ranks emulated in ONE process (a library context per rank, the three collectives done by hand):
    same ROI as the unsharded path and as the CPU oracle, heatmap within the association error of the time sum;
  * two real processes on the one GPU of the test box (gloo stages the tensors through the host; on the
    8-GPU node the same code runs over RCCL): dist.locate_sharded end to end."""
import ctypes
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from respmon_amd import _capi
    return _capi.load()


def _buffer(oracle, T, H, W, seed):
    import torch
    from respmon_amd import synth
    frames = oracle.uint8_to_float(synth.synth_breathing(T, H, W, seed=seed))
    return frames, torch.from_numpy(frames).cuda()


def test_one_rank_equals_locate_bit_for_bit(hip, oracle):
    import torch
    from respmon_amd import dist as rdist
    from respmon_amd.base import RespiratoryMonitor
    for (T, H, W, L, S, seed) in [(64, 135, 241, 8, 3, 5), (48, 270, 480, 9, 4, 6), (40, 64, 64, 5, 1, 24)]:
        frames, buf = _buffer(oracle, T, H, W, seed)
        roi, heat = rdist.locate_sharded(buf, T, 10, pyramid_levels=L, skip_levels_at_top=S, return_heatmap=True)
        assert roi == RespiratoryMonitor.locate(buf, 10, pyramid_levels=L, skip_levels_at_top=S)
        assert torch.equal(heat, rdist.hip_calibrate(buf, 10, pyramid_levels=L, skip_levels_at_top=S))
        assert roi == oracle.locate(frames, 10, pyramid_levels=L, skip_levels_at_top=S)
    # nothing filtered (skip >= levels - 1): zero heatmap -> NaN after normalisation -> no contour -> None (base.py:569-570)
    frames, buf = _buffer(oracle, 16, 32, 48, 9)
    assert oracle.locate(frames, 10, pyramid_levels=3, skip_levels_at_top=2) is None
    assert RespiratoryMonitor.locate(buf, 10, pyramid_levels=3, skip_levels_at_top=2) is None
    assert rdist.locate_sharded(buf, 16, 10, pyramid_levels=3, skip_levels_at_top=2) is None


class _Rank:
    """One emulated rank: its own library context, so the per-rank state of rm_shard_* is really separate."""

    def __init__(self, lib):
        from respmon_amd import _capi
        self.lib, self._capi = lib, _capi
        self.ctx = ctypes.c_void_p()
        _capi.check(lib, lib.rm_ctx_create(0, ctypes.byref(self.ctx)), "rm_ctx_create")

    def close(self):
        self.lib.rm_ctx_destroy(self.ctx)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_emulated_ranks_match_unsharded(hip, oracle, world):
    import torch
    from respmon_amd import _capi, device, dist as rdist
    from respmon_amd.base import RespiratoryMonitor
    lib = hip
    T, H, W, L, S = 67, 180, 320, 8, 3          # 67 frames: uneven shards for every world size
    frames, buf = _buffer(oracle, T, H, W, 31)
    sp = device.stream_ptr()
    n = ctypes.c_size_t()
    _capi.check(lib, lib.rm_shard_layout(H, W, L, S, ctypes.byref(n)), "rm_shard_layout")
    NP = int(n.value)
    ranks = [_Rank(lib) for _ in range(world)]
    try:
        spans = [rdist.shard_frames(T, r, world) for r in range(world)]
        lap_all = torch.empty((T, NP), dtype=torch.float64, device="cuda")
        for rk, (t0, t1) in zip(ranks, spans):        # stage 1 + "all-gather"
            local = buf[t0:t1].contiguous()
            _capi.check(lib, lib.rm_shard_pyramid(rk.ctx, device.ptr(local), _capi.RM_F64, t1 - t0, H, W, L, S, 0,
                                                  ctypes.c_void_p(lap_all[t0:t1].data_ptr()), sp), "rm_shard_pyramid")
        mms = []
        for rk, (t0, t1) in zip(ranks, spans):        # stage 2
            mm = torch.empty(2, dtype=torch.float64, device="cuda")
            _capi.check(lib, lib.rm_shard_collapse(rk.ctx, device.ptr(lap_all), T, t0, t1, H, W, 10.0, 0.1, 1.0, 500.0, L, S, 0.7, 0,
                                                   device.ptr(mm), sp), "rm_shard_collapse")
            mms.append(mm)
        mm = torch.stack(mms).max(dim=0).values        # "all-reduce(MAX)"
        total = torch.zeros((H, W), dtype=torch.float64, device="cuda")
        for rk in ranks:                               # stage 3 + "all-reduce(SUM)" in rank order
            hs = torch.empty((H, W), dtype=torch.float64, device="cuda")
            _capi.check(lib, lib.rm_shard_heat(rk.ctx, device.ptr(mm), 0.7, device.ptr(hs), sp), "rm_shard_heat")
            total += hs
        heat = torch.empty((H, W), dtype=torch.float64, device="cuda")
        xywh = (ctypes.c_int32 * 4)()
        rc = _capi.check(lib, lib.rm_shard_finish(ranks[0].ctx, device.ptr(total), T, H, W, 20, device.ptr(heat), xywh, sp),
                         "rm_shard_finish")
        roi = None if rc == _capi.RM_NO_CONTOUR else tuple(int(v) for v in xywh)
    finally:
        for rk in ranks:
            rk.close()
    ref_heat = rdist.hip_calibrate(buf, 10, pyramid_levels=L, skip_levels_at_top=S)
    assert roi == RespiratoryMonitor.locate(buf, 10, pyramid_levels=L, skip_levels_at_top=S)
    assert roi == oracle.locate(frames, 10, pyramid_levels=L, skip_levels_at_top=S)
    err = float((heat - ref_heat).abs().max() / ref_heat.abs().max())
    assert err <= 1e-12, err
    # the exact extrema found by the shards are the global ones
    masked, raw = oracle.eulerian_magnification_bandpass(frames, 10, 0.1, 1.0, 500, pyramid_levels=L, skip_levels_at_top=S)
    got_min, got_max = -float(mm[0]), float(mm[1])
    assert abs(got_min - raw.min()) <= 1e-9 * abs(raw.min()) and abs(got_max - raw.max()) <= 1e-9 * abs(raw.max())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, T, H, W, L, S):
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle import respmon_oracle as oracle
    from respmon_amd import dist as rdist, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = oracle.uint8_to_float(synth.synth_breathing(T, H, W, seed=41))
    t0, t1 = rdist.shard_frames(T, rank, world)
    local = torch.from_numpy(frames[t0:t1].copy()).cuda()
    roi, heat = rdist.locate_sharded(local, T, 10, pyramid_levels=L, skip_levels_at_top=S, return_heatmap=True)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), roi=np.array(roi if roi is not None else (-1, -1, -1, -1)),
             heat=heat.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_locate_sharded(hip, oracle, tmp_path):
    import torch
    import torch.multiprocessing as mp
    from respmon_amd import dist as rdist
    T, H, W, L, S = 65, 135, 240, 8, 3
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), T, H, W, L, S), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["roi"], r1["roi"]) and np.array_equal(r0["heat"], r1["heat"])
    frames, buf = _buffer(oracle, T, H, W, 41)
    assert tuple(int(v) for v in r0["roi"]) == oracle.locate(frames, 10, pyramid_levels=L, skip_levels_at_top=S)
    ref = rdist.hip_calibrate(buf, 10, pyramid_levels=L, skip_levels_at_top=S).cpu().numpy()
    assert np.abs(r0["heat"] - ref).max() <= 1e-12 * np.abs(ref).max()


# ---------------------------------------------------------------------------------------------------
# Mode B (one stream per GPU): sparse heatmap exchange == dense sum of the per-stream heatmaps
# ---------------------------------------------------------------------------------------------------
def test_sparse_heatmap_exchange_emulated_ranks(hip, oracle):
    """rm_heat_sparse_pack / rm_heat_sparse_merge_roi with the all-gather done by hand: the fused heatmap equals the
    rank-ordered dense sum bit for bit, the ROI equals the dense path's; overflow and missing bookkeeping fall back."""
    import torch
    from respmon_amd import _capi, device, dist as rdist
    lib = hip
    from respmon_amd import synth
    T, H, W, L, S = 128, 1080, 1920, 9, 4          # the pruning needs a long buffer to bite
    world, cap = 3, rdist.SPARSE_CAP_TILES
    sp = device.stream_ptr()
    pd = int(lib.rm_heat_sparse_packet_doubles(cap))
    heats, packets = [], []
    for r in range(world):
        buf = torch.from_numpy(synth.synth_breathing(T, H, W, seed=60 + r)).cuda()   # uint8 frame buffer (bit-identical results)
        heat = rdist.hip_calibrate(buf, 10, pyramid_levels=L, skip_levels_at_top=S)
        pk = torch.empty(pd, dtype=torch.float64, device="cuda")
        _capi.check(lib, lib.rm_heat_sparse_pack(device.ctx(), device.ptr(heat), H, W, cap, device.ptr(pk), sp), "pack")
        heats.append(heat); packets.append(pk)
    allp = torch.cat(packets)
    fused = torch.empty((H, W), dtype=torch.float64, device="cuda")
    xywh = (ctypes.c_int32 * 4)()
    rc = _capi.check(lib, lib.rm_heat_sparse_merge_roi(device.ctx(), device.ptr(allp), world, H, W, cap, 20, 0, device.ptr(fused), xywh, sp), "merge")
    dense = heats[0].clone()
    for h in heats[1:]:
        dense = dense + h                                   # rank order, like the merge kernel
    counts = [int(pk[:1].view(torch.int32)[0]) for pk in packets]
    assert rc in (_capi.RM_OK, _capi.RM_NO_CONTOUR), counts
    assert torch.equal(fused, dense)
    roi = None if rc == _capi.RM_NO_CONTOUR else tuple(int(v) for v in xywh)
    assert roi == rdist.hip_heatmap_to_roi(dense, 20)
    assert all(0 < c <= cap for c in counts), counts           # the sparse path was really taken
    # a packet that cannot hold its tiles -> every rank is told to fall back
    small = 2
    pd2 = int(lib.rm_heat_sparse_packet_doubles(small))
    pk2 = torch.empty(pd2, dtype=torch.float64, device="cuda")
    _capi.check(lib, lib.rm_heat_sparse_pack(device.ctx(), device.ptr(heats[-1]), H, W, small, device.ptr(pk2), sp), "pack")
    rc = _capi.check(lib, lib.rm_heat_sparse_merge_roi(device.ctx(), device.ptr(pk2), 1, H, W, small, 20, 0, device.ptr(fused), xywh, sp), "merge")
    assert rc == _capi.RM_SPARSE_FALLBACK
    # a heatmap the context has no pruning bookkeeping for (other geometry) -> fall back as well
    other = torch.rand((64, 96), dtype=torch.float64, device="cuda")
    pk3 = torch.empty(pd, dtype=torch.float64, device="cuda")
    _capi.check(lib, lib.rm_heat_sparse_pack(device.ctx(), device.ptr(other), 64, 96, cap, device.ptr(pk3), sp), "pack")
    f3 = torch.empty((64, 96), dtype=torch.float64, device="cuda")
    rc = _capi.check(lib, lib.rm_heat_sparse_merge_roi(device.ctx(), device.ptr(pk3), 1, 64, 96, cap, 20, 0, device.ptr(f3), xywh, sp), "merge")
    assert rc == _capi.RM_SPARSE_FALLBACK


@pytest.mark.parametrize("H,W,avg_T", [(1080, 1920, 0), (271, 1001, 0), (540, 962, 5)])
def test_sparse_merge_handmade_packets(hip, oracle, H, W, avg_T):
    """The merge kernels on hand-made packets (tests/test_emu_calibration.py::_handmade_packets): constant path for the
    tiles nobody sent (16-byte stores / odd-width scalar stores), per-pixel path for the others, partial edge tiles."""
    import torch
    from respmon_amd import _capi, device
    from tests.test_emu_calibration import _handmade_packets
    lib = hip
    world, cap = 4, 9
    allp_h, want = _handmade_packets(H, W, world, cap, seed=H + W, avg_T=avg_T)
    allp = torch.from_numpy(allp_h).cuda()
    fused = torch.full((H, W), -1.0, dtype=torch.float64, device="cuda")
    xywh = (ctypes.c_int32 * 4)()
    rc = _capi.check(lib, lib.rm_heat_sparse_merge_roi(device.ctx(), device.ptr(allp), world, H, W, cap, 20, avg_T, device.ptr(fused), xywh,
                                                       device.stream_ptr()), "merge")
    assert np.array_equal(fused.cpu().numpy(), want)
    u8 = oracle.float_to_uint8((want - want.min()) / (want.max() - want.min()))
    assert (tuple(int(v) for v in xywh) if rc == _capi.RM_OK else None) == oracle.roi_from_heatmap_u8(u8, 20)
    n = ctypes.c_int(-1)
    assert lib.rm_heat_sparse_tiles_needed(device.ctx(), ctypes.byref(n)) == 0 and n.value == world


def _worker_streams(rank, world, port, out_dir, T, H, W):
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle import respmon_oracle as oracle
    from respmon_amd import dist as rdist, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    buf = torch.from_numpy(synth.synth_breathing(T, H, W, seed=90 + rank)).cuda()   # uint8 frame buffer
    roi_s, fused_s = rdist.locate_streams(buf, 10, threshold=20, return_heatmap=True)                 # sparse exchange
    taken = rdist.hip_sparse_exchange_roi(rdist.hip_calibrate(buf, 10), 20)[0]
    roi_d, fused_d = rdist.locate_streams(buf, 10, threshold=20, sparse=False, return_heatmap=True)  # dense all-reduce
    local = rdist.hip_calibrate(buf, 10)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), roi_s=np.array(roi_s or (-1,) * 4), roi_d=np.array(roi_d or (-1,) * 4),
             fused_s=fused_s.cpu().numpy(), fused_d=fused_d.cpu().numpy(), local=local.cpu().numpy(), sparse_taken=np.array(taken))
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_locate_streams_sparse_equals_dense(hip, oracle, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker_streams, args=(2, port, str(tmp_path), 128, 1080, 1920), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["fused_s"], r1["fused_s"]) and np.array_equal(r0["roi_s"], r1["roi_s"])
    assert np.array_equal(r0["fused_s"], r0["local"] + r1["local"])          # rank-ordered sum of the per-stream heatmaps
    assert np.array_equal(r0["fused_s"], r0["fused_d"])                      # two ranks: the dense sum has only one order
    assert np.array_equal(r0["roi_s"], r0["roi_d"]) and r0["roi_s"][2] > 0
    assert bool(r0["sparse_taken"]) and bool(r1["sparse_taken"])


def _worker_sharded_big(rank, world, port, out_dir, T, H, W):
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from respmon_amd import dist as rdist, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    v8 = synth.synth_breathing(T, H, W, seed=123)
    t0, t1 = rdist.shard_frames(T, rank, world)
    local = torch.from_numpy(v8[t0:t1].copy()).cuda()          # uint8 frame shard
    roi, heat = rdist.locate_sharded(local, T, 10, return_heatmap=True)
    how = rdist.LAST_EXCHANGE
    roi_d, heat_d = rdist.locate_sharded(local, T, 10, return_heatmap=True, sparse=False)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), roi=np.array(roi or (-1,) * 4), heat=heat.cpu().numpy(),
             roi_d=np.array(roi_d or (-1,) * 4), heat_d=heat_d.cpu().numpy(), sparse=np.array(how == "sparse"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_locate_sharded_1080p_sparse_heat_sum(hip, tmp_path):
    """Mode A at a size where the pruning bites: the partial heat sums travel as sparse packets (summed in rank order,
    then / T) and give the same heatmap as the dense all-reduce and -- up to the association of the time sum -- as
    the unsharded calibration."""
    import torch
    import torch.multiprocessing as mp
    from respmon_amd import dist as rdist, synth
    T, H, W = 128, 1080, 1920
    port = _free_port()
    mp.spawn(_worker_sharded_big, args=(2, port, str(tmp_path), T, H, W), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert bool(r0["sparse"]) and bool(r1["sparse"])
    assert np.array_equal(r0["heat"], r1["heat"]) and np.array_equal(r0["roi"], r1["roi"])
    assert np.array_equal(r0["heat"], r0["heat_d"]) and np.array_equal(r0["roi"], r0["roi_d"])   # two ranks: one summation order
    buf = torch.from_numpy(synth.synth_breathing(T, H, W, seed=123)).cuda()
    ref = rdist.hip_calibrate(buf, 10)
    assert tuple(int(v) for v in r0["roi"]) == rdist.hip_heatmap_to_roi(ref, 20)
    ref = ref.cpu().numpy()
    assert np.abs(r0["heat"] - ref).max() <= 1e-12 * np.abs(ref).max()
