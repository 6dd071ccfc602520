"""Multi-process path on CPU (gloo, world_size 2): the collective sequencing of respmon_amd.dist --
one all-reduce(sum) of the [H,W] heatmap per step (Mode B) and the min/max all-reduce building block of
Mode A -- with the oracle standing in for the per-rank HIP calibration (test double; there is no CPU product path)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle import respmon_oracle as oracle
    from respmon_amd import dist as rdist, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vid = oracle.uint8_to_float(synth.synth_breathing(32, 60, 80, seed=1234 + rank))   # independent stream per rank

    def calibrate_double(buf, fps, **kw):
        masked, _raw = oracle.eulerian_magnification_bandpass(np.asarray(buf), fps, 0.1, 1.0, 500, pyramid_levels=6, skip_levels_at_top=2)
        return torch.from_numpy(np.average(masked, axis=0))

    def roi_double(heat, threshold):
        avg = heat.numpy()
        u8 = oracle.float_to_uint8((avg - avg.min()) / (avg.max() - avg.min()))
        return oracle.roi_from_heatmap_u8(u8, threshold)

    local = calibrate_double(vid, 10)
    roi = rdist.locate_streams(vid, 10, threshold=20, calibrate_fn=calibrate_double, roi_fn=roi_double)
    fused = local.clone()
    rdist.all_reduce_heatmap(fused)
    mn, mx = rdist.all_reduce_minmax(float(local.min()), float(local.max()), local)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), local=local.numpy(), fused=fused.numpy(),
             roi=np.array(roi if roi is not None else (-1, -1, -1, -1)), mn=mn, mx=mx)
    dist.barrier()
    dist.destroy_process_group()


def test_mode_b_heatmap_allreduce_world2(tmp_path, oracle):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert not np.array_equal(r0["local"], r1["local"])                       # different streams
    assert np.array_equal(r0["fused"], r1["fused"])                           # every rank holds the same fused heatmap
    assert np.array_equal(r0["fused"], r0["local"] + r1["local"])             # which is the sum of the per-stream heatmaps
    assert np.array_equal(r0["roi"], r1["roi"]) and r0["roi"][2] > 0          # and the same ROI
    avg = r0["fused"]
    u8 = oracle.float_to_uint8((avg - avg.min()) / (avg.max() - avg.min()))
    assert tuple(int(v) for v in r0["roi"]) == oracle.roi_from_heatmap_u8(u8, 20)
    assert r0["mn"] == r1["mn"] == min(r0["local"].min(), r1["local"].min())
    assert r0["mx"] == r1["mx"] == max(r0["local"].max(), r1["local"].max())


def test_single_rank_is_a_no_op():
    import torch
    from respmon_amd import dist as rdist
    h = torch.arange(12, dtype=torch.float64).reshape(3, 4)
    assert torch.equal(rdist.all_reduce_heatmap(h.clone()), h)
    assert rdist.all_reduce_minmax(-1.5, 2.5, h) == (-1.5, 2.5)


def test_exchange_policy_remembers_an_overflow():
    """Mode B / Mode A heatmap exchange (dist.ExchangePolicy): a refused sparse attempt grows the packet cap when the fullest rank
    fits 4 MB, and otherwise holds the dense all-reduce for DENSE_HOLD steps instead of paying for a refusal on every step."""
    from respmon_amd import dist as rdist
    pol = rdist.ExchangePolicy()
    assert pol.use_sparse() and pol.cap == rdist.SPARSE_CAP_TILES
    pol.overflowed(391)                       # bench.py's dense stream
    assert pol.cap == 512 and pol.use_sparse() and pol.dense_left == 0
    pol.overflowed(2040)                      # every tile of a 1080p heatmap: no packet holds that
    assert pol.dense_left == rdist.DENSE_HOLD and pol.cap == 512
    for _ in range(rdist.DENSE_HOLD):         # a dense step counts once its collective has returned (dense_step_done), not before
        assert not pol.use_sparse() and not pol.use_sparse()
        pol.dense_step_done()
    assert pol.use_sparse()                   # ... then one more sparse attempt
    assert rdist.exchange_policy(1080, 1920, "streams") is rdist.exchange_policy(1080, 1920, "streams")
    assert rdist.exchange_policy(1080, 1920, "streams") is not rdist.exchange_policy(1080, 1920, "sharded")
    # one policy per process group: ranks of different groups do not share a call history
    ga, gb = object(), object()
    pa = rdist.exchange_policy(1080, 1920, "streams", ga)
    assert pa is rdist.exchange_policy(1080, 1920, "streams", ga) and pa is not rdist.exchange_policy(1080, 1920, "streams", gb)
    assert pa is not rdist.exchange_policy(1080, 1920, "streams")
    pa.overflowed(2040)
    assert rdist.exchange_policy(1080, 1920, "streams", gb).dense_left == 0 and rdist.exchange_policy(1080, 1920, "streams", ga).dense_left
    rdist.reset_exchange_policy()
    assert rdist.exchange_policy(1080, 1920, "streams", ga).dense_left == 0


def test_frame_shards_partition_the_buffer():
    from respmon_amd import dist as rdist
    for T in (1, 7, 128, 256, 513):
        for world in (1, 2, 3, 4, 8):
            spans = [rdist.shard_frames(T, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == T
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---------------------------------------------------------------------------------------------------
# Mode A: one buffer sharded by frame index -- collective sequencing of dist.locate_sharded on CPU, with
# oracle-based doubles for the per-rank HIP stages (rm_shard_*)
# ---------------------------------------------------------------------------------------------------
class _OracleShardStages:
    def __init__(self, oracle):
        self.o = oracle

    def _sizes(self, H, W, L):
        hs, ws = [H], [W]
        for _ in range(1, L):
            hs.append((hs[-1] + 1) // 2)
            ws.append((ws[-1] + 1) // 2)
        return hs, ws

    def layout(self, H, W, L, S, flags=0):
        hs, ws = self._sizes(H, W, L)
        return 0 if S >= L - 1 else sum(hs[l] * ws[l] for l in range(S, L - 1))

    def pyramid(self, buf_local, L, S, flags, NP):
        import torch
        pyr = self.o.create_laplacian_video_pyramid(np.asarray(buf_local), L)
        rows = [pyr[l].reshape(pyr[l].shape[0], -1) for l in range(S, L - 1)]
        return torch.from_numpy(np.concatenate(rows, axis=1)) if rows else torch.zeros((buf_local.shape[0], 0), dtype=torch.float64)

    def collapse(self, lap_all, T, t0, t1, H, W, fps, fmin, fmax, amp, L, S, thr, flags):
        import torch
        hs, ws = self._sizes(H, W, L)
        lap = lap_all.numpy()
        pyr, o = [], 0
        for l in range(L):
            if S <= l < L - 1:
                n = hs[l] * ws[l]
                lev = lap[:, o:o + n].reshape(T, hs[l], ws[l])
                o += n
                pyr.append(self.o.temporal_bandpass_filter_fft(lev, fps, freq_min=fmin, freq_max=fmax, amplification_factor=amp))
            else:
                pyr.append(np.zeros((T, hs[l], ws[l])))
        raw = self.o.collapse_laplacian_video_pyramid(pyr)
        self.raw_local = raw[t0:t1]
        return torch.tensor([-self.raw_local.min(), self.raw_local.max()], dtype=torch.float64)

    def heat(self, mm, thr, H, W):
        import torch
        mn, mx = -float(mm[0]), float(mm[1])
        top = mx - (mx - mn) * thr
        masked = np.where(self.raw_local >= top, mn, self.raw_local)
        acc = np.zeros((H, W))
        for fr in masked:          # sequential in t, like np.average / np.add.reduce over axis 0
            acc = acc + fr
        return torch.from_numpy(acc)

    def finish(self, heat_sum, T, threshold):
        avg = heat_sum.numpy() / T
        u8 = self.o.float_to_uint8((avg - avg.min()) / (avg.max() - avg.min()))
        return self.o.roi_from_heatmap_u8(u8, threshold), heat_sum / T


def _worker_sharded(rank, world, port, out_dir, T):
    import sys
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle import respmon_oracle as oracle
    from respmon_amd import dist as rdist, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vid = oracle.uint8_to_float(synth.synth_breathing(T, 60, 80, seed=77))    # the SAME buffer on every rank ...
    t0, t1 = rdist.shard_frames(T, rank, world)
    local = torch.from_numpy(vid[t0:t1].copy())                               # ... of which this rank holds a frame shard
    roi, heat = rdist.locate_sharded(local, T, 10, pyramid_levels=6, skip_levels_at_top=2, threshold=20,
                                     stages=_OracleShardStages(oracle), return_heatmap=True)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), roi=np.array(roi if roi is not None else (-1, -1, -1, -1)),
             heat=heat.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T", [32, 33])      # even and uneven shards (padding path of the all-gather)
def test_mode_a_frame_sharded_world2(tmp_path, oracle, T):
    import torch.multiprocessing as mp
    from respmon_amd import synth
    port = _free_port()
    mp.spawn(_worker_sharded, args=(2, port, str(tmp_path), T), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["roi"], r1["roi"]) and np.array_equal(r0["heat"], r1["heat"])   # every rank agrees
    vid = oracle.uint8_to_float(synth.synth_breathing(T, 60, 80, seed=77))
    masked, _ = oracle.eulerian_magnification_bandpass(vid, 10, 0.1, 1.0, 500, pyramid_levels=6, skip_levels_at_top=2)
    avg = np.average(masked, axis=0)
    # the unsharded reference: identical ROI; the heatmap differs only by the association of the time sum
    assert tuple(int(v) for v in r0["roi"]) == oracle.locate(vid, 10, pyramid_levels=6, skip_levels_at_top=2)
    assert np.abs(r0["heat"] - avg).max() <= 1e-12 * np.abs(avg).max()
