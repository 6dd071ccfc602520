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


def test_frame_shards_partition_the_buffer():
    from respmon_amd import dist as rdist
    for T in (1, 7, 128, 256, 513):
        for world in (1, 2, 3, 4, 8):
            spans = [rdist.shard_frames(T, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == T
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
