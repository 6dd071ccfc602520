"""The un-staged RCCL branches of respmon_amd/dist.py on ONE GPU: a process group of one rank over the "nccl" backend (RCCL on
ROCm) with dist.COLLECTIVES_AT_WORLD_1, so that every collective of Mode A and Mode B is really issued on device tensors
(float64 MAX / SUM all-reduce, all_gather_into_tensor of the frame rows and of the sparse packets).  A collective over one rank
is the identity: ROI and heatmap must equal rm_locate / rm_calibrate bit for bit.  Runs in a child process: the
process group and the switch must not leak into the other tests."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, socket, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from respmon_amd import dist as rdist, synth
from respmon_amd.base import RespiratoryMonitor
T, H, W = 64, 272, 480
vid = synth.synth_breathing(T, H, W, seed=21)
buf = (torch.from_numpy(vid).cuda().to(torch.float64) * (1.0 / 255))
kw = dict(pyramid_levels=6, skip_levels_at_top=2)
roi_ref = RespiratoryMonitor.locate(buf, 10, **kw)
heat_ref = rdist.hip_calibrate(buf, 10, **kw).clone()
out = {"backend": dist.get_backend(), "roi_ref": list(roi_ref)}
calls = []
for name in ("all_reduce", "all_gather_into_tensor"):
    orig = getattr(dist, name)
    def wrap(*a, _o=orig, _n=name, **k):
        t = a[1] if _n == "all_gather_into_tensor" else a[0]
        calls.append((_n, bool(t.is_cuda), str(t.dtype)))
        return _o(*a, **k)
    setattr(dist, name, wrap)
rdist.COLLECTIVES_AT_WORLD_1 = True
# Mode B: sparse packets through all_gather_into_tensor, then the dense all-reduce(sum)
roi_b, fused_b = rdist.locate_streams(buf, 10, threshold=20, return_heatmap=True, **kw)
out["mode_b_sparse"] = {"roi": list(roi_b), "exchange": rdist.LAST_EXCHANGE, "heat_equal": bool(torch.equal(fused_b, heat_ref))}
roi_d, fused_d = rdist.locate_streams(buf, 10, threshold=20, return_heatmap=True, sparse=False, **kw)
out["mode_b_dense"] = {"roi": list(roi_d), "exchange": rdist.LAST_EXCHANGE, "heat_equal": bool(torch.equal(fused_d, heat_ref))}
# Mode A: all-gather of the frame rows, all-reduce(MAX) of {-min, max}, sparse exchange / all-reduce(SUM) of the heat sum
roi_a, heat_a = rdist.locate_sharded(buf, T, 10, threshold=20, return_heatmap=True, **kw)
out["mode_a_sparse"] = {"roi": list(roi_a), "exchange": rdist.LAST_EXCHANGE, "heat_equal": bool(torch.equal(heat_a, heat_ref))}
roi_a2, heat_a2 = rdist.locate_sharded(buf, T, 10, threshold=20, return_heatmap=True, sparse=False, **kw)
out["mode_a_dense"] = {"roi": list(roi_a2), "exchange": rdist.LAST_EXCHANGE, "heat_equal": bool(torch.equal(heat_a2, heat_ref))}
mn, mx = rdist.all_reduce_minmax(-1.5, 2.5, buf)
out["minmax"] = [mn, mx]
# host-side cost of the whole Mode B exchange at world 1 over RCCL (tools/time_mode_b.py reports it per round)
for fn, key in ((lambda: RespiratoryMonitor.locate(buf, 10, **kw), "locate_ms"),
                (lambda: rdist.locate_streams(buf, 10, threshold=20, **kw), "mode_b_sparse_ms"),
                (lambda: rdist.locate_streams(buf, 10, threshold=20, sparse=False, **kw), "mode_b_dense_ms")):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); out[key] = (time.perf_counter() - t0) / 50 * 1e3
out["calls"] = sorted(set(calls))
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def test_rccl_collectives_on_one_rank(tmp_path):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    assert p.returncode == 0 and lines, (p.stdout[-2000:], p.stderr[-4000:])
    r = json.loads(lines[-1][7:])
    assert r["backend"] == "nccl"
    for key, ex in (("mode_b_sparse", "sparse"), ("mode_b_dense", "dense"), ("mode_a_sparse", "sparse"), ("mode_a_dense", "dense")):
        assert r[key]["roi"] == r["roi_ref"], (key, r)
        assert r[key]["exchange"] == ex, (key, r)
    # Mode B sums one stream: the fused heatmap IS the stream's heatmap; Mode A with one rank equals rm_calibrate bit for bit
    for key in ("mode_b_sparse", "mode_b_dense", "mode_a_sparse", "mode_a_dense"):
        assert r[key]["heat_equal"], (key, r)
    assert r["minmax"] == [-1.5, 2.5]
    # every collective ran on DEVICE tensors (no host staging), in float64
    names = {c[0] for c in r["calls"]}
    assert names == {"all_reduce", "all_gather_into_tensor"}, r["calls"]
    assert all(c[1] and c[2] == "torch.float64" for c in r["calls"]), r["calls"]
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "rccl_world1.json"), "w") as f:
            json.dump(r, f)


CHILD_CABI = r'''
import ctypes, json, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
import torch
torch.cuda.set_device(0)
from respmon_amd import _capi, device, dist as rdist, synth
from respmon_amd.base import RespiratoryMonitor
lib = _capi.load()
T, H, W = 64, 272, 480
vid = synth.synth_breathing(T, H, W, seed=21)
buf = (torch.from_numpy(vid).cuda().to(torch.float64) * (1.0 / 255))
out = {}
for (L, S) in ((6, 2), (7, 4)):
    kw = dict(pyramid_levels=L, skip_levels_at_top=S)
    roi_ref = RespiratoryMonitor.locate(buf, 10, **kw)
    heat_ref = rdist.hip_calibrate(buf, 10, **kw).clone()
    # (a) one rank WITHOUT a library: the collectives are the identity
    _capi.check(lib, lib.rm_comm_init(device.ctx(), 0, 1, None), "rm_comm_init")
    roi0, fused0 = rdist.cabi_locate_streams(buf, 10, return_heatmap=True, **kw)
    # (b) one rank over a real RCCL communicator: every collective is issued by the library on the stream
    rank, world, seen = rdist.cabi_comm_init()
    res = {"roi_ref": list(roi_ref), "no_lib": {"roi": list(roi0), "heat_equal": bool(torch.equal(fused0, heat_ref))}, "rccl_ranks_seen": seen,
           "rank_world": [rank, world], "active": bool(rdist.cabi_comm_active())}
    for dense in (0, 1):
        device.debug_set("exchange_dense", dense)
        tag = "dense" if dense else "sparse"
        roi_b, fused_b = rdist.locate_streams(buf, 10, threshold=20, return_heatmap=True, **kw)      # routed to rm_locate_streams
        res["mode_b_" + tag] = {"roi": list(roi_b), "exchange": rdist.LAST_EXCHANGE, "heat_equal": bool(torch.equal(fused_b, heat_ref))}
        roi_a, heat_a = rdist.locate_sharded(buf, T, 10, threshold=20, return_heatmap=True, **kw)     # routed to rm_locate_sharded
        res["mode_a_" + tag] = {"roi": list(roi_a), "exchange": rdist.LAST_EXCHANGE, "heat_equal": bool(torch.equal(heat_a, heat_ref))}
    device.debug_set("exchange_dense", 0)
    if (L, S) == (6, 2):
        for fn, key in ((lambda: RespiratoryMonitor.locate(buf, 10, **kw), "locate_ms"),
                        (lambda: rdist.cabi_locate_streams(buf, 10, **kw), "cabi_mode_b_sparse_ms"),
                        (lambda: rdist.cabi_locate_sharded(buf, T, 10, **kw), "cabi_mode_a_sparse_ms")):
            for _ in range(10): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): fn()
            torch.cuda.synchronize(); res[key] = (time.perf_counter() - t0) / 50 * 1e3
    rdist.cabi_comm_destroy()
    res["active_after_destroy"] = bool(rdist.cabi_comm_active())
    out["L%%d_S%%d" %% (L, S)] = res
t0, t1 = ctypes.c_int(), ctypes.c_int()
spans = []
for r in range(3):
    _capi.check(lib, lib.rm_shard_frames(10, r, 3, ctypes.byref(t0), ctypes.byref(t1)), "rm_shard_frames")
    spans.append([t0.value, t1.value])
out["shard_frames_10_3"] = spans
out["shard_frames_py"] = [list(rdist.shard_frames(10, r, 3)) for r in range(3)]
print("RESULT " + json.dumps(out))
'''


def test_rccl_behind_the_c_abi_on_one_rank():
    """rm_comm_init / rm_locate_streams / rm_locate_sharded (include/respmon_hip.h): the library opens librccl itself, makes a real
    communicator of ONE rank and issues every collective of Mode A and Mode B on the stream (ncclAllGather of the frame rows and of
    the sparse packets, ncclAllReduce MAX / SUM).  A collective over one rank is the identity: ROI and heatmap equal rm_locate /
    rm_calibrate bit for bit, sparse and dense exchange, skip 2 and skip 4; ncclCommCount reports the rank."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", CHILD_CABI % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    assert p.returncode == 0 and lines, (p.stdout[-2000:], p.stderr[-4000:])
    r = json.loads(lines[-1][7:])
    for cfg in ("L6_S2", "L7_S4"):
        c = r[cfg]
        assert c["rccl_ranks_seen"] == 1 and c["rank_world"] == [0, 1] and c["active"] and not c["active_after_destroy"], c
        assert c["no_lib"]["roi"] == c["roi_ref"] and c["no_lib"]["heat_equal"], c
        for key, ex in (("mode_b_sparse", "sparse"), ("mode_b_dense", "dense"), ("mode_a_sparse", "sparse"), ("mode_a_dense", "dense")):
            assert c[key]["roi"] == c["roi_ref"] and c[key]["exchange"] == ex and c[key]["heat_equal"], (cfg, key, c)
    assert r["shard_frames_10_3"] == r["shard_frames_py"] == [[0, 4], [4, 7], [7, 10]]
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "rccl_cabi_world1.json"), "w") as f:
            json.dump(r, f)
