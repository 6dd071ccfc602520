"""Host-side logic without a GPU: the C-ABI library loads and exports every declared symbol, the temporal
operator (host function) matches the golden vectors, and the RespiratoryMonitor state machine reproduces the
reference's frame accounting (G6) when driven through a recording test double of the backend."""
import ctypes
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from respmon_amd import _capi
    lib = _capi.load()     # no GPU needed to load; raises if the library or a symbol is missing
    # the drop-in interface and the developer / test header: every symbol either declares
    header = "".join(open(os.path.join(ROOT, "include", f)).read() for f in ("respmon_hip.h", "respmon_hip_debug.h"))
    declared = set(re.findall(r"\b(rm_[a-z0-9_]+)\s*\(", header))
    public = open(os.path.join(ROOT, "include", "respmon_hip.h")).read()
    assert "rm_debug_" not in public and "test hook" not in public     # hooks live in respmon_hip_debug.h only
    assert declared, "no declarations found"
    for name in declared:
        assert hasattr(lib, name), "library does not export %s" % name
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    assert lib.rm_abi_version() == 1
    assert isinstance(lib.rm_last_error_string(), bytes)


def test_temporal_operator_host_function(golden):
    from respmon_amd import _capi
    lib = _capi.load()
    g = golden("g1_temporal_fft.npz")
    for i in range(int(g["ncases"])):
        n, fps, fmin, fmax, amp = g["meta%d" % i]
        if "M%d" % i not in g.files:
            continue
        n = int(n)
        M = np.empty((n, n))
        lo, hi = ctypes.c_int(), ctypes.c_int()
        rc = lib.rm_temporal_operator(n, fps, fmin, fmax, ctypes.c_void_p(M.ctypes.data), ctypes.byref(lo), ctypes.byref(hi))
        assert rc == 0 and np.abs(M - g["M%d" % i]).max() < 1e-15
        assert np.allclose((M @ g["x%d" % i].reshape(n, -1)) * amp, g["y%d" % i].reshape(n, -1), rtol=0, atol=1e-9)
    assert lib.rm_temporal_operator(0, 10.0, 0.1, 1.0, None, None, None) < 0       # bad argument -> negative code
    assert b"rm_temporal_operator" in lib.rm_last_error_string()


def test_no_cpu_fallback_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from respmon_amd import _capi, transforms
    with pytest.raises(_capi.RespmonError):
        transforms.uint8_to_float(np.zeros(4, np.uint8))
    from respmon_amd.base import RespiratoryMonitor
    with pytest.raises(_capi.RespmonError):
        RespiratoryMonitor.locate(np.zeros((4, 8, 8)), 10)


class RecordingBackend:
    """Test double: records the calls the state machine makes and answers with fixed values."""

    def __init__(self, roi=(12, 22, 28, 26), fail_first=0):
        self.roi, self.fail_first = roi, fail_first
        self.locate_calls, self.stored = [], []

    def alloc_buffer(self, T, H, W, dtype):
        return np.zeros((T, H, W))

    def bgr_to_gray(self, frame):
        return np.ascontiguousarray(frame[..., 0])

    def store_frame(self, buf, idx, gray):
        buf[idx] = gray * (1. / 255)
        self.stored.append(idx)

    def locate(self, buf, fps, fmin, fmax, amp, levels, skip, tthr, thr, flags=0):
        self.locate_calls.append((fps, fmin, fmax, amp, levels, skip, tthr, thr, float(buf.sum())))
        if len(self.locate_calls) <= self.fail_first:
            return None
        return self.roi

    def roi_mean(self, gray, x, y, w, h):
        return float(np.average(gray[y:y + h, x:x + w] * (1. / 255)))


def _monitor(frames, fps, backend, **kw):
    from respmon_amd import synth
    from respmon_amd.base import RespiratoryMonitor
    mon = RespiratoryMonitor(capture_target=synth.FakeCapture(frames, fps=fps), visualize=None, save_all_data=False,
                             run_on_init=False, backend=backend, **kw)
    mon.sync_to_fps = lambda: None
    return mon


def test_state_machine_frame_accounting_matches_reference_trace(golden):
    from respmon_amd import synth
    g = golden("g6_run_trace.npz")
    vid = synth.synth_breathing(150, 48, 64, seed=11)
    roi = tuple(int(v) for v in g["c2_roi"])
    be = RecordingBackend(roi=roi)
    mon = _monitor(vid, 30, be, motion_extraction_method="average")
    trace = []
    real_next = mon.next_frame

    def traced():
        trace.append((["initialize", "calibration", "measure", "error"].index(mon.state), mon.calibration_buffer_idx))
        return real_next()
    mon.next_frame = traced
    mon.run()
    assert np.array_equal(np.array(trace, dtype=np.int32), g["c2_trace"])     # same (state, buffer index) per frame
    assert be.stored == list(range(128))                                      # frames 1..128 fill the buffer
    fps, fmin, fmax, amp, levels, skip, tthr, thr, _ = be.locate_calls[0]
    assert [fps, fmin, fmax, tthr, thr] == [float(v) for v in g["c2_locate_kw"]]   # fps clamped to fps_limit = 10
    assert (amp, levels, skip) == (500, 9, 4)                                 # locate()'s defaults (base.py:549-550)
    assert (mon.x, mon.y, mon.w, mon.h) == roi and mon.fps == 10
    assert np.allclose(np.array(mon.data), g["c2_data"], rtol=1e-13, atol=0)
    assert np.array_equal(np.array(mon.t), g["c2_t"])


def test_state_machine_retries_when_locate_returns_none():
    from respmon_amd import synth
    vid = synth.synth_breathing(300, 24, 32, seed=3)
    be = RecordingBackend(fail_first=1)
    mon = _monitor(vid, 10, be)
    mon.run()
    assert len(be.locate_calls) == 2                    # base.py:451-454: buffer index reset, buffer refilled, retry
    assert be.stored == list(range(128)) * 2
    assert mon.state == 'measure' and len(mon.data) == 300 - (1 + 128 + 1 + 128 + 1)


def test_skip_calibration_config1_and_deque_cap(golden):
    from respmon_amd import synth
    g = golden("g6_run_trace.npz")
    frames = synth.synth_brightness_video(64, 240, 320)
    mon = _monitor(frames, 10, RecordingBackend())
    mon.skip_calibration(100, 80, 70, 51)
    assert mon.state == 'measure' and mon.peak_minimum_sample_distance == int(g["c1_peak_min_dist"])
    mon.run()
    assert np.allclose(np.array(mon.data), g["c1_data"], rtol=1e-13, atol=0) and np.array_equal(np.array(mon.t), g["c1_t"])
    # the measurement deques are capped at measure_buffer_length = 128 (base.py:473-475)
    mon2 = _monitor(synth.synth_brightness_video(200, 24, 32), 10, RecordingBackend())
    mon2.skip_calibration(2, 2, 8, 8)
    mon2.run()
    assert len(mon2.data) == 128 and len(mon2.t) == 128


def test_constructor_argument_contract():
    import pytest
    from respmon_amd import synth
    from respmon_amd.base import RespiratoryMonitor
    cap = synth.FakeCapture(synth.synth_brightness_video(2, 8, 8), fps=10)
    for bad in (dict(fps_limit=0), dict(save_calibration_image=1), dict(visualize="matplotlib"), dict(fig_size=(1,)),
                dict(error_reset_delay=-1), dict(save_all_data=None), dict(motion_extraction_method="median")):
        with pytest.raises(AssertionError):
            RespiratoryMonitor(capture_target=cap, run_on_init=False, backend=RecordingBackend(), **bad)


def test_tools_and_peaks(golden):
    from respmon_amd import peaks, tools
    g = golden("g7_misc.npz")
    for c, r in zip(g["rbb_in"], g["rbb_out"]):
        x, y, w, h, a = c
        assert tuple(tools.reduce_bounding_box(int(x), int(y), int(w), int(h), a)) == tuple(int(v) for v in r)
    from respmon_amd.transforms import butter_lowpass_filter
    assert np.array_equal(butter_lowpass_filter(g["sig"], 0.5, 10, 3), g["filt"])
    t = np.arange(128) / 10.0
    sig = np.sin(2 * np.pi * 0.3 * t)
    idx = peaks.indexes(sig, min_dist=10)
    assert len(idx) >= 3 and np.allclose(np.diff(t[idx]), 1 / 0.3, atol=0.11)
    p = peaks.gaussian_fit(t[20:40], peaks.gaussian(t[20:40], 2.0, 3.0, 0.5))
    assert np.allclose(p, [2.0, 3.0, 0.5], atol=1e-6)
    b = tools.Benchmarker(); b.add_tag("x"); b.tick_start("x"); b.tick_end("x")
    assert b.has_tag("x") and "x, " in b.get_report()


def test_bench_presets(monkeypatch):
    """bench.py host logic that needs no GPU: the --config presets (SURVEY 8 sizes P / Q / R) and explicit flags overriding them."""
    import sys
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.config, a.frames, a.height, a.width, a.levels, a.skip, a.in_dtype) == (1, "P", 256, 1080, 1920, 9, 4, "f64")
    assert a.steps * 1.0e-3 >= 0.08         # ~0.8 ms per step: the default timed region is about 0.1 s
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", "Q"])
    a = bench.parse()
    assert (a.frames, a.height, a.width, a.levels, a.skip, a.in_dtype) == (128, 720, 1280, 4, 2, "f64")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", "R", "--frames", "64", "--skip", "3"])
    a = bench.parse()
    assert (a.frames, a.height, a.width, a.levels, a.skip, a.in_dtype) == (64, 2160, 3840, 6, 3, "f16")


def test_library_does_not_read_the_environment():
    """VERDICT r2 Weak 12: no getenv anywhere in the product's native sources -- developer switches are rm_debug_set keys and
    flag bits, per context."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "respmon_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")):
            text = open(os.path.join(csrc, f)).read()
            assert not re.search(r"\bgetenv\s*\(", text), f
    for f in ("_capi.py", "base.py", "device.py", "dist.py", "transforms.py", "pyramid.py"):
        text = open(os.path.join(root, "respmon_amd", f)).read()
        assert "os.environ" not in text and "getenv" not in text, f


def test_measure_and_find_peaks_match_reference_fixture(golden):
    """SURVEY 8f row f2 / VERDICT r1 Next 7: G9 holds what the reference's own measure() / find_peaks() (base.py:312-352,
    run unbound on a namespace in the build container; filtfilt authentic, peakutils served by respmon_amd/peaks.py) make of
    three signals.  The drop-in's measure() must give the same filtered signal, peak indices and BPM."""
    from respmon_amd import synth
    g = golden("g9_measure.npz")
    for i in range(int(g["ncases"])):
        fps = float(g["fps%d" % i])
        mon = _monitor(synth.synth_brightness_video(2, 8, 8), 10, RecordingBackend())
        mon.fps = fps
        mon.data = list(g["data%d" % i]); mon.t = list(g["t%d" % i])
        mon.peak_minimum_sample_distance = int(np.floor(fps / mon.freq_max))       # base.py:171
        mon.freq = []
        mon.measure()
        assert np.allclose(mon.filtered_data, g["filtered%d" % i], rtol=1e-12, atol=1e-14)
        assert list(mon.peak_indices) == list(g["peaks%d" % i])
        assert np.allclose(mon.freq, g["freq%d" % i], rtol=1e-12)
    assert abs(float(g["freq0"][-1]) - 24.0) < 1.0          # config 1: 0.4 Hz brightness = 24 breaths per minute


def test_run_reports_bpm_for_config1(golden):
    """run() in 'average' mode on the config-1 brightness video (0.4 Hz): measure() fires once more than 12 samples exist
    (base.py:479-480) and the last estimate equals the reference's for the same data (G9 case 0)."""
    from respmon_amd import synth
    g = golden("g9_measure.npz")
    frames = synth.synth_brightness_video(64, 240, 320)
    mon = _monitor(frames, 10, RecordingBackend(), motion_extraction_method="average")
    mon.skip_calibration(100, 80, 70, 51)
    mon.run()
    assert len(mon.data) == 64 and len(mon.freq) > 0
    assert abs(mon.freq[-1] - float(g["freq0"][-1])) <= 1e-9 * float(g["freq0"][-1])
    assert abs(mon.freq[-1] - 24.0) < 1.0


def test_flag_constants_match_the_header():
    """respmon_amd/_capi.py mirrors the RM_FLAG_* values of include/respmon_hip.h (the ctypes boundary has no compiler to do it)."""
    import re
    from respmon_amd import _capi
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    text = open(os.path.join(inc, "respmon_hip.h")).read() + open(os.path.join(inc, "respmon_hip_debug.h")).read()
    flags = dict(re.findall(r"#define\s+(RM_FLAG_\w+)\s+(\d+)u", text))
    assert len(flags) >= 9
    for name, value in flags.items():
        assert getattr(_capi, name) == int(value), name
    assert len(set(flags.values())) == len(flags)          # one bit each
    assert all(int(v) & (int(v) - 1) == 0 for v in flags.values())


def test_multi_gpu_dry_run_plan_and_shard_coverage():
    """bench.py --gpus 8 --dry-run (no GPU): the per-rank plan a first 8-GPU run has to confirm -- and rm_shard_frames x 8 covers the
    T = 256 / 512 / ragged buffers exactly once, with the padded all-gather layout rm_locate_sharded uses."""
    import json
    import subprocess
    import sys
    from respmon_amd import _capi
    lib = _capi.load()
    for T in (256, 512, 257, 9):
        for world in (1, 2, 3, 8):
            seen = []
            sizes = []
            for r in range(world):
                t0, t1 = ctypes.c_int(), ctypes.c_int()
                assert lib.rm_shard_frames(T, r, world, ctypes.byref(t0), ctypes.byref(t1)) == 0
                seen += list(range(t0.value, t1.value))
                sizes.append(t1.value - t0.value)
            assert seen == list(range(T)), (T, world)                   # contiguous, in rank order, every frame once
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == (T + world - 1) // world       # cmax: what every rank sends
    assert lib.rm_shard_frames(4, 4, 4, ctypes.byref(ctypes.c_int()), ctypes.byref(ctypes.c_int())) < 0
    for mode, cfg in (("streams", "P"), ("sharded", "P"), ("sharded", "R")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--mode", mode, "--config", cfg],
                             capture_output=True, text=True, check=True).stdout
        plan = json.loads(out.strip().split("\n")[-1])
        assert plan["dry_run"] and plan["n_gpus"] == 8 and len(plan["ranks"]) == 8
        assert "--nproc-per-node 8" in plan["launch"] and "127.0.0.1" in plan["launch"]
        T = plan["workload"]["frames"]
        if mode == "sharded":
            assert [r["frames"] for r in plan["ranks"]] == [[T // 8 * i, T // 8 * (i + 1)] for i in range(8)]
            assert all(r["all_gather_send_bytes"] == (T // 8) * plan["workload"]["NP"] * 8 for r in plan["ranks"])
            assert plan["scaling"] == "strong"
        else:
            assert all(r["frames"] == [0, T] for r in plan["ranks"]) and plan["scaling"] == "weak"
            assert sorted(r["stream_seed"] for r in plan["ranks"]) == list(range(1234, 1242))
        assert any("ncclAllGather" in line for line in plan["step"])


def _bench_ranks(nproc, fault, extra_args=(), timeout=240):
    """bench.py as `nproc` ranks of a torch.distributed.run job on the CPU (gloo), with a fault injected (RESPMON_BENCH_FAULT)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.update({"RESPMON_BENCH_BACKEND": "gloo", "RESPMON_BENCH_FAULT": fault, "OMP_NUM_THREADS": "1"})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "1", "--warmup", "0", "--cpu-frames", "0"]
    p = subprocess.run(cmd + list(extra_args), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = []
    for line in p.stdout.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            lines.append(json.loads(line))
    return p, lines


def test_bench_multi_gpu_failure_leaves_one_json_line():
    """first contact with a multi-GPU node must be diagnosable.  Two gloo ranks on the CPU; rank 1 is killed right
    after the rendezvous (what a rank dying inside ncclCommInitRank looks like to the others).  Rank 0 -- stuck in the collective
    that follows, then told to stop by the launcher, or failing on its own -- still prints exactly ONE JSON line: n_gpus, "error",
    the stage it was in, and what every rank left behind (the stage it had reached, the injected fault)."""
    p, lines = _bench_ranks(2, "1:rendezvous_done:kill")
    assert p.returncode != 0
    assert len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    j = lines[0]
    assert j["n_gpus"] == 2 and j["value"] is None and j["error"] and j["steps"] == 1
    assert j["stage"] in ("rendezvous_done", "device"), j["stage"]
    assert j["earliest_stage_of_any_rank"] == "rendezvous_done"
    assert set(j["ranks"]) == {"0", "1"}
    assert j["ranks"]["1"]["stage"] == "rendezvous_done" and "fault injected" in (j["ranks"]["1"]["stderr_tail"] or "")
    # rank 0 stuck where no error ever reaches it (a hang inside a collective): the launcher's SIGTERM makes its watcher thread print
    p, lines = _bench_ranks(2, "1:rendezvous_done:kill,0:rendezvous_done:hang")
    assert p.returncode != 0 and len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    j = lines[0]
    assert j["n_gpus"] == 2 and "SIGTERM" in j["error"] and j["stage"] == "rendezvous_done"
    assert "fault injected" in (j["ranks"]["1"]["stderr_tail"] or "")
    # ... and the watchdog when nothing ends the job at all
    p, lines = _bench_ranks(2, "1:rendezvous_done:hang,0:rendezvous_done:hang", extra_args=("--stage-timeout", "3"))
    assert p.returncode != 0 and len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    assert "no progress" in lines[0]["error"] and lines[0]["stage"] == "rendezvous_done"
    # a rank that raises (a Python-level failure: IPC refused, out of memory) leaves its traceback for rank 0's line as well
    p, lines = _bench_ranks(2, "1:rendezvous_done:raise")
    assert p.returncode != 0 and len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    j = lines[0]
    assert j["n_gpus"] == 2 and j["error"]
    assert "fault injected at stage rendezvous_done" in (j["ranks"]["1"]["stderr_tail"] or "")


def test_bench_launcher_reports_and_retries_when_no_rank_prints(tmp_path):
    """`python bench.py --gpus 2` as its own launcher: the ranks die before any of them can print (here: no GPU at all, and rank 0 is
    killed at the device stage) -- the launcher still prints ONE JSON line with n_gpus, "error", the earliest stage, the per-rank
    records of the first attempt and of the retry on the torch.distributed collectives."""
    import subprocess
    env = dict(os.environ)
    env.update({"RESPMON_BENCH_BACKEND": "gloo", "RESPMON_BENCH_FAULT": "0:device:kill", "OMP_NUM_THREADS": "1"})
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-frames", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    lines = [json.loads(x) for x in p.stdout.splitlines() if x.strip().startswith("{")]
    assert p.returncode != 0 and len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    j = lines[0]
    assert j["n_gpus"] == 2 and j["value"] is None and j["error"]
    assert j["stage"] in ("device", "rendezvous_done"), j["stage"]
    assert j["first_attempt"]["ranks"]["0"]["stage"] == "device"
    assert j["retry_with_torch_distributed_collectives"] is not None


def test_committed_pmc_figures_follow_the_kernel_source_stamp():
    """profiles/hbm_traffic.json / valu_issue.json are reported only to a library built from the frame-buffer kernel
    sources they were measured on.  The library exports the stamp of its sources (rm_debug_kernel_source_stamp == the sha256 of
    csrc/Makefile's STAMP_SRCS, recomputed here); bench.py's look-up returns the figure for the matching stamp and (None, stale)
    for an edited one; every committed entry carries a stamp."""
    import hashlib
    import bench
    from respmon_amd import _capi
    lib = _capi.load()
    stamp = lib.rm_debug_kernel_source_stamp().decode()
    csrc = os.path.join(ROOT, "respmon_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    srcs = re.search(r"^STAMP_SRCS\s*=\s*(.+)$", mk, re.M).group(1).split()
    want = hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in srcs)).hexdigest()[:16]
    assert stamp == want == bench.library_kernel_stamp()
    table = {"f64_256x1080x1920": {"bytes_per_launch": 4.4e9, "kernel_source_sha": stamp}, "old": {"bytes_per_launch": 1.0}}
    assert bench.committed_figure(table, "f64_256x1080x1920", stamp, "bytes_per_launch") == (4.4e9, False)
    edited = stamp[:-1] + ("0" if stamp[-1] != "0" else "1")          # the hash of an edited header
    assert bench.committed_figure(table, "f64_256x1080x1920", edited, "bytes_per_launch") == (None, True)
    assert bench.committed_figure(table, "old", stamp, "bytes_per_launch") == (None, True)       # an unstamped figure is never reported
    assert bench.committed_figure(table, "absent", stamp, "bytes_per_launch") == (None, False)
    tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    for key, e in tj.items():
        assert e.get("kernel_source_sha"), key
    vj = json.load(open(os.path.join(ROOT, "profiles", "valu_issue.json")))
    assert set(vj["kernel_source_sha"]) == set(vj["valu_instructions_per_pixel"])


def test_library_exports_only_the_c_abi():
    """the helpers the translation units share are not visible outside the library (csrc/rm_exports.map): every dynamic
    symbol librespmon_hip.so defines is an rm_* entry point."""
    import subprocess
    from respmon_amd import _capi
    out = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    names = [line.split()[-1] for line in out.splitlines() if line.strip()]
    assert names and all(n.startswith("rm_") for n in names), [n for n in names if not n.startswith("rm_")][:10]
    assert set(_capi.SIGNATURES) <= set(names)
