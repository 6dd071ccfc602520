"""bench.py -- Eulerian-calibration frames/sec on a 1080p x 256 buffer (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config P|Q|R] [--mode streams|sharded]

One step = one locate() (calibration + ROI) over one [T,H,W] frame buffer already resident in HBM.
N>1: one process per GPU.  When the script is started WITHOUT a launcher (`python bench.py --gpus 8`, WORLD_SIZE unset) it
re-executes itself through `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`; started
by the launcher (RANK / LOCAL_RANK / WORLD_SIZE in the environment) it is one rank of that job.  Default mode "streams"
(Mode B of respmon_amd/dist.py, BASELINE config 4): an independent stream per GPU, weak scaling, one RCCL exchange of the
[H,W] heatmaps per step (an all-gather of sparse packets; dense all-reduce(sum) as the fallback).  --mode sharded: ONE
buffer split by frame index over the GPUs (Mode A, strong scaling).  Rank 0 prints ONE JSON line.

The default single-GPU P line also carries `configs`: summaries of BASELINE configs 2 (Q), 5 (R), 3 (F: LK on a 256x256 ROI with
1000 points) and of P with a float32 frame buffer, each measured in the same process after the headline and checked against the
CPU oracle (--no-configs skips them).

--config picks the workload (SURVEY 8 sizes):  P = 256 x 1080p, L=9, S=4, float64 buffer (the metric's own configuration,
default);  Q = 128 x 720p, L=4, S=2, float64 (BASELINE config 2);  R = 512 x 4K, L=6, S=2, float16 buffer (config 5).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
DT_BYTES = {"f64": 8, "f32": 4, "f16": 2, "u8": 1, "bgr8": 3}   # bgr8: [T,H,W,3] uint8 as captured (RM_BGR8; the synthetic video in three equal planes)
CONFIGS = {   # frames, height, width, levels, skip, frame-buffer dtype
    "P": (256, 1080, 1920, 9, 4, "f64"),
    "Q": (128, 720, 1280, 4, 2, "f64"),
    "R": (512, 2160, 3840, 6, 2, "f16"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (100 x ~1 ms: the timed region is >= 0.1 s)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="P", choices=sorted(CONFIGS), help="workload preset (see the module docstring); the "
                    "explicit size flags below override its fields")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--no-bgr-alt", action="store_true", help="skip the [T,H,W,3] uint8 BGR frame-buffer leg of the uint8 summary")
    ap.add_argument("--in-dtype", default=None, choices=list(DT_BYTES), help="device frame-buffer element type; "
                    "f64 is the reference's calibration_buffer dtype (base.py:119)")
    ap.add_argument("--levels", type=int, default=None)
    ap.add_argument("--skip", type=int, default=None)
    ap.add_argument("--prewarm-steps", type=int, default=None, help="untimed steps run BEFORE the W warmup steps (a fixed count so that "
                    "all ranks issue the same collectives): a fresh box starts in a low power state and its clocks take a few hundred ms "
                    "of load to settle -- measured 1.18 -> 1.14 ms/step (reported as pre_warm_steps; 0 disables; default ~0.7 s of work)")
    ap.add_argument("--mode", default="streams", choices=["streams", "sharded"], help="N>1 only.  streams (default, BASELINE config 4): one "
                    "independent [T,H,W] stream per GPU + one heatmap exchange, weak scaling.  sharded: ONE [T,H,W] buffer split by frame "
                    "index over the GPUs (respmon_amd.dist.locate_sharded), strong scaling")
    ap.add_argument("--no-prune", action="store_true")
    ap.add_argument("--video", default="breathing", choices=["breathing", "dense", "noise", "blobs16"], help="synthetic stream of the timed steps "
                    "(developer option: breathing is the metric's video; the others are the data-dependence / worst-case streams)")
    ap.add_argument("--debug-set", action="append", default=[], metavar="KEY=VALUE", help="developer switch of the library context "
                    "(include/respmon_hip.h rm_debug_set), e.g. temporal_wide=0; listed in the JSON line as debug_set")
    ap.add_argument("--no-extras", action="store_true", help="skip everything outside the contract line: uint8-buffer variant, ROI "
                    "flow, no-prune / dense-stream data-dependence figures")
    ap.add_argument("--no-u8-alt", action="store_true", help="skip the extra measurement with a uint8 frame buffer")
    ap.add_argument("--no-roi-flow", action="store_true", help="skip the per-frame ROI optical-flow measurement")
    ap.add_argument("--no-data-dependence", action="store_true", help="skip the no-prune and dense-stream measurements")
    ap.add_argument("--no-batches", action="store_true", help="skip the four extra batches of --steps behind the timed one (ms_per_step_batches)")
    ap.add_argument("--no-worst-case", action="store_true", help="skip the worst-case streams (full-frame noise, sixteen blobs) of `worst_case`")
    ap.add_argument("--no-configs", action="store_true", help="skip the summaries of the other BASELINE configs (Q, R, F, and P with a "
                    "float32 frame buffer) that the default single-GPU P line carries under `configs`")
    ap.add_argument("--configs", default="Q,R,F,P32", help="which of those summaries to run (comma separated)")
    ap.add_argument("--cpu-frames", type=int, default=-1, help="frames of the same workload timed on the CPU oracle (0 = skip; -1 = all T "
                    "frames if host memory allows, else 64)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU needed: print the per-rank launch plan of `--gpus N` (devices, frame shards, "
                    "the C-ABI call of a step, the collectives it issues with their byte counts) as one JSON object and exit")
    ap.add_argument("--stage-timeout", type=float, default=600.0, help="N>1: rank 0 prints an error line and ends the job when a stage "
                    "before the timed loop (rendezvous, communicator, first step, ...) makes no progress for this many seconds (0 = never)")
    ap.add_argument("--cpu-workers", type=int, default=-1, help="threads of the all-cores CPU figure (0 = skip, -1 = min(64, host cores))")
    a = ap.parse_args()
    T, H, W, L, S, dt = CONFIGS[a.config]
    a.frames = a.frames or T
    a.height = a.height or H
    a.width = a.width or W
    a.levels = a.levels if a.levels is not None else L
    a.skip = a.skip if a.skip is not None else S
    a.in_dtype = a.in_dtype or dt
    return a


def _json_lines(text):
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1).
    First contact with a multi-GPU node must leave a diagnosable record: the ranks' output is captured; when the
    job dies before rank 0 could print its line (a rank killed inside ncclCommInitRank, IPC refused, out of memory) the launcher
    prints ONE JSON line -- n_gpus, "error", the stage every rank had reached, the tails of their stderr -- after ONE retry with the
    collectives of torch.distributed instead of the library's own communicator (RESPMON_BENCH_TORCH_COLLECTIVES=1).  A successful
    retry prints the normal line with the first attempt's diagnosis under "first_attempt"."""
    def attempt(extra_env):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus, "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL between processes on this driver)
        env.setdefault("OMP_NUM_THREADS", "8")
        env.update(extra_env)
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, errors="replace")
        lines = _json_lines(p.stdout)
        good = [j for j in lines if "error" not in j and "value" in j]
        diag = {"returncode": p.returncode, "rank_reports": [j for j in lines if "error" in j][-1:],
                "launcher_stderr_tail": p.stderr[-3000:], "ranks": RankReporter.collect(port, a.gpus)}
        return good[-1] if good else None, diag, p
    line, diag, p = attempt({})
    if line is not None:
        sys.stderr.write(p.stderr)
        print(json.dumps(line))
        return p.returncode
    if os.environ.get("RESPMON_BENCH_TORCH_COLLECTIVES"):
        second = None
    else:
        line, second, p2 = attempt({"RESPMON_BENCH_TORCH_COLLECTIVES": "1"})
        if line is not None:
            line["first_attempt"] = diag
            sys.stderr.write(p2.stderr)
            print(json.dumps(line))
            return p2.returncode
    rep = (diag["rank_reports"] or [{}])[0]
    print(json.dumps({"metric": "Eulerian-calibration frames/sec on 1080p x 256 buffer; achieved HBM GB/s", "value": None, "unit": "frames/s",
                      "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "error": rep.get("error") or "the job ended without a result line",
                      "stage": rep.get("stage") or RankReporter.earliest_stage(diag["ranks"]), "first_attempt": diag,
                      "retry_with_torch_distributed_collectives": second}))
    return p.returncode or 1


class RankReporter:
    """Makes a failing multi-GPU run leave ONE JSON line.  Every rank keeps the stage it has reached -- and, when
    it fails, the tail of its traceback -- in a file of a directory named after the rendezvous port; rank 0 prints the error line:
    from its own exception handler, from the SIGTERM the launcher sends the survivors when another rank has died (a watcher thread
    on the signal wake-up descriptor: the main thread may be blocked inside a collective), or from a watchdog when a stage before
    the timed loop makes no progress for --stage-timeout seconds (a hung ncclCommInitRank)."""
    STAGES = ["start", "rendezvous", "rendezvous_done", "device", "communicator", "frame_buffer", "first_step", "warmup", "timed", "report", "done"]

    @staticmethod
    def directory(port):
        return os.path.join(os.environ.get("TMPDIR", "/tmp"), "respmon_bench_%s" % port)

    @classmethod
    def collect(cls, port, world):
        out = {}
        d = cls.directory(port)
        for r in range(world):
            rec = {"stage": None, "stderr_tail": None}
            try:
                rec.update(json.load(open(os.path.join(d, "rank%d.json" % r))))
            except Exception:   # noqa: BLE001 -- a rank that never got as far as writing its file
                pass
            out[str(r)] = rec
        return out

    @classmethod
    def earliest_stage(cls, ranks):
        idx = [cls.STAGES.index(v["stage"]) for v in ranks.values() if v.get("stage") in cls.STAGES]
        return cls.STAGES[min(idx)] if idx else None

    def __init__(self, a):
        import faulthandler
        import signal
        import threading
        self.a = a
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.port = os.environ.get("MASTER_PORT", str(os.getpid()))
        self.cur = "start"
        self.t_stage = time.time()
        self.printed = False
        self.lock = threading.Lock()
        self.fault = os.environ.get("RESPMON_BENCH_FAULT", "")   # "<rank>:<stage>:kill|raise|hang" -- test hook (tests/test_host_logic.py)
        if self.world > 1:
            os.makedirs(self.directory(self.port), exist_ok=True)
            self.errfile = open(os.path.join(self.directory(self.port), "rank%d.fault" % self.rank), "w")
            faulthandler.enable(self.errfile)       # a segfault inside librccl still leaves its Python stack in the rank's file
            self._write()
            if self.rank == 0:
                r, w = os.pipe()
                os.set_blocking(w, False)
                signal.signal(signal.SIGTERM, lambda *_: None)
                signal.set_wakeup_fd(w, warn_on_full_buffer=False)
                threading.Thread(target=self._watch_signal, args=(r,), daemon=True).start()
                threading.Thread(target=self._watchdog, daemon=True).start()

    def _write(self, err=None):
        if self.world <= 1:
            return
        try:
            with open(os.path.join(self.directory(self.port), "rank%d.json" % self.rank), "w") as f:
                json.dump({"stage": self.cur, "stderr_tail": err, "pid": os.getpid()}, f)
        except OSError:
            pass

    def stage(self, name):
        self.cur = name
        self.t_stage = time.time()
        self._write()
        for spec in filter(None, self.fault.split(",")):
            r, st, kind = (spec.split(":") + ["", ""])[:3]
            if int(r) == self.rank and st == name:
                if kind == "kill":
                    self._write("fault injected: os._exit(17) at stage %s" % name)
                    os._exit(17)
                if kind == "hang":
                    time.sleep(3600)
                raise RuntimeError("fault injected at stage %s" % name)

    def error_line(self, why):
        a = self.a
        ranks = self.collect(self.port, self.world)
        return {"metric": "Eulerian-calibration frames/sec on 1080p x 256 buffer; achieved HBM GB/s", "value": None, "unit": "frames/s",
                "n_gpus": self.world, "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "error": why, "stage": self.cur,
                "earliest_stage_of_any_rank": self.earliest_stage(ranks), "ranks": ranks,
                "hint": "stages: %s; `communicator` = rm_comm_init (ncclCommInitRank) -- RESPMON_BENCH_TORCH_COLLECTIVES=1 keeps the "
                        "collectives of torch.distributed instead" % " > ".join(self.STAGES)}

    def emit(self, why):
        with self.lock:
            if self.printed or self.rank != 0:
                return
            self.printed = True
            sys.stdout.write(json.dumps(self.error_line(why)) + "\n")
            sys.stdout.flush()

    def _watch_signal(self, r):
        import signal
        while True:
            try:
                data = os.read(r, 16)
            except OSError:
                return
            if any(b == signal.SIGTERM for b in data) and self.cur != "done":
                time.sleep(0.2)    # (the rank that died may still be writing its file)
                self.emit("SIGTERM from the launcher while at stage `%s`: another rank has ended" % self.cur)
                os._exit(143)

    def _watchdog(self):
        limit = float(getattr(self.a, "stage_timeout", 0) or 0)
        while limit > 0:
            time.sleep(1.0)
            if self.cur in ("timed", "report", "done"):
                return
            if time.time() - self.t_stage > limit:
                self.emit("no progress for %.0f s at stage `%s`" % (limit, self.cur))
                os._exit(124)

    def failed(self, exc):
        import traceback
        tb = "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))[-3000:]
        self._write(tb)
        if self.world > 1 or not isinstance(exc, SystemExit):
            self.emit("%s: %s" % (type(exc).__name__, exc))


def oracle_single(vid_u8, n_frames, levels, skip, in_dtype):
    """The oracle (a port: the reference's materialising algorithm with the build's C restatement of the cv2 calls)
    timed on this host, single thread like the reference, on the first n_frames frames."""
    from oracle import respmon_oracle as oracle
    oracle.build()
    frames = oracle.uint8_to_float(vid_u8[:n_frames])
    if in_dtype in ("f32", "f16"):   # what the device buffer holds, widened exactly
        frames = frames.astype({"f32": np.float32, "f16": np.float16}[in_dtype]).astype(np.float64)
    t0 = time.perf_counter()
    roi = oracle.locate(frames, 10, pyramid_levels=levels, skip_levels_at_top=skip)
    dt = time.perf_counter() - t0
    return frames, roi, dt


def cpu_baseline(vid_u8, n_frames, levels, skip, in_dtype, workers):
    from oracle import respmon_oracle as oracle
    T = vid_u8.shape[0]
    frames, roi, dt = oracle_single(vid_u8, n_frames, levels, skip, in_dtype)
    out = {"value": n_frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%s %d of the %d frames of the same %dx%d video, oracle.locate (L=%d,S=%d), %.1f s, host has %d cores"
                     % ("the first" if n_frames < T else "all", n_frames, T, vid_u8.shape[1], vid_u8.shape[2], levels, skip, dt, os.cpu_count()),
           "reduced_T": n_frames < T, "roi": roi}
    if workers:
        # the same algorithm with its per-frame / per-pixel-column loops spread over host threads (the C calls release the GIL):
        # bit-identical results, so the GPU is not only compared with a single-threaded strawman
        nw = min(64, os.cpu_count() or 1) if workers < 0 else workers
        t0 = time.perf_counter()
        roi_mt = oracle.locate_parallel(frames, 10, pyramid_levels=levels, skip_levels_at_top=skip, workers=nw)
        dtm = time.perf_counter() - t0
        out["all_cores"] = {"value": n_frames / dtm, "unit": "frames/s", "cores": nw, "seconds": dtm, "roi_equals_single_thread": roi_mt == roi}
    return out


def library_kernel_stamp():
    """The stamp of the frame-buffer kernel sources the LOADED library was built from (include/respmon_hip_debug.h)."""
    from respmon_amd import _capi
    v = _capi.load().rm_debug_kernel_source_stamp()
    return v.decode() if isinstance(v, bytes) else str(v)


def committed_figure(table, key, lib_stamp, field):
    """A committed PMC figure (profiles/hbm_traffic.json, profiles/valu_issue.json) is reported only while the library it was
    measured on is the library that runs: -> (value, stale).  `table[key]` is a dict that carries the figure
    under `field` and the stamp of its library under "kernel_source_sha"; no entry: (None, False); another library: (None, True)."""
    e = (table or {}).get(key)
    if not isinstance(e, dict) or e.get(field) is None:
        return None, False
    if e.get("kernel_source_sha") != lib_stamp:
        return None, True
    return e[field], False


def valu_roofline(dt, T, H, W, kernel_ms):
    """Instruction-issue roofline of the frame-buffer kernel: VALU instructions per input pixel (SQ_INSTS_VALU of a committed PMC
    pass) x ~4.8 cycles per wave64 instruction (bench_micro/valu_rate.hip) over 1024 SIMDs at 2.4 GHz; profiles/valu_issue.json."""
    try:
        v = json.load(open(os.path.join(ROOT, "profiles", "valu_issue.json")))
        key = "%s_%dx%dx%d" % (dt, T, H, W)
        entries = {k: {"ipp": x, "kernel_source_sha": v.get("kernel_source_sha", {}).get(k)} for k, x in v["valu_instructions_per_pixel"].items()}
        ipp, stale = committed_figure(entries, key, library_kernel_stamp(), "ipp")
    except Exception:
        return None
    if ipp is None:
        return {"bound": "valu_issue", "frac": None, "stale": True, "note": "profiles/valu_issue.json was measured on another build of the "
                "frame-buffer kernels (kernel_source_sha differs): re-run tools/pmc_narrow.sh"} if stale else None
    bound_ms = T * H * W * ipp / 64.0 * v["cycles_per_wave_instruction"] / (v["simds"] * v["clock_ghz"] * 1e9) * 1e3
    return {"bound": "valu_issue", "valu_instructions_per_pixel": ipp, "issue_bound_ms": bound_ms,
            "frac": bound_ms / kernel_ms if kernel_ms > 0 else None, "source": "profiles/valu_issue.json (committed PMC figures, not measured in this run)"}


def mode_b_exchange(rdist, need):
    """What respmon_amd.dist.ExchangePolicy does with a stream whose sparse packet needs `need` tiles: first step and steady state."""
    if need is None:
        return "dense all-reduce (no pruning bookkeeping)"
    pol = rdist.ExchangePolicy()
    if need <= pol.cap:
        return "sparse (%d-tile packets)" % pol.cap
    pol.overflowed(need)
    if pol.dense_left:
        return "first step: sparse attempt refused, dense all-reduce; then dense all-reduce for %d steps before the next sparse attempt" % pol.dense_left
    return "first step: sparse attempt refused, dense all-reduce; from the second step on sparse with the cap grown to %d tiles (%.1f MB packets)" % (
        pol.cap, 8e-6 * (4 + pol.cap * 1025))


def dry_run_plan(a):
    """What `bench.py --gpus N` will do, rank by rank, computed with the library's own host-side layout functions (rm_shard_frames,
    rm_shard_layout_flags, rm_heat_sparse_packet_doubles: no GPU needed).  First contact with an 8-GPU node then only has to
    confirm it: the launch line, the per-rank devices and frame shards, the collectives of a step and their sizes."""
    import ctypes
    from respmon_amd import _capi
    lib = _capi.load()
    N, T, H, W, L, S = a.gpus, a.frames, a.height, a.width, a.levels, a.skip
    esz = DT_BYTES[a.in_dtype]
    np_ = ctypes.c_size_t()
    _capi.check(lib, lib.rm_shard_layout_flags(H, W, L, S, 0, ctypes.byref(np_)), "rm_shard_layout_flags")
    NP = int(np_.value)
    cap0, cap_max = 128, 512   # RM_SPARSE_CAP_TILES / RM_SPARSE_MAX_TILES (include/respmon_hip.h)
    pk0, pk_max = int(lib.rm_heat_sparse_packet_doubles(cap0)) * 8, int(lib.rm_heat_sparse_packet_doubles(cap_max)) * 8
    heat_bytes = H * W * 8
    cmax = (T + N - 1) // N
    ranks = []
    for r in range(N):
        t0, t1 = ctypes.c_int(), ctypes.c_int()
        _capi.check(lib, lib.rm_shard_frames(T, r, N, ctypes.byref(t0), ctypes.byref(t1)), "rm_shard_frames")
        if a.mode == "sharded":
            ranks.append({"rank": r, "device": "cuda:%d" % r, "frames": [t0.value, t1.value], "frame_buffer_bytes": (t1.value - t0.value) * H * W * esz,
                          "all_gather_send_bytes": cmax * NP * 8, "padded_frames": cmax - (t1.value - t0.value)})
        else:
            ranks.append({"rank": r, "device": "cuda:%d" % r, "frames": [0, T], "stream_seed": 1234 + r, "frame_buffer_bytes": T * H * W * esz})
    if a.mode == "sharded":
        step = ["rm_locate_sharded (ONE C-ABI call per step, everything on the caller's stream):",
                "  rm_shard_pyramid: frame-buffer kernel over the local frames -> G_S [cmax=%d, NP=%d] float64" % (cmax, NP),
                "  ncclAllGather %d B per rank -> [%d, %d, NP] (%s)" % (cmax * NP * 8, N, cmax, "frames in order" if T % N == 0 else "padded shards, compacted by %d device copies" % N),
                "  rm_shard_collapse: temporal filter + small pyramid + bounds + pruning for all %d frames, evaluation of the local ones" % T,
                "  ncclAllReduce(max) 16 B: {-min, max} of raw",
                "  rm_shard_heat: masked time sum of the local frames -> heat_sum [%d, %d] float64" % (H, W),
                "  sparse exchange: ncclAllGather of %d B packets (cap %d tiles; grows to %d tiles = %d B) -> merge in rank order, / T, ROI on every rank" % (pk0, cap0, cap_max, pk_max),
                "  on a packet overflow: ncclAllReduce(sum) of %d B (dense), then the dense form is held for 64 steps or the cap grows" % heat_bytes]
        scaling = "strong"
    else:
        step = ["rm_locate_streams (ONE C-ABI call per step, everything on the caller's stream):",
                "  rm_calibrate of this rank's own [%d, %d, %d] %s buffer -> heatmap [%d, %d] float64" % (T, H, W, a.in_dtype, H, W),
                "  rm_heat_sparse_pack -> ncclAllGather of %d B packets (cap %d tiles; grows to %d tiles = %d B)" % (pk0, cap0, cap_max, pk_max),
                "  rm_heat_sparse_merge_roi: sum over the ranks in rank order + ROI (every rank the same); ONE host synchronisation",
                "  on a packet overflow: ncclAllReduce(sum) of the %d B heatmaps, second host synchronisation" % heat_bytes]
        scaling = "weak"
    return {"dry_run": True, "n_gpus": N, "mode": a.mode if N > 1 else "single", "scaling": scaling,
            "launch": "python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port <P> bench.py --gpus %d --steps %d --warmup %d%s"
                      % (N, N, a.steps, a.warmup, " --mode sharded" if a.mode == "sharded" else ""),
            "env": {"HSA_ENABLE_IPC_MODE_LEGACY": "0 (dmabuf IPC between the ranks' processes)", "MASTER_ADDR": "127.0.0.1"},
            "communicator": "rank 0: rm_comm_unique_id (128 B) -> torch.distributed broadcast -> every rank rm_comm_init (ncclCommInitRank); all ranks "
                            "agree on success (all_reduce MIN), else every rank keeps torch.distributed collectives",
            "workload": {"frames": T, "height": H, "width": W, "dtype": a.in_dtype, "levels": L, "skip": S, "NP": NP},
            "step": step, "ranks": ranks,
            "value": "frames of ALL ranks per second: %d x steps / max-over-ranks time (barrier + synchronize on both sides)" % ((1 if a.mode == "sharded" else N) * T),
            "xgmi_note": "per step and rank: %.2f MB over xGMI on the sparse path, %.1f MB on the dense one" % (
                (pk0 * (N - 1) + (cmax * NP * 8 * (N - 1) if a.mode == "sharded" else 0)) / 1e6, 2 * heat_bytes * (N - 1) / N / 1e6)}


def main():
    a = parse()
    if a.dry_run:
        print(json.dumps(dry_run_plan(a)))
        return
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))
    rep = RankReporter(a)
    try:
        run(a, rep)
    except BaseException as e:   # noqa: BLE001 -- whatever ends a rank early must leave its trace (and, on rank 0, the JSON line)
        if isinstance(e, SystemExit) and not e.code:
            raise
        rep.failed(e)
        raise


def run(a, rep):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend_name = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend_name = os.environ.get("RESPMON_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        rep.stage("rendezvous")
        if backend_name == "nccl":
            # one rank per GPU; the modulo only matters for the single-GPU dry run of this path, never for the 8-GPU node
            dev_index = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(dev_index)
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend_name)
        rep.stage("rendezvous_done")
        if backend_name != "nccl":
            dist.barrier()
    rep.stage("device")
    assert torch.cuda.is_available(), "bench.py needs the MI355X: torch.cuda.is_available() is False"
    from respmon_amd import _capi, device, synth
    from respmon_amd import dist as rdist
    if world > 1:
        dev_index = local_rank % torch.cuda.device_count()   # (RESPMON_BENCH_BACKEND=gloo: several ranks may share one device)
        torch.cuda.set_device(dev_index)
    else:
        torch.cuda.set_device(0)
    assert a.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (a.gpus, world)
    rep.stage("communicator")
    # N > 1 over RCCL: the library makes its own communicator (include/respmon_hip.h rm_comm_init; torch.distributed only ships the
    # 128-byte id) and a step is ONE C-ABI call (rm_locate_streams / rm_locate_sharded).  All ranks agree on whether that worked;
    # otherwise every rank keeps the torch.distributed collectives of respmon_amd/dist.py.
    comm_info = None
    if world > 1:
        comm_info = {"collectives": "torch.distributed (%s)" % backend_name, "rccl_ranks_seen": None}
        if backend_name == "nccl" and not os.environ.get("RESPMON_BENCH_TORCH_COLLECTIVES"):
            ok, err, seen = 1, None, None
            try:
                _, _, seen = rdist.cabi_comm_init()
            except Exception as e:   # noqa: BLE001 -- any failure means: fall back, together
                ok, err = 0, "%s: %s" % (type(e).__name__, e)
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 1:
                comm_info = {"collectives": "RCCL behind the C-ABI (rm_comm_init; one rm_locate_streams / rm_locate_sharded call per step)",
                             "rccl_ranks_seen": seen}
            else:
                rdist.cabi_comm_destroy()
                comm_info["c_abi_comm_error"] = err or "another rank could not create its communicator"

    T, H, W = a.frames, a.height, a.width
    sharded = world > 1 and a.mode == "sharded"
    big = T * H * W > 1 << 30
    gen = synth.synth_breathing_blocks if big else synth.synth_breathing
    # config 4: an independent stream per GPU; sharded mode: the same buffer everywhere, each rank keeps its frame shard
    if a.video != "breathing":
        gen = {"dense": synth.synth_breathing_dense, "noise": synth.synth_noise_only, "blobs16": synth.synth_breathing_16}[a.video]
        vid_u8 = gen(T, H, W)
    else:
        vid_u8 = gen(T, H, W, seed=1234 + (0 if sharded else rank))
    if sharded:
        t_lo, t_hi = rdist.shard_frames(T, rank, world)
        vid_u8 = vid_u8[t_lo:t_hi]
    TORCH_DT = {"f64": torch.float64, "f32": torch.float32, "f16": torch.float16, "u8": torch.uint8}

    def to_device(v8, dt_name=None):
        dt_name = dt_name or a.in_dtype
        if dt_name == "u8":
            return torch.from_numpy(v8).cuda()
        if dt_name == "bgr8":   # gray(x, x, x) == x: the same video, stored as captured
            b3 = torch.empty(tuple(v8.shape) + (3,), dtype=torch.uint8, device="cuda")
            for t0 in range(0, b3.shape[0], 16):
                b3[t0:t0 + 16] = torch.from_numpy(v8[t0:t0 + 16]).cuda().unsqueeze(-1).expand(-1, -1, -1, 3)
            return b3
        td = TORCH_DT[dt_name]
        b = torch.empty(tuple(v8.shape), dtype=td, device="cuda")
        for t0 in range(0, b.shape[0], 16):  # uint8_to_float in float64 (base.py:231), then the storage dtype
            b[t0:t0 + 16] = (torch.from_numpy(v8[t0:t0 + 16]).cuda().to(torch.float64) * (1.0 / 255)).to(td)
        return b

    rep.stage("frame_buffer")
    buf = to_device(vid_u8)
    torch.cuda.synchronize()

    lib = _capi.load()
    ctx = device.ctx()
    for kv in a.debug_set:
        k, v = kv.split("=", 1)
        device.debug_set(k, int(v))
    flags = _capi.RM_FLAG_NO_PRUNE if a.no_prune else 0
    kw = dict(pyramid_levels=a.levels, skip_levels_at_top=a.skip, flags=flags)

    from respmon_amd.base import _Backend
    backend = _Backend()

    def locate1(b, fl=flags):   # exactly RespiratoryMonitor.locate: one rm_locate call
        return backend.locate(b, 10, 0.1, 1.0, 500, a.levels, a.skip, 0.7, 20, fl)

    def step():
        if world == 1:
            return locate1(buf)
        if sharded:
            return rdist.locate_sharded(buf, T, 10, threshold=20, **kw)
        return rdist.locate_streams(buf, 10, threshold=20, **kw)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = None
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        return r, (time.perf_counter() - t0) / max(n, 1) * 1e3

    rep.stage("first_step")
    roi = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    roi = step()
    torch.cuda.synchronize()
    one = max(time.perf_counter() - t0, 1e-5)
    rep.stage("warmup")
    pre_steps = a.prewarm_steps if a.prewarm_steps is not None else int(min(600, max(10, 0.7 / one)))   # clock ramp of a fresh box
    if world > 1:   # every rank must issue the same number of collectives
        n_t = torch.tensor([pre_steps], dtype=torch.int64, device="cuda")
        dist.broadcast(n_t, 0)
        pre_steps = int(n_t[0])
    for _ in range(max(0, pre_steps)):
        roi = step()
    for _ in range(a.warmup):
        roi = step()
    single_ms = None
    if world > 1 and not sharded:
        # what ONE stream does on this GPU without any exchange, measured in the same run (every rank at once, rank 0 reports): the
        # N = 1 figure the scaling of `value` can be read against
        barrier()
        ts = time.perf_counter()
        n_single = max(3, min(a.steps, 20))
        for _ in range(n_single):
            locate1(buf)
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - ts) / n_single * 1e3
    barrier()
    rep.stage("timed")
    _capi.check(lib, lib.rm_profile_enable(ctx, 1), "rm_profile_enable")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        roi = step()
    barrier()
    elapsed = time.perf_counter() - t0
    rep.stage("report")
    ms = (ctypes.c_double * 4)()
    ncalls = ctypes.c_int()
    _capi.check(lib, lib.rm_profile_read(ctx, ms, ctypes.byref(ncalls)), "rm_profile_read")
    k_ms_total, k_calls = ms[0], ncalls.value     # frame-buffer kernel, HIP events inside the timed region
    # box noise made visible in one run: four more batches of --steps (`value` stays the first batch above)
    batch_ms = [elapsed / a.steps * 1e3]
    if world == 1 and not a.no_batches:
        for _ in range(4):
            barrier()
            tb = time.perf_counter()
            for _ in range(a.steps):
                roi = step()
            barrier()
            batch_ms.append((time.perf_counter() - tb) / a.steps * 1e3)
        _capi.check(lib, lib.rm_profile_read(ctx, (ctypes.c_double * 4)(), None), "rm_profile_read")
    # back-to-back buffers through rm_locate_submit / rm_locate_result: buffer k+1 is submitted before the ROI of buffer k is fetched,
    # so its frame-buffer kernel runs where the synchronous step leaves the GPU idle (host wait + contour stage + launch latency).
    # EXTRA key only -- `value` is the synchronous step above
    pipelined = None
    if world == 1 and not a.no_batches:
        _capi.check(lib, lib.rm_profile_enable(ctx, 0), "rm_profile_enable")
        def pipe(n):
            rois_equal = True
            tk = backend.locate_submit(buf, 10, 0.1, 1.0, 500, a.levels, a.skip, 0.7, 20, flags)
            for _ in range(n - 1):
                nxt = backend.locate_submit(buf, 10, 0.1, 1.0, 500, a.levels, a.skip, 0.7, 20, flags)
                rois_equal &= backend.locate_result(tk) == roi
                tk = nxt
            rois_equal &= backend.locate_result(tk) == roi
            return rois_equal
        pipe(max(2, a.warmup))
        runs, eq = [], True
        for _ in range(2):      # (two batches, both listed: the boxes of this pool drift by several per cent within a run)
            barrier()
            tp = time.perf_counter()
            eq &= pipe(a.steps)
            barrier()
            runs.append((time.perf_counter() - tp) / a.steps * 1e3)
        pipe_ms = min(runs)
        pipelined = {"in_flight": 2, "steps": a.steps, "ms_per_step": pipe_ms, "batches": runs, "frames_per_s": T / pipe_ms * 1e3,
                     "every_roi_equals_the_synchronous_one": bool(eq),
                     "note": "rm_locate_submit(k+1) before rm_locate_result(k), one stream, one context; every step still delivers its ROI to the host"}
        _capi.check(lib, lib.rm_profile_enable(ctx, 1), "rm_profile_enable")
        _capi.check(lib, lib.rm_profile_read(ctx, (ctypes.c_double * 4)(), None), "rm_profile_read")
    tn = ctypes.c_int(0)
    _capi.check(lib, lib.rm_heat_sparse_tiles_needed(ctx, ctypes.byref(tn)), "rm_heat_sparse_tiles_needed")
    tiles_needed = tn.value                       # largest per-rank tile count of the last sparse exchange (world > 1)
    dbg = (ctypes.c_longlong * 4)()
    _capi.check(lib, lib.rm_debug_counters(ctx, dbg, device.stream_ptr()), "rm_debug_counters")
    def sum_path(d):
        if d[3] == -1:
            return "tile-major (rm_tile_eval.h: extrema from the C pairs, every kept pair evaluated where it is summed; no value store)"
        return "dense (every pair recomputed by the sum kernel, no value store)" if d[3] == 0 and d[0] else "sparse (value store)"

    pairs = {"total": dbg[0], "evaluated_for_extrema" if dbg[3] == -1 else "evaluated": dbg[1], "kept_for_sum": dbg[2],
             "store_capacity": max(dbg[3], 0), "sum_path": sum_path(dbg)}
    cn, cl = ctypes.c_int(0), ctypes.c_int(0)
    _capi.check(lib, lib.rm_contour_stats(ctx, ctypes.byref(cn), ctypes.byref(cl)), "rm_contour_stats")
    contour_stage = {"components": cn.value,
                     "path": "device labelling + host border following of the candidates" if cl.value else "host border following"}
    # phase breakdown: separate untimed pass (bracketing every phase costs ~10 us of stream idle time each)
    _capi.check(lib, lib.rm_profile_enable(ctx, 2), "rm_profile_enable")
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _capi.check(lib, lib.rm_profile_read(ctx, ms, ctypes.byref(ncalls)), "rm_profile_read")
    _capi.check(lib, lib.rm_profile_enable(ctx, 0), "rm_profile_enable")
    phases = {"frame_buffer_kernel": ms[0] / max(ncalls.value, 1), "pyramid_rest_and_temporal": ms[1] / max(ncalls.value, 1),
              "collapse_passes": ms[2] / max(ncalls.value, 1), "heatmap_to_roi": ms[3] / max(ncalls.value, 1)}
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])

    extras = world == 1 and not a.no_extras
    n_extra = max(3, min(a.steps, 20))
    # where the host time of a synchronous step goes (include/respmon_hip_debug.h rm_debug_host_timeline): medians over a separate
    # batch of steps -- the read-out itself costs ~1 us per step, so it stays out of the timed region
    host_timeline = None
    if extras:
        _capi.check(lib, lib.rm_profile_enable(ctx, 0), "rm_profile_enable")   # (no event records on the host path being measured)
        marks = np.zeros((max(20, min(a.steps, 100)), 5))
        hm = (ctypes.c_double * 5)()
        for i in range(marks.shape[0]):
            step()
            lib.rm_debug_host_timeline(ctx, hm)
            marks[i] = list(hm)
        m = np.median(marks[3:], axis=0)
        host_timeline = {"between_calls": m[0], "entry_to_first_launch_issued": m[1], "all_launches_issued": m[2], "device_done_seen": m[3],
                         "contour_stage_done": m[4], "host_contour_stage": m[4] - m[3],
                         "note": "microseconds after the entry of rm_locate (between_calls: what the caller spent since the previous return); "
                                 "the GPU idles from the end of its last kernel until the next call's first kernel starts: completion "
                                 "seen by the poll + host_contour_stage + between_calls + entry_to_first_launch_issued + launch latency"}
    # same video kept as uint8 in HBM (what a camera delivers; kernels apply uint8_to_float on the fly):
    # bit-identical ROI at 1/8 of the frame-buffer bytes.  Reported beside the headline, never as `value`.
    alt = None
    if extras and a.in_dtype == "f64" and not a.no_u8_alt:
        buf8 = torch.from_numpy(vid_u8).cuda()
        for _ in range(a.warmup):
            locate1(buf8)
        _capi.check(lib, lib.rm_profile_enable(ctx, 1), "rm_profile_enable")
        roi8, ms8 = timed(lambda: locate1(buf8), max(n_extra, 100))   # (0.3 ms steps: a batch of 20 is shorter than the clocks take to settle)
        _capi.check(lib, lib.rm_profile_read(ctx, ms, ctypes.byref(ncalls)), "rm_profile_read")
        _capi.check(lib, lib.rm_profile_enable(ctx, 0), "rm_profile_enable")
        k8 = ms[0] / max(ncalls.value, 1)
        alt = {"frame_buffer_dtype": "u8", "value": T / ms8 * 1e3, "unit": "frames/s", "ms_per_step": ms8, "kernel_ms": k8,
               "roi": roi8, "roi_equals_headline": list(roi8 or []) == list(roi or []),
               # this kernel is bound by instruction issue, not by bytes: its second roofline
               "valu_roofline": valu_roofline("u8", T, H, W, k8)}
        del buf8
        # ... and as captured: [T,H,W,3] uint8 BGR (north_star's [T,H,W,C] buffer; RM_BGR8).  The frame-buffer kernel applies base.py:230's
        # cvtColor while it unpacks a row (3 bytes per pixel, 5.5 more integer instructions).  Three equal planes: gray(x, x, x) == x, so
        # the ROI must equal the headline's; the arithmetic does not depend on the data.
        if not a.no_bgr_alt:
            buf3 = torch.from_numpy(vid_u8).cuda().unsqueeze(-1).expand(-1, -1, -1, 3).contiguous()
            for _ in range(a.warmup):
                locate1(buf3)
            _capi.check(lib, lib.rm_profile_enable(ctx, 1), "rm_profile_enable")
            roi3, ms3 = timed(lambda: locate1(buf3), max(n_extra, 100))
            _capi.check(lib, lib.rm_profile_read(ctx, ms, ctypes.byref(ncalls)), "rm_profile_read")
            _capi.check(lib, lib.rm_profile_enable(ctx, 0), "rm_profile_enable")
            k3 = ms[0] / max(ncalls.value, 1)
            b3 = T * H * W * 3 + H * W * 8
            alt["bgr8"] = {"frame_buffer": "[T,H,W,3] uint8 (RM_BGR8)", "value": T / ms3 * 1e3, "unit": "frames/s", "ms_per_step": ms3, "kernel_ms": k3,
                           "algorithmic_bytes": b3, "frac": b3 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "kernel_frac": b3 / (k3 * 1e-3) / 1e9 / HBM_PEAK_GBS if k3 > 0 else None,
                           "roi": roi3, "roi_equals_headline": list(roi3 or []) == list(roi or [])}
            del buf3
    # BASELINE config 4 is "calibration + ROI flow": the per-frame motion extraction (base.py:354-407, 'flow' method) on the
    # ROI just found -- Shi-Tomasi corners once, then pyramidal LK + mean flow + PCA per frame.  Latency bound
    # (no roofline fraction is meaningful); reported beside the headline, never part of `value`.
    roi_flow = None
    if extras and roi is not None and not a.no_roi_flow and a.config == "P":
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
            # the same frames through rm_flow_begin / rm_flow_step: crop + LK + mean flow in ONE call per frame, crops and points
            # resident on the device (what RespiratoryMonitor.extract_motion('flow') runs); same numbers as the four calls above
            grays = [torch.from_numpy(vid_u8[i]).cuda() for i in range(n_fl + 1)]
            fstate = backend.flow_state()
            backend.flow_begin(fstate, grays[0], x, y, w, h, 100, 0.3, 7, 7)
            motion2 = []
            torch.cuda.synchronize()
            tf = time.perf_counter()
            for i in range(n_fl):
                mean, n_good = backend.flow_step(fstate, grays[i + 1], x, y, w, h, (15, 15), 2, (3, 10, 0.03))
                if n_good:
                    motion2.append([mean[0], mean[1]])
                if len(motion2) >= 2:
                    backend.pca_reduce(np.array(motion2, dtype=np.float32))
            torch.cuda.synchronize()
            dtr = (time.perf_counter() - tf) / max(n_fl, 1)
            roi_flow = {"roi": [x, y, w, h], "corners": int(len(pts)), "frames": n_fl, "ms_per_frame": dtr * 1e3,
                        "frames_per_s": 1.0 / dtr if dtr > 0 else None, "budget_ms_at_30fps": 33.3,
                        "path": "rm_flow_step: crop + pyramidal LK + mean flow per call, state resident on the device; + rm_pca_reduce",
                        "four_calls_ms_per_frame": dtf * 1e3,
                        "equal_to_four_calls": bool(np.array_equal(np.array(motion2, dtype=np.float32), np.array(motion, dtype=np.float32)))}
        else:
            roi_flow = {"roi": [x, y, w, h], "corners": 0}
    # Data dependence of the headline: the collapse passes skip every (frame, tile) pair that provably cannot
    # hold the extrema or a value below `top`.  (1) the same buffer with pruning off (every pair evaluated, bit-identical
    # heatmap); (2) a "dense" synthetic stream -- four breathing blobs and 3x the noise -- with its pair counts, its step
    # time and the tiles a Mode-B sparse packet would need against the 128-tile cap.
    no_prune = dense = None
    if extras and not a.no_data_dependence and not a.no_prune:
        for _ in range(2):
            locate1(buf, _capi.RM_FLAG_NO_PRUNE)
        roi_np, ms_np = timed(lambda: locate1(buf, _capi.RM_FLAG_NO_PRUNE), max(3, n_extra // 4))
        no_prune = {"ms_per_step": ms_np, "frames_per_s": T / ms_np * 1e3, "roi_equals_headline": list(roi_np or []) == list(roi or [])}
        del buf
        torch.cuda.empty_cache()
        dgen = synth.synth_breathing_dense
        dbuf = to_device(dgen(T, H, W, seed=4321, workers=None if big else 8))
        for _ in range(a.warmup + 20):
            locate1(dbuf)
        roi_d, ms_d = timed(lambda: locate1(dbuf), n_extra)
        _capi.check(lib, lib.rm_debug_counters(ctx, dbg, device.stream_ptr()), "rm_debug_counters")
        heat_d = rdist.hip_calibrate(dbuf, 10, pyramid_levels=a.levels, skip_levels_at_top=a.skip)
        need = rdist.hip_sparse_tiles(heat_d)
        dense = {"video": "four blobs (A=0.2, 0.4 Hz, phases 0/90/180/270 deg) + noise sigma 0.06 (3x), seed 4321",
                 "ms_per_step": ms_d, "frames_per_s": T / ms_d * 1e3, "roi": roi_d,
                 "collapse_pairs": {"total": dbg[0], "evaluated_for_extrema" if dbg[3] == -1 else "evaluated": dbg[1], "kept_for_sum": dbg[2],
                                    "store_capacity": max(dbg[3], 0), "sum_path": sum_path(dbg)},
                 "mode_b_sparse_tiles_needed": need, "mode_b_sparse_tile_cap": rdist.SPARSE_CAP_TILES,
                 "mode_b_exchange": mode_b_exchange(rdist, need)}
        del dbuf, heat_d
        torch.cuda.empty_cache()

    # Worst case under DEFAULT flags: streams on which the pruning of the collapse passes finds little to prune --
    # sensor noise of sigma 0.1 in every pixel and no breathing region; sixteen small blobs a sixteenth of a period apart -- with
    # their pair counts, against the headline's step time, and the oracle's ROI on the same video.
    worst = None
    if extras and not a.no_data_dependence and not a.no_worst_case and not a.no_prune and a.config == "P":
        from oracle import respmon_oracle as oracle
        oracle.build()
        worst = {"headline_ms_per_step": elapsed / a.steps * 1e3}
        for name, gen_w, desc in (("noise", synth.synth_noise_only, "static texture + noise sigma 0.1 in every pixel, no breathing region, seed 777"),
                                  ("blobs16", synth.synth_breathing_16, "sixteen blobs (A=0.2, 0.4 Hz, sigma 0.05H x 0.04W, phases 22.5 deg apart) + noise sigma 0.04, seed 888")):
            try:
                del buf
            except NameError:
                pass
            torch.cuda.empty_cache()
            v8w = gen_w(T, H, W, workers=None if big else 8)
            wbuf = to_device(v8w)
            for _ in range(a.warmup + 5):
                locate1(wbuf)
            roi_w, ms_w = timed(lambda: locate1(wbuf), n_extra)
            _capi.check(lib, lib.rm_debug_counters(ctx, dbg, device.stream_ptr()), "rm_debug_counters")
            del wbuf
            torch.cuda.empty_cache()
            fr = oracle.uint8_to_float(v8w)
            if a.in_dtype in ("f32", "f16"):
                fr = fr.astype({"f32": np.float32, "f16": np.float16}[a.in_dtype]).astype(np.float64)
            r_or = oracle.locate_parallel(fr, 10, pyramid_levels=a.levels, skip_levels_at_top=a.skip, workers=min(64, os.cpu_count() or 1))
            del fr, v8w
            worst[name] = {"video": desc, "ms_per_step": ms_w, "frames_per_s": T / ms_w * 1e3, "vs_headline": ms_w / worst["headline_ms_per_step"],
                           "roi": roi_w, "oracle_roi": r_or, "roi_equals_oracle": list(r_or or []) == list(roi_w or []),
                           "collapse_pairs": {"total": dbg[0], "evaluated_for_extrema" if dbg[3] == -1 else "evaluated": dbg[1],
                                              "kept_for_sum": dbg[2], "store_capacity": max(dbg[3], 0), "sum_path": sum_path(dbg)}}
        worst["slowest_vs_headline"] = max(worst[k]["vs_headline"] for k in ("noise", "blobs16"))

    # The other BASELINE configs as driver-verifiable summaries inside the default P line (config 2 = Q, config 5 = R,
    # config 3 = F; P32 = the headline workload with the float32 frame buffer BASELINE.md quotes its time on).  Each one is
    # measured like the headline (resident buffer, HIP events around the frame-buffer kernel inside the timed loop) and its ROI
    # is compared with the CPU oracle run on ALL frames of the same video.
    other = None
    if extras and a.config == "P" and not a.no_configs and not a.no_prune and not a.debug_set:
        other = {}
        try:
            del buf
        except NameError:
            pass
        torch.cuda.empty_cache()
        which = [c for c in a.configs.split(",") if c]

        def calib_summary(cT, cH, cW, cL, cS, cdt, n_steps, n_warm, check):
            cbig = cT * cH * cW > 1 << 30
            v8 = (synth.synth_breathing_blocks if cbig else synth.synth_breathing)(cT, cH, cW, seed=1234)
            cbuf = to_device(v8, cdt)
            fn = lambda: backend.locate(cbuf, 10, 0.1, 1.0, 500, cL, cS, 0.7, 20, 0)
            for _ in range(n_warm):
                c_roi = fn()
            torch.cuda.synchronize()
            _capi.check(lib, lib.rm_profile_enable(ctx, 1), "rm_profile_enable")
            c_roi, c_ms = timed(fn, n_steps)
            _capi.check(lib, lib.rm_profile_read(ctx, ms, ctypes.byref(ncalls)), "rm_profile_read")
            _capi.check(lib, lib.rm_profile_enable(ctx, 0), "rm_profile_enable")
            ck_ms = ms[0] / max(ncalls.value, 1)
            _capi.check(lib, lib.rm_debug_counters(ctx, dbg, device.stream_ptr()), "rm_debug_counters")
            _capi.check(lib, lib.rm_contour_stats(ctx, ctypes.byref(cn), ctypes.byref(cl)), "rm_contour_stats")
            cb = cT * cH * cW * DT_BYTES[cdt] + cH * cW * 8
            d = {"workload": "%dx%dx%d %s frame buffer, %d-level pyramid, skip %d" % (cT, cH, cW, cdt, cL, cS),
                 "steps": n_steps, "ms_per_step": c_ms, "frames_per_s": cT / c_ms * 1e3, "kernel_ms": ck_ms,
                 "algorithmic_bytes": cb, "frac": cb / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,      # whole step, as `roofline.frac`
                 "kernel_frac": cb / (ck_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ck_ms > 0 else None,
                 "step_frac": cb / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "roi": c_roi,
                 "collapse_pairs": {"total": dbg[0], "evaluated_for_extrema" if dbg[3] == -1 else "evaluated": dbg[1], "kept_for_sum": dbg[2],
                                    "sum_path": sum_path(dbg)},
                 "contour_components": cn.value, "contour_labelled": bool(cl.value)}
            d["valu_roofline"] = valu_roofline(cdt, cT, cH, cW, ck_ms)
            try:    # HBM bytes per launch of the frame-buffer kernel from the committed PMC passes of this configuration, if any
                d["traffic"] = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get(
                    "%s_%dx%dx%d" % (cdt, cT, cH, cW), {}).get("bytes_per_launch")
            except Exception:
                d["traffic"] = None
            del cbuf
            torch.cuda.empty_cache()
            if check:
                import psutil
                from oracle import respmon_oracle as oracle
                oracle.build()
                if psutil.virtual_memory().available > 1.3 * 6.0 * cT * cH * cW * 8:
                    fr = oracle.uint8_to_float(v8)
                    if cdt in ("f32", "f16"):
                        fr = fr.astype({"f32": np.float32, "f16": np.float16}[cdt]).astype(np.float64)
                    t_or = time.perf_counter()
                    r_or = oracle.locate_parallel(fr, 10, pyramid_levels=cL, skip_levels_at_top=cS, workers=min(64, os.cpu_count() or 1))
                    d["oracle_roi"] = r_or
                    d["oracle_seconds"] = time.perf_counter() - t_or
                    d["roi_equals_oracle"] = list(r_or or []) == list(c_roi or [])
                    del fr
                else:
                    d["roi_equals_oracle"] = None
                    d["oracle_skipped"] = "host memory"
            return d

        if "Q" in which:
            other["Q"] = calib_summary(*CONFIGS["Q"], 100, 10, True)
        if "P32" in which:
            other["P32"] = calib_summary(T, H, W, a.levels, a.skip, "f32", 50, 10, False)
            other["P32"]["roi_equals_headline"] = list(other["P32"]["roi"] or []) == list(roi or [])
        if "R" in which:
            other["R"] = calib_summary(*CONFIGS["R"], 10, 4, True)
        if "F" in which:
            # config 3: pyramidal LK on a 256x256 ROI, 1000 Shi-Tomasi points, one rm_flow_step per frame (30 fps stream)
            from oracle import respmon_oracle as oracle
            oracle.build()
            render = synth.synth_texture(256, 256, seed=4321)
            n_fr = 60
            fr8 = [render(1.5 * np.sin(2 * np.pi * 0.4 * t / 30), 0.5 * np.sin(2 * np.pi * 0.4 * t / 30 + np.pi / 3)) for t in range(n_fr + 1)]
            dev = [torch.from_numpy(f).cuda() for f in fr8]
            fstate3 = backend.flow_state()
            pts0 = backend.flow_begin(fstate3, dev[0], 0, 0, 256, 256, 1000, 0.01, 3, 7)
            n0 = 0 if pts0 is None else len(pts0)
            means = []
            torch.cuda.synchronize()
            tf = time.perf_counter()
            for i in range(n_fr):
                mean, n_good = backend.flow_step(fstate3, dev[i + 1], 0, 0, 256, 256, (15, 15), 2, (3, 10, 0.03))
                means.append((float(mean[0]), float(mean[1]), int(n_good)))
            torch.cuda.synchronize()
            dtf = (time.perf_counter() - tf) / n_fr
            # the oracle on the first 5 frames: corners, then LK + mean flow frame after frame
            # (the monitor's frames are float: base.py:231 uint8_to_float, then base.py:364 float_to_uint8 of the crop, which loses
            #  24 of the 256 levels -- the oracle sees the same round trip)
            fr8 = [oracle.float_to_uint8(oracle.uint8_to_float(f)) for f in fr8[:6]]
            ref = oracle.goodFeaturesToTrack(fr8[0], 1000, 0.01, 3, blockSize=7)
            ok = ref is not None and pts0 is not None and np.array_equal(np.asarray(pts0).reshape(-1, 2), ref.reshape(-1, 2))
            p = ref
            for i in range(5):
                if not ok:
                    break
                r1, rs, _ = oracle.calcOpticalFlowPyrLK(fr8[i], fr8[i + 1], p, None, winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))
                good = rs.ravel() == 1
                m = np.mean(p[rs == 1] - r1[rs == 1], axis=0) if good.any() else np.zeros(2, np.float32)
                ok = ok and means[i][2] == int(good.sum()) and np.array_equal(np.array(means[i][:2], np.float32), m.astype(np.float32))
                p = r1[rs == 1].reshape(-1, 1, 2)
            other["F"] = {"workload": "pyramidal LK on a 256x256 ROI, maxCorners=1000 (qualityLevel 0.01, minDistance 3), winSize 15, maxLevel 2; "
                                      "one rm_flow_step per frame", "corners": n0, "frames": n_fr, "ms_per_frame": dtf * 1e3,
                          "frames_per_s": 1.0 / dtf, "budget_ms_at_30fps": 33.3, "points_at_end": means[-1][2],
                          "first_5_frames_equal_oracle": bool(ok)}

    if rank == 0:
        frames_total = (1 if sharded else world) * T * a.steps
        t_local = int(vid_u8.shape[0])
        # one read of the (rank-local) frame buffer + the heatmap
        b_alg = t_local * H * W * DT_BYTES[a.in_dtype] + H * W * 8
        k_ms = k_ms_total / max(k_calls, 1)
        step_ms = elapsed / a.steps * 1e3
        achieved = b_alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None
        step_achieved = b_alg / (step_ms * 1e-3) / 1e9
        traffic, traffic_stale = None, False
        lib_stamp = library_kernel_stamp()
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic, traffic_stale = committed_figure(json.load(open(tpath)), "%s_%dx%dx%d" % (a.in_dtype, t_local, H, W), lib_stamp, "bytes_per_launch")
            except Exception:
                traffic = None
        metric = "Eulerian-calibration frames/sec on 1080p x 256 buffer; achieved HBM GB/s" if (T, H, W) == (256, 1080, 1920) else \
                 "Eulerian-calibration frames/sec on %dx%d x %d buffer; achieved HBM GB/s" % (W, H, T)
        out = {
            "metric": metric,
            "value": frames_total / elapsed, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": step_ms, "ms_per_step_batches": {"batches": batch_ms, "min": min(batch_ms), "median": float(np.median(batch_ms)),
                                                            "note": "batch 0 is the timed region `value` comes from; the others follow it"},
            "pipelined": pipelined,
            "pre_warm_steps": pre_steps, "higher_is_better": True, "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "f64", "frame_buffer_dtype": a.in_dtype, "data": "synthetic",
            "config": {"workload": "%s: Eulerian calibration + ROI (locate) on a %dx%dx%d %s frame buffer, %d-level Laplacian pyramid, "
                                   "skip %d, temporal FFT band-pass 0.1-1.0 Hz @10 fps; %s" % (
                                       a.config, T, H, W, a.in_dtype, a.levels, a.skip,
                                       "ONE buffer sharded by frame index over the GPUs: all-gather of the small pyramid, min/max all-reduce, "
                                       "one RCCL heatmap exchange" if sharded else "one independent stream per GPU + one RCCL exchange of the heatmaps"),
                       "preset": a.config, "frame_buffer_dtype": a.in_dtype, "frames": T, "height": H, "width": W, "levels": a.levels,
                       "skip": a.skip, "prune": not a.no_prune, "mode": a.mode if world > 1 else "single"},
            "video": a.video, "comm": comm_info,
            "single_stream_same_run": ({"ms_per_step": single_ms, "frames_per_s": T / single_ms * 1e3,
                                        "note": "rm_locate on this rank's own buffer, no exchange, all ranks at once: the N = 1 figure of this run"}
                                       if single_ms else None),
            "world": world, "backend": {"nccl": "nccl (RCCL)"}.get(backend_name, backend_name),
            "debug_set": a.debug_set,
            # `achieved` / `frac` are the contract figure -- algorithmic bytes over the WHOLE step (all kernels + the host
            # contour stage) against the peak; the frame-buffer kernel alone is `kernel_achieved` / `kernel_frac`
            "roofline": {"bound": "hbm", "achieved": step_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": step_achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_stale": traffic_stale,
                         "kernel_source_sha": lib_stamp,
                         "kernel_achieved": achieved, "kernel_frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "traffic_source": ("profiles/hbm_traffic.json (rocprofv3 --pmc passes of this command on a library with the same "
                                            "kernel_source_sha, committed; not measured in this run)" if traffic else
                                            ("profiles/hbm_traffic.json holds a figure of ANOTHER build of the frame-buffer kernels: not "
                                             "reported (tools/profile_round.sh refreshes it)" if traffic_stale else None)),
                         "kernel": "frame-buffer pyrDown kernel (reads [T,H,W] once)", "kernel_ms": k_ms,
                         "algorithmic_bytes": b_alg,
                         # the contract figure of SURVEY 8(d): the WHOLE step (all kernels + host contour stage) against the peak
                         "step_achieved": step_achieved, "step_frac": step_achieved / HBM_PEAK_GBS,
                         "valu_roofline": valu_roofline(a.in_dtype, t_local, H, W, k_ms)},
            "phases_ms_per_step": phases,
            "roi": roi,
            "heatmap_exchange": (rdist.LAST_EXCHANGE and {"sparse": "one all-gather of sparse packets (%d-tile cap, %.2f MB per rank)"
                                                          % (rdist.SPARSE_CAP_TILES, 8e-6 * (4 + rdist.SPARSE_CAP_TILES * 1025)),
                                                          "dense": "all-reduce(sum) of the [H,W] float64 heatmap"}[rdist.LAST_EXCHANGE])
                                if world > 1 else None,
            "heatmap_exchange_tiles_needed": tiles_needed if world > 1 else None,
            "alt_uint8_buffer": alt,
            "roi_flow": roi_flow,
            "collapse_pairs": pairs,
            "contour_stage": contour_stage,
            "no_prune": no_prune,
            "dense_stream": dense,
            "host_timeline_us": host_timeline,
            "worst_case": worst,
            "configs": other,
        }
        if world == 1 and a.cpu_frames != 0:
            n_cpu = a.cpu_frames
            if n_cpu < 0:
                import psutil
                need = 6.0 * T * H * W * 8   # the materialising algorithm holds ~5-6 float64 [T,H,W] arrays
                n_cpu = T if psutil.virtual_memory().available > 1.3 * need else 64
            n_cpu = min(n_cpu, T)
            out["cpu_baseline"] = cpu_baseline(vid_u8, n_cpu, a.levels, a.skip, a.in_dtype, a.cpu_workers)
            if n_cpu == T:
                out["cpu_baseline"]["roi_equals_gpu"] = list(out["cpu_baseline"]["roi"] or []) == list(roi or [])
        else:
            out["cpu_baseline"] = None
        with rep.lock:
            rep.printed = True
            print(json.dumps(out))
            sys.stdout.flush()
    rep.stage("done")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
