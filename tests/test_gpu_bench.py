"""bench.py on the GPU box: the single-GPU contract line, and `python bench.py --gpus 2` launching its own ranks (no
torchrun in the command) -- two ranks on the one GPU over gloo (RESPMON_BENCH_BACKEND=gloo stages device tensors
through the host; on the 8-GPU node the same code runs over RCCL), both multi-GPU modes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--frames", "64", "--height", "270", "--width", "480", "--prewarm-steps", "3", "--cpu-frames", "0"]


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    return out


def _json_line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-1500:], out.stderr[-3000:])
    return json.loads(lines[0])


def test_bench_single_gpu_contract_line():
    j = _json_line(_run(["--steps", "5", "--warmup", "2"] + SMALL + ["--cpu-frames", "16", "--cpu-workers", "4"]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 5 and j["warmup"] == 2 and j["dtype"] == "f64" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["step_frac"] == r["frac"]                                # `frac` is the contract figure: the whole step
    assert 0 < r["frac"] <= r["kernel_frac"] < 1.0                    # the whole step can only be slower than its main kernel
    assert r["algorithmic_bytes"] == 64 * 270 * 480 * 8 + 270 * 480 * 8
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["all_cores"]["cores"] == 4 and c["all_cores"]["roi_equals_single_thread"]
    assert j["no_prune"]["roi_equals_headline"] and j["dense_stream"]["collapse_pairs"]["total"] > 0
    p = j["pipelined"]      # rm_locate_submit / rm_locate_result: an EXTRA key, `value` is the synchronous step
    assert p["in_flight"] == 2 and p["every_roi_equals_the_synchronous_one"] and len(p["batches"]) == 2 and p["ms_per_step"] == min(p["batches"])
    assert len(j["ms_per_step_batches"]["batches"]) == 5 and j["ms_per_step_batches"]["batches"][0] == j["ms_per_step"]
    w = j["worst_case"]
    assert w["noise"]["roi_equals_oracle"] and w["blobs16"]["roi_equals_oracle"] and w["slowest_vs_headline"] >= 1.0


@pytest.mark.parametrize("mode", ["streams", "sharded"])
def test_bench_launches_its_own_ranks(mode):
    j = _json_line(_run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--mode", mode] + SMALL, {"RESPMON_BENCH_BACKEND": "gloo"}))
    assert j["n_gpus"] == 2 and j["world"] == 2 and j["backend"] == "gloo" and j["steps"] == 3
    assert j["scaling"] == ("weak" if mode == "streams" else "strong")
    assert j["heatmap_exchange"] is not None and j["roi"] is not None
    assert j["config"]["mode"] == mode
