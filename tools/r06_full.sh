#!/bin/bash
# round 6: the whole GPU suite, smoke(), and the default bench line (what the driver runs at round end)
OUT=gpurun_out/r06/full
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -6 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
timeout 1200 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; python - <<PY
import json
j = json.load(open("$OUT/bench_n1.json"))
print("ms_per_step", j["ms_per_step"], "frac", j["roofline"]["frac"], "kernel_ms", j["roofline"]["kernel_ms"], "traffic", j["roofline"]["traffic"], j["roofline"].get("traffic_stale"))
print("batches", j["ms_per_step_batches"]["batches"])
for k, v in (j.get("configs") or {}).items():
    print(k, {a: v.get(a) for a in ("ms_per_step", "frac", "roi_equals_oracle", "ms_per_frame") if a in v})
print("worst", {k: (v.get("ms_per_step"), v.get("vs_headline")) for k, v in (j.get("worst_case") or {}).items() if isinstance(v, dict)})
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"].get("roi_equals_gpu"))
PY
