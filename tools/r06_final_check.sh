#!/bin/bash
# round 6: final library -- the whole GPU suite, smoke(), the default bench line, the worst-case timelines again (the refinement of skip 3 / 4 bounds came after the evidence batch)
OUT=gpurun_out/r06/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests/ -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
timeout 1200 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python - <<PY
import json
j = json.load(open("$OUT/bench_n1.json"))
r = j["roofline"]
print("P ms", j["ms_per_step"], "frac", r["frac"], "kernel_ms", r["kernel_ms"], "traffic", r["traffic"], r.get("traffic_stale"), "valu", (r.get("valu_roofline") or {}).get("frac"))
for k, v in (j.get("configs") or {}).items():
    print(k, {a: v.get(a) for a in ("ms_per_step", "frac", "roi_equals_oracle", "ms_per_frame", "traffic") if a in v})
print("worst", {k: (v.get("ms_per_step"), v.get("vs_headline")) for k, v in (j.get("worst_case") or {}).items() if isinstance(v, dict)}, j["worst_case"].get("slowest_vs_headline"))
print("alt", j["alt_uint8_buffer"].get("ms_per_step"), "cpu", j["cpu_baseline"]["value"], j["cpu_baseline"].get("roi_equals_gpu"))
PY
for v in noise blobs16; do
  mkdir -p $OUT/wc_$v
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wc_$v -o k -- python bench.py --steps 60 --warmup 5 --prewarm-steps 30 --video $v --cpu-frames 0 --no-extras --no-batches > $OUT/wc_$v/bench.json 2> $OUT/wc_$v/err.txt < /dev/null
  python tools/gpu_timeline.py $OUT/wc_$v > $OUT/gpu_timeline_worst_case_$v.txt 2>&1
  rm -f $OUT/wc_$v/*kernel_trace.csv $OUT/wc_$v/*agent_info.csv
  head -14 $OUT/gpu_timeline_worst_case_$v.txt | cut -c1-90
done
