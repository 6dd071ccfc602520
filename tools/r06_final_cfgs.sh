#!/bin/bash
# round-6 evidence: headline (bench line, rocprof stats, PMC passes), synchronous-step timeline, the other presets, worst-case timelines
#   bash tools/r06_final_cfgs.sh      (on the GPU box, from the repo root; ~12 min)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r06/sync
bash tools/profile_round.sh r06 P > gpurun_out/r06/profile_P.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06/sync -o k -- python bench.py --steps 100 --warmup 5 --cpu-frames 0 --no-extras --no-batches > gpurun_out/r06/sync/bench.json 2> gpurun_out/r06/sync/err.txt < /dev/null
python tools/gpu_timeline.py gpurun_out/r06/sync > gpurun_out/r06/gpu_timeline.txt 2>&1
rm -f gpurun_out/r06/sync/*kernel_trace.csv gpurun_out/r06/sync/*agent_info.csv
cat gpurun_out/r06/gpu_timeline.txt
python tools/host_timeline.py 300 > gpurun_out/r06/host_timeline.txt 2>&1; tail -2 gpurun_out/r06/host_timeline.txt
for v in noise blobs16; do
  mkdir -p gpurun_out/r06/wc_$v
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06/wc_$v -o k -- python bench.py --steps 60 --warmup 5 --prewarm-steps 30 --video $v --cpu-frames 0 --no-extras --no-batches > gpurun_out/r06/wc_$v/bench.json 2> gpurun_out/r06/wc_$v/err.txt < /dev/null
  python tools/gpu_timeline.py gpurun_out/r06/wc_$v > gpurun_out/r06/gpu_timeline_worst_case_$v.txt 2>&1
  rm -f gpurun_out/r06/wc_$v/*kernel_trace.csv gpurun_out/r06/wc_$v/*agent_info.csv
  head -1 gpurun_out/r06/gpu_timeline_worst_case_$v.txt
done
for c in Q P32 P8 PBGR R; do
  bash tools/profile_round.sh r06 $c > gpurun_out/r06/profile_$c.log 2>&1
  python tools/gpu_timeline.py gpurun_out/r06/$c/stats > gpurun_out/r06/gpu_timeline_config_$c.txt 2>&1
  rm -f gpurun_out/r06/$c/stats/*kernel_trace.csv
  python - <<PY
import json
d = json.load(open("gpurun_out/r06/$c/bench_n1.json"))
r = d["roofline"]
print("$c", "ms", round(d["ms_per_step"], 4), "frac", round(r["frac"], 4), "kernel_ms", round(r["kernel_ms"], 4), "roi", d["roi"])
PY
done
rm -f gpurun_out/r06/stats/*kernel_trace.csv
python - <<PY
import json
d = json.load(open("gpurun_out/r06/bench_n1.json"))
r = d["roofline"]
print("P", "ms", round(d["ms_per_step"], 4), "frac", round(r["frac"], 4), "kernel_ms", round(r["kernel_ms"], 4), "roi", d["roi"], "worst", d.get("worst_case", {}).get("slowest_vs_headline"))
PY
