#!/bin/bash
# round-4 evidence for the other presets + the synchronous-step timeline of the headline: bash tools/r04_final_cfgs.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04/sync
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/sync -o k -- python bench.py --steps 100 --warmup 5 --cpu-frames 0 --no-extras --no-batches > gpurun_out/r04/sync/bench.json 2> gpurun_out/r04/sync/err.txt < /dev/null
python tools/gpu_timeline.py gpurun_out/r04/sync > gpurun_out/r04/gpu_timeline.txt 2>&1
rm -f gpurun_out/r04/sync/*kernel_trace.csv gpurun_out/r04/sync/*agent_info.csv
cat gpurun_out/r04/gpu_timeline.txt
for c in Q P32 P8 R; do
  bash tools/profile_round.sh r04 $c > gpurun_out/r04/profile_$c.log 2>&1
  python tools/gpu_timeline.py gpurun_out/r04/$c/stats > gpurun_out/r04/gpu_timeline_config_$c.txt 2>&1
  rm -f gpurun_out/r04/$c/stats/*kernel_trace.csv
  python - <<PY
import json
d = json.load(open("gpurun_out/r04/$c/bench_n1.json"))
r = d["roofline"]
print("$c", "ms", round(d["ms_per_step"], 4), "frac", round(r["frac"], 4), "kernel_ms", round(r["kernel_ms"], 4), "roi", d["roi"], "pipelined", d.get("pipelined", {}) and round(d["pipelined"]["ms_per_step"], 4))
PY
done
