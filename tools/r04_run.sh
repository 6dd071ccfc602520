#!/bin/bash
# round-4 measurement batch on the GPU box: bash tools/r04_run.sh <tag> [bench args]
# bench line (headline only) + rocprofv3 kernel trace -> per-step timeline
TAG=${1:-base}; shift
OUT=gpurun_out/r04/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py --steps 200 --warmup 10 --cpu-frames 0 --no-extras "$@" > $OUT/bench.json 2> $OUT/bench.err < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python bench.py --steps 100 --warmup 10 --cpu-frames 0 --no-extras --no-batches "$@" > $OUT/bench_rocprof.json 2> $OUT/rocprof.err < /dev/null
python tools/gpu_timeline.py $OUT/stats > $OUT/timeline.txt 2>&1
rm -f $OUT/stats/*kernel_trace.csv $OUT/stats/*agent_info.csv
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
r = d["roofline"]
print("$TAG ms_per_step %.4f kernel_ms %.4f step_frac %.4f roi %s" % (d["ms_per_step"], r["kernel_ms"], r["step_frac"], d["roi"]))
PY
cat $OUT/timeline.txt
