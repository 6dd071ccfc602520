#!/bin/bash
# developer: synchronous-step GPU timelines (kernel durations, gaps, idle between steps) of the given bench.py configurations
#   bash tools/r05_timelines.sh <outdir-tag> "Q:--config Q --steps 80" "R:--config R --steps 12" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
for spec in "$@"; do
  name=${spec%%:*}; args=${spec#*:}
  OUT=gpurun_out/$TAG/tl_$name; mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python bench.py $args --warmup 5 --cpu-frames 0 --no-extras --no-batches > $OUT/bench.json 2> $OUT/err.txt < /dev/null
  python tools/gpu_timeline.py $OUT > gpurun_out/$TAG/gpu_timeline_$name.txt 2>&1
  rm -f $OUT/*kernel_trace.csv $OUT/*agent_info.csv
  echo "== $name"; head -16 gpurun_out/$TAG/gpu_timeline_$name.txt
done
