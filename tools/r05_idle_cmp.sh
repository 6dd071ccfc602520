#!/bin/bash
# developer: GPU idle between synchronous steps, two checkouts on one box (rocprofv3 kernel trace of bench.py --no-batches)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for r in 1 2; do for d in . _old; do
  OUT=gpurun_out/idle_cmp/$r$(echo $d | tr -d ./); mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $d/bench.py --steps 100 --warmup 5 --cpu-frames 0 --no-extras --no-batches > $OUT/bench.json 2> $OUT/err.txt < /dev/null
  echo "== $d"; python tools/gpu_timeline.py $OUT 2>&1 | head -1
  rm -f $OUT/*kernel_trace.csv $OUT/*agent_info.csv
done; done
