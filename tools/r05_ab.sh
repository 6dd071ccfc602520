#!/bin/bash
# round 5: bench A/B of rm_debug_set combinations on ONE box, interleaved and repeated (boxes differ by ~8 %)
#   bash tools/r05_ab.sh "dc_prio=0" "dc_prio=2" "dc_prio=2,dc_split=500" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do
for combo in "$@"; do
  args=""
  for kv in ${combo//,/ }; do args="$args --debug-set $kv"; done
  python bench.py --steps 200 --warmup 10 --cpu-frames 0 --no-extras --no-batches $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('%-40s rep $rep ms_per_step %.4f kernel_ms %.4f roi %s' % ('$combo', d['ms_per_step'], r['kernel_ms'], d['roi']))"
done
done
