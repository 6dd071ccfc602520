#!/bin/bash
# round 5: issue priority of the frame-buffer kernel's waves (rm_debug_set dc_prio) -- bench A/B on one box + workgroup timelines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
for rep in 1 2; do
for p in 0 1 2 3; do
  python bench.py --steps 200 --warmup 10 --cpu-frames 0 --no-extras --no-batches --debug-set dc_prio=$p 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('dc_prio $p rep $rep ms_per_step %.4f kernel_ms %.4f' % (d['ms_per_step'], r['kernel_ms']))"
done
done
for p in 0 1 2; do
  python tools/trace_tail.py --config P --debug-set dc_prio=$p --out gpurun_out/r05/trace_prio$p 2>&1 | grep "k_down_chain \|^kernel"
done
