#!/bin/bash
# developer sweep on the GPU box: frame-buffer kernel geometry (segments x waves per workgroup) for bench config Q / R
# usage: bash tools/sweep_q.sh Q "2 3 4 6 8" "1 2"
CFG=${1:-Q}; SEGS=${2:-"2 3 4 6 8"}; WPGS=${3:-"1 2"}
cd "$GRAFT_REPO_ROOT" || exit 1
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f kernel_ms %.4f step_ms roi %s' % (d['roofline']['kernel_ms'], d['ms_per_step'], d.get('roi')))" 2>&1 | tail -1; }
echo "default: $(timeout 120 python bench.py --config $CFG --cpu-frames 0 --no-extras 2>/dev/null | one)"
for segs in $SEGS; do for wpg in $WPGS; do
  echo "segs=$segs wpg=$wpg : $(RM_DC_SEGS=$segs RM_DC_WPG=$wpg timeout 120 python bench.py --config $CFG --allow-env-knobs --cpu-frames 0 --no-extras 2>/dev/null | one)"
done; done
