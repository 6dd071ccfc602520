#!/bin/bash
# developer sweep on the GPU box: compile-time knobs of the frame-buffer kernel (prefetch depth, skip-2 strip width) for bench
# config Q; restores the default library afterwards.  usage: bash tools/sweep_q_build.sh "2 3 4" "64 80 88 108"
PFS=${1:-"2 3"}; SWS=${2:-"64 88 108"}
cd "$GRAFT_REPO_ROOT/respmon_amd/csrc" || exit 1
cp librespmon_hip.so /tmp/librespmon_hip.default.so
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f kernel_ms %.4f step_ms roi %s' % (d['roofline']['kernel_ms'], d['ms_per_step'], d.get('roi')))" 2>&1 | tail -1; }
echo "default: $(cd "$GRAFT_REPO_ROOT" && timeout 120 python bench.py --config Q --cpu-frames 0 --no-extras 2>/dev/null | one)"
for pf in $PFS; do for sw in $SWS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -DRM_DC_PREFETCH=$pf -DRM_DC_SW2=$sw -shared -o librespmon_hip.so rm_unity.hip rm_contour.cpp 2>/dev/null || { echo "build pf=$pf sw=$sw failed"; continue; }
  echo "pf=$pf sw2=$sw : $(cd "$GRAFT_REPO_ROOT" && timeout 120 python bench.py --config Q --cpu-frames 0 --no-extras 2>/dev/null | one)"
done; done
cp /tmp/librespmon_hip.default.so librespmon_hip.so
echo "default again: $(cd "$GRAFT_REPO_ROOT" && timeout 120 python bench.py --config Q --cpu-frames 0 --no-extras 2>/dev/null | one)"
