#!/bin/bash
# developer sweep of the frame-buffer kernel's knobs on the GPU box: prefetch depth (compile time) x segments x waves per
# workgroup (env); prints kernel ms per launch from bench.py's HIP events.  Restores the default library afterwards.
cd "$GRAFT_REPO_ROOT/respmon_amd/csrc" || exit 1
cp librespmon_hip.so /tmp/librespmon_hip.default.so
for pf in 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -DRM_DC_PREFETCH=$pf -shared -o librespmon_hip.so rm_unity.hip rm_contour.cpp 2>/dev/null || { echo "build pf=$pf failed"; continue; }
  for segs in 1 2 3; do
    for wpg in 1 2; do
      r=$(cd "$GRAFT_REPO_ROOT" && RM_DC_SEGS=$segs RM_DC_WPG=$wpg timeout 120 python bench.py --steps 100 --warmup 10 --prewarm-steps 200 --cpu-frames 0 --no-u8-alt --no-roi-flow 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f kernel_ms %.4f step_ms roi %s' % (d['roofline']['kernel_ms'], d['ms_per_step'], d.get('roi')))" 2>&1 | tail -1)
      echo "pf=$pf segs=$segs wpg=$wpg : $r"
    done
  done
done
cp /tmp/librespmon_hip.default.so librespmon_hip.so
