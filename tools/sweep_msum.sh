#!/bin/bash
# developer sweep: masked_sum_tiles batch depth x occupancy hint
cd "$GRAFT_REPO_ROOT/respmon_amd/csrc" || exit 1
cp librespmon_hip.so /tmp/librespmon_hip.default.so
for cfg in "4 5" "4 6" "3 4" "3 6" "5 3" "6 4" "5 4"; do
  set -- $cfg
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -DRM_MS_B=$1 -DRM_MS_MINBLK=$2 -shared -o librespmon_hip.so rm_api.hip rm_contour.cpp 2>/dev/null || { echo "build $cfg failed"; continue; }
  echo "== MS_B=$1 minblk=$2"
  (cd "$GRAFT_REPO_ROOT" && bash tools/kstats.sh | grep -E "masked_sum|eval_pairs")
done
cp /tmp/librespmon_hip.default.so librespmon_hip.so
