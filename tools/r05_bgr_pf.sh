#!/bin/bash
# developer A/B on the GPU box: RM_BGR_PREFETCH 2 (shipped) against 4 (librespmon_hip_pf4.so, built by hand), alternating processes
mkdir -p gpurun_out/r05b
cd respmon_amd/csrc; cp librespmon_hip.so /tmp/pf2.so; cp librespmon_hip_pf4.so /tmp/pf4.so; cd ../..
for r in 1 2; do
  for v in pf2 pf4; do
    cp /tmp/$v.so respmon_amd/csrc/librespmon_hip.so
    echo "== $v"; timeout 300 python tools/ab_inproc.py --in-dtype bgr8 --rounds 2 --steps 100 "dc_segs=0" "dc_segs=3" "dc_segs=2" 2>&1 | grep "step ms"
  done
done
cp /tmp/pf2.so respmon_amd/csrc/librespmon_hip.so
