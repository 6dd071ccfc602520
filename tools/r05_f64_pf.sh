#!/bin/bash
# developer A/B on the GPU box: rows in flight of the float64 frame-buffer kernel (RM_DC_PREFETCH 2 shipped; 3 / 4 built by hand into
# librespmon_hip_f64pf{3,4}.so), alternating processes
cd respmon_amd/csrc; cp librespmon_hip.so /tmp/pf2.so; cp librespmon_hip_f64pf3.so /tmp/pf3.so; cp librespmon_hip_f64pf4.so /tmp/pf4.so; cd ../..
for r in 1 2 3; do
  for v in pf2 pf3 pf4; do
    cp /tmp/$v.so respmon_amd/csrc/librespmon_hip.so
    echo "== $v"; timeout 300 python tools/ab_inproc.py --rounds 2 --steps 100 "dc_segs=0" 2>&1 | grep "step ms"
  done
done
cp /tmp/pf2.so respmon_amd/csrc/librespmon_hip.so
