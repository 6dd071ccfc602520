#!/bin/bash
# developer A/B on the GPU box: the float16 frame-buffer chain of preset R (4K x 512) -- shipped: hot blocks, 2 rows in flight; variants
# built by hand: row-at-a-time form (no hot blocks) with 4 / 8 rows in flight (librespmon_hip_f16h0_pf{4,8}.so), alternating processes
cd respmon_amd/csrc; cp librespmon_hip.so /tmp/base.so; cp librespmon_hip_f16h0_pf4.so /tmp/h0pf4.so; cp librespmon_hip_f16h0_pf8.so /tmp/h0pf8.so; cd ../..
for r in 1 2; do
  for v in base h0pf4 h0pf8; do
    cp /tmp/$v.so respmon_amd/csrc/librespmon_hip.so
    echo "== $v"; timeout 300 python tools/ab_inproc.py --config R --rounds 2 --steps 8 "dc_segs=0" "dc_segs=2" 2>&1 | grep "step ms"
  done
done
cp /tmp/base.so respmon_amd/csrc/librespmon_hip.so
