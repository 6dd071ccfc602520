#!/bin/bash
# sweep of the frame-buffer kernel's workgroup width / sync form (rm_debug_set dc_wpg, dc_sync): bash tools/r04_sweep_dc.sh
OUT=gpurun_out/r04/sweep_dc; mkdir -p $OUT
run() {
  tag=$1; shift
  timeout 300 python bench.py --steps 200 --warmup 10 --cpu-frames 0 --no-extras "$@" > $OUT/$tag.json 2> $OUT/$tag.err < /dev/null
  python - <<PY
import json
try:
    d = json.load(open("$OUT/$tag.json")); r = d["roofline"]
    print("%-14s ms_per_step %.4f kernel_ms %.4f roi %s" % ("$tag", d["ms_per_step"], r["kernel_ms"], d["roi"]))
except Exception as e:
    print("$tag failed", e)
PY
}
run base
run w2_s1 --debug-set dc_sync=1
run w2_s2 --debug-set dc_sync=2
run w3_s0 --debug-set dc_wpg=3
run w3_s1 --debug-set dc_wpg=3 --debug-set dc_sync=1
run w3_s2 --debug-set dc_wpg=3 --debug-set dc_sync=2
run w3_s4 --debug-set dc_wpg=3 --debug-set dc_sync=4
run w6_s1 --debug-set dc_wpg=6 --debug-set dc_sync=1
run w6_s2 --debug-set dc_wpg=6 --debug-set dc_sync=2
run w6_s3 --debug-set dc_wpg=6 --debug-set dc_sync=3
run w6_s5 --debug-set dc_wpg=6 --debug-set dc_sync=5
run base2
