#!/bin/bash
# worst-case streams under several value-store capacities: bash tools/r04_worst.sh
mkdir -p gpurun_out/r04
for v in ${VARIANTS:-default}; do
  args=""; [ "$v" != "default" ] && args="--debug-set $v"
  timeout 900 python bench.py --steps 50 --no-configs --cpu-frames 0 --no-u8-alt --no-roi-flow --no-batches $args > gpurun_out/r04/worst_$v.json 2> gpurun_out/r04/worst_$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r04/worst_$v.json"))
w = d["worst_case"]
print("$v headline %.4f no_prune %.3f dense %.3f" % (d["ms_per_step"], d["no_prune"]["ms_per_step"], d["dense_stream"]["ms_per_step"]))
for k in ("noise", "blobs16"):
    print("   %-8s %.4f ms  x%.2f  roi ok %s  %s" % (k, w[k]["ms_per_step"], w[k]["vs_headline"], w[k]["roi_equals_oracle"], w[k]["collapse_pairs"]))
PY
done
