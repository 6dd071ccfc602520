#!/bin/bash
# k_ccl_bbox with / without its LDS table on the speckled presets (Q, R) and the noise-blob stream: per-kernel averages
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "Q:--config Q:60" "R:--config R:10" "noise:--video noise:40"; do
  name=${cfg%%:*}; rest=${cfg#*:}; args=${rest%%:*}; st=${rest##*:}
  for tb in 1 0; do
    OUT=gpurun_out/r04/cclab_${name}_$tb; mkdir -p $OUT
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python bench.py --steps $st --warmup 3 $args --cpu-frames 0 --no-extras --no-batches --debug-set ccl_table=$tb > $OUT/b.json 2>/dev/null < /dev/null
    python - <<PY
import csv, glob, json
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
d = json.load(open("$OUT/b.json"))
ks = {r["Name"].split("(")[0].replace("void ", "").replace("rm::", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
print("$name table=$tb step %.4f ms  " % d["ms_per_step"], {k: round(v, 1) for k, v in ks.items() if "ccl" in k or "heat_to" in k}, d["roi"])
PY
    rm -rf $OUT
  done
done
