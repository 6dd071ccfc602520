#!/bin/bash
# headline + data-dependence extras of bench.py in one run: bash tools/r04_extras.sh <tag> [bench args]
TAG=${1:-x}; shift
mkdir -p gpurun_out/r04
timeout 900 python bench.py --steps 100 --no-configs --cpu-frames 0 "$@" > gpurun_out/r04/extras_$TAG.json 2> gpurun_out/r04/extras_$TAG.err
python - <<PY
import json
d = json.load(open("gpurun_out/r04/extras_$TAG.json"))
print("$TAG ms %.4f kernel %.4f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]), d["collapse_pairs"])
print("  no_prune", d["no_prune"])
print("  dense_stream", {k: d["dense_stream"][k] for k in ("ms_per_step", "roi", "collapse_pairs")})
print("  u8 %.4f" % d["alt_uint8_buffer"]["ms_per_step"], "worst_case", d.get("worst_case"))
PY
