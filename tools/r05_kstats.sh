#!/bin/bash
# round 5: per-kernel times of one bench.py configuration under rocprofv3 (kernel trace only)
#   bash tools/r05_kstats.sh <tag> <bench args...>
TAG=$1; shift
OUT=gpurun_out/r05/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python bench.py --cpu-frames 0 --no-extras --no-batches "$@" > $OUT/bench.json 2> $OUT/err.txt < /dev/null
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/stats/k_kernel_stats.csv")))
print("$TAG")
for r in rows[:16]:
    if "rm::" in r["Name"]:
        print("   %-64s calls %5s avg %10.1f us" % (r["Name"].split("(")[0][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -f $OUT/stats/*kernel_trace.csv $OUT/stats/*agent_info.csv
