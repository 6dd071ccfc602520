#!/bin/bash
# per-kernel average durations of a short bench run (rocprofv3 --kernel-trace --stats): bash tools/kstats.sh [bench args]
mkdir -p gpurun_out/ks; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -o k -- python bench.py --steps 100 --warmup 10 --cpu-frames 0 --no-u8-alt --no-roi-flow "$@" > gpurun_out/ks/log.txt 2>&1 < /dev/null
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/ks/k_kernel_stats.csv')):
    n = r['Name']
    if 'rm::' in n or n.startswith('k_'):
        print("%-30s calls %5s avg %8.1f us" % (n.split('(')[0].replace('void ', '').replace('rm::', '')[:30], r['Calls'], float(r['AverageNs']) / 1e3))
PY
tail -1 gpurun_out/ks/log.txt | cut -c1-200
