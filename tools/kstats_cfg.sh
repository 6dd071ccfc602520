#!/bin/bash
# per-kernel average durations of a short bench run of one preset (rocprofv3 --kernel-trace --stats):
#   bash tools/kstats_cfg.sh P|Q|R [bench args]      -> gpurun_out/ks_<cfg>/
CFG=${1:-P}; shift
OUT=gpurun_out/ks_$CFG
mkdir -p $OUT; export TMPDIR=/tmp
STEPS=100; [ "$CFG" = "R" ] && STEPS=20
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python bench.py --config $CFG --steps $STEPS --warmup 5 --cpu-frames 0 --no-extras "$@" > $OUT/log.txt 2>&1 < /dev/null
python - "$OUT" <<'PY'
import csv, sys
out = sys.argv[1]
tot = 0.0
for r in csv.DictReader(open(out + '/k_kernel_stats.csv')):
    n = r['Name']
    if 'rm::' in n or n.startswith('k_'):
        print("%-34s calls %5s avg %9.1f us" % (n.split('(')[0].replace('void ', '').replace('rm::', '')[:34], r['Calls'], float(r['AverageNs']) / 1e3))
PY
tail -1 $OUT/log.txt | cut -c1-400
