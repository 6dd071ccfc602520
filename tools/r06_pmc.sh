#!/bin/bash
# round 6: SQ counters of every kernel of one preset (where do the waves of the collapse kernels spend their cycles?)
#   bash tools/r06_pmc.sh <tag> <bench args...>
set -u
TAG=$1; shift
OUT=gpurun_out/r06/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 4 --warmup 1 --prewarm-steps 3 --cpu-frames 0 --no-extras --no-batches $*"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $OUT/a -o a -- $B > /dev/null 2> $OUT/a.err < /dev/null
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -o b -- $B > /dev/null 2> $OUT/b.err < /dev/null
python - <<PY
import csv, re
from collections import defaultdict
def load(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "rm::" not in r["Kernel_Name"]:
            continue
        k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("rm::", "")[:30]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
a = load("$OUT/a/a_counter_collection.csv"); b = load("$OUT/b/b_counter_collection.csv")
avg = lambda d, n: (sum(d[n]) / len(d[n]) if d[n] else 0.0)
print("$TAG")
print("%-30s %8s %13s %7s %8s %8s %9s" % ("kernel", "waves", "wave_cycles/4", "wait%", "istall%", "active%", "lds_stall%"))
for k in a:
    wc = avg(a[k], "SQ_WAVE_CYCLES") or 1.0
    print("%-30s %8.0f %13.0f %7.1f %8.1f %8.1f %9.1f" % (k, avg(a[k], "SQ_WAVES"), wc, 100 * avg(a[k], "SQ_WAIT_ANY") / wc, 100 * avg(a[k], "SQ_WAIT_INST_ANY") / wc, 100 * avg(a[k], "SQ_ACTIVE_INST_ANY") / wc, 100 * avg(a[k], "SQ_WAIT_INST_LDS") / wc))
print("%-30s %10s %10s %9s %9s %9s %10s" % ("kernel", "valu", "salu", "lds", "vmem_rd", "vmem_wr", "bank_confl"))
for k in b:
    print("%-30s %10.0f %10.0f %9.0f %9.0f %9.0f %10.0f" % (k, avg(b[k], "SQ_INSTS_VALU"), avg(b[k], "SQ_INSTS_SALU"), avg(b[k], "SQ_INSTS_LDS"), avg(b[k], "SQ_INSTS_VMEM_RD"), avg(b[k], "SQ_INSTS_VMEM_WR"), avg(b[k], "SQ_LDS_BANK_CONFLICT")))
PY
rm -rf $OUT/a/*agent_info.csv $OUT/b/*agent_info.csv
