#!/bin/bash
# developer: where k_temporal_sym_px (4K x 512, preset R) spends its cycles -- SQ / MFMA / TCP counters, rocprofv3 --pmc, separate passes
#   gpurun -- bash tools/pmc_temporal.sh   -> gpurun_out/pmc_temporal/summary.txt
set -u
OUT=gpurun_out/pmc_temporal; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --config R --steps 3 --warmup 1 --prewarm-steps 3 --cpu-frames 0 --no-extras"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVES --output-format csv -d $OUT/a -o a -- $B > /dev/null 2> $OUT/a.err < /dev/null
# (the TCP_* pass produced no rows on this image)
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_VALU TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/c -o c -- $B > /dev/null 2> $OUT/c.err < /dev/null
python - <<'PY' | tee gpurun_out/pmc_temporal/summary.txt
import csv, glob, collections
for p in ("a", "b", "c"):
    for f in glob.glob("gpurun_out/pmc_temporal/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "temporal_sym_px" in k or "k_dense_sum_t" in k or "k_frame_bounds_rows" in k:
                acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            print(p, k, {n: "%.4g" % (sum(v) / len(v)) for n, v in d.items()})
PY
