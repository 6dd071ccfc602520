#!/bin/bash
# developer: SQ counters of the register-resident chain kernels (uint8 / float32 buffers of preset P, float16 buffer of preset R)
#   gpurun -- bash tools/pmc_narrow.sh      -> gpurun_out/pmc_narrow/<dt>/a_counter_collection.csv ; python tools/summarize_narrow.py
set -u
OUT=gpurun_out/pmc_narrow; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for dt in u8 f32 f16; do
B="python bench.py --steps 5 --warmup 1 --prewarm-steps 5 --cpu-frames 0 --no-extras --in-dtype $dt"
[ "$dt" = "f16" ] && B="python bench.py --config R --steps 3 --warmup 1 --prewarm-steps 3 --cpu-frames 0 --no-extras"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/$dt -o a -- $B > /dev/null 2> $OUT/$dt.err < /dev/null
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/${dt}_b -o b -- $B > /dev/null 2>> $OUT/$dt.err < /dev/null
done
ls -R $OUT | head -30
