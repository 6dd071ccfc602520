#!/bin/bash
# developer: SQ counters of the register-resident chain kernel for uint8 / float32 frame buffers
set -u
OUT=gpurun_out/pmc_narrow; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for dt in u8 f32; do
B="python bench.py --steps 5 --warmup 1 --prewarm-steps 5 --cpu-frames 0 --no-u8-alt --no-roi-flow --in-dtype $dt"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/$dt -o a -- $B > /dev/null 2> $OUT/$dt.err < /dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${dt}_fetch -o f -- $B > /dev/null 2>> $OUT/$dt.err < /dev/null
done
ls -R $OUT | head -20
