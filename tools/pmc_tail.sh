#!/bin/bash
# developer: SQ counters of the small kernels after the frame-buffer kernel (where does each of them wait?)
set -u
OUT=gpurun_out/pmc_tail; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 5 --warmup 1 --prewarm-steps 5 --cpu-frames 0 --no-extras"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $OUT/a -o a -- $B > /dev/null 2> $OUT/a.err < /dev/null
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/b -o b -- $B > /dev/null 2> $OUT/b.err < /dev/null
ls -R $OUT | head; tail -3 $OUT/a.err $OUT/b.err
