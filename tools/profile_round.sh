#!/bin/bash
# Collect the evidence profiles/ holds for one round, on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r03 [P|Q|R|P32|P8]
# 1. bench.py run of the preset (P: the default command the driver records) -> gpurun_out/<tag>[/<cfg>]/bench_n1.json
# 2. rocprofv3 --kernel-trace --stats of the same command                      -> .../stats/
# 3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md PMC slots)
#    with no trace domains beside them                                         -> .../pmc_fetch, pmc_write
# tools/summarize_profiles.py (run afterwards, anywhere) turns these into profiles/<tag>/ + profiles/hbm_traffic.json.
set -u
TAG=${1:-r01}
CFG=${2:-P}
OUT=gpurun_out/$TAG
ARGS=""
STEPS=100
case $CFG in
  P) ;;
  Q) OUT=$OUT/Q; ARGS="--config Q" ;;
  R) OUT=$OUT/R; ARGS="--config R"; STEPS=20 ;;
  P32) OUT=$OUT/P32; ARGS="--in-dtype f32" ;;
  P8) OUT=$OUT/P8; ARGS="--in-dtype u8" ;;
  PBGR) OUT=$OUT/PBGR; ARGS="--in-dtype bgr8" ;;
esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python bench.py --steps $STEPS --warmup 5 $ARGS"
if [ "$CFG" = "P" ]; then
  timeout 900 $BENCH > $OUT/bench_n1.json 2> $OUT/bench_n1.err < /dev/null
else
  timeout 900 $BENCH --no-extras > $OUT/bench_n1.json 2> $OUT/bench_n1.err < /dev/null
fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- $BENCH --cpu-frames 0 --no-extras --no-batches > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err < /dev/null
PSTEPS=20; [ "$CFG" = "R" ] && PSTEPS=6
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python bench.py --steps $PSTEPS --warmup 2 --prewarm-steps 5 $ARGS --cpu-frames 0 --no-extras > /dev/null 2> $OUT/rocprof_fetch.err < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python bench.py --steps $PSTEPS --warmup 2 --prewarm-steps 5 $ARGS --cpu-frames 0 --no-extras > /dev/null 2> $OUT/rocprof_write.err < /dev/null
ls -R $OUT | head -30
