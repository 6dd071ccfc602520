#!/bin/bash
# Collect the evidence profiles/ holds for one round, on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01
# 1. bench.py default run (the line the driver records)          -> gpurun_out/<tag>/bench_n1.json
# 2. rocprofv3 --kernel-trace --stats of the same command         -> gpurun_out/<tag>/stats/
# 3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md PMC slots)
#    with no trace domains beside them                            -> gpurun_out/<tag>/pmc_fetch, pmc_write
# tools/summarize_profiles.py (run afterwards, anywhere) turns these into profiles/<tag>/ + profiles/hbm_traffic.json.
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python bench.py --steps 100 --warmup 5"
timeout 600 $BENCH > $OUT/bench_n1.json 2> $OUT/bench_n1.err < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- $BENCH --cpu-frames 0 --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/rocprof_stats.err < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- $BENCH --steps 20 --cpu-frames 0 --no-extras --no-u8-alt --no-roi-flow > /dev/null 2> $OUT/rocprof_fetch.err < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- $BENCH --steps 20 --cpu-frames 0 --no-extras --no-u8-alt --no-roi-flow > /dev/null 2> $OUT/rocprof_write.err < /dev/null
ls -R $OUT | head -40
