#!/bin/bash
# The round's randomised parity sweeps on the GPU box (gpurun -- bash tools/fuzz_round.sh <tag> <seed>): every mode of
# tools/fuzz_parity.py with fresh seeds, calibration once more in the reference's operation order.  -> gpurun_out/<tag>/fuzz_*.txt
set -u
TAG=${1:-r03}; SEED=${2:-930001}
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: python tools/fuzz_parity.py $*"; timeout 900 python tools/fuzz_parity.py "$@" > $OUT/fuzz_$name.txt 2>&1; tail -3 $OUT/fuzz_$name.txt; }
run calib 300 $SEED calib
run calib_reforder 300 $SEED calib reforder
run big 200 $((SEED + 1)) big
run api 120 $((SEED + 2)) api
run shard 120 $((SEED + 3)) shard
run run 120 $((SEED + 4)) run
run roi 120 $((SEED + 5)) roi
run flow 100 $((SEED + 6)) flow
