#!/bin/bash
# round 6, step A': restructured level-1 bounds kernel -- GPU parity of the bounds, kernel times on R and Q
OUT=gpurun_out/r06/a2
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_calibration.py -x -q -m gpu -k "level1 or streaming_tile" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for rows in 0 8 32; do
  echo "bounds_l1_rows=$rows"
  bash tools/r05_kstats.sh r06a2_R_$rows --config R --steps 10 --warmup 3 --debug-set bounds_l1_rows=$rows 2>&1 | grep "bounds_l1\|dense_sum"
done
for rows in 0 2 8; do
  echo "bounds_l1_rows=$rows"
  bash tools/r05_kstats.sh r06a2_Q_$rows --config Q --steps 50 --warmup 5 --debug-set bounds_l1_rows=$rows 2>&1 | grep "bounds_l1\|dense_sum"
done
