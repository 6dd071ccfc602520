#!/bin/bash
# round 6, step H: level-1 bounds kernel without the column-long lattice-row search (the tile row of the running extremes is tracked)
OUT=gpurun_out/r06/h
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_calibration.py -x -q -m gpu -k "level1 or refined or dense_sum_equals or streaming_tile" > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
bash tools/r05_kstats.sh r06h_R --config R --steps 30 --warmup 3 2>&1 | grep -E "bounds|dense_sum|select" 
bash tools/r05_kstats.sh r06h_Q --config Q --steps 100 --warmup 5 2>&1 | grep -E "bounds|dense_sum|select"
timeout 600 python tools/ab_inproc.py --config R --rounds 3 --steps 30 "bounds_l1=2@2" "bounds_l1=1" > $OUT/ab_R.txt 2>&1; tail -3 $OUT/ab_R.txt
timeout 600 python tools/ab_inproc.py --config Q --rounds 3 --steps 150 "bounds_l1=2@2" "bounds_l1=1" > $OUT/ab_Q.txt 2>&1; tail -3 $OUT/ab_Q.txt
