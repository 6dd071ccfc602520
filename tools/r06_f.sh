#!/bin/bash
# round 6, step F: level-1 bounds in packed float32 (rm_bounds_l1.h k_frame_bounds_l1f)
OUT=gpurun_out/r06/f
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_calibration.py -x -q -m gpu -k "level1 or streaming_tile or config_q or config_r_fp16" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
timeout 600 python tools/ab_inproc.py --config R --rounds 3 --steps 20 "bounds_l1=2@2" "bounds_l1=1" > $OUT/ab_R.txt 2>&1; tail -3 $OUT/ab_R.txt
timeout 600 python tools/ab_inproc.py --config Q --rounds 3 --steps 200 "bounds_l1=2@2" "bounds_l1=1" "bounds_l1=0" "bounds_l1_rows=32" "bounds_l1_rows=8" > $OUT/ab_Q.txt 2>&1; tail -6 $OUT/ab_Q.txt
bash tools/r05_kstats.sh r06f_R --config R --steps 10 --warmup 3 2>&1 | grep "bounds\|dense_sum\|select"
bash tools/r05_kstats.sh r06f_Q --config Q --steps 50 --warmup 5 2>&1 | grep "bounds\|dense_sum\|select"
