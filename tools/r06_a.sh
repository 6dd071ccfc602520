#!/bin/bash
# round 6, step A: level-1 tile bounds (rm_bounds_l1.h) -- parity on the GPU, then A/B inside one process on R and Q
OUT=gpurun_out/r06/a
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_calibration.py -x -q -m gpu -k "level1 or streaming_tile or dense_sum_equals or config_q or config_r_fp16" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 600 python tools/ab_inproc.py --config R --rounds 3 --steps 20 "bounds_l1=0@1" "bounds_l1=1" > $OUT/ab_R.txt 2>&1
cat $OUT/ab_R.txt | tail -4
timeout 600 python tools/ab_inproc.py --config Q --rounds 3 --steps 200 "bounds_l1=0@1" "bounds_l1=1" "bounds_l1=1,dense_t_low=1@-1" "bounds_l1=0,dense_t_low=1" > $OUT/ab_Q.txt 2>&1
cat $OUT/ab_Q.txt | tail -6
bash tools/r05_kstats.sh r06a_R --config R --steps 10 --warmup 3 2>&1 | tail -20
bash tools/r05_kstats.sh r06a_Q --config Q --steps 50 --warmup 5 2>&1 | tail -20
timeout 300 python tools/host_timeline.py 20 Q > $OUT/host_Q.txt 2>&1; tail -8 $OUT/host_Q.txt
timeout 300 python tools/host_timeline.py 8 R > $OUT/host_R.txt 2>&1; tail -8 $OUT/host_R.txt
