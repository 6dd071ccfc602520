#!/bin/bash
# round 6, step D: k_dense_sum_wf walking the kept list (720p x 128)
OUT=gpurun_out/r06/d
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_calibration.py -x -q -m gpu -k "level1 or dense_sum_equals or config_q or fused_collapse" > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 600 python tools/ab_inproc.py --config Q --rounds 3 --steps 200 "dense_wf_list=1@1" "dense_wf_list=0" "dense_wf_list=0,bounds_l1=0@1" "bounds_l1_rows=16" "dense_split=2" > $OUT/ab_Q.txt 2>&1; tail -6 $OUT/ab_Q.txt
bash tools/r05_kstats.sh r06d_Q --config Q --steps 50 --warmup 5 2>&1 | tail -15
