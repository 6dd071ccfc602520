#!/bin/bash
# round 6, step C: k_dense_sum_t walking the kept list (no level-1 stop behind level-1 bounds), 4 waves per SIMD at skip <= 2
OUT=gpurun_out/r06/c
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_calibration.py -x -q -m gpu -k "level1 or dense_sum_equals or exception_store or config_q or config_r_fp16 or fused_collapse or value_store" > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 600 python tools/ab_inproc.py --config R --rounds 3 --steps 20 "bounds_l1=1@1" "bounds_l1=0" "bounds_l1_rows=64" > $OUT/ab_R.txt 2>&1; tail -4 $OUT/ab_R.txt
timeout 600 python tools/ab_inproc.py --config Q --rounds 3 --steps 200 "bounds_l1=1@1" "bounds_l1=0" "dense_t_low=1@-1" "bounds_l1_rows=16" > $OUT/ab_Q.txt 2>&1; tail -5 $OUT/ab_Q.txt
timeout 600 python tools/ab_inproc.py --config P --video noise --rounds 3 --steps 100 "xs=0" "xs=1" > $OUT/ab_Pnoise.txt 2>&1; tail -3 $OUT/ab_Pnoise.txt
bash tools/r05_kstats.sh r06c_R --config R --steps 10 --warmup 3 2>&1 | tail -14
bash tools/r05_kstats.sh r06c_Pnoise --config P --video noise --steps 50 --warmup 5 2>&1 | tail -8
