#!/bin/bash
# round 6, step B: the exception store (rm_xstore.h) -- GPU parity, then per-kernel times and A/B on R, Q and the noise stream
OUT=gpurun_out/r06/b
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_calibration.py -x -q -m gpu -k "level1 or streaming_tile or dense_sum_equals or exception_store or config_q or config_r_fp16 or fused_collapse or value_store" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 600 python tools/ab_inproc.py --config R --rounds 3 --steps 20 "xs=0@1" "xs=1" > $OUT/ab_R.txt 2>&1; tail -3 $OUT/ab_R.txt
timeout 600 python tools/ab_inproc.py --config Q --rounds 3 --steps 200 "xs=0@1" "xs=1" "xs=1,xs_waves=2" "xs=1,xs_waves=1" "xs=1,bounds_l1_rows=16" > $OUT/ab_Q.txt 2>&1; tail -6 $OUT/ab_Q.txt
timeout 600 python tools/ab_inproc.py --config P --video noise --rounds 3 --steps 100 "xs=0@1" "xs=1" "xs=1,xs_waves=2" > $OUT/ab_Pnoise.txt 2>&1; tail -4 $OUT/ab_Pnoise.txt
timeout 600 python tools/ab_inproc.py --config P --video blobs16 --rounds 3 --steps 100 "xs=0@1" "xs=1" > $OUT/ab_Pblobs.txt 2>&1; tail -3 $OUT/ab_Pblobs.txt
bash tools/r05_kstats.sh r06b_R --config R --steps 10 --warmup 3 2>&1 | tail -16
bash tools/r05_kstats.sh r06b_Q --config Q --steps 50 --warmup 5 2>&1 | tail -16
bash tools/r05_kstats.sh r06b_Pnoise --config P --video noise --steps 50 --warmup 5 2>&1 | tail -16
