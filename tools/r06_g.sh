#!/bin/bash
# round 6, step G: skip 3 / 4 bounds refined one level down behind a dense selection (worst-case streams)
OUT=gpurun_out/r06/g
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_calibration.py -x -q -m gpu -k "refined or dense_sum_equals or value_store or fused_collapse" > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 600 python tools/ab_inproc.py --config P --video noise --rounds 3 --steps 100 "bounds_up1=-1@-1" "bounds_up1=0" "bounds_up1=1" > $OUT/ab_noise.txt 2>&1; tail -4 $OUT/ab_noise.txt
timeout 600 python tools/ab_inproc.py --config P --video blobs16 --rounds 3 --steps 100 "bounds_up1=-1@-1" "bounds_up1=1" > $OUT/ab_blobs.txt 2>&1; tail -3 $OUT/ab_blobs.txt
timeout 600 python tools/ab_inproc.py --config P --rounds 3 --steps 150 "bounds_up1=-1@-1" "bounds_up1=1" > $OUT/ab_P.txt 2>&1; tail -3 $OUT/ab_P.txt
bash tools/r05_kstats.sh r06g_noise --config P --video noise --steps 50 --warmup 5 2>&1 | tail -14
