#!/bin/bash
# round 6, step L: k_select_pairs makes its three reservations from three lanes (one round trip instead of three in a row)
OUT=gpurun_out/r06/l
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_calibration.py -x -q -m gpu -k "level1 or refined or value_store or dense_sum_equals or fused_collapse or prun or select" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
for c in P Q R; do bash tools/r05_kstats.sh r06l_$c --config $c --steps 40 --warmup 4 2>&1 | grep -E "select_pairs"; done
bash tools/r05_kstats.sh r06l_noise --config P --video noise --steps 40 --warmup 4 2>&1 | grep -E "select_pairs"
