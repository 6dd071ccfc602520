#!/bin/bash
# round 6, step O: the end of a bounds-kernel wave's life (six state reads requested together, the two lattice look-ups side by side)
OUT=gpurun_out/r06/o
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_calibration.py -x -q -m gpu -k "level1 or dense_sum_equals or streaming_tile" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
bash tools/r05_kstats.sh r06o_Q --config Q --steps 100 --warmup 4 2>&1 | grep -E "bounds|select"
bash tools/r05_kstats.sh r06o_R --config R --steps 20 --warmup 3 2>&1 | grep -E "bounds|select"
timeout 600 python tools/ab_inproc.py --config Q --rounds 3 --steps 150 "bounds_l1=2@2" "bounds_l1=1" > $OUT/ab_Q.txt 2>&1; tail -2 $OUT/ab_Q.txt
