#!/bin/bash
# round 6, step I: components of a 64 x 32 tile labelled in LDS (k_ccl_tile / k_ccl_seam / k_ccl_fold) against the global-memory kernels;
# waves per tile workgroup (the box arrays at the run starts only: 25 KB of LDS, four 8-wave workgroups per CU)
OUT=gpurun_out/r06/i
mkdir -p $OUT
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_calibration.py -x -q -m gpu -k "labelling or contour or roi" > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 600 python tools/ab_inproc.py --config P --video noise --rounds 3 --steps 100 "ccl_tile_waves=16@16" "ccl_tile_waves=8" "ccl_tile_waves=4" "ccl_tiles=0@1" > $OUT/ab_noise.txt 2>&1; tail -4 $OUT/ab_noise.txt
timeout 600 python tools/ab_inproc.py --config Q --rounds 3 --steps 150 "ccl_tile_waves=16@16" "ccl_tile_waves=8" "ccl_tile_waves=4" "ccl_tiles=0@1" > $OUT/ab_Q.txt 2>&1; tail -4 $OUT/ab_Q.txt
timeout 600 python tools/ab_inproc.py --config R --rounds 3 --steps 30 "ccl_tile_waves=16@16" "ccl_tile_waves=8" "ccl_tile_waves=4" "ccl_tiles=0@1" > $OUT/ab_R.txt 2>&1; tail -4 $OUT/ab_R.txt
timeout 400 python tools/fuzz_parity.py 240 960888 roi > $OUT/fuzz_roi.txt 2>&1; tail -1 $OUT/fuzz_roi.txt
