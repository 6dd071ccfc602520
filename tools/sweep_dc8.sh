#!/bin/bash
# developer: build bench_micro/dc8_bench.hip in several variants on the GPU box and run them (gpurun -- bash tools/sweep_dc8.sh)
set -u
cd "$GRAFT_REPO_ROOT/bench_micro"
OUT=../gpurun_out/dc8; mkdir -p $OUT
CC="/opt/rocm/bin/hipcc -I../respmon_amd/csrc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-function -Wno-unused-value"
build() { name=$1; shift; $CC "$@" -o dc_$name dc8_bench.hip 2> $OUT/build_$name.err || { echo "build $name failed"; tail -5 $OUT/build_$name.err; }; }
cp old_rm_down_chain_u8.h.txt /tmp/old_dc8.h
build old '-DRM_DC8_HEADER="/tmp/old_dc8.h"' &
build w2pf2 -DRM_NARROW_WAVES=2 -DRM_U8_PREFETCH=2 -DRM_F16_PREFETCH=2 &
build w2pf42 -DRM_NARROW_WAVES=2 -DRM_U8_PREFETCH=4 -DRM_F16_PREFETCH=2 &
wait
for v in old w2pf2 w2pf42; do for segs in 0 2 3 6 8; do echo "== $v segs=$segs"; [ -x dc_$v ] && timeout 120 ./dc_$v $segs; done; done 2>&1 | tee $OUT/sweep.txt
