#!/bin/bash
# developer: build bench_micro/dc8_bench.hip in several variants on the GPU box and run them (gpurun -- bash tools/sweep_dc8.sh)
set -u
cd "$GRAFT_REPO_ROOT/bench_micro"
OUT=../gpurun_out/dc8; mkdir -p $OUT
CC="/opt/rocm/bin/hipcc -I../respmon_amd/csrc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-function -Wno-unused-value"
build() { name=$1; shift; $CC "$@" -o dc_$name dc8_bench.hip 2> $OUT/build_$name.err || { echo "build $name failed"; tail -5 $OUT/build_$name.err; }; }
cp old_rm_down_chain_u8.h.txt /tmp/old_dc8.h
build old '-DRM_DC8_HEADER="/tmp/old_dc8.h"' &
build nodma -DRM_NARROW_DMA=0 &
build dma &
build dma_r4 -DRM_F16_RING=4 &
build dma_r16 -DRM_F16_RING=16 -DRM_F32_RING=8 &
build dma_w3 -DRM_NARROW_WAVES=3 &
wait
for v in old nodma dma dma_r4 dma_r16 dma_w3; do echo "== $v"; [ -x dc_$v ] && timeout 120 ./dc_$v "$@"; done 2>&1 | tee $OUT/sweep.txt
