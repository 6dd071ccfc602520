#!/bin/bash
# narrow-dtype chain variants after the scratch fix: gpurun -- bash tools/r04_sweep_narrow.sh
set -u
cd "$GRAFT_REPO_ROOT/bench_micro"
OUT=../gpurun_out/r04/narrow; mkdir -p $OUT
CC="/opt/rocm/bin/hipcc -I../respmon_amd/csrc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-function -Wno-unused-value"
build() { name=$1; shift; $CC "$@" -o dc_$name dc8_bench.hip 2> $OUT/build_$name.err || { echo "build $name failed"; tail -5 $OUT/build_$name.err; }; }
build base &
build f32hot -DRM_F32_HOT=1 &
build f32hot_pf4 -DRM_F32_HOT=1 -DRM_F32_PREFETCH=4 &
build f32_pf4 -DRM_F32_PREFETCH=4 &
build f16_pf4 -DRM_F16_PREFETCH=4 &
build u8_pf2 -DRM_U8_PREFETCH=2 &
wait
for v in base f32hot f32hot_pf4 f32_pf4 f16_pf4 u8_pf2; do for segs in 0 2 3; do echo "== $v segs=$segs"; [ -x dc_$v ] && timeout 120 ./dc_$v $segs; done; done 2>&1 | tee $OUT/sweep.txt
