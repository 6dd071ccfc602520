#!/bin/bash
# A/B of collapse variants on one box: bash tools/r04_ab.sh  (per-kernel averages from rocprofv3 for each variant)
OUT=gpurun_out/r04/ab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  tag=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o k -- python bench.py --steps 100 --warmup 10 --cpu-frames 0 --no-extras "$@" > $OUT/$tag.json 2> $OUT/$tag.err < /dev/null
  python tools/gpu_timeline.py $OUT/$tag > $OUT/$tag.timeline.txt 2>&1
  rm -rf $OUT/$tag
  echo "== $tag"; cat $OUT/$tag.timeline.txt
}
for v in "$@"; do
  case $v in
    fused) run fused ;;
    store_fast) run store_fast --debug-set collapse_fused=0 ;;
    store_old) run store_old --debug-set collapse_fused=0 --debug-set eval_fast=0 ;;
    *) run "$v" $(echo $v | tr ',' ' ' | sed 's/\([a-z_]*=[-0-9]*\)/--debug-set \1/g') ;;
  esac
done
