#!/bin/bash
# developer probe: what decides the fast / slow state of a process (frame-buffer kernel 0.69 / 0.74 ms for one binary on one box)?
# CPU placement of the process relative to the GPU's NUMA node, first.
cd "$GRAFT_REPO_ROOT"
lscpu | grep -i "numa\|socket\|model name\|^CPU(s)" | head -12
rocm-smi --showtopo 2>/dev/null | grep -i "numa" | head -6
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -4
run() { echo "== $*"; "$@" python tools/ab_inproc.py --rounds 1 --steps 80 "default" 2>&1 | grep "step ms"; }
NODES=$(ls -d /sys/devices/system/node/node* 2>/dev/null | wc -l)
echo "numa nodes: $NODES"
for rep in 1 2; do
  run env
  for n in $(seq 0 $((NODES - 1))); do
    CPUS=$(cat /sys/devices/system/node/node$n/cpulist)
    run taskset -c $CPUS
  done
done
