#!/bin/bash
# developer probe: the fast / slow state of consecutive identical processes on one box
cd "$GRAFT_REPO_ROOT"
for i in $(seq 1 12); do
  python tools/ab_inproc.py --rounds 1 --steps 60 "default" 2>&1 | grep "step ms" | sed "s/^/$i /"
done
