#!/bin/bash
# round 6, final library: the evidence batch (tools/r06_final_cfgs.sh) and the final check (tools/r06_final_check.sh) on ONE box
bash tools/r06_final_cfgs.sh 2>&1 | tail -40
bash tools/r06_final_check.sh 2>&1 | tail -60
