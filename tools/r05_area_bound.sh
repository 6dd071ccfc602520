#!/bin/bash
# developer check on the GPU box: labelled-path tests, then the noise / Q / R streams with and without the area-bound shortcut
#   (in-process A/B: tools/ab_inproc.py) and the noise stream's timeline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05b/wc_noise
timeout 900 python -m pytest tests/test_gpu_calibration.py tests/test_gpu_ingest.py -m gpu -x -q -k "labelling or contour or simple_shape" 2>&1 | tail -5
timeout 600 python tools/ab_inproc.py --video noise --rounds 3 --steps 60 "host_area_bound=1" "host_area_bound=0" 2>&1 | grep "step ms"
timeout 600 python tools/ab_inproc.py --config Q --rounds 2 --steps 100 "host_area_bound=1" "host_area_bound=0" 2>&1 | grep "step ms"
timeout 600 python tools/ab_inproc.py --config R --rounds 2 --steps 10 "host_area_bound=1" "host_area_bound=0" 2>&1 | grep "step ms"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05b/wc_noise -o k -- python bench.py --steps 60 --warmup 5 --prewarm-steps 30 --video noise --cpu-frames 0 --no-extras --no-batches > gpurun_out/r05b/wc_noise/bench.json 2> gpurun_out/r05b/wc_noise/err.txt < /dev/null
python tools/gpu_timeline.py gpurun_out/r05b/wc_noise > gpurun_out/r05b/gpu_timeline_worst_case_noise.txt 2>&1
rm -f gpurun_out/r05b/wc_noise/*kernel_trace.csv gpurun_out/r05b/wc_noise/*agent_info.csv
head -14 gpurun_out/r05b/gpu_timeline_worst_case_noise.txt
