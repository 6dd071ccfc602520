#!/bin/bash
# headline bench with the `pipelined` extra key: bash tools/r04_pipe_bench.sh
mkdir -p gpurun_out/r04
python bench.py --steps 200 --warmup 10 --cpu-frames 0 --no-extras > gpurun_out/r04/pipe_bench.json 2> gpurun_out/r04/pipe_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04/pipe_bench.json"))
print("sync", d["ms_per_step"], d["ms_per_step_batches"]["batches"], "kernel", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
print("pipelined", d["pipelined"])
PY
tail -3 gpurun_out/r04/pipe_bench.err
