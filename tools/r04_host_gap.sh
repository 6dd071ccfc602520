#!/bin/bash
# where the GPU idle between two synchronous steps goes: HIP API trace + kernel trace of a short bench run
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04/hostgap; mkdir -p $OUT
timeout 600 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $OUT -o g -- python bench.py --steps 40 --warmup 5 --prewarm-steps 20 --cpu-frames 0 --no-extras --no-batches > $OUT/bench.json 2> $OUT/err.txt < /dev/null
ls $OUT
python - <<'PY'
import csv, glob
d = "gpurun_out/r04/hostgap"
kt = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])) if "rm::" in r["Kernel_Name"])
api = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0])))
import bisect
starts = [a[0] for a in api]
rows = []
for i, k in enumerate(kt[:-1]):
    if "k_heat_to_u8" in k[2] and "k_down_chain" in kt[i + 1][2]:
        end = k[1]; nxt = kt[i + 1][0]
        j = bisect.bisect_left(starts, end - 2000000)
        seq = [(a[0] - end, a[1] - end, a[2]) for a in api[j:] if a[0] < nxt + 20000 and a[1] > end - 5000]
        rows.append((nxt - end, seq))
rows = rows[-16:]
for gap, seq in rows[-11:-7]:
    print("gap %.1f us" % (gap / 1e3))
    for s, e, f in seq:
        print("   %8.1f .. %8.1f  %s" % (s / 1e3, e / 1e3, f))
PY
rm -f $OUT/*/*kernel_trace.csv $OUT/*kernel_trace.csv
