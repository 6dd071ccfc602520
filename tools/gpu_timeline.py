"""Developer tool: per-step GPU timeline from a rocprofv3 --kernel-trace CSV of bench.py (kernel durations, gaps between the
kernels of a step, and the GPU idle time between steps = host stage + launch latency).
    python tools/gpu_timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys

import numpy as np


def main(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if "rm::" in r["Kernel_Name"])
    # a step starts at its frame-buffer kernel: the k_down_chain* instantiation with the largest total time (configs with a deep
    # small pyramid run a second, smaller k_down_chain inside the step)
    tot = {}
    for s0, e0, nm in ks:
        if "k_down_chain" in nm:
            tot[nm] = tot.get(nm, 0) + e0 - s0
    lead = max(tot, key=tot.get)
    starts = [i for i, k in enumerate(ks) if k[2] == lead]
    per, names = [], None
    starts = starts[-80:]
    for a, b in zip(starts[:-1], starts[1:]):
        seq = ks[a:b]
        if names is None:
            names = [k[2].split("(")[0].replace("void ", "").replace("rm::", "") for k in seq]
        if [k[2].split("(")[0].replace("void ", "").replace("rm::", "") for k in seq] != names or b >= len(ks):
            continue
        per.append([ks[b][0] - seq[0][0], ks[b][0] - seq[-1][1]] + [e - s for s, e, _ in seq] + [seq[i + 1][0] - seq[i][1] for i in range(len(seq) - 1)])
    per = np.array(per) / 1e3
    m = per.mean(0)
    n = len(names)
    print("steps %d   step %.1f us   GPU idle between steps %.1f us   kernels %.1f   gaps %.1f" % (len(per), m[0], m[1], m[2:2 + n].sum(), m[2 + n:].sum()))
    for i, nm in enumerate(names):
        print("  %-34s %7.1f us   gap after %5.1f" % (nm[:34], m[2 + i], m[2 + n + i] if i < n - 1 else 0.0))


if __name__ == "__main__":
    main(sys.argv[1])
