"""gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) -> profiles/<tag>/ + profiles/hbm_traffic.json.

    python tools/summarize_profiles.py r03 [P|Q|R|P32|P8]

HBM traffic per launch of every kernel, from the two PMC passes, corrected as /opt/skills/guides/MI355X_MICROARCH.md
(HBM section) prescribes for gfx950: FETCH_SIZE (KiB) reports half of the bytes of a wide coalesced streaming read, so
    hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.
"""
import csv
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(.*$", "", name).strip()
    return name.replace("void ", "")


def counter_avg(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and "rm::" in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    cfg_name = sys.argv[2] if len(sys.argv) > 2 else "P"
    src = os.path.join(ROOT, "gpurun_out", tag) if cfg_name == "P" else os.path.join(ROOT, "gpurun_out", tag, cfg_name)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    sfx = "" if cfg_name == "P" else "_config_" + cfg_name
    names = {"bench": "bench_n1.json" if cfg_name == "P" else "bench%s.json" % sfx,
             "rocprof": "bench_n1_under_rocprof.json" if cfg_name == "P" else "bench%s_under_rocprof.json" % sfx,
             "stats": "kernel_stats_bench_n1.csv" if cfg_name == "P" else "kernel_stats%s.csv" % sfx,
             "pmc": "pmc_hbm_traffic_per_kernel.csv" if cfg_name == "P" else "pmc_hbm_traffic_per_kernel%s.csv" % sfx}
    for a, b in [("bench_n1.json", names["bench"]), ("bench_under_rocprof.json", names["rocprof"]),
                 ("stats/k_kernel_stats.csv", names["stats"])]:
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(dst, b))
    fetch = counter_avg(os.path.join(src, "pmc_fetch", "f_counter_collection.csv"), "FETCH_SIZE")
    write = counter_avg(os.path.join(src, "pmc_write", "w_counter_collection.csv"), "WRITE_SIZE")
    rows = []
    for k in sorted(fetch, key=lambda k: -fetch[k][0]):
        fk, n = fetch[k]
        wk = write.get(k, (0.0, 0))[0]
        rows.append((k, n, fk, wk, 2 * fk * 1024 + wk * 1024))
    with open(os.path.join(dst, names["pmc"]), "w") as f:
        f.write("kernel,launches_sampled,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg,hbm_bytes_per_launch_corrected\n")
        for r in rows:
            f.write("%s,%d,%.1f,%.1f,%.0f\n" % r)
    chain = [r for r in rows if r[0].startswith("rm::k_down_chain")]   # the frame-buffer kernel: the largest fetch among the chains
    if chain:
        k, n, fk, wk, b = chain[0]
        bench = json.load(open(os.path.join(dst, names["bench"])))
        cfg = bench["config"]
        key = "%s_%dx%dx%d" % (cfg["frame_buffer_dtype"], cfg["frames"], cfg["height"], cfg["width"])
        path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        table = json.load(open(path)) if os.path.exists(path) else {}
        # the stamp of the library the passes ran on (bench.py prints it; bench.py reports the figure only to a library with the same stamp)
        sha = bench.get("roofline", {}).get("kernel_source_sha")
        table[key] = {"kernel": k, "kernel_source_sha": sha, "bytes_per_launch": b, "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
                      "correction": "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE reports half of a wide "
                                    "coalesced stream, MI355X_MICROARCH.md HBM section); separate --pmc passes",
                      "source": "profiles/%s/%s" % (tag, names["pmc"])}
        json.dump(table, open(path, "w"), indent=1)
        print(key, "traffic %.3f GB per launch = %.3fx algorithmic" % (b / 1e9, b / bench["roofline"]["algorithmic_bytes"]))
    for r in rows[:16]:
        print("%-60s n=%d fetch %.0f KiB write %.0f KiB -> %.1f MB" % (r[0][:60], r[1], r[2], r[3], r[4] / 1e6))


if __name__ == "__main__":
    main()
