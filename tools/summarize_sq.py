"""gpurun_out/pmc_tail (tools/pmc_tail.sh) -> profiles/<tag>/sq_counters_per_kernel.txt: where each kernel's waves spend
their cycles (SQ_WAIT_ANY = parked on s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stalls, SQ_ACTIVE_INST_ANY = issuing;
quad-cycle units, MI355X_MICROARCH.md PMC table) and what they issue.   python tools/summarize_sq.py r01"""
import csv
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "rm::" not in r["Kernel_Name"]:
            continue
        k = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("rm::", "")[:30]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "pmc_tail")
    a = load(os.path.join(src, "a", "a_counter_collection.csv"))
    b = load(os.path.join(src, "b", "b_counter_collection.csv"))
    avg = lambda d, n: (sum(d[n]) / len(d[n]) if d[n] else 0.0)
    lines = ["per launch averages, bench.py workload (256x1080x1920 f64); rocprofv3 --pmc, two passes (tools/pmc_tail.sh)", "",
             "%-30s %8s %13s %7s %8s %8s %9s" % ("kernel", "waves", "wave_cycles/4", "wait%", "istall%", "active%", "lds_stall%")]
    for k in a:
        wc = avg(a[k], "SQ_WAVE_CYCLES") or 1.0
        lines.append("%-30s %8.0f %13.0f %7.1f %8.1f %8.1f %9.1f" % (
            k, avg(a[k], "SQ_WAVES"), wc, 100 * avg(a[k], "SQ_WAIT_ANY") / wc, 100 * avg(a[k], "SQ_WAIT_INST_ANY") / wc,
            100 * avg(a[k], "SQ_ACTIVE_INST_ANY") / wc, 100 * avg(a[k], "SQ_WAIT_INST_LDS") / wc))
    lines += ["", "%-30s %10s %10s %9s %9s %9s %10s" % ("kernel", "valu", "salu", "lds", "vmem_rd", "vmem_wr", "bank_confl")]
    for k in b:
        lines.append("%-30s %10.0f %10.0f %9.0f %9.0f %9.0f %10.0f" % (
            k, avg(b[k], "SQ_INSTS_VALU"), avg(b[k], "SQ_INSTS_SALU"), avg(b[k], "SQ_INSTS_LDS"), avg(b[k], "SQ_INSTS_VMEM_RD"),
            avg(b[k], "SQ_INSTS_VMEM_WR"), avg(b[k], "SQ_LDS_BANK_CONFLICT")))
    out = os.path.join(ROOT, "profiles", tag, "sq_counters_per_kernel.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
