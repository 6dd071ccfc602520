"""gpurun_out/pmc_narrow (tools/pmc_narrow.sh) -> profiles/<tag>/narrow_chain_sq_counters.txt: what the register-resident chain kernels
issue per input pixel and where their waves spend their cycles, against the instruction-issue roofline of bench_micro/valu_rate.hip
(~4.8 cycles per wave64 fp64 / convert / DPP instruction on a SIMD).      python tools/summarize_narrow.py r03"""
import csv
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIX = {"u8": (256 * 1080 * 1920, "k_down_chain_u8<4, unsigned char>"), "f32": (256 * 1080 * 1920, "k_down_chain_narrow<4, float>"),
       "f16": (512 * 2160 * 3840, "k_down_chain_u8<2, __half>")}
CYC_PER_INSTR, SIMDS, GHZ = 4.8, 1024, 2.4


def load(path, kernel):
    acc = defaultdict(list)
    if not os.path.exists(path):
        return acc
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    src = os.path.join(ROOT, "gpurun_out", "pmc_narrow")
    lines = ["register-resident chain kernels (rm_down_chain_u8.h): SQ counters per launch, rocprofv3 --pmc (tools/pmc_narrow.sh)",
             "issue roofline: VALU instructions x %.1f cycles / (%d SIMDs x %.1f GHz)  [bench_micro/valu_rate.hip: profiles/%s/valu_issue_rates.txt]" % (CYC_PER_INSTR, SIMDS, GHZ, tag), ""]
    for dt, (npx, kern) in PIX.items():
        a = load(os.path.join(src, dt, "a_counter_collection.csv"), kern)
        b = load(os.path.join(src, dt + "_b", "b_counter_collection.csv"), kern)
        if not a:
            continue
        avg = lambda d, n: (sum(d[n]) / len(d[n]) if d.get(n) else 0.0)
        valu, salu, waves = avg(a, "SQ_INSTS_VALU"), avg(a, "SQ_INSTS_SALU"), avg(a, "SQ_WAVES")
        wc = avg(a, "SQ_WAVE_CYCLES") or 1.0
        issue_ms = valu * CYC_PER_INSTR / (SIMDS * GHZ * 1e9) * 1e3
        lines.append("%s  %s" % (dt, kern))
        lines.append("  waves %.0f   VALU wave-instructions %.3e = %.2f per input pixel and lane   SALU %.3e   LDS %.3e   VMEM rd %.3e" % (
            waves, valu, valu * 64 / npx, salu, avg(b, "SQ_INSTS_LDS"), avg(b, "SQ_INSTS_VMEM_RD")))
        lines.append("  issue-bound time of the VALU stream: %.3f ms" % issue_ms)
        lines.append("  wave cycles: waiting (s_waitcnt) %.0f %%, issue stalls %.0f %%, issuing %.0f %%" % (
            100 * avg(a, "SQ_WAIT_ANY") / wc, 100 * avg(a, "SQ_WAIT_INST_ANY") / wc, 100 * avg(a, "SQ_ACTIVE_INST_ANY") / wc))
        lines.append("")
    out = os.path.join(ROOT, "profiles", tag, "narrow_chain_sq_counters.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
