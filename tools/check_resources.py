"""Parse hipcc's -Rpass-analysis=kernel-resource-usage remarks (respmon_amd/csrc/build_resources.txt, written by the Makefile) into a
table and FAIL when a frame-buffer kernel of the shipped library uses scratch (spilled registers): those kernels stream the
[T,H,W] buffer and every spilled VGPR is HBM traffic on top of it (1.60x the algorithmic bytes on the uint8 chain).

    python tools/check_resources.py [remarks file] [--all]      exit code 1 on a violation
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = os.path.join(ROOT, "respmon_amd", "csrc", "build_resources.txt")
# kernels that must hold their state in registers: the hot (VB = false) instantiations of the float64 chain and every narrow chain
MUST_NOT_SPILL = re.compile(r"^(void )?rm::k_down_chain<[^>]*, false>|^(void )?rm::k_down_chain_u8<|^(void )?rm::k_down_chain_narrow<")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.strip().split("\n")
    except Exception:
        return names


def parse(path):
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        txt = m.group(1).strip()
        if txt.startswith("Function Name:"):
            cur = {"name": txt.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in txt:
            k, v = txt.rsplit(":", 1)
            cur[k.strip()] = v.strip()
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["demangled"] = n.split("(")[0]
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0] if args else DEFAULT
    rows = parse(path)
    if not rows:
        print("check_resources: no kernel-resource-usage remarks in %s" % path)
        return 1
    bad = []
    show_all = "--all" in sys.argv
    print("%-72s %5s %5s %8s %6s %4s" % ("kernel", "VGPR", "AGPR", "scratch", "spill", "occ"))
    for r in rows:
        scratch = int(r.get("ScratchSize [bytes/lane]", "0"))
        must = bool(MUST_NOT_SPILL.search(r["demangled"]))
        if must and scratch > 0:
            bad.append(r)
        if show_all or must or scratch > 0:
            print("%-72s %5s %5s %8d %6s %4s%s" % (r["demangled"][:72], r.get("VGPRs", "?"), r.get("AGPRs", "?"), scratch, r.get("VGPRs Spill", "?"),
                                                 r.get("Occupancy [waves/SIMD]", "?"), "   <-- must not spill" if must and scratch else ""))
    if bad:
        print("check_resources: %d frame-buffer kernel(s) use scratch" % len(bad))
        return 1
    print("check_resources: ok (%d kernels, frame-buffer kernels hold their state in registers)" % len(rows))
    return 0


if __name__ == "__main__":
    sys.exit(main())
