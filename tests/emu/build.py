"""tests/emu/build.py -- TEST INFRASTRUCTURE ONLY.

Compiles the PRODUCT sources respmon_amd/csrc/*.hip with g++ against the host emulation of
the HIP subset in tests/emu/include (which shadows <hip/hip_runtime.h> by include path), into
tests/emu/_build/librespmon_emu.so.  Used only by tests/test_emu_*.py to exercise the kernels'
index / border / ordering logic and the host orchestration on a machine without a GPU.
The product package never loads this library; the product library is built by hipcc for gfx950
(respmon_amd/csrc/Makefile)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "respmon_amd", "csrc")
OUT = os.path.join(HERE, "_build", "librespmon_emu.so")


def build(force=False):
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    srcs += [os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "respmon_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(s) <= os.path.getmtime(OUT) for s in srcs):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    import hashlib
    stamp_srcs = ["rm_down_chain.h", "rm_down_chain_u8.h", "rm_down_launch.h", "rm_down.hip", "rm_down_f64.hip", "rm_down_generic.hip",
                  "rm_down_narrow.hip", "rm_down_bgr.hip"]     # csrc/Makefile STAMP_SRCS
    sha = hashlib.sha256(b"".join(open(os.path.join(CSRC, f), "rb").read() for f in stamp_srcs)).hexdigest()[:16]
    units = ["rm_unity.hip", "rm_contour.cpp"]   # every product translation unit, as one unit (respmon_amd/csrc/rm_unity.hip)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-pthread",
           "-I", os.path.join(HERE, "include"), "-DRM_HIPEMU_DEFINE_TLS", '-DRM_FRAME_KERNEL_SRC_SHA="%s"' % sha, "-Wno-unused-function", "-Wno-attributes", "-Wno-psabi", "-o", OUT]
    for u in units:
        cmd += ["-x", "c++", os.path.join(CSRC, u)]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
