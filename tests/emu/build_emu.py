"""TEST INFRASTRUCTURE ONLY: build the host-thread emulation of the HIP library (see tests/emu/hip/hip_runtime.h).

g++ compiles the UNMODIFIED product sources (cubicsdr_amd/csrc/csdr_api.hip + kernel headers) against the shim
<hip/hip_runtime.h> in this directory.  Output: tests/emu/_build/libcsdr_emu[_asan|_tsan].so -- never placed inside the
cubicsdr_amd package, never loaded by it.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "cubicsdr_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")


def lib_path(flavor=""):
    return os.path.join(OUT_DIR, "libcsdr_emu%s.so" % (("_" + flavor) if flavor else ""))


def build(flavor="", force=False, verbose=False):
    """flavor: "" (plain -O2), "asan", "tsan" """
    os.makedirs(OUT_DIR, exist_ok=True)
    lab = os.environ.get("CSDR_BUILD_LAB") == "1"       # the measurement switches of common.hpp lab_int() compiled in
    out = lib_path(flavor + ("lab" if lab else ""))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "hip", "hip_runtime.h"),
                                                                os.path.join(HERE, "hip_emu_runtime.cpp"),
                                                                os.path.join(ROOT, "include", "csdr_hip.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    san = {"": ["-O2"], "asan": ["-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer"],
           "tsan": ["-O1", "-g", "-fsanitize=thread"]}[flavor]
    cmd = ["g++", "-std=c++20", "-shared", "-fPIC", "-pthread", "-w", "-ffp-contract=off", "-I", HERE] + san + \
          (["-DCSDR_LAB"] if lab else []) + ["-x", "c++", os.path.join(CSRC, "csdr_api.hip"), os.path.join(HERE, "hip_emu_runtime.cpp"), "-o", out]
    if verbose:
        print("[emu build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build(sys.argv[1] if len(sys.argv) > 1 else "", force=True, verbose=True)
