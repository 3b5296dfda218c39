"""TEST INFRASTRUCTURE ONLY: build the host-thread emulation of the HIP library (see tests/emu/hip/hip_runtime.h).

g++ compiles the UNMODIFIED product sources (cubicsdr_amd/csrc/csdr_*.hip + kernel headers) against the shim
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
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if os.path.isfile(os.path.join(CSRC, f))] + [os.path.join(HERE, "hip", "hip_runtime.h"),
                                                                os.path.join(HERE, "hip_emu_runtime.cpp"),
                                                                os.path.join(ROOT, "include", "csdr_hip.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    san = {"": ["-O2"], "asan": ["-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer"],
           "tsan": ["-O1", "-g", "-fsanitize=thread"]}[flavor]
    cmd = ["g++", "-std=c++20", "-shared", "-fPIC", "-pthread", "-w", "-ffp-contract=off", "-I", HERE] + san + \
          (["-DCSDR_LAB"] if lab else [])
    units = sorted(f for f in os.listdir(CSRC) if f.startswith("csdr_") and f.endswith(".hip"))
    objs = []

    def compile_one(u):
        obj = os.path.join(OUT_DIR, "%s%s.o" % (u[:-4], "_" + os.path.basename(out)[:-3]))
        c = cmd + ["-c", "-x", "c++", os.path.join(CSRC, u), "-o", obj]
        if verbose:
            print("[emu build]", " ".join(c), flush=True)
        subprocess.run(c, check=True)
        return obj
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(len(units)) as ex:
        objs = list(ex.map(compile_one, units))
    link = cmd + objs + [os.path.join(HERE, "hip_emu_runtime.cpp"), "-o", out]
    if verbose:
        print("[emu build]", " ".join(link), flush=True)
    subprocess.run(link, check=True)
    for o in objs:
        os.remove(o)
    return out


if __name__ == "__main__":
    build(sys.argv[1] if len(sys.argv) > 1 else "", force=True, verbose=True)
