"""Build the gfx950 C-ABI library in-tree: cubicsdr_amd/libcsdr_hip.so (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "csdr_api.hip")
OUT = os.path.join(HERE, "libcsdr_hip.so")
DEPS = [os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))] + [
    os.path.join(HERE, "..", "include", "csdr_hip.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


DESIGN_SRC = os.path.join(HERE, "csrc", "design_capi.cpp")
DESIGN_OUT = os.path.join(HERE, "libcsdr_design.so")


def build_design(force=False, verbose=True):
    """host-only helper library (g++): the cold-path filter design, testable without a GPU"""
    deps = [DESIGN_SRC, os.path.join(HERE, "csrc", "design.hpp")]
    if not force and os.path.exists(DESIGN_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(DESIGN_OUT) for d in deps):
        return DESIGN_OUT
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", DESIGN_SRC, "-o", DESIGN_OUT]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return DESIGN_OUT


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force=False, verbose=True):
    build_design(force, verbose)
    if not force and not needs_build():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall",
           "-Wno-unused-function", SRC, "-o", OUT]
    if os.environ.get("CSDR_BUILD_LAB") == "1":      # measurement build: the A/B switches of common.hpp lab_int() read the environment
        cmd.insert(1, "-DCSDR_LAB")
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
