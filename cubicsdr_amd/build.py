"""Build the gfx950 C-ABI library in-tree: cubicsdr_amd/libcsdr_hip.so (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# one translation unit per object of include/csdr_hip.h (an edit rebuilds its own unit; the units compile in parallel)
UNITS = ["csdr_ctx", "csdr_post", "csdr_bank", "csdr_spec", "csdr_io", "csdr_comm"]
SRCS = [os.path.join(HERE, "csrc", u + ".hip") for u in UNITS]
OBJ_DIR = os.path.join(HERE, "csrc", "_obj")
OUT = os.path.join(HERE, "libcsdr_hip.so")


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


def _unit_deps(src):
    """the unit itself plus every header of csrc/ it includes, transitively"""
    import re
    seen, todo = set(), [src]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(f):
            continue
        seen.add(f)
        for inc in re.findall(r'^#include "([^"]+)"', open(f).read(), re.M):
            todo.append(os.path.normpath(os.path.join(os.path.dirname(f), inc)))
    return seen


def build(force=False, verbose=True):
    build_design(force, verbose)
    lab = os.environ.get("CSDR_BUILD_LAB") == "1"      # measurement build: the A/B switches of common.hpp lab_int() read the environment
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + (["-DCSDR_LAB"] if lab else [])
    stamp = os.path.join(OBJ_DIR, "flavor")
    flavor = "lab" if lab else "ship"
    if not os.path.exists(stamp) or open(stamp).read() != flavor:
        force = True
    jobs = []
    for src in SRCS:
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in _unit_deps(src)):
            jobs.append((src, obj))
    objs = [os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o") for src in SRCS]
    if not jobs and os.path.exists(OUT) and all(os.path.getmtime(o) <= os.path.getmtime(OUT) for o in objs):
        return OUT
    cc = hipcc()

    def compile_one(job):
        cmd = [cc] + flags + ["-c", job[0], "-o", job[1]]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(compile_one, jobs))
    open(stamp, "w").write(flavor)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", OUT]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
