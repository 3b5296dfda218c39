#!/usr/bin/env python3
"""Static instruction statistics of the gfx950 kernels (CPU-side: hipcc --cuda-device-only -S, no GPU needed).

  python profiles/isa_stats.py csdr_spec [kernel-name-substring ...]      (unit = a file of cubicsdr_amd/csrc without .hip, or a path to any .hip)

Per kernel: VGPRs / SGPRs / scratch bytes, and the instruction mix of the whole code object text: packed fp32 (v_pk_*), other vector ALU,
64-bit integer / compare / select / move instructions (address and guard overhead), scalar ALU, LDS, global memory, waits, barriers.  The
kernels of the spectrum chain are bound by vector-ALU issue: these counts are what a change to their instruction stream is
judged by before it goes to the GPU.  (Counts are static -- every instruction once, loops not weighted.)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True)
        return r.stdout.split("\n")
    except Exception:
        return names


def classify(op):
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "v_mov"
    if op.startswith("v_cndmask") or op.startswith("v_cmp"):
        return "v_cmp/sel"
    if re.match(r"v_(lshl_add_u64|mad_u64|mad_i64|lshlrev_b64|add_co|addc_co|add_u64|ashrrev_i64|lshrrev_b64)", op):
        return "v_addr64"
    if op.startswith("v_") and "f64" in op:
        return "v_f64"
    if op.startswith("v_"):
        return "v_other"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "s_load"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_"):
        return "s_alu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
        return "vm_load"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store"):
        return "vm_store"
    if op.startswith("scratch_"):
        return "scratch"
    return "other"


def main():
    unit = sys.argv[1]
    pats = sys.argv[2:]
    src = unit if unit.endswith(".hip") else os.path.join(ROOT, "cubicsdr_amd", "csrc", unit + ".hip")
    flags = os.environ.get("ISA_FLAGS", "").split()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "u.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out, "-Wno-unused-command-line-argument"] + flags, check=True)
        text = open(out).read()
    # kernel bodies: from "<name>:" (after .type <name>,@function) to .Lfunc_end
    bodies = {}
    for m in re.finditer(r"^(\w+):\s*;\s*@\1\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    meta = {}
    for m in re.finditer(r"\.name:\s+(\w+)\n(.*?)(?=\n\s+- \.|\namdhsa\.target)", text, re.S):
        blk = m.group(0)
        g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
        meta[m.group(1)] = (g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("agpr_count"))
    names = sorted(bodies)
    dem = dict(zip(names, demangle(names)))
    cols = ["v_pk", "v_other", "v_f64", "v_mov", "v_cmp/sel", "v_addr64", "s_alu", "s_load", "s_waitcnt", "s_barrier", "s_nop", "lds", "vm_load", "vm_store", "scratch", "other"]
    print("%-58s %5s %5s %5s | %6s | " % ("kernel", "vgpr", "sgpr", "scr", "total") + " ".join("%9s" % c for c in cols))
    for n in names:
        short = re.sub(r"\(.*", "", dem[n]).replace("csdr::", "").replace("void ", "")
        if pats and not any(p in short for p in pats):
            continue
        cnt = collections.Counter()
        for line in bodies[n].split("\n"):
            line = line.strip()
            if not line or line.startswith(";") or line.startswith(".") or line.endswith(":"):
                continue
            op = line.split()[0]
            if not re.match(r"^[a-z_0-9]+$", op):
                continue
            cnt[classify(op)] += 1
        v, s, p, _ = meta.get(n, ("?", "?", "?", "?"))
        print("%-58s %5s %5s %5s | %6d | " % (short[:58], v, s, p, sum(cnt.values())) + " ".join("%9d" % cnt[c] for c in cols))


if __name__ == "__main__":
    main()
