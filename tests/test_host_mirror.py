"""The C++ host-side mirror of the reference's IOThread / ThreadBlockingQueue / ReBuffer / VisualProcessor surface
(cubicsdr_amd/host/): compiled with g++ against libcsdr_hip.so and exercised by tests/cpp/test_host.cpp."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_host")


def _build():
    from cubicsdr_amd import build
    build.build(verbose=False)
    src = os.path.join(ROOT, "tests", "cpp", "test_host.cpp")
    deps = [src] + [os.path.join(ROOT, "cubicsdr_amd", "host", f) for f in os.listdir(os.path.join(ROOT, "cubicsdr_amd", "host"))]
    if os.path.exists(EXE) and all(os.path.getmtime(d) <= os.path.getmtime(EXE) for d in deps):
        return
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-pthread", src, "-o", EXE, "-L" + os.path.join(ROOT, "cubicsdr_amd"),
                    "-lcsdr_hip", "-ldl", "-Wl,-rpath," + os.path.join(ROOT, "cubicsdr_amd")], check=True)


def test_queue_rebuffer_iothread_visualprocessor_semantics():
    _build()
    r = subprocess.run([EXE, "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host test ok" in r.stdout


def test_fft_data_distributor_line_pacing_matches_oracle():
    """FFTDataDistributor (waterfall line pacing, K19): the C++ mirror against the statement-by-statement restatement of
    FFTDataDistributor.cpp:28-144 -- which lines go out (by stream position), the accumulator and the buffered count
    after every run(), across a retune, a line-size change, buffer overflow and a full consumer queue."""
    from oracle.fft_distributor import FFTDataDistributorRef
    _build()
    r = subprocess.run([EXE, "distrib"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    got = {}
    for ln in r.stdout.splitlines():
        f = ln.split()
        got.setdefault(f[1] if f[0] in "LS" else "E", []).append(f)
    ref = FFTDataDistributorRef(4096, 30)
    pos = 0
    want = {k: [] for k in "ABCD"}

    def feed(tag, freq, rate, n, blocks_per_run):
        nonlocal pos
        for _ in range(blocks_per_run):
            for first, size, fq, rt in ref.push(list(range(pos, pos + n)), freq, rate):
                want[tag].append(["L", tag, str(first), str(size), str(fq), str(rt)])
            pos += n
        want[tag].append(["S", tag, "%.12f" % ref.accum, str(len(ref.buf))])

    for _ in range(60):
        feed("A", 100000000, 2400000, 40000, 1)
    ref.lps = 400
    for _ in range(6):
        feed("B", 101000000, 2400000, 40000, 3)
    ref.fft_size = 16384
    for _ in range(8):
        feed("C", 101000000, 2400000, 40000, 1)
    ref.fft_size = 2048
    ref.lps = 1000
    for _ in range(5):
        feed("D", 101000000, 96000, 40000, 1)
    for tag in "ABCD":
        assert len(got[tag]) == len(want[tag]), tag
        for g, w in zip(got[tag], want[tag]):
            if g[0] == "S":
                assert g[3] == w[3] and abs(float(g[2]) - float(w[2])) < 1e-9, (tag, g, w)
            else:
                assert g == w, (tag, g, w)
    n_lines = {t: sum(1 for g in got[t] if g[0] == "L") for t in "ABCD"}
    assert 28 <= n_lines["A"] <= 31 and n_lines["B"] > 100 and n_lines["D"] > 0, n_lines
    assert got["E"][0][1] == "1"          # default queue capacity 1: the other lines of that run were dropped


def test_level_squelch_state_machine_matches_oracle():
    """DemodulatorThread's level / floor / ceil / squelch bookkeeping (DemodulatorThread.cpp:142-220): the C++ mirror
    (cubicsdr_amd/host/DemodLevel.h) against the statement-by-statement restatement in oracle/cubicsdr_chain.py over 400 blocks
    with quiet and loud stretches, empty blocks and the squelch switched on and moved: squelch decisions and the squelch-break
    flag exact, the float trackers to one float32 ulp."""
    import numpy as np
    from oracle.cubicsdr_chain import RefLevelSquelch
    _build()
    r = subprocess.run([EXE, "level"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    ref = RefLevelSquelch()
    n = 0
    flips = 0
    last = None
    for ln in r.stdout.splitlines():
        f = ln.split()
        have, accum, count, st, sq, sl = int(f[1]), float(f[2]), int(f[3]), float(f[4]), int(f[5]), float(f[6])
        squelched = ref.step(bool(have), accum, count, st, bool(sq), sl)
        assert squelched == bool(int(f[7])), ln
        assert ref.squelch_break == bool(int(f[11])), ln
        for got, want in ((float(f[8]), ref.level), (float(f[9]), ref.floor), (float(f[10]), ref.ceil)):
            assert abs(got - float(want)) <= float(np.spacing(np.float32(abs(float(want))))), (ln, got, want)
        flips += last is not None and squelched != last
        last = squelched
        n += 1
    assert n == 400 and flips >= 2            # the sequence really opens and closes the squelch


@pytest.mark.parametrize("san", ["thread", "address"])
def test_host_mirror_under_sanitizers(san):
    """SURVEY 5 (race detection): the C++ host mirror's threads / queues / buffer pools / visual processors (test_host "cpu" mode: every
    IOThread, ThreadBlockingQueue, ReBuffer and VisualProcessor scenario) built with ThreadSanitizer and with AddressSanitizer and run to a clean exit"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    lib = build_emu.build("")                                  # the host-executing library only resolves the symbols: "cpu" mode never calls into it
    exe = os.path.join(os.path.dirname(lib), "test_host_" + san)
    src = os.path.join(ROOT, "tests", "cpp", "test_host.cpp")
    deps = [src] + [os.path.join(ROOT, "cubicsdr_amd", "host", f) for f in os.listdir(os.path.join(ROOT, "cubicsdr_amd", "host"))]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=" + san, "-fno-omit-frame-pointer", src, "-o", exe, "-L" + os.path.dirname(lib),
                        "-l:" + os.path.basename(lib), "-ldl", "-Wl,-rpath," + os.path.dirname(lib)], check=True)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=0 exitcode=66")
    r = subprocess.run([exe, "cpu"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "host test ok" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr


@pytest.mark.gpu
def test_threaded_pipeline_on_gpu():
    """SDRThreadIQData blocks -> SDRPostThread (HIP) -> NBFM audio queue + spectrum queue, through real threads/queues"""
    _build()
    r = subprocess.run([EXE, "gpu", ROOT], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gpu host test ok" in r.stdout
