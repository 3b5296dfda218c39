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
                    "-lcsdr_hip", "-Wl,-rpath," + os.path.join(ROOT, "cubicsdr_amd")], check=True)


def test_queue_rebuffer_iothread_visualprocessor_semantics():
    _build()
    r = subprocess.run([EXE, "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host test ok" in r.stdout


@pytest.mark.gpu
def test_threaded_pipeline_on_gpu():
    """SDRThreadIQData blocks -> SDRPostThread (HIP) -> NBFM audio queue + spectrum queue, through real threads/queues"""
    _build()
    r = subprocess.run([EXE, "gpu"], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gpu host test ok" in r.stdout
