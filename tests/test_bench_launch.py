"""bench.py's own launcher and its reporting helpers.  `python bench.py --gpus N` (the driver's command shape) must produce the N-rank line
unaided: it re-runs itself under torch.distributed.run, one rank per GPU; rank 0 prints ONE JSON line with n_gpus = N, the replica (weak)
value of the default configuration and a `strong` object (ONE C4 stream over all ranks, time slabs + all-to-all).  On a one-GPU box the same
control flow runs as a dry run: both ranks on the one device, collectives over gloo (staged through the host)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spread_and_sensors_without_a_gpu():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.spread([3.0, 1.0, 2.0]) == {"n": 3, "min": 1.0, "median": 2.0, "max": 3.0}
    assert bench.spread([]) == {}
    s = bench.Sensors(0)               # no HIP device here: no files, start / stop are no-ops
    s.start()
    assert s.stop().get("samples", 0) == 0 or s.files


def test_only_the_json_line_reaches_stdout():
    """libraries under bench.py write to file descriptor 1 (RCCL's version banner, gloo's connection notes): after claim_stdout() such writes land
    on stderr and the stream it returns is the only way to the caller's stdout"""
    code = ("import os, sys; sys.path.insert(0, %r); import bench\n"
            "out = bench.claim_stdout()\n"
            "os.write(1, b'banner from a library\\n'); print('a stray print')\n"
            "out.write('{\"ok\": 1}\\n'); out.flush()\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"ok": 1}\n', r.stdout
    assert "banner from a library" in r.stderr and "a stray print" in r.stderr


@pytest.mark.gpu
def test_bench_gpus_2_launches_its_own_ranks_dry_run():
    import torch
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env["CSDR_DIST_BACKEND"] = "gloo"      # two ranks on the one GPU: RCCL refuses that, gloo carries the collectives
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--blocks", "8", "--batches", "2",
           "--cpu-seconds", "0", "--no-latency", "--ring", "noise"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["step_ms"]["n"] == 1
    st = out["strong"]
    assert st.get("value"), st
    assert st["scaling"] == "strong" and st["n_gpus"] == 2
    if torch.cuda.device_count() >= 2:
        assert st["rccl_ranks"] == 2
