"""The N > 1 path over RCCL on real GPUs: two ranks, one GPU each -- through the C ABI's communicator (csdr_comm: no torch.distributed
process group at all, the id travels over a TCP store) and through torch.distributed (backend "nccl" = RCCL over xGMI); plus the C++
example tests/cpp/comm_ranks.cpp (one rank where one GPU is visible: the same RCCL calls).  Skipped where fewer than two GPUs
are visible (RCCL refuses two ranks on one device); the same stream classes run on CPU over gloo in tests/test_parallel_gloo.py and as
virtual ranks on one GPU in tests/test_gpu_parity.py.  What this adds: the collectives, the library's kernels and torch's allocator
really are ordered by the dedicated boundary stream (cubicsdr_amd/parallel.py: _Boundary) -- every sharded demodulator's audio must equal
the unsharded path's bit for bit over several back-to-back batches with no host synchronisation between them."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, mode, q, transport="torch"):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    cid = None
    if transport == "abi":
        from cubicsdr_amd.parallel import exchange_id
        cid = exchange_id(rank, world)                        # no process group: the collectives are the library's own
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        from cubicsdr_amd.engine import Context, DemodBank, SDRPost
        from cubicsdr_amd.parallel import ShardedStream, SlabStream
        from tests.util import demod_frequencies, synth_iq
        fs, M, block, nd, nb, nbat, center = 2400000, 4, 40000, 6, 4, 5, 400000000
        freqs = demod_frequencies(center, fs, nd); freqs[0] = center + 1500
        demods = [("NBFM" if i % 2 == 0 else "AM", 12500 if i % 2 == 0 else 6000, f) for i, f in enumerate(freqs)]
        x = synth_iq(nbat * nb * block, fs, center, [(k, f) for k, _, f in demods], seed=97)
        dev = torch.device("cuda", rank)
        xf = torch.from_numpy(x.view(np.float32).reshape(-1, 2).copy()).to(dev)
        # the unsharded answer, every batch, on this rank's GPU
        ctx = Context(rank)
        post = SDRPost(ctx, fs, M, block, max_blocks=nb); bank = DemodBank(ctx, nd, max_blocks=nb)
        for i, (k, b, f) in enumerate(demods):
            bank.configure(i, post, k, b, f)
        want = []
        for t in range(nbat):
            post.execute(xf[t * nb * block:(t + 1) * nb * block], nb, block, center)
            bank.execute(post)
            want.append({i: bank.audio(i) for i in range(nd)})
        if mode == "broadcast":
            st = ShardedStream(rank, rank, world, fs, M, block, demods, center, nb, comm_id=cid)
            batch = torch.empty(nb * block, 2, dtype=torch.float32, device=dev)
        else:
            st = SlabStream(rank, rank, world, fs, M, block, demods, center, nb, comm_id=cid)
        # batches back to back: only the LAST one is inspected after a synchronise, the earlier ones must have been ordered by the streams
        got_last = None
        for t in range(nbat):
            src = xf[t * nb * block:(t + 1) * nb * block]
            if mode == "broadcast":
                if rank == 0:
                    st.release()                              # the kernels of the previous batch have read `batch`
                    batch.copy_(src)
                st.step(batch, nb, src=0)
            else:
                st.step(st.scatter(src if rank == 0 else None, nb, src=0), nb, overlap=(mode == "slab_overlapped"))
        if mode == "slab_overlapped":
            st.flush()                                    # the last batch's rows are still between the two halves of their exchange
        st.synchronize()
        ok = all(np.array_equal(st.audio(i), want[-1][i]) for i in st.plan.demods)
        q.put((rank, ok, list(st.plan.demods)))
        st.close(); bank.close(); post.close(); ctx.close()
    finally:
        if transport != "abi":
            dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["abi", "torch"])
@pytest.mark.parametrize("mode", ["broadcast", "slab", "slab_overlapped"])
def test_sharded_streams_over_rccl_equal_unsharded(mode, transport):
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    if mode == "slab_overlapped" and transport == "torch":
        pytest.skip("the two-halves exchange on its own stream is the C ABI communicator's (torch.distributed collectives are the one-call order)")
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    owned = []
    for rank, ok, mine in res:
        assert ok, (mode, rank)
        owned += mine
    assert sorted(owned) == list(range(6))


def test_cpp_ranks_through_the_c_abi():
    """tests/cpp/comm_ranks.cpp: one process per GPU in C++ against include/csdr_hip.h alone -- csdr_comm_broadcast / _scatter / _all_to_all /
    _max on known patterns, then both sharded variants (csdr_post_exchange_rows) against an unsharded channelizer.  Two ranks where two
    GPUs are visible, else one (the same RCCL entry points with a one-rank communicator)."""
    import subprocess
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from cubicsdr_amd import build
    build.build(verbose=False)
    exe = os.path.join(ROOT, "tests", "cpp", "comm_ranks")
    src = exe + ".cpp"
    if not os.path.exists(exe) or os.path.getmtime(src) > os.path.getmtime(exe):
        subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", src, "-o", exe, "-L" + os.path.join(ROOT, "cubicsdr_amd"), "-lcsdr_hip", "-ldl",
                        "-Wl,-rpath," + os.path.join(ROOT, "cubicsdr_amd")], check=True)
    world = min(2, torch.cuda.device_count())
    r = subprocess.run([exe, str(world)], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("sharded rows: ok") == world
