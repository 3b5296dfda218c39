"""N > 1 path on CPU: world size 2, gloo.  Covers the demodulator sharding map, the per-rank channel sets, the IQ-batch
broadcast and the max-/sum-over-ranks reductions bench.py relies on."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from cubicsdr_amd import parallel as P
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_demods, M = 64, 20
        rng = np.random.default_rng(7)
        channels = [int(c) for c in rng.integers(0, M, n_demods)]
        pl = P.plan(n_demods, channels, world, rank)
        # every rank publishes its shard: together they must partition the demods, with disjoint channel sets
        gathered = [None] * world
        dist.all_gather_object(gathered, (pl.demods, pl.active_channels))
        batch = torch.zeros(4096, 2)
        if rank == 0:
            batch = torch.arange(8192, dtype=torch.float32).reshape(4096, 2)
        P.broadcast_iq(batch, src=0)
        tmax = P.max_over_ranks(1.0 + rank)
        tot = P.sum_over_ranks(len(pl.demods))
        q.put((rank, gathered, float(batch.sum()), tmax, tot, channels))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_broadcast():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, gathered, bsum, tmax, tot, channels in res:
        all_demods = sorted(i for shard, _ in gathered for i in shard)
        assert all_demods == list(range(64))                                   # a partition
        chsets = [set(c) for _, c in gathered]
        assert not (chsets[0] & chsets[1])                                      # one channel -> one rank
        for shard, chs in gathered:
            assert sorted({channels[i] for i in shard}) == chs
        assert abs(len(gathered[0][0]) - len(gathered[1][0])) <= 8              # balanced
        assert bsum == float(sum(range(8192)))                                  # broadcast delivered rank 0's batch
        assert tmax == 2.0 and tot == 64


def test_contiguous_shards_cover_everything():
    from cubicsdr_amd.parallel import shard_demods
    for n, w in [(64, 8), (256, 8), (7, 4), (3, 8), (1024, 2)]:
        got = [i for r in range(w) for i in shard_demods(n, w, r)]
        assert got == list(range(n))
        sizes = [len(shard_demods(n, w, r)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1


def test_channel_routing_twin_and_stream_plan():
    """the host twin of getChannelAt / updateChannels (SDRPostThread.cpp:116-139) that ShardedStream plans with: every channel
    centre routes to itself, the upper band edge to entry M (row M / 2), ties go to the first entry; and the plans of all ranks
    partition the demodulators of BASELINE config 4 (1024 channels, one demodulator each) with disjoint channel sets."""
    from cubicsdr_amd import parallel as P
    fs, M, center = 100000000, 1024, 400000000
    cc = P.channel_centers(center, fs, M)
    assert len(cc) == M + 1 and cc[0] == center and cc[M] == center + fs // 2 and cc[M // 2] == center - fs // 2
    for ch in (0, 1, 17, 511, 512, 513, 1023):
        assert P.channel_at(cc[ch], center, fs, M) == ch
        assert P.channel_at(cc[ch] + 1000, center, fs, M) == ch
    assert P.channel_at(center + fs // 2, center, fs, M) == M and P.data_channel(M, M) == M // 2
    assert P.channel_at(center + 3 * fs, center, fs, M) == -1
    assert P.channel_at(123, 456, 2400000, 1) == 0
    channels = [P.data_channel(P.channel_at(cc[ch] + 3700, center, fs, M), M) for ch in range(M)]
    for world in (2, 4, 8):
        plans = [P.plan(M, channels, world, r) for r in range(world)]
        assert sorted(i for p in plans for i in p.demods) == list(range(M))
        seen = set()
        for p in plans:
            assert not (seen & set(p.active_channels))
            seen |= set(p.active_channels)
            assert abs(len(p.demods) - M // world) <= 1


def _slab_worker(rank, world, port, q, overlap=False, nbat=2):
    """one rank of parallel.SlabStream over gloo, its kernels run by the host-thread emulation of tests/emu (numpy buffers)"""
    import ctypes as C
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import build_emu
        import cubicsdr_amd.hip as H
        lib = C.CDLL(build_emu.build(""))
        for name, (res, args) in H.ABI.items():
            fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
        H._lib = lib
        from cubicsdr_amd.engine import Context, DemodBank, SDRPost
        from cubicsdr_amd.parallel import SlabStream
        from tests.util import demod_frequencies, synth_iq
        fs, M, block, nd, nb, center = 480000, 8, 8000, 6, 4, 400000000
        freqs = demod_frequencies(center, fs, nd); freqs[0] = center + 1500
        demods = [("NBFM" if i % 2 == 0 else "AM", 12500 if i % 2 == 0 else 6000, f) for i, f in enumerate(freqs)]
        x = synth_iq(nbat * nb * block, fs, center, [(k, f) for k, _, f in demods], seed=97)
        stream = SlabStream(0, rank, world, fs, M, block, demods, center, nb, use_torch=False)
        # the unsharded answer for this rank's demodulators, computed locally
        ctx = Context(0)
        post = SDRPost(ctx, fs, M, block, max_blocks=nb); bank = DemodBank(ctx, nd, max_blocks=nb)
        for i, (k, b, f) in enumerate(demods):
            bank.configure(i, post, k, b, f)
        ok, n_cmp = True, 0
        xf = x.view(np.float32).reshape(-1, 2)
        want = []                                                  # the unsharded audio / counts of this rank's demodulators, batch by batch

        def compare(t):
            nonlocal ok, n_cmp
            for i in stream.plan.demods:
                ok = ok and np.array_equal(stream.audio(i), want[t][i][0])
                ok = ok and [(r.n_iq, r.n_audio, r.nco_theta) for r in stream.results(i)] == want[t][i][1]
                n_cmp += 1
        for t in range(nbat):
            post.execute(x[t * nb * block:(t + 1) * nb * block], nb, block, center)
            bank.execute(post)
            want.append({i: (bank.audio(i), [(r.n_iq, r.n_audio, r.nco_theta) for r in bank.results(i)]) for i in stream.plan.demods})
            window = stream.scatter(xf[t * nb * block:(t + 1) * nb * block] if rank == 0 else None, nb, src=0)
            stream.step(window, nb, overlap=overlap)
            if not overlap:
                compare(t)
            elif t > 0:
                compare(t - 1)                                     # the pipeline is one batch deep: batch t - 1 is complete after step t
        if overlap:
            stream.flush()
            compare(nbat - 1)
        stream.close(); bank.close(); post.close(); ctx.close()
        q.put((rank, ok, n_cmp, stream.plan.demods))
    finally:
        dist.destroy_process_group()


def test_two_rank_time_slab_stream_over_gloo():
    """SURVEY 8e option 2 end to end on two processes: scatter of [history | slab] windows, per-rank channelizer over its slab, the
    all-to-all of channel rows over gloo, assembly, demodulators -- each rank's audio equals the unsharded path's bit for bit (the HIP
    kernels run through the CPU emulation; the GPU twin of this test is tests/test_gpu_parity.py::test_time_slab_sharding_equals_unsharded)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_slab_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    owned = []
    for rank, ok, n_cmp, mine in res:
        assert ok and n_cmp == 2 * len(mine), (rank, ok, n_cmp)
        owned += mine
    assert sorted(owned) == list(range(6))


def test_two_rank_time_slab_stream_overlapped_over_gloo():
    """the same stream with step(overlap=True) -- batch i + 1 channelized and its rows exchanged BEFORE batch i is imported and demodulated
    (the order the C ABI's begin / finish pair runs on a GPU node) -- over five back-to-back batches: every batch's audio still equals the
    unsharded path's bit for bit."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_slab_worker, args=(r, world, port, q, True, 5)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, ok, n_cmp, mine in res:
        assert ok and n_cmp == 5 * len(mine), (rank, ok, n_cmp)


def test_strong_scaling_plan_picks_by_the_link_model():
    """parallel.strong_scaling_plan: one C4 batch (32 blocks = 427 MB of input, 0.777 ms of kernels on one GPU).  Two GPUs: the one link binds
    every sharded variant, one GPU alone is fastest -- unless the demodulators do not fit one GPU (infinite one-GPU time), then the time slabs;
    four and eight GPUs: time slabs, and only with distributed ingest does eight reach the north star's 6 x (rank-0 ingest scatters 7 / 8 of
    the input over rank 0's links)."""
    from cubicsdr_amd.parallel import strong_scaling_plan as plan
    b, k = 8.0 * 32 * 1667072, 0.777
    assert plan(1, b, k)["choice"] == "single"
    p2 = plan(2, b, k)
    assert p2["choice"] == "single" and p2["ms"]["slab"] < p2["ms"]["broadcast"]
    assert plan(2, b, float("inf"))["choice"] in ("slab", "broadcast")
    for w in (4, 8):
        assert plan(w, b, k)["choice"] == "slab"
    assert plan(8, b, k)["speedup_over_one_gpu"] >= 6.0
    assert plan(8, b, k, ingest="rank0")["speedup_over_one_gpu"] < 2.0
