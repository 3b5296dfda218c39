"""Multi-GPU partitioning of the hot path: one process per GPU.  The collectives of a sharded stream go through the C ABI's communicator
(csdr_comm: RCCL over xGMI, include/csdr_hip.h "one stream over several GPUs") when the stream is given an engine.Comm -- that is what
bench.py --config C4 runs on a GPU node; without one they go through torch.distributed ("gloo" in the CPU tests, where no RCCL exists).
This module only plans (who owns which channel) and calls; the data path is the library's.

The path shards two ways:
  * by IQ stream  -- independent receivers, one per GPU, no data-path collective (BASELINE config 5; bench.py default);
  * by demodulator -- ONE IQ stream whose DemodulatorInstances are spread over the GPUs (BASELINE config 4): the ingest
    rank broadcasts each raw IQ batch (8 B/sample; 13.3 MB per 1/60 s at 100 MS/s, far below one xGMI link), every rank
    runs the polyphase front of the channelizer but only the DFT bins of the channels ITS demodulators sit on
    (csdr_post_set_active_channels), then its own demodulator slots.  Audio stays on the owning rank; no reduction exists
    on this path, so the broadcast is the only collective.
"""
from dataclasses import dataclass
from typing import List, Sequence


def shard_demods(n_demods: int, world: int, rank: int, channels: Sequence[int] = None) -> List[int]:
    """Demodulator indices owned by `rank`.  With `channels` (channel index of every demod) the split is by channel so
    that all demods of one channel land on one rank (each rank then needs the fewest DFT bins); without, contiguous."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    if channels is None:
        base, rem = divmod(n_demods, world)
        start = rank * base + min(rank, rem)
        return list(range(start, start + base + (1 if rank < rem else 0)))
    if len(channels) != n_demods:
        raise ValueError("channels must list one channel per demod")
    # greedy balance: heaviest channel first onto the lightest rank (deterministic tie-break by rank index)
    by_chan = {}
    for i, c in enumerate(channels):
        by_chan.setdefault(int(c), []).append(i)
    loads = [0] * world
    owner = {}
    for c in sorted(by_chan, key=lambda k: (-len(by_chan[k]), k)):
        r = min(range(world), key=lambda k: (loads[k], k))
        owner[c] = r
        loads[r] += len(by_chan[c])
    return sorted(i for c, idx in by_chan.items() if owner[c] == rank for i in idx)


def channels_of(demod_indices: Sequence[int], channels: Sequence[int]) -> List[int]:
    return sorted({int(channels[i]) for i in demod_indices})


@dataclass
class ShardPlan:
    rank: int
    world: int
    demods: List[int]
    active_channels: List[int]


def plan(n_demods: int, channels: Sequence[int], world: int, rank: int) -> ShardPlan:
    mine = shard_demods(n_demods, world, rank, channels)
    return ShardPlan(rank, world, mine, channels_of(mine, channels))


def channel_centers(center: int, fs: int, M: int) -> List[int]:
    """chanCenters[0 .. M] of SDRPostThread::updateChannels (SDRPostThread.cpp:116-124), integer arithmetic as there"""
    if M == 1:
        return [center, center + fs // 2]
    bw = fs // M
    c = [0] * (M + 1)
    for i in range(M // 2):
        ofs = bw * i
        c[i] = center + ofs
        c[i + M // 2] = center - fs // 2 + ofs
    c[M] = center + fs // 2
    return c


def channel_at(frequency: int, center: int, fs: int, M: int) -> int:
    """SDRPostThread::getChannelAt (:128-139): the nearest chanCenters entry (first wins on ties); -1 when none is closer than
    the sample rate.  Host twin of csdr_post_channel_at, so that every rank can plan without touching a device."""
    if M == 1:
        return 0
    best, min_delta = -1, fs
    for i, cc in enumerate(channel_centers(center, fs, M)):
        d = abs(frequency - cc)
        if d < min_delta:
            min_delta, best = d, i
    return best


def data_channel(ch: int, M: int) -> int:
    """the channelizer row a routed channel reads: index M (the upper band edge) wraps to M / 2 (:359-361)"""
    return M // 2 if (M > 1 and ch == M) else ch


class _Boundary:
    """The stream every collective and every torch buffer operation of a sharded stream runs on, and which the library uses as its
    boundary stream (csdr_ctx_create, "Streams"): a DEDICATED torch stream -- its raw handle is never 0, which the C ABI would read as
    "no stream, make a private one" and then order nothing against torch.  `on()` makes it torch's current stream (NCCL / RCCL
    collectives are ordered behind what is enqueued there and torch makes it wait for them), after making it wait for the
    caller's own current stream, on which the caller produced the tensors it hands over.  Without CUDA (gloo tests, the
    host-executing test build) there is nothing to order: `handle` is None and `on()` does nothing."""

    def __init__(self, device_index, use_torch=True):
        self.ts = None
        self.handle = None
        if use_torch:
            try:
                import torch
                if torch.cuda.is_available():
                    self.ts = torch.cuda.Stream(device=device_index)
                    self.handle = self.ts.cuda_stream
                    assert self.handle != 0
            except ImportError:
                pass

    def on(self):
        import contextlib
        if self.ts is None:
            return contextlib.nullcontext()
        import torch
        self.ts.wait_stream(torch.cuda.current_stream(self.ts.device))
        return torch.cuda.stream(self.ts)

    def release_to_caller(self):
        """the caller's current stream waits for what was enqueued on the boundary stream (e.g. before it reuses a batch tensor)"""
        if self.ts is not None:
            import torch
            torch.cuda.current_stream(self.ts.device).wait_stream(self.ts)


class ShardedStream:
    """ONE IQ stream whose DemodulatorInstances are spread over the ranks (BASELINE config 4; fan-out point
    SDRPostThread.cpp:389-396).  Every rank builds the same plan from the full demodulator list; per batch:

        rank `src` holds the raw IQ batch  ->  broadcast (RCCL over xGMI; gloo in CPU tests)  ->  every rank runs the
        channelizer for the channels ITS demodulators sit on (csdr_post_set_active_channels)  ->  its own bank slots.

    No reduction exists on this path: audio stays on the owning rank.  `group=False` skips the collective (single process
    driving several "virtual ranks" on one GPU: the sharded-equals-unsharded test)."""

    def __init__(self, device_index, rank, world, fs, M, block, demods, center, max_blocks, group=None, oversampled=False, comm_id=None):
        """comm_id: the 128-byte id of engine.Comm.unique_id() (same on every rank): the broadcast then runs through the C ABI's
        communicator (RCCL) instead of torch.distributed"""
        from .engine import Comm, Context, DemodBank, SDRPost
        self.rank, self.world, self.group = rank, world, group
        self.fs, self.M, self.block, self.center, self.max_blocks = fs, M, block, center, max_blocks
        self.demods = list(demods)                               # (kind, bandwidth, frequency) of EVERY demodulator of the stream
        routed = [channel_at(f, center, fs, M) for _, _, f in self.demods]
        self.channels = [data_channel(c, M) for c in routed]
        self.plan = plan(len(self.demods), self.channels, world, rank)
        # the collective runs on the boundary stream, so the kernels of a batch are ordered behind its broadcast (lane_begin) and
        # the next broadcast behind their reads (join)
        self.boundary = _Boundary(device_index)
        self.ctx = Context(device_index, stream=self.boundary.handle)
        assert self.boundary.handle is None or not self.ctx.owns_stream
        self.comm = Comm(self.ctx, comm_id, rank, world) if comm_id is not None else None
        self.post = SDRPost(self.ctx, fs, M, block, max_blocks=max_blocks, oversampled=oversampled)
        self.post.set_active_channels(self.plan.active_channels)
        self.bank = DemodBank(self.ctx, max(1, len(self.plan.demods)), max_blocks=max_blocks)
        self.slot_of = {}
        for slot, i in enumerate(self.plan.demods):
            kind, bw, f = self.demods[i]
            self.bank.configure(slot, self.post, kind, bw, f)
            self.slot_of[i] = slot

    def step(self, iq, n_blocks, src=0):
        """iq: the batch tensor, pre-allocated on every rank (valid on `src`); returns after the work is enqueued"""
        with self.boundary.on():
            if self.comm is not None:
                self.comm.broadcast(iq, n_blocks * self.block, src)        # (joins the lanes itself: csdr_comm.hip)
            elif self.world > 1 and self.group is not False:
                self.ctx.join()                                  # the previous batch's kernels have read `iq` before it is overwritten
                broadcast_iq(iq, src=src, group=self.group)
            self.post.execute(iq, n_blocks, self.block, self.center)
            if self.plan.demods:
                self.bank.execute(self.post)

    def release(self):
        """the caller's stream waits for the kernels enqueued so far (call before overwriting a batch tensor from another stream)"""
        with self.boundary.on():
            self.ctx.join()
        self.boundary.release_to_caller()

    def audio(self, demod_index):
        return self.bank.audio(self.slot_of[demod_index])

    def results(self, demod_index):
        return self.bank.results(self.slot_of[demod_index])

    def synchronize(self):
        self.ctx.synchronize()

    def close(self):
        if self.comm is not None:
            self.comm.close()
        self.bank.close(); self.post.close(); self.ctx.close()


def strong_scaling_plan(world: int, batch_bytes: float, kernels_ms_one_gpu: float, ingest: str = "distributed", link_GBps: float = 115.0,
                        chan_share: float = 0.33, row_bytes: float = None):
    """Which way to run ONE stream on `world` GPUs, from a link model (xGMI is point-to-point: one link per peer pair, ~115 GB/s sustained of
    153 peak) and the one-GPU kernel time of a batch.  Returns {"choice", "ms": {variant: projected batch time}}; variants:
      single     one GPU does everything (the others idle): kernels_ms_one_gpu
      broadcast  ShardedStream: the raw batch crosses every link whole; each rank runs the channelizer's FIR over every frame (chan_share of
                 the one-GPU kernels stays undivided) and 1 / world of the rest
      slab       SlabStream with step(overlap=True): every rank channelizes 1 / world of the frames; rows_bytes / world^2 cross each link, hidden
                 behind the next batch's kernels when shorter; ingest "rank0" adds the scatter of 1 / world of the batch over each link
                 (the same links, in front of the kernels), "distributed" (every rank reads its own slab: local_window) adds nothing.
    The choice is the smallest projected time: at world = 2 the one link carries a quarter of all channel samples (and, with rank-0 ingest,
    half of the input): a free-running stream is then link-bound on two GPUs and `single` wins unless the demodulators do not FIT one GPU --
    pass kernels_ms_one_gpu = inf in that case.  At real-time rates (0.8 GB/s per 100 MS/s stream) every variant is far inside one link and the
    choice is by capacity only."""
    if world < 1:
        raise ValueError("world")
    row_bytes = batch_bytes if row_bytes is None else row_bytes          # channel rows of a batch: 8 B per input sample, like the input
    link = link_GBps * 1e6                                               # bytes per ms
    ms = {"single": kernels_ms_one_gpu}
    if world > 1:
        ms["broadcast"] = max(batch_bytes / link, kernels_ms_one_gpu * (chan_share + (1.0 - chan_share) / world))
        exch = row_bytes / (world * world) / link
        scat = batch_bytes / world / link if ingest == "rank0" else 0.0
        ms["slab"] = max(kernels_ms_one_gpu / world, exch + scat)
    choice = min(ms, key=lambda k: (ms[k], k))
    return {"choice": choice, "ms": ms, "speedup_over_one_gpu": kernels_ms_one_gpu / ms[choice] if ms[choice] > 0 else None}


def slab_blocks(n_blocks: int, world: int):
    """[(first block, blocks)] per rank: contiguous, the remainder on the first ranks"""
    base, rem = divmod(n_blocks, world)
    out, start = [], 0
    for r in range(world):
        cnt = base + (1 if r < rem else 0)
        out.append((start, cnt))
        start += cnt
    return out


class SlabStream:
    """ONE IQ stream sharded by TIME for the channelizer and by CHANNEL for the demodulators (SURVEY.md 8e option 2) -- the variant
    of ShardedStream that divides the channelizer's work as well.  Per batch of n_blocks blocks:

        ingest rank: scatter  -> rank r holds blocks [b_r, b_r + n_r) of the batch plus the history_length input samples in front of them
        every rank:  channelizer over ITS blocks, all M channels (SDRPost.set_history gives the polyphase windows their true contents;
                     channel 0's DC blocker is a recurrence over the whole stream and is left to the channel's owner)
        all-to-all:  rank r sends rank q the rows of q's channels for r's frames (RCCL over xGMI; gloo in CPU tests):
                     8 bytes x frames x channels_q -- in total every channel sample crosses a link once
        every rank:  assembles the full rows of ITS channels in a second post object (import_*: DC blocker on channel 0 there),
                     then its own bank slots, as in ShardedStream.

    The three phases are separate methods (produce / exchange / consume) so that one process can drive several "virtual ranks" on one
    GPU and do the exchange by slicing (local_exchange): the sharded-equals-unsharded test.  Buffers are torch tensors on the GPU
    (float32 [n, 2]); with use_torch=False (a host-executing test build of the library) they are numpy arrays."""

    def __init__(self, device_index, rank, world, fs, M, block, demods, center, max_blocks, group=None, use_torch=True, comm_id=None):
        """comm_id: see ShardedStream: scatter and the row exchange then run through the C ABI (csdr_comm_scatter, csdr_post_exchange_rows)"""
        from .engine import Comm, Context, DemodBank, SDRPost
        self.rank, self.world, self.group, self.use_torch = rank, world, group, use_torch
        self.fs, self.M, self.block, self.center, self.max_blocks = fs, M, block, center, max_blocks
        self.bc = block // M                                       # frames (samples per channel) of one block
        self.demods = list(demods)
        routed = [channel_at(f, center, fs, M) for _, _, f in self.demods]
        self.channels = [data_channel(c, M) for c in routed]
        self.plans = [plan(len(self.demods), self.channels, world, q) for q in range(world)]
        self.plan = self.plans[rank]
        self.owned = [p.active_channels for p in self.plans]        # rows each rank needs
        self.device_index = device_index
        self.boundary = _Boundary(device_index, use_torch)
        self.ctx = Context(device_index, stream=self.boundary.handle)
        assert self.boundary.handle is None or not self.ctx.owns_stream
        self.comm = Comm(self.ctx, comm_id, rank, world) if comm_id is not None else None
        self.producer = SDRPost(self.ctx, fs, M, block, max_blocks=max(1, -(-max_blocks // world)))
        self.producer.set_dc_blocker(False)
        if self.comm is not None:
            # the producer writes its rows grouped by owning rank: its output buffer is the all-to-all's send buffer (no export copy)
            self.producer.set_row_order([c for o in self.owned for c in o])
        self.rows = SDRPost(self.ctx, fs, M, block, max_blocks=max_blocks)
        self.rows.set_active_channels(self.owned[rank] if self.owned[rank] else [0])
        self.hist = self.producer.history_length
        self.bank = DemodBank(self.ctx, max(1, len(self.plan.demods)), max_blocks=max_blocks)
        self.slot_of = {}
        for slot, i in enumerate(self.plan.demods):
            kind, bw, f = self.demods[i]
            self.bank.configure(slot, self.rows, kind, bw, f)
            self.slot_of[i] = slot
        with self.boundary.on():
            self._tail = self._empty(self.hist)                     # ingest rank: the input in front of the next batch
            self._zero(self._tail)
        self._keep = []
        self._pend = None                                           # step(overlap=True): the batch between the two halves of its exchange

    # ---- buffers
    def _empty(self, n_samples):
        if self.use_torch:
            import torch
            dev = torch.device("cuda", self.device_index) if torch.cuda.is_available() else torch.device("cpu")
            return torch.empty((max(1, n_samples), 2), dtype=torch.float32, device=dev)
        import numpy as np
        return np.empty((max(1, n_samples), 2), np.float32)

    @staticmethod
    def _t(buf):
        """the torch view a collective needs (numpy buffers of the CPU emulation share their memory with it)"""
        if hasattr(buf, "data_ptr"):
            return buf
        import torch
        return torch.from_numpy(buf)

    @staticmethod
    def _zero(buf):
        if hasattr(buf, "zero_"):
            buf.zero_()
        else:
            buf[...] = 0

    # ---- ingest
    def extended(self, batch, n_blocks):
        """ingest rank: [history | batch] as one buffer and the new history; rank r's input is a window of it"""
        n = n_blocks * self.block
        with self.boundary.on():
            # buffers come from torch's allocator, which only knows about torch's streams: the library's kernels that read the
            # previous batch's buffers are joined into the boundary stream before anything is allocated or overwritten on it
            self.ctx.join()
            ext = self._empty(self.hist + n)
            ext[:self.hist] = self._tail
            ext[self.hist:self.hist + n] = batch[:n]
            self._tail = self._empty(self.hist)
            self._tail[...] = ext[n:n + self.hist]
        return ext

    def window(self, ext, n_blocks, r):
        start, cnt = slab_blocks(n_blocks, self.world)[r]
        return ext[start * self.block: start * self.block + self.hist + cnt * self.block]

    def scatter(self, batch, n_blocks, src=0):
        """collective: every rank gets its window [history | its blocks].  Through torch.distributed: one buffer, needs n_blocks % world == 0
        (equal windows).  Through the C ABI (comm_id given): returns (history, blocks) -- see _scatter_abi."""
        if self.comm is not None:
            return self._scatter_abi(batch, n_blocks, src)
        import torch.distributed as dist
        if n_blocks % self.world:
            raise ValueError("scatter needs the batch's blocks to divide evenly over the ranks")
        with self.boundary.on():
            self.ctx.join()                                         # the previous batch's kernels have read the window buffer
            each = self.hist + (n_blocks // self.world) * self.block
            mine = self._empty(each)
            parts = None
            if self.rank == src:
                ext = self.extended(batch, n_blocks)
                parts = [self._t(self.window(ext, n_blocks, r)).contiguous() for r in range(self.world)]
            if self._host_staged():
                # gloo moves host memory only (the one-GPU dry run of the multi-rank control flow): stage the windows through the host
                mine_h = self._t(mine).cpu()
                dist.scatter(mine_h, [p.cpu() for p in parts] if parts is not None else None, src=src, group=self.group)
                self._t(mine).copy_(mine_h)
            else:
                dist.scatter(self._t(mine), parts, src=src, group=self.group)
        return mine

    def _host_staged(self):
        """True when the collectives of this stream run through a gloo group while the buffers are device memory"""
        import torch.distributed as dist
        return bool(self.use_torch and self.boundary.ts is not None and dist.get_backend(self.group) == "gloo")

    def _scatter_abi(self, batch, n_blocks, src):
        """The ingest rank sends every OTHER rank its window straight out of the batch -- a window starts `hist` samples in front of the rank's
        first block: inside the batch for every slab but the first, whose history is the tail of the previous batch -- as one group of
        point-to-point transfers (csdr_comm_p2p); its own slab is used where it lies.  No packing, no copy of the batch on the ingest rank.
        Returns (history, blocks): `hist` samples and the rank's slab (uneven slabs are fine)."""
        sl = slab_blocks(n_blocks, self.world)
        start, cnt = sl[self.rank]
        n = n_blocks * self.block
        if any(s and s * self.block < self.hist for s, _ in sl):
            raise ValueError("blocks shorter than the channelizer's history: use torch.distributed's scatter path")
        with self.boundary.on():
            if self.rank == src:
                ops = []
                for r, (s0, c) in enumerate(sl):
                    if r == src or c == 0:
                        continue
                    if s0 == 0:
                        ops += [(r, False, self._tail, 0, self.hist), (r, False, batch, 0, c * self.block)]
                    else:
                        ops.append((r, False, batch, s0 * self.block - self.hist, self.hist + c * self.block))
                self.comm.p2p(ops)                                   # (joins the library's lanes first)
                tail = self._tail if start == 0 else batch[start * self.block - self.hist: start * self.block]
                blocks = batch[start * self.block:(start + cnt) * self.block]
                new_tail = self._empty(self.hist)                    # the input in front of the NEXT batch
                if n >= self.hist:
                    new_tail[...] = batch[n - self.hist:n]
                else:
                    new_tail[:self.hist - n] = self._tail[n:]
                    new_tail[self.hist - n:] = batch[:n]
                self._keep_scatter = (self._tail, batch)
                self._tail = new_tail
            else:
                recv = self._empty(self.hist + cnt * self.block)
                if cnt == 0:
                    ops = []
                elif start == 0:
                    ops = [(src, True, recv, 0, self.hist), (src, True, recv, self.hist, cnt * self.block)]
                else:
                    ops = [(src, True, recv, 0, self.hist + cnt * self.block)]
                self.comm.p2p(ops)
                tail, blocks = recv[:self.hist], recv[self.hist:]
                self._keep_scatter = recv
        return tail, blocks

    # ---- the three phases
    def produce(self, window, n_blocks):
        """channelize this rank's blocks; returns the packed rows for every peer: [peer q][q's channels][this rank's frames]"""
        start, cnt = slab_blocks(n_blocks, self.world)[self.rank]
        frames = cnt * self.bc
        with self.boundary.on():
            self.ctx.join()                                         # (see extended(): the previous batch's buffers are dropped below)
            send = self._empty(sum(len(o) for o in self.owned) * frames)
            tail, blocks = window if isinstance(window, tuple) else (window, window[self.hist:])      # (history, blocks) views, or one buffer [history | blocks]
            if cnt:
                self.producer.set_history(tail, self.hist)
                self.producer.execute(blocks, cnt, self.block, self.center)
                off = 0
                for q in range(self.world):
                    if self.owned[q]:
                        self.producer.export_rows(self.owned[q], send[off:], frames)
                        off += len(self.owned[q]) * frames
            self._keep = [window, send]
        return send

    def splits(self, n_blocks):
        """(samples this rank sends to each peer, samples it receives from each peer)"""
        sl = slab_blocks(n_blocks, self.world)
        mine = sl[self.rank][1] * self.bc
        return ([len(self.owned[q]) * mine for q in range(self.world)],
                [len(self.owned[self.rank]) * sl[p][1] * self.bc for p in range(self.world)])

    def exchange(self, send, n_blocks):
        import torch.distributed as dist
        ins, outs = self.splits(n_blocks)
        with self.boundary.on():
            self.ctx.join()                                         # the export kernels have filled `send`
            recv = self._empty(sum(outs))
            if self._host_staged():
                # (gloo has no all-to-all either: one broadcast per source rank of its whole send buffer, every rank keeps its part)
                import torch
                sizes = [None] * self.world
                dist.all_gather_object(sizes, int(sum(ins)), group=self.group)
                off = 0
                for p in range(self.world):
                    buf = self._t(send)[:sizes[p]].cpu().contiguous() if p == self.rank else torch.empty((max(1, sizes[p]), 2), dtype=torch.float32)
                    dist.broadcast(buf, src=p, group=self.group)
                    mine_from_p = outs[p]
                    o = sum(len(self.owned[q]) for q in range(self.rank)) * (slab_blocks(n_blocks, self.world)[p][1] * self.bc)
                    self._t(recv)[off:off + mine_from_p].copy_(buf[o:o + mine_from_p])
                    off += mine_from_p
            else:
                dist.all_to_all_single(self._t(recv)[:sum(outs)].view(-1), self._t(send)[:sum(ins)].view(-1), [2 * v for v in outs], [2 * v for v in ins], group=self.group)
        return recv

    def consume(self, recv, n_blocks):
        sl = slab_blocks(n_blocks, self.world)
        mine = self.owned[self.rank]
        with self.boundary.on():
            self.rows.import_begin(n_blocks, self.block, self.center)
            off = 0
            for p in range(self.world):
                frames = sl[p][1] * self.bc
                if mine and frames:
                    self.rows.import_rows(mine, recv[off:], frames, sl[p][0] * self.bc, frames)
                    off += len(mine) * frames
            self.rows.import_commit()
            self._keep.append(recv)
            if self.plan.demods:
                self.bank.execute(self.rows)

    def step(self, window, n_blocks, overlap=False):
        """one batch.  overlap=True runs the software pipeline of csdr_hip.h "The same exchange in two halves": this batch is channelized and its
        row transfers are started, THEN the previous batch is imported and demodulated -- the transfers of batch i run beside the channelizer of
        batch i + 1.  The results of a batch (audio, results) are then complete after the NEXT step(overlap=True) or after flush(); they are the
        sequential form's bit for bit."""
        if self.comm is None:
            if not overlap:
                self.consume(self.exchange(self.produce(window, n_blocks), n_blocks), n_blocks)
                return
            recv = self.exchange(self.produce(window, n_blocks), n_blocks)     # (torch.distributed collectives return when they are done: the order is what this mode checks)
            if self._pend is not None:
                self.consume(*self._pend)
            self._pend = (recv, n_blocks)
            return
        # through the C ABI: producer execute, then the row exchange (RCCL) -> import -> commit
        sl = slab_blocks(n_blocks, self.world)
        start, cnt = sl[self.rank]
        tail, blocks = window if isinstance(window, tuple) else (window[:self.hist], window[self.hist:])
        f0, fr = [b * self.bc for b, _ in sl], [c * self.bc for _, c in sl]
        with self.boundary.on():
            self.ctx.join()
            if cnt:
                self.producer.set_history(tail, self.hist)
                self.producer.execute(blocks, cnt, self.block, self.center)
            if not overlap:
                self.comm.exchange_rows(self.producer, self.rows, self.owned, f0, fr, n_blocks, self.block, self.center)
                self._keep = [window]
                if self.plan.demods:
                    self.bank.execute(self.rows)
                return
            self.comm.exchange_rows_begin(self.producer, self.owned, f0, fr)
            self._keep = [window]
            if self._pend is not None:
                self._finish_pending()
            self._pend = (n_blocks,)

    def _finish_pending(self):
        (n_blocks,) = self._pend
        self.comm.exchange_rows_finish(self.rows, n_blocks, self.block, self.center)
        if self.plan.demods:
            self.bank.execute(self.rows)
        self._pend = None

    def flush(self):
        """drain the pipeline of step(overlap=True): import and demodulate the batch whose transfers are still out"""
        if self._pend is None:
            return
        with self.boundary.on():
            if self.comm is None:
                pend, self._pend = self._pend, None
                self.consume(*pend)
            else:
                self._finish_pending()

    def local_window(self, ring, n_blocks):
        """distributed ingest: every rank already holds the stream's samples (its own reader, or -- the bench -- the same synthetic ring on every
        rank) and takes ITS window where it lies: (history, blocks) views, nothing moves.  `ring` holds the batch; the history in front of slab 0
        is the end of the previous batch, which for a ring that repeats is the ring's own end."""
        start, cnt = slab_blocks(n_blocks, self.world)[self.rank]
        n = n_blocks * self.block
        if start == 0:
            tail = ring[n - self.hist:n]
        else:
            tail = ring[start * self.block - self.hist: start * self.block]
        return tail, ring[start * self.block:(start + cnt) * self.block]

    def audio(self, demod_index):
        return self.bank.audio(self.slot_of[demod_index])

    def results(self, demod_index):
        return self.bank.results(self.slot_of[demod_index])

    def synchronize(self):
        self.ctx.synchronize()

    def close(self):
        if self.comm is not None:
            self.comm.close()
        self.bank.close(); self.rows.close(); self.producer.close(); self.ctx.close()


def local_exchange(streams, sends, n_blocks):
    """the all-to-all of SlabStream done by slicing, for virtual ranks inside one process: returns each rank's receive buffer"""
    world = len(streams)
    for s in streams:
        s.ctx.synchronize()
    recvs = []
    for q, sq in enumerate(streams):
        _, outs = sq.splits(n_blocks)
        with sq.boundary.on():
            recv = sq._empty(sum(outs))
            off = 0
            for p, sp in enumerate(streams):
                ins, _ = sp.splits(n_blocks)
                o = sum(ins[:q])
                recv[off:off + outs[p]] = sends[p][o:o + ins[q]]
                off += outs[p]
        recvs.append(recv)
    return recvs


def exchange_id(rank: int, world: int, host: str = None, port: int = None, timeout_s: float = 120.0) -> bytes:
    """the 128-byte communicator id: rank 0 makes it (csdr_comm_unique_id), the others read it from a TCP key-value store on
    MASTER_ADDR : MASTER_PORT + 1 (no process group, no collective library involved)"""
    import datetime
    import os
    from torch.distributed import TCPStore
    from .engine import Comm
    if world == 1:
        return Comm.unique_id()
    host = host or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = port or int(os.environ.get("MASTER_PORT", "29500")) + 1
    store = TCPStore(host, port, world, is_master=(rank == 0), timeout=datetime.timedelta(seconds=timeout_s))
    if rank == 0:
        store.set("csdr_comm_id", Comm.unique_id())
    return bytes(store.get("csdr_comm_id"))


def broadcast_iq(batch, src: int = 0, group=None):
    """Broadcast one raw IQ batch (float32 [n, 2] or complex64 [n] tensor, pre-allocated on every rank) from the ingest
    rank.  One collective per batch, issued on the current stream so it overlaps the previous batch's kernels on the
    library's own stream."""
    import torch.distributed as dist
    dist.broadcast(batch, src=src, group=group)
    return batch


def max_over_ranks(seconds: float, device=None) -> float:
    """max-reduce of a scalar time (bench.py contract: value uses the slowest rank)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: int, device=None) -> int:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
