"""Multi-GPU partitioning of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests).

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


class ShardedStream:
    """ONE IQ stream whose DemodulatorInstances are spread over the ranks (BASELINE config 4; fan-out point
    SDRPostThread.cpp:389-396).  Every rank builds the same plan from the full demodulator list; per batch:

        rank `src` holds the raw IQ batch  ->  broadcast (RCCL over xGMI; gloo in CPU tests)  ->  every rank runs the
        channelizer for the channels ITS demodulators sit on (csdr_post_set_active_channels)  ->  its own bank slots.

    No reduction exists on this path: audio stays on the owning rank.  `group=False` skips the collective (single process
    driving several "virtual ranks" on one GPU: the sharded-equals-unsharded test)."""

    def __init__(self, device_index, rank, world, fs, M, block, demods, center, max_blocks, group=None, oversampled=False):
        from .engine import Context, DemodBank, SDRPost
        self.rank, self.world, self.group = rank, world, group
        self.fs, self.M, self.block, self.center, self.max_blocks = fs, M, block, center, max_blocks
        self.demods = list(demods)                               # (kind, bandwidth, frequency) of EVERY demodulator of the stream
        routed = [channel_at(f, center, fs, M) for _, _, f in self.demods]
        self.channels = [data_channel(c, M) for c in routed]
        self.plan = plan(len(self.demods), self.channels, world, rank)
        # the collective runs on torch's current stream: that stream is the library's boundary stream, so the kernels of a batch
        # are ordered behind its broadcast (csdr_ctx_create, "Streams") and the next broadcast behind their reads (join)
        stream = None
        try:
            import torch
            if torch.cuda.is_available():
                stream = torch.cuda.current_stream(device_index).cuda_stream
        except Exception:
            stream = None
        self.ctx = Context(device_index, stream=stream)
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
        if self.world > 1 and self.group is not False:
            self.ctx.join()                                      # the previous batch's kernels have read `iq` before it is overwritten
            broadcast_iq(iq, src=src, group=self.group)
        self.post.execute(iq, n_blocks, self.block, self.center)
        if self.plan.demods:
            self.bank.execute(self.post)

    def audio(self, demod_index):
        return self.bank.audio(self.slot_of[demod_index])

    def results(self, demod_index):
        return self.bank.results(self.slot_of[demod_index])

    def synchronize(self):
        self.ctx.synchronize()

    def close(self):
        self.bank.close(); self.post.close(); self.ctx.close()


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
