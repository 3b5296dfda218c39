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
