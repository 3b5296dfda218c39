"""bench.py --config C4: BASELINE config 4 -- ONE 100 MS/s IQ stream, firpfbch M = 1024, one NBFM demodulator per channel, the
demodulators sharded over the ranks (cubicsdr_amd.parallel.ShardedStream): rank 0 owns the HBM-resident ring, every batch is
broadcast (RCCL over xGMI), every rank channelizes for its own channels and runs its own slots.  scaling = "strong": the value is
the one stream's MS/s (total work fixed as N grows).  --shard slab runs the time-slab variant instead (SURVEY.md 8e option 2,
cubicsdr_amd.parallel.SlabStream): scatter of [history | slab] windows, every rank channelizes ITS blocks for all channels, an
all-to-all hands every rank the rows of its channels -- the channelizer's work is divided too."""
import json
import os
import sys
import time

FS, M, BLOCK, CENTER = 100_000_000, 1024, 1_667_072, 400_000_000        # BASELINE config 4 names M = 1024; block = ceil(fs / 60 / M) M (SoapySDRThread.cpp:668-674)
NBFM_BW, AUDIO = 12_500, 48_000


def main(args, line_out=None):
    out = measure(args)
    if int(os.environ.get("RANK", "0")) == 0:
        line_out = line_out or sys.stdout
        line_out.write(json.dumps(out) + "\n"); line_out.flush()


def measure(args, dist=None):
    """one C4 measurement; collective over the ranks of the job, returns the bench line as a dict (rank 0's is the one that counts).
    `dist`: an initialised torch.distributed module (bench.py's strong-scaling leg calls with its own process group): "nccl" (= RCCL) -> the
    library's own communicator is used next to it; "gloo" (the one-GPU dry run) -> the collectives go through that group, staged on the host."""
    import torch
    from cubicsdr_amd import build as cbuild
    from cubicsdr_amd.parallel import ShardedStream, SlabStream, channel_centers
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        cbuild.build(verbose=False)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # transport of the collectives: the library's own communicator (csdr_comm: RCCL behind the C ABI, no torch.distributed process group; the
    # id travels over a TCP store on MASTER_PORT + 1), or CSDR_C4_TRANSPORT=torch: torch.distributed ("nccl" is RCCL; CSDR_DIST_BACKEND overrides
    # it for one-GPU dry runs of the multi-rank control flow)
    own_group = False
    cid = None
    if dist is not None:
        if dist.get_backend() == "nccl" and os.environ.get("CSDR_C4_TRANSPORT", "abi") == "abi":
            from cubicsdr_amd.parallel import exchange_id
            cid = exchange_id(rank, world)
            dist = None
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("CSDR_C4_TRANSPORT", "abi") == "abi" and os.environ.get("CSDR_DIST_BACKEND", "nccl") == "nccl":
            from cubicsdr_amd.parallel import exchange_id
            cid = exchange_id(rank, world)
        else:
            import torch.distributed as dist
            dist.init_process_group(os.environ.get("CSDR_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
            dist.barrier()
            own_group = True
    # the timed pipeline runs its stages on ONE HIP stream unless told otherwise (as bench.py's default configuration does): every kernel runs
    # alone, so the live per-kernel durations are the kernels' own; the library's default folding (three streams) overlaps the channelizer of
    # batch i + 1 with the demodulators of batch i: higher throughput, stretched kernel durations
    streams = int(os.environ.get("CSDR_STREAMS", getattr(args, "streams", 1)))
    os.environ["CSDR_STREAMS"] = str(streams)
    NB = args.blocks or 32          # 32 blocks = 427 MB of input per batch: beyond the 256 MB Infinity Cache (16 blocks fit in it and measure 0.39 instead of 0.27 of peak on the channelizer)
    NBATCH = args.batches or 24
    cc = channel_centers(CENTER, FS, M)
    demods = [("NBFM", NBFM_BW, cc[ch] + 3700) for ch in range(M)]          # one per channel, 3.7 kHz off the channel centre (forces the NCO)
    g = torch.Generator(device=device); g.manual_seed(0xC0B1C5D2)
    ring = torch.randn(NB * BLOCK, 2, generator=g, device=device, dtype=torch.float32) * 0.05 + 0.01
    # --shard auto: the link model of parallel.strong_scaling_plan picks the variant for this world size from the one-GPU kernel time of a batch
    # (profiles/r05_c4slab_bench.json: 0.777 ms per 32-block batch).  Where it says one GPU alone is fastest (two GPUs, free-running: the one link
    # between them carries a quarter of all channel samples) the slab variant is still what runs -- the measurement is of the multi-GPU path -- and
    # the line says so.
    shard = getattr(args, "shard", "broadcast")
    ingest = getattr(args, "ingest", None) or "distributed"
    policy = None
    if shard == "auto":
        from cubicsdr_amd.parallel import strong_scaling_plan
        policy = strong_scaling_plan(world, 8.0 * NB * BLOCK, 0.777 * NB / 32.0, ingest=ingest)
        shard = "broadcast" if (world == 1 or policy["choice"] == "broadcast") else "slab"      # (one GPU: the unsharded path, which the broadcast driver with one rank is)
    slab = shard == "slab"
    overlap = slab and bool(getattr(args, "overlap", True))
    if slab and cid is None and dist is None and os.environ.get("CSDR_C4_TRANSPORT", "abi") == "abi":
        from cubicsdr_amd.parallel import exchange_id
        cid = exchange_id(rank, world)              # one rank: the same calls (scatter, packed producer rows, csdr_post_exchange_rows) on a one-rank communicator
    if slab and NB % world:
        raise SystemExit("--shard slab needs --blocks divisible by the number of GPUs")
    if slab:
        st = SlabStream(local_rank, rank, world, FS, M, BLOCK, demods, CENTER, NB, comm_id=cid)
    else:
        st = ShardedStream(local_rank, rank, world, FS, M, BLOCK, demods, CENTER, NB, comm_id=cid)

    def step():
        for _ in range(NBATCH):
            if not slab:
                st.step(ring, NB, src=0)
            elif ingest == "distributed" and (world > 1 or st.comm is not None):
                # every rank holds the stream's samples (here: the same synthetic ring, same seed, on every rank -- a deployment: its own reader
                # hands each GPU its time slab over that GPU's own host link) and takes its window where it lies: no scatter
                st.step(st.local_window(ring, NB), NB, overlap=overlap)
            elif world > 1 or st.comm is not None:
                st.step(st.scatter(ring if rank == 0 else None, NB, src=0), NB, overlap=overlap)
            else:
                ext = st.extended(ring, NB)
                st.consume(st.produce(st.window(ext, NB, 0), NB), NB)      # one rank: its own rows come straight back
        if slab and overlap:
            st.flush()

    for _ in range(args.warmup):
        step()
    st.synchronize(); torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elif st.comm is not None:
        st.comm.barrier()
    profile_period = 4
    if not args.no_profile:
        st.ctx.profile_enable(profile_period)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    st.synchronize(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = {}
    if not args.no_profile:
        prof = st.ctx.profile()
        st.ctx.profile_enable(False)
    if dist:
        dist.barrier()
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    elif st.comm is not None:
        elapsed = st.comm.max(elapsed)                   # max over the ranks (also a barrier)
    samples = args.steps * NBATCH * NB * BLOCK                                   # ONE stream: not multiplied by the world size
    value = samples / elapsed / 1e6
    bytes_per_sample = 8 + 8 + 8.0 * 1.0 + 4.0 * M * AUDIO / FS                  # SURVEY.md 8d, C4 (no FFT): 26.0
    out = {"metric": "IQ MS/s sustained @ N demods + FFT size", "value": value, "unit": "MS/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "C4: 1024-channel firpfbch + 1024x NBFM (one per channel), 100 MS/s IQ, demodulators sharded over the ranks, "
                                  + ("time slabs scattered from rank 0, channel rows exchanged all-to-all (RCCL)" if slab else "IQ batches broadcast from rank 0 (RCCL)"),
                      "shard": "slab" if slab else "broadcast", "ingest": (ingest if slab else "rank0"), "exchange_overlapped_with_next_batch": bool(overlap),
                      "policy": policy,
                      "batches_per_step": NBATCH, "blocks_per_batch": NB, "block_len": BLOCK, "n_demods": M, "demods_on_rank0": len(st.plan.demods),
                      "channels_on_rank0": len(st.plan.active_channels), "realtime_multiple": value / (FS / 1e6), "timed_region_s": elapsed,
                      "parallelism": ("time slabs -> per-rank channelizer -> all-to-all of channel rows -> per-rank bank" if slab else
                                      "dp over demodulators (one IQ stream): broadcast + per-rank channel subset + per-rank bank"),
                      "streams": streams, "transport": ("torch.distributed (%s)" % dist.get_backend()) if dist else "csdr_comm (RCCL through the C ABI)" if cid is not None else "none (one rank)",
                      "rccl_ranks": st.comm.world_size if st.comm is not None else 0},
           "roofline": {"bound": "hbm", "whole_path": {"bytes_per_sample": round(bytes_per_sample, 1), "achieved": bytes_per_sample * value * 1e6 / 1e9,
                                                       "frac": bytes_per_sample * value * 1e6 / 1e9 / 8000.0 / world}}}
    if prof:
        # the dominant kernel of rank 0 (every rank runs the same kernels on its share): live HIP-event durations, as in bench.py
        n_batches = args.steps * NBATCH
        avg = {k: v[0] / v[1] for k, v in prof.items()}
        per_batch = {k: avg[k] * (v[2] / n_batches) for k, v in prof.items()}
        dom = max(per_batch, key=lambda k: per_batch[k])
        lpb = prof[dom][2] / n_batches
        nb_rank = NB // world if slab else NB                                     # blocks this rank channelizes per batch
        share_ch = 1.0 if slab else len(st.plan.active_channels) / float(M)      # channel rows this rank writes
        share_dm = len(st.plan.demods) / float(M)
        if dom == "chan_analyze":
            bps = 8.0 + 8.0 * share_ch
        elif dom.startswith("demod_frontend"):
            bps = 8.0 * share_dm
        elif dom == "demod_audio_interp":
            bps = 4.0 * len(st.plan.demods) * AUDIO / FS
        else:
            bps = 0.0
        units = (nb_rank if dom == "chan_analyze" else NB) * BLOCK
        alg = bps * units / lpb
        traffic, traffic_file = None, None
        if world == 1 and not slab:
            # HBM bytes of the dominant kernel from the committed PMC pass of this command (profiles/collect.sh; one rank, broadcast variant)
            try:
                import bench
                per_block, traffic_file = bench.measured_traffic("C4")
                if dom in per_block:
                    traffic = per_block[dom] * (units / BLOCK) / lpb
            except Exception:
                traffic = None
        out["roofline"].update({"kernel": dom, "achieved": alg / (avg[dom] * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / (avg[dom] * 1e-3) / 1e9 / 8000.0,
                                "traffic": traffic, "traffic_unit": ("HBM bytes per launch of the dominant kernel (PMC pass %s)" % traffic_file) if traffic is not None else None,
                                "avg_launch_ms": avg[dom], "launches_per_batch": lpb, "algorithmic_bytes_per_launch": alg,
                                "profile_sampling": "HIP events around every %d-th launch of each kernel id on rank 0, inside the timed region" % profile_period,
                                "kernels_ms_per_batch": {k: per_batch[k] for k in sorted(per_batch, key=lambda k: -per_batch[k])}})
    st.close()
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        # CPU baseline on a bounded sample of the same workload: the M = 1024 firpfbch over every block and 64 of the 1024 NBFM chains
        # (building all 1024 liquid resamplers alone takes minutes), no spectrum; one thread and thread-per-stage
        try:
            import bench
            sub = dict(fs=FS, M=M, block=BLOCK, n_demods=64, fft=0, kinds=["NBFM"], name="C4")
            cb = bench.cpu_baseline(sub, ring[: min(NB, 4) * BLOCK].cpu().numpy().view("complex64").reshape(-1), args.cpu_seconds)
            cb["sample"] = "channelizer M = 1024 over every block + 64 of the 1024 NBFM demodulator chains, no spectrum -- " + cb["sample"]
            out["cpu_baseline"] = cb
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "MS/s", "cores": 0, "kind": "unavailable", "sample": repr(e)}
    if dist and own_group:
        dist.destroy_process_group()
    return out
