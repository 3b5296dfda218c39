#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native CubicSDR hot path (contract: see task statement / DESIGN.md).

Workload = BASELINE.json configs[1] ("C2"): 64x NBFM demodulators, 10 MS/s complex-float IQ, firpfbch M = 20
(block = 166 680 samples, channel rate 500 kS/s), 16384-point spectrum FFT (internal 32768), every sample FFT'ed
("contiguous" frames).  One step = one pass of the whole hot path (channelizer + 64 demodulator chains
+ spectrum) over one batch of `--blocks` consecutive IQ blocks that are already resident in HBM.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Multi-GPU: each rank owns an independent IQ stream with its own demodulators (BASELINE config 5 style partitioning:
no data-path collective); value = samples processed by all ranks / max-over-ranks time; scaling = weak.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s peak

FS = 10_000_000
M = 20
BLOCK = 166_680            # ceil(floor(Fs/60)/M)*M, SoapySDRThread.cpp:668-674
N_DEMODS = 64
FFT_SIZE = 16384
PROFILE_PERIOD = 8          # per-kernel HIP events bracket every 8th launch (bracketing all of them costs ~7 % of the throughput)
CENTER = 100_000_000
NBFM_BW = 12_500
AUDIO_RATE = 48_000


def demod_frequencies(center, fs, n):
    return [int(center + (k + 0.37) * fs / n - fs / 2) for k in range(n)]


def make_ring(torch, device, n_blocks, seed):
    """synthetic IQ ring in HBM: noise sigma 0.05 + one NBFM carrier per demod + DC offset."""
    n = n_blocks * BLOCK
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.randn(n, 2, generator=g, device=device, dtype=torch.float32) * 0.05
    t = torch.arange(n, device=device, dtype=torch.float64) / FS
    amp = 0.5 / math.sqrt(N_DEMODS)
    acc_r = torch.zeros(n, device=device, dtype=torch.float32)
    acc_i = torch.zeros(n, device=device, dtype=torch.float32)
    mod = (2500.0 / 1000.0) * torch.sin(2 * math.pi * 1000.0 * t)
    for f in demod_frequencies(CENTER, FS, N_DEMODS):
        ph = (2 * math.pi * (f - CENTER)) * t + mod
        ph = torch.remainder(ph, 2 * math.pi)
        acc_r += (amp * torch.cos(ph)).float()
        acc_i += (amp * torch.sin(ph)).float()
    x[:, 0] += acc_r + 0.01
    x[:, 1] += acc_i + 0.01
    return x.contiguous()


# algorithmic HBM bytes per INPUT SAMPLE attributed to each kernel (DESIGN.md "Roofline accounting"):
#   ingest read 8 + channelizer write 8 ; demod reads 8 N/M + audio/IQ writes ; spectrum read 8 + display write 4
def algorithmic_bytes_per_sample(kernel, n_demods, m, fft_n):
    audio = 4.0 * n_demods * AUDIO_RATE / FS
    table = {
        "chan_analyze": 16.0,
        "demod_frontend": 8.0 * n_demods / m,
        "demod_modem": 0.0,
        "demod_audio_interp": audio,
        "spec_fft_radix": 8.0,         # the frame is read once from HBM ...
        "spec_fft_rows": 0.0,          # ... the second pass re-reads an intermediate that is not algorithmic traffic
        "spec_average": 0.0,
        "spec_display": 4.0,
    }
    return table.get(kernel, 0.0)


def measured_traffic_bytes(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC pass (profiles/collect.sh: FETCH_SIZE and WRITE_SIZE in their
    own rocprofv3 runs of this same command; gfx950 reports half of a wide coalesced read stream, MI355X_MICROARCH.md "HBM",
    so the read side is doubled), per IQ block of the launch.  None when no pass is on file for this kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    try:
        t = json.load(open(files[-1]))                      # the latest committed pass
    except Exception:
        return None
    blocks = float(t.get("_meta", {}).get("blocks_per_launch", 64))
    for name, v in t.items():
        if name.split("<")[0] in (kernel, kernel + "_s") and "FETCH_SIZE_KiB_avg_per_launch" in v and "WRITE_SIZE_KiB_avg_per_launch" in v:
            return (2.0 * v["FETCH_SIZE_KiB_avg_per_launch"] + v["WRITE_SIZE_KiB_avg_per_launch"]) * 1024.0 / blocks
    return None


def cpu_baseline(ring_host, target_seconds):
    """time the reference CPU path (single thread) on a bounded sample of the same workload"""
    import numpy as np
    import oracle.liquid_api as A
    kind = "reference" if A.available("ref") else "port"
    L = A.load("ref" if kind == "reference" else "port")
    L.oracle_chain_run.restype = C.c_double
    L.oracle_chain_run.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    chan_bw = FS // M
    centers = [CENTER + chan_bw * i for i in range(M // 2)] + [CENTER - FS // 2 + chan_bw * i for i in range(M // 2)] + [CENTER + FS // 2]
    ch, nf, md = [], [], []
    for f in demod_frequencies(CENTER, FS, N_DEMODS):
        i = min(range(M + 1), key=lambda k: (abs(f - centers[k]), k))
        shift = f - centers[i]
        ch.append(i)
        nf.append(np.float32(2.0 * math.pi * abs(shift) / chan_bw))
        md.append(0 if shift == 0 else (1 if shift < 0 else -1))
    ch = np.array(ch, np.int32); nf = np.array(nf, np.float32); md = np.array(md, np.int32)
    ring_blocks = ring_host.size // BLOCK
    t = np.zeros(3)
    na = C.c_longlong()

    def run(nb):
        return L.oracle_chain_run(M, BLOCK, nb, ring_blocks, ring_host.ctypes.data_as(C.c_void_p), N_DEMODS, ch.ctypes.data_as(C.c_void_p),
                                  nf.ctypes.data_as(C.c_void_p), md.ctypes.data_as(C.c_void_p), float(NBFM_BW) / chan_bw,
                                  float(AUDIO_RATE) / NBFM_BW, 2 * FFT_SIZE, t.ctypes.data_as(C.c_void_p), C.byref(na))
    probe = run(8)
    nb = max(8, int(target_seconds / (probe / 8)))
    T = run(nb)
    return {"value": nb * BLOCK / T / 1e6, "unit": "MS/s", "cores": 1, "kind": kind,
            "sample": "%d blocks x %d samples of the same C2 workload (64 NBFM chains + M=20 firpfbch + contiguous 32768-pt FFT frames), %.1f s single thread: channelizer %.0f%%, demodulators %.0f%%, spectrum %.0f%%"
                      % (nb, BLOCK, T, 100 * t[0] / T, 100 * t[1] / T, 100 * t[2] / T)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--blocks", type=int, default=0, help="IQ blocks per step (batch resident in HBM); default 256 (C2) / 16 (C3)")
    ap.add_argument("--config", default="C2", choices=["C2", "C3"],
                    help="BASELINE.json workload: C2 (default, the judged one) or C3 = 256 mixed NBFM/AM/USB demods, 61.44 MS/s, M=122, 65536-pt FFT (reported, not judged)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline sample budget (0 disables)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event profile")
    args = ap.parse_args()

    global FS, M, BLOCK, N_DEMODS, FFT_SIZE
    kinds = ["NBFM"]
    workload = "C2: 64x NBFM demods (12.5 kHz -> 48 kHz audio), 10 MS/s complex-float IQ, firpfbch M=20, 16384-pt spectrum FFT (internal 32768) over every sample"
    if args.config == "C3":
        FS, M, BLOCK, N_DEMODS, FFT_SIZE = 61_440_000, 122, 1_024_068, 256, 65536          # SoapySDRThread.cpp:668-693 for 61.44 MS/s
        kinds = ["NBFM", "AM", "USB"]
        workload = "C3: 256 mixed NBFM/AM/USB demods, 61.44 MS/s complex-float IQ, firpfbch M=122, 65536-pt spectrum FFT (internal 131072) over every sample"
        args.cpu_seconds = 0.0                          # the CPU sample is defined for the judged workload only
    if not args.blocks:
        args.blocks = 256 if args.config == "C2" else 16
    bytes_per_sample = 8 + 8 + 8.0 * N_DEMODS / M + 4.0 * N_DEMODS * AUDIO_RATE / FS + 12      # 54.8 (C2), 45.6 (C3)
    import torch
    from cubicsdr_amd import build as cbuild
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        cbuild.build(verbose=False)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
        dist.barrier()
    else:
        torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from cubicsdr_amd.engine import Context, DemodBank, SDRPost, SpectrumProcessor
    NB = args.blocks
    ring = make_ring(torch, device, NB, seed=0xC0B1C5D2 + rank)
    torch.cuda.synchronize()
    n_frames_max = (NB * BLOCK) // (2 * FFT_SIZE) + 2

    def make_pipeline():
        c = Context(local_rank)        # one HIP stream per pipeline stage inside (include/csdr_hip.h "Streams")
        p = SDRPost(c, FS, M, BLOCK, max_blocks=NB)
        b = DemodBank(c, N_DEMODS, max_blocks=NB)
        for i, f in enumerate(demod_frequencies(CENTER, FS, N_DEMODS)):
            kind = kinds[i % len(kinds)]
            b.configure(i, p, kind, {"NBFM": NBFM_BW, "AM": 6000, "USB": 5400}[kind], f, AUDIO_RATE)
        return c, p, b, SpectrumProcessor(c, FFT_SIZE, max_frames=n_frames_max)

    ctx, post, bank, spec = make_pipeline()

    def step():
        post.execute(ring, NB, BLOCK, CENTER)
        bank.execute(post)
        spec.process(ring, NB, BLOCK, contiguous=True)

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    if not args.no_profile:
        ctx.profile_enable(PROFILE_PERIOD)
    ctx.timer_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ev_ms = ctx.timer_stop()
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        dist.barrier()
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    prof = {}
    if not args.no_profile:
        prof.update(ctx.profile())
        ctx.profile_enable(False)
    audio_total = bank.total_audio()

    samples = args.steps * NB * BLOCK * world
    value = samples / elapsed / 1e6
    out = {
        "metric": "IQ MS/s sustained @ N demods + FFT size",
        "value": value, "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "blocks_per_step": NB, "block_len": BLOCK, "n_demods": N_DEMODS, "fft_size": FFT_SIZE,
                   "realtime_multiple": value / world / (FS / 1e6), "audio_samples_per_step": audio_total,
                   "event_ms_per_step": ev_ms / args.steps, "parallelism": "one independent IQ stream per GPU; one HIP stream per pipeline stage, consecutive batches overlap"},
    }
    if prof:
        dom = max(prof, key=lambda k: prof[k][0])
        ms, launches = prof[dom]
        avg_ms = ms / launches
        units = NB * BLOCK                      # input samples one launch covers
        bps = algorithmic_bytes_per_sample(dom, N_DEMODS, M, 2 * FFT_SIZE)
        achieved = bps * units / (avg_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS,
                           "traffic": (None if args.config != "C2" or measured_traffic_bytes(dom) is None else measured_traffic_bytes(dom) * NB),
                           "traffic_unit": "bytes per launch (committed PMC pass, per IQ block, times the blocks of this launch)",
                           "avg_launch_ms": avg_ms,
                           "algorithmic_bytes_per_launch": bps * units,
                           "whole_path": {"bytes_per_sample": round(bytes_per_sample, 1), "achieved": bytes_per_sample * value / world * 1e6 / 1e9,
                                          "frac": bytes_per_sample * value / world * 1e6 / 1e9 / HBM_PEAK_GBS},
                           "profile_sampling": "HIP events around every %d-th launch of each kernel inside the timed region" % PROFILE_PERIOD,
                           "kernels_ms_per_step": {k: v[0] * PROFILE_PERIOD / args.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}
    spec.close(); bank.close(); post.close(); ctx.close()
    if prof and rank == 0:
        # The live durations above include whatever the other stream's kernels took from the GPU at that moment (the two chains
        # overlap by design), so they move with the phase between the chains.  For a kernel-quality figure the same batch is run
        # once more, untimed, on ONE stream: every kernel alone on the device.
        os.environ["CSDR_STREAMS"] = "1"
        ctx1, post1, bank1, spec1 = make_pipeline()
        os.environ.pop("CSDR_STREAMS", None)
        for it in range(13):
            if it == 3:
                ctx1.synchronize(); ctx1.profile_enable(1)
            post1.execute(ring, NB, BLOCK, CENTER); bank1.execute(post1); spec1.process(ring, NB, BLOCK, contiguous=True)
        ctx1.synchronize()
        solo = {k: v[0] / v[1] for k, v in ctx1.profile().items()}
        dom = out["roofline"]["kernel"]
        alg = out["roofline"]["algorithmic_bytes_per_launch"]
        out["roofline"]["solo"] = {"note": "same batch, one stream, nothing else on the GPU; 10 launches", "avg_launch_ms": solo[dom],
                                   "achieved": alg / (solo[dom] * 1e-3) / 1e9, "frac": alg / (solo[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "kernels_ms_per_launch": {k: v for k, v in sorted(solo.items(), key=lambda kv: -kv[1])}}
        spec1.close(); bank1.close(); post1.close(); ctx1.close()
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        try:
            out["cpu_baseline"] = cpu_baseline(ring.cpu().numpy().view("complex64").reshape(-1), args.cpu_seconds)
        except Exception as e:  # the baseline is reported, never required for the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "MS/s", "cores": 0, "kind": "unavailable", "sample": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
