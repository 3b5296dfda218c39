#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native CubicSDR hot path (contract: see task statement / DESIGN.md).

Default workload = BASELINE.json configs[2] ("C3", the configuration the north-star target is quoted on): 256 mixed
NBFM / AM / USB demodulators, 61.44 MS/s complex-float IQ, firpfbch M = 122 (block = 1 024 068 samples, channel rate
503 606 S/s), 65536-point spectrum FFT (internal 131072), every sample FFT'ed ("contiguous" frames).
`--config C3N` is the same with 256 NBFM demodulators (the north-star sentence's wording), `--config C2` the 64-NBFM /
10 MS/s / 16384-point case.

One step = `--batches` consecutive passes of the whole hot path (channelizer + demodulator chains + spectrum), each over
one batch of `--blocks` consecutive IQ blocks that are already resident in HBM; the stream state (filter histories,
oscillators, averagers) carries on from batch to batch.  The defaults make one step ~0.1-0.2 s of GPU work so that the
driver's `--steps 20` is a multi-second timed region.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Multi-GPU: each rank owns an independent IQ stream with its own demodulators (BASELINE config 5 style partitioning:
no data-path collective); value = samples processed by all ranks / max-over-ranks time; scaling = weak.
`--config C4 --gpus N` is the demodulator-sharded one-stream case (see cubicsdr_amd/parallel.py).
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
PROFILE_PERIOD = 32       # per-kernel HIP events bracket every 32nd launch of each kernel id: 30 samples per kernel over the default timed region (bracketing every
                          # launch costs ~7 % of the throughput, every 8th still 4.5 %: an event pair is two barrier packets between back-to-back kernels)
CENTER = 100_000_000
AUDIO_RATE = 48_000
MODEM_BW = {"NBFM": 12_500, "AM": 6_000, "USB": 5_400}
MODEM_ID = {"NBFM": 0, "AM": 2, "USB": 3}

# fs, M, block (SoapySDRThread.cpp:668-693: numChannels = even floor of ceil(fs / 500 kHz), block = ceil(fs / 60 / M) M), demods,
# fftSize, modem round-robin, IQ blocks per batch, batches per step
CONFIGS = {
    "C2": dict(fs=10_000_000, M=20, block=166_680, n_demods=64, fft=16384, kinds=["NBFM"], blocks=256, batches=128,
               label="C2: 64x NBFM demods (12.5 kHz -> 48 kHz audio), 10 MS/s complex-float IQ, firpfbch M=20, 16384-pt spectrum FFT (internal 32768) over every sample"),
    "C3": dict(fs=61_440_000, M=122, block=1_024_068, n_demods=256, fft=65536, kinds=["NBFM", "AM", "USB"], blocks=128, batches=48,
               label="C3: 256 mixed NBFM/AM/USB demods, 61.44 MS/s complex-float IQ, firpfbch M=122, 65536-pt spectrum FFT (internal 131072) over every sample"),
    "C5": dict(fs=100_000_000, M=200, block=1_666_800, n_demods=512, fft=1_048_576, kinds=["NBFM", "AM", "USB"], blocks=32, batches=24,
               label="C5: one 100 MS/s complex-float IQ stream per GPU (BASELINE config 5: replicas, no cross-GPU traffic), firpfbch M=200, 512 mixed NBFM/AM/USB demods, "
                     "1048576-pt spectrum FFT (internal 2097152) over every sample"),
    "C3N": dict(fs=61_440_000, M=122, block=1_024_068, n_demods=256, fft=65536, kinds=["NBFM"], blocks=128, batches=48,
                label="C3N: 256 NBFM demods, 61.44 MS/s complex-float IQ, firpfbch M=122, 65536-pt spectrum FFT (internal 131072) over every sample"),
}


def demod_frequencies(center, fs, n):
    return [int(center + (k + 0.37) * fs / n - fs / 2) for k in range(n)]


def make_ring(torch, device, cfg, n_blocks, seed):
    """synthetic IQ ring in HBM: noise sigma 0.05 + one modulated carrier per demod (NBFM: 1 kHz tone, 2.5 kHz
    deviation; AM: 80 %; USB: single tone at +1 kHz) + DC offset.  Generated in slices so the temporaries stay small."""
    fs, n_demods, kinds = cfg["fs"], cfg["n_demods"], cfg["kinds"]
    n = n_blocks * cfg["block"]
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.randn(n, 2, generator=g, device=device, dtype=torch.float32) * 0.05
    amp = 0.5 / math.sqrt(n_demods)
    freqs = demod_frequencies(CENTER, fs, n_demods)
    SL = 1 << 22
    for s0 in range(0, n, SL):
        s1 = min(n, s0 + SL)
        t = torch.arange(s0, s1, device=device, dtype=torch.float64) / fs
        tone = torch.sin(2 * math.pi * 1000.0 * t)
        acc_r = torch.zeros(s1 - s0, device=device, dtype=torch.float32)
        acc_i = torch.zeros(s1 - s0, device=device, dtype=torch.float32)
        for i, f in enumerate(freqs):
            kind = kinds[i % len(kinds)]
            df = float(f - CENTER)
            if kind == "NBFM":
                ph = torch.remainder((2 * math.pi * df) * t + 2.5 * tone, 2 * math.pi).float()
                acc_r += amp * torch.cos(ph); acc_i += amp * torch.sin(ph)
            elif kind == "AM":
                ph = torch.remainder((2 * math.pi * df) * t, 2 * math.pi).float()
                env = (amp * (1 + 0.8 * tone)).float()
                acc_r += env * torch.cos(ph); acc_i += env * torch.sin(ph)
            else:
                ph = torch.remainder((2 * math.pi * (df + 1000.0)) * t, 2 * math.pi).float()
                acc_r += amp * torch.cos(ph); acc_i += amp * torch.sin(ph)
        x[s0:s1, 0] += acc_r + 0.01
        x[s0:s1, 1] += acc_i + 0.01
    return x.contiguous()


# algorithmic HBM bytes per INPUT SAMPLE attributed to each kernel (DESIGN.md "Roofline accounting"):
#   ingest read 8 + channelizer write 8 ; demod reads 8 N/M + audio writes ; spectrum read 8 + display write 4
def cascade_depth(bw, chan_rate):
    """half-band stages of the decimating msresamp for bandwidth / channel rate (liquid msresamp: while (r < 0.5) { S++; r *= 2 })"""
    r, S = float(bw) / float(chan_rate), 0
    while r < 0.5:
        S += 1
        r *= 2.0
    return S


MERGED_56 = None     # set from the kernel ids the library really launched (main); None: the library's rule with MI355X's 1024 resident workgroups


def frontend_groups(cfg):
    """demodulators per front-end kernel instance (one launch per cascade depth, csdr_bank.hip: csdr_bank_execute)"""
    out = {}
    for i in range(cfg["n_demods"]):
        kind = cfg["kinds"][i % len(cfg["kinds"])]
        S = cascade_depth(MODEM_BW[kind], cfg["fs"] // cfg["M"])
        name = "demod_frontend_s%d" % S if 3 <= S <= 6 else "demod_frontend_generic"
        out[name] = out.get(name, 0) + 1
    # depths 5 and 6 share one launch (demod_frontend_s56) when together they still get three ranges per demodulator (csdr_bank_execute: 4 (n5 + n6)
    # <= resident workgroups, 4 per CU x 256 CUs on MI355X)
    n5, n6 = out.get("demod_frontend_s5", 0), out.get("demod_frontend_s6", 0)
    if n5 and n6 and (MERGED_56 if MERGED_56 is not None else 4 * (n5 + n6) <= 1024):
        out["demod_frontend_s56"] = n5 + n6
        del out["demod_frontend_s5"], out["demod_frontend_s6"]
    return out


STAGE_OF = {"chan_analyze": "channelizer", "dc_tile_ends": "channelizer", "dc_apply": "channelizer",
            "demod_modem": "modem+audio", "demod_gain_scan": "modem+audio", "demod_audio_interp": "modem+audio", "fms_stages": "modem+audio", "fms_out": "modem+audio",
            "spec_fft_radix": "spectrum", "spec_fft_rows": "spectrum", "spec_average": "spectrum", "spec_extrema": "spectrum",
            "spec_display": "spectrum", "spec_misc": "spectrum"}


def stage_of(kernel):
    return "front-end" if kernel.startswith("demod_frontend") else STAGE_OF.get(kernel, kernel)


def algorithmic_bytes_per_sample(kernel, cfg):
    audio = 4.0 * cfg["n_demods"] * AUDIO_RATE / cfg["fs"]
    table = {
        "chan_analyze": 16.0,
        "demod_modem": 0.0,
        "demod_audio_interp": audio,
        "spec_fft_radix": 8.0,         # the frame is read once from HBM ...
        "spec_fft_rows": 0.0,          # ... later passes re-read intermediates that are not algorithmic traffic
        "spec_average": 0.0,
        "spec_display": 4.0,
    }
    if kernel.startswith("demod_frontend"):
        return 8.0 * frontend_groups(cfg).get(kernel, 0) / cfg["M"]          # each demodulator reads its channel once
    return table.get(kernel, 0.0)


def whole_path_bytes_per_sample(cfg):                      # 54.8 (C2), 45.6 (C3)
    return 8 + 8 + 8.0 * cfg["n_demods"] / cfg["M"] + 4.0 * cfg["n_demods"] * AUDIO_RATE / cfg["fs"] + 12


def measured_traffic(cfg_name):
    """-> (dict kernel -> HBM bytes per IQ block, file) from the latest committed PMC pass of this configuration
    (profiles/collect.sh: FETCH_SIZE and WRITE_SIZE in their own rocprofv3 runs of this same command; gfx950 reports half of a
    wide coalesced read stream, MI355X_MICROARCH.md "HBM", so the read side is doubled)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_pmc_traffic.json" % cfg_name.lower())))
    if not files:
        return {}, None
    try:
        t = json.load(open(files[-1]))
    except Exception:
        return {}, None
    blocks = float(t.get("_meta", {}).get("blocks_per_launch", 64))
    out = {}
    for name, v in t.items():
        if "FETCH_SIZE_KiB_avg_per_launch" in v and "WRITE_SIZE_KiB_avg_per_launch" in v:
            base = name.split("<")[0]
            if base == "demod_frontend_s" and "<" in name:
                base = "demod_frontend_s" + name.split("<")[1].split(",")[0].strip()          # demod_frontend_s<6, 2048, true> -> demod_frontend_s6
            if base == "demod_frontend_s56" or name.startswith("demod_frontend_s56"):
                base = "demod_frontend_s56"
            # (profile ids of the library, common.hpp CsdrKernelId: the fused spectrum chain's kernels run under the ids of the stages they replace)
            base = {"demod_frontend": "demod_frontend_generic", "spec_fft_rows4096": "spec_fft_rows", "chan_analyze_p2": "chan_analyze", "chan_analyze_fft": "chan_analyze",
                    "spec_cols512p": "spec_fft_radix", "spec_rows256_ema": "spec_average", "spec_display_p256": "spec_display"}.get(base, base)
            out[base] = out.get(base, 0.0) + (2.0 * v["FETCH_SIZE_KiB_avg_per_launch"] + v["WRITE_SIZE_KiB_avg_per_launch"]) * 1024.0 / blocks
    return out, os.path.relpath(files[-1], ROOT)


def cpu_baseline(cfg, ring_host, target_seconds):
    """time the reference CPU path on a bounded sample of the same workload: single thread, then thread-per-stage"""
    import numpy as np
    import oracle.liquid_api as A
    kind = "reference" if A.available("ref") else "port"
    L = A.load("ref" if kind == "reference" else "port")
    L.oracle_chain_create.restype = C.c_void_p
    L.oracle_chain_create.argtypes = [C.c_int] * 3 + [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int]
    L.oracle_chain_run.restype = C.c_double
    L.oracle_chain_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    fs, M, BLOCK, n_demods, kinds = cfg["fs"], cfg["M"], cfg["block"], cfg["n_demods"], cfg["kinds"]
    chan_bw = fs // M
    centers = [CENTER + chan_bw * i for i in range(M // 2)] + [CENTER - fs // 2 + chan_bw * i for i in range(M // 2)] + [CENTER + fs // 2]
    ch, nf, md, kd, iqr, aur = [], [], [], [], [], []
    for i, f in enumerate(demod_frequencies(CENTER, fs, n_demods)):
        k = min(range(M + 1), key=lambda q: (abs(f - centers[q]), q))
        shift = f - centers[k]
        name = kinds[i % len(kinds)]
        ch.append(k)
        nf.append(np.float32(2.0 * math.pi * abs(shift) / chan_bw))
        md.append(0 if shift == 0 else (1 if shift < 0 else -1))
        kd.append(MODEM_ID[name])
        iqr.append(np.float32(float(MODEM_BW[name]) / chan_bw))
        aur.append(np.float32(float(AUDIO_RATE) / MODEM_BW[name]))
    ch = np.array(ch, np.int32); nf = np.array(nf, np.float32); md = np.array(md, np.int32); kd = np.array(kd, np.int32)
    iqr = np.array(iqr, np.float32); aur = np.array(aur, np.float32)
    ring_blocks = ring_host.size // BLOCK
    t = np.zeros(3)
    na = C.c_longlong()
    p = lambda a: a.ctypes.data_as(C.c_void_p)

    ncpu = os.cpu_count() or 1
    # the DSP objects are built once (liquid designs two 3585-tap Kaiser prototypes per demodulator: ~0.3 s each) and kept
    h = C.c_void_p(L.oracle_chain_create(M, BLOCK, ring_blocks, p(ring_host), n_demods, p(ch), p(nf), p(md), p(kd), p(iqr), p(aur), 2 * cfg["fft"], ncpu))

    def run(nb, nthreads):
        return L.oracle_chain_run(h, nb, nthreads, p(t), C.byref(na))
    nprobe = 2 if BLOCK > 500_000 else 8
    probe = run(nprobe, 1)
    nb = max(nprobe, int(target_seconds / (probe / nprobe)))
    T = run(nb, 1)
    out = {"value": nb * BLOCK / T / 1e6, "unit": "MS/s", "cores": 1, "kind": kind,
           "sample": "%d blocks x %d samples of the same %s workload (%d demodulator chains + M=%d firpfbch + contiguous %d-pt FFT frames), %.1f s single thread: channelizer %.0f%%, demodulators %.0f%%, spectrum %.0f%%"
                     % (nb, BLOCK, cfg["name"], n_demods, M, 2 * cfg["fft"], T, 100 * t[0] / T, 100 * t[1] / T, 100 * t[2] / T)}
    if ncpu >= 3:
        nb2 = max(nprobe, int(nb * min(ncpu - 1, 8) * 0.6))
        T2 = run(nb2, ncpu)
        out["threaded"] = {"value": nb2 * BLOCK / T2 / 1e6, "unit": "MS/s", "cores": ncpu, "nproc": ncpu,
                           "sample": "%d blocks, %.1f s, thread-per-stage as CubicSDR runs it: 1 SDRPostThread + 1 spectrum thread + %d demodulator threads sharing the %d demodulators"
                                     % (nb2, T2, ncpu - 2, n_demods)}
    return out


def parity_sample(cfg, ring_host, make_pipeline, n_check=12, n_blocks=2):
    """Part of the cpu_baseline leg (the oracle as the CHECKER, never the thing measured): the first blocks of the bench's own ring through a
    fresh pipeline of the bench's own configuration as ONE batch, against the reference chain (oracle/cubicsdr_chain.py on the reference's
    liquid binary where it travelled) -- a spread of the demodulators (every modem kind of the configuration; resampled IQ, audio, counts)
    and the first spectrum frames.  Error metric as in the tests: max |gpu - reference| / peak |reference| per compared array; bound 1e-5."""
    import numpy as np
    import oracle.liquid_api as A
    from oracle.cubicsdr_chain import RefDemod, RefSDRPost, RefSpectrum
    be = "ref" if A.available("ref") else "port"
    fs, M, BLOCK, n_demods, kinds, F = cfg["fs"], cfg["M"], cfg["block"], cfg["n_demods"], cfg["kinds"], cfg["fft"]
    x = np.ascontiguousarray(ring_host[: n_blocks * BLOCK])
    c, post, bank, spec = make_pipeline(n_blocks)
    post.execute(x, n_blocks, BLOCK, CENTER); bank.execute(post)
    nfr = spec.process(x, n_blocks, BLOCK, contiguous=True)
    rp = RefSDRPost(be, fs, M)
    rp.frequency = CENTER; rp.update_channels()
    freqs = demod_frequencies(CENTER, fs, n_demods)
    pick = [i for i in sorted({(j * n_demods) // n_check + (j % len(kinds)) for j in range(n_check)}) if i < n_demods and rp.channel_at(freqs[i]) != 0]
    rds = {i: RefDemod(be, kinds[i % len(kinds)], MODEM_BW[kinds[i % len(kinds)]], freqs[i], rp.chan_bw, AUDIO_RATE) for i in pick}
    want = {i: {"iq": [], "audio": []} for i in pick}
    for b in range(n_blocks):
        rp.run_block(x[b * BLOCK:(b + 1) * BLOCK], CENTER)
        cache = {}
        for i, rd in rds.items():
            ch = rp.channel_at(rd.frequency)
            if ch not in cache:
                cache[ch] = rp.channel_data(ch)
            riq = rd.pre(*cache[ch])
            want[i]["iq"].append(riq); want[i]["audio"].append(rd.demodulate(riq)["audio"])
    rel = lambda g, w: float(np.max(np.abs(g - w)) / np.max(np.abs(w))) if w.size and g.size == w.size else float("inf")
    e_iq = e_au = e_sp = 0.0
    counts_exact = True
    for i in pick:
        wi, wa = np.concatenate(want[i]["iq"]), np.concatenate(want[i]["audio"])
        res = bank.results(i)
        counts_exact &= [r.n_iq for r in res] == [w.size for w in want[i]["iq"]] and [r.n_audio for r in res] == [w.size for w in want[i]["audio"]]
        e_iq = max(e_iq, rel(bank.iq(i), wi)); e_au = max(e_au, rel(bank.audio(i), wa))
    rs = RefSpectrum(be, F)
    nchk = min(nfr, 3)
    for k in range(nchk):
        wp = rs.process_frame(x[k * 2 * F:(k + 1) * 2 * F])[0]
        e_sp = max(e_sp, rel(spec.fetch(k)[0], wp))
    spec.close(); bank.close(); post.close(); c.close()
    return {"oracle": "reference liquid binary" if be == "ref" else "C restatement", "tolerance": 1e-5, "metric": "max|gpu-ref| / peak|ref| per array",
            "blocks": n_blocks, "demodulators_checked": len(pick), "kinds": sorted({kinds[i % len(kinds)] for i in pick}), "counts_exact": bool(counts_exact),
            "iq": e_iq, "audio": e_au, "spectrum_frames": nchk, "spectrum": e_sp, "ok": bool(counts_exact and max(e_iq, e_au, e_sp) < 1e-5),
            "full_suite": "tests/test_gpu_parity.py compares every demodulator x 3 blocks and ~1000 spectrum frames of this configuration (DESIGN.md 2)"}


class Sensors:
    """core / memory clock, package power and junction temperature of ONE GPU, read from the amdgpu hwmon files of its PCI device
    (/sys/bus/pci/devices/<bdf>/hwmon/hwmon*/{freq1,freq2,power1,temp2}_input) by a sampling thread while the timed region runs: the
    kernels on this path are bound by instruction issue, so their durations follow the core clock -- which differs from box to box and
    ramps for ~40 ms after the GPU was idle (profiles/r05_variance.txt).  Best effort: {} where the files do not exist."""

    def __init__(self, device_index, period_s=0.1):
        import glob
        self.files, self.samples, self.period, self._stop, self._th = {}, [], period_s, False, None
        try:
            hip = C.CDLL("libamdhip64.so")
            buf = C.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) != 0:
                return
            bdf = buf.value.decode().lower()
            for d in glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf):
                for key, name in (("sclk_MHz", "freq1_input"), ("mclk_MHz", "freq2_input"), ("power_W", "power1_input"), ("junction_C", "temp2_input")):
                    f = os.path.join(d, name)
                    if os.path.exists(f):
                        self.files[key] = f
        except Exception:
            self.files = {}

    def read(self):
        out = {}
        for key, f in self.files.items():
            try:
                v = float(open(f).read().strip())
                out[key] = v / 1e6 if key != "junction_C" else v / 1e3
            except Exception:
                pass
        return out

    def start(self):
        if not self.files:
            return
        import threading

        def loop():
            while not self._stop:
                self.samples.append(self.read())
                time.sleep(self.period)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()

    def stop(self):
        """-> {sensor: {min, median, max}} over the samples taken since start()"""
        self._stop = True
        if self._th:
            self._th.join()
        out = {"samples": len(self.samples)}
        for key in self.files:
            v = sorted(x[key] for x in self.samples if key in x)
            if v:
                out[key] = {"min": round(v[0], 1), "median": round(v[len(v) // 2], 1), "max": round(v[-1], 1)}
        return out


def spread(v):
    s = sorted(v)
    return {"n": len(s), "min": s[0], "median": s[len(s) // 2], "max": s[-1]} if s else {}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line under torch.distributed.run, one rank per GPU of this node
    (rendezvous on 127.0.0.1, a free port); rank 0 prints the one JSON line.  Never returns."""
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries under this process write there too (RCCL's version banner at communicator creation, gloo's
    connection notes in the one-GPU dry run): file descriptor 1 is pointed at stderr for the whole run and the line goes to the descriptor that was
    stdout.  Returns the stream to print the line on."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(keep, "w")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS) + ["C4"],
                    help="BASELINE.json workload: C3 (default: 256 mixed demods, 61.44 MS/s, M=122, 65536-pt FFT -- the configuration the target is quoted on), "
                         "C3N (same, all NBFM), C2 (64 NBFM, 10 MS/s, M=20, 16384-pt), C4 (M=1024 channelizer + 1024 NBFM, demodulators sharded over the ranks), "
                         "C5 (100 MS/s, M=200, 512 mixed demods, 2^20-pt FFT: with --gpus N one such stream per GPU)")
    ap.add_argument("--blocks", type=int, default=0, help="IQ blocks per batch (HBM-resident ring); default per config")
    ap.add_argument("--batches", type=int, default=0, help="batches per step; default per config (a step is ~0.1-0.2 s of GPU work)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline sample budget per leg (0 disables)")
    ap.add_argument("--shard", default="auto", choices=["auto", "broadcast", "slab"],
                    help="--config C4 only: how the ONE stream is spread over the GPUs (broadcast + channel subsets, or time slabs + all-to-all); auto: "
                         "parallel.strong_scaling_plan picks by the link model for this number of GPUs")
    ap.add_argument("--ingest", default="distributed", choices=["distributed", "rank0"],
                    help="--config C4 --shard slab: every rank holds its own time slab (its own reader; here the same synthetic ring on every rank) or "
                         "rank 0 holds the stream and scatters windows over xGMI")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false",
                    help="--config C4 --shard slab: row exchange of a batch NOT overlapped with the next batch's channelizer (the one-call form)")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2, 3, 5],
                    help="physical HIP streams the stages of the TIMED pipeline are folded onto (default 1: every kernel runs alone, so the live "
                         "per-kernel durations and roofline.frac are the kernel's own; the library's default folding is 3, measured next to it)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event profile")
    ap.add_argument("--no-latency", action="store_true", help="skip the small-batch (real-time shape) measurement")
    ap.add_argument("--ring", default="signal", choices=["signal", "noise"],
                    help="signal: noise + one modulated carrier per demodulator + DC (default, SURVEY.md 8d); noise: the noise and DC only "
                         "(counter-collection passes: rocprofv3 --pmc does not survive the thousands of small torch launches of the synthesis)")
    ap.add_argument("--no-strong", action="store_true", help="--gpus N > 1, default configuration: skip the strong-scaling leg (C4, time slabs) that follows the timed region")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                         # (does not return)
    line_out = claim_stdout()
    if args.config == "C4":
        from cubicsdr_amd import sharded_bench
        return sharded_bench.main(args, line_out)

    cfg = dict(CONFIGS[args.config]); cfg["name"] = args.config
    FS, M, BLOCK, N_DEMODS, FFT_SIZE, kinds = cfg["fs"], cfg["M"], cfg["block"], cfg["n_demods"], cfg["fft"], cfg["kinds"]
    NB = args.blocks or cfg["blocks"]
    NBATCH = args.batches or cfg["batches"]
    bytes_per_sample = whole_path_bytes_per_sample(cfg)
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
    local_rank = local_rank % torch.cuda.device_count()          # (more ranks than GPUs: the one-GPU dry run of the multi-rank control flow, CSDR_DIST_BACKEND=gloo)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(os.environ.get("CSDR_DIST_BACKEND", "nccl"), rank=rank, world_size=world)   # ("nccl" is RCCL; the override exists for one-GPU dry runs of the multi-rank control flow)
        dist.barrier()
    else:
        torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    sensors = Sensors(local_rank)

    from cubicsdr_amd.engine import Context, DemodBank, SDRPost, SpectrumProcessor
    if args.ring == "noise":
        g0 = torch.Generator(device=device); g0.manual_seed(0xC0B1C5D2 + rank)
        ring = torch.randn(NB * BLOCK, 2, generator=g0, device=device, dtype=torch.float32) * 0.05 + 0.01
    else:
        ring = make_ring(torch, device, cfg, NB, seed=0xC0B1C5D2 + rank)
    torch.cuda.synchronize()

    def make_pipeline(nb):
        c = Context(local_rank)        # one HIP stream per pipeline stage inside (include/csdr_hip.h "Streams")
        p = SDRPost(c, FS, M, BLOCK, max_blocks=nb)
        b = DemodBank(c, N_DEMODS, max_blocks=nb)
        for i, f in enumerate(demod_frequencies(CENTER, FS, N_DEMODS)):
            kind = kinds[i % len(kinds)]
            b.configure(i, p, kind, MODEM_BW[kind], f, AUDIO_RATE)
        return c, p, b, SpectrumProcessor(c, FFT_SIZE, max_frames=(nb * BLOCK) // (2 * FFT_SIZE) + 2)

    streams = int(os.environ.get("CSDR_STREAMS", args.streams))      # (an explicit environment setting wins)
    env_had = os.environ.get("CSDR_STREAMS")
    os.environ["CSDR_STREAMS"] = str(streams)
    ctx, post, bank, spec = make_pipeline(NB)
    if env_had is None:
        os.environ.pop("CSDR_STREAMS", None)

    def batch(p=post, b=bank, s=spec, nb=NB, x=ring):
        p.execute(x, nb, BLOCK, CENTER)
        b.execute(p)
        s.process(x, nb, BLOCK, contiguous=True)

    def step():
        for _ in range(NBATCH):
            batch()

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    if not args.no_profile:
        ctx.profile_enable(PROFILE_PERIOD)
    sensors.start()
    ctx.timer_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ev_ms = ctx.timer_stop()
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    clocks = sensors.stop()
    if dist:
        dist.barrier()
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    prof, prof_rng = {}, {}
    if not args.no_profile:
        prof.update(ctx.profile())
        prof_rng = ctx.profile_range()
        ctx.profile_enable(False)
    audio_total = bank.total_audio()
    # the spread behind the mean: the same steps once more, each synchronised and timed on its own (after the timed region, which runs unbroken)
    step_ms = []
    for _ in range(min(args.steps, 10)):
        ts = time.perf_counter()
        step()
        ctx.synchronize()
        step_ms.append(1e3 * (time.perf_counter() - ts))

    samples = args.steps * NBATCH * NB * BLOCK * world
    value = samples / elapsed / 1e6
    out = {
        "metric": "IQ MS/s sustained @ N demods + FFT size",
        "value": value, "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["label"], "batches_per_step": NBATCH, "blocks_per_batch": NB, "block_len": BLOCK,
                   "samples_per_step": NBATCH * NB * BLOCK, "n_demods": N_DEMODS, "fft_size": FFT_SIZE,
                   "realtime_multiple": value / world / (FS / 1e6), "audio_samples_per_batch": audio_total,
                   "timed_region_s": elapsed, "event_ms_per_step": ev_ms / args.steps,
                   "error_metric": "parity tests hold |gpu - reference| <= 1e-5 of the reference's peak magnitude per compared array (tests/util.py rel_err); integer items bit-exact",
                   "streams": streams,
                   "step_ms": dict(spread(step_ms), note="the same step run again after the timed region, each synchronised on its own"),
                   "clocks": dict(clocks, note="amdgpu hwmon of this GPU sampled every 0.1 s inside the timed region (the kernels are issue-bound: their durations follow sclk, "
                                               "which differs from box to box and ramps for ~40 ms after idle: profiles/r05_variance.txt)"),
                   "parallelism": "one independent IQ stream per GPU; stages of the timed pipeline on %d HIP stream(s)" % streams},
    }
    if prof:
        units = NB * BLOCK                      # input samples one batch covers
        n_batches = args.steps * NBATCH
        # per kernel id: average launch duration (HIP events, every PROFILE_PERIOD-th launch) x the launches per batch the library
        # really made (all launches are counted, bracketed or not) = its time per batch
        global MERGED_56
        MERGED_56 = "demod_frontend_s56" in prof          # what the library launched, not a rule restated here
        avg = {k: v[0] / v[1] for k, v in prof.items()}
        per_batch = {k: avg[k] * (v[2] / n_batches) for k, v in prof.items()}
        dom = max(per_batch, key=lambda k: per_batch[k])
        co_dominant = [k for k in sorted(per_batch, key=lambda k: -per_batch[k]) if k != dom and per_batch[k] >= 0.97 * per_batch[dom]]
        avg_ms = avg[dom]
        launches_dom = prof[dom][2] / n_batches
        bps = algorithmic_bytes_per_sample(dom, cfg)
        alg_launch = bps * units / launches_dom
        achieved = alg_launch / (avg_ms * 1e-3) / 1e9
        traffic, traffic_file = measured_traffic(args.config)
        stages = {}
        for k in prof:
            st = stages.setdefault(stage_of(k), {"ms_per_batch": 0.0, "algorithmic_bytes_per_sample": 0.0, "kernels": []})
            st["ms_per_batch"] += per_batch[k]
            st["algorithmic_bytes_per_sample"] += algorithmic_bytes_per_sample(k, cfg)
            st["kernels"].append(k)
        for st in stages.values():
            st["achieved_GBps"] = st["algorithmic_bytes_per_sample"] * units / (st["ms_per_batch"] * 1e-3) / 1e9 if st["ms_per_batch"] > 0 else None
            st["frac"] = st["achieved_GBps"] / HBM_PEAK_GBS if st["achieved_GBps"] is not None else None
        # per kernel: time per batch, the fraction of the roofline by ALGORITHMIC bytes and by the bytes the counters saw it move
        # (`frac_traffic`: committed PMC pass / this run's live time -- lower than `frac` where cache hits serve algorithmic bytes, higher where
        # a kernel re-reads)
        kernels = {}
        for k in sorted(per_batch, key=lambda k: -per_batch[k]):
            t_s = per_batch[k] * 1e-3
            ab = algorithmic_bytes_per_sample(k, cfg)
            kernels[k] = {"ms_per_batch": per_batch[k], "algorithmic_bytes_per_sample": ab,
                          "frac": (ab * units / t_s / 1e9 / HBM_PEAK_GBS) if t_s > 0 and ab > 0 else None,
                          "traffic_bytes_per_sample": (traffic[k] / BLOCK if k in traffic else None),
                          "frac_traffic": (traffic[k] * NB / t_s / 1e9 / HBM_PEAK_GBS) if t_s > 0 and k in traffic else None}
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS,
                           "frac_traffic": kernels[dom]["frac_traffic"],
                           "co_dominant": {k: {"ms_per_batch": per_batch[k], "frac": kernels[k]["frac"], "frac_traffic": kernels[k]["frac_traffic"]} for k in co_dominant},
                           "dominant_rule": "largest time per batch by this run's HIP events (agrees with rocprofv3: profiles/); kernels within 3 % of it are listed as co_dominant",
                           "traffic": (traffic[dom] * NB / launches_dom if dom in traffic else None),
                           "traffic_unit": "HBM bytes per launch of the dominant kernel (PMC pass %s: per IQ block, times the blocks of this launch)" % traffic_file,
                           "traffic_source": "NOT measured in this run: read from the committed rocprofv3 counter pass %s (profiles/collect.sh: FETCH_SIZE x 2 + WRITE_SIZE, own passes, noise ring)" % traffic_file,
                           "kernels": kernels,
                           "avg_launch_ms": avg_ms, "launch_ms_range": list(prof_rng.get(dom, (None, None))), "launches_per_batch": launches_dom,
                           "algorithmic_bytes_per_launch": alg_launch,
                           "whole_path": {"bytes_per_sample": round(bytes_per_sample, 1), "achieved": bytes_per_sample * value / world * 1e6 / 1e9,
                                          "frac": bytes_per_sample * value / world * 1e6 / 1e9 / HBM_PEAK_GBS,
                                          "traffic_bytes_per_sample": (sum(traffic.values()) / BLOCK if traffic else None),
                                          "sum_of_kernel_ms_per_batch": sum(per_batch.values()), "ms_per_batch": 1e3 * elapsed / n_batches},
                           "profile_sampling": "HIP events around every %d-th launch of EACH kernel id (one id per template instance) inside the timed region; "
                                               "ms_per_batch = average launch x launches per batch counted by the library" % PROFILE_PERIOD,
                           "concurrency": ("one stream: every kernel runs alone, the live duration is the kernel's own" if streams == 1 else
                                           "%d streams: other kernels share the GPU during a launch of the dominant kernel, so this live duration is longer than the kernel's own (roofline.solo: the same batch on one stream)" % streams),
                           "stages": {k: v for k, v in sorted(stages.items(), key=lambda kv: -kv[1]["ms_per_batch"])},
                           "kernels_avg_launch_ms": {k: avg[k] for k in sorted(avg, key=lambda k: -per_batch[k])},
                           "kernels_ms_per_batch": {k: per_batch[k] for k in sorted(avg, key=lambda k: -per_batch[k])}}
    spec.close(); bank.close(); post.close(); ctx.close()
    if prof and rank == 0 and streams != 1:
        # The live durations above include whatever the other stream's kernels took from the GPU at that moment (the two chains
        # overlap by design), so they move with the phase between the chains.  For a kernel-quality figure the same batch is run
        # once more, untimed, on ONE stream: every kernel alone on the device.
        os.environ["CSDR_STREAMS"] = "1"
        ctx1, post1, bank1, spec1 = make_pipeline(NB)
        os.environ.pop("CSDR_STREAMS", None)
        for it in range(13):
            if it == 3:
                ctx1.synchronize(); ctx1.profile_enable(1)
            batch(post1, bank1, spec1)
        ctx1.synchronize()
        solo = {k: v[0] / v[1] for k, v in ctx1.profile().items()}
        dom = out["roofline"]["kernel"]
        alg = out["roofline"]["algorithmic_bytes_per_launch"]
        out["roofline"]["solo"] = {"note": "same batch, one stream, nothing else on the GPU; 10 launches", "avg_launch_ms": solo[dom],
                                   "achieved": alg / (solo[dom] * 1e-3) / 1e9, "frac": alg / (solo[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "kernels_ms_per_launch": {k: v for k, v in sorted(solo.items(), key=lambda kv: -kv[1])}}
        spec1.close(); bank1.close(); post1.close(); ctx1.close()
    if rank == 0 and world == 1 and not args.no_latency and streams != 3 and env_had is None:
        # the library's default folding (three streams: channelizer | demodulators | spectrum) on the same batches, untimed part of the run
        os.environ["CSDR_STREAMS"] = "3"
        c3, p3, b3, s3 = make_pipeline(NB)
        os.environ.pop("CSDR_STREAMS", None)
        for _ in range(NBATCH):
            batch(p3, b3, s3)
        c3.synchronize()
        t3 = time.perf_counter()
        n3 = max(2, min(args.steps, 8))
        for _ in range(n3 * NBATCH):
            batch(p3, b3, s3)
        c3.synchronize()
        dt3 = time.perf_counter() - t3
        out["config"]["library_default_streams"] = {"streams": 3, "MS_per_s": n3 * NBATCH * NB * BLOCK / dt3 / 1e6, "steps": n3}
        s3.close(); b3.close(); p3.close(); c3.close()
    if rank == 0 and world == 1 and not args.no_latency:
        # the real-time shape: the reference hands ONE block (1/60 s of signal) per call (SoapySDRThread.cpp:12); small batches
        # through the same entry points, untimed part of the run, reported next to the throughput setting
        lat = {}
        for nb in (1, 4, 16):
            if nb >= NB:
                continue
            c2, p2, b2, s2 = make_pipeline(nb)
            calls = max(8, min(240, 960 // nb))
            sub = ring[: nb * BLOCK]
            for _ in range(4):
                batch(p2, b2, s2, nb, sub)
            c2.synchronize()
            tl = time.perf_counter()
            for _ in range(calls):
                batch(p2, b2, s2, nb, sub)
            c2.synchronize()
            dt = time.perf_counter() - tl
            lat[str(nb)] = {"MS_per_s": calls * nb * BLOCK / dt / 1e6, "blocks_per_s": calls * nb / dt, "ms_per_call": 1e3 * dt / calls,
                            "streams_at_60_blocks_per_s": calls * nb / dt / 60.0}
            if nb == 1:
                # the same one-block calls from TWO host threads, the reference's own thread cut (SDRPostThread drives the channelizer and the
                # demodulators, SpectrumVisualDataThread the spectrum: cubicsdr_amd/host/HipPipeline.h runs them that way): a one-block call is
                # bound by the host's enqueue time, which the two threads spend side by side (the C ABI calls release the GIL)
                import threading
                errs = []

                def demod_side():
                    try:
                        for _ in range(calls):
                            p2.execute(sub, 1, BLOCK, CENTER); b2.execute(p2)
                    except Exception as e:      # noqa: BLE001
                        errs.append(repr(e))

                def spec_side():
                    try:
                        for _ in range(calls):
                            s2.process(sub, 1, BLOCK, contiguous=True)
                    except Exception as e:      # noqa: BLE001
                        errs.append(repr(e))
                ta, tb = threading.Thread(target=demod_side), threading.Thread(target=spec_side)
                tl = time.perf_counter()
                ta.start(); tb.start(); ta.join(); tb.join()
                c2.synchronize()
                dt2 = time.perf_counter() - tl
                lat["1"]["two_host_threads"] = {"blocks_per_s": calls / dt2, "ms_per_call": 1e3 * dt2 / calls, "errors": errs,
                                                "note": "channelizer + demodulators on one host thread, spectrum on another (the reference's thread cut)"}
            s2.close(); b2.close(); p2.close(); c2.close()
        out["config"]["small_batches"] = lat
    if rank == 0 and world == 1 and not args.no_latency:
        # Host-fed rate (never `value`): the SAME pipeline when the IQ blocks start in page-locked HOST memory, as an SDR reader leaves them:
        # csdr_ingest moves each batch over the link ONCE (its own transfer stream, slot k + 1 in flight while slot k is processed) and the
        # channelizer and the spectrum both read that one HBM copy.  The slots are filled once, before the clock starts: a real reader's
        # DMA writes them; a host memcpy per batch would measure the host's copy loop instead of the link.
        try:
            from cubicsdr_amd.engine import Ingest
            nbh = min(16, NB)
            ch, ph, bh, sh = make_pipeline(nbh)
            ing = Ingest(ch, nbh * BLOCK, depth=3)
            src = ring[: nbh * BLOCK].cpu().numpy().view("complex64").reshape(-1)
            for _ in range(3):
                slot = ing.acquire(); slot[:src.size] = src; ing.commit(src.size)
            ch.synchronize()

            def host_batch():
                ing.acquire()
                dev = ing.commit(nbh * BLOCK)
                ph.execute(dev, nbh, BLOCK, CENTER); bh.execute(ph); sh.process(dev, nbh, BLOCK, contiguous=True)
            for _ in range(4):
                host_batch()
            ch.synchronize()
            th = time.perf_counter()
            nh = max(8, int(1.5e9 / (nbh * BLOCK)))                    # ~1.5 G samples
            for _ in range(nh):
                host_batch()
            ch.synchronize()
            dth = time.perf_counter() - th
            out["config"]["host_fed"] = {"MS_per_s": nh * nbh * BLOCK / dth / 1e6, "GB_per_s_over_the_link": nh * nbh * BLOCK * 8 / dth / 1e9, "blocks_per_call": nbh,
                                         "calls": nh, "note": "blocks in page-locked host memory -> csdr_ingest (ONE transfer per block, overlapped with compute) -> "
                                                              "channelizer + demodulators + spectrum all read the one HBM copy; PCIe-inclusive, never `value`"}
            ing.close(); sh.close(); bh.close(); ph.close(); ch.close()
        except Exception as e:
            out["config"]["host_fed"] = {"MS_per_s": None, "note": repr(e)}
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        try:
            ring_host = ring.cpu().numpy().view("complex64").reshape(-1)
            out["cpu_baseline"] = cpu_baseline(cfg, ring_host, args.cpu_seconds)
            if args.ring != "noise":
                try:
                    out["cpu_baseline"]["parity"] = parity_sample(cfg, ring_host, make_pipeline)
                except Exception as e:
                    out["cpu_baseline"]["parity"] = {"ok": None, "note": repr(e)}
        except Exception as e:  # the baseline is reported, never required for the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "MS/s", "cores": 0, "kind": "unavailable", "sample": repr(e)}
    if world > 1 and not args.no_strong:
        # Strong scaling next to the weak (replica) value: ONE 100 MS/s stream, M = 1024 channelizer + 1024 NBFM demodulators (BASELINE config 4),
        # time slabs scattered from rank 0 and channel rows exchanged all-to-all (SURVEY 8e option 2; cubicsdr_amd/sharded_bench.py).  Collective:
        # every rank runs it; a short, untimed-by-the-contract leg after the timed region.
        del ring
        torch.cuda.empty_cache()
        try:
            from cubicsdr_amd import sharded_bench
            sargs = argparse.Namespace(**vars(args))
            sargs.shard, sargs.blocks, sargs.batches, sargs.steps, sargs.warmup = "auto", 32, 6, max(2, min(args.steps, 5)), 1
            sargs.ingest, sargs.overlap = "distributed", True
            sargs.no_profile, sargs.cpu_seconds, sargs.streams = True, 0.0, 3
            so = sharded_bench.measure(sargs, dist=dist)
            out["strong"] = {"workload": so["config"]["workload"], "value": so["value"], "unit": "MS/s", "scaling": "strong", "n_gpus": world,
                             "ms_per_step": so["ms_per_step"], "steps": sargs.steps, "transport": so["config"]["transport"], "rccl_ranks": so["config"]["rccl_ranks"],
                             "note": "one stream over all ranks (total work fixed); `value` above is the replica (weak) figure of the default configuration"}
        except Exception as e:      # reported, never required for the headline number
            out["strong"] = {"value": None, "note": repr(e)}
    if rank == 0:
        line_out.write(json.dumps(out) + "\n"); line_out.flush()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
