#!/usr/bin/env python3
"""Generate tests/golden/liquid_1_5_0.npz from the reference's own liquid-dsp 1.5.0 binary.

Runs ONLY where /root/reference is mounted (it needs oracle/_ref/libliquid_ref.so + libliquid.dll, built by
`make -C oracle ref`).  The fixture pins, for every liquid function on CubicSDR's hot path, the output of
the reference binary on small seeded inputs; inputs are stored too, so the fixture is self-contained.
    python tests/golden/gen_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle.liquid_api as A  # noqa: E402
from oracle.cubicsdr_chain import RefDemod, RefSDRPost, RefSpectrum  # noqa: E402
from tests.util import demod_frequencies, synth_iq  # noqa: E402


def rnd(rng, n):
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.3).astype(np.complex64)


def main():
    L = A.load("ref")
    rng = np.random.default_rng(20260922)
    g = {}
    # --- design helpers
    for name, n, fc, As in [("kaiser33", 33, 0.1, 60.0), ("kaiser161", 161, 0.025, 60.0)]:
        h = np.zeros(n, np.float32)
        L.liquid_firdes_kaiser(n, fc, As, 0.0, A.ptr(h))
        g["firdes_" + name] = h
    h = np.zeros(51, np.float32)
    L.liquid_firdes_notch(25, 0.0, 30.0, A.ptr(h))
    g["firdes_notch_25_30"] = h
    g["estimate_req_filter_len"] = np.array([L.estimate_req_filter_len(df, As) for df, As in [(0.1, 60.0), (0.05, 65.0), (0.35, 65.0), (0.2, 65.0)]], np.int64)
    # half-band branch taps of the three designs CubicSDR uses (impulse through resamp2 interpolator, exact)
    for m in (3, 5, 10):
        q = L.resamp2_crcf_create(m, 0.0, 65.0)
        yy = np.zeros(8 * m + 4, np.complex64)
        for k in range(4 * m + 2):
            L.resamp2_crcf_interp_execute(q, A.cpx(1.0 if k == 0 else 0.0), A.ptr(yy[2 * k:2 * k + 2]))
        g["halfband_h1_m%d" % m] = yy.real[1::2][:2 * m].copy()
    # --- nco: phase words and mixing
    freqs = np.array([0.1, 1.234, 3.0, -0.5, 6.0, 2 * np.pi * 0.25, 2 * np.pi * (123456 / 500000)], np.float32)
    words, mixes = [], []
    x = rnd(rng, 600)
    g["nco_in"] = x
    for f in freqs:
        q = L.nco_crcf_create(A.LIQUID_VCO)
        L.nco_crcf_set_frequency(q, float(f))
        y = np.zeros_like(x)
        L.nco_crcf_mix_block_down(q, A.ptr(x), A.ptr(y), x.size)
        d = np.zeros(2, np.uint32)
        L.ref_peek(C.c_void_p(q), 0x1004, A.ptr(d), 8)
        words.append(d.copy()); mixes.append(y)
    g["nco_freqs"] = freqs; g["nco_words_after_600"] = np.array(words); g["nco_mix_down"] = np.array(mixes)
    # --- msresamp_crcf decimators, 3 blocks each (counts per block are the bit-exact item)
    for tag, r, bs in [("r0025", 12500 / 500000, 2500), ("r0119", 6000 / 503606, 2111), ("r0108", 5400 / 500000, 2500), ("r04", 0.4, 777), ("r06", 0.6, 500)]:
        q = L.msresamp_crcf_create(r, 60.0)
        xin = rnd(rng, 3 * bs)
        outs, cnts = [], []
        for b in range(3):
            xb = np.ascontiguousarray(xin[b * bs:(b + 1) * bs])
            y = np.zeros(bs + 600, np.complex64); ny = C.c_uint()
            L.msresamp_crcf_execute(q, A.ptr(xb), bs, A.ptr(y), C.byref(ny))
            outs.append(y[:ny.value].copy()); cnts.append(ny.value)
        g["msresamp_crcf_%s_in" % tag] = xin; g["msresamp_crcf_%s_out" % tag] = np.concatenate(outs)
        g["msresamp_crcf_%s_counts" % tag] = np.array(cnts, np.int64); g["msresamp_crcf_%s_rate" % tag] = np.float32(r)
        g["msresamp_crcf_%s_delay" % tag] = np.float32(L.msresamp_crcf_get_delay(q))
    for tag, r, bs in [("r384", 48000 / 12500, 120), ("r8", 8.0, 60), ("r889", 48000 / 5400, 55)]:
        q = L.msresamp_rrrf_create(r, 60.0)
        xin = rng.standard_normal(3 * bs).astype(np.float32)
        outs, cnts = [], []
        for b in range(3):
            xb = np.ascontiguousarray(xin[b * bs:(b + 1) * bs])
            y = np.zeros(int(bs * r) + 600, np.float32); ny = C.c_uint()
            L.msresamp_rrrf_execute(q, A.ptr(xb), bs, A.ptr(y), C.byref(ny))
            outs.append(y[:ny.value].copy()); cnts.append(ny.value)
        g["msresamp_rrrf_%s_in" % tag] = xin; g["msresamp_rrrf_%s_out" % tag] = np.concatenate(outs)
        g["msresamp_rrrf_%s_counts" % tag] = np.array(cnts, np.int64); g["msresamp_rrrf_%s_rate" % tag] = np.float32(r)
    # --- channelizer
    for M in (4, 20, 122):
        q = L.firpfbch_crcf_create_kaiser(A.LIQUID_ANALYZER, M, 4, 60.0)
        nf = 24
        xin = rnd(rng, M * nf); y = np.zeros(M * nf, np.complex64)
        L.oracle_firpfbch_analyzer_block(C.c_void_p(q), M, A.ptr(xin), nf, A.ptr(y))
        g["firpfbch_M%d_in" % M] = xin; g["firpfbch_M%d_out" % M] = y
    # --- iir dc blocker, lowpass; freqdem; AM fir; SSB chain; fft
    xin = rnd(rng, 1500) + np.complex64(0.02 + 0.01j)
    q = L.iirfilt_crcf_create_dc_blocker(0.0005); y = np.zeros_like(xin)
    L.iirfilt_crcf_execute_block(q, A.ptr(xin), xin.size, A.ptr(y))
    g["dcblock_in"] = xin; g["dcblock_out"] = y
    xin = rnd(rng, 600); q = L.iirfilt_crcf_create_lowpass(6, 0.25); y = np.zeros_like(xin)
    L.iirfilt_crcf_execute_block(q, A.ptr(xin), xin.size, A.ptr(y))
    g["butter6_in"] = xin; g["butter6_out"] = y
    xin = rnd(rng, 600); q = L.freqdem_create(0.5); yf = np.zeros(600, np.float32)
    L.freqdem_demodulate_block(q, A.ptr(xin), 600, A.ptr(yf))
    g["freqdem_in"] = xin; g["freqdem_out"] = yf
    xin = rnd(rng, 400); q = L.firfilt_rrrf_create_dc_blocker(25, 30.0); yf = np.zeros(400, np.float32)
    L.oracle_am_block(C.c_void_p(q), A.ptr(xin), 400, A.ptr(yf))
    g["am_in"] = xin; g["am_out"] = yf
    for usb in (1, 0):
        xin = rnd(rng, 400)
        nco = L.nco_crcf_create(A.LIQUID_NCO); L.nco_crcf_set_frequency(nco, float(np.float32(2 * np.pi * 0.25)))
        iir = L.iirfilt_crcf_create_lowpass(6, 0.25); hb = L.firhilbf_create(5, 90.0); yf = np.zeros(400, np.float32)
        L.oracle_ssb_block(C.c_void_p(nco), C.c_void_p(iir), C.c_void_p(hb), usb, A.ptr(xin), 400, A.ptr(yf))
        g["ssb_usb%d_in" % usb] = xin; g["ssb_usb%d_out" % usb] = yf
    for n in (8, 256, 4096):
        xin = rnd(rng, n); y = np.zeros(n, np.complex64)
        q = L.fft_create_plan(n, A.ptr(xin), A.ptr(y), A.LIQUID_FFT_FORWARD, 0); L.fft_execute(q)
        g["fft%d_in" % n] = xin; g["fft%d_out" % n] = y
    # --- end-to-end C1-shaped chain (2.4 MS/s, M = 4, block 40000 would be large: use 2 blocks of 8000 samples, Fc = 600k)
    fs, M, block, center = 2400000, 4, 8000, 100000000
    kinds = ["NBFM", "AM", "USB", "LSB"]; bws = [12500, 6000, 5400, 5400]
    fr = demod_frequencies(center, fs, len(kinds))
    xin = synth_iq(3 * block, fs, center, list(zip(kinds, fr)), seed=424242)
    rp = RefSDRPost("ref", fs, M)
    rds = [RefDemod("ref", k, bw, f, rp.chan_bw) for k, bw, f in zip(kinds, bws, fr)]
    audio = [[] for _ in kinds]; cnt_iq = [[] for _ in kinds]; cnt_au = [[] for _ in kinds]; lev = [[] for _ in kinds]
    for b in range(3):
        rp.run_block(xin[b * block:(b + 1) * block], center)
        cache = {}
        for i, rd in enumerate(rds):
            ch = rp.channel_at(rd.frequency)
            if ch not in cache:
                cache[ch] = rp.channel_data(ch)
            riq = rd.pre(*cache[ch]); o = rd.demodulate(riq)
            audio[i].append(o["audio"]); cnt_iq[i].append(riq.size); cnt_au[i].append(o["audio"].size); lev[i].append(o["level_accum"])
    g["chain_in"] = xin; g["chain_freqs"] = np.array(fr, np.int64)
    for i, k in enumerate(kinds):
        g["chain_%s_audio" % k] = np.concatenate(audio[i]); g["chain_%s_n_iq" % k] = np.array(cnt_iq[i], np.int64)
        g["chain_%s_n_audio" % k] = np.array(cnt_au[i], np.int64); g["chain_%s_level" % k] = np.array(lev[i], np.float64)
    sp = RefSpectrum("ref", 512)
    pts = []; ce = []; fl = []
    for b in range(3):
        p, c, f = sp.process_frame(xin[b * block:b * block + 1024]); pts.append(p); ce.append(c); fl.append(f)
    g["spec512_points"] = np.array(pts); g["spec512_ceil"] = np.array(ce); g["spec512_floor"] = np.array(fl)
    out = os.path.join(ROOT, "tests", "golden", "liquid_1_5_0.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes,", len(g), "arrays")


def main_b():
    """second fixture file (later additions keep their own seed so that the first file never changes):
    tests/golden/liquid_1_5_0_b.npz"""
    L = A.load("ref")
    rng = np.random.default_rng(20260923)
    g = {}
    # --- 2x oversampled channelizer (SDRPostThread.cpp:463, :505-507): M outputs per M/2 inputs
    for M in (4, 6, 20, 122):
        q = L.firpfbch2_crcf_create_kaiser(A.LIQUID_ANALYZER, M, 4, 60.0)
        nc = 40
        xin = rnd(rng, (M // 2) * nc); y = np.zeros(M * nc, np.complex64)
        L.oracle_firpfbch2_block(C.c_void_p(q), M, A.ptr(xin), nc, A.ptr(y))
        g["firpfbch2_M%d_in" % M] = xin; g["firpfbch2_M%d_out" % M] = y
    # --- ampmodem DSB, suppressed carrier (ModemDSB.cpp:6,49-51): Costas loop pulling in a carrier 0.002 cycles/sample off
    n = 3000
    t = np.arange(n)
    xin = ((np.sin(2 * np.pi * t * 0.01) * np.exp(1j * (0.3 + 2 * np.pi * 0.002 * t))) * 0.4).astype(np.complex64) + rnd(rng, n) * np.float32(0.05)
    q = L.ampmodem_create(0.5, 0, 1); y = np.zeros(n, np.float32)
    L.oracle_dsb_block(C.c_void_p(q), A.ptr(xin), n, A.ptr(y))
    g["ampmodem_dsb_in"] = xin; g["ampmodem_dsb_out"] = y
    # --- msresamp_cccf interpolation (ModemCW.cpp:124,163): 500 S/s -> 48 kS/s, six blocks of ten samples
    xin = rnd(rng, 60)
    q = L.msresamp_cccf_create(float(np.float32(48000 / 500)), 60.0)
    outs, cnts = [], []
    for b in range(6):
        xb = np.ascontiguousarray(xin[b * 10:(b + 1) * 10]); y = np.zeros(10 * 96 + 600, np.complex64); ny = C.c_uint()
        L.msresamp_cccf_execute(q, A.ptr(xb), 10, A.ptr(y), C.byref(ny))
        outs.append(y[:ny.value].copy()); cnts.append(ny.value)
    g["msresamp_cccf_in"] = xin; g["msresamp_cccf_out"] = np.concatenate(outs); g["msresamp_cccf_counts"] = np.array(cnts, np.int64)
    # --- the CW chain on those samples: beep oscillator + c2r Hilbert (ModemCW.cpp:171-178)
    lo = L.nco_crcf_create(A.LIQUID_NCO); hb = L.firhilbf_create(5, 60.0)
    L.nco_crcf_set_frequency(lo, float(np.float32(np.float32(2.0) * np.float32(np.pi) * np.float32(650.0) / np.float32(48000))))
    cw_in = np.ascontiguousarray(g["msresamp_cccf_out"]); cw = np.zeros(cw_in.size, np.float32)
    L.oracle_cw_block(C.c_void_p(lo), C.c_void_p(hb), A.ptr(cw_in), cw_in.size, A.ptr(cw))
    g["cw_chain_out"] = cw
    out = os.path.join(ROOT, "tests", "golden", "liquid_1_5_0_b.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes,", len(g), "arrays")


if __name__ == "__main__":
    if "--b-only" not in sys.argv:
        main()
    main_b()
