"""CPU tests that PIN the oracle: the plain-C restatement (oracle/liquid_port.c) and the CubicSDR control-flow
restatement (oracle/cubicsdr_chain.py) against tests/golden/liquid_1_5_0.npz, a fixture generated from the reference's
own liquid-dsp 1.5.0 binary by tests/golden/gen_golden.py.  Integer items must be exact; float samples within 5e-6 of
the peak (the restatement's rounding differs from the -ffast-math SSE build of the reference at the 1e-6 level).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle.liquid_api as A
from oracle.cubicsdr_chain import RefDemod, RefSDRPost, RefSpectrum
from tests.util import rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 5e-6


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(ROOT, "tests", "golden", "liquid_1_5_0.npz"))


@pytest.fixture(scope="module")
def P():
    if not A.available("port"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/liboracle_port.so"], check=True)
    return A.load("port")


def backends():
    out = ["port"]
    if A.available("ref"):
        out.append("ref")
    return out


def test_design_helpers(G, P):
    for name, n, fc, As in [("kaiser33", 33, 0.1, 60.0), ("kaiser161", 161, 0.025, 60.0)]:
        h = np.zeros(n, np.float32)
        P.liquid_firdes_kaiser(n, fc, As, 0.0, A.ptr(h))
        assert rel_err(h, G["firdes_" + name]) < TOL
    h = np.zeros(51, np.float32)
    P.liquid_firdes_notch(25, 0.0, 30.0, A.ptr(h))
    assert rel_err(h, G["firdes_notch_25_30"]) < TOL
    got = [P.estimate_req_filter_len(df, As) for df, As in [(0.1, 60.0), (0.05, 65.0), (0.35, 65.0), (0.2, 65.0)]]
    assert got == list(G["estimate_req_filter_len"])


def test_halfband_tables_are_the_reference_designs(G, P):
    for m in (3, 5, 10):
        q = P.resamp2_crcf_create(m, 0.0, 65.0)
        yy = np.zeros(8 * m + 4, np.complex64)
        for k in range(4 * m + 2):
            P.resamp2_crcf_interp_execute(q, A.cpx(1.0 if k == 0 else 0.0), A.ptr(yy[2 * k:2 * k + 2]))
        assert np.array_equal(yy.real[1::2][:2 * m], G["halfband_h1_m%d" % m])      # exact: the taps are tabulated


def test_nco_phase_words_bit_exact(G, P):
    x = G["nco_in"]
    for i, f in enumerate(G["nco_freqs"]):
        q = P.nco_crcf_create(A.LIQUID_VCO)
        P.nco_crcf_set_frequency(q, float(f))
        y = np.zeros_like(x)
        P.nco_crcf_mix_block_down(q, A.ptr(x), A.ptr(y), x.size)
        th, dth = C.c_uint32(), C.c_uint32()
        P.port_nco_get_state(C.c_void_p(q), C.byref(th), C.byref(dth))
        assert (th.value, dth.value) == tuple(int(v) for v in G["nco_words_after_600"][i])
        assert rel_err(y, G["nco_mix_down"][i]) < TOL


@pytest.mark.parametrize("tag,bs", [("r0025", 2500), ("r0119", 2111), ("r0108", 2500), ("r04", 777), ("r06", 500)])
def test_msresamp_crcf(G, P, tag, bs):
    q = P.msresamp_crcf_create(float(G["msresamp_crcf_%s_rate" % tag]), 60.0)
    xin = G["msresamp_crcf_%s_in" % tag]
    outs, cnts = [], []
    for b in range(3):
        xb = np.ascontiguousarray(xin[b * bs:(b + 1) * bs])
        y = np.zeros(bs + 600, np.complex64); ny = C.c_uint()
        P.msresamp_crcf_execute(q, A.ptr(xb), bs, A.ptr(y), C.byref(ny))
        outs.append(y[:ny.value].copy()); cnts.append(ny.value)
    assert cnts == list(G["msresamp_crcf_%s_counts" % tag])                       # bit-exact decimation indices
    assert rel_err(np.concatenate(outs), G["msresamp_crcf_%s_out" % tag]) < TOL


@pytest.mark.parametrize("tag,bs", [("r384", 120), ("r8", 60), ("r889", 55)])
def test_msresamp_rrrf(G, P, tag, bs):
    r = float(G["msresamp_rrrf_%s_rate" % tag])
    q = P.msresamp_rrrf_create(r, 60.0)
    xin = G["msresamp_rrrf_%s_in" % tag]
    outs, cnts = [], []
    for b in range(3):
        xb = np.ascontiguousarray(xin[b * bs:(b + 1) * bs])
        y = np.zeros(int(bs * r) + 600, np.float32); ny = C.c_uint()
        P.msresamp_rrrf_execute(q, A.ptr(xb), bs, A.ptr(y), C.byref(ny))
        outs.append(y[:ny.value].copy()); cnts.append(ny.value)
    assert cnts == list(G["msresamp_rrrf_%s_counts" % tag])
    assert rel_err(np.concatenate(outs), G["msresamp_rrrf_%s_out" % tag]) < TOL


@pytest.mark.parametrize("M", [4, 20, 122])
def test_firpfbch(G, P, M):
    q = P.firpfbch_crcf_create_kaiser(A.LIQUID_ANALYZER, M, 4, 60.0)
    xin = np.ascontiguousarray(G["firpfbch_M%d_in" % M]); y = np.zeros_like(xin)
    P.oracle_firpfbch_analyzer_block(C.c_void_p(q), M, A.ptr(xin), xin.size // M, A.ptr(y))
    assert rel_err(y, G["firpfbch_M%d_out" % M]) < TOL


@pytest.mark.parametrize("M", [4, 6, 20, 122])
def test_firpfbch2(P, M):
    """2x oversampled analyzer (runPFBCH2): the restatement against the reference binary's outputs"""
    G2 = np.load(os.path.join(ROOT, "tests", "golden", "liquid_1_5_0_b.npz"))
    q = P.firpfbch2_crcf_create_kaiser(A.LIQUID_ANALYZER, M, 4, 60.0)
    xin = np.ascontiguousarray(G2["firpfbch2_M%d_in" % M]); y = np.zeros(2 * xin.size, np.complex64)
    P.oracle_firpfbch2_block(C.c_void_p(q), M, A.ptr(xin), xin.size // (M // 2), A.ptr(y))
    assert rel_err(y, G2["firpfbch2_M%d_out" % M]) < TOL


def test_dsb_costas_loop_and_cw_chain(P):
    """ampmodem DSB (suppressed carrier) -- whole loop trajectory -- and the CW chain (msresamp_cccf interpolation, beep
    oscillator, c2r Hilbert): the restatements against the reference binary's outputs"""
    G2 = np.load(os.path.join(ROOT, "tests", "golden", "liquid_1_5_0_b.npz"))
    xin = np.ascontiguousarray(G2["ampmodem_dsb_in"]); y = np.zeros(xin.size, np.float32)
    q = P.ampmodem_create(0.5, 0, 1)
    P.oracle_dsb_block(C.c_void_p(q), A.ptr(xin), xin.size, A.ptr(y))
    assert rel_err(y, G2["ampmodem_dsb_out"]) < TOL
    xin = np.ascontiguousarray(G2["msresamp_cccf_in"])
    q = P.msresamp_cccf_create(float(np.float32(48000 / 500)), 60.0)
    outs, cnts = [], []
    for b in range(6):
        xb = np.ascontiguousarray(xin[b * 10:(b + 1) * 10]); yb = np.zeros(10 * 96 + 600, np.complex64); ny = C.c_uint()
        P.msresamp_cccf_execute(q, A.ptr(xb), 10, A.ptr(yb), C.byref(ny))
        outs.append(yb[:ny.value].copy()); cnts.append(ny.value)
    assert cnts == list(G2["msresamp_cccf_counts"])
    up = np.concatenate(outs)
    assert rel_err(up, G2["msresamp_cccf_out"]) < TOL
    lo = P.nco_crcf_create(A.LIQUID_NCO); hb = P.firhilbf_create(5, 60.0)
    P.nco_crcf_set_frequency(lo, float(np.float32(np.float32(2.0) * np.float32(np.pi) * np.float32(650.0) / np.float32(48000))))
    cw_in = np.ascontiguousarray(G2["msresamp_cccf_out"]); cw = np.zeros(cw_in.size, np.float32)
    P.oracle_cw_block(C.c_void_p(lo), C.c_void_p(hb), A.ptr(cw_in), cw_in.size, A.ptr(cw))
    assert rel_err(cw, G2["cw_chain_out"]) < TOL


def test_filters_and_modems(G, P):
    x = np.ascontiguousarray(G["dcblock_in"]); y = np.zeros_like(x)
    P.iirfilt_crcf_execute_block(P.iirfilt_crcf_create_dc_blocker(0.0005), A.ptr(x), x.size, A.ptr(y))
    assert rel_err(y, G["dcblock_out"]) < TOL
    x = np.ascontiguousarray(G["butter6_in"]); y = np.zeros_like(x)
    P.iirfilt_crcf_execute_block(P.iirfilt_crcf_create_lowpass(6, 0.25), A.ptr(x), x.size, A.ptr(y))
    assert rel_err(y, G["butter6_out"]) < TOL
    x = np.ascontiguousarray(G["freqdem_in"]); yf = np.zeros(x.size, np.float32)
    P.freqdem_demodulate_block(P.freqdem_create(0.5), A.ptr(x), x.size, A.ptr(yf))
    assert rel_err(yf, G["freqdem_out"]) < TOL
    x = np.ascontiguousarray(G["am_in"]); yf = np.zeros(x.size, np.float32)
    P.oracle_am_block(C.c_void_p(P.firfilt_rrrf_create_dc_blocker(25, 30.0)), A.ptr(x), x.size, A.ptr(yf))
    assert rel_err(yf, G["am_out"]) < TOL
    for usb in (1, 0):
        x = np.ascontiguousarray(G["ssb_usb%d_in" % usb]); yf = np.zeros(x.size, np.float32)
        nco = P.nco_crcf_create(A.LIQUID_NCO); P.nco_crcf_set_frequency(nco, float(np.float32(2 * np.pi * 0.25)))
        P.oracle_ssb_block(C.c_void_p(nco), C.c_void_p(P.iirfilt_crcf_create_lowpass(6, 0.25)), C.c_void_p(P.firhilbf_create(5, 90.0)),
                           usb, A.ptr(x), x.size, A.ptr(yf))
        assert rel_err(yf, G["ssb_usb%d_out" % usb]) < TOL


@pytest.mark.parametrize("n", [8, 256, 4096])
def test_fft(G, P, n):
    x = np.ascontiguousarray(G["fft%d_in" % n]); y = np.zeros_like(x)
    P.fft_execute(P.fft_create_plan(n, A.ptr(x), A.ptr(y), A.LIQUID_FFT_FORWARD, 0))
    assert rel_err(y, G["fft%d_out" % n]) < TOL
    assert rel_err(np.fft.fft(x.astype(np.complex128)), G["fft%d_out" % n]) < TOL   # unnormalised forward DFT


@pytest.mark.parametrize("backend", backends())
def test_cubicsdr_chain_end_to_end(G, backend):
    """SDRPostThread -> DemodulatorPreThread -> Modem -> audio, 3 blocks, 4 modem kinds; plus the spectrum processor."""
    fs, M, block, center = 2400000, 4, 8000, 100000000
    kinds = ["NBFM", "AM", "USB", "LSB"]; bws = [12500, 6000, 5400, 5400]
    fr = [int(v) for v in G["chain_freqs"]]
    xin = G["chain_in"]
    rp = RefSDRPost(backend, fs, M)
    rds = [RefDemod(backend, k, bw, f, rp.chan_bw) for k, bw, f in zip(kinds, bws, fr)]
    audio = [[] for _ in kinds]; n_iq = [[] for _ in kinds]; lev = [[] for _ in kinds]
    for b in range(3):
        rp.run_block(xin[b * block:(b + 1) * block], center)
        cache = {}
        for i, rd in enumerate(rds):
            ch = rp.channel_at(rd.frequency)
            if ch not in cache:
                cache[ch] = rp.channel_data(ch)
            riq = rd.pre(*cache[ch]); o = rd.demodulate(riq)
            audio[i].append(o["audio"]); n_iq[i].append(riq.size); lev[i].append(o["level_accum"])
    for i, k in enumerate(kinds):
        assert n_iq[i] == list(G["chain_%s_n_iq" % k])
        assert [a.size for a in audio[i]] == list(G["chain_%s_n_audio" % k])
        assert rel_err(np.concatenate(audio[i]), G["chain_%s_audio" % k]) < 2 * TOL, k
        assert np.allclose(lev[i], G["chain_%s_level" % k], rtol=1e-5)
    sp = RefSpectrum(backend, 512)
    for b in range(3):
        p, c, f = sp.process_frame(xin[b * block:b * block + 1024])
        assert rel_err(p, G["spec512_points"][b]) < 2 * TOL
        assert abs(c - G["spec512_ceil"][b]) <= 1e-6 * abs(c)


@pytest.mark.skipif(not A.available("ref"), reason="reference DLL not staged (oracle/_ref/libliquid.dll)")
def test_port_tracks_reference_on_fresh_inputs(P):
    """beyond the fixture: random ratios / block sizes, restatement vs the reference binary run here"""
    R = A.load("ref")
    rng = np.random.default_rng(99)
    for trial in range(6):
        r = float(rng.uniform(0.004, 0.95)); bs = int(rng.integers(300, 3000))
        qa, qb = R.msresamp_crcf_create(r, 60.0), P.msresamp_crcf_create(r, 60.0)
        for blk in range(3):
            x = ((rng.standard_normal(bs) + 1j * rng.standard_normal(bs)) * 0.3).astype(np.complex64)
            ya = np.zeros(bs + 600, np.complex64); yb = np.zeros(bs + 600, np.complex64); na, nb = C.c_uint(), C.c_uint()
            R.msresamp_crcf_execute(qa, A.ptr(x), bs, A.ptr(ya), C.byref(na)); P.msresamp_crcf_execute(qb, A.ptr(x), bs, A.ptr(yb), C.byref(nb))
            assert na.value == nb.value, (r, bs, blk)
            assert rel_err(yb[:nb.value], ya[:na.value]) < TOL if na.value else True
        S, bi, ph, st = C.c_uint(), C.c_uint(), C.c_uint32(), C.c_uint32()
        P.port_msresamp_get_state(C.c_void_p(qb), C.byref(S), C.byref(bi), C.byref(ph), C.byref(st))
        assert ph.value < (1 << 25)


@pytest.mark.skipif(not A.available("ref"), reason="reference DLL not staged (oracle/_ref/libliquid.dll)")
def test_reference_state_words_are_where_the_accessors_read_them(P):
    """oracle/liquid_api.py nco_state / msresamp_state read the reference binary's phase words at fixed offsets inside ITS objects (it has no
    accessor for them): on identical inputs they must be the words the restatement's hooks return -- oscillator phase and frequency words after
    mix_block_up / _down, half-band stage count, buffer_index, step and phase of msresamp_crcf / _rrrf after ragged blocks (decimating and
    interpolating ratios)."""
    R = A.load("ref")
    rng = np.random.default_rng(2024)
    for f in (0.0, 0.7123, 3.0001, 6.2):
        qa, qb = R.nco_crcf_create(A.LIQUID_VCO), P.nco_crcf_create(A.LIQUID_VCO)
        R.nco_crcf_set_frequency(qa, f); P.nco_crcf_set_frequency(qb, f)
        for n, up in ((1000, False), (777, True), (8394, False)):
            x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.3).astype(np.complex64)
            ya, yb = np.empty_like(x), np.empty_like(x)
            (R.nco_crcf_mix_block_up if up else R.nco_crcf_mix_block_down)(qa, A.ptr(x), A.ptr(ya), n)
            (P.nco_crcf_mix_block_up if up else P.nco_crcf_mix_block_down)(qb, A.ptr(x), A.ptr(yb), n)
            assert A.nco_state("ref", qa) == A.nco_state("port", qb), (f, n)
    for r in (0.0248, 0.3, 0.0107, 0.62, 1.33, 2.5):
        qa, qb = R.msresamp_crcf_create(r, 60.0), P.msresamp_crcf_create(r, 60.0)
        for bs in (8394, 8394, 777, 1):
            x = ((rng.standard_normal(bs) + 1j * rng.standard_normal(bs)) * 0.3).astype(np.complex64)
            cap = int(bs * max(r, 1.0)) + 600
            ya, yb = np.zeros(cap, np.complex64), np.zeros(cap, np.complex64)
            na, nb = C.c_uint(), C.c_uint()
            R.msresamp_crcf_execute(qa, A.ptr(x), bs, A.ptr(ya), C.byref(na)); P.msresamp_crcf_execute(qb, A.ptr(x), bs, A.ptr(yb), C.byref(nb))
            assert na.value == nb.value
            assert A.msresamp_state("ref", qa) == A.msresamp_state("port", qb), (r, bs)
    for r in (48000 / 12500.0, 48000 / 200000.0):
        qa, qb = R.msresamp_rrrf_create(r, 60.0), P.msresamp_rrrf_create(r, 60.0)
        for bs in (208, 209, 3333):
            x = (rng.standard_normal(bs) * 0.3).astype(np.float32)
            cap = int(bs * max(r, 1.0)) + 600
            ya, yb = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
            na, nb = C.c_uint(), C.c_uint()
            R.msresamp_rrrf_execute(qa, A.ptr(x), bs, A.ptr(ya), C.byref(na)); P.msresamp_rrrf_execute(qb, A.ptr(x), bs, A.ptr(yb), C.byref(nb))
            assert na.value == nb.value
            assert A.msresamp_state("ref", qa) == A.msresamp_state("port", qb), (r, bs)


_MODEM_CASES = [("NBFM", 12500, 48000, {}), ("FM", 200000, 48000, {}), ("AM", 6000, 48000, {}), ("USB", 5400, 48000, {}), ("LSB", 5401, 44100, {}),
                ("DSB", 5400, 48000, {}), ("CW", 500, 48000, {}), ("I/Q", 12345, 48000, {}), ("FMS", 200000, 48000, {}),
                ("FMS", 250000, 44100, {"demph": 50}), ("FMS", 50000, 48000, {"demph": 0}), ("FM", 400000, 48000, {}), ("AM", 300, 48000, {})]


@pytest.mark.parametrize("name,bw,audio_rate,settings", _MODEM_CASES)
def test_python_modem_glue_equals_the_reference_modem_classes(name, bw, audio_rate, settings):
    """oracle/cubicsdr_chain.py RefDemod.demodulate -- the checker of the GPU parity tests -- against the reference's OWN modem
    sources (src/modules/modem/analog/Modem*.cpp compiled unmodified into oracle/_ref/libref_modems.so) on the reference's own liquid
    binary: six consecutive blocks (ragged sizes, one empty), audio and channel count bit for bit, rate rules and useSignalOutput too."""
    from oracle import ref_modems as RM
    if not RM.available():
        pytest.skip("oracle/_ref/libref_modems.so is built only where /root/reference is")
    cpp = RM.RefModem(name, bw, audio_rate, settings)
    py = RefDemod("ref", name, bw, 100000000, 600000, audio_rate=audio_rate, demph=int(settings.get("demph", 75)))
    assert py.bandwidth == cpp.rate                                    # checkSampleRate
    assert py.use_signal_output == cpp.use_signal_output
    rng = np.random.default_rng(len(name) * 1000 + bw)
    total = 0
    for b in range(6):
        n = 0 if b == 3 else int(round(cpp.rate / 60.0)) + (b % 3) - 1
        t = (np.arange(n) + 1000 * b) / cpp.rate
        iq = (0.3 * np.exp(2j * np.pi * 700.0 * t * (1 + 0.2 * np.sin(2 * np.pi * 300 * t))) * (1 + 0.5 * np.sin(2 * np.pi * 440 * t))
              + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
        w = py.demodulate(iq)
        if n == 0:
            assert w is None                                           # the reference's modems return at once on an empty block
            continue
        a, ch = cpp.demodulate(iq)
        assert ch == w.get("channels", 1), (b, ch)
        assert a.size == w["audio"].size and np.array_equal(a, w["audio"]), (name, b, a.size, w["audio"].size)
        d = cpp.demod_output()
        if d is not None and name not in ("CW",):                      # ModemAnalog::getDemodOutputData holds the block's (scaled) demodulator output
            assert d.size == iq.size
        total += a.size
    assert total > 0
    cpp.close()


def _spectrum_pair(F, rate=2400000, freq=100000000):
    from oracle import ref_modems as RM
    cpp = RM.RefSpectrumCpp(F, rate)
    cpp.set_center(freq); cpp.set_bandwidth(rate)
    return cpp, RefSpectrum("ref", F)


@pytest.mark.parametrize("case", ["full", "short_overlap", "scale_rate", "peak_hold", "hide_dc", "view"])
def test_python_spectrum_glue_equals_the_reference_spectrum_processor(case):
    """oracle/cubicsdr_chain.py RefSpectrum.process_input -- the checker of the GPU spectrum tests -- against the reference's OWN
    src/process/SpectrumVisualProcessor.cpp (compiled unmodified into oracle/_ref/libref_spectrum.so) on its own liquid binary, one
    process() per block: which calls produce output, the points, the held points, fft_ceiling / fft_floor -- bit for bit."""
    from oracle import ref_modems as RM
    if not RM.spectrum_available():
        pytest.skip("oracle/_ref/libref_spectrum.so is built only where /root/reference is")
    rng = np.random.default_rng(3)
    freq, rate = 100000000, 2400000

    def sig(n, t0):
        t = np.arange(n) + t0
        return (0.3 * np.exp(2j * np.pi * 0.07 * t) + 0.1 * np.exp(-2j * np.pi * 0.21 * t) * (1 + 0.5 * np.sin(2 * np.pi * t / 5000))
                + 0.05 * (rng.standard_normal(n) + 1j * rng.standard_normal(n)) + 0.02).astype(np.complex64)

    F, n, nblk, events = 1024, 40000, 6, {}
    if case == "short_overlap":
        F, n, nblk = 2048, 1500, 12
    elif case == "scale_rate":
        F, n = 512, 5000
    elif case == "peak_hold":
        nblk = 40
    cpp, py = _spectrum_pair(F, rate, freq)
    if case == "scale_rate":
        cpp.set_scale(2.0); cpp.set_average_rate(0.3)
        py.sf = float(np.float32(2.0)); py.rate = float(np.float32(0.3))
    elif case == "peak_hold":
        cpp.set_peak_hold(True); py.set_peak_hold(True)
        events = {20: True, 33: False}                                   # a second "on" restarts the hold (PEAK_RESET_COUNT), then off
    elif case == "hide_dc":
        cpp.set_hide_dc(True); py.set_hide_dc(True, freq, rate, freq)
    elif case == "view":
        cpp.set_view(True, freq + 250000, 300000); py.set_view(True, freq + 250000, 300000)
    nout = 0
    for b in range(nblk):
        if b in events:
            cpp.set_peak_hold(events[b]); py.set_peak_hold(events[b])
        x = sig(n, b * n)
        a, w = cpp.process(x, freq, rate), py.process_input(x, freq, rate)
        assert (a is None) == (w is None), (case, b)
        if a is None:
            continue
        nout += 1
        assert np.array_equal(a[0], w[0]), (case, b, float(np.abs(a[0] - w[0]).max()))
        assert a[1] == w[1] and a[2] == w[2], (case, b)
        assert (a[3] is None) == (w[3] is None), (case, b)
        if a[3] is not None:
            assert np.array_equal(a[3], w[3]), (case, b)
    assert nout >= nblk - 2
    cpp.close()


@pytest.mark.parametrize("fft,lps,rate,blk", [(2048, 30, 2400000, 40000), (4096, 60, 10000000, 166680), (1024, 10, 250000, 4167), (2048, 120, 2400000, 40000)])
def test_python_line_pacing_equals_the_reference_distributor(fft, lps, rate, blk):
    """oracle/fft_distributor.py (the checker of the waterfall line pacing, host mirror and GPU tests) against the reference's OWN
    src/process/FFTDataDistributor.cpp (oracle/_ref/libref_distributor.so): 40 blocks with a retune in the middle -- which input
    produces which lines, each line's first sample and length, identical."""
    from oracle.fft_distributor import FFTDataDistributorRef
    path = os.path.join(os.path.dirname(A.lib_path("ref")), "libref_distributor.so")
    if not (A.available("ref") and os.path.exists(path)):
        pytest.skip("oracle/_ref/libref_distributor.so is built only where /root/reference is")
    A.load("ref")
    L = C.CDLL(path)
    L.refdist_create.restype = C.c_void_p; L.refdist_create.argtypes = [C.c_uint, C.c_uint]
    L.refdist_push.restype = C.c_int; L.refdist_push.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int]
    L.refdist_destroy.argtypes = [C.c_void_p]
    h = L.refdist_create(fft, lps); py = FFTDataDistributorRef(fft, lps)
    nid = lines = 0
    for b in range(40):
        f = 100000000 if b < 25 else 101000000
        ids = np.zeros(10000, np.int64); ln = np.zeros(10000, np.int32)
        m = L.refdist_push(h, blk, f, rate, A.ptr(ids), A.ptr(ln), 10000)
        want = py.push(list(range(nid, nid + blk)), f, rate); nid += blk
        assert [(int(ids[i]), int(ln[i])) for i in range(m)] == [(w[0], w[1]) for w in want], b
        lines += m
    assert lines > 3
    L.refdist_destroy(h)


@pytest.mark.parametrize("use_signal_output", [False, True])
def test_level_squelch_restatement_equals_the_reference_demodulator_thread(use_signal_output):
    """oracle/cubicsdr_chain.py RefLevelSquelch -- the checker cubicsdr_amd/host/DemodLevel.h is held to -- against the reference's OWN
    src/demod/DemodulatorThread.cpp running on its thread (oracle/_ref/libref_demodthread.so), fed 300 blocks through a feed modem: quiet and
    loud stretches, empty audio, the squelch switched on with a moving threshold, muting.  Signal level / floor / ceiling bit for bit
    (float32), the squelch-break flag, whether the audio was pushed on, the audio peak, and the scope tap rule (audio when it outnumbers the
    block's IQ samples, else the modem's demodulator output; at most DEMOD_VIS_SIZE samples; the rates it is labelled with)."""
    from oracle import ref_modems as RM
    from oracle.cubicsdr_chain import RefLevelSquelch
    if not RM.demodthread_available():
        pytest.skip("oracle/_ref/libref_demodthread.so is built only where /root/reference is")
    rng = np.random.default_rng(17)
    ref = RM.RefDemodThreadCpp(use_signal_output, 12500, 48000)
    py = RefLevelSquelch()
    flips = pushed = taps_audio = taps_demod = 0
    last = None
    for b in range(300):
        n_iq = 208 + b % 3
        loud = (b // 40) % 2 == 1
        amp = 0.25 if loud else 0.003
        iq = (amp * (rng.standard_normal(n_iq) + 1j * rng.standard_normal(n_iq))).astype(np.complex64)
        n_audio = 0 if b % 23 == 7 else (150 if b % 5 == 4 else 800 + b % 2)          # empty audio; fewer audio samples than IQ samples; the usual 800
        audio = (amp * 2.0 * rng.standard_normal(n_audio)).astype(np.float32)
        sq = 100 <= b < 260
        sl = -30.0 if b < 180 else -8.0
        muted = 270 <= b < 280
        ref.set(sq, sl, muted)
        got = ref.block(iq, audio)
        have = n_audio > 0
        if use_signal_output:
            accum, count = float(np.sum(np.abs(audio.astype(np.float64)))), n_audio
        else:
            accum, count = float(np.sum(np.sqrt(iq.real.astype(np.float64) ** 2 + iq.imag.astype(np.float64) ** 2))), n_iq
        squelched = py.step(have, accum, count, float(n_iq) / 12500.0, sq, sl)
        assert got["level"] == py.level and got["floor"] == py.floor and got["ceil"] == py.ceil, (b, got, py.level, py.floor, py.ceil)
        assert got["squelch_break"] == py.squelch_break, b
        assert got["pushed"] == (not squelched and not muted), (b, got["pushed"], squelched, muted)
        assert got["peak"] == (np.float32(np.max(np.abs(audio))) if n_audio else np.float32(0)), b
        # scope tap (:240-316): only un-squelched blocks; audio when numAudioWritten > bufSize, else getDemodOutputData (here: Re of the IQ)
        if squelched:
            assert got["tap"] is None, b
        else:
            assert got["tap"] is not None, b
            if n_audio > n_iq:
                assert np.array_equal(got["tap"], audio[:2048]) and got["tap_input_rate"] == 48000 and got["tap_sample_rate"] == 12500, b
                taps_audio += 1
            else:
                assert np.array_equal(got["tap"], iq.real[:2048]) and got["tap_input_rate"] == 12500 and got["tap_sample_rate"] == 12500, b
                taps_demod += 1
            assert got["tap_type"] == 0
        flips += last is not None and squelched != last
        last = squelched
        pushed += got["pushed"]
    ref.close()
    assert flips >= 2 and pushed > 50 and taps_audio > 20 and taps_demod > 5, (flips, pushed, taps_audio, taps_demod)


@pytest.mark.parametrize("fs,M,oversampled", [(2400000, 4, False), (6100000, 122, False), (2400000, 4, True), (480000, 1, False)])
def test_post_restatement_equals_the_reference_post_thread(fs, M, oversampled):
    """oracle/cubicsdr_chain.py RefSDRPost -- the checker of every GPU channelizer / routing test -- against the reference's OWN
    src/sdr/SDRPostThread.cpp running on its thread on the reference's liquid binary (oracle/_ref/libref_post.so): what every demodulator
    finds in its input pipe after each block -- the channel it was routed to (its centre frequency stamp), the channel rate, the samples
    (channel 0 behind the DC blocker, the wrap channel) -- bit for bit, for M = 4, M = 122 (the headline channel count), the oversampled
    analyzer and single-channel mode; the active list after a retune that moves demodulators out of range; the visual queues."""
    from oracle import ref_modems as RM
    if not RM.post_available():
        pytest.skip("oracle/_ref/libref_post.so is built only where /root/reference is")
    center = 100000000
    block = (fs // 60 // M) * M if M > 1 else 8000
    ref = RM.RefPostThreadCpp(center, fs, oversampled)
    py = RefSDRPost("ref", fs, M, oversampled=oversampled)
    chan_bw = fs // M
    # demodulators: off-centre ones, one exactly on a channel edge (tie -> the lower index wins), one at the wrap channel, one on channel 0
    freqs = [center + 3700, center + chan_bw // 2, center + fs // 2 - 1000, center - fs // 2 + chan_bw + 900, center - chan_bw - 250] if M > 1 else [center + 10000, center - 50000]
    for i, f in enumerate(freqs):
        ref.add_demod(f, current=(i == 0))
    ref.notify()
    rng = np.random.default_rng(9)
    t = np.arange(block)
    checked = 0
    for b in range(6):
        cf = center if b < 4 else center + fs // 2 + chan_bw            # the last blocks: a retune that leaves some demodulators out of range
        x = (0.1 * (rng.standard_normal(block) + 1j * rng.standard_normal(block)) + 0.3 * np.exp(2j * np.pi * 0.013 * (t + b * block)) + (0.01 + 0.01j)).astype(np.complex64)
        active = ref.block(x, cf, fs, M)                                # (the first block builds the channelizer AND the active list before it runs, :418-430)
        py.run_block(x, cf)
        cache = {}                                                       # one buffer per channel and block, shared by its demodulators (:341-396)
        for i, f in enumerate(freqs):
            got = ref.fetch(i)
            in_range = abs(cf - f) <= fs // 2
            if b >= 4 and not in_range:
                # the block that FOLLOWS the retune still runs the old list (:187-200): only from the next one on is the demodulator off
                if b == 5:
                    assert got is None and not active[i], (b, i)
                continue
            if got is None:
                continue
            ch = py.channel_at(f)
            if ch not in cache:
                cache[ch] = py.channel_data(ch)
            want, fc, rate = cache[ch]
            assert got[1] == fc and got[2] == rate, (b, i, got[1], fc, got[2], rate)
            assert got[0].size == want.size and np.array_equal(got[0].view(np.uint32), want.view(np.uint32)), (b, i)
            checked += 1
        vis = ref.fetch_visual(0)
        assert vis is not None and vis[1] == cf and vis[2] == fs
        if M == 1:
            assert np.array_equal(vis[0], py.data_out)                   # single-channel mode shows the DC-corrected block (:284-299)
        else:
            assert np.array_equal(vis[0], x)                             # the full-rate copy (:221-245)
    ref.close()
    assert checked >= (8 if M > 1 else 4), checked
