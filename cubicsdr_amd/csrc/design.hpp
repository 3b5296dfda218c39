// design.hpp -- host-side (cold path) filter design and integer bookkeeping for the HIP kernels.
//
// The reference builds its filters on a worker thread by calling liquid-dsp 1.5.0 create() functions
// (DemodulatorWorkerThread.cpp:63-101, ModemAnalog.cpp:21-33, SDRPostThread.cpp:29,406, ModemAM.cpp:9,
// ModemUSB.cpp:8-11).  The GPU kernels need the same coefficient sets as plain arrays, so this file
// computes them on the host with the arithmetic liquid 1.5.0 uses (float where the rounding decides an
// integer such as a filter length or a phase step) and uploads them once per
// (re)configuration.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <complex>
#include <vector>

namespace csdr {
namespace design {

constexpr float kPi = 3.14159265358979323846f;

// ---- Kaiser-window prototype design (liquid firdes.c / math.bessel.c / windows.c semantics) ----------------
inline float lngamma_f(float z) {
    // recursion below 10, Stirling-like closed form above (liquid math.gamma.c)
    float shift = 0.0f;
    while (z < 10.0f) { shift += std::log(z); z += 1.0f; }
    float g = 0.5f * (std::log(2.0f * kPi) - std::log(z));
    g += z * (std::log(z + (1.0f / (12.0f * z - 0.1f / z))) - 1.0f);
    return g - shift;
}

inline float bessel_i0_f(float z) {
    if (z == 0.0f) return 1.0f;
    float acc = 0.0f;
    const float lz = std::log(0.5f * z);
    for (unsigned k = 0; k < 32; ++k) acc += std::exp(2.0f * ((float)k * lz - lngamma_f((float)k + 1.0f)));
    return acc;
}

inline float sinc_f(float x) {
    if (std::fabs(x) < 0.01f) return std::cos(kPi * x / 2.0f) * std::cos(kPi * x / 4.0f) * std::cos(kPi * x / 8.0f);
    return std::sin(kPi * x) / (kPi * x);
}

inline float kaiser_beta(float as) {
    as = std::fabs(as);
    if (as > 50.0f) return 0.1102f * (as - 8.7f);
    if (as > 21.0f) return 0.5842f * std::pow(as - 21.0f, 0.4f) + 0.07886f * (as - 21.0f);
    return 0.0f;
}

inline float kaiser_window(unsigned i, unsigned n, float beta) {
    const float t = (float)i - (float)(n - 1) / 2;
    const float r = 2.0f * t / (float)(n - 1);
    return bessel_i0_f(beta * std::sqrt(1 - r * r)) / bessel_i0_f(beta);
}

inline unsigned required_filter_len(float df, float as) { return (unsigned)((as - 7.95f) / (14.26f * df)); }

inline std::vector<float> kaiser_lowpass(unsigned n, float fc, float as) {
    std::vector<float> h(n);
    const float beta = kaiser_beta(as);
    for (unsigned i = 0; i < n; ++i) {
        const float t = (float)i - (float)(n - 1) / 2;
        h[i] = sinc_f(2.0f * fc * t) * kaiser_window(i, n, beta);
    }
    return h;
}

// ---- half-band filters ---------------------------------------------------------------------------------------
// liquid 1.5.0 designs resamp2's prototype with an iterative equiripple optimiser whose result is not a closed
// form.  CubicSDR creates every msresamp with As = 60 dB (DemodulatorWorkerThread.cpp:100, ModemAnalog.cpp:25,
// SpectrumVisualProcessor.cpp:361), msresamp2 adds 5 dB, and the stage rule below only ever yields m = 10, 5, 3.
// Those three coefficient sets are design constants of the reference's DSP library, tabulated here (first half of
// the symmetric odd-tap branch; the centre tap acts as a pure delay).  Other (m, As) fall back to a Kaiser
// half-band and are reported as not parity-pinned.
inline bool halfband_branch(unsigned m, float as, std::vector<float> &h1, bool *pinned = nullptr) {
    static const float t3[3] = {0x1.31fb88p-6f, -0x1.d5fe0ep-4f, 0x1.31556cp-1f};
    static const float t5[5] = {0x1.4ae43ap-8f, -0x1.679e46p-6f, 0x1.084878p-4f, -0x1.57458ap-3f, 0x1.3da7d4p-1f};
    static const float t10[10] = {-0x1.7604f2p-10f, 0x1.caf71ap-9f, -0x1.e992e0p-8f, 0x1.cc3ab4p-7f, -0x1.8f23dcp-6f,
                                  0x1.495b22p-5f,   -0x1.0a5864p-4f, 0x1.b87790p-4f, -0x1.992bc8p-3f, 0x1.43c83ep-1f};
    const float *t = nullptr;
    if (std::fabs(as - 65.0f) < 1e-3f) t = (m == 3) ? t3 : (m == 5) ? t5 : (m == 10) ? t10 : nullptr;
    h1.assign(2 * m, 0.0f);
    if (pinned) *pinned = (t != nullptr);
    if (t) {
        for (unsigned i = 0; i < m; ++i) h1[i] = h1[2 * m - 1 - i] = t[i];
        return true;
    }
    const unsigned n = 4 * m + 1;
    const float beta = kaiser_beta(as);
    for (unsigned j = 0, i = 1; i < n; i += 2, ++j) {
        const float tt = (float)i - (float)(n - 1) / 2.0f;
        h1[j] = sinc_f(tt / 2.0f) * kaiser_window(i, n, beta);
    }
    return true;
}

// ---- multi-stage resampler plan (msresamp + msresamp2 + resamp of liquid 1.5.0) ----------------------------
struct MsresampPlan {
    bool interp = false;        // rate > 1
    unsigned S = 0;             // number of half-band stages
    float rate_arb = 1.0f;      // arbitrary stage rate in [0.5, 1) (decim) or (1, 2] (interp)
    uint32_t step = 0;          // round(2^24 / rate_arb)
    std::vector<unsigned> m;    // per design index (index 0 = lowest-rate stage)
    std::vector<std::vector<float>> h1;  // per design index: 2m symmetric taps of the filtered branch
    std::vector<float> arms;    // [256][14] arbitrary-resampler polyphase bank, oldest-sample-first
    bool pinned = true;         // false if a non-tabulated half-band was needed
    static constexpr unsigned kArms = 256, kArmTaps = 14;
};

inline MsresampPlan plan_msresamp(float rate, float as) {
    MsresampPlan p;
    p.interp = rate > 1.0f;
    p.rate_arb = rate;
    if (p.interp) while (p.rate_arb > 2.0f) { ++p.S; p.rate_arb *= 0.5f; }
    else          while (p.rate_arb < 0.5f) { ++p.S; p.rate_arb *= 2.0f; }
    // half-band stages: msresamp2_create(type, S, fc = 0.4, f0 = 0, As)
    float fc = 0.4f;
    const float as2 = as + 5.0f;
    for (unsigned i = 0; i < p.S; ++i) {
        fc = (i == 1) ? (0.5f - fc) * 0.5f : 0.5f * fc;
        const float ft = 2 * (0.25f - fc);
        const unsigned hl = required_filter_len(ft, as2);
        unsigned mm = (unsigned)std::ceil((float)(hl - 1) / 4.0f);
        if (mm < 3) mm = 3;
        std::vector<float> h;
        bool pin = true;
        halfband_branch(mm, as2, h, &pin);
        p.pinned = p.pinned && pin;
        p.m.push_back(mm);
        p.h1.push_back(h);
    }
    // arbitrary stage: resamp_create(rate_arb, m = 7, fc = min(0.515 rate_arb, 0.49), As, npfb = 256)
    float fca = 0.515f * p.rate_arb;
    if (fca > 0.49f) fca = 0.49f;
    p.step = (uint32_t)std::round((float)(1u << 24) / p.rate_arb);
    const unsigned npfb = MsresampPlan::kArms, sub = MsresampPlan::kArmTaps, n = sub * npfb + 1;
    std::vector<float> hf = kaiser_lowpass(n, fca / (float)npfb, as);
    float gain = 0.0f;
    for (float v : hf) gain += v;
    gain = (float)npfb / gain;
    p.arms.resize((size_t)npfb * sub);
    for (unsigned a = 0; a < npfb; ++a)
        for (unsigned k = 0; k < sub; ++k) p.arms[a * sub + (sub - 1 - k)] = hf[a + k * npfb] * gain;
    return p;
}

// number of arbitrary-stage outputs produced while consuming K inputs starting at 24-bit phase `phase`,
// and the phase afterwards (closed form of the while-loop in resamp_execute)
inline uint64_t resamp_count(uint64_t K, uint32_t phase, uint32_t step, uint32_t *phase_after) {
    const uint64_t lim = K << 24;
    uint64_t J = 0;
    if (lim > phase) J = (lim - phase + step - 1) / step;
    if (phase_after) *phase_after = (uint32_t)((uint64_t)phase + J * step - lim);
    return J;
}

// ---- NCO frequency word (nco_crcf_set_frequency of liquid 1.5.0: float -> uint32) ---------
inline uint32_t nco_phase_word(float theta) {
    const float p = (float)((double)theta * 0.159154943091895);
    float frac = p - (float)((long)p);
    if (frac < 0.0f) frac += 1.0f;
    return (uint32_t)(int64_t)(frac * 4294967296.0f);
}
inline std::vector<float> nco_sine_table() {
    std::vector<float> t(1024);
    // the argument is formed in float, its sine correctly rounded (the reference binary's table bit for bit; glibc's sinf differs in 15 entries)
    for (unsigned i = 0; i < 1024; ++i) t[i] = (float)std::sin((double)(2.0f * kPi * (float)i / 1024.0f));
    return t;
}

// ---- polyphase analysis channelizer prototype (firpfbch_crcf_create_kaiser(ANALYZER, M, m, As)) ------------
// returns taps[c][n] (c = commutator position 0..M-1, n = 0..2m-1 frames back) such that
//   X_t[c] = sum_n taps[c][n] * x[(t - n) * M + c]        and    y_t[k] = sum_c X_t[c] exp(-j 2 pi k c / M)
inline std::vector<float> channelizer_taps(unsigned M, unsigned m, float as) {
    const unsigned p = 2 * m, hl = 2 * M * m + 1;
    std::vector<float> h = kaiser_lowpass(hl, 0.5f / (float)M, as);
    std::vector<float> t((size_t)M * p);
    for (unsigned c = 0; c < M; ++c)
        for (unsigned n = 0; n < p; ++n) t[(size_t)c * p + n] = h[(M - 1 - c) + n * M];
    return t;
}

// ---- 2x oversampled polyphase analysis channelizer (firpfbch2_crcf_create_kaiser(ANALYZER, M, m, As)) --------
// liquid 1.5.0 firpfbch2: prototype Kaiser low-pass of 2 M m + 1 taps with cut-off 1/M (twice the analyzer bandwidth of
// firpfbch), scaled to sum M; every execute() takes M/2 new samples and applies the first 2 M m taps, an M-point inverse
// DFT and a 1/M gain, with the branch order rotating by M/2 on alternate calls.  Written in the commutator form of
// channelizer_taps (pinned against the reference DLL's impulse responses to 1e-7):
//   X_t[c] = sum_n taps[c][n] x[(t - 1) M/2 + c - n M]
//   y_t[k] = post[t & 1][k] sum_c X_t[c] exp(-j 2 pi k c / M),   post[p][k] = (p ? (-1)^k : 1) exp(-j 2 pi k / M) / M
inline std::vector<float> channelizer2_taps(unsigned M, unsigned m, float as) {
    const unsigned p = 2 * m, hl = 2 * M * m + 1;
    std::vector<float> h = kaiser_lowpass(hl, 1.0f / (float)M, as);
    float sum = 0.0f;
    for (float v : h) sum += v;
    for (float &v : h) v = v * (float)M / sum;
    std::vector<float> t((size_t)M * p);
    for (unsigned c = 0; c < M; ++c)
        for (unsigned n = 0; n < p; ++n) t[(size_t)c * p + n] = h[(M - 1 - c) + n * M];
    return t;
}
// post[2][M] as (re, im) pairs
inline std::vector<float> channelizer2_post(unsigned M) {
    std::vector<float> t((size_t)4 * M);
    for (unsigned par = 0; par < 2; ++par)
        for (unsigned k = 0; k < M; ++k) {
            const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)M;
            const double s = (par && (k & 1)) ? -1.0 : 1.0;
            t[2 * ((size_t)par * M + k)] = (float)(s * std::cos(a) / (double)M);
            t[2 * ((size_t)par * M + k) + 1] = (float)(s * std::sin(a) / (double)M);
        }
    return t;
}

// ---- AM DC-blocking FIR (firfilt_rrrf_create_dc_blocker(m, As) -> liquid_firdes_notch(m, 0, As)) -----------
inline std::vector<float> dc_notch_taps(unsigned m, float as) {
    const unsigned n = 2 * m + 1;
    const float beta = kaiser_beta(as);
    std::vector<float> h(n);
    float scale = 0.0f;
    for (unsigned i = 0; i < n; ++i) {
        const float p = -1.0f;  // -cos(2 pi f0 (i - m)) with f0 = 0
        h[i] = p * kaiser_window(i, n, beta);
        scale += h[i] * p;
    }
    for (float &v : h) v /= scale;
    h[m] += 1.0f;
    return h;
}

// ---- SSB: Butterworth low-pass as second-order sections (iirfilt_crcf_create_lowpass(order, fc)) -----------
struct Sos { float b[3]; float a[3]; };
inline std::vector<Sos> butter_lowpass_sos(unsigned order, float fc) {
    const unsigned r = order % 2, L = (order - r) / 2;
    const float mm = 1.0f / std::tan(kPi * fc);
    std::vector<Sos> out;
    double gr = 1.0, gi = 0.0;
    std::vector<double> a1, a2;
    for (unsigned i = 0; i < L; ++i) {
        const float th = (float)(2 * (i + 1) + order - 1) * kPi / (float)(2 * order);
        double pr = std::cos(th) / mm, pi = std::sin(th) / mm;
        // digital pole pd = (1 + p) / (1 - p), and its conjugate
        double dr = 1.0 - pr, di = -pi, nr = 1.0 + pr, ni = pi, den = dr * dr + di * di;
        double qr = (nr * dr + ni * di) / den, qi = (ni * dr - nr * di) / den;
        // gain *= (1 - pd)(1 - conj pd) / 4
        double f = ((1.0 - qr) * (1.0 - qr) + qi * qi) / 4.0;
        gr *= f;
        a1.push_back(-2.0 * qr);
        a2.push_back(qr * qr + qi * qi);
    }
    double real_pole = 0.0;
    if (r) { double pr = -1.0 / mm; real_pole = (1.0 + pr) / (1.0 - pr); gr *= (1.0 - real_pole) / 2.0; }
    (void)gi;
    const float kg = std::pow((float)gr, 1.0f / (float)(L + r));
    // the reference orders its sections by ascending a2 (pinned by tests against the reference binary)
    for (unsigned i = 0; i < L; ++i)
        for (unsigned j = i + 1; j < L; ++j)
            if (a2[j] < a2[i]) { std::swap(a2[i], a2[j]); std::swap(a1[i], a1[j]); }
    for (unsigned i = 0; i < L; ++i) out.push_back(Sos{{kg, 2.0f * kg, kg}, {1.0f, (float)a1[i], (float)a2[i]}});
    if (r) out.push_back(Sos{{kg, kg, 0.0f}, {1.0f, (float)(-real_pole), 0.0f}});
    return out;
}

// impulse response of a cascade of second-order sections (direct form II, as iirfilt_crcf_execute runs them), n samples, in
// double: the SSB low-pass has poles of radius <= 0.77, so its response is below 1e-14 of the peak after 128 samples and the
// recursive filter IS (to far below float32 resolution) the FIR filter with these taps -- which runs in parallel
inline std::vector<float> sos_impulse_response(const std::vector<Sos> &sos, unsigned n) {
    std::vector<double> v1(sos.size(), 0.0), v2(sos.size(), 0.0);
    std::vector<float> g(n);
    for (unsigned i = 0; i < n; ++i) {
        double t = i == 0 ? 1.0 : 0.0;
        for (size_t q = 0; q < sos.size(); ++q) {
            const double v0 = t - (double)sos[q].a[1] * v1[q] - (double)sos[q].a[2] * v2[q];
            t = (double)sos[q].b[0] * v0 + (double)sos[q].b[1] * v1[q] + (double)sos[q].b[2] * v2[q];
            v2[q] = v1[q]; v1[q] = v0;
        }
        g[i] = (float)t;
    }
    return g;
}

// ---- FM stereo: 19 kHz pilot band-pass (iirfilt_crcf_create_prototype(CHEBY2, BANDPASS, SOS, 5, fc, f0, 1, 60), ModemFMStereo.cpp:128-139) ----
// liquid_iirdes of liquid 1.5.0 restated with its single-precision data flow (cheby2_azpkf -> bilinear_zpkf -> iirdes_dzpk_lp2bp ->
// iirdes_dzpk2sosf): the pole radii are ~0.998, so the pass-band phase moves by 1e-4 rad per unit in the last place of a feedback
// coefficient, and only the same roundings give the same filter.  Complex quotients of floats are formed the plain way
// ((a conj b) / |b|^2), the bilinear map runs in double (its "1.0" literals promote the operands); pinned bit-for-bit against the
// reference binary on 400 sample rates (feedback taps always; 3 of 400 feed-forward taps differ by one unit in the last place).
inline std::vector<Sos> cheby2_bandpass_sos(unsigned n, float fc, float f0, float as) {
    typedef std::complex<float> cf;
    typedef std::complex<double> cd;
    const double pi = 3.14159265358979323846;
    auto qf = [](cf a, cf b) { const float d = b.real() * b.real() + b.imag() * b.imag();
                               return cf((a.real() * b.real() + a.imag() * b.imag()) / d, (a.imag() * b.real() - a.real() * b.imag()) / d); };
    auto qd = [](cd a, cd b) { const double d = b.real() * b.real() + b.imag() * b.imag();
                               return cd((a.real() * b.real() + a.imag() * b.imag()) / d, (a.imag() * b.real() - a.real() * b.imag()) / d); };
    const unsigned r = n % 2, L = (n - r) / 2;
    const float es = std::pow(10.0f, -as / 20.0f);
    const float t0 = (float)std::sqrt(1.0 + 1.0 / ((double)es * (double)es));
    const float tp = std::pow((float)((double)t0 + 1.0 / (double)es), (float)(1.0 / (double)(float)n));
    const float tm = std::pow((float)((double)t0 - 1.0 / (double)es), (float)(1.0 / (double)(float)n));
    const float eb = (float)(0.5 * ((double)tp + (double)tm)), ea = (float)(0.5 * ((double)tp - (double)tm));
    std::vector<cf> pa, za;
    for (unsigned i = 0; i < L; ++i) {
        const float th = (float)((double)(float)(2 * (i + 1) + n - 1) * pi / (double)(float)(2 * n));
        pa.push_back(qf(cf(1.0f, 0.f), cf(ea * std::cos(th), -eb * std::sin(th))));
        pa.push_back(qf(cf(1.0f, 0.f), cf(ea * std::cos(th), eb * std::sin(th))));
    }
    if (r) pa.push_back(cf(-1.0f / ea, 0.f));
    for (unsigned i = 0; i < L; ++i) {
        const float th = (float)(0.5 * pi * (double)(2 * (i + 1) - 1) / (double)(float)n);
        za.push_back(qf(cf(-1.0f, 0.f), cf(0.f, std::cos(th))));
        za.push_back(qf(cf(1.0f, 0.f), cf(0.f, std::cos(th))));
    }
    const float m = std::fabs((std::cos((float)(2 * pi * (double)fc)) - std::cos((float)(2 * pi * (double)f0))) / std::sin((float)(2 * pi * (double)fc)));
    std::vector<cf> zd(n), pd(n);
    cf G(1.0f, 0.f);
    for (unsigned i = 0; i < n; ++i) {
        if (i < 2 * L) { const cf zm = za[i] * m; zd[i] = (cf)qd(cd(1.0) + (cd)zm, cd(1.0) - (cd)zm); } else zd[i] = cf(-1.0f, 0.f);
        const cf pm = pa[i] * m;
        pd[i] = (cf)qd(cd(1.0) + (cd)pm, cd(1.0) - (cd)pm);
        G = (cf)((cd)G * qd(cd(1.0) - (cd)pd[i], cd(1.0) - (cd)zd[i]));
    }
    const float c0 = std::cos((float)(2 * pi * (double)f0));
    auto lp2bp = [&](const std::vector<cf> &v) {
        std::vector<cf> o;
        for (const cf &z : v) {
            const cf t = cf(1.0f) + z, sq = std::sqrt(c0 * c0 * t * t - 4.0f * z);
            o.push_back(0.5f * (c0 * t + sq));
            o.push_back(0.5f * (c0 * t - sq));
        }
        return o;
    };
    // liquid_cplxpair: conjugate pairs (negative imaginary part first) by ascending real part, then the purely real values
    auto pairup = [](std::vector<cf> v) {
        const float tol = 1e-6f;
        const size_t N = v.size();
        std::vector<bool> used(N, false);
        std::vector<cf> o;
        for (size_t i = 0; i < N; ++i) {
            if (used[i] || std::fabs(v[i].imag()) < tol) continue;
            for (size_t j = 0; j < N; ++j) {
                if (j == i || used[j] || std::fabs(v[j].imag()) < tol) continue;
                if (std::fabs(v[i].imag() + v[j].imag()) < tol && std::fabs(v[i].real() - v[j].real()) < tol) {
                    o.push_back(v[i]); o.push_back(v[j]); used[i] = used[j] = true;
                    break;
                }
            }
        }
        const size_t np = o.size() / 2;
        for (size_t i = 0; i < N; ++i) if (!used[i]) o.push_back(v[i]);
        for (size_t i = 0; i < np; ++i) if (o[2 * i].imag() > 0) std::swap(o[2 * i], o[2 * i + 1]);
        for (size_t i = 0; i < np; ++i)
            for (size_t j = i + 1; j < np; ++j)
                if (o[2 * j].real() < o[2 * i].real()) { std::swap(o[2 * i], o[2 * j]); std::swap(o[2 * i + 1], o[2 * j + 1]); }
        for (size_t i = 2 * np; i < N; ++i)
            for (size_t j = i + 1; j < N; ++j)
                if (o[j].real() < o[i].real()) std::swap(o[i], o[j]);
        return o;
    };
    const std::vector<cf> zp = pairup(lp2bp(zd)), pp = pairup(lp2bp(pd));
    const float kg = std::pow(G.real(), 1.0f / (float)n);
    std::vector<Sos> out;
    for (unsigned i = 0; i < n; ++i) {
        const cf p0 = -pp[2 * i], p1 = -pp[2 * i + 1], z0 = -zp[2 * i], z1 = -zp[2 * i + 1];
        Sos q;
        q.a[0] = 1.0f; q.a[1] = (p0 + p1).real(); q.a[2] = (p0 * p1).real();
        q.b[0] = kg; q.b[1] = (z0 + z1).real() * kg; q.b[2] = (z0 * z1).real() * kg;
        out.push_back(q);
    }
    return out;
}
inline std::vector<Sos> fms_pilot_sos(int64_t sample_rate) {          // ModemFMStereo.cpp:128-139
    float bw = (float)sample_rate;
    if (bw < 100000.0f) bw = 100000.0f;
    return cheby2_bandpass_sos(5, (float)19500 / bw, (float)19000 / bw, 60.0f);
}

// ---- FM stereo: output filter of one channel = de-emphasis (iirfilt_rrrf, one pole: ModemFMStereo.cpp:147-159) followed by the
// 16 kHz Kaiser low-pass (firfilt_rrrf, :107-126), as ONE impulse response (both are LTI and start from rest).  The pole of the
// de-emphasis is at most 0.9 in magnitude at the rates the reference runs (75 us at 96 kHz: -0.87), so its response is cut where
// it falls below 1e-10; demph_us == 0 leaves the low-pass alone.
inline std::vector<float> fms_output_fir(int audio_rate, int demph_us, unsigned max_taps) {
    const float as = 60.0f;
    float fcut = 16000.0f / (float)audio_rate;
    const float ft = 1000.0f / (float)audio_rate;
    if (fcut < 0) fcut = 0;
    if (fcut > 0.5f) fcut = 0.5f;
    const unsigned h_len = required_filter_len(ft, as);
    const std::vector<float> h = kaiser_lowpass(h_len, fcut, as);
    if (!demph_us) return h.size() <= max_taps ? h : std::vector<float>();
    const double f = 1.0 / (2.0 * M_PI * (double)demph_us * 1e-6);
    double t = 1.0 / (2.0 * M_PI * f);
    t = 1.0 / (2.0 * (double)audio_rate * std::tan(1.0 / (2.0 * (double)audio_rate * t)));
    const double tb = 1.0 + 2.0 * t * (double)audio_rate;
    const float b0 = (float)(1.0 / tb), b1 = (float)(1.0 / tb), a1 = (float)((1.0 - 2.0 * t * (double)audio_rate) / tb);
    std::vector<double> dm;
    double v1 = 0.0;
    for (unsigned i = 0; i < 4096; ++i) {                 // direct form II: v0 = x - a1 v1;  y = b0 v0 + b1 v1
        const double v0 = (i == 0 ? 1.0 : 0.0) - (double)a1 * v1;
        dm.push_back((double)b0 * v0 + (double)b1 * v1);
        v1 = v0;
        if (i > 8 && std::fabs(v0) < 1e-10) break;
    }
    if (h.size() + dm.size() - 1 > max_taps) return std::vector<float>();
    std::vector<double> g(h.size() + dm.size() - 1, 0.0);
    for (size_t i = 0; i < h.size(); ++i)
        for (size_t k = 0; k < dm.size(); ++k) g[i + k] += (double)h[i] * dm[k];
    return std::vector<float>(g.begin(), g.end());
}

// ---- SSB: Hilbert transformer taps (firhilbf_create(m, As)); h[n] for odd delays n = 1, 3, .., 4m-1 ---------
// y_q[k] = sum_{n odd} hq[(n-1)/2] * imag(x[k - n]),  y_i[k] = real(x[k - 2m])
inline std::vector<float> hilbert_taps(unsigned m, float as) {
    const unsigned n = 4 * m + 1;
    std::vector<float> h = kaiser_lowpass(n, 0.25f, std::fabs(as));
    std::vector<float> hq;
    for (unsigned i = 1; i < n; i += 2) {
        const float t = (float)i - (float)(n - 1) / 2.0f;
        hq.push_back(h[i] * std::sin(0.5f * kPi * t));
    }
    return hq;
}

// ---- block / channel sizing rules of the reference's SDR thread (SoapySDRThread.cpp:668-693) ---------------
inline int optimal_channel_count(int64_t sample_rate) {
    if (sample_rate <= 500000) return 1;
    int64_t c = (int64_t)std::ceil((double)sample_rate / 500000.0);
    if (c % 2 == 1) c--;
    if (c < 2) c = 2;
    return (int)c;
}
inline int optimal_element_count(int64_t sample_rate, int fps, int num_channels) {
    int n = (int)std::floor((double)sample_rate / (double)fps);
    n = (int)(std::ceil((double)n / (double)num_channels) * num_channels);
    return n;
}

}  // namespace design
}  // namespace csdr
