// design_capi.cpp -- host-only C entry points over design.hpp so the cold-path filter design of the product can be
// checked on a machine without a GPU (tests/test_host_design.py).  Built by cubicsdr_amd/build.py with g++ into
// cubicsdr_amd/libcsdr_design.so; not part of the GPU library's ABI.
#include <cstring>

#include "design.hpp"

using namespace csdr::design;

extern "C" {

int csdr_design_msresamp(float rate, float as, int *interp, int *S, unsigned *step, float *rate_arb, int *m /*[16]*/,
                         float *h1 /*[16][20]*/, float *arms /*[256*14]*/, int *pinned) {
    MsresampPlan p = plan_msresamp(rate, as);
    *interp = p.interp ? 1 : 0; *S = (int)p.S; *step = p.step; *rate_arb = p.rate_arb; *pinned = p.pinned ? 1 : 0;
    if (p.S > 16) return -1;
    for (unsigned i = 0; i < p.S; ++i) {
        m[i] = (int)p.m[i];
        for (unsigned j = 0; j < 2 * p.m[i] && j < 20; ++j) h1[i * 20 + j] = p.h1[i][j];
    }
    std::memcpy(arms, p.arms.data(), p.arms.size() * sizeof(float));
    return 0;
}
unsigned long long csdr_design_resamp_count(unsigned long long K, unsigned phase, unsigned step, unsigned *phase_after) {
    return resamp_count(K, phase, step, phase_after);
}
unsigned csdr_design_nco_word(float theta) { return nco_phase_word(theta); }
void csdr_design_sine_table(float *t) { auto v = nco_sine_table(); std::memcpy(t, v.data(), 1024 * sizeof(float)); }
void csdr_design_channelizer(unsigned M, unsigned m, float as, float *taps) { auto v = channelizer_taps(M, m, as); std::memcpy(taps, v.data(), v.size() * sizeof(float)); }
void csdr_design_dc_notch(unsigned m, float as, float *h) { auto v = dc_notch_taps(m, as); std::memcpy(h, v.data(), v.size() * sizeof(float)); }
int csdr_design_butter_sos(unsigned order, float fc, float *b, float *a) {
    auto v = butter_lowpass_sos(order, fc);
    for (size_t i = 0; i < v.size(); ++i) for (int k = 0; k < 3; ++k) { b[3 * i + k] = v[i].b[k]; a[3 * i + k] = v[i].a[k]; }
    return (int)v.size();
}
int csdr_design_fms_pilot_sos(long long sample_rate, float *b, float *a) {
    auto v = fms_pilot_sos(sample_rate);
    for (size_t q = 0; q < v.size(); ++q) for (int k = 0; k < 3; ++k) { b[3 * q + k] = v[q].b[k]; a[3 * q + k] = v[q].a[k]; }
    return (int)v.size();
}
int csdr_design_fms_output_fir(int audio_rate, int demph_us, float *g, unsigned cap) {
    auto v = fms_output_fir(audio_rate, demph_us, cap);
    std::memcpy(g, v.data(), v.size() * sizeof(float));
    return (int)v.size();
}
void csdr_design_hilbert(unsigned m, float as, float *hq) { auto v = hilbert_taps(m, as); std::memcpy(hq, v.data(), v.size() * sizeof(float)); }
int csdr_design_channel_count(long long rate) { return optimal_channel_count(rate); }
int csdr_design_element_count(long long rate, int fps, int nch) { return optimal_element_count(rate, fps, nch); }

}  // extern "C"
