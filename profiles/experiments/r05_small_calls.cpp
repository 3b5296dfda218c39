// One-block calls from a C++ host (measurement helper, not the product): the C3 pipeline through include/csdr_hip.h alone, ONE 1/60 s block per call, as the
// reference hands them over -- from one host thread, and from two (channelizer + demodulators | spectrum: the reference's own thread cut, which
// cubicsdr_amd/host/HipPipeline.h mirrors).  bench.py measures the same calls through the Python binding, whose own overhead is part of its figure.
//   hipcc -O2 -std=c++17 profiles/experiments/r05_small_calls.cpp -Iinclude -Lcubicsdr_amd -lcsdr_hip -Wl,-rpath,$PWD/cubicsdr_amd -lpthread -o /tmp/small_calls
#include <hip/hip_runtime_api.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>
#include "csdr_hip.h"

#define OK(x) do { int rc__ = (x); if (rc__ != 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc__, csdr_last_error()); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int64_t FS = 61440000, CENTER = 100000000;
    const int M = 122, BLOCK = 1024068, ND = 256, F = 65536, calls = argc > 1 ? atoi(argv[1]) : 600;
    std::vector<float> h((size_t)2 * BLOCK);
    std::mt19937 g(1); std::normal_distribution<float> nd(0.f, 0.05f);
    for (auto &v : h) v = nd(g);
    float *x = nullptr;
    if (hipMalloc((void **)&x, h.size() * sizeof(float)) != hipSuccess) return 2;
    (void)hipMemcpy(x, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    csdr_ctx *c; csdr_post *p; csdr_bank *b; csdr_spec *s;
    OK(csdr_ctx_create(0, nullptr, &c));
    OK(csdr_post_create(c, &p));
    OK(csdr_post_configure(p, FS, M, CSDR_POST_PFBCH, BLOCK, 1));
    OK(csdr_bank_create(c, ND, 1, &b));
    const int modem[3] = {CSDR_MODEM_NBFM, CSDR_MODEM_AM, CSDR_MODEM_USB}, bw[3] = {12500, 6000, 5400};
    for (int i = 0; i < ND; ++i) {
        csdr_demod_params prm = {modem[i % 3], bw[i % 3], 48000, 0, (int64_t)(CENTER + (i + 0.37) * (double)FS / ND - FS / 2)};
        OK(csdr_bank_configure_slot(b, i, &prm, p));
    }
    OK(csdr_spec_create(c, &s));
    OK(csdr_spec_setup(s, F, BLOCK / (2 * F) + 2));
    auto demod_side = [&](int n) { for (int i = 0; i < n; ++i) { OK(csdr_post_execute(p, x, 1, 1, BLOCK, CENTER)); OK(csdr_bank_execute(b, p)); } };
    auto spec_side = [&](int n) { for (int i = 0; i < n; ++i) OK(csdr_spec_process(s, x, 1, 1, BLOCK, CSDR_SPEC_CONTIGUOUS)); };
    demod_side(20); spec_side(20);
    OK(csdr_ctx_synchronize(c));
    using clk = std::chrono::steady_clock;
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = clk::now();
        for (int i = 0; i < calls; ++i) { demod_side(1); spec_side(1); }
        OK(csdr_ctx_synchronize(c));
        const double one = std::chrono::duration<double>(clk::now() - t0).count();
        t0 = clk::now();
        std::thread ta(demod_side, calls), tb(spec_side, calls);
        ta.join(); tb.join();
        OK(csdr_ctx_synchronize(c));
        const double two = std::chrono::duration<double>(clk::now() - t0).count();
        printf("{\"calls\": %d, \"one_thread_calls_per_s\": %.0f, \"one_thread_us_per_call\": %.1f, \"two_threads_calls_per_s\": %.0f, \"two_threads_us_per_call\": %.1f}\n",
               calls, calls / one, 1e6 * one / calls, calls / two, 1e6 * two / calls);
    }
    return 0;
}
