// csdr_spec.hip -- implementation of include/csdr_hip.h (gfx950): csdr_spec (SpectrumVisualProcessor).  Host-side bookkeeping mirrors the reference's control flow
// (file:line cited per function); all sample arithmetic is in the kernels_*.hpp kernels.
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <map>
#include <memory>

#define CSDR_TU_SPEC 1          // this unit is the home of its kernels (common.hpp)
#include "csdr_objects.hpp"
#include "kernels_spec.hpp"
#include "kernels_spec2.hpp"
#include "kernels_spec3.hpp"

using namespace csdr;

// =================================================================================================== spectrum
struct csdr_spec {
    csdr_ctx *ctx = nullptr;
    bool ready = false;
    SpecGeom g{};
    int max_frames = 0, nf_last = 0;
    float avg_rate = 0.65f, scale = 1.0f;
    DevBuf<float2> tw4096, tw_hi, tw_lo, tmp, carry, stage_in, raw;
    DevBuf<float2> blue_w, blue_B;           // sizes that are not powers of two: chirp w[N] and the transformed chirp filter Bf[L]
    int blue_L = 0;
    // ... with a convolution longer than 4096 points (fftSize above 1024): the two L-point transforms run as the power-of-two chain of size L
    bool blue_big = false;
    int blue_chunk = 1;                      // frames per pass of the big path
    SpecGeom gL{};
    DevBuf<float2> blue_a, blue_b, twL_hi, twL_lo;
    DevBuf<float> mag;                       // [2][max_frames][N]: the FFT lane fills one copy while the averaging lane reads the other
    DevBuf<float> pairsum, first_b, points;
    uint64_t seq = 0;
    hipEvent_t ev_fft_done[2] = {nullptr, nullptr}, ev_avg_done[2] = {nullptr, nullptr};
    bool avg_pending[2] = {false, false};
    const float2 *carry_fold_src = nullptr;  // contiguous mode, fused chain: the samples behind the last whole frame, copied to `carry` by the display launch (> 0: pending, -1: done)
    int carry_fold_n = 0;
    int tmp_reader = -1;                     // magnitude copy whose ev_avg_done also covers a reader of `tmp` on lane AVG (the fused chain's row pass), -1: none
    DevBuf<double> ma, maa;
    DevBuf<float2> ext_w, ext;
    int n_avg_tiles = 0, scal_parity = 0;
    DevBuf<SpecFrameOut> fo;
    DevBuf<SpecFrameScal> fsc;                      // per frame: point_ceil, point_floor, fft_floor_maa (spec_trackers -> spec_display)
    DevBuf<SpecScalars> scal;
    int carry_len = 0;
    // CSDR_SPEC_LINES: fftLastData (the previous FFT input, :399-421) in two copies written alternately, lastDataSize != 0
    DevBuf<float2> last[2], lines;
    int last_cur = 0;
    bool last_primed = false;
    // peak hold (:247-273): peakHold / peakReset as in the reference; device state allocated when first enabled
    bool peak_hold = false;
    int peak_reset = 0;
    DevBuf<double> peak;                     // fft_result_peak, pair layout like ma / maa
    DevBuf<float2> maaf;
    DevBuf<float> peaksum, peak_b, hold_points;
    DevBuf<SpecPeakScalars> pk;
    DevBuf<SpecFrameOut> pfo;
    std::vector<char> hold_valid;            // per frame of the last process: spectrum_hold_points present
    // hideDC (:578-623) and the frequencies it needs
    bool hide_dc = false;
    int64_t center_freq = 0, input_freq = 0;
    long bandwidth = 0;
    // zoomed view (setView :64-72; process :283-386, :454-492, :532-560)
    bool is_view = false, last_view = false, have_resampler = false;
    int64_t input_rate = 0;
    long last_bandwidth = 0, last_input_bandwidth = 0, shift_frequency = 0, resample_bw = 0;   // ctor :11-15, :30
    int desired_input_size = 0;
    csdr_post *vpost = nullptr;              // raw single-channel hand-over of the input block
    csdr_bank *vbank = nullptr;              // one front-end-only slot: NCO shift + msresamp_crcf
    int vpost_cap = 0;
    DevBuf<double> ma2, maa2;                // target of an averager remap (swapped with ma / maa afterwards)
    DevBuf<int2> vmap;                       // (first bin, bins) per display point for the current visualRatio
    long vmap_bw = -1, vmap_rbw = -1;
    DevBuf<float2> peakf;
    bool view_frame = false;                 // the frames being post-processed belong to the zoomed view
    bool fused_ok = false;                   // N = 2^17: the 512 x 256 chain with the averaging fused into the row pass exists (kernels_spec3.hpp)
    bool fused_now = false;                  // ... and this batch takes it (full-span view, no peak hold set or pending)
};

extern "C" int csdr_spec_create(csdr_ctx *ctx, csdr_spec **out) {
    DeviceScope dev__(ctx);
    if (!ctx || !out) return fail(CSDR_EINVAL, "null argument");
    std::unique_ptr<csdr_spec> s(new csdr_spec());
    s->ctx = ctx;
    for (int k = 0; k < 2; ++k) {
        CSDR_HIP_TRY(hipEventCreateWithFlags(&s->ev_fft_done[k], hipEventDisableTiming));
        CSDR_HIP_TRY(hipEventCreateWithFlags(&s->ev_avg_done[k], hipEventDisableTiming));
    }
    *out = s.release();
    return CSDR_OK;
}
extern "C" void csdr_spec_destroy(csdr_spec *s) {
    DeviceScope dev__(s ? s->ctx : nullptr);
    if (!s) return;
    (void)s->ctx->sync_all();
    for (int k = 0; k < 2; ++k) {
        if (s->ev_fft_done[k]) (void)hipEventDestroy(s->ev_fft_done[k]);
        if (s->ev_avg_done[k]) (void)hipEventDestroy(s->ev_avg_done[k]);
    }
    s->tw4096.release(); s->tw_hi.release(); s->tw_lo.release(); s->tmp.release(); s->carry.release();
    s->blue_w.release(); s->blue_B.release(); s->blue_a.release(); s->blue_b.release(); s->twL_hi.release(); s->twL_lo.release();
    s->stage_in.release(); s->raw.release(); s->mag.release(); s->ext_w.release(); s->ext.release(); s->pairsum.release(); s->first_b.release(); s->points.release();
    s->ma.release(); s->maa.release(); s->fo.release(); s->fsc.release(); s->scal.release();
    s->last[0].release(); s->last[1].release(); s->lines.release();
    s->peak.release(); s->maaf.release(); s->peaksum.release(); s->peak_b.release(); s->hold_points.release(); s->pk.release(); s->pfo.release();
    s->ma2.release(); s->maa2.release(); s->vmap.release(); s->peakf.release();
    if (s->vbank) csdr_bank_destroy(s->vbank);
    if (s->vpost) csdr_post_destroy(s->vpost);
    delete s;
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

extern "C" int csdr_spec_setup(csdr_spec *s, int fft_size, int max_frames) {
    DeviceScope dev__(s ? s->ctx : nullptr);
    if (!s) return fail(CSDR_EINVAL, "spec is null");
    if (fft_size < 2) return fail(CSDR_EINVAL, "fft_size %d", fft_size);
    const bool npot = (fft_size & (fft_size - 1)) != 0;
    if (max_frames <= 0) return fail(CSDR_EINVAL, "max_frames");
    const int N = 2 * fft_size;                                      // SPECTRUM_VZM 2, SpectrumVisualProcessor.h:11, .cpp:145
    if (N > (1 << 22) || (npot && 2 * (int64_t)N - 1 > (1 << 22))) return fail(CSDR_EUNSUPPORTED, "internal FFT of %d points exceeds 2^22%s", N, npot ? " (the chirp-z convolution of a size that is not a power of two is twice as long)" : "");
    if (int rc = s->ctx->sync_all()) return rc;
    s->ready = false;
    s->seq = 0; s->avg_pending[0] = s->avg_pending[1] = false; s->tmp_reader = -1;
    SpecGeom &g = s->g;
    g.N = N; g.F = fft_size; g.Ra = 1; g.Rb = 1; g.N2 = N; g.npot = npot ? 1 : 0;
    s->blue_L = 0; s->blue_big = false;
    if (npot) {
        // chirp-z tables, in double: w[n] = exp(-i pi n^2 / N) (n^2 taken mod 2 N, exactly) and Bf = FFT_L(b), b[m mod L] = conj(w[|m|]), |m| < N
        int L = 1;
        while (L < 2 * N - 1) L <<= 1;
        s->blue_L = L;
        std::vector<std::pair<double, double>> w((size_t)N), b((size_t)L, {0.0, 0.0});
        for (int n = 0; n < N; ++n) {
            const long long q = ((long long)n * n) % (2LL * N);
            const double a = -M_PI * (double)q / (double)N;
            w[(size_t)n] = {std::cos(a), std::sin(a)};
        }
        for (int m = 0; m < N; ++m) { b[(size_t)m] = {w[(size_t)m].first, -w[(size_t)m].second}; if (m) b[(size_t)(L - m)] = b[(size_t)m]; }
        // iterative radix-2 transform of b in double (one-time; the twiddles from ONE table of L / 2 entries: L reaches 2^22)
        {
            std::vector<std::pair<double, double>> tw((size_t)L / 2);
            for (int k = 0; k < L / 2; ++k) { const double a = -2.0 * M_PI * (double)k / (double)L; tw[(size_t)k] = {std::cos(a), std::sin(a)}; }
            for (int i = 1, j = 0; i < L; ++i) { int bit = L >> 1; for (; j & bit; bit >>= 1) j ^= bit; j ^= bit; if (i < j) std::swap(b[(size_t)i], b[(size_t)j]); }
            for (int len = 2; len <= L; len <<= 1) {
                const int step = L / len;
                for (int i = 0; i < L; i += len)
                    for (int k = 0; k < len / 2; ++k) {
                        const double c = tw[(size_t)k * step].first, sn = tw[(size_t)k * step].second;
                        const auto u = b[(size_t)(i + k)], v = b[(size_t)(i + k + len / 2)];
                        const double vr = v.first * c - v.second * sn, vi = v.first * sn + v.second * c;
                        b[(size_t)(i + k)] = {u.first + vr, u.second + vi}; b[(size_t)(i + k + len / 2)] = {u.first - vr, u.second - vi};
                    }
            }
        }
        std::vector<float2> wf((size_t)N), bf((size_t)L);
        for (int n = 0; n < N; ++n) wf[(size_t)n] = make_float2((float)w[(size_t)n].first, (float)w[(size_t)n].second);
        for (int i = 0; i < L; ++i) bf[(size_t)i] = make_float2((float)b[(size_t)i].first, (float)b[(size_t)i].second);
        if (int rc = s->blue_w.reserve((size_t)N)) return rc;
        if (int rc = s->blue_B.reserve((size_t)L)) return rc;
        CSDR_HIP_TRY(hipMemcpy(s->blue_w.p, wf.data(), wf.size() * sizeof(float2), hipMemcpyHostToDevice));
        CSDR_HIP_TRY(hipMemcpy(s->blue_B.p, bf.data(), bf.size() * sizeof(float2), hipMemcpyHostToDevice));
        s->blue_big = L > kTwTab;                                     // the LDS transforms of spec_fft_bluestein reach 4096 points
        if (!s->blue_big && (size_t)2 * L * sizeof(float2) > 64 * 1024)
            CSDR_HIP_TRY(hipFuncSetAttribute((const void *)spec_fft_bluestein, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)2 * L * sizeof(float2))));
        if (s->blue_big) {
            // the L-point transforms as the power-of-two chain of that size (radix passes + 4096-point rows, natural-order complex output)
            SpecGeom &q = s->gL;
            q = SpecGeom{};
            q.N = L; q.F = L / 2; q.N2 = 4096; q.npot = 0;
            const int R = L / 4096;
            q.Ra = std::min(R, 32); q.Rb = R / q.Ra;
            q.lgRa = ilog2(q.Ra); q.lgRb = ilog2(q.Rb);
            std::vector<float2> lo(1024), hi((size_t)L / 1024);
            for (int i = 0; i < 1024; i++) { double a = -2.0 * M_PI * i / L; lo[i] = make_float2((float)std::cos(a), (float)std::sin(a)); }
            for (size_t i = 0; i < hi.size(); i++) { double a = -2.0 * M_PI * (double)(i * 1024) / L; hi[i] = make_float2((float)std::cos(a), (float)std::sin(a)); }
            if (int rc = s->twL_lo.reserve(1024)) return rc;
            if (int rc = s->twL_hi.reserve(hi.size())) return rc;
            CSDR_HIP_TRY(hipMemcpy(s->twL_lo.p, lo.data(), 1024 * sizeof(float2), hipMemcpyHostToDevice));
            CSDR_HIP_TRY(hipMemcpy(s->twL_hi.p, hi.data(), hi.size() * sizeof(float2), hipMemcpyHostToDevice));
            s->blue_chunk = std::max(1, std::min(max_frames, (1 << 26) / L));
            const size_t nfL = (size_t)s->blue_chunk * (size_t)L;
            if (int rc = s->blue_a.reserve(nfL)) return rc;
            if (int rc = s->blue_b.reserve(nfL)) return rc;
            if (int rc = s->tmp.reserve(nfL)) return rc;
        }
    }
    if (!s->blue_big) { s->blue_a.release(); s->blue_b.release(); }      // (a smaller size after a large one that was not a power of two)
    if (N >= 4096 && !npot) {
        g.N2 = 4096;
        const int R = N / 4096;                                       // 1 .. 1024
        g.Ra = std::min(R, 32); g.Rb = R / g.Ra;
        // 2^21 points: ONE 512-point column pass through LDS (spec_cols512, kernels_spec2.hpp) instead of a radix-32 and a radix-16 pass through
        // HBM -- 16 instead of 32 B/sample in front of the 4096-point rows; bins k = k1 + 512 k3, row = k1 (the layout of Ra = 512, Rb = 1)
        if (N == kC512 * 4096 && lab_int("CSDR_SPEC_COLS512", 1) != 0) { g.Ra = kC512; g.Rb = 1; }
    }
    g.lgRa = ilog2(g.Ra); g.lgRb = ilog2(g.Rb);
    s->max_frames = max_frames;
    std::vector<float2> t(kTwTab);
    for (int i = 0; i < kTwTab; i++) { double a = -2.0 * M_PI * i / kTwTab; t[i] = make_float2((float)std::cos(a), (float)std::sin(a)); }
    if (int rc = s->tw4096.reserve(kTwTab)) return rc;
    CSDR_HIP_TRY(hipMemcpy(s->tw4096.p, t.data(), kTwTab * sizeof(float2), hipMemcpyHostToDevice));
    std::vector<float2> lo(1024), hi(std::max(1, N / 1024));
    for (int i = 0; i < 1024; i++) { double a = -2.0 * M_PI * i / N; lo[i] = make_float2((float)std::cos(a), (float)std::sin(a)); }
    for (size_t i = 0; i < hi.size(); i++) { double a = -2.0 * M_PI * (double)(i * 1024) / N; hi[i] = make_float2((float)std::cos(a), (float)std::sin(a)); }
    if (int rc = s->tw_lo.reserve(1024)) return rc;
    if (int rc = s->tw_hi.reserve(hi.size())) return rc;
    CSDR_HIP_TRY(hipMemcpy(s->tw_lo.p, lo.data(), 1024 * sizeof(float2), hipMemcpyHostToDevice));
    CSDR_HIP_TRY(hipMemcpy(s->tw_hi.p, hi.data(), hi.size() * sizeof(float2), hipMemcpyHostToDevice));
    const size_t nfN = (size_t)max_frames * N, F = (size_t)g.F;
    if (g.Ra == kC512) CSDR_HIP_TRY(hipFuncSetAttribute((const void *)spec_cols512, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kC512Lds));
    // the headline size: two passes with the averaging fused into the second (30 instead of 38 B per sample; CSDR_SPEC_FUSED=0 of the measurement
    // build keeps the three-kernel chain for A/B runs)
    s->fused_ok = N == kS3N && !npot && lab_int("CSDR_SPEC_FUSED", 1) != 0;
    if (s->fused_ok) {
        CSDR_HIP_TRY(hipFuncSetAttribute((const void *)spec_cols512p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kP1Lds));
        CSDR_HIP_TRY(hipFuncSetAttribute((const void *)spec_rows256_ema<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kR2Lds));
        CSDR_HIP_TRY(hipFuncSetAttribute((const void *)spec_rows256_ema<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kR2Lds));
    }
    if (g.Ra > 1 && disp_lds_bytes((g.Ra >> 1) * g.Rb, ilog2(g.Ra) - 1 + ilog2(g.Rb), true) > 64 * 1024)      // the display tiles of 2^21-point frames with peak hold
        CSDR_HIP_TRY(hipFuncSetAttribute((const void *)spec_display<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)disp_lds_bytes((g.Ra >> 1) * g.Rb, ilog2(g.Ra) - 1 + ilog2(g.Rb), true)));
    if (g.Ra > 1 && !s->blue_big) if (int rc = s->tmp.reserve(nfN)) return rc;
    if (int rc = s->mag.reserve(2 * nfN)) return rc;
    s->n_avg_tiles = (g.F + kAvgLanes - 1) / kAvgLanes;
    if (int rc = s->ext_w.reserve((size_t)max_frames * s->n_avg_tiles)) return rc;
    if (int rc = s->ext.reserve(max_frames)) return rc;
    if (int rc = s->pairsum.reserve(nfN / 2)) return rc;
    if (int rc = s->first_b.reserve(max_frames)) return rc;
    if (int rc = s->points.reserve(nfN / 2)) return rc;           // F floats per frame: the y of every point (x = i / F is filled in by the fetch)
    if (int rc = s->ma.reserve(2 * F)) return rc;
    if (int rc = s->maa.reserve(2 * F)) return rc;
    if (int rc = s->fo.reserve(max_frames)) return rc;
    if (int rc = s->fsc.reserve(max_frames)) return rc;
    if (int rc = s->scal.reserve(2)) return rc;
    if (int rc = s->carry.reserve(N)) return rc;
    CSDR_HIP_TRY(hipMemset(s->ma.p, 0, 2 * F * sizeof(double)));      // vector<double>::resize -> zeros (:243-257)
    CSDR_HIP_TRY(hipMemset(s->maa.p, 0, 2 * F * sizeof(double)));
    SpecScalars sc = {100.0, 100.0, 0.0, 0.0};                     // ctor :32-33
    CSDR_HIP_TRY(hipMemcpy(s->scal.p, &sc, sizeof sc, hipMemcpyHostToDevice));
    CSDR_HIP_TRY(hipMemcpy(s->scal.p + 1, &sc, sizeof sc, hipMemcpyHostToDevice));
    s->scal_parity = 0;
    s->carry_len = 0; s->nf_last = 0;
    s->last_cur = 0; s->last_primed = false;                         // lastDataSize = 0 (:166)
    s->peak.release(); s->maaf.release(); s->peaksum.release(); s->peak_b.release(); s->hold_points.release();   // sized per fft size
    if (s->peak_hold) s->peak_reset = 1;                              // fft_result_peak is rebuilt (:261): nothing held until a reset has run
    s->ma2.release(); s->maa2.release(); s->vmap.release(); s->peakf.release(); s->vmap_bw = s->vmap_rbw = -1;
    s->ready = true;
    return CSDR_OK;
}
extern "C" int csdr_spec_set_average_rate(csdr_spec *s, float r) { if (!s) return fail(CSDR_EINVAL, "null"); s->avg_rate = r; return CSDR_OK; }
extern "C" int csdr_spec_set_scale_factor(csdr_spec *s, float f) { if (!s) return fail(CSDR_EINVAL, "null"); s->scale = f; return CSDR_OK; }
extern "C" int csdr_spec_frames(const csdr_spec *s) { return s ? s->nf_last : 0; }
extern "C" int csdr_spec_set_peak_hold(csdr_spec *s, int enabled) {      // setPeakHold :115-125
    if (!s) return fail(CSDR_EINVAL, "null");
    if (s->peak_hold && enabled) s->peak_reset = 30;                      // PEAK_RESET_COUNT (.h:12)
    else { s->peak_hold = enabled != 0; s->peak_reset = 1; }
    return CSDR_OK;
}
extern "C" int csdr_spec_get_peak_hold(const csdr_spec *s) { return s && s->peak_hold ? 1 : 0; }
extern "C" int csdr_spec_set_hide_dc(csdr_spec *s, int enabled) { if (!s) return fail(CSDR_EINVAL, "null"); s->hide_dc = enabled != 0; return CSDR_OK; }
extern "C" int csdr_spec_set_center_frequency(csdr_spec *s, int64_t f) { if (!s) return fail(CSDR_EINVAL, "null"); s->center_freq = f; return CSDR_OK; }
extern "C" int csdr_spec_set_bandwidth(csdr_spec *s, int64_t bw) { if (!s) return fail(CSDR_EINVAL, "null"); s->bandwidth = (long)bw; return CSDR_OK; }
extern "C" int csdr_spec_set_input_frequency(csdr_spec *s, int64_t f) { if (!s) return fail(CSDR_EINVAL, "null"); s->input_freq = f; return CSDR_OK; }
extern "C" int csdr_spec_set_input_rate(csdr_spec *s, int64_t rate) { if (!s) return fail(CSDR_EINVAL, "null"); s->input_rate = rate; return CSDR_OK; }
extern "C" int csdr_spec_set_view(csdr_spec *s, int is_view) { if (!s) return fail(CSDR_EINVAL, "null"); s->is_view = is_view != 0; return CSDR_OK; }
extern "C" int csdr_spec_get_view(const csdr_spec *s) { return s && s->is_view ? 1 : 0; }
extern "C" int csdr_spec_desired_input_size(const csdr_spec *s) {        // getDesiredInputSize :133-137
    if (!s || !s->ready) return 0;
    return s->is_view && s->desired_input_size ? s->desired_input_size : s->g.N;
}

template <int COLS>
static void launch_radix(csdr_ctx *c, int R, const FrameSrc &fs, int L, unsigned tw_scale, int nseq, const float2 *hi, const float2 *lo, float2 *dst) {
    const int Lr = L / R;
    const dim3 grid((Lr / COLS + kFftThreads - 1) / kFftThreads, nseq), block(kFftThreads);
    switch (R) {
        case 2: CSDR_LAUNCH(c, LANE_FFT, KID_FFT_COLS, (spec_fft_radix<2, COLS>), grid, block, 0, fs, L, tw_scale, hi, lo, dst); break;
        case 4: CSDR_LAUNCH(c, LANE_FFT, KID_FFT_COLS, (spec_fft_radix<4, COLS>), grid, block, 0, fs, L, tw_scale, hi, lo, dst); break;
        case 8: CSDR_LAUNCH(c, LANE_FFT, KID_FFT_COLS, (spec_fft_radix<8, COLS>), grid, block, 0, fs, L, tw_scale, hi, lo, dst); break;
        case 16: CSDR_LAUNCH(c, LANE_FFT, KID_FFT_COLS, (spec_fft_radix<16, COLS>), grid, block, 0, fs, L, tw_scale, hi, lo, dst); break;
        default: CSDR_LAUNCH(c, LANE_FFT, KID_FFT_COLS, (spec_fft_radix<32, COLS>), grid, block, 0, fs, L, tw_scale, hi, lo, dst); break;
    }
}

// the power-of-two transform chain of geometry `g` (the spectrum's own size, or the convolution length of a chirp-z transform) over `nf` frames
static int spec_run_pow2(csdr_spec *s, const SpecGeom &g, const float2 *tw_hi, const float2 *tw_lo, const FrameSrc &fs, int nf, float *mag, float2 *raw) {
    csdr_ctx *c = s->ctx;
    if (g.N < 4096) {
        CSDR_LAUNCH(c, LANE_FFT, KID_FFT_ROWS, spec_fft_small, dim3(1, nf), dim3(kFftThreads), (size_t)2 * g.N * sizeof(float2), fs, g.N, s->tw4096.p, mag, raw);
    } else if (g.Ra == 1) {
        CSDR_LAUNCH(c, LANE_FFT, KID_FFT_ROWS, spec_fft_rows4096, dim3(1, nf), dim3(kFftThreads), kRowLdsBytes, fs, g, s->tw4096.p, mag, raw);
    } else if (g.Ra == kC512) {
        CSDR_LAUNCH(c, LANE_FFT, KID_FFT_COLS, spec_cols512, dim3(g.N / kC512 / kC512Cols, nf), dim3(kFftThreads), kC512Lds, fs, g.N, s->tw4096.p, tw_hi, tw_lo, s->tmp.p);
        FrameSrc rows{s->tmp.p, nullptr, s->tmp.p + g.N, g.N, 1 << 30};
        CSDR_LAUNCH(c, LANE_FFT, KID_FFT_ROWS, spec_fft_rows4096, dim3(g.Ra, nf), dim3(kFftThreads), kRowLdsBytes, rows, g, s->tw4096.p, mag, raw);
    } else {
        // radix passes: Ra-point columns of each frame, then (optionally) Rb-point columns inside each of the Ra sub-sequences
        if (g.Ra <= 16) launch_radix<2>(c, g.Ra, fs, g.N, 1u, nf, tw_hi, tw_lo, s->tmp.p);
        else launch_radix<1>(c, g.Ra, fs, g.N, 1u, nf, tw_hi, tw_lo, s->tmp.p);
        if (g.Rb > 1) {
            const int L2 = g.N / g.Ra;
            FrameSrc sub{s->tmp.p, nullptr, s->tmp.p + L2, L2, 1 << 30};
            if (g.Rb <= 16) launch_radix<2>(c, g.Rb, sub, L2, (unsigned)g.Ra, nf * g.Ra, tw_hi, tw_lo, s->tmp.p);
            else launch_radix<1>(c, g.Rb, sub, L2, (unsigned)g.Ra, nf * g.Ra, tw_hi, tw_lo, s->tmp.p);
        }
        FrameSrc rows{s->tmp.p, nullptr, s->tmp.p + g.N, g.N, 1 << 30};
        CSDR_LAUNCH(c, LANE_FFT, KID_FFT_ROWS, spec_fft_rows4096, dim3(g.Ra * g.Rb, nf), dim3(kFftThreads), kRowLdsBytes, rows, g,
                    s->tw4096.p, mag, raw);
    }
    CSDR_HIP_TRY(hipGetLastError());
    return CSDR_OK;
}


static int spec_run_fft(csdr_spec *s, const FrameSrc &fs, int nf, float *mag, float2 *raw) {
    const SpecGeom &g = s->g;
    csdr_ctx *c = s->ctx;
    if (s->fused_now && !raw) {
        // pass 1 of the 512 x 256 chain: Z[f][k1][n2] into `tmp`; pass 2 runs fused with the averaging (spec_post_range).  Persistent
        // workgroups: 16 column tiles x as many frame groups as are resident at once, each walking its frames
        const int ntile = kS3R / kP1Cols;
        const int groups = std::max(1, std::min(nf, c->wg_slots(spec_cols512p, kP1Threads, kP1Lds) / ntile));
        CSDR_LAUNCH(c, LANE_FFT, KID_FFT_COLS, spec_cols512p, dim3(ntile * groups), dim3(kP1Threads), kP1Lds, fs, nf, s->tw_hi.p, s->tw_lo.p, s->tmp.p);
        CSDR_HIP_TRY(hipGetLastError());
        return CSDR_OK;
    }
    if (g.npot && s->blue_big) {
        // chirp-z with a convolution longer than the LDS transforms: x w -> FFT_L -> times Bf, conjugated -> FFT_L -> conjugate, / L, times w
        // (the frames of a call in groups of blue_chunk: the three work arrays hold at most 2^26 complex each)
        const int L = s->blue_L;
        FrameSrc fa{s->blue_a.p, nullptr, s->blue_a.p + L, L, 1 << 30};
        for (int f0 = 0; f0 < nf; f0 += s->blue_chunk) {
            const int cnt = std::min(s->blue_chunk, nf - f0);
            const dim3 ew((L + kFftThreads - 1) / kFftThreads, cnt);
            CSDR_LAUNCH(c, LANE_FFT, KID_FFT_COLS, spec_blue_pre, ew, dim3(kFftThreads), 0, fs, f0, g.N, L, s->blue_w.p, s->blue_a.p);
            if (int rc = spec_run_pow2(s, s->gL, s->twL_hi.p, s->twL_lo.p, fa, cnt, nullptr, s->blue_b.p)) return rc;
            CSDR_LAUNCH(c, LANE_FFT, KID_FFT_COLS, spec_blue_mid, ew, dim3(kFftThreads), 0, s->blue_b.p, s->blue_B.p, L, s->blue_a.p);
            if (int rc = spec_run_pow2(s, s->gL, s->twL_hi.p, s->twL_lo.p, fa, cnt, nullptr, s->blue_b.p)) return rc;
            CSDR_LAUNCH(c, LANE_FFT, KID_FFT_ROWS, spec_blue_post, dim3((g.N + kFftThreads - 1) / kFftThreads, cnt), dim3(kFftThreads), 0, s->blue_b.p, s->blue_w.p, g.N, L,
                        mag ? mag + (size_t)f0 * g.N : nullptr, raw ? raw + (size_t)f0 * g.N : nullptr);
        }
    } else if (g.npot) {
        CSDR_LAUNCH(c, LANE_FFT, KID_FFT_ROWS, spec_fft_bluestein, dim3(1, nf), dim3(kFftThreads), (size_t)2 * s->blue_L * sizeof(float2), fs, g.N, s->blue_L, s->tw4096.p,
                    s->blue_w.p, s->blue_B.p, mag, raw);
    } else return spec_run_pow2(s, g, s->tw_hi.p, s->tw_lo.p, fs, nf, mag, raw);
    CSDR_HIP_TRY(hipGetLastError());
    return CSDR_OK;
}

// averaging .. display for the frames [f0, f0 + cnt) of the current batch; frames >= pk_from (relative to f0) hold peaks
static int spec_post_range(csdr_spec *s, const float *mag, int f0, int cnt, int pk_from) {
    csdr_ctx *c = s->ctx;
    const SpecGeom &g = s->g;
    const size_t F = (size_t)g.F;
    const bool hold = pk_from < cnt, view = s->view_frame;
    const bool bins = hold || view;                                  // per-bin averaged values are kept (maaf)
    if (s->fused_now) {
        if (view) return fail(CSDR_ESTATE, "internal: the fused spectrum pass was chosen for a zoomed view");
        const int npairs = kS3C / 2;
        // (peak hold live in this range -- frames >= pk_from hold: the row pass keeps the held maxima of its bins, round 6; else the plain instance)
        if (hold)
            CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_AVG, spec_rows256_ema<true>, dim3(npairs), dim3(kR2Threads), kR2Lds, s->tmp.p + (size_t)f0 * g.N, cnt, g, (double)s->avg_rate, s->tw4096.p,
                        s->ma.p, s->maa.p, s->pairsum.p + f0 * F, s->first_b.p + f0, s->ext_w.p + (size_t)f0 * npairs, s->peak.p, s->peaksum.p + f0 * F, s->peak_b.p + f0, pk_from);
        else
            CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_AVG, spec_rows256_ema<false>, dim3(npairs), dim3(kR2Threads), kR2Lds, s->tmp.p + (size_t)f0 * g.N, cnt, g, (double)s->avg_rate, s->tw4096.p,
                        s->ma.p, s->maa.p, s->pairsum.p + f0 * F, s->first_b.p + f0, s->ext_w.p + (size_t)f0 * npairs, (double *)nullptr, (float *)nullptr, (float *)nullptr, cnt);
        const SpecScalars *st_in = s->scal.p + s->scal_parity;
        if (!hold && f0 == 0 && cnt <= kTrackSmallFrames) {
            // a short batch (the one-block call: 7 or 8 frames): extrema and trackers in one launch
            CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_TRACK, spec_trackers, dim3(cnt), dim3(kDispThreads), kTrackLds + (size_t)cnt * sizeof(float2), s->ext.p, cnt, st_in,
                        s->scal.p + (s->scal_parity ^ 1), s->fo.p, s->fsc.p, cnt, (const SpecFrameOut *)nullptr, s->ext_w.p, npairs, s->ext.p);
        } else {
            CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_TRACK, spec_extrema, dim3(cnt), dim3(256), (size_t)(256 / 64) * sizeof(float2), s->ext_w.p + (size_t)f0 * npairs, npairs, s->ext.p + f0);
            if (hold) CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_MISC, spec_peak_trackers, dim3(1), dim3(64), 0, s->ext.p + f0, cnt, pk_from, st_in, s->pk.p, s->pfo.p + f0);
            CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_TRACK, spec_trackers, dim3(cnt), dim3(kDispThreads), kTrackLds, s->ext.p + f0, cnt, st_in, s->scal.p + (s->scal_parity ^ 1),
                        s->fo.p + f0, s->fsc.p + f0, hold ? pk_from : cnt, hold ? s->pfo.p + f0 : (const SpecFrameOut *)nullptr, (const float2 *)nullptr, 0, (float2 *)nullptr);
        }
        // (the carry of a contiguous stream rides on the display launch: csdr_spec_process sets carry_src for a batch whose range starts at frame 0)
        const bool carry_here = !hold && f0 == 0 && s->carry_fold_n > 0 && cnt <= kTrackSmallFrames;      // (a long batch keeps its own transfer: the copy inside the launch cost it 16 us)
        if (hold)
            CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_DISPLAY, (spec_display_p256<false, true>), dim3(64, cnt), dim3(kDispThreads), 32 * 33 * sizeof(float),
                        s->pairsum.p + f0 * F, s->first_b.p + f0, s->fsc.p + f0, g, s->scale, s->points.p + f0 * F, (const float2 *)nullptr, (float2 *)nullptr, 0,
                        s->peaksum.p + f0 * F, s->peak_b.p + f0, s->hold_points.p + f0 * F, pk_from);
        else if (carry_here)
            CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_DISPLAY, spec_display_p256<true>, dim3(64, cnt), dim3(kDispThreads), 32 * 33 * sizeof(float),
                        s->pairsum.p + f0 * F, s->first_b.p + f0, s->fsc.p + f0, g, s->scale, s->points.p + f0 * F, s->carry_fold_src, s->carry.p, s->carry_fold_n,
                        (const float *)nullptr, (const float *)nullptr, (float *)nullptr, 0);
        else
            CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_DISPLAY, spec_display_p256<false>, dim3(64, cnt), dim3(kDispThreads), 32 * 33 * sizeof(float),
                        s->pairsum.p + f0 * F, s->first_b.p + f0, s->fsc.p + f0, g, s->scale, s->points.p + f0 * F, (const float2 *)nullptr, (float2 *)nullptr, 0,
                        (const float *)nullptr, (const float *)nullptr, (float *)nullptr, 0);
        if (carry_here) s->carry_fold_n = -1;                        // done
        s->scal_parity ^= 1;
        CSDR_HIP_TRY(hipGetLastError());
        return CSDR_OK;
    }
    // frame groups per workgroup: up to 16 frames each, so a short batch does not pay the set-up of sixteen groups
    // (CSDR_AVG_GROUPS = 4 | 8 | 16 caps the groups: fewer, smaller workgroups let two of them share a CU -- one loads its round while the
    // other scans)
    static const int avg_cap = std::max(1, std::min(kAvgGroups, lab_int("CSDR_AVG_GROUPS", kAvgGroupsDefault)));
    const int avg_groups = std::max(1, std::min(avg_cap, (cnt + kAvgGMax - 1) / kAvgGMax));      // (fewer frames per group -- 8 / 4 / 2 -- measured on C5's 25-frame batches: 0.22 -> 0.25 - 0.27 ms)
    CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_AVG, spec_average, dim3(s->n_avg_tiles), dim3(kAvgLanes * avg_groups), avg_lds_bytes(avg_groups), mag + (size_t)f0 * g.N, cnt, g, (double)s->avg_rate,
                s->ma.p, s->maa.p, s->pairsum.p + f0 * F, s->first_b.p + f0, s->ext_w.p + (size_t)f0 * s->n_avg_tiles,
                bins ? s->maaf.p + f0 * F : (float2 *)nullptr, view ? 0 : (hold ? pk_from : cnt));
    const int ext_threads = s->n_avg_tiles >= 4096 ? kExtMaxThreads : 256;
    CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_TRACK, spec_extrema, dim3(cnt), dim3(ext_threads), (size_t)(ext_threads / 64) * sizeof(float2), s->ext_w.p + (size_t)f0 * s->n_avg_tiles, s->n_avg_tiles, s->ext.p + f0);
    const SpecScalars *st_in = s->scal.p + s->scal_parity;
    if (hold) {
        CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_MISC, spec_peak_track, dim3((g.F + 255) / 256), dim3(256), 0, s->maaf.p + f0 * F, cnt, pk_from, g.F, s->peak.p,
                    s->peaksum.p + f0 * F, s->peak_b.p + f0, view ? s->peakf.p + f0 * F : (float2 *)nullptr);
        CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_MISC, spec_peak_trackers, dim3(1), dim3(64), 0, s->ext.p + f0, cnt, pk_from, st_in, s->pk.p, s->pfo.p + f0);
    }
    // trackers of every frame (closed form, one workgroup per frame), then the display: the transposing path takes one tile per workgroup
    CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_TRACK, spec_trackers, dim3(cnt), dim3(kDispThreads), kTrackLds, s->ext.p + f0, cnt, st_in, s->scal.p + (s->scal_parity ^ 1),
                s->fo.p + f0, s->fsc.p + f0, hold ? pk_from : cnt, hold ? s->pfo.p + f0 : (const SpecFrameOut *)nullptr, (const float2 *)nullptr, 0, (float2 *)nullptr);
    const int npairs = g.Ra == 1 ? 1 : (g.Ra >> 1) * g.Rb;
    const bool transposing = !view && g.Ra > 1 && npairs <= kDispMaxPairs;
    const size_t disp_lds = transposing ? disp_lds_bytes(npairs, g.lgRa - 1 + g.lgRb, hold) : kDispLdsPlain;
    const int disp_gx = transposing ? std::max(1, g.F / disp_tile_points(npairs))
                                    : std::max(1, std::min((g.F / 2 + kDispThreads - 1) / kDispThreads, c->wg_slots(spec_display<false>, kDispThreads, kDispLdsPlain) / std::max(1, cnt)));
#define CSDR_DISPLAY(T_)                                                                                                                   \
    CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_DISPLAY, spec_display<T_>, dim3(disp_gx, cnt), dim3(kDispThreads), disp_lds,                         \
                s->pairsum.p + f0 * F, s->first_b.p + f0, s->fsc.p + f0, g, s->scale, s->points.p + f0 * F, hold ? pk_from : cnt,           \
                hold ? s->peaksum.p + f0 * F : (const float *)nullptr, hold ? s->peak_b.p + f0 : (const float *)nullptr,                    \
                hold ? s->hold_points.p + f0 * F : (float *)nullptr,                                                                        \
                view ? s->vmap.p : (const int2 *)nullptr, view ? s->maaf.p + f0 * F : (const float2 *)nullptr,                              \
                view && hold ? s->peakf.p + f0 * F : (const float2 *)nullptr)
    if (transposing) CSDR_DISPLAY(true); else CSDR_DISPLAY(false);
#undef CSDR_DISPLAY
    s->scal_parity ^= 1;
    CSDR_HIP_TRY(hipGetLastError());
    return CSDR_OK;
}

// The batch holds nf frames made from n_inputs process() inputs (input 0 makes no frame when it only primed fftLastData).
// peakReset counts inputs (:264-273); the reset uses the trackers as they stand before that input's frame.
static int spec_post_frames(csdr_spec *s, const float *mag, int nf, int n_inputs, bool first_input_has_frame) {
    csdr_ctx *c = s->ctx;
    const size_t F = (size_t)s->g.F;
    const int skip = first_input_has_frame ? 0 : 1;                  // frame of input i is i - skip
    // walk the inputs: doPeak(i) = peakHold && peakReset == 0 (before the decrement, :247)
    int reset_input = -1, first_peak_input = n_inputs;
    {
        int pr = s->peak_reset;
        for (int i = 0; i < n_inputs; ++i) {
            if (s->peak_hold && pr == 0 && first_peak_input == n_inputs) first_peak_input = i;
            if (pr != 0 && --pr == 0) reset_input = i;
        }
        s->peak_reset = pr;
    }
    if (s->peak_hold || reset_input >= 0) {
        const size_t nfF = (size_t)s->max_frames * F;
        if (int rc = s->peak.reserve(2 * F)) return rc;
        if (int rc = s->pk.reserve(1)) return rc;
        if (s->peak_hold) {
            if (!s->fused_now) if (int rc = s->maaf.reserve(nfF)) return rc;      // (the fused chain keeps the held maxima in its row pass: no per-bin copy)
            if (int rc = s->peaksum.reserve(nfF)) return rc;
            if (int rc = s->peak_b.reserve(s->max_frames)) return rc;
            if (int rc = s->hold_points.reserve(nfF)) return rc;
            if (int rc = s->pfo.reserve(s->max_frames)) return rc;
        }
    }
    s->hold_valid.assign((size_t)std::max(nf, 0), 0);
    auto frame_of = [&](int input) { return std::min(nf, std::max(0, input - skip)); };
    for (int f = frame_of(first_peak_input); f < nf; ++f) s->hold_valid[f] = 1;
    if (reset_input < 0) return nf > 0 ? spec_post_range(s, mag, 0, nf, frame_of(first_peak_input)) : CSDR_OK;
    // frames of the inputs before the reset, the reset, then the rest (the reset input itself never holds: :247)
    const int fr = frame_of(reset_input);
    if (fr > 0) if (int rc = spec_post_range(s, mag, 0, fr, fr)) return rc;
    CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_MISC, spec_peak_reset, dim3(std::max(1, std::min(256, (int)(2 * F + 255) / 256))), dim3(256), 0,
                s->scal.p + s->scal_parity, s->peak.p, (int)(2 * F), s->pk.p);
    CSDR_HIP_TRY(hipGetLastError());
    if (nf > fr) return spec_post_range(s, mag, fr, nf - fr, frame_of(first_peak_input) - fr);
    return CSDR_OK;
}

// ---- inputs shorter than the transform (:399-421).  `nl` inputs of `len` < N samples each at x (device memory).
// The very first one only primes fftLastData (zero padded, :406-412); every later one is appended to the previous FFT input
// shifted left by its length (:413-419).  On return x / nl describe the inputs that make frames and fs reads them.
static int spec_lines_begin(csdr_spec *s, const float2 *&x, int len, int &nl, FrameSrc &fs) {
    hipStream_t st = s->ctx->lanes[LANE_FFT];
    const int N = s->g.N;
    for (int k = 0; k < 2; ++k) if (int rc = s->last[k].reserve((size_t)N)) return rc;
    if (!s->last_primed && nl > 0) {
        float2 *L = s->last[s->last_cur].p;
        CSDR_HIP_TRY(hipMemsetAsync(L + len, 0, (size_t)(N - len) * sizeof(float2), st));
        if (len) CSDR_HIP_TRY(hipMemcpyAsync(L, x, (size_t)len * sizeof(float2), hipMemcpyDeviceToDevice, st));
        s->last_primed = true;
        x += len; --nl;
    }
    if (nl > s->max_frames) return fail(CSDR_ERANGE, "%d frames exceed max_frames %d", nl, s->max_frames);
    if (nl > 0) {
        // V = last ++ lines; frame j = V[(j + 1) len, (j + 1) len + N)
        const float2 *L = s->last[s->last_cur].p;
        if (2 * len >= N) {          // only frame 0 straddles the two buffers: read in place
            fs.first = L + len; fs.split = N - len; fs.first2 = x;
            fs.rest = x + (2 * len - N); fs.stride = len;
        } else {                     // several frames straddle: make the tail of `last` and the lines contiguous
            const size_t need = (size_t)(N - len) + (size_t)nl * len;
            if (int rc = s->lines.reserve(need)) return rc;
            CSDR_HIP_TRY(hipMemcpyAsync(s->lines.p, L + len, (size_t)(N - len) * sizeof(float2), hipMemcpyDeviceToDevice, st));
            if (len) CSDR_HIP_TRY(hipMemcpyAsync(s->lines.p + (N - len), x, (size_t)nl * len * sizeof(float2), hipMemcpyDeviceToDevice, st));
            fs.first = s->lines.p; fs.rest = s->lines.p + len; fs.stride = len;
        }
    }
    return CSDR_OK;
}
// fftLastData = the last FFT input (:417) = V[nl len, nl len + N), written to the other copy (lane FFT: behind the kernels
// that read the current one)
static int spec_lines_end(csdr_spec *s, const float2 *x, int len, int nl) {
    if (nl <= 0) return CSDR_OK;
    hipStream_t st = s->ctx->lanes[LANE_FFT];
    const int N = s->g.N;
    const float2 *L = s->last[s->last_cur].p;
    float2 *Ln = s->last[s->last_cur ^ 1].p;
    const int64_t from_x = (int64_t)nl * len;                    // samples of the inputs inside the new fftLastData (if < N)
    if (from_x >= N) {
        CSDR_HIP_TRY(hipMemcpyAsync(Ln, x + (from_x - N), (size_t)N * sizeof(float2), hipMemcpyDeviceToDevice, st));
    } else {
        CSDR_HIP_TRY(hipMemcpyAsync(Ln, L + from_x, (size_t)(N - from_x) * sizeof(float2), hipMemcpyDeviceToDevice, st));
        if (from_x) CSDR_HIP_TRY(hipMemcpyAsync(Ln + (N - from_x), x, (size_t)from_x * sizeof(float2), hipMemcpyDeviceToDevice, st));
    }
    s->last_cur ^= 1;
    return CSDR_OK;
}

// FFT of nf frames on lane FFT, then `post(mag)` on lane AVG
template <typename PostFn>
static int spec_fft_then(csdr_spec *s, const FrameSrc &fs, int nf, PostFn post) {
    csdr_ctx *c = s->ctx;
    // lane FFT fills magnitude copy `mp`; its previous reader was the averaging kernel two batches ago
    const int mp = (c->same(LANE_FFT, LANE_AVG) || s->fused_now) ? 0 : (int)(s->seq & 1);      // (fused: the one intermediate buffer is the hand-off)
    float *mag = s->mag.p + (size_t)mp * s->max_frames * s->g.N;
    if (s->avg_pending[mp]) if (int rc = c->wait(s->ev_avg_done[mp], LANE_AVG, LANE_FFT)) return rc;
    // `tmp` is the fused chain's hand-off: its row pass reads it on lane AVG.  Whatever runs next on lane FFT rewrites it (a radix pass of the
    // general chain as well, after peak hold was switched on), so that reader is waited for whichever magnitude copy this batch takes
    if (s->tmp_reader >= 0 && s->tmp_reader != mp) if (int rc = c->wait(s->ev_avg_done[s->tmp_reader], LANE_AVG, LANE_FFT)) return rc;
    s->tmp_reader = -1;
    if (int rc = spec_run_fft(s, fs, nf, mag, nullptr)) return rc;
    if (int rc = c->signal(s->ev_fft_done[mp], LANE_FFT, LANE_AVG)) return rc;
    // lane AVG: averaging, extrema, trackers + display points
    if (int rc = c->wait(s->ev_fft_done[mp], LANE_FFT, LANE_AVG)) return rc;
    if (int rc = post(mag)) return rc;
    if (int rc = c->signal(s->ev_avg_done[mp], LANE_AVG, LANE_FFT)) return rc;
    s->avg_pending[mp] = true;
    if (s->fused_now) s->tmp_reader = mp;
    s->seq++;
    return CSDR_OK;
}

// ---- zoomed view: one process() input (:283-386).  The frequency shift and the msresamp run on a private front-end-only
// demodulator slot (the same NCO + msresamp_crcf kernels the demodulators use); the frame rule, FFT, averaging and display
// follow with the view's bin walk.
static int spec_process_view(csdr_spec *s, const float *iq, int iq_is_dev, int block_len) {
    csdr_ctx *c = s->ctx;
    const int N = s->g.N, F = s->g.F;
    const int64_t rate = s->input_rate;
    s->fused_now = false;                                                        // the zoomed view keeps per-bin values: the general kernels
    s->nf_last = 0;
    s->hold_valid.clear();
    // head of process() (:247, :264-273): doPeak is taken before the countdown moves; a reset uses the trackers as they stand
    const bool do_peak = s->peak_hold && s->peak_reset == 0;
    if (int rc = s->peak.reserve((size_t)2 * F)) return rc;
    if (int rc = s->pk.reserve(1)) return rc;
    if (s->peak_reset != 0 && --s->peak_reset == 0) {
        CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_MISC, spec_peak_reset, dim3(std::max(1, std::min(256, (2 * F + 255) / 256))), dim3(256), 0,
                    s->scal.p + s->scal_parity, s->peak.p, 2 * F, s->pk.p);
        CSDR_HIP_TRY(hipGetLastError());
    }
    if (!rate) { s->last_view = true; return CSDR_OK; }                        // :286-289
    // the previous view frame is complete before its buffers are reused (display-rate path: a host wait is affordable)
    CSDR_HIP_TRY(hipStreamSynchronize(c->lanes[LANE_AVG]));
    CSDR_HIP_TRY(hipStreamSynchronize(c->lanes[LANE_FFT]));
    long resampleBw = (long)rate;
    while (resampleBw / 2 >= (long)s->bandwidth && resampleBw / 2 > 0) resampleBw /= 2;      // SPECTRUM_VZM, :291-293
    s->resample_bw = resampleBw;
    const double ratio = (double)resampleBw / (double)rate;                     // :295
    size_t desired = (size_t)((double)N / ratio);                               // :297
    s->desired_input_size = (int)desired;                                       // :299
    if ((size_t)block_len < desired) desired = (size_t)block_len;               // :301-304
    bool new_resampler = false;
    long bw_diff = 0;
    const bool mix = s->center_freq != s->input_freq;                           // :306
    if (int rc = s->ma2.reserve((size_t)2 * F)) return rc;
    if (int rc = s->maa2.reserve((size_t)2 * F)) return rc;
    auto remap = [&](int mode, int n) -> int {
        CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_MISC, spec_avg_remap, dim3(std::max(1, std::min(512, (N + 255) / 256))), dim3(256), 0,
                    s->ma.p, s->maa.p, s->ma2.p, s->maa2.p, N, mode, n, s->g);
        CSDR_HIP_TRY(hipGetLastError());
        std::swap(s->ma.p, s->ma2.p); std::swap(s->maa.p, s->maa2.p);
        return CSDR_OK;
    };
    if (mix) {
        if ((long)(s->center_freq - s->input_freq) != s->shift_frequency || s->last_input_bandwidth != (long)rate) {     // :307
            if (std::llabs(s->input_freq - s->center_freq) < rate / 2) {        // :308 (the application rate is the input rate)
                const long last_shift = s->shift_frequency;
                s->shift_frequency = (long)(s->center_freq - s->input_freq);    // the NCO frequency follows inside the slot (:311)
                const long freq_diff = s->shift_frequency - last_shift;
                if (s->last_bandwidth != 0) {                                   // the averagers follow the retune (:316-331)
                    const double bin_per_hz = double(s->last_bandwidth) / double(N);
                    const unsigned num_shift = (unsigned)std::floor(double(std::labs(freq_diff)) / bin_per_hz);
                    if (num_shift < (unsigned)N / 2 && num_shift) if (int rc = remap(freq_diff > 0 ? 0 : 1, (int)num_shift)) return rc;
                }
            }
            s->peak_reset = 30;                                                 // PEAK_RESET_COUNT :335
        }
    }
    // (re)build the resampler (:354-368)
    if (!s->vpost) { if (int rc = csdr_post_create(c, &s->vpost)) return rc; }
    if (!s->vbank) { if (int rc = csdr_bank_create(c, 1, 1, &s->vbank)) return rc; }
    if (!s->have_resampler || resampleBw != s->last_bandwidth || s->last_input_bandwidth != (long)rate) {
        uint32_t theta = 0;
        if (s->have_resampler) theta = s->vbank->slots[0].theta;                // freqShifter lives on across resamplers
        const int cap = std::max(std::max(s->vpost_cap, (int)((double)N / ratio) + 16), block_len);
        if (int rc = csdr_post_configure(s->vpost, rate, 1, CSDR_POST_SINGLE, cap, 1)) return rc;
        s->vpost->raw = true;
        s->vpost_cap = cap;
        csdr_demod_params prm = {CSDR_MODEM_FRONTEND_ONLY, (int32_t)resampleBw, (int32_t)resampleBw, 0, s->input_freq};
        if (int rc = bank_configure_slot(s->vbank, 0, &prm, s->vpost)) return rc;   // msresamp_crcf_create(resamplerRatio, 60) :361
        s->vbank->slots[0].theta = theta;
        bw_diff = resampleBw - s->last_bandwidth;
        s->last_bandwidth = resampleBw; s->last_input_bandwidth = (long)rate;
        s->have_resampler = true;
        new_resampler = true;
        s->peak_reset = 30;                                                     // :367
    } else if (block_len > s->vpost_cap) {
        return fail(CSDR_ERANGE, "view input of %d samples exceeds the %d the resampler was built for", block_len, s->vpost_cap);
    }
    // shift (:341-352) + resample (:379) of the first `desired` samples
    if (int rc = csdr_bank_set_frequency(s->vbank, 0, mix ? s->input_freq + s->shift_frequency : s->input_freq)) return rc;
    if (int rc = csdr_post_execute(s->vpost, iq, iq_is_dev, 1, (int)desired, s->input_freq)) return rc;
    if (int rc = csdr_bank_execute(s->vbank, s->vpost)) return rc;
    const SlotHost &sl = s->vbank->slots[0];
    const int nw = sl.last_J;                                                   // num_written
    const float2 *xr = sl.cfg.iq + (size_t)sl.last_parity * ((size_t)kIqHist + sl.cfg.cap_iq) + kIqHist;
    // the spectrum lanes read what the front-end lane wrote
    CSDR_HIP_TRY(hipStreamSynchronize(c->lanes[LANE_FE]));
    // frame rule (:399-421)
    if (int rc = c->lane_begin(LANE_FFT)) return rc;
    FrameSrc fs{nullptr, nullptr, nullptr, 0, 1 << 30};
    int nf = 0;
    const float2 *lx = xr;
    int nl = 1;
    if (nw >= N) { fs.first = xr; nf = 1; }
    else { if (int rc = spec_lines_begin(s, lx, nw, nl, fs)) return rc; nf = nl; }
    s->nf_last = nf;
    if (nf > 0) {
        // bins per display point for visualRatio = bandwidth / resampleBw (:532-560), walked with the reference's accumulator
        if (s->vmap_bw != s->bandwidth || s->vmap_rbw != resampleBw) {
            if (int rc = s->vmap.reserve((size_t)F)) return rc;
            std::vector<int2> vm((size_t)F);
            const double visualRatio = double(s->bandwidth) / double(resampleBw);
            const double visualStart = (double(N) / 2.0) - (double(N) * (visualRatio / 2.0));
            double visualAccum = 0, i = 0;
            for (int x = 0; x < F; ++x) {
                visualAccum += visualRatio * 2.0;
                int first = 0, cnt = 0;
                while (visualAccum >= 1.0) {
                    const unsigned idx = (unsigned)std::round(visualStart + i);
                    if (!cnt) first = (int)idx;
                    ++cnt; visualAccum -= 1.0; i += 1.0;
                }
                vm[x] = make_int2(first, cnt);
            }
            CSDR_HIP_TRY(hipMemcpy(s->vmap.p, vm.data(), vm.size() * sizeof(int2), hipMemcpyHostToDevice));
            s->vmap_bw = s->bandwidth; s->vmap_rbw = resampleBw;
        }
        const size_t nfF = (size_t)s->max_frames * F;
        if (int rc = s->maaf.reserve(nfF)) return rc;
        if (do_peak) {
            if (int rc = s->peaksum.reserve(nfF)) return rc;
            if (int rc = s->peak_b.reserve(s->max_frames)) return rc;
            if (int rc = s->hold_points.reserve(nfF)) return rc;
            if (int rc = s->pfo.reserve(s->max_frames)) return rc;
            if (int rc = s->peakf.reserve(nfF)) return rc;
        }
        const bool rescale = new_resampler && s->last_view;                     // :454
        int rc = spec_fft_then(s, fs, 1, [&](float *mag) -> int {
            if (rescale) if (int r2 = remap(bw_diff < 0 ? 2 : 3, 0)) return r2;  // :455-491
            s->view_frame = true;
            const int r3 = spec_post_range(s, mag, 0, 1, do_peak ? 0 : 1);
            s->view_frame = false;
            return r3;
        });
        if (rc) return rc;
        s->hold_valid.assign(1, do_peak ? 1 : 0);
        hipStream_t st = c->lanes[LANE_FFT];
        if (nw >= N) {                                                           // memcpy(fftLastData, fftInput) :404
            for (int k = 0; k < 2; ++k) if (int r4 = s->last[k].reserve((size_t)N)) return r4;
            CSDR_HIP_TRY(hipMemcpyAsync(s->last[s->last_cur ^ 1].p, xr, (size_t)N * sizeof(float2), hipMemcpyDeviceToDevice, st));
            s->last_cur ^= 1;
        } else if (int r5 = spec_lines_end(s, lx, nw, nl)) return r5;
    }
    s->last_view = true;                                                         // :631
    return CSDR_OK;
}

extern "C" int csdr_spec_process(csdr_spec *s, const float *iq, int iq_is_dev, int n_blocks, int block_len, int mode) {
    RangeScope range__("csdr_spec_process");
    DeviceScope dev__(s ? s->ctx : nullptr);
    if (!s || !s->ready) return fail(CSDR_ESTATE, "spec not set up");
    if (!iq || n_blocks <= 0 || block_len <= 0) return fail(CSDR_EINVAL, "bad block arguments");
    if (s->is_view) {
        if (n_blocks != 1) return fail(CSDR_EINVAL, "the zoomed view takes one process() input per call");
        return spec_process_view(s, iq, iq_is_dev, block_len);
    }
    csdr_ctx *c = s->ctx;
    hipStream_t st = c->lanes[LANE_FFT];
    const SpecGeom &g = s->g;
    const int N = g.N;
    // full-span view: the 512 x 256 chain with the averaging fused into its row pass (since round 6 with peak hold too: CSDR_SPEC_FUSED_HOLD=0 is the
    // earlier rule -- a batch with peak hold set or pending on the three-kernel chain)
    s->fused_now = s->fused_ok && ((!s->peak_hold && s->peak_reset == 0) || lab_int("CSDR_SPEC_FUSED_HOLD", 1) != 0);
    const int64_t n = (int64_t)n_blocks * block_len;
    const float2 *x = (const float2 *)iq;
    if (int rc = c->lane_begin(LANE_FFT)) return rc;
    if (!iq_is_dev) {
        if (int rc = s->stage_in.reserve((size_t)n)) return rc;
        CSDR_HIP_TRY(hipMemcpyAsync(s->stage_in.p, iq, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, st));
        x = s->stage_in.p;
    } else if ((uintptr_t)iq & 7) return fail(CSDR_EINVAL, "device IQ pointer must be 8-byte aligned");
    FrameSrc fs{nullptr, nullptr, nullptr, 0, 1 << 30};
    int nf = 0;
    const float2 *lines_x = nullptr;
    int lines_n = 0;
    if (mode == CSDR_SPEC_FIRST_FRAME) {
        if (block_len < N) return fail(CSDR_EINVAL, "block_len %d < internal FFT size %d: use CSDR_SPEC_LINES for short inputs", block_len, N);
        nf = n_blocks; fs.first = x; fs.rest = x + block_len; fs.stride = block_len;
    } else if (mode == CSDR_SPEC_CONTIGUOUS) {
        const int64_t total = s->carry_len + n;
        nf = (int)(total / N);
        if (nf > s->max_frames) return fail(CSDR_ERANGE, "%d frames exceed max_frames %d", nf, s->max_frames);
        if (nf > 0) {
            // frame 0 = carry ++ head of the new data (read in place, two pieces); the rest are contiguous in x
            if (s->carry_len > 0) { fs.first = s->carry.p; fs.first2 = x; fs.split = s->carry_len; }
            else fs.first = x;
            fs.rest = x + (N - s->carry_len); fs.stride = N;
        }
    } else if (mode == CSDR_SPEC_LINES) {
        // every block is one input of fewer than 2*fftSize samples (e.g. FFTDataDistributor lines of fftSize samples)
        if (block_len >= N) return fail(CSDR_EINVAL, "CSDR_SPEC_LINES takes blocks shorter than the internal FFT size %d", N);
        lines_x = x; lines_n = n_blocks;
        if (int rc = spec_lines_begin(s, lines_x, block_len, lines_n, fs)) return rc;
        nf = lines_n;
    } else return fail(CSDR_EINVAL, "mode");
    if (nf > s->max_frames) return fail(CSDR_ERANGE, "%d frames exceed max_frames %d", nf, s->max_frames);
    s->nf_last = nf;
    // process() inputs behind these frames (peakReset counts inputs, :264): a block or a line each; contiguous mode has no
    // reference input boundaries, every frame counts as one
    const bool first_input_has_frame = !(mode == CSDR_SPEC_LINES && nf < n_blocks);
    const int n_inputs = mode == CSDR_SPEC_CONTIGUOUS ? nf : n_blocks;
    if (nf == 0 && n_inputs > 0) { if (int rc = spec_post_frames(s, nullptr, 0, n_inputs, first_input_has_frame)) return rc; }
    s->carry_fold_n = 0;
    if (mode == CSDR_SPEC_CONTIGUOUS && nf > 0 && s->fused_now) {      // the fused chain's display launch takes the carry copy along (spec_post_range)
        const int rem = (int)(s->carry_len + n - (int64_t)nf * N);
        if (rem > 0) { s->carry_fold_src = x + (n - rem); s->carry_fold_n = rem; }
    }
    if (nf > 0)
        if (int rc = spec_fft_then(s, fs, nf, [&](float *mag) { return spec_post_frames(s, mag, nf, n_inputs, first_input_has_frame); })) return rc;
    if (mode == CSDR_SPEC_LINES) if (int rc = spec_lines_end(s, lines_x, block_len, lines_n)) return rc;
    if (mode == CSDR_SPEC_CONTIGUOUS) {
        // new carry = samples after the last whole frame (lane FFT: ordered behind the kernels that read the old carry)
        const int64_t total = s->carry_len + n;
        const int rem = (int)(total - (int64_t)nf * N);
        if (nf == 0) {
            CSDR_HIP_TRY(hipMemcpyAsync(s->carry.p + s->carry_len, x, (size_t)n * sizeof(float2), hipMemcpyDeviceToDevice, st));
        } else if (rem > 0 && s->carry_fold_n != -1) {               // (-1: the display launch of this call has taken it)
            CSDR_HIP_TRY(hipMemcpyAsync(s->carry.p, x + (n - rem), (size_t)rem * sizeof(float2), hipMemcpyDeviceToDevice, st));
        }
        s->carry_fold_n = 0;
        s->carry_len = rem;
    }
    s->last_view = false;                                                        // :631
    return CSDR_OK;
}

// DC-spike removal on the finished points (:578-623): the bins within 2 kHz of the input centre are overwritten by their
// mirror images just outside that span.  A few values on the host copy; integer arithmetic as in the reference.
static void spec_hide_dc(const csdr_spec *s, float *pts) {
    const long long centerFreq = s->center_freq, inFreq = s->input_freq;
    const long bandwidth = s->bandwidth;
    const long long fftSize = s->g.F;
    const long long freqMin = centerFreq - (bandwidth / 2), freqMax = centerFreq + (bandwidth / 2);
    const long long zeroPt = inFreq - freqMin;
    if (!(freqMin < inFreq && freqMax > inFreq)) return;
    const int freqRange = (int)(freqMax - freqMin);
    const int freqStep = freqRange / (int)fftSize;
    if (freqStep == 0) return;                                       // (the reference would divide by zero)
    int fftStart = (int)(zeroPt / freqStep) - (2000 / freqStep);
    int fftEnd = (int)(zeroPt / freqStep) + (2000 / freqStep);
    if (fftEnd - fftStart < 2) { fftEnd++; fftStart--; }
    const int numSteps = fftEnd - fftStart;
    const int halfWay = fftStart + (numSteps / 2);
    if ((fftEnd + numSteps / 2 + 1 < fftSize) && (fftStart - numSteps / 2 - 1 >= 0) && (fftEnd > fftStart)) {
        int n = 1;
        for (int i = fftStart; i < halfWay; i++) { pts[i * 2 + 1] = pts[(fftStart - n) * 2 + 1]; n++; }
        n = 1;
        for (int i = halfWay; i < fftEnd; i++) { pts[i * 2 + 1] = pts[(fftEnd + n) * 2 + 1]; n++; }
    }
}

// The device keeps only the y of every display point (the x of point i is i / F in every frame, SpectrumVisualProcessor.cpp:562: half of
// the display kernel's stores and of the fetch's transfer were that constant).  pts[F .. 2F) holds the F values just fetched: interleave
// in place, front to back (the value of point i is read before slots 2i, 2i + 1 <= F + i are written).
static void spec_expand_points(float *pts, int F) {
    // the reference's own quotient, (float)x / (float)xMax (:563): a product with 1 / F is the same value only when F is a power of two, and
    // setFFTSize takes any size
    for (int i = 0; i < F; ++i) { const float y = pts[F + i]; pts[2 * i] = (float)i / (float)F; pts[2 * i + 1] = y; }
}

extern "C" int csdr_spec_fetch_hold(csdr_spec *s, int frame, float *hold_host, int cap_floats, int *n_floats) {
    DeviceScope dev__(s ? s->ctx : nullptr);
    if (!s || !s->ready || !hold_host || !n_floats) return fail(CSDR_EINVAL, "bad argument");
    if (frame < 0 || frame >= s->nf_last) return fail(CSDR_EINVAL, "frame %d of %d", frame, s->nf_last);
    const int F = s->g.F;
    *n_floats = 0;
    if ((size_t)frame >= s->hold_valid.size() || !s->hold_valid[frame]) return CSDR_OK;     // spectrum_hold_points.resize(0) (:432)
    if (cap_floats < 2 * F) return fail(CSDR_ERANGE, "need %d floats", 2 * F);
    hipStream_t st = s->ctx->lanes[LANE_AVG];
    CSDR_HIP_TRY(hipMemcpyAsync(hold_host + F, s->hold_points.p + (size_t)frame * F, (size_t)F * sizeof(float), hipMemcpyDeviceToHost, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    spec_expand_points(hold_host, F);
    if (s->hide_dc) spec_hide_dc(s, hold_host);
    *n_floats = 2 * F;
    return CSDR_OK;
}

extern "C" int csdr_spec_fetch(csdr_spec *s, int frame, float *points_host, int cap_floats, double *fft_ceiling, double *fft_floor) {
    DeviceScope dev__(s ? s->ctx : nullptr);
    if (!s || !s->ready || !points_host) return fail(CSDR_EINVAL, "bad argument");
    if (frame < 0 || frame >= s->nf_last) return fail(CSDR_EINVAL, "frame %d of %d", frame, s->nf_last);
    const int F = s->g.F;
    if (cap_floats < 2 * F) return fail(CSDR_ERANGE, "need %d floats", 2 * F);
    hipStream_t st = s->ctx->lanes[LANE_AVG];
    SpecFrameOut fo;
    CSDR_HIP_TRY(hipMemcpyAsync(points_host + F, s->points.p + (size_t)frame * F, (size_t)F * sizeof(float), hipMemcpyDeviceToHost, st));
    CSDR_HIP_TRY(hipMemcpyAsync(&fo, s->fo.p + frame, sizeof fo, hipMemcpyDeviceToHost, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    spec_expand_points(points_host, F);
    if (s->hide_dc) spec_hide_dc(s, points_host);
    if (fft_ceiling) *fft_ceiling = fo.point_ceil / (double)s->scale;     // :626
    if (fft_floor) *fft_floor = fo.point_floor;                            // :627
    return CSDR_OK;
}

extern "C" int csdr_spec_fft_only(csdr_spec *s, const float *iq_host, float *out_host) {
    DeviceScope dev__(s ? s->ctx : nullptr);
    if (!s || !s->ready || !iq_host || !out_host) return fail(CSDR_EINVAL, "bad argument");
    hipStream_t st = s->ctx->lanes[LANE_FFT];
    const int N = s->g.N;
    if (s->tmp_reader >= 0) { CSDR_HIP_TRY(hipStreamSynchronize(s->ctx->lanes[LANE_AVG])); s->tmp_reader = -1; }      // (a fused batch's row pass may still read the intermediate rows)
    if (int rc = s->stage_in.reserve((size_t)N)) return rc;
    if (int rc = s->raw.reserve((size_t)N)) return rc;
    CSDR_HIP_TRY(hipMemcpyAsync(s->stage_in.p, iq_host, (size_t)N * sizeof(float2), hipMemcpyHostToDevice, st));
    FrameSrc fs{s->stage_in.p, nullptr, s->stage_in.p, 0, 1 << 30};
    if (int rc = spec_run_fft(s, fs, 1, nullptr, s->raw.p)) return rc;
    CSDR_HIP_TRY(hipMemcpyAsync(out_host, s->raw.p, (size_t)N * sizeof(float2), hipMemcpyDeviceToHost, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    return CSDR_OK;
}

