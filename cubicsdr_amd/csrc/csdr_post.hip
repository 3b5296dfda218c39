// csdr_post.hip -- implementation of include/csdr_hip.h (gfx950): csdr_post (SDRPostThread: DC blocker, polyphase channelizers, routing, time-slab export / import).  Host-side bookkeeping mirrors the reference's control flow
// (file:line cited per function); all sample arithmetic is in the kernels_*.hpp kernels.
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <map>
#include <memory>

#define CSDR_TU_POST 1          // this unit is the home of its kernels (common.hpp)
#include "csdr_objects.hpp"

using namespace csdr;

// =================================================================================================== SDRPostThread

static void post_update_channels(csdr_post *p) {   // SDRPostThread::updateChannels, SDRPostThread.cpp:116-124
    const int M = p->M;
    p->centers.assign(M + 1, 0);
    if (M == 1) { p->centers[0] = p->frequency; p->centers[1] = p->frequency + p->sample_rate / 2; return; }
    for (int i = 0; i < M / 2; i++) {
        int ofs = (int)(p->chan_bw * i);
        p->centers[i] = p->frequency + ofs;
        p->centers[i + M / 2] = p->frequency - (p->sample_rate / 2) + ofs;
    }
    p->centers[M] = p->frequency + (p->sample_rate / 2);
}

extern "C" int csdr_post_create(csdr_ctx *ctx, csdr_post **out) {
    DeviceScope dev__(ctx);
    if (!ctx || !out) return fail(CSDR_EINVAL, "null argument");
    std::unique_ptr<csdr_post> p(new csdr_post());
    p->ctx = ctx;
    for (int k = 0; k < csdr_post::kPostBufs; ++k) {
        CSDR_HIP_TRY(hipEventCreateWithFlags(&p->ev_ready[k], hipEventDisableTiming));
        for (int c = 0; c < csdr_post::kMaxConsumers; ++c) CSDR_HIP_TRY(hipEventCreateWithFlags(&p->ev_consumed[k][c], hipEventDisableTiming));
    }
    *out = p.release();
    return CSDR_OK;
}
extern "C" void csdr_post_destroy(csdr_post *p) {
    DeviceScope dev__(p ? p->ctx : nullptr);
    if (!p) return;
    (void)p->ctx->sync_all();
    for (int k = 0; k < csdr_post::kPostBufs; ++k) {
        if (p->ev_ready[k]) (void)hipEventDestroy(p->ev_ready[k]);
        for (int c = 0; c < csdr_post::kMaxConsumers; ++c) if (p->ev_consumed[k][c]) (void)hipEventDestroy(p->ev_consumed[k][c]);
    }
    p->out.release(); p->hist0.release(); p->hist1.release(); p->stage_in.release();
    p->twA.release(); p->twB.release(); p->twM.release(); p->post2.release(); p->perm.release();
    p->taps.release(); p->active.release(); p->dc_state.release(); p->tile_end.release();
    for (auto &kv : p->rowlists) (void)hipFree(kv.second);
    p->rowlists.clear();
    delete p;
}

#ifndef CSDR_P2_MX_MIN_A
#define CSDR_P2_MX_MIN_A 11           // critically sampled bank: the smallest A whose transforms go to the matrix pipe (M = 22 ... 62: 1.03 - 1.25 x the vector form, which keeps M = 6 / 10 / 14: profiles/r06_mx_channelizer.txt)
#endif
#ifndef CSDR_P2_OS2_MIN_A
#define CSDR_P2_OS2_MIN_A 19          // firpfbch2 with M / 2 odd: the smallest A = M / 2 that takes chan_analyze_p2 (M >= 38: 1.2 - 2.5 x the two-factor kernel, which wins below: profiles/r06_chan2_p2.txt)
#endif
#ifndef CSDR_CHAN_MX_DEFAULT
#define CSDR_CHAN_MX_DEFAULT 1          // (A/B builds: -DCSDR_CHAN_MX_DEFAULT=0 keeps the vector form of chan_analyze_p2's transform phase for every A)
#endif
// geometry of the channelizer kernel for M channels (see kernels_post.hpp)
static int chan_geometry(int M, int hop, ChanGeom &g) {
    memset(&g, 0, sizeof g);
    g.M = M; g.hop = hop;
    int B = 1;
    for (int d = 1; (int64_t)d * d <= M; ++d) if (M % d == 0) B = d;     // largest divisor <= sqrt(M)
    g.B = B; g.A = M / B;
    // DFT outputs per pass: the k range is cut into the fewest passes of <= 8 accumulators, then evened out (A = 5 -> one
    // pass of 5, not two of 4); rows of the twiddle tables are padded to whole passes
    auto passes = [](int n, int &K, int &nk, int &pitch) { nk = (n + 7) / 8; K = std::max(4, (n + nk - 1) / nk); pitch = nk * K; };
    passes(g.A, g.KA, g.nkA, g.PA);
    passes(g.B, g.KB, g.nkB, g.PB);
    g.oddA = (g.A >= 3 && (g.A & 1)) ? 1 : 0;
    if (g.oddA) {            // conjugate-pair form of phase 1: (A - 1) / 2 output pairs, up to four per pass
        const int H = (g.A - 1) / 2;
        g.nkA = (H + 3) / 4; g.KA = (H + g.nkA - 1) / g.nkA; g.PA = g.nkA * g.KA;
    }
    g.magicM = (unsigned)((1ull << 32) / (unsigned)M) + 1u;
    // M = 2 A with A odd <= 63: chan_analyze_p2.  Critically sampled where A is prime (a composite A factors over the FFT kernel's radices, which is
    // faster); firpfbch2 (hop == M / 2: every frame but one in two starts at an odd sample offset) for every odd A, in the kernel's matrix-pipe form with
    // the two lattices of frames dealt to its waves.
    const bool twice_odd = (M & 3) == 2 && M / 2 >= 3 && M / 2 <= kP2MaxA;
    const bool p2_os2 = twice_odd && M / 2 >= CSDR_P2_OS2_MIN_A && hop * 2 == M && CSDR_CHAN_MX_DEFAULT && lab_int("CSDR_CHAN_P2_OS2", 1);
    if (p2_os2) { g.B = 2; g.A = M / 2; g.oddA = 1; }
    if ((p2_os2 || (hop == M && g.B == 2 && g.oddA && g.A <= kP2MaxA)) && !lab_int("CSDR_CHAN_GENERIC", 0)) {
        // whole transform of a frame inside one lane (kernels_post.hpp, chan_analyze_p2): (A - 1) / 2 output pairs + the k = 0
        // pseudo pair, split evenly over (up to) four passes of at most eight slots
        const int slots = (g.A - 1) / 2 + 1;
        g.p2 = 1;
        g.KA = std::min(8, (slots + kP2Waves - 1) / kP2Waves);
        g.nkA = (slots + g.KA - 1) / g.KA;
        g.PA = g.nkA * g.KA;
        g.TF = kP2Frames; g.lgTF = 6; g.S = M; g.taps_lds = 0; g.stage_in = 1; g.threads = kP2Threads;
        // the A-point transforms on the fp32 matrix pipe (A >= 11; below 33 one row tile of sixteen outputs: half of the waves sit the phase out)
        g.mx = (hop != M || (g.A >= CSDR_P2_MX_MIN_A && lab_int("CSDR_CHAN_MX", CSDR_CHAN_MX_DEFAULT) != 0)) ? 1 : 0;
        return CSDR_OK;
    }
    g.taps_lds = (M <= 512) ? 1 : 0;
    g.stage_in = (M <= 256) ? 1 : 0;
    g.fpw = 0;   // set per launch
    // frames per workgroup: the largest power of two <= 64 whose two row arrays fit the LDS budget
    const size_t budget = (M <= 512) ? 64 * 1024 : 72 * 1024;
    for (int tf = 64; tf >= 1; tf >>= 1) {
        g.TF = tf;
        g.lgTF = 0; while ((1 << g.lgTF) < tf) ++g.lgTF;
        const int q = 32 / std::min(tf, 32);              // row stride = q * odd: lanes along t hit distinct banks
        int S = (M + q - 1) / q; if (!(S & 1)) ++S; S *= q;
        g.S = S;
        if (chan_lds_bytes(g) <= budget) break;
        if (tf == 1) return fail(CSDR_EUNSUPPORTED, "numChannels %d does not fit the channelizer's LDS tile", M);
    }
    if ((int64_t)g.TF * M * M >= (1ll << 31)) return fail(CSDR_EUNSUPPORTED, "numChannels %d too large", M);
    // workgroup size: four waves, one per SIMD (five waves balance M = 20 better on paper -- 10 FIR wave-iterations, 4 + 5 DFT
    // wave-items -- but measured 20 % slower on MI355X: the fifth wave doubles up on one SIMD)
    g.threads = 256;
    return CSDR_OK;
}

typedef void (*chan_kernel_t)(const float2 *, const float2 *, float2 *, const float *, const float2 *, const float2 *, const float2 *,
                              const int *, ChanGeom, int64_t, float2 *, int64_t, d2 *, double, const float2 *);
typedef void (*chan_p2_kernel_t)(const float2 *, const float2 *, float2 *, const float *, const float2 *, const float2 *, const int *, ChanGeom,
                                 int64_t, float2 *, int64_t, d2 *, double, const float2 *);
static chan_p2_kernel_t chan_p2_kernel(const ChanGeom &g) {
#define CSDR_P2_CASE(K_) case K_: return chan_analyze_p2<K_>
    // (A <= 63: at most 32 slots over eight waves = at most four per pass; wider passes were instances nothing ever launched -- one of them spilled)
    if (g.mx) return g.hop != g.M ? chan_analyze_p2<4, true, true> : chan_analyze_p2<4, true>;
#if CSDR_CHAN_MX_DEFAULT == 0 || defined(CSDR_LAB)      // (A/B builds: the vector form for every A)
    switch (g.KA) {
        CSDR_P2_CASE(1); CSDR_P2_CASE(2); CSDR_P2_CASE(3);
        default: return chan_analyze_p2<4>;
    }
#else
    return chan_analyze_p2<1>;                          // A <= 9: at most five slots over eight waves
#endif
#undef CSDR_P2_CASE
}
typedef void (*chanfft_kernel_t)(const float2 *, const float2 *, float2 *, const float *, const float2 *, const int *, const int *, ChanFftGeom, int64_t,
                                 float2 *, int64_t, d2 *, double, const float2 *);
static chanfft_kernel_t chanfft_kernel(const ChanFftGeom &g) {
    if (g.os2) return g.wide_odd ? chan_analyze_fft<true, true> : chan_analyze_fft<false, true>;
    if (g.wide_odd) return chan_analyze_fft<true, false>;
    if (g.bp) return chan_analyze_fft<false, false, 4>;            // a prime factor >= 157: the instance with the chirp-z pass
    if (g.dp) return chan_analyze_fft<false, false, 5>;            // a prime factor 29 .. 151: the instance with the direct prime pass
    switch (lab_int("CSDR_CHANFFT_PLAN", 1) ? cf_plan_of(g) : 0) {      // the BASELINE channel counts have an instance of their own (only their radices: fewer registers)
        case 1: return chan_analyze_fft<false, false, 1>;
        case 2: return chan_analyze_fft<false, false, 2>;
        case 3: return chan_analyze_fft<false, false, 3>;
        default: return chan_analyze_fft<false, false, 0>;
    }
}
static chan_kernel_t chan_kernel(const ChanGeom &g) {
    if (g.oddA) {
        if (g.hop != g.M) return g.stage_in ? chan_analyze<1, 1, 1, 1> : g.taps_lds ? chan_analyze<0, 1, 1, 1> : chan_analyze<0, 0, 1, 1>;
        return g.stage_in ? chan_analyze<1, 1, 0, 1> : g.taps_lds ? chan_analyze<0, 1, 0, 1> : chan_analyze<0, 0, 0, 1>;
    }
    if (g.hop != g.M) return g.stage_in ? chan_analyze<1, 1, 1, 0> : g.taps_lds ? chan_analyze<0, 1, 1, 0> : chan_analyze<0, 0, 1, 0>;
    return g.stage_in ? chan_analyze<1, 1, 0, 0> : g.taps_lds ? chan_analyze<0, 1, 0, 0> : chan_analyze<0, 0, 0, 0>;
}

extern "C" int csdr_post_configure(csdr_post *p, int64_t sample_rate, int num_channels, int mode, int max_block_len, int max_blocks) {
    DeviceScope dev__(p ? p->ctx : nullptr);
    if (!p) return fail(CSDR_EINVAL, "post is null");
    if (sample_rate <= 0 || num_channels < 1 || max_block_len <= 0 || max_blocks <= 0) return fail(CSDR_EINVAL, "bad sizes");
    if (mode != CSDR_POST_SINGLE && mode != CSDR_POST_PFBCH && mode != CSDR_POST_PFBCH2) return fail(CSDR_EINVAL, "channelizer mode %d", mode);
    if ((mode == CSDR_POST_SINGLE) != (num_channels == 1)) return fail(CSDR_EINVAL, "SINGLE mode <=> num_channels == 1");
    if (max_block_len % num_channels) return fail(CSDR_EINVAL, "max_block_len must be a multiple of num_channels");
    if (num_channels > 1 && (num_channels & 1)) return fail(CSDR_EUNSUPPORTED, "odd numChannels %d (the reference only produces even counts, SoapySDRThread.cpp:676-693)", num_channels);
    hipStream_t st = p->ctx->lanes[LANE_POST];
    if (int rc = p->ctx->sync_all()) return rc;
    p->configured = false;
    p->cur = 0; p->seq = 0;
    for (int k = 0; k < csdr_post::kPostBufs; ++k) p->n_consumed[k] = 0;
    p->mode = mode; p->M = num_channels; p->sample_rate = sample_rate;
    p->chan_bw = sample_rate / num_channels;                       // integer division, SDRPostThread.cpp:408
    // samples per channel: one per M inputs, or one per M / 2 (firpfbch2, whose channels are handed on at 2 * chanBw, :510)
    p->hop = mode == CSDR_POST_PFBCH2 ? num_channels / 2 : num_channels;
    p->chan_rate = mode == CSDR_POST_SINGLE ? sample_rate : (mode == CSDR_POST_PFBCH2 ? 2 * p->chan_bw : p->chan_bw);
    p->max_block_len = max_block_len; p->max_blocks = max_blocks;
    const int M = p->M;
    // Row pitch of the channel-major output: a multiple of 16 samples (128 bytes, one cache line), so that every row starts on a line: a
    // 16-frame tile of the M = 1024 channelizer then stores exactly one whole line per channel row, where an unaligned pitch splits every such
    // store into two partial lines.  Measured on the M = 1024 channelizer, 60 M samples per launch (profiles/r04_row_pitch.txt): pitch 58338
    // samples (16-byte aligned rows) 0.25 of the HBM roofline, any multiple of 16 .. 2048 samples 0.38.  (The multiple is made odd as well: a
    // pitch that is a multiple of a large power of two could land the rows a workgroup writes together on few memory channels; on MI355X
    // that was measured NOT to matter -- 52096 = 2^7 * 407 samples runs at 0.38 too -- so this costs 16 samples per row and buys insurance.)
    {
        int64_t q = ((int64_t)max_blocks * (max_block_len / p->hop) + 15) / 16;
        if (!(q & 1)) ++q;
        p->chan_stride = q * 16;
    }
    if (lab_int("CSDR_ROW_PAD", 1) == 0) p->chan_stride = ((int64_t)max_blocks * (max_block_len / p->hop) + 1) & ~(int64_t)1;      // the round-3 pitch (A/B)
    const int out_off_kb = std::min(8192, lab_int("CSDR_OUT_OFFSET_KB", -1));                                  // (measurement build: where the output starts inside its allocation;
    p->out_off = (size_t)std::max(0, out_off_kb) * 128;                                        //  the allocation itself is the same for every offset up to 8 MB)
    if (int rc = p->out.reserve((size_t)p->chan_stride * M * csdr_post::kPostBufs + (out_off_kb >= 0 ? (size_t)1 << 20 : 0))) return rc;
    if (int rc = p->dc_state.reserve(2)) return rc;
    CSDR_HIP_TRY(hipMemsetAsync(p->dc_state.p, 0, 2 * sizeof(d2), st));
    p->dc_parity = 0;
    const int64_t dc_n = (mode == CSDR_POST_SINGLE) ? (int64_t)max_blocks * max_block_len : p->chan_stride;
    const size_t ntiles = (size_t)(dc_n / 16 + 2);          // channelizer tiles hold >= 16 frames when they emit end values
    if (int rc = p->tile_end.reserve(ntiles)) return rc;
    // iirfilt_crcf_create_dc_blocker(0.0005f): b = {1, -1}, a = {1, -1 + alpha}  (float)  ->  v = x - a1 v'
    const float a1 = -1.0f + 0.0005f;
    p->dc_c = -(double)a1;
    if (mode != CSDR_POST_SINGLE) {
        if (int rc = chan_geometry(M, p->hop, p->geom)) return rc;
        const ChanGeom &g = p->geom;
        // every channel count that is not 2 * odd and factors over the small radices takes the FFT kernel (kernels_chanfft.hpp);
        // CSDR_CHAN_FFT=0 keeps the two-factor direct-DFT kernel (A/B measurements, bit-for-bit routing tests)
        std::vector<int> fperm;
        // (firpfbch2 too, since round 5: the oversampled hop is two interleaved lattices of frames in the same kernel, M % 4 == 0)
        p->use_fft = (mode == CSDR_POST_PFBCH || mode == CSDR_POST_PFBCH2) && !g.p2 && lab_int("CSDR_CHAN_FFT", 1) != 0 &&
                     chanfft_plan(M, (size_t)p->ctx->lds_per_cu, lab_int("CSDR_CHANFFT_TF", 0), lab_int("CSDR_CHANFFT_THREADS", 0), p->fgeom, fperm, mode == CSDR_POST_PFBCH2);
        // prototype taps transposed to [n][c]: tapsT[n M + c] multiplies x[(t - n) M + c]
        std::vector<float> taps = mode == CSDR_POST_PFBCH2 ? design::channelizer2_taps((unsigned)M, 4, 60.0f)      // initPFBCH2 :463
                                                           : design::channelizer_taps((unsigned)M, 4, 60.0f);      // initPFBCH :406
        if (mode == CSDR_POST_PFBCH2) {
            std::vector<float> post = design::channelizer2_post((unsigned)M);
            if (int rc = p->post2.reserve((size_t)2 * M)) return rc;
            CSDR_HIP_TRY(hipMemcpy(p->post2.p, post.data(), post.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        std::vector<float> tapsT((size_t)kChanTaps * M);
        for (int c = 0; c < M; c++) for (int n = 0; n < kChanTaps; n++) tapsT[(size_t)n * M + c] = taps[(size_t)c * kChanTaps + n];
        std::vector<float2> twA((size_t)g.A * g.PA, make_float2(0.f, 0.f)), twB((size_t)g.B * g.PB, make_float2(0.f, 0.f)), twM((size_t)g.A * g.B);
        auto W = [](int64_t num, int den) { const double a = -2.0 * M_PI * (double)(num % den) / (double)den; return make_float2((float)std::cos(a), (float)std::sin(a)); };
        if (g.p2) {   // slot q: output pair k = q + 1 (q < H), k = 0 as (1, 0) (q == H), unused (0, 0) beyond
            const int H = (g.A - 1) / 2;
            twA.assign((size_t)H * g.PA, make_float2(0.f, 0.f));
            if (g.mx) {                                          // the matrix-pipe form's coefficient fragments travel in the table's place
                twA.assign((size_t)2 * kMxSteps * 64, make_float2(0.f, 0.f));
                chan_mx_table(g.A, reinterpret_cast<float *>(twA.data()));
            } else
            for (int c = 1; c <= H; c++) for (int q = 0; q <= H; q++) {
                const int k = q < H ? q + 1 : 0;
                const double a = 2.0 * M_PI * (double)(((int64_t)c * k) % g.A) / (double)g.A;
                twA[(size_t)(c - 1) * g.PA + q] = make_float2((float)std::cos(a), (float)std::sin(a));
            }
        } else if (g.oddA) {        // (cos, sin)(2 pi kp c / A) at [(c - 1) PA + kp - 1], c, kp = 1 .. (A - 1) / 2
            const int H = (g.A - 1) / 2;
            for (int c = 1; c <= H; c++) for (int kp = 1; kp <= H; kp++) {
                const double a = 2.0 * M_PI * (double)(((int64_t)c * kp) % g.A) / (double)g.A;
                twA[(size_t)(c - 1) * g.PA + kp - 1] = make_float2((float)std::cos(a), (float)std::sin(a));
            }
        } else
        for (int c1 = 0; c1 < g.A; c1++) for (int k1 = 0; k1 < g.A; k1++) twA[(size_t)c1 * g.PA + k1] = W((int64_t)c1 * k1, g.A);
        for (int c2 = 0; c2 < g.B; c2++) for (int k2 = 0; k2 < g.B; k2++) twB[(size_t)c2 * g.PB + k2] = W((int64_t)c2 * k2, g.B);
        for (int k1 = 0; k1 < g.A; k1++) for (int c2 = 0; c2 < g.B; c2++) twM[(size_t)k1 * g.B + c2] = W((int64_t)k1 * c2, M);
        if (p->use_fft) {        // one table W_M^i serves every pass: W_N^(j r) = W_M^(j r M / N)
            twM.resize((size_t)M);
            for (int i = 0; i < M; i++) twM[(size_t)i] = W(i, M);
            if (int rc = p->perm.reserve(fperm.size())) return rc;
            CSDR_HIP_TRY(hipMemcpyAsync(p->perm.p, fperm.data(), fperm.size() * sizeof(int), hipMemcpyHostToDevice, st));
            if (p->fgeom.bp || p->fgeom.dp) {      // tables of the chirp-z / direct prime pass: they travel in the place of firpfbch2's post factors (never both)
                const std::vector<float2> bt = p->fgeom.bp ? chanfft_blue_tables(p->fgeom) : kCfPrimeMx ? chanfft_direct_mx_tables(p->fgeom) : chanfft_direct_tables(p->fgeom);
                if (int rc = p->post2.reserve(bt.size())) return rc;
                CSDR_HIP_TRY(hipMemcpy(p->post2.p, bt.data(), bt.size() * sizeof(float2), hipMemcpyHostToDevice));
            }
        }
        if (int rc = p->taps.reserve(tapsT.size())) return rc;
        if (int rc = p->twA.reserve(twA.size())) return rc;
        if (int rc = p->twB.reserve(twB.size())) return rc;
        if (int rc = p->twM.reserve(twM.size())) return rc;
        CSDR_HIP_TRY(hipMemcpyAsync(p->taps.p, tapsT.data(), tapsT.size() * sizeof(float), hipMemcpyHostToDevice, st));
        CSDR_HIP_TRY(hipMemcpyAsync(p->twA.p, twA.data(), twA.size() * sizeof(float2), hipMemcpyHostToDevice, st));
        CSDR_HIP_TRY(hipMemcpyAsync(p->twB.p, twB.data(), twB.size() * sizeof(float2), hipMemcpyHostToDevice, st));
        CSDR_HIP_TRY(hipMemcpyAsync(p->twM.p, twM.data(), twM.size() * sizeof(float2), hipMemcpyHostToDevice, st));
        const size_t H = (size_t)kChanTaps * M - p->hop;
        if (int rc = p->hist0.reserve(H)) return rc;
        if (int rc = p->hist1.reserve(H)) return rc;
        CSDR_HIP_TRY(hipMemsetAsync(p->hist0.p, 0, H * sizeof(float2), st));
        CSDR_HIP_TRY(hipMemsetAsync(p->hist1.p, 0, H * sizeof(float2), st));
        CSDR_HIP_TRY(hipStreamSynchronize(st));   // host vectors above go out of scope
        const size_t lds = p->use_fft ? chanfft_lds_bytes(p->fgeom) : g.p2 ? chan_p2_lds_bytes(M, g.mx != 0, g.hop != M) : chan_lds_bytes(g);
        if (lds > 64 * 1024) CSDR_HIP_TRY(hipFuncSetAttribute(p->use_fft ? (const void *)chanfft_kernel(p->fgeom) : g.p2 ? (const void *)chan_p2_kernel(g) : (const void *)chan_kernel(g), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    p->hist_parity = 0;
    p->active_host.resize(M);
    for (int i = 0; i < M; i++) p->active_host[i] = i;
    p->active_dirty = true;
    if (int rc = p->active.reserve(M)) return rc;
    p->frequency = 0;
    post_update_channels(p);
    p->n_blocks = 0; p->block_len = 0;
    p->configured = true;
    return CSDR_OK;
}

// optional: restrict the channelizer to the channels that have consumers (reference: SDRPostThread.cpp:336-339)
extern "C" int csdr_post_set_active_channels(csdr_post *p, const int *channels, int n) {
    if (!p || !p->configured) return fail(CSDR_ESTATE, "post not configured");
    if (n < 0 || n > p->M + 1) return fail(CSDR_EINVAL, "bad channel count");
    std::vector<int> v;
    if (!channels) { v.resize(p->M); for (int i = 0; i < p->M; i++) v[i] = i; }
    else {
        v.assign(channels, channels + n);
        for (int &c : v) { if (c == p->M) c = p->M / 2; if (c < 0 || c >= p->M) return fail(CSDR_EINVAL, "channel %d out of range", c); }
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
    }
    if (!p->row_order.empty()) { p->row_order.clear(); p->active_dirty = true; }
    if (v != p->active_host) { p->active_host = v; p->active_dirty = true; }
    return CSDR_OK;
}
// Time-slab producers: store the rows of the listed channels one after the other IN THIS ORDER (row i = channels[i]; the other channels are
// not produced) -- with the channels grouped by owning rank the output buffer is the all-to-all's send buffer as it stands
// (csdr_post_exchange_rows then skips the export copy).  n = 0 returns to "row = channel".
extern "C" int csdr_post_set_row_order(csdr_post *p, const int *channels, int n) {
    if (!p || !p->configured || p->mode == CSDR_POST_SINGLE) return fail(CSDR_ESTATE, "post is not a configured channelizer");
    if (n < 0 || n > p->M || (n && !channels)) return fail(CSDR_EINVAL, "bad channel count");
    std::vector<int> order(channels, channels + n), sorted;
    for (int c : order) if (c < 0 || c >= p->M) return fail(CSDR_EINVAL, "channel %d out of range", c);
    sorted = order;
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return fail(CSDR_EINVAL, "a channel is listed twice");
    if (n == 0) { sorted.resize(p->M); for (int i = 0; i < p->M; i++) sorted[i] = i; }
    p->row_order = order; p->active_host = sorted; p->active_dirty = true;
    return CSDR_OK;
}
// output row of channel ch (-1: not produced)
static int post_row_of(const csdr_post *p, int ch) {
    if (p->row_order.empty()) return std::binary_search(p->active_host.begin(), p->active_host.end(), ch) ? ch : -1;
    for (size_t i = 0; i < p->row_order.size(); ++i) if (p->row_order[i] == ch) return (int)i;
    return -1;
}

// DC blocker over n samples: `have_ends` = the mini-tile end values (tile_len samples each) are already in tile_end
// (the channelizer wrote them); otherwise a first pass computes them per kDcTile samples.
static int run_dc_blocker(csdr_post *p, const float2 *x, float2 *y, int64_t n, bool have_ends, int tile_len) {
    d2 *s_in = p->dc_state.p + p->dc_parity, *s_out = p->dc_state.p + (p->dc_parity ^ 1);
    if (!have_ends) {
        tile_len = kDcTile;
        const int nt = (int)((n + kDcTile - 1) / kDcTile);
        CSDR_LAUNCH(p->ctx, LANE_POST, KID_DC_ENDS, dc_tile_ends, dim3(nt), dim3(kDcThreads), kDcLds, x, n, p->dc_c, p->tile_end.p);
    }
    const int64_t blk_len = (int64_t)(kDcTile / tile_len) * tile_len;     // whole mini-tiles per block
    const int nblocks = (int)((n + blk_len - 1) / blk_len);
    CSDR_LAUNCH(p->ctx, LANE_POST, KID_DC_APPLY, dc_apply, dim3(nblocks), dim3(kDcThreads), kDcLds, x, y, n, p->dc_c, tile_len, p->tile_end.p, s_in, s_out);
    p->dc_parity ^= 1;
    CSDR_HIP_TRY(hipGetLastError());
    return CSDR_OK;
}


extern "C" int csdr_post_execute(csdr_post *p, const float *iq, int iq_is_dev, int n_blocks, int block_len, int64_t frequency) {
    RangeScope range__("csdr_post_execute");
    DeviceScope dev__(p ? p->ctx : nullptr);
    if (!p || !p->configured) return fail(CSDR_ESTATE, "post not configured");
    if (!iq || n_blocks <= 0 || block_len <= 0) return fail(CSDR_EINVAL, "bad block arguments");
    if (n_blocks > p->max_blocks || block_len > p->max_block_len) return fail(CSDR_ERANGE, "batch %d x %d exceeds configured %d x %d", n_blocks, block_len, p->max_blocks, p->max_block_len);
    if (block_len % p->M) return fail(CSDR_EINVAL, "block_len %d is not a multiple of numChannels %d", block_len, p->M);
    // (refused before anything of the object changes: a failed call leaves the post holding its previous batch)
    if (p->mode != CSDR_POST_SINGLE && !p->row_order.empty() && p->dc_enabled && !p->active_host.empty() && p->active_host[0] == 0)
        return fail(CSDR_ESTATE, "packed rows are for time-slab producers: csdr_post_set_dc_blocker(0) first (channel 0's owner runs the DC blocker)");
    csdr_ctx *c = p->ctx;
    hipStream_t st = c->lanes[LANE_POST];
    const int64_t n = (int64_t)n_blocks * block_len;
    const float2 *x = (const float2 *)iq;
    if (int rc = c->lane_begin(LANE_POST)) return rc;
    if (!iq_is_dev) {
        if (int rc = p->stage_in.reserve((size_t)p->max_blocks * p->max_block_len)) return rc;
        CSDR_HIP_TRY(hipMemcpyAsync(p->stage_in.p, iq, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, st));
        x = p->stage_in.p;
    } else if ((uintptr_t)iq & 15) return fail(CSDR_EINVAL, "device IQ pointer must be 16-byte aligned");
    if (frequency != p->frequency || p->centers.empty()) { p->frequency = frequency; post_update_channels(p); }
    p->n_blocks = n_blocks; p->block_len = block_len;
    // next output buffer of the rotation: its previous readers (demodulator front-ends, three batches ago) must be done
    // (stages that share one stream are ordered by it: a single buffer keeps the working set inside the Infinity Cache)
    const bool rotating = p->rotate || !c->same(LANE_POST, LANE_FE);
    const int k = rotating ? (int)(p->seq % csdr_post::kPostBufs) : 0;
    if (rotating)
        for (int q = 0; q < p->n_consumed[k]; ++q) CSDR_HIP_TRY(hipStreamWaitEvent(st, p->ev_consumed[k][q], 0));
    p->n_consumed[k] = 0;
    float2 *out = post_buf(p, k);
    int rc = CSDR_OK;
    if (p->mode == CSDR_POST_SINGLE && p->raw) CSDR_HIP_TRY(hipMemcpyAsync(out, x, (size_t)n * sizeof(float2), hipMemcpyDeviceToDevice, st));
    else if (p->mode == CSDR_POST_SINGLE) rc = run_dc_blocker(p, x, out, n, false, 0);       // runSingleCH :284
    else {
        const int M = p->M;
        if (p->active_dirty) {
            std::vector<int> flags(M, 0);                                        // output row of the channel + 1; 0 = not produced
            if (p->row_order.empty()) for (int ch : p->active_host) flags[ch] = ch + 1;
            else for (size_t i = 0; i < p->row_order.size(); ++i) flags[p->row_order[i]] = (int)i + 1;
            CSDR_HIP_TRY(hipStreamSynchronize(st));                              // earlier launches still read the old flags
            CSDR_HIP_TRY(hipMemcpy(p->active.p, flags.data(), flags.size() * sizeof(int), hipMemcpyHostToDevice));
            p->active_dirty = false;
        }
        const int64_t n_frames = n / p->hop;
        float2 *hist = p->hist_parity ? p->hist1.p : p->hist0.p, *hist_new = p->hist_parity ? p->hist0.p : p->hist1.p;
        ChanGeom g = p->geom;
        g.fpw = p->use_fft ? p->fgeom.TF : g.TF;      // frames per workgroup (full tiles measured fastest on MI355X)
        const int ntiles = (int)((n_frames + g.fpw - 1) / g.fpw);
        // channel 0 carries the DC spike: it is blocked after de-interleave (:364-375); when the tile size allows, the
        // channelizer itself emits the per-tile end values the blocked scan needs
        const bool dc0 = p->dc_enabled && !p->active_host.empty() && p->active_host[0] == 0;
        const bool fused_ends = dc0 && g.fpw >= 16;
        if (p->use_fft) {
            // persistent workgroups (as many as are resident at once) walk over the tiles
            ChanFftGeom fg = p->fgeom;
            fg.xcd = lab_int("CSDR_CHANFFT_XCD", fg.xcd);
            const size_t lds = chanfft_lds_bytes(fg);
            const chanfft_kernel_t kf = chanfft_kernel(fg);
            const int wgs = std::min(ntiles, std::max(1, c->wg_slots(kf, fg.threads, lds) * lab_int("CSDR_CHANFFT_PCT", 100) / 100));
            CSDR_LAUNCH(c, LANE_POST, KID_CHAN_ANALYZE, kf, dim3(wgs), dim3(fg.threads), lds, x, hist, hist_new, p->taps.p,
                        p->twM.p, p->perm.p, p->active.p, fg, n_frames, out, p->chan_stride, fused_ends ? p->tile_end.p : (d2 *)nullptr, p->dc_c,
                        (p->mode == CSDR_POST_PFBCH2 || fg.bp || fg.dp) ? p->post2.p : (const float2 *)nullptr);
        } else if (g.p2) {
            // persistent workgroups: as many as are resident at once, each walks over tiles blockIdx.x, + gridDim.x, ...
            const chan_p2_kernel_t k2 = chan_p2_kernel(g);
            const int chan_pct = std::max(10, std::min(100, lab_int("CSDR_CHAN_PCT", 100)));
            const int wgs = std::min(ntiles, std::max(1, c->wg_slots(k2, g.threads, chan_p2_lds_bytes(M, g.mx != 0, g.hop != M)) * chan_pct / 100));
            g.xcd = lab_int("CSDR_CHAN_XCD", g.xcd);
            if (lab_int("CSDR_LAB_TRACE", 0)) fprintf(stderr, "[csdr lab] chan_analyze_p2 x=%p out=%p hist=%p taps=%p cs=%p twM=%p wgs=%d xcd=%d\n", (const void *)x, (void *)out, (void *)hist, (void *)p->taps.p, (void *)p->twA.p, (void *)p->twM.p, wgs, g.xcd);
            CSDR_LAUNCH(c, LANE_POST, KID_CHAN_ANALYZE, k2, dim3(wgs), dim3(g.threads), chan_p2_lds_bytes(M, g.mx != 0, g.hop != M), x, hist, hist_new, p->taps.p,
                        p->twA.p, p->twM.p, p->active.p, g, n_frames, out, p->chan_stride, fused_ends ? p->tile_end.p : (d2 *)nullptr, p->dc_c,
                        p->mode == CSDR_POST_PFBCH2 ? p->post2.p : (const float2 *)nullptr);
        } else {
        const chan_kernel_t kern = chan_kernel(g);
        CSDR_LAUNCH(c, LANE_POST, KID_CHAN_ANALYZE, kern, dim3(ntiles), dim3(g.threads), chan_lds_bytes(g), x, hist, hist_new, p->taps.p,
                    p->twA.p, p->twB.p, p->twM.p, p->active.p, g, n_frames, out, p->chan_stride, fused_ends ? p->tile_end.p : (d2 *)nullptr, p->dc_c,
                    p->mode == CSDR_POST_PFBCH2 ? p->post2.p : (const float2 *)nullptr);
        }
        p->hist_parity ^= 1;
        CSDR_HIP_TRY(hipGetLastError());
        if (dc0) rc = run_dc_blocker(p, out, out, n_frames, fused_ends, g.fpw);
    }
    if (rc) return rc;
    if (int rc2 = c->signal(p->ev_ready[k], LANE_POST, LANE_FE)) return rc2;
    p->cur = k;
    p->seq++;
    return CSDR_OK;
}

extern "C" int64_t csdr_post_channel_bandwidth(const csdr_post *p) { return p ? (p->M == 1 ? p->sample_rate : p->chan_bw) : 0; }
extern "C" int64_t csdr_post_channel_rate(const csdr_post *p) { return p ? p->chan_rate : 0; }
extern "C" int csdr_post_num_channels(const csdr_post *p) { return p ? p->M : 0; }
extern "C" const char *csdr_post_kernel_name(const csdr_post *p) {
    if (!p || !p->configured) return "";
    return p->mode == CSDR_POST_SINGLE ? "dc_blocker" : p->use_fft ? "chan_analyze_fft" : p->geom.p2 ? "chan_analyze_p2" : "chan_analyze";
}
extern "C" int64_t csdr_post_channel_center(const csdr_post *p, int i) {
    if (!p || i < 0 || i >= (int)p->centers.size()) return 0;
    return p->centers[i];
}
extern "C" int csdr_post_channel_at(const csdr_post *p, int64_t frequency_in) {   // getChannelAt, :128-139
    if (!p || !p->configured) return -1;
    if (p->M == 1) return 0;
    int chan = -1;
    long long minDelta = p->sample_rate;
    for (int i = 0; i < p->M + 1; i++) {
        long long fdelta = std::llabs((long long)frequency_in - (long long)p->centers[i]);
        if (fdelta < minDelta) { minDelta = fdelta; chan = i; }
    }
    return chan;
}
extern "C" int csdr_post_read_channel(csdr_post *p, int ch, float *host_out, int cap_samples, int *n) {
    DeviceScope dev__(p ? p->ctx : nullptr);
    if (!p || !p->configured || !host_out || !n) return fail(CSDR_EINVAL, "bad argument");
    if (ch == p->M && p->M > 1) ch = p->M / 2;
    if (ch < 0 || ch >= p->M) return fail(CSDR_EINVAL, "channel out of range");
    const int64_t cnt = (int64_t)p->n_blocks * (p->block_len / p->hop);
    if (cnt > cap_samples) return fail(CSDR_ERANGE, "need %lld samples", (long long)cnt);
    hipStream_t st = p->ctx->lanes[LANE_POST];
    const int row = p->row_order.empty() ? ch : post_row_of(p, ch);
    if (row < 0) return fail(CSDR_EINVAL, "channel %d is not produced", ch);
    CSDR_HIP_TRY(hipMemcpyAsync(host_out, post_buf(p, p->cur) + (int64_t)row * p->chan_stride, (size_t)cnt * sizeof(float2), hipMemcpyDeviceToHost, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    *n = (int)cnt;
    return CSDR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Time-slab sharding of ONE stream over several GPUs (SURVEY 8e option 2; host side: cubicsdr_amd/parallel.py SlabStream).
// A producer rank runs the channelizer over ITS blocks of the batch for all channels -- csdr_post_set_history gives it the input
// samples in front of its slab, csdr_post_set_dc_blocker(0) leaves channel 0 unfiltered -- and csdr_post_export_rows packs the
// rows each peer owns for the all-to-all.  The owner assembles its channels' rows from every peer's frames into a second post
// object (import_begin / import_rows / import_commit: commit runs the carried DC blocker over channel 0 when it owns it), which
// its demodulator bank then reads exactly like an executed one.
// ---------------------------------------------------------------------------------------------------------------------------
namespace csdr {
CSDR_KERNEL __launch_bounds__(256) void rows_copy(const float2 *__restrict__ src, int64_t src_stride, const int *__restrict__ src_rows,
                                                 float2 *__restrict__ dst, int64_t dst_stride, const int *__restrict__ dst_rows, int64_t n_frames) {
    const int r = blockIdx.y;
    const float2 *s = src + (int64_t)(src_rows ? src_rows[r] : r) * src_stride;
    float2 *d = dst + (int64_t)(dst_rows ? dst_rows[r] : r) * dst_stride;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_frames; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
}
}  // namespace csdr
static int post_row_list(csdr_post *p, const int *channels, int n, const int **dev_list) {
    if (n <= 0 || n > p->M) return fail(CSDR_EINVAL, "bad channel count");
    for (int i = 0; i < n; ++i) if (channels[i] < 0 || channels[i] >= p->M) return fail(CSDR_EINVAL, "channel %d out of range", channels[i]);
    std::vector<int> key(channels, channels + n);
    auto it = p->rowlists.find(key);
    if (it == p->rowlists.end()) {
        if (p->rowlists.size() >= 64) return fail(CSDR_ERANGE, "too many distinct channel lists");
        int *d = nullptr;
        if (hipMalloc((void **)&d, (size_t)n * sizeof(int)) != hipSuccess) return fail(CSDR_ENOMEM, "channel list");
        if (hipMemcpy(d, channels, (size_t)n * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d);
            return fail(CSDR_EHIP, "channel list upload");
        }
        it = p->rowlists.emplace(std::move(key), d).first;
    }
    *dev_list = it->second;
    return CSDR_OK;
}
extern "C" int csdr_post_set_history(csdr_post *p, const float *dev_tail, int64_t n_samples) {
    DeviceScope dev__(p ? p->ctx : nullptr);
    if (!p || !p->configured || p->mode == CSDR_POST_SINGLE) return fail(CSDR_ESTATE, "post is not a configured channelizer");
    if (!dev_tail || n_samples < 0) return fail(CSDR_EINVAL, "bad argument");
    const int64_t H = (int64_t)kChanTaps * p->M - p->hop;
    hipStream_t st = p->ctx->lanes[LANE_POST];
    if (int rc = p->ctx->lane_begin(LANE_POST)) return rc;
    float2 *hist = p->hist_parity ? p->hist1.p : p->hist0.p;                 // what the next execute reads in front of its input
    const int64_t take = std::min(H, n_samples);
    if (take < H) CSDR_HIP_TRY(hipMemsetAsync(hist, 0, (size_t)(H - take) * sizeof(float2), st));
    if (take) CSDR_HIP_TRY(hipMemcpyAsync(hist + (H - take), (const float2 *)dev_tail + (n_samples - take), (size_t)take * sizeof(float2), hipMemcpyDeviceToDevice, st));
    return CSDR_OK;
}
extern "C" int csdr_post_history_length(const csdr_post *p) { return (p && p->configured && p->mode != CSDR_POST_SINGLE) ? kChanTaps * p->M - p->hop : 0; }
extern "C" int csdr_post_set_dc_blocker(csdr_post *p, int enabled) {
    if (!p) return fail(CSDR_EINVAL, "null argument");
    p->dc_enabled = enabled != 0;
    return CSDR_OK;
}
extern "C" int csdr_post_export_rows(csdr_post *p, const int *channels, int n, float *dst_dev, int64_t dst_stride) {
    DeviceScope dev__(p ? p->ctx : nullptr);
    if (!p || !p->configured || p->n_blocks <= 0) return fail(CSDR_ESTATE, "post has no data");
    if (!channels || !dst_dev) return fail(CSDR_EINVAL, "null argument");
    const int64_t nf = (int64_t)p->n_blocks * (p->block_len / p->hop);
    if (dst_stride < nf) return fail(CSDR_EINVAL, "destination stride %lld below %lld frames", (long long)dst_stride, (long long)nf);
    const int *rows = nullptr;
    if (p->row_order.empty()) {
        if (int rc = post_row_list(p, channels, n, &rows)) return rc;
    } else {                                                     // packed rows: the listed channels' positions
        std::vector<int> pos((size_t)std::max(n, 0));
        for (int i = 0; i < n; ++i) if ((pos[(size_t)i] = post_row_of(p, channels[i])) < 0) return fail(CSDR_EINVAL, "channel %d is not produced", channels[i]);
        if (int rc = post_row_list(p, pos.data(), n, &rows)) return rc;
    }
    CSDR_LAUNCH(p->ctx, LANE_POST, KID_ROWS_COPY, rows_copy, dim3((unsigned)std::min<int64_t>(64, (nf + 255) / 256), n), dim3(256), 0,
                post_buf(p, p->cur), p->chan_stride, rows, (float2 *)dst_dev, dst_stride, (const int *)nullptr, nf);
    CSDR_HIP_TRY(hipGetLastError());
    return CSDR_OK;
}
extern "C" int csdr_post_import_begin(csdr_post *p, int n_blocks, int block_len, int64_t frequency) {
    DeviceScope dev__(p ? p->ctx : nullptr);
    if (!p || !p->configured || p->mode == CSDR_POST_SINGLE) return fail(CSDR_ESTATE, "post is not a configured channelizer");
    if (!p->row_order.empty()) return fail(CSDR_ESTATE, "a post with packed rows is a producer: import into a second post object");
    if (n_blocks <= 0 || n_blocks > p->max_blocks || block_len <= 0 || block_len > p->max_block_len || block_len % p->M) return fail(CSDR_ERANGE, "bad batch %d x %d", n_blocks, block_len);
    csdr_ctx *c = p->ctx;
    hipStream_t st = c->lanes[LANE_POST];
    if (int rc = c->lane_begin(LANE_POST)) return rc;
    if (frequency != p->frequency || p->centers.empty()) { p->frequency = frequency; post_update_channels(p); }
    p->n_blocks = n_blocks; p->block_len = block_len;
    const int k = c->same(LANE_POST, LANE_FE) ? 0 : (int)(p->seq % csdr_post::kPostBufs);
    if (!c->same(LANE_FE, LANE_POST))
        for (int q = 0; q < p->n_consumed[k]; ++q) CSDR_HIP_TRY(hipStreamWaitEvent(st, p->ev_consumed[k][q], 0));
    p->n_consumed[k] = 0;
    p->import_k = k;
    return CSDR_OK;
}
extern "C" int csdr_post_import_rows(csdr_post *p, const int *channels, int n, const float *src_dev, int64_t src_stride, int64_t frame0, int64_t n_frames) {
    DeviceScope dev__(p ? p->ctx : nullptr);
    if (!p || p->import_k < 0) return fail(CSDR_ESTATE, "no import in progress");
    if (!channels || !src_dev) return fail(CSDR_EINVAL, "null argument");
    const int64_t nf = (int64_t)p->n_blocks * (p->block_len / p->hop);
    if (frame0 < 0 || n_frames < 0 || frame0 + n_frames > nf || src_stride < n_frames) return fail(CSDR_ERANGE, "frames [%lld, +%lld) outside the batch of %lld", (long long)frame0, (long long)n_frames, (long long)nf);
    if (n_frames == 0) return CSDR_OK;
    const int *rows = nullptr;
    if (int rc = post_row_list(p, channels, n, &rows)) return rc;
    CSDR_LAUNCH(p->ctx, LANE_POST, KID_ROWS_COPY, rows_copy, dim3((unsigned)std::min<int64_t>(64, (n_frames + 255) / 256), n), dim3(256), 0,
                (const float2 *)src_dev, src_stride, (const int *)nullptr, post_buf(p, p->import_k) + frame0, p->chan_stride, rows, n_frames);
    CSDR_HIP_TRY(hipGetLastError());
    return CSDR_OK;
}
extern "C" int csdr_post_import_commit(csdr_post *p) {
    DeviceScope dev__(p ? p->ctx : nullptr);
    if (!p || p->import_k < 0) return fail(CSDR_ESTATE, "no import in progress");
    csdr_ctx *c = p->ctx;
    const int k = p->import_k;
    p->import_k = -1;
    const int64_t nf = (int64_t)p->n_blocks * (p->block_len / p->hop);
    float2 *out = post_buf(p, k);
    if (p->dc_enabled && !p->active_host.empty() && p->active_host[0] == 0)
        if (int rc = run_dc_blocker(p, out, out, nf, false, 0)) return rc;
    if (int rc2 = c->signal(p->ev_ready[k], LANE_POST, LANE_FE)) return rc2;
    p->cur = k;
    p->seq++;
    return CSDR_OK;
}

