// csdr_api.hip -- implementation of include/csdr_hip.h (gfx950).  Host-side bookkeeping mirrors the reference's
// control flow (file:line cited per function); all sample arithmetic is in the kernels_*.hpp kernels.
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <map>
#include <memory>

#include "common.hpp"
#include "design.hpp"
#include "kernels_demod.hpp"
#include "kernels_fms.hpp"
#include "kernels_post.hpp"
#include "kernels_chanfft.hpp"
#include "kernels_spec.hpp"
#include "kernels_io.hpp"

using namespace csdr;

// =================================================================================================== context
extern "C" int csdr_abi_version(void) { return 1; }

extern "C" const char *csdr_strerror(int code) {
    switch (code) {
        case CSDR_OK: return "ok";
        case CSDR_EINVAL: return "invalid argument";
        case CSDR_ENOMEM: return "out of memory";
        case CSDR_EHIP: return "HIP runtime error";
        case CSDR_ESTATE: return "object not configured";
        case CSDR_ERANGE: return "capacity exceeded";
        case CSDR_EUNSUPPORTED: return "not supported yet";
        default: return "unknown error";
    }
}
extern "C" const char *csdr_last_error(void) { return last_error_ref().c_str(); }

extern "C" int csdr_ctx_create(int device, void *hip_stream, csdr_ctx **out) {
    if (!out) return fail(CSDR_EINVAL, "out is null");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(CSDR_EHIP, "no HIP device available: the HIP path cannot run (there is no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(CSDR_EINVAL, "device %d out of range (%d devices)", device, ndev);
    CSDR_HIP_TRY(hipSetDevice(device));
    std::unique_ptr<csdr_ctx> c(new csdr_ctx());
    c->device = device;
    if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->n_cu < 1) c->n_cu = 256;
    // Physical streams.  The five stage lanes are folded onto three streams by default: channelizer | demodulators | spectrum
    // (the reference's own cut: SDRPostThread, the demodulator threads, the spectrum thread), so that the channelizer of batch
    // n + 1 runs next to the demodulators of batch n.  Measured on MI355X / ROCm 7.2, C3: 1 / 2 / 3 / 5 streams = 51.0 / 50.9 /
    // 52.8 / 52.3 GS/s at 128-block batches and 5.8 / 7.9 / 9.3 / 8.9 thousand one-block calls per second.  CSDR_STREAMS = 1 | 2 |
    // 3 | 5 selects a folding (2: {channelizer + demodulators} | {spectrum}; 5: one stream per stage); the event protocol is the
    // same for all.
    int want = 3;
    if (const char *e = getenv("CSDR_STREAMS")) want = atoi(e);
    if (want != 1 && want != 2 && want != 3 && want != 5) want = 3;
    static const int kMap[6][LANE_COUNT] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 1, 1}, {0, 1, 1, 2, 2}, {0, 0, 0, 0, 0}, {0, 1, 2, 3, 4}};
    c->n_phys = want;
    for (int l = 0; l < want; ++l) {
        CSDR_HIP_TRY(hipStreamCreateWithFlags(&c->phys[l], hipStreamNonBlocking));
        CSDR_HIP_TRY(hipEventCreateWithFlags(&c->ev_lane[l], hipEventDisableTiming));
    }
    for (int l = 0; l < LANE_COUNT; ++l) c->lanes[l] = c->phys[kMap[want][l]];
    if (hip_stream == CSDR_STREAM_NULL) { c->stream = nullptr; c->own_stream = false; }          // the device's null stream
    else if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else { c->stream = c->phys[0]; c->own_stream = true; }      // a private boundary stream is just the first stage stream
    for (int l = 0; l < LANE_COUNT; ++l) CSDR_HIP_TRY(hipEventCreateWithFlags(&c->ev_in[l], hipEventDisableTiming));
    CSDR_HIP_TRY(hipEventCreate(&c->ev0));
    CSDR_HIP_TRY(hipEventCreate(&c->ev1));
    std::vector<float> tab = design::nco_sine_table();
    if (int rc = c->sintab.reserve(1024)) return rc;
    CSDR_HIP_TRY(hipMemcpy(c->sintab.p, tab.data(), 1024 * sizeof(float), hipMemcpyHostToDevice));
    *out = c.release();
    return CSDR_OK;
}
extern "C" void csdr_ctx_destroy(csdr_ctx *c) {
    DeviceScope dev__(c);
    if (!c) return;
    (void)c->sync_all();
    c->sintab.release();
    for (auto &r : c->prof_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : c->prof_pool) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (int l = 0; l < LANE_COUNT; ++l) if (c->ev_in[l]) (void)hipEventDestroy(c->ev_in[l]);
    for (int l = 0; l < c->n_phys; ++l) {
        if (c->ev_lane[l]) (void)hipEventDestroy(c->ev_lane[l]);
        if (c->phys[l]) (void)hipStreamDestroy(c->phys[l]);
    }
    delete c;
}
extern "C" int csdr_ctx_synchronize(csdr_ctx *c) {
    DeviceScope dev__(c);
    if (!c) return fail(CSDR_EINVAL, "ctx is null");
    return c->sync_all();
}
extern "C" int csdr_ctx_join(csdr_ctx *c) {
    DeviceScope dev__(c);
    if (!c) return fail(CSDR_EINVAL, "ctx is null");
    return c->join();
}
extern "C" int csdr_ctx_owns_stream(const csdr_ctx *c) { return c && c->own_stream ? 1 : 0; }
extern "C" void *csdr_ctx_stream(csdr_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" int csdr_ctx_timer_start(csdr_ctx *c) {
    DeviceScope dev__(c);
    if (!c) return fail(CSDR_EINVAL, "ctx is null");
    if (int rc = c->join()) return rc;
    CSDR_HIP_TRY(hipEventRecord(c->ev0, c->stream));
    // nothing the stage streams receive from now on may start before the timer's start mark
    for (int l = 0; l < c->n_phys; ++l) if (c->phys[l] != c->stream) CSDR_HIP_TRY(hipStreamWaitEvent(c->phys[l], c->ev0, 0));
    return CSDR_OK;
}
extern "C" int csdr_ctx_timer_stop(csdr_ctx *c, float *ms) {
    DeviceScope dev__(c);
    if (!c || !ms) return fail(CSDR_EINVAL, "null argument");
    if (int rc = c->join()) return rc;
    CSDR_HIP_TRY(hipEventRecord(c->ev1, c->stream));
    CSDR_HIP_TRY(hipEventSynchronize(c->ev1));
    CSDR_HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return CSDR_OK;
}
// ---- per-kernel HIP-event profile (bench.py roofline leg) ----
static const char *kKernelNames[KID_COUNT] = {
    "chan_analyze", "dc_tile_ends", "dc_apply", "rows_copy",
    "demod_frontend_generic", "demod_frontend_s3", "demod_frontend_s4", "demod_frontend_s5", "demod_frontend_s6", "demod_frontend_s56", "demod_frontend_interp",
    "demod_modem", "demod_gain_scan", "fms_stages", "demod_audio_interp", "fms_out", "audio_egress",
    "spec_fft_radix", "spec_fft_rows", "spec_average", "spec_extrema", "spec_display", "spec_misc"};
static int prof_drain(csdr_ctx *c) {
    if (int rc = c->sync_all()) return rc;
    std::lock_guard<std::mutex> lk(c->prof_mu);
    for (auto &r : c->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { c->prof_ms[r.id] += ms; c->prof_n[r.id] += 1; }
        c->prof_pool.push_back(r.a); c->prof_pool.push_back(r.b);
    }
    c->prof_pending.clear();
    return CSDR_OK;
}
extern "C" int csdr_ctx_profile_enable(csdr_ctx *c, int on) {
    DeviceScope dev__(c);
    if (!c) return fail(CSDR_EINVAL, "ctx is null");
    if (int rc = prof_drain(c)) return rc;
    c->prof_on = on != 0;
    c->prof_period = on > 1 ? on : 1;
    if (on) for (int i = 0; i < KID_COUNT; i++) { c->prof_ms[i] = 0.0; c->prof_n[i] = 0; c->prof_seen[i] = 0; }
    return CSDR_OK;
}
extern "C" int csdr_ctx_profile_launches(csdr_ctx *c, int id, int64_t *launches) {
    if (!c || id < 0 || id >= KID_COUNT || !launches) return fail(CSDR_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lk(c->prof_mu);
    *launches = (int64_t)c->prof_seen[id];
    return CSDR_OK;
}
extern "C" int csdr_ctx_profile_num_kernels(void) { return KID_COUNT; }
extern "C" const char *csdr_ctx_profile_kernel_name(int id) { return (id >= 0 && id < KID_COUNT) ? kKernelNames[id] : ""; }
extern "C" int csdr_ctx_profile_fetch(csdr_ctx *c, int id, double *total_ms, int64_t *launches) {
    DeviceScope dev__(c);
    if (!c || id < 0 || id >= KID_COUNT || !total_ms || !launches) return fail(CSDR_EINVAL, "bad argument");
    if (int rc = prof_drain(c)) return rc;
    *total_ms = c->prof_ms[id]; *launches = c->prof_n[id];
    return CSDR_OK;
}

extern "C" int csdr_dev_alloc(csdr_ctx *c, uint64_t bytes, void **dev) {
    DeviceScope dev__(c);
    if (!c || !dev) return fail(CSDR_EINVAL, "null argument");
    if (hipMalloc(dev, bytes) != hipSuccess) return fail(CSDR_ENOMEM, "hipMalloc(%llu) failed", (unsigned long long)bytes);
    return CSDR_OK;
}
extern "C" int csdr_dev_free(csdr_ctx *c, void *dev) {
    DeviceScope dev__(c); (void)c; if (dev) CSDR_HIP_TRY(hipFree(dev)); return CSDR_OK; }
extern "C" int csdr_dev_upload(csdr_ctx *c, void *dev, const void *host, uint64_t bytes) {
    DeviceScope dev__(c);
    if (!c) return fail(CSDR_EINVAL, "ctx is null");
    CSDR_HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    CSDR_HIP_TRY(hipStreamSynchronize(c->stream));
    return CSDR_OK;
}
extern "C" int csdr_dev_download(csdr_ctx *c, void *host, const void *dev, uint64_t bytes) {
    DeviceScope dev__(c);
    if (!c) return fail(CSDR_EINVAL, "ctx is null");
    if (int rc = c->sync_all()) return rc;
    CSDR_HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    CSDR_HIP_TRY(hipStreamSynchronize(c->stream));
    return CSDR_OK;
}

extern "C" int csdr_host_register(csdr_ctx *c, void *host, uint64_t bytes) {
    DeviceScope dev__(c);
    if (!c || !host || !bytes) return fail(CSDR_EINVAL, "bad argument");
    CSDR_HIP_TRY(hipHostRegister(host, bytes, hipHostRegisterDefault));
    return CSDR_OK;
}
extern "C" int csdr_host_unregister(csdr_ctx *c, void *host) {
    DeviceScope dev__(c);
    if (!c || !host) return fail(CSDR_EINVAL, "bad argument");
    CSDR_HIP_TRY(hipHostUnregister(host));
    return CSDR_OK;
}

// =================================================================================================== SDRPostThread
struct csdr_post {
    csdr_ctx *ctx = nullptr;
    bool configured = false;
    int mode = CSDR_POST_SINGLE, M = 1;
    int64_t sample_rate = 0, chan_bw = 0, chan_rate = 0, frequency = 0;
    int hop = 1;                             // input samples per output sample of a channel: M, M / 2 (PFBCH2) or 1 (single)
    int max_block_len = 0, max_blocks = 0;
    int64_t chan_stride = 0;                 // samples per channel row in `out` (even: rows stay 16-byte aligned)
    int n_blocks = 0, block_len = 0;         // of the last execute
    std::vector<int64_t> centers;            // chanCenters[M + 1]
    std::vector<int> active_host;            // sorted list of produced channels
    bool active_dirty = true;
    ChanGeom geom{};
    bool use_fft = false;                    // critically sampled, M = 2^a 3^b 5^c 7^d 11^e 13^f: chan_analyze_fft (kernels_chanfft.hpp)
    ChanFftGeom fgeom{};
    DevBuf<int> perm;                        // chan_analyze_fft: position after the last pass -> channel
    // `out` holds kPostBufs batches in rotation: the channelizer fills the next one while the demodulators still read
    // the previous (the reference hands ReBuffer blocks through a queue, SDRPostThread.cpp:341-396)
    static constexpr int kPostBufs = 3, kMaxConsumers = 4;
    int cur = 0;                             // buffer of the last execute
    uint64_t seq = 0;
    hipEvent_t ev_ready[kPostBufs] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_consumed[kPostBufs][kMaxConsumers] = {};
    int n_consumed[kPostBufs] = {0, 0, 0};
    DevBuf<float2> out, hist0, hist1, stage_in, twA, twB, twM, post2;
    DevBuf<float> taps;
    DevBuf<int> active;                      // [M] flags
    DevBuf<d2> dc_state, tile_end;           // dc_state[2]: ping-pong carried state
    int hist_parity = 0, dc_parity = 0;
    double dc_c = 0.0;                       // feedback coefficient of the DC blocker recurrence
    bool raw = false;                        // internal (zoomed spectrum view): SINGLE mode hands the input on unfiltered
    bool dc_enabled = true;                  // csdr_post_set_dc_blocker: a time-slab producer leaves channel 0 to the rank that owns it
    int import_k = -1;                       // buffer being assembled by csdr_post_import_begin .. commit
    std::map<std::vector<int>, int *> rowlists;   // device copies of the channel lists export / import calls name (a handful, reused every batch)
};

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
    if (hop == M && g.B == 2 && g.oddA && g.A <= 63 && !getenv("CSDR_CHAN_GENERIC")) {
        // whole transform of a frame inside one lane (kernels_post.hpp, chan_analyze_p2): (A - 1) / 2 output pairs + the k = 0
        // pseudo pair, split evenly over (up to) four passes of at most eight slots
        const int slots = (g.A - 1) / 2 + 1;
        g.p2 = 1;
        g.KA = std::min(8, (slots + kP2Waves - 1) / kP2Waves);
        g.nkA = (slots + g.KA - 1) / g.KA;
        g.PA = g.nkA * g.KA;
        g.TF = kP2Frames; g.lgTF = 6; g.S = M; g.taps_lds = 0; g.stage_in = 1; g.threads = kP2Threads;
        // The prime-length DFT is matrix-shaped work: A >= 17 runs it on the fp32 matrix pipe (bit-identical results; the kernel was
        // bound by vector-ALU issue, not by bandwidth).  CSDR_CHAN_MX=0 keeps the vector form (A/B measurements).
        // CSDR_CHAN_MX: 0 vector form | 1 matrix pipe, 64-frame tiles, stores straight from the accumulators | 2 the same with stores staged
        // through LDS (512-byte row runs) | 3 / 4: as 1 / 2 with 32-frame tiles in four-wave workgroups (four per CU instead of two)
        g.mx = 0;
        if (const char *e = getenv("CSDR_CHAN_MX")) g.mx = g.A >= 17 ? std::max(0, std::min(4, atoi(e))) : 0;
        // vector form: s = x_c + x_{A-c} / d = x_c - x_{A-c} are formed once, in the FIR phase (a lane trade), instead of by all eight waves in
        // their DFT passes, and the second row request of a trip is unconditional (no register-set copies).  Same sums, same order: the
        // output is bit-identical.  Measured on C3: 0.581 -> 0.564 ms (before the window went straight into registers; within noise after).
        // CSDR_CHAN_ALT=0 selects the round-2 form of the DFT phase (A/B, bit-identity test).
        // bit 1: the channel rows are stored with the streaming hint (nothing in this kernel reads them again): they stay out of the way of
        // the window rows two waves share -- 12.6 -> 10.5 B/sample fetched on C3, same kernel time
        g.alt = getenv("CSDR_CHAN_ALT") ? (atoi(getenv("CSDR_CHAN_ALT")) & 3) : 3;
        if (g.mx >= 3) { g.TF = 32; g.lgTF = 5; g.threads = P2Tile<32>::threads; }
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
                                 int64_t, float2 *, int64_t, d2 *, double);
static chan_p2_kernel_t chan_p2_kernel(const ChanGeom &g) {
    if (g.mx >= 3) return chan_analyze_p2<4, true, 32>;
    if (g.mx) return chan_analyze_p2<4, true, 64>;
#define CSDR_P2_CASE(K_) case K_: return chan_analyze_p2<K_>
    switch (g.KA) {
        CSDR_P2_CASE(1); CSDR_P2_CASE(2); CSDR_P2_CASE(3); CSDR_P2_CASE(4); CSDR_P2_CASE(5); CSDR_P2_CASE(6); CSDR_P2_CASE(7);
        default: return chan_analyze_p2<8>;
    }
#undef CSDR_P2_CASE
}
typedef void (*chanfft_kernel_t)(const float2 *, const float2 *, float2 *, const float *, const float2 *, const int *, const int *, ChanFftGeom, int64_t,
                                 float2 *, int64_t, d2 *, double);
static chanfft_kernel_t chanfft_kernel(const ChanFftGeom &) { return chan_analyze_fft; }
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
    p->chan_stride = ((int64_t)max_blocks * (max_block_len / p->hop) + 1) & ~(int64_t)1;
    if (int rc = p->out.reserve((size_t)p->chan_stride * M * csdr_post::kPostBufs)) return rc;
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
        p->use_fft = mode == CSDR_POST_PFBCH && !g.p2 && lab_int("CSDR_CHAN_FFT", 1) != 0 &&
                     chanfft_plan(M, (size_t)p->ctx->lds_per_cu, lab_int("CSDR_CHANFFT_TF", 0), lab_int("CSDR_CHANFFT_THREADS", 0), p->fgeom, fperm);
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
        if (g.p2 && g.mx) {  // coefficient fragments of the matrix-pipe form (kernels_post.hpp: chan_mx_table)
            twA.assign((size_t)2 * kMxSteps * 64, make_float2(0.f, 0.f));        // 2 x 2 x kMxSteps x 64 floats
            chan_mx_table(g.A, reinterpret_cast<float *>(twA.data()));
        } else if (g.p2) {   // slot q: output pair k = q + 1 (q < H), k = 0 as (1, 0) (q == H), unused (0, 0) beyond
            const int H = (g.A - 1) / 2;
            twA.assign((size_t)H * g.PA, make_float2(0.f, 0.f));
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
        const size_t lds = p->use_fft ? chanfft_lds_bytes(p->fgeom) : g.p2 ? chan_p2_lds_bytes(M, g.TF) : chan_lds_bytes(g);
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
    if (v != p->active_host) { p->active_host = v; p->active_dirty = true; }
    return CSDR_OK;
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

static float2 *post_buf(const csdr_post *p, int k) { return p->out.p + (size_t)k * p->chan_stride * p->M; }

extern "C" int csdr_post_execute(csdr_post *p, const float *iq, int iq_is_dev, int n_blocks, int block_len, int64_t frequency) {
    DeviceScope dev__(p ? p->ctx : nullptr);
    if (!p || !p->configured) return fail(CSDR_ESTATE, "post not configured");
    if (!iq || n_blocks <= 0 || block_len <= 0) return fail(CSDR_EINVAL, "bad block arguments");
    if (n_blocks > p->max_blocks || block_len > p->max_block_len) return fail(CSDR_ERANGE, "batch %d x %d exceeds configured %d x %d", n_blocks, block_len, p->max_blocks, p->max_block_len);
    if (block_len % p->M) return fail(CSDR_EINVAL, "block_len %d is not a multiple of numChannels %d", block_len, p->M);
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
    const int k = c->same(LANE_POST, LANE_FE) ? 0 : (int)(p->seq % csdr_post::kPostBufs);
    if (!c->same(LANE_FE, LANE_POST))
        for (int q = 0; q < p->n_consumed[k]; ++q) CSDR_HIP_TRY(hipStreamWaitEvent(st, p->ev_consumed[k][q], 0));
    p->n_consumed[k] = 0;
    float2 *out = post_buf(p, k);
    int rc = CSDR_OK;
    if (p->mode == CSDR_POST_SINGLE && p->raw) CSDR_HIP_TRY(hipMemcpyAsync(out, x, (size_t)n * sizeof(float2), hipMemcpyDeviceToDevice, st));
    else if (p->mode == CSDR_POST_SINGLE) rc = run_dc_blocker(p, x, out, n, false, 0);       // runSingleCH :284
    else {
        const int M = p->M;
        if (p->active_dirty) {
            std::vector<int> flags(M, 0);
            for (int ch : p->active_host) flags[ch] = 1;
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
                        p->twM.p, p->perm.p, p->active.p, fg, n_frames, out, p->chan_stride, fused_ends ? p->tile_end.p : (d2 *)nullptr, p->dc_c);
        } else if (g.p2) {
            // persistent workgroups: as many as are resident at once, each walks over tiles blockIdx.x, + gridDim.x, ...
            const chan_p2_kernel_t k2 = chan_p2_kernel(g);
            static const int chan_pct = getenv("CSDR_CHAN_PCT") ? std::max(10, std::min(100, atoi(getenv("CSDR_CHAN_PCT")))) : 100;
            const int wgs = std::min(ntiles, std::max(1, c->wg_slots(k2, g.threads, chan_p2_lds_bytes(M, g.TF)) * chan_pct / 100));
            CSDR_LAUNCH(c, LANE_POST, KID_CHAN_ANALYZE, k2, dim3(wgs), dim3(g.threads), chan_p2_lds_bytes(M, g.TF), x, hist, hist_new, p->taps.p,
                        p->twA.p, p->twM.p, p->active.p, g, n_frames, out, p->chan_stride, fused_ends ? p->tile_end.p : (d2 *)nullptr, p->dc_c);
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
    CSDR_HIP_TRY(hipMemcpyAsync(host_out, post_buf(p, p->cur) + (int64_t)ch * p->chan_stride, (size_t)cnt * sizeof(float2), hipMemcpyDeviceToHost, st));
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
__global__ __launch_bounds__(256) void rows_copy(const float2 *__restrict__ src, int64_t src_stride, const int *__restrict__ src_rows,
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
        CSDR_HIP_TRY(hipMemcpy(d, channels, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
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
    if (int rc = post_row_list(p, channels, n, &rows)) return rc;
    CSDR_LAUNCH(p->ctx, LANE_POST, KID_ROWS_COPY, rows_copy, dim3((unsigned)std::min<int64_t>(64, (nf + 255) / 256), n), dim3(256), 0,
                post_buf(p, p->cur), p->chan_stride, rows, (float2 *)dst_dev, dst_stride, (const int *)nullptr, nf);
    CSDR_HIP_TRY(hipGetLastError());
    return CSDR_OK;
}
extern "C" int csdr_post_import_begin(csdr_post *p, int n_blocks, int block_len, int64_t frequency) {
    DeviceScope dev__(p ? p->ctx : nullptr);
    if (!p || !p->configured || p->mode == CSDR_POST_SINGLE) return fail(CSDR_ESTATE, "post is not a configured channelizer");
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

// =================================================================================================== demodulator bank
namespace {
struct SlotHost {
    bool configured = false, active = false;
    csdr_demod_params prm{};
    design::MsresampPlan iq, au;
    int64_t chan_rate = 0;
    // integer state mirrored on the host (closed-form bookkeeping)
    uint32_t theta = 0, dtheta = 0, buf_idx = 0, phase = 0, aphase = 0, abuf = 0, ssb_theta = 0, cw_dtheta = 0;
    long long shift_frequency = 0;
    bool shift_valid = false;
    int hist_parity = 0, last_parity = 0;
    int prev_J = 0;                          // resampled-IQ samples of the previous executed batch
    int warm = 0;                            // cascade span in input samples (+ one output period)
    bool fms_sos_set = false;                // csdr_bank_set_fms_pilot: caller-supplied pilot band-pass sections
    float fms_b[15] = {0}, fms_a[15] = {0};
    void *slab = nullptr;
    SlotCfg cfg{};
    // results of the last execute
    std::vector<csdr_block_result> results;
    int last_J = 0, last_A = 0;
};
constexpr int kStageRing = 4;                // pinned staging sets for the per-batch uploads
}  // namespace

struct csdr_bank {
    csdr_ctx *ctx = nullptr;
    int max_demods = 0, max_blocks = 0;
    std::vector<SlotHost> slots;
    DevBuf<SlotCfg> cfgs;
    // per-batch device tables, two copies: the front-end of batch i+1 uploads its set while the audio kernels of batch i
    // still read theirs
    DevBuf<SlotDyn> dyns;                    // [2][max_demods]
    DevBuf<int> slot_list;                   // [2][3][max_demods]: all running slots | running auto-gain slots | grouped by front-end kernel
    DevBuf<BlockPlan> plans;                 // [2][max_demods][max_blocks + 1]
    uint64_t seq = 0;
    hipEvent_t ev_fe_done[2] = {nullptr, nullptr}, ev_audio_done[2] = {nullptr, nullptr};
    bool audio_pending[2] = {false, false};
    DevBuf<float> arms;
    DevBuf<ModemConsts> mconsts;
    PinBuf<SlotDyn> dyns_h[kStageRing];
    PinBuf<int> slot_list_h[kStageRing];
    PinBuf<BlockPlan> plans_h[kStageRing];
    hipEvent_t stage_ev[kStageRing] = {nullptr, nullptr, nullptr, nullptr};
    bool stage_used[kStageRing] = {false, false, false, false};
    int stage_next = 0;
    PinBuf<BlockOut> bout_h;
    std::map<uint32_t, int> arm_index;       // key: bit pattern of rate_arb
    std::vector<float> arms_host;
    int n_run = 0, last_nb = 0;
    size_t lds_attr[7] = {0, 0, 0, 0, 0, 0, 0};
    DevBuf<int16_t> pcm;                     // csdr_bank_fetch_pcm16: the converted audio of one slot
    DevBuf<PcmJob> pcm_jobs;
};

static int bank_arm_bank(csdr_bank *b, const design::MsresampPlan &p, int *idx) {
    uint32_t key;
    memcpy(&key, &p.rate_arb, 4);
    auto it = b->arm_index.find(key);
    if (it != b->arm_index.end()) { *idx = it->second; return CSDR_OK; }
    const int i = (int)b->arm_index.size();
    b->arms_host.insert(b->arms_host.end(), p.arms.begin(), p.arms.end());
    const size_t need = b->arms_host.size();
    if (need > b->arms.cap) {
        // grow: re-upload everything (cold path)
        if (int rc = b->ctx->sync_all()) return rc;
        if (int rc = b->arms.reserve(std::max(need, b->arms.cap * 2 + (size_t)kArms * kArmTaps * 8))) return rc;
        CSDR_HIP_TRY(hipMemcpy(b->arms.p, b->arms_host.data(), need * sizeof(float), hipMemcpyHostToDevice));
    } else {
        CSDR_HIP_TRY(hipMemcpy(b->arms.p + (size_t)i * kArms * kArmTaps, p.arms.data(), p.arms.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    b->arm_index[key] = i;
    *idx = i;
    return CSDR_OK;
}

static void fill_resamp_cfg(ResampCfg &rc, const design::MsresampPlan &p, int arms_idx) {
    memset(&rc, 0, sizeof rc);
    rc.interp = p.interp ? 1 : 0;
    rc.S = (int)p.S;
    rc.step = p.step;
    rc.arms_idx = arms_idx;
    for (unsigned e = 0; e < p.S; ++e) {
        // execution order: decimator runs design index S-1 first; interpolator runs design index 0 first
        const unsigned g = p.interp ? e : (p.S - 1 - e);
        rc.m_x[e] = (int)p.m[g];
        for (unsigned j = 0; j < p.m[g]; ++j) rc.h_x[e][j] = p.h1[g][j];
    }
}

extern "C" int csdr_bank_create(csdr_ctx *ctx, int max_demods, int max_blocks, csdr_bank **out) {
    DeviceScope dev__(ctx);
    if (!ctx || !out || max_demods <= 0 || max_blocks <= 0) return fail(CSDR_EINVAL, "bad argument");
    std::unique_ptr<csdr_bank> b(new csdr_bank());
    b->ctx = ctx; b->max_demods = max_demods; b->max_blocks = max_blocks;
    b->slots.resize(max_demods);
    if (int rc = b->cfgs.reserve(max_demods)) return rc;
    if (int rc = b->dyns.reserve(2 * (size_t)max_demods)) return rc;
    if (int rc = b->slot_list.reserve(2 * 3 * (size_t)max_demods)) return rc;
    if (int rc = b->plans.reserve(2 * (size_t)max_demods * (max_blocks + 1))) return rc;
    for (int k = 0; k < 2; ++k) {
        CSDR_HIP_TRY(hipEventCreateWithFlags(&b->ev_fe_done[k], hipEventDisableTiming));
        CSDR_HIP_TRY(hipEventCreateWithFlags(&b->ev_audio_done[k], hipEventDisableTiming));
    }
    if (int rc = b->mconsts.reserve(1)) return rc;
    for (int r = 0; r < kStageRing; ++r) {
        if (int rc = b->dyns_h[r].reserve(max_demods)) return rc;
        if (int rc = b->slot_list_h[r].reserve(3 * (size_t)max_demods)) return rc;
        if (int rc = b->plans_h[r].reserve((size_t)max_demods * (max_blocks + 1))) return rc;
        CSDR_HIP_TRY(hipEventCreate(&b->stage_ev[r]));
    }
    if (int rc = b->bout_h.reserve(max_blocks)) return rc;
    CSDR_HIP_TRY(hipMemset(b->cfgs.p, 0, max_demods * sizeof(SlotCfg)));
    // modem constants (cold): AM notch ModemAM.cpp:9, SSB filters ModemUSB.cpp:8-11
    ModemConsts mc;
    memset(&mc, 0, sizeof mc);
    std::vector<float> am = design::dc_notch_taps(25, 30.0f);
    for (int i = 0; i < kAmTaps; i++) mc.am_taps[i] = am[i];
    std::vector<design::Sos> sos = design::butter_lowpass_sos(6, 0.25f);
    for (int q = 0; q < 3; q++) for (int i = 0; i < 3; i++) { mc.sos_b[q][i] = sos[q].b[i]; mc.sos_a[q][i] = sos[q].a[i]; }
    std::vector<float> hq = design::hilbert_taps(kHilbM, 90.0f);
    for (int i = 0; i < 2 * kHilbM; i++) mc.hilb[i] = hq[i];
    std::vector<float> hq60 = design::hilbert_taps(kHilbM, 60.0f);           // ModemCW.cpp:23
    for (int i = 0; i < 2 * kHilbM; i++) mc.hilb60[i] = hq60[i];
    std::vector<float> g = design::sos_impulse_response(sos, kSsbFir);
    for (int i = 0; i < kSsbFir; i++) mc.ssb_fir[i] = g[i];
    CSDR_HIP_TRY(hipMemcpy(b->mconsts.p, &mc, sizeof mc, hipMemcpyHostToDevice));
    *out = b.release();
    return CSDR_OK;
}

extern "C" void csdr_bank_destroy(csdr_bank *b) {
    DeviceScope dev__(b ? b->ctx : nullptr);
    if (!b) return;
    (void)b->ctx->sync_all();
    for (int k = 0; k < 2; ++k) {
        if (b->ev_fe_done[k]) (void)hipEventDestroy(b->ev_fe_done[k]);
        if (b->ev_audio_done[k]) (void)hipEventDestroy(b->ev_audio_done[k]);
    }
    for (auto &s : b->slots) if (s.slab) (void)hipFree(s.slab);
    b->cfgs.release(); b->dyns.release(); b->slot_list.release(); b->plans.release(); b->arms.release(); b->mconsts.release();
    for (int r = 0; r < kStageRing; ++r) {
        b->dyns_h[r].release(); b->slot_list_h[r].release(); b->plans_h[r].release();
        if (b->stage_ev[r]) (void)hipEventDestroy(b->stage_ev[r]);
    }
    b->bout_h.release();
    b->pcm.release(); b->pcm_jobs.release();
    delete b;
}

// internal: NCO + msresamp only, no modem / audio stage (the zoomed spectrum view's shift + resample, SpectrumVisualProcessor.cpp:306-379)
#define CSDR_MODEM_FRONTEND_ONLY 100
// no device modem / audio stage: the internal front-end-only slot, or a host plug-in modem (CSDR_MODEM_HOST) that demodulates the fetched IQ
static inline bool is_fe_only(int modem) { return modem == CSDR_MODEM_FRONTEND_ONLY || modem == CSDR_MODEM_HOST; }
static int modem_check_rate(int modem, int bw, int audio_rate) {   // Modem*::checkSampleRate (ModemAnalog.cpp:14-19, ModemUSB.cpp:29-37, ModemIQ.cpp:31-33)
    if (modem == CSDR_MODEM_HOST) return bw;                       // the plug-in's own checkSampleRate ran on the host
    if (modem == CSDR_MODEM_IQ || modem == CSDR_MODEM_FRONTEND_ONLY) return audio_rate;
    if (modem == CSDR_MODEM_FMS) return bw < 100000 ? 100000 : bw;      // ModemFMStereo.cpp:27-35
    if (bw < 500) bw = 500;                          // MIN_BANDWIDTH, Modem.h:13
    if ((modem == CSDR_MODEM_USB || modem == CSDR_MODEM_LSB) && (bw % 2)) bw += 1;
    return bw;
}

static int bank_configure_slot(csdr_bank *b, int slot, const csdr_demod_params *prm, const csdr_post *post);
extern "C" int csdr_bank_configure_slot(csdr_bank *b, int slot, const csdr_demod_params *prm, const csdr_post *post) {
    DeviceScope dev__(b ? b->ctx : nullptr);
    if (prm && (prm->modem < CSDR_MODEM_NBFM || prm->modem > CSDR_MODEM_HOST)) return fail(CSDR_EUNSUPPORTED, "modem %d", prm->modem);
    return bank_configure_slot(b, slot, prm, post);
}
static int bank_configure_slot(csdr_bank *b, int slot, const csdr_demod_params *prm, const csdr_post *post) {
    if (!b || !prm || !post) return fail(CSDR_EINVAL, "null argument");
    if (slot < 0 || slot >= b->max_demods) return fail(CSDR_EINVAL, "slot out of range");
    if (!post->configured) return fail(CSDR_ESTATE, "post not configured");
    if (prm->bandwidth <= 0 || prm->audio_sample_rate <= 0) return fail(CSDR_EINVAL, "bad rates");
    SlotHost &s = b->slots[slot];
    if (int rc = b->ctx->sync_all()) return rc;
    s.configured = false;
    s.prm = *prm;
    s.prm.bandwidth = modem_check_rate(prm->modem, prm->bandwidth, prm->audio_sample_rate);
    s.chan_rate = csdr_post_channel_rate(post);
    const double iq_ratio = (double)s.prm.bandwidth / (double)s.chan_rate;        // DemodulatorWorkerThread.cpp:99-100
    s.iq = design::plan_msresamp((float)iq_ratio, 60.0f);        // bandwidth above the channel rate: the interpolating form (:97-101 creates it for any ratio)
    const double au_ratio = is_fe_only(s.prm.modem) ? 1.0 : double(s.prm.audio_sample_rate) / double(s.prm.bandwidth);   // ModemAnalog.cpp:29-30
    s.au = design::plan_msresamp((float)au_ratio, 60.0f);
    if (s.iq.S > kMaxHb || s.au.S > kMaxHb) return fail(CSDR_EUNSUPPORTED, "resampling ratio needs %u half-band stages", s.iq.S);
    if (!s.au.interp) {      // decimating audio resampler: its cascade must fit the carried demodulator-output history
        int64_t lo = -(int64_t)(kArmTaps - 1);
        for (int e = (int)s.au.S - 1; e >= 0; --e) lo = 2 * lo - (4 * (int)s.au.m[s.au.S - 1 - e] - 2);
        if (-lo + (1 << s.au.S) > kDHist) return fail(CSDR_EUNSUPPORTED, "audio decimation by %d / %d needs %lld samples of history", s.prm.bandwidth, s.prm.audio_sample_rate, (long long)-lo);
    }
    const bool fms = s.prm.modem == CSDR_MODEM_FMS;
    std::vector<float> fms_fir;
    if (fms) {
        // csdr_demod_params::modem_arg = the "demph" setting (ModemFMStereo.cpp:42-81): microseconds, 0 -> the default 75, < 0 -> none
        const int demph = s.prm.modem_arg == 0 ? 75 : (s.prm.modem_arg < 0 ? 0 : s.prm.modem_arg);
        fms_fir = design::fms_output_fir(s.prm.audio_sample_rate, demph, kFmsFirMax);
        if (fms_fir.empty()) return fail(CSDR_EUNSUPPORTED, "FM stereo output filter at %d Hz exceeds %d taps", s.prm.audio_sample_rate, kFmsFirMax);
    }
    int ia = 0, aa = 0;
    if (int rc = bank_arm_bank(b, s.iq, &ia)) return rc;
    if (int rc = bank_arm_bank(b, s.au, &aa)) return rc;
    // cascade span: input samples before an output that can influence it (front-end warm-up, carried history)
    if (s.iq.interp) {
        // interpolating: the first outputs of a batch reach back (arm length + the half-band windows, in input samples)
        int64_t lo = 0;
        for (int st = (int)s.iq.S - 1; st >= 0; --st) lo = (lo >> 1) - (2 * (int)s.iq.m[st] - 1);
        s.warm = (int)(((-lo) * (int64_t)s.iq.step) >> 24) + kArmTaps + 8;
    } else {
        const int S = (int)s.iq.S;
        int64_t lo = -(int64_t)(kArmTaps - 1);
        for (int e = S - 1; e >= 0; --e) lo = 2 * lo - (4 * (int)s.iq.m[S - 1 - e] - 2);
        s.warm = (int)(-lo) + (2 << S);
    }
    const int hist_len = (s.warm + 63) & ~63;
    if (hist_len > kMixHist) return fail(CSDR_EUNSUPPORTED, "cascade span %d exceeds the carried history", s.warm);
    // capacities for one execute
    const int64_t max_bc = post->max_block_len / post->hop;
    const int64_t cap_iq = (int64_t)std::ceil((double)b->max_blocks * (double)max_bc * iq_ratio) + b->max_blocks + 64;
    const int64_t cap_audio = is_fe_only(s.prm.modem) ? 64 : s.prm.modem == CSDR_MODEM_IQ ? 2 * cap_iq + 64      // two floats per IQ sample, no audio resampler
        : (fms ? 2 : 1) * ((int64_t)std::ceil((double)cap_iq * au_ratio) + (int64_t)b->max_blocks * (2 << (s.au.interp ? s.au.S : 0)) + 64);
    // one slab per slot
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_mix = carve((size_t)2 * hist_len * sizeof(float2));
    const size_t o_iq = carve(2 * (kIqHist + cap_iq) * sizeof(float2));
    const size_t o_d = carve(cap_iq * sizeof(float));
    const size_t o_dh = carve(2 * kDHist * sizeof(float));
    const size_t o_au = carve(cap_audio * sizeof(float));
    const size_t o_agc = carve(8 * sizeof(float));
    const size_t o_pll = carve(2 * sizeof(uint32_t));
    const size_t o_bm = carve(b->max_blocks * sizeof(float)), o_bma = carve(b->max_blocks * sizeof(float));
    const size_t o_bo = carve(b->max_blocks * sizeof(BlockOut));
    const size_t o_sc = carve(kScopeMax * sizeof(float)), o_scn = carve(sizeof(int32_t));
    size_t o_fx = 0, o_fth = 0, o_fm = 0, o_fs = 0, o_fyh = 0, o_fuh = 0, o_fst = 0, o_ffir = 0;
    if (fms) {
        o_fx = carve(cap_iq * sizeof(float2)); o_fth = carve(cap_iq * sizeof(uint32_t));
        o_fm = carve((cap_audio / 2) * sizeof(float)); o_fs = carve((cap_audio / 2) * sizeof(float));
        o_fyh = carve(2 * kFmsYHist * sizeof(float2)); o_fuh = carve((size_t)4 * kFmsFirMax * sizeof(float));
        o_fst = carve(kFmsStateWords * sizeof(float)); o_ffir = carve(kFmsFirMax * sizeof(float));
    }
    if (s.slab) { (void)hipFree(s.slab); s.slab = nullptr; }
    if (hipMalloc(&s.slab, off) != hipSuccess) return fail(CSDR_ENOMEM, "slot slab of %zu bytes", off);
    CSDR_HIP_TRY(hipMemset(s.slab, 0, off));
    char *base = (char *)s.slab;
    SlotCfg &c = s.cfg;
    memset(&c, 0, sizeof c);
    fill_resamp_cfg(c.rs_iq, s.iq, ia);
    fill_resamp_cfg(c.rs_au, s.au, aa);
    c.modem = s.prm.modem;
    c.hist_len = hist_len;
    c.mixhist = (float2 *)(base + o_mix); c.iq = (float2 *)(base + o_iq); c.d = (float *)(base + o_d); c.dh = (float *)(base + o_dh);
    c.audio = (float *)(base + o_au); c.agc = (float *)(base + o_agc); c.pll = (uint32_t *)(base + o_pll);   // slab is zeroed: nco_crcf_reset
    c.blockmax = (float *)(base + o_bm); c.blockmaa = (float *)(base + o_bma); c.bout = (BlockOut *)(base + o_bo);
    c.scope = (float *)(base + o_sc); c.scope_n = (int32_t *)(base + o_scn);
    c.cap_iq = (int)cap_iq; c.cap_audio = (int)cap_audio;
    if (fms) {
        c.fms_x = (float2 *)(base + o_fx); c.fms_theta = (uint32_t *)(base + o_fth); c.fms_m = (float *)(base + o_fm); c.fms_s = (float *)(base + o_fs);
        c.fms_yh = (float2 *)(base + o_fyh); c.fms_uh = (float *)(base + o_fuh); c.fms_state = (float *)(base + o_fst); c.fms_fir = (float *)(base + o_ffir);
        c.fms_fir_len = (int)fms_fir.size();
        CSDR_HIP_TRY(hipMemcpy(c.fms_fir, fms_fir.data(), fms_fir.size() * sizeof(float), hipMemcpyHostToDevice));
        const std::vector<design::Sos> sos = design::fms_pilot_sos(s.prm.bandwidth);
        for (int q = 0; q < 5; ++q) for (int k = 0; k < 3; ++k) { c.fms_b[3 * q + k] = sos[q].b[k]; c.fms_a[3 * q + k] = sos[q].a[k]; }
        if (s.fms_sos_set) { memcpy(c.fms_b, s.fms_b, sizeof c.fms_b); memcpy(c.fms_a, s.fms_a, sizeof c.fms_a); }
    }
    const float agc0[8] = {1.0f, 1.0f, 1.0f, 0.f, 1.0f, 1.0f, 1.0f, 0.f};   // ModemAnalog::ModemAnalog(): aOutputCeil(1), MA(1), MAA(1)
    CSDR_HIP_TRY(hipMemcpy(c.agc, agc0, sizeof agc0, hipMemcpyHostToDevice));
    CSDR_HIP_TRY(hipMemcpy(b->cfgs.p + slot, &c, sizeof c, hipMemcpyHostToDevice));
    // fresh objects: nco_crcf_create / msresamp create / modem ctor all start from zero state
    s.theta = 0; s.dtheta = 0; s.buf_idx = 0; s.phase = 0; s.aphase = 0; s.abuf = 0; s.hist_parity = 0; s.last_parity = 0; s.prev_J = 0;
    s.shift_valid = false; s.shift_frequency = 0;
    // ModemUSB/LSB ctor: nco_crcf_set_frequency(ssbShift, 2 pi 0.25) -> the oscillator advances 2^30 per sample
    s.ssb_theta = 0;
    // ModemCW: mLO runs at the audio rate, nco_crcf_set_frequency(mLO, 2 pi mBeepFrequency / audioSampleRate) every block (:171)
    s.cw_dtheta = s.prm.modem == CSDR_MODEM_CW ? design::nco_phase_word(2.0f * (float)M_PI * 650.0f / (float)s.prm.audio_sample_rate) : 0;
    s.configured = true; s.active = true;
    s.results.clear(); s.last_J = 0; s.last_A = 0;
    return CSDR_OK;
}

extern "C" int csdr_bank_set_frequency(csdr_bank *b, int slot, int64_t f) {
    if (!b || slot < 0 || slot >= b->max_demods || !b->slots[slot].configured) return fail(CSDR_EINVAL, "bad slot");
    b->slots[slot].prm.frequency = f;
    return CSDR_OK;
}
extern "C" int csdr_bank_set_active(csdr_bank *b, int slot, int active) {
    if (!b || slot < 0 || slot >= b->max_demods || !b->slots[slot].configured) return fail(CSDR_EINVAL, "bad slot");
    b->slots[slot].active = active != 0;
    return CSDR_OK;
}

static inline int64_t first_out(int64_t K, uint32_t phase0, uint32_t step) {
    const int64_t lim = K * (int64_t)(1 << 24) - (int64_t)phase0;
    if (lim <= 0) return -((-lim) / (int64_t)step);
    return (lim + step - 1) / step;
}

extern "C" int csdr_bank_execute(csdr_bank *b, const csdr_post *post) {
    DeviceScope dev__(b ? b->ctx : nullptr);
    if (!b || !post) return fail(CSDR_EINVAL, "null argument");
    if (!post->configured || post->n_blocks <= 0) return fail(CSDR_ESTATE, "post has no data");
    csdr_ctx *c = b->ctx;
    hipStream_t st = c->lanes[LANE_FE], st_a = c->lanes[LANE_AUDIO];
    const int NB = post->n_blocks, M = post->M, Bc = post->block_len / post->hop;
    if (NB > b->max_blocks) return fail(CSDR_ERANGE, "batch of %d blocks exceeds bank capacity %d", NB, b->max_blocks);
    const int bpar = (int)(b->seq & 1);      // which copy of the per-batch device tables this batch uses
    SlotDyn *dyns_d = b->dyns.p + (size_t)bpar * b->max_demods;
    int *lists_d = b->slot_list.p + (size_t)bpar * 3 * b->max_demods;
    BlockPlan *plans_d = b->plans.p + (size_t)bpar * b->max_demods * (b->max_blocks + 1);
    const int64_t rate = csdr_post_channel_rate(post);
    // pinned staging set for this batch: wait only for the upload that last used it (kStageRing batches ago)
    const int ring = b->stage_next;
    b->stage_next = (b->stage_next + 1) % kStageRing;
    if (b->stage_used[ring]) CSDR_HIP_TRY(hipEventSynchronize(b->stage_ev[ring]));
    SlotDyn *dyns_h = b->dyns_h[ring].p;
    int *slot_list_h = b->slot_list_h[ring].p;
    BlockPlan *plans_h = b->plans_h[ring].p;
    int n_run = 0, n_ag = 0, max_n_iq = 0, max_n_iq_ag = 0, max_n_audio = 0, warm_max = 0, max_aS = 0, max_cw_audio = 0;
    int max_n_iq_fms = 0, max_n_au_fms = 0;
    std::vector<int> fms_slots;              // FM-stereo slots of this batch (their list shares the auto-gain list's region, from its end)
    int *ag_list_h = slot_list_h + b->max_demods;
    // The per-slot walk below validates AND advances the host-side integer state (oscillator phases, resampler phases, buffer
    // parities).  A rejected batch must leave every slot as it was -- no kernel runs for it -- so the state is snapshotted and
    // put back on any error return of the walk.
    struct Snap { uint32_t theta, dtheta, buf_idx, phase, aphase, abuf, ssb_theta; long long shift_frequency; bool shift_valid; int hist_parity, last_parity, prev_J; };
    std::vector<Snap> snap((size_t)b->max_demods);
    for (int si = 0; si < b->max_demods; ++si) {
        const SlotHost &s = b->slots[si];
        snap[si] = Snap{s.theta, s.dtheta, s.buf_idx, s.phase, s.aphase, s.abuf, s.ssb_theta, s.shift_frequency, s.shift_valid, s.hist_parity, s.last_parity, s.prev_J};
    }
    const int ring_before = ring;
    auto reject = [&](int rc) {
        for (int si = 0; si < b->max_demods; ++si) {
            SlotHost &s = b->slots[si];
            const Snap &q = snap[si];
            s.theta = q.theta; s.dtheta = q.dtheta; s.buf_idx = q.buf_idx; s.phase = q.phase; s.aphase = q.aphase; s.abuf = q.abuf; s.ssb_theta = q.ssb_theta;
            s.shift_frequency = q.shift_frequency; s.shift_valid = q.shift_valid; s.hist_parity = q.hist_parity; s.last_parity = q.last_parity; s.prev_J = q.prev_J;
            s.results.clear(); s.last_J = 0; s.last_A = 0;
        }
        b->stage_next = ring_before;
        return rc;
    };
    for (int si = 0; si < b->max_demods; ++si) {
        SlotHost &s = b->slots[si];
        s.results.clear(); s.last_J = 0; s.last_A = 0;
        if (!s.configured || !s.active) continue;
        if (s.chan_rate != rate) return reject(fail(CSDR_ESTATE, "slot %d was built for channel rate %lld, post now runs %lld: reconfigure", si, (long long)s.chan_rate, (long long)rate));
        // channel routing: runDemodChannels, SDRPostThread.cpp:317-323 (nearest centre; M == wrap channel = M/2)
        int ch = csdr_post_channel_at(post, s.prm.frequency);
        if (ch < 0) continue;
        const int64_t centre = (M == 1) ? post->frequency : post->centers[ch];
        const int data_ch = (M > 1 && ch == M) ? M / 2 : ch;
        if (M > 1 && !std::binary_search(post->active_host.begin(), post->active_host.end(), data_ch))
            return reject(fail(CSDR_ESTATE, "slot %d needs channel %d which the channelizer was told not to produce", si, data_ch));
        // DemodulatorPreThread.cpp:154-165
        const long long shift = (long long)s.prm.frequency - (long long)centre;
        const int bound = (int)((double)(rate / 2) * 1.5);
        if (!s.shift_valid || shift != s.shift_frequency) {
            s.shift_frequency = shift; s.shift_valid = true;
            if (std::llabs(shift) <= bound)
                s.dtheta = design::nco_phase_word((float)((2.0 * M_PI) * (((double)std::llabs(shift)) / ((double)rate))));
        }
        const bool skipped = std::llabs(shift) > bound;
        s.results.resize(NB);
        if (skipped) {
            for (auto &r : s.results) { memset(&r, 0, sizeof r); r.skipped = 1; r.nco_theta = s.theta; r.resamp_phase = s.phase; r.buffer_index = s.buf_idx; }
            continue;
        }
        SlotDyn &d = dyns_h[si];
        memset(&d, 0, sizeof d);
        d.active = 1; d.chan = data_ch; d.theta0 = s.theta; d.dtheta = s.dtheta;
        d.mixdir = shift == 0 ? 0 : (shift < 0 ? +1 : -1);          // :186-191: shift < 0 -> mix up
        d.buf0 = s.buf_idx; d.phase0 = s.phase; d.aphase0 = s.aphase; d.abuf0 = s.abuf; d.ssb_theta0 = s.ssb_theta; d.cw_dtheta = s.cw_dtheta; d.hist_parity = s.hist_parity;
        d.prev_j = s.prev_J;
        // per-block plan
        BlockPlan *pl = plans_h + (size_t)si * (NB + 1);
        const int S = (int)s.iq.S, aS = (int)s.au.S;
        const bool fe_only = is_fe_only(s.prm.modem);
        const bool iq_modem = s.prm.modem == CSDR_MODEM_IQ || fe_only;      // no audio resampler: 2 floats per resampled IQ sample
        const bool au_interp = s.au.interp;
        const bool fms = s.prm.modem == CSDR_MODEM_FMS;          // two floats (left, right) per audio sample
        // floats written per arbitrary-stage output = 2^ash: an interpolating audio resampler fans every arbitrary-stage output out to
        // 2^aS samples; I/Q and FM stereo write two floats per sample (FM stereo with either kind of resampler, ModemFMStereo.cpp:91-105)
        const int ash = iq_modem ? 1 : (au_interp ? aS : 0) + (fms ? 1 : 0);
        const bool iq_interp = s.iq.interp;      // arbitrary stage first: it consumes the channel samples directly, each output fans out to 2^S
        for (int bb = 0; bb <= NB; ++bb) {
            const int64_t K = iq_interp ? (int64_t)bb * Bc : (((int64_t)s.buf_idx + (int64_t)bb * Bc) >> S);
            const int64_t J = iq_interp ? (first_out(K, s.phase, s.iq.step) << S) : first_out(K, s.phase, s.iq.step);
            // audio msresamp_rrrf (ModemAnalog.cpp:88): interpolating = arbitrary stage first (input index J);
            // decimating = half-band /2 stages first: the arbitrary stage sees (abuf + J) >> aS chain outputs
            const int64_t Ka = au_interp ? J : (((int64_t)s.abuf + J) >> aS);
            const int64_t Q = iq_modem ? J : first_out(Ka, s.aphase, s.au.step);
            pl[bb].j0 = (int)J; pl[bb].q0 = (int)Q;
        }
        const int64_t Jtot = pl[NB].j0, Qtot = pl[NB].q0;
        if (Jtot > s.cfg.cap_iq - 8 || (!fe_only && (Qtot << ash) > s.cfg.cap_audio - 8)) return reject(fail(CSDR_ERANGE, "slot %d output exceeds its buffers", si));
        int max_blk_audio = 0;
        for (int bb = 0; bb < NB; ++bb) {
            csdr_block_result &r = s.results[bb];
            memset(&r, 0, sizeof r);
            r.n_iq = pl[bb + 1].j0 - pl[bb].j0;
            r.n_audio = (int)(((int64_t)(pl[bb + 1].q0 - pl[bb].q0)) << ash);
            r.audio_offset = (int)(((int64_t)pl[bb].q0) << ash);
            if (fe_only) { r.n_audio = 0; r.audio_offset = 0; }
            else if (r.n_iq > kModemMaxBlockIq || r.n_audio > kAudioMaxOut) return reject(fail(CSDR_EUNSUPPORTED, "slot %d: %d IQ / %d audio samples per block exceed the per-workgroup limits", si, r.n_iq, r.n_audio));
            if (!fe_only) { max_n_iq = std::max(max_n_iq, r.n_iq); max_n_audio = std::max(max_n_audio, fms ? r.n_audio / 2 : r.n_audio); }
            if (fms) { max_n_iq_fms = std::max(max_n_iq_fms, r.n_iq); max_n_au_fms = std::max(max_n_au_fms, r.n_audio / 2); }
            max_blk_audio = std::max(max_blk_audio, r.n_audio);
            const int64_t Kb = iq_interp ? (int64_t)(bb + 1) * Bc : (((int64_t)s.buf_idx + (int64_t)(bb + 1) * Bc) >> S);
            r.buffer_index = iq_interp ? 0u : (uint32_t)(((int64_t)s.buf_idx + (int64_t)(bb + 1) * Bc) & ((1 << S) - 1));
            r.resamp_phase = (uint32_t)((int64_t)s.phase + (int64_t)(iq_interp ? pl[bb + 1].j0 >> S : pl[bb + 1].j0) * s.iq.step - (Kb << 24));
            r.nco_theta = d.mixdir ? (uint32_t)(s.theta + (uint32_t)((int64_t)(bb + 1) * Bc) * s.dtheta) : s.theta;
        }
        // advance host-side integer state
        const int64_t Ktot = iq_interp ? (int64_t)NB * Bc : (((int64_t)s.buf_idx + (int64_t)NB * Bc) >> S);
        s.phase = (uint32_t)((int64_t)s.phase + (iq_interp ? Jtot >> S : Jtot) * (int64_t)s.iq.step - (Ktot << 24));
        if (!iq_interp) s.buf_idx = (uint32_t)(((int64_t)s.buf_idx + (int64_t)NB * Bc) & ((1 << S) - 1));
        if (d.mixdir) s.theta += (uint32_t)((int64_t)NB * Bc) * s.dtheta;
        if (!iq_modem) {
            const int64_t Ka_tot = au_interp ? Jtot : (((int64_t)s.abuf + Jtot) >> aS);
            s.aphase = (uint32_t)((int64_t)s.aphase + Qtot * (int64_t)s.au.step - (Ka_tot << 24));
            if (!au_interp) s.abuf = (uint32_t)(((int64_t)s.abuf + Jtot) & ((1 << aS) - 1));
        }
        if (s.prm.modem == CSDR_MODEM_CW) { s.ssb_theta += (uint32_t)(Qtot << ash) * s.cw_dtheta; max_cw_audio = std::max(max_cw_audio, max_blk_audio); }
        else s.ssb_theta += (uint32_t)Jtot * (1u << 30);
        s.last_parity = s.hist_parity;
        s.hist_parity ^= 1;
        s.last_J = (int)Jtot; s.last_A = fe_only ? 0 : (int)(Qtot << ash);
        s.prev_J = (int)Jtot;
        warm_max = std::max(warm_max, s.warm); max_aS = std::max(max_aS, aS);
        if (fms) fms_slots.push_back(si);
        else if (s.prm.modem != CSDR_MODEM_NBFM && s.prm.modem != CSDR_MODEM_FM && s.prm.modem != CSDR_MODEM_IQ && !fe_only) {
            ag_list_h[n_ag++] = si;
            for (int bb = 0; bb < NB; ++bb) max_n_iq_ag = std::max(max_n_iq_ag, s.results[bb].n_iq);      // what the modem kernel stages per block
        }
        slot_list_h[n_run++] = si;
    }
    b->n_run = n_run; b->last_nb = NB;
    if (n_run == 0) return CSDR_OK;
    const int n_fms = (int)fms_slots.size(), fms_off = b->max_demods - n_fms;     // n_ag + n_fms <= n_run <= max_demods
    for (int i = 0; i < n_fms; ++i) ag_list_h[fms_off + i] = fms_slots[i];
    // running slots grouped by front-end kernel (filled before the staging set is handed to the copy engine)
    int *grp_h = slot_list_h + 2 * (size_t)b->max_demods;
    int grp_off[8] = {0}, grp_n[8] = {0};        // index 0: generic, 3..6: specialised by S
    {
        auto klass = [&](const SlotHost &s) {
            const int S = (int)s.iq.S;
            if (s.iq.interp) return 7;
            if (S < 3 || S > 6) return 0;
            for (int e = 0; e < S; ++e) if ((int)s.iq.m[S - 1 - e] != fes_m(S, e)) return 0;
            return S;
        };
        int pos = 0;
        for (int k = 0; k < 8; ++k) {
            grp_off[k] = pos;
            for (int i = 0; i < n_run; ++i) if (klass(b->slots[slot_list_h[i]]) == k) grp_h[pos++] = slot_list_h[i];
            grp_n[k] = pos - grp_off[k];
        }
    }
    // the audio stage runs the slots that have one: compact the head of the list (the front-end groups above are copies)
    int n_audio_run = 0;
    for (int i = 0; i < n_run; ++i) if (!is_fe_only(b->slots[slot_list_h[i]].prm.modem)) slot_list_h[n_audio_run++] = slot_list_h[i];
    // lane FE: the channelizer output of this batch must be complete; the tables and the resampled-IQ buffers of this
    // parity were last read by the audio kernels two batches ago
    const int pk = post->cur;
    if (post->ctx != c || !c->same(LANE_POST, LANE_FE)) {
        if (post->ctx != c) CSDR_HIP_TRY(hipEventRecord(post->ev_ready[pk], post->ctx->lanes[LANE_POST]));
        CSDR_HIP_TRY(hipStreamWaitEvent(st, post->ev_ready[pk], 0));
    }
    if (b->audio_pending[bpar]) if (int rc = c->wait(b->ev_audio_done[bpar], LANE_AUDIO, LANE_FE)) return rc;
    CSDR_HIP_TRY(hipMemcpyAsync(dyns_d, dyns_h, b->max_demods * sizeof(SlotDyn), hipMemcpyHostToDevice, st));
    CSDR_HIP_TRY(hipMemcpyAsync(lists_d, slot_list_h, 3 * (size_t)b->max_demods * sizeof(int), hipMemcpyHostToDevice, st));
    CSDR_HIP_TRY(hipMemcpyAsync(plans_d, plans_h, (size_t)b->max_demods * (NB + 1) * sizeof(BlockPlan), hipMemcpyHostToDevice, st));
    CSDR_HIP_TRY(hipEventRecord(b->stage_ev[ring], st));
    b->stage_used[ring] = true;
    // front-end geometry: every slot's batch is cut into P ranges; a range re-runs `warm` inputs in front of it.
    // Slots whose cascade has the reference's standard shape (m = 3..3, 5, 10; 3 <= S <= 6) run the specialised kernel,
    // one launch per depth S; anything else runs the generic one.
    const int64_t total = (int64_t)NB * Bc;
    // ranges per slot, PER LAUNCH (the slots are grouped by cascade depth, one launch per group on the same stream): as many as
    // make that launch's grid ONE round of resident workgroups (each range re-runs `warm` inputs, so fewer, longer ranges
    // waste less), but never shorter than 4 warm-up spans and never fewer than one
    static const int fe_pct = getenv("CSDR_FE_PCT") ? std::max(10, std::min(100, atoi(getenv("CSDR_FE_PCT")))) : 100;
    const int fe_slots = std::max(1, c->wg_slots(demod_frontend_s<5, 2048, true>, kFeThreads + 64, fes_lds_bytes<5, 2048>()) * fe_pct / 100);
    auto ranges_for = [&](int n_slots) {
        int P = (int)std::max<int64_t>(1, std::min<int64_t>(total / std::max<int64_t>(4096, 4 * (int64_t)warm_max), 4096));
        const int per_slot = fe_slots / std::max(1, n_slots) - 1;                          // one extra workgroup per slot carries the histories
        if (per_slot >= 1) P = std::min(P, per_slot);
        else {                                                           // more slots than resident workgroups: whole rounds
            const int rounds = (n_slots * 2 + fe_slots - 1) / fe_slots;
            P = std::max(1, std::min(P, rounds * fe_slots / std::max(1, n_slots) - 1));
        }
        return P;
    };
    size_t fe_lds = 0;
    for (int i = 0; i < grp_n[0]; ++i) fe_lds = std::max(fe_lds, fe_lds_bytes((int)b->slots[grp_h[grp_off[0] + i]].iq.S));
    const int cap_stream = (max_n_iq_ag + kSsbWarm + 64 + 3) & ~3;
    // CW blocks run the complex audio interpolator in LDS: IQ window + two stage arrays of (block audio + Hilbert reach)
    const int cap_cw = max_cw_audio ? ((max_cw_audio + 4 * kHilbM + 64 + 3) & ~3) : 0;
    // LDS of the modem kernel, sized by the modems that actually run (a DSB slot stages the sine table and one block of IQ, a CW
    // slot the complex interpolator's arrays; AM / SSB need four float streams): an oversized request costs resident waves
    bool any_dsb = false;
    for (int i = 0; i < n_ag; ++i) any_dsb = any_dsb || b->slots[ag_list_h[i]].prm.modem == CSDR_MODEM_DSB;
    const size_t dsb_lds = any_dsb ? 1024 * sizeof(float) + (size_t)(max_n_iq_ag + 64) * sizeof(float2) : 0;
    const size_t cw_lds = cap_cw ? ((size_t)kCwIqWin + 2 * (size_t)cap_cw) * sizeof(float2) : 0;
    const size_t modem_lds = std::max(std::max((size_t)4 * cap_stream * sizeof(float), cw_lds), dsb_lds) + 64;
    // LDS of the audio kernel: two ping-pong arrays (stage outputs) and the staged demodulator window (decimating
    // cascades reach back up to kDHist samples and their first stage outputs half the window)
    // samples in front of a block its audio cascade reaches back to (the backward range propagation of demod_audio_interp, taken
    // at A0 = 0): the staged window is the block's own samples plus this much history -- sized per configuration, not by the
    // largest history the slots could carry
    int hist_need = 2 * kArmTaps;
    for (int i = 0; i < n_run; ++i) {
        const SlotHost &s = b->slots[slot_list_h[i]];
        if (is_fe_only(s.prm.modem) || s.prm.modem == CSDR_MODEM_IQ) continue;
        const int aS = (int)s.au.S;
        int64_t lo = 0, need;
        if (s.au.interp) {
            for (int st = aS - 1; st >= 0; --st) lo = (lo >> 1) - (2 * (int)s.au.m[st] - 1);        // execution order = design order
            need = ((-lo * (int64_t)s.au.step) >> 24) + kArmTaps + 8;
        } else {
            lo = -(int64_t)(kArmTaps - 1);
            for (int e = aS - 1; e >= 0; --e) lo = 2 * lo - (4 * (int)s.au.m[aS - 1 - e] - 2);
            need = -lo + (1 << aS) + 8;
        }
        hist_need = std::max<int>(hist_need, (int)need);
    }
    const int cap_win = (max_n_iq + std::min(hist_need, kDHist) + 64 + 3) & ~3;
    const int cap_out = (std::max(max_n_audio + 32 * max_aS + 64, cap_win / 2 + 64) + 3) & ~3;
    const size_t audio_lds = (size_t)(2 * cap_out + cap_win) * sizeof(float) + 64;
    // a block is staged whole in LDS by the modem and audio kernels: that, not a fixed sample count, is what bounds the samples
    // per block and demodulator (a full-width 500 kS/s channel demodulated at its own rate is ~8400 samples per 1/60 s block)
    constexpr size_t kLdsPerWorkgroup = 160 * 1024;
    if (modem_lds > kLdsPerWorkgroup || audio_lds > kLdsPerWorkgroup)
        return reject(fail(CSDR_EUNSUPPORTED, "%d IQ / %d audio samples per block need %zu / %zu bytes of LDS (limit %zu)", max_n_iq, max_n_audio, modem_lds, audio_lds, kLdsPerWorkgroup));
    // FM stereo: a block of x / theta / the two matrix streams staged whole, like the modem kernel
    const int fms_blk = (max_n_iq_fms + 4 * kHilbM + 4 + 3) & ~3, fms_au = (max_n_au_fms + 4 + 3) & ~3;
    const size_t fms_pre_lds = (size_t)fms_blk * sizeof(float), fms_pll_lds = 1024 * sizeof(float) + (size_t)fms_blk * (sizeof(float2) + sizeof(uint32_t)),
                 fms_mix_lds = (size_t)2 * (fms_blk + 4 * kHilbM) * sizeof(float),
                 fms_out_lds = ((size_t)2 * (fms_au + kFmsFirMax) + kFmsFirMax) * sizeof(float) + 64;
    if (n_fms && std::max(std::max(fms_pre_lds, fms_pll_lds), std::max(fms_mix_lds, fms_out_lds)) > kLdsPerWorkgroup)
        return reject(fail(CSDR_EUNSUPPORTED, "FM stereo: %d IQ samples per block need more LDS than a workgroup has", max_n_iq_fms));
    const size_t want[7] = {fe_lds, modem_lds, audio_lds, n_fms ? fms_pre_lds : 0, n_fms ? fms_pll_lds : 0, n_fms ? fms_mix_lds : 0, n_fms ? fms_out_lds : 0};
    const void *fn[7] = {(const void *)demod_frontend, (const void *)demod_modem, (const void *)demod_audio_interp,
                         (const void *)fms_pre, (const void *)fms_pll, (const void *)fms_mix, (const void *)fms_out};
    for (int k = 0; k < 7; ++k)
        if (want[k] > 64 * 1024 && want[k] > b->lds_attr[k]) {
            CSDR_HIP_TRY(hipFuncSetAttribute(fn[k], hipFuncAttributeMaxDynamicSharedMemorySize, (int)want[k]));
            b->lds_attr[k] = want[k];
        }
    const dim3 grid(std::max(1, n_audio_run), NB);
    // one wave per (demodulator, block): the block's few hundred samples pass through five barrier-separated stages, and a
    // single wave crosses a barrier without waiting for anyone (measured 30 us against 41 us with four waves, 64 x 64 blocks)
    const int audio_threads = 64;
    const float2 *chan_out = post_buf(post, pk);
    const int *grp_d = lists_d + 2 * (size_t)b->max_demods;
    if (grp_n[0] > 0)
        CSDR_LAUNCH(c, LANE_FE, KID_FE_GENERIC, demod_frontend, dim3(ranges_for(grp_n[0]), grp_n[0]), dim3(kFeThreads), fe_lds, b->cfgs.p, dyns_d, grp_d + grp_off[0],
                    chan_out, post->chan_stride, total, b->arms.p, c->sintab.p);
#define CSDR_FE_S(S_, CH_)                                                                                                              \
    if (grp_n[S_] > 0)                                                                                                                  \
        CSDR_LAUNCH(c, LANE_FE, KID_FE_S##S_, (demod_frontend_s<S_, CH_>), dim3(ranges_for(grp_n[S_]) + 1, grp_n[S_]), dim3(kFeThreads), (fes_lds_bytes<S_, CH_>()), \
                    b->cfgs.p, dyns_d, grp_d + grp_off[S_], chan_out, post->chan_stride, total, b->arms.p, c->sintab.p)
    CSDR_FE_S(3, 2048); CSDR_FE_S(4, 2048);
    static const bool tw6 = !(getenv("CSDR_FE_TW6") && atoi(getenv("CSDR_FE_TW6")) == 0);
    // CSDR_FE_MERGE=1: depth-5 and depth-6 groups in ONE launch.  Measured on C3 (r3e): 0.749 ms against 0.440 + 0.231 ms for the two
    // launches (the depth-5 workgroups then carry the depth-6 LDS carve and fewer of them are resident), so it is off by default.
    static const bool fe_merge = getenv("CSDR_FE_MERGE") && atoi(getenv("CSDR_FE_MERGE")) == 1;
    const bool merged = fe_merge && tw6 && grp_n[6] > 0 && grp_n[5] > 0;
    if (merged) {
        // both tail-wave depths in one launch: each group gets the range count one round of resident workgroups would give it alone
        const int P6 = ranges_for(grp_n[6]), P5 = ranges_for(grp_n[5]);
        CSDR_LAUNCH(c, LANE_FE, KID_FE_S56, demod_frontend_s56, dim3(std::max(P6, P5) + 1, grp_n[6] + grp_n[5]), dim3(kFeThreads + 64), (fes_lds_bytes<6, 2048>()),
                    b->cfgs.p, dyns_d, grp_d + grp_off[6], grp_n[6], P6, grp_d + grp_off[5], P5, chan_out, post->chan_stride, total, b->arms.p, c->sintab.p);
    }
    if (grp_n[6] > 0 && !merged) {          // depth 6 (AM / SSB from ~500 kS/s channels): tail wave with three tail stages (CSDR_FE_TW6=0: without)
        if (tw6)
            CSDR_LAUNCH(c, LANE_FE, KID_FE_S6, (demod_frontend_s<6, 2048, true>), dim3(ranges_for(grp_n[6]) + 1, grp_n[6]), dim3(kFeThreads + 64), (fes_lds_bytes<6, 2048>()),
                        b->cfgs.p, dyns_d, grp_d + grp_off[6], chan_out, post->chan_stride, total, b->arms.p, c->sintab.p);
        else CSDR_FE_S(6, 2048);
    }
    if (grp_n[7] > 0) {          // interpolating IQ resamplers: chunks of output samples
        int64_t jmax = 0;
        for (int i = 0; i < grp_n[7]; ++i) jmax = std::max<int64_t>(jmax, b->slots[grp_h[grp_off[7] + i]].last_J);
        const int nchunks = (int)((jmax + kFiChunk - 1) / kFiChunk);
        CSDR_LAUNCH(c, LANE_FE, KID_FE_INTERP, demod_frontend_interp, dim3(nchunks + 1, grp_n[7]), dim3(kFeThreads), kFiLds, b->cfgs.p, dyns_d, grp_d + grp_off[7],
                    chan_out, post->chan_stride, total, b->arms.p, c->sintab.p);
    }
    if (grp_n[5] > 0 && !merged)            // depth 5 (NBFM from ~500 kS/s channels): a fifth wave runs the one-wave tail one chunk behind
        CSDR_LAUNCH(c, LANE_FE, KID_FE_S5, (demod_frontend_s<5, 2048, true>), dim3(ranges_for(grp_n[5]) + 1, grp_n[5]), dim3(kFeThreads + 64), (fes_lds_bytes<5, 2048>()),
                    b->cfgs.p, dyns_d, grp_d + grp_off[5], chan_out, post->chan_stride, total, b->arms.p, c->sintab.p);
#undef CSDR_FE_S
    CSDR_HIP_TRY(hipGetLastError());
    // the front-end was the only reader of the channelizer buffer: hand it back to the post object's rotation
    {
        csdr_post *pw = const_cast<csdr_post *>(post);
        if (pw->ctx != c || !c->same(LANE_POST, LANE_FE)) {
            if (pw->n_consumed[pk] >= csdr_post::kMaxConsumers) return fail(CSDR_ERANGE, "too many demodulator banks read one channelizer batch");
            CSDR_HIP_TRY(hipEventRecord(pw->ev_consumed[pk][pw->n_consumed[pk]++], st));
        }
    }
    if (int rc = c->signal(b->ev_fe_done[bpar], LANE_FE, LANE_AUDIO)) return rc;
    // lane AUDIO: modem + audio kernels of this batch
    if (int rc = c->wait(b->ev_fe_done[bpar], LANE_FE, LANE_AUDIO)) return rc;
    if (n_ag > 0)     // freqdem modems need no block-wide pre-pass: only the auto-gain modems run the modem kernel
        CSDR_LAUNCH(c, LANE_AUDIO, KID_MODEM, demod_modem, dim3(n_ag, NB), dim3(audio_threads) /* one wave per block, like the audio kernel */, modem_lds, b->cfgs.p, dyns_d, lists_d + b->max_demods,
                    plans_d, NB, cap_stream, b->mconsts.p, c->sintab.p, b->arms.p, cap_cw);
    if (n_ag > 0)     // the auto-gain recurrence over the blocks, once per demodulator
        CSDR_LAUNCH(c, LANE_AUDIO, KID_GAIN_SCAN, demod_gain_scan, dim3(n_ag), dim3(64), (size_t)NB * sizeof(float), b->cfgs.p, dyns_d, lists_d + b->max_demods, plans_d, NB);
    const int *fms_d = lists_d + b->max_demods + fms_off;
    if (n_fms > 0) {  // FM stereo, ahead of the audio stage: Hilbert r2c of the discriminator output, the pilot loop, the 38 kHz down-mix
        CSDR_LAUNCH(c, LANE_AUDIO, KID_FMS, fms_pre, dim3(n_fms, NB), dim3(64), fms_pre_lds, b->cfgs.p, dyns_d, fms_d, plans_d, NB, b->mconsts.p);
        CSDR_LAUNCH(c, LANE_AUDIO, KID_FMS, fms_pll, dim3(n_fms), dim3(kModemThreads), fms_pll_lds, b->cfgs.p, fms_d, plans_d, NB, fms_blk, c->sintab.p);
        CSDR_LAUNCH(c, LANE_AUDIO, KID_FMS, fms_mix, dim3(n_fms, NB), dim3(64), fms_mix_lds, b->cfgs.p, dyns_d, fms_d, plans_d, NB, fms_blk - 4 * kHilbM, b->mconsts.p, c->sintab.p);
    }
    if (n_audio_run > 0)
        CSDR_LAUNCH(c, LANE_AUDIO, KID_AUDIO, demod_audio_interp, grid, dim3(audio_threads), audio_lds, b->cfgs.p, dyns_d, lists_d, plans_d, NB,
                    cap_out, cap_win, b->arms.p, 0);
    if (n_fms > 0) {  // the second msresamp_rrrf (stereo difference), then matrix + de-emphasis + low-pass into interleaved frames
        CSDR_LAUNCH(c, LANE_AUDIO, KID_AUDIO, demod_audio_interp, dim3(n_fms, NB), dim3(audio_threads), audio_lds, b->cfgs.p, dyns_d, fms_d, plans_d, NB,
                    cap_out, cap_win, b->arms.p, 1);
        CSDR_LAUNCH(c, LANE_AUDIO, KID_FMS_OUT, fms_out, dim3(n_fms, NB), dim3(64), fms_out_lds, b->cfgs.p, dyns_d, fms_d, plans_d, NB, fms_au);
    }
    CSDR_HIP_TRY(hipGetLastError());
    if (int rc = c->signal(b->ev_audio_done[bpar], LANE_AUDIO, LANE_FE)) return rc;
    b->audio_pending[bpar] = true;
    (void)st_a;
    b->seq++;
    return CSDR_OK;
}

extern "C" int csdr_bank_fetch_results(csdr_bank *b, int slot, csdr_block_result *out, int cap_blocks, int *n_blocks) {
    DeviceScope dev__(b ? b->ctx : nullptr);
    if (!b || !out || !n_blocks || slot < 0 || slot >= b->max_demods) return fail(CSDR_EINVAL, "bad argument");
    SlotHost &s = b->slots[slot];
    const int nb = (int)s.results.size();
    if (nb > cap_blocks) return fail(CSDR_ERANGE, "need room for %d blocks", nb);
    *n_blocks = nb;
    if (!nb) return CSDR_OK;
    if (!s.results[0].skipped) {
        hipStream_t st = b->ctx->lanes[LANE_AUDIO];
        CSDR_HIP_TRY(hipMemcpyAsync(b->bout_h.p, s.cfg.bout, nb * sizeof(BlockOut), hipMemcpyDeviceToHost, st));
        CSDR_HIP_TRY(hipStreamSynchronize(st));
        for (int i = 0; i < nb; i++) {
            s.results[i].level_accum = b->bout_h.p[i].level_accum;
            s.results[i].level_count = b->bout_h.p[i].level_count;
            s.results[i].audio_peak = b->bout_h.p[i].audio_peak;
        }
    }
    memcpy(out, s.results.data(), nb * sizeof(csdr_block_result));
    return CSDR_OK;
}
extern "C" int csdr_bank_fetch_audio(csdr_bank *b, int slot, float *host_out, int cap_samples, int *n) {
    DeviceScope dev__(b ? b->ctx : nullptr);
    if (!b || !host_out || !n || slot < 0 || slot >= b->max_demods) return fail(CSDR_EINVAL, "bad argument");
    SlotHost &s = b->slots[slot];
    if (s.last_A > cap_samples) return fail(CSDR_ERANGE, "need room for %d samples", s.last_A);
    *n = s.last_A;
    if (s.last_A) {
        hipStream_t st = b->ctx->lanes[LANE_AUDIO];
        CSDR_HIP_TRY(hipMemcpyAsync(host_out, s.cfg.audio, (size_t)s.last_A * sizeof(float), hipMemcpyDeviceToHost, st));
        CSDR_HIP_TRY(hipStreamSynchronize(st));
    }
    return CSDR_OK;
}
extern "C" int csdr_bank_fetch_iq(csdr_bank *b, int slot, float *host_out, int cap_samples, int *n) {
    DeviceScope dev__(b ? b->ctx : nullptr);
    if (!b || !host_out || !n || slot < 0 || slot >= b->max_demods) return fail(CSDR_EINVAL, "bad argument");
    SlotHost &s = b->slots[slot];
    if (s.last_J > cap_samples) return fail(CSDR_ERANGE, "need room for %d samples", s.last_J);
    *n = s.last_J;
    if (s.last_J) {
        const float2 *cur = s.cfg.iq + (size_t)s.last_parity * ((size_t)kIqHist + s.cfg.cap_iq) + kIqHist;
        hipStream_t st = b->ctx->lanes[LANE_FE];
        CSDR_HIP_TRY(hipMemcpyAsync(host_out, cur, (size_t)s.last_J * sizeof(float2), hipMemcpyDeviceToHost, st));
        CSDR_HIP_TRY(hipStreamSynchronize(st));
    }
    return CSDR_OK;
}
// ModemAnalog::getDemodOutputData of the last block of the last batch: the scaled demodulator output before the audio resampler,
// at most DEMOD_VIS_SIZE samples (the scope tap of DemodulatorThread.cpp:293-305 reads it when the audio is decimated)
extern "C" int csdr_bank_fetch_demod_output(csdr_bank *b, int slot, float *host_out, int cap_samples, int *n) {
    DeviceScope dev__(b ? b->ctx : nullptr);
    if (!b || !host_out || !n || slot < 0 || slot >= b->max_demods) return fail(CSDR_EINVAL, "bad argument");
    SlotHost &s = b->slots[slot];
    *n = 0;
    if (!s.configured || s.last_A == 0 || s.prm.modem == CSDR_MODEM_IQ || s.prm.modem == CSDR_MODEM_CW || s.prm.modem == CSDR_MODEM_FMS) return CSDR_OK;     // (those modems keep no demodOutputData: not ModemAnalog)
    hipStream_t st = b->ctx->lanes[LANE_AUDIO];
    int32_t cnt = 0;
    CSDR_HIP_TRY(hipMemcpyAsync(&cnt, s.cfg.scope_n, sizeof cnt, hipMemcpyDeviceToHost, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    cnt = std::min<int32_t>(cnt, cap_samples);
    if (cnt > 0) {
        CSDR_HIP_TRY(hipMemcpyAsync(host_out, s.cfg.scope, (size_t)cnt * sizeof(float), hipMemcpyDeviceToHost, st));
        CSDR_HIP_TRY(hipStreamSynchronize(st));
    }
    *n = cnt;
    return CSDR_OK;
}
// FM stereo pilot band-pass: the sections this library designs for a modem input rate (five sections, b[15] / a[15] in execution order)
extern "C" int csdr_design_fms_pilot(int64_t sample_rate, float *b15, float *a15) {
    if (!b15 || !a15 || sample_rate <= 0) return fail(CSDR_EINVAL, "bad argument");
    const std::vector<design::Sos> sos = design::fms_pilot_sos(sample_rate);
    for (int q = 0; q < 5; ++q) for (int k = 0; k < 3; ++k) { b15[3 * q + k] = sos[q].b[k]; a15[3 * q + k] = sos[q].a[k]; }
    return CSDR_OK;
}
// replace the pilot band-pass sections of an FM-stereo slot (e.g. with the output of the host's own liquid_iirdes); takes effect now and
// survives reconfiguration of the slot.  b15 == NULL returns to the library's design.
extern "C" int csdr_bank_set_fms_pilot(csdr_bank *b, int slot, const float *b15, const float *a15) {
    DeviceScope dev__(b ? b->ctx : nullptr);
    if (!b || slot < 0 || slot >= b->max_demods) return fail(CSDR_EINVAL, "bad slot");
    SlotHost &s = b->slots[slot];
    if (!s.configured || s.prm.modem != CSDR_MODEM_FMS) return fail(CSDR_ESTATE, "slot %d is not an FM-stereo demodulator", slot);
    if (int rc = b->ctx->sync_all()) return rc;
    if (b15 && a15) { memcpy(s.fms_b, b15, sizeof s.fms_b); memcpy(s.fms_a, a15, sizeof s.fms_a); s.fms_sos_set = true; }
    else {
        s.fms_sos_set = false;
        const std::vector<design::Sos> sos = design::fms_pilot_sos(s.prm.bandwidth);
        for (int q = 0; q < 5; ++q) for (int k = 0; k < 3; ++k) { s.fms_b[3 * q + k] = sos[q].b[k]; s.fms_a[3 * q + k] = sos[q].a[k]; }
    }
    memcpy(s.cfg.fms_b, s.fms_b, sizeof s.fms_b); memcpy(s.cfg.fms_a, s.fms_a, sizeof s.fms_a);
    CSDR_HIP_TRY(hipMemcpy(b->cfgs.p + slot, &s.cfg, sizeof s.cfg, hipMemcpyHostToDevice));
    return CSDR_OK;
}
// FM stereo intermediates of the last batch, for stage-by-stage parity checks: which = 0 the pilot oscillator's phase word after each
// resampled-IQ sample's step (uint32), 1 the stereo-difference stream before its audio resampler (float)
extern "C" int csdr_bank_fetch_fms_stage(csdr_bank *b, int slot, int which, void *host_out, int cap_samples, int *n) {
    DeviceScope dev__(b ? b->ctx : nullptr);
    if (!b || !host_out || !n || slot < 0 || slot >= b->max_demods || which < 0 || which > 1) return fail(CSDR_EINVAL, "bad argument");
    SlotHost &s = b->slots[slot];
    if (!s.configured || s.prm.modem != CSDR_MODEM_FMS) return fail(CSDR_ESTATE, "slot %d is not an FM-stereo demodulator", slot);
    if (s.last_J > cap_samples) return fail(CSDR_ERANGE, "need room for %d samples", s.last_J);
    *n = s.last_J;
    if (s.last_J) {
        hipStream_t st = b->ctx->lanes[LANE_AUDIO];
        CSDR_HIP_TRY(hipMemcpyAsync(host_out, which == 0 ? (const void *)s.cfg.fms_theta : (const void *)s.cfg.d, (size_t)s.last_J * 4, hipMemcpyDeviceToHost, st));
        CSDR_HIP_TRY(hipStreamSynchronize(st));
    }
    return CSDR_OK;
}
extern "C" int csdr_bank_total_audio(csdr_bank *b, int64_t *n) {
    if (!b || !n) return fail(CSDR_EINVAL, "null argument");
    int64_t t = 0;
    for (auto &s : b->slots) t += s.last_A;
    *n = t;
    return CSDR_OK;
}

// =================================================================================================== spectrum
struct csdr_spec {
    csdr_ctx *ctx = nullptr;
    bool ready = false;
    SpecGeom g{};
    int max_frames = 0, nf_last = 0;
    float avg_rate = 0.65f, scale = 1.0f;
    DevBuf<float2> tw4096, tw_hi, tw_lo, tmp, carry, stage_in, raw;
    DevBuf<float> mag;                       // [2][max_frames][N]: the FFT lane fills one copy while the averaging lane reads the other
    DevBuf<float> pairsum, first_b, points;
    uint64_t seq = 0;
    hipEvent_t ev_fft_done[2] = {nullptr, nullptr}, ev_avg_done[2] = {nullptr, nullptr};
    bool avg_pending[2] = {false, false};
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
    if (fft_size < 2 || (fft_size & (fft_size - 1))) return fail(CSDR_EUNSUPPORTED, "fft_size %d: only powers of two are built", fft_size);
    if (max_frames <= 0) return fail(CSDR_EINVAL, "max_frames");
    const int N = 2 * fft_size;                                      // SPECTRUM_VZM 2, SpectrumVisualProcessor.h:11, .cpp:145
    if (N > (1 << 22)) return fail(CSDR_EUNSUPPORTED, "internal FFT of %d points exceeds 2^22", N);
    if (int rc = s->ctx->sync_all()) return rc;
    s->ready = false;
    s->seq = 0; s->avg_pending[0] = s->avg_pending[1] = false;
    SpecGeom &g = s->g;
    g.N = N; g.F = fft_size; g.Ra = 1; g.Rb = 1; g.N2 = N;
    if (N >= 4096) {
        g.N2 = 4096;
        const int R = N / 4096;                                       // 1 .. 1024
        g.Ra = std::min(R, 32); g.Rb = R / g.Ra;
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
    if (g.Ra > 1) if (int rc = s->tmp.reserve(nfN)) return rc;
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

static int spec_run_fft(csdr_spec *s, const FrameSrc &fs, int nf, float *mag, float2 *raw) {
    const SpecGeom &g = s->g;
    csdr_ctx *c = s->ctx;
    if (g.N < 4096) {
        CSDR_LAUNCH(c, LANE_FFT, KID_FFT_ROWS, spec_fft_small, dim3(1, nf), dim3(kFftThreads), (size_t)2 * g.N * sizeof(float2), fs, g.N, s->tw4096.p, mag, raw);
    } else if (g.Ra == 1) {
        CSDR_LAUNCH(c, LANE_FFT, KID_FFT_ROWS, spec_fft_rows4096, dim3(1, nf), dim3(kFftThreads), kRowLdsBytes, fs, g, s->tw4096.p, mag, raw);
    } else {
        // radix passes: Ra-point columns of each frame, then (optionally) Rb-point columns inside each of the Ra sub-sequences
        if (g.Ra <= 16) launch_radix<2>(c, g.Ra, fs, g.N, 1u, nf, s->tw_hi.p, s->tw_lo.p, s->tmp.p);
        else launch_radix<1>(c, g.Ra, fs, g.N, 1u, nf, s->tw_hi.p, s->tw_lo.p, s->tmp.p);
        if (g.Rb > 1) {
            const int L2 = g.N / g.Ra;
            FrameSrc sub{s->tmp.p, nullptr, s->tmp.p + L2, L2, 1 << 30};
            if (g.Rb <= 16) launch_radix<2>(c, g.Rb, sub, L2, (unsigned)g.Ra, nf * g.Ra, s->tw_hi.p, s->tw_lo.p, s->tmp.p);
            else launch_radix<1>(c, g.Rb, sub, L2, (unsigned)g.Ra, nf * g.Ra, s->tw_hi.p, s->tw_lo.p, s->tmp.p);
        }
        FrameSrc rows{s->tmp.p, nullptr, s->tmp.p + g.N, g.N, 1 << 30};
        CSDR_LAUNCH(c, LANE_FFT, KID_FFT_ROWS, spec_fft_rows4096, dim3(g.Ra * g.Rb, nf), dim3(kFftThreads), kRowLdsBytes, rows, g,
                    s->tw4096.p, mag, raw);
    }
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
    // frame groups per workgroup: up to 16 frames each, so a short batch does not pay the set-up of sixteen groups
    // (CSDR_AVG_GROUPS = 4 | 8 | 16 caps the groups: fewer, smaller workgroups let two of them share a CU -- one loads its round while the
    // other scans)
    static const int avg_cap = getenv("CSDR_AVG_GROUPS") ? std::max(1, std::min(kAvgGroups, atoi(getenv("CSDR_AVG_GROUPS")))) : kAvgGroupsDefault;
    const int avg_groups = std::max(1, std::min(avg_cap, (cnt + kAvgGMax - 1) / kAvgGMax));
    CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_AVG, spec_average, dim3(s->n_avg_tiles), dim3(kAvgLanes * avg_groups), avg_lds_bytes(avg_groups), mag + (size_t)f0 * g.N, cnt, g, (double)s->avg_rate,
                s->ma.p, s->maa.p, s->pairsum.p + f0 * F, s->first_b.p + f0, s->ext_w.p + (size_t)f0 * s->n_avg_tiles,
                bins ? s->maaf.p + f0 * F : (float2 *)nullptr, view ? 0 : (hold ? pk_from : cnt));
    CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_TRACK, spec_extrema, dim3(cnt), dim3(256), 64, s->ext_w.p + (size_t)f0 * s->n_avg_tiles, s->n_avg_tiles, s->ext.p + f0);
    const SpecScalars *st_in = s->scal.p + s->scal_parity;
    if (hold) {
        CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_MISC, spec_peak_track, dim3((g.F + 255) / 256), dim3(256), 0, s->maaf.p + f0 * F, cnt, pk_from, g.F, s->peak.p,
                    s->peaksum.p + f0 * F, s->peak_b.p + f0, view ? s->peakf.p + f0 * F : (float2 *)nullptr);
        CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_MISC, spec_peak_trackers, dim3(1), dim3(64), 0, s->ext.p + f0, cnt, pk_from, st_in, s->pk.p, s->pfo.p + f0);
    }
    // trackers of every frame (closed form, one workgroup per frame), then the display: the transposing path takes kDispTpi tiles per workgroup
    CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_TRACK, spec_trackers, dim3(cnt), dim3(kDispThreads), kTrackLds, s->ext.p + f0, cnt, st_in, s->scal.p + (s->scal_parity ^ 1),
                s->fo.p + f0, s->fsc.p + f0, hold ? pk_from : cnt, hold ? s->pfo.p + f0 : (const SpecFrameOut *)nullptr);
    const bool transposing = !view && g.Ra > 1 && (g.Ra >> 1) * g.Rb <= kDispTile && g.F >= kDispTile;
    const int disp_gx = transposing ? std::max(1, (g.F / kDispTile + kDispTpi - 1) / kDispTpi)
                                    : std::max(1, std::min((g.F / 2 + kDispThreads - 1) / kDispThreads, c->wg_slots(spec_display, kDispThreads, kDispLds) / std::max(1, cnt)));
    CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_DISPLAY, spec_display, dim3(disp_gx, cnt), dim3(kDispThreads), kDispLds,
                s->pairsum.p + f0 * F, s->first_b.p + f0, s->fsc.p + f0, g, s->scale, s->points.p + f0 * F, hold ? pk_from : cnt,
                hold ? s->peaksum.p + f0 * F : (const float *)nullptr, hold ? s->peak_b.p + f0 : (const float *)nullptr,
                hold ? s->hold_points.p + f0 * F : (float *)nullptr,
                view ? s->vmap.p : (const int2 *)nullptr, view ? s->maaf.p + f0 * F : (const float2 *)nullptr,
                view && hold ? s->peakf.p + f0 * F : (const float2 *)nullptr);
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
            if (int rc = s->maaf.reserve(nfF)) return rc;
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
    const int mp = c->same(LANE_FFT, LANE_AVG) ? 0 : (int)(s->seq & 1);
    float *mag = s->mag.p + (size_t)mp * s->max_frames * s->g.N;
    if (s->avg_pending[mp]) if (int rc = c->wait(s->ev_avg_done[mp], LANE_AVG, LANE_FFT)) return rc;
    if (int rc = spec_run_fft(s, fs, nf, mag, nullptr)) return rc;
    if (int rc = c->signal(s->ev_fft_done[mp], LANE_FFT, LANE_AVG)) return rc;
    // lane AVG: averaging, extrema, trackers + display points
    if (int rc = c->wait(s->ev_fft_done[mp], LANE_FFT, LANE_AVG)) return rc;
    if (int rc = post(mag)) return rc;
    if (int rc = c->signal(s->ev_avg_done[mp], LANE_AVG, LANE_FFT)) return rc;
    s->avg_pending[mp] = true;
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
                    s->ma.p, s->maa.p, s->ma2.p, s->maa2.p, N, mode, n);
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
    if (nf > 0)
        if (int rc = spec_fft_then(s, fs, nf, [&](float *mag) { return spec_post_frames(s, mag, nf, n_inputs, first_input_has_frame); })) return rc;
    if (mode == CSDR_SPEC_LINES) if (int rc = spec_lines_end(s, lines_x, block_len, lines_n)) return rc;
    if (mode == CSDR_SPEC_CONTIGUOUS) {
        // new carry = samples after the last whole frame (lane FFT: ordered behind the kernels that read the old carry)
        const int64_t total = s->carry_len + n;
        const int rem = (int)(total - (int64_t)nf * N);
        if (nf == 0) {
            CSDR_HIP_TRY(hipMemcpyAsync(s->carry.p + s->carry_len, x, (size_t)n * sizeof(float2), hipMemcpyDeviceToDevice, st));
        } else if (rem > 0) {
            CSDR_HIP_TRY(hipMemcpyAsync(s->carry.p, x + (n - rem), (size_t)rem * sizeof(float2), hipMemcpyDeviceToDevice, st));
        }
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
    const float inv_F = 1.0f / (float)F;                                  // F is a power of two: i * inv_F == i / F exactly
    for (int i = 0; i < F; ++i) { const float y = pts[F + i]; pts[2 * i] = (float)i * inv_F; pts[2 * i + 1] = y; }
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
    if (int rc = s->stage_in.reserve((size_t)N)) return rc;
    if (int rc = s->raw.reserve((size_t)N)) return rc;
    CSDR_HIP_TRY(hipMemcpyAsync(s->stage_in.p, iq_host, (size_t)N * sizeof(float2), hipMemcpyHostToDevice, st));
    FrameSrc fs{s->stage_in.p, nullptr, s->stage_in.p, 0, 1 << 30};
    if (int rc = spec_run_fft(s, fs, 1, nullptr, s->raw.p)) return rc;
    CSDR_HIP_TRY(hipMemcpyAsync(out_host, s->raw.p, (size_t)N * sizeof(float2), hipMemcpyDeviceToHost, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    return CSDR_OK;
}

#include "csdr_io_api.hpp"
