// csdr_api.hip -- implementation of include/csdr_hip.h (gfx950).  Host-side bookkeeping mirrors the reference's
// control flow (file:line cited per function); all sample arithmetic is in the kernels_*.hpp kernels.
#include <algorithm>
#include <cmath>
#include <map>
#include <memory>

#include "common.hpp"
#include "design.hpp"
#include "kernels_demod.hpp"
#include "kernels_post.hpp"
#include "kernels_spec.hpp"

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
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else { CSDR_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    CSDR_HIP_TRY(hipEventCreate(&c->ev0));
    CSDR_HIP_TRY(hipEventCreate(&c->ev1));
    std::vector<float> tab = design::nco_sine_table();
    if (int rc = c->sintab.reserve(1024)) return rc;
    CSDR_HIP_TRY(hipMemcpy(c->sintab.p, tab.data(), 1024 * sizeof(float), hipMemcpyHostToDevice));
    *out = c.release();
    return CSDR_OK;
}
extern "C" void csdr_ctx_destroy(csdr_ctx *c) {
    if (!c) return;
    (void)hipStreamSynchronize(c->stream);
    c->sintab.release();
    for (auto &r : c->prof_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : c->prof_pool) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
extern "C" int csdr_ctx_synchronize(csdr_ctx *c) {
    if (!c) return fail(CSDR_EINVAL, "ctx is null");
    CSDR_HIP_TRY(hipStreamSynchronize(c->stream));
    return CSDR_OK;
}
extern "C" void *csdr_ctx_stream(csdr_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" int csdr_ctx_timer_start(csdr_ctx *c) {
    if (!c) return fail(CSDR_EINVAL, "ctx is null");
    CSDR_HIP_TRY(hipEventRecord(c->ev0, c->stream));
    return CSDR_OK;
}
extern "C" int csdr_ctx_timer_stop(csdr_ctx *c, float *ms) {
    if (!c || !ms) return fail(CSDR_EINVAL, "null argument");
    CSDR_HIP_TRY(hipEventRecord(c->ev1, c->stream));
    CSDR_HIP_TRY(hipEventSynchronize(c->ev1));
    CSDR_HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return CSDR_OK;
}
// ---- per-kernel HIP-event profile (bench.py roofline leg) ----
static const char *kKernelNames[KID_COUNT] = {
    "chan_analyze", "chan_update_hist", "dc_tile_ends", "dc_tile_carry", "dc_apply",
    "demod_frontend", "demod_modem", "demod_gain", "demod_audio_interp", "demod_tails",
    "spec_fft_cols", "spec_fft_rows", "spec_average", "spec_trackers", "spec_display", "spec_misc"};
static int prof_drain(csdr_ctx *c) {
    CSDR_HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto &r : c->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { c->prof_ms[r.id] += ms; c->prof_n[r.id] += 1; }
        c->prof_pool.push_back(r.a); c->prof_pool.push_back(r.b);
    }
    c->prof_pending.clear();
    return CSDR_OK;
}
extern "C" int csdr_ctx_profile_enable(csdr_ctx *c, int on) {
    if (!c) return fail(CSDR_EINVAL, "ctx is null");
    if (int rc = prof_drain(c)) return rc;
    c->prof_on = on != 0;
    if (on) for (int i = 0; i < KID_COUNT; i++) { c->prof_ms[i] = 0.0; c->prof_n[i] = 0; }
    return CSDR_OK;
}
extern "C" int csdr_ctx_profile_num_kernels(void) { return KID_COUNT; }
extern "C" const char *csdr_ctx_profile_kernel_name(int id) { return (id >= 0 && id < KID_COUNT) ? kKernelNames[id] : ""; }
extern "C" int csdr_ctx_profile_fetch(csdr_ctx *c, int id, double *total_ms, int64_t *launches) {
    if (!c || id < 0 || id >= KID_COUNT || !total_ms || !launches) return fail(CSDR_EINVAL, "bad argument");
    if (int rc = prof_drain(c)) return rc;
    *total_ms = c->prof_ms[id]; *launches = c->prof_n[id];
    return CSDR_OK;
}

extern "C" int csdr_dev_alloc(csdr_ctx *c, uint64_t bytes, void **dev) {
    if (!c || !dev) return fail(CSDR_EINVAL, "null argument");
    if (hipMalloc(dev, bytes) != hipSuccess) return fail(CSDR_ENOMEM, "hipMalloc(%llu) failed", (unsigned long long)bytes);
    return CSDR_OK;
}
extern "C" int csdr_dev_free(csdr_ctx *c, void *dev) { (void)c; if (dev) CSDR_HIP_TRY(hipFree(dev)); return CSDR_OK; }
extern "C" int csdr_dev_upload(csdr_ctx *c, void *dev, const void *host, uint64_t bytes) {
    if (!c) return fail(CSDR_EINVAL, "ctx is null");
    CSDR_HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    CSDR_HIP_TRY(hipStreamSynchronize(c->stream));
    return CSDR_OK;
}
extern "C" int csdr_dev_download(csdr_ctx *c, void *host, const void *dev, uint64_t bytes) {
    if (!c) return fail(CSDR_EINVAL, "ctx is null");
    CSDR_HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    CSDR_HIP_TRY(hipStreamSynchronize(c->stream));
    return CSDR_OK;
}

// =================================================================================================== SDRPostThread
struct csdr_post {
    csdr_ctx *ctx = nullptr;
    bool configured = false;
    int mode = CSDR_POST_SINGLE, M = 1;
    int64_t sample_rate = 0, chan_bw = 0, frequency = 0;
    int max_block_len = 0, max_blocks = 0;
    int64_t chan_stride = 0;                 // samples per channel row in `out`
    int n_blocks = 0, block_len = 0;         // of the last execute
    std::vector<int64_t> centers;            // chanCenters[M + 1]
    std::vector<int> active_host;
    bool active_dirty = true;
    DevBuf<float2> out, hist0, hist1, stage_in, tw;
    DevBuf<float> taps;
    DevBuf<int> active;
    DevBuf<d2> dc_state, tile_end, tile_in;
    int hist_parity = 0;
    double dc_c = 0.0;                       // feedback coefficient of the DC blocker recurrence
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
    if (!ctx || !out) return fail(CSDR_EINVAL, "null argument");
    *out = new csdr_post();
    (*out)->ctx = ctx;
    return CSDR_OK;
}
extern "C" void csdr_post_destroy(csdr_post *p) {
    if (!p) return;
    (void)hipStreamSynchronize(p->ctx->stream);
    p->out.release(); p->hist0.release(); p->hist1.release(); p->stage_in.release(); p->tw.release();
    p->taps.release(); p->active.release(); p->dc_state.release(); p->tile_end.release(); p->tile_in.release();
    delete p;
}

extern "C" int csdr_post_configure(csdr_post *p, int64_t sample_rate, int num_channels, int mode, int max_block_len, int max_blocks) {
    if (!p) return fail(CSDR_EINVAL, "post is null");
    if (sample_rate <= 0 || num_channels < 1 || max_block_len <= 0 || max_blocks <= 0) return fail(CSDR_EINVAL, "bad sizes");
    if (mode != CSDR_POST_SINGLE && mode != CSDR_POST_PFBCH) return fail(CSDR_EUNSUPPORTED, "channelizer mode %d (PFBCH2 is a later tier)", mode);
    if ((mode == CSDR_POST_SINGLE) != (num_channels == 1)) return fail(CSDR_EINVAL, "SINGLE mode <=> num_channels == 1");
    if (max_block_len % num_channels) return fail(CSDR_EINVAL, "max_block_len must be a multiple of num_channels");
    hipStream_t st = p->ctx->stream;
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    p->mode = mode; p->M = num_channels; p->sample_rate = sample_rate;
    p->chan_bw = sample_rate / num_channels;                       // integer division, SDRPostThread.cpp:408
    p->max_block_len = max_block_len; p->max_blocks = max_blocks;
    const int M = p->M;
    p->chan_stride = (int64_t)max_blocks * (max_block_len / M);
    if (int rc = p->out.reserve((size_t)p->chan_stride * M)) return rc;
    if (int rc = p->dc_state.reserve(1)) return rc;
    CSDR_HIP_TRY(hipMemsetAsync(p->dc_state.p, 0, sizeof(d2), st));
    const int64_t dc_n = (mode == CSDR_POST_SINGLE) ? (int64_t)max_blocks * max_block_len : p->chan_stride;
    const size_t ntiles = (size_t)((dc_n + kDcTile - 1) / kDcTile);
    if (int rc = p->tile_end.reserve(ntiles)) return rc;
    if (int rc = p->tile_in.reserve(ntiles)) return rc;
    // iirfilt_crcf_create_dc_blocker(0.0005f): b = {1, -1}, a = {1, -1 + alpha}  (float)  ->  v = x - a1 v'
    const float a1 = -1.0f + 0.0005f;
    p->dc_c = -(double)a1;
    if (mode == CSDR_POST_PFBCH) {
        std::vector<float> taps = design::channelizer_taps((unsigned)M, 4, 60.0f);
        if (int rc = p->taps.reserve(taps.size())) return rc;
        CSDR_HIP_TRY(hipMemcpyAsync(p->taps.p, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice, st));
        std::vector<float2> tw(M);
        for (int i = 0; i < M; i++) { double a = -2.0 * M_PI * (double)i / (double)M; tw[i] = make_float2((float)std::cos(a), (float)std::sin(a)); }
        if (int rc = p->tw.reserve(M)) return rc;
        CSDR_HIP_TRY(hipMemcpyAsync(p->tw.p, tw.data(), M * sizeof(float2), hipMemcpyHostToDevice, st));
        const size_t H = (size_t)(kChanTaps - 1) * M;
        if (int rc = p->hist0.reserve(H)) return rc;
        if (int rc = p->hist1.reserve(H)) return rc;
        CSDR_HIP_TRY(hipMemsetAsync(p->hist0.p, 0, H * sizeof(float2), st));
        CSDR_HIP_TRY(hipMemsetAsync(p->hist1.p, 0, H * sizeof(float2), st));
        CSDR_HIP_TRY(hipStreamSynchronize(st));   // host vectors above go out of scope
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
    if (n < 0 || n > p->M) return fail(CSDR_EINVAL, "bad channel count");
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

static int run_dc_blocker(csdr_post *p, const float2 *x, float2 *y, int64_t n) {
    hipStream_t st = p->ctx->stream;
    const int ntiles = (int)((n + kDcTile - 1) / kDcTile);
    CSDR_LAUNCH(p->ctx, KID_DC_ENDS, dc_tile_ends, dim3(ntiles), dim3(kDcThreads), 0, x, n, p->dc_c, p->tile_end.p);
    CSDR_LAUNCH(p->ctx, KID_DC_CARRY, dc_tile_carry, dim3(1), dim3(64), 0, p->tile_end.p, ntiles, p->dc_c, p->dc_state.p, p->tile_in.p);
    CSDR_LAUNCH(p->ctx, KID_DC_APPLY, dc_apply, dim3(ntiles), dim3(kDcThreads), 0, x, y, n, p->dc_c, p->tile_in.p, p->dc_state.p);
    CSDR_HIP_TRY(hipGetLastError());
    return CSDR_OK;
}

extern "C" int csdr_post_execute(csdr_post *p, const float *iq, int iq_is_dev, int n_blocks, int block_len, int64_t frequency) {
    if (!p || !p->configured) return fail(CSDR_ESTATE, "post not configured");
    if (!iq || n_blocks <= 0 || block_len <= 0) return fail(CSDR_EINVAL, "bad block arguments");
    if (n_blocks > p->max_blocks || block_len > p->max_block_len) return fail(CSDR_ERANGE, "batch %d x %d exceeds configured %d x %d", n_blocks, block_len, p->max_blocks, p->max_block_len);
    if (block_len % p->M) return fail(CSDR_EINVAL, "block_len %d is not a multiple of numChannels %d", block_len, p->M);
    hipStream_t st = p->ctx->stream;
    const int64_t n = (int64_t)n_blocks * block_len;
    const float2 *x = (const float2 *)iq;
    if (!iq_is_dev) {
        if (int rc = p->stage_in.reserve((size_t)p->max_blocks * p->max_block_len)) return rc;
        CSDR_HIP_TRY(hipMemcpyAsync(p->stage_in.p, iq, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, st));
        x = p->stage_in.p;
    }
    if (frequency != p->frequency || p->centers.empty()) { p->frequency = frequency; post_update_channels(p); }
    p->n_blocks = n_blocks; p->block_len = block_len;
    if (p->mode == CSDR_POST_SINGLE) return run_dc_blocker(p, x, p->out.p, n);       // runSingleCH :284

    const int M = p->M;
    if (p->active_dirty) {
        CSDR_HIP_TRY(hipMemcpyAsync(p->active.p, p->active_host.data(), p->active_host.size() * sizeof(int), hipMemcpyHostToDevice, st));
        CSDR_HIP_TRY(hipStreamSynchronize(st));
        p->active_dirty = false;
    }
    const int n_active = (int)p->active_host.size();
    const int64_t n_frames = n / M;
    // tile size: keep LDS under ~60 KB
    int TF = 128;
    auto lds_bytes = [&](int tf) { return (size_t)((tf + kChanTaps - 1) * M + tf * (M | 1) + M) * sizeof(float2) + (size_t)M * kChanTaps * sizeof(float); };
    while (TF > 1 && lds_bytes(TF) > 60 * 1024) TF >>= 1;
    if (lds_bytes(TF) > 64 * 1024) return fail(CSDR_EUNSUPPORTED, "numChannels %d too large for the direct-DFT channelizer", M);
    float2 *hist = p->hist_parity ? p->hist1.p : p->hist0.p, *hist_new = p->hist_parity ? p->hist0.p : p->hist1.p;
    if (n_active > 0) {
        const int ntiles = (int)((n_frames + TF - 1) / TF);
        CSDR_LAUNCH(p->ctx, KID_CHAN_ANALYZE, chan_analyze, dim3(ntiles), dim3(kChanThreads), lds_bytes(TF), x, hist, p->taps.p, p->tw.p, p->active.p,
                           n_active, M, TF, n_frames, p->out.p, p->chan_stride);
    }
    const int H = (kChanTaps - 1) * M;
    CSDR_LAUNCH(p->ctx, KID_CHAN_HIST, chan_update_hist, dim3((H + 255) / 256), dim3(256), 0, x, n, hist, hist_new, H);
    p->hist_parity ^= 1;
    CSDR_HIP_TRY(hipGetLastError());
    // channel 0 carries the DC spike: block it after de-interleave (:364-375)
    if (n_active > 0 && p->active_host[0] == 0) return run_dc_blocker(p, p->out.p, p->out.p, n_frames);
    return CSDR_OK;
}

extern "C" int64_t csdr_post_channel_bandwidth(const csdr_post *p) { return p ? (p->M == 1 ? p->sample_rate : p->chan_bw) : 0; }
extern "C" int csdr_post_num_channels(const csdr_post *p) { return p ? p->M : 0; }
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
    if (!p || !p->configured || !host_out || !n) return fail(CSDR_EINVAL, "bad argument");
    if (ch == p->M && p->M > 1) ch = p->M / 2;
    if (ch < 0 || ch >= p->M) return fail(CSDR_EINVAL, "channel out of range");
    const int64_t cnt = (int64_t)p->n_blocks * (p->block_len / p->M);
    if (cnt > cap_samples) return fail(CSDR_ERANGE, "need %lld samples", (long long)cnt);
    CSDR_HIP_TRY(hipMemcpyAsync(host_out, p->out.p + (int64_t)ch * p->chan_stride, (size_t)cnt * sizeof(float2), hipMemcpyDeviceToHost, p->ctx->stream));
    CSDR_HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    *n = (int)cnt;
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
    uint32_t theta = 0, dtheta = 0, buf_idx = 0, phase = 0, aphase = 0, ssb_theta = 0;
    long long shift_frequency = 0;
    bool shift_valid = false;
    int hist_parity = 0;
    void *slab = nullptr;
    SlotCfg cfg{};
    // results of the last execute
    std::vector<csdr_block_result> results;
    int last_J = 0, last_A = 0;
};
}  // namespace

struct csdr_bank {
    csdr_ctx *ctx = nullptr;
    int max_demods = 0, max_blocks = 0;
    std::vector<SlotHost> slots;
    DevBuf<SlotCfg> cfgs;
    DevBuf<SlotDyn> dyns;
    DevBuf<int> slot_list;
    DevBuf<BlockPlan> plans;
    DevBuf<float> arms;
    DevBuf<ModemConsts> mconsts;
    PinBuf<SlotDyn> dyns_h;
    PinBuf<int> slot_list_h;
    PinBuf<BlockPlan> plans_h;
    PinBuf<BlockOut> bout_h;
    std::map<uint32_t, int> arm_index;       // key: bit pattern of rate_arb
    std::vector<float> arms_host;
    int n_run = 0, last_nb = 0;
    size_t fe_lds_attr = 0;
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
        CSDR_HIP_TRY(hipStreamSynchronize(b->ctx->stream));
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
    if (!ctx || !out || max_demods <= 0 || max_blocks <= 0) return fail(CSDR_EINVAL, "bad argument");
    std::unique_ptr<csdr_bank> b(new csdr_bank());
    b->ctx = ctx; b->max_demods = max_demods; b->max_blocks = max_blocks;
    b->slots.resize(max_demods);
    if (int rc = b->cfgs.reserve(max_demods)) return rc;
    if (int rc = b->dyns.reserve(max_demods)) return rc;
    if (int rc = b->slot_list.reserve(max_demods)) return rc;
    if (int rc = b->plans.reserve((size_t)max_demods * (max_blocks + 1))) return rc;
    if (int rc = b->mconsts.reserve(1)) return rc;
    if (int rc = b->dyns_h.reserve(max_demods)) return rc;
    if (int rc = b->slot_list_h.reserve(max_demods)) return rc;
    if (int rc = b->plans_h.reserve((size_t)max_demods * (max_blocks + 1))) return rc;
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
    CSDR_HIP_TRY(hipMemcpy(b->mconsts.p, &mc, sizeof mc, hipMemcpyHostToDevice));
    *out = b.release();
    return CSDR_OK;
}

extern "C" void csdr_bank_destroy(csdr_bank *b) {
    if (!b) return;
    (void)hipStreamSynchronize(b->ctx->stream);
    for (auto &s : b->slots) if (s.slab) (void)hipFree(s.slab);
    b->cfgs.release(); b->dyns.release(); b->slot_list.release(); b->plans.release(); b->arms.release(); b->mconsts.release();
    b->dyns_h.release(); b->slot_list_h.release(); b->plans_h.release(); b->bout_h.release();
    delete b;
}

static int modem_check_rate(int modem, int bw) {   // Modem*::checkSampleRate (ModemAnalog.cpp:14-19, ModemUSB.cpp:29-37)
    if (bw < 500) bw = 500;                          // MIN_BANDWIDTH, Modem.h:13
    if ((modem == CSDR_MODEM_USB || modem == CSDR_MODEM_LSB) && (bw % 2)) bw += 1;
    return bw;
}

extern "C" int csdr_bank_configure_slot(csdr_bank *b, int slot, const csdr_demod_params *prm, const csdr_post *post) {
    if (!b || !prm || !post) return fail(CSDR_EINVAL, "null argument");
    if (slot < 0 || slot >= b->max_demods) return fail(CSDR_EINVAL, "slot out of range");
    if (!post->configured) return fail(CSDR_ESTATE, "post not configured");
    if (prm->modem < CSDR_MODEM_NBFM || prm->modem > CSDR_MODEM_LSB) return fail(CSDR_EUNSUPPORTED, "modem %d", prm->modem);
    if (prm->bandwidth <= 0 || prm->audio_sample_rate <= 0) return fail(CSDR_EINVAL, "bad rates");
    SlotHost &s = b->slots[slot];
    CSDR_HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    s.prm = *prm;
    s.prm.bandwidth = modem_check_rate(prm->modem, prm->bandwidth);
    s.chan_rate = csdr_post_channel_bandwidth(post);
    const double iq_ratio = (double)s.prm.bandwidth / (double)s.chan_rate;        // DemodulatorWorkerThread.cpp:99-100
    if (iq_ratio > 1.0) return fail(CSDR_EUNSUPPORTED, "bandwidth %d above the channel rate %lld (interpolating IQ resampler)", s.prm.bandwidth, (long long)s.chan_rate);
    s.iq = design::plan_msresamp((float)iq_ratio, 60.0f);
    const double au_ratio = double(s.prm.audio_sample_rate) / double(s.prm.bandwidth);   // ModemAnalog.cpp:29-30
    s.au = design::plan_msresamp((float)au_ratio, 60.0f);
    if (!s.au.interp) return fail(CSDR_EUNSUPPORTED, "audio decimation (bandwidth %d > audio rate %d) is not built yet", s.prm.bandwidth, s.prm.audio_sample_rate);
    if (s.iq.S > 8 || s.au.S > kMaxHb) return fail(CSDR_EUNSUPPORTED, "resampling ratio needs %u half-band stages", s.iq.S);
    int ia = 0, aa = 0;
    if (int rc = bank_arm_bank(b, s.iq, &ia)) return rc;
    if (int rc = bank_arm_bank(b, s.au, &aa)) return rc;
    // capacities for one execute
    const int64_t max_bc = post->max_block_len / post->M;
    const int64_t cap_iq = (int64_t)std::ceil((double)b->max_blocks * (double)max_bc * iq_ratio) + b->max_blocks + 64;
    const int64_t cap_audio = (int64_t)std::ceil((double)cap_iq * au_ratio) + (int64_t)b->max_blocks * (2 << s.au.S) + 64;
    // one slab per slot
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_mix = carve(2 * kMixHist * sizeof(float2));
    const size_t o_iq = carve((kIqHist + cap_iq) * sizeof(float2));
    const size_t o_d = carve(cap_iq * sizeof(float));
    const size_t o_dh = carve(kDHist * sizeof(float));
    const size_t o_au = carve(cap_audio * sizeof(float));
    const size_t o_g = carve((b->max_blocks + 1) * sizeof(float));
    const size_t o_agc = carve(4 * sizeof(float));
    const size_t o_bm = carve(b->max_blocks * sizeof(float));
    const size_t o_bo = carve(b->max_blocks * sizeof(BlockOut));
    if (s.slab) { (void)hipFree(s.slab); s.slab = nullptr; }
    if (hipMalloc(&s.slab, off) != hipSuccess) return fail(CSDR_ENOMEM, "slot slab of %zu bytes", off);
    CSDR_HIP_TRY(hipMemset(s.slab, 0, off));
    char *base = (char *)s.slab;
    SlotCfg &c = s.cfg;
    memset(&c, 0, sizeof c);
    fill_resamp_cfg(c.rs_iq, s.iq, ia);
    fill_resamp_cfg(c.rs_au, s.au, aa);
    c.modem = s.prm.modem;
    c.mixhist = (float2 *)(base + o_mix); c.iq = (float2 *)(base + o_iq); c.d = (float *)(base + o_d); c.dh = (float *)(base + o_dh);
    c.audio = (float *)(base + o_au); c.gains = (float *)(base + o_g); c.agc = (float *)(base + o_agc);
    c.blockmax = (float *)(base + o_bm); c.bout = (BlockOut *)(base + o_bo);
    c.cap_iq = (int)cap_iq; c.cap_audio = (int)cap_audio;
    const float agc0[3] = {1.0f, 1.0f, 1.0f};                 // ModemAnalog::ModemAnalog(): aOutputCeil(1), MA(1), MAA(1)
    CSDR_HIP_TRY(hipMemcpy(c.agc, agc0, sizeof agc0, hipMemcpyHostToDevice));
    const float g0[1] = {1.0f};
    CSDR_HIP_TRY(hipMemcpy(c.gains, g0, sizeof g0, hipMemcpyHostToDevice));
    CSDR_HIP_TRY(hipMemcpy(b->cfgs.p + slot, &c, sizeof c, hipMemcpyHostToDevice));
    // fresh objects: nco_crcf_create / msresamp create / modem ctor all start from zero state
    s.theta = 0; s.dtheta = 0; s.buf_idx = 0; s.phase = 0; s.aphase = 0; s.hist_parity = 0;
    s.shift_valid = false; s.shift_frequency = 0;
    // ModemUSB/LSB ctor: nco_crcf_set_frequency(ssbShift, 2 pi 0.25) -> the oscillator advances 2^30 per sample
    s.ssb_theta = 0;
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
    if (!b || !post) return fail(CSDR_EINVAL, "null argument");
    if (!post->configured || post->n_blocks <= 0) return fail(CSDR_ESTATE, "post has no data");
    hipStream_t st = b->ctx->stream;
    const int NB = post->n_blocks, M = post->M, Bc = post->block_len / M;
    if (NB > b->max_blocks) return fail(CSDR_ERANGE, "batch of %d blocks exceeds bank capacity %d", NB, b->max_blocks);
    const int64_t rate = csdr_post_channel_bandwidth(post);
    // results of the previous execute are overwritten below: make sure the previous launches are done with the pinned plans
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    int n_run = 0;
    for (int si = 0; si < b->max_demods; ++si) {
        SlotHost &s = b->slots[si];
        s.results.clear(); s.last_J = 0; s.last_A = 0;
        if (!s.configured || !s.active) continue;
        if (s.chan_rate != rate) return fail(CSDR_ESTATE, "slot %d was built for channel rate %lld, post now runs %lld: reconfigure", si, (long long)s.chan_rate, (long long)rate);
        // channel routing: runDemodChannels, SDRPostThread.cpp:317-323 (nearest centre; M == wrap channel = M/2)
        int ch = csdr_post_channel_at(post, s.prm.frequency);
        if (ch < 0) continue;
        const int64_t centre = (M == 1) ? post->frequency : post->centers[ch];
        const int data_ch = (M > 1 && ch == M) ? M / 2 : ch;
        if (M > 1 && std::find(post->active_host.begin(), post->active_host.end(), data_ch) == post->active_host.end())
            return fail(CSDR_ESTATE, "slot %d needs channel %d which the channelizer was told not to produce", si, data_ch);
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
        SlotDyn &d = b->dyns_h.p[si];
        memset(&d, 0, sizeof d);
        d.active = 1; d.chan = data_ch; d.theta0 = s.theta; d.dtheta = s.dtheta;
        d.mixdir = shift == 0 ? 0 : (shift < 0 ? +1 : -1);          // :186-191: shift < 0 -> mix up
        d.buf0 = s.buf_idx; d.phase0 = s.phase; d.aphase0 = s.aphase; d.ssb_theta0 = s.ssb_theta; d.hist_parity = s.hist_parity;
        // per-block plan
        BlockPlan *pl = b->plans_h.p + (size_t)si * (NB + 1);
        const int S = (int)s.iq.S, aS = (int)s.au.S;
        for (int bb = 0; bb <= NB; ++bb) {
            const int64_t K = ((int64_t)s.buf_idx + (int64_t)bb * Bc) >> S;
            const int64_t J = first_out(K, s.phase, s.iq.step);
            const int64_t Q = first_out(J, s.aphase, s.au.step);
            pl[bb].j0 = (int)J; pl[bb].q0 = (int)Q;
        }
        const int64_t Jtot = pl[NB].j0, Qtot = pl[NB].q0;
        if (Jtot > s.cfg.cap_iq - 8 || (Qtot << aS) > s.cfg.cap_audio - 8) return fail(CSDR_ERANGE, "slot %d output exceeds its buffers", si);
        for (int bb = 0; bb < NB; ++bb) {
            csdr_block_result &r = s.results[bb];
            memset(&r, 0, sizeof r);
            r.n_iq = pl[bb + 1].j0 - pl[bb].j0;
            r.n_audio = (int)(((int64_t)(pl[bb + 1].q0 - pl[bb].q0)) << aS);
            r.audio_offset = (int)(((int64_t)pl[bb].q0) << aS);
            if (r.n_iq > kModemMaxBlockIq || r.n_audio > kAudioMaxOut) return fail(CSDR_EUNSUPPORTED, "slot %d: %d IQ / %d audio samples per block exceed the per-workgroup limits", si, r.n_iq, r.n_audio);
            const int64_t Kb = ((int64_t)s.buf_idx + (int64_t)(bb + 1) * Bc) >> S;
            r.buffer_index = (uint32_t)(((int64_t)s.buf_idx + (int64_t)(bb + 1) * Bc) & ((1 << S) - 1));
            r.resamp_phase = (uint32_t)((int64_t)s.phase + (int64_t)pl[bb + 1].j0 * s.iq.step - (Kb << 24));
            r.nco_theta = d.mixdir ? (uint32_t)(s.theta + (uint32_t)((int64_t)(bb + 1) * Bc) * s.dtheta) : s.theta;
        }
        // advance host-side integer state
        const int64_t Ktot = ((int64_t)s.buf_idx + (int64_t)NB * Bc) >> S;
        s.phase = (uint32_t)((int64_t)s.phase + Jtot * (int64_t)s.iq.step - (Ktot << 24));
        s.buf_idx = (uint32_t)(((int64_t)s.buf_idx + (int64_t)NB * Bc) & ((1 << S) - 1));
        if (d.mixdir) s.theta += (uint32_t)((int64_t)NB * Bc) * s.dtheta;
        s.aphase = (uint32_t)((int64_t)s.aphase + Qtot * (int64_t)s.au.step - (Jtot << 24));
        s.ssb_theta += (uint32_t)Jtot * (1u << 30);
        s.hist_parity ^= 1;
        s.last_J = (int)Jtot; s.last_A = (int)(Qtot << aS);
        b->slot_list_h.p[n_run++] = si;
    }
    b->n_run = n_run; b->last_nb = NB;
    if (n_run == 0) return CSDR_OK;
    CSDR_HIP_TRY(hipMemcpyAsync(b->dyns.p, b->dyns_h.p, b->max_demods * sizeof(SlotDyn), hipMemcpyHostToDevice, st));
    CSDR_HIP_TRY(hipMemcpyAsync(b->slot_list.p, b->slot_list_h.p, n_run * sizeof(int), hipMemcpyHostToDevice, st));
    CSDR_HIP_TRY(hipMemcpyAsync(b->plans.p, b->plans_h.p, (size_t)b->max_demods * (NB + 1) * sizeof(BlockPlan), hipMemcpyHostToDevice, st));
    // front-end geometry: split each block into P parts so that (part + warm-up) fits one LDS chunk of ~4K inputs
    int Smax = 0, warm_max = 0;
    for (int i = 0; i < n_run; ++i) {
        const SlotHost &s = b->slots[b->slot_list_h.p[i]];
        const int S = (int)s.iq.S;
        int64_t lo = -(int64_t)(kArmTaps - 1);
        for (int e = S - 1; e >= 0; --e) lo = 2 * lo - (4 * (int)s.iq.m[S - 1 - e] - 2);
        Smax = std::max(Smax, S);
        warm_max = std::max(warm_max, (int)(-lo) + (2 << S));
    }
    if (warm_max + (1 << Smax) > kMixHist) return fail(CSDR_EUNSUPPORTED, "cascade span %d exceeds the carried history", warm_max);
    const int P = std::max(1, (Bc + 3071) / 3072);
    const int gran = 1 << Smax;
    int chunk = ((Bc + P - 1) / P + warm_max + gran + gran - 1) / gran * gran;
    chunk = std::min(chunk, kFeChunkMax / gran * gran);
    if (chunk < gran) return fail(CSDR_EUNSUPPORTED, "half-band depth %d too deep for the front-end chunk", Smax);
    size_t fe_lds = 0;
    for (int i = 0; i < n_run; ++i) fe_lds = std::max(fe_lds, fe_lds_bytes((int)b->slots[b->slot_list_h.p[i]].iq.S, chunk));
    if (fe_lds > b->fe_lds_attr) {
        CSDR_HIP_TRY(hipFuncSetAttribute((const void *)demod_frontend, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fe_lds));
        b->fe_lds_attr = fe_lds;
    }
    const dim3 grid(n_run, NB);
    CSDR_LAUNCH(b->ctx, KID_FRONTEND, demod_frontend, dim3(n_run, NB, P), dim3(kFeThreads), fe_lds, b->cfgs.p, b->dyns.p, b->slot_list.p,
                       post->out.p, post->chan_stride, Bc, NB, chunk, b->arms.p, b->ctx->sintab.p);
    CSDR_LAUNCH(b->ctx, KID_MODEM, demod_modem, grid, dim3(kModemThreads), 0, b->cfgs.p, b->dyns.p, b->slot_list.p, b->plans.p, NB, b->mconsts.p, b->ctx->sintab.p);
    CSDR_LAUNCH(b->ctx, KID_GAIN, demod_gain, dim3((n_run + 63) / 64), dim3(64), 0, b->cfgs.p, b->slot_list.p, n_run, NB);
    CSDR_LAUNCH(b->ctx, KID_AUDIO, demod_audio_interp, grid, dim3(kModemThreads), 0, b->cfgs.p, b->dyns.p, b->slot_list.p, b->plans.p, NB, b->arms.p);
    CSDR_LAUNCH(b->ctx, KID_TAILS, demod_tails, dim3(n_run), dim3(256), 0, b->cfgs.p, b->slot_list.p, b->plans.p, NB);
    CSDR_HIP_TRY(hipGetLastError());
    return CSDR_OK;
}

extern "C" int csdr_bank_fetch_results(csdr_bank *b, int slot, csdr_block_result *out, int cap_blocks, int *n_blocks) {
    if (!b || !out || !n_blocks || slot < 0 || slot >= b->max_demods) return fail(CSDR_EINVAL, "bad argument");
    SlotHost &s = b->slots[slot];
    const int nb = (int)s.results.size();
    if (nb > cap_blocks) return fail(CSDR_ERANGE, "need room for %d blocks", nb);
    *n_blocks = nb;
    if (!nb) return CSDR_OK;
    if (!s.results[0].skipped) {
        CSDR_HIP_TRY(hipMemcpyAsync(b->bout_h.p, s.cfg.bout, nb * sizeof(BlockOut), hipMemcpyDeviceToHost, b->ctx->stream));
        CSDR_HIP_TRY(hipStreamSynchronize(b->ctx->stream));
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
    if (!b || !host_out || !n || slot < 0 || slot >= b->max_demods) return fail(CSDR_EINVAL, "bad argument");
    SlotHost &s = b->slots[slot];
    if (s.last_A > cap_samples) return fail(CSDR_ERANGE, "need room for %d samples", s.last_A);
    *n = s.last_A;
    if (s.last_A) {
        CSDR_HIP_TRY(hipMemcpyAsync(host_out, s.cfg.audio, (size_t)s.last_A * sizeof(float), hipMemcpyDeviceToHost, b->ctx->stream));
        CSDR_HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    }
    return CSDR_OK;
}
extern "C" int csdr_bank_fetch_iq(csdr_bank *b, int slot, float *host_out, int cap_samples, int *n) {
    if (!b || !host_out || !n || slot < 0 || slot >= b->max_demods) return fail(CSDR_EINVAL, "bad argument");
    SlotHost &s = b->slots[slot];
    if (s.last_J > cap_samples) return fail(CSDR_ERANGE, "need room for %d samples", s.last_J);
    *n = s.last_J;
    if (s.last_J) {
        // after demod_tails the batch samples still sit at [kIqHist, kIqHist + J) except the first kIqHist slots' worth of
        // history region, which is separate: the batch region itself is untouched by the tail copy.
        CSDR_HIP_TRY(hipMemcpyAsync(host_out, s.cfg.iq + kIqHist, (size_t)s.last_J * sizeof(float2), hipMemcpyDeviceToHost, b->ctx->stream));
        CSDR_HIP_TRY(hipStreamSynchronize(b->ctx->stream));
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
    int F = 0, N = 0, N1 = 1, N2 = 0, C = 1, R = 1, max_frames = 0, nf_last = 0;
    float avg_rate = 0.65f, scale = 1.0f;
    DevBuf<float2> tw4096, tw_hi, tw_lo, tmp, carry, frame0, stage_in, raw;
    DevBuf<float2> mag2;
    DevBuf<float> pairsum, first_b, points;
    DevBuf<double> ma, maa;
    DevBuf<float2> ext;
    int n_avg_waves = 0;
    DevBuf<SpecFrameOut> fo;
    DevBuf<SpecScalars> scal;
    int carry_len = 0;
    size_t stage_cap = 0;
};

extern "C" int csdr_spec_create(csdr_ctx *ctx, csdr_spec **out) {
    if (!ctx || !out) return fail(CSDR_EINVAL, "null argument");
    *out = new csdr_spec();
    (*out)->ctx = ctx;
    return CSDR_OK;
}
extern "C" void csdr_spec_destroy(csdr_spec *s) {
    if (!s) return;
    (void)hipStreamSynchronize(s->ctx->stream);
    s->tw4096.release(); s->tw_hi.release(); s->tw_lo.release(); s->tmp.release(); s->carry.release(); s->frame0.release();
    s->stage_in.release(); s->raw.release(); s->mag2.release(); s->ext.release(); s->pairsum.release(); s->first_b.release(); s->points.release();
    s->ma.release(); s->maa.release(); s->fo.release(); s->scal.release();
    delete s;
}

extern "C" int csdr_spec_setup(csdr_spec *s, int fft_size, int max_frames) {
    if (!s) return fail(CSDR_EINVAL, "spec is null");
    if (fft_size < 2 || (fft_size & (fft_size - 1))) return fail(CSDR_EUNSUPPORTED, "fft_size %d: only powers of two are built", fft_size);
    if (max_frames <= 0) return fail(CSDR_EINVAL, "max_frames");
    const int N = 2 * fft_size;                                      // SPECTRUM_VZM 2, SpectrumVisualProcessor.h:11, .cpp:145
    if (N > (1 << 21)) return fail(CSDR_EUNSUPPORTED, "internal FFT of %d points exceeds 2^21", N);
    hipStream_t st = s->ctx->stream;
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    s->F = fft_size; s->N = N; s->max_frames = max_frames;
    if (N <= kFftMaxLds) { s->N1 = 1; s->N2 = N; s->C = 1; s->R = 1; }
    else {
        s->N1 = std::max(128, N / (kFftMaxLds / 2)); s->N2 = N / s->N1;     // N2 <= 2048 so one workgroup holds a row PAIR
        s->C = kFftMaxLds / s->N1; s->R = kFftMaxLds / s->N2;
        if (s->C > s->N2) s->C = s->N2;
        if (s->R > s->N1) s->R = s->N1;
    }
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
    const size_t nfN = (size_t)max_frames * N;
    if (s->N1 > 1) if (int rc = s->tmp.reserve(nfN)) return rc;
    if (int rc = s->mag2.reserve(nfN / 2)) return rc;
    s->n_avg_waves = (N / 2 + kAvgThreads - 1) / kAvgThreads;
    if (int rc = s->ext.reserve(nfN / 2)) return rc;
    if (int rc = s->pairsum.reserve(nfN / 2)) return rc;
    if (int rc = s->first_b.reserve(max_frames)) return rc;
    if (int rc = s->points.reserve(nfN)) return rc;               // 2 * F floats per frame
    if (int rc = s->ma.reserve(N)) return rc;
    if (int rc = s->maa.reserve(N)) return rc;
    if (int rc = s->fo.reserve(max_frames)) return rc;
    if (int rc = s->scal.reserve(1)) return rc;
    if (int rc = s->carry.reserve(N)) return rc;
    if (int rc = s->frame0.reserve(N)) return rc;
    CSDR_HIP_TRY(hipMemset(s->ma.p, 0, N * sizeof(double)));      // vector<double>::resize -> zeros (:243-257)
    CSDR_HIP_TRY(hipMemset(s->maa.p, 0, N * sizeof(double)));
    SpecScalars sc = {100.0, 100.0, 0.0, 0.0};                     // ctor :32-33
    CSDR_HIP_TRY(hipMemcpy(s->scal.p, &sc, sizeof sc, hipMemcpyHostToDevice));
    s->carry_len = 0; s->nf_last = 0;
    s->ready = true;
    return CSDR_OK;
}
extern "C" int csdr_spec_set_average_rate(csdr_spec *s, float r) { if (!s) return fail(CSDR_EINVAL, "null"); s->avg_rate = r; return CSDR_OK; }
extern "C" int csdr_spec_set_scale_factor(csdr_spec *s, float f) { if (!s) return fail(CSDR_EINVAL, "null"); s->scale = f; return CSDR_OK; }
extern "C" int csdr_spec_frames(const csdr_spec *s) { return s ? s->nf_last : 0; }

static int spec_run_fft(csdr_spec *s, const FrameSrc &fs, int nf, float2 *mag, float2 *raw) {
    hipStream_t st = s->ctx->stream;
    FrameSrc rows = fs;
    if (s->N1 > 1) {
        CSDR_LAUNCH(s->ctx, KID_FFT_COLS, spec_fft_cols, dim3(s->N2 / s->C, nf), dim3(kFftThreads), 0, fs, s->N1, s->N2, s->C,
                           s->tw4096.p, s->tw_hi.p, s->tw_lo.p, s->tmp.p);
        rows.first = s->tmp.p; rows.rest = s->tmp.p + s->N; rows.stride = s->N;
    }
    CSDR_LAUNCH(s->ctx, KID_FFT_ROWS, spec_fft_rows, dim3(s->N1 / s->R, nf), dim3(kFftThreads), 0, rows, s->N1, s->N2, s->R, s->tw4096.p, mag, raw);
    CSDR_HIP_TRY(hipGetLastError());
    return CSDR_OK;
}

extern "C" int csdr_spec_process(csdr_spec *s, const float *iq, int iq_is_dev, int n_blocks, int block_len, int mode) {
    if (!s || !s->ready) return fail(CSDR_ESTATE, "spec not set up");
    if (!iq || n_blocks <= 0 || block_len <= 0) return fail(CSDR_EINVAL, "bad block arguments");
    hipStream_t st = s->ctx->stream;
    const int N = s->N;
    const int64_t n = (int64_t)n_blocks * block_len;
    const float2 *x = (const float2 *)iq;
    if (!iq_is_dev) {
        if (int rc = s->stage_in.reserve((size_t)n)) return rc;
        CSDR_HIP_TRY(hipMemcpyAsync(s->stage_in.p, iq, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, st));
        x = s->stage_in.p;
    }
    FrameSrc fs;
    int nf = 0;
    if (mode == CSDR_SPEC_FIRST_FRAME) {
        if (block_len < N) return fail(CSDR_EUNSUPPORTED, "block_len %d < internal FFT size %d (overlap priming path :399-421 not built)", block_len, N);
        nf = n_blocks; fs.first = x; fs.rest = x + block_len; fs.stride = block_len;
    } else if (mode == CSDR_SPEC_CONTIGUOUS) {
        const int64_t total = s->carry_len + n;
        nf = (int)(total / N);
        if (nf > 0) {
            if (s->carry_len > 0) {
                CSDR_LAUNCH(s->ctx, KID_SPEC_MISC, spec_assemble, dim3((N + 255) / 256), dim3(256), 0, s->carry.p, s->carry_len, x, N, s->frame0.p);
                fs.first = s->frame0.p;
            } else fs.first = x;
            fs.rest = x + (N - s->carry_len); fs.stride = N;
        }
    } else return fail(CSDR_EINVAL, "mode");
    if (nf > s->max_frames) return fail(CSDR_ERANGE, "%d frames exceed max_frames %d", nf, s->max_frames);
    s->nf_last = nf;
    if (nf > 0) {
        if (int rc = spec_run_fft(s, fs, nf, s->mag2.p, nullptr)) return rc;
        CSDR_LAUNCH(s->ctx, KID_SPEC_AVG, spec_average, dim3(s->n_avg_waves), dim3(kAvgThreads), 0, s->mag2.p, nf, s->N1, s->N2, (double)s->avg_rate,
                    s->ma.p, s->maa.p, s->pairsum.p, s->first_b.p, s->ext.p);
        CSDR_LAUNCH(s->ctx, KID_SPEC_MISC, spec_minmax, dim3(nf), dim3(256), 0, s->ext.p, N / 2, s->fo.p);
        CSDR_LAUNCH(s->ctx, KID_SPEC_TRACK, spec_trackers, dim3(1), dim3(64), 0, nf, s->scal.p, s->fo.p);
        CSDR_LAUNCH(s->ctx, KID_SPEC_DISPLAY, spec_display, dim3((s->F + 255) / 256, nf), dim3(256), 0, s->pairsum.p, s->first_b.p, s->fo.p, s->N1, s->N2, s->scale, s->points.p);
        CSDR_HIP_TRY(hipGetLastError());
    }
    if (mode == CSDR_SPEC_CONTIGUOUS) {
        // new carry = samples after the last whole frame
        const int64_t total = s->carry_len + n;
        const int rem = (int)(total - (int64_t)nf * N);
        if (nf == 0) {
            CSDR_HIP_TRY(hipMemcpyAsync(s->carry.p + s->carry_len, x, (size_t)n * sizeof(float2), hipMemcpyDeviceToDevice, st));
        } else if (rem > 0) {
            CSDR_HIP_TRY(hipMemcpyAsync(s->carry.p, x + (n - rem), (size_t)rem * sizeof(float2), hipMemcpyDeviceToDevice, st));
        }
        s->carry_len = rem;
    }
    return CSDR_OK;
}

extern "C" int csdr_spec_fetch(csdr_spec *s, int frame, float *points_host, int cap_floats, double *fft_ceiling, double *fft_floor) {
    if (!s || !s->ready || !points_host) return fail(CSDR_EINVAL, "bad argument");
    if (frame < 0 || frame >= s->nf_last) return fail(CSDR_EINVAL, "frame %d of %d", frame, s->nf_last);
    if (cap_floats < 2 * s->F) return fail(CSDR_ERANGE, "need %d floats", 2 * s->F);
    hipStream_t st = s->ctx->stream;
    SpecFrameOut fo;
    CSDR_HIP_TRY(hipMemcpyAsync(points_host, s->points.p + (size_t)frame * 2 * s->F, (size_t)2 * s->F * sizeof(float), hipMemcpyDeviceToHost, st));
    CSDR_HIP_TRY(hipMemcpyAsync(&fo, s->fo.p + frame, sizeof fo, hipMemcpyDeviceToHost, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    if (fft_ceiling) *fft_ceiling = fo.point_ceil / (double)s->scale;     // :626
    if (fft_floor) *fft_floor = fo.point_floor;                            // :627
    return CSDR_OK;
}

extern "C" int csdr_spec_fft_only(csdr_spec *s, const float *iq_host, float *out_host) {
    if (!s || !s->ready || !iq_host || !out_host) return fail(CSDR_EINVAL, "bad argument");
    hipStream_t st = s->ctx->stream;
    const int N = s->N;
    if (int rc = s->stage_in.reserve((size_t)N)) return rc;
    if (int rc = s->raw.reserve((size_t)N)) return rc;
    CSDR_HIP_TRY(hipMemcpyAsync(s->stage_in.p, iq_host, (size_t)N * sizeof(float2), hipMemcpyHostToDevice, st));
    FrameSrc fs{s->stage_in.p, s->stage_in.p, 0};
    if (int rc = spec_run_fft(s, fs, 1, nullptr, s->raw.p)) return rc;
    CSDR_HIP_TRY(hipMemcpyAsync(out_host, s->raw.p, (size_t)N * sizeof(float2), hipMemcpyDeviceToHost, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    return CSDR_OK;
}
