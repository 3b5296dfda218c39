// csdr_bank.hip -- implementation of include/csdr_hip.h (gfx950): csdr_bank (DemodulatorPreThread / DemodulatorThread / modems for a bank of demodulators).  Host-side bookkeeping mirrors the reference's control flow
// (file:line cited per function); all sample arithmetic is in the kernels_*.hpp kernels.
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <map>
#include <memory>

#define CSDR_TU_BANK 1          // this unit is the home of its kernels (common.hpp)
#include "csdr_objects.hpp"
#include "kernels_fms.hpp"

using namespace csdr;

// =================================================================================================== demodulator bank
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
    b->off_lists = ((size_t)max_demods * sizeof(SlotDyn) + 15) & ~(size_t)15;
    b->off_plans = (b->off_lists + (6 * (size_t)max_demods + 8 * 32) * sizeof(int) + 15) & ~(size_t)15;      // all running | auto-gain | grouped by front-end kernel (the specialised kernels' lists in rows of 8 G positions, padded)
    b->table_bytes = (b->off_plans + (size_t)max_demods * (max_blocks + 1) * sizeof(BlockPlan) + 255) & ~(size_t)255;
    if (int rc = b->tables.reserve(2 * b->table_bytes)) return rc;
    for (int k = 0; k < 2; ++k) {
        CSDR_HIP_TRY(hipEventCreateWithFlags(&b->ev_fe_done[k], hipEventDisableTiming));
        CSDR_HIP_TRY(hipEventCreateWithFlags(&b->ev_audio_done[k], hipEventDisableTiming));
        CSDR_HIP_TRY(hipEventCreateWithFlags(&b->ev_tables_read[k], hipEventDisableTiming));
    }
    if (int rc = b->mconsts.reserve(1)) return rc;
    for (int r = 0; r < kStageRing; ++r) {
        if (int rc = b->tables_h[r].reserve(b->table_bytes)) return rc;
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
        if (b->ev_tables_read[k]) (void)hipEventDestroy(b->ev_tables_read[k]);
    }
    for (auto &s : b->slots) if (s.slab) (void)hipFree(s.slab);
    b->cfgs.release(); b->tables.release(); b->arms.release(); b->mconsts.release();
    for (int r = 0; r < kStageRing; ++r) {
        b->tables_h[r].release();
        if (b->stage_ev[r]) (void)hipEventDestroy(b->stage_ev[r]);
    }
    b->bout_h.release();
    b->pcm.release(); b->pcm_jobs.release();
    delete b;
}

// no device modem / audio stage: the internal front-end-only slot, or a host plug-in modem (CSDR_MODEM_HOST) that demodulates the fetched IQ
static int modem_check_rate(int modem, int bw, int audio_rate) {   // Modem*::checkSampleRate (ModemAnalog.cpp:14-19, ModemUSB.cpp:29-37, ModemIQ.cpp:31-33)
    if (modem == CSDR_MODEM_HOST) return bw;                       // the plug-in's own checkSampleRate ran on the host
    if (modem == CSDR_MODEM_IQ || modem == CSDR_MODEM_FRONTEND_ONLY) return audio_rate;
    if (modem == CSDR_MODEM_FMS) return bw < 100000 ? 100000 : bw;      // ModemFMStereo.cpp:27-35
    if (bw < 500) bw = 500;                          // MIN_BANDWIDTH, Modem.h:13
    if ((modem == CSDR_MODEM_USB || modem == CSDR_MODEM_LSB) && (bw % 2)) bw += 1;
    return bw;
}

extern "C" int csdr_bank_configure_slot(csdr_bank *b, int slot, const csdr_demod_params *prm, const csdr_post *post) {
    DeviceScope dev__(b ? b->ctx : nullptr);
    if (prm && (prm->modem < CSDR_MODEM_NBFM || prm->modem > CSDR_MODEM_HOST)) return fail(CSDR_EUNSUPPORTED, "modem %d", prm->modem);
    return bank_configure_slot(b, slot, prm, post);
}
int bank_configure_slot(csdr_bank *b, int slot, const csdr_demod_params *prm, const csdr_post *post) {
    if (!b || !prm || !post) return fail(CSDR_EINVAL, "null argument");
    if (slot < 0 || slot >= b->max_demods) return fail(CSDR_EINVAL, "slot out of range");
    if (!post->configured) return fail(CSDR_ESTATE, "post not configured");
    if (!post->row_order.empty()) return fail(CSDR_ESTATE, "the post has packed rows (a time-slab producer): demodulators read the owner's post");
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

// the smallest j with j * step >= K 2^24 - phase0 (K may be negative): the device's closed form (kernels_demod.hpp: a double quotient and two
// integer corrections -- exact, and a third of the cost of the 64-bit division this walk used to pay twice per demodulator and block)
// The front-end's oscillator table in LDS (kernels_demod.hpp: fes_tab_slot): rows of 32 words, row r rotated by rot * r columns.  The table reads of
// 32 consecutive lanes are the progression (theta0 + 2 p dtheta) >> 22: the rotation with the fewest addresses per bank, over a fixed set of starting
// phases and both samples of a lane's pair.  A layout choice only: any value computes the same samples.
static uint32_t fe_table_rotation(uint32_t dtheta) {
#ifdef CSDR_FE_NOROT
    return 0;                                                      // (A/B builds: the plain table order)
#endif
    if (!dtheta) return 0;
    uint32_t best = 0;
    int best_cost = 1 << 30;
    for (uint32_t rot = 0; rot < 32; rot += 4) {       // multiples of 4: entry i + 256 (the cosine, eight rows on) then sits 256 words after entry i
        int cost = 0;
        for (uint32_t trial = 0; trial < 24; ++trial) {
            const uint32_t th0 = 0x9E3779B9u * (trial + 1u) + (trial & 1u) * dtheta, base = 32u * (trial * 5u);     // lane group of 32 inside a chunk
            uint16_t addr[32][32];
            int cnt[32] = {0};
            int worst = 0;
            for (uint32_t l = 0; l < 32; ++l) {
                const uint32_t th = th0 + 2u * (base + l) * dtheta, i = (th + (1u << 21)) >> 22;
                const uint32_t a = ((i + rot * (i >> 5)) & 31u) | (i & ~31u), bk = a & 31u;
                bool seen = false;
                for (int k = 0; k < cnt[bk]; ++k) seen |= addr[bk][k] == (uint16_t)a;
                if (!seen) { addr[bk][cnt[bk]++] = (uint16_t)a; worst = std::max(worst, cnt[bk]); }
            }
            cost += worst;
        }
        if (cost < best_cost) { best_cost = cost; best = rot; }
    }
    return best;
}

static inline int64_t first_out(int64_t K, uint32_t phase0, uint32_t step) { return resamp_first_out(K, phase0, step); }

extern "C" int csdr_bank_execute(csdr_bank *b, const csdr_post *post) {
    RangeScope range__("csdr_bank_execute");
    DeviceScope dev__(b ? b->ctx : nullptr);
    if (!b || !post) return fail(CSDR_EINVAL, "null argument");
    if (!post->configured || post->n_blocks <= 0) return fail(CSDR_ESTATE, "post has no data");
    if (!post->row_order.empty()) return fail(CSDR_ESTATE, "the post has packed rows (a time-slab producer): demodulators read the owner's post");
    csdr_ctx *c = b->ctx;
    hipStream_t st = c->lanes[LANE_FE], st_a = c->lanes[LANE_AUDIO];
    const int NB = post->n_blocks, M = post->M, Bc = post->block_len / post->hop;
    if (NB > b->max_blocks) return fail(CSDR_ERANGE, "batch of %d blocks exceeds bank capacity %d", NB, b->max_blocks);
    const int bpar = (int)(b->seq & 1);      // which copy of the per-batch device tables this batch uses
    char *table_d = b->tables.p + (size_t)bpar * b->table_bytes;
    SlotDyn *dyns_d = b->dyns_of(table_d);
    int *lists_d = b->lists_of(table_d);
    BlockPlan *plans_d = b->plans_of(table_d);
    const int64_t rate = csdr_post_channel_rate(post);
    // pinned staging set for this batch: wait only for the upload that last used it (kStageRing batches ago)
    const int ring = b->stage_next;
    b->stage_next = (b->stage_next + 1) % kStageRing;
    if (b->stage_used[ring]) CSDR_HIP_TRY(hipEventSynchronize(b->stage_ev[ring]));
    char *table_h = b->tables_h[ring].p;
    SlotDyn *dyns_h = b->dyns_of(table_h);
    int *slot_list_h = b->lists_of(table_h);
    BlockPlan *plans_h = b->plans_of(table_h);
    int n_run = 0, n_ag = 0, max_n_iq = 0, max_n_iq_ag = 0, max_n_audio = 0, warm_max = 0, max_aS = 0, max_cw_audio = 0;
    int max_n_iq_fms = 0, max_n_au_fms = 0;
    std::vector<int> fms_slots;              // FM-stereo slots of this batch (their list shares the auto-gain list's region, from its end)
    int *ag_list_h = slot_list_h + b->max_demods;
    // The per-slot walk below validates AND advances the host-side integer state (oscillator phases, resampler phases, buffer
    // parities).  A rejected batch must leave every slot as it was -- no kernel runs for it -- so the state is snapshotted and
    // put back on any error return of the walk.
    struct Snap { uint32_t theta, dtheta, buf_idx, phase, aphase, abuf, ssb_theta; long long shift_frequency; bool shift_valid; int hist_parity, last_parity, prev_J; };
    static_assert(sizeof(Snap) <= sizeof(csdr_bank::SnapBytes), "snapshot record");
    b->snap.resize((size_t)b->max_demods);
    Snap *snap = reinterpret_cast<Snap *>(b->snap.data());
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
        int ch;
        if (s.route_ch != -2 && s.route_freq == (int64_t)s.prm.frequency && s.route_post_freq == (int64_t)post->frequency && s.route_post_rate == (int64_t)post->sample_rate && s.route_M == M)
            ch = s.route_ch;
        else {
            ch = csdr_post_channel_at(post, s.prm.frequency);
            s.route_freq = (int64_t)s.prm.frequency; s.route_post_freq = (int64_t)post->frequency; s.route_post_rate = (int64_t)post->sample_rate; s.route_M = M; s.route_ch = ch;
        }
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
            if (std::llabs(shift) <= bound) {
                s.dtheta = design::nco_phase_word((float)((2.0 * M_PI) * (((double)std::llabs(shift)) / ((double)rate))));
                s.tab_rot = fe_table_rotation(s.dtheta);
            }
        }
        const bool skipped = std::llabs(shift) > bound;
        s.results.resize(NB);
        if (skipped) {
            for (auto &r : s.results) { memset(&r, 0, sizeof r); r.skipped = 1; r.nco_theta = s.theta; r.resamp_phase = s.phase; r.buffer_index = s.buf_idx; }
            continue;
        }
        SlotDyn &d = dyns_h[si];
        memset(&d, 0, sizeof d);
        d.active = 1; d.chan = data_ch; d.theta0 = s.theta; d.dtheta = s.dtheta; d.tab_rot = s.tab_rot;
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
    // running slots grouped by front-end kernel (filled before the staging set is handed to the copy engine).  The grouping is a function of
    // (running slots, their channels, their cascade classes): rebuilt only when that key changes (a one-block call has ~100 us to spend)
    int *grp_h = slot_list_h + 2 * (size_t)b->max_demods;
    int grp_off[8] = {0}, grp_n[8] = {0}, grp_rows[8] = {0};        // index 0: generic, 3..6: specialised by S (their lists: grp_rows rows of sixteen positions)
    bool merged56 = false;
    {
        auto depth = [](const SlotHost &s) {        // 3 .. 6: the reference's standard half-band pattern of that depth; 7: interpolating; 0: anything else
            const int S = (int)s.iq.S;
            if (s.iq.interp) return 7;
            if (S < 3 || S > 6) return 0;
            for (int e = 0; e < S; ++e) if ((int)s.iq.m[S - 1 - e] != fes_m(S, e)) return 0;
            return S;
        };
        std::vector<int> &key = b->grp_key_scratch;
        key.clear();
        for (int i = 0; i < n_run; ++i) { const int si = slot_list_h[i]; key.push_back(si); key.push_back(dyns_h[si].chan); key.push_back(depth(b->slots[si])); }
        if (key != b->grp_key) {
            b->grp_key = key;
            // Depths 5 and 6 (NBFM beside AM / SSB on ~500 kS/s channels) share ONE launch when the two together still get three or more ranges per
            // demodulator: the two demodulators of a channel usually differ in depth, and only inside one launch can they run side by side on one
            // XCD and read the channel row once (below).  C3 (86 + 170 demodulators): 0.37 + 0.19 -> 0.51 ms per batch; C5 (171 + 341: one range
            // each when merged) 0.318 -> 0.334 ms: not merged.  (Round 3's merged launch was slower: it had no pairing to pay for the two bodies.)
            int n5 = 0, n6 = 0;
            for (int i = 0; i < n_run; ++i) { const int d = key[3 * (size_t)i + 2]; n5 += d == 5; n6 += d == 6; }
            const int fe_resident = std::max(1, c->wg_slots(demod_frontend_s56<2048>, kFeThreads + 64, fes_lds_bytes<6, 2048>()));      // (memoised per context: another device, another answer)
            const bool merge56 = n5 > 0 && n6 > 0 && 4 * (n5 + n6) <= fe_resident && lab_int("CSDR_FE_MERGE56", 1) != 0;
            b->grp_merged56 = merge56;
            auto klass = [&](int i) { const int d = key[3 * (size_t)i + 2]; return (d == 5 && merge56) ? 6 : d; };
            std::vector<int> &list = b->grp_list;
            list.clear();
            std::vector<int> members, loose;
            std::vector<std::pair<int, int>> cols;
            std::map<int, int> waiting;
            for (int k = 0; k < 8; ++k) {
                b->grp_off[k] = (int)list.size(); b->grp_rows[k] = 0;
                if (k < 3 || k > 6) {
                    for (int i = 0; i < n_run; ++i) if (klass(i) == k) list.push_back(key[3 * (size_t)i]);
                    b->grp_n[k] = (int)list.size() - b->grp_off[k];
                    continue;
                }
                // the specialised kernels take their list in rows of sixteen positions: two demodulators of ONE channel at positions w and w + 8 of a
                // row (demod_frontend_s: they then run on the same XCD at the same time and the channel row crosses the fabric once); -1 = empty
                // (only the last row has any: demodulators alone on their channel pair up with each other).  Measured on MI355X, front-end ms per
                // batch: C3N (256 NBFM, 2.1 per channel) 0.59 -> 0.47, C5 0.217 + 0.133 -> 0.209 + 0.109, C2 unchanged.  Rows of 32 with up to
                // four of a channel together were measured as well and lose badly (C3N 0.89, C2 0.27 -> 0.43 ms): half of the grid is then
                // empty positions and a second round.  The pairs of one channel at the same position of CONSECUTIVE rows (a column-major fill;
                // C2 has 3.2, C5 2.6 demodulators per channel) change nothing: C2 0.2728 / 0.2720, C5 0.321 / 0.321, C3 0.506 / 0.501 ms.
                members.clear(); loose.clear(); cols.clear(); waiting.clear();
                for (int i = 0; i < n_run; ++i) if (klass(i) == k) members.push_back(i);
                b->grp_n[k] = (int)members.size();
                if (members.empty()) continue;
                for (int i : members) {
                    const int si = key[3 * (size_t)i], ch = key[3 * (size_t)i + 1];
                    auto it = waiting.find(ch);
                    if (it == waiting.end()) waiting[ch] = si;
                    else { cols.push_back({it->second, si}); waiting.erase(it); }
                }
                for (auto &kv : waiting) loose.push_back(kv.second);
                std::sort(loose.begin(), loose.end());
                for (size_t i = 0; i < loose.size(); i += 2) cols.push_back({loose[i], i + 1 < loose.size() ? loose[i + 1] : -1});
                b->grp_rows[k] = ((int)cols.size() + 7) / 8;
                const size_t base = list.size();
                list.resize(base + (size_t)16 * b->grp_rows[k], -1);
                for (size_t cidx = 0; cidx < cols.size(); ++cidx) {
                    list[base + (cidx / 8) * 16 + (cidx % 8)] = cols[cidx].first;
                    list[base + (cidx / 8) * 16 + (cidx % 8) + 8] = cols[cidx].second;
                }
            }
        }
        memcpy(grp_h, b->grp_list.data(), b->grp_list.size() * sizeof(int));
        for (int k = 0; k < 8; ++k) { grp_off[k] = b->grp_off[k]; grp_n[k] = b->grp_n[k]; grp_rows[k] = b->grp_rows[k]; }
        merged56 = b->grp_merged56;
    }
    // the audio stage runs the slots that have one: compact the head of the list (the front-end groups above are copies)
    int n_audio_run = 0;
    for (int i = 0; i < n_run; ++i) if (!is_fe_only(b->slots[slot_list_h[i]].prm.modem)) slot_list_h[n_audio_run++] = slot_list_h[i];
    // lane FE: the tables and the resampled-IQ buffers of this parity were last read by the audio kernels two batches ago; the tables are fetched
    // BEFORE the lane waits for the channelizer (on separate streams the fetch runs beside it), the front-end kernels behind that wait
    const int pk = post->cur;
    if (b->audio_pending[bpar]) if (int rc = c->wait(b->ev_audio_done[bpar], LANE_AUDIO, LANE_FE)) return rc;
    // Where the channelizer has a stream of its own, the fetch rides on THAT stream, behind the channelizer of this batch: a one-block call is bound by
    // the chain of the demodulators' stream (tables 4 + front-end 19 + modem / audio 22 us and three gaps), the channelizer's stream has
    // half of that; the front-end's wait for the channelizer becomes its wait for the fetch (one event either way).
    const bool fetch_on_post = post->ctx == c && !c->same(LANE_POST, LANE_FE);
    {   // the tables of this batch: fetched from the page-locked staging slot by a kernel (bank_tables_fetch: why not a copy-engine transfer)
        const size_t bytes = b->off_plans + (size_t)b->max_demods * (NB + 1) * sizeof(BlockPlan);
        const int n16 = (int)((bytes + 15) / 16);                   // (the slot and the device table are whole multiples of 256 bytes)
        // (the device copy of this parity was last read by the audio kernels two batches ago: the demodulators' stream orders that by itself,
        //  the channelizer's stream waits for the event recorded behind them)
        if (fetch_on_post && b->tables_read_pending[bpar]) CSDR_HIP_TRY(hipStreamWaitEvent(c->lanes[LANE_POST], b->ev_tables_read[bpar], 0));
        CSDR_LAUNCH(c, fetch_on_post ? LANE_POST : LANE_FE, KID_TABLES, bank_tables_fetch, dim3(std::max(1, std::min(64, (n16 + 255) / 256))), dim3(256), 0,
                    reinterpret_cast<const float4 *>(table_h), reinterpret_cast<float4 *>(table_d), n16);
        CSDR_HIP_TRY(hipGetLastError());
    }
    CSDR_HIP_TRY(hipEventRecord(b->stage_ev[ring], fetch_on_post ? c->lanes[LANE_POST] : st));     // the staging slot is free again once the fetch has run
    b->stage_used[ring] = true;
    if (fetch_on_post) CSDR_HIP_TRY(hipStreamWaitEvent(st, b->stage_ev[ring], 0));                  // (behind the channelizer of this batch on its stream)
    else if (post->ctx != c || !c->same(LANE_POST, LANE_FE)) {      // the channelizer output of this batch must be complete
        if (post->ctx != c) CSDR_HIP_TRY(hipEventRecord(post->ev_ready[pk], post->ctx->lanes[LANE_POST]));
        CSDR_HIP_TRY(hipStreamWaitEvent(st, post->ev_ready[pk], 0));
    }
    // front-end geometry: every slot's batch is cut into P ranges; a range re-runs `warm` inputs in front of it.
    // Slots whose cascade has the reference's standard shape (m = 3..3, 5, 10; 3 <= S <= 6) run the specialised kernel,
    // one launch per depth S; anything else runs the generic one.
    const int64_t total = (int64_t)NB * Bc;
    // ranges per slot, PER LAUNCH (the slots are grouped by cascade depth, one launch per group on the same stream): as many as
    // make that launch's grid ONE round of resident workgroups (each range re-runs `warm` inputs, so fewer, longer ranges
    // waste less), but never shorter than 4 warm-up spans and never fewer than one
    static const int fe_pct = std::max(10, std::min(100, lab_int("CSDR_FE_PCT", 100)));
    const int fe_slots = std::max(1, c->wg_slots(demod_frontend_s<5, 2048, true>, kFeThreads + 64, fes_lds_bytes<5, 2048>()) * fe_pct / 100);
    auto ranges_for = [&](int n_slots) {
        int P = (int)std::max<int64_t>(1, std::min<int64_t>(total / std::max<int64_t>(4096, 4 * (int64_t)warm_max), 4096));
        const int per_slot = fe_slots / std::max(1, n_slots) - 1;                          // one extra workgroup per slot carries the histories
        if (per_slot >= 1) {
            // a batch too short to fill the resident workgroups at that length (the real-time shape: one block per call) is cut down to ONE warm-up
            // span per range: the launch is then bound by the serial chunk chain of a workgroup, not by the work (one block, C3: 19 -> 12 us)
            P = std::max(P, (int)std::min<int64_t>(total / std::max<int64_t>(2048, (int64_t)warm_max), 4096));
            P = std::min(P, per_slot);
        }
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
    // a one-block batch without FM stereo runs modem and audio of a demodulator in ONE launch (demod_modem_audio1: one dependent kernel less in the call's chain)
    const bool fused1 = NB == 1 && n_fms == 0;
    const size_t want[8] = {fe_lds, modem_lds, audio_lds, n_fms ? fms_pre_lds : 0, n_fms ? fms_pll_lds : 0, n_fms ? fms_mix_lds : 0, n_fms ? fms_out_lds : 0,
                            fused1 ? std::max(modem_lds, audio_lds) : 0};
    const void *fn[8] = {(const void *)demod_frontend, (const void *)demod_modem, (const void *)demod_audio_interp,
                         (const void *)fms_pre, (const void *)fms_pll, (const void *)fms_mix, (const void *)fms_out, (const void *)demod_modem_audio1};
    for (int k = 0; k < 8; ++k)
        if (want[k] > 64 * 1024 && want[k] > b->lds_attr[k]) {
            CSDR_HIP_TRY(hipFuncSetAttribute(fn[k], hipFuncAttributeMaxDynamicSharedMemorySize, (int)want[k]));
            b->lds_attr[k] = want[k];
        }
    const dim3 grid(std::max(1, n_audio_run), NB);
    // one wave per (demodulator, block): the block's few hundred samples pass through five barrier-separated stages, and a
    // single wave crosses a barrier without waiting for anyone (measured 30 us against 41 us with four waves, 64 x 64 blocks)
    const int audio_threads = kAudioThreads;
    const float2 *chan_out = post_buf(post, pk);
    const int *grp_d = lists_d + 2 * (size_t)b->max_demods;
    if (grp_n[0] > 0)
        CSDR_LAUNCH(c, LANE_FE, KID_FE_GENERIC, demod_frontend, dim3(ranges_for(grp_n[0]), grp_n[0]), dim3(kFeThreads), fe_lds, b->cfgs.p, dyns_d, grp_d + grp_off[0],
                    chan_out, post->chan_stride, total, b->arms.p, c->sintab.p);
#define CSDR_FE_S(S_, CH_)                                                                                                              \
    if (grp_n[S_] > 0)                                                                                                                  \
        CSDR_LAUNCH(c, LANE_FE, KID_FE_S##S_, (demod_frontend_s<S_, CH_>), dim3(16, grp_rows[S_] * (ranges_for(grp_n[S_]) + 1)), dim3(kFeThreads), (fes_lds_bytes<S_, CH_>()), \
                    b->cfgs.p, dyns_d, grp_d + grp_off[S_], chan_out, post->chan_stride, total, b->arms.p, c->sintab.p, grp_rows[S_])
    CSDR_FE_S(3, 2048); CSDR_FE_S(4, 2048);
    static const bool tw6 = lab_int("CSDR_FE_TW6", 1) != 0;
    if (grp_n[6] > 0) {          // depth 6 (AM / SSB from ~500 kS/s channels): tail wave with three tail stages (CSDR_FE_TW6=0: without)
        if (merged56)
            CSDR_LAUNCH(c, LANE_FE, KID_FE_S56, (demod_frontend_s56<2048>), dim3(16, grp_rows[6] * (ranges_for(grp_n[6]) + 1)), dim3(kFeThreads + 64), (fes_lds_bytes<6, 2048>()),
                        b->cfgs.p, dyns_d, grp_d + grp_off[6], chan_out, post->chan_stride, total, b->arms.p, c->sintab.p, grp_rows[6]);
        else if (tw6)
            CSDR_LAUNCH(c, LANE_FE, KID_FE_S6, (demod_frontend_s<6, 2048, true>), dim3(16, grp_rows[6] * (ranges_for(grp_n[6]) + 1)), dim3(kFeThreads + 64), (fes_lds_bytes<6, 2048>()),
                        b->cfgs.p, dyns_d, grp_d + grp_off[6], chan_out, post->chan_stride, total, b->arms.p, c->sintab.p, grp_rows[6]);
        else CSDR_FE_S(6, 2048);
    }
    if (grp_n[7] > 0) {          // interpolating IQ resamplers: chunks of output samples
        int64_t jmax = 0;
        for (int i = 0; i < grp_n[7]; ++i) jmax = std::max<int64_t>(jmax, b->slots[grp_h[grp_off[7] + i]].last_J);
        const int nchunks = (int)((jmax + kFiChunk - 1) / kFiChunk);
        CSDR_LAUNCH(c, LANE_FE, KID_FE_INTERP, demod_frontend_interp, dim3(nchunks + 1, grp_n[7]), dim3(kFeThreads), kFiLds, b->cfgs.p, dyns_d, grp_d + grp_off[7],
                    chan_out, post->chan_stride, total, b->arms.p, c->sintab.p);
    }
    if (grp_n[5] > 0)                       // depth 5 (NBFM from ~500 kS/s channels): a fifth wave runs the one-wave tail one chunk behind
        CSDR_LAUNCH(c, LANE_FE, KID_FE_S5, (demod_frontend_s<5, 2048, true>), dim3(16, grp_rows[5] * (ranges_for(grp_n[5]) + 1)), dim3(kFeThreads + 64), (fes_lds_bytes<5, 2048>()),
                    b->cfgs.p, dyns_d, grp_d + grp_off[5], chan_out, post->chan_stride, total, b->arms.p, c->sintab.p, grp_rows[5]);
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
    if (fused1) {
        if (n_audio_run > 0)
            CSDR_LAUNCH(c, LANE_AUDIO, KID_AUDIO, demod_modem_audio1, dim3(n_audio_run), dim3(audio_threads), std::max(modem_lds, audio_lds), b->cfgs.p, dyns_d, lists_d, plans_d,
                        cap_stream, b->mconsts.p, c->sintab.p, b->arms.p, cap_cw, cap_out, cap_win);
    } else {
    // freqdem modems need no block-wide pre-pass: only the auto-gain modems run the modem kernel (grid: blocks first, see the kernel)
    if (n_ag > 0)
        CSDR_LAUNCH(c, LANE_AUDIO, KID_MODEM, demod_modem, dim3(NB, n_ag), dim3(audio_threads) /* one wave per block, like the audio kernel */, modem_lds, b->cfgs.p, dyns_d, lists_d + b->max_demods,
                    plans_d, NB, cap_stream, b->mconsts.p, c->sintab.p, b->arms.p, cap_cw);
    if (n_ag > 0 && NB > 1)     // the auto-gain recurrence over the blocks, once per demodulator (a one-block batch: the modem workgroup takes the single step)
        CSDR_LAUNCH(c, LANE_AUDIO, KID_GAIN_SCAN, demod_gain_scan, dim3(n_ag), dim3(64), (size_t)(2 * NB + 1) * sizeof(float), b->cfgs.p, dyns_d, lists_d + b->max_demods, plans_d, NB);
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
    }
    CSDR_HIP_TRY(hipGetLastError());
    if (int rc = c->signal(b->ev_audio_done[bpar], LANE_AUDIO, LANE_FE)) return rc;
    b->audio_pending[bpar] = true;
    if (fetch_on_post) { CSDR_HIP_TRY(hipEventRecord(b->ev_tables_read[bpar], st_a)); b->tables_read_pending[bpar] = true; }      // the last reader of this parity's device tables
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

