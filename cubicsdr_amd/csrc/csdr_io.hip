// csdr_io.hip -- C ABI of the path's edges (it uses the bank's internals: csdr_objects.hpp).
//   csdr_scope   ScopeVisualProcessor's arithmetic                      (src/process/ScopeVisualProcessor.cpp:24-217)
//   csdr_mix     AudioThread's mixing callback                          (src/audio/AudioThread.cpp:88-240)
//   PCM16        AudioFileWAV's payload conversion                      (src/audio/AudioFileWAV.cpp:133-157)
//   csdr_ingest  SDRThread's block buffers -> HBM, one transfer per block (src/sdr/SoapySDRThread.cpp:221-225, :258-266; SDRPostThread.cpp:227-245)
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <deque>
#include <memory>

#include "csdr_objects.hpp"
#include "kernels_io.hpp"

using namespace csdr;

// =================================================================================================== audio scope
struct csdr_scope {
    csdr_ctx *ctx = nullptr;
    bool ready = false, scope_on = true, spectrum_on = true;            // ctor :8-9
    int L = 0, max_frames = 0, max_n = 0, max_scope_samples = 1024;     // maxScopeSamples = DEFAULT_DMOD_FFT_SIZE (:13)
    float rate = 0.65f;                                                 // fft_average_rate (:10)
    DevBuf<float2> tw4096;
    DevBuf<double> ma, maa, trk;
    DevBuf<ScopeFrame> frames_d;
    DevBuf<float> stage, wave_pts, spec_pts;
    DevBuf<ScopeMeta> wave_meta, spec_meta;
    int nf_last = 0;
    bool last_wave = false, last_spec = false, trackers_zeroed = false;
};

extern "C" int csdr_scope_create(csdr_ctx *ctx, csdr_scope **out) {
    if (!ctx || !out) return fail(CSDR_EINVAL, "null argument");
    *out = new csdr_scope();
    (*out)->ctx = ctx;
    return CSDR_OK;
}
extern "C" void csdr_scope_destroy(csdr_scope *s) {
    DeviceScope dev__(s ? s->ctx : nullptr);
    if (!s) return;
    (void)s->ctx->sync_all();
    s->tw4096.release(); s->ma.release(); s->maa.release(); s->trk.release(); s->frames_d.release(); s->stage.release();
    s->wave_pts.release(); s->spec_pts.release(); s->wave_meta.release(); s->spec_meta.release();
    delete s;
}
// setup(fftSize_in) :24-35.  max_frames AudioThreadInputs per csdr_scope_process call, each of at most max_samples floats.
extern "C" int csdr_scope_setup(csdr_scope *s, int fft_size, int max_frames, int max_samples) {
    DeviceScope dev__(s ? s->ctx : nullptr);
    if (!s) return fail(CSDR_EINVAL, "scope is null");
    if (fft_size < 4 || (fft_size & (fft_size - 1)) || fft_size > kFftMaxLds) return fail(CSDR_EUNSUPPORTED, "scope fft size %d: powers of two up to %d", fft_size, kFftMaxLds);
    if (max_frames <= 0 || max_samples <= 0) return fail(CSDR_EINVAL, "bad capacities");
    if (int rc = s->ctx->sync_all()) return rc;
    s->ready = false;
    s->L = fft_size; s->max_frames = max_frames; s->max_n = max_samples;
    std::vector<float2> t(kTwTab);
    for (int i = 0; i < kTwTab; i++) { const double a = -2.0 * M_PI * i / kTwTab; t[i] = make_float2((float)std::cos(a), (float)std::sin(a)); }
    if (int rc = s->tw4096.reserve(kTwTab)) return rc;
    CSDR_HIP_TRY(hipMemcpy(s->tw4096.p, t.data(), kTwTab * sizeof(float2), hipMemcpyHostToDevice));
    const size_t H = (size_t)fft_size / 2;
    if (int rc = s->ma.reserve(H)) return rc;
    if (int rc = s->maa.reserve(H)) return rc;
    if (int rc = s->trk.reserve(4)) return rc;
    // the averagers start from zero (vector::resize, :158-162); the four trackers are constructed to zero (:11-12) and live on across setups
    CSDR_HIP_TRY(hipMemset(s->ma.p, 0, H * sizeof(double)));
    CSDR_HIP_TRY(hipMemset(s->maa.p, 0, H * sizeof(double)));
    if (!s->trackers_zeroed) { CSDR_HIP_TRY(hipMemset(s->trk.p, 0, 4 * sizeof(double))); s->trackers_zeroed = true; }
    if (int rc = s->frames_d.reserve((size_t)max_frames)) return rc;
    if (int rc = s->stage.reserve((size_t)max_frames * max_samples)) return rc;
    if (int rc = s->wave_pts.reserve((size_t)max_frames * 2 * max_samples)) return rc;
    if (int rc = s->spec_pts.reserve((size_t)max_frames * fft_size)) return rc;
    if (int rc = s->wave_meta.reserve((size_t)max_frames)) return rc;
    if (int rc = s->spec_meta.reserve((size_t)max_frames)) return rc;
    // the spectrum kernel's two L-point arrays + its reduction scratch pass the 64 KB default at fftSize 4096
    const size_t lds = (size_t)2 * fft_size * sizeof(float2) + 8 * sizeof(double);
    if (lds > 64 * 1024) CSDR_HIP_TRY(hipFuncSetAttribute((const void *)scope_spectrum, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    s->nf_last = 0;
    s->ready = true;
    return CSDR_OK;
}
extern "C" int csdr_scope_set_enabled(csdr_scope *s, int scope_on, int spectrum_on) {       // setScopeEnabled / setSpectrumEnabled :37-43
    if (!s) return fail(CSDR_EINVAL, "null");
    s->scope_on = scope_on != 0; s->spectrum_on = spectrum_on != 0;
    return CSDR_OK;
}
extern "C" int csdr_scope_set_max_scope_samples(csdr_scope *s, int n) { if (!s || n <= 0) return fail(CSDR_EINVAL, "bad argument"); s->max_scope_samples = n; return CSDR_OK; }
extern "C" int csdr_scope_set_average_rate(csdr_scope *s, float r) { if (!s) return fail(CSDR_EINVAL, "null"); s->rate = r; return CSDR_OK; }

// n_frames AudioThreadInputs through process() (:45-217), in order: each yields a waveform item (scope enabled) and a spectrum
// item (spectrum enabled).  Frame data is host memory (staged with ONE copy per call) or device memory read in place.
extern "C" int csdr_scope_process(csdr_scope *s, const csdr_scope_frame *frames, int n_frames, int data_is_dev) {
    RangeScope range__("csdr_scope_process");
    DeviceScope dev__(s ? s->ctx : nullptr);
    if (!s || !s->ready) return fail(CSDR_ESTATE, "scope not set up");
    if (!frames || n_frames <= 0) return fail(CSDR_EINVAL, "no frames");
    if (n_frames > s->max_frames) return fail(CSDR_ERANGE, "%d frames exceed max_frames %d", n_frames, s->max_frames);
    csdr_ctx *c = s->ctx;
    hipStream_t st = c->lanes[LANE_AVG];
    if (int rc = c->lane_begin(LANE_AVG)) return rc;
    if (data_is_dev) {
        // frames that lie in HBM were produced by kernels on another stage's stream (the bank's audio kernels): everything enqueued on
        // the other streams so far precedes the scope kernels
        for (int l = 0; l < c->n_phys; ++l) {
            if (c->phys[l] == st) continue;
            CSDR_HIP_TRY(hipEventRecord(c->ev_lane[l], c->phys[l]));
            CSDR_HIP_TRY(hipStreamWaitEvent(st, c->ev_lane[l], 0));
        }
    }
    std::vector<ScopeFrame> fr((size_t)n_frames);
    std::vector<float> host_stage;
    size_t off = 0;
    for (int i = 0; i < n_frames; ++i) {
        const csdr_scope_frame &f = frames[i];
        if (!f.data || f.n <= 0) return fail(CSDR_EINVAL, "frame %d is empty (the reference discards such inputs before process() does anything)", i);
        if (f.n > s->max_n) return fail(CSDR_ERANGE, "frame %d: %d samples exceed %d", i, f.n, s->max_n);
        if (f.channels != 1 && f.channels != 2) return fail(CSDR_EINVAL, "frame %d: %d channels", i, f.channels);
        if (f.layout < 0 || f.layout > 2) return fail(CSDR_EINVAL, "frame %d: layout", i);
        ScopeFrame &d = fr[(size_t)i];
        d.data = f.data; d.n_dev = data_is_dev ? f.n_dev : nullptr; d.n = f.n; d.channels = f.channels; d.type = f.type; d.layout = f.layout;
        d.scale = f.layout ? f.scale : 1.0f;
        d.sample_rate = f.sample_rate; d.input_rate = f.input_rate; d.pad = 0;
        // spectrum points kept: fftSize / 2, scaled down when the tap runs below the rate it is labelled with (:194-200, float arithmetic)
        unsigned out_size = (unsigned)s->L / 2;
        if (f.sample_rate != f.input_rate) out_size = (unsigned)(int)std::floor((float)out_size * ((float)f.sample_rate / (float)f.input_rate));
        d.out_size = (int)std::min<unsigned>(out_size, (unsigned)s->L / 2);
        if (!data_is_dev) {
            host_stage.insert(host_stage.end(), f.data, f.data + f.n);
            d.data = s->stage.p + off;
            off += (size_t)f.n;
        }
    }
    if (!data_is_dev) CSDR_HIP_TRY(hipMemcpyAsync(s->stage.p, host_stage.data(), off * sizeof(float), hipMemcpyHostToDevice, st));
    CSDR_HIP_TRY(hipMemcpyAsync(s->frames_d.p, fr.data(), fr.size() * sizeof(ScopeFrame), hipMemcpyHostToDevice, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));                                      // the pageable staging vectors go out of scope below
    if (s->scope_on)
        CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_MISC, scope_wave, dim3(n_frames), dim3(kScopeThreads), 64, s->frames_d.p, s->max_scope_samples, s->max_n, s->wave_pts.p, s->wave_meta.p);
    if (s->spectrum_on)
        CSDR_LAUNCH(c, LANE_AVG, KID_SPEC_MISC, scope_spectrum, dim3(1), dim3(kFftThreads), (size_t)2 * s->L * sizeof(float2) + 8 * sizeof(double), s->frames_d.p, n_frames, s->L,
                    (double)s->rate, s->tw4096.p, s->ma.p, s->maa.p, s->trk.p, s->spec_pts.p, s->spec_meta.p);
    CSDR_HIP_TRY(hipGetLastError());
    s->nf_last = n_frames; s->last_wave = s->scope_on; s->last_spec = s->spectrum_on;
    return CSDR_OK;
}
extern "C" int csdr_scope_frames(const csdr_scope *s) { return s ? s->nf_last : 0; }
// item of frame `frame` of the last process: which = 0 the waveform (ScopeRenderData with spectrum == false), 1 the spectrum.
// info->n_floats == 0 when that item was not produced (disabled).
extern "C" int csdr_scope_fetch(csdr_scope *s, int frame, int which, float *points, int cap_floats, csdr_scope_info *info) {
    DeviceScope dev__(s ? s->ctx : nullptr);
    if (!s || !s->ready || !points || !info) return fail(CSDR_EINVAL, "bad argument");
    if (frame < 0 || frame >= s->nf_last || which < 0 || which > 1) return fail(CSDR_EINVAL, "frame %d of %d", frame, s->nf_last);
    memset(info, 0, sizeof *info);
    if ((which == 0 && !s->last_wave) || (which == 1 && !s->last_spec)) return CSDR_OK;
    hipStream_t st = s->ctx->lanes[LANE_AVG];
    ScopeMeta m;
    CSDR_HIP_TRY(hipMemcpyAsync(&m, (which ? s->spec_meta.p : s->wave_meta.p) + frame, sizeof m, hipMemcpyDeviceToHost, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    if (m.n_floats > cap_floats) return fail(CSDR_ERANGE, "need %d floats", m.n_floats);
    const float *src = which ? s->spec_pts.p + (size_t)frame * s->L : s->wave_pts.p + (size_t)frame * 2 * s->max_n;
    if (m.n_floats > 0) {
        CSDR_HIP_TRY(hipMemcpyAsync(points, src, (size_t)m.n_floats * sizeof(float), hipMemcpyDeviceToHost, st));
        CSDR_HIP_TRY(hipStreamSynchronize(st));
    }
    info->mode = m.mode; info->spectrum = m.spectrum; info->channels = m.channels; info->input_rate = m.input_rate; info->sample_rate = m.sample_rate;
    info->fft_size = m.fft_size; info->n_floats = m.n_floats; info->fft_floor = m.fft_floor; info->fft_ceil = m.fft_ceil;
    return CSDR_OK;
}

// The audio-scope tap of a demodulator (DemodulatorThread.cpp:240-316) for the LAST block of the bank's last execute, as a frame
// whose data pointers lie in HBM: stereo modems hand their interleaved audio (re-ordered by the frame's layout as it is read),
// mono modems their audio when it outnumbers the block's IQ samples, else the gain-scaled demodulator output.
// out->n == 0: the block produced nothing to show.
extern "C" int csdr_bank_scope_frame(csdr_bank *b, int slot, csdr_scope_frame *out) {
    if (!b || !out || slot < 0 || slot >= b->max_demods) return fail(CSDR_EINVAL, "bad argument");
    memset(out, 0, sizeof *out);
    const SlotHost &s = b->slots[slot];
    if (!s.configured || s.results.empty()) return CSDR_OK;
    const csdr_block_result &r = s.results.back();
    if (r.skipped || r.n_iq == 0 || is_fe_only(s.prm.modem)) return CSDR_OK;
    const int bw = s.prm.bandwidth, arate = s.prm.audio_sample_rate;
    out->sample_rate = bw; out->input_rate = bw;                                 // inp->sampleRate (:254-255)
    out->scale = 1.0f;
    const float *audio = s.cfg.audio + r.audio_offset;
    if (s.prm.modem == CSDR_MODEM_IQ || s.prm.modem == CSDR_MODEM_FMS) {          // ati->channels == 2 (:269-291)
        out->channels = 2; out->type = 1;
        out->n = std::min(r.n_audio, 2 * kScopeMax);
        out->data = audio;
        if (s.prm.modem == CSDR_MODEM_IQ) { out->layout = 2; out->scale = 0.75f; }       // real * 0.75 | imag * 0.75 of the resampled IQ = (b, a) of the (imag, real) audio pairs
        else { out->layout = 1; out->input_rate = arate; out->sample_rate = 36000; }     // left | right
        return CSDR_OK;
    }
    out->channels = 1; out->type = 0; out->layout = 0;
    const bool has_demod_out = !(s.prm.modem == CSDR_MODEM_CW);                   // every other mono modem keeps ModemAnalog's demodOutputData
    if (r.n_audio > r.n_iq || !has_demod_out) {                                   // :295-300
        out->input_rate = arate;
        out->n = std::min(r.n_audio, kScopeMax);
        out->data = audio;
    } else {                                                                      // :301-306
        out->n = std::min(r.n_iq, kScopeMax);
        out->data = s.cfg.scope;
        out->n_dev = s.cfg.scope_n;
    }
    return CSDR_OK;
}

// =================================================================================================== audio mix-down
namespace {
struct MixSeg { uint32_t ring_pos; int32_t n; int32_t channels, rate; uint32_t peak_idx; };   // one AudioThreadInput of a source's queue
struct MixSource {
    bool bound = false, active = true;                                   // boundThreads membership, AudioThread::isActive
    float gain = 1.0f;                                                   // AudioThread::gain
    DevBuf<float> ring, peaks;
    uint32_t mask = 0, pmask = 0, wpos = 0, pwpos = 0;
    std::deque<MixSeg> queue;                                            // inputQueue
    size_t queue_cap = 0;
    bool have_cur = false;                                               // currentInput != nullptr
    MixSeg cur{};
    size_t ptr = 0;                                                      // audioQueuePtr
    int64_t buffered = 0;                                                // floats in ring not yet released (overrun guard)
};
}  // namespace
struct csdr_mix {
    csdr_ctx *ctx = nullptr;
    int rate = 48000;                                                    // the controller's sample rate
    std::vector<MixSource> src;
    DevBuf<MixBuffer> bufs_d; DevBuf<MixPiece> pieces_d; DevBuf<MixPeakRef> refs_d; DevBuf<RingPush> push_d; DevBuf<PcmJob> pcm_d;
    DevBuf<float> out, out_peak, fixed_peak;
    DevBuf<int16_t> pcm;
    int last_frames = 0, last_buffers = 0;
};

static uint32_t pow2_at_least(size_t n) { uint32_t p = 1; while (p < n) p <<= 1; return p; }

extern "C" int csdr_mix_create(csdr_ctx *ctx, int max_sources, int ring_floats, int sample_rate, csdr_mix **out) {
    DeviceScope dev__(ctx);
    if (!ctx || !out || max_sources <= 0 || max_sources > kMixMaxSources || ring_floats <= 0) return fail(CSDR_EINVAL, "bad argument");
    std::unique_ptr<csdr_mix> m(new csdr_mix());
    m->ctx = ctx; m->rate = sample_rate;
    m->src.resize((size_t)max_sources);
    const uint32_t cap = pow2_at_least((size_t)ring_floats);
    for (auto &s : m->src) {
        if (int rc = s.ring.reserve(cap)) return rc;
        if (int rc = s.peaks.reserve(1024)) return rc;
        s.mask = cap - 1; s.pmask = 1023;
    }
    *out = m.release();
    return CSDR_OK;
}
extern "C" void csdr_mix_destroy(csdr_mix *m) {
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m) return;
    (void)m->ctx->sync_all();
    for (auto &s : m->src) { s.ring.release(); s.peaks.release(); }
    m->bufs_d.release(); m->pieces_d.release(); m->refs_d.release(); m->push_d.release(); m->pcm_d.release(); m->out.release(); m->out_peak.release(); m->fixed_peak.release(); m->pcm.release();
    delete m;
}
// bindThread / removeThread (:52-74), setActive, setGain; queue_blocks = the source's inputQueue capacity (0: unbounded)
extern "C" int csdr_mix_set_source(csdr_mix *m, int source, int bound, int active, float gain, int queue_blocks) {
    if (!m || source < 0 || source >= (int)m->src.size()) return fail(CSDR_EINVAL, "bad source");
    MixSource &s = m->src[(size_t)source];
    if (s.bound && !bound) { s.queue.clear(); s.have_cur = false; s.ptr = 0; s.buffered = 0; }
    s.bound = bound != 0; s.active = active != 0; s.gain = gain; s.queue_cap = (size_t)std::max(0, queue_blocks);
    return CSDR_OK;
}
static int mix_enqueue(csdr_mix *m, MixSource &s, int n, int channels, int rate) {
    if (s.queue_cap && s.queue.size() >= s.queue_cap) return 1;                    // try_push on a full queue: the block is dropped
    if (s.queue.size() + (s.have_cur ? 1 : 0) >= (size_t)s.pmask) return 1;         // one peak slot per queued block: an unbounded queue (queue_blocks 0) is full here
    if ((int64_t)n + s.buffered > (int64_t)s.mask + 1) return fail(CSDR_ERANGE, "mixer ring overrun: render before pushing more");
    s.queue.push_back(MixSeg{s.wpos, n, channels, rate, s.pwpos});
    s.wpos = (s.wpos + (uint32_t)n) & s.mask; s.pwpos = (s.pwpos + 1) & s.pmask; s.buffered += n;
    (void)m;
    return CSDR_OK;
}
// one AudioThreadInput (host or device memory) onto a source's queue; returns 1 when the queue was full and the block was dropped
extern "C" int csdr_mix_push(csdr_mix *m, int source, const float *audio, int is_dev, int n_floats, int channels, int sample_rate, float peak) {
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m || source < 0 || source >= (int)m->src.size() || n_floats < 0 || (n_floats && !audio)) return fail(CSDR_EINVAL, "bad argument");
    MixSource &s = m->src[(size_t)source];
    if (!s.bound) return fail(CSDR_ESTATE, "source %d is not bound", source);
    hipStream_t st = m->ctx->lanes[LANE_AUDIO];
    const uint32_t w0 = s.wpos, pw0 = s.pwpos;
    if (int rc = mix_enqueue(m, s, n_floats, channels, sample_rate)) return rc;
    if (is_dev) {
        // device audio may have been produced on any lane of this context or on the boundary stream: the copy starts behind all of them
        csdr_ctx *c = m->ctx;
        for (int l = 0; l < c->n_phys; ++l) {
            if (c->phys[l] == st) continue;
            CSDR_HIP_TRY(hipEventRecord(c->ev_lane[l], c->phys[l]));
            CSDR_HIP_TRY(hipStreamWaitEvent(st, c->ev_lane[l], 0));
        }
        if (int rc = c->lane_begin(LANE_AUDIO)) return rc;
    }
    const hipMemcpyKind k = is_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const uint32_t first = std::min<uint32_t>((uint32_t)n_floats, s.mask + 1 - w0);
    if (first) CSDR_HIP_TRY(hipMemcpyAsync(s.ring.p + w0, audio, (size_t)first * sizeof(float), k, st));
    if ((uint32_t)n_floats > first) CSDR_HIP_TRY(hipMemcpyAsync(s.ring.p, audio + first, ((size_t)n_floats - first) * sizeof(float), k, st));
    CSDR_HIP_TRY(hipMemcpyAsync(s.peaks.p + pw0, &peak, sizeof(float), hipMemcpyHostToDevice, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));                                       // `peak` lives on this stack frame, a pageable source may be reused on return
    return CSDR_OK;
}
// The audio of EVERY block of the bank's last execute, for n (slot, source) pairs, appended in HBM by ONE kernel: samples and
// per-block peaks never visit the host (the block sizes are host bookkeeping already: csdr_block_result.n_audio).
extern "C" int csdr_mix_push_bank(csdr_mix *m, csdr_bank *b, const int *slots, const int *sources, int n) {
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m || !b || !slots || !sources || n <= 0) return fail(CSDR_EINVAL, "bad argument");
    if (b->ctx != m->ctx) return fail(CSDR_EINVAL, "bank and mixer belong to different contexts");
    std::vector<RingPush> jobs;
    int max_n = 0;
    for (int i = 0; i < n; ++i) {                    // every pair is checked before anything is enqueued: a bad pair leaves no half-pushed queue entries
        if (slots[i] < 0 || slots[i] >= b->max_demods || sources[i] < 0 || sources[i] >= (int)m->src.size()) return fail(CSDR_EINVAL, "pair %d", i);
        if (!m->src[(size_t)sources[i]].bound) return fail(CSDR_ESTATE, "source %d is not bound", sources[i]);
    }
    struct Undo { MixSource *s; uint32_t wpos, pwpos; size_t qsize; int64_t buffered; };
    std::vector<Undo> undo;
    for (int i = 0; i < n; ++i) {
        const SlotHost &sl = b->slots[(size_t)slots[i]];
        MixSource &s = m->src[(size_t)sources[i]];
        if (!sl.configured || sl.results.empty() || sl.results[0].skipped || is_fe_only(sl.prm.modem)) continue;
        const int channels = (sl.prm.modem == CSDR_MODEM_IQ || sl.prm.modem == CSDR_MODEM_FMS) ? 2 : 1;
        RingPush jb{};
        jb.src = sl.cfg.audio; jb.ring = s.ring.p; jb.mask = s.mask; jb.wpos = s.wpos; jb.n = 0;
        jb.peaks_src = sl.cfg.bout; jb.peaks_dst = s.peaks.p; jb.peaks_mask = s.pmask; jb.peaks_wpos = s.pwpos; jb.n_peaks = 0;
        undo.push_back(Undo{&s, s.wpos, s.pwpos, s.queue.size(), s.buffered});
        // blocks the queue refuses (full) still occupy ring space of this copy: simplest is to stop at the first refusal
        for (const csdr_block_result &r : sl.results) {
            const int rc = mix_enqueue(m, s, r.n_audio, channels, sl.prm.audio_sample_rate);
            if (rc < 0) {                             // ring overrun: the copy kernel will not run, so nothing of this call may stay queued
                for (const Undo &u : undo) { u.s->queue.resize(u.qsize); u.s->wpos = u.wpos; u.s->pwpos = u.pwpos; u.s->buffered = u.buffered; }
                return rc;
            }
            if (rc == 1) break;
            jb.n += r.n_audio; jb.n_peaks += 1;
        }
        if (jb.n_peaks) { jobs.push_back(jb); max_n = std::max(max_n, jb.n); }
    }
    if (jobs.empty()) return CSDR_OK;
    csdr_ctx *c = m->ctx;
    hipStream_t st = c->lanes[LANE_AUDIO];
    if (int rc = m->push_d.reserve(std::max<size_t>(jobs.size(), 256))) return rc;
    CSDR_HIP_TRY(hipMemcpyAsync(m->push_d.p, jobs.data(), jobs.size() * sizeof(RingPush), hipMemcpyHostToDevice, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));                                       // (`jobs` is pageable and local)
    CSDR_LAUNCH(c, LANE_AUDIO, KID_MIX, ring_push, dim3((unsigned)std::max(1, std::min(64, (max_n + 255) / 256)), (unsigned)jobs.size()), dim3(256), 0,
                m->push_d.p, (int)sizeof(BlockOut), (int)offsetof(BlockOut, audio_peak));
    CSDR_HIP_TRY(hipGetLastError());
    return CSDR_OK;
}
extern "C" int csdr_mix_queued(const csdr_mix *m, int source) {
    if (!m || source < 0 || source >= (int)m->src.size()) return 0;
    return (int)m->src[(size_t)source].queue.size();
}

// n_buffers consecutive device callbacks of `frames` stereo frames (audioCallback, :88-240).  The walk below decides, per buffer and
// source, exactly what the callback decides -- whether the source takes part, which block is current, when the next one is taken --
// and hands the arithmetic to audio_mix as pieces.  out_host (may be null: csdr_mix_fetch_pcm16 / a later render read the device
// copy) receives n_buffers * frames * 2 floats.
extern "C" int csdr_mix_render(csdr_mix *m, int frames, int n_buffers, float *out_host) {
    RangeScope range__("csdr_mix_render");
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m || frames <= 0 || n_buffers <= 0) return fail(CSDR_EINVAL, "bad argument");
    csdr_ctx *c = m->ctx;
    std::vector<MixBuffer> bufs((size_t)n_buffers);
    std::vector<MixPiece> pieces;
    std::vector<MixPeakRef> refs;
    std::vector<int64_t> released(m->src.size(), 0);
    for (int k = 0; k < n_buffers; ++k) {
        MixBuffer mb{};
        mb.piece0 = (int)pieces.size(); mb.ref0 = (int)refs.size();
        int slot_in_buffer = 0;
        for (size_t j = 0; j < m->src.size(); ++j) {
            MixSource &s = m->src[j];
            if (!s.bound || !s.active || s.queue.empty()) continue;            // terminated / no queue / EMPTY queue / inactive (:117)
            auto take = [&]() -> bool {                                        // try_pop(currentInput)
                if (s.queue.empty()) return false;
                if (s.have_cur) released[j] += s.cur.n;
                s.cur = s.queue.front(); s.queue.pop_front(); s.have_cur = true;
                return true;
            };
            auto drop = [&]() { if (s.have_cur) released[j] += s.cur.n; s.have_cur = false; };
            if (!s.have_cur) { s.ptr = 0; (void)take(); continue; }           // the first callback only latches a block (:121-129)
            if (s.cur.rate != m->rate) {                                       // blocks at another rate are discarded (:131-149)
                drop();
                while (take()) { if (s.cur.rate == m->rate) break; drop(); }
                s.ptr = 0;
                if (!s.have_cur) continue;
            }
            if (s.cur.channels == 0 || s.cur.n == 0) {                         // an empty block is replaced, next callback mixes (:152-164)
                if (!s.queue.empty()) { s.ptr = 0; drop(); (void)take(); }
                continue;
            }
            // this source takes part in buffer k
            const int me = slot_in_buffer++;
            refs.push_back(MixPeakRef{s.peaks.p + s.cur.peak_idx, s.gain, me});
            const bool mono = s.cur.channels == 1;
            const int total = mono ? frames : s.cur.channels * frames;         // loop trips (:169, :196)
            int i = 0;
            while (i < total) {
                if (s.ptr >= (size_t)s.cur.n) {                                 // the current block is used up (:171-187, :198-213)
                    s.ptr = 0;
                    drop();
                    if (!take()) break;                                         // nothing queued: the rest of the buffer gets nothing, currentInput stays null
                    refs.push_back(MixPeakRef{s.peaks.p + s.cur.peak_idx, s.gain, me});
                }
                const int run = (int)std::min<size_t>((size_t)(total - i), (size_t)s.cur.n - s.ptr);
                if (run > 0) {
                    MixPiece pc{};
                    pc.ring = s.ring.p; pc.ring_mask = s.mask; pc.ring_pos = (s.cur.ring_pos + (uint32_t)s.ptr) & s.mask;
                    pc.mono = mono ? 1 : 0; pc.gain = s.gain;
                    pc.out_begin = mono ? 2 * i : i; pc.out_end = mono ? 2 * (i + run) : i + run;
                    if (pc.out_end > 2 * frames) pc.out_end = 2 * frames;       // (channels > 2 would run past the buffer in the reference; clipped here)
                    if (pc.out_begin < pc.out_end) pieces.push_back(pc);
                    s.ptr += (size_t)run; i += run;
                } else { s.ptr++; i++; }                                        // zero-length current block: the pointer still advances (:193, :218)
            }
        }
        mb.piece1 = (int)pieces.size(); mb.ref1 = (int)refs.size(); mb.n_sources = slot_in_buffer;
        bufs[(size_t)k] = mb;
    }
    hipStream_t st = c->lanes[LANE_AUDIO];
    if (int rc = m->bufs_d.reserve(bufs.size())) return rc;
    if (int rc = m->pieces_d.reserve(std::max<size_t>(pieces.size(), 1))) return rc;
    if (int rc = m->refs_d.reserve(std::max<size_t>(refs.size(), 1))) return rc;
    if (int rc = m->out.reserve((size_t)n_buffers * 2 * frames)) return rc;
    if (int rc = m->out_peak.reserve((size_t)n_buffers)) return rc;
    CSDR_HIP_TRY(hipMemcpyAsync(m->bufs_d.p, bufs.data(), bufs.size() * sizeof(MixBuffer), hipMemcpyHostToDevice, st));
    if (!pieces.empty()) CSDR_HIP_TRY(hipMemcpyAsync(m->pieces_d.p, pieces.data(), pieces.size() * sizeof(MixPiece), hipMemcpyHostToDevice, st));
    if (!refs.empty()) CSDR_HIP_TRY(hipMemcpyAsync(m->refs_d.p, refs.data(), refs.size() * sizeof(MixPeakRef), hipMemcpyHostToDevice, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    CSDR_LAUNCH(c, LANE_AUDIO, KID_MIX, audio_mix, dim3((unsigned)n_buffers), dim3(kMixThreads), (size_t)kMixMaxSources * sizeof(double) + 16, m->bufs_d.p, m->pieces_d.p, m->refs_d.p,
                frames, m->out.p, m->out_peak.p);
    CSDR_HIP_TRY(hipGetLastError());
    for (size_t j = 0; j < m->src.size(); ++j) m->src[j].buffered -= released[j];
    m->last_frames = frames; m->last_buffers = n_buffers;
    if (out_host) {
        CSDR_HIP_TRY(hipMemcpyAsync(out_host, m->out.p, (size_t)n_buffers * 2 * frames * sizeof(float), hipMemcpyDeviceToHost, st));
        CSDR_HIP_TRY(hipStreamSynchronize(st));
    }
    return CSDR_OK;
}
// the last render as 16-bit PCM, each buffer scaled by the WAV writer's anti-clipping rule with `peak` = that buffer's summed peak
// when per_buffer_peak != 0, else with `peak` (AudioFileWAV.cpp:136)
extern "C" int csdr_mix_fetch_pcm16(csdr_mix *m, int16_t *out_host, int cap_samples, float peak, int per_buffer_peak, int *n) {
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m || !out_host || !n) return fail(CSDR_EINVAL, "bad argument");
    const int per = 2 * m->last_frames, total = per * m->last_buffers;
    if (total > cap_samples) return fail(CSDR_ERANGE, "need %d samples", total);
    *n = total;
    if (!total) return CSDR_OK;
    csdr_ctx *c = m->ctx;
    hipStream_t st = c->lanes[LANE_AUDIO];
    if (int rc = m->pcm.reserve((size_t)total)) return rc;
    if (int rc = m->pcm_d.reserve((size_t)m->last_buffers)) return rc;
    if (int rc = m->fixed_peak.reserve(1)) return rc;
    std::vector<PcmJob> jobs((size_t)m->last_buffers);
    float *fixed = m->fixed_peak.p;
    if (!per_buffer_peak) CSDR_HIP_TRY(hipMemcpyAsync(fixed, &peak, sizeof(float), hipMemcpyHostToDevice, st));
    for (int k = 0; k < m->last_buffers; ++k) jobs[(size_t)k] = PcmJob{m->out.p + (size_t)k * per, m->pcm.p + (size_t)k * per, per, 0, per_buffer_peak ? m->out_peak.p + k : fixed};
    CSDR_HIP_TRY(hipMemcpyAsync(m->pcm_d.p, jobs.data(), jobs.size() * sizeof(PcmJob), hipMemcpyHostToDevice, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    CSDR_LAUNCH(c, LANE_AUDIO, KID_MIX, pcm16_convert, dim3((unsigned)std::max(1, std::min(16, (per + 255) / 256)), (unsigned)m->last_buffers), dim3(256), 0, m->pcm_d.p);
    CSDR_HIP_TRY(hipGetLastError());
    CSDR_HIP_TRY(hipMemcpyAsync(out_host, m->pcm.p, (size_t)total * sizeof(int16_t), hipMemcpyDeviceToHost, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    return CSDR_OK;
}

// 16-bit PCM of a demodulator's audio of the last execute: every block converted with ITS OWN peak (each block is one
// AudioThreadInput for AudioFileWAV::writePayloadToFileStream), on the device; half the bytes of the float audio cross the link.
extern "C" int csdr_bank_fetch_pcm16(csdr_bank *b, int slot, int16_t *out_host, int cap_samples, int *n) {
    DeviceScope dev__(b ? b->ctx : nullptr);
    if (!b || !out_host || !n || slot < 0 || slot >= b->max_demods) return fail(CSDR_EINVAL, "bad argument");
    SlotHost &s = b->slots[(size_t)slot];
    if (s.last_A > cap_samples) return fail(CSDR_ERANGE, "need room for %d samples", s.last_A);
    *n = s.last_A;
    if (!s.last_A || s.results.empty()) return CSDR_OK;
    csdr_ctx *c = b->ctx;
    hipStream_t st = c->lanes[LANE_AUDIO];
    if (int rc = b->pcm.reserve((size_t)s.cfg.cap_audio)) return rc;
    if (int rc = b->pcm_jobs.reserve((size_t)b->max_blocks)) return rc;
    std::vector<PcmJob> jobs;
    int max_n = 1;
    for (size_t k = 0; k < s.results.size(); ++k) {
        const csdr_block_result &r = s.results[k];
        if (r.n_audio <= 0) continue;
        jobs.push_back(PcmJob{s.cfg.audio + r.audio_offset, b->pcm.p + r.audio_offset, r.n_audio, 0, &s.cfg.bout[k].audio_peak});
        max_n = std::max(max_n, r.n_audio);
    }
    if (!jobs.empty()) {
        CSDR_HIP_TRY(hipMemcpyAsync(b->pcm_jobs.p, jobs.data(), jobs.size() * sizeof(PcmJob), hipMemcpyHostToDevice, st));
        CSDR_HIP_TRY(hipStreamSynchronize(st));
        CSDR_LAUNCH(c, LANE_AUDIO, KID_MIX, pcm16_convert, dim3((unsigned)std::min(16, (max_n + 255) / 256), (unsigned)jobs.size()), dim3(256), 0, b->pcm_jobs.p);
        CSDR_HIP_TRY(hipGetLastError());
    }
    CSDR_HIP_TRY(hipMemcpyAsync(out_host, b->pcm.p, (size_t)s.last_A * sizeof(int16_t), hipMemcpyDeviceToHost, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    return CSDR_OK;
}

// =================================================================================================== ingest
// A ring of page-locked host slots the SDR reader fills in place (SoapySDRThread.cpp:221-225 takes its blocks from a buffer pool)
// and their HBM twins.  commit() moves a slot over the link ONCE; csdr_post_execute, csdr_spec_process and any other consumer read
// the returned device pointer (iq_is_dev = 1) -- the reference likewise hands ONE buffer to the demodulator and visual queues
// (SDRPostThread.cpp:227-245).  The transfer runs on its own stream: slot k + 1 crosses the link while slot k is processed.
struct csdr_ingest {
    csdr_ctx *ctx = nullptr;
    int depth = 0, next = 0, acquired = -1;
    int64_t cap = 0;
    hipStream_t copy = nullptr;
    std::vector<float2 *> host, dev;
    std::vector<hipEvent_t> ev_copied;                                   // slot's transfer finished
    std::vector<std::vector<hipEvent_t>> ev_done;                        // [slot][physical stream]: its consumers enqueued up to the next commit
    std::vector<char> copied_valid, done_valid;
};
extern "C" void csdr_ingest_destroy(csdr_ingest *g);
extern "C" int csdr_ingest_create(csdr_ctx *ctx, int64_t max_samples, int depth, csdr_ingest **out) {
    DeviceScope dev__(ctx);
    if (!ctx || !out || max_samples <= 0 || depth < 2 || depth > 16) return fail(CSDR_EINVAL, "bad argument (depth 2..16)");
    csdr_ingest *g = new csdr_ingest();
    g->ctx = ctx; g->depth = depth; g->cap = max_samples;
    g->host.assign((size_t)depth, nullptr); g->dev.assign((size_t)depth, nullptr);
    g->ev_copied.assign((size_t)depth, nullptr); g->ev_done.assign((size_t)depth, std::vector<hipEvent_t>());
    g->copied_valid.assign((size_t)depth, 0); g->done_valid.assign((size_t)depth, 0);
    // a failure anywhere below tears down what was built so far (pinned slots are large: nothing may leak under memory pressure)
    auto build = [&]() -> int {
        CSDR_HIP_TRY(hipStreamCreateWithFlags(&g->copy, hipStreamNonBlocking));
        for (int k = 0; k < depth; ++k) {
            if (hipHostMalloc((void **)&g->host[(size_t)k], (size_t)max_samples * sizeof(float2), hipHostMallocDefault) != hipSuccess) return fail(CSDR_ENOMEM, "pinned ingest slot of %lld samples", (long long)max_samples);
            if (hipMalloc((void **)&g->dev[(size_t)k], (size_t)max_samples * sizeof(float2)) != hipSuccess) return fail(CSDR_ENOMEM, "device ingest slot");
            CSDR_HIP_TRY(hipEventCreateWithFlags(&g->ev_copied[(size_t)k], hipEventDisableTiming));
            g->ev_done[(size_t)k].assign((size_t)ctx->n_phys + 1, nullptr);
            for (auto &e : g->ev_done[(size_t)k]) CSDR_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        return CSDR_OK;
    };
    if (int rc = build()) { const std::string why = last_error_ref(); csdr_ingest_destroy(g); last_error_ref() = why; return rc; }
    *out = g;
    return CSDR_OK;
}
extern "C" void csdr_ingest_destroy(csdr_ingest *g) {
    DeviceScope dev__(g ? g->ctx : nullptr);
    if (!g) return;
    (void)g->ctx->sync_all();
    if (g->copy) { (void)hipStreamSynchronize(g->copy); (void)hipStreamDestroy(g->copy); }
    for (int k = 0; k < g->depth; ++k) {
        if (g->host[(size_t)k]) (void)hipHostFree(g->host[(size_t)k]);
        if (g->dev[(size_t)k]) (void)hipFree(g->dev[(size_t)k]);
        if (g->ev_copied[(size_t)k]) (void)hipEventDestroy(g->ev_copied[(size_t)k]);
        for (auto &e : g->ev_done[(size_t)k]) if (e) (void)hipEventDestroy(e);
    }
    delete g;
}
// the page-locked slot the next batch is assembled in; waits until the transfer that last read this slot is over
extern "C" int csdr_ingest_acquire(csdr_ingest *g, float **host_slot) {
    DeviceScope dev__(g ? g->ctx : nullptr);
    if (!g || !host_slot) return fail(CSDR_EINVAL, "bad argument");
    const int k = g->next;
    if (g->copied_valid[(size_t)k]) CSDR_HIP_TRY(hipEventSynchronize(g->ev_copied[(size_t)k]));
    g->acquired = k;
    *host_slot = (float *)g->host[(size_t)k];
    return CSDR_OK;
}
// the transfer of slot k from `src` (its own page-locked twin, or caller memory), ordered against the slot's previous consumers
static int ingest_transfer(csdr_ingest *g, int k, const float2 *src, bool src_is_slot, int64_t n_samples, int iq_swap, const float **dev_iq) {
    csdr_ctx *c = g->ctx;
    const int prev = (k + g->depth - 1) % g->depth;
    // everything enqueued so far may still read the PREVIOUS slot's device copy: mark it (one event per stream that can hold consumers)
    if (g->copied_valid[(size_t)prev]) {
        for (int l = 0; l < c->n_phys; ++l) CSDR_HIP_TRY(hipEventRecord(g->ev_done[(size_t)prev][(size_t)l], c->phys[l]));
        CSDR_HIP_TRY(hipEventRecord(g->ev_done[(size_t)prev][(size_t)c->n_phys], c->stream));
        g->done_valid[(size_t)prev] = 1;
    }
    // the device twin of THIS slot was last read by the consumers of `depth` transfers ago
    if (g->done_valid[(size_t)k]) for (auto &e : g->ev_done[(size_t)k]) CSDR_HIP_TRY(hipStreamWaitEvent(g->copy, e, 0));
    const int grid = std::max(1, std::min(4 * c->n_cu, (int)((n_samples + 255) / 256)));
    if (iq_swap && src_is_slot) {
        // the slot is mapped into the device's address space: the exchanging kernel IS the transfer
        hipLaunchKernelGGL(ingest_swap, dim3((unsigned)grid), dim3(256), 0, g->copy, src, g->dev[(size_t)k], n_samples);
        CSDR_HIP_TRY(hipGetLastError());
    } else {
        CSDR_HIP_TRY(hipMemcpyAsync(g->dev[(size_t)k], src, (size_t)n_samples * sizeof(float2), hipMemcpyHostToDevice, g->copy));
        if (iq_swap) {                                                                // caller memory may not be device-visible: exchange in HBM behind the DMA
            hipLaunchKernelGGL(ingest_swap, dim3((unsigned)grid), dim3(256), 0, g->copy, (const float2 *)g->dev[(size_t)k], g->dev[(size_t)k], n_samples);
            CSDR_HIP_TRY(hipGetLastError());
        }
    }
    CSDR_HIP_TRY(hipEventRecord(g->ev_copied[(size_t)k], g->copy));
    g->copied_valid[(size_t)k] = 1;
    // consumers on any of the library's streams (and on the boundary stream) start behind the transfer
    for (int l = 0; l < c->n_phys; ++l) CSDR_HIP_TRY(hipStreamWaitEvent(c->phys[l], g->ev_copied[(size_t)k], 0));
    if (!c->own_stream) CSDR_HIP_TRY(hipStreamWaitEvent(c->stream, g->ev_copied[(size_t)k], 0));
    *dev_iq = (const float *)g->dev[(size_t)k];
    g->next = (k + 1) % g->depth;
    return CSDR_OK;
}
extern "C" int csdr_ingest_commit(csdr_ingest *g, int64_t n_samples, int iq_swap, const float **dev_iq) {
    DeviceScope dev__(g ? g->ctx : nullptr);
    if (!g || !dev_iq) return fail(CSDR_EINVAL, "bad argument");
    if (g->acquired < 0) return fail(CSDR_ESTATE, "commit without acquire");
    if (n_samples <= 0 || n_samples > g->cap) return fail(CSDR_ERANGE, "%lld samples (slot holds %lld)", (long long)n_samples, (long long)g->cap);
    const int k = g->acquired;
    g->acquired = -1;
    return ingest_transfer(g, k, g->host[(size_t)k], true, n_samples, iq_swap, dev_iq);
}
// blocks until the last transfer has left its source buffer (a caller that is about to rewrite or recycle the block it just uploaded)
extern "C" int csdr_ingest_wait(csdr_ingest *g) {
    DeviceScope dev__(g ? g->ctx : nullptr);
    if (!g) return fail(CSDR_EINVAL, "ingest is null");
    const int last = (g->next + g->depth - 1) % g->depth;
    if (g->copied_valid[(size_t)last]) CSDR_HIP_TRY(hipEventSynchronize(g->ev_copied[(size_t)last]));
    return CSDR_OK;
}
// the slot the next commit / upload will use (a caller that ties a slot's lifetime to its own block objects asks before it transfers)
extern "C" int csdr_ingest_next_slot(const csdr_ingest *g) { return g ? g->next : -1; }
// One transfer of a block the caller assembled in ITS OWN memory (the pooled SDRThreadIQData blocks; page-lock them once with
// csdr_host_register and the transfer is a DMA).  The previous upload is waited for first, so at most one is in flight and a caller that
// alternates between at least two buffers never rewrites one that is still being read.
extern "C" int csdr_ingest_upload(csdr_ingest *g, const float *host_iq, int64_t n_samples, int iq_swap, const float **dev_iq) {
    RangeScope range__("csdr_ingest_upload");
    DeviceScope dev__(g ? g->ctx : nullptr);
    if (!g || !host_iq || !dev_iq) return fail(CSDR_EINVAL, "bad argument");
    if (n_samples <= 0 || n_samples > g->cap) return fail(CSDR_ERANGE, "%lld samples (slot holds %lld)", (long long)n_samples, (long long)g->cap);
    const int k = g->next, prev = (k + g->depth - 1) % g->depth;
    if (g->copied_valid[(size_t)prev]) CSDR_HIP_TRY(hipEventSynchronize(g->ev_copied[(size_t)prev]));
    return ingest_transfer(g, k, (const float2 *)host_iq, false, n_samples, iq_swap, dev_iq);
}
