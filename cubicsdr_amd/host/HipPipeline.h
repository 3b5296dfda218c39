// HipPipeline.h -- the reference's hot-path objects re-hosted on the HIP library (include/csdr_hip.h).
//
//   DemodulatorInstance / DemodulatorMgr : setters/getters of src/demod/DemodulatorInstance.h:36-136 and
//       DemodulatorMgr.cpp:35-48; each instance owns one slot of a csdr_bank instead of three threads.
//   SDRPostThread : IOThread with the queue names of src/sdr/SDRPostThread.cpp:162-165; per input block it runs the
//       channelizer + every active demodulator on the GPU, then applies DemodulatorThread::run's host-side state
//       machine (level / floor / ceil / squelch, DemodulatorThread.cpp:142-220) and try_pushes the audio
//       (:318-328) -- drop-on-full exactly where the reference drops.
//   SpectrumVisualProcessor : VisualProcessor<DemodulatorThreadIQData, SpectrumVisualData> with the setters of
//       src/process/SpectrumVisualProcessor.h:29-58 (full-span view, peak hold, DC hiding, short-input overlap rule).
//   FFTVisualDataThread / SpectrumVisualDataThread : the 10 ms pump threads around it (FFTDataDistributor.h holds the
//       waterfall line pacing).
#pragma once
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/csdr_hip.h"
#include "Adapters.h"
#include "DataTypes.h"
#include "DemodLevel.h"
#include "IOThread.h"
#include "Modem.h"
#include "VisualProcessor.h"

#define HEARTBEAT_CHECK_PERIOD_MICROS (50 * 1000)
#define DEMOD_VIS_SIZE 2048                      // DemodulatorThread.h:15
#include "FFTDataDistributor.h"

// Construction-time failures (no device, no memory) throw: the object cannot exist.
inline void csdr_must(int rc, const char *what) {
    if (rc != CSDR_OK) throw std::runtime_error(std::string(what) + ": " + csdr_strerror(rc) + " (" + csdr_last_error() + ")");
}
// Failures INSIDE a running stage never leave run() / process() as exceptions (IOThread::threadMain would mark the thread
// terminated and rethrow into the GUI, IOThread.cpp:44-51): the block is dropped, the failure counted and its text kept for the
// owner to poll, and the stage goes on with the next block -- the reference's stages have no error path at all.
class CsdrErrorLog {
public:
    bool ok(int rc, const char *what) {
        if (rc == CSDR_OK) return true;
        errors_.fetch_add(1);
        std::lock_guard<std::mutex> g(mu_);
        last_ = std::string(what) + ": " + csdr_strerror(rc) + " (" + csdr_last_error() + ")";
        return false;
    }
    long long errorCount() const { return errors_.load(); }
    std::string lastError() { std::lock_guard<std::mutex> g(mu_); return last_; }
private:
    std::atomic<long long> errors_{0};
    std::mutex mu_;
    std::string last_;
};
#define CSDR_STAGE_TRY(expr, what) do { if (!errlog.ok((expr), (what))) return; } while (0)

class DemodulatorMgr;

typedef ThreadBlockingQueue<AudioThreadInputPtr> DemodulatorThreadOutputQueue;           // DemodDefs.h: the scope ("audio visual") queue
typedef std::shared_ptr<DemodulatorThreadOutputQueue> DemodulatorThreadOutputQueuePtr;

class DemodulatorInstance {
public:
    DemodulatorInstance(DemodulatorMgr *mgr, int slot) : mgr_(mgr), slot_(slot) {
        audioQueue_ = std::make_shared<AudioThreadInputQueue>();
        audioQueue_->set_max_num_items(100);
        pipeIQInputData_ = std::make_shared<DemodulatorThreadInputQueue>();                 // DemodulatorInstance.cpp:54-56
        pipeIQInputData_->set_max_num_items(100);
        setDemodulatorType("NBFM");
    }
    // --- lifecycle (DemodulatorInstance.cpp:111-196): the reference starts a pre-demod, a demod and an audio thread per instance;
    // here the arithmetic of all instances runs inside SDRPostThread's csdr_bank_execute, so run() only marks the instance live
    ~DemodulatorInstance() { if (kit_ && modem_) modem_->disposeKit(kit_); }
    void run() { terminated_.store(false); active_.store(true); }
    void terminate() { active_.store(false); terminated_.store(true); if (pipeIQInputData_) pipeIQInputData_->flush(); audioQueue_->flush(); }
    bool isTerminated() { return terminated_.load(); }
    std::string getLabel() { std::lock_guard<std::mutex> g(mu_); return label_; }
    void setLabel(std::string l) { std::lock_guard<std::mutex> g(mu_); label_ = std::move(l); }
    // --- configuration (applied at the next block, like the atomics of DemodulatorPreThread.cpp:293-336)
    void setDemodulatorType(const std::string &t) {                                          // DemodulatorInstance.cpp:318-333
        std::unique_ptr<Modem> m(Modem::makeModem(t));
        if (!m) return;                                                                      // unknown type: keep the current modem
        std::lock_guard<std::mutex> g(mu_);
        if (kit_ && modem_) { modem_->disposeKit(kit_); kit_ = nullptr; }                   // DemodulatorThread.cpp:98-108
        type_ = t; bandwidth_ = Modem::getModemDefaultSampleRate(t); modem_ = std::move(m); dirty_ = true;
        auto it = lastModemSettings_.find(t);
        if (it != lastModemSettings_.end()) modem_->writeSettings(it->second);
    }
    std::string getDemodulatorType() { std::lock_guard<std::mutex> g(mu_); return type_; }
    std::string getModemType() { std::lock_guard<std::mutex> g(mu_); return modem_ ? modem_->getType() : ""; }
    bool isModemInitialized() { std::lock_guard<std::mutex> g(mu_); return modem_ != nullptr; }
    void setBandwidth(int bw) {                                                              // through the modem's checkSampleRate (DemodulatorPreThread.cpp:98-104)
        std::lock_guard<std::mutex> g(mu_);
        bandwidth_ = modem_ ? modem_->checkSampleRate(bw, audioRate_) : bw; dirty_ = true;
    }
    int getBandwidth() { std::lock_guard<std::mutex> g(mu_); return bandwidth_; }
    void setFrequency(long long f) { frequency_.store(f); }
    long long getFrequency() { return frequency_.load(); }
    void setAudioSampleRate(int r) { std::lock_guard<std::mutex> g(mu_); audioRate_ = r; dirty_ = true; }
    int getAudioSampleRate() { std::lock_guard<std::mutex> g(mu_); return audioRate_; }
    void setGain(float g) { gain_.store(g < 0.005f ? 0.005f : (g > 40.0f ? 40.0f : g)); }   // AudioThread::setGain clamps (AudioThread.cpp:523-531)
    float getGain() { return gain_.load(); }
    void setActive(bool a) { active_.store(a); }
    bool isActive() { return active_.load(); }
    void setSquelchEnabled(bool e) { squelchEnabled_.store(e); }
    bool isSquelchEnabled() { return squelchEnabled_.load(); }
    void setSquelchLevel(float l) { if (!squelchEnabled_.load()) squelchEnabled_.store(true); squelchLevel_.store(l); }   // setting a level switches the squelch on (DemodulatorThread.cpp:392-397)
    float getSquelchLevel() { return squelchLevel_.load(); }
    void setMuted(bool m) { muted_.store(m); }
    bool isMuted() { return muted_.load(); }
    float getSignalLevel() { return signalLevel_.load(); }
    float getSignalFloor() { return signalFloor_.load(); }
    float getSignalCeil() { return signalCeil_.load(); }
    AudioThreadInputQueuePtr getAudioOutputQueue() { return audioQueue_; }   // "AudioDataOutput" (DemodulatorThread.cpp:80)
    // the block hand-over pipe of the reference (DemodulatorInstance.cpp:447-449).  SDRPostThread does not push blocks through it
    // (the channelizer output stays on the device); it exists so that code holding the pipe keeps linking, and is flushed on terminate
    DemodulatorThreadInputQueuePtr getIQInputDataPipe() { return pipeIQInputData_; }
    // the audio-scope queue (DemodulatorInstance.cpp:103-105 -> DemodulatorThread::setOutputQueue("AudioVisualOutput"))
    void setVisualOutputQueue(const DemodulatorThreadOutputQueuePtr &q) { std::lock_guard<std::mutex> g(mu_); audioVisQueue_ = q; }
    // the sound output of this demodulator (the reference owns an AudioThread bound to the device controller, DemodulatorInstance.cpp:60-66):
    // with a source set AND SDRPostThread::setAudioMixer, the block's audio goes from the bank to the mixer's ring inside HBM
    void setAudioMixSource(const std::shared_ptr<AudioMixSource> &s) { std::lock_guard<std::mutex> g(mu_); mixSource_ = s; }
    std::shared_ptr<AudioMixSource> getAudioMixSource() { std::lock_guard<std::mutex> g(mu_); return mixSource_; }
    // modem settings (DemodulatorInstance.cpp:451-498)
    ModemArgInfoList getModemArgs() { std::lock_guard<std::mutex> g(mu_); return modem_ ? modem_->getSettings() : ModemArgInfoList(); }
    std::string readModemSetting(const std::string &setting) { std::lock_guard<std::mutex> g(mu_); return modem_ ? modem_->readSetting(setting) : ""; }
    ModemSettings readModemSettings() { std::lock_guard<std::mutex> g(mu_); return modem_ ? modem_->readSettings() : ModemSettings(); }
    void writeModemSetting(const std::string &setting, std::string value) {
        std::lock_guard<std::mutex> g(mu_);
        if (!modem_) return;
        modem_->writeSetting(setting, value);
        lastModemSettings_[type_][setting] = value;
        if (modem_->shouldRebuildKit()) { dirty_ = true; modem_->clearRebuildKit(); }
    }
    void writeModemSettings(ModemSettings settings) { for (auto &kv : settings) writeModemSetting(kv.first, kv.second); }
    ModemSettings getLastModemSettings(const std::string &demodType) { std::lock_guard<std::mutex> g(mu_); return lastModemSettings_[demodType]; }
    int slot() const { return slot_; }

private:
    friend class SDRPostThread;
    DemodulatorMgr *mgr_;
    int slot_;
    std::mutex mu_;
    std::string type_ = "NBFM", label_;
    std::unique_ptr<Modem> modem_;
    ModemKit *kit_ = nullptr;                 // host plug-in modems (CSDR_MODEM_HOST): the kit their buildKit() made, disposed by the same modem
    std::map<std::string, ModemSettings> lastModemSettings_;
    int bandwidth_ = 12500, audioRate_ = 48000;
    bool dirty_ = true;                       // needs csdr_bank_configure_slot
    long long builtRate_ = 0;
    std::atomic<long long> frequency_{0};
    std::atomic_bool active_{false}, terminated_{false}, squelchEnabled_{false}, muted_{false};
    std::atomic<float> squelchLevel_{-100.0f}, signalLevel_{-100.0f}, signalFloor_{-30.0f}, signalCeil_{30.0f}, gain_{1.0f};
    bool squelchBreak_ = false;
    AudioThreadInputQueuePtr audioQueue_;
    DemodulatorThreadInputQueuePtr pipeIQInputData_;
    DemodulatorThreadOutputQueuePtr audioVisQueue_;
    std::shared_ptr<AudioMixSource> mixSource_;
    ReBuffer<AudioThreadInput> outputBuffers_{"DemodulatorThreadBuffers"};
};
typedef std::shared_ptr<DemodulatorInstance> DemodulatorInstancePtr;

class DemodulatorMgr {
public:
    explicit DemodulatorMgr(int maxDemods = 256) : max_(maxDemods) {}
    DemodulatorInstancePtr newThread() {                     // DemodulatorMgr.cpp:35-48
        std::lock_guard<std::recursive_mutex> g(mu_);
        if ((int)demods_.size() >= max_) throw std::runtime_error("DemodulatorMgr: bank is full");
        int slot = 0;
        while (used_.count(slot)) ++slot;
        used_[slot] = true;
        auto d = std::make_shared<DemodulatorInstance>(this, slot);
        demods_.push_back(d);
        return d;
    }
    std::vector<DemodulatorInstancePtr> getDemodulators() { std::lock_guard<std::recursive_mutex> g(mu_); return demods_; }
    void deleteThread(const DemodulatorInstancePtr &d) {
        std::lock_guard<std::recursive_mutex> g(mu_);
        for (auto it = demods_.begin(); it != demods_.end(); ++it)
            if (*it == d) { used_.erase(d->slot()); demods_.erase(it); break; }
    }
    int capacity() const { return max_; }
    // the selected ("current") modem: its channel is what SDRPostThread taps for the demodulator spectrum view
    // (DemodulatorMgr.cpp:207-250, :285-290; runDemodChannels :304, :334, :383-387)
    void setActiveDemodulator(const DemodulatorInstancePtr &d, bool temporary = true) {
        std::lock_guard<std::recursive_mutex> g(mu_);
        if (!temporary) currentModem_ = d;
    }
    DemodulatorInstancePtr getCurrentModem() { std::lock_guard<std::recursive_mutex> g(mu_); return currentModem_; }

private:
    std::recursive_mutex mu_;
    DemodulatorInstancePtr currentModem_;
    std::vector<DemodulatorInstancePtr> demods_;
    std::map<int, bool> used_;
    int max_;
};

class SDRPostThread : public IOThread {
public:
    SDRPostThread(csdr_ctx *ctx, DemodulatorMgr *mgr) : ctx_(ctx), mgr_(mgr) {
        csdr_must(csdr_post_create(ctx_, &post_), "csdr_post_create");
        csdr_must(csdr_bank_create(ctx_, mgr->capacity(), 1, &bank_), "csdr_bank_create");
    }
    ~SDRPostThread() override { if (bank_) csdr_bank_destroy(bank_); if (post_) csdr_post_destroy(post_); }

    void run() override {
        auto iqIn = std::static_pointer_cast<SDRThreadIQDataQueue>(getInputQueue("IQDataInput"));
        auto iqOut = std::static_pointer_cast<DemodulatorThreadInputQueue>(getOutputQueue("IQDataOutput"));
        auto iqVisual = std::static_pointer_cast<DemodulatorThreadInputQueue>(getOutputQueue("IQVisualDataOutput"));
        if (!iqIn) throw std::runtime_error("SDRPostThread: IQDataInput is not bound");
        while (!stopping) {
            SDRThreadIQDataPtr in;
            if (!iqIn->pop(in, HEARTBEAT_CHECK_PERIOD_MICROS)) continue;      // SDRPostThread.cpp:170
            if (!in || in->data.empty()) continue;
            processBlock(*in, iqOut, iqVisual);
            ++blocksProcessed;
        }
        if (iqVisual) iqVisual->flush();
        iqIn->flush();
        if (iqOut) iqOut->flush();
    }
    void terminate() override {
        IOThread::terminate();
        auto iqIn = std::static_pointer_cast<SDRThreadIQDataQueue>(getInputQueue("IQDataInput"));
        if (iqIn) iqIn->flush();
    }
    std::atomic<long long> blocksProcessed{0};
    CsdrErrorLog errlog;                                       // failures of the device library inside run(): counted, never thrown

    // SDRPostThreadChannelizerType (SDRPostThread.h:9-12, setChannelizerType :142-149): takes effect at the next block,
    // which rebuilds the channelizer (chanMode != lastChanMode, :418 / :474)
    enum SDRPostThreadChannelizerType { SDRPostPFBCH = 1, SDRPostPFBCH2 = 2 };
    void setChannelizerType(SDRPostThreadChannelizerType t) { chanMode.store((int)t); }
    // sound output on the device: demodulators that carry an AudioMixSource hand their audio to this mixer without a host round trip
    void setAudioMixer(AudioMixer *m) { mixer_.store(m); }
    SDRPostThreadChannelizerType getChannelizerType() { return (SDRPostThreadChannelizerType)chanMode.load(); }

private:
    void processBlock(SDRThreadIQData &in, const DemodulatorThreadInputQueuePtr &iqOut, const DemodulatorThreadInputQueuePtr &iqVisual) {
        const int M = in.numChannels > 1 ? in.numChannels : 1;
        // whole frames only (the reference's channelizer loop steps by numChannels, :449; SoapySDRThread hands out multiples, :668-674)
        const int n = ((int)in.data.size() / M) * M;
        if (n <= 0) return;
        const int mode = chanMode.load();
        if (in.sampleRate != sampleRate_ || M != numChannels_ || n > maxBlock_ || mode != lastChanMode_) {      // initPFBCH :401-414, initPFBCH2 :458-470
            // room for blocks up to twice the nominal 1/60 s (a longer one re-initialises, which also rebuilds every demodulator:
            // their buffers are sized from the block length)
            const long long nominal = ((in.sampleRate / 30 + M - 1) / M) * M;
            sampleRate_ = in.sampleRate; numChannels_ = M; lastChanMode_ = mode;
            maxBlock_ = (int)std::max<long long>(n, std::min<long long>(nominal, 1LL << 26));
            const int kind = M > 1 ? (mode == SDRPostPFBCH2 ? CSDR_POST_PFBCH2 : CSDR_POST_PFBCH) : CSDR_POST_SINGLE;
            CSDR_STAGE_TRY(csdr_post_configure(post_, sampleRate_, M, kind, maxBlock_, 1), "csdr_post_configure");
            for (auto &d : mgr_->getDemodulators()) { std::lock_guard<std::mutex> g(d->mu_); d->dirty_ = true; }
        }
        if (M == 1) {
            // runSingleCH (:248-299): the DC blocker runs on EVERY block; the DC-corrected data is what the main spectrum, the
            // waterfall and (when a demodulator is active) the demodulator spectrum see
            CSDR_STAGE_TRY(csdr_post_execute(post_, (const float *)in.data.data(), 0, 1, n, in.frequency), "csdr_post_execute");
            // the corrected block comes back to the host only for a consumer that is bound (a D2H copy + stream wait per block
            // otherwise bought nothing and kept the demodulators waiting behind it); the demodulators read it on the device
            auto iqActiveQ = std::static_pointer_cast<DemodulatorThreadInputQueue>(getOutputQueue("IQActiveDemodVisualDataOutput"));
            singleOut_.reset();
            if (iqOut || iqActiveQ) {
                singleOut_ = visualBuffers_.getBuffer();
                singleOut_->frequency = in.frequency; singleOut_->sampleRate = in.sampleRate;
                singleOut_->data.resize((size_t)n);
                int got = 0;
                CSDR_STAGE_TRY(csdr_post_read_channel(post_, 0, (float *)singleOut_->data.data(), n, &got), "csdr_post_read_channel");
                singleOut_->data.resize((size_t)got);
                if (iqOut) { iqOut->try_push(singleOut_); if (iqVisual) iqVisual->try_push(singleOut_); }   // pushVisualData :233-245
            }
        } else if (iqOut) {
            // full-rate copy to the visual queues first (getFullSampleRateIqData + pushVisualData, :221-245): never blocks
            DemodulatorThreadIQDataPtr vis = visualBuffers_.getBuffer();
            vis->frequency = in.frequency; vis->sampleRate = in.sampleRate; vis->data = in.data;
            vis->shareDeviceCopy(in);                                               // the spectrum reads the block where the ingest put it
            iqOut->try_push(vis);
            if (iqVisual) iqVisual->try_push(vis);
        }
        // active set: in range of this block's span (updateActiveDemodulators, :44-98)
        auto demods = mgr_->getDemodulators();
        const long long chanRate = csdr_post_channel_rate(post_);            // chanBw, or 2 * chanBw behind firpfbch2 (:510)
        std::vector<DemodulatorInstancePtr> run;
        for (auto &d : demods) {
            const bool inRange = std::llabs(in.frequency - d->getFrequency()) <= in.sampleRate / 2;
            d->setActive(inRange);
            bool rebuild;
            csdr_demod_params p{};
            {
                std::lock_guard<std::mutex> g(d->mu_);
                rebuild = d->dirty_ || d->builtRate_ != chanRate;
                p.modem = d->modem_ ? d->modem_->csdrModemId() : -1; p.bandwidth = d->bandwidth_; p.audio_sample_rate = d->audioRate_;
                p.modem_arg = d->modem_ ? d->modem_->csdrModemArg() : 0;
                p.frequency = d->getFrequency();
                if (rebuild && inRange) {
                    d->dirty_ = false; d->builtRate_ = chanRate;
                    if (d->modem_ && p.modem == CSDR_MODEM_HOST) {                  // DemodulatorWorkerThread.cpp:63-76: a fresh kit per (re)build
                        if (d->kit_) d->modem_->disposeKit(d->kit_);
                        d->kit_ = d->modem_->buildKit(d->bandwidth_, d->audioRate_);
                    }
                }
            }
            if (!inRange) { (void)csdr_bank_set_active(bank_, d->slot(), 0); continue; }
            if (rebuild) CSDR_STAGE_TRY(csdr_bank_configure_slot(bank_, d->slot(), &p, post_), "csdr_bank_configure_slot");
            CSDR_STAGE_TRY(csdr_bank_set_frequency(bank_, d->slot(), d->getFrequency()), "csdr_bank_set_frequency");
            CSDR_STAGE_TRY(csdr_bank_set_active(bank_, d->slot(), 1), "csdr_bank_set_active");
            run.push_back(d);
        }
        if (run.empty()) return;                                                     // :436 "if (!runDemods.empty())"
        auto iqActive = std::static_pointer_cast<DemodulatorThreadInputQueue>(getOutputQueue("IQActiveDemodVisualDataOutput"));
        if (M == 1) {
            if (iqActive && singleOut_) iqActive->try_push(singleOut_);              // :289-292
        } else if (in.deviceData && in.deviceSamples >= (size_t)n)                  // already in HBM (DeviceIngest): no second transfer
            CSDR_STAGE_TRY(csdr_post_execute(post_, in.deviceData, 1, 1, n, in.frequency), "csdr_post_execute");
        else CSDR_STAGE_TRY(csdr_post_execute(post_, (const float *)in.data.data(), 0, 1, n, in.frequency), "csdr_post_execute");
        CSDR_STAGE_TRY(csdr_bank_execute(bank_, post_), "csdr_bank_execute");
        // the active demodulator's channel also feeds the demodulator spectrum (:334, :383-387)
        DemodulatorInstancePtr cur = mgr_->getCurrentModem();
        if (M > 1 && iqActive && cur && cur->isActive()) {
            const int ch = csdr_post_channel_at(post_, cur->getFrequency());
            if (ch >= 0) {
                DemodulatorThreadIQDataPtr tap = visualBuffers_.getBuffer();
                const int cnt = n / std::max(1, (int)(in.sampleRate / std::max(1LL, (long long)csdr_post_channel_rate(post_)))) + 8;
                tap->data.resize((size_t)cnt);
                int got = 0;
                CSDR_STAGE_TRY(csdr_post_read_channel(post_, ch, (float *)tap->data.data(), cnt, &got), "csdr_post_read_channel");
                tap->data.resize((size_t)got);
                tap->frequency = csdr_post_channel_center(post_, ch);
                tap->sampleRate = csdr_post_channel_rate(post_);
                iqActive->try_push(tap);                                              // never blocks (:386)
            }
        }
        mixSlots_.clear(); mixSources_.clear();
        for (auto &d : run) finishDemod(*d);
        if (!mixSlots_.empty()) { AudioMixer *mx = mixer_.load(); if (mx) (void)mx->takeBankAudio(bank_, mixSlots_, mixSources_); }
    }

    // DemodulatorThread::run after demodulate(): :142-233, :318-328
    void finishDemod(DemodulatorInstance &d) {
        csdr_block_result r;
        int nb = 0;
        CSDR_STAGE_TRY(csdr_bank_fetch_results(bank_, d.slot(), &r, 1, &nb), "csdr_bank_fetch_results");
        if (nb != 1 || r.skipped || r.n_iq == 0) return;
        AudioThreadInputPtr ati = d.outputBuffers_.getBuffer();
        ati->sampleRate = d.getAudioSampleRate(); ati->inputRate = d.getBandwidth(); ati->channels = (d.getDemodulatorType() == "I/Q" || d.getDemodulatorType() == "FMS") ? 2 : 1; ati->frequency = d.getFrequency();
        int got = 0;
        bool hostModem = false;
        std::shared_ptr<AudioMixSource> mixSource;
        DemodulatorThreadOutputQueuePtr vis;
        {
            std::lock_guard<std::mutex> g(d.mu_);
            hostModem = d.modem_ && d.kit_ && d.modem_->csdrModemId() == CSDR_MODEM_HOST;
            if (mixer_.load() && !hostModem) mixSource = d.mixSource_;
            vis = d.audioVisQueue_;
        }
        // with a device mixer the audio stays in HBM; it comes to the host only for the scope tap, when a scope is bound and waiting
        const bool audioToHost = !mixSource || (vis && vis->empty());
        if (hostModem) {
            // a plug-in modem (Modem.h:127-166): the device ran DemodulatorPreThread's arithmetic; the block's resampled IQ comes back
            // and the plug-in demodulates it here, on the thread that owns the instance, as DemodulatorThread::run does (:119-135).
            // Level and peak are formed from what it produced with the reference's statements (:142-160, :223-233).
            hostIq_.sampleRate = d.getBandwidth();
            hostIq_.data.resize((size_t)r.n_iq);
            CSDR_STAGE_TRY(csdr_bank_fetch_iq(bank_, d.slot(), (float *)hostIq_.data.data(), r.n_iq, &got), "csdr_bank_fetch_iq");
            hostIq_.data.resize((size_t)got);
            ati->channels = 1;
            ati->data.resize(0);
            bool useOut;
            {
                std::lock_guard<std::mutex> g(d.mu_);
                if (d.modem_->getType() == "digital") ati->sampleRate = (int)d.kit_->sampleRate;     // :131-136
                d.modem_->demodulate(d.kit_, &hostIq_, ati.get());
                useOut = d.modem_->useSignalOutput();
            }
            double accum = 0;
            if (!ati->data.empty()) {
                if (useOut) for (float v : ati->data) accum += std::sqrt((double)v * (double)v);
                else for (auto &x : hostIq_.data) accum += std::sqrt((double)x.real * (double)x.real + (double)x.imag * (double)x.imag);
            }
            r.level_accum = accum; r.level_count = (int)(useOut ? ati->data.size() : hostIq_.data.size());
            r.audio_peak = 0.f;
            for (float v : ati->data) r.audio_peak = std::max(r.audio_peak, std::fabs(v));
            r.n_audio = (int)ati->data.size();
        } else if (audioToHost) {
            ati->data.resize(r.n_audio);
            if (r.n_audio) CSDR_STAGE_TRY(csdr_bank_fetch_audio(bank_, d.slot(), ati->data.data(), r.n_audio, &got), "csdr_bank_fetch_audio");
        } else ati->data.resize(0);
        const double sampleTime = double(r.n_iq) / double(d.getBandwidth());
        DemodLevelState st;
        st.signalLevel = d.signalLevel_; st.signalFloor = d.signalFloor_; st.signalCeil = d.signalCeil_; st.squelchBreak = d.squelchBreak_;
        const bool squelched = demodLevelStep(st, hostModem ? !ati->data.empty() : r.n_audio > 0, r.level_accum, r.level_count, sampleTime, d.squelchEnabled_, d.squelchLevel_);
        d.signalLevel_ = st.signalLevel; d.signalFloor_ = st.signalFloor; d.signalCeil_ = st.signalCeil; d.squelchBreak_ = st.squelchBreak;
        ati->peak = r.audio_peak;
        ati->is_squelch_active = squelched;
        // the audio scope tap (:240-316): only when the scope queue is bound and empty
        if (!squelched && vis && vis->empty() && !ati->data.empty()) {
            AudioThreadInputPtr ati_vis = std::make_shared<AudioThreadInput>();
            ati_vis->sampleRate = d.getBandwidth(); ati_vis->inputRate = d.getBandwidth();       // inp->sampleRate
            size_t num_vis = DEMOD_VIS_SIZE;
            if (ati->channels == 2) {                                                             // :269-291
                ati_vis->channels = 2;
                int stereoSize = (int)ati->data.size();
                if (stereoSize > DEMOD_VIS_SIZE * 2) stereoSize = DEMOD_VIS_SIZE * 2;
                ati_vis->data.resize((size_t)stereoSize);
                if (d.getDemodulatorType() == "I/Q") {
                    // inputData = the resampled IQ of the block; the I/Q modem's audio is (imag, real) of it (ModemIQ.cpp:41-61)
                    for (int i = 0; i < stereoSize / 2; i++) {
                        ati_vis->data[i] = ati->data[2 * i + 1] * 0.75f;
                        ati_vis->data[i + stereoSize / 2] = ati->data[2 * i] * 0.75f;
                    }
                } else {
                    ati_vis->inputRate = d.getAudioSampleRate(); ati_vis->sampleRate = 36000;
                    for (int i = 0; i < stereoSize / 2; i++) { ati_vis->data[i] = ati->data[i * 2]; ati_vis->data[i + stereoSize / 2] = ati->data[i * 2 + 1]; }
                }
                ati_vis->type = 1;
            } else {                                                                              // :292-312
                const size_t numAudioWritten = ati->data.size();
                ati_vis->channels = 1;
                std::vector<float> demodOut((size_t)DEMOD_VIS_SIZE);
                int nd = 0;
                const bool haveDemodOut = csdr_bank_fetch_demod_output(bank_, d.slot(), demodOut.data(), DEMOD_VIS_SIZE, &nd) == CSDR_OK && nd > 0;
                if (numAudioWritten > (size_t)r.n_iq || !haveDemodOut) {
                    ati_vis->inputRate = d.getAudioSampleRate();
                    if (num_vis > numAudioWritten) num_vis = numAudioWritten;
                    ati_vis->data.assign(ati->data.begin(), ati->data.begin() + (long)num_vis);
                } else {
                    if (num_vis > (size_t)nd) num_vis = (size_t)nd;
                    ati_vis->data.assign(demodOut.begin(), demodOut.begin() + (long)num_vis);
                }
                ati_vis->type = 0;
            }
            (void)vis->try_push(ati_vis);                                                         // non-blocking (:314)
        }
        if (!squelched && !d.muted_) {
            if (mixSource) { mixSlots_.push_back(d.slot()); mixSources_.push_back(mixSource); }      // HBM -> the mixer's ring, after the loop
            else (void)d.audioQueue_->try_push(ati);                                   // never blocks (:322)
        }
    }

    csdr_ctx *ctx_;
    DemodulatorMgr *mgr_;
    csdr_post *post_ = nullptr;
    csdr_bank *bank_ = nullptr;
    long long sampleRate_ = 0;
    int numChannels_ = 0, maxBlock_ = 0, lastChanMode_ = 0;
    std::atomic<int> chanMode{(int)SDRPostPFBCH};                                // ctor :23
    ReBuffer<DemodulatorThreadIQData> visualBuffers_{"SDRPostThreadVisualDataBuffers"};
    DemodulatorThreadIQDataPtr singleOut_;                                        // single-channel mode: the DC-corrected block
    std::atomic<AudioMixer *> mixer_{nullptr};
    std::vector<int> mixSlots_;
    std::vector<std::shared_ptr<AudioMixSource>> mixSources_;
    ModemIQData hostIq_;                                                          // a host plug-in modem's input block (modemData, DemodulatorThread.h)
};

class SpectrumVisualProcessor : public VisualProcessor<DemodulatorThreadIQData, SpectrumVisualData> {
public:
    explicit SpectrumVisualProcessor(csdr_ctx *ctx) : ctx_(ctx) { csdr_must(csdr_spec_create(ctx_, &spec_), "csdr_spec_create"); }
    ~SpectrumVisualProcessor() override { if (spec_) csdr_spec_destroy(spec_); }

    void setup(unsigned int fftSize_in) {                                            // :140-178
        std::lock_guard<std::mutex> g(busy_run);
        fftSize = fftSize_in;
        csdr_must(csdr_spec_setup(spec_, (int)fftSize, 1), "csdr_spec_setup");
        csdr_must(csdr_spec_set_average_rate(spec_, fft_average_rate), "csdr_spec_set_average_rate");
        csdr_must(csdr_spec_set_scale_factor(spec_, scaleFactor), "csdr_spec_set_scale_factor");
    }
    CsdrErrorLog errlog;
    void setFFTSize(unsigned int n) { std::lock_guard<std::mutex> g(busy_run); if (n != fftSize) { newFFTSize = n; fftSizeChanged = true; } }
    unsigned int getFFTSize() { std::lock_guard<std::mutex> g(busy_run); return fftSizeChanged ? newFFTSize : fftSize; }
    void setFFTAverageRate(float r) { std::lock_guard<std::mutex> g(busy_run); fft_average_rate = r; if (fftSize) csdr_spec_set_average_rate(spec_, r); }
    float getFFTAverageRate() { std::lock_guard<std::mutex> g(busy_run); return fft_average_rate; }
    void setScaleFactor(float sf) { std::lock_guard<std::mutex> g(busy_run); scaleFactor = sf; if (fftSize) csdr_spec_set_scale_factor(spec_, sf); }
    float getScaleFactor() { std::lock_guard<std::mutex> g(busy_run); return scaleFactor; }
    void setCenterFrequency(long long f) { std::lock_guard<std::mutex> g(busy_run); centerFreq = f; csdr_spec_set_center_frequency(spec_, f); }
    long long getCenterFrequency() { std::lock_guard<std::mutex> g(busy_run); return centerFreq; }
    void setBandwidth(long b) { std::lock_guard<std::mutex> g(busy_run); bandwidth = b; csdr_spec_set_bandwidth(spec_, b); }
    long getBandwidth() { std::lock_guard<std::mutex> g(busy_run); return bandwidth; }
    void setPeakHold(bool on) { std::lock_guard<std::mutex> g(busy_run); csdr_spec_set_peak_hold(spec_, on ? 1 : 0); }       // :115-125
    bool getPeakHold() { std::lock_guard<std::mutex> g(busy_run); return csdr_spec_get_peak_hold(spec_) != 0; }
    void setHideDC(bool on) { std::lock_guard<std::mutex> g(busy_run); csdr_spec_set_hide_dc(spec_, on ? 1 : 0); }           // :204-209
    // zoomed view (K17, :64-72): shift + resample to the view bandwidth, averager retune / zoom steps, fractional bins per point
    void setView(bool v) { std::lock_guard<std::mutex> g(busy_run); is_view = v; csdr_spec_set_view(spec_, v ? 1 : 0); }
    void setView(bool v, long long centerFreq_in, long bandwidth_in) { setView(v); setCenterFrequency(centerFreq_in); setBandwidth(bandwidth_in); }
    bool isView() { std::lock_guard<std::mutex> g(busy_run); return is_view; }
    int getDesiredInputSize() { std::lock_guard<std::mutex> g(busy_run); return fftSize ? csdr_spec_desired_input_size(spec_) : 0; }

protected:
    bool trySetup(unsigned int n) {                                                  // setup() from inside process(): no throw
        std::lock_guard<std::mutex> g(busy_run);
        if (!errlog.ok(csdr_spec_setup(spec_, (int)n, 1), "csdr_spec_setup")) return false;
        fftSize = n;
        (void)csdr_spec_set_average_rate(spec_, fft_average_rate);
        (void)csdr_spec_set_scale_factor(spec_, scaleFactor);
        return true;
    }
    void process() override {                                                        // :212-637, full-span branch
        if (!isOutputEmpty()) return;
        if (!input || input->empty()) return;
        bool doSetup = false;
        { std::lock_guard<std::mutex> g(busy_run); if (fftSizeChanged) { doSetup = true; fftSizeChanged = false; } }
        if (doSetup && !trySetup(newFFTSize)) return;
        DemodulatorThreadIQDataPtr iq;
        if (!input->pop(iq, HEARTBEAT_CHECK_PERIOD_MICROS) || !iq) return;
        std::lock_guard<std::mutex> g(busy_run);
        if (!fftSize || iq->data.empty()) return;
        const size_t N = 2 * (size_t)fftSize;
        CSDR_STAGE_TRY(csdr_spec_set_input_frequency(spec_, iq->frequency), "csdr_spec_set_input_frequency");
        CSDR_STAGE_TRY(csdr_spec_set_input_rate(spec_, iq->sampleRate), "csdr_spec_set_input_rate");
        // inputs of at least 2*fftSize samples are transformed directly (:401-404); shorter ones go through the
        // fftLastData priming / overlap rule (:406-420)
        const int mode = iq->data.size() >= N ? CSDR_SPEC_FIRST_FRAME : CSDR_SPEC_LINES;
        const bool inHbm = iq->deviceData && iq->deviceSamples == iq->data.size();     // the ingest's copy: read it in place
        CSDR_STAGE_TRY(csdr_spec_process(spec_, inHbm ? iq->deviceData : (const float *)iq->data.data(), inHbm ? 1 : 0, 1, (int)iq->data.size(), mode), "csdr_spec_process");
        if (csdr_spec_frames(spec_) < 1) return;                                     // the input only primed fftLastData
        SpectrumVisualDataPtr out = outputBuffers.getBuffer();
        out->spectrum_points.resize(fftSize * 2);
        CSDR_STAGE_TRY(csdr_spec_fetch(spec_, 0, out->spectrum_points.data(), (int)out->spectrum_points.size(), &out->fft_ceiling, &out->fft_floor), "csdr_spec_fetch");
        out->spectrum_hold_points.resize(fftSize * 2);
        int nh = 0;
        CSDR_STAGE_TRY(csdr_spec_fetch_hold(spec_, 0, out->spectrum_hold_points.data(), (int)out->spectrum_hold_points.size(), &nh), "csdr_spec_fetch_hold");
        out->spectrum_hold_points.resize((size_t)nh);                                // empty unless peak hold is live (:432)
        out->centerFreq = centerFreq; out->bandwidth = (int)bandwidth;
        distribute(out);
    }

private:
    csdr_ctx *ctx_;
    csdr_spec *spec_ = nullptr;
    std::mutex busy_run;
    unsigned int fftSize = 0, newFFTSize = 0;
    bool fftSizeChanged = false, is_view = false;
    float fft_average_rate = 0.65f, scaleFactor = 1.0f;
    long long centerFreq = 0;
    long bandwidth = 0;
    ReBuffer<SpectrumVisualData> outputBuffers{"SpectrumVisualProcessorBuffers"};
};

// The two pump threads around the spectrum path (src/process/FFTVisualDataThread.cpp:26-82: distributor -> processor for the
// waterfall; SpectrumVisualDataThread.cpp:14-25: the processor alone), both ticking every 10 ms.
class FFTVisualDataThread : public IOThread {
public:
    explicit FFTVisualDataThread(csdr_ctx *ctx) : wproc(ctx), linesPerSecond(DEFAULT_WATERFALL_LPS), lpsChanged(true) {}
    void setLinesPerSecond(int lps) { linesPerSecond.store(lps); lpsChanged.store(true); }
    int getLinesPerSecond() { return linesPerSecond.load(); }
    SpectrumVisualProcessor *getProcessor() { return &wproc; }
    void run() override {
        auto in = std::static_pointer_cast<DemodulatorThreadInputQueue>(getInputQueue("IQDataInput"));
        auto out = std::static_pointer_cast<SpectrumVisualDataQueue>(getOutputQueue("FFTDataOutput"));
        fftQueue->set_max_num_items(100);
        out->set_max_num_items(100);
        fftDistrib.setInput(in);
        fftDistrib.attachOutput(fftQueue);
        wproc.setInput(fftQueue);
        wproc.attachOutput(out);
        wproc.setup(DEFAULT_FFT_SIZE);
        while (!stopping) {
            std::this_thread::sleep_for(std::chrono::milliseconds((int)(FFT_DISTRIBUTOR_BUFFER_IN_SECONDS * 1000.0 / 25.0)));
            const int want = wproc.getDesiredInputSize();
            fftDistrib.setFFTSize(want ? (unsigned)want : DEFAULT_FFT_SIZE * 2);     // SPECTRUM_VZM
            if (lpsChanged.load()) { fftDistrib.setLinesPerSecond((unsigned)linesPerSecond.load()); lpsChanged.store(false); }
            fftDistrib.run();
            while (!stopping && !wproc.isInputEmpty()) wproc.run();
        }
        in->flush();
        out->flush();
    }
    void terminate() override { IOThread::terminate(); fftDistrib.flushQueues(); wproc.flushQueues(); }

protected:
    FFTDataDistributor fftDistrib;
    DemodulatorThreadInputQueuePtr fftQueue = std::make_shared<DemodulatorThreadInputQueue>();
    SpectrumVisualProcessor wproc;
    std::atomic_int linesPerSecond;
    std::atomic_bool lpsChanged;
};

class SpectrumVisualDataThread : public IOThread {
public:
    explicit SpectrumVisualDataThread(csdr_ctx *ctx) : sproc(ctx) {}
    SpectrumVisualProcessor *getProcessor() { return &sproc; }
    void run() override {
        while (!stopping) {
            std::this_thread::sleep_for(std::chrono::milliseconds((int)(FFT_DISTRIBUTOR_BUFFER_IN_SECONDS * 1000.0 / 25.0)));
            sproc.run();
        }
    }
    void terminate() override { IOThread::terminate(); sproc.flushQueues(); }

protected:
    SpectrumVisualProcessor sproc;
};
