// demod_thread_harness.cpp -- TEST INFRASTRUCTURE ONLY (oracle/).  Never linked into the product.
//
// The reference's OWN DemodulatorThread::run (src/demod/DemodulatorThread.cpp, compiled where it lies, unmodified) on a real thread, fed
// DemodulatorThreadPostIQData blocks whose modem is a FeedModem: an "analog" Modem (derived from the reference's ModemAnalog) whose
// demodulate() hands out the audio the test supplies.  Everything around that call is the reference's: the level sum (:142-160), the
// floor / ceiling trackers (:165-190), the level smoothing and the squelch state machine (:192-220), the audio peak (:223-233), the
// scope tap (:240-316) and the decision whether the audio is pushed (:318-328).  tests/test_oracle_pin.py pins the Python restatement
// (oracle/cubicsdr_chain.py RefLevelSquelch) and, through it, cubicsdr_amd/host/DemodLevel.h against this.
#include <chrono>
#include <memory>
#include <thread>
#include <vector>

#include "DemodulatorThread.h"
#include "ModemAnalog.h"
#include "DemodulatorInstance.h"
#include "CubicSDR.h"

// the few application symbols DemodulatorThread.cpp references (solo-mode squelch lock, the GUI's squelch-break cue)
static DemodulatorMgr *g_mgr_unused = nullptr;
DemodulatorMgr &OracleApp::getDemodMgr() { return *g_mgr_unused; }                // only reached in solo mode (never set here)
void DemodulatorMgr::setActiveDemodulator(const DemodulatorInstancePtr &, bool) {}
void DemodulatorMgr::setActiveDemodulatorByRawPointer(DemodulatorInstance *, bool) {}
DemodulatorInstancePtr DemodulatorMgr::getCurrentModem() { return nullptr; }
static int g_cue_count = 0;
DemodVisualCue::DemodVisualCue() {}
DemodVisualCue::~DemodVisualCue() {}
void DemodVisualCue::triggerSquelchBreak(int) { ++g_cue_count; }
static DemodVisualCue *g_cue = nullptr;
DemodVisualCue *DemodulatorInstance::getVisualCue() { if (!g_cue) g_cue = new DemodVisualCue(); return g_cue; }

namespace {
class FeedModem : public ModemAnalog {
public:
    std::vector<float> next_audio;
    int next_channels = 1;
    std::string getName() override { return "FEED"; }
    static ModemBase *factory() { return new FeedModem(); }
    int getDefaultSampleRate() override { return 12500; }
    void demodulate(ModemKit *, ModemIQData *input, AudioThreadInput *audioOut) override {
        // ModemAnalog keeps the (scaled) demodulator output of the block for the scope tap: here simply the real parts
        demodOutputData.resize(input->data.size());
        for (size_t i = 0; i < input->data.size(); ++i) demodOutputData[i] = input->data[i].real;
        audioOut->channels = next_channels;
        audioOut->data = next_audio;
    }
};
struct RefDemodThread {
    DemodulatorThread *thread;
    FeedModem *modem;
    ModemKit *kit;
    DemodulatorThreadPostInputQueuePtr in = std::make_shared<DemodulatorThreadPostInputQueue>();
    AudioThreadInputQueuePtr audio = std::make_shared<AudioThreadInputQueue>();
    DemodulatorThreadOutputQueuePtr vis = std::make_shared<DemodulatorThreadOutputQueue>(), sink = std::make_shared<DemodulatorThreadOutputQueue>();
    std::thread runner;
};
}  // namespace

extern "C" {
void *refdt_create(int use_signal_output, long long modem_rate, int audio_rate) {
    RefDemodThread *r = new RefDemodThread();
    r->thread = new DemodulatorThread(reinterpret_cast<DemodulatorInstance *>(0x1000));     // only compared / passed on, never dereferenced (stubs above)
    r->modem = new FeedModem();
    r->modem->useSignalOutput(use_signal_output != 0);
    r->kit = r->modem->buildKit(modem_rate, audio_rate);
    r->in->set_max_num_items(4); r->audio->set_max_num_items(4); r->vis->set_max_num_items(1); r->sink->set_max_num_items(4);
    r->thread->setInputQueue("IQDataInput", r->in);
    r->thread->setOutputQueue("AudioDataOutput", r->audio);
    r->thread->setOutputQueue("AudioVisualOutput", r->vis);
    r->thread->setOutputQueue("AudioSink", r->sink);
    r->runner = std::thread(&IOThread::threadMain, r->thread);
    return r;
}
void refdt_set(void *h, int squelch_enabled, float squelch_level, int muted) {
    RefDemodThread *r = (RefDemodThread *)h;
    // (setSquelchLevel switches the squelch ON when it is off, DemodulatorThread.cpp:392-397: the level goes first, then the switch)
    r->thread->setSquelchLevel(squelch_level); r->thread->setSquelchEnabled(squelch_enabled != 0); r->thread->setMuted(muted != 0);
}
// One block through run().  out[0..5] = signalLevel, signalFloor, signalCeil, squelchBreak, audio pushed to the audio queue (0 / 1), its peak;
// out[6] = scope-tap items pushed (0 / 1), out[7] = tap length, out[8] = tap inputRate, out[9] = tap sampleRate, out[10] = tap type.
// tap receives up to cap floats of the tap.  Returns 0 when the thread did not answer in time.
int refdt_block(void *h, const float *iq, int n_iq, long long rate, const float *audio, int n_audio, int channels, double *out, float *tap, int cap) {
    RefDemodThread *r = (RefDemodThread *)h;
    r->modem->next_audio.assign(audio, audio + n_audio);
    r->modem->next_channels = channels;
    auto d = std::make_shared<DemodulatorThreadPostIQData>();
    d->sampleRate = rate; d->modem = r->modem; d->modemKit = r->kit; d->modemName = "FEED"; d->modemType = "analog";
    d->data.resize((size_t)n_iq);
    for (int i = 0; i < n_iq; ++i) { d->data[i].real = iq[2 * i]; d->data[i].imag = iq[2 * i + 1]; }
    AudioThreadInputPtr drop;
    while (r->vis->try_pop(drop)) {}
    if (!r->in->push(d, 2000000)) return 0;
    AudioThreadInputPtr sunk;
    if (!r->sink->pop(sunk, 5000000)) return 0;                                  // every block reaches the sink queue last (:336-346): the block is done
    out[0] = r->thread->getSignalLevel(); out[1] = r->thread->getSignalFloor(); out[2] = r->thread->getSignalCeil();
    out[3] = r->thread->getSquelchBreak() ? 1.0 : 0.0;
    AudioThreadInputPtr a;
    out[4] = r->audio->try_pop(a) ? 1.0 : 0.0;
    out[5] = sunk ? sunk->peak : 0.0;
    AudioThreadInputPtr v;
    out[6] = r->vis->try_pop(v) ? 1.0 : 0.0;
    out[7] = out[8] = out[9] = out[10] = 0.0;
    if (v) {
        out[7] = (double)v->data.size(); out[8] = v->inputRate; out[9] = v->sampleRate; out[10] = v->type;
        const int m = (int)std::min<size_t>(v->data.size(), (size_t)cap);
        for (int i = 0; i < m; ++i) tap[i] = v->data[i];
    }
    return 1;
}
void refdt_destroy(void *h) {
    RefDemodThread *r = (RefDemodThread *)h;
    r->thread->terminate();
    r->runner.join();
    // (the thread deletes the modem it adopted, DemodulatorThread.cpp:98-108 / dtor)
    delete r->thread;
    delete r;
}
}
