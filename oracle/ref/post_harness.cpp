// post_harness.cpp -- TEST INFRASTRUCTURE ONLY (oracle/).  Never linked into the product.
//
// The reference's OWN SDRPostThread (src/sdr/SDRPostThread.cpp, compiled where it lies, unmodified; stub SoapySDR / wx headers in
// ref/stub_post only satisfy declarations its headers pull in) running run() on a real thread on the reference's liquid binary:
// DC blocker / firpfbch / firpfbch2, updateChannels, getChannelAt, the active-demodulator bookkeeping, the per-channel de-interleave and
// the try_push into every demodulator's input pipe.  The demodulators are DemodulatorInstance objects whose few methods SDRPostThread
// calls are defined HERE (frequency, active flag, input pipe; DemodulatorInstance.cpp needs the whole application and is not built).
// tests/test_oracle_pin.py holds oracle/cubicsdr_chain.py RefSDRPost -- the checker of every GPU channelizer / routing test -- to it.
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "SDRPostThread.h"
#include "DemodulatorInstance.h"
#include "CubicSDR.h"

namespace {
struct FakeDemod { long long freq = 0; bool active = false, follow = false, tracking = false; DemodulatorThreadInputQueuePtr pipe; };
std::map<const DemodulatorInstance *, FakeDemod> g_fake;
std::mutex g_mu;
FakeDemod &fake(const DemodulatorInstance *d) { std::lock_guard<std::mutex> g(g_mu); return g_fake[d]; }
DemodulatorMgr *g_mgr = nullptr;
std::vector<DemodulatorInstancePtr> g_demods;
DemodulatorInstancePtr g_current;
}  // namespace

// ---- the application side SDRPostThread.cpp talks to
DemodulatorMgr &OracleApp::getDemodMgr() { return *g_mgr; }
DemodulatorMgr::DemodulatorMgr() {}
DemodulatorMgr::~DemodulatorMgr() {}
std::vector<DemodulatorInstancePtr> DemodulatorMgr::getDemodulators() { std::lock_guard<std::mutex> g(g_mu); return g_demods; }
DemodulatorInstancePtr DemodulatorMgr::getCurrentModem() { std::lock_guard<std::mutex> g(g_mu); return g_current; }
void DemodulatorMgr::setActiveDemodulator(const DemodulatorInstancePtr &d, bool temporary) { std::lock_guard<std::mutex> g(g_mu); if (!temporary) g_current = d; }
DemodVisualCue::DemodVisualCue() {}
DemodVisualCue::~DemodVisualCue() {}
DemodulatorInstance::DemodulatorInstance() { fake(this).pipe = std::make_shared<DemodulatorThreadInputQueue>(); fake(this).pipe->set_max_num_items(8); }
DemodulatorInstance::~DemodulatorInstance() {}
bool DemodulatorInstance::isDeltaLock() { return false; }
int DemodulatorInstance::getDeltaLockOfs() { return 0; }
long long DemodulatorInstance::getFrequency() { return fake(this).freq; }
void DemodulatorInstance::setFrequency(long long f) { fake(this).freq = f; }
void DemodulatorInstance::updateLabel(long long) {}
void DemodulatorInstance::setFollow(bool f) { fake(this).follow = f; }
bool DemodulatorInstance::isFollow() { return fake(this).follow; }
void DemodulatorInstance::setTracking(bool t) { fake(this).tracking = t; }
bool DemodulatorInstance::isTracking() { return fake(this).tracking; }
bool DemodulatorInstance::isActive() { return fake(this).active; }
void DemodulatorInstance::setActive(bool a) { fake(this).active = a; }
DemodulatorThreadInputQueuePtr DemodulatorInstance::getIQInputDataPipe() { return fake(this).pipe; }

namespace {
struct RefPost {
    SDRPostThread *post;
    SDRThreadIQDataQueuePtr in = std::make_shared<SDRThreadIQDataQueue>();
    DemodulatorThreadInputQueuePtr out = std::make_shared<DemodulatorThreadInputQueue>(), vis = std::make_shared<DemodulatorThreadInputQueue>(),
                                   act = std::make_shared<DemodulatorThreadInputQueue>();
    std::thread runner;
};
}
extern "C" {
void *refpost_create(int oversampled) {
    if (!g_mgr) g_mgr = new DemodulatorMgr();
    { std::lock_guard<std::mutex> g(g_mu); g_demods.clear(); g_current = nullptr; }
    RefPost *r = new RefPost();
    r->post = new SDRPostThread();
    r->post->setChannelizerType(oversampled ? SDRPostPFBCH2 : SDRPostPFBCH);
    r->in->set_max_num_items(2); r->out->set_max_num_items(2); r->vis->set_max_num_items(2); r->act->set_max_num_items(2);
    r->post->setInputQueue("IQDataInput", r->in);
    r->post->setOutputQueue("IQDataOutput", r->out);
    r->post->setOutputQueue("IQVisualDataOutput", r->vis);
    r->post->setOutputQueue("IQActiveDemodVisualDataOutput", r->act);
    r->runner = std::thread(&IOThread::threadMain, r->post);
    return r;
}
// add a demodulator at `frequency`; returns its index.  make_current: it becomes DemodulatorMgr::getCurrentModem() (the demodulator spectrum's tap)
int refpost_add_demod(void *h, long long frequency, int make_current) {
    auto d = std::make_shared<DemodulatorInstance>();
    d->setFrequency(frequency);
    std::lock_guard<std::mutex> g(g_mu);
    g_demods.push_back(d);
    if (make_current) g_current = d;
    return (int)g_demods.size() - 1;
}
void refpost_set_demod_frequency(void *, int i, long long f) { DemodulatorInstancePtr d; { std::lock_guard<std::mutex> g(g_mu); d = g_demods[i]; } d->setFrequency(f); }
void refpost_notify(void *h) { ((RefPost *)h)->post->notifyDemodulatorsChanged(); }
void refpost_set_app(long long center, long long rate) { wxGetApp().frequency = center; wxGetApp().sampleRate = rate; }
// One SDRThreadIQData block through run(); returns when the thread has let go of it (its loop iteration -- the active-list update included --
// is over).  active[i] receives each demodulator's isActive() afterwards; returns 0 on a timeout.
int refpost_block(void *h, const float *iq, int n, long long frequency, long long rate, int num_channels, int *active, int n_demods) {
    RefPost *r = (RefPost *)h;
    auto b = std::make_shared<SDRThreadIQData>();
    b->frequency = frequency; b->sampleRate = rate; b->numChannels = num_channels; b->dcCorrected = false;
    b->data.resize((size_t)n);
    for (int i = 0; i < n; ++i) { b->data[i].real = iq[2 * i]; b->data[i].imag = iq[2 * i + 1]; }
    DemodulatorThreadIQDataPtr drop;
    while (r->out->try_pop(drop)) {}
    while (r->vis->try_pop(drop)) {}
    while (r->act->try_pop(drop)) {}
    if (!r->in->push(b, 5000000)) return 0;
    for (int spin = 0; spin < 20000 && (b.use_count() > 1 || !r->in->empty()); ++spin) std::this_thread::sleep_for(std::chrono::microseconds(250));
    if (b.use_count() > 1) return 0;
    std::vector<DemodulatorInstancePtr> ds;
    { std::lock_guard<std::mutex> g(g_mu); ds = g_demods; }
    for (int i = 0; i < n_demods && i < (int)ds.size(); ++i) active[i] = ds[i]->isActive() ? 1 : 0;
    return 1;
}
// what demodulator i found in its input pipe after the last block: returns the sample count (0: nothing was pushed), meta = {frequency, sampleRate}
int refpost_fetch(void *, int i, float *iq, int cap, long long *meta) {
    DemodulatorInstancePtr d; { std::lock_guard<std::mutex> g(g_mu); d = g_demods[i]; }
    DemodulatorThreadIQDataPtr p;
    if (!d->getIQInputDataPipe()->try_pop(p) || !p) return 0;
    const int n = (int)p->data.size();
    if (n > cap) return -n;
    for (int k = 0; k < n; ++k) { iq[2 * k] = p->data[k].real; iq[2 * k + 1] = p->data[k].imag; }
    meta[0] = p->frequency; meta[1] = p->sampleRate;
    return n;
}
// the queues: which = 0 "IQDataOutput" (waterfall), 1 "IQVisualDataOutput" (main spectrum), 2 "IQActiveDemodVisualDataOutput"
int refpost_fetch_visual(void *h, int which, float *iq, int cap, long long *meta) {
    RefPost *r = (RefPost *)h;
    DemodulatorThreadIQDataPtr p;
    DemodulatorThreadInputQueuePtr q = which == 0 ? r->out : (which == 1 ? r->vis : r->act);
    if (!q->try_pop(p) || !p) return 0;
    const int n = (int)p->data.size();
    if (n > cap) return -n;
    for (int k = 0; k < n; ++k) { iq[2 * k] = p->data[k].real; iq[2 * k + 1] = p->data[k].imag; }
    meta[0] = p->frequency; meta[1] = p->sampleRate;
    return n;
}
void refpost_destroy(void *h) {
    RefPost *r = (RefPost *)h;
    r->post->terminate();
    r->runner.join();
    delete r->post;
    { std::lock_guard<std::mutex> g(g_mu); g_demods.clear(); g_current = nullptr; g_fake.clear(); }
    delete r;
}
}
