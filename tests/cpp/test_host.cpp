// test_host.cpp -- tests of the C++ host mirror (cubicsdr_amd/host/).
//   ./test_host cpu   : ThreadBlockingQueue timeout semantics, ReBuffer reuse rule, IOThread lifecycle, VisualProcessor
//                       distribute (reference behaviours listed in SURVEY.md section 4, item 2); no GPU needed.
//   ./test_host gpu   : end-to-end: synthetic SDRThreadIQData blocks -> SDRPostThread (HIP) -> audio queue of an
//                       NBFM DemodulatorInstance + SpectrumVisualProcessor output queue.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <thread>

#include "../../cubicsdr_amd/host/HipPipeline.h"
#include "../../cubicsdr_amd/host/Adapters.h"
#include "../../cubicsdr_amd/host/ScopeVisualProcessor.h"

static int g_fail = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } } while (0)

struct Item { int v; };
typedef std::shared_ptr<Item> ItemPtr;

static void test_queue() {
    ThreadBlockingQueue<ItemPtr> q;
    ItemPtr a = std::make_shared<Item>(Item{1}), b = std::make_shared<Item>(Item{2}), out;
    CHECK(q.empty() && !q.full() && q.size() == 0);
    CHECK(q.try_push(a));                      // capacity defaults to 1
    CHECK(q.full() && !q.try_push(b));
    CHECK(!q.push(b, NON_BLOCKING_TIMEOUT));   // <= 100 us == try_push
    auto t0 = std::chrono::steady_clock::now();
    CHECK(!q.push(b, 20000));                  // timed wait expires
    auto dt = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    CHECK(dt >= 15 && dt < 500);
    q.set_max_num_items(3);
    CHECK(q.push(b, 20000) && q.size() == 2);
    q.set_max_num_items(1);                    // never shrinks
    CHECK(q.try_push(a) && q.full());
    CHECK(q.pop(out) && out->v == 1);          // FIFO
    CHECK(q.try_pop(out) && out->v == 2);
    q.flush();
    CHECK(q.empty() && !q.try_pop(out) && !q.pop(out, NON_BLOCKING_TIMEOUT));
    t0 = std::chrono::steady_clock::now();
    CHECK(!q.pop(out, 20000));
    dt = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    CHECK(dt >= 15 && dt < 500);
    // blocking pop is released by a producer thread; blocking push by a consumer
    std::thread prod([&] { std::this_thread::sleep_for(std::chrono::milliseconds(20)); q.push(a); });
    CHECK(q.pop(out) && out->v == 1);
    prod.join();
    ThreadBlockingQueue<ItemPtr> q1;
    q1.push(a);
    std::thread cons([&] { std::this_thread::sleep_for(std::chrono::milliseconds(20)); ItemPtr o; q1.pop(o); });
    CHECK(q1.push(b));                         // infinite wait until the consumer makes room
    cons.join();
    CHECK(q1.size() == 1);
}

static void test_rebuffer() {
    ReBuffer<Item> pool("test");
    ItemPtr a = pool.getBuffer();
    Item *pa = a.get();
    ItemPtr b = pool.getBuffer();              // a still referenced -> a new buffer
    CHECK(b.get() != pa && pool.size() == 2);
    a.reset();
    ItemPtr c = pool.getBuffer();              // a's buffer is free again (use_count == 1) -> re-used
    CHECK(c.get() == pa && pool.size() == 2);
    b.reset(); c.reset();
    for (int i = 0; i < 3 * REBUFFER_GC_LIMIT; ++i) { ItemPtr t = pool.getBuffer(); }
    CHECK(pool.size() == 1);                   // the idle second entry aged out
    pool.purge();
    CHECK(pool.size() == 0);
}

struct CountThread : IOThread {
    std::atomic<int> n{0};
    void run() override {
        auto q = std::static_pointer_cast<ThreadBlockingQueue<ItemPtr>>(getInputQueue("In"));
        while (!stopping) { ItemPtr it; if (q->pop(it, 10000)) n += it->v; }
    }
};
static void test_iothread() {
    CountThread t;
    auto q = std::make_shared<ThreadBlockingQueue<ItemPtr>>();
    q->set_max_num_items(10);
    t.setInputQueue("In", q);
    CHECK(t.getInputQueue("In") == q && t.getInputQueue("nope") == nullptr);
    std::thread th(&IOThread::threadMain, &t);
    for (int i = 0; i < 5; ++i) q->push(std::make_shared<Item>(Item{2}));
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    CHECK(!t.isTerminated());
    t.terminate();
    CHECK(t.isTerminated(1000));
    th.join();
    CHECK(t.n == 10);
}

struct Doubler : VisualProcessor<Item, Item> {
    int processed = 0;
    void process() override {
        ItemPtr in;
        if (!input->pop(in, 1000)) return;
        ++processed;
        distribute(std::make_shared<Item>(Item{in->v * 2}), 1000);
    }
};
static void test_visual_processor() {
    Doubler p;
    auto in = std::make_shared<Doubler::VisualInputQueueType>();
    auto o1 = std::make_shared<Doubler::VisualOutputQueueType>(), o2 = std::make_shared<Doubler::VisualOutputQueueType>();
    in->set_max_num_items(4);
    p.setInput(in); p.attachOutput(o1); p.attachOutput(o2);
    CHECK(p.isInputEmpty() && p.isOutputEmpty() && p.isAnyOutputEmpty());
    p.run();                                   // empty input: nothing happens
    CHECK(p.processed == 0);
    in->push(std::make_shared<Item>(Item{21}));
    p.run();
    ItemPtr a, b;
    CHECK(p.processed == 1 && o1->full() && !p.isOutputEmpty() && !p.isAnyOutputEmpty());
    CHECK(o1->try_pop(a) && o2->try_pop(b) && a == b && a->v == 42);      // the SAME shared_ptr fans out
    p.removeOutput(o2);
    in->push(std::make_shared<Item>(Item{1}));
    p.run();
    CHECK(o1->size() == 1 && o2->empty());
    p.flushQueues();
    CHECK(o1->empty() && in->empty());
}

// VisualDataDistributor shares the instance, VisualDataReDistributor hands out one pooled deep copy (VisualProcessor.h:164-224)
static void test_distributors() {
    VisualDataDistributor<Item> dist;
    auto in = std::make_shared<VisualDataDistributor<Item>::VisualInputQueueType>();
    auto o1 = std::make_shared<VisualDataDistributor<Item>::VisualOutputQueueType>(), o2 = std::make_shared<VisualDataDistributor<Item>::VisualOutputQueueType>();
    in->set_max_num_items(8); o1->set_max_num_items(2); o2->set_max_num_items(4);
    dist.setInput(in); dist.attachOutput(o1); dist.attachOutput(o2);
    auto a = std::make_shared<Item>(Item{7});
    in->push(a); in->push(std::make_shared<Item>(Item{8}));
    dist.run();
    ItemPtr x, y;
    CHECK(o1->size() == 2 && o2->size() == 2);
    CHECK(o1->try_pop(x) && o2->try_pop(y) && x == a && y == a);            // the same instance on both outputs
    struct Re : VisualDataReDistributor<Item> {} re;
    auto rin = std::make_shared<VisualDataReDistributor<Item>::VisualInputQueueType>();
    auto r1 = std::make_shared<VisualDataReDistributor<Item>::VisualOutputQueueType>();
    rin->set_max_num_items(4); r1->set_max_num_items(4);
    re.setInput(rin); re.attachOutput(r1);
    auto b = std::make_shared<Item>(Item{9});
    rin->push(b);
    re.run();
    CHECK(r1->try_pop(x) && x != b && x->v == 9);                            // a copy, not the instance
    // all outputs full: the popped item is dropped and the drain stops (as in the reference)
    auto f1 = std::make_shared<VisualDataDistributor<Item>::VisualOutputQueueType>();
    VisualDataDistributor<Item> d2;
    d2.setInput(in); d2.attachOutput(f1);
    f1->push(std::make_shared<Item>(Item{1}));                               // capacity 1: full
    in->flush(); in->push(std::make_shared<Item>(Item{2})); in->push(std::make_shared<Item>(Item{3}));
    d2.run();
    CHECK(f1->size() == 1 && in->size() == 1);
}

// the Modem registry / descriptor surface (Modem.h:127-166) and the DemodulatorInstance calls that go through it
// cubicsdr_amd/host/ScopeVisualProcessor.h against the reference's OWN ScopeVisualProcessor.cpp (oracle/_ref/libref_scope.so, built by
// oracle/Makefile from the unmodified source; test infrastructure, loaded only here): the same audio frames through both -- scope modes Y /
// 2Y / XY, mono and stereo spectra, a decimated demodulator-output tap (sampleRate != inputRate), more samples than the scope shows.
#include <dlfcn.h>
static void test_scope_against_reference(csdr_ctx *ctx, const char *root) {
    const std::string path = std::string(root) + "/oracle/_ref/libref_scope.so";
    void *lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!lib) { std::printf("scope reference not built (%s): skipped\n", dlerror()); return; }
    auto r_create = (void *(*)(int))dlsym(lib, "refscope_create");
    auto r_push = (int (*)(void *, const float *, int, int, int, int, int))dlsym(lib, "refscope_push");
    auto r_get = (int (*)(void *, int, float *, int, int *, double *))dlsym(lib, "refscope_get");
    auto r_destroy = (void (*)(void *))dlsym(lib, "refscope_destroy");
    CHECK(r_create && r_push && r_get && r_destroy);
    if (!r_create || !r_push || !r_get || !r_destroy) return;
    void *ref = r_create(DEFAULT_SCOPE_FFT_SIZE);
    ScopeVisualProcessor scope(ctx);
    auto in = std::make_shared<DemodulatorThreadOutputQueue>();
    auto out = std::make_shared<ScopeRenderDataQueue>();
    in->set_max_num_items(4); out->set_max_num_items(8);
    scope.setInput(in); scope.attachOutput(out);
    scope.setup(DEFAULT_SCOPE_FFT_SIZE);
    struct Case { int n, channels, inputRate, sampleRate, type; };
    const Case cases[] = {{800, 1, 48000, 48000, 0}, {800, 1, 48000, 48000, 0}, {1600, 2, 48000, 48000, 1}, {1600, 2, 48000, 48000, 2},
                          {2048, 1, 200000, 48000, 0}, {3000, 1, 48000, 48000, 0}, {300, 1, 12500, 12500, 0}, {800, 1, 48000, 48000, 0}};
    double worst = 0.0;
    int frames = 0;
    unsigned seed = 12345u;
    for (const Case &c : cases) {
        std::vector<float> data((size_t)c.n);
        for (int i = 0; i < c.n; ++i) {
            seed = seed * 1664525u + 1013904223u;
            data[i] = (float)(0.7 * std::sin(2 * M_PI * 1000.0 * i / 48000.0 + 0.3 * c.type) + 0.4 * std::sin(2 * M_PI * 5200.0 * i / 48000.0)
                              + 0.05 * ((double)(seed >> 8) / (1 << 24) - 0.5)) * (c.type == 2 ? 1.7f : 1.0f);
        }
        const int nref = r_push(ref, data.data(), c.n, c.channels, c.inputRate, c.sampleRate, c.type);
        auto a = std::make_shared<AudioThreadInput>();
        a->channels = c.channels; a->inputRate = c.inputRate; a->sampleRate = c.sampleRate; a->type = c.type; a->data = data;
        in->push(a);
        scope.run();
        std::vector<ScopeRenderDataPtr> got;
        ScopeRenderDataPtr o;
        while (out->try_pop(o)) got.push_back(o);
        CHECK((int)got.size() == nref && nref == 2);
        for (int k = 0; k < nref && k < (int)got.size(); ++k) {
            std::vector<float> pts(16384);
            int meta[6]; double fc[2];
            const int m = r_get(ref, k, pts.data(), (int)pts.size(), meta, fc);
            CHECK(m == (int)got[k]->waveform_points.size());
            CHECK(meta[0] == (int)got[k]->mode && (meta[1] != 0) == got[k]->spectrum && meta[2] == got[k]->channels && meta[3] == got[k]->inputRate && meta[4] == got[k]->sampleRate);
            if (got[k]->spectrum) {
                CHECK(meta[5] == got[k]->fft_size);
                CHECK(std::fabs(fc[0] - got[k]->fft_floor) <= 1e-5 * std::fabs(fc[1]) && std::fabs(fc[1] - got[k]->fft_ceil) <= 1e-5 * std::fabs(fc[1]));
            }
            double peak = 1e-30, err = 0.0;
            for (int i = 0; i < m && i < (int)got[k]->waveform_points.size(); ++i) {
                peak = std::max(peak, (double)std::fabs(pts[i]));
                err = std::max(err, (double)std::fabs(pts[i] - got[k]->waveform_points[i]));
            }
            worst = std::max(worst, err / peak);
            ++frames;
        }
    }
    std::printf("scope against the reference's ScopeVisualProcessor: %d frames, worst error %.2e of the frame's peak\n", frames, worst);
    CHECK(worst < 1e-5);
    r_destroy(ref);
}

static void test_modem_shim() {
    CHECK(Modem::getFactories().size() == 9);
    CHECK(Modem::getModemDefaultSampleRate("NBFM") == 12500 && Modem::getModemDefaultSampleRate("FM") == 200000 && Modem::getModemDefaultSampleRate("I/Q") == 48000);
    CHECK(Modem::makeModem("nope") == nullptr && Modem::getModemDefaultSampleRate("nope") == 0);
    std::unique_ptr<Modem> usb(Modem::makeModem("USB")), iq(Modem::makeModem("I/Q")), nb(Modem::makeModem("NBFM")), cw(Modem::makeModem("CW"));
    CHECK(usb && usb->getName() == "USB" && usb->getType() == "analog" && usb->csdrModemId() == CSDR_MODEM_USB && usb->useSignalOutput());
    CHECK(usb->checkSampleRate(5401, 48000) == 5402 && usb->checkSampleRate(100, 48000) == MIN_BANDWIDTH);      // ModemUSB.cpp:29-37
    CHECK(iq->checkSampleRate(12345, 44100) == 44100 && !nb->useSignalOutput() && cw->getDefaultSampleRate() == MIN_BANDWIDTH);
    ModemKit *kit = nb->buildKit(12500, 48000);
    CHECK(kit->sampleRate == 12500 && kit->audioSampleRate == 48000);
    nb->disposeKit(kit);
    bool threw = false;
    try { nb->demodulate(nullptr, nullptr, nullptr); } catch (const std::logic_error &) { threw = true; }
    CHECK(threw);                                                             // no host demodulation path exists
    // FM stereo: rate rule and the "demph" setting (ModemFMStereo.cpp:27-89)
    std::unique_ptr<Modem> fms(Modem::makeModem("FMS"));
    CHECK(fms && fms->csdrModemId() == CSDR_MODEM_FMS && fms->getDefaultSampleRate() == 200000 && !fms->useSignalOutput());
    CHECK(fms->checkSampleRate(50000, 48000) == 100000 && fms->checkSampleRate(250000, 48000) == 250000);
    CHECK(fms->getSettings().size() == 1 && fms->getSettings()[0].key == "demph" && fms->getSettings()[0].options.size() == 6);
    CHECK(fms->readSetting("demph") == "75" && fms->csdrModemArg() == 75 && !fms->shouldRebuildKit());
    fms->writeSetting("demph", "50");
    CHECK(fms->readSetting("demph") == "50" && fms->csdrModemArg() == 50 && fms->shouldRebuildKit());
    fms->writeSetting("demph", "0");
    CHECK(fms->csdrModemArg() == -1 && fms->readSettings()["demph"] == "0" && nb->getSettings().empty() && nb->csdrModemArg() == 0);
    DemodulatorMgr mgr(4);
    auto d = mgr.newThread();
    CHECK(d->getDemodulatorType() == "NBFM" && d->getBandwidth() == 12500 && d->isModemInitialized() && d->getModemType() == "analog");
    d->setDemodulatorType("USB");
    CHECK(d->getBandwidth() == 5400);
    d->setBandwidth(5401);
    CHECK(d->getBandwidth() == 5402);
    d->setDemodulatorType("bogus");
    CHECK(d->getDemodulatorType() == "USB");
    d->setGain(100.f);
    CHECK(d->getGain() == 40.0f);
    CHECK(d->getIQInputDataPipe() != nullptr && d->getModemArgs().empty() && d->readModemSetting("x").empty());
    d->run();
    CHECK(!d->isTerminated() && d->isActive());
    d->terminate();
    CHECK(d->isTerminated() && !d->isActive());
}

// SDRThread::readStream block semantics (SoapySDRThread.cpp:195-402): MTU chunks, overflow carry, IQ swap, full-queue discard
struct RampSource : IQStreamSource {
    long long pos = 0; int chunk; int fail_at = -1; int calls = 0;
    explicit RampSource(int c) : chunk(c) {}
    int readStream(float *buff, int maxElems) override {
        if (++calls == fail_at) return -1;
        const int n = std::min(chunk, maxElems);
        for (int i = 0; i < n; ++i) { buff[2 * i] = (float)(pos + i); buff[2 * i + 1] = -(float)(pos + i); }
        pos += n;
        return n;
    }
};
static void test_block_assembler() {
    CHECK(SDRBlockAssembler::getOptimalChannelCount(2400000) == 4 && SDRBlockAssembler::getOptimalChannelCount(500000) == 1);
    CHECK(SDRBlockAssembler::getOptimalChannelCount(10000000) == 20 && SDRBlockAssembler::getOptimalChannelCount(61440000) == 122);
    CHECK(SDRBlockAssembler::getOptimalChannelCount(100000000) == 200 && SDRBlockAssembler::getOptimalChannelCount(600000) == 2);
    CHECK(SDRBlockAssembler::getOptimalElementCount(10000000, 60, 20) == 166680 && SDRBlockAssembler::getOptimalElementCount(61440000, 60, 122) == 1024068);
    SDRBlockAssembler a;                          // no ctx: plain pageable buffers
    a.setSampleRate(2400000);
    a.setFrequency(100000000);
    CHECK(a.getNumChannels() == 4 && a.getNumElems() == 40000);
    a.setMTU(16384);                              // 40000 = 2 x 16384 + 7232: every block ends inside a read
    RampSource dev(16384);
    auto q = std::make_shared<SDRThreadIQDataQueue>();
    q->set_max_num_items(8);
    std::atomic_bool stopping{false};
    long long expect = 0;
    for (int b = 0; b < 5; ++b) {
        CHECK(a.readStream(dev, q, stopping) > 0);
        SDRThreadIQDataPtr blk;
        CHECK(q->try_pop(blk) && blk->data.size() == 40000 && blk->numChannels == 4 && blk->sampleRate == 2400000 && blk->frequency == 100000000);
        bool ok = true;
        for (int i = 0; i < 40000; ++i) ok = ok && blk->data[i].real == (float)(expect + i) && blk->data[i].imag == -(float)(expect + i);
        CHECK(ok);                                // the stream is continuous across blocks: nothing lost, nothing repeated
        expect += 40000;
        CHECK(a.pendingOverflow() == (int)(dev.pos - expect));
    }
    a.setIQSwap(true);
    CHECK(a.readStream(dev, q, stopping) > 0);
    SDRThreadIQDataPtr blk;
    CHECK(q->try_pop(blk));
    // samples carried over from before the swap keep their orientation; fresh ones are swapped (the reference swaps at copy time)
    CHECK(blk->data[39999].imag == (float)(expect + 39999) && blk->data[39999].real == -(float)(expect + 39999));
    expect += 40000;
    a.setIQSwap(false);
    // full consumer queue: the block is discarded, the code is 0, the overflow bookkeeping stays consistent
    auto q1 = std::make_shared<SDRThreadIQDataQueue>();
    q1->push(std::make_shared<SDRThreadIQData>());
    CHECK(a.readStream(dev, q1, stopping) == 0 && q1->size() == 1);
    // a stream error ends the block early: what was read so far is posted
    dev.fail_at = dev.calls + 2;
    const int code = a.readStream(dev, q, stopping);
    CHECK(code < 0 && q->try_pop(blk) && blk->data.size() < 40000 && !blk->data.empty());
}

// the WAV writer (AudioFileWAV.cpp:63-170) with host floats: header fields, 16-bit payload with the anti-clipping scale, sizes patched on
// close, roll-over at the size limit.  (Device-made PCM through the same writer: test_audio_egress_device; byte-for-byte against the
// reference's own AudioFileWAV: tests/test_gpu_io.py.)
static void test_wav_writer() {
    auto mk = [](int ch, int rate, int n, float v, float peak) { auto a = std::make_shared<AudioThreadInput>(); a->channels = ch; a->sampleRate = rate; a->data.assign((size_t)n, v); a->peak = peak; return a; };
    const std::string base = "/tmp/csdr_test_wav";
    WavWriter w(base, 44 + 300);           // tiny limit; the reference counts the file size from the data chunk header (36): 154 samples fit
    auto a = mk(1, 48000, 200, 0.5f, 0.5f);
    CHECK(w.writeToFile(a));
    w.closeFile();
    std::ifstream f0(base + ".wav", std::ios::binary), f1(base + "_001.wav", std::ios::binary);
    std::vector<unsigned char> b0((std::istreambuf_iterator<char>(f0)), std::istreambuf_iterator<char>()), b1((std::istreambuf_iterator<char>(f1)), std::istreambuf_iterator<char>());
    auto u32 = [](const std::vector<unsigned char> &b, size_t o) { return (unsigned)b[o] | ((unsigned)b[o + 1] << 8) | ((unsigned)b[o + 2] << 16) | ((unsigned)b[o + 3] << 24); };
    CHECK(b0.size() == 44 + 308 && b1.size() == 44 + 92);
    CHECK(std::memcmp(b0.data(), "RIFF", 4) == 0 && std::memcmp(b0.data() + 8, "WAVEfmt ", 8) == 0 && std::memcmp(b0.data() + 36, "data", 4) == 0);
    CHECK(u32(b0, 4) == b0.size() - 8 && u32(b0, 40) == 308 && u32(b0, 24) == 48000 && b0[22] == 1 && b0[34] == 16);
    const short v = (short)(b0[44] | (b0[45] << 8));
    CHECK(v == (short)int(0.5f * 32767.0f));
    // a loud block: scaled by 32767 / peak
    WavWriter w2(base + "2");
    auto loud = mk(2, 44100, 64, 2.0f, 4.0f);
    CHECK(w2.writeToFile(loud));
    w2.closeFile();
    std::ifstream f2(base + "2.wav", std::ios::binary);
    std::vector<unsigned char> b2((std::istreambuf_iterator<char>(f2)), std::istreambuf_iterator<char>());
    CHECK(b2.size() == 44 + 128 && b2[22] == 2 && u32(b2, 24) == 44100 && u32(b2, 28) == 44100 * 4 && b2[32] == 4);
    CHECK((short)(b2[44] | (b2[45] << 8)) == (short)int(2.0f * (32767.0f / 4.0f)));
    std::remove((base + ".wav").c_str()); std::remove((base + "_001.wav").c_str()); std::remove((base + "2.wav").c_str());
}

// AudioMixer over csdr_mix: the queue rules of the sound callback (AudioThread.cpp:88-240) seen through the host class; bit-for-bit
// parity with the reference's own callback is tests/test_gpu_io.py::test_mixer_is_the_reference_callback_bit_for_bit
static void test_audio_egress_device(csdr_ctx *ctx) {
    auto mk = [](int ch, int rate, int n, float v, float peak) { auto a = std::make_shared<AudioThreadInput>(); a->channels = ch; a->sampleRate = rate; a->data.assign((size_t)n, v); a->peak = peak; return a; };
    AudioMixer mix(ctx, 48000, 8, 8);
    auto s1 = mix.bindThread(), s2 = mix.bindThread();
    CHECK(s1 && s2 && s1->index() == 0 && s2->index() == 1);
    s2->setGain(0.5f);
    for (int k = 0; k < 3; ++k) CHECK(s1->try_push(mk(1, 48000, 100, 0.25f, 0.25f)));        // mono
    for (int k = 0; k < 3; ++k) CHECK(s2->try_push(mk(2, 48000, 200, 0.5f, 0.5f)));          // stereo, gain 0.5
    CHECK(s1->queued() == 3);
    std::vector<float> out(2 * 64);
    mix.callback(out.data(), 64);                 // the first call only latches each source's first block (:121-129)
    CHECK(out[0] == 0.f && out[127] == 0.f && s1->queued() == 2);
    mix.callback(out.data(), 64);
    CHECK(std::fabs(out[0] - 0.5f) < 1e-6f && std::fabs(out[127] - 0.5f) < 1e-6f);             // 0.25 + 0.5 * 0.5 on both channels
    mix.callback(out.data(), 64);                 // crosses the first mono block (100 samples) into the second one
    CHECK(std::fabs(out[2 * 40] - 0.5f) < 1e-6f);
    // the last buffer as PCM made on the device, written by the WAV writer
    std::vector<int16_t> pcm;
    CHECK(mix.lastBufferPcm16(pcm, 0.5f) && pcm.size() == 128 && pcm[0] == (int16_t)int(out[0] * 32767.0f));
    WavWriter w("/tmp/csdr_test_mix");
    CHECK(w.writePcm16(pcm.data(), pcm.size(), 2, 48000));
    w.closeFile();
    std::ifstream fm("/tmp/csdr_test_mix.wav", std::ios::binary);
    std::vector<unsigned char> bm((std::istreambuf_iterator<char>(fm)), std::istreambuf_iterator<char>());
    CHECK(bm.size() == 44 + 256 && (short)(bm[44] | (bm[45] << 8)) == pcm[0]);
    std::remove("/tmp/csdr_test_mix.wav");
    // a loud source: the sum of the peaks exceeds 1 -> the buffer is scaled by 1 / peak
    AudioMixer loud(ctx, 48000, 2, 8);
    auto s3 = loud.bindThread();
    for (int k = 0; k < 4; ++k) CHECK(s3->try_push(mk(1, 48000, 64, 2.0f, 2.0f)));
    loud.callback(out.data(), 64); loud.callback(out.data(), 64);
    CHECK(std::fabs(out[10] - 1.0f) < 1e-6f);
    // a bounded queue refuses the block that does not fit (try_push, DemodulatorThread.cpp:322)
    AudioMixer small(ctx, 48000, 1, 2);
    auto s4 = small.bindThread();
    CHECK(s4->try_push(mk(1, 48000, 64, 0.1f, 0.1f)) && s4->try_push(mk(1, 48000, 64, 0.1f, 0.1f)) && !s4->try_push(mk(1, 48000, 64, 0.1f, 0.1f)));
    // blocks at another sample rate are skipped (:131-149)
    AudioMixer m4(ctx, 48000, 1, 8);
    auto s5 = m4.bindThread();
    s5->try_push(mk(1, 44100, 64, 0.3f, 0.3f)); s5->try_push(mk(1, 44100, 64, 0.3f, 0.3f)); s5->try_push(mk(1, 48000, 64, 0.1f, 0.1f)); s5->try_push(mk(1, 48000, 64, 0.1f, 0.1f));
    m4.callback(out.data(), 32); m4.callback(out.data(), 32);
    CHECK(std::fabs(out[0] - 0.1f) < 1e-6f);
    m4.removeThread(s5);
    CHECK(!s5->try_push(mk(1, 48000, 64, 0.1f, 0.1f)));
}

// a modem that is NOT one of the nine built-in descriptors: an integrator's plug-in registered through Modem::addModemFactory.  Its
// demodulate() is host code (here: envelope detection at the modem rate, audio = |x|); the pipeline runs the front end on the device
// (CSDR_MODEM_HOST), fetches the block's resampled IQ and calls it -- DemodulatorThread.cpp:119-135.
struct EnvelopeModem : Modem {
    static ModemBase *factory() { return new EnvelopeModem(); }
    static std::atomic<int> kits, calls;
    std::string getType() override { return "analog"; }
    std::string getName() override { return "ENV"; }
    int checkSampleRate(long long rate, int) override { return rate < 2000 ? 2000 : (int)rate; }
    int getDefaultSampleRate() override { return 8000; }
    ModemKit *buildKit(long long sampleRate, int audioSampleRate) override { kits++; return Modem::buildKit(sampleRate, audioSampleRate); }
    void demodulate(ModemKit *kit, ModemIQData *input, AudioThreadInput *audioOut) override {
        calls++;
        lastRate = kit->sampleRate;
        audioOut->channels = 1; audioOut->sampleRate = (int)kit->sampleRate;
        audioOut->data.resize(input->data.size());
        for (size_t i = 0; i < input->data.size(); ++i) audioOut->data[i] = std::sqrt(input->data[i].real * input->data[i].real + input->data[i].imag * input->data[i].imag);
        lastInput = input->data;
    }
    long long lastRate = 0;
    std::vector<liquid_float_complex_t> lastInput;
};
std::atomic<int> EnvelopeModem::kits{0}, EnvelopeModem::calls{0};

// FFTDataDistributor scenarios (no GPU): the sample values carry their stream position so that the emitted lines can be
// identified; output is compared with oracle/fft_distributor.py by tests/test_host_mirror.py.
struct DistribHarness : FFTDataDistributor {
    using FFTDataDistributor::process;
};
static DemodulatorThreadIQDataPtr make_block(long long freq, long long rate, size_t n, long long &pos) {
    auto b = std::make_shared<DemodulatorThreadIQData>();
    b->frequency = freq; b->sampleRate = rate; b->data.resize(n);
    for (size_t i = 0; i < n; ++i) { b->data[i].real = (float)(pos + (long long)i); b->data[i].imag = 0.f; }
    pos += (long long)n;
    return b;
}
static int run_distrib() {
    DistribHarness d;
    auto in = std::make_shared<DemodulatorThreadInputQueue>();
    auto out = std::make_shared<DemodulatorThreadInputQueue>();
    in->set_max_num_items(1000); out->set_max_num_items(100000);
    d.setInput(in); d.attachOutput(out);
    long long pos = 0;
    auto drain = [&](const char *tag) {
        d.run();
        DemodulatorThreadIQDataPtr l;
        while (out->try_pop(l)) std::printf("L %s %lld %zu %lld %lld\n", tag, (long long)l->data[0].real, l->data.size(), l->frequency, l->sampleRate);
        std::printf("S %s %.12f %zu\n", tag, d.lineRateAccumulator(), d.buffered());
    };
    // A: 2.4 MS/s, 40000-sample blocks, 4096-sample lines at 30 lines/s, one block per run()
    d.setFFTSize(4096); d.setLinesPerSecond(30);
    for (int b = 0; b < 60; ++b) { in->push(make_block(100000000, 2400000, 40000, pos)); drain("A"); }
    // B: retune (buffer dropped), faster waterfall, several blocks queued per run()
    d.setLinesPerSecond(400);
    for (int b = 0; b < 6; ++b) { for (int k = 0; k < 3; ++k) in->push(make_block(101000000, 2400000, 40000, pos)); drain("B"); }
    // C: line size change without retune
    d.setFFTSize(16384);
    for (int b = 0; b < 8; ++b) { in->push(make_block(101000000, 2400000, 40000, pos)); drain("C"); }
    // D: low rate -> small buffer; oversized blocks overflow and lose their tail
    d.setFFTSize(2048); d.setLinesPerSecond(1000);
    for (int b = 0; b < 5; ++b) { in->push(make_block(101000000, 96000, 40000, pos)); drain("D"); }
    // E: a full consumer queue loses lines
    auto small = std::make_shared<DemodulatorThreadInputQueue>();
    d.removeOutput(out); d.attachOutput(small);
    in->push(make_block(101000000, 96000, 20000, pos));
    d.run();
    std::printf("E %zu\n", small->size());
    return 0;
}

// the level / squelch state machine on a deterministic sequence of block sums (no GPU): printed for the oracle comparison
static int run_level() {
    DemodLevelState st;
    unsigned lcg = 12345u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (double)(lcg >> 8) / 16777216.0; };
    for (int b = 0; b < 400; ++b) {
        // quiet / loud stretches, empty blocks, the squelch switched on in the middle third with a moving threshold
        const bool have = (b % 17) != 5;
        const int count = 208 + (b % 3);
        const double amp = (b / 50) % 2 ? 0.2 + 0.05 * rnd() : 0.002 + 0.001 * rnd();
        const double accum = amp * count;
        const double sampleTime = (double)count / 12500.0;
        const bool sq = b >= 130 && b < 330;
        const float sl = b < 230 ? -35.0f : -12.0f;
        const bool squelched = demodLevelStep(st, have, accum, count, sampleTime, sq, sl);
        std::printf("%d %d %.17g %d %.17g %d %.9g %d %.9g %.9g %.9g %d\n", b, have ? 1 : 0, accum, count, sampleTime, sq ? 1 : 0, (double)sl,
                    squelched ? 1 : 0, (double)st.signalLevel, (double)st.signalFloor, (double)st.signalCeil, st.squelchBreak ? 1 : 0);
    }
    return 0;
}

// plug-in modem + the one-transfer ingest, end to end: StreamReblocker blocks (I/Q exchanged by the "device") -> DeviceIngest ->
// SDRPostThread reads the HBM copy -> front end on the device -> EnvelopeModem::demodulate on the host -> audio queue
struct ToneSource : IQStreamSource {
    long long pos = 0; double fs, f0; bool qi;
    ToneSource(double fs_, double f0_, bool qi_) : fs(fs_), f0(f0_), qi(qi_) {}
    int readStream(float *buff, int maxElems) override {
        for (int i = 0; i < maxElems; ++i) {
            const double ph = 2 * M_PI * f0 * double(pos + i) / fs;
            const float re = (float)(0.4 * std::cos(ph)), im = (float)(0.4 * std::sin(ph));
            buff[2 * i] = qi ? im : re; buff[2 * i + 1] = qi ? re : im;       // a device that delivers Q, I
        }
        pos += maxElems;
        return maxElems;
    }
};
static void test_plugin_modem_and_device_ingest(csdr_ctx *ctx) {
    Modem::addModemFactory(EnvelopeModem::factory, "ENV", 8000);
    CHECK(Modem::getFactories().size() == 10 && Modem::getModemDefaultSampleRate("ENV") == 8000);
    const long long fs = 2400000, center = 100000000;
    DemodulatorMgr mgr(4);
    SDRPostThread post(ctx, &mgr);
    auto in = std::make_shared<SDRThreadIQDataQueue>();
    in->set_max_num_items(4);
    post.setInputQueue("IQDataInput", in);
    auto d = mgr.newThread();
    d->setDemodulatorType("ENV");
    CHECK(d->getDemodulatorType() == "ENV" && d->getBandwidth() == 8000);
    d->setFrequency(center + 200000);
    StreamReblocker rb(ctx);
    rb.setSampleRate(fs); rb.setFrequency(center); rb.setMTU(16384); rb.setIQSwap(true);
    DeviceIngest ingest(ctx, rb.getNumElems() + 16384, 4);
    rb.bindIngest(&ingest);
    ToneSource dev((double)fs, 200000.0 + 500.0, true);                       // 500 Hz off the demodulator's centre
    std::atomic_bool stopping{false};
    std::thread tp(&IOThread::threadMain, &post);
    int inHbm = 0;
    for (int b = 0; b < 8; ++b) {
        CHECK(rb.readStream(dev, in, stopping) > 0);
        while (post.blocksProcessed.load() <= b) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    // the blocks really went through the device copy: the pooled block objects still carry their HBM address
    {
        auto probe = std::make_shared<SDRThreadIQDataQueue>();
        probe->set_max_num_items(2);
        CHECK(rb.readStream(dev, probe, stopping) > 0);
        SDRThreadIQDataPtr blk;
        CHECK(probe->try_pop(blk) && blk->data.size() == 40000);
        if (blk->deviceData && blk->deviceSamples == 40000) ++inHbm;
        // host readers see I, Q: the carrier rotates counter-clockwise at +200.5 kHz (positive frequency)
        const double cr = (double)blk->data[10].real * blk->data[11].imag - (double)blk->data[10].imag * blk->data[11].real;
        CHECK(cr > 0);
    }
    CHECK(inHbm == 1);
    CHECK(post.errlog.errorCount() == 0);
    auto aq = d->getAudioOutputQueue();
    AudioThreadInputPtr ati;
    int nblk = 0; double mean = 0; size_t cnt = 0;
    while (aq->try_pop(ati)) {
        ++nblk;
        CHECK(ati->channels == 1 && !ati->data.empty());
        if (nblk > 3) for (float v : ati->data) { mean += v; ++cnt; }
    }
    std::printf("plug-in modem: kits %d, demodulate calls %d, audio blocks %d, envelope %.4f (carrier 0.4 x the 4-channel analyzer's gain of 4)\n", EnvelopeModem::kits.load(),
                EnvelopeModem::calls.load(), nblk, cnt ? mean / (double)cnt : 0.0);
    CHECK(EnvelopeModem::kits.load() >= 1 && EnvelopeModem::calls.load() == 8 && nblk == 8);
    CHECK(cnt > 0 && std::fabs(mean / (double)cnt - 1.6) < 0.1);                     // a constant-envelope carrier: |x| = amplitude x analyzer gain (M = 4)
    CHECK(d->getSignalLevel() > -10.0f);                                              // level from the IQ magnitudes (:150-158): 20 log10(1.55) = +3.8 dB
    post.terminate();
    tp.join();
}

static int run_gpu(const char *root) {
    csdr_ctx *ctx = nullptr;
    csdr_must(csdr_ctx_create(0, nullptr, &ctx), "csdr_ctx_create");
    test_scope_against_reference(ctx, root);
    test_audio_egress_device(ctx);
    test_plugin_modem_and_device_ingest(ctx);
    {
        DemodulatorMgr mgr(8);
        SDRPostThread post(ctx, &mgr);
        SpectrumVisualProcessor spec(ctx);
        auto pipeSDRIQData = std::make_shared<SDRThreadIQDataQueue>();               // CubicSDR.cpp:352-353
        pipeSDRIQData->set_max_num_items(100);
        auto pipeIQVisualData = std::make_shared<DemodulatorThreadInputQueue>();      // :342-343
        auto spectrumOut = std::make_shared<SpectrumVisualDataQueue>();
        post.setInputQueue("IQDataInput", pipeSDRIQData);
        post.setOutputQueue("IQVisualDataOutput", pipeIQVisualData);
        // pushVisualData (SDRPostThread.cpp:233-245) feeds the spectrum queue only when the waterfall queue is bound too
        auto pipeWaterfallIQVisualData = std::make_shared<DemodulatorThreadInputQueue>();      // CubicSDR.cpp:346-348
        pipeWaterfallIQVisualData->set_max_num_items(1000);
        post.setOutputQueue("IQDataOutput", pipeWaterfallIQVisualData);
        spec.setInput(pipeIQVisualData);
        spec.attachOutput(spectrumOut);
        spec.setup(2048);
        spec.setCenterFrequency(100000000); spec.setBandwidth(2400000);

        const long long fs = 2400000, center = 100000000;
        const int M = 4, block = 40000;
        auto d = mgr.newThread();
        d->setDemodulatorType("NBFM");
        d->setFrequency(center + 250000);
        // a second instance, FM stereo at its default 200 kHz with the 50 us de-emphasis setting: stereo frames on its audio queue
        auto dfms = mgr.newThread();
        dfms->setDemodulatorType("FMS");
        dfms->setFrequency(center - 600000);
        dfms->writeModemSetting("demph", "50");
        CHECK(dfms->getBandwidth() == 200000 && dfms->readModemSetting("demph") == "50");
        // a third instance whose sound goes to the device mixer: bank audio -> the mixer's ring inside HBM, nothing on its host audio queue
        AudioMixer mixer(ctx, 48000, 4, 100);
        auto dmix = mgr.newThread();
        dmix->setDemodulatorType("NBFM");
        dmix->setFrequency(center + 250000);
        auto mixSrc = mixer.bindThread();
        mixSrc->setGain(0.5f);
        dmix->setAudioMixSource(mixSrc);
        post.setAudioMixer(&mixer);
        // demodulator spectrum (CubicSDR.cpp:373-381): the selected modem's channel -> a second processor in view mode
        mgr.setActiveDemodulator(d, false);
        SpectrumVisualProcessor demodSpec(ctx);
        auto pipeDemodIQVisualData = std::make_shared<DemodulatorThreadInputQueue>();
        auto demodSpectrumOut = std::make_shared<SpectrumVisualDataQueue>();
        post.setOutputQueue("IQActiveDemodVisualDataOutput", pipeDemodIQVisualData);
        demodSpec.setInput(pipeDemodIQVisualData);
        demodSpec.attachOutput(demodSpectrumOut);
        demodSpec.setup(1024);                                                       // DEFAULT_DMOD_FFT_SIZE
        demodSpec.setView(true, center + 250000, 300000);
        int ndemodspec = 0;
        // audio scope: DemodulatorInstance::setVisualOutputQueue -> the <= 2048-sample tap -> ScopeVisualProcessor (CubicSDR.cpp:383-394)
        auto pipeAudioVisualData = std::make_shared<DemodulatorThreadOutputQueue>();
        pipeAudioVisualData->set_max_num_items(1);
        d->setVisualOutputQueue(pipeAudioVisualData);
        ScopeVisualProcessor scope(ctx);
        auto scopeOut = std::make_shared<ScopeRenderDataQueue>();
        scopeOut->set_max_num_items(4);
        scope.setInput(pipeAudioVisualData);
        scope.attachOutput(scopeOut);
        scope.setup(DEFAULT_SCOPE_FFT_SIZE);
        int nscope = 0, nscopespec = 0;
        std::thread tp(&IOThread::threadMain, &post);
        // FM carrier at +250 kHz, 1 kHz tone, 2.5 kHz deviation; 12 blocks = 0.2 s
        const double dev = 2500.0, ft = 1000.0, amp = 0.5;
        long long n0 = 0;
        int nspec = 0;
        for (int b = 0; b < 12; ++b) {
            auto blk = std::make_shared<SDRThreadIQData>();
            blk->frequency = center; blk->sampleRate = fs; blk->numChannels = M; blk->data.resize(block);
            for (int i = 0; i < block; ++i) {
                const double t = double(n0 + i) / fs;
                const double ph = 2 * M_PI * 250000.0 * t + (dev / ft) * std::sin(2 * M_PI * ft * t);
                blk->data[i].real = (float)(amp * std::cos(ph)); blk->data[i].imag = (float)(amp * std::sin(ph));
            }
            n0 += block;
            CHECK(pipeSDRIQData->push(blk, 2000000));
            while (post.blocksProcessed.load() <= b) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            if (b == 5) { spec.setPeakHold(true); spec.setHideDC(true); }
            demodSpec.run();
            SpectrumVisualDataPtr dv;
            if (demodSpectrumOut->try_pop(dv)) {
                ++ndemodspec;
                CHECK(dv->spectrum_points.size() == 2048);
                // channel 0 is centred on `center`; the view is centred on the carrier (+250 kHz): it peaks mid-display
                int best = 0; float bv = -1e9f;
                for (int x = 0; x < 1024; ++x) if (dv->spectrum_points[2 * x + 1] > bv) { bv = dv->spectrum_points[2 * x + 1]; best = x; }
                if (b >= 3) CHECK(std::abs(best - 512) <= 4);
            }
            scope.run();
            ScopeRenderDataPtr rd;
            while (scopeOut->try_pop(rd)) {
                if (!rd->spectrum) {
                    ++nscope;
                    // NBFM: 800 audio samples > 208 IQ samples -> the tap carries the audio (:295-300), clipped to maxScopeSamples
                    CHECK(rd->mode == ScopePanel::SCOPE_MODE_Y && rd->channels == 1 && rd->inputRate == 48000 && rd->waveform_points.size() >= 2 * 790);
                } else {
                    ++nscopespec;
                    CHECK(rd->fft_size == DEFAULT_SCOPE_FFT_SIZE / 2);
                    // sampleRate (12500, the bandwidth) != inputRate (48000): outSize = floor(512 * 12500 / 48000) = 133
                    CHECK(rd->waveform_points.size() == 2 * 133);
                    if (b >= 6) {   // the 1 kHz tone at 48 kHz sits in bin 1000 / 48000 * 1024 = 21 of the 1024-point transform
                        int best = 0; float bv = -1e9f;
                        for (int x = 1; x < 133; ++x) if (rd->waveform_points[2 * x + 1] > bv) { bv = rd->waveform_points[2 * x + 1]; best = x; }
                        CHECK(std::abs(best - 21) <= 1);
                    }
                }
            }
            spec.run();
            SpectrumVisualDataPtr sv;
            if (spectrumOut->try_pop(sv)) {
                ++nspec;
                CHECK(sv->spectrum_points.size() == 4096);
                // setPeakHold: the first input after the call resets the held arrays (b = 5), later ones carry hold points >= the live ones
                CHECK(sv->spectrum_hold_points.size() == (b >= 6 ? 4096u : 0u));
                if (b >= 6) for (int x = 1; x < 2048; x += 97) CHECK(sv->spectrum_hold_points[2 * x + 1] >= sv->spectrum_points[2 * x + 1] - 1e-5f);
                // carrier at +250 kHz of a 2.4 MHz span -> display point ~ (0.5 + 250/2400) * 2048 = 1237
                int best = 0; float bv = -1e9f;
                for (int x = 0; x < 2048; ++x) if (sv->spectrum_points[2 * x + 1] > bv) { bv = sv->spectrum_points[2 * x + 1]; best = x; }
                if (b >= 3) CHECK(std::abs(best - 1237) <= 3);
            }
        }
        CHECK(nspec >= 10);
        std::printf("scope frames %d, scope spectra %d\n", nscope, nscopespec);
        CHECK(nscope >= 9 && nscopespec >= 9);
        std::printf("demodulator-view spectra %d\n", ndemodspec);
        CHECK(ndemodspec >= 9);
        // audio: ~800 samples per block at 48 kHz; after the filters settle the output is a 1 kHz tone of amplitude
        // dev / (kf * fs_demod) = 2500 / (0.5 * 12500) = 0.4
        auto aq = d->getAudioOutputQueue();
        std::vector<float> audio;
        AudioThreadInputPtr ati;
        int nblk = 0;
        while (aq->try_pop(ati)) { ++nblk; CHECK(ati->sampleRate == 48000 && ati->channels == 1); audio.insert(audio.end(), ati->data.begin(), ati->data.end()); }
        CHECK(nblk == 12);
        CHECK(std::abs((long)audio.size() - 9600) <= 4);
        {
            auto fq = dfms->getAudioOutputQueue();
            int nf = 0; size_t fl = 0;
            while (fq->try_pop(ati)) { ++nf; CHECK(ati->sampleRate == 48000 && ati->channels == 2 && ati->data.size() % 2 == 0); fl += ati->data.size(); }
            std::printf("FM stereo audio blocks %d floats %zu\n", nf, fl);
            CHECK(nf == 12 && std::abs((long)fl - 2 * 9600) <= 8);
        }
        {
            // the mixer holds dmix's 12 blocks (~9600 samples): 9 callbacks of 1024 frames; the 1 kHz tone comes out at gain 0.5 on both channels
            CHECK(dmix->getAudioOutputQueue()->empty() && mixSrc->queued() == 12);
            std::vector<float> mo, cb(2048);
            for (int k = 0; k < 9; ++k) { mixer.callback(cb.data(), 1024); mo.insert(mo.end(), cb.begin(), cb.end()); }
            double mr = 0, mi = 0;
            const int f0 = 5000, nfm = 9 * 1024 - f0 - 600;
            for (int i = 0; i < nfm; ++i) { mr += mo[2 * (f0 + i)] * std::cos(2 * M_PI * 1000.0 * i / 48000.0); mi += mo[2 * (f0 + i)] * std::sin(2 * M_PI * 1000.0 * i / 48000.0); }
            const double mt = 2.0 * std::sqrt(mr * mr + mi * mi) / nfm;
            std::printf("device mixer: tone amplitude %.4f (expect 0.20), left == right %d\n", mt, mo[2 * 6000] == mo[2 * 6000 + 1] ? 1 : 0);
            CHECK(std::fabs(mt - 0.2) < 0.01 && mo[2 * 6000] == mo[2 * 6000 + 1]);
        }
        double re = 0, im = 0;
        const int a0 = 4800, na = (int)audio.size() - a0;
        for (int i = 0; i < na; ++i) { re += audio[a0 + i] * std::cos(2 * M_PI * 1000.0 * i / 48000.0); im += audio[a0 + i] * std::sin(2 * M_PI * 1000.0 * i / 48000.0); }
        const double tone = 2.0 * std::sqrt(re * re + im * im) / na;
        std::printf("audio blocks %d samples %zu tone amplitude %.4f (expect 0.40) level %.1f dB\n", nblk, audio.size(), tone, d->getSignalLevel());
        CHECK(std::fabs(tone - 0.4) < 0.02);
        // out-of-range demodulator is deactivated like updateActiveDemodulators() does (:66-72)
        d->setFrequency(center + 5000000);
        auto blk = std::make_shared<SDRThreadIQData>();
        blk->frequency = center; blk->sampleRate = fs; blk->numChannels = M; blk->data.assign(block, liquid_float_complex_t{0.1f, 0.0f});
        const long long before = post.blocksProcessed.load();
        pipeSDRIQData->push(blk);
        while (post.blocksProcessed.load() == before) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        CHECK(!d->isActive() && aq->empty());
        post.terminate();
        tp.join();
        CHECK(post.isTerminated());
    }
    {
        // waterfall pump: IQ blocks -> FFTDataDistributor lines (2 * fftSize samples each, FFTVisualDataThread.cpp:55-61) ->
        // SpectrumVisualProcessor -> "FFTDataOutput"
        FFTVisualDataThread wf(ctx);
        auto iqIn = std::make_shared<DemodulatorThreadInputQueue>();
        auto fftOut = std::make_shared<SpectrumVisualDataQueue>();
        iqIn->set_max_num_items(100);
        wf.setInputQueue("IQDataInput", iqIn);
        wf.setOutputQueue("FFTDataOutput", fftOut);
        wf.setLinesPerSecond(200);
        std::thread tw(&IOThread::threadMain, &wf);
        const long long fs = 2400000;
        long long n0 = 0;
        for (int b = 0; b < 12; ++b) {
            auto blk = std::make_shared<DemodulatorThreadIQData>();
            blk->frequency = 100000000; blk->sampleRate = fs; blk->data.resize(40000);
            for (int i = 0; i < 40000; ++i) {
                const double ph = 2 * M_PI * 250000.0 * double(n0 + i) / fs;
                blk->data[i].real = (float)(0.5 * std::cos(ph)); blk->data[i].imag = (float)(0.5 * std::sin(ph));
            }
            n0 += 40000;
            CHECK(iqIn->push(blk, 2000000));
            std::this_thread::sleep_for(std::chrono::milliseconds(15));
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
        int lines = 0;
        SpectrumVisualDataPtr sv;
        while (fftOut->try_pop(sv)) {
            ++lines;
            CHECK(sv->spectrum_points.size() == 2 * DEFAULT_FFT_SIZE);
            int best = 0; float bv = -1e9f;
            for (int x = 0; x < DEFAULT_FFT_SIZE; ++x) if (sv->spectrum_points[2 * x + 1] > bv) { bv = sv->spectrum_points[2 * x + 1]; best = x; }
            if (lines > 3) CHECK(std::abs(best - 1237) <= 3);
        }
        // 0.2 s of samples at 200 lines/s: about 40 lines (4096 samples each, 117 available)
        std::printf("waterfall lines %d\n", lines);
        CHECK(lines >= 20 && lines <= 60);            // (pump timing: 10 ms ticks on a possibly busy host)
        wf.terminate();
        tw.join();
        CHECK(wf.isTerminated());
    }
    {
        // single-channel mode (numChannels = 1, runSingleCH :248-299): the DC blocker runs on every block and the DC-corrected data is
        // what the visual queues receive -- with or without an active demodulator
        DemodulatorMgr mgr(2);
        SDRPostThread post(ctx, &mgr);
        auto in = std::make_shared<SDRThreadIQDataQueue>();
        auto vis = std::make_shared<DemodulatorThreadInputQueue>();
        in->set_max_num_items(10); vis->set_max_num_items(10);
        post.setInputQueue("IQDataInput", in);
        post.setOutputQueue("IQDataOutput", vis);
        std::thread tp(&IOThread::threadMain, &post);
        for (int b = 0; b < 3; ++b) {
            auto blk = std::make_shared<SDRThreadIQData>();
            blk->frequency = 50000000; blk->sampleRate = 480000; blk->numChannels = 1;
            blk->data.assign(8000, liquid_float_complex_t{0.25f, -0.125f});                 // pure DC
            const long long before = post.blocksProcessed.load();
            CHECK(in->push(blk, 2000000));
            while (post.blocksProcessed.load() == before) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        DemodulatorThreadIQDataPtr o;
        int nv = 0;
        float last = 1.f;
        while (vis->try_pop(o)) { ++nv; CHECK(o->data.size() == 8000 && o->sampleRate == 480000); last = std::fabs(o->data.back().real); }
        // y[n] = DC (1 - 0.0005)^n: after 24000 samples the offset has decayed to 0.25 e^-12 ~ 1.5e-6
        std::printf("single-channel blocks on the visual queue %d, residual DC %.3g\n", nv, last);
        CHECK(nv == 3 && last < 1e-4f);
        post.terminate();
        tp.join();
    }
    csdr_ctx_destroy(ctx);
    return g_fail;
}

int main(int argc, char **argv) {
    if (argc > 1 && !std::strcmp(argv[1], "distrib")) return run_distrib();
    if (argc > 1 && !std::strcmp(argv[1], "level")) return run_level();
    const bool gpu = argc > 1 && !std::strcmp(argv[1], "gpu");
    if (gpu) { int f = run_gpu(argc > 2 ? argv[2] : "."); std::printf(f ? "GPU HOST TEST FAILED (%d)\n" : "gpu host test ok\n", f); return f ? 1 : 0; }
    test_queue();
    test_rebuffer();
    test_iothread();
    test_visual_processor();
    test_distributors();
    test_modem_shim();
    test_block_assembler();
    test_wav_writer();
    std::printf(g_fail ? "HOST TEST FAILED (%d)\n" : "host test ok\n", g_fail);
    return g_fail ? 1 : 0;
}
