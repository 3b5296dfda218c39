// distributor_harness.cpp -- TEST INFRASTRUCTURE ONLY (oracle/).  Never linked into the product.
// C entry points over the reference's OWN FFTDataDistributor (src/process/FFTDataDistributor.cpp, compiled where it lies, unmodified):
// one process() per call; each emitted line is reported as (index of its first sample in the pushed stream, length).  The samples pushed
// are their own running index (real part), so a line's origin can be read back from its first sample.  Pins oracle/fft_distributor.py.
#include <memory>
#include "FFTDataDistributor.h"

namespace {
struct RefDist {
    FFTDataDistributor d;
    DemodulatorThreadInputQueuePtr in = std::make_shared<DemodulatorThreadInputQueue>();
    DemodulatorThreadInputQueuePtr out = std::make_shared<DemodulatorThreadInputQueue>();
    double next = 0.0;
};
}
extern "C" {
void *refdist_create(unsigned fft_size, unsigned lines_per_second) {
    RefDist *r = new RefDist();
    r->in->set_max_num_items(4); r->out->set_max_num_items(100000);
    r->d.setInput(r->in); r->d.attachOutput(r->out);
    r->d.setFFTSize(fft_size); r->d.setLinesPerSecond(lines_per_second);
    return r;
}
// push n samples (ids continue from the previous call), run process(), report the emitted lines
int refdist_push(void *h, int n, long long frequency, long long sample_rate, long long *first_ids, int *lengths, int cap) {
    RefDist *r = (RefDist *)h;
    auto d = std::make_shared<DemodulatorThreadIQData>();
    d->frequency = frequency; d->sampleRate = sample_rate;
    d->data.resize((size_t)n);
    // ids exceed float precision for long streams: carry them in BOTH parts (id = re * 2^20 + im)
    for (int i = 0; i < n; ++i) { const long long id = (long long)r->next + i; d->data[i].real = (float)(id >> 20); d->data[i].imag = (float)(id & 0xFFFFF); }
    r->next += n;
    r->in->push(d);
    r->d.run();
    int m = 0;
    DemodulatorThreadIQDataPtr o;
    while (r->out->try_pop(o)) {
        if (m < cap && o && !o->data.empty()) {
            first_ids[m] = ((long long)o->data[0].real << 20) + (long long)o->data[0].imag;
            lengths[m] = (int)o->data.size();
        }
        ++m;
    }
    return m;
}
void refdist_destroy(void *h) { delete (RefDist *)h; }
}
