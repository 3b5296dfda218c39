// spectrum_harness.cpp -- TEST INFRASTRUCTURE ONLY (oracle/).  Never linked into the product.
//
// C entry points over the reference's OWN SpectrumVisualProcessor (src/process/SpectrumVisualProcessor.cpp, compiled where it lies,
// unmodified, with oracle/ref/stub/CubicSDR.h in front of its include path) on the reference's own liquid binary: one process() per
// call, input pushed to its queue, output popped from the queue it distributes to.  oracle/cubicsdr_chain.py's RefSpectrum is pinned
// against it in tests/test_oracle_pin.py.
#include <cstring>
#include <memory>

#include "SpectrumVisualProcessor.h"
#include "CubicSDR.h"

namespace {
struct RefSpec {
    SpectrumVisualProcessor proc;
    DemodulatorThreadInputQueuePtr in = std::make_shared<DemodulatorThreadInputQueue>();
    SpectrumVisualDataQueuePtr out = std::make_shared<SpectrumVisualDataQueue>();
};
}  // namespace

extern "C" {
void *refspec_create(unsigned fft_size, long long app_sample_rate) {
    wxGetApp().sampleRate = app_sample_rate;
    RefSpec *r = new RefSpec();
    r->in->set_max_num_items(4);
    r->out->set_max_num_items(4);
    r->proc.setInput(r->in);
    r->proc.attachOutput(r->out);
    r->proc.setup(fft_size);
    return r;
}
void refspec_set_average_rate(void *h, float rate) { ((RefSpec *)h)->proc.setFFTAverageRate(rate); }
void refspec_set_scale(void *h, float sf) { ((RefSpec *)h)->proc.setScaleFactor(sf); }
void refspec_set_peak_hold(void *h, int on) { ((RefSpec *)h)->proc.setPeakHold(on != 0); }
void refspec_set_hide_dc(void *h, int on) { ((RefSpec *)h)->proc.setHideDC(on != 0); }
void refspec_set_center(void *h, long long f) { ((RefSpec *)h)->proc.setCenterFrequency(f); }
void refspec_set_bandwidth(void *h, long bw) { ((RefSpec *)h)->proc.setBandwidth(bw); }
void refspec_set_view(void *h, int on, long long center, long bw) { if (on) ((RefSpec *)h)->proc.setView(true, center, bw); else ((RefSpec *)h)->proc.setView(false); }
int refspec_desired_input_size(void *h) { return ((RefSpec *)h)->proc.getDesiredInputSize(); }
// one block through SpectrumVisualProcessor::process(); returns the number of spectrum points written (0: the processor produced nothing)
int refspec_process(void *h, const float *iq, int n, long long frequency, long long sample_rate, float *points, float *hold_points, int cap,
                    double *ceil_floor /* [2] */, int *n_hold) {
    RefSpec *r = (RefSpec *)h;
    auto d = std::make_shared<DemodulatorThreadIQData>();
    d->frequency = frequency;
    d->sampleRate = sample_rate;
    d->data.resize((size_t)n);
    std::memcpy(d->data.data(), iq, (size_t)n * sizeof(liquid_float_complex));
    r->in->push(d);
    r->proc.run();                                   // VisualProcessor::run(): process() under the busy lock
    SpectrumVisualDataPtr o;
    if (!r->out->try_pop(o) || !o) return 0;
    const int m = (int)o->spectrum_points.size();
    if (m > cap) return -m;
    std::memcpy(points, o->spectrum_points.data(), (size_t)m * sizeof(float));
    const int mh = (int)o->spectrum_hold_points.size();
    if (n_hold) *n_hold = mh;
    if (mh && hold_points && mh <= cap) std::memcpy(hold_points, o->spectrum_hold_points.data(), (size_t)mh * sizeof(float));
    if (ceil_floor) { ceil_floor[0] = o->fft_ceiling; ceil_floor[1] = o->fft_floor; }
    return m;
}
void refspec_destroy(void *h) { delete (RefSpec *)h; }
}
