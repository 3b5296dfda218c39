// scope_harness.cpp -- TEST INFRASTRUCTURE ONLY (oracle/).  Never linked into the product.
// C entry points over the reference's OWN ScopeVisualProcessor (src/process/ScopeVisualProcessor.cpp, compiled where it lies, unmodified;
// oracle/ref/stub/ScopePanel.h supplies the mode enumeration of the GL panel header it includes) on the reference's liquid binary.
// tests/cpp/test_host.cpp loads it to check cubicsdr_amd/host/ScopeVisualProcessor.h frame by frame.
#include <cstring>
#include <memory>
#include <vector>
#include "ScopeVisualProcessor.h"

namespace {
struct RefScope {
    ScopeVisualProcessor proc;
    DemodulatorThreadOutputQueuePtr in = std::make_shared<DemodulatorThreadOutputQueue>();
    ScopeRenderDataQueuePtr out = std::make_shared<ScopeRenderDataQueue>();
    std::vector<ScopeRenderDataPtr> last;
};
}
extern "C" {
void *refscope_create(int fft_size) {
    RefScope *r = new RefScope();
    r->in->set_max_num_items(4); r->out->set_max_num_items(8);
    r->proc.setInput(r->in); r->proc.attachOutput(r->out);
    r->proc.setup(fft_size);
    return r;
}
void refscope_enable(void *h, int scope, int spectrum) { ((RefScope *)h)->proc.setScopeEnabled(scope != 0); ((RefScope *)h)->proc.setSpectrumEnabled(spectrum != 0); }
// one AudioThreadInput through process(); returns the number of render-data items it distributed (0..2)
int refscope_push(void *h, const float *data, int n, int channels, int input_rate, int sample_rate, int type) {
    RefScope *r = (RefScope *)h;
    auto a = std::make_shared<AudioThreadInput>();
    a->channels = channels; a->inputRate = input_rate; a->sampleRate = sample_rate; a->type = type;
    a->data.assign(data, data + n);
    r->in->push(a);
    r->proc.run();
    r->last.clear();
    ScopeRenderDataPtr o;
    while (r->out->try_pop(o)) r->last.push_back(o);
    return (int)r->last.size();
}
// item idx of the last push: points (returns their count, negative if cap is too small); meta = {mode, spectrum, channels, inputRate, sampleRate, fft_size}
int refscope_get(void *h, int idx, float *points, int cap, int *meta, double *floor_ceil) {
    RefScope *r = (RefScope *)h;
    if (idx < 0 || idx >= (int)r->last.size()) return 0;
    const ScopeRenderData &d = *r->last[idx];
    const int m = (int)d.waveform_points.size();
    if (m > cap) return -m;
    if (m) std::memcpy(points, d.waveform_points.data(), (size_t)m * sizeof(float));
    meta[0] = (int)d.mode; meta[1] = d.spectrum ? 1 : 0; meta[2] = d.channels; meta[3] = d.inputRate; meta[4] = d.sampleRate; meta[5] = d.fft_size;
    floor_ceil[0] = d.fft_floor; floor_ceil[1] = d.fft_ceil;
    return m;
}
void refscope_destroy(void *h) { delete (RefScope *)h; }
}
