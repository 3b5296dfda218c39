// audio_harness.cpp -- TEST INFRASTRUCTURE ONLY (oracle/).  Never linked into the product.
//
// C entry points over the reference's OWN audio egress: the RtAudio callback `audioCallback` of src/audio/AudioThread.cpp (a file-static
// function: this translation unit includes the source where it lies, unmodified, so that it can be called) driving real AudioThread
// objects bound to a controller, and AudioFileWAV (src/audio/AudioFileWAV.cpp + AudioFile.cpp, compiled where they lie).  RtAudio is the
// reference's vendored external/rtaudio built with its dummy API (no sound device is opened).  tests/ compare csdr_mix / the PCM16
// conversion / the host AudioSinkWAV with these bit for bit.
// (every standard header first: the access-control override below must not reach the C++ library)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <typeinfo>
#include <unordered_map>
#include <vector>
#include "RtAudio.h"
#include "liquid/liquid.h"
#define private public                 // the harness sets AudioThread::sampleRate / active directly instead of opening a device
#define protected public
#include "AudioThread.cpp"
#undef private
#undef protected
#include "AudioFileWAV.h"

#include <memory>
#include <vector>

// never reached (AudioThread::setSampleRate / setupDevice are not called); defined so that the library has no loose ends
DemodulatorMgr &OracleApp::getDemodMgr() { std::abort(); }
std::vector<DemodulatorInstancePtr> DemodulatorMgr::getDemodulators() { std::abort(); }
int DemodulatorInstance::getOutputDevice() { std::abort(); }
void DemodulatorInstance::setAudioSampleRate(int) { std::abort(); }

namespace {
struct RefMixer {
    AudioThread controller;
    std::vector<std::unique_ptr<AudioThread>> sources;
};
}
extern "C" {
void *refaudio_create(int sample_rate, int n_sources, int queue_blocks) {
    RefMixer *m = new RefMixer();
    m->controller.sampleRate = sample_rate;
    for (int i = 0; i < n_sources; ++i) {
        std::unique_ptr<AudioThread> s(new AudioThread());
        s->inputQueue = std::make_shared<AudioThreadInputQueue>();
        s->inputQueue->set_max_num_items(queue_blocks > 0 ? queue_blocks : 1000000);
        s->active.store(true);
        m->controller.bindThread(s.get());
        m->sources.push_back(std::move(s));
    }
    return m;
}
void refaudio_set_source(void *h, int i, int active, float gain) { RefMixer *m = (RefMixer *)h; m->sources[i]->active.store(active != 0); m->sources[i]->gain = gain; }
// returns 1 when the queue took the block
int refaudio_push(void *h, int i, const float *data, int n, int channels, int sample_rate, float peak) {
    RefMixer *m = (RefMixer *)h;
    auto a = std::make_shared<AudioThreadInput>();
    a->channels = channels; a->sampleRate = sample_rate; a->peak = peak;
    a->data.assign(data, data + n);
    return m->sources[i]->inputQueue->try_push(a) ? 1 : 0;
}
int refaudio_queued(void *h, int i) { return (int)((RefMixer *)h)->sources[i]->inputQueue->size(); }
int refaudio_callback(void *h, float *out, int frames) { return audioCallback(out, nullptr, (unsigned)frames, 0.0, 0, &((RefMixer *)h)->controller); }
void refaudio_destroy(void *h) {
    RefMixer *m = (RefMixer *)h;
    for (auto &s : m->sources) m->controller.removeThread(s.get());
    delete m;
}

// AudioFileWAV: one file (plus roll-over files) written under `dir` with base name `base`
void *refwav_create(const char *dir, const char *base) {
    wxGetApp().getConfig()->recordingPath = dir;
    AudioFileWAV *w = new AudioFileWAV();
    w->setOutputFileName(base);
    return w;
}
int refwav_write(void *h, const float *data, int n, int channels, int sample_rate, float peak) {
    auto a = std::make_shared<AudioThreadInput>();
    a->channels = channels; a->sampleRate = sample_rate; a->peak = peak;
    a->data.assign(data, data + n);
    return ((AudioFileWAV *)h)->writeToFile(a) ? 1 : 0;
}
void refwav_close(void *h) { ((AudioFileWAV *)h)->closeFile(); }
void refwav_destroy(void *h) { delete (AudioFileWAV *)h; }
}
