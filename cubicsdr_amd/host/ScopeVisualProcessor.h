// ScopeVisualProcessor.h -- the audio scope / audio spectrum processor behind the reference's interface (src/process/
// ScopeVisualProcessor.h:11-66) with ALL of its arithmetic on the device (csdr_scope, cubicsdr_amd/csrc/kernels_io.hpp): one frame
// goes up, the waveform kernel and the spectrum kernel (transform, double averagers, trackers, log scaling) run, and the finished
// ScopeRenderData items come back -- no per-sample loop on the host.  Frames that already lie in HBM (the demodulator's tap,
// csdr_bank_scope_frame) skip the upload: processDeviceFrame().
#pragma once
#include <atomic>
#include <cmath>
#include <vector>

#include "../../include/csdr_hip.h"
#include "DataTypes.h"
#include "IOThread.h"
#include "VisualProcessor.h"

#ifndef DEFAULT_FFT_SIZE
#define DEFAULT_FFT_SIZE 2048
#endif
#define DEFAULT_DMOD_FFT_SIZE (DEFAULT_FFT_SIZE / 2)           // CubicSDRDefs.h:45
#define DEFAULT_SCOPE_FFT_SIZE (DEFAULT_FFT_SIZE / 2)

struct ScopePanel { typedef enum ScopeMode { SCOPE_MODE_Y, SCOPE_MODE_2Y, SCOPE_MODE_XY } ScopeMode; };     // ScopePanel.h:11

class ScopeRenderData {                                         // ScopeVisualProcessor.h:11-23
public:
    std::vector<float> waveform_points;
    ScopePanel::ScopeMode mode = ScopePanel::SCOPE_MODE_Y;
    int inputRate = 0;
    int sampleRate = 0;
    int channels = 0;
    bool spectrum = false;
    int fft_size = 0;
    double fft_floor = 0, fft_ceil = 0;
    virtual ~ScopeRenderData() = default;
};
typedef std::shared_ptr<ScopeRenderData> ScopeRenderDataPtr;
typedef ThreadBlockingQueue<ScopeRenderDataPtr> ScopeRenderDataQueue;
typedef std::shared_ptr<ScopeRenderDataQueue> ScopeRenderDataQueuePtr;

class ScopeVisualProcessor : public VisualProcessor<AudioThreadInput, ScopeRenderData> {
public:
    explicit ScopeVisualProcessor(csdr_ctx *ctx) : pool_("ScopeVisualProcessorBuffers") {
        if (csdr_scope_create(ctx, &scope_) != CSDR_OK) throw std::runtime_error(std::string("csdr_scope_create: ") + csdr_last_error());
    }
    ~ScopeVisualProcessor() override { if (scope_) csdr_scope_destroy(scope_); }

    void setup(int fftSize_in) {                                 // setup :24-35
        // one frame per call; a frame is at most 2 * DEMOD_VIS_SIZE floats (the stereo tap, DemodulatorThread.cpp:271-275)
        if (csdr_scope_setup(scope_, fftSize_in, 1, kMaxFrameFloats) != CSDR_OK) throw std::runtime_error(std::string("csdr_scope_setup: ") + csdr_last_error());
        fftSize_ = fftSize_in;
    }
    void setScopeEnabled(bool on) { scopeOn_.store(on); pushEnables(); }               // :37-43
    void setSpectrumEnabled(bool on) { spectrumOn_.store(on); pushEnables(); }
    long long errorCount() const { return errors_.load(); }

    // a frame that is already device memory (csdr_bank_scope_frame): same outputs, nothing uploaded
    void processDeviceFrame(const csdr_scope_frame &f) { if (f.n > 0 && isOutputEmpty()) runFrame(f, true); }

protected:
    void process() override {                                    // :45-217
        if (!isOutputEmpty()) return;                            // the previous items have not been drawn yet
        AudioThreadInputPtr in;
        if (!input || !input->try_pop(in) || !in || in->data.empty()) return;
        csdr_scope_frame f{};
        f.data = in->data.data(); f.n = (int)std::min<size_t>(in->data.size(), (size_t)kMaxFrameFloats);
        f.channels = in->channels; f.type = in->type; f.sample_rate = in->sampleRate; f.input_rate = in->inputRate; f.scale = 1.0f;
        runFrame(f, false);
    }

private:
    static constexpr int kMaxFrameFloats = 4096;
    void pushEnables() { (void)csdr_scope_set_enabled(scope_, scopeOn_.load() ? 1 : 0, spectrumOn_.load() ? 1 : 0); }
    void runFrame(const csdr_scope_frame &f, bool onDevice) {
        if (!fftSize_ || (f.channels != 1 && f.channels != 2)) return;
        if (csdr_scope_process(scope_, &f, 1, onDevice ? 1 : 0) != CSDR_OK) { errors_.fetch_add(1); return; }
        for (int which = 0; which < 2; ++which) {                // the waveform item first, then the spectrum item (:118, :214)
            ScopeRenderDataPtr item = pool_.getBuffer();
            item->waveform_points.resize((size_t)std::max(2 * kMaxFrameFloats, fftSize_));
            csdr_scope_info info{};
            if (csdr_scope_fetch(scope_, 0, which, item->waveform_points.data(), (int)item->waveform_points.size(), &info) != CSDR_OK) { errors_.fetch_add(1); return; }
            if (info.n_floats == 0) continue;                    // that half is switched off
            item->waveform_points.resize((size_t)info.n_floats);
            item->mode = (ScopePanel::ScopeMode)info.mode; item->spectrum = info.spectrum != 0;
            item->channels = info.channels; item->inputRate = info.input_rate; item->sampleRate = info.sample_rate;
            item->fft_size = info.fft_size; item->fft_floor = info.fft_floor; item->fft_ceil = info.fft_ceil;
            distribute(item);
        }
    }
    csdr_scope *scope_ = nullptr;
    int fftSize_ = 0;
    std::atomic_bool scopeOn_{true}, spectrumOn_{true};
    std::atomic<long long> errors_{0};
    ReBuffer<ScopeRenderData> pool_;
};
