// ScopeVisualProcessor.h -- the audio scope / audio spectrum processor (reference src/process/ScopeVisualProcessor.cpp:45-217,
// ScopeVisualProcessor.h:11-66) over the HIP library: the waveform path is a copy with peak normalisation (host, a few thousand
// floats per frame), the spectrum path's fft_execute (:163) runs on the GPU (csdr_spec_fft_only on a private csdr_spec of
// fftSize points); its averaging / log scaling of fftSize / 2 bins follows in double on the host exactly as the reference's.
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
    explicit ScopeVisualProcessor(csdr_ctx *ctx) : ctx_(ctx), outputBuffers("ScopeVisualProcessorBuffers") {
        scopeEnabled.store(true);
        spectrumEnabled.store(true);
        if (csdr_spec_create(ctx_, &spec_) != CSDR_OK) throw std::runtime_error(std::string("csdr_spec_create: ") + csdr_last_error());
    }
    ~ScopeVisualProcessor() override { if (spec_) csdr_spec_destroy(spec_); }
    void setup(int fftSize_in) {                                 // :24-35: fft_create_plan(fftSize, FORWARD)
        fftSize = (unsigned)fftSize_in;
        desiredInputSize = fftSize_in;
        fftInData.assign(fftSize, liquid_float_complex_t{0.f, 0.f});
        fftOutput.assign(fftSize, liquid_float_complex_t{0.f, 0.f});
        // csdr_spec transforms 2 * fft_size points: a private instance of fftSize / 2 "display points" is an fftSize-point plan
        if (csdr_spec_setup(spec_, fftSize_in / 2, 1) != CSDR_OK) throw std::runtime_error(std::string("csdr_spec_setup: ") + csdr_last_error());
    }
    void setScopeEnabled(bool e) { scopeEnabled.store(e); }
    void setSpectrumEnabled(bool e) { spectrumEnabled.store(e); }

protected:
    void process() override {                                    // :45-217
        if (!isOutputEmpty()) return;
        AudioThreadInputPtr audioInputData;
        if (!input->try_pop(audioInputData) || !audioInputData) return;
        size_t i, iMax = audioInputData->data.size();
        if (!iMax) return;
        ScopeRenderDataPtr renderData;
        if (scopeEnabled) {
            if (iMax > maxScopeSamples) iMax = maxScopeSamples;
            renderData = outputBuffers.getBuffer();
            renderData->channels = audioInputData->channels;
            renderData->inputRate = audioInputData->inputRate;
            renderData->sampleRate = audioInputData->sampleRate;
            if (renderData->waveform_points.size() != iMax * 2) renderData->waveform_points.resize(iMax * 2);
            float peak = 1.0f;
            for (i = 0; i < iMax; i++) { const float p = std::fabs(audioInputData->data[i]); if (p > peak) peak = p; }
            if (audioInputData->type == 1) {
                iMax = audioInputData->data.size();
                if (renderData->waveform_points.size() != iMax * 2) renderData->waveform_points.resize(iMax * 2);
                for (i = 0; i < iMax; i++) {
                    renderData->waveform_points[i * 2] = (float)((((double)(i % (iMax / 2)) / (double)iMax) * 2.0 - 0.5) * 2.0);
                    renderData->waveform_points[i * 2 + 1] = audioInputData->data[i] / peak;
                }
                renderData->mode = ScopePanel::SCOPE_MODE_2Y;
            } else if (audioInputData->type == 2) {
                iMax = audioInputData->data.size();
                if (renderData->waveform_points.size() != iMax) renderData->waveform_points.resize(iMax);
                for (i = 0; i < iMax / 2; i++) {
                    renderData->waveform_points[i * 2] = audioInputData->data[i * 2] / peak;
                    renderData->waveform_points[i * 2 + 1] = audioInputData->data[i * 2 + 1] / peak;
                }
                renderData->mode = ScopePanel::SCOPE_MODE_XY;
            } else {
                for (i = 0; i < iMax; i++) {
                    renderData->waveform_points[i * 2] = (float)((((double)i / (double)iMax) - 0.5) * 2.0);
                    renderData->waveform_points[i * 2 + 1] = audioInputData->data[i] / peak;
                }
                renderData->mode = ScopePanel::SCOPE_MODE_Y;
            }
            renderData->spectrum = false;
            distribute(renderData);
        }
        if (spectrumEnabled && fftSize) {
            iMax = audioInputData->data.size();
            if (audioInputData->channels == 1) {
                for (i = 0; i < fftSize; i++) { fftInData[i].real = i < iMax ? audioInputData->data[i] : 0.f; fftInData[i].imag = 0.f; }
            } else if (audioInputData->channels == 2) {
                iMax = iMax / 2;
                for (i = 0; i < fftSize; i++) { fftInData[i].real = i < iMax ? audioInputData->data[i] + audioInputData->data[iMax + i] : 0.f; fftInData[i].imag = 0.f; }
            }
            renderData = outputBuffers.getBuffer();
            renderData->channels = audioInputData->channels;
            renderData->inputRate = audioInputData->inputRate;
            renderData->sampleRate = audioInputData->sampleRate;
            audioInputData = nullptr;
            double fft_ceil = 0, fft_floor = 1;
            if (fft_result.size() < fftSize / 2) { fft_result.resize(fftSize / 2); fft_result_ma.resize(fftSize / 2); fft_result_maa.resize(fftSize / 2); }
            // fft_execute(fftPlan) (:163) on the device
            if (csdr_spec_fft_only(spec_, (const float *)fftInData.data(), (float *)fftOutput.data()) != CSDR_OK)
                throw std::runtime_error(std::string("csdr_spec_fft_only: ") + csdr_last_error());
            for (i = 0; i < fftSize / 2; i++) {
                const double a = (double)fftOutput[i].real, b = (double)fftOutput[i].imag;
                fft_result[i] = std::sqrt(a * a + b * b);
            }
            for (i = 0; i < fftSize / 2; i++) {
                fft_result_ma[i] += (fft_result[i] - fft_result_ma[i]) * fft_average_rate;
                fft_result_maa[i] += (fft_result_ma[i] - fft_result_maa[i]) * fft_average_rate;
                if (fft_result_maa[i] > fft_ceil) fft_ceil = fft_result_maa[i];
                if (fft_result_maa[i] < fft_floor) fft_floor = fft_result_maa[i];
            }
            fft_ceil_ma = fft_ceil_ma + (fft_ceil - fft_ceil_ma) * 0.05;
            fft_ceil_maa = fft_ceil_maa + (fft_ceil_ma - fft_ceil_maa) * 0.05;
            fft_floor_ma = fft_floor_ma + (fft_floor - fft_floor_ma) * 0.05;
            fft_floor_maa = fft_floor_maa + (fft_floor_ma - fft_floor_maa) * 0.05;
            unsigned int outSize = fftSize / 2;
            if (renderData->sampleRate != renderData->inputRate)
                outSize = (unsigned)(int)std::floor((float)outSize * ((float)renderData->sampleRate / (float)renderData->inputRate));
            if (outSize > fftSize / 2) outSize = fftSize / 2;     // (the reference would read past fft_result_maa when sampleRate > inputRate)
            if (renderData->waveform_points.size() != outSize * 2) renderData->waveform_points.resize(outSize * 2);
            for (i = 0; i < outSize; i++) {
                const float v = (float)(std::log10(fft_result_maa[i] + 0.25 - (fft_floor_maa - 0.75)) / std::log10((fft_ceil_maa + 0.25) - (fft_floor_maa - 0.75)));
                renderData->waveform_points[i * 2] = (float)((double)i / (double)outSize);
                renderData->waveform_points[i * 2 + 1] = v;
            }
            renderData->fft_floor = fft_floor_maa;
            renderData->fft_ceil = fft_ceil_maa;
            renderData->fft_size = (int)(fftSize / 2);
            renderData->spectrum = true;
            distribute(renderData);
        }
    }

    csdr_ctx *ctx_;
    csdr_spec *spec_ = nullptr;
    ReBuffer<ScopeRenderData> outputBuffers;
    std::atomic_bool scopeEnabled, spectrumEnabled;
    std::vector<liquid_float_complex_t> fftInData, fftOutput;
    unsigned int fftSize = 0;
    int desiredInputSize = 0;
    unsigned int maxScopeSamples = DEFAULT_DMOD_FFT_SIZE;
    double fft_ceil_ma = 0, fft_ceil_maa = 0, fft_floor_ma = 0, fft_floor_maa = 0;
    double fft_average_rate = 0.65f;
    std::vector<double> fft_result, fft_result_ma, fft_result_maa;
};
