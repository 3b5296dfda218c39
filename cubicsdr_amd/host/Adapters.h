// Adapters.h -- the ingest / egress edges of the hot path, device-agnostic (no SoapySDR / RtAudio / wx dependency):
//
//   SDRBlockAssembler : SDRThread::readStream's block semantics (reference src/sdr/SoapySDRThread.cpp:195-402) over any
//       CF32 stream source: MTU-sized reads appended until numElems samples are in the block, the excess of the last read carried
//       to the next block (overflowBuffer), optional I/Q swap, drop when the consumer queue is full; numChannels / numElems by
//       getOptimalChannelCount / getOptimalElementCount (:668-693).  The pooled block buffers can be page-locked once
//       (csdr_host_register) so that csdr_post_execute's host-to-device copy is a DMA from the block itself.
//   AudioMixer : the RtAudio callback's mixing of the bound demodulators' audio queues into one interleaved stereo buffer
//       (src/audio/AudioThread.cpp:88-240): per source gain, mono fan-out, sample-rate filtering, peak-normalised sum.
//   AudioSinkWAV : AudioFileWAV::writeToFile (src/audio/AudioFileWAV.cpp:63-170): 16-bit PCM with the peak-based anti-clipping
//       scale, header patched on close, 2 GB roll-over into numbered files.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/csdr_hip.h"
#include "DataTypes.h"
#include "IOThread.h"

#define CHANNELIZER_RATE_MAX 500000                  // CubicSDRDefs.h:63
#define TARGET_DISPLAY_FPS 60                        // CubicSDRDefs.h

// what SoapySDR::Device::readStream is to SDRThread: up to `maxElems` interleaved CF32 samples into `buff`; returns the count,
// 0 for "would block", < 0 for an error code
struct IQStreamSource {
    virtual ~IQStreamSource() = default;
    virtual int readStream(float *buff, int maxElems) = 0;
};

class SDRBlockAssembler {
public:
    explicit SDRBlockAssembler(csdr_ctx *ctx = nullptr) : ctx_(ctx), buffers("SDRThreadBuffers") {}
    ~SDRBlockAssembler() { for (void *p : registered_) if (ctx_) (void)csdr_host_unregister(ctx_, p); }

    static int getOptimalChannelCount(long long sampleRate_in) {                       // :676-693
        if (sampleRate_in <= CHANNELIZER_RATE_MAX) return 1;
        int optimal_count = int(std::ceil(double(sampleRate_in) / double(CHANNELIZER_RATE_MAX)));
        if (optimal_count % 2 == 1) optimal_count--;
        if (optimal_count < 2) optimal_count = 2;
        return optimal_count;
    }
    static int getOptimalElementCount(long long sampleRate_in, int fps, int nch) {     // :668-674
        int elemCount = (int)std::floor((double)sampleRate_in / (double)fps);
        return int(std::ceil((double)elemCount / (double)nch)) * nch;
    }
    void setSampleRate(long long rate) {                                               // updateSettings :505-516
        sampleRate.store(rate);
        numChannels.store(getOptimalChannelCount(rate));
        numElems.store(getOptimalElementCount(rate, TARGET_DISPLAY_FPS, numChannels.load()));
    }
    void setFrequency(long long f) { frequency.store(f < sampleRate.load() / 2 ? sampleRate.load() / 2 : f); }   // :696-701
    void setMTU(int mtu) { mtuElems.store(mtu); mtuBuf_.resize((size_t)2 * mtu); }
    void setIQSwap(bool s) { iq_swap.store(s); }
    int getNumChannels() const { return numChannels.load(); }
    int getNumElems() const { return numElems.load(); }

    // one block: returns the last read code (> 0 samples of the last read, 0 when nothing was posted, < 0 stream error)
    int readStream(IQStreamSource &device, const SDRThreadIQDataQueuePtr &iqDataOutQueue, const std::atomic_bool &stopping) {
        int n_read = 0;
        const int nElems = numElems.load(), mtElems = mtuElems.load();
        SDRThreadIQDataPtr dataOut = buffers.getBuffer();
        assure(dataOut.get(), nElems);
        if (numOverflow > 0) {                                                         // 1. the previous read's excess comes first
            const int n_overflow = std::min(numOverflow, nElems);
            std::memcpy(&dataOut->data[0], &overflowBuffer.data[0], (size_t)n_overflow * sizeof(liquid_float_complex_t));
            n_read = n_overflow;
            numOverflow -= n_overflow;
            if (numOverflow > 0) std::memmove(&overflowBuffer.data[0], &overflowBuffer.data[n_overflow], (size_t)numOverflow * sizeof(liquid_float_complex_t));
        }
        int readStreamCode = 0;
        while (n_read < nElems && !stopping) {                                         // 2. MTU-sized reads until the block is full
            const int n_stream_read = device.readStream(mtuBuf_.data(), mtElems);
            readStreamCode = n_stream_read;
            if (n_stream_read <= 0) break;
            const float *pp = mtuBuf_.data();
            const bool swap = iq_swap.load();
            auto put = [&](liquid_float_complex_t &d, const float *s) { if (swap) { d.imag = s[0]; d.real = s[1]; } else { d.real = s[0]; d.imag = s[1]; } };
            if (n_read + n_stream_read > nElems) {
                const int n_requested = nElems - n_read;
                assure(dataOut.get(), n_read + n_requested);
                for (int i = 0; i < n_requested; i++) put(dataOut->data[n_read + i], pp + 2 * i);
                pp += 2 * n_requested;
                const int numNewOverflow = n_stream_read - n_requested;
                assure(&overflowBuffer, numOverflow + numNewOverflow);
                for (int i = 0; i < numNewOverflow; i++) put(overflowBuffer.data[numOverflow + i], pp + 2 * i);
                numOverflow += numNewOverflow;
                n_read += n_requested;
            } else {
                assure(dataOut.get(), n_read + n_stream_read);
                for (int i = 0; i < n_stream_read; i++) put(dataOut->data[n_read + i], pp + 2 * i);
                n_read += n_stream_read;
            }
        }
        if (n_read > 0 && !stopping && !iqDataOutQueue->full()) {                      // 3. post, or discard when the consumer is saturated
            dataOut->data.resize((size_t)n_read);
            dataOut->frequency = frequency.load();
            dataOut->sampleRate = sampleRate.load();
            dataOut->dcCorrected = false;
            dataOut->numChannels = numChannels.load();
            if (!iqDataOutQueue->try_push(dataOut)) readStreamCode = 0;
        } else readStreamCode = 0;
        return readStreamCode;
    }
    int pendingOverflow() const { return numOverflow; }

private:
    // resize to at least n samples; a buffer whose storage moved (or is new) is page-locked for the H2D copy
    void assure(SDRThreadIQData *d, int n) {
        if ((int)d->data.size() >= n) return;
        const void *before = d->data.data();
        d->data.resize((size_t)std::max(n, numElems.load()));
        if (ctx_ && d != &overflowBuffer && d->data.data() != before) {
            if (before) { auto it = std::find(registered_.begin(), registered_.end(), (void *)before); if (it != registered_.end()) { (void)csdr_host_unregister(ctx_, *it); registered_.erase(it); } }
            if (csdr_host_register(ctx_, d->data.data(), d->data.size() * sizeof(liquid_float_complex_t)) == CSDR_OK) registered_.push_back(d->data.data());
        }
    }
    csdr_ctx *ctx_;
    ReBuffer<SDRThreadIQData> buffers;
    SDRThreadIQData overflowBuffer;
    int numOverflow = 0;
    std::vector<float> mtuBuf_;
    std::vector<void *> registered_;
    std::atomic<long long> sampleRate{0}, frequency{0};
    std::atomic_int numChannels{1}, numElems{0}, mtuElems{0};
    std::atomic_bool iq_swap{false};
};

// ---- audio egress: one mixer source per demodulator (an AudioThread bound to the device controller, AudioThread.cpp:52-72)
struct AudioMixSource {
    AudioThreadInputQueuePtr inputQueue;
    AudioThreadInputPtr currentInput;
    size_t audioQueuePtr = 0;
    float gain = 1.0f;
    std::atomic_bool active{true}, terminated{false};
    std::recursive_mutex mu;
};

class AudioMixer {
public:
    explicit AudioMixer(int sampleRate) : sampleRate_(sampleRate) {}
    void bindThread(const std::shared_ptr<AudioMixSource> &s) { std::lock_guard<std::recursive_mutex> g(mu_); if (std::find(bound_.begin(), bound_.end(), s) == bound_.end()) bound_.push_back(s); }
    void removeThread(const std::shared_ptr<AudioMixSource> &s) { std::lock_guard<std::recursive_mutex> g(mu_); bound_.erase(std::remove(bound_.begin(), bound_.end(), s), bound_.end()); }
    int getSampleRate() const { return sampleRate_; }

    // the body of audioCallback (:88-240): `out` receives nBufferFrames interleaved stereo frames
    int callback(float *out, unsigned int nBufferFrames) {
        std::memset(out, 0, (size_t)nBufferFrames * 2 * sizeof(float));
        std::lock_guard<std::recursive_mutex> lock(mu_);
        double peak = 0.0;
        for (auto &sp : bound_) {
            AudioMixSource *srcmix = sp.get();
            std::lock_guard<std::recursive_mutex> l2(srcmix->mu);
            if (srcmix->terminated || !srcmix->inputQueue || srcmix->inputQueue->empty() || !srcmix->active) continue;
            if (!srcmix->currentInput) {
                srcmix->audioQueuePtr = 0;
                (void)srcmix->inputQueue->try_pop(srcmix->currentInput);
                continue;
            }
            if (srcmix->currentInput->sampleRate != sampleRate_) {
                while (srcmix->inputQueue->try_pop(srcmix->currentInput)) {
                    if (srcmix->currentInput && srcmix->currentInput->sampleRate == sampleRate_) break;
                    srcmix->currentInput = nullptr;
                }
                srcmix->audioQueuePtr = 0;
                if (!srcmix->currentInput) continue;
            }
            if (srcmix->currentInput->channels == 0 || srcmix->currentInput->data.empty()) {
                if (!srcmix->inputQueue->empty()) {
                    srcmix->audioQueuePtr = 0;
                    srcmix->currentInput = nullptr;
                    if (!srcmix->inputQueue->try_pop(srcmix->currentInput)) continue;
                }
                continue;
            }
            double mixPeak = srcmix->currentInput->peak * srcmix->gain;
            auto next_input = [&]() -> bool {                                          // the block is used up: take the next one
                srcmix->audioQueuePtr = 0;
                srcmix->currentInput = nullptr;
                if (!srcmix->inputQueue->try_pop(srcmix->currentInput)) return false;
                const double srcPeak = srcmix->currentInput->peak * srcmix->gain;
                if (mixPeak < srcPeak) mixPeak = srcPeak;
                return true;
            };
            if (srcmix->currentInput->channels == 1) {
                for (unsigned int i = 0; i < nBufferFrames; i++) {
                    if (srcmix->audioQueuePtr >= srcmix->currentInput->data.size() && !next_input()) break;
                    if (srcmix->currentInput && !srcmix->currentInput->data.empty()) {
                        const float v = srcmix->currentInput->data[srcmix->audioQueuePtr] * srcmix->gain;
                        out[i * 2] += v; out[i * 2 + 1] += v;
                    }
                    srcmix->audioQueuePtr++;
                }
            } else {
                for (unsigned int i = 0, iMax = srcmix->currentInput->channels * nBufferFrames; i < iMax; i++) {
                    if (srcmix->audioQueuePtr >= srcmix->currentInput->data.size() && !next_input()) break;
                    if (srcmix->currentInput && !srcmix->currentInput->data.empty()) out[i] = out[i] + srcmix->currentInput->data[srcmix->audioQueuePtr] * srcmix->gain;
                    srcmix->audioQueuePtr++;
                }
            }
            peak += mixPeak;
        }
        if (peak > 1.0) {                                                             // normalise the volume
            const float invPeak = (float)(1.0 / peak);
            for (unsigned int i = 0; i < nBufferFrames * 2; i++) out[i] *= invPeak;
        }
        return 0;
    }

private:
    int sampleRate_;
    std::recursive_mutex mu_;
    std::vector<std::shared_ptr<AudioMixSource>> bound_;
};

// ---- WAV sink (AudioFileWAV.cpp:63-170; the file name policy of AudioFile / AudioSinkFileThread is the caller's)
class AudioSinkWAV {
public:
    static constexpr long long kMaxFileSize = 0x7FFFFFFFLL - 1024;                   // MAX_WAV_FILE_SIZE
    explicit AudioSinkWAV(std::string base, long long maxFileSize = kMaxFileSize) : base_(std::move(base)), maxFileSize_(maxFileSize) {}
    ~AudioSinkWAV() { closeFile(); }
    std::string getOutputFileName() const {
        std::stringstream n;
        n << base_;
        if (seq_ > 0) n << "_" << std::setfill('0') << std::setw(3) << seq_;
        n << ".wav";
        return n.str();
    }
    bool writeToFile(const AudioThreadInputPtr &input) {
        if (!out_.is_open()) { out_.open(getOutputFileName().c_str(), std::ios::binary); currentFileSize_ = 0; writeHeader(input); }
        const size_t room = (size_t)((maxFileSize_ - currentFileSize_) / (input->channels * 2));
        if (room >= input->data.size()) writePayload(input, 0, input->data.size());
        else {
            writePayload(input, 0, room);
            closeFile();
            seq_++;
            currentFileSize_ = 0;
            out_.open(getOutputFileName().c_str(), std::ios::binary);
            writeHeader(input);
            writePayload(input, room, input->data.size());
        }
        return true;
    }
    bool closeFile() {
        if (out_.is_open()) {
            const size_t file_length = (size_t)out_.tellp();
            out_.seekp((std::streamoff)dataChunkPos_ + 4); word(file_length - (dataChunkPos_ + 8), 4);
            out_.seekp(4); word(file_length - 8, 4);
            out_.close();
            currentFileSize_ = 0;
        }
        return true;
    }

private:
    template <typename W> void word(W value, unsigned size) { for (; size; --size, value >>= 8) out_.put(static_cast<char>(value & 0xFF)); }
    void writeHeader(const AudioThreadInputPtr &input) {
        out_ << "RIFF----WAVEfmt ";
        word(16, 4); word(1, 2); word(input->channels, 2); word(input->sampleRate, 4);
        word((input->sampleRate * 16 * input->channels) / 8, 4); word(input->channels * 2, 2); word(16, 2);
        dataChunkPos_ = (size_t)out_.tellp();
        currentFileSize_ = (long long)dataChunkPos_;
        out_ << "data----";
    }
    void writePayload(const AudioThreadInputPtr &input, size_t start, size_t end) {
        const float intScale = (input->peak < 1.0) ? 32767.0f : (32767.0f / input->peak);     // prevent clipping
        if (input->channels == 1) {
            for (size_t i = start; i < end; i++) { word(int(input->data[i] * intScale), 2); currentFileSize_ += 2; }
        } else if (input->channels == 2) {
            for (size_t i = start, iMax = end / 2; i < iMax; i++) {
                word(int(input->data[i * 2] * intScale), 2); word(int(input->data[i * 2 + 1] * intScale), 2);
                currentFileSize_ += 4;
            }
        }
    }
    std::string base_;
    long long maxFileSize_, currentFileSize_ = 0;
    int seq_ = 0;
    size_t dataChunkPos_ = 0;
    std::ofstream out_;
};
