// Adapters.h -- the ingest / egress edges of the hot path over the device library (no SoapySDR / RtAudio / wx dependency).
//
// The reference does this work sample by sample on the CPU; here the host keeps only bookkeeping and every per-sample operation
// happens where the samples already are:
//
//   StreamReblocker   what SDRThread::readStream does to the device stream (src/sdr/SoapySDRThread.cpp:195-402): MTU-sized reads become
//                     blocks of numElems samples, the surplus of the last read opens the next block, a saturated consumer loses the
//                     block.  Reads land DIRECTLY in the block (no intermediate MTU buffer, no per-sample copy loop); the I/Q
//                     exchange (:258-266) is not done here at all: the block carries a flag and the exchange happens while the block
//                     crosses the link (csdr_ingest_upload).  numChannels / numElems: :668-693.
//   DeviceIngest      ONE transfer per block into a ring of HBM slots (csdr_ingest); the block then carries its device address, which
//                     SDRPostThread and the spectrum processors read instead of uploading the host copy again (SDRPostThread.cpp:227-245
//                     hands one buffer to all consumers).
//   AudioMixer        the sound device's callback (src/audio/AudioThread.cpp:88-240) over csdr_mix: the demodulators' audio goes from the
//                     bank to per-source rings in HBM, the mix-down of a callback buffer is one kernel, one buffer comes back.
//   WavWriter         AudioFileWAV (src/audio/AudioFileWAV.cpp:63-170): RIFF bookkeeping on the host, the float -> int16 conversion with
//                     the anti-clipping scale on the device (csdr_bank_fetch_pcm16 / csdr_mix_fetch_pcm16).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
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

// ---- block geometry of a sample rate (SoapySDRThread.cpp:668-693) -------------------------------------------------------------
struct BlockGeometry {
    int channels = 1, elems = 0;
    static BlockGeometry forRate(long long rate, int fps = TARGET_DISPLAY_FPS) {
        BlockGeometry g;
        if (rate > CHANNELIZER_RATE_MAX) {
            // as many <= 500 kHz channels as cover the rate, rounded DOWN to an even count, at least two
            const int covering = (int)std::ceil((double)rate / (double)CHANNELIZER_RATE_MAX);
            g.channels = std::max(2, covering & ~1);
        }
        const int perFrame = (int)std::floor((double)rate / (double)fps);
        g.elems = ((perFrame + g.channels - 1) / g.channels) * g.channels;      // whole channelizer frames
        return g;
    }
};

// ---- one transfer per block into a ring of HBM slots ----------------------------------------------------------------------------
class DeviceIngest {
public:
    DeviceIngest(csdr_ctx *ctx, long long maxSamples, int depth = 4) : holds_((size_t)depth) {
        if (csdr_ingest_create(ctx, maxSamples, depth, &ing_) != CSDR_OK) throw std::runtime_error(std::string("csdr_ingest_create: ") + csdr_last_error());
    }
    ~DeviceIngest() { if (ing_) csdr_ingest_destroy(ing_); }
    // Moves blk.data over the link (exchanging I and Q on the way when blk.iqSwapPending) and records the HBM address in the block.
    // false: the slot that is next in the ring is still held by a consumer (or the transfer failed) -- the block stays host-only.
    // The transfer is a DMA from blk.data that may still be running on return: wait() before the buffer is rewritten or recycled.
    bool upload(SDRThreadIQData &blk) {
        const int k = csdr_ingest_next_slot(ing_);
        if (k < 0 || (holds_[(size_t)k] && holds_[(size_t)k].use_count() > 1)) return false;
        const float *dev = nullptr;
        if (csdr_ingest_upload(ing_, reinterpret_cast<const float *>(blk.data.data()), (long long)blk.data.size(), blk.iqSwapPending ? 1 : 0, &dev) != CSDR_OK) return false;
        holds_[(size_t)k] = std::make_shared<int>(k);
        blk.deviceData = dev; blk.deviceSamples = blk.data.size(); blk.deviceHold = holds_[(size_t)k];
        return true;
    }
    void wait() { (void)csdr_ingest_wait(ing_); }

private:
    csdr_ingest *ing_ = nullptr;
    std::vector<std::shared_ptr<void>> holds_;
};

// ---- the device stream cut into blocks ----------------------------------------------------------------------------------------
// A block buffer holds numElems samples plus room for one whole read: every read is written straight behind what the block already
// holds; when that passes numElems the surplus is the head of the NEXT block and is moved there once, as one memmove.
class StreamReblocker {
public:
    explicit StreamReblocker(csdr_ctx *ctx = nullptr) : ctx_(ctx), pool_("SDRThreadBuffers") {}
    ~StreamReblocker() { for (void *p : pinned_) if (ctx_) (void)csdr_host_unregister(ctx_, p); }

    static int getOptimalChannelCount(long long rate) { return BlockGeometry::forRate(rate).channels; }
    static int getOptimalElementCount(long long rate, int fps, int nch) {
        const int perFrame = (int)std::floor((double)rate / (double)fps);
        return ((perFrame + nch - 1) / nch) * nch;
    }
    void setSampleRate(long long rate) {                                               // updateSettings :505-516
        const BlockGeometry g = BlockGeometry::forRate(rate);
        rate_.store(rate); channels_.store(g.channels); elems_.store(g.elems);
    }
    void setFrequency(long long f) { const long long lo = rate_.load() / 2; freq_.store(f < lo ? lo : f); }   // :696-701
    void setMTU(int mtu) { mtu_.store(mtu); }
    void setIQSwap(bool s) { swap_.store(s); }
    int getNumChannels() const { return channels_.load(); }
    int getNumElems() const { return elems_.load(); }
    int pendingOverflow() const { return (int)spill_.size(); }

    // Assemble and post ONE block.  Returns the code of the last read: > 0 its sample count, 0 when nothing was posted (would-block
    // read with an empty block, stop request, saturated consumer), < 0 the stream's error code (what was read so far is still posted).
    int readStream(IQStreamSource &dev, const SDRThreadIQDataQueuePtr &out, const std::atomic_bool &stopping) {
        const int want = elems_.load(), mtu = std::max(1, mtu_.load());
        // a pooled block may be one whose upload is still reading it (pushed, released early by its consumers, taken again): the one transfer
        // in flight is waited for before any block is refilled
        if (uploading_ && ingest_) ingest_->wait();
        uploading_ = false;
        SDRThreadIQDataPtr blk = pool_.getBuffer();
        reserve(*blk, (size_t)want + (size_t)mtu);
        liquid_float_complex_t *base = blk->data.data();
        // the surplus of the previous block comes first; with a small block and a large MTU it can exceed a whole block
        size_t have = std::min(spill_.size(), (size_t)want);
        bool swapped = false;                                                          // what the block holds so far still needs the I/Q exchange
        if (have) {
            std::memcpy(base, spill_.data(), have * sizeof *base);
            spill_.erase(spill_.begin(), spill_.begin() + (long)have);
            swapped = spillSwapped_;
        }
        int code = 0;
        while ((int)have < want && !stopping.load()) {
            const bool sw = swap_.load();
            code = dev.readStream(reinterpret_cast<float *>(base + have), mtu);         // straight into the block
            if (code <= 0) break;
            if (have == 0) swapped = sw;
            else if (sw != swapped) {
                // the option changed inside this block (a user action): exchange what is there on the host, once, so that the whole
                // block needs the same treatment again (the exchange is its own inverse)
                exchange(base, have);
                swapped = sw;
            }
            have += (size_t)code;
        }
        if ((int)have > want) {                                                        // the last read ran past the block: carry the rest
            spill_.insert(spill_.end(), base + want, base + have);
            spillSwapped_ = swapped;
            have = (size_t)want;
        }
        if (have == 0 || stopping.load() || out->full()) return 0;                     // nothing to hand over / the consumer is saturated
        blk->data.resize(have);
        blk->frequency = freq_.load(); blk->sampleRate = rate_.load(); blk->numChannels = channels_.load(); blk->dcCorrected = false;
        blk->iqSwapPending = swapped;
        blk->dropDeviceCopy();
        // host readers of `data` need the exchanged orientation anyway (one pass over the host copy, only while the option is on): it is made
        // BEFORE the transfer -- the DMA reads this buffer asynchronously, an exchange issued behind it would race with it -- and the block then
        // crosses the link ONCE, as it is
        if (blk->iqSwapPending) exchange(blk->data.data(), have);
        blk->iqSwapPending = false;
        const bool inHbm = ingest_ && ingest_->upload(*blk);
        uploading_ = inHbm;
        if (!out->try_push(blk)) return 0;       // (the block goes back to the pool; the next read waits for its transfer before refilling anything)
        return code;
    }
    void bindIngest(DeviceIngest *ing) { ingest_ = ing; }

private:
    static void exchange(liquid_float_complex_t *p, size_t n) { for (size_t i = 0; i < n; ++i) std::swap(p[i].real, p[i].imag); }
    // capacity only: the pooled vector keeps its storage (page-locked once per storage so that the transfer is a DMA from the block)
    void reserve(SDRThreadIQData &blk, size_t n) {
        if (blk.data.size() >= n) return;
        const void *before = blk.data.empty() ? nullptr : blk.data.data();
        blk.data.resize(n);
        if (!ctx_ || blk.data.data() == before) return;
        auto it = std::find(pinned_.begin(), pinned_.end(), (void *)before);
        if (it != pinned_.end()) { (void)csdr_host_unregister(ctx_, *it); pinned_.erase(it); }
        if (csdr_host_register(ctx_, blk.data.data(), blk.data.size() * sizeof(liquid_float_complex_t)) == CSDR_OK) pinned_.push_back(blk.data.data());
    }
    csdr_ctx *ctx_;
    DeviceIngest *ingest_ = nullptr;
    ReBuffer<SDRThreadIQData> pool_;
    std::vector<liquid_float_complex_t> spill_;
    bool spillSwapped_ = false;
    bool uploading_ = false;                      // the last block's asynchronous upload has not been waited for yet
    std::vector<void *> pinned_;
    std::atomic<long long> rate_{0}, freq_{0};
    std::atomic_int channels_{1}, elems_{0}, mtu_{0};
    std::atomic_bool swap_{false};
};
typedef StreamReblocker SDRBlockAssembler;          // (the name earlier revisions and INTEGRATION.md use)

// ---- audio egress -------------------------------------------------------------------------------------------------------------
// One source per demodulator (the reference binds one AudioThread per DemodulatorInstance to the device's controller thread,
// AudioThread.cpp:52-74).  A source's queue lives in the device library: blocks go in from the host (try_push) or straight from the
// bank in HBM (AudioMixer::takeBankAudio).
class AudioMixer;
class AudioMixSource {
public:
    // AudioThreadInputQueue::try_push of the reference's inputQueue: false when the queue is full (the block is lost, :322)
    bool try_push(const AudioThreadInputPtr &a);
    size_t queued() const;
    void setGain(float g) { gain_ = g < 0.005f ? 0.005f : (g > 40.0f ? 40.0f : g); apply(); }    // AudioThread::setGain clamps (:523-531)
    float getGain() const { return gain_; }
    void setActive(bool a) { active_ = a; apply(); }
    bool isActive() const { return active_; }
    int index() const { return index_; }

private:
    friend class AudioMixer;
    void apply();
    AudioMixer *mixer_ = nullptr;
    int index_ = -1, queueBlocks_ = 0;
    float gain_ = 1.0f;
    bool active_ = true;
};

class AudioMixer {
public:
    AudioMixer(csdr_ctx *ctx, int sampleRate, int maxSources = 64, int queueBlocks = 100, int ringFloats = 1 << 20) : rate_(sampleRate), queueBlocks_(queueBlocks) {
        if (csdr_mix_create(ctx, maxSources, ringFloats, sampleRate, &mix_) != CSDR_OK) throw std::runtime_error(std::string("csdr_mix_create: ") + csdr_last_error());
        slots_.assign((size_t)maxSources, nullptr);
    }
    ~AudioMixer() { if (mix_) csdr_mix_destroy(mix_); }
    int getSampleRate() const { return rate_; }

    std::shared_ptr<AudioMixSource> bindThread() {                                    // bindThread (:52-62): binding order = mixing order
        std::lock_guard<std::mutex> g(mu_);
        for (size_t i = 0; i < slots_.size(); ++i)
            if (!slots_[i]) {
                auto s = std::make_shared<AudioMixSource>();
                s->mixer_ = this; s->index_ = (int)i; s->queueBlocks_ = queueBlocks_;
                slots_[i] = s;
                (void)csdr_mix_set_source(mix_, (int)i, 1, 1, 1.0f, queueBlocks_);
                return s;
            }
        return nullptr;
    }
    void removeThread(const std::shared_ptr<AudioMixSource> &s) {                     // :64-74
        std::lock_guard<std::mutex> g(mu_);
        if (!s || s->index_ < 0 || slots_[(size_t)s->index_] != s) return;
        (void)csdr_mix_set_source(mix_, s->index_, 0, 0, 1.0f, 0);
        slots_[(size_t)s->index_] = nullptr;
        s->mixer_ = nullptr;
    }
    // the audio of every block of the bank's last execute for these demodulators, HBM to HBM (squelched / muted ones are simply not listed)
    bool takeBankAudio(csdr_bank *bank, const std::vector<int> &bankSlots, const std::vector<std::shared_ptr<AudioMixSource>> &sources) {
        std::lock_guard<std::mutex> g(mu_);
        std::vector<int> idx;
        for (auto &s : sources) idx.push_back(s ? s->index_ : -1);
        return !bankSlots.empty() && csdr_mix_push_bank(mix_, bank, bankSlots.data(), idx.data(), (int)bankSlots.size()) == CSDR_OK;
    }
    // the sound device's callback: nBufferFrames interleaved stereo frames into `out`; always 0 (the reference returns 1 only when terminated)
    int callback(float *out, unsigned int nBufferFrames) {
        std::lock_guard<std::mutex> g(mu_);
        if (csdr_mix_render(mix_, (int)nBufferFrames, 1, out) != CSDR_OK) std::memset(out, 0, (size_t)nBufferFrames * 2 * sizeof(float));
        return 0;
    }
    // the last callback buffer as 16-bit PCM (a recording of the mixed output)
    bool lastBufferPcm16(std::vector<int16_t> &pcm, float peak) {
        std::lock_guard<std::mutex> g(mu_);
        int n = 0;
        pcm.resize(1 << 16);
        if (csdr_mix_fetch_pcm16(mix_, pcm.data(), (int)pcm.size(), peak, 0, &n) != CSDR_OK) return false;
        pcm.resize((size_t)n);
        return true;
    }

private:
    friend class AudioMixSource;
    csdr_mix *mix_ = nullptr;
    int rate_, queueBlocks_;
    std::mutex mu_;
    std::vector<std::shared_ptr<AudioMixSource>> slots_;
};
inline bool AudioMixSource::try_push(const AudioThreadInputPtr &a) {
    if (!mixer_ || !a) return false;
    std::lock_guard<std::mutex> g(mixer_->mu_);
    return csdr_mix_push(mixer_->mix_, index_, a->data.empty() ? nullptr : a->data.data(), 0, (int)a->data.size(), a->channels, a->sampleRate, a->peak) == CSDR_OK;
}
inline size_t AudioMixSource::queued() const { return mixer_ ? (size_t)csdr_mix_queued(mixer_->mix_, index_) : 0; }
inline void AudioMixSource::apply() {
    if (!mixer_) return;
    std::lock_guard<std::mutex> g(mixer_->mu_);
    (void)csdr_mix_set_source(mixer_->mix_, index_, 1, active_ ? 1 : 0, gain_, queueBlocks_);
}

// ---- WAV files ----------------------------------------------------------------------------------------------------------------
// Canonical 44-byte PCM header (AudioFileWAV.cpp:108-131), sizes patched on close (:93-106), a new numbered file when the size limit
// would be passed (:76-91).  Samples arrive as 16-bit PCM made on the device; AudioThreadInput blocks with host floats are accepted
// for callers that have nothing else (same conversion, :133-157).
class WavWriter {
public:
    static constexpr long long kMaxFileSize = 0x7FFFFFFFLL - 1024;                   // MAX_WAV_FILE_SIZE (:9)
    explicit WavWriter(std::string base, long long maxFileSize = kMaxFileSize) : base_(std::move(base)), limit_(maxFileSize) {}
    ~WavWriter() { closeFile(); }
    std::string getOutputFileName() const {
        char suffix[16] = "";
        if (seq_ > 0) std::snprintf(suffix, sizeof suffix, "_%03d", seq_);
        return base_ + suffix + ".wav";
    }
    // n 16-bit samples (interleaved when channels == 2)
    bool writePcm16(const int16_t *pcm, size_t n, int channels, int sampleRate) {
        if (channels < 1 || channels > 2) return false;
        while (n) {
            if (!f_ && !open(channels, sampleRate)) return false;
            // whole samples that still fit under the limit; the reference counts the file size from the data chunk header on
            const size_t room = (size_t)std::max<long long>(0, (limit_ - size_) / (2 * channels)) * (size_t)channels;   // whole frames
            const size_t take = std::min(n, room);
            if (take) {
                if (std::fwrite(pcm, sizeof(int16_t), take, f_) != take) return false;
                size_ += (long long)take * 2; pcm += take; n -= take;
            }
            if (n) { closeFile(); ++seq_; }                                           // the rest opens the next file of the sequence
        }
        return true;
    }
    // an AudioThreadInput with host floats: int(x * scale) with scale = peak < 1 ? 32767 : 32767 / peak, low 16 bits
    bool writeToFile(const AudioThreadInputPtr &in) {
        if (!in || in->channels < 1 || in->channels > 2) return false;
        const float scale = in->peak < 1.0f ? 32767.0f : 32767.0f / in->peak;
        const size_t n = in->channels == 2 ? in->data.size() & ~(size_t)1 : in->data.size();
        conv_.resize(n);
        for (size_t i = 0; i < n; ++i) conv_[i] = (int16_t)(int)(in->data[i] * scale);
        return writePcm16(conv_.data(), n, in->channels, in->sampleRate);
    }
    bool closeFile() {
        if (!f_) return true;
        const long end = std::ftell(f_);
        put32(40, (uint32_t)(end - 44));                                               // data chunk size
        put32(4, (uint32_t)(end - 8));                                                 // RIFF chunk size
        std::fclose(f_);
        f_ = nullptr; size_ = 0;
        return true;
    }

private:
    bool open(int channels, int rate) {
        f_ = std::fopen(getOutputFileName().c_str(), "wb");
        if (!f_) return false;
        unsigned char h[44];
        std::memcpy(h, "RIFF\0\0\0\0WAVEfmt ", 16);
        le(h + 16, 16, 4); le(h + 20, 1, 2); le(h + 22, (uint32_t)channels, 2); le(h + 24, (uint32_t)rate, 4);
        le(h + 28, (uint32_t)(rate * 2 * channels), 4); le(h + 32, (uint32_t)(2 * channels), 2); le(h + 34, 16, 2);
        std::memcpy(h + 36, "data\0\0\0\0", 8);
        size_ = 36;                                                                    // currentFileSize = dataChunkPos (:128-129)
        return std::fwrite(h, 1, 44, f_) == 44;
    }
    static void le(unsigned char *p, uint32_t v, int bytes) { for (int i = 0; i < bytes; ++i) p[i] = (unsigned char)(v >> (8 * i)); }
    void put32(long at, uint32_t v) { unsigned char b[4]; le(b, v, 4); std::fseek(f_, at, SEEK_SET); (void)std::fwrite(b, 1, 4, f_); }
    std::string base_;
    long long limit_, size_ = 0;
    int seq_ = 0;
    std::FILE *f_ = nullptr;
    std::vector<int16_t> conv_;
};
typedef WavWriter AudioSinkWAV;
