// DataTypes.h -- the message structs carried through the queues of the hot path.  Field names and meaning follow the
// reference: SDRThreadIQData (src/sdr/SoapySDRThread.h:20-42), DemodulatorThreadIQData (src/demod/DemodDefs.h:18-38),
// AudioThreadInput (src/audio/AudioThread.h:16-51), SpectrumVisualData (src/process/SpectrumVisualProcessor.h:14-23).
#pragma once
#include <complex>
#include <memory>
#include <vector>

#include "ThreadBlockingQueue.h"

struct liquid_float_complex_t { float real, imag; };     // layout of liquid_float_complex (liquid.h:149-157)

// Where a block's samples ALSO live in HBM once the ingest has moved them there (null: host copy only).  Consumers that run on the
// device read this address instead of uploading `data` again (the reference hands ONE buffer to all consumers, SDRPostThread.cpp:227-245).
// `hold` keeps the HBM slot from being overwritten while any block that points into it is alive -- the ReBuffer rule (use_count() == 1
// means free, IOThread.h:62-110) applied to device slots.
struct DeviceResidentIQ {
    const float *deviceData = nullptr;
    size_t deviceSamples = 0;
    std::shared_ptr<void> deviceHold;
    void shareDeviceCopy(const DeviceResidentIQ &o) { deviceData = o.deviceData; deviceSamples = o.deviceSamples; deviceHold = o.deviceHold; }
    void dropDeviceCopy() { deviceData = nullptr; deviceSamples = 0; deviceHold.reset(); }
};

class SDRThreadIQData : public DeviceResidentIQ {
public:
    long long frequency = 0;
    long long sampleRate = 0;
    bool dcCorrected = true;
    int numChannels = 0;
    bool iqSwapPending = false;               // internal to the block assembler: `data` still holds Q, I (exchanged during the transfer)
    std::vector<liquid_float_complex_t> data;
    virtual ~SDRThreadIQData() = default;
};
typedef std::shared_ptr<SDRThreadIQData> SDRThreadIQDataPtr;
typedef ThreadBlockingQueue<SDRThreadIQDataPtr> SDRThreadIQDataQueue;
typedef std::shared_ptr<SDRThreadIQDataQueue> SDRThreadIQDataQueuePtr;

class DemodulatorThreadIQData : public DeviceResidentIQ {
public:
    long long frequency = 0;
    long long sampleRate = 0;
    std::vector<liquid_float_complex_t> data;
    virtual ~DemodulatorThreadIQData() = default;
};
typedef std::shared_ptr<DemodulatorThreadIQData> DemodulatorThreadIQDataPtr;
typedef ThreadBlockingQueue<DemodulatorThreadIQDataPtr> DemodulatorThreadInputQueue;
typedef std::shared_ptr<DemodulatorThreadInputQueue> DemodulatorThreadInputQueuePtr;

class AudioThreadInput {
public:
    long long frequency = 0;
    int inputRate = 0;
    int sampleRate = 0;
    int channels = 0;
    float peak = 0;
    int type = 0;
    bool is_squelch_active = false;
    std::vector<float> data;
    virtual ~AudioThreadInput() = default;
};
typedef std::shared_ptr<AudioThreadInput> AudioThreadInputPtr;
typedef ThreadBlockingQueue<AudioThreadInputPtr> AudioThreadInputQueue;
typedef std::shared_ptr<AudioThreadInputQueue> AudioThreadInputQueuePtr;

class SpectrumVisualData {
public:
    std::vector<float> spectrum_points;
    std::vector<float> spectrum_hold_points;
    double fft_ceiling = 0, fft_floor = 0;
    long long centerFreq = 0;
    int bandwidth = 0;
    virtual ~SpectrumVisualData() = default;
};
typedef std::shared_ptr<SpectrumVisualData> SpectrumVisualDataPtr;
typedef ThreadBlockingQueue<SpectrumVisualDataPtr> SpectrumVisualDataQueue;
typedef std::shared_ptr<SpectrumVisualDataQueue> SpectrumVisualDataQueuePtr;
