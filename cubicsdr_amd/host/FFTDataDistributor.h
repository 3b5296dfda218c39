// FFTDataDistributor.h -- waterfall line pacing: buffers the full-rate IQ stream and cuts it into fftSize-sample
// "lines" at linesPerSecond; API and behaviour of the reference's src/process/FFTDataDistributor.{h,cpp} (own
// implementation).  K19 in SURVEY.md: integer offsets + double accumulators, the samples themselves are only copied.
//
// Behaviour kept from the reference (FFTDataDistributor.cpp:28-144):
//  * a change of sample rate or frequency drops everything buffered and resizes the buffer to
//    max(0.25 s of samples, 1.2 * fftSize) (:43-55); an fftSize change alone only grows it (:58-61)
//  * incoming samples that do not fit are dropped from the END of the incoming block, after compacting (:68-79)
//  * lineRateStep = linesPerSecond * inputTime / inputLines, accumulated once per whole line; a line is emitted when the
//    accumulator reaches 1 (then wrapped below 1); if even all buffered lines together stay below 1 the buffer is
//    skipped outright, partial line included (:89-104)
//  * lines go out with a non-blocking push: a full consumer queue loses the line (:121)
#pragma once
#include <algorithm>
#include <atomic>
#include <cstring>

#include "DataTypes.h"
#include "IOThread.h"
#include "VisualProcessor.h"

#ifndef HEARTBEAT_CHECK_PERIOD_MICROS
#define HEARTBEAT_CHECK_PERIOD_MICROS (50 * 1000)
#endif
#define DEFAULT_FFT_SIZE 2048                        // CubicSDRDefs.h:44
#define DEFAULT_WATERFALL_LPS 30                     // CubicSDRDefs.h:56
#define FFT_DISTRIBUTOR_BUFFER_IN_SECONDS 0.250      // CubicSDRDefs.h:69

class FFTDataDistributor : public VisualProcessor<DemodulatorThreadIQData, DemodulatorThreadIQData> {
public:
    FFTDataDistributor() : fftSize(DEFAULT_FFT_SIZE), linesPerSecond(DEFAULT_WATERFALL_LPS) {}
    void setFFTSize(unsigned int size) { fftSize.store(size); }
    void setLinesPerSecond(unsigned int lines) { linesPerSecond = lines; }
    unsigned int getLinesPerSecond() const { return linesPerSecond; }

    // test hooks (not in the reference): the pacing state
    double lineRateAccumulator() const { return lineRateAccum; }
    size_t buffered() const { return count; }

protected:
    void process() override {
        while (!input->empty()) {
            if (!isAnyOutputEmpty()) return;                         // every consumer queue is full: try again later
            DemodulatorThreadIQDataPtr inp;
            if (!input->pop(inp, HEARTBEAT_CHECK_PERIOD_MICROS) || !inp) continue;
            take(*inp);
            cutLines();
        }
    }

private:
    // append one incoming block to the stream buffer
    void take(const DemodulatorThreadIQData &inp) {
        const size_t fft = fftSize.load();
        if (stream.sampleRate != inp.sampleRate || stream.frequency != inp.frequency) {
            capacity = std::max((size_t)(inp.sampleRate * FFT_DISTRIBUTOR_BUFFER_IN_SECONDS), (size_t)(1.2 * fft));
            head = 0; count = 0;
            stream.sampleRate = inp.sampleRate;
            stream.frequency = inp.frequency;
            stream.data.resize(capacity);
        }
        if (capacity < (size_t)(1.2 * fft)) { capacity = (size_t)(1.2 * fft); stream.data.resize(capacity); }
        size_t n = inp.data.size();
        if (head + count + n > capacity) {
            std::memmove(stream.data.data(), stream.data.data() + head, count * sizeof(liquid_float_complex_t));
            head = 0;
            if (count + n > capacity) n = capacity - count;          // overflow: the tail of the incoming block is lost
        }
        if (n) std::memcpy(stream.data.data() + head + count, inp.data.data(), n * sizeof(liquid_float_complex_t));
        count += n;
    }
    // emit the lines the pacing allows and consume what was looked at
    void cutLines() {
        const size_t fft = fftSize.load();
        if (!fft || count < fft) return;
        const double inputTime = (double)count / (double)stream.sampleRate;
        const double inputLines = (double)count / (double)fft;
        const double step = ((double)linesPerSecond * inputTime) / inputLines;
        size_t used = 0;
        if (lineRateAccum + step * inputLines < 1.0) {
            lineRateAccum += step * inputLines;
            used = count;
        } else {
            for (size_t at = 0; at + fft <= count; at += fft) {
                lineRateAccum += step;
                if (lineRateAccum >= 1.0) {
                    DemodulatorThreadIQDataPtr line = lineBuffers.getBuffer();
                    line->frequency = stream.frequency;
                    line->sampleRate = stream.sampleRate;
                    line->data.assign(stream.data.begin() + head + at, stream.data.begin() + head + at + fft);
                    distribute(line, NON_BLOCKING_TIMEOUT);
                    while (lineRateAccum >= 1.0) lineRateAccum -= 1.0;
                }
                used += fft;
            }
        }
        count -= used; head += used;
        if (count == 0) head = 0;
    }

    DemodulatorThreadIQData stream;                  // inputBuffer
    ReBuffer<DemodulatorThreadIQData> lineBuffers{"FFTDataDistributorBuffers"};
    std::atomic<unsigned int> fftSize;
    unsigned int linesPerSecond;
    double lineRateAccum = 0.0;
    size_t capacity = 0, head = 0, count = 0;        // bufferMax, bufferOffset, bufferedItems
};
