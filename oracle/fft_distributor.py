"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's waterfall line pacing,
src/process/FFTDataDistributor.cpp:28-144 (integer offsets + double accumulators; the samples are only copied).
Only tests/ may import this.  Parity pins: the reference holds no golden vectors for this path ("parity unpinned" at the
CubicSDR level); this restatement follows the reference statement by statement and the C++ host mirror
(cubicsdr_amd/host/FFTDataDistributor.h) is compared against it.
"""

FFT_DISTRIBUTOR_BUFFER_IN_SECONDS = 0.250     # CubicSDRDefs.h:69


class FFTDataDistributorRef:
    def __init__(self, fft_size=2048, lines_per_second=30):            # ctor :11-13, CubicSDRDefs.h:44,56
        self.fft_size = int(fft_size)
        self.lps = int(lines_per_second)
        self.accum = 0.0                                               # lineRateAccum
        self.rate = 0
        self.freq = 0
        self.buf = []                                                  # inputBuffer.data[bufferOffset : bufferOffset + bufferedItems]
        self.buffer_max = 0
        self.offset = 0                                                # bufferOffset (only matters for the overflow test :68)

    def push(self, data, freq, rate):
        """one popped input (:41-143); data: list of sample ids.  Returns the emitted lines [(first id, n, freq, rate)]."""
        if self.rate != rate or self.freq != freq:                     # :43-55
            self.buffer_max = max(int(rate * FFT_DISTRIBUTOR_BUFFER_IN_SECONDS), int(1.2 * self.fft_size))
            self.offset = 0
            self.buf = []
            self.rate, self.freq = rate, freq
        if self.buffer_max < int(1.2 * self.fft_size):                 # :58-61
            self.buffer_max = int(1.2 * self.fft_size)
        n_add = len(data)
        if self.offset + len(self.buf) + len(data) > self.buffer_max:  # :68-79
            self.offset = 0
            if len(self.buf) + len(data) > self.buffer_max:
                n_add = self.buffer_max - len(self.buf)
        self.buf.extend(data[:n_add])                                  # :82-83
        items = len(self.buf)
        fft = self.fft_size
        input_time = float(items) / float(self.rate)                   # :92
        input_lines = float(items) / float(fft)                        # :94
        step = (float(self.lps) * input_time) / input_lines            # :99
        out = []
        if items >= fft:                                               # :102
            processed = 0
            if self.accum + step * (float(items) / float(fft)) < 1.0:  # :104-107
                self.accum += step * (float(items) / float(fft))
                processed = items
            else:
                i = 0
                while i < items:                                       # :109-131
                    if i + fft > items:
                        break
                    self.accum += step
                    if self.accum >= 1.0:
                        out.append((self.buf[i], fft, self.freq, self.rate))
                        while self.accum >= 1.0:
                            self.accum -= 1.0
                    processed += fft
                    i += fft
            if processed:                                              # :135-138
                self.buf = self.buf[processed:]
                self.offset += processed
            if not self.buf:                                           # :140-143
                self.offset = 0
        return out
