"""CPU ORACLE (TEST INFRASTRUCTURE ONLY): restatement of CubicSDR's hot-path control flow around liquid-dsp.

Each class follows one reference object and calls the liquid functions that object calls, in the same order,
through oracle/liquid_api.py -- so it runs either on the reference's own liquid-dsp 1.5.0 binary (backend "ref")
or on the C restatement oracle/liquid_port.c (backend "port").  File:line citations are into /root/reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import math

import numpy as np

from . import liquid_api as A


def _cptr(a):
    return a.ctypes.data_as(C.c_void_p)


_LIBM = C.CDLL("libm.so.6")
_LIBM.log10f.restype = C.c_float; _LIBM.log10f.argtypes = [C.c_float]
_LIBM.powf.restype = C.c_float; _LIBM.powf.argtypes = [C.c_float, C.c_float]


class RefSDRPost:
    """SDRPostThread (src/sdr/SDRPostThread.cpp): runSingleCH :248-299, runPFBCH :416-455, runDemodChannels :303-398."""

    def __init__(self, backend, sample_rate, num_channels, oversampled=False):
        self.L = A.load(backend)
        self.sample_rate = int(sample_rate)
        self.M = int(num_channels)
        self.dc = self.L.iirfilt_crcf_create_dc_blocker(0.0005)          # :29
        self.frequency = 0
        self.oversampled = bool(oversampled) and self.M > 1              # chanMode 2: runPFBCH2 :472-512
        if self.oversampled:
            self.chan = self.L.firpfbch2_crcf_create_kaiser(A.LIQUID_ANALYZER, self.M, 4, 60.0)  # :463
            self.chan_bw = self.sample_rate // self.M                    # :465
        elif self.M > 1:
            self.chan = self.L.firpfbch_crcf_create_kaiser(A.LIQUID_ANALYZER, self.M, 4, 60.0)   # :406
            self.chan_bw = self.sample_rate // self.M                    # :408 (integer division)
        else:
            self.chan = None
            self.chan_bw = self.sample_rate
        self.centers = [0] * (self.M + 1)
        self.data_out = None

    def update_channels(self):                                           # :116-124
        M = self.M
        for i in range(M // 2):
            ofs = self.chan_bw * i
            self.centers[i] = self.frequency + ofs
            self.centers[i + M // 2] = self.frequency - (self.sample_rate // 2) + ofs
        self.centers[M] = self.frequency + (self.sample_rate // 2)

    def channel_at(self, f):                                             # :128-139
        if self.M == 1:
            return 0
        chan, min_delta = -1, self.sample_rate
        for i in range(self.M + 1):
            d = abs(f - self.centers[i])
            if d < min_delta:
                min_delta, chan = d, i
        return chan

    def run_block(self, x, frequency):
        """x: complex64[block_len].  Leaves channelizer output (time-major) in self.data_out."""
        x = A.as_c64(x)
        self.frequency = int(frequency)
        if self.M == 1:
            y = np.empty_like(x)
            self.L.iirfilt_crcf_execute_block(self.dc, _cptr(x), x.size, _cptr(y))      # :284
            self.data_out = y
            return
        self.update_channels()
        if self.oversampled:                                             # :493-507: dataOut holds 2x the input
            y = np.empty(2 * x.size, np.complex64)
            self.L.oracle_firpfbch2_block(C.c_void_p(self.chan), self.M, _cptr(x), x.size // (self.M // 2), _cptr(y))
            self.data_out = y
            return
        y = np.empty_like(x)
        self.L.oracle_firpfbch_analyzer_block(C.c_void_p(self.chan), self.M, _cptr(x), x.size // self.M, _cptr(y))   # :449-451
        self.data_out = y

    def channel_data(self, i):
        """runDemodChannels :341-382 -> (samples, centre frequency, sample rate) for channel index i in [0, M]."""
        if self.M == 1:
            return self.data_out, self.frequency, self.sample_rate
        idx = self.M // 2 if i == self.M else i                         # :359-361
        d = np.ascontiguousarray(self.data_out[idx::self.M])
        if i == 0:                                                       # :364-375
            y = np.empty_like(d)
            self.L.iirfilt_crcf_execute_block(self.dc, _cptr(d), d.size, _cptr(y))
            d = y
        return d, self.centers[i], (2 * self.chan_bw if self.oversampled else self.chan_bw)   # runDemodChannels(chanBw * 2) :510


class RefDemod:
    """DemodulatorPreThread::run (src/demod/DemodulatorPreThread.cpp:154-220) + DemodulatorThread::run
    (src/demod/DemodulatorThread.cpp:119-233) + the analog modems (src/modules/modem/)."""

    def __init__(self, backend, modem, bandwidth, frequency, chan_rate, audio_rate=48000, demph=75, pilot_sos=None):
        """demph / pilot_sos: FM stereo only -- the "demph" setting (microseconds, 0 = none) and, when given, (b15, a15) second-order
        sections to build the pilot band-pass from (iirfilt_crcf_create_sos) instead of the reference's own design call"""
        L = self.L = A.load(backend)
        self.backend = backend
        self.modem = modem
        self.frequency = int(frequency)
        self.audio_rate = int(audio_rate)
        bw = max(int(bandwidth), 500)                                    # checkSampleRate, ModemAnalog.cpp:14-19
        if modem in ("USB", "LSB") and bw % 2:
            bw += 1                                                      # ModemUSB.cpp:29-37
        if modem == "I/Q":
            bw = int(audio_rate)                                         # ModemIQ.cpp:31-33
        if modem == "FMS":
            bw = max(int(bandwidth), 100000)                             # ModemFMStereo.cpp:27-35
        self.cw_offset = 650.0                                           # mBeepFrequency, ModemCW.cpp:17
        self.bandwidth = bw
        self.chan_rate = int(chan_rate)
        self.nco = L.nco_crcf_create(A.LIQUID_VCO)                       # DemodulatorPreThread.cpp:22
        self.shift = None
        self.iq_ratio = float(bw) / float(chan_rate)                     # DemodulatorWorkerThread.cpp:99
        self.resamp = L.msresamp_crcf_create(self.iq_ratio, 60.0)        # :100
        self.au_ratio = float(audio_rate) / float(bw)                    # ModemAnalog.cpp:29
        self.au = L.msresamp_rrrf_create(self.au_ratio, 60.0)            # :30
        self.ceil, self.ceil_ma, self.ceil_maa = 1.0, 1.0, 1.0           # ModemAnalog ctor
        self.use_signal_output = modem in ("AM", "USB", "LSB", "DSB")
        if modem in ("NBFM", "FM"):
            self.fm = L.freqdem_create(0.5)                              # ModemNBFM.cpp:7
        elif modem == "I/Q":
            pass                                                         # ModemIQ::buildKit: no DSP objects
        elif modem == "FMS":                                             # ModemFMStereo::buildKit, ModemFMStereo.cpp:91-162
            f32 = np.float32
            self.fm = L.freqdem_create(0.5)                              # :7
            self.au2 = L.msresamp_rrrf_create(self.au_ratio, 60.0)       # stereoResampler :103
            fcut = float(f32(16000.0) / f32(audio_rate))                 # :106
            ft = float(f32(1000.0) / f32(audio_rate))                    # :108
            fcut = min(max(fcut, 0.0), 0.5)
            h_len = L.estimate_req_filter_len(ft, 60.0)                  # :120
            h = np.zeros(h_len, np.float32)
            L.liquid_firdes_kaiser(h_len, fcut, 60.0, 0.0, A.ptr(h))     # :122
            self.fir_l = L.firfilt_rrrf_create(A.ptr(h), h_len)          # :124-125
            self.fir_r = L.firfilt_rrrf_create(A.ptr(h), h_len)
            bwf = f32(max(float(f32(bw)), 100000.0))                     # :128-131
            f0 = float(f32(19000) / bwf)
            fc = float(f32(19500) / bwf)
            if pilot_sos is None:
                self.pilot = L.iirfilt_crcf_create_prototype(A.LIQUID_IIRDES_CHEBY2, A.LIQUID_IIRDES_BANDPASS, A.LIQUID_IIRDES_SOS,
                                                             5, fc, f0, 1.0, 60.0)     # :138
            else:
                b15 = A.as_f32(pilot_sos[0]).copy(); a15 = A.as_f32(pilot_sos[1]).copy()
                self.pilot = L.iirfilt_crcf_create_sos(A.ptr(b15), A.ptr(a15), 5)
            self.r2c = L.firhilbf_create(5, 60.0)                        # :140-141
            self.c2r = L.firhilbf_create(5, 60.0)
            self.pll = L.nco_crcf_create(A.LIQUID_VCO)                   # :143-145
            L.nco_crcf_reset(self.pll)
            L.nco_crcf_pll_set_bandwidth(self.pll, 0.25)
            self.demph = int(demph)
            self.dem_l = self.dem_r = None
            if self.demph:                                               # :149-158
                f = 1.0 / (2.0 * math.pi * float(self.demph) * 1e-6)
                t = 1.0 / (2.0 * math.pi * f)
                t = 1.0 / (2.0 * float(audio_rate) * math.tan(1.0 / (2.0 * float(audio_rate) * t)))
                tb = 1.0 + 2.0 * t * float(audio_rate)
                bd = np.array([1.0 / tb, 1.0 / tb], np.float32)
                ad = np.array([1.0, (1.0 - 2.0 * t * float(audio_rate)) / tb], np.float32)
                self.dem_l = L.iirfilt_rrrf_create(A.ptr(bd), 2, A.ptr(ad), 2)
                self.dem_r = L.iirfilt_rrrf_create(A.ptr(bd), 2, A.ptr(ad), 2)
        elif modem == "CW":
            self.cw_lo = L.nco_crcf_create(A.LIQUID_NCO)                 # ModemCW.cpp:22
            self.cw_hilb = L.firhilbf_create(5, 60.0)                    # :23
            self.cw_resamp = L.msresamp_cccf_create(float(np.float32(self.au_ratio)), 60.0)   # buildKit :124
            self.cw_gain = np.float32(15.0)                              # mGain :18 (overwritten while mAutoGain is on)
            self.use_signal_output = True                                # :24
        elif modem == "DSB":
            self.dsb = L.ampmodem_create(0.5, 0, 1)                      # ModemDSB.cpp:6 (LIQUID_AMPMODEM_DSB = 0, suppressed carrier)
        elif modem == "AM":
            self.dcb = L.firfilt_rrrf_create_dc_blocker(25, 30.0)        # ModemAM.cpp:9
        else:
            self.ssb_filt = L.iirfilt_crcf_create_lowpass(6, 0.25)       # ModemUSB.cpp:8
            self.ssb_nco = L.nco_crcf_create(A.LIQUID_NCO)               # :9
            L.nco_crcf_set_frequency(self.ssb_nco, float(np.float32((2.0 * math.pi) * 0.25)))   # :10
            self.hilb = L.firhilbf_create(5, 90.0)                       # :11

    def state(self):
        """the integer state of the front-end as it stands (after the blocks run so far): the oscillator's phase word, the arbitrary
        resampler's 24-bit phase and the half-band input fill of msresamp_crcf -- what the HIP path reports per block as bit-exact items"""
        be = self.backend
        th, _ = A.nco_state(be, self.nco)
        m = A.msresamp_state(be, self.resamp)
        return dict(nco_theta=th, resamp_phase=m["phase"], buffer_index=m["buffer_index"], S=m["S"], step=m["step"])

    def pre(self, data, in_freq, in_rate):
        """NCO shift + decimate; returns resampled IQ or None when the block is skipped (:154-165)."""
        L = self.L
        shift = self.frequency - int(in_freq)
        bound = int(float(in_rate // 2) * 1.5)
        if shift != self.shift:
            self.shift = shift
            if abs(shift) <= bound:
                L.nco_crcf_set_frequency(self.nco, float(np.float32((2.0 * math.pi) * (float(abs(shift)) / float(in_rate)))))
        if abs(shift) > bound:
            return None
        x = A.as_c64(data).copy()
        if shift != 0:                                                   # :186-195
            y = np.empty_like(x)
            if shift < 0:
                L.nco_crcf_mix_block_up(self.nco, _cptr(x), _cptr(y), x.size)
            else:
                L.nco_crcf_mix_block_down(self.nco, _cptr(x), _cptr(y), x.size)
            x = y
        out = np.empty(int(math.ceil(x.size * self.iq_ratio)) + 512, np.complex64)   # :199
        nw = C.c_uint()
        L.msresamp_crcf_execute(self.resamp, _cptr(x), x.size, _cptr(out), C.byref(nw))   # :209
        return out[:nw.value].copy()

    def demodulate(self, iq):
        """Modem::demodulate + buildAudioOutput -> dict(audio, level_accum, level_count, peak, demod)"""
        L = self.L
        n = iq.size
        if n == 0:
            return None
        iq = A.as_c64(iq)
        if self.modem == "I/Q":                                          # ModemIQ.cpp:41-61: stereo (imag, real), 2 channels
            audio = np.empty(2 * n, np.float32)
            audio[0::2] = iq.imag
            audio[1::2] = iq.real
            accum = float(np.sum(np.sqrt(iq.real.astype(np.float64) ** 2 + iq.imag.astype(np.float64) ** 2)))
            return dict(audio=audio, level_accum=accum, level_count=n, peak=float(np.max(np.abs(audio))), demod=audio.copy(), channels=2)
        if self.modem == "FMS":                                          # ModemFMStereo::demodulate, ModemFMStereo.cpp:163-289
            d = np.empty(n, np.float32)
            L.freqdem_demodulate_block(self.fm, _cptr(iq), n, _cptr(d))                  # :178
            cap = int(math.ceil(n * self.au_ratio)) + 512                               # :176
            mono = np.empty(cap, np.float32)
            nw = C.c_uint()
            L.msresamp_rrrf_execute(self.au, _cptr(d), n, _cptr(mono), C.byref(nw))      # :189
            st = np.empty(n, np.float32)
            th = np.empty(n, np.uint32)
            L.oracle_fms_pilot_block(C.c_void_p(self.r2c), C.c_void_p(self.pilot), C.c_void_p(self.pll), C.c_void_p(self.c2r),
                                     _cptr(d), n, _cptr(st), _cptr(th))                  # :198-226
            ster = np.empty(cap, np.float32)
            nw2 = C.c_uint()
            L.msresamp_rrrf_execute(self.au2, _cptr(st), n, _cptr(ster), C.byref(nw2))   # :236
            m = nw2.value                                                                # numAudioWritten of the SECOND call sizes the output (:238-262)
            audio = np.empty(2 * m, np.float32)
            L.oracle_fms_matrix_block(C.c_void_p(self.dem_l) if self.dem_l else None, C.c_void_p(self.dem_r) if self.dem_r else None,
                                      C.c_void_p(self.fir_l), C.c_void_p(self.fir_r), _cptr(mono), _cptr(ster), m, _cptr(audio))   # :263-287
            accum = float(np.sum(np.sqrt(iq.real.astype(np.float64) ** 2 + iq.imag.astype(np.float64) ** 2)))
            peak = float(np.max(np.abs(audio))) if m else 0.0
            return dict(audio=audio, level_accum=accum, level_count=n, peak=peak, demod=d.copy(), channels=2,
                        fms_stereo=st.copy(), fms_theta=th.copy(), fms_mono_audio=mono[:m].copy(), fms_stereo_audio=ster[:m].copy())
        if self.modem == "CW":                                           # ModemCW::demodulate :155-209
            f32 = np.float32
            cx = np.empty(int(math.ceil(n * self.au_ratio)) + 512, np.complex64)      # initOutputBuffers :140
            nw = C.c_uint()
            L.msresamp_cccf_execute(self.cw_resamp, _cptr(iq), n, _cptr(cx), C.byref(nw))   # :163
            m = nw.value
            L.nco_crcf_set_frequency(self.cw_lo, float(f32(f32(2.0) * f32(math.pi) * f32(self.cw_offset) / f32(self.audio_rate))))   # :171
            d = np.empty(m, np.float32)
            L.oracle_cw_block(C.c_void_p(self.cw_lo), C.c_void_p(self.cw_hilb), _cptr(np.ascontiguousarray(cx[:m])), m, _cptr(d))   # :174-178
            demod_unscaled = d.copy()
            self.ceil_ma = f32(self.ceil_ma + f32(f32(self.ceil - self.ceil_ma) * f32(0.025)))            # :182-183
            self.ceil_maa = f32(self.ceil_maa + f32(f32(self.ceil_ma - self.ceil_maa) * f32(0.025)))
            self.ceil = f32(max(0.0, float(d.max()))) if m else f32(0)                                    # :184-190 (signed maximum from 0)
            # std::log10(float) / std::pow(float, float) are the C library's log10f / powf (numpy's float32 routines differ in the last place)
            self.cw_gain = f32(f32(10.0) * f32(_LIBM.log10f(float(f32(f32(0.5) / self.ceil_maa)))))        # :192
            audio = (d * f32(_LIBM.powf(10.0, float(f32(self.cw_gain / f32(10.0)))))).astype(np.float32)   # :196-198
            accum = float(np.sum(np.abs(audio.astype(np.float64))))
            peak = float(np.max(np.abs(audio))) if m else 0.0
            return dict(audio=audio, level_accum=accum, level_count=m, peak=peak, demod=demod_unscaled)
        d = np.empty(n, np.float32)
        if self.modem in ("NBFM", "FM"):
            L.freqdem_demodulate_block(self.fm, _cptr(iq), n, _cptr(d))                  # ModemNBFM.cpp:36
            autogain = False
        elif self.modem == "DSB":
            L.oracle_dsb_block(C.c_void_p(self.dsb), _cptr(iq), n, _cptr(d))             # ModemDSB.cpp:49-51
            autogain = True
        elif self.modem == "AM":
            L.oracle_am_block(C.c_void_p(self.dcb), _cptr(iq), n, _cptr(d))              # ModemAM.cpp:41-47
            autogain = True
        else:
            L.oracle_ssb_block(C.c_void_p(self.ssb_nco), C.c_void_p(self.ssb_filt), C.c_void_p(self.hilb),
                               1 if self.modem == "USB" else 0, _cptr(iq), n, _cptr(d))  # ModemUSB.cpp:54-61
            autogain = True
        demod_unscaled = d.copy()
        if autogain:                                                     # ModemAnalog.cpp:70-86 (float arithmetic)
            f32 = np.float32
            self.ceil_ma = f32(self.ceil_ma + f32(f32(self.ceil - self.ceil_ma) * f32(0.025)))
            self.ceil_maa = f32(self.ceil_maa + f32(f32(self.ceil_ma - self.ceil_maa) * f32(0.025)))
            self.ceil = f32(max(0.0, float(d.max())))
            gain = f32(0.5) / f32(self.ceil_maa)
            d = (d * gain).astype(np.float32)
        out = np.empty(int(math.ceil(n * self.au_ratio)) + 512, np.float32)               # ModemAnalog.cpp:51
        nw = C.c_uint()
        L.msresamp_rrrf_execute(self.au, _cptr(d), n, _cptr(out), C.byref(nw))           # :88
        audio = out[:nw.value].copy()
        # DemodulatorThread.cpp:142-152 (double accumulation of magnitudes)
        if self.use_signal_output:
            accum = float(np.sum(np.abs(audio.astype(np.float64))))
            cnt = audio.size
        else:
            accum = float(np.sum(np.sqrt(iq.real.astype(np.float64) ** 2 + iq.imag.astype(np.float64) ** 2)))
            cnt = n
        peak = float(np.max(np.abs(audio))) if audio.size else 0.0      # :223-233
        return dict(audio=audio, level_accum=accum, level_count=cnt, peak=peak, demod=demod_unscaled)


class RefLevelSquelch:
    """DemodulatorThread::run's level / floor / ceil / squelch bookkeeping (src/demod/DemodulatorThread.cpp:142-220), one step per
    block, in the reference's types: float32 trackers, double currentSignalLevel (TEST INFRASTRUCTURE: pins cubicsdr_amd/host/DemodLevel.h)."""

    def __init__(self):
        f32 = np.float32
        self.level, self.floor, self.ceil = f32(-100.0), f32(-30.0), f32(30.0)     # ctor :20-27
        self.squelch_break = False

    @staticmethod
    def linear_to_db(x):                                                    # :59-67
        return 20.0 * math.log10(max(x, 1e-20))

    def step(self, have_level, accum, count, sample_time, squelch_enabled, squelch_level):
        f32 = np.float32
        cur = 0.0
        if have_level:
            cur = self.linear_to_db(accum / float(count))                   # :152 / :162
            sf, sc, sl = f32(self.floor), f32(self.ceil), f32(squelch_level)
            if cur > float(sc):
                sc = f32(cur)
            if cur < float(sf):
                sf = f32(cur)
            if float(f32(sl + f32(1.0))) > float(sc):
                sc = f32(sl + f32(1.0))
            if float(f32(sf + f32(2.0))) > float(sc):
                sc = f32(sf + f32(2.0))
            sc = f32(float(sc) - (float(sc) - (cur + 2.0)) * sample_time * float(f32(0.05)))       # double arithmetic, stored as float
            sf = f32(float(sf) + ((cur - 5.0) - float(sf)) * sample_time * float(f32(0.15)))
            self.floor, self.ceil = sf, sc
        lvl = float(self.level)
        if cur > lvl:
            lvl = lvl + (cur - lvl) * 0.5
        else:
            lvl = lvl + (cur - lvl) * 0.05 * sample_time * 30.0
        self.level = f32(lvl)
        squelched = bool(squelch_enabled) and float(self.level) < float(f32(squelch_level))
        if squelch_enabled:
            if not squelched and not self.squelch_break:
                self.squelch_break = True
            elif squelched and self.squelch_break:
                self.squelch_break = False
        return squelched


class RefSpectrum:
    """SpectrumVisualProcessor::process, full-span view (src/process/SpectrumVisualProcessor.cpp:387-576, 626-627)."""

    def __init__(self, backend, fft_size, average_rate=0.65, scale=1.0):
        self.L = A.load(backend)
        self.F = int(fft_size)
        self.N = 2 * self.F                                              # SPECTRUM_VZM, .h:11, .cpp:145
        self.x = np.zeros(self.N, np.complex64)
        self.y = np.zeros(self.N, np.complex64)
        self.plan = self.L.fft_create_plan(self.N, _cptr(self.x), _cptr(self.y), A.LIQUID_FFT_FORWARD, 0)   # :177
        self.ma = np.zeros(self.N, np.float64)
        self.maa = np.zeros(self.N, np.float64)
        self.ceil_ma = self.ceil_maa = 100.0                             # :32
        self.floor_ma = self.floor_maa = 0.0                             # :33
        self.rate = float(np.float32(average_rate))                      # float member, :36
        self.sf = float(np.float32(scale))

    def select_input(self, data):
        """Frame selection of process() for one popped input, full-span view (:387-421): returns the 2*fftSize samples that
        are transformed, or None when the input only primes fftLastData."""
        N = self.N
        if not hasattr(self, "last"):
            self.last = np.zeros(N, np.complex64)                        # fftLastData
            self.last_size = 0                                           # lastDataSize, setup :166
        data = A.as_c64(data)
        n = min(len(data), N)
        fin = np.zeros(N, np.complex64)                                  # fftInData: data, zero padded (:388-396)
        fin[:n] = data[:n]
        num_written = len(data)
        if num_written >= N:                                             # :401-404
            self.last[:] = fin
            return fin
        if self.last_size + num_written < N:                             # priming :406-412
            num_copy = max(N - self.last_size, num_written)
            self.last[:num_copy] = fin[:num_copy]
            self.last_size += num_copy
            return None
        num_last = N - num_written                                       # :413-419
        frame = np.concatenate([self.last[self.last_size - num_last:self.last_size], fin[:num_written]])
        self.last[:] = frame
        return frame

    def fft(self, frame):
        self.x[:] = A.as_c64(frame)
        self.L.fft_execute(self.plan)                                    # :439
        return self.y.copy()

    # ---- peak hold and DC hiding (setPeakHold :115-125, reset :264-273, setHideDC :204-209) -------------------------
    def set_peak_hold(self, on):
        if getattr(self, "peak_hold", False) and on:
            self.peak_reset = 30                                         # PEAK_RESET_COUNT, .h:12
        else:
            self.peak_hold = bool(on)
            self.peak_reset = 1

    def set_hide_dc(self, on, center_freq, bandwidth, input_freq):
        self.hide_dc, self.center_freq, self.bandwidth, self.input_freq = bool(on), int(center_freq), int(bandwidth), int(input_freq)

    def begin_input(self):
        """the head of process() for one popped input (:247, :264-273): returns doPeak for this input"""
        if not hasattr(self, "peak_hold"):
            self.peak_hold, self.peak_reset = False, 0
        do_peak = self.peak_hold and self.peak_reset == 0
        if self.peak_reset != 0:
            self.peak_reset -= 1
            if self.peak_reset == 0:
                self.peak = np.full(self.N, self.floor_maa, np.float64)
                self.ceil_peak = self.floor_maa
                self.floor_peak = self.ceil_maa
        self.do_peak = do_peak
        return do_peak

    def _hide_dc(self, pts):
        """:578-623, the reference's integer arithmetic (C division truncates toward zero; operands here are positive)"""
        F = self.F
        fmin = self.center_freq - self.bandwidth // 2
        fmax = self.center_freq + self.bandwidth // 2
        zero_pt = self.input_freq - fmin
        if not (fmin < self.input_freq < fmax):
            return
        step = int(fmax - fmin) // F
        start = zero_pt // step - 2000 // step
        end = zero_pt // step + 2000 // step
        if end - start < 2:
            end += 1
            start -= 1
        steps = end - start
        half = start + steps // 2
        if end + steps // 2 + 1 < F and start - steps // 2 - 1 >= 0 and end > start:
            n = 1
            for i in range(start, half):
                pts[2 * i + 1] = pts[2 * (start - n) + 1]
                n += 1
            n = 1
            for i in range(half, end):
                pts[2 * i + 1] = pts[2 * (end + n) + 1]
                n += 1

    def process_input(self, data, frequency=None, sample_rate=None):
        """one process() call: returns None (no FFT ran) or (points, fft_ceiling, fft_floor, hold_points or None).
        With set_view(True, ...) the input goes through the zoomed-view branch (:283-386) first."""
        self.begin_input()
        if getattr(self, "is_view", False):
            data = self._view_input(A.as_c64(data), int(frequency), int(sample_rate))
            if data is None:
                self.last_view = True
                return None
        frame = self.select_input(data)
        if frame is None:
            self.last_view = getattr(self, "is_view", False)
            return None
        pts, ce, fl = self.process_frame(frame)
        self.last_view = getattr(self, "is_view", False)                 # :631
        return pts, ce, fl, self.hold

    # ---- zoomed view (setView :64-72, process :283-386, :454-492, :532-560) -------------------------------------------
    def set_view(self, on, center_freq=None, bandwidth=None):
        self.is_view = bool(on)
        if center_freq is not None:
            self.center_freq = int(center_freq)
        if bandwidth is not None:
            self.bandwidth = int(bandwidth)
        if not hasattr(self, "last_bandwidth"):
            self.last_bandwidth = 0                                      # ctor :11-15
            self.last_input_bandwidth = 0
            self.shift_frequency = 0                                     # ctor :30
            self.resampler = None
            self.shifter = self.L.nco_crcf_create(A.LIQUID_NCO)          # ctor :29
            self.last_view = False
            self.new_resampler = False
            self.bw_diff = 0
            self.desired_input_size = 0
        if not hasattr(self, "peak_hold"):
            self.peak_hold, self.peak_reset = False, 0

    def _view_input(self, x, frequency, sample_rate):
        """:283-386: returns the resampler output (num_written samples) of this input"""
        L, N = self.L, self.N
        self.input_freq = frequency
        if not sample_rate:
            return None                                                  # :286-289
        resample_bw = sample_rate
        while resample_bw // 2 >= self.bandwidth:                        # SPECTRUM_VZM, :291-293 (long division)
            resample_bw //= 2
        self.resample_bw = resample_bw
        ratio = float(resample_bw) / float(sample_rate)                  # :295
        desired = int(N / ratio)                                         # :297 size_t <- double
        self.desired_input_size = desired                                # :299
        if x.size < desired:
            desired = x.size                                             # :301-304
        self.new_resampler = False
        self.bw_diff = 0
        if self.center_freq != frequency:                                # :306
            if (self.center_freq - frequency) != self.shift_frequency or self.last_input_bandwidth != sample_rate:   # :307
                if abs(frequency - self.center_freq) < (sample_rate // 2):          # :308 (app rate == input rate here)
                    last_shift = self.shift_frequency
                    self.shift_frequency = self.center_freq - frequency
                    L.nco_crcf_set_frequency(self.shifter, float(np.float32((2.0 * math.pi) * (float(abs(self.shift_frequency)) / float(sample_rate)))))   # :311
                    freq_diff = self.shift_frequency - last_shift        # :314
                    if self.last_bandwidth != 0:                         # :316
                        bin_per_hz = float(self.last_bandwidth) / float(N)
                        num_shift = int(math.floor(float(abs(freq_diff)) / bin_per_hz))
                        if num_shift < N // 2 and num_shift:             # :321
                            if freq_diff > 0:                            # memmove left :323-324
                                self.ma[:N - num_shift] = self.ma[num_shift:].copy()
                                self.maa[:N - num_shift] = self.maa[num_shift:].copy()
                            else:                                        # memmove right :328-329
                                self.ma[num_shift:] = self.ma[:N - num_shift].copy()
                                self.maa[num_shift:] = self.maa[:N - num_shift].copy()
                self.peak_reset = 30                                     # :335 (inside the outer if, whatever the range test said)
            xin = np.ascontiguousarray(x[:desired])
            y = np.empty(desired, np.complex64)
            if self.shift_frequency < 0:                                 # :345-349
                L.nco_crcf_mix_block_up(self.shifter, _cptr(xin), _cptr(y), desired)
            else:
                L.nco_crcf_mix_block_down(self.shifter, _cptr(xin), _cptr(y), desired)
            shifted = y
        else:
            shifted = np.ascontiguousarray(x[:desired])                  # :351
        if self.resampler is None or resample_bw != self.last_bandwidth or self.last_input_bandwidth != sample_rate:   # :354
            self.resampler = L.msresamp_crcf_create(float(np.float32(ratio)), 60.0)   # :361 (destroying the old one)
            self.bw_diff = resample_bw - self.last_bandwidth
            self.last_bandwidth = resample_bw
            self.last_input_bandwidth = sample_rate
            self.new_resampler = True
            self.peak_reset = 30                                         # :367
        out = np.empty(int(math.ceil(float(desired) * ratio)) + 512, np.complex64)   # :370
        nw = C.c_uint()
        L.msresamp_crcf_execute(self.resampler, _cptr(shifted), desired, _cptr(out), C.byref(nw))   # :379
        return out[:nw.value].copy()

    def _rescale_averagers(self):
        """newResampler && lastView (:454-492): the averaged spectrum is stretched / squeezed about its centre"""
        N = self.N
        i = np.arange(N)
        if self.bw_diff < 0:
            src = N // 4 + i // 2
            self.ma = self.ma[src].copy()
            self.maa = self.maa[src].copy()
        else:
            inside = (i >= N // 4) & (i < N - N // 4)
            src = np.where(inside, (i - N // 4) * 2, 0)
            self.ma = np.where(inside, self.ma[src], 0.0)
            self.maa = np.where(inside, self.maa[src], 0.0)

    def _view_map(self):
        """the bin walk of :532-560 for visualRatio = bandwidth / resampleBw: for every visited bin its index, whether it is
        inside (0, N), and the display point that owns it; plus the bins per point.  The double accumulator is stepped
        exactly as the reference does."""
        N, F = self.N, self.F
        key = (self.bandwidth, self.resample_bw)
        if getattr(self, "_vm_key", None) == key:
            return self._vm
        ratio = float(self.bandwidth) / float(self.resample_bw)          # :532
        start = (float(N) / 2.0) - (float(N) * (ratio / 2.0))            # :533
        accum = 0.0
        i = 0.0
        idx, owner, cnt = [], [], np.zeros(F, np.int64)
        for x in range(F):
            accum += ratio * 2.0                                         # SPECTRUM_VZM :541
            while accum >= 1.0:
                v = start + i
                idx.append(int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5)))   # C round(): half away from zero
                owner.append(x)
                cnt[x] += 1
                accum -= 1.0
                i += 1.0
        idx = np.array(idx, np.int64)
        valid = (idx > 0) & (idx < N)                                    # unsigned idx: negative wraps to huge -> invalid
        self._vm_key, self._vm = key, (idx, valid, np.array(owner, np.int64), cnt)
        return self._vm

    def process_frame(self, frame):
        """frame: the 2*fftSize samples process() would FFT.  Returns (spectrum_points[2F], fft_ceiling, fft_floor)."""
        N, F = self.N, self.F
        Y = self.fft(frame)
        mag = np.sqrt((Y.real * Y.real + Y.imag * Y.imag).astype(np.float32)).astype(np.float32)   # float sqrt :443-448
        res = np.concatenate([mag[N // 2:], mag[:N // 2]]).astype(np.float64)                       # :450-451
        if getattr(self, "is_view", False) and self.new_resampler and self.last_view:
            self._rescale_averagers()
        self.maa += (self.ma - self.maa) * self.rate                     # :494-497
        self.ma += (res - self.ma) * self.rate
        fft_ceil = np.float32(max(0.0, self.maa.max()))                  # float locals :436, :499-504
        fft_floor = np.float32(min(1.0, self.maa.min()))
        self.ceil_ma += (float(fft_ceil) - self.ceil_ma) * 0.05          # :513-516
        self.ceil_maa += (self.ceil_ma - self.ceil_maa) * 0.05
        self.floor_ma += (float(fft_floor) - self.floor_ma) * 0.05       # :518-521
        self.floor_maa += (self.floor_ma - self.floor_maa) * 0.05
        do_peak = getattr(self, "do_peak", False)
        if do_peak:                                                      # :506-510, :523-530
            self.peak = np.maximum(self.peak, self.maa)
            self.ceil_peak = max(self.ceil_peak, self.ceil_maa)
            self.floor_peak = min(self.floor_peak, self.floor_maa)
        pc, pf = (self.ceil_peak, self.floor_peak) if do_peak else (self.ceil_maa, self.floor_maa)   # :539-540
        view_map = None
        if getattr(self, "is_view", False):
            view_map = self._view_map()
            idx, valid, owner, cnt = view_map
            vals = np.where(valid, self.maa[np.clip(idx, 0, N - 1)], self.floor_maa)
            acc = np.bincount(owner, weights=vals, minlength=F)
            acc_n = cnt.astype(np.float64)
        else:
            acc = self.maa[0::2] + self.maa[1::2]                        # visualRatio = 1 -> 2 bins per point :538-560
            acc[0] = self.floor_maa + self.maa[1]                        # idx == 0 is replaced by fft_floor_maa :546-551
            acc_n = 2.0
        den = np.log10((pc + 0.25) - (pf - 0.75))
        y = (np.log10(acc / acc_n + 0.25 - (pf - 0.75)) / den) * self.sf   # :566
        pts = np.empty(2 * F, np.float32)
        pts[0::2] = (np.arange(F, dtype=np.float32) / np.float32(F))     # :562
        pts[1::2] = y.astype(np.float32)
        self.hold = None
        if do_peak:
            if view_map is not None:
                idx, valid, owner, cnt = view_map
                pvals = np.where(valid, self.peak[np.clip(idx, 0, N - 1)], self.floor_maa)
                pacc = np.bincount(owner, weights=pvals, minlength=F)
            else:
                pacc = self.peak[0::2] + self.peak[1::2]
                pacc[0] = self.floor_maa + self.peak[1]
            hold = np.empty(2 * F, np.float32)
            hold[0::2] = pts[0::2]
            hold[1::2] = ((np.log10(pacc / acc_n + 0.25 - (pf - 0.75)) / den) * self.sf).astype(np.float32)   # :569
            self.hold = hold
        if getattr(self, "hide_dc", False):
            self._hide_dc(pts)
            if self.hold is not None:
                self._hide_dc(self.hold)
        return pts, pc / self.sf, pf                                     # :626-627
