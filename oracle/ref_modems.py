"""TEST INFRASTRUCTURE ONLY (oracle/).  ctypes front of oracle/_ref/libref_modems.so: the reference's OWN modem classes
(/root/reference/src/modules/modem/*.cpp, compiled in place by oracle/Makefile) running on the reference's own liquid binary.
The strongest anchor this repository has for Modem::demodulate: oracle/cubicsdr_chain.py's RefDemod.demodulate -- the checker
the GPU parity tests use -- is pinned against it, modem by modem, in tests/test_oracle_pin.py."""
import ctypes as C
import os

import numpy as np

from . import liquid_api as A

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib_path():
    return os.path.join(_HERE, "_ref", "libref_modems.so")


def available():
    return A.available("ref") and os.path.exists(lib_path())


def load():
    global _LIB
    if _LIB is None:
        A.load("ref")                                    # the liquid binary is mapped before the first create(); the modem library names
        L = C.CDLL(lib_path())                           # libliquid_ref.so as a dependency (same file: one instance), nothing goes global
        L.refmodem_create.restype = C.c_void_p; L.refmodem_create.argtypes = [C.c_char_p]
        L.refmodem_default_rate.restype = C.c_int; L.refmodem_default_rate.argtypes = [C.c_char_p]
        L.refmodem_check_rate.restype = C.c_longlong; L.refmodem_check_rate.argtypes = [C.c_void_p, C.c_longlong, C.c_int]
        L.refmodem_use_signal_output.restype = C.c_int; L.refmodem_use_signal_output.argtypes = [C.c_void_p]
        L.refmodem_write_setting.restype = None; L.refmodem_write_setting.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.refmodem_build.restype = C.c_int; L.refmodem_build.argtypes = [C.c_void_p, C.c_longlong, C.c_int]
        L.refmodem_demodulate.restype = C.c_int
        L.refmodem_demodulate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.refmodem_demod_output.restype = C.c_int; L.refmodem_demod_output.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.refmodem_destroy.restype = None; L.refmodem_destroy.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


class RefModem:
    """Modem::makeModem(name) + buildKit(checkSampleRate(bandwidth), audio_rate) + demodulate() block by block"""

    def __init__(self, name, bandwidth, audio_rate=48000, settings=None):
        self.L = load()
        self.h = self.L.refmodem_create(name.encode())
        if not self.h:
            raise ValueError("the reference has no modem called %r" % name)
        self.name = name
        self.audio_rate = int(audio_rate)
        self.rate = int(self.L.refmodem_check_rate(self.h, int(bandwidth), self.audio_rate))
        for k, v in (settings or {}).items():
            self.L.refmodem_write_setting(self.h, k.encode(), str(v).encode())
        if self.L.refmodem_build(self.h, self.rate, self.audio_rate):
            raise RuntimeError("buildKit failed")
        self.use_signal_output = bool(self.L.refmodem_use_signal_output(self.h))

    def demodulate(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        cap = 4 * iq.size + 4096
        out = np.empty(cap, np.float32)
        ch = C.c_int(0)
        m = self.L.refmodem_demodulate(self.h, iq.ctypes.data_as(C.c_void_p), iq.size, self.rate, out.ctypes.data_as(C.c_void_p), cap, C.byref(ch))
        if m < 0:
            raise RuntimeError("audio buffer too small")
        return out[:m].copy(), ch.value

    def demod_output(self, cap=1 << 20):
        out = np.empty(cap, np.float32)
        m = self.L.refmodem_demod_output(self.h, out.ctypes.data_as(C.c_void_p), cap)
        return None if m < 0 else out[:m].copy()

    def close(self):
        if self.h:
            self.L.refmodem_destroy(self.h)
            self.h = None


# ---- the reference's own SpectrumVisualProcessor (oracle/_ref/libref_spectrum.so, oracle/ref/spectrum_harness.cpp) -------------
_SLIB = None


def spectrum_lib_path():
    return os.path.join(_HERE, "_ref", "libref_spectrum.so")


def spectrum_available():
    return A.available("ref") and os.path.exists(spectrum_lib_path())


def load_spectrum():
    global _SLIB
    if _SLIB is None:
        A.load("ref")
        L = C.CDLL(spectrum_lib_path())
        L.refspec_create.restype = C.c_void_p; L.refspec_create.argtypes = [C.c_uint, C.c_longlong]
        for name, args in (("refspec_set_average_rate", [C.c_void_p, C.c_float]), ("refspec_set_scale", [C.c_void_p, C.c_float]),
                           ("refspec_set_peak_hold", [C.c_void_p, C.c_int]), ("refspec_set_hide_dc", [C.c_void_p, C.c_int]),
                           ("refspec_set_center", [C.c_void_p, C.c_longlong]), ("refspec_set_bandwidth", [C.c_void_p, C.c_long]),
                           ("refspec_set_view", [C.c_void_p, C.c_int, C.c_longlong, C.c_long]), ("refspec_destroy", [C.c_void_p])):
            getattr(L, name).restype = None; getattr(L, name).argtypes = args
        L.refspec_desired_input_size.restype = C.c_int; L.refspec_desired_input_size.argtypes = [C.c_void_p]
        L.refspec_process.restype = C.c_int
        L.refspec_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        _SLIB = L
    return _SLIB


class RefSpectrumCpp:
    """SpectrumVisualProcessor of the reference: setup(fft_size), the setters, one process() per call"""

    def __init__(self, fft_size, app_sample_rate=2400000):
        self.L = load_spectrum()
        self.F = int(fft_size)
        self.h = self.L.refspec_create(self.F, int(app_sample_rate))

    def set_average_rate(self, r): self.L.refspec_set_average_rate(self.h, float(r))
    def set_scale(self, sf): self.L.refspec_set_scale(self.h, float(sf))
    def set_peak_hold(self, on): self.L.refspec_set_peak_hold(self.h, 1 if on else 0)
    def set_hide_dc(self, on): self.L.refspec_set_hide_dc(self.h, 1 if on else 0)
    def set_center(self, f): self.L.refspec_set_center(self.h, int(f))
    def set_bandwidth(self, bw): self.L.refspec_set_bandwidth(self.h, int(bw))
    def set_view(self, on, center=0, bw=0): self.L.refspec_set_view(self.h, 1 if on else 0, int(center), int(bw))
    def desired_input_size(self): return self.L.refspec_desired_input_size(self.h)

    def process(self, iq, frequency, sample_rate):
        """-> None (no output this call) or (points[2F], fft_ceiling, fft_floor, hold_points or None)"""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        cap = 2 * self.F + 16
        pts = np.empty(cap, np.float32); hold = np.empty(cap, np.float32)
        cf = np.zeros(2, np.float64); nh = C.c_int(0)
        m = self.L.refspec_process(self.h, iq.ctypes.data_as(C.c_void_p), iq.size, int(frequency), int(sample_rate), pts.ctypes.data_as(C.c_void_p),
                                   hold.ctypes.data_as(C.c_void_p), cap, cf.ctypes.data_as(C.c_void_p), C.byref(nh))
        if m <= 0:
            return None
        return pts[:m].copy(), float(cf[0]), float(cf[1]), (hold[:nh.value].copy() if nh.value else None)

    def close(self):
        if self.h:
            self.L.refspec_destroy(self.h)
            self.h = None


# ---- the reference's own ScopeVisualProcessor (oracle/_ref/libref_scope.so, oracle/ref/scope_harness.cpp) -----------------------
_SCLIB = None


def scope_available():
    return A.available("ref") and os.path.exists(os.path.join(_HERE, "_ref", "libref_scope.so"))


def load_scope():
    global _SCLIB
    if _SCLIB is None:
        A.load("ref")
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref_scope.so"))
        L.refscope_create.restype = C.c_void_p; L.refscope_create.argtypes = [C.c_int]
        L.refscope_enable.restype = None; L.refscope_enable.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.refscope_push.restype = C.c_int; L.refscope_push.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.refscope_get.restype = C.c_int; L.refscope_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.refscope_destroy.restype = None; L.refscope_destroy.argtypes = [C.c_void_p]
        _SCLIB = L
    return _SCLIB


class RefScopeCpp:
    """ScopeVisualProcessor of the reference: one AudioThreadInput per push -> the ScopeRenderData items it distributed"""

    def __init__(self, fft_size=1024):
        self.L = load_scope()
        self.h = self.L.refscope_create(int(fft_size))

    def enable(self, scope=True, spectrum=True):
        self.L.refscope_enable(self.h, int(scope), int(spectrum))

    def push(self, data, channels, input_rate, sample_rate, typ):
        """-> list of dicts {points, mode, spectrum, channels, input_rate, sample_rate, fft_size, fft_floor, fft_ceil}"""
        a = np.ascontiguousarray(data, dtype=np.float32)
        n = self.L.refscope_push(self.h, a.ctypes.data_as(C.c_void_p), a.size, int(channels), int(input_rate), int(sample_rate), int(typ))
        out = []
        for i in range(n):
            pts = np.empty(1 << 16, np.float32); meta = np.zeros(6, np.int32); fc = np.zeros(2, np.float64)
            m = self.L.refscope_get(self.h, i, pts.ctypes.data_as(C.c_void_p), pts.size, meta.ctypes.data_as(C.c_void_p), fc.ctypes.data_as(C.c_void_p))
            assert m >= 0
            out.append(dict(points=pts[:m].copy(), mode=int(meta[0]), spectrum=bool(meta[1]), channels=int(meta[2]), input_rate=int(meta[3]),
                            sample_rate=int(meta[4]), fft_size=int(meta[5]), fft_floor=float(fc[0]), fft_ceil=float(fc[1])))
        return out

    def close(self):
        if self.h:
            self.L.refscope_destroy(self.h)
            self.h = None


# ---- the reference's own audioCallback and AudioFileWAV (oracle/_ref/libref_audio.so, oracle/ref/audio_harness.cpp) --------------
_AULIB = None


def audio_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_audio.so"))


def load_audio():
    global _AULIB
    if _AULIB is None:
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref_audio.so"))
        L.refaudio_create.restype = C.c_void_p; L.refaudio_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.refaudio_set_source.restype = None; L.refaudio_set_source.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float]
        L.refaudio_push.restype = C.c_int; L.refaudio_push.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]
        L.refaudio_queued.restype = C.c_int; L.refaudio_queued.argtypes = [C.c_void_p, C.c_int]
        L.refaudio_callback.restype = C.c_int; L.refaudio_callback.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.refaudio_destroy.restype = None; L.refaudio_destroy.argtypes = [C.c_void_p]
        L.refwav_create.restype = C.c_void_p; L.refwav_create.argtypes = [C.c_char_p, C.c_char_p]
        L.refwav_write.restype = C.c_int; L.refwav_write.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float]
        L.refwav_close.restype = None; L.refwav_close.argtypes = [C.c_void_p]
        L.refwav_destroy.restype = None; L.refwav_destroy.argtypes = [C.c_void_p]
        _AULIB = L
    return _AULIB


class RefAudioMixCpp:
    """a controller AudioThread with n bound source threads, driven through the reference's audioCallback"""

    def __init__(self, sample_rate, n_sources, queue_blocks=0):
        self.L = load_audio()
        self.h = self.L.refaudio_create(int(sample_rate), int(n_sources), int(queue_blocks))

    def set_source(self, i, active=True, gain=1.0):
        self.L.refaudio_set_source(self.h, int(i), int(active), float(gain))

    def push(self, i, data, channels, sample_rate, peak):
        a = np.ascontiguousarray(data, dtype=np.float32)
        return bool(self.L.refaudio_push(self.h, int(i), a.ctypes.data_as(C.c_void_p), a.size, int(channels), int(sample_rate), float(peak)))

    def queued(self, i):
        return self.L.refaudio_queued(self.h, int(i))

    def callback(self, frames):
        out = np.empty(2 * frames, np.float32)
        self.L.refaudio_callback(self.h, out.ctypes.data_as(C.c_void_p), int(frames))
        return out

    def close(self):
        if self.h:
            self.L.refaudio_destroy(self.h)
            self.h = None


class RefWavCpp:
    """AudioFileWAV of the reference writing under `directory` (its getOutputFileName appends -N when the name exists)"""

    def __init__(self, directory, base):
        self.L = load_audio()
        self.h = self.L.refwav_create(directory.encode(), base.encode())

    def write(self, data, channels, sample_rate, peak):
        a = np.ascontiguousarray(data, dtype=np.float32)
        return bool(self.L.refwav_write(self.h, a.ctypes.data_as(C.c_void_p), a.size, int(channels), int(sample_rate), float(peak)))

    def close(self):
        if self.h:
            self.L.refwav_close(self.h)
            self.L.refwav_destroy(self.h)
            self.h = None


# ---- the reference's own DemodulatorThread::run (oracle/_ref/libref_demodthread.so, oracle/ref/demod_thread_harness.cpp) ------------
_DTLIB = None


def demodthread_available():
    return A.available("ref") and os.path.exists(os.path.join(_HERE, "_ref", "libref_demodthread.so"))


def load_demodthread():
    global _DTLIB
    if _DTLIB is None:
        A.load("ref")
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref_demodthread.so"))
        L.refdt_create.restype = C.c_void_p; L.refdt_create.argtypes = [C.c_int, C.c_longlong, C.c_int]
        L.refdt_set.restype = None; L.refdt_set.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int]
        L.refdt_block.restype = C.c_int
        L.refdt_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.refdt_destroy.restype = None; L.refdt_destroy.argtypes = [C.c_void_p]
        _DTLIB = L
    return _DTLIB


class RefDemodThreadCpp:
    """DemodulatorThread of the reference on its own thread; block(iq, audio) runs one input through run() and reports what it left behind"""

    def __init__(self, use_signal_output, modem_rate=12500, audio_rate=48000):
        self.L = load_demodthread()
        self.rate = int(modem_rate)
        self.h = self.L.refdt_create(int(use_signal_output), self.rate, int(audio_rate))

    def set(self, squelch_enabled=False, squelch_level=-100.0, muted=False):
        self.L.refdt_set(self.h, int(squelch_enabled), float(squelch_level), int(muted))

    def block(self, iq, audio, channels=1):
        iq = np.ascontiguousarray(iq, dtype=np.complex64); audio = np.ascontiguousarray(audio, dtype=np.float32)
        out = np.zeros(11, np.float64); tap = np.zeros(8192, np.float32)
        ok = self.L.refdt_block(self.h, iq.ctypes.data_as(C.c_void_p), iq.size, self.rate, audio.ctypes.data_as(C.c_void_p), audio.size, int(channels),
                                out.ctypes.data_as(C.c_void_p), tap.ctypes.data_as(C.c_void_p), tap.size)
        if not ok:
            raise RuntimeError("the reference demodulator thread did not answer")
        return dict(level=np.float32(out[0]), floor=np.float32(out[1]), ceil=np.float32(out[2]), squelch_break=bool(out[3]), pushed=bool(out[4]), peak=np.float32(out[5]),
                    tap=(tap[:int(out[7])].copy() if out[6] else None), tap_input_rate=int(out[8]), tap_sample_rate=int(out[9]), tap_type=int(out[10]))

    def close(self):
        if self.h:
            self.L.refdt_destroy(self.h)
            self.h = None


# ---- the reference's own SDRPostThread::run (oracle/_ref/libref_post.so, oracle/ref/post_harness.cpp) ------------------------------
_PLIB = None


def post_available():
    return A.available("ref") and os.path.exists(os.path.join(_HERE, "_ref", "libref_post.so"))


def load_post():
    global _PLIB
    if _PLIB is None:
        A.load("ref")
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref_post.so"))
        L.refpost_create.restype = C.c_void_p; L.refpost_create.argtypes = [C.c_int]
        L.refpost_add_demod.restype = C.c_int; L.refpost_add_demod.argtypes = [C.c_void_p, C.c_longlong, C.c_int]
        L.refpost_set_demod_frequency.restype = None; L.refpost_set_demod_frequency.argtypes = [C.c_void_p, C.c_int, C.c_longlong]
        L.refpost_notify.restype = None; L.refpost_notify.argtypes = [C.c_void_p]
        L.refpost_set_app.restype = None; L.refpost_set_app.argtypes = [C.c_longlong, C.c_longlong]
        L.refpost_block.restype = C.c_int; L.refpost_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_longlong, C.c_int, C.c_void_p, C.c_int]
        L.refpost_fetch.restype = C.c_int; L.refpost_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.refpost_fetch_visual.restype = C.c_int; L.refpost_fetch_visual.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.refpost_destroy.restype = None; L.refpost_destroy.argtypes = [C.c_void_p]
        _PLIB = L
    return _PLIB


class RefPostThreadCpp:
    """SDRPostThread of the reference on its own thread with n demodulators attached"""

    def __init__(self, center, rate, oversampled=False):
        self.L = load_post()
        self.L.refpost_set_app(int(center), int(rate))
        self.h = self.L.refpost_create(int(oversampled))
        self.n = 0

    def add_demod(self, frequency, current=False):
        self.n = self.L.refpost_add_demod(self.h, int(frequency), int(current)) + 1
        return self.n - 1

    def set_demod_frequency(self, i, f):
        self.L.refpost_set_demod_frequency(self.h, int(i), int(f))

    def notify(self):
        self.L.refpost_notify(self.h)

    def block(self, x, frequency, rate, num_channels):
        """-> list of isActive() per demodulator after the block"""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        act = np.zeros(max(self.n, 1), np.int32)
        if not self.L.refpost_block(self.h, x.ctypes.data_as(C.c_void_p), x.size, int(frequency), int(rate), int(num_channels), act.ctypes.data_as(C.c_void_p), self.n):
            raise RuntimeError("the reference post thread did not finish the block")
        return [bool(a) for a in act[:self.n]]

    def _fetch(self, fn, *args):
        buf = np.empty(1 << 22, np.complex64); meta = np.zeros(2, np.int64)
        n = fn(self.h, *args, buf.ctypes.data_as(C.c_void_p), buf.size, meta.ctypes.data_as(C.c_void_p))
        assert n >= 0
        return None if n == 0 else (buf[:n].copy(), int(meta[0]), int(meta[1]))

    def fetch(self, i):
        """what demodulator i's input pipe received for the last block: (samples, frequency, sampleRate) or None"""
        return self._fetch(self.L.refpost_fetch, int(i))

    def fetch_visual(self, which):
        return self._fetch(self.L.refpost_fetch_visual, int(which))

    def close(self):
        if self.h:
            self.L.refpost_destroy(self.h)
            self.h = None
