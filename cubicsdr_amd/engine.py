"""Thin Python host objects over the C ABI (tests, bench, smoke).  Names follow the reference objects whose
arithmetic each handle replaces: SDRPostThread (src/sdr/SDRPostThread.cpp), DemodulatorInstance's Pre/Demod
threads + Modem (src/demod/, src/modules/modem/), SpectrumVisualProcessor (src/process/).

Inputs may be numpy complex64 arrays (host, staged by the library) or torch CUDA tensors (HBM resident, passed by
device pointer).  No computation happens in Python; without the HIP library or a GPU everything raises.
"""
import ctypes as C

import numpy as np

from . import hip as H


def _as_iq_arg(iq):
    """-> (pointer, is_dev, n_complex, keepalive)"""
    if isinstance(iq, np.ndarray):
        if iq.dtype == np.float32 and iq.ndim == 2 and iq.shape[1] == 2:          # interleaved (re, im) pairs
            iq = np.ascontiguousarray(iq).view(np.complex64).reshape(-1)
        a = np.ascontiguousarray(iq, dtype=np.complex64)
        return a.ctypes.data_as(C.c_void_p), 0, a.size, a
    # torch tensor on the GPU: complex64 [n] or float32 [n, 2] / [2n]
    import torch
    if not isinstance(iq, torch.Tensor) or not iq.is_cuda:
        raise TypeError("iq must be a numpy array or a CUDA torch tensor")
    t = iq.contiguous()
    if t.dtype == torch.complex64:
        n = t.numel()
    elif t.dtype == torch.float32:
        n = t.numel() // 2
    else:
        raise TypeError("iq tensor must be complex64 or float32")
    return C.c_void_p(t.data_ptr()), 1, n, t


class Context:
    """device + streams (csdr_ctx): one internal HIP stream per pipeline stage; `stream` is the boundary stream the
    caller's own GPU work is ordered on: None creates a private one; a raw hipStream_t handle (e.g. torch's
    `torch.cuda.Stream.cuda_stream`) chains with the caller's work -- 0 is the device's null stream (torch's default
    stream), passed on as CSDR_STREAM_NULL."""

    def __init__(self, device=0, stream=None):
        self._l = H.lib()
        self.h = C.c_void_p()
        if stream is None:
            arg = None
        elif int(stream) == 0:
            arg = C.c_void_p(-1)                                  # CSDR_STREAM_NULL
        else:
            arg = C.c_void_p(int(stream))
        H.check(self._l.csdr_ctx_create(device, arg, C.byref(self.h)))

    @property
    def owns_stream(self):
        """True when the boundary stream is private to the library (nothing the caller enqueues is ordered against it)"""
        return bool(self._l.csdr_ctx_owns_stream(self.h))

    def synchronize(self):
        H.check(self._l.csdr_ctx_synchronize(self.h))

    def join(self):
        """the boundary stream waits for everything enqueued on the internal stage streams so far"""
        H.check(self._l.csdr_ctx_join(self.h))

    def timer_start(self):
        H.check(self._l.csdr_ctx_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        H.check(self._l.csdr_ctx_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        """True / 1: time every launch; an integer P > 1: every P-th launch of each kernel; False: off"""
        H.check(self._l.csdr_ctx_profile_enable(self.h, int(on)))

    def profile(self):
        """-> {kernel name: (total_ms of the bracketed launches, bracketed launches, ALL launches)} since profile_enable(...)"""
        out = {}
        for i in range(self._l.csdr_ctx_profile_num_kernels()):
            ms, n, seen = C.c_double(), C.c_int64(), C.c_int64()
            H.check(self._l.csdr_ctx_profile_fetch(self.h, i, C.byref(ms), C.byref(n)))
            H.check(self._l.csdr_ctx_profile_launches(self.h, i, C.byref(seen)))
            if n.value:
                out[self._l.csdr_ctx_profile_kernel_name(i).decode()] = (ms.value, n.value, seen.value)
        return out

    def close(self):
        if self.h:
            self._l.csdr_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SDRPost:
    """SDRPostThread's arithmetic (csdr_post): DC blocker (1 channel) or firpfbch analyzer (M channels)."""

    def __init__(self, ctx, sample_rate, num_channels, max_block_len, max_blocks=1, oversampled=False):
        """oversampled=True: SDRPostPFBCH2 (firpfbch2, channels at twice the channel spacing)"""
        self._l = H.lib()
        self.ctx = ctx
        self.h = C.c_void_p()
        H.check(self._l.csdr_post_create(ctx.h, C.byref(self.h)))
        mode = H.CSDR_POST_SINGLE if num_channels == 1 else (H.CSDR_POST_PFBCH2 if oversampled else H.CSDR_POST_PFBCH)
        self.hop = 1 if num_channels == 1 else (num_channels // 2 if oversampled else num_channels)
        H.check(self._l.csdr_post_configure(self.h, int(sample_rate), int(num_channels), mode, int(max_block_len), int(max_blocks)))
        self.num_channels = num_channels
        self.sample_rate = sample_rate

    def set_active_channels(self, channels=None):
        if channels is None:
            H.check(self._l.csdr_post_set_active_channels(self.h, None, 0))
        else:
            a = np.ascontiguousarray(channels, dtype=np.int32)
            H.check(self._l.csdr_post_set_active_channels(self.h, a.ctypes.data_as(C.c_void_p), a.size))

    def execute(self, iq, n_blocks, block_len, frequency):
        p, is_dev, n, keep = _as_iq_arg(iq)
        if n < n_blocks * block_len:
            raise ValueError("iq holds %d samples, need %d" % (n, n_blocks * block_len))
        H.check(self._l.csdr_post_execute(self.h, p, is_dev, int(n_blocks), int(block_len), int(frequency)))
        self._keep = keep
        self._last = (n_blocks, block_len)

    # ---- time-slab sharding (csdr_hip.h: producer / owner halves; parallel.SlabStream drives them).  `buf` arguments are DEVICE
    # buffers: torch tensors on the context's GPU (float32 [.., 2] or complex64); numpy arrays only when the library in use runs its kernels on the host (the test suite has such a build)
    @staticmethod
    def _dev_ptr(buf):
        if isinstance(buf, np.ndarray):
            return buf.ctypes.data_as(C.c_void_p)
        return C.c_void_p(buf.data_ptr())

    @property
    def history_length(self):
        return self._l.csdr_post_history_length(self.h)

    def set_history(self, tail, n_samples):
        H.check(self._l.csdr_post_set_history(self.h, self._dev_ptr(tail), int(n_samples)))
        self._keep_hist = tail

    def set_dc_blocker(self, enabled):
        H.check(self._l.csdr_post_set_dc_blocker(self.h, 1 if enabled else 0))

    def export_rows(self, channels, dst, dst_stride):
        a = np.ascontiguousarray(channels, dtype=np.int32)
        H.check(self._l.csdr_post_export_rows(self.h, a.ctypes.data_as(C.c_void_p), a.size, self._dev_ptr(dst), int(dst_stride)))

    def import_begin(self, n_blocks, block_len, frequency):
        H.check(self._l.csdr_post_import_begin(self.h, int(n_blocks), int(block_len), int(frequency)))
        self._last = (n_blocks, block_len)

    def import_rows(self, channels, src, src_stride, frame0, n_frames):
        a = np.ascontiguousarray(channels, dtype=np.int32)
        H.check(self._l.csdr_post_import_rows(self.h, a.ctypes.data_as(C.c_void_p), a.size, self._dev_ptr(src), int(src_stride), int(frame0), int(n_frames)))

    def import_commit(self):
        H.check(self._l.csdr_post_import_commit(self.h))

    @property
    def channel_bandwidth(self):
        return self._l.csdr_post_channel_bandwidth(self.h)

    @property
    def channel_rate(self):
        return self._l.csdr_post_channel_rate(self.h)

    def channel_center(self, i):
        return self._l.csdr_post_channel_center(self.h, i)

    def channel_at(self, f):
        return self._l.csdr_post_channel_at(self.h, int(f))

    def read_channel(self, ch):
        nb, bl = self._last
        cap = nb * (bl // self.hop)
        out = np.empty(cap, np.complex64)
        n = C.c_int()
        H.check(self._l.csdr_post_read_channel(self.h, int(ch), out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return out[:n.value]

    def close(self):
        if self.h:
            self._l.csdr_post_destroy(self.h)
            self.h = C.c_void_p()


class DemodBank:
    """N demodulator slots (csdr_bank); one slot = one DemodulatorInstance's Pre + Demod thread arithmetic."""

    def __init__(self, ctx, max_demods, max_blocks=1):
        self._l = H.lib()
        self.ctx = ctx
        self.h = C.c_void_p()
        self.max_blocks = max_blocks
        H.check(self._l.csdr_bank_create(ctx.h, int(max_demods), int(max_blocks), C.byref(self.h)))

    def configure(self, slot, post, modem, bandwidth, frequency, audio_sample_rate=48000, modem_arg=0):
        """modem_arg: FM stereo de-emphasis in microseconds (0: the reference's default 75, negative: none)"""
        m = H.MODEM_BY_NAME[modem] if isinstance(modem, str) else int(modem)
        p = H.DemodParams(m, int(bandwidth), int(audio_sample_rate), int(modem_arg), int(frequency))
        H.check(self._l.csdr_bank_configure_slot(self.h, int(slot), C.byref(p), post.h))

    def set_frequency(self, slot, f):
        H.check(self._l.csdr_bank_set_frequency(self.h, int(slot), int(f)))

    def set_active(self, slot, active):
        H.check(self._l.csdr_bank_set_active(self.h, int(slot), int(bool(active))))

    def execute(self, post):
        H.check(self._l.csdr_bank_execute(self.h, post.h))

    def results(self, slot):
        arr = (H.BlockResult * self.max_blocks)()
        n = C.c_int()
        H.check(self._l.csdr_bank_fetch_results(self.h, int(slot), arr, self.max_blocks, C.byref(n)))
        return [arr[i] for i in range(n.value)]

    def audio(self, slot, cap=1 << 22):
        out = np.empty(cap, np.float32)
        n = C.c_int()
        H.check(self._l.csdr_bank_fetch_audio(self.h, int(slot), out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return out[:n.value].copy()

    def iq(self, slot, cap=1 << 22):
        out = np.empty(cap, np.complex64)
        n = C.c_int()
        H.check(self._l.csdr_bank_fetch_iq(self.h, int(slot), out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return out[:n.value].copy()

    def demod_output(self, slot, cap=2048):
        """scaled demodulator output of the last block (ModemAnalog::getDemodOutputData), at most 2048 samples"""
        out = np.empty(cap, np.float32)
        n = C.c_int()
        H.check(self._l.csdr_bank_fetch_demod_output(self.h, int(slot), out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return out[:n.value].copy()

    def set_fms_pilot(self, slot, b15=None, a15=None):
        """replace (or, with None, restore) the pilot band-pass sections of an FM-stereo slot"""
        if b15 is None:
            H.check(self._l.csdr_bank_set_fms_pilot(self.h, int(slot), None, None))
            return
        b = np.ascontiguousarray(b15, dtype=np.float32); a = np.ascontiguousarray(a15, dtype=np.float32)
        assert b.size == 15 and a.size == 15
        H.check(self._l.csdr_bank_set_fms_pilot(self.h, int(slot), b.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p)))

    def fms_stage(self, slot, which, cap=1 << 22):
        """FM stereo intermediates of the last batch: which = 0 pilot oscillator phase words (uint32), 1 stereo-difference stream (float32)"""
        out = np.empty(cap, dtype=np.uint32 if which == 0 else np.float32)
        n = C.c_int(0)
        H.check(self._l.csdr_bank_fetch_fms_stage(self.h, int(slot), int(which), out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return out[:n.value].copy()

    def total_audio(self):
        n = C.c_int64()
        H.check(self._l.csdr_bank_total_audio(self.h, C.byref(n)))
        return n.value

    def close(self):
        if self.h:
            self._l.csdr_bank_destroy(self.h)
            self.h = C.c_void_p()


class SpectrumProcessor:
    """SpectrumVisualProcessor's arithmetic (csdr_spec), full-span view."""

    def __init__(self, ctx, fft_size, max_frames=1):
        self._l = H.lib()
        self.ctx = ctx
        self.h = C.c_void_p()
        self.fft_size = fft_size
        H.check(self._l.csdr_spec_create(ctx.h, C.byref(self.h)))
        H.check(self._l.csdr_spec_setup(self.h, int(fft_size), int(max_frames)))

    def set_average_rate(self, r):
        H.check(self._l.csdr_spec_set_average_rate(self.h, float(r)))

    def set_peak_hold(self, enabled):
        H.check(self._l.csdr_spec_set_peak_hold(self.h, int(bool(enabled))))

    def set_hide_dc(self, enabled, center_freq=None, bandwidth=None, input_freq=None):
        H.check(self._l.csdr_spec_set_hide_dc(self.h, int(bool(enabled))))
        if center_freq is not None:
            H.check(self._l.csdr_spec_set_center_frequency(self.h, int(center_freq)))
        if bandwidth is not None:
            H.check(self._l.csdr_spec_set_bandwidth(self.h, int(bandwidth)))
        if input_freq is not None:
            H.check(self._l.csdr_spec_set_input_frequency(self.h, int(input_freq)))

    def set_view(self, on, center_freq=None, bandwidth=None):
        H.check(self._l.csdr_spec_set_view(self.h, int(bool(on))))
        if center_freq is not None:
            H.check(self._l.csdr_spec_set_center_frequency(self.h, int(center_freq)))
        if bandwidth is not None:
            H.check(self._l.csdr_spec_set_bandwidth(self.h, int(bandwidth)))

    def process_view_input(self, iq, frequency, sample_rate):
        """one process() input in zoomed-view mode; returns the number of frames it produced (0 or 1)"""
        H.check(self._l.csdr_spec_set_input_frequency(self.h, int(frequency)))
        H.check(self._l.csdr_spec_set_input_rate(self.h, int(sample_rate)))
        p, is_dev, n, keep = _as_iq_arg(iq)
        H.check(self._l.csdr_spec_process(self.h, p, is_dev, 1, int(n), H.CSDR_SPEC_FIRST_FRAME))
        self._keep = keep
        return self._l.csdr_spec_frames(self.h)

    @property
    def desired_input_size(self):
        return self._l.csdr_spec_desired_input_size(self.h)

    def fetch_hold(self, frame):
        """spectrum_hold_points of a frame, or None when it carries none"""
        pts = np.empty(2 * self.fft_size, np.float32)
        n = C.c_int()
        H.check(self._l.csdr_spec_fetch_hold(self.h, int(frame), pts.ctypes.data_as(C.c_void_p), pts.size, C.byref(n)))
        return pts if n.value else None

    def process(self, iq, n_blocks, block_len, contiguous=False, lines=False):
        """frames per `mode`: first 2*fftSize samples of every block (default), every non-overlapping frame
        (contiguous=True), or one overlapped frame per short block (lines=True: FFTDataDistributor's fftSize-sample lines)"""
        p, is_dev, n, keep = _as_iq_arg(iq)
        if n < n_blocks * block_len:
            raise ValueError("iq holds %d samples, need %d" % (n, n_blocks * block_len))
        mode = H.CSDR_SPEC_LINES if lines else (H.CSDR_SPEC_CONTIGUOUS if contiguous else H.CSDR_SPEC_FIRST_FRAME)
        H.check(self._l.csdr_spec_process(self.h, p, is_dev, int(n_blocks), int(block_len), mode))
        self._keep = keep
        return self._l.csdr_spec_frames(self.h)

    def fetch(self, frame):
        pts = np.empty(2 * self.fft_size, np.float32)
        ce, fl = C.c_double(), C.c_double()
        H.check(self._l.csdr_spec_fetch(self.h, int(frame), pts.ctypes.data_as(C.c_void_p), pts.size, C.byref(ce), C.byref(fl)))
        return pts, ce.value, fl.value

    def fft_only(self, x):
        a = np.ascontiguousarray(x, dtype=np.complex64)
        assert a.size == 2 * self.fft_size
        out = np.empty(a.size, np.complex64)
        H.check(self._l.csdr_spec_fft_only(self.h, a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if self.h:
            self._l.csdr_spec_destroy(self.h)
            self.h = C.c_void_p()
