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
    if isinstance(iq, DevicePointer):
        return C.c_void_p(iq.ptr), 1, iq.n, iq
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

    def profile_range(self):
        """-> {kernel name: (shortest, longest) bracketed launch in ms} since profile_enable(...)"""
        out = {}
        for i in range(self._l.csdr_ctx_profile_num_kernels()):
            lo, hi = C.c_double(), C.c_double()
            H.check(self._l.csdr_ctx_profile_range(self.h, i, C.byref(lo), C.byref(hi)))
            if hi.value > 0.0:
                out[self._l.csdr_ctx_profile_kernel_name(i).decode()] = (lo.value, hi.value)
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

    def set_row_order(self, channels=None):
        """time-slab producers: rows stored in this channel order (csdr_post_set_row_order); None = row is the channel"""
        a = np.ascontiguousarray(channels if channels is not None else [], dtype=np.int32)
        H.check(self._l.csdr_post_set_row_order(self.h, a.ctypes.data_as(C.c_void_p) if a.size else None, a.size))

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
    def kernel_name(self):
        """which kernel the channel count maps to (csdr_post_kernel_name)"""
        return self._l.csdr_post_kernel_name(self.h).decode()

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

    def pcm16(self, slot, cap=1 << 22):
        """the slot's audio of the last execute as 16-bit PCM, every block scaled by its own peak (AudioFileWAV.cpp:133-157), converted on the device"""
        out = np.empty(cap, np.int16)
        n = C.c_int()
        H.check(self._l.csdr_bank_fetch_pcm16(self.h, int(slot), out.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        return out[:n.value].copy()

    def scope_frame(self, slot):
        """the audio-scope tap of the last block as a device-resident frame (H.ScopeFrame; n == 0: nothing to show)"""
        f = H.ScopeFrame()
        H.check(self._l.csdr_bank_scope_frame(self.h, int(slot), C.byref(f)))
        return f

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

    def set_scale_factor(self, f):
        H.check(self._l.csdr_spec_set_scale_factor(self.h, float(f)))

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


class ScopeProcessor:
    """ScopeVisualProcessor's arithmetic (csdr_scope): waveform normalisation + audio spectrum of AudioThreadInput frames."""

    def __init__(self, ctx, fft_size=1024, max_frames=8, max_samples=8192):
        self._l = H.lib()
        self.ctx = ctx
        self.h = C.c_void_p()
        H.check(self._l.csdr_scope_create(ctx.h, C.byref(self.h)))
        H.check(self._l.csdr_scope_setup(self.h, int(fft_size), int(max_frames), int(max_samples)))

    def set_enabled(self, scope=True, spectrum=True):
        H.check(self._l.csdr_scope_set_enabled(self.h, int(scope), int(spectrum)))

    def process(self, frames):
        """frames: list of dicts {data (numpy float32), channels, type, sample_rate, input_rate} (host frames), or of H.ScopeFrame
        structures whose data lies in HBM (DemodBank.scope_frame)"""
        n = len(frames)
        arr = (H.ScopeFrame * n)()
        keep = []
        dev = isinstance(frames[0], H.ScopeFrame)
        for i, f in enumerate(frames):
            if dev:
                arr[i] = f
            else:
                a = np.ascontiguousarray(f["data"], dtype=np.float32)
                keep.append(a)
                arr[i] = H.ScopeFrame(a.ctypes.data, None, a.size, int(f["channels"]), int(f.get("type", 0)), int(f["sample_rate"]), int(f["input_rate"]),
                                      int(f.get("layout", 0)), float(f.get("scale", 1.0)))
        H.check(self._l.csdr_scope_process(self.h, arr, n, 1 if dev else 0))

    def fetch(self, frame, spectrum):
        """-> dict like oracle.ref_modems.RefScopeCpp.push items, or None when that item was not produced"""
        pts = np.empty(1 << 15, np.float32)
        info = H.ScopeInfo()
        H.check(self._l.csdr_scope_fetch(self.h, int(frame), 1 if spectrum else 0, pts.ctypes.data_as(C.c_void_p), pts.size, C.byref(info)))
        if info.n_floats == 0:
            return None
        return dict(points=pts[:info.n_floats].copy(), mode=info.mode, spectrum=bool(info.spectrum), channels=info.channels, input_rate=info.input_rate,
                    sample_rate=info.sample_rate, fft_size=info.fft_size, fft_floor=info.fft_floor, fft_ceil=info.fft_ceil)

    def close(self):
        if self.h:
            self._l.csdr_scope_destroy(self.h)
            self.h = C.c_void_p()


class AudioMixer:
    """AudioThread's mixing callback (csdr_mix): per-source block queues, rings in HBM, bit-exact mix-down."""

    def __init__(self, ctx, n_sources, sample_rate=48000, ring_floats=1 << 20, queue_blocks=0):
        self._l = H.lib()
        self.ctx = ctx
        self.h = C.c_void_p()
        H.check(self._l.csdr_mix_create(ctx.h, int(n_sources), int(ring_floats), int(sample_rate), C.byref(self.h)))
        for i in range(n_sources):
            self.set_source(i, queue_blocks=queue_blocks)

    def set_source(self, i, bound=True, active=True, gain=1.0, queue_blocks=0):
        H.check(self._l.csdr_mix_set_source(self.h, int(i), int(bound), int(active), float(gain), int(queue_blocks)))

    def push(self, i, data, channels, sample_rate, peak):
        """-> True when the queue took the block (False: full, dropped -- try_push semantics)"""
        a = np.ascontiguousarray(data, dtype=np.float32)
        rc = self._l.csdr_mix_push(self.h, int(i), a.ctypes.data_as(C.c_void_p), 0, a.size, int(channels), int(sample_rate), float(peak))
        if rc == 1:
            return False
        H.check(rc)
        return True

    def push_bank(self, bank, slots, sources=None):
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        so = np.ascontiguousarray(sources if sources is not None else slots, dtype=np.int32)
        H.check(self._l.csdr_mix_push_bank(self.h, bank.h, sl.ctypes.data_as(C.c_void_p), so.ctypes.data_as(C.c_void_p), sl.size))

    def queued(self, i):
        return self._l.csdr_mix_queued(self.h, int(i))

    def render(self, frames, n_buffers=1, fetch=True):
        out = np.empty(n_buffers * frames * 2, np.float32) if fetch else None
        H.check(self._l.csdr_mix_render(self.h, int(frames), int(n_buffers), out.ctypes.data_as(C.c_void_p) if fetch else None))
        return out

    def pcm16(self, peak=1.0, per_buffer_peak=False, cap=1 << 22):
        out = np.empty(cap, np.int16)
        n = C.c_int()
        H.check(self._l.csdr_mix_fetch_pcm16(self.h, out.ctypes.data_as(C.c_void_p), cap, float(peak), int(per_buffer_peak), C.byref(n)))
        return out[:n.value].copy()

    def close(self):
        if self.h:
            self._l.csdr_mix_destroy(self.h)
            self.h = C.c_void_p()


class Comm:
    """csdr_comm: the collectives of ONE stream over the GPUs of a node, RCCL over xGMI behind the C ABI (csdr_comm.hip).  Every rank
    creates one on its own Context with the same 128-byte id (rank 0 makes it: Comm.unique_id(); how it reaches the other ranks is the
    host's business -- parallel.exchange_id uses a TCP store).  Buffers are device memory (torch CUDA tensors / DevicePointer; numpy
    arrays with the host-executing test build); counts are complex samples."""

    ID_BYTES = 128

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(Comm.ID_BYTES)
        H.check(H.lib().csdr_comm_unique_id(buf))
        return buf.raw

    def __init__(self, ctx, unique_id, rank, world):
        self._l = H.lib()
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self.h = C.c_void_p()
        assert len(unique_id) == Comm.ID_BYTES
        H.check(self._l.csdr_comm_create(ctx.h, C.c_char_p(bytes(unique_id)), self.rank, self.world, C.byref(self.h)))

    @property
    def world_size(self):
        """ranks of the communicator as the library reports them (csdr_comm_world)"""
        return int(self._l.csdr_comm_world(self.h))

    @staticmethod
    def _ptr(buf):
        if buf is None:
            return None
        if isinstance(buf, np.ndarray):
            return buf.ctypes.data_as(C.c_void_p)
        if isinstance(buf, DevicePointer):
            return C.c_void_p(buf.ptr)
        return C.c_void_p(buf.data_ptr())

    def broadcast(self, iq, n_samples, root=0):
        H.check(self._l.csdr_comm_broadcast(self.h, self._ptr(iq), int(n_samples), int(root)))

    def scatter(self, send, recv, n_samples, root=0):
        H.check(self._l.csdr_comm_scatter(self.h, self._ptr(send), self._ptr(recv), int(n_samples), int(root)))

    def all_to_all(self, send, send_samples, recv, recv_samples):
        a = np.ascontiguousarray(send_samples, dtype=np.int64)
        b = np.ascontiguousarray(recv_samples, dtype=np.int64)
        assert a.size == self.world and b.size == self.world
        H.check(self._l.csdr_comm_all_to_all(self.h, self._ptr(send), a.ctypes.data_as(C.c_void_p), self._ptr(recv), b.ctypes.data_as(C.c_void_p)))

    def p2p(self, ops):
        """ops: [(peer, recv, buffer, byte_offset_in_samples, n_samples)] -- one grouped set of sends (recv False) and receives"""
        arr = (H.P2pOp * max(1, len(ops)))()
        for i, (peer, recv, buf, off, n) in enumerate(ops):
            base = self._ptr(buf).value or 0
            arr[i] = H.P2pOp(int(peer), 1 if recv else 0, base + 8 * int(off), int(n))
        H.check(self._l.csdr_comm_p2p(self.h, arr, len(ops)))

    def max(self, value):
        v = C.c_double(float(value))
        H.check(self._l.csdr_comm_max(self.h, C.byref(v)))
        return v.value

    def barrier(self):
        H.check(self._l.csdr_comm_barrier(self.h))

    def exchange_rows(self, producer, owner, owned, frame0, frames, n_blocks, block_len, frequency):
        """owned: per-rank channel lists; frame0 / frames: per-rank slab position inside the batch (frames = samples per channel)"""
        ch = np.ascontiguousarray([c for o in owned for c in o] or [0], dtype=np.int32)
        nch = np.ascontiguousarray([len(o) for o in owned], dtype=np.int32)
        f0 = np.ascontiguousarray(frame0, dtype=np.int64)
        fr = np.ascontiguousarray(frames, dtype=np.int64)
        assert nch.size == self.world and f0.size == self.world and fr.size == self.world
        H.check(self._l.csdr_post_exchange_rows(self.h, producer.h, owner.h, ch.ctypes.data_as(C.c_void_p), nch.ctypes.data_as(C.c_void_p),
                                                f0.ctypes.data_as(C.c_void_p), fr.ctypes.data_as(C.c_void_p), int(n_blocks), int(block_len), int(frequency)))

    def exchange_rows_begin(self, producer, owned, frame0, frames):
        """first half of exchange_rows: the transfers of the producer's current batch, on the communicator's own stream"""
        ch = np.ascontiguousarray([c for o in owned for c in o] or [0], dtype=np.int32)
        nch = np.ascontiguousarray([len(o) for o in owned], dtype=np.int32)
        f0 = np.ascontiguousarray(frame0, dtype=np.int64)
        fr = np.ascontiguousarray(frames, dtype=np.int64)
        assert nch.size == self.world and f0.size == self.world and fr.size == self.world
        H.check(self._l.csdr_post_exchange_rows_begin(self.h, producer.h, ch.ctypes.data_as(C.c_void_p), nch.ctypes.data_as(C.c_void_p),
                                                      f0.ctypes.data_as(C.c_void_p), fr.ctypes.data_as(C.c_void_p)))

    def exchange_rows_finish(self, owner, n_blocks, block_len, frequency):
        """second half, for the oldest batch begun: import into the owner behind that batch's transfers, commit"""
        H.check(self._l.csdr_post_exchange_rows_finish(self.h, owner.h, int(n_blocks), int(block_len), int(frequency)))
        owner._last = (int(n_blocks), int(block_len))

    @property
    def exchanges_pending(self):
        return int(self._l.csdr_comm_exchanges_pending(self.h))

    def abort(self):
        H.check(self._l.csdr_comm_abort(self.h))

    def async_error(self):
        """raises if a transfer of this communicator has failed (the communicator is then aborted on this rank too)"""
        H.check(self._l.csdr_comm_async_error(self.h))

    def close(self):
        if self.h:
            self._l.csdr_comm_destroy(self.h)
            self.h = C.c_void_p()


class Ingest:
    """page-locked block ring -> HBM, ONE transfer per block (csdr_ingest); commit returns the device pointer as an integer"""

    def __init__(self, ctx, max_samples, depth=3):
        self._l = H.lib()
        self.ctx = ctx
        self.h = C.c_void_p()
        self.max_samples = int(max_samples)
        H.check(self._l.csdr_ingest_create(ctx.h, self.max_samples, int(depth), C.byref(self.h)))

    def acquire(self):
        """-> numpy complex64 view of the page-locked slot (max_samples long)"""
        p = C.c_void_p()
        H.check(self._l.csdr_ingest_acquire(self.h, C.byref(p)))
        buf = (C.c_float * (2 * self.max_samples)).from_address(p.value)
        return np.frombuffer(buf, dtype=np.complex64)

    def commit(self, n_samples, iq_swap=False):
        p = C.c_void_p()
        H.check(self._l.csdr_ingest_commit(self.h, int(n_samples), int(iq_swap), C.byref(p)))
        return DevicePointer(p.value, int(n_samples))

    def close(self):
        if self.h:
            self._l.csdr_ingest_destroy(self.h)
            self.h = C.c_void_p()


class DevicePointer:
    """a raw device IQ buffer (complex64 samples) that SDRPost.execute / SpectrumProcessor.process accept like a CUDA tensor"""

    def __init__(self, ptr, n):
        self.ptr, self.n = ptr, n
