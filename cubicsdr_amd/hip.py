"""ctypes binding of include/csdr_hip.h (the C ABI of libcsdr_hip.so).

There is no CPU fallback: importing works anywhere (so symbol checks can run without a GPU), but creating a
context without a HIP device raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcsdr_hip.so")

CSDR_POST_SINGLE, CSDR_POST_PFBCH, CSDR_POST_PFBCH2 = 0, 1, 2
CSDR_MODEM_NBFM, CSDR_MODEM_FM, CSDR_MODEM_AM, CSDR_MODEM_USB, CSDR_MODEM_LSB, CSDR_MODEM_IQ, CSDR_MODEM_CW, CSDR_MODEM_DSB, CSDR_MODEM_FMS, CSDR_MODEM_HOST = range(10)
CSDR_SPEC_FIRST_FRAME, CSDR_SPEC_CONTIGUOUS, CSDR_SPEC_LINES = 0, 1, 2
MODEM_BY_NAME = {"NBFM": 0, "FM": 1, "AM": 2, "USB": 3, "LSB": 4, "I/Q": 5, "IQ": 5, "CW": 6, "DSB": 7, "FMS": 8, "HOST": 9}


class DemodParams(C.Structure):
    _fields_ = [("modem", C.c_int32), ("bandwidth", C.c_int32), ("audio_sample_rate", C.c_int32),
                ("modem_arg", C.c_int32), ("frequency", C.c_int64)]


class BlockResult(C.Structure):
    _fields_ = [("n_iq", C.c_int32), ("n_audio", C.c_int32), ("audio_offset", C.c_int32), ("skipped", C.c_int32),
                ("level_accum", C.c_double), ("level_count", C.c_int32), ("audio_peak", C.c_float),
                ("nco_theta", C.c_uint32), ("resamp_phase", C.c_uint32), ("buffer_index", C.c_uint32),
                ("reserved", C.c_uint32)]


class P2pOp(C.Structure):
    _fields_ = [("peer", C.c_int32), ("recv", C.c_int32), ("buf", C.c_void_p), ("n_samples", C.c_int64)]


class ScopeFrame(C.Structure):
    _fields_ = [("data", C.c_void_p), ("n_dev", C.c_void_p), ("n", C.c_int32), ("channels", C.c_int32), ("type", C.c_int32),
                ("sample_rate", C.c_int32), ("input_rate", C.c_int32), ("layout", C.c_int32), ("scale", C.c_float)]


class ScopeInfo(C.Structure):
    _fields_ = [("mode", C.c_int32), ("spectrum", C.c_int32), ("channels", C.c_int32), ("input_rate", C.c_int32), ("sample_rate", C.c_int32),
                ("fft_size", C.c_int32), ("n_floats", C.c_int32), ("reserved", C.c_int32), ("fft_floor", C.c_double), ("fft_ceil", C.c_double)]


_p, _i, _i64, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
_pp = C.POINTER(C.c_void_p)

# every symbol include/csdr_hip.h declares: name -> (restype, argtypes)
ABI = {
    "csdr_abi_version": (_i, []),
    "csdr_strerror": (C.c_char_p, [_i]),
    "csdr_last_error": (C.c_char_p, []),
    "csdr_ctx_create": (_i, [_i, _p, _pp]),
    "csdr_ctx_destroy": (None, [_p]),
    "csdr_ctx_synchronize": (_i, [_p]),
    "csdr_ctx_join": (_i, [_p]),
    "csdr_ctx_stream": (_p, [_p]),
    "csdr_ctx_owns_stream": (_i, [_p]),
    "csdr_ctx_timer_start": (_i, [_p]),
    "csdr_ctx_timer_stop": (_i, [_p, C.POINTER(_f)]),
    "csdr_ctx_profile_enable": (_i, [_p, _i]),
    "csdr_ctx_profile_num_kernels": (_i, []),
    "csdr_ctx_profile_kernel_name": (C.c_char_p, [_i]),
    "csdr_ctx_profile_fetch": (_i, [_p, _i, C.POINTER(_d), C.POINTER(_i64)]),
    "csdr_ctx_profile_launches": (_i, [_p, _i, C.POINTER(_i64)]),
    "csdr_ctx_profile_range": (_i, [_p, _i, C.POINTER(_d), C.POINTER(_d)]),
    "csdr_dev_alloc": (_i, [_p, C.c_uint64, _pp]),
    "csdr_dev_free": (_i, [_p, _p]),
    "csdr_dev_upload": (_i, [_p, _p, _p, C.c_uint64]),
    "csdr_dev_download": (_i, [_p, _p, _p, C.c_uint64]),
    "csdr_host_register": (_i, [_p, _p, C.c_uint64]),
    "csdr_host_unregister": (_i, [_p, _p]),
    "csdr_post_create": (_i, [_p, _pp]),
    "csdr_post_destroy": (None, [_p]),
    "csdr_post_configure": (_i, [_p, _i64, _i, _i, _i, _i]),
    "csdr_post_execute": (_i, [_p, _p, _i, _i, _i, _i64]),
    "csdr_post_set_active_channels": (_i, [_p, _p, _i]),
    "csdr_post_channel_bandwidth": (_i64, [_p]),
    "csdr_post_channel_rate": (_i64, [_p]),
    "csdr_post_num_channels": (_i, [_p]),
    "csdr_post_kernel_name": (C.c_char_p, [_p]),
    "csdr_post_set_row_order": (_i, [_p, _p, _i]),
    "csdr_post_channel_center": (_i64, [_p, _i]),
    "csdr_post_channel_at": (_i, [_p, _i64]),
    "csdr_post_read_channel": (_i, [_p, _i, _p, _i, C.POINTER(_i)]),
    "csdr_bank_create": (_i, [_p, _i, _i, _pp]),
    "csdr_bank_destroy": (None, [_p]),
    "csdr_bank_configure_slot": (_i, [_p, _i, C.POINTER(DemodParams), _p]),
    "csdr_bank_set_frequency": (_i, [_p, _i, _i64]),
    "csdr_bank_set_active": (_i, [_p, _i, _i]),
    "csdr_bank_execute": (_i, [_p, _p]),
    "csdr_bank_fetch_results": (_i, [_p, _i, C.POINTER(BlockResult), _i, C.POINTER(_i)]),
    "csdr_bank_fetch_audio": (_i, [_p, _i, _p, _i, C.POINTER(_i)]),
    "csdr_bank_fetch_iq": (_i, [_p, _i, _p, _i, C.POINTER(_i)]),
    "csdr_bank_fetch_demod_output": (_i, [_p, _i, _p, _i, C.POINTER(_i)]),
    "csdr_post_history_length": (_i, [_p]),
    "csdr_post_set_history": (_i, [_p, _p, _i64]),
    "csdr_post_set_dc_blocker": (_i, [_p, _i]),
    "csdr_post_export_rows": (_i, [_p, _p, _i, _p, _i64]),
    "csdr_post_import_begin": (_i, [_p, _i, _i, _i64]),
    "csdr_post_import_rows": (_i, [_p, _p, _i, _p, _i64, _i64, _i64]),
    "csdr_post_import_commit": (_i, [_p]),
    "csdr_design_fms_pilot": (_i, [_i64, _p, _p]),
    "csdr_bank_set_fms_pilot": (_i, [_p, _i, _p, _p]),
    "csdr_bank_fetch_fms_stage": (_i, [_p, _i, _i, _p, _i, C.POINTER(_i)]),
    "csdr_bank_total_audio": (_i, [_p, C.POINTER(_i64)]),
    "csdr_spec_create": (_i, [_p, _pp]),
    "csdr_spec_destroy": (None, [_p]),
    "csdr_spec_setup": (_i, [_p, _i, _i]),
    "csdr_spec_set_average_rate": (_i, [_p, _f]),
    "csdr_spec_set_scale_factor": (_i, [_p, _f]),
    "csdr_spec_set_peak_hold": (_i, [_p, _i]),
    "csdr_spec_get_peak_hold": (_i, [_p]),
    "csdr_spec_set_hide_dc": (_i, [_p, _i]),
    "csdr_spec_set_center_frequency": (_i, [_p, _i64]),
    "csdr_spec_set_bandwidth": (_i, [_p, _i64]),
    "csdr_spec_set_input_frequency": (_i, [_p, _i64]),
    "csdr_spec_set_view": (_i, [_p, _i]),
    "csdr_spec_get_view": (_i, [_p]),
    "csdr_spec_set_input_rate": (_i, [_p, _i64]),
    "csdr_spec_desired_input_size": (_i, [_p]),
    "csdr_spec_process": (_i, [_p, _p, _i, _i, _i, _i]),
    "csdr_spec_frames": (_i, [_p]),
    "csdr_spec_fetch": (_i, [_p, _i, _p, _i, C.POINTER(_d), C.POINTER(_d)]),
    "csdr_spec_fetch_hold": (_i, [_p, _i, _p, _i, C.POINTER(_i)]),
    "csdr_spec_fft_only": (_i, [_p, _p, _p]),
    "csdr_scope_create": (_i, [_p, _pp]),
    "csdr_scope_destroy": (None, [_p]),
    "csdr_scope_setup": (_i, [_p, _i, _i, _i]),
    "csdr_scope_set_enabled": (_i, [_p, _i, _i]),
    "csdr_scope_set_max_scope_samples": (_i, [_p, _i]),
    "csdr_scope_set_average_rate": (_i, [_p, _f]),
    "csdr_scope_process": (_i, [_p, C.POINTER(ScopeFrame), _i, _i]),
    "csdr_scope_frames": (_i, [_p]),
    "csdr_scope_fetch": (_i, [_p, _i, _i, _p, _i, C.POINTER(ScopeInfo)]),
    "csdr_bank_scope_frame": (_i, [_p, _i, C.POINTER(ScopeFrame)]),
    "csdr_mix_create": (_i, [_p, _i, _i, _i, _pp]),
    "csdr_mix_destroy": (None, [_p]),
    "csdr_mix_set_source": (_i, [_p, _i, _i, _i, _f, _i]),
    "csdr_mix_push": (_i, [_p, _i, _p, _i, _i, _i, _i, _f]),
    "csdr_mix_push_bank": (_i, [_p, _p, _p, _p, _i]),
    "csdr_mix_queued": (_i, [_p, _i]),
    "csdr_mix_render": (_i, [_p, _i, _i, _p]),
    "csdr_mix_fetch_pcm16": (_i, [_p, _p, _i, _f, _i, C.POINTER(_i)]),
    "csdr_bank_fetch_pcm16": (_i, [_p, _i, _p, _i, C.POINTER(_i)]),
    "csdr_ingest_create": (_i, [_p, _i64, _i, _pp]),
    "csdr_ingest_destroy": (None, [_p]),
    "csdr_ingest_acquire": (_i, [_p, _pp]),
    "csdr_ingest_commit": (_i, [_p, _i64, _i, _pp]),
    "csdr_ingest_upload": (_i, [_p, _p, _i64, _i, _pp]),
    "csdr_ingest_wait": (_i, [_p]),
    "csdr_comm_unique_id": (_i, [_p]),
    "csdr_comm_create": (_i, [_p, _p, _i, _i, _pp]),
    "csdr_comm_destroy": (None, [_p]),
    "csdr_comm_rank": (_i, [_p]),
    "csdr_comm_world": (_i, [_p]),
    "csdr_comm_broadcast": (_i, [_p, _p, _i64, _i]),
    "csdr_comm_scatter": (_i, [_p, _p, _p, _i64, _i]),
    "csdr_comm_all_to_all": (_i, [_p, _p, _p, _p, _p]),
    "csdr_comm_p2p": (_i, [_p, _p, _i]),
    "csdr_comm_max": (_i, [_p, C.POINTER(_d)]),
    "csdr_comm_barrier": (_i, [_p]),
    "csdr_post_exchange_rows": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i64]),
    "csdr_post_exchange_rows_begin": (_i, [_p, _p, _p, _p, _p, _p]),
    "csdr_post_exchange_rows_finish": (_i, [_p, _p, _i, _i, _i64]),
    "csdr_comm_exchanges_pending": (_i, [_p]),
    "csdr_comm_abort": (_i, [_p]),
    "csdr_comm_async_error": (_i, [_p]),
    "csdr_ingest_next_slot": (_i, [_p]),
}

_lib = None


class CsdrError(RuntimeError):
    pass


def lib():
    """Load libcsdr_hip.so (built in-tree by cubicsdr_amd/build.py).  Fails loudly when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CsdrError("HIP extension %s is missing: run `python -m cubicsdr_amd.build` (no CPU fallback exists)" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in ABI.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        l = lib()
        raise CsdrError("%s: %s" % (l.csdr_strerror(rc).decode(), l.csdr_last_error().decode()))


def _host_ptr(a):
    return a.ctypes.data_as(C.c_void_p)
