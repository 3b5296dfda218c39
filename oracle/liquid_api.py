"""ctypes binding shared by the two oracle back-ends (TEST INFRASTRUCTURE ONLY).

* ``load("ref")``  -> oracle/_ref/libliquid_ref.so : the reference's own vendored liquid-dsp 1.5.0 DLL
  (reference external/liquid-dsp/gcc/64/libliquid.dll) executed through oracle/ref/pe_loader.c.
* ``load("port")`` -> oracle/_ref/liboracle_port.so : the plain-C restatement in oracle/liquid_port.c.

Both export the liquid function names CubicSDR calls (reference external/liquid-dsp/include/liquid/liquid.h),
so the same harness drives either.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product (cubicsdr_amd/) never does.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

c_f = C.c_float
c_u = C.c_uint
c_i = C.c_int
c_p = C.c_void_p


class cf32(C.Structure):
    _fields_ = [("re", c_f), ("im", c_f)]


_SIGS = {
    # name: (restype, argtypes)
    "nco_crcf_create": (c_p, [c_i]),
    "nco_crcf_destroy": (c_i, [c_p]),
    "nco_crcf_reset": (c_i, [c_p]),
    "nco_crcf_set_frequency": (c_i, [c_p, c_f]),
    "nco_crcf_get_frequency": (c_f, [c_p]),
    "nco_crcf_set_phase": (c_i, [c_p, c_f]),
    "nco_crcf_get_phase": (c_f, [c_p]),
    "nco_crcf_step": (c_i, [c_p]),
    "nco_crcf_cexpf": (c_i, [c_p, c_p]),
    "nco_crcf_mix_up": (c_i, [c_p, cf32, c_p]),
    "nco_crcf_mix_down": (c_i, [c_p, cf32, c_p]),
    "nco_crcf_mix_block_up": (c_i, [c_p, c_p, c_p, c_u]),
    "nco_crcf_mix_block_down": (c_i, [c_p, c_p, c_p, c_u]),
    "msresamp_crcf_create": (c_p, [c_f, c_f]),
    "msresamp_crcf_destroy": (c_i, [c_p]),
    "msresamp_crcf_print": (c_i, [c_p]),
    "msresamp_crcf_get_delay": (c_f, [c_p]),
    "msresamp_crcf_execute": (c_i, [c_p, c_p, c_u, c_p, c_p]),
    "ampmodem_create": (c_p, [c_f, c_i, c_i]),
    "ampmodem_destroy": (c_i, [c_p]),
    "msresamp_cccf_create": (c_p, [c_f, c_f]),
    "msresamp_cccf_destroy": (c_i, [c_p]),
    "msresamp_cccf_execute": (c_i, [c_p, c_p, c_u, c_p, c_p]),
    "msresamp_rrrf_create": (c_p, [c_f, c_f]),
    "msresamp_rrrf_destroy": (c_i, [c_p]),
    "msresamp_rrrf_print": (c_i, [c_p]),
    "msresamp_rrrf_get_delay": (c_f, [c_p]),
    "msresamp_rrrf_execute": (c_i, [c_p, c_p, c_u, c_p, c_p]),
    "msresamp2_crcf_create": (c_p, [c_i, c_u, c_f, c_f, c_f]),
    "msresamp2_crcf_destroy": (c_i, [c_p]),
    "msresamp2_crcf_print": (c_i, [c_p]),
    "msresamp2_crcf_execute": (c_i, [c_p, c_p, c_p]),
    "resamp2_crcf_create": (c_p, [c_u, c_f, c_f]),
    "resamp2_crcf_destroy": (c_i, [c_p]),
    "resamp2_crcf_print": (c_i, [c_p]),
    "resamp2_crcf_decim_execute": (c_i, [c_p, c_p, c_p]),
    "resamp2_crcf_interp_execute": (c_i, [c_p, cf32, c_p]),
    "resamp2_rrrf_create": (c_p, [c_u, c_f, c_f]),
    "resamp2_rrrf_destroy": (c_i, [c_p]),
    "resamp2_rrrf_interp_execute": (c_i, [c_p, c_f, c_p]),
    "resamp2_rrrf_decim_execute": (c_i, [c_p, c_p, c_p]),
    "resamp_crcf_create": (c_p, [c_f, c_u, c_f, c_f, c_u]),
    "resamp_crcf_destroy": (c_i, [c_p]),
    "resamp_crcf_print": (c_i, [c_p]),
    "resamp_crcf_execute_block": (c_i, [c_p, c_p, c_u, c_p, c_p]),
    "resamp_rrrf_create": (c_p, [c_f, c_u, c_f, c_f, c_u]),
    "resamp_rrrf_destroy": (c_i, [c_p]),
    "resamp_rrrf_execute_block": (c_i, [c_p, c_p, c_u, c_p, c_p]),
    "firpfbch_crcf_create_kaiser": (c_p, [c_i, c_u, c_u, c_f]),
    "firpfbch_crcf_destroy": (c_i, [c_p]),
    "firpfbch_crcf_reset": (c_i, [c_p]),
    "firpfbch_crcf_analyzer_execute": (c_i, [c_p, c_p, c_p]),
    "firpfbch2_crcf_create_kaiser": (c_p, [c_i, c_u, c_u, c_f]),
    "firpfbch2_crcf_destroy": (c_i, [c_p]),
    "firpfbch2_crcf_execute": (c_i, [c_p, c_p, c_p]),
    "iirfilt_crcf_create_dc_blocker": (c_p, [c_f]),
    "iirfilt_crcf_create_lowpass": (c_p, [c_u, c_f]),
    "iirfilt_crcf_destroy": (c_i, [c_p]),
    "iirfilt_crcf_print": (c_i, [c_p]),
    "iirfilt_crcf_reset": (c_i, [c_p]),
    "iirfilt_crcf_execute": (c_i, [c_p, cf32, c_p]),
    "iirfilt_crcf_execute_block": (c_i, [c_p, c_p, c_u, c_p]),
    "fft_create_plan": (c_p, [c_u, c_p, c_p, c_i, c_i]),
    "fft_destroy_plan": (c_i, [c_p]),
    "fft_execute": (c_i, [c_p]),
    "freqdem_create": (c_p, [c_f]),
    "freqdem_destroy": (c_i, [c_p]),
    "freqdem_reset": (c_i, [c_p]),
    "freqdem_demodulate_block": (c_i, [c_p, c_p, c_u, c_p]),
    "firfilt_rrrf_create_dc_blocker": (c_p, [c_u, c_f]),
    "firfilt_rrrf_destroy": (c_i, [c_p]),
    "firfilt_rrrf_push": (c_i, [c_p, c_f]),
    "firfilt_rrrf_execute": (c_i, [c_p, c_p]),
    "firfilt_rrrf_execute_block": (c_i, [c_p, c_p, c_u, c_p]),
    "firfilt_rrrf_get_length": (c_u, [c_p]),
    "firhilbf_create": (c_p, [c_u, c_f]),
    "firhilbf_destroy": (c_i, [c_p]),
    "firhilbf_print": (c_i, [c_p]),
    "firhilbf_c2r_execute": (c_i, [c_p, cf32, c_p, c_p]),
    "firhilbf_r2c_execute": (c_i, [c_p, c_f, c_p]),
    "iirfilt_crcf_create_prototype": (c_p, [c_i, c_i, c_i, c_u, c_f, c_f, c_f, c_f]),
    "liquid_iirdes": (c_i, [c_i, c_i, c_i, c_u, c_f, c_f, c_f, c_f, c_p, c_p]),
    "iirfilt_crcf_create_sos": (c_p, [c_p, c_p, c_u]),
    "nco_crcf_pll_set_bandwidth": (c_i, [c_p, c_f]),
    "nco_crcf_pll_step": (c_i, [c_p, c_f]),
    "iirfilt_rrrf_create": (c_p, [c_p, c_u, c_p, c_u]),
    "iirfilt_rrrf_destroy": (c_i, [c_p]),
    "iirfilt_rrrf_execute": (c_i, [c_p, c_f, c_p]),
    "firfilt_rrrf_create": (c_p, [c_p, c_u]),
    "oracle_fms_pilot_block": (c_i, [c_p, c_p, c_p, c_p, c_p, c_u, c_p, c_p]),
    "oracle_fms_matrix_block": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_u, c_p]),
    "estimate_req_filter_len": (c_u, [c_f, c_f]),
    "kaiser_beta_As": (c_f, [c_f]),
    "liquid_firdes_kaiser": (c_i, [c_u, c_f, c_f, c_f, c_p]),
    "liquid_firdes_notch": (c_i, [c_u, c_f, c_f, c_p]),
    "liquid_besseli0f": (c_f, [c_f]),
    "liquid_kaiser": (c_f, [c_u, c_u, c_f]),
    "sincf": (c_f, [c_f]),
}

LIQUID_NCO, LIQUID_VCO = 0, 1
LIQUID_IIRDES_CHEBY2, LIQUID_IIRDES_BANDPASS, LIQUID_IIRDES_SOS = 2, 2, 0
LIQUID_ANALYZER, LIQUID_SYNTHESIZER = 0, 1
LIQUID_RESAMP_INTERP, LIQUID_RESAMP_DECIM = 0, 1
LIQUID_FFT_FORWARD, LIQUID_FFT_BACKWARD = 1, -1


def lib_path(kind):
    name = {"ref": "libliquid_ref.so", "port": "liboracle_port.so"}[kind]
    return os.path.join(_HERE, "_ref", name)


def available(kind):
    if not os.path.exists(lib_path(kind)):
        return False
    if kind == "ref":
        return os.path.exists(os.path.join(_HERE, "_ref", "libliquid.dll"))
    return True


def load(kind):
    """Return a ctypes library exposing the liquid API for back-end ``kind`` ("ref" | "port")."""
    if kind in _LIBS:
        return _LIBS[kind]
    lib = C.CDLL(lib_path(kind), mode=os.RTLD_LOCAL) if hasattr(os, "RTLD_LOCAL") else C.CDLL(lib_path(kind))
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.restype = res
        fn.argtypes = args
    if kind == "ref":
        lib.liquid_ref_load.argtypes = [C.c_char_p]
        rc = lib.liquid_ref_load(None)
        if rc:
            raise OSError("reference liquid DLL failed to load (rc=%d)" % rc)
    _LIBS[kind] = lib
    return lib


def ptr(a):
    return a.ctypes.data_as(c_p)


# ---- integer state of the objects on the path (the bit-exact items the HIP path reports per block: csdr_block_result) ----------------------
# The restatement has test hooks (port_nco_get_state, port_msresamp_get_state).  The reference binary has none: its words are read where they
# lie in its own objects -- liquid 1.5.0's structures as the reference's libliquid.dll lays them out (x86-64, LLP64):
#   nco_crcf   { int type; float sintab[1024]; uint32 theta @0x1004; uint32 d_theta @0x1008; ... }                  (disassembled)
#   msresamp   { float rate, As; int type; uint num_halfband_stages @12; msresamp2 *halfband @16; float rate_halfband @24;
#                resamp *arbitrary @32; float rate_arbitrary @40; uint buffer_len @44; T *buffer @48; uint buffer_index @56 }
#   resamp     { uint m; float As, fc, rate; uint32 step @16; uint32 phase @20; uint bits_index @24; uint npfb @28; firpfb pfb }
# tests/test_oracle_pin.py pins these offsets against the restatement's hooks on identical inputs.
_REF_NCO_THETA, _REF_NCO_DTHETA = 0x1004, 0x1008
_REF_MSR_S, _REF_MSR_ARB, _REF_MSR_BUFIDX = 12, 32, 56
_REF_RES_STEP, _REF_RES_PHASE = 16, 20


def _addr(q):
    return q if isinstance(q, int) else q.value


def nco_state(kind, q):
    """-> (theta, d_theta): the oscillator's 32-bit phase and frequency words"""
    if kind == "port":
        th, dth = C.c_uint32(), C.c_uint32()
        fn = load("port").port_nco_get_state
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        fn(C.c_void_p(_addr(q)), C.byref(th), C.byref(dth))
        return th.value, dth.value
    a = _addr(q)
    return C.c_uint32.from_address(a + _REF_NCO_THETA).value, C.c_uint32.from_address(a + _REF_NCO_DTHETA).value


def msresamp_state(kind, q):
    """-> dict(S, buffer_index, phase, step) of a msresamp_crcf / _rrrf / _cccf object"""
    if kind == "port":
        S, bi, ph, st = C.c_uint(), C.c_uint(), C.c_uint32(), C.c_uint32()
        fn = load("port").port_msresamp_get_state
        fn.argtypes = [C.c_void_p] * 5
        fn(C.c_void_p(_addr(q)), C.byref(S), C.byref(bi), C.byref(ph), C.byref(st))
        return dict(S=S.value, buffer_index=bi.value, phase=ph.value, step=st.value)
    a = _addr(q)
    arb = C.c_uint64.from_address(a + _REF_MSR_ARB).value
    return dict(S=C.c_uint32.from_address(a + _REF_MSR_S).value, buffer_index=C.c_uint32.from_address(a + _REF_MSR_BUFIDX).value,
                phase=C.c_uint32.from_address(arb + _REF_RES_PHASE).value, step=C.c_uint32.from_address(arb + _REF_RES_STEP).value)


def as_c64(a):
    return np.ascontiguousarray(a, dtype=np.complex64)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def cpx(z):
    return cf32(float(np.real(z)), float(np.imag(z)))
