/*
 * liquid_ref.c -- TEST INFRASTRUCTURE ONLY (oracle/). Never linked into the product.
 *
 * SysV-ABI wrappers around the reference's vendored liquid-dsp 1.5.0 binary (loaded by
 * pe_loader.c).  Each wrapper has the name and argument meaning of the liquid function that
 * CubicSDR calls on the hot path (declarations: reference external/liquid-dsp/include/liquid/liquid.h;
 * call sites: SURVEY.md section 2.3) and forwards to the ms_abi export of the DLL.
 *
 * `liquid_float_complex` passed BY VALUE travels as one 64-bit integer register in the
 * mingw ms_abi (verified by disassembling firhilbf_c2r_execute), hence the u64 packing.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MS __attribute__((ms_abi))

int pe_load(const char *path);
void *pe_sym(const char *name);

typedef struct { float re, im; } cf32;
typedef void *obj;

static int g_loaded;

int liquid_ref_load(const char *path)
{
    if (g_loaded) return 0;
    char buf[4096];
    if (!path || !*path) path = getenv("LIQUID_REF_DLL");
    if (!path || !*path) {
        Dl_info di;
        if (dladdr((void *)liquid_ref_load, &di) && di.dli_fname) {
            snprintf(buf, sizeof buf, "%s", di.dli_fname);
            char *s = strrchr(buf, '/');
            if (s) { strcpy(s + 1, "libliquid.dll"); path = buf; }
        }
    }
    if (!path) return -100;
    int rc = pe_load(path);
    if (rc == 0) g_loaded = 1;
    return rc;
}

static void *must(const char *name)
{
    if (!g_loaded && liquid_ref_load(NULL)) { fprintf(stderr, "[liquid_ref] DLL not loaded\n"); abort(); }
    void *p = pe_sym(name);
    if (!p) { fprintf(stderr, "[liquid_ref] missing export %s\n", name); abort(); }
    return p;
}

#define FN(ret, name, ...) \
    typedef ret(MS *name##_t)(__VA_ARGS__); \
    static name##_t name##_p; \
    static inline name##_t name##_get(void) { if (!name##_p) name##_p = (name##_t)must(#name); return name##_p; }

static inline uint64_t pack(cf32 x) { uint64_t u; memcpy(&u, &x, 8); return u; }

/* ---- version ---- */
FN(const char *, liquid_libversion, void)
const char *liquid_libversion(void) { return liquid_libversion_get()(); }
FN(int, liquid_libversion_number, void)
int liquid_libversion_number(void) { return liquid_libversion_number_get()(); }

/* ---- nco_crcf (liquid.h nco section; call sites DemodulatorPreThread.cpp:22,157,188,190; ModemUSB.cpp:9-10,55-58) ---- */
FN(obj, nco_crcf_create, int)
obj nco_crcf_create(int t) { return nco_crcf_create_get()(t); }
FN(int, nco_crcf_destroy, obj)
int nco_crcf_destroy(obj q) { return nco_crcf_destroy_get()(q); }
FN(int, nco_crcf_reset, obj)
int nco_crcf_reset(obj q) { return nco_crcf_reset_get()(q); }
FN(int, nco_crcf_set_frequency, obj, float)
int nco_crcf_set_frequency(obj q, float f) { return nco_crcf_set_frequency_get()(q, f); }
FN(float, nco_crcf_get_frequency, obj)
float nco_crcf_get_frequency(obj q) { return nco_crcf_get_frequency_get()(q); }
FN(int, nco_crcf_set_phase, obj, float)
int nco_crcf_set_phase(obj q, float f) { return nco_crcf_set_phase_get()(q, f); }
FN(float, nco_crcf_get_phase, obj)
float nco_crcf_get_phase(obj q) { return nco_crcf_get_phase_get()(q); }
FN(int, nco_crcf_step, obj)
int nco_crcf_step(obj q) { return nco_crcf_step_get()(q); }
FN(int, nco_crcf_cexpf, obj, cf32 *)
int nco_crcf_cexpf(obj q, cf32 *y) { return nco_crcf_cexpf_get()(q, y); }
FN(int, nco_crcf_mix_up, obj, uint64_t, cf32 *)
int nco_crcf_mix_up(obj q, cf32 x, cf32 *y) { return nco_crcf_mix_up_get()(q, pack(x), y); }
FN(int, nco_crcf_mix_down, obj, uint64_t, cf32 *)
int nco_crcf_mix_down(obj q, cf32 x, cf32 *y) { return nco_crcf_mix_down_get()(q, pack(x), y); }
FN(int, nco_crcf_mix_block_up, obj, cf32 *, cf32 *, unsigned)
int nco_crcf_mix_block_up(obj q, cf32 *x, cf32 *y, unsigned n) { return nco_crcf_mix_block_up_get()(q, x, y, n); }
FN(int, nco_crcf_mix_block_down, obj, cf32 *, cf32 *, unsigned)
int nco_crcf_mix_block_down(obj q, cf32 *x, cf32 *y, unsigned n) { return nco_crcf_mix_block_down_get()(q, x, y, n); }

/* ---- msresamp (DemodulatorWorkerThread.cpp:100; DemodulatorPreThread.cpp:209; ModemAnalog.cpp:30,88) ---- */
FN(obj, msresamp_crcf_create, float, float)
obj msresamp_crcf_create(float r, float as) { return msresamp_crcf_create_get()(r, as); }
FN(int, msresamp_crcf_destroy, obj)
int msresamp_crcf_destroy(obj q) { return msresamp_crcf_destroy_get()(q); }
FN(int, msresamp_crcf_print, obj)
int msresamp_crcf_print(obj q) { int r = msresamp_crcf_print_get()(q); fflush(stdout); return r; }
FN(int, msresamp_crcf_reset, obj)
int msresamp_crcf_reset(obj q) { return msresamp_crcf_reset_get()(q); }
FN(float, msresamp_crcf_get_delay, obj)
float msresamp_crcf_get_delay(obj q) { return msresamp_crcf_get_delay_get()(q); }
FN(int, msresamp_crcf_execute, obj, cf32 *, unsigned, cf32 *, unsigned *)
int msresamp_crcf_execute(obj q, cf32 *x, unsigned nx, cf32 *y, unsigned *ny) { return msresamp_crcf_execute_get()(q, x, nx, y, ny); }

/* msresamp_cccf (ModemCW.cpp:124,163: complex interpolation of the IQ stream to the audio rate) */
FN(obj, msresamp_cccf_create, float, float)
obj msresamp_cccf_create(float r, float as) { return msresamp_cccf_create_get()(r, as); }
FN(int, msresamp_cccf_destroy, obj)
int msresamp_cccf_destroy(obj q) { return msresamp_cccf_destroy_get()(q); }
FN(int, msresamp_cccf_execute, obj, cf32 *, unsigned, cf32 *, unsigned *)
int msresamp_cccf_execute(obj q, cf32 *x, unsigned nx, cf32 *y, unsigned *ny) { return msresamp_cccf_execute_get()(q, x, nx, y, ny); }

FN(obj, msresamp_rrrf_create, float, float)
obj msresamp_rrrf_create(float r, float as) { return msresamp_rrrf_create_get()(r, as); }
FN(int, msresamp_rrrf_destroy, obj)
int msresamp_rrrf_destroy(obj q) { return msresamp_rrrf_destroy_get()(q); }
FN(int, msresamp_rrrf_print, obj)
int msresamp_rrrf_print(obj q) { int r = msresamp_rrrf_print_get()(q); fflush(stdout); return r; }
FN(float, msresamp_rrrf_get_delay, obj)
float msresamp_rrrf_get_delay(obj q) { return msresamp_rrrf_get_delay_get()(q); }
FN(int, msresamp_rrrf_execute, obj, float *, unsigned, float *, unsigned *)
int msresamp_rrrf_execute(obj q, float *x, unsigned nx, float *y, unsigned *ny) { return msresamp_rrrf_execute_get()(q, x, nx, y, ny); }

/* ---- msresamp2 / resamp2 / resamp building blocks (used to pin the restatement stage by stage) ---- */
FN(obj, msresamp2_crcf_create, int, unsigned, float, float, float)
obj msresamp2_crcf_create(int t, unsigned s, float fc, float f0, float as) { return msresamp2_crcf_create_get()(t, s, fc, f0, as); }
FN(int, msresamp2_crcf_destroy, obj)
int msresamp2_crcf_destroy(obj q) { return msresamp2_crcf_destroy_get()(q); }
FN(int, msresamp2_crcf_print, obj)
int msresamp2_crcf_print(obj q) { int r = msresamp2_crcf_print_get()(q); fflush(stdout); return r; }
FN(int, msresamp2_crcf_execute, obj, cf32 *, cf32 *)
int msresamp2_crcf_execute(obj q, cf32 *x, cf32 *y) { return msresamp2_crcf_execute_get()(q, x, y); }

FN(obj, resamp2_crcf_create, unsigned, float, float)
obj resamp2_crcf_create(unsigned m, float f0, float as) { return resamp2_crcf_create_get()(m, f0, as); }
FN(int, resamp2_crcf_destroy, obj)
int resamp2_crcf_destroy(obj q) { return resamp2_crcf_destroy_get()(q); }
FN(int, resamp2_crcf_print, obj)
int resamp2_crcf_print(obj q) { int r = resamp2_crcf_print_get()(q); fflush(stdout); return r; }
FN(int, resamp2_crcf_decim_execute, obj, cf32 *, cf32 *)
int resamp2_crcf_decim_execute(obj q, cf32 *x, cf32 *y) { return resamp2_crcf_decim_execute_get()(q, x, y); }
FN(int, resamp2_crcf_interp_execute, obj, uint64_t, cf32 *)
int resamp2_crcf_interp_execute(obj q, cf32 x, cf32 *y) { return resamp2_crcf_interp_execute_get()(q, pack(x), y); }

FN(obj, resamp2_rrrf_create, unsigned, float, float)
obj resamp2_rrrf_create(unsigned m, float f0, float as) { return resamp2_rrrf_create_get()(m, f0, as); }
FN(int, resamp2_rrrf_destroy, obj)
int resamp2_rrrf_destroy(obj q) { return resamp2_rrrf_destroy_get()(q); }
FN(int, resamp2_rrrf_interp_execute, obj, float, float *)
int resamp2_rrrf_interp_execute(obj q, float x, float *y) { return resamp2_rrrf_interp_execute_get()(q, x, y); }
FN(int, resamp2_rrrf_decim_execute, obj, float *, float *)
int resamp2_rrrf_decim_execute(obj q, float *x, float *y) { return resamp2_rrrf_decim_execute_get()(q, x, y); }

FN(obj, resamp_crcf_create, float, unsigned, float, float, unsigned)
obj resamp_crcf_create(float r, unsigned m, float fc, float as, unsigned npfb) { return resamp_crcf_create_get()(r, m, fc, as, npfb); }
FN(int, resamp_crcf_destroy, obj)
int resamp_crcf_destroy(obj q) { return resamp_crcf_destroy_get()(q); }
FN(int, resamp_crcf_print, obj)
int resamp_crcf_print(obj q) { int r = resamp_crcf_print_get()(q); fflush(stdout); return r; }
FN(int, resamp_crcf_execute_block, obj, cf32 *, unsigned, cf32 *, unsigned *)
int resamp_crcf_execute_block(obj q, cf32 *x, unsigned nx, cf32 *y, unsigned *ny) { return resamp_crcf_execute_block_get()(q, x, nx, y, ny); }

FN(obj, resamp_rrrf_create, float, unsigned, float, float, unsigned)
obj resamp_rrrf_create(float r, unsigned m, float fc, float as, unsigned npfb) { return resamp_rrrf_create_get()(r, m, fc, as, npfb); }
FN(int, resamp_rrrf_destroy, obj)
int resamp_rrrf_destroy(obj q) { return resamp_rrrf_destroy_get()(q); }
FN(int, resamp_rrrf_execute_block, obj, float *, unsigned, float *, unsigned *)
int resamp_rrrf_execute_block(obj q, float *x, unsigned nx, float *y, unsigned *ny) { return resamp_rrrf_execute_block_get()(q, x, nx, y, ny); }

/* ---- firpfbch (SDRPostThread.cpp:406,449-451) / firpfbch2 (:463,505-507) ---- */
FN(obj, firpfbch_crcf_create_kaiser, int, unsigned, unsigned, float)
obj firpfbch_crcf_create_kaiser(int t, unsigned M, unsigned m, float as) { return firpfbch_crcf_create_kaiser_get()(t, M, m, as); }
FN(int, firpfbch_crcf_destroy, obj)
int firpfbch_crcf_destroy(obj q) { return firpfbch_crcf_destroy_get()(q); }
FN(int, firpfbch_crcf_reset, obj)
int firpfbch_crcf_reset(obj q) { return firpfbch_crcf_reset_get()(q); }
FN(int, firpfbch_crcf_analyzer_execute, obj, cf32 *, cf32 *)
int firpfbch_crcf_analyzer_execute(obj q, cf32 *x, cf32 *y) { return firpfbch_crcf_analyzer_execute_get()(q, x, y); }

FN(obj, firpfbch2_crcf_create_kaiser, int, unsigned, unsigned, float)
obj firpfbch2_crcf_create_kaiser(int t, unsigned M, unsigned m, float as) { return firpfbch2_crcf_create_kaiser_get()(t, M, m, as); }
FN(int, firpfbch2_crcf_destroy, obj)
int firpfbch2_crcf_destroy(obj q) { return firpfbch2_crcf_destroy_get()(q); }
FN(int, firpfbch2_crcf_execute, obj, cf32 *, cf32 *)
int firpfbch2_crcf_execute(obj q, cf32 *x, cf32 *y) { return firpfbch2_crcf_execute_get()(q, x, y); }

/* ---- iirfilt_crcf (SDRPostThread.cpp:29,284,375; ModemUSB.cpp:8,56) ---- */
FN(obj, iirfilt_crcf_create_dc_blocker, float)
obj iirfilt_crcf_create_dc_blocker(float a) { return iirfilt_crcf_create_dc_blocker_get()(a); }
FN(obj, iirfilt_crcf_create_lowpass, unsigned, float)
obj iirfilt_crcf_create_lowpass(unsigned o, float fc) { return iirfilt_crcf_create_lowpass_get()(o, fc); }
FN(int, iirfilt_crcf_destroy, obj)
int iirfilt_crcf_destroy(obj q) { return iirfilt_crcf_destroy_get()(q); }
FN(int, iirfilt_crcf_print, obj)
int iirfilt_crcf_print(obj q) { int r = iirfilt_crcf_print_get()(q); fflush(stdout); return r; }
FN(int, iirfilt_crcf_reset, obj)
int iirfilt_crcf_reset(obj q) { return iirfilt_crcf_reset_get()(q); }
FN(int, iirfilt_crcf_execute, obj, uint64_t, cf32 *)
int iirfilt_crcf_execute(obj q, cf32 x, cf32 *y) { return iirfilt_crcf_execute_get()(q, pack(x), y); }
FN(int, iirfilt_crcf_execute_block, obj, cf32 *, unsigned, cf32 *)
int iirfilt_crcf_execute_block(obj q, cf32 *x, unsigned n, cf32 *y) { return iirfilt_crcf_execute_block_get()(q, x, n, y); }

/* ---- fft (SpectrumVisualProcessor.cpp:177,439) ---- */
FN(obj, fft_create_plan, unsigned, cf32 *, cf32 *, int, int)
obj fft_create_plan(unsigned n, cf32 *x, cf32 *y, int dir, int fl) { return fft_create_plan_get()(n, x, y, dir, fl); }
FN(int, fft_destroy_plan, obj)
int fft_destroy_plan(obj q) { return fft_destroy_plan_get()(q); }
FN(int, fft_execute, obj)
int fft_execute(obj q) { return fft_execute_get()(q); }
FN(int, fft_print_plan, obj)
int fft_print_plan(obj q) { int r = fft_print_plan_get()(q); fflush(stdout); return r; }

/* ---- freqdem (ModemNBFM.cpp:7,36) ---- */
FN(obj, freqdem_create, float)
obj freqdem_create(float kf) { return freqdem_create_get()(kf); }
FN(int, freqdem_destroy, obj)
int freqdem_destroy(obj q) { return freqdem_destroy_get()(q); }
FN(int, freqdem_reset, obj)
int freqdem_reset(obj q) { return freqdem_reset_get()(q); }
FN(int, freqdem_demodulate_block, obj, cf32 *, unsigned, float *)
int freqdem_demodulate_block(obj q, cf32 *x, unsigned n, float *y) { return freqdem_demodulate_block_get()(q, x, n, y); }

/* ---- firfilt_rrrf (ModemAM.cpp:9,45-46) ---- */
FN(obj, firfilt_rrrf_create_dc_blocker, unsigned, float)
obj firfilt_rrrf_create_dc_blocker(unsigned m, float as) { return firfilt_rrrf_create_dc_blocker_get()(m, as); }
FN(int, firfilt_rrrf_destroy, obj)
int firfilt_rrrf_destroy(obj q) { return firfilt_rrrf_destroy_get()(q); }
FN(int, firfilt_rrrf_push, obj, float)
int firfilt_rrrf_push(obj q, float x) { return firfilt_rrrf_push_get()(q, x); }
FN(int, firfilt_rrrf_execute, obj, float *)
int firfilt_rrrf_execute(obj q, float *y) { return firfilt_rrrf_execute_get()(q, y); }
FN(int, firfilt_rrrf_execute_block, obj, float *, unsigned, float *)
int firfilt_rrrf_execute_block(obj q, float *x, unsigned n, float *y) { return firfilt_rrrf_execute_block_get()(q, x, n, y); }
FN(unsigned, firfilt_rrrf_get_length, obj)
unsigned firfilt_rrrf_get_length(obj q) { return firfilt_rrrf_get_length_get()(q); }

/* ---- firhilbf (ModemUSB.cpp:11,60) ---- */
FN(obj, firhilbf_create, unsigned, float)
obj firhilbf_create(unsigned m, float as) { return firhilbf_create_get()(m, as); }
FN(int, firhilbf_destroy, obj)
int firhilbf_destroy(obj q) { return firhilbf_destroy_get()(q); }
FN(int, firhilbf_print, obj)
int firhilbf_print(obj q) { int r = firhilbf_print_get()(q); fflush(stdout); return r; }
FN(int, firhilbf_c2r_execute, obj, uint64_t, float *, float *)
int firhilbf_c2r_execute(obj q, cf32 x, float *y0, float *y1) { return firhilbf_c2r_execute_get()(q, pack(x), y0, y1); }

/* ---- filter design helpers (called inside the create() functions above; exposed to pin the restatement) ---- */
FN(unsigned, estimate_req_filter_len, float, float)
unsigned estimate_req_filter_len(float df, float as) { return estimate_req_filter_len_get()(df, as); }
FN(float, kaiser_beta_As, float)
float kaiser_beta_As(float as) { return kaiser_beta_As_get()(as); }
FN(int, liquid_firdes_kaiser, unsigned, float, float, float, float *)
int liquid_firdes_kaiser(unsigned n, float fc, float as, float mu, float *h) { return liquid_firdes_kaiser_get()(n, fc, as, mu, h); }
FN(int, liquid_firdes_notch, unsigned, float, float, float *)
int liquid_firdes_notch(unsigned m, float f0, float as, float *h) { return liquid_firdes_notch_get()(m, f0, as, h); }
FN(float, liquid_besseli0f, float)
float liquid_besseli0f(float z) { return liquid_besseli0f_get()(z); }
FN(float, liquid_kaiser, unsigned, unsigned, float)
float liquid_kaiser(unsigned i, unsigned n, float beta) { return liquid_kaiser_get()(i, n, beta); }
FN(float, sincf, float)
float sincf(float x) { return sincf_get()(x); }

/* ---- block helpers (loops the reference runs in C++; kept in C so the CPU baseline is not python-bound) ---- */
#include <math.h>
/* SDRPostThread.cpp:449-451 */
int oracle_firpfbch_analyzer_block(obj q, unsigned M, cf32 *x, unsigned nframes, cf32 *y)
{ firpfbch_crcf_analyzer_execute_t f = firpfbch_crcf_analyzer_execute_get(); for (unsigned i = 0; i < nframes; i++) f(q, x + (size_t)i * M, y + (size_t)i * M); return 0; }
/* ampmodem (ModemDSB.cpp:6,49-51): cf32 by value is an 8-byte integer-register aggregate in the MS ABI */
FN(obj, ampmodem_create, float, int, int)
obj ampmodem_create(float mod_index, int type, int suppressed) { return ampmodem_create_get()(mod_index, type, suppressed); }
FN(int, ampmodem_destroy, obj)
int ampmodem_destroy(obj q) { return ampmodem_destroy_get()(q); }
FN(int, ampmodem_demodulate, obj, uint64_t, float *)
int ampmodem_demodulate(obj q, cf32 x, float *y) { return ampmodem_demodulate_get()(q, pack(x), y); }
int oracle_dsb_block(obj q, cf32 *in, unsigned n, float *out)
{ ampmodem_demodulate_t f = ampmodem_demodulate_get(); for (unsigned i = 0; i < n; i++) { uint64_t v; memcpy(&v, &in[i], 8); f(q, v, &out[i]); } return 0; }

/* SDRPostThread.cpp:505-507 */
int oracle_firpfbch2_block(obj q, unsigned M, cf32 *x, unsigned ncalls, cf32 *y)
{ firpfbch2_crcf_execute_t f = firpfbch2_crcf_execute_get(); for (unsigned i = 0; i < ncalls; i++) f(q, x + (size_t)i * (M / 2), y + (size_t)i * M); return 0; }
/* ModemAM.cpp:41-47 */
int oracle_am_block(obj dcblock, cf32 *x, unsigned n, float *y)
{ for (unsigned i = 0; i < n; i++) { float I = x[i].re, Q = x[i].im; firfilt_rrrf_push(dcblock, sqrtf(I * I + Q * Q)); firfilt_rrrf_execute(dcblock, &y[i]); } return 0; }
/* ModemUSB.cpp:54-61 (usb=1) / ModemLSB.cpp (usb=0) */
/* ModemCW.cpp:175-180: mix up by the beep-frequency oscillator, step it, keep the upper-sideband output of the c2r Hilbert transform */
int oracle_cw_block(obj nco, obj hilb, cf32 *in, unsigned n, float *out)
{
    for (unsigned i = 0; i < n; i++) {
        cf32 sig; float lsb;
        nco_crcf_mix_up(nco, in[i], &sig);
        nco_crcf_step(nco);
        firhilbf_c2r_execute(hilb, sig, &lsb, &out[i]);
    }
    return 0;
}
int oracle_ssb_block(obj nco, obj iir, obj hilb, int usb, cf32 *in, unsigned n, float *out)
{
    for (unsigned i = 0; i < n; i++) {
        cf32 x, y; float lo, up;
        nco_crcf_step(nco);
        if (usb) nco_crcf_mix_down(nco, in[i], &x); else nco_crcf_mix_up(nco, in[i], &x);
        iirfilt_crcf_execute(iir, x, &y);
        if (usb) nco_crcf_mix_up(nco, y, &x); else nco_crcf_mix_down(nco, y, &x);
        firhilbf_c2r_execute(hilb, x, &lo, &up);
        out[i] = usb ? up : lo;
    }
    return 0;
}
static inline void ref_peek_u32(const void *p, unsigned off, unsigned *dst) { memcpy(dst, (const char *)p + off, 4); }

/* ---- FM stereo (ModemFMStereo.cpp:113-152 kit, :163-289 demodulate) ---- */
FN(int, firhilbf_r2c_execute, obj, float, cf32 *)
int firhilbf_r2c_execute(obj q, float x, cf32 *y) { return firhilbf_r2c_execute_get()(q, x, y); }
FN(obj, iirfilt_crcf_create_prototype, int, int, int, unsigned, float, float, float, float)
obj iirfilt_crcf_create_prototype(int ft, int bt, int fmt, unsigned order, float fc, float f0, float ap, float as)
{ return iirfilt_crcf_create_prototype_get()(ft, bt, fmt, order, fc, f0, ap, as); }
FN(int, nco_crcf_pll_set_bandwidth, obj, float)
int nco_crcf_pll_set_bandwidth(obj q, float bw) { return nco_crcf_pll_set_bandwidth_get()(q, bw); }
FN(int, nco_crcf_pll_step, obj, float)
int nco_crcf_pll_step(obj q, float dphi) { return nco_crcf_pll_step_get()(q, dphi); }
FN(obj, iirfilt_rrrf_create, float *, unsigned, float *, unsigned)
obj iirfilt_rrrf_create(float *b, unsigned nb, float *a, unsigned na) { return iirfilt_rrrf_create_get()(b, nb, a, na); }
FN(int, iirfilt_rrrf_destroy, obj)
int iirfilt_rrrf_destroy(obj q) { return iirfilt_rrrf_destroy_get()(q); }
FN(int, iirfilt_rrrf_execute, obj, float, float *)
int iirfilt_rrrf_execute(obj q, float x, float *y) { return iirfilt_rrrf_execute_get()(q, x, y); }
FN(obj, firfilt_rrrf_create, float *, unsigned)
obj firfilt_rrrf_create(float *h, unsigned n) { return firfilt_rrrf_create_get()(h, n); }
FN(int, liquid_iirdes, int, int, int, unsigned, float, float, float, float, float *, float *)
int liquid_iirdes(int ft, int bt, int fmt, unsigned n, float fc, float f0, float ap, float as, float *b, float *a)
{ return liquid_iirdes_get()(ft, bt, fmt, n, fc, f0, ap, as, b, a); }
FN(obj, iirfilt_crcf_create_sos, float *, float *, unsigned)
obj iirfilt_crcf_create_sos(float *b, float *a, unsigned nsos) { return iirfilt_crcf_create_sos_get()(b, a, nsos); }
/* the per-sample pilot loop of ModemFMStereo.cpp:198-226; theta_out (optional) records the oscillator phase word after each step */
int oracle_fms_pilot_block(obj r2c, obj bp, obj pll, obj c2r, float *d, unsigned n, float *stereo, unsigned *theta_out)
{
    for (unsigned i = 0; i < n; i++) {
        cf32 x, v, w, u, y; float usb;
        firhilbf_r2c_execute(r2c, d[i], &x);
        iirfilt_crcf_execute(bp, x, &v);
        nco_crcf_cexpf(pll, &w);
        w.im = -w.im;
        u.re = v.re * w.re - v.im * w.im;
        u.im = v.re * w.im + v.im * w.re;
        float pe = atan2f(u.im, u.re);
        nco_crcf_pll_step(pll, pe);
        nco_crcf_step(pll);
        if (theta_out) ref_peek_u32(pll, 0x1004, &theta_out[i]);
        nco_crcf_mix_down(pll, x, &y);
        nco_crcf_mix_down(pll, y, &x);
        firhilbf_c2r_execute(c2r, x, &stereo[i], &usb);
    }
    return 0;
}
/* ModemFMStereo.cpp:263-288: matrix, de-emphasis (dl/dr NULL when demph == 0), 16 kHz low-pass, interleave */
int oracle_fms_matrix_block(obj dl, obj dr, obj fl, obj fr, float *mono, float *st, unsigned n, float *out)
{
    for (unsigned i = 0; i < n; i++) {
        float l, r, ld = 0.568f * (mono[i] - st[i]), rd = 0.568f * (mono[i] + st[i]);
        if (dl) { float a = ld, b = rd; iirfilt_rrrf_execute(dl, a, &ld); iirfilt_rrrf_execute(dr, b, &rd); }
        firfilt_rrrf_push(fl, ld); firfilt_rrrf_execute(fl, &l);
        firfilt_rrrf_push(fr, rd); firfilt_rrrf_execute(fr, &r);
        out[2 * i] = l; out[2 * i + 1] = r;
    }
    return 0;
}
/* raw object peek for pinning integer state (nco theta/d_theta at +0x1004/+0x1008) */
void ref_peek(const void *p, unsigned off, void *dst, unsigned n) { memcpy(dst, (const char *)p + off, n); }
/* and poke: set the oscillator phase word per sample when a stage is checked against given phases (tests/test_gpu_parity.py, FM stereo) */
void ref_poke(void *p, unsigned off, const void *src, unsigned n) { memcpy((char *)p + off, src, n); }
#include "../chain_bench.inc"
