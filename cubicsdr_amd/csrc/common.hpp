// common.hpp -- shared host/device helpers for the csdr_hip library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/csdr_hip.h"

namespace csdr {

inline std::string &last_error_ref() {
    static thread_local std::string s;
    return s;
}
inline int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

#define CSDR_HIP_TRY(expr)                                                                      \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return ::csdr::fail(CSDR_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

// device buffer that only ever grows
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return CSDR_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc((void **)&p, n * sizeof(T)) != hipSuccess) return fail(CSDR_ENOMEM, "hipMalloc(%zu bytes) failed", n * sizeof(T));
        cap = n;
        return CSDR_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// pinned host buffer
template <typename T>
struct PinBuf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return CSDR_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        if (hipHostMalloc((void **)&p, n * sizeof(T), hipHostMallocDefault) != hipSuccess) return fail(CSDR_ENOMEM, "hipHostMalloc(%zu bytes) failed", n * sizeof(T));
        cap = n;
        return CSDR_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// ---- small device helpers shared by the kernel headers ----
__device__ inline float2 cmul(float2 a, float2 b) { return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)); }
__device__ inline float2 cfma(float2 v, float2 w, float2 acc) {   // acc + v * w
    acc.x = fmaf(v.x, w.x, acc.x); acc.x = fmaf(-v.y, w.y, acc.x);
    acc.y = fmaf(v.x, w.y, acc.y); acc.y = fmaf(v.y, w.x, acc.y);
    return acc;
}
// a value that is the same in every lane of the wave by construction: tell the compiler (scalar registers, scalar loads)
__device__ inline int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// 16 bytes that are only guaranteed 8-byte aligned (two adjacent complex samples at an odd sample offset)
struct __attribute__((aligned(8))) f4u { float x, y, z, w; };

}  // namespace csdr

// kernel ids for the optional per-kernel HIP-event profile (csdr_ctx_profile_*)
enum CsdrKernelId {
    KID_CHAN_ANALYZE = 0, KID_DC_ENDS, KID_DC_APPLY,
    KID_FRONTEND, KID_MODEM, KID_AUDIO,
    KID_FFT_COLS, KID_FFT_ROWS, KID_SPEC_AVG, KID_SPEC_TRACK, KID_SPEC_DISPLAY, KID_SPEC_MISC,
    KID_COUNT
};

struct csdr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    csdr::DevBuf<float> sintab;  // 1024-entry sine table of the reference's NCO
    // per-kernel profile: event pairs recorded around launches while enabled
    bool prof_on = false;
    struct ProfRec { int id; hipEvent_t a, b; };
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[KID_COUNT] = {0};
    long long prof_n[KID_COUNT] = {0};
    hipEvent_t prof_event() {
        if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};

// bracket one kernel launch with events when profiling is on
struct ProfScope {
    csdr_ctx *c; int id; hipEvent_t a = nullptr;
    ProfScope(csdr_ctx *c_, int id_) : c(c_), id(id_) { if (c->prof_on) { a = c->prof_event(); (void)hipEventRecord(a, c->stream); } }
    ~ProfScope() { if (a) { hipEvent_t b = c->prof_event(); (void)hipEventRecord(b, c->stream); c->prof_pending.push_back({id, a, b}); } }
};
#define CSDR_LAUNCH(ctx_, kid_, kern_, grid_, block_, lds_, ...) \
    do { ProfScope ps__((ctx_), (kid_)); hipLaunchKernelGGL(kern_, grid_, block_, lds_, (ctx_)->stream, __VA_ARGS__); } while (0)
