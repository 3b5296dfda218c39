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

}  // namespace csdr

struct csdr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    csdr::DevBuf<float> sintab;  // 1024-entry sine table of the reference's NCO
};
