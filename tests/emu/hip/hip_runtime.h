// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.  NOT part of the product, never shipped, never loaded by
// cubicsdr_amd/.
//
// A host-thread emulation of the tiny subset of the HIP runtime + device language that cubicsdr_amd/csrc uses, so
// that the *logic* of the kernels (indexing, LDS carve-outs, barrier placement, integer bookkeeping) can be exercised
// in the CPU-only container -- under AddressSanitizer / ThreadSanitizer -- before GPU minutes are spent on them.
// tests/emu/build_emu.py compiles the UNMODIFIED product sources with g++ and `-I tests/emu`, which makes
// `#include <hip/hip_runtime.h>` resolve to this file.  The result (tests/emu/_build/libcsdr_emu*.so) is loaded only by
// tests/test_emu_logic.py.  It proves nothing about performance or about the real device; the parity tests proper are
// the `-m gpu` tests against libcsdr_hip.so.
//
// Model: one workgroup at a time; every work-item is a std::thread; __syncthreads() is a std::barrier; a work-item
// that returns from the kernel drops out of the barrier (as an exited wave does on the hardware); dynamic LDS is one
// global buffer; wave64 cross-lane ops go through an exchange buffer and a per-wave barrier (wave-uniform control flow).
#pragma once
#define CSDR_HIP_EMULATION 1      /* host-executing test build: no RCCL, no device */
#include <algorithm>
#include <barrier>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <tuple>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ /* only `extern __shared__ ... char smem[]` is supported: resolves to the global csdr::smem */

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct int2 { int x, y; };
struct uint2 { unsigned x, y; };
struct int4 { int x, y, z, w; };
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct hipEmuStream *hipStream_t;
typedef struct hipEmuEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDisableTiming = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace hip_emu {
struct State {
    dim3 grid, block;
    std::unique_ptr<std::barrier<>> bar;   // __syncthreads of the running workgroup
    std::unique_ptr<std::barrier<>> wbar[16];   // wave-level barriers (64 work-items each)
    unsigned char xch[1024][16];           // cross-lane exchange
};
State &state();
extern thread_local dim3 t_threadIdx, t_blockIdx;
void run(dim3 grid, dim3 block, size_t lds_bytes, void (*thunk)(void *), void *arg);
inline void sync() { state().bar->arrive_and_wait(); }
inline unsigned flat_tid() { return t_threadIdx.x; }
template <typename T>
inline T lane_exchange(T v, int src_lane_in_wave, bool valid) {
    static_assert(sizeof(T) <= 16, "exchange slot");
    State &s = state();
    const unsigned tid = flat_tid();
    std::memcpy(s.xch[tid], &v, sizeof(T));
    s.wbar[tid >> 6]->arrive_and_wait();           // cross-lane ops involve the 64 lanes of one wave only
    T r = v;
    if (valid) {
        const unsigned src = (tid & ~63u) + (unsigned)src_lane_in_wave;
        if (src < s.block.x) std::memcpy(&r, s.xch[src], sizeof(T));
    }
    s.wbar[tid >> 6]->arrive_and_wait();
    return r;
}
}  // namespace hip_emu

#define threadIdx (hip_emu::t_threadIdx)
#define blockIdx (hip_emu::t_blockIdx)
#define blockDim (hip_emu::state().block)
#define gridDim (hip_emu::state().grid)
static const int warpSize = 64;

static inline void __syncthreads() { hip_emu::sync(); }
// wave-level synchronisation: on the device the 64 lanes run in lockstep; here they are 64 host threads
static inline void __builtin_amdgcn_wave_barrier() { hip_emu::state().wbar[hip_emu::flat_tid() >> 6]->arrive_and_wait(); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
template <typename T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    const int lane = (int)(hip_emu::flat_tid() & 63u);
    const int src = lane + (int)delta;
    return hip_emu::lane_exchange(v, src, (lane % width) + (int)delta < width && src < 64);
}
template <typename T>
static inline T __shfl_up(T v, unsigned delta, int width = 64) {
    const int lane = (int)(hip_emu::flat_tid() & 63u);
    const int src = lane - (int)delta;
    return hip_emu::lane_exchange(v, src, (lane % width) - (int)delta >= 0);
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    const int lane = (int)(hip_emu::flat_tid() & 63u);
    (void)width;
    return hip_emu::lane_exchange(v, lane ^ mask, (lane ^ mask) < 64);
}
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
    const int lane = (int)(hip_emu::flat_tid() & 63u);
    const int base = lane - (lane % width);
    return hip_emu::lane_exchange(v, base + (src % width), true);
}

// v_mfma_f32_16x16x4_f32 on one wave: lane l supplies A[l & 15][l >> 4], B[l >> 4][l & 15]; element r of the result is
// D[4 (l >> 4) + r][l & 15] = fma chain over k ascending (exact f32, as the hardware's)
typedef float hip_emu_f32x4 __attribute__((vector_size(16)));
static inline hip_emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hip_emu_f32x4 c, int, int, int) {
    hip_emu::State &s = hip_emu::state();
    const unsigned tid = hip_emu::flat_tid(), w0 = tid & ~63u, l = tid & 63u;
    float ab[2] = {a, b};
    std::memcpy(s.xch[tid], ab, sizeof ab);
    s.wbar[tid >> 6]->arrive_and_wait();
    hip_emu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const unsigned i = 4 * (l >> 4) + (unsigned)r, j = l & 15u;
        float acc = c[r];
        for (unsigned k = 0; k < 4; ++k) {
            float pa[2], pb[2];
            std::memcpy(pa, s.xch[w0 + i + 16 * k], sizeof pa);      // A[i][k] lives in lane i + 16 k
            std::memcpy(pb, s.xch[w0 + 16 * k + j], sizeof pb);      // B[k][j] lives in lane 16 k + j
            acc = std::fmaf(pa[0], pb[1], acc);
        }
        d[r] = acc;
    }
    s.wbar[tid >> 6]->arrive_and_wait();
    return d;
}

// device math the kernels use beyond <cmath>
using std::max;
using std::min;
static inline float __fmul_rn(float a, float b) { return a * b; }      // (the emulation is built with -ffp-contract=off)
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline unsigned __float_as_uint(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline void sincosf_emu(float x, float *s, float *c) { *s = sinf(x); *c = cosf(x); }

// ---- host runtime subset ------------------------------------------------------------------------------------
const char *hipGetErrorString(hipError_t e);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipGetLastError();
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
enum { hipHostRegisterDefault = 0 };
static inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemset(void *dst, int v, size_t n);
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipFuncSetAttribute(const void *f, hipFuncAttribute a, int v);
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 16 };
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int dev);
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *nb, const void *f, int threads, size_t lds);

template <typename... P, typename... A>
static inline void hipLaunchKernelGGL(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t, A &&...a) {
    std::tuple<P...> args(static_cast<P>(a)...);
    struct Ctx { void (*k)(P...); std::tuple<P...> *t; } ctx{kernel, &args};
    hip_emu::run(grid, block, lds, [](void *p) { Ctx *c = (Ctx *)p; std::apply(c->k, *c->t); }, &ctx);
}
