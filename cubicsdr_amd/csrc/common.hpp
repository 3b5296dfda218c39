// common.hpp -- shared host/device helpers for the csdr_hip library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/csdr_hip.h"

// every kernel has internal linkage: the kernel headers are included by several translation units (for the object layouts and constants next to
// the kernels).  A kernel has ONE home unit, the one that launches it: csdr_post.hip (CSDR_TU_POST: channelizers, DC blocker), csdr_bank.hip
// (CSDR_TU_BANK: front-ends, modems, audio), csdr_spec.hip (CSDR_TU_SPEC: the spectrum chain), csdr_io.hip (scope, mixer, ingest).  Elsewhere a
// kernel that is not a template already is declared as one that is never instantiated: parsed, not compiled (round 4 built chan_analyze_fft and the
// demodulator kernels three times and the spectrum chain twice).
#define CSDR_KERNEL static __global__
#define CSDR_KERNEL_ELSEWHERE template <typename CsdrNotEmittedHere_ = void> static __global__
#if defined(CSDR_TU_POST)
#define CSDR_KERNEL_POST CSDR_KERNEL
#else
#define CSDR_KERNEL_POST CSDR_KERNEL_ELSEWHERE
#endif
#if defined(CSDR_TU_BANK)
#define CSDR_KERNEL_BANK CSDR_KERNEL
#else
#define CSDR_KERNEL_BANK CSDR_KERNEL_ELSEWHERE
#endif
#if defined(CSDR_TU_SPEC)
#define CSDR_KERNEL_SPEC CSDR_KERNEL
#else
#define CSDR_KERNEL_SPEC CSDR_KERNEL_ELSEWHERE
#endif

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

// measurement switches (A/B runs recorded under profiles/): compiled in only with -DCSDR_LAB (CSDR_BUILD_LAB=1 python -m cubicsdr_amd.build);
// the shipping library reads none of them
inline int lab_int(const char *name, int dflt) {
#ifdef CSDR_LAB
    if (const char *e = getenv(name)) return atoi(e);
#else
    (void)name;
#endif
    return dflt;
}

// roctx ranges around the stage calls of the C ABI (csdr_post_execute, csdr_bank_execute, csdr_spec_process, the collectives): named spans
// on the host timeline next to the kernel trace (rocprofv3 --marker-trace --kernel-trace).  Off unless CSDR_ROCTX=1 is set when the first
// context is created: the tracing library (rocprofiler-sdk's roctx, else roctracer's) is then loaded with dlopen; nothing is linked.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
};
Roctx &roctx();            // csdr_ctx.hip
struct RangeScope {
    bool on;
    explicit RangeScope(const char *name) : on(roctx().push != nullptr) { if (on) (void)roctx().push(name); }
    ~RangeScope() { if (on) (void)roctx().pop(); }
    RangeScope(const RangeScope &) = delete;
    RangeScope &operator=(const RangeScope &) = delete;
};

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

// one AudioThreadInput (a block) of the float -> 16-bit PCM conversion (kernels_io.hpp: pcm16_convert)
struct PcmJob { const float *src; int16_t *dst; int32_t n; int32_t pad; const float *peak; };

// ---- small device helpers shared by the kernel headers ----
__device__ inline float2 cmul(float2 a, float2 b) { return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)); }
__device__ inline float2 cfma(float2 v, float2 w, float2 acc) {   // acc + v * w
    acc.x = fmaf(v.x, w.x, acc.x); acc.x = fmaf(-v.y, w.y, acc.x);
    acc.y = fmaf(v.x, w.y, acc.y); acc.y = fmaf(v.y, w.x, acc.y);
    return acc;
}
// a value that is the same in every lane of the wave by construction: tell the compiler (scalar registers, scalar loads)
__device__ inline int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// LDS hand-off between lanes of ONE wave (no other wave involved): the wave executes its LDS operations in order, so
// only the compiler has to be kept from moving them across this point
__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// issue priority of this wave among the waves of its SIMD (s_setprio 0 .. 3; 0 is what a wave starts with)
__device__ __forceinline__ void wave_priority(const int p) {
#if defined(__AMDGCN__)
    if (p == 0) __builtin_amdgcn_s_setprio(0); else if (p == 1) __builtin_amdgcn_s_setprio(1); else if (p == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
#else
    (void)p;
#endif
}
// keep the compiler's scheduler from moving instructions across this point (software-pipelined loops: requests for the next
// iteration stay in front of the arithmetic of the current one instead of being sunk to their first use)
__device__ __forceinline__ void sched_fence() {
#if defined(__AMDGCN__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}
// workgroup barrier that orders LDS traffic only: global loads / stores already issued stay in flight across it (__syncthreads
// also waits for every outstanding global access of the wave -- vmcnt(0) -- which exposes the full store latency at each barrier)
__device__ __forceinline__ void lds_barrier() {
#if defined(__AMDGCN__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#else
    __syncthreads();
#endif
}
// make a value opaque to the optimiser at this point: address arithmetic derived from it is redone where it is used instead of
// being hoisted out of an enclosing loop and kept (or spilled) across it
__device__ __forceinline__ void opaque(int &v) {
#if defined(__AMDGCN__)
    asm volatile("" : "+v"(v));
#endif
}
// pin loaded values where they are: without it a load whose result is only stored under a guard is sunk into the guarded block, behind its
// own wait (one memory round trip per element instead of several loads in flight)
__device__ __forceinline__ void pin_loaded(float4 &v) {
#if defined(__AMDGCN__)
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
#endif
}
// a value whose computation (the loads feeding it) must not be removed although nothing reads it
__device__ __forceinline__ void keep_alive(float v) {
#if defined(__AMDGCN__)
    asm volatile("" :: "v"(v));
#else
    (void)v;
#endif
}
// a product (or sum) that must be rounded on its own: hipcc contracts a * b + c into one fused operation by default (and its
// __fmul_rn / __fadd_rn are plain operators), which differs from the reference's separately rounded multiply and add in the last bit
__device__ __forceinline__ float rounded(float v) {
#if defined(__AMDGCN__)
    asm volatile("" : "+v"(v));
#endif
    return v;
}
// value of the neighbouring lane (lane ^ 1): a DPP quad permute on the device, no LDS traffic
__device__ __forceinline__ float lane_xor1(float v) {
#if defined(__AMDGCN__)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
#else
    return __shfl_xor(v, 1, 64);
#endif
}
// true in every lane when the predicate holds in any lane of the wave
__device__ __forceinline__ bool wave_any(bool p) {
#if defined(__AMDGCN__)
    return __builtin_amdgcn_ballot_w64(p) != 0;
#else
    int v = p ? 1 : 0;
    for (int o = 1; o < 64; o <<= 1) v |= __shfl_xor(v, o, 64);
    return v != 0;
#endif
}
// maximum / minimum over the 64 lanes of the wave, valid in lane 63: six DPP steps on the vector ALU (quad permutes, row mirrors,
// row broadcasts), no LDS crossbar traffic (__shfl_down goes through ds_bpermute: an LDS-pipe round trip per step)
#if defined(__AMDGCN__)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_max_to_lane63(float v) {
    v = fmaxf(v, dpp_f<0xB1, 0xF>(v));      // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_f<0x4E, 0xF>(v));      // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_f<0x141, 0xF>(v));     // row_half_mirror
    v = fmaxf(v, dpp_f<0x140, 0xF>(v));     // row_mirror: every lane of a row holds the row's maximum
    v = fmaxf(v, dpp_f<0x142, 0xA>(v));     // row_bcast:15 into rows 1 and 3
    v = fmaxf(v, dpp_f<0x143, 0xC>(v));     // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ float wave_min_to_lane63(float v) {
    v = fminf(v, dpp_f<0xB1, 0xF>(v));
    v = fminf(v, dpp_f<0x4E, 0xF>(v));
    v = fminf(v, dpp_f<0x141, 0xF>(v));
    v = fminf(v, dpp_f<0x140, 0xF>(v));
    v = fminf(v, dpp_f<0x142, 0xA>(v));
    v = fminf(v, dpp_f<0x143, 0xC>(v));
    return v;
}
#else
__device__ __forceinline__ float wave_max_to_lane63(float v) {
    for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min_to_lane63(float v) {
    for (int o = 1; o < 64; o <<= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
#endif
// v_mfma_f32_16x16x4_f32: D = A B + C on one wave, 16 x 16 outputs, four terms; lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] and holds
// D[4 (l >> 4) + r][l & 15] in element r.  Exact f32: a k-ordered fmaf chain per output, bit for bit what the vector pipe computes.
#if defined(__AMDGCN__)
typedef float csdr_f32x4 __attribute__((ext_vector_type(4)));
#else
typedef float csdr_f32x4 __attribute__((vector_size(16)));
#endif
__device__ __forceinline__ csdr_f32x4 csdr_mfma16(float a, float b, csdr_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// 16 bytes that are only guaranteed 8-byte aligned (two adjacent complex samples at an odd sample offset)
struct __attribute__((aligned(8))) f4u { float x, y, z, w; };

// N + 1 LDS reads of 16 bytes: N consecutive ones from pe, one from po (both 16-byte aligned), issued back to back and
// waited for once.  Written as ds_read_b128 by hand: when a window is only partly used the compiler narrows float4 loads to
// ds_read2_b64 pairs, and lanes 16 bytes apart then hit every bank twice (measured: half of the front-end's LDS cycles).
template <int N> __device__ __forceinline__ void lds_read128(const float2 *pe, const float2 *po, float4 (&e)[N], float4 &o);
#if defined(__AMDGCN__)
typedef float csdr_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p; }
template <> __device__ __forceinline__ void lds_read128<4>(const float2 *pe, const float2 *po, float4 (&e)[4], float4 &o) {
    csdr_v4f a, b, c, d, q;
    asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %5 offset:32\n\tds_read_b128 %3, %5 offset:48\n\t"
                 "ds_read_b128 %4, %6\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(q) : "v"(lds_addr(pe)), "v"(lds_addr(po)) : "memory");
    e[0] = make_float4(a.x, a.y, a.z, a.w); e[1] = make_float4(b.x, b.y, b.z, b.w); e[2] = make_float4(c.x, c.y, c.z, c.w);
    e[3] = make_float4(d.x, d.y, d.z, d.w); o = make_float4(q.x, q.y, q.z, q.w);
}
template <> __device__ __forceinline__ void lds_read128<6>(const float2 *pe, const float2 *po, float4 (&e)[6], float4 &o) {
    csdr_v4f a, b, c, d, f, g, q;
    asm volatile("ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:16\n\tds_read_b128 %2, %7 offset:32\n\tds_read_b128 %3, %7 offset:48\n\t"
                 "ds_read_b128 %4, %7 offset:64\n\tds_read_b128 %5, %7 offset:80\n\tds_read_b128 %6, %8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(f), "=&v"(g), "=&v"(q) : "v"(lds_addr(pe)), "v"(lds_addr(po)) : "memory");
    e[0] = make_float4(a.x, a.y, a.z, a.w); e[1] = make_float4(b.x, b.y, b.z, b.w); e[2] = make_float4(c.x, c.y, c.z, c.w);
    e[3] = make_float4(d.x, d.y, d.z, d.w); e[4] = make_float4(f.x, f.y, f.z, f.w); e[5] = make_float4(g.x, g.y, g.z, g.w);
    o = make_float4(q.x, q.y, q.z, q.w);
}
#else
template <int N> __device__ __forceinline__ void lds_read128(const float2 *pe, const float2 *po, float4 (&e)[N], float4 &o) {
    for (int i = 0; i < N; ++i) e[i] = reinterpret_cast<const float4 *>(pe)[i];
    o = *reinterpret_cast<const float4 *>(po);
}
#endif

}  // namespace csdr

// kernel ids for the optional per-kernel HIP-event profile (csdr_ctx_profile_*)
// (one id per kernel that can appear in a batch: two template instances of a kernel launched alternately under ONE id would
// always be sampled at the same phase of the sampling period and the per-stage sums would miss the other instance)
enum CsdrKernelId {
    KID_CHAN_ANALYZE = 0, KID_DC_ENDS, KID_DC_APPLY, KID_ROWS_COPY,
    KID_FE_GENERIC, KID_FE_S3, KID_FE_S4, KID_FE_S5, KID_FE_S6, KID_FE_S56, KID_FE_INTERP,
    KID_MODEM, KID_GAIN_SCAN, KID_FMS, KID_AUDIO, KID_FMS_OUT, KID_MIX, KID_TABLES,
    KID_FFT_COLS, KID_FFT_ROWS, KID_SPEC_AVG, KID_SPEC_TRACK, KID_SPEC_DISPLAY, KID_SPEC_MISC,
    KID_COUNT
};

// One HIP stream per pipeline stage, like one IOThread per stage in the reference: consecutive batches overlap across
// stages (SDRPostThread works on block i+1 while the demodulators work on block i); events order the hand-offs.
enum CsdrLane { LANE_POST = 0, LANE_FE, LANE_AUDIO, LANE_FFT, LANE_AVG, LANE_COUNT };

struct csdr_ctx {
    int device = 0;
    int n_cu = 256;                          // compute units (grid sizing: whole rounds of resident workgroups)
    int lds_per_cu = 160 * 1024;             // LDS of one compute unit (gfx950)
    hipStream_t stream = nullptr;            // boundary stream: the caller's producer / consumer work is ordered on it
    bool own_stream = false;
    bool boundary_shared = false;            // a csdr_comm enqueues collectives on the boundary stream: the lanes order against it even when the library created it
    hipStream_t lanes[LANE_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // logical stage -> physical stream
    hipStream_t phys[LANE_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr};    // streams this ctx created
    int n_phys = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr; // timing
    hipEvent_t ev_in[LANE_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // fork: boundary stream -> lane (one per lane: stage threads fork concurrently)
    std::mutex prof_mu;                      // the profile bookkeeping is shared by the stage threads
    hipEvent_t ev_lane[LANE_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // join: lane -> boundary stream
    csdr::DevBuf<float> sintab;  // 1024-entry sine table of the reference's NCO
    // per-kernel profile: event pairs recorded around launches while enabled
    bool prof_on = false;
    int prof_period = 1;                     // bracket every prof_period-th launch of each kernel
    unsigned prof_seen[KID_COUNT] = {0};
    struct ProfRec { int id; hipEvent_t a, b; };
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[KID_COUNT] = {0};
    double prof_min[KID_COUNT] = {0}, prof_max[KID_COUNT] = {0};    // shortest / longest bracketed launch
    long long prof_n[KID_COUNT] = {0};
    hipEvent_t prof_event() {                // (callers hold prof_mu)
        if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    // work the caller enqueued on the boundary stream (e.g. the producer of a device IQ buffer) precedes this lane's next work
    int lane_begin(int lane) {
        if (own_stream && !boundary_shared) return CSDR_OK;      // nobody else can enqueue on a stream we created
        CSDR_HIP_TRY(hipEventRecord(ev_in[lane], stream));
        CSDR_HIP_TRY(hipStreamWaitEvent(lanes[lane], ev_in[lane], 0));
        return CSDR_OK;
    }
    // everything enqueued on the lanes so far precedes whatever is enqueued on the boundary stream next
    int join() {
        for (int l = 0; l < n_phys; ++l) {
            if (phys[l] == stream) continue;
            CSDR_HIP_TRY(hipEventRecord(ev_lane[l], phys[l]));
            CSDR_HIP_TRY(hipStreamWaitEvent(stream, ev_lane[l], 0));
        }
        return CSDR_OK;
    }
    // hand-off between two stages: only a real cross-stream edge needs an event (on this stack one costs far more than
    // a kernel boundary), stages that share a physical stream are ordered by it
    bool same(int lane_a, int lane_b) const { return lanes[lane_a] == lanes[lane_b]; }
    int signal(hipEvent_t ev, int from_lane, int to_lane) {
        if (same(from_lane, to_lane)) return CSDR_OK;
        CSDR_HIP_TRY(hipEventRecord(ev, lanes[from_lane]));
        return CSDR_OK;
    }
    int wait(hipEvent_t ev, int from_lane, int to_lane) {
        if (same(from_lane, to_lane)) return CSDR_OK;
        CSDR_HIP_TRY(hipStreamWaitEvent(lanes[to_lane], ev, 0));
        return CSDR_OK;
    }
    // workgroups of this kernel that are resident at once on the whole chip
    // (the occupancy query is a runtime call of several microseconds and its answer never changes: asked once per (kernel, shape); a one-block call
    // has ~100 us of host time in all)
    struct SlotsKey { const void *k; int threads; size_t lds; int nb; };
    mutable std::vector<SlotsKey> slots_cache;
    mutable std::mutex slots_mu;
    template <typename K>
    int wg_slots(K kernel, int threads, size_t lds) const {
        std::lock_guard<std::mutex> lk(slots_mu);
        for (const SlotsKey &e : slots_cache) if (e.k == (const void *)kernel && e.threads == threads && e.lds == lds) return e.nb * n_cu;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kernel, threads, lds) != hipSuccess || nb < 1) nb = 1;
        slots_cache.push_back(SlotsKey{(const void *)kernel, threads, lds, nb});
        return nb * n_cu;
    }
    int sync_all() {
        for (int l = 0; l < n_phys; ++l) CSDR_HIP_TRY(hipStreamSynchronize(phys[l]));
        CSDR_HIP_TRY(hipStreamSynchronize(stream));
        return CSDR_OK;
    }
};

// Every entry point of the C ABI that touches HIP runs on the device of its context, whichever host thread calls it (the
// reference's pipeline calls from one IOThread per stage) and whatever device that thread had selected before (a caller
// that also drives torch or another HIP library may change it between two calls).  The thread's current device is queried,
// changed only on a mismatch, and put back when the entry point returns.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(const csdr_ctx *c) {
        if (!c) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
    }
    ~DeviceScope() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// bracket one kernel launch with events when profiling is on
struct ProfScope {
    csdr_ctx *c; int id; hipStream_t st; hipEvent_t a = nullptr;
    ProfScope(csdr_ctx *c_, int id_, hipStream_t st_) : c(c_), id(id_), st(st_) {
        if (!c->prof_on) return;
        std::lock_guard<std::mutex> lk(c->prof_mu);
        if ((c->prof_seen[id]++ % (unsigned)c->prof_period) == 0) { a = c->prof_event(); (void)hipEventRecord(a, st); }
    }
    ~ProfScope() {
        if (!a) return;
        std::lock_guard<std::mutex> lk(c->prof_mu);
        hipEvent_t b = c->prof_event(); (void)hipEventRecord(b, st); c->prof_pending.push_back({id, a, b});
    }
};
#define CSDR_LAUNCH(ctx_, lane_, kid_, kern_, grid_, block_, lds_, ...) \
    do { ProfScope ps__((ctx_), (kid_), (ctx_)->lanes[lane_]); hipLaunchKernelGGL(kern_, grid_, block_, lds_, (ctx_)->lanes[lane_], __VA_ARGS__); } while (0)

// streaming-hint stores / loads for data that is written or read once per launch and far exceeds the caches: the spectrum chain's outputs
// (radix intermediate, magnitudes, pair sums, display values) and the radix pass's input.  Measured on C3 (-DCSDR_NT=0 compiles them as
// plain accesses): radix 0.357 -> 0.347 ms, rows 0.286 -> 0.270 ms, average 0.186 -> 0.178 ms.  The same hint on the loads of the row
// pass (the intermediate) costs it 10 %, on the averaging / display loads it changes nothing: those stay plain.
#ifndef CSDR_NT
#define CSDR_NT 1
#endif
namespace csdr {
#if defined(__AMDGCN__) && CSDR_NT
typedef float csdr_nt2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_stream(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_stream(float2 *p, float2 v) { const csdr_nt2 t = {v.x, v.y}; __builtin_nontemporal_store(t, reinterpret_cast<csdr_nt2 *>(p)); }
__device__ __forceinline__ float2 ld_stream(const float2 *p) { const csdr_nt2 t = __builtin_nontemporal_load(reinterpret_cast<const csdr_nt2 *>(p)); return make_float2(t.x, t.y); }
__device__ __forceinline__ float ld_stream(const float *p) { return __builtin_nontemporal_load(p); }
#else
__device__ __forceinline__ void st_stream(float *p, float v) { *p = v; }
__device__ __forceinline__ void st_stream(float2 *p, float2 v) { *p = v; }
__device__ __forceinline__ float2 ld_stream(const float2 *p) { return *p; }
__device__ __forceinline__ float ld_stream(const float *p) { return *p; }
#endif
}  // namespace csdr
