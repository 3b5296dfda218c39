// tests/emu/hip_emu_runtime.cpp -- TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h in this directory).
#include <hip/hip_runtime.h>

#include <mutex>

namespace csdr {
// the one dynamic-LDS region every kernel declares as `extern __shared__ ... char smem[]` (160 KiB per CU on gfx950)
alignas(256) char smem[160 * 1024];
}  // namespace csdr

namespace hip_emu {
thread_local dim3 t_threadIdx, t_blockIdx;
State &state() { static State s; return s; }
static std::mutex g_launch_mutex;

void run(dim3 grid, dim3 block, size_t lds_bytes, void (*thunk)(void *), void *arg) {
    std::lock_guard<std::mutex> lk(g_launch_mutex);
    if (lds_bytes > sizeof(csdr::smem)) { std::fprintf(stderr, "[hip_emu] dynamic LDS request %zu > 160 KiB\n", lds_bytes); std::abort(); }
    if (block.y != 1 || block.z != 1 || block.x > 1024 || block.x == 0) { std::fprintf(stderr, "[hip_emu] unsupported block shape\n"); std::abort(); }
    State &s = state();
    s.grid = grid; s.block = block;
    const unsigned n = block.x;
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    std::barrier<> end_bar((std::ptrdiff_t)n);
    s.bar.reset(new std::barrier<>((std::ptrdiff_t)n));
    auto reset_wave_barriers = [&]() {
        for (unsigned w = 0; w * 64 < n; ++w) s.wbar[w].reset(new std::barrier<>((std::ptrdiff_t)std::min(64u, n - w * 64)));
    };
    reset_wave_barriers();
    // canary behind the requested dynamic-LDS size: a kernel that carves more than it asked for is a bug on the device
    const size_t canary = std::min<size_t>(8192, sizeof(csdr::smem) - lds_bytes);
    std::memset(csdr::smem + lds_bytes, 0xEE, canary);
    auto worker = [&](unsigned tid) {
        t_threadIdx = dim3(tid, 0, 0);
        for (unsigned bz = 0; bz < grid.z; ++bz)
            for (unsigned by = 0; by < grid.y; ++by)
                for (unsigned bx = 0; bx < grid.x; ++bx) {
                    t_blockIdx = dim3(bx, by, bz);
                    thunk(arg);
                    s.bar->arrive_and_drop();       // an exited work-item no longer takes part in __syncthreads
                    s.wbar[tid >> 6]->arrive_and_drop();
                    end_bar.arrive_and_wait();
                    if (tid == 0) {
                        for (size_t i = 0; i < canary; ++i)
                            if ((unsigned char)csdr::smem[lds_bytes + i] != 0xEE) {
                                std::fprintf(stderr, "[hip_emu] workgroup (%u,%u,%u) wrote past its %zu bytes of dynamic LDS (offset +%zu)\n", bx, by, bz, lds_bytes, i);
                                std::abort();
                            }
                        // poison LDS between workgroups: nothing may rely on another workgroup's leftovers
                        std::memset(csdr::smem, 0xCD, lds_bytes);
                        s.bar.reset(new std::barrier<>((std::ptrdiff_t)n));
                        reset_wave_barriers();
                    }
                    end_bar.arrive_and_wait();
                }
    };
    std::vector<std::thread> th;
    th.reserve(n);
    for (unsigned t = 0; t < n; ++t) th.emplace_back(worker, t);
    for (auto &t : th) t.join();
}
}  // namespace hip_emu

const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipError(emu)"; }
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) {
    // exact-size allocation so AddressSanitizer sees every out-of-bounds device access; 0xA5 fill = "uninitialised HBM"
    *p = std::malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    std::memset(*p, 0xA5, n);
    return hipSuccess;
}
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)std::malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 8; return hipSuccess; }      // a small "chip": exercises multi-round grids
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *nb, const void *, int, size_t) { *nb = 2; return hipSuccess; }
