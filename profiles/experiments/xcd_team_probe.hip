// xcd_team_probe.hip -- measurement helper (not product code): can the workgroups that share ONE L2 (one XCD) hand a 1 MB array to each other through
// that L2, without the agent-scope write-back, and does the array stay out of HBM?  Gate for a two-pass spectrum transform whose intermediate rows
// never leave the XCD.
//   hipcc --offload-arch=gfx950 -O3 -o xcd_team_probe xcd_team_probe.hip && ./xcd_team_probe [iters] [mode]
//   mode 0: hand-off = s_waitcnt vmcnt(0) + L2 atomic (no sc1) + buffer_inv sc1 on the reader     (the protocol under test)
//   mode 1: hand-off = agent-scope release / acquire (buffer_wbl2 sc1 ...)                         (the documented protocol, for comparison)
//   mode 2: no team: every workgroup writes and reads its own slice (the ceiling)
//   mode 3: as mode 0 with the arrival counter at agent scope (relaxed atomics, no fences): only the DATA goes the short way
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kThreads = 256, kElems = 1 << 17;          // 2^17 float2 = 1 MB per team

struct Team { int count; int arrive; int pad[30]; };    // one 128-byte line each

__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}
__device__ __forceinline__ int l2_add(int *p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// all workgroups of the team have arrived `target` times
__device__ int g_timeout;
template <int MODE>
__device__ __forceinline__ void team_barrier(Team *t, int target) {
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        int polls = 0;
        if (MODE == 1 || MODE == 3) {
            __hip_atomic_fetch_add(&t->arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(&t->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++polls < (1 << 20)) __builtin_amdgcn_s_sleep(1);
        } else {
            l2_add(&t->arrive, 1);
            while (l2_add(&t->arrive, 0) < target && ++polls < (1 << 20)) __builtin_amdgcn_s_sleep(1);
        }
        if (polls >= (1 << 20)) atomicAdd(&g_timeout, 1);
    }
    __syncthreads();
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else asm volatile("buffer_inv sc1" ::: "memory");
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void probe(Team *teams, int *grid_arrive, float2 *bufs, int iters, int *xcc_of_wg, unsigned long long *bad, float *sink) {
    __shared__ int s_team, s_rank, s_n;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int x = xcc_id();
        xcc_of_wg[blockIdx.x] = x;
        s_team = x;
        s_rank = __hip_atomic_fetch_add(&teams[x].count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(grid_arrive, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(grid_arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x) __builtin_amdgcn_s_sleep(2);
        s_n = __hip_atomic_load(&teams[x].count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int team = s_team, rank = s_rank, n = s_n;
    Team *t = teams + team;
    float2 *buf = bufs + (size_t)team * kElems;
    unsigned long long nbad = 0;
    float acc = 0.f;
    const int per = kElems / n;                          // (n divides 2^17 when it is a power of two; the rest of the array is left alone otherwise)
    for (int it = 0; it < iters; ++it) {
        // phase A: this workgroup's contiguous slice
        for (int i = tid; i < per; i += kThreads) {
            const int e = rank * per + i;
            buf[e] = make_float2((float)(e + it), (float)(e ^ it));
        }
        if (MODE != 2) team_barrier<MODE>(t, n * (2 * it + 1));
        // phase B: a strided slice -- elements every other workgroup of the team wrote
        for (int i = tid; i < per; i += kThreads) {
            const int e = MODE == 2 ? rank * per + i : i * n + rank;
            const float2 v = buf[e];
            if (v.x != (float)(e + it) || v.y != (float)(e ^ it)) ++nbad;
            acc += v.x;
        }
        if (MODE != 2) team_barrier<MODE>(t, n * (2 * it + 2));
    }
    if (nbad) atomicAdd(bad, nbad);
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200, mode = argc > 2 ? atoi(argv[2]) : 0, wgs = argc > 3 ? atoi(argv[3]) : 256;
    Team *teams; int *ga, *xw; float2 *bufs; unsigned long long *bad; float *sink;
    CK(hipMalloc(&teams, 16 * sizeof(Team))); CK(hipMalloc(&ga, 4)); CK(hipMalloc(&xw, wgs * 4)); CK(hipMalloc(&bufs, (size_t)16 * kElems * sizeof(float2)));
    CK(hipMalloc(&bad, 8)); CK(hipMalloc(&sink, 4));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(teams, 0, 16 * sizeof(Team))); CK(hipMemset(ga, 0, 4)); CK(hipMemset(bad, 0, 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(wgs), dim3(kThreads), 0, 0, teams, ga, bufs, iters, xw, bad, sink);
        else if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(kThreads), 0, 0, teams, ga, bufs, iters, xw, bad, sink);
        else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(wgs), dim3(kThreads), 0, 0, teams, ga, bufs, iters, xw, bad, sink);
        else hipLaunchKernelGGL(probe<2>, dim3(wgs), dim3(kThreads), 0, 0, teams, ga, bufs, iters, xw, bad, sink);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<int> x(wgs); unsigned long long b; std::vector<Team> tm(16);
        CK(hipMemcpy(x.data(), xw, wgs * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(tm.data(), teams, 16 * sizeof(Team), hipMemcpyDeviceToHost));
        int to = 0; CK(hipMemcpyFromSymbol(&to, HIP_SYMBOL(g_timeout), 4));
        if (to) printf("  (barrier timeouts so far: %d; arrive counters:", to);
        if (to) { for (int i = 0; i < 8; ++i) printf(" %d", tm[i].arrive); printf(")\n"); }
        int hist[16] = {0}; int rr = 0;
        for (int i = 0; i < wgs; ++i) { hist[x[i] & 15]++; rr += (x[i] == (i & 7)); }
        printf("mode %d wgs %d iters %d: %.3f ms = %.2f us per iteration (1 MB written + 1 MB read per team), mismatches %llu; teams:", mode, wgs, iters, ms, 1e3 * ms / iters, b);
        for (int i = 0; i < 16; ++i) if (hist[i]) printf(" xcc%d=%d", i, hist[i]);
        printf("; blockIdx %% 8 == xcc for %d of %d\n", rr, wgs);
    }
    return 0;
}
