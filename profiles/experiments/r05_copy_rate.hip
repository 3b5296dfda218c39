// What a plain streaming kernel reaches on this box (measurement helper, not the product): float4 copy / read / write of 1 GiB arrays,
// grid-stride and one-tile-per-workgroup forms, with and without the streaming (nt) hint.   hipcc --offload-arch=gfx950 -O3 r05_copy_rate.hip -o /tmp/copy_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NTL, bool NTS, int U>
__global__ __launch_bounds__(256) void copy_gs(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(a + i + u * 256) : a[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NTS) __builtin_nontemporal_store(v[u], b + i + u * 256); else b[i + u * 256] = v[u]; }
    }
}
template <int U>
__global__ __launch_bounds__(256) void read_gs(const f4 *__restrict__ a, float *__restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += a[i + u * 256];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}
template <int U>
__global__ __launch_bounds__(256) void write_gs(f4 *__restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    const f4 v = {1, 2, 3, 4};
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) b[i + u * 256] = v;
    }
}
template <typename F> static float timed(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    f4 *a, *b; float *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    const int reps = 20;
    for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64}) {
        float t;
        t = timed([&] { copy_gs<false, false, 4><<<grid, 256>>>(a, b, n); }, reps); printf("copy     U4 grid %6d: %.3f ms  %.2f TB/s\n", grid, t, 2.0 * bytes / t / 1e9);
        t = timed([&] { copy_gs<false, true, 4><<<grid, 256>>>(a, b, n); }, reps);  printf("copy nts U4 grid %6d: %.3f ms  %.2f TB/s\n", grid, t, 2.0 * bytes / t / 1e9);
        t = timed([&] { copy_gs<true, true, 4><<<grid, 256>>>(a, b, n); }, reps);   printf("copy ntb U4 grid %6d: %.3f ms  %.2f TB/s\n", grid, t, 2.0 * bytes / t / 1e9);
        t = timed([&] { copy_gs<false, true, 8><<<grid, 256>>>(a, b, n); }, reps);  printf("copy nts U8 grid %6d: %.3f ms  %.2f TB/s\n", grid, t, 2.0 * bytes / t / 1e9);
        t = timed([&] { read_gs<4><<<grid, 256>>>(a, o, n); }, reps);               printf("read     U4 grid %6d: %.3f ms  %.2f TB/s\n", grid, t, 1.0 * bytes / t / 1e9);
        t = timed([&] { write_gs<4><<<grid, 256>>>(b, n); }, reps);                 printf("write    U4 grid %6d: %.3f ms  %.2f TB/s\n", grid, t, 1.0 * bytes / t / 1e9);
    }
    {   // one launch per element tile (no loop): n / (256 * 4) workgroups
        const int grid = (int)(n / (256 * 4));
        float t = timed([&] { copy_gs<false, true, 4><<<grid, 256>>>(a, b, n); }, reps); printf("copy nts U4 one tile per workgroup (%d): %.3f ms  %.2f TB/s\n", grid, t, 2.0 * bytes / t / 1e9);
        t = timed([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, reps); printf("hipMemcpyDtoD: %.3f ms  %.2f TB/s\n", t, 2.0 * bytes / t / 1e9);
    }
    return 0;
}
