// micro-benchmark: how fast can persistent 512-thread workgroups (two per CU, 70 KB of LDS each) stream tiles of 69 KB into LDS?
// (the load skeleton of chan_analyze_p2)   build: hipcc --offload-arch=gfx950 -O3 stream_read.hip -o stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int kThreads = 512, kPre = 9, kTileF4 = kThreads * kPre;      // float4 per tile (73.7 KB)

template <int MODE>   // 0: request, wait, commit   1: two register sets (the next tile is requested before this one is committed)   2: request split in thirds, commit per third
__global__ __launch_bounds__(kThreads, 4) void skeleton(const float4 *__restrict__ x, long n_tiles, float *sink) {
    extern __shared__ float4 lds[];
    const int tid = threadIdx.x;
    float4 a[kPre], b[kPre];
    float acc = 0.f;
    long tile = blockIdx.x;
    if (tile < n_tiles) {
#pragma unroll
        for (int i = 0; i < kPre; ++i) a[i] = x[tile * kTileF4 + tid + i * kThreads];
    }
    for (; tile < n_tiles; tile += gridDim.x) {
        const long nxt = tile + gridDim.x;
        if (MODE == 1) {
            if (nxt < n_tiles) {
#pragma unroll
                for (int i = 0; i < kPre; ++i) b[i] = x[nxt * kTileF4 + tid + i * kThreads];
            }
        }
#pragma unroll
        for (int i = 0; i < kPre; ++i) lds[tid + i * kThreads] = a[i];
        __syncthreads();
        acc += lds[(tid * 7 + 3) % kTileF4].x;
        __syncthreads();
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < kPre; ++i) a[i] = b[i];
        } else {
            if (nxt < n_tiles) {
#pragma unroll
                for (int i = 0; i < kPre; ++i) a[i] = x[nxt * kTileF4 + tid + i * kThreads];
            }
            if (MODE == 2) {      // some independent work between the request and the next commit (stands in for the DFT phase)
                float v = acc;
                for (int k = 0; k < 2000; ++k) v = fmaf(v, 1.0001f, 0.5f);
                acc = v;
            }
        }
    }
    if (acc == 12345.f) sink[0] = acc;
}
__global__ void plain_read(const float4 *__restrict__ x, long n, float *sink) {
    float acc = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { const float4 v = x[i]; acc += v.x + v.w; }
    if (acc == 12345.f) sink[0] = acc;
}
int main() {
    const long n_tiles = 14000;                       // ~1.03 GB
    const long n = n_tiles * kTileF4;
    float4 *x; float *sink;
    CK(hipMalloc(&x, n * sizeof(float4))); CK(hipMalloc(&sink, 4));
    CK(hipMemset(x, 0, n * sizeof(float4)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = kTileF4 * sizeof(float4);
    CK(hipFuncSetAttribute((const void *)skeleton<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)skeleton<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)skeleton<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto time = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-40s %.3f ms  %.2f TB/s\n", name, ms, n * 16.0 / ms / 1e9);
    };
    for (int wgs : {256, 512}) {
        char nm[64];
        snprintf(nm, 64, "skeleton<0> wgs=%d", wgs); time(nm, [&] { skeleton<0><<<wgs, kThreads, lds>>>(x, n_tiles, sink); });
        snprintf(nm, 64, "skeleton<1> two sets wgs=%d", wgs); time(nm, [&] { skeleton<1><<<wgs, kThreads, lds>>>(x, n_tiles, sink); });
        snprintf(nm, 64, "skeleton<2> +work wgs=%d", wgs); time(nm, [&] { skeleton<2><<<wgs, kThreads, lds>>>(x, n_tiles, sink); });
    }
    time("plain_read 256x8 wgs x 256", [&] { plain_read<<<2048, 256>>>(x, n, sink); });
    time("plain_read 256x16 wgs x 256", [&] { plain_read<<<4096, 256>>>(x, n, sink); });
    time("plain_read 256x32 wgs x 256", [&] { plain_read<<<8192, 256>>>(x, n, sink); });
    return 0;
}
