// The access pattern of the spectrum's column pass without its arithmetic (measurement helper, not the product): frames of 512 rows x 256 complex,
// a 1024-thread workgroup copies one COLUMN TILE of a frame -- 512 pieces of W x 8 bytes at 2 KB stride -- rows permuted on the way out as the
// transform does.  W = 16 (128-byte pieces: the shipped pass), 32, 64; tiles of a frame on one XCD or spread; against whole contiguous frames.
//   hipcc --offload-arch=gfx950 -O3 r05_pattern_rate.hip -o /tmp/pattern_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int R = 256, C = 512, N = R * C;
// W columns per workgroup; thread (g = tid / W, col = tid % W); rows g + (1024 / W) k
template <int W, int MAP>
__global__ __launch_bounds__(1024) void tile_copy(const f2 *__restrict__ x, f2 *__restrict__ z, int nf) {
    constexpr int NT = R / W, G = 1024 / W, K = C / G;
    int b = blockIdx.x;
    if (MAP == 1) {               // the NT tiles of a frame group on ONE XCD: consecutive ids of an XCD (b % 8 fixed) walk the tiles
        const int xcd = b & 7, s = b >> 3;                       // s-th workgroup of this XCD
        b = ((s / NT) * 8 + xcd) * NT + (s % NT);                // frame group (s / NT) * 8 + xcd, tile s % NT
    }
    const int ct = b % NT, fg = b / NT, nfg = gridDim.x / NT;
    const int tid = threadIdx.x, col = tid % W, g = tid / W;
    const int n2 = ct * W + col;
    f2 nx[K];
    auto request = [&](int f) {
        const f2 *xb = x + (size_t)f * N;
#pragma unroll
        for (int k = 0; k < K; ++k) nx[k] = xb[(size_t)(G * k + g) * R + n2];
    };
    if (fg < nf) request(fg);
    for (int f = fg; f < nf; f += nfg) {
        f2 a[K];
#pragma unroll
        for (int k = 0; k < K; ++k) a[k] = nx[k] * 1.0001f;
        if (f + nfg < nf) request(f + nfg);
        f2 *zb = z + (size_t)f * N;
        const int orow = (g % 8) * (G / 8) + g / 8;              // a permutation of the G rows of a slab
#pragma unroll
        for (int k = 0; k < K; ++k) __builtin_nontemporal_store(a[k], zb + (size_t)(G * k + orow) * R + n2);
    }
}
template <typename F> static float timed(F f, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main() {
    const int nf = 1000;
    const size_t bytes = (size_t)nf * N * 8;
    f2 *a, *b;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes);
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 0, bytes);
    const int reps = 10;
    auto rep = [&](const char *name, float t) { printf("%-52s %.3f ms  %.2f TB/s\n", name, t, 2.0 * bytes / t / 1e9); };
    for (int wgs : {256, 512}) {
        printf("-- %d workgroups\n", wgs);
        rep("W=16 (128 B pieces), tiles of a frame spread", timed([&] { tile_copy<16, 0><<<wgs, 1024>>>(a, b, nf); }, reps));
        rep("W=16, tiles of a frame on one XCD", timed([&] { tile_copy<16, 1><<<wgs, 1024>>>(a, b, nf); }, reps));
        rep("W=32 (256 B pieces), spread", timed([&] { tile_copy<32, 0><<<wgs, 1024>>>(a, b, nf); }, reps));
        rep("W=32, one XCD", timed([&] { tile_copy<32, 1><<<wgs, 1024>>>(a, b, nf); }, reps));
        rep("W=64 (512 B pieces), spread", timed([&] { tile_copy<64, 0><<<wgs, 1024>>>(a, b, nf); }, reps));
        rep("W=64, one XCD", timed([&] { tile_copy<64, 1><<<wgs, 1024>>>(a, b, nf); }, reps));
        rep("W=256 (whole rows: contiguous 64 KB per step)", timed([&] { tile_copy<256, 0><<<wgs, 1024>>>(a, b, nf); }, reps));
    }
    return 0;
}
