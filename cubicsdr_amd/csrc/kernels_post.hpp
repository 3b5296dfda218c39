// kernels_post.hpp -- SDRPostThread arithmetic on the GPU (K1 DC blocker, K2 polyphase channelizer, K4 de-interleave).
//
// Replaces (reference file:line): iirfilt_crcf_execute_block SDRPostThread.cpp:284,375 (DC blocker, liquid
// iirfilt_crcf_create_dc_blocker(0.0005) :29), firpfbch_crcf_analyzer_execute :449-451 (liquid firpfbch, Kaiser
// prototype m=4, As=60 :406) and the strided channel gather :364-381.
#pragma once
#include "common.hpp"

namespace csdr {

// ------------------------------------------------------------------------------------------------------------
// K1: first-order DC blocker  v[n] = x[n] - a1 v[n-1];  y[n] = v[n] - v[n-1]   (direct form II, b={1,-1}, a={1,a1})
// A linear recurrence: evaluated as a three-kernel blocked affine scan in fp64 (tile-local ends, cross-tile carry,
// apply), so a stream of any length runs in parallel while the carried state (one complex v) stays exact.
// Tile = 256 threads x 16 samples, staged through LDS so global accesses stay coalesced.
// ------------------------------------------------------------------------------------------------------------
constexpr int kDcSeg = 16;
constexpr int kDcThreads = 256;
constexpr int kDcTile = kDcSeg * kDcThreads;

struct d2 { double x, y; };

__device__ inline double dc_pow(double c, int n) {
    double r = 1.0, b = c;
    while (n) { if (n & 1) r *= b; b *= b; n >>= 1; }
    return r;
}

// scan of the recurrence V[t+1] = A V[t] + b[t] over 256 threads; returns V[t] (value ENTERING thread t's segment)
// given V_in entering thread 0; *total receives V[256].
__device__ inline d2 dc_block_scan(d2 b, double A, d2 v_in, d2 *lds_b, double *lds_a, d2 *total) {
    const int t = threadIdx.x;
    // inclusive Hillis-Steele on affine maps (a, b): compose(f_later o f_earlier)
    double a = A;
    lds_a[t] = a; lds_b[t] = b;
    __syncthreads();
    for (int off = 1; off < kDcThreads; off <<= 1) {
        double pa = 1.0; d2 pb = {0.0, 0.0};
        if (t >= off) { pa = lds_a[t - off]; pb = lds_b[t - off]; }
        __syncthreads();
        if (t >= off) { b.x = a * pb.x + b.x; b.y = a * pb.y + b.y; a = a * pa; }
        lds_a[t] = a; lds_b[t] = b;
        __syncthreads();
    }
    // inclusive result for thread t: V[t+1] = a * v_in + b
    d2 incl = {a * v_in.x + b.x, a * v_in.y + b.y};
    lds_b[t] = incl;
    __syncthreads();
    d2 ent = (t == 0) ? v_in : lds_b[t - 1];
    if (total) *total = lds_b[kDcThreads - 1];
    __syncthreads();
    return ent;
}

// pass 1: each tile computes its end value assuming zero entering state
__global__ __launch_bounds__(kDcThreads) void dc_tile_ends(const float2 *__restrict__ x, int64_t n, double c, d2 *tile_end) {
    __shared__ float2 sx[kDcTile];
    __shared__ d2 sb[kDcThreads];
    __shared__ double sa[kDcThreads];
    const int64_t base = (int64_t)blockIdx.x * kDcTile;
    for (int i = threadIdx.x; i < kDcTile; i += kDcThreads) {
        int64_t g = base + i;
        sx[i] = g < n ? x[g] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    d2 v = {0.0, 0.0};
    int cnt = 0;
    for (int i = 0; i < kDcSeg; ++i) {
        int64_t g = base + threadIdx.x * kDcSeg + i;
        if (g < n) { float2 s = sx[threadIdx.x * kDcSeg + i]; v.x = (double)s.x + c * v.x; v.y = (double)s.y + c * v.y; ++cnt; }
    }
    // threads past the end of the stream act as identity maps (A = 1, b = 0): handled by using A^cnt
    // -> the scan below assumes a uniform A, so give short segments their own multiplier through b only when cnt == kDcSeg;
    // partial segments occur only in the last tile, whose end value is never consumed by a later tile except as the
    // final carried state, which dc_apply recomputes exactly.  So a uniform A is sufficient here.
    d2 total;
    (void)dc_block_scan(v, dc_pow(c, kDcSeg), d2{0.0, 0.0}, sb, sa, &total);
    if (threadIdx.x == 0) tile_end[blockIdx.x] = total;
}

// pass 2: sequential carry across tiles (a few hundred tiles at most); state[0] = v entering the stream
__global__ void dc_tile_carry(const d2 *tile_end, int ntiles, double c, d2 *state, d2 *tile_in) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double A = dc_pow(c, kDcTile);
    d2 v = state[0];
    for (int t = 0; t < ntiles; ++t) {
        tile_in[t] = v;
        d2 e = tile_end[t];
        v.x = A * v.x + e.x; v.y = A * v.y + e.y;
    }
}

// pass 3: recompute with the true entering state and write y (in place allowed); the last tile stores the new state
__global__ __launch_bounds__(kDcThreads) void dc_apply(const float2 *x, float2 *y, int64_t n, double c, const d2 *tile_in, d2 *state) {
    __shared__ float2 sx[kDcTile];
    __shared__ d2 sb[kDcThreads];
    __shared__ double sa[kDcThreads];
    const int64_t base = (int64_t)blockIdx.x * kDcTile;
    for (int i = threadIdx.x; i < kDcTile; i += kDcThreads) {
        int64_t g = base + i;
        sx[i] = g < n ? x[g] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    d2 v = {0.0, 0.0};
    for (int i = 0; i < kDcSeg; ++i) {
        float2 s = sx[threadIdx.x * kDcSeg + i];
        v.x = (double)s.x + c * v.x; v.y = (double)s.y + c * v.y;
    }
    d2 ent = dc_block_scan(v, dc_pow(c, kDcSeg), tile_in[blockIdx.x], sb, sa, nullptr);
    v = ent;
    for (int i = 0; i < kDcSeg; ++i) {
        int64_t g = base + threadIdx.x * kDcSeg + i;
        float2 s = sx[threadIdx.x * kDcSeg + i];
        d2 v0 = {(double)s.x + c * v.x, (double)s.y + c * v.y};
        sx[threadIdx.x * kDcSeg + i] = make_float2((float)(v0.x - v.x), (float)(v0.y - v.y));
        if (g < n) {
            v = v0;
            if (g == n - 1) state[0] = v0;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kDcTile; i += kDcThreads) {
        int64_t g = base + i;
        if (g < n) y[g] = sx[i];
    }
}

// ------------------------------------------------------------------------------------------------------------
// K2 + K4: critically-sampled polyphase analysis bank, M channels, 8 taps per branch, channel-major output.
//   X_t[c] = sum_{n<8} taps[c][n] x[(t-n) M + c];   y_t[k] = sum_c X_t[c] exp(-j 2 pi k c / M);   out[k][t]
// A workgroup owns TF consecutive frames: it stages (TF+7) M input samples in LDS with coalesced loads, forms X
// (8 real x complex MACs per sample), then evaluates the M-point DFT for the REQUESTED channels only (the reference
// computes all M with an FFT and then copies just the channels that have demodulators, SDRPostThread.cpp:336-339).
// Lanes run along t so the channel-major stores are coalesced; the twiddle index is wave-uniform.
// M is arbitrary (4, 20, 122, 200 ...): direct DFT, cost ~ n_active * M per frame.
// ------------------------------------------------------------------------------------------------------------
constexpr int kChanThreads = 256;
constexpr int kChanTaps = 8;

__global__ __launch_bounds__(kChanThreads) void chan_analyze(
    const float2 *__restrict__ x,        // batch input, n_frames * M samples
    const float2 *__restrict__ hist,     // 7 * M samples preceding x
    const float *__restrict__ taps,      // [M][8]
    const float2 *__restrict__ tw,       // [M] exp(-j 2 pi i / M)
    const int *__restrict__ active,      // channel indices to produce
    int n_active, int M, int TF, int64_t n_frames,
    float2 *__restrict__ out, int64_t out_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Mp = M | 1;  // padded row length (odd number of float2) against bank conflicts
    float2 *s_in = (float2 *)smem;                        // (TF + 7) * M
    float2 *s_x = s_in + (size_t)(TF + kChanTaps - 1) * M;  // TF * Mp
    float2 *s_tw = s_x + (size_t)TF * Mp;                 // M
    float *s_taps = (float *)(s_tw + M);                  // M * 8

    const int64_t f0 = (int64_t)blockIdx.x * TF;          // first frame of this tile
    const int nf = (int)min((int64_t)TF, n_frames - f0);
    const int tid = threadIdx.x;

    // stage input: frames f0-7 .. f0+nf-1
    const int64_t s0 = (f0 - (kChanTaps - 1)) * M;        // global sample index of s_in[0] (may be negative)
    const int n_in = (nf + kChanTaps - 1) * M;
    for (int i = tid; i < n_in; i += kChanThreads) {
        int64_t g = s0 + i;
        s_in[i] = g >= 0 ? x[g] : hist[g + (int64_t)(kChanTaps - 1) * M];
    }
    for (int i = tid; i < M; i += kChanThreads) s_tw[i] = tw[i];
    for (int i = tid; i < M * kChanTaps; i += kChanThreads) s_taps[i] = taps[i];
    __syncthreads();

    // polyphase FIR
    for (int i = tid; i < nf * M; i += kChanThreads) {
        const int t = i / M, c = i - t * M;
        float ar = 0.f, ai = 0.f;
#pragma unroll
        for (int n = 0; n < kChanTaps; ++n) {
            const float h = s_taps[c * kChanTaps + n];
            const float2 v = s_in[(t + kChanTaps - 1 - n) * M + c];
            ar = fmaf(h, v.x, ar); ai = fmaf(h, v.y, ai);
        }
        s_x[t * Mp + c] = make_float2(ar, ai);
    }
    __syncthreads();

    // DFT for requested channels: thread = (t, kgroup)
    const int t = tid % TF, kg = tid / TF, KG = kChanThreads / TF;
    if (t < nf) {
        for (int a = kg; a < n_active; a += KG) {
            const int k = active[a];
            float yr = 0.f, yi = 0.f;
            int wi = 0;
            const float2 *row = s_x + t * Mp;
            for (int c = 0; c < M; ++c) {
                const float2 w = s_tw[wi];
                const float2 v = row[c];
                yr = fmaf(v.x, w.x, yr); yr = fmaf(-v.y, w.y, yr);
                yi = fmaf(v.x, w.y, yi); yi = fmaf(v.y, w.x, yi);
                wi += k; if (wi >= M) wi -= M;
            }
            out[(int64_t)k * out_stride + f0 + t] = make_float2(yr, yi);
        }
    }
}

// new history = last 7*M samples of (old history ++ x[0..n))
__global__ void chan_update_hist(const float2 *x, int64_t n, float2 *hist, float2 *hist_new, int H) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H) return;
    int64_t g = n - H + i;  // index into x; negative -> old history
    hist_new[i] = g >= 0 ? x[g] : hist[g + H];
}

}  // namespace csdr
