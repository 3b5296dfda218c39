// kernels_spec.hpp -- SpectrumVisualProcessor arithmetic on the GPU (K13 FFT, K14 magnitude+shift, K15 averaging,
// K16 display resampling / log scaling).
//
// Replaces (reference file:line): fft_execute SpectrumVisualProcessor.cpp:439 (liquid radix-2 plan created :177),
// magnitude + fftshift :441-452, double EMA + running min/max :494-511, floor/ceil EMAs :513-521,
// display loop :532-576.
//
// FFT: N = 2 * fftSize = N1 * N2.  N <= 4096: one LDS Stockham pass (N1 = 1).  Larger: four-step --
// pass A does N1-point column FFTs (C adjacent columns per workgroup, 8C-byte contiguous segments), multiplies by
// W_N^(k1 n2) and writes tmp[k1][n2]; pass B does contiguous N2-point row FFTs and takes |X| straight from LDS.
// Bin k = k1 + N1 k2 is kept in the permuted position p = k1 N2 + k2 for everything element-wise (averagers live in
// that order too); only the final F display points are gathered back to natural order.
#pragma once
#include "common.hpp"

namespace csdr {

constexpr int kFftThreads = 256;
constexpr int kFftMaxLds = 4096;           // complex points per workgroup (2 x 32 KB ping-pong)
constexpr int kTwTab = 4096;               // base twiddle table exp(-2 pi i k / 4096)

__device__ inline float2 cmul(float2 a, float2 b) { return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)); }

// In-LDS Stockham autosort FFT of `nseq` independent sequences of length L (power of two, nseq * L <= 4096).
// Sequence s occupies buf[s * L .. s * L + L).  Radix-4 passes, one radix-2 pass when log2 L is odd.
// Returns the buffer holding the result (a or b).  tw4096[i] = exp(-2 pi i / 4096).
__device__ inline float2 *lds_fft(float2 *a, float2 *b, int L, int nseq, const float2 *__restrict__ tw4096) {
    const int tid = threadIdx.x;
    int Ns = 1;
    float2 *src = a, *dst = b;
    // radix-2 first if odd number of bits
    int bits = 0;
    while ((1 << bits) < L) ++bits;
    if (bits & 1) {
        const int half = L >> 1;
        for (int i = tid; i < nseq * half; i += kFftThreads) {
            const int s = i / half, j = i - s * half;
            const float2 u = src[s * L + j], v = src[s * L + j + half];
            // Ns = 1: twiddle = 1
            dst[s * L + 2 * j] = make_float2(u.x + v.x, u.y + v.y);
            dst[s * L + 2 * j + 1] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
        float2 *t = src; src = dst; dst = t;
        Ns = 2;
    }
    while (Ns < L) {
        const int q = L >> 2;
        const int tstep = kTwTab / (Ns * 4);
        for (int i = tid; i < nseq * q; i += kFftThreads) {
            const int s = i / q, j = i - s * q;
            const int k = j & (Ns - 1);
            const float2 *in = src + s * L;
            float2 v0 = in[j], v1 = in[j + q], v2 = in[j + 2 * q], v3 = in[j + 3 * q];
            if (k) {
                const float2 w1 = tw4096[k * tstep], w2 = tw4096[2 * k * tstep], w3 = tw4096[3 * k * tstep];
                v1 = cmul(v1, w1); v2 = cmul(v2, w2); v3 = cmul(v3, w3);
            }
            // 4-point DFT (forward: -j rotation)
            const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y), a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
            const float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y), a3 = make_float2(v1.x - v3.x, v1.y - v3.y);
            const int j0 = ((j - k) << 2) + k;   // (j / Ns) * Ns * 4 + k
            float2 *out = dst + s * L;
            out[j0] = make_float2(a0.x + a2.x, a0.y + a2.y);
            out[j0 + Ns] = make_float2(a1.x + a3.y, a1.y - a3.x);       // a1 - j a3
            out[j0 + 2 * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
            out[j0 + 3 * Ns] = make_float2(a1.x - a3.y, a1.y + a3.x);   // a1 + j a3
        }
        __syncthreads();
        float2 *t = src; src = dst; dst = t;
        Ns <<= 2;
    }
    return src;
}

struct FrameSrc {             // where frame f starts: frame 0 may live in the carry buffer
    const float2 *first;      // frame 0
    const float2 *rest;       // frame f >= 1 starts at rest + (f - 1) * stride
    int64_t stride;
};
__device__ inline const float2 *frame_ptr(const FrameSrc &fs, int f) { return f == 0 ? fs.first : fs.rest + (int64_t)(f - 1) * fs.stride; }

// pass A: column FFTs.  grid = (N2 / C, frames)
__global__ __launch_bounds__(kFftThreads) void spec_fft_cols(FrameSrc fs, int N1, int N2, int C,
                                                             const float2 *__restrict__ tw4096,
                                                             const float2 *__restrict__ tw_hi, const float2 *__restrict__ tw_lo,
                                                             float2 *__restrict__ tmp) {
    __shared__ float2 sa[kFftMaxLds], sb[kFftMaxLds];
    const int f = blockIdx.y, c0 = blockIdx.x * C, tid = threadIdx.x;
    const float2 *x = frame_ptr(fs, f);
    for (int i = tid; i < N1 * C; i += kFftThreads) {
        const int n1 = i / C, c = i - n1 * C;
        sa[c * N1 + n1] = x[(int64_t)n1 * N2 + c0 + c];
    }
    __syncthreads();
    float2 *r = lds_fft(sa, sb, N1, C, tw4096);
    float2 *o = tmp + (int64_t)f * N1 * N2;
    for (int i = tid; i < N1 * C; i += kFftThreads) {
        const int k1 = i / C, c = i - k1 * C;
        const unsigned q = (unsigned)k1 * (unsigned)(c0 + c);          // < N
        const float2 w = cmul(tw_hi[q >> 10], tw_lo[q & 1023]);
        o[(int64_t)k1 * N2 + c0 + c] = cmul(r[c * N1 + k1], w);
    }
}

// pass B: row FFTs + magnitude.  grid = (N1 / R, frames).  src rows are contiguous (tmp, or the frame itself if N1 == 1).
// Magnitudes are stored as float2 pairs {|X[k_a]|, |X[k_a + 1]|} of the two adjacent bins that one display point
// averages: pair index t = (k1 / 2) N2 + k2 for N1 > 1 (rows k1, k1 + 1 of one workgroup), t = k / 2 for N1 == 1.
__global__ __launch_bounds__(kFftThreads) void spec_fft_rows(FrameSrc fs, int N1, int N2, int R,
                                                             const float2 *__restrict__ tw4096,
                                                             float2 *__restrict__ mag2, float2 *__restrict__ raw_out) {
    __shared__ float2 sa[kFftMaxLds], sb[kFftMaxLds];
    const int f = blockIdx.y, r0 = blockIdx.x * R, tid = threadIdx.x;
    const float2 *x = frame_ptr(fs, f) + (int64_t)r0 * N2;
    for (int i = tid; i < R * N2; i += kFftThreads) sa[i] = x[i];
    __syncthreads();
    float2 *r = lds_fft(sa, sb, N2, R, tw4096);
    const int64_t N = (int64_t)N1 * N2;
    if (mag2) {
        float2 *o = mag2 + (int64_t)f * (N / 2);
        if (N1 > 1) {
            for (int i = tid; i < (R / 2) * N2; i += kFftThreads) {
                const int rp = i / N2, k2 = i - rp * N2;
                const float2 va = r[(2 * rp) * N2 + k2], vb = r[(2 * rp + 1) * N2 + k2];
                o[(int64_t)(r0 / 2 + rp) * N2 + k2] = make_float2(sqrtf(va.x * va.x + va.y * va.y), sqrtf(vb.x * vb.x + vb.y * vb.y));
            }
        } else {
            for (int i = tid; i < N2 / 2; i += kFftThreads) {
                const float2 va = r[2 * i], vb = r[2 * i + 1];
                o[i] = make_float2(sqrtf(va.x * va.x + va.y * va.y), sqrtf(vb.x * vb.x + vb.y * vb.y));
            }
        }
    }
    if (raw_out) {   // natural-order complex output (parity tests of K13 alone): bin k = k1 + N1 k2
        for (int i = tid; i < R * N2; i += kFftThreads) {
            const int rr = i / N2, k2 = i - rr * N2;
            raw_out[(int64_t)f * N + (int64_t)(r0 + rr) + (int64_t)N1 * k2] = r[i];
        }
    }
}

// K15: per display point (= two adjacent shifted bins) run the averaging recurrence over the frames of the batch.
// thread t owns pair t; ma / maa (fft_result_ma / _maa, double) are kept per pair as [2][N/2] arrays.
// One wave per workgroup so the N/2 pairs spread over all CUs; loads of 4 frames are issued ahead of the recurrence.
// The per-frame extrema the reference tracks are (float) max / min of maa; float rounding is monotonic, so each thread
// emits its own (float max, float min) per frame and spec_minmax reduces them -- nothing cross-lane in the serial loop.
constexpr int kAvgThreads = 64;
constexpr int kAvgUnroll = 4;

__global__ __launch_bounds__(kAvgThreads) void spec_average(const float2 *__restrict__ mag2, int nf, int N1, int N2, double rate,
                                                            double *__restrict__ ma, double *__restrict__ maa,
                                                            float *__restrict__ pairsum, float *__restrict__ first_b,
                                                            float2 *__restrict__ ext) {
    const int H = (N1 * N2) / 2;
    const int t = blockIdx.x * kAvgThreads + threadIdx.x;
    if (t >= H) return;
    // shifted index 0 <-> bin N/2: k1 = 0, k2 = N2/2 (N1 > 1)  or  pair N/4 (N1 == 1)
    const bool is_x0 = (N1 > 1 ? (t == N2 / 2) : (t == H / 2));
    double ma_a = ma[t], ma_b = ma[H + t], maa_a = maa[t], maa_b = maa[H + t];
    for (int f0 = 0; f0 < nf; f0 += kAvgUnroll) {
        float2 m[kAvgUnroll];
#pragma unroll
        for (int u = 0; u < kAvgUnroll; ++u) m[u] = (f0 + u < nf) ? mag2[(int64_t)(f0 + u) * H + t] : make_float2(0.f, 0.f);
#pragma unroll
        for (int u = 0; u < kAvgUnroll; ++u) {
            const int f = f0 + u;
            if (f < nf) {
                const double xa = (double)m[u].x, xb = (double)m[u].y;
                if (maa_a != maa_a) maa_a = xa;
                maa_a += (ma_a - maa_a) * rate;
                if (ma_a != ma_a) ma_a = xa;
                ma_a += (xa - ma_a) * rate;
                if (maa_b != maa_b) maa_b = xb;
                maa_b += (ma_b - maa_b) * rate;
                if (ma_b != ma_b) ma_b = xb;
                ma_b += (xb - ma_b) * rate;
                pairsum[(int64_t)f * H + t] = (float)(maa_a + maa_b);
                ext[(int64_t)f * H + t] = make_float2((float)fmax(maa_a, maa_b), (float)fmin(maa_a, maa_b));
                if (is_x0) first_b[f] = (float)maa_b;
            }
        }
    }
    ma[t] = ma_a; ma[H + t] = ma_b; maa[t] = maa_a; maa[H + t] = maa_b;
}

// per-frame reduction of the extrema: grid = frames, 256 threads
struct SpecFrameOut { double point_ceil, point_floor; };
__global__ __launch_bounds__(256) void spec_minmax(const float2 *__restrict__ ext, int H, SpecFrameOut *fo) {
    __shared__ float smx[4], smn[4];
    const int f = blockIdx.x, tid = threadIdx.x;
    float mx = 0.f, mn = 3.0e38f;
    for (int i = tid; i < H; i += 256) { const float2 v = ext[(int64_t)f * H + i]; mx = fmaxf(mx, v.x); mn = fminf(mn, v.y); }
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_down(mx, o, 64)); mn = fminf(mn, __shfl_down(mn, o, 64)); }
    if ((tid & 63) == 0) { smx[tid >> 6] = mx; smn[tid >> 6] = mn; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 4; ++i) { mx = fmaxf(mx, smx[i]); mn = fminf(mn, smn[i]); }
        fo[f].point_ceil = (double)mx; fo[f].point_floor = (double)mn;
    }
}

// floor / ceil trackers across the frames of the batch (short serial recurrences)  SpectrumVisualProcessor.cpp:494-521
struct SpecScalars { double ceil_ma, ceil_maa, floor_ma, floor_maa; };

__global__ void spec_trackers(int nf, SpecScalars *st, SpecFrameOut *fo) {
    if (threadIdx.x || blockIdx.x) return;
    SpecScalars s = *st;
    for (int f = 0; f < nf; ++f) {
        const float mx = (float)fo[f].point_ceil, mn = (float)fo[f].point_floor;
        float fft_ceil = 0.f, fft_floor = 1.f;          // the reference keeps these two in float (:436)
        if (mx > fft_ceil) fft_ceil = mx;
        if (mn < fft_floor) fft_floor = mn;
        if (s.ceil_ma != s.ceil_ma) s.ceil_ma = fft_ceil;
        s.ceil_ma = s.ceil_ma + ((double)fft_ceil - s.ceil_ma) * 0.05;
        if (s.ceil_maa != s.ceil_maa) s.ceil_maa = fft_ceil;
        s.ceil_maa = s.ceil_maa + (s.ceil_ma - s.ceil_maa) * 0.05;
        if (s.floor_ma != s.floor_ma) s.floor_ma = fft_floor;
        s.floor_ma = s.floor_ma + ((double)fft_floor - s.floor_ma) * 0.05;
        if (s.floor_maa != s.floor_maa) s.floor_maa = fft_floor;
        s.floor_maa = s.floor_maa + (s.floor_ma - s.floor_maa) * 0.05;
        fo[f].point_ceil = s.ceil_maa;
        fo[f].point_floor = s.floor_maa;
    }
    *st = s;
}

// K16: display points, full-span view (visualRatio = 1: two bins per point).  grid = (F / 256, frames)
__global__ __launch_bounds__(256) void spec_display(const float *__restrict__ pairsum, const float *__restrict__ first_b,
                                                    const SpecFrameOut *__restrict__ fo, int N1, int N2, float sf,
                                                    float *__restrict__ points) {
    const int N = N1 * N2, F = N / 2;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    if (x >= F) return;
    const int ka = (2 * x + N / 2) & (N - 1);
    int t;
    if (N1 > 1) { const int k1 = ka & (N1 - 1), k2 = ka / N1; t = (k1 >> 1) * N2 + k2; }
    else t = ka >> 1;
    const double pc = fo[f].point_ceil, pf = fo[f].point_floor;
    double acc;
    if (x == 0) acc = pf + (double)first_b[f];      // idx == 0 is replaced by fft_floor_maa (:546-556)
    else acc = (double)pairsum[(int64_t)f * F + t];
    const double v = (log10((acc / 2.0) + 0.25 - (pf - 0.75)) / log10((pc + 0.25) - (pf - 0.75))) * (double)sf;
    float *o = points + ((int64_t)f * F + x) * 2;
    o[0] = (float)x / (float)F;
    o[1] = (float)v;
}

// assemble frame 0 of a contiguous run from (carry ++ head of the new data)
__global__ void spec_assemble(const float2 *carry, int ncarry, const float2 *x, int n, float2 *dst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = i < ncarry ? carry[i] : x[i - ncarry];
}

}  // namespace csdr
