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

// pass B: row FFTs + magnitude.  grid = (N1 / R, frames).  src rows are contiguous (tmp, or the frame itself if N1 == 1)
__global__ __launch_bounds__(kFftThreads) void spec_fft_rows(FrameSrc fs, int N1, int N2, int R,
                                                             const float2 *__restrict__ tw4096,
                                                             float *__restrict__ mag, float2 *__restrict__ raw_out) {
    __shared__ float2 sa[kFftMaxLds], sb[kFftMaxLds];
    const int f = blockIdx.y, r0 = blockIdx.x * R, tid = threadIdx.x;
    const float2 *x = frame_ptr(fs, f) + (int64_t)r0 * N2;
    for (int i = tid; i < R * N2; i += kFftThreads) sa[i] = x[i];
    __syncthreads();
    float2 *r = lds_fft(sa, sb, N2, R, tw4096);
    const int64_t base = (int64_t)f * N1 * N2 + (int64_t)r0 * N2;
    if (mag) {
        for (int i = tid; i < R * N2; i += kFftThreads) {
            const float2 v = r[i];
            mag[base + i] = sqrtf(v.x * v.x + v.y * v.y);
        }
    }
    if (raw_out) {   // natural-order complex output (parity tests of K13 alone): bin k = k1 + N1 k2
        for (int i = tid; i < R * N2; i += kFftThreads) {
            const int rr = i / N2, k2 = i - rr * N2;
            raw_out[(int64_t)f * N1 * N2 + (int64_t)(r0 + rr) + (int64_t)N1 * k2] = r[i];
        }
    }
}

// K15: per display point (= two adjacent shifted bins) run the averaging recurrence over the frames of the batch.
// thread t <-> (k1 pair, k2): bins k_a = 2 k1p + N1 k2 and k_a + 1 at permuted positions p_a, p_a + N2 (N1 > 1)
// or p_a = 2 t, p_a + 1 (N1 == 1).  ma / maa (fft_result_ma / _maa, double) live in permuted order.
struct SpecMinMax { unsigned long long mx, mn; };   // bit patterns of non-negative doubles (order-preserving)

__global__ __launch_bounds__(256) void spec_average(const float *__restrict__ mag, int nf, int N1, int N2, double rate,
                                                    double *__restrict__ ma, double *__restrict__ maa,
                                                    float *__restrict__ pairsum, float *__restrict__ first_b,
                                                    SpecMinMax *__restrict__ mm) {
    const int N = N1 * N2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < N / 2;
    int pa = 0, pb = 0;
    bool is_x0 = false;
    if (live) {
        if (N1 > 1) { const int k1p = t / N2, k2 = t - k1p * N2; pa = (2 * k1p) * N2 + k2; pb = pa + N2; is_x0 = (k1p == 0 && k2 == N2 / 2); }
        else { pa = 2 * t; pb = pa + 1; is_x0 = (pa == N / 2); }
    }
    // shifted index 0 <-> bin N/2: k1 = 0, k2 = N2/2 (N1 > 1) ; that thread's first bin is the "idx == 0" element
    double ma_a = 0, ma_b = 0, maa_a = 0, maa_b = 0;
    if (live) { ma_a = ma[pa]; ma_b = ma[pb]; maa_a = maa[pa]; maa_b = maa[pb]; }
    for (int f = 0; f < nf; ++f) {
        double lmx = 0.0, lmn = 1e300;
        if (live) {
            const float *m = mag + (int64_t)f * N;
            const double xa = (double)m[pa], xb = (double)m[pb];
            if (maa_a != maa_a) maa_a = xa;
            maa_a += (ma_a - maa_a) * rate;
            if (ma_a != ma_a) ma_a = xa;
            ma_a += (xa - ma_a) * rate;
            if (maa_b != maa_b) maa_b = xb;
            maa_b += (ma_b - maa_b) * rate;
            if (ma_b != ma_b) ma_b = xb;
            ma_b += (xb - ma_b) * rate;
            lmx = fmax(maa_a, maa_b); lmn = fmin(maa_a, maa_b);
            pairsum[(int64_t)f * (N / 2) + t] = (float)(maa_a + maa_b);
            if (is_x0) first_b[f] = (float)maa_b;
        }
        for (int o = 32; o > 0; o >>= 1) { lmx = fmax(lmx, __shfl_down(lmx, o, 64)); lmn = fmin(lmn, __shfl_down(lmn, o, 64)); }
        if ((threadIdx.x & 63) == 0) {
            atomicMax(&mm[f].mx, (unsigned long long)__double_as_longlong(lmx));
            atomicMin(&mm[f].mn, (unsigned long long)__double_as_longlong(lmn));
        }
    }
    if (live) { ma[pa] = ma_a; ma[pb] = ma_b; maa[pa] = maa_a; maa[pb] = maa_b; }
}

// floor / ceil trackers across the frames of the batch (sequential, one thread)  SpectrumVisualProcessor.cpp:494-521
struct SpecScalars { double ceil_ma, ceil_maa, floor_ma, floor_maa; };
struct SpecFrameOut { double point_ceil, point_floor; };

__global__ void spec_trackers(const SpecMinMax *__restrict__ mm, int nf, SpecScalars *st, SpecFrameOut *fo) {
    if (threadIdx.x || blockIdx.x) return;
    SpecScalars s = *st;
    for (int f = 0; f < nf; ++f) {
        const double mx = __longlong_as_double((long long)mm[f].mx), mn = __longlong_as_double((long long)mm[f].mn);
        float fft_ceil = 0.f, fft_floor = 1.f;          // the reference keeps these two in float (:436)
        if (mx > (double)fft_ceil) fft_ceil = (float)mx;
        if (mn < (double)fft_floor) fft_floor = (float)mn;
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

__global__ void spec_reset_minmax(SpecMinMax *mm, int nf) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nf) { mm[i].mx = 0ull; mm[i].mn = (unsigned long long)__double_as_longlong(1e300); }
}

// assemble frame 0 of a contiguous run from (carry ++ head of the new data)
__global__ void spec_assemble(const float2 *carry, int ncarry, const float2 *x, int n, float2 *dst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = i < ncarry ? carry[i] : x[i - ncarry];
}

}  // namespace csdr
