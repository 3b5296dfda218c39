// kernels_spec.hpp -- SpectrumVisualProcessor arithmetic on the GPU (K13 FFT, K14 magnitude+shift, K15 averaging,
// K16 display resampling / log scaling).
//
// Replaces (reference file:line): fft_execute SpectrumVisualProcessor.cpp:439 (liquid radix-2 plan created :177),
// magnitude + fftshift :441-452, double EMA + running min/max :494-511, floor/ceil EMAs :513-521,
// display loop :532-576.
//
// FFT of N = 2 * fftSize points:
//   N <= 2048              one LDS Stockham pass per frame (spec_fft_small)
//   N == 4096              one register/LDS pass per frame (spec_fft_rows4096)
//   N == Ra * 4096         spec_fft_radix<Ra> (Ra-point column DFTs in registers, stride N / Ra, times W_N^(k1 n))
//                          then 4096-point row FFTs;  bin k = k1 + Ra k3
//   N == Ra * Rb * 4096    a second radix pass inside each of the Ra sub-sequences; bin k = k1 + Ra (k2 + Rb k3)
// The 4096-point row FFT keeps 16 points per thread in registers: three 16-point DFTs with two LDS transposes
// (conflict-free padded layouts) instead of six radix-4 passes through LDS.
// Row FFTs run on row PAIRS (k1 even, k1 + 1): the two adjacent bins one display point averages come out together
// as a float2, stored at pair index t = ((k1 / 2) Rb + k2) 4096 + k3  (t = k / 2 when there is a single row).
// Everything after the FFT works in display order (point x <-> bins ka = (2 x + N / 2) mod N and ka + 1).
//
// All LDS is dynamic (`smem`).
#pragma once
#include "common.hpp"

namespace csdr {

constexpr int kFftThreads = 256;
constexpr int kFftMaxLds = 4096;           // complex points per workgroup in the Stockham path (2 x 32 KB ping-pong)
constexpr int kTwTab = 4096;               // base twiddle table exp(-2 pi i k / 4096)
constexpr size_t kRowLdsBytes = 8 * 272 * sizeof(float2);   // exchange buffer of the 4096-point row FFT (half of the points at a time)

struct SpecGeom {
    int N, F;                 // internal FFT size, display points (= N / 2)
    int Ra, Rb, lgRa, lgRb;   // radix passes in front of the 4096-point rows (1 = absent)
    int N2;                   // row length: 4096 when N >= 4096, else N
    int npot;                 // 1: N is not a power of two (N <= 2048: chirp-z transform, spec_fft_bluestein); the fftshift then is a modulo, not a mask
};

// ---- twiddle table lookups ----------------------------------------------------------------------------------
__device__ inline float2 tw_split(const float2 *__restrict__ tw_hi, const float2 *__restrict__ tw_lo, unsigned q) {
    return cmul(tw_hi[q >> 10], tw_lo[q & 1023u]);     // exp(-2 pi i q / N), q < N
}

// ---- in-register DFT of R points (R = 2, 4, 8, 16, 32), natural order in and out, forward transform -----------
__device__ inline float2 w32(int k) {    // exp(-2 pi i k / 32), k in [0, 16)
    constexpr float c[16] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                             0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f, 0.0f, -0.19509032201612825f,
                             -0.38268343236508977f, -0.55557023301960218f, -0.70710678118654752f, -0.83146961230254524f,
                             -0.92387953251128674f, -0.98078528040323043f};
    constexpr float s[16] = {0.0f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f, 0.70710678118654752f,
                             0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f, 1.0f, 0.98078528040323043f,
                             0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960218f,
                             0.38268343236508977f, 0.19509032201612825f};
    return make_float2(c[k], -s[k]);
}
template <int R>
__device__ inline constexpr int bitrev(int i) {
    int r = 0;
    for (int b = 1; b < R; b <<= 1) { r = (r << 1) | (i & 1); i >>= 1; }
    return r;
}
template <int R>
__device__ inline void dft_reg(float2 (&v)[R]) {
    float2 t[R];
#pragma unroll
    for (int i = 0; i < R; ++i) t[bitrev<R>(i)] = v[i];
#pragma unroll
    for (int len = 2; len <= R; len <<= 1) {
#pragma unroll
        for (int i = 0; i < R; i += len) {
#pragma unroll
            for (int j = 0; j < len / 2; ++j) {
                const int tk = j * (32 / len);      // twiddle exponent on the 32-point circle
                const float2 a = t[i + j];
                float2 b = t[i + j + len / 2];
                if (tk == 8) b = make_float2(b.y, -b.x);            // times -i
                else if (tk != 0) b = cmul(b, w32(tk));
                t[i + j] = make_float2(a.x + b.x, a.y + b.y);
                t[i + j + len / 2] = make_float2(a.x - b.x, a.y - b.y);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = t[i];
}

// v[k] *= p^k for k = 1 .. R-1, given the exact leaves p^1, p^2, p^4, p^8, p^16 (products are at most 4 deep)
template <int R>
__device__ inline void twiddle_powers(float2 (&v)[R], const float2 (&leaf)[5]) {
    float2 p[R];
    p[0] = make_float2(1.f, 0.f);
#pragma unroll
    for (int k = 1; k < R; ++k) {
        const int hb = (k >= 16) ? 16 : (k >= 8) ? 8 : (k >= 4) ? 4 : (k >= 2) ? 2 : 1;
        const int li = (hb == 16) ? 4 : (hb == 8) ? 3 : (hb == 4) ? 2 : (hb == 2) ? 1 : 0;
        p[k] = (k == hb) ? leaf[li] : cmul(leaf[li], p[k - hb]);
        v[k] = cmul(v[k], p[k]);
    }
}

// ---- 4096-point FFT, 256 threads x 16 points -------------------------------------------------------------------
// in : v[r] = x[tid + 256 r]        out: v[r] = X[tid + 256 r]
// n = n0 + 16 n1 + 256 n2, k = 256 k0 + 16 k1 + k2:
//   W^(nk) = W16^(n2 k2) W256^(n1 k2) . W16^(n1 k1) W4096^(n0 (16 k1 + k2)) . W16^(n0 k0)
__device__ inline void fft4096_regs(float2 (&v)[16], float2 *lds, const float2 *__restrict__ tw4096) {
    const int tid = threadIdx.x, n0 = tid & 15, hi = tid >> 4;
    float2 leaf[5];
    // stage 1: DFT over n2 (registers), times W256^(n1 k2), n1 = hi
    dft_reg<16>(v);
    {
        const int b = 16 * hi;
        leaf[0] = tw4096[b]; leaf[1] = tw4096[2 * b]; leaf[2] = tw4096[4 * b]; leaf[3] = tw4096[8 * b]; leaf[4] = leaf[3];
        twiddle_powers<16>(v, leaf);
    }
    // exchange 1, [k2][n1][n0] with rows padded to 272, in two halves of eight k2 (the buffer holds 2048 points):
    // the threads whose k2 (= hi) lies in the half that is resident pick up their sixteen n1 values
    float2 u[16];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                                       // previous users of the exchange buffer are done
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) lds[tid + 272 * k2] = v[8 * half + k2];
        __syncthreads();
        if ((hi >> 3) == half) {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) u[n1] = lds[n0 + 16 * n1 + 272 * (hi & 7)];
        }
    }
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = u[n1];
    // stage 2: thread (n0, k2 = hi): DFT over n1, times W4096^(n0 (16 k1 + k2))
    dft_reg<16>(v);
    {
        const int b = 16 * n0;
        leaf[0] = tw4096[b]; leaf[1] = tw4096[2 * b]; leaf[2] = tw4096[4 * b]; leaf[3] = tw4096[8 * b]; leaf[4] = leaf[3];
        twiddle_powers<16>(v, leaf);
        const float2 b0 = tw4096[n0 * hi];
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) v[k1] = cmul(v[k1], b0);
    }
    // exchange 2, [n0][k1][k2] with rows padded to 257, in two halves of eight n0: the threads whose n0 lies in the half
    // write their sixteen k1 values, every thread (tid = k2 + 16 k1) reads eight n0 values
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        if ((n0 >> 3) == half) {
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) lds[hi + 16 * k1 + 257 * (n0 & 7)] = v[k1];
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 8; ++m) u[8 * half + m] = lds[tid + 257 * m];
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = u[m];
    // stage 3: thread tid = k2 + 16 k1: DFT over n0
    dft_reg<16>(v);
}

// ---- In-LDS Stockham autosort FFT of one sequence of length L <= 2048 (generic small sizes) ---------------------
__device__ inline float2 *lds_fft(float2 *a, float2 *b, int L, const float2 *__restrict__ tw4096) {
    const int tid = threadIdx.x;
    int Ns = 1;
    float2 *src = a, *dst = b;
    int bits = 0;
    while ((1 << bits) < L) ++bits;
    if (bits & 1) {
        const int half = L >> 1;
        for (int j = tid; j < half; j += kFftThreads) {
            const float2 u = src[j], w = src[j + half];
            dst[2 * j] = make_float2(u.x + w.x, u.y + w.y);
            dst[2 * j + 1] = make_float2(u.x - w.x, u.y - w.y);
        }
        __syncthreads();
        float2 *t = src; src = dst; dst = t;
        Ns = 2;
    }
    while (Ns < L) {
        const int q = L >> 2;
        const int tstep = kTwTab / (Ns * 4);
        for (int j = tid; j < q; j += kFftThreads) {
            const int k = j & (Ns - 1);
            float2 v0 = src[j], v1 = src[j + q], v2 = src[j + 2 * q], v3 = src[j + 3 * q];
            if (k) {
                v1 = cmul(v1, tw4096[k * tstep]); v2 = cmul(v2, tw4096[2 * k * tstep]); v3 = cmul(v3, tw4096[3 * k * tstep]);
            }
            const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y), a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
            const float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y), a3 = make_float2(v1.x - v3.x, v1.y - v3.y);
            const int j0 = ((j - k) << 2) + k;
            dst[j0] = make_float2(a0.x + a2.x, a0.y + a2.y);
            dst[j0 + Ns] = make_float2(a1.x + a3.y, a1.y - a3.x);       // a1 - j a3
            dst[j0 + 2 * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
            dst[j0 + 3 * Ns] = make_float2(a1.x - a3.y, a1.y + a3.x);   // a1 + j a3
        }
        __syncthreads();
        float2 *t = src; src = dst; dst = t;
        Ns <<= 2;
    }
    return src;
}

struct FrameSrc {             // where sequence f starts
    const float2 *first;      // sequence 0: elements [0, split) ...
    const float2 *first2;     // ... and elements [split, L) at first2[i - split] (contiguous run: carry ++ head of the new data)
    const float2 *rest;       // sequence f >= 1 starts at rest + (f - 1) * stride
    int64_t stride;
    int split;                // >= L when sequence 0 is one piece
};
__device__ inline const float2 *frame_ptr(const FrameSrc &fs, int f) { return f == 0 ? fs.first : fs.rest + (int64_t)(f - 1) * fs.stride; }
// element i of sequence f (sequence 0 may be split in two pieces)
__device__ inline float2 frame_at(const FrameSrc &fs, int f, const float2 *base, int64_t i) {
    if (f == 0 && i >= fs.split) return fs.first2[i - fs.split];
    return base[i];
}

// |v|: the hardware square root (v_sqrt_f32, one ulp) -- the library sqrtf wraps it in a range-scaling sequence of seven instructions per value,
// which at sixteen magnitudes per thread was a tenth of the row pass's instruction stream (the pass is issue-bound: DESIGN 12)
__device__ inline float cabs_f(float2 v) {
#if defined(__AMDGCN__)
    return __builtin_amdgcn_sqrtf(v.x * v.x + v.y * v.y);
#else
    return sqrtf(v.x * v.x + v.y * v.y);
#endif
}

// ---- radix pass: R-point column DFTs over stride Lr = L / R, times W_L^(k c); grid = (Lr / (256 COLS), sequences) --
// W_L^q = exp(-2 pi i q tw_scale / N) is looked up in the split tables of the full transform.
template <int R, int COLS>
CSDR_KERNEL __launch_bounds__(kFftThreads) void spec_fft_radix(FrameSrc fs, int L, unsigned tw_scale,
                                                              const float2 *__restrict__ tw_hi, const float2 *__restrict__ tw_lo,
                                                              float2 *dst) {
    const int Lr = L / R;
    const int c = COLS * (blockIdx.x * kFftThreads + threadIdx.x);
    if (c >= Lr) return;
    const int s = blockIdx.y;
    const float2 *xb = frame_ptr(fs, s);
    const float2 *x = xb + c;
    float2 *o = dst + (int64_t)s * L + c;
    float2 a[R], b[R];
    if (s == 0 && fs.split < L) {                       // block-uniform: the split sequence is read element-wise
#pragma unroll
        for (int r = 0; r < R; ++r) {
            a[r] = frame_at(fs, 0, xb, (int64_t)r * Lr + c);
            if (COLS == 2) b[r] = frame_at(fs, 0, xb, (int64_t)r * Lr + c + 1);
        }
    } else if (COLS == 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const f4u v = *reinterpret_cast<const f4u *>(x + (int64_t)r * Lr);
            a[r] = make_float2(v.x, v.y); b[r] = make_float2(v.z, v.w);
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = ld_stream(x + (int64_t)r * Lr);
    }
    float2 leaf[5];
    dft_reg<R>(a);
#pragma unroll
    for (int l = 0; l < 5; ++l) leaf[l] = ((1 << l) < R) ? tw_split(tw_hi, tw_lo, tw_scale * (unsigned)(c << l)) : make_float2(1.f, 0.f);
    twiddle_powers<R>(a, leaf);
    if (COLS == 2) {
        dft_reg<R>(b);
#pragma unroll
        for (int l = 0; l < 5; ++l) leaf[l] = ((1 << l) < R) ? tw_split(tw_hi, tw_lo, tw_scale * (unsigned)((c + 1) << l)) : make_float2(1.f, 0.f);
        twiddle_powers<R>(b, leaf);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            f4u v; v.x = a[k].x; v.y = a[k].y; v.z = b[k].x; v.w = b[k].y;
            *reinterpret_cast<f4u *>(o + (int64_t)k * Lr) = v;
        }
    } else {
#pragma unroll
        for (int k = 0; k < R; ++k) st_stream(o + (int64_t)k * Lr, a[k]);
    }
}

// ---- 4096-point row FFTs + magnitude.  grid = (rows = Ra Rb, frames) ---------------------------------------------
// row r = k1 Rb + k2 of frame f in `fs` ([f][r][4096]; the frame itself when there is a single row) holds the bins
// k = k1 + Ra (k2 + Rb k3); |X| is stored as mag[f][r][k3] (float), i.e. natural bin order when there is one row.
CSDR_KERNEL_SPEC __launch_bounds__(kFftThreads) void spec_fft_rows4096(FrameSrc fs, SpecGeom g, const float2 *__restrict__ tw4096,
                                                                 float *__restrict__ mag, float2 *__restrict__ raw_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    const int f = blockIdx.y, row = blockIdx.x, tid = threadIdx.x;
    const float2 *xb = frame_ptr(fs, f);
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = frame_at(fs, f, xb, (int64_t)row * 4096 + tid + 256 * r);
    fft4096_regs(v, lds, tw4096);
    if (mag) {
        float *o = mag + (int64_t)f * g.N + (int64_t)row * 4096;
#pragma unroll
        for (int r = 0; r < 16; ++r) st_stream(o + tid + 256 * r, cabs_f(v[r]));
    }
    if (raw_out) {   // natural-order complex output (parity tests of K13 alone)
        const int k1 = row >> g.lgRb, k2 = row & (g.Rb - 1);
        const int64_t rb = k1 + (int64_t)g.Ra * k2, rs = (int64_t)g.Ra * g.Rb;
#pragma unroll
        for (int r = 0; r < 16; ++r) raw_out[(int64_t)f * g.N + rb + rs * (tid + 256 * r)] = v[r];
    }
}

// ---- small transforms (N <= 2048): one frame per workgroup, Stockham through LDS.  grid = (1, frames) ----------
CSDR_KERNEL_SPEC __launch_bounds__(kFftThreads) void spec_fft_small(FrameSrc fs, int N, const float2 *__restrict__ tw4096,
                                                              float *__restrict__ mag, float2 *__restrict__ raw_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *sa = reinterpret_cast<float2 *>(smem), *sb = sa + N;
    const int f = blockIdx.y, tid = threadIdx.x;
    const float2 *x = frame_ptr(fs, f);
    for (int i = tid; i < N; i += kFftThreads) sa[i] = frame_at(fs, f, x, i);
    __syncthreads();
    const float2 *r = lds_fft(sa, sb, N, tw4096);
    if (mag) {
        float *o = mag + (int64_t)f * N;
        for (int i = tid; i < N; i += kFftThreads) o[i] = cabs_f(r[i]);
    }
    if (raw_out) for (int i = tid; i < N; i += kFftThreads) raw_out[(int64_t)f * N + i] = r[i];
}

// ---- any even N (setFFTSize takes any size, SpectrumVisualProcessor.cpp:180-190; liquid plans a mixed-radix / Rader transform): the
// chirp-z transform.  With w[n] = exp(-i pi n^2 / N):  X[k] = w[k] sum_n (x[n] w[n]) conj(w[k - n]) -- a circular convolution of length
// L = 2^p >= 2 N - 1 with the (precomputed, transformed in double on the host) chirp filter Bf.  One frame per workgroup, both L-point
// transforms in LDS (the inverse one as conj(FFT(conj(.))) / L).  grid = (1, frames), LDS 2 L float2.
CSDR_KERNEL_SPEC __launch_bounds__(kFftThreads) void spec_fft_bluestein(FrameSrc fs, int N, int L, const float2 *__restrict__ tw4096, const float2 *__restrict__ chirp /* [N] w */,
                                                                  const float2 *__restrict__ Bf /* [L] */, float *__restrict__ mag, float2 *__restrict__ raw_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *sa = reinterpret_cast<float2 *>(smem), *sb = sa + L;
    const int f = blockIdx.y, tid = threadIdx.x;
    const float2 *x = frame_ptr(fs, f);
    for (int i = tid; i < L; i += kFftThreads) sa[i] = i < N ? cmul(frame_at(fs, f, x, i), chirp[i]) : make_float2(0.f, 0.f);
    __syncthreads();
    float2 *r = lds_fft(sa, sb, L, tw4096);
    float2 *o = r == sa ? sb : sa;
    for (int i = tid; i < L; i += kFftThreads) { const float2 v = cmul(r[i], Bf[i]); o[i] = make_float2(v.x, -v.y); }
    __syncthreads();
    const float2 *y = lds_fft(o, r, L, tw4096);
    const float inv = 1.0f / (float)L;
    for (int k = tid; k < N; k += kFftThreads) {
        const float2 v = cmul(chirp[k], make_float2(y[k].x * inv, -y[k].y * inv));
        if (mag) mag[(int64_t)f * N + k] = cabs_f(v);
        if (raw_out) raw_out[(int64_t)f * N + k] = v;
    }
}

// ---- the same transform with a convolution longer than 4096 points (fftSize above 1024): the two L-point transforms run as the power-of-two chain
// of size L (spec_run_pow2: natural-order complex output), these three kernels do the element-wise steps between them.  grid = (ceil(n / 256), frames)
CSDR_KERNEL_SPEC __launch_bounds__(kFftThreads) void spec_blue_pre(FrameSrc fs, int f0, int N, int L, const float2 *__restrict__ chirp, float2 *__restrict__ a) {
    const int f = blockIdx.y, i = blockIdx.x * kFftThreads + threadIdx.x;      // frame f0 + f of the call into slot f of the work array
    if (i >= L) return;
    a[(int64_t)f * L + i] = i < N ? cmul(frame_at(fs, f0 + f, frame_ptr(fs, f0 + f), i), chirp[i]) : make_float2(0.f, 0.f);
}
CSDR_KERNEL_SPEC __launch_bounds__(kFftThreads) void spec_blue_mid(const float2 *__restrict__ b, const float2 *__restrict__ Bf, int L, float2 *__restrict__ a) {
    const int f = blockIdx.y, i = blockIdx.x * kFftThreads + threadIdx.x;
    if (i >= L) return;
    const float2 v = cmul(b[(int64_t)f * L + i], Bf[i]);
    a[(int64_t)f * L + i] = make_float2(v.x, -v.y);               // the inverse transform as conj(FFT(conj(.))) / L
}
CSDR_KERNEL_SPEC __launch_bounds__(kFftThreads) void spec_blue_post(const float2 *__restrict__ y, const float2 *__restrict__ chirp, int N, int L,
                                                              float *__restrict__ mag, float2 *__restrict__ raw_out) {
    const int f = blockIdx.y, k = blockIdx.x * kFftThreads + threadIdx.x;
    if (k >= N) return;
    const float inv = 1.0f / (float)L;
    const float2 q = y[(int64_t)f * L + k];
    const float2 v = cmul(chirp[k], make_float2(q.x * inv, -q.y * inv));
    if (mag) mag[(int64_t)f * N + k] = cabs_f(v);
    if (raw_out) raw_out[(int64_t)f * N + k] = v;
}

// ---- K15: averaging recurrences, display order -----------------------------------------------------------------
// Display point x owns the two adjacent shifted bins ka = (2 x + N / 2) mod N and ka + 1; fft_result_ma / _maa (double)
// are kept per point as [2][F] arrays.  The reference runs, per bin and per frame,
//     maa += (ma - maa) rate;   ma += (x - ma) rate                                  (SpectrumVisualProcessor.cpp:494-511)
// a linear recurrence in the frame index.  A workgroup owns 64 points; its 16 waves split the frames of the batch into
// 16 consecutive groups: every thread loads the magnitudes of its group, runs the recurrence from a zero state, the
// group end states are combined through LDS (state_in(g) = M^G state_in(g-1) + local_end(g-1), M = [[a,0],[rate,a]],
// a = 1 - rate), and the thread re-runs its frames from the true entering state with the reference's statements.
// In exact arithmetic this is the sequential result; in double it differs by ~1e-16 relative.  A round of a tile that holds a
// magnitude that is not finite (or enters with a NaN state) is run in frame order instead, with the reference's NaN repairs
// applied frame by frame (:494-497): the reference recovers two frames after a NaN sample, and so does this.
// The per-frame extrema the reference tracks are (float) max / min of maa; float rounding is monotonic, so each thread
// forms (float max, float min) per frame and the wave (= 64 points of one frame) reduces them.
constexpr int kAvgLanes = 64;
constexpr int kAvgGroups = 16;             // most frame groups (waves) per workgroup
constexpr int kAvgGroupsDefault = 8;       // what the host launches unless told otherwise: two workgroups share a CU (measured on C3: 16 / 8 / 4 groups = 0.213 / 0.186 / 0.189 ms)
constexpr int kAvgThreads = kAvgLanes * kAvgGroups;
constexpr int kAvgGMax = 16;               // frames per thread per round -> 256 frames per round

// offsets (in floats, inside one frame of `mag`) of the two bins of display point x; db = distance between them
__device__ inline int64_t spec_pair_offset(const SpecGeom &g, int x, int64_t &db) {
    if (g.npot) {                                                     // any even N: bins (2 x + N / 2) mod N and (2 x + 1 + N / 2) mod N (:441-452, :532-560)
        const int ka = (2 * x + g.N / 2) % g.N, kb = (2 * x + 1 + g.N / 2) % g.N;
        db = kb - ka;
        return ka;
    }
    const int ka = (2 * x + g.N / 2) & (g.N - 1);
    if (g.Ra == 1) { db = 1; return ka; }
    const int k1 = ka & (g.Ra - 1), rest = ka >> g.lgRa;
    const int k2 = rest & (g.Rb - 1), k3 = rest >> g.lgRb;
    db = (int64_t)g.Rb * 4096;                                        // bin ka + 1 lives in row (k1 + 1) Rb + k2
    return ((int64_t)k1 * g.Rb + k2) * 4096 + k3;
}
// index of display point x in the PAIR order of the last FFT pass (pair row ((k1 / 2) Rb + k2), position k3): the averaging kernel
// reads the magnitudes and writes the averaged pair sums in this order (whole 256-byte runs per wave; display-order stores were
// 8 bytes per 64: measured 4.2 x the bytes as 32-byte partial writes); the display kernel permutes on its read side
__device__ inline int64_t spec_pair_index(const SpecGeom &g, int x) {
    if (g.npot) return ((2 * x + g.N / 2) % g.N) >> 1;
    const int ka = (2 * x + g.N / 2) & (g.N - 1);
    if (g.Ra == 1) return ka >> 1;
    const int k1 = ka & (g.Ra - 1), rest = ka >> g.lgRa;
    const int k2 = rest & (g.Rb - 1), k3 = rest >> g.lgRb;
    return ((int64_t)(k1 >> 1) * g.Rb + k2) * 4096 + k3;
}
// where the averagers of display point x live in ma / maa (bin a at [i], bin b at [F + i]): display order for the single-pass sizes,
// pair order behind a multi-pass transform
__device__ inline int64_t spec_state_index(const SpecGeom &g, int x) { return (g.Ra == 1 || g.npot) ? (int64_t)x : spec_pair_index(g, x); }
__device__ inline float2 spec_load_pair(const float *__restrict__ mag, int64_t off, int64_t db) {
    if (db == 1) return *reinterpret_cast<const float2 *>(mag + off);   // ka is even: 8-byte aligned
    return make_float2(mag[off], mag[off + db]);
}

struct AvgState { double ma_a, maa_a, ma_b, maa_b; };
__device__ inline void avg_step(AvgState &s, double xa, double xb, double rate) {   // the reference's statements, both bins
    if (s.maa_a != s.maa_a) s.maa_a = xa;
    s.maa_a += (s.ma_a - s.maa_a) * rate;
    if (s.ma_a != s.ma_a) s.ma_a = xa;
    s.ma_a += (xa - s.ma_a) * rate;
    if (s.maa_b != s.maa_b) s.maa_b = xb;
    s.maa_b += (s.ma_b - s.maa_b) * rate;
    if (s.ma_b != s.ma_b) s.ma_b = xb;
    s.ma_b += (xb - s.ma_b) * rate;
}
// the same statements when no state is NaN (the repairs are no-ops): checked once per round, see below
__device__ __forceinline__ void avg_step_fast(AvgState &s, double xa, double xb, double rate) {
    s.maa_a += (s.ma_a - s.maa_a) * rate; s.ma_a += (xa - s.ma_a) * rate;
    s.maa_b += (s.ma_b - s.maa_b) * rate; s.ma_b += (xb - s.ma_b) * rate;
}
// loads / stores at a wave-uniform base plus a 32-bit per-lane byte offset (scalar-base addressing)
__device__ __forceinline__ float ldf(const float *base, unsigned byte_off) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off); }
__device__ __forceinline__ float2 ldf2(const float *base, unsigned byte_off) { return *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(base) + byte_off); }
__device__ __forceinline__ void stf(float *base, unsigned byte_off, float v) { st_stream(reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off), v); }
// max / min over each row of 16 lanes (every lane of the row gets the result): four DPP steps, operands fused into the ALU op
__device__ __forceinline__ void row16_max_min(float &mx, float &mn) {
#if defined(__AMDGCN__)
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_min_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                 : "+v"(mx), "+v"(mn));
#else
    for (int o = 1; o < 16; o <<= 1) { mx = fmaxf(mx, __shfl_xor(mx, o, 64)); mn = fminf(mn, __shfl_xor(mn, o, 64)); }
#endif
}

constexpr int kAvgExtFrames = 4;           // frames whose per-lane extrema sit in LDS before one transposed reduction
constexpr size_t kAvgLds = (size_t)(kAvgGroups + 1) * kAvgLanes * 4 * sizeof(double) + (size_t)kAvgGroups * 2 * kAvgExtFrames * kAvgLanes * sizeof(float) + 16;      // the largest request (hipFuncSetAttribute)
__host__ __device__ constexpr size_t avg_lds_bytes(int ng) {      // dynamic LDS of a launch with ng frame groups
    return (size_t)(ng + 1) * kAvgLanes * 4 * sizeof(double) + (size_t)ng * 2 * kAvgExtFrames * kAvgLanes * sizeof(float) + 16;
}

CSDR_KERNEL_SPEC __launch_bounds__(kAvgThreads) void spec_average(const float *__restrict__ mag, int nf, SpecGeom g, double rate,
                                                            double *__restrict__ ma, double *__restrict__ maa,
                                                            float *__restrict__ pairsum /* pair order */, float *__restrict__ first_b,
                                                            float2 *__restrict__ ext_w,
                                                            float2 *__restrict__ maaf /* peak hold / zoomed view: both averaged bins of every point, frames >= pk_from */,
                                                            int pk_from) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ng = blockDim.x >> 6;                                      // frame groups in this launch (1 .. kAvgGroups: few frames, few groups)
    AvgState *s_loc = reinterpret_cast<AvgState *>(smem);               // [ng][64] group end states (zero entering state)
    AvgState *s_carry = s_loc + ng * kAvgLanes;                          // [64] state after the round
    const int F = g.F, lane = threadIdx.x & 63, grp = wave_uniform((int)(threadIdx.x >> 6));
    float *s_ex_all = reinterpret_cast<float *>(s_carry + kAvgLanes);
    float *s_ex = s_ex_all + grp * 2 * kAvgExtFrames * kAvgLanes;       // this wave's [max | min][4 frames][64 lanes]
    // The magnitudes lie in the row order of the last FFT pass: bin k1 + Ra (k2 + Rb k3) at [(k1 Rb + k2) 4096 + k3].  A tile is
    // 64 consecutive k3 of ONE row pair (k1 even, k1 + 1): two 256-byte runs per load, and the averaged pair sums are written in
    // the same pair order (one 256-byte run per store); the display kernel does the permutation to display order on its read side.
    int x;
    bool valid;
    int64_t t, db, pt;
    if (g.Ra == 1) {
        x = blockIdx.x * kAvgLanes + lane;
        valid = x < F;
        t = spec_pair_offset(g, valid ? x : 0, db);
        pt = t >> 1;
    } else {
        const int prow = blockIdx.x >> 6, k3 = ((blockIdx.x & 63) << 6) + lane;      // 4096 / 64 = 64 tiles per row pair
        const int k1 = 2 * (prow >> g.lgRb), k2 = prow & (g.Rb - 1);
        const int ka = k1 + g.Ra * (k2 + g.Rb * k3);
        x = ((ka - g.N / 2) & (g.N - 1)) >> 1;
        valid = true;
        t = ((int64_t)k1 * g.Rb + k2) * 4096 + k3;
        db = (int64_t)g.Rb * 4096;
        pt = (int64_t)prow * 4096 + k3;
    }
    const int xs = valid ? x : 0;
    const int64_t NN = g.N;
    const int ntiles = gridDim.x;
    const double a = 1.0 - rate;
    const unsigned off_a = (unsigned)t * 4u, off_b = (unsigned)(t + db) * 4u, off_p = (unsigned)pt * 4u;
    const bool adjacent = g.Ra == 1 && !g.npot;                          // (uniform) single-pass power-of-two sizes: the two bins are one 8-byte load
    // the averagers of a multi-pass transform live in PAIR order like the pair sums (spec_state_index): a tile's 64 points are 64 consecutive
    // doubles, not one cache line each (display order: 2^21-point frames moved 2.5 x their magnitudes in state lines per 25-frame batch)
    const int64_t si = g.Ra == 1 ? (int64_t)xs : pt;
    AvgState s0 = {ma[si], maa[si], ma[F + si], maa[F + si]};            // state entering the batch
    int *s_flag = reinterpret_cast<int *>(s_ex_all + (size_t)ng * 2 * kAvgExtFrames * kAvgLanes);   // [2] "this round needs the repairs", by round parity
    if (threadIdx.x == 0) { s_flag[0] = 0; s_flag[1] = 0; }
    __syncthreads();
    int round = 0;
    for (int fb = 0; fb < nf; fb += ng * kAvgGMax) {
        const int nfb = min(ng * kAvgGMax, nf - fb);
        const int G = (nfb + ng - 1) / ng;                               // frames per group (block-uniform)
        const int fg = grp * G;                                           // first frame of my group inside the round (wave-uniform)
        const int cnt = max(0, min(G, nfb - fg));                         // frames this wave really has (wave-uniform)
        // magnitudes of my frames (frames past the end read the round's last frame: loaded, never used; requesting the next round's
        // during the scan was measured and buys nothing: 0.186 ms either way)
        float2 m[kAvgGMax];
#pragma unroll
        for (int i = 0; i < kAvgGMax; ++i) {
            const float *mf = mag + (int64_t)(fb + min(fg + i, nfb - 1)) * NN;      // wave-uniform base
            m[i] = adjacent ? ldf2(mf, off_a) : make_float2(ldf(mf, off_a), ldf(mf, off_b));
        }
        // M^G = [[aG, 0], [cG, aG]] with aG = a^G, cG = G rate a^(G-1)
        double aG = 1.0, aGm1 = 1.0;
        for (int i = 0; i < G; ++i) { aGm1 = aG; aG *= a; }
        const double cG = (double)G * rate * aGm1;
        // 1. local pass from a zero state (a zero state never needs the NaN repairs of avg_step: plain recurrences)
        AvgState loc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < kAvgGMax; ++i)
            if (i < cnt) avg_step_fast(loc, (double)m[i].x, (double)m[i].y, rate);
        s_loc[grp * kAvgLanes + lane] = loc;
        {   // anything not finite among my magnitudes (the zero-state pass has summed them all with positive weights: a NaN or Inf magnitude
            // leaves a NaN or Inf there; x - x is NaN for both), or a NaN state entering the round?
            const bool odd = (s0.ma_a != s0.ma_a) | (s0.maa_a != s0.maa_a) | (s0.ma_b != s0.ma_b) | (s0.maa_b != s0.maa_b)
                           | ((loc.ma_a - loc.ma_a) != 0.0) | ((loc.ma_b - loc.ma_b) != 0.0);
            if (wave_any(odd) && lane == 0) s_flag[round & 1] = 1;
        }
        __syncthreads();
        // 2. entering state of my group
        AvgState s = s0;
        for (int q = 0; q < grp; ++q) {
            const AvgState e = s_loc[q * kAvgLanes + lane];
            const AvgState o = s;
            s.ma_a = aG * o.ma_a + e.ma_a;  s.maa_a = aG * o.maa_a + cG * o.ma_a + e.maa_a;
            s.ma_b = aG * o.ma_b + e.ma_b;  s.maa_b = aG * o.maa_b + cG * o.ma_b + e.maa_b;
        }
        const bool repairs = s_flag[round & 1] != 0;                     // (block-uniform; read between the two barriers of the round)
        if (threadIdx.x == 0) s_flag[(round + 1) & 1] = 0;               // nobody reads or sets the other flag before this round's closing barrier
        if (!repairs) {
            // 3. final pass from the true entering state.  Every magnitude of the round and the entering state are finite: the NaN repairs
            // of :494-497 are no-ops and the blocked scan is the sequential result
#pragma unroll
            for (int i = 0; i < kAvgGMax; ++i) {
                if (i < cnt) {                                                // wave-uniform
                    const int f = fb + fg + i;
                    avg_step_fast(s, (double)m[i].x, (double)m[i].y, rate);
                    const float fa = (float)s.maa_a, fbb = (float)s.maa_b;    // float rounding is monotonic: extrema of the rounded values
                    float mx = 0.f, mn = 3.0e38f;
                    if (valid) {
                        stf(pairsum + (int64_t)f * F, off_p, (float)(s.maa_a + s.maa_b));
                        if (f >= pk_from) maaf[(int64_t)f * F + x] = make_float2(fa, fbb);
                        mx = fmaxf(fa, fbb); mn = fminf(fa, fbb);
                        if (x == 0) first_b[f] = fbb;
                    }
                    s_ex[(i & (kAvgExtFrames - 1)) * kAvgLanes + lane] = mx;
                    s_ex[(kAvgExtFrames + (i & (kAvgExtFrames - 1))) * kAvgLanes + lane] = mn;
                }
                // every four frames (and after the last one): transposed reduction through LDS -- lane = (frame q, sixteenth p) folds four
                // lanes' values in registers, then a 16-lane row reduction: ~10 instructions per frame instead of 36
                if ((i & (kAvgExtFrames - 1)) == kAvgExtFrames - 1 && i - (kAvgExtFrames - 1) < cnt) {
                    wave_sync();
                    const int q = lane >> 4, p16 = lane & 15, fq = i - (kAvgExtFrames - 1) + q;
                    const float4 vx = *reinterpret_cast<const float4 *>(s_ex + q * kAvgLanes + 4 * p16);
                    const float4 vn = *reinterpret_cast<const float4 *>(s_ex + (kAvgExtFrames + q) * kAvgLanes + 4 * p16);
                    float mx = fmaxf(fmaxf(vx.x, vx.y), fmaxf(vx.z, vx.w)), mn = fminf(fminf(vn.x, vn.y), fminf(vn.z, vn.w));
                    row16_max_min(mx, mn);
                    if (p16 == 0 && fq < cnt) ext_w[(int64_t)(fb + fg + fq) * ntiles + blockIdx.x] = make_float2(mx, mn);
                    wave_sync();
                }
            }
            // 4. the group that holds the last frame of the round publishes the state entering the next round
            if (grp == (nfb - 1) / G) s_carry[lane] = s;
            __syncthreads();
        } else {
            // a NaN / Inf magnitude (or a NaN state) somewhere in this tile's round: the repairs make the recurrence non-linear, so the groups
            // take turns in frame order, each entering with the state the previous one left, statement by statement as the reference does
            // (a rare path kept small: magnitudes re-read, the extrema folded by one lane; fmaxf / fminf skip a NaN operand as the reference's
            // comparisons do)
            for (int q = 0; q < ng; ++q) {
                if (grp == q && cnt > 0) {
                    s = (q == 0) ? s0 : s_carry[lane];
#pragma unroll 1
                    for (int i = 0; i < cnt; ++i) {
                        const int f = fb + fg + i;
                        const float *mf = mag + (int64_t)f * NN;
                        const float2 v = adjacent ? ldf2(mf, off_a) : make_float2(ldf(mf, off_a), ldf(mf, off_b));
                        avg_step(s, (double)v.x, (double)v.y, rate);
                        const float fa = (float)s.maa_a, fbb = (float)s.maa_b;
                        float mx = 0.f, mn = 3.0e38f;
                        if (valid) {
                            stf(pairsum + (int64_t)f * F, off_p, (float)(s.maa_a + s.maa_b));
                            if (f >= pk_from) maaf[(int64_t)f * F + x] = make_float2(fa, fbb);
                            mx = fmaxf(fa, fbb); mn = fminf(fa, fbb);
                            if (x == 0) first_b[f] = fbb;
                        }
                        s_ex[lane] = mx; s_ex[kAvgLanes + lane] = mn;
                        wave_sync();
                        if (lane == 0) {
                            for (int l = 1; l < kAvgLanes; ++l) { mx = fmaxf(mx, s_ex[l]); mn = fminf(mn, s_ex[kAvgLanes + l]); }
                            ext_w[(int64_t)f * ntiles + blockIdx.x] = make_float2(mx, mn);
                        }
                        wave_sync();
                    }
                    s_carry[lane] = s;
                }
                __syncthreads();
            }
        }
        s0 = s_carry[lane];
        ++round;
    }
    if (grp == 0 && valid) { ma[si] = s0.ma_a; maa[si] = s0.maa_a; ma[F + si] = s0.ma_b; maa[F + si] = s0.maa_b; }
}

// ---- per-frame extrema over the tiles of spec_average.  grid = frames, 256 threads (1024 when a frame has thousands of tiles: a 2^21-point
// frame has 16384 and a batch only a few dozen frames = workgroups) -------------------------------
constexpr int kExtMaxThreads = 1024;
CSDR_KERNEL_SPEC __launch_bounds__(kExtMaxThreads) void spec_extrema(const float2 *__restrict__ ext_w, int ntiles, float2 *__restrict__ ext) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *s_r = reinterpret_cast<float2 *>(smem);              // [waves]
    const int f = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    float mx = 0.f, mn = 3.0e38f;
    for (int w0 = tid; w0 < ntiles; w0 += 8 * nthr) {               // eight loads in flight (an index past the end re-reads the last tile: extrema do not mind)
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ext_w[(int64_t)f * ntiles + min(w0 + u * nthr, ntiles - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) { mx = fmaxf(mx, v[u].x); mn = fminf(mn, v[u].y); }
    }
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_down(mx, o, 64)); mn = fminf(mn, __shfl_down(mn, o, 64)); }
    if ((tid & 63) == 0) s_r[tid >> 6] = make_float2(mx, mn);
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < (nthr >> 6); ++i) { mx = fmaxf(mx, s_r[i].x); mn = fminf(mn, s_r[i].y); }
        ext[f] = make_float2(mx, mn);
    }
}

// log(1 + u) for u > -1 from the hardware base-2 logarithm: log(w) u / (w - 1) with w = fl(1 + u) cancels the rounding of
// the sum (the classic log1p identity), so the relative error stays at a few float ulps however small u is
__device__ __forceinline__ float log1p_fast(float u) {
    const float w = 1.0f + u;
#if defined(__AMDGCN__)
    const float l = __log2f(w) * 0.69314718055994530942f;             // v_log_f32
#else
    const float l = log2f(w) * 0.69314718055994530942f;
#endif
    const float d = w - 1.0f;
#if defined(__AMDGCN__)
    return d == 0.0f ? u : l * (u * __builtin_amdgcn_rcpf(d));         // (v_rcp_f32, one ulp: the full-precision quotient is ten instructions per point)
#else
    return d == 0.0f ? u : l * (u / d);
#endif
}

struct SpecFrameOut { double point_ceil, point_floor; };
struct SpecScalars { double ceil_ma, ceil_maa, floor_ma, floor_maa; };
struct SpecPeakScalars { double ceil_peak, floor_peak; };

// ---- peak hold (SpectrumVisualProcessor.cpp:247-273, :506-510, :523-530) --------------------------------------------
// reset: fft_result_peak[i] = fft_floor_maa, fft_ceil_peak = fft_floor_maa, fft_floor_peak = fft_ceil_maa (:266-272)
CSDR_KERNEL_SPEC __launch_bounds__(256) void spec_peak_reset(const SpecScalars *__restrict__ st, double *__restrict__ peak, int n2f,
                                                       SpecPeakScalars *__restrict__ pk) {
    const SpecScalars s = *st;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n2f; i += 256 * gridDim.x) peak[i] = s.floor_maa;
    if (blockIdx.x == 0 && threadIdx.x == 0) { pk->ceil_peak = s.floor_maa; pk->floor_peak = s.ceil_maa; }
}
// running maximum of the averaged bins over the frames [pk_from, nf) of a batch, one thread per display point (both of
// its bins); peaksum[f][x] = peak[2x] + peak[2x+1] after frame f, peak_b[f] = the second bin of point 0 (:546-556)
CSDR_KERNEL_SPEC __launch_bounds__(256) void spec_peak_track(const float2 *__restrict__ maaf, int nf, int pk_from, int F,
                                                       double *__restrict__ peak, float *__restrict__ peaksum, float *__restrict__ peak_b,
                                                       float2 *__restrict__ peakf /* zoomed view: both held bins per frame, else null */) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= F) return;
    double pa = peak[x], pb = peak[F + x];
    for (int f = pk_from; f < nf; ++f) {
        const float2 v = maaf[(int64_t)f * F + x];
        if ((double)v.x > pa) pa = (double)v.x;
        if ((double)v.y > pb) pb = (double)v.y;
        peaksum[(int64_t)f * F + x] = (float)(pa + pb);
        if (peakf) peakf[(int64_t)f * F + x] = make_float2((float)pa, (float)pb);
        if (x == 0) peak_b[f] = (float)pb;
    }
    peak[x] = pa; peak[F + x] = pb;
}
// zoomed view: the averagers follow a retune or a zoom step (SpectrumVisualProcessor.cpp:316-331, :454-492).  Display-order
// bin i lives at [(i & 1) F + (i >> 1)] (pair layout).  mode 0/1: memmove left / right by n bins (the vacated end keeps its
// old values); 2: zoom in, dst[i] = src[N/4 + i/2]; 3: zoom out, dst[i] = src[(i - N/4) 2] inside the middle half, else 0.
CSDR_KERNEL_SPEC __launch_bounds__(256) void spec_avg_remap(const double *__restrict__ ma, const double *__restrict__ maa,
                                                      double *__restrict__ ma_o, double *__restrict__ maa_o, int N, int mode, int n, SpecGeom g) {
    const int F = N >> 1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += 256 * gridDim.x) {
        int src = i;
        bool zero = false;
        if (mode == 0) src = i < N - n ? i + n : i;
        else if (mode == 1) src = i >= n ? i - n : i;
        else if (mode == 2) src = N / 4 + i / 2;
        else {
            zero = i < N / 4 || i >= N - N / 4; src = zero ? 0 : (i - N / 4) * 2;
            if (src >= N) { zero = true; src = 0; }      // N not a multiple of 4 (an odd fftSize): the reference reads one element past its vector here (:471); a zero instead
        }
        const int64_t so = (int64_t)(src & 1) * F + spec_state_index(g, src >> 1), dn = (int64_t)(i & 1) * F + spec_state_index(g, i >> 1);
        ma_o[dn] = zero ? 0.0 : ma[so];
        maa_o[dn] = zero ? 0.0 : maa[so];
    }
}

// the four trackers frame by frame (the reference's statements, :513-521) and their held extremes (:523-530) for the
// frames [pk_from, nf): pfo[f] = {fft_ceil_peak, fft_floor_peak} after frame f.  One thread: nf short double recurrences.
CSDR_KERNEL_SPEC void spec_peak_trackers(const float2 *__restrict__ ext, int nf, int pk_from, const SpecScalars *__restrict__ st_in,
                                   SpecPeakScalars *__restrict__ pk, SpecFrameOut *__restrict__ pfo) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    SpecScalars s = *st_in;
    SpecPeakScalars p = *pk;
    for (int f = 0; f < nf; ++f) {
        const float2 e = ext[f];
        float fft_ceil = 0.f, fft_floor = 1.f;
        if (e.x > fft_ceil) fft_ceil = e.x;
        if (e.y < fft_floor) fft_floor = e.y;
        s.ceil_ma = s.ceil_ma + ((double)fft_ceil - s.ceil_ma) * 0.05;
        s.ceil_maa = s.ceil_maa + (s.ceil_ma - s.ceil_maa) * 0.05;
        s.floor_ma = s.floor_ma + ((double)fft_floor - s.floor_ma) * 0.05;
        s.floor_maa = s.floor_maa + (s.floor_ma - s.floor_maa) * 0.05;
        if (f >= pk_from) {
            if (s.ceil_maa > p.ceil_peak) p.ceil_peak = s.ceil_maa;
            if (s.floor_maa < p.floor_peak) p.floor_peak = s.floor_maa;
            pfo[f].point_ceil = p.ceil_peak; pfo[f].point_floor = p.floor_peak;
        }
    }
    *pk = p;
}
constexpr int kDispThreads = 256;
constexpr int kDispTile = 2 * kDispThreads;   // display points per step of the plain path
// the transposing path: a tile is W = U x 256 consecutive display points = W / npairs consecutive k3 of EVERY row pair.  W = 2048 up to 64 row
// pairs; 32 k3 per row pair (128-byte runs on the read side) beyond: 4096 / 8192 points for 128 / 256 row pairs (2^20 / 2^21-point frames --
// with 512-point tiles their runs were 16 / 8 bytes: 0.103 ms per C5 batch)
constexpr int kDispRun = 32;
__host__ __device__ constexpr int disp_tile_points(int npairs) { return npairs * kDispRun > 2048 ? npairs * kDispRun : 2048; }
__host__ __device__ constexpr int disp_lg_pad(int lg_npairs) { return lg_npairs > 4 ? lg_npairs : 4; }      // one float of padding per max(16, npairs): a wave's stores are npairs floats apart
__host__ __device__ constexpr size_t disp_lds_bytes(int npairs, int lg_npairs, bool hold) {
    return (size_t)(hold ? 2 : 1) * (size_t)(disp_tile_points(npairs) + (disp_tile_points(npairs) >> disp_lg_pad(lg_npairs))) * sizeof(float);
}
constexpr int kDispMaxPairs = 256;
constexpr size_t kDispLdsPlain = 16;
constexpr size_t kTrackLds = 4 * (kDispThreads / 64) * sizeof(double);

struct SpecFrameScal { double pc, pf, fl; };  // point_ceil, point_floor, fft_floor_maa of a frame

// ---- floor / ceil trackers of every frame of the batch.  grid = frames, 256 threads --------------------------------
// The trackers (SpectrumVisualProcessor.cpp:513-521) are short linear recurrences over the frames; workgroup f evaluates them in
// closed form from the batch-entering state up to its own frame (a = 0.95, b = 0.05; c_i = float ceiling, d_i = float floor of
// frame i):      ma_f  = a^(f+1) ma_in + b sum_i a^(f-i) c_i
//                maa_f = a^(f+1) maa_in + b (f+1) a^(f+1) ma_in + b^2 sum_i (f-i+1) a^(f-i) c_i      (maa uses the NEW ma)
// as parallel weighted sums over i <= f (double; equal to the serial loop to ~1e-15 relative); the last frame publishes the end
// state (ping-pong copy).
// (ext_w != nullptr -- short batches, the one-block call of the real-time shape: the per-frame extrema are formed HERE from the averaging pass's
//  per-tile values, every workgroup for the frames up to its own, into LDS behind the reduction scratch: one launch less in a chain of 5 us launches)
constexpr int kTrackSmallFrames = 32;
CSDR_KERNEL_SPEC __launch_bounds__(kDispThreads) void spec_trackers(const float2 *__restrict__ ext, int nf, const SpecScalars *__restrict__ st_in,
                                                              SpecScalars *__restrict__ st_out, SpecFrameOut *__restrict__ fo,
                                                              SpecFrameScal *__restrict__ fsc, int pk_from, const SpecFrameOut *__restrict__ pfo,
                                                              const float2 *__restrict__ ext_w, int ntiles, float2 *__restrict__ ext_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *s_sum = reinterpret_cast<double *>(smem);          // reduction scratch [waves][4]
    const int f = blockIdx.x, tid = threadIdx.x;
    const SpecScalars s_in = *st_in;
    if (ext_w) {
        float2 *s_ext = reinterpret_cast<float2 *>(smem + kTrackLds);      // [nf <= kTrackSmallFrames]
        for (int i = tid >> 6; i <= f; i += kDispThreads / 64) {            // a wave per frame (spec_extrema's arithmetic: max from 0, min from 3e38)
            float mx = 0.f, mn = 3.0e38f;
            for (int w0 = tid & 63; w0 < ntiles; w0 += 64) { const float2 v = ext_w[(int64_t)i * ntiles + w0]; mx = fmaxf(mx, v.x); mn = fminf(mn, v.y); }
            for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_down(mx, o, 64)); mn = fminf(mn, __shfl_down(mn, o, 64)); }
            if ((tid & 63) == 0) { s_ext[i] = make_float2(mx, mn); if (i == f && ext_out) ext_out[f] = make_float2(mx, mn); }
        }
        __syncthreads();
        ext = s_ext;
    }
    double w_c = 0.0, w_c2 = 0.0, w_d = 0.0, w_d2 = 0.0;
    for (int i = tid; i <= f; i += kDispThreads) {
        const float2 e = ext[i];
        float fft_ceil = 0.f, fft_floor = 1.f;              // the reference keeps these two in float (:436)
        if (e.x > fft_ceil) fft_ceil = e.x;
        if (e.y < fft_floor) fft_floor = e.y;
        const double w = dc_pow(0.95, f - i), w2 = (double)(f - i + 1) * w;
        w_c += w * (double)fft_ceil; w_c2 += w2 * (double)fft_ceil;
        w_d += w * (double)fft_floor; w_d2 += w2 * (double)fft_floor;
    }
    for (int o = 32; o > 0; o >>= 1) {
        w_c += __shfl_down(w_c, o, 64); w_c2 += __shfl_down(w_c2, o, 64);
        w_d += __shfl_down(w_d, o, 64); w_d2 += __shfl_down(w_d2, o, 64);
    }
    if ((tid & 63) == 0) { double *q = s_sum + 4 * (tid >> 6); q[0] = w_c; q[1] = w_c2; q[2] = w_d; q[3] = w_d2; }
    __syncthreads();
    if (tid == 0) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        for (int q = 0; q < kDispThreads / 64; ++q) { t0 += s_sum[4 * q]; t1 += s_sum[4 * q + 1]; t2 += s_sum[4 * q + 2]; t3 += s_sum[4 * q + 3]; }
        const double af = dc_pow(0.95, f + 1), b = 0.05;
        SpecScalars s;
        s.ceil_ma = af * s_in.ceil_ma + b * t0;
        s.ceil_maa = af * s_in.ceil_maa + b * (double)(f + 1) * af * s_in.ceil_ma + b * b * t1;
        s.floor_ma = af * s_in.floor_ma + b * t2;
        s.floor_maa = af * s_in.floor_maa + b * (double)(f + 1) * af * s_in.floor_ma + b * b * t3;
        // point_ceil / point_floor: the held extremes while peak hold is live (:539-540)
        const bool hold = f >= pk_from;
        SpecFrameScal o;
        o.pc = hold ? pfo[f].point_ceil : s.ceil_maa; o.pf = hold ? pfo[f].point_floor : s.floor_maa; o.fl = s.floor_maa;
        fsc[f] = o;
        fo[f].point_ceil = o.pc; fo[f].point_floor = o.pf;
        if (f == nf - 1) *st_out = s;
    }
}

// the transposing path of spec_display for one frame.  A tile of W points is read in passes of 2048 (eight loads per thread in flight together:
// all of a wide tile's at once cost 16 registers a point -- 482 at W = 8192), formed, dropped at its display position in LDS; one barrier; written out
constexpr int kDispPass = 8;                  // points per thread and pass
__device__ __forceinline__ void spec_display_tiles(const float *__restrict__ pairsum, const float *__restrict__ first_b, const SpecGeom &g, float sf,
                                                   float *__restrict__ points, bool hold, const float *__restrict__ peaksum, const float *__restrict__ peak_b,
                                                   float *__restrict__ hold_points, int f, double pf, double fl, float inv_den, int npairs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = disp_tile_points(npairs);
    const int tid = threadIdx.x, F = g.F;
    const int lg_np = g.lgRa - 1 + g.lgRb, lg_pad = disp_lg_pad(lg_np);
    float *s_y = reinterpret_cast<float *>(smem);                    // [W + W >> lg_pad] y, then the same for the held y
    float *s_h = s_y + W + (W >> lg_pad);
    const int nk3 = W >> lg_np, lg_nk3 = 31 - __clz(nk3), lg_half = g.lgRa - 1;
    const int ntile = F / W;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int k3base = tile * nk3;
        const int x0 = (npairs * k3base - (g.N >> 2)) & (F - 1);     // display point of (row pair 0, k3base); tiles do not wrap (nk3 divides 2048)
#pragma unroll 1
        for (int i0 = 0; i0 < W; i0 += kDispPass * kDispThreads) {
            float a[kDispPass], ph[kDispPass];
#pragma unroll
            for (int u = 0; u < kDispPass; ++u) {
                const int i = i0 + tid + u * kDispThreads;
                const int prow = i >> lg_nk3, k3i = i & (nk3 - 1);
                a[u] = pairsum[(int64_t)f * F + (int64_t)prow * 4096 + k3base + k3i];
                if (hold) {
                    const int k1h = prow >> g.lgRb, k2 = prow & (g.Rb - 1);
                    ph[u] = peaksum[(int64_t)f * F + x0 + k3i * npairs + k1h + (k2 << lg_half)];
                }
            }
#pragma unroll
            for (int u = 0; u < kDispPass; ++u) {
                const int i = i0 + tid + u * kDispThreads;
                const int prow = i >> lg_nk3, k3i = i & (nk3 - 1);
                const int k1h = prow >> g.lgRb, k2 = prow & (g.Rb - 1);
                const int pos = k3i * npairs + k1h + (k2 << lg_half);    // position inside the tile, display order
                const int x = x0 + pos;
                const double acc = (x == 0) ? fl + (double)first_b[f] : (double)a[u];      // idx == 0 is replaced by fft_floor_maa (:546-556)
                s_y[pos + (pos >> lg_pad)] = log1p_fast((float)(acc * 0.5 - pf)) * inv_den * sf;
                if (hold) {
                    const double pacc = (x == 0) ? fl + (double)peak_b[f] : (double)ph[u];
                    s_h[pos + (pos >> lg_pad)] = log1p_fast((float)(pacc * 0.5 - pf)) * inv_den * sf;
                }
            }
        }
        __syncthreads();
        for (int j = 2 * tid; j < W; j += 2 * kDispThreads) {
            const int q = j + (j >> lg_pad);
            // only y is stored: the x of point i is i / F for every frame (:562) and is filled in when a frame is fetched
            st_stream(reinterpret_cast<float2 *>(points + (int64_t)f * F + x0 + j), make_float2(s_y[q], s_y[q + 1]));
            if (hold) *reinterpret_cast<float2 *>(hold_points + (int64_t)f * F + x0 + j) = make_float2(s_h[q], s_h[q + 1]);
        }
        __syncthreads();
    }
}

// ---- K16.  Two display points per thread, full-span view (visualRatio = 1: two bins per point), :532-576:
//     y = log10(acc / 2 + 0.25 - (floor - 0.75)) / log10(ceil + 0.25 - (floor - 0.75)) * scale
// Both arguments are 1 + u with u formed in double; the logarithms are taken as log(1 + u) with the rounding of the sum
// cancelled (log1p_fast: a few float ulps also when the dynamic range is tiny), their ratio needs no base conversion.
// grid = (column blocks, frames): the transposing path takes one tile per workgroup (grid.x = F / W or fewer: it strides), the plain path
// 2 x 256 points per step.
// TILES: the transposing path (full-span view behind a multi-pass transform) or the plain one -- two kernels, the registers of one are not the other's
template <bool TILES>
CSDR_KERNEL __launch_bounds__(kDispThreads) void spec_display(const float *__restrict__ pairsum /* pair order: spec_pair_index */, const float *__restrict__ first_b,
                                                             const SpecFrameScal *__restrict__ fsc, SpecGeom g, float sf, float *__restrict__ points,
                                                             int pk_from, const float *__restrict__ peaksum,
                                                             const float *__restrict__ peak_b, float *__restrict__ hold_points,
                                                             const int2 *__restrict__ vmap /* zoomed view: (first bin, bins) per point, else null */,
                                                             const float2 *__restrict__ maaf, const float2 *__restrict__ peakf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int f = blockIdx.y, tid = threadIdx.x, F = g.F;
    const SpecFrameScal sc = fsc[f];
    const double pc = sc.pc, pf = sc.pf, fl = sc.fl;
    const bool hold = f >= pk_from;
    const float inv_den = 1.0f / log1pf((float)(pc - pf));          // (pc + 0.25) - (pf - 0.75) = 1 + (pc - pf)
    // Full-span view of a multi-pass transform: the pair sums lie in PAIR order (spec_pair_index).  A tile of W consecutive display
    // points = nk3 consecutive k3 of every row pair: the threads read it row by row (runs of nk3 floats), form y, drop it at its display
    // position in LDS (padded: a wave's stores are npairs floats apart) and write the tile out in display order.  All loads of a tile are
    // issued before the first value is used.
    const int npairs = g.Ra == 1 ? 1 : (g.Ra >> 1) * g.Rb;           // row pairs = display points per k3
    if constexpr (TILES) {
        spec_display_tiles(pairsum, first_b, g, sf, points, hold, peaksum, peak_b, hold_points, f, pf, fl, inv_den, npairs);
        return;
    }
    for (int x0 = 2 * (blockIdx.x * kDispThreads + tid); x0 < F; x0 += 2 * kDispThreads * gridDim.x) {
    float y[2], yh[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int x = x0 + u;
        double acc = 0.0, pacc = 0.0, inv_n = 0.5;
        if (vmap) {
            // zoomed view (:532-560): visualRatio = bandwidth / resampleBw in (0.5, 1] -> one or two bins per point, walked on
            // the host with the reference's double accumulator; bins outside (0, N) read as fft_floor_maa
            if (x < F) {
                const int2 m = vmap[x];
                const float *row = reinterpret_cast<const float *>(maaf + (int64_t)f * F);
                const float *prow = hold ? reinterpret_cast<const float *>(peakf + (int64_t)f * F) : nullptr;
                const int N = 2 * F;
                for (int k = 0; k < m.y; ++k) {
                    const int idx = m.x + k;
                    const bool in = idx > 0 && idx < N;
                    acc += in ? (double)row[idx] : fl;
                    if (hold) pacc += in ? (double)prow[idx] : fl;
                }
                inv_n = 1.0 / (double)max(m.y, 1);
            }
        } else {
            if (x < F) acc = (x == 0) ? fl + (double)first_b[f]   // idx == 0 is replaced by fft_floor_maa (:546-556)
                                      : (double)pairsum[(int64_t)f * F + spec_pair_index(g, x)];
            if (hold && x < F) pacc = (x == 0) ? fl + (double)peak_b[f] : (double)peaksum[(int64_t)f * F + x];
        }
        y[u] = log1p_fast((float)(acc * inv_n - pf)) * inv_den * sf;  // acc / n + 0.25 - (pf - 0.75) = 1 + (acc / n - pf)
        if (hold) yh[u] = log1p_fast((float)(pacc * inv_n - pf)) * inv_den * sf;
    }
    float *o = points + (int64_t)f * F + x0;
    if (x0 + 1 < F) *reinterpret_cast<float2 *>(o) = make_float2(y[0], y[1]);
    else o[0] = y[0];
    if (hold) {
        float *h = hold_points + (int64_t)f * F + x0;
        if (x0 + 1 < F) *reinterpret_cast<float2 *>(h) = make_float2(yh[0], yh[1]);
        else h[0] = yh[0];
    }
    }
}

}  // namespace csdr
