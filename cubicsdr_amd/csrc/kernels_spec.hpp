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
constexpr int kRowLdsPts = 16 * 272;       // float2 slots of the 4096-point row FFT's exchange buffer

struct SpecGeom {
    int N, F;                 // internal FFT size, display points (= N / 2)
    int Ra, Rb, lgRa, lgRb;   // radix passes in front of the 4096-point rows (1 = absent)
    int N2;                   // row length: 4096 when N >= 4096, else N
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
    __syncthreads();                                           // previous users of the exchange buffer are done
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) lds[tid + 272 * k2] = v[k2];          // [k2][n1][n0], rows padded to 272
    __syncthreads();
    // stage 2: thread (n0, k2 = hi): DFT over n1, times W4096^(n0 (16 k1 + k2))
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = lds[n0 + 16 * n1 + 272 * hi];
    __syncthreads();
    dft_reg<16>(v);
    {
        const int b = 16 * n0;
        leaf[0] = tw4096[b]; leaf[1] = tw4096[2 * b]; leaf[2] = tw4096[4 * b]; leaf[3] = tw4096[8 * b]; leaf[4] = leaf[3];
        twiddle_powers<16>(v, leaf);
        const float2 b0 = tw4096[n0 * hi];
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) v[k1] = cmul(v[k1], b0);
    }
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) lds[hi + 16 * k1 + 257 * n0] = v[k1];  // [n0][k1][k2], rows padded to 257
    __syncthreads();
    // stage 3: thread tid = k2 + 16 k1: DFT over n0
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = lds[tid + 257 * m];
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

struct FrameSrc {             // where sequence f starts: sequence 0 may live in the carry buffer
    const float2 *first;      // sequence 0
    const float2 *rest;       // sequence f >= 1 starts at rest + (f - 1) * stride
    int64_t stride;
};
__device__ inline const float2 *frame_ptr(const FrameSrc &fs, int f) { return f == 0 ? fs.first : fs.rest + (int64_t)(f - 1) * fs.stride; }

__device__ inline float cabs_f(float2 v) { return sqrtf(v.x * v.x + v.y * v.y); }

// ---- radix pass: R-point column DFTs over stride Lr = L / R, times W_L^(k c); grid = (Lr / (256 COLS), sequences) --
// W_L^q = exp(-2 pi i q tw_scale / N) is looked up in the split tables of the full transform.
template <int R, int COLS>
__global__ __launch_bounds__(kFftThreads) void spec_fft_radix(FrameSrc fs, int L, unsigned tw_scale,
                                                              const float2 *__restrict__ tw_hi, const float2 *__restrict__ tw_lo,
                                                              float2 *dst) {
    const int Lr = L / R;
    const int c = COLS * (blockIdx.x * kFftThreads + threadIdx.x);
    if (c >= Lr) return;
    const int s = blockIdx.y;
    const float2 *x = frame_ptr(fs, s) + c;
    float2 *o = dst + (int64_t)s * L + c;
    float2 a[R], b[R];
    if (COLS == 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const f4u v = *reinterpret_cast<const f4u *>(x + (int64_t)r * Lr);
            a[r] = make_float2(v.x, v.y); b[r] = make_float2(v.z, v.w);
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = x[(int64_t)r * Lr];
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
        for (int k = 0; k < R; ++k) o[(int64_t)k * Lr] = a[k];
    }
}

// ---- 4096-point row FFTs + magnitude.  grid = (rows = Ra Rb, frames) ---------------------------------------------
// row r = k1 Rb + k2 of frame f in `fs` ([f][r][4096]; the frame itself when there is a single row) holds the bins
// k = k1 + Ra (k2 + Rb k3); |X| is stored as mag[f][r][k3] (float), i.e. natural bin order when there is one row.
__global__ __launch_bounds__(kFftThreads) void spec_fft_rows4096(FrameSrc fs, SpecGeom g, const float2 *__restrict__ tw4096,
                                                                 float *__restrict__ mag, float2 *__restrict__ raw_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *lds = reinterpret_cast<float2 *>(smem);
    const int f = blockIdx.y, row = blockIdx.x, tid = threadIdx.x;
    const float2 *x = frame_ptr(fs, f) + (int64_t)row * 4096;
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = x[tid + 256 * r];
    fft4096_regs(v, lds, tw4096);
    if (mag) {
        float *o = mag + (int64_t)f * g.N + (int64_t)row * 4096;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[tid + 256 * r] = cabs_f(v[r]);
    }
    if (raw_out) {   // natural-order complex output (parity tests of K13 alone)
        const int k1 = row >> g.lgRb, k2 = row & (g.Rb - 1);
        const int64_t rb = k1 + (int64_t)g.Ra * k2, rs = (int64_t)g.Ra * g.Rb;
#pragma unroll
        for (int r = 0; r < 16; ++r) raw_out[(int64_t)f * g.N + rb + rs * (tid + 256 * r)] = v[r];
    }
}

// ---- small transforms (N <= 2048): one frame per workgroup, Stockham through LDS.  grid = (1, frames) ----------
__global__ __launch_bounds__(kFftThreads) void spec_fft_small(FrameSrc fs, int N, const float2 *__restrict__ tw4096,
                                                              float *__restrict__ mag, float2 *__restrict__ raw_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *sa = reinterpret_cast<float2 *>(smem), *sb = sa + N;
    const int f = blockIdx.y, tid = threadIdx.x;
    const float2 *x = frame_ptr(fs, f);
    for (int i = tid; i < N; i += kFftThreads) sa[i] = x[i];
    __syncthreads();
    const float2 *r = lds_fft(sa, sb, N, tw4096);
    if (mag) {
        float *o = mag + (int64_t)f * N;
        for (int i = tid; i < N; i += kFftThreads) o[i] = cabs_f(r[i]);
    }
    if (raw_out) for (int i = tid; i < N; i += kFftThreads) raw_out[(int64_t)f * N + i] = r[i];
}

// ---- K15: averaging recurrences, display order -----------------------------------------------------------------
// thread x owns display point x = the two adjacent shifted bins ka = (2 x + N / 2) mod N and ka + 1; ma / maa
// (fft_result_ma / _maa, double) are kept per pair as [2][F] arrays.  One wave per workgroup so the F pairs spread
// over all CUs; the magnitudes of kAvgU frames are loaded ahead of the serial recurrence.  The per-frame extrema the
// reference tracks are (float) max / min of maa; float rounding is monotonic, so each thread keeps (float max, float
// min) per frame and the wave reduces kAvgU frames at a time -- nothing cross-lane inside the serial chain.
constexpr int kAvgThreads = 64;
constexpr int kAvgU = 8;

// offsets (in floats, inside one frame of `mag`) of the two bins of display point x; db = distance between them
__device__ inline int64_t spec_pair_offset(const SpecGeom &g, int x, int64_t &db) {
    const int ka = (2 * x + g.N / 2) & (g.N - 1);
    if (g.Ra == 1) { db = 1; return ka; }
    const int k1 = ka & (g.Ra - 1), rest = ka >> g.lgRa;
    const int k2 = rest & (g.Rb - 1), k3 = rest >> g.lgRb;
    db = (int64_t)g.Rb * 4096;                                        // bin ka + 1 lives in row (k1 + 1) Rb + k2
    return ((int64_t)k1 * g.Rb + k2) * 4096 + k3;
}
__device__ inline float2 spec_load_pair(const float *__restrict__ mag, int64_t off, int64_t db) {
    if (db == 1) return *reinterpret_cast<const float2 *>(mag + off);   // ka is even: 8-byte aligned
    return make_float2(mag[off], mag[off + db]);
}

__global__ __launch_bounds__(kAvgThreads) void spec_average(const float *__restrict__ mag, int nf, SpecGeom g, double rate,
                                                            double *__restrict__ ma, double *__restrict__ maa,
                                                            float *__restrict__ pairsum, float *__restrict__ first_b,
                                                            float2 *__restrict__ ext_w) {
    const int F = g.F;
    const int x = blockIdx.x * kAvgThreads + threadIdx.x;
    const bool valid = x < F;
    const int xs = valid ? x : 0;
    int64_t db;
    const int64_t t = spec_pair_offset(g, xs, db);
    const int64_t NN = g.N;
    const int nwaves = gridDim.x;
    double ma_a = ma[xs], ma_b = ma[F + xs], maa_a = maa[xs], maa_b = maa[F + xs];
    float2 cur[kAvgU], nxt[kAvgU];
#pragma unroll
    for (int u = 0; u < kAvgU; ++u) cur[u] = (u < nf) ? spec_load_pair(mag, (int64_t)u * NN + t, db) : make_float2(0.f, 0.f);
    for (int f0 = 0; f0 < nf; f0 += kAvgU) {
#pragma unroll
        for (int u = 0; u < kAvgU; ++u) nxt[u] = (f0 + kAvgU + u < nf) ? spec_load_pair(mag, (int64_t)(f0 + kAvgU + u) * NN + t, db) : make_float2(0.f, 0.f);
        float mx[kAvgU], mn[kAvgU];
#pragma unroll
        for (int u = 0; u < kAvgU; ++u) {
            const int f = f0 + u;
            mx[u] = 0.f; mn[u] = 3.0e38f;
            if (f < nf) {
                const double xa = (double)cur[u].x, xb = (double)cur[u].y;
                if (maa_a != maa_a) maa_a = xa;
                maa_a += (ma_a - maa_a) * rate;
                if (ma_a != ma_a) ma_a = xa;
                ma_a += (xa - ma_a) * rate;
                if (maa_b != maa_b) maa_b = xb;
                maa_b += (ma_b - maa_b) * rate;
                if (ma_b != ma_b) ma_b = xb;
                ma_b += (xb - ma_b) * rate;
                if (valid) {
                    pairsum[(int64_t)f * F + x] = (float)(maa_a + maa_b);
                    mx[u] = (float)fmax(maa_a, maa_b); mn[u] = (float)fmin(maa_a, maa_b);
                    if (x == 0) first_b[f] = (float)maa_b;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kAvgU; ++u) {
            float a = mx[u], b = mn[u];
            for (int o = 32; o > 0; o >>= 1) { a = fmaxf(a, __shfl_down(a, o, 64)); b = fminf(b, __shfl_down(b, o, 64)); }
            if (threadIdx.x == 0 && f0 + u < nf) ext_w[(int64_t)(f0 + u) * nwaves + blockIdx.x] = make_float2(a, b);
            cur[u] = nxt[u];
        }
    }
    if (valid) { ma[x] = ma_a; ma[F + x] = ma_b; maa[x] = maa_a; maa[F + x] = maa_b; }
}

// ---- floor / ceil trackers across the frames of the batch (SpectrumVisualProcessor.cpp:494-521) ------------------
// one workgroup: the waves reduce the per-wave extrema of each frame, then thread 0 runs the short serial recurrences.
struct SpecFrameOut { double point_ceil, point_floor; };
struct SpecScalars { double ceil_ma, ceil_maa, floor_ma, floor_maa; };
constexpr int kTrackThreads = 1024;
constexpr int kTrackChunk = 2048;          // frames staged in LDS per round

__global__ __launch_bounds__(kTrackThreads) void spec_trackers(const float2 *__restrict__ ext_w, int nwaves, int nf,
                                                               SpecScalars *st, SpecFrameOut *fo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *s_ext = reinterpret_cast<float2 *>(smem);          // [kTrackChunk] (max, min) per frame
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SpecScalars s = *st;
    for (int fc = 0; fc < nf; fc += kTrackChunk) {
        const int cn = min(kTrackChunk, nf - fc);
        for (int i0 = 0; i0 < cn; i0 += kTrackThreads / 64) {     // block-uniform trip count
            const int i = i0 + wave;
            float mx = 0.f, mn = 3.0e38f;
            if (i < cn) for (int w = lane; w < nwaves; w += 64) { const float2 v = ext_w[(int64_t)(fc + i) * nwaves + w]; mx = fmaxf(mx, v.x); mn = fminf(mn, v.y); }
            for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_down(mx, o, 64)); mn = fminf(mn, __shfl_down(mn, o, 64)); }
            if (lane == 0 && i < cn) s_ext[i] = make_float2(mx, mn);
        }
        __syncthreads();
        if (tid == 0) {
            for (int i = 0; i < cn; ++i) {
                const float mx = s_ext[i].x, mn = s_ext[i].y;
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
                fo[fc + i].point_ceil = s.ceil_maa;
                fo[fc + i].point_floor = s.floor_maa;
            }
        }
        __syncthreads();
    }
    if (tid == 0) *st = s;
}

// ---- K16: display points, full-span view (visualRatio = 1: two bins per point).  grid = (F / 512, frames) ---------
__global__ __launch_bounds__(256) void spec_display(const float *__restrict__ pairsum, const float *__restrict__ first_b,
                                                    const SpecFrameOut *__restrict__ fo, int F, float sf,
                                                    float *__restrict__ points) {
    const int x0 = 2 * (blockIdx.x * blockDim.x + threadIdx.x), f = blockIdx.y;
    if (x0 >= F) return;
    const double pc = fo[f].point_ceil, pf = fo[f].point_floor;
    const double den = log10((pc + 0.25) - (pf - 0.75));
    float y[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int x = x0 + u;
        double acc = 0.0;
        if (x < F) acc = (x == 0) ? pf + (double)first_b[f]       // idx == 0 is replaced by fft_floor_maa (:546-556)
                                  : (double)pairsum[(int64_t)f * F + x];
        y[u] = (float)((log10((acc / 2.0) + 0.25 - (pf - 0.75)) / den) * (double)sf);
    }
    float *o = points + ((int64_t)f * F + x0) * 2;
    if (x0 + 1 < F) *reinterpret_cast<float4 *>(o) = make_float4((float)x0 / (float)F, y[0], (float)(x0 + 1) / (float)F, y[1]);
    else { o[0] = (float)x0 / (float)F; o[1] = y[0]; }
}

// assemble frame 0 of a contiguous run from (carry ++ head of the new data)
__global__ void spec_assemble(const float2 *carry, int ncarry, const float2 *x, int n, float2 *dst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = i < ncarry ? carry[i] : x[i - ncarry];
}

}  // namespace csdr
