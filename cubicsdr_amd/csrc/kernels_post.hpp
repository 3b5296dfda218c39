// kernels_post.hpp -- SDRPostThread arithmetic on the GPU (K1 DC blocker, K2 polyphase channelizer, K4 de-interleave).
//
// Replaces (reference file:line): iirfilt_crcf_execute_block SDRPostThread.cpp:284,375 (DC blocker, liquid
// iirfilt_crcf_create_dc_blocker(0.0005) :29), firpfbch_crcf_analyzer_execute :449-451 (liquid firpfbch, Kaiser
// prototype m=4, As=60 :406) and the strided channel gather :364-381.
//
// All LDS is dynamic (`smem`, 16-byte aligned base, carve offsets multiples of 16).
#pragma once
#include <type_traits>
#include "common.hpp"

namespace csdr {

// ------------------------------------------------------------------------------------------------------------
// K1: first-order DC blocker  v[n] = x[n] - a1 v[n-1];  y[n] = v[n] - v[n-1]   (direct form II, b={1,-1}, a={1,a1})
// A linear recurrence: evaluated as a two-kernel blocked affine scan in fp64 (tile-local ends; then every tile
// derives its entering state from the ends of the tiles before it and applies), so a stream of any length runs in
// parallel while the carried state (one complex v) stays exact to fp64 rounding.
// Tile = 256 threads x 16 samples, staged through LDS so global accesses stay coalesced.
// ------------------------------------------------------------------------------------------------------------
constexpr int kDcSeg = 16;
constexpr int kDcThreads = 256;
constexpr int kDcTile = kDcSeg * kDcThreads;
constexpr size_t kDcLds = kDcTile * sizeof(float2) + kDcThreads * (2 * sizeof(double) + sizeof(double));

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

__device__ inline void dc_stage_in(const float2 *__restrict__ x, int64_t n, int64_t base, float2 *sx) {
    // 16-byte loads: two samples per lane (the block start is only guaranteed 8-byte aligned)
    for (int i = threadIdx.x; i < kDcTile / 2; i += kDcThreads) {
        const int64_t g = base + 2 * (int64_t)i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g + 1 < n) { const f4u t = *reinterpret_cast<const f4u *>(x + g); v = make_float4(t.x, t.y, t.z, t.w); }
        else if (g < n) { const float2 s = x[g]; v.x = s.x; v.y = s.y; }
        sx[2 * i] = make_float2(v.x, v.y);
        sx[2 * i + 1] = make_float2(v.z, v.w);
    }
}

// pass 1: each tile computes its end value assuming zero entering state
CSDR_KERNEL_POST __launch_bounds__(kDcThreads) void dc_tile_ends(const float2 *__restrict__ x, int64_t n, double c, d2 *__restrict__ tile_end) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *sx = reinterpret_cast<float2 *>(smem);
    d2 *sb = reinterpret_cast<d2 *>(smem + kDcTile * sizeof(float2));
    double *sa = reinterpret_cast<double *>(smem + kDcTile * sizeof(float2) + kDcThreads * sizeof(d2));
    const int64_t base = (int64_t)blockIdx.x * kDcTile;
    dc_stage_in(x, n, base, sx);
    __syncthreads();
    d2 v = {0.0, 0.0};
    for (int i = 0; i < kDcSeg; ++i) {
        int64_t g = base + threadIdx.x * kDcSeg + i;
        if (g < n) { float2 s = sx[threadIdx.x * kDcSeg + i]; v.x = (double)s.x + c * v.x; v.y = (double)s.y + c * v.y; }
    }
    // partial segments occur only in the last tile, whose end value is never consumed (the carried state is recomputed
    // exactly by dc_apply), so a uniform per-thread multiplier is sufficient here.
    d2 total;
    (void)dc_block_scan(v, dc_pow(c, kDcSeg), d2{0.0, 0.0}, sb, sa, &total);
    if (threadIdx.x == 0) tile_end[blockIdx.x] = total;
}

// inclusive scan of V[t+1] = a V[t] + b[t] over the 256 threads of the workgroup (same multiplier `a` for every
// thread): Kogge-Stone inside each wave with shuffles (multipliers a^(2^k)), wave totals through LDS.
// Returns the value ENTERING thread t's segment, given v_in entering thread 0.
__device__ inline d2 dc_scan256(d2 b, double a, d2 v_in, d2 *lds4) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    d2 cur = b;
    double ap = a;                                           // a^(2^k)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double ox = __shfl_up(cur.x, 1u << k, 64), oy = __shfl_up(cur.y, 1u << k, 64);
        if (lane >= (1 << k)) { cur.x += ap * ox; cur.y += ap * oy; }
        ap *= ap;
    }
    // ap == a^64 now; cur = inclusive value of this thread inside its wave (zero entering state)
    if (lane == 63) lds4[w] = cur;
    double px = __shfl_up(cur.x, 1, 64), py = __shfl_up(cur.y, 1, 64);   // exclusive inside the wave
    if (lane == 0) { px = 0.0; py = 0.0; }
    __syncthreads();
    d2 pre = v_in;                                           // value entering this wave
    for (int u = 0; u < w; ++u) { const d2 t = lds4[u]; pre.x = ap * pre.x + t.x; pre.y = ap * pre.y + t.y; }
    const double al = dc_pow(a, lane);
    __syncthreads();
    return d2{al * pre.x + px, al * pre.y + py};
}

// pass 2: block = kDcTile consecutive samples.  The entering state of the block comes from the carried state and the
// end values e_i (zero entering state) of the mini-tiles of `tile_len` samples before it:
//     v_in(n0) = c^n0 state + sum_{i < n0 / tile_len} c^(n0 - (i+1) tile_len) e_i
// (terms older than ~80000 samples are below 1e-17 of the newest and are dropped), then the recurrence is re-run with
// that state and y is written (in place allowed); the block holding the last sample stores the new carried state.
CSDR_KERNEL_POST __launch_bounds__(kDcThreads) void dc_apply(const float2 *x, float2 *y, int64_t n, double c, int tile_len,
                                                       const d2 *__restrict__ tile_end, const d2 *__restrict__ state_in,
                                                       d2 *__restrict__ state_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *sx = reinterpret_cast<float2 *>(smem);
    d2 *sb = reinterpret_cast<d2 *>(smem + kDcTile * sizeof(float2));
    const int tid = threadIdx.x;
    const int blk_tiles = kDcTile / tile_len;                // whole mini-tiles per block
    const int64_t base = (int64_t)blockIdx.x * blk_tiles * tile_len;
    const int64_t nb_end = min(n, base + (int64_t)blk_tiles * tile_len);    // this block's samples are [base, nb_end)
    dc_stage_in(x, nb_end, base, sx);
    // carry from the mini-tiles before this block
    const int64_t m0 = (int64_t)blockIdx.x * blk_tiles;
    const int64_t reach = (80000 + tile_len - 1) / tile_len;
    const int64_t mlo = m0 > reach ? m0 - reach : 0;
    const double A = dc_pow(c, tile_len);
    d2 part = {0.0, 0.0};
    for (int64_t i = mlo + tid; i < m0; i += kDcThreads) {
        const double w = dc_pow(A, (int)(m0 - 1 - i));
        const d2 e = tile_end[i];
        part.x += w * e.x; part.y += w * e.y;
    }
    for (int o = 32; o > 0; o >>= 1) { part.x += __shfl_down(part.x, o, 64); part.y += __shfl_down(part.y, o, 64); }
    if ((tid & 63) == 0) sb[tid >> 6] = part;
    __syncthreads();
    const d2 s0 = state_in[0];
    const double wT = base < 200000 ? dc_pow(c, (int)base) : 0.0;
    d2 v_tile = {wT * s0.x, wT * s0.y};
    for (int u = 0; u < kDcThreads / 64; ++u) { v_tile.x += sb[u].x; v_tile.y += sb[u].y; }
    __syncthreads();

    d2 v = {0.0, 0.0};
    for (int i = 0; i < kDcSeg; ++i) {
        float2 s = sx[tid * kDcSeg + i];
        v.x = (double)s.x + c * v.x; v.y = (double)s.y + c * v.y;
    }
    v = dc_scan256(v, dc_pow(c, kDcSeg), v_tile, sb);
    for (int i = 0; i < kDcSeg; ++i) {
        int64_t g = base + tid * kDcSeg + i;
        float2 s = sx[tid * kDcSeg + i];
        d2 v0 = {(double)s.x + c * v.x, (double)s.y + c * v.y};
        sx[tid * kDcSeg + i] = make_float2((float)(v0.x - v.x), (float)(v0.y - v.y));
        if (g < nb_end) {
            v = v0;
            if (g == n - 1) state_out[0] = v0;
        }
    }
    __syncthreads();
    for (int i = tid; i < kDcTile / 2; i += kDcThreads) {
        const int64_t g = base + 2 * (int64_t)i;
        const float2 a = sx[2 * i], b = sx[2 * i + 1];
        if (g + 1 < nb_end) *reinterpret_cast<f4u *>(y + g) = f4u{a.x, a.y, b.x, b.y};
        else if (g < nb_end) y[g] = a;
    }
}

// ------------------------------------------------------------------------------------------------------------
// K2 + K4: critically-sampled polyphase analysis bank, M channels, 8 taps per branch, channel-major output.
//   X_t[c] = sum_{n<8} taps[c][n] x[(t-n) M + c];   y_t[k] = sum_c X_t[c] exp(-j 2 pi k c / M);   out[k][t]
//
// A workgroup owns TF consecutive frames.
//  phase 0  polyphase FIR straight from HBM/L2: the input is one flat stream, so lane i reads x[i - n M], n = 0..7,
//           fully coalesced 16-byte loads (two samples per lane); X goes to LDS as X[t][c] with a padded row stride.
//  phase 1  M = A B.  c = c1 B + c2, k = k1 + A k2:  Z[t][c2][k1] = W_M^(k1 c2) sum_c1 X[t][c1 B + c2] W_A^(k1 c1)
//           work item = (t, c2); KA (4..8) k1 accumulators per pass; the W_A twiddles are wave-uniform (scalar loads).
//  phase 2  y[t][k1 + A k2] = sum_c2 Z[t][c2][k1] W_B^(k2 c2);  work item = (t, k1); lanes run along t, so every
//           channel-major store instruction writes up to 512 contiguous bytes of one channel row.
// Cost per frame M (A + B) complex MACs for any M (4, 20, 122, 200, 1024 ...); the reference computes the same DFT
// with liquid's mixed-radix FFT and then copies out the channels that have consumers (SDRPostThread.cpp:336-339,
// :364-381); here only rows whose `active` flag is set are stored.
// ------------------------------------------------------------------------------------------------------------
constexpr int kChanTaps = 8;
constexpr int kChanMaxWaves = 8;

struct ChanGeom {
    int M, A, B;              // M = A * B
    int KA, KB;               // DFT outputs accumulated per pass (4..8): ceil(A / nkA), ceil(B / nkB)
    int nkA, nkB;             // passes over the k1 / k2 range
    int PA, PB;               // twiddle row pitch: nkA * KA, nkB * KB (rows zero padded)
    int TF, lgTF;             // frames per workgroup (power of two)
    int S;                    // LDS row stride in float2 units (>= M, conflict-free for lanes along t)
    unsigned magicM;          // floor(2^32 / M) + 1 : i / M for i < 2^20
    int taps_lds;             // 1: the [8][M] tap table is staged in LDS
    int stage_in;             // 1: the (TF + 7) M input samples of the tile are staged in LDS (aliasing the Z array)
    int fpw;                  // frames one workgroup processes (<= TF)
    int hop;                  // input samples between frames: M (firpfbch) or M / 2 (firpfbch2, 2x oversampled)
    int oddA;                 // 1: A is odd (>= 3) and phase 1 uses the conjugate-pair form: KA / nkA / PA then count output PAIRS (k, A - k)
    int threads;              // workgroup size (whole waves, 256..512): chosen so each phase splits evenly over the waves
    int p2;                   // 1: M = 2 A, A odd <= 63, critically sampled: chan_analyze_p2 (KA = slots per pass, nkA = passes, PA = row pitch)
    int mx;                   // chan_analyze_p2: 1 = the A-point transforms run on the fp32 matrix pipe (A >= 33: chan_analyze_p2<.., true>)
    int xcd;                  // chan_analyze_p2: 1 = the workgroups of one XCD (blockIdx.x % 8) take CONSECUTIVE tiles of a round (grid a multiple of 8)
};
__host__ __device__ inline size_t chan_zin_floats2(const ChanGeom &g) {      // Z array, or the staged input tile if larger
    const size_t z = (size_t)g.TF * g.S, in = g.stage_in ? (size_t)(g.TF - 1) * g.hop + (size_t)kChanTaps * g.M : 0;
    return ((z > in ? z : in) + 1) & ~(size_t)1;
}
__host__ __device__ inline size_t chan_lds_bytes(const ChanGeom &g) {
    size_t b = ((size_t)g.TF * g.S + chan_zin_floats2(g)) * sizeof(float2);
    if (g.taps_lds) b += (size_t)kChanTaps * g.M * sizeof(float);
    return (b + 15) & ~(size_t)15;
}

// K accumulators of an n-point DFT: acc[j] = sum_c v[c * vstride] * w[c * wpitch + j]; `w` is wave-uniform (scalar loads)
template <int K>
__device__ __forceinline__ void chan_dft(const float2 *v, int vstride, const float2 *__restrict__ w, int wpitch, int n, float2 (&acc)[K]) {
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] = make_float2(0.f, 0.f);
    for (int c = 0; c < n; ++c, w += wpitch) {
        const float2 x = v[c * vstride];
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] = cfma(x, w[j], acc[j]);
    }
}

// phase 1 for one wave item: A-point DFTs over c1 (K of the k1 outputs) for 64 (t, c2) pairs, times W_M^(k1 c2)
template <int K>
__device__ __forceinline__ void chan_phase1(const ChanGeom &g, const float2 *s_x, float2 *s_z, const float2 *__restrict__ twA,
                                            const float2 *__restrict__ twM, int k1b, int t, int c2) {
    const int A = g.A, B = g.B;
    // the W_M factors are fetched (row index clamped) before the DFT so one memory latency covers all of them
    float2 m[K];
#pragma unroll
    for (int j = 0; j < K; ++j) m[j] = twM[(size_t)min(k1b + j, A - 1) * B + c2];
    float2 acc[K];
    chan_dft<K>(s_x + (size_t)t * g.S + c2, B, twA + k1b, g.PA, A, acc);
    float2 *zr = s_z + (size_t)t * g.S + c2 * A + k1b;
#pragma unroll
    for (int j = 0; j < K; ++j) if (k1b + j < A) zr[j] = cmul(acc[j], m[j]);
}

// phase 1 when A is odd: outputs k and A - k share their products.  With s_c = x_c + x_{A-c}, d_c = x_c - x_{A-c} (c = 1 .. H,
// H = (A - 1) / 2):  P_k = x_0 + sum_c s_c cos(2 pi k c / A),  Q_k = sum_c d_c sin(2 pi k c / A),  X_k = P_k - j Q_k,
// X_{A-k} = P_k + j Q_k -- a quarter of the multiplies of the direct form (M = 20: A = 5; M = 122: A = 61).
// `cs` holds (cos, sin) of pair kp = 1 .. H and term c at [(c - 1) pitch + kp - 1], wave-uniform; KP pairs per pass.
template <int KP>
__device__ __forceinline__ void chan_phase1_odd(const ChanGeom &g, const float2 *s_x, float2 *s_z, const float2 *__restrict__ cs,
                                                const float2 *__restrict__ twM, int kpb, int t, int c2) {
    const int A = g.A, B = g.B, H = (A - 1) >> 1;
    const float2 *xr = s_x + (size_t)t * g.S + c2;
    float2 mk[KP], mn[KP];                                    // W_M factors of outputs k and A - k, fetched ahead
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        const int k = min(kpb + 1 + j, H);
        mk[j] = twM[(size_t)k * B + c2]; mn[j] = twM[(size_t)(A - k) * B + c2];
    }
    const float2 m0 = twM[c2];                                // k = 0 (row 0 of W_M is all ones, kept for uniformity)
    const float2 x0 = xr[0];
    float2 P[KP], Q[KP], sum0 = x0;
#pragma unroll
    for (int j = 0; j < KP; ++j) { P[j] = x0; Q[j] = make_float2(0.f, 0.f); }
    const float2 *w = cs + kpb;
    for (int c = 1; c <= H; ++c, w += g.PA) {
        const float2 a = xr[c * B], b = xr[(A - c) * B];
        const float2 sc = make_float2(a.x + b.x, a.y + b.y), dc = make_float2(a.x - b.x, a.y - b.y);
        sum0.x += sc.x; sum0.y += sc.y;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const float2 e = w[j];
            P[j].x = fmaf(sc.x, e.x, P[j].x); P[j].y = fmaf(sc.y, e.x, P[j].y);
            Q[j].x = fmaf(dc.x, e.y, Q[j].x); Q[j].y = fmaf(dc.y, e.y, Q[j].y);
        }
    }
    float2 *zr = s_z + (size_t)t * g.S + c2 * A;
    if (kpb == 0) zr[0] = cmul(sum0, m0);
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        const int k = kpb + 1 + j;
        if (k <= H) {
            zr[k] = cmul(make_float2(P[j].x + Q[j].y, P[j].y - Q[j].x), mk[j]);          // P - jQ
            zr[A - k] = cmul(make_float2(P[j].x - Q[j].y, P[j].y + Q[j].x), mn[j]);      // P + jQ
        }
    }
}

// phase 2 for one wave item: B-point DFTs over c2 (K of the k2 outputs) for 64 (t, k1) pairs; channel-major stores
template <int K>
__device__ __forceinline__ void chan_phase2(const ChanGeom &g, float2 *s_x, const float2 *s_z, const float2 *__restrict__ twB,
                                            const int *__restrict__ active, float2 *__restrict__ out, int64_t out_stride,
                                            int64_t f0, bool keep0, int k2b, int t, int k1,
                                            const float2 *__restrict__ post /* firpfbch2: [2][M] output factors by frame parity, else null */) {
    const int A = g.A, B = g.B, M = g.M;
    const int k = k1 + A * k2b;
    // consumer flags of the K rows, fetched (index clamped) ahead of the DFT: one latency, not K in a chain
    int on[K];
#pragma unroll
    for (int j = 0; j < K; ++j) on[j] = active[min(k + j * A, M - 1)];
    float2 acc[K];
    chan_dft<K>(s_z + (size_t)t * g.S + k1, A, twB + k2b, g.PB, B, acc);
    if (post) {
        const float2 *pr = post + (size_t)((f0 + t) & 1) * M;          // every batch holds an even number of frames
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] = cmul(acc[j], pr[min(k + j * A, M - 1)]);
    }
    float2 *o = out + f0 + t;
    if (k == 0 && keep0) s_x[t] = acc[0];                 // channel 0 of this tile (s_x is free after phase 1)
#pragma unroll
    for (int j = 0; j < K; ++j) if (k2b + j < B && on[j]) o[(int64_t)(on[j] - 1) * out_stride] = acc[j];      // on = output row of the channel + 1
}

template <int STAGE_IN, int TAPS_LDS, int OS2 /* 1: frames hop by M / 2 and may start at odd sample offsets */,
          int ODDA /* 1: phase 1 in the conjugate-pair form (its own variant: the registers it needs would cost the others occupancy) */>
CSDR_KERNEL __launch_bounds__(64 * kChanMaxWaves) void chan_analyze(
    const float2 *__restrict__ x,        // batch input, n_frames * M samples
    const float2 *__restrict__ hist,     // 8 * M - hop samples preceding x
    float2 *__restrict__ hist_new,       // receives the last 8 * M - hop samples of (hist ++ x)
    const float *__restrict__ tapsT,     // [8][M]  tapsT[n M + c] multiplies x[(t - n) M + c]
    const float2 *__restrict__ twA,      // [A][PA] exp(-j 2 pi k1 c1 / A) at [c1 PA + k1], zero padded
    const float2 *__restrict__ twB,      // [B][PB] exp(-j 2 pi k2 c2 / B) at [c2 PB + k2], zero padded
    const float2 *__restrict__ twM,      // [A][B]  exp(-j 2 pi k1 c2 / M) at [k1 B + c2]
    const int *__restrict__ active,      // [M] output row of channel k + 1 (the channel itself unless the rows are packed); 0: not stored
    ChanGeom g, int64_t n_frames,
    float2 *__restrict__ out, int64_t out_stride,
    d2 *__restrict__ dc_ends, double dc_c /* DC blocker of channel 0: end value of this tile's recurrence (zero entering state) */,
    const float2 *__restrict__ post) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = g.M, TF = g.TF, S = g.S, hop = g.hop;
    float2 *s_x = reinterpret_cast<float2 *>(smem);
    float2 *s_z = s_x + (size_t)TF * S;                   // also the staged input tile during phase 0
    float *s_taps = reinterpret_cast<float *>(s_z + chan_zin_floats2(g));
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int64_t f0 = (int64_t)blockIdx.x * g.fpw;       // first frame of this tile (g.fpw <= TF frames per workgroup)
    const int nf = (int)min((int64_t)g.fpw, n_frames - f0);
    const int64_t H = (int64_t)kChanTaps * M - hop;       // samples in front of frame 0's newest hop: 7 M, or 7.5 M when oversampled

    // frame t of the tile reads x[base + t hop + c - n M], c < M, n < 8 (base = first sample of frame f0's window end row)
    const int64_t base = f0 * hop - (M - hop);
    const int64_t Hs = (int64_t)(kChanTaps - 1) * M;      // staged samples in front of `base`
    if (STAGE_IN) {
        // one memory round trip: every input sample of the tile (the halo first) is loaded once, 16 bytes per lane
        const int n_in2 = ((nf - 1) * hop + kChanTaps * M + 1) >> 1;
        if (OS2) {
            for (int p = tid; p < n_in2; p += nthr) {
                const int64_t gi = base - Hs + 2 * (int64_t)p;
                const float2 *src = gi >= 0 ? x + gi : hist + (gi + H);
                // M / 2 may be odd: 8-byte aligned pairs; the pair may straddle hist | x
                const float2 a = src[0];
                float2 b2 = make_float2(0.f, 0.f);                              // one sample past the batch (odd count): never read, never used
                if (gi + 1 < n_frames * hop) b2 = (gi + 1 >= 0) ? x[gi + 1] : hist[gi + 1 + H];
                s_z[2 * p] = a; s_z[2 * p + 1] = b2;
            }
        } else {
            // eight 16-byte loads of a thread in flight at a time (one load, one LDS store per trip was a memory round trip per trip: fourteen
            // of them in a row at M = 200); indices past the end re-read the last pair and are not stored
            if (base - Hs >= 0) {                          // (block-uniform: every tile but the first lies wholly inside the batch)
                const float4 *src4 = reinterpret_cast<const float4 *>(x + (base - Hs));
                for (int p0 = tid; p0 < n_in2; p0 += 8 * nthr) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = src4[min(p0 + u * nthr, n_in2 - 1)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) pin_loaded(v[u]);
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int p = p0 + u * nthr; if (p < n_in2) reinterpret_cast<float4 *>(s_z)[p] = v[u]; }
                }
            } else {
                for (int p = tid; p < n_in2; p += nthr) {
                    const int64_t gi = base - Hs + 2 * (int64_t)p;
                    const float2 *src = gi >= 0 ? x + gi : hist + (gi + H);
                    reinterpret_cast<float4 *>(s_z)[p] = *reinterpret_cast<const float4 *>(src);
                }
            }
        }
    }
    if (TAPS_LDS) for (int i = tid; i < kChanTaps * M; i += nthr) s_taps[i] = tapsT[i];
    if (STAGE_IN || TAPS_LDS) __syncthreads();
    const float *tp = TAPS_LDS ? s_taps : tapsT;

    // ---- phase 0: polyphase FIR, two adjacent samples (same frame: M is even) per lane
    const int npairs = (nf * M) >> 1;
    for (int p = tid; p < npairs; p += nthr) {
        const unsigned i = 2u * (unsigned)p;
        const unsigned t = __umulhi(i, g.magicM);
        const unsigned c = i - t * (unsigned)M;
        const unsigned pos = OS2 ? t * (unsigned)hop + c : i;               // offset of x[.. + t hop + c] from `base`
        float2 a0 = make_float2(0.f, 0.f), a1 = make_float2(0.f, 0.f);
        if (STAGE_IN) {
            const float2 *sp = s_z + (size_t)(kChanTaps - 1) * M + pos;    // x[base + t hop + c] inside the staged tile
#pragma unroll
            for (int n = 0; n < kChanTaps; ++n) {
                float4 v;
                if (OS2) { const f4u u = *reinterpret_cast<const f4u *>(sp - n * M); v = make_float4(u.x, u.y, u.z, u.w); }
                else v = *reinterpret_cast<const float4 *>(sp - n * M);
                const float2 h = *reinterpret_cast<const float2 *>(tp + n * M + c);
                a0.x = fmaf(h.x, v.x, a0.x); a0.y = fmaf(h.x, v.y, a0.y);
                a1.x = fmaf(h.y, v.z, a1.x); a1.y = fmaf(h.y, v.w, a1.y);
            }
        } else {
            const int64_t gi0 = base + pos;
#pragma unroll
            for (int n = 0; n < kChanTaps; ++n) {
                const int64_t gi = gi0 - (int64_t)n * M;
                float4 v;
                if (OS2) {                                  // 8-byte aligned pair that may straddle hist | x
                    const float2 a = gi >= 0 ? x[gi] : hist[gi + H], b2 = gi + 1 >= 0 ? x[gi + 1] : hist[gi + 1 + H];
                    v = make_float4(a.x, a.y, b2.x, b2.y);
                } else {
                    const float2 *src = gi >= 0 ? x + gi : hist + (gi + H);
                    v = *reinterpret_cast<const float4 *>(src);
                }
                const float2 h = *reinterpret_cast<const float2 *>(tp + n * M + c);
                a0.x = fmaf(h.x, v.x, a0.x); a0.y = fmaf(h.x, v.y, a0.y);
                a1.x = fmaf(h.y, v.z, a1.x); a1.y = fmaf(h.y, v.w, a1.y);
            }
        }
        float2 *d = s_x + (size_t)t * S + c;
        d[0] = a0; d[1] = a1;
    }
    // the last workgroup also writes the new input history (the launch runs even with no consumers)
    if (blockIdx.x == gridDim.x - 1) {
        const int64_t n = n_frames * hop;
        for (int64_t j = tid; j < H; j += nthr) {
            const int64_t gsrc = n - H + j;
            hist_new[j] = gsrc >= 0 ? x[gsrc] : hist[gsrc + H];
        }
    }
    __syncthreads();

    // ---- phase 1: A-point DFTs over c1 for every (t, c2), times W_M^(k1 c2).
    // wave item = (block of KA k1, 64 consecutive (t, c2) items): k1b is wave-uniform, so the W_A rows are scalar loads.
    const int lane = tid & 63, wave = wave_uniform(tid >> 6), nw = nthr >> 6;
    const int tmask = TF - 1;
    {
        const int items = TF * g.B, nch = (items + 63) >> 6;
        for (int w = wave; w < g.nkA * nch; w += nw) {
            const int kb = w / nch, k1b = kb * g.KA;
            const int it = (w - kb * nch) * 64 + lane;
            const int t = it & tmask, c2 = it >> g.lgTF;
            if (it >= items || t >= nf) continue;
            if (ODDA) {                                         // k1b counts output pairs here
                switch (g.KA) {
                    case 1: chan_phase1_odd<1>(g, s_x, s_z, twA, twM, k1b, t, c2); break;
                    case 2: chan_phase1_odd<2>(g, s_x, s_z, twA, twM, k1b, t, c2); break;
                    case 3: chan_phase1_odd<3>(g, s_x, s_z, twA, twM, k1b, t, c2); break;
                    default: chan_phase1_odd<4>(g, s_x, s_z, twA, twM, k1b, t, c2); break;
                }
                continue;
            }
            switch (g.KA) {
                case 4: chan_phase1<4>(g, s_x, s_z, twA, twM, k1b, t, c2); break;
                case 5: chan_phase1<5>(g, s_x, s_z, twA, twM, k1b, t, c2); break;
                case 6: chan_phase1<6>(g, s_x, s_z, twA, twM, k1b, t, c2); break;
                case 7: chan_phase1<7>(g, s_x, s_z, twA, twM, k1b, t, c2); break;
                default: chan_phase1<8>(g, s_x, s_z, twA, twM, k1b, t, c2); break;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: B-point DFTs over c2 for every (t, k1); channel-major stores, lanes along t
    {
        const int items = TF * g.A, nch = (items + 63) >> 6;
        const bool keep0 = dc_ends != nullptr;
        for (int w = wave; w < g.nkB * nch; w += nw) {
            const int kb = w / nch, k2b = kb * g.KB;
            const int it = (w - kb * nch) * 64 + lane;
            const int t = it & tmask, k1 = it >> g.lgTF;
            if (it >= items || t >= nf) continue;
            switch (g.KB) {
                case 4: chan_phase2<4>(g, s_x, s_z, twB, active, out, out_stride, f0, keep0, k2b, t, k1, OS2 ? post : nullptr); break;
                case 5: chan_phase2<5>(g, s_x, s_z, twB, active, out, out_stride, f0, keep0, k2b, t, k1, OS2 ? post : nullptr); break;
                case 6: chan_phase2<6>(g, s_x, s_z, twB, active, out, out_stride, f0, keep0, k2b, t, k1, OS2 ? post : nullptr); break;
                case 7: chan_phase2<7>(g, s_x, s_z, twB, active, out, out_stride, f0, keep0, k2b, t, k1, OS2 ? post : nullptr); break;
                default: chan_phase2<8>(g, s_x, s_z, twB, active, out, out_stride, f0, keep0, k2b, t, k1, OS2 ? post : nullptr); break;
            }
        }
    }
    if (dc_ends) {
        // v_end = sum_t c^(nf-1-t) y0[t]: the DC blocker's state after this tile if it entered with zero (iirfilt, :375)
        __syncthreads();
        if (wave == 0) {
            double vx = 0.0, vy = 0.0;
            for (int t = lane; t < nf; t += 64) {
                const double wgt = dc_pow(dc_c, nf - 1 - t);
                const float2 v = s_x[t];
                vx += wgt * (double)v.x; vy += wgt * (double)v.y;
            }
            for (int o = 32; o > 0; o >>= 1) { vx += __shfl_down(vx, o, 64); vy += __shfl_down(vy, o, 64); }
            if (lane == 0) dc_ends[blockIdx.x] = d2{vx, vy};
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// K2 + K4 for M = 2 A with A odd (M = 6, 10, 14, 22, ... 122 = the 61.44 MS/s case, A <= 63): the Cooley-Tukey split has B = 2,
// so the B-point pass is one butterfly and the whole transform of a frame stays inside one lane.
// Persistent workgroups of eight waves (two per CU) walk over tiles of 64 consecutive frames; LDS holds ONE array of 64 rows
// X[t][c] (the FIR's output):
//  window   lane = column pair (c1; c2 = 0, 1 = one float4), wave = 8 consecutive frames: the fifteen input rows those frames
//           reach (16 A contiguous bytes per wave and row) are loaded straight from global memory into the registers the FIR
//           reads -- requested one tile ahead, eight rows before the DFT phase of the previous tile and seven after it (all
//           fifteen at once do not fit next to the accumulators).  Until the middle of round 3 the tile went through a flat
//           register stage, a commit to LDS and halo reads: two more barriers per tile, 36 registers, and a load skeleton that alone
//           took half of the kernel.  Neighbouring waves read seven rows twice: cache hits, the HBM traffic is the same.
//  FIR      the eight frames in ascending order from the window; the taps (a guarded load is compiled behind a wait for everything in
//           flight: they are read without a lane guard) live in registers; then s = x_c + x_{A-c} / d = x_c - x_{A-c} by a lane trade,
//           X[t] to LDS.
//  DFT      lane = frame t, wave = KP of the (A - 1) / 2 conjugate output pairs (k, A - k) for BOTH c2 at once
//           (4 KP accumulators): per term c one ds_read_b128 pair x_c, x_{A-c} (row stride 4 A dwords: conflict-free for
//           odd A), the (cos, sin) rows are wave-uniform scalar loads, both requested one term ahead; then the radix-2
//           butterfly with W_M^k and four channel-major stores of 512 contiguous bytes per wave (scalar row base + lane
//           offset).  k = 0 rides along as a pseudo pair with (cos, sin) = (1, 0).  No Z array, no third phase.
// (v_pk_fma_f32 issues every ~5 clk per SIMD with >= 2 waves resident on it, 13 clk with one: measured with a micro-benchmark.)
// ------------------------------------------------------------------------------------------------------------
constexpr int kP2Frames = 64;          // frames per tile (lane = frame in the DFT phase)
constexpr int kP2Waves = 8;
#ifndef CSDR_P2_PRIO_DFT
#define CSDR_P2_PRIO_DFT 1
#endif
#ifndef CSDR_P2_REQ_PRIO
#define CSDR_P2_REQ_PRIO 2
#endif
constexpr int kP2ReqPrio = CSDR_P2_REQ_PRIO;
constexpr int kP2DftPrio = CSDR_P2_PRIO_DFT;            // (A/B builds: -DCSDR_P2_PRIO_DFT=0 is the round-5 kernel)
#ifndef CSDR_P2_EARLY
#define CSDR_P2_EARLY 8
#endif
constexpr int kP2EarlyRows = CSDR_P2_EARLY;          // rows of the next tile's FIR window requested before the DFT phase (the first frame's whole window); the other seven after it
constexpr int kP2Threads = 64 * kP2Waves;
#ifndef CSDR_P2_MIRROR
#define CSDR_P2_MIRROR 1
#endif
constexpr bool kP2Mirror = CSDR_P2_MIRROR != 0;          // matrix-pipe form: odd waves hold their FIR window in descending order (A/B builds: -DCSDR_P2_MIRROR=0)
constexpr int kP2MaxA = 63;
// (the matrix-pipe form keeps two small tables behind the rows: the per-output constants of its epilogue and the tile's channel-0 samples)
constexpr int kMxSteps = 8;            // K steps of four terms: n = 0 .. 31 (H <= 31)
constexpr int kMxRows = 32;            // two row tiles of sixteen outputs: k = 0 .. 31
__host__ __device__ inline size_t chan_p2_lds_bytes(int M, bool mx = false, bool os2 = false) {
    return (size_t)kP2Frames * M * sizeof(float2) + (mx ? (size_t)kMxRows * (sizeof(float4) + 4 * sizeof(int)) + kP2Frames * sizeof(float2) + (size_t)(kChanTaps / 2) * (M / 2) * sizeof(float4) : 0) +
           (os2 ? (size_t)4 * kMxRows * sizeof(float4) : 0);
}
// coefficient fragments of the matrix-pipe form, [2 (cos | sin)][2 row tiles][kMxSteps][64 lanes]: lane l of step J holds the coefficient of output
// k = 16 rt + (l & 15) and term n = 4 J + (l >> 4) -- the A operand of v_mfma_f32_16x16x4_f32 (A[i = l & 15][k = l >> 4]); term 0 is x_0 (cos = 1, sin = 0),
// terms and outputs past H are zero.  The angles are the vector form's expression, so the products are the same products.
__host__ inline void chan_mx_table(int A, float *tab /* [2][2][kMxSteps][64] */) {
    const int H = (A - 1) / 2;
    for (int kind = 0; kind < 2; ++kind) for (int rt = 0; rt < 2; ++rt) for (int J = 0; J < kMxSteps; ++J) for (int l = 0; l < 64; ++l) {
        const int k = 16 * rt + (l & 15), n = 4 * J + (l >> 4);
        float v = 0.f;
        if (k <= H && n <= H) {
            const double ang = 2.0 * M_PI * (double)(((long long)n * k) % A) / (double)A;
            if (kind == 0) v = n == 0 ? 1.0f : (float)std::cos(ang);
            else v = n == 0 ? 0.0f : (float)std::sin(ang);
        }
        tab[(((size_t)kind * 2 + rt) * kMxSteps + J) * 64 + l] = v;
    }
}
// store to a wave-uniform row base plus a 32-bit per-lane byte offset (scalar-base addressing: no 64-bit address arithmetic per lane)
__device__ __forceinline__ void store_row(float2 *row_base, unsigned byte_off, float2 v) {
    *reinterpret_cast<float2 *>(reinterpret_cast<char *>(row_base) + byte_off) = v;
}
// the same with the streaming hint (the output is not read again by this kernel: kept out of the way of the window rows in L2)
__device__ __forceinline__ void store_row_nt(float2 *row_base, unsigned byte_off, float2 v) {
#if defined(__AMDGCN__)
    typedef float csdr_v2f __attribute__((ext_vector_type(2)));
    const csdr_v2f t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<csdr_v2f *>(reinterpret_cast<char *>(row_base) + byte_off));
#else
    store_row(row_base, byte_off, v);
#endif
}
// streaming-hint store through a per-lane address (the matrix-pipe form: a store instruction covers four channel rows)
__device__ __forceinline__ void store_nt(float2 *p, float2 v) {
#if defined(__AMDGCN__)
    typedef float csdr_v2f __attribute__((ext_vector_type(2)));
    const csdr_v2f t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<csdr_v2f *>(p));
#else
    *p = v;
#endif
}
// one term of the conjugate-pair sums for KP slots and both c2: s = x_c + x_{A-c} (the FIR phase left it at row c), d = x_c - x_{A-c}
// (at row A - c); e = (cos, sin) rows
template <int KP>
__device__ __forceinline__ void chan_p2_term(const float4 s, const float4 d, const float2 (&e)[KP], float2 (&P0)[KP], float2 (&Q0)[KP],
                                             float2 (&P1)[KP], float2 (&Q1)[KP]) {
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        P0[j].x = fmaf(s.x, e[j].x, P0[j].x); P0[j].y = fmaf(s.y, e[j].x, P0[j].y);
        P1[j].x = fmaf(s.z, e[j].x, P1[j].x); P1[j].y = fmaf(s.w, e[j].x, P1[j].y);
        Q0[j].x = fmaf(d.x, e[j].y, Q0[j].x); Q0[j].y = fmaf(d.y, e[j].y, Q0[j].y);
        Q1[j].x = fmaf(d.z, e[j].y, Q1[j].x); Q1[j].y = fmaf(d.w, e[j].y, Q1[j].y);
    }
}

// the conjugate-pair sums of one pass (KP slots, both c2) over the terms c = 1 .. H.  Software pipeline, two terms per trip with two
// register sets: the rows and the (cos, sin) row of the next term are requested before the current one is accumulated; the second request
// of a trip is unconditional (the last trip re-reads term H).  (Round 3 measured the guarded request and s / d formed here, in every wave's
// passes, as well: 3 % slower before the window went straight into registers, bit-identical results; those forms are gone.)
template <int KP>
__device__ __forceinline__ void chan_p2_accumulate(const float4 *row, const int A, const int H, const float2 *__restrict__ w, const int PA,
                                                   float2 (&P0)[KP], float2 (&Q0)[KP], float2 (&P1)[KP], float2 (&Q1)[KP]) {
    float4 a = row[1], b = row[A - 1], a2 = a, b2 = b;
    float2 eA[KP], eB[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) { eA[j] = w[j]; eB[j] = eA[j]; }
    int c = 1;
    for (; c + 1 <= H; c += 2) {
        w += PA;
        a2 = row[c + 1]; b2 = row[A - c - 1];
#pragma unroll
        for (int j = 0; j < KP; ++j) eB[j] = w[j];
        sched_fence();
        chan_p2_term<KP>(a, b, eA, P0, Q0, P1, Q1);
        sched_fence();
        const int cn = min(c + 2, H);
        w += (c + 2 <= H) ? PA : 0;
        a = row[cn]; b = row[A - cn];
#pragma unroll
        for (int j = 0; j < KP; ++j) eA[j] = w[j];
        sched_fence();
        chan_p2_term<KP>(a2, b2, eB, P0, Q0, P1, Q1);
        sched_fence();
    }
    if (c <= H) chan_p2_term<KP>(a, b, eA, P0, Q0, P1, Q1);        // H odd: the last term
}

// the window of one wave's FIR range, straight into registers: the 8 frames [ta, ta + 8) of tile `tile` need the 15 input rows
// f0 + ta - 7 .. f0 + ta + 7; lane = column pair, a row is 16 A contiguous bytes per wave.  Rows in front of the batch come from the carried
// history, rows past its end are zero (their frames are never stored).  The row index and the source select are wave-uniform: fifteen
// plain loads under one lane mask, nothing between them.
template <int J0 = 0, int J1 = 2 * kChanTaps - 1, bool OS2 = false>
__device__ __forceinline__ void chan_p2_request_window(const float2 *__restrict__ x, const float2 *__restrict__ hist, int M, int A, int64_t n_frames,
                                                       int64_t tile, bool valid, int wave, int lane, float4 (&win)[2 * kChanTaps - 1], bool mir = false) {
    // (mir: the window is held in descending order -- win[j] = row r0 + 14 - j.  The matrix-pipe form's odd waves do that, so that the seven rows a wave
    //  shares with each neighbour are requested at the same positions of the sequence by both -- the early rows with the wave above, the late ones with
    //  the wave below -- and the second request meets the first in the cache: 10.7 -> 8.9 B/sample fetched)
    // OS2 (firpfbch2, frames hop by A = M / 2): the frames of even and of odd index are two lattices of rows M apart, the odd one's rows start A samples
    // later; wave = (lattice l = wave & 1, eight frames of it): row r of lattice l holds the samples r M + (l - 1) A .. + M - 1 (frame f = 2 r + l ends with
    // it), the history is the 7.5 M samples in front of the batch.  Row 0 of lattice 0 lies half in the history: the lane of its middle pair reads one
    // sample from each side.
    const bool col = lane < A;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (OS2) {
        const int l = wave & 1;
        const int64_t r0 = tile * (kP2Frames / 2) + (int64_t)(wave >> 1) * kChanTaps - (kChanTaps - 1);
        const int64_t Hs = (int64_t)kChanTaps * M - A;
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const int64_t r = r0 + (mir ? 2 * kChanTaps - 2 - j : j);               // wave-uniform
            const int64_t s0 = r * M + (int64_t)(l - 1) * A;                        // first sample of the row (batch-relative)
            win[j] = z4;
            if (valid && 2 * r + l < n_frames) {                                    // (wave-uniform)
                if (s0 >= 0 || s0 + M <= 0) {
                    const float2 *src = s0 >= 0 ? x + s0 : hist + (s0 + Hs);
                    if (col) { const f4u v = *reinterpret_cast<const f4u *>(src + 2 * lane); win[j] = make_float4(v.x, v.y, v.z, v.w); }
                } else if (col) {                                                   // the row that straddles the batch's start (s0 = - A)
                    const int64_t sa = s0 + 2 * lane, sb = sa + 1;
                    const float2 a = sa >= 0 ? x[sa] : hist[sa + Hs], b = sb >= 0 ? x[sb] : hist[sb + Hs];
                    win[j] = make_float4(a.x, a.y, b.x, b.y);
                }
            }
        }
        return;
    }
    const int64_t r0 = tile * kP2Frames + (int64_t)wave * kChanTaps - (kChanTaps - 1);      // input row of win[0] (win[14] when mirrored)
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        const int64_t r = r0 + (mir ? 2 * kChanTaps - 2 - j : j);                   // wave-uniform
        const float2 *src = r >= 0 ? x + r * M : hist + (r + (kChanTaps - 1)) * M;
        win[j] = z4;
        if (valid && r < n_frames) {                                                // (wave-uniform)
            if (col) { const f4u v = *reinterpret_cast<const f4u *>(src + 2 * lane); win[j] = make_float4(v.x, v.y, v.z, v.w); }
        }
    }
}

// MX = true (A >= 33: two row tiles of outputs): the transform phase runs on the fp32 matrix pipe -- see "DFT on the matrix pipe" below; `cs` then is
// the coefficient-fragment table of chan_mx_table().
// OS2 (matrix-pipe form only): firpfbch2 -- frames hop by A = M / 2, `post` = [2 frame parities][M] output factors (design::channelizer2_post).
template <int KP, bool MX = false, bool OS2 = false>
CSDR_KERNEL __launch_bounds__(kP2Threads, 4) void chan_analyze_p2(
    const float2 *__restrict__ x, const float2 *__restrict__ hist, float2 *__restrict__ hist_new,
    const float *__restrict__ tapsT,      // [8][M]
    const float2 *__restrict__ cs,        // [(A-1)/2][PA]: (cos, sin)(2 pi k(q) c / A) at [(c - 1) PA + q]; slot q: k = q + 1 (q < H), k = 0 (q == H), else (0, 0)
    const float2 *__restrict__ twM,       // [A][2]: exp(-j 2 pi k1 c2 / M) at [2 k1 + c2]
    const int *__restrict__ active, ChanGeom g, int64_t n_frames,
    float2 *__restrict__ out, int64_t out_stride, d2 *__restrict__ dc_ends, double dc_c, const float2 *__restrict__ post) {
    static_assert(!OS2 || MX, "the oversampled bank exists in the matrix-pipe form only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *rows = reinterpret_cast<float4 *>(smem);          // X[t] at rows + t A
    const int M = g.M, A = g.A, H = (A - 1) >> 1;
    const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave = wave_uniform(tid0 >> 6);
    const int64_t n_tiles = (n_frames + kP2Frames - 1) / kP2Frames;
    const int64_t Hs = (int64_t)kChanTaps * M - (OS2 ? A : M);   // samples in front of a tile's own first row = carried history length (7 M; oversampled 7.5 M)
    // the workgroup that finishes last in program order is unknown: the new input history is written by workgroup 0 up front
    // (it only reads x / hist, which nobody writes during this launch)
    if (blockIdx.x == 0) {
        const int64_t n = n_frames * (OS2 ? A : M);
        for (int64_t j = tid0; j < Hs; j += kP2Threads) {
            const int64_t gsrc = n - Hs + j;
            hist_new[j] = gsrc >= 0 ? x[gsrc] : hist[gsrc + Hs];
        }
    }
    // matrix-pipe form: this wave's coefficient fragments stay in registers for the whole launch (persistent workgroup); the constants of the
    // epilogue (W_M^k of an output and of its partner A - k, the four output rows) sit in LDS behind the rows, one entry per output k
    float mxc[kMxSteps], mxs[kMxSteps];
    float4 *epi_w = reinterpret_cast<float4 *>(smem + (size_t)kP2Frames * M * sizeof(float2));
    int4 *epi_on = reinterpret_cast<int4 *>(epi_w + kMxRows);
    float2 *s_y0 = reinterpret_cast<float2 *>(epi_on + kMxRows);                  // channel 0 of the tile (the DC blocker's end value)
    float4 *s_taps = reinterpret_cast<float4 *>(s_y0 + kP2Frames);                // [4][A]: taps 2 j and 2 j + 1 of a column pair
    float4 *epi_post = s_taps + (kChanTaps / 2) * A;                              // OS2: [2 parities][32 outputs k][2]: post factors of rows (k, k + A), (A - k, 2 A - k)
    if constexpr (OS2) {
        if (tid0 < 2 * kMxRows) {
            const int par = tid0 >> 5, k = tid0 & 31, kn = A - k;
            float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
            if (k <= H) {
                const float2 *pr = post + (size_t)par * M;
                pa = make_float4(pr[k].x, pr[k].y, pr[k + A].x, pr[k + A].y);
                if (k > 0) pb = make_float4(pr[kn].x, pr[kn].y, pr[kn + A].x, pr[kn + A].y);
            }
            epi_post[2 * tid0] = pa; epi_post[2 * tid0 + 1] = pb;
        }
    }
    if constexpr (MX) {
        for (int i = tid0; i < (kChanTaps / 2) * A; i += kP2Threads) {
            const int j = i / A, lc = i - j * A;
            const float2 h0 = *reinterpret_cast<const float2 *>(tapsT + (2 * j) * M + 2 * lc), h1 = *reinterpret_cast<const float2 *>(tapsT + (2 * j + 1) * M + 2 * lc);
            s_taps[i] = make_float4(h0.x, h0.y, h1.x, h1.y);
        }
        const float *tab = reinterpret_cast<const float *>(cs) + (size_t)(wave >> 2) * kMxSteps * 64 + lane0;
#pragma unroll
        for (int J = 0; J < kMxSteps; ++J) { mxc[J] = tab[J * 64]; mxs[J] = tab[(2 * kMxSteps + J) * 64]; }
        if (tid0 < kMxRows) {
            const int k = tid0, kn = A - k;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            int4 on = make_int4(0, 0, 0, 0);
            if (k <= H) {
                const float2 wk = twM[2 * k + 1];
                w.x = wk.x; w.y = wk.y; on.x = active[k]; on.y = active[k + A];
                if (k > 0) { const float2 wn = twM[2 * kn + 1]; w.z = wn.x; w.w = wn.y; on.z = active[kn]; on.w = active[kn + A]; }      // k = 0 has no conjugate partner
            }
            epi_w[k] = w; epi_on[k] = on;
        }
        __syncthreads();
    }
    float4 win[2 * kChanTaps - 1];                              // this wave's FIR window of the tile: input rows f0 + ta - 7 .. f0 + ta + 7
    // tile walk: workgroup b takes tiles b, b + grid, ...; with `xcd` the workgroups that share an L2 (b % 8: the dispatcher's round robin over the
    // XCDs) take consecutive tiles of the round, so that the seven window rows two neighbouring tiles share meet in one L2
    int64_t tile = blockIdx.x, tstep = gridDim.x, tend = n_tiles;
    if (g.xcd == 1 && !(gridDim.x & 7)) tile = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (g.xcd == 2) {                                          // every workgroup takes a contiguous range of tiles: the seven window rows two consecutive tiles share are its own
        const int64_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
        tile = (int64_t)blockIdx.x * per; tstep = 1; tend = min(n_tiles, tile + per);
    }
    const bool mirw = MX && kP2Mirror && (((OS2 ? wave >> 1 : wave) & 1) != 0);      // (the neighbours that share rows: the next wave, oversampled the next wave of the same lattice)
    chan_p2_request_window<0, 2 * kChanTaps - 1, OS2>(x, hist, M, A, n_frames, tile, tile < tend, wave, lane0, win, mirw);
    for (; tile < tend; tile += tstep) {
        const int64_t f0 = tile * kP2Frames;
        const int nf = (int)min((int64_t)kP2Frames, n_frames - f0);
        int lane = lane0, tid = tid0;                         // per-tile copies: their address arithmetic is not worth carrying across tiles
        opaque(lane); opaque(tid);
        const bool col = lane < A;
        float2 h[kChanTaps];
        if constexpr (MX) {   // from the workgroup's LDS copy: four 16-byte reads, no global round trip behind the window's
            const int lc = min(lane, A - 1);
#pragma unroll
            for (int n = 0; n < kChanTaps; n += 2) { const float4 v = s_taps[(n >> 1) * A + lc]; h[n] = make_float2(v.x, v.y); h[n + 1] = make_float2(v.z, v.w); }
        } else {   // (lanes past the last column read the last column's taps: nothing of theirs is stored, and a guarded load is compiled behind a
            // wait for everything in flight -- eight serialised round trips per tile)
            const int lc = min(lane, A - 1);
#pragma unroll
            for (int n = 0; n < kChanTaps; ++n) h[n] = *reinterpret_cast<const float2 *>(tapsT + n * M + 2 * lc);
        }
        // ---- FIR: frames [ta, ta + 8) of this wave from the window in registers (requested one tile ahead); X[t] goes to row t of the LDS array
        {
            constexpr int kRange = kP2Frames / kP2Waves;
            static_assert(kRange == kChanTaps, "a wave's range is eight frames: its window is fifteen rows");
            // tile-local frame of the wave's i-th output: eight consecutive ones, or (oversampled) eight of its lattice
            const int ta = OS2 ? 2 * kRange * (wave >> 1) + (wave & 1) : wave * kRange, ts = OS2 ? 2 : 1;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            auto fir_frame = [&](const int i, auto mirrored) -> float4 {      // tap n multiplies input row t - n = window row i + 7 - n
                float4 acc = z4;
#pragma unroll
                for (int n = 0; n < kChanTaps; ++n) {
                    constexpr bool kMir = decltype(mirrored)::value;
                    const int wr = i + kChanTaps - 1 - n;
                    const float4 v = win[kMir ? 2 * kChanTaps - 2 - wr : wr];
                    acc.x = fmaf(h[n].x, v.x, acc.x); acc.y = fmaf(h[n].x, v.y, acc.y);
                    acc.z = fmaf(h[n].y, v.z, acc.z); acc.w = fmaf(h[n].y, v.w, acc.w);
                }
                return acc;
            };
            if (mirw) {                                       // (wave-uniform) the mirrored window: nothing but the stores follows a frame's sum
#pragma unroll
                for (int i = kRange - 1; i >= 0; --i) {       // last frame first: its window is the eight rows that were requested early
                    const float4 acc = fir_frame(i, std::true_type{});
                    if (col) rows[(ta + ts * i) * A + lane] = acc;
                    sched_fence();
                }
            } else
#pragma unroll
            for (int i = 0; i < kRange; ++i) {
                float4 acc = fir_frame(i, std::false_type{});
                if constexpr (!MX) {                          // columns c and A - c trade: s_c = x_c + x_{A-c} at row c, d_c = x_c - x_{A-c} at row A - c
                    // (one component at a time: a float4 of partner values at once is one more spilled float4 in this phase)
                    // p + sg acc with sg = +1 on the s lanes, -1 on the d lanes: ONE multiply-add per component in place of two selects, an add and a subtract
                    // (x + y and fma(1, x, y) are the same single rounding: bit-identical); the lanes outside 1 .. A - 1 keep their own value
                    const int pl = (lane >= 1 && lane < A) ? A - lane : lane;
                    const bool is_sd = lane >= 1 && lane < A;
                    const float sg = lane <= H ? 1.0f : -1.0f;
                    float p;
                    p = __shfl(acc.x, pl, 64); acc.x = is_sd ? fmaf(sg, acc.x, p) : acc.x;
                    p = __shfl(acc.y, pl, 64); acc.y = is_sd ? fmaf(sg, acc.y, p) : acc.y;
                    p = __shfl(acc.z, pl, 64); acc.z = is_sd ? fmaf(sg, acc.z, p) : acc.z;
                    p = __shfl(acc.w, pl, 64); acc.w = is_sd ? fmaf(sg, acc.w, p) : acc.w;
                }
                if (col) rows[(ta + ts * i) * A + lane] = acc;
                sched_fence();                                // frame after frame
            }
        }
        lds_barrier();
        // the transform phase issues at raised priority: of the four waves of a SIMD (two workgroups) the ones inside their 480 packed multiply-adds
        // go first, the ones in the FIR / load / store phases fill in (0.504 -> 0.489 ms on C3, profiles/r06_p2_variants.txt; raising the FIR phase
        // instead costs 1 %); the window requests one step above it (- 1.5 %)
        wave_priority(kP2ReqPrio);
        // the next tile's window is on its way while this one is transformed (into the registers the FIR has just finished with)
        chan_p2_request_window<0, kP2EarlyRows, OS2>(x, hist, M, A, n_frames, tile + tstep, tile + tstep < tend, wave, lane, win, mirw);
        wave_priority(kP2DftPrio);
        if constexpr (MX) {
            // ---- DFT on the matrix pipe.  The conjugate-pair sums are two real matrix products per component:
            //        P[k][t] = sum_n Cos[k][n] s_n[t]      Q[k][t] = sum_n Sin[k][n] d_n[t]        k, n = 0 .. H   (s_0 = x_0, Cos[k][0] = 1, Sin[k][0] = 0)
            //      for the four components (c2 = 0 / 1) x (re / im) of s and d: eight products, tiled 16 (k) x 16 (t) x 4 (n) on v_mfma_f32_16x16x4_f32.
            //      Wave = (row tile rt = wave >> 2: k = 16 rt .. 16 rt + 15, column tile ct = wave & 3: frames 16 ct .. 16 ct + 15); lane (q = lane >> 4,
            //      j = lane & 15) feeds term n = 4 J + q of frame t = 16 ct + j in step J -- the two ds_read_b128 of rows n and A - n (x_n, x_{A-n}: their sum and
            //      difference) are the B operands of all eight products -- and receives outputs k = 16 rt + 4 q + r (r = 0..3) of that frame for all
            //      eight, so the radix-2 butterfly and the stores stay in-lane.  An MFMA is a k-ordered fmaf chain: the accumulation order (n ascending,
            //      starting from x_0) is the vector form's, and so are the results, bit for bit.  64 MFMAs (2048 matrix-pipe cycles) per wave and tile
            //      in place of 465 packed multiply-adds on the vector pipe, 16 LDS reads in place of 61 per pass.
            const int q = lane >> 4, t = 16 * (wave & 3) + (lane & 15), rt = wave >> 2;
            const float4 *row = rows + t * A;
            const bool tv = t < nf;
            csdr_f32x4 P0r = {0.f, 0.f, 0.f, 0.f}, P0i = P0r, P1r = P0r, P1i = P0r, Q0r = P0r, Q0i = P0r, Q1r = P0r, Q1i = P0r;
            // s_n = x_n + x_{A-n}, d_n = x_n - x_{A-n} are formed here from the two rows (each rounded once, as the vector form's lane trade does in its FIR
            // phase: the same values); term 0 is x_0 itself.  Terms past H have zero coefficients: they read a row of the frame all the same (a finite operand).
            // (A < 33 -- the oversampled bank's small counts: the second row tile holds no output, its waves sit the phase out; steps past the last term are skipped)
            const int KS = (H + 4) >> 2;                               // ceil((H + 1) / 4) steps, (wave-uniform)
            const bool rt_live = 16 * rt <= H;
            auto row_s = [&](int n) { return row[min(n, A - 1)]; };
            auto row_d = [&](int n) { return row[n >= 1 && n < A ? A - n : 1]; };
            float4 a = row_s(q), b = q ? row_d(q) : make_float4(0.f, 0.f, 0.f, 0.f);      // step 0: n = q
            if (rt_live) {
#pragma unroll
                for (int J = 0; J < kMxSteps; ++J) {
                    if (J < KS) {
                        float4 a2 = a, b2 = b;
                        if (J + 1 < kMxSteps) { const int n2 = 4 * (J + 1) + q; a2 = row_s(n2); b2 = row_d(n2); }      // the next step's rows are requested ahead of this step's products
                        const float4 sv = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
                        const float4 dv = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
                        P0r = csdr_mfma16(mxc[J], sv.x, P0r); P0i = csdr_mfma16(mxc[J], sv.y, P0i);
                        P1r = csdr_mfma16(mxc[J], sv.z, P1r); P1i = csdr_mfma16(mxc[J], sv.w, P1i);
                        Q0r = csdr_mfma16(mxs[J], dv.x, Q0r); Q0i = csdr_mfma16(mxs[J], dv.y, Q0i);
                        Q1r = csdr_mfma16(mxs[J], dv.z, Q1r); Q1i = csdr_mfma16(mxs[J], dv.w, Q1i);
                        a = a2; b = b2;
                    }
                }
            }
            // each accumulator holds four outputs k of one frame: a store instruction of the wave covers four channel rows, 128 contiguous bytes of each
            float2 *ob = out + f0 + t;
            if (rt_live)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * rt + 4 * q + r;
                const float4 w = epi_w[k];
                const int4 on = epi_on[k];
                const float2 P0 = make_float2(P0r[r], P0i[r]), Q0 = make_float2(Q0r[r], Q0i[r]), P1 = make_float2(P1r[r], P1i[r]), Q1 = make_float2(Q1r[r], Q1i[r]);
                const float2 z0k = make_float2(P0.x + Q0.y, P0.y - Q0.x), z0n = make_float2(P0.x - Q0.y, P0.y + Q0.x);
                const float2 u = cmul(make_float2(P1.x + Q1.y, P1.y - Q1.x), make_float2(w.x, w.y));
                const float2 v = cmul(make_float2(P1.x - Q1.y, P1.y + Q1.x), make_float2(w.z, w.w));
                float2 y0 = make_float2(z0k.x + u.x, z0k.y + u.y), y1 = make_float2(z0k.x - u.x, z0k.y - u.y);
                float2 y2 = make_float2(z0n.x + v.x, z0n.y + v.y), y3 = make_float2(z0n.x - v.x, z0n.y - v.y);
                if constexpr (OS2) {                                // firpfbch2: times the channel's post factor of this frame's parity (tiles start on even frames)
                    const float4 pa = epi_post[2 * (32 * (t & 1) + k)], pb = epi_post[2 * (32 * (t & 1) + k) + 1];
                    y0 = cmul(y0, make_float2(pa.x, pa.y)); y1 = cmul(y1, make_float2(pa.z, pa.w));
                    y2 = cmul(y2, make_float2(pb.x, pb.y)); y3 = cmul(y3, make_float2(pb.z, pb.w));
                }
                if (tv) {
                    if (on.x) { if (k) store_nt(ob + (int64_t)(on.x - 1) * out_stride, y0); else ob[(int64_t)(on.x - 1) * out_stride] = y0; }      // (channel 0 is read again by the DC blocker)
                    if (on.y) store_nt(ob + (int64_t)(on.y - 1) * out_stride, y1);
                    if (on.z) store_nt(ob + (int64_t)(on.z - 1) * out_stride, y2);
                    if (on.w) store_nt(ob + (int64_t)(on.w - 1) * out_stride, y3);
                }
                if (r == 0 && dc_ends && k == 0) s_y0[t] = y0;        // k = 0: Q = 0, W = 1 -- y0 = P0 + P1 as in the vector form
            }
        } else {   // ---- DFT: lane = frame t; wave = pass of KP output-pair slots
            const int t = lane;
            const float4 *row = rows + t * A;
            const bool tv = t < nf;
            for (int p = wave; p < g.nkA; p += kP2Waves) {
                const int q0 = p * KP;
                const float4 x0 = row[0];
                float2 P0[KP], Q0[KP], P1[KP], Q1[KP];
#pragma unroll
                for (int j = 0; j < KP; ++j) {
                    P0[j] = make_float2(x0.x, x0.y); P1[j] = make_float2(x0.z, x0.w);
                    Q0[j] = make_float2(0.f, 0.f); Q1[j] = make_float2(0.f, 0.f);
                }
                chan_p2_accumulate<KP>(row, A, H, cs + q0, g.PA, P0, Q0, P1, Q1);
                float2 *ob = out + f0;                                  // wave-uniform row bases + a 32-bit lane offset: scalar-base stores
                const unsigned tb = (unsigned)t * (unsigned)sizeof(float2);   // byte offset of this lane inside a channel row
#pragma unroll
                for (int j = 0; j < KP; ++j) {
                    const int q = q0 + j;                               // wave-uniform
                    if (q < H) {
                        const int k = q + 1, kn = A - k;
                        const float2 wk = twM[2 * k + 1], wn = twM[2 * kn + 1];
                        const int on0 = active[k], on1 = active[k + A], on2 = active[kn], on3 = active[kn + A];
                        const float2 z0k = make_float2(P0[j].x + Q0[j].y, P0[j].y - Q0[j].x), z0n = make_float2(P0[j].x - Q0[j].y, P0[j].y + Q0[j].x);
                        const float2 u = cmul(make_float2(P1[j].x + Q1[j].y, P1[j].y - Q1[j].x), wk);
                        const float2 v = cmul(make_float2(P1[j].x - Q1[j].y, P1[j].y + Q1[j].x), wn);
                        if (tv) {                                       // streaming-hint stores: nothing in this kernel reads the rows again, they stay out of the way of the window rows two waves share (12.6 -> 10.5 B/sample fetched on C3)
                            // (on = output row of the channel + 1: the channel itself unless the rows are packed, csdr_post_set_row_order)
                            if (on0) store_row_nt(ob + (int64_t)(on0 - 1) * out_stride, tb, make_float2(z0k.x + u.x, z0k.y + u.y));
                            if (on1) store_row_nt(ob + (int64_t)(on1 - 1) * out_stride, tb, make_float2(z0k.x - u.x, z0k.y - u.y));
                            if (on2) store_row_nt(ob + (int64_t)(on2 - 1) * out_stride, tb, make_float2(z0n.x + v.x, z0n.y + v.y));
                            if (on3) store_row_nt(ob + (int64_t)(on3 - 1) * out_stride, tb, make_float2(z0n.x - v.x, z0n.y - v.y));
                        }
                    } else if (q == H) {                                // k = 0: P = sum of the column, Q = 0
                        const float2 y0 = make_float2(P0[j].x + P1[j].x, P0[j].y + P1[j].y);
                        if (tv) {
                            const int r0 = active[0], rA = active[A];
                            if (r0) store_row(ob + (int64_t)(r0 - 1) * out_stride, tb, y0);
                            if (rA) store_row(ob + (int64_t)(rA - 1) * out_stride, tb, make_float2(P0[j].x - P1[j].x, P0[j].y - P1[j].y));
                        }
                        if (dc_ends) {
                            // v_end = sum_t c^(nf-1-t) y0[t]: the DC blocker's state after this tile if it entered with zero (iirfilt, :375)
                            const double wgt = tv ? dc_pow(dc_c, nf - 1 - t) : 0.0;
                            double vx = tv ? wgt * (double)y0.x : 0.0, vy = tv ? wgt * (double)y0.y : 0.0;      // rows past the last frame hold no data
                            for (int s2 = 32; s2 > 0; s2 >>= 1) { vx += __shfl_down(vx, s2, 64); vy += __shfl_down(vy, s2, 64); }
                            if (lane == 0) dc_ends[tile] = d2{vx, vy};
                        }
                    }
                }
            }
        }
        wave_priority(kP2ReqPrio);
        // the rest of the next window (the transform above leaves no room for all fifteen rows: they would be spilled -- which waits for them)
        chan_p2_request_window<kP2EarlyRows, 2 * kChanTaps - 1, OS2>(x, hist, M, A, n_frames, tile + tstep, tile + tstep < tend, wave, lane0, win, mirw);
        wave_priority(0);
        lds_barrier();                                      // the rows are free for the next tile
        if constexpr (MX) {
            if (dc_ends && wave == 0) {
                // v_end = sum_t c^(nf-1-t) y0[t], added up in the vector form's order (one wave, lane = frame); the next write of s_y0 is behind the next tile's first barrier
                const int t = lane0;
                const bool tv = t < nf;
                const float2 y0 = s_y0[t];
                const double wgt = tv ? dc_pow(dc_c, nf - 1 - t) : 0.0;
                double vx = tv ? wgt * (double)y0.x : 0.0, vy = tv ? wgt * (double)y0.y : 0.0;
                for (int s2 = 32; s2 > 0; s2 >>= 1) { vx += __shfl_down(vx, s2, 64); vy += __shfl_down(vy, s2, 64); }
                if (lane0 == 0) dc_ends[tile] = d2{vx, vy};
            }
        }
    }
}

}  // namespace csdr
