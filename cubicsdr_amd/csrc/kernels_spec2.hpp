// kernels_spec2.hpp -- the 512-point column pass of the N = 512 x R factorisation, and (an experiment of the measurement build) the spectrum chain
// of the headline size N = 2^17 = 512 x 256 with the averaging fused into the second transform pass.
//
// IN THE PRODUCT: spec_cols512 is pass 1 of every 2^21-point frame (BASELINE config 5: fftSize 1 048 576): one 512-point column pass through LDS
// in front of the 4096-point rows of kernels_spec.hpp, instead of a radix-32 and a radix-16 pass through HBM -- 16 instead of 32 B/sample
// (C5: 0.31 -> 0.19 ms per batch, and the rows / averaging kernels gain from the one-level row layout: 35.0 -> 38.6 GS/s; against a float64
// transform the 2^21-point display values are 9.5e-6 off where the two-pass form was 2.2e-5 and the reference's own class is 1.8e-5).
//
// Replaces (reference file:line): fft_execute SpectrumVisualProcessor.cpp:439, magnitude + fftshift :441-452, the double EMA and the running
// extrema :494-511 -- for the full-span view without peak hold (the other cases run the kernels of kernels_spec.hpp).
//
// Why another factorisation.  The averagers recur over FRAMES per bin; a transform pass that is to carry them in registers has to own its
// bins for the whole batch and walk the frames in order.  With 4096-point rows (kernels_spec.hpp) a frame of 2^17 points has 16 row pairs:
// sixteen workgroups.  With N = 512 x 256 the second pass has 512 rows of 256 points: 256 row PAIRS (rows k1 even and k1 + 1 hold the two
// adjacent bins k1 + 512 k2, k1 + 1 + 512 k2 of one display point), one workgroup each, every thread owning ONE display point's two bins:
// four doubles of averager state, the reference's statements as they are (NaN repairs included), no frame groups, no blocked scan.  The
// magnitudes never leave the CU: 8 B/sample of traffic less than row FFT -> magnitudes -> averaging kernel.
//   spec_cols512      pass 1: 512-point column transforms (radix 32 in registers, an LDS exchange, radix 16), times W_N^(n2 k1); 16 adjacent
//                     columns per workgroup (128-byte runs on both sides).  Z[f][k1][n2].
//   spec_rows256_avg  pass 2 + K15: workgroup = row pair; eight waves transform eight consecutive frames (two 256-point rows each, Stockham radix 4
//                     in wave-private LDS) while four waves run their display points' bins through the eight frames before, in order; pair sums
//                     in pair-row order [f][row pair][k2], per-frame extrema per row pair.
//   spec_display_rows256  K16 for that order: 32 x 32 tiles transposed through LDS (128-byte runs on both sides).
#pragma once
#include "kernels_spec.hpp"

namespace csdr {

constexpr int kC512 = 512, kC512Cols = 16;
constexpr int kC512KaPitch = 16 * 16 + 16;                       // [k_a][v][col] with 16 float2 of padding per k_a: two k_a of a half-wave hit different bank halves
constexpr size_t kC512Lds = (size_t)16 * kC512KaPitch * sizeof(float2);      // the exchange runs in two halves of sixteen k_a: 35 KB, four workgroups per CU by LDS

// pass 1.  Frame f, columns n2 in [16 blockIdx.x, + 16): X[k1][n2] = W_N^(n2 k1) sum_n1 x[n1 R + n2] W_512^(n1 k1), n1 = 16 u + v, k1 = k_a + 32 k_b.
// thread = (v = tid >> 4, col = tid & 15): radix 32 over u in registers, times W_512^(v k_a); exchange; thread = (k_a = tid >> 4 and + 16, col):
// radix 16 over v, times W_N^(n2 k1); stores Z[f][k1][n2] (16 columns = 128 contiguous bytes per k1).
CSDR_KERNEL __launch_bounds__(kFftThreads) void spec_cols512(FrameSrc fs, int N, const float2 *__restrict__ tw4096, const float2 *__restrict__ tw_hi,
                                                                const float2 *__restrict__ tw_lo, float2 *__restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *s_x = reinterpret_cast<float2 *>(smem);
    const int R = N / kC512;
    const int f = blockIdx.y, tid = threadIdx.x, col = tid & 15, hi = tid >> 4;
    const int n2 = blockIdx.x * kC512Cols + col;
    const float2 *xb = frame_ptr(fs, f);
    float2 a[32];
    {
        const int v = hi;
        if (f == 0 && fs.split < N) {                                // (block-uniform) the one frame that lies in two pieces
#pragma unroll
            for (int u = 0; u < 32; ++u) a[u] = frame_at(fs, f, xb, (int64_t)(16 * u + v) * R + n2);
        } else {
            const float2 *xc = xb + (int64_t)v * R + n2;
#pragma unroll
            for (int u = 0; u < 32; ++u) a[u] = xc[(int64_t)(16 * u) * R];
        }
        dft_reg<32>(a);
        float2 leaf[5];
#pragma unroll
        for (int l = 0; l < 5; ++l) leaf[l] = tw4096[(8 * v) << l];              // W_512^(v 2^l) = exp(-2 pi i 8 v 2^l / 4096), 8 * 15 * 16 < 4096
        twiddle_powers<32>(a, leaf);
    }
    float2 *o = dst + (int64_t)f * N + n2;
    // W_N^(n2 (ka + 32 kb)) = W_N^(n2 ka) (W_N^(32 n2))^kb
    float2 leaf[5];
#pragma unroll
    for (int l = 0; l < 4; ++l) leaf[l] = tw_split(tw_hi, tw_lo, (unsigned)(32 * n2) << l);      // 32 n2 8 < N (n2 < R = N / 512)
    leaf[4] = leaf[3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();                                      // the first half has been picked up
#pragma unroll
        for (int ka = 0; ka < 16; ++ka) s_x[ka * kC512KaPitch + hi * 16 + col] = a[16 * h + ka];      // (hi = v here)
        __syncthreads();
        const int ka = hi + 16 * h;
        float2 b[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) b[v] = s_x[hi * kC512KaPitch + v * 16 + col];
        dft_reg<16>(b);
        twiddle_powers<16>(b, leaf);
        const float2 w0 = tw_split(tw_hi, tw_lo, (unsigned)(n2 * ka));
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) st_stream(o + (int64_t)(ka + 32 * kb) * R, cmul(b[kb], w0));
    }
}

// ---- pass 2 + averaging, R = 256
constexpr int kR2 = 256;
constexpr int kR2Frames = 8;                                      // frames per round = transforming waves per workgroup
constexpr int kR2AvgWaves = kR2 / 64;                             // waves 0 .. 3 own the 256 display points of the row pair and only average
constexpr int kR2Threads = 64 * (kR2AvgWaves + kR2Frames);
constexpr size_t kR2Lds = (size_t)kR2Frames * 4 * kR2 * sizeof(float2) /* per transforming wave: two rows, ping + pong */ +
                          (size_t)2 * kR2Frames * 2 * kR2 * sizeof(float) /* magnitudes of two rounds */ +
                          (size_t)2 * kR2Frames * kR2AvgWaves * 2 * sizeof(float) /* per-frame extrema of the averaging waves, two rounds */;

// one radix-4 Stockham pass over BOTH rows of a wave (64 butterflies per row, one per lane): src -> dst
// (the lane's three twiddles of the pass sit in registers for the whole launch: w = null for the first pass, whose twiddles are 1)
__device__ __forceinline__ void r2_pass(const float2 *sa, const float2 *sb, float2 *da, float2 *db, int Ns, int lane, const float2 *w) {
    constexpr int q = kR2 / 4;
    const int j = lane, k = j & (Ns - 1);
    float2 a0 = sa[j], a1 = sa[j + q], a2 = sa[j + 2 * q], a3 = sa[j + 3 * q];
    float2 b0 = sb[j], b1 = sb[j + q], b2 = sb[j + 2 * q], b3 = sb[j + 3 * q];
    if (w) {
        a1 = cmul(a1, w[0]); a2 = cmul(a2, w[1]); a3 = cmul(a3, w[2]);
        b1 = cmul(b1, w[0]); b2 = cmul(b2, w[1]); b3 = cmul(b3, w[2]);
    }
    const int j0 = ((j - k) << 2) + k;
    {
        const float2 p0 = make_float2(a0.x + a2.x, a0.y + a2.y), p1 = make_float2(a0.x - a2.x, a0.y - a2.y);
        const float2 p2 = make_float2(a1.x + a3.x, a1.y + a3.y), p3 = make_float2(a1.x - a3.x, a1.y - a3.y);
        da[j0] = make_float2(p0.x + p2.x, p0.y + p2.y);
        da[j0 + Ns] = make_float2(p1.x + p3.y, p1.y - p3.x);           // p1 - j p3
        da[j0 + 2 * Ns] = make_float2(p0.x - p2.x, p0.y - p2.y);
        da[j0 + 3 * Ns] = make_float2(p1.x - p3.y, p1.y + p3.x);       // p1 + j p3
    }
    {
        const float2 p0 = make_float2(b0.x + b2.x, b0.y + b2.y), p1 = make_float2(b0.x - b2.x, b0.y - b2.y);
        const float2 p2 = make_float2(b1.x + b3.x, b1.y + b3.y), p3 = make_float2(b1.x - b3.x, b1.y - b3.y);
        db[j0] = make_float2(p0.x + p2.x, p0.y + p2.y);
        db[j0 + Ns] = make_float2(p1.x + p3.y, p1.y - p3.x);
        db[j0 + 2 * Ns] = make_float2(p0.x - p2.x, p0.y - p2.y);
        db[j0 + 3 * Ns] = make_float2(p1.x - p3.y, p1.y + p3.x);
    }
}

// grid = 256 row pairs (N = 2^17), twelve waves: eight transform (wave 4 + q takes frame 8 r + q of round r: its two rows, magnitudes into LDS),
// four average (thread = display point: the frames of round r - 1 in order) -- both at once, one workgroup barrier per round.
// Z: [frames][512][256].  pairsum[f][pair][k2] (float), ext_w[f][pair] = (max, min) of the float-rounded averaged bins, first_b[f] = bin 1's maa
// (display point 0).
CSDR_KERNEL __launch_bounds__(kR2Threads) void spec_rows256_avg(const float2 *__restrict__ Z, int nf, SpecGeom g, double rate, const float2 *__restrict__ tw4096,
                                                                 double *__restrict__ ma, double *__restrict__ maa, float *__restrict__ pairsum,
                                                                 float *__restrict__ first_b, float2 *__restrict__ ext_w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
    float2 *s_rows = reinterpret_cast<float2 *>(smem);                                   // [transforming wave][row a ping | row a pong | row b ping | row b pong]
    float *s_mag = reinterpret_cast<float *>(s_rows + (size_t)kR2Frames * 4 * kR2);      // [round parity][frame of the round][row][k2]
    float *s_ex = s_mag + 2 * kR2Frames * 2 * kR2;                                       // [round parity][frame][averaging wave][max | min]
    const int pair = blockIdx.x, npairs = gridDim.x, F = g.F;
    const int64_t N = g.N;
    const int nrounds = (nf + kR2Frames - 1) / kR2Frames;
    if (w >= kR2AvgWaves) {
        // ================= transforming waves
        const int pw = w - kR2AvgWaves;
        float2 *s_w = s_rows + (size_t)pw * 4 * kR2;
        float2 tw[3][3];                                           // passes Ns = 4, 16, 64: W^(k), W^(2k), W^(3k), k = lane & (Ns - 1)
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) {
            const int Ns = 4 << (2 * ps), k = lane & (Ns - 1), ts = kTwTab / (Ns * 4);
            tw[ps][0] = tw4096[k * ts]; tw[ps][1] = tw4096[2 * k * ts]; tw[ps][2] = tw4096[3 * k * ts];
        }
        const float2 *za = Z + (int64_t)(2 * pair) * kR2, *zb = za + kR2;           // rows 2 pair and 2 pair + 1 of frame 0
        float2 ra[4], rb[4];
        {
            const int f = min(pw, nf - 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) { ra[i] = za[(int64_t)f * N + lane + 64 * i]; rb[i] = zb[(int64_t)f * N + lane + 64 * i]; }
        }
        for (int it = 0; it <= nrounds; ++it) {
            if (it < nrounds) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { s_w[lane + 64 * i] = ra[i]; s_w[2 * kR2 + lane + 64 * i] = rb[i]; }
                {   // the next round's rows are requested before this round's arithmetic (frames past the end re-read the last one)
                    const int fn = min((it + 1) * kR2Frames + pw, nf - 1);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { ra[i] = za[(int64_t)fn * N + lane + 64 * i]; rb[i] = zb[(int64_t)fn * N + lane + 64 * i]; }
                }
                wave_sync();
                r2_pass(s_w, s_w + 2 * kR2, s_w + kR2, s_w + 3 * kR2, 1, lane, nullptr);   wave_sync();
                r2_pass(s_w + kR2, s_w + 3 * kR2, s_w, s_w + 2 * kR2, 4, lane, tw[0]);     wave_sync();
                r2_pass(s_w, s_w + 2 * kR2, s_w + kR2, s_w + 3 * kR2, 16, lane, tw[1]);    wave_sync();
                r2_pass(s_w + kR2, s_w + 3 * kR2, s_w, s_w + 2 * kR2, 64, lane, tw[2]);    wave_sync();
                float *mg = s_mag + (size_t)((it & 1) * kR2Frames + pw) * 2 * kR2;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    mg[lane + 64 * i] = cabs_f(s_w[lane + 64 * i]);
                    mg[kR2 + lane + 64 * i] = cabs_f(s_w[2 * kR2 + lane + 64 * i]);
                }
            }
            __syncthreads();
        }
        return;
    }
    // ================= averaging waves: this thread's display point, bins ka = 2 pair + 512 tid and ka + 1
    const int ka = 2 * pair + kC512 * tid;
    const int x = (int)(((ka - N / 2) & (N - 1)) >> 1);
    AvgState s = {ma[x], maa[x], ma[F + x], maa[F + x]};
    auto publish = [&](int r) {                                    // the per-frame extrema of round r, left in LDS one barrier ago
        const int nfr = min(kR2Frames, nf - r * kR2Frames);
        if (tid < nfr) {
            const float *e = s_ex + (size_t)((r & 1) * kR2Frames + tid) * kR2AvgWaves * 2;
            float mx = 0.f, mn = 3.0e38f;                          // the starting values of spec_average's tiles
            for (int q = 0; q < kR2AvgWaves; ++q) { mx = fmaxf(mx, e[2 * q]); mn = fminf(mn, e[2 * q + 1]); }
            ext_w[(int64_t)(r * kR2Frames + tid) * npairs + pair] = make_float2(mx, mn);
        }
    };
    for (int it = 0; it <= nrounds; ++it) {
        if (it >= 2) publish(it - 2);
        if (it >= 1) {
            const int r = it - 1, fb = r * kR2Frames, nfr = min(kR2Frames, nf - fb);
            const float *mg = s_mag + (size_t)(r & 1) * kR2Frames * 2 * kR2;
            float *ex = s_ex + (size_t)(r & 1) * kR2Frames * kR2AvgWaves * 2;
            // the magnitudes of the whole round first (independent loads), then the recurrence -- the only serial chain -- frame after frame; the
            // stores and the extrema hang off it and are folded for all frames together at the end (eight independent reductions in flight)
            float xa[kR2Frames], xb[kR2Frames], mxs[kR2Frames], mns[kR2Frames];
#pragma unroll
            for (int i = 0; i < kR2Frames; ++i) { xa[i] = mg[(i * 2 + 0) * kR2 + tid]; xb[i] = mg[(i * 2 + 1) * kR2 + tid]; }
#pragma unroll
            for (int i = 0; i < kR2Frames; ++i) {
                mxs[i] = 0.f; mns[i] = 3.0e38f;
                if (i < nfr) {                                     // (block-uniform)
                    const int f = fb + i;
                    avg_step(s, (double)xa[i], (double)xb[i], rate);           // the reference's statements, NaN repairs included
                    const float fa = (float)s.maa_a, fbb = (float)s.maa_b;         // float rounding is monotonic: extrema of the rounded values
                    stf(pairsum + (int64_t)f * F, (unsigned)(pair * kR2 + tid) * 4u, (float)(s.maa_a + s.maa_b));
                    if (x == 0) first_b[f] = fbb;
                    mxs[i] = fmaxf(fa, fbb); mns[i] = fminf(fa, fbb);            // (fmaxf / fminf skip a NaN operand, as the reference's comparisons do)
                }
            }
#pragma unroll
            for (int i = 0; i < kR2Frames; ++i) { mxs[i] = wave_max_to_lane63(mxs[i]); mns[i] = wave_min_to_lane63(mns[i]); }
            if (lane == 63) {
#pragma unroll
                for (int i = 0; i < kR2Frames; ++i) { ex[(i * kR2AvgWaves + w) * 2] = mxs[i]; ex[(i * kR2AvgWaves + w) * 2 + 1] = mns[i]; }
            }
        }
        __syncthreads();
    }
    publish(nrounds - 1);
    ma[x] = s.ma_a; maa[x] = s.maa_a; ma[F + x] = s.ma_b; maa[F + x] = s.maa_b;
}

// ---- K16 for the pair-row order of spec_rows256_avg: pairsum[f][pair][k2], display point x = (pair + 256 k2 - N / 4) mod F.
// grid = (8 x 8 tiles of 32 pairs x 32 k2, frames); reads 32 runs of 128 bytes, writes 32 runs of 128 bytes.
CSDR_KERNEL __launch_bounds__(kDispThreads) void spec_display_rows256(const float *__restrict__ pairsum, const float *__restrict__ first_b,
                                                                     const SpecFrameScal *__restrict__ fsc, SpecGeom g, float sf, float *__restrict__ points) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_y = reinterpret_cast<float *>(smem);                    // [32 k2][33]
    const int f = blockIdx.y, tid = threadIdx.x, F = g.F;
    const int npairs = kC512 / 2;
    const int p0 = (blockIdx.x & 7) * 32, t0 = (blockIdx.x >> 3) * 32;
    const SpecFrameScal sc = fsc[f];
    const double pf = sc.pf, fl = sc.fl;
    const float inv_den = 1.0f / log1pf((float)(sc.pc - pf));
    float a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = (tid >> 5) + 8 * u, j = tid & 31;              // pair p0 + i, k2 t0 + j
        a[u] = pairsum[(int64_t)f * F + (int64_t)(p0 + i) * kR2 + t0 + j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = (tid >> 5) + 8 * u, j = tid & 31;
        const int x = (p0 + i + npairs * (t0 + j) - (int)(g.N >> 2)) & (F - 1);
        const double acc = (x == 0) ? fl + (double)first_b[f] : (double)a[u];      // idx == 0 is replaced by fft_floor_maa (:546-556)
        s_y[j * 33 + i] = log1p_fast((float)(acc * 0.5 - pf)) * inv_den * sf;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = (tid >> 5) + 8 * u, i = tid & 31;              // 32 consecutive display points per k2
        const int x = (p0 + i + npairs * (t0 + j) - (int)(g.N >> 2)) & (F - 1);
        st_stream(points + (int64_t)f * F + x, s_y[j * 33 + i]);
    }
}

}  // namespace csdr
