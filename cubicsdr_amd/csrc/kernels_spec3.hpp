// kernels_spec3.hpp -- the spectrum chain of the headline size N = 2^17 = 512 x 256 (BASELINE config 3: fftSize 65536), full-span view (with or
// without peak hold): TWO transform passes with the magnitudes and the double averaging fused into the second one.
//
// Replaces (reference file:line): fft_execute SpectrumVisualProcessor.cpp:439 (liquid's radix-2 plan, :177), magnitude + fftshift :441-452, the
// double EMA and the running extrema :494-511; the display loop :532-576 for the order the second pass leaves its pair sums in.
//
// Bytes.  radix pass + 4096-point rows + averaging kernel + display moved 16 + 12 + 6 + 4 = 38 B per input sample for 12 algorithmic ones; this
// chain moves 16 (pass 1) + 8 + 2 (pass 2: rows in, pair sums out) + 4 (display) = 30: the magnitudes never leave the CU.
//   spec_cols512p     pass 1: 512-point column transforms as 8 x 8 x 8 -- EIGHT points per thread (a 1024-thread workgroup owns 16 adjacent
//                     columns = 128-byte runs, two workgroups per CU, <= 64 registers), two exchanges through LDS, every butterfly on the packed
//                     fp32 pipe (cpx.hpp); the workgroup keeps its columns and walks over the frames, so the output twiddles W_N^(n2 k1) of its
//                     threads are loop invariants (eight register pairs).  Z[f][k1][n2].
//   spec_rows256_ema  pass 2 + K14 + K15: workgroup = row PAIR (rows k1 even and k1 + 1 hold the two adjacent bins of one display point), which
//                     it owns for the whole batch and walks frame by frame: eight waves transform sixteen frames per round (a 16-lane group per
//                     256-point row: 16-point transforms in registers, one 16 x 16 transpose through LDS, the next round's rows requested
//                     before this round's arithmetic) while four waves run their 256 display points' bins through the round before, in
//                     order, with the reference's statements; one workgroup barrier per round.
//   spec_display_p256 K16 for the pair-row order of pass 2: 32 x 32 tiles transposed through LDS (128-byte runs on both sides).
// Round 4 built this factorisation once (exact, parity-green, slower than the three-kernel chain: 0.47 + 0.42 ms against 0.35 + 0.27
// + 0.18 ms, bound by instruction issue at 224 registers / two waves per SIMD in pass 1 and ~57 instructions per point in pass 2); these are
// new kernels on the packed arithmetic of cpx.hpp.
#pragma once
#include "cpx.hpp"
#include "kernels_spec.hpp"

namespace csdr {

constexpr int kS3N = 1 << 17, kS3C = 512, kS3R = 256;             // N = C x R: C-point columns (stride R), R-point rows

// ------------------------------------------------------------------------------------------------------------------------------ pass 1
constexpr int kP1Threads = 1024, kP1Cols = 16;
constexpr int kP1QPitch = 8 * kP1Cols + kP1Cols;                  // exchange 2: [p][q] slabs of 8 c x 16 columns, one column group of padding (the four (p, q) of a wave's read hit alternate bank halves)
constexpr int kP1Xchg1 = kS3C * kP1Cols;                          // exchange 1: [p][r][col], 8192 complex
constexpr int kP1Xchg2 = 64 * kP1QPitch;                          // exchange 2: 9216 complex
constexpr size_t kP1Lds = (size_t)(kP1Xchg1 + kP1Xchg2 + kS3C) * sizeof(float2);      // + the 512-entry table exp(-2 pi i k / 512): 143 360 B, one workgroup per CU

// n1 = 64 a + r (r = 8 b + c),  k1 = p + 8 q + 64 t:
//   X[k1] = sum_c W64^(c (q + 8 t)) [ W512^(c p) ... ] -- three 8-point transforms: over a (times W512^(r p)), over b (times W64^(c q)), over c.
// thread = (g = tid >> 4, col = tid & 15):  step 1 g = r;  step 2 g = (p, c);  step 3 g = (p, q).
// One 1024-thread workgroup per CU (16 waves: four per SIMD, <= 128 registers); the two exchanges have a buffer each, so a frame costs two
// barriers (the writes of exchange 1 for frame f + 1 are behind the barrier that follows exchange 2's writes of frame f, which every thread
// passes after its step-2 reads; likewise exchange 2), and the next frame's rows are requested before this frame's first butterfly.
CSDR_KERNEL __launch_bounds__(kP1Threads, 4) void spec_cols512p(FrameSrc fs, int nf, const float2 *__restrict__ tw_hi, const float2 *__restrict__ tw_lo,
                                                               float2 *__restrict__ Z) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cpx *s_x1 = reinterpret_cast<cpx *>(smem);
    cpx *s_x2 = s_x1 + kP1Xchg1;
    cpx *s_w = s_x2 + kP1Xchg2;                                       // W512 table
    const int tid = threadIdx.x, col = tid & 15, g = tid >> 4;
    const int ntile = kS3R / kP1Cols;                                 // 16 column tiles per frame
    const int ct = blockIdx.x % ntile, fg = blockIdx.x / ntile, nfg = gridDim.x / ntile;
    const int n2 = ct * kP1Cols + col;
    if (tid < kS3C) {                                                 // exp(-2 pi i tid / 512) = exp(-2 pi i (256 tid) / N): exact table entries
        const float2 w = tw_split(tw_hi, tw_lo, (unsigned)tid * (unsigned)(kS3N / kS3C));
        s_w[tid] = cpx_from(w);
    }
    // output twiddles of this thread's eight results (step 3: p = g >> 3, q = g & 7): W_N^(n2 (p + 8 q + 64 t)), loop invariants
    cpx twf[8];
    {
        const int p = g >> 3, q = g & 7;
#pragma unroll
        for (int t = 0; t < 8; ++t) twf[t] = cpx_from(tw_split(tw_hi, tw_lo, (unsigned)n2 * (unsigned)(p + 8 * q + 64 * t)));      // n2 k1 < 256 * 512 = N
    }
    __syncthreads();
    const unsigned in_off = (unsigned)((g * kS3R + n2) * (int)sizeof(float2));          // row r = g of step 1, byte offset inside a frame
    const unsigned out_off = (unsigned)((((g >> 3) + 8 * (g & 7)) * kS3R + n2) * (int)sizeof(float2));   // row p + 8 q of step 3
    constexpr unsigned kRowStep = 64u * kS3R * sizeof(float2);        // 64 rows
    cpx nx[8];
    auto request = [&](int f) {                                       // rows 64 a + r of frame f, a = 0 .. 7
        const float2 *xb = frame_ptr(fs, f);
        if (f == 0 && fs.split < kS3N) {                             // (block-uniform) the one frame that lies in two pieces
#pragma unroll
            for (int k = 0; k < 8; ++k) nx[k] = cpx_from(frame_at(fs, 0, xb, (int64_t)(64 * k + g) * kS3R + n2));
        } else {
            const char *base = reinterpret_cast<const char *>(xb);
#pragma unroll
            for (int k = 0; k < 8; ++k) nx[k] = cpx_from(*reinterpret_cast<const float2 *>(base + in_off + (unsigned)k * kRowStep));
        }
    };
    if (fg < nf) request(fg);
    for (int f = fg; f < nf; f += nfg) {
        cpx a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = nx[k];
        if (f + nfg < nf) request(f + nfg);
        // ---- step 1: over a; times W512^(r p)
        cpx_dft<8>(a);
#pragma unroll
        for (int p = 1; p < 8; ++p) a[p] = cpx_mul(a[p], s_w[g * p]);
#pragma unroll
        for (int p = 0; p < 8; ++p) s_x1[(p * 64 + g) * kP1Cols + col] = a[p];
        lds_barrier();                                                // (LDS only: the requested rows stay in flight across it)
        {   // ---- step 2: thread (p, c): over b; times W64^(c q)
            const int p = g >> 3, c = g & 7;
#pragma unroll
            for (int b = 0; b < 8; ++b) a[b] = s_x1[(p * 64 + 8 * b + c) * kP1Cols + col];
            cpx_dft<8>(a);
#pragma unroll
            for (int q = 1; q < 8; ++q) a[q] = cpx_mul(a[q], s_w[8 * c * q]);
#pragma unroll
            for (int q = 0; q < 8; ++q) s_x2[(p * 8 + q) * kP1QPitch + c * kP1Cols + col] = a[q];
        }
        lds_barrier();                                                // (LDS only: the requested rows stay in flight across it)
        {   // ---- step 3: thread (p, q): over c; outputs k1 = p + 8 q + 64 t
#pragma unroll
            for (int c = 0; c < 8; ++c) a[c] = s_x2[g * kP1QPitch + c * kP1Cols + col];
            cpx_dft<8>(a);
            char *ob = reinterpret_cast<char *>(Z + (int64_t)f * kS3N);
#pragma unroll
            for (int t = 0; t < 8; ++t) st_stream(reinterpret_cast<float2 *>(ob + out_off + (unsigned)t * kRowStep), cpx_to(cpx_mul(a[t], twf[t])));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------ pass 2
constexpr int kR2TWaves = 8;                                      // transforming waves: two frames (four rows) each
constexpr int kR2AWaves = kS3R / 64;                              // averaging waves: one display point (two bins) per thread
constexpr int kR2Round = 2 * kR2TWaves;                           // frames per round
constexpr int kR2Threads = 64 * (kR2AWaves + kR2TWaves);
constexpr int kR2RowPitch = 18;                                   // transpose rows of 16 complex padded to 18: sixteen 16-byte reads of a quarter wave hit distinct banks
constexpr int kR2SubPitch = 16 * kR2RowPitch + 16;                // one 16-lane group's region; the + 16 puts the two groups of a half wave on alternate bank halves
constexpr int kR2MagPitch = kS3R + 16;                            // magnitude planes [frame of the round][row]: the four planes a wave writes together start 16 banks apart
constexpr int kR2ExtFrames = 4;
constexpr size_t kR2LdsXchg = (size_t)kR2TWaves * 4 * kR2SubPitch * sizeof(float2);
constexpr size_t kR2LdsMag = (size_t)2 * kR2Round * 2 * kR2MagPitch * sizeof(float);
constexpr size_t kR2LdsExt = (size_t)kR2AWaves * 2 * kR2ExtFrames * 64 * sizeof(float);
constexpr size_t kR2LdsPart = (size_t)2 * kR2Round * kR2AWaves * sizeof(float2);
constexpr size_t kR2Lds = kR2LdsXchg + kR2LdsMag + kR2LdsExt + kR2LdsPart;

// Z: [frames][512][256] (pass 1).  pairsum[f][pair][k2] (float), ext_w[f][pair] = (max, min) of the float-rounded averaged bins of the row pair,
// first_b[f] = the averaged second bin of display point 0.  ma / maa: the averagers, at spec_state_index(g, x) of the geometry `g` the three-kernel
// chain uses for this size (the two chains trade places when peak hold or the zoomed view is switched: one state layout).
// HOLD (peak hold live, SpectrumVisualProcessor.cpp:247-273, :506-510): the frames >= pk_from also raise the held maximum of their two bins (the running
// maximum of the float-rounded averaged values, spec_peak_track's statements) -- peak[x] / peak[F + x] by display point -- and leave peaksum[f][pair][k2] in
// the pair-sum's order, peak_b[f] = the held second bin of display point 0.
template <bool HOLD>
CSDR_KERNEL __launch_bounds__(kR2Threads) void spec_rows256_ema(const float2 *__restrict__ Z, int nf, SpecGeom g, double rate, const float2 *__restrict__ tw4096,
                                                               double *__restrict__ ma, double *__restrict__ maa, float *__restrict__ pairsum,
                                                               float *__restrict__ first_b, float2 *__restrict__ ext_w,
                                                               double *__restrict__ peak, float *__restrict__ peaksum, float *__restrict__ peak_b, int pk_from) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = wave_uniform(tid >> 6);
    cpx *s_xch = reinterpret_cast<cpx *>(smem);
    float *s_mag = reinterpret_cast<float *>(smem + kR2LdsXchg);                       // [round parity][frame][row][kR2MagPitch]
    float *s_ex = reinterpret_cast<float *>(smem + kR2LdsXchg + kR2LdsMag);           // [averaging wave][max | min][4 frames][64 lanes]
    float2 *s_part = reinterpret_cast<float2 *>(smem + kR2LdsXchg + kR2LdsMag + kR2LdsExt);   // [round parity][frame][averaging wave]
    const int pair = blockIdx.x, npairs = gridDim.x, F = g.F;
    const int nrounds = (nf + kR2Round - 1) / kR2Round;
    if (w >= kR2AWaves) {
        // ================= transforming waves: lane = (sub = (frame parity, row), j); row x[j + 16 m] -> bins ka + 16 kb in lane ka
        const int tw = w - kR2AWaves, sub = lane >> 4, j = lane & 15, fp = sub >> 1, rho = sub & 1;
        cpx *xs = s_xch + (size_t)(tw * 4 + sub) * kR2SubPitch;
        cpx w256[16];                                              // W256^(j ka) = exp(-2 pi i 16 j ka / 4096), ka = 1 .. 15 (j ka <= 225)
#pragma unroll
        for (int ka = 1; ka < 16; ++ka) w256[ka] = cpx_from(tw4096[16 * j * ka]);
        // byte offset of this lane's first element inside the round's first frame: frame (2 tw + fp), row 2 pair + rho, element j
        const unsigned lane_off = (unsigned)(((2 * pair + rho) * kS3R + j) * (int)sizeof(float2));      // row 2 pair + rho, element j: byte offset inside a frame
        const int my_f = 2 * tw + fp;                              // frame of the round this lane works on
        cpx nx[16];
        auto request = [&](int round) {                            // the rows of `round` (frames past the end re-read the batch's last frame)
            const int f = min(round * kR2Round + my_f, nf - 1);
            const char *base = reinterpret_cast<const char *>(Z) + (int64_t)f * (int64_t)(kS3N * sizeof(float2)) + lane_off;
#pragma unroll
            for (int m = 0; m < 16; ++m) nx[m] = cpx_from(*reinterpret_cast<const float2 *>(base + m * 128));
        };
        request(0);
        for (int it = 0; it <= nrounds; ++it) {
            if (it < nrounds) {
                cpx v[16];
#pragma unroll
                for (int m = 0; m < 16; ++m) v[m] = nx[m];
                if (it + 1 < nrounds) request(it + 1);
                cpx_dft<16>(v);                                    // over m: Y[ka]
#pragma unroll
                for (int ka = 1; ka < 16; ++ka) v[ka] = cpx_mul(v[ka], w256[ka]);
#pragma unroll
                for (int ka = 0; ka < 16; ++ka) xs[ka * kR2RowPitch + j] = v[ka];
                wave_sync();
                {   // lane ka = j picks up its row: sixteen consecutive complex, eight 16-byte reads
                    const float4 *rp = reinterpret_cast<const float4 *>(xs + j * kR2RowPitch);
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const float4 q = rp[u]; v[2 * u] = cpx_make(q.x, q.y); v[2 * u + 1] = cpx_make(q.z, q.w); }
                }
                cpx_dft<16>(v);                                    // over j: X[ka + 16 kb], kb = register
                float *mg = s_mag + (size_t)(((it & 1) * kR2Round + my_f) * 2 + rho) * kR2MagPitch + j;
#pragma unroll
                for (int kb = 0; kb < 16; ++kb) mg[16 * kb] = cpx_abs(v[kb]);
            }
            __syncthreads();
        }
        return;
    }
    // ================= averaging waves: this thread's display point, bins ka = 2 pair + 512 tid and ka + 1
    const int ka = 2 * pair + kS3C * tid;
    const int x = (int)(((ka - kS3N / 2) & (kS3N - 1)) >> 1);
    const int64_t si = spec_state_index(g, x);
    AvgState s = {ma[si], maa[si], ma[F + si], maa[F + si]};
    double pka = 0.0, pkb = 0.0;
    if constexpr (HOLD) { pka = peak[x]; pkb = peak[F + x]; }
    float *ex = s_ex + (size_t)w * 2 * kR2ExtFrames * 64;
    auto publish = [&](int r) {                                    // the per-frame extrema of round r, left in LDS one barrier ago
        const int nfr = min(kR2Round, nf - r * kR2Round);
        if (tid < nfr) {
            const float2 *e = s_part + (size_t)((r & 1) * kR2Round + tid) * kR2AWaves;
            float mx = 0.f, mn = 3.0e38f;                          // the starting values of spec_average's tiles
#pragma unroll
            for (int q = 0; q < kR2AWaves; ++q) { mx = fmaxf(mx, e[q].x); mn = fminf(mn, e[q].y); }
            ext_w[(int64_t)(r * kR2Round + tid) * npairs + pair] = make_float2(mx, mn);
        }
    };
    for (int it = 0; it <= nrounds; ++it) {
        if (it >= 2) publish(it - 2);
        if (it >= 1) {
            const int r = it - 1, fb = r * kR2Round, nfr = min(kR2Round, nf - fb);
            const float *mg = s_mag + (size_t)(r & 1) * kR2Round * 2 * kR2MagPitch + tid;
            float2 *part = s_part + (size_t)(r & 1) * kR2Round * kR2AWaves;
            // the magnitudes of the whole round first (independent LDS reads), then the recurrence -- the only serial chain -- frame after frame
            float xa[kR2Round], xb[kR2Round];
#pragma unroll
            for (int i = 0; i < kR2Round; ++i) { xa[i] = mg[(2 * i) * kR2MagPitch]; xb[i] = mg[(2 * i + 1) * kR2MagPitch]; }
            const AvgState s_in = s;
            float ps[kR2Round], mxs[kR2Round], mns[kR2Round], fbv[kR2Round];
            float fav[HOLD ? kR2Round : 1];                                                // HOLD: the first bin's rounded value too (the second is fbv)
#pragma unroll
            for (int i = 0; i < kR2Round; ++i) {
                if (i < nfr) avg_step_fast(s, (double)xa[i], (double)xb[i], rate);        // (block-uniform guard)
                const float fa = (float)s.maa_a, fbb = (float)s.maa_b;                     // float rounding is monotonic: extrema of the rounded values
                ps[i] = (float)(s.maa_a + s.maa_b); fbv[i] = fbb;
                if constexpr (HOLD) fav[i] = fa;
                mxs[i] = fmaxf(fa, fbb); mns[i] = fminf(fa, fbb);
            }
            // A magnitude that is not finite (a NaN / Inf IQ sample), or a NaN state entering the round, leaves a state that is not finite at the
            // end of the plain recurrence (NaN and Inf absorb): then -- never on a finite stream -- the round is run again with the reference's
            // statements, NaN repairs included (:494-497); on finite values both forms are the same arithmetic
            const bool odd = ((s.ma_a - s.ma_a) != 0.0) | ((s.maa_a - s.maa_a) != 0.0) | ((s.ma_b - s.ma_b) != 0.0) | ((s.maa_b - s.maa_b) != 0.0);
            if (wave_any(odd)) {
                s = s_in;
#pragma unroll
                for (int i = 0; i < kR2Round; ++i) {
                    if (i < nfr) avg_step(s, (double)xa[i], (double)xb[i], rate);
                    const float fa = (float)s.maa_a, fbb = (float)s.maa_b;
                    ps[i] = (float)(s.maa_a + s.maa_b); fbv[i] = fbb;
                    if constexpr (HOLD) fav[i] = fa;
                    mxs[i] = fmaxf(fa, fbb); mns[i] = fminf(fa, fbb);                       // (fmaxf / fminf skip a NaN operand, as the reference's comparisons do)
                }
            }
            if constexpr (HOLD) {
                // the held maxima, frame after frame (spec_peak_track: a NaN never replaces a held value -- the comparison is false)
#pragma unroll
                for (int i = 0; i < kR2Round; ++i) {
                    const int f = fb + i;
                    if (i < nfr && f >= pk_from) {
                        if ((double)fav[i] > pka) pka = (double)fav[i];
                        if ((double)fbv[i] > pkb) pkb = (double)fbv[i];
                        stf(peaksum + (int64_t)f * F, (unsigned)(pair * kS3R + tid) * 4u, (float)(pka + pkb));
                        if (x == 0) peak_b[f] = (float)pkb;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < kR2Round; ++i) {
                if (i < nfr) {
                    const int f = fb + i;
                    stf(pairsum + (int64_t)f * F, (unsigned)(pair * kS3R + tid) * 4u, ps[i]);
                    if (x == 0) first_b[f] = fbv[i];
                }
            }
            // per-frame extrema over the 64 lanes, four frames at a time: transposed through LDS, lane = (frame q, sixteenth p) folds four lanes'
            // values, then a 16-lane row reduction (spec_average's scheme)
#pragma unroll
            for (int i0 = 0; i0 < kR2Round; i0 += kR2ExtFrames) {
#pragma unroll
                for (int q = 0; q < kR2ExtFrames; ++q) { ex[q * 64 + lane] = mxs[i0 + q]; ex[(kR2ExtFrames + q) * 64 + lane] = mns[i0 + q]; }
                wave_sync();
                const int q = lane >> 4, p16 = lane & 15;
                const float4 vx = *reinterpret_cast<const float4 *>(ex + q * 64 + 4 * p16);
                const float4 vn = *reinterpret_cast<const float4 *>(ex + (kR2ExtFrames + q) * 64 + 4 * p16);
                float mx = fmaxf(fmaxf(vx.x, vx.y), fmaxf(vx.z, vx.w)), mn = fminf(fminf(vn.x, vn.y), fminf(vn.z, vn.w));
                row16_max_min(mx, mn);
                if (p16 == 0) part[(i0 + q) * kR2AWaves + w] = make_float2(mx, mn);
                wave_sync();
            }
        }
        __syncthreads();
    }
    publish(nrounds - 1);
    ma[si] = s.ma_a; maa[si] = s.maa_a; ma[F + si] = s.ma_b; maa[F + si] = s.maa_b;
    if constexpr (HOLD) { peak[x] = pka; peak[F + x] = pkb; }
}

// ---- K16 for the pair-row order of spec_rows256_ema: pairsum[f][pair][k2], display point x = (pair + 256 k2 - N / 4) mod F.
// grid = (8 x 8 tiles of 32 pairs x 32 k2, frames); reads 32 runs of 128 bytes, writes 32 runs of 128 bytes.
// (carry_n > 0: the workgroups of frame 0 also move the samples behind the batch's last whole frame to the carry buffer -- the last launch of a call's chain
//  takes the copy that was a 4 us transfer of its own; nothing of this launch reads either buffer)
// HOLD: frames >= pk_from also form spectrum_hold_points from the held sums, with the frame's own scalars (:539-556)
template <bool CARRY /* the short-batch instance that takes the carry copy along */, bool HOLD = false>
CSDR_KERNEL __launch_bounds__(kDispThreads) void spec_display_p256(const float *__restrict__ pairsum, const float *__restrict__ first_b,
                                                                  const SpecFrameScal *__restrict__ fsc, SpecGeom g, float sf, float *__restrict__ points,
                                                                  const float2 *__restrict__ carry_src, float2 *__restrict__ carry_dst, int carry_n,
                                                                  const float *__restrict__ peaksum = nullptr, const float *__restrict__ peak_b = nullptr,
                                                                  float *__restrict__ hold_points = nullptr, int pk_from = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (CARRY) if (carry_n > 0 && blockIdx.y == 0)
        for (int i = (int)(blockIdx.x * kDispThreads + threadIdx.x); i < carry_n; i += (int)(gridDim.x * kDispThreads)) carry_dst[i] = carry_src[i];
    float *s_y = reinterpret_cast<float *>(smem);                    // [32 k2][33]
    const int f = blockIdx.y, tid = threadIdx.x, F = g.F;
    const int npairs = kS3C / 2;
    const int p0 = (blockIdx.x & 7) * 32, t0 = (blockIdx.x >> 3) * 32;
    const SpecFrameScal sc = fsc[f];
    const double pf = sc.pf, fl = sc.fl;
    const float inv_den = 1.0f / log1pf((float)(sc.pc - pf));
    float a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = (tid >> 5) + 8 * u, j = tid & 31;              // pair p0 + i, k2 t0 + j
        a[u] = pairsum[(int64_t)f * F + (int64_t)(p0 + i) * kS3R + t0 + j];
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
    if constexpr (HOLD) {
        if (f < pk_from) return;                                     // (block-uniform)
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (tid >> 5) + 8 * u, j = tid & 31;
            a[u] = peaksum[(int64_t)f * F + (int64_t)(p0 + i) * kS3R + t0 + j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (tid >> 5) + 8 * u, j = tid & 31;
            const int x = (p0 + i + npairs * (t0 + j) - (int)(g.N >> 2)) & (F - 1);
            const double pacc = (x == 0) ? fl + (double)peak_b[f] : (double)a[u];
            s_y[j * 33 + i] = log1p_fast((float)(pacc * 0.5 - pf)) * inv_den * sf;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = (tid >> 5) + 8 * u, i = tid & 31;
            const int x = (p0 + i + npairs * (t0 + j) - (int)(g.N >> 2)) & (F - 1);
            st_stream(hold_points + (int64_t)f * F + x, s_y[j * 33 + i]);
        }
    }
}

}  // namespace csdr
