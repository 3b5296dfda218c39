// kernels_chanfft.hpp -- K2 + K4 for every even channel count whose factors are small (M = 4, 8, 20, 40, 112, 200, 1024 ...):
// the critically sampled polyphase analysis bank with a real mixed-radix FFT.
//
// Replaces (reference file:line): firpfbch_crcf_analyzer_execute SDRPostThread.cpp:449-451 (liquid firpfbch: Kaiser prototype m = 4,
// As = 60, :406; its M-point transform is liquid's own mixed-radix FFT plan) and the strided channel gather :364-381.
//
//   X_t[c] = sum_{n<8} taps[c][n] x[(t-n) M + c];   y_t[k] = sum_c X_t[c] exp(-j 2 pi k c / M);   out[k][t]
//
// Until round 4 these channel counts ran chan_analyze (kernels_post.hpp): ONE Cooley-Tukey split M = A B with direct A- and B-point
// DFTs, M (A + B) complex MACs per frame -- 64 per sample at M = 1024, 0.11 of the HBM roofline.  Here the transform is an in-place
// decimation-in-frequency FFT over radices 16 / 8 / 4 / 2 and odd 3 / 5 / 7 / 9 / 11 / 13 (17 / 19 / 23 in a second instance; M = 1024 = 16 * 8 * 8: ~9 real operations per
// sample and pass), and the tile is TRANSPOSED in LDS: X[c][t], t contiguous.
//  * Persistent workgroups walk over tiles of TF consecutive frames (TF a power of two, 8 .. 256: the largest whose tile fits the LDS budget).
//  * FIR: a work item owns a column PAIR (one float4) and eight consecutive frames; the fifteen input rows they reach go straight
//    from global memory into registers (lanes along the columns: every load of a wave is one contiguous run; neighbouring items share
//    seven rows: cache hits), the taps of its two columns sit in registers, and each of its two rows of X receives 64 contiguous bytes
//    (four ds_write_b128; row pitch TF + 2 samples: lanes two rows apart are 8 banks apart).
//  * FFT pass p (radix R, span s, block N = R s): work item = (butterfly, frame), lanes along the frames, so the R operands of a lane
//    are R row reads at a pitch of s rows, conflict-free, and with TF >= 64 the butterfly -- hence its twiddles W_N^(j r), read from an
//    M-entry table in LDS -- is wave-uniform (a broadcast read).  The R-point DFT runs in registers; results go back in place.
//  * The last pass does not write LDS: output r of the butterfly at position pos is channel k = perm[pos + r] (the mixed-radix digit
//    reversal, a table) and goes straight to out[k][f0 + t]: with the lanes along t every store instruction writes runs of 8 TF bytes
//    of one channel row.  Rows without consumers are skipped (SDRPostThread.cpp:336-339).
//  * Channel 0 of the tile is left in LDS for the DC blocker's per-tile end value (as chan_analyze does).
// LDS traffic per input sample: one 8-byte write + read per pass (three at M = 1024); arithmetic ~ 16 FIR + 9 * passes operations.
#pragma once
#include "common.hpp"
#include "kernels_post.hpp"

namespace csdr {

constexpr int kCfMaxPasses = 8;
constexpr int kCfSeg = 8;                 // frames one FIR work item produces
constexpr int kCfMaxThreads = 1024;
#ifndef CSDR_CF_PRIO_FIR
#define CSDR_CF_PRIO_FIR 2
#endif
#ifndef CSDR_CF_PRIO_PASS
#define CSDR_CF_PRIO_PASS 1
#endif
constexpr int kCfPrioFir = CSDR_CF_PRIO_FIR, kCfPrioPass = CSDR_CF_PRIO_PASS;      // (A/B builds: -DCSDR_CF_PRIO_FIR=0 -DCSDR_CF_PRIO_PASS=0 is the round-5 kernel)

struct ChanFftGeom {
    int M, TF, lgTF, TFs;                 // TFs: row pitch of X in samples (TF + 2)
    int threads;                          // workgroup size (whole waves)
    int npass;
    int radix[kCfMaxPasses];              // R_p;  N_p = M / (R_0 .. R_{p-1}) is the block size of pass p
    int span[kCfMaxPasses];               // s_p = N_p / R_p
    unsigned magic_span[kCfMaxPasses];    // floor(2^32 / s_p) + 1 (s_p > 1)
    int twstep[kCfMaxPasses];             // M / N_p: W_{N_p}^(j r) = W_M^(j r twstep)
    unsigned magic_half;                  // floor(2^32 / (M / 2)) + 1 (M > 2)
    int xcd;                              // 1: workgroups of one XCD (blockIdx.x % 8) take CONSECUTIVE tiles (grid a multiple of 8)
    int wide_odd;                         // 1: a radix of 17 / 19 / 23 is in the plan: the kernel instance that carries those butterflies
    int os2;                              // 1: firpfbch2 -- frames hop by M / 2 (two interleaved lattices of frames), outputs times the post factors
    // a prime factor >= 29 (M = 116, 134, 146 ... 398: a third of the even channel counts getOptimalChannelCount returns, SoapySDRThread.cpp:676-693):
    // pass 0 is its bp-point transform, run as a chirp-z convolution of length bL (the power of two >= 2 bp - 1) in a work array beside the tile
    int bp, bL, blgL;                     // 0: no such factor
    int bnpass, bradix[4], bspan[4];      // the bL-point transform: radices 16 / 8 / 4 / 2, span of sub-pass k inside bL
    // a prime factor 29 .. 199: pass 0 is its direct transform in the conjugate-pair form of chan_analyze_p2's transform phase, out of place into a second
    // tile -- on the fp32 matrix pipe (cf_prime_pass_mx; the vector form -- lane = (column, frame), wave = four output pairs, (cos, sin) rows wave-uniform -- is
    // kept for A/B builds)
    int dp, dnk, dPA;                     // the prime (0: none), groups of four output-pair slots, pitch of a (cos, sin) row
    unsigned magic_s0;                    // floor(2^32 / (M / dp)) + 1
};
constexpr int kCfDirectKP = 4;
#ifndef CSDR_CF_PRIME_MX
#define CSDR_CF_PRIME_MX 1
#endif
constexpr bool kCfPrimeMx = CSDR_CF_PRIME_MX != 0;      // the direct prime pass on the fp32 matrix pipe (A/B builds: -DCSDR_CF_PRIME_MX=0 is the vector form)
// (measured, profiles/r06_chirpz_channel_counts.txt: the convolution costs 2.2 - 2.6 x the factor's own data in LDS work space and six trips through it;
//  against the VECTOR form of the direct prime pass it won from p ~ 157 on.  Against the matrix-pipe form of that pass (cf_prime_pass_mx,
//  profiles/r06_prime_mx.txt) it loses for every prime the fragments' registers reach -- M = 314: 1.27 against 0.67 ms, M = 398: 1.06 against 0.73 ms --
//  so it now starts at p = 211, i.e. M >= 422: beyond the counts getOptimalChannelCount returns for rates up to 200 MS/s)
#ifndef CSDR_CF_BLUE_MIN
#define CSDR_CF_BLUE_MIN 211
#endif
constexpr int kCfBlueMinPrime = CSDR_CF_BLUE_MIN, kCfBlueMaxPrime = 509;

__host__ __device__ inline size_t chanfft_lds_bytes(const ChanFftGeom &g) {
    // (oversampled: + the M post factors; chirp-z pass: + the work array of M / bp transforms of bL points per frame, W_bL, the transformed chirp, the chirp)
    const size_t blue = g.bp ? (size_t)(g.M / g.bp) * g.bL * g.TFs + 2 * (size_t)g.bL + g.bp : g.dp ? (size_t)g.M * g.TFs : 0;      // (direct prime pass: the second tile)
    return ((size_t)g.M * g.TFs + g.M + g.TF + (g.os2 ? g.M : 0) + blue) * sizeof(float2) + (size_t)g.M * sizeof(int);
}

// plan for M channels: radices (odd ones first, then the powers of two from the widest), tile size, workgroup size.
// Returns false when M has a prime factor this kernel has no butterfly for (chan_analyze takes those).
// (force_tf / force_threads: measurement overrides, 0 = automatic)
__host__ inline bool chanfft_plan(int M, size_t lds_limit, int force_tf, int force_threads, ChanFftGeom &g, std::vector<int> &perm, bool os2 = false) {
    memset(&g, 0, sizeof g);
    g.M = M; g.os2 = os2 ? 1 : 0;
    if (os2 && (M & 3)) return false;             // the half-frame offset M / 2 of the second lattice must keep the column pairs 16-byte aligned
    if (M < 2 || (M & 1) || M > 65536) return false;
    std::vector<int> rad;
    int m = M, e2 = 0, e3 = 0;
    while (!(m & 1)) { m >>= 1; ++e2; }
    while (m % 3 == 0) { m /= 3; ++e3; }
    for (; e3 >= 2; e3 -= 2) rad.push_back(9);
    if (e3) rad.push_back(3);
    for (int p : {5, 7, 11, 13, 17, 19, 23}) while (m % p == 0) { m /= p; rad.push_back(p); if (p >= 17) g.wide_odd = 1; }
    if (m != 1) {
        // what is left: ONE prime >= 29 can go through the chirp-z pass (critically sampled bank, no 17 / 19 / 23 beside it: one kernel instance)
        bool prime = m >= 29 && m <= kCfBlueMaxPrime;
        for (int d = 3; prime && d * d <= m; d += 2) prime = m % d != 0;
        if (!prime || os2 || g.wide_odd) return false;
        if (m < kCfBlueMinPrime) {                // the direct pass
            g.dp = m;
            const int H = (m - 1) / 2;
            g.dnk = (H + 1 + kCfDirectKP - 1) / kCfDirectKP; g.dPA = g.dnk * kCfDirectKP;
            g.magic_s0 = (unsigned)((1ull << 32) / (unsigned)(M / m)) + 1u;
        } else g.bp = m;
        g.blgL = 0; while ((1 << g.blgL) < 2 * m - 1) ++g.blgL;
        g.bL = 1 << g.blgL;
        g.bnpass = (g.blgL + 3) / 4;
        int N = g.bL;
        for (int k = 0; k < g.bnpass; ++k) { const int lg = g.blgL / g.bnpass + (k < g.blgL % g.bnpass ? 1 : 0); g.bradix[k] = 1 << lg; g.bspan[k] = N >> lg; N >>= lg; }
        rad.insert(rad.begin(), m);               // pass 0
    }
    const int n2 = (e2 + 3) / 4;                               // passes over the power of two: as even as possible, widest first
    for (int i = 0; i < n2; ++i) rad.push_back(1 << (e2 / n2 + (i < e2 % n2 ? 1 : 0)));
    if ((int)rad.size() > kCfMaxPasses) return false;
    g.npass = (int)rad.size();
    int N = M;
    for (int p = 0; p < g.npass; ++p) {
        g.radix[p] = rad[p]; g.span[p] = N / rad[p]; g.twstep[p] = M / N;
        g.magic_span[p] = g.span[p] > 1 ? (unsigned)((1ull << 32) / (unsigned)g.span[p]) + 1u : 0u;
        N /= rad[p];
    }
    g.magic_half = M > 2 ? (unsigned)((1ull << 32) / (unsigned)(M / 2)) + 1u : 0u;
    // tile and workgroup size, from the sweep in profiles/r04_chan_geometry.txt (11 channel counts x 15 geometries on MI355X, working sets beyond the
    // Infinity Cache): about 5 K samples per tile up to M = 200 (several workgroups per CU overlap their phases), 16 K samples from M = 256 on (the
    // runs a store instruction writes are 8 TF bytes: 512 at M = 256, 128 at M = 1024, where the 160 KB of LDS end the choice); one thread per FIR
    // work item, a power of two (320- and 448-thread workgroups measured 10 - 25 % slower than 256 / 512)
    auto fits = [&](int tf) { g.TF = tf; g.TFs = tf + 2; return chanfft_lds_bytes(g) <= lds_limit; };
    const int target = M >= 256 ? 16384 : 5120;
    int tf = 16;
    if (g.bp) { g.radix[0] = g.bp; g.span[0] = M / g.bp; }        // (chanfft_lds_bytes of a chirp-z plan)
    while (tf < 256 && 141 * tf * M <= 100 * target) tf <<= 1;    // the power of two nearest target / M (on a log scale)
    while (tf > (os2 ? 2 * kCfSeg : kCfSeg) && !fits(tf)) tf >>= 1;      // (oversampled: eight frames of EACH lattice per tile at least)
    bool dp_wide = false;
    if (g.dp) {
        // direct prime pass (two sweeps in profiles/r06_prime_channel_counts.txt): the largest tile that leaves two workgroups per CU; where that fills less
        // than a wave with (column, frame) lanes, twice the tile and 1024 threads instead
        tf = 32;
        while (tf > kCfSeg && (!fits(tf) || chanfft_lds_bytes(g) > 80 * 1024)) tf >>= 1;
        if ((M / g.dp) * tf < 64 && fits(2 * tf)) { tf *= 2; dp_wide = true; }
    }
    if (g.bp) { tf = M >= 280 ? 16 : 8; while (tf > kCfSeg && !fits(tf)) tf >>= 1; }      // chirp-z plans: small tiles, several workgroups per CU (sweep in profiles/r06_chirpz_channel_counts.txt)
    if (force_tf >= kCfSeg && !(force_tf & (force_tf - 1)) && fits(force_tf)) tf = force_tf;
    if (!fits(tf)) return false;
    g.lgTF = 0; while ((1 << g.lgTF) < tf) ++g.lgTF;
    if (os2 && tf < 2 * kCfSeg) return false;
    const int fir_items = (M / 2) * (tf / kCfSeg);
    g.threads = 256;
    while (g.threads < kCfMaxThreads && g.threads < fir_items) g.threads <<= 1;
    if (g.bp) g.threads = M >= 280 ? 1024 : 512;
    if (g.dp) g.threads = dp_wide ? 1024 : 512;
    if (force_threads >= 64 && force_threads <= kCfMaxThreads && !(force_threads & 63)) g.threads = force_threads;
    g.xcd = M >= 64;             // (C4: + 5 %, M = 20: nothing)
    // pos = sum_p r_p s_p holds channel k = r_0 + R_0 (r_1 + R_1 (r_2 + ...)) after the last pass
    perm.assign(M, 0);
    for (int pos = 0; pos < M; ++pos) {
        int k = 0, w = 1, rest = pos;
        for (int p = 0; p < g.npass; ++p) { const int r = rest / g.span[p]; rest -= r * g.span[p]; k += r * w; w *= g.radix[p]; }
        perm[pos] = k;
    }
    return true;
}

// (cos, sin)(2 pi k(q) c / p) of the direct prime pass at [(c - 1) dPA + q], c = 1 .. (p - 1) / 2; slot q: k = q + 1 (q < H), k = 0 (q == H: (1, 0)), (0, 0) beyond
__host__ inline std::vector<float2> chanfft_direct_tables(const ChanFftGeom &g) {
    const int p = g.dp, H = (p - 1) / 2;
    std::vector<float2> t((size_t)H * g.dPA, make_float2(0.f, 0.f));
    for (int c = 1; c <= H; ++c) for (int q = 0; q <= H; ++q) {
        const int k = q < H ? q + 1 : 0;
        const double a = 2.0 * M_PI * (double)(((int64_t)c * k) % p) / (double)p;
        t[(size_t)(c - 1) * g.dPA + q] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    return t;
}

// The same pass on the fp32 matrix pipe (cf_prime_pass_mx): coefficient fragments [2 (cos | sin)][row tile][step][64 lanes] -- lane l of step J holds the
// coefficient of output k = 16 rt + (l & 15) and term n = 4 J + (l >> 4), the A operand of v_mfma_f32_16x16x4_f32; term 0 is x_0 (cos = 1, sin = 0),
// outputs and terms past H are zero.  The angles are the vector form's expression: the products are the same products.
__host__ __device__ inline int chanfft_mx_row_tiles(int p) { return ((p - 1) / 2 + 16) / 16; }      // outputs k = 0 .. H
__host__ __device__ inline int chanfft_mx_steps(int p) { return ((p - 1) / 2 + 4) / 4; }            // terms n = 0 .. H, four per step
constexpr int kCfMxMaxSteps = 25;                                                                   // p <= 199
__host__ inline std::vector<float2> chanfft_direct_mx_tables(const ChanFftGeom &g) {
    const int p = g.dp, H = (p - 1) / 2, RT = chanfft_mx_row_tiles(p), KS = chanfft_mx_steps(p);
    std::vector<float> t((size_t)2 * RT * KS * 64, 0.f);
    for (int kind = 0; kind < 2; ++kind) for (int rt = 0; rt < RT; ++rt) for (int J = 0; J < KS; ++J) for (int l = 0; l < 64; ++l) {
        const int k = 16 * rt + (l & 15), n = 4 * J + (l >> 4);
        float v = 0.f;
        if (k <= H && n <= H) {
            const double a = 2.0 * M_PI * (double)(((int64_t)n * k) % p) / (double)p;
            v = kind == 0 ? (n == 0 ? 1.0f : (float)std::cos(a)) : (n == 0 ? 0.0f : (float)std::sin(a));
        }
        t[(((size_t)kind * RT + rt) * KS + J) * 64 + l] = v;
    }
    std::vector<float2> r((t.size() + 1) / 2);
    memcpy(r.data(), t.data(), t.size() * sizeof(float));
    return r;
}

// tables of the chirp-z pass, in double on the host: W_L^i (L), the transformed chirp filter at the positions the forward sub-passes leave the
// frequencies in, already divided by L (L), the chirp c[n] = exp(-j pi n^2 / p) (p)
__host__ inline std::vector<float2> chanfft_blue_tables(const ChanFftGeom &g) {
    const int p = g.bp, L = g.bL;
    std::vector<float2> t((size_t)2 * L + p);
    for (int i = 0; i < L; ++i) { const double a = -2.0 * M_PI * (double)i / (double)L; t[(size_t)i] = make_float2((float)std::cos(a), (float)std::sin(a)); }
    std::vector<double> cr((size_t)p), ci((size_t)p);
    for (int n = 0; n < p; ++n) {
        const double a = -M_PI * (double)(((int64_t)n * n) % (2 * p)) / (double)p;
        cr[(size_t)n] = std::cos(a); ci[(size_t)n] = std::sin(a);
        t[(size_t)2 * L + n] = make_float2((float)cr[(size_t)n], (float)ci[(size_t)n]);
    }
    // b[m mod L] = conj(c[|m|]), |m| < p;  B = DFT_L(b): b is even, so B[k] = b[0] + 2 sum_{m=1}^{p-1} b[m] cos(2 pi k m / L)
    std::vector<double> Br((size_t)L), Bi((size_t)L);
    for (int k = 0; k < L; ++k) {
        double sr = cr[0], si = -ci[0];
        for (int m = 1; m < p; ++m) { const double w = 2.0 * std::cos(2.0 * M_PI * (double)(((int64_t)k * m) % L) / (double)L); sr += w * cr[(size_t)m]; si -= w * ci[(size_t)m]; }
        Br[(size_t)k] = sr / L; Bi[(size_t)k] = si / L;
    }
    for (int pos = 0; pos < L; ++pos) {            // position after the forward sub-passes -> frequency (the digit reversal of the sub-plan)
        int k = 0, w = 1, rest = pos;
        for (int q = 0; q < g.bnpass; ++q) { const int r = rest / g.bspan[q]; rest -= r * g.bspan[q]; k += r * w; w *= g.bradix[q]; }
        t[(size_t)L + pos] = make_float2((float)Br[(size_t)k], (float)Bi[(size_t)k]);
    }
    return t;
}

// ---- R-point forward DFTs in registers: v[r] <- sum_q v[q] exp(-j 2 pi q r / R)
__device__ __forceinline__ float2 cf_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 cf_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cf_mj(float2 a) { return make_float2(a.y, -a.x); }        // a * (-j)
__device__ __forceinline__ float2 cf_mulc(float2 a, float c, float s) {                      // a * (c + j s), constants
    return make_float2(fmaf(a.x, c, -a.y * s), fmaf(a.x, s, a.y * c));
}
__device__ __forceinline__ void cf_dft2(float2 &a, float2 &b) { const float2 t = cf_sub(a, b); a = cf_add(a, b); b = t; }
__device__ __forceinline__ void cf_dft4(float2 &a, float2 &b, float2 &c, float2 &d) {
    const float2 t0 = cf_add(a, c), t1 = cf_sub(a, c), t2 = cf_add(b, d), t3 = cf_mj(cf_sub(b, d));
    a = cf_add(t0, t2); c = cf_sub(t0, t2); b = cf_add(t1, t3); d = cf_sub(t1, t3);
}
template <int R> struct CfOdd;
template <> struct CfOdd<3> {
    static constexpr float c[3] = {1.000000000e+00f, -5.000000000e-01f, -5.000000000e-01f};
    static constexpr float s[3] = {0.000000000e+00f, 8.660254038e-01f, -8.660254038e-01f};
};
template <> struct CfOdd<5> {
    static constexpr float c[5] = {1.000000000e+00f, 3.090169944e-01f, -8.090169944e-01f, -8.090169944e-01f, 3.090169944e-01f};
    static constexpr float s[5] = {0.000000000e+00f, 9.510565163e-01f, 5.877852523e-01f, -5.877852523e-01f, -9.510565163e-01f};
};
template <> struct CfOdd<7> {
    static constexpr float c[7] = {1.000000000e+00f, 6.234898019e-01f, -2.225209340e-01f, -9.009688679e-01f, -9.009688679e-01f, -2.225209340e-01f, 6.234898019e-01f};
    static constexpr float s[7] = {0.000000000e+00f, 7.818314825e-01f, 9.749279122e-01f, 4.338837391e-01f, -4.338837391e-01f, -9.749279122e-01f, -7.818314825e-01f};
};
template <> struct CfOdd<9> {
    static constexpr float c[9] = {1.000000000e+00f, 7.660444431e-01f, 1.736481777e-01f, -5.000000000e-01f, -9.396926208e-01f, -9.396926208e-01f, -5.000000000e-01f, 1.736481777e-01f, 7.660444431e-01f};
    static constexpr float s[9] = {0.000000000e+00f, 6.427876097e-01f, 9.848077530e-01f, 8.660254038e-01f, 3.420201433e-01f, -3.420201433e-01f, -8.660254038e-01f, -9.848077530e-01f, -6.427876097e-01f};
};
template <> struct CfOdd<11> {
    static constexpr float c[11] = {1.000000000e+00f, 8.412535328e-01f, 4.154150130e-01f, -1.423148383e-01f, -6.548607339e-01f, -9.594929736e-01f, -9.594929736e-01f, -6.548607339e-01f, -1.423148383e-01f, 4.154150130e-01f, 8.412535328e-01f};
    static constexpr float s[11] = {0.000000000e+00f, 5.406408175e-01f, 9.096319954e-01f, 9.898214419e-01f, 7.557495744e-01f, 2.817325568e-01f, -2.817325568e-01f, -7.557495744e-01f, -9.898214419e-01f, -9.096319954e-01f, -5.406408175e-01f};
};
template <> struct CfOdd<13> {
    static constexpr float c[13] = {1.000000000e+00f, 8.854560257e-01f, 5.680647467e-01f, 1.205366803e-01f, -3.546048870e-01f, -7.485107482e-01f, -9.709418174e-01f, -9.709418174e-01f, -7.485107482e-01f, -3.546048870e-01f, 1.205366803e-01f, 5.680647467e-01f, 8.854560257e-01f};
    static constexpr float s[13] = {0.000000000e+00f, 4.647231720e-01f, 8.229838659e-01f, 9.927088741e-01f, 9.350162427e-01f, 6.631226582e-01f, 2.393156643e-01f, -2.393156643e-01f, -6.631226582e-01f, -9.350162427e-01f, -9.927088741e-01f, -8.229838659e-01f, -4.647231720e-01f};
};
// 17 / 19 / 23: the wide-odd instance of the kernel only (chan_analyze_fft<true>: M = 68, 76, 92, 136 ... -- channel counts getOptimalChannelCount
// returns between 34 and 46 MS/s and their multiples); their 2 R registers of operands stay out of the instance every other count runs
template <> struct CfOdd<17> {
    static constexpr float c[17] = {1.000000000e+00f, 9.324722294e-01f, 7.390089172e-01f, 4.457383558e-01f, 9.226835946e-02f, -2.736629901e-01f, -6.026346364e-01f, -8.502171357e-01f, -9.829730997e-01f, -9.829730997e-01f, -8.502171357e-01f, -6.026346364e-01f, -2.736629901e-01f, 9.226835946e-02f, 4.457383558e-01f, 7.390089172e-01f, 9.324722294e-01f};
    static constexpr float s[17] = {0.000000000e+00f, 3.612416662e-01f, 6.736956436e-01f, 8.951632914e-01f, 9.957341763e-01f, 9.618256432e-01f, 7.980172273e-01f, 5.264321629e-01f, 1.837495178e-01f, -1.837495178e-01f, -5.264321629e-01f, -7.980172273e-01f, -9.618256432e-01f, -9.957341763e-01f, -8.951632914e-01f, -6.736956436e-01f, -3.612416662e-01f};
};
template <> struct CfOdd<19> {
    static constexpr float c[19] = {1.000000000e+00f, 9.458172417e-01f, 7.891405094e-01f, 5.469481581e-01f, 2.454854871e-01f, -8.257934547e-02f, -4.016954247e-01f, -6.772815716e-01f, -8.794737512e-01f, -9.863613034e-01f, -9.863613034e-01f, -8.794737512e-01f, -6.772815716e-01f, -4.016954247e-01f, -8.257934547e-02f, 2.454854871e-01f, 5.469481581e-01f, 7.891405094e-01f, 9.458172417e-01f};
    static constexpr float s[19] = {0.000000000e+00f, 3.246994692e-01f, 6.142127127e-01f, 8.371664783e-01f, 9.694002659e-01f, 9.965844930e-01f, 9.157733267e-01f, 7.357239107e-01f, 4.759473930e-01f, 1.645945903e-01f, -1.645945903e-01f, -4.759473930e-01f, -7.357239107e-01f, -9.157733267e-01f, -9.965844930e-01f, -9.694002659e-01f, -8.371664783e-01f, -6.142127127e-01f, -3.246994692e-01f};
};
template <> struct CfOdd<23> {
    static constexpr float c[23] = {1.000000000e+00f, 9.629172873e-01f, 8.544194045e-01f, 6.825531432e-01f, 4.600650377e-01f, 2.034560131e-01f, -6.824241336e-02f, -3.348796122e-01f, -5.766803221e-01f, -7.757112907e-01f, -9.172113015e-01f, -9.906859460e-01f, -9.906859460e-01f, -9.172113015e-01f, -7.757112907e-01f, -5.766803221e-01f, -3.348796122e-01f, -6.824241336e-02f, 2.034560131e-01f, 4.600650377e-01f, 6.825531432e-01f, 8.544194045e-01f, 9.629172873e-01f};
    static constexpr float s[23] = {0.000000000e+00f, 2.697967712e-01f, 5.195839500e-01f, 7.308359643e-01f, 8.878852184e-01f, 9.790840877e-01f, 9.976687692e-01f, 9.422609221e-01f, 8.169698930e-01f, 6.310879443e-01f, 3.984010898e-01f, 1.361666491e-01f, -1.361666491e-01f, -3.984010898e-01f, -6.310879443e-01f, -8.169698930e-01f, -9.422609221e-01f, -9.976687692e-01f, -9.790840877e-01f, -8.878852184e-01f, -7.308359643e-01f, -5.195839500e-01f, -2.697967712e-01f};
};

template <int R> struct CfDft {
    // odd R, conjugate-pair form: s_c = v_c + v_{R-c}, d_c = v_c - v_{R-c};  P_k = v_0 + sum_c s_c cos(2 pi k c / R),
    // Q_k = sum_c d_c sin(2 pi k c / R);  y_k = P_k - j Q_k,  y_{R-k} = P_k + j Q_k
    static __device__ __forceinline__ void run(float2 (&v)[R]) {
        static_assert(R & 1, "odd radix");
        constexpr int H = (R - 1) / 2;
        float2 s[H + 1], d[H + 1];
        float2 y0 = v[0];
#pragma unroll
        for (int c = 1; c <= H; ++c) { s[c] = cf_add(v[c], v[R - c]); d[c] = cf_sub(v[c], v[R - c]); y0 = cf_add(y0, s[c]); }
        const float2 x0 = v[0];
        v[0] = y0;
#pragma unroll
        for (int k = 1; k <= H; ++k) {
            float2 P = x0, Q = make_float2(0.f, 0.f);
#pragma unroll
            for (int c = 1; c <= H; ++c) {
                const float co = CfOdd<R>::c[(k * c) % R], si = CfOdd<R>::s[(k * c) % R];
                P.x = fmaf(s[c].x, co, P.x); P.y = fmaf(s[c].y, co, P.y);
                Q.x = fmaf(d[c].x, si, Q.x); Q.y = fmaf(d[c].y, si, Q.y);
            }
            v[k] = make_float2(P.x + Q.y, P.y - Q.x);
            v[R - k] = make_float2(P.x - Q.y, P.y + Q.x);
        }
    }
};
template <> struct CfDft<2> { static __device__ __forceinline__ void run(float2 (&v)[2]) { cf_dft2(v[0], v[1]); } };
template <> struct CfDft<4> { static __device__ __forceinline__ void run(float2 (&v)[4]) { cf_dft4(v[0], v[1], v[2], v[3]); } };
template <> struct CfDft<8> {
    // a_q = v_q + v_{q+4}, b_q = (v_q - v_{q+4}) W_8^q;  y_{2m} = DFT4(a)_m,  y_{2m+1} = DFT4(b)_m
    static __device__ __forceinline__ void run(float2 (&v)[8]) {
        constexpr float h = 7.071067812e-01f;
        float2 a0 = cf_add(v[0], v[4]), a1 = cf_add(v[1], v[5]), a2 = cf_add(v[2], v[6]), a3 = cf_add(v[3], v[7]);
        float2 b0 = cf_sub(v[0], v[4]), b1 = cf_sub(v[1], v[5]), b2 = cf_mj(cf_sub(v[2], v[6])), b3 = cf_sub(v[3], v[7]);
        b1 = make_float2((b1.x + b1.y) * h, (b1.y - b1.x) * h);          // (1 - j) / sqrt 2
        b3 = make_float2((b3.y - b3.x) * h, -(b3.x + b3.y) * h);         // (-1 - j) / sqrt 2
        cf_dft4(a0, a1, a2, a3); cf_dft4(b0, b1, b2, b3);
        v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3; v[1] = b0; v[3] = b1; v[5] = b2; v[7] = b3;
    }
};
template <> struct CfDft<16> {
    // v[4 a + b]: u_b = DFT4 over a, times W_16^(b r1); y[r1 + 4 r2] = DFT4 over b of u[r1][.]
    static __device__ __forceinline__ void run(float2 (&v)[16]) {
        constexpr float h = 7.071067812e-01f, c1 = 9.238795325e-01f, s1 = 3.826834324e-01f;
#pragma unroll
        for (int b = 0; b < 4; ++b) cf_dft4(v[b], v[b + 4], v[b + 8], v[b + 12]);      // v[4 r1 + b] now holds u[r1][b]
        // W_16^m = (cos(m pi / 8), -sin(m pi / 8)):  m = b r1
        v[5] = cf_mulc(v[5], c1, -s1);                                                 // m = 1
        v[6] = make_float2((v[6].x + v[6].y) * h, (v[6].y - v[6].x) * h);              // m = 2
        v[7] = cf_mulc(v[7], s1, -c1);                                                 // m = 3
        v[9] = make_float2((v[9].x + v[9].y) * h, (v[9].y - v[9].x) * h);              // m = 2
        v[10] = cf_mj(v[10]);                                                          // m = 4
        v[11] = make_float2((v[11].y - v[11].x) * h, -(v[11].x + v[11].y) * h);        // m = 6
        v[13] = cf_mulc(v[13], s1, -c1);                                               // m = 3
        v[14] = make_float2((v[14].y - v[14].x) * h, -(v[14].x + v[14].y) * h);        // m = 6
        v[15] = cf_mulc(v[15], -c1, s1);                                               // m = 9
        float2 y[16];
#pragma unroll
        for (int r1 = 0; r1 < 4; ++r1) {
            cf_dft4(v[4 * r1], v[4 * r1 + 1], v[4 * r1 + 2], v[4 * r1 + 3]);
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2) y[r1 + 4 * r2] = v[4 * r1 + r2];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = y[i];
    }
};

__device__ __forceinline__ unsigned cf_div(unsigned n, int d, unsigned magic) { return d == 1 ? n : __umulhi(n, magic); }

// one butterfly of a pass that stays in LDS: R rows at pitch `pitch` samples, twiddles W_M^(r jj), in place
template <int R>
__device__ __forceinline__ void cf_pass_item(float2 *px, int pitch, const float2 *s_tw, int jj) {
    float2 v[R];
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = px[(size_t)q * pitch];
    if constexpr (R >= 17) {
        // the wide odd radices: 2 R operand registers and as many of sums and differences inside the butterfly already; the R - 1 twiddles are read
        // one by one where they are used instead of up front (all in registers, the instance spilled 52 - 61 of them)
        CfDft<R>::run(v);
        px[0] = v[0];
#pragma unroll
        for (int r = 1; r < R; ++r) px[(size_t)r * pitch] = cmul(v[r], s_tw[r * jj]);
    } else {
        float2 w[R];
#pragma unroll
        for (int r = 1; r < R; ++r) w[r] = s_tw[r * jj];
        CfDft<R>::run(v);
        px[0] = v[0];
#pragma unroll
        for (int r = 1; r < R; ++r) px[(size_t)r * pitch] = cmul(v[r], w[r]);
    }
}
// one butterfly of an INVERSE sub-pass of the chirp-z convolution, on conjugated data (conj(IDFT(u)) = DFT(conj(u)): the array stays conjugated from
// the first inverse sub-pass to the read-out): twiddle first, then the R-point transform -- the forward stage run backwards.  `bh` (first inverse
// sub-pass only): the transformed chirp filter at these positions; its product with the transformed data is what gets conjugated.
template <int R>
__device__ __forceinline__ void cf_ipass_item(float2 *px, int pitch, const float2 *s_wl, int jj, const float2 *bh, int bh_pitch) {
    float2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float2 u = px[(size_t)r * pitch];
        if (bh) { u = cmul(u, bh[(size_t)r * bh_pitch]); u.y = -u.y; }
        v[r] = r ? cmul(u, s_wl[r * jj]) : u;
    }
    CfDft<R>::run(v);
#pragma unroll
    for (int q = 0; q < R; ++q) px[(size_t)q * pitch] = v[q];
}
template <int R>
__device__ __forceinline__ void cf_blue_subpass(const ChanFftGeom &g, int k, bool inverse, float2 *s_x, float2 *s_ws, const float2 *s_tw, const float2 *s_wl, const float2 *s_bh,
                                                const float2 *s_c, int tid, int nthr) {
    const int TF = g.TF, TFs = g.TFs, L = g.bL, p = g.bp, s0 = g.M / p, span = g.bspan[k], lgspan = __builtin_ctz((unsigned)span), lgR = __builtin_ctz((unsigned)R);
    const int per = L >> lgR, lgper = g.blgL - lgR, items = (s0 * per) << g.lgTF, twstep = L / (R * span);
    for (int it = tid; it < items; it += nthr) {
        const int t = it & (TF - 1), rest = it >> g.lgTF, bf = rest & (per - 1), j = rest >> lgper;
        const int b = bf >> lgspan, jj = bf & (span - 1), pos0 = b * R * span + jj;
        float2 *px = s_ws + ((size_t)j * L + pos0) * TFs + t;
        const int pitch = span * TFs;
        if (k == 0 && !inverse) {
            // first forward sub-pass (one block: pos0 = jj): its operands are the tile's rows times the chirp, zeros past the factor's length -- read in place
            float2 v[R];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int n = jj + span * q;
                v[q] = n < p ? cmul(s_x[(size_t)(j + s0 * n) * TFs + t], s_c[n]) : make_float2(0.f, 0.f);
            }
            CfDft<R>::run(v);
            px[0] = v[0];
#pragma unroll
            for (int r = 1; r < R; ++r) px[(size_t)r * pitch] = cmul(v[r], s_wl[r * jj * twstep]);
        } else if (k == 0) {
            // last inverse sub-pass: its results are the convolution (conjugated): times the chirp and the pass's own twiddle, straight back into the tile
            float2 v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float2 u = px[(size_t)r * pitch];
                if (g.bnpass == 1) { u = cmul(u, s_bh[pos0 + r * span]); u.y = -u.y; }
                v[r] = r ? cmul(u, s_wl[r * jj * twstep]) : u;
            }
            CfDft<R>::run(v);
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int n = jj + span * q;
                if (n < p) s_x[(size_t)(j + s0 * n) * TFs + t] = cmul(make_float2(v[q].x, -v[q].y), cmul(s_c[n], s_tw[j * n]));      // j n < M
            }
        } else if (!inverse) cf_pass_item<R>(px, pitch, s_wl, jj * twstep);
        else cf_ipass_item<R>(px, pitch, s_wl, jj * twstep, k == g.bnpass - 1 ? s_bh + pos0 : nullptr, span);
    }
}
__device__ __forceinline__ void cf_blue_dispatch(const ChanFftGeom &g, int k, bool inverse, float2 *s_x, float2 *s_ws, const float2 *s_tw, const float2 *s_wl, const float2 *s_bh,
                                                 const float2 *s_c, int tid, int nthr) {
    switch (g.bradix[k]) {
        case 2: cf_blue_subpass<2>(g, k, inverse, s_x, s_ws, s_tw, s_wl, s_bh, s_c, tid, nthr); break;
        case 4: cf_blue_subpass<4>(g, k, inverse, s_x, s_ws, s_tw, s_wl, s_bh, s_c, tid, nthr); break;
        case 8: cf_blue_subpass<8>(g, k, inverse, s_x, s_ws, s_tw, s_wl, s_bh, s_c, tid, nthr); break;
        default: cf_blue_subpass<16>(g, k, inverse, s_x, s_ws, s_tw, s_wl, s_bh, s_c, tid, nthr); break;
    }
}
// pass 0 of a plan with a prime factor bp: y[r] = W_M^(j r) sum_q x[q] W_bp^(q r) for every column j < s0 = M / bp and frame, rows j + s0 q -> j + s0 r,
// as the chirp-z convolution  y[r] = W_M^(j r) c[r] sum_q (x[q] c[q]) conj(c[r - q]),  c[n] = exp(-j pi n^2 / bp):
//   forward bL-point sub-passes (the first reads the tile's rows times the chirp, zero padded) | times the transformed chirp filter, inverse sub-passes
//   on conjugated data (the last writes the tile's rows: times the chirp and W_M^(j r))
__device__ __forceinline__ void cf_blue_pass(const ChanFftGeom &g, float2 *s_x, float2 *s_ws, const float2 *s_tw, const float2 *s_wl, const float2 *s_bh, const float2 *s_c,
                                             int tid, int nthr) {
    for (int k = 0; k < g.bnpass; ++k) { cf_blue_dispatch(g, k, false, s_x, s_ws, s_tw, s_wl, s_bh, s_c, tid, nthr); lds_barrier(); }
    for (int k = g.bnpass - 1; k >= 0; --k) { cf_blue_dispatch(g, k, true, s_x, s_ws, s_tw, s_wl, s_bh, s_c, tid, nthr); if (k) lds_barrier(); }
}

// pass 0 of a plan with a prime factor dp = 29 .. 89, rows j + s0 q of s_x -> rows j + s0 r of s_y:  y[r] = W_M^(j r) sum_q x[q] W_dp^(q r).
// Step A forms s_c = x_c + x_{dp-c} and d_c = x_c - x_{dp-c} in place (rows c and dp - c of every column); step B is chan_analyze_p2's transform phase:
// a lane owns one (column j, frame t), a wave four output-pair slots; P_k = x_0 + sum_c s_c cos, Q_k = sum_c d_c sin with the (cos, sin) rows wave-uniform
// (scalar loads); y_k = P_k - j Q_k, y_{dp-k} = P_k + j Q_k, times the pass's twiddle, into the second tile (another wave still reads the first).
template <int KP>
__device__ __forceinline__ void cf_prime_pass(const ChanFftGeom &g, float2 *s_x, float2 *s_y, const float2 *s_tw, const float2 *__restrict__ cs, int tid, int nthr) {
    const int TF = g.TF, TFs = g.TFs, p = g.dp, s0 = g.M / p, H = (p - 1) >> 1;
    for (int it = tid; it < ((s0 * H) << g.lgTF); it += nthr) {
        const int t = it & (TF - 1);
        const unsigned rest = (unsigned)it >> g.lgTF, c1 = s0 == 1 ? rest : __umulhi(rest, g.magic_s0), j = rest - c1 * (unsigned)s0;
        float2 *pa = s_x + (size_t)(j + s0 * (c1 + 1)) * TFs + t, *pb = s_x + (size_t)(j + s0 * (p - 1 - c1)) * TFs + t;
        const float2 a = *pa, b = *pb;
        *pa = make_float2(a.x + b.x, a.y + b.y); *pb = make_float2(a.x - b.x, a.y - b.y);
    }
    lds_barrier();
    const int lane = tid & 63, wave = wave_uniform(tid >> 6), nw = nthr >> 6;
    const int n_li = s0 << g.lgTF, lgroups = (n_li + 63) >> 6, tasks = lgroups * g.dnk;
    for (int task = wave; task < tasks; task += nw) {
        const int sg = task / lgroups, lg = task - sg * lgroups;          // (wave-uniform)
        const int li = min(lg * 64 + lane, n_li - 1);
        const bool live = lg * 64 + lane < n_li;
        const int t = li & (TF - 1), j = li >> g.lgTF;
        const float2 *col = s_x + (size_t)j * TFs + t;
        const int rs = s0 * TFs;                                           // row c of this column: col + c rs
        const float2 x0 = col[0];
        float2 P[KP], Q[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) { P[k] = x0; Q[k] = make_float2(0.f, 0.f); }
        const float2 *w = cs + sg * KP;
        float2 sa = col[rs], da = col[(size_t)(p - 1) * rs];
        for (int c = 1; c <= H; ++c) {
            const int cn = min(c + 1, H);
            const float2 sn = col[(size_t)cn * rs], dn = col[(size_t)(p - cn) * rs];      // the next term's rows, requested before this term's arithmetic
            float2 e[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) e[k] = w[k];
            w += g.dPA;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                P[k].x = fmaf(sa.x, e[k].x, P[k].x); P[k].y = fmaf(sa.y, e[k].x, P[k].y);
                Q[k].x = fmaf(da.x, e[k].y, Q[k].x); Q[k].y = fmaf(da.y, e[k].y, Q[k].y);
            }
            sa = sn; da = dn;
        }
        float2 *ycol = s_y + (size_t)j * TFs + t;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int q = sg * KP + k;                                         // (wave-uniform)
            if (q < H) {
                const int kk = q + 1, kn = p - kk;
                if (live) {
                    ycol[(size_t)kk * rs] = cmul(make_float2(P[k].x + Q[k].y, P[k].y - Q[k].x), s_tw[j * kk]);      // j r < M
                    ycol[(size_t)kn * rs] = cmul(make_float2(P[k].x - Q[k].y, P[k].y + Q[k].x), s_tw[j * kn]);
                }
            } else if (q == H) { if (live) ycol[0] = P[k]; }
        }
    }
}

// The direct prime pass on the fp32 matrix pipe.  P = Cos s and Q = Sin d are real matrix products per component (re / im), tiled 16 (k) x 16 (columns) x 4
// (terms) on v_mfma_f32_16x16x4_f32: a wave takes (row tile of sixteen outputs, column tile of sixteen (column j, frame t) lanes); lane (q = lane >> 4,
// i = lane & 15) feeds term n = 4 J + q of its column in step J -- s_n = x_n + x_{p-n} and d_n formed from the two rows as they are read, so step A and
// its barrier are gone -- and receives outputs k = 16 rt + 4 q + r of that column.  An MFMA is a k-ordered fmaf chain: with the terms in ascending
// order the sums are the vector form's, bit for bit.  The coefficient fragments of a wave's row tile (2 x steps registers) are fetched once per tile.
__device__ __forceinline__ void cf_prime_pass_mx(const ChanFftGeom &g, float2 *s_x, float2 *s_y, const float2 *s_tw, const float *__restrict__ tab, int tid, int nthr) {
    const int TF = g.TF, TFs = g.TFs, p = g.dp, s0 = g.M / p, H = (p - 1) >> 1;
    const int RT = chanfft_mx_row_tiles(p), KS = chanfft_mx_steps(p);
    int lane = tid & 63;
    opaque(lane);                                                          // (what is derived from it -- fragment and column addresses -- is formed here, per tile, instead of being carried, spilled, across the FIR phase)
    const int wave = wave_uniform(tid >> 6), nw = nthr >> 6;
    const int n_li = s0 << g.lgTF, CT = (n_li + 15) >> 4, tasks = RT * CT;
    const int q = lane >> 4, rs = s0 * TFs;
    constexpr int kCh = 5, kNch = (kCfMxMaxSteps + kCh - 1) / kCh;      // the coefficient fragments arrive in chunks of five steps, one chunk ahead of the products (all of them at once do not fit the registers)
    for (int task = wave; task < tasks; task += nw) {
        const int ct = task / RT, rt = task - ct * RT;                     // (wave-uniform) neighbouring waves share a column tile's rows
        const float *tc = tab + (size_t)rt * KS * 64 + lane, *ts = tab + ((size_t)RT + rt) * KS * 64 + lane;
        const int li = min(16 * ct + (lane & 15), n_li - 1);
        const bool live = 16 * ct + (lane & 15) < n_li;
        const int t = li & (TF - 1), j = li >> g.lgTF;
        const float2 *col = s_x + (size_t)j * TFs + t;
        csdr_f32x4 Pr = {0.f, 0.f, 0.f, 0.f}, Pi = Pr, Qr = Pr, Qi = Pr;
        float2 a = col[(size_t)q * rs], b = q ? col[(size_t)(p - q) * rs] : make_float2(0.f, 0.f);      // step 0: n = q
        float mc[2][kCh], ms[2][kCh];                                      // two register sets: chunk c + 1 is on its way while chunk c is multiplied
#pragma unroll
        for (int i = 0; i < kCh; ++i) { const int J = min(i, KS - 1); mc[0][i] = tc[J * 64]; ms[0][i] = ts[J * 64]; }
#pragma unroll
        for (int c = 0; c < kNch; ++c) {
            if (c * kCh < KS) {                                            // (wave-uniform)
                if ((c + 1) * kCh < KS) {
#pragma unroll
                    for (int i = 0; i < kCh; ++i) { const int J = min((c + 1) * kCh + i, KS - 1); mc[(c + 1) & 1][i] = tc[J * 64]; ms[(c + 1) & 1][i] = ts[J * 64]; }
                }
#pragma unroll
                for (int i = 0; i < kCh; ++i) {
                    const int J = c * kCh + i;
                    if (J < KS) {                                          // (wave-uniform)
                        float2 a2 = a, b2 = b;
                        if (J + 1 < KS) { const int n2 = 4 * (J + 1) + q; a2 = col[(size_t)n2 * rs]; b2 = col[(size_t)(p - n2) * rs]; }      // (n2 <= H + 3 < p: a row of the tile; past H the coefficients are zero)
                        const float2 sv = make_float2(a.x + b.x, a.y + b.y), dv = make_float2(a.x - b.x, a.y - b.y);
                        Pr = csdr_mfma16(mc[c & 1][i], sv.x, Pr); Pi = csdr_mfma16(mc[c & 1][i], sv.y, Pi);
                        Qr = csdr_mfma16(ms[c & 1][i], dv.x, Qr); Qi = csdr_mfma16(ms[c & 1][i], dv.y, Qi);
                        a = a2; b = b2;
                    }
                }
            }
        }
        float2 *ycol = s_y + (size_t)j * TFs + t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * rt + 4 * q + r;
            if (live && k <= H) {
                const float2 P = make_float2(Pr[r], Pi[r]), Q = make_float2(Qr[r], Qi[r]);
                if (k == 0) ycol[0] = P;
                else {
                    const int kn = p - k;
                    ycol[(size_t)k * rs] = cmul(make_float2(P.x + Q.y, P.y - Q.x), s_tw[j * k]);        // j r < M
                    ycol[(size_t)kn * rs] = cmul(make_float2(P.x - Q.y, P.y + Q.x), s_tw[j * kn]);
                }
            }
        }
    }
}

// one butterfly of the last pass (span 1): results go to their channel rows
// (OS2, the oversampled bank: s_pa holds (row << 1) | (channel is odd) -- -1 stays -1 --, s_post the post factor W_M^k / M of firpfbch2 of the channel at
//  each position; odd channels change sign in the frames of odd parity, design::channelizer2_post)
template <int R, bool OS2 = false>
__device__ __forceinline__ void cf_last_item(const float2 *px, int pitch, const int *s_pa, float2 *s_dc /* non-null: this butterfly holds channel 0 (position 0) and its tile values are wanted */,
                                             int t, bool live, float2 *__restrict__ o /* out + f0 + t */, int64_t out_stride, const float2 *s_post = nullptr) {
    float2 v[R];
    if constexpr (R >= 17) {
        // (wide odd radices: the output rows -- and the oversampled bank's post factors -- are looked up one by one at the stores: R more live
        //  registers spilled)
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = px[(size_t)q * pitch];
        CfDft<R>::run(v);
        const bool odd_frame = (t & 1) != 0;                       // tiles start on even frames
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int kr = s_pa[r];
            float2 val = v[r];
            if constexpr (OS2) {
                float2 w = s_post[r];
                if (odd_frame && (kr & 1)) w = make_float2(-w.x, -w.y);
                val = cmul(val, w);
                kr >>= 1;                                          // (-1 stays -1)
            }
            if (r == 0 && s_dc) s_dc[t] = val;
            if (kr >= 0 && live) st_stream(o + (int64_t)kr * out_stride, val);
        }
        return;
    }
    int k[R];
#pragma unroll
    for (int q = 0; q < R; ++q) { v[q] = px[(size_t)q * pitch]; k[q] = s_pa[q]; }
    CfDft<R>::run(v);
    if constexpr (OS2) {
        const bool odd_frame = (t & 1) != 0;                       // tiles start on even frames
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float2 w = s_post[r];
            if (odd_frame && (k[r] & 1)) w = make_float2(-w.x, -w.y);
            v[r] = cmul(v[r], w);
            k[r] >>= 1;                                            // (-1 stays -1)
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r == 0 && s_dc) s_dc[t] = v[0];
        if (k[r] >= 0 && live) st_stream(o + (int64_t)k[r] * out_stride, v[r]);
    }
}

// the fifteen input rows the eight frames [f0 + 8 seg, + 8) reach, column pair cp: row r of the batch is x[r M ..], rows -7 .. -1 are the
// carried history, rows past the end are zero (their frames are never stored).  `inside` (tile-uniform): every row lies in x -- fifteen
// plain loads with nothing between them (a guarded load is compiled behind a wait of its own).
__device__ __forceinline__ void cf_load_window(const float2 *__restrict__ x, const float2 *__restrict__ hist, int M, int half, int64_t n_frames,
                                               int64_t f0, bool inside, int seg, int cp, float4 (&win)[kCfSeg + kChanTaps - 1]) {
    const int64_t r0 = f0 + seg * kCfSeg - (kChanTaps - 1);
    if (inside) {
        const float4 *src = reinterpret_cast<const float4 *>(x + r0 * M) + cp;
#pragma unroll
        for (int j = 0; j < kCfSeg + kChanTaps - 1; ++j) win[j] = src[(size_t)j * half];
    } else {
#pragma unroll
        for (int j = 0; j < kCfSeg + kChanTaps - 1; ++j) {
            const int64_t r = r0 + j;
            win[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < n_frames) win[j] = reinterpret_cast<const float4 *>(r >= 0 ? x + r * M : hist + (r + kChanTaps - 1) * M)[cp];
        }
    }
}
// the same for the oversampled bank (firpfbch2): frames of parity `par` live on a lattice of rows of M samples whose row r starts at sample
// r M - (par ? 0 : M / 2); the eight frames u0 .. u0 + 7 of that lattice reach rows u0 - 7 .. u0 + 7.  Samples before the batch come from the
// carried history (H = 7.5 M of them), samples past its end are zero.
__device__ __forceinline__ void cf_load_window_os2(const float2 *__restrict__ x, const float2 *__restrict__ hist, int M, int half, int64_t n_samples,
                                                   int64_t u0, int par, bool inside, int cp, float4 (&win)[kCfSeg + kChanTaps - 1]) {
    const int64_t s0 = (u0 - (kChanTaps - 1)) * M - (par ? 0 : half) + 2 * cp;       // sample index of win[0]'s first value
    if (inside) {
        const float4 *src = reinterpret_cast<const float4 *>(x + s0);                 // (M % 4 == 0: 16-byte aligned)
#pragma unroll
        for (int j = 0; j < kCfSeg + kChanTaps - 1; ++j) win[j] = src[(size_t)j * half];
    } else {
        const int64_t H = (int64_t)kChanTaps * M - half;
#pragma unroll
        for (int j = 0; j < kCfSeg + kChanTaps - 1; ++j) {
            const int64_t si = s0 + (int64_t)j * M;
            win[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (si + 1 < n_samples && si >= -H) win[j] = *reinterpret_cast<const float4 *>(si >= 0 ? x + si : hist + (si + H));      // (a pair never straddles the boundary: si and H are even)
        }
    }
}
// eight frames of the two columns of a pair from the window; each row of X receives 64 contiguous bytes
// (STRIDE2: the eight frames are every other frame of the tile -- one lattice of the oversampled bank: 8-byte stores two frames apart)
template <bool STRIDE2 = false>
__device__ __forceinline__ void cf_fir(const float4 (&win)[kCfSeg + kChanTaps - 1], const float *__restrict__ tapsT, int M, int cp, float4 *d0, float4 *d1) {
    float2 h[kChanTaps];
#pragma unroll
    for (int n = 0; n < kChanTaps; ++n) h[n] = *reinterpret_cast<const float2 *>(tapsT + (size_t)n * M + 2 * cp);
#pragma unroll
    for (int tt = 0; tt < kCfSeg; tt += 2) {
        float2 a[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)}, b[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int n = 0; n < kChanTaps; ++n) {
                const float4 v = win[kChanTaps - 1 + tt + u - n];
                a[u].x = fmaf(h[n].x, v.x, a[u].x); a[u].y = fmaf(h[n].x, v.y, a[u].y);
                b[u].x = fmaf(h[n].y, v.z, b[u].x); b[u].y = fmaf(h[n].y, v.w, b[u].y);
            }
        }
        if (STRIDE2) {
            float2 *e0 = reinterpret_cast<float2 *>(d0), *e1 = reinterpret_cast<float2 *>(d1);
            e0[2 * tt] = a[0]; e0[2 * tt + 2] = a[1]; e1[2 * tt] = b[0]; e1[2 * tt + 2] = b[1];
        } else {
            d0[tt >> 1] = make_float4(a[0].x, a[0].y, a[1].x, a[1].y);
            d1[tt >> 1] = make_float4(b[0].x, b[0].y, b[1].x, b[1].y);
        }
    }
}

// which radices an instance of the kernel carries: PLAN 0 = every small radix (+ 17 / 19 / 23 when WIDE); 1 = {16, 8} (M = 1024 = 16 * 8 * 8: BASELINE
// config 4), 2 = {5, 8} (M = 200 = 5 * 5 * 8: config 5), 3 = {5, 4} (M = 20: config 2)
__host__ __device__ constexpr bool cf_plan_has(int plan, int R, bool wide) {
    return plan == 1 ? (R == 16 || R == 8) : plan == 2 ? (R == 5 || R == 8) : plan == 3 ? (R == 5 || R == 4) : (R < 17 || wide);      // (plan 4: every small radix + the chirp-z pass)
}
__host__ inline int cf_plan_of(const ChanFftGeom &g) {
    if (g.os2 || g.wide_odd) return 0;
    return g.M == 1024 ? 1 : g.M == 200 ? 2 : g.M == 20 ? 3 : 0;
}
// one pass over the tile: R is a compile-time constant of the call
template <int R, bool LAST, bool OS2>
__device__ __forceinline__ void cf_run_pass(const ChanFftGeom &g, int p, float2 *s_x, const float2 *s_tw, const int *s_pa, float2 *s_dc, const float2 *s_post,
                                            int tid, int nthr, int nf, float2 *__restrict__ out_f0, int64_t out_stride) {
    const int TF = g.TF, TFs = g.TFs, s = g.span[p], items = (g.M / R) << g.lgTF, pitch = s * TFs;
    for (int it = tid; it < items; it += nthr) {
        const int t = it & (TF - 1);
        const unsigned bf = (unsigned)it >> g.lgTF;
        const unsigned b = cf_div(bf, s, g.magic_span[p]);
        const int j = (int)(bf - b * (unsigned)s), pos0 = (int)b * R * s + j;
        float2 *px = s_x + (size_t)pos0 * TFs + t;
        if constexpr (!LAST) cf_pass_item<R>(px, pitch, s_tw, j * g.twstep[p]);
        else cf_last_item<R, OS2>(px, pitch, s_pa + pos0, (s_dc && pos0 == 0) ? s_dc : nullptr /* position 0 is channel 0 in every digit order */, t, t < nf,
                                  out_f0 + t, out_stride, s_post + pos0);
    }
}

template <bool WIDE, bool OS2 = false, int PLAN = 0>
CSDR_KERNEL __launch_bounds__(kCfMaxThreads) void chan_analyze_fft(
    const float2 *__restrict__ x,        // batch input, n_frames * M samples
    const float2 *__restrict__ hist,     // 7 * M samples preceding x
    float2 *__restrict__ hist_new,       // receives the last 7 * M samples of (hist ++ x)
    const float *__restrict__ tapsT,     // [8][M]  tapsT[n M + c] multiplies x[(t - n) M + c]
    const float2 *__restrict__ twM,      // [M] exp(-j 2 pi i / M)
    const int *__restrict__ perm,        // [M] position after the last pass -> channel
    const int *__restrict__ active,      // [M] output row of channel k + 1; 0: not stored
    ChanFftGeom g, int64_t n_frames,
    float2 *__restrict__ out, int64_t out_stride,
    d2 *__restrict__ dc_ends, double dc_c /* DC blocker of channel 0: end value of each tile's recurrence (zero entering state) */,
    const float2 *__restrict__ post /* OS2 (firpfbch2): [M] output factors W_M^k / M of the even frames; odd channels change sign in odd frames */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int M = g.M, TF = g.TF, TFs = g.TFs;
    float2 *s_x = reinterpret_cast<float2 *>(smem);                  // X[c][t], pitch TFs
    float2 *s_tw = s_x + (size_t)M * TFs;                            // W_M^i
    float2 *s_dc = s_tw + M;                                         // channel 0 of the tile
    float2 *s_post = s_dc + TF;                                      // OS2: post factor of the channel at each position
    // chirp-z plan (PLAN 4): the work array and its three tables sit between the post factors' place and the row list; `post` carries the tables
    float2 *s_ws = s_post + (OS2 ? M : 0);
    const size_t blue_ws = (PLAN == 4 && g.bp) ? (size_t)(M / g.bp) * g.bL * TFs : (PLAN == 5 && g.dp) ? (size_t)M * TFs : 0;      // (PLAN 5, direct prime pass: the second tile)
    float2 *s_wl = s_ws + blue_ws, *s_bh = s_wl + ((PLAN == 4) ? g.bL : 0), *s_c = s_bh + ((PLAN == 4) ? g.bL : 0);
    int *s_pa = reinterpret_cast<int *>(s_c + ((PLAN == 4) ? g.bp : 0));     // position -> output row of its channel (the channel itself unless the rows are packed), or -1 when it has no consumer
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int i = tid; i < M; i += nthr) {
        const int ch = perm[i];
        const int row = active[ch] - 1;                              // active[k] = output row of channel k + 1, 0 = not stored
        s_tw[i] = twM[i];
        s_pa[i] = !OS2 ? row : (row < 0 ? -1 : ((row << 1) | (ch & 1)));      // (oversampled: the channel's parity rides along, cf_last_item)
        if (OS2) s_post[i] = post[ch];
    }
    if constexpr (PLAN == 4) for (int i = tid; i < 2 * g.bL + g.bp; i += nthr) s_wl[i] = post[i];

    const int64_t ntiles = (n_frames + TF - 1) >> g.lgTF;
    const int half = M >> 1, nfir = half * (TF / kCfSeg);
    // tile walk: workgroup b takes tiles b, b + grid, ...; with `xcd` the workgroups that share an L2 (b % 8, the dispatcher's round robin: a
    // speed assumption only) take consecutive tiles of every round, so that the window rows two neighbouring tiles share and the pieces of a
    // channel row they write meet in ONE L2
    int64_t tile0 = blockIdx.x;
    if (g.xcd && !(gridDim.x & 7)) tile0 = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    for (int64_t tile = tile0; tile < ntiles; tile += gridDim.x) {
        const int64_t f0 = tile << g.lgTF;
        const int nf = (int)min((int64_t)TF, n_frames - f0);
        // wave priorities (round 6, profiles/r06_chanfft_priority.txt): the FIR phase -- global loads straight into the registers the taps read -- issues at
        // priority 2, the passes at 1, whatever is left of a tile (set-up, the waits at the barriers) at 0: of the waves that share a SIMD the ones that
        // can put loads in flight go first.  M = 200: 0.225 -> 0.204 ms (C5), M = 40 / 112: - 6 %, M = 20: - 2 %, M = 1024 (one workgroup per CU): - 1 %
        wave_priority(kCfPrioFir);
        // ---- FIR: item = (column pair cp, segment of 8 frames)
        if constexpr (!OS2) {
            const bool inside = f0 >= kChanTaps - 1 && f0 + TF <= n_frames;      // (tile-uniform) every row any item of the tile reaches lies in x
            for (int it = tid; it < nfir; it += nthr) {
                const int seg = (int)cf_div((unsigned)it, half, g.magic_half), cp = it - seg * half;
                float4 win[kCfSeg + kChanTaps - 1];
                cf_load_window(x, hist, M, half, n_frames, f0, inside, seg, cp, win);
                cf_fir(win, tapsT, M, cp, reinterpret_cast<float4 *>(s_x + (size_t)(2 * cp) * TFs + seg * kCfSeg),
                       reinterpret_cast<float4 *>(s_x + (size_t)(2 * cp + 1) * TFs + seg * kCfSeg));
            }
        } else {
            // oversampled: frame f0 + t = frame u = (f0 + t) >> 1 of the lattice of its parity; a segment = eight frames of ONE lattice: tile positions
            // t = 16 useg + par, + 2, ... (segment index = 2 useg + par)
            const int64_t n_samples = n_frames * half;
            const bool inside = (f0 >> 1) >= kChanTaps && f0 + TF <= n_frames;   // (tile-uniform) row u - 7 of the lattice that starts M / 2 early lies in x
            for (int it = tid; it < nfir; it += nthr) {
                const int seg = (int)cf_div((unsigned)it, half, g.magic_half), cp = it - seg * half;
                const int par = seg & 1, useg = seg >> 1;
                float4 win[kCfSeg + kChanTaps - 1];
                cf_load_window_os2(x, hist, M, half, n_samples, (f0 >> 1) + (int64_t)useg * kCfSeg, par, inside, cp, win);
                cf_fir<true>(win, tapsT, M, cp, reinterpret_cast<float4 *>(s_x + (size_t)(2 * cp) * TFs + 2 * useg * kCfSeg + par),
                             reinterpret_cast<float4 *>(s_x + (size_t)(2 * cp + 1) * TFs + 2 * useg * kCfSeg + par));
            }
        }
        // the workgroup that owns the last tile also writes the new input history (the launch runs even with no consumers)
        if (tile == ntiles - 1) {
            const int64_t n = n_frames * (OS2 ? half : M), H = (int64_t)kChanTaps * M - (OS2 ? half : M);
            for (int64_t j = tid; j < H; j += nthr) {
                const int64_t gsrc = n - H + j;
                hist_new[j] = gsrc >= 0 ? x[gsrc] : hist[gsrc + H];
            }
        }
        lds_barrier();
        wave_priority(kCfPrioPass);

        // ---- FFT passes: item = (butterfly bf, frame t), lanes along t.  The radix is a property of the PASS: one dispatch per pass, the item loop
        // inside it has a compile-time radix (and an instance built for a plan -- PLAN != 0 -- carries only its own radices: registers sized by the plan)
        float2 *s_t = s_x;                                          // the tile the passes work on (a direct prime pass moves it to the second array)
        for (int p = 0; p < g.npass; ++p) {
            const bool lastp = p == g.npass - 1;
            float2 *dcs = dc_ends ? s_dc : nullptr;
            if constexpr (PLAN == 4) if (p == 0) { cf_blue_pass(g, s_x, s_ws, s_tw, s_wl, s_bh, s_c, tid, nthr); lds_barrier(); continue; }
            if constexpr (PLAN == 5) if (p == 0) {
                if (kCfPrimeMx) cf_prime_pass_mx(g, s_x, s_ws, s_tw, reinterpret_cast<const float *>(post), tid, nthr);
                else cf_prime_pass<kCfDirectKP>(g, s_x, s_ws, s_tw, post, tid, nthr);
                lds_barrier(); s_t = s_ws; continue;
            }
#define CSDR_CF_CASE(R_)                                                                                                                         \
            case R_:                                                                                                                             \
                if constexpr (cf_plan_has(PLAN, R_, WIDE)) {                                                                                     \
                    if (lastp) cf_run_pass<R_, true, OS2>(g, p, s_t, s_tw, s_pa, dcs, s_post, tid, nthr, nf, out + f0, out_stride);               \
                    else cf_run_pass<R_, false, OS2>(g, p, s_t, s_tw, s_pa, dcs, s_post, tid, nthr, nf, out + f0, out_stride);                    \
                }                                                                                                                                \
                break;
            switch (g.radix[p]) {
                CSDR_CF_CASE(2) CSDR_CF_CASE(3) CSDR_CF_CASE(4) CSDR_CF_CASE(5) CSDR_CF_CASE(7) CSDR_CF_CASE(8) CSDR_CF_CASE(9) CSDR_CF_CASE(11) CSDR_CF_CASE(13)
                CSDR_CF_CASE(16) CSDR_CF_CASE(17) CSDR_CF_CASE(19) CSDR_CF_CASE(23)
                default: break;
            }
#undef CSDR_CF_CASE
            lds_barrier();
        }
        wave_priority(0);
        if (dc_ends) {
            // v_end = sum_t c^(nf-1-t) y0[t]: the DC blocker's state after this tile if it entered with zero (iirfilt, :375)
            if (tid < 64) {
                double vx = 0.0, vy = 0.0;
                for (int t = tid; t < nf; t += 64) {
                    const double wgt = dc_pow(dc_c, nf - 1 - t);
                    const float2 v = s_dc[t];
                    vx += wgt * (double)v.x; vy += wgt * (double)v.y;
                }
                for (int o = 32; o > 0; o >>= 1) { vx += __shfl_down(vx, o, 64); vy += __shfl_down(vy, o, 64); }
                if (tid == 0) dc_ends[tile] = d2{vx, vy};
            }
        }
    }
}

}  // namespace csdr
