// kernels_demod.hpp -- DemodulatorPreThread + DemodulatorThread + Modem arithmetic, batched over demodulators.
//
// Replaces (reference file:line): nco_crcf_mix_block_{up,down} DemodulatorPreThread.cpp:186-195, msresamp_crcf_execute
// :209 (K5, K6); freqdem_demodulate_block ModemNBFM.cpp:36 / ModemFM.cpp:36 (K7); ModemAM.cpp:41-47 (K8);
// ModemUSB.cpp:54-61 / ModemLSB.cpp (K9); ModemAnalog::buildAudioOutput ModemAnalog.cpp:67-93 (K10);
// level / peak loops DemodulatorThread.cpp:142-152,223-233 (K11, K12).
//
// Design: every filter on this path is FIR except the SSB Butterworth (whose impulse response is below fp32
// resolution after ~100 samples).  So instead of carrying liquid's per-stage windows as state, each workgroup
// re-derives the windows by running the cascade over a short warm-up span that precedes its range: the only carried
// state is (i) the integer bookkeeping (NCO phase word, half-band input fill, 24-bit resampler phase), tracked in
// closed form, and (ii) the tail of each sample stream.  Ranges are therefore independent and run concurrently;
// outputs are bit-for-bit independent of the batching.
//
// All LDS is dynamic (`smem`).
#pragma once
#include "common.hpp"
#include "cpx.hpp"

namespace csdr {

constexpr int kMaxHb = 10;        // half-band stages supported per resampler
constexpr int kHbMaxM = 10;       // largest half-band m (taps: 2m on the filtered branch)
constexpr int kArmTaps = 14;      // arbitrary resampler: 2 * 7 taps per arm
constexpr int kArms = 256;
constexpr int kMixHist = 65536;   // upper bound of the mixed-input history kept per slot (cascade span; S <= 10)
constexpr int kIqHist = 1024;     // resampled-IQ history kept per slot (modem warm-ups; the discriminator in front of a decimating audio cascade)
constexpr int kDHist = 1024;      // scaled demodulator-output history kept per slot (>= span of a decimating audio cascade: 302 samples at 400 kHz -> 48 kHz)
constexpr int kScopeMax = 2048;   // DEMOD_VIS_SIZE (DemodulatorThread.h:15)
constexpr int kFeThreads = 256;
constexpr int kFeChunk = 2048;    // input samples one inner iteration of the front-end stages through LDS
constexpr int kFePairs = kFeChunk / 2 / kFeThreads;   // 16-byte loads per thread per chunk
constexpr int kFeTail = 24;       // per-stage carried tail (>= 2 * kHbMaxM)
constexpr int kFeZTail = 16;      // carried tail of the half-band chain output (>= 13)

struct ResampCfg {                // one msresamp (complex decimator or real interpolator/decimator)
    int32_t interp;               // 1: arbitrary stage first then x2 stages; 0: /2 stages first then arbitrary
    int32_t S;
    int32_t m_x[kMaxHb];          // execution order
    float h_x[kMaxHb][kHbMaxM];   // execution order; first half of the symmetric filtered-branch taps
    uint32_t step;
    int32_t arms_idx;             // which [256][14] bank
};

struct SlotCfg {                  // static per configuration, lives in HBM
    ResampCfg rs_iq;              // msresamp_crcf  (DemodulatorWorkerThread.cpp:100)
    ResampCfg rs_au;              // msresamp_rrrf  (ModemAnalog.cpp:30)
    int32_t modem;
    int32_t hist_len;             // mixed-input history actually kept (cascade span of this slot, multiple of 64)
    float2 *mixhist;              // [2][hist_len]  ping-pong by SlotDyn::hist_parity
    float2 *iq;                   // [2][kIqHist + cap_iq]  resampled IQ: [history | batch], ping-pong by hist_parity
    float *d;                     // [cap_iq]     unscaled demodulator output of the batch
    float *dh;                    // [2][kDHist]  scaled demodulator-output history, ping-pong
    float *audio;                 // [cap_audio]
    float *agc;                   // [2][4] aOutputCeil, aOutputCeilMA, aOutputCeilMAA (ModemAnalog.h), ping-pong
    uint32_t *pll;                // [2] DSB Costas loop: oscillator phase word, frequency word (carried across batches)
    float *blockmax;              // [max_blocks]
    float *blockmaa;              // [max_blocks] aOutputCeilMAA in force for each block (demod_gain_scan)
    struct BlockOut *bout;        // [max_blocks]
    float *scope;                 // [kScopeMax] scaled demodulator output of the LAST block of the batch (ModemAnalog::getDemodOutputData: the scope tap)
    int32_t *scope_n;             // how many of them
    int32_t cap_iq, cap_audio;
    // FM stereo (ModemFMStereo.cpp), allocated for CSDR_MODEM_FMS slots only (kernels_fms.hpp)
    float2 *fms_x;                // [cap_iq]  r2c Hilbert output of the batch
    uint32_t *fms_theta;          // [cap_iq]  pilot oscillator phase word after each sample's step
    float *fms_m, *fms_s;         // [cap_audio / 2]  mono / stereo-difference audio of the batch
    float2 *fms_yh;               // [2][kFmsYHist]  down-mixed samples in front of the batch (c2r window), ping-pong
    float *fms_uh;                // [2][2][kFmsFirMax]  matrix outputs in front of the batch (left | right), ping-pong
    float *fms_state;             // [kFmsStateWords]  pilot filter state + oscillator phase / frequency words (carried in place)
    float *fms_fir;               // [fms_fir_len]  de-emphasis * 16 kHz low-pass of one output channel
    int32_t fms_fir_len;
    float fms_b[15], fms_a[15];   // pilot band-pass sections, execution order
};

struct SlotDyn {                  // per batch
    int32_t active;
    int32_t chan;                 // data channel index in the post buffer
    uint32_t theta0, dtheta;      // NCO phase word at batch start, increment
    int32_t mixdir;               // 0: no shift, +1: mix up, -1: mix down
    uint32_t buf0;                // msresamp_crcf buffer_index at batch start
    uint32_t phase0;              // arbitrary resampler phase at batch start
    uint32_t aphase0;             // audio arbitrary resampler phase at batch start
    uint32_t abuf0;               // audio msresamp buffer_index (decimating audio path)
    uint32_t ssb_theta0;          // modem oscillator phase word at batch start: SSB fs/4 shifter (IQ rate), CW beep oscillator (audio rate)
    uint32_t cw_dtheta;           // CW beep oscillator increment per audio sample
    int32_t hist_parity;
    int32_t prev_j;               // resampled-IQ samples the previous batch produced (its tail is this batch's history)
    uint32_t tab_rot;             // demod_frontend_s: column rotation per row of the oscillator table in LDS (fes_tab_slot; the host picks it for dtheta)
};

struct BlockPlan { int32_t j0, q0; };   // first IQ output / first audio-arbitrary output of the block (batch-relative)

struct BlockOut {
    double level_accum;
    int32_t level_count;
    float audio_peak;
};

// phase word of an angle in radians, as nco_crcf_set_frequency / pll_step quantise it (liquid 1.5.0; host twin: design::nco_phase_word)
__device__ inline uint32_t nco_phase_word_dev(float theta) {
    float p = (float)((double)theta * 0.159154943091895);
    p -= truncf(p);
    if (p < 0.0f) p += 1.0f;
    return (uint32_t)(long long)(p * 4294967296.0f);
}

// --- NCO: 1024-entry table, no interpolation (liquid 1.5.0 nco_crcf, both NCO and VCO types) -----------------
__device__ inline void nco_sincos(const float *tab, uint32_t theta, float &s, float &c) {
    const uint32_t idx = (theta + (1u << 21)) >> 22;
    s = tab[idx & 1023u];
    c = tab[(idx + 256u) & 1023u];
}

// closed form of the resampler loop: first output index whose phase lands at or after input K (K may be negative):
// the smallest j with j * step >= K 2^24 - phase0.  |lim| < 2^53, so a double quotient is within one of the answer
// and two integer corrections make it exact (a 64-bit integer division costs several hundred cycles on the device).
__host__ __device__ inline int64_t resamp_first_out(int64_t K, uint32_t phase0, uint32_t step) {
    const int64_t lim = K * (int64_t)(1 << 24) - (int64_t)phase0;   // need j*step >= lim
    int64_t j = (int64_t)((double)lim / (double)step);
    while (j * (int64_t)step < lim) ++j;
    while ((j - 1) * (int64_t)step >= lim) --j;
    return j;
}

// ------------------------------------------------------------------------------------------------------------
// D1: NCO shift + half-band decimator cascade + arbitrary polyphase resampler.   grid = (part, slot)
//
// A workgroup owns a contiguous range [Ka, Kb) of half-band-chain outputs of one demodulator over the WHOLE batch
// (block boundaries only matter for the per-block output counts, which are closed-form integers) and the resampler
// outputs that fall on them.  It starts `warm` input samples earlier (the span of the whole cascade) with empty
// windows, so every window is fully populated by true samples when the first wanted output is formed.
// Per chunk of kFeChunk inputs: 16-byte coalesced loads (prefetched one chunk ahead into registers), table-NCO mix,
// even/odd-split LDS arrays per stage (unit-stride ds_read_b64), per-stage tails carried in LDS between chunks.
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int fe_off(int e) { return e * kFeTail + kFeChunk - (kFeChunk >> e); }   // stage e input region (tail first)
__host__ __device__ inline int fe_arr_len(int S) { return S * kFeTail + kFeChunk - (kFeChunk >> S); }
__host__ __device__ inline int fe_z_len(int S) { return kFeZTail + (kFeChunk >> S); }
__host__ __device__ inline size_t fe_lds_bytes(int S) {
    return (size_t)(2 * fe_arr_len(S) + fe_z_len(S)) * sizeof(float2) + 1024 * sizeof(float) + kMaxHb * kHbMaxM * sizeof(float);
}

// two adjacent stream samples (rel, rel + 1): batch samples come raw from the channel row, samples before the batch
// come (already mixed) from the slot's history; positions outside both are zero.
__device__ inline float4 fe_fetch_pair(const float2 *__restrict__ chan, const float2 *__restrict__ hist, int hist_len,
                                       int64_t rel, int64_t total, bool inside) {
    if (inside || (rel >= 0 && rel + 1 < total)) {     // `inside` is chunk-uniform: the whole chunk lies in the batch
        const f4u v = *reinterpret_cast<const f4u *>(chan + rel);
        return make_float4(v.x, v.y, v.z, v.w);
    }
    float2 a = make_float2(0.f, 0.f), b = a;
    if (rel >= 0) { if (rel < total) a = chan[rel]; }
    else if (rel >= -(int64_t)hist_len) a = hist[hist_len + rel];
    const int64_t r1 = rel + 1;
    if (r1 >= 0) { if (r1 < total) b = chan[r1]; }
    else if (r1 >= -(int64_t)hist_len) b = hist[hist_len + r1];
    return make_float4(a.x, a.y, b.x, b.y);
}

// the NPF pairs of a thread for one chunk (pair p = tid + q kFeThreads starts at rel_first + 2 q kFeThreads).  A chunk that lies wholly inside the
// batch takes NPF independent 16-byte loads in a block of its own; routed pair by pair through fe_fetch_pair, every load was compiled
// behind an s_waitcnt vmcnt(0) (its destination registers are also written on the history path): four serialised HBM round trips per chunk
template <int NPF>
__device__ __forceinline__ void fe_fetch_chunk(float4 (&pf)[NPF], const float2 *__restrict__ chan, const float2 *__restrict__ hist, int hist_len,
                                               int64_t rel_first, int64_t total, bool inside /* chunk-uniform */) {
    if (inside) {
        const f4u *src = reinterpret_cast<const f4u *>(chan + rel_first);
#pragma unroll
        for (int q = 0; q < NPF; ++q) { const f4u v = src[q * kFeThreads]; pf[q] = make_float4(v.x, v.y, v.z, v.w); }
    } else {
#pragma unroll
        for (int q = 0; q < NPF; ++q) pf[q] = fe_fetch_pair(chan, hist, hist_len, rel_first + 2 * q * kFeThreads, total, false);
    }
}

__device__ inline float2 fe_mix(float2 x, int64_t rel, const SlotDyn &dyn, const float *tab) {
    if (dyn.mixdir == 0 || rel < 0) return x;          // history samples are stored mixed
    float s, c;
    nco_sincos(tab, dyn.theta0 + (uint32_t)rel * dyn.dtheta, s, c);
    if (dyn.mixdir < 0) return make_float2(fmaf(x.x, c, x.y * s), fmaf(x.y, c, -x.x * s));   // x (c - j s)
    return make_float2(fmaf(x.x, c, -x.y * s), fmaf(x.y, c, x.x * s));                        // x (c + j s)
}

// one half-band stage through LDS:  y[k] = O[k - m] + sum_{j<m} h[j] (E[k - j] + E[k - (2m-1) + j])
// Each thread forms the two adjacent outputs k = 2t, 2t + 1 from E[2t - 2m .. 2t + 1] (m + 1 aligned 16-byte reads,
// lanes contiguous: conflict-free) and O[2t - m], O[2t - m + 1]; y[2t] goes to the next stage's even array and
// y[2t + 1] to its odd array at index t (contiguous 8-byte writes, no lane divergence).
template <int M>
__device__ inline void fe_stage_lds(const float2 *__restrict__ Ein, const float2 *__restrict__ Oin, const float *h, int cnt, float zeta,
                                    bool to_z, float2 *__restrict__ outE, float2 *__restrict__ outO, float2 *__restrict__ outZ) {
    for (int t = threadIdx.x; 2 * t < cnt; t += kFeThreads) {
        float2 ev[2 * M + 2];                               // ev[i] = E[2t - 2M + i]
        const float4 *e4 = reinterpret_cast<const float4 *>(Ein + 2 * t - 2 * M);
#pragma unroll
        for (int i = 0; i <= M; ++i) { const float4 v = e4[i]; ev[2 * i] = make_float2(v.x, v.y); ev[2 * i + 1] = make_float2(v.z, v.w); }
        float2 o0, o1;
        if (M % 2 == 0) { const float4 v = *reinterpret_cast<const float4 *>(Oin + 2 * t - M); o0 = make_float2(v.x, v.y); o1 = make_float2(v.z, v.w); }
        else { o0 = Oin[2 * t - M]; o1 = Oin[2 * t - M + 1]; }
        float a0r = o0.x, a0i = o0.y, a1r = o1.x, a1i = o1.y;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const float hj = h[j];
            a0r = fmaf(hj, ev[2 * M - j].x + ev[1 + j].x, a0r); a0i = fmaf(hj, ev[2 * M - j].y + ev[1 + j].y, a0i);
            a1r = fmaf(hj, ev[2 * M + 1 - j].x + ev[2 + j].x, a1r); a1i = fmaf(hj, ev[2 * M + 1 - j].y + ev[2 + j].y, a1i);
        }
        if (to_z) {
            outZ[2 * t] = make_float2(a0r * zeta, a0i * zeta);
            if (2 * t + 1 < cnt) outZ[2 * t + 1] = make_float2(a1r * zeta, a1i * zeta);
        } else {
            outE[t] = make_float2(a0r, a0i);
            outO[t] = make_float2(a1r, a1i);             // cnt is even whenever another stage follows
        }
    }
}
// one output per thread with a compile-time m: fe_stage_any's arithmetic (same order), unrolled
template <int M>
__device__ inline void fe_stage_one(const float2 *__restrict__ Ein, const float2 *__restrict__ Oin, const float *h, int cnt, float zeta,
                                    bool to_z, float2 *__restrict__ outE, float2 *__restrict__ outO, float2 *__restrict__ outZ) {
    for (int k = threadIdx.x; k < cnt; k += kFeThreads) {
        const float2 d = Oin[k - M];
        float ar = d.x, ai = d.y;
#pragma unroll 5
        for (int j = 0; j < M; ++j) {
            const float hj = h[j];
            const float2 p = Ein[k - j], q = Ein[k - (2 * M - 1) + j];
            ar = fmaf(hj, p.x + q.x, ar); ai = fmaf(hj, p.y + q.y, ai);
        }
        if (to_z) outZ[k] = make_float2(ar * zeta, ai * zeta);
        else if (k & 1) outO[k >> 1] = make_float2(ar, ai);
        else outE[k >> 1] = make_float2(ar, ai);
    }
}
__device__ inline void fe_stage_any(int m, const float2 *__restrict__ Ein, const float2 *__restrict__ Oin, const float *h, int cnt, float zeta,
                                    bool to_z, float2 *__restrict__ outE, float2 *__restrict__ outO, float2 *__restrict__ outZ) {
    for (int k = threadIdx.x; k < cnt; k += kFeThreads) {
        const float2 d = Oin[k - m];
        float ar = d.x, ai = d.y;
        for (int j = 0; j < m; ++j) {
            const float hj = h[j];
            const float2 p = Ein[k - j], q = Ein[k - (2 * m - 1) + j];
            ar = fmaf(hj, p.x + q.x, ar); ai = fmaf(hj, p.y + q.y, ai);
        }
        if (to_z) outZ[k] = make_float2(ar * zeta, ai * zeta);
        else if (k & 1) outO[k >> 1] = make_float2(ar, ai);
        else outE[k >> 1] = make_float2(ar, ai);
    }
}

CSDR_KERNEL_BANK __launch_bounds__(kFeThreads, 4) void demod_frontend(
    const SlotCfg *__restrict__ cfgs, const SlotDyn *__restrict__ dyns, const int *__restrict__ slot_list,
    const float2 *__restrict__ chan_base, int64_t chan_stride, int64_t total /* batch samples per channel */,
    const float *__restrict__ arms_all, const float *__restrict__ sintab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int slot = slot_list[blockIdx.y];
    const int part = blockIdx.x, P = gridDim.x;
    const SlotCfg &cfg = cfgs[slot];
    const SlotDyn dyn = dyns[slot];
    const int tid = threadIdx.x;
    const int S = cfg.rs_iq.S;
    const uint32_t step = cfg.rs_iq.step;
    const int hist_len = cfg.hist_len;
    const float2 *__restrict__ chan = chan_base + (int64_t)dyn.chan * chan_stride;
    const float2 *__restrict__ hist = cfg.mixhist + (size_t)dyn.hist_parity * hist_len;
    const float *__restrict__ arms = arms_all + (size_t)cfg.rs_iq.arms_idx * kArms * kArmTaps;
    const size_t iq_stride = (size_t)kIqHist + cfg.cap_iq;
    float2 *__restrict__ iq_cur = cfg.iq + (size_t)dyn.hist_parity * iq_stride;
    if (part == 0) {
        // resampled-IQ history of this batch = the last kIqHist samples of (previous history ++ previous batch)
        const float2 *iq_prev = cfg.iq + (size_t)(dyn.hist_parity ^ 1) * iq_stride;
        for (int i = tid; i < kIqHist; i += kFeThreads) iq_cur[i] = iq_prev[dyn.prev_j + i];
    }

    const int alen = fe_arr_len(S);
    float2 *LE = reinterpret_cast<float2 *>(smem);
    float2 *LO = LE + alen;
    float2 *LZ = LO + alen;
    float *tab = reinterpret_cast<float *>(LZ + fe_z_len(S));
    float *hb = tab + 1024;
    unsigned mcode = 0;                                // 2 bits per stage: 0 -> m = 3, 1 -> 5, 2 -> 10, 3 -> anything else
    for (int e = 0; e < S; ++e) {
        const int me = cfg.rs_iq.m_x[e];
        mcode |= (unsigned)(me == 3 ? 0 : me == 5 ? 1 : me == 10 ? 2 : 3) << (2 * e);
    }

    for (int i = tid; i < 1024; i += kFeThreads) tab[i] = sintab[i];
    for (int i = tid; i < kMaxHb * kHbMaxM; i += kFeThreads) hb[i] = cfg.rs_iq.h_x[i / kHbMaxM][i % kHbMaxM];
    for (int i = tid; i < S * kFeTail; i += kFeThreads) {
        const int e = i / kFeTail, k = i % kFeTail;
        LE[fe_off(e) + k] = make_float2(0.f, 0.f);
        LO[fe_off(e) + k] = make_float2(0.f, 0.f);
    }
    if (tid < kFeZTail) LZ[tid] = make_float2(0.f, 0.f);

    // --- index ranges (u-space: u = batch-relative input index + buf0; half-band output k covers u in [k 2^S, (k+1) 2^S))
    const int64_t K1 = ((int64_t)dyn.buf0 + total) >> S;
    const int64_t Ka = (K1 * part) / P, Kb = (K1 * (part + 1)) / P;
    const int64_t j0 = resamp_first_out(Ka, dyn.phase0, step), j1 = resamp_first_out(Kb, dyn.phase0, step);
    int64_t lo = Ka - (kArmTaps - 1);
    for (int e = S - 1; e >= 0; --e) lo = 2 * lo - (4 * cfg.rs_iq.m_x[e] - 2);
    const int64_t u_lo = (lo >> S) << S;           // floor to a multiple of 2^S (also for negatives)
    const int64_t u_stop = Kb << S;
    const float zeta = 1.0f / (float)(1 << S);

    // prefetch of the first chunk
    float4 pf[kFePairs];
    {
        const int64_t rel0 = u_lo - (int64_t)dyn.buf0;
        const int n = (int)min((int64_t)kFeChunk, u_stop - u_lo);
        const bool inside = rel0 >= 0 && rel0 + kFeChunk <= total;
        // a whole chunk inside the batch: kFePairs independent 16-byte loads in a block of their own (pair by pair through fe_fetch_pair every
        // load is compiled behind a wait for the one before: four serialised round trips per chunk -- the C4 front-end ran at 1.8 TB/s)
        if (inside && n == kFeChunk) fe_fetch_chunk<kFePairs>(pf, chan, hist, hist_len, rel0 + 2 * tid, total, true);
        else {
#pragma unroll
            for (int q = 0; q < kFePairs; ++q) {
                const int p = tid + q * kFeThreads;
                pf[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (2 * p < n) pf[q] = fe_fetch_pair(chan, hist, hist_len, rel0 + 2 * p, total, false);
            }
        }
    }
    __syncthreads();

    for (int64_t uc = u_lo; uc < u_stop; uc += kFeChunk) {
        const int n = (int)min((int64_t)kFeChunk, u_stop - uc);
        const int64_t rel0 = uc - (int64_t)dyn.buf0;          // batch-relative index of the chunk's first input
        // mix the prefetched chunk into the stage-0 arrays (or straight into Z when there is no half-band stage)
#pragma unroll
        for (int q = 0; q < kFePairs; ++q) {
            const int p = tid + q * kFeThreads;
            if (2 * p < n) {
                const int64_t r = rel0 + 2 * p;
                const float2 a = fe_mix(make_float2(pf[q].x, pf[q].y), r, dyn, tab);
                const float2 b = fe_mix(make_float2(pf[q].z, pf[q].w), r + 1, dyn, tab);
                if (S == 0) { LZ[kFeZTail + 2 * p] = a; if (2 * p + 1 < n) LZ[kFeZTail + 2 * p + 1] = b; }
                else { LE[kFeTail + p] = a; LO[kFeTail + p] = b; }
            }
        }
        // arbitrary-resampler outputs whose newest input falls in this chunk: fetch their filter arms NOW (before the
        // prefetch below: vector-memory waits retire in order, so the arms must not queue behind the next chunk's loads)
        const int64_t kz0 = uc >> S;
        const int cz = n >> S;
        int64_t ja = resamp_first_out(kz0, dyn.phase0, step), jb = resamp_first_out(kz0 + cz, dyn.phase0, step);
        if (ja < j0) ja = j0;
        if (jb > j1) jb = j1;
        const int64_t jmine = ja + tid;                          // first (for S >= 3: only) output of this thread in the chunk
        float2 hv[kArmTaps / 2];
        int kj = 0;
        if (jmine < jb) {
            const int64_t Pj = (int64_t)dyn.phase0 + jmine * (int64_t)step;
            kj = (int)((Pj >> 24) - kz0);
            const int arm = (int)((Pj & 0xFFFFFF) >> 16);
            const float2 *h2 = reinterpret_cast<const float2 *>(arms + arm * kArmTaps);
#pragma unroll
            for (int t = 0; t < kArmTaps / 2; ++t) hv[t] = h2[t];
        }
        // prefetch the next chunk while this one runs through the cascade
        if (uc + kFeChunk < u_stop) {
            const int64_t un = uc + kFeChunk;
            const int nn = (int)min((int64_t)kFeChunk, u_stop - un);
            const int64_t reln = un - (int64_t)dyn.buf0;
            const bool inside = reln >= 0 && reln + kFeChunk <= total;
            if (inside && nn == kFeChunk) fe_fetch_chunk<kFePairs>(pf, chan, hist, hist_len, reln + 2 * tid, total, true);
            else {
#pragma unroll
                for (int q = 0; q < kFePairs; ++q) {
                    const int p = tid + q * kFeThreads;
                    if (2 * p < nn) pf[q] = fe_fetch_pair(chan, hist, hist_len, reln + 2 * p, total, false);
                }
            }
        }
        __syncthreads();
        int cnt = n;
        for (int e = 0; e < S; ++e) {
            cnt >>= 1;
            const unsigned mc = (mcode >> (2 * e)) & 3u;
            const float2 *Ein = LE + fe_off(e) + kFeTail, *Oin = LO + fe_off(e) + kFeTail;
            const float *he = hb + e * kHbMaxM;
            const bool tz = (e == S - 1);
            float2 *oE = LE + (tz ? 0 : fe_off(e + 1) + kFeTail), *oO = LO + (tz ? 0 : fe_off(e + 1) + kFeTail), *oZ = LZ + kFeZTail;
            if (mc == 0) fe_stage_lds<3>(Ein, Oin, he, cnt, zeta, tz, oE, oO, oZ);
            else if (mc == 1) fe_stage_lds<5>(Ein, Oin, he, cnt, zeta, tz, oE, oO, oZ);
            else if (mc == 2) fe_stage_one<10>(Ein, Oin, he, cnt, zeta, tz, oE, oO, oZ);      // (the pair form with its 22-sample window spills next to the prefetched chunk: 0.19 -> 0.40 ms on C4)
            else fe_stage_any(cfg.rs_iq.m_x[e], Ein, Oin, he, cnt, zeta, tz, oE, oO, oZ);
            __syncthreads();
        }
        // arbitrary resampler on Z
        if (jmine < jb) {
            const float2 *z = LZ + kFeZTail + kj - (kArmTaps - 1);
            float ar = 0.f, ai = 0.f;
#pragma unroll
            for (int t = 0; t < kArmTaps / 2; ++t) {
                ar = fmaf(hv[t].x, z[2 * t].x, ar); ai = fmaf(hv[t].x, z[2 * t].y, ai);
                ar = fmaf(hv[t].y, z[2 * t + 1].x, ar); ai = fmaf(hv[t].y, z[2 * t + 1].y, ai);
            }
            iq_cur[kIqHist + jmine] = make_float2(ar, ai);
        }
        for (int64_t j = jmine + kFeThreads; j < jb; j += kFeThreads) {      // shallow cascades (S <= 2) only
            const int64_t Pj = (int64_t)dyn.phase0 + j * (int64_t)step;
            const int kq = (int)((Pj >> 24) - kz0);
            const int arm = (int)((Pj & 0xFFFFFF) >> 16);
            const float *h = arms + arm * kArmTaps;
            const float2 *z = LZ + kFeZTail + kq - (kArmTaps - 1);
            float ar = 0.f, ai = 0.f;
#pragma unroll
            for (int t = 0; t < kArmTaps; ++t) { ar = fmaf(h[t], z[t].x, ar); ai = fmaf(h[t], z[t].y, ai); }
            iq_cur[kIqHist + j] = make_float2(ar, ai);
        }
        if (uc + kFeChunk < u_stop) {      // another chunk follows: carry tails to the front of every buffer
            // (the tails are read behind the last stage's barrier; ONE barrier separates those reads and the resampler's reads of the Z tail
            // from the writes; the writes touch tail entries only, which nothing reads before the barrier after the next chunk's mix)
            float2 cv[2];
            const int ntail = 2 * kFeTail * S;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int idx = tid + r * kFeThreads;
                if (idx < ntail) {
                    const int e = idx / (2 * kFeTail), k = idx % (2 * kFeTail);
                    const float2 *arr = k < kFeTail ? LE : LO;
                    cv[r] = arr[fe_off(e) + (n >> (e + 1)) + (k % kFeTail)];
                }
            }
            float2 vz;
            if (tid < kFeZTail) vz = LZ[cz + tid];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int idx = tid + r * kFeThreads;
                if (idx < ntail) {
                    const int e = idx / (2 * kFeTail), k = idx % (2 * kFeTail);
                    float2 *arr = k < kFeTail ? LE : LO;
                    arr[fe_off(e) + (k % kFeTail)] = cv[r];
                }
            }
            if (tid < kFeZTail) LZ[tid] = vz;
        }
    }

    // new mixed-input history (other parity): the last hist_len samples of (old history ++ mixed batch)
    if (part == P - 1) {
        float2 *hnew = cfg.mixhist + (size_t)(dyn.hist_parity ^ 1) * hist_len;
        for (int i = tid; i < hist_len; i += kFeThreads) {
            const int64_t rel = total - hist_len + i;
            float2 v = make_float2(0.f, 0.f);
            if (rel >= 0) v = fe_mix(chan[rel], rel, dyn, tab);
            else if (rel >= -(int64_t)hist_len) v = hist[hist_len + rel];
            hnew[i] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// D1s: the same arithmetic, specialised for the half-band pattern every msresamp_crcf of the reference has with
// As = 60 dB: execution order m = 3, ..., 3, 5, 10 (the two lowest-rate stages are m = 5 and m = 10), S stages.
// Compile-time stage sizes (every chunk is full: the warm-up is lengthened to a whole number of chunks), S + 1
// barriers per chunk (the per-stage tails move to the front of their arrays while the next stage computes), the sine
// table carries a 256-entry wrap so sin/cos come from one two-address LDS read, and the carried streams (mixed-input
// history, resampled-IQ history) are written by one extra workgroup per demodulator (blockIdx.x == P).
// grid = (P + 1, slots of this S).  Bit-identical to demod_frontend.
// ------------------------------------------------------------------------------------------------------------
constexpr int kFeTabLen = 1024 + 256;
#ifndef CSDR_FE_PRIO_STAGE
#define CSDR_FE_PRIO_STAGE 1
#endif
#ifndef CSDR_FE_PRIO_TAIL
#define CSDR_FE_PRIO_TAIL 2
#endif
#ifndef CSDR_FE_SCHED3
#define CSDR_FE_SCHED3 1
#endif
constexpr bool kFeSched3 = CSDR_FE_SCHED3 != 0;      // tail-wave instances: three barrier intervals per chunk (A/B builds: -DCSDR_FE_SCHED3=0 is the four-interval schedule)
constexpr int kFePrioStage = CSDR_FE_PRIO_STAGE, kFePrioTail = CSDR_FE_PRIO_TAIL;      // (A/B builds: both 0 is the round-5 kernel)
// Where entry i of the oscillator table sits in LDS: row i >> 5 keeps its 32 words, its columns are rotated by rot * row.  The table reads of a wave are
// an arithmetic progression of the demodulator's phase increment over 32 banks: in the plain order thirty-two lanes meet 3.6 times per bank on average
// (C3), with the rotation the host picks for the increment (csdr_bank.hip: fe_table_rotation: a multiple of 4, so that the cosine -- 256 entries = eight rows on -- keeps the sine's column and the
// two come from one two-address read) 2.2 times.  Values and order of the
// arithmetic are untouched: bit-identical.
__device__ __forceinline__ uint32_t fes_tab_slot(uint32_t i, uint32_t t /* i + rot * (i >> 5) (+ a multiple of 32) */) { return (t & 31u) | (i & ~31u); }
template <int CH>
__host__ __device__ constexpr int fes_off(int e) { return e * kFeTail + CH - (CH >> e); }
template <int S, int CH>
__host__ __device__ constexpr size_t fes_lds_bytes() {
    return (size_t)(2 * (S * kFeTail + CH - (CH >> S)) + 2 + kFeZTail + (CH >> S)) * sizeof(float2) + kFeTabLen * sizeof(float) +
           (S * kHbMaxM) * sizeof(float);
}
constexpr int fes_m(int S, int e) { return e == S - 1 ? 10 : (e == S - 2 ? 5 : 3); }
// the odd-sample array of a stage with odd m starts one entry later, so that the pair O[2t - m], O[2t - m + 1] a thread
// reads is 16-byte aligned like its even-sample window
// (Where an odd-m stage is followed by an even-m one -- m = 5 then m = 10 -- the shifted region's last odd sample and the next region's tail
// entry 0 are ONE address.  Tail entry 0 is never read (the deepest reach is 19 of the 24 entries); it is WRITTEN by the tail carry, which runs
// after the stage that read the shifted region and a barrier before the stage that refills it, so the refill always lands last.  A depth-2
// instance would break that order -- its mix refills the region right behind the carry with no barrier between, safe only inside one wave:
// measured in round 4, slower than the generic kernel, not instantiated.)
template <int S, int CH>
__host__ __device__ constexpr int fes_offo(int e) { return fes_off<CH>(e) + (fes_m(S, e) & 1); }

// CNT outputs of stage `E` (compile-time), two per thread
template <int M, int CNT>
__device__ inline void fes_stage_pairs(const float2 *__restrict__ Ein, const float2 *__restrict__ Oin, const float *__restrict__ h,
                                       float2 *__restrict__ outE, float2 *__restrict__ outO, int tx /* thread index inside the group that runs the stage */) {
    constexpr int NP = CNT / 2;
#pragma unroll
    for (int t0 = 0; t0 < NP; t0 += kFeThreads) {
        const int t = t0 + tx;
        if (t0 + kFeThreads <= NP || t < NP) {                   // (compile-time for whole rounds of the workgroup)
            float2 ev[2 * M + 2];                               // ev[i] = E[2t - 2M + i]
            float4 e4[M + 1], o4;                               // both windows are 16-byte aligned (fes_offo)
            lds_read128<M + 1>(Ein + 2 * t - 2 * M, Oin + 2 * t - M, e4, o4);
#pragma unroll
            for (int i = 0; i <= M; ++i) { ev[2 * i] = make_float2(e4[i].x, e4[i].y); ev[2 * i + 1] = make_float2(e4[i].z, e4[i].w); }
            float a0r = o4.x, a0i = o4.y, a1r = o4.z, a1i = o4.w;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const float hj = h[j];
                a0r = fmaf(hj, ev[2 * M - j].x + ev[1 + j].x, a0r); a0i = fmaf(hj, ev[2 * M - j].y + ev[1 + j].y, a0i);
                a1r = fmaf(hj, ev[2 * M + 1 - j].x + ev[2 + j].x, a1r); a1i = fmaf(hj, ev[2 * M + 1 - j].y + ev[2 + j].y, a1i);
            }
            outE[t] = make_float2(a0r, a0i);
            outO[t] = make_float2(a1r, a1i);
        }
    }
}
// last stage (m = 10, CNT <= 256 outputs, one per thread) -> Z, scaled by 2^-S
template <int CNT>
__device__ inline void fes_stage_last(const float2 *__restrict__ Ein, const float2 *__restrict__ Oin, const float *__restrict__ h, float zeta,
                                      float2 *__restrict__ outZ, int tx) {
    constexpr int M = 10;
    const int k = tx;
    if (k < CNT) {
        const float2 d = Oin[k - M];
        float ar = d.x, ai = d.y;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const float hj = h[j];
            const float2 p = Ein[k - j], q = Ein[k - (2 * M - 1) + j];
            ar = fmaf(hj, p.x + q.x, ar); ai = fmaf(hj, p.y + q.y, ai);
        }
        outZ[k] = make_float2(ar * zeta, ai * zeta);
    }
}
// move the last kFeTail entries of stage region E (data count CNT >= kFeTail) to its front; threads 256-48 .. 255
template <int CNT>
__device__ inline void fes_carry_tail(float2 *__restrict__ LE, float2 *__restrict__ LO, int off, int offo) {
    const int c = (int)threadIdx.x - (kFeThreads - 2 * kFeTail);
    if (c >= 0 && c < 2 * kFeTail) {
        float2 *arr = c < kFeTail ? LE + off : LO + offo;
        const int k = c < kFeTail ? c : c - kFeTail;
        arr[k] = arr[CNT + k];
    }
}

// Stages [0, fes_blk) run on the whole workgroup, one barrier each; when the remaining stages are small enough for
// one wave (<= 64 output pairs, and <= 64 chain outputs for the resampler) they run on wave 0 alone with wave-level
// hand-offs while the other waves already start on the next chunk.
template <int S, int CH>
__host__ __device__ constexpr int fes_blk() {
    if ((CH >> S) > 64) return S;                                  // resampler needs more than one wave: no wave tail
    int e = 0;
    while (e < S && (CH >> (e + 2)) > 64) ++e;
    return e;
}
template <int S, int CH, int E, int END, bool WAVE>
struct FesStages {
    static __device__ inline void run(float2 *LE, float2 *LO, float2 *LZ, const float *hb, float zeta, int tx) {
        constexpr int CNT = CH >> (E + 1);                       // outputs of stage E
        constexpr int M = fes_m(S, E);
        const float2 *Ein = LE + fes_off<CH>(E) + kFeTail, *Oin = LO + fes_offo<S, CH>(E) + kFeTail;
        if constexpr (E == S - 1) fes_stage_last<CNT>(Ein, Oin, hb + E * kHbMaxM, zeta, LZ + kFeZTail, tx);
        else fes_stage_pairs<M, CNT>(Ein, Oin, hb + E * kHbMaxM, LE + fes_off<CH>(E + 1) + kFeTail, LO + fes_offo<S, CH>(E + 1) + kFeTail, tx);
        if constexpr (!WAVE) {
            // the tail of the PREVIOUS stage's input region is free to move now (its consumer finished at the last barrier)
            if constexpr (E >= 1) fes_carry_tail<(CH >> E)>(LE, LO, fes_off<CH>(E - 1), fes_offo<S, CH>(E - 1));
            __syncthreads();
        } else {
            // one wave alone: its own stage input is free once every lane has read it
            wave_sync();
            const int c = tx;
            if (c < 2 * kFeTail) {
                float2 *arr = c < kFeTail ? LE + fes_off<CH>(E) : LO + fes_offo<S, CH>(E);
                const int k = c < kFeTail ? c : c - kFeTail;
                arr[k] = arr[(CH >> (E + 1)) + k];
            }
        }
        FesStages<S, CH, E + 1, END, WAVE>::run(LE, LO, LZ, hb, zeta, tx);
    }
};
template <int S, int CH, int END, bool WAVE>
struct FesStages<S, CH, END, END, WAVE> {
    static __device__ inline void run(float2 *, float2 *, float2 *, const float *, float, int) {}
};

// The mix of one prefetched chunk into the stage-0 arrays: x (c + j sgn s) with (s, c) from the 1024-entry table at the oscillator's phase word
// (nco_crcf_mix_down / _up, DemodulatorPreThread.cpp:186-199); pair p = tid + q kFeThreads holds the samples rel0 + 2 p, + 1.  Samples before the
// batch (rel < 0) come from the carried history and are mixed already.  A chunk that lies wholly in the batch (rel0 >= 0: all but a range's first
// chunks) takes a form without per-sample guards: the 4 NPF table reads issued together, the direction of the mix chosen once per chunk (its sign
// rides on the multiply-adds); the products are formed exactly as in the guarded form: bit-identical.
// What the mix costs (round 5, C3, ablations of the merged depth-5 / 6 launch): 0.51 ms as is, 0.40 without it, 0.45 with conflict-free table addresses --
// the table gathers are arithmetic progressions of arbitrary stride over 64 banks (~ 4 bank cycles each where one would do: 16 gathers per wave and
// chunk, 11 % of the kernel), and the reference's oscillator IS that table.  The unguarded form itself changed nothing measurable (0.513 -> 0.515 ms:
// the LDS pipe's conflict cycles are the cost, not the waits between the reads).
template <int S, int CH, int NPF>
__device__ __forceinline__ void fes_mix_chunk(const float4 (&pf)[NPF], int64_t rel0, const SlotDyn &dyn, float sgn, const float *__restrict__ tab,
                                              float2 *__restrict__ LE, float2 *__restrict__ LO, int tid) {
    float2 *e0 = LE + kFeTail, *o0 = LO + fes_offo<S, CH>(0) + kFeTail;
    const uint32_t th0 = dyn.theta0 + (uint32_t)rel0 * dyn.dtheta, rot = dyn.tab_rot;
    if (dyn.mixdir != 0 && rel0 >= 0) {                          // (chunk-uniform)
        float ts[NPF][4];
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int p = tid + q * kFeThreads;
            const uint32_t tha = th0 + (uint32_t)(2 * p) * dyn.dtheta, thb = tha + dyn.dtheta;
            const uint32_t ia = (tha + (1u << 21)) >> 22, ib = (thb + (1u << 21)) >> 22;          // 0 .. 1023
            const uint32_t sa = fes_tab_slot(ia, ia + rot * (ia >> 5)), sb = fes_tab_slot(ib, ib + rot * (ib >> 5));     // cosine: eight rows on, the same column (rot is a multiple of 4)
            ts[q][0] = tab[sa]; ts[q][1] = tab[sa + 256]; ts[q][2] = tab[sb]; ts[q][3] = tab[sb + 256];
        }
        if (dyn.mixdir < 0) {
#pragma unroll
            for (int q = 0; q < NPF; ++q) {
                const int p = tid + q * kFeThreads;
                const float4 v = pf[q];
                e0[p] = make_float2(fmaf(v.x, ts[q][1], v.y * ts[q][0]), fmaf(v.y, ts[q][1], -(v.x * ts[q][0])));
                o0[p] = make_float2(fmaf(v.z, ts[q][3], v.w * ts[q][2]), fmaf(v.w, ts[q][3], -(v.z * ts[q][2])));
            }
        } else {
#pragma unroll
            for (int q = 0; q < NPF; ++q) {
                const int p = tid + q * kFeThreads;
                const float4 v = pf[q];
                e0[p] = make_float2(fmaf(v.x, ts[q][1], -(v.y * ts[q][0])), fmaf(v.y, ts[q][1], v.x * ts[q][0]));
                o0[p] = make_float2(fmaf(v.z, ts[q][3], -(v.w * ts[q][2])), fmaf(v.w, ts[q][3], v.z * ts[q][2]));
            }
        }
        return;
    }
    const bool do_mix = dyn.mixdir != 0;
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
        const int p = tid + q * kFeThreads;
        float2 a = make_float2(pf[q].x, pf[q].y), b = make_float2(pf[q].z, pf[q].w);
        if (do_mix) {
            const uint32_t tha = th0 + (uint32_t)(2 * p) * dyn.dtheta, thb = tha + dyn.dtheta;
            const uint32_t ia = (tha + (1u << 21)) >> 22, ib = (thb + (1u << 21)) >> 22;
            const uint32_t ja = fes_tab_slot(ia, ia + rot * (ia >> 5)), jb = fes_tab_slot(ib, ib + rot * (ib >> 5));
            const float sa = tab[ja] * sgn, ca = tab[ja + 256], sb = tab[jb] * sgn, cb = tab[jb + 256];
            if (rel0 + 2 * p >= 0) a = make_float2(fmaf(a.x, ca, -(a.y * sa)), fmaf(a.y, ca, a.x * sa));
            if (rel0 + 2 * p + 1 >= 0) b = make_float2(fmaf(b.x, cb, -(b.y * sb)), fmaf(b.y, cb, b.x * sb));
        }
        e0[p] = a; o0[p] = b;
    }
}

// TW (S = 5, 6): a FIFTH wave owns the one-wave tail (the last two -- depth 6: three -- stages and the arbitrary resampler).  It works one chunk behind
// the four worker waves, one piece per barrier interval, so the workers never wait for the tail at the head of the next chunk.  Since round 6 a chunk is
// THREE barrier intervals (kFeSched3, the schedule written out inside fes_body: the next chunk's mix rides beside stage 2); the four-interval one it
// replaced (A/B builds: -DCSDR_FE_SCHED3=0) was:
//   workers      mix k | B1 | stage 0 | B2 | stage 1 | B3 | stage 2 (writes the tail's input of chunk k) | B4
//   tail wave    stage S-2 of chunk k-1 (+ its carry) | B1 | Z tail, stage S-1 | B2 | resampler | B3 | - | B4
// The tail's input region is read before B1 of chunk k and rewritten only after B3 of chunk k; everything else it touches
// is its own.  After the last chunk it runs once more without barriers.
// the workgroup's work: demodulator `slot`, range `part` of P (part == P: the carried streams)
template <int S, int CH, bool TW>
__device__ __forceinline__ void fes_body(
    const SlotCfg *__restrict__ cfgs, const SlotDyn *__restrict__ dyns, const int slot, const int part, const int P,
    const float2 *__restrict__ chan_base, int64_t chan_stride, int64_t total /* batch samples per channel */,
    const float *__restrict__ arms_all, const float *__restrict__ sintab) {
    static_assert(S >= 2 && (CH >> S) >= kFeTail && (CH >> S) <= kFeThreads && CH % (2 * kFeThreads) == 0 && (CH >> S) % 2 == 0, "chunk does not suit this cascade depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NPF = CH / 2 / kFeThreads;
    constexpr int CZ = CH >> S;                                  // half-band chain outputs per chunk
    constexpr int ALEN = S * kFeTail + CH - (CH >> S);

    const SlotCfg &cfg = cfgs[slot];
    const SlotDyn dyn = dyns[slot];
    const int tid = threadIdx.x;
    const uint32_t step = cfg.rs_iq.step;
    const int hist_len = cfg.hist_len;
    const float2 *__restrict__ chan = chan_base + (int64_t)dyn.chan * chan_stride;
    const float2 *__restrict__ hist = cfg.mixhist + (size_t)dyn.hist_parity * hist_len;
    const size_t iq_stride = (size_t)kIqHist + cfg.cap_iq;
    float2 *__restrict__ iq_cur = cfg.iq + (size_t)dyn.hist_parity * iq_stride;
    const float sgn = dyn.mixdir < 0 ? -1.0f : 1.0f;             // x (c + j sgn s)

    float2 *LE = reinterpret_cast<float2 *>(smem);
    float2 *LO = LE + ALEN;
    float2 *LZ = LO + ALEN + 2;
    float *tab = reinterpret_cast<float *>(LZ + kFeZTail + CZ);
    float *hb = tab + kFeTabLen;
    const uint32_t rot = part == P ? 0u : dyn.tab_rot;            // (the carried-stream workgroup reads the table through nco_sincos: plain order)
    for (int i = tid; i < kFeTabLen; i += kFeThreads) tab[fes_tab_slot((uint32_t)i, (uint32_t)i + rot * ((uint32_t)i >> 5))] = sintab[i & 1023];

    if (part == P) {
        // ---- carried streams of this demodulator (one workgroup): resampled-IQ history and mixed-input history
        {
            const float2 *iq_prev = cfg.iq + (size_t)(dyn.hist_parity ^ 1) * iq_stride;
            for (int i = tid; i < kIqHist; i += (int)blockDim.x) iq_cur[i] = iq_prev[dyn.prev_j + i];   // last kIqHist samples of (previous history ++ previous batch)
        }
        __syncthreads();
        float2 *hnew = cfg.mixhist + (size_t)(dyn.hist_parity ^ 1) * hist_len;
        for (int i = tid; i < hist_len; i += kFeThreads) {
            const int64_t rel = total - hist_len + i;
            float2 v = make_float2(0.f, 0.f);
            if (rel >= 0) v = fe_mix(chan[rel], rel, dyn, tab);
            else if (rel >= -(int64_t)hist_len) v = hist[hist_len + rel];
            hnew[i] = v;
        }
        return;
    }

    const float *__restrict__ arms = arms_all + (size_t)cfg.rs_iq.arms_idx * kArms * kArmTaps;
    for (int i = tid; i < S * kHbMaxM; i += kFeThreads) hb[i] = cfg.rs_iq.h_x[i / kHbMaxM][i % kHbMaxM];
    for (int i = tid; i < S * kFeTail; i += kFeThreads) {
        const int e = i / kFeTail, k = i % kFeTail;
        LE[fes_off<CH>(e) + k] = make_float2(0.f, 0.f);
        LO[fes_offo<S, CH>(e) + k] = make_float2(0.f, 0.f);
    }
    if (tid < kFeZTail) LZ[tid] = make_float2(0.f, 0.f);

    // --- index ranges (u-space: u = batch-relative input index + buf0; half-band output k covers u in [k 2^S, (k+1) 2^S))
    const int64_t K1 = ((int64_t)dyn.buf0 + total) >> S;
    const int64_t Ka = (K1 * part) / P, Kb = (K1 * (part + 1)) / P;
    const int64_t j0 = resamp_first_out(Ka, dyn.phase0, step), j1 = resamp_first_out(Kb, dyn.phase0, step);
    int64_t lo = Ka - (kArmTaps - 1);
#pragma unroll
    for (int e = S - 1; e >= 0; --e) lo = 2 * lo - (4 * fes_m(S, e) - 2);
    const int64_t u_stop = Kb << S;
    int64_t u_lo = (lo >> S) << S;                               // floor to a multiple of 2^S (also for negatives)
    u_lo = u_stop - ((u_stop - u_lo + CH - 1) / CH) * CH;        // ... and a whole number of chunks before u_stop
    const float zeta = 1.0f / (float)(1 << S);

    if constexpr (TW) {
        constexpr int BLK = fes_blk<S, CH>();
        constexpr int NT = S - BLK;                               // tail stages: 2 (depth 5) or 3 (depth 6)
        static_assert((NT == 2 || NT == 3) && BLK >= 1 && NT + 1 <= BLK + 1, "the tail wave schedule needs one barrier interval per tail piece");
        const bool tailw = tid >= kFeThreads;
        const int ltid = tid - kFeThreads;
        // wave priorities (round 6): the tail wave -- one chunk behind, its pieces must fit the workers' barrier intervals -- at 2, the workers' stages at 1,
        // their mix (LDS gathers, the next chunk's loads) and everything else at 0: 0.46 -> 0.42 - 0.43 ms on C3 / C3N (either alone: nothing / - 2 %)
        if (tailw) wave_priority(kFePrioTail);
        const int nch = (int)((u_stop - u_lo) / CH);
        float4 pf[NPF];
        if (!tailw) {
            const int64_t rel0 = u_lo - (int64_t)dyn.buf0;
            const bool inside = rel0 >= 0 && rel0 + CH <= total;
            fe_fetch_chunk<NPF>(pf, chan, hist, hist_len, rel0 + 2 * tid, total, inside);
        }
        __syncthreads();
        if constexpr (kFeSched3) {
            // THREE barrier intervals per chunk (round 6, third session): the next chunk's mix rides in the interval of the last block-wide stage (the stage-0
            // arrays are free once stage 0 has read them and their tail has moved), the carry of that stage's input tail in stage 0's interval:
            //   workers      | B1 | stage 0 (k) + carry E(BLK-1) | Ba | stage 1 (k) + carry E0 | Bb | stage 2 (k) + carry E1, mix (k + 1), loads (k + 2) |
            //   tail wave    | B1 | stage BLK of chunk k - 1     | Ba | next piece             | Bb | last stage -> Z, resampler                        |
            // The tail's input region is written in the last interval of chunk k - 1 and read in the first of chunk k; a closing barrier after the last chunk
            // stands for the B1 of a chunk that does not come.
            static_assert(BLK == 3, "the three-interval schedule is written for three block-wide stages");
            if (!tailw) {
                {
                    const int64_t rel0 = u_lo - (int64_t)dyn.buf0;
                    fes_mix_chunk<S, CH, NPF>(pf, rel0, dyn, sgn, tab, LE, LO, tid);
                    if (1 < nch) {
                        const int64_t reln = rel0 + CH;
                        const bool inside = reln >= 0 && reln + CH <= total;
                        fe_fetch_chunk<NPF>(pf, chan, hist, hist_len, reln + 2 * tid, total, inside);
                    }
                }
                for (int k = 0; k < nch; ++k) {
                    __syncthreads();                                                            // B1
                    wave_priority(kFePrioStage);
                    if (k > 0) fes_carry_tail<(CH >> BLK)>(LE, LO, fes_off<CH>(BLK - 1), fes_offo<S, CH>(BLK - 1));      // the tail of stage 2's input: its consumer finished before B1
                    FesStages<S, CH, 0, BLK - 1, false>::run(LE, LO, LZ, hb, zeta, tid);        // stage 0 | Ba | stage 1 + carry E0 | Bb
                    {
                        constexpr int E = BLK - 1, CNT = CH >> (E + 1), M = fes_m(S, E);
                        fes_stage_pairs<M, CNT>(LE + fes_off<CH>(E) + kFeTail, LO + fes_offo<S, CH>(E) + kFeTail, hb + E * kHbMaxM,
                                                LE + fes_off<CH>(E + 1) + kFeTail, LO + fes_offo<S, CH>(E + 1) + kFeTail, tid);
                        fes_carry_tail<(CH >> E)>(LE, LO, fes_off<CH>(E - 1), fes_offo<S, CH>(E - 1));
                    }
                    wave_priority(0);
                    if (k + 1 < nch) {
                        const int64_t rel1 = u_lo + (int64_t)(k + 1) * CH - (int64_t)dyn.buf0;
                        fes_mix_chunk<S, CH, NPF>(pf, rel1, dyn, sgn, tab, LE, LO, tid);
                        if (k + 2 < nch) {
                            const int64_t reln = rel1 + CH;
                            const bool inside = reln >= 0 && reln + CH <= total;
                            fe_fetch_chunk<NPF>(pf, chan, hist, hist_len, reln + 2 * tid, total, inside);
                        }
                    }
                }
                __syncthreads();                                                                // the tail's B1 of the chunk that does not come
            } else {
                for (int k = 0; k <= nch; ++k) {
                    const bool have = k >= 1, bar = k < nch;          // chunk k - 1 has a tail to run; the workers are at chunk k
                    const int64_t uc = u_lo + (int64_t)(k - 1) * CH;
                    const int64_t kz0 = uc >> S;
                    int64_t jmine = 0, jhi = 0;
                    int kj = 0;
                    float2 hv[kArmTaps / 2];
                    if (have) {
                        const int64_t ja = resamp_first_out(kz0, dyn.phase0, step), jb = resamp_first_out(kz0 + CZ, dyn.phase0, step);
                        const int64_t jlo = ja < j0 ? j0 : ja;
                        jhi = jb > j1 ? j1 : jb;
                        jmine = jlo + ltid;                           // CZ <= 64 chain outputs, rate < 1: at most one per lane
                        if (jmine < jhi) {
                            const int64_t Pj = (int64_t)dyn.phase0 + jmine * (int64_t)step;
                            kj = (int)((Pj >> 24) - kz0);
                            const float2 *h2 = reinterpret_cast<const float2 *>(arms + (int)((Pj & 0xFFFFFF) >> 16) * kArmTaps);
#pragma unroll
                            for (int t = 0; t < kArmTaps / 2; ++t) hv[t] = h2[t];
                        }
                    }
                    __syncthreads();                                                            // B1 (after the last chunk: the closing barrier)
                    if (have) FesStages<S, CH, BLK, BLK + 1, true>::run(LE, LO, LZ, hb, zeta, ltid);      // stage BLK of chunk k - 1 and the carry of its input
                    if (bar) __syncthreads();                                                   // Ba
                    if constexpr (NT == 3) {
                        if (have) { wave_sync(); FesStages<S, CH, BLK + 1, BLK + 2, true>::run(LE, LO, LZ, hb, zeta, ltid); }   // middle tail stage
                    } else if (have) {
                        if (k >= 2 && ltid < kFeZTail) LZ[ltid] = LZ[CZ + ltid];                // Z tail of the chunk before
                        wave_sync();
                        FesStages<S, CH, S - 1, S, true>::run(LE, LO, LZ, hb, zeta, ltid);      // stage S-1 -> Z
                    }
                    if (bar) __syncthreads();                                                   // Bb
                    if constexpr (NT == 3) {
                        if (have) {
                            if (k >= 2 && ltid < kFeZTail) LZ[ltid] = LZ[CZ + ltid];
                            wave_sync();
                            FesStages<S, CH, S - 1, S, true>::run(LE, LO, LZ, hb, zeta, ltid);
                        }
                    }
                    if (have) {
                        wave_sync();
                        if (jmine < jhi) {
                            const float2 *z = LZ + kFeZTail + kj - (kArmTaps - 1);
                            float ar = 0.f, ai = 0.f;
#pragma unroll
                            for (int t = 0; t < kArmTaps / 2; ++t) {
                                ar = fmaf(hv[t].x, z[2 * t].x, ar); ai = fmaf(hv[t].x, z[2 * t].y, ai);
                                ar = fmaf(hv[t].y, z[2 * t + 1].x, ar); ai = fmaf(hv[t].y, z[2 * t + 1].y, ai);
                            }
                            iq_cur[kIqHist + jmine] = make_float2(ar, ai);
                        }
                        wave_sync();
                    }
                }
            }
            return;
        }
        if (!tailw) {
            for (int k = 0; k < nch; ++k) {
                const int64_t uc = u_lo + (int64_t)k * CH;
                const int64_t rel0 = uc - (int64_t)dyn.buf0;
                fes_mix_chunk<S, CH, NPF>(pf, rel0, dyn, sgn, tab, LE, LO, tid);
                if (k + 1 < nch) {
                    const int64_t reln = rel0 + CH;
                    const bool inside = reln >= 0 && reln + CH <= total;
                    fe_fetch_chunk<NPF>(pf, chan, hist, hist_len, reln + 2 * tid, total, inside);
                }
                __syncthreads();                                                        // B1
                wave_priority(kFePrioStage);
                FesStages<S, CH, 0, BLK, false>::run(LE, LO, LZ, hb, zeta, tid);        // B2 .. B(BLK+1)
                wave_priority(0);
                fes_carry_tail<(CH >> BLK)>(LE, LO, fes_off<CH>(BLK - 1), fes_offo<S, CH>(BLK - 1));
            }
        } else {
            for (int k = 0; k <= nch; ++k) {
                const bool have = k >= 1, bar = k < nch;          // chunk k - 1 has a tail to run; the workers are at chunk k
                const int64_t uc = u_lo + (int64_t)(k - 1) * CH;
                const int64_t kz0 = uc >> S;
                int64_t jmine = 0, jhi = 0;
                int kj = 0;
                float2 hv[kArmTaps / 2];
                if (have) {
                    const int64_t ja = resamp_first_out(kz0, dyn.phase0, step), jb = resamp_first_out(kz0 + CZ, dyn.phase0, step);
                    const int64_t jlo = ja < j0 ? j0 : ja;
                    jhi = jb > j1 ? j1 : jb;
                    jmine = jlo + ltid;                           // CZ <= 64 chain outputs, rate < 1: at most one per lane
                    if (jmine < jhi) {
                        const int64_t Pj = (int64_t)dyn.phase0 + jmine * (int64_t)step;
                        kj = (int)((Pj >> 24) - kz0);
                        const float2 *h2 = reinterpret_cast<const float2 *>(arms + (int)((Pj & 0xFFFFFF) >> 16) * kArmTaps);
#pragma unroll
                        for (int t = 0; t < kArmTaps / 2; ++t) hv[t] = h2[t];
                    }
                    FesStages<S, CH, BLK, BLK + 1, true>::run(LE, LO, LZ, hb, zeta, ltid);      // stage S-2 and the carry of its input
                }
                if (bar) __syncthreads();                                                       // B1
                if constexpr (NT == 3) {
                    if (have) { wave_sync(); FesStages<S, CH, BLK + 1, BLK + 2, true>::run(LE, LO, LZ, hb, zeta, ltid); }   // middle tail stage
                    if (bar) __syncthreads();                                                   // B2
                }
                if (have) {
                    if (k >= 2 && ltid < kFeZTail) LZ[ltid] = LZ[CZ + ltid];                    // Z tail of the chunk before
                    wave_sync();
                    FesStages<S, CH, S - 1, S, true>::run(LE, LO, LZ, hb, zeta, ltid);          // stage S-1 -> Z
                }
                if (bar) __syncthreads();                                                       // B2 (B3 with three tail stages)
                if (have) {
                    wave_sync();
                    if (jmine < jhi) {
                        const float2 *z = LZ + kFeZTail + kj - (kArmTaps - 1);
                        float ar = 0.f, ai = 0.f;
#pragma unroll
                        for (int t = 0; t < kArmTaps / 2; ++t) {
                            ar = fmaf(hv[t].x, z[2 * t].x, ar); ai = fmaf(hv[t].x, z[2 * t].y, ai);
                            ar = fmaf(hv[t].y, z[2 * t + 1].x, ar); ai = fmaf(hv[t].y, z[2 * t + 1].y, ai);
                        }
                        iq_cur[kIqHist + jmine] = make_float2(ar, ai);
                    }
                    wave_sync();
                }
                if (bar) { for (int q = NT; q <= BLK; ++q) __syncthreads(); }                   // the workers' remaining barriers of this chunk
            }
        }
        return;
    }

    float4 pf[NPF];
    {
        const int64_t rel0 = u_lo - (int64_t)dyn.buf0;
        const bool inside = rel0 >= 0 && rel0 + CH <= total;
        fe_fetch_chunk<NPF>(pf, chan, hist, hist_len, rel0 + 2 * tid, total, inside);
    }
    int64_t ja = resamp_first_out(u_lo >> S, dyn.phase0, step);
    __syncthreads();

    for (int64_t uc = u_lo; uc < u_stop; uc += CH) {
        const int64_t rel0 = uc - (int64_t)dyn.buf0;          // batch-relative index of the chunk's first input
        // ---- mix the prefetched chunk into the stage-0 arrays
        fes_mix_chunk<S, CH, NPF>(pf, rel0, dyn, sgn, tab, LE, LO, tid);
        // ---- resampler outputs of this chunk: fetch their filter arms before the prefetch (vector-memory waits retire in order)
        const int64_t kz0 = uc >> S;
        int64_t jb = resamp_first_out(kz0 + CZ, dyn.phase0, step);
        const int64_t jlo = ja < j0 ? j0 : ja, jhi = jb > j1 ? j1 : jb;
        ja = jb;
        const int64_t jmine = jlo + tid;                         // CZ <= 256 half-band outputs, rate < 1: at most one per thread
        float2 hv[kArmTaps / 2];
        int kj = 0;
        if (jmine < jhi) {
            const int64_t Pj = (int64_t)dyn.phase0 + jmine * (int64_t)step;
            kj = (int)((Pj >> 24) - kz0);
            const int arm = (int)((Pj & 0xFFFFFF) >> 16);
            const float2 *h2 = reinterpret_cast<const float2 *>(arms + arm * kArmTaps);
#pragma unroll
            for (int t = 0; t < kArmTaps / 2; ++t) hv[t] = h2[t];
        }
        if (uc + CH < u_stop) {
            const int64_t reln = rel0 + CH;
            const bool inside = reln >= 0 && reln + CH <= total;
            fe_fetch_chunk<NPF>(pf, chan, hist, hist_len, reln + 2 * tid, total, inside);
        }
        __syncthreads();
        // the previous chunk's resampler is done (barrier above): its Z tail may move to the front now
        if (uc > u_lo && tid >= kFeThreads - kFeZTail) { const int k = tid - (kFeThreads - kFeZTail); LZ[k] = LZ[CZ + k]; }
        constexpr int BLK = fes_blk<S, CH>();
        FesStages<S, CH, 0, BLK, false>::run(LE, LO, LZ, hb, zeta, tid);
        // tail of the last block-wide stage's input region (threads 208 .. 255)
        if constexpr (BLK >= 1) fes_carry_tail<(CH >> BLK)>(LE, LO, fes_off<CH>(BLK - 1), fes_offo<S, CH>(BLK - 1));
        if (BLK < S && tid >= 64) continue;                      // waves 1..3 go on to the next chunk
        FesStages<S, CH, BLK, S, true>::run(LE, LO, LZ, hb, zeta, tid);
        // ---- the arbitrary resampler on Z
        if (jmine < jhi) {
            const float2 *z = LZ + kFeZTail + kj - (kArmTaps - 1);
            float ar = 0.f, ai = 0.f;
#pragma unroll
            for (int t = 0; t < kArmTaps / 2; ++t) {
                ar = fmaf(hv[t].x, z[2 * t].x, ar); ai = fmaf(hv[t].x, z[2 * t].y, ai);
                ar = fmaf(hv[t].y, z[2 * t + 1].x, ar); ai = fmaf(hv[t].y, z[2 * t + 1].y, ai);
            }
            iq_cur[kIqHist + jmine] = make_float2(ar, ai);
        }
    }
}

// (Five waves per SIMD: the tail-wave form must stay at or below 80 registers -- four resident workgroups of five waves need five wave slots
// on every SIMD; at 84 the fourth workgroup is lost and the kernel runs 60 % longer.  1536-sample chunks with five workgroups per CU were
// measured too: 0.97 ms against 0.68 ms, the per-chunk barrier chain does not shrink with the chunk.)
template <int S, int CH, bool TW = false>
CSDR_KERNEL __launch_bounds__(kFeThreads + (TW ? 64 : 0), TW ? 6 : 4) void demod_frontend_s(
    const SlotCfg *__restrict__ cfgs, const SlotDyn *__restrict__ dyns, const int *__restrict__ slot_list,
    const float2 *__restrict__ chan_base, int64_t chan_stride, int64_t total, const float *__restrict__ arms_all, const float *__restrict__ sintab, int nq) {
    // grid = (16, nq * (P + 1)): sixteen list positions per row, nq rows per range of the batch.  Workgroups are dispatched in linear order
    // (x fastest) round robin over the eight XCDs, each with its own L2: the positions w and w + 8 of a row -- where the host puts two
    // demodulators that share a CHANNEL (csdr_bank_execute) -- run on the same XCD at the same time over the same range of the batch, so
    // the channel row crosses the fabric once instead of twice (C3N, 2.1 demodulators per channel: 0.59 -> 0.47 ms; a speed assumption
    // only: any placement computes the same thing).  Positions without a demodulator hold -1.
    const int q = (int)blockIdx.y, part = q / nq, slot = slot_list[(q - part * nq) * (int)gridDim.x + (int)blockIdx.x];
    if (slot < 0) return;
    fes_body<S, CH, TW>(cfgs, dyns, slot, part, (int)gridDim.y / nq - 1, chan_base, chan_stride, total, arms_all, sintab);
}
// depths 5 and 6 in ONE launch (the NBFM and the AM / SSB demodulators of a ~500 kS/s channel): the two demodulators of a channel usually differ in
// depth, and only inside one launch can they run side by side on one XCD and share the channel row (demod_frontend_s above)
template <int CH>
CSDR_KERNEL __launch_bounds__(kFeThreads + 64, 6) void demod_frontend_s56(
    const SlotCfg *__restrict__ cfgs, const SlotDyn *__restrict__ dyns, const int *__restrict__ slot_list,
    const float2 *__restrict__ chan_base, int64_t chan_stride, int64_t total, const float *__restrict__ arms_all, const float *__restrict__ sintab, int nq) {
    const int q = (int)blockIdx.y, part = q / nq, slot = slot_list[(q - part * nq) * (int)gridDim.x + (int)blockIdx.x];
    if (slot < 0) return;
    if (cfgs[slot].rs_iq.S == 5) fes_body<5, CH, true>(cfgs, dyns, slot, part, (int)gridDim.y / nq - 1, chan_base, chan_stride, total, arms_all, sintab);
    else fes_body<6, CH, true>(cfgs, dyns, slot, part, (int)gridDim.y / nq - 1, chan_base, chan_stride, total, arms_all, sintab);
}
// (Depths 5 and 6 in ONE launch were measured in round 3: 0.749 ms against 0.440 + 0.231 ms for the two launches -- the depth-5 workgroups
// then carry the depth-6 LDS carve and fewer of them are resident; that kernel is gone.)

// ------------------------------------------------------------------------------------------------------------
// D1i: NCO shift + INTERPOLATING msresamp_crcf (demodulator bandwidth above the channel rate: DemodulatorWorkerThread.cpp:97-101
// creates the resampler for any ratio): arbitrary polyphase stage first (1 < rate_arb <= 2: one or two outputs per input), then
// S x2 half-band stages -- the structure of the audio interpolator, on complex samples.
// grid = (chunks + 1, slots): a workgroup owns a chunk of kFiChunk consecutive OUTPUT samples of the batch, propagates the range
// back through the stages (closed-form indices, no carried window), stages the mixed inputs it needs in LDS and runs the cascade
// through two ping-pong arrays; the extra workgroup carries the histories (resampled IQ, mixed input).
// ------------------------------------------------------------------------------------------------------------
constexpr int kFiChunk = 2048;
constexpr int kFiArr = kFiChunk + 256;        // LDS array length (outputs of a stage + half-band reach)
constexpr size_t kFiLds = (size_t)3 * kFiArr * sizeof(float2);

CSDR_KERNEL_BANK __launch_bounds__(kFeThreads) void demod_frontend_interp(
    const SlotCfg *__restrict__ cfgs, const SlotDyn *__restrict__ dyns, const int *__restrict__ slot_list,
    const float2 *__restrict__ chan_base, int64_t chan_stride, int64_t total /* batch samples per channel */,
    const float *__restrict__ arms_all, const float *__restrict__ sintab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *s_in = reinterpret_cast<float2 *>(smem), *w0 = s_in + kFiArr, *w1 = w0 + kFiArr;
    const int slot = slot_list[blockIdx.y], tid = threadIdx.x;
    const SlotCfg &cfg = cfgs[slot];
    const SlotDyn dyn = dyns[slot];
    const ResampCfg &rs = cfg.rs_iq;
    const int S = rs.S, hist_len = cfg.hist_len;
    const uint32_t step = rs.step;
    const float2 *__restrict__ chan = chan_base + (int64_t)dyn.chan * chan_stride;
    const float2 *__restrict__ hist = cfg.mixhist + (size_t)dyn.hist_parity * hist_len;
    const size_t iq_stride = (size_t)kIqHist + cfg.cap_iq;
    float2 *__restrict__ iq_cur = cfg.iq + (size_t)dyn.hist_parity * iq_stride;
    const float *__restrict__ arms = arms_all + (size_t)rs.arms_idx * kArms * kArmTaps;
    if (blockIdx.x == gridDim.x - 1) {
        // ---- carried streams: resampled-IQ history and mixed-input history (other parity)
        const float2 *iq_prev = cfg.iq + (size_t)(dyn.hist_parity ^ 1) * iq_stride;
        for (int i = tid; i < kIqHist; i += kFeThreads) iq_cur[i] = iq_prev[dyn.prev_j + i];
        float2 *hnew = cfg.mixhist + (size_t)(dyn.hist_parity ^ 1) * hist_len;
        for (int i = tid; i < hist_len; i += kFeThreads) {
            const int64_t rel = total - hist_len + i;
            float2 v = make_float2(0.f, 0.f);
            if (rel >= 0) v = fe_mix(chan[rel], rel, dyn, sintab);
            else if (rel >= -(int64_t)hist_len) v = hist[hist_len + rel];
            hnew[i] = v;
        }
        return;
    }
    const int64_t Jtot = resamp_first_out(total, dyn.phase0, step) << S;      // resampled samples of the batch
    const int64_t A0 = (int64_t)blockIdx.x * kFiChunk, A1 = min(Jtot, A0 + kFiChunk);
    if (A0 >= A1) return;
    // backward range propagation: lo[s] / hi[s] = input range of x2 stage s (s = 0: arbitrary-stage outputs)
    int64_t lo[kMaxHb + 1], hi[kMaxHb + 1];
    lo[S] = A0; hi[S] = A1;
    for (int st = S - 1; st >= 0; --st) { lo[st] = (lo[st + 1] >> 1) - (2 * rs.m_x[st] - 1); hi[st] = (hi[st + 1] + 1) >> 1; }
    const int nv = (int)(hi[0] - lo[0]);
    const int64_t jlo = (((int64_t)dyn.phase0 + lo[0] * (int64_t)step) >> 24) - (kArmTaps - 1);
    const int64_t jhi = (((int64_t)dyn.phase0 + (hi[0] - 1) * (int64_t)step) >> 24) + 1;
    const int nwin = (int)(jhi - jlo);
    for (int i = tid; i < nwin; i += kFeThreads) {
        const int64_t rel = jlo + i;                                           // batch-relative input index
        float2 v = make_float2(0.f, 0.f);
        if (rel >= 0) { if (rel < total) v = fe_mix(chan[rel], rel, dyn, sintab); }
        else if (rel >= -(int64_t)hist_len) v = hist[hist_len + rel];          // history samples are stored mixed
        s_in[i] = v;
    }
    __syncthreads();
    // arbitrary stage: v[q], q in [lo[0], hi[0])  (outputs with q < 0 belong to the previous batch: recomputed from its history)
    for (int i = tid; i < nv; i += kFeThreads) {
        const int64_t P = (int64_t)dyn.phase0 + (lo[0] + i) * (int64_t)step;
        const float *h = arms + (int)((P & 0xFFFFFF) >> 16) * kArmTaps;
        const float2 *z = s_in + ((P >> 24) - (kArmTaps - 1) - jlo);
        float ar = 0.f, ai = 0.f;
#pragma unroll
        for (int t = 0; t < kArmTaps; ++t) { ar = fmaf(h[t], z[t].x, ar); ai = fmaf(h[t], z[t].y, ai); }
        w0[i] = make_float2(ar, ai);
    }
    __syncthreads();
    // x2 stages: w'[2q] = w[q - m], w'[2q + 1] = sum_j h1[j] (w[q - j] + w[q - (2m - 1) + j])
    float2 *src = w0, *dst = w1;
    for (int st = 0; st < S; ++st) {
        const int m = rs.m_x[st];
        const int64_t olo = lo[st + 1], ilo = lo[st];
        const int nout = (int)(hi[st + 1] - olo);
        const int qoff = (int)((olo >> 1) - ilo), par0 = (int)(olo & 1);
        for (int i = tid; i < nout; i += kFeThreads) {
            const int a = i + par0, qi = qoff + (a >> 1);
            float2 v;
            if ((a & 1) == 0) v = src[qi - m];
            else {
                v = make_float2(0.f, 0.f);
                for (int j = 0; j < m; ++j) {
                    const float hj = rs.h_x[st][j];
                    const float2 p = src[qi - j], q2 = src[qi - (2 * m - 1) + j];
                    v.x = fmaf(hj, p.x + q2.x, v.x); v.y = fmaf(hj, p.y + q2.y, v.y);
                }
            }
            dst[i] = v;
        }
        __syncthreads();
        float2 *t = src; src = dst; dst = t;
    }
    for (int i = tid; i < (int)(A1 - A0); i += kFeThreads) iq_cur[kIqHist + A0 + i] = src[i];
}

// ------------------------------------------------------------------------------------------------------------
// D2a: modem core of the auto-gain modems -> unscaled demodulator output d[j] for the block and the block maximum
// (auto-gain input).   grid = (auto-gain slot, block), 256 threads
//   AM      : d = FIR51(|x|)                                                     (ModemAM.cpp:41-47)
//   USB/LSB : fs/4 shift, 3 biquads, shift back, Hilbert c2r, keep upper/lower    (ModemUSB.cpp:54-61)
// ------------------------------------------------------------------------------------------------------------
constexpr int kModemThreads = 256;
constexpr int kAudioThreads = 64;          // demod_modem / demod_audio_interp: ONE wave per (demodulator, block) -- the barriers between their stages are wave-local
constexpr int kModemMaxBlockIq = 16384;    // resampled samples of one block one workgroup may have to stage (the real bound is the LDS its kernels need: csdr_bank_execute)
constexpr int kAmTaps = 51;
constexpr int kSsbFir = 128;               // taps of the SSB low-pass run as an FIR filter (pole radius <= 0.77: 0.77^128 ~ 3e-15)
constexpr int kSsbWarm = kSsbFir - 1;      // samples the filter reaches back
constexpr int kHilbM = 5;                  // firhilbf_create(5, As): 21-tap half-band, 10 odd taps
constexpr int kCwIqWin = 512;              // resampled-IQ samples one CW block can reach (interpolation by >= 2: far fewer)
// dynamic LDS: two float streams of `cap_stream` samples (max block + warm-up, multiple of 4) + 64 bytes of reduction scratch

struct ModemConsts {
    float am_taps[kAmTaps];                // h[i] multiplies |x|[j - i]
    float sos_b[3][3], sos_a[3][3];        // Butterworth sections, execution order
    float hilb[2 * kHilbM];                // hq[(n-1)/2] for odd delay n   (firhilbf_create(5, 90), ModemUSB.cpp:11)
    float hilb60[2 * kHilbM];              // the same for firhilbf_create(5, 60)          (ModemCW.cpp:23)
    float ssb_fir[kSsbFir];                // impulse response of the three sections (iirfilt_crcf_create_lowpass(6, 0.25), ModemUSB.cpp:8)
};

// sum / maximum over the ONE wave of a modem / audio workgroup (kAudioThreads): valid in thread 0
__device__ inline double wave_sum_double(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ inline float wave_max_float(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}

// The per-batch tables of a bank (slot dynamics, launch lists, block plans) from their page-locked staging slot into device memory, fetched BY A KERNEL of
// the front-end's own stream.  A hipMemcpyAsync puts a copy-engine transfer into the dependent chain of a call: 6 us of transfer behind ~ 20 us of host
// time inside the API call, and ~ 12 us until the compute queue sees the engine's completion signal (rocprofv3 traces of one-block calls, round 5) -- a
// third of a one-block call.  The staging slot is host memory mapped into the device's address space (hipHostMalloc): n16 16-byte words.
CSDR_KERNEL_BANK __launch_bounds__(256) void bank_tables_fetch(const float4 *__restrict__ staged, float4 *__restrict__ tables, int n16) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) tables[i] = staged[i];
}

// A ONE-block batch (the real-time shape): the auto-gain recurrence over the blocks is a single step, taken by the modem workgroup itself instead of a
// demod_gain_scan launch behind it (one dependent kernel less in the chain of a call).  The statements of demod_gain_scan, below.
__device__ inline void gain_step_single(const SlotCfg &cfg, const SlotDyn &dyn, const BlockPlan *pl, float block_max) {
    const float *agc_in = cfg.agc + 4 * dyn.hist_parity;
    float ceil_ = agc_in[0], ma = agc_in[1], maa = agc_in[2];
    if (pl[1].j0 != pl[0].j0) {
        ma = ma + (ceil_ - ma) * 0.025f;
        maa = maa + (ma - maa) * 0.025f;
        ceil_ = block_max;
    }
    cfg.blockmaa[0] = maa;
    float *agc_out = cfg.agc + 4 * (dyn.hist_parity ^ 1);
    agc_out[0] = ceil_; agc_out[1] = ma; agc_out[2] = maa;
}

// (the work of one workgroup = one wave, as a function: the kernel below calls it, and so does the fused modem + audio kernel of a one-block batch)
__device__ __forceinline__ void demod_modem_body(
    const SlotCfg *__restrict__ cfgs, const SlotDyn *__restrict__ dyns, const int slot, const int b,
    const BlockPlan *__restrict__ plans, int NB, int cap_stream, const ModemConsts *__restrict__ mc, const float *__restrict__ sintab,
    const float *__restrict__ arms_all, int cap_cw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_a = reinterpret_cast<float *>(smem);              // AM: |x| ; SSB: real part stream (the imaginary part and the filtered pair follow, cap_stream apart)
    // (the block sum and maximum are wave reductions -- wave_sum_double / wave_max_float: no LDS scratch)

    const int tid = threadIdx.x;
    constexpr int nthr = kAudioThreads;
    const SlotCfg &cfg = cfgs[slot];
    const SlotDyn dyn = dyns[slot];
    const BlockPlan *pl = plans + (size_t)slot * (NB + 1);
    const int j0 = pl[b].j0, j1 = pl[b + 1].j0, n = j1 - j0;
    const float2 *iq = cfg.iq + (size_t)dyn.hist_parity * ((size_t)kIqHist + cfg.cap_iq) + kIqHist;   // iq[j] valid for j >= -kIqHist
    float *d = cfg.d;
    float lmax = 0.0f;                         // aOutputCeil starts at 0 each block (ModemAnalog.cpp:73)
    double lsum = 0.0;
    int lcount = 0;

    if (cfg.modem == CSDR_MODEM_NBFM || cfg.modem == CSDR_MODEM_FM) {
        return;                                 // freqdem needs no block-wide maximum: done inside demod_audio_interp
    } else if (cfg.modem == CSDR_MODEM_AM) {
        const int halo = kAmTaps - 1;
        for (int i = tid; i < n + halo; i += nthr) {
            const float2 x = iq[j0 - halo + i];
            s_a[i] = sqrtf(x.x * x.x + x.y * x.y);
        }
        __syncthreads();
        // outputs i and i + 64 of a lane as one pair: a two-address LDS read and one packed multiply-add per tap serve both (the partner past the
        // block's end reads staged or stale LDS inside the carve -- cap_stream >= n + 127 + 64 -- and is never stored)
        for (int i = tid; i < n; i += 2 * nthr) {
            cpx acc = cpx_make(0.f, 0.f);
#pragma unroll 17
            for (int t = 0; t < kAmTaps; ++t) acc = cpx_fma_s(mc->am_taps[t], cpx_make(s_a[i + halo - t], s_a[i + nthr + halo - t]), acc);
            d[j0 + i] = acc.x;
            lmax = fmaxf(lmax, acc.x);
            if (i + nthr < n) { d[j0 + i + nthr] = acc.y; lmax = fmaxf(lmax, acc.y); }
        }
    } else if (cfg.modem == CSDR_MODEM_DSB) {
        // ModemDSB::demodulate -> ampmodem_demodulate, DSB with suppressed carrier (liquid 1.5.0 ampmodem_demod_dsb_pll_costas):
        //   v = x e^{-j theta};  e = Re v > 0 ? Im v : -Im v;  dtheta += W(alpha e);  theta += W(beta e);  theta += dtheta;
        //   y = Re v / mod_index          (alpha = 0.001, beta = sqrt(0.001), W = the oscillator's phase-word quantisation)
        // Every sample's phase depends on the previous one: ONE thread walks the whole batch of this demodulator (the
        // workgroup of block 0; the table and each block's samples are staged in LDS by all of its threads).
        if (b != 0) return;
        float *s_tab = reinterpret_cast<float *>(smem);                        // 1024-entry sine table
        float2 *s_x = reinterpret_cast<float2 *>(s_tab + 1024);                // one block of resampled IQ (<= kModemMaxBlockIq)
        for (int i = tid; i < 1024; i += nthr) s_tab[i] = sintab[i];
        uint32_t th = cfg.pll[0], dth = cfg.pll[1];
        const float alpha = 0.001f, beta = sqrtf(0.001f);                      // nco_crcf_pll_set_bandwidth(0.001)
        for (int bb = 0; bb < NB; ++bb) {
            const int jb = pl[bb].j0, nb = pl[bb + 1].j0 - jb;
            __syncthreads();
            for (int i = tid; i < nb; i += nthr) s_x[i] = iq[jb + i];
            __syncthreads();
            if (tid == 0) {
                float mx = 0.0f;
                for (int i = 0; i < nb; ++i) {
                    const float2 x = s_x[i];
                    const unsigned idx = (th + (1u << 21)) >> 22;
                    const float sn = s_tab[idx & 1023], cs = s_tab[(idx + 256) & 1023];
                    const float vr = __fadd_rn(__fmul_rn(x.x, cs), __fmul_rn(x.y, sn));       // mix down, the reference's operation order
                    const float vi = __fsub_rn(__fmul_rn(x.y, cs), __fmul_rn(x.x, sn));
                    const float e = vr > 0.0f ? vi : -vi;
                    dth += nco_phase_word_dev(alpha * e);
                    th += nco_phase_word_dev(beta * e);
                    th += dth;
                    const float y = vr / 0.5f;
                    d[jb + i] = y;
                    mx = fmaxf(mx, y);
                }
                cfg.blockmax[bb] = mx;
                cfg.bout[bb].level_accum = 0.0; cfg.bout[bb].level_count = 0; cfg.bout[bb].audio_peak = 0.f;
                if (NB == 1) gain_step_single(cfg, dyn, pl, mx);
            }
        }
        if (tid == 0) { cfg.pll[0] = th; cfg.pll[1] = dth; }
        return;
    } else if (cfg.modem == CSDR_MODEM_CW) {
        // ModemCW::demodulate (ModemCW.cpp:155-198) up to the gain: msresamp_cccf interpolation of the IQ stream to the audio
        // rate (arbitrary stage, then x2 half-band stages: the structure of the real audio interpolator, on complex samples),
        // mix up by the beep oscillator, upper-sideband output of the c2r Hilbert transform.  The block's outputs [A0, A1)
        // and the 4m samples the Hilbert window reaches back are recomputed from the resampled-IQ stream (history included).
        const ResampCfg &au = cfg.rs_au;
        const int aS = au.S, H = 4 * kHilbM;
        const int64_t A0 = (int64_t)pl[b].q0 << aS, A1 = (int64_t)pl[b + 1].q0 << aS;
        const int n_audio = (int)(A1 - A0);
        float2 *s_iq = reinterpret_cast<float2 *>(smem);          // staged IQ window [kCwIqWin]
        float2 *w0 = s_iq + kCwIqWin, *w1 = w0 + cap_cw;          // ping-pong stage arrays
        const float *arms = arms_all + (size_t)au.arms_idx * kArms * kArmTaps;
        int64_t lo[kMaxHb + 1], hi[kMaxHb + 1];
        lo[aS] = A0 - H; hi[aS] = A1;
        for (int st = aS - 1; st >= 0; --st) { lo[st] = (lo[st + 1] >> 1) - (2 * au.m_x[st] - 1); hi[st] = (hi[st + 1] + 1) >> 1; }
        const int nv = (int)(hi[0] - lo[0]);
        const int64_t jlo = (((int64_t)dyn.aphase0 + lo[0] * (int64_t)au.step) >> 24) - (kArmTaps - 1);
        const int64_t jhi = nv > 0 ? (((int64_t)dyn.aphase0 + (hi[0] - 1) * (int64_t)au.step) >> 24) + 1 : jlo;
        const int nwin = (int)(jhi - jlo);
        for (int i = tid; i < nwin; i += nthr) {
            const int64_t j = jlo + i;
            s_iq[i] = j >= -(int64_t)kIqHist ? iq[j] : make_float2(0.f, 0.f);
        }
        __syncthreads();
        // arbitrary stage: v[q], q in [lo[0], hi[0])
        for (int i = tid; i < nv; i += nthr) {
            const int64_t P = (int64_t)dyn.aphase0 + (lo[0] + i) * (int64_t)au.step;
            const float *h = arms + (int)((P & 0xFFFFFF) >> 16) * kArmTaps;
            const float2 *z = s_iq + ((P >> 24) - (kArmTaps - 1) - jlo);
            float ar = 0.f, ai = 0.f;
#pragma unroll
            for (int t = 0; t < kArmTaps; ++t) { ar = fmaf(h[t], z[t].x, ar); ai = fmaf(h[t], z[t].y, ai); }
            w0[i] = make_float2(ar, ai);
        }
        __syncthreads();
        // x2 stages: w'[2q] = w[q - m], w'[2q + 1] = sum_j h1[j] (w[q - j] + w[q - (2m - 1) + j])
        float2 *src = w0, *dst = w1;
        for (int st = 0; st < aS; ++st) {
            const int m = au.m_x[st];
            const int64_t olo = lo[st + 1], ilo = lo[st];
            const int nout = (int)(hi[st + 1] - olo);
            const int qoff = (int)((olo >> 1) - ilo), par0 = (int)(olo & 1);
            for (int i = tid; i < nout; i += nthr) {
                const int a = i + par0, qi = qoff + (a >> 1);
                float2 v;
                if ((a & 1) == 0) v = src[qi - m];
                else {
                    v = make_float2(0.f, 0.f);
                    for (int j = 0; j < m; ++j) {
                        const float hj = au.h_x[st][j];
                        const float2 p = src[qi - j], q2 = src[qi - (2 * m - 1) + j];
                        v.x = fmaf(hj, p.x + q2.x, v.x); v.y = fmaf(hj, p.y + q2.y, v.y);
                    }
                }
                dst[i] = v;
            }
            __syncthreads();
            float2 *t = src; src = dst; dst = t;
        }
        // beep oscillator (mix up, then step: audio sample a uses theta0 + a dtheta), in place; src[i] is audio sample A0 - H + i
        for (int i = tid; i < n_audio + H; i += nthr) {
            const int64_t a = A0 - H + i;
            float sn, cs;
            nco_sincos(sintab, dyn.ssb_theta0 + (uint32_t)a * dyn.cw_dtheta, sn, cs);
            const float2 v = src[i];
            src[i] = make_float2(v.x * cs - v.y * sn, v.y * cs + v.x * sn);
        }
        __syncthreads();
        // Hilbert c2r, upper sideband: yi - yq (as in the SSB path, taps of firhilbf_create(5, 60))
        for (int i = tid; i < n_audio; i += nthr) {
            const int k = i + H;
            const float yi = src[k - 2 * kHilbM].x;
            float yq = 0.f;
#pragma unroll
            for (int t = 0; t < 2 * kHilbM; ++t) yq = fmaf(mc->hilb60[t], src[k - (2 * t + 1)].y, yq);
            const float v = yi - yq;
            cfg.audio[A0 + i] = v;                                // unscaled: demod_audio_interp applies the block gain
            lmax = fmaxf(lmax, v);                                // signed maximum from 0 (:184-190)
        }
    } else {  // USB / LSB
        const bool usb = (cfg.modem == CSDR_MODEM_USB);
        const int hh = 4 * kHilbM;                 // Hilbert span
        const int pre = kSsbWarm + hh;             // samples before j0 that are processed
        const int tot = n + pre;
        // the two streams (real, imaginary) travel interleaved: one 8-byte LDS read and one packed multiply-add per tap serve both
        cpx *s_ab = reinterpret_cast<cpx *>(smem), *s_cd = s_ab + cap_stream;
        // 1. shift by fs/4 (oscillator is stepped BEFORE use: theta_j = theta0 + (j+1) * 2^30)
        for (int i = tid; i < tot; i += nthr) {
            const int j = j0 - pre + i;
            const float2 x = iq[j];
            float s, c;
            nco_sincos(sintab, dyn.ssb_theta0 + (uint32_t)(j + 1) * (1u << 30), s, c);
            s_ab[i] = usb ? cpx_make(x.x * c + x.y * s, x.y * c - x.x * s)      // mix down
                          : cpx_make(x.x * c - x.y * s, x.y * c + x.x * s);     // mix up
        }
        __syncthreads();
        // 2. the three Butterworth sections (iirfilt_crcf_execute, ModemUSB.cpp:57) as their 128-tap impulse response: every
        //    output is an independent dot product (the recursion ran 300 dependent steps on two lanes), real taps on both streams
        for (int i = kSsbWarm + tid; i < tot; i += nthr) {
            cpx acc = cpx_make(0.f, 0.f);
#pragma unroll 32
            for (int k = 0; k < kSsbFir; ++k) acc = cpx_fma_s(mc->ssb_fir[k], s_ab[i - k], acc);
            s_cd[i] = acc;
        }
        __syncthreads();
        // 3. shift back (same oscillator phase), in place
        for (int i = kSsbWarm + tid; i < tot; i += nthr) {
            const int j = j0 - pre + i;
            float s, c;
            nco_sincos(sintab, dyn.ssb_theta0 + (uint32_t)(j + 1) * (1u << 30), s, c);
            const float xr = s_cd[i].x, xi = s_cd[i].y;
            s_cd[i] = usb ? cpx_make(xr * c - xi * s, xi * c + xr * s) : cpx_make(xr * c + xi * s, xi * c - xr * s);
        }
        __syncthreads();
        // 4. Hilbert c2r: yi = re[k - 2m], yq = sum_{n odd} hq[(n-1)/2] im[k - n]; lower = yi + yq, upper = yi - yq
        for (int i = tid; i < n; i += nthr) {
            const int k = i + pre;
            const float yi = s_cd[k - 2 * kHilbM].x;
            float yq = 0.f;
#pragma unroll
            for (int t = 0; t < 2 * kHilbM; ++t) yq = fmaf(mc->hilb[t], s_cd[k - (2 * t + 1)].y, yq);
            const float v = usb ? (yi - yq) : (yi + yq);
            d[j0 + i] = v;
            lmax = fmaxf(lmax, v);
        }
    }
    const float bm = wave_max_float(lmax);
    const double bs = wave_sum_double(lsum);
    if (tid == 0) {
        cfg.blockmax[b] = bm;
        cfg.bout[b].level_accum = bs;
        cfg.bout[b].level_count = lcount;
        cfg.bout[b].audio_peak = 0.f;
        if (NB == 1) gain_step_single(cfg, dyn, pl, bm);
    }
}
// grid = (blocks, demodulators): the blocks of a demodulator are neighbours in dispatch order (measured on C3, ms per batch: 0.048; demodulators
// first 0.060; one XCD per demodulator -- id % 8 -- 0.061).  The audio kernel below is the other way round: 0.105 against 0.125.
CSDR_KERNEL_BANK __launch_bounds__(kAudioThreads) void demod_modem(
    const SlotCfg *__restrict__ cfgs, const SlotDyn *__restrict__ dyns, const int *__restrict__ slot_list,
    const BlockPlan *__restrict__ plans, int NB, int cap_stream, const ModemConsts *__restrict__ mc, const float *__restrict__ sintab,
    const float *__restrict__ arms_all, int cap_cw) {
    demod_modem_body(cfgs, dyns, slot_list[blockIdx.y], (int)blockIdx.x, plans, NB, cap_stream, mc, sintab, arms_all, cap_cw);
}

// ------------------------------------------------------------------------------------------------------------
// D2a': the auto-gain recurrence over the blocks of the batch (ModemAnalog.cpp:70-77, ModemCW.cpp:181-190), once per demodulator:
//     MA += (ceil - MA) 0.025;  MAA += (MA - MAA) 0.025;  ceil = max of the block           (blocks without samples change nothing)
// blockmaa[b] = MAA in force for block b; the end state goes to the other parity copy.  (Every audio workgroup used to replay the
// recurrence up to its own block: quadratic in the blocks per batch.)   grid = auto-gain slots, 64 threads
// ------------------------------------------------------------------------------------------------------------
CSDR_KERNEL_BANK __launch_bounds__(64) void demod_gain_scan(const SlotCfg *__restrict__ cfgs, const SlotDyn *__restrict__ dyns, const int *__restrict__ slot_list,
                                                      const BlockPlan *__restrict__ plans, int NB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_max = reinterpret_cast<float *>(smem);             // [NB] block maxima, then the MAA per block (coalesced traffic on both sides of the serial walk)
    int *s_j0 = reinterpret_cast<int *>(s_max + NB);            // [NB + 1] first resampled sample of every block
    const int slot = slot_list[blockIdx.x], tid = threadIdx.x;
    const SlotCfg &cfg = cfgs[slot];
    const SlotDyn dyn = dyns[slot];
    const BlockPlan *pl = plans + (size_t)slot * (NB + 1);
    for (int i = tid; i < NB; i += 64) s_max[i] = cfg.blockmax[i];
    for (int i = tid; i <= NB; i += 64) s_j0[i] = pl[i].j0;
    __syncthreads();
    const float *agc_in = cfg.agc + 4 * dyn.hist_parity;
    float ceil_ = agc_in[0], ma = agc_in[1], maa = agc_in[2];
    if (tid == 0)
        for (int bb = 0; bb < NB; ++bb) {
            const float mx = s_max[bb];
            if (s_j0[bb + 1] != s_j0[bb]) {                     // a block without samples never reaches demodulate() (ModemAM.cpp:33-36)
                ma = ma + (ceil_ - ma) * 0.025f;
                maa = maa + (ma - maa) * 0.025f;
                ceil_ = mx;
            }
            s_max[bb] = maa;
        }
    __syncthreads();
    for (int i = tid; i < NB; i += 64) cfg.blockmaa[i] = s_max[i];
    if (tid != 0) return;
    float *agc_out = cfg.agc + 4 * (dyn.hist_parity ^ 1);
    agc_out[0] = ceil_; agc_out[1] = ma; agc_out[2] = maa;
}

// ------------------------------------------------------------------------------------------------------------
// D2b: auto-gain (ModemAnalog.cpp:70-86) + msresamp_rrrf to the audio rate (interpolating form: arbitrary stage then
// x2 stages), audio peak and (for useSignalOutput modems) the audio-based level sum.   grid = (slot, block)
// The gain of block b depends on the maxima of the blocks before it: every workgroup replays that short recurrence
// from the batch-start state (ping-pong copy, so the last block's workgroup can publish the end state).  The
// workgroup of the last block also carries the stream tails (resampled IQ, scaled demodulator output) to the
// history regions for the next batch.
// ------------------------------------------------------------------------------------------------------------
// one x2 half-band stage of the audio interpolator on a single wave: a thread forms the output PAIR (2p, 2p + 1) relative to the even index at or
// below the stage's first output: the delayed sample and the filtered one (one lane per output would have half of every wave copy while the other
// half filters).  M = the stage's m when it is one of the reference's (taps in registers, straight-line); M = 0: any m <= kHbMaxM
template <int M>
__device__ __forceinline__ void audio_x2_stage(const float *__restrict__ h, const float *__restrict__ src, float *__restrict__ dst, int qoff, int par0, int nout, int tid,
                                               int m_any = 0) {
    constexpr int NT = M ? M : kHbMaxM;
    const int m = M ? M : m_any;
    float hs[NT];                                           // the stage's taps, fetched once (wave-uniform)
#pragma unroll
    for (int j = 0; j < NT; ++j) hs[j] = h[j];
    for (int p = tid; 2 * p - par0 < nout; p += kAudioThreads) {
        const int qi = qoff + p, i0 = 2 * p - par0;
        const float ve = src[qi - m];
        float vo = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
            if (M || j < m) vo = fmaf(hs[j], src[qi - j] + src[qi - (2 * m - 1) + j], vo);
        if (i0 >= 0) dst[i0] = ve;
        if (i0 + 1 < nout) dst[i0 + 1] = vo;
    }
}

constexpr int kAudioMaxOut = 16384;        // audio samples of one block handled by one workgroup (likewise bounded by the LDS request)
// dynamic LDS: two ping-pong arrays of `cap_out` floats, `cap_win` staged demodulator samples, 64 bytes of scratch

__device__ __forceinline__ void demod_audio_body(
    const SlotCfg *__restrict__ cfgs, const SlotDyn *__restrict__ dyns, const int slot, const int b,
    const BlockPlan *__restrict__ plans, int NB, int cap_out, int cap_win, const float *__restrict__ arms_all, int pass) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_w0 = reinterpret_cast<float *>(smem);
    float *s_w1 = s_w0 + cap_out;
    float *s_d = s_w1 + cap_out;                               // scaled demodulator samples [jlo, jhi)

    const int tid = threadIdx.x;
    constexpr int nthr = kAudioThreads;
    const SlotCfg &cfg = cfgs[slot];
    const SlotDyn dyn = dyns[slot];
    const BlockPlan *pl = plans + (size_t)slot * (NB + 1);
    const ResampCfg &au = cfg.rs_au;
    const int aS = au.S;
    const int64_t Q0 = pl[b].q0, Q1 = pl[b + 1].q0;               // arbitrary-stage outputs of this block
    const bool interp = au.interp != 0;
    // audio samples of this block: interpolating form = arbitrary stage then x2 stages; decimating form (bandwidth above
    // the audio rate, e.g. FM 200 kHz) = /2 stages then arbitrary stage, one audio sample per arbitrary-stage output
    const int64_t A0 = interp ? (Q0 << aS) : Q0, A1 = interp ? (Q1 << aS) : Q1;
    const int n_audio = (int)(A1 - A0);
    const float *arms = arms_all + (size_t)au.arms_idx * kArms * kArmTaps;
    // FM stereo runs this kernel twice (ModemFMStereo.cpp:189,236: two msresamp_rrrf of the same ratio, so the same integer
    // bookkeeping): pass 0 is the FM path into fms_m, pass 1 resamples the stereo-difference stream fms_mix left in cfg.d into fms_s
    const bool fms = cfg.modem == CSDR_MODEM_FMS, plain = fms && pass == 1;
    const bool autogain = !(cfg.modem == CSDR_MODEM_NBFM || cfg.modem == CSDR_MODEM_FM || fms);
    const float2 *iq = cfg.iq + (size_t)dyn.hist_parity * ((size_t)kIqHist + cfg.cap_iq) + kIqHist;   // iq[j], j >= -kIqHist
    if (cfg.modem == CSDR_MODEM_CW) {
        // ModemCW.cpp:181-203: the auto-gain of block b from the maxima of the blocks before it (recurrence replayed from the
        // batch-entering state), gain in dB and back as the reference does, applied to the unscaled audio demod_modem wrote
        const float maa = cfg.blockmaa[b];                       // demod_gain_scan
        const float gain_db = 10.0f * log10f(0.5f / maa);
        const float g = powf(10.0f, gain_db / 10.0f);
        const int aS = cfg.rs_au.S;
        const int a0 = pl[b].q0 << aS, n_audio = (pl[b + 1].q0 << aS) - a0;
        float lpk = 0.f;
        double lsum = 0.0;
        for (int i = tid; i < n_audio; i += nthr) {
            const float v = cfg.audio[a0 + i] * g;
            cfg.audio[a0 + i] = v;
            lpk = fmaxf(lpk, fabsf(v));
            lsum += (double)fabsf(v);
        }
        const float pk = wave_max_float(lpk);
        const double sm = wave_sum_double(lsum);
        if (tid == 0) {
            cfg.bout[b].audio_peak = pk; cfg.bout[b].level_accum = sm; cfg.bout[b].level_count = n_audio;
        }
        return;
    }
    if (cfg.modem == CSDR_MODEM_IQ) {
        // ModemIQ::demodulate (ModemIQ.cpp:41-61): stereo frames (imag, real) of the resampled IQ, no filtering, no gain;
        // level from the IQ magnitudes like the other non-signal-output modems (DemodulatorThread.cpp:156-162)
        const int j0 = pl[b].j0, n_iq = pl[b + 1].j0 - j0;
        float lpk = 0.f;
        double lsum = 0.0;
        float2 *ao = reinterpret_cast<float2 *>(cfg.audio) + j0;
        for (int i = tid; i < n_iq; i += nthr) {
            const float2 x = iq[j0 + i];
            ao[i] = make_float2(x.y, x.x);
            lpk = fmaxf(lpk, fmaxf(fabsf(x.x), fabsf(x.y)));
            lsum += sqrt((double)x.x * (double)x.x + (double)x.y * (double)x.y);
        }
        const float pk = wave_max_float(lpk);
        const double sm = wave_sum_double(lsum);
        if (tid == 0) { cfg.bout[b].audio_peak = pk; cfg.bout[b].level_accum = sm; cfg.bout[b].level_count = n_iq; }
        return;
    }
    const float fm_ref = 1.0f / (2.0f * 3.14159265358979323846f * 0.5f);   // freqdem_create(kf = 0.5): 1 / (2 pi kf)
    const float *dh_in = cfg.dh + (size_t)kDHist * dyn.hist_parity;

    // gains of block b (g_cur) and of the block before it (g_prev); block -1 means "previous batch" (already scaled)
    float g_cur = 1.0f, g_prev = 1.0f;
    if (autogain) {                                              // MAA per block from demod_gain_scan
        g_cur = 0.5f / cfg.blockmaa[b];
        if (b > 0) g_prev = 0.5f / cfg.blockmaa[b - 1];
    }

    // backward range propagation.
    //  interpolating: lo[s] / hi[s] = range of the input of x2 stage s (s = 0 is v = arbitrary-stage output), in samples
    //  decimating:    lo[e] / hi[e] = range of the input of /2 stage e in its own index space (e = 0: u = j + abuf0;
    //                 e = aS: the half-band chain output Z the arbitrary stage reads)
    int64_t lo[kMaxHb + 1], hi[kMaxHb + 1];
    int64_t jlo, jhi;
    int nv = 0;
    if (interp) {
        lo[aS] = A0;
        for (int s = aS - 1; s >= 0; --s) lo[s] = (lo[s + 1] >> 1) - (2 * au.m_x[s] - 1);
        hi[aS] = A1;
        for (int s = aS - 1; s >= 0; --s) hi[s] = (hi[s + 1] + 1) >> 1;
        nv = (int)(hi[0] - lo[0]);
        jlo = (((int64_t)dyn.aphase0 + lo[0] * (int64_t)au.step) >> 24) - (kArmTaps - 1);
        jhi = nv > 0 ? (((int64_t)dyn.aphase0 + (hi[0] - 1) * (int64_t)au.step) >> 24) + 1 : jlo;
    } else {
        lo[aS] = (((int64_t)dyn.aphase0 + Q0 * (int64_t)au.step) >> 24) - (kArmTaps - 1);
        hi[aS] = n_audio > 0 ? (((int64_t)dyn.aphase0 + (Q1 - 1) * (int64_t)au.step) >> 24) + 1 : lo[aS];
        for (int e = aS - 1; e >= 0; --e) { lo[e] = 2 * lo[e + 1] - (4 * au.m_x[e] - 2); hi[e] = 2 * hi[e + 1] - 1; }
        jlo = lo[0] - (int64_t)dyn.abuf0;
        jhi = hi[0] > lo[0] ? hi[0] - (int64_t)dyn.abuf0 : jlo;
    }

    // 0. stage the scaled demodulator samples [jlo, jhi) the cascade touches
    const int nwin = (int)(jhi - jlo);
    const int jb0 = pl[b].j0, jbp = b > 0 ? pl[b - 1].j0 : 0;
    // g_cur / g_prev are the gains of blocks b and b - 1 only when both hold samples (empty blocks do not step the gain)
    const bool fast_gain = pl[b + 1].j0 > jb0 && (b == 0 || jb0 > jbp);
    // the scaled demodulator sample j of this batch's stream (j < 0: the carried history), as the modem hands it to its audio resampler
    auto demod_sample = [&](int64_t j) -> float {
        if (plain) return j < 0 ? (j >= -(int64_t)kDHist ? dh_in[kDHist + j] : 0.f) : cfg.d[j];
        if (!autogain) {
            // NBFM / FM (ModemNBFM.cpp:36, ModemFM.cpp:36): m[j] = atan2f(Im(x_j conj x_{j-1}), Re(..)) / (2 pi kf), gain 1.
            // Formed here from the resampled IQ stream (history included: x_{-1} of a fresh demodulator is 0 -> m = 0).
            if (j < -(int64_t)(kIqHist - 1)) return 0.f;
            const float2 c = iq[j], p = iq[j - 1];
            return atan2f(c.y * p.x - c.x * p.y, c.x * p.x + c.y * p.y) * fm_ref;
        }
        if (j < 0) return j >= -(int64_t)kDHist ? dh_in[kDHist + j] : 0.f;
        if (fast_gain && j >= jb0) return cfg.d[j] * g_cur;
        if (fast_gain && j >= jbp) return cfg.d[j] * g_prev;
        // more than one block back, or empty blocks nearby (tiny blocks): replay the gain of the block that holds j
        int bb = b;
        while (bb > 0 && j < pl[bb].j0) --bb;
        return cfg.d[j] * (0.5f / cfg.blockmaa[bb]);                  // block bb holds sample j, so it stepped the gain
    };
    if (!plain && !autogain) {
        // the FM discriminator's window, four samples of a lane at a time: all eight IQ loads are in flight before the first arctangent (the
        // general form below is a load, a wait and an arctangent per sample).  Indices are clamped instead of guarded, the results selected.
        const int64_t jmin = -(int64_t)(kIqHist - 1);
        for (int i0 = tid; i0 < nwin; i0 += 4 * nthr) {
            float2 c4[4], p4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t j = jlo + min(i0 + u * nthr, nwin - 1);
                const int64_t jc = j < jmin ? jmin : j;
                c4[u] = iq[jc]; p4[u] = iq[jc - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * nthr;
                const float2 c = c4[u], pq = p4[u];
                const float v = atan2f(c.y * pq.x - c.x * pq.y, c.x * pq.x + c.y * pq.y) * fm_ref;
                if (i < nwin) s_d[i] = (jlo + i) < jmin ? 0.f : v;
            }
        }
    } else
    for (int i = tid; i < nwin; i += nthr) s_d[i] = demod_sample(jlo + i);
    // the last block's own scaled samples are the scope tap (ModemAnalog::getDemodOutputData, DemodulatorThread.cpp:293-305): the WHOLE
    // block, up to DEMOD_VIS_SIZE -- those the cascade of this block already staged, and the trailing ones a decimating cascade leaves
    // to the next block's outputs
    if (b == NB - 1 && !fms) {
        const int n_own = pl[b + 1].j0 - jb0;
        const int ns = max(0, min(n_own, kScopeMax));
        const int staged = max(0, min(ns, (int)(jhi - (int64_t)jb0)));
        __syncthreads();
        for (int i = tid; i < ns; i += nthr) cfg.scope[i] = (i < staged && (int64_t)jb0 >= jlo) ? s_d[(int)((int64_t)jb0 - jlo) + i] : demod_sample((int64_t)jb0 + i);
        if (tid == 0) *cfg.scope_n = ns;
    }
    // the filter arms of this thread's first two arbitrary-stage outputs travel while the staging above lands
    float2 hv[2][kArmTaps / 2];
    int zoff[2] = {0, 0};
    float *src = s_w0, *dst = s_w1;
    if (interp) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = tid + r * nthr;
        if (i < nv) {
            const int64_t P = (int64_t)dyn.aphase0 + (lo[0] + i) * (int64_t)au.step;
            zoff[r] = (int)((P >> 24) - (kArmTaps - 1) - jlo);
            const float2 *h2 = reinterpret_cast<const float2 *>(arms + (int)((P & 0xFFFFFF) >> 16) * kArmTaps);
#pragma unroll
            for (int t = 0; t < kArmTaps / 2; ++t) hv[r][t] = h2[t];
        }
    }
    __syncthreads();
    // 1. arbitrary stage: v[q] for q in [lo[0], hi[0]) into s_w0
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = tid + r * nthr;
        if (i < nv) {
            const float *z = s_d + zoff[r];
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < kArmTaps / 2; ++t) { acc = fmaf(hv[r][t].x, z[2 * t], acc); acc = fmaf(hv[r][t].y, z[2 * t + 1], acc); }
            s_w0[i] = acc;
        }
    }
    // the outputs after a thread's first two, two at a time: the filter arms of both are requested before either dot product
    for (int i0 = tid + 2 * nthr; i0 < nv; i0 += 2 * nthr) {
        float2 ha[2][kArmTaps / 2];
        int zo[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = min(i0 + r * nthr, nv - 1);                       // (an index past the end re-reads the last output's arm: never stored)
            const int64_t P = (int64_t)dyn.aphase0 + (lo[0] + i) * (int64_t)au.step;
            zo[r] = (int)((P >> 24) - (kArmTaps - 1) - jlo);
            const float2 *h2 = reinterpret_cast<const float2 *>(arms + (int)((P & 0xFFFFFF) >> 16) * kArmTaps);
#pragma unroll
            for (int t = 0; t < kArmTaps / 2; ++t) ha[r][t] = h2[t];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = i0 + r * nthr;
            const float *z = s_d + zo[r];
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < kArmTaps / 2; ++t) { acc = fmaf(ha[r][t].x, z[2 * t], acc); acc = fmaf(ha[r][t].y, z[2 * t + 1], acc); }
            if (i < nv) s_w0[i] = acc;
        }
    }
    __syncthreads();
    // 2. x2 stages: w'[2q] = w[q - m], w'[2q+1] = sum_j h1[j] w[q - j]
    for (int s = 0; s < aS; ++s) {
        const int m = au.m_x[s];
        const int64_t olo = lo[s + 1], ohi = hi[s + 1], ilo = lo[s];
        const int nout = (int)(ohi - olo);
        const int qoff = (int)((olo >> 1) - ilo), par0 = (int)(olo & 1);
        // the reference's interpolators are m = 10, 5, 3, 3, ... (msresamp2 at 60 dB): those run with the tap count known to the compiler
        switch (m) {
            case 3: audio_x2_stage<3>(au.h_x[s], src, dst, qoff, par0, nout, tid); break;
            case 5: audio_x2_stage<5>(au.h_x[s], src, dst, qoff, par0, nout, tid); break;
            case 10: audio_x2_stage<10>(au.h_x[s], src, dst, qoff, par0, nout, tid); break;
            default: audio_x2_stage<0>(au.h_x[s], src, dst, qoff, par0, nout, tid, m); break;
        }
        __syncthreads();
        float *t = src; src = dst; dst = t;
    }
    } else {
        // ---- decimating form (msresamp_rrrf with rate < 1): /2 half-band stages, then the arbitrary stage
        __syncthreads();
        const float *in = s_d;                                  // in[i - lo[e]] = input i of stage e
        float *ping = s_w0, *pong = s_w1;
        const float zeta = 1.0f / (float)(1 << aS);
        for (int e = 0; e < aS; ++e) {
            const int m = au.m_x[e];
            const int64_t ilo = lo[e], olo = lo[e + 1];
            const int nout = (int)(hi[e + 1] - olo);
            const float sc = (e == aS - 1) ? zeta : 1.0f;
            // y[k] = x[2k - 2m + 1] + sum_j h[j] (x[2k - 2j] + x[2k - 4m + 2 + 2j])
            float hs[kHbMaxM];
#pragma unroll
            for (int j = 0; j < kHbMaxM; ++j) hs[j] = au.h_x[e][j];
            const int xoff = (int)(2 * olo - ilo);
            for (int i = tid; i < nout; i += nthr) {
                const float *x = in + (xoff + 2 * i);
                float v = x[-2 * m + 1];
#pragma unroll
                for (int j = 0; j < kHbMaxM; ++j)
                    if (j < m) v = fmaf(hs[j], x[-2 * j] + x[-4 * m + 2 + 2 * j], v);
                ping[i] = v * sc;
            }
            __syncthreads();
            in = ping;
            float *t = ping; ping = pong; pong = t;
        }
        // arbitrary stage on the chain output Z = in[k - lo[aS]]
        for (int i = tid; i < n_audio; i += nthr) {
            const int64_t P = (int64_t)dyn.aphase0 + (Q0 + i) * (int64_t)au.step;
            const int64_t kq = P >> 24;
            const float *h = arms + (int)((P & 0xFFFFFF) >> 16) * kArmTaps;
            const float *z = in + (kq - (kArmTaps - 1) - lo[aS]);
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < kArmTaps; ++t) acc = fmaf(h[t], z[t], acc);
            ping[i] = acc;
        }
        __syncthreads();
        src = ping;
    }
    // 3. write audio, peak, level (audio-based for the useSignalOutput modems, IQ-based |x| sum for NBFM / FM:
    //    DemodulatorThread.cpp:142-152, abMagnitude :49-57)
    float lpk = 0.f;
    double lsum = 0.0;
    const int aoff = (int)A0;
    float *aout = fms ? (pass ? cfg.fms_s : cfg.fms_m) : cfg.audio;
    for (int i = tid; i < n_audio; i += nthr) {
        const float v = src[i];
        aout[aoff + i] = v;
        lpk = fmaxf(lpk, fabsf(v));
        if (autogain) lsum += (double)fabsf(v);
    }
    const int n_iq = pl[b + 1].j0 - jb0;
    if (!autogain && !plain)
        for (int i = tid; i < n_iq; i += nthr) {
            const float2 x = iq[jb0 + i];
            lsum += sqrt((double)x.x * (double)x.x + (double)x.y * (double)x.y);
        }
    const float pk = wave_max_float(lpk);
    const double sm = wave_sum_double(lsum);
    if (tid == 0 && !plain) {                                    // (FM stereo: fms_out sets the peak of the finished stereo frames)
        cfg.bout[b].audio_peak = pk;
        cfg.bout[b].level_accum = sm;
        cfg.bout[b].level_count = autogain ? n_audio : n_iq;
    }
    // 4. last block of an auto-gain modem: publish the gain state and the scaled demodulator tail (other parity)
    if (b == NB - 1 && (autogain || plain)) {
        const int J = pl[NB].j0;
        for (int td = tid; td < kDHist; td += nthr) {
            const int j = J - kDHist + td;
            float dv;
            if (j < 0) dv = dh_in[kDHist + j];
            else if (plain) dv = cfg.d[j];
            else if (fast_gain && j >= jb0) dv = cfg.d[j] * g_cur;
            else if (fast_gain && j >= jbp) dv = cfg.d[j] * g_prev;
            else {
                int bb = b;
                while (bb > 0 && j < pl[bb].j0) --bb;
                dv = cfg.d[j] * (0.5f / cfg.blockmaa[bb]);
            }
            (cfg.dh + (size_t)kDHist * (dyn.hist_parity ^ 1))[td] = dv;
        }
    }
}
// grid = (demodulators, blocks)
CSDR_KERNEL_BANK __launch_bounds__(kAudioThreads) void demod_audio_interp(
    const SlotCfg *__restrict__ cfgs, const SlotDyn *__restrict__ dyns, const int *__restrict__ slot_list,
    const BlockPlan *__restrict__ plans, int NB, int cap_out, int cap_win, const float *__restrict__ arms_all, int pass) {
    demod_audio_body(cfgs, dyns, slot_list[blockIdx.x], (int)blockIdx.y, plans, NB, cap_out, cap_win, arms_all, pass);
}

// A ONE-block batch (the real-time shape): modem and audio of a demodulator in ONE launch -- its single block is one wave in either kernel, the
// auto-gain of the block depends on the carried state alone (gain_step_single), and a call's time is the chain of its dependent launches (DESIGN 6:
// front-end -> modem -> audio was 19 + 8 + 17 us).  The wave's own stores (the modem's output, the gain) are read back behind a workgroup fence.
// grid = demodulators of the audio stage
CSDR_KERNEL_BANK __launch_bounds__(kAudioThreads) void demod_modem_audio1(
    const SlotCfg *__restrict__ cfgs, const SlotDyn *__restrict__ dyns, const int *__restrict__ slot_list, const BlockPlan *__restrict__ plans,
    int cap_stream, const ModemConsts *__restrict__ mc, const float *__restrict__ sintab, const float *__restrict__ arms_all, int cap_cw,
    int cap_out, int cap_win) {
    const int slot = slot_list[blockIdx.x];
    const int modem = cfgs[slot].modem;
    if (modem == CSDR_MODEM_AM || modem == CSDR_MODEM_USB || modem == CSDR_MODEM_LSB || modem == CSDR_MODEM_DSB || modem == CSDR_MODEM_CW) {
        demod_modem_body(cfgs, dyns, slot, 0, plans, 1, cap_stream, mc, sintab, arms_all, cap_cw);
#if defined(__AMDGCN__)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // the wave's global stores (the modem's output, the gain) before its loads of them
        __builtin_amdgcn_s_dcache_inv();                            // ... also where the compiler reads a wave-uniform one of them (the block's gain) through the scalar cache, which vector stores do not update
#endif
        __syncthreads();
    }
    demod_audio_body(cfgs, dyns, slot, 0, plans, 1, cap_out, cap_win, arms_all, 0);
}

}  // namespace csdr
