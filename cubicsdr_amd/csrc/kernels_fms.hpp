// kernels_fms.hpp -- FM stereo (ModemFMStereo.cpp:163-289) on the resampled-IQ stream of a demodulator slot.
//
//   reference, per block:                                                  here, per batch of blocks:
//     d      = freqdem(iq)                                 (:178)            demod_audio_interp pass 0 (mono: the FM path) and fms_pre
//     mono   = msresamp_rrrf(d)                            (:189)            demod_audio_interp pass 0 -> fms_m
//     per sample  x = r2c(d); v = pilotBP(x);              (:198-205)        fms_pre  (x: every sample independent)
//                 PLL on v, step; theta                    (:206-218)        fms_pll  (ONE thread walks the batch: v and theta recur)
//                 y = x e^{-j theta} e^{-j theta}; s = c2r(y).lower (:220-226) fms_mix  (independent given theta)
//     stereo = msresamp_rrrf(s)                            (:236)            demod_audio_interp pass 1 -> fms_s
//     l / r  = lowpass(deemph(0.568 (mono -/+ stereo)))    (:263-287)        fms_out  (one FIR per channel: design::fms_output_fir)
//
// The pilot loop is the one sequential piece: a 10th-order band-pass with pole radii 0.998 feeding a phase-locked loop of
// bandwidth 0.25 around a 1024-entry table oscillator.  That loop limit-cycles on the table's 6e-3 rad phase steps, so its
// trajectory -- and with it the L-R channel -- is reproducible only to ~2e-3 between ANY two executions that differ in the last
// bit of one input ("FM stereo"); the arithmetic below keeps the reference's operation order all the same.
#pragma once
#include "kernels_demod.hpp"

namespace csdr {

constexpr int kFmsYHist = 32;        // down-mixed samples kept in front of a batch (c2r Hilbert window: 4 m = 20)
constexpr int kFmsFirMax = 2048;     // longest de-emphasis * low-pass response (175 + ~70 taps at 48 kHz; ~1400 at 192 kHz)
constexpr int kFmsStateWords = 24;   // v1, v2 of five sections (re, im) + phase word + frequency word

__device__ inline float fms_freqdem(const float2 *iq, int64_t j) {
    // freqdem_demodulate (liquid 1.5.0, kf = 0.5): arg(x_j conj x_{j-1}) / (2 pi kf); before the first sample the state is 0
    if (j < -(int64_t)(kIqHist - 1)) return 0.f;
    const float2 c = iq[j], p = iq[j - 1];
    return atan2f(c.y * p.x - c.x * p.y, c.x * p.x + c.y * p.y) * (1.0f / (2.0f * 3.14159265358979323846f * 0.5f));
}

// ---- fms_pre: x[j] = firhilbf_r2c_execute(d[j]) = d[j - 2m] + i sum_t hq[t] d[j - (2t + 1)]      grid = (slot, block)
// dynamic LDS: (cap_blk + 4 m) floats
CSDR_KERNEL __launch_bounds__(64) void fms_pre(const SlotCfg *__restrict__ cfgs, const SlotDyn *__restrict__ dyns, const int *__restrict__ slot_list,
                                              const BlockPlan *__restrict__ plans, int NB, const ModemConsts *__restrict__ mc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_d = reinterpret_cast<float *>(smem);
    const int slot = slot_list[blockIdx.x], b = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
    const SlotCfg &cfg = cfgs[slot];
    const SlotDyn dyn = dyns[slot];
    const BlockPlan *pl = plans + (size_t)slot * (NB + 1);
    const int j0 = pl[b].j0, n = pl[b + 1].j0 - j0, H = 4 * kHilbM;
    const float2 *iq = cfg.iq + (size_t)dyn.hist_parity * ((size_t)kIqHist + cfg.cap_iq) + kIqHist;
    for (int i = tid; i < n + H; i += nthr) s_d[i] = fms_freqdem(iq, (int64_t)j0 - H + i);
    __syncthreads();
    for (int i = tid; i < n; i += nthr) {
        const int k = i + H;
        float yq = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * kHilbM; ++t) yq = fmaf(mc->hilb60[t], s_d[k - (2 * t + 1)], yq);
        cfg.fms_x[j0 + i] = make_float2(s_d[k - 2 * kHilbM], yq);
    }
}

// ---- fms_pll: the pilot band-pass and the phase-locked loop, one thread per demodulator over the whole batch.   grid = slots
// dynamic LDS: 1024-entry sine table, one block of x (float2) and of theta (uint32)
CSDR_KERNEL __launch_bounds__(kModemThreads) void fms_pll(const SlotCfg *__restrict__ cfgs, const int *__restrict__ slot_list,
                                                         const BlockPlan *__restrict__ plans, int NB, int cap_blk, const float *__restrict__ sintab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_tab = reinterpret_cast<float *>(smem);
    float2 *s_x = reinterpret_cast<float2 *>(s_tab + 1024);
    uint32_t *s_th = reinterpret_cast<uint32_t *>(s_x + cap_blk);
    const int slot = slot_list[blockIdx.x], tid = threadIdx.x, nthr = blockDim.x;
    const SlotCfg &cfg = cfgs[slot];
    const BlockPlan *pl = plans + (size_t)slot * (NB + 1);
    for (int i = tid; i < 1024; i += nthr) s_tab[i] = sintab[i];
    float v1r[5], v1i[5], v2r[5], v2i[5], B0[5], B1[5], B2[5], A1[5], A2[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        v1r[q] = cfg.fms_state[4 * q]; v1i[q] = cfg.fms_state[4 * q + 1]; v2r[q] = cfg.fms_state[4 * q + 2]; v2i[q] = cfg.fms_state[4 * q + 3];
        B0[q] = cfg.fms_b[3 * q]; B1[q] = cfg.fms_b[3 * q + 1]; B2[q] = cfg.fms_b[3 * q + 2]; A1[q] = cfg.fms_a[3 * q + 1]; A2[q] = cfg.fms_a[3 * q + 2];
    }
    uint32_t th = __float_as_uint(cfg.fms_state[20]), dth = __float_as_uint(cfg.fms_state[21]);
    const float alpha = 0.25f, beta = sqrtf(0.25f);                       // nco_crcf_pll_set_bandwidth(stereoPilot, 0.25f)  (:143)
    for (int bb = 0; bb < NB; ++bb) {
        const int jb = pl[bb].j0, nb = pl[bb + 1].j0 - jb;
        __syncthreads();
        for (int i = tid; i < nb; i += nthr) s_x[i] = cfg.fms_x[jb + i];
        __syncthreads();
        if (tid == 0) {
            for (int i = 0; i < nb; ++i) {
                float tr = s_x[i].x, ti = s_x[i].y;
                // iirfilt_crcf_execute, second-order sections in direct form II, the reference binary's summation order:
                //   v0 = x - (a1 v1 + a2 v2);   y = (b1 v1 + b2 v2) + b0 v0
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const float v0r = __fsub_rn(tr, __fadd_rn(__fmul_rn(A1[q], v1r[q]), __fmul_rn(A2[q], v2r[q])));
                    const float v0i = __fsub_rn(ti, __fadd_rn(__fmul_rn(A1[q], v1i[q]), __fmul_rn(A2[q], v2i[q])));
                    tr = __fadd_rn(__fadd_rn(__fmul_rn(B1[q], v1r[q]), __fmul_rn(B2[q], v2r[q])), __fmul_rn(B0[q], v0r));
                    ti = __fadd_rn(__fadd_rn(__fmul_rn(B1[q], v1i[q]), __fmul_rn(B2[q], v2i[q])), __fmul_rn(B0[q], v0i));
                    v2r[q] = v1r[q]; v2i[q] = v1i[q]; v1r[q] = v0r; v1i[q] = v0i;
                }
                // u = v conj(w), w = the oscillator's table value;  phase error = arg u;  pll_step, then step   (:203-218)
                const unsigned idx = (th + (1u << 21)) >> 22;
                const float wr = s_tab[(idx + 256) & 1023], wi = -s_tab[idx & 1023];
                const float ur = __fsub_rn(__fmul_rn(tr, wr), __fmul_rn(ti, wi));
                const float ui = __fadd_rn(__fmul_rn(tr, wi), __fmul_rn(ti, wr));
                const float pe = atan2f(ui, ur);
                dth += nco_phase_word_dev(__fmul_rn(alpha, pe));
                th += nco_phase_word_dev(__fmul_rn(beta, pe));
                th += dth;
                s_th[i] = th;
            }
        }
        __syncthreads();
        for (int i = tid; i < nb; i += nthr) cfg.fms_theta[jb + i] = s_th[i];
    }
    if (tid == 0) {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            cfg.fms_state[4 * q] = v1r[q]; cfg.fms_state[4 * q + 1] = v1i[q]; cfg.fms_state[4 * q + 2] = v2r[q]; cfg.fms_state[4 * q + 3] = v2i[q];
        }
        cfg.fms_state[20] = __uint_as_float(th); cfg.fms_state[21] = __uint_as_float(dth);
    }
}

// ---- fms_mix: y = (x conj w) conj w with w of the stepped oscillator, s = lower-sideband output of firhilbf_c2r_execute   grid = (slot, block)
// dynamic LDS: 2 (cap_blk + 4 m) floats.  s goes to cfg.d (the stream the audio kernel's second pass resamples).
CSDR_KERNEL __launch_bounds__(64) void fms_mix(const SlotCfg *__restrict__ cfgs, const SlotDyn *__restrict__ dyns, const int *__restrict__ slot_list,
                                              const BlockPlan *__restrict__ plans, int NB, int cap_blk, const ModemConsts *__restrict__ mc,
                                              const float *__restrict__ sintab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int H = 4 * kHilbM;
    float *s_r = reinterpret_cast<float *>(smem), *s_i = s_r + cap_blk + H;
    const int slot = slot_list[blockIdx.x], b = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
    const SlotCfg &cfg = cfgs[slot];
    const SlotDyn dyn = dyns[slot];
    const BlockPlan *pl = plans + (size_t)slot * (NB + 1);
    const int j0 = pl[b].j0, n = pl[b + 1].j0 - j0;
    const float2 *yh_in = cfg.fms_yh + (size_t)kFmsYHist * dyn.hist_parity;
    auto mixed = [&](int j) -> float2 {
        if (j < 0) return j >= -kFmsYHist ? yh_in[kFmsYHist + j] : make_float2(0.f, 0.f);
        const float2 x = cfg.fms_x[j];
        float s, c;
        nco_sincos(sintab, cfg.fms_theta[j], s, c);
        const float2 y = make_float2(x.x * c + x.y * s, x.y * c - x.x * s);         // nco_crcf_mix_down, twice (:220-221)
        return make_float2(y.x * c + y.y * s, y.y * c - y.x * s);
    };
    for (int i = tid; i < n + H; i += nthr) {
        const float2 y = mixed(j0 - H + i);
        s_r[i] = y.x; s_i[i] = y.y;
    }
    __syncthreads();
    for (int i = tid; i < n; i += nthr) {
        const int k = i + H;
        float yq = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * kHilbM; ++t) yq = fmaf(mc->hilb60[t], s_i[k - (2 * t + 1)], yq);
        cfg.d[j0 + i] = s_r[k - 2 * kHilbM] + yq;                                    // lower sideband: yi + yq (:225)
    }
    if (b == NB - 1) {                                                               // the window the next batch starts from
        const int J = pl[NB].j0;
        float2 *yh_out = cfg.fms_yh + (size_t)kFmsYHist * (dyn.hist_parity ^ 1);
        for (int t = tid; t < kFmsYHist; t += nthr) yh_out[t] = mixed(J - kFmsYHist + t);
    }
}

// ---- fms_out: left / right = FIR(0.568 (mono -/+ stereo)), interleaved; audio peak of the block      grid = (slot, block)
// dynamic LDS: 2 (cap_au + kFmsFirMax) floats + the taps + 64 bytes
CSDR_KERNEL __launch_bounds__(64) void fms_out(const SlotCfg *__restrict__ cfgs, const SlotDyn *__restrict__ dyns, const int *__restrict__ slot_list,
                                              const BlockPlan *__restrict__ plans, int NB, int cap_au) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int slot = slot_list[blockIdx.x], b = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
    const SlotCfg &cfg = cfgs[slot];
    const SlotDyn dyn = dyns[slot];
    const BlockPlan *pl = plans + (size_t)slot * (NB + 1);
    const int L = cfg.fms_fir_len, H = L - 1;
    float *s_l = reinterpret_cast<float *>(smem), *s_rt = s_l + cap_au + kFmsFirMax, *s_g = s_rt + cap_au + kFmsFirMax;
    // audio samples of this block per channel: one per arbitrary-stage output of the two audio resamplers, 2^S each when they interpolate
    const int ush = cfg.rs_au.interp ? cfg.rs_au.S : 0;
    const int a0 = pl[b].q0 << ush, n = (pl[b + 1].q0 << ush) - a0;
    const float *uh_in = cfg.fms_uh + (size_t)2 * kFmsFirMax * dyn.hist_parity;
    for (int i = tid; i < L; i += nthr) s_g[i] = cfg.fms_fir[i];
    for (int i = tid; i < n + H; i += nthr) {
        const int a = a0 - H + i;
        float ul, ur;
        if (a < 0) { ul = uh_in[kFmsFirMax + a]; ur = uh_in[2 * kFmsFirMax + a]; }      // a >= -H > -kFmsFirMax
        else { const float m = cfg.fms_m[a], s = cfg.fms_s[a]; ul = 0.568f * (m - s); ur = 0.568f * (m + s); }
        s_l[i] = ul; s_rt[i] = ur;
    }
    __syncthreads();
    float lpk = 0.f;
    float2 *out = reinterpret_cast<float2 *>(cfg.audio) + a0;
    for (int i = tid; i < n; i += nthr) {
        const int k = i + H;
        float al = 0.f, ar = 0.f;
        for (int t = 0; t < L; ++t) { const float g = s_g[t]; al = fmaf(g, s_l[k - t], al); ar = fmaf(g, s_rt[k - t], ar); }
        out[i] = make_float2(al, ar);
        lpk = fmaxf(lpk, fmaxf(fabsf(al), fabsf(ar)));
    }
    const float pk = wave_max_float(lpk);
    if (tid == 0) cfg.bout[b].audio_peak = pk;
    if (b == NB - 1) {
        const int A = pl[NB].q0 << ush;
        float *uh_out = cfg.fms_uh + (size_t)2 * kFmsFirMax * (dyn.hist_parity ^ 1);
        for (int t = tid; t < kFmsFirMax; t += nthr) {
            const int a = A - kFmsFirMax + t;
            float ul, ur;
            if (a < 0) { ul = a >= -kFmsFirMax ? uh_in[kFmsFirMax + a] : 0.f; ur = a >= -kFmsFirMax ? uh_in[2 * kFmsFirMax + a] : 0.f; }
            else { const float m = cfg.fms_m[a], s = cfg.fms_s[a]; ul = 0.568f * (m - s); ur = 0.568f * (m + s); }
            uh_out[t] = ul; uh_out[kFmsFirMax + t] = ur;
        }
    }
}

}  // namespace csdr
