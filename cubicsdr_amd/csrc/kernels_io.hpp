// kernels_io.hpp -- the edges of the hot path on the GPU: audio scope, audio mix-down, PCM conversion, ingest swap.
//
// Replaces the arithmetic of (reference file:line):
//   ScopeVisualProcessor::process      src/process/ScopeVisualProcessor.cpp:45-217   (waveform normalisation :64-117, audio FFT +
//                                      double EMA + trackers + log scaling :119-214)
//   audioCallback                      src/audio/AudioThread.cpp:88-240              (per-source gain, mono fan-out, sum, peak normalisation)
//   AudioFileWAV::writePayloadToFileStream  src/audio/AudioFileWAV.cpp:133-157       (anti-clipping scale, float -> int16)
//   SDRThread::readStream IQ swap      src/sdr/SoapySDRThread.cpp:258-266, :300-308  (applied while the block crosses the link)
// Host control flow (which block is current, queue rules, file headers) lives in csdr_io.hip; only arithmetic is here.
// All LDS is dynamic (`smem`).
#pragma once
#include "common.hpp"
#include "kernels_post.hpp"
#include "kernels_spec.hpp"

namespace csdr {

// ---- one AudioThreadInput as the scope sees it ------------------------------------------------------------------------------
// `data` holds n floats.  layout 0: exactly the order AudioThreadInput::data has.  layout 1 / 2: the n floats are n / 2 interleaved
// pairs (a, b) still lying in a demodulator's audio buffer; element i of the AudioThreadInput the reference would have built is
//   layout 1: i < n/2 ? a[i] * scale : b[i - n/2] * scale        (FM stereo tap, DemodulatorThread.cpp:283-289: left | right)
//   layout 2: i < n/2 ? b[i] * scale : a[i - n/2] * scale        (I/Q tap, :273-281: imag * 0.75 | real * 0.75)
// so that a stereo tap is read in place in HBM instead of being re-packed on the host.
struct ScopeFrame {
    const float *data;
    const int32_t *n_dev;        // null, or a device int that bounds n (a count only the producing kernel knows)
    int32_t n, channels, type, layout;
    float scale;
    int32_t sample_rate, input_rate;
    int32_t out_size;            // spectrum points of this frame (fftSize / 2, scaled by sampleRate / inputRate: :196-200), host-computed
    int32_t pad;
};
__device__ __forceinline__ float scope_at(const ScopeFrame &f, int i) {
    if (f.layout == 0) return f.data[i];
    const int half = f.n >> 1;
    const bool first = i < half;
    const int k = first ? i : i - half;
    const float v = f.data[2 * k + ((f.layout == 1) == first ? 0 : 1)];
    return rounded(v * f.scale);
}
__device__ __forceinline__ ScopeFrame scope_load(const ScopeFrame *frames, int i) {
    ScopeFrame f = frames[i];
    if (f.n_dev) f.n = max(0, min(f.n, *f.n_dev));
    return f;
}

struct ScopeMeta {               // per produced item
    int32_t mode, spectrum, channels, input_rate, sample_rate, fft_size, n_floats, pad;
    double fft_floor, fft_ceil;
};

constexpr int kScopeThreads = 256;

// ---- waveform (:64-117).  grid = frames.  points[f] has room for 2 * max_n floats ------------------------------------------------
CSDR_KERNEL __launch_bounds__(kScopeThreads) void scope_wave(const ScopeFrame *__restrict__ frames, int max_scope_samples, int max_n,
                                                            float *__restrict__ points, ScopeMeta *__restrict__ meta) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_red = reinterpret_cast<float *>(smem);
    const ScopeFrame fr = scope_load(frames, blockIdx.x);
    const int tid = threadIdx.x;
    float *out = points + (size_t)blockIdx.x * 2 * max_n;
    const int shown = min(fr.n, max_scope_samples);                       // iMax of the peak search and of the plain Y trace
    float pk = 1.0f;                                                       // "float peak = 1.0f": quiet frames are not blown up
    for (int i = tid; i < shown; i += kScopeThreads) pk = fmaxf(pk, fabsf(scope_at(fr, i)));
    pk = wave_max_to_lane63(pk);
    if ((tid & 63) == 63) s_red[tid >> 6] = pk;
    __syncthreads();
    pk = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    int n_floats, mode;
    if (fr.type == 1) {            // two traces side by side: x runs over [-1, 1) once per half
        const int cnt = fr.n, half = cnt >> 1;
        for (int i = tid; i < cnt; i += kScopeThreads) {
            const double t = (double)(half ? i % half : 0) / (double)cnt;  // position inside its half, as a fraction of the whole
            out[2 * i] = (float)(4.0 * t - 1.0);                           // == ((t * 2 - 0.5) * 2): powers of two commute with the rounding
            out[2 * i + 1] = __fdiv_rn(scope_at(fr, i), pk);
        }
        n_floats = 2 * cnt; mode = 1;
    } else if (fr.type == 2) {     // X / Y pairs
        const int cnt = fr.n;
        for (int i = tid; i < (cnt >> 1); i += kScopeThreads) {
            out[2 * i] = __fdiv_rn(scope_at(fr, 2 * i), pk);
            out[2 * i + 1] = __fdiv_rn(scope_at(fr, 2 * i + 1), pk);
        }
        n_floats = cnt; mode = 2;
    } else {                       // one trace over [-1, 1)
        for (int i = tid; i < shown; i += kScopeThreads) {
            const double q = (double)i / (double)shown;
            out[2 * i] = (float)(2.0 * q - 1.0);                           // == ((q - 0.5) * 2)
            out[2 * i + 1] = __fdiv_rn(scope_at(fr, i), pk);
        }
        n_floats = 2 * shown; mode = 0;
    }
    if (tid == 0) {
        ScopeMeta m;
        m.mode = mode; m.spectrum = 0; m.channels = fr.channels; m.input_rate = fr.input_rate; m.sample_rate = fr.sample_rate;
        m.fft_size = 0; m.n_floats = n_floats; m.pad = 0; m.fft_floor = 0.0; m.fft_ceil = 0.0;
        meta[blockIdx.x] = m;
    }
}

// ---- audio spectrum (:119-214).  ONE workgroup walks the frames in order: the averagers and trackers are recurrences over them.
// L = fftSize <= 4096 complex points through the in-LDS transform of the main spectrum; bins [0, L/2) are kept.
// state: ma / maa double[L/2], trk = {ceil_ma, ceil_maa, floor_ma, floor_maa}.  points[f] has room for L floats.
CSDR_KERNEL __launch_bounds__(kFftThreads) void scope_spectrum(const ScopeFrame *__restrict__ frames, int nf, int L, double rate,
                                                             const float2 *__restrict__ tw4096, double *__restrict__ ma, double *__restrict__ maa,
                                                             double *__restrict__ trk, float *__restrict__ points, ScopeMeta *__restrict__ meta) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *sa = reinterpret_cast<float2 *>(smem), *sb = sa + L;
    double *s_red = reinterpret_cast<double *>(sb + L);                    // [2][4 waves]
    const int tid = threadIdx.x, H = L >> 1;
    double t_ceil_ma = trk[0], t_ceil_maa = trk[1], t_floor_ma = trk[2], t_floor_maa = trk[3];
    for (int f = 0; f < nf; ++f) {
        const ScopeFrame fr = scope_load(frames, f);
        const int cnt = fr.channels == 2 ? (fr.n >> 1) : fr.n;            // samples per channel handed to the transform
        for (int i = tid; i < L; i += kFftThreads) {
            float v = 0.f;
            if (i < cnt) v = fr.channels == 2 ? scope_at(fr, i) + scope_at(fr, cnt + i) : scope_at(fr, i);   // stereo: left + right (:132-141)
            sa[i] = make_float2(v, 0.f);
        }
        __syncthreads();
        const float2 *X = lds_fft(sa, sb, L, tw4096);
        // magnitude -> first averager -> second averager (which sees the NEW first one, :176-177); extrema in double (:179-184)
        double mx = 0.0, mn = 1.0;
        for (int i = tid; i < H; i += kFftThreads) {
            const double a = (double)X[i].x, b = (double)X[i].y;
            const double mag = sqrt(a * a + b * b);
            double m1 = ma[i], m2 = maa[i];
            m1 += (mag - m1) * rate;
            m2 += (m1 - m2) * rate;
            ma[i] = m1; maa[i] = m2;
            mx = fmax(mx, m2); mn = fmin(mn, m2);
        }
        for (int o = 32; o > 0; o >>= 1) { mx = fmax(mx, __shfl_down(mx, o, 64)); mn = fmin(mn, __shfl_down(mn, o, 64)); }
        if ((tid & 63) == 0) { s_red[tid >> 6] = mx; s_red[4 + (tid >> 6)] = mn; }
        __syncthreads();
        mx = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
        mn = fmin(fmin(s_red[4], s_red[5]), fmin(s_red[6], s_red[7]));
        t_ceil_ma = t_ceil_ma + (mx - t_ceil_ma) * 0.05;                   // every thread carries the four trackers (:186-190)
        t_ceil_maa = t_ceil_maa + (t_ceil_ma - t_ceil_maa) * 0.05;
        t_floor_ma = t_floor_ma + (mn - t_floor_ma) * 0.05;
        t_floor_maa = t_floor_maa + (t_floor_ma - t_floor_maa) * 0.05;
        const double lo = t_floor_maa - 0.75;
        const double den = log10((t_ceil_maa + 0.25) - lo);
        const int osz = min(fr.out_size, H);
        float *out = points + (size_t)f * L;
        __syncthreads();                                                    // every thread's maa[] store is visible
        for (int i = tid; i < osz; i += kFftThreads) {
            out[2 * i] = (float)((double)i / (double)osz);
            out[2 * i + 1] = (float)(log10(maa[i] + 0.25 - lo) / den);
        }
        if (tid == 0) {
            ScopeMeta m;
            m.mode = 0; m.spectrum = 1; m.channels = fr.channels; m.input_rate = fr.input_rate; m.sample_rate = fr.sample_rate;
            m.fft_size = H; m.n_floats = 2 * osz; m.pad = 0; m.fft_floor = t_floor_maa; m.fft_ceil = t_ceil_maa;
            meta[f] = m;
        }
        __syncthreads();                                                    // the LDS arrays are reused by the next frame
    }
    if (tid == 0) { trk[0] = t_ceil_ma; trk[1] = t_ceil_maa; trk[2] = t_floor_ma; trk[3] = t_floor_maa; }
}

// ---- audio mix-down (audioCallback) ---------------------------------------------------------------------------------------
// The host walks the sources' block queues (which block is current, when the next one is taken: csdr_io.hip) and describes
// every callback buffer as a list of PIECES: a run of one source's stream that lands on a run of the buffer's interleaved stereo
// floats.  The kernel does the arithmetic in the callback's own order -- sources in binding order, v = sample * gain, out += v,
// then the whole buffer times (float)(1 / peak) when the summed peaks exceed 1 -- with explicitly rounded operations, so that the
// result is the callback's bit for bit.  Peaks are read from device memory (they come out of the audio kernel's block results).
struct MixPiece {
    const float *ring;           // the source's sample ring (device)
    uint32_t ring_mask;          // capacity - 1 (power of two), in floats
    uint32_t ring_pos;           // ring index of the sample that lands on out_begin
    int32_t out_begin, out_end;  // float range inside the buffer [0, 2 * frames)
    int32_t mono;                // 1: every sample feeds both floats of its frame
    float gain;
};
struct MixPeakRef {              // one (buffer, source) visit of one block: mixPeak = max over them of peak * gain, per source
    const float *peak;           // device float (a block's audio peak)
    float gain;
    int32_t source_slot;         // position of the source among the sources that contribute to this buffer (0, 1, ...)
};
struct MixBuffer { int32_t piece0, piece1, ref0, ref1, n_sources, pad; };

constexpr int kMixThreads = 256;
constexpr int kMixMaxSources = 1024;       // per buffer (LDS: one double each)

CSDR_KERNEL __launch_bounds__(kMixThreads) void audio_mix(const MixBuffer *__restrict__ bufs, const MixPiece *__restrict__ pieces,
                                                        const MixPeakRef *__restrict__ refs, int frames, float *__restrict__ out,
                                                        float *__restrict__ out_peak /* per buffer: the summed peak (diagnostic / PCM scale) */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *s_peak = reinterpret_cast<double *>(smem);                     // [kMixMaxSources] mixPeak of each contributing source
    float *s_inv = reinterpret_cast<float *>(s_peak + kMixMaxSources);
    const MixBuffer mb = bufs[blockIdx.x];
    const int tid = threadIdx.x;
    for (int s = tid; s < mb.n_sources; s += kMixThreads) s_peak[s] = -1.0;
    __syncthreads();
    if (tid == 0) {
        // mixPeak per source: the first visited block sets it, later ones raise it (:175, :190-193); summed in source order (:233)
        for (int r = mb.ref0; r < mb.ref1; ++r) {
            const double p = (double)rounded(*refs[r].peak * refs[r].gain);
            double &m = s_peak[refs[r].source_slot];
            m = m < 0.0 ? p : fmax(m, p);
        }
        double peak = 0.0;
        for (int s = 0; s < mb.n_sources; ++s) peak += s_peak[s] < 0.0 ? 0.0 : s_peak[s];
        *s_inv = peak > 1.0 ? (float)(1.0 / peak) : 1.0f;
        out_peak[blockIdx.x] = (float)peak;
    }
    __syncthreads();
    const float inv = *s_inv;
    float *o = out + (size_t)blockIdx.x * 2 * frames;
    for (int j = tid; j < 2 * frames; j += kMixThreads) {
        float acc = 0.f;                                                    // memset(out, 0, ...) (:96)
        for (int p = mb.piece0; p < mb.piece1; ++p) {
            const MixPiece pc = pieces[p];
            if (j < pc.out_begin || j >= pc.out_end) continue;
            const uint32_t k = pc.mono ? (uint32_t)((j >> 1) - (pc.out_begin >> 1)) : (uint32_t)(j - pc.out_begin);
            const float v = rounded(pc.ring[(pc.ring_pos + k) & pc.ring_mask] * pc.gain);        // v = sample * gain; out += v (:188-190, :215)
            acc = rounded(acc + v);
        }
        o[j] = inv != 1.0f ? acc * inv : acc;
    }
}

// append `n` floats at src to a ring at write position wpos (ring of mask + 1 floats); grid-stride.  One launch moves the batch audio
// of many demodulators: job j = blockIdx.y.
struct RingPush { const float *src; float *ring; uint32_t mask, wpos; int32_t n; int32_t n_peaks; const void *peaks_src /* BlockOut[n_peaks] */; float *peaks_dst; uint32_t peaks_mask, peaks_wpos; };
CSDR_KERNEL __launch_bounds__(256) void ring_push(const RingPush *__restrict__ jobs, int peak_stride_bytes, int peak_offset_bytes) {
    const RingPush jb = jobs[blockIdx.y];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < jb.n; i += 256 * gridDim.x) jb.ring[(jb.wpos + (uint32_t)i) & jb.mask] = jb.src[i];
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < jb.n_peaks; i += 256)
            jb.peaks_dst[(jb.peaks_wpos + (uint32_t)i) & jb.peaks_mask] =
                *reinterpret_cast<const float *>(reinterpret_cast<const char *>(jb.peaks_src) + (size_t)i * peak_stride_bytes + peak_offset_bytes);
}

// ---- float -> 16-bit PCM with the WAV writer's anti-clipping scale (:136): int(x * (peak < 1 ? 32767 : 32767 / peak)), low 16 bits.
// job = one AudioThreadInput (a block): grid = (chunks, jobs)
CSDR_KERNEL __launch_bounds__(256) void pcm16_convert(const PcmJob *__restrict__ jobs) {
    const PcmJob jb = jobs[blockIdx.y];
    const float pk = *jb.peak;
    const float scale = pk < 1.0f ? 32767.0f : __fdiv_rn(32767.0f, pk);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < jb.n; i += 256 * gridDim.x)
        jb.dst[i] = (int16_t)(int)rounded(jb.src[i] * scale);              // float -> int truncates toward zero; the writer keeps the low 16 bits
}

// ---- ingest: host block -> HBM with I and Q exchanged on the way (the reference swaps while it copies the block together).
// `src` is page-locked host memory mapped into the device's address space: the kernel IS the transfer over the link.
CSDR_KERNEL __launch_bounds__(256) void ingest_swap(const float2 *src, float2 *dst /* may be src: exchange in place */, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)256 * gridDim.x) {
        const float2 v = src[i];
        dst[i] = make_float2(v.y, v.x);
    }
}

}  // namespace csdr
