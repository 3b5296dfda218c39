/*
 * csdr_hip.h -- C ABI of the MI355X-native streaming-IQ DSP hot path of CubicSDR.
 *
 * The reference (cjcliffe/CubicSDR v0.2.8) has no FFI: its hot path is C++ classes calling liquid-dsp
 * (SURVEY.md section 8b).  This header is the boundary a maintainer binds from those classes; every entry
 * point names the reference code it replaces (file:line relative to the reference tree).  Plain pointers
 * and sizes only; every call returns 0 on success or a negative CSDR_E* code and never throws.  A handle
 * must be used from one thread at a time (same rule as the reference objects: one owning IOThread each); different
 * handles of one context may be used from different threads at once -- the reference's thread cut: SDRPostThread with
 * its demodulators (csdr_post + csdr_bank: a call that takes two handles uses both) beside the spectrum thread (csdr_spec).
 *
 * Sample format everywhere: interleaved complex float32 {re, im} (liquid_float_complex, liquid.h:149-157).
 * "dev" pointers are HIP device pointers resident in HBM; "host" pointers are ordinary host memory.
 *
 * Batching: the reference handles one SDRThreadIQData block per loop turn (60 blocks/s,
 * SoapySDRThread.cpp:12,668-674).  Every *_execute call here takes `n_blocks` consecutive blocks of
 * `block_len` samples and produces exactly the per-block results the reference would produce for that
 * sequence (per-block output counts included); n_blocks = 1 is the real-time case.
 */
#ifndef CSDR_HIP_H
#define CSDR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSDR_OK            0
#define CSDR_EINVAL       -1   /* bad argument / unsupported parameter */
#define CSDR_ENOMEM       -2   /* host or device allocation failed */
#define CSDR_EHIP         -3   /* a HIP runtime call failed (see csdr_last_error) */
#define CSDR_ESTATE       -4   /* object not configured / wrong call order */
#define CSDR_ERANGE       -5   /* capacity exceeded (block too long, too many demods, ...) */
#define CSDR_EUNSUPPORTED -6   /* valid in the reference, not built yet (see DESIGN.md out-of-scope) */

typedef struct csdr_ctx   csdr_ctx;    /* device + stream */
typedef struct csdr_post  csdr_post;   /* SDRPostThread's arithmetic: DC blocker | firpfbch channelizer */
typedef struct csdr_bank  csdr_bank;   /* N x {DemodulatorPreThread + DemodulatorThread + Modem} arithmetic */
typedef struct csdr_spec  csdr_spec;   /* SpectrumVisualProcessor's arithmetic */
typedef struct csdr_scope csdr_scope;  /* ScopeVisualProcessor's arithmetic (audio scope + audio spectrum) */
typedef struct csdr_mix   csdr_mix;    /* AudioThread's mixing callback, PCM conversion */
typedef struct csdr_ingest csdr_ingest; /* page-locked block ring -> HBM, one transfer per block */
typedef struct csdr_comm  csdr_comm;   /* one IQ stream over the GPUs of a node: RCCL over xGMI */

/* ------------------------------------------------------------------ context */
int         csdr_abi_version(void);
const char *csdr_strerror(int code);
const char *csdr_last_error(void);                 /* thread-local detail string of the last failure */

/* Streams.  The reference runs one IOThread per stage (SDRPostThread, DemodulatorPreThread/DemodulatorThread per
 * demodulator, SpectrumVisualDataThread) joined by queues, so block i+1 is channelized while block i is demodulated.
 * The ctx mirrors that with one internal HIP stream per stage (channelizer, demodulator front-end, modem + audio,
 * spectrum FFT, spectrum averaging + display); events order the hand-offs and the buffer rotations, so consecutive
 * *_execute / *_process calls overlap on the device and every call returns as soon as its work is enqueued.
 * `hip_stream` is the BOUNDARY stream: an existing hipStream_t (e.g. PyTorch's current stream) on which the caller
 * produces device-resident IQ -- every execute first waits for what is enqueued there -- or NULL for a private one.
 * csdr_ctx_join makes the boundary stream wait for everything enqueued so far (for consumers chained on it);
 * csdr_ctx_synchronize blocks the host until all of it is done.  The fetch / read calls synchronise what they need.
 * The device's NULL (default) stream has the handle 0 and cannot be told from "no stream" by its value: pass CSDR_STREAM_NULL to
 * make it the boundary stream (e.g. PyTorch's default stream, whose raw handle is 0). */
#define CSDR_STREAM_NULL ((void *)(intptr_t)-1)
int  csdr_ctx_create(int device, void *hip_stream, csdr_ctx **out);
int  csdr_ctx_owns_stream(const csdr_ctx *ctx);    /* 1: the boundary stream is private (created by csdr_ctx_create), 0: the caller's */
void csdr_ctx_destroy(csdr_ctx *ctx);
int  csdr_ctx_synchronize(csdr_ctx *ctx);
int  csdr_ctx_join(csdr_ctx *ctx);
void *csdr_ctx_stream(csdr_ctx *ctx);              /* the boundary hipStream_t */
/* HIP-event timer (bench.py roofline leg): start / stop marks on the boundary stream, joined with every internal
 * stream on both sides, so the bracket covers exactly the work enqueued between the two calls; returns milliseconds. */
int  csdr_ctx_timer_start(csdr_ctx *ctx);
int  csdr_ctx_timer_stop(csdr_ctx *ctx, float *ms);
/* Optional per-kernel profile: while enabled (on = 1) every kernel launch of this ctx -- or, with on = P > 1, every P-th
 * launch of each kernel -- is bracketed by HIP events on the stream it is launched on; fetch returns the accumulated
 * device time and the number of bracketed launches of kernel `id`. */
int  csdr_ctx_profile_enable(csdr_ctx *ctx, int on);
int  csdr_ctx_profile_num_kernels(void);
const char *csdr_ctx_profile_kernel_name(int id);
int  csdr_ctx_profile_fetch(csdr_ctx *ctx, int id, double *total_ms, int64_t *launches);
/* ALL launches of kernel `id` since the profile was enabled, bracketed or not (launches per batch = this / batches) */
int  csdr_ctx_profile_launches(csdr_ctx *ctx, int id, int64_t *launches);
/* shortest / longest bracketed launch of kernel `id` since the profile was enabled (the spread behind csdr_ctx_profile_fetch's mean: a kernel
   whose duration moves with the clocks or the placement of its buffers shows it here); both 0 when nothing was bracketed */
int  csdr_ctx_profile_range(csdr_ctx *ctx, int id, double *min_ms, double *max_ms);
/* raw device memory for callers without a GPU array library (tests written in C/C++) */
int  csdr_dev_alloc(csdr_ctx *ctx, uint64_t bytes, void **dev);
int  csdr_dev_free(csdr_ctx *ctx, void *dev);
int  csdr_dev_upload(csdr_ctx *ctx, void *dev, const void *host, uint64_t bytes);
int  csdr_dev_download(csdr_ctx *ctx, void *host, const void *dev, uint64_t bytes);
/* Page-lock a caller-owned host buffer (e.g. the pooled SDRThreadIQData blocks, SoapySDRThread.cpp:221-225) so that the
 * host-to-device copies of csdr_post_execute / csdr_spec_process read it by DMA; unregister before freeing it. */
int  csdr_host_register(csdr_ctx *ctx, void *host, uint64_t bytes);
int  csdr_host_unregister(csdr_ctx *ctx, void *host);

/* ------------------------------------------------------------------ SDRPostThread (src/sdr/SDRPostThread.cpp)
 * replaces: iirfilt_crcf_create_dc_blocker :29, runSingleCH :248-299 (iirfilt_crcf_execute_block :284),
 * initPFBCH :401-414 (firpfbch_crcf_create_kaiser(ANALYZER,M,4,60) :406, chanBw = sampleRate/numChannels :408),
 * runPFBCH :416-455 (firpfbch_crcf_analyzer_execute :449-451), de-interleave + channel-0 DC block :364-382,
 * updateChannels :116-124, getChannelAt :128-139. */
#define CSDR_POST_SINGLE 0   /* numChannels == 1 : DC-blocked full-rate stream is "channel 0" */
#define CSDR_POST_PFBCH  1   /* SDRPostPFBCH  (SDRPostThread.h:9-12), critically sampled analyzer */
#define CSDR_POST_PFBCH2 2   /* SDRPostPFBCH2: initPFBCH2 :458-470 (firpfbch2_crcf_create_kaiser(ANALYZER,M,4,60) :463),
                              * runPFBCH2 :472-512 (firpfbch2_crcf_execute per M/2 inputs :505-507): every channel comes
                              * out at 2*chanBw (runDemodChannels(chanBw * 2) :510), channel centres are unchanged */

int  csdr_post_create(csdr_ctx *ctx, csdr_post **out);
void csdr_post_destroy(csdr_post *post);
/* (re)build for a sample rate / channel count; resets filter state like initPFBCH(). max_* size the HBM buffers. */
int  csdr_post_configure(csdr_post *post, int64_t sample_rate, int num_channels, int mode,
                         int max_block_len, int max_blocks);
/* Process n_blocks x block_len input samples (block_len % num_channels == 0).  `iq` is device memory when
 * iq_is_dev != 0, else host memory that is copied to a device staging buffer with hipMemcpyAsync on the stage's stream: from
 * pageable memory the runtime stages the copy before the call returns; from page-locked memory (csdr_host_register) it is a
 * DMA that may still be in flight, so the buffer must stay unchanged until the next synchronising call on this object
 * (csdr_post_read_channel, csdr_bank_fetch_*, csdr_ctx_synchronize).  Output stays in HBM, channel-major. */
int  csdr_post_execute(csdr_post *post, const float *iq, int iq_is_dev, int n_blocks, int block_len,
                       int64_t frequency);
/* Optional: produce only these channels (the reference skips channels without consumers, :336-339).  NULL = all.
 * The wrap channel index M is accepted as an alias of M/2. */
int  csdr_post_set_active_channels(csdr_post *post, const int *channels, int n);
int64_t csdr_post_channel_bandwidth(const csdr_post *post);                  /* chanBw (:408) */
int64_t csdr_post_channel_rate(const csdr_post *post);                       /* sampleRate stamped on the channel data (:344): chanBw, 2*chanBw for PFBCH2 */
int     csdr_post_num_channels(const csdr_post *post);
const char *csdr_post_kernel_name(const csdr_post *post);                    /* diagnostic: the kernel runPFBCH's transform (:449-451) maps to for this channel count: "dc_blocker" (single channel), "chan_analyze_p2" (M = 2 * prime <= 122; firpfbch2 with M / 2 odd, 38 <= M <= 126), "chan_analyze_fft" (small factors, one prime factor 29 .. 509; firpfbch2 with M % 4 == 0), "chan_analyze" (any other even M) */
int64_t csdr_post_channel_center(const csdr_post *post, int i);              /* chanCenters[i], i in [0, M] (:116-124) */
int     csdr_post_channel_at(const csdr_post *post, int64_t frequency);      /* getChannelAt (:128-139) */
/* copy one channel's samples of the last execute to the host (tests / demod-visual tap); ch == M is the wrap
 * channel (alias of M/2, :359-361).  *n receives the number of complex samples written. */
int  csdr_post_read_channel(csdr_post *post, int ch, float *host_out, int cap_samples, int *n);
/* Time-slab sharding of ONE stream over several GPUs (SURVEY 8e option 2: each rank channelizes ITS blocks of a batch for all
 * channels; an all-to-all gives every rank the full time series of the channels its demodulators sit on).  No reference
 * counterpart: SDRPostThread runs on one thread.  Producer side: csdr_post_set_history (the csdr_post_history_length() input samples
 * in front of the slab, device pointer; fewer = zeros in front), csdr_post_set_dc_blocker(0) (channel 0's DC blocker, SDRPostThread.cpp:375,
 * is a recurrence over the whole stream: the owner of channel 0 runs it), csdr_post_execute, csdr_post_export_rows (rows of the listed
 * channels of the last execute -> dst_dev[i * dst_stride + frame], complex float).  Owner side, on a second post object configured alike:
 * csdr_post_import_begin, csdr_post_import_rows per peer (its frames at frame0), csdr_post_import_commit; then csdr_bank_execute. */
int  csdr_post_history_length(const csdr_post *post);
/* producer only: store the rows of the listed channels one after the other in THIS order (row i = channels[i]; other channels are not
 * produced; n = 0: back to "row = channel").  With the channels grouped by owning rank the output buffer IS the all-to-all's send buffer:
 * csdr_post_exchange_rows then sends it as it stands (no export copy).  Such a post cannot feed a bank or run the DC blocker. */
int  csdr_post_set_row_order(csdr_post *post, const int *channels, int n);
int  csdr_post_set_history(csdr_post *post, const float *dev_tail, int64_t n_samples);
int  csdr_post_set_dc_blocker(csdr_post *post, int enabled);
int  csdr_post_export_rows(csdr_post *post, const int *channels, int n, float *dst_dev, int64_t dst_stride);
int  csdr_post_import_begin(csdr_post *post, int n_blocks, int block_len, int64_t frequency);
int  csdr_post_import_rows(csdr_post *post, const int *channels, int n, const float *src_dev, int64_t src_stride, int64_t frame0, int64_t n_frames);
int  csdr_post_import_commit(csdr_post *post);

/* ------------------------------------------------------------------ demodulator bank
 * One slot == one DemodulatorInstance's DSP state: NCO shift + msresamp_crcf decimator
 * (src/demod/DemodulatorPreThread.cpp:154-209, built by DemodulatorWorkerThread.cpp:63-101), the modem
 * (src/modules/modem/analog/Modem{NBFM,FM,AM,USB,LSB}.cpp ::demodulate), ModemAnalog::buildAudioOutput
 * (ModemAnalog.cpp:67-93) and the level / peak measurements of DemodulatorThread::run
 * (DemodulatorThread.cpp:142-152,223-233). */
#define CSDR_MODEM_NBFM 0   /* ModemNBFM.cpp:26-39  freqdem kf=0.5, no auto-gain */
#define CSDR_MODEM_FM   1   /* ModemFM.cpp:26-39    same arithmetic, 200 kHz default bandwidth */
#define CSDR_MODEM_AM   2   /* ModemAM.cpp:29-50    |x| -> 51-tap DC notch, auto-gain */
#define CSDR_MODEM_USB  3   /* ModemUSB.cpp:43-64   fs/4 shift, 6th-order Butterworth, Hilbert, upper sideband */
#define CSDR_MODEM_LSB  4   /* ModemLSB.cpp         mirror of USB, lower sideband */
#define CSDR_MODEM_CW   6   /* ModemCW.cpp:155-209  msresamp_cccf interpolation to the audio rate, 650 Hz beep oscillator, c2r Hilbert
                             * (upper sideband), auto-gain in dB; bandwidth floor 500 Hz (:100-104) */
#define CSDR_MODEM_DSB  7   /* ModemDSB.cpp:38-53   ampmodem(0.5, DSB, suppressed carrier): Costas loop around the table oscillator
                             * (a per-sample feedback loop: one thread per demodulator walks the batch), auto-gain */
#define CSDR_MODEM_IQ   5   /* ModemIQ.cpp:41-61    stereo pass-through of the resampled IQ (L = imag, R = real); the
                             * bandwidth is forced to the audio rate (checkSampleRate :31-33); 2 floats per IQ sample */

#define CSDR_MODEM_FMS  8   /* ModemFMStereo.cpp:178-287  FM stereo: pilot band-pass + PLL, L-R down-mix, two audio resamplers, de-emphasis */
#define CSDR_MODEM_HOST 9   /* a modem registered through Modem::addModemFactory whose demodulate() is host code (Modem.h:127-166): the slot runs
                             * DemodulatorPreThread's arithmetic only (NCO shift + msresamp_crcf to `bandwidth`); the block's resampled IQ is
                             * fetched with csdr_bank_fetch_iq and handed to the plug-in's demodulate(kit, iq, audioOut) on the host.
                             * n_audio / level / peak of the block results stay 0: they are the plug-in's to produce. */

typedef struct csdr_demod_params {
    int32_t modem;             /* CSDR_MODEM_* (DemodulatorInstance::setDemodulatorType) */
    int32_t bandwidth;         /* Hz, modem input rate after checkSampleRate() (setBandwidth) */
    int32_t audio_sample_rate; /* Hz (setAudioSampleRate; 48000 in the reference, DemodulatorInstance.cpp:345) */
    int32_t modem_arg;         /* FM stereo: de-emphasis in microseconds ("demph", ModemFMStereo.cpp:42-81; 0 = the default 75, < 0 = none); else 0 */
    int64_t frequency;         /* Hz, demodulator centre (setFrequency) */
} csdr_demod_params;

/* per (slot, block) results, fetched after csdr_bank_execute */
typedef struct csdr_block_result {
    int32_t  n_iq;          /* samples written by msresamp_crcf_execute for this block (numWritten, :209) */
    int32_t  n_audio;       /* numAudioWritten (ModemAnalog.cpp:88) */
    int32_t  audio_offset;  /* offset of this block's audio inside the slot's audio buffer of this execute */
    int32_t  skipped;       /* 1 when the block was skipped by the |shift| > 0.75*rate rule (:161-165) */
    double   level_accum;   /* sum of |x| (IQ for NBFM/FM, audio for AM/USB/LSB: useSignalOutput) */
    int32_t  level_count;   /* number of terms in level_accum */
    float    audio_peak;    /* max |audio| (DemodulatorThread.cpp:223-233) */
    uint32_t nco_theta;     /* NCO phase word after the block (bit-exact item) */
    uint32_t resamp_phase;  /* arbitrary-resampler 24-bit phase after the block (bit-exact item) */
    uint32_t buffer_index;  /* msresamp half-band input buffer fill after the block (bit-exact item) */
    uint32_t reserved;
} csdr_block_result;

int  csdr_bank_create(csdr_ctx *ctx, int max_demods, int max_blocks, csdr_bank **out);
void csdr_bank_destroy(csdr_bank *bank);
/* (Re)build slot `slot` (MAKE_DEMOD / BUILD_FILTERS of DemodulatorWorkerThread.cpp:40-101): designs the filters
 * on the host for the CURRENT channel rate of `post` and resets the slot's state, like the worker thread
 * handing over a fresh msresamp/modem/kit.  Changing only `frequency` later: csdr_bank_set_frequency. */
int  csdr_bank_configure_slot(csdr_bank *bank, int slot, const csdr_demod_params *p, const csdr_post *post);
int  csdr_bank_set_frequency(csdr_bank *bank, int slot, int64_t frequency);
int  csdr_bank_set_active(csdr_bank *bank, int slot, int active);
/* Route every active slot to its channel of `post`'s last execute (runDemodChannels :303-398) and run
 * NCO + decimator + modem + audio resampler for all of them, all blocks, in a handful of launches. */
int  csdr_bank_execute(csdr_bank *bank, const csdr_post *post);
/* results of the last execute (blocks in order); synchronises the stream */
int  csdr_bank_fetch_results(csdr_bank *bank, int slot, csdr_block_result *out, int cap_blocks, int *n_blocks);
int  csdr_bank_fetch_audio(csdr_bank *bank, int slot, float *host_out, int cap_samples, int *n);
int  csdr_bank_fetch_iq(csdr_bank *bank, int slot, float *host_out, int cap_samples, int *n);   /* resampled IQ */
/* ModemAnalog::getDemodOutputData() of the LAST block of the last execute (ModemAnalog.cpp:95-97): the gain-scaled demodulator
 * output in front of the audio resampler, at most DEMOD_VIS_SIZE = 2048 samples (DemodulatorThread.h:15) -- what the scope tap
 * hands to the audio scope when the audio is decimated (DemodulatorThread.cpp:293-305).  *n = 0 for the I/Q and CW modems. */
int  csdr_bank_fetch_demod_output(csdr_bank *bank, int slot, float *host_out, int cap_samples, int *n);
/* FM stereo (CSDR_MODEM_FMS).  The 19 kHz pilot band-pass is iirfilt_crcf_create_prototype(CHEBY2, BANDPASS, SOS, 5, 19500/fs, 19000/fs, 1, 60)
 * (ModemFMStereo.cpp:128-139): csdr_design_fms_pilot returns the five sections the library designs for a modem input rate
 * (b15 / a15: three taps per section, execution order); csdr_bank_set_fms_pilot replaces a slot's sections (NULL: back to the design) --
 * e.g. with the host liquid's own liquid_iirdes output, whose last-place roundings depend on its libm;
 * csdr_bank_fetch_fms_stage returns intermediates of the last batch for stage-by-stage checks (which = 0: pilot oscillator phase
 * words, uint32 per resampled-IQ sample; 1: the stereo-difference stream before its audio resampler, float). */
int  csdr_design_fms_pilot(int64_t sample_rate, float *b15, float *a15);
int  csdr_bank_set_fms_pilot(csdr_bank *bank, int slot, const float *b15, const float *a15);
int  csdr_bank_fetch_fms_stage(csdr_bank *bank, int slot, int which, void *host_out, int cap_samples, int *n);
/* device-side total of audio samples produced by the last execute over all slots (bench sanity) */
int  csdr_bank_total_audio(csdr_bank *bank, int64_t *n);

/* ------------------------------------------------------------------ SpectrumVisualProcessor (src/process/SpectrumVisualProcessor.cpp)
 * replaces: setup :140-178 (fft_create_plan(2*fftSize, FORWARD)), process :212-637 full-span view:
 * frame selection :387-421, fft_execute :439, magnitude + fftshift :441-452, double EMA + min/max :494-530,
 * floor/ceil EMAs :518-530, display resample + log10 scaling :532-576. */
#define CSDR_SPEC_FIRST_FRAME 0  /* reference cadence: only the first 2*fftSize samples of each block (:387-397) */
#define CSDR_SPEC_CONTIGUOUS  1  /* every sample belongs to one non-overlapping 2*fftSize frame */
#define CSDR_SPEC_LINES       2  /* every block is one input shorter than 2*fftSize (the fftSize-sample lines FFTDataDistributor
                                  * emits, FFTDataDistributor.cpp:112-131): the first one primes fftLastData, each later one is
                                  * appended to the previous FFT input shifted left by its length (:399-421) -> one frame each */

int  csdr_spec_create(csdr_ctx *ctx, csdr_spec **out);
void csdr_spec_destroy(csdr_spec *spec);
int  csdr_spec_setup(csdr_spec *spec, int fft_size, int max_frames);      /* setup(fftSize_in): a power of two up to 2^21, or any other size up to 2^20 (setFFTSize :180-190 takes any) */
int  csdr_spec_set_average_rate(csdr_spec *spec, float rate);             /* setFFTAverageRate, default 0.65 (:36) */
int  csdr_spec_set_scale_factor(csdr_spec *spec, float sf);               /* setScaleFactor, default 1 */
/* setPeakHold (:115-125): enabling starts a one-input countdown to the reset of fft_result_peak / fft_ceil_peak /
 * fft_floor_peak (:264-273), enabling again while enabled restarts it at PEAK_RESET_COUNT = 30 inputs; frames after the
 * reset carry spectrum_hold_points and are scaled by the held ceiling / floor (:506-510, :523-541). */
int  csdr_spec_set_peak_hold(csdr_spec *spec, int enabled);
int  csdr_spec_get_peak_hold(const csdr_spec *spec);
/* setHideDC (:204-209) and the frequencies its bin arithmetic uses (:578-623): setCenterFrequency, setBandwidth, and
 * iqData->frequency of the inputs that follow.  Applied to the points as they are fetched. */
int  csdr_spec_set_hide_dc(csdr_spec *spec, int enabled);
int  csdr_spec_set_center_frequency(csdr_spec *spec, int64_t center_freq);
int  csdr_spec_set_bandwidth(csdr_spec *spec, int64_t bandwidth);
int  csdr_spec_set_input_frequency(csdr_spec *spec, int64_t frequency);
/* Zoomed view: setView(bView[, centerFreq, bandwidth]) :64-72 with setCenterFrequency / setBandwidth above.  While set, every
 * csdr_spec_process call is ONE process() input (n_blocks == 1; `mode` is ignored) taken through :283-386: the sample rate
 * is halved while half of it still covers `bandwidth`, the first fftSizeInternal / ratio samples are shifted by
 * centerFreq - input frequency (nco_crcf) and resampled (msresamp_crcf_create(ratio, 60)), the averagers follow retunes and
 * zoom steps (:316-331, :454-492), and the display walks bandwidth / resampleBw bins per point (:532-560).
 * csdr_spec_set_input_rate gives iqData->sampleRate of the inputs that follow (it also stands in for the application
 * sample rate of the range test :308). */
int  csdr_spec_set_view(csdr_spec *spec, int is_view);
int  csdr_spec_get_view(const csdr_spec *spec);
int  csdr_spec_set_input_rate(csdr_spec *spec, int64_t sample_rate);
int  csdr_spec_desired_input_size(const csdr_spec *spec);                 /* getDesiredInputSize :133-137 */
/* Run the spectrum path over n_blocks x block_len samples; frames are taken per `mode`.  Every frame updates the
 * averagers in order, exactly as one process() call per frame would. */
int  csdr_spec_process(csdr_spec *spec, const float *iq, int iq_is_dev, int n_blocks, int block_len, int mode);
int  csdr_spec_frames(const csdr_spec *spec);                             /* frames produced by the last process */
/* SpectrumVisualData of frame `frame` of the last process: spectrum_points[2*fftSize] = (x, y) pairs,
 * fft_ceiling, fft_floor (SpectrumVisualProcessor.h:14-23; :626-627). */
int  csdr_spec_fetch(csdr_spec *spec, int frame, float *points_host, int cap_floats, double *fft_ceiling, double *fft_floor);
/* spectrum_hold_points[2*fftSize] of that frame; *n_floats = 0 when the frame carries none (peak hold off or not reset yet) */
int  csdr_spec_fetch_hold(csdr_spec *spec, int frame, float *hold_host, int cap_floats, int *n_floats);
/* raw forward FFT of one 2*fftSize frame (K13 alone), for parity tests against fft_execute */
int  csdr_spec_fft_only(csdr_spec *spec, const float *iq_host, float *out_host);

/* ------------------------------------------------------------------ ScopeVisualProcessor (src/process/ScopeVisualProcessor.cpp)
 * replaces: setup :24-35 (fft_create_plan(fftSize)), process :45-217 -- waveform normalisation by max(1, peak) in the modes Y / 2Y / XY
 * (:64-117), the audio spectrum: zero-padded real input (stereo: left + right) -> fft_execute :163 -> |X| in double -> the two averagers
 * (the second one sees the UPDATED first one, :176-177) -> double extrema -> floor / ceil trackers :186-190 -> log10 scaling :202-206,
 * outSize = floor(fftSize/2 * sampleRate / inputRate) :194-200.  Every frame is one AudioThreadInput; frames of a call are processed in
 * order through ONE state, exactly as consecutive process() calls. */
typedef struct csdr_scope_frame {
    const float *data;          /* n floats; host memory, or device memory when csdr_scope_process(..., data_is_dev = 1) */
    const int32_t *n_dev;       /* device frames only: NULL, or a device int that bounds n (csdr_bank_scope_frame sets it) */
    int32_t n;                  /* AudioThreadInput::data.size() */
    int32_t channels;           /* 1 | 2 */
    int32_t type;               /* AudioThreadInput::type: 0 -> SCOPE_MODE_Y, 1 -> SCOPE_MODE_2Y, 2 -> SCOPE_MODE_XY (:84, :95, :104) */
    int32_t sample_rate, input_rate;
    int32_t layout;             /* 0: data is in AudioThreadInput order.  1 / 2: data is n/2 interleaved pairs (a, b) as a stereo demodulator
                                 * wrote them; the frame the scope sees is  1: [a.. | b..] * scale   2: [b.. | a..] * scale  (the re-packing
                                 * of DemodulatorThread.cpp:269-291 done while the tap is read, in place in HBM) */
    float   scale;
} csdr_scope_frame;
typedef struct csdr_scope_info {   /* ScopeRenderData (ScopeVisualProcessor.h:10-21) without the points */
    int32_t mode, spectrum, channels, input_rate, sample_rate, fft_size, n_floats, reserved;
    double  fft_floor, fft_ceil;
} csdr_scope_info;
int  csdr_scope_create(csdr_ctx *ctx, csdr_scope **out);
void csdr_scope_destroy(csdr_scope *scope);
int  csdr_scope_setup(csdr_scope *scope, int fft_size, int max_frames, int max_samples);   /* setup(fftSize_in); fft_size <= 4096 */
int  csdr_scope_set_enabled(csdr_scope *scope, int scope_enabled, int spectrum_enabled);  /* :37-43 */
int  csdr_scope_set_max_scope_samples(csdr_scope *scope, int n);                            /* maxScopeSamples, default DEFAULT_DMOD_FFT_SIZE = 1024 (:13) */
int  csdr_scope_set_average_rate(csdr_scope *scope, float rate);                            /* fft_average_rate, default 0.65 (:10) */
int  csdr_scope_process(csdr_scope *scope, const csdr_scope_frame *frames, int n_frames, int data_is_dev);
int  csdr_scope_frames(const csdr_scope *scope);
/* item of frame `frame` of the last process: which = 0 the waveform, 1 the spectrum; info->n_floats == 0 when it was not produced */
int  csdr_scope_fetch(csdr_scope *scope, int frame, int which, float *points_host, int cap_floats, csdr_scope_info *info);
/* the audio-scope tap of a demodulator for the LAST block of the bank's last execute (DemodulatorThread.cpp:240-316), as a frame
 * whose data lies in HBM (pass it to csdr_scope_process with data_is_dev = 1); out->n == 0: nothing to show */
int  csdr_bank_scope_frame(csdr_bank *bank, int slot, csdr_scope_frame *out);

/* ------------------------------------------------------------------ audio egress
 * csdr_mix replaces the arithmetic AND the queue rules of audioCallback (src/audio/AudioThread.cpp:88-240): sources in binding order;
 * a source takes part in a callback buffer only while bound, active and with a non-empty queue (:117); its first callback only latches a
 * block (:121-129); blocks at another sample rate are discarded (:131-149); mono samples feed both output channels (:169-194), stereo
 * ones are added float by float (:196-219); per source mixPeak = max(peak * gain) over the blocks it visited, and the buffer is scaled
 * by (float)(1 / sum) when the sum exceeds 1 (:222-238).  Samples live in per-source rings in HBM; every operation is individually
 * rounded in the callback's order, so the mix is the callback's bit for bit. */
int  csdr_mix_create(csdr_ctx *ctx, int max_sources, int ring_floats, int sample_rate, csdr_mix **out);
void csdr_mix_destroy(csdr_mix *mix);
int  csdr_mix_set_source(csdr_mix *mix, int source, int bound, int active, float gain, int queue_blocks);
/* one AudioThreadInput onto a source's queue (host or device memory); returns 1 when the queue was full and the block was dropped */
int  csdr_mix_push(csdr_mix *mix, int source, const float *audio, int is_dev, int n_floats, int channels, int sample_rate, float peak);
/* every block of the bank's last execute for n (slot, source) pairs: samples and peaks are appended in HBM by one kernel */
int  csdr_mix_push_bank(csdr_mix *mix, csdr_bank *bank, const int *slots, const int *sources, int n);
int  csdr_mix_queued(const csdr_mix *mix, int source);                  /* blocks waiting in the source's queue */
/* n_buffers consecutive callbacks of `frames` stereo frames -> out_host[n_buffers * frames * 2] (NULL: keep the result on the device) */
int  csdr_mix_render(csdr_mix *mix, int frames, int n_buffers, float *out_host);
/* AudioFileWAV::writePayloadToFileStream's conversion (src/audio/AudioFileWAV.cpp:133-157): int(x * (peak < 1 ? 32767 : 32767 / peak)),
 * low 16 bits.  Of the last render (peak: one value, or each buffer's summed peak), or of a demodulator's audio of the last execute --
 * every block with its own peak, as one AudioThreadInput each. */
int  csdr_mix_fetch_pcm16(csdr_mix *mix, int16_t *out_host, int cap_samples, float peak, int per_buffer_peak, int *n);
int  csdr_bank_fetch_pcm16(csdr_bank *bank, int slot, int16_t *out_host, int cap_samples, int *n);

/* ------------------------------------------------------------------ ingest
 * The reference's reader fills pooled SDRThreadIQData blocks (SoapySDRThread.cpp:221-225) and SDRPostThread hands ONE buffer to the
 * demodulator and the visual queues (SDRPostThread.cpp:227-245).  Here the reader fills a page-locked slot in place, commit moves it
 * over the link ONCE on a transfer stream of its own (optionally exchanging I and Q on the way: the iq_swap option of
 * SoapySDRThread.cpp:258-266) and returns the device pointer every consumer reads (csdr_post_execute / csdr_spec_process with
 * iq_is_dev = 1); it stays valid until `depth - 1` further commits.  Slot k + 1 crosses the link while slot k is processed. */
int  csdr_ingest_create(csdr_ctx *ctx, int64_t max_samples, int depth, csdr_ingest **out);
void csdr_ingest_destroy(csdr_ingest *ing);
int  csdr_ingest_acquire(csdr_ingest *ing, float **host_slot);
int  csdr_ingest_commit(csdr_ingest *ing, int64_t n_samples, int iq_swap, const float **dev_iq);
/* the same single transfer for a block assembled in the caller's own memory (pooled SDRThreadIQData blocks, page-locked once with
 * csdr_host_register); waits for the previous upload first.  csdr_ingest_next_slot: the ring slot the next transfer will overwrite. */
int  csdr_ingest_upload(csdr_ingest *ing, const float *host_iq, int64_t n_samples, int iq_swap, const float **dev_iq);
int  csdr_ingest_next_slot(const csdr_ingest *ing);
int  csdr_ingest_wait(csdr_ingest *ing);             /* blocks until the last transfer has left its source buffer: call before rewriting or recycling the block just uploaded */

/* ------------------------------------------------------------------ one stream over several GPUs
 * Replaces the fan-out point SDRPostThread.cpp:389-396 (one block pushed to every demodulator's queue) when the DemodulatorInstances
 * of ONE stream are spread over the GPUs of a node: one process per GPU, each with its own csdr_ctx.  Rank 0 makes the 128-byte id
 * (csdr_comm_unique_id) and the HOST hands it to every rank (pipe, socket, MPI ...); csdr_comm_create is collective.  Every call below
 * is collective too and is enqueued on the context's boundary stream: behind all the library's earlier work on this context, and the
 * library's next work starts behind it -- no host synchronisation on the data path.  RCCL is loaded on first use (dlopen): a
 * single-GPU user never maps it.  Buffers are device memory; counts are complex samples.
 *   broadcast     the ingest rank's raw IQ batch (then csdr_post_set_active_channels + csdr_post_execute on every rank: SURVEY 8e option 1)
 *   scatter       root holds world x n_samples (rank r's part at 2 * r * n_samples floats); every rank receives its part (time slabs)
 *   all_to_all    send_samples[q] to rank q, recv_samples[p] from rank p, consecutive in the buffers: one send / receive per xGMI peer pair
 *   max           max over the ranks of a host scalar; also a barrier (returns when this rank's enqueued work is done and all ranks arrived)
 *   csdr_post_exchange_rows  one batch of the time-slab variant: export of the rows every peer owns from `producer` (which has just
 *                 executed this rank's blocks for all channels), all-to-all, import into `owner` at each peer's frame offset, commit.
 *                 channels: the ranks' channel lists one after the other (n_channels[q] entries for rank q); frame0 / frames [world]. */
#define CSDR_COMM_ID_BYTES 128
/* Errors.  Every entry point validates its arguments and allocates before the first call the peers take part in, so a refused call (CSDR_EINVAL,
 * CSDR_ENOMEM, CSDR_ESTATE) has enqueued nothing and the communicator stays usable -- provided EVERY rank is refused alike: the ranks of one
 * collective must pass consistent arguments (the same channel lists and row order in csdr_post_exchange_rows; whether the producers' rows
 * travel as they lie is decided from the producer's row order, which therefore has to be the same on every rank).  A rank that fails INSIDE a
 * collective (CSDR_EHIP after its peers may have entered the matching calls) aborts its communicators (ncclCommAbort): its connections are torn
 * down, so the peers' pending transfers end with an error instead of waiting for good -- they see it through csdr_comm_async_error (poll it where
 * the host would otherwise block on the stream), which aborts theirs too.  An aborted communicator refuses every later call (CSDR_ESTATE): destroy
 * it on every rank and make a new one.  csdr_comm_abort does the same on request (a rank that must leave for a reason of its own). */
int  csdr_comm_unique_id(char *id_out /* [CSDR_COMM_ID_BYTES] */);
int  csdr_comm_create(csdr_ctx *ctx, const char *unique_id, int rank, int world, csdr_comm **out);
void csdr_comm_destroy(csdr_comm *comm);
int  csdr_comm_rank(const csdr_comm *comm);
int  csdr_comm_world(const csdr_comm *comm);
int  csdr_comm_broadcast(csdr_comm *comm, float *iq_dev, int64_t n_samples, int root);
int  csdr_comm_scatter(csdr_comm *comm, const float *send_dev, float *recv_dev, int64_t n_samples, int root);
int  csdr_comm_all_to_all(csdr_comm *comm, const float *send_dev, const int64_t *send_samples, float *recv_dev, const int64_t *recv_samples);
typedef struct csdr_p2p_op { int32_t peer; int32_t recv; float *buf; int64_t n_samples; } csdr_p2p_op;
/* n point-to-point transfers as one group (a scatter of overlapping windows, any irregular exchange): op i sends n_samples from buf to peer
 * (recv == 0) or receives them into buf; peer == this rank is refused (a rank's own part needs no transfer) */
int  csdr_comm_p2p(csdr_comm *comm, const csdr_p2p_op *ops, int n);
int  csdr_comm_max(csdr_comm *comm, double *value);
int  csdr_comm_barrier(csdr_comm *comm);
int  csdr_post_exchange_rows(csdr_comm *comm, csdr_post *producer, csdr_post *owner, const int *channels, const int *n_channels,
                             const int64_t *frame0, const int64_t *frames, int n_blocks, int block_len, int64_t frequency);
/* The same exchange in two halves, so that the transfers of batch i run beside the channelizer of batch i + 1 (SDRPostThread.cpp:389-396 hands a
 * block to the demodulators' queues and goes straight on to the next one: the queue is the overlap there).  Host order per batch:
 *     csdr_post_execute(producer, batch i + 1);  _begin(...batch i + 1...);  _finish(owner, batch i);  csdr_bank_execute(bank, owner);
 *   _begin   enqueues the grouped sends / receives on the communicator's own transfer stream (and, where RCCL can split one off, its own
 *            communicator), behind the producer's kernels only; no lane of the library waits for them.  From its first _begin on, the producer
 *            rotates its output buffers whatever the stream folding and rewrites a buffer only behind the transfers that read it.
 *   _finish  the oldest batch begun and not finished: the owner's lane waits for that batch's transfers, imports and commits (n_blocks,
 *            block_len, frequency describe THAT batch).
 * At most two batches may be between their halves (CSDR_ESTATE beyond); csdr_comm_exchanges_pending counts them.  The owner's rows -- and the
 * audio behind them -- equal the one-call form's bit for bit. */
int  csdr_post_exchange_rows_begin(csdr_comm *comm, csdr_post *producer, const int *channels, const int *n_channels,
                                   const int64_t *frame0, const int64_t *frames);
int  csdr_post_exchange_rows_finish(csdr_comm *comm, csdr_post *owner, int n_blocks, int block_len, int64_t frequency);
int  csdr_comm_exchanges_pending(const csdr_comm *comm);
int  csdr_comm_abort(csdr_comm *comm);
int  csdr_comm_async_error(csdr_comm *comm);        /* CSDR_OK: no transfer of this communicator has failed; otherwise it has been aborted here too */

#ifdef __cplusplus
}
#endif
#endif /* CSDR_HIP_H */
