// csdr_objects.hpp -- the object layouts behind the opaque handles of include/csdr_hip.h that more than one translation unit looks into
// (the demodulator bank reads the channelizer's output rotation; the audio / scope edges read the bank's slots).
#pragma once
#include <map>
#include <vector>

#include "common.hpp"
#include "design.hpp"
#include "kernels_post.hpp"
#include "kernels_chanfft.hpp"
#include "kernels_demod.hpp"

using namespace csdr;      // (this header is only included by the library's own translation units, all of which do the same)

// =================================================================================================== SDRPostThread
struct csdr_post {
    csdr_ctx *ctx = nullptr;
    bool configured = false;
    int mode = CSDR_POST_SINGLE, M = 1;
    int64_t sample_rate = 0, chan_bw = 0, chan_rate = 0, frequency = 0;
    int hop = 1;                             // input samples per output sample of a channel: M, M / 2 (PFBCH2) or 1 (single)
    int max_block_len = 0, max_blocks = 0;
    int64_t chan_stride = 0;                 // samples per channel row in `out` (even: rows stay 16-byte aligned)
    int n_blocks = 0, block_len = 0;         // of the last execute
    std::vector<int64_t> centers;            // chanCenters[M + 1]
    std::vector<int> active_host;            // sorted list of produced channels
    std::vector<int> row_order;              // csdr_post_set_row_order: channel of output row i (empty: row = channel).  A post with packed rows is a
                                             // time-slab PRODUCER: its buffer is the all-to-all's send buffer; no bank reads it, no DC blocker runs on it
    bool active_dirty = true;
    ChanGeom geom{};
    bool use_fft = false;                    // critically sampled, M = 2^a 3^b 5^c 7^d 11^e 13^f: chan_analyze_fft (kernels_chanfft.hpp)
    ChanFftGeom fgeom{};
    DevBuf<int> perm;                        // chan_analyze_fft: position after the last pass -> channel
    // `out` holds kPostBufs batches in rotation: the channelizer fills the next one while the demodulators still read
    // the previous (the reference hands ReBuffer blocks through a queue, SDRPostThread.cpp:341-396)
    static constexpr int kPostBufs = 3, kMaxConsumers = 4;
    int cur = 0;                             // buffer of the last execute
    uint64_t seq = 0;
    hipEvent_t ev_ready[kPostBufs] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_consumed[kPostBufs][kMaxConsumers] = {};
    int n_consumed[kPostBufs] = {0, 0, 0};
    DevBuf<float2> out, hist0, hist1, stage_in, twA, twB, twM, post2;
    DevBuf<float> taps;
    DevBuf<int> active;                      // [M] flags
    DevBuf<d2> dc_state, tile_end;           // dc_state[2]: ping-pong carried state
    int hist_parity = 0, dc_parity = 0;
    double dc_c = 0.0;                       // feedback coefficient of the DC blocker recurrence
    bool raw = false;                        // internal (zoomed spectrum view): SINGLE mode hands the input on unfiltered
    bool dc_enabled = true;                  // csdr_post_set_dc_blocker: a time-slab producer leaves channel 0 to the rank that owns it
    size_t out_off = 0;                      // samples between the allocation's base and the first output buffer (0 unless the measurement build moves it: CSDR_OUT_OFFSET_KB)
    int import_k = -1;                       // buffer being assembled by csdr_post_import_begin .. commit
    bool rotate = false;                     // the output is read from a stream that is no lane (csdr_post_exchange_rows_begin's transfers): the buffers rotate whatever the stream folding
    std::map<std::vector<int>, int *> rowlists;   // device copies of the channel lists export / import calls name (a handful, reused every batch)
};
static inline float2 *post_buf(const csdr_post *p, int k) { return p->out.p + p->out_off + (size_t)k * p->chan_stride * p->M; }

// =================================================================================================== demodulator bank
namespace csdr {
struct SlotHost {
    bool configured = false, active = false;
    csdr_demod_params prm{};
    design::MsresampPlan iq, au;
    int64_t chan_rate = 0;
    // integer state mirrored on the host (closed-form bookkeeping)
    uint32_t theta = 0, dtheta = 0, buf_idx = 0, phase = 0, aphase = 0, abuf = 0, ssb_theta = 0, cw_dtheta = 0;
    uint32_t tab_rot = 0;                    // layout of the oscillator table in the front-end's LDS for this dtheta (fe_table_rotation)
    long long shift_frequency = 0;
    bool shift_valid = false;
    int hist_parity = 0, last_parity = 0;
    int prev_J = 0;                          // resampled-IQ samples of the previous executed batch
    // routing of the last batch: getChannelAt is a scan over M + 1 centres (1 M comparisons per batch with 1024 demodulators behind M = 1024);
    // the centres are a function of (post frequency, sample rate, M): same key, same channel
    int64_t route_freq = 0, route_post_freq = 0, route_post_rate = 0;
    int route_M = 0, route_ch = -2;          // -2: nothing cached
    int warm = 0;                            // cascade span in input samples (+ one output period)
    bool fms_sos_set = false;                // csdr_bank_set_fms_pilot: caller-supplied pilot band-pass sections
    float fms_b[15] = {0}, fms_a[15] = {0};
    void *slab = nullptr;
    SlotCfg cfg{};
    // results of the last execute
    std::vector<csdr_block_result> results;
    int last_J = 0, last_A = 0;
};
constexpr int kStageRing = 4;                // pinned staging sets for the per-batch uploads
}  // namespace csdr

struct csdr_bank {
    csdr_ctx *ctx = nullptr;
    int max_demods = 0, max_blocks = 0;
    std::vector<SlotHost> slots;
    DevBuf<SlotCfg> cfgs;
    // per-batch device tables, two copies: the front-end of batch i+1 uploads its set while the audio kernels of batch i
    // still read theirs
    // ONE table per copy, uploaded by one transfer per batch (three transfers were three stream round trips: 75 us of a 0.65 ms C4 batch):
    //   SlotDyn [max_demods] | int [3][max_demods]: all running slots | running auto-gain slots | grouped by front-end kernel | BlockPlan [max_demods][blocks + 1]
    DevBuf<char> tables;                     // [2][table_bytes]
    size_t table_bytes = 0, off_lists = 0, off_plans = 0;
    SlotDyn *dyns_of(char *t) const { return reinterpret_cast<SlotDyn *>(t); }
    int *lists_of(char *t) const { return reinterpret_cast<int *>(t + off_lists); }
    BlockPlan *plans_of(char *t) const { return reinterpret_cast<BlockPlan *>(t + off_plans); }
    uint64_t seq = 0;
    hipEvent_t ev_fe_done[2] = {nullptr, nullptr}, ev_audio_done[2] = {nullptr, nullptr};
    bool audio_pending[2] = {false, false};
    hipEvent_t ev_tables_read[2] = {nullptr, nullptr};      // behind the audio kernels of a batch: the device tables of its parity may be refetched (by the channelizer's stream)
    bool tables_read_pending[2] = {false, false};
    DevBuf<float> arms;
    DevBuf<ModemConsts> mconsts;
    PinBuf<char> tables_h[kStageRing];
    hipEvent_t stage_ev[kStageRing] = {nullptr, nullptr, nullptr, nullptr};
    bool stage_used[kStageRing] = {false, false, false, false};
    int stage_next = 0;
    PinBuf<BlockOut> bout_h;
    std::map<uint32_t, int> arm_index;       // key: bit pattern of rate_arb
    std::vector<float> arms_host;
    int n_run = 0, last_nb = 0;
    size_t lds_attr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // the front-end launch lists of the last batch and what they were built from (slot, channel, cascade class per running slot): csdr_bank_execute
    std::vector<int> grp_key, grp_key_scratch, grp_list;
    int grp_off[8] = {0}, grp_n[8] = {0}, grp_rows[8] = {0};
    bool grp_merged56 = false;
    struct SnapBytes { char b[64]; };
    std::vector<SnapBytes> snap;             // host-state snapshot of a batch being planned (a rejected batch puts it back)
    DevBuf<int16_t> pcm;                     // csdr_bank_fetch_pcm16: the converted audio of one slot
    DevBuf<PcmJob> pcm_jobs;
};

// internal modem id: NCO + msresamp only, no modem / audio stage (the zoomed spectrum view's shift + resample, SpectrumVisualProcessor.cpp:306-379)
#define CSDR_MODEM_FRONTEND_ONLY 100
// csdr_bank_configure_slot without the public entry point's modem-range check (csdr_bank.hip; the zoomed view configures a front-end-only slot)
int bank_configure_slot(csdr_bank *b, int slot, const csdr_demod_params *prm, const csdr_post *post);
static inline bool is_fe_only(int modem) { return modem == CSDR_MODEM_FRONTEND_ONLY || modem == CSDR_MODEM_HOST; }
