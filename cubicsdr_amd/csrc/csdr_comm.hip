// csdr_comm.hip -- implementation of include/csdr_hip.h (gfx950): csdr_comm, ONE IQ stream over the GPUs of a node (RCCL over xGMI).
//
// Replaces the fan-out point of SDRPostThread::runDemodChannels (SDRPostThread.cpp:389-396: one ReBuffer block pushed to every demodulator's
// queue) when the DemodulatorInstances of one stream are spread over several GPUs (BASELINE config 4):
//   * csdr_comm_broadcast      the ingest rank's raw IQ batch to every rank (8 B / sample; every rank then channelizes for ITS channels);
//   * csdr_comm_scatter        time slabs [history | the rank's blocks] from the ingest rank (the time-slab variant: the channelizer's work is divided too);
//   * csdr_comm_all_to_all     the rows of each owner's channels for each producer's frames: every channel sample crosses a link once;
//   * csdr_post_exchange_rows  export -> all-to-all -> import -> commit of one batch, between a producer and an owner csdr_post.
// One process per GPU; the communicator is created from a 128-byte id that rank 0 makes and the HOST distributes (a pipe, a socket, MPI, a
// torch.distributed store: csdr_hip.h does not care).  Every collective is enqueued on the context's BOUNDARY stream, behind every lane of
// the library (csdr_ctx::join), and the lanes' next work starts behind it (lane_begin): no host synchronisation on the data path.
//
// RCCL is loaded at the first csdr_comm_unique_id / csdr_comm_create (dlopen: a single-GPU user of the library never maps it).  xGMI is
// point-to-point: the all-to-all is one grouped send / receive per peer pair, sized by the caller's channel plan.
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "csdr_objects.hpp"

using namespace csdr;

namespace {
// the few declarations of <rccl/rccl.h> this file needs (the header drags the whole HIP surface in; the ABI below is stable since NCCL 2.x)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { kNcclSuccess = 0, kNcclFloat = 7, kNcclDouble = 8, kNcclMax = 2 };
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommAbort)(ncclComm_t) = nullptr;                                   // optional (older libraries): without it a failed communicator is only marked
    int (*CommGetAsyncError)(ncclComm_t, int *) = nullptr;                     // optional
    int (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, void *) = nullptr;    // optional: a second communicator for the pipelined row exchange
};
Rccl &rccl() { static Rccl r; return r; }
int rccl_load() {
    Rccl &r = rccl();
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);          // (taken on every call: communicators are created once per process, collectives do not come here)
    if (r.lib) return CSDR_OK;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return fail(CSDR_EUNSUPPORTED, "RCCL is not available: %s", dlerror());
#define CSDR_SYM(field_, sym_) \
    if (!(*(void **)(&r.field_) = dlsym(h, sym_))) { dlclose(h); return fail(CSDR_EUNSUPPORTED, "librccl lacks %s", sym_); }
    CSDR_SYM(GetUniqueId, "ncclGetUniqueId"); CSDR_SYM(CommInitRank, "ncclCommInitRank"); CSDR_SYM(CommDestroy, "ncclCommDestroy");
    CSDR_SYM(Broadcast, "ncclBroadcast"); CSDR_SYM(AllReduce, "ncclAllReduce"); CSDR_SYM(Send, "ncclSend"); CSDR_SYM(Recv, "ncclRecv");
    CSDR_SYM(GroupStart, "ncclGroupStart"); CSDR_SYM(GroupEnd, "ncclGroupEnd"); CSDR_SYM(GetErrorString, "ncclGetErrorString");
#undef CSDR_SYM
    *(void **)(&r.CommAbort) = dlsym(h, "ncclCommAbort");
    *(void **)(&r.CommGetAsyncError) = dlsym(h, "ncclCommGetAsyncError");
    *(void **)(&r.CommSplit) = dlsym(h, "ncclCommSplit");
    r.lib = h;
    return CSDR_OK;
}
#define CSDR_RCCL_TRY(expr)                                                                                               \
    do {                                                                                                                  \
        const int e__ = (expr);                                                                                           \
        if (e__ != kNcclSuccess) return ::csdr::fail(CSDR_EHIP, "%s failed: %s", #expr, rccl().GetErrorString(e__));      \
    } while (0)
// a group of sends / receives: closed on every path out of the scope (an early error return must not leave the communicator inside an open group)
struct GroupScope {
    bool open = false;
    int start() { const int e = rccl().GroupStart(); open = e == kNcclSuccess; return e; }
    int end() { open = false; return rccl().GroupEnd(); }
    ~GroupScope() { if (open) (void)rccl().GroupEnd(); }
};
}  // namespace

// one batch of the pipelined row exchange between its two halves (csdr_post_exchange_rows_begin / _finish)
struct PendingExchange {
    int slot = 0;                            // receive (and packing) buffer of this batch
    bool direct = false;
    const float2 *send = nullptr;            // where this rank's own rows lie (the producer's buffer of that batch, or the packing buffer)
    std::vector<int> channels, n_channels;
    std::vector<int64_t> frame0, frames, soff, rcount, rstride;
};
struct csdr_comm {
    csdr_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    bool loopback = false;                   // one rank without RCCL (the host-executing test build): collectives are copies on the boundary stream
    bool aborted = false;                    // a collective failed on this rank (or csdr_comm_abort): every later call is refused
    ncclComm_t nc = nullptr;
    DevBuf<float2> send, recv;               // csdr_post_exchange_rows
    DevBuf<double> scalar;                   // barrier / max over ranks
    // pipelined row exchange: its own stream (and, where the library can split one off, its own communicator: transfers of one communicator
    // are serialised by RCCL whatever streams they sit on, and the scatter of the NEXT batch's windows runs on the boundary stream meanwhile)
    hipStream_t xstream = nullptr;
    ncclComm_t nc_x = nullptr;
    bool x_ready = false;
    DevBuf<float2> xsend[2], xrecv[2];
    hipEvent_t ev_prod = nullptr, ev_xdone[2] = {nullptr, nullptr}, ev_imported[2] = {nullptr, nullptr};
    bool imported_pending[2] = {false, false};
    uint64_t xseq = 0;
    std::vector<PendingExchange> pending;    // FIFO, at most two deep
};

extern "C" int csdr_comm_unique_id(char *id_out) {
    if (!id_out) return fail(CSDR_EINVAL, "null argument");
    memset(id_out, 0, CSDR_COMM_ID_BYTES);
#if !defined(CSDR_HIP_EMULATION)
    if (int rc = rccl_load()) return rc;
    ncclUniqueId id;
    CSDR_RCCL_TRY(rccl().GetUniqueId(&id));
    memcpy(id_out, id.internal, sizeof id.internal);
#endif
    return CSDR_OK;
}

extern "C" int csdr_comm_create(csdr_ctx *ctx, const char *unique_id, int rank, int world, csdr_comm **out) {
    DeviceScope dev__(ctx);
    if (!ctx || !out || !unique_id) return fail(CSDR_EINVAL, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(CSDR_EINVAL, "rank %d of %d", rank, world);
    std::unique_ptr<csdr_comm> m(new csdr_comm());
    m->ctx = ctx; m->rank = rank; m->world = world;
#if defined(CSDR_HIP_EMULATION)
    if (world != 1) return fail(CSDR_EUNSUPPORTED, "the host-executing test build has no RCCL: one rank only");
    m->loopback = true;
#else
    if (int rc = rccl_load()) return rc;
    ncclUniqueId id;
    memcpy(id.internal, unique_id, sizeof id.internal);
    CSDR_RCCL_TRY(rccl().CommInitRank(&m->nc, world, id, rank));      // collective: every rank of the node calls it with the same id
#endif
    if (int rc = m->scalar.reserve(2)) {
        if (m->nc) (void)rccl().CommDestroy(m->nc);
        return rc;
    }
    ctx->boundary_shared = true;              // the library itself enqueues on the boundary stream from now on: the lanes must order against it
    *out = m.release();
    return CSDR_OK;
}
extern "C" void csdr_comm_destroy(csdr_comm *m) {
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m) return;
    if (!m->aborted) (void)m->ctx->sync_all();               // (after an abort the streams may hold transfers that never end: nothing to wait for)
    if (m->xstream && !m->aborted) (void)hipStreamSynchronize(m->xstream);
    if (m->nc_x) (void)rccl().CommDestroy(m->nc_x);
    if (m->nc) (void)rccl().CommDestroy(m->nc);
    for (int k = 0; k < 2; ++k) {
        m->xsend[k].release(); m->xrecv[k].release();
        if (m->ev_xdone[k]) (void)hipEventDestroy(m->ev_xdone[k]);
        if (m->ev_imported[k]) (void)hipEventDestroy(m->ev_imported[k]);
    }
    if (m->ev_prod) (void)hipEventDestroy(m->ev_prod);
    if (m->xstream) (void)hipStreamDestroy(m->xstream);
    m->send.release(); m->recv.release(); m->scalar.release();
    delete m;
}
extern "C" int csdr_comm_rank(const csdr_comm *m) { return m ? m->rank : -1; }
extern "C" int csdr_comm_world(const csdr_comm *m) { return m ? m->world : 0; }

// A rank that fails INSIDE a collective (an RCCL or HIP error after its peers may have entered the matching calls) aborts its communicators:
// ncclCommAbort tears the rank's connections down, so the peers' pending transfers end with an error they can see (csdr_comm_async_error)
// instead of waiting for good; every later call on this object is refused.  The only way on is csdr_comm_destroy + a new communicator on every rank.
static void comm_abort(csdr_comm *m) {
    if (!m || m->aborted) return;
    m->aborted = true;
    if (rccl().CommAbort) {
        if (m->nc_x) { (void)rccl().CommAbort(m->nc_x); m->nc_x = nullptr; }
        if (m->nc) { (void)rccl().CommAbort(m->nc); m->nc = nullptr; }
    }
}
struct AbortOnFailure {                       // armed once the peers can be inside the collective; disarmed on the success path
    csdr_comm *m; bool armed = false;
    explicit AbortOnFailure(csdr_comm *m_) : m(m_) {}
    ~AbortOnFailure() { if (armed) comm_abort(m); }
};
static int comm_usable(const csdr_comm *m) { return m->aborted ? fail(CSDR_ESTATE, "the communicator was aborted after a failed collective: destroy it on every rank and create a new one") : CSDR_OK; }
// every collective: behind all the library's lanes (they may still read or write the buffers), on the boundary stream
static int comm_begin(csdr_comm *m) { if (int rc = comm_usable(m)) return rc; return m->ctx->join(); }

extern "C" int csdr_comm_abort(csdr_comm *m) {
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m) return fail(CSDR_EINVAL, "null argument");
    comm_abort(m);
    return CSDR_OK;
}
// has a transfer of this communicator failed asynchronously (a peer died or aborted)?  CSDR_OK: no; otherwise the communicator has been aborted here too
extern "C" int csdr_comm_async_error(csdr_comm *m) {
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m) return fail(CSDR_EINVAL, "null argument");
    if (int rc = comm_usable(m)) return rc;
    if (m->loopback || !rccl().CommGetAsyncError) return CSDR_OK;
    for (ncclComm_t c : {m->nc, m->nc_x}) {
        int e = kNcclSuccess;
        if (c && (rccl().CommGetAsyncError(c, &e) != kNcclSuccess || e != kNcclSuccess)) {
            comm_abort(m);
            return fail(CSDR_EHIP, "RCCL reports an asynchronous error: %s (communicator aborted)", rccl().GetErrorString(e));
        }
    }
    return CSDR_OK;
}

extern "C" int csdr_comm_broadcast(csdr_comm *m, float *iq_dev, int64_t n_samples, int root) {
    RangeScope range__("csdr_comm_broadcast");
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m || !iq_dev || n_samples < 0 || root < 0 || root >= m->world) return fail(CSDR_EINVAL, "bad argument");
    if (int rc = comm_begin(m)) return rc;
    if (m->loopback || n_samples == 0) return CSDR_OK;
    AbortOnFailure guard(m); guard.armed = true;
    CSDR_RCCL_TRY(rccl().Broadcast(iq_dev, iq_dev, (size_t)2 * (size_t)n_samples, kNcclFloat, root, m->nc, m->ctx->stream));
    guard.armed = false;
    return CSDR_OK;
}

// rank `root` holds world x n_samples samples (rank r's part at send_dev + 2 * r * n_samples floats); every rank receives its part
extern "C" int csdr_comm_scatter(csdr_comm *m, const float *send_dev, float *recv_dev, int64_t n_samples, int root) {
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m || !recv_dev || n_samples < 0 || root < 0 || root >= m->world || (m->rank == root && !send_dev)) return fail(CSDR_EINVAL, "bad argument");
    if (int rc = comm_begin(m)) return rc;
    if (n_samples == 0) return CSDR_OK;
    hipStream_t st = m->ctx->stream;
    const size_t cnt = (size_t)2 * (size_t)n_samples;
    if (m->loopback) {
        CSDR_HIP_TRY(hipMemcpyAsync(recv_dev, send_dev, cnt * sizeof(float), hipMemcpyDeviceToDevice, st));
        return CSDR_OK;
    }
    AbortOnFailure guard(m); guard.armed = true;
    GroupScope grp;
    CSDR_RCCL_TRY(grp.start());
    if (m->rank == root)
        for (int r = 0; r < m->world; ++r) CSDR_RCCL_TRY(rccl().Send(send_dev + (size_t)r * cnt, cnt, kNcclFloat, r, m->nc, st));
    CSDR_RCCL_TRY(rccl().Recv(recv_dev, cnt, kNcclFloat, root, m->nc, st));
    CSDR_RCCL_TRY(grp.end());
    guard.armed = false;
    return CSDR_OK;
}

// send_samples[q] samples to rank q (consecutive in send_dev), recv_samples[p] from rank p (consecutive in recv_dev): one grouped
// send / receive per peer pair -- xGMI is point-to-point, every pair has its own link
extern "C" int csdr_comm_all_to_all(csdr_comm *m, const float *send_dev, const int64_t *send_samples, float *recv_dev, const int64_t *recv_samples) {
    RangeScope range__("csdr_comm_all_to_all");
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m || !send_samples || !recv_samples) return fail(CSDR_EINVAL, "null argument");
    int64_t ts = 0, tr = 0;
    for (int q = 0; q < m->world; ++q) {
        if (send_samples[q] < 0 || recv_samples[q] < 0) return fail(CSDR_EINVAL, "negative count");
        ts += send_samples[q]; tr += recv_samples[q];
    }
    if ((ts && !send_dev) || (tr && !recv_dev)) return fail(CSDR_EINVAL, "null buffer");
    // every rank-local check comes BEFORE the first call the peers take part in: a rank that returned from here after the others had entered
    // the grouped transfers would leave them waiting for good
    if (send_samples[m->rank] != recv_samples[m->rank]) return fail(CSDR_EINVAL, "a rank's counts to and from itself differ");
    if (int rc = comm_begin(m)) return rc;
    hipStream_t st = m->ctx->stream;
    if (m->loopback) {
        if (ts) CSDR_HIP_TRY(hipMemcpyAsync(recv_dev, send_dev, (size_t)ts * sizeof(float2), hipMemcpyDeviceToDevice, st));
        return CSDR_OK;
    }
    AbortOnFailure guard(m); guard.armed = true;
    GroupScope grp;
    CSDR_RCCL_TRY(grp.start());
    size_t so = 0, ro = 0, self_so = 0, self_ro = 0;
    for (int q = 0; q < m->world; ++q) {
        if (q == m->rank) { self_so = so; self_ro = ro; }                 // the part that stays on this GPU is a device copy, not a transfer
        else {
            if (send_samples[q]) CSDR_RCCL_TRY(rccl().Send(send_dev + 2 * so, (size_t)2 * (size_t)send_samples[q], kNcclFloat, q, m->nc, st));
            if (recv_samples[q]) CSDR_RCCL_TRY(rccl().Recv(recv_dev + 2 * ro, (size_t)2 * (size_t)recv_samples[q], kNcclFloat, q, m->nc, st));
        }
        so += (size_t)send_samples[q]; ro += (size_t)recv_samples[q];
    }
    CSDR_RCCL_TRY(grp.end());
    if (send_samples[m->rank])
        CSDR_HIP_TRY(hipMemcpyAsync(recv_dev + 2 * self_ro, send_dev + 2 * self_so, (size_t)send_samples[m->rank] * sizeof(float2), hipMemcpyDeviceToDevice, st));
    guard.armed = false;
    return CSDR_OK;
}

// n point-to-point transfers as ONE group (what a scatter of overlapping windows, or any irregular exchange, is made of): op i sends
// n_samples from `buf` to `peer` (recv == 0) or receives them into `buf` from `peer`.  A rank's own part needs no transfer: peer == rank is refused.
static int comm_p2p_on(csdr_comm *m, ncclComm_t nc, hipStream_t st, const csdr_p2p_op *ops, int n) {
    if (m->loopback || n == 0) return CSDR_OK;
    AbortOnFailure guard(m); guard.armed = true;
    GroupScope grp;
    CSDR_RCCL_TRY(grp.start());
    for (int i = 0; i < n; ++i) {
        if (!ops[i].n_samples) continue;
        if (ops[i].recv) CSDR_RCCL_TRY(rccl().Recv(ops[i].buf, (size_t)2 * (size_t)ops[i].n_samples, kNcclFloat, ops[i].peer, nc, st));
        else CSDR_RCCL_TRY(rccl().Send(ops[i].buf, (size_t)2 * (size_t)ops[i].n_samples, kNcclFloat, ops[i].peer, nc, st));
    }
    CSDR_RCCL_TRY(grp.end());
    guard.armed = false;
    return CSDR_OK;
}
static int p2p_check(const csdr_comm *m, const csdr_p2p_op *ops, int n) {
    if (!m || n < 0 || (n && !ops)) return fail(CSDR_EINVAL, "bad argument");
    for (int i = 0; i < n; ++i)
        if (ops[i].peer < 0 || ops[i].peer >= m->world || ops[i].peer == m->rank || ops[i].n_samples < 0 || (ops[i].n_samples && !ops[i].buf)) return fail(CSDR_EINVAL, "operation %d", i);
    return CSDR_OK;
}
extern "C" int csdr_comm_p2p(csdr_comm *m, const csdr_p2p_op *ops, int n) {
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (int rc = p2p_check(m, ops, n)) return rc;
    if (int rc = comm_begin(m)) return rc;
    return comm_p2p_on(m, m->nc, m->ctx->stream, ops, n);
}

// max over the ranks of a host scalar (a timing: bench.py takes the slowest rank), which is also a barrier: returns when every rank has
// reached it and everything this rank enqueued before it has finished
extern "C" int csdr_comm_max(csdr_comm *m, double *value) {
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m || !value) return fail(CSDR_EINVAL, "null argument");
    if (int rc = comm_begin(m)) return rc;
    hipStream_t st = m->ctx->stream;
    AbortOnFailure guard(m); guard.armed = !m->loopback;
    CSDR_HIP_TRY(hipMemcpyAsync(m->scalar.p, value, sizeof(double), hipMemcpyHostToDevice, st));
    if (!m->loopback) CSDR_RCCL_TRY(rccl().AllReduce(m->scalar.p, m->scalar.p, 1, kNcclDouble, kNcclMax, m->nc, st));
    CSDR_HIP_TRY(hipMemcpyAsync(value, m->scalar.p, sizeof(double), hipMemcpyDeviceToHost, st));
    CSDR_HIP_TRY(hipStreamSynchronize(st));
    guard.armed = false;
    return CSDR_OK;
}
extern "C" int csdr_comm_barrier(csdr_comm *m) { double v = 0.0; return csdr_comm_max(m, &v); }

// One batch of the time-slab variant between this rank's producer post (it has just executed ITS blocks for all channels) and its owner
// post (the rows of ITS channels for the whole batch, read by its demodulator bank):
//   export: the rows each peer owns, packed [peer q][q's channels][this rank's frames]      (csdr_post_export_rows)
//   all-to-all over RCCL
//   import: [peer p][my channels][p's frames] into the owner's rows at p's frame offset      (csdr_post_import_begin / _rows / _commit)
// channels: the ranks' channel lists one after the other (n_channels[q] entries for rank q); frame0[p] / frames[p]: rank p's slab inside the
// batch, in frames (samples per channel).
extern "C" int csdr_post_exchange_rows(csdr_comm *m, csdr_post *producer, csdr_post *owner, const int *channels, const int *n_channels,
                                       const int64_t *frame0, const int64_t *frames, int n_blocks, int block_len, int64_t frequency) {
    RangeScope range__("csdr_post_exchange_rows");
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m || !producer || !owner || !channels || !n_channels || !frame0 || !frames) return fail(CSDR_EINVAL, "null argument");
    if (producer->ctx != m->ctx || owner->ctx != m->ctx) return fail(CSDR_EINVAL, "posts and communicator belong to different contexts");
    const int W = m->world, me = m->rank;
    std::vector<const int *> list((size_t)W);
    int64_t total_ch = 0;
    for (int q = 0; q < W; ++q) {
        if (n_channels[q] < 0 || frames[q] < 0) return fail(CSDR_EINVAL, "negative count");
        list[(size_t)q] = channels + total_ch;
        total_ch += n_channels[q];
    }
    const int64_t mine_f = frames[me];
    // Producers whose rows are packed in this very order (csdr_post_set_row_order with the concatenated lists) send their output buffer as it
    // stands: rows at the producer's pitch, the few padding samples of a row travel along (every rank's producer is configured alike, so the
    // pitch is the same everywhere).  Otherwise the rows each peer owns are packed by a copy first (csdr_post_export_rows).
    const bool direct = producer->row_order.size() == (size_t)total_ch && std::equal(producer->row_order.begin(), producer->row_order.end(), channels);
    const int64_t pitch = producer->chan_stride;
    // the part of this rank's own rows stays on the GPU: it is imported straight from where it lies, not sent to itself
    std::vector<int64_t> sc((size_t)W), rc_((size_t)W), rstride((size_t)W), soff((size_t)W);
    int64_t ts = 0, tr = 0;
    for (int q = 0; q < W; ++q) {
        const int64_t row_len = direct ? pitch : mine_f;
        soff[(size_t)q] = ts;
        sc[(size_t)q] = (int64_t)n_channels[q] * row_len;
        rstride[(size_t)q] = direct ? pitch : frames[q];
        rc_[(size_t)q] = q == me ? 0 : (int64_t)n_channels[me] * rstride[(size_t)q];
        if (direct && frames[q] > pitch) return fail(CSDR_ERANGE, "rank %d's slab of %lld frames exceeds the producers' row pitch %lld", q, (long long)frames[q], (long long)pitch);
        ts += sc[(size_t)q]; tr += rc_[(size_t)q];
    }
    if (int rc = m->recv.reserve((size_t)std::max<int64_t>(tr, 1))) return rc;
    const float2 *send = nullptr;
    if (direct) {
        // the batch the producer holds must be the slab this call hands out (csdr_post_export_rows checks the same on the packing path): a stale or
        // short batch would be sent as it stands
        if (mine_f && (producer->n_blocks <= 0 || (int64_t)producer->n_blocks * (producer->block_len / producer->hop) != mine_f))
            return fail(CSDR_ESTATE, "the producer holds %lld frames, this rank's slab has %lld", (long long)producer->n_blocks * (producer->block_len / std::max(1, producer->hop)), (long long)mine_f);
        send = post_buf(producer, producer->cur);
    } else {
        if (int rc = m->send.reserve((size_t)std::max<int64_t>(ts, 1))) return rc;
        if (mine_f)
            for (int q = 0; q < W; ++q)
                if (n_channels[q])
                    if (int rc = csdr_post_export_rows(producer, list[(size_t)q], n_channels[q], (float *)(m->send.p + soff[(size_t)q]), mine_f)) return rc;
        send = m->send.p;
    }
    {   // the transfers: one send / receive per peer pair (nothing to itself)
        std::vector<csdr_p2p_op> ops;
        int64_t ro = 0;
        for (int q = 0; q < W; ++q) {
            if (q == me) continue;
            if (sc[(size_t)q] && mine_f) ops.push_back(csdr_p2p_op{q, 0, (float *)(send + soff[(size_t)q]), sc[(size_t)q]});
            if (rc_[(size_t)q] && frames[q]) ops.push_back(csdr_p2p_op{q, 1, (float *)(m->recv.p + ro), rc_[(size_t)q]});
            ro += rc_[(size_t)q];
        }
        if (int rc = csdr_comm_p2p(m, ops.data(), (int)ops.size())) return rc;          // (joins the lanes: the producer's kernels / export copies are done)
    }
    if (int rc = csdr_post_import_begin(owner, n_blocks, block_len, frequency)) return rc;      // (lane_begin: the owner's lane starts behind the transfers)
    int64_t off = 0;
    for (int p = 0; p < W; ++p) {
        if (n_channels[me] && frames[p]) {
            const float2 *src = p == me ? send + soff[(size_t)me] : m->recv.p + off;
            if (int rc = csdr_post_import_rows(owner, list[(size_t)me], n_channels[me], (const float *)src, rstride[(size_t)p], frame0[p], frames[p])) return rc;
        }
        off += rc_[(size_t)p];
    }
    return csdr_post_import_commit(owner);
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// The same exchange in two halves, so that the transfers of batch i run BESIDE the channelizer of batch i + 1 (and the scatter of its windows):
//   host order per batch:   csdr_post_execute(producer, batch i + 1)  ->  _begin(i + 1)  ->  _finish(i)  ->  csdr_bank_execute(owner)   [batch i]
//   _begin   behind the producer's kernels only (one event), on the communicator's own transfer stream: the grouped sends / receives into one of
//            two receive buffers.  No lane waits for them and they wait for no other lane.  The producer keeps rotating its output buffers from
//            now on (whatever the stream folding) and rewrites one only after the transfers that read it.
//   _finish  the OLDEST batch begun: the owner's lane waits for that batch's transfers, imports every peer's frames (its own straight from the
//            producer's buffer of that batch) and commits; the receive buffer is handed back to the transfer stream by an event.
// At most two batches may be between their halves.  The results equal csdr_post_exchange_rows' bit for bit: the same copies in the same order.
static int exchange_plan(csdr_comm *m, csdr_post *producer, const int *channels, const int *n_channels, const int64_t *frame0, const int64_t *frames,
                         PendingExchange &px, int64_t &ts, int64_t &tr) {
    const int W = m->world, me = m->rank;
    int64_t total_ch = 0;
    for (int q = 0; q < W; ++q) {
        if (n_channels[q] < 0 || frames[q] < 0 || frame0[q] < 0) return fail(CSDR_EINVAL, "negative count");
        total_ch += n_channels[q];
    }
    px.channels.assign(channels, channels + total_ch);
    px.n_channels.assign(n_channels, n_channels + W);
    px.frame0.assign(frame0, frame0 + W); px.frames.assign(frames, frames + W);
    px.direct = producer->row_order.size() == (size_t)total_ch && std::equal(producer->row_order.begin(), producer->row_order.end(), channels);
    const int64_t pitch = producer->chan_stride, mine_f = frames[me];
    px.soff.assign((size_t)W, 0); px.rcount.assign((size_t)W, 0); px.rstride.assign((size_t)W, 0);
    ts = tr = 0;
    for (int q = 0; q < W; ++q) {
        px.soff[(size_t)q] = ts;
        ts += (int64_t)n_channels[q] * (px.direct ? pitch : mine_f);
        px.rstride[(size_t)q] = px.direct ? pitch : frames[q];
        px.rcount[(size_t)q] = q == me ? 0 : (int64_t)n_channels[me] * px.rstride[(size_t)q];
        if (px.direct && frames[q] > pitch) return fail(CSDR_ERANGE, "rank %d's slab of %lld frames exceeds the producers' row pitch %lld", q, (long long)frames[q], (long long)pitch);
        tr += px.rcount[(size_t)q];
    }
    if (mine_f && (producer->n_blocks <= 0 || (int64_t)producer->n_blocks * (producer->block_len / std::max(1, producer->hop)) != mine_f))
        return fail(CSDR_ESTATE, "the producer holds %lld frames, this rank's slab has %lld", (long long)producer->n_blocks * (producer->block_len / std::max(1, producer->hop)), (long long)mine_f);
    return CSDR_OK;
}
static int exchange_resources(csdr_comm *m) {
    if (m->x_ready) return CSDR_OK;
    CSDR_HIP_TRY(hipStreamCreateWithFlags(&m->xstream, hipStreamNonBlocking));
    CSDR_HIP_TRY(hipEventCreateWithFlags(&m->ev_prod, hipEventDisableTiming));
    for (int k = 0; k < 2; ++k) {
        CSDR_HIP_TRY(hipEventCreateWithFlags(&m->ev_xdone[k], hipEventDisableTiming));
        CSDR_HIP_TRY(hipEventCreateWithFlags(&m->ev_imported[k], hipEventDisableTiming));
    }
    // a communicator of its own for the row transfers (collective: every rank reaches its first _begin at the same place of the stream);
    // without ncclCommSplit the one communicator carries both kinds of transfer, in the order they are issued
    if (!m->loopback && m->world > 1 && rccl().CommSplit) {
        AbortOnFailure guard(m); guard.armed = true;
        CSDR_RCCL_TRY(rccl().CommSplit(m->nc, 0, m->rank, &m->nc_x, nullptr));
        guard.armed = false;
    }
    m->x_ready = true;
    return CSDR_OK;
}

extern "C" int csdr_post_exchange_rows_begin(csdr_comm *m, csdr_post *producer, const int *channels, const int *n_channels, const int64_t *frame0, const int64_t *frames) {
    RangeScope range__("csdr_post_exchange_rows_begin");
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m || !producer || !channels || !n_channels || !frame0 || !frames) return fail(CSDR_EINVAL, "null argument");
    if (producer->ctx != m->ctx) return fail(CSDR_EINVAL, "post and communicator belong to different contexts");
    if (int rc = comm_usable(m)) return rc;
    if (m->pending.size() >= 2) return fail(CSDR_ESTATE, "two row exchanges are between their halves: finish one first");
    PendingExchange px;
    int64_t ts = 0, tr = 0;
    if (int rc = exchange_plan(m, producer, channels, n_channels, frame0, frames, px, ts, tr)) return rc;
    const int W = m->world, me = m->rank, slot = (int)(m->xseq & 1);
    px.slot = slot;
    if (int rc = m->xrecv[slot].reserve((size_t)std::max<int64_t>(tr, 1))) return rc;
    if (!px.direct) if (int rc = m->xsend[slot].reserve((size_t)std::max<int64_t>(ts, 1))) return rc;
    if (int rc = exchange_resources(m)) return rc;                  // (every rank-local refusal lies above: the first collective call is in here)
    csdr_ctx *c = m->ctx;
    hipStream_t lane = c->lanes[LANE_POST];
    const int64_t mine_f = frames[me];
    if (px.direct) px.send = post_buf(producer, producer->cur);
    else {
        // packing copies on the producer's lane, into this slot's buffer: behind the transfers that last read it (two batches ago)
        if (m->xseq >= 2) CSDR_HIP_TRY(hipStreamWaitEvent(lane, m->ev_xdone[slot], 0));
        int64_t ch0 = 0;
        for (int q = 0; q < W; ++q) {
            if (mine_f && n_channels[q])
                if (int rc = csdr_post_export_rows(producer, channels + ch0, n_channels[q], (float *)(m->xsend[slot].p + px.soff[(size_t)q]), mine_f)) return rc;
            ch0 += n_channels[q];
        }
        px.send = m->xsend[slot].p;
    }
    CSDR_HIP_TRY(hipEventRecord(m->ev_prod, lane));
    CSDR_HIP_TRY(hipStreamWaitEvent(m->xstream, m->ev_prod, 0));
    if (m->imported_pending[slot]) CSDR_HIP_TRY(hipStreamWaitEvent(m->xstream, m->ev_imported[slot], 0));      // the import that last read this receive buffer
    {
        std::vector<csdr_p2p_op> ops;
        int64_t ro = 0;
        for (int q = 0; q < W; ++q) {
            if (q == me) continue;
            const int64_t send_cnt = (int64_t)n_channels[q] * (px.direct ? producer->chan_stride : mine_f);
            if (send_cnt && mine_f) ops.push_back(csdr_p2p_op{q, 0, (float *)(px.send + px.soff[(size_t)q]), send_cnt});
            if (px.rcount[(size_t)q] && frames[q]) ops.push_back(csdr_p2p_op{q, 1, (float *)(m->xrecv[slot].p + ro), px.rcount[(size_t)q]});
            ro += px.rcount[(size_t)q];
        }
        if (int rc = comm_p2p_on(m, m->nc_x ? m->nc_x : m->nc, m->xstream, ops.data(), (int)ops.size())) return rc;
    }
    CSDR_HIP_TRY(hipEventRecord(m->ev_xdone[slot], m->xstream));
    // the producer's buffer of this batch is read by the transfers (and by this rank's own import): it rotates its buffers from now on and
    // rewrites this one only behind them
    producer->rotate = true;
    if (px.direct && mine_f) {
        const int pk = producer->cur;
        if (producer->n_consumed[pk] >= csdr_post::kMaxConsumers) return fail(CSDR_ERANGE, "too many readers of one channelizer batch");
        CSDR_HIP_TRY(hipEventRecord(producer->ev_consumed[pk][producer->n_consumed[pk]++], m->xstream));
    }
    m->pending.push_back(std::move(px));
    m->xseq++;
    return CSDR_OK;
}

extern "C" int csdr_post_exchange_rows_finish(csdr_comm *m, csdr_post *owner, int n_blocks, int block_len, int64_t frequency) {
    RangeScope range__("csdr_post_exchange_rows_finish");
    DeviceScope dev__(m ? m->ctx : nullptr);
    if (!m || !owner) return fail(CSDR_EINVAL, "null argument");
    if (owner->ctx != m->ctx) return fail(CSDR_EINVAL, "post and communicator belong to different contexts");
    if (int rc = comm_usable(m)) return rc;
    if (m->pending.empty()) return fail(CSDR_ESTATE, "no row exchange has been begun");
    const PendingExchange &px = m->pending.front();
    const int W = m->world, me = m->rank, slot = px.slot;
    csdr_ctx *c = m->ctx;
    if (int rc = csdr_post_import_begin(owner, n_blocks, block_len, frequency)) return rc;
    CSDR_HIP_TRY(hipStreamWaitEvent(c->lanes[LANE_POST], m->ev_xdone[slot], 0));            // this batch's transfers
    int64_t off = 0, ch_me = 0;
    for (int q = 0; q < me; ++q) ch_me += px.n_channels[(size_t)q];
    for (int p = 0; p < W; ++p) {
        if (px.n_channels[(size_t)me] && px.frames[(size_t)p]) {
            const float2 *src = p == me ? px.send + px.soff[(size_t)me] : m->xrecv[slot].p + off;
            if (int rc = csdr_post_import_rows(owner, px.channels.data() + ch_me, px.n_channels[(size_t)me], (const float *)src, px.rstride[(size_t)p], px.frame0[(size_t)p], px.frames[(size_t)p])) return rc;
        }
        off += px.rcount[(size_t)p];
    }
    if (int rc = csdr_post_import_commit(owner)) return rc;
    CSDR_HIP_TRY(hipEventRecord(m->ev_imported[slot], c->lanes[LANE_POST]));
    m->imported_pending[slot] = true;
    m->pending.erase(m->pending.begin());
    return CSDR_OK;
}
extern "C" int csdr_comm_exchanges_pending(const csdr_comm *m) { return m ? (int)m->pending.size() : 0; }
