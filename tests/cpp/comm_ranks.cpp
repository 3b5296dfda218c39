// tests/cpp/comm_ranks.cpp -- ONE IQ stream over the GPUs of a node, in C++ against the C ABI alone (include/csdr_hip.h: csdr_comm_*,
// csdr_post_exchange_rows).  What a C++ host (the reference's SDRPostThread, one process per GPU) would do at the fan-out point
// SDRPostThread.cpp:389-396; no Python, no torch.
//
//   comm_ranks WORLD      forks WORLD processes, rank r on HIP device r (WORLD = 1 runs the same calls on one GPU)
//
// Rank 0 makes the communicator id and hands it to the others through pipes.  Every rank then checks
//   1. csdr_comm_broadcast / _scatter / _all_to_all / _max on known patterns,
//   2. the broadcast variant: rank 0's batch reaches every rank, which channelizes for ITS channels only,
//   3. the time-slab variant: scatter of [history | slab] windows, per-rank channelizer over its blocks, csdr_post_exchange_rows,
// against an unsharded csdr_post that every rank also runs on the whole batch: rows bit for bit (channel 0, whose DC blocker scans
// other tile sizes in the slab variant, to 1e-6).  Exit code 0 = every rank passed.
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/csdr_hip.h"

#define CHECK(expr)                                                                                                        \
    do {                                                                                                                   \
        const int rc__ = (expr);                                                                                           \
        if (rc__ != CSDR_OK) { std::fprintf(stderr, "[rank %d] %s -> %s: %s\n", g_rank, #expr, csdr_strerror(rc__), csdr_last_error()); return 1; } \
    } while (0)
#define REQUIRE(cond)                                                                                        \
    do {                                                                                                     \
        if (!(cond)) { std::fprintf(stderr, "[rank %d] %s:%d: %s\n", g_rank, __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

static int g_rank = 0;

static std::vector<float> pattern(size_t n_samples, unsigned seed) {       // reproducible noise-like IQ in [-0.5, 0.5)
    std::vector<float> v(2 * n_samples);
    unsigned s = seed * 2654435761u + 12345u;
    for (float &f : v) { s = s * 1664525u + 1013904223u; f = (float)((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    return v;
}

static int run_rank(int rank, int world, const char *id) {
    g_rank = rank;
    const int M = 8, frames_per_block = 96, block = M * frames_per_block, nb = 2 * world;      // every rank owns two blocks of a batch
    const int64_t fs = 4000000, center = 100000000;
    csdr_ctx *ctx = nullptr;
    CHECK(csdr_ctx_create(rank, nullptr, &ctx));
    csdr_comm *comm = nullptr;
    CHECK(csdr_comm_create(ctx, id, rank, world, &comm));
    REQUIRE(csdr_comm_rank(comm) == rank && csdr_comm_world(comm) == world);

    // ---- 1. the collectives on known patterns
    const size_t n = 4096;
    void *a = nullptr, *b = nullptr, *c = nullptr;
    CHECK(csdr_dev_alloc(ctx, n * 8 * (size_t)world, &a));
    CHECK(csdr_dev_alloc(ctx, n * 8 * (size_t)world, &b));
    CHECK(csdr_dev_alloc(ctx, n * 8 * (size_t)world, &c));
    const std::vector<float> root_data = pattern(n * (size_t)world, 7);
    if (rank == 0) CHECK(csdr_dev_upload(ctx, a, root_data.data(), root_data.size() * 4));
    CHECK(csdr_comm_broadcast(comm, (float *)a, (int64_t)n * world, 0));
    std::vector<float> got(root_data.size());
    CHECK(csdr_ctx_join(ctx)); CHECK(csdr_ctx_synchronize(ctx));
    CHECK(csdr_dev_download(ctx, got.data(), a, got.size() * 4));
    REQUIRE(got == root_data);
    CHECK(csdr_comm_scatter(comm, (const float *)a, (float *)b, (int64_t)n, 0));                  // rank r gets part r
    CHECK(csdr_ctx_synchronize(ctx));
    std::vector<float> part(2 * n);
    CHECK(csdr_dev_download(ctx, part.data(), b, part.size() * 4));
    REQUIRE(std::memcmp(part.data(), root_data.data() + 2 * n * (size_t)rank, part.size() * 4) == 0);
    {   // all-to-all: rank r sends (q + 1) * 100 samples of pattern(seed 100 r + q) to rank q
        std::vector<int64_t> sc((size_t)world), rc((size_t)world);
        std::vector<float> send;
        for (int q = 0; q < world; ++q) {
            sc[(size_t)q] = (q + 1) * 100; rc[(size_t)q] = (rank + 1) * 100;
            const std::vector<float> p = pattern((size_t)sc[(size_t)q], 100u * (unsigned)rank + (unsigned)q);
            send.insert(send.end(), p.begin(), p.end());
        }
        CHECK(csdr_dev_upload(ctx, b, send.data(), send.size() * 4));
        CHECK(csdr_comm_all_to_all(comm, (const float *)b, sc.data(), (float *)c, rc.data()));
        CHECK(csdr_ctx_synchronize(ctx));
        std::vector<float> recv((size_t)2 * (size_t)(rank + 1) * 100 * (size_t)world);
        CHECK(csdr_dev_download(ctx, recv.data(), c, recv.size() * 4));
        for (int p = 0; p < world; ++p) {
            const std::vector<float> want = pattern((size_t)(rank + 1) * 100, 100u * (unsigned)p + (unsigned)rank);
            REQUIRE(std::memcmp(recv.data() + (size_t)p * want.size(), want.data(), want.size() * 4) == 0);
        }
    }
    double v = (double)(rank + 1);
    CHECK(csdr_comm_max(comm, &v));
    REQUIRE(v == (double)world);

    // ---- 2 + 3. the two sharded drivers against an unsharded channelizer
    const int n_batches = 2;
    const std::vector<float> stream = pattern((size_t)n_batches * nb * block, 99);
    csdr_post *whole = nullptr, *shard = nullptr, *producer = nullptr, *owner = nullptr;
    CHECK(csdr_post_create(ctx, &whole)); CHECK(csdr_post_configure(whole, fs, M, CSDR_POST_PFBCH, block, nb));
    CHECK(csdr_post_create(ctx, &shard)); CHECK(csdr_post_configure(shard, fs, M, CSDR_POST_PFBCH, block, nb));
    CHECK(csdr_post_create(ctx, &producer)); CHECK(csdr_post_configure(producer, fs, M, CSDR_POST_PFBCH, block, 2));
    CHECK(csdr_post_create(ctx, &owner)); CHECK(csdr_post_configure(owner, fs, M, CSDR_POST_PFBCH, block, nb));
    // channel plan: channel k belongs to rank k % world
    std::vector<int> mine, all_ch, n_ch((size_t)world, 0);
    for (int q = 0; q < world; ++q) for (int k = q; k < M; k += world) { all_ch.push_back(k); n_ch[(size_t)q]++; if (q == rank) mine.push_back(k); }
    CHECK(csdr_post_set_active_channels(shard, mine.data(), (int)mine.size()));
    CHECK(csdr_post_set_active_channels(owner, mine.data(), (int)mine.size()));
    CHECK(csdr_post_set_dc_blocker(producer, 0));
    CHECK(csdr_post_set_row_order(producer, all_ch.data(), (int)all_ch.size()));      // rows grouped by owner: the output buffer is the send buffer
    const int H = csdr_post_history_length(producer);
    const size_t batch = (size_t)nb * block, each = (size_t)H + 2 * (size_t)block;
    void *d_batch = nullptr, *d_packed = nullptr, *d_win = nullptr;
    CHECK(csdr_dev_alloc(ctx, batch * 8, &d_batch));
    CHECK(csdr_dev_alloc(ctx, each * 8 * (size_t)world, &d_packed));
    CHECK(csdr_dev_alloc(ctx, each * 8, &d_win));
    std::vector<int64_t> frame0((size_t)world), frames((size_t)world);
    for (int q = 0; q < world; ++q) { frame0[(size_t)q] = (int64_t)q * 2 * frames_per_block; frames[(size_t)q] = 2 * frames_per_block; }
    const int nrow = nb * frames_per_block;
    std::vector<float> want((size_t)2 * nrow), row((size_t)2 * nrow);
    std::vector<std::vector<float>> kept;                                   // [batch][my channel]: the unsharded rows, for the pipelined pass below
    for (int t = 0; t < n_batches; ++t) {
        const float *host_batch = stream.data() + (size_t)t * batch * 2;
        CHECK(csdr_post_execute(whole, host_batch, 0, nb, block, center));                        // the unsharded reference (every rank runs it)
        // broadcast variant
        if (rank == 0) CHECK(csdr_dev_upload(ctx, d_batch, host_batch, batch * 8));
        CHECK(csdr_comm_broadcast(comm, (float *)d_batch, (int64_t)batch, 0));
        CHECK(csdr_post_execute(shard, (const float *)d_batch, 1, nb, block, center));
        // time-slab variant: the windows [history | two blocks] of every rank, packed by rank 0
        if (rank == 0) {
            std::vector<float> packed(each * 2 * (size_t)world, 0.0f);
            for (int q = 0; q < world; ++q) {
                const long long first = ((long long)t * (long long)batch + (long long)q * 2 * block - H) * 2;      // float index into `stream`; negative = before the stream (zeros)
                for (size_t i = 0; i < each * 2; ++i) { const long long src = first + (long long)i; packed[(size_t)q * each * 2 + i] = src >= 0 ? stream[(size_t)src] : 0.0f; }
            }
            CHECK(csdr_dev_upload(ctx, d_packed, packed.data(), packed.size() * 4));
        }
        CHECK(csdr_comm_scatter(comm, (const float *)d_packed, (float *)d_win, (int64_t)each, 0));
        CHECK(csdr_post_set_history(producer, (const float *)d_win, H));
        CHECK(csdr_post_execute(producer, (const float *)d_win + 2 * (size_t)H, 1, 2, block, center));
        CHECK(csdr_post_exchange_rows(comm, producer, owner, all_ch.data(), n_ch.data(), frame0.data(), frames.data(), nb, block, center));
        for (int k : mine) {
            int got_n = 0;
            CHECK(csdr_post_read_channel(whole, k, want.data(), nrow, &got_n)); REQUIRE(got_n == nrow);
            kept.push_back(want);
            CHECK(csdr_post_read_channel(shard, k, row.data(), nrow, &got_n)); REQUIRE(got_n == nrow);
            REQUIRE(std::memcmp(row.data(), want.data(), want.size() * 4) == 0);                   // broadcast variant: bit for bit
            CHECK(csdr_post_read_channel(owner, k, row.data(), nrow, &got_n)); REQUIRE(got_n == nrow);
            if (k != 0) REQUIRE(std::memcmp(row.data(), want.data(), want.size() * 4) == 0);       // slab variant: bit for bit ...
            else {                                                                                // ... but channel 0 (fp64 blocked scan over other tiles)
                double worst = 0.0, peak = 1e-30;
                for (size_t i = 0; i < want.size(); ++i) { worst = std::fmax(worst, std::fabs((double)row[i] - (double)want[i])); peak = std::fmax(peak, std::fabs((double)want[i])); }
                REQUIRE(worst <= 1e-6 * peak);
            }
        }
    }
    // 4. the same time-slab stream with the exchange in two halves (csdr_post_exchange_rows_begin / _finish): batch t is channelized and its row
    //    transfers are started before batch t - 1 is imported -- the order in which the transfers run beside the next batch's channelizer
    {
        csdr_post *producer2 = nullptr, *owner2 = nullptr;
        CHECK(csdr_post_create(ctx, &producer2)); CHECK(csdr_post_configure(producer2, fs, M, CSDR_POST_PFBCH, block, 2));
        CHECK(csdr_post_create(ctx, &owner2)); CHECK(csdr_post_configure(owner2, fs, M, CSDR_POST_PFBCH, block, nb));
        CHECK(csdr_post_set_active_channels(owner2, mine.data(), (int)mine.size()));
        CHECK(csdr_post_set_dc_blocker(producer2, 0));
        CHECK(csdr_post_set_row_order(producer2, all_ch.data(), (int)all_ch.size()));
        auto check_batch = [&](int t) -> int {
            for (size_t j = 0; j < mine.size(); ++j) {
                int got_n = 0;
                CHECK(csdr_post_read_channel(owner2, mine[j], row.data(), nrow, &got_n)); REQUIRE(got_n == nrow);
                const std::vector<float> &w = kept[(size_t)t * mine.size() + j];
                if (mine[j] != 0) REQUIRE(std::memcmp(row.data(), w.data(), w.size() * 4) == 0);
                else {
                    double worst = 0.0, peak = 1e-30;
                    for (size_t i = 0; i < w.size(); ++i) { worst = std::fmax(worst, std::fabs((double)row[i] - (double)w[i])); peak = std::fmax(peak, std::fabs((double)w[i])); }
                    REQUIRE(worst <= 1e-6 * peak);
                }
            }
            return 0;
        };
        for (int t = 0; t < n_batches; ++t) {
            if (rank == 0) {
                std::vector<float> packed(each * 2 * (size_t)world, 0.0f);
                for (int q = 0; q < world; ++q) {
                    const long long first = ((long long)t * (long long)batch + (long long)q * 2 * block - H) * 2;
                    for (size_t i = 0; i < each * 2; ++i) { const long long src = first + (long long)i; packed[(size_t)q * each * 2 + i] = src >= 0 ? stream[(size_t)src] : 0.0f; }
                }
                CHECK(csdr_dev_upload(ctx, d_packed, packed.data(), packed.size() * 4));
            }
            CHECK(csdr_comm_scatter(comm, (const float *)d_packed, (float *)d_win, (int64_t)each, 0));
            CHECK(csdr_post_set_history(producer2, (const float *)d_win, H));
            CHECK(csdr_post_execute(producer2, (const float *)d_win + 2 * (size_t)H, 1, 2, block, center));
            CHECK(csdr_post_exchange_rows_begin(comm, producer2, all_ch.data(), n_ch.data(), frame0.data(), frames.data()));
            if (t > 0) {
                REQUIRE(csdr_comm_exchanges_pending(comm) == 2);
                CHECK(csdr_post_exchange_rows_finish(comm, owner2, nb, block, center));
                if (check_batch(t - 1)) return 1;
            }
        }
        CHECK(csdr_post_exchange_rows_finish(comm, owner2, nb, block, center));
        REQUIRE(csdr_comm_exchanges_pending(comm) == 0);
        if (check_batch(n_batches - 1)) return 1;
        REQUIRE(csdr_comm_async_error(comm) == CSDR_OK);
        csdr_post_destroy(producer2); csdr_post_destroy(owner2);
    }
    CHECK(csdr_comm_barrier(comm));
    csdr_post_destroy(whole); csdr_post_destroy(shard); csdr_post_destroy(producer); csdr_post_destroy(owner);
    csdr_dev_free(ctx, a); csdr_dev_free(ctx, b); csdr_dev_free(ctx, c); csdr_dev_free(ctx, d_batch); csdr_dev_free(ctx, d_packed); csdr_dev_free(ctx, d_win);
    csdr_comm_destroy(comm);
    csdr_ctx_destroy(ctx);
    std::printf("[rank %d of %d] broadcast, scatter, all-to-all, max, sharded rows: ok\n", rank, world);
    std::fflush(stdout);                                                                          // (the rank leaves through _exit)
    return 0;
}

int main(int argc, char **argv) {
    const int world = argc > 1 ? std::atoi(argv[1]) : 1;
    if (world < 1 || world > 8) { std::fprintf(stderr, "usage: comm_ranks WORLD (1..8)\n"); return 2; }
    // pipes first, then fork: no HIP / RCCL state may exist in the parent
    std::vector<int> rd((size_t)world, -1), wr((size_t)world, -1);
    for (int r = 1; r < world; ++r) { int p[2]; if (pipe(p)) return 2; rd[(size_t)r] = p[0]; wr[(size_t)r] = p[1]; }
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        const pid_t pid = fork();
        if (pid < 0) return 2;
        if (pid == 0) {
            char id[CSDR_COMM_ID_BYTES];
            if (r == 0) {
                g_rank = 0;
                if (csdr_comm_unique_id(id) != CSDR_OK) { std::fprintf(stderr, "csdr_comm_unique_id: %s\n", csdr_last_error()); _exit(1); }
                for (int q = 1; q < world; ++q) if (write(wr[(size_t)q], id, sizeof id) != (ssize_t)sizeof id) _exit(1);
            } else if (read(rd[(size_t)r], id, sizeof id) != (ssize_t)sizeof id) _exit(1);
            _exit(run_rank(r, world, id));
        }
        kids.push_back(pid);
    }
    int bad = 0;
    for (pid_t k : kids) { int st = 0; waitpid(k, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = 1; }
    return bad;
}
