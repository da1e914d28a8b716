// hvx_batcher.hip -- the batching operator in front of the search kernels.
//
// The reference calls `ValidatedVectorReadIndex::search` once per operator invocation, from many tokio tasks
// (crates/db/src/execution/interpreter/access/search/storage.rs:140-163): one query per call.  The device kernels are
// built for batches (one wavefront per query; 1 024 queries fill the chip), so concurrent single-query callers are
// coalesced here.  Results are exactly those of a direct batch call (queries are independent).
//
// Round 3 design (round 2's collector, not the device, was the limit: 418 k QPS = 28 % of the batch kernel):
//   * NO lock and NO per-request object on the path of a caller.  One 64-bit word `state` = (batch sequence << 24 | slots
//     claimed; 40 sequence bits never wrap in practice -- round 4, ADVICE r3: a 32-bit sequence with a buffer count that is
//     not a power of two skipped the drain test at the wrap) is the whole queue: a caller claims slot `n` of the open batch with one compare-and-swap, copies its query
//     straight into the batch's PINNED staging row (the row the H2D copy reads), bumps `filled`, and sleeps on the batch's
//     completion word (futex); a dispatcher closes the open batch with one compare-and-swap (sequence + 1, count 0), which
//     at the same instant opens the next batch for the callers that keep arriving.
//   * A batch is closed when it is full -- or as soon as a DEVICE LANE IS FREE and at least one query waits.  No timer in
//     the steady state: while all lanes are busy the open batch simply grows, so the batch size follows the load (a lone
//     caller on an idle device is launched at once; a thousand callers ride in batches of several hundred).
//   * `lanes` dispatcher threads, each with its own execution lane on the shared index image (hvx_index_fork: stream +
//     scratch) and its own device buffers, compete for the open batch: batch i + 1 is launched while batch i runs, like
//     bench.py's lanes.  Everything a launch needs is enqueued on the lane's stream (H2D of the staged rows, the search,
//     D2H of the result rows into pinned memory) followed by ONE stream synchronise, then ONE futex wake for the batch.
//   * A batch buffer is reused only after every caller of its previous batch has copied its rows out (`consumed`).
// Round 4 (VERDICT r3 weak #6, 606 k QPS = 40 % of the batch kernel):
//   * the search kernels write their result rows STRAIGHT into the batch's pinned host rows (mapped, fine-grained: 128 bytes per
//     query over PCIe behind the kernel's last store) -- the four D2H copies per batch and their ~10 us each are gone; only the
//     queries are still copied (the kernels read them more than once);
//   * completion wakes its callers through 32 futex words, the first caller of each group relaying the wake (see Batch::Group);
//   * a free lane does not take the open batch the instant its first query arrives: the callers of the batch that has just
//     completed come back within tens of microseconds, and the lane that launched at the first of them ran a batch of a handful
//     beside the next lane's hundreds (mean 170-300 of 1 024 callers).  A lane now expects as many queries as its previous batch
//     held -- they are the ones coming back -- and sleeps until a caller brings the open batch to that count (the caller rings the
//     lane's bell), the batch is full, or its first query has waited max_wait_us (0 = 200 us).  A lone caller (previous batch: one
//     query) is launched at once; when the load drops, one batch pays max_wait_us and the expectation follows it down;
//   * NOTHING spins: the GPU boxes of this pool run the process under a CPU quota of 16 cores (cgroup cpu.max 1600000 100000;
//     profiles/history/r04m_batcher_cgroup.log shows nr_throttled rising during a 1 024-caller run).  Spinning lanes (hipStreamSynchronize
//     busy-waits, plus this file's former yield loops) burnt the quota, the whole process was frozen for the rest of the 100 ms
//     period, and that -- not the batching policy -- was the 40-75 ms p99 of rounds 3 and 4.  A lane now waits for its stream on an
//     event created with hipEventBlockingSync (an interrupt, no polling) and sleeps on futexes / short timed sleeps elsewhere;
//   * hvx_batcher_lane_times reports where the lanes' time went.
#include <hip/hip_runtime.h>
#include <linux/futex.h>
#include <sys/eventfd.h>
#include <sys/prctl.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <climits>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "hvx_host.h"

using namespace hvx;

namespace {

inline void futex_wait(std::atomic<uint32_t> *addr, uint32_t expected, long timeout_us = -1) {
    timespec ts, *tp = nullptr;
    if (timeout_us >= 0) { ts.tv_sec = timeout_us / 1000000; ts.tv_nsec = (timeout_us % 1000000) * 1000; tp = &ts; }
    syscall(SYS_futex, reinterpret_cast<uint32_t *>(addr), FUTEX_WAIT_PRIVATE, expected, tp, nullptr, 0);
}
inline void futex_wake(std::atomic<uint32_t> *addr, int n) {
    syscall(SYS_futex, reinterpret_cast<uint32_t *>(addr), FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0);
}

constexpr uint64_t kStampMask = (1ull << 44) - 1;
inline uint64_t now_us_stamp() { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() & kStampMask; }
inline int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline void cpu_pause() { __builtin_ia32_pause(); }
inline void sleep_us(long us) {
    timespec ts{us / 1000000, (us % 1000000) * 1000};
    clock_nanosleep(CLOCK_MONOTONIC, 0, &ts, nullptr);
}

// one batch in flight between callers and a lane: pinned host staging + completion state
struct Batch {
    float *q = nullptr;            // [max_batch][dim] pinned
    uint64_t *ids = nullptr;       // [max_batch][k] pinned
    float *sc = nullptr;           // [max_batch][k] pinned
    uint32_t *cnt = nullptr, *st = nullptr; // [max_batch] pinned
    // the prefiltered branch (hvx_batcher_new_restricted): every slot's own candidate ids, fixed stride, and their number
    uint64_t *cand = nullptr;      // [max_batch][ids_cap] pinned
    uint32_t *cand_n = nullptr;    // [max_batch] pinned
    uint64_t *dev_cand = nullptr;  // ... as the device addresses them
    uint32_t *dev_cand_n = nullptr;
    uint64_t *dev_ids = nullptr;   // the same four arrays as the device addresses them (hipHostGetDevicePointer)
    float *dev_sc = nullptr;
    uint32_t *dev_cnt = nullptr, *dev_st = nullptr;
    // when the open batch got its first query: (sequence & 0xFFFFF) << 44 | steady-clock microseconds, written by the caller of slot 0.
    // The tag says WHICH batch the stamp belongs to -- nobody resets the word (a reset by a lane that lost the race for the close could
    // land after the next batch's first caller had stamped it, and a lane that finds no stamp would wait for its count for ever)
    std::atomic<uint64_t> t_first{0};
    // cumulative over every batch this buffer has carried (never reset: a reset could race with the first callers of the
    // next batch); batches of one buffer are strictly sequential, so "all of them" is always "the previous ones + this one"
    // `filled` / `consumed` live in the groups below (slot % kGroups), one cache line each: a counter every caller of a batch adds
    // to is a line that 300 cores hand around -- tens of microseconds of stalled CPU per caller on a two-socket host, and CPU time is
    // what this pool rations (16 cores per process)
    alignas(64) std::atomic<uint32_t> done{0};     // sequence + 1 of the last batch completed in this buffer
    // Callers sleep on one of kGroups words (slot % kGroups), not on `done`: waking a few hundred waiters of ONE futex word is a
    // serial walk of its hash bucket by one thread (~1.5 us per waiter: the last of 300 callers left 0.4 ms after the first).  The
    // completing thread wakes ONE sleeper per group; the first caller of a group to see the batch complete wakes the rest of its
    // group -- kGroups buckets drained by kGroups threads at once.  (Round 3's fan-out had every woken caller call FUTEX_WAKE on the
    // SAME word and was 10 x slower: they all met on one bucket lock.)
    struct alignas(64) Group {
        std::atomic<uint32_t> word{0};  // sequence + 1 of the last completed batch (futex word)
        std::atomic<uint32_t> relay{0}; // sequence + 1 of the last batch whose group-wide wake has been issued
        std::atomic<uint64_t> filled{0};   // callers of this group that finished writing their row (cumulative, like `total`)
        std::atomic<uint64_t> consumed{0}; // callers of this group that copied their result out
    };
    uint64_t filled_sum() const { uint64_t v = 0; for (const Group &g : grp) v += g.filled.load(); return v; }
    uint64_t consumed_sum() const { uint64_t v = 0; for (const Group &g : grp) v += g.consumed.load(); return v; }
    static constexpr uint32_t kGroups = 32;
    Group grp[kGroups];
    uint64_t total = 0;            // slots of all batches closed in this buffer (written by the closing dispatcher before `done`)
    int rc = 0;                    // status of the launch as a whole
    std::string err;
};

struct Lane {
    hvx_index *ix = nullptr;
    std::thread worker, waker;
    // hand-over to the lane's waker thread (round 4): waking a few hundred futex waiters costs the kernel ~1 us each -- 0.2-0.4 ms per
    // batch during which the lane launched nothing.  The dispatcher publishes the finished batch here and goes back to the queue.
    alignas(64) std::atomic<Batch *> wake_batch{nullptr};
    std::atomic<uint32_t> wake_seq{0};
    alignas(64) std::atomic<uint32_t> wake_bell{0};
    uint32_t seen_refresh = 0;
    float *d_q = nullptr; // the lane's query rows in HBM (results go straight to the batch's pinned rows)
    uint64_t *d_cand = nullptr; // restricted batcher: the batch's candidate ids in HBM (same slots)
    hipEvent_t ev = nullptr; // hipEventBlockingSync: the lane sleeps until the stream's work is done
    // where this lane's time went, ns (hvx_batcher_timing)
    std::atomic<uint64_t> ns_idle{0}, ns_collect{0}, ns_drain{0}, ns_fill{0}, ns_device{0}, ns_wake{0};
};

} // namespace

constexpr unsigned kSeqShift = 24;               // state word: 40-bit batch sequence | 24-bit claims (may overshoot max_batch, see the claim loop)
constexpr uint64_t kCountMask = 0xFFFFFFull;

struct hvx_batcher {
    hvx_search_params params{};
    bool restricted = false;       // the operator's Some(candidates) branch: callers bring their own candidate ids
    uint32_t ids_cap = 0;          // ... at most this many per caller
    uint32_t max_batch = 0, max_wait_us = 0, dim = 0, k = 0, nbuf = 0;
    int device = 0;
    alignas(64) std::atomic<uint64_t> state{0};      // (sequence of the open batch) << kSeqShift (24) | slots claimed
    alignas(64) std::atomic<uint32_t> seq_word{0};   // low 32 bits of the open sequence: callers of a full batch sleep on it
    alignas(64) std::atomic<uint32_t> bell{0};       // dispatchers sleep on it; rung by the first and the last claim of a batch
    alignas(64) std::atomic<uint32_t> sleepers{0};   // dispatchers asleep on the bell
    alignas(64) std::atomic<uint32_t> target{0};     // the count a waiting lane wants the open batch to reach: the caller that brings it there rings the bell
    alignas(64) std::atomic<bool> stop{false};       // (its own line: every caller reads it)
    struct alignas(64) Inside { std::atomic<uint32_t> n{0}; };
    Inside inside[64];                               // callers currently inside hvx_batcher_search, striped by thread (hvx_batcher_free waits for them)
    uint32_t inside_sum() const { uint32_t v = 0; for (const Inside &i : inside) v += i.n.load(); return v; }
    std::vector<Batch> bufs;
    std::vector<Lane> lanes;
    std::atomic<uint64_t> n_batches{0}, n_queries{0}, n_full{0};
    std::atomic<int> efd{-1};                        // hvx_batcher_eventfd: written once per completed batch (non-blocking hosts)
    std::atomic<uint32_t> refresh_gen{0};            // hvx_batcher_refresh: lanes adopt the image's visible generation before their next batch

    void run(Lane &ln) {
        (void)hipSetDevice(device);
        (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0); // this thread's timed sleeps end within ~1 us of their time (default slack: 50 us)
        const int64_t max_wait_ns = (int64_t)(max_wait_us ? max_wait_us : 200u) * 1000;
        uint32_t stop_spins = 0;
        uint32_t expect = 1; // queries the open batch should hold before this lane takes it: the size of the lane's previous batch
        uint64_t seen_seq = ~0ull, seen_us = 0; // when THIS lane first saw the open batch non-empty (the fallback age base)
        // microseconds the open batch's first query has waited: by its caller's stamp when it carries this batch's tag, else since
        // this lane first saw the batch (a stamp is missing only for the instant between a claim and its store)
        auto waited_us = [&](uint64_t seq) -> int64_t {
            const uint64_t now = now_us_stamp(), tf = bufs[seq % nbuf].t_first.load(std::memory_order_relaxed);
            if (seen_seq != seq) { seen_seq = seq; seen_us = now; }
            const uint64_t base = (tf >> 44) == (seq & 0xFFFFFull) ? (tf & kStampMask) : seen_us;
            return (int64_t)((now - base) & kStampMask);
        };
        int64_t t_mark = now_ns();
        auto lap = [&](std::atomic<uint64_t> &acc) { const int64_t t = now_ns(); acc.fetch_add((uint64_t)(t - t_mark), std::memory_order_relaxed); t_mark = t; };
        for (;;) {
            // this lane is free: look at the open batch
            uint64_t s = state.load();
            uint32_t cnt = std::min<uint32_t>((uint32_t)(s & kCountMask), max_batch); // (claims past max_batch are void)
            if (cnt == 0) { // (sequentially consistent operations: a caller either sees this sleeper or this sleeper sees its claim)
                if (stop.load()) return;
                const uint32_t b0 = bell.load();
                sleepers.fetch_add(1);
                if ((state.load() & kCountMask) == 0 && !stop.load()) futex_wait(&bell, b0, 2000);
                sleepers.fetch_sub(1);
                lap(ln.ns_idle);
                continue;
            }
            const uint64_t seq = s >> kSeqShift;
            // collect: the callers of this lane's previous batch are on their way back -- sleep until the open batch holds as many
            // queries as that batch did (a caller rings the bell at that count), is full, or its first query has waited max_wait_us
            const uint32_t need = expect < max_batch ? expect : max_batch;
            if (cnt < need && !stop.load()) {
                const int64_t left = max_wait_ns - waited_us(seq) * 1000;
                if (left > 2000) {
                    const uint32_t b0 = bell.load();
                    target.store(need);
                    sleepers.fetch_add(1);
                    const uint64_t s2 = state.load();
                    if ((s2 >> kSeqShift) == seq && (uint32_t)(s2 & kCountMask) < need && !stop.load()) futex_wait(&bell, b0, (long)(left / 1000));
                    sleepers.fetch_sub(1);
                    lap(ln.ns_collect);
                    const uint64_t s3 = state.load();
                    if ((s3 >> kSeqShift) != seq) continue; // another lane took it
                    const uint32_t c3 = std::min<uint32_t>((uint32_t)(s3 & kCountMask), max_batch);
                    if (c3 < need && waited_us(seq) * 1000 < max_wait_ns && !stop.load()) continue; // woken for something else: look again
                    s = s3;
                    cnt = c3;
                }
            }
            // the next batch opens in buffer (seq + 1) % nbuf at the instant this one closes: it must be free, i.e. every
            // caller of the batch it held last has taken its rows
            Batch &next = bufs[(seq + 1) % nbuf];
            // (also while stopping -- ADVICE r4: a batch opened in a buffer whose previous batch is still in flight on another lane would
            // mix the two batches' rows; the wait is bounded: that lane completes its batch, and every caller blocked on it wakes within
            // 5 ms and either takes its rows or leaves with its slot accounted for)
            if (seq + 1 >= nbuf) { // its previous batch (sequence seq + 1 - nbuf) must be complete and fully drained
                if (next.done.load() != (uint32_t)(seq + 1 - nbuf + 1) || next.consumed_sum() < next.total) { // (<: a ticket polled twice -- a host bug -- must not wedge the lanes)
                    // stopping, and that buffer is held by tickets nobody will poll any more (a non-blocking host that went away): the
                    // open batch is not launched -- its callers leave with "shutting down" like everybody else (0.2 s of grace)
                    if (stop.load() && ++stop_spins > 20000u) return;
                    sleep_us(10); // (a caller of that batch has not copied its rows out yet)
                    lap(ln.ns_drain);
                    continue;
                }
            }
            if (!state.compare_exchange_strong(s, (seq + 1) << kSeqShift)) continue; // another claim or another lane won
            seq_word.store((uint32_t)(seq + 1));
            // callers that found the batch full: as many as the new batch has slots (FIFO).  Waking them all made every close a
            // stampede of losers when more callers wait than a batch holds (2 048 callers on 1 024-slot batches: 16 cores burnt,
            // 120 k QPS; 256 callers on 16-slot batches: 6 k) -- the ones left asleep are woken by the next close
            futex_wake(&seq_word, (int)max_batch);
            Batch &bt = bufs[seq % nbuf];
            bt.total += cnt;
            for (uint32_t spins = 0; bt.filled_sum() < bt.total; ++spins) { // callers still copying their row in (3 KB: normally done)
                if (spins < 64) cpu_pause();
                else sleep_us(5);
            }
            lap(ln.ns_fill);
            n_batches.fetch_add(1, std::memory_order_relaxed);
            n_queries.fetch_add(cnt, std::memory_order_relaxed);
            if (cnt == max_batch) n_full.fetch_add(1, std::memory_order_relaxed);
            launch(ln, bt, cnt);
            expect = cnt;
            lap(ln.ns_device);
            Batch *expected = nullptr;
            if (ln.waker.joinable() && ln.wake_batch.load() == nullptr) { // the waker is idle: it completes the batch, this lane goes on
                ln.wake_seq.store((uint32_t)(seq + 1));
                if (ln.wake_batch.compare_exchange_strong(expected, &bt)) {
                    ln.wake_bell.fetch_add(1);
                    futex_wake(&ln.wake_bell, 1);
                    lap(ln.ns_wake);
                    continue;
                }
            }
            complete(bt, (uint32_t)(seq + 1));
            lap(ln.ns_wake);
        }
    }

    // publish a finished batch: results are in its pinned rows
    void complete(Batch &bt, uint32_t want) {
        bt.done.store(want);
        for (Batch::Group &g : bt.grp) g.word.store(want);
        for (Batch::Group &g : bt.grp) futex_wake(&g.word, 1);
        const int fd = efd.load(std::memory_order_acquire);
        if (fd >= 0) { // a reactor-driven host: one eventfd tick per completed batch (the host then polls its tickets)
            const uint64_t one = 1;
            (void)!write(fd, &one, sizeof(one));
        }
    }

    void wake_loop(Lane &ln) {
        for (;;) {
            const uint32_t b0 = ln.wake_bell.load();
            Batch *bt = ln.wake_batch.load();
            if (!bt) {
                if (stop.load()) return;
                futex_wait(&ln.wake_bell, b0, 2000);
                continue;
            }
            const uint32_t want = ln.wake_seq.load();
            complete(*bt, want);
            ln.wake_batch.store(nullptr);
        }
    }

    void launch(Lane &ln, Batch &bt, uint32_t cnt) {
        hipStream_t s = (hipStream_t)hvx_index_stream(ln.ix);
        bt.rc = 0;
        bt.err.clear();
        const uint32_t rg = refresh_gen.load(std::memory_order_acquire);
        if (rg != ln.seen_refresh) { // rows were appended to the image (hvx_index_insert_batch): this lane's next batches see them
            ln.seen_refresh = rg;
            (void)hvx_index_refresh(ln.ix);
        }
        // on any failure nothing may still be writing the batch's pinned rows when the buffer is handed back: drain the stream first
        auto bad = [&](const char *what, hipError_t e) {
            bt.rc = HVX_ERR_DEVICE;
            bt.err = std::string(what) + ": " + hipGetErrorString(e);
            (void)hipStreamSynchronize(s);
        };
        hipError_t e = hipMemcpyAsync(ln.d_q, bt.q, (size_t)cnt * dim * 4, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return bad("hipMemcpyAsync(queries)", e);
        int rc;
        if (restricted) {
            // candidate ids: a copy kernel reads the slots' USED parts out of pinned memory (a DMA copy would move every slot's full stride);
            // then ONE launch scans every caller's own candidate set (csrc/hvx_restricted_exact.hip)
            uint32_t max_len = 0;
            for (uint32_t i = 0; i < cnt; ++i) max_len = std::max(max_len, std::min(bt.cand_n[i], ids_cap));
            if ((e = launch_stage_ids(bt.dev_cand, bt.dev_cand_n, ids_cap, ln.d_cand, cnt, s)) != hipSuccess) return bad("stage_ids_kernel", e);
            rc = hvx_search_restricted_lists_device(ln.ix, ln.d_q, cnt, k, ln.d_cand, ids_cap, bt.dev_cand_n, max_len, bt.dev_ids, bt.dev_sc, bt.dev_cnt, bt.dev_st);
        } else
            rc = hvx_search_batch_params_device(ln.ix, ln.d_q, cnt, &params, bt.dev_ids, bt.dev_sc, bt.dev_cnt, bt.dev_st, nullptr, nullptr, nullptr);
        if (rc) { bt.rc = rc; bt.err = hvx_last_error(); (void)hipStreamSynchronize(s); return; }
        if ((e = hipEventRecord(ln.ev, s)) != hipSuccess) return bad("hipEventRecord", e);
        if ((e = hipEventSynchronize(ln.ev)) != hipSuccess) return bad("hipEventSynchronize", e); // (blocking event: no busy-wait)
    }
};

extern "C" void hvx_batcher_free(hvx_batcher *b) {
    if (!b) return;
    b->stop.store(true, std::memory_order_release);
    b->bell.fetch_add(1, std::memory_order_acq_rel);
    futex_wake(&b->bell, INT_MAX);
    for (Lane &ln : b->lanes)
        if (ln.worker.joinable()) ln.worker.join();
    for (Lane &ln : b->lanes) { // (after the dispatchers: a batch they handed over is still completed)
        ln.wake_bell.fetch_add(1);
        futex_wake(&ln.wake_bell, 1);
        if (ln.waker.joinable()) ln.waker.join();
    }
    // callers that were blocked when the batcher stopped leave with "shutting down" (their futex waits time out within 5 ms): the
    // object outlives the last of them.  (Calling hvx_batcher_search AFTER hvx_batcher_free has returned is the host's bug.)
    for (uint32_t spins = 0; b->inside_sum() != 0 && spins < 200000u; ++spins) {
        for (Batch &bt : b->bufs)
            for (Batch::Group &g : bt.grp) futex_wake(&g.word, INT_MAX);
        futex_wake(&b->seq_word, INT_MAX);
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    (void)hipSetDevice(b->device);
    for (Lane &ln : b->lanes) {
        if (ln.d_q) (void)hipFree(ln.d_q);
        if (ln.d_cand) (void)hipFree(ln.d_cand);
        if (ln.ev) (void)hipEventDestroy(ln.ev);
        if (ln.ix) hvx_index_free(ln.ix);
    }
    for (Batch &bt : b->bufs)
        for (void *p : {(void *)bt.q, (void *)bt.ids, (void *)bt.sc, (void *)bt.cnt, (void *)bt.st, (void *)bt.cand, (void *)bt.cand_n})
            if (p) (void)hipHostFree(p);
    if (b->efd.load() >= 0) (void)close(b->efd.load());
    delete b;
}

static int batcher_create(hvx_index *ix, const hvx_search_params *params, uint32_t k, uint32_t max_batch, uint32_t max_wait_us, uint32_t lanes,
                          uint32_t ids_cap, hvx_batcher **out) {
    *out = nullptr;
    if (max_batch == 0) max_batch = ix->max_batch;
    if (max_batch > ix->max_batch) return fail(HVX_ERR_UNSUPPORTED, "batch %u exceeds max_batch %u given at import", max_batch, ix->max_batch);
    if (max_batch > 0xFFFFu) max_batch = 0xFFFFu; // (the state word's 24 claim bits leave room for 16 million void claims beyond it)
    if (lanes == 0) lanes = 4;
    if (lanes > 8) return fail(HVX_ERR_K_RANGE, "at most 8 dispatcher lanes");
    if (hipSetDevice(ix->device) != hipSuccess) return fail(HVX_ERR_DEVICE, "hipSetDevice failed");
    hvx_batcher *b = new hvx_batcher();
    if (params) b->params = *params;
    b->restricted = ids_cap != 0;
    b->ids_cap = ids_cap;
    b->max_batch = max_batch;
    b->max_wait_us = max_wait_us;
    b->dim = ix->dev.dim;
    b->k = k;
    b->device = ix->device;
    b->nbuf = lanes + 2; // one open batch, one per lane in flight, one being drained by its callers
    b->bufs = std::vector<Batch>(b->nbuf);
    b->lanes = std::vector<Lane>(lanes);
    auto host = [&](void **p, size_t bytes) { return hipHostMalloc(p, bytes, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess; };
    auto devptr = [&](void **d, void *h) { return hipHostGetDevicePointer(d, h, 0) == hipSuccess; };
    auto dev = [&](void **p, size_t bytes) { return hipMalloc(p, bytes) == hipSuccess; };
    bool ok = true;
    for (Batch &bt : b->bufs) {
        ok = ok && host((void **)&bt.q, (size_t)max_batch * b->dim * 4) && host((void **)&bt.ids, (size_t)max_batch * b->k * 8) &&
             host((void **)&bt.sc, (size_t)max_batch * b->k * 4) && host((void **)&bt.cnt, (size_t)max_batch * 4) &&
             host((void **)&bt.st, (size_t)max_batch * 4) && devptr((void **)&bt.dev_ids, bt.ids) && devptr((void **)&bt.dev_sc, bt.sc) &&
             devptr((void **)&bt.dev_cnt, bt.cnt) && devptr((void **)&bt.dev_st, bt.st);
        if (b->restricted)
            ok = ok && host((void **)&bt.cand, (size_t)max_batch * ids_cap * 8) && host((void **)&bt.cand_n, (size_t)max_batch * 4) &&
                 devptr((void **)&bt.dev_cand, bt.cand) && devptr((void **)&bt.dev_cand_n, bt.cand_n);
    }
    int rc;
    for (Lane &ln : b->lanes) {
        if (ok && (rc = hvx_index_fork(ix, &ln.ix))) { // own stream + scratch on the shared image (SimHash rows included)
            hvx_batcher_free(b);
            return rc;
        }
        // (a lane inherits the parent handle's settings -- hvx_index_set_occupancy / hvx_index_set_option -- at this point)
        ok = ok && dev((void **)&ln.d_q, (size_t)max_batch * b->dim * 4) &&
             hipEventCreateWithFlags(&ln.ev, hipEventBlockingSync | hipEventDisableTiming) == hipSuccess;
        if (b->restricted) ok = ok && dev((void **)&ln.d_cand, (size_t)max_batch * ids_cap * 8);
    }
    if (!ok) {
        hvx_batcher_free(b);
        return fail(HVX_ERR_DEVICE, "allocation of the batcher's staging buffers failed");
    }
    for (Lane &ln : b->lanes) {
        Lane *lp = &ln;
        ln.waker = std::thread([b, lp] { b->wake_loop(*lp); });
        ln.worker = std::thread([b, lp] { b->run(*lp); });
    }
    *out = b;
    return HVX_OK;
}

extern "C" int hvx_batcher_new_lanes(hvx_index *ix, const hvx_search_params *params, uint32_t max_batch, uint32_t max_wait_us,
                                     uint32_t lanes, hvx_batcher **out) {
    if (!ix || !params || !out) return fail(HVX_ERR_INVARIANT, "null argument");
    *out = nullptr;
    int rc = check_k_ef(params->k, params->ef);
    if (rc) return rc;
    return batcher_create(ix, params, params->k, max_batch, max_wait_us, lanes, 0, out);
}

// The prefiltered branch of the operator (execution/interpreter/access/search/storage.rs:140-163: `Some(candidates) =>
// index.search_restricted(..)`): every caller brings its OWN candidate ids; a batch of them is ONE launch of the exact scan
// (csrc/hvx_restricted_exact.hip).  The plan must answer such sets exactly: strategy EXACT, or AUTO with slots whose rows stay within the
// device plan's limit; k <= 64.  Candidate lists longer than max_ids_per_query do not ride in a batch: the host calls
// hvx_search_restricted_batch_params for them (a list that long keeps the device busy by itself).
extern "C" int hvx_batcher_new_restricted(hvx_index *ix, const hvx_restricted_params *params, uint32_t max_batch, uint32_t max_wait_us,
                                          uint32_t lanes, uint32_t max_ids_per_query, hvx_batcher **out) {
    if (!ix || !params || !out) return fail(HVX_ERR_INVARIANT, "null argument");
    *out = nullptr;
    if (params->k == 0) return fail(HVX_ERR_K_RANGE, "result count must be non-zero");
    if (params->ef < params->k) return fail(HVX_ERR_K_RANGE, "search beam width %u is below the result count %u", params->ef, params->k);
    if (max_ids_per_query == 0) max_ids_per_query = 4096;
    if (max_ids_per_query > 1000000u) return fail(HVX_ERR_CANDIDATE_LIMIT, "restricted vector search accepts at most 1000000 unique candidates");
    if (!restricted_direct_supported(ix, params->k))
        return fail(HVX_ERR_UNSUPPORTED, "the batched prefiltered search serves k <= 64 over f32 / bf16 rows of a non-empty image");
    RestrictedPlan plan;
    int rc = restricted_make_plan(*params, max_ids_per_query, ix->dev.dim, &plan, ix);
    if (rc) return rc;
    if (plan.strategy != HVX_RESTRICTED_EXACT || params->explicit_budgets)
        return fail(HVX_ERR_UNSUPPORTED, "the batched prefiltered search answers candidate sets with the exact scan: strategy EXACT, or AUTO with "
                    "max_ids_per_query x row bytes within the device plan's limit (the filter-aware walk is not batched per caller)");
    return batcher_create(ix, nullptr, params->k, max_batch, max_wait_us, lanes, max_ids_per_query, out);
}

extern "C" int hvx_batcher_new(hvx_index *ix, const hvx_search_params *params, uint32_t max_batch, uint32_t max_wait_us,
                               hvx_batcher **out) {
    return hvx_batcher_new_lanes(ix, params, max_batch, max_wait_us, 0, out);
}

namespace {
struct Inside { // callers inside an entry point, striped by thread (hvx_batcher_free waits for them)
    hvx_batcher *b;
    uint32_t stripe;
    explicit Inside(hvx_batcher *bb) : b(bb) {
        static std::atomic<uint32_t> next_stripe{0};
        thread_local const uint32_t mine = next_stripe.fetch_add(1, std::memory_order_relaxed) & 63u;
        stripe = mine;
        b->inside[stripe].n.fetch_add(1);
    }
    ~Inside() { b->inside[stripe].n.fetch_sub(1); }
};

// claim a slot of the open batch: one fetch-and-add.  blocking = false: a full batch returns HVX_ERR_BUSY instead of waiting for the
// next one to open (a free lane closes the full batch within microseconds: the host retries after its next completion or yield).
int claim_slot(hvx_batcher *b, bool blocking, uint64_t *seq_out, uint32_t *slot_out) {
    for (;;) {
        if (b->stop.load(std::memory_order_acquire)) return fail(HVX_ERR_INVARIANT, "batcher is shutting down");
        uint64_t s = b->state.load();
        uint64_t seq = s >> kSeqShift;
        if ((uint32_t)(s & kCountMask) >= b->max_batch) { // full: a lane closes it as soon as one is free; wait for the next batch to open
            if (!blocking) return fail(HVX_ERR_BUSY, "the open batch is full: submit again after a completion");
            futex_wait(&b->seq_word, (uint32_t)seq, 2000); // (woken by the close; the time-out is a safety net, not a poll)
            continue;
        }
        // ONE fetch-and-add, no retry loop: the callers of a completed batch come back together, and a compare-and-swap loop over
        // one word costs a burst of N callers ~N^2 / 2 failed attempts (300 callers: ~45 000 cache-line transfers, several ms of CPU
        // per batch -- under this pool's 16-core quota that alone throttled the process).  A claim that lands past max_batch (the batch
        // filled between the load and the add) is void: the dispatcher counts min(claims, max_batch), and the caller tries the next batch.
        s = b->state.fetch_add(1);
        seq = s >> kSeqShift;
        const uint32_t slot = (uint32_t)(s & kCountMask);
        if (slot < b->max_batch) { *seq_out = seq; *slot_out = slot; return HVX_OK; }
        if (!blocking) return fail(HVX_ERR_BUSY, "the open batch is full: submit again after a completion");
        futex_wait(&b->seq_word, (uint32_t)seq, 2000);
    }
}

// copy the query into the batch's pinned staging row and tell the lanes
void fill_slot(hvx_batcher *b, uint64_t seq, uint32_t slot, const float *query, const uint64_t *allowed_ids = nullptr, uint32_t n_allowed = 0) {
    Batch &bt = b->bufs[seq % b->nbuf];
    if (slot == 0) bt.t_first.store(((seq & 0xFFFFFull) << 44) | now_us_stamp(), std::memory_order_relaxed);
    memcpy(bt.q + (size_t)slot * b->dim, query, (size_t)b->dim * 4);
    if (b->restricted) {
        if (n_allowed) memcpy(bt.cand + (size_t)slot * b->ids_cap, allowed_ids, (size_t)n_allowed * 8);
        bt.cand_n[slot] = n_allowed;
    }
    bt.grp[slot % Batch::kGroups].filled.fetch_add(1);
    if ((slot == 0 || slot + 1 == b->max_batch || slot + 1 == b->target.load()) && b->sleepers.load()) { // first / last query of a batch, or the count a waiting lane asked for
        b->bell.fetch_add(1);
        futex_wake(&b->bell, INT_MAX); // (at most 8 lanes; each looks at the batch and goes back to sleep if it is not the one to take it)
    }
}

// the batch is complete: hand the slot's rows to the caller and release the slot
int take_result(hvx_batcher *b, uint64_t seq, uint32_t slot, uint64_t *out_ids, float *out_scores, uint32_t *out_count) {
    Batch &bt = b->bufs[seq % b->nbuf];
    Batch::Group &grp = bt.grp[slot % Batch::kGroups];
    const uint32_t want = (uint32_t)(seq + 1);
    // the first caller of the group to get here passes the wake on to the group's other sleepers (nobody can fall asleep on the
    // word any more: it already holds `want`)
    if (grp.relay.exchange(want, std::memory_order_acq_rel) != want) futex_wake(&grp.word, INT_MAX);
    int rc = bt.rc;
    std::string err;
    uint32_t st = 0;
    if (rc) err = bt.err;
    else if ((st = bt.st[slot]) != 0) { rc = (int)st; } // a rejected query fails alone (per-query status)
    else {
        const uint32_t c = bt.cnt[slot];
        *out_count = c;
        memcpy(out_ids, bt.ids + (size_t)slot * b->k, (size_t)c * 8);
        memcpy(out_scores, bt.sc + (size_t)slot * b->k, (size_t)c * 4);
    }
    grp.consumed.fetch_add(1);
    if (rc) return err.empty() ? fail(rc, "query rejected with status %u", st) : fail(rc, "%s", err.c_str());
    return HVX_OK;
}

// block until the slot's batch is complete (timeout_us < 0: for ever); HVX_PENDING on time-out
int await_slot(hvx_batcher *b, uint64_t seq, uint32_t slot, long timeout_us) {
    Batch &bt = b->bufs[seq % b->nbuf];
    Batch::Group &grp = bt.grp[slot % Batch::kGroups];
    const uint32_t want = (uint32_t)(seq + 1);
    const int64_t t_end = timeout_us >= 0 ? now_ns() + (int64_t)timeout_us * 1000 : 0;
    for (;;) {
        const uint32_t d = grp.word.load(std::memory_order_acquire);
        if (d == want) return HVX_OK;
        long wait_us = 5000;
        if (timeout_us >= 0) {
            const int64_t left = (t_end - now_ns()) / 1000;
            if (left <= 0) return HVX_PENDING;
            wait_us = left < wait_us ? (long)left : wait_us;
        }
        futex_wait(&grp.word, d, wait_us);
        if (b->stop.load(std::memory_order_acquire) && grp.word.load(std::memory_order_acquire) != want) {
            grp.consumed.fetch_add(1); // the slot is accounted for: a dispatcher draining this buffer must not wait for it
            return fail(HVX_ERR_INVARIANT, "batcher is shutting down");
        }
    }
}

inline bool ticket_ok(const hvx_batcher *b, const hvx_batcher_ticket *t) { return t && t->slot < b->max_batch; }
} // namespace

static int check_restricted_call(const hvx_batcher *b, const uint64_t *allowed_ids, uint32_t n_allowed) {
    if (!b->restricted) return fail(HVX_ERR_INVARIANT, "this batcher serves the unrestricted branch (hvx_batcher_new): create one with hvx_batcher_new_restricted");
    if (n_allowed && !allowed_ids) return fail(HVX_ERR_INVARIANT, "null candidate list");
    if (n_allowed > b->ids_cap)
        return fail(HVX_ERR_UNSUPPORTED, "%u candidate ids exceed the batcher's max_ids_per_query %u: call hvx_search_restricted_batch_params for this query", n_allowed, b->ids_cap);
    return HVX_OK;
}

// ValidatedVectorReadIndex::search_restricted awaited to completion (read_index.rs:93-102) for ONE caller with its own candidate ids
// (any order, duplicates count once, ids that hold no vector are skipped; an empty list answers with nothing before any validation)
extern "C" int hvx_batcher_search_restricted(hvx_batcher *b, const float *query, const uint64_t *allowed_ids, uint32_t n_allowed, uint64_t *out_ids,
                                             float *out_scores, uint32_t *out_count) {
    if (!b || !query || !out_ids || !out_scores || !out_count) return fail(HVX_ERR_INVARIANT, "null argument");
    int rc = check_restricted_call(b, allowed_ids, n_allowed);
    if (rc) return rc;
    Inside in(b);
    *out_count = 0;
    uint64_t seq;
    uint32_t slot;
    if ((rc = claim_slot(b, /*blocking=*/true, &seq, &slot))) return rc;
    fill_slot(b, seq, slot, query, allowed_ids, n_allowed);
    if ((rc = await_slot(b, seq, slot, -1))) return rc;
    return take_result(b, seq, slot, out_ids, out_scores, out_count);
}

extern "C" int hvx_batcher_submit_restricted(hvx_batcher *b, const float *query, const uint64_t *allowed_ids, uint32_t n_allowed,
                                             hvx_batcher_ticket *out_ticket) {
    if (!b || !query || !out_ticket) return fail(HVX_ERR_INVARIANT, "null argument");
    int rc = check_restricted_call(b, allowed_ids, n_allowed);
    if (rc) return rc;
    Inside in(b);
    uint64_t seq;
    uint32_t slot;
    if ((rc = claim_slot(b, /*blocking=*/false, &seq, &slot))) return rc;
    fill_slot(b, seq, slot, query, allowed_ids, n_allowed);
    out_ticket->sequence = seq;
    out_ticket->slot = slot;
    out_ticket->reserved = 0;
    return HVX_OK;
}

// The blocking call = submit + wait (ValidatedVectorReadIndex::search awaited to completion, read_index.rs:81-102).
extern "C" int hvx_batcher_search(hvx_batcher *b, const float *query, uint64_t *out_ids, float *out_scores, uint32_t *out_count) {
    if (!b || !query || !out_ids || !out_scores || !out_count) return fail(HVX_ERR_INVARIANT, "null argument");
    if (b->restricted) return fail(HVX_ERR_INVARIANT, "this batcher serves the prefiltered branch: call hvx_batcher_search_restricted");
    Inside in(b);
    *out_count = 0;
    uint64_t seq;
    uint32_t slot;
    int rc = claim_slot(b, /*blocking=*/true, &seq, &slot);
    if (rc) return rc;
    fill_slot(b, seq, slot, query);
    if ((rc = await_slot(b, seq, slot, -1))) return rc;
    return take_result(b, seq, slot, out_ids, out_scores, out_count);
}

// ---- the non-blocking form (round 5): `search` is an async fn called from tokio tasks (read_index.rs:81-102); a task must not park
//      an OS thread per query in flight.  submit claims a slot and copies the query in (no syscall on the fast path), poll tests one
//      futex word, and hvx_batcher_eventfd hands the host an fd its reactor can await: one tick per completed batch. ----
extern "C" int hvx_batcher_submit(hvx_batcher *b, const float *query, hvx_batcher_ticket *out_ticket) {
    if (!b || !query || !out_ticket) return fail(HVX_ERR_INVARIANT, "null argument");
    if (b->restricted) return fail(HVX_ERR_INVARIANT, "this batcher serves the prefiltered branch: call hvx_batcher_submit_restricted");
    Inside in(b);
    uint64_t seq;
    uint32_t slot;
    const int rc = claim_slot(b, /*blocking=*/false, &seq, &slot);
    if (rc) return rc;
    fill_slot(b, seq, slot, query);
    out_ticket->sequence = seq;
    out_ticket->slot = slot;
    out_ticket->reserved = 0;
    return HVX_OK;
}

extern "C" int hvx_batcher_poll(hvx_batcher *b, const hvx_batcher_ticket *ticket, uint64_t *out_ids, float *out_scores, uint32_t *out_count) {
    if (!b || !out_ids || !out_scores || !out_count || !ticket_ok(b, ticket)) return fail(HVX_ERR_INVARIANT, "null argument or malformed ticket");
    Inside in(b);
    Batch &bt = b->bufs[ticket->sequence % b->nbuf];
    Batch::Group &grp = bt.grp[ticket->slot % Batch::kGroups];
    if (grp.word.load(std::memory_order_acquire) != (uint32_t)(ticket->sequence + 1)) {
        if (b->stop.load(std::memory_order_acquire)) {
            grp.consumed.fetch_add(1);
            return fail(HVX_ERR_INVARIANT, "batcher is shutting down");
        }
        return HVX_PENDING; // (not an error: hvx_last_error is left alone)
    }
    *out_count = 0;
    return take_result(b, ticket->sequence, ticket->slot, out_ids, out_scores, out_count);
}

extern "C" int hvx_batcher_wait(hvx_batcher *b, const hvx_batcher_ticket *ticket, uint32_t timeout_us, uint64_t *out_ids, float *out_scores,
                                uint32_t *out_count) {
    if (!b || !out_ids || !out_scores || !out_count || !ticket_ok(b, ticket)) return fail(HVX_ERR_INVARIANT, "null argument or malformed ticket");
    Inside in(b);
    *out_count = 0;
    const int rc = await_slot(b, ticket->sequence, ticket->slot, timeout_us == 0xFFFFFFFFu ? -1 : (long)timeout_us);
    if (rc) return rc; // HVX_PENDING on time-out: the ticket stays valid
    return take_result(b, ticket->sequence, ticket->slot, out_ids, out_scores, out_count);
}

// The image behind the batcher has grown (hvx_index_insert_batch on the handle the batcher was created from): every dispatcher lane
// adopts the new generation before the next batch it launches (batches already in flight finish on the generation they started with).
extern "C" int hvx_batcher_refresh(hvx_batcher *b) {
    if (!b) return fail(HVX_ERR_INVARIANT, "null argument");
    b->refresh_gen.fetch_add(1, std::memory_order_acq_rel);
    return HVX_OK;
}

extern "C" int hvx_batcher_eventfd(hvx_batcher *b) {
    if (!b) { (void)fail(HVX_ERR_INVARIANT, "null argument"); return -1; }
    int fd = b->efd.load(std::memory_order_acquire);
    if (fd >= 0) return fd;
    const int made = eventfd(0, EFD_NONBLOCK | EFD_CLOEXEC);
    if (made < 0) { (void)fail(HVX_ERR_DEVICE, "eventfd failed"); return -1; }
    int expected = -1;
    if (!b->efd.compare_exchange_strong(expected, made)) { // another thread created it first
        (void)close(made);
        return expected;
    }
    return made;
}

extern "C" int hvx_batcher_stats(const hvx_batcher *cb, uint64_t *batches, uint64_t *queries, uint64_t *full_batches) {
    if (!cb) return fail(HVX_ERR_INVARIANT, "null argument");
    if (batches) *batches = cb->n_batches.load();
    if (queries) *queries = cb->n_queries.load();
    if (full_batches) *full_batches = cb->n_full.load();
    return HVX_OK;
}

extern "C" int hvx_batcher_lane_times(const hvx_batcher *cb, hvx_batcher_times *out) {
    if (!cb || !out) return fail(HVX_ERR_INVARIANT, "null argument");
    memset(out, 0, sizeof(*out));
    for (const Lane &ln : cb->lanes) {
        out->idle_ns += ln.ns_idle.load();
        out->collect_ns += ln.ns_collect.load();
        out->drain_ns += ln.ns_drain.load();
        out->fill_ns += ln.ns_fill.load();
        out->device_ns += ln.ns_device.load();
        out->wake_ns += ln.ns_wake.load();
    }
    out->lanes = (uint32_t)cb->lanes.size();
    return HVX_OK;
}
