// hvx_batcher.hip -- the batching operator in front of the search kernels.
//
// The reference calls `ValidatedVectorReadIndex::search` once per operator invocation, from many tokio tasks
// (crates/db/src/execution/interpreter/access/search/storage.rs:140-163): one query per call.  The device kernels are
// built for batches (one wavefront per query; 1 024 queries fill the chip), so concurrent single-query callers are
// coalesced here: a caller blocks in hvx_batcher_search, a dispatcher thread owned by the batcher gathers whatever is
// waiting -- up to max_batch queries, or after the oldest has waited max_wait_us -- into ONE hvx_search_batch_params
// launch and hands every caller its own rows of the result.  A batcher owns `lanes` dispatchers (default 2), each with its
// own execution lane on the shared index image (hvx_index_fork): batch i+1 is collected and launched while batch i is on
// the device, and nothing serialises on the caller's index handle.  Results are exactly those of a direct batch call
// (queries are independent; tests/test_gpu_parity.py::test_batcher_*).
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hvx_host.h"

using namespace hvx;

namespace {
struct Request {
    const float *query;
    uint64_t *out_ids;
    float *out_scores;
    uint32_t *out_count;
    int rc = 0;
    std::string err;
    std::chrono::steady_clock::time_point t_in;
    // completion is signalled to THIS caller only (no thundering herd across a thousand blocked callers)
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
};

// one dispatcher: its own execution lane (hvx_index_fork: stream + scratch) and host staging
struct Lane {
    hvx_index *ix = nullptr;
    std::thread worker;
    std::vector<float> q;
    std::vector<uint64_t> ids;
    std::vector<float> sc;
    std::vector<uint32_t> cnt, st;
};
} // namespace

struct hvx_batcher {
    hvx_search_params params{};
    uint32_t max_batch = 0, max_wait_us = 0, dim = 0;
    std::mutex mu;
    std::condition_variable cv_work;
    std::deque<Request *> pending;
    bool stop = false;
    std::vector<Lane> lanes;
    uint64_t n_batches = 0, n_queries = 0, n_full = 0;

    // Every lane runs this loop: while one lane's batch is on the device, another lane collects and launches the next
    // (double buffering; the launches overlap on the device as independent streams).
    void run(Lane &ln) {
        std::unique_lock<std::mutex> lock(mu);
        for (;;) {
            cv_work.wait(lock, [&] { return stop || !pending.empty(); });
            if (stop && pending.empty()) return;
            // wait for more callers: until the batch is full or the oldest request has waited max_wait_us
            const auto deadline = pending.front()->t_in + std::chrono::microseconds(max_wait_us);
            while (!stop && !pending.empty() && pending.size() < max_batch && std::chrono::steady_clock::now() < deadline)
                cv_work.wait_until(lock, deadline);
            if (pending.empty()) continue; // another lane took them
            std::vector<Request *> batch;
            while (!pending.empty() && batch.size() < max_batch) {
                batch.push_back(pending.front());
                pending.pop_front();
            }
            n_batches += 1;
            n_queries += batch.size();
            n_full += batch.size() == max_batch ? 1 : 0;
            if (!pending.empty()) cv_work.notify_one(); // leftovers: the next lane starts collecting now
            lock.unlock();
            const uint32_t b = (uint32_t)batch.size(), k = params.k;
            for (uint32_t i = 0; i < b; ++i) memcpy(ln.q.data() + (size_t)i * dim, batch[i]->query, (size_t)dim * 4);
            const int rc = hvx_search_batch_params(ln.ix, ln.q.data(), b, &params, ln.ids.data(), ln.sc.data(), ln.cnt.data(), ln.st.data(),
                                                   nullptr, nullptr, nullptr);
            const std::string err = rc ? hvx_last_error() : "";
            for (uint32_t i = 0; i < b; ++i) {
                Request *r = batch[i];
                r->rc = rc ? rc : (int)ln.st[i]; // a rejected query fails alone (per-query status)
                if (rc) r->err = err;
                else if (ln.st[i]) r->err = "query rejected with status " + std::to_string(ln.st[i]);
                else {
                    *r->out_count = ln.cnt[i];
                    memcpy(r->out_ids, ln.ids.data() + (size_t)i * k, (size_t)ln.cnt[i] * 8);
                    memcpy(r->out_scores, ln.sc.data() + (size_t)i * k, (size_t)ln.cnt[i] * 4);
                }
                {   // notify UNDER the request's lock: the request lives on its caller's stack, and a caller that sees `done`
                    // may return and destroy it -- it cannot do so before this scope releases r->m
                    std::lock_guard<std::mutex> g(r->m);
                    r->done = true;
                    r->cv.notify_one();
                }
            }
            lock.lock();
        }
    }
};

extern "C" void hvx_batcher_free(hvx_batcher *b) {
    if (!b) return;
    {
        std::lock_guard<std::mutex> lock(b->mu);
        b->stop = true;
    }
    b->cv_work.notify_all();
    for (Lane &ln : b->lanes)
        if (ln.worker.joinable()) ln.worker.join();
    for (Lane &ln : b->lanes)
        if (ln.ix) hvx_index_free(ln.ix);
    delete b;
}

extern "C" int hvx_batcher_new_lanes(hvx_index *ix, const hvx_search_params *params, uint32_t max_batch, uint32_t max_wait_us,
                                     uint32_t lanes, hvx_batcher **out) {
    if (!ix || !params || !out) return fail(HVX_ERR_INVARIANT, "null argument");
    *out = nullptr;
    if (max_batch == 0) max_batch = ix->max_batch;
    if (max_batch > ix->max_batch) return fail(HVX_ERR_UNSUPPORTED, "batch %u exceeds max_batch %u given at import", max_batch, ix->max_batch);
    if (lanes == 0) lanes = 2;
    if (lanes > 8) return fail(HVX_ERR_K_RANGE, "at most 8 dispatcher lanes");
    int rc = check_k_ef(params->k, params->ef);
    if (rc) return rc;
    hvx_batcher *b = new hvx_batcher();
    b->params = *params;
    b->max_batch = max_batch;
    b->max_wait_us = max_wait_us;
    b->dim = ix->dev.dim;
    b->lanes.resize(lanes);
    for (Lane &ln : b->lanes) {
        if ((rc = hvx_index_fork(ix, &ln.ix))) { // own stream + scratch on the shared image (SimHash rows included)
            hvx_batcher_free(b);
            return rc;
        }
        ln.q.resize((size_t)max_batch * b->dim);
        ln.ids.resize((size_t)max_batch * params->k);
        ln.sc.resize((size_t)max_batch * params->k);
        ln.cnt.resize(max_batch);
        ln.st.resize(max_batch);
    }
    for (Lane &ln : b->lanes) {
        Lane *lp = &ln;
        ln.worker = std::thread([b, lp] { b->run(*lp); });
    }
    *out = b;
    return HVX_OK;
}

extern "C" int hvx_batcher_new(hvx_index *ix, const hvx_search_params *params, uint32_t max_batch, uint32_t max_wait_us,
                               hvx_batcher **out) {
    return hvx_batcher_new_lanes(ix, params, max_batch, max_wait_us, 0, out);
}

extern "C" int hvx_batcher_search(hvx_batcher *b, const float *query, uint64_t *out_ids, float *out_scores, uint32_t *out_count) {
    if (!b || !query || !out_ids || !out_scores || !out_count) return fail(HVX_ERR_INVARIANT, "null argument");
    Request r;
    r.query = query;
    r.out_ids = out_ids;
    r.out_scores = out_scores;
    r.out_count = out_count;
    *out_count = 0;
    r.t_in = std::chrono::steady_clock::now();
    {
        std::lock_guard<std::mutex> lock(b->mu);
        if (b->stop) return fail(HVX_ERR_INVARIANT, "batcher is shutting down");
        b->pending.push_back(&r);
        if (b->pending.size() == 1 || b->pending.size() >= b->max_batch) b->cv_work.notify_one();
    }
    {
        std::unique_lock<std::mutex> g(r.m);
        r.cv.wait(g, [&] { return r.done; });
    }
    if (r.rc) return fail(r.rc, "%s", r.err.c_str());
    return HVX_OK;
}

extern "C" int hvx_batcher_stats(const hvx_batcher *cb, uint64_t *batches, uint64_t *queries, uint64_t *full_batches) {
    if (!cb) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_batcher *b = const_cast<hvx_batcher *>(cb);
    std::lock_guard<std::mutex> lock(b->mu);
    if (batches) *batches = b->n_batches;
    if (queries) *queries = b->n_queries;
    if (full_batches) *full_batches = b->n_full;
    return HVX_OK;
}
