// hvx_batcher.hip -- the batching operator in front of the search kernels.
//
// The reference calls `ValidatedVectorReadIndex::search` once per operator invocation, from many tokio tasks
// (crates/db/src/execution/interpreter/access/search/storage.rs:140-163): one query per call.  The device kernels are
// built for batches (one wavefront per query; 1 024 queries fill the chip), so concurrent single-query callers are
// coalesced here: a caller blocks in hvx_batcher_search, a dispatcher thread owned by the batcher gathers whatever is
// waiting -- up to max_batch queries, or after the oldest has waited max_wait_us -- into ONE hvx_search_batch_params
// launch and hands every caller its own rows of the result.  Results are exactly those of a direct batch call
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
    bool done = false;
    std::string err;
    std::chrono::steady_clock::time_point t_in;
};
} // namespace

struct hvx_batcher {
    hvx_index *ix = nullptr;
    hvx_search_params params{};
    uint32_t max_batch = 0, max_wait_us = 0, dim = 0;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Request *> pending;
    bool stop = false;
    std::thread worker;
    uint64_t n_batches = 0, n_queries = 0, n_full = 0;
    // staging reused by the dispatcher
    std::vector<float> q;
    std::vector<uint64_t> ids;
    std::vector<float> sc;
    std::vector<uint32_t> cnt, st;

    void run() {
        std::unique_lock<std::mutex> lock(mu);
        for (;;) {
            cv_work.wait(lock, [&] { return stop || !pending.empty(); });
            if (stop && pending.empty()) return;
            // wait for more callers: until the batch is full or the oldest request has waited max_wait_us
            const auto deadline = pending.front()->t_in + std::chrono::microseconds(max_wait_us);
            while (!stop && pending.size() < max_batch && std::chrono::steady_clock::now() < deadline)
                cv_work.wait_until(lock, deadline);
            std::vector<Request *> batch;
            while (!pending.empty() && batch.size() < max_batch) {
                batch.push_back(pending.front());
                pending.pop_front();
            }
            n_batches += 1;
            n_queries += batch.size();
            n_full += batch.size() == max_batch ? 1 : 0;
            lock.unlock();
            const uint32_t b = (uint32_t)batch.size(), k = params.k;
            for (uint32_t i = 0; i < b; ++i) memcpy(q.data() + (size_t)i * dim, batch[i]->query, (size_t)dim * 4);
            const int rc = hvx_search_batch_params(ix, q.data(), b, &params, ids.data(), sc.data(), cnt.data(), st.data(), nullptr,
                                                   nullptr, nullptr);
            const std::string err = rc ? hvx_last_error() : "";
            for (uint32_t i = 0; i < b; ++i) {
                Request *r = batch[i];
                r->rc = rc ? rc : (int)st[i]; // a rejected query fails alone (per-query status)
                if (rc) r->err = err;
                else if (st[i]) r->err = "query rejected with status " + std::to_string(st[i]);
                else {
                    *r->out_count = cnt[i];
                    memcpy(r->out_ids, ids.data() + (size_t)i * k, (size_t)cnt[i] * 8);
                    memcpy(r->out_scores, sc.data() + (size_t)i * k, (size_t)cnt[i] * 4);
                }
            }
            lock.lock();
            for (Request *r : batch) r->done = true;
            cv_done.notify_all();
        }
    }
};

extern "C" int hvx_batcher_new(hvx_index *ix, const hvx_search_params *params, uint32_t max_batch, uint32_t max_wait_us,
                               hvx_batcher **out) {
    if (!ix || !params || !out) return fail(HVX_ERR_INVARIANT, "null argument");
    *out = nullptr;
    if (max_batch == 0) max_batch = ix->max_batch;
    if (max_batch > ix->max_batch) return fail(HVX_ERR_UNSUPPORTED, "batch %u exceeds max_batch %u given at import", max_batch, ix->max_batch);
    int rc = check_k_ef(params->k, params->ef);
    if (rc) return rc;
    hvx_batcher *b = new hvx_batcher();
    b->ix = ix;
    b->params = *params;
    b->max_batch = max_batch;
    b->max_wait_us = max_wait_us;
    b->dim = ix->dev.dim;
    b->q.resize((size_t)max_batch * b->dim);
    b->ids.resize((size_t)max_batch * params->k);
    b->sc.resize((size_t)max_batch * params->k);
    b->cnt.resize(max_batch);
    b->st.resize(max_batch);
    b->worker = std::thread([b] { b->run(); });
    *out = b;
    return HVX_OK;
}

extern "C" void hvx_batcher_free(hvx_batcher *b) {
    if (!b) return;
    {
        std::lock_guard<std::mutex> lock(b->mu);
        b->stop = true;
    }
    b->cv_work.notify_all();
    if (b->worker.joinable()) b->worker.join();
    delete b;
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
    std::unique_lock<std::mutex> lock(b->mu);
    if (b->stop) return fail(HVX_ERR_INVARIANT, "batcher is shutting down");
    b->pending.push_back(&r);
    if (b->pending.size() == 1 || b->pending.size() >= b->max_batch) b->cv_work.notify_one();
    b->cv_done.wait(lock, [&] { return r.done; });
    lock.unlock();
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
