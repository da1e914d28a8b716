// hvx_prefilter.hip -- graph prefilter on the device (CSR BFS / one-hop expand -> candidate bitmap)
// and the restricted (prefiltered) vector search entry point.
//
// Reference: crates/graph-algorithms/src/model.rs:370-417 (Csr), algorithms/traversal.rs:216-318
// (breadth_first, suppresses_hub), crates/db/src/execution/interpreter/access/expand.rs:16-80,
// crates/db/src/search/vector/restricted.rs:196-260,303-371,426-453,529-613,753-835.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "hvx_host.h"

using namespace hvx;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(HVX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

struct hvx_csr {
    int device = 0;
    uint32_t n = 0;
    uint64_t e = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::vector<void *> allocs;
    // outgoing and incoming adjacency (crates/graph-algorithms/src/model.rs:376-417 builds both)
    uint64_t *out_off = nullptr, *in_off = nullptr;
    uint32_t *out_tgt = nullptr, *in_tgt = nullptr;
    uint32_t *out_lab = nullptr, *in_lab = nullptr; // null when the graph is unlabeled
    uint32_t *visited = nullptr;                    // [words] bitmap
    uint32_t *depth = nullptr;                      // [n]
    uint32_t *front[2] = {nullptr, nullptr};        // frontier queues [n]
    uint32_t *counter = nullptr;                    // [2] next-frontier size
    uint32_t *labels = nullptr;                     // allowed-label scratch
    uint32_t labels_cap = 0;
};

namespace {

struct CsrView {
    const uint64_t *out_off, *in_off;
    const uint32_t *out_tgt, *in_tgt, *out_lab, *in_lab;
    uint32_t n;
};

__device__ __forceinline__ bool label_ok(uint32_t lab, const uint32_t *allowed, uint32_t n_allowed) {
    if (n_allowed == 0) return true; // empty allow-set means every label (traversal.rs:53)
    for (uint32_t i = 0; i < n_allowed; ++i)
        if (allowed[i] == lab) return true;
    return false;
}

// One BFS level: one wavefront per frontier node, lanes stride over its arcs (out, then in).
// mark_visited=false gives the `expand` union (no visited exclusion, bitmap only).
__global__ __launch_bounds__(256) void bfs_level_kernel(CsrView g, const uint32_t *frontier, uint32_t n_front,
                                                        uint32_t *next, uint32_t *next_n, uint32_t *visited,
                                                        uint32_t *depth, uint32_t next_depth, uint32_t direction,
                                                        const uint32_t *allowed, uint32_t n_allowed,
                                                        uint32_t hub_degree, uint32_t is_seed_level,
                                                        uint32_t expand_only) {
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = (int)(threadIdx.x & 63u);
    if (w >= n_front) return;
    const uint32_t node = frontier[w];
    const uint64_t o0 = g.out_off[node], o1 = g.out_off[node + 1];
    const uint64_t i0 = g.in_off[node], i1 = g.in_off[node + 1];
    // suppresses_hub (traversal.rs:311-318): non-seed nodes at/above the total-degree threshold are
    // emitted but never expanded
    if (hub_degree && !is_seed_level && (o1 - o0) + (i1 - i0) >= hub_degree) return;
    for (int pass = 0; pass < 2; ++pass) {
        const bool use_out = pass == 0;
        if (use_out && direction == 1u) continue;  // In only
        if (!use_out && direction == 0u) continue; // Out only
        const uint64_t a0 = use_out ? o0 : i0, a1 = use_out ? o1 : i1;
        const uint32_t *tgt = use_out ? g.out_tgt : g.in_tgt;
        const uint32_t *lab = use_out ? g.out_lab : g.in_lab;
        for (uint64_t a = a0 + (uint64_t)lane; a < a1; a += 64) {
            if (lab && !label_ok(lab[a], allowed, n_allowed)) continue;
            const uint32_t v = tgt[a];
            const uint32_t bit = 1u << (v & 31u);
            const uint32_t old = atomicOr(&visited[v >> 5], bit);
            if (expand_only) continue;
            if (!(old & bit)) {
                depth[v] = next_depth;
                next[atomicAdd(next_n, 1u)] = v;
            }
        }
    }
}

__global__ void bfs_seed_kernel(const uint32_t *seeds, uint32_t n_seeds, uint32_t *visited, uint32_t *depth) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seeds) return;
    const uint32_t v = seeds[i];
    atomicOr(&visited[v >> 5], 1u << (v & 31u));
    depth[v] = 0;
}

__global__ void clear_bits_kernel(const uint32_t *ids, uint32_t n, uint32_t *bitmap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    atomicAnd(&bitmap[ids[i] >> 5], ~(1u << (ids[i] & 31u)));
}

int csr_alloc(hvx_csr *g, void **p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return fail(HVX_ERR_DEVICE, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    g->allocs.push_back(*p);
    return HVX_OK;
}

void csr_free(hvx_csr *g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    for (void *p : g->allocs) (void)hipFree(p);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

} // namespace

extern "C" void hvx_csr_free(hvx_csr *g) { csr_free(g); }

extern "C" int hvx_csr_import(uint64_t n_nodes, uint64_t n_edges, const uint64_t *out_offsets,
                              const uint64_t *out_targets, const uint32_t *edge_labels, int32_t device,
                              hvx_csr **out) {
    if (!out) return fail(HVX_ERR_INVARIANT, "null argument");
    *out = nullptr;
    if (n_nodes >= (1ull << 32) - 1) return fail(HVX_ERR_UNSUPPORTED, "graph too large (n < 2^32-1)");
    if (n_nodes && !out_offsets) return fail(HVX_ERR_INVARIANT, "null offsets");
    if (n_edges && !out_targets) return fail(HVX_ERR_INVARIANT, "null targets");
    if (n_nodes && (out_offsets[0] != 0 || out_offsets[n_nodes] != n_edges))
        return fail(HVX_ERR_INVARIANT, "offsets do not span the edge array");
    int dev = device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipSetDevice(dev));
    // incoming CSR by counting sort (stable: incoming arcs keep source order)
    std::vector<uint64_t> in_off(n_nodes + 1, 0);
    std::vector<uint32_t> tgt32(std::max<uint64_t>(n_edges, 1)), in_tgt(std::max<uint64_t>(n_edges, 1)),
        in_lab(edge_labels ? std::max<uint64_t>(n_edges, 1) : 0);
    for (uint64_t u = 0; u < n_nodes; ++u) {
        if (out_offsets[u + 1] < out_offsets[u]) return fail(HVX_ERR_INVARIANT, "offsets not monotone");
        for (uint64_t a = out_offsets[u]; a < out_offsets[u + 1]; ++a) {
            if (out_targets[a] >= n_nodes) return fail(HVX_ERR_INVARIANT, "edge target out of range");
            tgt32[a] = (uint32_t)out_targets[a];
            in_off[out_targets[a] + 1]++;
        }
    }
    for (uint64_t v = 0; v < n_nodes; ++v) in_off[v + 1] += in_off[v];
    {
        std::vector<uint64_t> cur(in_off.begin(), in_off.end() - 1);
        for (uint64_t u = 0; u < n_nodes; ++u)
            for (uint64_t a = out_offsets[u]; a < out_offsets[u + 1]; ++a) {
                uint64_t slot = cur[out_targets[a]]++;
                in_tgt[slot] = (uint32_t)u;
                if (edge_labels) in_lab[slot] = edge_labels[a];
            }
    }
    hvx_csr *g = new hvx_csr();
    g->device = dev;
    g->n = (uint32_t)n_nodes;
    g->e = n_edges;
    auto bail = [&](int code) { csr_free(g); return code; };
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(HVX_ERR_DEVICE, "stream creation failed"));
    int rc;
    auto up = [&](const void *src, size_t bytes, void **dst) -> int {
        int r = csr_alloc(g, dst, bytes);
        if (r) return r;
        if (bytes && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(HVX_ERR_DEVICE, "upload failed");
        return HVX_OK;
    };
    std::vector<uint64_t> zero_off(1, 0);
    if ((rc = up(n_nodes ? out_offsets : zero_off.data(), (n_nodes + 1) * 8, (void **)&g->out_off))) return bail(rc);
    if ((rc = up(in_off.data(), (n_nodes + 1) * 8, (void **)&g->in_off))) return bail(rc);
    if ((rc = up(tgt32.data(), n_edges * 4, (void **)&g->out_tgt))) return bail(rc);
    if ((rc = up(in_tgt.data(), n_edges * 4, (void **)&g->in_tgt))) return bail(rc);
    if (edge_labels) {
        if ((rc = up(edge_labels, n_edges * 4, (void **)&g->out_lab))) return bail(rc);
        if ((rc = up(in_lab.data(), n_edges * 4, (void **)&g->in_lab))) return bail(rc);
    }
    const size_t words = ((n_nodes + 63) / 64) * 2;
    if ((rc = csr_alloc(g, (void **)&g->visited, std::max<size_t>(words, 2) * 4))) return bail(rc);
    if ((rc = csr_alloc(g, (void **)&g->depth, std::max<uint64_t>(n_nodes, 1) * 4))) return bail(rc);
    if ((rc = csr_alloc(g, (void **)&g->front[0], std::max<uint64_t>(n_nodes, 1) * 4))) return bail(rc);
    if ((rc = csr_alloc(g, (void **)&g->front[1], std::max<uint64_t>(n_nodes, 1) * 4))) return bail(rc);
    if ((rc = csr_alloc(g, (void **)&g->counter, 16))) return bail(rc);
    *out = g;
    return HVX_OK;
}

static int upload_labels(hvx_csr *g, const uint32_t *labels, uint32_t n) {
    if (n > g->labels_cap) {
        int rc = csr_alloc(g, (void **)&g->labels, (size_t)n * 4);
        if (rc) return rc;
        g->labels_cap = n;
    }
    if (n) HIP_TRY(hipMemcpyAsync(g->labels, labels, (size_t)n * 4, hipMemcpyHostToDevice, g->stream));
    return HVX_OK;
}

// caller holds g->mu; the visited bitmap stays in g->visited (device) whether or not it is copied out
static int run_bfs_locked(hvx_csr *g, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth, uint32_t direction,
                          const uint32_t *labels, uint32_t n_labels, uint32_t hub_degree, uint32_t include_seeds,
                          bool expand_only, uint64_t *out_bitmap, uint32_t *out_depth) {
    if (direction > HVX_DIR_BOTH) return fail(HVX_ERR_INVARIANT, "bad direction");
    if (n_seeds == 0) return fail(HVX_ERR_INVARIANT, "traversal requires at least one seed"); // traversal.rs:198-202
    HIP_TRY(hipSetDevice(g->device));
    // dedupe seeds preserving first occurrence (traversal.rs:203-210); unknown node => error
    std::vector<uint32_t> s32;
    {
        std::vector<uint64_t> seen(seeds, seeds + n_seeds);
        std::sort(seen.begin(), seen.end());
        seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
        for (uint64_t v : seen) {
            if (v >= g->n) return fail(HVX_ERR_INVARIANT, "unknown node %llu", (unsigned long long)v);
            s32.push_back((uint32_t)v);
        }
    }
    const size_t words32 = ((g->n + 63) / 64) * 2;
    int rc = upload_labels(g, labels, n_labels);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(g->visited, 0, std::max<size_t>(words32, 2) * 4, g->stream));
    HIP_TRY(hipMemsetAsync(g->depth, 0xFF, std::max<uint32_t>(g->n, 1) * 4, g->stream));
    HIP_TRY(hipMemcpyAsync(g->front[0], s32.data(), s32.size() * 4, hipMemcpyHostToDevice, g->stream));
    uint32_t nf = (uint32_t)s32.size();
    if (!expand_only)
        hipLaunchKernelGGL(bfs_seed_kernel, dim3((nf + 255) / 256), dim3(256), 0, g->stream, g->front[0], nf, g->visited, g->depth);
    CsrView v{g->out_off, g->in_off, g->out_tgt, g->in_tgt, g->out_lab, g->in_lab, g->n};
    int cur = 0;
    const uint32_t levels = expand_only ? 1u : max_depth;
    for (uint32_t d = 0; d < levels && nf; ++d) {
        HIP_TRY(hipMemsetAsync(g->counter, 0, 4, g->stream));
        const uint32_t blocks = (uint32_t)(((uint64_t)nf * 64 + 255) / 256);
        hipLaunchKernelGGL(bfs_level_kernel, dim3(blocks), dim3(256), 0, g->stream, v, g->front[cur], nf,
                           g->front[cur ^ 1], g->counter, g->visited, g->depth, d + 1, direction, g->labels,
                           n_labels, hub_degree, d == 0 ? 1u : 0u, expand_only ? 1u : 0u);
        HIP_TRY(hipGetLastError());
        uint32_t next_n = 0;
        HIP_TRY(hipMemcpyAsync(&next_n, g->counter, 4, hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
        nf = next_n;
        cur ^= 1;
    }
    if (!expand_only && !include_seeds) {
        // seeds were inserted into front[0]; re-upload (front[0] may have been reused) and clear
        HIP_TRY(hipMemcpyAsync(g->front[cur ^ 1], s32.data(), s32.size() * 4, hipMemcpyHostToDevice, g->stream));
        hipLaunchKernelGGL(clear_bits_kernel, dim3(((uint32_t)s32.size() + 255) / 256), dim3(256), 0, g->stream,
                           g->front[cur ^ 1], (uint32_t)s32.size(), g->visited);
    }
    if (out_bitmap) HIP_TRY(hipMemcpyAsync(out_bitmap, g->visited, words32 * 4, hipMemcpyDeviceToHost, g->stream));
    if (out_depth && g->n) HIP_TRY(hipMemcpyAsync(out_depth, g->depth, (size_t)g->n * 4, hipMemcpyDeviceToHost, g->stream));
    HIP_TRY(hipStreamSynchronize(g->stream));
    return HVX_OK;
}

static int run_bfs(hvx_csr *g, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth, uint32_t direction,
                   const uint32_t *labels, uint32_t n_labels, uint32_t hub_degree, uint32_t include_seeds,
                   bool expand_only, uint64_t *out_bitmap, uint32_t *out_depth) {
    std::lock_guard<std::mutex> lock(g->mu);
    return run_bfs_locked(g, seeds, n_seeds, max_depth, direction, labels, n_labels, hub_degree, include_seeds, expand_only,
                          out_bitmap, out_depth);
}

extern "C" int hvx_traverse_filter(const hvx_csr *cg, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth,
                                   uint32_t direction, const uint32_t *allowed_label_ids, uint32_t n_labels,
                                   uint32_t hub_degree, uint32_t include_seeds, uint64_t *out_bitmap_words,
                                   uint32_t *out_depth) {
    if (!cg) return fail(HVX_ERR_INVARIANT, "null graph");
    return run_bfs(const_cast<hvx_csr *>(cg), seeds, n_seeds, max_depth, direction, allowed_label_ids, n_labels,
                   hub_degree, include_seeds, false, out_bitmap_words, out_depth);
}

extern "C" int hvx_expand_filter(const hvx_csr *cg, const uint64_t *rows, uint32_t n_rows, uint32_t direction,
                                 const uint32_t *allowed_label_ids, uint32_t n_labels, uint64_t *out_bitmap_words) {
    if (!cg) return fail(HVX_ERR_INVARIANT, "null graph");
    if (n_rows == 0) { // empty input stream expands to nothing
        const hvx_csr *g = cg;
        memset(out_bitmap_words, 0, ((g->n + 63) / 64) * 8);
        return HVX_OK;
    }
    return run_bfs(const_cast<hvx_csr *>(cg), rows, n_rows, 1, direction, allowed_label_ids, n_labels, 0, 0, true,
                   out_bitmap_words, nullptr);
}

// ---------------------------------------------------------------------------------------------
// Restricted (prefiltered) search: search_restricted_observed_with_beam_percent
// (restricted.rs:529-613).  Plan `Exact` is the reference's restricted_exact_scan; candidate sets
// above the exact thresholds are ALSO scanned exactly on the device (a gathered flat scan over the
// allowed rows): on MI355X a 100k x 1536 f32 candidate set is 0.6 GB = ~0.1 ms of HBM, and the
// result is the exact answer that the reference's ACORN-style walk (restricted.rs:837-1148)
// approximates (recall gate >= 0.92/0.95).  See DESIGN.md "restricted search".
// ---------------------------------------------------------------------------------------------
static int restricted_one(hvx_index *ix, const float *queries, uint32_t b, uint32_t k, const uint64_t *allowed,
                          uint64_t n_allowed, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                          uint32_t *out_status, hvx_stats *stats) {
    // RestrictedVectorCandidates::from_ids (restricted.rs:356-371): dedupe, cap 1,000,000
    std::vector<uint64_t> ids(allowed, allowed + n_allowed);
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    if (ids.size() > 1000000) return fail(HVX_ERR_CANDIDATE_LIMIT, "restricted vector search accepts at most 1000000 unique candidates");
    for (uint32_t q = 0; q < b; ++q) {
        out_counts[q] = 0;
        if (out_status) out_status[q] = HVX_OK;
    }
    if (ids.empty()) return HVX_OK; // RestrictedVectorCandidates::Empty => no results, before any validation
    // RestrictedResultCount::try_new (restricted.rs:200-213)
    const uint32_t kk = (uint32_t)std::min<uint64_t>(k, ids.size());
    if (kk == 0) return fail(HVX_ERR_K_RANGE, "result count must be non-zero");
    if (kk > 800) return fail(HVX_ERR_K_RANGE, "restricted vector search result count %u is above the maximum 800", kk);
    // ids that are not indexed are omitted (restricted.rs:615-659; tests .../restricted.rs:788-800)
    std::vector<uint32_t> subset;
    subset.reserve(ids.size());
    for (uint64_t id : ids) {
        uint32_t x = ix->find(id);
        if (x != kSentinel) subset.push_back(x);
    }
    // query validation happens before the empty-index / empty-subset outcome (restricted.rs:556-567)
    int rc;
    if (subset.size() > ix->cap_subset) {
        if ((rc = ix->regrow((void **)&ix->f_subset, subset.size() * 4))) return rc;
        ix->cap_subset = subset.size();
    }
    if (!subset.empty())
        HIP_TRY(hipMemcpyAsync(ix->f_subset, subset.data(), subset.size() * 4, hipMemcpyHostToDevice, ix->stream));
    std::vector<uint64_t> t_ids((size_t)b * kk);
    std::vector<float> t_sc((size_t)b * kk);
    std::vector<uint32_t> t_cnt(b), t_st(b);
    rc = flat_scan_host(ix, queries, b, kk, ix->f_subset, (uint32_t)subset.size(), t_ids.data(), t_sc.data(),
                        t_cnt.data(), t_st.data(), stats);
    if (rc) return rc;
    for (uint32_t q = 0; q < b; ++q) {
        if (t_st[q]) {
            if (!out_status) return fail((int)t_st[q], "query %u rejected with status %u", q, t_st[q]);
            out_status[q] = t_st[q];
            continue;
        }
        out_counts[q] = t_cnt[q];
        memcpy(out_ids + (size_t)q * k, t_ids.data() + (size_t)q * kk, (size_t)t_cnt[q] * 8);
        memcpy(out_scores + (size_t)q * k, t_sc.data() + (size_t)q * kk, (size_t)t_cnt[q] * 4);
    }
    return HVX_OK;
}

extern "C" int hvx_search_restricted_batch(const hvx_index *cix, const float *queries, uint32_t b, uint32_t k,
                                           uint32_t ef, const uint64_t *allowed_ids, const uint64_t *allowed_offsets,
                                           uint64_t n_allowed, uint64_t *out_ids, float *out_scores,
                                           uint32_t *out_counts, uint32_t *out_status, hvx_stats *stats) {
    if (!cix) return fail(HVX_ERR_INVARIANT, "null index");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    if (k == 0) return fail(HVX_ERR_K_RANGE, "result count must be non-zero");
    if (ef < k) return fail(HVX_ERR_K_RANGE, "search beam width %u is below the result count %u", ef, k);
    if (b == 0) return HVX_OK;
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    if (!allowed_offsets)
        return restricted_one(ix, queries, b, k, allowed_ids, n_allowed, out_ids, out_scores, out_counts, out_status, stats);
    for (uint32_t q = 0; q < b; ++q) {
        const uint64_t a0 = allowed_offsets[q], a1 = allowed_offsets[q + 1];
        int rc = restricted_one(ix, queries + (size_t)q * ix->dev.dim, 1, k, allowed_ids + a0, a1 - a0,
                                out_ids + (size_t)q * k, out_scores + (size_t)q * k, out_counts + q,
                                out_status ? out_status + q : nullptr, stats);
        if (rc) return rc;
    }
    return HVX_OK;
}

// ---------------------------------------------------------------------------------------------
// Fused prefilter + restricted kNN: the candidate bitmap never leaves the device.
// ---------------------------------------------------------------------------------------------
namespace {

// external node id -> internal row of the index (ids ascending), kSentinel when the node holds no vector
__device__ __forceinline__ uint32_t find_row(const uint64_t *ids, uint32_t n, uint64_t id, bool contiguous) {
    if (n == 0) return kSentinel;
    if (contiguous) return (id >= ids[0] && id - ids[0] < n) ? (uint32_t)(id - ids[0]) : kSentinel;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (ids[mid] < id) lo = mid + 1;
        else hi = mid;
    }
    return (lo < n && ids[lo] == id) ? lo : kSentinel;
}

// pass 1: per 256-word block, how many set bits map to an indexed row (restricted.rs:615-659: ids that are not
// indexed are omitted) and how many bits are set at all (the RestrictedVectorCandidates population, :356-371)
__global__ __launch_bounds__(256) void bitmap_count_kernel(const uint32_t *bitmap, uint32_t n_words, const uint64_t *ids, uint32_t n,
                                                           uint32_t contiguous, uint32_t *block_rows, uint32_t *total_bits) {
    __shared__ uint32_t s_rows, s_bits;
    if (threadIdx.x == 0) { s_rows = 0; s_bits = 0; }
    __syncthreads();
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    uint32_t word = w < n_words ? bitmap[w] : 0u, rows = 0;
    const uint32_t bits = (uint32_t)__builtin_popcount(word);
    while (word) {
        const uint32_t b = (uint32_t)__builtin_ctz(word);
        word &= word - 1u;
        rows += find_row(ids, n, (uint64_t)w * 32u + b, contiguous != 0u) != kSentinel ? 1u : 0u;
    }
    if (rows) atomicAdd(&s_rows, rows);
    if (bits) atomicAdd(&s_bits, bits);
    __syncthreads();
    if (threadIdx.x == 0) {
        block_rows[blockIdx.x] = s_rows;
        if (s_bits) atomicAdd(total_bits, s_bits);
    }
}

// exclusive scan of the block counts (one block; n_blocks <= a few thousand), total into block_rows[n_blocks]
__global__ __launch_bounds__(1024) void block_scan_kernel(uint32_t *block_rows, uint32_t n_blocks) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (n_blocks + 1023u) / 1024u, t = threadIdx.x;
    uint32_t sum = 0;
    for (uint32_t i = t * per; i < (t + 1) * per && i < n_blocks; ++i) sum += block_rows[i];
    part[t] = sum;
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < 1024; ++i) { const uint32_t v = part[i]; part[i] = run; run += v; }
        block_rows[n_blocks] = run;
    }
    __syncthreads();
    uint32_t run = part[t];
    for (uint32_t i = t * per; i < (t + 1) * per && i < n_blocks; ++i) { const uint32_t v = block_rows[i]; block_rows[i] = run; run += v; }
}

// pass 2: rows of the set bits, ascending id order (thread order inside a block by an LDS scan of the per-word counts)
__global__ __launch_bounds__(256) void bitmap_compact_kernel(const uint32_t *bitmap, uint32_t n_words, const uint64_t *ids, uint32_t n,
                                                             uint32_t contiguous, const uint32_t *block_base, uint32_t *subset) {
    __shared__ uint32_t cnt[256];
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    const uint32_t word0 = w < n_words ? bitmap[w] : 0u;
    uint32_t word = word0, rows = 0;
    while (word) {
        const uint32_t b = (uint32_t)__builtin_ctz(word);
        word &= word - 1u;
        rows += find_row(ids, n, (uint64_t)w * 32u + b, contiguous != 0u) != kSentinel ? 1u : 0u;
    }
    cnt[threadIdx.x] = rows;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) { // Hillis-Steele inclusive scan
        const uint32_t v = threadIdx.x >= off ? cnt[threadIdx.x - off] : 0u;
        __syncthreads();
        cnt[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t out = block_base[blockIdx.x] + cnt[threadIdx.x] - rows;
    word = word0;
    while (word) {
        const uint32_t b = (uint32_t)__builtin_ctz(word);
        word &= word - 1u;
        const uint32_t r = find_row(ids, n, (uint64_t)w * 32u + b, contiguous != 0u);
        if (r != kSentinel) subset[out++] = r;
    }
}

} // namespace

extern "C" int hvx_prefilter_search_batch(const hvx_index *cix, const hvx_csr *cg, const float *queries, uint32_t b, uint32_t k,
                                          uint32_t ef, uint32_t mode, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth,
                                          uint32_t direction, const uint32_t *allowed_label_ids, uint32_t n_labels,
                                          uint32_t hub_degree, uint32_t include_seeds, uint64_t *out_ids, float *out_scores,
                                          uint32_t *out_counts, uint32_t *out_status, uint64_t *out_candidates, hvx_stats *stats) {
    if (!cix || !cg) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    hvx_csr *g = const_cast<hvx_csr *>(cg);
    if (k == 0) return fail(HVX_ERR_K_RANGE, "result count must be non-zero");
    if (ef < k) return fail(HVX_ERR_K_RANGE, "search beam width %u is below the result count %u", ef, k);
    if (mode > HVX_PREFILTER_TRAVERSE) return fail(HVX_ERR_INVARIANT, "bad prefilter mode");
    if (ix->device != g->device) return fail(HVX_ERR_INVARIANT, "index and graph live on different devices");
    if (out_candidates) *out_candidates = 0;
    for (uint32_t q = 0; q < b; ++q) {
        out_counts[q] = 0;
        if (out_status) out_status[q] = HVX_OK;
    }
    if (b == 0) return HVX_OK;
    if (mode == HVX_PREFILTER_EXPAND && n_seeds == 0) return HVX_OK; // an empty stream expands to nothing
    std::lock_guard<std::mutex> glock(g->mu);
    std::lock_guard<std::mutex> lock(ix->mu);
    int rc = run_bfs_locked(g, seeds, n_seeds, mode == HVX_PREFILTER_EXPAND ? 1u : max_depth, direction, allowed_label_ids, n_labels,
                            mode == HVX_PREFILTER_EXPAND ? 0u : hub_degree, mode == HVX_PREFILTER_EXPAND ? 0u : include_seeds,
                            mode == HVX_PREFILTER_EXPAND, nullptr, nullptr);
    if (rc) return rc; // (run_bfs_locked ends with a stream synchronise: the bitmap is complete)
    const uint32_t n_words = ((g->n + 63u) / 64u) * 2u, n_blocks = (n_words + 255u) / 256u;
    if (n_blocks + 2 > ix->cap_pf_blocks) {
        if ((rc = ix->regrow((void **)&ix->pf_blocks, (size_t)(n_blocks + 2) * 4))) return rc;
        ix->cap_pf_blocks = n_blocks + 2;
    }
    uint32_t *d_total_bits = ix->pf_blocks + n_blocks + 1;
    HIP_TRY(hipMemsetAsync(d_total_bits, 0, 4, ix->stream));
    hipLaunchKernelGGL(bitmap_count_kernel, dim3(n_blocks), dim3(256), 0, ix->stream, g->visited, n_words, ix->dev.ids, ix->dev.n,
                       ix->contiguous ? 1u : 0u, ix->pf_blocks, d_total_bits);
    hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(1024), 0, ix->stream, ix->pf_blocks, n_blocks);
    HIP_TRY(hipGetLastError());
    uint32_t totals[2] = {0, 0}; // rows to scan, candidate population
    HIP_TRY(hipMemcpyAsync(totals, ix->pf_blocks + n_blocks, 8, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    const uint32_t n_rows = totals[0], population = totals[1];
    if (out_candidates) *out_candidates = population;
    // RestrictedVectorCandidates::from_ids (restricted.rs:356-371) / RestrictedResultCount::try_new (:200-213)
    if (population > 1000000) return fail(HVX_ERR_CANDIDATE_LIMIT, "restricted vector search accepts at most 1000000 unique candidates");
    if (population == 0) return HVX_OK;
    const uint32_t kk = std::min<uint32_t>(k, population);
    if (kk > 800) return fail(HVX_ERR_K_RANGE, "restricted vector search result count %u is above the maximum 800", kk);
    if (n_rows > ix->cap_subset) {
        if ((rc = ix->regrow((void **)&ix->f_subset, (size_t)n_rows * 4))) return rc;
        ix->cap_subset = n_rows;
    }
    if (n_rows)
        hipLaunchKernelGGL(bitmap_compact_kernel, dim3(n_blocks), dim3(256), 0, ix->stream, g->visited, n_words, ix->dev.ids, ix->dev.n,
                           ix->contiguous ? 1u : 0u, ix->pf_blocks, ix->f_subset);
    HIP_TRY(hipGetLastError());
    std::vector<uint64_t> t_ids((size_t)b * kk);
    std::vector<float> t_sc((size_t)b * kk);
    std::vector<uint32_t> t_cnt(b), t_st(b);
    rc = flat_scan_host(ix, queries, b, kk, ix->f_subset, n_rows, t_ids.data(), t_sc.data(), t_cnt.data(), t_st.data(), stats);
    if (rc) return rc;
    for (uint32_t q = 0; q < b; ++q) {
        if (t_st[q]) {
            if (!out_status) return fail((int)t_st[q], "query %u rejected with status %u", q, t_st[q]);
            out_status[q] = t_st[q];
            continue;
        }
        out_counts[q] = t_cnt[q];
        memcpy(out_ids + (size_t)q * k, t_ids.data() + (size_t)q * kk, (size_t)t_cnt[q] * 8);
        memcpy(out_scores + (size_t)q * k, t_sc.data() + (size_t)q * kk, (size_t)t_cnt[q] * 4);
    }
    return HVX_OK;
}
