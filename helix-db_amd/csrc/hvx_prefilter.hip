// hvx_prefilter.hip -- graph prefilter on the device (CSR BFS / one-hop expand -> candidate bitmap)
// and the restricted (prefiltered) vector search entry point.
//
// Reference: crates/graph-algorithms/src/model.rs:370-417 (Csr), algorithms/traversal.rs:216-318
// (breadth_first, suppresses_hub), crates/db/src/execution/interpreter/access/expand.rs:16-80,
// crates/db/src/search/vector/restricted.rs:196-260,303-371,426-453,529-613,753-835.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <memory>
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
    // ordered traversal (hvx_traverse_ordered)
    bool rows_sorted = true;                        // every outgoing row ascends by target (the reference's rows do: model.rs:656-666)
    uint32_t *in_arc = nullptr;                     // [e] incoming slot -> index of the stored edge in the outgoing arrays (e < 2^32)
    uint32_t *owner = nullptr;                      // [n] frontier position that claimed the node at its level
    uint32_t *pc_out = nullptr, *pc_in = nullptr;   // [e] per arc: bit 31 = discovery arc, low bits = discovery arcs before it in its row
    uint32_t *cnt = nullptr;                        // [n + 1] per frontier position: discovery arcs (then their exclusive scan)
    uint32_t *ord_node = nullptr, *ord_parent = nullptr, *ord_arc = nullptr; // [n] visits in discovery order
    uint32_t *lvl = nullptr;                        // [4] level start, level size, next level size
    void *host = nullptr;                           // HostCsr mirror (hvx_traverse_dfs), fetched on first use
    uint32_t *h_seeds = nullptr;                    // pinned: the seeds of the traversal in flight (a pageable upload is a synchronous staged copy)
    size_t cap_h_seeds = 0;
    hipEvent_t done = nullptr;                      // recorded behind a traversal whose consumer runs on another stream (fused prefilter search)
    uint64_t max_out_deg = 0, max_in_deg = 0;       // largest row of either adjacency: bounds what one hop from s seeds can reach (round 6)
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

__device__ __forceinline__ void bfs_expand_node(const CsrView &g, uint32_t node, int lane, uint32_t *next, uint32_t *next_n,
                                                uint32_t *visited, uint32_t *depth, uint32_t next_depth, uint32_t direction,
                                                const uint32_t *allowed, uint32_t n_allowed, uint32_t hub_degree,
                                                uint32_t is_seed_level, uint32_t expand_only);

// One BFS level: one wavefront per frontier node, lanes stride over its arcs (out, then in).
// mark_visited=false gives the `expand` union (no visited exclusion, bitmap only).
__global__ __launch_bounds__(256) void bfs_level_kernel(CsrView g, const uint32_t *frontier, uint32_t n_front,
                                                        uint32_t *next, uint32_t *next_n, uint32_t *visited,
                                                        uint32_t *depth, uint32_t next_depth, uint32_t direction,
                                                        const uint32_t *allowed, uint32_t n_allowed,
                                                        uint32_t hub_degree, uint32_t is_seed_level,
                                                        uint32_t expand_only, const uint32_t *n_front_dev) {
    // n_front_dev != NULL: the frontier size is whatever the previous level's launch counted (no host round trip between
    // levels); the grid is then a fixed one and the wavefronts stride over the frontier
    if (n_front_dev) n_front = *n_front_dev;
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n_front; w += n_waves)
        bfs_expand_node(g, frontier[w], lane, next, next_n, visited, depth, next_depth, direction, allowed, n_allowed, hub_degree,
                        is_seed_level, expand_only);
}

__device__ __forceinline__ void bfs_expand_node(const CsrView &g, uint32_t node, int lane, uint32_t *next, uint32_t *next_n,
                                                uint32_t *visited, uint32_t *depth, uint32_t next_depth, uint32_t direction,
                                                const uint32_t *allowed, uint32_t n_allowed, uint32_t hub_degree,
                                                uint32_t is_seed_level, uint32_t expand_only) {
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

void host_csr_free(void *p); // the host mirror of hvx_traverse_dfs (defined with HostCsr below)

void csr_free(hvx_csr *g) {
    if (!g) return;
    host_csr_free(g->host);
    (void)hipSetDevice(g->device);
    for (void *p : g->allocs) (void)hipFree(p);
    if (g->h_seeds) (void)hipHostFree(g->h_seeds);
    if (g->done) (void)hipEventDestroy(g->done);
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
        in_lab(edge_labels ? std::max<uint64_t>(n_edges, 1) : 0), in_arc(n_edges < (1ull << 32) ? std::max<uint64_t>(n_edges, 1) : 0);
    bool rows_sorted = true;
    for (uint64_t u = 0; u < n_nodes; ++u) {
        if (out_offsets[u + 1] < out_offsets[u]) return fail(HVX_ERR_INVARIANT, "offsets not monotone");
        for (uint64_t a = out_offsets[u]; a < out_offsets[u + 1]; ++a) {
            if (out_targets[a] >= n_nodes) return fail(HVX_ERR_INVARIANT, "edge target out of range");
            tgt32[a] = (uint32_t)out_targets[a];
            if (a > out_offsets[u] && out_targets[a] < out_targets[a - 1]) rows_sorted = false;
            in_off[out_targets[a] + 1]++;
        }
    }
    uint64_t max_out = 0, max_in = 0;
    for (uint64_t u = 0; u < n_nodes; ++u) max_out = std::max<uint64_t>(max_out, out_offsets[u + 1] - out_offsets[u]);
    for (uint64_t v = 0; v < n_nodes; ++v) max_in = std::max<uint64_t>(max_in, in_off[v + 1]);
    for (uint64_t v = 0; v < n_nodes; ++v) in_off[v + 1] += in_off[v];
    {
        std::vector<uint64_t> cur(in_off.begin(), in_off.end() - 1);
        for (uint64_t u = 0; u < n_nodes; ++u)
            for (uint64_t a = out_offsets[u]; a < out_offsets[u + 1]; ++a) {
                uint64_t slot = cur[out_targets[a]]++;
                in_tgt[slot] = (uint32_t)u;
                if (n_edges < (1ull << 32)) in_arc[slot] = (uint32_t)a;
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
    g->rows_sorted = rows_sorted;
    g->max_out_deg = max_out;
    g->max_in_deg = max_in;
    if (!in_arc.empty() && (rc = up(in_arc.data(), n_edges * 4, (void **)&g->in_arc))) return bail(rc);
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
// `consumer`: a stream that reads the bitmap next (the fused prefilter search): it is made to wait on the traversal with an
// event instead of the host waiting for it.
static int run_bfs_locked(hvx_csr *g, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth, uint32_t direction,
                          const uint32_t *labels, uint32_t n_labels, uint32_t hub_degree, uint32_t include_seeds,
                          bool expand_only, uint64_t *out_bitmap, uint32_t *out_depth, hipStream_t consumer = nullptr) {
    if (direction > HVX_DIR_BOTH) return fail(HVX_ERR_INVARIANT, "bad direction");
    if (n_seeds == 0) return fail(HVX_ERR_INVARIANT, "traversal requires at least one seed"); // traversal.rs:198-202
    HIP_TRY(hipSetDevice(g->device));
    // seeds: unknown node => error; duplicates (traversal.rs:203-210 keeps the first occurrence) are harmless for a visited SET --
    // test-and-set makes every step idempotent -- so they are only removed when the list could overflow the frontier queue
    if (n_seeds > g->cap_h_seeds) {
        HIP_TRY(hipStreamSynchronize(g->stream)); // no upload from the old buffer is still in flight
        if (g->h_seeds) (void)hipHostFree(g->h_seeds);
        g->h_seeds = nullptr;
        g->cap_h_seeds = 0;
        const size_t want = std::max<size_t>((size_t)n_seeds + n_seeds / 2, 4096);
        if (hipHostMalloc((void **)&g->h_seeds, want * 4, hipHostMallocDefault) != hipSuccess) return fail(HVX_ERR_DEVICE, "hipHostMalloc(%zu) seeds", want * 4);
        g->cap_h_seeds = want;
    }
    struct SeedView { // what the code below reads of the former std::vector
        const uint32_t *p; size_t n;
        const uint32_t *data() const { return p; }
        size_t size() const { return n; }
    } s32{g->h_seeds, 0};
    if (n_seeds <= g->n) {
        for (uint32_t i = 0; i < n_seeds; ++i) {
            const uint64_t v = seeds[i];
            if (v >= g->n) return fail(HVX_ERR_INVARIANT, "unknown node %llu", (unsigned long long)v);
            g->h_seeds[i] = (uint32_t)v;
        }
        s32.n = n_seeds;
    } else {
        std::vector<uint64_t> seen(seeds, seeds + n_seeds);
        std::sort(seen.begin(), seen.end());
        seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
        for (uint64_t v : seen) {
            if (v >= g->n) return fail(HVX_ERR_INVARIANT, "unknown node %llu", (unsigned long long)v);
            g->h_seeds[s32.n++] = (uint32_t)v;
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
    // Level 0 is sized by the host (it knows the seeds); every later level reads its frontier size from the counter the
    // level before it incremented -- the levels are enqueued back to back, the host looks at the counter only every
    // kLevelsPerSync levels (to stop an unbounded traversal whose frontier has emptied).
    constexpr uint32_t kLevelsPerSync = 8;
    HIP_TRY(hipMemsetAsync(g->counter, 0, 8, g->stream));
    for (uint32_t d = 0; d < levels && nf;) {
        const uint32_t until = std::min<uint64_t>(levels, (uint64_t)d + kLevelsPerSync);
        for (; d < until; ++d) {
            uint32_t *cnt_next = g->counter + ((d + 1) & 1u);
            HIP_TRY(hipMemsetAsync(cnt_next, 0, 4, g->stream));
            const uint32_t blocks = d == 0 ? (uint32_t)(((uint64_t)nf * 64 + 255) / 256) : 1024u;
            hipLaunchKernelGGL(bfs_level_kernel, dim3(blocks), dim3(256), 0, g->stream, v, g->front[cur], nf,
                               g->front[cur ^ 1], cnt_next, g->visited, g->depth, d + 1, direction, g->labels,
                               n_labels, hub_degree, d == 0 ? 1u : 0u, expand_only ? 1u : 0u,
                               d == 0 ? (const uint32_t *)nullptr : (const uint32_t *)(g->counter + (d & 1u)));
            HIP_TRY(hipGetLastError());
            cur ^= 1;
        }
        if (d < levels) { // more levels wanted: is there a frontier left?
            uint32_t next_n = 0;
            HIP_TRY(hipMemcpyAsync(&next_n, g->counter + (d & 1u), 4, hipMemcpyDeviceToHost, g->stream));
            HIP_TRY(hipStreamSynchronize(g->stream));
            nf = next_n;
        }
    }
    if (!expand_only && !include_seeds) {
        // seeds were inserted into front[0]; re-upload (front[0] may have been reused) and clear
        HIP_TRY(hipMemcpyAsync(g->front[cur ^ 1], s32.data(), s32.size() * 4, hipMemcpyHostToDevice, g->stream));
        hipLaunchKernelGGL(clear_bits_kernel, dim3(((uint32_t)s32.size() + 255) / 256), dim3(256), 0, g->stream,
                           g->front[cur ^ 1], (uint32_t)s32.size(), g->visited);
    }
    if (out_bitmap) HIP_TRY(hipMemcpyAsync(out_bitmap, g->visited, words32 * 4, hipMemcpyDeviceToHost, g->stream));
    if (out_depth && g->n) HIP_TRY(hipMemcpyAsync(out_depth, g->depth, (size_t)g->n * 4, hipMemcpyDeviceToHost, g->stream));
    if (consumer && !out_bitmap && !out_depth) { // the bitmap stays on the device: order the consumer behind the traversal, the host moves on
        if (!g->done) HIP_TRY(hipEventCreateWithFlags(&g->done, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(g->done, g->stream));
        HIP_TRY(hipStreamWaitEvent(consumer, g->done, 0));
        return HVX_OK;
    }
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
// Ordered traversal: Graph::traverse (BreadthFirst) with its visit ORDER and discovery edges
// (traversal.rs:216-261).  The reference's FIFO order is: by the discovery position of the node that reached you, then
// by the arc order of that node (model.rs:635-725: rows sorted by neighbour, direction Both merges the two rows by
// neighbour, outgoing first on equal neighbours, incoming self-loops skipped).  Level-synchronous form of exactly that
// order, four launches per level, no host round trip between levels:
//   claim  -- every frontier position i offers itself to the unvisited nodes its allowed arcs reach: owner[v] = min i;
//   count  -- wave i walks its rows: an arc is v's discovery arc iff owner[v] == i and it is the first allowed arc to v
//             in the merged arc order; per arc it stores (flag, discovery arcs before it in its row), per position the total;
//   scan   -- exclusive prefix sum of the totals (one workgroup);
//   emit   -- a discovery arc's rank among its node's = (those before it in its own row) + (those of the other row with
//             a smaller neighbour: one binary search); visit slot = level start + scan[i] + rank.
// ---------------------------------------------------------------------------------------------
namespace {

constexpr uint32_t kUnset = 0xFFFFFFFFu;

struct OrdArgs {
    CsrView g;
    const uint32_t *in_arc;
    uint32_t *owner, *pc_out, *pc_in, *cnt, *depth;
    uint32_t *ord_node, *ord_parent, *ord_arc;
    uint32_t *lvl;               // [0] start of the current level in ord_*, [1] its size, [2] size of the next one
    const uint32_t *allowed;
    uint32_t n_allowed, direction, hub_degree, cur_depth;
};

__device__ __forceinline__ bool ord_expands(const OrdArgs &a, uint32_t node) {
    if (!a.hub_degree || a.cur_depth == 0) return true; // seeds are expanded whatever their degree
    return (a.g.out_off[node + 1] - a.g.out_off[node]) + (a.g.in_off[node + 1] - a.g.in_off[node]) < a.hub_degree;
}
__device__ __forceinline__ bool ord_allowed(const uint32_t *lab, uint64_t a, const OrdArgs &o) {
    return !lab || label_ok(lab[a], o.allowed, o.n_allowed);
}

__global__ __launch_bounds__(256) void ord_claim_kernel(OrdArgs a) {
    const uint32_t start = a.lvl[0], nf = a.lvl[1];
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < nf; i += n_waves) {
        const uint32_t node = a.ord_node[start + i];
        if (!ord_expands(a, node)) continue;
        for (int pass = 0; pass < 2; ++pass) {
            const bool use_out = pass == 0;
            if (use_out ? a.direction == 1u : a.direction == 0u) continue;
            const uint64_t a0 = use_out ? a.g.out_off[node] : a.g.in_off[node], a1 = use_out ? a.g.out_off[node + 1] : a.g.in_off[node + 1];
            const uint32_t *tgt = use_out ? a.g.out_tgt : a.g.in_tgt;
            const uint32_t *lab = use_out ? a.g.out_lab : a.g.in_lab;
            for (uint64_t e = a0 + (uint64_t)lane; e < a1; e += 64) {
                if (!ord_allowed(lab, e, a)) continue;
                const uint32_t v = tgt[e];
                if (a.depth[v] == kUnset) atomicMin(&a.owner[v], i);
            }
        }
    }
}

// is arc e (a row of `tgt`, starting at row0) the first ALLOWED arc to its neighbour in its row?  (rows ascend by neighbour)
__device__ __forceinline__ bool ord_first_in_row(const uint32_t *tgt, const uint32_t *lab, uint64_t row0, uint64_t e, const OrdArgs &o) {
    const uint32_t v = tgt[e];
    for (uint64_t p = e; p > row0 && tgt[p - 1] == v; --p)
        if (ord_allowed(lab, p - 1, o)) return false;
    return true;
}
// does the outgoing row [o0, o1) hold an allowed arc to v?
__device__ __forceinline__ bool ord_out_reaches(const OrdArgs &o, uint64_t o0, uint64_t o1, uint32_t v) {
    uint64_t lo = o0, hi = o1;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (o.g.out_tgt[mid] < v) lo = mid + 1; else hi = mid;
    }
    for (; lo < o1 && o.g.out_tgt[lo] == v; ++lo)
        if (ord_allowed(o.g.out_lab, lo, o)) return true;
    return false;
}

__global__ __launch_bounds__(256) void ord_count_kernel(OrdArgs a) {
    const uint32_t start = a.lvl[0], nf = a.lvl[1];
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < nf; i += n_waves) {
        const uint32_t node = a.ord_node[start + i];
        uint32_t total = 0;
        if (ord_expands(a, node)) {
            const uint64_t o0 = a.g.out_off[node], o1 = a.g.out_off[node + 1];
            for (int pass = 0; pass < 2; ++pass) {
                const bool use_out = pass == 0;
                if (use_out ? a.direction == 1u : a.direction == 0u) continue;
                const uint64_t a0 = use_out ? o0 : a.g.in_off[node], a1 = use_out ? o1 : a.g.in_off[node + 1];
                const uint32_t *tgt = use_out ? a.g.out_tgt : a.g.in_tgt;
                const uint32_t *lab = use_out ? a.g.out_lab : a.g.in_lab;
                uint32_t *pc = use_out ? a.pc_out : a.pc_in;
                uint32_t row_count = 0;
                for (uint64_t base = a0; base < a1; base += 64) {
                    const uint64_t e = base + (uint64_t)lane;
                    bool flag = false;
                    if (e < a1 && ord_allowed(lab, e, a)) {
                        const uint32_t v = tgt[e];
                        flag = a.depth[v] == kUnset && a.owner[v] == i && ord_first_in_row(tgt, lab, a0, e, a) &&
                               (use_out || a.direction != 2u || !ord_out_reaches(a, o0, o1, v));
                    }
                    const unsigned long long m = __ballot(flag);
                    if (e < a1) pc[e] = (flag ? 0x80000000u : 0u) | (row_count + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)));
                    row_count += (uint32_t)__popcll(m);
                }
                total += row_count;
            }
        }
        if (lane == 0) a.cnt[i] = total;
    }
}

// exclusive scan of cnt[0 .. lvl[1]) in place; lvl[2] = the total (the next level's size)
__global__ __launch_bounds__(1024) void ord_scan_kernel(uint32_t *cnt, uint32_t *lvl) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry_s;
    const uint32_t nf = lvl[1];
    const int tid = (int)threadIdx.x;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nf; base += 1024) {
        const uint32_t idx = base + (uint32_t)tid;
        const uint32_t v = idx < nf ? cnt[idx] : 0u;
        part[tid] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan
            const uint32_t t = tid >= off ? part[tid - off] : 0u;
            __syncthreads();
            part[tid] += t;
            __syncthreads();
        }
        const uint32_t carry = carry_s;
        if (idx < nf) cnt[idx] = carry + part[tid] - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + part[1023];
        __syncthreads();
    }
    if (tid == 0) lvl[2] = carry_s;
}

__global__ __launch_bounds__(256) void ord_emit_kernel(OrdArgs a) {
    const uint32_t start = a.lvl[0], nf = a.lvl[1];
    const uint32_t next_start = start + nf;
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < nf; i += n_waves) {
        const uint32_t node = a.ord_node[start + i];
        if (!ord_expands(a, node)) continue;
        const uint32_t base = a.cnt[i];
        const uint64_t o0 = a.g.out_off[node], o1 = a.g.out_off[node + 1], i0 = a.g.in_off[node], i1 = a.g.in_off[node + 1];
        const bool both = a.direction == 2u;
        if (a.direction != 1u)
            for (uint64_t e = o0 + (uint64_t)lane; e < o1; e += 64) {
                const uint32_t p = a.pc_out[e];
                if (!(p & 0x80000000u)) continue;
                const uint32_t v = a.g.out_tgt[e];
                uint32_t rank = p & 0x7FFFFFFFu;
                if (both) { // + discovery arcs of the incoming row with a smaller neighbour
                    uint64_t lo = i0, hi = i1;
                    while (lo < hi) {
                        const uint64_t mid = (lo + hi) >> 1;
                        if (a.g.in_tgt[mid] < v) lo = mid + 1; else hi = mid;
                    }
                    // pc_in[lo] counts the flags before slot lo; past the row's end: every flag of the row
                    if (lo < i1) rank += a.pc_in[lo] & 0x7FFFFFFFu;
                    else if (i1 > i0) rank += (a.pc_in[i1 - 1] & 0x7FFFFFFFu) + (a.pc_in[i1 - 1] >> 31);
                }
                const uint32_t slot = next_start + base + rank;
                a.ord_node[slot] = v;
                a.ord_parent[slot] = node;
                a.ord_arc[slot] = (uint32_t)e;               // followed along the stored edge
                a.depth[v] = a.cur_depth + 1u;
            }
        if (a.direction != 0u)
            for (uint64_t e = i0 + (uint64_t)lane; e < i1; e += 64) {
                const uint32_t p = a.pc_in[e];
                if (!(p & 0x80000000u)) continue;
                const uint32_t v = a.g.in_tgt[e];
                uint32_t rank = p & 0x7FFFFFFFu;
                if (both) { // + discovery arcs of the outgoing row with a neighbour <= v (an equal one cannot be flagged too)
                    uint64_t lo = o0, hi = o1;
                    while (lo < hi) {
                        const uint64_t mid = (lo + hi) >> 1;
                        if (a.g.out_tgt[mid] <= v) lo = mid + 1; else hi = mid;
                    }
                    if (lo < o1) rank += a.pc_out[lo] & 0x7FFFFFFFu;
                    else if (o1 > o0) rank += (a.pc_out[o1 - 1] & 0x7FFFFFFFu) + (a.pc_out[o1 - 1] >> 31);
                }
                const uint32_t slot = next_start + base + rank;
                a.ord_node[slot] = v;
                a.ord_parent[slot] = node;
                a.ord_arc[slot] = a.in_arc[e] | 0x80000000u; // followed against the stored edge
                a.depth[v] = a.cur_depth + 1u;
            }
    }
}

// next level: start += size, size = what the scan counted
__global__ void ord_advance_kernel(uint32_t *lvl) {
    lvl[0] += lvl[1];
    lvl[1] = lvl[2];
    lvl[2] = 0;
}

__global__ void ord_seed_kernel(const uint32_t *seeds, uint32_t n_seeds, uint32_t *depth, uint32_t *lvl) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_seeds) depth[seeds[i]] = 0;
    if (i == 0) { lvl[0] = 0; lvl[1] = n_seeds; lvl[2] = 0; }
}

} // namespace

extern "C" int hvx_traverse_ordered(const hvx_csr *cg, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth,
                                    uint32_t direction, const uint32_t *allowed_label_ids, uint32_t n_labels,
                                    uint32_t hub_degree, uint64_t capacity, uint64_t *out_nodes, uint32_t *out_depths,
                                    uint64_t *out_parents, uint64_t *out_edges, uint32_t *out_against, uint64_t *out_count) {
    if (!cg || !out_count) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_csr *g = const_cast<hvx_csr *>(cg);
    *out_count = 0;
    if (direction > HVX_DIR_BOTH) return fail(HVX_ERR_INVARIANT, "bad direction");
    if (n_seeds == 0) return fail(HVX_ERR_INVARIANT, "traversal requires at least one seed"); // traversal.rs:198-202
    if (!g->rows_sorted) return fail(HVX_ERR_UNSUPPORTED, "ordered traversal needs outgoing rows sorted by target (model.rs:656-666)");
    if (g->e >= (1ull << 31)) return fail(HVX_ERR_UNSUPPORTED, "ordered traversal serves graphs below 2^31 edges");
    std::lock_guard<std::mutex> lock(g->mu);
    HIP_TRY(hipSetDevice(g->device));
    // seeds: duplicates collapse, first occurrence keeps its place (traversal.rs:203-210); unknown node => error
    std::vector<uint32_t> s32;
    {
        std::vector<uint64_t> seen;
        for (uint32_t i = 0; i < n_seeds; ++i) {
            if (seeds[i] >= g->n) return fail(HVX_ERR_INVARIANT, "unknown node %llu", (unsigned long long)seeds[i]);
            if (std::find(seen.begin(), seen.end(), seeds[i]) == seen.end()) {
                seen.push_back(seeds[i]);
                s32.push_back((uint32_t)seeds[i]);
            }
        }
    }
    int rc;
    if (!g->owner) {
        const size_t nn = std::max<uint32_t>(g->n, 1), ee = std::max<uint64_t>(g->e, 1);
        if ((rc = csr_alloc(g, (void **)&g->owner, nn * 4))) return rc;
        if ((rc = csr_alloc(g, (void **)&g->pc_out, ee * 4))) return rc;
        if ((rc = csr_alloc(g, (void **)&g->pc_in, ee * 4))) return rc;
        if ((rc = csr_alloc(g, (void **)&g->cnt, (nn + 1) * 4))) return rc;
        if ((rc = csr_alloc(g, (void **)&g->ord_node, nn * 4))) return rc;
        if ((rc = csr_alloc(g, (void **)&g->ord_parent, nn * 4))) return rc;
        if ((rc = csr_alloc(g, (void **)&g->ord_arc, nn * 4))) return rc;
        if ((rc = csr_alloc(g, (void **)&g->lvl, 16))) return rc;
    }
    if ((rc = upload_labels(g, allowed_label_ids, n_labels))) return rc;
    const uint32_t nn = std::max<uint32_t>(g->n, 1);
    HIP_TRY(hipMemsetAsync(g->depth, 0xFF, (size_t)nn * 4, g->stream));
    HIP_TRY(hipMemsetAsync(g->owner, 0xFF, (size_t)nn * 4, g->stream));
    HIP_TRY(hipMemcpyAsync(g->ord_node, s32.data(), s32.size() * 4, hipMemcpyHostToDevice, g->stream));
    const uint32_t ns = (uint32_t)s32.size();
    hipLaunchKernelGGL(ord_seed_kernel, dim3((ns + 255) / 256), dim3(256), 0, g->stream, g->ord_node, ns, g->depth, g->lvl);
    HIP_TRY(hipGetLastError());
    OrdArgs a;
    a.g = CsrView{g->out_off, g->in_off, g->out_tgt, g->in_tgt, g->out_lab, g->in_lab, g->n};
    a.in_arc = g->in_arc; a.owner = g->owner; a.pc_out = g->pc_out; a.pc_in = g->pc_in; a.cnt = g->cnt; a.depth = g->depth;
    a.ord_node = g->ord_node; a.ord_parent = g->ord_parent; a.ord_arc = g->ord_arc; a.lvl = g->lvl;
    a.allowed = g->labels; a.n_allowed = n_labels; a.direction = direction; a.hub_degree = hub_degree;
    constexpr uint32_t kLevelsPerSync = 8;
    uint32_t h_lvl[4] = {0, ns, 0, 0};
    for (uint32_t d = 0; d < max_depth && h_lvl[1];) {
        const uint32_t until = (uint32_t)std::min<uint64_t>(max_depth, (uint64_t)d + kLevelsPerSync);
        for (; d < until; ++d) {
            a.cur_depth = d;
            hipLaunchKernelGGL(ord_claim_kernel, dim3(1024), dim3(256), 0, g->stream, a);
            hipLaunchKernelGGL(ord_count_kernel, dim3(1024), dim3(256), 0, g->stream, a);
            hipLaunchKernelGGL(ord_scan_kernel, dim3(1), dim3(1024), 0, g->stream, g->cnt, g->lvl);
            hipLaunchKernelGGL(ord_emit_kernel, dim3(1024), dim3(256), 0, g->stream, a);
            hipLaunchKernelGGL(ord_advance_kernel, dim3(1), dim3(1), 0, g->stream, g->lvl);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemcpyAsync(h_lvl, g->lvl, 16, hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
    }
    const uint64_t total = (uint64_t)h_lvl[0] + h_lvl[1];
    *out_count = total;
    if (total > capacity) return fail(HVX_ERR_INVARIANT, "traversal visited %llu nodes, the output holds %llu", (unsigned long long)total,
                                      (unsigned long long)capacity);
    std::vector<uint32_t> hn(total), hp(total), ha(total), hd(g->n);
    if (total) {
        HIP_TRY(hipMemcpyAsync(hn.data(), g->ord_node, total * 4, hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(hipMemcpyAsync(hp.data(), g->ord_parent, total * 4, hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(hipMemcpyAsync(ha.data(), g->ord_arc, total * 4, hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(hipMemcpyAsync(hd.data(), g->depth, (size_t)g->n * 4, hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(hipStreamSynchronize(g->stream));
    }
    for (uint64_t i = 0; i < total; ++i) {
        const bool seed = i < ns;
        if (out_nodes) out_nodes[i] = hn[i];
        if (out_depths) out_depths[i] = hd[hn[i]];
        if (out_parents) out_parents[i] = seed ? UINT64_MAX : hp[i];
        if (out_edges) out_edges[i] = seed ? UINT64_MAX : (ha[i] & 0x7FFFFFFFu);
        if (out_against) out_against[i] = seed ? 0u : (ha[i] >> 31);
    }
    return HVX_OK;
}

// ---------------------------------------------------------------------------------------------
// Graph::traverse on the host (traversal.rs:197-309), both strategies.  DepthFirst is one dependent chain of stack pops -- there
// is nothing for a GPU in it -- so it runs here, over host arrays (hvx_traverse_host: callers that hold the CSR on the host, and
// the CPU tests) or over a host mirror of a device CSR fetched on first use (hvx_traverse_dfs).  BreadthFirst is included as the
// cross-check of the device's level-synchronous order (hvx_traverse_ordered).
// ---------------------------------------------------------------------------------------------
namespace {

struct HostCsr {
    uint32_t n = 0;
    std::vector<uint64_t> out_off, in_off;
    std::vector<uint32_t> out_tgt, in_tgt, out_lab, in_lab, in_arc;
    bool labeled = false;
};

void host_csr_free(void *p) { delete static_cast<HostCsr *>(p); }

struct HostArc { uint32_t nb; uint32_t edge; uint32_t against; };

// the arcs of `u` in the reference's order (model.rs:635-725), label filter applied
void host_arcs(const HostCsr &g, uint32_t u, uint32_t direction, const std::vector<uint32_t> &allowed, std::vector<HostArc> &out) {
    out.clear();
    auto ok = [&](const std::vector<uint32_t> &lab, uint64_t a) {
        if (!g.labeled || allowed.empty()) return true;
        return std::find(allowed.begin(), allowed.end(), lab[a]) != allowed.end();
    };
    uint64_t i = g.out_off[u], i1 = g.out_off[u + 1], j = g.in_off[u], j1 = g.in_off[u + 1];
    if (direction == HVX_DIR_OUT) j = j1;
    if (direction == HVX_DIR_IN) i = i1;
    while (i < i1 || j < j1) {
        if (direction == HVX_DIR_BOTH && j < j1 && g.in_tgt[j] == u) { ++j; continue; } // incoming self-loops are skipped
        const bool take_out = j >= j1 || (i < i1 && g.out_tgt[i] <= g.in_tgt[j]);
        if (take_out) { if (ok(g.out_lab, i)) out.push_back({g.out_tgt[i], (uint32_t)i, 0u}); ++i; }
        else { if (ok(g.in_lab, j)) out.push_back({g.in_tgt[j], g.in_arc[j], 1u}); ++j; }
    }
}

int host_traverse(const HostCsr &g, uint32_t strategy, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth, uint32_t direction,
                  const uint32_t *allowed_label_ids, uint32_t n_labels, uint32_t hub_degree, uint64_t capacity, uint64_t *out_nodes,
                  uint32_t *out_depths, uint64_t *out_parents, uint64_t *out_edges, uint32_t *out_against, uint64_t *out_count) {
    if (!out_count) return fail(HVX_ERR_INVARIANT, "null argument");
    *out_count = 0;
    if (direction > HVX_DIR_BOTH || strategy > 1u) return fail(HVX_ERR_INVARIANT, "bad direction / strategy");
    if (n_seeds == 0) return fail(HVX_ERR_INVARIANT, "traversal requires at least one seed"); // traversal.rs:198-202
    std::vector<uint32_t> seed_list;
    for (uint32_t i = 0; i < n_seeds; ++i) {
        if (seeds[i] >= g.n) return fail(HVX_ERR_INVARIANT, "unknown node %llu", (unsigned long long)seeds[i]);
        if (std::find(seed_list.begin(), seed_list.end(), (uint32_t)seeds[i]) == seed_list.end()) seed_list.push_back((uint32_t)seeds[i]);
    }
    const std::vector<uint32_t> allowed(allowed_label_ids, allowed_label_ids + n_labels);
    std::vector<uint8_t> visited(g.n, 0), is_seed(g.n, 0);
    for (uint32_t s : seed_list) is_seed[s] = 1;
    struct Visit { uint32_t node, depth, parent, edge, against; bool has_edge; };
    std::vector<Visit> visits;
    std::vector<HostArc> arcs;
    auto expands = [&](uint32_t u, uint32_t depth) {
        if (depth >= max_depth) return false;
        if (!is_seed[u] && hub_degree && (g.out_off[u + 1] - g.out_off[u]) + (g.in_off[u + 1] - g.in_off[u]) >= hub_degree) return false;
        return true;
    };
    if (strategy == 0u) { // BreadthFirst (traversal.rs:216-261)
        size_t head = 0;
        for (uint32_t s : seed_list) { visited[s] = 1; visits.push_back({s, 0, 0, 0, 0, false}); }
        while (head < visits.size()) {
            const Visit cur = visits[head++];
            if (!expands(cur.node, cur.depth)) continue;
            host_arcs(g, cur.node, direction, allowed, arcs);
            for (const HostArc &a : arcs) {
                if (visited[a.nb]) continue;
                visited[a.nb] = 1;
                visits.push_back({a.nb, cur.depth + 1, cur.node, a.edge, a.against, true});
            }
        }
    } else { // DepthFirst (traversal.rs:263-309): marked when scheduled, recorded when popped
        std::vector<Visit> stack;
        for (size_t i = seed_list.size(); i-- > 0;)
            if (!visited[seed_list[i]]) { visited[seed_list[i]] = 1; stack.push_back({seed_list[i], 0, 0, 0, 0, false}); }
        std::vector<HostArc> chosen;
        while (!stack.empty()) {
            const Visit cur = stack.back();
            stack.pop_back();
            visits.push_back(cur);
            if (!expands(cur.node, cur.depth)) continue;
            host_arcs(g, cur.node, direction, allowed, arcs);
            chosen.clear();
            for (const HostArc &a : arcs) { // one arc per unvisited neighbour: the first in arc order (`discovered.insert`)
                if (visited[a.nb]) continue;
                bool dup = false;
                for (const HostArc &c : chosen) dup |= c.nb == a.nb;
                if (!dup) chosen.push_back(a);
            }
            for (size_t i = chosen.size(); i-- > 0;) {
                visited[chosen[i].nb] = 1;
                stack.push_back({chosen[i].nb, cur.depth + 1, cur.node, chosen[i].edge, chosen[i].against, true});
            }
        }
    }
    *out_count = visits.size();
    if (visits.size() > capacity) return fail(HVX_ERR_INVARIANT, "traversal visited %zu nodes, the output holds %llu", visits.size(), (unsigned long long)capacity);
    for (size_t i = 0; i < visits.size(); ++i) {
        const Visit &v = visits[i];
        if (out_nodes) out_nodes[i] = v.node;
        if (out_depths) out_depths[i] = v.depth;
        if (out_parents) out_parents[i] = v.has_edge ? v.parent : UINT64_MAX;
        if (out_edges) out_edges[i] = v.has_edge ? v.edge : UINT64_MAX;
        if (out_against) out_against[i] = v.has_edge ? v.against : 0u;
    }
    return HVX_OK;
}

// outgoing arrays -> HostCsr (incoming rows by stable counting sort: sources ascending, parallel edges in outgoing-row order)
int host_csr_build(uint64_t n_nodes, uint64_t n_edges, const uint64_t *out_offsets, const uint64_t *out_targets, const uint32_t *edge_labels,
                   HostCsr &g) {
    if (n_nodes >= (1ull << 32) - 1 || n_edges >= (1ull << 31)) return fail(HVX_ERR_UNSUPPORTED, "graph too large for the host traversal");
    if (n_nodes && (!out_offsets || out_offsets[0] != 0 || out_offsets[n_nodes] != n_edges)) return fail(HVX_ERR_INVARIANT, "offsets do not span the edge array");
    if (n_edges && !out_targets) return fail(HVX_ERR_INVARIANT, "null targets");
    g.n = (uint32_t)n_nodes;
    g.labeled = edge_labels != nullptr;
    g.out_off.assign(out_offsets, out_offsets + (n_nodes ? n_nodes + 1 : 0));
    if (!n_nodes) g.out_off.assign(1, 0);
    g.out_tgt.resize(n_edges);
    g.in_off.assign(n_nodes + 1, 0);
    for (uint64_t u = 0; u < n_nodes; ++u) {
        if (out_offsets[u + 1] < out_offsets[u]) return fail(HVX_ERR_INVARIANT, "offsets not monotone");
        for (uint64_t a = out_offsets[u]; a < out_offsets[u + 1]; ++a) {
            if (out_targets[a] >= n_nodes) return fail(HVX_ERR_INVARIANT, "edge target out of range");
            if (a > out_offsets[u] && out_targets[a] < out_targets[a - 1])
                return fail(HVX_ERR_UNSUPPORTED, "ordered traversal needs outgoing rows sorted by target (model.rs:656-666)");
            g.out_tgt[a] = (uint32_t)out_targets[a];
            g.in_off[out_targets[a] + 1]++;
        }
    }
    for (uint64_t v = 0; v < n_nodes; ++v) g.in_off[v + 1] += g.in_off[v];
    g.in_tgt.resize(n_edges);
    g.in_arc.resize(n_edges);
    if (g.labeled) { g.out_lab.assign(edge_labels, edge_labels + n_edges); g.in_lab.resize(n_edges); }
    std::vector<uint64_t> cur(g.in_off.begin(), g.in_off.end() - 1);
    for (uint64_t u = 0; u < n_nodes; ++u)
        for (uint64_t a = out_offsets[u]; a < out_offsets[u + 1]; ++a) {
            const uint64_t slot = cur[out_targets[a]]++;
            g.in_tgt[slot] = (uint32_t)u;
            g.in_arc[slot] = (uint32_t)a;
            if (g.labeled) g.in_lab[slot] = edge_labels[a];
        }
    return HVX_OK;
}

} // namespace

extern "C" int hvx_traverse_host(uint64_t n_nodes, uint64_t n_edges, const uint64_t *out_offsets, const uint64_t *out_targets,
                                 const uint32_t *edge_labels, uint32_t strategy, const uint64_t *seeds, uint32_t n_seeds,
                                 uint32_t max_depth, uint32_t direction, const uint32_t *allowed_label_ids, uint32_t n_labels,
                                 uint32_t hub_degree, uint64_t capacity, uint64_t *out_nodes, uint32_t *out_depths,
                                 uint64_t *out_parents, uint64_t *out_edges, uint32_t *out_against, uint64_t *out_count) {
    HostCsr g;
    int rc = host_csr_build(n_nodes, n_edges, out_offsets, out_targets, edge_labels, g);
    if (rc) return rc;
    return host_traverse(g, strategy, seeds, n_seeds, max_depth, direction, allowed_label_ids, n_labels, hub_degree, capacity, out_nodes,
                         out_depths, out_parents, out_edges, out_against, out_count);
}

extern "C" int hvx_traverse_dfs(const hvx_csr *cg, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth, uint32_t direction,
                                const uint32_t *allowed_label_ids, uint32_t n_labels, uint32_t hub_degree, uint64_t capacity,
                                uint64_t *out_nodes, uint32_t *out_depths, uint64_t *out_parents, uint64_t *out_edges,
                                uint32_t *out_against, uint64_t *out_count) {
    if (!cg) return fail(HVX_ERR_INVARIANT, "null graph");
    hvx_csr *g = const_cast<hvx_csr *>(cg);
    if (!g->rows_sorted) return fail(HVX_ERR_UNSUPPORTED, "ordered traversal needs outgoing rows sorted by target (model.rs:656-666)");
    if (g->e >= (1ull << 31)) return fail(HVX_ERR_UNSUPPORTED, "ordered traversal serves graphs below 2^31 edges");
    std::lock_guard<std::mutex> lock(g->mu);
    if (!g->host) { // host mirror of the device CSR, fetched once
        HIP_TRY(hipSetDevice(g->device));
        auto h = std::make_unique<HostCsr>();
        h->n = g->n;
        h->labeled = g->out_lab != nullptr;
        h->out_off.resize((size_t)g->n + 1); h->in_off.resize((size_t)g->n + 1);
        h->out_tgt.resize(g->e); h->in_tgt.resize(g->e); h->in_arc.resize(g->e);
        HIP_TRY(hipMemcpy(h->out_off.data(), g->out_off, ((size_t)g->n + 1) * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(h->in_off.data(), g->in_off, ((size_t)g->n + 1) * 8, hipMemcpyDeviceToHost));
        if (g->e) {
            HIP_TRY(hipMemcpy(h->out_tgt.data(), g->out_tgt, g->e * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(h->in_tgt.data(), g->in_tgt, g->e * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(h->in_arc.data(), g->in_arc, g->e * 4, hipMemcpyDeviceToHost));
            if (h->labeled) {
                h->out_lab.resize(g->e); h->in_lab.resize(g->e);
                HIP_TRY(hipMemcpy(h->out_lab.data(), g->out_lab, g->e * 4, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(h->in_lab.data(), g->in_lab, g->e * 4, hipMemcpyDeviceToHost));
            }
        }
        g->host = h.release();
    }
    return host_traverse(*static_cast<const HostCsr *>(g->host), 1u, seeds, n_seeds, max_depth, direction, allowed_label_ids, n_labels,
                         hub_degree, capacity, out_nodes, out_depths, out_parents, out_edges, out_against, out_count);
}

// ---------------------------------------------------------------------------------------------
// Restricted (prefiltered) search: search_restricted_observed_with_beam_percent
// (restricted.rs:529-613).  Plan `Exact` is the reference's restricted_exact_scan; candidate sets
// above the exact thresholds are ALSO scanned exactly on the device (a gathered flat scan over the
// allowed rows): on MI355X a 100k x 1536 f32 candidate set is 0.6 GB = ~0.1 ms of HBM, and the
// result is the exact answer that the reference's ACORN-style walk (restricted.rs:837-1148)
// approximates (recall gate >= 0.92/0.95).  See DESIGN.md "restricted search".
// ---------------------------------------------------------------------------------------------
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
    ix->sync_rewrites();
    HIP_TRY(hipSetDevice(ix->device));
    hvx_restricted_params rp; // this entry point answers every candidate-set size with the exact gathered scan
    memset(&rp, 0, sizeof(rp));
    rp.k = k;
    rp.ef = ef;
    rp.strategy = HVX_RESTRICTED_EXACT;
    return restricted_search_host(ix, queries, b, rp, allowed_ids, allowed_offsets, n_allowed, out_ids, out_scores, out_counts, out_status, nullptr, stats);
}

// ---------------------------------------------------------------------------------------------
// Fused prefilter + restricted kNN: the candidate bitmap never leaves the device.
// ---------------------------------------------------------------------------------------------
namespace {

// external node id -> internal row of the index (ids ascending), kSentinel when the node holds no vector
// (or whose vector has been deleted: `dead` = the image's deleted-row bitmap, NULL while nothing has been deleted)
__device__ __forceinline__ uint32_t find_row(const uint64_t *ids, uint32_t n, uint64_t id, bool contiguous, const uint32_t *dead) {
    if (n == 0) return kSentinel;
    uint32_t row;
    if (contiguous) {
        if (!(id >= ids[0] && id - ids[0] < n)) return kSentinel;
        row = (uint32_t)(id - ids[0]);
    } else {
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (ids[mid] < id) lo = mid + 1;
            else hi = mid;
        }
        if (!(lo < n && ids[lo] == id)) return kSentinel;
        row = lo;
    }
    if (dead && ((dead[row >> 5] >> (row & 31u)) & 1u)) return kSentinel;
    return row;
}

// pass 1: per 256-word block, how many set bits map to an indexed row (restricted.rs:615-659: ids that are not
// indexed are omitted) and how many bits are set at all (the RestrictedVectorCandidates population, :356-371)
__global__ __launch_bounds__(256) void bitmap_count_kernel(const uint32_t *bitmap, uint32_t n_words, const uint64_t *ids, uint32_t n,
                                                           uint32_t contiguous, const uint32_t *dead, uint32_t *block_rows, uint32_t *total_bits, uint32_t *block_bits) {
    __shared__ uint32_t s_rows, s_bits;
    if (threadIdx.x == 0) { s_rows = 0; s_bits = 0; }
    __syncthreads();
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    uint32_t word = w < n_words ? bitmap[w] : 0u, rows = 0;
    const uint32_t bits = (uint32_t)__builtin_popcount(word);
    while (word) {
        const uint32_t b = (uint32_t)__builtin_ctz(word);
        word &= word - 1u;
        rows += find_row(ids, n, (uint64_t)w * 32u + b, contiguous != 0u, dead) != kSentinel ? 1u : 0u;
    }
    if (rows) atomicAdd(&s_rows, rows);
    if (bits) atomicAdd(&s_bits, bits);
    __syncthreads();
    if (threadIdx.x == 0) {
        block_rows[blockIdx.x] = s_rows;
        if (block_bits) block_bits[blockIdx.x] = s_bits; // for the deterministic sample: ranks count every candidate id
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
                                                             uint32_t contiguous, const uint32_t *dead, const uint32_t *block_base, uint32_t *subset) {
    __shared__ uint32_t cnt[256];
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    const uint32_t word0 = w < n_words ? bitmap[w] : 0u;
    uint32_t word = word0, rows = 0;
    while (word) {
        const uint32_t b = (uint32_t)__builtin_ctz(word);
        word &= word - 1u;
        rows += find_row(ids, n, (uint64_t)w * 32u + b, contiguous != 0u, dead) != kSentinel ? 1u : 0u;
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
        const uint32_t r = find_row(ids, n, (uint64_t)w * 32u + b, contiguous != 0u, dead);
        if (r != kSentinel) subset[out++] = r;
    }
}

// deterministic_sample_ids (restricted.rs:321-342) on the device: the candidate with rank ranks[t] in the ascending id order
// of the bitmap, as an index row (kSentinel when that id holds no vector).  bits_prefix = exclusive scan of the per-block counts.
__global__ __launch_bounds__(64) void bitmap_select_kernel(const uint32_t *bitmap, uint32_t n_words, const uint32_t *bits_prefix, uint32_t n_blocks,
                                                           const uint64_t *ids, uint32_t n, uint32_t contiguous, const uint32_t *dead, const uint32_t *ranks,
                                                           uint32_t n_ranks, uint32_t *out_rows) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_ranks) return;
    const uint32_t r = ranks[t];
    uint32_t lo = 0, hi = n_blocks; // last block whose prefix is <= r
    while (hi - lo > 1) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (bits_prefix[mid] <= r) lo = mid;
        else hi = mid;
    }
    uint32_t rem = r - bits_prefix[lo], row = kSentinel;
    for (uint32_t w = lo * 256u; w < n_words && w < (lo + 1u) * 256u; ++w) {
        uint32_t word = bitmap[w];
        const uint32_t pc = (uint32_t)__builtin_popcount(word);
        if (rem >= pc) { rem -= pc; continue; }
        while (rem--) word &= word - 1u;
        row = find_row(ids, n, (uint64_t)w * 32u + (uint32_t)__builtin_ctz(word), contiguous != 0u, dead);
        break;
    }
    out_rows[t] = row;
}

// Round 6: one `expand` hop (expand.rs:16-80) straight into the scan's row list.  One wavefront per seed, lanes over its arcs: a target
// whose bit was clear joins the candidate population (a bitmap test-and-set makes the union a set) and, when it holds a live vector, the
// row list -- in whatever order the wavefronts get there: the one-launch scan (hvx_restricted_exact.hip) orders by (score, row) itself.
// counters[0] = rows of the list, counters[1] = candidate population.  Replaces the level kernel + count + scan + compact sequence and
// the host read-back between them for plans that are exact whatever the population turns out to be.
__global__ __launch_bounds__(1024) void expand_collect_kernel(CsrView g, const uint32_t *seeds, uint32_t n_seeds, uint32_t direction, const uint32_t *allowed,
                                                             uint32_t n_allowed, uint32_t *visited, const uint64_t *ids, uint32_t n, uint32_t contiguous,
                                                             const uint32_t *dead, uint32_t *rows_out, uint32_t rows_cap, uint32_t *counters) {
    const int lane = (int)(threadIdx.x & 63u);
    // one arc -> (joined the population, row of its target); the wavefront's arcs of one step share ONE pair of counter updates (a where_()
    // group of 10 000 sources with one edge each was 20 000 atomics on two words with a wavefront per seed: 0.25 ms of a 0.34 ms call)
    auto take = [&](bool active, const uint32_t *tgt, const uint32_t *lab, uint64_t at) __attribute__((always_inline)) {
        bool fresh = false;
        uint32_t row = kSentinel;
        if (active && (!lab || label_ok(lab[at], allowed, n_allowed))) {
            const uint32_t v = tgt[at];
            const uint32_t bit = 1u << (v & 31u);
            fresh = (atomicOr(&visited[v >> 5], bit) & bit) == 0u;
            if (fresh) row = find_row(ids, n, (uint64_t)v, contiguous != 0u, dead);
        }
        const unsigned long long fm = __ballot(fresh), rm = __ballot(row != kSentinel);
        if (fm == 0ull) return;
        uint32_t base = 0;
        if (lane == 0) {
            atomicAdd(&counters[1], (uint32_t)__builtin_popcountll(fm));
            if (rm) base = atomicAdd(&counters[0], (uint32_t)__builtin_popcountll(rm));
        }
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (row != kSentinel) {
            const uint32_t pos = base + (uint32_t)__builtin_popcountll(rm & ((1ull << lane) - 1ull));
            if (pos < rows_cap) rows_out[pos] = row;
        }
    };
    constexpr uint32_t kLight = 16; // rows up to this long are walked one arc per step by the seed's own lane (64 seeds per wavefront)
    // Round 6 (second build): the sixteen wavefronts of a workgroup walk their light rows in lock step and share ONE pair of counter updates per
    // step (a group of 100 000 sources with one edge each was still 3 126 updates of two words -- same-address atomics serialise at the
    // memory side: 121 us of a 0.43-ms call)
    __shared__ uint32_t s_f[16], s_r[16], s_base, s_steps;
    const uint32_t wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    for (uint32_t c0 = blockIdx.x * blockDim.x; c0 < n_seeds; c0 += gridDim.x * blockDim.x) {
        const uint32_t s0 = c0 + wave * 64u;
        const bool have = s0 + (uint32_t)lane < n_seeds;
        const uint32_t node = have ? seeds[s0 + (uint32_t)lane] : 0u;
        for (int pass = 0; pass < 2; ++pass) {
            const bool use_out = pass == 0;
            if (use_out && direction == 1u) continue;  // In only
            if (!use_out && direction == 0u) continue; // Out only
            const uint64_t *off = use_out ? g.out_off : g.in_off;
            const uint32_t *tgt = use_out ? g.out_tgt : g.in_tgt;
            const uint32_t *lab = use_out ? g.out_lab : g.in_lab;
            const uint64_t a0 = have ? off[node] : 0ull, a1 = have ? off[node + 1] : 0ull;
            const uint32_t deg = (uint32_t)((a1 - a0) < 0xFFFFFFFFull ? (a1 - a0) : 0xFFFFFFFFull);
            const bool light = deg <= kLight;
            uint32_t steps = light ? deg : 0u;
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)steps, sh, 64); steps = o > steps ? o : steps; }
            if (threadIdx.x == 0) s_steps = 0u;
            __syncthreads();
            if (lane == 0 && steps) atomicMax(&s_steps, steps);
            __syncthreads();
            const uint32_t wg_steps = s_steps;
            for (uint32_t i = 0; i < wg_steps; ++i) {
                bool fresh = false;
                uint32_t row = kSentinel;
                if (light && i < deg && (!lab || label_ok(lab[a0 + i], allowed, n_allowed))) {
                    const uint32_t v = tgt[a0 + i];
                    const uint32_t bit = 1u << (v & 31u);
                    fresh = (atomicOr(&visited[v >> 5], bit) & bit) == 0u;
                    if (fresh) row = find_row(ids, n, (uint64_t)v, contiguous != 0u, dead);
                }
                const unsigned long long fm = __ballot(fresh), rm = __ballot(row != kSentinel);
                if (lane == 0) { s_f[wave] = (uint32_t)__builtin_popcountll(fm); s_r[wave] = (uint32_t)__builtin_popcountll(rm); }
                __syncthreads();
                if (threadIdx.x == 0) {
                    uint32_t tf = 0, tr = 0;
                    for (uint32_t w = 0; w < waves; ++w) { tf += s_f[w]; tr += s_r[w]; }
                    if (tf) atomicAdd(&counters[1], tf);
                    s_base = tr ? atomicAdd(&counters[0], tr) : 0u;
                }
                __syncthreads();
                if (row != kSentinel) {
                    uint32_t pos = s_base + (uint32_t)__builtin_popcountll(rm & ((1ull << lane) - 1ull));
                    for (uint32_t w = 0; w < wave; ++w) pos += s_r[w];
                    if (pos < rows_cap) rows_out[pos] = row;
                }
                __syncthreads(); // (the next step writes s_f / s_r again)
            }
            unsigned long long heavy = __ballot(have && !light); // long rows (hubs): the whole wavefront strides over the arcs of one seed at a time
            while (heavy) {
                const int l = __builtin_ctzll(heavy);
                heavy &= heavy - 1ull;
                const uint64_t h0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(a0 >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)a0, l);
                const uint64_t h1 = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(a1 >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)a1, l);
                for (uint64_t a = h0; a < h1; a += 64) take(a + (uint64_t)lane < h1, tgt, lab, a + (uint64_t)lane);
            }
        }
    }
}

} // namespace

static int prefilter_search_impl(const hvx_index *cix, const hvx_csr *cg, const float *queries, uint32_t b, const hvx_restricted_params &rp,
                                 uint32_t mode, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth, uint32_t direction,
                                 const uint32_t *allowed_label_ids, uint32_t n_labels, uint32_t hub_degree, uint32_t include_seeds,
                                 uint64_t *out_ids, float *out_scores, uint32_t *out_counts, uint32_t *out_status, uint64_t *out_candidates,
                                 hvx_restricted_stats *rstats, hvx_stats *stats);

// One hop + exact scan with ONE host wait (round 6; the reference issues one query per call, index_lifecycle_scale.rs:1893-1912: the
// latency of this call is what a request sees).  Taken when the plan is exact WHATEVER the hop reaches -- strategy EXACT, or a planned
// strategy whose bound on the population (seeds x the largest adjacency row) is inside the plan's exact region -- and the one-launch scan
// serves the shape.  Everything is enqueued on the index's stream: clear the bitmap, expand_collect_kernel (seeds read out of the pinned
// staging buffer), queries copied + validated, the scan with its row count read ON THE DEVICE, results into mapped host rows.
static int prefilter_expand_lean(hvx_index *ix, hvx_csr *g, const float *queries, uint32_t b, const hvx_restricted_params &rp, const uint64_t *seeds,
                                 uint32_t n_seeds, uint32_t direction, const uint32_t *allowed_label_ids, uint32_t n_labels, uint64_t bound,
                                 uint64_t *out_ids, float *out_scores, uint32_t *out_counts, uint32_t *out_status, uint64_t *out_candidates,
                                 hvx_restricted_stats *rstats, hvx_stats *stats) {
    int rc;
    const uint32_t k = rp.k;
    hipStream_t s = ix->stream;
    if (n_seeds > g->cap_h_seeds) {
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipStreamSynchronize(g->stream));
        if (g->h_seeds) (void)hipHostFree(g->h_seeds);
        g->h_seeds = nullptr;
        g->cap_h_seeds = 0;
        const size_t want = std::max<size_t>((size_t)n_seeds + n_seeds / 2, 4096);
        if (hipHostMalloc((void **)&g->h_seeds, want * 4, hipHostMallocDefault) != hipSuccess) return fail(HVX_ERR_DEVICE, "hipHostMalloc(%zu) seeds", want * 4);
        g->cap_h_seeds = want;
    }
    for (uint32_t i = 0; i < n_seeds; ++i) {
        if (seeds[i] >= g->n) return fail(HVX_ERR_INVARIANT, "unknown node %llu", (unsigned long long)seeds[i]);
        g->h_seeds[i] = (uint32_t)seeds[i];
    }
    if ((rc = upload_labels(g, allowed_label_ids, n_labels))) return rc; // (on g->stream: ordered below)
    if (n_labels) {
        if (!g->done) HIP_TRY(hipEventCreateWithFlags(&g->done, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(g->done, g->stream));
        HIP_TRY(hipStreamWaitEvent(s, g->done, 0));
    }
    const uint32_t rows_cap = (uint32_t)std::min<uint64_t>(bound, ix->dev.n);
    if (rows_cap > ix->cap_subset) {
        if ((rc = ix->regrow((void **)&ix->f_subset, (size_t)std::max<uint32_t>(rows_cap, 1u) * 4))) return rc;
        ix->cap_subset = rows_cap;
    }
    if (ix->cap_pf_blocks < 4) {
        if ((rc = ix->regrow((void **)&ix->pf_blocks, 64))) return rc;
        ix->cap_pf_blocks = 16;
    }
    const size_t words32 = ((g->n + 63) / 64) * 2;
    HIP_TRY(hipMemsetAsync(g->visited, 0, std::max<size_t>(words32, 2) * 4, s));
    HIP_TRY(hipMemsetAsync(ix->pf_blocks, 0, 8, s));
    CsrView v{g->out_off, g->in_off, g->out_tgt, g->in_tgt, g->out_lab, g->in_lab, g->n};
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(512, ((uint64_t)n_seeds + 1023) / 1024); // (a wavefront takes 64 seeds at a time, a workgroup 1 024)
    hipLaunchKernelGGL(expand_collect_kernel, dim3(std::max(blocks, 1u)), dim3(1024), 0, s, v, g->h_seeds, n_seeds, direction, g->labels, n_labels, g->visited,
                       ix->dev.ids, ix->dev.n, ix->contiguous ? 1u : 0u, ix->dev.dead, ix->f_subset, rows_cap, ix->pf_blocks);
    HIP_TRY(hipGetLastError());
    return restricted_direct_shared_devcount(ix, queries, b, k, ix->f_subset, rows_cap, ix->pf_blocks, out_ids, out_scores, out_counts, out_status,
                                             out_candidates, rstats, stats);
}

extern "C" int hvx_prefilter_search_batch(const hvx_index *cix, const hvx_csr *cg, const float *queries, uint32_t b, uint32_t k,
                                          uint32_t ef, uint32_t mode, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth,
                                          uint32_t direction, const uint32_t *allowed_label_ids, uint32_t n_labels,
                                          uint32_t hub_degree, uint32_t include_seeds, uint64_t *out_ids, float *out_scores,
                                          uint32_t *out_counts, uint32_t *out_status, uint64_t *out_candidates, hvx_stats *stats) {
    hvx_restricted_params rp;
    memset(&rp, 0, sizeof(rp));
    rp.k = k;
    rp.ef = ef;
    rp.strategy = HVX_RESTRICTED_EXACT; // this entry point answers every candidate-set size with the exact gathered scan
    return prefilter_search_impl(cix, cg, queries, b, rp, mode, seeds, n_seeds, max_depth, direction, allowed_label_ids, n_labels, hub_degree,
                                 include_seeds, out_ids, out_scores, out_counts, out_status, out_candidates, nullptr, stats);
}

extern "C" int hvx_prefilter_search_batch_params(const hvx_index *cix, const hvx_csr *cg, const float *queries, uint32_t b,
                                                 const hvx_restricted_params *params, uint32_t mode, const uint64_t *seeds, uint32_t n_seeds,
                                                 uint32_t max_depth, uint32_t direction, const uint32_t *allowed_label_ids, uint32_t n_labels,
                                                 uint32_t hub_degree, uint32_t include_seeds, uint64_t *out_ids, float *out_scores,
                                                 uint32_t *out_counts, uint32_t *out_status, uint64_t *out_candidates,
                                                 hvx_restricted_stats *out_restricted_stats, hvx_stats *stats) {
    if (!params) return fail(HVX_ERR_INVARIANT, "null argument");
    if (params->strategy > HVX_RESTRICTED_REFERENCE_PLAN) return fail(HVX_ERR_INVARIANT, "unknown restricted strategy %u", params->strategy);
    return prefilter_search_impl(cix, cg, queries, b, *params, mode, seeds, n_seeds, max_depth, direction, allowed_label_ids, n_labels, hub_degree,
                                 include_seeds, out_ids, out_scores, out_counts, out_status, out_candidates, out_restricted_stats, stats);
}

static int prefilter_search_impl(const hvx_index *cix, const hvx_csr *cg, const float *queries, uint32_t b, const hvx_restricted_params &rp,
                                 uint32_t mode, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth, uint32_t direction,
                                 const uint32_t *allowed_label_ids, uint32_t n_labels, uint32_t hub_degree, uint32_t include_seeds,
                                 uint64_t *out_ids, float *out_scores, uint32_t *out_counts, uint32_t *out_status, uint64_t *out_candidates,
                                 hvx_restricted_stats *rstats, hvx_stats *stats) {
    const uint32_t k = rp.k, ef = rp.ef;
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
        if (rstats) memset(&rstats[q], 0, sizeof(hvx_restricted_stats));
    }
    if (b == 0) return HVX_OK;
    if (mode == HVX_PREFILTER_EXPAND && n_seeds == 0) return HVX_OK; // an empty stream expands to nothing
    std::lock_guard<std::mutex> glock(g->mu);
    std::lock_guard<std::mutex> lock(ix->mu);
    ix->sync_rewrites();
    if (mode == HVX_PREFILTER_EXPAND && direction <= HVX_DIR_BOTH && ix->opt[HVX_OPT_RESTRICTED_DIRECT] != 1u && !rp.explicit_budgets && k <= 64u &&
        restricted_direct_supported(ix, k) && ix->dev.n != 0u) {
        // what one hop can reach at most: seeds x the largest adjacency row (both rows for direction Both), never more than the graph
        const uint64_t per_seed = (direction != HVX_DIR_IN ? g->max_out_deg : 0) + (direction != HVX_DIR_OUT ? g->max_in_deg : 0);
        const uint64_t bound = std::min<uint64_t>((uint64_t)n_seeds * per_seed, g->n);
        RestrictedPlan plan;
        const bool exact_any = bound == 0 || (bound <= 1000000ull && restricted_make_plan(rp, bound, ix->dev.dim, &plan, ix) == HVX_OK && plan.strategy == HVX_RESTRICTED_EXACT &&
                                             (rp.strategy != HVX_RESTRICTED_REFERENCE_PLAN || bound <= 256)); // (a plan is monotone in the population)
        const uint64_t work = (uint64_t)b * std::min<uint64_t>(bound, ix->dev.n) * ix->dev.dim;
        // (the other path costs ~0.2 ms of host round trips more: the one-launch scan is the better choice up to 2^28 here, not 2^26)
        if (exact_any && bound != 0 && b <= ix->max_batch && (work <= (1ull << 28) || ix->opt[HVX_OPT_RESTRICTED_DIRECT] == 2u))
            return prefilter_expand_lean(ix, g, queries, b, rp, seeds, n_seeds, direction, allowed_label_ids, n_labels, bound, out_ids, out_scores, out_counts,
                                         out_status, out_candidates, rstats, stats);
    }
    int rc = run_bfs_locked(g, seeds, n_seeds, mode == HVX_PREFILTER_EXPAND ? 1u : max_depth, direction, allowed_label_ids, n_labels,
                            mode == HVX_PREFILTER_EXPAND ? 0u : hub_degree, mode == HVX_PREFILTER_EXPAND ? 0u : include_seeds,
                            mode == HVX_PREFILTER_EXPAND, nullptr, nullptr, ix->stream);
    if (rc) return rc; // (the index's stream waits for the traversal: the bitmap is complete before anything below reads it)
    const uint32_t n_words = ((g->n + 63u) / 64u) * 2u, n_blocks = (n_words + 255u) / 256u;
    if (2 * (n_blocks + 2) > ix->cap_pf_blocks) {
        if ((rc = ix->regrow((void **)&ix->pf_blocks, (size_t)2 * (n_blocks + 2) * 4))) return rc;
        ix->cap_pf_blocks = 2 * (n_blocks + 2);
    }
    uint32_t *d_total_bits = ix->pf_blocks + n_blocks + 1;
    uint32_t *d_block_bits = ix->pf_blocks + n_blocks + 2; // [n_blocks + 1]: per-block candidate counts, then their scan
    HIP_TRY(hipMemsetAsync(d_total_bits, 0, 4, ix->stream));
    hipLaunchKernelGGL(bitmap_count_kernel, dim3(n_blocks), dim3(256), 0, ix->stream, g->visited, n_words, ix->dev.ids, ix->dev.n,
                       ix->contiguous ? 1u : 0u, ix->dev.dead, ix->pf_blocks, d_total_bits, d_block_bits);
    hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(1024), 0, ix->stream, ix->pf_blocks, n_blocks);
    HIP_TRY(hipGetLastError());
    if ((rc = ix->pin(64))) return rc;
    uint32_t *totals = reinterpret_cast<uint32_t *>(ix->h_pin); // rows to scan, candidate population (read back through the pinned mirror)
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
                           ix->contiguous ? 1u : 0u, ix->dev.dead, ix->pf_blocks, ix->f_subset);
    HIP_TRY(hipGetLastError());
    // restricted_execution_plan_with_beam_percent (restricted.rs:426-453) over the candidate population
    RestrictedPlan plan;
    if ((rc = restricted_make_plan(rp, population, ix->dev.dim, &plan, ix))) return rc;
    (void)kk;
    const uint32_t *d_samples = nullptr;
    if (plan.strategy == HVX_RESTRICTED_FILTERED && plan.p.n_sample) {
        // the deterministic seeds: ranks into the candidate population, resolved against the bitmap on the device
        std::vector<uint64_t> ranks64;
        restricted_sample_ranks(population, plan.p.n_sample, ranks64);
        std::vector<uint32_t> ranks(ranks64.begin(), ranks64.end());
        const uint32_t ns = plan.p.n_sample;
        if ((size_t)2 * ns * 4 > ix->cap_w_samples) {
            if ((rc = ix->regrow((void **)&ix->w_samples, (size_t)2 * ns * 4))) return rc;
            ix->cap_w_samples = (size_t)2 * ns * 4;
        }
        HIP_TRY(hipMemcpyAsync(ix->w_samples + ns, ranks.data(), (size_t)ns * 4, hipMemcpyHostToDevice, ix->stream));
        hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(1024), 0, ix->stream, d_block_bits, n_blocks);
        hipLaunchKernelGGL(bitmap_select_kernel, dim3((ns + 63u) / 64u), dim3(64), 0, ix->stream, g->visited, n_words, d_block_bits, n_blocks,
                           ix->dev.ids, ix->dev.n, ix->contiguous ? 1u : 0u, ix->dev.dead, ix->w_samples + ns, ns, ix->w_samples);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(ix->stream)); // `ranks` lives on this frame
        d_samples = ix->w_samples;
    }
    return restricted_run_plan(ix, queries, b, k, plan, ix->f_subset, n_rows, d_samples, out_ids, out_scores, out_counts, out_status, rstats, stats);
}
