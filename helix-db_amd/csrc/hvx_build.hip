// hvx_build.hip -- GPU-assisted HNSW build: the reference's insert path
// (crates/db/src/search/vector/mutation.rs:642-895 insert_with_mutation_cache / insert_hnsw) for BATCHES of nodes on
// the device, over an index image whose rows are already resident in HBM.
//
// Per batch of B new nodes (consecutive ids; every node of a batch sees the graph as of the batch start):
//   1. search   hnsw_wave_kernel<BUILD> -- greedy descent above min(level, max_layer), search_layer_beam
//               (mutation.rs:904-1005) on every layer from there to 0; per layer the first 2*Mmax entries of W
//   2. select   build_select_kernel    -- select_neighbors_heuristic (mutation.rs:1072-1097) = select_diverse
//               (mod.rs:809-856) over those hydrated candidates, backfill; writes the new node's canonical row
//   3. link     build_link_kernel / build_link_wg_kernel -- add_bidirectional_link (mutation.rs:1498-1583) for every selected neighbour IN
//               SELECTION ORDER: append, and when the row exceeds Mmax rank its neighbours by distance to the row's owner,
//               select_diverse + backfill, canonical row (neighbor_set.rs:1-9), remove the reverse edge of every dropped
//               neighbour (mutation.rs:1890-1908).  One wavefront per new node; a row is changed under its owner's lock.
// With B = 1 this IS the reference's sequential insertion: rows equal the oracle's row for row (tests).  With B > 1 the
// nodes of one batch do not see each other and the order in which concurrent links reach a shared neighbour is not
// defined; the result is a valid (symmetric, degree-bounded, canonical) HNSW graph whose recall / work match the
// sequential one (bench.py graph_equivalence).  A node whose level exceeds the current top layer always forms its own
// batch (it becomes the entry point, mutation.rs:769-772).
//
// Every distance is the reference-order f32 distance of hvx_device.h (group_distance), so `dist(c,s) < dist(c,q)` decisions
// are the CPU path's.  Served shapes: f32 rows of any dimension, L2 / cosine / Manhattan, every summation tree (the unrolled
// search builds for L2 / cosine + AVX+FMA + dim in {128,...,1536} + ef_construction <= 352, the GENERIC build of the same kernel
// otherwise; select / link kernels per metric and tree), m0 <= 32, ef_construction <= 800.
// Round 3: the batched link step runs one WORKGROUP per link with the prune evaluated from LDS (build_link_wg_kernel below) and
// the search side two wavefronts per SIMD for batches > 1 024 nodes: 1M x 768 in 4.1 s (round 2: 10.0 s).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "hvx_host.h"
#include "hvx_graph_dev.h"

using namespace hvx;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(HVX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

namespace hvx {

#ifdef HVX_TUNING
#define HVX_DBG_ADD(a, i, v) do { if ((a).dbg) atomicAdd(&(a).dbg[i], (uint32_t)(v)); } while (0)
#else
#define HVX_DBG_ADD(a, i, v) do { } while (0)
#endif

constexpr uint32_t kCand = 64; // candidates kept per layer and node (2 * Mmax <= 64)

struct BuildArgs {
    DevIndex ix;
    uint32_t *l0, *up;          // the same rows as ix.l0 / ix.up, writable
    uint32_t *locks;            // [n] one lock per row owner (all its layers)
    const uint32_t *nodes;      // [b] internal ids of the batch
    uint32_t b, layers;         // layers = old max_layer + 1
    const uint64_t *cand_ids;   // [layers][b][kCand] internal ids (search output)
    const float *cand_sc;       // [layers][b][kCand]
    const uint32_t *cand_cnt;   // [layers][b]
    uint32_t *sel;              // [layers][b][32] selected neighbours in selection order
    uint32_t *sel_cnt;          // [layers][b]
    uint32_t m, m0;             // degree limits: upper layers / layer 0 (m0 = max(m0, 2m), mutation.rs:178-196)
    uint32_t *err;              // [1] set when a row would overflow its stride (invariant violation)
    uint32_t ldp, ncmax;        // build_link_wg_kernel: row stride of a column block in LDS (floats), candidate rows the LDS holds
    uint32_t link_ck;           // 32-float chunks per column block
    uint32_t *dbg;              // tuning builds (HVX_BUILD_DEBUG): [0] lock spins [1] prunes [2] reverse-edge removals [3] plain appends
    float *gdm;                 // one-node steps: [layers][kSeqLayerDm] distance matrices (device-scope stores / loads)
    uint32_t *tick;             // ... [2][layers] workgroups that have delivered (zero between launches)
    uint32_t g0, gu;            // ... workgroups of layer 0 / of every upper layer
};

__device__ __forceinline__ void lock_row(uint32_t *locks, uint32_t node, int lane) {
    if (lane == 0) {
        while (__hip_atomic_exchange(&locks[node], 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
}
__device__ __forceinline__ void unlock_row(uint32_t *locks, uint32_t node, int lane) {
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) __hip_atomic_store(&locks[node], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// row of `node` on `layer`, and its stride
__device__ __forceinline__ uint32_t *row_ptr(const BuildArgs &a, uint32_t node, uint32_t layer, uint32_t &stride) {
    if (layer == 0u) { stride = a.ix.s0; return a.l0 + (size_t)node * a.ix.s0; }
    stride = a.ix.su;
    return a.up + (size_t)(a.ix.up_base[node] + layer - 1u) * a.ix.su;
}

// ---- step 2: the new node's own neighbour lists ----
template <uint32_t METRIC, bool FUSED> __global__ __launch_bounds__(64) void build_select_kernel(BuildArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevIndex &ix = a.ix;
    const uint32_t q = blockIdx.x, layer = blockIdx.y;
    const int lane = (int)threadIdx.x;
    const uint32_t node = a.nodes[q];
    const uint32_t lv = ix.level[node];
    const uint32_t top = lv < a.layers - 1u ? lv : a.layers - 1u; // min(node level, old max_layer)
    if (layer > top) return; // layers above the old top stay empty rows (mutation.rs:883-894)
    BuildLds L = carve_build(smem, ix.ld);
    const uint32_t maxn = layer == 0u ? a.m0 : a.m;
    const size_t slot = (size_t)layer * a.b + q;
    const uint32_t cnt = a.cand_cnt[slot];
    const uint32_t hyd = cnt < 2u * maxn ? cnt : 2u * maxn; // select_neighbors_heuristic hydrates the first 2*Mmax only
    if ((uint32_t)lane < hyd) {
        L.cid[lane] = (uint32_t)a.cand_ids[slot * kCand + lane];
        L.csc[lane] = a.cand_sc[slot * kCand + lane];
    }
    __syncthreads();
    const uint32_t ns = select_diverse_dev<METRIC, FUSED>(ix, L, hyd, maxn, lane);
    if ((uint32_t)lane < ns) a.sel[slot * 32u + lane] = L.kept[lane];
    if (lane == 0) a.sel_cnt[slot] = ns;
    uint32_t stride;
    uint32_t *row = row_ptr(a, node, layer, stride);
    store_canonical(row, stride, L.kept, ns, lane, false); // nobody else can reach this row before the link step
}

// remove `victim` from the row of `owner` on `layer` (mutation.rs:1890-1908), under owner's lock
__device__ __forceinline__ void remove_edge_dev(const BuildArgs &a, uint32_t layer, uint32_t owner, uint32_t victim, int lane) {
    lock_row(a.locks, owner, lane);
    uint32_t stride;
    uint32_t *row = row_ptr(a, owner, layer, stride);
    const uint32_t v = (uint32_t)lane < stride ? ld_row(row + lane) : kSentinel;
    const bool keep = v != kSentinel && v != victim;
    const unsigned long long km = __ballot(keep);
    const uint32_t pos = (uint32_t)__builtin_popcountll(km & ((1ull << lane) - 1ull));
    const uint32_t nk = (uint32_t)__builtin_popcountll(km);
    __syncthreads();
    if (keep) st_row(row + pos, v);
    if ((uint32_t)lane >= nk && (uint32_t)lane < stride) st_row(row + lane, kSentinel);
    unlock_row(a.locks, owner, lane);
}

// ---- step 3: bidirectional links of the new node, in selection order, top layer first ----
template <uint32_t METRIC, bool FUSED> __global__ __launch_bounds__(64) void build_link_kernel(BuildArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevIndex &ix = a.ix;
    const uint32_t q = blockIdx.x;
    const int lane = (int)threadIdx.x, grp = lane >> 3, j = lane & 7;
    const uint32_t me = a.nodes[q];
    const uint32_t lv = ix.level[me];
    const uint32_t top = lv < a.layers - 1u ? lv : a.layers - 1u;
    BuildLds L = carve_build(smem, ix.ld);
    for (int32_t layer = (int32_t)top; layer >= 0; --layer) {
        const uint32_t maxn = layer == 0 ? a.m0 : a.m;
        const size_t slot = (size_t)layer * a.b + q;
        const uint32_t ns = a.sel_cnt[slot];
        for (uint32_t s = 0; s < ns; ++s) {
            const uint32_t to = a.sel[slot * 32u + s];
            // add_bidirectional_link(from = me, to) (mutation.rs:1498-1583)
            lock_row(a.locks, to, lane);
            uint32_t stride;
            uint32_t *row = row_ptr(a, to, (uint32_t)layer, stride);
            uint32_t v = (uint32_t)lane < stride ? ld_row(row + lane) : kSentinel;
            uint32_t deg = (uint32_t)__builtin_popcountll(__ballot(v != kSentinel));
            const bool present = __ballot(v == me) != 0ull;
            if (!present) {
                if (deg >= 64u) { if (lane == 0) *a.err = 1u; unlock_row(a.locks, to, lane); continue; }
                if ((uint32_t)lane == deg) v = me; // rows are canonical: the valid ids occupy lanes 0..deg-1
                ++deg;
            }
            const uint32_t nc = deg;
            uint32_t dropped_id = kSentinel; // per lane: a candidate this prune removed
            if (nc > maxn) {
                // rank the row's neighbours by distance to its owner, select_diverse with the owner as the reference point
                stage_row(ix, L.qv, to, lane);
                const float thdr = ix.hdr[to];
                __syncthreads();
                if ((uint32_t)lane < nc) L.kept[lane] = v; // scratch: unsorted candidate ids
                __syncthreads();
                for (uint32_t p0 = 0; p0 < nc; p0 += 8) {
                    const uint32_t g = p0 + (uint32_t)grp;
                    const uint32_t other = L.kept[g < nc ? g : nc - 1u];
                    const float d = group_distance<METRIC, FUSED>(ix, L.qv, thdr, other, j);
                    if (g < nc && j == 0) L.dtmp[g] = d;
                }
                __syncthreads();
                const float dmine = (uint32_t)lane < nc ? L.dtmp[lane] : 0.f;
                uint32_t rank = 0;
                for (uint32_t t = 0; t < nc; ++t) { // Candidate order: score, then id (model.rs:55-61)
                    const float dt = L.dtmp[t];
                    const uint32_t it = L.kept[t];
                    rank += (dt < dmine || (dt == dmine && it < v)) ? 1u : 0u;
                }
                __syncthreads();
                if ((uint32_t)lane < nc) { L.cid[rank] = v; L.csc[rank] = dmine; }
                __syncthreads();
                const uint32_t keepn = select_diverse_dev<METRIC, FUSED>(ix, L, nc, maxn, lane);
                bool kept_mine = false;
                for (uint32_t t = 0; t < keepn; ++t) kept_mine |= L.kept[t] == v;
                if ((uint32_t)lane < nc && !kept_mine) dropped_id = v;
                store_canonical(row, stride, L.kept, keepn, lane, true);
            } else if (!present) {
                __syncthreads();
                if ((uint32_t)lane < nc) L.kept[lane] = v;
                __syncthreads();
                store_canonical(row, stride, L.kept, nc, lane, true);
            }
            unlock_row(a.locks, to, lane);
            // every neighbour dropped by the prune loses its edge to `to` as well: the graph stays symmetric
            unsigned long long dm = __ballot(dropped_id != kSentinel);
            while (dm) {
                const uint32_t src = (uint32_t)__builtin_ctzll(dm);
                dm &= dm - 1ull;
                const uint32_t x = __builtin_amdgcn_readlane(dropped_id, src);
                remove_edge_dev(a, (uint32_t)layer, x, to, lane);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ONE node at a time (sequential mode = the reference's order exactly; every upsert; promotions): round 6.  build_select_kernel and
// build_link_kernel evaluate select_diverse lazily on one wavefront -- a chain of dependent row gathers per candidate: 0.33 ms + 2.0 ms of
// a 2.9-ms insert at 600 000 x 768 (profiles/r06o_seq_insert_kernel_stats.csv).  The two kernels below evaluate the distance matrices of
// ALL prunes of a step up front, spread over the layer's workgroups (one 8-lane group per pair, reference summation order), and the last
// workgroup to deliver replays the decisions from registers (hvx_graph_dev.h: replay_rows) -- as the delete steps do (hvx_delete.hip):
//   build_select_seq_kernel   the node's own lists: <= 64 search candidates per layer, 2 016 pairs, one replay per layer;
//   build_link_seq_kernel     add_bidirectional_link for its <= 32 selected neighbours per layer IN SELECTION ORDER.  A link reads the
//                             neighbour's row as earlier links of the same node left it -- and the only thing an earlier link can do to it
//                             is REMOVE an id (a neighbour its prune dropped loses the reverse edge).  So: every link's list (row + the
//                             node) and its matrix are taken from the rows as the kernel finds them -- a superset of what the link will
//                             see --, sixteen wavefronts replay the links speculatively in parallel, and one wavefront then walks them in
//                             order: a link whose row an earlier link has touched is replayed again over the ids that are still there
//                             (replay_rows' `alive` mask), everything else stands.  The removals (row of the dropped id loses the
//                             neighbour) go out last, all victims of a row at once.
// Same distances, same comparisons, same order of decisions as the one-wavefront kernels: tests/test_gpu_build.py holds both to the oracle
// row for row (hvx_build_params.link_mode = 1 selects the one-wavefront kernels).
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kSeqRow = 35;                 // ids of a link's list (row + the node); wider rows take the one-wavefront kernel
constexpr uint32_t kSeqRS = 36;                  // stride of a link's matrix: rows 0 .. nc (row nc: the owner's), columns 0 .. nc - 1
constexpr uint32_t kSeqSelDm = 65 * 64;          // the select's matrix
constexpr uint32_t kSeqLinkDm = kSeqRS * kSeqRS;
constexpr uint32_t kSeqLayerDm = kSeqSelDm + 32u * kSeqLinkDm;
constexpr uint32_t kSeqPairs = 128;              // (row, victim) removals one layer's links can log

template <uint32_t METRIC, bool FUSED, bool BF> __global__ __launch_bounds__(256) void build_select_seq_kernel(BuildArgs a) {
    __shared__ uint32_t s_cid[64], s_wsc[128], s_last;
    const DevIndex &ix = a.ix;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    uint32_t L, g, G;
    if (blockIdx.x < a.g0) { L = 0u; g = blockIdx.x; G = a.g0; }
    else { L = 1u + (blockIdx.x - a.g0) / a.gu; g = (blockIdx.x - a.g0) % a.gu; G = a.gu; }
    const uint32_t node = a.nodes[0];
    const uint32_t lv = ix.level[node];
    const uint32_t top = lv < a.layers - 1u ? lv : a.layers - 1u; // min(node level, old max_layer)
    if (L > top) return; // layers above the old top stay empty rows (mutation.rs:883-894)
    const uint32_t maxn = L == 0u ? a.m0 : a.m;
    const size_t slot = (size_t)L * a.b;
    const uint32_t cnt = a.cand_cnt[slot];
    const uint32_t hyd = cnt < 2u * maxn ? cnt : 2u * maxn; // select_neighbors_heuristic hydrates the first 2*Mmax only
    float *gl = a.gdm + (size_t)L * kSeqLayerDm;
    if (tid < hyd) s_cid[tid] = (uint32_t)a.cand_ids[slot * kCand + tid];
    __syncthreads();
    bool last = g == 0u;
    const uint32_t npairs = hyd * (hyd - (hyd ? 1u : 0u)) / 2u;
    if (npairs != 0u) {
        if (g == 0u && tid < hyd) st_agent(gl + hyd * 64u + tid, a.cand_sc[slot * kCand + tid]); // the owner's row: the search's scores
        const int j = (int)(lane & 7u);
        for (uint32_t q = g * 32u + (tid >> 3); q < npairs; q += G * 32u) {
            uint32_t i, jj;
            pair_of(q, i, jj); // 0 <= jj < i < hyd
            const uint32_t ni = s_cid[i], nj = s_cid[jj];
            const float d = pair_distance<METRIC, FUSED, BF>(ix, ni, nj, j);
            if (j == 0) { st_agent(gl + i * 64u + jj, d); st_agent(gl + jj * 64u + i, d); }
        }
        stores_done(); // the stores have been acknowledged before the ticket is taken
        __syncthreads();
        if (tid == 0) s_last = __hip_atomic_fetch_add(a.tick + L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == G ? 1u : 0u;
        __syncthreads();
        last = s_last != 0u;
        if (last && tid == 0) st_agent(a.tick + L, 0u);
    }
    if (!last || wave != 0u) return;
    const WaveScratch W{s_wsc, s_wsc + 64};
    uint32_t ns = hyd;
    bool bad = false;
    if (hyd >= 2u) ns = replay_rows<64>(gl, s_cid, hyd, maxn, lane, W, &bad);
    else if (lane < hyd) W.kept[lane] = s_cid[lane];
    lds_order();
    const uint32_t kf = lane < ns ? W.kept[lane] : kSentinel;
    if (lane < ns) a.sel[slot * 32u + lane] = kf;
    if (lane == 0) a.sel_cnt[slot] = ns;
    uint32_t stride;
    uint32_t *row = row_ptr(a, node, L, stride);
    store_canonical_reg(row, stride, kf, ns, lane); // nobody else can reach this row before the link step
}

struct SeqLds {
    uint32_t *rcur, *rkept;   // [32][kSeqRS] a link's list (row + the node) / what stays of it
    uint32_t *rdeg, *rkn;     // [32] ids in the list (0: nothing to do) / ids that stay
    uint32_t *rdlo, *rdhi;    // [32] dropped ids by list position
    uint32_t *to;             // [32] the selected neighbours, selection order
    uint32_t *px, *pv;        // [kSeqPairs] removals: row px loses pv
    uint32_t *pbase;          // [34]
    uint32_t *wsc;            // [16][128]
};
__device__ __forceinline__ SeqLds carve_seq(char *smem) {
    SeqLds S;
    uint32_t *p = reinterpret_cast<uint32_t *>(smem);
    S.rcur = p; p += 32 * kSeqRS;
    S.rkept = p; p += 32 * kSeqRS;
    S.rdeg = p; p += 32; S.rkn = p; p += 32; S.rdlo = p; p += 32; S.rdhi = p; p += 32; S.to = p; p += 32;
    S.px = p; p += kSeqPairs; S.pv = p; p += kSeqPairs;
    S.pbase = p; p += 36;
    S.wsc = p;
    return S;
}
static size_t seq_lds_bytes() { return (size_t)(2 * 32 * kSeqRS + 5 * 32 + 2 * kSeqPairs + 36 + 16 * 128) * 4; }

template <uint32_t METRIC, bool FUSED, bool BF> __global__ __launch_bounds__(1024) void build_link_seq_kernel(BuildArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t s_err, s_last, s_np;
    const DevIndex &ix = a.ix;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    uint32_t L, g, G;
    if (blockIdx.x < a.g0) { L = 0u; g = blockIdx.x; G = a.g0; }
    else { L = 1u + (blockIdx.x - a.g0) / a.gu; g = (blockIdx.x - a.g0) % a.gu; G = a.gu; }
    const uint32_t me = a.nodes[0];
    const uint32_t lv = ix.level[me];
    const uint32_t top = lv < a.layers - 1u ? lv : a.layers - 1u;
    if (L > top) return;
    const uint32_t maxn = L == 0u ? a.m0 : a.m;
    const size_t slot = (size_t)L * a.b;
    const uint32_t ns = a.sel_cnt[slot];
    if (ns == 0u) return;
    const SeqLds S = carve_seq(smem);
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (tid == 0) s_err = 0u;
    if (tid < ns) S.to[tid] = a.sel[slot * 32u + tid];
    __syncthreads();
    // ---- the links' lists, from the rows as they are now (every workgroup of the layer arrives at the same lists) ----
    for (uint32_t t = wave; t < ns; t += 16u) {
        const uint32_t to = S.to[t];
        uint32_t stride;
        const uint32_t *row = row_ptr(a, to, L, stride);
        uint32_t v = lane < stride ? ld_row(row + lane) : kSentinel;
        uint32_t deg = (uint32_t)__builtin_popcountll(__ballot(v != kSentinel));
        const bool present = __ballot(v == me) != 0ull;
        if (!present) {
            if (deg + 1u > kSeqRow) { if (lane == 0) s_err = 1u; deg = 0u; }
            else { if (lane == deg) v = me; ++deg; } // rows are canonical: the valid ids occupy lanes 0..deg-1
        } else if (deg > kSeqRow) { if (lane == 0) s_err = 1u; deg = 0u; }
        if (lane < deg) S.rcur[t * kSeqRS + lane] = v;
        if (lane == 0) S.rdeg[t] = deg;
    }
    __syncthreads();
    if (s_err) { if (g == 0u && tid == 0) *a.err = 1u; return; }
    if (wave == 0) { // first pair of every link's prune (a list within its limit needs no matrix)
        const uint32_t rd = lane < ns ? S.rdeg[lane] : 0u;
        const uint32_t np = rd > maxn ? (rd + 1u) * rd / 2u : 0u;
        uint32_t incl = np;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const uint32_t o = __shfl_up(incl, sft, 64);
            if ((int)lane >= sft) incl += o;
        }
        if (lane < 32u) S.pbase[lane] = incl - np;
        if (lane == 31u) S.pbase[32] = incl;
    }
    __syncthreads();
    const uint32_t total = S.pbase[32];
    float *gl = a.gdm + (size_t)L * kSeqLayerDm + kSeqSelDm;
    bool last;
    { // (the ticket is taken even when no list needs a matrix: the last workgroup rewrites rows the others are still reading their lists from)
        const int j = (int)(lane & 7u);
        for (uint32_t p = g * 128u + (tid >> 3); p < total; p += G * 128u) {
            uint32_t lo = 0u, hi = ns; // the link t with pbase[t] <= p < pbase[t + 1]
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (S.pbase[mid] <= p) lo = mid; else hi = mid;
            }
            const uint32_t t = lo;
            uint32_t i, jj;
            pair_of(p - S.pbase[t], i, jj);
            const uint32_t *rc = S.rcur + t * kSeqRS;
            const uint32_t nc = S.rdeg[t];
            const uint32_t node_i = i < nc ? rc[i] : S.to[t], node_j = rc[jj];
            const float d = pair_distance<METRIC, FUSED, BF>(ix, node_i, node_j, j);
            if (j == 0) {
                float *out = gl + (size_t)t * kSeqLinkDm;
                st_agent(out + i * kSeqRS + jj, d);
                if (i < nc) st_agent(out + jj * kSeqRS + i, d);
            }
        }
        stores_done();
        __syncthreads();
        if (tid == 0) s_last = __hip_atomic_fetch_add(a.tick + a.layers + L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == G ? 1u : 0u;
        __syncthreads();
        last = s_last != 0u;
        if (last && tid == 0) st_agent(a.tick + a.layers + L, 0u);
    }
    if (!last) return;
    // ---- the last workgroup: every link replayed on its list as found (sixteen at a time) ...
    const WaveScratch W{S.wsc + wave * 128u, S.wsc + wave * 128u + 64u};
    bool bad = false; // (a distance that is no valid score: add_bidirectional_link does not look, neither does the one-wavefront kernel)
    for (uint32_t t = wave; t < ns; t += 16u) {
        const uint32_t deg = S.rdeg[t];
        const uint32_t *rc = S.rcur + t * kSeqRS;
        uint32_t kn = deg;
        unsigned long long dropped = 0ull;
        if (deg > maxn) kn = replay_rows<(int)kSeqRS>(gl + (size_t)t * kSeqLinkDm, rc, deg, maxn, lane, W, &bad, ~0ull, &dropped);
        else if (lane < deg) W.kept[lane] = rc[lane];
        lds_order();
        if (lane < kn) S.rkept[t * kSeqRS + lane] = W.kept[lane];
        if (lane == 0) { S.rkn[t] = kn; S.rdlo[t] = (uint32_t)dropped; S.rdhi[t] = (uint32_t)(dropped >> 32); }
        lds_order();
    }
    if (tid == 0) s_np = 0u;
    __syncthreads();
    // ---- ... then in selection order: a link whose row an earlier link has taken an id from is replayed over the ids still there
    if (wave == 0) {
        uint32_t np = 0;
        for (uint32_t s = 0; s < ns; ++s) {
            const uint32_t to = S.to[s], deg = S.rdeg[s];
            if (deg == 0u) continue;
            const uint32_t *rc = S.rcur + s * kSeqRS;
            const uint32_t mine = lane < deg ? rc[lane] : kSentinel;
            bool gone = false; // my id has been removed from this row by an earlier link
            for (uint32_t i = 0; i < np; ++i) gone |= S.px[i] == to && S.pv[i] == mine;
            const unsigned long long alive = __ballot(lane < deg && !gone);
            unsigned long long dropped = ((unsigned long long)S.rdhi[s] << 32) | S.rdlo[s];
            if (alive != (deg >= 64u ? ~0ull : (1ull << deg) - 1ull)) {
                const uint32_t nlive = (uint32_t)__builtin_popcountll(alive);
                uint32_t kn;
                dropped = 0ull;
                if (nlive > maxn) {
                    kn = replay_rows<(int)kSeqRS>(gl + (size_t)s * kSeqLinkDm, rc, deg, maxn, lane, W, &bad, alive, &dropped);
                } else {
                    if ((alive >> lane) & 1ull) W.kept[(uint32_t)__builtin_popcountll(alive & lt)] = mine;
                    kn = nlive;
                    lds_order();
                }
                if (lane < kn) S.rkept[s * kSeqRS + lane] = W.kept[lane];
                if (lane == 0) S.rkn[s] = kn;
                lds_order();
            }
            // every neighbour dropped by the prune loses its edge to `to` as well: the graph stays symmetric (mutation.rs:1890-1908)
            const bool drop_mine = ((dropped >> lane) & 1ull) != 0ull;
            const uint32_t at = np + (uint32_t)__builtin_popcountll(dropped & lt);
            if (drop_mine && at < kSeqPairs) { S.px[at] = mine; S.pv[at] = to; }
            np += (uint32_t)__builtin_popcountll(dropped);
            if (np > kSeqPairs) { if (lane == 0) *a.err = 1u; np = kSeqPairs; }
            lds_order();
        }
        if (lane == 0) s_np = np;
    }
    __syncthreads();
    // ---- the rows.  Every neighbour dropped by a prune loses its edge back (mutation.rs:1890-1908): a neighbour row (a link's own row) gets
    // its removals before it is stored -- one store per row, nothing in this kernel reads a row it has written --, every other row that
    // loses edges (the node's own among them) is read, compacted and stored by the first removal that names it
    const uint32_t np = s_np;
    for (uint32_t t = wave; t < ns; t += 16u) {
        if (S.rdeg[t] == 0u) continue;
        const uint32_t kn = S.rkn[t], to = S.to[t];
        const uint32_t v = lane < kn ? S.rkept[t * kSeqRS + lane] : kSentinel;
        bool victim = false;
        for (uint32_t k2 = 0; k2 < np; ++k2) victim |= S.px[k2] == to && S.pv[k2] == v;
        const bool keep = v != kSentinel && !victim;
        const unsigned long long km = __ballot(keep);
        const uint32_t nk = (uint32_t)__builtin_popcountll(km);
        // (compact: the ids that stay move to the low lanes)
        const uint32_t src = (uint32_t)__builtin_amdgcn_ds_permute((int)(((keep ? (uint32_t)__builtin_popcountll(km & lt) : 63u - (uint32_t)__builtin_popcountll(~km & lt))) << 2), (int)v);
        uint32_t stride;
        uint32_t *row = row_ptr(a, to, L, stride);
        if (nk > stride) { if (lane == 0) *a.err = 1u; continue; }
        store_canonical_reg(row, stride, lane < nk ? src : kSentinel, nk, lane);
    }
    for (uint32_t i = wave; i < np; i += 16u) {
        const uint32_t x = S.px[i];
        bool first = true;
        for (uint32_t k2 = 0; k2 < i; ++k2) first &= S.px[k2] != x;
        for (uint32_t t = 0; t < ns; ++t) first &= S.to[t] != x; // (a link's row: done above)
        if (!first) continue;
        uint32_t stride;
        uint32_t *row = row_ptr(a, x, L, stride);
        const uint32_t v = lane < stride ? ld_row(row + lane) : kSentinel;
        bool victim = false;
        for (uint32_t k2 = i; k2 < np; ++k2) victim |= S.px[k2] == x && S.pv[k2] == v;
        const bool keep = v != kSentinel && !victim;
        const unsigned long long km = __ballot(keep);
        const uint32_t pos = (uint32_t)__builtin_popcountll(km & lt), nk = (uint32_t)__builtin_popcountll(km);
        if (keep) st_row(row + pos, v); // every lane holds its id in a register: the order of the stores does not matter
        if (lane >= nk && lane < stride) st_row(row + lane, kSentinel);
    }
}

// ---- step 3, batched mode: one 256-thread workgroup per LINK (new node q, layer, selected neighbour s) ----
// build_link_kernel walks a node's <= 32 links one after the other, and every prune inside it is a chain of ~100 dependent row
// gathers (select_diverse stages candidate i, then scores it against the kept rows eight at a time, stops at the first hit):
// 9.4 ms per 2 048-node batch, 70 % of the build (profiles/history/r02f).  In a batch the order in which links reach the graph is not
// defined anyway, so every link gets its own workgroup, and the prune is evaluated EAGERLY from LDS: the nc <= Mmax + 1 rows of
// the overflowing row and its owner's cross HBM once, and ALL pairwise distances among them (every pair independent of every
// other: 561 pairs for 33 + 1 rows, eight per wavefront step) are computed with the reference's summation order -- the
// distance is symmetric bit for bit (squares / products commute), so the pair set does not depend on the (score, id) order that
// select_diverse walks.  The rows pass through LDS in COLUMN blocks of <= 256 floats: a pair's four AVX-lane accumulators are
// carried in registers from block to block (each lane's fma chain runs over the depth in the same order as in one pass), the
// next block is in flight in registers while this one is being used, and a workgroup holds ~46 KB of LDS: three per CU, so
// the lock / row / store latencies of one link sit under the arithmetic of the others.  Then: owner distances -> Candidate order
// (model.rs:55-61), P[i] bit j = D[c_i][c_j] < D[c_i][owner] (the test of mod.rs:832), and select_diverse + backfill
// (mod.rs:809-856) is a walk over 64-bit masks: candidate i is diverse iff P[i] & kept == 0.  Same decisions as the lazy
// evaluation, bit for bit.
constexpr int kLinkTasks = 18; // wave-steps of 8 pairs per wavefront: 4 x 18 x 8 >= 561 pairs of 33 candidates + owner

struct LinkLds {
    float *rows;              // [ncmax + 1][ldp]: this column block of the candidate rows (row order of the neighbour row), then the owner's
    float *D;                 // [ncmax + 1][ncmax + 1] pairwise distances (index nc = the owner)
    uint32_t *cand;           // [64] ids in row order
    uint32_t *cid;            // [64] ids sorted by (distance to the owner, id)
    float *csc;               // [64] their distances
    uint32_t *srow;           // [64] row index of sorted candidate r
    unsigned long long *P;    // [64] predicate masks, sorted order
    uint32_t *fin;            // [64] ids of the pruned row
    uint32_t *sh;             // [8] nc, prune, present, overflow
    unsigned char *pa, *pb;   // [pairs] the two rows of pair p
};
__device__ __host__ __forceinline__ size_t link_pairs_max(uint32_t ncmax) { return (size_t)(ncmax + 1u) * ncmax / 2u; }
__device__ __forceinline__ LinkLds carve_link(char *smem, uint32_t ldp, uint32_t ncmax) {
    LinkLds L;
    L.rows = reinterpret_cast<float *>(smem);
    char *p = smem + (size_t)(ncmax + 1u) * ldp * 4u;
    L.P = reinterpret_cast<unsigned long long *>(p); p += 512;
    L.D = reinterpret_cast<float *>(p); p += (size_t)(ncmax + 1u) * (ncmax + 1u) * 4u;
    L.cand = reinterpret_cast<uint32_t *>(p); p += 256;
    L.cid = reinterpret_cast<uint32_t *>(p); p += 256;
    L.csc = reinterpret_cast<float *>(p); p += 256;
    L.srow = reinterpret_cast<uint32_t *>(p); p += 256;
    L.fin = reinterpret_cast<uint32_t *>(p); p += 256;
    L.sh = reinterpret_cast<uint32_t *>(p); p += 32;
    L.pa = reinterpret_cast<unsigned char *>(p); p += (link_pairs_max(ncmax) + 15u) & ~(size_t)15u;
    L.pb = reinterpret_cast<unsigned char *>(p);
    return L;
}
static size_t link_lds_bytes(uint32_t ldp, uint32_t ncmax) {
    return (size_t)(ncmax + 1u) * ldp * 4u + 512u + (size_t)(ncmax + 1u) * (ncmax + 1u) * 4u + 5u * 256u + 32u +
           2u * ((link_pairs_max(ncmax) + 15u) & ~(size_t)15u);
}

// the tail of a link runs on ONE wavefront of the workgroup (the others have left): wavefront-level ordering instead of s_barrier
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// Row locks of build_link_wg_kernel.  Everything a lock protects (the neighbour rows) is read and written with agent-scope
// atomics (ld_row / st_row: coherent across the XCDs' L2s by themselves), so taking and dropping a lock needs ORDER only, not
// cache maintenance: no acquire / release at agent scope (on gfx950 that is an L2 invalidate / write-back of the whole XCD per
// link, with hundreds of links in flight), but relaxed atomics and an explicit wait for this wavefront's outstanding stores.
__device__ __forceinline__ void lock_row_w(uint32_t *locks, uint32_t node, int lane) {
    if (lane == 0) {
        while (__hip_atomic_exchange(&locks[node], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_wave_barrier(); // the row is read after lane 0 has left the loop (one wavefront: program order)
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void unlock_row_w(uint32_t *locks, uint32_t node, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every row store of this wavefront has been performed
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(&locks[node], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// canonical row (ascending id, sentinel padded) of ids_lds[0..ns), written with agent-scope stores by one wavefront
__device__ __forceinline__ void store_canonical_w(uint32_t *row, uint32_t stride, const uint32_t *ids_lds, uint32_t ns, int lane) {
    const uint32_t mine = (uint32_t)lane < ns ? ids_lds[lane] : kSentinel;
    uint32_t rank = 0;
    for (uint32_t s = 0; s < ns; ++s) rank += ids_lds[s] < mine ? 1u : 0u;
    if ((uint32_t)lane >= ns && (uint32_t)lane < stride) st_row(row + lane, kSentinel);
    if ((uint32_t)lane < ns) st_row(row + rank, mine);
}
__device__ __forceinline__ void remove_edge_w(const BuildArgs &a, uint32_t layer, uint32_t owner, uint32_t victim, int lane) {
    lock_row_w(a.locks, owner, lane);
    uint32_t stride;
    uint32_t *row = row_ptr(a, owner, layer, stride);
    const uint32_t v = (uint32_t)lane < stride ? ld_row(row + lane) : kSentinel;
    const bool keep = v != kSentinel && v != victim;
    const unsigned long long km = __ballot(keep);
    const uint32_t pos = (uint32_t)__builtin_popcountll(km & ((1ull << lane) - 1ull));
    const uint32_t nk = (uint32_t)__builtin_popcountll(km);
    if (keep) st_row(row + pos, v);   // every lane holds its id in a register: the order of the stores does not matter
    if ((uint32_t)lane >= nk && (uint32_t)lane < stride) st_row(row + lane, kSentinel);
    unlock_row_w(a.locks, owner, lane);
}

template <uint32_t METRIC, bool FUSED> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void build_link_wg_kernel(BuildArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevIndex &ix = a.ix;
    const uint32_t q = blockIdx.x >> 5, s = blockIdx.x & 31u, layer = blockIdx.y;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 3, j = lane & 7;
    const uint32_t me = a.nodes[q];
    const uint32_t lv = ix.level[me];
    const uint32_t top = lv < a.layers - 1u ? lv : a.layers - 1u;
    if (layer > top) return;
    const size_t slot = (size_t)layer * a.b + q;
    if (s >= a.sel_cnt[slot]) return;
    const uint32_t to = a.sel[slot * 32u + s];
    const uint32_t maxn = layer == 0u ? a.m0 : a.m;
    LinkLds L = carve_link(smem, a.ldp, a.ncmax);
    uint32_t stride;
    uint32_t *row = row_ptr(a, to, layer, stride);

    // ---- add_bidirectional_link(from = me, to) (mutation.rs:1498-1583): append under the row owner's lock ----
    if (wave == 0) {
#ifdef HVX_TUNING
        if (a.dbg && lane == 0) { // contention probe: how often the target's lock is found taken
            uint32_t spins = 0;
            while (__hip_atomic_load(&a.locks[to], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u && spins < 1000000u) { ++spins; __builtin_amdgcn_s_sleep(2); }
            atomicAdd(&a.dbg[0], spins);
        }
#endif
        lock_row_w(a.locks, to, lane);
        uint32_t v = (uint32_t)lane < stride ? ld_row(row + lane) : kSentinel;
        uint32_t deg = (uint32_t)__builtin_popcountll(__ballot(v != kSentinel));
        const bool present = __ballot(v == me) != 0ull;
        bool overflow = false;
        if (!present) {
            if (deg >= 64u) overflow = true;
            else {
                if ((uint32_t)lane == deg) v = me; // rows are canonical: the valid ids occupy lanes 0..deg-1
                ++deg;
            }
        }
        if (deg > maxn && deg > a.ncmax) overflow = true; // more rows than the LDS was sized for: cannot happen on rows this build wrote
        L.cand[lane] = v;
        if (lane == 0) {
            L.sh[0] = deg;
            L.sh[1] = (deg > maxn && !overflow) ? 1u : 0u;
            L.sh[2] = present ? 1u : 0u;
            L.sh[3] = overflow ? 1u : 0u;
            if (overflow) *a.err = 1u;
        }
    }
    __syncthreads();
    const uint32_t nc = L.sh[0];
    if (L.sh[1] == 0u) { // no prune: the appended id takes its place in the canonical row
        if (wave == 0) {
            if (L.sh[2] == 0u && L.sh[3] == 0u) store_canonical_w(row, stride, L.cand, nc, lane);
            unlock_row_w(a.locks, to, lane);
            if (lane == 0) HVX_DBG_ADD(a, 3, 1);
        }
        return;
    }

    // ---- all pairwise distances among the nc candidate rows and the owner's row (index nc) ----
    const uint32_t nrows = nc + 1u, npairs = nrows * nc / 2u;
    if ((uint32_t)tid >= 1u && (uint32_t)tid < nrows) { // pair p = b (b - 1) / 2 + a  <->  rows a < b
        const uint32_t b = (uint32_t)tid, base = b * (b - 1u) / 2u;
        for (uint32_t aa = 0; aa < b; ++aa) { L.pa[base + aa] = (unsigned char)aa; L.pb[base + aa] = (unsigned char)b; }
    }
    const uint32_t nk = ix.dim_main >> 5;                 // 32-float chunks of a row (dim == dim_main == ld: the host checked)
    const uint32_t ck = a.link_ck;                        // chunks per column block (even)
    const uint32_t nblocks = (nk + ck - 1u) / ck;
    const uint32_t w4 = ck * 8u;                          // float4 per row and block
    constexpr int kPre = 9;                               // float4 a thread carries for the next block: 34 rows x 64 float4 / 256 threads
    float4 pre[kPre];
#pragma unroll
    for (int u = 0; u < kPre; ++u) pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto prefetch = [&](uint32_t blk) __attribute__((always_inline)) {
        const uint32_t c0 = blk * w4, cw = (nk - blk * ck < ck ? nk - blk * ck : ck) * 8u;
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const uint32_t e = (uint32_t)tid + 256u * (uint32_t)u;
            const uint32_t r = e / w4, c = e - r * w4;
            if (r < nrows && c < cw) {
                const uint32_t node = r < nc ? L.cand[r] : to;
                pre[u] = reinterpret_cast<const float4 *>(ix.vec + (size_t)node * ix.ld)[c0 + c];
            }
        }
    };
    auto commit = [&](uint32_t blk) __attribute__((always_inline)) {
        const uint32_t cw = (nk - blk * ck < ck ? nk - blk * ck : ck) * 8u;
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const uint32_t e = (uint32_t)tid + 256u * (uint32_t)u;
            const uint32_t r = e / w4, c = e - r * w4;
            if (r < nrows && c < cw) reinterpret_cast<float4 *>(L.rows + (size_t)r * a.ldp)[c] = pre[u];
        }
    };
    float4 acc[kLinkTasks];
#pragma unroll
    for (int t = 0; t < kLinkTasks; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int slot4 = chunk_slot(j);
    prefetch(0);
    if (tid < 64) L.P[tid] = 0ull;
    for (uint32_t blk = 0; blk < nblocks; ++blk) {
        commit(blk);
        __syncthreads(); // block blk is in LDS (first round: and the pair table)
        if (blk + 1u < nblocks) prefetch(blk + 1u); // in flight underneath the arithmetic
        const uint32_t ckb = nk - blk * ck < ck ? nk - blk * ck : ck;
#pragma unroll
        for (int t = 0; t < kLinkTasks; ++t) {
            const uint32_t p = ((uint32_t)wave + 4u * (uint32_t)t) * 8u + (uint32_t)grp;
            if (((uint32_t)wave + 4u * (uint32_t)t) * 8u >= npairs) continue; // uniform in the wavefront
            const uint32_t pp = p < npairs ? p : npairs - 1u;
            const float4 *qp = reinterpret_cast<const float4 *>(L.rows + (size_t)L.pa[pp] * a.ldp) + slot4;
            const float4 *rp = reinterpret_cast<const float4 *>(L.rows + (size_t)L.pb[pp] * a.ldp) + slot4;
            float4 ac = acc[t];
#pragma unroll 4
            for (uint32_t k = 0; k < ckb; ++k) {
                const float4 x = rp[k * 8u];
                const float4 qq = qp[k * 8u];
                if (METRIC == kL2) {
                    const float d0 = qq.x - x.x, d1 = qq.y - x.y, d2 = qq.z - x.z, d3 = qq.w - x.w;
                    if (FUSED) {
                        ac.x = __builtin_fmaf(d0, d0, ac.x); ac.y = __builtin_fmaf(d1, d1, ac.y);
                        ac.z = __builtin_fmaf(d2, d2, ac.z); ac.w = __builtin_fmaf(d3, d3, ac.w);
                    } else {
                        ac.x = d0 * d0 + ac.x; ac.y = d1 * d1 + ac.y;
                        ac.z = d2 * d2 + ac.z; ac.w = d3 * d3 + ac.w;
                    }
                } else {
                    if (FUSED) {
                        ac.x = __builtin_fmaf(qq.x, x.x, ac.x); ac.y = __builtin_fmaf(qq.y, x.y, ac.y);
                        ac.z = __builtin_fmaf(qq.z, x.z, ac.z); ac.w = __builtin_fmaf(qq.w, x.w, ac.w);
                    } else {
                        ac.x = qq.x * x.x + ac.x; ac.y = qq.y * x.y + ac.y;
                        ac.z = qq.z * x.z + ac.z; ac.w = qq.w * x.w + ac.w;
                    }
                }
            }
            acc[t] = ac;
        }
        __syncthreads(); // everybody is done with block blk before the next one overwrites it
    }
#pragma unroll
    for (int t = 0; t < kLinkTasks; ++t) {
        const uint32_t p = ((uint32_t)wave + 4u * (uint32_t)t) * 8u + (uint32_t)grp;
        if (((uint32_t)wave + 4u * (uint32_t)t) * 8u >= npairs) continue;
        float r = avx_tree_reduce(acc[t]); // every lane of the group takes part
        if (p < npairs) {
            const uint32_t ra = L.pa[p], rb = L.pb[p];
            if (METRIC == kCosine) {
                const uint32_t na = ra < nc ? L.cand[ra] : to, nb = rb < nc ? L.cand[rb] : to;
                r = cosine_finish(r, ix.hdr[na], ix.hdr[nb], ix.vec + (size_t)na * ix.ld, ix.vec + (size_t)nb * ix.ld, ix.dim);
            }
            if (j == 0) { L.D[ra * nrows + rb] = r; L.D[rb * nrows + ra] = r; }
        }
    }
    __syncthreads();

    // ---- rank the row's neighbours by distance to its owner (Candidate order: score, then id; model.rs:55-61) ----
    if ((uint32_t)tid < nc) {
        const float dmine = L.D[nc * nrows + (uint32_t)tid];
        const uint32_t v = L.cand[tid];
        uint32_t rank = 0;
        for (uint32_t t = 0; t < nc; ++t) {
            const float dt = L.D[nc * nrows + t];
            const uint32_t it = L.cand[t];
            rank += (dt < dmine || (dt == dmine && it < v)) ? 1u : 0u;
        }
        L.cid[rank] = v;
        L.csc[rank] = dmine;
        L.srow[rank] = (uint32_t)tid;
    }
    __syncthreads();
    if (wave != 0) return;
    // ---- P[i] bit jj = dist(c_i, c_jj) < dist(c_i, owner), jj < i in sorted order (strict <: mod.rs:832) ----
    if ((uint32_t)lane < nc) {
        const uint32_t ri = L.srow[lane];
        const float si = L.csc[lane];
        unsigned long long bits = 0ull;
        for (uint32_t jj = 0; jj < (uint32_t)lane; ++jj)
            if (L.D[ri * nrows + L.srow[jj]] < si) bits |= 1ull << jj;
        L.P[lane] = bits;
    }
    wave_sync();

    // ---- select_diverse + backfill over the masks (mod.rs:809-856); all lanes walk the same chain ----
    unsigned long long kept = 0ull;
    uint32_t ns = 0;
    for (uint32_t i = 0; i < nc && ns < maxn; ++i)
        if ((L.P[i] & kept) == 0ull) { kept |= 1ull << i; ++ns; }
    for (uint32_t i = 0; i < nc && ns < maxn; ++i)
        if (((kept >> i) & 1ull) == 0ull) { kept |= 1ull << i; ++ns; }
    const bool have = (uint32_t)lane < nc;
    const uint32_t mine = have ? L.cid[lane] : kSentinel;
    const bool in = have && ((kept >> lane) & 1ull) != 0ull;
    const unsigned long long im = __ballot(in);
    if (in) L.fin[__builtin_popcountll(im & ((1ull << lane) - 1ull))] = mine;
    const uint32_t dropped_id = (have && !in) ? mine : kSentinel;
    wave_sync();
    store_canonical_w(row, stride, L.fin, ns, lane);
    unlock_row_w(a.locks, to, lane);
    // every neighbour dropped by the prune loses its edge to `to` as well (mutation.rs:1890-1908): the graph stays symmetric
    unsigned long long dm = __ballot(dropped_id != kSentinel);
    while (dm) {
        const uint32_t src = (uint32_t)__builtin_ctzll(dm);
        dm &= dm - 1ull;
        const uint32_t x = __builtin_amdgcn_readlane(dropped_id, src);
        remove_edge_w(a, layer, x, to, lane);
        if (lane == 0) HVX_DBG_ADD(a, 2, 1);
    }
    if (lane == 0) HVX_DBG_ADD(a, 1, 1);
}

template <typename K> static hipError_t launch_build(K kern, dim3 grid, const BuildArgs &a, hipStream_t s) {
    const size_t lds = build_lds_bytes(a.ix.ld);
    hipLaunchKernelGGL(kern, grid, dim3(64), lds, s, a);
    return hipGetLastError();
}

template <typename K> static hipError_t launch_link_wg(K kern, const BuildArgs &a, uint32_t layers, size_t lds, hipStream_t s) {
    if (lds > 48 * 1024) { // (46 KB at <= 33 candidate rows: only larger Mmax would need it; the attribute is per function AND device)
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(a.b * 32u, layers), dim3(256), lds, s, a);
    return hipGetLastError();
}

// insertion order: position i inserts row (i * stride) mod n; stride 1 = id order, a stride coprime with n = a permutation that puts
// the nodes of a batch far apart in id space (hvx_build_params.scatter)
__global__ void iota_kernel(uint32_t *p, uint32_t count, uint32_t first, uint32_t stride, uint32_t mod) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < count) p[i] = stride == 1u ? first + i : (uint32_t)(((unsigned long long)(first + i) * stride) % mod);
}

// one instantiation per (metric, summation tree): the reference picks both per index (spaces/*.rs, distance/*.rs)
using BuildKernel = void (*)(BuildArgs);
struct BuildKernels {
    BuildKernel select, link, link_wg; // link_wg: null where the workgroup kernel has no instantiation (Manhattan)
    BuildKernel select_seq, link_seq;  // one node per step
};
template <uint32_t METRIC, bool FUSED> static BuildKernels build_kernels_of() {
    BuildKernels k{build_select_kernel<METRIC, FUSED>, build_link_kernel<METRIC, FUSED>, nullptr, build_select_seq_kernel<METRIC, FUSED, false>,
                   build_link_seq_kernel<METRIC, FUSED, false>};
    if constexpr (METRIC != kL1) k.link_wg = build_link_wg_kernel<METRIC, FUSED>;
    return k;
}
// bf16 images (round 6): one node at a time only -- the many-workgroup select / link steps over the interleaved bf16 rows
template <uint32_t METRIC> static BuildKernels build_kernels_bf16() {
    return BuildKernels{nullptr, nullptr, nullptr, build_select_seq_kernel<METRIC, true, true>, build_link_seq_kernel<METRIC, true, true>};
}
static BuildKernels pick_build_kernels(uint32_t metric, bool fused, bool bf16 = false) {
    if (bf16) return metric == kL2 ? build_kernels_bf16<kL2>() : build_kernels_bf16<kCosine>();
    if (metric == kL2) return fused ? build_kernels_of<kL2, true>() : build_kernels_of<kL2, false>();
    if (metric == kCosine) return fused ? build_kernels_of<kCosine, true>() : build_kernels_of<kCosine, false>();
    return fused ? build_kernels_of<kL1, true>() : build_kernels_of<kL1, false>();
}

} // namespace hvx

extern "C" void hvx_build_params_default(hvx_build_params *p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->ef_construction = 200; // DEFAULT_EF_CONSTRUCTION (mod.rs:702-708)
    p->max_batch = 2048;
    p->batch_divisor = 32;
}

// insert_hnsw (mutation.rs:787-895) for the rows at insertion positions [first, first + count) of an image that already holds their
// vectors, levels and (empty) neighbour rows: batches of nodes searched on the handle's stream while the previous batch is selected
// and linked on a second one.  hvx_index_build runs it over a whole image (positions 0 .. n - 1, optionally scattered);
// hvx_index_insert_batch over the rows it has just appended (id order).  levels_in[0] is the level of row level_row0.  Position p inserts row
// (p * stride) mod `mod` (stride 1: row p).  The handle is private to the caller for the duration (its lock is held or it has not
// been returned yet).
static int insert_range(hvx_index *ix, uint64_t first, uint64_t count, const uint16_t *levels_in, uint64_t level_row0, const hvx_build_params *params,
                        uint32_t stride, uint64_t mod, hvx_build_stats *stats, const float *build_q = nullptr) {
    // build_q (bf16 images): the nodes' ROUNDED vectors as f32 [count][dim] on the device -- the queries of their build searches
    if (count == 0) return HVX_OK;
    struct LevelOf { // levels_in[0] is the level of row level_row0
        const uint16_t *p;
        uint64_t r0;
        explicit operator bool() const { return p != nullptr; }
        uint16_t operator[](uint64_t row) const { return p[row - r0]; }
    } levels{levels_in, level_row0};
    const uint32_t m = ix->desc.m ? ix->desc.m : 16u;
    const uint32_t m0 = std::max(ix->desc.m0 ? ix->desc.m0 : 2u * m, 2u * m);
    const uint32_t efc = params->ef_construction ? params->ef_construction : 200u;
    const uint32_t ef0 = std::max(efc, m0), efu = std::max(efc, 2u * m);
    uint32_t bmax = params->max_batch ? params->max_batch : 2048u;
    if (bmax > ix->max_batch) bmax = ix->max_batch; // per-batch scratch of the search kernel is sized by max_batch
    const uint64_t rows_total = std::max<uint64_t>(first + count, mod);
    auto row_at = [&](uint64_t pos) -> uint64_t { return stride == 1 ? pos : (pos * stride) % mod; };
    DevIndex &d = ix->dev;
    uint32_t top_level = d.has_entry ? d.max_layer : 0u;
    for (uint64_t p = first; p < first + count; ++p) top_level = std::max<uint32_t>(top_level, levels ? levels[row_at(p)] : 0);
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t s = ix->stream;
    const uint32_t layers_max = top_level + 1u;
    uint32_t *d_iota, *d_locks, *d_cnt, *d_sel, *d_selcnt, *d_status, *d_err, *d_tick;
    uint64_t *d_cids;
    float *d_csc, *d_gdm;
    // Round 6: the scratch, the link stream and its events stay with the handle (a one-node insert / upsert paid nine hipMalloc + hipFree --
    // each a device synchronisation --, a stream and four events per call: ~2 ms of a 6.4-ms upsert).  Only tuning buffers are per call.
    // Round 4: the search of batch i + 1 runs on the handle's stream WHILE batch i is selected and linked on a second stream.  Batch i + 1
    // then does not see batch i -- which it tolerates exactly as the nodes of one batch tolerate not seeing each other: rows are only ever
    // read as stale-or-current (agent-scope stores by the link step, immutable vectors), never torn into invalid ids.  Candidate /
    // selection buffers are double-buffered; a promotion (new top layer = new entry point), a one-node batch and sequential mode do not overlap.
    std::vector<void *> scratch;
    auto salloc = [&](void **p, size_t bytes) -> int {
        if (hipMalloc(p, std::max<size_t>(bytes, 16)) != hipSuccess) return fail(HVX_ERR_DEVICE, "hipMalloc(%zu) build scratch", bytes);
        scratch.push_back(*p);
        return HVX_OK;
    };
    auto release = [&]() { for (void *p : scratch) (void)hipFree(p); scratch.clear(); };
    auto sbail = [&](int code) { // (a failed call may leave locks / tickets set: the next call starts from fresh scratch)
        (void)hipStreamSynchronize(s);
        release();
        ix->ins_cap = 0;
        ix->ins_locks_rows = 0;
        return code;
    };
    const size_t sz_cids = (size_t)layers_max * bmax * kCand, sz_cnt = (size_t)layers_max * bmax, sz_sel = (size_t)layers_max * bmax * 32;
    int rc;
    {
        auto up = [](size_t b) { return (b + 255u) & ~(size_t)255u; };
        const size_t b_iota = up(count * 4), b_cids = up(2 * sz_cids * 8), b_csc = up(2 * sz_cids * 4), b_cnt = up(2 * sz_cnt * 4), b_sel = up(2 * sz_sel * 4),
                     b_status = up((size_t)bmax * 4), b_err = 256, b_tick = up(2 * 64 * 4), b_gdm = up((size_t)layers_max * kSeqLayerDm * 4);
        const size_t need = b_iota + b_cids + b_csc + 2 * b_cnt + b_sel + b_status + b_err + b_tick + b_gdm;
        if (need > ix->ins_cap) {
            ix->ins_cap = 0;
            if ((rc = ix->regrow(&ix->ins_scratch, need + need / 4))) return rc;
            if (hipMemsetAsync(ix->ins_scratch, 0, need + need / 4, s) != hipSuccess) return fail(HVX_ERR_DEVICE, "memset");
            ix->ins_cap = need + need / 4;
        }
        char *p = reinterpret_cast<char *>(ix->ins_scratch);
        d_tick = reinterpret_cast<uint32_t *>(p); p += b_tick; // (first: zero between calls, set once by the memset above)
        d_iota = reinterpret_cast<uint32_t *>(p); p += b_iota;
        d_cids = reinterpret_cast<uint64_t *>(p); p += b_cids;
        d_csc = reinterpret_cast<float *>(p); p += b_csc;
        d_cnt = reinterpret_cast<uint32_t *>(p); p += b_cnt;
        d_selcnt = reinterpret_cast<uint32_t *>(p); p += b_cnt;
        d_sel = reinterpret_cast<uint32_t *>(p); p += b_sel;
        d_status = reinterpret_cast<uint32_t *>(p); p += b_status;
        d_err = reinterpret_cast<uint32_t *>(p); p += b_err;
        d_gdm = reinterpret_cast<float *>(p);
        const uint64_t lock_rows = std::max<uint64_t>(rows_total, ix->cap_rows);
        if (lock_rows > ix->ins_locks_rows) { // one lock per row the image can hold: all zero between calls (every lock taken is released)
            ix->ins_locks_rows = 0;
            if ((rc = ix->regrow((void **)&ix->ins_locks, lock_rows * 4))) return rc;
            if (hipMemsetAsync(ix->ins_locks, 0, lock_rows * 4, s) != hipSuccess) return fail(HVX_ERR_DEVICE, "memset");
            ix->ins_locks_rows = lock_rows;
        }
        d_locks = ix->ins_locks;
    }
    if (!ix->ins_stream && hipStreamCreateWithFlags(&ix->ins_stream, hipStreamNonBlocking) != hipSuccess) return fail(HVX_ERR_DEVICE, "stream creation failed");
    for (int i = 0; i < 4; ++i)
        if (!ix->ins_ev[i] && hipEventCreateWithFlags(&ix->ins_ev[i], hipEventDisableTiming) != hipSuccess) return fail(HVX_ERR_DEVICE, "event creation failed");
    hipStream_t s2 = ix->ins_stream;
    hipEvent_t ev_search[2] = {ix->ins_ev[0], ix->ins_ev[1]}, ev_link[2] = {ix->ins_ev[2], ix->ins_ev[3]};
    bool link_pending[2] = {false, false};
    auto drop_streams = [&]() { (void)hipStreamSynchronize(s2); };
    auto sbail2 = [&](int code) { (void)hipStreamSynchronize(s); drop_streams(); return sbail(code); };
    uint32_t *d_dbg = nullptr;
    if (tuning_env("HVX_BUILD_DEBUG")) {
        if ((rc = salloc((void **)&d_dbg, 32))) return sbail2(rc);
        if (hipMemsetAsync(d_dbg, 0, 32, s) != hipSuccess) return sbail2(fail(HVX_ERR_DEVICE, "memset"));
    }
    hipLaunchKernelGGL(iota_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, s, d_iota, (uint32_t)count, (uint32_t)first, stride, (uint32_t)std::max<uint64_t>(mod, 1));
    if (hipMemsetAsync(d_err, 0, 4, s) != hipSuccess) return sbail2(fail(HVX_ERR_DEVICE, "memset"));

    uint32_t *l0w = const_cast<uint32_t *>(d.l0), *upw = const_cast<uint32_t *>(d.up);
    const bool fused = kernel_fused(d.fkernel);
    // build_link_wg_kernel: column blocks of <= 8 chunks (256 floats) of Mmax + 1 candidate rows and the owner's in LDS, row stride
    // = 128 B mod 256 B (the eight row groups of a wavefront read different rows: conflict-free ds_read_b128 for neighbouring rows)
    const uint32_t ncmax = std::max(m0, m) + 1u;
    const uint32_t nk_rows = d.dim_main >> 5;
    const uint32_t link_ck = std::min<uint32_t>(8u, (nk_rows + 1u) & ~1u);
    const uint32_t ldp = link_ck * 32u + 32u;
    const size_t link_lds = link_lds_bytes(ldp, ncmax);
    // serves rows without a scalar tail (dim % 32 == 0, no padding) and <= 33 candidates (561 pairs = 4 wavefronts x 18 steps x 8)
    const bool bf16 = d.dtype == HVX_BF16;
    if (bf16 && (!build_q || stride != 1u)) return sbail2(fail(HVX_ERR_INVARIANT, "bf16 rows are linked from their rounded f32 vectors"));
    const BuildKernels kern = pick_build_kernels(d.metric, fused, bf16);
    const bool link_wg = params->link_mode != 1u && kern.link_wg && !kernel_w4(d.fkernel) /* 32-lane tree only */ && ncmax <= 33u && nk_rows > 0 && d.dim_main == d.dim && d.ld == d.dim &&
                         (size_t)(ncmax + 1u) * link_ck * 8u <= 9u * 256u;
    uint64_t done = first, batches = 0, singles = 0;
    if (!d.has_entry) { // the first node of an index: the entry point with empty rows on its layers (mutation.rs:706-739)
        d.has_entry = 1;
        d.entry = (uint32_t)row_at(first);
        d.max_layer = levels ? levels[row_at(first)] : 0;
        done = first + 1;
        singles = 1;
    }
    const uint32_t divisor = params->batch_divisor ? params->batch_divisor : 32u;
    bool prev_serial = true; // the previous batch must be complete before the next search starts
    // the iota / lock / err initialisation above ran on `s`: the link stream starts behind it
    if (hipEventRecord(ev_search[0], s) != hipSuccess || hipStreamWaitEvent(s2, ev_search[0], 0) != hipSuccess)
        return sbail2(fail(HVX_ERR_DEVICE, "stream ordering failed"));
    const uint64_t end = first + count;
    while (done < end) {
        // batch = consecutive positions of the insertion order; a node above the current top layer is inserted alone and becomes the entry point
        uint32_t bsz = 1;
        const uint16_t lv0 = levels ? levels[row_at(done)] : 0;
        const bool promotes = lv0 > d.max_layer;
        if (!promotes && !params->sequential && !bf16) { // (bf16 images: one node per step)
            uint64_t want = std::min<uint64_t>(std::max<uint64_t>(done / divisor, 1), bmax);
            want = std::min<uint64_t>(want, end - done);
            while (bsz < want && !((levels ? levels[row_at(done + bsz)] : 0) > d.max_layer)) ++bsz;
        }
        const uint32_t pb = (uint32_t)(batches & 1u); // buffer set of this batch
        const bool serial = promotes || bsz == 1u || params->sequential != 0u || params->link_mode == 1u;
        // the buffers of set pb were last read by the link step of batch - 2; and a serial batch (or the batch behind one) starts
        // only when everything before it is in the graph
        if (link_pending[pb] && hipStreamWaitEvent(s, ev_link[pb], 0) != hipSuccess) return sbail2(fail(HVX_ERR_DEVICE, "stream ordering failed"));
        if ((serial || prev_serial) && link_pending[pb ^ 1u] && hipStreamWaitEvent(s, ev_link[pb ^ 1u], 0) != hipSuccess)
            return sbail2(fail(HVX_ERR_DEVICE, "stream ordering failed"));
        const uint32_t layers = d.max_layer + 1u; // old max_layer + 1
        uint64_t *b_cids = d_cids + (size_t)pb * sz_cids;
        float *b_csc = d_csc + (size_t)pb * sz_cids;
        uint32_t *b_cnt = d_cnt + (size_t)pb * sz_cnt, *b_sel = d_sel + (size_t)pb * sz_sel, *b_selcnt = d_selcnt + (size_t)pb * sz_cnt;
        HnswArgs a{};
        a.ix = d;
        a.bitmap = ix->d_bitmap;
        a.words_per_query = ix->words_per_query;
        a.k = kCand;
        a.ef = ef0;
        a.build_ef_upper = efu;
        a.out_ids = b_cids;
        a.out_scores = b_csc;
        a.out_counts = b_cnt;
        a.out_status = d_status;
        a.tie_flags = ix->d_tie;
        a.build_nodes = d_iota + (done - first);
        if (bf16) a.queries = build_q + (size_t)(done - first) * d.dim;
        a.occupancy = (bsz > 1024u && params->link_mode != 1u) ? 2 : 1; // more nodes than SIMDs: two searches per SIMD instead of two rounds
        if (launch_hnsw_wave(a, bsz, s) != hipSuccess) return sbail2(fail(HVX_ERR_DEVICE, "build search launch failed: %s", hipGetErrorString(hipGetLastError())));
        if (hipEventRecord(ev_search[pb], s) != hipSuccess || hipStreamWaitEvent(s2, ev_search[pb], 0) != hipSuccess)
            return sbail2(fail(HVX_ERR_DEVICE, "stream ordering failed"));
        BuildArgs ba{};
        ba.ix = d;
        ba.l0 = l0w;
        ba.up = upw;
        ba.locks = d_locks;
        ba.nodes = d_iota + (done - first);
        ba.b = bsz;
        ba.layers = layers;
        ba.cand_ids = b_cids;
        ba.cand_sc = b_csc;
        ba.cand_cnt = b_cnt;
        ba.sel = b_sel;
        ba.sel_cnt = b_selcnt;
        ba.m = m;
        ba.m0 = m0;
        ba.err = d_err;
        ba.dbg = d_dbg;
        // one node: its select and its links as two many-workgroup steps with every prune's distance matrix evaluated up front
        const bool seq_step = bsz == 1u && (params->link_mode != 1u || bf16) && d.s0 + 1u <= kSeqRow && d.su + 1u <= kSeqRow;
        if (bf16 && !seq_step) return sbail2(fail(HVX_ERR_UNSUPPORTED, "a bf16 image links rows of at most %u ids", kSeqRow - 1u));
        hipError_t e;
        if (seq_step) {
            ba.gdm = d_gdm;
            ba.tick = d_tick;
            ba.g0 = 63u; ba.gu = 16u; // 32 row groups per workgroup: 2 016 pairs among 64 candidates on layer 0, 496 among 32 above
            hipLaunchKernelGGL(kern.select_seq, dim3(ba.g0 + (layers - 1u) * ba.gu), dim3(256), 0, s2, ba);
            ba.g0 = 96u; ba.gu = 12u; // 128 row groups per workgroup: <= 32 links x 561 pairs on layer 0, <= 16 x 153 above
            hipLaunchKernelGGL(kern.link_seq, dim3(ba.g0 + (layers - 1u) * ba.gu), dim3(1024), seq_lds_bytes(), s2, ba);
            e = hipGetLastError();
        } else {
            e = launch_build(kern.select, dim3(bsz, layers), ba, s2);
        }
        if (e == hipSuccess && !seq_step) {
            if (bsz > 1u && link_wg) { // batched mode: one workgroup per link, prunes evaluated from LDS
                ba.ldp = ldp;
                ba.ncmax = ncmax;
                ba.link_ck = link_ck;
                e = launch_link_wg(kern.link_wg, ba, layers, link_lds, s2);
            } else { // one node (the reference's order exactly), or rows the workgroup kernel does not serve: one wavefront per node
                e = launch_build(kern.link, dim3(bsz), ba, s2);
            }
        }
        if (e != hipSuccess) return sbail2(fail(HVX_ERR_DEVICE, "build launch failed: %s", hipGetErrorString(e)));
        if (hipEventRecord(ev_link[pb], s2) != hipSuccess) return sbail2(fail(HVX_ERR_DEVICE, "stream ordering failed"));
        link_pending[pb] = true;
        prev_serial = serial;
        if (promotes) { // mutation.rs:769-772
            d.entry = (uint32_t)row_at(done);
            d.max_layer = lv0;
        }
        done += bsz;
        batches += 1;
        singles += bsz == 1 ? 1 : 0;
        if ((batches & 63u) == 0u && (hipStreamSynchronize(s) != hipSuccess || hipStreamSynchronize(s2) != hipSuccess)) // bound the launch queues
            return sbail2(fail(HVX_ERR_DEVICE, "build kernels failed: %s", hipGetErrorString(hipGetLastError())));
    }
    if (hipStreamSynchronize(s2) != hipSuccess) return sbail2(fail(HVX_ERR_DEVICE, "build did not complete: %s", hipGetErrorString(hipGetLastError())));
    drop_streams();
    uint32_t err = 0;
    if (hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return sbail(fail(HVX_ERR_DEVICE, "build did not complete: %s", hipGetErrorString(hipGetLastError())));
    if (d_dbg) {
        uint32_t h[8] = {0};
        (void)hipMemcpy(h, d_dbg, 32, hipMemcpyDeviceToHost);
        fprintf(stderr, "[hvx build] link step: lock spins %u, prunes %u, reverse-edge removals %u, plain appends %u\n", h[0], h[1], h[2], h[3]);
    }
    release();
    if (err) return fail(HVX_ERR_INVARIANT, "a neighbour row overflowed its stride during the build");
    if (stats) {
        stats->batches += batches;
        stats->single_node_batches += singles;
        stats->nodes += count;
    }
    return HVX_OK;
}

extern "C" int hvx_index_build(const hvx_index_desc *desc, const uint64_t *node_ids, const float *vectors, const uint16_t *levels,
                               const hvx_build_params *params, hvx_index **out, hvx_build_stats *stats) {
    if (!desc || !out || !params) return fail(HVX_ERR_INVARIANT, "null argument");
    *out = nullptr;
    const uint64_t n = desc->n;
    if (n && (!node_ids || !vectors)) return fail(HVX_ERR_INVARIANT, "null array");
    const uint32_t m = desc->m ? desc->m : 16u;
    const uint32_t m0 = std::max(desc->m0 ? desc->m0 : 2u * m, 2u * m); // MutationDegreeLimits (mutation.rs:178-196)
    const uint32_t efc = params->ef_construction ? params->ef_construction : 200u;
    if (desc->dtype != HVX_F32) return fail(HVX_ERR_UNSUPPORTED, "the device build reads f32 rows (import the built graph with a reduced-precision dtype afterwards)");
    if (m0 > 32u || m > 32u) return fail(HVX_ERR_UNSUPPORTED, "device build serves m0 <= 32");
    if (std::max(efc, m0) + 32u > 832u) return fail(HVX_ERR_UNSUPPORTED, "device build serves ef_construction <= 800");
    const uint32_t ef0 = std::max(efc, m0);

    // ---- the image: rows + EMPTY graph with rows sized for m0 / m (+ room for rows appended later: hvx_index_insert_batch) ----
    hvx_index_desc d0 = *desc;
    d0.m = m;
    d0.m0 = m0;
    d0.has_entry = 0;
    d0.max_layer = 0;
    if (d0.max_batch == 0) d0.max_batch = 1024;
    uint64_t up_rows = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint16_t lv = levels ? levels[i] : 0;
        if (lv > 63) return fail(HVX_ERR_INVARIANT, "node level > 63");
        up_rows += lv;
    }
    std::vector<uint64_t> zeros0(n + 1, 0), zerosu(up_rows + 1, 0);
    uint64_t dummy = 0;
    hvx_index *ix = nullptr;
    int rc = import_index(&d0, node_ids, vectors, zeros0.data(), &dummy, levels, zerosu.data(), &dummy, m0, m, &ix, params->reserve_rows,
                          params->reserve_upper_rows);
    if (rc) return rc;
    auto bail = [&](int code) { hvx_index_free(ix); return code; };
    HnswArgs probe{};
    probe.ix = ix->dev;
    probe.ef = ef0;
    // the unrolled builds serve L2 / cosine, the AVX+FMA tree, dim in {128,...,1536}, ef_construction <= 352; everything else (any
    // dimension, Manhattan, the scalar / AVX summation trees, ef_construction <= 800) takes the GENERIC build of the same kernel
    if (!hnsw_wave_supported(probe) && (ix->dev.s0 > 64u || ix->dev.su > 64u))
        return bail(fail(HVX_ERR_UNSUPPORTED, "device build serves neighbour rows of <= 64 ids"));
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n == 0) { *out = ix; return HVX_OK; }
    // scatter: rows whose ORDER follows the data (a dump sorted by topic, an index hydrated in key order) would put each other's nearest
    // neighbours into one batch, where they cannot see each other: insert in the order (i * stride) mod n instead, stride ~ 0.618 n and
    // coprime with n (position 0 stays row 0).  Sequential mode keeps id order: it IS the reference's order.
    uint32_t stride = 1;
    if (params->scatter && !params->sequential && n > 2) {
        auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
        uint64_t st = (uint64_t)((double)n * 0.6180339887498949);
        while (st > 1 && gcd(st, n) != 1) --st;
        stride = (uint32_t)std::max<uint64_t>(st, 1);
    }
    if ((rc = insert_range(ix, 0, n, levels, 0, params, stride, n, stats))) return bail(rc);
    ix->desc.has_entry = 1;
    ix->desc.entry_point = node_ids[ix->dev.entry];
    ix->desc.max_layer = ix->dev.max_layer;
    ix->publish_view(/*bump=*/false); // generation 1 = the built image (import_index published the empty graph under the same number)
    *out = ix;
    return HVX_OK;
}

// insert_hnsw for rows APPENDED to a live image (mutation.rs:642-780 insert -> :787-895 insert_hnsw): the reference inserts one node at
// a time into the store its readers snapshot; here a batch of new rows lands in the spare capacity of a growable image
// (hvx_build_params.reserve_rows), is validated and given its headers / SimHash rows like imported rows, and is then linked into the
// graph by the SAME loop the device build runs -- sequential mode = the reference's insertion, row for row.  The new generation
// becomes visible (visible_seq + 1: rows, entry point, top layer) when the batch is complete; forks adopt it with hvx_index_refresh.
// Searches that run on forks meanwhile keep their generation's entry point and read neighbour rows as stale-or-current, never torn:
// a new node's vector, header and own rows are complete before the first link to it is stored (hvx_build.hip, row locks +
// agent-scope stores), exactly what the overlapped batches of the device build rely on.
extern "C" int hvx_index_insert_batch(hvx_index *ix, const uint64_t *node_ids, const float *vectors, const uint16_t *levels, uint32_t count,
                                      const hvx_build_params *params, hvx_build_stats *stats) {
    if (!ix || (count && (!node_ids || !vectors))) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_build_params dflt;
    hvx_build_params_default(&dflt);
    if (!params) params = &dflt;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (count == 0) return HVX_OK;
    if (ix->is_fork) return fail(HVX_ERR_UNSUPPORTED, "rows are inserted through the handle that owns the image, not a fork");
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    DevIndex &d = ix->dev;
    if (d.dtype != HVX_F32 && d.dtype != HVX_BF16) return fail(HVX_ERR_UNSUPPORTED, "rows are inserted into f32 and bf16 images (fp8 images are read-only)");
    const bool bf16 = d.dtype == HVX_BF16; // (round 6: rounded, validated, packed, linked one node at a time over the bf16 rows)
    const uint32_t m = ix->desc.m ? ix->desc.m : 16u;
    const uint32_t m0 = std::max(ix->desc.m0 ? ix->desc.m0 : 2u * m, 2u * m);
    const uint32_t efc = params->ef_construction ? params->ef_construction : 200u;
    if (m0 > 32u || m > 32u || d.s0 < m0 || d.su < m) return fail(HVX_ERR_UNSUPPORTED, "the image's neighbour rows are narrower than the degree limits (build it with hvx_index_build)");
    if (std::max(efc, m0) + 32u > 832u) return fail(HVX_ERR_UNSUPPORTED, "device build serves ef_construction <= 800");
    const uint64_t n0 = d.n;
    if (n0 + count > ix->cap_rows)
        return fail(HVX_ERR_CANDIDATE_LIMIT, "the image holds %llu of %llu rows: %u more do not fit (hvx_build_params.reserve_rows)", (unsigned long long)n0,
                    (unsigned long long)ix->cap_rows, count);
    uint64_t up_need = 0;
    for (uint32_t i = 0; i < count; ++i) {
        const uint16_t lv = levels ? levels[i] : 0;
        if (lv > 63) return fail(HVX_ERR_INVARIANT, "node level > 63");
        up_need += lv;
        // ids stay strictly ascending over the whole image (the reference allocates dense ascending u64 ids: index_lifecycle_scale.rs:1504-1514)
        const uint64_t prev = i ? node_ids[i - 1] : (n0 ? ix->ids_ref()[n0 - 1] : 0);
        if ((i || n0) && node_ids[i] <= prev) return fail(HVX_ERR_INVARIANT, "inserted node ids must be ascending and above every id of the image (id %llu)", (unsigned long long)node_ids[i]);
    }
    if (ix->up_rows_used + up_need > ix->cap_up_rows)
        return fail(HVX_ERR_CANDIDATE_LIMIT, "the image's upper-layer rows are exhausted (%llu + %llu > %llu: hvx_build_params.reserve_upper_rows)",
                    (unsigned long long)ix->up_rows_used, (unsigned long long)up_need, (unsigned long long)ix->cap_up_rows);
    HnswArgs probe{};
    probe.ix = d;
    probe.ef = std::max(efc, m0);
    if (!hnsw_wave_supported(probe) && (d.s0 > 64u || d.su > 64u)) return fail(HVX_ERR_UNSUPPORTED, "device build serves neighbour rows of <= 64 ids");
    hipStream_t s = ix->stream;
    // ---- the rows: upload into the spare capacity, validate, headers (nothing is visible yet: d.n still ends before them) ----
    float *vdst;
    if (bf16) { // the rounded vectors as f32: validation, headers, SimHash rows and the build searches read them; the image gets the packed rows
        if ((size_t)count * d.ld * 4 > ix->ins_rows_cap) {
            ix->ins_rows_cap = 0;
            int rc0 = ix->regrow((void **)&ix->ins_rows, (size_t)count * d.ld * 4);
            if (rc0) return rc0;
            ix->ins_rows_cap = (size_t)count * d.ld * 4;
        }
        vdst = ix->ins_rows;
    } else {
        vdst = const_cast<float *>(d.vec) + (size_t)n0 * d.ld;
    }
    if (d.ld == d.dim) {
        HIP_TRY(hipMemcpyAsync(vdst, vectors, (size_t)count * d.dim * 4, hipMemcpyDefault, s));
    } else {
        HIP_TRY(hipMemsetAsync(vdst, 0, (size_t)count * d.ld * 4, s));
        HIP_TRY(hipMemcpy2DAsync(vdst, (size_t)d.ld * 4, vectors, (size_t)d.dim * 4, (size_t)d.dim * 4, count, hipMemcpyDefault, s));
    }
    if (bf16) HIP_TRY(launch_round_bf16_inplace(vdst, (size_t)count * d.ld, s)); // the index IS the rounded vectors
    uint32_t *d_rowstatus = nullptr;
    HIP_TRY(hipMalloc((void **)&d_rowstatus, (size_t)count * 4));
    {
        DevIndex view = d;
        view.vec = vdst;
        std::vector<uint32_t> st(count);
        hipError_t e = launch_validate_rows(view, count, ix->limit, d_rowstatus, const_cast<float *>(d.hdr) + n0, s);
        if (e == hipSuccess) e = hipMemcpyAsync(st.data(), d_rowstatus, (size_t)count * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipFree(d_rowstatus);
        if (e != hipSuccess) return fail(HVX_ERR_DEVICE, "row validation: %s", hipGetErrorString(e));
        for (uint32_t i = 0; i < count; ++i)
            if (st[i]) return fail((int)st[i], "vector of node %llu is invalid for this metric (status %u)", (unsigned long long)node_ids[i], st[i]);
    }
    // ---- levels, upper-row bases, ids ----
    std::vector<uint16_t> h_lv(count, 0);
    std::vector<uint32_t> h_base(count, kSentinel);
    uint64_t r = ix->up_rows_used;
    for (uint32_t i = 0; i < count; ++i) {
        h_lv[i] = levels ? levels[i] : 0;
        if (h_lv[i]) { h_base[i] = (uint32_t)r; r += h_lv[i]; }
    }
    HIP_TRY(hipMemcpyAsync(const_cast<uint16_t *>(d.level) + n0, h_lv.data(), (size_t)count * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(const_cast<uint32_t *>(d.up_base) + n0, h_base.data(), (size_t)count * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(const_cast<uint64_t *>(d.ids) + n0, node_ids, (size_t)count * 8, hipMemcpyHostToDevice, s));
    if (bf16) HIP_TRY(launch_pack_bf16(vdst, const_cast<uint16_t *>(d.vecb) + (size_t)n0 * d.dim, count, d.dim, s));
    if (ix->has_simhash) { // SimHash rows of the new nodes (SimHasher::hash at insert time: mutation.rs:700-705)
        HIP_TRY(launch_simhash_rows(ix->d_planes_t, vdst, d.dim, d.ld, count, ix->d_node_hash + n0, s));
    }
    HIP_TRY(hipStreamSynchronize(s)); // (the host vectors above leave scope)
    auto grown = std::make_shared<std::vector<uint64_t>>();
    grown->reserve(n0 + count);
    grown->insert(grown->end(), ix->ids_ref().begin(), ix->ids_ref().end());
    grown->insert(grown->end(), node_ids, node_ids + count);
    bool contiguous = ix->contiguous && (n0 == 0 || node_ids[0] == ix->ids_ref()[n0 - 1] + 1);
    for (uint32_t i = 1; i < count && contiguous; ++i) contiguous = node_ids[i] == node_ids[i - 1] + 1;
    ix->ids_p = grown;
    ix->contiguous = contiguous;
    d.n = (uint32_t)(n0 + count);
    ix->desc.n = d.n;
    ix->up_rows_used = r;
    // ---- link them into the graph ----
    int rc = insert_range(ix, n0, count, h_lv.data(), n0, params, 1u, 0, stats, bf16 ? vdst : nullptr);
    if (rc) return rc; // (the image is partially linked: the host discards the handle and re-hydrates)
    ix->desc.has_entry = 1;
    ix->desc.entry_point = ix->ids_ref()[d.entry];
    ix->desc.max_layer = d.max_layer;
    ix->desc.shard_id_hi = ix->ids_ref()[d.n - 1];
    ix->publish_view();
    return HVX_OK;
}

namespace hvx {
__global__ void clear_dead_bit_kernel(uint32_t *dead, uint32_t row) { atomicAnd(&dead[row >> 5], ~(1u << (row & 31u))); }
}

// VectorInsertContract::Upsert (mutation.rs:642-780: an insert of an id the index already holds deletes it first, index.rs:2018-2060):
// for every id, in order -- a LIVE id is deleted (hvx_index_delete_batch), then the vector is linked back in under the same id: into the
// node's own row slot when the image holds one (it keeps its position in the ascending id order, so every id tie-break stays the
// reference's), appended when the id is above every id of the image.  The node keeps the level of its slot (the reference draws the
// level of an insert at random: any draw is a valid one); levels[i] is used for appended ids only.
extern "C" int hvx_index_upsert_batch(hvx_index *ix, const uint64_t *node_ids, const float *vectors, const uint16_t *levels, uint32_t count,
                                      const hvx_build_params *params, hvx_build_stats *stats) {
    if (!ix || (count && (!node_ids || !vectors))) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_build_params seq;
    hvx_build_params_default(&seq);
    if (params) seq = *params;
    seq.sequential = 1; // one node at a time: the reference's order
    if (stats) memset(stats, 0, sizeof(*stats));
    if (count == 0) return HVX_OK;
    if (ix->is_fork) return fail(HVX_ERR_UNSUPPORTED, "rows are written through the handle that owns the image, not a fork");
    const uint32_t dim = ix->dev.dim, ld = ix->dev.ld;
    if (ix->dev.dtype != HVX_F32 && ix->dev.dtype != HVX_BF16) return fail(HVX_ERR_UNSUPPORTED, "rows are written into f32 and bf16 images");
    const bool bf16 = ix->dev.dtype == HVX_BF16; // (round 6: an id the image holds gets its new vector in its slot; bf16 images have no spare rows to append to)
    // ---- every vector is validated before anything changes (an invalid one fails the call: mutation.rs:660-690) ----
    float *d_tmp = nullptr, *d_tmph = nullptr;
    uint32_t *d_st = nullptr;
    {
        std::lock_guard<std::mutex> lock(ix->mu);
        HIP_TRY(hipSetDevice(ix->device));
        hipStream_t s = ix->stream;
        if (hipMalloc((void **)&d_tmp, (size_t)count * ld * 4) != hipSuccess || hipMalloc((void **)&d_tmph, (size_t)count * 4) != hipSuccess ||
            hipMalloc((void **)&d_st, (size_t)count * 4) != hipSuccess) {
            (void)hipFree(d_tmp); (void)hipFree(d_tmph); (void)hipFree(d_st);
            return fail(HVX_ERR_DEVICE, "hipMalloc of the upsert staging rows failed");
        }
        std::vector<uint32_t> st(count);
        hipError_t e = hipMemsetAsync(d_tmp, 0, (size_t)count * ld * 4, s);
        if (e == hipSuccess) e = hipMemcpy2DAsync(d_tmp, (size_t)ld * 4, vectors, (size_t)dim * 4, (size_t)dim * 4, count, hipMemcpyDefault, s);
        if (e == hipSuccess && bf16) e = launch_round_bf16_inplace(d_tmp, (size_t)count * ld, s); // the index IS the rounded vectors: validation and headers see them
        DevIndex view = ix->dev;
        view.vec = d_tmp;
        if (e == hipSuccess) e = launch_validate_rows(view, count, ix->limit, d_st, d_tmph, s);
        if (e == hipSuccess) e = hipMemcpyAsync(st.data(), d_st, (size_t)count * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        (void)hipFree(d_st); (void)hipFree(d_tmph);
        if (e != hipSuccess) { (void)hipFree(d_tmp); return fail(HVX_ERR_DEVICE, "row validation: %s", hipGetErrorString(e)); }
        for (uint32_t i = 0; i < count; ++i)
            if (st[i]) { (void)hipFree(d_tmp); return fail((int)st[i], "vector of node %llu is invalid for this metric (status %u)", (unsigned long long)node_ids[i], st[i]); }
        // ids that are neither in the image nor above it cannot be placed (rows ascend with ids)
        const std::vector<uint64_t> &ids = ix->ids_ref();
        uint64_t last = ids.empty() ? 0 : ids.back();
        bool any = !ids.empty();
        for (uint32_t i = 0; i < count; ++i) {
            if (ix->find_slot(node_ids[i]) != kSentinel) continue;
            if (any && node_ids[i] <= last) { (void)hipFree(d_tmp); return fail(HVX_ERR_UNSUPPORTED, "node %llu lies between the ids of the image: it has no row slot (hydrate the image again)", (unsigned long long)node_ids[i]); }
            last = node_ids[i];
            any = true;
        }
    }
    auto done = [&](int rc) { (void)hipSetDevice(ix->device); (void)hipFree(d_tmp); return rc; };
    uint64_t nodes = 0, batches = 0, singles = 0;
    for (uint32_t i = 0; i < count; ++i) {
        int rc;
        if (ix->find(node_ids[i]) != kSentinel && (rc = hvx_index_delete_batch(ix, node_ids + i, 1, nullptr))) return done(rc); // live: the delete half
        hvx_build_stats one{};
        if (ix->find_slot(node_ids[i]) == kSentinel) { // above the image: an ordinary append
            const uint16_t lv = levels ? levels[i] : 0;
            if ((rc = hvx_index_insert_batch(ix, node_ids + i, d_tmp + (size_t)i * ld, &lv, 1, &seq, &one))) return done(rc);
        } else { // the node's own (deleted) slot
            std::lock_guard<std::mutex> lock(ix->mu);
            if (hipSetDevice(ix->device) != hipSuccess) return done(fail(HVX_ERR_DEVICE, "hipSetDevice failed"));
            DevIndex &d = ix->dev;
            hipStream_t s = ix->stream;
            const uint32_t row = ix->find_slot(node_ids[i]);
            float *vsrc = d_tmp + (size_t)i * ld; // the validated (bf16 images: rounded) vector
            float *vdst = bf16 ? vsrc : const_cast<float *>(d.vec) + (size_t)row * d.ld;
            uint16_t lv = 0;
            hipError_t e = bf16 ? launch_pack_bf16(vsrc, const_cast<uint16_t *>(d.vecb) + (size_t)row * d.dim, 1u, d.dim, s)
                                : hipMemcpyAsync(vdst, vsrc, (size_t)ld * 4, hipMemcpyDeviceToDevice, s);
            uint32_t *d_one = nullptr;
            if (e == hipSuccess) e = hipMalloc((void **)&d_one, 4);
            DevIndex view = d;
            view.vec = vdst;
            if (e == hipSuccess) e = launch_validate_rows(view, 1, ix->limit, d_one, const_cast<float *>(d.hdr) + row, s); // (the header of the new vector)
            if (e == hipSuccess && ix->has_simhash) e = launch_simhash_rows(ix->d_planes_t, vdst, d.dim, d.ld, 1, ix->d_node_hash + row, s);
            if (e == hipSuccess) e = hipMemcpyAsync(&lv, d.level + row, 2, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess && ix->d_dead) { hipLaunchKernelGGL(clear_dead_bit_kernel, dim3(1), dim3(1), 0, s, ix->d_dead, row); e = hipGetLastError(); }
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            (void)hipFree(d_one);
            if (e != hipSuccess) return done(fail(HVX_ERR_DEVICE, "upsert of node %llu: %s", (unsigned long long)node_ids[i], hipGetErrorString(e)));
            { // caches derived from the rows' VECTORS (per-handle |x|^2, the image's bf16 shadow) are rebuilt by their next user
                std::lock_guard<std::mutex> g(ix->shared->mu);
                ix->shared->vec_epoch += 1;
                ix->shared->shadow_rows = 0;
            }
            auto flags = std::make_shared<std::vector<uint8_t>>(*ix->dead_p); // this generation's flags: the slot is live again
            (*flags)[row] = 0;
            ix->dead_p = flags;
            ix->n_dead -= 1;
            if ((rc = insert_range(ix, row, 1, &lv, row, &seq, 1u, d.n, &one, bf16 ? vsrc : nullptr))) return done(rc);
            ix->desc.has_entry = 1;
            ix->desc.entry_point = ix->ids_ref()[d.entry];
            ix->desc.max_layer = d.max_layer;
            ix->publish_view();
            ix->seen_rewrite = ix->shared->rewrite_epoch.fetch_add(1, std::memory_order_acq_rel) + 1;
        }
        nodes += 1; batches += one.batches; singles += one.single_node_batches;
    }
    if (stats) { stats->nodes = nodes; stats->batches = batches; stats->single_node_batches = singles; }
    return done(HVX_OK);
}

// add_bidirectional_link on layer 0 of an existing image, through build_link_wg_kernel (include/helix_vec.h): the kernel that links
// every batched build, driven link by link so that tests can hold each prune against the oracle's select_diverse + backfill.
extern "C" int hvx_index_link_rows(hvx_index *ix, const uint64_t *from_ids, const uint64_t *to_ids, uint32_t n_links, uint32_t concurrent) {
    if (!ix || (n_links && (!from_ids || !to_ids))) return fail(HVX_ERR_INVARIANT, "null argument");
    if (n_links == 0) return HVX_OK;
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    DevIndex &d = ix->dev;
    if (ix->is_fork) return fail(HVX_ERR_UNSUPPORTED, "rows are linked through the handle that owns the image, not a fork");
    const uint32_t m = ix->desc.m ? ix->desc.m : 16u;
    const uint32_t m0 = std::max(ix->desc.m0 ? ix->desc.m0 : 2u * m, 2u * m);
    const bool fused = kernel_fused(d.fkernel);
    const BuildKernels kern = pick_build_kernels(d.metric, fused);
    const uint32_t ncmax = std::max(m0, m) + 1u;
    const uint32_t nk_rows = d.dim_main >> 5;
    const uint32_t link_ck = std::min<uint32_t>(8u, (nk_rows + 1u) & ~1u);
    const uint32_t ldp = link_ck * 32u + 32u;
    if (d.dtype != HVX_F32 || !kern.link_wg || kernel_w4(d.fkernel) || ncmax > 33u || nk_rows == 0 || d.dim_main != d.dim || d.ld != d.dim || d.s0 < m0 ||
        (size_t)(ncmax + 1u) * link_ck * 8u > 9u * 256u)
        return fail(HVX_ERR_UNSUPPORTED, "the link workgroups serve f32 rows, L2 / cosine, dim %% 32 == 0, m0 <= 32 (row stride >= m0)");
    std::vector<uint32_t> h_nodes(n_links), h_sel((size_t)n_links * 32u, kSentinel), h_cnt(n_links, 1u);
    for (uint32_t i = 0; i < n_links; ++i) {
        const uint32_t f = ix->find(from_ids[i]), t = ix->find(to_ids[i]);
        if (f == kSentinel || t == kSentinel || f == t) return fail(HVX_ERR_INVARIANT, "link %u: unknown id or a self link", i);
        h_nodes[i] = f;
        h_sel[(size_t)i * 32u] = t;
    }
    hipStream_t s = ix->stream;
    uint32_t *d_nodes = nullptr, *d_sel = nullptr, *d_cnt = nullptr, *d_locks = nullptr, *d_err = nullptr;
    auto release = [&]() {
        for (void *p : {(void *)d_nodes, (void *)d_sel, (void *)d_cnt, (void *)d_locks, (void *)d_err})
            if (p) (void)hipFree(p);
    };
    auto bail = [&](int rc) { (void)hipStreamSynchronize(s); release(); return rc; };
    if (hipMalloc((void **)&d_nodes, (size_t)n_links * 4) != hipSuccess || hipMalloc((void **)&d_sel, (size_t)n_links * 32 * 4) != hipSuccess ||
        hipMalloc((void **)&d_cnt, (size_t)n_links * 4) != hipSuccess || hipMalloc((void **)&d_locks, (size_t)d.n * 4) != hipSuccess ||
        hipMalloc((void **)&d_err, 4) != hipSuccess)
        return bail(fail(HVX_ERR_DEVICE, "hipMalloc of the link scratch failed"));
    if (hipMemcpyAsync(d_nodes, h_nodes.data(), (size_t)n_links * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_sel, h_sel.data(), h_sel.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_cnt, h_cnt.data(), (size_t)n_links * 4, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemsetAsync(d_locks, 0, (size_t)d.n * 4, s) != hipSuccess || hipMemsetAsync(d_err, 0, 4, s) != hipSuccess)
        return bail(fail(HVX_ERR_DEVICE, "upload of the link list failed"));
    BuildArgs ba{};
    ba.ix = d;
    ba.l0 = const_cast<uint32_t *>(d.l0);
    ba.up = const_cast<uint32_t *>(d.up);
    ba.locks = d_locks;
    ba.layers = 1; // layer 0 only
    ba.sel = d_sel;
    ba.sel_cnt = d_cnt;
    ba.m = m;
    ba.m0 = m0;
    ba.err = d_err;
    ba.ldp = ldp;
    ba.ncmax = ncmax;
    ba.link_ck = link_ck;
    const size_t lds = link_lds_bytes(ldp, ncmax);
    hipError_t e = hipSuccess;
    if (concurrent) {
        ba.nodes = d_nodes;
        ba.b = n_links;
        e = launch_link_wg(kern.link_wg, ba, 1, lds, s);
    } else {
        for (uint32_t i = 0; i < n_links && e == hipSuccess; ++i) { // one launch per link: launches on a stream run in order
            ba.nodes = d_nodes + i;
            ba.sel = d_sel + (size_t)i * 32u;
            ba.sel_cnt = d_cnt + i;
            ba.b = 1;
            e = launch_link_wg(kern.link_wg, ba, 1, lds, s);
        }
    }
    if (e != hipSuccess) return bail(fail(HVX_ERR_DEVICE, "link launch failed: %s", hipGetErrorString(e)));
    uint32_t err = 0;
    if (hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return bail(fail(HVX_ERR_DEVICE, "link kernels failed: %s", hipGetErrorString(hipGetLastError())));
    release();
    if (err) return fail(HVX_ERR_INVARIANT, "a neighbour row overflowed its stride during the link step");
    return HVX_OK;
}

// ---- graph read-back: what the host persists (values/vectors.rs rows) and what tests compare with the oracle ----
extern "C" int hvx_index_graph_sizes(const hvx_index *cix, uint64_t *l0_edges, uint64_t *up_rows, uint64_t *up_edges,
                                     uint64_t *entry_point, uint32_t *max_layer, uint32_t *has_entry) {
    if (!cix) return fail(HVX_ERR_INVARIANT, "null index");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    const DevIndex &d = ix->dev;
    std::vector<uint32_t> h((size_t)d.n * d.s0);
    if (d.n) HIP_TRY(hipMemcpy(h.data(), d.l0, h.size() * 4, hipMemcpyDeviceToHost));
    uint64_t e0 = 0;
    for (uint32_t v : h) e0 += v != kSentinel;
    std::vector<uint16_t> lv(d.n);
    if (d.n) HIP_TRY(hipMemcpy(lv.data(), d.level, (size_t)d.n * 2, hipMemcpyDeviceToHost));
    uint64_t ur = 0;
    for (uint16_t v : lv) ur += v;
    std::vector<uint32_t> hu((size_t)ur * d.su);
    if (ur) HIP_TRY(hipMemcpy(hu.data(), d.up, hu.size() * 4, hipMemcpyDeviceToHost));
    uint64_t eu = 0;
    for (uint32_t v : hu) eu += v != kSentinel;
    if (l0_edges) *l0_edges = e0;
    if (up_rows) *up_rows = ur;
    if (up_edges) *up_edges = eu;
    if (entry_point) *entry_point = d.has_entry ? ix->ids_ref()[d.entry] : 0;
    if (max_layer) *max_layer = d.max_layer;
    if (has_entry) *has_entry = d.has_entry;
    return HVX_OK;
}

extern "C" int hvx_index_export_graph(const hvx_index *cix, uint64_t *l0_offsets, uint64_t *l0_neighbors, uint16_t *level,
                                      uint64_t *up_offsets, uint64_t *up_neighbors) {
    if (!cix || !l0_offsets || !l0_neighbors) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    const DevIndex &d = ix->dev;
    const std::vector<uint64_t> &ids = ix->ids_ref();
    std::vector<uint32_t> h((size_t)d.n * d.s0);
    if (d.n) HIP_TRY(hipMemcpy(h.data(), d.l0, h.size() * 4, hipMemcpyDeviceToHost));
    uint64_t w = 0;
    for (uint32_t i = 0; i < d.n; ++i) {
        l0_offsets[i] = w;
        for (uint32_t t = 0; t < d.s0; ++t) {
            const uint32_t v = h[(size_t)i * d.s0 + t];
            if (v != kSentinel) l0_neighbors[w++] = ids[v];
        }
    }
    l0_offsets[d.n] = w;
    std::vector<uint16_t> lv(d.n);
    if (d.n) HIP_TRY(hipMemcpy(lv.data(), d.level, (size_t)d.n * 2, hipMemcpyDeviceToHost));
    if (level) memcpy(level, lv.data(), (size_t)d.n * 2);
    uint64_t ur = 0;
    for (uint16_t v : lv) ur += v;
    if (up_offsets && up_neighbors) {
        std::vector<uint32_t> hu((size_t)ur * d.su);
        if (ur) HIP_TRY(hipMemcpy(hu.data(), d.up, hu.size() * 4, hipMemcpyDeviceToHost));
        uint64_t wu = 0;
        for (uint64_t r = 0; r < ur; ++r) {
            up_offsets[r] = wu;
            for (uint32_t t = 0; t < d.su; ++t) {
                const uint32_t v = hu[(size_t)r * d.su + t];
                if (v != kSentinel) up_neighbors[wu++] = ids[v];
            }
        }
        up_offsets[ur] = wu;
    }
    return HVX_OK;
}
