// hvx_delete.hip -- delete of nodes from a LIVE device image (VERDICT r4 missing #4, second half): the reference's
// stage_delete_with_metadata (crates/db/src/search/vector/mutation.rs:1658-1774) -> delete_from_layer (:1819-1888) ->
// remove_edge_from_neighbor (:1890-1908) + relink_neighbor (:1916-2055), one node after the other, on the rows resident in HBM.
//
// What the reference does for ONE node x, per layer (highest first; the layers are independent of each other -- each touches
// only its own neighbour rows -- so the device runs them side by side, one workgroup per layer):
//   sources      every row of the layer that holds x: x's own out-neighbours (they relink whether or not they hold the edge) and the
//                reverse-locator sources (update_reverse_edge_locator :1134-1153 keeps that set equal to "rows that contain x")
//   unlink       x is removed from each of them
//   candidates   C = the sources and everything their remaining rows hold (x excluded)
//   relink(s)    for every source s in ascending id order: the Mmax closest members of C \ {s} that s does not hold yet are appended
//                to its row; a row beyond Mmax is ranked by distance to s and pruned by select_diverse + backfill (mod.rs:809-856);
//                every NEW neighbour t gets s appended to ITS row, pruned the same way when that overflows
// then x's rows, item, SimHash row and entry-candidate rows are deleted and, when x was the entry point, the entry moves to the best
// remaining entry candidate: highest layer first, then the smallest id (keys/vectors.rs:1097 [inv_layer:2][node_id:8]).
//
// Kernels per deleted node (one host read-back per node: the number of relink steps, behind the prep kernel, while the ranking runs):
//   delete_scan_kernel        grid over every row slot: which rows hold x, per layer         (128-260 MB of rows at 1M nodes: ~50 us)
//   delete_prep_kernel        one workgroup per layer: dedupe + sort the sources, unlink x, collect C (a bitmap marks membership)
//   delete_rank_kernel        workgroups x layers: for every source its distances to all of C in the reference's summation order and
//                             the Mmax smallest in Candidate order (model.rs:55-61) -- independent of the graph, so this part runs in
//                             parallel over the sources (sixteen wavefronts per source, register top lists)
//   delete_step_fused_kernel  ONE launch per relinked row (round 6, below): all prunes' distance matrices evaluated up front over the
//                             layer's workgroups, the last workgroup replays select_diverse from registers; NARROW (rows <= 34 ids) and
//                             WIDE (rows <= 64 ids: M 32 / M0 64 graphs) builds, f32 and bf16 rows.  The relinks happen IN ORDER (a relink
//                             reads rows earlier relinks of the same delete changed): one step after the other on the stream
//   delete_step_own / _recip  the same step as two launches (HVX_OPT_DELETE_SEQUENTIAL = 2)
//   delete_relink_kernel      one wavefront per layer: empties x's own rows and sets its deleted bit; with HVX_OPT_DELETE_SEQUENTIAL = 1
//                             it also runs every relink itself (lazy select_diverse: the round-5 first build, 26.9 ms per delete)
// A deleted node keeps its row slot: unreachable (nothing links to it, the entry point is repaired), absent from every id -> row
// lookup (hvx_index::find), from exact scans (they run over the live rows, hvx_index::ensure_live), from the SimHash directory, the
// prefilter's candidate mapping and the audit.  Sequential semantics: the nodes of a batch are deleted one after the other, and
// the rows equal the CPU restatement's sequential deletes row for row -- tests/test_gpu_delete.py.
// Limits (exceeded => HVX_ERR_UNSUPPORTED, loudly): 4 096 rows holding one node per layer, 16 384 candidates per layer, rows of more than
// 64 ids, fp8 images.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "hvx_host.h"
#include "hvx_graph_dev.h"
#include "hvx_toplist.h"

using namespace hvx;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(HVX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

namespace hvx {

constexpr uint32_t kDelSrcCap = 4096;   // rows that hold the node, per layer
constexpr uint32_t kDelRelCap = 4096;   // relink sources per layer (sorted in LDS)
constexpr uint32_t kDelCandCap = 16384; // candidates per layer (one source's scores live in LDS)
constexpr uint32_t kDelTop = 64;        // Mmax (round 6: degree limits up to 64 through the wide build of the fused step)
constexpr uint32_t kDelMinLayers = 16;
constexpr uint32_t kDelWaves = 4;  // wavefronts of a step workgroup (32 row groups of 8 lanes)
constexpr uint32_t kDelDm = 65 * 64; // one prune's distance matrix

struct DeleteArgs {
    DevIndex ix;
    uint32_t *l0, *up;      // ix.l0 / ix.up, writable
    uint32_t node;          // row being deleted
    uint32_t layers;        // max_layer + 1
    uint32_t m, m0, words;  // degree limits; words of one layer's mark bitmap
    uint32_t *src, *src_cnt;   // [layers][kDelSrcCap], [layers]: rows that hold the node (unordered)
    uint32_t *rel, *rel_cnt;   // [layers][kDelRelCap], [layers]: relink sources, ascending
    uint32_t *cand, *cand_cnt; // [layers][kDelCandCap], [layers]
    uint32_t *top, *top_cnt;   // [layers][kDelRelCap][kDelTop], [layers][kDelRelCap]: a source's closest candidates, Candidate order
    uint32_t *mark;            // [layers][words] all zero between kernels
    uint32_t *newl, *new_cnt;  // [layers][kDelTop], [layers]: the neighbours the current step's source has gained (step kernels)
    float *gdm;                // [layers][1 + kDelTop][65 x 64]: the distance matrices of a step's prunes (slot 0: the source's own row,
    uint32_t *tick;            //  1 + t: its t-th new neighbour's), and per matrix the workgroups that have delivered (zero between launches)
    uint32_t g_own, g_recip;   // workgroups that share one matrix
    uint32_t first;            // delete_relink_kernel starts with this source (the ones before it ran as steps)
    uint32_t *dead;            // the image's deleted-row bitmap
    uint32_t *ctl;             // [0] error (1 sources, 2 relink sources, 3 candidates, 4 invalid score, 5 row overflow) [1] relinked rows
                               // [2..3] best entry key (u64)
};

__device__ __forceinline__ uint32_t *del_row(const DeleteArgs &a, uint32_t node, uint32_t layer, uint32_t &stride) {
    if (layer == 0u) { stride = a.ix.s0; return a.l0 + (size_t)node * a.ix.s0; }
    stride = a.ix.su;
    return a.up + (size_t)(a.ix.up_base[node] + layer - 1u) * a.ix.su;
}

// reverse_sources_for_target (storage.rs, mutation.rs:1677): every row that holds the node, per layer
__global__ __launch_bounds__(256) void delete_scan_kernel(DeleteArgs a) {
    const DevIndex &ix = a.ix;
    const unsigned long long total = (unsigned long long)ix.n * ix.s0, step = (unsigned long long)gridDim.x * 256ull;
    for (unsigned long long t = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; t < total; t += step) {
        if (a.l0[t] != a.node) continue;
        const uint32_t owner = (uint32_t)(t / ix.s0);
        if (owner == a.node) continue;
        const uint32_t pos = atomicAdd(&a.src_cnt[0], 1u);
        if (pos < kDelSrcCap) a.src[pos] = owner;
    }
    for (unsigned long long u = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; u < ix.n; u += step) {
        uint32_t lv = ix.level[u];
        if (lv == 0u || (uint32_t)u == a.node) continue;
        if (lv >= a.layers) lv = a.layers - 1u;
        const uint32_t base = ix.up_base[u];
        for (uint32_t l = 1; l <= lv; ++l) {
            const uint32_t *row = a.up + (size_t)(base + l - 1u) * ix.su;
            for (uint32_t p = 0; p < ix.su; ++p) {
                const uint32_t v = row[p];
                if (v == kSentinel) break; // canonical rows: the valid ids come first
                if (v == a.node) {
                    const uint32_t pos = atomicAdd(&a.src_cnt[l], 1u);
                    if (pos < kDelSrcCap) a.src[(size_t)l * kDelSrcCap + pos] = (uint32_t)u;
                    break;
                }
            }
        }
    }
}

// delete_from_layer up to the relinks (mutation.rs:1829-1875): sources, unlink, candidates.  One workgroup per layer.
__global__ __launch_bounds__(256) void delete_prep_kernel(DeleteArgs a) {
    __shared__ uint32_t aff[kDelRelCap], srt[kDelRelCap];
    __shared__ uint32_t s_na, s_nc, s_err;
    const DevIndex &ix = a.ix;
    const uint32_t L = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t node = a.node, lvl = ix.level[node];
    const uint32_t ns = a.src_cnt[L];
    if (tid == 0) { a.rel_cnt[L] = 0; a.cand_cnt[L] = 0; s_na = 0; s_nc = 0; s_err = a.ctl[0]; }
    __syncthreads();
    if (s_err != 0u) return; // (one read for the whole workgroup: another layer's workgroup may be raising the flag right now)
    if (ns > kDelSrcCap) { if (tid == 0) atomicMax(&a.ctl[0], 1u); return; }
    if (L > lvl && ns == 0u) return;
    uint32_t *mark = a.mark + (size_t)L * a.words;
    auto claim = [&](uint32_t v) -> bool { // true for the first claim of a row
        const uint32_t bit = 1u << (v & 31u);
        return (atomicOr(&mark[v >> 5], bit) & bit) == 0u;
    };
    if (tid == 0) (void)claim(node);
    __syncthreads();
    // the node's out-neighbours (mandatory_relink :1833-1837), then the rows that hold it (:1838-1844); BTreeSet = ascending ids
    if (L <= lvl) {
        uint32_t stride;
        const uint32_t *row = del_row(a, node, L, stride);
        const uint32_t v = tid < stride ? ld_row(row + tid) : kSentinel;
        if (v != kSentinel && v < ix.n && ix.level[v] >= L && claim(v)) { const uint32_t pos = atomicAdd(&s_na, 1u); if (pos < kDelRelCap) aff[pos] = v; }
    }
    __syncthreads();
    for (uint32_t t = tid; t < ns; t += 256u) {
        const uint32_t v = a.src[(size_t)L * kDelSrcCap + t];
        if (claim(v)) { const uint32_t pos = atomicAdd(&s_na, 1u); if (pos < kDelRelCap) aff[pos] = v; }
    }
    __syncthreads();
    const uint32_t na_all = s_na, na = na_all < kDelRelCap ? na_all : kDelRelCap;
    for (uint32_t i = tid; i < na; i += 256u) { // rows = ids ascending: rank by counting (the ids are distinct)
        const uint32_t v = aff[i];
        uint32_t r = 0;
        for (uint32_t t = 0; t < na; ++t) r += aff[t] < v ? 1u : 0u;
        srt[r] = v;
    }
    __syncthreads();
    uint32_t lstride = L == 0u ? ix.s0 : ix.su;
    if (na_all <= kDelRelCap) {
        // remove_edge_from_neighbor (:1890-1908) on every source: one wavefront per row (the rows are distinct)
        for (uint32_t i = wave; i < na; i += 4u) {
            uint32_t stride;
            uint32_t *row = del_row(a, srt[i], L, stride);
            const uint32_t v = lane < stride ? ld_row(row + lane) : kSentinel;
            if (__ballot(v == node) == 0ull) continue;
            const bool keep = v != kSentinel && v != node;
            const unsigned long long km = __ballot(keep);
            const uint32_t pos = (uint32_t)__builtin_popcountll(km & ((1ull << lane) - 1ull)), nk = (uint32_t)__builtin_popcountll(km);
            if (keep) st_row(row + pos, v);
            if (lane >= nk && lane < stride) st_row(row + lane, kSentinel);
        }
        __threadfence();
        __syncthreads();
        // candidates (:1862-1875): the relink sources (their bits are set already), then what their rows hold now
        uint32_t *cand = a.cand + (size_t)L * kDelCandCap;
        for (uint32_t i = tid; i < na; i += 256u) cand[i] = srt[i];
        if (tid == 0) s_nc = na;
        __syncthreads();
        for (uint32_t t = tid; t < na * lstride; t += 256u) {
            uint32_t stride;
            const uint32_t *row = del_row(a, srt[t / lstride], L, stride);
            const uint32_t v = ld_row(row + (t % lstride));
            if (v != kSentinel && v < ix.n && ix.level[v] >= L && claim(v)) { const uint32_t pos = atomicAdd(&s_nc, 1u); if (pos < kDelCandCap) cand[pos] = v; }
        }
        __syncthreads();
        const uint32_t nc_all = s_nc, nc = nc_all < kDelCandCap ? nc_all : kDelCandCap;
        for (uint32_t i = tid; i < nc; i += 256u) { const uint32_t v = cand[i]; atomicAnd(&mark[v >> 5], ~(1u << (v & 31u))); }
        for (uint32_t i = tid; i < na; i += 256u) a.rel[(size_t)L * kDelRelCap + i] = srt[i];
        if (tid == 0) {
            atomicAnd(&mark[node >> 5], ~(1u << (node & 31u)));
            if (nc_all > kDelCandCap) atomicMax(&a.ctl[0], 3u);
            else { a.rel_cnt[L] = na; a.cand_cnt[L] = nc; }
        }
    } else if (tid == 0) {
        atomicMax(&a.ctl[0], 2u);
    }
}

// relink_neighbor's ranking (mutation.rs:1936-1957) for every source of every layer: distances to all candidates, the Mmax smallest
template <uint32_t METRIC, bool FUSED, bool BF> __global__ __launch_bounds__(1024) void delete_rank_kernel(DeleteArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t s_bad;
    const DevIndex &ix = a.ix;
    const uint32_t L = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int grp = (int)(lane >> 3), j = (int)(lane & 7u);
    if (tid == 0) { s_bad = a.ctl[0]; }
    __syncthreads();
    if (s_bad != 0u) return;
    const uint32_t nr = a.rel_cnt[L], nc = a.cand_cnt[L];
    if (nr == 0u || nc == 0u) return;
    const uint32_t maxn = L == 0u ? a.m0 : a.m;
    float *qv = reinterpret_cast<float *>(smem);
    // Round 6: sixteen wavefronts; every wavefront keeps the Mmax smallest (score, id) pairs of ITS candidates in a register list
    // (hvx_toplist.h), the lists meet in LDS and wavefront 0 merges them -- rounds 3-5 kept every score in LDS and ran Mmax block-wide
    // arg-min reductions per source (59 us per delete)
    float *m_sc = reinterpret_cast<float *>(smem + (((size_t)ix.ld * 4u + 15u) & ~(size_t)15u));
    uint32_t *m_id = reinterpret_cast<uint32_t *>(m_sc + 16 * 64);
    const uint32_t *cand = a.cand + (size_t)L * kDelCandCap;
    for (uint32_t ri = blockIdx.x; ri < nr; ri += gridDim.x) {
        const uint32_t r = a.rel[(size_t)L * kDelRelCap + ri];
        __syncthreads();
        if constexpr (BF) { // the source's row as f32, element order
            const uint16_t *rb = ix.vecb + (size_t)r * ix.dim;
            for (uint32_t t = tid; t < ix.dim; t += 1024u) qv[t] = bf16_to_f32(rb[bf16_slot_of(t)]);
        } else {
            const float *rv = ix.vec + (size_t)r * ix.ld;
            for (uint32_t t = tid; t < ix.ld; t += 1024u) qv[t] = rv[t];
        }
        __syncthreads();
        const float rh = ix.hdr[r];
        bool bad = false;
        TopList l;
        l.init();
        for (uint32_t c0 = 0; c0 < nc; c0 += 128u) {
            const uint32_t idx = c0 + wave * 8u + (uint32_t)grp;
            const uint32_t c = cand[idx < nc ? idx : nc - 1u];
            float d;
            if constexpr (BF) d = group_distance_bf16<METRIC == kL1 ? kL2 : METRIC>(ix, qv, rh, c, j);
            else d = group_distance<METRIC, FUSED>(ix, qv, rh, c, j);
            bool take = idx < nc && j == 0 && c != r; // (the source itself is no candidate of its own relink: :1937-1939)
            if (take && !score_valid(d)) { bad = true; take = false; } // Candidate::try_new fails: the reference aborts the delete
            l.offer(take, d, c, maxn, (int)lane);
        }
        if (__ballot(bad) != 0ull && lane == 0) s_bad = 1u;
        m_sc[wave * 64u + lane] = l.sc;
        m_id[wave * 64u + lane] = l.id;
        __syncthreads();
        if (wave == 0) {
            TopList m;
            m.init();
            for (uint32_t w = 0; w < 16u; ++w) {
                const uint32_t id = m_id[w * 64u + lane];
                m.offer(id != kSentinel && lane < maxn, m_sc[w * 64u + lane], id, maxn, (int)lane);
            }
            const uint32_t got = m.count < maxn ? m.count : maxn;
            uint32_t *out = a.top + ((size_t)L * kDelRelCap + ri) * kDelTop;
            if (lane < got) out[lane] = m.id;
            if (lane == 0) a.top_cnt[(size_t)L * kDelRelCap + ri] = got;
        }
    }
    __syncthreads();
    if (tid == 0 && s_bad) atomicMax(&a.ctl[0], 4u);
}

// the relinks of one layer, in order (mutation.rs:1876-1887 -> relink_neighbor :1916-2055), then the node's own row goes
template <uint32_t METRIC, bool FUSED> __global__ __launch_bounds__(64) void delete_relink_kernel(DeleteArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevIndex &ix = a.ix;
    const uint32_t L = blockIdx.x;
    const int lane = (int)threadIdx.x;
    BuildLds Ld = carve_build(smem, ix.ld);
    uint32_t *oldl = reinterpret_cast<uint32_t *>(smem + ((((size_t)ix.ld * 4u + 15u) & ~(size_t)15u) + 4 * 256)); // [64] the row before the relink
    const uint32_t node = a.node, lvl = ix.level[node];
    const bool failed = a.ctl[0] != 0u;
    const uint32_t nr = failed ? 0u : a.rel_cnt[L];
    const uint32_t maxn = L == 0u ? a.m0 : a.m;
    const unsigned long long lt = (1ull << lane) - 1ull;
    bool bad = false, overflow = false;
    for (uint32_t ri = a.first; ri < nr; ++ri) {
        const uint32_t nb = a.rel[(size_t)L * kDelRelCap + ri];
        uint32_t stride;
        uint32_t *row = del_row(a, nb, L, stride);
        const uint32_t v = (uint32_t)lane < stride ? ld_row(row + lane) : kSentinel;
        const uint32_t nold = (uint32_t)__builtin_popcountll(__ballot(v != kSentinel));
        const uint32_t tn = a.top_cnt[(size_t)L * kDelRelCap + ri];
        const uint32_t tv = (uint32_t)lane < tn ? a.top[((size_t)L * kDelRelCap + ri) * kDelTop + lane] : kSentinel;
        __syncthreads();
        oldl[lane] = v; // canonical: the valid ids are lanes 0..nold-1
        __syncthreads();
        // :1953-1957 the closest candidates the row does not hold yet, appended in Candidate order
        bool in_old = false;
        for (uint32_t s = 0; s < nold; ++s) in_old |= oldl[s] == tv;
        const bool add = tv != kSentinel && !in_old;
        const unsigned long long am = __ballot(add);
        const uint32_t ncur = nold + (uint32_t)__builtin_popcountll(am);
        if (ncur > 64u) { overflow = true; continue; }
        if ((uint32_t)lane < nold) Ld.kept[lane] = v;
        if (add) Ld.kept[nold + (uint32_t)__builtin_popcountll(am & lt)] = tv;
        __syncthreads();
        const uint32_t cv = (uint32_t)lane < ncur ? Ld.kept[lane] : kSentinel;
        uint32_t keepn = ncur;
        if (ncur > maxn) keepn = prune_row_dev<METRIC, FUSED>(ix, Ld, nb, cv, ncur, maxn, lane, &bad); // :1959-1984
        __syncthreads();
        const uint32_t f = (uint32_t)lane < keepn ? Ld.kept[lane] : kSentinel;
        bool was_old = false;
        for (uint32_t s = 0; s < nold; ++s) was_old |= oldl[s] == f;
        unsigned long long nm = __ballot(f != kSentinel && !was_old);
        if (keepn > stride) { overflow = true; continue; }
        store_canonical(row, stride, Ld.kept, keepn, lane, true); // :1986-1993 (staged rows are sorted by id, :1299)
        __syncthreads();
        while (nm) { // :1994-2052 the reciprocal row of every new neighbour
            const uint32_t srcl = (uint32_t)__builtin_ctzll(nm);
            nm &= nm - 1ull;
            const uint32_t nw = (uint32_t)__builtin_amdgcn_readlane((int)f, (int)srcl);
            uint32_t rstride;
            uint32_t *rrow = del_row(a, nw, L, rstride);
            uint32_t rv = (uint32_t)lane < rstride ? ld_row(rrow + lane) : kSentinel;
            uint32_t rdeg = (uint32_t)__builtin_popcountll(__ballot(rv != kSentinel));
            if (__ballot(rv == nb) != 0ull) continue;
            if (rdeg >= 64u) { overflow = true; continue; } // (the id is appended in a register: the row itself never holds more than Mmax)
            if ((uint32_t)lane == rdeg) rv = nb;
            ++rdeg;
            uint32_t kn = rdeg;
            if (rdeg > maxn) {
                kn = prune_row_dev<METRIC, FUSED>(ix, Ld, nw, rv, rdeg, maxn, lane, &bad); // :2007-2040
            } else {
                __syncthreads();
                if ((uint32_t)lane < rdeg) Ld.kept[lane] = rv;
                __syncthreads();
            }
            store_canonical(rrow, rstride, Ld.kept, kn, lane, true);
            __syncthreads();
        }
        __threadfence(); // the next relink of this layer reads rows this one has written
        __syncthreads();
    }
    __threadfence();
    __syncthreads();
    if (!failed && L <= lvl) { // :1726-1730 the node's neighbour rows are deleted
        uint32_t stride;
        uint32_t *row = del_row(a, node, L, stride);
        if ((uint32_t)lane < stride) st_row(row + lane, kSentinel);
    }
    if (lane == 0) {
        a.src_cnt[L] = 0;
        if (nr) atomicAdd(&a.ctl[1], nr);
        if (__builtin_expect(bad, 0)) atomicMax(&a.ctl[0], 4u);
        if (__builtin_expect(overflow, 0)) atomicMax(&a.ctl[0], 5u);
        if (L == 0u && !failed) atomicOr(&a.dead[node >> 5], 1u << (node & 31u));
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same relinks as STEPS (the default): the one-wavefront kernel below evaluates every prune lazily -- candidate after candidate, each a
// dependent row gather -- ~0.5 ms per prune and ~15 prunes per source (26.9 ms per delete at 200 000 x 768, profiles/r05t_delete_bench.json).
// A step runs one source: delete_step_own_kernel (one 16-wavefront workgroup per layer) merges the source's row with its closest
// candidates and, beyond Mmax, evaluates the WHOLE distance matrix of the <= 64 ids at once (every wavefront stages one id's vector and
// scores it against all the others, reference summation order) before one wavefront replays select_diverse + backfill from LDS;
// delete_step_recip_kernel (one workgroup per new neighbour) does the same for the reciprocal rows, which are independent of each
// other.  Same distances, same comparisons, same order of decisions as the lazy evaluation: the rows are identical (both modes are
// held to the oracle by tests/test_gpu_delete.py; HVX_OPT_DELETE_SEQUENTIAL selects the one-wavefront kernel).
// (A pair distance that is not a valid score aborts the reference only if select_diverse reaches that pair; the eager matrix treats
// it as "not closer" -- rows validated at import cannot produce one under the component limit.)
// ---------------------------------------------------------------------------------------------------------------------------------
struct StepLds {
    float *Dm;                                // [65][64]: Dm[i][j] = distance(staged id i, stored id j); row nc = the owner
    uint32_t *cur, *ord, *sel, *kept, *oldl;  // [64] each
    float *osc;                               // [64]
    uint32_t *misc;                           // [8]
};
__device__ __forceinline__ StepLds carve_step(char *smem) {
    StepLds S;
    char *p = smem;
    S.Dm = reinterpret_cast<float *>(p); p += 65 * 64 * 4;
    S.cur = reinterpret_cast<uint32_t *>(p); p += 256;
    S.ord = reinterpret_cast<uint32_t *>(p); p += 256;
    S.sel = reinterpret_cast<uint32_t *>(p); p += 256;
    S.kept = reinterpret_cast<uint32_t *>(p); p += 256;
    S.oldl = reinterpret_cast<uint32_t *>(p); p += 256;
    S.osc = reinterpret_cast<float *>(p); p += 256;
    S.misc = reinterpret_cast<uint32_t *>(p);
    return S;
}
static size_t step_lds_bytes() { return (size_t)kDelDm * 4 + 6 * 256 + 64; }


__device__ __forceinline__ void store_canonical_wave(uint32_t *row, uint32_t stride, const uint32_t *ids_lds, uint32_t ns, uint32_t lane) {
    const uint32_t mine = lane < ns ? ids_lds[lane] : kSentinel;
    uint32_t rank = 0;
    for (uint32_t s = 0; s < ns; ++s) rank += ids_lds[s] < mine ? 1u : 0u;
    for (uint32_t t = lane; t < stride; t += 64u)
        if (t >= ns) st_row(row + t, kSentinel);
    if (lane < ns) st_row(row + rank, mine);
}

// S.cur[0..nc) pruned to at most maxn ids around `owner` (prune_row_dev's result, evaluated eagerly): S.kept[0..return).
// G workgroups share the prune: every one of them evaluates its share of the distance matrix into gdm; the LAST one to deliver (one
// relaxed fetch-add behind a workgroup-scope release, hvx_restricted_exact.hip) loads the matrix and replays select_diverse -- the
// others get kSentinel back and leave.
template <uint32_t METRIC, bool FUSED>
__device__ __forceinline__ uint32_t prune_eager(const DevIndex &ix, const StepLds &S, uint32_t owner, uint32_t nc, uint32_t maxn, uint32_t tid, uint32_t g,
                                                uint32_t G, float *gdm, uint32_t *tick, bool *bad) {
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const int j = (int)(lane & 7u);
    // Round 6: the matrix is SYMMETRIC bit for bit -- (a - b)^2 = (b - a)^2, a b = b a, |a - b| = |b - a|, and the cosine finish is symmetric
    // in its two norms -- so only the pairs i > j are evaluated, each by one 8-lane group with BOTH rows read in place (the "query" row
    // through the same pointer arithmetic as a staged copy: the same operands in the same order).  One workgroup evaluating all <= 2 080
    // pairs pulls 12.5 MB through ONE compute unit's vector cache: 48 - 51 us per step kernel, 2 ms of a 2.6-ms delete (rounds 3-5 and the
    // first build of this round).  Spread over G workgroups of 32 row groups every group has about one pair: the matrix costs one row gather.
    __syncthreads();
    const uint32_t npairs = (nc + 1u) * nc / 2u;
    for (uint32_t p = g * (kDelWaves * 8u) + (tid >> 3); p < npairs; p += G * (kDelWaves * 8u)) {
        // p = i (i - 1) / 2 + jj, 0 <= jj < i <= nc
        uint32_t i = (uint32_t)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
        while (i * (i - 1u) / 2u > p) --i;
        while ((i + 1u) * i / 2u <= p) ++i;
        const uint32_t jj = p - i * (i - 1u) / 2u;
        const uint32_t node_i = i < nc ? S.cur[i] : owner, node_j = S.cur[jj];
        const float d = group_distance<METRIC, FUSED>(ix, ix.vec + (size_t)node_i * ix.ld, ix.hdr[node_i], node_j, j);
        if (j == 0) {
            st_agent(gdm + i * 64u + jj, d);
            if (i < nc) st_agent(gdm + jj * 64u + i, d);
        }
    }
    stores_done(); // the stores have been acknowledged before the ticket is taken
    __syncthreads();
    if (tid == 0) S.misc[1] = __hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == G ? 1u : 0u;
    __syncthreads();
    if (!S.misc[1]) return kSentinel;
    if (tid == 0) st_agent(tick, 0u); // (the next launch starts from zero)
    for (uint32_t t = tid; t < (nc + 1u) * 64u; t += kDelWaves * 64u) S.Dm[t] = ld_agent(gdm + t); // (the diagonal is never read)
    __syncthreads();
    if (wave == 0) {
        const float dmine = lane < nc ? S.Dm[nc * 64u + lane] : 0.f;
        const uint32_t v = lane < nc ? S.cur[lane] : kSentinel;
        float chk = dmine;
        if (__ballot(lane < nc && !score_valid(chk)) != 0ull) *bad = true;
        uint32_t rank = 0;
        for (uint32_t t = 0; t < nc; ++t) { // Candidate order: score, then id
            const float dt = S.Dm[nc * 64u + t];
            const uint32_t idt = S.cur[t];
            rank += (dt < dmine || (dt == dmine && idt < v)) ? 1u : 0u;
        }
        if (lane < nc) { S.ord[rank] = lane; S.osc[rank] = dmine; }
        lds_order();
        // select_diverse (mod.rs:822-842): strict < rejects.  The ranked candidates and the selection live in REGISTERS (lane r holds
        // the r-th candidate and its score, lane l the l-th selected one): a step is two lane broadcasts, one LDS gather of the matrix and a
        // ballot -- no LDS write + fence per accepted candidate
        const uint32_t my_ord = lane < nc ? S.ord[lane] : 0u;
        const float my_osc = lane < nc ? S.osc[lane] : 0.f;
        uint32_t my_sel = 0u; // the index (into S.cur) of the lane-th selected candidate
        uint32_t ns = 0;
        for (uint32_t r = 0; r < nc && ns < maxn; ++r) {
            const uint32_t ci = (uint32_t)__builtin_amdgcn_readlane((int)my_ord, (int)r);
            const float sc = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(my_osc), (int)r));
            const bool closer = lane < ns && S.Dm[ci * 64u + my_sel] < sc;
            if (__ballot(closer) == 0ull) {
                if (lane == ns) my_sel = ci;
                ++ns;
            }
        }
        if (lane < ns) S.sel[lane] = my_sel;
        lds_order();
        if (ns < maxn) { // backfill, closest first (mod.rs:845-854)
            const bool have = lane < nc;
            const uint32_t mine = have ? my_ord : kSentinel;
            bool in = false;
            for (uint32_t s = 0; s < ns; ++s) in |= S.sel[s] == mine;
            const unsigned long long fm = __ballot(have && !in);
            const uint32_t rk = (uint32_t)__builtin_popcountll(fm & ((1ull << lane) - 1ull));
            if (have && !in && ns + rk < maxn) S.sel[ns + rk] = mine;
            const uint32_t add = (uint32_t)__builtin_popcountll(fm);
            ns = ns + add < maxn ? ns + add : maxn;
            lds_order();
        }
        if (lane < ns) S.kept[lane] = S.cur[S.sel[lane]];
        if (lane == 0) S.misc[0] = ns;
    }
    __syncthreads();
    return S.misc[0];
}

// relink_neighbor (mutation.rs:1916-1993) for source `ri` of every layer: merge, prune, stage; the neighbours it has gained go to newl.
// grid (layers, g_own): every workgroup of a layer reads the same row and closest-candidate list and arrives at the same merged row;
// without a prune workgroup 0 finishes alone, with one the last workgroup to deliver its share of the matrix does
template <uint32_t METRIC, bool FUSED> __global__ __launch_bounds__(256) void delete_step_own_kernel(DeleteArgs a, uint32_t ri) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t s_go, s_ncur, s_nold;
    const DevIndex &ix = a.ix;
    const uint32_t L = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const StepLds S = carve_step(smem);
    if (tid == 0) s_go = (a.ctl[0] == 0u && ri < a.rel_cnt[L]) ? 1u : 0u;
    __syncthreads();
    if (!s_go) { if (g == 0u && tid == 0) a.new_cnt[L] = 0; return; }
    const uint32_t nb = a.rel[(size_t)L * kDelRelCap + ri];
    const uint32_t maxn = L == 0u ? a.m0 : a.m;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t stride;
    uint32_t *row = del_row(a, nb, L, stride);
    if (wave == 0) {
        const uint32_t v = lane < stride ? ld_row(row + lane) : kSentinel;
        const uint32_t nold = (uint32_t)__builtin_popcountll(__ballot(v != kSentinel));
        const uint32_t tn = a.top_cnt[(size_t)L * kDelRelCap + ri];
        const uint32_t tv = lane < tn ? a.top[((size_t)L * kDelRelCap + ri) * kDelTop + lane] : kSentinel;
        S.oldl[lane] = v;
        lds_order();
        bool in_old = false;
        for (uint32_t s = 0; s < nold; ++s) in_old |= S.oldl[s] == tv;
        const bool add = tv != kSentinel && !in_old;
        const unsigned long long am = __ballot(add);
        const uint32_t ncur = nold + (uint32_t)__builtin_popcountll(am);
        if (ncur <= 64u) {
            if (lane < nold) S.cur[lane] = v;
            if (add) S.cur[nold + (uint32_t)__builtin_popcountll(am & lt)] = tv;
        }
        if (lane == 0) { s_ncur = ncur; s_nold = nold; }
    }
    __syncthreads();
    const uint32_t ncur = s_ncur, nold = s_nold;
    if (ncur > 64u) { if (g == 0u && tid == 0) { a.new_cnt[L] = 0; atomicMax(&a.ctl[0], 5u); } return; }
    bool bad = false;
    uint32_t keepn = ncur;
    if (ncur > maxn) {
        keepn = prune_eager<METRIC, FUSED>(ix, S, nb, ncur, maxn, tid, g, a.g_own, a.gdm + (size_t)L * (1u + kDelTop) * kDelDm, a.tick + L * (1u + kDelTop), &bad);
        if (keepn == kSentinel) return;
    } else {
        if (g != 0u) return;
        if (wave == 0 && lane < ncur) S.kept[lane] = S.cur[lane];
        __syncthreads();
    }
    if (wave == 0) {
        const uint32_t f = lane < keepn ? S.kept[lane] : kSentinel;
        bool was_old = false;
        for (uint32_t s = 0; s < nold; ++s) was_old |= S.oldl[s] == f;
        const bool isnew = f != kSentinel && !was_old;
        const unsigned long long nm = __ballot(isnew);
        if (keepn > stride) { if (lane == 0) { a.new_cnt[L] = 0; atomicMax(&a.ctl[0], 5u); } return; }
        store_canonical_wave(row, stride, S.kept, keepn, lane);
        if (isnew) a.newl[(size_t)L * kDelTop + (uint32_t)__builtin_popcountll(nm & lt)] = f;
        if (lane == 0) {
            a.new_cnt[L] = (uint32_t)__builtin_popcountll(nm);
            if (bad) atomicMax(&a.ctl[0], 4u);
        }
    }
}

// ... and the reciprocal row of each of them (mutation.rs:1994-2052): g_recip workgroups per new neighbour (grid x = neighbour x g_recip)
template <uint32_t METRIC, bool FUSED> __global__ __launch_bounds__(256) void delete_step_recip_kernel(DeleteArgs a, uint32_t ri) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t s_go, s_rdeg, s_has;
    const DevIndex &ix = a.ix;
    const uint32_t L = blockIdx.y, bx = blockIdx.x / a.g_recip, g = blockIdx.x % a.g_recip, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const StepLds S = carve_step(smem);
    if (tid == 0) s_go = (a.ctl[0] == 0u && ri < a.rel_cnt[L] && bx < a.new_cnt[L]) ? 1u : 0u;
    __syncthreads();
    if (!s_go) return;
    const uint32_t nb = a.rel[(size_t)L * kDelRelCap + ri], nw = a.newl[(size_t)L * kDelTop + bx];
    const uint32_t maxn = L == 0u ? a.m0 : a.m;
    uint32_t rstride;
    uint32_t *rrow = del_row(a, nw, L, rstride);
    if (wave == 0) {
        uint32_t rv = lane < rstride ? ld_row(rrow + lane) : kSentinel;
        uint32_t rdeg = (uint32_t)__builtin_popcountll(__ballot(rv != kSentinel));
        const bool has = __ballot(rv == nb) != 0ull;
        if (!has && rdeg < 64u) {
            if (lane == rdeg) rv = nb;
            ++rdeg;
            if (lane < rdeg) S.cur[lane] = rv;
        } else if (!has) {
            rdeg = 65u;
        }
        if (lane == 0) { s_rdeg = rdeg; s_has = has ? 1u : 0u; }
    }
    __syncthreads();
    if (s_has) return;
    const uint32_t rdeg = s_rdeg;
    if (rdeg > 64u) { if (g == 0u && tid == 0) atomicMax(&a.ctl[0], 5u); return; }
    bool bad = false;
    uint32_t kn = rdeg;
    if (rdeg > maxn) {
        kn = prune_eager<METRIC, FUSED>(ix, S, nw, rdeg, maxn, tid, g, a.g_recip, a.gdm + ((size_t)L * (1u + kDelTop) + 1u + bx) * kDelDm,
                                        a.tick + L * (1u + kDelTop) + 1u + bx, &bad);
        if (kn == kSentinel) return;
    } else {
        if (g != 0u) return;
        if (wave == 0 && lane < rdeg) S.kept[lane] = S.cur[lane];
        __syncthreads();
    }
    if (wave == 0) {
        if (kn <= rstride) store_canonical_wave(rrow, rstride, S.kept, kn, lane);
        if (lane == 0 && (bad || kn > rstride)) atomicMax(&a.ctl[0], bad ? 4u : 5u);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// A step as ONE launch (round 6): the reciprocal rows' matrices do not have to wait
// for the source's prune -- the neighbours a source can gain are known before it (its closest candidates that its row does not hold: the
// prune only decides which of them stay), and their rows do not change during the step until their own reciprocal update.  So every
// workgroup of a layer derives the same task list -- task 0: the source's merged row, task t: "candidate t's row + the source" --, the
// pairs of ALL tasks are spread over the layer's workgroups (one 8-lane group per pair), and the LAST workgroup to deliver replays the
// prunes from LDS: wavefront 0 the source's, then sixteen wavefronts the reciprocal rows of the candidates that stayed.  One launch and one
// ticket per relinked row instead of two launches and up to 33 tickets (37 us per step -> see profiles/r06n_*).
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kFuseWaves = 16;
// NARROW: rows of at most 34 ids (degree limits <= 32 on canonical rows): a source's merged row <= 64 ids, a reciprocal list <= 35 -- one
// candidate per lane, the matrix rows in registers (replay_rows).  WIDE (round 6): rows of up to 64 ids (the reference's scale fixture
// runs M 32 / M0 64, scale_contracts.rs:167-173): merged row <= 128 ids, reciprocal list <= 65 -- two candidates per lane, rows streamed
// (replay_rows2).
template <bool WIDE> struct FuseT;
template <> struct FuseT<false> {
    static constexpr bool wide = false;
    static constexpr uint32_t NCMAX = 64, OWN_RW = 64, ROWMAX = 35, RS = 36, RECIP_RW = 36, WSC = 128;
};
template <> struct FuseT<true> {
    static constexpr bool wide = true;
    static constexpr uint32_t NCMAX = 128, OWN_RW = 128, ROWMAX = 66, RS = 68, RECIP_RW = 96, WSC = kWide2Words;
};
// a prune's distances in HBM: the square matrix over [ids..., owner], row i = distances of id i to the others (row nc: the owner's), RW wide
template <typename T> constexpr uint32_t fuse_own_dm() { return (T::NCMAX + 1u) * T::OWN_RW; }
template <typename T> constexpr uint32_t fuse_recip_dm() { return (T::ROWMAX + 1u) * T::RECIP_RW; }
template <typename T> constexpr uint32_t fuse_layer_dm() { return fuse_own_dm<T>() + kDelTop * fuse_recip_dm<T>(); } // floats of one layer's matrices

struct FuseLds {
    uint32_t *rcur;       // [kDelTop][RS] the candidates' rows with the source appended
    uint32_t *rdeg;       // [kDelTop] ids of that list; 0: the row holds the source already
    uint32_t *cur, *oldl; // [NCMAX] the source's merged row / [64] its row before
    uint32_t *addl;       // [kDelTop] the candidates the source's row does not hold, Candidate order
    uint32_t *isnew;      // [kDelTop] ... that stayed after the prune
    uint32_t *pbase;      // [kDelTop + 2] first pair of every task
    uint32_t *wsc;        // [16][WSC] per-wavefront replay scratch
    uint32_t *misc;       // [8]
};
template <typename T> __device__ __forceinline__ FuseLds carve_fuse(char *smem) {
    FuseLds S;
    char *p = smem;
    S.rcur = reinterpret_cast<uint32_t *>(p); p += kDelTop * T::RS * 4;
    S.rdeg = reinterpret_cast<uint32_t *>(p); p += kDelTop * 4;
    S.cur = reinterpret_cast<uint32_t *>(p); p += T::NCMAX * 4;
    S.oldl = reinterpret_cast<uint32_t *>(p); p += 256;
    S.addl = reinterpret_cast<uint32_t *>(p); p += kDelTop * 4;
    S.isnew = reinterpret_cast<uint32_t *>(p); p += kDelTop * 4;
    S.pbase = reinterpret_cast<uint32_t *>(p); p += (kDelTop + 2) * 4 + 8;
    S.wsc = reinterpret_cast<uint32_t *>(p); p += kFuseWaves * T::WSC * 4;
    S.misc = reinterpret_cast<uint32_t *>(p);
    return S;
}
template <typename T> static size_t fuse_lds_bytes() {
    return (size_t)kDelTop * T::RS * 4 + kDelTop * 4 * 3 + T::NCMAX * 4 + 256 + (kDelTop + 2) * 4 + 8 + kFuseWaves * T::WSC * 4 + 64;
}

template <uint32_t METRIC, bool FUSED, bool BF, bool WIDE> __global__ __launch_bounds__(1024) void delete_step_fused_kernel(DeleteArgs a, uint32_t ri) {
    using T = FuseT<WIDE>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t s_go, s_ncur, s_nadd, s_err, s_last, s_keepn;
    const DevIndex &ix = a.ix;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    // workgroup -> (layer, share): layer 0 has g_own workgroups, every upper layer g_recip
    uint32_t L, g, G;
    if (blockIdx.x < a.g_own) { L = 0u; g = blockIdx.x; G = a.g_own; }
    else { L = 1u + (blockIdx.x - a.g_own) / a.g_recip; g = (blockIdx.x - a.g_own) % a.g_recip; G = a.g_recip; }
    const FuseLds S = carve_fuse<T>(smem);
    if (tid == 0) { s_go = (a.ctl[0] == 0u && ri < a.rel_cnt[L]) ? 1u : 0u; s_err = 0u; }
    __syncthreads();
    if (!s_go) return;
    const uint32_t nb = a.rel[(size_t)L * kDelRelCap + ri];
    const uint32_t maxn = L == 0u ? a.m0 : a.m;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t stride;
    uint32_t *row = del_row(a, nb, L, stride);
    // ---- the tasks (every workgroup of the layer arrives at the same lists) ----
    if (wave == 0) {
        const uint32_t v = lane < stride ? ld_row(row + lane) : kSentinel;
        const uint32_t nold = (uint32_t)__builtin_popcountll(__ballot(v != kSentinel));
        const uint32_t tn = a.top_cnt[(size_t)L * kDelRelCap + ri];
        const uint32_t tv = lane < tn ? a.top[((size_t)L * kDelRelCap + ri) * kDelTop + lane] : kSentinel;
        S.oldl[lane] = v;
        lds_order();
        bool in_old = false;
        for (uint32_t s = 0; s < nold; ++s) in_old |= S.oldl[s] == tv;
        const bool add = tv != kSentinel && !in_old;
        const unsigned long long am = __ballot(add);
        const uint32_t nadd = (uint32_t)__builtin_popcountll(am), ncur = nold + nadd;
        if (ncur <= T::NCMAX) {
            if (lane < nold) S.cur[lane] = v;
            if (add) {
                const uint32_t at = (uint32_t)__builtin_popcountll(am & lt);
                S.cur[nold + at] = tv;
                S.addl[at] = tv;
            }
        }
        if (lane == 0) { s_ncur = ncur; s_nadd = nadd; }
    }
    __syncthreads();
    const uint32_t ncur = s_ncur, nadd = s_nadd;
    if (ncur > T::NCMAX) { if (g == 0u && tid == 0) atomicMax(&a.ctl[0], 5u); return; }
    for (uint32_t t = wave; t < nadd; t += kFuseWaves) { // the candidates' rows, the source appended (mutation.rs:1994-2006)
        const uint32_t nw = S.addl[t];
        uint32_t rstride;
        const uint32_t *rrow = del_row(a, nw, L, rstride);
        uint32_t rv = lane < rstride ? ld_row(rrow + lane) : kSentinel;
        uint32_t rdeg = (uint32_t)__builtin_popcountll(__ballot(rv != kSentinel));
        const bool has = __ballot(rv == nb) != 0ull;
        if (has) {
            rdeg = 0u;
        } else if (rdeg + 1u > T::ROWMAX) {
            if (lane == 0) s_err = 1u;
            rdeg = 0u;
        } else { // rows are canonical: the valid ids occupy lanes 0..rdeg-1; the source goes behind them (position 64 of a full wide row)
            if (lane < rdeg) S.rcur[t * T::RS + lane] = rv;
            if (lane == 0) S.rcur[t * T::RS + rdeg] = nb;
            ++rdeg;
        }
        if (lane == 0) S.rdeg[t] = rdeg;
    }
    __syncthreads();
    if (s_err) { if (g == 0u && tid == 0) atomicMax(&a.ctl[0], 5u); return; }
    if (wave == 0) { // first pair of every task: 0 the source's prune, 1 + t candidate t's (a list within its limit needs no matrix)
        const uint32_t n0 = ncur > maxn ? (ncur + 1u) * ncur / 2u : 0u;
        const uint32_t rd = lane < nadd ? S.rdeg[lane] : 0u;
        const uint32_t np = rd > maxn ? (rd + 1u) * rd / 2u : 0u;
        uint32_t incl = np;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const uint32_t o = __shfl_up(incl, sft, 64);
            if ((int)lane >= sft) incl += o;
        }
        if (lane == 0) S.pbase[0] = 0u;
        if (lane < kDelTop) S.pbase[1u + lane] = n0 + incl - np;
        if (lane == kDelTop - 1u) S.pbase[kDelTop + 1u] = n0 + incl;
    }
    __syncthreads();
    const uint32_t total = S.pbase[kDelTop + 1u];
    float *gl = a.gdm + (size_t)L * fuse_layer_dm<T>();
    bool last;
    { // (the ticket is taken even when nothing needs a matrix: the last workgroup rewrites rows the others are still reading their tasks from)
        const int j = (int)(lane & 7u);
        for (uint32_t p = g * (kFuseWaves * 8u) + (tid >> 3); p < total; p += G * (kFuseWaves * 8u)) {
            uint32_t lo = 0u, hi = nadd + 1u; // the task t with pbase[t] <= p < pbase[t + 1]
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (S.pbase[mid] <= p) lo = mid; else hi = mid;
            }
            const uint32_t t = lo, q = p - S.pbase[t];
            // q = i (i - 1) / 2 + jj, 0 <= jj < i <= nc
            uint32_t i = (uint32_t)((1.0f + sqrtf(1.0f + 8.0f * (float)q)) * 0.5f);
            while (i * (i - 1u) / 2u > q) --i;
            while ((i + 1u) * i / 2u <= q) ++i;
            const uint32_t jj = q - i * (i - 1u) / 2u;
            uint32_t node_i, node_j, nc, st;
            float *out;
            if (t == 0u) {
                nc = ncur; st = T::OWN_RW; out = gl;
                node_i = i < nc ? S.cur[i] : nb; node_j = S.cur[jj];
            } else {
                const uint32_t *rc = S.rcur + (t - 1u) * T::RS;
                nc = S.rdeg[t - 1u]; st = T::RECIP_RW; out = gl + fuse_own_dm<T>() + (size_t)(t - 1u) * fuse_recip_dm<T>();
                node_i = i < nc ? rc[i] : S.addl[t - 1u]; node_j = rc[jj];
            }
            const float d = pair_distance<METRIC, FUSED, BF>(ix, node_i, node_j, j);
            if (j == 0) {
                st_agent(out + i * st + jj, d);
                if (i < nc) st_agent(out + jj * st + i, d);
            }
        }
        stores_done(); // the stores have been acknowledged before the ticket is taken
        __syncthreads();
        if (tid == 0) s_last = __hip_atomic_fetch_add(a.tick + L, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == G ? 1u : 0u;
        __syncthreads();
        last = s_last != 0u;
        if (last && tid == 0) st_agent(a.tick + L, 0u); // (the next launch starts from zero)
    }
    if (!last) return;
    // ---- the last workgroup.  The reciprocal prunes do not depend on the source's (only whether they are APPLIED does): wavefront 0
    // replays the source's prune while wavefronts 1 .. 15 replay the candidates' lists, then the rows of the candidates that stayed go out
    bool bad = false;
    uint32_t *wsc = S.wsc + wave * T::WSC;
    const WaveScratch W{wsc, wsc + (T::wide ? 256u + 512u + 128u : 64u)}; // (W.kept = the replay's kept list in either layout)
    const uint32_t first = wave - 1u; // (wavefront 0: none)
    __syncthreads();
    if (wave == 0) {
        uint32_t keepn = ncur;
        if (ncur > maxn) {
            if constexpr (T::wide) keepn = replay_rows2<(int)T::OWN_RW>(gl, S.cur, ncur, maxn, lane, wsc, &bad);
            else keepn = replay_rows<(int)T::OWN_RW>(gl, S.cur, ncur, maxn, lane, W, &bad);
        }
        else { if (lane < ncur) W.kept[lane] = S.cur[lane]; lds_order(); }
        const uint32_t kf = lane < keepn ? W.kept[lane] : kSentinel;
        const uint32_t at = lane < nadd ? S.addl[lane] : kSentinel;
        bool stays = false;
        for (uint32_t s = 0; s < keepn; ++s) stays |= (uint32_t)__builtin_amdgcn_readlane((int)kf, (int)s) == at;
        if (lane < kDelTop) S.isnew[lane] = (lane < nadd && stays) ? 1u : 0u;
        if (keepn > stride) { if (lane == 0) atomicMax(&a.ctl[0], 5u); keepn = kSentinel; }
        else store_canonical_reg(row, stride, kf, keepn, lane);
        if (lane == 0) { s_keepn = keepn; if (bad) atomicMax(&a.ctl[0], 4u); }
    } else {
        for (uint32_t t = first; t < nadd; t += kFuseWaves - 1u) {
            const uint32_t rdeg = S.rdeg[t];
            if (rdeg == 0u) continue;
            uint32_t *rc = S.rcur + t * T::RS;
            if (rdeg > maxn) {
                bool rbad = false;
                const float *gm = gl + fuse_own_dm<T>() + (size_t)t * fuse_recip_dm<T>();
                uint32_t kn;
                if constexpr (T::wide) kn = replay_rows2<(int)T::RECIP_RW>(gm, rc, rdeg, maxn, lane, wsc, &rbad);
                else kn = replay_rows<(int)T::RECIP_RW>(gm, rc, rdeg, maxn, lane, W, &rbad);
                if (lane < kn) rc[lane] = W.kept[lane]; // the list is replaced by what stays of it
                if (lane == 0) { S.rdeg[t] = kn; if (rbad) atomicMax(&a.ctl[0], 4u); }
                lds_order();
            }
        }
    }
    __syncthreads();
    if (s_keepn == kSentinel) return;
    // the reciprocal row of every neighbour the source has gained (mutation.rs:1994-2052)
    for (uint32_t t = wave; t < nadd; t += kFuseWaves) {
        const uint32_t kn = S.rdeg[t];
        if (!S.isnew[t] || kn == 0u) continue;
        uint32_t rstride;
        uint32_t *rrow = del_row(a, S.addl[t], L, rstride);
        if (kn <= rstride) store_canonical_reg(rrow, rstride, lane < kn ? S.rcur[t * T::RS + lane] : kSentinel, kn, lane);
        else if (lane == 0) atomicMax(&a.ctl[0], 5u);
    }
}

// find_best_entry_candidate (mutation.rs:350-394): the live node on the highest layer, smallest id first
__global__ __launch_bounds__(256) void delete_best_entry_kernel(DevIndex ix, const uint32_t *dead, unsigned long long *out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    unsigned long long key = 0ull;
    if (i < ix.n && ((dead[i >> 5] >> (i & 31u)) & 1u) == 0u) key = (((unsigned long long)ix.level[i] + 1ull) << 32) | (0xFFFFFFFFu - i);
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const unsigned long long o = __shfl_xor(key, s, 64);
        key = o > key ? o : key;
    }
    if ((threadIdx.x & 63u) == 0u && key) atomicMax(out, key);
}

using DeleteKernel = void (*)(DeleteArgs);
using DeleteStepKernel = void (*)(DeleteArgs, uint32_t);
struct DeleteKernels { DeleteKernel rank, relink; DeleteStepKernel own, recip, fused, fused_wide; };
template <uint32_t METRIC, bool FUSED> static DeleteKernels delete_kernels_of() {
    return {delete_rank_kernel<METRIC, FUSED, false>, delete_relink_kernel<METRIC, FUSED>, delete_step_own_kernel<METRIC, FUSED>, delete_step_recip_kernel<METRIC, FUSED>,
            delete_step_fused_kernel<METRIC, FUSED, false, false>, delete_step_fused_kernel<METRIC, FUSED, false, true>};
}
// bf16 images: the ranking and the fused steps read the interleaved rows (the one-wavefront kernel only retires the node there)
template <uint32_t METRIC> static DeleteKernels delete_kernels_bf16() {
    return {delete_rank_kernel<METRIC, true, true>, delete_relink_kernel<METRIC, true>, nullptr, nullptr, delete_step_fused_kernel<METRIC, true, true, false>,
            delete_step_fused_kernel<METRIC, true, true, true>};
}
static DeleteKernels pick_delete_kernels(uint32_t metric, bool fused, bool bf16) {
    if (bf16) return metric == kL2 ? delete_kernels_bf16<kL2>() : delete_kernels_bf16<kCosine>();
    if (metric == kL2) return fused ? delete_kernels_of<kL2, true>() : delete_kernels_of<kL2, false>();
    if (metric == kCosine) return fused ? delete_kernels_of<kCosine, true>() : delete_kernels_of<kCosine, false>();
    return fused ? delete_kernels_of<kL1, true>() : delete_kernels_of<kL1, false>();
}

} // namespace hvx

extern "C" int hvx_index_delete_batch(hvx_index *ix, const uint64_t *node_ids, uint32_t count, hvx_delete_stats *stats) {
    if (!ix || (count && !node_ids)) return fail(HVX_ERR_INVARIANT, "null argument");
    if (stats) memset(stats, 0, sizeof(*stats));
    if (stats) stats->requested = count;
    if (count == 0) return HVX_OK;
    if (ix->is_fork) return fail(HVX_ERR_UNSUPPORTED, "nodes are deleted through the handle that owns the image, not a fork");
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    DevIndex &d = ix->dev;
    if (d.dtype != HVX_F32 && d.dtype != HVX_BF16)
        return fail(HVX_ERR_UNSUPPORTED, "nodes are deleted from f32 and bf16 images (fp8 rows: import the changed graph afterwards)");
    const bool bf16 = d.dtype == HVX_BF16;
    const uint32_t m = ix->desc.m ? ix->desc.m : 16u;
    const uint32_t m0 = std::max(ix->desc.m0 ? ix->desc.m0 : 2u * m, 2u * m);
    if (m0 > kDelTop || m > kDelTop || d.s0 > 64u || d.su > 64u || d.s0 < m0 || d.su < m)
        return fail(HVX_ERR_UNSUPPORTED, "device delete serves degree limits <= 64 on rows at least that wide and at most 64 ids (m %u m0 %u strides %u / %u)", m, m0, d.su, d.s0);
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t s = ix->stream;
    const uint64_t cap = std::max<uint64_t>(ix->cap_rows, d.n);
    const uint32_t words = (uint32_t)((cap + 31u) / 32u) + 1u;
    int rc;
    if (!ix->d_dead) { // the image's deleted-row bitmap: one bit per row it can ever hold
        if ((rc = ix->dalloc((void **)&ix->d_dead, (size_t)words * 4))) return rc;
        HIP_TRY(hipMemsetAsync(ix->d_dead, 0, (size_t)words * 4, s));
    }
    d.dead = ix->d_dead;
    const uint32_t layers_now = d.max_layer + 1u;
    if (!ix->del_scratch || ix->del_layers < layers_now) {
        const uint32_t layers = std::max(kDelMinLayers, layers_now);
        const size_t per_layer = ((size_t)kDelSrcCap + kDelRelCap + kDelCandCap + (size_t)kDelRelCap * kDelTop + kDelRelCap + words + 4u + kDelTop +
                                  std::max<size_t>((size_t)(1u + kDelTop) * (kDelDm + 1u), (size_t)fuse_layer_dm<FuseT<true>>() + 2u + kDelTop)) * 4u;
        void *p = nullptr;
        if ((rc = ix->dalloc(&p, per_layer * layers + 64))) return rc;
        HIP_TRY(hipMemsetAsync(p, 0, per_layer * layers + 64, s));
        ix->del_scratch = p; // (an earlier, smaller scratch stays with the image's allocations until the handle goes)
        ix->del_layers = layers;
    }
    DeleteArgs a{};
    {
        const uint32_t layers = ix->del_layers;
        uint32_t *p = reinterpret_cast<uint32_t *>(ix->del_scratch);
        a.ctl = p; p += 16;
        a.src_cnt = p; p += layers;
        a.rel_cnt = p; p += layers;
        a.cand_cnt = p; p += layers;
        a.new_cnt = p; p += layers;
        a.newl = p; p += (size_t)layers * kDelTop;
        a.src = p; p += (size_t)layers * kDelSrcCap;
        a.rel = p; p += (size_t)layers * kDelRelCap;
        a.cand = p; p += (size_t)layers * kDelCandCap;
        a.top_cnt = p; p += (size_t)layers * kDelRelCap;
        a.top = p; p += (size_t)layers * kDelRelCap * kDelTop;
        a.gdm = reinterpret_cast<float *>(p); p += (size_t)layers * std::max<size_t>((size_t)(1u + kDelTop) * kDelDm, fuse_layer_dm<FuseT<true>>());
        a.tick = p; p += (size_t)layers * (1u + kDelTop);
        a.mark = p;
    }
    a.l0 = const_cast<uint32_t *>(d.l0);
    a.up = const_cast<uint32_t *>(d.up);
    a.m = m; a.m0 = m0; a.words = words;
    a.dead = ix->d_dead;
    HIP_TRY(hipMemsetAsync(a.ctl, 0, 16, s));
    const DeleteKernels kern = pick_delete_kernels(d.metric, kernel_fused(d.fkernel), bf16);
    const size_t rank_lds = (((size_t)d.ld * 4u + 15u) & ~(size_t)15u) + 16u * 64u * 8u;
    const size_t relink_lds = (((size_t)d.ld * 4u + 15u) & ~(size_t)15u) + 4 * 256 + 256;
    if (rank_lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kern.rank, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rank_lds));
    if (relink_lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kern.relink, hipFuncAttributeMaxDynamicSharedMemorySize, (int)relink_lds));
    const size_t step_lds = step_lds_bytes();
    const bool steps = ix->opt[HVX_OPT_DELETE_SEQUENTIAL] != 1u;
    // one launch per step while a reciprocal list (row + the source) fits the fused kernel's LDS matrices; HVX_OPT_DELETE_SEQUENTIAL = 2
    // keeps the two-launch steps (A/B, and what wider rows take)
    const bool narrow = d.s0 + 1u <= FuseT<false>::ROWMAX && d.su + 1u <= FuseT<false>::ROWMAX && m0 <= 32u && m <= 32u;
    const bool fused_steps = steps && (ix->opt[HVX_OPT_DELETE_SEQUENTIAL] != 2u || !narrow); // (rows of up to 64 ids: the wide build of the fused step)
    if (bf16 && !fused_steps)
        return fail(HVX_ERR_UNSUPPORTED, "a bf16 image relinks by fused steps only (HVX_OPT_DELETE_SEQUENTIAL unset)");
    if (!narrow && !fused_steps)
        return fail(HVX_ERR_UNSUPPORTED, "rows wider than 34 ids / degree limits above 32 relink by fused steps only (HVX_OPT_DELETE_SEQUENTIAL unset)");
    const DeleteStepKernel fused_kernel = narrow ? kern.fused : kern.fused_wide;
    const size_t fused_lds = narrow ? fuse_lds_bytes<FuseT<false>>() : fuse_lds_bytes<FuseT<true>>();
    const uint32_t groups = kDelWaves * 8u;
    if (fused_steps) {
        // workgroups (128 row groups each) of layer 0 / of every upper layer: a typical step has ~8 000 pairs on layer 0 (the source's
        // <= 2 080 and ~10 reciprocal lists of 561), at most 20 032; the wide build: <= 8 256 + 64 x 2 145
        a.g_own = narrow ? 96u : 192u;
        a.g_recip = narrow ? 12u : 24u;
        HIP_TRY(hipFuncSetAttribute((const void *)fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds));
    } else {
        // workgroups (32 row groups each) that share one prune's distance matrix: about one pair per row group.  A source's row grows to
        // at most 64 ids (65 x 64 / 2 pairs), a reciprocal row to Mmax + 1
        a.g_own = (65u * 64u / 2u + groups - 1u) / groups;
        a.g_recip = ((m0 + 2u) * (m0 + 1u) / 2u + groups - 1u) / groups;
    }
    a.first = steps ? kDelRelCap : 0u; // every source runs as a step (their number is read back below); the one-wavefront kernel only retires the node
    // this generation's deleted-row flags: a copy of the visible ones (forks keep theirs until hvx_index_refresh)
    auto flags = std::make_shared<std::vector<uint8_t>>((size_t)cap, (uint8_t)0);
    if (ix->dead_p) std::copy(ix->dead_p->begin(), ix->dead_p->begin() + std::min<size_t>(ix->dead_p->size(), (size_t)cap), flags->begin());
    uint32_t deleted = 0, missing = 0, entry_moves = 0, n_dead = ix->n_dead;
    auto read_ctl = [&](uint32_t *out4) -> int {
        if ((rc = ix->pin_flags(4))) return rc;
        HIP_TRY(hipMemcpyAsync(ix->h_flags, a.ctl, 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        memcpy(out4, ix->h_flags, 16);
        return HVX_OK;
    };
    for (uint32_t i = 0; i < count; ++i) {
        const uint32_t row = ix->find_slot(node_ids[i]);
        if (row == kSentinel || row >= d.n || (*flags)[row]) { ++missing; continue; } // an unknown id succeeds and stages nothing (index.rs:2263, 2294-2295)
        a.ix = d;
        a.node = row;
        a.layers = d.max_layer + 1u;
        const unsigned long long slots = (unsigned long long)d.n * d.s0;
        const uint32_t scan_blocks = (uint32_t)std::min<unsigned long long>(4096ull, std::max<unsigned long long>(1ull, (slots + 255ull) / 256ull));
        hipLaunchKernelGGL(delete_scan_kernel, dim3(scan_blocks), dim3(256), 0, s, a);
        hipLaunchKernelGGL(delete_prep_kernel, dim3(a.layers), dim3(256), 0, s, a);
        // how many relink sources the layers have: read back behind the prep kernel WHILE the rank kernel runs (round 6: the r05w trace had
        // 64 000 own + 64 000 reciprocal step dispatches for 2 000 deletes of ~25 relinked rows each -- two launches of ~4 us per absent step)
        uint32_t n_steps = 0;
        if (steps) {
            if ((rc = ix->pin_flags(4 + kDelMinLayers + a.layers))) return rc;
            HIP_TRY(hipMemcpyAsync(ix->h_flags + 4, a.rel_cnt, (size_t)a.layers * 4, hipMemcpyDeviceToHost, s));
            if (!ix->del_ev) HIP_TRY(hipEventCreateWithFlags(&ix->del_ev, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(ix->del_ev, s));
        }
        hipLaunchKernelGGL(kern.rank, dim3(64, a.layers), dim3(1024), rank_lds, s, a);
        if (steps) {
            HIP_TRY(hipEventSynchronize(ix->del_ev));
            uint32_t most = 0;
            for (uint32_t l = 0; l < a.layers; ++l) most = std::max(most, ix->h_flags[4 + l]);
            n_steps = std::min(kDelRelCap, most);
        }
        for (uint32_t ri = 0; steps && ri < n_steps; ++ri) { // (a step past a layer's last source returns at once)
            if (fused_steps) {
                hipLaunchKernelGGL(fused_kernel, dim3(a.g_own + (a.layers - 1u) * a.g_recip), dim3(kFuseWaves * 64u), fused_lds, s, a, ri);
                continue;
            }
            hipLaunchKernelGGL(kern.own, dim3(a.layers, a.g_own), dim3(kDelWaves * 64u), step_lds, s, a, ri);
            hipLaunchKernelGGL(kern.recip, dim3(kDelTop * a.g_recip, a.layers), dim3(kDelWaves * 64u), step_lds, s, a, ri);
        }
        hipLaunchKernelGGL(kern.relink, dim3(a.layers), dim3(64), relink_lds, s, a);
        HIP_TRY(hipGetLastError());
        (*flags)[row] = 1;
        ++n_dead;
        ++deleted;
        if (d.has_entry && row == d.entry) { // :1756-1767 the entry point moves to the best remaining entry candidate
            HIP_TRY(hipMemsetAsync(a.ctl + 2, 0, 8, s));
            hipLaunchKernelGGL(delete_best_entry_kernel, dim3((d.n + 255u) / 256u), dim3(256), 0, s, d, ix->d_dead, reinterpret_cast<unsigned long long *>(a.ctl + 2));
            HIP_TRY(hipGetLastError());
            uint32_t c4[4];
            if ((rc = read_ctl(c4))) return rc;
            if (c4[0]) break; // reported below
            const unsigned long long key = ((unsigned long long)c4[3] << 32) | c4[2];
            if (key == 0ull) { d.has_entry = 0; d.entry = 0; d.max_layer = 0; }
            else { d.entry = 0xFFFFFFFFu - (uint32_t)key; d.max_layer = (uint32_t)(key >> 32) - 1u; }
            ++entry_moves;
        }
    }
    uint32_t c4[4];
    if ((rc = read_ctl(c4))) return rc;
    if (c4[0]) {
        static const char *why[] = {"", "more than 4096 rows hold one node on a layer", "more than 4096 relink sources on a layer", "more than 16384 relink candidates on a layer",
                                    "a distance is not a valid score (Candidate::try_new fails: the reference aborts the delete)", "a neighbour row overflowed its stride"};
        const uint32_t code = c4[0] < 6u ? c4[0] : 5u;
        (void)hipMemsetAsync(ix->del_scratch, 0, 64 + (size_t)ix->del_layers * 4u * 4u, s); // counters; the marks of an aborted layer may be dirty:
        ix->del_scratch = nullptr;                                                           // ... the next call starts from fresh scratch
        ix->del_layers = 0;
        return fail(code == 4u ? HVX_ERR_INVARIANT : HVX_ERR_UNSUPPORTED, "device delete: %s -- the image is partially relinked: discard the handle and hydrate again", why[code]);
    }
    // ---- the new generation ----
    ix->dead_p = flags;
    ix->n_dead = n_dead;
    ix->desc.has_entry = d.has_entry;
    ix->desc.max_layer = d.max_layer;
    ix->desc.entry_point = d.has_entry ? ix->ids_ref()[d.entry] : 0;
    if (deleted) {
        ix->publish_view();
        ix->seen_rewrite = ix->shared->rewrite_epoch.fetch_add(1, std::memory_order_acq_rel) + 1; // lanes adopt this generation at their next launch
    }
    if (stats) {
        stats->deleted = deleted;
        stats->missing = missing;
        stats->entry_moves = entry_moves;
        stats->relinked_rows = c4[1];
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return HVX_OK;
}
