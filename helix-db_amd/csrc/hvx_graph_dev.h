// hvx_graph_dev.h -- device helpers shared by the kernels that CHANGE neighbour rows: the build / insert path (hvx_build.hip) and the
// delete path (hvx_delete.hip).  One wavefront works on one row owner at a time: its vector is staged in LDS as the "query" of the
// reference-order distance evaluator (hvx_device.h: group_distance), candidates are ranked in Candidate order (model.rs:55-61)
// and pruned by select_diverse + backfill (mod.rs:809-856), rows are stored in canonical form (neighbor_set.rs:1-9).
#pragma once
#include <hip/hip_runtime.h>

#include "hvx_device.h"

namespace hvx {

// neighbour rows that other workgroups may be changing are read and written past the non-coherent L1
__device__ __forceinline__ uint32_t ld_row(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_row(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }


// LDS of one wavefront: the "query" row of the distance evaluator + small id / score lists
struct BuildLds {
    float *qv;       // [ld]
    uint32_t *kept;  // [64]
    uint32_t *cid;   // [64] candidates sorted by (score, id)
    float *csc;      // [64]
    float *dtmp;     // [64]
};
__device__ __forceinline__ BuildLds carve_build(char *smem, uint32_t ld) {
    BuildLds L;
    L.qv = reinterpret_cast<float *>(smem);
    char *p = smem + (((size_t)ld * 4u + 15u) & ~(size_t)15u);
    L.kept = reinterpret_cast<uint32_t *>(p); p += 256;
    L.cid = reinterpret_cast<uint32_t *>(p); p += 256;
    L.csc = reinterpret_cast<float *>(p); p += 256;
    L.dtmp = reinterpret_cast<float *>(p);
    return L;
}
static size_t build_lds_bytes(uint32_t ld) { return (((size_t)ld * 4u + 15u) & ~(size_t)15u) + 4 * 256; }

__device__ __forceinline__ void stage_row(const DevIndex &ix, float *qv, uint32_t node, int lane) {
    __syncthreads();
    const float *r = ix.vec + (size_t)node * ix.ld;
    for (uint32_t t = (uint32_t)lane; t < ix.ld; t += 64) qv[t] = r[t];
    __syncthreads();
}

// mod.rs:809-856 select_diverse over L.cid/L.csc[0..hyd) (sorted closest first, all hydrated), at most m kept, then the
// backfill with the closest remaining candidates (:845-854).  L.kept[0..ns) = the selection in selection order.
template <uint32_t METRIC, bool FUSED>
__device__ __forceinline__ uint32_t select_diverse_dev(const DevIndex &ix, const BuildLds &L, uint32_t hyd, uint32_t m, int lane) {
    const int grp = lane >> 3, j = lane & 7;
    uint32_t ns = 0;
    for (uint32_t i = 0; i < hyd && ns < m; ++i) {
        const uint32_t ci = L.cid[i];
        const float si = L.csc[i];
        bool diverse = true;
        if (ns) {
            stage_row(ix, L.qv, ci, lane);
            const float chdr = ix.hdr[ci];
            for (uint32_t p0 = 0; p0 < ns; p0 += 8) {
                const uint32_t g = p0 + (uint32_t)grp;
                const uint32_t other = L.kept[g < ns ? g : ns - 1u];
                const float pd = group_distance<METRIC, FUSED>(ix, L.qv, chdr, other, j);
                if (__ballot(g < ns && pd < si)) { diverse = false; break; } // strict < rejects (mod.rs:832)
            }
        }
        if (diverse) {
            __syncthreads();
            if (lane == 0) L.kept[ns] = ci;
            ++ns;
            __syncthreads();
        }
    }
    if (ns < m) { // backfill, closest first
        const bool have = (uint32_t)lane < hyd;
        const uint32_t mine = have ? L.cid[lane] : kSentinel;
        bool in = false;
        for (uint32_t s = 0; s < ns; ++s) in |= L.kept[s] == mine;
        const unsigned long long free_m = __ballot(have && !in);
        const uint32_t rank = (uint32_t)__builtin_popcountll(free_m & ((1ull << lane) - 1ull));
        __syncthreads();
        if (have && !in && ns + rank < m) L.kept[ns + rank] = mine;
        const uint32_t add = (uint32_t)__builtin_popcountll(free_m);
        ns = ns + add < m ? ns + add : m;
        __syncthreads();
    }
    return ns;
}

// canonical row (ascending id, sentinel padded) of the ids in L.kept[0..ns)
__device__ __forceinline__ void store_canonical(uint32_t *row, uint32_t stride, const uint32_t *ids_lds, uint32_t ns, int lane, bool coherent) {
    const uint32_t mine = (uint32_t)lane < ns ? ids_lds[lane] : kSentinel;
    uint32_t rank = 0;
    for (uint32_t s = 0; s < ns; ++s) rank += ids_lds[s] < mine ? 1u : 0u;
    __syncthreads();
    for (uint32_t t = (uint32_t)lane; t < stride; t += 64)
        if (t >= ns) { if (coherent) st_row(row + t, kSentinel); else row[t] = kSentinel; }
    if ((uint32_t)lane < ns) { if (coherent) st_row(row + rank, mine); else row[rank] = mine; }
}


// `nc` ids held one per lane (lanes 0..nc-1, nc <= 64) are ranked by their distance to `owner` in Candidate order and pruned to at most
// `maxn` by select_diverse + backfill with the owner as the reference point: the prune of add_bidirectional_link (mutation.rs:1545-1575) and of
// relink_neighbor (mutation.rs:1959-1984, 2012-2040).  Returns the number kept; L.kept[0..kept) holds them in selection order.
// *bad is set when a distance is not a valid score (Candidate::try_new fails: the reference aborts the mutation).
template <uint32_t METRIC, bool FUSED>
__device__ __forceinline__ uint32_t prune_row_dev(const DevIndex &ix, const BuildLds &L, uint32_t owner, uint32_t v, uint32_t nc, uint32_t maxn, int lane,
                                                  bool *bad) {
    const int grp = lane >> 3, j = lane & 7;
    stage_row(ix, L.qv, owner, lane);
    const float ohdr = ix.hdr[owner];
    __syncthreads();
    if ((uint32_t)lane < nc) L.kept[lane] = v; // scratch: unsorted candidate ids
    __syncthreads();
    for (uint32_t p0 = 0; p0 < nc; p0 += 8) {
        const uint32_t g = p0 + (uint32_t)grp;
        const uint32_t other = L.kept[g < nc ? g : nc - 1u];
        const float d = group_distance<METRIC, FUSED>(ix, L.qv, ohdr, other, j);
        if (g < nc && j == 0) L.dtmp[g] = d;
    }
    __syncthreads();
    const float dmine = (uint32_t)lane < nc ? L.dtmp[lane] : 0.f;
    float dchk = dmine;
    if (__ballot((uint32_t)lane < nc && !score_valid(dchk))) *bad = true;
    uint32_t rank = 0;
    for (uint32_t t = 0; t < nc; ++t) { // Candidate order: score, then id (model.rs:55-61)
        const float dt = L.dtmp[t];
        const uint32_t it = L.kept[t];
        rank += (dt < dmine || (dt == dmine && it < v)) ? 1u : 0u;
    }
    __syncthreads();
    if ((uint32_t)lane < nc) { L.cid[rank] = v; L.csc[rank] = dmine; }
    __syncthreads();
    return select_diverse_dev<METRIC, FUSED>(ix, L, nc, maxn, lane);
}

} // namespace hvx
