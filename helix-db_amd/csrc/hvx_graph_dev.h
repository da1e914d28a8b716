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


// distance between two RESIDENT rows, one 8-lane group, in the reference's summation order.  f32 images: the first row is the "query"
// of group_distance, read in place.  bf16 images (round 6; dim % 64 == 0, AVX+FMA tree, L2 / cosine -- what the import accepts): both rows
// are decoded from the interleaved layout piece by piece -- the values are exact f32 numbers and the arithmetic and its order are those
// of group_distance_bf16 (hvx_device.h), so the distance equals the reference's on the rounded vectors bit for bit.
template <uint32_t METRIC, bool FUSED, bool BF>
__device__ __forceinline__ float pair_distance(const DevIndex &ix, uint32_t ni, uint32_t nj, int j) {
    if constexpr (!BF) {
        return group_distance<METRIC, FUSED>(ix, ix.vec + (size_t)ni * ix.ld, ix.hdr[ni], nj, j);
    } else {
        const int slot = chunk_slot(j);
        const uint16_t *qb = ix.vecb + (size_t)ni * ix.dim, *rb = ix.vecb + (size_t)nj * ix.dim;
        const float4 *qp = reinterpret_cast<const float4 *>(qb) + slot; // one 16-byte piece = this lane's virtual lanes of TWO chunks
        const float4 *rp = reinterpret_cast<const float4 *>(rb) + slot;
        const uint32_t np = ix.dim >> 6;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        auto step = [&](const float4 qw, const float4 xw) __attribute__((always_inline)) {
            const uint32_t q0 = __float_as_uint(qw.x), q1 = __float_as_uint(qw.y), q2 = __float_as_uint(qw.z), q3 = __float_as_uint(qw.w);
            const uint32_t w0 = __float_as_uint(xw.x), w1 = __float_as_uint(xw.y), w2 = __float_as_uint(xw.z), w3 = __float_as_uint(xw.w);
            const float qa0 = __uint_as_float(q0 << 16), qa1 = __uint_as_float(q0 & 0xFFFF0000u), qa2 = __uint_as_float(q1 << 16), qa3 = __uint_as_float(q1 & 0xFFFF0000u);
            const float qb0 = __uint_as_float(q2 << 16), qb1 = __uint_as_float(q2 & 0xFFFF0000u), qb2 = __uint_as_float(q3 << 16), qb3 = __uint_as_float(q3 & 0xFFFF0000u);
            const float a0 = __uint_as_float(w0 << 16), a1 = __uint_as_float(w0 & 0xFFFF0000u), a2 = __uint_as_float(w1 << 16), a3 = __uint_as_float(w1 & 0xFFFF0000u);
            const float b0 = __uint_as_float(w2 << 16), b1 = __uint_as_float(w2 & 0xFFFF0000u), b2 = __uint_as_float(w3 << 16), b3 = __uint_as_float(w3 & 0xFFFF0000u);
            if (METRIC == kL2) {
                float d0 = qa0 - a0, d1 = qa1 - a1, d2 = qa2 - a2, d3 = qa3 - a3;
                acc.x = __builtin_fmaf(d0, d0, acc.x); acc.y = __builtin_fmaf(d1, d1, acc.y);
                acc.z = __builtin_fmaf(d2, d2, acc.z); acc.w = __builtin_fmaf(d3, d3, acc.w);
                d0 = qb0 - b0; d1 = qb1 - b1; d2 = qb2 - b2; d3 = qb3 - b3;
                acc.x = __builtin_fmaf(d0, d0, acc.x); acc.y = __builtin_fmaf(d1, d1, acc.y);
                acc.z = __builtin_fmaf(d2, d2, acc.z); acc.w = __builtin_fmaf(d3, d3, acc.w);
            } else {
                acc.x = __builtin_fmaf(qa0, a0, acc.x); acc.y = __builtin_fmaf(qa1, a1, acc.y);
                acc.z = __builtin_fmaf(qa2, a2, acc.z); acc.w = __builtin_fmaf(qa3, a3, acc.w);
                acc.x = __builtin_fmaf(qb0, b0, acc.x); acc.y = __builtin_fmaf(qb1, b1, acc.y);
                acc.z = __builtin_fmaf(qb2, b2, acc.z); acc.w = __builtin_fmaf(qb3, b3, acc.w);
            }
        };
        uint32_t m = 0;
        for (; m + 6u <= np; m += 6u) { // twelve independent 16-byte loads in flight per lane before the first use
            float4 q[6], x[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) { q[u] = qp[(size_t)(m + u) * 8]; x[u] = rp[(size_t)(m + u) * 8]; }
#pragma unroll
            for (int u = 0; u < 6; ++u) step(q[u], x[u]);
        }
        for (; m < np; ++m) step(qp[(size_t)m * 8], rp[(size_t)m * 8]);
        float r = avx_tree_reduce(acc);
        if (METRIC == kCosine)
            r = cosine_finish_fn(r, ix.hdr[ni], ix.hdr[nj], [&]() {
                return stable_half_cosine_fn(ix.dim, [&](uint32_t i) { return bf16_to_f32(qb[bf16_slot_of(i)]); }, [&](uint32_t i) { return bf16_to_f32(rb[bf16_slot_of(i)]); });
            });
        return r;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Eager prunes (round 6): a prune's distance matrix is evaluated up front by many workgroups (one 8-lane group per pair), written to HBM
// with device-scope stores, and ONE wavefront of the last workgroup to deliver replays select_diverse + backfill from registers.  Shared
// by the delete steps (hvx_delete.hip) and the sequential insert's select / link steps (hvx_build.hip).
// ---------------------------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void st_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every global store of this wavefront has been performed.  A workgroup-scope release fence does NOT wait for them (all wavefronts of a
// workgroup sit behind one vector cache: the compiler omits vmcnt(0) outside threadgroup-split mode), and an agent-scope one writes back
// and invalidates the XCD's L2 -- so the "last workgroup" tickets wait explicitly before the barrier in front of the ticket
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); } // one wavefront's LDS writes before its next reads
struct WaveScratch { uint32_t *ord, *kept; }; // [64] each, per wavefront

// pair q of a prune over nc ids + the owner (index nc): q = i (i - 1) / 2 + j, 0 <= j < i <= nc
__device__ __forceinline__ void pair_of(uint32_t q, uint32_t &i, uint32_t &j) {
    i = (uint32_t)((1.0f + sqrtf(1.0f + 8.0f * (float)q)) * 0.5f);
    while (i * (i - 1u) / 2u > q) --i;
    while ((i + 1u) * i / 2u <= q) ++i;
    j = q - i * (i - 1u) / 2u;
}

// select_diverse + backfill (mod.rs:809-856) over a finished matrix, by ONE wavefront, without LDS: lane c reads ROW c of the matrix
// (RW floats, device-scope 8-byte loads all in flight: the matrix was written by other workgroups, and this compute unit's L2 may hold
// the previous step's lines) and its distance to the owner; ids[c] = cur[c].  -> W.kept[0..return).
//   * Candidate order (score, then id: model.rs:55-61) by lane broadcasts of a 64-bit key (valid scores are non-negative: their bit
//     patterns order like the numbers; an invalid one aborts the delete as the reference does);
//   * who would reject whom is decided for all pairs up front from the registers: bit s of conf(c) <=> candidate s is strictly closer to
//     c than the owner is; the sequential pass over the ranking is then scalar work on two 64-bit masks -- nothing in the dependent
//     chain touches memory (first builds: an LDS gather + ballot per candidate, 6 - 7 us per prune of 64 ids, 3 us per reciprocal one).
// Same decisions as testing each candidate against the selected ones when its turn comes: a candidate is rejected iff an EARLIER selected
// one is strictly closer to it than the owner is (mod.rs:822-842), and the matrix is symmetric bit for bit.
//   * `alive`: bit c clear = id c has left the list since the matrix was evaluated (a sequential insert: an earlier link removed it from this
//     row) -- it ranks behind every live id, rejects nobody and is neither selected nor backfilled; *dropped = the live ids that do not stay.
template <int RW>
__device__ __forceinline__ uint32_t replay_rows(const float *G, const uint32_t *cur, uint32_t nc, uint32_t maxn, uint32_t lane, const WaveScratch &W, bool *bad,
                                                unsigned long long alive = ~0ull, unsigned long long *dropped = nullptr) {
    const unsigned long long *rp = reinterpret_cast<const unsigned long long *>(G + (size_t)(lane < nc ? lane : 0u) * RW);
    unsigned long long x[RW / 2];
#pragma unroll
    for (int u = 0; u < RW / 2; ++u) x[u] = ld_agent(rp + u);
    const float dmine = lane < nc ? ld_agent(G + (size_t)nc * RW + lane) : 0.f;
    const uint32_t v = lane < nc ? cur[lane] : kSentinel;
    alive &= nc >= 64u ? ~0ull : (1ull << nc) - 1ull;
    const bool live = ((alive >> lane) & 1ull) != 0ull;
    const uint32_t nlive = (uint32_t)__builtin_popcountll(alive);
    float chk = dmine;
    if (__ballot(live && !score_valid(chk)) != 0ull) *bad = true;
    const unsigned long long key = live ? ((unsigned long long)__float_as_uint(dmine + 0.0f) << 32) | v : (0xFFFFFFFFull << 32) | lane;
    uint32_t rank = 0;
    for (uint32_t t = 0; t < nc; ++t) {
        const unsigned long long kt = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), (int)t) << 32) |
                                      (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, (int)t);
        rank += kt < key ? 1u : 0u;
    }
    // lane r <- the candidate ranked r (every lane sends its index to the lane of its rank; lanes past nc keep their own place)
    const uint32_t my_ord = (uint32_t)__builtin_amdgcn_ds_permute((int)((lane < nc ? rank : lane) << 2), (int)lane);
    unsigned long long conf = 0ull;
#pragma unroll
    for (int u = 0; u < RW / 2; ++u) {
        const float d0 = __uint_as_float((uint32_t)x[u]), d1 = __uint_as_float((uint32_t)(x[u] >> 32));
        conf |= (unsigned long long)((d0 < dmine ? 1u : 0u) | (d1 < dmine ? 2u : 0u)) << (2 * u);
    }
    conf &= alive & ~(1ull << lane); // (columns past nc and the diagonal hold no distance; an id that has left rejects nobody)
    // ... in rank order (lane r holds the r-th candidate's mask)
    const uint32_t clo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(my_ord << 2), (int)(uint32_t)conf);
    const uint32_t chi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(my_ord << 2), (int)(uint32_t)(conf >> 32));
    unsigned long long selm = 0ull, selr = 0ull; // selected candidates by index / by rank
    uint32_t ns = 0;
    for (uint32_t r = 0; r < nlive && ns < maxn; ++r) { // (the live ids hold the first nlive ranks)
        const unsigned long long cm = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)chi, (int)r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)clo, (int)r);
        if (cm & selm) continue;
        selm |= 1ull << (uint32_t)__builtin_amdgcn_readlane((int)my_ord, (int)r);
        selr |= 1ull << r;
        ++ns;
    }
    // lane r holds the id ranked r; the selected ones go to their slots in selection (= rank) order, then the backfill: the others, closest first
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t rid = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(my_ord << 2), (int)v);
    const bool mine_sel = ((selr >> lane) & 1ull) != 0ull;
    if (mine_sel) W.kept[(uint32_t)__builtin_popcountll(selr & lt)] = rid;
    bool stays = mine_sel;
    if (ns < maxn) { // (mod.rs:845-854)
        const bool free = lane < nlive && !mine_sel;
        const unsigned long long fm = __ballot(free);
        const uint32_t rk = (uint32_t)__builtin_popcountll(fm & lt);
        if (free && ns + rk < maxn) { W.kept[ns + rk] = rid; stays = true; }
        const uint32_t add = (uint32_t)__builtin_popcountll(fm);
        ns = ns + add < maxn ? ns + add : maxn;
    }
    if (dropped) { // by list position (lane r holds the id ranked r = position my_ord): few -- a row over its limit by one loses one
        unsigned long long dr = __ballot(lane < nlive && !stays), dp = 0ull;
        while (dr) {
            const uint32_t r = (uint32_t)__builtin_ctzll(dr);
            dr &= dr - 1ull;
            dp |= 1ull << (uint32_t)__builtin_amdgcn_readlane((int)my_ord, (int)r);
        }
        *dropped = dp;
    }
    lds_order();
    return ns;
}

// a staged row (mutation.rs:1299: sorted by id) from ids held one per lane (lane < ns), ranks by lane broadcasts
__device__ __forceinline__ void store_canonical_reg(uint32_t *row, uint32_t stride, uint32_t mine, uint32_t ns, uint32_t lane) {
    uint32_t rank = 0;
    for (uint32_t s = 0; s < ns; ++s) rank += (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)s) < mine ? 1u : 0u;
    for (uint32_t t = lane; t < stride; t += 64u)
        if (t >= ns) st_row(row + t, kSentinel);
    if (lane < ns) st_row(row + rank, mine);
}


// The same replay for prunes of up to 128 ids (degree limits up to 64: the reference's scale fixture runs M 32 / M0 64,
// scale_contracts.rs:167-173): two candidates per lane (c = lane, lane + 64), 128-bit masks, the matrix rows STREAMED (32 floats at a
// time: only "closer than the owner" bits are kept), Candidate order and the masks in rank order through LDS.  RW = row width of G in
// floats, a multiple of 32.  W2: per-wavefront scratch of kWide2Words uint32 (keys [128] u64 | conf [128][4] | ord [128] | kept [64]).
constexpr uint32_t kWide2Words = 128 * 2 + 128 * 4 + 128 + 64;
template <int RW>
__device__ __forceinline__ uint32_t replay_rows2(const float *G, const uint32_t *cur, uint32_t nc, uint32_t maxn, uint32_t lane, uint32_t *W2, bool *bad) {
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(W2);
    uint32_t *conf = W2 + 256, *ord = W2 + 256 + 512, *kept = W2 + 256 + 512 + 128;
    float d[2];
    uint32_t v[2], rank[2];
    unsigned long long key[2];
    bool isbad = false;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint32_t c = lane + 64u * (uint32_t)s;
        const bool valid = c < nc;
        d[s] = valid ? ld_agent(G + (size_t)nc * RW + c) : 0.f;
        v[s] = valid ? cur[c] : kSentinel;
        float chk = d[s];
        isbad |= valid && !score_valid(chk);
        key[s] = valid ? ((unsigned long long)__float_as_uint(d[s] + 0.0f) << 32) | v[s] : (0xFFFFFFFFull << 32) | c;
        keys[c] = key[s];
        rank[s] = 0;
    }
    if (__ballot(isbad) != 0ull) *bad = true;
    lds_order();
    for (uint32_t t = 0; t < nc; ++t) { // Candidate order: score, then id
        const unsigned long long kt = keys[t];
        rank[0] += kt < key[0] ? 1u : 0u;
        rank[1] += kt < key[1] ? 1u : 0u;
    }
    const unsigned long long alo = nc >= 64u ? ~0ull : (1ull << nc) - 1ull;
    const unsigned long long ahi = nc <= 64u ? 0ull : (nc >= 128u ? ~0ull : (1ull << (nc - 64u)) - 1ull);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint32_t c = lane + 64u * (uint32_t)s;
        if (c >= nc) continue;
        // bit t of conf(c) <=> candidate t is strictly closer to c than the owner is
        unsigned long long lo = 0ull, hi = 0ull;
        const unsigned long long *rp = reinterpret_cast<const unsigned long long *>(G + (size_t)c * RW);
#pragma unroll
        for (int ch = 0; ch < RW / 32; ++ch) {
            unsigned long long x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = ld_agent(rp + ch * 16 + u);
            unsigned long long bits = 0ull;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float d0 = __uint_as_float((uint32_t)x[u]), d1 = __uint_as_float((uint32_t)(x[u] >> 32));
                bits |= (unsigned long long)((d0 < d[s] ? 1u : 0u) | (d1 < d[s] ? 2u : 0u)) << (2 * u);
            }
            // 32 columns: ch * 32 ..
            if (ch == 0) lo |= bits;
            else if (ch == 1) lo |= bits << 32;
            else if (ch == 2) hi |= bits;
            else if (ch == 3) hi |= bits << 32;
        }
        lo &= alo; hi &= ahi;
        if (c < 64u) lo &= ~(1ull << c); else hi &= ~(1ull << (c - 64u));
        const uint32_t r = rank[s];
        conf[r * 4u + 0u] = (uint32_t)lo; conf[r * 4u + 1u] = (uint32_t)(lo >> 32);
        conf[r * 4u + 2u] = (uint32_t)hi; conf[r * 4u + 3u] = (uint32_t)(hi >> 32);
        ord[r] = c;
    }
    lds_order();
    unsigned long long slo = 0ull, shi = 0ull, rlo = 0ull, rhi = 0ull; // selected by index / by rank
    uint32_t ns = 0;
    for (uint32_t r = 0; r < nc && ns < maxn; ++r) {
        const unsigned long long clo = ((unsigned long long)conf[r * 4u + 1u] << 32) | conf[r * 4u + 0u];
        const unsigned long long chi = ((unsigned long long)conf[r * 4u + 3u] << 32) | conf[r * 4u + 2u];
        if ((clo & slo) | (chi & shi)) continue;
        const uint32_t ci = ord[r];
        if (ci < 64u) slo |= 1ull << ci; else shi |= 1ull << (ci - 64u);
        if (r < 64u) rlo |= 1ull << r; else rhi |= 1ull << (r - 64u);
        ++ns;
    }
    // rank r = lane, lane + 64: the selected ones to their slots in selection order, then the backfill: the others, closest first
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t nsel = ns;
    const unsigned long long flo = ~rlo & alo, fhi = ~rhi & ahi; // (the live ids hold the first nc ranks)
    const uint32_t nfree = (uint32_t)__builtin_popcountll(flo) + (uint32_t)__builtin_popcountll(fhi);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint32_t r = lane + 64u * (uint32_t)s;
        if (r >= nc) continue;
        const uint32_t id = cur[ord[r]];
        const bool sel = s == 0 ? ((rlo >> lane) & 1ull) != 0ull : ((rhi >> lane) & 1ull) != 0ull;
        if (sel) {
            const uint32_t slot = s == 0 ? (uint32_t)__builtin_popcountll(rlo & lt) : (uint32_t)__builtin_popcountll(rlo) + (uint32_t)__builtin_popcountll(rhi & lt);
            kept[slot] = id;
        } else if (nsel < maxn) {
            const uint32_t rk = s == 0 ? (uint32_t)__builtin_popcountll(flo & lt) : (uint32_t)__builtin_popcountll(flo) + (uint32_t)__builtin_popcountll(fhi & lt);
            if (nsel + rk < maxn) kept[nsel + rk] = id;
        }
    }
    if (nsel < maxn) ns = nsel + nfree < maxn ? nsel + nfree : maxn;
    lds_order();
    return ns;
}

} // namespace hvx
