// hvx_restricted_exact.hip -- the restricted (prefiltered) exact scan as ONE launch (round 6; SURVEY.md rows a10 / a12, BASELINE config #3).
//
// Reference: restricted_exact_scan + restricted_score_keys (crates/db/src/search/vector/restricted.rs:753-835, :661-704): score every
// candidate with D::distance, keep the k smallest by Candidate order (score, then id: model.rs:55-61), return them sorted; candidate
// ids that hold no vector are skipped (:615-659); RestrictedVectorCandidates is a SET (:303-371: duplicates count once).  The
// operator above it serves one query with ITS OWN candidate set per call (execution/interpreter/access/search/storage.rs:140-163).
//
// Why a second exact scan beside hvx_flat*.hip: the matrix-core pipeline (approximate contraction -> selection -> exact re-rank ->
// certificate read back by the host) is built for candidate sets that are streams -- 10^5 rows and more per launch.  The sets a
// where_() filter produces are mostly small (10^2 .. 10^4 ids), there are MANY of them in flight (one per request), and each request
// brings its own: four dependent launches and a host round trip per set left the device idle (0.001 - 0.12 of HBM, one
// set at a time).  This kernel answers a whole batch of (query, candidate set) pairs in one launch, exactly:
//   * grid = (slices, query tiles); a workgroup = 4 wavefronts = 32 row groups of 8 lanes (hvx_device.h); a tile's TQ queries sit in
//     LDS, every row group gathers 2 candidate rows at a time (8 lanes x 16 B = whole 128-byte lines, 8 chunks ahead) and scores them
//     against the TQ queries in the reference's summation order -- the scores ARE the reference's f32 bits, nothing to certify;
//     the FMAs are packed (v_pk_add_f32 / v_pk_fma_f32: two virtual lanes per instruction, same IEEE operations);
//   * per-query candidate sets arrive as EXTERNAL ids (CSR): id -> row (arithmetic for contiguous ids, binary search otherwise), deleted
//     rows and unknown ids drop out, duplicates are rejected where the result list is kept;
//   * every wavefront keeps its k smallest (score, row) pairs per query in a register list (hvx_beam.h: ballot rank + one wave shift);
//     the four lists of a workgroup meet in LDS, the slices' lists in HBM, and the LAST workgroup of a tile to finish (one fetch-add)
//     merges them and writes ids / scores / counts / status -- straight into pinned host rows when the caller passes such.
// Algorithmic bytes per (query tile, candidate) = dim x sizeof(row element); the roofline is HBM for one query per set, the packed
// FMA rate for tiles of several queries over a shared set (8 lane-operations per row chunk and query).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "hvx_toplist.h"
#include "hvx_host.h"

using namespace hvx;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(HVX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

struct DirectArgs {
    DevIndex ix;
    const float *queries;     // [b][dim] (device copy written by stage_validate_kernel)
    const uint32_t *qstatus;  // [b]
    const float *qhdr;        // [b]
    const uint32_t *rows;     // shared candidate set: internal rows (unique; any order) ...
    uint32_t n_rows;
    const uint32_t *n_rows_dev; // ... whose number a kernel earlier on the stream left on the device (then n_rows = the host's bound, the grid's size)
    const uint64_t *ext_ids;  // ... or per-query sets: external ids, query q owns [offsets[q], offsets[q + 1]) ...
    const uint64_t *offsets;
    const uint32_t *lens;     // ... or, with offsets == NULL, [q * ext_stride, + lens[q]) (the batching operator's fixed slots)
    uint32_t ext_stride;
    uint32_t contiguous;
    uint32_t b, k, k_stride;
    uint32_t chunk, slices;   // candidate positions per slice (a multiple of 64), slices per query
    float *part_sc;           // [b][slices][k] every slice's k smallest, padded with (+inf, kSentinel)
    uint32_t *part_row;
    uint32_t *bad;            // [b] a score failed Candidate::try_new (zero between launches)
    uint32_t *done;           // [query tiles] workgroups of the tile that have delivered (zero between launches)
    uint64_t *out_ids;        // [b][k_stride]
    float *out_scores;
    uint32_t *out_counts, *out_status;
};

// external node id -> internal row (ids ascending), kSentinel when the id holds no (live) vector
__device__ __forceinline__ uint32_t row_of_id(const DevIndex &ix, uint64_t id, bool contiguous) {
    uint32_t row;
    if (contiguous) {
        const uint64_t first = ix.ids[0];
        if (!(id >= first && id - first < ix.n)) return kSentinel;
        row = (uint32_t)(id - first);
    } else {
        uint32_t lo = 0, hi = ix.n;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (ix.ids[mid] < id) lo = mid + 1;
            else hi = mid;
        }
        if (!(lo < ix.n && ix.ids[lo] == id)) return kSentinel;
        row = lo;
    }
    if (row_dead(ix, row)) return kSentinel;
    return row;
}

struct Acc { f2 lo, hi; }; // a float4 accumulator as two packed halves
constexpr int load_group(int nl) {
    for (int g = 8; g > 1; --g)
        if (nl % g == 0) return g;
    return 1;
}
// the slices' lists, the flags and the counters cross workgroups that may sit on different XCDs (one L2 each): device-scope accesses
template <typename T> __device__ __forceinline__ T ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void st_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <uint32_t METRIC> __device__ __forceinline__ void fma_chunk_pk(Acc &acc, const float4 qq, const float4 xv) {
    const f2 ql = {qq.x, qq.y}, qh = {qq.z, qq.w}, xl = {xv.x, xv.y}, xh = {xv.z, xv.w};
    if (METRIC == kL2) {
        const f2 d0 = ql - xl, d1 = qh - xh;
        acc.lo = __builtin_elementwise_fma(d0, d0, acc.lo);
        acc.hi = __builtin_elementwise_fma(d1, d1, acc.hi);
    } else {
        acc.lo = __builtin_elementwise_fma(ql, xl, acc.lo);
        acc.hi = __builtin_elementwise_fma(qh, xh, acc.hi);
    }
}

// distance/cosine.rs:39-59: the f64 fallback of the cosine kernel (extreme norms only) -- out of line: it is rare, and inlined once per
// (row, query) pair it made the cosine builds several times the size of the Euclidean ones
__device__ __noinline__ float slow_half_cosine(const DevIndex &ix, const float *qv, uint32_t node) {
    if (ix.dtype == HVX_BF16) {
        const uint16_t *rb = ix.vecb + (size_t)node * ix.dim;
        return stable_half_cosine_fn(ix.dim, [&](uint32_t i) { return qv[i]; }, [&](uint32_t i) { return bf16_to_f32(rb[bf16_slot_of(i)]); });
    }
    const float *rf = ix.vec + (size_t)node * ix.ld;
    return stable_half_cosine_fn(ix.dim, [&](uint32_t i) { return qv[i]; }, [&](uint32_t i) { return rf[i]; });
}


// NK = dim / 32 (unrolled shapes: AVX+FMA tree, dim == ld == dim_main); NK == 0: any dimension / metric / summation tree through
// group_distance (one row per group at a time).  BF: bf16 rows (interleaved layout).  EXT: per-query external id lists (TQ == 1).
template <uint32_t METRIC, int NK, bool BF, int TQ, bool EXT, bool FUSED>
__global__ __launch_bounds__(256) void restricted_direct_kernel(DirectArgs a) {
    constexpr int P = 2;
    constexpr bool GEN = NK == 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float m_sc[TQ][4][64];
    __shared__ uint32_t m_id[TQ][4][64];
    __shared__ uint32_t s_last;
    const DevIndex &ix = a.ix;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 3, j = lane & 7, slot = chunk_slot(j);
    const uint32_t G = (uint32_t)(wave * 8 + grp);
    const uint32_t tile = blockIdx.y, q0 = tile * TQ, slice = blockIdx.x;
    const uint32_t ld = ix.ld;
    float *qs = reinterpret_cast<float *>(smem); // [TQ][ld]
    const float inf = __uint_as_float(0x7F800000u);

    bool qok[TQ];
    float qh[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
        const uint32_t q = q0 + (uint32_t)t;
        qok[t] = q < a.b && a.qstatus[q] == 0u;
        qh[t] = (METRIC == kCosine && qok[t]) ? a.qhdr[q] : 0.f;
        for (uint32_t i = (uint32_t)tid; i < ld; i += 256)
            qs[(size_t)t * ld + i] = (qok[t] && i < ix.dim) ? a.queries[(size_t)q * ix.dim + i] : 0.f;
    }
    uint64_t off = 0;
    uint32_t n = a.n_rows;
    if (EXT) {
        if (a.offsets) {
            off = a.offsets[q0];
            n = (uint32_t)(a.offsets[q0 + 1] - off);
        } else {
            off = (uint64_t)q0 * a.ext_stride;
            n = a.lens[q0];
        }
    }
    __syncthreads();

    uint32_t chunk = a.chunk;
    if (!EXT && a.n_rows_dev) { // the list's real length: the slices divide IT (whole 64-row passes), not the bound the grid was sized for
        n = *a.n_rows_dev < a.n_rows ? *a.n_rows_dev : a.n_rows;
        chunk = ((n + a.slices * 64u - 1u) / (a.slices * 64u)) * 64u;
    }
    TopList top[TQ];
#pragma unroll
    for (int t = 0; t < TQ; ++t) top[t].init();
    uint32_t badmask = 0;
    const uint32_t lo = slice * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (uint32_t pass0 = lo; pass0 < hi; pass0 += 32u * P) {
        uint32_t nd[P];
        bool ok[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t pos = pass0 + (uint32_t)p * 32u + G;
            uint32_t row = kSentinel;
            if (pos < hi) row = EXT ? row_of_id(ix, a.ext_ids[off + pos], a.contiguous != 0u) : a.rows[pos];
            ok[p] = row != kSentinel;
            nd[p] = ok[p] ? row : 0u; // (a group without a row of its own re-reads row 0: no divergence in the gather)
        }
        if (!__ballot(ok[0] || ok[1])) continue;
        float sc[P][TQ];
        if constexpr (GEN) {
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int t = 0; t < TQ; ++t) {
                    float d = 0.f;
                    if (ok[p]) { // group-uniform
                        if (BF) d = group_distance_bf16<METRIC == kL1 ? kL2 : METRIC>(ix, qs + (size_t)t * ld, qh[t], nd[p], j);
                        else d = group_distance<METRIC, FUSED>(ix, qs + (size_t)t * ld, qh[t], nd[p], j);
                    }
                    sc[p][t] = d;
                }
        } else {
            constexpr int NL = BF ? NK / 2 : NK;     // 16-byte pieces per lane and row
            constexpr int NG = load_group(NL);       // pieces requested together: the largest divisor of NL that is <= 8
            constexpr int STAGES = NL / NG;
            static_assert(NL % NG == 0, "a row is a whole number of load groups");
            float hdr[P];
            const float4 *rp[P];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                hdr[p] = METRIC == kCosine ? ix.hdr[nd[p]] : 0.f;
                rp[p] = BF ? reinterpret_cast<const float4 *>(ix.vecb + (size_t)nd[p] * ix.dim) + slot
                           : reinterpret_cast<const float4 *>(ix.vec + (size_t)nd[p] * ld) + slot;
            }
            Acc acc[P][TQ];
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int t = 0; t < TQ; ++t) { acc[p][t].lo = f2{0.f, 0.f}; acc[p][t].hi = f2{0.f, 0.f}; }
            float4 xa[P][NG], xb[P][NG];
            auto request = [&](float4 (&x)[P][NG]) __attribute__((always_inline)) { // the next NG pieces of both rows; the row pointers move on
#pragma unroll
                for (int p = 0; p < P; ++p) {
#pragma unroll
                    for (int u = 0; u < NG; ++u) x[p][u] = rp[p][u * 8];
                    rp[p] += NG * 8;
                }
            };
            const uint32_t ld4 = ld >> 2;
            const float4 *qp = reinterpret_cast<const float4 *>(qs) + slot; // piece 0 of query 0; moves on with every stage consumed
            auto consume = [&](const float4 (&x)[P][NG]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < NG; ++u) {
#pragma unroll
                    for (int t = 0; t < TQ; ++t) {
                        if (!BF) {
                            const float4 qq = qp[(size_t)t * ld4 + u * 8];
#pragma unroll
                            for (int p = 0; p < P; ++p) fma_chunk_pk<METRIC>(acc[p][t], qq, x[p][u]);
                        } else { // piece u = the lane's virtual lanes of chunks 2u, 2u + 1
                            const float4 qa = qp[(size_t)t * ld4 + (2 * u) * 8], qb = qp[(size_t)t * ld4 + (2 * u + 1) * 8];
#pragma unroll
                            for (int p = 0; p < P; ++p) { // bf16 -> f32 is exact: the halfword becomes the high half of the word
                                const uint32_t w0 = __float_as_uint(x[p][u].x), w1 = __float_as_uint(x[p][u].y);
                                const uint32_t w2 = __float_as_uint(x[p][u].z), w3 = __float_as_uint(x[p][u].w);
                                fma_chunk_pk<METRIC>(acc[p][t], qa, make_float4(__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xFFFF0000u),
                                                                                __uint_as_float(w1 << 16), __uint_as_float(w1 & 0xFFFF0000u)));
                                fma_chunk_pk<METRIC>(acc[p][t], qb, make_float4(__uint_as_float(w2 << 16), __uint_as_float(w2 & 0xFFFF0000u),
                                                                                __uint_as_float(w3 << 16), __uint_as_float(w3 & 0xFFFF0000u)));
                            }
                        }
                    }
                }
                qp += (BF ? 2 : 1) * NG * 8;
            };
            request(xa);
#pragma unroll 1
            for (int s = 0; s < STAGES; s += 2) { // stage s + 1 is requested before stage s is multiplied
                if (s + 1 < STAGES) request(xb);
                __builtin_amdgcn_sched_barrier(0);
                consume(xa);
                if (s + 1 < STAGES) {
                    if (s + 2 < STAGES) request(xa);
                    __builtin_amdgcn_sched_barrier(0);
                    consume(xb);
                }
            }
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int t = 0; t < TQ; ++t) {
                    float r = avx_tree_reduce(make_float4(acc[p][t].lo.x, acc[p][t].lo.y, acc[p][t].hi.x, acc[p][t].hi.y));
                    if (METRIC == kCosine) {
                        const uint32_t node = nd[p];
                        const float *qv = qs + (size_t)t * ld;
                        r = cosine_finish_fn(r, qh[t], hdr[p], [&]() { return slow_half_cosine(ix, qv, node); });
                    }
                    sc[p][t] = r;
                }
        }
        // admission: Candidate::try_new per score (model.rs:21-29), then the wavefront's list
#pragma unroll
        for (int t = 0; t < TQ; ++t)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                float d = sc[p][t];
                const bool live = ok[p] && qok[t];
                const bool valid = score_valid(d);
                if (live && !valid) badmask |= 1u << t;
                top[t].offer(live && valid && j == 0 && top[t].admits(d, nd[p]), d, nd[p], a.k, lane);
            }
    }

    // the workgroup's four lists of a query -> one (wavefront t mod 4), padded with (+inf, kSentinel) -> HBM
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
        m_sc[t][wave][lane] = top[t].sc;
        m_id[t][wave][lane] = top[t].id;
        const unsigned long long anybad = __ballot((badmask >> t) & 1u);
        if (anybad && lane == 0 && q0 + (uint32_t)t < a.b) atomicOr(&a.bad[q0 + (uint32_t)t], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
        if ((t & 3) != wave) continue;
        const uint32_t q = q0 + (uint32_t)t;
        if (q >= a.b) continue;
        TopList l;
        l.init();
        for (int w = 0; w < 4; ++w) {
            const float es = m_sc[t][w][lane];
            const uint32_t ei = m_id[t][w][lane];
            l.offer(ei != kSentinel, es, ei, a.k, lane);
        }
        if ((uint32_t)lane < a.k) {
            const size_t at = ((size_t)q * a.slices + slice) * a.k + (uint32_t)lane;
            st_agent(a.part_sc + at, l.sc);
            st_agent(a.part_row + at, l.id);
        }
    }
    // the last workgroup of the tile to get here merges the slices' lists
    // No device-scope fence here: an agent-scope release / acquire pair is an L2 write-back and an L2 INVALIDATE on this part (one L2 per
    // XCD), and hundreds of short workgroups doing that to the L2 their neighbours are streaming rows through cost more than the scan
    // (first build: 200 us at 100 000 x 32).  The lists are written with device-scope (write-through) stores and read with device-scope
    // loads; the stores have been acknowledged (vscnt = 0, workgroup-scope release) before the ticket is taken.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (a workgroup-scope release fence does not wait for global stores outside threadgroup-split mode)
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(&a.done[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == a.slices ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    const uint32_t per_query = a.slices * a.k;
    for (int t0 = 0; t0 < TQ; t0 += (TQ >= 4 ? 4 : 1)) {
        // TQ >= 4: wavefront w merges query t0 + w on its own; fewer queries: the four wavefronts split one query's lists
        const int t = TQ >= 4 ? t0 + wave : t0;
        const uint32_t q = q0 + (uint32_t)t;
        TopList l;
        l.init();
        if (q < a.b) {
            const float *ps = a.part_sc + (size_t)q * per_query;
            const uint32_t *pr = a.part_row + (size_t)q * per_query;
            const uint32_t share = TQ >= 4 ? per_query : (per_query + 3u) / 4u;
            const uint32_t e0 = TQ >= 4 ? 0u : (uint32_t)wave * share, e1 = e0 + share < per_query ? e0 + share : per_query;
            for (uint32_t e = e0; e < e1; e += 256u) { // four coalesced requests in flight
                float es[4];
                uint32_t ei[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t i = e + (uint32_t)u * 64u + (uint32_t)lane;
                    const bool in = i < e1;
                    es[u] = in ? ld_agent(ps + i) : inf;
                    ei[u] = in ? ld_agent(pr + i) : kSentinel;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) l.offer(ei[u] != kSentinel && l.admits(es[u], ei[u]), es[u], ei[u], a.k, lane);
            }
        }
        if (TQ < 4) { // the four partial merges of the one query meet in LDS (wavefront 0 finishes)
            __syncthreads();
            m_sc[0][wave][lane] = l.sc;
            m_id[0][wave][lane] = l.id;
            __syncthreads();
            if (wave != 0) continue;
            l.init();
            for (int w = 0; w < 4; ++w) {
                const float es = m_sc[0][w][lane];
                const uint32_t ei = m_id[0][w][lane];
                l.offer(ei != kSentinel, es, ei, a.k, lane);
            }
        }
        if (q >= a.b) continue;
        // results (restricted.rs:820-835): the k smallest, sorted; a rejected query keeps its status, an invalid score is an invariant error
        uint32_t st = a.qstatus[q];
        if (EXT && n == 0u) st = 0u; // an empty candidate set answers with nothing BEFORE the query is validated (restricted.rs:539-541)
        uint32_t isbad = 0;
        if (lane == 0) { isbad = ld_agent(a.bad + q); if (isbad) st_agent(a.bad + q, 0u); }
        isbad = __builtin_amdgcn_readfirstlane(isbad);
        uint32_t outn = l.count < a.k ? l.count : a.k;
        if (st != 0u || isbad) outn = 0;
        if ((uint32_t)lane < outn) {
            a.out_ids[(size_t)q * a.k_stride + (uint32_t)lane] = ix.ids[l.id];
            a.out_scores[(size_t)q * a.k_stride + (uint32_t)lane] = l.sc;
        }
        if (lane == 0) {
            a.out_counts[q] = outn;
            if (a.out_status) a.out_status[q] = st != 0u ? st : (isbad ? 8u /*HVX_ERR_INVARIANT*/ : 0u);
        }
    }
    if (tid == 0) st_agent(a.done + tile, 0u); // (the next launch starts from zero)
}

// Queries from (pinned) host memory: validation (ValidatedMetricVector::try_new, domain.rs:113-157), the cosine header, and the device
// copy the scan reads -- one kernel instead of a DMA copy + a kernel (a copy engine hop costs more than these few KB are worth)
__global__ __launch_bounds__(256) void stage_queries_kernel(const float *src, float *dst, uint32_t dim, uint32_t b) {
    const uint32_t q = blockIdx.x;
    if (q >= b) return;
    for (uint32_t i = threadIdx.x; i < dim; i += 256) dst[(size_t)q * dim + i] = src[(size_t)q * dim + i];
}

template <typename K> hipError_t launch_direct_kernel(K kern, const DirectArgs &a, uint32_t tiles, size_t lds, hipStream_t s) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(a.slices, tiles), dim3(256), lds, s, a);
    return hipGetLastError();
}

// Queries per tile of a SHARED set: registers bound it (8 accumulator + 4 list registers per query next to the gather's 128) -- the builds
// below are the ones that keep everything in registers (profiles/r06_kernel_meta.json); the any-shape build scores one row at a time: two.
// Every candidate row is then read b / TQ times, from L2 after the first.
constexpr bool tq_built(uint32_t metric, int nk, bool bf, int tq) {
    if (tq <= 1) return true;
    if (nk == 0) return tq <= 2;
    if (metric == kCosine) {
        if (bf) return nk == 32 ? false : tq <= 2;
        return tq <= 4;
    }
    if (bf) return nk <= 24 ? tq <= 8 : (nk <= 48 ? tq <= 4 : tq <= 2);
    return nk <= 24 ? tq <= 8 : tq <= 4;
}
uint32_t direct_tq(const DevIndex &d, uint32_t b, bool unrolled) {
    const int nk = unrolled ? (int)(d.dim >> 5) : 0;
    uint32_t tq = 8;
    while (tq > 1 && (tq > b || !tq_built(d.metric, nk, d.dtype == HVX_BF16, (int)tq))) tq >>= 1;
    return tq;
}

template <uint32_t METRIC, int NK, bool BF, bool FUSED> hipError_t launch_direct_q(const DirectArgs &a, uint32_t tq, bool ext, hipStream_t s) {
    const size_t row = (size_t)a.ix.ld * 4;
    const uint32_t tiles = (a.b + tq - 1) / tq;
    if (ext) return launch_direct_kernel(restricted_direct_kernel<METRIC, NK, BF, 1, true, FUSED>, a, a.b, row, s);
    switch (tq) {
    case 1: return launch_direct_kernel(restricted_direct_kernel<METRIC, NK, BF, 1, false, FUSED>, a, tiles, row, s);
    case 2: if constexpr (tq_built(METRIC, NK, BF, 2)) return launch_direct_kernel(restricted_direct_kernel<METRIC, NK, BF, 2, false, FUSED>, a, tiles, 2 * row, s); else break;
    case 4: if constexpr (tq_built(METRIC, NK, BF, 4)) return launch_direct_kernel(restricted_direct_kernel<METRIC, NK, BF, 4, false, FUSED>, a, tiles, 4 * row, s); else break;
    case 8: if constexpr (tq_built(METRIC, NK, BF, 8)) return launch_direct_kernel(restricted_direct_kernel<METRIC, NK, BF, 8, false, FUSED>, a, tiles, 8 * row, s); else break;
    default: break;
    }
    return hipErrorInvalidValue;
}

template <uint32_t METRIC, bool BF> hipError_t launch_direct_nk(const DirectArgs &a, uint32_t tq, bool ext, bool unrolled, hipStream_t s) {
    if constexpr (METRIC != kL1) {
        if (unrolled) {
            switch (a.ix.dim >> 5) {
            case 4: return launch_direct_q<METRIC, 4, BF, true>(a, tq, ext, s);
            case 8: return launch_direct_q<METRIC, 8, BF, true>(a, tq, ext, s);
            case 12: return launch_direct_q<METRIC, 12, BF, true>(a, tq, ext, s);
            case 16: return launch_direct_q<METRIC, 16, BF, true>(a, tq, ext, s);
            case 24: return launch_direct_q<METRIC, 24, BF, true>(a, tq, ext, s);
            case 32: return launch_direct_q<METRIC, 32, BF, true>(a, tq, ext, s);
            case 48: return launch_direct_q<METRIC, 48, BF, true>(a, tq, ext, s);
            case 64: return launch_direct_q<METRIC, 64, BF, true>(a, tq, ext, s);
            case 96: return launch_direct_q<METRIC, 96, BF, true>(a, tq, ext, s);
            default: break;
            }
        }
    }
    if (BF) return launch_direct_q<METRIC, 0, BF, true>(a, tq, ext, s);
    return kernel_fused(a.ix.fkernel) ? launch_direct_q<METRIC, 0, false, true>(a, tq, ext, s) : launch_direct_q<METRIC, 0, false, false>(a, tq, ext, s);
}

bool direct_unrolled_shape(const DevIndex &d) {
    const uint32_t nk = d.dim >> 5;
    const bool dims = nk == 4 || nk == 8 || nk == 12 || nk == 16 || nk == 24 || nk == 32 || nk == 48 || nk == 64 || nk == 96;
    return d.dim % 32u == 0u && dims && d.fkernel == kKernelAvxFma && d.metric != kL1 && (d.dtype == HVX_BF16 || (d.ld == d.dim && d.dim_main == d.dim));
}

} // namespace

namespace hvx {

bool restricted_direct_supported(const hvx_index *ix, uint32_t k) {
    const DevIndex &d = ix->dev;
    if (d.n == 0 || k == 0 || k > 64u) return false;
    if (d.dtype == HVX_F32) return true;                       // any dimension / metric / summation tree (group_distance)
    if (d.dtype == HVX_BF16) return d.metric != kL1 && d.dim % 64u == 0u; // what the import accepts for bf16 rows
    return false;                                              // fp8 rows: the matrix-core pipeline
}

// scratch of the one-launch scan: the slices' lists and the two self-cleaning counter arrays
static int direct_scratch(hvx_index *ix, uint32_t b, uint32_t slices, uint32_t k, uint32_t tiles) {
    int rc;
    const size_t need = (size_t)b * slices * k;
    if (need > ix->cap_x_part) {
        if ((rc = ix->regrow((void **)&ix->x_part_sc, need * 4))) return rc;
        if ((rc = ix->regrow((void **)&ix->x_part_row, need * 4))) return rc;
        ix->cap_x_part = need;
    }
    const uint32_t words = std::max(b, tiles);
    if (words > ix->cap_x_ctl) {
        const uint32_t cap = std::max<uint32_t>(words, 1024u);
        if ((rc = ix->regrow((void **)&ix->x_bad, (size_t)cap * 4))) return rc;
        if ((rc = ix->regrow((void **)&ix->x_done, (size_t)cap * 4))) return rc;
        HIP_TRY(hipMemsetAsync(ix->x_bad, 0, (size_t)cap * 4, ix->stream));
        HIP_TRY(hipMemsetAsync(ix->x_done, 0, (size_t)cap * 4, ix->stream));
        ix->cap_x_ctl = cap;
    }
    return HVX_OK;
}

// Enqueue the one-launch exact scan for b device-resident, VALIDATED queries (ix->d_qstatus / d_qhdr hold their status and headers).
// Shared set: d_rows / n_rows (internal rows, ascending, unique).  Per-query sets: d_ext_ids + d_offsets[b + 1] (external ids, any
// order, duplicates allowed), max_set = the largest set's length.  Outputs: device-visible arrays (HBM or mapped host rows) of row
// length k_stride.  Nothing is synchronised here.
int restricted_direct_enqueue(hvx_index *ix, const float *d_queries, uint32_t b, uint32_t k, uint32_t k_stride, const uint32_t *d_rows,
                              uint32_t n_rows, const uint64_t *d_ext_ids, const uint64_t *d_offsets, uint32_t max_set, uint64_t *d_ids,
                              float *d_scores, uint32_t *d_counts, uint32_t *d_status, const uint32_t *d_lens, uint32_t ext_stride,
                              const uint32_t *d_n_rows) {
    const DevIndex &d = ix->dev;
    const bool ext = d_ext_ids != nullptr;
    const bool unrolled = direct_unrolled_shape(d);
    const uint32_t n = ext ? max_set : n_rows;
    const uint32_t tq = ext ? 1u : direct_tq(d, b, unrolled);
    const uint32_t tiles = ext ? b : (b + tq - 1) / tq;
    // slices: a whole number of 64-row passes each, enough workgroups to fill the chip four times over, at most 1 024 per query
    const uint32_t passes = std::max<uint32_t>(1u, (n + 63u) / 64u);
    const uint32_t want = std::max<uint32_t>(1u, std::min<uint32_t>(1024u, (1024u + tiles - 1u) / tiles));
    const uint32_t per = (passes + want - 1u) / want;
    const uint32_t chunk = per * 64u, slices = std::max<uint32_t>(1u, (n + chunk - 1u) / chunk);
    int rc = direct_scratch(ix, b, slices, k, tiles);
    if (rc) return rc;
    DirectArgs a{};
    a.ix = d;
    a.queries = d_queries;
    a.qstatus = ix->d_qstatus;
    a.qhdr = ix->d_qhdr;
    a.rows = d_rows;
    a.n_rows = n_rows;
    a.n_rows_dev = d_n_rows;
    a.ext_ids = d_ext_ids;
    a.offsets = d_offsets;
    a.lens = d_lens;
    a.ext_stride = ext_stride;
    a.contiguous = ix->contiguous ? 1u : 0u;
    a.b = b; a.k = k; a.k_stride = k_stride;
    a.chunk = chunk; a.slices = slices;
    a.part_sc = ix->x_part_sc; a.part_row = ix->x_part_row; a.bad = ix->x_bad; a.done = ix->x_done;
    a.out_ids = d_ids; a.out_scores = d_scores; a.out_counts = d_counts; a.out_status = d_status;
    hipError_t e;
    const bool bf = d.dtype == HVX_BF16;
    switch (d.metric) {
    case kCosine: e = bf ? launch_direct_nk<kCosine, true>(a, tq, ext, unrolled, ix->stream) : launch_direct_nk<kCosine, false>(a, tq, ext, unrolled, ix->stream); break;
    case kL2: e = bf ? launch_direct_nk<kL2, true>(a, tq, ext, unrolled, ix->stream) : launch_direct_nk<kL2, false>(a, tq, ext, unrolled, ix->stream); break;
    default: e = launch_direct_nk<kL1, false>(a, tq, ext, false, ix->stream); break;
    }
    if (e != hipSuccess) return fail(HVX_ERR_DEVICE, "restricted exact scan launch: %s", hipGetErrorString(e));
    ix->last_scan_path = HVX_PATH_DIRECT;
    return HVX_OK;
}

// candidate ids of a batch from the batching operator's pinned slots (fixed stride) into device memory: only the ids that are there
__global__ __launch_bounds__(256) void stage_ids_kernel(const uint64_t *src, const uint32_t *lens, uint32_t stride, uint64_t *dst, uint32_t b) {
    const uint32_t q = blockIdx.x;
    if (q >= b) return;
    const uint32_t n = lens[q] < stride ? lens[q] : stride;
    for (uint32_t i = threadIdx.x; i < n; i += 256) dst[(size_t)q * stride + i] = src[(size_t)q * stride + i];
}
hipError_t launch_stage_ids(const uint64_t *src, const uint32_t *lens, uint32_t stride, uint64_t *dst, uint32_t b, hipStream_t s) {
    if (b == 0) return hipSuccess;
    hipLaunchKernelGGL(stage_ids_kernel, dim3(b), dim3(256), 0, s, src, lens, stride, dst, b);
    return hipGetLastError();
}

hipError_t launch_stage_queries(const float *src, float *dst, uint32_t dim, uint32_t b, hipStream_t s) {
    if (b == 0) return hipSuccess;
    hipLaunchKernelGGL(stage_queries_kernel, dim3(b), dim3(256), 0, s, src, dst, dim, b);
    return hipGetLastError();
}

} // namespace hvx

// Device-resident surface of the prefiltered branch: b queries in HBM, query q with its own candidate ids d_allowed_ids[q * stride ..
// + d_lens[q]) (external ids, any order, duplicates allowed, unknown ids skipped); exact strategy, one launch, nothing synchronised.
extern "C" int hvx_search_restricted_lists_device(const hvx_index *cix, const float *d_queries, uint32_t b, uint32_t k, const uint64_t *d_allowed_ids,
                                                  uint32_t stride, const uint32_t *d_lens, uint32_t max_len, uint64_t *d_out_ids, float *d_out_scores,
                                                  uint32_t *d_out_counts, uint32_t *d_out_status) {
    if (!cix || !d_queries || !d_allowed_ids || !d_lens || !d_out_ids || !d_out_scores || !d_out_counts) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    if (k == 0) return fail(HVX_ERR_K_RANGE, "result count must be non-zero");
    if (b == 0) return HVX_OK;
    if (b > ix->max_batch) return fail(HVX_ERR_UNSUPPORTED, "batch %u exceeds max_batch %u given at import", b, ix->max_batch);
    if (max_len > stride) return fail(HVX_ERR_INVARIANT, "a candidate list longer than its slot");
    std::lock_guard<std::mutex> lock(ix->mu);
    ix->sync_rewrites();
    if (!restricted_direct_supported(ix, k))
        return fail(HVX_ERR_UNSUPPORTED, "the one-launch restricted scan serves k <= 64 over f32 / bf16 rows of a non-empty image");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(launch_validate_queries(ix->dev, d_queries, b, ix->limit, ix->d_qstatus, ix->d_qhdr, ix->stream));
    return restricted_direct_enqueue(ix, d_queries, b, k, k, nullptr, 0, d_allowed_ids, nullptr, max_len, d_out_ids, d_out_scores, d_out_counts,
                                     d_out_status, d_lens, stride);
}
