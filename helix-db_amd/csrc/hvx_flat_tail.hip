// hvx_flat_tail.hip -- the exact tail of the small-batch matrix-core scan in ONE launch, without a certificate (round 6).
//
// Where it sits: restricted_exact_scan (crates/db/src/search/vector/restricted.rs:753-835) for a handful of queries over 10^4 .. 10^6
// candidate rows (BASELINE config #3's large groups).  hvx_flat_smallb.hip streams the rows once and leaves the APPROXIMATE dot products
// [b][rows] (bf16-rounded operands on the matrix cores).  Rounds 3-5 turned them into the exact answer with three more dependent steps:
// a per-slice radix selection of the m + 1 smallest approximate scores (37 us at 100 000 x 32), a re-rank kernel that selects again,
// re-scores 64 rows per query and checks a certificate (22 us), and the host reading the certificates back -- repeating the whole scan
// with the full hi + lo split when one was missed (always, on tightly clustered rows: the one-pass bound exceeds the gap to the 64th row).
//
// This kernel needs no candidate COUNT and no certificate: it keeps, per query, a device-wide upper bound T of the k-th smallest EXACT
// score (one atomicMin word), and re-scores -- in the reference's summation order, on the reference's f32 / bf16 values -- every row whose
// approximate score could be below it:  s~ - E <= T,  E = the contraction's worst-case error (the bound the certificate used).
//   * a row of the true top-k has  s <= (k-th smallest exact score) <= T  and  s~ - E <= s : it passes the filter and is re-scored.  Every
//     other row either fails the filter or loses in the exact list.  No approximate number ever reaches the result: exact by construction,
//     for ANY error bound -- a loose bound (one rounded pass over bf16 shadows) only re-scores more rows;
//   * T starts at +inf.  A workgroup = (query, slice of 4 096 scores) bounds the k-th smallest exact score of its slice WITHOUT touching a
//     row -- k rows with approximate score <= t have exact score <= t + E; t = the k-th smallest of a wavefront's 64 lane minima (one
//     shuffle sort) -- publishes t + E (atomicMin), reads back the minimum over all workgroups so far, filters its 4 096 scores once and
//     re-scores what passes: ONE dependent round of row gathers.  (A first build re-scored the slice's k best rows to publish an exact
//     bound before filtering: two dependent gather rounds and ~100 list insertions per wavefront -- 72 us per workgroup alone, 200 us
//     at 100 000 x 32 with four workgroups per CU, profiles/r06c_tail_timeline.txt.)  On the clustered config-#3 rows what passes is
//     the query's own cluster (~100 of 100 000 rows), not 64 x 25 candidates;
//   * the slices' exact lists meet like the one-launch scan's (hvx_restricted_exact.hip): the last workgroup of a query merges them and
//     writes ids / scores / counts / status.  One launch behind the contraction, nothing for the host to read back or repeat.
// Algorithmic bytes of the pair (contraction + this kernel) = rows x dim x sizeof(row element) + 4 b rows (the score matrix, twice).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "hvx_toplist.h"
#include "hvx_flat_mfma.h"
#include "hvx_host.h"

namespace hvx {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ T ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void st_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int kTailR = 16;                  // scores per thread
constexpr int kTailSlice = kTailR * 256;    // 4 096 scores per workgroup

struct TailArgs {
    DevIndex ix;
    const float *queries;     // [b][dim] f32
    const uint32_t *qstatus;  // [b]
    const float *qhdr;        // [b] cosine norm header
    const float *qn2;         // [b] |q|^2
    const float *dots;        // [b][chunk_ld] approximate dot products
    uint32_t chunk_ld, rows;
    const uint32_t *subset;   // scan position -> row, or NULL
    const float *rowterm;     // [n] |x|^2 (L2) / |x| (cosine)
    float xmax2, erel;        // max |x|^2; relative error bound of the contraction (absolute for cosine)
    uint32_t b, k, slices;
    uint32_t *thr;            // [b] bits of the smallest k-th exact score published so far (+inf between launches)
    float *part_sc;           // [b][slices][k]
    uint32_t *part_row;
    uint32_t *bad, *done;     // [b] (zero between launches)
    uint64_t *out_ids;        // [b][k]
    float *out_scores;
    uint32_t *out_counts, *out_status;
};


struct Acc { f2 lo, hi; };
constexpr int tail_load_group(int nl) {
    for (int g = 16; g > 1; --g)
        if (nl % g == 0) return g;
    return 1;
}
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
__device__ __noinline__ float tail_slow_half_cosine(const DevIndex &ix, const float *qv, uint32_t node) {
    if (ix.dtype == HVX_BF16) {
        const uint16_t *rb = ix.vecb + (size_t)node * ix.dim;
        return stable_half_cosine_fn(ix.dim, [&](uint32_t i) { return qv[i]; }, [&](uint32_t i) { return bf16_to_f32(rb[bf16_slot_of(i)]); });
    }
    const float *rf = ix.vec + (size_t)node * ix.ld;
    return stable_half_cosine_fn(ix.dim, [&](uint32_t i) { return qv[i]; }, [&](uint32_t i) { return rf[i]; });
}

// one row per 8-lane group in the reference's order (AVX+FMA tree over f32 rows, or bf16 rows in the interleaved layout): pieces of 16
// bytes per lane requested NG at a time -- the register budget that keeps three workgroups per CU resident
template <uint32_t METRIC, int NK, bool BF>
__device__ __forceinline__ float exact_row(const DevIndex &ix, const float *qs, float qhdr, uint32_t node, int slot) {
    constexpr int NL = BF ? NK / 2 : NK;
    constexpr int NG = tail_load_group(NL); // the largest divisor of NL that is <= 16
    static_assert(NL % NG == 0, "a row is a whole number of load groups");
    const float hdr = METRIC == kCosine ? ix.hdr[node] : 0.f;
    const float4 *rp = BF ? reinterpret_cast<const float4 *>(ix.vecb + (size_t)node * ix.dim) + slot
                          : reinterpret_cast<const float4 *>(ix.vec + (size_t)node * ix.ld) + slot;
    const float4 *qp = reinterpret_cast<const float4 *>(qs) + slot;
    Acc acc;
    acc.lo = f2{0.f, 0.f};
    acc.hi = f2{0.f, 0.f};
#pragma unroll 1
    for (int s = 0; s < NL / NG; ++s) {
        float4 x[NG];
#pragma unroll
        for (int u = 0; u < NG; ++u) x[u] = rp[u * 8];
        rp += NG * 8;
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            if (!BF) {
                fma_chunk_pk<METRIC>(acc, qp[u * 8], x[u]);
            } else {
                const uint32_t w0 = __float_as_uint(x[u].x), w1 = __float_as_uint(x[u].y), w2 = __float_as_uint(x[u].z), w3 = __float_as_uint(x[u].w);
                fma_chunk_pk<METRIC>(acc, qp[(2 * u) * 8], make_float4(__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xFFFF0000u),
                                                                      __uint_as_float(w1 << 16), __uint_as_float(w1 & 0xFFFF0000u)));
                fma_chunk_pk<METRIC>(acc, qp[(2 * u + 1) * 8], make_float4(__uint_as_float(w2 << 16), __uint_as_float(w2 & 0xFFFF0000u),
                                                                          __uint_as_float(w3 << 16), __uint_as_float(w3 & 0xFFFF0000u)));
            }
        }
        qp += (BF ? 2 : 1) * NG * 8;
    }
    float r = avx_tree_reduce(make_float4(acc.lo.x, acc.lo.y, acc.hi.x, acc.hi.y));
    if (METRIC == kCosine) r = cosine_finish_fn(r, qhdr, hdr, [&]() { return tail_slow_half_cosine(ix, qs, node); });
    return r;
}

template <uint32_t METRIC, int NK, bool BF>
__global__ __launch_bounds__(256) void flat_exact_tail_kernel(TailArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float m_sc[5][64];
    __shared__ uint32_t m_id[5][64];
    __shared__ uint32_t s_cnt, s_T, s_last;
    const DevIndex &ix = a.ix;
    const uint32_t q = blockIdx.y, sl = blockIdx.x, k = a.k;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 3, j = lane & 7, slot = chunk_slot(j);
    const uint32_t G = (uint32_t)(wave * 8 + grp);
    float *qs = reinterpret_cast<float *>(smem);                              // [ld] the query, for the exact scores
    uint32_t *queue = reinterpret_cast<uint32_t *>(qs + (size_t)NK * 32);      // [4 096] rows to re-score
    const float inf = __uint_as_float(0x7F800000u);
    const uint32_t st = a.qstatus[q];
    const uint32_t lo = sl * (uint32_t)kTailSlice, hi = lo + (uint32_t)kTailSlice < a.rows ? lo + (uint32_t)kTailSlice : a.rows;
    bool bad = false;
    TopList ex; // this wavefront's exact list
    ex.init();
    if (st == 0u && lo < hi) { // (block-uniform)
        const float qhdr = METRIC == kCosine ? a.qhdr[q] : 0.f;
        for (uint32_t i = (uint32_t)tid; i < (uint32_t)NK * 8u; i += 256)
            reinterpret_cast<float4 *>(qs)[i] = reinterpret_cast<const float4 *>(a.queries + (size_t)q * ix.dim)[i];
        if (tid == 0) s_cnt = 0;
        // ---- the slice's approximate scores ----
        const float *dq = a.dots + (size_t)q * a.chunk_ld;
        const float qn2 = a.qn2[q], qn = sqrtf(qn2);
        // worst-case |approximate - reference-order score| (hvx_flat_mfma.hip, rerank_bf16_kernel: the certificate's bound)
        const float E = METRIC == kL2 ? a.erel * 0.5f * (qn2 + a.xmax2) : a.erel;
        float as[kTailR];
        uint32_t an[kTailR];
#pragma unroll
        for (int r = 0; r < kTailR; ++r) {
            const uint32_t i = lo + (uint32_t)r * 256u + (uint32_t)tid;
            float s = inf;
            uint32_t node = kSentinel;
            if (i < hi) {
                node = a.subset ? a.subset[i] : i;
                const float dot = dq[i], term = a.rowterm[node];
                if (METRIC == kL2) {
                    s = (qn2 + term) - 2.0f * dot;
                    s = s < 0.f ? 0.f : s;
                } else {
                    const float den = qn * term;
                    float c = dot / den;
                    c = c < -1.f ? -1.f : (c > 1.f ? 1.f : c);
                    s = (1.0f - c) * 0.5f;
                    if (!(den > 0.f) || !f32_is_normal(den)) s = -inf; // extreme norms: the reference's f64 fallback decides -- always re-scored
                }
                if (!f32_is_finite(s) || !f32_is_finite(dot)) s = -inf; // never trust a non-finite approximation: the row is re-scored
            }
            as[r] = s;
            an[r] = node;
        }
        // ---- an upper bound of the slice's k-th smallest EXACT score, without touching a row: k rows with approximate score <= t have exact
        //      score <= t + E.  t = the k-th smallest of a wavefront's 64 lane minima (64 different rows: at least k of them are <= it), the
        //      smallest such t over the four wavefronts.  Forced rows have no usable approximation and stay out of it. ----
        float lmin = inf;
#pragma unroll
        for (int r = 0; r < kTailR; ++r) lmin = (an[r] != kSentinel && as[r] != -inf && as[r] < lmin) ? as[r] : lmin;
        {
            float v = lmin; // ascending bitonic sort of the lane minima across the wavefront (21 exchange steps, no LDS)
#pragma unroll
            for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    const float o = __shfl_xor(v, stride, 64);
                    const bool lower = (lane & stride) == 0, up = (lane & size) == 0;
                    const float mn = o < v ? o : v, mx = o < v ? v : o;
                    v = (lower == up) ? mn : mx;
                }
            const float kth = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), (int)(k - 1u)));
            if (lane == 0) m_sc[4][wave] = kth;
        }
        __syncthreads();
        if (tid == 0) {
            float t = m_sc[4][0];
            for (int w = 1; w < 4; ++w) t = m_sc[4][w] < t ? m_sc[4][w] : t;
            const float U = t + E; // (+inf stays +inf)
            const uint32_t mine = f32_is_finite(U) && U >= 0.f ? __float_as_uint(U) : 0x7F800000u;
            const uint32_t old = atomicMin(&a.thr[q], mine); // (bounds are >= +0: their bit patterns order like the values)
            s_T = old < mine ? old : mine;
        }
        __syncthreads();
        // ---- filter: every row whose exact score could be <= T; re-scored in the reference's order ----
        const float T = __uint_as_float(s_T);
#pragma unroll
        for (int r = 0; r < kTailR; ++r) {
            if (an[r] != kSentinel && as[r] - E <= T) queue[atomicAdd(&s_cnt, 1u)] = an[r];
        }
        __syncthreads();
        const uint32_t nq = s_cnt;
        for (uint32_t f0 = 0; f0 < nq; f0 += 32u) {
            const uint32_t f = f0 + G;
            const uint32_t node = queue[f < nq ? f : 0u];
            float d = exact_row<METRIC, NK, BF>(ix, qs, qhdr, node, slot);
            const bool live = f < nq;
            const bool valid = score_valid(d);
            if (live && !valid) bad = true;
            ex.offer(live && valid && j == 0 && ex.admits(d, node), d, node, k, lane);
        }
    }
    // ---- the workgroup's exact list -> HBM; the query's last workgroup merges the slices ----
    if (__ballot(bad) && lane == 0) atomicOr(&a.bad[q], 1u);
    __syncthreads();
    m_sc[wave][lane] = ex.sc;
    m_id[wave][lane] = ex.id;
    __syncthreads();
    if (wave == 0) {
        TopList l;
        l.init();
        for (int w = 0; w < 4; ++w) {
            const float es = m_sc[w][lane];
            const uint32_t ei = m_id[w][lane];
            l.offer(ei != kSentinel, es, ei, k, lane);
        }
        if ((uint32_t)lane < k) {
            const size_t at = ((size_t)q * a.slices + sl) * k + (uint32_t)lane;
            st_agent(a.part_sc + at, l.sc);
            st_agent(a.part_row + at, l.id);
        }
    }
    // No device-scope fence here: an agent-scope release / acquire pair is an L2 write-back and an L2 INVALIDATE on this part (one L2 per
    // XCD), and hundreds of short workgroups doing that to the L2 their neighbours are streaming rows through cost more than the scan
    // (first build: 200 us at 100 000 x 32).  The lists are written with device-scope (write-through) stores and read with device-scope
    // loads; the stores have been acknowledged (vscnt = 0, workgroup-scope release) before the ticket is taken.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (a workgroup-scope release fence does not wait for global stores outside threadgroup-split mode)
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(&a.done[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == a.slices ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    const uint32_t per_query = a.slices * k;
    TopList l;
    l.init();
    {
        const float *ps = a.part_sc + (size_t)q * per_query;
        const uint32_t *pr = a.part_row + (size_t)q * per_query;
        const uint32_t share = (per_query + 3u) / 4u, e0 = (uint32_t)wave * share, e1 = e0 + share < per_query ? e0 + share : per_query;
        for (uint32_t e = e0; e < e1; e += 256u) {
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
            for (int u = 0; u < 4; ++u) l.offer(ei[u] != kSentinel && l.admits(es[u], ei[u]), es[u], ei[u], k, lane);
        }
    }
    __syncthreads();
    m_sc[wave][lane] = l.sc;
    m_id[wave][lane] = l.id;
    __syncthreads();
    if (wave != 0) return;
    l.init();
    for (int w = 0; w < 4; ++w) {
        const float es = m_sc[w][lane];
        const uint32_t ei = m_id[w][lane];
        l.offer(ei != kSentinel, es, ei, k, lane);
    }
    uint32_t isbad = 0;
    if (lane == 0) { isbad = ld_agent(a.bad + q); if (isbad) st_agent(a.bad + q, 0u); }
    isbad = __builtin_amdgcn_readfirstlane(isbad);
    uint32_t outn = l.count < k ? l.count : k;
    if (st != 0u || isbad) outn = 0; // a rejected query keeps its status; an invalid score is Candidate::try_new failing (model.rs:21-29)
    if ((uint32_t)lane < outn) {
        a.out_ids[(size_t)q * k + (uint32_t)lane] = ix.ids[l.id];
        a.out_scores[(size_t)q * k + (uint32_t)lane] = l.sc;
    }
    if (lane == 0) {
        a.out_counts[q] = outn;
        if (a.out_status) a.out_status[q] = st != 0u ? st : (isbad ? 8u /*HVX_ERR_INVARIANT*/ : 0u);
        st_agent(a.done + q, 0u);
        st_agent(a.thr + q, 0x7F800000u);
    }
}

__global__ void tail_fill_kernel(uint32_t *p, uint32_t n, uint32_t v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

template <uint32_t METRIC, bool BF> hipError_t launch_tail_nk(const TailArgs &a, hipStream_t s) {
    const dim3 grid(a.slices, a.b);
    const size_t lds = (size_t)a.ix.dim * 4 + (size_t)kTailSlice * 4;
    switch (a.ix.dim >> 5) {
#define HVX_TAIL(N) case N: hipLaunchKernelGGL((flat_exact_tail_kernel<METRIC, N, BF>), grid, dim3(256), lds, s, a); break;
        HVX_TAIL(4) HVX_TAIL(8) HVX_TAIL(16) HVX_TAIL(24) HVX_TAIL(32) HVX_TAIL(48)
#undef HVX_TAIL
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace

bool flat_tail_supported(const hvx_index *ix, uint32_t b, uint32_t k, uint32_t n_rows) {
    const DevIndex &d = ix->dev;
    const uint32_t nk = d.dim >> 5;
    if (k == 0 || k > 64u || n_rows == 0 || d.dim % 32u != 0u) return false;
    if (!(nk == 4 || nk == 8 || nk == 16 || nk == 24 || nk == 32 || nk == 48)) return false;
    if (d.metric == kL1 || d.fkernel != kKernelAvxFma) return false;
    if (d.dtype == HVX_F32) return d.ld == d.dim && d.dim_main == d.dim;
    return d.dtype == HVX_BF16;
}

// `dots` = the approximate dot products [b][chunk_ld] of `rows` scan positions (hvx_flat_smallb.hip); erel as RerankArgs::extra_rel + the
// split's own bound.  Writes the exact top-k of every query; nothing is read back.
int flat_tail_enqueue(hvx_index *ix, const float *d_queries, uint32_t b, uint32_t k, const float *dots, uint32_t chunk_ld, uint32_t rows,
                      const uint32_t *d_subset, const float *rowterm, const float *qn2, float erel, uint64_t *d_ids, float *d_scores,
                      uint32_t *d_counts, uint32_t *d_status) {
    const DevIndex &d = ix->dev;
    const uint32_t slices = (rows + (uint32_t)kTailSlice - 1u) / (uint32_t)kTailSlice;
    int rc;
    const size_t need = (size_t)b * slices * k;
    if (need > ix->cap_t_part) {
        if ((rc = ix->regrow((void **)&ix->t_part_sc, need * 4))) return rc;
        if ((rc = ix->regrow((void **)&ix->t_part_row, need * 4))) return rc;
        ix->cap_t_part = need;
    }
    if (b > ix->cap_t_ctl) {
        const uint32_t cap = std::max<uint32_t>(b, 256u);
        if ((rc = ix->regrow((void **)&ix->t_ctl, (size_t)cap * 3 * 4))) return rc;
        if (hipMemsetAsync(ix->t_ctl, 0, (size_t)cap * 2 * 4, ix->stream) != hipSuccess) return fail(HVX_ERR_DEVICE, "clearing the tail kernel's counters failed");
        hipLaunchKernelGGL(tail_fill_kernel, dim3((cap + 255u) / 256u), dim3(256), 0, ix->stream, ix->t_ctl + (size_t)cap * 2, cap, 0x7F800000u);
        ix->cap_t_ctl = cap;
    }
    TailArgs a{};
    a.ix = d;
    a.queries = d_queries; a.qstatus = ix->d_qstatus; a.qhdr = ix->d_qhdr; a.qn2 = qn2;
    a.dots = dots; a.chunk_ld = chunk_ld; a.rows = rows; a.subset = d_subset; a.rowterm = rowterm;
    a.xmax2 = ix->m_xmax2;
    // worst-case |approximate - reference-order score| relative to (|q|^2 + |x|^2) / 2 (L2), absolute (cosine): the certificate's bound
    // (hvx_flat_mfma.hip rerank_bf16_kernel) -- split residuals 2e-5 + f32 accumulation in both orders 12 K 2^-24 + what a one-pass
    // contraction dropped
    a.erel = 2.0e-5f + 12.0f * (float)d.dim * 5.9604645e-8f + erel;
    a.b = b; a.k = k; a.slices = slices;
    a.bad = ix->t_ctl; a.done = ix->t_ctl + ix->cap_t_ctl; a.thr = ix->t_ctl + (size_t)ix->cap_t_ctl * 2;
    a.part_sc = ix->t_part_sc; a.part_row = ix->t_part_row;
    a.out_ids = d_ids; a.out_scores = d_scores; a.out_counts = d_counts; a.out_status = d_status;
    hipError_t e;
    const bool bf = d.dtype == HVX_BF16;
    if (d.metric == kL2) e = bf ? launch_tail_nk<kL2, true>(a, ix->stream) : launch_tail_nk<kL2, false>(a, ix->stream);
    else e = bf ? launch_tail_nk<kCosine, true>(a, ix->stream) : launch_tail_nk<kCosine, false>(a, ix->stream);
    if (e != hipSuccess) return fail(HVX_ERR_DEVICE, "exact tail launch: %s", hipGetErrorString(e));
    return HVX_OK;
}

} // namespace hvx
