// hvx_flat.hip -- exact scan kernels: query validation, bit-exact distance tiles, exact top-k
// selection, result finish, and the multi-shard top-k merge.
//
// Semantics: restricted_exact_scan + restricted_score_keys
// (crates/db/src/search/vector/restricted.rs:753-835, :661-704): score every candidate row with
// D::distance, keep the k smallest by Candidate order (score, then node id; model.rs:55-61), return
// them sorted.  The flat scan is the same with candidates = all rows.
#include "hvx_device.h"
#include "hvx_kernels.h"

namespace hvx {

// ---------------------------------------------------------------------------------------------
// ValidatedMetricVector::try_new on the device (domain.rs:113-157) + Distance::new_header
// ---------------------------------------------------------------------------------------------
// Cosine header = scaled_l2_norm (distance/cosine.rs:12-36): a running scale (the prefix maximum of |v|) and a sum of squared ratios,
// serial f64 by definition.  Round 5: what is serial is only the CHAIN of additions; the scale in effect at element i is the prefix
// maximum of the magnitudes before it, so the wavefront computes that with a scan, every lane derives the terms of its own block of
// consecutive elements (ratio^2 = (mag / scale)^2 for an ordinary element, the ratio scale / mag for a new maximum: the f64 divisions
// run 64 wide), and the chain `sum += term` / `sum = 1 + sum * ratio * ratio` then walks the terms in element order out of LDS --
// the same operations on the same operands in the same order as the reference's loop, hence the same bits.  (Lane 0 alone walking
// the row with a division per element took 0.15 ms per 1 024-query batch at dim 768: a fifth of a cosine search step.)
constexpr uint32_t kNormLdsDim = 2048; // rows up to this dimension take the parallel form (18 KiB of LDS); longer ones the serial loop
__device__ __forceinline__ double scaled_l2_norm_wave(const float *v, uint32_t dim, int lane, double *terms, unsigned char *rec) {
    const uint32_t c = (dim + 63u) / 64u, i0 = (uint32_t)lane * c, i1 = i0 + c < dim ? i0 + c : dim;
    float bmax = 0.0f;
    for (uint32_t i = i0; i < i1; ++i) bmax = fmaxf(bmax, fabsf(v[i]));
    // exclusive prefix maximum over the lanes (magnitudes are finite and >= 0: their bit patterns order like the values)
    uint32_t run = __float_as_uint(bmax);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)run, off, 64);
        if (lane >= off) run = o > run ? o : run;
    }
    uint32_t before = (uint32_t)__shfl_up((int)run, 1, 64);
    if (lane == 0) before = 0u;
    double scale = (double)__uint_as_float(before);
    for (uint32_t i = i0; i < i1; ++i) {
        const double mag = (double)fabsf(v[i]);
        double t = 0.0;
        unsigned char r = 0;
        if (mag != 0.0) {
            if (scale < mag) { t = scale / mag; r = 1; scale = mag; }
            else { const double ratio = mag / scale; t = ratio * ratio; }
        }
        terms[i] = t;
        rec[i] = r;
    }
    __syncthreads();
    double scaled_sum = 1.0;
    for (uint32_t i = 0; i < dim; ++i) { // every lane walks the same chain (LDS broadcasts)
        const double t = terms[i];
        if (rec[i]) scaled_sum = 1.0 + scaled_sum * t * t;
        else scaled_sum += t; // (a zero element contributes +0.0: the sum is >= 1 and keeps its bits, as `continue` does)
    }
    const double total = (double)__uint_as_float((uint32_t)__shfl((int)run, 63, 64));
    if (total == 0.0) return 0.0;
    return total * sqrt(scaled_sum);
}

__global__ __launch_bounds__(64) void validate_vectors_kernel(DevIndex ix, const float *vectors, uint32_t stride,
                                                               uint32_t b, float limit, uint32_t *status, float *qhdr) {
    __shared__ double terms[kNormLdsDim];
    __shared__ unsigned char rec[kNormLdsDim];
    const uint32_t q = blockIdx.x;
    if (q >= b) return;
    const int lane = lane_id();
    const float *v = vectors + (size_t)q * stride;
    bool nonfinite = false, nonzero = false, over = false;
    for (uint32_t i = (uint32_t)lane; i < ix.dim; i += 64) {
        float x = v[i];
        if (!f32_is_finite(x)) nonfinite = true;
        if (x != 0.0f) nonzero = true;
        if (fabsf(x) > limit) over = true;
    }
    const bool any_nf = __ballot(nonfinite) != 0ull;
    const bool any_nz = __ballot(nonzero) != 0ull;
    const bool any_ov = __ballot(over) != 0ull;
    uint32_t st = 0;
    if (any_nf) st = 2;                                   // HVX_ERR_NONFINITE
    else if (ix.metric == kCosine && !any_nz) st = 3;     // HVX_ERR_ZERO_NORM
    else if (ix.metric != kCosine && any_ov) st = 4;      // HVX_ERR_MAGNITUDE
    float h = 0.0f;
    if (st == 0 && ix.metric == kCosine) { // (uniform)
        double norm;
        if (ix.dim <= kNormLdsDim) norm = scaled_l2_norm_wave(v, ix.dim, lane, terms, rec);
        else norm = scaled_l2_norm(ix.dim, [&](uint32_t i) { return v[i]; });
        const double mx = 3.4028234663852886e+38;
        if (norm > mx) norm = mx;
        h = (float)norm;
    }
    if (lane == 0) {
        status[q] = st;
        if (qhdr) qhdr[q] = h;
    }
}

hipError_t launch_validate_queries(const DevIndex &ix, const float *d_queries, uint32_t b, float limit,
                                   uint32_t *d_status, float *d_qhdr, hipStream_t s) {
    if (b == 0) return hipSuccess;
    hipLaunchKernelGGL(validate_vectors_kernel, dim3(b), dim3(64), 0, s, ix, d_queries, ix.dim, b, limit, d_status, d_qhdr);
    return hipGetLastError();
}

// stored rows (stride ld): decode_item_borrowed's per-fetch validation (mod.rs:889-949) done once
hipError_t launch_validate_rows(const DevIndex &ix, uint32_t n, float limit, uint32_t *d_status, float *d_hdr,
                                hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(validate_vectors_kernel, dim3(n), dim3(64), 0, s, ix, ix.vec, ix.ld, n, limit, d_status, d_hdr);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Distance tile: TQ queries staged in LDS, each row group streams one row from HBM and scores it
// against all TQ queries before moving on (row bytes are read once per query tile).
// ---------------------------------------------------------------------------------------------
constexpr int kRowsPerGroup = 8; // 32 groups x 8 rows = 256 rows per workgroup

template <uint32_t METRIC, bool FUSED, int TQ>
__global__ __launch_bounds__(256) void flat_distance_kernel(FlatArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *ql = reinterpret_cast<float *>(smem); // [TQ][ld]
    const DevIndex &ix = a.ix;
    const int tid = (int)threadIdx.x, lane = tid & 63, j = lane & 7;
    const int g = tid >> 3; // row group 0..31
    const uint32_t q0 = blockIdx.y * TQ;
    const uint32_t ld = ix.ld;
    for (uint32_t i = (uint32_t)tid; i < (uint32_t)TQ * ld; i += 256) {
        uint32_t t = i / ld, c = i % ld;
        uint32_t q = q0 + t;
        ql[i] = (q < a.b && c < ix.dim) ? a.queries[(size_t)q * ix.dim + c] : 0.0f;
    }
    __syncthreads();
    const uint32_t nk = ix.dim_main >> 5; // (the 32-lane tree; the 128-bit kernels take the branch below)
    const uint32_t t0 = (METRIC == kL1) ? 0u : ix.dim_main;
    for (int rr = 0; rr < kRowsPerGroup; ++rr) {
        const uint32_t idx = blockIdx.x * (32u * kRowsPerGroup) + (uint32_t)rr * 32u + (uint32_t)g;
        if (idx >= a.rows) continue; // group-uniform
        const uint32_t scan = a.row0 + idx;
        const uint32_t node = a.subset ? a.subset[scan] : scan;
        const float *row = ix.vec + (size_t)node * ld;
        float res[TQ];
        if (METRIC != kL1 && kernel_w4(ix.fkernel)) { // SSE / NEON host kernels: 16 virtual lanes, two per lane of the group (hvx_device.h)
#pragma unroll
            for (int t = 0; t < TQ; ++t) res[t] = w4_main<METRIC, FUSED>(row, ql + (size_t)t * ld, ix.dim_main, j);
        } else if (METRIC != kL1) {
            float4 acc[TQ];
#pragma unroll
            for (int t = 0; t < TQ; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int slot = chunk_slot(j);
            const float4 *rp = reinterpret_cast<const float4 *>(row) + slot;
            for (uint32_t k = 0; k < nk; ++k) {
                const float4 x = rp[(size_t)k * 8];
#pragma unroll
                for (int t = 0; t < TQ; ++t) {
                    const float4 qq = reinterpret_cast<const float4 *>(ql + (size_t)t * ld)[k * 8 + slot];
                    if (METRIC == kL2) {
                        float d0 = qq.x - x.x, d1 = qq.y - x.y, d2 = qq.z - x.z, d3 = qq.w - x.w;
                        if (FUSED) {
                            acc[t].x = __builtin_fmaf(d0, d0, acc[t].x); acc[t].y = __builtin_fmaf(d1, d1, acc[t].y);
                            acc[t].z = __builtin_fmaf(d2, d2, acc[t].z); acc[t].w = __builtin_fmaf(d3, d3, acc[t].w);
                        } else {
                            acc[t].x = d0 * d0 + acc[t].x; acc[t].y = d1 * d1 + acc[t].y;
                            acc[t].z = d2 * d2 + acc[t].z; acc[t].w = d3 * d3 + acc[t].w;
                        }
                    } else {
                        if (FUSED) {
                            acc[t].x = __builtin_fmaf(qq.x, x.x, acc[t].x); acc[t].y = __builtin_fmaf(qq.y, x.y, acc[t].y);
                            acc[t].z = __builtin_fmaf(qq.z, x.z, acc[t].z); acc[t].w = __builtin_fmaf(qq.w, x.w, acc[t].w);
                        } else {
                            acc[t].x = qq.x * x.x + acc[t].x; acc[t].y = qq.y * x.y + acc[t].y;
                            acc[t].z = qq.z * x.z + acc[t].z; acc[t].w = qq.w * x.w + acc[t].w;
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < TQ; ++t) res[t] = nk ? avx_tree_reduce(acc[t]) : 0.0f;
        } else {
#pragma unroll
            for (int t = 0; t < TQ; ++t) res[t] = 0.0f;
        }
        for (uint32_t e = t0; e < ix.dim; ++e) {
            const float bv = row[e];
#pragma unroll
            for (int t = 0; t < TQ; ++t) {
                const float av = ql[(size_t)t * ld + e];
                if (METRIC == kL2) {
                    float d = av - bv;
                    float pr = d * d;
                    res[t] += pr;
                } else if (METRIC == kCosine) {
                    float pr = av * bv;
                    res[t] += pr;
                } else {
                    res[t] += fabsf(av - bv);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            const uint32_t q = q0 + (uint32_t)t;
            if (q >= a.b) continue;
            float r = res[t];
            if (METRIC == kCosine) r = cosine_finish(r, a.qhdr[q], ix.hdr[node], ql + (size_t)t * ld, row, ix.dim);
            if (j == 0) a.dist[(size_t)q * a.chunk_ld + idx] = r;
        }
    }
}

template <uint32_t METRIC, bool FUSED>
static hipError_t launch_flat_tq(const FlatArgs &a, hipStream_t s) {
    const uint32_t ld = a.ix.ld;
    const uint32_t rows_per_block = 32u * kRowsPerGroup;
    const uint32_t gx = (a.rows + rows_per_block - 1) / rows_per_block;
    // query tile: as many as fit comfortably in LDS (<= 64 KB) and registers
    int tq = 16;
    while (tq > 1 && ((size_t)tq * ld * 4 > 65536 || (uint32_t)tq > a.b * 2u)) tq >>= 1;
    if ((size_t)tq * ld * 4 > 160 * 1024) return hipErrorInvalidValue;
    const size_t lds = (size_t)tq * ld * 4;
    const dim3 grid(gx, (a.b + tq - 1) / tq);
    switch (tq) {
    case 16: hipLaunchKernelGGL((flat_distance_kernel<METRIC, FUSED, 16>), grid, dim3(256), lds, s, a); break;
    case 8: hipLaunchKernelGGL((flat_distance_kernel<METRIC, FUSED, 8>), grid, dim3(256), lds, s, a); break;
    case 4: hipLaunchKernelGGL((flat_distance_kernel<METRIC, FUSED, 4>), grid, dim3(256), lds, s, a); break;
    case 2: hipLaunchKernelGGL((flat_distance_kernel<METRIC, FUSED, 2>), grid, dim3(256), lds, s, a); break;
    default: hipLaunchKernelGGL((flat_distance_kernel<METRIC, FUSED, 1>), grid, dim3(256), lds, s, a); break;
    }
    return hipGetLastError();
}

hipError_t launch_flat_distances(const FlatArgs &a, hipStream_t s) {
    if (a.rows == 0 || a.b == 0) return hipSuccess;
    const bool fused = kernel_fused(a.ix.fkernel);
    switch (a.ix.metric) {
    case kCosine: return fused ? launch_flat_tq<kCosine, true>(a, s) : launch_flat_tq<kCosine, false>(a, s);
    case kL2: return fused ? launch_flat_tq<kL2, true>(a, s) : launch_flat_tq<kL2, false>(a, s);
    default: return launch_flat_tq<kL1, true>(a, s);
    }
}

// ---------------------------------------------------------------------------------------------
// Exact top-k selection: one workgroup per query keeps a 2048-entry candidate pool in LDS, admits
// only (score,id) pairs below the current k-th best, and re-sorts (bitonic) when the pool fills.
// ---------------------------------------------------------------------------------------------
constexpr int kPool = 2048;

__device__ __forceinline__ bool pair_less(float sa, uint32_t ia, float sb, uint32_t ib) {
    return sa < sb || (sa == sb && ia < ib);
}

// in-place bitonic sort of kPool (score,id) pairs by 256 threads
__device__ inline void bitonic_sort_pool(float *ps, uint32_t *pi, int tid) {
    for (int size = 2; size <= kPool; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < kPool / 2; t += 256) {
                int lo = 2 * t - (t & (stride - 1));
                int hi = lo + stride;
                bool up = ((lo & size) == 0);
                float sl = ps[lo], sh = ps[hi];
                uint32_t il = pi[lo], ih = pi[hi];
                bool sw = up ? pair_less(sh, ih, sl, il) : pair_less(sl, il, sh, ih);
                if (sw) { ps[lo] = sh; ps[hi] = sl; pi[lo] = ih; pi[hi] = il; }
            }
        }
    }
    __syncthreads();
}

// the same over the first n (a power of two <= kPool) pool entries
__device__ inline void bitonic_sort_pool_n(float *ps, uint32_t *pi, int tid, int n) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < n / 2; t += 256) {
                int lo = 2 * t - (t & (stride - 1));
                int hi = lo + stride;
                bool up = ((lo & size) == 0);
                float sl = ps[lo], sh = ps[hi];
                uint32_t il = pi[lo], ih = pi[hi];
                bool sw = up ? pair_less(sh, ih, sl, il) : pair_less(sl, il, sh, ih);
                if (sw) { ps[lo] = sh; ps[hi] = sl; pi[lo] = ih; pi[hi] = il; }
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void flat_select_kernel(FlatArgs a, uint32_t *status) {
    __shared__ float ps[kPool];
    __shared__ uint32_t pi[kPool];
    __shared__ uint32_t cnt, bad;
    const uint32_t q = blockIdx.x;
    const int tid = (int)threadIdx.x;
    const uint32_t k = a.k;
    const float inf = __uint_as_float(0x7F800000u);
    // a query rejected by validation (NaN / zero-norm / magnitude: domain.rs:113-157) keeps its status and gets no rows:
    // its scores would be NaN and must not be reported as an invariant violation
    if (status && status[q] != 0u) {
        if (tid == 0) a.top_counts[q] = 0u;
        return;
    }
    const uint32_t have = a.top_counts[q];
    for (int i = tid; i < kPool; i += 256) {
        bool in = (uint32_t)i < have;
        ps[i] = in ? a.top_scores[(size_t)q * k + i] : inf;
        pi[i] = in ? a.top_ids[(size_t)q * k + i] : 0xFFFFFFFFu;
    }
    if (tid == 0) { cnt = have; bad = 0; }
    __syncthreads();
    float thr_s = have >= k ? ps[k - 1] : inf;
    uint32_t thr_i = have >= k ? pi[k - 1] : 0xFFFFFFFFu;
    const float *dq = a.dist + (size_t)q * a.chunk_ld;
    // segments of 1024 scores: one 16-byte load per thread, pushes through an LDS counter, ONE barrier
    // per segment; the pool (2048) is re-sorted and cut back to k whenever it is more than half full,
    // so a segment can never overflow it
    const bool vec_ok = (a.chunk_ld & 3u) == 0u;
    for (uint32_t base = 0; base < a.rows; base += 1024) {
        const uint32_t i0 = base + (uint32_t)tid * 4u;
        float d4[4] = {inf, inf, inf, inf};
        if (vec_ok && i0 + 3 < a.rows) {
            const float4 v = *reinterpret_cast<const float4 *>(dq + i0);
            d4[0] = v.x; d4[1] = v.y; d4[2] = v.z; d4[3] = v.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i0 + e < a.rows) d4[e] = dq[i0 + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t i = i0 + (uint32_t)e;
            if (i < a.rows) {
                float d = d4[e];
                const uint32_t scan = a.row0 + i;
                const uint32_t node = a.subset ? a.subset[scan] : scan;
                if (!score_valid(d)) {
                    bad = 1; // Candidate::try_new rejects the score (model.rs:21-29)
                } else if (pair_less(d, node, thr_s, thr_i)) {
                    uint32_t slot = atomicAdd(&cnt, 1u);
                    ps[slot] = d;
                    pi[slot] = node;
                }
            }
        }
        __syncthreads();
        // re-sort + cut when the pool is more than half full -- or EARLY, while no threshold exists yet: the first 4k scores give the
        // scan a finite bound, after which a segment pushes a handful of scores instead of all 1 024 (round 5: two 2 048-entry sorts
        // per query were most of this kernel's 250 us on the exact scan's first chunk; the selection itself is exact either way)
        const bool early = thr_i == 0xFFFFFFFFu && thr_s == inf && cnt >= 4u * k && cnt >= 256u;
        if (cnt > (uint32_t)(kPool / 2) || early) {
            int np = 64;
            while ((uint32_t)np < cnt) np <<= 1; // entries beyond cnt are +inf padding: the next power of two is enough
            bitonic_sort_pool_n(ps, pi, tid, np);
            const uint32_t keep = cnt < k ? cnt : k;
            for (int t = tid; t < kPool; t += 256)
                if ((uint32_t)t >= keep) { ps[t] = inf; pi[t] = 0xFFFFFFFFu; }
            __syncthreads();
            if (tid == 0) cnt = keep;
            if (keep >= k) { thr_s = ps[k - 1]; thr_i = pi[k - 1]; }
            __syncthreads();
        }
    }
    {
        int np = 64;
        while ((uint32_t)np < cnt) np <<= 1;
        bitonic_sort_pool_n(ps, pi, tid, np);
    }
    const uint32_t keep = cnt < k ? cnt : k;
    for (uint32_t t = (uint32_t)tid; t < keep; t += 256) {
        a.top_scores[(size_t)q * k + t] = ps[t];
        a.top_ids[(size_t)q * k + t] = pi[t];
    }
    if (tid == 0) {
        a.top_counts[q] = keep;
        if (bad && status && status[q] == 0u) status[q] = 8u; // HVX_ERR_INVARIANT
    }
}

// The same selection for ONE long score row per query (the small-batch scan writes the whole [b][candidates] matrix in one
// launch): grid (b, S), every workgroup keeps the top-k of its slice and appends it to the query's pair list
// (cand_sc / cand_id / cand_cnt: the inputs of flat_merge_pairs_kernel, which sorts the S x k pairs into the top list).
__global__ __launch_bounds__(256) void flat_select_slices_kernel(FlatArgs a, uint32_t *status, uint32_t slice_rows, float *cand_sc,
                                                                 uint32_t *cand_id, uint32_t *cand_cnt, uint32_t cand_cap) {
    __shared__ float ps[kPool];
    __shared__ uint32_t pi[kPool];
    __shared__ uint32_t cnt, bad, base_out;
    const uint32_t q = blockIdx.x;
    const int tid = (int)threadIdx.x;
    const uint32_t k = a.k;
    const float inf = __uint_as_float(0x7F800000u);
    if (status && status[q] != 0u) return;
    for (int i = tid; i < kPool; i += 256) { ps[i] = inf; pi[i] = 0xFFFFFFFFu; }
    if (tid == 0) { cnt = 0; bad = 0; }
    __syncthreads();
    float thr_s = inf;
    uint32_t thr_i = 0xFFFFFFFFu;
    const float *dq = a.dist + (size_t)q * a.chunk_ld;
    const uint32_t lo = blockIdx.y * slice_rows, hi = lo + slice_rows < a.rows ? lo + slice_rows : a.rows;
    const bool vec_ok = (a.chunk_ld & 3u) == 0u;
    for (uint32_t base = lo; base < hi; base += 1024) {
        const uint32_t i0 = base + (uint32_t)tid * 4u;
        float d4[4] = {inf, inf, inf, inf};
        if (vec_ok && i0 + 3 < hi) {
            const float4 v = *reinterpret_cast<const float4 *>(dq + i0);
            d4[0] = v.x; d4[1] = v.y; d4[2] = v.z; d4[3] = v.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i0 + e < hi) d4[e] = dq[i0 + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t i = i0 + (uint32_t)e;
            if (i < hi) {
                float d = d4[e];
                const uint32_t scan = a.row0 + i;
                const uint32_t node = a.subset ? a.subset[scan] : scan;
                if (!score_valid(d)) {
                    bad = 1;
                } else if (pair_less(d, node, thr_s, thr_i)) {
                    uint32_t slot = atomicAdd(&cnt, 1u);
                    ps[slot] = d;
                    pi[slot] = node;
                }
            }
        }
        __syncthreads();
        const uint32_t have = cnt;
        // sort (only the filled part of the pool) as soon as k entries can give a threshold, and whenever the pool is half full
        if (have > (uint32_t)(kPool / 2) || (have >= k && thr_i == 0xFFFFFFFFu && thr_s == inf)) {
            int np = 2;
            while ((uint32_t)np < have) np <<= 1;
            bitonic_sort_pool_n(ps, pi, tid, np);
            const uint32_t keep = have < k ? have : k;
            for (int t = tid; t < np; t += 256)
                if ((uint32_t)t >= keep) { ps[t] = inf; pi[t] = 0xFFFFFFFFu; }
            __syncthreads();
            if (tid == 0) cnt = keep;
            if (keep >= k) { thr_s = ps[k - 1]; thr_i = pi[k - 1]; }
            __syncthreads();
        }
    }
    {
        int np = 2;
        while ((uint32_t)np < cnt) np <<= 1;
        bitonic_sort_pool_n(ps, pi, tid, np);
    }
    const uint32_t keep = cnt < k ? cnt : k;
    if (tid == 0) {
        base_out = atomicAdd(&cand_cnt[q], keep);
        if (bad && status && status[q] == 0u) status[q] = 8u; // HVX_ERR_INVARIANT
    }
    __syncthreads();
    for (uint32_t t = (uint32_t)tid; t < keep; t += 256)
        if (base_out + t < cand_cap) {
            cand_sc[(size_t)q * cand_cap + base_out + t] = ps[t];
            cand_id[(size_t)q * cand_cap + base_out + t] = pi[t];
        }
}

hipError_t launch_flat_select_slices(const FlatArgs &a, uint32_t slices, float *cand_sc, uint32_t *cand_id, uint32_t *cand_cnt, uint32_t cand_cap,
                                     hipStream_t s) {
    if (a.b == 0 || slices == 0) return hipSuccess;
    uint32_t slice_rows = ((a.rows + slices - 1) / slices + 1023u) / 1024u * 1024u;
    if (slice_rows == 0) slice_rows = 1024;
    const uint32_t used = (a.rows + slice_rows - 1) / slice_rows;
    hipLaunchKernelGGL(flat_select_slices_kernel, dim3(a.b, used ? used : 1), dim3(256), 0, s, a, const_cast<uint32_t *>(a.qstatus), slice_rows, cand_sc,
                       cand_id, cand_cnt, cand_cap);
    return hipGetLastError();
}

static uint32_t *g_unused_status = nullptr;

hipError_t launch_flat_select(const FlatArgs &a, hipStream_t s) {
    (void)g_unused_status;
    if (a.b == 0) return hipSuccess;
    hipLaunchKernelGGL(flat_select_kernel, dim3(a.b), dim3(256), 0, s, a, const_cast<uint32_t *>(a.qstatus));
    return hipGetLastError();
}

__global__ void flat_finish_kernel(FlatArgs a, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                                   uint32_t *out_status) {
    const uint32_t q = blockIdx.x;
    const uint32_t st = a.qstatus[q];
    const uint32_t n = st == 0 ? a.top_counts[q] : 0u;
    for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) {
        out_ids[(size_t)q * a.k + t] = a.ix.ids[a.top_ids[(size_t)q * a.k + t]];
        out_scores[(size_t)q * a.k + t] = a.top_scores[(size_t)q * a.k + t];
    }
    if (threadIdx.x == 0) {
        out_counts[q] = n;
        if (out_status) out_status[q] = st;
    }
}

hipError_t launch_flat_finish(const FlatArgs &a, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                              uint32_t *out_status, hipStream_t s) {
    if (a.b == 0) return hipSuccess;
    hipLaunchKernelGGL(flat_finish_kernel, dim3(a.b), dim3(64), 0, s, a, out_ids, out_scores, out_counts, out_status);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Multi-shard merge: g sorted lists of <= k per query -> top-k by Candidate order (model.rs:55-61).
// Rank-by-counting: entry x goes to position #{y : y < x}; ids are unique across shards.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_topk_kernel(uint32_t g, uint32_t b, uint32_t k, const uint64_t *ids,
                                                         const float *scores, const uint32_t *counts, size_t ids_stride,
                                                         size_t scores_stride, size_t counts_stride, uint64_t *out_ids,
                                                         float *out_scores, uint32_t *out_counts) {
    // list s of query q: ids + s*ids_stride + q*k (strides in elements: b*k / b*k / b for three dense arrays, or the
    // per-rank payload size for the packed exchange buffer of pyhvx/shard.py)
    const uint32_t q = blockIdx.x;
    uint32_t total = 0;
    for (uint32_t s = 0; s < g; ++s) total += counts[(size_t)s * counts_stride + q];
    const uint32_t outn = total < k ? total : k;
    for (uint32_t e = threadIdx.x; e < g * k; e += blockDim.x) {
        const uint32_t s = e / k, i = e % k;
        if (i >= counts[(size_t)s * counts_stride + q]) continue;
        const float sc = scores[(size_t)s * scores_stride + (size_t)q * k + i];
        const uint64_t id = ids[(size_t)s * ids_stride + (size_t)q * k + i];
        // rank = (entries before it in its own list) + for every other list, a binary search
        uint32_t rank = i;
        for (uint32_t s2 = 0; s2 < g; ++s2) {
            if (s2 == s) continue;
            const uint32_t c2 = counts[(size_t)s2 * counts_stride + q];
            const float *sp = scores + (size_t)s2 * scores_stride + (size_t)q * k;
            const uint64_t *ip = ids + (size_t)s2 * ids_stride + (size_t)q * k;
            uint32_t lo = 0, hi = c2;
            while (lo < hi) {
                uint32_t mid = (lo + hi) >> 1;
                bool less = sp[mid] < sc || (sp[mid] == sc && ip[mid] < id);
                if (less) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < outn) {
            out_ids[(size_t)q * k + rank] = id;
            out_scores[(size_t)q * k + rank] = sc;
        }
    }
    if (threadIdx.x == 0) out_counts[q] = outn;
}

hipError_t launch_merge_topk(uint32_t g, uint32_t b, uint32_t k, const uint64_t *ids, const float *scores,
                             const uint32_t *counts, uint64_t *out_ids, float *out_scores,
                             uint32_t *out_counts, hipStream_t s) {
    return launch_merge_topk_strided(g, b, k, ids, scores, counts, (size_t)b * k, (size_t)b * k, b, out_ids, out_scores, out_counts, s);
}

hipError_t launch_merge_topk_strided(uint32_t g, uint32_t b, uint32_t k, const uint64_t *ids, const float *scores,
                                     const uint32_t *counts, size_t ids_stride, size_t scores_stride, size_t counts_stride,
                                     uint64_t *out_ids, float *out_scores, uint32_t *out_counts, hipStream_t s) {
    if (b == 0) return hipSuccess;
    hipLaunchKernelGGL(merge_topk_kernel, dim3(b), dim3(256), 0, s, g, b, k, ids, scores, counts, ids_stride, scores_stride,
                       counts_stride, out_ids, out_scores, out_counts);
    return hipGetLastError();
}

} // namespace hvx
