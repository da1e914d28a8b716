// hvx_flat_smallb.hip -- candidate generation of the exact scan for SMALL query batches (b <= 128) in ONE pass over the rows.
//
// Where it sits: restricted_exact_scan (crates/db/src/search/vector/restricted.rs:753-835) over a prefiltered candidate set
// is the common shape of BASELINE config #3: a handful of queries (the benchmark issues them one at a time; 32 share a
// batch here) against 10^3..10^6 candidate rows.  With so few queries the scan is a pure stream: every candidate row has
// to cross HBM once and nothing else is big.  The 128 x 128 / 256 x 256 contraction kernels are built for large batches:
// they pad the batch to their query tile, stage both operands through LDS behind barriers, and the filtered pipeline
// around them needs 8-10 launches -- 0.10 of the HBM roofline on the 100 000-candidate group.
//
// This kernel: one wavefront owns 32 candidate rows at a time and keeps them in REGISTERS -- lane l reads 8 consecutive
// stored elements of row (l & 31), depth half (l >> 5), which is exactly the B-operand layout of v_mfma_f32_32x32x16_bf16
// (f32 rows are rounded to bf16 in registers: the one-pass contraction's rounding, covered by the certificate's bound);
// the queries (bf16 hi parts, <= 4 tiles of 32) sit in LDS once per workgroup, padded so that the A-operand
// ds_read_b128 is conflict-free.  No barrier after the query tile is staged, no LDS traffic for the rows, the next group
// of row loads is issued before the current group's MFMAs.  16 wavefronts per CU x 8-16 KB in flight each.  The epilogue
// writes the [b][candidates] score matrix (b x 4 bytes per row against dim x 4 bytes read: 2 % at b = 32, dim = 1536).
// Selection (flat_select_slices: grid b x S), merge, exact re-rank and certificate are the scan's usual tail.
// Algorithmic bytes per launch = candidates x dim x sizeof(row element).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "hvx_device.h"
#include "hvx_flat_mfma.h"

namespace hvx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// two f32 -> two bf16 (RNE) in one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
}

// KIND 0: bf16 rows (a bf16 index, or the bf16 shadow of an f32 index), 2: f32 rows.  NQT: 32-query tiles held in LDS.
// FULL: the hi + lo split (acc += q_hi.x_hi + q_lo.x_hi [+ q_hi.x_lo for f32 rows]; the dropped lo.lo term is covered by the
// certificate's bound) -- false = the one-pass build (q_hi.x_hi only, RerankArgs::extra_rel widens the bound).
// WAVES: wavefronts per workgroup = per CU (the query tiles fill most of the LDS).  U: MFMA steps (16 deep) per register group.
// Output = the raw dot products [b][rows] in a.dist (the selection kernel below turns them into scores).  The query tiles of
// one workgroup must fit the LDS: when they do not (dim 1536 with the lo tile) the depth is split over gridDim.y workgroup
// rows, each streaming its depth range [k0, k0 + klen) of every row -- still one pass over the bytes -- and the partial
// dot products meet in a.dist by float atomic add (two or three addends onto a zeroed word: order-independent).
template <int KIND, int NQT, int WAVES, int U, bool FULL>
__global__ __launch_bounds__(WAVES * 64) void flat_smallb_kernel(MfmaArgs a, uint32_t n_blocks, uint32_t lds_stride, uint32_t klen) {
    const uint32_t k0 = blockIdx.y * klen;
    const bool accumulate = gridDim.y > 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char qlds[];
    constexpr bool F32 = KIND == 2;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t fr = lane & 31u, half = lane >> 5;
    // the query tile(s) of this depth range: [NQT * 32][klen] bf16 hi (then lo), rows padded by 16 bytes
    unsigned char *qlo_lds = qlds + (size_t)NQT * 32u * lds_stride;
    const uint32_t vec_per_row = klen / 8u;
    for (uint32_t i = tid; i < (uint32_t)NQT * 32u * vec_per_row; i += WAVES * 64) {
        const uint32_t r = i / vec_per_row, c = i - r * vec_per_row;
        *reinterpret_cast<uint4 *>(qlds + (size_t)r * lds_stride + c * 16u) = *reinterpret_cast<const uint4 *>(a.qhi + (size_t)r * a.dim + k0 + c * 8u);
        if (FULL)
            *reinterpret_cast<uint4 *>(qlo_lds + (size_t)r * lds_stride + c * 16u) = *reinterpret_cast<const uint4 *>(a.qlo + (size_t)r * a.dim + k0 + c * 8u);
    }
    __syncthreads();
    const unsigned char *qa = qlds + (size_t)fr * lds_stride + half * 16u; // this lane's A fragment of tile 0, step 0
    const unsigned char *qal = qlo_lds + (size_t)fr * lds_stride + half * 16u;
    const uint32_t ngroups = klen / (16u * U);
    struct Group {
        uint4 v[U][F32 ? 2 : 1];
    };
    for (uint32_t blk = blockIdx.x * WAVES + wave; blk < n_blocks; blk += gridDim.x * WAVES) {
        const uint32_t rloc = blk * 32u + fr;
        const uint32_t rcl = rloc < a.nrows ? rloc : a.nrows - 1u; // the ragged tail re-reads the last row; masked below
        const size_t node = a.subset ? a.subset[a.row0 + rcl] : a.row0 + rcl;
        const unsigned char *rp = reinterpret_cast<const unsigned char *>(a.rows) + (node * a.dim + k0) * (F32 ? 4u : 2u) + half * (F32 ? 32u : 16u);
        auto load_group = [&](Group &g, uint32_t grp) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned char *p = rp + (size_t)(grp * U + u) * (F32 ? 64u : 32u);
                g.v[u][0] = *reinterpret_cast<const uint4 *>(p);
                if (F32) g.v[u][F32 ? 1 : 0] = *reinterpret_cast<const uint4 *>(p + 16);
            }
        };
        f32x16 acc[NQT];
#pragma unroll
        for (int t = 0; t < NQT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        auto compute_group = [&](const Group &g, uint32_t grp) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint4 bw, bl = make_uint4(0, 0, 0, 0);
                if (F32) {
                    float x[8];
                    const uint4 x0 = g.v[u][0], x1 = g.v[u][F32 ? 1 : 0];
                    x[0] = __uint_as_float(x0.x); x[1] = __uint_as_float(x0.y); x[2] = __uint_as_float(x0.z); x[3] = __uint_as_float(x0.w);
                    x[4] = __uint_as_float(x1.x); x[5] = __uint_as_float(x1.y); x[6] = __uint_as_float(x1.z); x[7] = __uint_as_float(x1.w);
                    uint32_t h[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) h[i] = pack_bf16(x[2 * i], x[2 * i + 1]);
                    bw = make_uint4(h[0], h[1], h[2], h[3]);
                    if (FULL) { // residuals x - bf16(x), rounded to bf16 again
                        uint32_t l[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            l[i] = pack_bf16(x[2 * i] - __uint_as_float(h[i] << 16), x[2 * i + 1] - __uint_as_float(h[i] & 0xFFFF0000u));
                        bl = make_uint4(l[0], l[1], l[2], l[3]);
                    }
                } else {
                    bw = g.v[u][0];
                }
                const bf16x8 fb = __builtin_bit_cast(bf16x8, bw);
                const bf16x8 fbl = __builtin_bit_cast(bf16x8, bl);
                const uint32_t koff = (grp * U + (uint32_t)u) * 32u; // bytes of bf16 depth before this step
#pragma unroll
                for (int t = 0; t < NQT; ++t) {
                    const bf16x8 fa = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(qa + (size_t)t * 32u * lds_stride + koff));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t], 0, 0, 0);
                    if (FULL) {
                        const bf16x8 fal = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(qal + (size_t)t * 32u * lds_stride + koff));
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal, fb, acc[t], 0, 0, 0);
                        if (F32) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fbl, acc[t], 0, 0, 0);
                    }
                }
            }
        };
        // two register groups: group g + 1 is requested before group g is consumed
        Group g0, g1;
        load_group(g0, 0);
        for (uint32_t grp = 0; grp < ngroups; grp += 2) {
            if (grp + 1 < ngroups) load_group(g1, grp + 1);
            compute_group(g0, grp);
            if (grp + 1 < ngroups) {
                if (grp + 2 < ngroups) load_group(g0, grp + 2);
                compute_group(g1, grp + 1);
            }
        }
        // epilogue: C[m = query][n = row]; lane holds n = lane & 31, m = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        if (rloc < a.nrows) {
#pragma unroll
            for (int t = 0; t < NQT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t qq = (uint32_t)t * 32u + (uint32_t)(e & 3) + 8u * (uint32_t)(e >> 2) + 4u * half;
                    if (qq >= a.b) continue;
                    float *dp = a.dist + (size_t)qq * a.chunk_ld + rloc;
                    if (accumulate) atomicAdd(dp, acc[t][e]);
                    else *dp = acc[t][e];
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Selection without sorting.  The certificate (hvx_flat_mfma.hip) needs, per query, a candidate set C of m + 1 rows and a
// threshold t such that every row outside C has approximate score >= t: the m + 1 smallest scores and the largest of them.
// A workgroup holds 4 096 scores of its slice in registers (16 per thread) and finds the kc-th smallest by a bitwise
// descent over the f32 bit patterns (non-negative scores order like their bits): 32 rounds of "how many keys carry this
// prefix with a 0 here" -- 16 compares per thread, one wave reduction, one LDS add, one barrier.  Keys below the result are
// emitted, ties fill the remaining quota.  Slices longer than one register chunk carry their survivors into the next chunk.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSelR = 16;              // scores per thread and chunk
constexpr int kSelChunk = kSelR * 256; // 4 096

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

struct SelectArgs {
    const float *dots;        // [b][chunk_ld] raw dot products
    uint32_t chunk_ld, rows, row0;
    const uint32_t *subset;   // scan position -> row (restricted scans), or NULL
    const float *rowterm;     // [n] |x|^2 (L2) / |x| (cosine)
    const float *qn2;         // [b]
    const float *rowscale;    // unused (fp8 rows do not take this path)
    uint32_t metric, b, kc, slice_rows;
    uint32_t *status;         // [b] query status: rejected queries are skipped, an invalid score sets 8
    float *sl_sc;             // [b][slices * kc] out: every slice's kc smallest (score, row) pairs, padded with +inf
    uint32_t *sl_id;
    uint32_t sl_stride;
};

__global__ __launch_bounds__(256) void flat_select_radix_kernel(SelectArgs a) {
    __shared__ uint32_t sv_key[256], sv_id[256], cnt3[3], n_lt, n_tie, bad;
    const uint32_t q = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x, lane = tid & 63u;
    const uint32_t kc = a.kc;
    float *out_sc = a.sl_sc + (size_t)q * a.sl_stride + (size_t)sl * kc;
    uint32_t *out_id = a.sl_id + (size_t)q * a.sl_stride + (size_t)sl * kc;
    const float inf = __uint_as_float(0x7F800000u);
    if (a.status && a.status[q] != 0u) {
        for (uint32_t i = tid; i < kc; i += 256) { out_sc[i] = inf; out_id[i] = 0xFFFFFFFFu; }
        return;
    }
    const uint32_t lo = sl * a.slice_rows, hi = lo + a.slice_rows < a.rows ? lo + a.slice_rows : a.rows;
    const float *dq = a.dots + (size_t)q * a.chunk_ld;
    const float qn2 = a.qn2[q], qn = sqrtf(qn2);
    uint32_t n_surv = 0; // survivors carried from the previous chunk (in sv_key / sv_id)
    if (tid == 0) bad = 0;
    __syncthreads();
    for (uint32_t base = lo; base < hi; base += kSelChunk) {
        if (tid == 0) { cnt3[0] = cnt3[1] = cnt3[2] = 0; n_lt = 0; n_tie = 0; }
        uint32_t key[kSelR + 1];
#pragma unroll
        for (int r = 0; r < kSelR; ++r) {
            const uint32_t i = base + (uint32_t)r * 256u + tid;
            uint32_t kbits = 0xFFFFFFFFu;
            if (i < hi) {
                const uint32_t scan = a.row0 + i;
                const uint32_t node = a.subset ? a.subset[scan] : scan;
                const float dot = dq[i], term = a.rowterm[node];
                float s;
                if (a.metric == kL2) {
                    s = (qn2 + term) - 2.0f * dot;
                    s = s < 0.f ? 0.f : s;
                } else {
                    const float den = qn * term;
                    float c = den > 0.f ? dot / den : 0.f;
                    c = c < -1.f ? -1.f : (c > 1.f ? 1.f : c);
                    s = (1.0f - c) * 0.5f;
                }
                if (!score_valid(s)) bad = 1; // Candidate::try_new rejects the score (model.rs:21-29)
                else kbits = __float_as_uint(s);
            }
            key[r] = kbits;
        }
        key[kSelR] = tid < n_surv ? sv_key[tid] : 0xFFFFFFFFu;
        const uint32_t carried_id = tid < n_surv ? sv_id[tid] : 0xFFFFFFFFu;
        const uint32_t in_chunk = (hi - base) < (uint32_t)kSelChunk ? (hi - base) : (uint32_t)kSelChunk;
        const uint32_t total = in_chunk + n_surv;
        const uint32_t kth = kc < total ? kc : total;
        __syncthreads(); // counters zeroed; every thread holds its carried survivor in a register
        // bitwise descent to the kth smallest key (an invalid score holds the all-ones key and sorts last)
        uint32_t prefix = 0, kk = kth, less = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t hi_mask = bit == 31 ? 0u : (0xFFFFFFFFu << (bit + 1));
            // count per wavefront with ballots (a compare into a scalar mask + s_bcnt1): no cross-lane traffic
            const uint32_t want = prefix, sel_mask = hi_mask | (1u << bit); // prefix on the bits above, 0 at `bit`
            uint32_t c = 0;
#pragma unroll
            for (int r = 0; r <= kSelR; ++r) c += (uint32_t)__builtin_popcountll(__ballot((key[r] & sel_mask) == want));
            const int slot = bit % 3;
            if (lane == 0 && c) atomicAdd(&cnt3[slot], c);
            if (tid == 0) cnt3[(bit + 2) % 3] = 0; // next round's counter: last read two rounds ago
            __syncthreads();
            const uint32_t zeros = cnt3[slot];
            if (kk > zeros) { prefix |= 1u << bit; kk -= zeros; less += zeros; }
        }
        // emit into the survivor arrays: keys below the kth value, then ties up to the quota
        const uint32_t quota = kth - less;
#pragma unroll
        for (int r = 0; r <= kSelR; ++r) {
            const uint32_t kb = key[r];
            bool take = kb < prefix;
            if (!take && kb == prefix && kb != 0xFFFFFFFFu) take = atomicAdd(&n_tie, 1u) < quota;
            if (take) {
                uint32_t id = carried_id;
                if (r < kSelR) {
                    const uint32_t scan = a.row0 + base + (uint32_t)r * 256u + tid;
                    id = a.subset ? a.subset[scan] : scan;
                }
                const uint32_t pos = atomicAdd(&n_lt, 1u);
                if (pos < 256u) { sv_key[pos] = kb; sv_id[pos] = id; }
            }
        }
        __syncthreads();
        n_surv = n_lt < kth ? n_lt : kth;
        __syncthreads(); // n_lt is re-zeroed at the top of the next chunk
    }
    __syncthreads();
    for (uint32_t i = tid; i < kc; i += 256) {
        out_sc[i] = i < n_surv ? __uint_as_float(sv_key[i]) : inf;
        out_id[i] = i < n_surv ? sv_id[i] : 0xFFFFFFFFu;
    }
    if (tid == 0 && lo < hi && bad && a.status && a.status[q] == 0u) a.status[q] = 8u; // HVX_ERR_INVARIANT
}

constexpr size_t kSmallbLds = 150u * 1024u;

static uint32_t smallb_tiles(uint32_t b) {
    const uint32_t nqt = (b + 31u) / 32u;
    return nqt == 3 ? 4u : nqt;
}
// depth passes needed so that the query tiles of one pass fit the LDS (0 = not servable)
static uint32_t smallb_passes(uint32_t dim, uint32_t b, bool full) {
    const uint32_t tiles = smallb_tiles(b) * (full ? 2u : 1u);
    for (uint32_t p = 1; p <= 4; ++p) {
        if (dim % (p * 128u) != 0u) continue; // a pass is a whole number of 8-step register groups
        if ((size_t)tiles * 32u * ((size_t)(dim / p) * 2u + 16u) <= kSmallbLds) return p;
    }
    return 0;
}

bool flat_smallb_supported(uint32_t dim, uint32_t b, int kind) {
    if (kind != 0 && kind != 2) return false;
    if (b == 0 || b > 128u || dim % 128u != 0u) return false;
    return smallb_passes(dim, b, true) != 0 && smallb_passes(dim, b, false) != 0;
}

template <int KIND, int NQT, bool FULL>
static hipError_t launch_smallb_t(const MfmaArgs &a, uint32_t n_blocks, uint32_t cus, hipStream_t s) {
    // 8 wavefronts per CU with 256 registers each: two groups of 8 steps in flight (16 KB of f32 rows per wavefront)
    constexpr int WAVES = 8, U = NQT == 4 ? (KIND == 2 ? 2 : 4) : ((NQT == 2 && KIND == 2 && FULL) ? 4 : 8);
    const uint32_t passes = smallb_passes(a.dim, a.b, FULL);
    if (passes == 0) return hipErrorInvalidValue;
    const uint32_t klen = a.dim / passes, stride = klen * 2u + 16u;
    const size_t lds = (size_t)NQT * (FULL ? 2u : 1u) * 32u * stride;
    auto kern = flat_smallb_kernel<KIND, NQT, WAVES, U, FULL>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (passes > 1) { // partial dot products are added into zeroed words
        hipError_t e = hipMemsetAsync(a.dist, 0, (size_t)a.b * a.chunk_ld * 4, s);
        if (e != hipSuccess) return e;
    }
    const uint32_t per = std::max<uint32_t>(1u, cus / passes);
    const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>(per, (n_blocks + WAVES - 1) / WAVES));
    hipLaunchKernelGGL(kern, dim3(grid, passes), dim3(WAVES * 64), lds, s, a, n_blocks, stride, klen);
    return hipGetLastError();
}

template <int KIND, bool FULL> static hipError_t launch_smallb_q(const MfmaArgs &a, uint32_t n_blocks, uint32_t cus, hipStream_t s) {
    switch (smallb_tiles(a.b)) {
    case 1: return launch_smallb_t<KIND, 1, FULL>(a, n_blocks, cus, s);
    case 2: return launch_smallb_t<KIND, 2, FULL>(a, n_blocks, cus, s);
    default: return launch_smallb_t<KIND, 4, FULL>(a, n_blocks, cus, s);
    }
}

// raw dot products of rows [a.row0, a.row0 + a.nrows) of the scan order against a.b <= 128 queries into a.dist (row length a.chunk_ld)
hipError_t launch_flat_smallb(const MfmaArgs &a, int kind, bool full, uint32_t cus, hipStream_t s) {
    if (a.nrows == 0) return hipSuccess;
    if (!flat_smallb_supported(a.dim, a.b, kind)) return hipErrorInvalidValue;
    const uint32_t n_blocks = (a.nrows + 31u) / 32u;
    if (kind == 2) return full ? launch_smallb_q<2, true>(a, n_blocks, cus, s) : launch_smallb_q<2, false>(a, n_blocks, cus, s);
    return full ? launch_smallb_q<0, true>(a, n_blocks, cus, s) : launch_smallb_q<0, false>(a, n_blocks, cus, s);
}

// slices per query for a scan of `rows` rows with kc <= 256: <= 32, each a whole number of 4 096-score register chunks
uint32_t flat_smallb_slices(uint32_t rows) {
    const uint32_t chunks = (rows + (uint32_t)kSelChunk - 1) / (uint32_t)kSelChunk;
    return std::max<uint32_t>(1u, std::min<uint32_t>(32u, chunks));
}

// the kc smallest approximate scores of every slice: a.dist (raw dots) -> sl_sc / sl_id [b][slices * kc] (row length sl_stride)
hipError_t launch_flat_select_radix(const MfmaArgs &a, uint32_t kc, uint32_t *status, float *sl_sc, uint32_t *sl_id, uint32_t sl_stride,
                                    uint32_t *out_slices, hipStream_t s) {
    if (a.b == 0 || kc == 0 || kc > 256u) return hipErrorInvalidValue;
    const uint32_t slices = flat_smallb_slices(a.nrows);
    const uint32_t chunks = (a.nrows + (uint32_t)kSelChunk - 1) / (uint32_t)kSelChunk;
    SelectArgs g{};
    g.dots = a.dist; g.chunk_ld = a.chunk_ld; g.rows = a.nrows; g.row0 = a.row0; g.subset = a.subset; g.rowterm = a.rowterm; g.qn2 = a.qn2;
    g.metric = a.metric; g.b = a.b; g.kc = kc; g.slice_rows = ((chunks + slices - 1) / slices) * (uint32_t)kSelChunk;
    g.status = status; g.sl_sc = sl_sc; g.sl_id = sl_id; g.sl_stride = sl_stride;
    const uint32_t used = (a.nrows + g.slice_rows - 1) / g.slice_rows;
    if ((size_t)used * kc > sl_stride) return hipErrorInvalidValue;
    *out_slices = used;
    hipLaunchKernelGGL(flat_select_radix_kernel, dim3(a.b, used), dim3(256), 0, s, g);
    return hipGetLastError();
}

} // namespace hvx
