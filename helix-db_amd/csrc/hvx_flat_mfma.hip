// hvx_flat_mfma.hip -- exact scan over bf16 / fp8 rows as a dense batched-query x corpus contraction on the
// matrix cores (BASELINE configs #4/#5: "MFMA GEMM path only for the brute-force flat scan").
//
// Semantics are still restricted_exact_scan (crates/db/src/search/vector/restricted.rs:753-835): the k
// smallest (score, id) pairs, scores produced by D::distance in the host kernel's summation order.  An
// MFMA sums in a different order, so the pipeline is candidate generation + exact re-rank + certificate:
//   1. queries are split into bf16 hi + lo parts (residual <= 2^-18 |q|) in the rows' interleaved layout
//      (a dot product does not care about the permutation as long as both operands use it);
//   2. `flat_mfma_bf16_kernel`: 128 queries x 128 rows per workgroup, v_mfma_f32_32x32x16_bf16,
//      acc += q_hi.x + q_lo.x, epilogue turns the dot product into the metric's score and writes one
//      chunk of the [b][rows] score matrix (fp8 rows: codes widened to bf16 -- exactly -- on the way into
//      LDS, the row scale applied in the epilogue);
//   3. the exact top-(m+1) by approximate score per query is kept (flat_select_kernel, m = max(63, 2k));
//   4. `rerank_kernel`: one wavefront per query re-scores those candidates with the reference's
//      summation order (the same gather/FMA code as the HNSW kernel), sorts by (score, id), keeps k;
//   5. certificate: every row that was NOT re-scored has approximate score >= t (the (m+1)-th), hence a
//      reference-order score >= t - E; if the k-th exact score is < t - E the answer is provably the
//      exact scan's.  E bounds the bf16 split residual plus the f32 accumulation error of both orders
//      (2e-5 + 12 dim 2^-24, relative to (|q|^2 + max|x|^2)/2 for L2, absolute for cosine).
//      Queries that fail are re-run with m = 1023; if that fails too they are reported, never guessed.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "hvx_hnsw_wave.h"
#include "hvx_host.h"
#include "hvx_flat_mfma.h"

using namespace hvx;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(HVX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

namespace hvx {

constexpr int HVX_MFMA_FALLBACK = -1; // flat_mfma_device over f32 rows: certificate not reached, run the exact VALU scan

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---- per-row term of the approximate score: |x|^2 for L2 (cosine uses the norm header) ----
__global__ __launch_bounds__(64) void bf16_row_norm2_kernel(const uint16_t *rows, uint32_t n, uint32_t dim, float *out) {
    const uint32_t r = blockIdx.x;
    if (r >= n) return;
    const uint16_t *p = rows + (size_t)r * dim;
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < dim; i += 64) {
        const double v = (double)bf16_to_f32(p[i]);
        acc += v * v;
    }
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if (threadIdx.x == 0) out[r] = (float)acc;
}

// ---- bf16 shadow of f32 rows (plain element order): what the 256 x 256 kernel streams instead of the f32 rows ----
__global__ __launch_bounds__(256) void bf16_shadow_kernel(const float *rows, size_t count4, uint16_t *out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4 *>(rows)[i];
        const uint32_t a = f32_to_bf16_rne(v.x), b = f32_to_bf16_rne(v.y), c = f32_to_bf16_rne(v.z), d = f32_to_bf16_rne(v.w);
        reinterpret_cast<uint2 *>(out)[i] = make_uint2(a | (b << 16), c | (d << 16));
    }
}

// ---- queries: f32 -> bf16 hi + lo in the interleaved layout, padded with zero rows; |q|^2 ----
__global__ __launch_bounds__(64) void split_queries_kernel(const float *q, uint32_t b, uint32_t bpad, uint32_t dim,
                                                           uint16_t *qhi, uint16_t *qlo, float *qn2, uint32_t fp8_layout,
                                                           uint16_t *qhi_tile8) {
    const uint32_t r = blockIdx.x;
    if (r >= bpad) return;
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < dim; i += 64) {
        float v = r < b ? q[(size_t)r * dim + i] : 0.f;
        if (!f32_is_finite(v)) v = 0.f; // rejected queries are masked by their status; keep the GEMM finite
        const uint16_t h = f32_to_bf16_rne(v);
        const float res = v - bf16_to_f32(h);
        const uint32_t s = fp8_layout == 2u ? i : (fp8_layout ? fp8_slot_of(i) : bf16_slot_of(i)); // 2: f32 rows, plain order
        qhi[(size_t)r * dim + s] = h;
        qlo[(size_t)r * dim + s] = f32_to_bf16_rne(res);
        if (qhi_tile8) qhi_tile8[(size_t)r * dim + tile_slot_fp8(s)] = h; // operand order of the 256 x 256 fp8 kernel
        acc += (double)v * (double)v;
    }
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if (threadIdx.x == 0 && r < b) qn2[r] = (float)acc;
}

// The query operand of the MX-scaled fp8 tile build (hvx_flat_tile.hip, flat_tile2mx_kernel): v_mfma_scale_f32_32x32x64_f8f6f4 multiplies
// fp8 by fp8, so a query value is carried as TWO e4m3 pieces -- hi = RNE(q / s), lo = RNE((q - s hi) 16 / s), s = 2^E the power of two
// that puts max |q| into (224, 448] -- and the instruction's block scales (E8M0: 127 + E for the hi piece, 4 less for the lo piece)
// restore the magnitudes: q ~ s hi + (s / 16) lo with |q - that| <= 2^-8 |q| per element (3 mantissa bits twice; subnormal pieces add
// <= 2^-18 max |q| absolute).  Layout: [query][dim / 64 stages][64 hi codes | 64 lo codes], element i at position fp8_slot_of(i) of the
// row (the order the fp8 rows are stored in): a lane of the tile kernel reads 32 consecutive codes of a row and of a query.
__global__ __launch_bounds__(64) void split_queries_mx_kernel(const float *q, uint32_t b, uint32_t bpad, uint32_t dim, unsigned char *qmx, uint32_t *qexp) {
    const uint32_t r = blockIdx.x;
    if (r >= bpad) return;
    float mx = 0.f;
    for (uint32_t i = threadIdx.x; i < dim; i += 64) {
        const float v = r < b ? q[(size_t)r * dim + i] : 0.f;
        if (f32_is_finite(v)) mx = fmaxf(mx, fabsf(v));
    }
    for (int s = 32; s > 0; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s, 64));
    int e = 0;
    if (mx > 0.f) {
        int ex;
        (void)frexpf(mx, &ex); // mx = f 2^ex, f in [0.5, 1): mx / 2^(ex - 9) in [256, 512) would overflow 448 -> one more
        e = ex - 8;            // mx / 2^e in [128, 256)
    }
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    const float inv = ldexpf(1.0f, -e), sc = ldexpf(1.0f, e);
    for (uint32_t i = threadIdx.x; i < dim; i += 64) {
        float v = r < b ? q[(size_t)r * dim + i] : 0.f;
        if (!f32_is_finite(v)) v = 0.f;
        const uint8_t hi = fp8_e4m3_encode(v * inv);
        const float res = v - fp8_e4m3_decode(hi) * sc; // exact: both are multiples of the hi step, within f32's 24 bits
        const uint8_t lo = fp8_e4m3_encode(res * inv * 16.0f);
        const uint32_t s = fp8_slot_of(i), at = (s >> 6) * 128u + (s & 63u);
        qmx[(size_t)r * dim * 2 + at] = hi;
        qmx[(size_t)r * dim * 2 + at + 64u] = lo;
    }
    if (threadIdx.x == 0) qexp[r] = (uint32_t)(127 + e);
}

// ---- the contraction ----
constexpr int kBM = 128, kBN = 128, kBK = 32; // queries x rows x depth per stage
constexpr int kLdsStride = kBK * 2 + 16;      // bytes per tile row: 64 B of data + 16 B pad => conflict-free ds_read_b128


// KIND: 0 = bf16 rows, 1 = fp8 rows, 2 = f32 rows (split into bf16 hi + lo on the way into LDS: acc += q_hi.x_hi +
// q_lo.x_hi + q_hi.x_lo; the dropped q_lo.x_lo term is <= 2^-16 |q||x| and is covered by the certificate's bound).
// FULL = false: the ONE-pass build -- only q_hi.x_hi; the dropped residual terms (<= 2^-9 |q||x| per rounded operand) are
// added to the certificate's bound instead (RerankArgs::extra_rel), queries that then fail it are repeated with FULL.
// FILT = true: the score matrix is not written; a score leaves the tile only if it is below the query's running threshold.
template <int KIND, bool FULL, bool FILT>
__global__ __launch_bounds__(256) void flat_mfma_bf16_kernel(MfmaArgs a) {
    constexpr bool FP8 = KIND == 1, F32 = KIND == 2, LOB = F32 && FULL;
    __shared__ __attribute__((aligned(16))) unsigned char lds[(2 + (FULL ? 1 : 0) + (LOB ? 1 : 0)) * kBM * kLdsStride]; // A_hi | B (| A_lo) (| B_lo)
    unsigned char *sAh = lds, *sB = lds + kBM * kLdsStride, *sAl = lds + 2 * kBM * kLdsStride;
    unsigned char *sBl = lds + 3 * kBM * kLdsStride;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1; // 64x64 sub-tile of the wave
    uint32_t qt, rt;
    {
        const uint32_t per_group = a.group_tiles * a.nq_tiles;
        const uint32_t grp = blockIdx.x / per_group, rem = blockIdx.x % per_group;
        const uint32_t in_group = a.nr_tiles - grp * a.group_tiles < a.group_tiles ? a.nr_tiles - grp * a.group_tiles : a.group_tiles;
        qt = rem / in_group;
        rt = grp * a.group_tiles + rem % in_group;
        if (qt >= a.nq_tiles) return; // the last (short) group leaves workgroup ids over
    }
    const uint32_t q0 = qt * kBM, r0 = rt * kBN;
    // staging: thread owns 16 bytes (8 bf16) of tile rows sr and sr+64, depth chunk sc
    const int sr = tid >> 2, sc = tid & 3;
    const uint32_t rowA0 = q0 + sr, rowA1 = q0 + sr + 64; // queries are padded to a multiple of 128
    uint32_t rowB0 = r0 + sr, rowB1 = r0 + sr + 64;
    if (rowB0 >= a.nrows) rowB0 = a.nrows - 1; // clamp: duplicates are masked in the epilogue
    if (rowB1 >= a.nrows) rowB1 = a.nrows - 1;
    const uint16_t *gAh0 = a.qhi + (size_t)rowA0 * a.dim + sc * 8, *gAh1 = a.qhi + (size_t)rowA1 * a.dim + sc * 8;
    const uint16_t *gAl0 = a.qlo + (size_t)rowA0 * a.dim + sc * 8, *gAl1 = a.qlo + (size_t)rowA1 * a.dim + sc * 8;
    // restricted scans walk a row list: the tile's rows are gathered through it
    const size_t nodeB0 = a.subset ? a.subset[a.row0 + rowB0] : a.row0 + rowB0, nodeB1 = a.subset ? a.subset[a.row0 + rowB1] : a.row0 + rowB1;
    const uint16_t *gB0 = reinterpret_cast<const uint16_t *>(a.rows) + nodeB0 * a.dim + sc * 8;
    const uint16_t *gB1 = reinterpret_cast<const uint16_t *>(a.rows) + nodeB1 * a.dim + sc * 8;
    const int so0 = sr * kLdsStride + sc * 16, so1 = (sr + 64) * kLdsStride + sc * 16;
    // fp8 rows: ONE 16-byte load per thread and stage = 16 codes of tile row tid/2, depth half tid&1,
    // widened to bf16 (exact) on the way into LDS
    uint32_t rowF = r0 + (uint32_t)(tid >> 1);
    if (rowF >= a.nrows) rowF = a.nrows - 1;
    const size_t nodeF = a.subset ? a.subset[a.row0 + rowF] : a.row0 + rowF;
    const uint8_t *gF = reinterpret_cast<const uint8_t *>(a.rows) + nodeF * a.dim + (tid & 1) * 16;
    const int soF = (tid >> 1) * kLdsStride + (tid & 1) * 32;
    // f32 rows: 8 consecutive floats (two 16-byte loads) of tile rows sr and sr+64 per thread and stage
    const float *gX0 = reinterpret_cast<const float *>(a.rows) + nodeB0 * a.dim + sc * 8;
    const float *gX1 = reinterpret_cast<const float *>(a.rows) + nodeB1 * a.dim + sc * 8;
    auto split8 = [](const float4 &u, const float4 &v, uint4 &hi, uint4 &lo) { // 8 f32 -> 8 bf16 hi + 8 bf16 residuals
        const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
        uint32_t h[8], l[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint16_t hb = f32_to_bf16_rne(x[i]);
            h[i] = hb;
            l[i] = f32_to_bf16_rne(x[i] - bf16_to_f32(hb));
        }
        hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
        lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
    };
    float4 pX0a, pX0b, pX1a, pX1b;
    auto widen = [](uint32_t w, uint32_t &lo, uint32_t &hi) { // 4 fp8 -> 4 bf16 (two words)
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 a01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false), a23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
        lo = (__float_as_uint(a01[1]) & 0xFFFF0000u) | (__float_as_uint(a01[0]) >> 16);
        hi = (__float_as_uint(a23[1]) & 0xFFFF0000u) | (__float_as_uint(a23[0]) >> 16);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    // fragment addresses: lane holds 8 consecutive depth values of tile row (lane & 31), depth group lane >> 5
    const int fr = lane & 31, fk = (lane >> 5) * 16; // byte offset of the lane's 8 bf16 inside a 16-deep step
    uint4 pAh0 = *reinterpret_cast<const uint4 *>(gAh0), pAh1 = *reinterpret_cast<const uint4 *>(gAh1);
    uint4 pAl0 = make_uint4(0, 0, 0, 0), pAl1 = pAl0;
    if (FULL) { pAl0 = *reinterpret_cast<const uint4 *>(gAl0); pAl1 = *reinterpret_cast<const uint4 *>(gAl1); }
    uint4 pB0 = make_uint4(0, 0, 0, 0), pB1 = pB0;
    if (F32) {
        pX0a = *reinterpret_cast<const float4 *>(gX0); pX0b = *reinterpret_cast<const float4 *>(gX0 + 4);
        pX1a = *reinterpret_cast<const float4 *>(gX1); pX1b = *reinterpret_cast<const float4 *>(gX1 + 4);
    } else if (FP8) { pB0 = *reinterpret_cast<const uint4 *>(gF); pB1 = pB0; }
    else { pB0 = *reinterpret_cast<const uint4 *>(gB0); pB1 = *reinterpret_cast<const uint4 *>(gB1); }
    for (uint32_t k0 = 0; k0 < a.dim; k0 += kBK) {
        __syncthreads(); // previous stage fully consumed
        *reinterpret_cast<uint4 *>(sAh + so0) = pAh0; *reinterpret_cast<uint4 *>(sAh + so1) = pAh1;
        if (FULL) { *reinterpret_cast<uint4 *>(sAl + so0) = pAl0; *reinterpret_cast<uint4 *>(sAl + so1) = pAl1; }
        if (F32) {
            uint4 h0, l0, h1, l1;
            split8(pX0a, pX0b, h0, l0);
            split8(pX1a, pX1b, h1, l1);
            *reinterpret_cast<uint4 *>(sB + so0) = h0; *reinterpret_cast<uint4 *>(sB + so1) = h1;
            if (LOB) { *reinterpret_cast<uint4 *>(sBl + so0) = l0; *reinterpret_cast<uint4 *>(sBl + so1) = l1; }
        } else if (FP8) {
            uint4 w0, w1;
            widen(pB0.x, w0.x, w0.y); widen(pB0.y, w0.z, w0.w); widen(pB0.z, w1.x, w1.y); widen(pB0.w, w1.z, w1.w);
            *reinterpret_cast<uint4 *>(sB + soF) = w0; *reinterpret_cast<uint4 *>(sB + soF + 16) = w1;
        } else {
            *reinterpret_cast<uint4 *>(sB + so0) = pB0; *reinterpret_cast<uint4 *>(sB + so1) = pB1;
        }
        __syncthreads();
        if (k0 + kBK < a.dim) { // next stage's global loads fly under this stage's MFMAs
            const uint32_t kn = k0 + kBK;
            pAh0 = *reinterpret_cast<const uint4 *>(gAh0 + kn); pAh1 = *reinterpret_cast<const uint4 *>(gAh1 + kn);
            if (FULL) { pAl0 = *reinterpret_cast<const uint4 *>(gAl0 + kn); pAl1 = *reinterpret_cast<const uint4 *>(gAl1 + kn); }
            if (F32) {
                pX0a = *reinterpret_cast<const float4 *>(gX0 + kn); pX0b = *reinterpret_cast<const float4 *>(gX0 + kn + 4);
                pX1a = *reinterpret_cast<const float4 *>(gX1 + kn); pX1b = *reinterpret_cast<const float4 *>(gX1 + kn + 4);
            } else if (FP8) pB0 = *reinterpret_cast<const uint4 *>(gF + kn);
            else { pB0 = *reinterpret_cast<const uint4 *>(gB0 + kn); pB1 = *reinterpret_cast<const uint4 *>(gB1 + kn); }
        }
#pragma unroll
        for (int kk = 0; kk < kBK / 16; ++kk) {
            bf16x8 fah[2], fal[2], fb[2], fbl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int off = (wm * 64 + i * 32 + fr) * kLdsStride + kk * 32 + fk;
                fah[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(sAh + off));
                if (FULL) fal[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(sAl + off));
            }
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const int off = (wn * 64 + jn * 32 + fr) * kLdsStride + kk * 32 + fk;
                fb[jn] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(sB + off));
                if (LOB) fbl[jn] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(sBl + off));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fb[jn], acc[i][jn], 0, 0, 0);
                    if (FULL) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fb[jn], acc[i][jn], 0, 0, 0);
                    if (LOB) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbl[jn], acc[i][jn], 0, 0, 0);
                }
        }
    }
    // epilogue: C[m = query][n = row]; lane holds n = lane & 31, m = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float thr_r[2][16]; // FILT: the running thresholds of this lane's 32 queries
    if (FILT) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t qq = q0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                thr_r[i][e] = qq < a.b ? a.thr[qq] : 0.f;
            }
    }
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const uint32_t rloc = r0 + wn * 64 + jn * 32 + (lane & 31);
        if (rloc >= a.nrows) continue;
        const size_t node = a.subset ? a.subset[a.row0 + rloc] : a.row0 + rloc;
        const float term = a.rowterm[node];
        const float rsc = FP8 ? a.rowscale[node] : 1.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t qq = q0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (qq >= a.b) continue;
                const float dot = FP8 ? acc[i][jn][e] * rsc : acc[i][jn][e];
                float s;
                if (a.metric == kL2) {
                    s = (a.qn2[qq] + term) - 2.0f * dot;
                    s = s < 0.f ? 0.f : s;
                } else {
                    const float den = sqrtf(a.qn2[qq]) * term;
                    float c = den > 0.f ? dot / den : 0.f;
                    c = c < -1.f ? -1.f : (c > 1.f ? 1.f : c);
                    s = (1.0f - c) * 0.5f;
                }
                if (FILT) {
                    if (s < thr_r[i][e]) { // rare once the threshold has settled: ~ (m+1) / rows-seen per row
                        const uint32_t pos = atomicAdd(&a.cand_cnt[qq], 1u);
                        if (pos < a.cand_cap) {
                            a.cand_sc[(size_t)qq * a.cand_cap + pos] = s;
                            a.cand_id[(size_t)qq * a.cand_cap + pos] = (uint32_t)node;
                        }
                    }
                } else {
                    a.dist[(size_t)qq * a.chunk_ld + rloc] = s;
                }
            }
    }
}

// small-batch path, kc > 256: turn the raw dot products of hvx_flat_smallb.hip into scores in place (the sorted-pool selection reads scores)
__global__ __launch_bounds__(256) void flat_dots_to_scores_kernel(MfmaArgs a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, q = blockIdx.y;
    if (i >= a.nrows) return;
    const size_t node = a.subset ? a.subset[a.row0 + i] : a.row0 + i;
    const float dot = a.dist[(size_t)q * a.chunk_ld + i], term = a.rowterm[node];
    float s;
    if (a.metric == kL2) {
        s = (a.qn2[q] + term) - 2.0f * dot;
        s = s < 0.f ? 0.f : s;
    } else {
        const float den = sqrtf(a.qn2[q]) * term;
        float c = den > 0.f ? dot / den : 0.f;
        c = c < -1.f ? -1.f : (c > 1.f ? 1.f : c);
        s = (1.0f - c) * 0.5f;
    }
    a.dist[(size_t)q * a.chunk_ld + i] = s;
}

// ---- FILT pipeline: merge the (score, row) pairs a filtered launch produced into the query's running top-kc list ----
constexpr int kPairPool = 2048;
struct PairMergeArgs {
    float *top_scores;        // [b][kc] ascending by (score, id)
    uint32_t *top_ids;        // [b][kc]
    uint32_t *top_counts;     // [b]
    float *cand_sc;           // [b][cap]
    uint32_t *cand_id;
    uint32_t *cand_cnt;       // [b], reset to 0 here
    float *thr;               // [b] out: score of the kc-th entry, +inf while the list is shorter
    uint32_t *overflow;       // [1] set when a query produced more pairs than cap
    const uint32_t *qstatus;  // [b]
    uint32_t kc, cap;
};
__global__ __launch_bounds__(256) void flat_merge_pairs_kernel(PairMergeArgs a) {
    __shared__ float ps[kPairPool];
    __shared__ uint32_t pi[kPairPool];
    const uint32_t q = blockIdx.x;
    const int tid = (int)threadIdx.x;
    const float inf = __uint_as_float(0x7F800000u);
    const uint32_t have = a.top_counts[q];
    uint32_t nc = a.cand_cnt[q];
    if (nc > a.cap) { if (tid == 0) *a.overflow = 1u; nc = a.cap; }
    if (a.qstatus[q] != 0u) nc = 0;
    // the pool holds have + nc entries: sort the next power of two (a settled threshold lets ~200 pairs through, not 2 048)
    int pool = 64;
    while ((uint32_t)pool < have + nc) pool <<= 1;
    for (int i = tid; i < pool; i += 256) {
        float sv = inf;
        uint32_t iv = 0xFFFFFFFFu;
        if ((uint32_t)i < have) { sv = a.top_scores[(size_t)q * a.kc + i]; iv = a.top_ids[(size_t)q * a.kc + i]; }
        else if ((uint32_t)i - have < nc) { sv = a.cand_sc[(size_t)q * a.cap + (i - have)]; iv = a.cand_id[(size_t)q * a.cap + (i - have)]; }
        ps[i] = sv;
        pi[i] = iv;
    }
    // bitonic sort of the pool by (score, id)
    for (int size = 2; size <= pool; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < pool / 2; t += 256) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const float sl = ps[lo], sh = ps[hi];
                const uint32_t il = pi[lo], ih = pi[hi];
                const bool lt = sh < sl || (sh == sl && ih < il), gt = sl < sh || (sl == sh && il < ih);
                if (up ? lt : gt) { ps[lo] = sh; ps[hi] = sl; pi[lo] = ih; pi[hi] = il; }
            }
        }
    __syncthreads();
    const uint32_t tot = have + nc;
    const uint32_t keep = tot < a.kc ? tot : a.kc;
    for (uint32_t t = (uint32_t)tid; t < keep; t += 256) {
        a.top_scores[(size_t)q * a.kc + t] = ps[t];
        a.top_ids[(size_t)q * a.kc + t] = pi[t];
    }
    if (tid == 0) {
        a.top_counts[q] = keep;
        a.cand_cnt[q] = 0u;
        // a rejected query lets nothing through (its scores are not reported: domain.rs:113-157 failed it before any index work)
        a.thr[q] = a.qstatus[q] != 0u ? 0.f : (keep >= a.kc ? ps[a.kc - 1] : inf);
    }
}

// ---- exact re-rank: one wavefront per query, candidates scored in the reference's summation order ----
struct RerankArgs {
    DevIndex ix;
    const float *queries;     // [b][dim] f32
    const uint32_t *qstatus;  // [b]
    const float *qhdr;        // [b]
    const float *qn2;         // [b]
    const float *cand_scores; // [b][kc] approximate scores, ascending
    const uint32_t *cand_ids; // [b][kc] internal ids
    const uint32_t *cand_counts; // [b]
    uint32_t kc, k, m;        // kc = m + 1 slots; results k
    float xmax2;              // max |x|^2 over the rows
    float extra_rel;          // added to the certificate's relative bound: the residual terms a one-pass contraction dropped
    uint64_t *out_ids;
    float *out_scores;
    uint32_t *out_counts, *out_status;
    uint32_t *cert;           // [b] 1 = proven exact, 0 = needs a wider candidate set
    // small-batch path (hvx_flat_smallb.hip): instead of one sorted candidate list per query, sl_n unsorted (score, row)
    // pairs -- every slice's kc smallest, +inf padded; the kernel takes their kc smallest itself (bitwise descent, no sort)
    const float *sl_sc;       // [b][sl_stride] or NULL
    const uint32_t *sl_id;
    uint32_t sl_n, sl_stride;
};

// P fp8 rows per 8-lane group scored in the reference's summation order: NK/4 16-byte loads per lane and row
// (the lane's four virtual lanes of four consecutive chunks), codes widened by v_cvt_pk_f32_fp8 and scaled by
// the row's f32 scale -- exactly the dequantised value the import validated.
template <uint32_t METRIC, int NK, int P>
__device__ __forceinline__ void score_rows_fp8(const DevIndex &ix, const float *qs, const uint32_t (&node)[P], int slot,
                                               float qhdr, const float *qglobal, float (&out)[P]) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int NL = NK / 4;
    uint4 x[P][NL];
    float sc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const uint4 *rp = reinterpret_cast<const uint4 *>(ix.vec8 + (size_t)node[p] * ix.dim) + slot;
#pragma unroll
        for (int m = 0; m < NL; ++m) x[p][m] = rp[m * 8];
        sc[p] = ix.rowscale[node[p]];
    }
    __builtin_amdgcn_sched_barrier(0);
    float4 acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 *qp = reinterpret_cast<const float4 *>(qs) + slot;
#pragma unroll
    for (int m = 0; m < NL; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 qq = qp[(4 * m + c) * 8];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const uint32_t w = c == 0 ? x[p][m].x : (c == 1 ? x[p][m].y : (c == 2 ? x[p][m].z : x[p][m].w));
                const f2 a01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false), a23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
                fma_chunk<METRIC>(acc[p], qq, make_float4(sc[p] * a01[0], sc[p] * a01[1], sc[p] * a23[0], sc[p] * a23[1]));
            }
        }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        float r = avx_tree_reduce(acc[p]);
        if (METRIC == kCosine) {
            const uint32_t nd = node[p];
            r = cosine_finish_fn(r, qhdr, ix.hdr[nd], [&]() {
                const uint8_t *rb = ix.vec8 + (size_t)nd * ix.dim;
                const float s = ix.rowscale[nd];
                return stable_half_cosine_fn(ix.dim, [&](uint32_t i) { return qglobal[i]; },
                                             [&](uint32_t i) { return s * fp8_e4m3_decode(rb[fp8_slot_of(i)]); });
            });
        }
        out[p] = r;
    }
}

// Four wavefronts per query (one per SIMD, each with its SIMD's whole register file): every wavefront re-scores a quarter of
// the candidates with the HNSW kernel's reference-order gather / FMA code, 16 rows per pass; selection (small-batch path),
// the final sort and the certificate are cooperative.
template <uint32_t METRIC, int NK, int KIND> // KIND as in flat_mfma_bf16_kernel: 0 bf16, 1 fp8, 2 f32 rows
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rerank_bf16_kernel(RerankArgs a) {
    constexpr bool FP8 = KIND == 1, BFR = KIND != 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t cnt3[3], s_valid, s_bad, s_nout;
    __shared__ __attribute__((aligned(16))) uint32_t rhist[4][256]; // small-batch path: one histogram per 8-bit selection round
    constexpr int P = 2; // 16 rows per wavefront and pass
    const DevIndex &ix = a.ix;
    const uint32_t q = blockIdx.x;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 3, j = lane & 7, slot = chunk_slot(j);
    float *qs = reinterpret_cast<float *>(smem);                     // [dim]
    float *ss = qs + (size_t)NK * 32;                                // [1024] exact scores
    uint32_t *si = reinterpret_cast<uint32_t *>(ss + 1024);          // [1024] ids
    const uint32_t st = a.qstatus[q];
    if (st != 0u) {
        if (tid == 0) { a.out_counts[q] = 0; if (a.out_status) a.out_status[q] = st; a.cert[q] = 1u; }
        return;
    }
    const float *qglobal = a.queries + (size_t)q * ix.dim;
    for (uint32_t i = (uint32_t)tid; i < (uint32_t)NK * 8u; i += 256)
        reinterpret_cast<float4 *>(qs)[i] = reinterpret_cast<const float4 *>(qglobal)[i];
    const float inf = __uint_as_float(0x7F800000u);
    uint32_t nc;
    float t_thr = inf; // approximate score of the last candidate: every row that is not re-scored has at least this score
    if (tid == 0) { cnt3[0] = cnt3[1] = cnt3[2] = 0; s_valid = 0; s_bad = 0; s_nout = 0; }
    for (uint32_t i = (uint32_t)tid; i < 1024u; i += 256) { ss[i] = inf; si[i] = 0xFFFFFFFFu; }
#pragma unroll
    for (int d = 0; d < 4; ++d) rhist[d][tid] = 0;
    __syncthreads();
    if (a.sl_sc) {
        // the kc smallest of the slices' pairs: keys staged in LDS, kth key by a cooperative bitwise descent (ballot counts)
        uint32_t *sk = si + 1024; // [sl_n]
        const float *gs = a.sl_sc + (size_t)q * a.sl_stride;
        const uint32_t *gi = a.sl_id + (size_t)q * a.sl_stride;
        uint32_t valid = 0;
        for (uint32_t i0 = 0; i0 < a.sl_n; i0 += 256) {
            const uint32_t i = i0 + (uint32_t)tid;
            uint32_t kb = 0xFFFFFFFFu;
            if (i < a.sl_n) {
                kb = gi[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : __float_as_uint(gs[i]);
                sk[i] = kb;
            }
            valid += (uint32_t)__builtin_popcountll(__ballot(kb != 0xFFFFFFFFu));
        }
        if (lane == 0 && valid) atomicAdd(&s_valid, valid);
        __syncthreads();
        const uint32_t kth = a.kc < s_valid ? a.kc : s_valid;
        // four rounds of 8 bits (hvx_flat_mfma.h: radix_digit_of_rank), one barrier each; the histograms were zeroed above
        uint32_t prefix = 0, kk = kth, less = 0;
#pragma unroll
        for (int d = 3; d >= 0; --d) {
            const int sh = 8 * d;
            const uint32_t hi_mask = d == 3 ? 0u : (0xFFFFFFFFu << (sh + 8));
            for (uint32_t i0 = 0; i0 < a.sl_n; i0 += 256) { // (whole wavefronts stay in the loop: radix_count votes)
                const uint32_t i = i0 + (uint32_t)tid;
                const uint32_t kb = i < a.sl_n ? sk[i] : 0xFFFFFFFFu;
                radix_count(rhist[d], i < a.sl_n && (kb & hi_mask) == prefix, (kb >> sh) & 255u);
            }
            __syncthreads();
            uint32_t g = 0, below = 0;
            if (kk) radix_digit_of_rank(rhist[d], kk, g, below);
            prefix |= g << sh; kk -= below; less += below;
        }
        // candidates: keys below the kth value, then ties up to the quota (which ties does not matter: see hvx_flat_smallb.hip);
        // cnt3[0] (zero since the kernel's start) counts the ties
        const uint32_t quota = kth - less;
        for (uint32_t i0 = 0; i0 < a.sl_n && kth; i0 += 256) {
            const uint32_t i = i0 + (uint32_t)tid;
            const uint32_t kb = i < a.sl_n ? sk[i] : 0xFFFFFFFFu;
            bool take = kb < prefix;
            if (!take && kb == prefix && kb != 0xFFFFFFFFu) take = atomicAdd(&cnt3[0], 1u) < quota;
            if (take) {
                const uint32_t pos = atomicAdd(&s_nout, 1u);
                if (pos < 1024u) si[pos] = gi[i];
            }
        }
        __syncthreads();
        nc = s_nout < 1024u ? s_nout : 1024u;
        t_thr = kth ? __uint_as_float(prefix) : inf;
    } else {
        nc = a.cand_counts[q];
        for (uint32_t i = (uint32_t)tid; i < nc && i < 1024u; i += 256) si[i] = a.cand_ids[(size_t)q * a.kc + i];
        if (nc) t_thr = a.cand_scores[(size_t)q * a.kc + (nc - 1)];
    }
    __syncthreads();
    const float qhdr = a.qhdr[q];
    bool bad = false;
    for (uint32_t f0 = (uint32_t)wave * 8u * P; f0 < nc; f0 += 4u * 8u * P) {
        uint32_t nd[P];
        float o[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t f = f0 + (uint32_t)(p * 8 + grp);
            nd[p] = si[f < nc ? f : f0];
        }
        if (FP8) {
            score_rows_fp8<METRIC, NK, P>(ix, qs, nd, slot, qhdr, qglobal, o);
        } else {
            Gather<NK, P, BFR> g;
            gather_issue<NK, P, BFR, P, METRIC == kCosine>(ix, nd, slot, g);
            gather_consume<METRIC, NK, P, BFR>(ix, qs, g, nd, slot, qhdr, qglobal, o);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t f = f0 + (uint32_t)(p * 8 + grp);
            float d = o[p];
            if (f < nc) {
                if (!score_valid(d)) bad = true;
                if (j == 0) ss[f] = d;
            }
        }
    }
    if (__ballot(bad) && lane == 0) s_bad = 1;
    __syncthreads();
    if (s_bad) { // Candidate::try_new rejects the score (model.rs:21-29)
        if (tid == 0) { a.out_counts[q] = 0; if (a.out_status) a.out_status[q] = 8u; a.cert[q] = 1u; }
        return;
    }
    // sort of the (score, id) pairs.  Up to 64 candidates (k <= 31: m + 1 = 64): one pair per lane of ONE wavefront, bitonic
    // exchanges by lane shuffles -- no LDS traffic, no barriers (21 exchange steps against 28 barrier-separated LDS passes)
    if (nc <= 64u) {
        if (wave == 0) {
            float sv = ss[lane];
            uint32_t iv = si[lane]; // (entries past nc hold +inf / 0xFFFFFFFF)
#pragma unroll
            for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    const float os = __shfl_xor(sv, stride, 64);
                    const uint32_t oi = __shfl_xor(iv, stride, 64);
                    const bool lower = (lane & stride) == 0, up = (lane & size) == 0;
                    const bool other_first = os < sv || (os == sv && oi < iv), mine_first = sv < os || (sv == os && iv < oi);
                    if ((lower == up) ? other_first : mine_first) { sv = os; iv = oi; }
                }
            ss[lane] = sv;
            si[lane] = iv;
        }
        __syncthreads();
    }
    // ... more: bitonic sort in LDS over the next power of two above the candidate count (the rest is +inf)
    int npairs = 2;
    while ((uint32_t)npairs < nc) npairs <<= 1;
    for (int size = 2; size <= npairs && nc > 64u; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < npairs / 2; t += 256) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const float sl = ss[lo], sh = ss[hi];
                const uint32_t il = si[lo], ih = si[hi];
                const bool lt = sh < sl || (sh == sl && ih < il); // hi before lo
                const bool gt = sl < sh || (sl == sh && il < ih);
                if (up ? lt : gt) { ss[lo] = sh; ss[hi] = sl; si[lo] = ih; si[hi] = il; }
            }
            __syncthreads();
        }
    const uint32_t outn = nc < a.k ? nc : a.k;
    for (uint32_t t = (uint32_t)tid; t < outn; t += 256) {
        a.out_ids[(size_t)q * a.k + t] = ix.ids[si[t]];
        a.out_scores[(size_t)q * a.k + t] = ss[t];
    }
    if (tid == 0) {
        a.out_counts[q] = outn;
        if (a.out_status) a.out_status[q] = 0u;
        // certificate (see the file header).  nc <= m: every row of the scan was re-scored.
        uint32_t ok = 1u;
        if (nc > a.m) {
            const float t = t_thr;
            // worst-case |approximate - reference-order score| relative to (|q|^2 + |x|^2)/2 (L2) resp. absolute (cosine):
            // hi/lo split residuals (<= 2^-17 each way, the dropped lo.lo term and the f32 roundings of the norms:
            // 2e-5 in all) + f32 accumulation over K = dim terms in BOTH summation orders (6 K 2^-24 worst case),
            // the latter doubled for whatever order the matrix core accumulates in.  K = 1536 -> 1.1e-3, K = 128 -> 1.1e-4.
            const float erel = 2.0e-5f + 12.0f * (float)(NK * 32) * 5.9604645e-8f + a.extra_rel;
            const float e = METRIC == kL2 ? erel * 0.5f * (a.qn2[q] + a.xmax2) : erel;
            const float kth = outn ? ss[outn - 1] : inf;
            ok = (outn == a.k && kth < t - e) ? 1u : 0u;
        }
        a.cert[q] = ok;
    }
}

template <uint32_t METRIC>
static hipError_t launch_rerank(const RerankArgs &a, uint32_t b, hipStream_t s) {
    const size_t lds = (size_t)a.ix.dim * 4 + 8192 + (a.sl_sc ? (size_t)a.sl_n * 4 : 0);
    switch (a.ix.dim >> 5) {
#define HVX_RR(N)                                                                                                   \
    case N:                                                                                                         \
        if (a.ix.dtype == HVX_FP8_E4M3) hipLaunchKernelGGL((rerank_bf16_kernel<METRIC, N, 1>), dim3(b), dim3(256), lds, s, a); \
        else if (a.ix.dtype == HVX_F32) hipLaunchKernelGGL((rerank_bf16_kernel<METRIC, N, 2>), dim3(b), dim3(256), lds, s, a);  \
        else hipLaunchKernelGGL((rerank_bf16_kernel<METRIC, N, 0>), dim3(b), dim3(256), lds, s, a);                  \
        break;
        HVX_RR(4) HVX_RR(8) HVX_RR(16) HVX_RR(24) HVX_RR(32) HVX_RR(48)
#undef HVX_RR
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_bf16_row_norm2(const uint16_t *rows, uint32_t n, uint32_t dim, float *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(bf16_row_norm2_kernel, dim3(n), dim3(64), 0, s, rows, n, dim, out);
    return hipGetLastError();
}

static int flat_mfma_impl(hvx_index *ix, const float *d_queries, uint32_t b, uint32_t k, const uint32_t *d_subset, uint32_t n_rows,
                          uint64_t *d_ids, float *d_scores, uint32_t *d_counts, uint32_t *d_status, bool timed, bool allow_filter);

// scan all rows of a bf16 index for b device-resident queries (see the file header)
int flat_mfma_device(hvx_index *ix, const float *d_queries, uint32_t b, uint32_t k, const uint32_t *d_subset, uint32_t n_rows,
                     uint64_t *d_ids, float *d_scores, uint32_t *d_counts, uint32_t *d_status, bool timed) {
    return flat_mfma_impl(ix, d_queries, b, k, d_subset, n_rows, d_ids, d_scores, d_counts, d_status, timed, true);
}

// bf16 shadow of an f32 index's rows (plain order, RNE: the values the one-pass contraction rounds to anyway): built once per
// IMAGE -- the imported handle and its forks share it -- by the first scan that reaches a filtered slice; +50 % of the row
// bytes, so an image that cannot afford it stays on the 128 x 128 kernel (which splits the f32 rows on the fly).  The builder
// waits for its conversion kernel before publishing the pointer: another lane may use it from its own stream right away.
static int ensure_shadow(hvx_index *ix) {
    const DevIndex &d = ix->dev;
    hvx_image_shared &sh = *ix->shared;
    std::lock_guard<std::mutex> lock(sh.mu);
    if (sh.shadow_failed) { ix->m_shadow = nullptr; return HVX_OK; }
    if (!sh.shadow) { // sized for every row the image can ever hold (hvx_build_params.reserve_rows): appended rows are converted below
        void *p = nullptr;
        const size_t rows_cap = std::max<uint64_t>(ix->cap_rows, d.n);
        if (hipMalloc(&p, std::max<size_t>(rows_cap * d.dim * 2, 16)) != hipSuccess) {
            (void)hipGetLastError();
            sh.shadow_failed = true;
            return HVX_OK;
        }
        sh.device = ix->device;
        sh.shadow = reinterpret_cast<uint16_t *>(p);
        sh.shadow_rows = 0;
    }
    if (sh.shadow_rows < d.n) { // the rows that are new since the shadow was last extended (all of them the first time)
        const size_t r0 = sh.shadow_rows, cnt = d.n - r0;
        hipLaunchKernelGGL(bf16_shadow_kernel, dim3(2048), dim3(256), 0, ix->stream, d.vec + r0 * d.ld, cnt * d.dim / 4, sh.shadow + r0 * d.dim);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ix->stream) != hipSuccess) return fail(HVX_ERR_DEVICE, "bf16 shadow conversion failed");
        sh.shadow_rows = d.n;
    }
    ix->m_shadow = sh.shadow;
    return HVX_OK;
}

static int flat_mfma_impl(hvx_index *ix, const float *d_queries, uint32_t b, uint32_t k, const uint32_t *d_subset, uint32_t n_rows,
                          uint64_t *d_ids, float *d_scores, uint32_t *d_counts, uint32_t *d_status, bool timed, bool allow_filter) {
    const DevIndex &d = ix->dev;
    const uint32_t n = n_rows; // rows of the scan: the whole index, or the restricted row list
    if (k == 0) return fail(HVX_ERR_K_RANGE, "result count must be non-zero");
    if (k > 511) return fail(HVX_ERR_UNSUPPORTED, "bf16 exact scan supports k <= 511");
    const uint32_t nk = d.dim >> 5;
    if (nk != 4 && nk != 8 && nk != 16 && nk != 24 && nk != 32 && nk != 48)
        return fail(HVX_ERR_UNSUPPORTED, "bf16 / fp8 exact scan serves dim in {128,256,512,768,1024,1536}");
    HIP_TRY(launch_validate_queries(d, d_queries, b, ix->limit, ix->d_qstatus, ix->d_qhdr, ix->stream));
    if (n == 0) {
        if (timed) { // the caller reads ev0 -> ev1: an empty scan is a zero-length interval, not a stale one
            HIP_TRY(hipEventRecord(ix->ev0, ix->stream));
            HIP_TRY(hipEventRecord(ix->ev1, ix->stream));
        }
        HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)b * 4, ix->stream));
        if (d_status) HIP_TRY(hipMemcpyAsync(d_status, ix->d_qstatus, (size_t)b * 4, hipMemcpyDeviceToDevice, ix->stream));
        return HVX_OK;
    }
    const uint32_t bpad = (b + 255u) / 256u * 256u; // a multiple of both kernels' query tiles (128 and 256)
    int rc;
    const bool fp8 = d.dtype == HVX_FP8_E4M3, f32 = d.dtype == HVX_F32;
    if ((size_t)bpad * d.dim > ix->cap_qsplit) {
        if ((rc = ix->regrow((void **)&ix->m_qhi, (size_t)bpad * d.dim * 2))) return rc;
        if ((rc = ix->regrow((void **)&ix->m_qlo, (size_t)bpad * d.dim * 2))) return rc;
        if (fp8 && (rc = ix->regrow((void **)&ix->m_qhi8, (size_t)bpad * d.dim * 2))) return rc;
        if ((rc = ix->regrow((void **)&ix->m_qn2, (size_t)bpad * 4))) return rc;
        if ((rc = ix->regrow((void **)&ix->m_cert, (size_t)bpad * 4))) return rc;
        ix->cap_qsplit = (size_t)bpad * d.dim;
    }
    // the 256 x 256 filtered contraction (hvx_flat_tile.hip) serves the one-pass attempt of scans with dim % 64 == 0
    // (its query tile is 256 wide: a batch of <= 128 queries wastes less on the 128 x 128 kernel)
    const bool tile_ok = d.dim % 64u == 0u && (b > 128u || ix->opt[HVX_OPT_FLAT_FIRST_CHUNK]) && !ix->opt[HVX_OPT_FLAT_NO_TILE];
    const bool rows_bf16 = d.dtype == HVX_BF16; // (round 6: bf16 images take upserts too; their norms were computed at import)
    if ((f32 || rows_bf16) && ix->m_rowterm) { // a vector that changed in place (an upsert into the node's own slot) invalidates the cached norms
        uint64_t epoch;
        { std::lock_guard<std::mutex> g(ix->shared->mu); epoch = ix->shared->vec_epoch; }
        if (epoch != ix->rowterm_epoch) { ix->rowterm_rows = 0; ix->m_xmax2 = 0.f; ix->rowterm_epoch = epoch; }
    }
    if ((f32 || rows_bf16) && (!ix->m_rowterm || ix->rowterm_rows < d.n)) { // |x|^2 per row and its maximum: on first use, and for rows appended since
        if (!ix->m_rowterm) {
            if ((rc = ix->dalloc((void **)&ix->m_rowterm, std::max<size_t>(std::max<uint64_t>(ix->cap_rows, d.n), 1) * 4))) return rc;
            ix->rowterm_rows = 0;
            ix->m_xmax2 = 0.f;
            { std::lock_guard<std::mutex> g(ix->shared->mu); ix->rowterm_epoch = ix->shared->vec_epoch; }
        }
        const uint32_t r0 = ix->rowterm_rows, cnt = d.n - r0;
        std::vector<float> h_n2(cnt);
        if (rows_bf16) HIP_TRY(launch_bf16_row_norm2(d.vecb + (size_t)r0 * d.dim, cnt, d.dim, ix->m_rowterm + r0, ix->stream));
        else HIP_TRY(launch_f32_row_norm2(d.vec + (size_t)r0 * d.ld, cnt, d.ld, d.dim, ix->m_rowterm + r0, ix->stream));
        HIP_TRY(hipMemcpyAsync(h_n2.data(), ix->m_rowterm + r0, (size_t)cnt * 4, hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream));
        for (float v : h_n2) ix->m_xmax2 = std::max(ix->m_xmax2, v);
        ix->rowterm_rows = d.n;
    }
    hipLaunchKernelGGL(split_queries_kernel, dim3(bpad), dim3(64), 0, ix->stream, d_queries, b, bpad, d.dim, ix->m_qhi, ix->m_qlo, ix->m_qn2,
                       f32 ? 2u : (fp8 ? 1u : 0u), fp8 && tile_ok ? ix->m_qhi8 : nullptr);
    HIP_TRY(hipGetLastError());
    // fp8 rows take the MX-scaled fp8 build of the tile kernel unless another build is asked for (4 = the bf16-widening two-workgroup build)
    const bool mx_build = fp8 && tile_ok && (ix->opt[HVX_OPT_FLAT_TILE_BUILD] == 3u || ix->opt[HVX_OPT_FLAT_TILE_BUILD] == 0u) && d.dim % 128u == 0u;
    if (mx_build) {
        if ((size_t)bpad * d.dim > ix->cap_qmx) {
            if ((rc = ix->regrow((void **)&ix->m_qmx, (size_t)bpad * d.dim * 2))) return rc;
            if ((rc = ix->regrow((void **)&ix->m_qexp, (size_t)bpad * 4))) return rc;
            ix->cap_qmx = (size_t)bpad * d.dim;
        }
        hipLaunchKernelGGL(split_queries_mx_kernel, dim3(bpad), dim3(64), 0, ix->stream, d_queries, b, bpad, d.dim, ix->m_qmx, ix->m_qexp);
        HIP_TRY(hipGetLastError());
    }
    if (timed) HIP_TRY(hipEventRecord(ix->ev0, ix->stream));
    // Attempts, cheapest first; a query whose certificate is not reached sends the batch to the next one:
    //   0. ONE-pass contraction (q_hi.x_hi only; the dropped residual terms widen the certificate's bound), m = max(63, 2k)
    //   1. the full split (bf16 / fp8 rows: + q_lo.x; f32 rows: + q_lo.x_hi + q_hi.x_lo), same m
    //   2. the full split with m = 1023 (f32 rows: the exact VALU scan for the failing queries instead)
    // With m + 1 <= 256 only the first chunk writes its score matrix; every later launch covers a geometrically growing
    // slice of the rows and lets a score out of the tile only below the query's running threshold (FILT).
    const uint32_t m0 = std::max<uint32_t>(63u, 2u * k); // (63: m + 1 = 64 candidates are ONE pass of the re-rank kernel's four wavefronts x 16 rows, and one wavefront's sort)
    const bool debug = tuning_env("HVX_FLAT_DEBUG") != nullptr;
    const bool allow_fast = !ix->opt[HVX_OPT_FLAT_NO_FAST];
    const bool no_filter = ix->opt[HVX_OPT_FLAT_NO_FILTER] != 0;
    if (allow_filter) ix->last_scan_path = 0; // (the unfiltered repeat after a pair overflow keeps the first pass's flags)
    constexpr uint32_t kCandCap = 1024;
    // a handle whose one-pass attempt keeps missing its certificate (tightly clustered scores: the widened bound exceeds the gap
    // between the k-th and the (m+1)-th score) starts with the full split; the one-pass attempt is probed again every 32 scans
    const bool skip_fast = ix->m_fast_misses >= 2 && (ix->m_fast_skipped++ % 32u) != 31u;
    // small batches with the one-launch exact tail (hvx_flat_tail.hip) always take the one-pass contraction: a loose bound only re-scores
    // more rows there, it never sends the scan round again
    const bool tail_first = !fp8 && !ix->opt[HVX_OPT_FLAT_NO_TAIL] && ix->opt[HVX_OPT_FLAT_NO_SMALLB] != 1u && flat_tail_supported(ix, b, k, n) &&
                            flat_smallb_supported(d.dim, b, f32 ? 2 : 0) && (size_t)((n + 3u) & ~3u) * b * 4 <= (512u << 20);
    for (int attempt = (allow_fast && (!skip_fast || tail_first)) ? 0 : 1; attempt < 3; ++attempt) {
        const bool full = attempt >= 1;
        const uint32_t m = attempt == 2 ? 1023u : m0;
        if (attempt == 2 && m0 >= 1023u) break;
        const uint32_t kc = m + 1;
        // small batches (b <= 128): ONE pass over the rows with the register-resident kernel of hvx_flat_smallb.hip writes the
        // whole [b][n] score matrix; sliced selection + pair merge replace the chunk loop.  One-pass attempt only.
        // (a shadow that exists may be behind the rows: appended since, or a vector replaced in place by an upsert -- it catches up here)
        if (f32 && ix->m_shadow && !full && (rc = ensure_shadow(ix))) return rc;
        const int sb_kind = f32 ? ((ix->m_shadow && !full) ? 0 : 2) : (fp8 ? 1 : 0); // (the shadow has no lo parts: the full split reads the f32 rows)
        const bool smallb = ix->opt[HVX_OPT_FLAT_NO_SMALLB] != 1u && flat_smallb_supported(d.dim, b, sb_kind) &&
                            (size_t)((n + 3u) & ~3u) * b * 4 <= (512u << 20) && kc <= 1024u;
        constexpr uint32_t kSmallbCandCap = 8192; // 32 slices x kc <= 256 pairs per query
        const bool sb_radix = smallb && kc <= 256u; // (kc > 256: the widened attempt of bf16 rows keeps the sorted-pool selection)
        uint32_t sb_slices = 0;
        // first chunk (scored by the 128 x 128 kernel into the score matrix, top-(m + 1) selected from it): 16 384 rows when the
        // large-tile filtered slices follow (1024 x 1M x 768: 2.43 ms vs 2.59 with 65 536), 65 536 as before otherwise
        uint32_t chunk = tile_ok && !full && allow_filter && m + 1 <= 256u && !no_filter ? 16384u : 65536u;
        if (ix->opt[HVX_OPT_FLAT_FIRST_CHUNK]) chunk = std::max<uint32_t>(1024u, ix->opt[HVX_OPT_FLAT_FIRST_CHUNK] / 1024u * 1024u); // tests: small first chunks
        while ((size_t)chunk * b * 4 > (512u << 20) && chunk > 1024) chunk >>= 1;
        if (chunk > n || smallb) chunk = (n + 3u) & ~3u;
        if ((rc = ix->flat_scratch(b, kc, chunk))) return rc;
        const bool filt = allow_filter && kc <= 256u && n > chunk && !no_filter;
        const size_t cand_need = (size_t)bpad * (smallb ? kSmallbCandCap : kCandCap);
        if ((filt || smallb) && ix->cap_cand < cand_need) { // pair buffers
            if ((rc = ix->regrow((void **)&ix->m_csc, cand_need * 4))) return rc;
            if ((rc = ix->regrow((void **)&ix->m_cid, cand_need * 4))) return rc;
            ix->cap_cand = cand_need;
        }
        if ((filt || smallb) && ix->cap_cand_b < (size_t)bpad) { // per-query thresholds / pair counters (+ the overflow word)
            if ((rc = ix->regrow((void **)&ix->m_thr, (size_t)bpad * 4))) return rc;
            if ((rc = ix->regrow((void **)&ix->m_ccnt, (size_t)bpad * 4 + 4))) return rc;
            ix->cap_cand_b = bpad;
        }
        // (the small-batch path with the sort-free selection uses neither the top lists' counters nor the pair counters: its two
        // kernels hand over through the slice lists -- three fills of ~4 us each were 7 % of a 100 000-candidate scan)
        if (!sb_radix) {
            HIP_TRY(hipMemsetAsync(ix->f_top_c, 0, (size_t)b * 4, ix->stream));
            if (filt || smallb) HIP_TRY(hipMemsetAsync(ix->m_ccnt, 0, (size_t)bpad * 4 + 4, ix->stream));
        }
        FlatArgs fa;
        fa.ix = d; fa.queries = d_queries; fa.qstatus = ix->d_qstatus; fa.qhdr = ix->d_qhdr; fa.subset = d_subset;
        fa.n_rows = n; fa.dist = ix->f_dist; fa.chunk_ld = chunk; fa.b = b; fa.k = kc;
        fa.top_scores = ix->f_top_s; fa.top_ids = ix->f_top_i; fa.top_counts = ix->f_top_c;
        MfmaArgs ma{};
        ma.qhi = ix->m_qhi; ma.qlo = ix->m_qlo; ma.subset = d_subset;
        ma.rows = f32 ? (const void *)d.vec : (fp8 ? (const void *)d.vec8 : (const void *)d.vecb);
        ma.rowscale = d.rowscale;
        ma.rowterm = d.metric == kL2 ? ix->m_rowterm : d.hdr;
        ma.qn2 = ix->m_qn2; ma.dim = d.dim; ma.b = b; ma.metric = d.metric; ma.dist = ix->f_dist; ma.chunk_ld = chunk;
        ma.thr = ix->m_thr; ma.cand_sc = ix->m_csc; ma.cand_id = ix->m_cid; ma.cand_cnt = ix->m_ccnt; ma.cand_cap = kCandCap;
        auto contraction = [&](uint32_t r0, uint32_t rows, bool filtered) -> hipError_t {
            ma.row0 = r0; ma.nrows = rows;
            ma.nq_tiles = (b + kBM - 1) / kBM; // (bpad is a multiple of 256 for the tile kernels: no all-padding 128-query tile here)
            ma.nr_tiles = (rows + kBN - 1) / kBN;
            const size_t row_bytes = (size_t)d.dim * (f32 ? 4 : (fp8 ? 1 : 2));
            ma.group_tiles = (uint32_t)std::max<size_t>(1, std::min<size_t>(ma.nr_tiles, (64u << 20) / ((size_t)kBN * row_bytes)));
            const uint32_t groups = (ma.nr_tiles + ma.group_tiles - 1) / ma.group_tiles;
            const dim3 grid(groups * ma.group_tiles * ma.nq_tiles);
#define HVX_FM(KIND)                                                                                                              \
    do {                                                                                                                          \
        if (full) { if (filtered) hipLaunchKernelGGL((flat_mfma_bf16_kernel<KIND, true, true>), grid, dim3(256), 0, ix->stream, ma);   \
                    else hipLaunchKernelGGL((flat_mfma_bf16_kernel<KIND, true, false>), grid, dim3(256), 0, ix->stream, ma); }          \
        else { if (filtered) hipLaunchKernelGGL((flat_mfma_bf16_kernel<KIND, false, true>), grid, dim3(256), 0, ix->stream, ma);        \
               else hipLaunchKernelGGL((flat_mfma_bf16_kernel<KIND, false, false>), grid, dim3(256), 0, ix->stream, ma); }              \
    } while (0)
            if (f32) HVX_FM(2); else if (fp8) HVX_FM(1); else HVX_FM(0);
#undef HVX_FM
            return hipGetLastError();
        };
        PairMergeArgs pm;
        pm.top_scores = ix->f_top_s; pm.top_ids = ix->f_top_i; pm.top_counts = ix->f_top_c; pm.cand_sc = ix->m_csc; pm.cand_id = ix->m_cid;
        pm.cand_cnt = ix->m_ccnt; pm.thr = ix->m_thr; pm.overflow = ix->m_ccnt ? ix->m_ccnt + bpad : nullptr; pm.qstatus = ix->d_qstatus;
        pm.kc = kc; pm.cap = smallb ? kSmallbCandCap : kCandCap;
        uint32_t r0 = 0;
        bool used_tile = false, used_mx = false;
        if (smallb) {
            MfmaArgs sa = ma;
            sa.row0 = 0; sa.nrows = n;
            if (f32 && sb_kind == 0) sa.rows = ix->m_shadow; // one-pass attempt: the bf16 shadow, once some large scan has built it (half the bytes)
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ix->device);
            sa.dist = ix->f_dist; sa.chunk_ld = chunk;
            HIP_TRY(launch_flat_smallb(sa, sb_kind, full, (uint32_t)cus, ix->opt[HVX_OPT_FLAT_NO_SMALLB] == 2u ? 1u : 0u, ix->stream));
            if (!full && !ix->opt[HVX_OPT_FLAT_NO_TAIL] && flat_tail_supported(ix, b, k, n)) {
                // round 6: the approximate dot products become the exact answer in ONE more launch (hvx_flat_tail.hip): rows whose
                // approximate score could be below the query's k-th exact score are re-scored in the reference's order -- no candidate
                // count, no certificate, nothing to read back, never a second pass over the rows
                const float dropped = f32 ? 0.0078125f : 0.00390625f; // what the one-pass contraction left out (RerankArgs::extra_rel)
                if ((rc = flat_tail_enqueue(ix, d_queries, b, k, ix->f_dist, chunk, n, d_subset, ma.rowterm, ix->m_qn2, dropped, d_ids, d_scores, d_counts, d_status))) return rc;
                if (timed) HIP_TRY(hipEventRecord(ix->ev1, ix->stream));
                ix->last_scan_path |= HVX_PATH_SMALL_BATCH | HVX_PATH_EXACT_TAIL;
                return HVX_OK;
            }
            if (sb_radix) { // sort-free selection: slices' kc smallest pairs, the query's kc smallest of those inside the re-rank kernel
                HIP_TRY(launch_flat_select_radix(sa, kc, ix->d_qstatus, ix->m_csc, reinterpret_cast<uint32_t *>(ix->m_cid), kSmallbCandCap, &sb_slices, ix->stream));
            } else {
                hipLaunchKernelGGL(flat_dots_to_scores_kernel, dim3((n + 255u) / 256u, b), dim3(256), 0, ix->stream, sa);
                fa.row0 = 0; fa.rows = n;
                const uint32_t slices = std::max<uint32_t>(1u, std::min<uint32_t>(31u, 2048u / kc));
                HIP_TRY(launch_flat_select_slices(fa, slices, ix->m_csc, ix->m_cid, ix->m_ccnt, kSmallbCandCap, ix->stream));
                hipLaunchKernelGGL(flat_merge_pairs_kernel, dim3(b), dim3(256), 0, ix->stream, pm);
            }
            HIP_TRY(hipGetLastError());
            r0 = n;
            ix->last_scan_path |= HVX_PATH_SMALL_BATCH;
        }
        while (r0 < n) {
            if (filt && r0 > 0) {
                // thresholds are those of the rows seen so far: a slice three times that long lets ~3 (m + 1) pairs per query through
                const uint32_t rows = (uint32_t)std::min<uint64_t>((uint64_t)n - r0, (uint64_t)r0 * 3u);
                if (!full && tile_ok && f32 && (rc = ensure_shadow(ix))) return rc;
                if (!full && tile_ok && (!f32 || ix->m_shadow)) { // one-pass attempt: the 256 x 256 kernel
                    MfmaArgs ta = ma;
                    ta.row0 = r0; ta.nrows = rows;
                    if (f32) ta.rows = ix->m_shadow;
                    if (fp8) ta.qhi = ix->m_qhi8;
                    if (mx_build) { ta.qhi = reinterpret_cast<const uint16_t *>(ix->m_qmx); ta.qexp = ix->m_qexp; }
                    const uint32_t tb = ix->opt[HVX_OPT_FLAT_TILE_BUILD];
                    HIP_TRY(launch_flat_tile256(ta, fp8 ? 1 : 0, bpad, ix->m_xmax2, pm.overflow, mx_build ? 3u : ((tb == 3u || tb == 4u) ? 0u : tb), ix->stream));
                    used_mx |= mx_build;
                    used_tile = true;
                } else
                    HIP_TRY(contraction(r0, rows, true));
                hipLaunchKernelGGL(flat_merge_pairs_kernel, dim3(b), dim3(256), 0, ix->stream, pm);
                HIP_TRY(hipGetLastError());
                r0 += rows;
            } else {
                const uint32_t rows = std::min(chunk, n - r0);
                HIP_TRY(contraction(r0, rows, false));
                fa.row0 = r0; fa.rows = rows;
                HIP_TRY(launch_flat_select(fa, ix->stream));
                r0 += rows;
                if (filt) { // thresholds out of the first chunk's top list (no pairs yet)
                    hipLaunchKernelGGL(flat_merge_pairs_kernel, dim3(b), dim3(256), 0, ix->stream, pm);
                    HIP_TRY(hipGetLastError());
                }
            }
        }
        RerankArgs ra;
        ra.ix = d; ra.queries = d_queries; ra.qstatus = ix->d_qstatus; ra.qhdr = ix->d_qhdr; ra.qn2 = ix->m_qn2;
        ra.cand_scores = ix->f_top_s; ra.cand_ids = ix->f_top_i; ra.cand_counts = ix->f_top_c; ra.kc = kc; ra.k = k; ra.m = m;
        ra.xmax2 = ix->m_xmax2; ra.out_ids = d_ids; ra.out_scores = d_scores; ra.out_counts = d_counts; ra.out_status = d_status;
        ra.cert = ix->m_cert;
        ra.sl_sc = nullptr; ra.sl_id = nullptr; ra.sl_n = 0; ra.sl_stride = 0;
        if (smallb && sb_radix) { ra.sl_sc = ix->m_csc; ra.sl_id = ix->m_cid; ra.sl_n = sb_slices * kc; ra.sl_stride = kSmallbCandCap; }
        // one rounded operand drops a term <= 2^-9 |q||x| of the dot product = 2^-8 of (|q|^2 + |x|^2)/2 in the L2 score
        // (cosine: <= 2^-10 absolute); f32 rows round BOTH operands in the one-pass build
        ra.extra_rel = full ? 0.f : (f32 ? 0.0078125f : 0.00390625f);
        // the MX build's query is two e4m3 pieces: |q - q^| <= 2^-8 |q| per element (+ subnormal pieces: 2^-18 max |q|) -- twice the bf16 hi part's
        if (used_mx) ra.extra_rel = 0.0079f;
        HIP_TRY(d.metric == kL2 ? launch_rerank<kL2>(ra, b, ix->stream) : launch_rerank<kCosine>(ra, b, ix->stream));
        if (timed) HIP_TRY(hipEventRecord(ix->ev1, ix->stream));
        if ((rc = ix->pin_flags((size_t)b + 1))) return rc; // read back through pinned memory: a pageable copy is a synchronous staged one
        uint32_t *cert = ix->h_flags;
        cert[b] = 0;
        HIP_TRY(hipMemcpyAsync(cert, ix->m_cert, (size_t)b * 4, hipMemcpyDeviceToHost, ix->stream));
        if ((filt || smallb) && !sb_radix) HIP_TRY(hipMemcpyAsync(cert + b, ix->m_ccnt + bpad, 4, hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream));
        const uint32_t overflow = cert[b];
        uint32_t failed = 0, first = 0;
        for (uint32_t i = 0; i < b; ++i)
            if (!cert[i]) { if (!failed) first = i; ++failed; }
        ix->last_scan_path |= (smallb ? 0u : (used_tile ? HVX_PATH_TILE_256 : HVX_PATH_MFMA_128)) | (filt ? HVX_PATH_FILTERED : 0u) | (full ? HVX_PATH_FULL_SPLIT : 0u) |
                              (attempt == 2 ? HVX_PATH_WIDENED : 0u);
        if (debug)
            fprintf(stderr, "[hvx flat] attempt %d (%s contraction, m = %u%s%s): %u of %u certificates missing%s\n", attempt,
                    full ? "full" : "one-pass", m, filt ? ", filtered epilogue" : "", used_tile ? ", 256 x 256 tiles" : "", failed, b,
                    overflow ? ", PAIR OVERFLOW" : "");
        if (overflow) { // a query produced more pairs than the buffer holds: its list is incomplete -- never guess
            ix->last_scan_path |= HVX_PATH_PAIR_OVERFLOW_REPEAT;
            return flat_mfma_impl(ix, d_queries, b, k, d_subset, n_rows, d_ids, d_scores, d_counts, d_status, timed, false);
        }
        if (attempt == 0) ix->m_fast_misses = failed ? ix->m_fast_misses + 1 : 0;
        if (!failed) return HVX_OK;
        if (f32 && full) { // f32 rows have an exact VALU scan to fall back to: never widen, never guess
            ix->m_failed.clear();
            for (uint32_t i = 0; i < b; ++i)
                if (!cert[i]) ix->m_failed.push_back(i);
            return HVX_MFMA_FALLBACK;
        }
        if (attempt == 2 || (attempt == 1 && m0 >= 1023u))
            return fail(HVX_ERR_INVARIANT, "exact-scan certificate failed for %u queries (first %u): more than 1023 rows within the "
                        "error bound of the k-th score", failed, first);
    }
    return fail(HVX_ERR_INVARIANT, "exact-scan certificate failed after widening");
}

} // namespace hvx
