// hvx_device.h -- device-side building blocks shared by the gfx950 kernels.
//
// The central piece is the bit-exact distance evaluator: HelixDB ranks by (f32 score, node id)
// (crates/db/src/search/vector/model.rs:55-61) and the f32 score depends on the summation tree of
// the host SIMD kernel that produced it (spaces/simple.rs:127-143).  The AVX kernels keep 4 x 8
// lane accumulators = 32 "virtual lanes", virtual lane v summing elements v, v+32, v+64, ... in
// order (spaces/simple_avx.rs:128-181).  On CDNA4 eight consecutive lanes of a wavefront (a "row
// group") own one embedding row; lane j carries virtual lanes 4j..4j+3 in a float4 accumulator fed
// by 16-byte global loads, so the per-virtual-lane FMA chain is reproduced exactly, and the final
// (s1+s2)+(s3+s4) / hsum256 tree maps onto three DPP lane exchanges.  Scores therefore match the
// reference CPU path bit for bit, not only in rank.
//
// Built with -ffp-contract=off: every fused multiply-add below is an explicit __builtin_fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hvx {

constexpr uint32_t kSentinel = 0xFFFFFFFFu; // empty slot in a fixed-stride neighbour row
constexpr uint32_t kExpandedBit = 0x80000000u;

enum : uint32_t { kCosine = 0, kL2 = 1, kL1 = 2 };
enum : uint32_t { kKernelScalar = 0, kKernelSse = 1, kKernelAvx = 2, kKernelAvxFma = 3, kKernelNeon = 4 }; // FloatSimd (spaces/simple.rs:45-62)
// the kernels that accumulate with fused multiply-adds (simple_avx.rs:128-238 *_avx_fma; simple_neon.rs vfmaq_f32)
__host__ __device__ __forceinline__ bool kernel_fused(uint32_t fk) { return fk == kKernelAvxFma || fk == kKernelNeon; }
// the 128-bit kernels: four 4-lane accumulators, 16 floats per iteration (simple_sse.rs:17-67, simple_neon.rs:10-95)
__host__ __device__ __forceinline__ bool kernel_w4(uint32_t fk) { return fk == kKernelSse || fk == kKernelNeon; }
// elements covered by the kernel's SIMD loop (the rest is its scalar tail): simple.rs:120-178 dispatch, MIN_DIM_SIZE_AVX = 32 / _SIMD = 16
__host__ __device__ __forceinline__ uint32_t kernel_dim_main(uint32_t fk, uint32_t dim) {
    if (fk == kKernelScalar) return 0u;
    const uint32_t step = kernel_w4(fk) ? 16u : 32u;
    return dim < step ? 0u : dim - dim % step;
}

// Read-only index image in HBM (see DESIGN.md "HBM layout").
struct DevIndex {
    const float *vec;         // [n][ld] row-major f32, rows 16-byte aligned, zero padded to ld (dtype f32)
    const uint16_t *vecb;     // [n][dim] bf16 rows in the interleaved device layout below (dtype bf16)
    const uint8_t *vec8;      // [n][dim] fp8 e4m3fn codes, interleaved (dtype fp8); value = rowscale[row] * decode(code)
    const float *rowscale;    // [n] per-row scale of fp8 rows
    const float *hdr;         // [n] cosine norm header (distance/cosine.rs:73-75); 0 otherwise
    const uint32_t *l0;       // [n][s0] layer-0 rows: internal ids ascending, kSentinel padded
    const uint32_t *up;       // [up_rows][su] upper-layer rows, same format
    const uint32_t *up_base;  // [n] first upper row (layer 1) of a node, kSentinel if level 0
    const uint16_t *level;    // [n] top layer of the node
    const uint64_t *ids;      // [n] external node ids, ascending
    uint32_t n, dim, ld;      // ld = dim rounded up to 4 floats
    uint32_t dim_main;        // elements covered by the 32-lane SIMD tree (0 in scalar mode / dim<32)
    uint32_t s0, su;          // row strides of l0 / up (multiples of 32 / 16)
    uint32_t metric, fkernel;
    uint32_t entry, max_layer, has_entry;
    uint32_t dtype;           // hvx_dtype of the stored rows
    const uint32_t *dead;     // one bit per row: deleted (hvx_index_delete_batch); NULL while nothing has been deleted
};
__device__ __forceinline__ bool row_dead(const DevIndex &ix, uint32_t row) { return ix.dead && ((ix.dead[row >> 5] >> (row & 31u)) & 1u); }

// bf16 device layout (dim % 64 == 0): element i = 32*chunk + 4*slot + e is stored at
// 64*(chunk/2) + 8*slot + 4*(chunk%2) + e, so that the 16 bytes a lane of a row group owns hold its
// four AVX "virtual lanes" of TWO consecutive 32-float chunks: one dwordx4 load per chunk pair, the
// eight lanes of a group still read 128 contiguous bytes.  Dot products (MFMA flat scan) are
// indifferent to the permutation as long as the query operand uses the same one.
__host__ __device__ __forceinline__ uint32_t bf16_slot_of(uint32_t i) {
    const uint32_t k = i >> 5, s = (i >> 2) & 7u, e = i & 3u;
    return (k >> 1) * 64u + s * 8u + (k & 1u) * 4u + e;
}
// fp8 device layout (dim % 128 == 0): element i = 32*chunk + 4*slot + e is stored at
// 128*(chunk/4) + 16*slot + 4*(chunk%4) + e: a lane's 16 bytes hold its four virtual lanes of FOUR chunks.
__host__ __device__ __forceinline__ uint32_t fp8_slot_of(uint32_t i) {
    const uint32_t k = i >> 5, s = (i >> 2) & 7u, e = i & 3u;
    return (k >> 2) * 128u + s * 16u + (k & 3u) * 4u + e;
}
// OCP e4m3fn: 1 sign, 4 exponent (bias 7), 3 mantissa bits; max finite 448; no infinities.
__host__ __device__ __forceinline__ float fp8_e4m3_decode(uint8_t c) {
    const uint32_t m = c & 7u, e = (c >> 3) & 15u;
    float v = e == 0u ? (float)m * 0.001953125f /* m/8 * 2^-6 */ : (1.0f + (float)m * 0.125f) * __builtin_ldexpf(1.0f, (int)e - 7);
    return (c & 0x80u) ? -v : v;
}
// round-to-nearest-even of y (|y| <= 448) to e4m3fn
__host__ __device__ __forceinline__ uint8_t fp8_e4m3_encode(float y) {
    const uint8_t sign = y < 0.0f ? 0x80u : 0u;
    float a = __builtin_fabsf(y);
    if (!(a < 464.0f)) a = 448.0f; // beyond the midpoint to the next (non-existent) step: saturate
    int ex;
    (void)__builtin_frexpf(a, &ex); // a = f * 2^ex, f in [0.5, 1)
    int e = ex - 1;                 // floor(log2(a)) for a > 0
    if (a == 0.0f || e < -6) e = -6; // subnormal range shares the exponent of the smallest normal
    const float step = __builtin_ldexpf(1.0f, e - 3);
    float qv = __builtin_rintf(a / step); // exact division by a power of two; ties to even
    float v = qv * step;
    if (v > 448.0f) v = 448.0f;
    // encode the (exactly representable) value
    if (v == 0.0f) return sign;
    int ex2;
    (void)__builtin_frexpf(v, &ex2);
    int e2 = ex2 - 1;
    uint32_t code;
    if (e2 < -6) code = (uint32_t)(v * 512.0f); // subnormal: m = v / 2^-9
    else code = ((uint32_t)(e2 + 7) << 3) | ((uint32_t)(v * __builtin_ldexpf(1.0f, 3 - e2)) - 8u);
    return (uint8_t)(sign | code);
}

// round-to-nearest-even f32 -> bf16 bits (finite inputs; NaN/inf are rejected before)
__host__ __device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    union { float f; uint32_t u; } v;
    v.f = f;
    const uint32_t lsb = (v.u >> 16) & 1u;
    return (uint16_t)((v.u + 0x7FFFu + lsb) >> 16);
}
__host__ __device__ __forceinline__ float bf16_to_f32(uint16_t h) {
    union { float f; uint32_t u; } v;
    v.u = (uint32_t)h << 16;
    return v.f;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// Lane exchanges of the 8-lane row group as DPP modifiers (no LDS crossbar traffic):
//   quad_perm [1,0,3,2] = lane^1, quad_perm [2,3,0,1] = lane^2, row_half_mirror = lane -> 7-lane.
__device__ __forceinline__ float dpp_xor1(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_half_mirror(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x141, 0xF, 0xF, true));
}

// lane ^ 4 inside the 8-lane row group: half-row mirror (7 - l) followed by a reversal inside the quad
__device__ __forceinline__ float dpp_xor4(float v) {
    const int m = __builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x141, 0xF, 0xF, true);
    return __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp(m, 0x1B, 0xF, 0xF, true));
}

// The 128-bit host kernels (round 5; FloatSimd::Sse on x86 without AVX, FloatSimd::Neon on aarch64 -- the reference's own CI runs on
// ubuntu-24.04-arm): four accumulators of four lanes = 16 virtual lanes, virtual lane v summing elements v, v+16, v+32 ... in order.
// Lane j of a row group owns virtual lanes 2j, 2j+1 (accumulator r = j >> 1, half h = j & 1) in a float2 fed by 8-byte loads -- the
// eight lanes read the 64 contiguous bytes of one iteration.  (sum1+sum2)+(sum3+sum4) is a lane^2 and a lane^4 exchange; the final
// horizontal add is hsum128_ps_sse (simple_sse.rs:10-14: x + movehl, then lane 0 + lane 1) or vaddvq_f32 (pairwise:
// (x0+x1)+(x2+x3), simple_neon.rs:41,83).  NEON == the fused (vfmaq_f32) kernel, SSE == the mul-then-add kernel.
template <bool NEON> __device__ __forceinline__ float w4_tree_reduce(float2 acc) {
    float2 p, t;
    p.x = acc.x + dpp_xor2(acc.x); p.y = acc.y + dpp_xor2(acc.y);   // sum1+sum2 | sum3+sum4
    t.x = p.x + dpp_xor4(p.x); t.y = p.y + dpp_xor4(p.y);           // (sum1+sum2)+(sum3+sum4): half h holds components 2h, 2h+1
    if (NEON) {
        const float pp = t.x + t.y;         // x0+x1 (h = 0) / x2+x3 (h = 1)
        return pp + dpp_xor1(pp);
    }
    const float u0 = t.x + dpp_xor1(t.x);   // x0+x2
    const float u1 = t.y + dpp_xor1(t.y);   // x1+x3
    return u0 + u1;
}
// SIMD part (elements [0, dim_main), dim_main % 16 == 0) of one row against one query, every lane of the group returns the same value
template <uint32_t METRIC, bool FUSED>
__device__ __forceinline__ float w4_main(const float *row, const float *qv, uint32_t dim_main, int j) {
    const uint32_t nb = dim_main >> 4;
    if (nb == 0u) return 0.0f;
    float2 acc = make_float2(0.f, 0.f);
    const float2 *rp = reinterpret_cast<const float2 *>(row) + j;
    const float2 *qp = reinterpret_cast<const float2 *>(qv) + j;
    auto step = [&](const float2 x, const float2 qq) __attribute__((always_inline)) {
        if (METRIC == kL2) {
            const float d0 = qq.x - x.x, d1 = qq.y - x.y;
            if (FUSED) { acc.x = __builtin_fmaf(d0, d0, acc.x); acc.y = __builtin_fmaf(d1, d1, acc.y); }
            else { acc.x = d0 * d0 + acc.x; acc.y = d1 * d1 + acc.y; }
        } else {
            if (FUSED) { acc.x = __builtin_fmaf(qq.x, x.x, acc.x); acc.y = __builtin_fmaf(qq.y, x.y, acc.y); }
            else { acc.x = qq.x * x.x + acc.x; acc.y = qq.y * x.y + acc.y; }
        }
    };
    uint32_t b = 0;
    for (; b + 8 <= nb; b += 8) { // eight independent loads in flight per lane before the first use
        float2 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = rp[(size_t)(b + u) * 8];
#pragma unroll
        for (int u = 0; u < 8; ++u) step(x[u], qp[(b + u) * 8]);
    }
    for (; b < nb; ++b) step(rp[(size_t)b * 8], qp[b * 8]);
    return w4_tree_reduce<FUSED>(acc);
}

// Which float4 of a 32-float chunk lane j of a row group owns.  AVX register r (sum256_{r+1}) covers
// floats 8r..8r+7 = (lo128, hi128).  Lanes 0..3 hold s1lo,s1hi,s2lo,s2hi and lanes 4..7 hold
// s4hi,s4lo,s3hi,s3lo, so that (s1+s2),(s3+s4) is a lane^2 exchange, (s12+s34) a half-row mirror
// and hi128+lo128 a lane^1 exchange -- all three are DPP row operations.
__device__ __forceinline__ int chunk_slot(int j) { return j < 4 ? j : 11 - j; }

// (sum256_1+sum256_2)+(sum256_3+sum256_4), then hsum256_ps_avx (simple_avx.rs:7-12,59-63):
// every lane of the group returns the same value.
__device__ __forceinline__ float avx_tree_reduce(float4 acc) {
    float4 p, t, u;
    p.x = acc.x + dpp_xor2(acc.x); p.y = acc.y + dpp_xor2(acc.y);
    p.z = acc.z + dpp_xor2(acc.z); p.w = acc.w + dpp_xor2(acc.w);
    t.x = p.x + dpp_half_mirror(p.x); t.y = p.y + dpp_half_mirror(p.y);
    t.z = p.z + dpp_half_mirror(p.z); t.w = p.w + dpp_half_mirror(p.w);
    u.x = t.x + dpp_xor1(t.x); u.y = t.y + dpp_xor1(t.y);
    u.z = t.z + dpp_xor1(t.z); u.w = t.w + dpp_xor1(t.w);
    const float x64_0 = u.x + u.z;
    const float x64_1 = u.y + u.w;
    return x64_0 + x64_1;
}

__device__ __forceinline__ bool f32_is_normal(float x) {
    uint32_t e = (__float_as_uint(x) >> 23) & 0xFFu;
    return e != 0u && e != 0xFFu;
}
__device__ __forceinline__ bool f32_is_finite(float x) {
    return ((__float_as_uint(x) >> 23) & 0xFFu) != 0xFFu;
}

// DistanceScore::try_new (parameters.rs:243-274): finite, non-negative, -0 -> +0.
__device__ __forceinline__ bool score_valid(float &s) {
    if (!f32_is_finite(s)) return false;
    if (s < 0.0f) return false;
    if (s == 0.0f) s = 0.0f;
    return true;
}

// distance/cosine.rs:12-36 scaled_l2_norm, serial f64.  `at(i)` yields component i.
template <typename F> __device__ inline double scaled_l2_norm(uint32_t n, F at) {
    double scale = 0.0, scaled_sum = 1.0;
    for (uint32_t i = 0; i < n; ++i) {
        double mag = (double)fabsf(at(i));
        if (mag == 0.0) continue;
        if (scale < mag) {
            double ratio = scale / mag;
            scaled_sum = 1.0 + scaled_sum * ratio * ratio;
            scale = mag;
        } else {
            double ratio = mag / scale;
            scaled_sum += ratio * ratio;
        }
    }
    if (scale == 0.0) return 0.0;
    return scale * sqrt(scaled_sum);
}

// distance/cosine.rs:39-59 stable_half_cosine (f64 fallback; rare: extreme norms only)
__device__ inline float stable_half_cosine(const float *p, const float *q, uint32_t n) {
    double pn = scaled_l2_norm(n, [&](uint32_t i) { return p[i]; });
    double qn = scaled_l2_norm(n, [&](uint32_t i) { return q[i]; });
    if (pn == 0.0 || qn == 0.0) return __uint_as_float(0x7FC00000u);
    double dot = 0.0;
    for (uint32_t i = 0; i < n; ++i) dot += (double)p[i] * (double)q[i];
    double c = dot / (pn * qn);
    if (c < -1.0) c = -1.0;
    if (c > 1.0) c = 1.0;
    return (float)((1.0 - c) * 0.5);
}

// distance/cosine.rs:39-59 over arbitrary component accessors (bf16 rows are dequantised on the fly)
template <typename FP, typename FQ> __device__ inline float stable_half_cosine_fn(uint32_t n, FP pat, FQ qat) {
    double pn = scaled_l2_norm(n, pat);
    double qn = scaled_l2_norm(n, qat);
    if (pn == 0.0 || qn == 0.0) return __uint_as_float(0x7FC00000u);
    double dot = 0.0;
    for (uint32_t i = 0; i < n; ++i) dot += (double)pat(i) * (double)qat(i);
    double c = dot / (pn * qn);
    if (c < -1.0) c = -1.0;
    if (c > 1.0) c = 1.0;
    return (float)((1.0 - c) * 0.5);
}

// distance/cosine.rs:96-118: fast path on cached norms, else the f64 fallback `slow()`
template <typename F> __device__ __forceinline__ float cosine_finish_fn(float pq, float pn, float qn, F slow) {
    float pnqn = pn * qn;
    if (pn > 0.0f && qn > 0.0f && pn != 3.402823466e+38f && qn != 3.402823466e+38f &&
        f32_is_normal(pnqn) && f32_is_finite(pq)) {
        float c = __fdiv_rn(pq, pnqn);
        if (c < -1.0f) c = -1.0f;
        if (c > 1.0f) c = 1.0f;
        return (1.0f - c) / 2.0f;
    }
    return slow();
}
__device__ __forceinline__ float cosine_finish(float pq, float pn, float qn, const float *qv,
                                               const float *row, uint32_t dim) {
    return cosine_finish_fn(pq, pn, qn, [&]() { return stable_half_cosine(qv, row, dim); });
}

// One embedding row scored by the 8 lanes of a row group (all 8 lanes return the same value).
//   qv   : query vector in LDS (ld floats, zero padded)
//   row  : embedding row in HBM
//   j    : lane & 7
// METRIC kL2 / kCosine use the AVX tree for the first dim_main elements and the reference's scalar
// tail for the rest (simple_avx.rs:172-177); kL1 is sequential everywhere (simple.rs:186-202).
template <uint32_t METRIC, bool FUSED, int UNR = 8>
__device__ __forceinline__ float group_distance(const DevIndex &ix, const float *qv, float qhdr,
                                                uint32_t node, int j) {
    const float *row = ix.vec + (size_t)node * ix.ld;
    float result;
    uint32_t t0;
    if (METRIC == kL1) {
        result = 0.0f;
        t0 = 0;
    } else if (kernel_w4(ix.fkernel)) { // SSE / NEON host kernels (FUSED == NEON by construction: kernel_fused)
        t0 = ix.dim_main;
        result = w4_main<METRIC, FUSED>(row, qv, t0, j);
    } else {
        t0 = ix.dim_main;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const uint32_t nk = t0 >> 5;
        const int slot = chunk_slot(j);
        const float4 *rp = reinterpret_cast<const float4 *>(row) + slot;
        const float4 *qp = reinterpret_cast<const float4 *>(qv) + slot;
        uint32_t k = 0;
        // UNR (8) independent 16-byte loads in flight per lane (8 KiB per wavefront) before the first use; callers with 16 wavefronts per
        // workgroup (128 registers each) ask for 4
        for (; k + UNR <= nk; k += UNR) {
            float4 x[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) x[u] = rp[(size_t)(k + u) * 8];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                float4 qq = qp[(k + u) * 8];
                if (METRIC == kL2) {
                    float d0 = qq.x - x[u].x, d1 = qq.y - x[u].y, d2 = qq.z - x[u].z, d3 = qq.w - x[u].w;
                    if (FUSED) {
                        acc.x = __builtin_fmaf(d0, d0, acc.x); acc.y = __builtin_fmaf(d1, d1, acc.y);
                        acc.z = __builtin_fmaf(d2, d2, acc.z); acc.w = __builtin_fmaf(d3, d3, acc.w);
                    } else {
                        acc.x = d0 * d0 + acc.x; acc.y = d1 * d1 + acc.y;
                        acc.z = d2 * d2 + acc.z; acc.w = d3 * d3 + acc.w;
                    }
                } else {
                    if (FUSED) {
                        acc.x = __builtin_fmaf(qq.x, x[u].x, acc.x); acc.y = __builtin_fmaf(qq.y, x[u].y, acc.y);
                        acc.z = __builtin_fmaf(qq.z, x[u].z, acc.z); acc.w = __builtin_fmaf(qq.w, x[u].w, acc.w);
                    } else {
                        acc.x = qq.x * x[u].x + acc.x; acc.y = qq.y * x[u].y + acc.y;
                        acc.z = qq.z * x[u].z + acc.z; acc.w = qq.w * x[u].w + acc.w;
                    }
                }
            }
        }
        for (; k < nk; ++k) {
            float4 x = rp[(size_t)k * 8];
            float4 qq = qp[k * 8];
            if (METRIC == kL2) {
                float d0 = qq.x - x.x, d1 = qq.y - x.y, d2 = qq.z - x.z, d3 = qq.w - x.w;
                if (FUSED) {
                    acc.x = __builtin_fmaf(d0, d0, acc.x); acc.y = __builtin_fmaf(d1, d1, acc.y);
                    acc.z = __builtin_fmaf(d2, d2, acc.z); acc.w = __builtin_fmaf(d3, d3, acc.w);
                } else {
                    acc.x = d0 * d0 + acc.x; acc.y = d1 * d1 + acc.y;
                    acc.z = d2 * d2 + acc.z; acc.w = d3 * d3 + acc.w;
                }
            } else {
                if (FUSED) {
                    acc.x = __builtin_fmaf(qq.x, x.x, acc.x); acc.y = __builtin_fmaf(qq.y, x.y, acc.y);
                    acc.z = __builtin_fmaf(qq.z, x.z, acc.z); acc.w = __builtin_fmaf(qq.w, x.w, acc.w);
                } else {
                    acc.x = qq.x * x.x + acc.x; acc.y = qq.y * x.y + acc.y;
                    acc.z = qq.z * x.z + acc.z; acc.w = qq.w * x.w + acc.w;
                }
            }
        }
        result = nk ? avx_tree_reduce(acc) : 0.0f;
    }
    // sequential part: scalar tail of the SIMD kernels, the whole vector for dim<32 / scalar mode,
    // and all of Manhattan.  Every lane of the group walks the same chain (addresses are
    // group-uniform, so the loads coalesce to one request per row).
    for (uint32_t t = t0; t < ix.dim; ++t) {
        float a = qv[t], b = row[t];
        if (METRIC == kL2) {
            float d = a - b;
            float pr = d * d;
            result += pr;
        } else if (METRIC == kCosine) {
            float pr = a * b;
            result += pr;
        } else {
            result += fabsf(a - b);
        }
    }
    if (METRIC == kCosine) result = cosine_finish(result, qhdr, ix.hdr[node], qv, row, ix.dim);
    return result;
}

// The same distance over a bf16 row (device layout above; dim % 64 == 0, AVX+FMA tree, L2 / cosine -- what the import accepts for
// bf16 rows): the row's values are exact f32 numbers, the arithmetic and its order are those of group_distance, so the score equals
// the reference's on the rounded vector bit for bit.  Used where rows are scored one per 8-lane group outside the unrolled HNSW
// kernels (the restricted walk, round 4).
template <uint32_t METRIC, int UNR = 8>
__device__ __forceinline__ float group_distance_bf16(const DevIndex &ix, const float *qv, float qhdr, uint32_t node, int j) {
    const int slot = chunk_slot(j);
    const uint16_t *rb = ix.vecb + (size_t)node * ix.dim;
    const float4 *rp = reinterpret_cast<const float4 *>(rb) + slot; // one 16-byte piece = this lane's virtual lanes of TWO chunks
    const float4 *qp = reinterpret_cast<const float4 *>(qv) + slot;
    const uint32_t np = ix.dim >> 6;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto step = [&](const float4 xw, uint32_t m) __attribute__((always_inline)) {
        const uint32_t w0 = __float_as_uint(xw.x), w1 = __float_as_uint(xw.y), w2 = __float_as_uint(xw.z), w3 = __float_as_uint(xw.w);
        const float4 q0 = qp[(2u * m) * 8u], q1 = qp[(2u * m + 1u) * 8u];
        const float a0 = __uint_as_float(w0 << 16), a1 = __uint_as_float(w0 & 0xFFFF0000u), a2 = __uint_as_float(w1 << 16), a3 = __uint_as_float(w1 & 0xFFFF0000u);
        const float b0 = __uint_as_float(w2 << 16), b1 = __uint_as_float(w2 & 0xFFFF0000u), b2 = __uint_as_float(w3 << 16), b3 = __uint_as_float(w3 & 0xFFFF0000u);
        if (METRIC == kL2) {
            float d0 = q0.x - a0, d1 = q0.y - a1, d2 = q0.z - a2, d3 = q0.w - a3;
            acc.x = __builtin_fmaf(d0, d0, acc.x); acc.y = __builtin_fmaf(d1, d1, acc.y);
            acc.z = __builtin_fmaf(d2, d2, acc.z); acc.w = __builtin_fmaf(d3, d3, acc.w);
            d0 = q1.x - b0; d1 = q1.y - b1; d2 = q1.z - b2; d3 = q1.w - b3;
            acc.x = __builtin_fmaf(d0, d0, acc.x); acc.y = __builtin_fmaf(d1, d1, acc.y);
            acc.z = __builtin_fmaf(d2, d2, acc.z); acc.w = __builtin_fmaf(d3, d3, acc.w);
        } else {
            acc.x = __builtin_fmaf(q0.x, a0, acc.x); acc.y = __builtin_fmaf(q0.y, a1, acc.y);
            acc.z = __builtin_fmaf(q0.z, a2, acc.z); acc.w = __builtin_fmaf(q0.w, a3, acc.w);
            acc.x = __builtin_fmaf(q1.x, b0, acc.x); acc.y = __builtin_fmaf(q1.y, b1, acc.y);
            acc.z = __builtin_fmaf(q1.z, b2, acc.z); acc.w = __builtin_fmaf(q1.w, b3, acc.w);
        }
    };
    uint32_t m = 0;
    for (; m + UNR <= np; m += UNR) { // UNR (8) independent 16-byte loads in flight per lane before the first use
        float4 x[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) x[u] = rp[(size_t)(m + u) * 8];
#pragma unroll
        for (int u = 0; u < UNR; ++u) step(x[u], m + (uint32_t)u);
    }
    for (; m < np; ++m) step(rp[(size_t)m * 8], m);
    float r = avx_tree_reduce(acc);
    if (METRIC == kCosine)
        r = cosine_finish_fn(r, qhdr, ix.hdr[node], [&]() {
            return stable_half_cosine_fn(ix.dim, [&](uint32_t i) { return qv[i]; }, [&](uint32_t i) { return bf16_to_f32(rb[bf16_slot_of(i)]); });
        });
    return r;
}

} // namespace hvx
