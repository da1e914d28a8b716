/*
 * hvx_oracle.c -- CPU ORACLE (test infrastructure only; see hvx_oracle.h).
 *
 * Restates, function by function, the HelixDB reference's vector-search hot path.
 * Paths are relative to /root/reference/crates/db/src/search/vector/.
 *
 * Build with -ffp-contract=off: the reference is Rust, which never contracts a*b+c into an
 * FMA on its own; every fused operation below is an explicit fmaf().
 */
#include "hvx_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#define ORC_HW_AVXFMA 1
#else
#define ORC_HW_AVXFMA 0
#endif

/* ------------------------------------------------------------------------------------------
 * Distance kernels
 * ---------------------------------------------------------------------------------------- */

/* spaces/simple.rs:204-219 euclidean_distance_scalar: distance += (l-r)*(l-r), left to right */
static float l2sq_scalar(const float *a, const float *b, uint32_t n) {
    float d = 0.0f;
    for (uint32_t i = 0; i < n; ++i) {
        float x = a[i] - b[i];
        float y = a[i] - b[i];
        d += x * y;
    }
    return d;
}

/* spaces/simple.rs:221-234 dot_product_scalar */
static float dot_scalar(const float *a, const float *b, uint32_t n) {
    float p = 0.0f;
    for (uint32_t i = 0; i < n; ++i) p += a[i] * b[i];
    return p;
}

/* spaces/simple.rs:186-202 manhattan_distance: sequential, no SIMD variant exists */
float orc_manhattan(const float *a, const float *b, uint32_t n) {
    float d = 0.0f;
    for (uint32_t i = 0; i < n; ++i) d += fabsf(a[i] - b[i]);
    return d;
}

/*
 * Generic "lanes x 4 accumulators" SIMD tree, emulated with scalar ops in the same order.
 *   width   = lanes per register (8 for AVX, 4 for SSE/NEON); 4 registers => 4*width floats/iter
 *   fused   = accumulate with fmaf (simple_avx.rs:128-238 *_avx_fma, simple_neon.rs vfmaq_f32)
 *             or mul-then-add (simple_avx.rs:15-125 *_avx, simple_sse.rs:17-67)
 *   is_l2   = (a-b)^2 terms, else a*b
 *   neon    = final horizontal add is vaddvq_f32 (pairwise) instead of the x86 movehl/shuffle tree
 * Tail (n % (4*width)) is the reference's scalar loop: result += d*d (unfused).
 */
static float simd_tree(const float *a, const float *b, uint32_t n, int width, int fused, int is_l2,
                       int neon) {
    const uint32_t step = 4u * (uint32_t)width;
    const uint32_t m = n - (n % step);
    float acc[32];
    for (int v = 0; v < 32; ++v) acc[v] = 0.0f;
    for (uint32_t i = 0; i < m; i += step) {
        for (uint32_t v = 0; v < step; ++v) {
            float x, y;
            if (is_l2) {
                x = a[i + v] - b[i + v];
                y = x;
            } else {
                x = a[i + v];
                y = b[i + v];
            }
            if (fused) {
                acc[v] = fmaf(x, y, acc[v]);
            } else {
                float p = x * y;
                acc[v] = p + acc[v];
            }
        }
    }
    /* (sum1 + sum2) + (sum3 + sum4), lane-wise */
    float t[8];
    for (int c = 0; c < width; ++c) {
        float s12 = acc[c] + acc[width + c];
        float s34 = acc[2 * width + c] + acc[3 * width + c];
        t[c] = s12 + s34;
    }
    float x128[4];
    if (width == 8) {
        /* hsum256_ps_avx (simple_avx.rs:7-12): hi128 + lo128 */
        for (int c = 0; c < 4; ++c) x128[c] = t[4 + c] + t[c];
    } else {
        for (int c = 0; c < 4; ++c) x128[c] = t[c];
    }
    float result;
    if (neon) {
        /* vaddvq_f32: pairwise (x0+x1)+(x2+x3) */
        float p0 = x128[0] + x128[1];
        float p1 = x128[2] + x128[3];
        result = p0 + p1;
    } else {
        /* hsum128 (simple_sse.rs:10-14 / simple_avx.rs:9-11): x + movehl(x,x), then lane0+lane1 */
        float x64_0 = x128[0] + x128[2];
        float x64_1 = x128[1] + x128[3];
        result = x64_0 + x64_1;
    }
    for (uint32_t i = m; i < n; ++i) {
        if (is_l2) {
            float d = a[i] - b[i];
            float p = d * d;
            result += p;
        } else {
            float p = a[i] * b[i];
            result += p;
        }
    }
    return result;
}

/* spaces/simple.rs:120-144 euclidean_distance (dispatch incl. MIN_DIM_SIZE_AVX=32 / _SIMD=16) */
float orc_euclidean(const float *a, const float *b, uint32_t n, int kernel) {
    switch (kernel) {
    case ORC_KERNEL_AVX_FMA_HW:
        if (orc_have_avxfma_hw()) return orc_euclidean_avxfma_hw(a, b, n);
        if (n >= 32) return simd_tree(a, b, n, 8, 1, 1, 0);
        break;
    case ORC_KERNEL_AVX_FMA:
        if (n >= 32) return simd_tree(a, b, n, 8, 1, 1, 0);
        break;
    case ORC_KERNEL_AVX:
        if (n >= 32) return simd_tree(a, b, n, 8, 0, 1, 0);
        break;
    case ORC_KERNEL_SSE:
        if (n >= 16) return simd_tree(a, b, n, 4, 0, 1, 0);
        break;
    case ORC_KERNEL_NEON:
        if (n >= 16) return simd_tree(a, b, n, 4, 1, 1, 1);
        break;
    default:
        break;
    }
    return l2sq_scalar(a, b, n);
}

/* spaces/simple.rs:155-178 dot_product */
float orc_dot(const float *a, const float *b, uint32_t n, int kernel) {
    switch (kernel) {
    case ORC_KERNEL_AVX_FMA_HW:
        if (orc_have_avxfma_hw()) return orc_dot_avxfma_hw(a, b, n);
        if (n >= 32) return simd_tree(a, b, n, 8, 1, 0, 0);
        break;
    case ORC_KERNEL_AVX_FMA:
        if (n >= 32) return simd_tree(a, b, n, 8, 1, 0, 0);
        break;
    case ORC_KERNEL_AVX:
        if (n >= 32) return simd_tree(a, b, n, 8, 0, 0, 0);
        break;
    case ORC_KERNEL_SSE:
        if (n >= 16) return simd_tree(a, b, n, 4, 0, 0, 0);
        break;
    case ORC_KERNEL_NEON:
        if (n >= 16) return simd_tree(a, b, n, 4, 1, 0, 1);
        break;
    default:
        break;
    }
    return dot_scalar(a, b, n);
}

#if ORC_HW_AVXFMA
static inline float hsum256(__m256 x) { /* simple_avx.rs:7-12 */
    __m128 x128 = _mm_add_ps(_mm256_extractf128_ps(x, 1), _mm256_castps256_ps128(x));
    __m128 x64 = _mm_add_ps(x128, _mm_movehl_ps(x128, x128));
    __m128 x32 = _mm_add_ss(x64, _mm_shuffle_ps(x64, x64, 0x55));
    return _mm_cvtss_f32(x32);
}
float orc_euclidean_avxfma_hw(const float *a, const float *b, uint32_t n) {
    if (n < 32) return l2sq_scalar(a, b, n);
    uint32_t m = n - (n % 32);
    __m256 s1 = _mm256_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
    for (uint32_t i = 0; i < m; i += 32) {
        __m256 d1 = _mm256_sub_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i));
        s1 = _mm256_fmadd_ps(d1, d1, s1);
        __m256 d2 = _mm256_sub_ps(_mm256_loadu_ps(a + i + 8), _mm256_loadu_ps(b + i + 8));
        s2 = _mm256_fmadd_ps(d2, d2, s2);
        __m256 d3 = _mm256_sub_ps(_mm256_loadu_ps(a + i + 16), _mm256_loadu_ps(b + i + 16));
        s3 = _mm256_fmadd_ps(d3, d3, s3);
        __m256 d4 = _mm256_sub_ps(_mm256_loadu_ps(a + i + 24), _mm256_loadu_ps(b + i + 24));
        s4 = _mm256_fmadd_ps(d4, d4, s4);
    }
    float r = hsum256(_mm256_add_ps(_mm256_add_ps(s1, s2), _mm256_add_ps(s3, s4)));
    for (uint32_t i = m; i < n; ++i) {
        float d = a[i] - b[i];
        float p = d * d;
        r += p;
    }
    return r;
}
float orc_dot_avxfma_hw(const float *a, const float *b, uint32_t n) {
    if (n < 32) return dot_scalar(a, b, n);
    uint32_t m = n - (n % 32);
    __m256 s1 = _mm256_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
    for (uint32_t i = 0; i < m; i += 32) {
        s1 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i), _mm256_loadu_ps(b + i), s1);
        s2 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 8), _mm256_loadu_ps(b + i + 8), s2);
        s3 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 16), _mm256_loadu_ps(b + i + 16), s3);
        s4 = _mm256_fmadd_ps(_mm256_loadu_ps(a + i + 24), _mm256_loadu_ps(b + i + 24), s4);
    }
    float r = hsum256(_mm256_add_ps(_mm256_add_ps(s1, s2), _mm256_add_ps(s3, s4)));
    for (uint32_t i = m; i < n; ++i) {
        float p = a[i] * b[i];
        r += p;
    }
    return r;
}
int orc_have_avxfma_hw(void) { return 1; }
#else
float orc_euclidean_avxfma_hw(const float *a, const float *b, uint32_t n) {
    (void)a; (void)b; (void)n;
    return NAN;
}
float orc_dot_avxfma_hw(const float *a, const float *b, uint32_t n) {
    (void)a; (void)b; (void)n;
    return NAN;
}
int orc_have_avxfma_hw(void) { return 0; }
#endif

/* The SSE kernels (simple_sse.rs:10-67,69-110) executed with the real instructions: every x86-64 host has SSE, so the emulated
 * 4 x 4-lane tree (simd_tree, width 4, unfused, hsum128) can be pinned against hardware wherever the oracle is built.  (The NEON
 * tree differs from it only in the fused accumulate -- pinned through the AVX+FMA hardware twin's fmaf semantics -- and in the
 * pairwise final add, which no x86 instruction reproduces: it stays a restatement of vaddvq_f32's documented order.) */
#if defined(__SSE__)
#include <xmmintrin.h>
static inline float hsum128_sse(__m128 x) { /* simple_sse.rs:10-14 */
    __m128 x64 = _mm_add_ps(x, _mm_movehl_ps(x, x));
    __m128 x32 = _mm_add_ss(x64, _mm_shuffle_ps(x64, x64, 0x55));
    return _mm_cvtss_f32(x32);
}
float orc_euclidean_sse_hw(const float *a, const float *b, uint32_t n) {
    if (n < 16) return l2sq_scalar(a, b, n);
    uint32_t m = n - (n % 16);
    __m128 s1 = _mm_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
    for (uint32_t i = 0; i < m; i += 16) {
        __m128 d1 = _mm_sub_ps(_mm_loadu_ps(a + i), _mm_loadu_ps(b + i));
        s1 = _mm_add_ps(_mm_mul_ps(d1, d1), s1);
        __m128 d2 = _mm_sub_ps(_mm_loadu_ps(a + i + 4), _mm_loadu_ps(b + i + 4));
        s2 = _mm_add_ps(_mm_mul_ps(d2, d2), s2);
        __m128 d3 = _mm_sub_ps(_mm_loadu_ps(a + i + 8), _mm_loadu_ps(b + i + 8));
        s3 = _mm_add_ps(_mm_mul_ps(d3, d3), s3);
        __m128 d4 = _mm_sub_ps(_mm_loadu_ps(a + i + 12), _mm_loadu_ps(b + i + 12));
        s4 = _mm_add_ps(_mm_mul_ps(d4, d4), s4);
    }
    float r = hsum128_sse(_mm_add_ps(_mm_add_ps(s1, s2), _mm_add_ps(s3, s4)));
    for (uint32_t i = m; i < n; ++i) {
        float d = a[i] - b[i];
        float p = d * d;
        r += p;
    }
    return r;
}
float orc_dot_sse_hw(const float *a, const float *b, uint32_t n) {
    if (n < 16) return dot_scalar(a, b, n);
    uint32_t m = n - (n % 16);
    __m128 s1 = _mm_setzero_ps(), s2 = s1, s3 = s1, s4 = s1;
    for (uint32_t i = 0; i < m; i += 16) {
        s1 = _mm_add_ps(_mm_mul_ps(_mm_loadu_ps(a + i), _mm_loadu_ps(b + i)), s1);
        s2 = _mm_add_ps(_mm_mul_ps(_mm_loadu_ps(a + i + 4), _mm_loadu_ps(b + i + 4)), s2);
        s3 = _mm_add_ps(_mm_mul_ps(_mm_loadu_ps(a + i + 8), _mm_loadu_ps(b + i + 8)), s3);
        s4 = _mm_add_ps(_mm_mul_ps(_mm_loadu_ps(a + i + 12), _mm_loadu_ps(b + i + 12)), s4);
    }
    float r = hsum128_sse(_mm_add_ps(_mm_add_ps(s1, s2), _mm_add_ps(s3, s4)));
    for (uint32_t i = m; i < n; ++i) {
        float p = a[i] * b[i];
        r += p;
    }
    return r;
}
int orc_have_sse_hw(void) { return 1; }
#else
float orc_euclidean_sse_hw(const float *a, const float *b, uint32_t n) { (void)a; (void)b; (void)n; return NAN; }
float orc_dot_sse_hw(const float *a, const float *b, uint32_t n) { (void)a; (void)b; (void)n; return NAN; }
int orc_have_sse_hw(void) { return 0; }
#endif

/* distance/cosine.rs:12-36 scaled_l2_norm */
double orc_scaled_l2_norm(const float *v, uint32_t n) {
    double scale = 0.0, scaled_sum = 1.0;
    for (uint32_t i = 0; i < n; ++i) {
        double magnitude = (double)fabsf(v[i]);
        if (magnitude == 0.0) continue;
        if (scale < magnitude) {
            double ratio = scale / magnitude;
            scaled_sum = 1.0 + scaled_sum * ratio * ratio;
            scale = magnitude;
        } else {
            double ratio = magnitude / scale;
            scaled_sum += ratio * ratio;
        }
    }
    if (scale == 0.0) return 0.0;
    return scale * sqrt(scaled_sum);
}

/* distance/cosine.rs:39-59 stable_half_cosine */
static float stable_half_cosine(const float *p, const float *q, uint32_t n) {
    double pn = orc_scaled_l2_norm(p, n);
    double qn = orc_scaled_l2_norm(q, n);
    if (pn == 0.0 || qn == 0.0) return NAN;
    double dot = 0.0;
    for (uint32_t i = 0; i < n; ++i) dot += (double)p[i] * (double)q[i];
    double c = dot / (pn * qn);
    if (c < -1.0) c = -1.0;
    if (c > 1.0) c = 1.0;
    return (float)((1.0 - c) * 0.5);
}

/* Distance::new_header: cosine.rs:89-93,120-122 (norm), euclidean.rs:42-44 / manhattan.rs (bias 0) */
float orc_header(int metric, const float *v, uint32_t n) {
    if (metric != ORC_COSINE) return 0.0f;
    double norm = orc_scaled_l2_norm(v, n);
    double mx = (double)FLT_MAX;
    if (norm > mx) norm = mx; /* f64::min */
    return (float)norm;
}

/* distance/cosine.rs:96-118, euclidean.rs:46-48, manhattan.rs:45-47 */
float orc_distance(int metric, int kernel, const float *p, float pn, const float *q, float qn,
                   uint32_t n) {
    if (metric == ORC_L2SQ) return orc_euclidean(p, q, n, kernel);
    if (metric == ORC_L1) return orc_manhattan(p, q, n);
    float pq = orc_dot(p, q, n, kernel);
    float pnqn = pn * qn;
    if (pn > 0.0f && qn > 0.0f && pn != FLT_MAX && qn != FLT_MAX && isnormal(pnqn) && isfinite(pq)) {
        float c = pq / pnqn;
        if (c < -1.0f) c = -1.0f;
        if (c > 1.0f) c = 1.0f;
        return (1.0f - c) / 2.0f;
    }
    return stable_half_cosine(p, q, n);
}

/* ------------------------------------------------------------------------------------------
 * Validation
 * ---------------------------------------------------------------------------------------- */

/* domain.rs:26-78 VectorComponentLimit::try_new */
float orc_component_limit(int metric, uint32_t dim) {
    if (metric == ORC_COSINE) return 0.0f;
    double factor = (metric == ORC_L2SQ) ? 8.0 : 4.0;
    double divisor = (double)((uint64_t)dim * (uint64_t)factor);
    double exact = (metric == ORC_L2SQ) ? sqrt((double)FLT_MAX / divisor) : (double)FLT_MAX / divisor;
    float rounded = (float)exact;
    if ((double)rounded > exact) {
        uint32_t bits;
        memcpy(&bits, &rounded, 4);
        bits -= 1;
        memcpy(&rounded, &bits, 4);
    }
    return rounded;
}

/* domain.rs:113-157 ValidatedMetricVector::try_new (check order is part of the contract) */
int orc_validate_vector(int metric, const float *v, uint32_t n_actual, uint32_t dim, uint32_t *bad) {
    if (bad) *bad = 0;
    if (n_actual != dim) return ORC_ERR_DIMENSION;
    for (uint32_t i = 0; i < dim; ++i) {
        if (!isfinite(v[i])) {
            if (bad) *bad = i;
            return ORC_ERR_NONFINITE;
        }
    }
    if (metric == ORC_COSINE) {
        int zero = 1;
        for (uint32_t i = 0; i < dim; ++i)
            if (v[i] != 0.0f) { zero = 0; break; }
        if (zero) return ORC_ERR_ZERO_NORM;
        return ORC_OK;
    }
    float limit = orc_component_limit(metric, dim);
    for (uint32_t i = 0; i < dim; ++i) {
        if (fabsf(v[i]) > limit) {
            if (bad) *bad = i;
            return ORC_ERR_MAGNITUDE;
        }
    }
    return ORC_OK;
}

/* parameters.rs:243-274 DistanceScore::try_new + normalize_zero */
int orc_distance_score(float *s) {
    if (!isfinite(*s)) return ORC_ERR_INVARIANT;
    if (*s < 0.0f) return ORC_ERR_INVARIANT;
    if (*s == 0.0f) *s = 0.0f; /* -0 -> +0 */
    return ORC_OK;
}

/* mod.rs:705-708 */
float orc_default_ml_for_m(uint32_t m) {
    float em = (float)(m < 2 ? 2 : m);
    return 1.0f / logf(em);
}

/* mod.rs:776-796 select_layer_from_uniform */
uint16_t orc_select_layer_from_uniform(float ml, float uniform) {
    if (!(isfinite(ml) && ml > 0.0f)) ml = orc_default_ml_for_m(16);
    if (isfinite(uniform)) {
        float lo = FLT_MIN, hi = 1.0f - FLT_EPSILON;
        if (uniform < lo) uniform = lo;
        if (uniform > hi) uniform = hi;
    } else {
        uniform = 0.5f;
    }
    float sampled = floorf(-logf(uniform) * ml);
    if (!isfinite(sampled) || sampled <= 0.0f) return 0;
    if (sampled > 63.0f) sampled = 63.0f;
    return (uint16_t)sampled;
}

/* ------------------------------------------------------------------------------------------
 * Index
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    float score;
    uint32_t idx;
} cand_t;

struct orc_index {
    uint32_t dim, m, m0, m0_eff, efc;
    int metric, kernel;
    uint64_t n, cap;
    uint64_t *ids;
    float *vec;
    float *hdr;
    uint16_t *level;
    /* layer 0: fixed stride rows of internal indices kept sorted by external id */
    uint32_t s0;
    uint32_t *l0;
    uint32_t *l0_deg;
    /* upper layers: up_base[node] = first row (layer 1) in the upper table, level[node] rows */
    uint32_t su;
    uint64_t *up_base;
    uint32_t *up;
    uint32_t *up_deg;
    uint64_t up_rows, up_cap;
    int has_entry;
    uint32_t entry;
    uint16_t max_layer;
    /* id -> index open-addressing map */
    uint64_t map_cap;
    uint64_t *map_key;
    uint32_t *map_val;
    /* per-node SimHash rows + the hasher's hyperplanes (orc_index_set_simhash) */
    uint64_t *simhash;
    float *planes;
    uint64_t simhash_seed;
    int has_simhash;
    /* orc_index_delete: a deleted node keeps its slot (rows emptied, unreachable, absent from the id map); n counts slots */
    uint8_t *dead;
    uint64_t n_dead;
};

#define NO_ROW UINT64_MAX
#define MAP_TOMB (UINT32_MAX - 1u)

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

static void map_rebuild(orc_index *ix, uint64_t cap) {
    free(ix->map_key); free(ix->map_val);
    ix->map_cap = cap;
    ix->map_key = (uint64_t *)malloc(cap * 8);
    ix->map_val = (uint32_t *)malloc(cap * 4);
    for (uint64_t i = 0; i < cap; ++i) ix->map_val[i] = UINT32_MAX;
    for (uint64_t i = 0; i < ix->n; ++i) {
        if (ix->dead && ix->dead[i]) continue;
        uint64_t h = mix64(ix->ids[i]) & (cap - 1);
        while (ix->map_val[h] != UINT32_MAX) h = (h + 1) & (cap - 1);
        ix->map_key[h] = ix->ids[i];
        ix->map_val[h] = (uint32_t)i;
    }
}

static uint32_t map_find(const orc_index *ix, uint64_t id) {
    if (!ix->map_cap) return UINT32_MAX;
    uint64_t h = mix64(id) & (ix->map_cap - 1);
    while (ix->map_val[h] != UINT32_MAX) {
        if (ix->map_key[h] == id && ix->map_val[h] != MAP_TOMB) return ix->map_val[h];
        h = (h + 1) & (ix->map_cap - 1);
    }
    return UINT32_MAX;
}

static void map_remove(orc_index *ix, uint64_t id) { /* the slot stays in its probe chain */
    if (!ix->map_cap) return;
    uint64_t h = mix64(id) & (ix->map_cap - 1);
    while (ix->map_val[h] != UINT32_MAX) {
        if (ix->map_key[h] == id && ix->map_val[h] != MAP_TOMB) { ix->map_val[h] = MAP_TOMB; return; }
        h = (h + 1) & (ix->map_cap - 1);
    }
}

static void map_insert(orc_index *ix, uint64_t id, uint32_t idx) {
    if ((ix->n + 1) * 2 > ix->map_cap) map_rebuild(ix, ix->map_cap ? ix->map_cap * 2 : 1024);
    uint64_t h = mix64(id) & (ix->map_cap - 1);
    while (ix->map_val[h] != UINT32_MAX) h = (h + 1) & (ix->map_cap - 1);
    ix->map_key[h] = id;
    ix->map_val[h] = idx;
}

orc_index *orc_index_new(uint32_t dim, int metric, int kernel, uint32_t m, uint32_t m0, uint32_t efc) {
    orc_index *ix = (orc_index *)calloc(1, sizeof(orc_index));
    ix->dim = dim; ix->metric = metric; ix->kernel = kernel;
    ix->m = m; ix->m0 = m0; ix->efc = efc;
    /* mutation.rs:178-196: layer0 limit = max(m0, 2m) */
    ix->m0_eff = m0 >= 2 * m ? m0 : 2 * m;
    ix->s0 = ix->m0_eff + 1;
    ix->su = m + 1;
    return ix;
}

void orc_index_free(orc_index *ix) {
    if (!ix) return;
    free(ix->ids); free(ix->vec); free(ix->hdr); free(ix->level); free(ix->l0); free(ix->l0_deg);
    free(ix->up_base); free(ix->up); free(ix->up_deg); free(ix->map_key); free(ix->map_val);
    free(ix->simhash); free(ix->planes); free(ix->dead);
    free(ix);
}

static void grow_nodes(orc_index *ix, uint64_t need) {
    if (need <= ix->cap) return;
    uint64_t cap = ix->cap ? ix->cap : 1024;
    while (cap < need) cap *= 2;
    ix->ids = (uint64_t *)realloc(ix->ids, cap * 8);
    ix->vec = (float *)realloc(ix->vec, cap * (size_t)ix->dim * 4);
    ix->hdr = (float *)realloc(ix->hdr, cap * 4);
    ix->level = (uint16_t *)realloc(ix->level, cap * 2);
    ix->l0 = (uint32_t *)realloc(ix->l0, cap * (size_t)ix->s0 * 4);
    ix->l0_deg = (uint32_t *)realloc(ix->l0_deg, cap * 4);
    ix->up_base = (uint64_t *)realloc(ix->up_base, cap * 8);
    ix->dead = (uint8_t *)realloc(ix->dead, cap);
    memset(ix->dead + ix->cap, 0, cap - ix->cap);
    ix->cap = cap;
}

static void grow_up(orc_index *ix, uint64_t need) {
    if (need <= ix->up_cap) return;
    uint64_t cap = ix->up_cap ? ix->up_cap : 256;
    while (cap < need) cap *= 2;
    ix->up = (uint32_t *)realloc(ix->up, cap * (size_t)ix->su * 4);
    ix->up_deg = (uint32_t *)realloc(ix->up_deg, cap * 4);
    ix->up_cap = cap;
}

uint64_t orc_index_count(const orc_index *ix) { return ix->n - ix->n_dead; } /* metadata.count: live rows */

int orc_index_entry(const orc_index *ix, uint64_t *entry, uint16_t *max_layer) {
    if (!ix->has_entry) return 0;
    if (entry) *entry = ix->ids[ix->entry];
    if (max_layer) *max_layer = ix->max_layer;
    return 1;
}

static inline const float *row(const orc_index *ix, uint32_t i) { return ix->vec + (size_t)i * ix->dim; }

static inline float dist_q(const orc_index *ix, const float *q, float qh, uint32_t i) {
    return orc_distance(ix->metric, ix->kernel, q, qh, row(ix, i), ix->hdr[i], ix->dim);
}

/* neighbour row accessors: layer 0 or upper; rows absent at that layer read as empty */
static uint32_t *nbr_row(const orc_index *ix, uint32_t layer, uint32_t node, uint32_t **deg) {
    if (layer == 0) {
        *deg = &ix->l0_deg[node];
        return ix->l0 + (size_t)node * ix->s0;
    }
    if (ix->level[node] < layer || ix->up_base[node] == NO_ROW) {
        *deg = NULL;
        return NULL;
    }
    uint64_t r = ix->up_base[node] + (layer - 1);
    *deg = &ix->up_deg[r];
    return ix->up + (size_t)r * ix->su;
}

/* model.rs:55-61 Candidate::cmp: score, then node id */
static inline int cand_less(const orc_index *ix, cand_t a, cand_t b) {
    if (a.score < b.score) return 1;
    if (a.score > b.score) return 0;
    return ix->ids[a.idx] < ix->ids[b.idx];
}

/* binary heaps (std::collections::BinaryHeap semantics: only the extremum matters) */
typedef struct {
    cand_t *a;
    uint32_t n, cap;
    int is_max;
} heap_t;

static void heap_init(heap_t *h, int is_max) { h->a = NULL; h->n = 0; h->cap = 0; h->is_max = is_max; }
static void heap_free(heap_t *h) { free(h->a); h->a = NULL; h->n = h->cap = 0; }
static inline int heap_before(const orc_index *ix, const heap_t *h, cand_t x, cand_t y) {
    return h->is_max ? cand_less(ix, y, x) : cand_less(ix, x, y);
}
static void heap_push(const orc_index *ix, heap_t *h, cand_t c) {
    if (h->n == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 64;
        h->a = (cand_t *)realloc(h->a, h->cap * sizeof(cand_t));
    }
    uint32_t i = h->n++;
    h->a[i] = c;
    while (i > 0) {
        uint32_t p = (i - 1) / 2;
        if (!heap_before(ix, h, h->a[i], h->a[p])) break;
        cand_t t = h->a[i]; h->a[i] = h->a[p]; h->a[p] = t;
        i = p;
    }
}
static cand_t heap_pop(const orc_index *ix, heap_t *h) {
    cand_t top = h->a[0];
    h->a[0] = h->a[--h->n];
    uint32_t i = 0;
    for (;;) {
        uint32_t l = 2 * i + 1, r = l + 1, b = i;
        if (l < h->n && heap_before(ix, h, h->a[l], h->a[b])) b = l;
        if (r < h->n && heap_before(ix, h, h->a[r], h->a[b])) b = r;
        if (b == i) break;
        cand_t t = h->a[i]; h->a[i] = h->a[b]; h->a[b] = t;
        i = b;
    }
    return top;
}

static const orc_index *g_sort_ix;
static int cand_qsort_cmp(const void *pa, const void *pb) {
    cand_t a = *(const cand_t *)pa, b = *(const cand_t *)pb;
    if (cand_less(g_sort_ix, a, b)) return -1;
    if (cand_less(g_sort_ix, b, a)) return 1;
    return 0;
}
static void sort_cands(const orc_index *ix, cand_t *a, uint32_t n) {
    /* insertion sort for small n keeps this re-entrant; larger n via merge-free qsort + global */
    if (n <= 64) {
        for (uint32_t i = 1; i < n; ++i) {
            cand_t x = a[i];
            uint32_t j = i;
            while (j > 0 && cand_less(ix, x, a[j - 1])) { a[j] = a[j - 1]; --j; }
            a[j] = x;
        }
        return;
    }
    /* heap sort: re-entrant, O(n log n) */
    heap_t h; heap_init(&h, 0);
    for (uint32_t i = 0; i < n; ++i) heap_push(ix, &h, a[i]);
    for (uint32_t i = 0; i < n; ++i) a[i] = heap_pop(ix, &h);
    heap_free(&h);
    (void)cand_qsort_cmp; (void)g_sort_ix;
}

/* visited set: epoch-stamped array, one per thread */
typedef struct {
    uint32_t *stamp;
    uint64_t cap;
    uint32_t epoch;
} visit_t;
static __thread visit_t tl_visit;

static void visit_begin(visit_t *v, uint64_t n) {
    if (v->cap < n) {
        free(v->stamp);
        v->cap = n + n / 2 + 16;
        v->stamp = (uint32_t *)calloc(v->cap, 4);
        v->epoch = 0;
    }
    if (++v->epoch == 0) {
        memset(v->stamp, 0, v->cap * 4);
        v->epoch = 1;
    }
}
/* returns 1 if newly inserted (HashSet::insert) */
static inline int visit_insert(visit_t *v, uint32_t i) {
    if (v->stamp[i] == v->epoch) return 0;
    v->stamp[i] = v->epoch;
    return 1;
}
static inline int visit_contains(const visit_t *v, uint32_t i) { return v->stamp[i] == v->epoch; }

/* search.rs:169-224 search_layer_greedy (== mutation.rs:1008-1064 for the write side) */
static int greedy_layer(const orc_index *ix, const float *q, float qh, uint32_t entry, uint32_t layer,
                        uint32_t *out) {
    visit_t *vis = &tl_visit;
    visit_begin(vis, ix->n);
    uint32_t cur = entry;
    float cur_d = dist_q(ix, q, qh, cur);
    if (orc_distance_score(&cur_d)) return ORC_ERR_INVARIANT;
    visit_insert(vis, cur);
    for (;;) {
        uint32_t *deg;
        uint32_t *nb = nbr_row(ix, layer, cur, &deg);
        uint32_t nd = nb ? *deg : 0;
        int changed = 0;
        /* the loop walks the list that was loaded for the node current at loop entry */
        for (uint32_t j = 0; j < nd; ++j) {
            uint32_t x = nb[j];
            if (!visit_insert(vis, x)) continue;
            float d = dist_q(ix, q, qh, x);
            if (orc_distance_score(&d)) return ORC_ERR_INVARIANT;
            if (d < cur_d) { cur = x; cur_d = d; changed = 1; }
        }
        if (!changed) break;
    }
    *out = cur;
    return ORC_OK;
}

/* search.rs:267-1067 with STRICT_EXHAUSTIVE=true; also
 * mutation.rs:904-1005 search_layer_beam when `layer` is given and stats==NULL.
 * Returns W sorted ascending by (score,id) in *out (caller frees). */
static int beam_layer(const orc_index *ix, const float *q, float qh, uint32_t entry, uint32_t layer,
                      uint32_t ef, cand_t **out, uint32_t *out_n, orc_stats *st) {
    visit_t *vis = &tl_visit;
    visit_begin(vis, ix->n);
    heap_t C, W;
    heap_init(&C, 0);
    heap_init(&W, 1);
    int rc = ORC_OK;
    float d0 = dist_q(ix, q, qh, entry);
    if (st) st->distance_computations += 1;
    if (orc_distance_score(&d0)) { rc = ORC_ERR_INVARIANT; goto done; }
    cand_t e = {d0, entry};
    heap_push(ix, &C, e);
    heap_push(ix, &W, e);
    visit_insert(vis, entry);
    uint32_t frontier[4096];
    while (C.n) {
        if (st) st->expansion_steps += 1;
        cand_t cur = heap_pop(ix, &C);
        if (W.n >= ef && cur.score > W.a[0].score) break;
        uint32_t *deg;
        uint32_t *nb = nbr_row(ix, layer, cur.idx, &deg);
        uint32_t nd = nb ? *deg : 0;
        if (st) st->neighbors_examined += nd;
        uint32_t nf = 0;
        for (uint32_t j = 0; j < nd; ++j) {
            if (visit_contains(vis, nb[j])) continue;
            frontier[nf++] = nb[j];
        }
        if (!nf) continue;
        for (uint32_t j = 0; j < nf; ++j) visit_insert(vis, frontier[j]); /* mark before scoring */
        if (st) st->vectors_loaded += nf;
        for (uint32_t j = 0; j < nf; ++j) {
            float d = dist_q(ix, q, qh, frontier[j]);
            if (st) st->distance_computations += 1;
            if (orc_distance_score(&d)) { rc = ORC_ERR_INVARIANT; goto done; }
            if (d < W.a[0].score || W.n < ef) {
                cand_t c = {d, frontier[j]};
                heap_push(ix, &C, c);
                heap_push(ix, &W, c);
                if (W.n > ef) heap_pop(ix, &W);
            }
        }
    }
    *out_n = W.n;
    *out = (cand_t *)malloc((W.n ? W.n : 1) * sizeof(cand_t));
    memcpy(*out, W.a, W.n * sizeof(cand_t));
    sort_cands(ix, *out, W.n);
done:
    heap_free(&C);
    heap_free(&W);
    return rc;
}

/* search.rs:1101-1230 SearchSession::run (strict-exhaustive parameters) */
int orc_search(const orc_index *ix, const float *query, uint32_t qlen, uint32_t k, uint32_t ef,
               uint64_t *out_ids, float *out_scores, uint32_t *out_count, orc_stats *stats) {
    if (out_count) *out_count = 0;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (k == 0 || ef < k) return ORC_ERR_K_RANGE; /* parameters.rs:100-133 */
    uint32_t bad;
    int rc = orc_validate_vector(ix->metric, query, qlen, ix->dim, &bad);
    if (rc) return rc;
    if (!ix->has_entry) return ORC_OK; /* VectorIndexState::Empty */
    float qh = orc_header(ix->metric, query, ix->dim);
    uint32_t entry = ix->entry;
    for (uint32_t layer = ix->max_layer; layer >= 1; --layer) {
        rc = greedy_layer(ix, query, qh, entry, layer, &entry);
        if (rc) return rc;
    }
    cand_t *w = NULL;
    uint32_t wn = 0;
    rc = beam_layer(ix, query, qh, entry, 0, ef, &w, &wn, stats);
    if (rc) return rc;
    uint32_t cnt = wn < k ? wn : k;
    for (uint32_t i = 0; i < cnt; ++i) {
        out_ids[i] = ix->ids[w[i].idx];
        out_scores[i] = w[i].score;
    }
    *out_count = cnt;
    free(w);
    return ORC_OK;
}

/* BASELINE.md CPU-baseline plan (ii): one query per thread at a time, static interleaved split */
#include <pthread.h>
typedef struct {
    const orc_index *ix;
    const float *queries;
    uint32_t nq, k, ef, tid, nthreads;
    uint64_t *ids;
    float *scores;
    uint32_t *counts;
    orc_stats *stats;
    int rc;
} mt_job;
static void *mt_worker(void *arg) {
    mt_job *j = (mt_job *)arg;
    for (uint32_t q = j->tid; q < j->nq; q += j->nthreads) {
        orc_stats st;
        int rc = orc_search(j->ix, j->queries + (size_t)q * j->ix->dim, j->ix->dim, j->k, j->ef,
                            j->ids + (size_t)q * j->k, j->scores + (size_t)q * j->k, j->counts + q, &st);
        if (j->stats) j->stats[q] = st;
        if (rc && !j->rc) j->rc = rc;
    }
    return NULL;
}
int orc_search_batch_mt(const orc_index *ix, const float *queries, uint32_t nq, uint32_t k, uint32_t ef,
                        uint32_t threads, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                        orc_stats *stats) {
    if (threads == 0) threads = 1;
    if (threads > nq) threads = nq ? nq : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    mt_job *jobs = (mt_job *)malloc(sizeof(mt_job) * threads);
    for (uint32_t t = 0; t < threads; ++t) {
        mt_job j = {ix, queries, nq, k, ef, t, threads, out_ids, out_scores, out_counts, stats, 0};
        jobs[t] = j;
        pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
    }
    int rc = 0;
    for (uint32_t t = 0; t < threads; ++t) {
        pthread_join(th[t], NULL);
        if (jobs[t].rc && !rc) rc = jobs[t].rc;
    }
    free(th);
    free(jobs);
    return rc;
}

/* restricted.rs:753-835 restricted_exact_scan (+ :661-704): bounded max-heap of k, final sort */
static int exact_scan(const orc_index *ix, const float *rows, const float *hdrs, uint64_t n,
                      int metric, int kernel, uint32_t dim, const float *q, uint32_t k,
                      const uint32_t *subset, uint64_t n_subset, cand_t *top, uint32_t *top_n) {
    float qh = orc_header(metric, q, dim);
    /* a plain array max-heap keyed on (score,id); ids == indices here unless ix is given */
    heap_t H;
    heap_init(&H, 1);
    orc_index fake;
    uint64_t *ident = NULL;
    const orc_index *cx = ix;
    if (!cx) {
        memset(&fake, 0, sizeof(fake));
        ident = (uint64_t *)malloc((n ? n : 1) * 8);
        for (uint64_t i = 0; i < n; ++i) ident[i] = i;
        fake.ids = ident;
        cx = &fake;
    }
    uint64_t total = subset ? n_subset : n;
    int rc = ORC_OK;
    for (uint64_t t = 0; t < total; ++t) {
        uint32_t i = subset ? subset[t] : (uint32_t)t;
        float h = hdrs ? hdrs[i] : orc_header(metric, rows + (size_t)i * dim, dim);
        float d = orc_distance(metric, kernel, q, qh, rows + (size_t)i * dim, h, dim);
        if (orc_distance_score(&d)) { rc = ORC_ERR_INVARIANT; break; }
        cand_t c = {d, i};
        heap_push(cx, &H, c);
        if (H.n > k) heap_pop(cx, &H);
    }
    if (!rc) {
        memcpy(top, H.a, H.n * sizeof(cand_t));
        *top_n = H.n;
        sort_cands(cx, top, H.n);
    }
    heap_free(&H);
    free(ident);
    return rc;
}

int orc_flat_search(const orc_index *ix, const float *query, uint32_t qlen, uint32_t k,
                    const uint64_t *allowed, uint64_t n_allowed, uint64_t *out_ids, float *out_scores,
                    uint32_t *out_count) {
    *out_count = 0;
    if (k == 0) return ORC_ERR_K_RANGE;
    uint32_t bad;
    int rc = orc_validate_vector(ix->metric, query, qlen, ix->dim, &bad);
    if (rc) return rc;
    uint32_t *subset = NULL;
    uint64_t ns = 0;
    if (allowed) {
        /* RestrictedVectorCandidates::from_ids dedupes (restricted.rs:356-371); ids that are not
         * indexed at all are omitted silently (restricted.rs:615-659: only a present row with a
         * missing companion fails closed; tests/production_support/vector/restricted.rs:788-800). */
        subset = (uint32_t *)malloc((n_allowed ? n_allowed : 1) * 4);
        visit_t *vis = &tl_visit;
        visit_begin(vis, ix->n);
        for (uint64_t t = 0; t < n_allowed; ++t) {
            uint32_t i = map_find(ix, allowed[t]);
            if (i == UINT32_MAX) continue;
            if (visit_insert(vis, i)) subset[ns++] = i;
        }
        if (ns == 0) { free(subset); return ORC_OK; }
    } else if (ix->n == ix->n_dead) {
        return ORC_OK;
    } else if (ix->n_dead) { /* deleted nodes have no item row: the scan sees the live rows only */
        subset = (uint32_t *)malloc(ix->n * 4);
        for (uint64_t i = 0; i < ix->n; ++i)
            if (!ix->dead[i]) subset[ns++] = (uint32_t)i;
    }
    cand_t *top = (cand_t *)malloc(((size_t)k + 1) * sizeof(cand_t));
    uint32_t tn = 0;
    rc = exact_scan(ix, ix->vec, ix->hdr, ix->n, ix->metric, ix->kernel, ix->dim, query, k, subset, ns,
                    top, &tn);
    if (!rc) {
        for (uint32_t i = 0; i < tn; ++i) {
            out_ids[i] = ix->ids[top[i].idx];
            out_scores[i] = top[i].score;
        }
        *out_count = tn;
    }
    free(top);
    free(subset);
    return rc;
}

int orc_flat_search_matrix(int metric, int kernel, const float *rows, uint64_t n, uint32_t dim,
                           const float *query, uint32_t k, uint64_t *out_ids, float *out_scores,
                           uint32_t *out_count) {
    *out_count = 0;
    if (k == 0) return ORC_ERR_K_RANGE;
    uint32_t bad;
    int rc = orc_validate_vector(metric, query, dim, dim, &bad);
    if (rc) return rc;
    if (n == 0) return ORC_OK;
    cand_t *top = (cand_t *)malloc(((size_t)k + 1) * sizeof(cand_t));
    uint32_t tn = 0;
    rc = exact_scan(NULL, rows, NULL, n, metric, kernel, dim, query, k, NULL, 0, top, &tn);
    if (!rc) {
        for (uint32_t i = 0; i < tn; ++i) {
            out_ids[i] = top[i].idx;
            out_scores[i] = top[i].score;
        }
        *out_count = tn;
    }
    free(top);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Build (mutation.rs)
 * ---------------------------------------------------------------------------------------- */

/* NeighborSet canonical form (neighbor_set.rs:1-9): sorted by node id, deduped, self-free */
static void canon_row(const orc_index *ix, uint32_t *r, uint32_t *deg, uint32_t self) {
    uint32_t n = *deg, w = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (r[i] != self) r[w++] = r[i];
    n = w;
    for (uint32_t i = 1; i < n; ++i) {
        uint32_t x = r[i];
        uint32_t j = i;
        while (j > 0 && ix->ids[x] < ix->ids[r[j - 1]]) { r[j] = r[j - 1]; --j; }
        r[j] = x;
    }
    w = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (i == 0 || r[i] != r[i - 1]) r[w++] = r[i];
    *deg = w;
}

static int row_contains(const uint32_t *r, uint32_t deg, uint32_t x) {
    for (uint32_t i = 0; i < deg; ++i)
        if (r[i] == x) return 1;
    return 0;
}

/* mod.rs:809-856 select_diverse.  `hydrated` = number of leading candidates that have items
 * (select_neighbors_heuristic hydrates only the first 2*Mmax, mutation.rs:1072-1097). */
static int select_diverse(const orc_index *ix, const cand_t *c, uint32_t nc, uint32_t hydrated,
                          uint32_t m, uint32_t *sel, uint32_t *nsel) {
    uint32_t ns = 0;
    if (hydrated > nc) hydrated = nc;
    for (uint32_t i = 0; i < hydrated && ns < m; ++i) {
        int diverse = 1;
        for (uint32_t s = 0; s < ns; ++s) {
            float pd = orc_distance(ix->metric, ix->kernel, row(ix, c[i].idx), ix->hdr[c[i].idx],
                                    row(ix, sel[s]), ix->hdr[sel[s]], ix->dim);
            if (orc_distance_score(&pd)) return ORC_ERR_INVARIANT;
            if (pd < c[i].score) { diverse = 0; break; }
        }
        if (diverse) sel[ns++] = c[i].idx;
    }
    if (ns < m) { /* backfill with the closest remaining hydrated candidates */
        for (uint32_t i = 0; i < hydrated && ns < m; ++i) {
            if (!row_contains(sel, ns, c[i].idx)) sel[ns++] = c[i].idx;
        }
    }
    *nsel = ns;
    return ORC_OK;
}

/* mutation.rs:1890-1908 remove_edge_from_neighbor */
static void remove_edge(orc_index *ix, uint32_t layer, uint32_t node, uint32_t to_remove) {
    uint32_t *deg;
    uint32_t *r = nbr_row(ix, layer, node, &deg);
    if (!r) return;
    uint32_t w = 0;
    for (uint32_t i = 0; i < *deg; ++i)
        if (r[i] != to_remove) r[w++] = r[i];
    *deg = w;
}

/* mutation.rs:1498-1583 add_bidirectional_link */
static int add_link(orc_index *ix, uint32_t layer, uint32_t from, uint32_t to, uint32_t maxn) {
    uint32_t *deg;
    uint32_t *r = nbr_row(ix, layer, to, &deg);
    if (!r) return ORC_ERR_INVARIANT;
    uint32_t stride = layer == 0 ? ix->s0 : ix->su;
    if (!row_contains(r, *deg, from)) {
        if (*deg >= stride) return ORC_ERR_INVARIANT;
        r[(*deg)++] = from;
    }
    uint32_t nc = *deg;
    uint32_t cand_ids[4096];
    memcpy(cand_ids, r, nc * 4);
    if (nc > maxn) {
        cand_t d[4096];
        for (uint32_t i = 0; i < nc; ++i) {
            float x = orc_distance(ix->metric, ix->kernel, row(ix, to), ix->hdr[to], row(ix, r[i]),
                                   ix->hdr[r[i]], ix->dim);
            if (orc_distance_score(&x)) return ORC_ERR_INVARIANT;
            d[i].score = x;
            d[i].idx = r[i];
        }
        sort_cands(ix, d, nc);
        uint32_t sel[4096], ns = 0;
        int rc = select_diverse(ix, d, nc, nc, maxn, sel, &ns);
        if (rc) return rc;
        memcpy(r, sel, ns * 4);
        *deg = ns;
    }
    canon_row(ix, r, deg, to);
    for (uint32_t i = 0; i < nc; ++i) {
        if (!row_contains(r, *deg, cand_ids[i])) remove_edge(ix, layer, cand_ids[i], to);
    }
    return ORC_OK;
}

/* Test hook: the prune step of add_bidirectional_link (mutation.rs:1498-1583) for ONE row, without touching the graph --
 * rank `cand_ids` (all stored in the index) by distance to `owner_id` in Candidate order, select_diverse (mod.rs:809-856) +
 * backfill with at most maxn survivors.  out_ids[0..*out_n) = the surviving ids in selection order.  The device's eager
 * evaluation (build_link_wg_kernel: all pairwise distances, predicate masks) is checked against this by a CPU twin. */
int orc_index_prune_candidates(const orc_index *ix, uint64_t owner_id, const uint64_t *cand_ids, uint32_t nc, uint32_t maxn,
                               uint64_t *out_ids, uint32_t *out_n) {
    if (!ix || !cand_ids || !out_ids || !out_n || nc > 4096) return ORC_ERR_INVARIANT;
    uint32_t to = map_find(ix, owner_id);
    if (to == UINT32_MAX) return ORC_ERR_INVARIANT;
    cand_t d[4096];
    for (uint32_t i = 0; i < nc; ++i) {
        uint32_t r = map_find(ix, cand_ids[i]);
        if (r == UINT32_MAX) return ORC_ERR_INVARIANT;
        float x = orc_distance(ix->metric, ix->kernel, row(ix, to), ix->hdr[to], row(ix, r), ix->hdr[r], ix->dim);
        if (orc_distance_score(&x)) return ORC_ERR_INVARIANT;
        d[i].score = x;
        d[i].idx = r;
    }
    sort_cands(ix, d, nc);
    uint32_t sel[4096], ns = 0;
    int rc = select_diverse(ix, d, nc, nc, maxn, sel, &ns);
    if (rc) return rc;
    for (uint32_t i = 0; i < ns; ++i) out_ids[i] = ix->ids[sel[i]];
    *out_n = ns;
    return ORC_OK;
}

/* mutation.rs:642-780 insert_with_mutation_cache + :787-895 insert_hnsw */
int orc_index_insert(orc_index *ix, uint64_t node_id, const float *v, uint16_t level) {
    uint32_t bad;
    int rc = orc_validate_vector(ix->metric, v, ix->dim, ix->dim, &bad);
    if (rc) return rc;
    if (map_find(ix, node_id) != UINT32_MAX) return ORC_ERR_INVARIANT; /* upsert not restated */
    if (level > 63) level = 63;
    grow_nodes(ix, ix->n + 1);
    uint32_t me = (uint32_t)ix->n;
    ix->ids[me] = node_id;
    memcpy(ix->vec + (size_t)me * ix->dim, v, (size_t)ix->dim * 4);
    ix->hdr[me] = orc_header(ix->metric, v, ix->dim);
    ix->level[me] = level;
    ix->l0_deg[me] = 0;
    if (level > 0) {
        grow_up(ix, ix->up_rows + level);
        ix->up_base[me] = ix->up_rows;
        for (uint32_t l = 0; l < level; ++l) ix->up_deg[ix->up_rows + l] = 0;
        ix->up_rows += level;
    } else {
        ix->up_base[me] = NO_ROW;
    }
    map_insert(ix, node_id, me);
    ix->n += 1;

    if (!ix->has_entry) { /* first row: becomes the entry point with empty rows */
        ix->has_entry = 1;
        ix->entry = me;
        ix->max_layer = level;
        return ORC_OK;
    }
    const float *q = row(ix, me);
    float qh = ix->hdr[me];
    uint32_t old_max = ix->max_layer;
    uint32_t cur = ix->entry;
    /* the new row must be invisible to its own insertion searches: it has no in-edges yet, so it
     * cannot be reached; visit stamps are sized with n already including it. */
    if (level < old_max) {
        for (uint32_t layer = old_max; layer >= (uint32_t)level + 1; --layer) {
            rc = greedy_layer(ix, q, qh, cur, layer, &cur);
            if (rc) return rc;
        }
    }
    uint32_t top = old_max < level ? old_max : level;
    for (int32_t layer = (int32_t)top; layer >= 0; --layer) {
        uint32_t maxn = layer == 0 ? ix->m0_eff : ix->m;
        uint32_t ef = layer == 0 ? (ix->efc > ix->m0_eff ? ix->efc : ix->m0_eff)
                                 : (ix->efc > 2 * ix->m ? ix->efc : 2 * ix->m);
        cand_t *cands = NULL;
        uint32_t nc = 0;
        rc = beam_layer(ix, q, qh, cur, (uint32_t)layer, ef, &cands, &nc, NULL);
        if (rc) return rc;
        uint32_t sel[4096], ns = 0;
        rc = select_diverse(ix, cands, nc, 2 * maxn, maxn, sel, &ns);
        if (rc) { free(cands); return rc; }
        uint32_t *deg;
        uint32_t *r = nbr_row(ix, (uint32_t)layer, me, &deg);
        memcpy(r, sel, ns * 4);
        *deg = ns;
        canon_row(ix, r, deg, me);
        for (uint32_t i = 0; i < ns; ++i) {
            rc = add_link(ix, (uint32_t)layer, me, sel[i], maxn);
            if (rc) { free(cands); return rc; }
        }
        if (nc) cur = cands[0].idx;
        free(cands);
    }
    if (level > old_max) {
        ix->entry = me;
        ix->max_layer = level;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * Delete (mutation.rs:1606-2055).  Parity unpinned by goldens: the reference's tests hold behaviour (index.rs:2263 unknown id,
 * :2294-2295 double delete, :3540-3568 entry repair, :3571-3605 the deleted id is never returned), not relinked rows.
 * ---------------------------------------------------------------------------------------- */
static int id_order_cmp(const void *a, const void *b, void *ctx);

static int remove_edge_found(orc_index *ix, uint32_t layer, uint32_t node, uint32_t to_remove) {
    uint32_t *deg;
    uint32_t *r = nbr_row(ix, layer, node, &deg);
    if (!r || !row_contains(r, *deg, to_remove)) return 0;
    remove_edge(ix, layer, node, to_remove);
    return 1;
}

/* rank `members` by distance to `owner` (Candidate order) and select_diverse + backfill; members without an item (deleted) are
 * skipped as get_item_for_layer_cached -> None does (mutation.rs:1957-1975, 2012-2033) */
static int prune_row(const orc_index *ix, uint32_t owner, const uint32_t *members, uint32_t nm, uint32_t maxn, uint32_t *out, uint32_t *nout) {
    cand_t *d = (cand_t *)malloc((nm ? nm : 1) * sizeof(cand_t));
    uint32_t nd = 0;
    for (uint32_t i = 0; i < nm; ++i) {
        if (ix->dead[members[i]]) continue;
        float x = orc_distance(ix->metric, ix->kernel, row(ix, owner), ix->hdr[owner], row(ix, members[i]), ix->hdr[members[i]], ix->dim);
        if (orc_distance_score(&x)) { free(d); return ORC_ERR_INVARIANT; }
        d[nd].score = x;
        d[nd].idx = members[i];
        ++nd;
    }
    sort_cands(ix, d, nd);
    int rc = select_diverse(ix, d, nd, nd, maxn, out, nout);
    free(d);
    return rc;
}

/* mutation.rs:1916-2055 relink_neighbor */
static int relink_neighbor(orc_index *ix, uint32_t layer, uint32_t nb, const uint32_t *cands, uint32_t ncand, uint32_t maxn) {
    if (ix->dead[nb]) return ORC_OK; /* :1925-1930 no item: nothing to relink */
    uint32_t *deg;
    uint32_t *r = nbr_row(ix, layer, nb, &deg);
    if (!r) return ORC_ERR_INVARIANT;
    const uint32_t stride = layer == 0 ? ix->s0 : ix->su;
    const uint32_t nold = *deg;
    uint32_t *old = (uint32_t *)malloc((nold + 1) * 4);
    memcpy(old, r, nold * 4);
    uint32_t *cur = (uint32_t *)malloc(((size_t)nold + maxn + 1) * 4);
    uint32_t ncur = nold;
    memcpy(cur, old, nold * 4);
    cand_t *cd = (cand_t *)malloc((ncand ? ncand : 1) * sizeof(cand_t));
    uint32_t nd = 0;
    int rc = ORC_OK;
    for (uint32_t i = 0; i < ncand; ++i) { /* :1936-1951 (the set's iteration order does not survive the sort) */
        const uint32_t c = cands[i];
        if (c == nb || ix->dead[c]) continue;
        float x = orc_distance(ix->metric, ix->kernel, row(ix, nb), ix->hdr[nb], row(ix, c), ix->hdr[c], ix->dim);
        if (orc_distance_score(&x)) { rc = ORC_ERR_INVARIANT; goto out; }
        cd[nd].score = x;
        cd[nd].idx = c;
        ++nd;
    }
    sort_cands(ix, cd, nd);
    for (uint32_t i = 0; i < nd && i < maxn; ++i) /* :1953-1957 */
        if (!row_contains(cur, ncur, cd[i].idx)) cur[ncur++] = cd[i].idx;
    if (ncur > maxn) { /* :1959-1984 */
        uint32_t *sel = (uint32_t *)malloc((ncur + 1) * 4), ns = 0;
        rc = prune_row(ix, nb, cur, ncur, maxn, sel, &ns);
        if (!rc) { memcpy(cur, sel, ns * 4); ncur = ns; }
        free(sel);
        if (rc) goto out;
    }
    if (ncur > stride) { rc = ORC_ERR_INVARIANT; goto out; }
    memcpy(r, cur, ncur * 4); /* :1986-1993 stage (stage_neighbors_vec_for_mutation sorts by id, :1299) */
    *deg = ncur;
    canon_row(ix, r, deg, nb);
    for (uint32_t t = 0; t < ncur; ++t) { /* :1994-2052 the reciprocal row of every NEW neighbour */
        const uint32_t nw = cur[t];
        if (row_contains(old, nold, nw)) continue;
        uint32_t *rdeg;
        uint32_t *rr = nbr_row(ix, layer, nw, &rdeg);
        if (!rr) { rc = ORC_ERR_INVARIANT; goto out; }
        if (row_contains(rr, *rdeg, nb)) continue;
        if (*rdeg >= stride) { rc = ORC_ERR_INVARIANT; goto out; }
        rr[(*rdeg)++] = nb;
        if (*rdeg > maxn) {
            if (ix->dead[nw]) { rc = ORC_ERR_INVARIANT; goto out; } /* :2007-2017: the over-long row fails its degree limit when staged */
            uint32_t sel[4096], ns = 0;
            rc = prune_row(ix, nw, rr, *rdeg, maxn, sel, &ns);
            if (rc) goto out;
            memcpy(rr, sel, ns * 4);
            *rdeg = ns;
        }
        canon_row(ix, rr, rdeg, nw);
    }
out:
    free(old); free(cur); free(cd);
    return rc;
}

/* mutation.rs:1819-1888 delete_from_layer; `extra` = the reverse-locator sources of this layer (every row that holds the node) */
static int delete_from_layer(orc_index *ix, uint32_t node, uint32_t layer, uint32_t maxn, const uint32_t *extra, uint32_t nextra, uint8_t *mark) {
    uint32_t *deg;
    uint32_t *own = nbr_row(ix, layer, node, &deg);
    const uint32_t nout = own ? *deg : 0;
    uint32_t *aff = (uint32_t *)malloc(((size_t)nout + nextra + 1) * 4);
    uint32_t na = 0;
    for (uint32_t i = 0; i < nout; ++i)
        if (own[i] != node && !mark[own[i]]) { mark[own[i]] = 1; aff[na++] = own[i]; } /* mandatory_relink (:1833-1837) */
    const uint32_t nmand = na;
    for (uint32_t i = 0; i < nextra; ++i)
        if (extra[i] != node && !mark[extra[i]]) { mark[extra[i]] = 2; aff[na++] = extra[i]; }
    if (na == 0) { free(aff); return ORC_OK; }
    qsort_r(aff, na, 4, id_order_cmp, (void *)ix); /* BTreeSet order */
    uint32_t *rel = (uint32_t *)malloc((na + 1) * 4);
    uint32_t nr = 0;
    for (uint32_t i = 0; i < na; ++i) { /* :1849-1857: mandatory sources relink whether or not they held the edge */
        const int had = remove_edge_found(ix, layer, aff[i], node);
        if (had || mark[aff[i]] == 1) rel[nr++] = aff[i];
    }
    (void)nmand;
    for (uint32_t i = 0; i < na; ++i) mark[aff[i]] = 0;
    int rc = ORC_OK;
    if (nr) {
        /* :1862-1875 candidates: the relink sources and their remaining neighbourhoods */
        size_t cap = (size_t)nr * ((layer == 0 ? ix->s0 : ix->su) + 1) + 1;
        uint32_t *cand = (uint32_t *)malloc(cap * 4);
        uint32_t nc = 0;
        mark[node] = 1;
        for (uint32_t i = 0; i < nr; ++i)
            if (!mark[rel[i]]) { mark[rel[i]] = 1; cand[nc++] = rel[i]; }
        for (uint32_t i = 0; i < nr; ++i) {
            uint32_t *d2;
            uint32_t *r2 = nbr_row(ix, layer, rel[i], &d2);
            for (uint32_t t = 0; r2 && t < *d2; ++t)
                if (!mark[r2[t]]) { mark[r2[t]] = 1; cand[nc++] = r2[t]; }
        }
        for (uint32_t i = 0; i < nc; ++i) mark[cand[i]] = 0;
        mark[node] = 0;
        for (uint32_t i = 0; i < nr && !rc; ++i) rc = relink_neighbor(ix, layer, rel[i], cand, nc, maxn);
        free(cand);
    }
    free(rel); free(aff);
    return rc;
}

/* mutation.rs:1658-1774 stage_delete_with_metadata.  An unknown id succeeds and changes nothing (index.rs:2263). */
int orc_index_delete(orc_index *ix, uint64_t node_id, int *existed) {
    const uint32_t x = map_find(ix, node_id);
    if (existed) *existed = x != UINT32_MAX;
    if (x == UINT32_MAX) return ORC_OK;
    uint8_t *mark = (uint8_t *)calloc(ix->n, 1);
    uint32_t *src = (uint32_t *)malloc(ix->n * 4);
    int rc = ORC_OK;
    uint32_t top = ix->level[x] > ix->max_layer ? ix->level[x] : ix->max_layer;
    for (int32_t layer = (int32_t)top; layer >= 0 && !rc; --layer) { /* layers_to_process, highest first (:1681-1702) */
        uint32_t ns = 0; /* reverse_sources_for_target: every row of this layer that holds the node (update_reverse_edge_locator :1134-1153) */
        for (uint64_t i = 0; i < ix->n; ++i) {
            if (i == x || ix->dead[i]) continue;
            uint32_t *deg;
            const uint32_t *r = nbr_row(ix, (uint32_t)layer, (uint32_t)i, &deg);
            if (r && row_contains(r, *deg, x)) src[ns++] = (uint32_t)i;
        }
        if ((uint32_t)layer > ix->level[x] && ns == 0) continue;
        rc = delete_from_layer(ix, x, (uint32_t)layer, layer == 0 ? ix->m0_eff : ix->m, src, ns, mark);
    }
    free(mark); free(src);
    if (rc) return rc;
    /* the node's rows, item, SimHash row and entry-candidate rows go (:1708-1745) */
    ix->l0_deg[x] = 0;
    for (uint32_t l = 0; l < ix->level[x]; ++l) ix->up_deg[ix->up_base[x] + l] = 0;
    ix->dead[x] = 1;
    ix->n_dead += 1;
    map_remove(ix, node_id);
    if (ix->has_entry && ix->entry == x) { /* :1756-1767 find_best_entry_candidate: highest layer first, then ascending id
                                               (keys/vectors.rs:1097 [inv_layer:2][node_id:8]) */
        int found = 0;
        uint32_t best = 0;
        for (uint64_t i = 0; i < ix->n; ++i) {
            if (ix->dead[i]) continue;
            if (!found || ix->level[i] > ix->level[best] || (ix->level[i] == ix->level[best] && ix->ids[i] < ix->ids[best])) { best = (uint32_t)i; found = 1; }
        }
        ix->has_entry = found;
        ix->entry = found ? best : 0;
        ix->max_layer = found ? ix->level[best] : 0;
    }
    return ORC_OK;
}

int orc_index_is_live(const orc_index *ix, uint64_t node_id) { return map_find(ix, node_id) != UINT32_MAX; }

int orc_index_seed(orc_index *ix, uint64_t n, const uint64_t *node_ids, const float *vectors,
                   const uint64_t *l0_off, const uint64_t *l0_nb, const uint16_t *level,
                   const uint64_t *up_off, const uint64_t *up_nb, int has_entry, uint64_t entry,
                   uint16_t max_layer) {
    if (ix->n) return ORC_ERR_INVARIANT;
    uint64_t maxdeg = ix->m0_eff;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t d = l0_off[i + 1] - l0_off[i];
        if (d > maxdeg) maxdeg = d;
    }
    ix->s0 = (uint32_t)maxdeg + 1;
    uint64_t upmax = ix->m;
    uint64_t rows = 0;
    if (level) {
        for (uint64_t i = 0; i < n; ++i) {
            for (uint32_t l = 0; l < level[i]; ++l) {
                uint64_t d = up_off[rows + 1] - up_off[rows];
                if (d > upmax) upmax = d;
                ++rows;
            }
        }
    }
    ix->su = (uint32_t)upmax + 1;
    grow_nodes(ix, n);
    grow_up(ix, rows);
    memcpy(ix->ids, node_ids, n * 8);
    memcpy(ix->vec, vectors, n * (size_t)ix->dim * 4);
    ix->n = n;
    {
        uint64_t cap = 1024;
        while (cap < 2 * n + 2) cap *= 2;
        map_rebuild(ix, cap);
    }
    uint64_t r = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t bad;
        int rc = orc_validate_vector(ix->metric, vectors + i * ix->dim, ix->dim, ix->dim, &bad);
        if (rc) return rc;
        ix->hdr[i] = orc_header(ix->metric, vectors + i * ix->dim, ix->dim);
        ix->level[i] = level ? level[i] : 0;
        uint32_t d = 0;
        uint32_t *rw = ix->l0 + (size_t)i * ix->s0;
        for (uint64_t e = l0_off[i]; e < l0_off[i + 1]; ++e) {
            uint32_t x = map_find(ix, l0_nb[e]);
            if (x == UINT32_MAX) return ORC_ERR_INVARIANT;
            rw[d++] = x;
        }
        ix->l0_deg[i] = d;
        canon_row(ix, rw, &ix->l0_deg[i], (uint32_t)i);
        if (ix->level[i] > 0) {
            ix->up_base[i] = r;
            for (uint32_t l = 0; l < ix->level[i]; ++l) {
                uint32_t *uw = ix->up + (size_t)r * ix->su;
                uint32_t ud = 0;
                for (uint64_t e = up_off[r]; e < up_off[r + 1]; ++e) {
                    uint32_t x = map_find(ix, up_nb[e]);
                    if (x == UINT32_MAX) return ORC_ERR_INVARIANT;
                    uw[ud++] = x;
                }
                ix->up_deg[r] = ud;
                canon_row(ix, uw, &ix->up_deg[r], (uint32_t)i);
                ++r;
            }
        } else {
            ix->up_base[i] = NO_ROW;
        }
    }
    ix->up_rows = r;
    ix->has_entry = has_entry && n > 0;
    if (ix->has_entry) {
        uint32_t e = map_find(ix, entry);
        if (e == UINT32_MAX) return ORC_ERR_INVARIANT;
        ix->entry = e;
        ix->max_layer = max_layer;
    }
    return ORC_OK;
}

/* export, ids ascending (the layout hvx_index_import takes) */
static int id_order_cmp(const void *a, const void *b, void *ctx) {
    const orc_index *ix = (const orc_index *)ctx;
    uint64_t x = ix->ids[*(const uint32_t *)a], y = ix->ids[*(const uint32_t *)b];
    return x < y ? -1 : (x > y ? 1 : 0);
}

uint64_t orc_index_export_sizes(const orc_index *ix, uint64_t *l0_edges, uint64_t *up_rows,
                                uint64_t *up_edges) {
    uint64_t e0 = 0, ue = 0, rows = 0;
    for (uint64_t i = 0; i < ix->n; ++i) {
        if (ix->dead[i]) continue; /* (a deleted node's rows are gone: mutation.rs:1726-1734) */
        e0 += ix->l0_deg[i];
        for (uint32_t l = 0; l < ix->level[i]; ++l) ue += ix->up_deg[ix->up_base[i] + l];
        rows += ix->level[i];
    }
    if (l0_edges) *l0_edges = e0;
    if (up_rows) *up_rows = rows;
    if (up_edges) *up_edges = ue;
    return ix->n - ix->n_dead;
}

int orc_index_export(const orc_index *ix, uint64_t *node_ids, float *vectors, uint64_t *l0_off,
                     uint64_t *l0_nb, uint16_t *level, uint64_t *up_off, uint64_t *up_nb) {
    uint64_t n = 0;
    uint32_t *ord = (uint32_t *)malloc((ix->n ? ix->n : 1) * 4);
    for (uint64_t i = 0; i < ix->n; ++i)
        if (!ix->dead[i]) ord[n++] = (uint32_t)i;
    qsort_r(ord, n, 4, id_order_cmp, (void *)ix);
    uint64_t e0 = 0, r = 0, ue = 0;
    l0_off[0] = 0;
    up_off[0] = 0;
    for (uint64_t t = 0; t < n; ++t) {
        uint32_t i = ord[t];
        node_ids[t] = ix->ids[i];
        memcpy(vectors + t * ix->dim, row(ix, i), (size_t)ix->dim * 4);
        level[t] = ix->level[i];
        const uint32_t *rw = ix->l0 + (size_t)i * ix->s0;
        for (uint32_t j = 0; j < ix->l0_deg[i]; ++j) l0_nb[e0++] = ix->ids[rw[j]];
        l0_off[t + 1] = e0;
        for (uint32_t l = 0; l < ix->level[i]; ++l) {
            uint64_t ur = ix->up_base[i] + l;
            const uint32_t *uw = ix->up + (size_t)ur * ix->su;
            for (uint32_t j = 0; j < ix->up_deg[ur]; ++j) up_nb[ue++] = ix->ids[uw[j]];
            up_off[++r] = ue;
        }
    }
    free(ord);
    return ORC_OK;
}

/* non-strict layer-0 arms (SimHash filter, sampling, adaptive bypass) */
#include "hvx_oracle_adaptive.inc"
#include "hvx_oracle_restricted.inc"
