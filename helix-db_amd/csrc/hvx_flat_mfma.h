// hvx_flat_mfma.h -- launch interface shared by the two contraction kernels of the matrix-core exact scan
// (hvx_flat_mfma.hip: 128 x 128 tiles, writes the score matrix or filters; hvx_flat_tile.hip: 256 x 256 tiles, filters only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hvx {

struct MfmaArgs {
    const uint16_t *qhi, *qlo; // [bpad][dim]
    const uint32_t *subset;    // optional: scan position -> row (restricted scans); NULL = the rows themselves
    const void *rows;          // [n][dim] bf16, or fp8 codes
    const float *rowscale;     // [n] fp8 only
    const float *rowterm;      // [n]: |x|^2 (L2) or |x| (cosine)
    const float *qn2;          // [b]
    uint32_t dim, b, row0, nrows, metric;
    float *dist;               // [b][chunk_ld]
    uint32_t chunk_ld;
    // FILT launches: only scores below the query's running threshold leave the tile, as (score, row) pairs
    const float *thr;          // [b] approximate score of the (m+1)-th candidate so far (+inf while fewer are known)
    float *cand_sc;            // [b][cand_cap]
    uint32_t *cand_id;         // [b][cand_cap] internal ids
    uint32_t *cand_cnt;        // [b] pairs appended (may exceed cand_cap: overflow, detected by the merge)
    uint32_t cand_cap;
    // 1-D launch: workgroup id -> (row tile, query tile).  Rows are walked in groups of `group_tiles` row tiles (~64 MB of
    // rows: they stay in the Infinity Cache), inside a group the QUERY tile is the outer loop: a query tile's 128 x dim
    // operand stays in L2 while the group's row tiles stream past it, and the group's rows come from HBM once.
    uint32_t nq_tiles, nr_tiles, group_tiles;
    // 256 x 256 kernel: XCD-aware super-tiles (hvx_flat_tile.hip)
    uint32_t sup_q, sup_r, sup_qblocks;
    const uint32_t *qexp; // MX build of the fp8 tile kernel: [bpad] E8M0 scale of a query's hi codes (a.qhi = the e4m3 operand there)
    uint32_t ablate; // measurement only (HVX_FLAT_TILE_ABLATE, two-buffer build): 1 no operand copies after stage 0, 2 no MFMAs, 4 no epilogue
};

// 256 x 256 filtered contraction (hvx_flat_tile.hip).  kind: 0 = bf16 rows (a bf16 index, or the bf16 shadow of an f32
// index; a.rows in the order a.qhi uses), 1 = fp8 codes (a.qhi in tile order, see tile_slot_fp8).  a.dim % 64 == 0.
// `wg_overflow` is set when a workgroup's pair list overflowed (the caller repeats the scan unfiltered).
// build: 0 = two 256-thread workgroups per CU (256 x 128 tiles), 1 = one 512-thread workgroup per CU (256 x 256 tiles), 2 = role-split,
// 3 (fp8 codes only) = the MX-scaled fp8 build: a.qhi = two e4m3 pieces per query value, a.qexp their scales
hipError_t launch_flat_tile256(const MfmaArgs &a, int kind, uint32_t bpad, float xmax2, uint32_t *wg_overflow, uint32_t build, hipStream_t s);

// one-pass score matrix for small batches (hvx_flat_smallb.hip): b <= 128 queries, kind 0 = bf16 rows / shadow, 2 = f32 rows
bool flat_smallb_supported(uint32_t dim, uint32_t b, int kind);
hipError_t launch_flat_smallb(const MfmaArgs &a, int kind, bool full, uint32_t cus, uint32_t build, hipStream_t s); // raw dot products into a.dist; build 1 = never the ring build
// sort-free selection over those dot products: every slice's kc (<= 256) smallest approximate scores as (score, row) pairs
hipError_t launch_flat_select_radix(const MfmaArgs &a, uint32_t kc, uint32_t *status, float *sl_sc, uint32_t *sl_id, uint32_t sl_stride,
                                    uint32_t *out_slices, hipStream_t s);

// position of stored code `slot` (its index in the fp8 row) in the query operand of the 256 x 256 fp8 kernel: inside a
// 64-code stage, MFMA step kk (0..3), lane half h, element e read code (2 (kk >> 1) + h) * 16 + (kk & 1) * 8 + e, so one
// ds_read_b128 of the code tile feeds two steps.
__host__ __device__ inline uint32_t tile_slot_fp8(uint32_t slot) {
    const uint32_t c = slot & 63u, u = c >> 4, sb = (c >> 3) & 1u, e = c & 7u;
    const uint32_t kk = (u >> 1) * 2u + sb, h = u & 1u;
    return (slot & ~63u) + kk * 16u + h * 8u + e;
}

#if defined(__HIPCC__)
// histogram increment with the wavefront's most common digits combined first: approximate scores of one query share their
// leading bytes, and 64 lanes adding to ONE LDS word serialise (measured: the four selection rounds of a 4 096-score slice cost
// 26 us with per-lane atomics alone, the rest of the kernel 12; two fixed rounds of combining still left 40-70 us kernels on the
// topic-ordered C3 corpus, whose scores share three bytes).
__device__ __forceinline__ void radix_count(uint32_t *hist, bool active, uint32_t digit) {
    // groups of equal digits are taken out one at a time (the first active lane's digit: one add for all lanes holding it) for as long
    // as they are BIG: a group of a dozen lanes costs a dozen serialised adds against one round here, a group of four costs the same
    // either way, and a wavefront of many small groups is served best by its per-lane adds (the clustered corpus: a 12-round loop that
    // went on at >= 4 lanes per group was 15-40 % slower than this; the topic-ordered C3 corpus has the big groups)
    for (int round = 0; round < 6; ++round) {
        const unsigned long long todo = __ballot(active);
        if (!todo) return;
        if (__builtin_popcountll(todo) < 12) break;
        const int leader = __builtin_ctzll(todo);
        const uint32_t pivot = (uint32_t)__shfl((int)digit, leader, 64);
        const bool same = active && digit == pivot;
        const unsigned long long votes = __ballot(same);
        if ((int)(threadIdx.x & 63u) == leader) atomicAdd(&hist[pivot], (uint32_t)__builtin_popcountll(votes));
        active = active && !same;
        if (__builtin_popcountll(votes) < 12) break;
    }
    if (active) atomicAdd(&hist[digit], 1u);
}

// One step of an 8-bit radix selection, run by EVERY wavefront of the workgroup on the same 256-bin histogram (no broadcast, no
// extra barrier): the digit g whose bin holds the kk-th smallest key (1-based) among the keys counted in `hist`, and the number
// of counted keys with a smaller digit.  kk <= the histogram's total by construction of the callers.
__device__ __forceinline__ void radix_digit_of_rank(const uint32_t *hist, uint32_t kk, uint32_t &g, uint32_t &below) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint4 c = reinterpret_cast<const uint4 *>(hist)[lane]; // bins 4 lane .. 4 lane + 3
    const uint32_t s = c.x + c.y + c.z + c.w;
    uint32_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += up;
    }
    const uint32_t excl = incl - s;
    const bool mine = excl < kk && kk <= incl;
    uint32_t gg = 0, bb = 0;
    if (mine) {
        uint32_t run = excl;
        gg = 4u * lane; bb = run;
        if (run + c.x < kk) { run += c.x; gg = 4u * lane + 1u; bb = run;
            if (run + c.y < kk) { run += c.y; gg = 4u * lane + 2u; bb = run;
                if (run + c.z < kk) { run += c.z; gg = 4u * lane + 3u; bb = run; } } }
    }
    const unsigned long long m = __ballot(mine);
    const int src = m ? __builtin_ctzll(m) : 0;
    g = __shfl(gg, src, 64);
    below = __shfl(bb, src, 64);
}
#endif

} // namespace hvx
