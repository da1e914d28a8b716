// hvx_hnsw_wave.h -- the fast HNSW search kernel: ONE wavefront per query, no workgroup barriers.
//
// Same algorithm, same results, same counters as hvx_hnsw.hip (SearchSession::run,
// crates/db/src/search/vector/search.rs:1101-1230; greedy upper layers :169-224; strict-exhaustive
// layer-0 beam :267-1067) -- restructured around what bounded the first kernel
// on MI355X (profiles/history/r01a: 11.6 us per expansion, three dependent HBM round trips + barriers):
//
//   * 1024 queries = 1024 wavefronts = one wave per SIMD on all 256 CUs: each wave owns a whole SIMD's
//     register file, so a full pass of P x 8 neighbour rows (P*NK 16-byte loads per lane, 8 lanes per
//     row, 128 B coalesced per load) is in flight before the first FMA;
//   * the visited set is an open-addressing hash table in LDS (ds_cmpst, ~100 cycles) instead of an
//     HBM bitmap (atomicOr round trip); it spills to the exact HBM bitmap if it ever fills;
//   * the next pop is predicted (right 97.7 % of the time on the 1M x 768 workload): 93 % of pops are the
//     next unexpanded entry already in the beam, whose neighbour row is requested AHEAD of the expansion's
//     gathers; if a fresh candidate beats it, that one's row is fetched underneath the admission loop;
//   * a frontier of up to 8, 16, 24 or 32 rows is gathered with ONE latency: the pass width is chosen per
//     expansion (1, 2, 3 or 4 x P rows per 8-lane group in flight), because the kernel ends with its
//     slowest query and the slow queries are the ones with large frontiers (profiles/history/r01c);
//   * distances keep the host SIMD summation order (hvx_device.h) => scores are bit-identical to the
//     reference CPU path; the beam (hvx_beam.h) is exact.
//
// Served shapes: f32 or bf16 rows, metric L2 / cosine, AVX+FMA summation tree, dim = 32*NK with NK in
// {4,8,12,16,24,32,48} (dim 128 ... 1536; 12 = dim 384 since round 6), neighbour rows <= 64 ids, ef + 32 <= 384.  Everything else runs on the general
// kernel in hvx_hnsw.hip.
#pragma once
#include <type_traits>

#include "hvx_beam.h"
#include "hvx_device.h"
#include "hvx_kernels.h"

namespace hvx {

constexpr uint32_t kTabEmpty = 0xFFFFFFFFu;
constexpr uint32_t kTentativeBit = 0x80000000u; // visited-table entry claimed by the non-strict arms, not (yet) visited
constexpr uint32_t kHashedBit = 0x40000000u;    // ... whose SimHash row is already in the query-local cache (uncached-handle read accounting)

__device__ __forceinline__ uint32_t umin_dpp_row(uint32_t v) {
    // butterfly inside each 16-lane DPP row: xor1, xor2, half-mirror (7-l), mirror (15-l)
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); v = v < t ? v : t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); v = v < t ? v : t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true); v = v < t ? v : t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true); v = v < t ? v : t;
    return v;
}
// wave-wide unsigned minimum, result uniform
__device__ __forceinline__ uint32_t wave_umin(uint32_t v) {
    v = umin_dpp_row(v);
    uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    a = a < b ? a : b;
    c = c < d ? c : d;
    return a < c ? a : c;
}

// Visited set of one query: LDS hash table, exact; falls back to the HBM bitmap when 3/4 full.
// WGSYNC: the table is cleared behind a workgroup barrier (the one-wavefront kernels: the workgroup IS the wavefront); false: the
// owner wavefront of the pair kernel (hvx_hnsw_pair.h) works on it alone while its partner sits in a barrier of its own -- LDS
// operations of one wavefront execute in order, a wave-level fence is all it needs.
template <bool WGSYNC = true> struct VisitedT {
    __device__ __forceinline__ static void vsync() {
        if (WGSYNC) {
            __syncthreads();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
    uint32_t *tab;      // LDS [cap]
    uint32_t *bm;       // HBM bitmap of this query (all-zero on entry, handed back all-zero)
    uint32_t *bm2;      // second bitmap: SimHash-row-cached ids in bitmap mode (uncached-handle read accounting only; may be NULL)
    uint32_t cap, words; // cap: any slot count (a power of two is not required: first slot = high word of hash x cap)
    uint32_t limit;     // uniform: spill to the bitmap once count + 64 would exceed this many table entries
    uint32_t count;     // uniform: ids inserted into the table since the last clear
    bool spilled;       // uniform: bitmap mode

    __device__ __forceinline__ void clear(int lane) {
        if (spilled) {
            for (uint32_t w = (uint32_t)lane; w < words; w += 64) { bm[w] = 0u; if (bm2) bm2[w] = 0u; }
            spilled = false;
        }
        for (uint32_t i = (uint32_t)lane; i < cap; i += 64) tab[i] = kTabEmpty;
        count = 0;
        vsync();
    }
    __device__ __forceinline__ void spill(int lane) {
        for (uint32_t i = (uint32_t)lane; i < cap; i += 64) {
            uint32_t v = tab[i];
            if (v == kTabEmpty) continue;
            const uint32_t vid = v & ~(kTentativeBit | kHashedBit);
            if (!(v & kTentativeBit)) atomicOr(&bm[vid >> 5], 1u << (vid & 31u));
            else if ((v & kHashedBit) && bm2) atomicOr(&bm2[vid >> 5], 1u << (vid & 31u));
        }
        spilled = true;
        __threadfence_block();
    }
    // Non-strict arms: a neighbour becomes visited only when it is filtered or admitted, later in the expansion.
    // claim() probes ONCE: an unknown id takes an empty slot as TENTATIVE (bit 31; tentative ids count as unseen, also for
    // later expansions, and keep their slot), commit() turns the remembered slot into a visited entry with one plain
    // store -- one LDS probe loop per expansion instead of a test loop plus an insert loop.  In bitmap mode claim() is
    // a bit test and commit() the atomicOr.  Returns true for unseen ids; ids must be < 2^31 - 1.
    __device__ __forceinline__ bool claim(uint32_t id, bool valid, int lane, uint32_t &slot_out, bool &hashed_out) {
        if (!spilled && count + 64u > limit) spill(lane);
        hashed_out = false;
        if (spilled) {
            if (valid && bm2) hashed_out = (__hip_atomic_load(&bm2[id >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (id & 31u)) & 1u;
            return !contains(id, valid) & valid;
        }
        bool pending = valid, unseen = false, taken = false;
        uint32_t slot = __umulhi(id * 2654435761u, cap);
        while (__ballot(pending)) {
            if (pending) {
                const uint32_t old = atomicCAS(&tab[slot], kTabEmpty, id | kTentativeBit);
                if (old == kTabEmpty) { unseen = true; taken = true; pending = false; }
                else if ((old & ~(kTentativeBit | kHashedBit)) == id) {
                    unseen = (old & kTentativeBit) != 0u;
                    hashed_out = (old & kHashedBit) != 0u;
                    pending = false;
                }
                else slot = slot + 1u == cap ? 0u : slot + 1u;
            }
        }
        slot_out = slot;
        count += (uint32_t)__builtin_popcountll(__ballot(taken));
        return unseen;
    }
    __device__ __forceinline__ void commit(uint32_t id, bool doit, uint32_t slot) {
        if (spilled) {
            if (doit) atomicOr(&bm[id >> 5], 1u << (id & 31u));
        } else if (doit) {
            tab[slot] = id;
        }
    }
    // the id's SimHash row entered the query-local cache (it stays a tentative, unseen id)
    __device__ __forceinline__ void mark_hashed(uint32_t id, bool doit, uint32_t slot) {
        if (spilled) {
            if (doit && bm2) atomicOr(&bm2[id >> 5], 1u << (id & 31u));
        } else if (doit) {
            tab[slot] = id | kTentativeBit | kHashedBit;
        }
    }
    // membership test only
    __device__ __forceinline__ bool contains(uint32_t id, bool valid) const {
        if (spilled) {
            bool hit = false;
            if (valid) hit = (__hip_atomic_load(&bm[id >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (id & 31u)) & 1u;
            return hit;
        }
        bool pending = valid, found = false;
        uint32_t slot = __umulhi(id * 2654435761u, cap);
        while (__ballot(pending)) {
            if (pending) {
                const uint32_t v = tab[slot];
                if (v == id) { found = true; pending = false; }
                else if (v == kTabEmpty) pending = false;
                else slot = slot + 1u == cap ? 0u : slot + 1u;
            }
        }
        return found;
    }
    // test-and-set for the lanes with valid==true (ids distinct across lanes); true = newly inserted
    __device__ __forceinline__ bool insert(uint32_t id, bool valid, int lane) {
        if (!spilled && count + 64u > limit) spill(lane);
        bool isnew = false;
        if (spilled) {
            if (valid) {
                uint32_t bit = 1u << (id & 31u);
                uint32_t old = atomicOr(&bm[id >> 5], bit);
                isnew = !(old & bit);
            }
            return isnew;
        }
        bool pending = valid;
        uint32_t slot = __umulhi(id * 2654435761u, cap);
        while (__ballot(pending)) {
            if (pending) {
                uint32_t old = atomicCAS(&tab[slot], kTabEmpty, id);
                if (old == kTabEmpty) { isnew = true; pending = false; }
                else if (old == id) { pending = false; }
                else slot = slot + 1u == cap ? 0u : slot + 1u;
            }
        }
        count += (uint32_t)__builtin_popcountll(__ballot(isnew));
        return isnew;
    }
};
using Visited = VisitedT<true>;

// One gather pass in flight: P rows per 8-lane group.  f32 rows: NK 16-byte loads per lane and row;
// bf16 rows (interleaved layout, hvx_device.h): NK/2 loads, each carrying the lane's virtual lanes of
// two consecutive chunks.
template <int NK, int P, bool BF> struct Gather {
    float4 x[P][BF ? NK / 2 : NK];
    float h[P]; // cosine: the row's cached norm header, requested together with the row
};

// Straight-line issue of every load of a pass.  Groups without a row of their own are pointed at
// another group's row by the caller (identical addresses coalesce inside the load instruction): no
// divergence, nothing for the compiler to sink behind a branch.
template <int NK, int P, bool BF, int PMAX = P, bool HDR = false>
__device__ __forceinline__ void gather_issue(const DevIndex &ix, const uint32_t (&node)[P], int slot, Gather<NK, PMAX, BF> &g) {
    constexpr int NL = BF ? NK / 2 : NK;
    if (HDR) { // ahead of the rows: returns in order, so it is there when the reduction tree ends
#pragma unroll
        for (int p = 0; p < P; ++p) g.h[p] = ix.hdr[node[p]];
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float4 *rp = BF ? reinterpret_cast<const float4 *>(ix.vecb + (size_t)node[p] * ix.dim) + slot
                              : reinterpret_cast<const float4 *>(ix.vec + (size_t)node[p] * ix.ld) + slot;
#pragma unroll
        for (int k = 0; k < NL; ++k) g.x[p][k] = rp[k * 8];
    }
    __builtin_amdgcn_sched_barrier(0); // every load is issued before anything that follows is scheduled
}

template <uint32_t METRIC> __device__ __forceinline__ void fma_chunk(float4 &acc, const float4 qq, const float4 xv) {
    if (METRIC == kL2) {
        const float d0 = qq.x - xv.x, d1 = qq.y - xv.y, d2 = qq.z - xv.z, d3 = qq.w - xv.w;
        acc.x = __builtin_fmaf(d0, d0, acc.x); acc.y = __builtin_fmaf(d1, d1, acc.y);
        acc.z = __builtin_fmaf(d2, d2, acc.z); acc.w = __builtin_fmaf(d3, d3, acc.w);
    } else {
        acc.x = __builtin_fmaf(qq.x, xv.x, acc.x); acc.y = __builtin_fmaf(qq.y, xv.y, acc.y);
        acc.z = __builtin_fmaf(qq.z, xv.z, acc.z); acc.w = __builtin_fmaf(qq.w, xv.w, acc.w);
    }
}

// FMAs in the host SIMD order against the query staged in LDS (conflict-free: the 8 lanes of a group
// read 128 contiguous bytes, the 8 groups broadcast), then the AVX reduction tree.
template <uint32_t METRIC, int NK, int P, bool BF, int PMAX = P>
__device__ __forceinline__ void gather_consume(const DevIndex &ix, const float *qs, const Gather<NK, PMAX, BF> &g,
                                               const uint32_t (&node)[P], int slot, float qhdr,
                                               const float *qglobal, float (&out)[P]) {
    float4 acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 *qp = reinterpret_cast<const float4 *>(qs) + slot;
    if (!BF) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const float4 qq = qp[k * 8];
#pragma unroll
            for (int p = 0; p < P; ++p) fma_chunk<METRIC>(acc[p], qq, g.x[p][k]);
        }
    } else {
#pragma unroll
        for (int m = 0; m < NK / 2; ++m) {
            const float4 q0 = qp[(2 * m) * 8], q1 = qp[(2 * m + 1) * 8];
#pragma unroll
            for (int p = 0; p < P; ++p) { // bf16 -> f32 is exact: the halfword becomes the high half of the word
                const uint32_t w0 = __float_as_uint(g.x[p][BF ? m : 0].x), w1 = __float_as_uint(g.x[p][BF ? m : 0].y);
                const float4 x0 = make_float4(__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xFFFF0000u),
                                              __uint_as_float(w1 << 16), __uint_as_float(w1 & 0xFFFF0000u));
                fma_chunk<METRIC>(acc[p], q0, x0);
            }
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const uint32_t w2 = __float_as_uint(g.x[p][BF ? m : 0].z), w3 = __float_as_uint(g.x[p][BF ? m : 0].w);
                const float4 x1 = make_float4(__uint_as_float(w2 << 16), __uint_as_float(w2 & 0xFFFF0000u),
                                              __uint_as_float(w3 << 16), __uint_as_float(w3 & 0xFFFF0000u));
                fma_chunk<METRIC>(acc[p], q1, x1);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        float r = avx_tree_reduce(acc[p]);
        if (METRIC == kCosine) {
            const uint32_t nd = node[p];
            r = cosine_finish_fn(r, qhdr, g.h[p], [&]() {
                if (BF) {
                    const uint16_t *rb = ix.vecb + (size_t)nd * ix.dim;
                    return stable_half_cosine_fn(ix.dim, [&](uint32_t i) { return qglobal[i]; },
                                                 [&](uint32_t i) { return bf16_to_f32(rb[bf16_slot_of(i)]); });
                }
                const float *rf = ix.vec + (size_t)nd * ix.ld;
                return stable_half_cosine_fn(ix.dim, [&](uint32_t i) { return qglobal[i]; }, [&](uint32_t i) { return rf[i]; });
            });
        }
        out[p] = r;
    }
}


// ------------------------------------------------------------------------------------------------------------
// Non-strict layer-0 arms (search.rs:595-829, policy.rs, randomness.rs)
// ------------------------------------------------------------------------------------------------------------

// The query-local generator of SearchSession (randomness.rs:127-164): rand 0.10 StdRng = ChaCha12 keyed by
// seed_from_u64's PCG32 expansion (pinned by the SimHasher known answer, hvx_simhash.hip).  The stream is consumed
// strictly in order, but the NUMBER of draws of a sampling stage is known before the stage runs, so the words are
// produced a window of blocks at a time -- lane L computes block base/16 + L -- and parked in LDS, and a stage
// hands word (pos + rank) to the lane whose candidate is the rank-th one to draw.
// RW = words of the LDS window: 1 024 (64 blocks, one per lane) for the one-query-per-SIMD builds; 256 (16 blocks, lanes 0..15) for
// the two-per-SIMD builds, whose 20 KiB of LDS are better spent on the visited table -- the production-default parameters draw
// ~120 words per query (profiles/history/r04s), a window is rarely refilled either way.
constexpr uint32_t kRngWords = 1024;
constexpr uint32_t kRngWordsOcc2 = 256;
template <uint32_t RW> struct QueryRngT {
    static constexpr uint32_t kBlocks = RW / 16u;
    uint32_t *buf;   // LDS [RW], word w of the window at buf[(w % 16) * kBlocks + w / 16] (conflict-free fill)
    uint32_t key[8]; // uniform
    uint32_t base;   // stream index of the window's first word
    uint32_t pos;    // words consumed so far (uniform)
    bool ready;

    __device__ __forceinline__ void seed(uint64_t state) { // rand_core SeedableRng::seed_from_u64
        const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            state = state * MUL + INC;
            const uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
            key[i] = (xs >> rot) | (xs << ((32u - rot) & 31u));
        }
        base = 0;
        pos = 0;
        ready = false; // SearchSession::seeded: nothing is generated until the first draw
    }
    // make words [pos, pos + 66) readable (a stage draws <= 64 words, choose_index <= 2)
    __device__ __forceinline__ void ensure(int lane) {
        if (ready && pos + 66u <= base + RW) return;
        base = pos & ~15u;
        const uint32_t ctr = (base >> 4) + (uint32_t)lane; // block counter (64-bit in ChaCha; < 2^32 here)
        uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                           key[4], key[5], key[6], key[7], ctr, 0u, 0u, 0u};
        uint32_t x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = st[i];
#define HVX_QR(a, b, c, d)                                                                                   \
    x[a] += x[b]; x[d] = __builtin_rotateleft32(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = __builtin_rotateleft32(x[b] ^ x[c], 12); \
    x[a] += x[b]; x[d] = __builtin_rotateleft32(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = __builtin_rotateleft32(x[b] ^ x[c], 7)
#pragma unroll 1
        for (int r = 0; r < 6; ++r) { // 12 rounds
            HVX_QR(0, 4, 8, 12); HVX_QR(1, 5, 9, 13); HVX_QR(2, 6, 10, 14); HVX_QR(3, 7, 11, 15);
            HVX_QR(0, 5, 10, 15); HVX_QR(1, 6, 11, 12); HVX_QR(2, 7, 8, 13); HVX_QR(3, 4, 9, 14);
        }
#undef HVX_QR
        __syncthreads(); // every reader of the previous window is done
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (kBlocks == 64u || (uint32_t)lane < kBlocks) buf[i * (int)kBlocks + lane] = x[i] + st[i];
        ready = true;
        __syncthreads();
    }
    __device__ __forceinline__ uint32_t word(uint32_t idx) const {
        const uint32_t r = idx - base;
        return buf[(r & 15u) * kBlocks + (r >> 4)];
    }
    // random::<f32>() < ratio for stream word idx
    __device__ __forceinline__ bool below(uint32_t idx, float ratio) const {
        return (float)(word(idx) >> 8) * (1.0f / 16777216.0f) < ratio;
    }
    // SearchSession::choose_index (randomness.rs:162-164) -> rand `random_range(0..n)`, n <= 64: high word of
    // x*n with one bias-correction draw (restated; the reference holds no known answer for it).  Uniform.
    __device__ __forceinline__ uint32_t choose_index(uint32_t n, int lane) {
        ensure(lane);
        const unsigned long long m = (unsigned long long)word(pos) * n;
        ++pos;
        uint32_t hi = (uint32_t)(m >> 32);
        const uint32_t lo = (uint32_t)m;
        if (lo > 0u - n) {
            const uint32_t nhi = (uint32_t)(((unsigned long long)word(pos) * n) >> 32);
            ++pos;
            hi += (lo + nhi < lo) ? 1u : 0u;
        }
        return hi;
    }
};

__device__ __forceinline__ float rs_clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); } // f32::clamp

// one decision epoch (Layer0Policy::decide, policy.rs:119-183); everything uniform
struct AdaptDecision {
    bool filter;          // fetch_missing == filter_cached
    uint32_t threshold;   // active collision threshold (0 when not filtering)
    uint32_t pre_kind;    // SamplingDecision of the pre-sampling stage: 0 Exhaustive, 1 Fixed
    float pre_p;
    uint32_t samp_kind;   // 0 Exhaustive, 1 Fixed, 2 Adaptive (after activation)
    float samp_p;         // probability() of that decision (1.0 for Exhaustive)
    float base_p;
    bool bypassed;
    uint32_t trigger;
};

// per-query state of the non-strict arms (uniform) + SearchStats counters
struct AdaptState {
    uint32_t fill;                        // simhash_fill_slots (search.rs:532)
    uint32_t win_ex, win_filt, win_exp;   // rolling yield window (search.rs:790-801)
    uint32_t bstate, bremain;             // AdaptiveBypassState: 0 Ready, 1 Bypassing, 2 CoolingDown
    uint32_t reads;                       // _txn_get_simhash_filter: stable-view SimHash reads (uncached handle only)
    hvx_adaptive_stats st;
};

__device__ __forceinline__ float adaptive_sampling_ratio_fn(float base, uint32_t wlen, uint32_t ef, float current, float delta) {
    if (base >= 1.0f) return base;                                   // policy.rs:559-575
    const uint32_t lim = ef / 3u > 8u ? ef / 3u : 8u;
    if (wlen < lim) return 1.0f;
    if (delta <= 1e-6f) return base;
    const float rq = rs_clampf(1.0f - rs_clampf(current / delta, 0.0f, 1.0f), 0.0f, 1.0f);
    const float u = (1.0f - base) * rq;
    const float v = rs_clampf(base + u, base, 1.0f);
    return __builtin_fminf(v, __builtin_fmaxf(0.90f, base));
}

__device__ __forceinline__ AdaptDecision adapt_decide(const AdaptArgs &p, AdaptState &s, uint32_t ef, bool topk_ready, uint32_t wlen,
                                                      uint32_t frontier, float current, float delta, float brk_lane) {
    AdaptDecision d;
    // AdaptiveBypassPolicy::decide (policy.rs:228-296); the device index is the resident snapshot: 0 SimHash reads
    bool bypassed = false;
    uint32_t trigger = 0, nstate = 0, nremain = 0;
    if (p.bypass_windowed) {
        if (s.bstate == 1u) {
            bypassed = true;
            nstate = s.bremain - 1u ? 1u : 2u;
            nremain = s.bremain - 1u ? s.bremain - 1u : p.window_expansions;
        } else if (s.bstate == 2u && s.bremain > 1u) {
            nstate = 2u;
            nremain = s.bremain - 1u;
        } else {
            const float rate = s.win_ex == 0u ? 1.0f : (float)s.win_filt / (float)s.win_ex;
            const bool low_yield = (s.win_exp >= p.window_expansions) & (rate < p.min_filter_rate);
            const bool budget = (p.count_reads != 0u) & (s.reads >= p.read_budget); // a resident snapshot reads nothing
            if (frontier >= p.min_frontier && (low_yield | budget)) {
                bypassed = true;
                trigger = (budget ? 1u : 0u) | (low_yield ? 2u : 0u); // ReadBudget / LowYield / both
                nstate = p.window_expansions - 1u ? 1u : 2u;
                nremain = p.window_expansions - 1u ? p.window_expansions - 1u : p.window_expansions;
            }
        }
    }
    s.bstate = nstate;
    s.bremain = nremain;
    // sampling (policy.rs:123-138, 526-557)
    float base_p = 1.0f;
    uint32_t kind = p.sampling;
    if (kind == 1u) base_p = p.ratio;
    else if (kind == 2u) base_p = adaptive_sampling_ratio_fn(p.ratio, wlen, ef, current, delta);
    const uint32_t lim = ef / 4u > 8u ? ef / 4u : 8u;
    d.base_p = base_p;
    d.samp_kind = kind;
    d.samp_p = base_p;
    if (!(base_p <= 0.0f || base_p >= 1.0f || frontier > lim)) { d.samp_kind = 0u; d.samp_p = 1.0f; }
    if (kind == 0u) d.samp_p = 1.0f;
    const float pre_base = p.pre_override >= 0.0f ? p.pre_override : base_p;
    if (pre_base >= 1.0f || frontier <= lim) {
        d.pre_kind = 0u;
        d.pre_p = 1.0f;
    } else {
        float r = rs_clampf(pre_base * 0.65f, 0.25f, 0.9f);
        const uint32_t wide = ef * 2u > 32u ? ef * 2u : 32u;
        if (pre_base <= 0.0f) r = 0.0f;
        else if (frontier > wide) r = __builtin_fmaxf(r * 0.8f, 0.20f);
        d.pre_kind = 1u;
        d.pre_p = r;
    }
    d.bypassed = bypassed;
    d.trigger = trigger;
    d.filter = !bypassed && p.filtering != 0u;
    d.threshold = 0u;
    if (d.filter) {
        if (p.filtering == 1u) d.threshold = p.configured;
        else if (p.configured == 0u) d.threshold = 0u;
        else if (!topk_ready) d.threshold = 1u;
        else d.threshold = (uint32_t)__builtin_popcountll(__ballot(delta <= brk_lane)); // adaptive_threshold as a table
    }
    return d;
}

// SamplingDecision::candidate_probability (policy.rs:417-433), per lane
__device__ __forceinline__ float candidate_probability_fn(uint32_t kind, float base, uint32_t sim_bits, bool has_thr, uint32_t thr) {
    if (kind != 2u) return base;
    if (base <= 0.0f || base >= 1.0f) return base;
    const float sr = (float)(sim_bits < 64u ? sim_bits : 64u) / 64.0f;
    const float tr = has_thr ? (float)thr / 64.0f : 0.0f;
    const float t = __builtin_fmaxf(sr - tr, 0.0f);
    const float u = (1.0f - base) * t;
    return rs_clampf(base + u, base, 1.0f);
}

// PROF=true builds the phase-timing variant (s_memtime around each phase of a layer-0 expansion,
// hard waits at the phase boundaries); it is only launched when HVX_WAVE_PROF is set.
// AD=true builds the non-strict layer-0 arms (SimHash filter, pre/post sampling, adaptive bypass) into the same
// beam search; AD=false is the strict-exhaustive arm and compiles to exactly the code it was before.
// ST=false drops the SearchStats counters of the non-strict stages (the reference's COLLECT_DIAGNOSTICS=false
// specialisation, search.rs:267-270): seventeen fewer live scalars, which is what keeps the AD kernels out of SGPR spills.
// OCC = wavefronts per SIMD the build is register-budgeted for: 1 = one query owns the SIMD's whole 512-register file
// (passes of up to 32 rows in flight), 2 = two queries share a SIMD (256 registers each, passes of up to 16 rows, half the
// LDS): the row gathers of one query run underneath the beam bookkeeping of the other -- the build for callers that keep
// >= 2 batches in flight (execution lanes, hvx_index_fork).
// BUILD = the search side of a batched insert_hnsw (mutation.rs:787-895; hvx_build.hip): the "query" of wavefront q is the
// stored row of node build_nodes[q], the descent is greedy above min(level, max_layer) and a full beam
// (search_layer_beam, mutation.rs:904-1005 -- the same algorithm as the strict layer-0 search) on every layer from
// there down to 0; the first k entries of every layer's W (internal ids, scores) are written per layer.
// One query's whole search on one wavefront (q = blockIdx.x, or the query a re-run workgroup picked from the re-run list).
template <uint32_t METRIC, int R, int NK, bool BF, bool PROF, bool AD, bool ST, int OCC, bool BUILD>
__device__ __forceinline__ void hnsw_wave_query(const HnswArgs &a, const uint32_t vcap, const uint32_t q) {
    // rows per 8-lane group in flight: P*NK <= 24 float4 per lane for a narrow pass, x2 and x4 for wider
    // frontiers (4P*NK <= 96 float4 = 384 registers, VGPR+AGPR file of one wave per SIMD)
    // NK == 0: the GENERIC build -- any dimension (scalar tail, simple_avx.rs:172-177), any metric (Manhattan's sequential
    // order, simple.rs:186-202), the AVX / AVX+FMA / scalar summation trees: rows are scored one per 8-lane group with
    // hvx_device.h's group_distance instead of the unrolled register passes.  Same beam, visited set and non-strict arms.
    constexpr bool GEN = NK == 0;
    constexpr int NL = GEN ? 1 : (BF ? NK / 2 : NK);        // 16-byte loads per lane and row
    constexpr int P = NL <= 8 ? 2 : 1;
    // (cosine over bf16 rows at dim 1536: the widened pieces + norm headers of a 4P-row pass spilled 70-240 registers in rounds 3-4 --
    // VERDICT r4 weak #6; those builds keep at most 2P rows (one per SIMD) / P rows (two per SIMD) in flight and nothing in scratch)
    constexpr bool kCosBfBig = BF && METRIC == kCosine && NL >= 24;
    constexpr bool kWide4 = OCC == 1 && 4 * P * NL <= 96 && !kCosBfBig;
    // 192 of the 256 registers of a half-SIMD wave.  The cosine non-strict build at dim 768 keeps ONE row per group in flight: with
    // two its per-lane filter / sampling state spills 17-34 registers to scratch and the spilled build is 11 % slower on four lanes
    // (profiles/r05b_ad_lanes_ab.log: 0.754 -> 0.673 ms per step; the squared-Euclidean build has no filter state and keeps two rows).
    // (bf16 rows widen every 16-byte piece to two float4 in front of the FMAs: their non-strict builds spill from 32 float4 per pass on)
    constexpr bool kWide2 = OCC == 1 || (2 * P * NL <= 48 && !kCosBfBig && !(AD && ((METRIC == kCosine && 2 * P * NL > 32) || (BF && 2 * P * NL >= 32))));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevIndex &ix = a.ix;
    const int lane = (int)threadIdx.x, grp = lane >> 3, j = lane & 7;
    const int slot = chunk_slot(j);

    Visited V;
    V.tab = reinterpret_cast<uint32_t *>(smem);
    V.cap = vcap;
    V.limit = OCC == 1 ? V.cap - (V.cap >> 2) : V.cap - (V.cap >> 3); // 3/4 full; 7/8 for the half-LDS build
    V.words = a.words_per_query;
    V.bm = a.bitmap + (size_t)q * a.words_per_query;
    V.bm2 = (AD && a.bitmap2) ? a.bitmap2 + (size_t)q * a.words_per_query : nullptr;
    V.count = 0;
    V.spilled = false;
    uint32_t *fr_id = V.tab + V.cap;                      // [64] frontier ids, row order
    float *fr_d = reinterpret_cast<float *>(fr_id + 64);   // [64] their distances
    float *qs = fr_d + 64;                                 // [dim] query, 16-byte aligned
    uint32_t *rng_buf = reinterpret_cast<uint32_t *>(qs + (GEN ? ix.ld : (uint32_t)NK * 32u)); // [kRngWords] (AD only)

    const unsigned long long wclk0 = (a.wave_clock && !a.only_flagged) ? wall_clock64() : 0ull;
    const uint32_t status_in = BUILD ? 0u : (a.qstatus ? a.qstatus[q] : 0u);
    if (status_in != 0u || !ix.has_entry) {
        if (lane == 0) {
            if (a.wave_clock) { a.wave_clock[2 * (size_t)q] = wclk0; a.wave_clock[2 * (size_t)q + 1] = wclk0; }
            a.out_counts[q] = 0;
            if (a.out_status) a.out_status[q] = status_in;
            if (a.qstats) a.qstats[q] = hvx_query_stats{0, 0, 0, 0};
            if (a.tie_flags) a.tie_flags[q] = 0u;
            if (AD && a.ad.stats) a.ad.stats[q] = hvx_adaptive_stats{};
        }
        return;
    }
    const uint32_t bnode = BUILD ? a.build_nodes[q] : 0u;
    // BUILD over bf16 rows (round 6: one-node inserts / upserts into bf16 images): the node's ROUNDED vector arrives as an f32 query
    // (a.queries, unused by f32 build searches), its stored row is bf16
    const float *qglobal = (BUILD && ix.vec) ? ix.vec + (size_t)bnode * ix.ld : a.queries + (size_t)q * ix.dim;
    if (GEN) { // zero padded to ld floats, as group_distance reads it
        for (uint32_t i = (uint32_t)lane; i < ix.ld; i += 64) qs[i] = i < ix.dim ? qglobal[i] : 0.f;
    } else {
        for (uint32_t i = (uint32_t)lane; i < (uint32_t)NK * 8u; i += 64)
            reinterpret_cast<float4 *>(qs)[i] = reinterpret_cast<const float4 *>(qglobal)[i];
    }
    __syncthreads();
    const float qhdr = BUILD ? ix.hdr[bnode] : (a.qhdr ? a.qhdr[q] : 0.f);
    const float inf = __uint_as_float(0x7F800000u);

    // one pass of W rows per group over fr_id[f0..nf): issue everything, then FMA, then publish
    // GENERIC build: one row per 8-lane group and step, the general distance evaluator
    auto generic_distance = [&](uint32_t node) __attribute__((always_inline)) -> float {
        if (kernel_fused(ix.fkernel)) return group_distance<METRIC, true>(ix, qs, qhdr, node, j);
        return group_distance<METRIC, false>(ix, qs, qhdr, node, j);
    };
    auto pass = [&](auto width, uint32_t f0, uint32_t nf) __attribute__((always_inline)) {
        constexpr int W = decltype(width)::value;
        uint32_t nd[W];
        float o[W];
        Gather<GEN ? 1 : NK, W, BF> g;
#pragma unroll
        for (int p = 0; p < W; ++p) {
            const uint32_t f = f0 + (uint32_t)(p * 8 + grp);
            nd[p] = fr_id[f < nf ? f : f0]; // idle groups shadow the pass's first row
        }
        if constexpr (!GEN) {
            gather_issue<NK, W, BF, W, METRIC == kCosine>(ix, nd, slot, g);
            gather_consume<METRIC, NK, W, BF>(ix, qs, g, nd, slot, qhdr, qglobal, o);
        }
#pragma unroll
        for (int p = 0; p < W; ++p) {
            const uint32_t f = f0 + (uint32_t)(p * 8 + grp);
            if (f < nf && j == 0) fr_d[f] = o[p];
        }
    };
    // distances of fr_id[0..nf) -> fr_d, widest pass that the remaining rows fill
    auto score_frontier = [&](uint32_t nf) __attribute__((always_inline)) {
        uint32_t f0 = 0;
        if constexpr (GEN) {
            for (; f0 < nf; f0 += 8) {
                const uint32_t f = f0 + (uint32_t)grp;
                const float d = generic_distance(fr_id[f < nf ? f : f0]);
                if (f < nf && j == 0) fr_d[f] = d;
            }
            __syncthreads();
            return;
        }
        while (f0 < nf) {
            const uint32_t rem = nf - f0;
            if (kWide4 && rem > 24u * P) { pass(std::integral_constant<int, kWide4 ? 4 * P : P>{}, f0, nf); f0 += 32u * P; }
            else if (kWide4 && rem > 16u * P) { pass(std::integral_constant<int, kWide4 ? 3 * P : P>{}, f0, nf); f0 += 24u * P; }
            else if (kWide2 && rem > 8u * P) { pass(std::integral_constant<int, kWide2 ? 2 * P : P>{}, f0, nf); f0 += 16u * P; }
            else { pass(std::integral_constant<int, P>{}, f0, nf); f0 += 8u * P; }
        }
        __syncthreads();
    };
    // distance of ONE node, uniform result
    auto score_one = [&](uint32_t node) __attribute__((always_inline)) -> float {
        if constexpr (GEN) {
            return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(generic_distance(node)), 0));
        } else {
            uint32_t nd[1] = {node};
            float o[1];
            Gather<NK, 1, BF> g;
            gather_issue<NK, 1, BF, 1, METRIC == kCosine>(ix, nd, slot, g);
            gather_consume<METRIC, NK, 1, BF>(ix, qs, g, nd, slot, qhdr, qglobal, o);
            return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(o[0]), 0));
        }
    };
    // visited test-and-set + in-order compaction of one neighbour row held one id per lane
    auto frontier_from = [&](uint32_t nid, uint32_t &deg) __attribute__((always_inline)) -> uint32_t {
        const bool valid = nid != kSentinel;
        const bool isnew = V.insert(nid, valid, lane);
        const unsigned long long um = __ballot(isnew);
        deg = (uint32_t)__builtin_popcountll(__ballot(valid));
        if (isnew) fr_id[__builtin_popcountll(um & ((1ull << lane) - 1ull))] = nid;
        __syncthreads();
        return (uint32_t)__builtin_popcountll(um);
    };

    // non-strict arms: visited TEST only; the unvisited neighbours land in fr_id in row order
    // non-strict arms: ONE probe of the visited table claims the unseen neighbours tentatively (Visited::claim); the
    // selection stages then work in row-lane space (lane = position in the neighbour row = frontier order)
    uint32_t vslot = 0;
    bool vhashed = false;
    auto frontier_claim = [&](uint32_t nid, uint32_t &deg, unsigned long long &um) __attribute__((always_inline)) -> uint32_t {
        const bool valid = nid != kSentinel;
        const bool unseen = V.claim(nid, valid, lane, vslot, vhashed);
        um = __ballot(unseen);
        deg = (uint32_t)__builtin_popcountll(__ballot(valid));
        return (uint32_t)__builtin_popcountll(um);
    };

    bool bad_score = false;
    uint32_t cur = ix.entry;

    // ---------------- upper layers: search_layer_greedy (search.rs:169-224) ----------------
    // The reference recomputes distance(query, current) when it enters a layer (search.rs:177-181): same inputs, same
    // value, so it is computed once and carried down.  The upper-row pointer and the level of every candidate of a hop
    // are requested together with the candidates' rows: the next hop's neighbour row is one round trip away instead of two.
    float cur_d = 0.f;
    bool have_d = false;
    uint32_t cur_base = kSentinel, cur_level = 0;
    if (ix.max_layer >= 1) {
        cur_base = ix.up_base[cur];
        cur_level = ix.level[cur];
    }
    uint32_t beam_top = 0; // BUILD: the beam runs on every layer from min(node level, max_layer) down to 0
    if (BUILD) {
        const uint32_t lv = ix.level[bnode];
        beam_top = lv < ix.max_layer ? lv : ix.max_layer;
    }
    for (uint32_t layer = ix.max_layer; layer >= 1 && layer > beam_top; --layer) {
        V.clear(lane); // fresh visited set per layer
        if (!have_d) {
            cur_d = score_one(cur);
            have_d = true;
            if (!score_valid(cur_d)) bad_score = true;
        }
        V.insert(cur, lane == 0, lane);
        while (!bad_score) {
            uint32_t nid = kSentinel;
            if (cur_base != kSentinel && cur_level >= layer && (uint32_t)lane < ix.su)
                nid = ix.up[(size_t)(cur_base + layer - 1) * ix.su + (uint32_t)lane];
            uint32_t deg;
            const uint32_t nf = frontier_from(nid, deg);
            if (nf == 0) break;
            const uint32_t fnode = (uint32_t)lane < nf ? fr_id[lane] : cur;
            const uint32_t f_base = ix.up_base[fnode];
            const uint32_t f_level = ix.level[fnode];
            score_frontier(nf);
            // sequential `if distance < current_dist` over the row == first minimum, if it improves
            float d = (uint32_t)lane < nf ? fr_d[lane] : inf;
            bool ok = true;
            if ((uint32_t)lane < nf) ok = score_valid(d);
            if (__ballot(!ok)) { bad_score = true; break; }
            const uint32_t m = wave_umin(__float_as_uint(d)); // scores are >= 0: bit order == value order
            const float mf = __uint_as_float(m);
            if (!(mf < cur_d)) break;
            const uint32_t w = (uint32_t)__builtin_ctzll(__ballot((uint32_t)lane < nf && d == mf));
            cur = __builtin_amdgcn_readlane(fnode, w);
            cur_base = __builtin_amdgcn_readlane(f_base, w);
            cur_level = __builtin_amdgcn_readlane(f_level, w);
            cur_d = mf;
            __syncthreads();
        }
        if (bad_score) break;
    }

    // ---------------- layer 0: strict-exhaustive beam (search.rs:267-1067) ----------------
    Beam<R> S;
    uint32_t st_exp = 0, st_nb = 0, st_vl = 0, st_dc = 0;
    bool tie_overflow = false;
    uint32_t bl = beam_top; // the layer the beam runs on (always 0 outside BUILD)
    // neighbour row of node c on the beam's layer, one id per lane
    auto load_row = [&](uint32_t c) __attribute__((always_inline)) -> uint32_t {
        if (!BUILD || bl == 0u) return (uint32_t)lane < ix.s0 ? ix.l0[(size_t)c * ix.s0 + (uint32_t)lane] : kSentinel;
        const uint32_t base = ix.up_base[c]; // every node reached on layer bl >= 1 lives there: its row exists
        return (uint32_t)lane < ix.su ? ix.up[(size_t)(base + bl - 1u) * ix.su + (uint32_t)lane] : kSentinel;
    };
    uint32_t dropped_unexpanded = 0;
    uint32_t ef = a.ef;
    uint32_t pf_id = kSentinel, pf_row = kSentinel; // predicted next candidate and its prefetched row
    // non-strict arms with a SimHash filter: the SimHash rows of the prefetched neighbour row are requested underneath
    // the admission loop, so the filter of the next expansion does not wait for an 8-byte gather of its own
    uint32_t pf_hash_for = kSentinel;
    unsigned long long pf_hash = 0ull;
    const bool want_hash = AD && a.ad.filtering != 0u;
    auto prefetch_hash = [&]() __attribute__((always_inline)) {
        if (AD && want_hash && pf_id != kSentinel && pf_hash_for != pf_id) {
            pf_hash = pf_row != kSentinel ? a.ad.node_hash[pf_row] : 0ull;
            pf_hash_for = pf_id;
        }
    };
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0, t_begin = 0; // PROF only
    auto tick = [&](int phase, bool wait) __attribute__((always_inline)) {
        if (PROF) {
            if (wait) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            t1 = __builtin_readcyclecounter();
            pt[phase] += t1 - t0;
            t0 = t1;
        }
    };
    AdaptState A;
    QueryRngT<(OCC == 2 ? kRngWordsOcc2 : kRngWords)> G;
    uint64_t qh = 0;
    float brk_lane = -1.0f;
    if (AD) {
        A.fill = A.win_ex = A.win_filt = A.win_exp = A.bstate = A.bremain = A.reads = 0u;
        A.st = hvx_adaptive_stats{};
        qh = a.ad.qhash[q];
        brk_lane = a.ad.thr_break[lane];
        G.buf = rng_buf;
        // SearchRandomness::QueryDerived (randomness.rs:104-120): the LAYER-0 entry after the descent, external id
        const uint64_t ep = ix.ids[cur], efw = (uint64_t)a.ef;
        G.seed(qh ^ ((ep << 17) | (ep >> 47)) ^ ((efw << 7) | (efw >> 57)));
    }
  for (;;) { // one pass outside BUILD; BUILD: one beam per layer, bl = beam_top .. 0
    S.init();
    dropped_unexpanded = 0;
    ef = (BUILD && bl > 0u) ? a.build_ef_upper : a.ef;
    pf_id = kSentinel;
    pf_row = kSentinel;
    if (!bad_score) {
        V.clear(lane);
        float d0 = have_d ? cur_d : score_one(cur); // search.rs:500-512: the same distance once more
        st_dc = 1;
        if (!score_valid(d0)) bad_score = true;
        V.insert(cur, lane == 0, lane);
        float ds;
        S.insert(d0, cur, lane, ds);
    }
    t_begin = PROF ? __builtin_readcyclecounter() : 0ull;
    if (PROF) t0 = t_begin;
    while (!bad_score) {
        const uint32_t pos = S.first_unexpanded(lane);
        if (pos >= S.count) {
            // the reference would still pop an evicted candidate that we no longer hold, count the
            // step and stop on `current_dist > w.peek()` (search.rs:549)
            if (dropped_unexpanded) {
                ++st_exp;
                if (AD && ST) { A.st.effective_beam_len_sum += (S.count < ef ? S.count : ef) + A.fill; A.st.effective_beam_len_samples += 1u; }
            }
            break;
        }
        ++st_exp;
        const float dc = S.score_at(pos);
        const uint32_t c = S.id_at(pos);
        const uint32_t wlen = S.count < ef ? S.count : ef;
        float wmax = S.score_at(wlen - 1);
        if (AD && ST) { A.st.effective_beam_len_sum += wlen + A.fill; A.st.effective_beam_len_samples += 1u; }
        if (wlen + (AD ? A.fill : 0u) >= ef && dc > wmax) break; // effective_len counts the virtual fill slots
        S.mark_expanded(pos, lane);

        uint32_t nid;
        if (c == pf_id) { nid = pf_row; if (PROF && !AD) pt[6] += 1; }
        else nid = load_row(c);
        tick(0, true); // pop + neighbour row available
        unsigned long long nh = 0ull;
        if (AD && want_hash) nh = c == pf_hash_for ? pf_hash : (nid != kSentinel ? a.ad.node_hash[nid] : 0ull);
        uint32_t deg;
        unsigned long long um = 0ull; // AD: lanes (row positions) of the unseen neighbours
        uint32_t nf = AD ? frontier_claim(nid, deg, um) : frontier_from(nid, deg);
        st_nb += deg;
        tick(1, true); // visited test-and-set + compaction
        // 93 % of the time the next pop is simply the next unexpanded entry already in the beam: its
        // neighbour row goes out AHEAD of this expansion's gathers and is back together with them
        const uint32_t p2 = S.first_unexpanded(lane);
        uint32_t e2 = kSentinel, row2 = kSentinel;
        float s2 = inf;
        if (p2 < S.count) {
            e2 = S.id_at(p2);
            s2 = S.score_at(p2);
            row2 = load_row(e2);
        }
        pf_id = e2;
        pf_row = row2;
        if (nf == 0) { prefetch_hash(); continue; }
        if (AD) {
            // ---- one decision epoch + candidate selection (search.rs:595-829) ----
            const AdaptArgs &P = a.ad;
            const uint32_t nf0 = nf;
            const uint32_t node = nid;
            const unsigned long long node_h = nh;
            const bool unseen = (um >> lane) & 1ull;
            const uint32_t frank = (uint32_t)__builtin_popcountll(um & ((1ull << lane) - 1ull)); // index in the frontier
            const uint32_t kt = a.k > 1u ? a.k : 1u;              // topk_target
            const float delta = S.score_at((kt < wlen ? kt : wlen) - 1u); // topk == the first min(k,|W|) of W
            const AdaptDecision D = adapt_decide(P, A, ef, wlen >= kt, wlen, nf0, dc, delta, brk_lane);
            if (ST) A.st.simhash_bypass_trigger_low_yield += (D.trigger & 2u) ? 1u : 0u;
            if (ST) A.st.simhash_bypass_trigger_budget += (D.trigger & 1u) ? 1u : 0u;
            if (ST) A.st.active_sampling_ratio_sum += (double)D.base_p;
            if (ST) A.st.active_sampling_ratio_samples += 1u;
            if (ST && D.filter) { A.st.active_simhash_threshold_sum += D.threshold; A.st.active_simhash_threshold_samples += 1u; }
            // stage 0: pre-sampling, one draw per frontier neighbour in row order (search.rs:651-679)
            bool keep = unseen;
            const bool pre_enabled = D.pre_kind != 0u;
            if (pre_enabled) {
                if (D.pre_p <= 0.0f) keep = false;
                else if (D.pre_p < 1.0f) {
                    G.ensure(lane);
                    keep = keep && G.below(G.pos + frank, D.pre_p);
                    G.pos += nf0;
                }
                uint32_t nk = (uint32_t)__builtin_popcountll(__ballot(keep));
                if (ST) A.st.pre_simhash_sample_dropped += nf0 - nk;
                if (nk == 0u) { keep = unseen & (frank == G.choose_index(nf0, lane)); nk = 1u; }
                if (ST) A.st.pre_simhash_sample_kept += nk;
            }
            const uint32_t ns = (uint32_t)__builtin_popcountll(__ballot(keep));
            if (ST && D.bypassed) { A.st.simhash_bypass_expansions += 1u; A.st.simhash_skipped_candidates += ns; }
            // threshold screening (search.rs:708-757): filtered rows are marked visited and, while the beam is
            // still filling, consume a virtual fill slot
            uint32_t sim = 32u;
            bool pass = keep, vfail = false;
            if (D.filter) {
                if (P.count_reads) { // fill_simhash_cache (memory_store.rs:314-372): every row not yet in the query-local cache is one read
                    const bool fresh_row = keep & !vhashed;
                    A.reads += (uint32_t)__builtin_popcountll(__ballot(fresh_row));
                    V.mark_hashed(node, fresh_row, vslot);
                }
                const uint64_t h = keep ? node_h : 0ull;
                sim = 64u - (uint32_t)__builtin_popcountll(h ^ qh); // collision_count == 64 - hamming_distance
                const bool failed = keep & !(sim >= D.threshold);
                const uint32_t nfail = (uint32_t)__builtin_popcountll(__ballot(failed));
                if (ST) A.st.simhash_examined += ns;
                if (ST) A.st.simhash_filtered += nfail;
                // rows are canonical (deduped) and these neighbours were unseen a moment ago: every filtered row is a
                // fresh `visited.insert`; the insertion itself happens once, together with the admitted rows below
                vfail = failed;
                const uint32_t nfresh = nfail;
                const uint32_t eff = wlen + A.fill;
                const uint32_t room = eff < ef ? ef - eff : 0u;
                A.fill += nfresh < room ? nfresh : room;
                pass = keep & !failed;
                if (ns > 0u) { // rolling yield window (search.rs:790-801)
                    A.win_ex += ns;
                    A.win_filt += nfail;
                    A.win_exp += 1u;
                    if (A.win_exp > P.window_expansions) { A.win_ex >>= 1; A.win_filt >>= 1; A.win_exp = P.window_expansions >> 1; }
                }
            }
            if (ST) A.st.simhash_passed_before_sampling += (uint32_t)__builtin_popcountll(__ballot(pass));
            // proximity-aware probabilistic expansion (search.rs:759-788): draws only for 0 < p < 1, in row order
            bool sampled = pass, deferred = false;
            const bool should_sample = !pre_enabled & (D.samp_kind != 0u) & (D.samp_p > 0.0f);
            if (should_sample) {
                const float pr = candidate_probability_fn(D.samp_kind, D.samp_p, sim, D.filter, D.threshold);
                const bool draws = pass & (pr > 0.0f) & (pr < 1.0f);
                const unsigned long long dm = __ballot(draws);
                G.ensure(lane);
                bool hit = pr >= 1.0f;
                if (draws) hit = G.below(G.pos + (uint32_t)__builtin_popcountll(dm & ((1ull << lane) - 1ull)), pr);
                G.pos += (uint32_t)__builtin_popcountll(dm);
                sampled = pass & hit;
                deferred = pass & !hit;
            } else if (D.samp_p <= 0.0f) {
                sampled = false;
                deferred = pass;
            }
            // avoid getting stuck on sparse frontier expansions (search.rs:803-823)
            if (D.samp_p > 0.0f && !__ballot(sampled) && __ballot(deferred)) {
                const uint32_t best = 64u - wave_umin(deferred ? 64u - sim : 0xFFFFFFFFu);
                unsigned long long bm = __ballot(deferred & (sim == best));
                uint32_t idx = G.choose_index((uint32_t)__builtin_popcountll(bm), lane);
                while (idx--) bm &= bm - 1ull;
                sampled = (uint32_t)lane == (uint32_t)__builtin_ctzll(bm);
            }
            // mark_sampled_neighbors_visited (search.rs:89-97): only admitted rows become visited
            // ... and so do the rows the filter rejected (search.rs:739-748): ONE pass over the visited table for both
            V.commit(node, sampled | vfail, vslot);
            const bool acc = sampled;
            const unsigned long long am = __ballot(acc);
            if (ST) A.st.simhash_passed_after_sampling += (uint32_t)__builtin_popcountll(__ballot(sampled));
            if (acc) fr_id[__builtin_popcountll(am & ((1ull << lane) - 1ull))] = node;
            __syncthreads();
            nf = (uint32_t)__builtin_popcountll(am);
            tick(6, true); // (PROF) the decision epoch + candidate selection of the non-strict arms
            if (nf == 0) { prefetch_hash(); continue; }
        }
        st_vl += nf;
        st_dc += nf;
        score_frontier(nf);
        prefetch_hash(); // the prefetched row came back with the gathers: its SimHash rows go out under the admission loop
        tick(2, true); // row gathers + FMAs
        const float d_l = (uint32_t)lane < nf ? fr_d[lane] : inf;
        const uint32_t id_l = (uint32_t)lane < nf ? fr_id[lane] : kSentinel;
        __syncthreads();
        {
            // a fresh candidate of this expansion that beats that entry becomes the prediction instead;
            // its row is fetched underneath the admission loop
            const uint32_t wl0 = (S.count < ef ? S.count : ef) + (AD ? A.fill : 0u);
            const bool adm = ((uint32_t)lane < nf) & ((d_l < wmax) | (wl0 < ef)) & (d_l >= 0.f) & (d_l < inf);
            const uint32_t key = adm ? __float_as_uint(d_l) : 0xFFFFFFFFu;
            const uint32_t kmin = wave_umin(key);
            if (kmin != 0xFFFFFFFFu && __uint_as_float(kmin) < s2) {
                const uint32_t pred = __builtin_amdgcn_readlane(id_l, (uint32_t)__builtin_ctzll(__ballot(key == kmin)));
                pf_id = pred;
                pf_row = load_row(pred);
                if (PROF) pt[5] += 1; // the predicted next pop is a candidate discovered by THIS expansion
            }
        }
        tick(3, false); // prediction (the row load stays in flight)
        // admission in row order with the running bound (search.rs:928-952).  Once W is full the
        // bound only ever tightens, so a candidate that fails it at entry fails it at its turn: only
        // the lanes passing the entry bound are visited (in row order), each re-checked against the
        // running bound.  (While W is still filling its max can grow, so nothing is skipped then.)
        // Invalid scores fail the whole query wherever they sit.
        {
            float dv = d_l;
            const bool okv = (uint32_t)lane >= nf || score_valid(dv);
            if (__ballot(!okv)) { bad_score = true; break; }
            const uint32_t wl_in = (S.count < ef ? S.count : ef) + (AD ? A.fill : 0u);
            unsigned long long todo = __ballot(((uint32_t)lane < nf) & ((dv < wmax) | (wl_in < ef)));
            while (todo) {
                const uint32_t f = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1ull;
                const float d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dv), f));
                const uint32_t wl = (S.count < ef ? S.count : ef) + (AD ? A.fill : 0u); // effective_len (search.rs:927)
                if (d < wmax || wl < ef) {
                    // a real candidate replaces a virtual fill slot before anything is trimmed (search.rs:943-950)
                    if (AD) A.fill -= (wl >= ef && A.fill > 0u) ? 1u : 0u;
                    float ds = 0.f;
                    const bool drop = S.insert(d, __builtin_amdgcn_readlane(id_l, f), lane, ds);
                    const uint32_t wl2 = S.count < ef ? S.count : ef;
                    wmax = S.score_at(wl2 - 1);
                    if (drop) {
                        ++dropped_unexpanded;
                        if (!(ds > wmax)) tie_overflow = true; // an equal-score candidate left the beam
                    }
                }
            }
        }
        prefetch_hash(); // a fresh-candidate prediction replaced the prefetched row during this expansion
        if (PROF) { t1 = __builtin_readcyclecounter(); pt[4] += t1 - t0; t0 = t1; } // admission loop
    }
    if (!BUILD) break;
    // ---- BUILD: this layer's candidates (W sorted by (score, id), first k) -> [layer][b][k]; next layer starts at the best ----
    {
        const uint32_t wl = bad_score ? 0u : (S.count < ef ? S.count : ef);
        const uint32_t outn = wl < a.k ? wl : a.k;
        const size_t slot0 = ((size_t)bl * gridDim.x + q) * a.k;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t e = (uint32_t)r * 64u + (uint32_t)lane;
            if (e < outn) {
                a.out_ids[slot0 + e] = (uint64_t)(S.id[r] & ~kExpandedBit);
                a.out_scores[slot0 + e] = S.sc[r];
            }
        }
        if (lane == 0) a.out_counts[(size_t)bl * gridDim.x + q] = outn;
        if (bad_score || bl == 0u) break;
        cur = S.id_at(0) & ~kExpandedBit; // mutation.rs:876-878: the closest candidate enters the next layer
        cur_d = S.score_at(0);
        have_d = true;
        --bl;
    }
  }
    if (PROF && a.prof && lane == 0) {
        // cycles: [0] pop+row wait [1] visited [2] gather wait+FMA [3] predict [4] admit [7] whole layer-0 loop;
        // counts: [5] predictions that chose a fresh candidate [6] row-prefetch hits
        pt[7] = __builtin_readcyclecounter() - t_begin;
        for (int i = 0; i < 8; ++i) a.prof[(size_t)q * 8 + i] = pt[i];
    }

    // ---------------- results: w sorted by (score,id), take k (search.rs:995-1004,1229) ----------------
    uint32_t outn = 0;
    if (!BUILD && !bad_score) {
        const uint32_t ef = a.ef;
        const uint32_t wl = S.count < ef ? S.count : ef;
        outn = wl < a.k ? wl : a.k;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t e = (uint32_t)r * 64u + (uint32_t)lane;
            if (e < outn) {
                a.out_ids[(size_t)q * a.k + e] = ix.ids[S.id[r] & ~kExpandedBit];
                a.out_scores[(size_t)q * a.k + e] = S.sc[r];
            }
        }
    }
    if (V.spilled) { // leave the HBM bitmap(s) zeroed for the next launch
        for (uint32_t w = (uint32_t)lane; w < V.words; w += 64) { V.bm[w] = 0u; if (V.bm2) V.bm2[w] = 0u; }
    }
    if (lane == 0) {
        if (a.wave_clock && !a.only_flagged) { a.wave_clock[2 * (size_t)q] = wclk0; a.wave_clock[2 * (size_t)q + 1] = wall_clock64(); }
        if (!BUILD) a.out_counts[q] = outn;
        if (a.out_status) a.out_status[q] = bad_score ? 8u /*HVX_ERR_INVARIANT*/ : 0u;
        if (a.qstats) a.qstats[q] = hvx_query_stats{st_exp, st_nb, st_vl, st_dc};
        if (a.tie_flags) a.tie_flags[q] = tie_overflow ? 1u : 0u;
        // a query whose beam evicted equal-score candidates beyond its slack joins the re-run list (launch_hnsw_wave)
        if (!BUILD && tie_overflow && a.rerun_ctl && !a.only_flagged) a.rerun_list[atomicAdd(&a.rerun_ctl[0], 1u)] = q;
        if (AD && ST && a.ad.stats) {
            A.st.rng_words = G.pos;
            A.st.txn_get_simhash_filter = A.reads;
            a.ad.stats[q] = A.st;
        }
    }
}

// The kernel.  A search launch runs one query per workgroup (q = blockIdx.x).  In the RE-RUN launch (a.only_flagged) workgroup i
// searches the i-th query of the re-run list again, from scratch, with the next beam size; workgroups beyond the list leave at once
// -- all of them on every corpus without masses of duplicate vectors.  Round 4: the list replaces "every wavefront tests its own
// flag" and the re-run keeps the launch's register budget where a wider build exists for it: round 3's re-run was a one-query-
// per-SIMD launch whose 1 024 exit-at-once workgroups each waited for an EMPTY SIMD behind the two-per-SIMD batches of the other
// execution lanes (216 us per step on average; profiles/history/r03a_kernel_stats_hvx.csv).  The last workgroup hands the list back empty.
// (A loop "one workgroup takes every n-th listed query" was tried first: it costs every instantiation 30-70 spilled SGPRs.)
template <uint32_t METRIC, int R, int NK, bool BF, bool PROF = false, bool AD = false, bool ST = true, int OCC = 1, bool BUILD = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void hnsw_wave_kernel(HnswArgs a, uint32_t vcap) {
    uint32_t q = blockIdx.x;
    if (!BUILD && a.only_flagged) {
        const uint32_t listed = __hip_atomic_load(&a.rerun_ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool mine = q < listed;
        if (mine) q = a.rerun_list[q];
        if (threadIdx.x == 0u && atomicAdd(&a.rerun_ctl[1], 1u) == gridDim.x - 1u) { // every workgroup has read the count
            __hip_atomic_store(&a.rerun_ctl[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.rerun_ctl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (an entry left behind by a launch that failed before its re-run could empty the list must not send a later, smaller batch
        // past its output arrays: ADVICE r4)
        if (!mine || q >= gridDim.x) return;
    }
    hnsw_wave_query<METRIC, R, NK, BF, PROF, AD, ST, OCC, BUILD>(a, vcap, q);
}

// launch geometry shared by the per-metric translation units
struct WaveGeom {
    uint32_t log2cap;
    uint32_t cap; // slots of the LDS visited table (1 << log2cap unless the launch sized it to fill its LDS budget)
    size_t lds;
    uint32_t occ; // wavefronts per SIMD the launch is budgeted for (1 or 2)
};

// per-metric launchers, defined in hvx_hnsw_wave_l2.hip / hvx_hnsw_wave_cos.hip / hvx_hnsw_wave_prof.hip
hipError_t launch_hnsw_wave_l2(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_cos(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_l2_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_cos_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_prof(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
// two queries per SIMD (OCC = 2 builds), hvx_hnsw_wave_occ2.hip
hipError_t launch_hnsw_wave_occ2(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_occ2_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
// search side of the device HNSW build (BUILD instantiations), hvx_hnsw_wave_build.hip
hipError_t launch_hnsw_wave_build(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_build_gen(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s); // GENERIC build, hvx_hnsw_wave_build_gen.hip
hipError_t launch_hnsw_wave_build_occ2(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s); // two per SIMD, hvx_hnsw_wave_build_occ2.hip
// GENERIC builds (NK = 0) of the non-strict arms: any dim / metric / summation tree, ef <= 800; hvx_hnsw_wave_gen_*.hip
hipError_t launch_hnsw_wave_gen_cos(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_gen_l2(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_gen_l1(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
// non-strict arms (AD instantiations), hvx_hnsw_wave_l2_ad.hip / hvx_hnsw_wave_cos_ad.hip
hipError_t launch_hnsw_wave_l2_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_cos_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_l2_bf16_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_cos_bf16_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
// ... budgeted for two queries per SIMD (f32 rows), hvx_hnsw_wave_occ2_l2_ad.hip / hvx_hnsw_wave_occ2_cos_ad.hip
hipError_t launch_hnsw_wave_occ2_l2_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_occ2_cos_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_occ2_l2_bf16_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_occ2_cos_bf16_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);

template <typename K> static hipError_t launch_wave_kernel(K kern, const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    if (g.lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(b), dim3(64), g.lds, s, a, g.cap);
    return hipGetLastError();
}

template <uint32_t METRIC, int R, bool BF, bool AD = false, bool ST = true, int OCC = 1, bool BUILD = false>
static hipError_t launch_wave_nk(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    switch (a.ix.dim >> 5) {
    case 4: return launch_wave_kernel(hnsw_wave_kernel<METRIC, R, 4, BF, false, AD, ST, OCC, BUILD>, a, b, g, s);
    case 8: return launch_wave_kernel(hnsw_wave_kernel<METRIC, R, 8, BF, false, AD, ST, OCC, BUILD>, a, b, g, s);
    case 12: return launch_wave_kernel(hnsw_wave_kernel<METRIC, R, 12, BF, false, AD, ST, OCC, BUILD>, a, b, g, s); // dim 384 (round 6)
    case 16: return launch_wave_kernel(hnsw_wave_kernel<METRIC, R, 16, BF, false, AD, ST, OCC, BUILD>, a, b, g, s);
    case 24: return launch_wave_kernel(hnsw_wave_kernel<METRIC, R, 24, BF, false, AD, ST, OCC, BUILD>, a, b, g, s);
    case 32: return launch_wave_kernel(hnsw_wave_kernel<METRIC, R, 32, BF, false, AD, ST, OCC, BUILD>, a, b, g, s);
    case 48: return launch_wave_kernel(hnsw_wave_kernel<METRIC, R, 48, BF, false, AD, ST, OCC, BUILD>, a, b, g, s);
    default: return hipErrorInvalidValue;
    }
}

template <uint32_t METRIC, bool BF, bool AD = false, bool ST = true, int OCC = 1>
static hipError_t launch_wave_r(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    const uint32_t need = a.ef + 32u; // beam capacity 64*R must hold ef plus slack for equal-score evictions
    if (need <= 192 && !a.only_flagged) return launch_wave_nk<METRIC, 3, BF, AD, ST, OCC>(a, b, g, s);
    if (need <= 384) return launch_wave_nk<METRIC, 6, BF, AD, ST, OCC>(a, b, g, s); // also the re-run of an R = 3 launch: slack 32 -> 224+
    return hipErrorInvalidValue;
}
// wide register beams of the strict arm (round 4; hvx_hnsw_wave_wide_*.hip): 448 entries (ef <= 416) and 832 entries (ef <= 800), one
// query per SIMD; the 832-entry build is also the re-run of a 384- or 448-entry launch
template <uint32_t METRIC, bool BF, int OCC> static hipError_t launch_wave_wide_o(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    const uint32_t need = a.ef + 32u;
    if (need <= 448 && !a.only_flagged) return launch_wave_nk<METRIC, 7, BF, false, true, OCC>(a, b, g, s);
    if (need <= 832) return launch_wave_nk<METRIC, 13, BF, false, true, OCC>(a, b, g, s);
    return hipErrorInvalidValue;
}
// one query per SIMD, or (handles with two queries per SIMD: several batches in flight) the half-register build, whose second
// resident query hides the first one's round trips exactly as it does for the narrow beams
template <uint32_t METRIC, bool BF> static hipError_t launch_wave_wide_r(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    return g.occ == 2 ? launch_wave_wide_o<METRIC, BF, 2>(a, b, g, s) : launch_wave_wide_o<METRIC, BF, 1>(a, b, g, s);
}
// GENERIC build of the non-strict arms: beam of 64*R >= ef + 32 entries
template <uint32_t METRIC, bool ST> static hipError_t launch_wave_gen_r(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    const uint32_t need = a.ef + 32u;
    if (need <= 192 && !a.only_flagged) return launch_wave_kernel(hnsw_wave_kernel<METRIC, 3, 0, false, false, true, ST>, a, b, g, s);
    if (need <= 448 && !(a.only_flagged && need > 192)) return launch_wave_kernel(hnsw_wave_kernel<METRIC, 7, 0, false, false, true, ST>, a, b, g, s);
    if (need <= 832) return launch_wave_kernel(hnsw_wave_kernel<METRIC, 13, 0, false, false, true, ST>, a, b, g, s);
    return hipErrorInvalidValue;
}
template <uint32_t METRIC> static hipError_t launch_wave_gen(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    return a.ad.stats ? launch_wave_gen_r<METRIC, true>(a, b, g, s) : launch_wave_gen_r<METRIC, false>(a, b, g, s);
}
// non-strict arms: with the per-query SearchStats of the filter / sampling stages when the caller asked for them,
// else the diagnostics-free build
template <uint32_t METRIC, bool BF>
static hipError_t launch_wave_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    return a.ad.stats ? launch_wave_r<METRIC, BF, true, true>(a, b, g, s) : launch_wave_r<METRIC, BF, true, false>(a, b, g, s);
}

} // namespace hvx
