// hvx_hnsw.hip -- batched HNSW search on gfx950: one 256-thread workgroup (4 wavefronts) per query.
//
// Restates SearchSession::run (crates/db/src/search/vector/search.rs:1101-1230): greedy descent of
// the upper layers (search_layer_greedy, :169-224) then the strict-exhaustive layer-0 beam
// (search_layer0_with_simhash<_, STRICT_EXHAUSTIVE=true>, :267-1067), step for
// step, so results, scores AND the SearchStats counters equal the reference's CPU path.
//
// Mapping onto CDNA4:
//   * query vector staged once in LDS; each expansion gathers the unvisited neighbour rows with
//     16-byte coalesced loads, 8 lanes per row, 8 rows per wavefront, 32 rows per workgroup pass;
//   * distances are produced in the host kernel's summation order (hvx_device.h) => bit-exact;
//   * the reference's two BinaryHeaps (`candidates`, `w`) are ONE sorted beam held in registers,
//     striped across the 64 lanes (entry e in register e/64, lane e%64), replicated in all four
//     wavefronts so no broadcast is needed: W = first min(count,ef) entries, candidates = entries
//     without the expanded bit.  Insert = ballot-popcount position + one DPP wave-shift;
//   * visited set = one bit per node per query in HBM, test-and-set with atomicOr (exact for any
//     shard size, sized for 288 GB: 125 KB/query at 1M rows).
#include <cstdlib>

#include "hvx_beam.h"
#include "hvx_device.h"
#include "hvx_kernels.h"

#ifdef HVX_TUNING
static inline const char *tuning_env(const char *name) { return getenv(name); }
#else
static inline const char *tuning_env(const char *) { return nullptr; }
#endif

namespace hvx {

constexpr int kMaxStride = 128;  // largest neighbour-row stride served
constexpr int kClearCap = 1024;  // upper-layer visited ids remembered for targeted clearing

// LDS carve (dynamic, 16-byte aligned): query | frontier ids x2 | frontier dist x2 | clear list | misc
struct LdsView {
    float *qv;
    uint32_t *fr_id_base; // [2][kMaxStride]
    float *fr_d_base;     // [2][kMaxStride]
    uint32_t *clr;
    uint32_t *misc; // [0..1] frontier sizes, [2] clr_n, [4] row degree
    __device__ __forceinline__ uint32_t *fr_id(int p) const { return fr_id_base + p * kMaxStride; }
    __device__ __forceinline__ float *fr_d(int p) const { return fr_d_base + p * kMaxStride; }
};

__device__ __forceinline__ LdsView carve(char *smem, uint32_t ld) {
    LdsView v;
    uint32_t qbytes = ((ld * 4u) + 15u) & ~15u;
    v.qv = reinterpret_cast<float *>(smem);
    char *p = smem + qbytes;
    v.fr_id_base = reinterpret_cast<uint32_t *>(p); p += 2 * kMaxStride * 4;
    v.fr_d_base = reinterpret_cast<float *>(p); p += 2 * kMaxStride * 4;
    v.clr = reinterpret_cast<uint32_t *>(p); p += kClearCap * 4;
    v.misc = reinterpret_cast<uint32_t *>(p);
    return v;
}

size_t hnsw_lds_bytes(uint32_t ld) {
    return (((size_t)ld * 4u + 15u) & ~(size_t)15u) + 4u * kMaxStride * 4u + kClearCap * 4u + 64u;
}

// wavefront 0: test-and-set the visited bits of one neighbour row and compact the unvisited ids,
// in row order (= ascending id), into fr[]; returns frontier size, *deg = valid ids in the row.
__device__ __forceinline__ uint32_t frontier_from_row(const uint32_t *row, uint32_t stride, uint32_t *bm,
                                                      uint32_t *fr, int lane, uint32_t *deg,
                                                      uint32_t *clr, uint32_t *clr_n, bool record) {
    uint32_t base = 0, d = 0;
    for (uint32_t c = 0; c < stride; c += 64) {
        uint32_t slot = c + (uint32_t)lane;
        uint32_t nid = slot < stride ? row[slot] : kSentinel;
        bool valid = nid != kSentinel;
        bool unv = false;
        if (valid) {
            uint32_t bit = 1u << (nid & 31u);
            uint32_t old = atomicOr(&bm[nid >> 5], bit);
            unv = !(old & bit);
        }
        unsigned long long vm = __ballot(valid);
        unsigned long long um = __ballot(unv);
        d += (uint32_t)__builtin_popcountll(vm);
        if (unv) {
            uint32_t pos = base + (uint32_t)__builtin_popcountll(um & ((1ull << lane) - 1ull));
            fr[pos] = nid;
            if (record) {
                uint32_t cp = *clr_n + pos;
                if (cp < (uint32_t)kClearCap) clr[cp] = nid;
            }
        }
        base += (uint32_t)__builtin_popcountll(um);
        if (!vm) break; // rows are sentinel-padded at the end
    }
    if (record) *clr_n += base;
    *deg = d;
    return base;
}

template <uint32_t METRIC, bool FUSED, int R>
__global__ __launch_bounds__(256) void hnsw_search_kernel(HnswArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevIndex &ix = a.ix;
    const uint32_t q = blockIdx.x;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 3, j = lane & 7;
    LdsView L = carve(smem, ix.ld);

    const uint32_t status_in = a.qstatus ? a.qstatus[q] : 0u;
    if (status_in != 0u || !ix.has_entry) {
        if (tid == 0) {
            a.out_counts[q] = 0;
            if (a.out_status) a.out_status[q] = status_in;
            if (a.qstats) a.qstats[q] = hvx_query_stats{0, 0, 0, 0};
        }
        return;
    }
    for (uint32_t i = (uint32_t)tid; i < ix.ld; i += 256) L.qv[i] = i < ix.dim ? a.queries[(size_t)q * ix.dim + i] : 0.f;
    if (tid < 8) L.misc[tid] = 0;
    const float qhdr = a.qhdr ? a.qhdr[q] : 0.f;
    uint32_t *bm = a.bitmap + (size_t)q * a.words_per_query;
    __syncthreads();

    bool bad_score = false;
    uint32_t cur = ix.entry;

    // ---------------- upper layers: search_layer_greedy (search.rs:169-224) ----------------
    for (uint32_t layer = ix.max_layer; layer >= 1; --layer) {
        // fresh visited set per layer: bits set here are cleared again before the next layer
        float cur_d = group_distance<METRIC, FUSED>(ix, L.qv, qhdr, cur, j);
        if (!score_valid(cur_d)) bad_score = true;
        if (tid == 0) {
            atomicOr(&bm[cur >> 5], 1u << (cur & 31u));
            L.clr[0] = cur;
            L.misc[2] = 1;
        }
        int p = 0;
        for (;;) {
            __syncthreads();
            if (wave == 0) {
                uint32_t nf = 0, deg = 0;
                uint32_t base_row = ix.up_base[cur];
                uint32_t clr_n = L.misc[2];
                if (base_row != kSentinel && ix.level[cur] >= layer) {
                    const uint32_t *row = ix.up + (size_t)(base_row + layer - 1) * ix.su;
                    nf = frontier_from_row(row, ix.su, bm, L.fr_id(p), lane, &deg, L.clr, &clr_n, true);
                }
                if (lane == 0) { L.misc[p] = nf; L.misc[2] = clr_n; }
            }
            __syncthreads();
            const uint32_t nf = L.misc[p];
            if (nf == 0) break;
            for (uint32_t f = (uint32_t)(grp * 4 + wave); f < nf; f += 32) {
                float d = group_distance<METRIC, FUSED>(ix, L.qv, qhdr, L.fr_id(p)[f], j);
                if (j == 0) L.fr_d(p)[f] = d;
            }
            __syncthreads();
            // sequential `if distance < current_dist` over the row == first minimum, if it improves
            bool changed = false;
            for (uint32_t c = 0; c < nf; c += 64) {
                uint32_t f = c + (uint32_t)lane;
                float d = f < nf ? L.fr_d(p)[f] : __uint_as_float(0x7F800000u);
                bool ok = true;
                if (f < nf) ok = score_valid(d);
                if (__ballot(!ok)) bad_score = true;
                float m = d;
#pragma unroll
                for (int s = 1; s < 64; s <<= 1) m = fminf(m, __shfl_xor(m, s, 64));
                if (m < cur_d) {
                    unsigned long long eq = __ballot(f < nf && d == m);
                    uint32_t first = c + (uint32_t)__builtin_ctzll(eq);
                    cur = L.fr_id(p)[first];
                    cur_d = m;
                    changed = true;
                }
            }
            if (bad_score || !changed) break;
            p ^= 1;
        }
        __syncthreads();
        // clear this layer's visited bits
        const uint32_t clr_n = L.misc[2];
        if (clr_n > (uint32_t)kClearCap) {
            for (uint32_t w = (uint32_t)tid; w < a.words_per_query; w += 256) bm[w] = 0u;
        } else if (wave == 0) {
            for (uint32_t i = (uint32_t)lane; i < clr_n; i += 64) {
                uint32_t nid = L.clr[i];
                atomicAnd(&bm[nid >> 5], ~(1u << (nid & 31u)));
            }
        }
        __syncthreads();
        if (bad_score) break;
    }

    // ---------------- layer 0: strict-exhaustive beam (search.rs:267-1067) ----------------
    Beam<R> S;
    S.init();
    uint32_t st_exp = 0, st_nb = 0, st_vl = 0, st_dc = 0;
    bool tie_overflow = false;
    uint32_t dropped_unexpanded = 0;
    const uint32_t ef = a.ef;
    if (!bad_score) {
        float d0 = group_distance<METRIC, FUSED>(ix, L.qv, qhdr, cur, j);
        st_dc = 1;
        if (!score_valid(d0)) bad_score = true;
        if (tid == 0) atomicOr(&bm[cur >> 5], 1u << (cur & 31u));
        float ds;
        S.insert(d0, cur, lane, ds);
    }
    int p = 0;
    while (!bad_score) {
        uint32_t pos = S.first_unexpanded(lane);
        if (pos >= S.count) {
            // the reference would still pop an evicted candidate that we no longer hold, count the
            // step and stop on `current_dist > w.peek()` (search.rs:549)
            if (dropped_unexpanded) ++st_exp;
            break;
        }
        ++st_exp;
        const float dc = S.score_at(pos);
        const uint32_t c = S.id_at(pos);
        const uint32_t wlen = S.count < ef ? S.count : ef;
        float wmax = S.score_at(wlen - 1);
        if (wlen >= ef && dc > wmax) break;
        S.mark_expanded(pos, lane);

        if (wave == 0) {
            uint32_t deg = 0, dummy = 0;
            const uint32_t *row = ix.l0 + (size_t)c * ix.s0;
            uint32_t nf = frontier_from_row(row, ix.s0, bm, L.fr_id(p), lane, &deg, L.clr, &dummy, false);
            if (lane == 0) { L.misc[p] = nf; L.misc[4] = deg; }
        }
        __syncthreads();
        const uint32_t nf = L.misc[p];
        st_nb += L.misc[4];
        if (nf == 0) { __syncthreads(); continue; }
        st_vl += nf;
        st_dc += nf;
        for (uint32_t f = (uint32_t)(grp * 4 + wave); f < nf; f += 32) {
            float d = group_distance<METRIC, FUSED>(ix, L.qv, qhdr, L.fr_id(p)[f], j);
            if (j == 0) L.fr_d(p)[f] = d;
        }
        __syncthreads();
        // admission in row order with the running bound (search.rs:928-952)
        for (uint32_t f = 0; f < nf; ++f) {
            float d = L.fr_d(p)[f];
            if (!score_valid(d)) { bad_score = true; break; }
            const uint32_t wl = S.count < ef ? S.count : ef;
            if (d < wmax || wl < ef) {
                float ds = 0.f;
                bool drop = S.insert(d, L.fr_id(p)[f], lane, ds);
                const uint32_t wl2 = S.count < ef ? S.count : ef;
                wmax = S.score_at(wl2 - 1);
                if (drop) {
                    ++dropped_unexpanded;
                    if (!(ds > wmax)) tie_overflow = true; // an equal-score candidate left the beam
                }
            }
        }
        p ^= 1;
    }

    // ---------------- results: w sorted by (score,id), take k (search.rs:995-1004,1229) ----------------
    uint32_t outn = 0;
    if (!bad_score) {
        const uint32_t wl = S.count < ef ? S.count : ef;
        outn = wl < a.k ? wl : a.k;
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                uint32_t e = (uint32_t)r * 64u + (uint32_t)lane;
                if (e < outn) {
                    a.out_ids[(size_t)q * a.k + e] = ix.ids[S.id[r] & ~kExpandedBit];
                    a.out_scores[(size_t)q * a.k + e] = S.sc[r];
                }
            }
        }
    }
    if (tid == 0) {
        a.out_counts[q] = outn;
        if (a.out_status) a.out_status[q] = bad_score ? 8u /*HVX_ERR_INVARIANT*/ : 0u;
        if (a.qstats) a.qstats[q] = hvx_query_stats{st_exp, st_nb, st_vl, st_dc};
        if (a.tie_flags) a.tie_flags[q] = tie_overflow ? 1u : 0u;
    }
}

template <uint32_t METRIC, bool FUSED>
static hipError_t launch_r(const HnswArgs &a, uint32_t b, hipStream_t s) {
    const size_t lds = hnsw_lds_bytes(a.ix.ld);
    // beam capacity 64*R must hold ef plus slack for equal-score evictions
    const uint32_t need = a.ef + 32u;
    if (need <= 128) hipLaunchKernelGGL((hnsw_search_kernel<METRIC, FUSED, 2>), dim3(b), dim3(256), lds, s, a);
    else if (need <= 192) hipLaunchKernelGGL((hnsw_search_kernel<METRIC, FUSED, 3>), dim3(b), dim3(256), lds, s, a);
    else if (need <= 256) hipLaunchKernelGGL((hnsw_search_kernel<METRIC, FUSED, 4>), dim3(b), dim3(256), lds, s, a);
    else if (need <= 512) hipLaunchKernelGGL((hnsw_search_kernel<METRIC, FUSED, 8>), dim3(b), dim3(256), lds, s, a);
    else if (need <= 1024) hipLaunchKernelGGL((hnsw_search_kernel<METRIC, FUSED, 16>), dim3(b), dim3(256), lds, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_hnsw_search(const HnswArgs &a, uint32_t b, hipStream_t s) {
    const bool fused = kernel_fused(a.ix.fkernel);
    switch (a.ix.metric) {
    case kCosine: return fused ? launch_r<kCosine, true>(a, b, s) : launch_r<kCosine, false>(a, b, s);
    case kL2: return fused ? launch_r<kL2, true>(a, b, s) : launch_r<kL2, false>(a, b, s);
    default: return launch_r<kL1, true>(a, b, s);
    }
}

// ---------------------------------------------------------------------------------------------
// dispatch to the one-wavefront-per-query kernel (hvx_hnsw_wave.h)
// ---------------------------------------------------------------------------------------------
struct WaveGeom {
    uint32_t log2cap;
    uint32_t cap; // slots of the LDS visited table
    size_t lds;
    uint32_t occ;
};
hipError_t launch_hnsw_wave_occ2(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_occ2_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_build(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_build_occ2(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_build_gen(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_build_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s); // one-node inserts into bf16 images
hipError_t launch_hnsw_wave_gen_cos(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_gen_l2(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_gen_l1(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_l2(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_cos(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_prof(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_l2_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_cos_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_l2_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_cos_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_l2_bf16_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_cos_bf16_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_occ2_l2_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);   // non-strict arms, two queries per SIMD
hipError_t launch_hnsw_wave_occ2_cos_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_occ2_l2_bf16_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_occ2_cos_bf16_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_pair_l2(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);       // owner / gatherer kernel (hvx_hnsw_pair.h)
hipError_t launch_hnsw_pair_cos(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_pair_l2_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_pair_cos_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_wide_l2(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);  // strict arm, beams of 448 / 832
hipError_t launch_hnsw_wave_wide_cos(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_wide_l2_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
hipError_t launch_hnsw_wave_wide_cos_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s);
constexpr uint32_t kRngWords = 1024, kRngWordsOcc2 = 256; // LDS window of the query RNG (hvx_hnsw_wave.h)

// wide beams (round 4): the strict arm on the unrolled shapes runs register beams of 448 / 832 entries too (hvx_hnsw_wave_wide*.hip:
// ef up to 800 = the reference's restricted-path limit; search.rs:267-1067 itself has no limit, ef beyond that takes the general kernel)
static uint32_t wave_beam_limit(const HnswArgs &a) { return (a.adaptive || a.build_nodes || a.prof) ? 384u : 832u; }
bool hnsw_wave_supported(const HnswArgs &a) {
    const DevIndex &ix = a.ix;
    if (ix.metric != kL2 && ix.metric != kCosine) return false;
    if (ix.fkernel != kKernelAvxFma) return false;
    if (ix.dim % 32u != 0u || ix.dim_main != ix.dim || ix.ld != ix.dim) return false;
    if (ix.dtype != HVX_F32 && ix.dtype != HVX_BF16) return false;
    const uint32_t nk = ix.dim >> 5;
    if (nk != 4 && nk != 8 && nk != 12 && nk != 16 && nk != 24 && nk != 32 && nk != 48) return false; // (12 = dim 384: round 6)
    if (ix.s0 > 64 || ix.su > 64) return false;
    if (a.ef + 32u > wave_beam_limit(a)) return false;
    return true;
}

// the GENERIC builds of the non-strict arms: any dimension, metric and summation tree over f32 rows; one id per lane still
// bounds the neighbour rows at 64 ids, the register beam bounds ef at 800 (the restricted path's k limit)
static bool hnsw_wave_generic_supported(const HnswArgs &a) {
    const DevIndex &ix = a.ix;
    if (ix.dtype != HVX_F32) return false;
    if (ix.s0 > 64 || ix.su > 64) return false;
    if (a.ef + 32u > 832u) return false;
    return true;
}
bool hnsw_wave_adaptive_supported(const HnswArgs &a) { return hnsw_wave_supported(a) || hnsw_wave_generic_supported(a); }

static hipError_t launch_hnsw_wave_once(const HnswArgs &a, uint32_t b, hipStream_t s) {
    WaveGeom g;
    // visited hash: 64 slots per beam entry (load factor ~0.15-0.3 at the measured ~10 distance evaluations
    // per expansion); the kernel spills to the exact HBM bitmap beyond 3/4 full
    g.log2cap = 11;
    while ((1u << g.log2cap) < 64u * a.ef && g.log2cap < 15) ++g.log2cap;
    const bool generic = a.adaptive && !hnsw_wave_supported(a);
    if (generic && g.log2cap > 14) g.log2cap = 14; // 64 KiB table (two workgroups per CU); larger visited sets spill to the bitmap
    if (a.build_nodes && g.log2cap > 13) g.log2cap = 13; // build searches (ef_construction ~200): keep four workgroups per CU
    // the 448 / 832-entry register beams (hvx_hnsw_wave_wide*.hip): strict searches with ef 353..800, and the re-run of a 384-entry beam
    const bool wide = !a.adaptive && !a.build_nodes && !a.prof && (a.ef + 32u > 384u || (a.only_flagged && a.ef + 32u > 192u));
    // Unrolled builds: 8 192 slots (32 KiB) whatever the beam width -- FOUR wavefronts per CU, one per SIMD.  (Rounds 1-3 sized the
    // table at 64 slots per beam entry: 64 KiB from ef = 129, 128 KiB from ef = 257, i.e. two / one wavefronts per CU, which is
    // where the ef sweep lost its throughput; a search visits ~10 rows per expansion, ~14 slots per beam entry at 3/4 load.)  A
    // query that visits more than 6 144 rows continues on the exact HBM bitmap.
    if (!generic && g.log2cap > 13) g.log2cap = 13;
    if (a.log2cap >= 7 && a.log2cap <= 15) g.log2cap = a.log2cap; // HVX_OPT_WAVE_LOG2CAP: a tiny table exercises the spill path
    // 160 KiB / 4: exactly four resident wavefronts per CU, one per SIMD, each with the SIMD's whole register file.
    // occ = 2 (a.occupancy): eight per CU, two per SIMD -- the table shrinks until query + frontier + table fit 20 KiB
    bool build_generic = false; // build searches outside the unrolled shapes: the GENERIC build (one query per SIMD)
    if (a.build_nodes) {
        HnswArgs probe = a; // the beam of a build search holds max(ef_construction, 2 M) entries on every layer
        probe.ef = a.ef > a.build_ef_upper ? a.ef : a.build_ef_upper;
        build_generic = !hnsw_wave_supported(probe);
    }
    // (the wide beams have two-per-SIMD builds too; the one for bf16 rows at dim 1536 spills ~200 registers: it stays one per SIMD)
    // Round 5: the non-strict arms (the production default, SearchParams::new(k): access/search/storage.rs:140-141) have two-per-SIMD builds
    // as well -- f32 and bf16 rows, the unrolled shapes; their RNG window shrinks to 256 words so that the visited table keeps its size.
    const bool ad_occ2 = a.adaptive && !generic; // f32 and bf16 rows, the unrolled shapes
    g.occ = (a.occupancy == 2 && (!a.adaptive || ad_occ2) && !a.prof && !build_generic && !(wide && a.ix.dtype == HVX_BF16 && (a.ix.dim >> 5) == 48u)) ? 2u : 1u;
    size_t fixed = 512 + (size_t)a.ix.ld * 4 + (a.adaptive ? (g.occ == 2 ? kRngWordsOcc2 : kRngWords) * 4 : 0);
    g.cap = 1u << g.log2cap;
    if (g.occ == 2) {
        // the table takes what the 20 KiB of a half-SIMD wavefront leave (any multiple of 64 slots: the hash maps onto [0, cap) by a
        // multiply-high, hvx_hnsw_wave.h) -- 4 224 slots at dim 768, 3 456 at dim 1536 where a power of two allowed 4 096 / 2 048
        const size_t room = 20 * 1024 > fixed ? (20 * 1024 - fixed) / 4 / 64 * 64 : 0;
        const bool forced = a.log2cap >= 7 && a.log2cap <= 15;
        if (!(forced && g.cap <= room)) { // (HVX_OPT_WAVE_LOG2CAP: a tiny table exercises the spill path)
            g.cap = (uint32_t)std::min<size_t>(room, 8192);
            if (g.cap < 512) { // no room for a useful table next to the query: one query per SIMD
                g.occ = 1;
                g.cap = 1u << g.log2cap;
                fixed = 512 + (size_t)a.ix.ld * 4 + (a.adaptive ? kRngWords * 4 : 0);
            }
        }
    }
    const size_t budget = g.occ == 2 ? 20 * 1024 : 40 * 1024;
    const size_t need = (size_t)4 * g.cap + fixed;
    g.lds = need < budget ? budget : need;
    if (a.build_nodes) {
        if (a.ix.dtype == HVX_BF16) return (build_generic || g.occ != 1u) ? hipErrorInvalidValue : launch_hnsw_wave_build_bf16(a, b, g, s);
        if (build_generic) return launch_hnsw_wave_build_gen(a, b, g, s);
        return g.occ == 2 ? launch_hnsw_wave_build_occ2(a, b, g, s) : launch_hnsw_wave_build(a, b, g, s);
    }
    // one batch in flight: two wavefronts per query (owner + gatherer); the strict arm, beams of 192 / 384 entries
    if (a.pair && !a.adaptive && !a.prof && !a.only_flagged && !wide && g.occ == 1 && a.ef + 32u <= 384u) {
        WaveGeom pg = g;
        const size_t pneed = (size_t)4 * g.cap + 528 + (size_t)a.ix.ld * 4;
        pg.lds = pneed < 40 * 1024 ? 40 * 1024 : pneed;
        if (a.ix.dtype == HVX_BF16) return a.ix.metric == kL2 ? launch_hnsw_pair_l2_bf16(a, b, pg, s) : launch_hnsw_pair_cos_bf16(a, b, pg, s);
        return a.ix.metric == kL2 ? launch_hnsw_pair_l2(a, b, pg, s) : launch_hnsw_pair_cos(a, b, pg, s);
    }
    if (wide) {
        if (a.ix.dtype == HVX_BF16) return a.ix.metric == kL2 ? launch_hnsw_wave_wide_l2_bf16(a, b, g, s) : launch_hnsw_wave_wide_cos_bf16(a, b, g, s);
        return a.ix.metric == kL2 ? launch_hnsw_wave_wide_l2(a, b, g, s) : launch_hnsw_wave_wide_cos(a, b, g, s);
    }
    if (g.occ == 2 && a.adaptive) {
        if (a.ix.dtype == HVX_BF16) return a.ix.metric == kL2 ? launch_hnsw_wave_occ2_l2_bf16_ad(a, b, g, s) : launch_hnsw_wave_occ2_cos_bf16_ad(a, b, g, s);
        return a.ix.metric == kL2 ? launch_hnsw_wave_occ2_l2_ad(a, b, g, s) : launch_hnsw_wave_occ2_cos_ad(a, b, g, s);
    }
    if (g.occ == 2) return a.ix.dtype == HVX_BF16 ? launch_hnsw_wave_occ2_bf16(a, b, g, s) : launch_hnsw_wave_occ2(a, b, g, s);
    if (generic) {
        switch (a.ix.metric) {
        case kCosine: return launch_hnsw_wave_gen_cos(a, b, g, s);
        case kL2: return launch_hnsw_wave_gen_l2(a, b, g, s);
        default: return launch_hnsw_wave_gen_l1(a, b, g, s);
        }
    }
    if (a.adaptive && a.prof) return launch_hnsw_wave_prof(a, b, g, s); // tuning builds only
    if (a.adaptive) {
        if (a.ix.dtype == HVX_BF16) return a.ix.metric == kL2 ? launch_hnsw_wave_l2_bf16_ad(a, b, g, s) : launch_hnsw_wave_cos_bf16_ad(a, b, g, s);
        return a.ix.metric == kL2 ? launch_hnsw_wave_l2_ad(a, b, g, s) : launch_hnsw_wave_cos_ad(a, b, g, s);
    }
    if (a.prof) return launch_hnsw_wave_prof(a, b, g, s);
    if (a.ix.dtype == HVX_BF16) return a.ix.metric == kL2 ? launch_hnsw_wave_l2_bf16(a, b, g, s) : launch_hnsw_wave_cos_bf16(a, b, g, s);
    return a.ix.metric == kL2 ? launch_hnsw_wave_l2(a, b, g, s) : launch_hnsw_wave_cos(a, b, g, s);
}

// The search launch + its re-run: a beam of 64*R entries holds ef + >= 32 of slack; a query that evicted an EQUAL-score
// candidate past that slack (many duplicate vectors) may differ from the reference, so the kernel flags it, appends it to the
// handle's re-run list, and a second launch -- whose workgroups leave at once beyond the end of the list -- searches the
// listed queries again with the next beam size (unrolled strict builds 192 -> 384 -> 832, non-strict 192 -> 384, generic 192 ->
// 448 -> 832).  Queries that overflow even that stay flagged (hvx_stats.tie_overflow_queries).
hipError_t launch_hnsw_wave(const HnswArgs &a0, uint32_t b, hipStream_t s) {
    HnswArgs a = a0;
    const bool generic = a.adaptive && !hnsw_wave_supported(a);
    const uint32_t need = a.ef + 32u;
    // a wider instantiation exists
    const bool wider = generic ? need <= 448u : (a.adaptive ? need <= 192u : need <= 448u);
    const bool rerun = !a.prof && !a.build_nodes && a.tie_flags && a.rerun_ctl && wider && !tuning_env("HVX_NO_RERUN");
    if (!rerun) a.rerun_ctl = nullptr; // the search launch lists nothing when nobody empties the list
    hipError_t e = launch_hnsw_wave_once(a, b, s);
    if (e != hipSuccess || !rerun) return e;
    auto reset_list = [&](hipError_t err) { // a failed launch leaves the list to nobody: empty it here (the kernel also ignores entries >= b)
        if (err != hipSuccess) (void)hipMemsetAsync(a.rerun_ctl, 0, 8, s);
        return err;
    };
    HnswArgs r = a;
    r.only_flagged = 1;
    // the re-run's workgroups leave at once (unless duplicates overflowed a beam): give them the build that fits NEXT TO whatever is
    // resident -- two per SIMD, 20 KiB of LDS -- wherever it exists (the strict unrolled builds).  A one-per-SIMD re-run needs a SIMD
    // with nothing else on it: behind the batches of other lanes that is a wait of 0.2 ms (round 3), behind the batcher's lanes
    // running the four-wavefront pair kernel it starved for tens of milliseconds (gpurun r04c: p99 47 ms).
    if (!a.adaptive || (hnsw_wave_supported(a) && !tuning_env("HVX_AD_RERUN_OCC1"))) r.occupancy = 2;
    // the re-run keeps the launch's register budget where the wider build exists for it (two queries per SIMD: its few wavefronts
    // fit next to the resident batches of the other lanes); the 832-entry beams are one-per-SIMD builds
    return reset_list(launch_hnsw_wave_once(r, b, s));
}

} // namespace hvx
