// hvx_hnsw_pair.h -- the HNSW search kernel for ONE batch in flight: two wavefronts per query.
//
// Same algorithm, same results, same counters as hvx_hnsw_wave.h (SearchSession::run, search.rs:1101-1230; greedy upper layers
// :169-224; strict-exhaustive layer-0 beam :267-1067).  What it changes is what a LONE batch of 1 024 queries leaves idle: the
// one-wavefront kernel runs every phase of an expansion back to back on one wavefront per SIMD (pop + visited test 0.8 us, row
// gathers 3.1 us, prediction 0.2 us, admission 0.9 us; profiles/history/r01, docs/next_kernel.md), so the memory system idles while the
// beam is updated and the other half of every SIMD's wave slots is empty.  Here a query is a 128-thread workgroup of two
// wavefronts (2 048 wavefronts = two per SIMD, 256 registers each):
//
//   * the OWNER (wavefront 0) holds everything order-dependent: the sorted beam (hvx_beam.h), the visited table, the counters, the
//     pop / stop rule, the admission loop with its running bound;
//   * the GATHERER (wavefront 1) has no state: it scores the id list the owner publishes in LDS (the same register passes, FMA
//     order and reduction tree as the one-wavefront kernel: bit-identical distances) and publishes the distances.
//
// The overlap: as soon as the distances of expansion i are known the owner names the NEXT pop -- the best admissible fresh
// candidate of this expansion or the best unexpanded beam entry (exact unless scores tie: checked at the pop) --, claims that
// node's unseen neighbours TENTATIVELY in the visited table (VisitedT::claim: a tentative entry still counts as unseen) and
// publishes them; the gatherer scores expansion i + 1 while the owner runs the admission loop of expansion i and the real pop.
// If the pop is the predicted node (97-98 %) the claims are committed with plain LDS stores; if not, the gatherer's pass is
// discarded (its claims stay tentative) and the expansion is redone the plain way.  The visited set is only ever consulted when
// a node is expanded, and expansion i's neighbours are committed before expansion i + 1's are claimed, so nothing the reference
// decides depends on the reordering; counters are taken at the real pop.
//
// Two workgroup barriers per expansion order the hand-over (B0: list published, B1: distances published); every publish by the
// owner is matched by exactly one collect, and the owner's last act is an EXIT token behind a final B0.
//
// Served: the strict arm, f32 / bf16 rows, L2 / cosine, AVX+FMA tree, dim in {128,...,1536}, rows <= 64 ids, ef + 32 <= 384.
// Launched for handles that run one query per SIMD (hvx_index_set_occupancy(1): one batch at a time -- the batcher's small
// batches, hosts that cannot keep several batches in flight); handles with two queries per SIMD keep the one-wavefront kernel,
// whose second resident batch does the overlapping.  The re-run of equal-score overflows uses the one-wavefront kernel's list.
#pragma once
#include "hvx_hnsw_wave.h"

namespace hvx {

constexpr uint32_t kPairExit = 0xFFFFFFFFu;

// G = gatherer wavefronts per query.  G = 1: 128-thread workgroups, two wavefronts per SIMD, 256 registers each, passes of up to 16
// rows.  G = 3 (rows of <= 24 16-byte pieces per lane: f32 up to dim 768, bf16 up to dim 1536): 256-thread workgroups, FOUR
// wavefronts per SIMD at 128 registers; gatherer g takes every third 8-row (16-row for the short rows) slice of the list, so a
// frontier of up to 24 (48) rows is scored with ONE memory latency -- the heavy queries, whose 20-30-row frontiers end a batch, are
// the ones a single 16-row gatherer serves in two round trips.
template <uint32_t METRIC, int R, int NK, bool BF, int G>
__global__ __launch_bounds__(64 * (1 + G)) __attribute__((amdgpu_waves_per_eu(1 + G, 1 + G))) void hnsw_pair_kernel(HnswArgs a, uint32_t vcap) {
    constexpr int NL = BF ? NK / 2 : NK; // 16-byte loads per lane and row
    constexpr int P = NL <= 8 ? 2 : 1;
    constexpr bool kWide2 = G == 1 && 2 * P * NL <= 48 && !(BF && METRIC == kCosine && NL >= 24); // 192 of the gatherer's 256 registers (cosine over bf16 rows at dim 1536 spilled 72: one row per group)
    constexpr uint32_t kThreads = 64u * (1u + G);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevIndex &ix = a.ix;
    const uint32_t q = blockIdx.x;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane >> 3, j = lane & 7;
    const int slot = chunk_slot(j);

    uint32_t *tab = reinterpret_cast<uint32_t *>(smem);
    const uint32_t cap = vcap;
    uint32_t *fr_id = tab + cap;                           // [64] the published id list, row order
    float *fr_d = reinterpret_cast<float *>(fr_id + 64);   // [64] its distances
    uint32_t *ctl = reinterpret_cast<uint32_t *>(fr_d + 64); // [4] ctl[0] = rows of the published list, or kPairExit
    float *qs = reinterpret_cast<float *>(ctl + 4);        // [dim] query, 16-byte aligned

    const unsigned long long wclk0 = a.wave_clock ? wall_clock64() : 0ull;
    const uint32_t status_in = a.qstatus ? a.qstatus[q] : 0u;
    if (status_in != 0u || !ix.has_entry) { // uniform over the workgroup: nobody reaches a barrier
        if (tid == 0) {
            if (a.wave_clock) { a.wave_clock[2 * (size_t)q] = wclk0; a.wave_clock[2 * (size_t)q + 1] = wclk0; }
            a.out_counts[q] = 0;
            if (a.out_status) a.out_status[q] = status_in;
            if (a.qstats) a.qstats[q] = hvx_query_stats{0, 0, 0, 0};
            if (a.tie_flags) a.tie_flags[q] = 0u;
        }
        return;
    }
    const float *qglobal = a.queries + (size_t)q * ix.dim;
    for (uint32_t i = (uint32_t)tid; i < (uint32_t)NK * 8u; i += kThreads)
        reinterpret_cast<float4 *>(qs)[i] = reinterpret_cast<const float4 *>(qglobal)[i];
    __syncthreads();
    const float qhdr = a.qhdr ? a.qhdr[q] : 0.f;
    const float inf = __uint_as_float(0x7F800000u);

    // =========================================== gatherer ===========================================
    if (wave >= 1) {
        auto pass = [&](auto width, uint32_t f0, uint32_t nf) __attribute__((always_inline)) {
            constexpr int W = decltype(width)::value;
            uint32_t nd[W];
            float o[W];
            Gather<NK, W, BF> g;
#pragma unroll
            for (int p = 0; p < W; ++p) {
                const uint32_t f = f0 + (uint32_t)(p * 8 + grp);
                nd[p] = fr_id[f < nf ? f : f0]; // idle groups shadow the pass's first row
            }
            gather_issue<NK, W, BF, W, METRIC == kCosine>(ix, nd, slot, g);
            gather_consume<METRIC, NK, W, BF>(ix, qs, g, nd, slot, qhdr, qglobal, o);
#pragma unroll
            for (int p = 0; p < W; ++p) {
                const uint32_t f = f0 + (uint32_t)(p * 8 + grp);
                if (f < nf && j == 0) fr_d[f] = o[p];
            }
        };
        for (;;) {
            __syncthreads(); // B0: a list is published (or the owner is done)
            const uint32_t nf = ctl[0];
            if (nf == kPairExit) return;
            uint32_t f0 = (uint32_t)(wave - 1) * 8u * P; // G > 1: the gatherers take the list's slices in turn
            while (f0 < nf) {
                const uint32_t rem = nf - f0;
                if (kWide2 && rem > 8u * P) { pass(std::integral_constant<int, kWide2 ? 2 * P : P>{}, f0, nf); f0 += 16u * P; }
                else { pass(std::integral_constant<int, P>{}, f0, nf); f0 += 8u * P * G; }
            }
            __syncthreads(); // B1: the distances are published
        }
    }

    // ============================================= owner =============================================
    using Vis = VisitedT<false>;
    Vis V;
    V.tab = tab;
    V.cap = cap;
    V.limit = cap - (cap >> 2);
    V.words = a.words_per_query;
    V.bm = a.bitmap + (size_t)q * a.words_per_query;
    V.bm2 = nullptr;
    V.count = 0;
    V.spilled = false;

    auto publish = [&](uint32_t nf) __attribute__((always_inline)) { // B0: hand the list in fr_id[0..nf) to the gatherer
        if (lane == 0) ctl[0] = nf;
        __syncthreads();
    };
    auto collect = [&]() __attribute__((always_inline)) { __syncthreads(); }; // B1: its distances are in fr_d
    auto own_fence = [&]() __attribute__((always_inline)) { Vis::vsync(); };
    // distance of ONE node through the gatherer, uniform result
    auto score_one = [&](uint32_t node) __attribute__((always_inline)) -> float {
        if (lane == 0) fr_id[0] = node;
        publish(1u);
        collect();
        const float d = fr_d[0];
        own_fence();
        return d;
    };
    // strict test-and-set + in-order compaction of one neighbour row held one id per lane (upper layers)
    auto frontier_from = [&](uint32_t nid, uint32_t &deg) __attribute__((always_inline)) -> uint32_t {
        const bool valid = nid != kSentinel;
        const bool isnew = V.insert(nid, valid, lane);
        const unsigned long long um = __ballot(isnew);
        deg = (uint32_t)__builtin_popcountll(__ballot(valid));
        if (isnew) fr_id[__builtin_popcountll(um & ((1ull << lane) - 1ull))] = nid;
        return (uint32_t)__builtin_popcountll(um);
    };

    bool bad_score = false;
    uint32_t cur = ix.entry;

    // ---------------- upper layers: search_layer_greedy (search.rs:169-224), as in hvx_hnsw_wave.h ----------------
    float cur_d = 0.f;
    bool have_d = false;
    uint32_t cur_base = kSentinel, cur_level = 0;
    if (ix.max_layer >= 1) {
        cur_base = ix.up_base[cur];
        cur_level = ix.level[cur];
    }
    for (uint32_t layer = ix.max_layer; layer >= 1; --layer) {
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
            own_fence();
            const uint32_t fnode = (uint32_t)lane < nf ? fr_id[lane] : cur;
            const uint32_t f_base = ix.up_base[fnode];
            const uint32_t f_level = ix.level[fnode];
            publish(nf);
            collect();
            // sequential `if distance < current_dist` over the row == first minimum, if it improves
            float d = (uint32_t)lane < nf ? fr_d[lane] : inf;
            own_fence();
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
        }
        if (bad_score) break;
    }

    // ---------------- layer 0: strict-exhaustive beam (search.rs:267-1067), one expansion ahead ----------------
    Beam<R> S;
    S.init();
    uint32_t st_exp = 0, st_nb = 0, st_vl = 0, st_dc = 0;
    bool tie_overflow = false;
    uint32_t dropped_unexpanded = 0;
    const uint32_t ef = a.ef;
    auto load_row = [&](uint32_t c) __attribute__((always_inline)) -> uint32_t {
        return (uint32_t)lane < ix.s0 ? ix.l0[(size_t)c * ix.s0 + (uint32_t)lane] : kSentinel;
    };
    if (!bad_score) {
        V.clear(lane);
        float d0 = have_d ? cur_d : score_one(cur); // search.rs:500-512: the same distance once more
        st_dc = 1;
        if (!score_valid(d0)) bad_score = true;
        V.insert(cur, lane == 0, lane);
        float ds;
        S.insert(d0, cur, lane, ds);
    }
    uint32_t pf_id = kSentinel, pf_row = kSentinel; // the row requested ahead: the entry most likely popped after the next pop
    bool in_flight = false;                         // a published list whose distances have not been collected
    uint32_t nf_cur = 0;                            // rows of the expansion whose distances are in d_l / id_l
    float d_l = inf;
    uint32_t id_l = kSentinel;
    while (!bad_score) {
        // ---- A. name the next pop from what is known now and let the gatherer start on it ----
        const uint32_t wl0 = S.count < ef ? S.count : ef;
        float wmax = wl0 ? S.score_at(wl0 - 1u) : inf; // the bound the admission below starts from
        const uint32_t p2 = S.first_unexpanded(lane);
        uint32_t e2 = kSentinel;
        float s2 = inf;
        if (p2 < S.count) { e2 = S.id_at(p2); s2 = S.score_at(p2); }
        uint32_t next = e2;
        {
            const bool adm = ((uint32_t)lane < nf_cur) & ((d_l < wmax) | (wl0 < ef)) & (d_l >= 0.f) & (d_l < inf);
            const uint32_t key = adm ? __float_as_uint(d_l) : 0xFFFFFFFFu;
            const uint32_t kmin = wave_umin(key);
            if (kmin != 0xFFFFFFFFu && __uint_as_float(kmin) < s2)
                next = __builtin_amdgcn_readlane(id_l, (uint32_t)__builtin_ctzll(__ballot(key == kmin)));
        }
        bool have_spec = false;
        uint32_t spec_nid = kSentinel, spec_slot = 0, spec_deg = 0, spec_nf = 0;
        bool spec_unseen = false;
        if (next != kSentinel) {
            spec_nid = next == pf_id ? pf_row : load_row(next);
            const bool valid = spec_nid != kSentinel;
            bool hashed;
            spec_unseen = V.claim(spec_nid, valid, lane, spec_slot, hashed); // tentative: still unseen for everybody else
            const unsigned long long um = __ballot(spec_unseen);
            spec_deg = (uint32_t)__builtin_popcountll(__ballot(valid));
            spec_nf = (uint32_t)__builtin_popcountll(um);
            if (spec_unseen) fr_id[__builtin_popcountll(um & ((1ull << lane) - 1ull))] = spec_nid;
            have_spec = true;
            if (spec_nf) { publish(spec_nf); in_flight = true; }
        }
        // ---- B. admission of the current expansion, in row order with the running bound (search.rs:928-952) ----
        if (nf_cur) {
            const bool okv = (uint32_t)lane >= nf_cur || score_valid(d_l);
            if (__ballot(!okv)) { bad_score = true; break; }
            unsigned long long todo = __ballot(((uint32_t)lane < nf_cur) & ((d_l < wmax) | (wl0 < ef)));
            while (todo) {
                const uint32_t f = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1ull;
                const float d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d_l), f));
                const uint32_t wl = S.count < ef ? S.count : ef;
                if (d < wmax || wl < ef) {
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
        // ---- C. the real pop (search.rs:537-594) ----
        const uint32_t pos = S.first_unexpanded(lane);
        if (pos >= S.count) {
            if (dropped_unexpanded) ++st_exp; // the reference would still pop an evicted candidate, count the step and stop
            break;
        }
        ++st_exp;
        const float dc = S.score_at(pos);
        const uint32_t c = S.id_at(pos);
        const uint32_t wlen = S.count < ef ? S.count : ef;
        if (wlen >= ef && dc > S.score_at(wlen - 1)) break;
        S.mark_expanded(pos, lane);
        if (have_spec && c == next) {
            V.commit(spec_nid, spec_unseen, spec_slot); // the claims become visited: plain LDS stores
            st_nb += spec_deg;
            nf_cur = spec_nf;
        } else {
            if (in_flight) { collect(); in_flight = false; } // the gatherer scored the wrong node: discard (its claims stay tentative)
            own_fence();
            const uint32_t nid = c == pf_id ? pf_row : load_row(c);
            const bool valid = nid != kSentinel;
            uint32_t vslot = 0;
            bool hashed;
            const bool unseen = V.claim(nid, valid, lane, vslot, hashed);
            V.commit(nid, unseen, vslot);
            const unsigned long long um = __ballot(unseen);
            st_nb += (uint32_t)__builtin_popcountll(__ballot(valid));
            nf_cur = (uint32_t)__builtin_popcountll(um);
            if (unseen) fr_id[__builtin_popcountll(um & ((1ull << lane) - 1ull))] = nid;
            if (nf_cur) { publish(nf_cur); in_flight = true; }
        }
        // the row of the entry most likely popped after this one goes out under the gatherer's pass
        {
            const uint32_t p3 = S.first_unexpanded(lane);
            pf_id = kSentinel;
            pf_row = kSentinel;
            if (p3 < S.count) { pf_id = S.id_at(p3); pf_row = load_row(pf_id); }
        }
        st_vl += nf_cur;
        st_dc += nf_cur;
        if (nf_cur == 0) { d_l = inf; id_l = kSentinel; continue; } // nothing new to score: straight to the next pop
        collect(); // B1
        in_flight = false;
        d_l = (uint32_t)lane < nf_cur ? fr_d[lane] : inf;
        id_l = (uint32_t)lane < nf_cur ? fr_id[lane] : kSentinel;
        own_fence(); // the list is in registers before the next one overwrites fr_id
    }
    if (in_flight) collect();
    if (lane == 0) ctl[0] = kPairExit;
    __syncthreads(); // the gatherer's last B0

    // ---------------- results: w sorted by (score,id), take k (search.rs:995-1004,1229) ----------------
    uint32_t outn = 0;
    if (!bad_score) {
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
    if (V.spilled) { // leave the HBM bitmap zeroed for the next launch
        for (uint32_t w = (uint32_t)lane; w < V.words; w += 64) V.bm[w] = 0u;
    }
    if (lane == 0) {
        if (a.wave_clock) { a.wave_clock[2 * (size_t)q] = wclk0; a.wave_clock[2 * (size_t)q + 1] = wall_clock64(); }
        a.out_counts[q] = outn;
        if (a.out_status) a.out_status[q] = bad_score ? 8u /*HVX_ERR_INVARIANT*/ : 0u;
        if (a.qstats) a.qstats[q] = hvx_query_stats{st_exp, st_nb, st_vl, st_dc};
        if (a.tie_flags) a.tie_flags[q] = tie_overflow ? 1u : 0u;
        if (tie_overflow && a.rerun_ctl) a.rerun_list[atomicAdd(&a.rerun_ctl[0], 1u)] = q;
    }
}

template <typename K> static hipError_t launch_pair_kernel(K kern, uint32_t threads, const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    if (g.lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(b), dim3(threads), g.lds, s, a, g.cap);
    return hipGetLastError();
}
// three gatherers where a row is <= 24 pieces per lane, else one
template <uint32_t METRIC, int R, int NK, bool BF> static hipError_t launch_pair_g(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    constexpr int NL = BF ? NK / 2 : NK;
    if constexpr (NL <= 24) {
        if (a.pair_gatherers != 1u) return launch_pair_kernel(hnsw_pair_kernel<METRIC, R, NK, BF, 3>, 256u, a, b, g, s);
    }
    return launch_pair_kernel(hnsw_pair_kernel<METRIC, R, NK, BF, 1>, 128u, a, b, g, s);
}

template <uint32_t METRIC, int R, bool BF> static hipError_t launch_pair_nk(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    switch (a.ix.dim >> 5) {
    case 4: return launch_pair_g<METRIC, R, 4, BF>(a, b, g, s);
    case 8: return launch_pair_g<METRIC, R, 8, BF>(a, b, g, s);
    case 12: return launch_pair_g<METRIC, R, 12, BF>(a, b, g, s);
    case 16: return launch_pair_g<METRIC, R, 16, BF>(a, b, g, s);
    case 24: return launch_pair_g<METRIC, R, 24, BF>(a, b, g, s);
    case 32: return launch_pair_g<METRIC, R, 32, BF>(a, b, g, s);
    case 48: return launch_pair_g<METRIC, R, 48, BF>(a, b, g, s);
    default: return hipErrorInvalidValue;
    }
}
template <uint32_t METRIC, bool BF> static hipError_t launch_pair_r(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    const uint32_t need = a.ef + 32u;
    if (need <= 192) return launch_pair_nk<METRIC, 3, BF>(a, b, g, s);
    if (need <= 384) return launch_pair_nk<METRIC, 6, BF>(a, b, g, s);
    return hipErrorInvalidValue;
}

} // namespace hvx
