// hvx_toplist.h -- the k <= 64 smallest (score, row) pairs offered to ONE wavefront, kept sorted in a register pair per lane (round 6).
//
// Candidate order (crates/db/src/search/vector/model.rs:55-61): score ascending, then id ascending -- rows are numbered in id order, so
// (score, row) sorts like (score, id).  Lane l holds the l-th smallest pair; unused lanes hold (+inf, kSentinel), so that the rank of a
// new pair is one ballot without a length test.  An insertion = two ballots + one DPP wave shift; a pair whose row is already in the
// list is dropped (RestrictedVectorCandidates is a set, restricted.rs:303-371: a duplicate candidate id counts once).  Shared by the
// one-launch restricted scan (hvx_restricted_exact.hip) and the exact tail of the small-batch scan (hvx_flat_tail.hip).
#pragma once
#include "hvx_beam.h"

namespace hvx {

struct TopList {
    float sc;
    uint32_t id;
    uint32_t count; // uniform
    float thr_s;    // the k-th pair once the list holds k (else +inf / all ones): only pairs before it are admitted
    uint32_t thr_i;
    __device__ __forceinline__ void init() {
        sc = __uint_as_float(0x7F800000u);
        id = kSentinel;
        count = 0;
        thr_s = __uint_as_float(0x7F800000u);
        thr_i = 0xFFFFFFFFu;
    }
    __device__ __forceinline__ bool admits(float d, uint32_t row) const { return d < thr_s || (d == thr_s && row < thr_i); }
    // (d, row) wave-uniform, d finite
    __device__ __forceinline__ void insert(float d, uint32_t row, uint32_t k, int lane) {
        const bool less = (sc < d) | ((sc == d) & (id < row));
        const unsigned long long before = __ballot(less);
        if (__ballot(id == row)) return; // the same candidate twice: a set holds it once
        const uint32_t p = (uint32_t)__builtin_popcountll(before);
        if (p >= k) return;
        const uint32_t ss = shr1(__float_as_uint(sc), 0u), si = shr1(id, 0u);
        const bool at = (uint32_t)lane == p, after = (uint32_t)lane > p;
        sc = at ? d : (after ? __uint_as_float(ss) : sc);
        id = at ? row : (after ? si : id);
        count += count < 64u ? 1u : 0u;
        if (count >= k) {
            thr_s = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(sc), (int)(k - 1u)));
            thr_i = __builtin_amdgcn_readlane(id, (int)(k - 1u));
        }
    }
    // every lane holding a pair with take == true offers it, in lane order
    __device__ __forceinline__ void offer(bool take, float d, uint32_t row, uint32_t k, int lane) {
        unsigned long long m = __ballot(take);
        while (m) {
            const int l = __builtin_ctzll(m);
            m &= m - 1ull;
            const float dd = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), l));
            const uint32_t rr = __builtin_amdgcn_readlane(row, l);
            if (admits(dd, rr)) insert(dd, rr, k, lane);
        }
    }
    // the pair this lane holds, padded
    __device__ __forceinline__ float lane_score() const { return sc; }
    __device__ __forceinline__ uint32_t lane_row() const { return id; }
};

} // namespace hvx
