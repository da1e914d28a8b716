// hvx_beam.h -- the register-resident sorted beam shared by the HNSW kernels.
//
// The reference keeps two BinaryHeaps during the layer-0 search (`candidates` min-heap and `w`
// max-heap, crates/db/src/search/vector/search.rs:500-525).  On CDNA4 both are ONE array sorted by
// Candidate order (score asc, then id asc; model.rs:55-61) striped across the 64 lanes of a wavefront
// (entry e lives in register e/64 of lane e%64): W = the first min(count, ef) entries, `candidates` =
// the entries whose expanded bit is clear.  Insert = ballot/popcount rank + one DPP wave-shift.
#pragma once
#include "hvx_device.h"

namespace hvx {

// wave-shift right by one lane; lane 0 receives `carry`
__device__ __forceinline__ uint32_t shr1(uint32_t v, uint32_t carry) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
}

// The beam: CAP = 64*R entries sorted by (score, id) ascending; id bit31 = already expanded.
template <int R> struct Beam {
    float sc[R];
    uint32_t id[R];
    uint32_t count;

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int r = 0; r < R; ++r) { sc[r] = 0.f; id[r] = 0u; }
        count = 0;
    }
    // uniform broadcast of entry `pos` (pos uniform): R readlanes + scalar selects, no branches
    __device__ __forceinline__ uint32_t bcast_u32(const uint32_t (&v)[R], uint32_t pos) const {
        uint32_t out = __builtin_amdgcn_readlane(v[0], pos & 63u);
#pragma unroll
        for (int r = 1; r < R; ++r) {
            const uint32_t t = __builtin_amdgcn_readlane(v[r], pos & 63u);
            out = (pos >> 6) == (uint32_t)r ? t : out;
        }
        return out;
    }
    __device__ __forceinline__ float score_at(uint32_t pos) const {
        uint32_t out = __builtin_amdgcn_readlane(__float_as_uint(sc[0]), pos & 63u);
#pragma unroll
        for (int r = 1; r < R; ++r) {
            const uint32_t t = __builtin_amdgcn_readlane(__float_as_uint(sc[r]), pos & 63u);
            out = (pos >> 6) == (uint32_t)r ? t : out;
        }
        return __uint_as_float(out);
    }
    __device__ __forceinline__ uint32_t id_at(uint32_t pos) const { return bcast_u32(id, pos); }
    // first entry without the expanded bit, searching from entry `from` on; returns count if none
    __device__ __forceinline__ uint32_t first_unexpanded(int lane, uint32_t from = 0) const {
        uint32_t res = count;
#pragma unroll
        for (int r = R - 1; r >= 0; --r) {
            const uint32_t e = (uint32_t)r * 64u + (uint32_t)lane;
            const unsigned long long m = __ballot((e >= from) & (e < count) & ((id[r] & kExpandedBit) == 0u));
            res = m ? (uint32_t)r * 64u + (uint32_t)__builtin_ctzll(m) : res;
        }
        return res;
    }
    __device__ __forceinline__ void mark_expanded(uint32_t pos, int lane) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            id[r] |= (pos == (uint32_t)r * 64u + (uint32_t)lane) ? kExpandedBit : 0u;
    }
    // sorted insert; returns true when an unexpanded entry fell off the end
    __device__ __forceinline__ bool insert(float d, uint32_t nid, int lane, float &dropped_score) {
        uint32_t p = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t e = (uint32_t)r * 64u + (uint32_t)lane;
            const uint32_t eid = id[r] & ~kExpandedBit;
            const bool less = (e < count) & ((sc[r] < d) | ((sc[r] == d) & (eid < nid))); // bitwise: no exec-mask branches
            p += (uint32_t)__builtin_popcountll(__ballot(less));
        }
        constexpr uint32_t CAP = 64u * R;
        bool dropped = false;
        if (count == CAP) {
            const uint32_t last = __builtin_amdgcn_readlane(id[R - 1], 63);
            dropped_score = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(sc[R - 1]), 63));
            dropped = !(last & kExpandedBit);
            if (p == CAP) { // the new entry itself would land past the end of a full beam
                dropped_score = d;
                return true;
            }
        }
#pragma unroll
        for (int r = R - 1; r >= 0; --r) {
            const uint32_t e = (uint32_t)r * 64u + (uint32_t)lane;
            uint32_t cs = 0, ci = 0;
            if (r > 0) {
                cs = __builtin_amdgcn_readlane(__float_as_uint(sc[r - 1]), 63);
                ci = __builtin_amdgcn_readlane(id[r - 1], 63);
            }
            const uint32_t ss = shr1(__float_as_uint(sc[r]), cs);
            const uint32_t si = shr1(id[r], ci);
            const bool at = e == p, after = e > p;
            const uint32_t keep_s = __float_as_uint(sc[r]), keep_i = id[r];
            sc[r] = __uint_as_float(at ? __float_as_uint(d) : (after ? ss : keep_s));
            id[r] = at ? nid : (after ? si : keep_i);
        }
        count += count < CAP ? 1u : 0u;
        return dropped;
    }
};

} // namespace hvx
