// hvx_walk_core.h -- the filter-aware restricted walk as ONE algorithm body that is compiled twice:
// by hipcc into restricted_walk_kernel (hvx_restricted_walk.hip; one 256-thread workgroup per query) and by g++ into the
// test-only host twin (tests/native/walk_twin.cpp), which runs the very same phases with the workgroup's threads executed
// one after another (in a shuffled order, so that a phase that depends on thread order shows up on the CPU).
//
// Reference: crates/db/src/search/vector/restricted.rs
//   :837-1148 restricted_filter_aware_search        :615-659 restricted_candidate_keys
//   :661-704  restricted_score_keys                 :711-751 restricted_enqueue_bridges
//   :230-259  FilteredGraphBudgets                  :455-462 directory_prefix_offsets
//   storage.rs:1942-2010 simhash_directory_window_measured (rows in (order_code, node_id) order, row + byte caps)
//
// How the reference's per-query containers map onto a workgroup (all exact, none approximate):
//   * `frontier` (min-heap of every scored candidate) and `top` (max-heap of the best ef_filtered) are ONE array `S` of
//     64-bit keys  score_bits << 32 | row << 1 | expanded  kept sorted in LDS: top = its first min(|S|, ef_filtered)
//     entries, frontier = the entries without the expanded bit.  At most `vector_payloads` (<= 800) rows are ever scored.
//   * `attempted` (members) and `bridge_state.queued` (non-members) are disjoint by construction, so both are ONE bitmap
//     `seen` over the index rows, per query, in HBM.  `scored`, `expanded` and the SimHash cache never reject anything a
//     second time (every scored id was attempted exactly once, every frontier entry is popped at most once, a member is
//     never queued as a bridge): they exist only as the counters they feed.
//   * `eligible_seen` + `eligible.truncate(cut)` + `attempted.extend`: the first `cut` eligible ids in (batch, row) order
//     are the only ones that matter, so an id is marked attempted the moment it is appended; ids past the cut are
//     dropped unmarked, exactly like the reference's truncated tail.
//   * `bridge_state.frontier` (min-heap by (hamming, node id)) is a sorted array `B` bounded by the bridge-row budget that
//     is left: an entry ranked beyond the number of pops that can still happen can never be popped (pushes only move it
//     back, every pop moves it and the budget by one), so it is counted (`bridge_n`, the heap's true length: the
//     termination tests only ask whether it is empty) but not stored.  Node ids ascend with the row number.
//
// Phase discipline (what makes the host twin a faithful model): inside one `c.phase` a thread writes only locations no
// other thread reads or writes in that phase (atomics excepted); everything a later phase or the uniform control flow
// reads was written in an EARLIER phase.  Uniform values (counters, budgets, statistics) live in plain locals that every
// thread computes identically from LDS words written by a completed phase.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HVX_WALK_FN __device__ __forceinline__          // the algorithm body: device only in the product
#define HVX_WALK_HD __host__ __device__ __forceinline__ // small helpers the library's host code shares (directory build)
#else
#define HVX_WALK_FN inline
#define HVX_WALK_HD inline
#endif

namespace hvx {
namespace walk {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kFrontierBatch = 16;    // FRONTIER_BATCH_SIZE     restricted.rs:50
constexpr uint32_t kBridgeBatch = 256;     // BRIDGE_BATCH_SIZE       restricted.rs:51
constexpr uint32_t kMaxProbes = 64;        // DIRECTORY_MAX_PROBES    restricted.rs:45
constexpr uint32_t kConcurrentScans = 8;   // DIRECTORY_MAX_CONCURRENT_SCANS :48
constexpr uint32_t kWindowRows = 65536 / kMaxProbes;            // DIRECTORY_MAX_ROWS / probes
constexpr uint32_t kWindowBytes = (4u * 1024 * 1024) / kMaxProbes;
constexpr uint32_t kDirectoryRowBytes = 27; // 26-byte key + 1-byte V1 marker (values/vectors/markers.rs:36-38)
constexpr uint32_t kScoredCap = 1024;      // >= vector_payloads (FILTERED_VECTOR_PAYLOAD_LIMIT = 800)
constexpr uint32_t kStageCap = 2048;       // staging of new bridge keys / directory seeds between two sorts
constexpr uint32_t kSeedCap = 1024;        // largest sampled_seeds / directory_seeds an explicit budget may ask for

enum : uint32_t { kTermNone = 0, kTermExhausted = 1, kTermBeamComplete = 2, kTermRoutingBudget = 3, kTermBridgeBudget = 4, kTermVectorBudget = 5 };

// FilteredGraphBudgets (restricted.rs:216-259) + what the planner resolved for this query
struct Plan {
    uint32_t k, ef_filtered, routing_rows, bridge_rows, vector_payloads, sampled_seeds, directory_seeds;
    uint32_t directory_enabled; // VectorIndex::simhash_directory_enabled()
    uint32_t n_sample;          // entries of the query's sample list (= min(sampled_seeds, candidates))
};

// RestrictedSearchStats (restricted.rs:147-166) as the kernel counts them (32-bit is ample: every counter is bounded by a
// budget); widened into hvx_restricted_stats on the way out
struct Counters {
    uint32_t termination, directory_scan_calls, directory_rows, directory_decoded_bytes, directory_hits, simhash_row_requests,
        companion_row_requests, routing_rows, bridge_rows, bridge_frontier_pushes, neighbor_multi_get_calls,
        vector_payload_requests, vector_bytes, distance_computations;
};

struct View {
    const uint32_t *l0;        // [n][s0] layer-0 rows, internal ids ascending, kNone padded
    uint32_t s0, n, dim;
    const uint64_t *node_hash; // [n] SimHash row of every node
    const uint64_t *dir_code;  // [n] order codes ascending  } the SimHash directory: (order_code, node) rows in key order
    const uint32_t *dir_row;   // [n] the row of each entry   }
    const uint32_t *dir_prefix; // [65537] first directory entry whose 16-bit order-code prefix is >= p (entry 65536 = n): window bounds
                                //         without a binary search over the directory in HBM (round 4)
    uint32_t entry, has_entry;
    const uint32_t *allowed;   // [words] membership bitmap of this query's candidate set over the index rows (read-only)
    uint32_t *seen;            // [words] attempted (members) / queued (non-members), zero on entry
};

// workgroup memory (LDS on the device)
struct Mem {
    uint64_t *S;     // [kScoredCap]
    uint64_t *B;     // [b_cap]
    uint64_t *G;     // [kStageCap]
    uint64_t *N;     // [kScoredCap] keys of the rows scored last, before they are merged into S (may alias rows + scan: dead then)
    uint32_t *Gr;    // [kStageCap]
    uint32_t *E;     // [kScoredCap]
    uint32_t *rows;  // [W]
    uint32_t *scan;  // [W]
    uint8_t *flag;   // [W]
    uint64_t *tv;    // [W]
    uint32_t *tp;    // [W]
    uint32_t *batch; // [kBridgeBatch]
    uint32_t *ctl;   // [kCtlWords]
    uint32_t b_cap;
    uint32_t stage_cap; // entries of G / Gr (>= kStageCap; >= 2 W for the wide geometry)
};

enum : uint32_t { kGN = 0, kPush = 1, kBad = 2, kUnknown = 3, kTmp = 4, kMin = 5, kWinLo = 8, kWinCnt = 16, kWinHit = 24, kCtlWords = 32 };

// order_code_from_simhash_bits (simhash.rs:44-59): the four 16-bit bands interleaved MSB first
HVX_WALK_HD uint64_t order_code(uint64_t bits) {
    uint64_t code = 0;
    for (int bit = 15; bit >= 0; --bit) {
        code = (code << 1) | ((bits >> (48 + bit)) & 1u);
        code = (code << 1) | ((bits >> (32 + bit)) & 1u);
        code = (code << 1) | ((bits >> (16 + bit)) & 1u);
        code = (code << 1) | ((bits >> bit) & 1u);
    }
    return code;
}

// directory_prefix_offsets()[i] for i < 64 (restricted.rs:455-462: all u16 sorted by (popcount, value)): 0, the 16
// single-bit values ascending, then the two-bit values ascending
HVX_WALK_HD uint32_t prefix_offset(uint32_t i) {
    if (i == 0) return 0u;
    if (i <= 16) return 1u << (i - 1);
    uint32_t r = i - 17, h = 1;
    while (r >= h) { r -= h; ++h; } // two-bit values ascending: high bit h = 1, 2, ... with low bit 0..h-1
    return (1u << h) | (1u << r);
}

HVX_WALK_HD uint32_t popc64(uint64_t x) {
    return (uint32_t)__builtin_popcountll(x);
}
HVX_WALK_HD uint32_t pow2_ceil(uint32_t x) {
    uint32_t p = 1;
    while (p < x) p <<= 1;
    return p;
}
HVX_WALK_HD uint32_t lower_bound(const uint64_t *a, uint32_t n, uint64_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
HVX_WALK_HD uint32_t lower_bound(const uint32_t *a, uint32_t n, uint32_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
HVX_WALK_HD bool row_contains(const uint32_t *row, uint32_t s0, uint32_t x) { // rows ascend, kNone (the largest u32) pads
    const uint32_t p = lower_bound(row, s0, x);
    return p < s0 && row[p] == x;
}
HVX_WALK_HD bool bit_of(const uint32_t *bm, uint32_t x) { return (bm[x >> 5] >> (x & 31u)) & 1u; }
HVX_WALK_HD uint32_t sat_sub(uint32_t a, uint32_t b) { return a > b ? a - b : 0u; }
HVX_WALK_HD uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// The context `C` supplies the workgroup:
//   C::T                      threads;  C::W  items handled per collective step (a multiple of T)
//   c.phase(f)                f(tid) on every thread, a barrier before and after
//   c.atomic_or_global(p, v)  -> old     c.atomic_add_shared(p, v) -> old     c.atomic_min_shared(p, v)
//   c.atomic_cas_shared(p, expected, desired) -> old
//   c.load_seen(p)            coherent read of a word other threads of the workgroup may have OR-ed into in an earlier phase
//   c.scan(a, len, &total)    in-place exclusive prefix sum over a[0..len), len <= W (collective)
//   c.sort64(a, L)            ascending sort of L (a power of two) keys (collective)
//   c.score(rows, n, keys)    keys[i] = score_bits(rows[i]) << 32 | rows[i] << 1; returns non-zero (uniform) if a score was
//                             rejected by DistanceScore::try_new (collective)
template <class C>
HVX_WALK_FN uint32_t run(C &c, const View &v, const Plan &pl, const uint32_t *sample_rows, uint64_t qhash, const Mem &m,
                         Counters &st, uint32_t &s_n_out) {
    constexpr uint32_t T = C::T, W = C::W;
    st = Counters{};
    uint32_t s_n = 0;                 // scored candidates in S
    uint32_t head = 0, nb = 0;        // B[head, head + nb) = the stored part of the bridge frontier; head = bridge rows popped
    uint32_t bridge_n = 0;            // true length of bridge_state.frontier
    uint32_t bad = 0;
    s_n_out = 0;

    c.phase([&](uint32_t t) {
        for (uint32_t i = t; i < kCtlWords; i += T) m.ctl[i] = 0;
    });

    auto pad_sort = [&](uint64_t *a, uint32_t n) {
        if (n < 2) return;
        const uint32_t L = pow2_ceil(n);
        if (L > n)
            c.phase([&](uint32_t t) {
                for (uint32_t i = n + t; i < L; i += T) a[i] = ~0ull;
            });
        c.sort64(a, L);
    };

    // merge the sorted keys X[0, nx) into the sorted array A[0, n) (keys are unique), keeping at most `cap` entries: an entry of
    // A moves up by the number of new keys below it (highest block first, so a block's targets have been vacated: positions
    // strictly increase with the entry), a new key lands at its index + the number of A's entries below it
    auto merge_into = [&](uint64_t *A, uint32_t &n, uint32_t cap, const uint64_t *X, uint32_t nx) {
        for (uint32_t base = 0; base < nx; base += T)
            c.phase([&](uint32_t t) {
                const uint32_t i = base + t;
                if (i < nx) m.Gr[i] = i + lower_bound(A, n, X[i]);
            });
        const uint32_t jmin = lower_bound(A, n, X[0]);
        for (uint32_t blk = (n + W - 1) / W; blk-- > jmin / W;) {
            c.phase([&](uint32_t t) {
                for (uint32_t s = t; s < W; s += T) {
                    const uint32_t j = blk * W + s;
                    if (j < n) {
                        const uint64_t x = A[j];
                        m.tv[s] = x;
                        m.tp[s] = j + lower_bound(X, nx, x);
                    }
                }
            });
            c.phase([&](uint32_t t) {
                for (uint32_t s = t; s < W; s += T) {
                    const uint32_t j = blk * W + s;
                    if (j < n && m.tp[s] < cap) A[m.tp[s]] = m.tv[s];
                }
            });
        }
        c.phase([&](uint32_t t) {
            for (uint32_t i = t; i < nx; i += T)
                if (m.Gr[i] < cap) A[m.Gr[i]] = X[i];
        });
        n = umin(n + nx, cap);
    };

    // restricted_score_keys (restricted.rs:661-704) for E[0, ne): the new keys are sorted on their own and merged into S
    auto score_list = [&](uint32_t ne) {
        if (ne == 0 || bad) return;
        st.vector_payload_requests += ne;
        st.vector_bytes += ne * (4u + v.dim * 4u);
        st.distance_computations += ne;
        if (c.score(m.E, ne, m.N)) { bad = 1; return; }
        pad_sort(m.N, ne);
        merge_into(m.S, s_n, kScoredCap, m.N, ne);
    };

    // merge the staged bridge keys G[0, ctl[kGN]) into B
    auto flush = [&]() {
        const uint32_t ng = m.ctl[kGN];
        if (ng == 0) return;
        const uint32_t cap_rem = m.b_cap - head;
        if (cap_rem != 0) {
            pad_sort(m.G, ng);
            merge_into(m.B + head, nb, cap_rem, m.G, ng);
        }
        c.phase([&](uint32_t t) {
            if (t == 0) m.ctl[kGN] = 0;
        });
    };

    // ---------------------------------------------------------------- directory seeds (restricted.rs:866-923)
    uint32_t n_ds = 0;
    if (pl.directory_enabled) {
        const uint64_t qorder = order_code(qhash);
        const uint32_t qprefix = (uint32_t)(qorder >> 48);
        bool done = false;
        for (uint32_t p0 = 0; p0 < kMaxProbes && !done; p0 += kConcurrentScans) {
            // the 8 windows of a batch are all read (try_join_all) before any is consumed
            c.phase([&](uint32_t t) {
                for (uint32_t w = t; w < kConcurrentScans; w += T) {
                    const uint32_t prefix = qprefix ^ prefix_offset(p0 + w);
                    const uint32_t lo = v.dir_prefix[prefix];
                    const uint32_t hi = v.dir_prefix[prefix + 1u]; // (prefix 0xFFFF: entry 65536 = n)
                    uint32_t cnt = hi - lo;
                    if (cnt > kWindowRows) cnt = kWindowRows;
                    if (cnt > kWindowBytes / kDirectoryRowBytes) cnt = kWindowBytes / kDirectoryRowBytes;
                    m.ctl[kWinLo + w] = lo;
                    m.ctl[kWinCnt + w] = cnt;
                    m.ctl[kWinHit + w] = 0;
                }
            });
            st.directory_scan_calls += kConcurrentScans;
            for (uint32_t w = 0; w < kConcurrentScans; ++w) {
                st.directory_rows += m.ctl[kWinCnt + w];
                st.directory_decoded_bytes += m.ctl[kWinCnt + w] * kDirectoryRowBytes;
            }
            // Round 4: the windows are consumed in order and the scan stops behind the first one that completes the seed quota --
            // so the members of ALL eight windows are counted in one phase (one dependent pair of HBM reads for the whole batch
            // instead of one per window), the uniform control flow picks the windows that are consumed, and a second phase stores
            // exactly their members: the same set the window-by-window loop produced (a node has one directory row: the
            // reference's `seen.insert` never rejects).
            c.phase([&](uint32_t t) {
                for (uint32_t s = t; s < kConcurrentScans * kWindowRows; s += T) {
                    const uint32_t w = s / kWindowRows, i = s - w * kWindowRows;
                    if (i < m.ctl[kWinCnt + w] && bit_of(v.allowed, v.dir_row[m.ctl[kWinLo + w] + i])) c.atomic_add_shared(&m.ctl[kWinHit + w], 1u);
                }
            });
            uint32_t used = 0;
            for (uint32_t w = 0; w < kConcurrentScans && !done; ++w) {
                n_ds += m.ctl[kWinHit + w];
                used = w + 1;
                if (n_ds >= pl.directory_seeds) done = true;
            }
            c.phase([&](uint32_t t) {
                for (uint32_t s = t; s < used * kWindowRows; s += T) {
                    const uint32_t w = s / kWindowRows, i = s - w * kWindowRows;
                    if (i < m.ctl[kWinCnt + w] && m.ctl[kWinHit + w] != 0u) {
                        const uint32_t lo = m.ctl[kWinLo + w], x = v.dir_row[lo + i];
                        if (bit_of(v.allowed, x)) {
                            const uint32_t pos = c.atomic_add_shared(&m.ctl[kGN], 1u);
                            m.G[pos] = ((uint64_t)popc64(v.dir_code[lo + i] ^ qorder) << 32) | x;
                        }
                    }
                }
            });
        }
        st.directory_hits = n_ds;
        pad_sort(m.G, n_ds); // sort_unstable_by_key((hamming, node_id)); rows ascend with node ids
        if (n_ds > pl.directory_seeds) n_ds = pl.directory_seeds;
        c.phase([&](uint32_t t) {
            if (t == 0) m.ctl[kGN] = 0;
        });
    }

    // ---------------------------------------------------------------- initial keys (restricted.rs:925-962)
    uint32_t n_init = 0;
    const uint32_t vp = pl.vector_payloads;
    // deterministic sample: every id is `attempted`; ids that are not indexed cost a SimHash read and a companion-row probe
    for (uint32_t base = 0; base < pl.n_sample; base += W) {
        const uint32_t len = umin(W, pl.n_sample - base);
        c.phase([&](uint32_t t) {
            for (uint32_t s = t; s < len; s += T) {
                const uint32_t r = sample_rows[base + s];
                uint32_t f = 0;
                if (r != kNone) {
                    c.atomic_or_global(&v.seen[r >> 5], 1u << (r & 31u));
                    f = 1;
                } else {
                    c.atomic_add_shared(&m.ctl[kUnknown], 1u);
                }
                m.flag[s] = (uint8_t)f;
                m.scan[s] = f;
            }
        });
        uint32_t total = 0;
        c.scan(m.scan, len, &total);
        c.phase([&](uint32_t t) {
            for (uint32_t s = t; s < len; s += T)
                if (m.flag[s]) {
                    const uint32_t pos = n_init + m.scan[s];
                    if (pos < vp) m.E[pos] = sample_rows[base + s];
                }
        });
        n_init += total;
    }
    st.simhash_row_requests += pl.n_sample;
    {
        const uint32_t unknown = m.ctl[kUnknown];
        st.companion_row_requests += unknown;
        if (unknown) st.neighbor_multi_get_calls += 1; // layer0_rows_exist of the ids without a SimHash row
    }
    // directory seeds carry their own key: no SimHash read
    for (uint32_t base = 0; base < n_ds; base += W) {
        const uint32_t len = umin(W, n_ds - base);
        c.phase([&](uint32_t t) {
            for (uint32_t s = t; s < len; s += T) {
                const uint32_t x = (uint32_t)m.G[base + s];
                const uint32_t bit = 1u << (x & 31u);
                const uint32_t f = (c.atomic_or_global(&v.seen[x >> 5], bit) & bit) ? 0u : 1u;
                m.flag[s] = (uint8_t)f;
                m.scan[s] = f;
            }
        });
        uint32_t total = 0;
        c.scan(m.scan, len, &total);
        c.phase([&](uint32_t t) {
            for (uint32_t s = t; s < len; s += T)
                if (m.flag[s]) {
                    const uint32_t pos = n_init + m.scan[s];
                    if (pos < vp) m.E[pos] = (uint32_t)m.G[base + s];
                }
        });
        n_init += total;
    }
    const bool entry_allowed = bit_of(v.allowed, v.entry);
    if (entry_allowed) {
        c.phase([&](uint32_t t) {
            if (t == 0) {
                const uint32_t bit = 1u << (v.entry & 31u);
                const uint32_t fresh = (c.atomic_or_global(&v.seen[v.entry >> 5], bit) & bit) ? 0u : 1u;
                m.ctl[kTmp] = fresh;
                if (fresh && n_init < vp) m.E[n_init] = v.entry;
            }
        });
        if (m.ctl[kTmp]) {
            n_init += 1;
            st.simhash_row_requests += 1;
        }
    }
    if (n_init > vp) n_init = vp; // initial_keys.truncate(budgets.vector_payloads)
    score_list(n_init);

    // stage one rejected node for the bridge frontier (restricted_enqueue_bridges); runs inside a phase
    auto stage_bridge = [&](uint32_t x, bool store_all, uint32_t cap_rem, uint64_t bmax) {
        const uint32_t bit = 1u << (x & 31u);
        if (c.atomic_or_global(&v.seen[x >> 5], bit) & bit) return; // already queued
        c.atomic_add_shared(&m.ctl[kPush], 1u);
        const uint64_t key = ((uint64_t)popc64(v.node_hash[x] ^ qhash) << 32) | x;
        if (store_all || (cap_rem != 0 && key < bmax)) m.G[c.atomic_add_shared(&m.ctl[kGN], 1u)] = key;
    };
    auto account_pushes = [&]() {
        const uint32_t pushes = m.ctl[kPush];
        st.bridge_frontier_pushes += pushes;
        st.simhash_row_requests += pushes;
        bridge_n += pushes;
    };

    if (!bad && !entry_allowed) {
        c.phase([&](uint32_t t) {
            if (t == 0) stage_bridge(v.entry, true, m.b_cap, 0);
        });
        account_pushes();
        flush();
    }

    // ---------------------------------------------------------------- the walk (restricted.rs:992-1128)
    uint32_t n_elig = 0, cut = 0;
    // the layer-0 rows of batch[0, nrows): members -> eligible (first `cut`, in (batch, row) order), the rest -> bridges
    auto process_rows = [&](uint32_t nrows) {
        const uint32_t CH = W / v.s0; // rows per chunk
        for (uint32_t r0 = 0; r0 < nrows; r0 += CH) {
            const uint32_t rc = umin(CH, nrows - r0), slots = rc * v.s0;
            c.phase([&](uint32_t t) {
                if (t == 0) m.ctl[kPush] = 0;
                for (uint32_t s = t; s < slots; s += T) {
                    const uint32_t r = s / v.s0, j = s - r * v.s0;
                    m.rows[s] = v.l0[(size_t)m.batch[r0 + r] * v.s0 + j];
                }
            });
            const bool full = n_elig >= cut;
            const uint32_t cap_rem = m.b_cap - head;
            const bool store_all = nb < cap_rem;
            const uint64_t bmax = nb ? m.B[head + nb - 1] : 0ull;
            // eligible_seen inside the chunk: of the slots that list a member x, the FIRST in (row, position) order wins.  Round 4: an
            // open-addressing table in LDS (2 W cells: ids in the words of `tv`, smallest listing slot in `Gr` -- both are scratch of
            // merge_into, dead here) replaces "binary-search every earlier row of the chunk" (up to 63 searches per listed member):
            // claim the id's cell, atomic-min the slot index into it, read the winner back in the next phase.
            uint32_t *hk = reinterpret_cast<uint32_t *>(m.tv), *hv = m.Gr;
            constexpr uint32_t HC = 2u * W;
            if (!full)
                c.phase([&](uint32_t t) {
                    for (uint32_t i = t; i < HC; i += T) { hk[i] = kNone; hv[i] = kNone; }
                });
            c.phase([&](uint32_t t) {
                for (uint32_t s = t; s < slots; s += T) {
                    const uint32_t x = m.rows[s];
                    uint32_t f = 0;
                    if (x != kNone) {
                        if (bit_of(v.allowed, x)) {
                            if (!full && !((c.load_seen(&v.seen[x >> 5]) >> (x & 31u)) & 1u)) {
                                uint32_t h = ((x * 2654435761u) >> 9) & (HC - 1u);
                                for (;;) {
                                    const uint32_t old = c.atomic_cas_shared(&hk[h], kNone, x);
                                    if (old == kNone || old == x) break;
                                    h = (h + 1u) & (HC - 1u);
                                }
                                c.atomic_min_shared(&hv[h], s);
                                m.tp[s] = h;
                                f = 2u; // a member not yet attempted: eligible if it is the first slot that lists it
                            }
                        } else {
                            stage_bridge(x, store_all, cap_rem, bmax);
                        }
                    }
                    m.flag[s] = (uint8_t)f;
                }
            });
            c.phase([&](uint32_t t) {
                for (uint32_t s = t; s < slots; s += T) {
                    const uint32_t f = (m.flag[s] == 2u && hv[m.tp[s]] == s) ? 1u : 0u;
                    m.flag[s] = (uint8_t)f;
                    m.scan[s] = f;
                }
            });
            account_pushes();
            if (!full) {
                uint32_t total = 0;
                c.scan(m.scan, slots, &total);
                c.phase([&](uint32_t t) {
                    for (uint32_t s = t; s < slots; s += T)
                        if (m.flag[s]) {
                            const uint32_t pos = n_elig + m.scan[s];
                            if (pos < cut) {
                                const uint32_t x = m.rows[s];
                                m.E[pos] = x;
                                c.atomic_or_global(&v.seen[x >> 5], 1u << (x & 31u)); // attempted
                            }
                        }
                });
                n_elig = umin(cut, n_elig + total);
            }
            if (m.ctl[kGN] + W > m.stage_cap) flush();
        }
    };

    while (!bad) {
        if (st.vector_payload_requests >= vp) { st.termination = kTermVectorBudget; break; }
        const uint32_t top_n = umin(s_n, pl.ef_filtered);
        if (bridge_n == 0 && top_n >= pl.ef_filtered && top_n != 0) {
            // frontier.peek() = the smallest entry without the expanded bit; top.peek() = S[ef_filtered - 1]
            c.phase([&](uint32_t t) {
                if (t == 0) m.ctl[kMin] = kNone;
            });
            c.phase([&](uint32_t t) {
                for (uint32_t i = t; i < s_n; i += T)
                    if (!(m.S[i] & 1ull)) { c.atomic_min_shared(&m.ctl[kMin], i); break; }
            });
            const uint32_t fu = m.ctl[kMin];
            if (fu != kNone && (m.S[fu] | 1ull) > (m.S[pl.ef_filtered - 1] | 1ull)) { st.termination = kTermBeamComplete; break; }
        }
        // routing batch: up to 16 pops of the frontier
        uint32_t nr = 0;
        for (uint32_t base = 0; base < s_n && nr < kFrontierBatch; base += W) {
            const uint32_t len = umin(W, s_n - base);
            c.phase([&](uint32_t t) {
                for (uint32_t s = t; s < len; s += T) {
                    const uint32_t f = (m.S[base + s] & 1ull) ? 0u : 1u;
                    m.flag[s] = (uint8_t)f;
                    m.scan[s] = f;
                }
            });
            uint32_t total = 0;
            c.scan(m.scan, len, &total);
            c.phase([&](uint32_t t) {
                for (uint32_t s = t; s < len; s += T)
                    if (m.flag[s]) {
                        const uint32_t pos = nr + m.scan[s];
                        if (pos < kFrontierBatch) {
                            m.batch[pos] = (uint32_t)(m.S[base + s] >> 1) & 0x7FFFFFFFu;
                            m.S[base + s] |= 1ull;
                        }
                    }
            });
            nr = umin(kFrontierBatch, nr + total);
        }
        if (nr == 0 && bridge_n == 0) { st.termination = kTermExhausted; break; }
        uint32_t routing_remaining = sat_sub(pl.routing_rows, st.routing_rows);
        if (routing_remaining == 0) { st.termination = kTermRoutingBudget; break; }
        if (nr > routing_remaining) nr = routing_remaining; // routing_batch.truncate: the popped tail stays expanded
        n_elig = 0;
        cut = umin(sat_sub(vp, st.vector_payload_requests), pl.ef_filtered);
        if (nr) {
            st.routing_rows += nr;
            st.neighbor_multi_get_calls += 1;
            routing_remaining -= nr;
            process_rows(nr);
        }
        flush(); // the bridge frontier must hold everything queued so far before it is popped (one merge per iteration)
        const uint32_t bl = umin(umin(sat_sub(pl.bridge_rows, st.bridge_rows), routing_remaining), umin(kBridgeBatch, bridge_n));
        if (bl) {
            c.phase([&](uint32_t t) {
                for (uint32_t i = t; i < bl; i += T) m.batch[i] = (uint32_t)m.B[head + i];
            });
            head += bl;
            nb -= bl;
            bridge_n -= bl;
            st.routing_rows += bl;
            st.bridge_rows += bl;
            st.neighbor_multi_get_calls += 1;
            process_rows(bl);
        } else if (nr == 0 && bridge_n != 0) {
            st.termination = kTermBridgeBudget;
            break;
        }
        if (sat_sub(vp, st.vector_payload_requests) == 0) { st.termination = kTermVectorBudget; break; }
        if (n_elig == 0) continue;
        st.simhash_row_requests += n_elig; // restricted_candidate_keys of the eligible ids (all indexed: they came from rows)
        score_list(n_elig);
    }
    s_n_out = s_n;
    return bad;
}

} // namespace walk
} // namespace hvx
