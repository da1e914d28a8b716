// walk_twin.cpp -- TEST INFRASTRUCTURE: the restricted-walk algorithm body of the product (helix-db_amd/csrc/hvx_walk_core.h,
// the exact text the gfx950 kernel is compiled from) compiled for the host, with the workgroup's threads executed one after
// another.  tests/test_walk_twin.py runs it against the oracle's restatement of restricted_filter_aware_search on the
// reference's own fixtures, so the kernel's control flow, budgets, ordering rules and counters are checked on the CPU before
// the kernel ever runs on a GPU.  Never linked into libhelix_vec_gfx950.so; distances come from a caller-supplied callback.
//
// Every phase runs its threads in a different pseudo-random order: a phase whose outcome depends on the order in which the
// threads of a workgroup run (i.e. a data race on the device) produces different results from run to run here.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../helix-db_amd/csrc/hvx_walk_core.h"

using namespace hvx::walk;

typedef float (*dist_fn)(uint32_t row);

template <uint32_t TT, uint32_t WW> struct HostCtx {
    static constexpr uint32_t T = TT, W = WW;
    dist_fn dist;
    uint64_t rng;
    std::vector<uint32_t> order;
    HostCtx(dist_fn d, uint64_t seed) : dist(d), rng(seed * 2654435761u + 12345u), order(TT) {
        for (uint32_t i = 0; i < TT; ++i) order[i] = i;
    }
    uint32_t next() {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        return (uint32_t)(rng >> 33);
    }
    template <class F> void phase(F f) {
        for (uint32_t i = TT - 1; i > 0; --i) std::swap(order[i], order[next() % (i + 1)]);
        for (uint32_t i = 0; i < TT; ++i) f(order[i]);
    }
    uint32_t atomic_or_global(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o | v; return o; }
    uint32_t atomic_add_shared(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
    void atomic_min_shared(uint32_t *p, uint32_t v) { if (v < *p) *p = v; }
    uint32_t atomic_cas_shared(uint32_t *p, uint32_t expected, uint32_t desired) { const uint32_t o = *p; if (o == expected) *p = desired; return o; }
    uint32_t load_seen(const uint32_t *p) { return *p; }
    void scan(uint32_t *a, uint32_t len, uint32_t *total) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < len; ++i) { const uint32_t x = a[i]; a[i] = run; run += x; }
        *total = run;
    }
    void sort64(uint64_t *a, uint32_t L) { std::sort(a, a + L); }
    uint32_t score(const uint32_t *rows, uint32_t n, uint64_t *keys) {
        uint32_t bad = 0;
        for (uint32_t i = 0; i < n; ++i) {
            float d = dist(rows[i]);
            if (!std::isfinite(d) || d < 0.0f) bad = 1; // DistanceScore::try_new (parameters.rs:243-274)
            if (d == 0.0f) d = 0.0f;                   // -0 -> +0
            uint32_t bits;
            memcpy(&bits, &d, 4);
            keys[i] = ((uint64_t)bits << 32) | ((uint64_t)rows[i] << 1);
        }
        return bad;
    }
};

struct twin_args {
    uint32_t n, s0, dim, entry, has_entry, b_cap, variant;
    uint64_t qhash, order_seed;
    const uint32_t *l0;
    const uint64_t *node_hash, *dir_code;
    const uint32_t *dir_row, *allowed, *sample_rows;
    Plan plan;
};

// the 65 537-entry prefix table of the directory (the library builds it on the device next to the sorted directory)
static std::vector<uint32_t> prefix_table(const uint64_t *dir_code, uint32_t n) {
    std::vector<uint32_t> p(65537);
    for (uint32_t v = 0; v <= 65536u; ++v) p[v] = v == 65536u ? n : lower_bound(dir_code, n, (uint64_t)v << 48);
    return p;
}

template <uint32_t TT, uint32_t WW, uint32_t STAGE = kStageCap>
static uint32_t run_variant(const twin_args &a, dist_fn dist, uint64_t *out_keys, uint32_t *out_n, Counters *out_st) {
    static_assert(STAGE >= kStageCap && STAGE >= 2 * WW || WW <= 1024, "staging as the kernel sizes it");
    std::vector<uint64_t> S(kScoredCap, 0), B(std::max<uint32_t>(a.b_cap, 1), 0), G(STAGE, 0), N(kScoredCap, 0), tv(WW, 0);
    // like the kernel, the geometry with W = 1024 keeps N where rows + scan live (dead while rows are scored)
    std::vector<uint32_t> rs(2 * WW + 2, 0);
    uint32_t *rows_p = rs.data() + (reinterpret_cast<uintptr_t>(rs.data()) % 8 ? 1 : 0), *scan_p = rows_p + WW;
    std::vector<uint32_t> Gr(STAGE, 0), E(kScoredCap, 0), tp(WW, 0), batch(kBridgeBatch, 0), ctl(kCtlWords, 0xDEADBEEFu);
    uint64_t *N_p = (WW * 8 == kScoredCap * 8) ? reinterpret_cast<uint64_t *>(rows_p) : N.data();
    std::vector<uint8_t> flag(WW, 0);
    std::vector<uint32_t> seen((a.n + 31) / 32 + 1, 0);
    Mem m{S.data(), B.data(), G.data(), N_p, Gr.data(), E.data(), rows_p, scan_p, flag.data(), tv.data(), tp.data(), batch.data(), ctl.data(), a.b_cap, STAGE};
    const std::vector<uint32_t> prefix = prefix_table(a.dir_code, a.n);
    View v{a.l0, a.s0, a.n, a.dim, a.node_hash, a.dir_code, a.dir_row, prefix.data(), a.entry, a.has_entry, a.allowed, seen.data()};
    HostCtx<TT, WW> c(dist, a.order_seed);
    uint32_t s_n = 0;
    const uint32_t bad = run(c, v, a.plan, a.sample_rows, a.qhash, m, *out_st, s_n);
    memcpy(out_keys, S.data(), (size_t)s_n * 8);
    *out_n = s_n;
    return bad;
}

extern "C" uint32_t walk_twin_run(const twin_args *a, dist_fn dist, uint64_t *out_keys /*[1024]*/, uint32_t *out_n, Counters *out_st) {
    if (a->s0 == 0 || a->plan.vector_payloads > kScoredCap || a->plan.sampled_seeds > kSeedCap || a->plan.directory_seeds > kSeedCap ||
        a->b_cap < a->plan.bridge_rows)
        return 99;
    switch (a->variant) {
    case 0: return a->s0 <= 1024 ? run_variant<256, 1024>(*a, dist, out_keys, out_n, out_st) : 99; // the kernel's geometry
    case 1: return a->s0 <= 256 ? run_variant<64, 256>(*a, dist, out_keys, out_n, out_st) : 99;
    case 2: return a->s0 <= 64 ? run_variant<8, 64>(*a, dist, out_keys, out_n, out_st) : 99;    // tiny: every loop chunks
    case 3: return a->s0 <= 1024 ? run_variant<1, 1024>(*a, dist, out_keys, out_n, out_st) : 99;  // one thread
    case 4: return a->s0 <= 2048 ? run_variant<1024, 2048, 4096>(*a, dist, out_keys, out_n, out_st) : 99; // the kernel's WIDE geometry (small batches)
    default: return 99;
    }
}

extern "C" uint32_t walk_twin_prefix_offset(uint32_t i) { return prefix_offset(i); }
extern "C" uint64_t walk_twin_order_code(uint64_t bits) { return order_code(bits); }
