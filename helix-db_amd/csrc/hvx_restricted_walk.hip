// hvx_restricted_walk.hip -- restricted (prefiltered) vector search under the reference's execution plan, with the
// filter-aware walk on the device.
//
// Reference (crates/db/src/search/vector/restricted.rs): search_restricted_observed_with_beam_percent :528-613,
// restricted_execution_plan_with_beam_percent :426-453, FilteredGraphBudgets :230-259, deterministic_sample_ids :321-342,
// restricted_filter_aware_search :837-1148 (the algorithm body: hvx_walk_core.h, shared with the host twin of the tests).
//
// Mapping onto CDNA4: one 256-thread workgroup per query (4 wavefronts, one per SIMD).  The walk is a chain of dependent
// batches -- pop <= 16 frontier rows / <= 256 bridge rows, read their layer-0 rows, classify every listed id against the
// membership bitmap, score the new members -- so the parallelism is INSIDE a batch: a phase classifies 1 024 ids at once
// (32 neighbour rows of 32 ids; four independent HBM reads in flight per thread), scoring runs 32 rows at a time on the
// 8-lane row groups of hvx_device.h (bit-exact distances), and the two priority queues are sorted arrays in LDS maintained
// with a bitonic network and a rank-merge.  Per query: <= 800 row gathers (<= 4.9 MB at 1536-D) + <= ef_filtered*16
// neighbour rows + one bit probe per listed id -- the algorithmic bytes of SURVEY.md 8(d) "filtered graph".
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <numeric>
#include <vector>

#include "hvx_host.h"
#include "hvx_walk_core.h"

using namespace hvx;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(HVX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

namespace {

// Two geometries of the same algorithm body (round 4).  NARROW: 256 threads classify 1 024 ids per step -- many queries in flight
// (two workgroups per CU).  WIDE: 1 024 threads, 2 048 ids per step (64 neighbour rows), staging for 4 096 bridge keys -- a batch of
// a few dozen queries (the DBpedia benchmark runs 32) leaves most CUs idle, so each query gets a whole CU: every phase has one
// dependent chain of HBM reads per thread instead of four, a 256-row bridge batch is four chunks instead of eight, 128 rows are
// scored per step instead of 32.
constexpr uint32_t kW = 1024;              // neighbour rows of up to 1 024 ids are served (narrow geometry's step)
constexpr uint32_t kWideBatch = 256;       // batches up to this many queries take the wide geometry
constexpr uint32_t kMaxBridgeRows = 9600;  // B in LDS: 75 KiB at most (ef 800 -> ef_filtered 1 200 -> 9 600 bridge rows)

struct WalkArgs {
    DevIndex ix;
    const float *queries;       // [b][dim]
    const uint32_t *qstatus;    // [b]
    const float *qhdr;          // [b]
    const uint64_t *qhash;      // [b] query SimHash
    const uint64_t *node_hash;  // [n]
    const uint64_t *dir_code;   // [n]
    const uint32_t *dir_row;    // [n]
    const uint32_t *dir_prefix; // [65537]
    const uint32_t *allowed;    // membership bitmap(s) over the rows
    uint32_t *seen;             // [b][words], zero
    uint32_t words, allowed_stride; // allowed_stride = 0: one candidate set for the whole batch
    const walk::Plan *plans;    // [b]
    const uint32_t *samples;    // sample rows of the deterministic seeds
    uint32_t sample_stride;     // 0: shared
    uint32_t b_cap;             // capacity of the bridge array in LDS (>= every plan's bridge_rows)
    uint32_t k_stride;          // row length of the output arrays
    uint64_t *out_ids;
    float *out_scores;
    uint32_t *out_counts, *out_status;
    walk::Counters *counters;   // [b]
};

template <uint32_t METRIC, bool FUSED, uint32_t TT, uint32_t WW> struct DevCtx {
    static constexpr uint32_t T = TT, W = WW;
    const DevIndex &ix;
    const float *qv;
    float qhdr;
    uint32_t *wsum; // [W / 64] LDS scratch of the prefix sums

    // a barrier on both sides: the uniform control flow between two phases READS words a completed phase wrote (counters, queue
    // lengths); the barrier in front keeps a fast wavefront's next phase from overwriting such a word before a slow one has read it
    template <class F> __device__ __forceinline__ void phase(F f) {
        __syncthreads();
        f(threadIdx.x);
        __syncthreads();
    }
    __device__ __forceinline__ uint32_t atomic_or_global(uint32_t *p, uint32_t v) { return atomicOr(p, v); }
    __device__ __forceinline__ uint32_t atomic_add_shared(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
    __device__ __forceinline__ void atomic_min_shared(uint32_t *p, uint32_t v) { atomicMin(p, v); }
    __device__ __forceinline__ uint32_t atomic_cas_shared(uint32_t *p, uint32_t expected, uint32_t desired) { return atomicCAS(p, expected, desired); }
    // the bits were OR-ed in at the L2 by other wavefronts of this workgroup: read them there, not from the CU's L1
    __device__ __forceinline__ uint32_t load_seen(const uint32_t *p) {
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // exclusive prefix sum over a[0, len), len <= W: element p*T + t belongs to thread t
    __device__ __forceinline__ void scan(uint32_t *a, uint32_t len, uint32_t *total) {
        constexpr uint32_t P = W / T;
        const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
        __syncthreads();
        uint32_t v[P], inc[P];
#pragma unroll
        for (uint32_t p = 0; p < P; ++p) {
            const uint32_t i = p * T + t;
            v[p] = i < len ? a[i] : 0u;
            uint32_t x = v[p];
#pragma unroll
            for (uint32_t d = 1; d < 64; d <<= 1) {
                const uint32_t u = __shfl_up(x, d, 64);
                if (lane >= d) x += u;
            }
            inc[p] = x;
            if (lane == 63u) wsum[p * (T / 64) + wave] = x;
        }
        __syncthreads();
        uint32_t run = 0, base[P];
#pragma unroll
        for (uint32_t e = 0; e < W / 64; ++e) {
#pragma unroll
            for (uint32_t p = 0; p < P; ++p)
                if (e == p * (T / 64) + wave) base[p] = run;
            run += wsum[e];
        }
#pragma unroll
        for (uint32_t p = 0; p < P; ++p) {
            const uint32_t i = p * T + t;
            if (i < len) a[i] = base[p] + inc[p] - v[p];
        }
        *total = run;
        __syncthreads();
    }

    // bitonic network over L keys in LDS
    __device__ __forceinline__ void sort64(uint64_t *a, uint32_t L) {
        const uint32_t t = threadIdx.x;
        __syncthreads();
        for (uint32_t k = 2; k <= L; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = t; i < (L >> 1); i += T) {
                    const uint32_t lo = ((i & ~(j - 1u)) << 1) | (i & (j - 1u)), hi = lo | j;
                    const bool up = (lo & k) == 0u;
                    const uint64_t x = a[lo], y = a[hi];
                    if ((x > y) == up) {
                        a[lo] = y;
                        a[hi] = x;
                    }
                }
                __syncthreads();
            }
    }

    // restricted_score_keys' distances: 32 rows at a time, one per 8-lane row group, in the host kernel's summation order
    __device__ __forceinline__ uint32_t score(const uint32_t *rows, uint32_t n, uint64_t *keys) {
        const uint32_t t = threadIdx.x, grp = t >> 3;
        const int j = (int)(t & 7u);
        __syncthreads();
        int bad = 0;
        for (uint32_t f = grp; f < n; f += T / 8) {
            const uint32_t node = rows[f];
            float d;
            // (the wide geometry runs 16 wavefronts per workgroup = 128 registers each: four loads in flight per lane instead of eight keep
            // the scoring loop out of scratch -- round 6; 13 spilled registers / 56 B of scratch before, profiles/r05z_kernel_meta.json)
            constexpr int UNR = TT >= 1024 ? 4 : 8;
            if (METRIC != kL1 && FUSED && ix.dtype == HVX_BF16) d = group_distance_bf16<METRIC == kL1 ? kL2 : METRIC, UNR>(ix, qv, qhdr, node, j);
            else d = group_distance<METRIC, FUSED, UNR>(ix, qv, qhdr, node, j);
            if (!score_valid(d)) bad = 1;
            if (j == 0) keys[f] = ((uint64_t)__float_as_uint(d) << 32) | ((uint64_t)node << 1);
        }
        return (uint32_t)__syncthreads_or(bad);
    }
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15u) & ~(size_t)15u; }

size_t walk_lds_bytes(uint32_t ld, uint32_t b_cap, uint32_t W, uint32_t stage_cap) {
    return align16((size_t)ld * 4) + walk::kScoredCap * 8 + align16((size_t)b_cap * 8) + (size_t)stage_cap * 8 + (size_t)stage_cap * 4 +
           walk::kScoredCap * 4 + (size_t)W * 4 + (size_t)W * 4 + W + (size_t)W * 8 + (size_t)W * 4 + walk::kBridgeBatch * 4 + walk::kCtlWords * 4 + (W / 64) * 4;
}

template <uint32_t METRIC, bool FUSED, uint32_t TT, uint32_t WW, uint32_t STAGE>
__global__ __launch_bounds__(TT) void restricted_walk_kernel(WalkArgs a) {
    constexpr uint32_t kT = TT, kW = WW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DevIndex &ix = a.ix;
    const uint32_t q = blockIdx.x, t = threadIdx.x;
    const walk::Plan pl = a.plans[q];
    const uint32_t status_in = a.qstatus ? a.qstatus[q] : 0u;
    if (status_in != 0u || !ix.has_entry) {
        if (t == 0) {
            a.out_counts[q] = 0;
            if (a.out_status) a.out_status[q] = status_in;
            a.counters[q] = walk::Counters{};
        }
        return;
    }
    char *p = smem;
    float *qv = reinterpret_cast<float *>(p); p += align16((size_t)ix.ld * 4);
    walk::Mem m;
    m.S = reinterpret_cast<uint64_t *>(p); p += walk::kScoredCap * 8;
    m.B = reinterpret_cast<uint64_t *>(p); p += align16((size_t)a.b_cap * 8);
    m.G = reinterpret_cast<uint64_t *>(p); p += STAGE * 8;
    m.tv = reinterpret_cast<uint64_t *>(p); p += kW * 8;
    m.Gr = reinterpret_cast<uint32_t *>(p); p += STAGE * 4;
    m.E = reinterpret_cast<uint32_t *>(p); p += walk::kScoredCap * 4;
    m.N = reinterpret_cast<uint64_t *>(p); // the keys of freshly scored rows live where `rows` + `scan` do: those are dead while rows are scored
    static_assert(walk::kScoredCap * 8 <= 2 * kW * 4, "N aliases rows + scan");
    static_assert(STAGE >= walk::kStageCap && STAGE >= 2 * kW, "a chunk's bridge keys fit behind what a flush left");
    m.rows = reinterpret_cast<uint32_t *>(p); p += kW * 4;
    m.scan = reinterpret_cast<uint32_t *>(p); p += kW * 4;
    m.tp = reinterpret_cast<uint32_t *>(p); p += kW * 4;
    m.batch = reinterpret_cast<uint32_t *>(p); p += walk::kBridgeBatch * 4;
    m.ctl = reinterpret_cast<uint32_t *>(p); p += walk::kCtlWords * 4;
    uint32_t *wsum = reinterpret_cast<uint32_t *>(p); p += (kW / 64) * 4;
    m.flag = reinterpret_cast<uint8_t *>(p);
    m.b_cap = a.b_cap;
    m.stage_cap = STAGE;
    for (uint32_t i = t; i < ix.ld; i += kT) qv[i] = i < ix.dim ? a.queries[(size_t)q * ix.dim + i] : 0.f;

    walk::View v;
    v.l0 = ix.l0; v.s0 = ix.s0; v.n = ix.n; v.dim = ix.dim;
    v.node_hash = a.node_hash; v.dir_code = a.dir_code; v.dir_row = a.dir_row; v.dir_prefix = a.dir_prefix;
    v.entry = ix.entry; v.has_entry = ix.has_entry;
    v.allowed = a.allowed + (size_t)q * a.allowed_stride;
    v.seen = a.seen + (size_t)q * a.words;
    DevCtx<METRIC, FUSED, TT, WW> c{ix, qv, a.qhdr ? a.qhdr[q] : 0.f, wsum};
    walk::Counters st;
    uint32_t s_n = 0;
    const uint32_t bad = walk::run(c, v, pl, a.samples + (size_t)q * a.sample_stride, a.qhash[q], m, st, s_n);
    __syncthreads();
    // results = `top` sorted by (score, id), first k (restricted.rs:1130-1147)
    uint32_t out_n = s_n < pl.ef_filtered ? s_n : pl.ef_filtered;
    if (out_n > pl.k) out_n = pl.k;
    if (bad) out_n = 0;
    for (uint32_t i = t; i < out_n; i += kT) {
        const uint64_t key = m.S[i];
        a.out_ids[(size_t)q * a.k_stride + i] = ix.ids[(uint32_t)(key >> 1) & 0x7FFFFFFFu];
        a.out_scores[(size_t)q * a.k_stride + i] = __uint_as_float((uint32_t)(key >> 32));
    }
    if (t == 0) {
        a.out_counts[q] = out_n;
        if (a.out_status) a.out_status[q] = bad ? 8u /*HVX_ERR_INVARIANT*/ : 0u;
        a.counters[q] = st;
    }
}

template <typename K> hipError_t launch_walk_kernel(K kern, const WalkArgs &a, uint32_t b, uint32_t threads, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(b), dim3(threads), lds, s, a);
    return hipGetLastError();
}
template <uint32_t METRIC, bool FUSED> hipError_t launch_walk_t(const WalkArgs &a, uint32_t b, hipStream_t s) {
    // the wide geometry for small batches, where its staging fits next to the plan's bridge queue (160 KiB of LDS per CU)
    const size_t wide_lds = walk_lds_bytes(a.ix.ld, a.b_cap, 2048, 4096);
    if (b <= kWideBatch && wide_lds <= 160 * 1024)
        return launch_walk_kernel(restricted_walk_kernel<METRIC, FUSED, 1024, 2048, 4096>, a, b, 1024u, wide_lds, s);
    const size_t lds = walk_lds_bytes(a.ix.ld, a.b_cap, 1024, 2048);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    return launch_walk_kernel(restricted_walk_kernel<METRIC, FUSED, 256, 1024, 2048>, a, b, 256u, lds, s);
}

hipError_t launch_walk(const WalkArgs &a, uint32_t b, hipStream_t s) {
    const bool fused = kernel_fused(a.ix.fkernel);
    switch (a.ix.metric) {
    case kCosine: return fused ? launch_walk_t<kCosine, true>(a, b, s) : launch_walk_t<kCosine, false>(a, b, s);
    case kL2: return fused ? launch_walk_t<kL2, true>(a, b, s) : launch_walk_t<kL2, false>(a, b, s);
    default: return launch_walk_t<kL1, true>(a, b, s);
    }
}

// membership bitmap of a row list
__global__ void set_bits_kernel(const uint32_t *rows, uint32_t n, uint32_t *bitmap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicOr(&bitmap[rows[i] >> 5], 1u << (rows[i] & 31u));
}
// samples[t] = rows[ranks[t]] (candidate sets whose every id is indexed: rank in the population = rank in the row list)
__global__ void gather_rows_kernel(const uint32_t *rows, const uint32_t *ranks, uint32_t n, uint32_t *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rows[ranks[i]];
}

using HostPlan = hvx::RestrictedPlan;

// restricted_execution_plan_with_beam_percent (restricted.rs:426-453) + FilteredGraphBudgets::with_beam_percent (:230-259)
bool walk_supported(const hvx_index *ix);
// `ix` (nullable): the handle the plan will run on.  An AUTO plan without explicit budgets that the walk of this build cannot
// run -- budgets beyond the LDS-resident queues (k >= 301 or ef > 800 under the reference's own budget rule), rows that are not
// f32, neighbour rows beyond the walk's width -- is answered by the EXACT gathered scan instead (round 4): a superset
// of the walk's answer in recall, reported as strategy EXACT.  Explicit budgets / strategy FILTERED outside the limits stay errors.
int make_plan(const hvx_restricted_params &rp, uint64_t candidates, uint32_t dim, HostPlan *out, const hvx_index *ix = nullptr) {
    HostPlan hp{};
    const uint64_t kk = std::min<uint64_t>(rp.k, candidates);
    hp.p.k = (uint32_t)kk;
    hp.p.directory_enabled = rp.directory_enabled ? 1u : 0u;
    const bool planned = rp.strategy == HVX_RESTRICTED_AUTO || rp.strategy == HVX_RESTRICTED_REFERENCE_PLAN;
    // the reference's line (restricted.rs:426-453): 256 ids / 4 MiB of f32 payload -- drawn where a CPU that pays a KV get per row stops
    // winning with a scan
    bool exact_ok = candidates <= 256 && candidates * (uint64_t)dim * 4ull <= 4ull * 1024 * 1024;
    // the DEVICE's line (round 6, HVX_RESTRICTED_AUTO): an exact gathered scan costs candidates x row bytes / ~4 TB/s + one launch, the
    // filter-aware walk 0.3 - 0.9 ms per 32-query batch and ~3.4 us per query with the chip full (profiles/r05w_bench_full.json config3:
    // exact 8 - 15 us per query at 100 - 100 000 candidates, walk 10 - 46 us, with recall 0.19 / 0.85 at 1 000 / 10 000 candidates on the
    // clustered stand-in) -- the scan wins until the candidate rows take about a GiB, and it is exact
    if (rp.strategy == HVX_RESTRICTED_AUTO && ix) {
        const uint64_t row_bytes = (uint64_t)dim * (ix->dev.dtype == HVX_F32 ? 4u : (ix->dev.dtype == HVX_BF16 ? 2u : 1u));
        const uint64_t limit = (uint64_t)(ix->opt[HVX_OPT_RESTRICTED_EXACT_MIB] ? ix->opt[HVX_OPT_RESTRICTED_EXACT_MIB] : 1024u) << 20;
        exact_ok = exact_ok || candidates * row_bytes <= limit;
    }
    if (rp.strategy == HVX_RESTRICTED_EXACT || (planned && exact_ok && !rp.explicit_budgets)) {
        hp.strategy = HVX_RESTRICTED_EXACT;
        *out = hp;
        return HVX_OK;
    }
    hp.strategy = HVX_RESTRICTED_FILTERED;
    if (rp.explicit_budgets) {
        hp.p.ef_filtered = rp.ef_filtered;
        hp.p.routing_rows = rp.routing_rows;
        hp.p.bridge_rows = rp.bridge_rows;
        hp.p.vector_payloads = rp.vector_payloads;
        hp.p.sampled_seeds = rp.sampled_seeds;
        hp.p.directory_seeds = rp.directory_seeds;
    } else {
        const uint64_t pct = rp.beam_percent ? rp.beam_percent : 150u;
        const uint64_t scaled = (uint64_t)rp.ef * pct / 100ull;
        const uint64_t eff = std::min<uint64_t>(std::max<uint64_t>(scaled, kk * 4), candidates);
        hp.p.ef_filtered = (uint32_t)eff;
        hp.p.routing_rows = (uint32_t)std::min<uint64_t>(eff * 16, 0xFFFFFFFFull);
        hp.p.bridge_rows = (uint32_t)std::min<uint64_t>(eff * 8, 0xFFFFFFFFull);
        hp.p.vector_payloads = (uint32_t)std::min<uint64_t>(800, candidates);
        hp.p.sampled_seeds = (uint32_t)std::min<uint64_t>(64, candidates);
        hp.p.directory_seeds = (uint32_t)std::min<uint64_t>(256, candidates);
    }
    const bool beyond = hp.p.vector_payloads > walk::kScoredCap || hp.p.sampled_seeds > walk::kSeedCap || hp.p.directory_seeds > walk::kSeedCap ||
                        hp.p.bridge_rows > kMaxBridgeRows;
    if (planned && !rp.explicit_budgets && (beyond || (ix && !walk_supported(ix)))) {
        HostPlan ex{};
        ex.p.k = hp.p.k;
        ex.p.directory_enabled = hp.p.directory_enabled;
        ex.strategy = HVX_RESTRICTED_EXACT;
        *out = ex;
        return HVX_OK;
    }
    if (beyond)
        return fail(HVX_ERR_UNSUPPORTED, "filtered walk budgets outside this build: vector_payloads %u (<= %u), sampled / directory seeds %u / %u "
                    "(<= %u), bridge_rows %u (<= %u)", hp.p.vector_payloads, walk::kScoredCap, hp.p.sampled_seeds, hp.p.directory_seeds,
                    walk::kSeedCap, hp.p.bridge_rows, kMaxBridgeRows);
    hp.p.n_sample = (uint32_t)std::min<uint64_t>(hp.p.sampled_seeds, candidates);
    *out = hp;
    return HVX_OK;
}

// NonEmptyCandidateSet::deterministic_sample_ids (restricted.rs:321-342) as ranks into the ascending candidate list
void sample_ranks(uint64_t candidates, uint32_t count, std::vector<uint64_t> &out) {
    out.resize(count);
    for (uint64_t t = 0; t < count; ++t) {
        if (count == candidates) out[t] = t;
        else if (count == 1) out[t] = 0;
        else out[t] = (uint64_t)(((unsigned __int128)t * (unsigned __int128)(candidates - 1)) / (unsigned __int128)(count - 1));
    }
}

// the SimHash directory of this image, built once from the attached SimHash rows -- on the device (round 4; rounds 2-3 sorted the
// 1M order codes with std::sort on the host): order codes, a stable radix sort of (code, row) pairs (rows ascend inside equal
// codes = the (order_code, node_id) key order of storage.rs:1942-2010), and the 65 537-entry prefix table the window bounds come from
// `live` = the ascending list of live rows of an image with deleted nodes (a delete removes the node's directory row:
// mutation.rs:1718-1722), NULL = every row
__global__ void dir_codes_kernel(const uint64_t *node_hash, uint32_t n, const uint32_t *live, uint64_t *codes, uint32_t *rows) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint32_t r = live ? live[i] : i; codes[i] = walk::order_code(node_hash[r]); rows[i] = r; }
}
__global__ void dir_prefix_kernel(const uint64_t *sorted_codes, uint32_t n, uint32_t *prefix) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > 65536u) return;
    prefix[p] = p == 65536u ? n : walk::lower_bound(sorted_codes, n, (uint64_t)p << 48);
}
// A plain stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass, 256 elements per workgroup: a one-off per image
// (1M pairs: ~2 ms), written out here because the library links nothing that reads the environment (hipCUB's dispatch does).
constexpr uint32_t kRsBlock = 256;
__global__ __launch_bounds__(256) void rs_hist_kernel(const uint64_t *keys, uint32_t n, uint32_t shift, uint32_t *hist, uint32_t nblocks) {
    __shared__ uint32_t h[256];
    const uint32_t t = threadIdx.x, i = blockIdx.x * kRsBlock + t;
    h[t] = 0;
    __syncthreads();
    if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    __syncthreads();
    hist[(size_t)t * nblocks + blockIdx.x] = h[t]; // digit-major: a scan over the whole array yields every (digit, block) start
}
__global__ __launch_bounds__(1024) void rs_scan_kernel(uint32_t *a, uint32_t len) { // exclusive scan, one workgroup
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, per = (len + 1023u) / 1024u, lo = t * per, hi = lo + per < len ? lo + per : len;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += a[i];
    part[t] = sum;
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < 1024u; ++i) { const uint32_t x = part[i]; part[i] = run; run += x; }
    }
    __syncthreads();
    uint32_t run = part[t];
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t x = a[i]; a[i] = run; run += x; }
}
__global__ __launch_bounds__(256) void rs_scatter_kernel(const uint64_t *keys, const uint32_t *vals, uint32_t n, uint32_t shift, const uint32_t *starts,
                                                         uint32_t nblocks, uint64_t *out_keys, uint32_t *out_vals) {
    __shared__ uint32_t dg[256];
    const uint32_t t = threadIdx.x, i = blockIdx.x * kRsBlock + t;
    const uint64_t key = i < n ? keys[i] : 0ull;
    const uint32_t d = i < n ? (uint32_t)(key >> shift) & 255u : 0xFFFFFFFFu;
    dg[t] = d;
    __syncthreads();
    if (i >= n) return;
    uint32_t rank = 0; // stable: elements of this workgroup with the same digit that come earlier
    for (uint32_t u = 0; u < t; ++u) rank += dg[u] == d ? 1u : 0u;
    const uint32_t pos = starts[(size_t)d * nblocks + blockIdx.x] + rank;
    out_keys[pos] = key;
    out_vals[pos] = vals[i];
}

int ensure_directory(hvx_index *ix) {
    hvx_image_shared &sh = *ix->shared;
    const uint32_t *live = nullptr;
    if (ix->n_dead) { // the directory of an image with deleted nodes holds its live rows only
        int rc = ix->ensure_live();
        if (rc) return rc;
        live = ix->f_live;
    }
    std::lock_guard<std::mutex> lock(sh.mu);
    if (sh.dir_code && sh.dir_for == ix->d_node_hash && sh.dir_rows == ix->live_rows() && sh.dir_dead == ix->dead_p)
        return HVX_OK; // (an image that has grown, or lost nodes, gets a new directory)
    const uint32_t n = ix->live_rows(), n1 = std::max<uint32_t>(n, 1);
    hipStream_t s = ix->stream;
    for (void **p : {(void **)&sh.dir_code, (void **)&sh.dir_row, (void **)&sh.dir_prefix})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    sh.device = ix->device;
    HIP_TRY(hipMalloc((void **)&sh.dir_code, (size_t)n1 * 8));
    HIP_TRY(hipMalloc((void **)&sh.dir_row, (size_t)n1 * 4));
    HIP_TRY(hipMalloc((void **)&sh.dir_prefix, 65537u * 4));
    uint64_t *codes = nullptr;
    uint32_t *rows = nullptr, *hist = nullptr;
    auto release = [&]() {
        for (void *p : {(void *)codes, (void *)rows, (void *)hist})
            if (p) (void)hipFree(p);
    };
    auto bail = [&](int rc) { (void)hipStreamSynchronize(s); release(); return rc; };
    const uint32_t nblocks = (n1 + kRsBlock - 1u) / kRsBlock;
    if (hipMalloc((void **)&codes, (size_t)n1 * 8) != hipSuccess || hipMalloc((void **)&rows, (size_t)n1 * 4) != hipSuccess ||
        hipMalloc((void **)&hist, (size_t)256 * nblocks * 4) != hipSuccess)
        return bail(fail(HVX_ERR_DEVICE, "hipMalloc of the directory scratch failed"));
    if (n) {
        hipLaunchKernelGGL(dir_codes_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, ix->d_node_hash, n, live, sh.dir_code, sh.dir_row);
        uint64_t *ka = sh.dir_code, *kb = codes;
        uint32_t *va = sh.dir_row, *vb = rows;
        for (uint32_t pass = 0; pass < 8; ++pass) { // eight passes: the sorted pairs end where they started (dir_code / dir_row)
            hipLaunchKernelGGL(rs_hist_kernel, dim3(nblocks), dim3(256), 0, s, ka, n, pass * 8u, hist, nblocks);
            hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, s, hist, 256u * nblocks);
            hipLaunchKernelGGL(rs_scatter_kernel, dim3(nblocks), dim3(256), 0, s, ka, va, n, pass * 8u, hist, nblocks, kb, vb);
            std::swap(ka, kb);
            std::swap(va, vb);
        }
    }
    hipLaunchKernelGGL(dir_prefix_kernel, dim3((65537u + 255u) / 256u), dim3(256), 0, s, sh.dir_code, n, sh.dir_prefix);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return bail(fail(HVX_ERR_DEVICE, "building the SimHash directory failed: %s", hipGetErrorString(hipGetLastError())));
    release();
    sh.dir_for = ix->d_node_hash;
    sh.dir_rows = n;
    sh.dir_dead = ix->dead_p;
    return HVX_OK;
}

int grow(hvx_index *ix, void **p, size_t *cap, size_t bytes) {
    if (bytes <= *cap && *p) return HVX_OK;
    int rc = ix->regrow(p, bytes);
    if (rc) return rc;
    *cap = bytes;
    return HVX_OK;
}

bool walk_supported(const hvx_index *ix) { return (ix->dev.dtype == HVX_F32 || ix->dev.dtype == HVX_BF16) && ix->dev.s0 <= kW && ix->dev.s0 != 0; }

void widen(const walk::Counters &c, uint32_t ef_filtered, hvx_restricted_stats *o) {
    o->strategy = HVX_RESTRICTED_FILTERED;
    o->termination = c.termination;
    o->ef_filtered = ef_filtered;
    o->directory_scan_calls = c.directory_scan_calls;
    o->directory_rows = c.directory_rows;
    o->directory_decoded_bytes = c.directory_decoded_bytes;
    o->directory_hits = c.directory_hits;
    o->simhash_row_requests = c.simhash_row_requests;
    o->companion_row_requests = c.companion_row_requests;
    o->routing_rows = c.routing_rows;
    o->bridge_rows = c.bridge_rows;
    o->bridge_frontier_pushes = c.bridge_frontier_pushes;
    o->neighbor_multi_get_calls = c.neighbor_multi_get_calls;
    o->vector_payload_requests = c.vector_payload_requests;
    o->vector_bytes = c.vector_bytes;
    o->distance_computations = c.distance_computations;
}

// One candidate set shared by `b` host-resident queries, walk strategy.  d_rows / n_rows: the indexed candidates as rows on
// the device (ascending); d_samples: the deterministic seeds as rows (kSentinel = not indexed), n_sample of them.
int walk_shared_set(hvx_index *ix, const float *queries, uint32_t b, uint32_t k_stride, const HostPlan &hp, const uint32_t *d_rows,
                    uint32_t n_rows, const uint32_t *d_samples, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                    uint32_t *out_status, hvx_restricted_stats *rstats, hvx_stats *stats) {
    int rc;
    if (!ix->has_simhash)
        return fail(HVX_ERR_INVARIANT, "the filtered restricted search needs the index's SimHash rows (hvx_index_set_simhash): "
                    "missing SimHash companion rows are index corruption in the reference");
    if (!walk_supported(ix)) return fail(HVX_ERR_UNSUPPORTED, "the filtered restricted walk serves f32 / bf16 rows with neighbour rows <= %u ids", kW);
    if ((rc = ensure_directory(ix))) return rc;
    const uint32_t words = (ix->dev.n + 31u) / 32u + 1u, mb = ix->max_batch;
    if ((rc = grow(ix, (void **)&ix->w_allowed, &ix->cap_w_allowed, (size_t)words * 4))) return rc;
    if ((rc = grow(ix, (void **)&ix->w_seen, &ix->cap_w_seen, (size_t)std::min(b, mb) * words * 4))) return rc;
    if ((rc = grow(ix, (void **)&ix->w_plans, &ix->cap_w_q, (size_t)mb * (sizeof(walk::Plan) + sizeof(walk::Counters))))) return rc;
    ix->w_counters = (char *)ix->w_plans + (size_t)mb * sizeof(walk::Plan);
    HIP_TRY(hipMemsetAsync(ix->w_allowed, 0, (size_t)words * 4, ix->stream));
    if (n_rows) hipLaunchKernelGGL(set_bits_kernel, dim3((n_rows + 255u) / 256u), dim3(256), 0, ix->stream, d_rows, n_rows, ix->w_allowed);
    HIP_TRY(hipGetLastError());
    std::vector<walk::Plan> plans(std::min(b, mb), hp.p);
    HIP_TRY(hipMemcpyAsync(ix->w_plans, plans.data(), plans.size() * sizeof(walk::Plan), hipMemcpyHostToDevice, ix->stream));
    std::vector<uint32_t> status(b, 0);
    std::vector<walk::Counters> cnt(b);
    for (uint32_t c0 = 0; c0 < b; c0 += mb) {
        const uint32_t cb = std::min(mb, b - c0);
        if ((rc = ix->stage(cb, k_stride))) return rc;
        if ((rc = ix->stage_in(queries + (size_t)c0 * ix->dev.dim, cb))) return rc;
        HIP_TRY(launch_validate_queries(ix->dev, ix->s_queries, cb, ix->limit, ix->d_qstatus, ix->d_qhdr, ix->stream));
        HIP_TRY(launch_simhash_rows(ix->d_planes_t, ix->s_queries, ix->dev.dim, ix->dev.dim, cb, ix->d_qhash, ix->stream));
        HIP_TRY(hipMemsetAsync(ix->w_seen, 0, (size_t)cb * words * 4, ix->stream));
        WalkArgs a;
        a.ix = ix->dev;
        a.queries = ix->s_queries;
        a.qstatus = ix->d_qstatus;
        a.qhdr = ix->d_qhdr;
        a.qhash = ix->d_qhash;
        a.node_hash = ix->d_node_hash;
        a.dir_code = ix->shared->dir_code;
        a.dir_row = ix->shared->dir_row;
        a.dir_prefix = ix->shared->dir_prefix;
        a.allowed = ix->w_allowed;
        a.seen = ix->w_seen;
        a.words = words;
        a.allowed_stride = 0;
        a.plans = (const walk::Plan *)ix->w_plans;
        a.samples = d_samples;
        a.sample_stride = 0;
        a.b_cap = hp.p.bridge_rows;
        a.k_stride = k_stride;
        a.out_ids = ix->s_ids;
        a.out_scores = ix->s_scores;
        a.out_counts = ix->s_counts;
        a.out_status = ix->s_status;
        a.counters = (walk::Counters *)ix->w_counters;
        if (stats) HIP_TRY(hipEventRecord(ix->ev0, ix->stream));
        HIP_TRY(launch_walk(a, cb, ix->stream));
        if (stats) HIP_TRY(hipEventRecord(ix->ev1, ix->stream));
        if ((rc = ix->stage_out(cb, k_stride))) return rc;
        HIP_TRY(hipMemcpyAsync(cnt.data() + c0, ix->w_counters, (size_t)cb * sizeof(walk::Counters), hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream));
        ix->deliver(cb, k_stride, out_ids + (size_t)c0 * k_stride, out_scores + (size_t)c0 * k_stride, out_counts + c0, status.data() + c0);
        if (stats) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
            stats->device_ms += ms;
            stats->queries += cb;
            for (uint32_t i = 0; i < cb; ++i) {
                stats->distance_computations += cnt[c0 + i].distance_computations;
                stats->vectors_loaded += cnt[c0 + i].vector_payload_requests;
                stats->expansion_steps += cnt[c0 + i].routing_rows;
            }
        }
    }
    for (uint32_t q = 0; q < b; ++q) {
        if (rstats) {
            memset(&rstats[q], 0, sizeof(hvx_restricted_stats));
            if (!status[q] && ix->dev.has_entry) widen(cnt[q], hp.p.ef_filtered, &rstats[q]);
        }
        if (status[q]) {
            out_counts[q] = 0;
            if (!out_status) return fail((int)status[q], "query %u rejected with status %u", q, status[q]);
            out_status[q] = status[q];
        }
    }
    return HVX_OK;
}


// ---- the one-launch exact scan (hvx_restricted_exact.hip) behind the host-pointer entry points ----
// Which exact scan serves a SHARED candidate set: the reference-order kernel scores every row against every query on the vector ALUs and
// is a chain of short dependent passes per workgroup; the matrix-core pipeline streams the rows once and pays ~40 us for its exact tail.
// Measured at dim 1536 (profiles/r06b_restricted_direct.log): one launch wins at 100 and 1 000 candidates for 1 - 32 queries (24 - 60 us
// against 31 - 199) and at 10 000 for one query (59 against 86); it loses from 10 000 x 8 on (119 against 87).  The line: b x rows x dim
// <= 2^26 -- and up to 2^31 where the other choice would be the distance-matrix VALU kernels (dimensions the contraction does not serve).
bool use_direct(const hvx_index *ix, uint32_t b, uint32_t k, uint32_t n_rows) {
    if (ix->opt[HVX_OPT_RESTRICTED_DIRECT] == 1u || n_rows == 0 || !restricted_direct_supported(ix, k)) return false;
    if (ix->opt[HVX_OPT_RESTRICTED_DIRECT] == 2u) return true;
    const uint64_t work = (uint64_t)b * n_rows * ix->dev.dim;
    return work <= (1ull << 26) || (work <= (1ull << 31) && !flat_scan_on_matrix_cores(ix, b, k, n_rows));
}

// pinned mirror layout of one chunk: queries | ids | scores | counts | status (the kernels write the four outputs THERE: mapped host rows)
struct PinView { float *q; uint64_t *ids; float *sc; uint32_t *cnt, *st; unsigned char *extra; };
int pin_view(hvx_index *ix, uint32_t cb, uint32_t k, PinView *v, size_t extra_bytes = 0) {
    const size_t qb = (((size_t)cb * ix->dev.dim * 4) + 63u) & ~(size_t)63u, n = (size_t)cb * k;
    const size_t fixed = (qb + n * 12 + (size_t)cb * 8 + 63u) & ~(size_t)63u;
    int rc = ix->pin(fixed + extra_bytes);
    if (rc) return rc;
    v->extra = ix->h_pin + fixed;
    v->q = reinterpret_cast<float *>(ix->h_pin);
    v->ids = reinterpret_cast<uint64_t *>(ix->h_pin + qb);
    v->sc = reinterpret_cast<float *>(ix->h_pin + qb + n * 8);
    v->cnt = reinterpret_cast<uint32_t *>(ix->h_pin + qb + n * 12);
    v->st = v->cnt + cb;
    return HVX_OK;
}

// queries -> pinned mirror -> (copy kernel) device rows, validation; afterwards s_queries / d_qstatus / d_qhdr describe the chunk
int stage_and_validate(hvx_index *ix, const float *queries, uint32_t cb, const PinView &v) {
    memcpy(v.q, queries, (size_t)cb * ix->dev.dim * 4);
    HIP_TRY(launch_stage_queries(v.q, ix->s_queries, ix->dev.dim, cb, ix->stream));
    HIP_TRY(launch_validate_queries(ix->dev, ix->s_queries, cb, ix->limit, ix->d_qstatus, ix->d_qhdr, ix->stream));
    return HVX_OK;
}

int direct_shared_host(hvx_index *ix, const float *queries, uint32_t b, uint32_t k, const uint32_t *d_rows, uint32_t n_rows, uint64_t *out_ids,
                       float *out_scores, uint32_t *out_counts, uint32_t *out_status, hvx_stats *stats) {
    int rc;
    const uint32_t mb = ix->max_batch;
    for (uint32_t c0 = 0; c0 < b; c0 += mb) {
        const uint32_t cb = std::min(mb, b - c0);
        if ((rc = ix->stage(cb, k))) return rc;
        PinView v;
        if ((rc = pin_view(ix, cb, k, &v))) return rc;
        if ((rc = stage_and_validate(ix, queries + (size_t)c0 * ix->dev.dim, cb, v))) return rc;
        if (stats) HIP_TRY(hipEventRecord(ix->ev0, ix->stream));
        if ((rc = restricted_direct_enqueue(ix, ix->s_queries, cb, k, k, d_rows, n_rows, nullptr, nullptr, 0, v.ids, v.sc, v.cnt, v.st))) return rc;
        if (stats) HIP_TRY(hipEventRecord(ix->ev1, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream));
        memcpy(out_ids + (size_t)c0 * k, v.ids, (size_t)cb * k * 8);
        memcpy(out_scores + (size_t)c0 * k, v.sc, (size_t)cb * k * 4);
        memcpy(out_counts + c0, v.cnt, (size_t)cb * 4);
        memcpy(out_status + c0, v.st, (size_t)cb * 4);
        if (stats) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
            stats->queries += cb;
            stats->vectors_loaded += (uint64_t)cb * n_rows;
            stats->distance_computations += (uint64_t)cb * n_rows;
            stats->device_ms += ms;
        }
    }
    return HVX_OK;
}

// Every query with ITS OWN candidate list (allowed_offsets), all answered by the exact strategy: ONE launch per chunk of max_batch queries.
// Returns -1 when a list needs the host's attention first (more than 1 000 000 entries: duplicates decide whether the set is too big),
// after which the caller serves the batch list by list.
int direct_per_query_host(hvx_index *ix, const float *queries, uint32_t b, const hvx_restricted_params &rp, const uint64_t *allowed_ids,
                          const uint64_t *allowed_offsets, uint64_t *out_ids, float *out_scores, uint32_t *out_counts, uint32_t *out_status,
                          hvx_restricted_stats *rstats, hvx_stats *stats) {
    int rc;
    const uint32_t mb = ix->max_batch, k = rp.k;
    std::vector<uint64_t> off;
    for (uint32_t c0 = 0; c0 < b; c0 += mb) {
        const uint32_t cb = std::min(mb, b - c0);
        const uint64_t a0 = allowed_offsets[c0], a1 = allowed_offsets[c0 + cb];
        off.assign((size_t)cb + 1, 0);
        uint64_t max_set = 0;
        for (uint32_t q = 0; q <= cb; ++q) off[q] = allowed_offsets[c0 + q] - a0;
        for (uint32_t q = 0; q < cb; ++q) max_set = std::max<uint64_t>(max_set, off[q + 1] - off[q]);
        if ((rc = ix->stage(cb, k))) return rc;
        const uint64_t n_ids = a1 - a0;
        // id lists up to 1 MiB ride in the pinned mirror behind the outputs and are read by the kernel IN PLACE (every id once, over PCIe): a
        // copy from the caller's pageable memory is a synchronous staged DMA of ~40 us; longer lists take it
        const bool ids_in_pin = n_ids * 8 <= (1u << 20);
        PinView v;
        if ((rc = pin_view(ix, cb, k, &v, ids_in_pin ? (size_t)n_ids * 8 + ((size_t)cb + 1) * 8 : 0))) return rc;
        const uint64_t *d_ids_src = nullptr, *d_off_src = nullptr;
        if (ids_in_pin) {
            uint64_t *pids = reinterpret_cast<uint64_t *>(v.extra), *poff = pids + n_ids;
            if (n_ids) memcpy(pids, allowed_ids + a0, (size_t)n_ids * 8);
            memcpy(poff, off.data(), ((size_t)cb + 1) * 8);
            d_ids_src = pids;
            d_off_src = poff;
        } else {
            if (n_ids > ix->cap_x_ids) {
                if ((rc = ix->regrow((void **)&ix->x_ids, std::max<size_t>(n_ids, 1) * 8))) return rc;
                ix->cap_x_ids = n_ids;
            }
            if ((size_t)cb + 1 > ix->cap_x_off) {
                if ((rc = ix->regrow((void **)&ix->x_off, ((size_t)mb + 1) * 8))) return rc;
                ix->cap_x_off = (size_t)mb + 1;
            }
            HIP_TRY(hipMemcpyAsync(ix->x_ids, allowed_ids + a0, n_ids * 8, hipMemcpyHostToDevice, ix->stream)); // (pageable source: staged, returns when the source is consumed)
            HIP_TRY(hipMemcpyAsync(ix->x_off, off.data(), ((size_t)cb + 1) * 8, hipMemcpyHostToDevice, ix->stream));
            d_ids_src = ix->x_ids;
            d_off_src = ix->x_off;
        }
        if ((rc = stage_and_validate(ix, queries + (size_t)c0 * ix->dev.dim, cb, v))) return rc;
        if (stats) HIP_TRY(hipEventRecord(ix->ev0, ix->stream));
        if ((rc = restricted_direct_enqueue(ix, ix->s_queries, cb, k, k, nullptr, 0, d_ids_src, d_off_src, (uint32_t)max_set, v.ids, v.sc, v.cnt, v.st))) return rc;
        if (stats) HIP_TRY(hipEventRecord(ix->ev1, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream)); // (`off` is consumed as well)
        for (uint32_t q = 0; q < cb; ++q) {
            const uint32_t g = c0 + q;
            out_counts[g] = 0;
            if (out_status) out_status[g] = HVX_OK;
            if (rstats) memset(&rstats[g], 0, sizeof(hvx_restricted_stats));
            if (off[q + 1] == off[q]) continue; // Empty: no results, before any validation (restricted.rs:539-541)
            if (v.st[q]) {
                if (!out_status) return fail((int)v.st[q], "query %u rejected with status %u", g, v.st[q]);
                out_status[g] = v.st[q];
                continue;
            }
            if (rstats && ix->dev.has_entry) rstats[g].strategy = HVX_RESTRICTED_EXACT;
            out_counts[g] = v.cnt[q];
            memcpy(out_ids + (size_t)g * k, v.ids + (size_t)q * k, (size_t)v.cnt[q] * 8);
            memcpy(out_scores + (size_t)g * k, v.sc + (size_t)q * k, (size_t)v.cnt[q] * 4);
        }
        if (stats) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
            stats->queries += cb;
            stats->vectors_loaded += n_ids;
            stats->distance_computations += n_ids;
            stats->device_ms += ms;
        }
    }
    return HVX_OK;
}

// exact strategy for one shared candidate set (rows already on the device): the gathered exact scan of hvx_flat*.hip
int exact_shared_set(hvx_index *ix, const float *queries, uint32_t b, uint32_t k_stride, uint32_t kk, const uint32_t *d_rows, uint32_t n_rows,
                     uint64_t *out_ids, float *out_scores, uint32_t *out_counts, uint32_t *out_status, hvx_restricted_stats *rstats,
                     hvx_stats *stats) {
    std::vector<uint64_t> t_ids((size_t)b * kk);
    std::vector<float> t_sc((size_t)b * kk);
    std::vector<uint32_t> t_cnt(b), t_st(b);
    int rc;
    if (use_direct(ix, b, kk, n_rows)) rc = direct_shared_host(ix, queries, b, kk, d_rows, n_rows, t_ids.data(), t_sc.data(), t_cnt.data(), t_st.data(), stats);
    else rc = flat_scan_host(ix, queries, b, kk, d_rows, n_rows, t_ids.data(), t_sc.data(), t_cnt.data(), t_st.data(), stats);
    if (rc) return rc;
    for (uint32_t q = 0; q < b; ++q) {
        if (rstats) {
            memset(&rstats[q], 0, sizeof(hvx_restricted_stats));
            if (!t_st[q] && ix->dev.has_entry) {
                rstats[q].strategy = HVX_RESTRICTED_EXACT;
            }
        }
        if (t_st[q]) {
            if (!out_status) return fail((int)t_st[q], "query %u rejected with status %u", q, t_st[q]);
            out_status[q] = t_st[q];
            continue;
        }
        out_counts[q] = t_cnt[q];
        memcpy(out_ids + (size_t)q * k_stride, t_ids.data() + (size_t)q * kk, (size_t)t_cnt[q] * 8);
        memcpy(out_scores + (size_t)q * k_stride, t_sc.data() + (size_t)q * kk, (size_t)t_cnt[q] * 4);
    }
    return HVX_OK;
}

// search_restricted_observed_with_beam_percent (restricted.rs:528-613) for one candidate list shared by b queries
int restricted_set(hvx_index *ix, const float *queries, uint32_t b, const hvx_restricted_params &rp, const uint64_t *allowed,
                   uint64_t n_allowed, uint64_t *out_ids, float *out_scores, uint32_t *out_counts, uint32_t *out_status,
                   hvx_restricted_stats *rstats, hvx_stats *stats) {
    // ONE query with its candidate ids (the operator's call, storage.rs:140-163) under a plan that is exact for a list this long: the
    // one-launch scan maps, dedupes and scans the ids on the device -- no host-side sort / unique / id -> row pass (round 6: those and two
    // staged copies were ~90 us of a 145-us call at 10 000 ids)
    if (b == 1 && n_allowed != 0 && n_allowed <= 1000000ull && rp.k <= 64u && !rp.explicit_budgets && ix->opt[HVX_OPT_RESTRICTED_DIRECT] != 1u &&
        restricted_direct_supported(ix, rp.k) && (n_allowed * (uint64_t)ix->dev.dim <= (1ull << 28) || ix->opt[HVX_OPT_RESTRICTED_DIRECT] == 2u)) {
        HostPlan hp1;
        if (make_plan(rp, n_allowed, ix->dev.dim, &hp1, ix) == HVX_OK && hp1.strategy == HVX_RESTRICTED_EXACT) {
            const uint64_t offs[2] = {0, n_allowed};
            return direct_per_query_host(ix, queries, 1, rp, allowed, offs, out_ids, out_scores, out_counts, out_status, rstats, stats);
        }
    }
    // RestrictedVectorCandidates::from_ids (restricted.rs:356-371): dedupe, cap 1,000,000
    std::vector<uint64_t> ids(allowed, allowed + n_allowed);
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    if (ids.size() > 1000000) return fail(HVX_ERR_CANDIDATE_LIMIT, "restricted vector search accepts at most 1000000 unique candidates");
    for (uint32_t q = 0; q < b; ++q) {
        out_counts[q] = 0;
        if (out_status) out_status[q] = HVX_OK;
        if (rstats) memset(&rstats[q], 0, sizeof(hvx_restricted_stats));
    }
    if (ids.empty()) return HVX_OK; // Empty: no results, before any validation (restricted.rs:539-541)
    const uint32_t kk = (uint32_t)std::min<uint64_t>(rp.k, ids.size());
    if (kk == 0) return fail(HVX_ERR_K_RANGE, "result count must be non-zero");
    if (kk > 800) return fail(HVX_ERR_K_RANGE, "restricted vector search result count %u is above the maximum 800", kk);
    HostPlan hp;
    int rc = make_plan(rp, ids.size(), ix->dev.dim, &hp, ix);
    if (rc) return rc;
    // ids that are not indexed are omitted from the scan (restricted.rs:615-659) but count as candidates for the plan
    std::vector<uint32_t> subset;
    subset.reserve(ids.size());
    for (uint64_t id : ids) {
        const uint32_t x = ix->find(id);
        if (x != kSentinel) subset.push_back(x);
    }
    if (subset.size() > ix->cap_subset) {
        if ((rc = ix->regrow((void **)&ix->f_subset, subset.size() * 4))) return rc;
        ix->cap_subset = subset.size();
    }
    if (!subset.empty())
        HIP_TRY(hipMemcpyAsync(ix->f_subset, subset.data(), subset.size() * 4, hipMemcpyHostToDevice, ix->stream));
    if (hp.strategy == HVX_RESTRICTED_EXACT) {
        HIP_TRY(hipStreamSynchronize(ix->stream)); // `subset` lives on this frame
        return exact_shared_set(ix, queries, b, rp.k, kk, ix->f_subset, (uint32_t)subset.size(), out_ids, out_scores, out_counts, out_status,
                                rstats, stats);
    }
    std::vector<uint64_t> ranks;
    sample_ranks(ids.size(), hp.p.n_sample, ranks);
    std::vector<uint32_t> samples(std::max<size_t>(ranks.size(), 1), kSentinel);
    for (size_t t = 0; t < ranks.size(); ++t) samples[t] = ix->find(ids[ranks[t]]);
    if ((rc = grow(ix, (void **)&ix->w_samples, &ix->cap_w_samples, samples.size() * 4))) return rc;
    HIP_TRY(hipMemcpyAsync(ix->w_samples, samples.data(), samples.size() * 4, hipMemcpyHostToDevice, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream)); // `subset` / `samples` live on this frame
    return walk_shared_set(ix, queries, b, rp.k, hp, ix->f_subset, (uint32_t)subset.size(), ix->w_samples, out_ids, out_scores, out_counts,
                           out_status, rstats, stats);
}

} // namespace

int hvx::restricted_direct_shared_devcount(hvx_index *ix, const float *queries, uint32_t b, uint32_t k, const uint32_t *d_rows, uint32_t rows_cap,
                                           const uint32_t *d_counters, uint64_t *out_ids, float *out_scores, uint32_t *out_counts, uint32_t *out_status,
                                           uint64_t *out_candidates, hvx_restricted_stats *rstats, hvx_stats *stats) {
    int rc;
    if ((rc = ix->stage(b, k))) return rc;
    PinView v;
    if ((rc = pin_view(ix, b, k, &v))) return rc;
    if ((rc = ix->pin_flags(8))) return rc;
    if ((rc = stage_and_validate(ix, queries, b, v))) return rc;
    if (stats) HIP_TRY(hipEventRecord(ix->ev0, ix->stream));
    if ((rc = restricted_direct_enqueue(ix, ix->s_queries, b, k, k, d_rows, std::max<uint32_t>(rows_cap, 1u), nullptr, nullptr, 0, v.ids, v.sc, v.cnt, v.st,
                                        nullptr, 0, d_counters)))
        return rc;
    if (stats) HIP_TRY(hipEventRecord(ix->ev1, ix->stream));
    HIP_TRY(hipMemcpyAsync(ix->h_flags, d_counters, 8, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    const uint32_t n_rows = ix->h_flags[0], population = ix->h_flags[1];
    if (out_candidates) *out_candidates = population;
    // RestrictedVectorCandidates::from_ids (restricted.rs:356-371): checked behind the scan here -- its answer is discarded
    if (population > 1000000u) return fail(HVX_ERR_CANDIDATE_LIMIT, "restricted vector search accepts at most 1000000 unique candidates");
    for (uint32_t q = 0; q < b; ++q) {
        out_counts[q] = 0;
        if (out_status) out_status[q] = HVX_OK;
        if (rstats) memset(&rstats[q], 0, sizeof(hvx_restricted_stats));
    }
    if (population == 0) return HVX_OK; // Empty: no results, before any validation (restricted.rs:539-541)
    for (uint32_t q = 0; q < b; ++q) {
        if (v.st[q]) {
            if (!out_status) return fail((int)v.st[q], "query %u rejected with status %u", q, v.st[q]);
            out_status[q] = v.st[q];
            continue;
        }
        if (rstats && ix->dev.has_entry) rstats[q].strategy = HVX_RESTRICTED_EXACT;
        out_counts[q] = v.cnt[q];
        memcpy(out_ids + (size_t)q * k, v.ids + (size_t)q * k, (size_t)v.cnt[q] * 8);
        memcpy(out_scores + (size_t)q * k, v.sc + (size_t)q * k, (size_t)v.cnt[q] * 4);
    }
    if (stats) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
        stats->queries += b;
        stats->vectors_loaded += (uint64_t)b * n_rows;
        stats->distance_computations += (uint64_t)b * n_rows;
        stats->device_ms += ms;
    }
    return HVX_OK;
}

int hvx::restricted_make_plan(const hvx_restricted_params &rp, uint64_t candidates, uint32_t dim, RestrictedPlan *out, const hvx_index *ix) {
    return make_plan(rp, candidates, dim, out, ix);
}
void hvx::restricted_sample_ranks(uint64_t candidates, uint32_t count, std::vector<uint64_t> &out) { sample_ranks(candidates, count, out); }
int hvx::restricted_run_plan(hvx_index *ix, const float *queries, uint32_t b, uint32_t k_stride, const RestrictedPlan &plan, const uint32_t *d_rows,
                             uint32_t n_rows, const uint32_t *d_samples, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                             uint32_t *out_status, hvx_restricted_stats *rstats, hvx_stats *stats) {
    if (plan.strategy == HVX_RESTRICTED_EXACT)
        return exact_shared_set(ix, queries, b, k_stride, plan.p.k, d_rows, n_rows, out_ids, out_scores, out_counts, out_status, rstats, stats);
    return walk_shared_set(ix, queries, b, k_stride, plan, d_rows, n_rows, d_samples, out_ids, out_scores, out_counts, out_status, rstats, stats);
}

extern "C" void hvx_restricted_params_default(hvx_restricted_params *p, uint32_t k, uint32_t ef) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->k = k;
    p->ef = ef;
    p->strategy = HVX_RESTRICTED_AUTO;
    p->beam_percent = 150; // FILTERED_BEAM_PERCENT (restricted.rs:52)
    p->directory_enabled = 1;
}

int hvx::restricted_search_host(hvx_index *ix, const float *queries, uint32_t b, const hvx_restricted_params &rp, const uint64_t *allowed_ids,
                                const uint64_t *allowed_offsets, uint64_t n_allowed, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                                uint32_t *out_status, hvx_restricted_stats *out_rstats, hvx_stats *stats) {
    if (!allowed_offsets)
        return restricted_set(ix, queries, b, rp, allowed_ids, n_allowed, out_ids, out_scores, out_counts, out_status, out_rstats, stats);
    // One candidate list per query (the operator's shape: storage.rs:140-163).  Lists that the plan answers exactly -- all of them under
    // the device plan, up to a GiB of rows each -- are scanned in ONE launch for the whole batch (round 6); anything else list by list.
    bool batched = rp.k <= 64u && restricted_direct_supported(ix, rp.k) && ix->opt[HVX_OPT_RESTRICTED_DIRECT] != 1u && !rp.explicit_budgets;
    for (uint32_t q = 0; q < b && batched; ++q) {
        const uint64_t len = allowed_offsets[q + 1] - allowed_offsets[q];
        if (len == 0) continue;
        if (len > 1000000ull) { batched = false; break; } // (duplicates decide whether the SET is within the limit: the host path dedupes)
        HostPlan hp;
        // planned against the list's length: an upper bound of the set's size (a list with duplicates is planned as the longer set it could be)
        if (make_plan(rp, len, ix->dev.dim, &hp, ix) != HVX_OK || hp.strategy != HVX_RESTRICTED_EXACT) batched = false;
    }
    if (batched)
        return direct_per_query_host(ix, queries, b, rp, allowed_ids, allowed_offsets, out_ids, out_scores, out_counts, out_status, out_rstats, stats);
    for (uint32_t q = 0; q < b; ++q) {
        const uint64_t a0 = allowed_offsets[q], a1 = allowed_offsets[q + 1];
        int rc = restricted_set(ix, queries + (size_t)q * ix->dev.dim, 1, rp, allowed_ids + a0, a1 - a0, out_ids + (size_t)q * rp.k,
                                out_scores + (size_t)q * rp.k, out_counts + q, out_status ? out_status + q : nullptr,
                                out_rstats ? out_rstats + q : nullptr, stats);
        if (rc) return rc;
    }
    return HVX_OK;
}

extern "C" int hvx_search_restricted_batch_params(const hvx_index *cix, const float *queries, uint32_t b, const hvx_restricted_params *params,
                                                  const uint64_t *allowed_ids, const uint64_t *allowed_offsets, uint64_t n_allowed,
                                                  uint64_t *out_ids, float *out_scores, uint32_t *out_counts, uint32_t *out_status,
                                                  hvx_restricted_stats *out_rstats, hvx_stats *stats) {
    if (!cix || !params) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    if (params->strategy > HVX_RESTRICTED_REFERENCE_PLAN) return fail(HVX_ERR_INVARIANT, "unknown restricted strategy %u", params->strategy);
    if (params->k == 0) return fail(HVX_ERR_K_RANGE, "result count must be non-zero");
    if (params->ef < params->k) return fail(HVX_ERR_K_RANGE, "search beam width %u is below the result count %u", params->ef, params->k);
    if (b == 0) return HVX_OK;
    std::lock_guard<std::mutex> lock(ix->mu);
    ix->sync_rewrites();
    HIP_TRY(hipSetDevice(ix->device));
    return restricted_search_host(ix, queries, b, *params, allowed_ids, allowed_offsets, n_allowed, out_ids, out_scores, out_counts, out_status, out_rstats, stats);
}
