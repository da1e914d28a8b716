// hvx_params.hip -- host side of the parameterised search: full SearchParams (crates/db/src/search/vector/mod.rs:410-621)
// resolved against the index-level VectorIndexConfig knobs, projected ONCE per call into the kernel's AdaptArgs
// (Layer0Policy::from_deployed, AdaptiveBypassPolicy::from_deployed: policy.rs:54-110,203-226), plus the per-node SimHash
// rows and the per-batch query fingerprints the non-strict arms need.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "hvx_host.h"

using namespace hvx;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(HVX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

extern "C" void hvx_search_params_default(hvx_search_params *p, uint32_t k) { // SearchParams::new (mod.rs:482-500)
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->k = k;
    p->ef = k > 100 ? k : 100;
    p->simhash_mode = HVX_SIMHASH_ADAPTIVE;
    p->pre_simhash_sampling_ratio_override = -1.0f;
    p->bypass_min_frontier = 24;
    p->bypass_window_expansions = 4;
    p->bypass_min_filter_rate = 0.12f;
    p->read_budget_multiplier = 3;
    p->simhash_sampling_ratio_override = -1.0f;
    p->simhash_failure_prob_override = -1.0f;
}

extern "C" void hvx_simhash_config_default(hvx_simhash_config *c) { // mod.rs:313-329, generation.rs:25
    if (!c) return;
    c->seed = 42;
    c->simhash_threshold = 43;
    c->sampling_ratio = 0.8f;
    c->adaptive_enabled = 1;
    c->adaptive_failure_prob = 0.1f;
    c->resident_snapshot = 1;
}

static bool unit_interval(float v) { return std::isfinite(v) && v >= 0.0f && v <= 1.0f; }      // parameters.rs:170-184
static bool open_unit(float v) { return std::isfinite(v) && v > 0.0f && v < 1.0f; }            // parameters.rs:196-210

extern "C" int hvx_index_set_simhash(hvx_index *ix, const hvx_simhash_config *cfg, const uint64_t *node_hashes) {
    if (!ix || !cfg) return fail(HVX_ERR_INVARIANT, "null argument");
    // VectorIndexConfig validation (mod.rs:331-375): threshold <= SIMHASH_BITS, ratio in [0,1], failure in (0,1)
    if (cfg->simhash_threshold > 64) return fail(HVX_ERR_K_RANGE, "collision threshold %u exceeds the 64-bit SimHash width", cfg->simhash_threshold);
    if (!unit_interval(cfg->sampling_ratio)) return fail(HVX_ERR_K_RANGE, "sampling ratio outside the closed unit interval");
    if (!open_unit(cfg->adaptive_failure_prob)) return fail(HVX_ERR_K_RANGE, "failure probability outside the open unit interval");
    if (ix->is_fork) return fail(HVX_ERR_INVARIANT, "SimHash rows belong to the index image: attach them to the handle returned by import, then fork");
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    const uint32_t n = ix->dev.n, dim = ix->dev.dim;
    int rc;
    if (!ix->d_planes_t || ix->sh_cfg.seed != cfg->seed) {
        std::vector<float> planes_t((size_t)64 * dim);
        simhash_planes_transposed(dim, cfg->seed, planes_t.data());
        if (!ix->d_planes_t && (rc = ix->dalloc((void **)&ix->d_planes_t, planes_t.size() * 4))) return rc;
        HIP_TRY(hipMemcpy(ix->d_planes_t, planes_t.data(), planes_t.size() * 4, hipMemcpyHostToDevice));
    }
    if (!ix->d_node_hash && (rc = ix->dalloc((void **)&ix->d_node_hash, (size_t)std::max<uint64_t>(std::max<uint64_t>(ix->cap_rows, n), 1) * 8))) return rc; // (room for appended rows)
    if (!ix->d_qhash && (rc = ix->dalloc((void **)&ix->d_qhash, (size_t)ix->max_batch * 8))) return rc;
    if (!ix->d_thr_break && (rc = ix->dalloc((void **)&ix->d_thr_break, 64 * 4))) return rc;
    if (!ix->d_astats && (rc = ix->dalloc((void **)&ix->d_astats, (size_t)ix->max_batch * sizeof(hvx_adaptive_stats)))) return rc;
    if (node_hashes) {
        HIP_TRY(hipMemcpyAsync(ix->d_node_hash, node_hashes, (size_t)n * 8, hipMemcpyDefault, ix->stream)); // host or device source
        HIP_TRY(hipStreamSynchronize(ix->stream));
    } else {
        if (ix->dev.dtype == HVX_FP8_E4M3)
            return fail(HVX_ERR_UNSUPPORTED, "SimHash rows are not recomputed from fp8 storage; pass node_hashes");
        if (ix->dev.dtype == HVX_BF16) // the stored (rounded) values are the vectors of this index
            HIP_TRY(launch_simhash_rows_bf16(ix->d_planes_t, ix->dev.vecb, dim, n, ix->d_node_hash, ix->stream));
        else
            HIP_TRY(launch_simhash_rows(ix->d_planes_t, ix->dev.vec, dim, ix->dev.ld, n, ix->d_node_hash, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream));
    }
    if (!cfg->resident_snapshot && !ix->d_bitmap2) { // second spill bitmap, all-zero between launches like the first
        const size_t bytes = (size_t)ix->max_batch * ix->words_per_query * 4;
        if ((rc = ix->dalloc((void **)&ix->d_bitmap2, bytes))) return rc;
        HIP_TRY(hipMemset(ix->d_bitmap2, 0, bytes));
    }
    ix->sh_cfg = *cfg;
    ix->has_simhash = true;
    ix->thr_configured = 0xFFFFFFFFu;
    return HVX_OK;
}

extern "C" int hvx_index_get_simhash(const hvx_index *cix, uint64_t *out) {
    if (!cix || !out) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    if (!ix->has_simhash) return fail(HVX_ERR_INVARIANT, "no SimHash rows attached (hvx_index_set_simhash)");
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    HIP_TRY(hipMemcpy(out, ix->d_node_hash, (size_t)ix->dev.n * 8, hipMemcpyDeviceToHost));
    return HVX_OK;
}

// policy.rs:577-599 adaptive_threshold for topk_ready and configured >= 1, evaluated with THIS host's libm exactly as the
// reference evaluates it on this host (f32 `acos`, `ln`, `sqrt`, `floor`)
static uint32_t adaptive_threshold_host(float delta, uint32_t configured, float failure) {
    const float normalized = delta < 0.0f ? 0.0f : (delta > 1.0f ? 1.0f : delta);
    float cs = 1.0f - 2.0f * normalized;
    cs = cs < -1.0f ? -1.0f : (cs > 1.0f ? 1.0f : cs);
    const float collision = 1.0f - acosf(cs) / 3.14159274101257324f;
    const float bits = 64.0f;
    const float margin = sqrtf((bits * logf(1.0f / failure)) / 2.0f);
    float v = floorf(bits * collision - margin);
    v = v < 1.0f ? 1.0f : (v > bits ? bits : v);
    const uint32_t t = (uint32_t)v;
    return t < configured ? t : configured;
}

// The threshold is a non-increasing step function of delta with at most 64 steps; the kernel evaluates it as
// #{t : delta <= brk[t-1]} so that no transcendental runs on the device.  brk[t-1] = the largest f32 delta in [0, +inf)
// whose threshold is >= t (bit patterns of non-negative floats order like the floats), -1 if there is none.
static int build_threshold_table(uint32_t configured, float failure, float brk[64]) {
    auto thr = [&](uint32_t bits) {
        float d;
        memcpy(&d, &bits, 4);
        return adaptive_threshold_host(d, configured, failure);
    };
    const uint32_t top = 0x7F7FFFFFu; // f32::MAX: DistanceScore is finite
    for (uint32_t t = 1; t <= 64; ++t) {
        if (thr(0u) < t) { brk[t - 1] = -1.0f; continue; }
        uint32_t lo = 0u, hi = top; // invariant: thr(lo) >= t
        if (thr(hi) >= t) lo = hi;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo + 1u) / 2u;
            if (thr(mid) >= t) lo = mid;
            else hi = mid - 1u;
        }
        // monotonicity check around the step: the table is only valid for a monotone libm
        for (uint32_t d = 1; d <= 64; ++d) {
            if (lo >= d && thr(lo - d) < t) return fail(HVX_ERR_INVARIANT, "host acosf is not monotone near %#x", lo);
            if (lo + d <= top && thr(lo + d) >= t) return fail(HVX_ERR_INVARIANT, "host acosf is not monotone near %#x", lo);
        }
        memcpy(&brk[t - 1], &lo, 4);
    }
    return HVX_OK;
}

// host-only utility (no device work): the table the kernels evaluate adaptive_threshold with, for inspection and tests
extern "C" int hvx_adaptive_threshold_table(uint32_t configured, float failure, float *out_brk /*[64]*/) {
    if (!out_brk) return fail(HVX_ERR_INVARIANT, "null argument");
    if (configured > 64) return fail(HVX_ERR_K_RANGE, "collision threshold %u exceeds the 64-bit SimHash width", configured);
    if (!open_unit(failure)) return fail(HVX_ERR_K_RANGE, "failure probability outside the open unit interval");
    if (configured == 0) { // threshold is 0 whatever delta is (policy.rs:582-583): no step reaches 1
        for (int t = 0; t < 64; ++t) out_brk[t] = -1.0f;
        return HVX_OK;
    }
    return build_threshold_table(configured, failure, out_brk);
}

// validate + project; strict = the params describe the strict-exhaustive arm (no fingerprint, no RNG)
static int project_params(hvx_index *ix, const hvx_search_params *p, AdaptArgs *ad, bool *strict) {
    int rc = check_k_ef(p->k, p->ef);
    if (rc) return rc;
    if (p->simhash_mode > HVX_SIMHASH_OFF) return fail(HVX_ERR_K_RANGE, "unknown SimHash mode %u", p->simhash_mode);
    if (p->bypass_min_frontier == 0 || p->bypass_window_expansions == 0 || p->read_budget_multiplier == 0)
        return fail(HVX_ERR_K_RANGE, "SimHash bypass tuning parameters must be non-zero");              // mod.rs:555-583
    if (!unit_interval(p->bypass_min_filter_rate)) return fail(HVX_ERR_K_RANGE, "bypass filter rate outside the closed unit interval");
    const bool has_pre = p->pre_simhash_sampling_ratio_override >= 0.0f, has_sr = p->simhash_sampling_ratio_override >= 0.0f;
    const bool has_fp = p->simhash_failure_prob_override >= 0.0f;
    if (has_pre && !unit_interval(p->pre_simhash_sampling_ratio_override)) return fail(HVX_ERR_K_RANGE, "pre-sampling ratio outside the closed unit interval");
    if (has_sr && !unit_interval(p->simhash_sampling_ratio_override)) return fail(HVX_ERR_K_RANGE, "sampling ratio outside the closed unit interval");
    if (has_fp && !open_unit(p->simhash_failure_prob_override)) return fail(HVX_ERR_K_RANGE, "failure probability outside the open unit interval");
    // SearchParams::requires_query_simhash (mod.rs:546-552)
    *strict = p->simhash_mode == HVX_SIMHASH_OFF && !(has_pre && p->pre_simhash_sampling_ratio_override < 1.0f);
    if (*strict) return HVX_OK;
    if (!ix->has_simhash)
        return fail(HVX_ERR_INVARIANT, "layer-zero filtering or sampling requires a query fingerprint: attach the SimHash rows first (hvx_index_set_simhash)");
    const hvx_simhash_config &c = ix->sh_cfg;
    const uint32_t metric = ix->dev.metric, mode = p->simhash_mode;
    const float failure = has_fp ? p->simhash_failure_prob_override : c.adaptive_failure_prob;
    memset(ad, 0, sizeof(*ad));
    ad->node_hash = ix->d_node_hash;
    ad->qhash = ix->d_qhash;
    ad->thr_break = ix->d_thr_break;
    ad->stats = nullptr; // set per call: the diagnostics-free kernel build runs when the caller wants no SearchStats
    ad->configured = c.simhash_threshold;
    ad->ratio = has_sr ? p->simhash_sampling_ratio_override : c.sampling_ratio;
    ad->pre_override = has_pre ? p->pre_simhash_sampling_ratio_override : -1.0f;
    // Layer0Policy::from_deployed (policy.rs:54-110): filtering is cosine-only, the sampling policy is metric-independent
    if (mode == HVX_SIMHASH_OFF || metric != kCosine) ad->filtering = 0;
    else if (mode == HVX_SIMHASH_ALWAYS) ad->filtering = 1;
    else ad->filtering = c.adaptive_enabled ? 2 : 1;
    if (mode == HVX_SIMHASH_OFF) ad->sampling = 0;
    else if (mode == HVX_SIMHASH_ALWAYS) ad->sampling = 1;
    else ad->sampling = c.adaptive_enabled ? 2 : 1;
    // AdaptiveBypassPolicy::from_deployed (policy.rs:203-226); the read budget only ever fills for an uncached handle
    ad->bypass_windowed = mode == HVX_SIMHASH_ADAPTIVE ? 1 : 0;
    ad->count_reads = c.resident_snapshot ? 0u : 1u;
    {
        const uint64_t budget = (uint64_t)p->ef * p->read_budget_multiplier;
        ad->read_budget = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(budget, p->bypass_min_frontier), 0xFFFFFFFFull);
    }
    ad->min_frontier = p->bypass_min_frontier;
    ad->window_expansions = p->bypass_window_expansions;
    ad->min_filter_rate = p->bypass_min_filter_rate;
    if (ad->filtering == 2 && (ix->thr_configured != ad->configured || ix->thr_failure != failure)) {
        float brk[64];
        if ((rc = build_threshold_table(ad->configured, failure, brk))) return rc;
        HIP_TRY(hipMemcpyAsync(ix->d_thr_break, brk, sizeof(brk), hipMemcpyHostToDevice, ix->stream));
        HIP_TRY(hipStreamSynchronize(ix->stream)); // brk lives on this stack frame
        ix->thr_configured = ad->configured;
        ix->thr_failure = failure;
    }
    return HVX_OK;
}

static int enqueue_params(hvx_index *ix, const float *d_queries, uint32_t b, const hvx_search_params *p, const AdaptArgs *ad,
                          bool strict, uint64_t *d_ids, float *d_scores, uint32_t *d_counts, uint32_t *d_status,
                          hvx_query_stats *d_qstats, hvx_adaptive_stats *d_astats, bool timed) {
    if (strict || p->ef > 992u) { // (beams beyond the kernels' widest are answered by the exact scan: no filter / sampling stage runs)
        if (d_astats) HIP_TRY(hipMemsetAsync(d_astats, 0, (size_t)b * sizeof(hvx_adaptive_stats), ix->stream));
        return enqueue_search(ix, d_queries, b, p->k, p->ef, d_ids, d_scores, d_counts, d_status, d_qstats, timed);
    }
    AdaptArgs a = *ad;
    a.stats = d_astats;
    // query fingerprints (search.rs:1143-1154); invalid queries are rejected by the search kernel's status check
    HIP_TRY(launch_simhash_rows(ix->d_planes_t, d_queries, ix->dev.dim, ix->dev.dim, b, ix->d_qhash, ix->stream));
    return enqueue_search(ix, d_queries, b, p->k, p->ef, d_ids, d_scores, d_counts, d_status, d_qstats, timed, &a);
}

// project + enqueue for a caller that already holds the handle's lock (hvx_shard.hip)
int hvx::enqueue_search_params(hvx_index *ix, const float *d_queries, uint32_t b, const hvx_search_params *params, uint64_t *d_ids, float *d_scores,
                               uint32_t *d_counts, uint32_t *d_status) {
    ix->sync_rewrites();
    AdaptArgs ad;
    bool strict = false;
    int rc = project_params(ix, params, &ad, &strict);
    if (rc) return rc;
    return enqueue_params(ix, d_queries, b, params, &ad, strict, d_ids, d_scores, d_counts, d_status, nullptr, nullptr, false);
}

extern "C" int hvx_search_batch_params_device(const hvx_index *cix, const float *d_queries, uint32_t b, const hvx_search_params *params,
                                              uint64_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts,
                                              uint32_t *d_out_status, hvx_query_stats *d_query_stats,
                                              hvx_adaptive_stats *d_adaptive_stats, hvx_stats *stats) {
    if (!cix || !params) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    std::lock_guard<std::mutex> lock(ix->mu);
    ix->sync_rewrites();
    HIP_TRY(hipSetDevice(ix->device));
    AdaptArgs ad;
    bool strict = false;
    int rc = project_params(ix, params, &ad, &strict);
    if (rc) return rc;
    if (b == 0) return HVX_OK;
    if (b > ix->max_batch) return fail(HVX_ERR_UNSUPPORTED, "batch %u exceeds max_batch %u given at import", b, ix->max_batch);
    rc = enqueue_params(ix, d_queries, b, params, &ad, strict, d_out_ids, d_out_scores, d_out_counts, d_out_status, d_query_stats,
                        d_adaptive_stats, stats != nullptr);
    if (rc) return rc;
    if (stats) return collect_stats(ix, b, d_query_stats, stats);
    return HVX_OK;
}

extern "C" int hvx_search_batch_params(const hvx_index *cix, const float *queries, uint32_t b, const hvx_search_params *params,
                                       uint64_t *out_ids, float *out_scores, uint32_t *out_counts, uint32_t *out_status,
                                       hvx_stats *stats, hvx_query_stats *query_stats, hvx_adaptive_stats *adaptive_stats) {
    if (!cix || !params) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    std::lock_guard<std::mutex> lock(ix->mu);
    ix->sync_rewrites();
    HIP_TRY(hipSetDevice(ix->device));
    AdaptArgs ad;
    bool strict = false;
    int rc = project_params(ix, params, &ad, &strict);
    if (rc) return rc;
    if (b == 0) return HVX_OK;
    const uint32_t mb = ix->max_batch, k = params->k;
    std::vector<uint32_t> status(b, 0);
    if (adaptive_stats && !ix->d_astats && (rc = ix->dalloc((void **)&ix->d_astats, (size_t)mb * sizeof(hvx_adaptive_stats)))) return rc;
    for (uint32_t c0 = 0; c0 < b; c0 += mb) {
        const uint32_t cb = std::min(mb, b - c0);
        if ((rc = ix->stage(cb, k))) return rc;
        if ((rc = ix->stage_in(queries + (size_t)c0 * ix->dev.dim, cb))) return rc;
        rc = enqueue_params(ix, ix->s_queries, cb, params, &ad, strict, ix->s_ids, ix->s_scores, ix->s_counts, ix->s_status, nullptr,
                            adaptive_stats ? ix->d_astats : nullptr, stats != nullptr);
        if (rc) return rc;
        if ((rc = ix->stage_out(cb, k))) return rc;
        if (query_stats)
            HIP_TRY(hipMemcpyAsync(query_stats + c0, ix->d_qstats, (size_t)cb * sizeof(hvx_query_stats), hipMemcpyDeviceToHost, ix->stream));
        if (adaptive_stats)
            HIP_TRY(hipMemcpyAsync(adaptive_stats + c0, ix->d_astats, (size_t)cb * sizeof(hvx_adaptive_stats), hipMemcpyDeviceToHost, ix->stream));
        if (stats) {
            if ((rc = collect_stats(ix, cb, nullptr, stats))) return rc;
        } else {
            HIP_TRY(hipStreamSynchronize(ix->stream));
        }
        ix->deliver(cb, k, out_ids + (size_t)c0 * k, out_scores + (size_t)c0 * k, out_counts + c0, status.data() + c0);
    }
    if (out_status) {
        memcpy(out_status, status.data(), (size_t)b * 4);
        return HVX_OK;
    }
    for (uint32_t i = 0; i < b; ++i)
        if (status[i]) return fail((int)status[i], "query %u rejected with status %u", i, status[i]);
    return HVX_OK;
}
