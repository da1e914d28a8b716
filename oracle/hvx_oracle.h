/*
 * hvx_oracle.h -- CPU ORACLE for the HelixDB vector-search hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call it.  The shipped library
 * (helix-db_amd/csrc, libhelix_vec_gfx950.so) never includes, links or calls anything here.
 *
 * It is a plain-C restatement of the reference's algorithm (the Rust reference cannot be
 * compiled in this environment: no cargo/rustc).  Every function
 * cites the reference file:line it follows; paths are relative to
 * /root/reference/crates/db/src/search/vector/ unless written out.
 *
 * Parity pin: tests/test_oracle_golden.py checks this oracle against every golden vector /
 * known-answer test the reference's own tests hold for this path.
 */
#ifndef HVX_ORACLE_H
#define HVX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* distance metrics: mod.rs:283-290, distance/semantics.rs:31-56 */
enum { ORC_COSINE = 0, ORC_L2SQ = 1, ORC_L1 = 2 };

/* float kernel = which summation tree the reference takes on a given host
 * (spaces/simple.rs:45-112).  AVX_FMA is what an x86-64 host with avx+fma runs. */
enum {
    ORC_KERNEL_SCALAR = 0,
    ORC_KERNEL_SSE = 1,
    ORC_KERNEL_AVX = 2,
    ORC_KERNEL_AVX_FMA = 3,
    ORC_KERNEL_NEON = 4,
    /* same summation tree as ORC_KERNEL_AVX_FMA executed with the real AVX2+FMA instructions
     * (bit-identical, proven by tests/test_oracle_golden.py); this is what the timed cpu_baseline
     * uses.  Falls back to the portable emulation when the host lacks AVX2+FMA. */
    ORC_KERNEL_AVX_FMA_HW = 5
};

/* statuses mirror include/helix_vec.h (hvx_status) */
enum {
    ORC_OK = 0,
    ORC_ERR_DIMENSION = 1,
    ORC_ERR_NONFINITE = 2,
    ORC_ERR_ZERO_NORM = 3,
    ORC_ERR_MAGNITUDE = 4,
    ORC_ERR_K_RANGE = 5,
    ORC_ERR_CANDIDATE_LIMIT = 6,
    ORC_ERR_DEVICE = 7,
    ORC_ERR_INVARIANT = 8
};

/* ---- distance kernels (spaces/simple.rs, simple_avx.rs, simple_sse.rs, simple_neon.rs) ---- */
float orc_euclidean(const float *a, const float *b, uint32_t n, int kernel);
float orc_dot(const float *a, const float *b, uint32_t n, int kernel);
float orc_manhattan(const float *a, const float *b, uint32_t n);
/* hardware AVX2+FMA twin of ORC_KERNEL_AVX_FMA (used by the timed cpu_baseline and to prove the
 * portable fmaf emulation equals the real instructions); returns NaN if not compiled in. */
float orc_euclidean_avxfma_hw(const float *a, const float *b, uint32_t n);
float orc_dot_avxfma_hw(const float *a, const float *b, uint32_t n);
int orc_have_avxfma_hw(void);
/* hardware SSE twin of ORC_KERNEL_SSE (simple_sse.rs executed with the real instructions; NAN / 0 when the host has no SSE) */
float orc_euclidean_sse_hw(const float *a, const float *b, uint32_t n);
float orc_dot_sse_hw(const float *a, const float *b, uint32_t n);
int orc_have_sse_hw(void);

double orc_scaled_l2_norm(const float *v, uint32_t n);                 /* distance/cosine.rs:12-36 */
float orc_header(int metric, const float *v, uint32_t n);              /* new_header            */
float orc_distance(int metric, int kernel, const float *p, float p_hdr, const float *q, float q_hdr,
                   uint32_t n);                                        /* Distance::distance    */

/* ---- validation (domain.rs:18-157) ---- */
float orc_component_limit(int metric, uint32_t dim);                   /* 0 => no limit (cosine) */
int orc_validate_vector(int metric, const float *v, uint32_t n_actual, uint32_t dim, uint32_t *bad_index);
/* DistanceScore::try_new (parameters.rs:243-274): 0 ok (normalises -0), else ORC_ERR_INVARIANT */
int orc_distance_score(float *score);

/* layer draw (mod.rs:769-796) */
uint16_t orc_select_layer_from_uniform(float ml, float uniform);
float orc_default_ml_for_m(uint32_t m);                                 /* mod.rs:705-708 */

/* ---- index ---- */
typedef struct orc_index orc_index;

typedef struct {
    uint32_t expansion_steps;
    uint32_t neighbors_examined;
    uint32_t vectors_loaded;
    uint32_t distance_computations;
} orc_stats;

orc_index *orc_index_new(uint32_t dim, int metric, int kernel, uint32_t m, uint32_t m0,
                         uint32_t ef_construction);
void orc_index_free(orc_index *);
/* insert one row (mutation.rs:642-780 + insert_hnsw :787-895); node ids may come in any order.
 * level is the scripted layer (reference: with_scripted_layers / select_layer). */
int orc_index_insert(orc_index *, uint64_t node_id, const float *vector, uint16_t level);
/* test hook: the prune of add_bidirectional_link for one row (rank by distance to the owner, select_diverse + backfill) */
int orc_index_prune_candidates(const orc_index *, uint64_t owner_id, const uint64_t *cand_ids, uint32_t nc, uint32_t maxn,
                               uint64_t *out_ids, uint32_t *out_n);
/* seed a pre-built graph (scale_contracts.rs:95-155 style): rows given as CSR over node ids. */
int orc_index_seed(orc_index *, uint64_t n, const uint64_t *node_ids, const float *vectors,
                   const uint64_t *l0_offsets, const uint64_t *l0_neighbors,
                   const uint16_t *level, const uint64_t *up_offsets, const uint64_t *up_neighbors,
                   int has_entry, uint64_t entry_point, uint16_t max_layer);
/* delete one node (mutation.rs:1606-2055 stage_delete -> delete_from_layer -> relink_neighbor): every row that holds it loses the
 * edge, the affected sources are relinked from their joint neighbourhood, the entry point moves to the best remaining entry
 * candidate.  An unknown id succeeds with *existed = 0 (index.rs:2263).  Parity unpinned (no golden rows in the reference). */
int orc_index_delete(orc_index *, uint64_t node_id, int *existed);
int orc_index_is_live(const orc_index *, uint64_t node_id);
uint64_t orc_index_count(const orc_index *);   /* metadata.count: live nodes */
int orc_index_entry(const orc_index *, uint64_t *entry_point, uint16_t *max_layer);
/* export in the import layout of include/helix_vec.h (ids ascending).  Pass NULL to size. */
uint64_t orc_index_export_sizes(const orc_index *, uint64_t *l0_edges, uint64_t *up_rows, uint64_t *up_edges);
int orc_index_export(const orc_index *, uint64_t *node_ids, float *vectors, uint64_t *l0_offsets,
                     uint64_t *l0_neighbors, uint16_t *level, uint64_t *up_offsets,
                     uint64_t *up_neighbors);

/* SearchSession::run, strict-exhaustive arm (search.rs:1101-1230, :169-224, :267-1067 STRICT). */
int orc_search(const orc_index *, const float *query, uint32_t query_len, uint32_t k, uint32_t ef,
               uint64_t *out_ids, float *out_scores, uint32_t *out_count, orc_stats *stats);
/* nq independent orc_search calls spread over `threads` pthreads (one query per thread at a time:
 * BASELINE.md "CPU-baseline plan" (ii)); outputs are [nq][k]; per-query stats optional.
 * Returns the first non-OK status, 0 otherwise. */
int orc_search_batch_mt(const orc_index *, const float *queries, uint32_t nq, uint32_t k, uint32_t ef,
                        uint32_t threads, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                        orc_stats *stats);
/* exact scan over all rows or over an allowed id list (restricted.rs:753-835, :661-704). */
int orc_flat_search(const orc_index *, const float *query, uint32_t query_len, uint32_t k,
                    const uint64_t *allowed_ids, uint64_t n_allowed, uint64_t *out_ids,
                    float *out_scores, uint32_t *out_count);
/* exact scan over a bare matrix (no index object): rows [n][dim], ids = row numbers. */
int orc_flat_search_matrix(int metric, int kernel, const float *rows, uint64_t n, uint32_t dim,
                           const float *query, uint32_t k, uint64_t *out_ids, float *out_scores,
                           uint32_t *out_count);

/* ---- SimHash projections (hvx_oracle_simhash.c): unaligned_vector/simhash.rs, simhash.rs:44-59 ---- */
void orc_stdrng_key(uint64_t seed, uint32_t key[8]);                       /* seed_from_u64 -> ChaCha key */
void orc_chacha12_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]);
void orc_stdrng_u32(uint64_t seed, uint32_t *out, uint32_t n);            /* rand 0.10 StdRng stream (pinned by KAT) */
int orc_simhash_planes(uint32_t dim, uint64_t seed, float *planes /*[64][dim]*/);
uint64_t orc_simhash_hash(const float *planes, const float *vec, uint32_t dim);
uint64_t orc_order_code(uint64_t bits);
uint32_t orc_simhash_collisions(uint64_t a, uint64_t b);
uint64_t orc_query_seed(uint64_t query_simhash, uint64_t entry_point, uint64_t ef);

/* ---- non-strict layer-0 arms (hvx_oracle_adaptive.inc): policy.rs, search.rs:595-829, randomness.rs ---- */
enum { ORC_SIMHASH_ALWAYS = 0, ORC_SIMHASH_ADAPTIVE = 1, ORC_SIMHASH_OFF = 2 };          /* mod.rs SimHashMode */
enum { ORC_SAMPLING_EXHAUSTIVE = 0, ORC_SAMPLING_FIXED = 1, ORC_SAMPLING_ADAPTIVE = 2 }; /* policy.rs SamplingDecision */
enum { ORC_BYPASS_READY = 0, ORC_BYPASS_BYPASSING = 1, ORC_BYPASS_COOLING = 2 };         /* policy.rs AdaptiveBypassState */
enum { ORC_TRIGGER_NONE = 0, ORC_TRIGGER_READ_BUDGET = 1, ORC_TRIGGER_LOW_YIELD = 2, ORC_TRIGGER_BOTH = 3 };

/* SearchParams (mod.rs:410-621) resolved against the index-level VectorConfig knobs
 * (metadata.rs:38-44; defaults 43 / 0.8 / true / 0.1, mod.rs:313-329).  Overrides: < 0 means None. */
typedef struct {
    uint32_t k, ef;
    uint32_t simhash_mode;
    float pre_simhash_sampling_ratio_override;
    uint32_t bypass_min_frontier, bypass_window_expansions;
    float bypass_min_filter_rate;
    uint32_t read_budget_multiplier;
    float simhash_sampling_ratio_override;
    float simhash_failure_prob_override;
    /* index configuration */
    uint32_t simhash_threshold;
    float sampling_ratio;
    uint32_t adaptive_enabled;
    float adaptive_failure_prob;
    /* 1: SimHash rows come from the resident snapshot (memory_store.rs:329-335: not counted as reads);
     * 0: Uncached handle, every first-seen row is one stable-view read (memory_store.rs:338-347). */
    uint32_t resident_simhash;
    /* SearchRandomness::Seeded (test only, randomness.rs:94-96); 0 = QueryDerived */
    uint32_t use_fixed_seed;
    uint64_t fixed_seed;
} orc_search_params;
void orc_search_params_default(orc_search_params *p, uint32_t k); /* SearchParams::new(k) + default config */

typedef struct {   /* SearchStats (mod.rs:629-700), the fields the traversal itself determines */
    uint32_t expansion_steps, neighbors_examined, vectors_loaded, distance_computations;
    uint32_t simhash_filtered, simhash_examined, simhash_missing_hash;
    uint32_t simhash_passed_before_sampling, simhash_passed_after_sampling;
    uint32_t pre_simhash_sample_kept, pre_simhash_sample_dropped;
    uint32_t simhash_bypass_expansions, simhash_skipped_candidates;
    uint32_t simhash_bypass_trigger_budget, simhash_bypass_trigger_low_yield;
    uint32_t txn_get_simhash_filter;
    uint32_t active_simhash_threshold_sum, active_simhash_threshold_samples;
    uint32_t effective_beam_len_sum, effective_beam_len_samples;
    uint32_t active_sampling_ratio_samples;
    uint32_t rng_words;                 /* u32 outputs drawn from the query RNG (not a reference field) */
    double active_sampling_ratio_sum;
} orc_adaptive_stats;

/* Layer0Policy::from_deployed(..).with_adaptive_bypass(..).decide(ctx) as one pure call (policy.rs:54-183). */
typedef struct {
    uint32_t metric, simhash_mode, configured_threshold;
    float sampling_ratio, pre_sampling_override /* <0 None */;
    uint32_t adaptive_enabled;
    float failure;
    uint32_t bypass_from_deployed;      /* 0: AdaptiveBypassPolicy::Disabled (from_deployed default) */
    uint32_t bypass_ef, bypass_min_frontier, bypass_window_expansions;
    float bypass_min_filter_rate;
    uint32_t bypass_read_budget_multiplier;
    /* SimHashContext */
    uint32_t topk_ready, ef, search_frontier_len, candidate_frontier_len;
    float current, delta;
    uint32_t state, state_remaining;
    uint64_t simhash_filter_reads, window_examined, window_filtered, window_expansions;
} orc_policy_input;
typedef struct {
    uint32_t fetch_missing, filter_cached, has_threshold, threshold;
    uint32_t pre_sampling_kind; float pre_sampling_probability;
    uint32_t sampling_kind; float sampling_probability;
    float base_sampling_probability;
    uint32_t bypassed, next_state, next_state_remaining, trigger;
} orc_policy_decision;
void orc_policy_decide(const orc_policy_input *in, orc_policy_decision *out);
float orc_candidate_probability(uint32_t kind, float probability, uint32_t similarity_bits, int has_threshold,
                                uint32_t threshold);                        /* policy.rs:417-433 */
uint32_t orc_pre_sampling_decision(float base_ratio, uint32_t frontier, uint32_t ef, float *probability); /* :540-557 */
float orc_adaptive_sampling_ratio(float base, uint32_t search_frontier_len, uint32_t ef, float current, float delta);
uint32_t orc_adaptive_threshold(uint32_t topk_ready, float delta, uint32_t configured, float failure);   /* :577-599 */

/* query RNG session (randomness.rs:127-164).  should_sample is pinned through the SimHasher KAT (same
 * generator, same random::<f32>()); choose_index restates rand's `random_range(0..n)` from its published
 * algorithm (widening multiply with one bias-correction draw) and has NO reference known answer:
 * PARITY UNPINNED for the two fallback sites that call it (search.rs:667,818). */
typedef struct { uint64_t seed; uint32_t key[8]; uint32_t buf[16]; uint64_t block; uint32_t pos; int ready; uint32_t words; } orc_rng;
void orc_rng_seed(orc_rng *r, uint64_t seed);
int orc_rng_should_sample(orc_rng *r, float ratio);
int64_t orc_rng_choose_index(orc_rng *r, uint64_t n);   /* -1 when n == 0 */

/* per-node SimHash rows: either given (e.g. decoded from the persisted keys) or computed with
 * SimHasher(dim, seed) (unaligned_vector/simhash.rs).  Required by orc_search_params(). */
int orc_index_set_simhash(orc_index *, uint64_t seed, const uint64_t *node_hashes /* NULL: compute */);
int orc_index_get_simhash(const orc_index *, uint64_t *node_hashes /* export order = ascending ids */);
uint64_t orc_index_query_simhash(const orc_index *, const float *query);

/* SearchSession::run with full SearchParams: strict-exhaustive parameters take the STRICT arm,
 * everything else search_layer0_with_simhash::<_, false> (search.rs:267-1067). */
int orc_search_params_run(const orc_index *, const float *query, uint32_t query_len, const orc_search_params *p,
                          uint64_t *out_ids, float *out_scores, uint32_t *out_count, orc_adaptive_stats *stats);
int orc_search_params_batch_mt(const orc_index *, const float *queries, uint32_t nq, const orc_search_params *p,
                               uint32_t threads, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                               orc_adaptive_stats *stats);

/* ---- restricted search: planner + filter-aware walk (hvx_oracle_restricted.inc; restricted.rs:196-260,303-462,528-1148) ---- */
enum { ORC_RESTRICTED_NONE = 0, ORC_RESTRICTED_EXACT = 1, ORC_RESTRICTED_FILTERED = 2 };              /* RestrictedSearchStrategy */
enum { ORC_TERM_NONE = 0, ORC_TERM_EXHAUSTED = 1, ORC_TERM_BEAM_COMPLETE = 2, ORC_TERM_ROUTING_BUDGET = 3,
       ORC_TERM_BRIDGE_BUDGET = 4, ORC_TERM_VECTOR_BUDGET = 5 };                                        /* RestrictedSearchTermination */
typedef struct {   /* RestrictedExecutionPlan + FilteredGraphBudgets (restricted.rs:216-281) */
    uint32_t strategy, k;
    uint64_t ef_filtered, routing_rows, bridge_rows, vector_payloads, sampled_seeds, directory_seeds;
} orc_restricted_plan_t;
typedef struct {   /* RestrictedSearchStats (restricted.rs:147-166) */
    uint32_t strategy, termination;
    uint64_t ef_filtered, directory_scan_calls, directory_rows, directory_decoded_bytes, directory_hits, simhash_row_requests,
        companion_row_requests, routing_rows, bridge_rows, bridge_frontier_pushes, neighbor_multi_get_calls,
        vector_payload_requests, vector_bytes, distance_computations;
} orc_restricted_stats;
/* beam_percent 0 = FILTERED_BEAM_PERCENT (150) */
void orc_restricted_plan(uint64_t candidates, uint32_t dim, uint32_t k, uint32_t ef, uint32_t beam_percent, orc_restricted_plan_t *out);
uint64_t orc_deterministic_sample_ranks(uint64_t candidates, uint64_t limit, uint64_t *out_ranks); /* restricted.rs:321-342 */
/* restricted_filter_aware_search with explicit budgets (plan->k, ->ef_filtered, ... as the reference's tests pass them) */
int orc_restricted_filter_aware_search(const orc_index *, const float *query, uint32_t query_len, const uint64_t *allowed_ids,
                                       uint64_t n_allowed, const orc_restricted_plan_t *plan, int directory_enabled,
                                       uint64_t *out_ids, float *out_scores, uint32_t *out_count, orc_restricted_stats *stats);
/* VectorIndex::search_restricted (restricted.rs:466-613).  directory_enabled = VectorIndex::with_simhash_directory().
 * The filtered walk needs the per-node SimHash rows (orc_index_set_simhash). */
int orc_search_restricted(const orc_index *, const float *query, uint32_t query_len, uint32_t k, uint32_t ef,
                          const uint64_t *allowed_ids, uint64_t n_allowed, uint32_t beam_percent, int directory_enabled,
                          uint64_t *out_ids, float *out_scores, uint32_t *out_count, orc_restricted_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
