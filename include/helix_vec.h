/*
 * helix_vec.h -- C ABI of libhelix_vec_gfx950.so, the MI355X-native vector-search executor that
 * sits behind HelixDB's vector index facade.
 *
 * The reference has no FFI for this path; these are the entry points a Rust
 * `extern "C"` shim at the B2 seam would bind.  Every entry point cites the reference interface
 * it replaces (paths relative to /root/reference/).  Conventions follow the project's UniFFI
 * style: POD buffers in, caller-owned out-buffers, integer status, thread-local error string.
 *
 * All host-pointer entry points are re-entrant on an immutable index; calls on ONE index are
 * serialised on that index's HIP stream.  *_device entry points take device pointers, enqueue on
 * the index's stream and return without synchronising (use hvx_index_sync).
 */
#ifndef HELIX_VEC_H
#define HELIX_VEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hvx_index hvx_index; /* opaque, device-resident, immutable after import */
typedef struct hvx_csr hvx_csr;     /* opaque device-resident CSR graph for traverse/prefilter */

/* crates/db/src/search/vector/mod.rs:283-290, distance/semantics.rs:31-56 */
enum hvx_metric { HVX_COSINE_HALF = 0, HVX_L2_SQUARED = 1, HVX_MANHATTAN = 2 };
/* only F32 is an active codec in the reference (distance/mod.rs:17-41); BF16/FP8 are new */
enum hvx_dtype { HVX_F32 = 0, HVX_BF16 = 1, HVX_FP8_E4M3 = 2 };
/* which host summation tree scores must be bit-identical to (spaces/simple.rs:45-112):
 * the reference's result depends on the CPU it runs on; the device reproduces the chosen tree. */
enum hvx_float_kernel {
    HVX_KERNEL_SCALAR = 0,  /* FloatSimd::Scalar (feature force-vector-scalar-kernel, or no SIMD unit) */
    HVX_KERNEL_SSE = 1,     /* FloatSimd::Sse: x86 without AVX -- 4 x 4 lanes, mul then add (simple_sse.rs:17-67), dim >= 16 */
    HVX_KERNEL_AVX = 2,     /* FloatSimd::Avx: 4 x 8 lanes, mul then add (simple_avx.rs:15-125), dim >= 32 */
    HVX_KERNEL_AVX_FMA = 3, /* FloatSimd::AvxFma: 4 x 8 lanes, fused (simple_avx.rs:128-238), dim >= 32 */
    HVX_KERNEL_NEON = 4     /* FloatSimd::Neon: aarch64 -- 4 x 4 lanes, vfmaq_f32 + vaddvq_f32 (simple_neon.rs:10-95), dim >= 16 */
};

/* HelixDbError variants of the path (crates/db/src/error.rs via search.rs:1101-1230,
 * restricted.rs:196-260,356-371) */
enum hvx_status {
    HVX_OK = 0,
    HVX_ERR_DIMENSION = 1,       /* InvalidDimension                       domain.rs:118-123 */
    HVX_ERR_NONFINITE = 2,       /* InvalidVectorComponent{index}          domain.rs:124-128 */
    HVX_ERR_ZERO_NORM = 3,       /* ZeroNormCosineVector                   domain.rs:129-131 */
    HVX_ERR_MAGNITUDE = 4,       /* ComponentMagnitudeExceeded             domain.rs:132-146 */
    HVX_ERR_K_RANGE = 5,         /* k==0, ef<k (parameters.rs:100-133), restricted k>800 (restricted.rs:55) */
    HVX_ERR_CANDIDATE_LIMIT = 6, /* >1,000,000 restricted candidates       restricted.rs:40,356-371 */
    HVX_ERR_DEVICE = 7,          /* HIP failure / extension unavailable */
    HVX_ERR_INVARIANT = 8,       /* InvariantViolation (bad graph row, invalid score, ...) */
    HVX_ERR_UNSUPPORTED = 9,     /* configuration outside what this build implements */
    HVX_PENDING = 10,            /* NOT an error: hvx_batcher_poll / _wait -- the ticket's batch has not completed yet */
    HVX_ERR_BUSY = 11,           /* hvx_batcher_submit would have to block (the open batch is full): submit again later */
    HVX_ERR_TIMEOUT = 12         /* hvx_shard_group_wait: the step (local search + exchange + merge) has not drained in the time given */
};

enum hvx_direction { HVX_DIR_OUT = 0, HVX_DIR_IN = 1, HVX_DIR_BOTH = 2 };

/* VectorIndexMetadata + config subset (values/vectors/metadata.rs:22-62) needed by search */
typedef struct hvx_index_desc {
    uint32_t dim;
    uint32_t metric;       /* hvx_metric */
    uint32_t dtype;        /* hvx_dtype (storage on device; import source is always f32) */
    uint32_t float_kernel; /* hvx_float_kernel */
    uint64_t n;            /* rows in this shard */
    uint32_t m, m0;        /* degree limits (upper layers / layer 0) */
    uint32_t has_entry;    /* 0 => VectorIndexState::Empty */
    uint32_t max_layer;
    uint64_t entry_point;  /* external node id */
    uint64_t shard_id_lo, shard_id_hi; /* informational: id range held by this shard */
    int32_t device;        /* HIP device ordinal, -1 = current */
    uint32_t max_batch;    /* largest query batch the index must serve (sizes scratch); 0 => 1024 */
} hvx_index_desc;

/* SearchStats subset (mod.rs:629-700) -- exactly the algorithmic-bytes numerators,
 * summed over the batch, plus device timing of the call. */
typedef struct hvx_stats {
    uint64_t queries;
    uint64_t expansion_steps;
    uint64_t neighbors_examined;
    uint64_t vectors_loaded;
    uint64_t distance_computations;
    uint64_t tie_overflow_queries; /* queries whose beam had > slack equal-score entries (exactness not proven) */
    double device_ms;              /* HIP-event time of the search kernel(s) of this call (HNSW: the beam-search
                                      kernel alone; flat: distance+select+finish kernels) */
} hvx_stats;

/* per-query counters, same order as the reference's golden test (index.rs:2396-2399) */
typedef struct hvx_query_stats {
    uint32_t expansion_steps, neighbors_examined, vectors_loaded, distance_computations;
} hvx_query_stats;

/*
 * Import a read-only index image (replaces VectorMemoryStore hydration, memory_store.rs:97-105,
 * and every row read under VectorIndex::search: storage.rs:1713-2066).
 *   node_ids      [n] external ids, strictly ascending
 *   vectors       [n][dim] row-major f32, host memory (or memory already on a device); validated like decode_item_borrowed
 *                 (mod.rs:889-949) ONCE here instead of on every fetch
 *   l0_offsets    [n+1], l0_neighbors: CSR of layer-0 rows over external ids (ids sorted per row:
 *                 values/vectors.rs:97-111)
 *   level         [n] top layer of each node (0 = layer 0 only); may be NULL when max_layer==0
 *   up_offsets    [rows+1], up_neighbors: CSR of upper rows; rows ordered by node, then layer 1..level
 */
int hvx_index_import(const hvx_index_desc *desc, const uint64_t *node_ids, const float *vectors,
                     const uint64_t *l0_offsets, const uint64_t *l0_neighbors, const uint16_t *level,
                     const uint64_t *up_offsets, const uint64_t *up_neighbors, hvx_index **out);
/* The same import with spare capacity (round 6): room for reserve_rows more nodes and reserve_upper_rows more upper-layer rows, so that
 * hvx_index_insert_batch / appending upserts can grow a HYDRATED image (f32 or bf16 rows; hvx_index_build takes its reserve from
 * hvx_build_params).  Declare m / m0 in the descriptor: the neighbour rows are sized for them. */
int hvx_index_import_reserve(const hvx_index_desc *desc, const uint64_t *node_ids, const float *vectors,
                             const uint64_t *l0_offsets, const uint64_t *l0_neighbors, const uint16_t *level,
                             const uint64_t *up_offsets, const uint64_t *up_neighbors, uint64_t reserve_rows, uint64_t reserve_upper_rows,
                             hvx_index **out);
void hvx_index_free(hvx_index *);
int hvx_index_sync(const hvx_index *);
/* the HIP stream (hipStream_t) search kernels are enqueued on */
void *hvx_index_stream(const hvx_index *);
/* The f32 values the index holds for rows [row0, row0 + n) (row = position in node-id order), into device memory
 * [n][dim]: the rows themselves (f32), the rounded values (bf16) or fl32(scale x decode(code)) (fp8) -- exactly what every
 * distance of this index is computed on; lets a host (or a checker) score the stored values.  Synchronises. */
int hvx_index_read_rows_device(const hvx_index *, uint64_t row0, uint64_t n, float *d_out);
/* enqueue on a caller-owned stream instead (e.g. the host runtime's current stream, so that
 * collectives and searches order without host synchronisation); NULL restores the index's own. */
int hvx_index_set_stream(hvx_index *, void *hip_stream);
/*
 * Execution lanes.  The reference's facade is called concurrently from many tokio tasks on one immutable index
 * (read_index.rs:83-102, access/search/storage.rs:140-163).  hvx_index_fork returns another handle on the SAME device image
 * (rows, graph, ids, headers, SimHash rows are shared, not copied) with its own HIP stream, events, per-batch scratch and
 * lock: calls on different handles of one image run concurrently on the device -- a host thread (or a pipelined caller
 * that keeps several batches in flight) uses one handle per lane.  The image lives until the last handle referencing it
 * is freed, in any order.  Attach SimHash rows (hvx_index_set_simhash) to the imported handle BEFORE forking.
 */
int hvx_index_fork(const hvx_index *, hvx_index **out);
/* Execution-path selectors of ONE handle.  Every setting returns the same results; they exist so that tests and A/B
 * measurements can pin a code path.  (The library reads no environment variables: tuning-only switches -- kernel
 * ablation, phase profiling, experimental tile builds -- exist only in builds made with `make TUNING=1`.)  Forks inherit
 * the options their parent has at fork time.  value 0 = the library's own choice. */
enum hvx_option {
    HVX_OPT_HNSW_GENERAL_KERNEL = 0, /* 1: strict searches over f32 rows take the general 4-wavefront kernel */
    HVX_OPT_WAVE_LOG2CAP = 1,        /* 7..15: log2 slots of the wave kernel's LDS visited table (small => HBM-bitmap spill path) */
    HVX_OPT_FLAT_FORCE_VALU = 2,     /* 1: f32 exact scans stay on the reference-order VALU kernel (no matrix-core candidates) */
    HVX_OPT_FLAT_FIRST_CHUNK = 3,    /* rows of the first (unfiltered) chunk of the matrix-core scan, >= 1024 */
    HVX_OPT_FLAT_NO_TILE = 4,        /* 1: matrix-core scans stay on the 128 x 128 kernel */
    HVX_OPT_FLAT_NO_FILTER = 5,      /* 1: no filtered epilogue (every chunk writes its score matrix) */
    HVX_OPT_FLAT_NO_FAST = 6,        /* 1: start with the full hi + lo split */
    HVX_OPT_FLAT_TILE_BUILD = 7,     /* large-tile kernel: 0 = two 256-thread workgroups per CU (256 x 128 tiles), fp8 rows on the MX-scaled fp8
                                        matrix instruction (query as two e4m3 pieces); 1 = one 512-thread workgroup (256 x 256); 2 = 512 threads
                                        role-split; 3 = as 0; 4 = as 0 but fp8 codes widened to bf16 (the round-2 build: the A/B switch) */
    HVX_OPT_FLAT_NO_SMALLB = 8,      /* 1: batches of <= 128 queries do not take the one-pass streaming kernels; 2: batches of <= 32 queries stay on
                                        the register-fragment build (rows straight into MFMA operands) instead of the LDS-ring build */
    HVX_OPT_HNSW_PAIR = 9,           /* owner / gatherer kernel (an owner wavefront + 1 or 3 gatherer wavefronts per query): 0 = when the
                                        handle runs one query per SIMD (hvx_index_set_occupancy(1): one batch in flight), 1 = never,
                                        2 = always where it is built, 3 = always, with ONE gatherer even where three are built */
    HVX_OPT_DELETE_SEQUENTIAL = 10,  /* 0: one fused launch per relinked row (round 6); 1: every source in the one-wavefront kernel; 2: two launches per row
                                        (rows of at most 34 ids, degree limits <= 32) -- all three produce the reference's rows */
    HVX_OPT_RESTRICTED_DIRECT = 11,  /* restricted exact scans of k <= 64: 0 = the one-launch reference-order kernel (csrc/hvx_restricted_exact.hip) where it is
                                        the cheaper one (b x candidates x dim <= 2^31), 1 = never, 2 = always */
    HVX_OPT_RESTRICTED_EXACT_MIB = 12, /* device plan of HVX_RESTRICTED_AUTO: candidate sets whose rows take at most this many MiB are scanned
                                        exactly; 0 = 1024 */
    HVX_OPT_FLAT_NO_TAIL = 13,       /* 1: small-batch matrix-core scans keep the selection / re-rank / certificate steps of rounds 3-5 instead of the
                                        one-launch exact tail (csrc/hvx_flat_tail.hip) */
    HVX_OPT_COUNT = 14
};
int hvx_index_set_option(hvx_index *, uint32_t option, uint32_t value);
/* which kernels the handle's last exact scan ran (bit flags) */
enum hvx_scan_path {
    HVX_PATH_VALU = 1, HVX_PATH_MFMA_128 = 2, HVX_PATH_TILE_256 = 4, HVX_PATH_FILTERED = 8, HVX_PATH_FULL_SPLIT = 16,
    HVX_PATH_VALU_FALLBACK_QUERIES = 32, HVX_PATH_WIDENED = 64,
    HVX_PATH_SMALL_BATCH = 256,       /* the one-pass small-batch kernel (hvx_flat_smallb.hip) produced the candidates */
    HVX_PATH_PAIR_OVERFLOW_REPEAT = 128, /* a filtered slice let more pairs through than its buffer holds: the scan was repeated unfiltered */
    HVX_PATH_DIRECT = 512,            /* the one-launch reference-order scan of a restricted candidate set (hvx_restricted_exact.hip) */
    HVX_PATH_EXACT_TAIL = 1024        /* small-batch candidates turned into the exact answer by the one-launch tail (hvx_flat_tail.hip): no certificate */
};
uint32_t hvx_index_last_scan_path(const hvx_index *);
/* HNSW kernel build used by this handle: 1 (default) = one query per SIMD with the SIMD's whole register file (lowest
 * latency of a lone batch); 2 = two queries per SIMD, half the registers / LDS each -- higher throughput when >= 2 batches
 * are in flight on the device (lanes).  Results are identical.  Shapes without a 2-per-SIMD build run the default. */
int hvx_index_set_occupancy(hvx_index *, uint32_t queries_per_simd);
/* Asynchronous kernel timing for pipelined callers: after _begin(capacity), every HNSW search call that is NOT asked
 * for hvx_stats brackets its search kernel with the next HIP-event pair of a ring instead of synchronising;
 * _collect waits for the stream once and returns the per-call kernel durations (ms) in call order, then disarms. */
int hvx_index_timing_begin(hvx_index *, uint32_t capacity);
int hvx_index_timing_collect(hvx_index *, float *out_ms, uint32_t cap, uint32_t *out_n);
/* Wave-slot residency evidence for the same launches (call before _collect): out[launch][rows_per_launch][2] = the device's
 * constant-rate 100 MHz clock when the query's wavefront started / finished (rows beyond a launch's batch are 0).  Only the
 * one-wavefront-per-query HNSW kernel writes them. */
int hvx_index_wave_clocks(hvx_index *, uint64_t *out, uint32_t cap_launches, uint32_t rows_per_launch, uint32_t *out_n);

/*
 * ValidatedVectorReadIndex::search (read_index.rs:83-92) -> VectorIndex::search (index.rs:1578-1587)
 * -> SearchSession::run (search.rs:1101-1230), strict-exhaustive arm, for b queries at once.
 * Results per query sorted (score asc, id asc), count <= min(k, population).
 * out_status: per-query hvx_status, nullable; when NULL the first failing query's status is returned (the rows of
 * the queries that did succeed have been written by then; a rejected query's row is left untouched, count 0).
 * Limits of this build: ef <= 992; the one-wavefront-per-query kernel serves dim in {128,256,512,768,1024,1536} with
 * rows <= 64 ids and ef <= 352, the general kernel everything else (any dim / metric, rows <= 128 ids).
 * A query whose beam evicted equal-score candidates beyond the 32-entry slack is re-run with a wider beam; only if
 * that is not possible is it counted in hvx_stats.tie_overflow_queries.
 */
int hvx_search_batch(const hvx_index *, const float *queries /*[b][dim]*/, uint32_t b, uint32_t k,
                     uint32_t ef, uint64_t *out_ids /*[b][k]*/, float *out_scores /*[b][k]*/,
                     uint32_t *out_counts /*[b]*/, uint32_t *out_status /*[b] nullable*/,
                     hvx_stats *stats /*nullable*/);
/* same, queries/outputs already in HBM; d_query_stats nullable; validation runs on the device */
int hvx_search_batch_device(const hvx_index *, const float *d_queries, uint32_t b, uint32_t k,
                            uint32_t ef, uint64_t *d_out_ids, float *d_out_scores,
                            uint32_t *d_out_counts, uint32_t *d_out_status,
                            hvx_query_stats *d_query_stats, hvx_stats *stats);

/*
 * Full SearchParams (mod.rs:410-621): the NON-strict arms of the layer-0 search -- SimHash threshold filter,
 * pre / post sampling, adaptive bypass (search.rs:595-829, policy.rs, randomness.rs).
 * `SearchParams::new(k)` -- what the query path runs (access/search/storage.rs:140-141) -- is
 * hvx_search_params_default(): ef = max(k, 100), SimHashMode::Adaptive, bypass tuning (24, 4, 0.12, 3).
 */
typedef enum hvx_simhash_mode { HVX_SIMHASH_ALWAYS = 0, HVX_SIMHASH_ADAPTIVE = 1, HVX_SIMHASH_OFF = 2 } hvx_simhash_mode;

typedef struct hvx_search_params {
    uint32_t k, ef;                              /* ResultCount, SearchBeamWidth (ef >= k) */
    uint32_t simhash_mode;                       /* hvx_simhash_mode */
    float pre_simhash_sampling_ratio_override;   /* [0,1]; < 0 = None */
    uint32_t bypass_min_frontier;                /* with_simhash_bypass_tuning (mod.rs:555-583), all non-zero */
    uint32_t bypass_window_expansions;
    float bypass_min_filter_rate;
    uint32_t read_budget_multiplier;
    float simhash_sampling_ratio_override;       /* [0,1]; < 0 = None */
    float simhash_failure_prob_override;         /* (0,1); < 0 = None */
} hvx_search_params;
void hvx_search_params_default(hvx_search_params *, uint32_t k);

/* index-level knobs of VectorIndexConfig the params resolve against (values/vectors/metadata.rs:38-44;
 * defaults 43 / 0.8 / true / 0.1, SimHasher seed 42: mod.rs:313-329, generation.rs:25) */
typedef struct hvx_simhash_config {
    uint64_t seed;
    uint32_t simhash_threshold;                  /* <= 64 */
    float sampling_ratio;                        /* [0,1] */
    uint32_t adaptive_enabled;
    float adaptive_failure_prob;                 /* (0,1) */
    uint32_t resident_snapshot;                  /* 1 (default): the index stands in for a Ready resident store -- SimHash
                                                    lookups are not stable-view reads (memory_store.rs:329-335);
                                                    0: an uncached handle -- every row first seen by a query is one read
                                                    (memory_store.rs:338-347) and the read-budget bypass trigger is live */
} hvx_simhash_config;
void hvx_simhash_config_default(hvx_simhash_config *);

/* Attach the per-node SimHash rows ([0xF0][index_id][0x13] rows / order codes of the vector keys) to an imported
 * index: node_hashes [n] in node-id order (host memory), or NULL to compute them on the device with
 * SimHasher(dim, cfg->seed) from the stored f32 / bf16 rows (unaligned_vector/simhash.rs:263-291).  Required before
 * a search whose params need the query fingerprint (SearchParams::requires_query_simhash, mod.rs:546-552). */
int hvx_index_set_simhash(hvx_index *, const hvx_simhash_config *cfg, const uint64_t *node_hashes);
int hvx_index_get_simhash(const hvx_index *, uint64_t *out_node_hashes /*[n] host*/);

/* Host-only: the step table the kernels evaluate `adaptive_threshold` (policy.rs:577-599) with -- out_brk[t-1] = the largest
 * f32 delta whose threshold is >= t (-1: none), built with THIS host's libm; threshold(delta) = #{t : delta <= out_brk[t-1]}. */
int hvx_adaptive_threshold_table(uint32_t configured, float failure, float *out_brk /*[64]*/);

/* SearchStats fields of the non-strict arms (mod.rs:629-700), per query.  With hvx_simhash_config.resident_snapshot = 1
 * (default) SimHash lookups are not stable-view reads: txn_get_simhash_filter stays 0 and the read-budget bypass trigger
 * cannot fire -- exactly as for the reference with a Ready resident store; with 0 both follow the uncached handle. */
typedef struct hvx_adaptive_stats {
    uint32_t simhash_filtered, simhash_examined;
    uint32_t simhash_passed_before_sampling, simhash_passed_after_sampling;
    uint32_t pre_simhash_sample_kept, pre_simhash_sample_dropped;
    uint32_t simhash_bypass_expansions, simhash_skipped_candidates;
    uint32_t simhash_bypass_trigger_budget, simhash_bypass_trigger_low_yield;
    uint32_t active_simhash_threshold_sum, active_simhash_threshold_samples; /* avg_active_simhash_threshold */
    uint32_t effective_beam_len_sum, effective_beam_len_samples;             /* avg_effective_beam_len */
    uint32_t active_sampling_ratio_samples;                                  /* avg_active_sampling_ratio */
    uint32_t rng_words;                          /* u32 outputs drawn from the query's StdRng (not a reference field) */
    uint32_t txn_get_simhash_filter;             /* SimHash rows read for filtering (uncached accounting; 0 for a resident snapshot) */
    double active_sampling_ratio_sum;
} hvx_adaptive_stats;

/*
 * SearchSession::run (search.rs:1101-1230) under full SearchParams, b queries at once.  Strict-exhaustive
 * params (mode Off, no pre-sampling below 1.0) take the same kernel as hvx_search_batch.  Otherwise the query
 * SimHash is computed on the device, the query RNG is rand's StdRng seeded with
 * simhash ^ rotl(entry,17) ^ rotl(ef,7) (randomness.rs:104-120), and every decision epoch follows policy.rs.
 * Results, SearchStats counters and the number of RNG words drawn equal the reference CPU path's for the same
 * (index, query, params) -- with one caveat: the two `choose_index` fallback sites (search.rs:667,818) use
 * rand's `random_range`, for which the reference holds no known answer (restated, parity unpinned).
 * query_stats / adaptive_stats: per query, host (…_device: device) memory, nullable.
 */
int hvx_search_batch_params(const hvx_index *, const float *queries, uint32_t b, const hvx_search_params *params,
                            uint64_t *out_ids, float *out_scores, uint32_t *out_counts, uint32_t *out_status,
                            hvx_stats *stats, hvx_query_stats *query_stats, hvx_adaptive_stats *adaptive_stats);
int hvx_search_batch_params_device(const hvx_index *, const float *d_queries, uint32_t b, const hvx_search_params *params,
                                   uint64_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts,
                                   uint32_t *d_out_status, hvx_query_stats *d_query_stats,
                                   hvx_adaptive_stats *d_adaptive_stats, hvx_stats *stats);

/*
 * Exact scan over all rows -- the reference has no general flat operator; semantics are
 * restricted_exact_scan (restricted.rs:753-835) with allowed = every id and no 256-id admission cap.
 */
int hvx_flat_search_batch(const hvx_index *, const float *queries, uint32_t b, uint32_t k,
                          uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                          uint32_t *out_status, hvx_stats *stats);
int hvx_flat_search_batch_device(const hvx_index *, const float *d_queries, uint32_t b, uint32_t k,
                                 uint64_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts,
                                 uint32_t *d_out_status, hvx_stats *stats);

/*
 * ValidatedVectorReadIndex::search_restricted (read_index.rs:93-102) -> restricted.rs:466-613.
 * allowed_ids/allowed_offsets: per-query candidate id lists (RestrictedVectorCandidates::from_ids,
 * restricted.rs:356-371: deduped, <= 1,000,000).  allowed_offsets==NULL => one list of n_allowed ids
 * shared by all queries.
 */
int hvx_search_restricted_batch(const hvx_index *, const float *queries, uint32_t b, uint32_t k,
                                uint32_t ef, const uint64_t *allowed_ids,
                                const uint64_t *allowed_offsets /*[b+1] or NULL*/, uint64_t n_allowed,
                                uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                                uint32_t *out_status, hvx_stats *stats);

/*
 * The same with the reference's execution plan (restricted.rs:426-453) and its observability (RestrictedSearchStats,
 * restricted.rs:147-166).  strategy AUTO = the reference's planner: candidate sets of <= 256 ids AND <= 4 MiB of f32 payload
 * take restricted_exact_scan (restricted.rs:753-835), everything else restricted_filter_aware_search (restricted.rs:837-1148):
 * seeds = deterministic sample of the candidate set (:321-342) + SimHash-directory windows around the query's order code
 * (:866-923) + the entry point; an ACORN-style walk over layer-0 rows in which members are scored and non-members become
 * bridges ranked by SimHash Hamming distance; the six budgets of FilteredGraphBudgets (:230-259).  One workgroup per query
 * (csrc/hvx_restricted_walk.hip: 1 024 threads for batches of <= 256 queries, 256 threads above); ids, score bits, every counter
 * below and the termination reason equal the reference CPU path's.  Needs the SimHash rows (hvx_index_set_simhash; missing =>
 * HVX_ERR_INVARIANT like the reference's missing_simhash_error) and f32 or bf16 rows (bf16: the reference's arithmetic on the
 * stored, rounded values; fp8 rows take the exact scan); strategy EXACT forces the device's exact gathered scan for any
 * candidate-set size (what hvx_search_restricted_batch does), FILTERED forces the walk (the reference's tests call it that way).
 * An AUTO plan (no explicit budgets) that the walk of this build cannot run -- the reference's own budget rule gives
 * bridge_rows = 8 x max(1.5 ef, 4 k) > 9 600 for k >= 301 or ef > 800; bf16 / fp8 rows; neighbour rows beyond the walk's width
 * -- is answered by the exact gathered scan (a superset of the walk's answer in recall; hvx_restricted_stats.strategy says
 * EXACT), never by an error: the reference serves k up to 800 (MAX_RESTRICTED_RESULT_COUNT, restricted.rs:55).
 */
enum hvx_restricted_strategy { HVX_RESTRICTED_AUTO = 0, HVX_RESTRICTED_EXACT = 1, HVX_RESTRICTED_FILTERED = 2, HVX_RESTRICTED_REFERENCE_PLAN = 3 };
/* RestrictedSearchTermination (restricted.rs:131-143) */
enum hvx_restricted_termination {
    HVX_TERM_NONE = 0, HVX_TERM_EXHAUSTED = 1, HVX_TERM_BEAM_COMPLETE = 2, HVX_TERM_ROUTING_BUDGET = 3,
    HVX_TERM_BRIDGE_BUDGET = 4, HVX_TERM_VECTOR_BUDGET = 5
};
typedef struct hvx_restricted_params {
    uint32_t k, ef;              /* requested result count (clamped to the candidate count, then <= 800) and beam width */
    uint32_t strategy;           /* hvx_restricted_strategy */
    uint32_t beam_percent;       /* filtered beam = ef * percent / 100; 0 = FILTERED_BEAM_PERCENT (150) */
    uint32_t directory_enabled;  /* VectorIndex::simhash_directory_enabled(): seed from the SimHash directory */
    uint32_t explicit_budgets;   /* != 0: the six budgets below instead of FilteredGraphBudgets::with_beam_percent (how the
                                    reference's tests drive restricted_filter_aware_search); limits of this build:
                                    vector_payloads <= 1024, sampled_seeds / directory_seeds <= 1024, bridge_rows <= 9 600 */
    uint32_t ef_filtered, routing_rows, bridge_rows, vector_payloads, sampled_seeds, directory_seeds;
} hvx_restricted_params;
void hvx_restricted_params_default(hvx_restricted_params *, uint32_t k, uint32_t ef); /* AUTO, 150 %, directory on */
/* RestrictedSearchStats (restricted.rs:147-166), per query.  The two *_multi_get_calls counters that depend on the
 * storage engine's chunking (simhash_, vector_) are not reproduced. */
typedef struct hvx_restricted_stats {
    uint32_t strategy, termination; /* hvx_restricted_strategy that ran (EXACT / FILTERED; 0 = no plan: empty set, rejected query), hvx_restricted_termination */
    uint64_t ef_filtered, directory_scan_calls, directory_rows, directory_decoded_bytes, directory_hits, simhash_row_requests,
        companion_row_requests, routing_rows, bridge_rows, bridge_frontier_pushes, neighbor_multi_get_calls,
        vector_payload_requests, vector_bytes, distance_computations;
} hvx_restricted_stats;
int hvx_search_restricted_batch_params(const hvx_index *, const float *queries, uint32_t b, const hvx_restricted_params *params,
                                       const uint64_t *allowed_ids, const uint64_t *allowed_offsets /*[b+1] or NULL*/,
                                       uint64_t n_allowed, uint64_t *out_ids /*[b][k]*/, float *out_scores, uint32_t *out_counts,
                                       uint32_t *out_status /*nullable*/, hvx_restricted_stats *out_restricted_stats /*[b] nullable*/,
                                       hvx_stats *stats /*nullable*/);

/* Device-resident surface of the prefiltered branch, exact strategy (csrc/hvx_restricted_exact.hip): b queries in HBM, query q with its OWN
 * candidate ids d_allowed_ids[q * stride .. + d_lens[q]) -- external ids in any order; duplicates count once (RestrictedVectorCandidates is a
 * set, restricted.rs:303-371), ids that hold no (live) vector are skipped (:615-659), an empty list answers with nothing and status 0 before
 * the query is validated (:539-541).  k <= 64, f32 / bf16 rows.  ONE launch for the batch; nothing is synchronised; outputs may be HBM or
 * mapped host rows.  max_len = the longest list (<= stride). */
int hvx_search_restricted_lists_device(const hvx_index *, const float *d_queries, uint32_t b, uint32_t k, const uint64_t *d_allowed_ids,
                                       uint32_t stride, const uint32_t *d_lens, uint32_t max_len, uint64_t *d_out_ids /*[b][k]*/,
                                       float *d_out_scores, uint32_t *d_out_counts, uint32_t *d_out_status /*nullable*/);

/* k-way merge of per-shard results by Candidate order (model.rs:55-61); inputs in HBM, laid out
 * [g][b][k] as an all-gather over shards delivers them. */
int hvx_merge_topk_device(const hvx_index *, uint32_t g, uint32_t b, uint32_t k, const uint64_t *d_ids,
                          const float *d_scores, const uint32_t *d_counts, uint64_t *d_out_ids,
                          float *d_out_scores, uint32_t *d_out_counts);
/* The same merge over ONE packed exchange buffer (one all-gather per step instead of three): rank r's payload starts at
 * r * hvx_topk_payload_bytes(b, k) and holds ids [b][k] u64, scores [b][k] f32, counts [b] u32 in that order.  A search
 * can write its outputs straight into the local payload (the three sub-arrays are ordinary device pointers). */
size_t hvx_topk_payload_bytes(uint32_t b, uint32_t k);
int hvx_merge_topk_packed_device(const hvx_index *, uint32_t g, uint32_t b, uint32_t k, const void *d_packed /*[g][payload]*/,
                                 uint64_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts);

/*
 * The sharded search step in ONE call: every rank of a node holds one id-range shard and runs
 *     local search -> ncclAllGather of the packed per-shard top-k (RCCL over xGMI) -> merge by Candidate order
 * on its shard's stream, without host synchronisation.  Bootstrap as NCCL does: rank 0 obtains a 128-byte unique id
 * (hvx_shard_group_unique_id), the host distributes it (any transport), every rank calls hvx_shard_group_init with the
 * same id; then all ranks call hvx_shard_group_search_batch_device in the same order with the same (b, k).  Every rank
 * receives the merged top-k of the whole corpus.  RCCL is loaded at run time (librccl.so.1); a group of one rank needs
 * none.  The group borrows the shard handle (use one handle -- e.g. one hvx_index_fork lane -- per group).
 */
#define HVX_SHARD_UNIQUE_ID_BYTES 128
/* A rank whose LOCAL search fails still joins the exchange (the others would block in the all-gather otherwise): its payload
 * carries no results and, for every query, the status word HVX_SHARD_RANK_FAILED | its hvx_status.  The failing rank returns
 * that status from the call; on the other ranks the device-output steps return HVX_OK with d_out_status[q] = the marked word
 * and count 0 for every query (pass d_out_status when shards can fail independently), the host-output restricted step
 * returns the failed shard's status on EVERY rank. */
#define HVX_SHARD_RANK_FAILED 0x100u
typedef struct hvx_shard_group hvx_shard_group;
int hvx_shard_group_unique_id(uint8_t *out /*[HVX_SHARD_UNIQUE_ID_BYTES]*/);
int hvx_shard_group_init(hvx_index *local_shard, const uint8_t *unique_id, uint32_t rank, uint32_t world, uint32_t max_batch, uint32_t max_k,
                         hvx_shard_group **out);
/* Execution lanes of ONE rank (hvx_index_fork handles of its shard) share the rank's communicator: hvx_shard_group_attach gives
 * `lane` a group of its own (own payload buffers) on `primary`'s communicator.  The all-gathers of all groups that share a
 * communicator run on one dedicated exchange stream in the order the steps are issued: as with any NCCL communicator, every rank
 * must issue its steps in the same order (round-robin over the lanes from one thread, as bench.py does).  Free the groups in any
 * order; the communicator goes with the last one. */
int hvx_shard_group_attach(hvx_shard_group *primary, hvx_index *lane, hvx_shard_group **out);
/* Every step below is  local search -> ONE all-gather of the packed per-shard top-k + per-query status -> merge by Candidate
 * order.  d_out_status (nullable): per query, the element-wise maximum of the ranks' statuses -- a NaN / zero-norm / oversized
 * query is rejected by every shard's validation (search.rs:1120-1125 InvalidVectorComponent ...) and comes back with count 0. */
int hvx_shard_group_search_batch_device(hvx_shard_group *, const float *d_queries, uint32_t b, uint32_t k, uint32_t ef,
                                        uint64_t *d_out_ids /*[b][k]*/, float *d_out_scores, uint32_t *d_out_counts, uint32_t *d_out_status);
/* full SearchParams per shard (hvx_search_batch_params_device: the production default and the other non-strict arms) */
int hvx_shard_group_search_batch_params_device(hvx_shard_group *, const float *d_queries, uint32_t b, const hvx_search_params *params,
                                               uint64_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts, uint32_t *d_out_status);
/* exact scan of every shard (BASELINE configs[4]: the batched-query x corpus scan over fp8 / bf16 / f32 rows, sharded by id range) */
int hvx_shard_group_flat_search_batch_device(hvx_shard_group *, const float *d_queries, uint32_t b, uint32_t k, uint64_t *d_out_ids,
                                             float *d_out_scores, uint32_t *d_out_counts, uint32_t *d_out_status);
/* restricted search: every rank passes the SAME candidate id list and keeps the ids of its own range
 * [desc.shard_id_lo, desc.shard_id_hi]; host buffers like hvx_search_restricted_batch_params; the set-wide limits
 * (<= 1 000 000 ids, k clamped to the candidate count then <= 800) apply to the whole list; each shard plans its own slice. */
int hvx_shard_group_search_restricted_batch(hvx_shard_group *, const float *queries, uint32_t b, const hvx_restricted_params *params,
                                            const uint64_t *allowed_ids, uint64_t n_allowed, uint64_t *out_ids /*[b][params->k]*/,
                                            float *out_scores, uint32_t *out_counts, uint32_t *out_status /*nullable*/);
void hvx_shard_group_free(hvx_shard_group *);
/* A sharded step is enqueued without a host wait; a peer rank that never joins the collective would park this rank's stream -- and a host
 * blocked in hipStreamSynchronize -- for ever.  hvx_shard_group_wait bounds it: HVX_OK once everything the group's last step enqueued has
 * drained, HVX_ERR_TIMEOUT after timeout_ms (the host then calls hvx_shard_group_abort = ncclCommAbort: the pending collective is
 * cancelled, later steps of the groups that share the communicator fail with HVX_ERR_DEVICE, and the host re-forms the group). */
int hvx_shard_group_wait(hvx_shard_group *, uint32_t timeout_ms);
int hvx_shard_group_abort(hvx_shard_group *);
/* NCCL_VERSION_CODE of the RCCL library bound at run time (e.g. 22203), 0 when none could be loaded or it was refused: the
 * ncclUniqueId / ncclCommInitRank declarations this library binds are restated for the NCCL 2.x ABI and checked through
 * ncclGetVersion before the first collective (crates/db has no counterpart: sharding across GPUs is this library's own seam). */
int hvx_shard_rccl_version(void);

/*
 * Graph prefilter (crates/graph-algorithms/src/model.rs:370-417 Csr; algorithms/traversal.rs:197-318
 * breadth_first; crates/db/src/execution/interpreter/access/expand.rs:16-80 one-hop expand).
 */
int hvx_csr_import(uint64_t n_nodes, uint64_t n_edges, const uint64_t *out_offsets /*[n+1]*/,
                   const uint64_t *out_targets /*[e]*/, const uint32_t *edge_labels /*[e] or NULL*/,
                   int32_t device, hvx_csr **out);
void hvx_csr_free(hvx_csr *);
/* Graph::traverse, BreadthFirst (traversal.rs:216-261): multi-seed BFS with depth cap, direction,
 * edge-label allow-set (n_labels==0 => every label) and hub policy (hub_degree>0 => non-seed nodes
 * whose total degree >= hub_degree are emitted but not expanded, :311-318).  Writes a bitmap (1 bit
 * per node, n_nodes bits rounded up to 64) of every visited node (seeds included iff
 * include_seeds!=0) and, when out_depth!=NULL, each node's BFS depth (UINT32_MAX = unreached).
 * Unknown seeds => HVX_ERR_INVARIANT (GraphError::UnknownNode). */
int hvx_traverse_filter(const hvx_csr *, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth,
                        uint32_t direction, const uint32_t *allowed_label_ids, uint32_t n_labels,
                        uint32_t hub_degree, uint32_t include_seeds, uint64_t *out_bitmap_words,
                        uint32_t *out_depth);
/* Graph::traverse, BreadthFirst, WITH its order (traversal.rs:216-261: `visits` in discovery order, one discovery edge per
 * non-seed visit).  out_nodes / out_depths [capacity]: the visits, seeds first in their given order (duplicates collapse);
 * out_parents[i] / out_edges[i] / out_against[i] describe the discovery edge of visit i: the node whose arc reached it, the
 * index of the stored edge in the out_targets array handed to hvx_csr_import, and whether the edge was followed against its
 * stored direction (TraversedEdge::traversal_direction); UINT64_MAX / 0 for seeds.  Arc order as model.rs:635-725 (rows by
 * neighbour; direction Both merges the outgoing and incoming rows by neighbour, the outgoing arc first on equal
 * neighbours -- the reference breaks that tie by graphify key and edge id, which only decides WHICH of two edges to the
 * same node is reported).  Needs outgoing rows sorted by target (HVX_ERR_UNSUPPORTED otherwise) and < 2^31 edges.
 * *out_count = visits; more than `capacity` => HVX_ERR_INVARIANT with *out_count set.  Any output array may be NULL. */
int hvx_traverse_ordered(const hvx_csr *, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth, uint32_t direction,
                         const uint32_t *allowed_label_ids, uint32_t n_labels, uint32_t hub_degree, uint64_t capacity,
                         uint64_t *out_nodes, uint32_t *out_depths, uint64_t *out_parents, uint64_t *out_edges,
                         uint32_t *out_against, uint64_t *out_count);
/* Graph::traverse with TraversalStrategy::DepthFirst (traversal.rs:263-309): a node is marked when it is scheduled and recorded
 * when it is popped; one dependent chain of stack pops, so it runs on the HOST over a mirror of the device CSR fetched on first
 * use.  Same outputs, arc order, limits and statuses as hvx_traverse_ordered. */
int hvx_traverse_dfs(const hvx_csr *, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth, uint32_t direction,
                     const uint32_t *allowed_label_ids, uint32_t n_labels, uint32_t hub_degree, uint64_t capacity,
                     uint64_t *out_nodes, uint32_t *out_depths, uint64_t *out_parents, uint64_t *out_edges,
                     uint32_t *out_against, uint64_t *out_count);
/* The same traversal entirely on the host, over caller-held arrays (no device, no hvx_csr): strategy 0 = BreadthFirst,
 * 1 = DepthFirst.  For hosts that keep their Csr in RAM, and as the CPU cross-check of the device's level-synchronous order. */
int hvx_traverse_host(uint64_t n_nodes, uint64_t n_edges, const uint64_t *out_offsets, const uint64_t *out_targets,
                      const uint32_t *edge_labels /*nullable*/, uint32_t strategy, const uint64_t *seeds, uint32_t n_seeds,
                      uint32_t max_depth, uint32_t direction, const uint32_t *allowed_label_ids, uint32_t n_labels,
                      uint32_t hub_degree, uint64_t capacity, uint64_t *out_nodes, uint32_t *out_depths, uint64_t *out_parents,
                      uint64_t *out_edges, uint32_t *out_against, uint64_t *out_count);
/* One interpreter `expand` hop (crates/db/src/execution/interpreter/access/expand.rs:16-80): the
 * union of the neighbours of every input row (an input row that is itself a neighbour of another
 * input row IS included) as the candidate bitmap handed to the restricted vector search. */
int hvx_expand_filter(const hvx_csr *, const uint64_t *rows, uint32_t n_rows, uint32_t direction,
                      const uint32_t *allowed_label_ids, uint32_t n_labels, uint64_t *out_bitmap_words);

/*
 * Fused graph prefilter + restricted kNN (BASELINE config #3: `where_()` / traversal filter, then vector_search):
 * hvx_expand_filter or hvx_traverse_filter followed by hvx_search_restricted_batch with the candidate bitmap kept ON
 * THE DEVICE -- the hop's bitmap is compacted into the scan's row list by two small kernels (ids without a vector are
 * omitted, restricted.rs:615-659), no candidate list crosses PCIe.  The CSR's node numbers are the index's node ids.
 * mode: HVX_PREFILTER_EXPAND (one hop from `seeds`, expand.rs:16-80; max_depth / hub_degree / include_seeds ignored)
 * or HVX_PREFILTER_TRAVERSE (Graph::traverse, traversal.rs:197-261).  Same limits and statuses as the two-call form
 * (<= 1 000 000 candidates, k <= 800); out_candidates receives the candidate population.
 */
enum { HVX_PREFILTER_EXPAND = 0, HVX_PREFILTER_TRAVERSE = 1 };
int hvx_prefilter_search_batch(const hvx_index *, const hvx_csr *, const float *queries, uint32_t b, uint32_t k, uint32_t ef,
                               uint32_t mode, const uint64_t *seeds, uint32_t n_seeds, uint32_t max_depth, uint32_t direction,
                               const uint32_t *allowed_label_ids, uint32_t n_labels, uint32_t hub_degree,
                               uint32_t include_seeds, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                               uint32_t *out_status /*nullable*/, uint64_t *out_candidates /*nullable*/, hvx_stats *stats);

/* The fused call under the reference's execution plan: `params` as for hvx_search_restricted_batch_params (k, ef, strategy,
 * beam, directory, budgets); the candidate bitmap, its compaction, the deterministic sample (ranks into the candidate
 * population) and the membership bitmap of the walk all stay on the device. */
int hvx_prefilter_search_batch_params(const hvx_index *, const hvx_csr *, const float *queries, uint32_t b,
                                      const hvx_restricted_params *params, uint32_t mode, const uint64_t *seeds, uint32_t n_seeds,
                                      uint32_t max_depth, uint32_t direction, const uint32_t *allowed_label_ids, uint32_t n_labels,
                                      uint32_t hub_degree, uint32_t include_seeds, uint64_t *out_ids, float *out_scores,
                                      uint32_t *out_counts, uint32_t *out_status /*nullable*/, uint64_t *out_candidates /*nullable*/,
                                      hvx_restricted_stats *out_restricted_stats /*[b] nullable*/, hvx_stats *stats);

/*
 * Batching operator: the reference calls ValidatedVectorReadIndex::search once per operator invocation
 * from many tokio tasks (access/search/storage.rs:140-163).  Concurrent single-query callers are coalesced into ONE
 * hvx_search_batch_params launch: a caller blocks in hvx_batcher_search until its rows are ready.  A batch is launched when it
 * is full (max_batch, 0 = the index's max_batch, at most 65 535) or when a DISPATCHER LANE IS FREE and the batch holds as many
 * queries as that lane EXPECTS: the size of the lane's previous batch (the callers of a batch that has just completed come back
 * in a burst: a lane that launched at the first of them would run a handful of queries beside the next lane's hundreds), or once
 * the batch's first query has waited max_wait_us (0 = 200 us).  A lane's first batch -- and a lone caller's every batch --
 * expects one query and is launched at once; when the load drops, one batch pays max_wait_us and the expectation follows it
 * down.  While every lane is busy the open batch simply keeps growing, so the batch size follows the load.  Result rows are written by the kernels straight into pinned host memory.
 * All callers of one batcher share `params`.  Thread-safe; results equal a direct batch call's.  A rejected query fails
 * alone.  hvx_batcher_free may be called while callers are blocked: they return HVX_ERR_INVARIANT.
 */
typedef struct hvx_batcher hvx_batcher;
int hvx_batcher_new(hvx_index *, const hvx_search_params *params, uint32_t max_batch, uint32_t max_wait_us, hvx_batcher **out);
/* the same with an explicit number of dispatcher lanes (1..8; 0 = 4): every lane is an hvx_index_fork of the index with its own
 * dispatcher thread, so batch i+1 is collected and launched while batch i runs.  Attach SimHash rows before creating it. */
int hvx_batcher_new_lanes(hvx_index *, const hvx_search_params *params, uint32_t max_batch, uint32_t max_wait_us, uint32_t lanes,
                          hvx_batcher **out);
void hvx_batcher_free(hvx_batcher *);
int hvx_batcher_search(hvx_batcher *, const float *query /*[dim]*/, uint64_t *out_ids /*[k]*/, float *out_scores /*[k]*/,
                       uint32_t *out_count);
/*
 * The non-blocking form (ValidatedVectorReadIndex::search is an `async fn` called from tokio tasks, read_index.rs:81-102: a task
 * must not park an OS thread for every query in flight).  hvx_batcher_submit claims a slot of the open batch and copies the query
 * into its pinned staging row -- no system call on the fast path -- and returns at once: HVX_OK + a ticket, or HVX_ERR_BUSY when
 * the open batch is full (a free lane closes it within microseconds: submit again after the next completion / yield).
 * hvx_batcher_poll tests the ticket: HVX_PENDING while its batch is in flight, else the rows are copied out and the ticket is
 * CONSUMED (HVX_OK, or the query's own rejection status); a ticket must be polled / waited to completion exactly once -- a batch
 * buffer is reused only after every one of its tickets has been consumed.  hvx_batcher_wait blocks up to timeout_us (0xFFFFFFFF =
 * for ever; HVX_PENDING on time-out, the ticket stays valid).  hvx_batcher_eventfd returns an eventfd (created on first use, owned
 * by the batcher, EFD_NONBLOCK) that receives one tick per completed batch: the host registers it with its reactor (tokio
 * AsyncFd / epoll), reads it when it fires and polls its outstanding tickets -- INTEGRATION.md 4c.  hvx_batcher_search is
 * submit + wait.
 */
typedef struct hvx_batcher_ticket {
    uint64_t sequence; /* batch the query rides in */
    uint32_t slot;     /* its row in that batch */
    uint32_t reserved;
} hvx_batcher_ticket;
int hvx_batcher_submit(hvx_batcher *, const float *query /*[dim]*/, hvx_batcher_ticket *out_ticket);
int hvx_batcher_poll(hvx_batcher *, const hvx_batcher_ticket *ticket, uint64_t *out_ids /*[k]*/, float *out_scores /*[k]*/, uint32_t *out_count);
int hvx_batcher_wait(hvx_batcher *, const hvx_batcher_ticket *ticket, uint32_t timeout_us, uint64_t *out_ids /*[k]*/, float *out_scores /*[k]*/,
                     uint32_t *out_count);
int hvx_batcher_eventfd(hvx_batcher *); /* -1 on failure */
/* The operator's OTHER branch (execution/interpreter/access/search/storage.rs:140-163: `Some(candidates) => index.search_restricted(..)`):
 * every caller brings its own candidate ids (a where_() / traversal filter evaluated by the host).  A batch of such callers is ONE launch of
 * the exact scan with per-query candidate lists -- 256 callers x 1 000 ids x 1536 floats: 0.26 ms of device time (profiles/r06d_*).
 * params: k <= 64; the plan must answer a max_ids_per_query-id set exactly (strategy EXACT, or AUTO within the device plan's limit); lists
 * beyond max_ids_per_query (0 = 4 096; pinned memory: lanes + 2 buffers x max_batch x max_ids_per_query x 8 B) are refused with
 * HVX_ERR_UNSUPPORTED -- the host sends those through hvx_search_restricted_batch_params.  Tickets of a restricted batcher are polled /
 * waited for / awaited through the eventfd like any other; hvx_batcher_search / _submit refuse a restricted batcher and vice versa. */
int hvx_batcher_new_restricted(hvx_index *, const hvx_restricted_params *params, uint32_t max_batch, uint32_t max_wait_us, uint32_t lanes,
                               uint32_t max_ids_per_query, hvx_batcher **out);
int hvx_batcher_search_restricted(hvx_batcher *, const float *query /*[dim]*/, const uint64_t *allowed_ids, uint32_t n_allowed,
                                  uint64_t *out_ids /*[k]*/, float *out_scores /*[k]*/, uint32_t *out_count);
int hvx_batcher_submit_restricted(hvx_batcher *, const float *query /*[dim]*/, const uint64_t *allowed_ids, uint32_t n_allowed,
                                  hvx_batcher_ticket *out_ticket);
/* after hvx_index_insert_batch on the handle the batcher was created from: the dispatcher lanes (forks of that handle) adopt the new
 * generation before the next batch each of them launches */
int hvx_batcher_refresh(hvx_batcher *);
int hvx_batcher_stats(const hvx_batcher *, uint64_t *batches, uint64_t *queries, uint64_t *full_batches);
/* where the dispatcher lanes' time went since creation, summed over the lanes (nanoseconds): asleep with nothing to do, watching
 * an open batch grow, waiting for a batch buffer's previous callers to take their rows, waiting for callers to finish copying
 * their query in, query copy + kernels + stream synchronise, handing the batch to its callers */
typedef struct hvx_batcher_times {
    uint64_t idle_ns, collect_ns, drain_ns, fill_ns, device_ns, wake_ns;
    uint32_t lanes, reserved;
} hvx_batcher_times;
int hvx_batcher_lane_times(const hvx_batcher *, hvx_batcher_times *out);

/*
 * SimHash projections (crates/db/src/search/vector/unaligned_vector/simhash.rs:123-178 SimHasher::new_with_seed,
 * :263-291 hash_from_slice; simhash.rs:44-59 order_code_from_simhash_bits, which keys the canonical vector rows
 * `[0xF1][index_id][0x02][order_code][node_id]`).  `vectors` is [n][dim] f32 in host or device memory.
 */
typedef struct hvx_simhasher hvx_simhasher;
int hvx_simhasher_new(uint32_t dim, uint64_t seed, int32_t device, hvx_simhasher **out);
void hvx_simhasher_free(hvx_simhasher *);
int hvx_simhash_batch(const hvx_simhasher *, const float *vectors, uint64_t n, uint64_t *out_bits /*[n] host*/);
uint64_t hvx_order_code_from_simhash_bits(uint64_t bits);

/* VectorComponentLimit::try_new (domain.rs:26-78): the inclusive |component| maximum the index validates rows and queries
 * against -- floor-to-f32 of sqrt(f32::MAX / (8 dim)) for squared-L2, f32::MAX / (4 dim) for Manhattan, +inf (no limit) for cosine */
float hvx_component_limit(uint32_t metric, uint32_t dim);

/*
 * Hydration from HelixDB's persisted rows (replaces VectorMemoryStore hydration, memory_store.rs:97-105).
 * Value codecs restated from crates/db/src/encoding/v1/values/vectors.rs:97-210 (layer-0 rows, tags 0x12 / 0x13, empty),
 * values/vectors/neighbors.rs:57-110 (upper rows), values/vectors/item.rs:34-60 (header f32 + dim f32, native-endian);
 * keys from keys/vectors.rs:23-50 (legacy or tenant-scoped, keys/tenant.rs:69-95); the rkyv metadata row through
 * hvx_decode_index_metadata / hvx_hydrator_set_metadata.
 */
typedef struct hvx_hydrator hvx_hydrator;
int hvx_decode_layer0_row(const uint8_t *value, size_t len, uint64_t *out_ids, uint32_t cap, uint32_t *out_count,
                          uint64_t *out_simhash, uint32_t *out_has_simhash);
int hvx_decode_upper_row(const uint8_t *value, size_t len, uint64_t *out_ids, uint32_t cap, uint32_t *out_count);
/* values/vectors/simhash.rs:39-61 (standalone SimHash row: exactly 8 bytes, little-endian) and values/vectors/entry.rs:25-47
 * (entry-candidate node row: its HNSW layer as exactly 2 big-endian bytes); non-exact lengths are rejected like the reference's */
int hvx_decode_simhash_row(const uint8_t *value, size_t len, uint64_t *out_bits);
int hvx_decode_entry_candidate_layer(const uint8_t *value, size_t len, uint32_t *out_layer);
/* keys/tenant.rs:69-95: length of the tenant envelope in front of a key (17 = [0xFD][tenant_id u128 BE], 0 = legacy namespace) */
uint32_t hvx_strip_tenant_envelope(const uint8_t *key, size_t len, uint64_t *tenant_hi, uint64_t *tenant_lo);
/* The index metadata row [0x03][0x03][index_id][0x01] (keys/vectors.rs:23-38): rkyv 0.8 archive of VectorIndexMetadata
 * (values/vectors/metadata.rs:22-62).  rkyv is not vendored in the reference and the reference holds no byte fixture of the row:
 * the layout is restated from rkyv 0.8's published format -- PARITY UNPINNED (csrc/hvx_hydrate.hip has the layout). */
typedef struct hvx_index_metadata {
    char index_name[256], property_name[256];
    uint64_t dimension, m, m0, ef_construction, simhash_threshold;
    float ml, sampling_ratio, adaptive_failure_prob;
    uint32_t adaptive_enabled, has_entry_point, max_layer;
    uint64_t entry_point, count;
} hvx_index_metadata;
int hvx_decode_index_metadata(const uint8_t *value, size_t len, hvx_index_metadata *out);
/* returns the key kind (0x01 index metadata, 0x02 canonical vector, 0x16 layer-0 neighbours, 0x11 upper neighbours) or 0;
 * tenant-scoped keys (a 17-byte envelope in front) are accepted */
uint32_t hvx_parse_vector_key(const uint8_t *key, size_t len, uint64_t *index_id, uint64_t *node_id, uint64_t *order_code,
                              uint32_t *layer);
int hvx_hydrator_new(uint32_t dim, uint32_t metric, hvx_hydrator **out);
void hvx_hydrator_free(hvx_hydrator *);
int hvx_hydrator_add_item(hvx_hydrator *, uint64_t node_id, const uint8_t *value, size_t len);
int hvx_hydrator_add_layer0_row(hvx_hydrator *, uint64_t node_id, const uint8_t *value, size_t len);
int hvx_hydrator_add_upper_row(hvx_hydrator *, uint64_t node_id, uint32_t layer, const uint8_t *value, size_t len);
int hvx_hydrator_set_entry(hvx_hydrator *, uint64_t entry_point, uint32_t max_layer);
/* entry point / top layer from the metadata row instead (checked against the hydrator's dimension) */
int hvx_hydrator_set_metadata(hvx_hydrator *, const uint8_t *value, size_t len);
/* `tmpl` supplies dtype, float_kernel, m, m0, device, max_batch; the rest comes from the collected rows */
int hvx_hydrator_finish(const hvx_hydrator *, const hvx_index_desc *tmpl, hvx_index **out);
/* ... with spare capacity (hvx_index_import_reserve): the image a serving host hydrates at start-up and then feeds its writes to */
int hvx_hydrator_finish_reserve(const hvx_hydrator *, const hvx_index_desc *tmpl, uint64_t reserve_rows, uint64_t reserve_upper_rows, hvx_index **out);

/*
 * GPU-assisted HNSW build: the reference's insert path (mutation.rs:642-895 insert_with_mutation_cache /
 * insert_hnsw: greedy descent, search_layer_beam per layer :904-1005, select_neighbors_heuristic :1072-1097 = select_diverse
 * mod.rs:809-856, add_bidirectional_link with prune + reverse-edge removal :1498-1583,1890-1908) over rows resident in HBM,
 * for batches of consecutive nodes.  `levels[i]` = the node's top layer (drawn by the host with select_layer_from_uniform,
 * mod.rs:776-796 -- the reference's draw is intentionally non-deterministic, randomness.rs:41-46).  Nodes are inserted in
 * node-id order.  With params->sequential != 0 every batch holds ONE node and the graph equals the reference's sequential
 * insertion row for row; otherwise a batch holds up to min(max_batch, inserted / batch_divisor) nodes that do not see each
 * other (a node above the current top layer is always inserted alone and becomes the entry point).  The result is an
 * ordinary searchable hvx_index; hvx_index_export_graph hands the rows back for the host to persist (values/vectors.rs).
 * Served shapes: f32 rows of any dimension, cosine / Euclidean / Manhattan, every summation tree of the reference's float kernels
 * (the unrolled kernels for cosine / Euclidean + AVX+FMA + dim in {128,256,512,768,1024,1536} + ef_construction <= 352, the GENERIC
 * build otherwise), m0 <= 32, ef_construction <= 800; desc->max_batch bounds the batch size.
 */
typedef struct hvx_build_params {
    uint32_t ef_construction; /* 0 => 200 (mod.rs:702-708) */
    uint32_t max_batch;       /* 0 => 2048 */
    uint32_t batch_divisor;   /* 0 => 32: batch <= nodes already inserted / divisor */
    uint32_t sequential;      /* 1 => one node per batch: the reference's insertion order exactly */
    uint32_t link_mode;       /* batched link step: 0 => one workgroup per link with the prune evaluated from LDS whenever the rows fit
                                 (Mmax + 2 rows of ld floats <= 160 KB), 1 => one wavefront per node (links one after the other).
                                 One node at a time (sequential = 1, upserts, promotions): 0 => its select and its links as two many-workgroup
                                 steps with every prune's distance matrix evaluated up front (round 6), 1 => the one-wavefront kernels */
    uint32_t scatter;         /* 1 => batched mode inserts in the order (i * stride) mod n, stride ~ 0.618 n coprime with n, instead of
                                 id order.  For rows whose order follows the data (dumps sorted by topic, indexes hydrated in key order):
                                 the nodes of one batch do not see each other, and consecutive rows of such data are each other's
                                 nearest neighbours.  Ignored by sequential mode (= the reference's insertion order). */
    uint64_t reserve_rows;       /* hvx_index_build: room for this many rows beyond desc->n (hvx_index_insert_batch appends into it) */
    uint64_t reserve_upper_rows; /* ... and for this many upper-layer rows (sum of the levels of the nodes appended later) */
} hvx_build_params;
typedef struct hvx_build_stats {
    uint64_t nodes, batches, single_node_batches;
} hvx_build_stats;
void hvx_build_params_default(hvx_build_params *);
int hvx_index_build(const hvx_index_desc *desc, const uint64_t *node_ids, const float *vectors /*host or device*/,
                    const uint16_t *levels /*[n] or NULL = all layer 0*/, const hvx_build_params *params, hvx_index **out,
                    hvx_build_stats *stats /*nullable*/);
/*
 * Incremental insert into a LIVE image (VectorIndex::insert -> insert_hnsw, mutation.rs:642-895): `count` new nodes -- ids ascending
 * and above every id of the image, f32 vectors (host or device memory), levels as hvx_index_build takes them -- are appended to
 * the spare capacity of an image built with hvx_build_params.reserve_rows or hydrated with hvx_index_import_reserve
 * (HVX_ERR_CANDIDATE_LIMIT when they do not fit; bf16 images: the vectors are rounded first and linked one node at a time),
 * validated like imported rows (an invalid vector fails the call before anything changes), given headers and -- when SimHash
 * rows are attached -- SimHash rows, and linked into the graph by the loop hvx_index_build runs: params->sequential = 1 is the
 * reference's insertion row for row (tests/test_gpu_build.py), the batched mode trades that for throughput exactly as the build
 * does.  params = NULL: the defaults.  The call goes through the handle that owns the image; when it returns the new generation
 * is visible on that handle (hvx_index_visible_seq + 1, hvx_index_rows grown) and every fork adopts it with hvx_index_refresh --
 * the host's registry decides when (read_index.rs:55-61: a resident copy serves a snapshot only when its sequence matches).
 * Forks may keep searching during the call: they see their own generation's entry point, and neighbour rows as stale-or-current,
 * never torn.  A failure after validation leaves the image partially linked: discard the handle and hydrate again.
 */
int hvx_index_insert_batch(hvx_index *, const uint64_t *node_ids /*[count]*/, const float *vectors /*[count][dim]*/,
                           const uint16_t *levels /*[count] or NULL*/, uint32_t count, const hvx_build_params *params /*nullable*/,
                           hvx_build_stats *stats /*nullable*/);
/*
 * Delete of nodes from a LIVE image (VectorIndex::delete -> stage_delete_with_metadata, mutation.rs:1606-1774; delete_from_layer
 * :1819-1888; relink_neighbor :1916-2055): for every id, in the order given, every row that holds the node loses the edge, the
 * affected sources are relinked from their joint neighbourhood (the Mmax closest candidates, select_diverse + backfill beyond Mmax, the
 * reciprocal rows of new neighbours), the node's own rows go, and a deleted entry point is replaced by the best remaining entry
 * candidate (highest layer, then smallest id).  Rows equal the oracle's sequential deletes row for row (tests/test_gpu_delete.py).
 * An id that is not in the image (or was deleted before) succeeds and changes nothing (index.rs:2263, 2294-2295; counted in
 * stats->missing).  A deleted node keeps its row slot but is gone from searches, exact scans, restricted candidate sets, the
 * prefilter's candidate mapping, the SimHash directory and hvx_index_contains; hvx_index_rows still counts the slot,
 * hvx_index_live_rows does not; hvx_index_export_graph returns empty rows for it.  Same generation rules as
 * hvx_index_insert_batch: owner handle only, visible_seq + 1 per call that deleted something, forks adopt with hvx_index_refresh.
 * Its id comes back through hvx_index_upsert_batch (into the same slot), not through hvx_index_insert_batch (ids ascend).
 * f32 and bf16 images (bf16: relinked on the rounded vectors), degree limits <= 64 on rows of at most 64 ids; fp8 images are refused.
 * HVX_ERR_UNSUPPORTED when more than 4 096 rows of one layer hold a node or their joint
 * neighbourhood exceeds 16 384 rows (after such a failure the image is partially relinked: discard the handle).
 */
typedef struct hvx_delete_stats {
    uint32_t requested, deleted, missing; /* ids given / nodes removed / ids that were not (or no longer) in the image */
    uint32_t entry_moves;                  /* deletes that took the entry point */
    uint32_t relinked_rows;                /* relink_neighbor calls */
    uint32_t reserved;
    double seconds;
} hvx_delete_stats;
int hvx_index_delete_batch(hvx_index *, const uint64_t *node_ids /*[count]*/, uint32_t count, hvx_delete_stats *stats /*nullable*/);
/*
 * VectorInsertContract::Upsert (mutation.rs:642-780; index.rs:2018-2060): for every id, in order, a LIVE id is deleted first (as
 * hvx_index_delete_batch does), then the vector is linked in under that id by the reference's sequential insertion: into the node's
 * own row slot when the image holds one -- a deleted node keeps its slot, so the id returns to its place in the ascending id order
 * and every id tie-break stays the reference's -- or appended when the id is above every id of the image (levels[i], NULL = 0, is
 * used there; a revived node keeps the level of its slot: the reference's level draw is random, any draw is a valid one).  Every
 * vector is validated before anything changes.  HVX_ERR_UNSUPPORTED for an id that lies between the image's ids without a slot.
 * Rows equal the oracle's delete + insert of the same ids (tests/test_gpu_delete.py).
 * bf16 images (round 6): the new vector is rounded to bf16, validated and linked over the bf16 rows (== the oracle on the rounded vectors);
 * appended ids need spare rows (hvx_index_import_reserve).  fp8 images are read-only.
 */
int hvx_index_upsert_batch(hvx_index *, const uint64_t *node_ids /*[count]*/, const float *vectors /*[count][dim] host or device*/,
                           const uint16_t *levels /*[count] or NULL*/, uint32_t count, const hvx_build_params *params /*nullable*/,
                           hvx_build_stats *stats /*nullable*/);
uint64_t hvx_index_live_rows(const hvx_index *);    /* rows visible to this handle that are not deleted */
int hvx_index_contains(const hvx_index *, uint64_t node_id); /* 1 when the id holds a (live) vector in this handle's generation */
int hvx_index_refresh(hvx_index *);                 /* adopt the image's visible generation (a fork; a no-op when nothing changed) */
uint64_t hvx_index_visible_seq(const hvx_index *);  /* generation this handle searches: 1 after import / build, +1 per insert batch */
uint64_t hvx_index_rows(const hvx_index *);         /* rows visible to this handle */
uint64_t hvx_index_row_capacity(const hvx_index *); /* rows the image can hold */
/*
 * add_bidirectional_link(from -> to) on layer 0 of an index image (mutation.rs:1498-1583): `from` is appended to the row of
 * `to` under the row's lock; a row beyond Mmax is pruned (rank by distance to the owner, select_diverse + backfill,
 * mod.rs:809-856), and every neighbour the prune drops loses its reverse edge (mutation.rs:1890-1908).  It runs the kernel the
 * BATCHED build links with (one 256-thread workgroup per link, the prune evaluated eagerly from LDS): the links are applied one
 * launch after the other in list order (concurrent = 0: a defined order, what the parity tests compare with the oracle's
 * select_diverse) or all in ONE launch (concurrent = 1: the batched build's situation -- rows change under their locks in an
 * undefined order).  Served shapes = the workgroup kernel's: f32 rows, L2 / cosine, dim % 32 == 0, m0 <= 32.  The index must
 * not be searched concurrently.
 */
int hvx_index_link_rows(hvx_index *, const uint64_t *from_ids, const uint64_t *to_ids, uint32_t n_links, uint32_t concurrent);
/*
 * Audit of a graph image on the device: the invariants of every persisted row (neighbor_set.rs:1-9: ascending, deduped,
 * self-free, degree-bounded; mutation.rs:1498-1583,1890-1908: links are bidirectional on their layer) and reachability of every
 * node from the entry point on layer 0.  A graph the reference could have written has every counter below `bfs_levels_l0` at 0
 * (unreachable_l0 may be > 0 for a reference-built graph too: HNSW does not guarantee connectivity).
 */
typedef struct hvx_graph_audit {
    uint64_t nodes, up_rows, edges_l0, edges_up;
    uint64_t asymmetric_edges_l0, asymmetric_edges_up; /* u lists v on a layer, v does not list u there */
    uint64_t unsorted_entries;      /* an id not greater than its predecessor in the row (covers duplicates) */
    uint64_t self_loops, out_of_range_ids, holes /* a valid id behind padding */, level_violations /* edge to a node below the layer */;
    uint64_t degree_overflow_rows;  /* rows longer than m0 (layer 0) / m (upper layers) */
    uint64_t unreachable_l0;        /* nodes a layer-0 BFS from the entry point does not reach */
    uint32_t max_degree_l0, max_degree_up, bfs_levels_l0, max_layer, has_entry, reserved;
} hvx_graph_audit;
int hvx_index_audit_graph(const hvx_index *, hvx_graph_audit *out);
/* Read the graph of an index back in hvx_index_import's CSR layout (external ids, rows ascending): sizes first, then the arrays
 * (l0_offsets [n+1], l0_neighbors [l0_edges], level [n] nullable, up_offsets [up_rows+1] / up_neighbors [up_edges] nullable). */
int hvx_index_graph_sizes(const hvx_index *, uint64_t *l0_edges, uint64_t *up_rows, uint64_t *up_edges, uint64_t *entry_point,
                          uint32_t *max_layer, uint32_t *has_entry);
int hvx_index_export_graph(const hvx_index *, uint64_t *l0_offsets, uint64_t *l0_neighbors, uint16_t *level, uint64_t *up_offsets,
                           uint64_t *up_neighbors);

/*
 * Process-level preparation, called ONCE by the host's single-threaded start-up code before its first HIP call.  The HIP runtime
 * multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4) and two execution lanes that land on
 * one queue run back to back; this sets the variable to `hw_queues` (0 = 8) unless the host has already exported a value.  (Up to
 * round 4 a load-time constructor did this; a library should not write its host's environment behind its back.)
 */
int hvx_runtime_prepare(uint32_t hw_queues);
/*
 * Roofline denominator ("use the rocprof-measured stream-read peak as the denominator and state both"): times a
 * read-only streaming kernel (global_load_dwordx4, every byte of a `bytes`-sized HBM buffer read once per launch, nothing written)
 * `iters` times on `device` and reports the best and the mean rate in GB/s (1e9 bytes).  bytes >= 1 GiB keeps the 256 MB
 * Infinity Cache out of the number.  Diagnostic: allocates and frees its own buffer.
 */
int hvx_device_stream_read_gbs(int32_t device, uint64_t bytes, uint32_t iters, float *out_best_gbs, float *out_mean_gbs);

const char *hvx_last_error(void); /* thread-local */
const char *hvx_version(void);

#ifdef __cplusplus
}
#endif
#endif
