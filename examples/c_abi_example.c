/*
 * c_abi_example.c -- the C ABI of libhelix_vec_gfx950.so from plain C (what a cgo / Rust `extern "C"` / JNI shim sees).
 *
 *   gcc -std=c99 -I include examples/c_abi_example.c -L helix-db_amd -lhelix_vec_gfx950 -Wl,-rpath,$PWD/helix-db_amd -o c_abi_example
 *
 * Builds a toy index (8 nodes on a line, dim 128 -- the smallest dimension the one-wavefront-per-query kernel serves),
 * then runs the three calls a HelixDB host makes: the strict search of the reference's golden tests, the production
 * default `SearchParams::new(k)` through hvx_search_batch_params, a restricted search over an id list, and a delete.
 * Needs an MI355X to RUN: tests/test_abi_and_host.py compiles and links it (CPU), tests/test_gpu_parity.py runs it and checks its
 * output against the oracle (-m gpu).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "helix_vec.h"

#define N 8
#define DIM 128
#define K 3

static int check(int rc, const char *what) {
    if (rc != HVX_OK) fprintf(stderr, "%s failed (%d): %s\n", what, rc, hvx_last_error());
    return rc;
}

int main(void) {
    /* node i sits at (i, 0, 0, ...); layer-0 rows link the line; ids are external u64 node ids */
    static float vectors[N][DIM];
    uint64_t ids[N], l0_offsets[N + 1], l0_neighbors[2 * N];
    uint64_t e = 0;
    for (int i = 0; i < N; ++i) {
        ids[i] = 100 + (uint64_t)i;
        vectors[i][0] = (float)i;
        l0_offsets[i] = e;
        if (i > 0) l0_neighbors[e++] = ids[i] - 1;      /* rows hold ascending external ids */
        if (i + 1 < N) l0_neighbors[e++] = 100 + (uint64_t)i + 1;
    }
    l0_offsets[N] = e;

    hvx_index_desc desc;
    memset(&desc, 0, sizeof(desc));
    desc.dim = DIM; desc.metric = HVX_L2_SQUARED; desc.dtype = HVX_F32; desc.float_kernel = HVX_KERNEL_AVX_FMA;
    desc.n = N; desc.m = 16; desc.m0 = 32; desc.has_entry = 1; desc.max_layer = 0; desc.entry_point = ids[0];
    desc.device = -1; desc.max_batch = 16;
    hvx_index *ix = NULL;
    if (check(hvx_index_import(&desc, ids, &vectors[0][0], l0_offsets, l0_neighbors, NULL, NULL, NULL, &ix), "hvx_index_import")) return 1;

    float query[DIM] = {0};
    query[0] = 5.2f;
    uint64_t out_ids[K];
    float out_scores[K];
    uint32_t count = 0;

    /* 1. strict-exhaustive search (the reference's golden-test parameters) */
    hvx_stats stats;
    memset(&stats, 0, sizeof(stats));
    if (check(hvx_search_batch(ix, query, 1, K, 16, out_ids, out_scores, &count, NULL, &stats), "hvx_search_batch")) return 1;
    printf("strict : %u results, nearest id %llu score %g, %llu distance computations\n", count,
           (unsigned long long)out_ids[0], out_scores[0], (unsigned long long)stats.distance_computations);

    /* 2. what the query path runs: SearchParams::new(k) = SimHashMode::Adaptive (needs the per-node SimHash rows) */
    hvx_simhash_config cfg;
    hvx_simhash_config_default(&cfg);
    if (check(hvx_index_set_simhash(ix, &cfg, NULL), "hvx_index_set_simhash")) return 1;
    hvx_search_params params;
    hvx_search_params_default(&params, K);
    hvx_adaptive_stats astats;
    if (check(hvx_search_batch_params(ix, query, 1, &params, out_ids, out_scores, &count, NULL, NULL, NULL, &astats), "hvx_search_batch_params")) return 1;
    printf("default: %u results, nearest id %llu, %u RNG words drawn\n", count, (unsigned long long)out_ids[0], astats.rng_words);

    /* 3. restricted search: only these ids may be returned (RestrictedVectorCandidates) */
    const uint64_t allowed[3] = {100, 103, 107};
    if (check(hvx_search_restricted_batch(ix, query, 1, K, 100, allowed, NULL, 3, out_ids, out_scores, &count, NULL, NULL), "hvx_search_restricted_batch")) return 1;
    printf("restricted: %u results, nearest allowed id %llu\n", count, (unsigned long long)out_ids[0]);

    /* 4. a write: VectorIndex::delete of the nearest node (unlink, relink of its neighbours, entry repair), then the same search */
    const uint64_t gone[2] = {105, 9999};          /* an unknown id succeeds and counts as `missing` (index.rs:2263) */
    hvx_delete_stats dstats;
    if (check(hvx_index_delete_batch(ix, gone, 2, &dstats), "hvx_index_delete_batch")) return 1;
    if (check(hvx_search_batch(ix, query, 1, K, 16, out_ids, out_scores, &count, NULL, NULL), "hvx_search_batch")) return 1;
    printf("after delete: %u deleted, %u missing, %llu live rows, nearest id %llu, contains(105) = %d\n", dstats.deleted, dstats.missing,
           (unsigned long long)hvx_index_live_rows(ix), (unsigned long long)out_ids[0], hvx_index_contains(ix, 105));

    hvx_index_free(ix);
    return 0;
}
