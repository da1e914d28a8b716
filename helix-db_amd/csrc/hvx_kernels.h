// hvx_kernels.h -- host-visible launch interface of the gfx950 kernels (internal to the library).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/helix_vec.h"
#include "hvx_device.h"

namespace hvx {

// Non-strict layer-0 arms: the per-launch projection of SearchParams + VectorIndexConfig
// (Layer0Policy::from_deployed / AdaptiveBypassPolicy::from_deployed, policy.rs:54-110,203-226) and the per-query
// fingerprints.  Only read by the AD instantiations of the wave kernel.
struct AdaptArgs {
    const uint64_t *node_hash;  // [n] SimHash row of every node, internal order
    const uint64_t *qhash;      // [b] query SimHash
    const float *thr_break;     // [64] adaptive_threshold as a table: brk[t-1] = largest delta with threshold(delta) >= t,
                                //      -1 when no delta reaches t (policy.rs:577-599 evaluated with the HOST libm)
    hvx_adaptive_stats *stats;  // [b] nullable
    uint32_t filtering;         // SimHashFilteringPolicy: 0 Disabled, 1 Fixed, 2 Adaptive
    uint32_t configured;        // configured collision threshold
    uint32_t sampling;          // FrontierSamplingPolicy: 0 Exhaustive, 1 Fixed, 2 Adaptive
    float ratio;                // base sampling ratio (override or index sampling_ratio)
    float pre_override;         // pre-sampling override, < 0 = None
    uint32_t count_reads;       // 1: uncached handle -- SimHash rows are stable-view reads (memory_store.rs:338-347), budget trigger live
    uint32_t read_budget;       // max(ef * multiplier, min_frontier) (policy.rs:214-218)
    uint32_t bypass_windowed;   // AdaptiveBypassPolicy::Windowed
    uint32_t min_frontier, window_expansions;
    float min_filter_rate;
};

struct HnswArgs {
    DevIndex ix;
    const float *queries;      // [b][dim]
    const uint32_t *qstatus;   // [b] validation status (0 = ok), nullable
    const float *qhdr;         // [b] query cosine norm header, nullable (non-cosine)
    uint32_t *bitmap;          // [b][words_per_query] visited bits, zeroed before launch
    uint32_t *bitmap2;         // same shape, SimHash-cached ids in bitmap mode (non-strict arms with read accounting), nullable
    uint32_t words_per_query;
    uint32_t k, ef;
    uint64_t *out_ids;         // [b][k]
    float *out_scores;         // [b][k]
    uint32_t *out_counts;      // [b]
    uint32_t *out_status;      // [b] nullable
    hvx_query_stats *qstats;   // [b] nullable
    uint32_t *tie_flags;       // [b] nullable
    uint32_t *rerun_list;      // wave kernel: [b] queries whose beam overflowed on equal scores, appended by the search launch ...
    uint32_t *rerun_ctl;       // ... [0] their number, [1] workgroups of the re-run launch that are done; NULL = no re-run follows
    unsigned long long *prof;  // [b][8] phase cycle counters of the PROF kernel variant, nullable
    unsigned long long *wave_clock; // [b][2] constant-rate (100 MHz) clock at the start / end of every query's wavefront, nullable
    uint32_t adaptive;         // 0 = strict-exhaustive search; 1 = non-strict arms, policy in `ad`
    const uint32_t *build_nodes; // BUILD instantiations (hvx_build.hip): [b] internal ids of the nodes being inserted (their rows are the queries)
    uint32_t build_ef_upper;   // ... beam width on the layers above 0 (ef is layer 0's)
    uint32_t only_flagged;     // wave kernel: 1 = the re-run launch -- only queries whose tie flag is set do anything (with a wider beam)
    uint32_t occupancy;        // wave kernel: 2 = the two-queries-per-SIMD build (callers with >= 2 batches in flight), else 1
    uint32_t pair;             // 1 = the owner / gatherer kernel (hvx_hnsw_pair.h) where it serves the launch
    uint32_t pair_gatherers;   // ... 1 = one gatherer wavefront per query even where three are built (A/B); 0 = the library's choice
    uint32_t log2cap;          // wave kernel: log2 slots of the LDS visited table (HVX_OPT_WAVE_LOG2CAP; small => bitmap spill); 0 = auto
    AdaptArgs ad;
};

size_t hnsw_lds_bytes(uint32_t ld);
// general kernel (4 wavefronts per query, any metric/dim/summation tree); dirties the HBM bitmap
hipError_t launch_hnsw_search(const HnswArgs &a, uint32_t b, hipStream_t s);
// fast kernel (1 wavefront per query, hvx_hnsw_wave.h); leaves the HBM bitmap zeroed
bool hnsw_wave_supported(const HnswArgs &a);
// non-strict arms: the AD instantiations of the wave kernel (f32 rows only)
bool hnsw_wave_adaptive_supported(const HnswArgs &a);
// SimHash of n device-resident f32 rows (row stride ld) against transposed hyperplanes [dim][64]
hipError_t launch_simhash_rows(const float *planes_t, const float *rows, uint32_t dim, uint32_t ld, uint64_t n, uint64_t *out,
                               hipStream_t s);
hipError_t launch_simhash_rows_bf16(const float *planes_t, const uint16_t *rows, uint32_t dim, uint64_t n, uint64_t *out,
                                    hipStream_t s);
// host: the 64 unit hyperplanes of SimHasher(dim, seed), transposed to [dim][64]
void simhash_planes_transposed(uint32_t dim, uint64_t seed, float *planes_t);
hipError_t launch_hnsw_wave(const HnswArgs &a, uint32_t b, hipStream_t s);

// validate queries on the device (domain.rs:113-157) and compute the cosine header
hipError_t launch_validate_queries(const DevIndex &ix, const float *d_queries, uint32_t b, float limit,
                                   uint32_t *d_status, float *d_qhdr, hipStream_t s);

hipError_t launch_validate_rows(const DevIndex &ix, uint32_t n, float limit, uint32_t *d_status, float *d_hdr,
                                hipStream_t s);

// bf16 row storage (hvx_dtype.hip): round an f32 staging copy in place, then pack it interleaved
hipError_t launch_round_bf16_inplace(float *v, size_t count, hipStream_t s);
hipError_t launch_pack_bf16(const float *staging, uint16_t *dst, uint32_t n, uint32_t dim, hipStream_t s);
// fp8 row storage: quantise rows (staging is overwritten with the dequantised values), |x|^2 of f32 rows
hipError_t launch_quantize_fp8(float *staging, uint8_t *dst, float *rowscale, uint32_t n, uint32_t dim, hipStream_t s);
hipError_t launch_read_rows(const DevIndex &ix, uint64_t row0, uint64_t n, float *out, hipStream_t s);
hipError_t launch_f32_row_norm2(const float *rows, uint32_t n, uint32_t ld, uint32_t dim, float *out, hipStream_t s);

// exact distance matrix tile + exact top-k selection (flat scan / restricted exact scan)
struct FlatArgs {
    DevIndex ix;
    const float *queries;     // [b][dim]
    const uint32_t *qstatus;  // [b]
    const float *qhdr;        // [b]
    const uint32_t *subset;   // optional [n_rows] internal ids to scan (restricted); NULL = all rows
    uint32_t n_rows;          // rows to scan (n or subset size)
    uint32_t row0, rows;      // this chunk: rows [row0, row0+rows) of the scan order
    float *dist;              // [b][chunk_ld] workspace
    uint32_t chunk_ld;
    uint32_t b, k;
    // running top-k per query, kept sorted; merged with every chunk
    float *top_scores;        // [b][k]
    uint32_t *top_ids;        // [b][k] internal ids
    uint32_t *top_counts;     // [b]
};
hipError_t launch_flat_distances(const FlatArgs &a, hipStream_t s);
hipError_t launch_flat_select(const FlatArgs &a, hipStream_t s);
// selection over ONE long score row per query, S slices per query; appends every slice's top-k to the query's pair list
hipError_t launch_flat_select_slices(const FlatArgs &a, uint32_t slices, float *cand_sc, uint32_t *cand_id, uint32_t *cand_cnt, uint32_t cand_cap,
                                     hipStream_t s);
hipError_t launch_flat_finish(const FlatArgs &a, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                              uint32_t *out_status, hipStream_t s);

hipError_t launch_merge_topk(uint32_t g, uint32_t b, uint32_t k, const uint64_t *ids, const float *scores,
                             const uint32_t *counts, uint64_t *out_ids, float *out_scores,
                             uint32_t *out_counts, hipStream_t s);

hipError_t launch_merge_topk_strided(uint32_t g, uint32_t b, uint32_t k, const uint64_t *ids, const float *scores,
                                     const uint32_t *counts, size_t ids_stride, size_t scores_stride, size_t counts_stride,
                                     uint64_t *out_ids, float *out_scores, uint32_t *out_counts, hipStream_t s);

} // namespace hvx
