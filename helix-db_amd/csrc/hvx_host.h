// hvx_host.h -- host-side state behind the opaque handles of include/helix_vec.h.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <vector>

#include "hvx_kernels.h"
#include "hvx_walk_core.h"

// Device allocations of one handle.  A fork (hvx_index_fork) keeps its parent's holder alive: the immutable index image
// (rows, graph, ids, headers, SimHash rows) is freed when the last handle that references it is gone.
struct hvx_allocs {
    int device = 0;
    std::vector<void *> v;
    ~hvx_allocs() {
        if (v.empty()) return;
        (void)hipSetDevice(device);
        for (void *p : v) (void)hipFree(p);
    }
};

// State that every handle of one device image shares (the imported handle and its forks): built once, by whichever lane
// needs it first, freed with the image.
struct hvx_image_shared {
    std::mutex mu;
    int device = 0;
    uint16_t *shadow = nullptr;   // f32 rows: bf16 (RNE) shadow of the rows for the large-tile exact-scan kernels (hvx_flat_tile.hip)
    uint32_t shadow_rows = 0;     // rows of it that are converted (a growable image extends it on the next scan)
    bool shadow_failed = false;   // no memory for it: the scan stays on the 128 x 128 kernel
    // the SimHash directory of the restricted walk (hvx_restricted_walk.hip): the rows [0xF1][index][0x17][order_code][node]
    // in key order = (order code, row) ascending, built from the attached SimHash rows on first use
    uint64_t *dir_code = nullptr;
    uint32_t *dir_row = nullptr;
    uint32_t *dir_prefix = nullptr;    // [65537] first directory entry per 16-bit order-code prefix
    const uint64_t *dir_for = nullptr; // the node_hash array the directory was derived from
    uint32_t dir_rows = 0;             // rows the directory covers (an image that has grown since needs a new one)
    // Generation view of a GROWABLE image (hvx_build_params.reserve_rows; hvx_index_insert_batch): the rows, entry point and top
    // layer that are visible.  Written by the owner handle when an insert batch is complete (visible_seq + 1), adopted by a fork
    // when the host calls hvx_index_refresh on it -- the registry's attach rule (read_index.rs:55-61: a cache guard serves a
    // snapshot only when hydration_seq == snapshot_seq) decides when.
    uint64_t visible_seq = 0;
    uint32_t v_n = 0, v_entry = 0, v_max_layer = 0, v_has_entry = 0;
    // ... and the rows that are deleted (hvx_index_delete_batch): host flags of this generation, their number, the device bitmap
    std::shared_ptr<const std::vector<uint8_t>> v_dead;
    uint32_t v_n_dead = 0;
    const uint32_t *v_dead_dev = nullptr;
    // the deleted-row flags the SimHash directory was built without.  Held, not just compared (round 6): a flags vector freed by
    // an in-slot upsert can be handed the same address again by the next delete -- a raw pointer + the live-row count then matched a
    // DIFFERENT generation (delete A, scan, upsert A, delete B) and the stale directory was reused
    std::shared_ptr<const std::vector<uint8_t>> dir_dead;
    uint64_t vec_epoch = 0;            // bumped when a row's VECTOR changes in place (hvx_index_upsert_batch into the node's own slot): per-handle
                                       // row norms are rebuilt when their epoch is behind; the bf16 shadow is reconverted (shadow_rows = 0)
    // In-place REWRITES of the shared rows (hvx_index_delete_batch relinks rows and empties the node's own; an in-slot upsert replaces a
    // vector): unlike appended rows they cannot be hidden from a handle that still shows an older generation -- its entry point may be
    // the deleted node, whose rows are now empty (round 6).  The owner bumps this word with the publish; every handle
    // compares it at the top of each search / scan entry point and adopts the visible generation first (hvx_index::sync_rewrites).
    std::atomic<uint64_t> rewrite_epoch{0};
    uint64_t v_entry_point = 0;
    bool v_contiguous = true;
    std::shared_ptr<const std::vector<uint64_t>> v_ids;
    ~hvx_image_shared() {
        if (shadow || dir_code || dir_row) (void)hipSetDevice(device);
        if (shadow) (void)hipFree(shadow);
        if (dir_code) (void)hipFree(dir_code);
        if (dir_row) (void)hipFree(dir_row);
        if (dir_prefix) (void)hipFree(dir_prefix);
    }
};

struct hvx_index {
    int device = 0;
    hvx_index_desc desc{};
    hvx::DevIndex dev{};
    float limit = 0.f;               // VectorComponentLimit (domain.rs:26-78); +inf for cosine
    uint32_t max_batch = 1024;
    uint32_t words_per_query = 0;
    uint32_t occupancy = 1;          // wave kernel build: queries per SIMD (hvx_index_set_occupancy)
    uint32_t opt[HVX_OPT_COUNT] = {}; // execution-path selectors (hvx_index_set_option); 0 = the library's own choice
    uint32_t last_scan_path = 0;     // hvx_scan_path flags of the handle's last exact scan (hvx_index_last_scan_path)
    bool bitmap_dirty = false;       // d_bitmap holds stale visited bits (general kernel ran last)
    hipStream_t stream = nullptr;      // stream in use
    hipStream_t own_stream = nullptr;  // created at import
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> ring;    // asynchronous timing: event pairs of the searches since hvx_index_timing_begin
    uint32_t ring_cap = 0, ring_n = 0;
    unsigned long long *d_wclk = nullptr; // [ring_cap][max_batch][2] wave start/end clocks of the searches since timing_begin
    uint32_t wclk_cap = 0;
    std::mutex mu;                   // calls on one index are serialised on its stream
    std::shared_ptr<hvx_allocs> allocs = std::make_shared<hvx_allocs>();
    std::vector<std::shared_ptr<hvx_allocs>> image; // fork: the allocations of the handles it descends from (the image it aliases)
    bool is_fork = false;
    // growable image: rows / upper rows the arrays were allocated for, upper rows in use, the generation this handle's view shows
    uint64_t cap_rows = 0, cap_up_rows = 0, up_rows_used = 0, seen_seq = 0;
    void publish_view(bool bump = true); // owner: this handle's view becomes the image's visible generation (visible_seq + 1 when bump)
    bool adopt_view();     // any handle: take the image's visible generation; true if the view changed
    uint64_t seen_rewrite = 0;
    void sync_rewrites() { // rows of the image were rewritten in place since this handle last looked: its generation is gone
        const uint64_t e = shared->rewrite_epoch.load(std::memory_order_acquire);
        if (e != seen_rewrite) { (void)adopt_view(); seen_rewrite = e; }
    }
    std::shared_ptr<const std::vector<uint64_t>> ids_p = std::make_shared<std::vector<uint64_t>>(); // host copy of node ids
    const std::vector<uint64_t> &ids_ref() const { return *ids_p; }
    bool contiguous = false;
    // deleted rows of this handle's generation (hvx_delete.hip): a deleted node keeps its row slot -- emptied, unreachable, absent
    // from every id -> row lookup and from every exact scan (which then runs over `f_live`, the ascending list of live rows)
    std::shared_ptr<const std::vector<uint8_t>> dead_p; // [rows] 1 = deleted; null while nothing has been deleted
    uint32_t n_dead = 0;
    uint32_t *d_dead = nullptr;      // owner: the image's device bitmap (one bit per row of capacity), allocated by the first delete
    uint32_t *f_live = nullptr;      // live rows ascending, built on demand for the generation in `live_for`
    std::shared_ptr<const std::vector<uint8_t>> live_for; // (held: its address cannot be recycled while this list is cached)
    bool live_valid = false;
    uint32_t live_rows_n = 0, cap_live = 0;
    int ensure_live();               // f_live / live_rows_n for this handle's generation
    uint32_t live_rows() const { return dev.n - n_dead; }
    void *del_scratch = nullptr;     // owner: scratch of the delete kernels (hvx_delete.hip)
    hipEvent_t del_ev = nullptr;     // ... and the event behind a delete's prep kernel (the relink-source counts are read back there)
    uint32_t del_layers = 0;
    // owner: scratch of the insert path (hvx_build.hip insert_range, upsert staging), kept across calls: a one-node insert paid nine
    // hipMalloc + hipFree, a stream and four events per call (round 6: ~2 ms of a 6.4-ms upsert)
    void *ins_scratch = nullptr;     // candidate / selection buffers, status words, the sequential link's matrices
    size_t ins_cap = 0;
    uint32_t *ins_locks = nullptr;   // one lock per row (all zero between calls)
    uint64_t ins_locks_rows = 0;
    hipStream_t ins_stream = nullptr;
    hipEvent_t ins_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    float *ins_rows = nullptr;       // upsert: the validated vectors before they reach their slots
    size_t ins_rows_cap = 0;
    // per-batch device scratch
    uint32_t *d_bitmap = nullptr, *d_qstatus = nullptr, *d_tie = nullptr;
    float *d_qhdr = nullptr;
    hvx_query_stats *d_qstats = nullptr;
    unsigned long long *d_prof = nullptr; // HVX_WAVE_PROF tuning buffer
    // staging of the host-pointer API
    float *s_queries = nullptr;
    uint64_t *s_ids = nullptr;
    float *s_scores = nullptr;
    uint32_t *s_counts = nullptr, *s_status = nullptr;
    size_t cap_q = 0, cap_o = 0;
    uint32_t cap_b = 0;
    // pinned host mirror of the staging buffers.  A hipMemcpyAsync from / to pageable memory is a synchronous staged copy (35-50 us
    // of host time EACH, five per call: measured with rocprofv3 --hip-trace, profiles/history/r03r); through pinned memory the five copies
    // are enqueued in ~5 us each and the host pays two memcpys of a few KB.
    unsigned char *h_pin = nullptr;
    size_t cap_pin = 0;
    int pin(size_t bytes);                                   // >= bytes of pinned host memory in h_pin
    uint32_t *h_flags = nullptr;                             // pinned: small per-batch read-backs of the scan pipelines (certificates, overflow word)
    size_t cap_flags = 0;
    int pin_flags(size_t words);
    int stage_in(const float *queries, uint32_t cb);         // user queries -> h_pin -> s_queries (enqueued)
    int stage_out(uint32_t cb, uint32_t k);                  // s_ids / s_scores / s_counts / s_status -> h_pin (enqueued)
    // after the stream has been synchronised: h_pin -> the caller's arrays
    void deliver(uint32_t cb, uint32_t k, uint64_t *out_ids, float *out_scores, uint32_t *out_counts, uint32_t *out_status) const;
    // exact-scan scratch
    float *f_dist = nullptr, *f_top_s = nullptr;
    uint32_t *f_top_i = nullptr, *f_top_c = nullptr, *f_subset = nullptr;
    size_t cap_dist = 0, cap_top = 0, cap_subset = 0;
    // per-query fallback of the f32 matrix-core exact scan (queries whose certificate was not reached)
    std::vector<uint32_t> m_failed;
    uint32_t *fb_idx = nullptr, *fb_cnt = nullptr, *fb_st = nullptr;
    float *fb_q = nullptr, *fb_sc = nullptr;
    uint64_t *fb_ids = nullptr;
    uint32_t cap_fb = 0;
    uint32_t *pf_blocks = nullptr;   // fused prefilter: per-block candidate counts / scan
    uint32_t cap_pf_blocks = 0;
    uint32_t cap_topc = 0;
    // bf16 exact scan on the matrix cores (hvx_flat_mfma.hip)
    uint16_t *m_qhi = nullptr, *m_qlo = nullptr;
    uint16_t *m_qhi8 = nullptr;      // fp8 rows: the hi parts again, in the operand order of the 256 x 256 kernel (hvx_flat_tile.hip)
    unsigned char *m_qmx = nullptr;  // fp8 rows, MX build: [bpad][dim / 64][64 e4m3 hi codes | 64 e4m3 lo codes] (stored order of the rows)
    uint32_t *m_qexp = nullptr;      // [bpad] E8M0 scale of the hi codes (the lo codes': 4 less)
    size_t cap_qmx = 0;
    uint16_t *m_shadow = nullptr;    // f32 rows: this handle's view of shared->shadow (set once it is complete)
    std::shared_ptr<hvx_image_shared> shared = std::make_shared<hvx_image_shared>(); // forks share their parent's
    float *m_qn2 = nullptr, *m_rowterm = nullptr; // |q|^2 per query; |x|^2 per row
    uint32_t *m_cert = nullptr;
    // filtered-epilogue pipeline: running thresholds and the (score, row) pairs a filtered launch lets through
    float *m_thr = nullptr, *m_csc = nullptr;
    uint32_t *m_cid = nullptr, *m_ccnt = nullptr;
    size_t cap_cand = 0, cap_cand_b = 0;
    size_t cap_qsplit = 0;
    float m_xmax2 = 0.f;
    uint32_t rowterm_rows = 0;       // rows m_rowterm / m_xmax2 cover
    uint64_t rowterm_epoch = 0;      // shared->vec_epoch they were computed at
    uint32_t m_fast_misses = 0, m_fast_skipped = 0; // consecutive scans whose one-pass attempt missed a certificate / scans that skipped it
    // non-strict search arms (hvx_params.hip): per-node SimHash rows, the hasher, per-batch fingerprints
    bool has_simhash = false;
    hvx_simhash_config sh_cfg{};
    uint64_t *d_node_hash = nullptr, *d_qhash = nullptr;
    uint32_t *d_bitmap2 = nullptr;   // SimHash-cached ids of the bitmap (spill) mode, uncached-handle accounting only
    float *d_planes_t = nullptr, *d_thr_break = nullptr;
    hvx_adaptive_stats *d_astats = nullptr;
    uint32_t thr_configured = 0xFFFFFFFFu; // what d_thr_break was built for
    float thr_failure = -1.f;
    // restricted walk (hvx_restricted_walk.hip): membership / seen bitmaps, plans, sample rows, per-query stats
    uint32_t *w_allowed = nullptr, *w_seen = nullptr, *w_samples = nullptr, *w_rows = nullptr;
    void *w_plans = nullptr, *w_counters = nullptr;
    size_t cap_w_allowed = 0, cap_w_seen = 0, cap_w_samples = 0, cap_w_rows = 0, cap_w_q = 0;
    // one-launch restricted exact scan (hvx_restricted_exact.hip): the slices' result lists, the self-cleaning "invalid score" / "workgroups
    // done" words, and the per-query candidate id lists of a batch (external ids + CSR offsets)
    float *x_part_sc = nullptr;
    uint32_t *x_part_row = nullptr, *x_bad = nullptr, *x_done = nullptr;
    uint64_t *x_ids = nullptr, *x_off = nullptr;
    size_t cap_x_part = 0, cap_x_ids = 0, cap_x_off = 0;
    uint32_t cap_x_ctl = 0;
    // exact tail of the small-batch matrix-core scan (hvx_flat_tail.hip): the slices' exact lists; [bad | done | threshold] words per query
    float *t_part_sc = nullptr;
    uint32_t *t_part_row = nullptr, *t_ctl = nullptr;
    size_t cap_t_part = 0;
    uint32_t cap_t_ctl = 0;

    int dalloc(void **p, size_t bytes);
    int regrow(void **p, size_t bytes);   // dalloc after releasing *p (scratch buffers that grow)
    int stage(uint32_t b, uint32_t k);
    int flat_scratch(uint32_t b, uint32_t k, uint32_t chunk_rows);
    // external id -> internal row, kSentinel when absent
    uint32_t find(uint64_t id) const {
        const uint32_t row = find_slot(id);
        if (row != hvx::kSentinel && dead_p && (*dead_p)[row]) return hvx::kSentinel; // deleted: the id holds no vector any more
        return row;
    }
    // the row slot of an id, deleted or not
    uint32_t find_slot(uint64_t id) const {
        const std::vector<uint64_t> &ids = *ids_p;
        if (ids.empty()) return hvx::kSentinel;
        if (contiguous) {
            if (id < ids[0] || id - ids[0] >= ids.size()) return hvx::kSentinel;
            return (uint32_t)(id - ids[0]);
        }
        auto it = std::lower_bound(ids.begin(), ids.end(), id);
        if (it == ids.end() || *it != id) return hvx::kSentinel;
        return (uint32_t)(it - ids.begin());
    }
};

namespace hvx {
// Environment switches exist only in tuning builds (make TUNING=1 -> -DHVX_TUNING): the release library never calls getenv,
// and the measurement-only code they select (ablation, phase profiling, experimental tile builds) is not compiled in.
#ifdef HVX_TUNING
inline const char *tuning_env(const char *name) { return getenv(name); }
#else
inline const char *tuning_env(const char *) { return nullptr; }
#endif
int fail(int code, const char *fmt, ...);
int import_index(const hvx_index_desc *desc, const uint64_t *node_ids, const float *vectors, const uint64_t *l0_offsets,
                 const uint64_t *l0_neighbors, const uint16_t *level, const uint64_t *up_offsets, const uint64_t *up_neighbors,
                 uint32_t min_s0, uint32_t min_su, hvx_index **out, uint64_t reserve_rows = 0, uint64_t reserve_up_rows = 0);
int check_k_ef(uint32_t k, uint32_t ef);
// enqueue validation + the search kernel for one chunk of <= max_batch device-resident queries;
// ad != NULL selects the non-strict arms
int enqueue_search(const hvx_index *ix, const float *d_queries, uint32_t b, uint32_t k, uint32_t ef, uint64_t *d_ids,
                   float *d_scores, uint32_t *d_counts, uint32_t *d_status, hvx_query_stats *d_qstats, bool timed,
                   const AdaptArgs *ad = nullptr);
int enqueue_search_params(hvx_index *ix, const float *d_queries, uint32_t b, const hvx_search_params *params, uint64_t *d_ids, float *d_scores,
                          uint32_t *d_counts, uint32_t *d_status);
int collect_stats(hvx_index *ix, uint32_t b, const hvx_query_stats *d_qstats, hvx_stats *stats);
float component_limit(uint32_t metric, uint32_t dim);
int flat_scan_device(hvx_index *ix, const float *d_queries, uint32_t b, uint32_t k, const uint32_t *d_subset,
                     uint32_t n_rows, uint64_t *d_ids, float *d_scores, uint32_t *d_counts, uint32_t *d_status,
                     bool timed);
bool flat_scan_on_matrix_cores(const hvx_index *ix, uint32_t b, uint32_t k, uint32_t n_rows);
int flat_scan_valu(hvx_index *ix, const float *d_queries, uint32_t b, uint32_t k, const uint32_t *d_subset, uint32_t n_rows,
                   uint64_t *d_ids, float *d_scores, uint32_t *d_counts, uint32_t *d_status, bool timed, bool record_begin);
int flat_mfma_device(hvx_index *ix, const float *d_queries, uint32_t b, uint32_t k, const uint32_t *d_subset, uint32_t n_rows,
                     uint64_t *d_ids, float *d_scores, uint32_t *d_counts, uint32_t *d_status, bool timed);
// restricted search under the reference's execution plan (hvx_restricted_walk.hip)
struct RestrictedPlan {
    uint32_t strategy; // HVX_RESTRICTED_EXACT / HVX_RESTRICTED_FILTERED
    walk::Plan p;
};
// restricted_execution_plan_with_beam_percent (restricted.rs:426-453) for a candidate population (ids incl. unindexed ones)
int restricted_make_plan(const hvx_restricted_params &rp, uint64_t candidates, uint32_t dim, RestrictedPlan *out, const hvx_index *ix = nullptr);
// deterministic_sample_ids (restricted.rs:321-342) as ranks into the ascending candidate list
void restricted_sample_ranks(uint64_t candidates, uint32_t count, std::vector<uint64_t> &out);
// run a plan for b host-resident queries over ONE candidate set given as device rows (ascending); d_samples = the plan's
// p.n_sample deterministic seeds as rows (kSentinel = not indexed); out arrays have row length k_stride
int restricted_run_plan(hvx_index *ix, const float *queries, uint32_t b, uint32_t k_stride, const RestrictedPlan &plan, const uint32_t *d_rows,
                        uint32_t n_rows, const uint32_t *d_samples, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                        uint32_t *out_status, hvx_restricted_stats *rstats, hvx_stats *stats);
hipError_t launch_bf16_row_norm2(const uint16_t *rows, uint32_t n, uint32_t dim, float *out, hipStream_t s);
int flat_scan_host(hvx_index *ix, const float *queries, uint32_t b, uint32_t k, const uint32_t *d_subset,
                   uint32_t n_rows, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                   uint32_t *out_status, hvx_stats *stats);
// the one-launch restricted exact scan (hvx_restricted_exact.hip): k <= 64 over f32 / bf16 rows of a non-empty image
bool restricted_direct_supported(const hvx_index *ix, uint32_t k);
int restricted_direct_enqueue(hvx_index *ix, const float *d_queries, uint32_t b, uint32_t k, uint32_t k_stride, const uint32_t *d_rows,
                              uint32_t n_rows, const uint64_t *d_ext_ids, const uint64_t *d_offsets, uint32_t max_set, uint64_t *d_ids,
                              float *d_scores, uint32_t *d_counts, uint32_t *d_status, const uint32_t *d_lens = nullptr, uint32_t ext_stride = 0,
                              const uint32_t *d_n_rows = nullptr);
// the one-launch scan over a row list whose LENGTH lives on the device (counters[0] = rows, counters[1] = candidate population; rows_cap = the
// host's bound): host-resident queries, results + the two counters delivered with ONE stream wait (hvx_restricted_walk.hip)
int restricted_direct_shared_devcount(hvx_index *ix, const float *queries, uint32_t b, uint32_t k, const uint32_t *d_rows, uint32_t rows_cap,
                                      const uint32_t *d_counters, uint64_t *out_ids, float *out_scores, uint32_t *out_counts, uint32_t *out_status,
                                      uint64_t *out_candidates, hvx_restricted_stats *rstats, hvx_stats *stats);
hipError_t launch_stage_ids(const uint64_t *src, const uint32_t *lens, uint32_t stride, uint64_t *dst, uint32_t b, hipStream_t s);
// exact tail of the small-batch matrix-core scan (hvx_flat_tail.hip): k <= 64, AVX+FMA tree, L2 / cosine, f32 or bf16 rows
bool flat_tail_supported(const hvx_index *ix, uint32_t b, uint32_t k, uint32_t n_rows);
int flat_tail_enqueue(hvx_index *ix, const float *d_queries, uint32_t b, uint32_t k, const float *dots, uint32_t chunk_ld, uint32_t rows,
                      const uint32_t *d_subset, const float *rowterm, const float *qn2, float erel, uint64_t *d_ids, float *d_scores,
                      uint32_t *d_counts, uint32_t *d_status);
hipError_t launch_stage_queries(const float *src, float *dst, uint32_t dim, uint32_t b, hipStream_t s);
// hvx_search_restricted_batch_params behind the handle's lock (hvx_restricted_walk.hip)
int restricted_search_host(hvx_index *ix, const float *queries, uint32_t b, const hvx_restricted_params &rp, const uint64_t *allowed_ids,
                           const uint64_t *allowed_offsets, uint64_t n_allowed, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                           uint32_t *out_status, hvx_restricted_stats *out_rstats, hvx_stats *stats);
} // namespace hvx
