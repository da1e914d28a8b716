// hvx_api.hip -- host side of the C ABI (include/helix_vec.h): index import, batched search entry
// points, scratch management.  No torch types, no CPU compute fallback: every search runs on the
// gfx950 kernels or fails with HVX_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hvx_host.h"
#include "hvx_flat_mfma.h"

using namespace hvx;

namespace {
thread_local std::string g_err;
}

namespace hvx {
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
} // namespace hvx

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(HVX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

extern "C" const char *hvx_last_error(void) { return g_err.c_str(); }
extern "C" const char *hvx_version(void) { return "helix_vec_gfx950 0.6 (round 6)"; }

// domain.rs:26-78 VectorComponentLimit::try_new
float hvx::component_limit(uint32_t metric, uint32_t dim) {
    if (metric == kCosine) return INFINITY;
    const double factor = metric == kL2 ? 8.0 : 4.0;
    const double divisor = (double)((uint64_t)dim * (uint64_t)factor);
    const double fmax = 3.4028234663852886e+38;
    const double exact = metric == kL2 ? std::sqrt(fmax / divisor) : fmax / divisor;
    float rounded = (float)exact;
    if ((double)rounded > exact) {
        uint32_t bits;
        memcpy(&bits, &rounded, 4);
        bits -= 1;
        memcpy(&rounded, &bits, 4);
    }
    return rounded;
}

extern "C" float hvx_component_limit(uint32_t metric, uint32_t dim) { return hvx::component_limit(metric, dim); }

// rows of `words` 32-bit words: gather (dst[i] = src[idx[i]]) or scatter (dst[idx[i]] = src[i])
__global__ void move_rows_kernel(const uint32_t *src, const uint32_t *idx, uint32_t *dst, uint32_t words, uint32_t n, uint32_t gather) {
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const size_t s0 = (size_t)(gather ? idx[i] : i) * words, d0 = (size_t)(gather ? i : idx[i]) * words;
    for (uint32_t w = threadIdx.x; w < words; w += blockDim.x) dst[d0 + w] = src[s0 + w];
}
static hipError_t launch_move_rows(const uint32_t *src, const uint32_t *idx, uint32_t *dst, uint32_t words, uint32_t n, bool gather,
                                   hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(move_rows_kernel, dim3(n), dim3(128), 0, s, src, idx, dst, words, n, gather ? 1u : 0u);
    return hipGetLastError();
}

template <typename F> static void parallel_for(uint64_t n, F f) {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 16) nt = 16;
    if (n < 4096) nt = 1;
    std::vector<std::thread> th;
    const uint64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        uint64_t lo = t * per, hi = std::min(n, lo + per);
        if (lo >= hi) break;
        th.emplace_back([=] { f(lo, hi); });
    }
    for (auto &x : th) x.join();
}

static void free_index(hvx_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream); // nothing in flight reads the scratch released below
    ix->allocs->device = ix->device;
    ix->allocs.reset(); // own scratch (and, for the root handle, its reference on the index image)
    ix->image.clear();
    if (ix->ev0) (void)hipEventDestroy(ix->ev0);
    if (ix->ev1) (void)hipEventDestroy(ix->ev1);
    if (ix->del_ev) (void)hipEventDestroy(ix->del_ev);
    for (hipEvent_t e : ix->ins_ev) if (e) (void)hipEventDestroy(e);
    if (ix->ins_stream) (void)hipStreamDestroy(ix->ins_stream);
    for (hipEvent_t e : ix->ring) (void)hipEventDestroy(e);
    if (ix->own_stream) (void)hipStreamDestroy(ix->own_stream);
    if (ix->h_pin) (void)hipHostFree(ix->h_pin);
    if (ix->h_flags) (void)hipHostFree(ix->h_flags);
    delete ix;
}

extern "C" void hvx_index_free(hvx_index *ix) { free_index(ix); }

extern "C" int hvx_index_sync(const hvx_index *ix) {
    if (!ix) return fail(HVX_ERR_INVARIANT, "null index");
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    return HVX_OK;
}

extern "C" int hvx_index_set_occupancy(hvx_index *ix, uint32_t queries_per_simd) {
    if (!ix) return fail(HVX_ERR_INVARIANT, "null index");
    if (queries_per_simd != 1 && queries_per_simd != 2) return fail(HVX_ERR_K_RANGE, "occupancy must be 1 or 2 queries per SIMD");
    std::lock_guard<std::mutex> lock(ix->mu);
    ix->occupancy = queries_per_simd;
    return HVX_OK;
}

// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order, and two
// execution lanes that land on one queue run their kernels back to back (r03: 0.65 -> 0.44 ms per step on the bf16 leg once its lanes
// had queues of their own).  A host with lanes + other handles + its own copy streams needs more than four; the runtime reads the
// variable when it initialises.  Round 5: the library no longer writes the environment from a load-time constructor
// (setenv races with a concurrent getenv in a multithreaded host, and silently changed the queue count of every other HIP user in
// the process).  The host calls hvx_runtime_prepare ONCE from its single-threaded start-up code, before its first HIP call
//  -- or exports the variable itself.  A value the host has already chosen is left alone.
extern "C" int hvx_runtime_prepare(uint32_t hw_queues) {
    if (hw_queues == 0) hw_queues = 8;
    if (hw_queues > 64) return fail(HVX_ERR_K_RANGE, "hardware queue count %u outside 1..64", hw_queues);
    char buf[16];
    snprintf(buf, sizeof(buf), "%u", hw_queues);
    if (setenv("GPU_MAX_HW_QUEUES", buf, /*overwrite=*/0) != 0) return fail(HVX_ERR_INVARIANT, "setenv(GPU_MAX_HW_QUEUES) failed");
    return HVX_OK;
}

extern "C" int hvx_index_set_option(hvx_index *ix, uint32_t option, uint32_t value) {
    if (!ix) return fail(HVX_ERR_INVARIANT, "null index");
    if (option >= HVX_OPT_COUNT) return fail(HVX_ERR_INVARIANT, "unknown option %u", option);
    if (option == HVX_OPT_WAVE_LOG2CAP && value != 0 && (value < 7 || value > 15)) return fail(HVX_ERR_K_RANGE, "visited-table size must be 2^7 .. 2^15 slots");
    if (option == HVX_OPT_FLAT_FIRST_CHUNK && value != 0 && value < 1024) return fail(HVX_ERR_K_RANGE, "the first chunk holds at least 1024 rows");
    if (option == HVX_OPT_HNSW_PAIR && value > 3) return fail(HVX_ERR_K_RANGE, "pair kernel selector is 0 (one query per SIMD handles), 1 (never), 2 (always) or 3 (always, one gatherer)");
    if (option == HVX_OPT_FLAT_TILE_BUILD && value > 4) return fail(HVX_ERR_K_RANGE, "tile build is 0 (default: two 256-thread workgroups per CU; fp8 rows on the MX-scaled fp8 matrix instruction), 1 (one 512-thread workgroup), 2 (512 threads, role-split), 3 (as 0) or 4 (as 0 with fp8 codes widened to bf16)");
    if (option == HVX_OPT_FLAT_NO_SMALLB && value > 2) return fail(HVX_ERR_K_RANGE, "small-batch selector is 0 (streaming kernels), 1 (never) or 2 (register-fragment build only)");
    std::lock_guard<std::mutex> lock(ix->mu);
    ix->opt[option] = value;
    return HVX_OK;
}

extern "C" int hvx_index_read_rows_device(const hvx_index *cix, uint64_t row0, uint64_t n, float *d_out) {
    if (!cix || (!d_out && n)) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    if (row0 > ix->dev.n || n > ix->dev.n - row0) return fail(HVX_ERR_INVARIANT, "rows [%llu, %llu) outside the index (%u rows)", (unsigned long long)row0,
                                                               (unsigned long long)(row0 + n), ix->dev.n);
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(launch_read_rows(ix->dev, row0, n, d_out, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    return HVX_OK;
}

extern "C" uint32_t hvx_index_last_scan_path(const hvx_index *ix) { return ix ? ix->last_scan_path : 0u; }

extern "C" void *hvx_index_stream(const hvx_index *ix) { return ix ? (void *)ix->stream : nullptr; }

extern "C" int hvx_index_set_stream(hvx_index *ix, void *hip_stream) {
    if (!ix) return fail(HVX_ERR_INVARIANT, "null index");
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->stream = hip_stream ? (hipStream_t)hip_stream : ix->own_stream;
    return HVX_OK;
}

int hvx_index::dalloc(void **p, size_t bytes) {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return fail(HVX_ERR_DEVICE, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    allocs->device = device;
    allocs->v.push_back(*p);
    return HVX_OK;
}

// scratch that is re-sized: the previous buffer is released first (hipFree waits for the device, so nothing in flight reads it)
int hvx_index::regrow(void **p, size_t bytes) {
    if (*p) {
        (void)hipFree(*p);
        allocs->v.erase(std::remove(allocs->v.begin(), allocs->v.end(), *p), allocs->v.end());
        *p = nullptr;
    }
    return dalloc(p, bytes);
}

static uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

extern "C" int hvx_index_import(const hvx_index_desc *desc, const uint64_t *node_ids, const float *vectors,
                                const uint64_t *l0_offsets, const uint64_t *l0_neighbors,
                                const uint16_t *level, const uint64_t *up_offsets,
                                const uint64_t *up_neighbors, hvx_index **out) {
    return hvx::import_index(desc, node_ids, vectors, l0_offsets, l0_neighbors, level, up_offsets, up_neighbors, 0, 0, out);
}

extern "C" int hvx_index_import_reserve(const hvx_index_desc *desc, const uint64_t *node_ids, const float *vectors, const uint64_t *l0_offsets,
                                        const uint64_t *l0_neighbors, const uint16_t *level, const uint64_t *up_offsets, const uint64_t *up_neighbors,
                                        uint64_t reserve_rows, uint64_t reserve_upper_rows, hvx_index **out) {
    if (desc && desc->dtype == HVX_FP8_E4M3 && (reserve_rows || reserve_upper_rows)) return fail(HVX_ERR_UNSUPPORTED, "fp8 images are read-only: no spare rows");
    return hvx::import_index(desc, node_ids, vectors, l0_offsets, l0_neighbors, level, up_offsets, up_neighbors, 0, 0, out, reserve_rows, reserve_upper_rows);
}

// min_s0 / min_su: lower bounds of the row strides (the device builder imports an empty graph and fills rows of up to
// m0 / m ids afterwards)
int hvx::import_index(const hvx_index_desc *desc, const uint64_t *node_ids, const float *vectors, const uint64_t *l0_offsets,
                      const uint64_t *l0_neighbors, const uint16_t *level, const uint64_t *up_offsets, const uint64_t *up_neighbors,
                      uint32_t min_s0, uint32_t min_su, hvx_index **out, uint64_t reserve_rows, uint64_t reserve_up_rows) {
    if (!desc || !out) return fail(HVX_ERR_INVARIANT, "null argument");
    *out = nullptr;
    if (desc->dim == 0) return fail(HVX_ERR_DIMENSION, "dimension must be non-zero");
    if (desc->metric > HVX_MANHATTAN) return fail(HVX_ERR_UNSUPPORTED, "unknown metric %u", desc->metric);
    if (desc->dtype != HVX_F32 && desc->dtype != HVX_BF16 && desc->dtype != HVX_FP8_E4M3)
        return fail(HVX_ERR_UNSUPPORTED, "unknown device dtype %u", desc->dtype);
    if (desc->dtype == HVX_FP8_E4M3 && (desc->dim % 128u != 0u || desc->float_kernel != HVX_KERNEL_AVX_FMA || desc->metric == HVX_MANHATTAN))
        return fail(HVX_ERR_UNSUPPORTED, "fp8 rows need dim %% 128 == 0, the AVX+FMA summation tree and an L2/cosine metric");
    if (desc->dtype == HVX_BF16 && (desc->dim % 64u != 0u || desc->float_kernel != HVX_KERNEL_AVX_FMA || desc->metric == HVX_MANHATTAN))
        return fail(HVX_ERR_UNSUPPORTED, "bf16 rows need dim %% 64 == 0, the AVX+FMA summation tree and an L2/cosine metric");
    if (desc->float_kernel > HVX_KERNEL_NEON)
        return fail(HVX_ERR_UNSUPPORTED, "float kernel %u is not one of the reference's FloatSimd kernels", desc->float_kernel);
    if (desc->n + reserve_rows >= (1ull << 31)) return fail(HVX_ERR_UNSUPPORTED, "shard too large (n < 2^31)");
    if ((reserve_rows || reserve_up_rows) && desc->dtype == HVX_FP8_E4M3) return fail(HVX_ERR_UNSUPPORTED, "fp8 images are read-only: they cannot reserve rows for later inserts");
    const uint64_t n = desc->n;
    const uint64_t cap = n + reserve_rows; // rows the arrays are allocated for (hvx_index_insert_batch appends into the spare ones)
    if (n && (!node_ids || !vectors || !l0_offsets)) return fail(HVX_ERR_INVARIANT, "null array");

    int dev = desc->device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipSetDevice(dev));

    hvx_index *ix = new hvx_index();
    ix->device = dev;
    ix->desc = *desc;
    ix->limit = component_limit(desc->metric, desc->dim);
    ix->max_batch = desc->max_batch ? desc->max_batch : 1024;
    DevIndex &d = ix->dev;
    memset(&d, 0, sizeof(d));
    d.n = (uint32_t)n;
    d.dim = desc->dim;
    d.ld = round_up(desc->dim, 4);
    d.metric = desc->metric;
    d.fkernel = desc->float_kernel;
    d.dim_main = kernel_dim_main(desc->float_kernel, desc->dim);
    d.max_layer = desc->max_layer;
    d.has_entry = (desc->has_entry && n) ? 1u : 0u;
    d.dtype = desc->dtype;
    if (desc->max_layer > 63) { free_index(ix); return fail(HVX_ERR_INVARIANT, "max_layer > 63"); }

    auto bail = [&](int code) { free_index(ix); return code; };
    if (hipStreamCreateWithFlags(&ix->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ix->ev0) != hipSuccess || hipEventCreate(&ix->ev1) != hipSuccess)
        return bail(fail(HVX_ERR_DEVICE, "stream/event creation failed"));
    ix->stream = ix->own_stream;

    // ---- ids: strictly ascending; contiguous ranges get an arithmetic id->index map ----
    ix->ids_p = std::make_shared<std::vector<uint64_t>>(node_ids, node_ids + n);
    ix->contiguous = true;
    for (uint64_t i = 1; i < n; ++i) {
        if (node_ids[i] <= node_ids[i - 1]) return bail(fail(HVX_ERR_INVARIANT, "node_ids must be strictly ascending (row %llu)", (unsigned long long)i));
        if (node_ids[i] != node_ids[0] + i) ix->contiguous = false;
    }
    if (d.has_entry) {
        uint32_t e = ix->find(desc->entry_point);
        if (e == kSentinel) return bail(fail(HVX_ERR_INVARIANT, "entry point %llu is not a row of this shard", (unsigned long long)desc->entry_point));
        d.entry = e;
    }

    // ---- graph: CSR over external ids -> fixed-stride rows of internal ids ----
    uint64_t max_deg = 0, up_rows = 0, max_up = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (l0_offsets[i + 1] < l0_offsets[i]) return bail(fail(HVX_ERR_INVARIANT, "l0_offsets not monotone"));
        max_deg = std::max<uint64_t>(max_deg, l0_offsets[i + 1] - l0_offsets[i]);
        if (level) up_rows += level[i];
    }
    if (level && up_rows && (!up_offsets || !up_neighbors)) return bail(fail(HVX_ERR_INVARIANT, "upper rows missing"));
    for (uint64_t r = 0; r < up_rows; ++r) {
        if (up_offsets[r + 1] < up_offsets[r]) return bail(fail(HVX_ERR_INVARIANT, "up_offsets not monotone"));
        max_up = std::max<uint64_t>(max_up, up_offsets[r + 1] - up_offsets[r]);
    }
    if (max_deg > 128 || max_up > 128) // before any narrowing: a huge row length must not wrap into a small stride
        return bail(fail(HVX_ERR_UNSUPPORTED, "neighbour rows longer than 128 are not supported (l0 %llu, upper %llu)", (unsigned long long)max_deg, (unsigned long long)max_up));
    // an image that declares its degree limits gets rows its nodes can grow into (deletes / upserts relink in place: a row may reach Mmax
    // although the persisted graph's longest row is shorter -- upper layers of small graphs)
    if (desc->m0 && desc->m0 <= 64u) min_s0 = std::max(min_s0, desc->m0);
    if (desc->m && desc->m <= 64u) { min_su = std::max(min_su, desc->m); min_s0 = std::max(min_s0, 2u * desc->m <= 64u ? 2u * desc->m : desc->m); }
    d.s0 = round_up((uint32_t)std::max<uint64_t>(std::max<uint64_t>(max_deg, min_s0), 1), 32);
    d.su = round_up((uint32_t)std::max<uint64_t>(std::max<uint64_t>(max_up, min_su), 1), 16);
    if (d.s0 > 128 || d.su > 128) return bail(fail(HVX_ERR_UNSUPPORTED, "neighbour rows longer than 128 are not supported (l0 %llu, upper %llu)", (unsigned long long)max_deg, (unsigned long long)max_up));

    std::vector<uint32_t> h_l0((size_t)cap * d.s0, kSentinel);
    std::atomic<int> graph_err{0};
    auto convert_row = [&](const uint64_t *src, uint64_t cnt, uint32_t *dst, uint64_t self) -> bool {
        uint64_t prev = 0;
        for (uint64_t t = 0; t < cnt; ++t) {
            // rows are canonical: ascending, deduped, self-free (neighbor_set.rs:1-9, values/vectors.rs:97-111)
            if (t && src[t] <= prev) return false;
            prev = src[t];
            uint32_t x = ix->find(src[t]);
            if (x == kSentinel || x == self) return false;
            dst[t] = x;
        }
        return true;
    };
    parallel_for(n, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i)
            if (!convert_row(l0_neighbors + l0_offsets[i], l0_offsets[i + 1] - l0_offsets[i], &h_l0[(size_t)i * d.s0], i))
                graph_err = 1;
    });
    if (graph_err) return bail(fail(HVX_ERR_INVARIANT, "layer-0 row is not canonical (ascending, deduped, self-free, known ids)"));

    const uint64_t cap_up = up_rows + reserve_up_rows;
    std::vector<uint32_t> h_up((size_t)std::max<uint64_t>(cap_up, 1) * d.su, kSentinel);
    std::vector<uint32_t> h_up_base(std::max<uint64_t>(cap, 1), kSentinel);
    std::vector<uint16_t> h_level(std::max<uint64_t>(cap, 1), 0);
    ix->cap_rows = cap;
    ix->cap_up_rows = cap_up;
    ix->up_rows_used = up_rows;
    {
        uint64_t r = 0;
        for (uint64_t i = 0; i < n; ++i) {
            uint16_t lv = level ? level[i] : 0;
            if (lv > 63) return bail(fail(HVX_ERR_INVARIANT, "node level > 63"));
            h_level[i] = lv;
            if (lv) {
                h_up_base[i] = (uint32_t)r;
                for (uint16_t l = 0; l < lv; ++l, ++r)
                    if (!convert_row(up_neighbors + up_offsets[r], up_offsets[r + 1] - up_offsets[r], &h_up[(size_t)r * d.su], i))
                        return bail(fail(HVX_ERR_INVARIANT, "upper row of node %llu is not canonical", (unsigned long long)node_ids[i]));
            }
        }
    }
    if (d.has_entry && h_level[d.entry] < d.max_layer)
        return bail(fail(HVX_ERR_INVARIANT, "entry point level %u below max_layer %u", h_level[d.entry], d.max_layer));

    // ---- upload ----
    void *p;
    int rc;
    const size_t vec_bytes = (size_t)n * d.ld * 4, vec_cap_bytes = (size_t)cap * d.ld * 4;
    const bool bf16 = desc->dtype == HVX_BF16, fp8 = desc->dtype == HVX_FP8_E4M3;
    float *staging = nullptr; // bf16 / fp8: temporary f32 copy, quantised in place, validated, packed, freed
    void *fp8_codes = nullptr;
    float *fp8_scale = nullptr;
    if (bf16 || fp8) {
        if (hipMalloc((void **)&staging, std::max<size_t>(vec_bytes, 16)) != hipSuccess) return bail(fail(HVX_ERR_DEVICE, "hipMalloc(%zu) staging", vec_bytes));
        p = staging;
    } else if ((rc = ix->dalloc(&p, vec_cap_bytes))) return bail(rc);
    d.vec = (const float *)p;
    auto bail_free = [&](int code) { if (staging) (void)hipFree(staging); return bail(code); };
    if (n) {
        // hipMemcpyDefault: `vectors` may be host memory or memory already resident on a device.  A device-to-device
        // hipMemcpy does not wait for the copy on the host, and the index's stream is non-blocking: the copy is ordered
        // on that stream and waited for, so that the kernels below (rounding, validation, packing) see every row
        hipError_t ce;
        if (d.ld == d.dim) {
            ce = hipMemcpyAsync(p, vectors, vec_bytes, hipMemcpyDefault, ix->stream);
        } else {
            ce = hipMemsetAsync(p, 0, vec_bytes, ix->stream);
            if (ce == hipSuccess)
                ce = hipMemcpy2DAsync(p, (size_t)d.ld * 4, vectors, (size_t)d.dim * 4, (size_t)d.dim * 4, n, hipMemcpyDefault, ix->stream);
        }
        if (ce == hipSuccess) ce = hipStreamSynchronize(ix->stream);
        if (ce != hipSuccess) return bail_free(fail(HVX_ERR_DEVICE, "vector upload failed: %s", hipGetErrorString(ce)));
        // bf16: the index IS the rounded vectors -- validation and cosine headers see the rounded values
        if (bf16 && launch_round_bf16_inplace(staging, (size_t)n * d.ld, ix->stream) != hipSuccess)
            return bail_free(fail(HVX_ERR_DEVICE, "bf16 rounding failed"));
    }
    if (fp8) { // quantise now: validation and headers below see the dequantised rows
        if ((rc = ix->dalloc(&fp8_codes, std::max<size_t>((size_t)n * d.dim, 16)))) return bail_free(rc);
        if ((rc = ix->dalloc((void **)&fp8_scale, std::max<size_t>(n, 1) * 4))) return bail_free(rc);
        if (launch_quantize_fp8(staging, (uint8_t *)fp8_codes, fp8_scale, (uint32_t)n, d.dim, ix->stream) != hipSuccess)
            return bail_free(fail(HVX_ERR_DEVICE, "fp8 quantisation failed"));
    }
    auto upload = [&](const void *src, size_t bytes, const void **dst) -> int {
        void *q;
        int r2 = ix->dalloc(&q, bytes);
        if (r2) return r2;
        if (bytes && hipMemcpy(q, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(HVX_ERR_DEVICE, "upload failed");
        *dst = q;
        return HVX_OK;
    };
    if ((rc = upload(h_l0.data(), h_l0.size() * 4, (const void **)&d.l0))) return bail_free(rc);
    if ((rc = upload(h_up.data(), h_up.size() * 4, (const void **)&d.up))) return bail_free(rc);
    if ((rc = upload(h_up_base.data(), h_up_base.size() * 4, (const void **)&d.up_base))) return bail_free(rc);
    if ((rc = upload(h_level.data(), h_level.size() * 2, (const void **)&d.level))) return bail_free(rc);
    {
        void *q;
        if ((rc = ix->dalloc(&q, std::max<size_t>(cap, 1) * 8))) return bail_free(rc);
        if (n && hipMemcpy(q, ix->ids_ref().data(), (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) return bail_free(fail(HVX_ERR_DEVICE, "upload failed"));
        d.ids = (const uint64_t *)q;
    }

    // ---- validate rows + headers ONCE on the device (decode_item_borrowed does it per fetch:
    //      mod.rs:889-949) ----
    float *d_hdr;
    uint32_t *d_rowstatus;
    if ((rc = ix->dalloc((void **)&d_hdr, std::max<size_t>(cap, 1) * 4))) return bail_free(rc);
    d.hdr = d_hdr;
    if (n) {
        if (hipMalloc((void **)&d_rowstatus, n * 4) != hipSuccess) return bail_free(fail(HVX_ERR_DEVICE, "hipMalloc row status"));
        // rows are laid out with stride ld; validate through a DevIndex view whose "dim" stride matches
        DevIndex view = d;
        hipError_t e = launch_validate_rows(view, (uint32_t)n, ix->limit, d_rowstatus, d_hdr, ix->stream);
        std::vector<uint32_t> st(n);
        if (e == hipSuccess) e = hipMemcpyAsync(st.data(), d_rowstatus, n * 4, hipMemcpyDeviceToHost, ix->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
        (void)hipFree(d_rowstatus);
        if (e != hipSuccess) return bail_free(fail(HVX_ERR_DEVICE, "row validation: %s", hipGetErrorString(e)));
        for (uint64_t i = 0; i < n; ++i)
            if (st[i]) return bail_free(fail((int)st[i], "stored vector of node %llu is invalid for this metric (status %u)", (unsigned long long)node_ids[i], st[i]));
    }
    if (bf16) {
        void *pb;
        if ((rc = ix->dalloc(&pb, std::max<size_t>((size_t)cap * d.dim * 2, 16)))) return bail_free(rc); // (cap: spare rows of hvx_index_import_reserve)
        hipError_t e = launch_pack_bf16(staging, (uint16_t *)pb, (uint32_t)n, d.dim, ix->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
        (void)hipFree(staging);
        staging = nullptr;
        if (e != hipSuccess) return bail(fail(HVX_ERR_DEVICE, "bf16 packing: %s", hipGetErrorString(e)));
        d.vecb = (const uint16_t *)pb;
        d.vec = nullptr;
        // |x|^2 per row and its maximum: score term and error bound of the MFMA exact scan
        if ((rc = ix->dalloc((void **)&ix->m_rowterm, std::max<size_t>(cap, 1) * 4))) return bail(rc);
        if (n) {
            std::vector<float> h_n2(n);
            e = launch_bf16_row_norm2(d.vecb, (uint32_t)n, d.dim, ix->m_rowterm, ix->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(h_n2.data(), ix->m_rowterm, n * 4, hipMemcpyDeviceToHost, ix->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
            if (e != hipSuccess) return bail(fail(HVX_ERR_DEVICE, "row norms: %s", hipGetErrorString(e)));
            for (float v : h_n2) ix->m_xmax2 = std::max(ix->m_xmax2, v);
        }
        ix->rowterm_rows = (uint32_t)n;
    }
    if (fp8) {
        if ((rc = ix->dalloc((void **)&ix->m_rowterm, std::max<size_t>(n, 1) * 4))) return bail_free(rc);
        hipError_t e = hipSuccess;
        std::vector<float> h_n2(n);
        if (n) {
            e = launch_f32_row_norm2(staging, (uint32_t)n, d.ld, d.dim, ix->m_rowterm, ix->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(h_n2.data(), ix->m_rowterm, n * 4, hipMemcpyDeviceToHost, ix->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
        }
        (void)hipFree(staging);
        staging = nullptr;
        if (e != hipSuccess) return bail(fail(HVX_ERR_DEVICE, "fp8 row norms: %s", hipGetErrorString(e)));
        for (float v : h_n2) ix->m_xmax2 = std::max(ix->m_xmax2, v);
        ix->rowterm_rows = (uint32_t)n;
        d.vec8 = (const uint8_t *)fp8_codes;
        d.rowscale = fp8_scale;
        d.vec = nullptr;
    }

    // ---- per-batch scratch ----
    const uint32_t mb = ix->max_batch;
    ix->words_per_query = round_up((uint32_t)((cap + 31) / 32), 4);
    if (ix->words_per_query == 0) ix->words_per_query = 4;
    if ((rc = ix->dalloc((void **)&ix->d_bitmap, (size_t)mb * ix->words_per_query * 4))) return bail(rc);
    if (hipMemset(ix->d_bitmap, 0, (size_t)mb * ix->words_per_query * 4) != hipSuccess) return bail(fail(HVX_ERR_DEVICE, "bitmap clear failed"));
    if ((rc = ix->dalloc((void **)&ix->d_qstatus, (size_t)mb * 4))) return bail(rc);
    if ((rc = ix->dalloc((void **)&ix->d_qhdr, (size_t)mb * 4))) return bail(rc);
    if ((rc = ix->dalloc((void **)&ix->d_tie, ((size_t)mb * 2 + 4) * 4))) return bail(rc); // [mb] flags, [mb] re-run list, re-run count / done
    if (hipMemset(ix->d_tie, 0, ((size_t)mb * 2 + 4) * 4) != hipSuccess) return bail(fail(HVX_ERR_DEVICE, "re-run list clear failed"));
    if ((rc = ix->dalloc((void **)&ix->d_qstats, (size_t)mb * sizeof(hvx_query_stats)))) return bail(rc);
    ix->publish_view();
    *out = ix;
    return HVX_OK;
}

// ---- generation view of a growable image (hvx_host.h) ----
void hvx_index::publish_view(bool bump) {
    std::lock_guard<std::mutex> g(shared->mu);
    if (bump) shared->visible_seq += 1;
    shared->v_n = dev.n;
    shared->v_entry = dev.entry;
    shared->v_max_layer = dev.max_layer;
    shared->v_has_entry = dev.has_entry;
    shared->v_entry_point = desc.entry_point;
    shared->v_contiguous = contiguous;
    shared->v_ids = ids_p;
    shared->v_dead = dead_p;
    shared->v_n_dead = n_dead;
    shared->v_dead_dev = dev.dead;
    seen_seq = shared->visible_seq;
}
bool hvx_index::adopt_view() {
    std::lock_guard<std::mutex> g(shared->mu);
    if (seen_seq == shared->visible_seq) return false;
    dev.n = shared->v_n;
    dev.entry = shared->v_entry;
    dev.max_layer = shared->v_max_layer;
    dev.has_entry = shared->v_has_entry;
    desc.n = shared->v_n;
    desc.has_entry = shared->v_has_entry;
    desc.max_layer = shared->v_max_layer;
    desc.entry_point = shared->v_entry_point;
    contiguous = shared->v_contiguous;
    ids_p = shared->v_ids;
    dead_p = shared->v_dead;
    n_dead = shared->v_n_dead;
    dev.dead = shared->v_dead_dev;
    seen_seq = shared->visible_seq;
    return true;
}

// the ascending list of live rows of this handle's generation (exact scans over "all rows" of an image with deleted nodes)
int hvx_index::ensure_live() {
    if (f_live && live_valid && live_for == dead_p && live_rows_n == dev.n - n_dead) return HVX_OK;
    std::vector<uint32_t> rows;
    rows.reserve(dev.n);
    const std::vector<uint8_t> *dd = dead_p.get();
    for (uint32_t i = 0; i < dev.n; ++i)
        if (!dd || i >= dd->size() || !(*dd)[i]) rows.push_back(i);
    if (rows.size() > cap_live || !f_live) {
        const size_t cap = std::max<size_t>(std::max<size_t>(rows.size(), cap_rows), 1);
        int rc = regrow((void **)&f_live, cap * 4);
        if (rc) return rc;
        cap_live = (uint32_t)cap;
    }
    if (!rows.empty()) {
        if (hipMemcpyAsync(f_live, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
            return hvx::fail(HVX_ERR_DEVICE, "upload of the live-row list failed");
    }
    live_for = dead_p;
    live_valid = true;
    live_rows_n = (uint32_t)rows.size();
    return HVX_OK;
}

extern "C" int hvx_index_refresh(hvx_index *ix) {
    if (!ix) return fail(HVX_ERR_INVARIANT, "null index");
    std::lock_guard<std::mutex> lock(ix->mu);
    (void)ix->adopt_view(); // per-image caches (row norms, bf16 shadow, SimHash directory) extend themselves on their next use
    return HVX_OK;
}
extern "C" uint64_t hvx_index_visible_seq(const hvx_index *ix) { return ix ? ix->seen_seq : 0; }
extern "C" uint64_t hvx_index_rows(const hvx_index *ix) { return ix ? ix->dev.n : 0; }
extern "C" uint64_t hvx_index_row_capacity(const hvx_index *ix) { return ix ? ix->cap_rows : 0; }
extern "C" uint64_t hvx_index_live_rows(const hvx_index *ix) { return ix ? ix->live_rows() : 0; }
extern "C" int hvx_index_contains(const hvx_index *ix, uint64_t node_id) { return ix && ix->find(node_id) != kSentinel ? 1 : 0; }

// An execution lane on the same index image: own stream, events, per-batch scratch and mutex; rows, graph, ids, headers
// and SimHash rows are shared with (and kept alive by) the handle it was forked from.
extern "C" int hvx_index_fork(const hvx_index *parent, hvx_index **out) {
    if (!parent || !out) return fail(HVX_ERR_INVARIANT, "null argument");
    *out = nullptr;
    hvx_index *src = const_cast<hvx_index *>(parent);
    std::lock_guard<std::mutex> lock(src->mu);
    HIP_TRY(hipSetDevice(src->device));
    hvx_index *ix = new hvx_index();
    ix->device = src->device;
    ix->desc = src->desc;
    ix->dev = src->dev;
    ix->limit = src->limit;
    ix->max_batch = src->max_batch;
    ix->words_per_query = src->words_per_query;
    ix->ids_p = src->ids_p;
    ix->contiguous = src->contiguous;
    ix->dead_p = src->dead_p;
    ix->n_dead = src->n_dead;
    ix->cap_rows = src->cap_rows;
    ix->cap_up_rows = src->cap_up_rows;
    ix->up_rows_used = src->up_rows_used;
    ix->seen_seq = src->seen_seq;
    ix->occupancy = src->occupancy;
    memcpy(ix->opt, src->opt, sizeof(ix->opt));
    ix->is_fork = true;
    ix->seen_rewrite = src->seen_rewrite;
    ix->image = src->image;
    ix->image.push_back(src->allocs);
    ix->m_rowterm = src->m_rowterm;
    ix->rowterm_rows = src->rowterm_rows;
    ix->m_xmax2 = src->m_xmax2;
    ix->shared = src->shared; // one bf16 shadow (and whatever else is built lazily per image) for all lanes
    ix->m_shadow = src->m_shadow;
    ix->has_simhash = src->has_simhash;
    ix->sh_cfg = src->sh_cfg;
    ix->d_node_hash = src->d_node_hash;
    ix->d_planes_t = src->d_planes_t;
    auto bail = [&](int code) { free_index(ix); return code; };
    if (hipStreamCreateWithFlags(&ix->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ix->ev0) != hipSuccess || hipEventCreate(&ix->ev1) != hipSuccess)
        return bail(fail(HVX_ERR_DEVICE, "stream/event creation failed"));
    ix->stream = ix->own_stream;
    int rc;
    const uint32_t mb = ix->max_batch;
    const size_t bm_bytes = (size_t)mb * ix->words_per_query * 4;
    if ((rc = ix->dalloc((void **)&ix->d_bitmap, bm_bytes))) return bail(rc);
    if (hipMemset(ix->d_bitmap, 0, bm_bytes) != hipSuccess) return bail(fail(HVX_ERR_DEVICE, "bitmap clear failed"));
    if ((rc = ix->dalloc((void **)&ix->d_qstatus, (size_t)mb * 4))) return bail(rc);
    if ((rc = ix->dalloc((void **)&ix->d_qhdr, (size_t)mb * 4))) return bail(rc);
    if ((rc = ix->dalloc((void **)&ix->d_tie, ((size_t)mb * 2 + 4) * 4))) return bail(rc); // [mb] flags, [mb] re-run list, re-run count / done
    if (hipMemset(ix->d_tie, 0, ((size_t)mb * 2 + 4) * 4) != hipSuccess) return bail(fail(HVX_ERR_DEVICE, "re-run list clear failed"));
    if ((rc = ix->dalloc((void **)&ix->d_qstats, (size_t)mb * sizeof(hvx_query_stats)))) return bail(rc);
    if (ix->has_simhash) { // per-batch state of the non-strict arms (hvx_params.hip)
        if ((rc = ix->dalloc((void **)&ix->d_qhash, (size_t)mb * 8))) return bail(rc);
        if ((rc = ix->dalloc((void **)&ix->d_thr_break, 64 * 4))) return bail(rc);
        if ((rc = ix->dalloc((void **)&ix->d_astats, (size_t)mb * sizeof(hvx_adaptive_stats)))) return bail(rc);
        if (!ix->sh_cfg.resident_snapshot) {
            if ((rc = ix->dalloc((void **)&ix->d_bitmap2, bm_bytes))) return bail(rc);
            if (hipMemset(ix->d_bitmap2, 0, bm_bytes) != hipSuccess) return bail(fail(HVX_ERR_DEVICE, "bitmap clear failed"));
        }
    }
    *out = ix;
    return HVX_OK;
}

// ---------------------------------------------------------------------------------------------
// staging buffers of the host-pointer API (grown on demand, owned by the index)
// ---------------------------------------------------------------------------------------------
int hvx_index::stage(uint32_t b, uint32_t k) {
    const size_t need_q = (size_t)b * dev.dim * 4, need_o = (size_t)b * k;
    int rc;
    if (need_q > cap_q) {
        if ((rc = regrow((void **)&s_queries, need_q))) return rc;
        cap_q = need_q;
    }
    if (need_o > cap_o) {
        if ((rc = regrow((void **)&s_ids, need_o * 8))) return rc;
        if ((rc = regrow((void **)&s_scores, need_o * 4))) return rc;
        cap_o = need_o;
    }
    if (b > cap_b) {
        if ((rc = regrow((void **)&s_counts, (size_t)b * 4))) return rc;
        if ((rc = regrow((void **)&s_status, (size_t)b * 4))) return rc;
        cap_b = b;
    }
    return pin(((need_q + 63u) & ~(size_t)63u) + need_o * 12 + (size_t)b * 8); // the pinned mirror holds one batch: queries, then the four outputs
}

static size_t pin_off_ids(const hvx_index *ix, uint32_t cb) { return (((size_t)cb * ix->dev.dim * 4) + 63u) & ~(size_t)63u; }
int hvx_index::pin(size_t bytes) {
    if (bytes <= cap_pin) return HVX_OK;
    if (h_pin) {
        HIP_TRY(hipStreamSynchronize(stream)); // nothing enqueued still reads or writes the old mirror
        (void)hipHostFree(h_pin);
        h_pin = nullptr;
        cap_pin = 0;
    }
    const size_t want = std::max<size_t>(bytes + bytes / 2, 1u << 16);
    if (hipHostMalloc((void **)&h_pin, want, hipHostMallocDefault) != hipSuccess) return fail(HVX_ERR_DEVICE, "hipHostMalloc(%zu) staging mirror", want);
    cap_pin = want;
    return HVX_OK;
}
int hvx_index::pin_flags(size_t words) {
    if (words <= cap_flags) return HVX_OK;
    if (h_flags) {
        HIP_TRY(hipStreamSynchronize(stream));
        (void)hipHostFree(h_flags);
        h_flags = nullptr;
        cap_flags = 0;
    }
    const size_t want = std::max<size_t>(words + words / 2, 4096);
    if (hipHostMalloc((void **)&h_flags, want * 4, hipHostMallocDefault) != hipSuccess) return fail(HVX_ERR_DEVICE, "hipHostMalloc(%zu) read-back mirror", want * 4);
    cap_flags = want;
    return HVX_OK;
}
int hvx_index::stage_in(const float *queries, uint32_t cb) {
    const size_t qbytes = (size_t)cb * dev.dim * 4;
    int rc = pin(qbytes);
    if (rc) return rc;
    memcpy(h_pin, queries, qbytes);
    HIP_TRY(hipMemcpyAsync(s_queries, h_pin, qbytes, hipMemcpyHostToDevice, stream));
    return HVX_OK;
}
int hvx_index::stage_out(uint32_t cb, uint32_t k) {
    const size_t o0 = pin_off_ids(this, cb), n = (size_t)cb * k;
    int rc = pin(o0 + n * 12 + (size_t)cb * 8);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h_pin + o0, s_ids, n * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_pin + o0 + n * 8, s_scores, n * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_pin + o0 + n * 12, s_counts, (size_t)cb * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_pin + o0 + n * 12 + (size_t)cb * 4, s_status, (size_t)cb * 4, hipMemcpyDeviceToHost, stream));
    return HVX_OK;
}
void hvx_index::deliver(uint32_t cb, uint32_t k, uint64_t *out_ids, float *out_scores, uint32_t *out_counts, uint32_t *out_status) const {
    const size_t o0 = pin_off_ids(this, cb), n = (size_t)cb * k;
    memcpy(out_ids, h_pin + o0, n * 8);
    memcpy(out_scores, h_pin + o0 + n * 8, n * 4);
    memcpy(out_counts, h_pin + o0 + n * 12, (size_t)cb * 4);
    memcpy(out_status, h_pin + o0 + n * 12 + (size_t)cb * 4, (size_t)cb * 4);
}

int hvx::check_k_ef(uint32_t k, uint32_t ef) {
    // ResultCount / SearchBeamWidth (parameters.rs:100-133)
    if (k == 0) return fail(HVX_ERR_K_RANGE, "result count must be non-zero");
    if (ef < k) return fail(HVX_ERR_K_RANGE, "search beam width %u is below the result count %u", ef, k);
    // SearchBeamWidth::try_new has no upper bound (parameters.rs:118-133).  Beams of ef + 32 <= 1024 entries run on the HNSW kernels; a
    // wider beam is answered by the EXACT scan of the index (enqueue_search): the true top-k, which is what a beam of that width
    // converges to -- the same fall-back the restricted path takes beyond its LDS limits.  The scan serves k <= 1024.
    if (ef > 992u && k > 1024) return fail(HVX_ERR_UNSUPPORTED, "result count %u exceeds the exact scan's limit of 1024 (beams beyond ef 992 are answered by the exact scan)", k);
    return HVX_OK;
}

// per-query counters of a search that was answered by the exact scan: every stored row loaded and scored once
__global__ void exact_fallback_stats_kernel(hvx_query_stats *qs, uint32_t *tie, const uint32_t *status, uint32_t b, uint32_t n) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= b) return;
    const uint32_t rows = (status && status[q]) ? 0u : n;
    if (qs) qs[q] = hvx_query_stats{0u, 0u, rows, rows};
    if (tie) tie[q] = 0u;
}

static void add_stats(hvx_stats *stats, const std::vector<hvx_query_stats> &qs, const std::vector<uint32_t> &tie,
                      float ms) {
    stats->queries += qs.size();
    for (size_t i = 0; i < qs.size(); ++i) {
        stats->expansion_steps += qs[i].expansion_steps;
        stats->neighbors_examined += qs[i].neighbors_examined;
        stats->vectors_loaded += qs[i].vectors_loaded;
        stats->distance_computations += qs[i].distance_computations;
        if (i < tie.size() && tie[i]) stats->tie_overflow_queries += 1;
    }
    stats->device_ms += ms;
}

// enqueue validation + memset + search kernel for one chunk of <= max_batch queries
// the event pair a search kernel is bracketed with: the synchronous stats pair, else the next slot of the timing ring
static void pick_events(hvx_index *ix, bool timed, hipEvent_t *e0, hipEvent_t *e1, unsigned long long **wclk = nullptr) {
    if (timed) { *e0 = ix->ev0; *e1 = ix->ev1; return; }
    if (ix->ring_n < ix->ring_cap) {
        *e0 = ix->ring[2 * ix->ring_n];
        *e1 = ix->ring[2 * ix->ring_n + 1];
        if (wclk && ix->d_wclk && ix->ring_n < ix->wclk_cap) *wclk = ix->d_wclk + (size_t)ix->ring_n * ix->max_batch * 2;
        ix->ring_n += 1;
    }
}

int hvx::enqueue_search(const hvx_index *cix, const float *d_queries, uint32_t b, uint32_t k, uint32_t ef,
                        uint64_t *d_ids, float *d_scores, uint32_t *d_counts, uint32_t *d_status,
                        hvx_query_stats *d_qstats, bool timed, const AdaptArgs *ad) {
    hvx_index *ix = const_cast<hvx_index *>(cix);
    ix->sync_rewrites();
    if (ef > 992u) { // beyond the widest beam of the HNSW kernels: the exact scan (see check_k_ef); validation happens inside it
        int rc = flat_scan_device(ix, d_queries, b, k, nullptr, ix->dev.n, d_ids, d_scores, d_counts, d_status ? d_status : ix->d_qstatus, timed);
        if (rc) return rc;
        hipLaunchKernelGGL(exact_fallback_stats_kernel, dim3((b + 255u) / 256u), dim3(256), 0, ix->stream, d_qstats ? d_qstats : ix->d_qstats, ix->d_tie,
                           d_status ? d_status : ix->d_qstatus, b, ix->live_rows());
        HIP_TRY(hipGetLastError());
        return HVX_OK;
    }
    HIP_TRY(launch_validate_queries(ix->dev, d_queries, b, ix->limit, ix->d_qstatus, ix->d_qhdr, ix->stream));
    HnswArgs a;
    a.ix = ix->dev;
    a.queries = d_queries;
    a.qstatus = ix->d_qstatus;
    a.qhdr = ix->d_qhdr;
    a.bitmap = ix->d_bitmap;
    a.bitmap2 = (ad && ad->count_reads) ? ix->d_bitmap2 : nullptr;
    a.words_per_query = ix->words_per_query;
    a.k = k;
    a.ef = ef;
    a.out_ids = d_ids;
    a.out_scores = d_scores;
    a.out_counts = d_counts;
    a.out_status = d_status;
    a.qstats = d_qstats ? d_qstats : ix->d_qstats;
    a.tie_flags = ix->d_tie;
    a.rerun_list = ix->d_tie + ix->max_batch;
    a.rerun_ctl = ix->d_tie + 2 * (size_t)ix->max_batch;
    a.prof = nullptr;
    a.wave_clock = nullptr;
    a.adaptive = ad ? 1u : 0u;
    a.ad = ad ? *ad : AdaptArgs{};
    a.build_nodes = nullptr;
    a.build_ef_upper = 0;
    a.only_flagged = 0;
    a.occupancy = ix->occupancy;
    a.pair = ix->opt[HVX_OPT_HNSW_PAIR] >= 2u || (ix->opt[HVX_OPT_HNSW_PAIR] == 0u && ix->occupancy != 2u) ? 1u : 0u;
    a.pair_gatherers = ix->opt[HVX_OPT_HNSW_PAIR] == 3u ? 1u : 0u;
    a.log2cap = ix->opt[HVX_OPT_WAVE_LOG2CAP];
    if (ad) {
        if (!hnsw_wave_adaptive_supported(a))
            return fail(HVX_ERR_UNSUPPORTED, "the non-strict search arms serve f32 rows of any dimension / metric (bf16 rows: dim in "
                        "{128,256,512,768,1024,1536}, cosine / Euclidean), neighbour rows <= 64 ids, ef <= 800");
        if (ix->bitmap_dirty) {
            HIP_TRY(hipMemsetAsync(ix->d_bitmap, 0, (size_t)ix->max_batch * ix->words_per_query * 4, ix->stream));
            ix->bitmap_dirty = false;
        }
        const bool ad_prof = tuning_env("HVX_WAVE_PROF") != nullptr && hnsw_wave_supported(a) && ix->occupancy != 2u;
        if (ad_prof) {
            if (!ix->d_prof && ix->dalloc((void **)&ix->d_prof, (size_t)ix->max_batch * 64)) return HVX_ERR_DEVICE;
            a.prof = ix->d_prof;
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        pick_events(ix, timed, &e0, &e1, &a.wave_clock);
        if (e0) HIP_TRY(hipEventRecord(e0, ix->stream));
        HIP_TRY(launch_hnsw_wave(a, b, ix->stream));
        if (e1) HIP_TRY(hipEventRecord(e1, ix->stream));
        if (ad_prof) {
            std::vector<unsigned long long> h((size_t)b * 8);
            HIP_TRY(hipMemcpyAsync(h.data(), ix->d_prof, h.size() * 8, hipMemcpyDeviceToHost, ix->stream));
            HIP_TRY(hipStreamSynchronize(ix->stream));
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t i = 0; i < b; ++i)
                for (int j = 0; j < 8; ++j) acc[j] += (double)h[(size_t)i * 8 + j];
            fprintf(stderr, "[hvx prof, non-strict] per query: row-wait %.0f  visited-claim %.0f  decide+select %.0f  gather+fma %.0f  predict %.0f  admit %.0f  (layer-0 loop %.0f) cycles\n",
                    acc[0] / b, acc[1] / b, acc[6] / b, acc[2] / b, acc[3] / b, acc[4] / b, acc[7] / b);
        }
        return HVX_OK;
    }
    const bool prof = tuning_env("HVX_WAVE_PROF") != nullptr; // tuning builds: phase-timing kernel + stderr report
    if (prof) {
        if (!ix->d_prof && ix->dalloc((void **)&ix->d_prof, (size_t)ix->max_batch * 64)) return HVX_ERR_DEVICE;
        a.prof = ix->d_prof;
    }
    // the HBM visited bitmap is all-zero at import; the general kernel dirties it, the wave kernel
    // (LDS visited set, bitmap only as overflow) hands it back zeroed
    const bool force_general = ix->opt[HVX_OPT_HNSW_GENERAL_KERNEL] != 0;
    const bool wave = !(force_general && ix->dev.dtype == HVX_F32) && hnsw_wave_supported(a);
    if (!wave && ix->dev.dtype != HVX_F32)
        return fail(HVX_ERR_UNSUPPORTED, ix->dev.dtype == HVX_FP8_E4M3 ? "fp8 rows serve the exact scan only (HNSW over fp8 rows is not built)"
                    : "bf16 rows are served by the one-wavefront-per-query kernel only (dim in {128,256,512,768,1024,1536}, rows <= 64 ids, ef <= 800)");
    if (ix->bitmap_dirty) {
        HIP_TRY(hipMemsetAsync(ix->d_bitmap, 0, (size_t)ix->max_batch * ix->words_per_query * 4, ix->stream));
        ix->bitmap_dirty = false;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr; // device_ms = the search kernel alone
    pick_events(ix, timed, &e0, &e1, &a.wave_clock);
    if (e0) HIP_TRY(hipEventRecord(e0, ix->stream));
    if (wave) {
        HIP_TRY(launch_hnsw_wave(a, b, ix->stream));
        if (prof) {
            std::vector<unsigned long long> h((size_t)b * 8);
            HIP_TRY(hipMemcpyAsync(h.data(), ix->d_prof, h.size() * 8, hipMemcpyDeviceToHost, ix->stream));
            HIP_TRY(hipStreamSynchronize(ix->stream));
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t i = 0; i < b; ++i)
                for (int j = 0; j < 8; ++j) acc[j] += (double)h[(size_t)i * 8 + j];
            fprintf(stderr, "[hvx prof] per query: row-wait %.0f  visited %.0f  gather+fma %.0f  predict %.0f  admit %.0f  (layer-0 loop %.0f) cycles;"
                            " fresh-candidate predictions %.1f  row-prefetch hits %.1f\n",
                    acc[0] / b, acc[1] / b, acc[2] / b, acc[3] / b, acc[4] / b, acc[7] / b, acc[5] / b, acc[6] / b);
        }
    } else {
        HIP_TRY(launch_hnsw_search(a, b, ix->stream));
        ix->bitmap_dirty = true;
    }
    if (e1) HIP_TRY(hipEventRecord(e1, ix->stream));
    return HVX_OK;
}

// ---- asynchronous kernel timing (hvx_index_timing_begin / _collect) ----
extern "C" int hvx_index_timing_begin(hvx_index *ix, uint32_t capacity) {
    if (!ix) return fail(HVX_ERR_INVARIANT, "null index");
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    while (ix->ring.size() < (size_t)capacity * 2) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        ix->ring.push_back(e);
    }
    if (capacity > ix->wclk_cap) { // per-wavefront start / end clocks of the same launches (hvx_index_wave_clocks)
        int rc = ix->regrow((void **)&ix->d_wclk, (size_t)capacity * ix->max_batch * 2 * sizeof(unsigned long long));
        if (rc) return rc;
        ix->wclk_cap = capacity;
    }
    if (ix->d_wclk) HIP_TRY(hipMemsetAsync(ix->d_wclk, 0, (size_t)ix->wclk_cap * ix->max_batch * 2 * sizeof(unsigned long long), ix->stream));
    ix->ring_cap = capacity;
    ix->ring_n = 0;
    return HVX_OK;
}

extern "C" int hvx_index_wave_clocks(hvx_index *ix, uint64_t *out, uint32_t cap_launches, uint32_t rows_per_launch, uint32_t *out_n) {
    if (!ix || !out || !out_n) return fail(HVX_ERR_INVARIANT, "null argument");
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    const uint32_t n = std::min(std::min(ix->ring_n, ix->wclk_cap), cap_launches);
    const uint32_t rows = std::min(rows_per_launch, ix->max_batch);
    const size_t rb = 2 * sizeof(unsigned long long);
    // device layout [launch][max_batch][2]; the caller's [launch][rows_per_launch][2]
    if (n && rows)
        HIP_TRY(hipMemcpy2D(out, (size_t)rows_per_launch * rb, ix->d_wclk, (size_t)ix->max_batch * rb, (size_t)rows * rb, n, hipMemcpyDeviceToHost));
    *out_n = n;
    return HVX_OK;
}

extern "C" int hvx_index_timing_collect(hvx_index *ix, float *out_ms, uint32_t cap, uint32_t *out_n) {
    if (!ix || !out_n) return fail(HVX_ERR_INVARIANT, "null argument");
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    const uint32_t n = std::min(ix->ring_n, cap);
    for (uint32_t i = 0; i < n; ++i) HIP_TRY(hipEventElapsedTime(&out_ms[i], ix->ring[2 * i], ix->ring[2 * i + 1]));
    *out_n = n;
    ix->ring_cap = 0;
    ix->ring_n = 0;
    return HVX_OK;
}

int hvx::collect_stats(hvx_index *ix, uint32_t b, const hvx_query_stats *d_qstats, hvx_stats *stats) {
    std::vector<hvx_query_stats> qs(b);
    std::vector<uint32_t> tie(b);
    HIP_TRY(hipMemcpyAsync(qs.data(), d_qstats ? d_qstats : ix->d_qstats, (size_t)b * sizeof(hvx_query_stats), hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipMemcpyAsync(tie.data(), ix->d_tie, (size_t)b * 4, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
    add_stats(stats, qs, tie, ms);
    return HVX_OK;
}

extern "C" int hvx_search_batch_device(const hvx_index *cix, const float *d_queries, uint32_t b, uint32_t k,
                                       uint32_t ef, uint64_t *d_out_ids, float *d_out_scores,
                                       uint32_t *d_out_counts, uint32_t *d_out_status,
                                       hvx_query_stats *d_query_stats, hvx_stats *stats) {
    if (!cix) return fail(HVX_ERR_INVARIANT, "null index");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    int rc = check_k_ef(k, ef);
    if (rc) return rc;
    if (b == 0) return HVX_OK;
    if (b > ix->max_batch) return fail(HVX_ERR_UNSUPPORTED, "batch %u exceeds max_batch %u given at import", b, ix->max_batch);
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    rc = enqueue_search(ix, d_queries, b, k, ef, d_out_ids, d_out_scores, d_out_counts, d_out_status, d_query_stats, stats != nullptr);
    if (rc) return rc;
    if (stats) return collect_stats(ix, b, d_query_stats, stats);
    return HVX_OK;
}

extern "C" int hvx_search_batch(const hvx_index *cix, const float *queries, uint32_t b, uint32_t k, uint32_t ef,
                                uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                                uint32_t *out_status, hvx_stats *stats) {
    if (!cix) return fail(HVX_ERR_INVARIANT, "null index");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    int rc = check_k_ef(k, ef);
    if (rc) return rc;
    if (b == 0) return HVX_OK;
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    const uint32_t mb = ix->max_batch;
    std::vector<uint32_t> status(b, 0);
    for (uint32_t c0 = 0; c0 < b; c0 += mb) {
        const uint32_t cb = std::min(mb, b - c0);
        if ((rc = ix->stage(cb, k))) return rc;
        if ((rc = ix->stage_in(queries + (size_t)c0 * ix->dev.dim, cb))) return rc;
        rc = enqueue_search(ix, ix->s_queries, cb, k, ef, ix->s_ids, ix->s_scores, ix->s_counts, ix->s_status, nullptr, stats != nullptr);
        if (rc) return rc;
        if ((rc = ix->stage_out(cb, k))) return rc;
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

// ---------------------------------------------------------------------------------------------
// exact scans
// ---------------------------------------------------------------------------------------------
int hvx_index::flat_scratch(uint32_t b, uint32_t k, uint32_t chunk_rows) {
    int rc;
    const size_t need_d = (size_t)b * chunk_rows * 4;
    if (need_d > cap_dist) {
        if ((rc = regrow((void **)&f_dist, need_d))) return rc;
        cap_dist = need_d;
    }
    const size_t need_t = (size_t)b * k;
    if (need_t > cap_top) {
        if ((rc = regrow((void **)&f_top_s, need_t * 4))) return rc;
        if ((rc = regrow((void **)&f_top_i, need_t * 4))) return rc;
        cap_top = need_t;
    }
    if (b > cap_topc) {
        if ((rc = regrow((void **)&f_top_c, (size_t)b * 4))) return rc;
        cap_topc = b;
    }
    return HVX_OK;
}

// Does an exact scan of this shape produce its candidates on the matrix cores?  bf16 / fp8 rows: always (their only pipeline).  f32 rows:
// the AVX+FMA tree over one of the unrolled dimensions >= 256 (below that the top-(m+1) selection over the score matrix outweighs the
// contraction), and enough work to amortise the extra passes (b x rows x dim >= 2^33 MACs) -- or a small batch (b <= 128) over enough
// rows (rows x dim >= 2^22) to make it a stream: the one-pass kernels of hvx_flat_smallb.hip read every row once at HBM speed, where the
// reference-order VALU kernel is bound by the b x rows x dim subtract / FMA pairs.
bool hvx::flat_scan_on_matrix_cores(const hvx_index *ix, uint32_t b, uint32_t k, uint32_t n_rows) {
    const DevIndex &d = ix->dev;
    if (d.dtype != HVX_F32) return true;
    const uint32_t nk = d.dim >> 5;
    const bool shape = d.dim % 32u == 0u && d.ld == d.dim && d.dim_main == d.dim && d.fkernel == kKernelAvxFma &&
                       (nk == 4 || nk == 8 || nk == 16 || nk == 24 || nk == 32 || nk == 48) && (d.metric == kL2 || d.metric == kCosine);
    const bool big = (uint64_t)b * n_rows * d.dim >= (1ull << 33);
    const bool small_stream = ix->opt[HVX_OPT_FLAT_NO_SMALLB] != 1u && flat_smallb_supported(d.dim, b, 2) && (uint64_t)n_rows * d.dim >= (1ull << 22);
    return shape && d.dim >= 256 && k <= 511 && (big || small_stream) && !ix->opt[HVX_OPT_FLAT_FORCE_VALU];
}

// scan `n_rows` rows (all rows, or d_subset internal ids) for b device-resident queries
int hvx::flat_scan_device(hvx_index *ix, const float *d_queries, uint32_t b, uint32_t k, const uint32_t *d_subset,
                          uint32_t n_rows, uint64_t *d_ids, float *d_scores, uint32_t *d_counts,
                          uint32_t *d_status, bool timed) {
    if (k == 0) return fail(HVX_ERR_K_RANGE, "result count must be non-zero");
    if (k > 1024) return fail(HVX_ERR_UNSUPPORTED, "flat scan supports k <= 1024");
    if (!d_subset) ix->sync_rewrites(); // (a restricted row list was resolved against the caller's generation: the callers sync before they resolve)
    if (!d_subset && ix->n_dead) { // "all rows" of an image with deleted nodes = its live rows (a deleted node has no item row: mutation.rs:1708-1745)
        int rc = ix->ensure_live();
        if (rc) return rc;
        d_subset = ix->f_live;
        n_rows = ix->live_rows_n;
    }
    if (ix->dev.dtype != HVX_F32) // bf16 / fp8 rows: the matrix-core pipeline, over all rows or over the restricted row list
        return flat_mfma_device(ix, d_queries, b, k, d_subset, n_rows, d_ids, d_scores, d_counts, d_status, timed);
    // f32 rows: candidates on the matrix cores where that pays (flat_scan_on_matrix_cores), exact re-rank / tail; any query whose
    // certificate is not reached sends the batch to the exact VALU scan
    {
        const DevIndex &d = ix->dev;
        if (flat_scan_on_matrix_cores(ix, b, k, n_rows)) {
            const int rc = flat_mfma_device(ix, d_queries, b, k, d_subset, n_rows, d_ids, d_scores, d_counts, d_status, timed);
            if (rc != -1) return rc;
            // certificate not reached for some queries: those -- and only those, unless they are many -- are answered
            // by the exact VALU scan below; the rest of the batch keeps its certified rows
            const std::vector<uint32_t> failed = ix->m_failed;
            const uint32_t nf = (uint32_t)failed.size();
            ix->last_scan_path |= nf * 4u <= b ? HVX_PATH_VALU_FALLBACK_QUERIES : HVX_PATH_VALU;
            if (tuning_env("HVX_FLAT_DEBUG"))
                fprintf(stderr, "[hvx flat] certificate not reached for %u of %u queries: exact VALU scan for %s\n", nf, b,
                        nf * 4u <= b ? "those queries only" : "the whole batch");
            if (nf * 4u <= b) {
                const size_t dim = d.dim;
                int r2;
                if (nf > ix->cap_fb) {
                    if ((r2 = ix->regrow((void **)&ix->fb_idx, (size_t)nf * 4))) return r2;
                    if ((r2 = ix->regrow((void **)&ix->fb_q, (size_t)nf * dim * 4))) return r2;
                    if ((r2 = ix->regrow((void **)&ix->fb_ids, (size_t)nf * 1024 * 8))) return r2;
                    if ((r2 = ix->regrow((void **)&ix->fb_sc, (size_t)nf * 1024 * 4))) return r2;
                    if ((r2 = ix->regrow((void **)&ix->fb_cnt, (size_t)nf * 4))) return r2;
                    if ((r2 = ix->regrow((void **)&ix->fb_st, (size_t)nf * 4))) return r2;
                    ix->cap_fb = nf;
                }
                HIP_TRY(hipMemcpyAsync(ix->fb_idx, failed.data(), (size_t)nf * 4, hipMemcpyHostToDevice, ix->stream));
                HIP_TRY(hipStreamSynchronize(ix->stream)); // `failed` lives on this stack frame
                HIP_TRY(launch_move_rows(reinterpret_cast<const uint32_t *>(d_queries), ix->fb_idx, reinterpret_cast<uint32_t *>(ix->fb_q),
                                         (uint32_t)dim, nf, true, ix->stream));
                if ((r2 = flat_scan_valu(ix, ix->fb_q, nf, k, d_subset, n_rows, ix->fb_ids, ix->fb_sc, ix->fb_cnt, ix->fb_st, timed, false))) return r2;
                HIP_TRY(launch_move_rows(reinterpret_cast<const uint32_t *>(ix->fb_ids), ix->fb_idx, reinterpret_cast<uint32_t *>(d_ids), 2 * k, nf, false, ix->stream));
                HIP_TRY(launch_move_rows(reinterpret_cast<const uint32_t *>(ix->fb_sc), ix->fb_idx, reinterpret_cast<uint32_t *>(d_scores), k, nf, false, ix->stream));
                HIP_TRY(launch_move_rows(ix->fb_cnt, ix->fb_idx, d_counts, 1, nf, false, ix->stream));
                if (d_status) HIP_TRY(launch_move_rows(ix->fb_st, ix->fb_idx, d_status, 1, nf, false, ix->stream));
                if (timed) HIP_TRY(hipEventRecord(ix->ev1, ix->stream));
                return HVX_OK;
            }
        }
    }
    return flat_scan_valu(ix, d_queries, b, k, d_subset, n_rows, d_ids, d_scores, d_counts, d_status, timed, true);
}

// the exact VALU scan (flat_distance_kernel + flat_select_kernel + finish); record_begin=false keeps an earlier ev0
int hvx::flat_scan_valu(hvx_index *ix, const float *d_queries, uint32_t b, uint32_t k, const uint32_t *d_subset, uint32_t n_rows,
                        uint64_t *d_ids, float *d_scores, uint32_t *d_counts, uint32_t *d_status, bool timed, bool record_begin) {
    if (record_begin) ix->last_scan_path = HVX_PATH_VALU;
    // chunk the scan so the distance workspace stays <= 256 MiB
    uint32_t chunk = 65536;
    while ((size_t)chunk * b * 4 > (256u << 20) && chunk > 256) chunk >>= 1;
    if (chunk > n_rows) chunk = std::max<uint32_t>(n_rows, 1);
    chunk = (chunk + 3) & ~3u;
    int rc = ix->flat_scratch(b, k, chunk);
    if (rc) return rc;
    HIP_TRY(launch_validate_queries(ix->dev, d_queries, b, ix->limit, ix->d_qstatus, ix->d_qhdr, ix->stream));
    if (timed && record_begin) HIP_TRY(hipEventRecord(ix->ev0, ix->stream));
    HIP_TRY(hipMemsetAsync(ix->f_top_c, 0, (size_t)b * 4, ix->stream));
    FlatArgs a;
    a.ix = ix->dev;
    a.queries = d_queries;
    a.qstatus = ix->d_qstatus;
    a.qhdr = ix->d_qhdr;
    a.subset = d_subset;
    a.n_rows = n_rows;
    a.dist = ix->f_dist;
    a.chunk_ld = chunk;
    a.b = b;
    a.k = k;
    a.top_scores = ix->f_top_s;
    a.top_ids = ix->f_top_i;
    a.top_counts = ix->f_top_c;
    for (uint32_t r0 = 0; r0 < n_rows; r0 += chunk) {
        a.row0 = r0;
        a.rows = std::min(chunk, n_rows - r0);
        HIP_TRY(launch_flat_distances(a, ix->stream));
        HIP_TRY(launch_flat_select(a, ix->stream));
    }
    a.row0 = 0;
    a.rows = 0;
    HIP_TRY(launch_flat_finish(a, d_ids, d_scores, d_counts, d_status, ix->stream));
    if (timed) HIP_TRY(hipEventRecord(ix->ev1, ix->stream));
    return HVX_OK;
}

static int flat_stats(hvx_index *ix, uint32_t b, uint64_t rows, hvx_stats *stats) {
    HIP_TRY(hipStreamSynchronize(ix->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ix->ev0, ix->ev1));
    stats->queries += b;
    stats->vectors_loaded += (uint64_t)b * rows;
    stats->distance_computations += (uint64_t)b * rows;
    stats->device_ms += ms;
    return HVX_OK;
}

extern "C" int hvx_flat_search_batch_device(const hvx_index *cix, const float *d_queries, uint32_t b, uint32_t k,
                                            uint64_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts,
                                            uint32_t *d_out_status, hvx_stats *stats) {
    if (!cix) return fail(HVX_ERR_INVARIANT, "null index");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    if (b == 0) return HVX_OK;
    if (b > ix->max_batch) return fail(HVX_ERR_UNSUPPORTED, "batch %u exceeds max_batch %u", b, ix->max_batch);
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    int rc = flat_scan_device(ix, d_queries, b, k, nullptr, ix->dev.n, d_out_ids, d_out_scores, d_out_counts, d_out_status, stats != nullptr);
    if (rc) return rc;
    if (stats) return flat_stats(ix, b, ix->live_rows(), stats);
    return HVX_OK;
}

int hvx::flat_scan_host(hvx_index *ix, const float *queries, uint32_t b, uint32_t k, const uint32_t *d_subset,
                        uint32_t n_rows, uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                        uint32_t *out_status, hvx_stats *stats) {
    int rc;
    const uint32_t mb = ix->max_batch;
    std::vector<uint32_t> status(b, 0);
    for (uint32_t c0 = 0; c0 < b; c0 += mb) {
        const uint32_t cb = std::min(mb, b - c0);
        if ((rc = ix->stage(cb, k))) return rc;
        if ((rc = ix->stage_in(queries + (size_t)c0 * ix->dev.dim, cb))) return rc;
        rc = flat_scan_device(ix, ix->s_queries, cb, k, d_subset, n_rows, ix->s_ids, ix->s_scores, ix->s_counts, ix->s_status, stats != nullptr);
        if (rc) return rc;
        if ((rc = ix->stage_out(cb, k))) return rc;
        if (stats) {
            if ((rc = flat_stats(ix, cb, (!d_subset && ix->n_dead) ? ix->live_rows() : n_rows, stats))) return rc;
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

extern "C" int hvx_flat_search_batch(const hvx_index *cix, const float *queries, uint32_t b, uint32_t k,
                                     uint64_t *out_ids, float *out_scores, uint32_t *out_counts,
                                     uint32_t *out_status, hvx_stats *stats) {
    if (!cix) return fail(HVX_ERR_INVARIANT, "null index");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    if (k == 0) return fail(HVX_ERR_K_RANGE, "result count must be non-zero");
    if (b == 0) return HVX_OK;
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    return flat_scan_host(ix, queries, b, k, nullptr, ix->dev.n, out_ids, out_scores, out_counts, out_status, stats);
}

extern "C" int hvx_merge_topk_device(const hvx_index *cix, uint32_t g, uint32_t b, uint32_t k, const uint64_t *d_ids,
                                     const float *d_scores, const uint32_t *d_counts, uint64_t *d_out_ids,
                                     float *d_out_scores, uint32_t *d_out_counts) {
    if (!cix) return fail(HVX_ERR_INVARIANT, "null index");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    if (k == 0 || g == 0) return fail(HVX_ERR_K_RANGE, "k and shard count must be non-zero");
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(launch_merge_topk(g, b, k, d_ids, d_scores, d_counts, d_out_ids, d_out_scores, d_out_counts, ix->stream));
    return HVX_OK;
}

// One rank's payload of the packed exchange buffer: ids [b][k] u64, then scores [b][k] f32, then counts [b] u32,
// padded to a multiple of 8 bytes -- so that ONE all-gather carries everything a merge needs.
extern "C" size_t hvx_topk_payload_bytes(uint32_t b, uint32_t k) {
    const size_t raw = (size_t)b * k * 12 + (size_t)b * 4;
    return (raw + 7) & ~(size_t)7;
}

extern "C" int hvx_merge_topk_packed_device(const hvx_index *cix, uint32_t g, uint32_t b, uint32_t k, const void *d_packed,
                                            uint64_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts) {
    if (!cix || !d_packed) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_index *ix = const_cast<hvx_index *>(cix);
    if (k == 0 || g == 0) return fail(HVX_ERR_K_RANGE, "k and shard count must be non-zero");
    const size_t payload = hvx_topk_payload_bytes(b, k);
    const char *base = static_cast<const char *>(d_packed);
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    HIP_TRY(launch_merge_topk_strided(g, b, k, reinterpret_cast<const uint64_t *>(base),
                                      reinterpret_cast<const float *>(base + (size_t)b * k * 8),
                                      reinterpret_cast<const uint32_t *>(base + (size_t)b * k * 12), payload / 8, payload / 4, payload / 4,
                                      d_out_ids, d_out_scores, d_out_counts, ix->stream));
    return HVX_OK;
}
