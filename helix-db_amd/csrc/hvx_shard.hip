// hvx_shard.hip -- the sharded search step through the C ABI (north star: "per-shard top-k merged via RCCL
// all-gather over xGMI"): one process per GPU holds one id-range shard (hvx_index) and calls
// hvx_shard_group_search_batch_device, which enqueues on the shard's stream
//     local search (writes ids / scores / counts straight into this rank's payload)
//  -> ncclAllGather of the packed payload (b x k x (u64 id, f32 score) + b counts; 124 KB per rank at b = 1024, k = 10)
//  -> merge of the `world` sorted lists by Candidate order (model.rs:55-61) on every rank.
// No host synchronisation; one collective per step; nothing else crosses the links.
//
// RCCL is bound at run time (dlopen of librccl.so.1, or the copy the host runtime already loaded): the library itself keeps
// no link-time dependency on it, single-GPU deployments never touch it.  The communicator is bootstrapped the NCCL way:
// rank 0 calls hvx_shard_group_unique_id, the host hands the 128 bytes to every rank by whatever transport it owns.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "hvx_host.h"

using namespace hvx;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(HVX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

namespace {
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl &rccl() {
    std::call_once(g_rccl_once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.handle) break;
        }
        if (!g_rccl.handle) return;
        g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.handle, "ncclGetUniqueId");
        g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.handle, "ncclCommInitRank");
        g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(g_rccl.handle, "ncclAllGather");
        g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.handle, "ncclCommDestroy");
        g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.handle, "ncclGetErrorString");
        g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllGather && g_rccl.CommDestroy && g_rccl.GetErrorString;
    });
    return g_rccl;
}
} // namespace

struct hvx_shard_group {
    hvx_index *ix = nullptr;
    ncclComm_t comm = nullptr;
    uint32_t rank = 0, world = 1, max_batch = 0, max_k = 0;
    char *send = nullptr, *recv = nullptr; // [payload(max)] / [world][payload(max)]
    uint32_t *status = nullptr;            // [max_batch]
};

extern "C" int hvx_shard_group_unique_id(uint8_t *out /*[128]*/) {
    if (!out) return fail(HVX_ERR_INVARIANT, "null argument");
    const Rccl &r = rccl();
    if (!r.ok) return fail(HVX_ERR_DEVICE, "RCCL is not available (librccl.so.1 could not be loaded)");
    ncclUniqueId id;
    const ncclResult_t e = r.GetUniqueId(&id);
    if (e != ncclSuccess) return fail(HVX_ERR_DEVICE, "ncclGetUniqueId: %s", r.GetErrorString(e));
    static_assert(sizeof(id) == HVX_SHARD_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(out, &id, sizeof(id));
    return HVX_OK;
}

extern "C" void hvx_shard_group_free(hvx_shard_group *g) {
    if (!g) return;
    if (g->ix) {
        (void)hipSetDevice(g->ix->device);
        (void)hipStreamSynchronize(g->ix->stream);
    }
    if (g->comm && rccl().ok) (void)rccl().CommDestroy(g->comm);
    if (g->send) (void)hipFree(g->send);
    if (g->recv) (void)hipFree(g->recv);
    if (g->status) (void)hipFree(g->status);
    delete g;
}

extern "C" int hvx_shard_group_init(hvx_index *local_shard, const uint8_t *unique_id, uint32_t rank, uint32_t world, uint32_t max_batch,
                                    uint32_t max_k, hvx_shard_group **out) {
    if (!local_shard || !out || (world > 1 && !unique_id)) return fail(HVX_ERR_INVARIANT, "null argument");
    *out = nullptr;
    if (world == 0 || rank >= world) return fail(HVX_ERR_K_RANGE, "rank %u outside a group of %u", rank, world);
    if (max_batch == 0 || max_k == 0) return fail(HVX_ERR_K_RANGE, "max_batch and max_k must be non-zero");
    if (max_batch > local_shard->max_batch) return fail(HVX_ERR_UNSUPPORTED, "max_batch %u exceeds the shard's %u", max_batch, local_shard->max_batch);
    HIP_TRY(hipSetDevice(local_shard->device));
    hvx_shard_group *g = new hvx_shard_group();
    g->ix = local_shard;
    g->rank = rank;
    g->world = world;
    g->max_batch = max_batch;
    g->max_k = max_k;
    const size_t payload = hvx_topk_payload_bytes(max_batch, max_k);
    if (hipMalloc((void **)&g->send, payload) != hipSuccess || hipMalloc((void **)&g->recv, payload * world) != hipSuccess ||
        hipMalloc((void **)&g->status, (size_t)max_batch * 4) != hipSuccess) {
        hvx_shard_group_free(g);
        return fail(HVX_ERR_DEVICE, "hipMalloc of the exchange buffers failed");
    }
    if (world > 1) {
        const Rccl &r = rccl();
        if (!r.ok) { hvx_shard_group_free(g); return fail(HVX_ERR_DEVICE, "RCCL is not available (librccl.so.1 could not be loaded)"); }
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof(id));
        const ncclResult_t e = r.CommInitRank(&g->comm, (int)world, id, (int)rank);
        if (e != ncclSuccess) {
            g->comm = nullptr;
            hvx_shard_group_free(g);
            return fail(HVX_ERR_DEVICE, "ncclCommInitRank(rank %u of %u): %s", rank, world, r.GetErrorString(e));
        }
    } else if (unique_id && rccl().ok) { // a group of one still exercises the collective when an id is given (tests)
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof(id));
        const ncclResult_t e = rccl().CommInitRank(&g->comm, 1, id, 0);
        if (e != ncclSuccess) g->comm = nullptr;
    }
    *out = g;
    return HVX_OK;
}

extern "C" int hvx_shard_group_search_batch_device(hvx_shard_group *g, const float *d_queries, uint32_t b, uint32_t k, uint32_t ef,
                                                   uint64_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts) {
    if (!g || !d_queries || !d_out_ids || !d_out_scores || !d_out_counts) return fail(HVX_ERR_INVARIANT, "null argument");
    if (b == 0) return HVX_OK;
    if (b > g->max_batch || k > g->max_k) return fail(HVX_ERR_UNSUPPORTED, "batch %u / k %u exceed the group's %u / %u", b, k, g->max_batch, g->max_k);
    int rc = check_k_ef(k, ef);
    if (rc) return rc;
    hvx_index *ix = g->ix;
    std::lock_guard<std::mutex> lock(ix->mu);
    HIP_TRY(hipSetDevice(ix->device));
    const size_t payload = hvx_topk_payload_bytes(b, k);
    uint64_t *s_ids = reinterpret_cast<uint64_t *>(g->send);
    float *s_sc = reinterpret_cast<float *>(g->send + (size_t)b * k * 8);
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(g->send + (size_t)b * k * 12);
    rc = enqueue_search(ix, d_queries, b, k, ef, s_ids, s_sc, s_cnt, g->status, nullptr, false);
    if (rc) return rc;
    const char *gathered = g->send;
    if (g->comm) {
        const ncclResult_t e = rccl().AllGather(g->send, g->recv, payload, ncclUint8, g->comm, ix->stream);
        if (e != ncclSuccess) return fail(HVX_ERR_DEVICE, "ncclAllGather: %s", rccl().GetErrorString(e));
        gathered = g->recv;
    } else if (g->world > 1) {
        return fail(HVX_ERR_INVARIANT, "shard group of %u ranks has no communicator", g->world);
    }
    HIP_TRY(launch_merge_topk_strided(g->comm ? g->world : 1u, b, k, reinterpret_cast<const uint64_t *>(gathered),
                                      reinterpret_cast<const float *>(gathered + (size_t)b * k * 8),
                                      reinterpret_cast<const uint32_t *>(gathered + (size_t)b * k * 12), payload / 8, payload / 4, payload / 4,
                                      d_out_ids, d_out_scores, d_out_counts, ix->stream));
    return HVX_OK;
}
