// hvx_probe.hip -- device capability probe behind hvx_device_stream_read_gbs: the measured streaming-read rate of the box the
// library runs on, the denominator SURVEY.md 8(d) asks the HBM-bound kernels to be quoted against next to the 8 TB/s spec.
//
// The kernel is the upper bound of what the HNSW row gathers can reach: every wavefront issues 16-byte loads (global_load_dwordx4),
// 1 KiB per wavefront and load instruction, eight independent loads in flight per lane, nothing is written.  The loaded words are
// folded into one value per lane that is stored only if it equals a value the host never passes -- the loads cannot be removed.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "hvx_host.h"

using namespace hvx;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(HVX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

namespace {
constexpr int kUnroll = 8;
constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void stream_read_kernel(const uint4 *__restrict__ src, size_t n16, uint32_t magic, uint32_t *sink) {
    // grid-stride over tiles of kThreads * kUnroll 16-byte words; consecutive lanes read consecutive words (coalesced 1 KiB per wave)
    const size_t tile = (size_t)kThreads * kUnroll;
    uint32_t acc = 0;
    for (size_t t0 = (size_t)blockIdx.x * tile; t0 + tile <= n16; t0 += (size_t)gridDim.x * tile) {
        uint4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = src[t0 + (size_t)u * kThreads + threadIdx.x];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == magic) sink[0] = acc; // never true for the all-zero buffer and magic != 0
}
} // namespace

extern "C" int hvx_device_stream_read_gbs(int32_t device, uint64_t bytes, uint32_t iters, float *out_best_gbs, float *out_mean_gbs) {
    if (!out_best_gbs && !out_mean_gbs) return fail(HVX_ERR_INVARIANT, "null argument");
    if (iters == 0) iters = 5;
    const size_t tile_bytes = (size_t)kThreads * kUnroll * 16;
    bytes = bytes / tile_bytes * tile_bytes;
    if (bytes < tile_bytes) return fail(HVX_ERR_K_RANGE, "buffer of the stream-read probe is below one tile (%zu bytes)", tile_bytes);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    void *buf = nullptr;
    uint32_t *sink = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = HVX_OK;
    auto done = [&](int code) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (s) (void)hipStreamDestroy(s);
        if (buf) (void)hipFree(buf);
        if (sink) (void)hipFree(sink);
        return code;
    };
    if (hipMalloc(&buf, bytes) != hipSuccess) return done(fail(HVX_ERR_DEVICE, "hipMalloc of the %llu-byte probe buffer failed", (unsigned long long)bytes));
    if (hipMalloc((void **)&sink, 64) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        return done(fail(HVX_ERR_DEVICE, "stream-read probe: resource allocation failed"));
    if (hipMemsetAsync(buf, 0, bytes, s) != hipSuccess) return done(fail(HVX_ERR_DEVICE, "stream-read probe: memset failed"));
    const size_t n16 = bytes / 16;
    // enough workgroups to keep every CU's wave slots busy (8 workgroups of 4 wavefronts per CU), each walking its stride of tiles
    const uint32_t grid = (uint32_t)std::min<size_t>((size_t)prop.multiProcessorCount * 8, n16 / ((size_t)kThreads * kUnroll));
    float best = 0.f, sum = 0.f;
    for (uint32_t it = 0; it < iters + 1; ++it) { // first launch untimed (code object load, TLB warm-up)
        if (hipEventRecord(e0, s) != hipSuccess) { rc = fail(HVX_ERR_DEVICE, "hipEventRecord failed"); break; }
        hipLaunchKernelGGL(stream_read_kernel, dim3(grid), dim3(kThreads), 0, s, (const uint4 *)buf, n16, 0xA5A5A5A5u, sink);
        if (hipGetLastError() != hipSuccess || hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) {
            rc = fail(HVX_ERR_DEVICE, "stream-read probe launch failed");
            break;
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) { rc = fail(HVX_ERR_DEVICE, "stream-read probe timing failed"); break; }
        if (it == 0) continue;
        const float gbs = (float)((double)bytes / ((double)ms * 1e-3) / 1e9);
        best = std::max(best, gbs);
        sum += gbs;
    }
    if (rc == HVX_OK) {
        if (out_best_gbs) *out_best_gbs = best;
        if (out_mean_gbs) *out_mean_gbs = sum / (float)iters;
    }
    return done(rc);
}
