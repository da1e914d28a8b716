// hvx_simhash.hip -- SimHash projections on the device.
//
// Reference: crates/db/src/search/vector/unaligned_vector/simhash.rs:123-178 (64 unit hyperplanes from a
// seeded StdRng), :263-291 (hash_from_slice: bit p = [sequential, unfused dot(v, plane_p) > 0]),
// simhash.rs:44-59 (order code that keys the canonical vector rows).  The hyperplanes are generated on the
// host exactly as the reference does (rand 0.10 StdRng = ChaCha12 keyed through seed_from_u64's PCG32
// expansion; pinned by the reference's 0x6d91_a757_8862_6786 known answer, tests/test_gpu_parity.py).
// On the device ONE wavefront hashes one vector: lane p owns hyperplane p and walks the dimension in the
// reference's order (mul, then add -- the library is built with -ffp-contract=off), so the 64-bit SimHash
// is literally `__ballot(dot > 0)`.  Planes are stored transposed ([dim][64]) so a step is one coalesced
// 256-byte load; the vector component is a wave-uniform scalar load.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "hvx_host.h"

using namespace hvx;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) return fail(HVX_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

struct hvx_simhasher {
    int device = 0;
    uint32_t dim = 0;
    float *d_planes_t = nullptr; // [dim][64]
    float *d_stage = nullptr;    // staging of host vectors
    uint64_t *d_bits = nullptr;
    size_t cap_rows = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
};

namespace {

inline uint32_t rotl32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }

// rand_core SeedableRng::seed_from_u64: PCG32 XSH-RR words into the 32-byte ChaCha key
void seed_from_u64(uint64_t state, uint32_t key[8]) {
    const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    for (int i = 0; i < 8; ++i) {
        state = state * MUL + INC;
        const uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
        key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
}

void chacha12_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                       key[4], key[5], key[6], key[7], (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t x[16];
    memcpy(x, st, sizeof(x));
    auto qr = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
    };
    for (int r = 0; r < 6; ++r) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + st[i];
}

// One wavefront per row, lane p = hyperplane p: dot_p = sum_d v[d] * plane_p[d], mul then add, left to right (SimHasher::hash_from_slice,
// unaligned_vector/simhash.rs:263-291).  The chain of one lane is sequential by definition; what the kernel can hide is the memory under
// it.  Round 5: the loads of the next 32 elements (the row: wave-uniform scalar loads; the planes: 256 coalesced bytes per element, L2
// resident) are issued before the 32 dependent mul / add pairs of the current ones -- the first version took one L2 round trip per
// element (0.25 ms for a 1 024-query batch at dim 768, a quarter of the search that follows it on the non-strict arms).
constexpr int kHashUnroll = 32;
__global__ __launch_bounds__(64) void simhash_kernel(const float *__restrict__ planes_t, const float *__restrict__ vectors, uint32_t dim, uint32_t ld,
                                                     uint64_t n, uint64_t *__restrict__ out) {
    const uint64_t r = blockIdx.x;
    if (r >= n) return;
    const float *v = vectors + r * ld;
    const int lane = (int)threadIdx.x;
    const float *pl = planes_t + lane;
    float dot = 0.0f;
    uint32_t d = 0;
    if (dim >= (uint32_t)kHashUnroll) {
        float pv[kHashUnroll], vv[kHashUnroll];
#pragma unroll
        for (int u = 0; u < kHashUnroll; ++u) { pv[u] = pl[(size_t)u * 64]; vv[u] = v[u]; }
        for (; d + 2 * kHashUnroll <= dim; d += kHashUnroll) {
            float pn[kHashUnroll], vn[kHashUnroll];
#pragma unroll
            for (int u = 0; u < kHashUnroll; ++u) { pn[u] = pl[(size_t)(d + kHashUnroll + u) * 64]; vn[u] = v[d + kHashUnroll + u]; }
#pragma unroll
            for (int u = 0; u < kHashUnroll; ++u) { const float t = vv[u] * pv[u]; dot += t; } // unfused (-ffp-contract=off)
#pragma unroll
            for (int u = 0; u < kHashUnroll; ++u) { pv[u] = pn[u]; vv[u] = vn[u]; }
        }
#pragma unroll
        for (int u = 0; u < kHashUnroll; ++u) { const float t = vv[u] * pv[u]; dot += t; }
        d += kHashUnroll;
    }
    for (; d < dim; ++d) {
        const float t = v[d] * pl[(size_t)d * 64];
        dot += t;
    }
    const unsigned long long bits = __ballot(dot > 0.0f);
    if (lane == 0) out[r] = bits;
}

} // namespace

namespace {
// the same projection over bf16 rows in the interleaved device layout (hvx_device.h): element d sits at bf16_slot_of(d)
__global__ __launch_bounds__(64) void simhash_bf16_kernel(const float *planes_t, const uint16_t *rows, uint32_t dim, uint64_t n,
                                                          uint64_t *out) {
    const uint64_t r = blockIdx.x;
    if (r >= n) return;
    const uint16_t *v = rows + r * dim;
    const int lane = (int)threadIdx.x;
    float dot = 0.0f;
    for (uint32_t d = 0; d < dim; ++d) {
        const float t = bf16_to_f32(v[bf16_slot_of(d)]) * planes_t[(size_t)d * 64 + lane];
        dot += t;
    }
    const unsigned long long bits = __ballot(dot > 0.0f);
    if (lane == 0) out[r] = bits;
}
} // namespace

namespace hvx {
hipError_t launch_simhash_rows_bf16(const float *planes_t, const uint16_t *rows, uint32_t dim, uint64_t n, uint64_t *out,
                                    hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(simhash_bf16_kernel, dim3((uint32_t)n), dim3(64), 0, s, planes_t, rows, dim, n, out);
    return hipGetLastError();
}

hipError_t launch_simhash_rows(const float *planes_t, const float *rows, uint32_t dim, uint32_t ld, uint64_t n, uint64_t *out,
                               hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(simhash_kernel, dim3((uint32_t)n), dim3(64), 0, s, planes_t, rows, dim, ld, n, out);
    return hipGetLastError();
}

// 64 hyperplanes, plane-major draws; component = u*2-1 with u = (next_u32 >> 8) * 2^-24; normalised in f32
void simhash_planes_transposed(uint32_t dim, uint64_t seed, float *planes_t) {
    std::vector<float> planes((size_t)64 * dim);
    uint32_t key[8], blk[16];
    seed_from_u64(seed, key);
    uint64_t ctr = 0;
    size_t have = 16, total = (size_t)64 * dim;
    for (size_t i = 0; i < total; ++i) {
        if (have == 16) { chacha12_block(key, ctr++, blk); have = 0; }
        const float value = (float)(blk[have++] >> 8) * (1.0f / 16777216.0f);
        planes[i] = value * 2.0f - 1.0f;
    }
    for (uint32_t p = 0; p < 64; ++p) {
        float *pl = planes.data() + (size_t)p * dim;
        float s = 0.0f;
        for (uint32_t d = 0; d < dim; ++d) { const float t = pl[d] * pl[d]; s += t; }
        const float norm = std::sqrt(s);
        if (norm > 1e-10f)
            for (uint32_t d = 0; d < dim; ++d) pl[d] /= norm;
        for (uint32_t d = 0; d < dim; ++d) planes_t[(size_t)d * 64 + p] = pl[d];
    }
}
} // namespace hvx

extern "C" uint64_t hvx_order_code_from_simhash_bits(uint64_t bits) {
    const uint16_t b0 = (uint16_t)(bits >> 48), b1 = (uint16_t)(bits >> 32), b2 = (uint16_t)(bits >> 16), b3 = (uint16_t)bits;
    uint64_t code = 0;
    for (int bit = 15; bit >= 0; --bit) {
        code = (code << 1) | ((b0 >> bit) & 1u);
        code = (code << 1) | ((b1 >> bit) & 1u);
        code = (code << 1) | ((b2 >> bit) & 1u);
        code = (code << 1) | ((b3 >> bit) & 1u);
    }
    return code;
}

extern "C" void hvx_simhasher_free(hvx_simhasher *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->d_planes_t) (void)hipFree(h->d_planes_t);
    if (h->d_stage) (void)hipFree(h->d_stage);
    if (h->d_bits) (void)hipFree(h->d_bits);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int hvx_simhasher_new(uint32_t dim, uint64_t seed, int32_t device, hvx_simhasher **out) {
    if (!out) return fail(HVX_ERR_INVARIANT, "null argument");
    *out = nullptr;
    if (dim == 0) return fail(HVX_ERR_DIMENSION, "dimension must be non-zero");
    if ((uint64_t)dim * 64ull * 4ull > (1ull << 32)) return fail(HVX_ERR_DIMENSION, "SimHasher allocation too large");
    int dev = device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipSetDevice(dev));
    const size_t total = (size_t)64 * dim;
    std::vector<float> planes_t(total);
    simhash_planes_transposed(dim, seed, planes_t.data());
    hvx_simhasher *h = new hvx_simhasher();
    h->device = dev;
    h->dim = dim;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void **)&h->d_planes_t, total * 4) != hipSuccess ||
        hipMemcpy(h->d_planes_t, planes_t.data(), total * 4, hipMemcpyHostToDevice) != hipSuccess) {
        hvx_simhasher_free(h);
        return fail(HVX_ERR_DEVICE, "SimHasher upload failed");
    }
    *out = h;
    return HVX_OK;
}

extern "C" int hvx_simhash_batch(const hvx_simhasher *ch, const float *vectors, uint64_t n, uint64_t *out_bits) {
    if (!ch || (n && (!vectors || !out_bits))) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_simhasher *h = const_cast<hvx_simhasher *>(ch);
    if (n == 0) return HVX_OK;
    std::lock_guard<std::mutex> lock(h->mu);
    HIP_TRY(hipSetDevice(h->device));
    const uint64_t chunk_rows = 1u << 20;
    for (uint64_t r0 = 0; r0 < n; r0 += chunk_rows) {
        const uint64_t rows = std::min(chunk_rows, n - r0);
        if (rows > h->cap_rows) {
            if (h->d_stage) (void)hipFree(h->d_stage);
            if (h->d_bits) (void)hipFree(h->d_bits);
            h->d_stage = nullptr; h->d_bits = nullptr; h->cap_rows = 0;
            HIP_TRY(hipMalloc((void **)&h->d_stage, rows * h->dim * 4));
            HIP_TRY(hipMalloc((void **)&h->d_bits, rows * 8));
            h->cap_rows = rows;
        }
        HIP_TRY(hipMemcpyAsync(h->d_stage, vectors + r0 * h->dim, rows * h->dim * 4, hipMemcpyDefault, h->stream));
        HIP_TRY(launch_simhash_rows(h->d_planes_t, h->d_stage, h->dim, h->dim, rows, h->d_bits, h->stream));
        HIP_TRY(hipMemcpyAsync(out_bits + r0, h->d_bits, rows * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return HVX_OK;
}
