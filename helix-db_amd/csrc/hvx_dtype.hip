// hvx_dtype.hip -- reduced-precision row storage.  The reference's only active codec is f32
// (crates/db/src/search/vector/distance/mod.rs:17-41); bf16 rows are new (BASELINE config #4): the
// index image is rounded once at import (round-to-nearest-even), every distance is then computed in
// f32 on the exactly-representable rounded values in the reference's summation order, so results are
// bit-identical to the reference CPU path run on the rounded vectors.
#include "hvx_device.h"
#include "hvx_kernels.h"

namespace hvx {

__global__ __launch_bounds__(256) void round_bf16_inplace_kernel(float *v, size_t count) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
        v[i] = bf16_to_f32(f32_to_bf16_rne(v[i]));
}

// staging [n][dim] f32 (already rounded) -> dst [n][dim] bf16 in the interleaved device layout
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float *staging, uint16_t *dst, uint32_t n, uint32_t dim) {
    const size_t total = (size_t)n * dim;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const uint32_t row = (uint32_t)(t / dim), i = (uint32_t)(t % dim);
        dst[(size_t)row * dim + bf16_slot_of(i)] = f32_to_bf16_rne(staging[t]);
    }
}

hipError_t launch_round_bf16_inplace(float *v, size_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(round_bf16_inplace_kernel, dim3(4096), dim3(256), 0, s, v, count);
    return hipGetLastError();
}

hipError_t launch_pack_bf16(const float *staging, uint16_t *dst, uint32_t n, uint32_t dim, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(4096), dim3(256), 0, s, staging, dst, n, dim);
    return hipGetLastError();
}

} // namespace hvx
