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

// fp8 e4m3fn rows with one f32 scale per row (BASELINE config #5): code = RNE(x / scale), scale = max|x| / 448.
// The staging row is overwritten with the dequantised values fl32(scale * decode(code)) -- these ARE the index:
// validation, cosine headers and every distance see exactly them.  Rows holding a non-finite value are left
// untouched so that validation reports them.
__global__ __launch_bounds__(64) void quantize_fp8_kernel(float *staging, uint8_t *dst, float *rowscale, uint32_t n, uint32_t dim) {
    const uint32_t r = blockIdx.x;
    if (r >= n) return;
    float *row = staging + (size_t)r * dim;
    float amax = 0.f;
    bool finite = true;
    for (uint32_t i = threadIdx.x; i < dim; i += 64) {
        const float v = row[i];
        if (!f32_is_finite(v)) finite = false;
        amax = fmaxf(amax, fabsf(v));
    }
    for (int s = 32; s > 0; s >>= 1) amax = fmaxf(amax, __shfl_xor(amax, s, 64));
    if (__ballot(!finite)) {
        if (threadIdx.x == 0) rowscale[r] = 1.0f;
        return;
    }
    const float scale = amax > 0.f ? __fdiv_rn(amax, 448.0f) : 1.0f;
    if (threadIdx.x == 0) rowscale[r] = scale;
    for (uint32_t i = threadIdx.x; i < dim; i += 64) {
        const uint8_t c = fp8_e4m3_encode(__fdiv_rn(row[i], scale));
        dst[(size_t)r * dim + fp8_slot_of(i)] = c;
        row[i] = scale * fp8_e4m3_decode(c);
    }
}

__global__ __launch_bounds__(64) void f32_row_norm2_kernel(const float *rows, uint32_t n, uint32_t ld, uint32_t dim, float *out) {
    const uint32_t r = blockIdx.x;
    if (r >= n) return;
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < dim; i += 64) {
        const double v = (double)rows[(size_t)r * ld + i];
        acc += v * v;
    }
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
    if (threadIdx.x == 0) out[r] = (float)acc;
}

// the f32 values the index holds for rows [row0, row0 + n): what every distance is computed on (f32 rows: the rows; bf16:
// the rounded values; fp8: fl32(scale * decode(code)), the expression the quantiser stored).  out [n][dim], plain order.
__global__ __launch_bounds__(256) void read_rows_kernel(DevIndex ix, uint64_t row0, uint64_t n, float *out) {
    const size_t total = (size_t)n * ix.dim;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const size_t r = row0 + t / ix.dim;
        const uint32_t i = (uint32_t)(t % ix.dim);
        float v;
        if (ix.dtype == 2u) v = ix.rowscale[r] * fp8_e4m3_decode(ix.vec8[r * ix.dim + fp8_slot_of(i)]);
        else if (ix.dtype == 1u) v = bf16_to_f32(ix.vecb[r * ix.dim + bf16_slot_of(i)]);
        else v = ix.vec[r * ix.ld + i];
        out[t] = v;
    }
}

hipError_t launch_read_rows(const DevIndex &ix, uint64_t row0, uint64_t n, float *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(read_rows_kernel, dim3(4096), dim3(256), 0, s, ix, row0, n, out);
    return hipGetLastError();
}

hipError_t launch_quantize_fp8(float *staging, uint8_t *dst, float *rowscale, uint32_t n, uint32_t dim, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(quantize_fp8_kernel, dim3(n), dim3(64), 0, s, staging, dst, rowscale, n, dim);
    return hipGetLastError();
}

hipError_t launch_f32_row_norm2(const float *rows, uint32_t n, uint32_t ld, uint32_t dim, float *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(f32_row_norm2_kernel, dim3(n), dim3(64), 0, s, rows, n, ld, dim, out);
    return hipGetLastError();
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
