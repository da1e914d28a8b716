// mfma_rate_probe.hip -- issue-rate probe for the three matrix instructions the fp8 exact scan could use on gfx950:
//   v_mfma_f32_32x32x16_bf16                 (what hvx_flat_tile.hip issues after widening e4m3 codes to bf16 in registers)
//   v_mfma_f32_32x32x16_fp8_fp8              (non-scaled fp8: same K per instruction)
//   v_mfma_scale_f32_32x32x64_f8f6f4 (fp8)   (MX-scaled, K = 64 per instruction)
// 4 workgroups of 4 wavefronts per CU (4 per SIMD) on every CU, 4 independent accumulator tiles per wavefront (64 registers),
// operands in registers, no memory traffic.  Prints TFLOP/s of each.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate_probe mfma_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const int lane = threadIdx.x;
    bf16x8 a16, b16;
    for (int e = 0; e < 8; ++e) { a16[e] = (__bf16)(float)((lane + e) & 3); b16[e] = (__bf16)(float)((lane * 3 + e) & 3); }
    long a8 = 0x3838383838383838L + lane, b8 = 0x3030303030303030L + lane; // e4m3 codes
    i32x8 a32, b32;
    for (int e = 0; e < 8; ++e) { a32[e] = 0x38383838 + lane + e; b32[e] = 0x30303030 + lane * 3 + e; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16, b16, acc[t], 0, 0, 0);
            if (KIND == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a8, b8, acc[t], 0, 0, 0);
            if (KIND == 2) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a32, b32, acc[t], 0 /* A = fp8 e4m3 */, 0 /* B = fp8 e4m3 */, 0, 127, 0, 127);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(const char *name, double flop_per_mfma) {
    const int blocks = 256 * 4, iters = 20000; // 4 workgroups of 4 waves per CU: 4 waves per SIMD keep the pipe fed
    float *out;
    (void)hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 /* waves */ * iters * 4 /* tiles */ * flop_per_mfma;
    printf("{\"instruction\": \"%s\", \"ms\": %.3f, \"tflops\": %.1f}\n", name, ms, flops / ms / 1e9);
    (void)hipFree(out);
}

int main() {
    run<0>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16);
    run<1>("v_mfma_f32_32x32x16_fp8_fp8", 2.0 * 32 * 32 * 16);
    run<2>("v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 x fp8)", 2.0 * 32 * 32 * 64);
    return 0;
}
