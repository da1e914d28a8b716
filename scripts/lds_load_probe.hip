#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float *p, float *out) {
    extern __shared__ char smem[];
    float *dump = reinterpret_cast<float *>(smem + 1024);
    const float *addr = p + threadIdx.x * 32;
    uint32_t lds_off = (uint32_t)(uintptr_t)dump; // LDS byte address
    uint32_t m0save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0save) : "s"(lds_off), "v"(addr) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = dump[threadIdx.x];
}
int main() {
    std::vector<float> h(64 * 32);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 64 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o);
    std::vector<float> r(64);
    hipMemcpy(r.data(), o, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += r[i] != (float)(i * 32);
    printf("lds-prefetch-test bad=%d r[0]=%g r[1]=%g r[63]=%g\n", bad, r[0], r[1], r[63]);
    return bad != 0;
}
