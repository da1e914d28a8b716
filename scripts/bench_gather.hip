// bench_gather.hip -- microbenchmark (tuning aid, not product): what HBM bandwidth does the HNSW
// access pattern admit on MI355X?  Each wavefront repeatedly gathers 16 random 3 KiB rows (8 lanes per
// row, 24 x 16-byte loads per lane and row, all 48 loads in flight) from a 1M x 768 f32 matrix.
//   hipcc --offload-arch=gfx950 -O3 scripts/bench_gather.hip -o /tmp/bench_gather && /tmp/bench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int P>
__global__ __launch_bounds__(64) void gather_kernel(const float *vec, const uint32_t *idx, uint32_t rounds, uint32_t n_idx,
                                                    float *out, uint32_t real_rows, uint32_t spin, unsigned long long *lat) {
    const int lane = threadIdx.x & 63, grp = lane >> 3, j = lane & 7;
    const uint32_t wave = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    float4 acc = make_float4(0, 0, 0, 0);
    uint32_t cursor = (wave * 9973u) % n_idx;
    unsigned long long lat_acc = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        float4 x[P][24];
        const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t f = p * 8 + grp;
            const uint32_t node = idx[(cursor + (f < real_rows ? f : 0)) % n_idx]; // idle groups shadow row 0
            const float4 *rp = reinterpret_cast<const float4 *>(vec + (size_t)node * 768) + j;
#pragma unroll
            for (int k = 0; k < 24; ++k) x[p][k] = rp[k * 8];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int k = 0; k < 24; ++k) { acc.x += x[p][k].x; acc.y += x[p][k].y; acc.z += x[p][k].z; acc.w += x[p][k].w; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lat_acc += __builtin_readcyclecounter() - t0;
        cursor = (cursor + P * 8 + (uint32_t)(acc.x != 12345.f)) % n_idx; // dependent chain like the real search
        uint32_t my_spin = spin & 0xFFFFu;
        if (spin >> 16) { // desynchronised: uniform random in [0, 2*spin]
            uint32_t hsh = (wave * 7919u + r * 104729u) * 2654435761u;
            my_spin = (hsh >> 8) % (2u * my_spin + 1u);
        }
        for (uint32_t sp = 0; sp < my_spin; ++sp) __builtin_amdgcn_s_sleep(8); // ~512 cycles of "admission" per unit
    }
    if (lane == 0) lat[wave] = lat_acc;
    out[wave * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}

int main(int argc, char **argv) {
    const size_t n = 1000000, dim = 768;
    float *vec;
    CK(hipMalloc(&vec, n * dim * 4));
    CK(hipMemset(vec, 0, n * dim * 4));
    const uint32_t n_idx = 1 << 22;
    std::vector<uint32_t> h(n_idx);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s % n); }
    std::vector<uint32_t> hs(n_idx);
    for (uint32_t i = 0; i < n_idx; ++i) hs[i] = i % n; // sequential rows for comparison
    uint32_t *idx, *idx_seq;
    CK(hipMalloc(&idx, n_idx * 4));
    CK(hipMalloc(&idx_seq, n_idx * 4));
    CK(hipMemcpy(idx, h.data(), n_idx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(idx_seq, hs.data(), n_idx * 4, hipMemcpyHostToDevice));
    float *out;
    CK(hipMalloc(&out, 16384 * 64 * 4));
    unsigned long long *lat;
    CK(hipMalloc(&lat, 16384 * 8));
    std::vector<unsigned long long> hlat(16384);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t rounds = 100;
    auto run = [&](const char *name, int P, uint32_t waves, uint32_t wpb, const uint32_t *ix, uint32_t real_rows = 16, uint32_t spin = 0) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            dim3 grid(waves / wpb), block(64 * wpb);
            if (P == 2) hipLaunchKernelGGL(gather_kernel<2>, grid, block, 0, 0, vec, ix, rounds, n_idx, out, real_rows, spin, lat);
            else hipLaunchKernelGGL(gather_kernel<1>, grid, block, 0, 0, vec, ix, rounds, n_idx, out, real_rows, spin, lat);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const uint32_t rr = real_rows < (uint32_t)P * 8 ? real_rows : (uint32_t)P * 8;
        const double bytes = (double)waves * rounds * rr * 3072;
        CK(hipMemcpy(hlat.data(), lat, waves * 8, hipMemcpyDeviceToHost));
        double la = 0;
        for (uint32_t i = 0; i < waves; ++i) la += (double)hlat[i];
        la /= (double)waves * rounds;
        printf("%-24s waves %5u P=%d real %2u spin %2u: %.3f ms  %5.0f GB/s  %.2f us/round  gather latency %.0f cycles (clock ~%.2f GHz if s_memtime=shader clk)\n",
               name, waves, P, rr, spin & 0xFFFFu, ms, bytes / ms / 1e6, ms * 1e3 / rounds, la, 0.0);
    };
    for (uint32_t waves : {256u, 512u, 1024u, 2048u}) {
        run("random rows", 2, waves, 1, idx);
        run("random rows", 1, waves, 1, idx);
    }
    for (uint32_t spin : {0u, 4u, 8u, 12u, 16u, 24u})
        run("random, 10 of 16 real", 2, 1024, 1, idx, 10, spin);
    for (uint32_t spin : {4u, 8u, 12u, 16u, 24u, 32u})
        run("10/16 real, desync", 2, 1024, 1, idx, 10, spin | (1u << 16));
    for (uint32_t spin : {8u, 16u, 24u})
        run("10/16 real, desync P=1x2", 1, 1024, 1, idx, 8, spin | (1u << 16));
    run("sequential rows", 2, 1024, 1, idx_seq);
    return 0;
}
