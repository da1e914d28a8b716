// mx_probe.hip -- test harness: operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands, as the MX build of the
// exact-scan tile kernel (hvx_flat_tile.hip, flat_tile2mx_kernel) relies on it: lane l holds row (l & 31), the 32 consecutive k of
// block (l >> 5), byte i of the eight registers = k 32 (l >> 5) + i; C as the 32 x 32 bf16 form (test 0); a scale (E8M0, 127 = 1.0)
// that is the same in both lanes of a row scales that row (test 1).  Tests 2 and 3 are informational: scales that differ between
// the two blocks of a row do NOT follow "lane = (row, block)" (measured: 1 020 of 1 024 entries off) -- the kernel does not use them.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void mx_kernel(const unsigned char *A, const unsigned char *B, const int *sa, const int *sb, float *C) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    i32x8 a, b;
    for (int v = 0; v < 8; ++v) {
        a[v] = *reinterpret_cast<const int *>(A + r * 64 + h * 32 + v * 4);
        b[v] = *reinterpret_cast<const int *>(B + r * 64 + h * 32 + v * 4);
    }
    f32x16 c;
    for (int e = 0; e < 16; ++e) c[e] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[lane], 0, sb[lane]);
    for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = c[e]; // C[m][n]
}

static unsigned char enc(int v) { // small integers as OCP e4m3 (exact for |v| <= 16): sign, 4 exponent bits (bias 7), 3 mantissa bits
    if (v == 0) return 0;
    const unsigned char s = v < 0 ? 0x80 : 0;
    int m = std::abs(v), e = 0;
    while ((1 << (e + 1)) <= m) ++e;
    const int frac = ((m << 3) >> e) & 7; // exact while m < 16 * ...
    return (unsigned char)(s | ((e + 7) << 3) | frac);
}

int main() {
    std::vector<unsigned char> A(32 * 64), B(32 * 64);
    std::vector<int> av(32 * 64), bv(32 * 64), sa(64), sb(64);
    unsigned x = 12345;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (int)((x >> 16) % 17) - 8; };
    for (int i = 0; i < 32 * 64; ++i) { av[i] = rnd(); bv[i] = rnd(); A[i] = enc(av[i]); B[i] = enc(bv[i]); }
    unsigned char *dA, *dB; int *dsa, *dsb; float *dC;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    int failed = 0;
    // test 0: all scales 1.0 (operand layout alone); 1: A scaled per row (same in both blocks); 2: A scaled per (row, block);
    // 3: B scaled per (row, block) as well
    for (int test = 0; test < 4; ++test) {
        for (int l = 0; l < 64; ++l) {
            sa[l] = test == 0 ? 127 : (test == 1 ? 127 + ((l & 31) % 3) : 127 + ((l & 31) % 3) + 2 * (l >> 5));
            sb[l] = test == 3 ? 127 - ((l & 31) % 2) - (l >> 5) : 127;
        }
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
        std::vector<float> C(32 * 32);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                double want = 0;
                for (int k = 0; k < 64; ++k) {
                    const int blk = k >> 5;
                    want += std::ldexp((double)av[m * 64 + k], sa[m + 32 * blk] - 127) * std::ldexp((double)bv[n * 64 + k], sb[n + 32 * blk] - 127);
                }
                if (std::fabs(C[m * 32 + n] - want) > 1e-3 * (1 + std::fabs(want))) { if (bad++ < 3) printf("  test %d: C[%d][%d] = %g, want %g\n", test, m, n, C[m * 32 + n], want); }
            }
        printf("mx probe test %d: %d of 1024 entries off\n", test, bad);
        if (test < 2) failed |= bad != 0;
    }
    return failed;
}
