// smallq_probe.hip -- test harness (not part of the library): runs the two builds of the small-batch streaming kernel
// (hvx_flat_smallb.hip) on random rows and prints how their raw dot products compare with a double-precision dot product of
// the bf16-rounded operands.  usage: smallq_probe <kind 0|2> <dim> <rows> <b> <full 0|1> <subset 0|1> [timing iterations]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../helix-db_amd/csrc/hvx_flat_smallb.hip"

using namespace hvx;

static uint16_t to_bf16(float v) { return f32_to_bf16_rne(v); }
static float from_bf16(uint16_t h) { return bf16_to_f32(h); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

int main(int argc, char **argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 0;
    const uint32_t dim = argc > 2 ? atoi(argv[2]) : 768, n = argc > 3 ? atoi(argv[3]) : 1000, b = argc > 4 ? atoi(argv[4]) : 32;
    const bool full = argc > 5 && atoi(argv[5]) != 0, use_subset = argc > 6 && atoi(argv[6]) != 0;
    const uint32_t total = use_subset ? n * 2 + 7 : n;
    uint64_t lcg = 0x9E3779B97F4A7C15ull ^ ((uint64_t)dim << 32) ^ n;
    auto nd = [&]() { // sum of four uniforms, centred: cheap and bell-shaped enough
        float acc = 0.f;
        for (int t = 0; t < 4; ++t) { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; acc += (float)(lcg >> 40) * (1.0f / 16777216.0f); }
        return (acc - 2.0f) * 1.7320508f;
    };
    std::vector<float> rows_f((size_t)total * dim), q((size_t)32 * dim, 0.f);
    for (auto &v : rows_f) v = nd();
    for (size_t i = 0; i < (size_t)b * dim; ++i) q[i] = nd();
    std::vector<uint16_t> rows_b((size_t)total * dim), qhi((size_t)32 * dim), qlo((size_t)32 * dim);
    for (size_t i = 0; i < rows_f.size(); ++i) { rows_b[i] = to_bf16(rows_f[i]); if (kind == 0) rows_f[i] = from_bf16(rows_b[i]); }
    for (size_t i = 0; i < q.size(); ++i) { qhi[i] = to_bf16(q[i]); qlo[i] = to_bf16(q[i] - from_bf16(qhi[i])); }
    std::vector<uint32_t> subset(n);
    for (uint32_t i = 0; i < n; ++i) subset[i] = use_subset ? (uint32_t)(((uint64_t)i * 2654435761u) % total) : i;
    void *d_rows; uint16_t *d_qhi, *d_qlo; uint32_t *d_subset; float *d_dist;
    const size_t row_bytes = (size_t)total * dim * (kind == 2 ? 4 : 2);
    CK(hipMalloc(&d_rows, row_bytes)); CK(hipMalloc(&d_qhi, qhi.size() * 2)); CK(hipMalloc(&d_qlo, qlo.size() * 2));
    CK(hipMalloc(&d_subset, n * 4)); const uint32_t ld = (n + 3u) & ~3u; CK(hipMalloc(&d_dist, (size_t)32 * ld * 4));
    CK(hipMemcpy(d_rows, kind == 2 ? (const void *)rows_f.data() : (const void *)rows_b.data(), row_bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_qhi, qhi.data(), qhi.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_qlo, qlo.data(), qlo.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_subset, subset.data(), n * 4, hipMemcpyHostToDevice));
    MfmaArgs a{};
    a.qhi = d_qhi; a.qlo = d_qlo; a.subset = use_subset ? d_subset : nullptr; a.rows = d_rows; a.dim = dim; a.b = b; a.row0 = 0; a.nrows = n;
    a.dist = d_dist; a.chunk_ld = ld;
    std::vector<float> out[2];
    for (uint32_t build = 0; build < 2; ++build) {
        CK(hipMemset(d_dist, 0xFF, (size_t)32 * ld * 4));
        CK(launch_flat_smallb(a, kind, full, 256, build, 0));
        CK(hipDeviceSynchronize());
        out[build].resize((size_t)32 * ld);
        CK(hipMemcpy(out[build].data(), d_dist, out[build].size() * 4, hipMemcpyDeviceToHost));
    }
    if (argc > 7) { // timing: <iters> launches of each build, then of the selection kernel over their output
        const int iters = atoi(argv[7]);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (uint32_t build = 0; build < 2; ++build) {
            for (int it = 0; it < 3; ++it) CK(launch_flat_smallb(a, kind, full, 256, build, 0));
            CK(hipEventRecord(e0, 0));
            for (int it = 0; it < iters; ++it) CK(launch_flat_smallb(a, kind, full, 256, build, 0));
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters, bytes = (double)n * dim * (kind == 2 ? 4 : 2);
            printf("timing %s build (ring stages %d): %.1f us per launch, %.2f TB/s\n", build ? "register" : "ring", kSqStages, us, bytes / us / 1e6);
        }
        std::vector<float> rowterm(total, 1.0f), qn2(32, 1.0f);
        float *d_rt, *d_qn2, *d_slsc; uint32_t *d_slid;
        CK(hipMalloc(&d_rt, total * 4)); CK(hipMalloc(&d_qn2, 128)); CK(hipMalloc(&d_slsc, 32 * 8192 * 4)); CK(hipMalloc(&d_slid, 32 * 8192 * 4));
        CK(hipMemcpy(d_rt, rowterm.data(), total * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_qn2, qn2.data(), 128, hipMemcpyHostToDevice));
        a.rowterm = d_rt; a.qn2 = d_qn2; a.metric = kCosine;
        uint32_t slices = 0;
        for (int it = 0; it < 3; ++it) CK(launch_flat_select_radix(a, 64, nullptr, d_slsc, d_slid, 8192, &slices, 0));
        CK(hipEventRecord(e0, 0));
        for (int it = 0; it < iters; ++it) CK(launch_flat_select_radix(a, 64, nullptr, d_slsc, d_slid, 8192, &slices, 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("timing selection kernel (%u slices x 64 of %u scores per query): %.1f us per launch\n", slices, n, ms * 1e3 / iters);
        return 0;
    }
    // reference: double dot of what the build multiplies (one-pass: bf16 hi parts; full: the f32 values up to the dropped lo.lo term)
    double worst[2] = {0, 0}; uint32_t bad[2] = {0, 0}; int shown = 0;
    for (uint32_t qi = 0; qi < b; ++qi)
        for (uint32_t r = 0; r < n; ++r) {
            const size_t node = subset[r];
            double ref = 0, mag = 0;
            for (uint32_t d = 0; d < dim; ++d) {
                const float x = rows_f[node * dim + d];
                const double xv = full ? (double)x : (double)from_bf16(to_bf16(x));
                const double qv = full ? (double)q[(size_t)qi * dim + d] : (double)from_bf16(qhi[(size_t)qi * dim + d]);
                ref += xv * qv; mag += std::fabs(xv * qv);
            }
            for (int build = 0; build < 2; ++build) {
                const double got = out[build][(size_t)qi * ld + r], err = std::fabs(got - ref) / (mag + 1e-30);
                if (!(err < (full ? 3e-5 : 1e-5))) {
                    if (bad[build]++ < 1 || (shown < 3 && build == 0)) { printf("build %d: query %u row %u (block %u, row in block %u): got %.6f want %.6f\n", build, qi, r, r / 32, r % 32, got, ref); ++shown; }
                }
                if (err == err && err > worst[build]) worst[build] = err;
            }
        }
    if (bad[0]) { // which (query, row) cells of the first two row blocks are off: one line per query, one character per row
        for (uint32_t blk = 0; blk < 2 && blk * 32 < n; ++blk) {
            printf("ring build, row block %u (line = query, column = row in block; X = off)\n", blk);
            for (uint32_t qi = 0; qi < b; ++qi) {
                char line[33] = {0};
                for (uint32_t rr = 0; rr < 32 && blk * 32 + rr < n; ++rr) {
                    const uint32_t r = blk * 32 + rr; const size_t node = subset[r];
                    double ref = 0, mag = 0;
                    for (uint32_t d = 0; d < dim; ++d) {
                        const float x = rows_f[node * dim + d];
                        const double xv = full ? (double)x : (double)from_bf16(to_bf16(x));
                        const double qv = full ? (double)q[(size_t)qi * dim + d] : (double)from_bf16(qhi[(size_t)qi * dim + d]);
                        ref += xv * qv; mag += std::fabs(xv * qv);
                    }
                    line[rr] = std::fabs(out[0][(size_t)qi * ld + r] - ref) / (mag + 1e-30) < (full ? 3e-5 : 1e-5) ? '.' : 'X';
                }
                printf("  q%02u %s\n", qi, line);
            }
        }
    }
    printf("kind %d dim %u rows %u b %u full %d subset %d: ring build bad %u worst %.3g | register build bad %u worst %.3g\n", kind, dim, n, b, (int)full,
           (int)use_subset, bad[0], worst[0], bad[1], worst[1]);
    return bad[0] || bad[1] ? 1 : 0;
}
