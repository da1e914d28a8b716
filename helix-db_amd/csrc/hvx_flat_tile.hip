// hvx_flat_tile.hip -- the large-tile contraction kernels of the matrix-core exact scan (BASELINE configs #4 / #5 and the
// f32 exact scan): candidate generation for restricted_exact_scan (crates/db/src/search/vector/restricted.rs:753-835), see
// the pipeline description at the top of hvx_flat_mfma.hip.  These kernels only FILTER: a score leaves the tile when it is
// below the query's running threshold, as a (score, row) pair; the first chunk of a scan (which has no threshold yet) and
// the repeat passes stay on the 128 x 128 kernel.
//
// Common structure (gfx950): every wavefront owns 128 queries x 64 rows = 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16
// (128 registers).  Operand stages are copied HBM/L2 -> LDS by global_load_lds_dwordx4 (no registers, no ds_write pass).
// The LDS image is lane-linear per wave instruction (1 KB = 8 tile rows x 128 B, or 16 x 64 B), so the XOR swizzle that
// makes the fragment reads (ds_read_b128, lane = tile row) conflict-free (SQ_LDS_BANK_CONFLICT = 0 measured) is applied to
// the SOURCE address and again to the read address: 128-B rows: 16-B slot ^ ((row >> 1) & 7); 64-B rows: slot ^ ((row >> 2) & 3).
// fp8 codes are widened to bf16 (exactly) in registers between the ds_read and the MFMA: v_cvt_pk_f32_fp8 + v_perm; one
// ds_read_b128 of the code tile feeds two MFMA steps (the query operand is stored in the matching order: tile_slot_fp8).
// Workgroups are mapped to tiles XCD-aware (tile_coords): the workgroups an XCD runs at a time form a super-tile that shares
// its operand tiles in that XCD's L2 (measured: L2 hit rate 0.91, HBM traffic = the rows once).
//
// Two builds:
//   flat_tile2_kernel    (default) 256 threads = 2 x 2 wavefronts on a 256 x 128 tile, a ring of three 32-deep stages
//                        (bf16, 72 KB) or two 64-deep stages (fp8, 80 KB), counted vmcnt + raw s_barrier, fragment reads as
//                        inline asm with stated lgkmcnt waits: TWO independent workgroups per CU.
//   flat_tile256_kernel  512 threads = 2 x 4 wavefronts on a 256 x 256 tile, two 64-deep LDS buffers, one workgroup per CU;
//                        carries the measurement switches (HVX_FLAT_TILE_ABLATE).
// What bounds them (profiles/history/r02g_*): the matrix cores are NOT the limit.  Ablation of the 512-thread build, 1024 x 1M x 768 bf16
// (ms per launch): MFMAs alone 0.35 (= the dense peak at the 2.1 GHz the chip holds), fragment reads + barriers alone 0.24,
// + operand copies 0.17-0.29, + epilogue 0.17; the full kernel takes 0.93 = their SUM.  A four-buffer ring with counted
// waits, fragment registers double-buffered across the stage barrier, and two workgroups per CU all land within 4 % of that
// (fp8, K = 1536: two workgroups per CU + 10 %), i.e. the tile is bound by what goes through the LDS -- 48 KB of
// fragment reads + 16 KB of LDS-DMA writes per 16-deep step and CU against 512 matrix-core cycles -- not by exposed
// latency.  The next step is therefore fewer LDS bytes per flop (128 x 128 per wavefront, accumulators in AGPRs, one
// wavefront per SIMD), not more pipelining.
//
// Epilogue: score < threshold is tested in the accumulator's own units -- fma(acc, alpha_row, -beta_row) > h_query with a
// conservative margin -- 2 VALU operations per score; the rare survivors are pushed to a workgroup list in LDS and
// finished (exact score formula, exact `s < thr` test of the 128 x 128 kernel, global append) by the first threads.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "hvx_device.h"
#include "hvx_flat_mfma.h"

namespace hvx {

// Measurement switches (MfmaArgs::ablate: parts of the tile kernel switched off -- results WRONG by construction) exist only
// in tuning builds (-DHVX_TUNING); in the release library the tests below are compile-time false.
#ifdef HVX_TUNING
#define HVX_ABLATE(a, bit) (((a).ablate & (bit)) != 0u)
#else
#define HVX_ABLATE(a, bit) (false)
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int kTM = 256, kTN = 256;      // queries x rows per workgroup
constexpr int kAStage = kTM * 128;       // bytes of query values per stage (64 bf16 per tile row)
constexpr int kWgList = 2048;            // survivors a workgroup can hold

#define HVX_GLDS16(gptr, lptr)                                                                                       \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                          \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

// workgroup -> tile, XCD-aware (workgroup b runs on XCD b % 8, every XCD has its own 4 MB L2): the 32 workgroups an XCD
// runs at a time form a super-tile of sup_r row tiles x sup_q query tiles, so each operand tile is fetched into that L2
// once per super-tile and shared; the query blocks of one row block follow each other (the rows come from HBM once).
__device__ __forceinline__ bool tile_coords(const MfmaArgs &a, uint32_t &qt, uint32_t &rt) {
    const uint32_t x = blockIdx.x & 7u, l = blockIdx.x >> 3;
    const uint32_t rq = a.sup_r * a.sup_q, per_rblock = a.sup_qblocks * rq;
    const uint32_t rblock = l / per_rblock, rem = l % per_rblock;
    const uint32_t qblock = rem / rq, rem2 = rem % rq;
    rt = (rblock * a.sup_r + rem2 / a.sup_q) * 8u + x;
    qt = qblock * a.sup_q + rem2 % a.sup_q;
    return rt < a.nr_tiles && qt < a.nq_tiles;
}

// ---- epilogue shared by all builds.  acc = C[m = query][n = row] of this wavefront's 128 x (32 NJ) block.
template <bool FP8, int NJ>
__device__ __forceinline__ void tile_epilogue(const MfmaArgs &a, f32x16 (&acc)[4][NJ], unsigned char *lds, uint32_t q0, uint32_t r0, int wm,
                                              int wn, int tid, float xmax2, uint32_t *wg_overflow) { // 256 or 512 threads
    const int lane = tid & 63, fr = lane & 31, h = lane >> 5;
    // ---- epilogue.  C[m = query][n = row]: lane holds n = lane & 31, m = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
    __syncthreads(); // the stage buffers are free
    float *sH = reinterpret_cast<float *>(lds);                      // [256] per-query pass level
    uint32_t *sCnt = reinterpret_cast<uint32_t *>(lds + 1024);       // [1] survivors of this workgroup
    float *sLv = reinterpret_cast<float *>(lds + 2048);              // [kWgList] accumulator value
    uint32_t *sLc = reinterpret_cast<uint32_t *>(lds + 2048 + kWgList * 4); // [kWgList] (query << 8) | row, tile-local
    const float inf = __uint_as_float(0x7F800000u);
    if (tid < kTM) {
        const uint32_t qq = q0 + (uint32_t)tid;
        float hq = inf; // padded query rows let nothing through
        if (qq < a.b) {
            const float t = a.thr[qq], n2 = a.qn2[qq];
            // s < t  <=>  dot - |x|^2/2 > (|q|^2 - t)/2 (L2)   resp.   dot / |x| > (1 - 2t) |q| (cosine), up to the rounding of
            // the two evaluation orders: the margin lets a few more in, the exact test below decides
            if (a.metric == kL2) hq = 0.5f * (n2 - t) - 4e-6f * (n2 + xmax2);
            else { const float qn = sqrtf(n2); hq = (1.0f - 2.0f * t) * qn - 8e-6f * qn; }
        }
        sH[tid] = hq;
    }
    if (tid == 0) *sCnt = 0u;
    float alpha[NJ], nbeta[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const uint32_t rloc = r0 + (uint32_t)(wn * (32 * NJ) + j * 32 + fr);
        alpha[j] = 0.f;
        nbeta[j] = -inf; // rows past the end of the scan never pass
        if (rloc < a.nrows) {
            const size_t node = a.subset ? a.subset[a.row0 + rloc] : (size_t)a.row0 + rloc;
            const float term = a.rowterm[node];
            const float rsc = FP8 ? a.rowscale[node] : 1.0f;
            if (a.metric == kL2) { alpha[j] = rsc; nbeta[j] = -0.5f * term; }
            else { alpha[j] = term > 0.f ? rsc / term : 0.f; nbeta[j] = 0.f; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h; // tile-local query
            const float hq = sH[m];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (__builtin_fmaf(acc[i][j][e], alpha[j], nbeta[j]) > hq) {
                    const uint32_t pos = atomicAdd(sCnt, 1u);
                    if (pos < (uint32_t)kWgList) {
                        sLv[pos] = acc[i][j][e];
                        sLc[pos] = ((uint32_t)m << 8) | (uint32_t)(wn * (32 * NJ) + j * 32 + fr);
                    }
                }
            }
        }
    __syncthreads();
    uint32_t cnt = *sCnt;
    if (cnt > (uint32_t)kWgList) { // more survivors than the list holds: the scan is repeated unfiltered, never guessed
        if (tid == 0) atomicOr(wg_overflow, 1u);
        cnt = kWgList;
    }
    for (uint32_t t = (uint32_t)tid; t < cnt; t += blockDim.x) {
        const uint32_t code = sLc[t];
        const uint32_t qq = q0 + (code >> 8), rloc = r0 + (code & 255u);
        if (qq >= a.b || rloc >= a.nrows) continue;
        const size_t node = a.subset ? a.subset[a.row0 + rloc] : (size_t)a.row0 + rloc;
        const float term = a.rowterm[node];
        const float dot = FP8 ? sLv[t] * a.rowscale[node] : sLv[t];
        float sc;
        if (a.metric == kL2) {
            sc = (a.qn2[qq] + term) - 2.0f * dot;
            sc = sc < 0.f ? 0.f : sc;
        } else {
            const float den = sqrtf(a.qn2[qq]) * term;
            float c = den > 0.f ? dot / den : 0.f;
            c = c < -1.f ? -1.f : (c > 1.f ? 1.f : c);
            sc = (1.0f - c) * 0.5f;
        }
        if (sc < a.thr[qq]) {
            const uint32_t pos = atomicAdd(&a.cand_cnt[qq], 1u);
            if (pos < a.cand_cap) {
                a.cand_sc[(size_t)qq * a.cand_cap + pos] = sc;
                a.cand_id[(size_t)qq * a.cand_cap + pos] = (uint32_t)node;
            }
        }
    }
}

// ---- the 512-thread build: one workgroup per CU on a 256 x 256 tile, two 64-deep LDS buffers, compiler-visible fragment reads
// (vmcnt(0) + __syncthreads per stage).  Carries the measurement switches (MfmaArgs::ablate).  KIND 0: bf16 rows, 1: fp8 codes.
// PIPE: pin the fragment reads of step kk + 1 ahead of the MFMAs of step kk (the compiler otherwise sinks them next to their use
// and waits for every read pair; measured equal).
template <int KIND, bool PIPE>
__global__ __launch_bounds__(512) void flat_tile256_kernel(MfmaArgs a, float xmax2, uint32_t *wg_overflow) {
    constexpr bool FP8 = KIND == 1;
    constexpr int ROWB = FP8 ? 64 : 128;            // bytes of one row per stage
    constexpr int BSTAGE = kTN * ROWB;
    constexpr int STAGE = kAStage + BSTAGE;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    uint32_t qt, rt;
    if (!tile_coords(a, qt, rt)) return;
    const uint32_t q0 = qt * kTM, r0 = rt * kTN;

    // ---- staging addresses.  Query values: wave w copies tile rows 32 w .. 32 w + 31 as four 1-KB pieces (8 rows each);
    // lane l of piece t lands on row 32 w + 8 t + (l >> 3), physical 16-B slot l & 7, and fetches the logical slot
    // (l & 7) ^ ((row >> 1) & 7) of that row.
    const unsigned char *gA[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t row = (uint32_t)(32 * wave + 8 * t + (lane >> 3));
        const uint32_t slot = (uint32_t)(lane & 7) ^ ((row >> 1) & 7u);
        gA[t] = reinterpret_cast<const unsigned char *>(a.qhi) + (size_t)(q0 + row) * a.dim * 2 + slot * 16; // queries are padded to 256
    }
    constexpr int NB = FP8 ? 2 : 4; // 1-KB pieces of the row tile per wave and stage
    const unsigned char *gB[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        uint32_t row, slot;
        if (FP8) { // 16 rows x 64 B per piece
            row = (uint32_t)(32 * wave + 16 * t + (lane >> 2));
            slot = (uint32_t)(lane & 3) ^ ((row >> 2) & 3u);
        } else {
            row = (uint32_t)(32 * wave + 8 * t + (lane >> 3));
            slot = (uint32_t)(lane & 7) ^ ((row >> 1) & 7u);
        }
        uint32_t rloc = r0 + row;
        if (rloc >= a.nrows) rloc = a.nrows - 1; // clamp: the duplicate is masked in the epilogue
        const size_t node = a.subset ? a.subset[a.row0 + rloc] : (size_t)a.row0 + rloc;
        gB[t] = reinterpret_cast<const unsigned char *>(a.rows) + node * a.dim * (FP8 ? 1 : 2) + slot * 16;
    }
    auto issue_stage = [&](uint32_t kbyteA, uint32_t kbyteB, int buf) {
        unsigned char *sA = lds + buf * STAGE + wave * 4096;          // 32 rows x 128 B of this wave
        unsigned char *sB = lds + buf * STAGE + kAStage + wave * (32 * ROWB);
#pragma unroll
        for (int t = 0; t < 4; ++t) HVX_GLDS16(gA[t] + kbyteA, sA + t * 1024);
#pragma unroll
        for (int t = 0; t < NB; ++t) HVX_GLDS16(gB[t] + kbyteB, sB + t * 1024);
    };

    // ---- fragment read offsets: lane = tile row fr of the 32-row block, depth half h
    const int fr = lane & 31, h = lane >> 5;
    int offA[4]; // per step kk: 16-B slot (2 kk + h) ^ ((fr >> 1) & 7) of row fr (the row's block offset is added as an immediate)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) offA[kk] = fr * 128 + (((2 * kk + h) ^ ((fr >> 1) & 7)) << 4);
    int offB8[2]; // fp8: per step pair: slot (2 jj + h) ^ ((fr >> 2) & 3) of the 64-B row
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) offB8[jj] = fr * 64 + (((2 * jj + h) ^ ((fr >> 2) & 3)) << 4);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const uint32_t nstage = a.dim >> 6;
    issue_stage(0, 0, 0);
    for (uint32_t s = 0; s < nstage; ++s) {
        const int buf = (int)(s & 1u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's pieces of stage s have landed ...
        __syncthreads();                                  // ... everyone's have, and buffer buf ^ 1 is no longer being read
        if (s + 1 < nstage && !HVX_ABLATE(a, 1u)) issue_stage((s + 1) * 128u, (s + 1) * (uint32_t)ROWB, buf ^ 1);
        const unsigned char *sA = lds + buf * STAGE + wm * (128 * 128);
        const unsigned char *sB = lds + buf * STAGE + kAStage + wn * (64 * ROWB);
        // fragments of step kk + 1 are read (fp8: and widened) while the matrix core works on step kk
        uint4 c8[2]; // fp8: the codes of both steps of a pair
        auto load_frags = [&](int kk, bf16x8 (&fa)[4], bf16x8 (&fb)[2]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(sA + i * 4096 + offA[kk]));
            if (FP8) {
                if ((kk & 1) == 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) c8[j] = *reinterpret_cast<const uint4 *>(sB + j * (32 * 64) + offB8[kk >> 1]);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const uint32_t w0 = (kk & 1) ? c8[j].z : c8[j].x, w1 = (kk & 1) ? c8[j].w : c8[j].y;
                    // 4 codes -> 4 f32 (exact) -> their high halves = 4 bf16 (exact)
                    const f32x2 a01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w0, false), a23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w0, true);
                    const f32x2 b01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w1, false), b23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w1, true);
                    uint4 wv;
                    wv.x = __builtin_amdgcn_perm(__float_as_uint(a01[1]), __float_as_uint(a01[0]), 0x07060302u);
                    wv.y = __builtin_amdgcn_perm(__float_as_uint(a23[1]), __float_as_uint(a23[0]), 0x07060302u);
                    wv.z = __builtin_amdgcn_perm(__float_as_uint(b01[1]), __float_as_uint(b01[0]), 0x07060302u);
                    wv.w = __builtin_amdgcn_perm(__float_as_uint(b23[1]), __float_as_uint(b23[0]), 0x07060302u);
                    fb[j] = __builtin_bit_cast(bf16x8, wv);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(sB + j * 4096 + offA[kk]));
            }
        };
        bf16x8 fa0[4], fb0[2], fa1[4], fb1[2];
        load_frags(0, fa0, fb0);
#pragma unroll
        for (int kk = 0; kk < 4; kk += 2) {
            load_frags(kk + 1, fa1, fb1);
            if (PIPE) __builtin_amdgcn_sched_barrier(0); // keep the reads of the next step ahead of this step's MFMAs
            if HVX_ABLATE(a, 2u) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa0[i]));
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(fb0[j]));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[i], fb0[j], acc[i][j], 0, 0, 0);
            }
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < 4) load_frags(kk + 2, fa0, fb0);
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
            if HVX_ABLATE(a, 2u) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fa1[i]));
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(fb1[j]));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[i], fb1[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    if HVX_ABLATE(a, 4u) { // measurement builds only (HVX_FLAT_TILE_ABLATE): keep the accumulators alive, skip the epilogue
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
        if (sum == 123456.789f) *wg_overflow = 2u;
        return;
    }
    tile_epilogue<FP8, 2>(a, acc, lds, q0, r0, wm, wn, tid, xmax2, wg_overflow);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ds_read_b128 as inline asm: the compiler then neither sees an LDS load that "may alias" the LDS-DMA copies in flight (it
// answers that with s_waitcnt vmcnt(0), draining the copy ring -- it did so for the fp8 code reads of an earlier build) nor
// places its own lgkmcnt waits; the kernel states both waits itself, each followed by a sched_barrier (an MFMA may not be
// hoisted over them).
template <int OFF>
__device__ __forceinline__ uint4 lds_read16(uint32_t addr) {
    uint4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void wait_lgkm0() {
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0), nothing else
    __builtin_amdgcn_sched_barrier(0);
}

template <int LGKM> // everything but the LGKM youngest LDS requests has arrived (they return in order)
__device__ __forceinline__ void wait_lgkm() {
    __builtin_amdgcn_s_waitcnt(0xC07F | (LGKM << 8));
    __builtin_amdgcn_sched_barrier(0);
}

// ---- the two-workgroups-per-CU build (see the file header): a workgroup is four wavefronts (one per SIMD, 2 x 2, each 128
// queries x 64 rows) on a 256 x 128 tile, so a CU holds two independent workgroups: while one waits at its barrier, starts a
// tile or runs its epilogue, the other one's MFMAs have the matrix cores.

template <int KIND>
__global__ __launch_bounds__(256, 2) void flat_tile2_kernel(MfmaArgs a, float xmax2, uint32_t *wg_overflow) {
    constexpr bool FP8 = KIND == 1;
    constexpr int TNR = 128;                        // rows per workgroup (queries: kTM = 256)
    constexpr int KS = FP8 ? 64 : 32, SPS = KS / 16;
    constexpr int AROWB = KS * 2, BROWB = 64;       // bytes per tile row and stage
    constexpr int ASTAGE = kTM * AROWB, STAGE = ASTAGE + TNR * BROWB; // 24 KB (bf16) / 40 KB (fp8)
    constexpr int NBUF = FP8 ? 2 : 3;               // 72 KB / 80 KB per workgroup
    constexpr int GA = AROWB / 16, GB = 2, G = GA + GB; // 1-KB copies per wave and stage: queries 64 rows, rows 32 rows per wave
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NBUF * STAGE];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    uint32_t qt, rt;
    if (!tile_coords(a, qt, rt)) return;
    const uint32_t q0 = qt * kTM, r0 = rt * TNR;

    // staging: wave w copies query rows 64 w .. 64 w + 63 and tile rows 32 w .. 32 w + 31 (64-byte rows: 16 rows per 1-KB copy, slot (l & 3) ^ ((row >> 2) & 3); 128-byte rows: 8 rows, slot (l & 7) ^ ((row >> 1) & 7))
    const unsigned char *gA[GA], *gB[GB];
#pragma unroll
    for (int t = 0; t < GA; ++t) {
        uint32_t row, slot;
        if (AROWB == 64) { row = (uint32_t)(64 * wave + 16 * t + (lane >> 2)); slot = (uint32_t)(lane & 3) ^ ((row >> 2) & 3u); }
        else { row = (uint32_t)(64 * wave + 8 * t + (lane >> 3)); slot = (uint32_t)(lane & 7) ^ ((row >> 1) & 7u); }
        gA[t] = reinterpret_cast<const unsigned char *>(a.qhi) + (size_t)(q0 + row) * a.dim * 2 + slot * 16;
    }
#pragma unroll
    for (int t = 0; t < GB; ++t) {
        const uint32_t row = (uint32_t)(32 * wave + 16 * t + (lane >> 2));
        const uint32_t slot = (uint32_t)(lane & 3) ^ ((row >> 2) & 3u);
        uint32_t rloc = r0 + row;
        if (rloc >= a.nrows) rloc = a.nrows - 1;
        const size_t node = a.subset ? a.subset[a.row0 + rloc] : (size_t)a.row0 + rloc;
        gB[t] = reinterpret_cast<const unsigned char *>(a.rows) + node * a.dim * (FP8 ? 1 : 2) + slot * 16;
    }
    auto issue_stage = [&](uint32_t s) {
        unsigned char *sA = lds + (s % NBUF) * STAGE + wave * (64 * AROWB);
        unsigned char *sB = lds + (s % NBUF) * STAGE + ASTAGE + wave * (32 * BROWB);
#pragma unroll
        for (int t = 0; t < GA; ++t) HVX_GLDS16(gA[t] + s * (uint32_t)AROWB, sA + t * 1024);
#pragma unroll
        for (int t = 0; t < GB; ++t) HVX_GLDS16(gB[t] + s * (uint32_t)BROWB, sB + t * 1024);
    };
    const int fr = lane & 31, h = lane >> 5;
    int off128[4], off64[2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) off128[kk] = fr * 128 + (((2 * kk + h) ^ ((fr >> 1) & 7)) << 4);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) off64[jj] = fr * 64 + (((2 * jj + h) ^ ((fr >> 2) & 3)) << 4);
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(lds);
    const uint32_t baseA = lds0 + (uint32_t)(wm * (128 * AROWB)), baseB = lds0 + (uint32_t)(ASTAGE + wn * (64 * BROWB));

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    uint4 c8[2];
    auto read_raw = [&](uint32_t buf, int kk, bf16x8 (&fa)[4], bf16x8 (&fb)[2]) { // 6 reads (bf16; fp8 even steps) or 4 (fp8 odd steps)
        const uint32_t pa = baseA + buf * (uint32_t)STAGE + (uint32_t)(FP8 ? off128[kk] : off64[kk]);
        fa[0] = __builtin_bit_cast(bf16x8, lds_read16<0>(pa));
        fa[1] = __builtin_bit_cast(bf16x8, lds_read16<32 * AROWB>(pa));
        fa[2] = __builtin_bit_cast(bf16x8, lds_read16<64 * AROWB>(pa));
        fa[3] = __builtin_bit_cast(bf16x8, lds_read16<96 * AROWB>(pa));
        if (FP8) {
            if ((kk & 1) == 0) {
                const uint32_t pb = baseB + buf * (uint32_t)STAGE + (uint32_t)off64[kk >> 1];
                c8[0] = lds_read16<0>(pb);
                c8[1] = lds_read16<32 * 64>(pb);
            }
        } else {
            const uint32_t pb = baseB + buf * (uint32_t)STAGE + (uint32_t)off64[kk];
            fb[0] = __builtin_bit_cast(bf16x8, lds_read16<0>(pb));
            fb[1] = __builtin_bit_cast(bf16x8, lds_read16<32 * 64>(pb));
        }
    };
    auto widen = [&](int kk, bf16x8 (&fb)[2]) {
        if (!FP8) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t w0 = (kk & 1) ? c8[j].z : c8[j].x, w1 = (kk & 1) ? c8[j].w : c8[j].y;
            const f32x2 a01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w0, false), a23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w0, true);
            const f32x2 b01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w1, false), b23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w1, true);
            uint4 wv;
            wv.x = __builtin_amdgcn_perm(__float_as_uint(a01[1]), __float_as_uint(a01[0]), 0x07060302u);
            wv.y = __builtin_amdgcn_perm(__float_as_uint(a23[1]), __float_as_uint(a23[0]), 0x07060302u);
            wv.z = __builtin_amdgcn_perm(__float_as_uint(b01[1]), __float_as_uint(b01[0]), 0x07060302u);
            wv.w = __builtin_amdgcn_perm(__float_as_uint(b23[1]), __float_as_uint(b23[0]), 0x07060302u);
            fb[j] = __builtin_bit_cast(bf16x8, wv);
        }
    };
    auto mfma8 = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };

    const uint32_t nstage = a.dim / (uint32_t)KS;
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
        if ((uint32_t)s < nstage) issue_stage((uint32_t)s);
    bf16x8 fa0[4], fb0[2], fa1[4], fb1[2];
    for (uint32_t s = 0; s < nstage; ++s) {
        // stage s has landed when at most the copies of the younger stages in flight (NBUF - 2 of them) are outstanding
        const uint32_t younger = nstage - 1u - s < (uint32_t)(NBUF - 2) ? nstage - 1u - s : (uint32_t)(NBUF - 2);
        if (younger >= 1u) wait_vmcnt<G>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier(); // everyone's copies of stage s are in LDS; nobody still reads the buffer of stage s - 1
        if (s + (uint32_t)(NBUF - 1) < nstage) issue_stage(s + (uint32_t)(NBUF - 1));
        const uint32_t buf = s % NBUF;
        read_raw(buf, 0, fa0, fb0);
#pragma unroll
        for (int kk = 0; kk < SPS; kk += 2) {
            read_raw(buf, kk + 1, fa1, fb1);                  // step kk + 1 requested ...
            if (FP8) wait_lgkm<4>(); else wait_lgkm<6>();      // ... step kk has arrived (fp8: with the codes of both steps)
            widen(kk, fb0);
            widen(kk + 1, fb1);                                // before the next pair's codes replace c8
            mfma8(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < SPS) {
                read_raw(buf, kk + 2, fa0, fb0);
                wait_lgkm<6>();
            } else {
                wait_lgkm<0>();
            }
            mfma8(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    tile_epilogue<FP8, 2>(a, acc, lds, q0, r0, wm, wn, tid, xmax2, wg_overflow);
}

// ---- the MX-scaled fp8 build (round 4; VERDICT r3 missing #6): fp8 rows x fp8 query pieces on v_mfma_scale_f32_32x32x64_f8f6f4, the
// only large-K low-precision matrix instruction of gfx950 (measured 4 447 TFLOP/s against 1 820 for bf16 32x32x16 in
// scripts/mfma_rate_probe.hip).  Same tile, same stage geometry and the same copies as flat_tile2_kernel<1>: a stage is 64 deep,
// a query's stage row is 128 bytes -- but they hold the query as TWO e4m3 pieces (64 hi codes | 64 lo codes, split_queries_mx_kernel)
// instead of 64 bf16 values, and the row codes go into the matrix cores as they are: no v_cvt_pk_f32_fp8 / v_perm widening (8 VALU
// operations per fragment), and K = 64 per instruction.  Per stage and wavefront: 4 + 16 fragment reads (ds_read_b128: 32 consecutive
// codes of a tile row per lane = the operand layout, tests/native/mx_probe.hip), 16 matrix instructions (8 hi, 8 lo; 64 cycles each by
// the bf16-relative rate, i.e. the same 1 024 cycles per stage as 32 bf16 steps -- the gain, if any, is the measured rate and the
// missing VALU work).  The query's power-of-two scale rides in the instruction's block scale (E8M0, the lo piece 4 less); the rows'
// f32 scales stay in the epilogue as before.
typedef __attribute__((ext_vector_type(8))) int i32x8;

__global__ __launch_bounds__(256, 2) void flat_tile2mx_kernel(MfmaArgs a, float xmax2, uint32_t *wg_overflow) {
    constexpr int TNR = 128;
    constexpr int AROWB = 128, BROWB = 64;
    constexpr int ASTAGE = kTM * AROWB, STAGE = ASTAGE + TNR * BROWB; // 40 KB
    constexpr int NBUF = 2;
    constexpr int GA = AROWB / 16, GB = 2;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NBUF * STAGE];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    uint32_t qt, rt;
    if (!tile_coords(a, qt, rt)) return;
    const uint32_t q0 = qt * kTM, r0 = rt * TNR;

    const unsigned char *gA[GA], *gB[GB];
#pragma unroll
    for (int t = 0; t < GA; ++t) {
        const uint32_t row = (uint32_t)(64 * wave + 8 * t + (lane >> 3)), slot = (uint32_t)(lane & 7) ^ ((row >> 1) & 7u);
        gA[t] = reinterpret_cast<const unsigned char *>(a.qhi) + (size_t)(q0 + row) * a.dim * 2 + slot * 16;
    }
#pragma unroll
    for (int t = 0; t < GB; ++t) {
        const uint32_t row = (uint32_t)(32 * wave + 16 * t + (lane >> 2));
        const uint32_t slot = (uint32_t)(lane & 3) ^ ((row >> 2) & 3u);
        uint32_t rloc = r0 + row;
        if (rloc >= a.nrows) rloc = a.nrows - 1;
        const size_t node = a.subset ? a.subset[a.row0 + rloc] : (size_t)a.row0 + rloc;
        gB[t] = reinterpret_cast<const unsigned char *>(a.rows) + node * a.dim + slot * 16;
    }
    auto issue_stage = [&](uint32_t s) {
        unsigned char *sA = lds + (s % NBUF) * STAGE + wave * (64 * AROWB);
        unsigned char *sB = lds + (s % NBUF) * STAGE + ASTAGE + wave * (32 * BROWB);
#pragma unroll
        for (int t = 0; t < GA; ++t) HVX_GLDS16(gA[t] + s * (uint32_t)AROWB, sA + t * 1024);
#pragma unroll
        for (int t = 0; t < GB; ++t) HVX_GLDS16(gB[t] + s * (uint32_t)BROWB, sB + t * 1024);
    };
    const int fr = lane & 31, h = lane >> 5;
    // fragment reads: 32 consecutive codes = two 16-byte slots; queries: hi codes in slots 0-3, lo codes in slots 4-7 of the 128-byte row
    int offA[2][2], offB[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        offA[0][t] = fr * 128 + (((2 * h + t) ^ ((fr >> 1) & 7)) << 4);
        offA[1][t] = fr * 128 + (((4 + 2 * h + t) ^ ((fr >> 1) & 7)) << 4);
        offB[t] = fr * 64 + (((2 * h + t) ^ ((fr >> 2) & 3)) << 4);
    }
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(lds);
    const uint32_t baseA = lds0 + (uint32_t)(wm * (128 * AROWB)), baseB = lds0 + (uint32_t)(ASTAGE + wn * (64 * BROWB));
    // block scales of this lane's four query rows (one per 32-query tile): E8M0 of the hi piece, the lo piece is 2^-4 of it
    int sc_hi[4], sc_lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = (int)a.qexp[q0 + (uint32_t)(wm * 128 + i * 32 + fr)]; // (queries are padded to 256: the padding rows hold zero codes)
        sc_hi[i] = e;
        sc_lo[i] = e >= 4 ? e - 4 : 0;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto frag = [&](uint32_t addr0, uint32_t addr1) -> i32x8 {
        const uint4 x = lds_read16<0>(addr0), y = lds_read16<0>(addr1);
        i32x8 v;
        v[0] = (int)x.x; v[1] = (int)x.y; v[2] = (int)x.z; v[3] = (int)x.w;
        v[4] = (int)y.x; v[5] = (int)y.y; v[6] = (int)y.z; v[7] = (int)y.w;
        return v;
    };
    const uint32_t nstage = a.dim / 64u;
    issue_stage(0);
    for (uint32_t s = 0; s < nstage; ++s) {
        wait_vmcnt<0>();              // this wave's copies of stage s have landed ...
        __builtin_amdgcn_s_barrier(); // ... everyone's have; nobody still reads the other buffer
        if (s + 1 < nstage) issue_stage(s + 1);
        const uint32_t buf = s % NBUF, pa = baseA + buf * (uint32_t)STAGE, pb = baseB + buf * (uint32_t)STAGE;
        // all 20 fragment reads of the stage first, into registers of their own (a read that lands in a register an earlier matrix
        // instruction may still be fetching its operand from would be a hazard nobody checks for an inline-asm read)
        i32x8 fb[2], fh[4], fl[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = frag(pb + (uint32_t)(j * 32 * BROWB + offB[0]), pb + (uint32_t)(j * 32 * BROWB + offB[1]));
#pragma unroll
        for (int i = 0; i < 4; ++i) fh[i] = frag(pa + (uint32_t)(i * 32 * AROWB + offA[0][0]), pa + (uint32_t)(i * 32 * AROWB + offA[0][1]));
#pragma unroll
        for (int i = 0; i < 4; ++i) fl[i] = frag(pa + (uint32_t)(i * 32 * AROWB + offA[1][0]), pa + (uint32_t)(i * 32 * AROWB + offA[1][1]));
        wait_lgkm<8>(); // the row codes and the hi pieces have arrived (LDS answers in order), the lo pieces are on their way
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fh[i], fb[j], acc[i][j], 0, 0, 0, sc_hi[i], 0, 127);
        __builtin_amdgcn_sched_barrier(0);
        wait_lgkm0();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fl[i], fb[j], acc[i][j], 0, 0, 0, sc_lo[i], 0, 127);
        __builtin_amdgcn_sched_barrier(0);
    }
    tile_epilogue<true, 2>(a, acc, lds, q0, r0, wm, wn, tid, xmax2, wg_overflow);
}

// ---- the role-split build (round 4; VERDICT r2 / r3 "the structure that is still untried"): 512 threads = 2 x 4 wavefronts on a
// 256 x 256 tile, 64-deep stages in two LDS buffers, and the two wavefronts of every SIMD in OPPOSITE roles at any instant.  The
// four wavefronts of query half 1 run one phase behind those of half 0 (one extra barrier at the start), and a stage is four
// phases separated by raw s_barriers:
//     half 0 of the wavefronts:   read steps 0-1 | MFMA steps 0-1 | read steps 2-3 | MFMA steps 2-3
//     half 1 (one phase later):                  | read steps 0-1 | MFMA steps 0-1 | read steps 2-3 | MFMA steps 2-3
// so while one wavefront of a SIMD issues its 16 MFMAs (s_setprio 1: nothing else of that SIMD competes for issue slots) the other
// one has its 12 fragment reads (and, at the start of a stage, its 8 LDS-DMA copies of the next stage) in flight.  In the lock-step
// builds above all wavefronts read at once and multiply at once: LDS time and matrix-core time ADD (profiles/history/r02g_ablate_bf16.txt:
// 0.93 ms = 0.35 MFMA + 0.24 reads + copies + epilogue).
// Hand-over rules (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that retires it"): a wavefront waits for
// ITS copies of stage s + 1 (vmcnt(0)) at the end of its second read phase of stage s, before that phase's barrier; the first read
// of stage s + 1 by anybody is at least one barrier later.  The buffer of stage s + 1 was last read (stage s - 1) one phase before
// the first copy into it is issued.
template <int KIND>
__global__ __launch_bounds__(512) void flat_tile8_kernel(MfmaArgs a, float xmax2, uint32_t *wg_overflow) {
    constexpr bool FP8 = KIND == 1;
    constexpr int ROWB = FP8 ? 64 : 128;            // bytes of one tile row per stage (queries: 128)
    constexpr int BSTAGE = kTN * ROWB;
    constexpr int STAGE = kAStage + BSTAGE;         // 64 KB (bf16) / 48 KB (fp8)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    uint32_t qt, rt;
    if (!tile_coords(a, qt, rt)) return;
    const uint32_t q0 = qt * kTM, r0 = rt * kTN;

    // staging: wave w copies query rows 32 w .. 32 w + 31 (four 1-KB pieces) and tile rows 32 w .. 32 w + 31 (four / two pieces)
    const unsigned char *gA[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t row = (uint32_t)(32 * wave + 8 * t + (lane >> 3));
        const uint32_t slot = (uint32_t)(lane & 7) ^ ((row >> 1) & 7u);
        gA[t] = reinterpret_cast<const unsigned char *>(a.qhi) + (size_t)(q0 + row) * a.dim * 2 + slot * 16; // queries are padded to 256
    }
    constexpr int NB = FP8 ? 2 : 4;
    const unsigned char *gB[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
        uint32_t row, slot;
        if (FP8) { row = (uint32_t)(32 * wave + 16 * t + (lane >> 2)); slot = (uint32_t)(lane & 3) ^ ((row >> 2) & 3u); }
        else { row = (uint32_t)(32 * wave + 8 * t + (lane >> 3)); slot = (uint32_t)(lane & 7) ^ ((row >> 1) & 7u); }
        uint32_t rloc = r0 + row;
        if (rloc >= a.nrows) rloc = a.nrows - 1; // clamp: the duplicate is masked in the epilogue
        const size_t node = a.subset ? a.subset[a.row0 + rloc] : (size_t)a.row0 + rloc;
        gB[t] = reinterpret_cast<const unsigned char *>(a.rows) + node * a.dim * (FP8 ? 1 : 2) + slot * 16;
    }
    auto issue_stage = [&](uint32_t s) {
        unsigned char *sA = lds + (s & 1u) * STAGE + wave * 4096;
        unsigned char *sB = lds + (s & 1u) * STAGE + kAStage + wave * (32 * ROWB);
#pragma unroll
        for (int t = 0; t < 4; ++t) HVX_GLDS16(gA[t] + s * 128u, sA + t * 1024);
#pragma unroll
        for (int t = 0; t < NB; ++t) HVX_GLDS16(gB[t] + s * (uint32_t)ROWB, sB + t * 1024);
    };
    const int fr = lane & 31, h = lane >> 5;
    int off128[4], off64[2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) off128[kk] = fr * 128 + (((2 * kk + h) ^ ((fr >> 1) & 7)) << 4);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) off64[jj] = fr * 64 + (((2 * jj + h) ^ ((fr >> 2) & 3)) << 4);
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(lds);
    const uint32_t baseA = lds0 + (uint32_t)(wm * (128 * 128)), baseB = lds0 + (uint32_t)(kAStage + wn * (64 * ROWB));

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    uint4 c8[2];
    auto read_raw = [&](uint32_t buf, int kk, bf16x8 (&fa)[4], bf16x8 (&fb)[2]) {
        const uint32_t pa = baseA + buf * (uint32_t)STAGE + (uint32_t)off128[kk];
        fa[0] = __builtin_bit_cast(bf16x8, lds_read16<0>(pa));
        fa[1] = __builtin_bit_cast(bf16x8, lds_read16<4096>(pa));
        fa[2] = __builtin_bit_cast(bf16x8, lds_read16<8192>(pa));
        fa[3] = __builtin_bit_cast(bf16x8, lds_read16<12288>(pa));
        if (FP8) {
            if ((kk & 1) == 0) {
                const uint32_t pb = baseB + buf * (uint32_t)STAGE + (uint32_t)off64[kk >> 1];
                c8[0] = lds_read16<0>(pb);
                c8[1] = lds_read16<32 * 64>(pb);
            }
        } else {
            const uint32_t pb = baseB + buf * (uint32_t)STAGE + (uint32_t)off128[kk];
            fb[0] = __builtin_bit_cast(bf16x8, lds_read16<0>(pb));
            fb[1] = __builtin_bit_cast(bf16x8, lds_read16<4096>(pb));
        }
    };
    auto widen = [&](int kk, bf16x8 (&fb)[2]) {
        if (!FP8) return;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t w0 = (kk & 1) ? c8[j].z : c8[j].x, w1 = (kk & 1) ? c8[j].w : c8[j].y;
            const f32x2 a01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w0, false), a23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w0, true);
            const f32x2 b01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w1, false), b23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w1, true);
            uint4 wv;
            wv.x = __builtin_amdgcn_perm(__float_as_uint(a01[1]), __float_as_uint(a01[0]), 0x07060302u);
            wv.y = __builtin_amdgcn_perm(__float_as_uint(a23[1]), __float_as_uint(a23[0]), 0x07060302u);
            wv.z = __builtin_amdgcn_perm(__float_as_uint(b01[1]), __float_as_uint(b01[0]), 0x07060302u);
            wv.w = __builtin_amdgcn_perm(__float_as_uint(b23[1]), __float_as_uint(b23[0]), 0x07060302u);
            fb[j] = __builtin_bit_cast(bf16x8, wv);
        }
    };
    auto mfma8 = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    bf16x8 fa0[4], fb0[2], fa1[4], fb1[2];
    auto read_phase = [&](uint32_t buf, int kk) __attribute__((always_inline)) { // the fragments of steps kk and kk + 1, complete before the barrier
        read_raw(buf, kk, fa0, fb0);
        read_raw(buf, kk + 1, fa1, fb1);
        wait_lgkm<0>();
        widen(kk, fb0);
        widen(kk + 1, fb1);
    };
    auto mfma_phase = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        mfma8(fa0, fb0);
        mfma8(fa1, fb1);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    const uint32_t nstage = a.dim >> 6;
    issue_stage(0);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                 // stage 0 is in LDS for everyone
    if (wm == 1) __builtin_amdgcn_s_barrier();    // query half 1 runs one phase behind half 0 from here on
    for (uint32_t s = 0; s < nstage; ++s) {
        const uint32_t buf = s & 1u;
        if (s + 1u < nstage) issue_stage(s + 1u); // into the other buffer: last read one phase ago (by the other half) or earlier
        read_phase(buf, 0);
        __builtin_amdgcn_s_barrier();
        mfma_phase();
        __builtin_amdgcn_s_barrier();
        read_phase(buf, 2);
        if (s + 1u < nstage) wait_vmcnt<0>();     // this wavefront's copies of stage s + 1 have landed: readable after the next barrier
        __builtin_amdgcn_s_barrier();
        mfma_phase();
        __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();    // half 1's last phase
    tile_epilogue<FP8, 2>(a, acc, lds, q0, r0, wm, wn, tid, xmax2, wg_overflow);
}

#ifdef HVX_TUNING // experimental: not in the release library until it has a hardware parity record (scripts/gpu_tile4_round.sh)
// ---- the 128 x 128-per-wavefront build: fewer LDS bytes per flop.  Four wavefronts (one per SIMD, 2 x 2) on a 256 x 256
// tile, each with 4 x 4 accumulators (256 registers: the kernel runs one wavefront per SIMD and spills into AGPRs): a
// 16-deep step is 16 MFMAs per 8 fragment reads (the 128 x 64 builds: 8 per 6), and the tile moves two thirds of the
// LDS-DMA bytes of two 256 x 128 tiles.  With a single wavefront per SIMD nothing overlaps by itself, so the fragments of
// step u + 1 are requested before the MFMAs of step u are issued -- also across the stage barrier -- and the fp8 widening
// of step u + 1 sits between the two halves of step u's MFMAs.  Ring of four 32-deep stages (bf16 rows, 128 KB) or three
// 64-deep stages (fp8 codes, 144 KB), counted vmcnt, raw s_barrier, fragment reads as inline asm with stated waits.
// INTER: the LDS-DMA copies of the stage a boundary step opens are issued BETWEEN that step's MFMAs (one 1-KB piece per two
// MFMAs) instead of in one burst behind the barrier: a piece costs its wavefront ~60 issue cycles, eight of them in a row leave the
// matrix core of a one-wavefront SIMD idle for half a stage's worth of MFMA time.
template <int KIND, bool INTER>
__global__ __launch_bounds__(256, 1) void flat_tile4_kernel(MfmaArgs a, float xmax2, uint32_t *wg_overflow) {
    constexpr bool FP8 = KIND == 1;
    constexpr int KS = FP8 ? 64 : 32, SPS = KS / 16;
    constexpr int AROWB = KS * 2, BROWB = 64;
    constexpr int ASTAGE = kTM * AROWB, STAGE = ASTAGE + kTN * BROWB; // 32 KB (bf16) / 48 KB (fp8)
    constexpr int NBUF = FP8 ? 3 : 4;
    constexpr int GA = AROWB / 16, GB = 4, G = GA + GB; // 1-KB copies per wave and stage: 64 query rows and 64 tile rows per wave
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NBUF * STAGE];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    uint32_t qt, rt;
    if (!tile_coords(a, qt, rt)) return;
    const uint32_t q0 = qt * kTM, r0 = rt * kTN;

    const unsigned char *gA[GA], *gB[GB];
#pragma unroll
    for (int t = 0; t < GA; ++t) {
        uint32_t row, slot;
        if (AROWB == 64) { row = (uint32_t)(64 * wave + 16 * t + (lane >> 2)); slot = (uint32_t)(lane & 3) ^ ((row >> 2) & 3u); }
        else { row = (uint32_t)(64 * wave + 8 * t + (lane >> 3)); slot = (uint32_t)(lane & 7) ^ ((row >> 1) & 7u); }
        gA[t] = reinterpret_cast<const unsigned char *>(a.qhi) + (size_t)(q0 + row) * a.dim * 2 + slot * 16;
    }
#pragma unroll
    for (int t = 0; t < GB; ++t) {
        const uint32_t row = (uint32_t)(64 * wave + 16 * t + (lane >> 2));
        const uint32_t slot = (uint32_t)(lane & 3) ^ ((row >> 2) & 3u);
        uint32_t rloc = r0 + row;
        if (rloc >= a.nrows) rloc = a.nrows - 1;
        const size_t node = a.subset ? a.subset[a.row0 + rloc] : (size_t)a.row0 + rloc;
        gB[t] = reinterpret_cast<const unsigned char *>(a.rows) + node * a.dim * (FP8 ? 1 : 2) + slot * 16;
    }
    auto issue_stage = [&](uint32_t s) {
        unsigned char *sA = lds + (s % NBUF) * STAGE + wave * (64 * AROWB);
        unsigned char *sB = lds + (s % NBUF) * STAGE + ASTAGE + wave * (64 * BROWB);
#pragma unroll
        for (int t = 0; t < GA; ++t) HVX_GLDS16(gA[t] + s * (uint32_t)AROWB, sA + t * 1024);
#pragma unroll
        for (int t = 0; t < GB; ++t) HVX_GLDS16(gB[t] + s * (uint32_t)BROWB, sB + t * 1024);
    };
    auto issue_piece = [&](uint32_t s, int p) { // piece p of stage s, same order as issue_stage (the vmcnt arithmetic counts pieces)
        if (p < GA) HVX_GLDS16(gA[p] + s * (uint32_t)AROWB, lds + (s % NBUF) * STAGE + wave * (64 * AROWB) + p * 1024);
        else HVX_GLDS16(gB[p - GA] + s * (uint32_t)BROWB, lds + (s % NBUF) * STAGE + ASTAGE + wave * (64 * BROWB) + (p - GA) * 1024);
    };
    const int fr = lane & 31, h = lane >> 5;
    int off128[4], off64[2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) off128[kk] = fr * 128 + (((2 * kk + h) ^ ((fr >> 1) & 7)) << 4);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) off64[jj] = fr * 64 + (((2 * jj + h) ^ ((fr >> 2) & 3)) << 4);
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(lds);
    const uint32_t baseA = lds0 + (uint32_t)(wm * (128 * AROWB)), baseB = lds0 + (uint32_t)(ASTAGE + wn * (128 * BROWB));

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    uint4 c8[4]; // fp8: the codes of both steps of a pair, four 32-row blocks
    auto read_raw = [&](uint32_t buf, int kk, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) { // 8 reads (bf16; fp8 even steps), 4 (fp8 odd steps)
        const uint32_t pa = baseA + buf * (uint32_t)STAGE + (uint32_t)(FP8 ? off128[kk] : off64[kk]);
        fa[0] = __builtin_bit_cast(bf16x8, lds_read16<0>(pa));
        fa[1] = __builtin_bit_cast(bf16x8, lds_read16<32 * AROWB>(pa));
        fa[2] = __builtin_bit_cast(bf16x8, lds_read16<64 * AROWB>(pa));
        fa[3] = __builtin_bit_cast(bf16x8, lds_read16<96 * AROWB>(pa));
        if (FP8) {
            if ((kk & 1) == 0) {
                const uint32_t pb = baseB + buf * (uint32_t)STAGE + (uint32_t)off64[kk >> 1];
                c8[0] = lds_read16<0>(pb);
                c8[1] = lds_read16<32 * 64>(pb);
                c8[2] = lds_read16<64 * 64>(pb);
                c8[3] = lds_read16<96 * 64>(pb);
            }
        } else {
            const uint32_t pb = baseB + buf * (uint32_t)STAGE + (uint32_t)off64[kk];
            fb[0] = __builtin_bit_cast(bf16x8, lds_read16<0>(pb));
            fb[1] = __builtin_bit_cast(bf16x8, lds_read16<32 * 64>(pb));
            fb[2] = __builtin_bit_cast(bf16x8, lds_read16<64 * 64>(pb));
            fb[3] = __builtin_bit_cast(bf16x8, lds_read16<96 * 64>(pb));
        }
    };
    auto widen = [&](int kk, bf16x8 (&fb)[4]) {
        if (!FP8) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t w0 = (kk & 1) ? c8[j].z : c8[j].x, w1 = (kk & 1) ? c8[j].w : c8[j].y;
            const f32x2 a01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w0, false), a23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w0, true);
            const f32x2 b01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w1, false), b23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w1, true);
            uint4 wv;
            wv.x = __builtin_amdgcn_perm(__float_as_uint(a01[1]), __float_as_uint(a01[0]), 0x07060302u);
            wv.y = __builtin_amdgcn_perm(__float_as_uint(a23[1]), __float_as_uint(a23[0]), 0x07060302u);
            wv.z = __builtin_amdgcn_perm(__float_as_uint(b01[1]), __float_as_uint(b01[0]), 0x07060302u);
            wv.w = __builtin_amdgcn_perm(__float_as_uint(b23[1]), __float_as_uint(b23[0]), 0x07060302u);
            fb[j] = __builtin_bit_cast(bf16x8, wv);
        }
    };
    auto mfma8 = [&](int half, const bf16x8 (&fa)[4], const bf16x8 (&fb)[4]) {
#pragma unroll
        for (int i = 2 * half; i < 2 * half + 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    const uint32_t nstage = a.dim / (uint32_t)KS;
    auto wait_stage = [&](uint32_t s) { // stage s has landed once at most the copies of the younger stages in flight are outstanding
        const uint32_t younger = nstage - 1u - s < (uint32_t)(NBUF - 2) ? nstage - 1u - s : (uint32_t)(NBUF - 2);
        if (younger >= 2u) wait_vmcnt<2 * G>();
        else if (younger == 1u) wait_vmcnt<G>();
        else wait_vmcnt<0>();
    };
    // one MFMA step: `cur` holds step u (fp8: already widened); `nxt` receives step u + 1 of buffer nbuf (boundary: it opens stage s + 1)
    auto step = [&](bool has_next, bool boundary, uint32_t s, uint32_t nbuf, int nkk, bf16x8 (&fa_c)[4], bf16x8 (&fb_c)[4], bf16x8 (&fa_n)[4],
                    bf16x8 (&fb_n)[4]) {
        wait_lgkm0(); // the reads of `cur` were requested sixteen MFMAs ago
        bool copies = false; // INTER: this step carries the copies of stage s + NBUF between its MFMAs
        if (has_next) {
            if (boundary) {
                wait_stage(s + 1);            // this wave's copies of stage s + 1 have landed (and it has read stage s out: the wait above)
                __builtin_amdgcn_s_barrier(); // everyone's have; the buffer of stage s is free
                if (s + (uint32_t)NBUF < nstage) {
                    if (INTER) copies = true;
                    else issue_stage(s + (uint32_t)NBUF);
                }
            }
            read_raw(nbuf, nkk, fa_n, fb_n);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (INTER && boundary) { // uniform branch (copies is the same for the whole workgroup); the MFMA order is the one of mfma8
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int i = 2 * half; i < 2 * half + 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_c[i], fb_c[j], acc[i][j], 0, 0, 0);
                        acc[i][j + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_c[i], fb_c[j + 1], acc[i][j + 1], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        const int slot = (i * 4 + j) >> 1; // 0 .. 7: the G pieces spread over eight slots
                        if (copies) {
#pragma unroll
                            for (int p = slot * G / 8; p < (slot + 1) * G / 8; ++p) issue_piece(s + (uint32_t)NBUF, p);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                if (half == 0 && FP8 && has_next) {
                    wait_lgkm0();
                    widen(nkk, fb_n);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            return;
        }
        mfma8(0, fa_c, fb_c);
        __builtin_amdgcn_sched_barrier(0);
        if (FP8 && has_next) { // the codes of step u + 1 were requested eight MFMAs ago
            wait_lgkm0();
            widen(nkk, fb_n);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma8(1, fa_c, fb_c);
        __builtin_amdgcn_sched_barrier(0);
    };

#pragma unroll
    for (int s = 0; s < NBUF; ++s)
        if ((uint32_t)s < nstage) issue_stage((uint32_t)s);
    bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
    {   // stage 0 has landed when the NBUF - 1 younger ones are all that is outstanding
        const uint32_t younger = nstage - 1u < (uint32_t)(NBUF - 1) ? nstage - 1u : (uint32_t)(NBUF - 1);
        if (younger >= 3u) wait_vmcnt<3 * G>();
        else if (younger == 2u) wait_vmcnt<2 * G>();
        else if (younger == 1u) wait_vmcnt<G>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        read_raw(0, 0, fa0, fb0);
        if (FP8) { wait_lgkm0(); widen(0, fb0); }
    }
    auto stage = [&](uint32_t s, bool more) {
        const uint32_t buf = s % NBUF;
#pragma unroll
        for (int kk = 0; kk < SPS; kk += 2) {
            step(true, false, s, buf, kk + 1, fa0, fb0, fa1, fb1);
            if (kk + 2 < SPS) step(true, false, s, buf, kk + 2, fa1, fb1, fa0, fb0);
            else step(more, true, s, (s + 1) % NBUF, 0, fa1, fb1, fa0, fb0);
        }
    };
    for (uint32_t s = 0; s + 1 < nstage; ++s) stage(s, true);
    stage(nstage - 1u, false); // peeled: the last stage requests nothing beyond itself (and the exit path carries no outstanding read)
    tile_epilogue<FP8, 4>(a, acc, lds, q0, r0, wm, wn, tid, xmax2, wg_overflow);
}

#endif // HVX_TUNING

hipError_t launch_flat_tile256(const MfmaArgs &a, int kind, uint32_t bpad, float xmax2, uint32_t *wg_overflow, uint32_t build, hipStream_t s) {
    if (a.nrows == 0) return hipSuccess;
    if (a.dim % 64u != 0u || bpad % (uint32_t)kTM != 0u) return hipErrorInvalidValue;
    MfmaArgs t = a;
    t.nq_tiles = bpad / kTM;
    t.nr_tiles = (a.nrows + kTN - 1) / kTN;
    t.sup_q = std::min<uint32_t>(t.nq_tiles, 8u);
    t.sup_r = 32u / t.sup_q;
    t.sup_qblocks = (t.nq_tiles + t.sup_q - 1) / t.sup_q;
    t.ablate = 0;
    const uint32_t rblocks = (t.nr_tiles + 8u * t.sup_r - 1) / (8u * t.sup_r);
    const dim3 grid(8u * rblocks * t.sup_qblocks * t.sup_r * t.sup_q);
#ifdef HVX_TUNING
    if (const char *e = getenv("HVX_FLAT_TILE_ABLATE")) t.ablate = (uint32_t)atoi(e);
    const int tuning_build = [] { const char *e = getenv("HVX_FLAT_TILE_BUILD"); return e ? atoi(e) : 0; }();
    if (tuning_build == 6 || tuning_build == 7) { // build 4 / 5 with the stage copies issued between the MFMAs (6: bf16 only, 7: fp8 too)
        if (kind == 1 && tuning_build == 7) { hipLaunchKernelGGL((flat_tile4_kernel<1, true>), grid, dim3(256), 0, s, t, xmax2, wg_overflow); return hipGetLastError(); }
        if (kind != 1) { hipLaunchKernelGGL((flat_tile4_kernel<0, true>), grid, dim3(256), 0, s, t, xmax2, wg_overflow); return hipGetLastError(); }
    }
    if ((tuning_build == 4 && kind != 1) || tuning_build == 5) { // 256 x 256 tiles, 256 threads of 128 x 128 each, one workgroup per CU:
        // bf16 rows passed their one parity run, the fp8 instantiation returned wrong candidates once (v_mov copies of fragment
        // registers right behind their inline-asm ds_read; scripts/lint_asm_lds.py finds them) and has not run since.  Untimed.
        if (kind == 1) hipLaunchKernelGGL((flat_tile4_kernel<1, false>), grid, dim3(256), 0, s, t, xmax2, wg_overflow);
        else hipLaunchKernelGGL((flat_tile4_kernel<0, false>), grid, dim3(256), 0, s, t, xmax2, wg_overflow);
        return hipGetLastError();
    }
#endif
    if (build == 3 && kind != 1) return hipErrorInvalidValue; // (the MX build multiplies fp8 codes)
    if (build == 0 || build == 3) { // 256 x 128 tiles, 256 threads, two workgroups per CU: super-tiles of 64 workgroups per XCD
        t.nr_tiles = (a.nrows + 127u) / 128u;
        t.sup_r = 64u / t.sup_q;
        const uint32_t rb = (t.nr_tiles + 8u * t.sup_r - 1) / (8u * t.sup_r);
        const dim3 grid2(8u * rb * t.sup_qblocks * t.sup_r * t.sup_q);
        if (build == 3) hipLaunchKernelGGL(flat_tile2mx_kernel, grid2, dim3(256), 0, s, t, xmax2, wg_overflow);
        else if (kind == 1) hipLaunchKernelGGL((flat_tile2_kernel<1>), grid2, dim3(256), 0, s, t, xmax2, wg_overflow);
        else hipLaunchKernelGGL((flat_tile2_kernel<0>), grid2, dim3(256), 0, s, t, xmax2, wg_overflow);
        return hipGetLastError();
    }
    if (build == 2) { // the role-split build: 512 threads, the two wavefronts of a SIMD in opposite roles (flat_tile8_kernel)
        if (kind == 1) hipLaunchKernelGGL((flat_tile8_kernel<1>), grid, dim3(512), 0, s, t, xmax2, wg_overflow);
        else hipLaunchKernelGGL((flat_tile8_kernel<0>), grid, dim3(512), 0, s, t, xmax2, wg_overflow);
        return hipGetLastError();
    }
    // one 512-thread workgroup per CU (256 x 256 tiles, two LDS buffers)
    bool pipe = true;
#ifdef HVX_TUNING
    if (const char *e = getenv("HVX_FLAT_TILE_PIPE")) pipe = e[0] != '0';
#endif
    if (kind == 1) {
        if (pipe) hipLaunchKernelGGL((flat_tile256_kernel<1, true>), grid, dim3(512), 0, s, t, xmax2, wg_overflow);
        else hipLaunchKernelGGL((flat_tile256_kernel<1, false>), grid, dim3(512), 0, s, t, xmax2, wg_overflow);
    } else {
        if (pipe) hipLaunchKernelGGL((flat_tile256_kernel<0, true>), grid, dim3(512), 0, s, t, xmax2, wg_overflow);
        else hipLaunchKernelGGL((flat_tile256_kernel<0, false>), grid, dim3(512), 0, s, t, xmax2, wg_overflow);
    }
    return hipGetLastError();
}

} // namespace hvx
