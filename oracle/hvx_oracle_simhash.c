/*
 * hvx_oracle_simhash.c -- CPU ORACLE (test infrastructure only; see hvx_oracle.h): SimHash projections.
 *
 * Restates crates/db/src/search/vector/unaligned_vector/simhash.rs:123-178 (hyperplane generation),
 * :263-291 (hash_from_slice), :36-55 (collision / hamming), simhash.rs:44-59 (order code) and
 * randomness.rs:104-120 (query-derived seed).
 *
 * Third-party arithmetic that is NOT under /root/reference: `rand = "0.10"` (Cargo.lock: rand 0.10.2,
 * rand_core 0.10.1, chacha20 0.10.1).  `StdRng` is ChaCha with 12 rounds; `SeedableRng::seed_from_u64`
 * expands the u64 into the 32-byte key with a PCG32 stream (multiplier 6364136223846793005, increment
 * 11634580027462260723, XSH-RR output, state advanced before each word, words little-endian);
 * block counter and stream id start at 0; `random::<f32>()` takes one u32, keeps its top 24 bits and
 * scales by 2^-24.  This restatement is PINNED by the reference's own known-answer test
 *   SimHasher(dim = 3, seed = 42).hash([1, 2, 3]) == 0x6d91_a757_8862_6786   (simhash_registry.rs:344-362)
 * which consumes 192 consecutive outputs of the stream (tests/test_oracle_golden.py).
 * The query-time sampler of the non-strict search arms (`random::<f32>() < p`, hvx_oracle_adaptive.inc) draws from this
 * same pinned generator; `random_range` (their two choose_index fallback sites) is restated there from rand's
 * published algorithm but has no known answer in the reference: parity unpinned for those two sites.
 */
#include "hvx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t rotl32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }

/* rand_core SeedableRng::seed_from_u64: PCG32 (XSH-RR) words, little-endian, into the ChaCha key */
static void seed_from_u64(uint64_t state, uint32_t key[8]) {
    const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    for (int i = 0; i < 8; ++i) {
        state = state * MUL + INC;
        uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
}

#define QR(a, b, c, d)                                                                     \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12); \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7)

static void chacha12_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
    uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                       key[4], key[5], key[6], key[7], (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t x[16];
    memcpy(x, st, sizeof(x));
    for (int r = 0; r < 6; ++r) { /* 12 rounds = 6 double rounds */
        QR(0, 4, 8, 12); QR(1, 5, 9, 13); QR(2, 6, 10, 14); QR(3, 7, 11, 15);
        QR(0, 5, 10, 15); QR(1, 6, 11, 12); QR(2, 7, 8, 13); QR(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + st[i];
}

/* exported for the query RNG session in hvx_oracle_adaptive.inc */
void orc_stdrng_key(uint64_t seed, uint32_t key[8]) { seed_from_u64(seed, key); }
void orc_chacha12_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) { chacha12_block(key, counter, out); }

/* the first n outputs of StdRng::seed_from_u64(seed).next_u32() */
void orc_stdrng_u32(uint64_t seed, uint32_t *out, uint32_t n) {
    uint32_t key[8], blk[16];
    seed_from_u64(seed, key);
    uint64_t ctr = 0;
    for (uint32_t i = 0; i < n; i += 16) {
        chacha12_block(key, ctr++, blk);
        uint32_t m = n - i < 16 ? n - i : 16;
        memcpy(out + i, blk, m * 4);
    }
}

/* unaligned_vector/simhash.rs:123-178: 64 hyperplanes, plane-major, component = u*2-1, each normalised in f32 */
int orc_simhash_planes(uint32_t dim, uint64_t seed, float *planes) {
    const uint64_t total = 64ull * dim;
    if (total > 0xFFFFFFFFull) return ORC_ERR_DIMENSION;
    uint32_t *w = (uint32_t *)malloc((size_t)total * 4);
    if (!w) return ORC_ERR_INVARIANT;
    orc_stdrng_u32(seed, w, (uint32_t)total);
    for (uint32_t p = 0; p < 64; ++p) {
        float *pl = planes + (size_t)p * dim;
        for (uint32_t d = 0; d < dim; ++d) {
            float value = (float)(w[(size_t)p * dim + d] >> 8) * (1.0f / 16777216.0f); /* random::<f32>() */
            pl[d] = value * 2.0f - 1.0f;
        }
        float s = 0.0f;
        for (uint32_t d = 0; d < dim; ++d) { float t = pl[d] * pl[d]; s += t; }
        float norm = sqrtf(s);
        if (norm > 1e-10f)
            for (uint32_t d = 0; d < dim; ++d) pl[d] /= norm;
    }
    free(w);
    return ORC_OK;
}

/* unaligned_vector/simhash.rs:263-291 hash_from_slice: bit p = (sequential unfused dot with plane p) > 0 */
uint64_t orc_simhash_hash(const float *planes, const float *vec, uint32_t dim) {
    uint64_t bits = 0;
    for (uint32_t p = 0; p < 64; ++p) {
        float dot = 0.0f;
        const float *pl = planes + (size_t)p * dim;
        for (uint32_t d = 0; d < dim; ++d) { float t = vec[d] * pl[d]; dot += t; }
        if (dot > 0.0f) bits |= 1ull << p;
    }
    return bits;
}

/* simhash.rs:44-59 order_code_from_simhash_bits: 4 bands of 16 bits, bit-planes interleaved high to low */
uint64_t orc_order_code(uint64_t bits) {
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

/* unaligned_vector/simhash.rs:36-55 */
uint32_t orc_simhash_collisions(uint64_t a, uint64_t b) { return 64u - (uint32_t)__builtin_popcountll(a ^ b); }

/* randomness.rs:104-120: seed of the query-local sampling RNG */
uint64_t orc_query_seed(uint64_t query_simhash, uint64_t entry_point, uint64_t ef) {
    return query_simhash ^ ((entry_point << 17) | (entry_point >> 47)) ^ ((ef << 7) | (ef >> 57));
}
