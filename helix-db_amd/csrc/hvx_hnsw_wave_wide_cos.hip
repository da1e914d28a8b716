// hvx_hnsw_wave_wide_cos.hip -- the strict-exhaustive arm with WIDE register beams (448 / 832 entries: ef 353 .. 800), f32 rows, metric kCosine:
// instantiations of the one-wavefront-per-query HNSW kernel (hvx_hnsw_wave.h), one query per SIMD.  search.rs:267-1067 has no beam
// limit (parameters.rs:118-133 only asks ef >= k); rounds 1-3 sent ef > 352 to the four-wavefront general kernel (4-12 x slower).
#include "hvx_hnsw_wave.h"

namespace hvx {
hipError_t launch_hnsw_wave_wide_cos(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    return launch_wave_wide_r<kCosine, false>(a, b, g, s);
}
} // namespace hvx
