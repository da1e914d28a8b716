// hvx_hnsw_wave_occ2_bf16.hip -- two-queries-per-SIMD build of the wave kernel over bf16 rows (strict-exhaustive arm).
#include "hvx_hnsw_wave.h"

namespace hvx {
hipError_t launch_hnsw_wave_occ2_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    if (a.ix.dtype != HVX_BF16 || a.adaptive) return hipErrorInvalidValue;
    return a.ix.metric == kL2 ? launch_wave_r<kL2, true, false, true, 2>(a, b, g, s) : launch_wave_r<kCosine, true, false, true, 2>(a, b, g, s);
}
} // namespace hvx
