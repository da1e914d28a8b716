// hvx_hnsw_wave_occ2_l2_bf16_ad.hip -- the non-strict layer-0 arms over bf16 rows (config #4 storage), budgeted for TWO queries per
// SIMD, squared-Euclidean: SearchParams::new(k) on execution lanes / the batcher for reduced-precision images.
#include "hvx_hnsw_wave.h"

namespace hvx {
hipError_t launch_hnsw_wave_occ2_l2_bf16_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    if (a.ix.dtype != HVX_BF16 || !a.adaptive) return hipErrorInvalidValue;
    return a.ad.stats ? launch_wave_r<kL2, true, true, true, 2>(a, b, g, s) : launch_wave_r<kL2, true, true, false, 2>(a, b, g, s);
}
} // namespace hvx
