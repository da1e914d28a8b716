// hvx_hnsw_wave_l2_bf16_ad.hip -- squared-Euclidean, bf16 rows, NON-strict layer-0 arms: AD instantiations of the
// one-wavefront-per-query HNSW kernel.
#include "hvx_hnsw_wave.h"

namespace hvx {
hipError_t launch_hnsw_wave_l2_bf16_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    return launch_wave_ad<kL2, true>(a, b, g, s);
}
} // namespace hvx
