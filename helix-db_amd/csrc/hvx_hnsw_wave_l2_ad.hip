// hvx_hnsw_wave_l2_ad.hip -- squared-Euclidean, f32 rows, NON-strict layer-0 arms (pre/post sampling; SimHash filtering
// is cosine-only, policy.rs:67-91): AD instantiations of the one-wavefront-per-query HNSW kernel.
#include "hvx_hnsw_wave.h"

namespace hvx {
hipError_t launch_hnsw_wave_l2_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    return launch_wave_ad<kL2, false>(a, b, g, s);
}
} // namespace hvx
