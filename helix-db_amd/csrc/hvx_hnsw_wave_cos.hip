// hvx_hnsw_wave_cos.hip -- half-cosine, f32 rows: instantiations of the one-wavefront-per-query HNSW kernel.
#include "hvx_hnsw_wave.h"

namespace hvx {
hipError_t launch_hnsw_wave_cos(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    return launch_wave_r<kCosine, false>(a, b, g, s);
}
} // namespace hvx
