// hvx_hnsw_wave_cos_bf16.hip -- half-cosine, bf16 rows: instantiations of the one-wavefront-per-query HNSW kernel.
#include "hvx_hnsw_wave.h"

namespace hvx {
hipError_t launch_hnsw_wave_cos_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    return launch_wave_r<kCosine, true>(a, b, g, s);
}
} // namespace hvx
