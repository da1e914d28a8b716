// hvx_hnsw_pair_cos.hip -- instantiations of the owner / gatherer HNSW kernel (two wavefronts per query, hvx_hnsw_pair.h): f32 rows, metric kCosine.
#include "hvx_hnsw_pair.h"

namespace hvx {
hipError_t launch_hnsw_pair_cos(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    return launch_pair_r<kCosine, false>(a, b, g, s);
}
} // namespace hvx
