// hvx_hnsw_pair_l2.hip -- instantiations of the owner / gatherer HNSW kernel (two wavefronts per query, hvx_hnsw_pair.h): f32 rows, metric kL2.
#include "hvx_hnsw_pair.h"

namespace hvx {
hipError_t launch_hnsw_pair_l2(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    return launch_pair_r<kL2, false>(a, b, g, s);
}
} // namespace hvx
