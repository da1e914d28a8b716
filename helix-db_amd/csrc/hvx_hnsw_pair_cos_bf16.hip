// hvx_hnsw_pair_cos_bf16.hip -- instantiations of the owner / gatherer HNSW kernel (two wavefronts per query, hvx_hnsw_pair.h): bf16 rows, metric kCosine.
#include "hvx_hnsw_pair.h"

namespace hvx {
hipError_t launch_hnsw_pair_cos_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    return launch_pair_r<kCosine, true>(a, b, g, s);
}
} // namespace hvx
