// hvx_hnsw_wave_cos.hip -- half-cosine instantiations of the one-wavefront-per-query HNSW kernel.
#include "hvx_hnsw_wave.h"

namespace hvx {
hipError_t launch_hnsw_wave_cos(const HnswArgs &a, uint32_t b, uint32_t log2cap, size_t lds, hipStream_t s) {
    return launch_wave_r<kCosine>(a, b, log2cap, lds, s);
}
} // namespace hvx
