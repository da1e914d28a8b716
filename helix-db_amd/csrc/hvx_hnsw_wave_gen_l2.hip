// hvx_hnsw_wave_gen_l2.hip -- GENERIC (any dimension / summation tree) builds of the one-wavefront-per-query kernel with the
// non-strict layer-0 arms, metric kL2: what serves `SearchParams::new(k)` on the shapes the unrolled builds do not.
#include "hvx_hnsw_wave.h"

namespace hvx {
hipError_t launch_hnsw_wave_gen_l2(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    return launch_wave_gen<kL2>(a, b, g, s);
}
} // namespace hvx
