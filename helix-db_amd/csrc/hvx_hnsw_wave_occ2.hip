// hvx_hnsw_wave_occ2.hip -- the one-wavefront-per-query HNSW kernel budgeted for TWO wavefronts per SIMD (256 registers,
// 20 KiB of LDS each): the strict-exhaustive arm over f32 rows, squared-Euclidean and cosine.  Launched for callers that
// keep two or more batches in flight on one device (execution lanes, hvx_index_fork), where a second resident query per
// SIMD hides the HBM round trips of the first.
#include "hvx_hnsw_wave.h"

namespace hvx {
hipError_t launch_hnsw_wave_occ2(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    if (a.ix.dtype != HVX_F32 || a.adaptive) return hipErrorInvalidValue;
    return a.ix.metric == kL2 ? launch_wave_r<kL2, false, false, true, 2>(a, b, g, s) : launch_wave_r<kCosine, false, false, true, 2>(a, b, g, s);
}
} // namespace hvx
