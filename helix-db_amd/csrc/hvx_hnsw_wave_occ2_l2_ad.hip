// hvx_hnsw_wave_occ2_l2_ad.hip -- the NON-strict layer-0 arms (SimHash filter, pre / post sampling, adaptive bypass: what
// SearchParams::new(k) selects, crates/db/src/execution/interpreter/access/search/storage.rs:140-141) budgeted for TWO queries per
// SIMD: f32 rows, squared-Euclidean.  For hosts that keep several batches in flight (execution lanes, the batcher): the row
// gathers of one query run underneath the policy / RNG / beam bookkeeping of the other.
#include "hvx_hnsw_wave.h"

namespace hvx {
hipError_t launch_hnsw_wave_occ2_l2_ad(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    if (a.ix.dtype != HVX_F32 || !a.adaptive) return hipErrorInvalidValue;
    return a.ad.stats ? launch_wave_r<kL2, false, true, true, 2>(a, b, g, s) : launch_wave_r<kL2, false, true, false, 2>(a, b, g, s);
}
} // namespace hvx
