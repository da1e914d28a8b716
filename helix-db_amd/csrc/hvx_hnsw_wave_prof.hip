// hvx_hnsw_wave_prof.hip -- phase-timing build of the wave kernel (L2, R=3, dim 768 only); launched
// instead of the production kernel when HVX_WAVE_PROF is set, for kernel tuning.
#include "hvx_hnsw_wave.h"

namespace hvx {
hipError_t launch_hnsw_wave_prof(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    if (a.ix.dtype != HVX_F32 || (a.ix.dim >> 5) != 24 || a.ef + 32u > 192u) return hipErrorInvalidValue;
#ifdef HVX_TUNING  // the non-strict arms, phase-timed (slot 6 = decision epoch + candidate selection)
    if (a.adaptive) return a.ix.metric == kL2 ? launch_wave_kernel(hnsw_wave_kernel<kL2, 3, 24, false, true, true, false>, a, b, g, s)
                                              : launch_wave_kernel(hnsw_wave_kernel<kCosine, 3, 24, false, true, true, false>, a, b, g, s);
#endif
    if (a.ix.metric != kL2) return hipErrorInvalidValue;
    return launch_wave_kernel(hnsw_wave_kernel<kL2, 3, 24, false, true>, a, b, g, s);
}
} // namespace hvx
