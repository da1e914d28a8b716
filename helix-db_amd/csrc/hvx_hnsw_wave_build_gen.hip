// hvx_hnsw_wave_build_gen.hip -- BUILD instantiations of the GENERIC one-wavefront-per-query kernel (NK = 0: any dimension incl. the
// scalar tail, any metric -- Manhattan's sequential order too --, the AVX / AVX+FMA / scalar summation trees): the search side of
// a batched insert_hnsw (mutation.rs:787-895, 904-1005) for the shapes the unrolled builds of hvx_hnsw_wave_build.hip do not
// serve, and for ef_construction up to 800.  f32 rows; consumed by hvx_build.hip.
#include "hvx_hnsw_wave.h"

namespace hvx {
template <uint32_t METRIC> static hipError_t launch_build_gen(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    const uint32_t need = (a.ef > a.build_ef_upper ? a.ef : a.build_ef_upper) + 32u;
    if (need <= 192) return launch_wave_kernel(hnsw_wave_kernel<METRIC, 3, 0, false, false, false, true, 1, true>, a, b, g, s);
    if (need <= 448) return launch_wave_kernel(hnsw_wave_kernel<METRIC, 7, 0, false, false, false, true, 1, true>, a, b, g, s);
    if (need <= 832) return launch_wave_kernel(hnsw_wave_kernel<METRIC, 13, 0, false, false, false, true, 1, true>, a, b, g, s);
    return hipErrorInvalidValue;
}
hipError_t launch_hnsw_wave_build_gen(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    if (a.ix.dtype != HVX_F32 || a.adaptive || !a.build_nodes) return hipErrorInvalidValue;
    switch (a.ix.metric) {
    case kCosine: return launch_build_gen<kCosine>(a, b, g, s);
    case kL2: return launch_build_gen<kL2>(a, b, g, s);
    default: return launch_build_gen<kL1>(a, b, g, s);
    }
}
} // namespace hvx
