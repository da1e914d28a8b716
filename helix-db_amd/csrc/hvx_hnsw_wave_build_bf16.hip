// hvx_hnsw_wave_build_bf16.hip -- BUILD instantiations of the one-wavefront-per-query kernel over bf16 rows (round 6): the search side of
// a one-node insert / upsert into a bf16 image (config #4's storage).  The node's rounded vector is the f32 query (HnswArgs::queries), the
// rows it is compared with are the image's bf16 rows: f32 arithmetic on the rounded values in the reference's order, as every bf16 search.
#include "hvx_hnsw_wave.h"

namespace hvx {
hipError_t launch_hnsw_wave_build_bf16(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    if (a.ix.dtype != HVX_BF16 || a.adaptive || !a.build_nodes || !a.queries) return hipErrorInvalidValue;
    const uint32_t need = (a.ef > a.build_ef_upper ? a.ef : a.build_ef_upper) + 32u;
    if (a.ix.metric == kL2) {
        if (need <= 192) return launch_wave_nk<kL2, 3, true, false, true, 1, true>(a, b, g, s);
        if (need <= 384) return launch_wave_nk<kL2, 6, true, false, true, 1, true>(a, b, g, s);
    } else if (a.ix.metric == kCosine) {
        if (need <= 192) return launch_wave_nk<kCosine, 3, true, false, true, 1, true>(a, b, g, s);
        if (need <= 384) return launch_wave_nk<kCosine, 6, true, false, true, 1, true>(a, b, g, s);
    }
    return hipErrorInvalidValue;
}
} // namespace hvx
