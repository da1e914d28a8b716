// hvx_hnsw_wave_build_occ2.hip -- BUILD instantiations of the one-wavefront-per-query kernel budgeted for TWO wavefronts per SIMD
// (256 registers, 20 KiB of LDS each): the search side of a batched insert_hnsw (mutation.rs:787-895, 904-1005) for batches of
// more than 1 024 nodes, where the one-per-SIMD build would run the batch in two rounds.  Same algorithm, same results (a visited
// table that fills spills to the exact bitmap); consumed by hvx_build.hip.
#include "hvx_hnsw_wave.h"

namespace hvx {
hipError_t launch_hnsw_wave_build_occ2(const HnswArgs &a, uint32_t b, const WaveGeom &g, hipStream_t s) {
    if (a.ix.dtype != HVX_F32 || a.adaptive || !a.build_nodes) return hipErrorInvalidValue;
    const uint32_t need = (a.ef > a.build_ef_upper ? a.ef : a.build_ef_upper) + 32u;
    if (a.ix.metric == kL2) {
        if (need <= 192) return launch_wave_nk<kL2, 3, false, false, true, 2, true>(a, b, g, s);
        if (need <= 384) return launch_wave_nk<kL2, 6, false, false, true, 2, true>(a, b, g, s);
    } else if (a.ix.metric == kCosine) {
        if (need <= 192) return launch_wave_nk<kCosine, 3, false, false, true, 2, true>(a, b, g, s);
        if (need <= 384) return launch_wave_nk<kCosine, 6, false, false, true, 2, true>(a, b, g, s);
    }
    return hipErrorInvalidValue;
}
} // namespace hvx
