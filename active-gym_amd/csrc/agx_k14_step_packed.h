// agx_k14_step_packed.h - the first launch of agx_step_flexible_packed: one whole step of a FlexibleFovealEnv batch in raw-crop
// mode with packed ragged observations (atari_env.py:119-148 + fov_env.py:300-330, 283-298) in TWO launches instead of three.
//
// The packed layout needs every env's NEW resolution before any crop can be placed (offsets = exclusive scan of fs * rh * rw), so
// agx_fovea_flexible_packed runs the state update + scan as a launch of its own: 4.9 us at N = 1024, nearly all of it launch latency
// and one dependent chain.  That launch reads the actions and the old fov state and nothing the ingest writes - and the ingest reads
// nothing it writes - so here its ceil(N / 256) workgroups ride in the ingest launch: the first rows of the grid are scan blocks
// (dispatched first, finished long before the bands are), the rest is the band12 ingest, env index shifted by the scan rows.
// Folding the scan into the CROP launch was measured slower (docs/HISTORY.md, round 3: every crop workgroup redoing the scan);
// this form does the scan once.
#pragma once
#include "agx_k1_ingest.h"
#include "agx_k4_raw3.h"

namespace agx {

// grid = (nbands, nb + N), nb = ceil(N / kScanEnvsPerBlock); dynamic LDS = band12_lds (>= 32 B, which is all a scan block uses)
template <bool GRAY, bool COMPACT>
__global__ __launch_bounds__(kThreads) void k_ingest_full12_flexscan(IngestParams p, FlexScanParams q, int nb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int y = blockIdx.y;
    if (y < nb) {                                            // wave-uniform
        if (blockIdx.x == 0) flex_state_scan_block(q, y, reinterpret_cast<int64_t *>(smem));
        return;
    }
    ingest_band12<GRAY, COMPACT>(p, blockIdx.x, y - nb, smem, (int)threadIdx.x);
}

}  // namespace agx
