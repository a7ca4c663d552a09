// entities_host.hip -- a host mirror's small frame as one launch (clapgpu_entities_update_tiles_hostio).
// Its own translation unit: see entities_row.h.
#include "entities_row.h"

namespace clapgpu {

// The three launches above as one, for the tile layout: a frame of a testbed-sized scene spent 50 of its 68 us waiting
// on three dependent launches of a few microseconds each.  Wave t walks tile t like k_entities_tiles' general loop; a
// lane the host touched this frame takes (flags, TRS) from the mirror's mapped upload image (and stores them into the
// device arrays, where later frames read them), every rebuilt lane also stores its results into the mapped result
// arrays, and the last workgroup to finish raises the completion word.  The touched bits of a tile's rows are read 64
// rows at a time (one trip over PCIe per tile, not per row); the touched lanes' inputs are in flight during the row before.
template <bool CULL>
__global__ __launch_bounds__(ENT_BLOCK)
void k_entities_tiles_host(lmd::FrustumK fr, EntK e, HostIO h, const uint32_t *tile_row_start, uint32_t n_tiles, uint32_t n,
                           uint32_t mode)
{
#define TILES_XV 0
#define TILES_XV_PTR nullptr
#include "entities_tiles_host_body.inc"
#undef TILES_XV
#undef TILES_XV_PTR
}

// ... with the frame's further views (clapgpu_entities.views)
__global__ __launch_bounds__(ENT_BLOCK)
void k_entities_tiles_host_xv(lmd::FrustumK fr, EntK e, HostIO h, const uint32_t *tile_row_start, uint32_t n_tiles, uint32_t n,
                              uint32_t mode, XViewsK xv)
{
    constexpr bool CULL = true;
#define TILES_XV 1
#define TILES_XV_PTR (&xv)
#include "entities_tiles_host_body.inc"
#undef TILES_XV
#undef TILES_XV_PTR
}

int launch_entities_tiles_host(hipStream_t stream, bool cull, const lmd::FrustumK &fr, const EntK &e, const HostIO &h,
                               const uint32_t *tile_row_start, uint32_t n_tiles, uint32_t n, uint32_t mode, const XViewsK &xv)
{
    const uint32_t per_block = ENT_BLOCK / WAVE;
    const dim3 grid(n_tiles ? (n_tiles + per_block - 1) / per_block : 1), block(ENT_BLOCK);   // an empty scene still raises the word
    if (cull && xv.n)
        hipLaunchKernelGGL(k_entities_tiles_host_xv, grid, block, 0, stream, fr, e, h, tile_row_start, n_tiles, n, mode, xv);
    else if (cull)
        hipLaunchKernelGGL(k_entities_tiles_host<true>, grid, block, 0, stream, fr, e, h, tile_row_start, n_tiles, n, mode);
    else
        hipLaunchKernelGGL(k_entities_tiles_host<false>, grid, block, 0, stream, fr, e, h, tile_row_start, n_tiles, n, mode);
    CLAPGPU_LAUNCH_CHECK("k_entities_tiles_host");
    return CLAPGPU_OK;
}

} // namespace clapgpu
