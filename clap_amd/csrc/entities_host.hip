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
    __shared__ float4 lds_tiles[ENT_BLOCK / WAVE][LDS_F4_PER_WAVE];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const uint32_t t = blockIdx.x * (ENT_BLOCK / WAVE) + wave;
    const uint32_t n_rows = (n + WAVE - 1) / WAVE;
    uint32_t row = 0, row_end = 0;
    if (t < n_tiles) {
        row = tile_row_start[t];
        row_end = tile_row_start[t + 1];
        if (row_end > n_rows) row_end = n_rows;
    }
    if (row < row_end) {
        float carry_mx[16];
#pragma unroll
        for (int k = 0; k < 16; k++) carry_mx[k] = 0.f;
        uint32_t carry_seq = 0;
        bool carry_valid = false, have_prev = false;
        uint32_t chunk = row;                                    // lane l holds the touched word of row chunk + l
        unsigned long long tw = (h.touched && chunk + lane < row_end) ? h.touched[chunk + lane] : 0ull;
        auto touched_of = [&](uint32_t r) -> uint64_t {          // r is wave-uniform and >= chunk
            if (r - chunk >= (uint32_t)WAVE) {
                chunk = r;
                tw = (h.touched && chunk + lane < row_end) ? h.touched[chunk + lane] : 0ull;
            }
            return (uint64_t)__shfl(tw, (int)(r - chunk));
        };
        // Three rows in the pipe: row r is worked on while row r + 1's second-level loads (what depends on its parent and
        // model indices) and row r + 2's first-level loads are in flight.  Past the tile's end the last row is re-read.
        auto first_of = [&](uint32_t r) { return (r < row_end ? r : row_end - 1) * (uint32_t)WAVE; };
        auto count_of = [&](uint32_t f) { return n - f < (uint32_t)WAVE ? n - f : (uint32_t)WAVE; };
        uint32_t f0 = first_of(row), f1 = first_of(row + 1);
        uint64_t t0 = touched_of(row), t1 = row + 1 < row_end ? touched_of(row + 1) : 0ull;
        RowPre b0, b1;
        RowIn a0 = load_row_host(e, h, lane, f0, count_of(f0), t0);
        load_row_box(e, b0, lane, f0, count_of(f0));
        RowIn a1 = load_row_host(e, h, lane, f1, count_of(f1), t1);
        load_row_box(e, b1, lane, f1, count_of(f1));
        load_row_pre(e, b0, a0);
        for (;;) {
            const bool more = row + 1 < row_end;
            const uint32_t f2 = first_of(row + 2);
            const uint64_t t2 = row + 2 < row_end ? touched_of(row + 2) : 0ull;
            load_row_pre(e, b1, a1);                             // waits for a1, asked for one row ago
            RowPre b2;
            const RowIn a2 = load_row_host(e, h, lane, f2, count_of(f2), t2);
            load_row_box(e, b2, lane, f2, count_of(f2));
            const uint32_t row_first = f0, row_count = count_of(f0);
            if ((uint32_t)lane < row_count && ((t0 >> lane) & 1ull)) {           // the device copy of what the host wrote
                const uint32_t i = row_first + lane;
                const_cast<float4 *>(e.pos_scale)[i] = a0.ps;
                const_cast<float4 *>(e.rot)[i] = a0.q;
                e.flags[i] = a0.fl;                              // process_row clears DIRTY behind this, same lane, program order
            }
            process_row<CULL, true, true>(e, a0, lds_tiles[wave], lane, row_first, row_count, mode, fr, have_prev,
                                          row_first - WAVE, carry_mx, carry_seq, carry_valid, &h, &b0);
            if (!more)
                break;
            have_prev = true;
            row++;
            a0 = a1; a1 = a2;
            b0 = b1;
            b1.bb0 = b2.bb0; b1.bb1 = b2.bb1; b1.bb2 = b2.bb2;    // its second level is asked for at the top
            f0 = f1; f1 = f2;
            t0 = t1; t1 = t2;
        }
    }
    // completion: every workgroup releases its stores to the system, the last one to arrive raises the word
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t arrived = atomicAdd(h.counter, 1u);
        if (arrived == gridDim.x - 1) {
            *h.counter = 0;                                       // ready for the next launch (stream order)
            __threadfence_system();
            __hip_atomic_store(h.done, h.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

int launch_entities_tiles_host(hipStream_t stream, bool cull, const lmd::FrustumK &fr, const EntK &e, const HostIO &h,
                               const uint32_t *tile_row_start, uint32_t n_tiles, uint32_t n, uint32_t mode)
{
    const uint32_t per_block = ENT_BLOCK / WAVE;
    const dim3 grid(n_tiles ? (n_tiles + per_block - 1) / per_block : 1), block(ENT_BLOCK);   // an empty scene still raises the word
    if (cull)
        hipLaunchKernelGGL(k_entities_tiles_host<true>, grid, block, 0, stream, fr, e, h, tile_row_start, n_tiles, n, mode);
    else
        hipLaunchKernelGGL(k_entities_tiles_host<false>, grid, block, 0, stream, fr, e, h, tile_row_start, n_tiles, n, mode);
    CLAPGPU_LAUNCH_CHECK("k_entities_tiles_host");
    return CLAPGPU_OK;
}

} // namespace clapgpu
