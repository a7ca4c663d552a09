// common.h -- shared host/device helpers for libclapgpu (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "clapgpu.h"

#define WAVE 64

namespace clapgpu {

// Records the HIP error text for clapgpu_last_error() and maps it to a cerr code.
int hip_fail(hipError_t err, const char *what);

#define CLAPGPU_HIP(call)                                                   \
    do {                                                                    \
        hipError_t err__ = (call);                                          \
        if (err__ != hipSuccess) return ::clapgpu::hip_fail(err__, #call);  \
    } while (0)

// Kernel launches return errors through hipGetLastError().
#define CLAPGPU_LAUNCH_CHECK(name)                                          \
    do {                                                                    \
        hipError_t err__ = hipGetLastError();                               \
        if (err__ != hipSuccess) return ::clapgpu::hip_fail(err__, name);   \
    } while (0)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ---------------------------------------------------------------- device side
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// Orders this wave's LDS writes before its later LDS reads of OTHER lanes' data.
// The waves of a block use disjoint LDS regions, so no s_barrier is needed: DS
// operations of one wave execute in issue order; the fence stops the compiler
// from hoisting the reads (it can prove a lane's own addresses never alias).
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Coalesced store of one mat4 per lane (64 consecutive matrices = 4 KiB) through a
// wave-private 4 KiB LDS tile: each global_store_dwordx4 writes 1 KiB contiguous
// instead of 64 x 16 B at a 64 B stride.  XOR swizzle keeps both the
// ds_write_b128 (8-lane groups) and ds_read_b128 (16-lane groups) conflict-free.
// All 64 lanes must call; `nvalid` = number of leading lanes whose matrix is stored.
__device__ __forceinline__ void wave_store_mat4(float4 *tile, float *dst, const float (&m)[16],
                                                int lane, int nvalid)
{
    const int sw = (lane >> 1) & 3;
#pragma unroll
    for (int c = 0; c < 4; c++)
        tile[lane * 4 + (c ^ sw)] = make_float4(m[4 * c], m[4 * c + 1], m[4 * c + 2], m[4 * c + 3]);
    wave_lds_fence();
    float4 *out = reinterpret_cast<float4 *>(dst);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int q = k * WAVE + lane;          // float4 index inside the wave's 4 KiB
        const int ent = q >> 2, col = q & 3;
        float4 v = tile[ent * 4 + (col ^ ((ent >> 1) & 3))];
        if (ent < nvalid)
            out[q] = v;
    }
    wave_lds_fence();                            // tile is reused by the caller
}

// Coalesced store of ROW floats per lane (ROW = 6: aabb, ROW = 3: aabb_center).
template <int ROW>
__device__ __forceinline__ void wave_store_rows(float *tile, float *dst, const float (&v)[ROW],
                                                int lane, int nvalid)
{
#pragma unroll
    for (int k = 0; k < ROW; k++)
        tile[lane * ROW + k] = v[k];
    wave_lds_fence();
    const int limit = nvalid * ROW;              // floats to store
    constexpr int CHUNKS = (WAVE * ROW + 3) / 4; // float4 chunks in the tile
#pragma unroll
    for (int k = 0; k < (CHUNKS + WAVE - 1) / WAVE; k++) {
        const int q = k * WAVE + lane;
        if (q < CHUNKS) {
            const int f0 = 4 * q;
            if (f0 + 4 <= limit) {
                reinterpret_cast<float4 *>(dst)[q] = reinterpret_cast<const float4 *>(tile)[q];
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (f0 + j < limit) dst[f0 + j] = tile[f0 + j];
            }
        }
    }
    wave_lds_fence();
}

} // namespace clapgpu
