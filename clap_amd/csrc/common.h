// common.h -- shared host/device helpers for libclapgpu (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "clapgpu.h"

#define WAVE 64

namespace clapgpu {

// Records the HIP error text for clapgpu_last_error() and maps it to a cerr code.
int hip_fail(hipError_t err, const char *what);
// ... and a refusal of our own (an asset or an argument combination the exactness does not cover)
void set_last_error(const char *what);

#define CLAPGPU_HIP(call)                                                   \
    do {                                                                    \
        hipError_t err__ = (call);                                          \
        if (err__ != hipSuccess) return ::clapgpu::hip_fail(err__, #call);  \
    } while (0)

// Kernel launches return errors through hipGetLastError() -- or, for the callers' failure-path tests, through
// clapgpu_test_fail_after() (runtime.hip).
hipError_t launch_error();
#define CLAPGPU_LAUNCH_CHECK(name)                                          \
    do {                                                                    \
        hipError_t err__ = ::clapgpu::launch_error();                       \
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

// Wave-private LDS staging for coalesced output: each lane holds one row (a mat4, an
// AABB, ...) of 64 consecutive entities; the rows are written to LDS lane-major and
// read back 16 B per lane in memory order, so every global_store_dwordx4 writes 1 KiB
// contiguous instead of 64 pieces at the row stride.  DS operations of one wave
// execute in issue order, so reusing a region needs only the compiler-level fence.
//
// mat4: the XOR swizzle keeps ds_write_b128 (8-lane groups) and ds_read_b128
// (16-lane groups) conflict-free.  All 64 lanes must call these.
__device__ __forceinline__ void stage_mat4(float4 *tile, const float (&m)[16], int lane)
{
    const int sw = (lane >> 1) & 3;
#pragma unroll
    for (int c = 0; c < 4; c++)
        tile[lane * 4 + (c ^ sw)] = make_float4(m[4 * c], m[4 * c + 1], m[4 * c + 2], m[4 * c + 3]);
}

__device__ __forceinline__ void unstage_mat4(const float4 *tile, float4 (&v)[4], int lane)
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int q = k * WAVE + lane;          // float4 index inside the wave's 4 KiB
        const int ent = q >> 2, col = q & 3;
        v[k] = tile[ent * 4 + (col ^ ((ent >> 1) & 3))];
    }
}

// ---- experiment switches --------------------------------------------------------------------------------------------
// A few A/B and sensitivity builds (tools/entities_sensitivity.sh, tools/profile_entities_scale.sh) compile parts of the
// kernels out or change their store policy; some of them produce WRONG results on purpose.  None of them can reach a
// release build by a stray EXTRA= flag: every such macro requires -DCLAPGPU_EXPERIMENT, and an experiment build
// reports an ABI version with the top bit set (runtime.hip), which clap_amd/_lib.py and any caller checking
// clapgpu_abi_version() against its header refuse to load.
#if defined(CLAPGPU_EXP_NO_INVERT) || defined(CLAPGPU_EXP_NO_AABB) || defined(CLAPGPU_PLAIN_STORES) || defined(BP_SEARCH_IN_FLIGHT)
#  ifndef CLAPGPU_EXPERIMENT
#    error "CLAPGPU_EXP_* / CLAPGPU_PLAIN_STORES / BP_SEARCH_IN_FLIGHT are experiment switches: add -DCLAPGPU_EXPERIMENT (the library then reports an experiment ABI version)"
#  endif
#endif

// Streaming store: the big per-frame outputs (mx, inverse_mx, aabb, palettes) are written once and
// read by a later kernel or the host, never by the writer.  Marked non-temporal they do not push the
// inputs out of the 256 MB infinity cache: neutral at 1 M entities (everything fits), 112 -> 78 us at
// 2 M entities, where the outputs alone are 330 MB.
typedef float clapgpu_f4 __attribute__((ext_vector_type(4)));
typedef float clapgpu_f2 __attribute__((ext_vector_type(2)));
typedef float clapgpu_f3 __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ void store_stream(float4 *dst, const float4 &v)
{
#ifdef CLAPGPU_PLAIN_STORES          // A/B builds only (tools/profile_entities_scale.sh): default-policy stores
    *dst = v;
#else
    const clapgpu_f4 t = { v.x, v.y, v.z, v.w };
    __builtin_nontemporal_store(t, reinterpret_cast<clapgpu_f4 *>(dst));
#endif
}

// dst = first matrix of the wave's 64; nvalid = leading lanes whose matrix is stored
template <bool STREAM = true>
__device__ __forceinline__ void store_mat4_rows(float *dst, const float4 (&v)[4], int lane, int nvalid)
{
    float4 *out = reinterpret_cast<float4 *>(dst);
    if (nvalid == WAVE) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (STREAM) store_stream(&out[k * WAVE + lane], v[k]);
            else out[k * WAVE + lane] = v[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int q = k * WAVE + lane;
            if ((q >> 2) < nvalid) out[q] = v[k];
        }
    }
}

// ROW floats per lane (ROW = 6: aabb, ROW = 3: aabb_center), tile region of 64*ROW floats.
template <int ROW>
__device__ __forceinline__ void stage_rows(float *tile, const float (&v)[ROW], int lane)
{
#pragma unroll
    for (int k = 0; k < ROW; k++)
        tile[lane * ROW + k] = v[k];
}

template <int ROW>
__device__ __forceinline__ void store_rows(const float *tile, float *dst, int lane, int nvalid)
{
    const int limit = nvalid * ROW;              // floats to store
    constexpr int CHUNKS = (WAVE * ROW + 3) / 4; // float4 chunks in the tile region
#pragma unroll
    for (int k = 0; k < (CHUNKS + WAVE - 1) / WAVE; k++) {
        const int q = k * WAVE + lane;
        if (q < CHUNKS) {
            const int f0 = 4 * q;
            if (f0 + 4 <= limit) {
                store_stream(&reinterpret_cast<float4 *>(dst)[q], reinterpret_cast<const float4 *>(tile)[q]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (f0 + j < limit) dst[f0 + j] = tile[f0 + j];
            }
        }
    }
}

// Sum of bytes [0, nbytes) of a 16-byte aligned array, nbytes % 16 == 0, over the whole wave.
// Eight independent 16-byte loads per lane are in flight per trip: the callers sit on the
// critical path right behind a kernel boundary and would otherwise serialise L2 round trips.
__device__ __forceinline__ uint32_t wave_byte_sum(const uint8_t *bytes, uint32_t nbytes, int lane)
{
    const uint4 *p = reinterpret_cast<const uint4 *>(bytes);
    const uint32_t chunks = nbytes >> 4;
    uint32_t acc = 0;
    for (uint32_t c = lane; c < chunks; c += WAVE * 8) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t idx = c + u * WAVE;
            v[u] = idx < chunks ? p[idx] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc = __builtin_amdgcn_sad_u8(v[u].x, 0u, acc);
            acc = __builtin_amdgcn_sad_u8(v[u].y, 0u, acc);
            acc = __builtin_amdgcn_sad_u8(v[u].z, 0u, acc);
            acc = __builtin_amdgcn_sad_u8(v[u].w, 0u, acc);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        acc += __shfl_xor(acc, off);
    return acc;
}

} // namespace clapgpu
