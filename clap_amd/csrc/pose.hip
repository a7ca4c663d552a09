// pose.hip -- skeletal pose blend + joint-matrix palette for gfx950.
//
// Replaces, per animated entity, channels_transform() (model.c:1266-1350: keyframe bracket,
// lerp T/S, slerp R; interp.h:59-118) and one_joint_transform() (model.c:1352-1404: global
// chain, joint_transforms = global * invmx, joint world position).  The host keeps
// animated_update()'s time base and queue logic (model.c:1563-1592) and passes each
// character's animation id and (float)frame_time.
//
// Mapping: one lane per joint, a 64-joint skeleton = one wavefront = one character; the
// joint tree is walked level by level with the parents' globals in LDS (they never go to HBM,
// as in the reference where `global` is scratch).  Keyframes are per model and stay in L2.
// HBM: ~200 B / joint: T/R/S 40 B, joint_transforms 64 B, joint pos 16 B written.
//
// Numerics: same mixed precision as the reference (double lerp, double acos/sin/cos in slerp);
// device libm differs from glibc in the last ulp of a double, so this path is held to 1e-5
// relative, not bit-exact.
#include <string.h>
#include "common.h"
#include "lm_dev.h"

namespace clapgpu {

struct PoseArgs {
    // skeleton
    uint32_t        J, n_levels;
    const int32_t  *parent;
    const int32_t  *depth;
    const float    *root_pose;
    const float4   *invmx;
    const float4   *bind;
    // animations
    const int32_t  *chan_of;
    const uint32_t *ch_nr, *ch_time_off, *ch_data_off;
    const float    *times, *data;
    // batch
    uint32_t        n_chars;
    const uint32_t *anim;
    const float    *frame_time;
    const uint32_t *entity;
    const float    *entity_mx;
    float          *trs;
    float          *joint_transforms;
    float          *joint_pos;
};

// model.c:1266-1288 for strictly increasing key times (glTF): the bracket does not depend on
// the reference's search cursor, so a binary search gives the same (prev, next).
__device__ __forceinline__ void key_bracket(const float *t, int nr, float time, int &prev, int &next)
{
    if (time < t[0] || time > t[nr - 1]) {      // before the first / past the last key: wrap
        prev = nr - 1;
        next = 0;
        return;
    }
    int lo = 0, hi = nr - 1;                    // first i with time <= t[i]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (time <= t[mid]) hi = mid; else lo = mid + 1;
    }
    prev = lo > 0 ? lo - 1 : 0;
    next = prev + 1 < nr - 1 ? prev + 1 : nr - 1;
}

// model.c:1312-1317
__device__ __forceinline__ float key_fac(float time, float p_time, float n_time)
{
    if (p_time > n_time) return time < n_time ? 1.f : 0.f;
    if (p_time < n_time) return (time - p_time) / (n_time - p_time);
    return 0.f;
}

// interp.h:25-29: a * (1.0 - blend) + b * blend with float operands (b * blend is an fp32 product)
__device__ __forceinline__ float lerp_ref(float a, float b, float fac)
{
    const float bf = b * fac;
    return (float)((double)a * (1.0 - (double)fac) + (double)bf);
}

// interp.h:67-118 quat_slerp / quat_interp
__device__ __forceinline__ void slerp_ref(float (&res)[4], const float (&a)[4], const float (&b_in)[4], float fac)
{
    float b[4] = { b_in[0], b_in[1], b_in[2], b_in[3] };
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) dot += b[i] * a[i];
    if (dot < 0.0f) {
        dot = -dot;
#pragma unroll
        for (int i = 0; i < 4; i++) b[i] = -b[i];
    }
    if ((double)dot > 0.9995) {                                  // nlerp + vec4_norm
        const float rfac = 1.f - fac;
        float t[4], d2 = 0.f, dd = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) d2 += b[i] * a[i];           // quat_interp recomputes the dot
#pragma unroll
        for (int i = 0; i < 4; i++)
            t[i] = d2 < 0 ? rfac * a[i] - fac * b[i] : rfac * a[i] + fac * b[i];
#pragma unroll
        for (int i = 0; i < 4; i++) dd += t[i] * t[i];
        const float k = (float)(1.0 / (double)sqrtf(dd));
#pragma unroll
        for (int i = 0; i < 4; i++) res[i] = t[i] * k;
        return;
    }
    const float theta_0 = (float)acos((double)dot);
    const float theta = fac * theta_0;
    const float sin_theta = (float)sin((double)theta);
    const float sin_theta_0 = (float)sin((double)theta_0);
    const float rf = (float)(cos((double)theta) - (double)(dot * sin_theta / sin_theta_0));
    const float f = sin_theta / sin_theta_0;
#pragma unroll
    for (int i = 0; i < 4; i++) res[i] = a[i] * rf + b[i] * f;
}

constexpr int POSE_BLOCK = 256;
constexpr int G_STRIDE = 16;            // floats per joint global in LDS

// LPC = lanes per character (64, 128, 192 or 256); CPB = characters per block.
template <int LPC>
__global__ __launch_bounds__(POSE_BLOCK)
void k_pose(PoseArgs a)
{
    constexpr int CPB = POSE_BLOCK / LPC;
    __shared__ float g_lds[CPB][LPC * G_STRIDE];                                   // joint globals
    __shared__ float4 stage[POSE_BLOCK / WAVE][256];                               // 4 KiB per wave

    const int tid = threadIdx.x;
    const int cib = tid / LPC, j = tid % LPC;
    const int lane = lane_id(), wave = tid / WAVE;
    const uint32_t c = blockIdx.x * CPB + cib;
    const bool char_ok = c < a.n_chars && cib < CPB;
    const uint32_t J = a.J;
    const bool joint_ok = char_ok && (uint32_t)j < J;
    const int depth = joint_ok ? a.depth[j] : -1;
    const bool reachable = depth >= 0;
    float *G = g_lds[cib < CPB ? cib : 0];

    // ---- 1. channels_transform: this joint's T, R, S at the character's frame time ----
    float T[3] = { 0, 0, 0 }, R[4] = { 0, 0, 0, 1 }, S[3] = { 1, 1, 1 };
    const size_t cj = (size_t)c * J + j;
    if (joint_ok) {
        const uint32_t an = a.anim[c];
        const float time = a.frame_time[c];
        const int32_t *co = a.chan_of + ((size_t)an * J + j) * 3;
        const int32_t c0 = co[0], c1 = co[1], c2 = co[2];
        if (c0 < 0 || c1 < 0 || c2 < 0) {                        // a path without a channel keeps its value
            const float *st = a.trs + 10 * cj;
            T[0] = st[0]; T[1] = st[1]; T[2] = st[2];
            R[0] = st[3]; R[1] = st[4]; R[2] = st[5]; R[3] = st[6];
            S[0] = st[7]; S[1] = st[8]; S[2] = st[9];
        }
        if (c0 >= 0) {
            const float *t = a.times + a.ch_time_off[c0];
            const float *d = a.data + a.ch_data_off[c0];
            int p, n;
            key_bracket(t, (int)a.ch_nr[c0], time, p, n);
            const float fac = key_fac(time, t[p], t[n]);
#pragma unroll
            for (int k = 0; k < 3; k++) T[k] = lerp_ref(d[3 * p + k], d[3 * n + k], fac);
        }
        if (c1 >= 0) {
            const float *t = a.times + a.ch_time_off[c1];
            const float *d = a.data + a.ch_data_off[c1];
            int p, n;
            key_bracket(t, (int)a.ch_nr[c1], time, p, n);
            const float fac = key_fac(time, t[p], t[n]);
            const float qa[4] = { d[4 * p], d[4 * p + 1], d[4 * p + 2], d[4 * p + 3] };
            const float qb[4] = { d[4 * n], d[4 * n + 1], d[4 * n + 2], d[4 * n + 3] };
            slerp_ref(R, qa, qb, fac);
        }
        if (c2 >= 0) {
            const float *t = a.times + a.ch_time_off[c2];
            const float *d = a.data + a.ch_data_off[c2];
            int p, n;
            key_bracket(t, (int)a.ch_nr[c2], time, p, n);
            const float fac = key_fac(time, t[p], t[n]);
#pragma unroll
            for (int k = 0; k < 3; k++) S[k] = lerp_ref(d[3 * p + k], d[3 * n + k], fac);
        }
    }

    // ---- 2. one_joint_transform, level by level: global = ((parent * I) * T) * R, scale_aniso ----
    // T is a pure translation and R a pure rotation matrix, so the products are evaluated on their
    // non-trivial terms only (the dropped terms are exact +-0 in the reference's full 4x4 products).
    float Gm[16];
#pragma unroll
    for (int k = 0; k < 16; k++) Gm[k] = 0.f;
    float Rm[16];
    lmd::from_quat(Rm, R[0], R[1], R[2], R[3]);
    const int32_t parent = joint_ok ? a.parent[j] : -1;
    for (uint32_t d = 0; d < a.n_levels; d++) {
        if (reachable && (uint32_t)depth == d) {
            float P[16];
            if (parent >= 0) {
                const float4 *src = reinterpret_cast<const float4 *>(G + parent * G_STRIDE);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 v = src[q];
                    P[4 * q] = v.x; P[4 * q + 1] = v.y; P[4 * q + 2] = v.z; P[4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) P[k] = a.root_pose[k];
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float t3 = 0.f;                                   // column 3 of P * T
                t3 += E_(P, 0, r) * T[0];
                t3 += E_(P, 1, r) * T[1];
                t3 += E_(P, 2, r) * T[2];
                t3 += E_(P, 3, r) * 1.f;
                E_(Gm, 3, r) = t3;
#pragma unroll
                for (int cc = 0; cc < 3; cc++) {                  // columns 0..2 of (P * T) * R, then scale
                    float s = 0.f;
                    s += E_(P, 0, r) * E_(Rm, cc, 0);
                    s += E_(P, 1, r) * E_(Rm, cc, 1);
                    s += E_(P, 2, r) * E_(Rm, cc, 2);
                    E_(Gm, cc, r) = s * S[cc];
                }
            }
            float4 *dst = reinterpret_cast<float4 *>(G + j * G_STRIDE);
#pragma unroll
            for (int q = 0; q < 4; q++)
                dst[q] = make_float4(Gm[4 * q], Gm[4 * q + 1], Gm[4 * q + 2], Gm[4 * q + 3]);
        }
        if (LPC == WAVE) wave_lds_fence(); else __syncthreads();
    }

    // ---- 3. palette: joint_transforms = global * invmx; pos = e->mx * (joint_transforms * bind) * (0,0,0,1) ----
    float JT[16], pos[4] = { 0, 0, 0, 0 };
    if (reachable) {
        float IM[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = a.invmx[4 * j + q];
            IM[4 * q] = v.x; IM[4 * q + 1] = v.y; IM[4 * q + 2] = v.z; IM[4 * q + 3] = v.w;
        }
        lmd::mul(JT, Gm, IM);                                     // model.c:1389
        const float4 b3 = a.bind[4 * j + 3];                      // only column 3 of bind reaches mpos
        const float bv[4] = { b3.x, b3.y, b3.z, b3.w };
        float mpos[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {                             // column 3 of JT * bind (model.c:1393-1397)
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) s += E_(JT, k, r) * bv[k];
            mpos[r] = s;
        }
        const uint32_t ei = a.entity ? a.entity[c] : c;
        float EM[16];
        const float4 *em = reinterpret_cast<const float4 *>(a.entity_mx + 16 * (size_t)ei);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = em[q];
            EM[4 * q] = v.x; EM[4 * q + 1] = v.y; EM[4 * q + 2] = v.z; EM[4 * q + 3] = v.w;
        }
        lmd::mul_vec4(pos, EM, mpos);                             // model.c:1400
    }

    // ---- stores (64 joints of one character per wave row) ----
    if (!char_ok || (uint32_t)(j - lane) >= J)
        return;                                                   // whole wave has nothing to store
    const uint32_t row_j0 = j - lane;                             // first joint of this wave's row
    const int nvalid = (int)(J - row_j0 < WAVE ? J - row_j0 : WAVE);
    const size_t row0 = (size_t)c * J + row_j0;
    float4 *tile = stage[wave];
    float *tile_f = reinterpret_cast<float *>(tile);

    const float trs_row[10] = { T[0], T[1], T[2], R[0], R[1], R[2], R[3], S[0], S[1], S[2] };
    stage_rows<10>(tile_f, trs_row, lane);                        // 2560 B
    wave_lds_fence();
    store_rows<10>(tile_f, a.trs + 10 * row0, lane, nvalid);
    wave_lds_fence();

    const uint64_t reach_mask = __ballot(reachable);
    const uint64_t full = nvalid == WAVE ? ~0ull : ((1ull << nvalid) - 1ull);
    if (reach_mask == full) {
        float4 v[4];
        stage_mat4(tile, JT, lane);
        wave_lds_fence();
        unstage_mat4(tile, v, lane);
        store_mat4_rows(a.joint_transforms + 16 * row0, v, lane, nvalid);
        if (lane < nvalid)
            reinterpret_cast<float4 *>(a.joint_pos)[row0 + lane] = make_float4(pos[0], pos[1], pos[2], pos[3]);
    } else if (reachable) {                                       // joints not under joint 0 are never written
        float4 *dj = reinterpret_cast<float4 *>(a.joint_transforms + 16 * cj);
#pragma unroll
        for (int q = 0; q < 4; q++)
            dj[q] = make_float4(JT[4 * q], JT[4 * q + 1], JT[4 * q + 2], JT[4 * q + 3]);
        reinterpret_cast<float4 *>(a.joint_pos)[cj] = make_float4(pos[0], pos[1], pos[2], pos[3]);
    }
}

} // namespace clapgpu

using namespace clapgpu;

extern "C" int clapgpu_pose_update(void *stream, const clapgpu_skeleton *sk, const clapgpu_animations *an,
                                   const clapgpu_pose_batch *pb)
{
    if (!sk || !an || !pb)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!sk->parent || !sk->depth || !sk->root_pose || !sk->invmx || !sk->bind || !an->chan_of ||
        !an->ch_nr || !an->ch_time_off || !an->ch_data_off || !an->times || !an->data)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (pb->n_chars == 0)
        return CLAPGPU_OK;
    if (!pb->anim || !pb->frame_time || !pb->entity_mx || !pb->trs || !pb->joint_transforms || !pb->joint_pos)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (sk->nr_joints == 0 || sk->nr_joints > 256)               // JOINTS_MAX is 200 (shader_constants.h:6)
        return CLAPGPU_ERR_TOO_LARGE;

    PoseArgs a;
    a.J = sk->nr_joints;
    a.n_levels = sk->n_levels;
    a.parent = sk->parent;
    a.depth = sk->depth;
    a.root_pose = sk->root_pose;
    a.invmx = reinterpret_cast<const float4 *>(sk->invmx);
    a.bind = reinterpret_cast<const float4 *>(sk->bind);
    a.chan_of = an->chan_of;
    a.ch_nr = an->ch_nr;
    a.ch_time_off = an->ch_time_off;
    a.ch_data_off = an->ch_data_off;
    a.times = an->times;
    a.data = an->data;
    a.n_chars = pb->n_chars;
    a.anim = pb->anim;
    a.frame_time = pb->frame_time;
    a.entity = pb->entity;
    a.entity_mx = pb->entity_mx;
    a.trs = pb->trs;
    a.joint_transforms = pb->joint_transforms;
    a.joint_pos = pb->joint_pos;

    const uint32_t lpc = (sk->nr_joints + 63) / 64 * 64;
    hipStream_t s = as_stream(stream);
    switch (lpc) {
    case 64:  hipLaunchKernelGGL(k_pose<64>,  dim3((pb->n_chars + 3) / 4), dim3(256), 0, s, a); break;
    case 128: hipLaunchKernelGGL(k_pose<128>, dim3((pb->n_chars + 1) / 2), dim3(256), 0, s, a); break;
    case 192: hipLaunchKernelGGL(k_pose<192>, dim3(pb->n_chars), dim3(192), 0, s, a); break;
    default:  hipLaunchKernelGGL(k_pose<256>, dim3(pb->n_chars), dim3(256), 0, s, a); break;
    }
    CLAPGPU_LAUNCH_CHECK("k_pose");
    return CLAPGPU_OK;
}
