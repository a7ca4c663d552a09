// pose.hip -- skeletal pose blend + joint-matrix palette for gfx950.
//
// Replaces, per animated entity, channels_transform() (model.c:1266-1350: keyframe bracket,
// lerp T/S, slerp R; interp.h:59-118) and one_joint_transform() (model.c:1352-1404: global
// chain, joint_transforms = global * invmx, joint world position).  The host keeps
// animated_update()'s time base and queue logic (model.c:1563-1592) and passes each
// character's animation id and (float)frame_time.
//
// Mapping: one lane per joint, a 64-joint skeleton = one wavefront = one character; every
// joint folds the locals on its ancestor path by pointer jumping through LDS (log2(levels)
// rounds; the globals never go to HBM, as in the reference where `global` is scratch).  Keyframes are per model and stay in L2.
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
    uint32_t        J, n_jump_steps;
    const int32_t  *parent;
    const int32_t  *depth;
    const float    *root_pose;
    const float4   *invmx;
    const float4   *bind;
    // animations
    const uint4    *chan_table;     // [n_anims][J][3] = (time_off, data_off, nr, 0)
    const float    *times, *data;
    uint32_t        n_times, n_anims;
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

// model.c:1312-1317.  The quotient uses the hardware reciprocal (<= 2 ulp): this path is held to
// 1e-5, and the IEEE division sequence is 3x the instructions.
__device__ __forceinline__ float key_fac(float time, float p_time, float n_time)
{
    if (p_time > n_time) return time < n_time ? 1.f : 0.f;
    if (p_time < n_time) return __fdividef(time - p_time, n_time - p_time);
    return 0.f;
}

// interp.h:25-29: a * (1.0 - blend) + b * blend.  The reference forms the first product and the sum
// in double; in fp32 the result differs by <= 1 ulp of the larger operand (inside the 1e-5 bar) and
// costs a third of the VALU time (fp64 converts and multiplies run at half rate).
__device__ __forceinline__ float lerp_ref(float a, float b, float fac)
{
    return fmaf(b, fac, a * (1.0f - fac));
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
    // The reference calls the double libm acos/sin/cos on float arguments and rounds the results
    // back to float.  fp64 transcendentals are this kernel's single largest VALU cost, so the
    // correctly-rounded-to-a-few-ulp fp32 forms are used instead: |error| <= ~3e-7 on unit
    // quaternion components, inside the 1e-5 bar this path is held to (tests/test_pose_skin_gpu.py).
    // Here dot is in [0, 0.9995]: sin(acos(dot)) = sqrt((1 - dot)(1 + dot)) to ~1e-7 relative (1 - dot
    // is exact for dot >= 0.5), and one sincos serves sin(theta) and cos(theta).
    const float theta_0 = acosf(dot);
    const float theta = fac * theta_0;
    float sin_theta, cos_theta;
    sincosf(theta, &sin_theta, &cos_theta);
    const float f = __fdividef(sin_theta, sqrtf((1.0f - dot) * (1.0f + dot)));
    const float rf = cos_theta - dot * f;
#pragma unroll
    for (int i = 0; i < 4; i++) res[i] = a[i] * rf + b[i] * f;
}

constexpr int POSE_WAVES = 4;           // 128 VGPRs, 40 KiB LDS per block: four blocks per CU
constexpr int G_STRIDE = 16;                 // floats per joint global in LDS
constexpr int POSE_TIMES_LDS_MAX = 6144;     // key times kept in LDS when the model's pool fits (24 KiB)

// LPC = lanes per character (64, 128, 192 or 256); BLOCK threads = CPB characters per block.
// LDS_TIMES: the model's whole key-time pool is staged in LDS once per block and the block is
// persistent (it strides over character groups), so the per-lane binary searches -- five
// dependent loads per path -- run at LDS latency instead of L2 latency.  Skeleton constants of
// the lane's joint (invmx, bind column 3, depth, parent) live in registers across characters.
template <int LPC, bool LDS_TIMES, int BLOCK>
__global__ __launch_bounds__(BLOCK, BLOCK == 256 ? POSE_WAVES : 1)
void k_pose(PoseArgs a)
{
    constexpr int CPB = BLOCK / LPC;
    // joint globals, 4 KiB per wave; once a character's chain is done the same 4 KiB are the
    // wave's staging tile for its coalesced stores
    __shared__ __attribute__((aligned(16))) float g_lds[CPB][LPC * G_STRIDE];
    __shared__ float times_lds[LDS_TIMES ? POSE_TIMES_LDS_MAX : 4];

    const int tid = threadIdx.x;
    const int cib = tid / LPC, j = tid % LPC;
    const int lane = lane_id();
    const uint32_t J = a.J;
    const bool lane_joint = cib < CPB && (uint32_t)j < J;
    const int depth = lane_joint ? a.depth[j] : -1;
    const bool reachable = depth >= 0;
    int32_t parent = lane_joint ? a.parent[j] : -1;
    if (parent >= (int32_t)J) parent = -1;
    float *G = g_lds[cib < CPB ? cib : 0];

    if (LDS_TIMES) {
        for (uint32_t q = tid; q < a.n_times; q += blockDim.x)
            times_lds[q] = a.times[q];
        __syncthreads();
    }
    const float *times = LDS_TIMES ? times_lds : a.times;

    const uint32_t n_groups = (a.n_chars + CPB - 1) / CPB;
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint32_t c = g * CPB + cib;
        const bool char_ok = cib < CPB && c < a.n_chars;
        const bool joint_ok = char_ok && lane_joint;
        const size_t cj = (size_t)c * J + j;

        // ---- 1. channels_transform: this joint's T, R, S at the character's frame time ----
        float T[3] = { 0, 0, 0 }, R[4] = { 0, 0, 0, 1 }, S[3] = { 1, 1, 1 };
        if (joint_ok) {
            uint32_t an = a.anim[c];
            if (an >= a.n_anims) an = 0;                         // an id outside the table would be a wild read
            const float time = a.frame_time[c];
            const uint4 *tab = a.chan_table + ((size_t)an * J + j) * 3;
            const uint4 e0 = tab[0], e1 = tab[1], e2 = tab[2];   // (time_off, data_off, nr, -)
            const int n0 = (int)e0.z, n1 = (int)e1.z, n2 = (int)e2.z;
            if (n0 <= 0 || n1 <= 0 || n2 <= 0) {                 // a path without a channel keeps its value
                const float *st = a.trs + 10 * cj;
                T[0] = st[0]; T[1] = st[1]; T[2] = st[2];
                R[0] = st[3]; R[1] = st[4]; R[2] = st[5]; R[3] = st[6];
                S[0] = st[7]; S[1] = st[8]; S[2] = st[9];
            }
            // one path at a time keeps the live state small (the searches run on LDS-resident times);
            // batching the three paths' key loads behind all three searches measured 205 us against 158
            if (n0 > 0) {
                int p, q;
                const float *t = times + e0.x;
                key_bracket(t, n0, time, p, q);
                const float fac = key_fac(time, t[p], t[q]);
                const float *d = a.data + e0.y;
#pragma unroll
                for (int k = 0; k < 3; k++) T[k] = lerp_ref(d[3 * p + k], d[3 * q + k], fac);
            }
            if (n1 > 0) {
                int p, q;
                const float *t = times + e1.x;
                key_bracket(t, n1, time, p, q);
                const float fac = key_fac(time, t[p], t[q]);
                const float *d = a.data + e1.y;
                const float qa[4] = { d[4 * p], d[4 * p + 1], d[4 * p + 2], d[4 * p + 3] };
                const float qb[4] = { d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3] };
                slerp_ref(R, qa, qb, fac);
            }
            if (n2 > 0) {
                int p, q;
                const float *t = times + e2.x;
                key_bracket(t, n2, time, p, q);
                const float fac = key_fac(time, t[p], t[q]);
                const float *d = a.data + e2.y;
#pragma unroll
                for (int k = 0; k < 3; k++) S[k] = lerp_ref(d[3 * p + k], d[3 * q + k], fac);
            }
        }

        // ---- 2. one_joint_transform (model.c:1352-1404): global_j = root_pose * L_0 * ... * L_j over the
        // joint's ancestor path, L = T * R * S.  The reference walks the tree top down; here every joint
        // folds its path by pointer jumping: after step s a lane holds the product of the last 2^s locals
        // of its path and the index of the ancestor 2^s above, so ceil(log2(levels)) LDS rounds replace
        // `levels` dependent ones.  Locals are affine, so the running products are kept as three rows
        // (the fourth is 0 0 0 1); only root_pose and invmx are treated as general 4x4.
        float4 M0, M1, M2;                                        // rows of the running product
        {
            float Rm[16];
            lmd::from_quat(Rm, R[0], R[1], R[2], R[3]);
            M0 = make_float4(E_(Rm, 0, 0) * S[0], E_(Rm, 1, 0) * S[1], E_(Rm, 2, 0) * S[2], T[0]);
            M1 = make_float4(E_(Rm, 0, 1) * S[0], E_(Rm, 1, 1) * S[1], E_(Rm, 2, 1) * S[2], T[1]);
            M2 = make_float4(E_(Rm, 0, 2) * S[0], E_(Rm, 1, 2) * S[1], E_(Rm, 2, 2) * S[2], T[2]);
        }
        int anc = joint_ok ? parent : -1;
        {
            float4 *slots = reinterpret_cast<float4 *>(G);
            const int sw_me = (j >> 2) & 3;                       // row swizzle: 16 neighbouring lanes hit 64 banks
            for (uint32_t st = 0; st < a.n_jump_steps; st++) {
                slots[4 * j + (0 ^ sw_me)] = M0;
                slots[4 * j + (1 ^ sw_me)] = M1;
                slots[4 * j + (2 ^ sw_me)] = M2;
                reinterpret_cast<int *>(&slots[4 * j + (3 ^ sw_me)])[0] = anc;
                if (LPC == WAVE) wave_lds_fence(); else __syncthreads();
                float4 A0, A1, A2;
                int anc2 = -1;
                const bool hop = anc >= 0;
                if (hop) {
                    const int sw = (anc >> 2) & 3;
                    A0 = slots[4 * anc + (0 ^ sw)];
                    A1 = slots[4 * anc + (1 ^ sw)];
                    A2 = slots[4 * anc + (2 ^ sw)];
                    anc2 = reinterpret_cast<const int *>(&slots[4 * anc + (3 ^ sw)])[0];
                }
                if (LPC == WAVE) wave_lds_fence(); else __syncthreads();
                if (hop) {
#pragma clang fp contract(fast)
                    const float4 B0 = M0, B1 = M1, B2 = M2;
#define CLAPGPU_AFFINE_ROW(A, OUT)                                                                  \
                    OUT = make_float4(A.x * B0.x + A.y * B1.x + A.z * B2.x,                          \
                                      A.x * B0.y + A.y * B1.y + A.z * B2.y,                          \
                                      A.x * B0.z + A.y * B1.z + A.z * B2.z,                          \
                                      A.x * B0.w + A.y * B1.w + A.z * B2.w + A.w)
                    CLAPGPU_AFFINE_ROW(A0, M0);
                    CLAPGPU_AFFINE_ROW(A1, M1);
                    CLAPGPU_AFFINE_ROW(A2, M2);
#undef CLAPGPU_AFFINE_ROW
                    anc = anc2;
                }
            }
        }
        float Gm[16];                                             // global = root_pose * path product
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma clang fp contract(fast)
            const float p0 = a.root_pose[r], p1 = a.root_pose[4 + r], p2 = a.root_pose[8 + r], p3 = a.root_pose[12 + r];
            E_(Gm, 0, r) = p0 * M0.x + p1 * M1.x + p2 * M2.x;
            E_(Gm, 1, r) = p0 * M0.y + p1 * M1.y + p2 * M2.y;
            E_(Gm, 2, r) = p0 * M0.z + p1 * M1.z + p2 * M2.z;
            E_(Gm, 3, r) = p0 * M0.w + p1 * M1.w + p2 * M2.w + p3;
        }

        // ---- 3. palette: joint_transforms = global * invmx; pos = e->mx * (joint_transforms * bind) * (0,0,0,1) ----
        float JT[16], pos[4] = { 0, 0, 0, 0 };
        if (joint_ok && reachable) {
            float IM[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 v = a.invmx[4 * j + q];
                IM[4 * q] = v.x; IM[4 * q + 1] = v.y; IM[4 * q + 2] = v.z; IM[4 * q + 3] = v.w;
            }
            const float4 b3 = a.bind[4 * j + 3];                  // only column 3 of bind reaches mpos
            const float bv[4] = { b3.x, b3.y, b3.z, b3.w };
            lmd::mul(JT, Gm, IM);                                 // model.c:1389
            float mpos[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {                         // column 3 of JT * bind (model.c:1393-1397)
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
            lmd::mul_vec4(pos, EM, mpos);                         // model.c:1400
        }

        // ---- stores (64 joints of one character per wave row) ----
        if (LPC != WAVE) __syncthreads();                         // every wave is done reading parents from G
        const uint32_t row_j0 = (uint32_t)(j - lane);             // first joint of this wave's row
        if (char_ok && row_j0 < J) {                              // wave-uniform
            const int nvalid = (int)(J - row_j0 < WAVE ? J - row_j0 : WAVE);
            const size_t row0 = (size_t)c * J + row_j0;
            float *tile_f = G + (row_j0 / WAVE) * (WAVE * G_STRIDE);      // the 4 KiB this wave's joints occupied
            float4 *tile = reinterpret_cast<float4 *>(tile_f);

            const float trs_row[10] = { T[0], T[1], T[2], R[0], R[1], R[2], R[3], S[0], S[1], S[2] };
            stage_rows<10>(tile_f, trs_row, lane);                // 2560 B
            wave_lds_fence();
            store_rows<10>(tile_f, a.trs + 10 * row0, lane, nvalid);
            wave_lds_fence();

            const uint64_t reach_mask = __ballot(joint_ok && reachable);
            const uint64_t full = nvalid == WAVE ? ~0ull : ((1ull << nvalid) - 1ull);
            if (reach_mask == full) {
                float4 v[4];
                stage_mat4(tile, JT, lane);
                wave_lds_fence();
                unstage_mat4(tile, v, lane);
                // the palette is what the skinning pass reads next: a plain store leaves it in the infinity
                // cache (205 MB at 50 k characters), unlike T/R/S, which nothing on the device reads back
                store_mat4_rows<false>(a.joint_transforms + 16 * row0, v, lane, nvalid);
                if (lane < nvalid)
                    reinterpret_cast<float4 *>(a.joint_pos)[row0 + lane] = make_float4(pos[0], pos[1], pos[2], pos[3]);
                wave_lds_fence();
            } else if (joint_ok && reachable) {                   // joints not under joint 0 are never written
                float4 *dj = reinterpret_cast<float4 *>(a.joint_transforms + 16 * cj);
#pragma unroll
                for (int q = 0; q < 4; q++)
                    dj[q] = make_float4(JT[4 * q], JT[4 * q + 1], JT[4 * q + 2], JT[4 * q + 3]);
                reinterpret_cast<float4 *>(a.joint_pos)[cj] = make_float4(pos[0], pos[1], pos[2], pos[3]);
            }
        }
        if (LPC != WAVE) __syncthreads();                         // next character reuses the globals in LDS
    }
}

// animated_update's clock (model.c:1563-1592): one lane per character
__global__ __launch_bounds__(256)
void k_animation_time(clapgpu_anim_clock k, double now, const double *now_dev)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= k.n_chars) return;
    if (now_dev) now = *now_dev;
    const double ft = (now - k.ani_time[c]) * (double)k.speed[c];
    k.frame_time[c] = (float)ft;
    const uint32_t an = k.anim[c];
    const bool ended = an < k.n_anims && ft >= (double)k.time_end[an];
    k.ended[c] = ended ? 1 : 0;
    if (ended && k.restart[c])
        k.ani_time[c] = now;                                        // animation_next -> animation_start
}

} // namespace clapgpu

using namespace clapgpu;

static int animation_time_launch(void *stream, const clapgpu_anim_clock *clk, double now, const double *now_dev);

extern "C" int clapgpu_animation_time(void *stream, const clapgpu_anim_clock *clk, double now)
{
    return animation_time_launch(stream, clk, now, nullptr);
}

extern "C" int clapgpu_animation_time_dev(void *stream, const clapgpu_anim_clock *clk, const double *now_dev)
{
    if (!now_dev)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    return animation_time_launch(stream, clk, 0.0, now_dev);
}

static int animation_time_launch(void *stream, const clapgpu_anim_clock *clk, double now, const double *now_dev)
{
    if (!clk)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (clk->n_chars == 0)
        return CLAPGPU_OK;
    if (!clk->anim || !clk->time_end || !clk->ani_time || !clk->speed || !clk->restart || !clk->frame_time ||
        !clk->ended)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    hipLaunchKernelGGL(k_animation_time, dim3((clk->n_chars + 255) / 256), dim3(256), 0, as_stream(stream), *clk, now,
                       now_dev);
    CLAPGPU_LAUNCH_CHECK("k_animation_time");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_pose_update(void *stream, const clapgpu_skeleton *sk, const clapgpu_animations *an,
                                   const clapgpu_pose_batch *pb)
{
    if (!sk || !an || !pb)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!sk->parent || !sk->depth || !sk->root_pose || !sk->invmx || !sk->bind || !an->chan_table ||
        !an->times || !an->data)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (pb->n_chars == 0)
        return CLAPGPU_OK;
    if (!pb->anim || !pb->frame_time || !pb->entity_mx || !pb->trs || !pb->joint_transforms || !pb->joint_pos)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (sk->nr_joints == 0 || sk->nr_joints > 256)               // JOINTS_MAX is 200 (shader_constants.h:6)
        return CLAPGPU_ERR_TOO_LARGE;

    PoseArgs a;
    a.J = sk->nr_joints;
    a.n_jump_steps = 0;                                        // ceil(log2(levels)): 2^steps >= longest path
    while ((1u << a.n_jump_steps) < sk->n_levels) a.n_jump_steps++;
    a.parent = sk->parent;
    a.depth = sk->depth;
    a.root_pose = sk->root_pose;
    a.invmx = reinterpret_cast<const float4 *>(sk->invmx);
    a.bind = reinterpret_cast<const float4 *>(sk->bind);
    a.chan_table = reinterpret_cast<const uint4 *>(an->chan_table);
    a.times = an->times;
    a.n_times = an->n_times;
    a.n_anims = an->n_anims ? an->n_anims : 1;
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
    const bool lds_times = an->n_times > 0 && an->n_times <= (uint32_t)POSE_TIMES_LDS_MAX;
    hipStream_t s = as_stream(stream);
    if (lds_times) {
        // persistent blocks (24 KiB key times + 16 KiB joint globals each): exactly as many as are
        // resident at once, so no block waits for a slot while the others hold their LDS copy
        const uint32_t threads = lpc == 192 ? 192 : 256, cpb = threads / lpc;
        const uint32_t n_groups = (pb->n_chars + cpb - 1) / cpb;
        const void *fn = lpc == 64 ? (const void *)k_pose<64, true, 256> : lpc == 128 ? (const void *)k_pose<128, true, 256>
                       : lpc == 192 ? (const void *)k_pose<192, true, 192> : (const void *)k_pose<256, true, 256>;
        static thread_local uint32_t resident[4] = { 0, 0, 0, 0 };
        uint32_t &res = resident[lpc / 64 - 1];
        if (!res) {
            int per_cu = 0, dev = 0;
            hipDeviceProp_t prop;
            CLAPGPU_HIP(hipGetDevice(&dev));
            CLAPGPU_HIP(hipGetDeviceProperties(&prop, dev));
            CLAPGPU_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)threads, 0));
            res = (uint32_t)(per_cu > 0 ? per_cu : 1) * (uint32_t)prop.multiProcessorCount;
        }
        const dim3 grid(n_groups < res ? n_groups : res), block(threads);
        switch (lpc) {
        case 64:  hipLaunchKernelGGL((k_pose<64, true, 256>), grid, block, 0, s, a); break;
        case 128: hipLaunchKernelGGL((k_pose<128, true, 256>), grid, block, 0, s, a); break;
        case 192: hipLaunchKernelGGL((k_pose<192, true, 192>), grid, block, 0, s, a); break;
        default:  hipLaunchKernelGGL((k_pose<256, true, 256>), grid, block, 0, s, a); break;
        }
    } else {
        const uint32_t threads = lpc == 192 ? 192 : 256, cpb = threads / lpc;
        const dim3 grid((pb->n_chars + cpb - 1) / cpb), block(threads);
        switch (lpc) {
        case 64:  hipLaunchKernelGGL((k_pose<64, false, 256>), grid, block, 0, s, a); break;
        case 128: hipLaunchKernelGGL((k_pose<128, false, 256>), grid, block, 0, s, a); break;
        case 192: hipLaunchKernelGGL((k_pose<192, false, 192>), grid, block, 0, s, a); break;
        default:  hipLaunchKernelGGL((k_pose<256, false, 256>), grid, block, 0, s, a); break;
        }
    }
    CLAPGPU_LAUNCH_CHECK("k_pose");
    return CLAPGPU_OK;
}
