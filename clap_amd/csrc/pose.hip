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
// Numerics: this path is held to 1e-5 relative, not bit-exact -- the reference itself
// goes through the host's double libm here.  The kernel is VALU-bound, so the arithmetic is fp32
// with FMA contraction, and slerp's acos / sin / cos are short polynomials valid on the only
// intervals slerp can reach (max error 1.7e-7, checked against libm in tests/test_pose_skin_gpu.py).
#include <string.h>
#include "common.h"
#include "lm_dev.h"

// Everything below may contract a*b+c into one FMA (the rest of the library is built with
// -ffp-contract=off for bit-exact parity with the reference's x86 arithmetic; k_animation_time
// at the end of this file has no multiply-add to contract).
#pragma clang fp contract(fast)

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
    // key-major copy of the pools (clapgpu_animations_pack), or nullptr: [anim][path][key row][64 lanes]
    const float    *pk_times;       // rows padded to pk_kp (a power of two > the longest channel) with +INF
    const float4   *pk_vals;        // pk_k rows of float4 (T / S: xyz, R: xyzw)
    uint32_t        pk_k, pk_kp;
    // batch
    uint32_t        n_chars;
    const uint32_t *anim;
    const float    *frame_time;
    const uint32_t *entity;
    const float    *entity_mx;
    float          *trs;
    float          *joint_transforms;
    float          *joint_pos;
    uint32_t        skip;           // CLAPGPU_POSE_SKIP_*
};

// model.c:1266-1288 for strictly increasing key times (glTF): the bracket does not depend on
// the reference's search cursor, so a search for lo = #{i : t[i] < time} gives the same (prev, next).
// `top` = the largest power of two <= the longest channel of the wave (wave-uniform), so the loop has
// no divergent exit: five steps of (add, compare, LDS read, compare, select) for 30 keys.
__device__ __forceinline__ void key_bracket(const float *t, int nr, float time, int top, int &prev, int &next)
{
    int lo = 0;
    for (int step = top; step > 0; step >>= 1) {
        const int cand = lo + step;
        const float tc = t[cand <= nr ? cand - 1 : 0];
        if (cand <= nr && tc < time) lo = cand;
    }
    const bool wrap = time < t[0] || time > t[nr - 1];            // before the first / past the last key
    prev = lo > 0 ? lo - 1 : 0;
    next = prev + 1 < nr - 1 ? prev + 1 : nr - 1;
    prev = wrap ? nr - 1 : prev;
    next = wrap ? 0 : next;
}

// model.c:1312-1317.  The quotient is the hardware reciprocal (1 ulp) corrected by one Newton step on the residual:
// correctly rounded but for rare ties, at 4 instructions instead of the IEEE division's ~12.  (The bare reciprocal put
// a 1e-7 relative error into the key fraction, which a lerp between keys of opposite sign turns into 4e-6 of the result.)
__device__ __forceinline__ float key_fac(float time, float p_time, float n_time)
{
    if (p_time > n_time) return time < n_time ? 1.f : 0.f;
    if (p_time < n_time) {
        const float d = n_time - p_time, x = time - p_time;
        const float r = __builtin_amdgcn_rcpf(d);
        const float q = x * r;
        return __builtin_fmaf(__builtin_fmaf(-d, q, x), r, q);
    }
    return 0.f;
}

// interp.h:25-29: (float)((double)a * (1.0 - (double)blend) + (double)(b * blend)) -- the first product and the sum in
// double, b * blend a rounded fp32 product.  In fp32 with the residual of 1 - blend carried along:
// 1 - blend = g + gl exactly (g the rounded difference, gl what the rounding dropped), so a * (1 - blend) + p1 =
// a * g + p1 (one FMA: a single rounding of the sum, exact under cancellation) + a * gl (a second FMA for the 2^-25-sized
// rest).  Within an ulp of the RESULT of the reference's value also where the two products cancel, which the plain
// fp32 form b * f + a * (1 - f) was not (errors of an ulp of the operands: up to 4e-6 of a cancelled result).
struct LerpFac { float f, g, gl; };
__device__ __forceinline__ LerpFac lerp_fac(float fac)
{
    LerpFac l;
    l.f = fac;
    l.g = 1.0f - fac;
    l.gl = (1.0f - l.g) - fac;          // exact: both differences are of neighbouring magnitudes
    return l;
}
__device__ __forceinline__ float lerp_ref(float a, float b, const LerpFac l)
{
    const float p1 = __fmul_rn(b, l.f);                     // the reference rounds this product to fp32 on its own
    return __builtin_fmaf(a, l.gl, __builtin_fmaf(a, l.g, p1));
}

// acos on [0, 1): the rational core of fdlibm's acosf (R(z) = z*P(z)/Q(z), |error| < 7e-9 on z <= 0.25)
// without its hi/lo splitting; max error 1.4e-7 on the interval slerp reaches.
__device__ __forceinline__ float acos01(float d)
{
    const bool big = d > 0.5f;
    const float z = big ? (1.0f - d) * 0.5f : d * d;
    const float pn = z * (1.6666586697e-01f + z * (-4.2743422091e-02f + z * -8.6563630030e-03f));
    const float r = pn * __builtin_amdgcn_rcpf(1.0f + z * -7.0662963390e-01f);
    const float x = big ? __builtin_amdgcn_sqrtf(z) : d;       // v_sqrt_f32 (1 ulp), z in [0, 0.25]: no denormal scaling needed
    const float y = x + x * r;                                   // asin(x)
    return big ? 2.0f * y : 1.5707963267948966f - y;
}

// sin and cos on [0, pi/2]: no range reduction needed (theta = fac * acos(dot), fac in [0, 1], dot >= 0);
// Taylor to x^11 / x^12, max error 1.7e-7 / 1.3e-7.
__device__ __forceinline__ void sincos_halfpi(float x, float &sn, float &cs)
{
    const float x2 = x * x;
    float ps = -2.5052108385e-08f;
    ps = ps * x2 + 2.7557319224e-06f;
    ps = ps * x2 - 1.9841269841e-04f;
    ps = ps * x2 + 8.3333333333e-03f;
    ps = ps * x2 - 1.6666666667e-01f;
    sn = x + x * x2 * ps;
    float pc = 2.0876756988e-09f;
    pc = pc * x2 - 2.7557319224e-07f;
    pc = pc * x2 + 2.4801587302e-05f;
    pc = pc * x2 - 1.3888888889e-03f;
    pc = pc * x2 + 4.1666666667e-02f;
    pc = pc * x2 - 0.5f;
    cs = 1.0f + x2 * pc;
}

// interp.h:67-118 quat_slerp / quat_interp
__device__ __forceinline__ void slerp_ref(float (&res)[4], const float (&a)[4], const float (&b_in)[4], float fac)
{
    float dot = b_in[0] * a[0] + b_in[1] * a[1] + b_in[2] * a[2] + b_in[3] * a[3];
    const float sgn = dot < 0.0f ? -1.0f : 1.0f;                 // shortest arc: b = -b, dot = -dot
    dot *= sgn;
    const float b[4] = { b_in[0] * sgn, b_in[1] * sgn, b_in[2] * sgn, b_in[3] * sgn };
    // (double)dot > 0.9995 in the reference; 0.9995f is the largest float below 0.9995, so the fp32
    // comparison takes the same branch
    if (dot > 0.9995f) {                                         // nlerp + vec4_norm (the recomputed dot is >= 0)
        const float rfac = 1.f - fac;
        float t[4];
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = rfac * a[i] + fac * b[i];
        const float k = rsqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3]);
#pragma unroll
        for (int i = 0; i < 4; i++) res[i] = t[i] * k;
        return;
    }
    // dot in [0, 0.9995]: sin(acos(dot)) = sqrt((1 - dot)(1 + dot)) (1 - dot is exact for dot >= 0.5)
    const float theta = fac * acos01(dot);
    float sin_theta, cos_theta;
    sincos_halfpi(theta, sin_theta, cos_theta);
    const float f = sin_theta * rsqrtf((1.0f - dot) * (1.0f + dot));
    const float rf = cos_theta - dot * f;
#pragma unroll
    for (int i = 0; i < 4; i++) res[i] = a[i] * rf + b[i] * f;
}

// A matrix row as two register pairs, so the affine products below are packed-fp32 FMAs whose scalar
// factor comes out of a pair through op_sel (no moves): 18 v_pk_fma/mul per 3x4 product.
typedef float v2f __attribute__((ext_vector_type(2)));
struct Row { v2f lo, hi; };
__device__ __forceinline__ Row row_of(const float4 v) { Row r; r.lo = v2f{v.x, v.y}; r.hi = v2f{v.z, v.w}; return r; }
__device__ __forceinline__ float4 f4_of(const Row r) { return make_float4(r.lo.x, r.lo.y, r.hi.x, r.hi.y); }
// one row of A * B for affine A, B (fourth rows 0 0 0 1): A.x B0 + A.y B1 + A.z B2 + (0, 0, 0, A.w)
__device__ __forceinline__ Row affine_row(const float4 A, const Row B0, const Row B1, const Row B2)
{
    Row o;
    const v2f ax = { A.x, A.x }, ay = { A.y, A.y }, az = { A.z, A.z };
    o.lo = ax * B0.lo + ay * B1.lo + az * B2.lo;
    o.hi = ax * B0.hi + ay * B1.hi + az * B2.hi;
    o.hi.y += A.w;
    return o;
}

// One key value (vec3 / quat) as ONE load per lane: the pools are only 4-byte aligned, which global loads take
// (unaligned access mode); as scalar loads every component was its own trip through the texture path -- a lane's keys
// are nobody else's, so nothing coalesces -- 20 per lane and character instead of 6.  (Time-neutral in a same-session
// A/B: 139-148 us either way; kept for the instruction count.)
typedef float key3 __attribute__((ext_vector_type(3), aligned(4)));
typedef float key4 __attribute__((ext_vector_type(4), aligned(4)));

constexpr int POSE_WAVES = 3;           // three waves per SIMD (168 VGPRs) run as fast as four (measured): the registers go to the joint constants
constexpr int G_STRIDE = 16;                 // floats per joint global in LDS
constexpr int POSE_TIMES_LDS_MAX = 6400;       // key times kept in LDS when the model's pool fits: 25 KiB (one animation of <= 31 keys per
                                             // channel in key-major form with its key counts; with the globals and joint constants 47 KiB per block)

// ---- the one-wavefront-per-character loop without store waits ---------------------------------------------------
// gfx950 retires vector loads and stores through ONE counter, in issue order: waiting for a load means waiting for
// every store issued before it, and the compiler can only wait for "all but the last N operations" when it can count
// the operations behind the load on every path.  The general loop below cannot offer that (its stores sit in lane
// and row tests, its key gathers follow the previous character's stores), so each character's first load waits for
// the previous character's 11 KB of stores to be acknowledged by memory while the SIMD's other two wavefronts do the
// same.  Here:
//   * every store is a buffer store issued by all 64 lanes under no branch; rows shorter than 64 joints, characters
//     past the end and masked outputs are clipped by the descriptor's size (range check per dword), not by exec;
//   * per-character scalars (animation id, frame time, entity matrix, root pose) come through the scalar cache;
//   * the order per character g is: interpolate g's keys -> search + gather the keys of g+1 -> hierarchy and palette
//     of g -> request the channel records of g+2's... of the character after next -> stores of g.  Every wait is then
//     for an operation with a known number of younger ones behind it, and the gathers of g+1 fly under g's
//     hierarchy rounds.
typedef int pose_v4i __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) float *pose_cfloat;
typedef const __attribute__((address_space(4))) uint32_t *pose_cu32;
constexpr int POSE_RSRC_FLAGS = 0x00020000;                      // raw buffer, 32-bit data format

__device__ __forceinline__ void buffer_store4(const float4 v, __amdgpu_buffer_rsrc_t r, uint32_t byte_off, bool stream)
{
    const pose_v4i d = { __float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w) };
    if (stream) __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)byte_off, 0, 2);      // nt
    else __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)byte_off, 0, 0);
}

struct PoseKeys { key3 ta, tb; key4 ra, rb; key3 sa, sb; float f0, f1, f2; };

// The three paths' searches as ONE loop: three independent LDS reads per step instead of three loops of five dependent
// ones (114.7 / 121.6 / 115.6 us against 131.6 / 121.9 / 129.4 in a same-session A/B; in the general loop, whose
// wavefronts wait on their stores anyway, the same interleaving measured slower).
__device__ __forceinline__ PoseKeys pose_gather_keys(const float *times, const float *kdata, int top, float time,
                                                     const uint4 e0, const uint4 e1, const uint4 e2)
{
    PoseKeys k;
    const float *t0 = times + e0.x, *t1 = times + e1.x, *t2 = times + e2.x;
    const int n0 = (int)e0.z, n1 = (int)e1.z, n2 = (int)e2.z;
    int l0 = 0, l1 = 0, l2 = 0;
    for (int step = top; step > 0; step >>= 1) {
        const int c0 = l0 + step, c1 = l1 + step, c2 = l2 + step;
        const float a0 = t0[c0 <= n0 ? c0 - 1 : 0], a1 = t1[c1 <= n1 ? c1 - 1 : 0], a2 = t2[c2 <= n2 ? c2 - 1 : 0];
        if (c0 <= n0 && a0 < time) l0 = c0;
        if (c1 <= n1 && a1 < time) l1 = c1;
        if (c2 <= n2 && a2 < time) l2 = c2;
    }
    // model.c:1266-1288's wrap (time before the first / past the last key) without reading t[0] and t[nr - 1] again:
    // lo counts the keys below `time`, so lo == nr <=> time > t[nr - 1], and with lo == 0 the bracket's own first
    // key IS t[0].  Six LDS reads per joint and character fewer; a wrapped lane (rare) re-reads its two keys.
    auto finish = [&](const float *t, int nr, int lo, int &prev, int &next, float &tp, float &tn) {
        prev = lo > 0 ? lo - 1 : 0;
        next = prev + 1 < nr - 1 ? prev + 1 : nr - 1;
        tp = t[prev]; tn = t[next];
        if (lo == nr || (lo == 0 && time < tp)) {
            prev = nr - 1; next = 0;
            tp = t[prev]; tn = t[next];
        }
    };
    int p0, q0, p1, q1, p2, q2;
    float tp0, tn0, tp1, tn1, tp2, tn2;
    finish(t0, n0, l0, p0, q0, tp0, tn0); finish(t1, n1, l1, p1, q1, tp1, tn1); finish(t2, n2, l2, p2, q2, tp2, tn2);
    const float *d0 = kdata + e0.y, *d1 = kdata + e1.y, *d2 = kdata + e2.y;
    k.ta = *reinterpret_cast<const key3 *>(d0 + 3 * p0); k.tb = *reinterpret_cast<const key3 *>(d0 + 3 * q0);
    k.ra = *reinterpret_cast<const key4 *>(d1 + 4 * p1); k.rb = *reinterpret_cast<const key4 *>(d1 + 4 * q1);
    k.sa = *reinterpret_cast<const key3 *>(d2 + 3 * p2); k.sb = *reinterpret_cast<const key3 *>(d2 + 3 * q2);
    k.f0 = key_fac(time, tp0, tn0);
    k.f1 = key_fac(time, tp1, tn1);
    k.f2 = key_fac(time, tp2, tn2);
    return k;
}

// The same for the KEY-MAJOR pools (clapgpu_animations_pack): key k of (path, joint j) sits at row k, column j.
//   * times in LDS, [path][row][64]: lane j's read of ANY row lands in bank j % 32 of its half-wave -- the searches'
//     ds_read_b32 are conflict-free whatever rows the 64 lanes are at (in the channel-major pool a lane's address was
//     its channel's offset + a data-dependent key: ~4 distinct addresses per bank, 48 % of all LDS cycles were conflicts);
//   * rows past a channel's last key hold +INF, so the search needs no bounds test: lo = #{k : t[k] < time} by five
//     (add, read, compare-select) steps for up to 31 keys;
//   * values in memory as float4 [path][row][64]: lanes whose brackets are the same key -- all of them when the channels
//     share their key times, as exported glTF samplers do -- read ONE contiguous 1 KiB row instead of 64 scattered 12-byte
//     pieces 360 bytes apart.
template <int LPC>
__device__ __forceinline__ PoseKeys pose_gather_keys_packed(const float *tl, const float4 *vals, const int kp, const int kk,
                                                            const float time, const int n0, const int n1, const int n2,
                                                            const int lane)
{
    constexpr int COLS = LPC;                                    // a row holds one key of every joint: LPC columns (64 per wavefront)
    PoseKeys k;
    const float *t0 = tl + lane, *t1 = t0 + kp * COLS, *t2 = t1 + kp * COLS;
    int l0 = 0, l1 = 0, l2 = 0;
    for (int step = kp >> 1; step > 0; step >>= 1) {
        const float a0 = t0[(l0 + step - 1) * COLS], a1 = t1[(l1 + step - 1) * COLS], a2 = t2[(l2 + step - 1) * COLS];
        l0 += a0 < time ? step : 0;
        l1 += a1 < time ? step : 0;
        l2 += a2 < time ? step : 0;
    }
    auto finish = [&](const float *t, int nr, int lo, int &prev, int &next, float &tp, float &tn) {
        prev = lo > 0 ? lo - 1 : 0;
        next = prev + 1 < nr - 1 ? prev + 1 : nr - 1;
        tp = t[prev * COLS]; tn = t[next * COLS];
        if (lo == nr || (lo == 0 && time < tp)) {                // model.c:1266-1288's wrap, as in pose_gather_keys
            prev = nr - 1; next = 0;
            tp = t[prev * COLS]; tn = t[next * COLS];
        }
    };
    int p0, q0, p1, q1, p2, q2;
    float tp0, tn0, tp1, tn1, tp2, tn2;
    finish(t0, n0, l0, p0, q0, tp0, tn0); finish(t1, n1, l1, p1, q1, tp1, tn1); finish(t2, n2, l2, p2, q2, tp2, tn2);
    const float4 *v0 = vals + lane, *v1 = v0 + kk * COLS, *v2 = v1 + kk * COLS;
    const float4 ta = v0[p0 * COLS], tb = v0[q0 * COLS], ra = v1[p1 * COLS], rb = v1[q1 * COLS], sa = v2[p2 * COLS], sb = v2[q2 * COLS];
    k.ta = key3{ ta.x, ta.y, ta.z }; k.tb = key3{ tb.x, tb.y, tb.z };
    k.ra = key4{ ra.x, ra.y, ra.z, ra.w }; k.rb = key4{ rb.x, rb.y, rb.z, rb.w };
    k.sa = key3{ sa.x, sa.y, sa.z }; k.sb = key3{ sb.x, sb.y, sb.z };
    k.f0 = key_fac(time, tp0, tn0);
    k.f1 = key_fac(time, tp1, tn1);
    k.f2 = key_fac(time, tp2, tn2);
    return k;
}

// PACKED: the key-major pools (times + key counts in LDS at `times`, values at a.pk_vals): no channel records are read
// in the loop at all -- 48 bytes per lane and character that the channel-major form fetched from L2.
// bits (1 ^ 3, 2) of the joint index: a bijection of bits (1, 2) for the stores' eight-lane groups and of bits (2, 3) for
// the reads of ancestors that share their low two bits; 0 for the identity slot (joint 64)
__device__ __forceinline__ int jump_swz(int j) { return (((j >> 1) ^ (j >> 3)) & 1) | ((j >> 1) & 2); }

// LDS hand-over between the wavefronts of one character (skeletons of more than 64 joints: 2-4 wavefronts each): the LDS
// counter only -- __syncthreads() would also wait for the wavefront's global stores, the very wait this loop exists to avoid
template <int LPC>
__device__ __forceinline__ void pose_lds_sync()
{
    if (LPC == WAVE) wave_lds_fence();
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int LPC, int CPB, bool PACKED>
__device__ __forceinline__ void pose_stream_loop(const PoseArgs &a, float *G, int *anc_lds, const float *times, const float *kdata,
                                                 const int top, uint4 e0, uint4 e1, uint4 e2, const float4 *jconst,
                                                 const int parent, const int j, const int cib_v)
{
    static_assert(LPC == WAVE || PACKED, "more than one wavefront per character: key-major pools only");
    const int lane = j & (WAVE - 1);                             // lane of the wavefront; j = the joint (column of the pools)
    const int row_j0 = j - lane;                                 // first joint of this wavefront's 64-joint row
    const uint32_t J = a.J;
    const uint32_t cib = (uint32_t)__builtin_amdgcn_readfirstlane(cib_v);
    const uint32_t n_groups = (a.n_chars + CPB - 1) / CPB;
    const uint32_t jc = (uint32_t)j < J ? (uint32_t)j : J - 1;
    const pose_cu32 anim_s = (pose_cu32)a.anim, entity_s = (pose_cu32)a.entity;
    const pose_cfloat time_s = (pose_cfloat)a.frame_time, root_s = (pose_cfloat)a.root_pose;
    const bool with_trs = !(a.skip & CLAPGPU_POSE_SKIP_TRS), with_pos = !(a.skip & CLAPGPU_POSE_SKIP_JOINT_POS);

    auto char_of = [&](uint32_t g_) {                            // past the end: a valid character whose stores are clipped
        const uint32_t c_ = g_ * CPB + cib;
        return c_ < a.n_chars ? c_ : a.n_chars - 1;
    };
    auto anim_of = [&](uint32_t c_) { const uint32_t an = anim_s[c_]; return an < a.n_anims ? an : 0u; };

    // Per-character scalars travel one character ahead in SGPRs: a scalar load consumed where it is issued costs its
    // round trip to L2 (the scalar cache does not hold 50 000 characters' worth), three of them per character.
    auto entity_of = [&](uint32_t c_) { return a.entity ? entity_s[c_] : c_; };
    uint32_t g = blockIdx.x;
    const uint32_t c0 = char_of(g), c1 = char_of(g + gridDim.x);
    // the character's entity matrix: element (lane & 15) per lane, one vector load a character ahead, read back with
    // v_readlane where joint positions are formed (as a scalar load its 64 bytes were waited for where they were asked for)
    float em_v = with_pos ? a.entity_mx[16 * (size_t)entity_of(c0) + (lane & 15)] : 0.f;
    float tm_next = time_s[c1];
    // PACKED: per animation 3 * kp rows of LPC key times, then (after all animations' times) 3 rows of LPC key counts
    const int kp = (int)a.pk_kp, kk = (int)a.pk_k;
    const uint32_t *nr_lds = reinterpret_cast<const uint32_t *>(times + (size_t)a.n_anims * 3 * kp * LPC);
    auto gather_packed = [&](uint32_t an, float tm) {
        const uint32_t *nr = nr_lds + an * 3 * LPC + j;
        return pose_gather_keys_packed<LPC>(times + (size_t)an * 3 * kp * LPC, a.pk_vals + (size_t)an * 3 * kk * LPC, kp, kk, tm,
                                            (int)nr[0], (int)nr[LPC], (int)nr[2 * LPC], j);
    };
    uint32_t an_next = PACKED ? anim_of(c1) : 0u;
    // the first character's keys (its channel records were requested by the caller) -- waited for HERE, so that no
    // wait for them is left pending into the loop, where it would stand for "all but a few operations" on the way round
    PoseKeys kv;
    if constexpr (PACKED) {
        kv = gather_packed(anim_of(c0), time_s[c0]);
        asm volatile("" : : "v"(kv.ta.x), "v"(kv.tb.x), "v"(kv.ra.x), "v"(kv.rb.x), "v"(kv.sa.x), "v"(kv.sb.x), "v"(em_v));
    } else {
        kv = pose_gather_keys(times, kdata, top, time_s[c0], e0, e1, e2);
        const uint4 *tab = a.chan_table + ((size_t)anim_of(c1) * J + jc) * 3;
        e0 = tab[0]; e1 = tab[1]; e2 = tab[2];
        asm volatile("" : : "v"(kv.ta.x), "v"(kv.tb.x), "v"(kv.ra.x), "v"(kv.rb.x), "v"(kv.sa.x), "v"(kv.sb.x),
                            "v"(e0.x), "v"(e1.x), "v"(e2.x), "v"(em_v));
    }

    for (; g < n_groups; g += gridDim.x) {
        const uint32_t c_raw = g * CPB + cib;
        const bool c_ok = c_raw < a.n_chars;                     // wave-uniform
        const uint32_t c = c_ok ? c_raw : a.n_chars - 1;
        // scalars of the characters after this one: issued now, consumed a phase or a character later
        const uint32_t c2 = char_of(g + 2 * gridDim.x);
        const uint32_t an2 = anim_of(c2);
        const float tm_next2 = time_s[c2];
        const uint32_t ei_next = entity_of(char_of(g + gridDim.x));

        // ---- 1. this character's T, R, S from its keys
        float T[3], R[4], S[3];
        const LerpFac lt = lerp_fac(kv.f0), ls = lerp_fac(kv.f2);
        T[0] = lerp_ref(kv.ta.x, kv.tb.x, lt); T[1] = lerp_ref(kv.ta.y, kv.tb.y, lt); T[2] = lerp_ref(kv.ta.z, kv.tb.z, lt);
        {
            const float qa[4] = { kv.ra.x, kv.ra.y, kv.ra.z, kv.ra.w };
            const float qb[4] = { kv.rb.x, kv.rb.y, kv.rb.z, kv.rb.w };
            slerp_ref(R, qa, qb, kv.f1);
        }
        S[0] = lerp_ref(kv.sa.x, kv.sb.x, ls); S[1] = lerp_ref(kv.sa.y, kv.sb.y, ls); S[2] = lerp_ref(kv.sa.z, kv.sb.z, ls);

        // ---- 2. the next character's key search (LDS) and key gathers, in flight under the hierarchy below
        if constexpr (PACKED) kv = gather_packed(an_next, tm_next);
        else kv = pose_gather_keys(times, kdata, top, tm_next, e0, e1, e2);

        // ---- 3. hierarchy by pointer jumping, palette, joint position: as in the general loop
        Row M0, M1, M2;
        {
            const float qa = R[3], qb = R[0], qc = R[1], qd = R[2];
            const float a2 = qa * qa, b2 = qb * qb, c2 = qc * qc, d2 = qd * qd;
            const float bc = qb * qc, ad = qa * qd, bd = qb * qd, ac = qa * qc, cd = qc * qd, ab = qa * qb;
            const v2f s01 = { S[0], S[1] };
            M0.lo = v2f{ a2 + b2 - c2 - d2, 2.f * (bc - ad) } * s01;  M0.hi = v2f{ 2.f * (bd + ac) * S[2], T[0] };
            M1.lo = v2f{ 2.f * (bc + ad), a2 - b2 + c2 - d2 } * s01;  M1.hi = v2f{ 2.f * (cd - ab) * S[2], T[1] };
            M2.lo = v2f{ 2.f * (bd - ac), 2.f * (cd + ab) } * s01;    M2.hi = v2f{ (a2 - b2 - c2 + d2) * S[2], T[2] };
        }
        int anc = parent;
        {
            // 64-byte slot per joint, its three rows XOR-swizzled by jump_swz(): the 16-byte stores of eight neighbouring
            // lanes then cover all 32 banks (ds_write_b128 is serviced eight lanes at a time, bank = dword mod 32), and
            // the 16-byte reads of sixteen different ancestors all 64 (ds_read_b128: dword mod 64).  The ancestor index
            // travels in its own dword array: in the slot's fourth row its ds_write_b32 hit eight banks with 32 lanes.
            float4 *slots = reinterpret_cast<float4 *>(G);
            const int sw_me = jump_swz(j);
            for (uint32_t st = 0; st < a.n_jump_steps; st++) {
                slots[4 * j + (0 ^ sw_me)] = f4_of(M0);
                slots[4 * j + (1 ^ sw_me)] = f4_of(M1);
                slots[4 * j + (2 ^ sw_me)] = f4_of(M2);
                anc_lds[j] = anc;
                pose_lds_sync<LPC>();
                const int src = anc >= 0 ? anc : LPC;
                const int sw = jump_swz(src);
                const float4 A0 = slots[4 * src + (0 ^ sw)];
                const float4 A1 = slots[4 * src + (1 ^ sw)];
                const float4 A2 = slots[4 * src + (2 ^ sw)];
                anc = anc_lds[src];
                pose_lds_sync<LPC>();
                const Row B0 = M0, B1 = M1, B2 = M2;
                M0 = affine_row(A0, B0, B1, B2);
                M1 = affine_row(A1, B0, B1, B2);
                M2 = affine_row(A2, B0, B1, B2);
            }
        }
        float Gm[16];
        {
            const float4 m0 = f4_of(M0), m1 = f4_of(M1), m2 = f4_of(M2);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float p0 = root_s[r], p1 = root_s[4 + r], p2 = root_s[8 + r], p3 = root_s[12 + r];
                E_(Gm, 0, r) = p0 * m0.x + p1 * m1.x + p2 * m2.x;
                E_(Gm, 1, r) = p0 * m0.y + p1 * m1.y + p2 * m2.y;
                E_(Gm, 2, r) = p0 * m0.z + p1 * m1.z + p2 * m2.z;
                E_(Gm, 3, r) = p0 * m0.w + p1 * m1.w + p2 * m2.w + p3;
            }
        }
        // the joint's constants (invmx, column 3 of bind) come from LDS here: held in registers across characters, as the
        // general loop holds them, they push this loop's extra live values (the next character's keys) into scratch
        float JT[16], pos[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            const float4 im = jconst[cc * LPC + j];              // column cc of invmx
#pragma unroll
            for (int r = 0; r < 4; r++)
                E_(JT, cc, r) = E_(Gm, 0, r) * im.x + E_(Gm, 1, r) * im.y + E_(Gm, 2, r) * im.z + E_(Gm, 3, r) * im.w;
        }
        if (with_pos) {                                          // uniform
            const float4 b3 = jconst[4 * LPC + j];
            const float bv[4] = { b3.x, b3.y, b3.z, b3.w };
            float mpos[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float sm = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) sm += E_(JT, k, r) * bv[k];
                mpos[r] = sm;
            }
            float em[16];
#pragma unroll
            for (int k = 0; k < 16; k++) em[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(em_v), k));
#pragma unroll
            for (int r = 0; r < 4; r++)
                pos[r] = E_(em, 0, r) * mpos[0] + E_(em, 1, r) * mpos[1] + E_(em, 2, r) * mpos[2] + E_(em, 3, r) * mpos[3];
        }

        // ---- 4. the channel records of the character after next, ahead of the stores (PACKED: only its animation id)
        if constexpr (PACKED) {
            an_next = an2;
        } else {
            const uint4 *tab = a.chan_table + ((size_t)an2 * J + jc) * 3;
            e0 = tab[0]; e1 = tab[1]; e2 = tab[2];
        }
        if (with_pos) em_v = a.entity_mx[16 * (size_t)ei_next + (lane & 15)];
        tm_next = tm_next2;

        // ---- 5. stores: all lanes, no branch; the descriptors clip (each wavefront's own: its 64-joint row of the character)
        const uint32_t rj0 = (uint32_t)__builtin_amdgcn_readfirstlane(row_j0);
        const uint32_t nrow = c_ok && J > rj0 ? (J - rj0 < (uint32_t)WAVE ? J - rj0 : (uint32_t)WAVE) : 0u;
        const size_t row0 = (size_t)c * J + rj0;
        const __amdgpu_buffer_rsrc_t rs_trs = __builtin_amdgcn_make_buffer_rsrc(a.trs + 10 * row0, 0, with_trs ? (int)(nrow * 40u) : 0, POSE_RSRC_FLAGS);
        const __amdgpu_buffer_rsrc_t rs_jt = __builtin_amdgcn_make_buffer_rsrc(a.joint_transforms + 16 * row0, 0, (int)(nrow * 64u), POSE_RSRC_FLAGS);
        const __amdgpu_buffer_rsrc_t rs_pos = __builtin_amdgcn_make_buffer_rsrc(a.joint_pos ? a.joint_pos + 4 * row0 : a.joint_transforms, 0, with_pos ? (int)(nrow * 16u) : 0, POSE_RSRC_FLAGS);
        // this wavefront's 64-joint row of the character: its own 4 KiB of the character's slots as the staging tile
        float *tile_f = G + row_j0 * G_STRIDE;
        float4 *tile = reinterpret_cast<float4 *>(tile_f);
        {
            const float trs_row[10] = { T[0], T[1], T[2], R[0], R[1], R[2], R[3], S[0], S[1], S[2] };
            stage_rows<10>(tile_f, trs_row, lane);
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < 3; k++)                           // 160 16-byte pieces; the third round's upper half lies past the row
                buffer_store4(tile[k * WAVE + lane], rs_trs, (uint32_t)(k * WAVE + lane) * 16u, true);
            wave_lds_fence();
        }
        {
            float4 v[4];
            stage_mat4(tile, JT, lane);
            wave_lds_fence();
            unstage_mat4(tile, v, lane);
#pragma unroll
            for (int k = 0; k < 4; k++)
                buffer_store4(v[k], rs_jt, (uint32_t)(k * WAVE + lane) * 16u, false);
            buffer_store4(make_float4(pos[0], pos[1], pos[2], pos[3]), rs_pos, (uint32_t)lane * 16u, false);
            wave_lds_fence();
        }
    }
}

// LPC = lanes per character (64, 128, 192 or 256); BLOCK threads = CPB characters per block.
// A producer / consumer split of this chain inside a block (keyframe waves -> LDS -> hierarchy waves) was built in
// round 2 and measured slower, 172 us against 145 (profiles/r02_experiments/pose_producer_consumer.md, commit af48632);
// so was a split into two launches (keyframe stage writing T/R/S, hierarchy stage reading them back): 66 + 78 us with
// the keyframe stage at six wavefronts per SIMD, 85 + 78 us with all of the model's keys in LDS
// (profiles/r02_experiments/pose_two_launches.md).
// MODE 0: keyframes read through L2.  MODE 1: key times in LDS.  (Key VALUES in LDS as well -- one
// 960-thread block per CU holding the model's whole 75 KiB pool -- was built and measured: no faster,
// profiles/r01_experiments/pose_bounds.md.)
// LDS_TIMES: the model's whole key-time pool is staged in LDS once per block and the block is
// persistent (it strides over character groups), so the per-lane binary searches -- five
// dependent loads per path -- run at LDS latency instead of L2 latency.  Skeleton constants of
// the lane's joint (invmx, bind column 3, depth, parent) live in registers across characters.
template <int LPC, int MODE, int BLOCK, bool PACKED = false>
__global__ __launch_bounds__(BLOCK, (BLOCK == 256 && !(PACKED && LPC > WAVE)) ? POSE_WAVES : 1)   // several wavefronts per character: LDS sets the occupancy
void k_pose(PoseArgs a)
{
    static_assert(!PACKED || MODE == 1, "the key-major pools are staged in LDS");
    constexpr int CPB = BLOCK / LPC;
    // joint globals, 4 KiB per wave; once a character's chain is done the same 4 KiB are the
    // wave's staging tile for its coalesced stores
    // slot LPC of every character is the identity with no ancestor: a lane that has reached the top of
    // its path keeps multiplying by it, so the jump rounds below have no divergent branch
    __shared__ __attribute__((aligned(16))) float g_lds[CPB][(LPC + 1) * G_STRIDE];
    constexpr bool LDS_TIMES = MODE >= 1;
    // key-major pools: LPC columns per key row, so the budget grows with the wavefronts per character
    __shared__ float times_lds[PACKED ? POSE_TIMES_LDS_MAX * (LPC / WAVE) : LDS_TIMES ? POSE_TIMES_LDS_MAX : 4];
    constexpr bool STREAM = MODE == 1 && (LPC == WAVE || PACKED);   // pose_stream_loop's instantiations
    __shared__ float4 jconst_lds[STREAM ? 5 * LPC : 1];          // per joint: the four columns of invmx, column 3 of bind
    __shared__ int anc_lds[STREAM ? CPB : 1][STREAM ? LPC + 4 : 1];    // the jump rounds' ancestor indices ([LPC] = -1: the identity slot's)

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

    if (cib < CPB && j == 0) {                                   // beyond the 4 KiB the store staging uses
        float4 *idn = reinterpret_cast<float4 *>(G) + 4 * LPC;   // (LPC >> 2) & 3 == 0: rows unswizzled
        idn[0] = make_float4(1.f, 0.f, 0.f, 0.f);
        idn[1] = make_float4(0.f, 1.f, 0.f, 0.f);
        idn[2] = make_float4(0.f, 0.f, 1.f, 0.f);
        idn[3] = make_float4(__int_as_float(-1), 0.f, 0.f, 0.f);
        if constexpr (STREAM) anc_lds[cib][LPC] = -1;
    }
    // first step of the key searches: the highest set bit of the model's longest channel (block-uniform)
    __shared__ uint32_t nr_or, not_streamable;
    if (tid == 0) { nr_or = 0; not_streamable = 0; }
    __syncthreads();
    {
        uint32_t m = 0, bad = 0;
        for (uint32_t q = tid; q < a.n_anims * J * 3; q += blockDim.x) {
            const uint32_t nr = a.chan_table[q].z;
            m |= nr;
            bad |= (int)nr <= 0;                                 // a path without a channel keeps its stored value: general loop
        }
        if (m) atomicOr(&nr_or, m);
        if (bad || (lane_joint && !reachable)) atomicOr(&not_streamable, 1u);
    }
    if (PACKED) {                                                // key-major times of every animation, then the key counts
        const uint32_t nt = a.n_anims * 3u * a.pk_kp * LPC + a.n_anims * 3u * LPC;
        for (uint32_t q = tid; q < nt; q += blockDim.x)
            times_lds[q] = a.pk_times[q];
    } else if (LDS_TIMES) {
        for (uint32_t q = tid; q < a.n_times; q += blockDim.x)
            times_lds[q] = a.times[q];
    }
    if (STREAM && tid < LPC) {
        const uint32_t jq = (uint32_t)tid < J ? (uint32_t)tid : J - 1;
#pragma unroll
        for (int q = 0; q < 4; q++) jconst_lds[q * LPC + tid] = a.invmx[4 * jq + q];
        jconst_lds[4 * LPC + tid] = a.bind[4 * jq + 3];
    }
    __syncthreads();
    // PACKED: the general loop below (taken when the skeleton is not streamable after all) searches the channel-major
    // times in memory; LDS holds the key-major copy
    const float *times = PACKED ? a.times : LDS_TIMES ? times_lds : a.times;
    const float *kdata = a.data;
    const int top = nr_or ? 1 << (31 - __clz((int)nr_or)) : 0;

    const uint32_t n_groups = (a.n_chars + CPB - 1) / CPB;

    // A character's inputs (animation id, frame time, the joint's three channel records) are requested
    // one character AHEAD, before the previous character's stores are issued: on gfx9-family hardware
    // loads and stores retire through one in-order counter (vmcnt), so a load issued behind 6.5 KB of
    // stores cannot be consumed until HBM has acknowledged them, while a load issued in front of them can.
    struct CharIn { uint32_t an; float time; uint4 e0, e1, e2; };
    const uint32_t jc = (uint32_t)j < J ? (uint32_t)j : J - 1;   // clamped: the loads below are always in bounds
    auto request = [&](uint32_t g_) {
        CharIn in;
        uint32_t c_ = g_ * CPB + (cib < CPB ? cib : 0);
        c_ = c_ < a.n_chars ? c_ : a.n_chars - 1;                 // past the end: a valid, unused character
        uint32_t an = a.anim[c_];
        if (an >= a.n_anims) an = 0;                             // an id outside the table would be a wild read
        in.an = an;
        in.time = a.frame_time[c_];
        const uint4 *tab = a.chan_table + ((size_t)an * J + jc) * 3;
        in.e0 = tab[0]; in.e1 = tab[1]; in.e2 = tab[2];          // (time_off, data_off, nr, -)
        return in;
    };
    CharIn cur = request(blockIdx.x);

    // the lane's joint constants, once per (persistent) block: 20 registers that three waves per SIMD afford
    float IM[16], bv[4];
    {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = a.invmx[4 * jc + q];
            IM[4 * q] = v.x; IM[4 * q + 1] = v.y; IM[4 * q + 2] = v.z; IM[4 * q + 3] = v.w;
        }
        const float4 b3 = a.bind[4 * jc + 3];                     // only column 3 of bind reaches mpos
        bv[0] = b3.x; bv[1] = b3.y; bv[2] = b3.z; bv[3] = b3.w;
    }

    if constexpr (STREAM) {
        // One wavefront per character, every joint animated on all three paths and under joint 0, outputs within
        // 2 GB: the loop below, in which the wavefront never waits for its own stores.  (A block's four wavefronts
        // share nothing but the key times; there is no block barrier past this point.)
        const uint64_t out_bytes = (uint64_t)a.n_chars * J * 64u;
        if (!not_streamable && out_bytes < (1ull << 31)) {
            pose_stream_loop<LPC, CPB, PACKED>(a, G, anc_lds[cib < CPB ? cib : 0], PACKED ? times_lds : times, kdata, top, cur.e0,
                                               cur.e1, cur.e2, jconst_lds, lane_joint ? parent : -1, j, cib);
            return;
        }
    }
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint32_t c = g * CPB + cib;
        const bool char_ok = cib < CPB && c < a.n_chars;
        const bool joint_ok = char_ok && lane_joint;
        const size_t cj = (size_t)c * J + j;

        // ---- 1. channels_transform: this joint's T, R, S at the character's frame time ----
        float T[3] = { 0, 0, 0 }, R[4] = { 0, 0, 0, 1 }, S[3] = { 1, 1, 1 };
        if (joint_ok) {
            const float time = cur.time;
            const uint4 e0 = cur.e0, e1 = cur.e1, e2 = cur.e2;
            const int n0 = (int)e0.z, n1 = (int)e1.z, n2 = (int)e2.z;
            if (n0 <= 0 || n1 <= 0 || n2 <= 0) {                 // a path without a channel keeps its value
                const float *st = a.trs + 10 * cj;
                T[0] = st[0]; T[1] = st[1]; T[2] = st[2];
                R[0] = st[3]; R[1] = st[4]; R[2] = st[5]; R[3] = st[6];
                S[0] = st[7]; S[1] = st[8]; S[2] = st[9];
            }
            // one path at a time (the searches run on LDS-resident times); the three paths' searches and key
            // loads interleaved in one block measured slower both at 128 VGPRs (205 us against 158) and at
            // 146 (141.6 against 135.1)
            if (n0 > 0) {
                int p, q;
                const float *t = times + e0.x;
                key_bracket(t, n0, time, top, p, q);
                const float fac = key_fac(time, t[p], t[q]);
                const float *d = kdata + e0.y;
                const key3 ka = *reinterpret_cast<const key3 *>(d + 3 * p), kb = *reinterpret_cast<const key3 *>(d + 3 * q);
                const LerpFac lf = lerp_fac(fac);
                T[0] = lerp_ref(ka.x, kb.x, lf); T[1] = lerp_ref(ka.y, kb.y, lf); T[2] = lerp_ref(ka.z, kb.z, lf);
            }
            if (n1 > 0) {
                int p, q;
                const float *t = times + e1.x;
                key_bracket(t, n1, time, top, p, q);
                const float fac = key_fac(time, t[p], t[q]);
                const float *d = kdata + e1.y;
                const key4 ka = *reinterpret_cast<const key4 *>(d + 4 * p), kb = *reinterpret_cast<const key4 *>(d + 4 * q);
                const float qa[4] = { ka.x, ka.y, ka.z, ka.w };
                const float qb[4] = { kb.x, kb.y, kb.z, kb.w };
                slerp_ref(R, qa, qb, fac);
            }
            if (n2 > 0) {
                int p, q;
                const float *t = times + e2.x;
                key_bracket(t, n2, time, top, p, q);
                const float fac = key_fac(time, t[p], t[q]);
                const float *d = kdata + e2.y;
                const key3 ka = *reinterpret_cast<const key3 *>(d + 3 * p), kb = *reinterpret_cast<const key3 *>(d + 3 * q);
                const LerpFac lf = lerp_fac(fac);
                S[0] = lerp_ref(ka.x, kb.x, lf); S[1] = lerp_ref(ka.y, kb.y, lf); S[2] = lerp_ref(ka.z, kb.z, lf);
            }
        }

        // ---- 2. one_joint_transform (model.c:1352-1404): global_j = root_pose * L_0 * ... * L_j over the
        // joint's ancestor path, L = T * R * S.  The reference walks the tree top down; here every joint
        // folds its path by pointer jumping: after step s a lane holds the product of the last 2^s locals
        // of its path and the index of the ancestor 2^s above, so ceil(log2(levels)) LDS rounds replace
        // `levels` dependent ones.  Locals are affine, so the running products are kept as three rows
        // (the fourth is 0 0 0 1); only root_pose and invmx are treated as general 4x4.
        Row M0, M1, M2;                                           // rows of the running product
        {
            // L = T * R * S: mat4x4_from_quat (linmath.h:959-987) with the scale folded into the columns
            const float qa = R[3], qb = R[0], qc = R[1], qd = R[2];
            const float a2 = qa * qa, b2 = qb * qb, c2 = qc * qc, d2 = qd * qd;
            const float bc = qb * qc, ad = qa * qd, bd = qb * qd, ac = qa * qc, cd = qc * qd, ab = qa * qb;
            const v2f s01 = { S[0], S[1] };
            M0.lo = v2f{ a2 + b2 - c2 - d2, 2.f * (bc - ad) } * s01;  M0.hi = v2f{ 2.f * (bd + ac) * S[2], T[0] };
            M1.lo = v2f{ 2.f * (bc + ad), a2 - b2 + c2 - d2 } * s01;  M1.hi = v2f{ 2.f * (cd - ab) * S[2], T[1] };
            M2.lo = v2f{ 2.f * (bd - ac), 2.f * (cd + ab) } * s01;    M2.hi = v2f{ (a2 - b2 - c2 + d2) * S[2], T[2] };
        }
        int anc = joint_ok ? parent : -1;
        {
            float4 *slots = reinterpret_cast<float4 *>(G);
            const int sw_me = (j >> 2) & 3;                       // row swizzle: 16 neighbouring lanes hit 64 banks
            for (uint32_t st = 0; st < a.n_jump_steps; st++) {
                slots[4 * j + (0 ^ sw_me)] = f4_of(M0);
                slots[4 * j + (1 ^ sw_me)] = f4_of(M1);
                slots[4 * j + (2 ^ sw_me)] = f4_of(M2);
                reinterpret_cast<int *>(&slots[4 * j + (3 ^ sw_me)])[0] = anc;
                if (LPC == WAVE) wave_lds_fence(); else __syncthreads();
                const int src = anc >= 0 ? anc : LPC;             // the identity slot once the path is folded
                const int sw = (src >> 2) & 3;
                const float4 A0 = slots[4 * src + (0 ^ sw)];
                const float4 A1 = slots[4 * src + (1 ^ sw)];
                const float4 A2 = slots[4 * src + (2 ^ sw)];
                anc = reinterpret_cast<const int *>(&slots[4 * src + (3 ^ sw)])[0];
                if (LPC == WAVE) wave_lds_fence(); else __syncthreads();
                const Row B0 = M0, B1 = M1, B2 = M2;
                M0 = affine_row(A0, B0, B1, B2);
                M1 = affine_row(A1, B0, B1, B2);
                M2 = affine_row(A2, B0, B1, B2);
            }
        }
        float Gm[16];                                             // global = root_pose * path product
        {
            const float4 m0 = f4_of(M0), m1 = f4_of(M1), m2 = f4_of(M2);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float p0 = a.root_pose[r], p1 = a.root_pose[4 + r], p2 = a.root_pose[8 + r], p3 = a.root_pose[12 + r];
                E_(Gm, 0, r) = p0 * m0.x + p1 * m1.x + p2 * m2.x;
                E_(Gm, 1, r) = p0 * m0.y + p1 * m1.y + p2 * m2.y;
                E_(Gm, 2, r) = p0 * m0.z + p1 * m1.z + p2 * m2.z;
                E_(Gm, 3, r) = p0 * m0.w + p1 * m1.w + p2 * m2.w + p3;
            }
        }

        // ---- 3. palette: joint_transforms = global * invmx; pos = e->mx * (joint_transforms * bind) * (0,0,0,1) ----
        float JT[16], pos[4] = { 0, 0, 0, 0 };
        if (joint_ok && reachable) {
#pragma unroll
            for (int cc = 0; cc < 4; cc++)                        // model.c:1389 (mat4x4_mul, contracted)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    E_(JT, cc, r) = E_(Gm, 0, r) * E_(IM, cc, 0) + E_(Gm, 1, r) * E_(IM, cc, 1) +
                                    E_(Gm, 2, r) * E_(IM, cc, 2) + E_(Gm, 3, r) * E_(IM, cc, 3);
            const bool with_pos = !(a.skip & CLAPGPU_POSE_SKIP_JOINT_POS);
            float mpos[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {                         // column 3 of JT * bind (model.c:1393-1397)
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 4; k++) s += E_(JT, k, r) * bv[k];
                mpos[r] = s;
            }
            if (with_pos) {
                const uint32_t ei = a.entity ? a.entity[c] : c;
                float EM[16];
                const float4 *em = reinterpret_cast<const float4 *>(a.entity_mx + 16 * (size_t)ei);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 v = em[q];
                    EM[4 * q] = v.x; EM[4 * q + 1] = v.y; EM[4 * q + 2] = v.z; EM[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int r = 0; r < 4; r++)                       // model.c:1400 (mat4x4_mul_vec4_post)
                    pos[r] = E_(EM, 0, r) * mpos[0] + E_(EM, 1, r) * mpos[1] + E_(EM, 2, r) * mpos[2] + E_(EM, 3, r) * mpos[3];
            }
        }

        cur = request(g + gridDim.x);                             // the next character's inputs, ahead of the stores

        // ---- stores (64 joints of one character per wave row) ----
        if (LPC != WAVE) __syncthreads();                         // every wave is done reading parents from G
        const uint32_t row_j0 = (uint32_t)(j - lane);             // first joint of this wave's row
        if (char_ok && row_j0 < J) {                              // wave-uniform
            const int nvalid = (int)(J - row_j0 < WAVE ? J - row_j0 : WAVE);
            const size_t row0 = (size_t)c * J + row_j0;
            float *tile_f = G + (row_j0 / WAVE) * (WAVE * G_STRIDE);      // the 4 KiB this wave's joints occupied
            float4 *tile = reinterpret_cast<float4 *>(tile_f);

            if (!(a.skip & CLAPGPU_POSE_SKIP_TRS)) {
                const float trs_row[10] = { T[0], T[1], T[2], R[0], R[1], R[2], R[3], S[0], S[1], S[2] };
                stage_rows<10>(tile_f, trs_row, lane);            // 2560 B
                wave_lds_fence();
                store_rows<10>(tile_f, a.trs + 10 * row0, lane, nvalid);
                wave_lds_fence();
            }

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
                if (lane < nvalid && !(a.skip & CLAPGPU_POSE_SKIP_JOINT_POS))
                    reinterpret_cast<float4 *>(a.joint_pos)[row0 + lane] = make_float4(pos[0], pos[1], pos[2], pos[3]);
                wave_lds_fence();
            } else if (joint_ok && reachable) {                   // joints not under joint 0 are never written
                float4 *dj = reinterpret_cast<float4 *>(a.joint_transforms + 16 * cj);
#pragma unroll
                for (int q = 0; q < 4; q++)
                    dj[q] = make_float4(JT[4 * q], JT[4 * q + 1], JT[4 * q + 2], JT[4 * q + 3]);
                if (!(a.skip & CLAPGPU_POSE_SKIP_JOINT_POS))
                    reinterpret_cast<float4 *>(a.joint_pos)[cj] = make_float4(pos[0], pos[1], pos[2], pos[3]);
            }
        }
        if (LPC != WAVE) __syncthreads();                         // next character reuses the globals in LDS
    }
}

// ---- key-major pools for the one-wavefront-per-character loop (clapgpu_animations_pack): once per model --------------
// layout of `packed`: times [n_anims][3][kp][L] f32 (+INF past a channel's last key) | key counts [n_anims][3][L] u32 |
// (16-byte aligned) values [n_anims][3][k][L] float4, L = the joints rounded up to whole wavefronts (64, 128, 192, 256).
// Columns past the last joint repeat the last joint's channels, as the loop's clamped joint index does.
__global__ __launch_bounds__(256)
void k_pose_pack(const uint4 *chan_table, const float *times, const float *data, uint32_t n_anims, uint32_t J, uint32_t kk,
                 uint32_t kp, uint32_t lanes, float *o_times, uint32_t *o_nr, float4 *o_vals)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;           // ((a * 3 + p) * kp + k) * lanes + lane
    const uint32_t lane = q % lanes, k = (q / lanes) % kp, ap = (q / lanes) / kp;
    if (ap >= n_anims * 3u) return;
    const uint32_t an = ap / 3u, p = ap % 3u, j = lane < J ? lane : J - 1;
    const uint4 e = chan_table[((size_t)an * J + j) * 3 + p];
    const int nr = (int)e.z > 0 ? (int)e.z : 0;
    o_times[q] = (int)k < nr ? times[e.x + k] : __builtin_inff();
    if (k == 0) o_nr[ap * lanes + lane] = (uint32_t)nr;
    if (k < kk) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((int)k < nr) {
            const float *d = data + e.y + (p == 1 ? 4u : 3u) * k;
            v = make_float4(d[0], d[1], d[2], p == 1 ? d[3] : 0.f);
        }
        o_vals[((size_t)ap * kk + k) * lanes + lane] = v;
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

static uint32_t pack_kp(uint32_t max_keys)
{
    uint32_t kp = 2;
    while (kp <= max_keys) kp <<= 1;                              // a power of two STRICTLY above the longest channel
    return kp;
}

static size_t pack_vals_offset(uint32_t n_anims, uint32_t kp, uint32_t lanes)
{
    const size_t head = ((size_t)n_anims * 3 * kp * lanes + (size_t)n_anims * 3 * lanes) * 4;
    return (head + 15) & ~(size_t)15;
}

extern "C" size_t clapgpu_animations_packed_bytes(uint32_t n_anims, uint32_t max_keys, uint32_t nr_joints)
{
    if (!n_anims || !max_keys || !nr_joints || nr_joints > 256) return 0;
    const uint32_t lanes = (nr_joints + 63) / 64 * 64;
    return pack_vals_offset(n_anims, pack_kp(max_keys), lanes) + (size_t)n_anims * 3 * max_keys * lanes * 16;
}

extern "C" int clapgpu_animations_pack(void *stream, const clapgpu_animations *an, uint32_t nr_joints, uint32_t max_keys,
                                       void *packed)
{
    if (!an || !packed || !an->chan_table || !an->times || !an->data || !an->n_anims || !max_keys)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (nr_joints == 0 || nr_joints > 256)                       // JOINTS_MAX is 200 (shader_constants.h:6)
        return CLAPGPU_ERR_TOO_LARGE;
    if ((reinterpret_cast<uintptr_t>(packed) & 15u) != 0)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const uint32_t kp = pack_kp(max_keys), lanes = (nr_joints + 63) / 64 * 64;
    char *base = static_cast<char *>(packed);
    float *o_times = reinterpret_cast<float *>(base);
    uint32_t *o_nr = reinterpret_cast<uint32_t *>(o_times + (size_t)an->n_anims * 3 * kp * lanes);
    float4 *o_vals = reinterpret_cast<float4 *>(base + pack_vals_offset(an->n_anims, kp, lanes));
    const uint32_t total = an->n_anims * 3u * kp * lanes;
    hipLaunchKernelGGL(k_pose_pack, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const uint4 *>(an->chan_table), an->times, an->data, an->n_anims, nr_joints, max_keys, kp,
                       lanes, o_times, o_nr, o_vals);
    CLAPGPU_LAUNCH_CHECK("k_pose_pack");
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
    if (!pb->anim || !pb->frame_time || !pb->trs || !pb->joint_transforms || (pb->joint_pos && !pb->entity_mx))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (pb->skip & ~(uint32_t)(CLAPGPU_POSE_SKIP_TRS | CLAPGPU_POSE_SKIP_JOINT_POS))
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
    a.skip = pb->skip | (pb->joint_pos ? 0u : (uint32_t)CLAPGPU_POSE_SKIP_JOINT_POS);

    const uint32_t lpc = (sk->nr_joints + 63) / 64 * 64;
    const bool lds_times = an->n_times > 0 && an->n_times <= (uint32_t)POSE_TIMES_LDS_MAX;
    // the key-major pools, if the caller made them (clapgpu_animations_pack) and every animation's rows fit in LDS
    a.pk_times = nullptr; a.pk_vals = nullptr; a.pk_k = a.pk_kp = 0;
    bool packed = false;
    if (an->packed && an->packed_keys) {
        const uint32_t kp = pack_kp(an->packed_keys);
        if ((uint64_t)a.n_anims * (3u * kp + 3u) * lpc <= (uint64_t)POSE_TIMES_LDS_MAX * (lpc / 64)) {
            packed = true;
            a.pk_times = static_cast<const float *>(an->packed);
            a.pk_vals = reinterpret_cast<const float4 *>(static_cast<const char *>(an->packed) + pack_vals_offset(a.n_anims, kp, lpc));
            a.pk_k = an->packed_keys; a.pk_kp = kp;
        }
    }
    hipStream_t s = as_stream(stream);
    static thread_local int n_cus = 0;
    if (!n_cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        CLAPGPU_HIP(hipGetDevice(&dev));
        CLAPGPU_HIP(hipGetDeviceProperties(&prop, dev));
        n_cus = prop.multiProcessorCount;
    }
    if (packed) {
        const uint32_t threads = lpc == 192 ? 192 : 256, cpb = threads / lpc;
        const uint32_t n_groups = (pb->n_chars + cpb - 1) / cpb;
        const void *fn = lpc == 64 ? (const void *)k_pose<64, 1, 256, true> : lpc == 128 ? (const void *)k_pose<128, 1, 256, true>
                       : lpc == 192 ? (const void *)k_pose<192, 1, 192, true> : (const void *)k_pose<256, 1, 256, true>;
        static thread_local uint32_t res_packed[4] = { 0, 0, 0, 0 };
        uint32_t &res = res_packed[lpc / 64 - 1];
        if (!res) {
            int per_cu = 0;
            CLAPGPU_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)threads, 0));
            res = (uint32_t)(per_cu > 0 ? per_cu : 1) * (uint32_t)n_cus;
        }
        const dim3 grid(n_groups < res ? n_groups : res), block(threads);
        switch (lpc) {
        case 64:  hipLaunchKernelGGL((k_pose<64, 1, 256, true>), grid, block, 0, s, a); break;
        case 128: hipLaunchKernelGGL((k_pose<128, 1, 256, true>), grid, block, 0, s, a); break;
        case 192: hipLaunchKernelGGL((k_pose<192, 1, 192, true>), grid, block, 0, s, a); break;
        default:  hipLaunchKernelGGL((k_pose<256, 1, 256, true>), grid, block, 0, s, a); break;
        }
    } else if (lds_times) {
        // persistent blocks (24 KiB key times + 16 KiB joint globals each): exactly as many as are
        // resident at once, so no block waits for a slot while the others hold their LDS copy
        const uint32_t threads = lpc == 192 ? 192 : 256, cpb = threads / lpc;
        const uint32_t n_groups = (pb->n_chars + cpb - 1) / cpb;
        const void *fn = lpc == 64 ? (const void *)k_pose<64, 1, 256> : lpc == 128 ? (const void *)k_pose<128, 1, 256>
                       : lpc == 192 ? (const void *)k_pose<192, 1, 192> : (const void *)k_pose<256, 1, 256>;
        static thread_local uint32_t resident[4] = { 0, 0, 0, 0 };
        uint32_t &res = resident[lpc / 64 - 1];
        if (!res) {
            int per_cu = 0;
            CLAPGPU_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)threads, 0));
            res = (uint32_t)(per_cu > 0 ? per_cu : 1) * (uint32_t)n_cus;
        }
        const dim3 grid(n_groups < res ? n_groups : res), block(threads);
        switch (lpc) {
        case 64:  hipLaunchKernelGGL((k_pose<64, 1, 256>), grid, block, 0, s, a); break;
        case 128: hipLaunchKernelGGL((k_pose<128, 1, 256>), grid, block, 0, s, a); break;
        case 192: hipLaunchKernelGGL((k_pose<192, 1, 192>), grid, block, 0, s, a); break;
        default:  hipLaunchKernelGGL((k_pose<256, 1, 256>), grid, block, 0, s, a); break;
        }
    } else {
        const uint32_t threads = lpc == 192 ? 192 : 256, cpb = threads / lpc;
        const dim3 grid((pb->n_chars + cpb - 1) / cpb), block(threads);
        switch (lpc) {
        case 64:  hipLaunchKernelGGL((k_pose<64, 0, 256>), grid, block, 0, s, a); break;
        case 128: hipLaunchKernelGGL((k_pose<128, 0, 256>), grid, block, 0, s, a); break;
        case 192: hipLaunchKernelGGL((k_pose<192, 0, 192>), grid, block, 0, s, a); break;
        default:  hipLaunchKernelGGL((k_pose<256, 0, 256>), grid, block, 0, s, a); break;
        }
    }
    CLAPGPU_LAUNCH_CHECK("k_pose");
    return CLAPGPU_OK;
}
