// pose.hip -- skeletal pose blend + joint-matrix palette for gfx950, in the reference's own arithmetic.
//
// Replaces, per animated entity, channels_transform() (model.c:1266-1350: keyframe bracket,
// lerp T/S, slerp R; interp.h:25-29, 59-118) and one_joint_transform() (model.c:1352-1404: global
// chain, joint_transforms = global * invmx, joint world position).  The host keeps
// animated_update()'s time base and queue logic (model.c:1563-1592) and passes each
// character's animation id and (float)frame_time.
//
// Numerics (round 4): every operation of the path is the reference's operation, in the reference's
// order, with the reference's roundings -- the file is compiled without FMA contraction like the
// rest of the library:
//   * key fraction: the fp32 quotient, correctly rounded (the compiler's IEEE division sequence);
//   * lerp (interp.h:25-29): (float)((double)a * (1.0 - (double)f) + (double)(b * f)) in fp64 on the device;
//   * slerp (interp.h:91-118): theta_0 = (float)acos(dot) and sin(theta_0) depend on the KEY PAIR alone, so
//     clapgpu_animations_pack() evaluates them once per model ON THE HOST with the host's libm -- the very
//     calls the reference makes -- and stores them per key interval; sin(theta) and cos(theta) of the frame
//     are fp64 polynomials on [0, pi/2].  MEASURED against glibc (tools/pose_exact_probe.c, round 5): (float)sin(theta)
//     is glibc's for EVERY float theta of [0, pi/2] (all 1 070 141 404 of them: exact by exhaustion); _rfac =
//     (float)(cos(theta) - u) cancels as fac -> 1 and rounds one float ulp differently in 27 of 10^10 slerps with fac
//     uniform in [0, 1] (the polynomial's cos differs from glibc's in its last fp64 bits for 0.32 % of the arguments):
//     at 3.2 M slerps a frame, one weight of one quaternion one ulp off every ~115 frames;
//     the two quotients by sin(theta_0) are fp64 products with its stored reciprocal, rounded to float: the fp32
//     quotient exactly (a quotient of two floats keeps 2^-49 away from every rounding boundary, the product errs by 2^-52);
//   * hierarchy: global[j] = ((global[parent] * T) * R) * S, evaluated level by level in THAT association
//     (model.c:1363-1383), then * invmx (model.c:1389), * bind's translation column, e->mx * (model.c:1392-1400).
// T / R / S, the palette and the joint positions therefore EQUAL the reference's, bit for bit, except where that one
// subtraction flips (2.7e-9 of the slerps): 0 of 3.2 M joints differ at BASELINE configs[2] in the tests' frames
// (tests/test_pose_skin_gpu.py, tools/pose_exact_check.py), and so do the skinned vertices.
// (Signed zeros included: the "0.f +" that opens mat4x4_mul's sums and turns a -0 sum into +0 is the zero addend of the
// v_pk_fma_f32 that forms the first product -- comb4<true>.)
//
// Mapping: one lane per joint, a 64-joint skeleton = one wavefront = one character for the keyframe work and the
// palette; the hierarchy runs as "level passes": the joints of one level, FOUR LANES EACH (one per column of the
// joint's global), so a level of up to LPC/4 joints is one pass of 16 packed multiply / add instructions whatever
// its width.  Globals and the joints' local columns live in LDS (never in HBM, as in the reference where
// `global` is scratch); keyframes are per model: times in LDS, values in L2.
// HBM: ~200 B / joint: T/R/S 40 B, joint_transforms 64 B, joint pos 16 B written.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "common.h"
#include "lm_dev.h"

namespace clapgpu {

struct PoseArgs {
    // skeleton
    uint32_t        J;
    const int32_t  *parent;
    const int32_t  *depth;
    const float    *root_pose;
    const float4   *invmx;
    const float4   *bind;
    // animations: the key-major pools of clapgpu_animations_pack(), L = J rounded up to whole wavefronts
    uint32_t        n_anims;
    const float    *pk_times;       // [anim][path][kp rows][L] key times, rows past a channel's last key +INF; then [anim][path][L] key counts
    const float4   *pk_vals;        // [anim][path][kk rows][L] key values (T / S: xyz, R: xyzw)
    const uint4    *pk_rc;          // [anim][kk rows][L] rotation interval constants (RotConst)
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
    uint32_t        prog_passes;    // level passes the dynamic LDS holds program words for (more: computed on the fly)
};

// What quat_slerp (interp.h:91-118) computes from the key pair (a, b) of one rotation interval alone, made by the host
// with the host's libm (clapgpu_animations_pack):
//   theta0        (float)acos((double)dot), dot = |quat_inner_product(a, b)|; -1 where dot > 0.9995 (quat_interp)
//   dot_flip      dot, its sign bit set where the inner product was negative (the reference then negates b)
//   inv_sin0      1.0 / (double)(float)sin((double)theta0)
struct RotConst { float theta0, dot_flip; double inv_sin0; };
static_assert(sizeof(RotConst) == 16, "one 16-byte load per lane");

typedef float v2f __attribute__((ext_vector_type(2)));

// ---- the reference's scalar arithmetic ------------------------------------------------------------------------------

// interp.h:25-29 linf_interp: a * (1.0 - blend) + b * blend with float a, b, blend -- the first product and the sum in
// double, b * blend a float product.  g = 1.0 - (double)blend.
__device__ __forceinline__ float lerp_ref(float a, float b, float blend, double g)
{
    const float bf = b * blend;
    const double t = (double)a * g;
    return (float)(t + (double)bf);
}

// sin and cos of x in [0, pi/2] in fp64: Taylor to x^21 / x^22 (|error| < 2 ulp of the double; rounded to float the
// results equal glibc's in 2 * 10^8 of 2 * 10^8 samples).  No range reduction: theta = fac * acos(dot), fac in [0, 1], dot >= 0.
// fma(a, b, c) with the coefficient c taken from an SGPR pair.  (Left to itself the compiler keeps the polynomials' 19
// coefficients in 38 VGPRs for the life of the kernel and copies one with v_mov_b64 in front of every v_fmac_f64.)
__device__ __forceinline__ double fma_coef(double a, double b, double c_uniform)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_uniform));
    return r;
}
// fma(c0, z, c1) with both coefficients from SGPR pairs (one scalar operand per VALU instruction: c0 goes through a move)
__device__ __forceinline__ double fma_coef2(double c0_uniform, double z, double c1_uniform)
{
    double r;
    asm("v_mov_b64 %0, %2\n\tv_fma_f64 %0, %0, %1, %3" : "=&v"(r) : "v"(z), "s"(c0_uniform), "s"(c1_uniform));
    return r;
}
__device__ __forceinline__ void sincos_halfpi(double x, double &sn, double &cs)
{
    const double z = x * x;
    double ps = fma_coef2(-1.9572941063391263e-20, z, 8.2206352466243295e-18);          // -1/21!, 1/19!
    ps = fma_coef(ps, z, -2.8114572543455206e-15);
    ps = fma_coef(ps, z, 7.6471637318198164e-13);
    ps = fma_coef(ps, z, -1.6059043836821613e-10);
    ps = fma_coef(ps, z, 2.5052108385441720e-08);
    ps = fma_coef(ps, z, -2.7557319223985893e-06);
    ps = fma_coef(ps, z, 1.9841269841269841e-04);
    ps = fma_coef(ps, z, -8.3333333333333332e-03);
    ps = fma_coef(ps, z, 1.6666666666666666e-01);
    sn = __builtin_fma(-(x * z), ps, x);
    double pc = fma_coef2(-8.8967913924505741e-22, z, 4.1103176233121648e-19);           // -1/22!, 1/20!
    pc = fma_coef(pc, z, -1.5619206968586225e-16);
    pc = fma_coef(pc, z, 4.7794773323873853e-14);
    pc = fma_coef(pc, z, -1.1470745597729725e-11);
    pc = fma_coef(pc, z, 2.0876756987868100e-09);
    pc = fma_coef(pc, z, -2.7557319223985888e-07);
    pc = fma_coef(pc, z, 2.4801587301587302e-05);
    pc = fma_coef(pc, z, -1.3888888888888889e-03);
    pc = fma_coef(pc, z, 4.1666666666666664e-02);
    pc = __builtin_fma(pc, z, -0.5);
    cs = __builtin_fma(z, pc, 1.0);
}
__device__ __forceinline__ void slerp_ref(float (&res)[4], const float (&a)[4], const float (&b_in)[4], float fac, const uint4 rcw)
{
    const float theta0 = __uint_as_float(rcw.x);
    const uint32_t flip = rcw.y & 0x80000000u;                   // dot < 0: b = -b, dot = -dot
    const float dot = __uint_as_float(rcw.y & 0x7fffffffu);
    float b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) b[i] = __uint_as_float(__float_as_uint(b_in[i]) ^ flip);
    if (theta0 < 0.0f) {                                         // dot > 0.9995: quat_interp's '+' branch (its own dot is >= 0), vec4_norm
        const float rfac = 1.f - fac;
        float t[4];
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = rfac * a[i] + fac * b[i];
        float p = t[0] * t[0];                                   // vec4_mul_inner: p = 0; p += b[i] * a[i]
        p += t[1] * t[1];
        p += t[2] * t[2];
        p += t[3] * t[3];
        const float len = sqrtf(p);                                // correctly rounded (__fsqrt_rn is the bare v_sqrt_f32: 1 ulp)
        // vec4_norm's k = 1.0 / len is a DOUBLE quotient rounded to float.  (A Newton / Markstein reciprocal in fp32 ties at
        // len = 1 - 2^-24, the length rounding gives nearly-unit quaternions half of the time, and rounds it to even.)
        const float k = (float)(1.0 / (double)len);
#pragma unroll
        for (int i = 0; i < 4; i++) res[i] = t[i] * k;
        return;
    }
    const float theta = fac * theta0;
    double sd, cd;
    sincos_halfpi((double)theta, sd, cd);
    const float sin_theta = (float)sd;
    const double inv_sin0 = __hiloint2double((int)rcw.w, (int)rcw.z);
    const float u = (float)((double)(dot * sin_theta) * inv_sin0);        // dot * sin_theta / sin_theta_0 in fp32
    const float rf = (float)(cd - (double)u);                             // cos(theta) is a double in the reference
    const float f = (float)((double)sin_theta * inv_sin0);
#pragma unroll
    for (int i = 0; i < 4; i++) res[i] = a[i] * rf + b[i] * f;             // quat_scale, quat_scale, quat_add
}

// One matrix column as two register pairs; out = ((A0 x + A1 y) + A2 z) + A3 w is mat4x4_mul's / mat4x4_mul_vec4_post's
// sum for one column (linmath.h:506-516, 297-305), with separately rounded products and sums: 4 v_pk_mul_f32 (or one
// v_pk_fma_f32 with a zero addend and 3 v_pk_mul_f32) + 3 v_pk_add_f32 per pair of rows.
struct Col { v2f lo, hi; };
__device__ __forceinline__ Col col_of(const float4 v) { Col r; r.lo = v2f{v.x, v.y}; r.hi = v2f{v.z, v.w}; return r; }
__device__ __forceinline__ float4 f4_of(const Col r) { return make_float4(r.lo.x, r.lo.y, r.hi.x, r.hi.y); }
// ZERO_FIRST: mat4x4_mul's "t = 0.f; t += ..." (linmath.h:506-516) -- the add that turns a -0 first product into +0, so that
// a sum of zeros comes out +0 as the reference's does; mat4x4_mul_vec4_post (linmath.h:297-305) has no such add.
template <bool ZERO_FIRST>
__device__ __forceinline__ Col comb4(const Col A0, const Col A1, const Col A2, const Col A3, float x, float y, float z, float w)
{
    Col o;
    if (ZERO_FIRST) {                                            // fma(a, x, +0) = round(a * x) with -0 turned into +0: "0.f + a * x" in one v_pk_fma_f32
        o.lo = __builtin_elementwise_fma(A0.lo, v2f{ x, x }, v2f{ 0.f, 0.f });
        o.hi = __builtin_elementwise_fma(A0.hi, v2f{ x, x }, v2f{ 0.f, 0.f });
    } else {
        o.lo = A0.lo * x; o.hi = A0.hi * x;
    }
    o.lo = o.lo + A1.lo * y; o.hi = o.hi + A1.hi * y;
    o.lo = o.lo + A2.lo * z; o.hi = o.hi + A2.hi * z;
    o.lo = o.lo + A3.lo * w; o.hi = o.hi + A3.hi * w;
    return o;
}

#ifndef POSE_WAVES_PER_SIMD
#define POSE_WAVES_PER_SIMD 3
#endif
#ifndef POSE_BLOCK64
#define POSE_BLOCK64 768
#endif
constexpr int POSE_WAVES = POSE_WAVES_PER_SIMD;   // wavefronts per SIMD the registers are budgeted for (168 VGPRs)
#ifndef POSE_TIMES_LDS_FLOATS
#define POSE_TIMES_LDS_FLOATS 6400             // (an occupancy experiment builds with less: profiles/r05_experiments/pose_occupancy.md)
#endif
constexpr int POSE_TIMES_LDS_MAX = POSE_TIMES_LDS_FLOATS;   // key times kept in LDS when the model's rows fit: 25 KiB per 64 lanes (one
                                             // animation of <= 31 keys per channel with its key counts)
constexpr int POSE_MAX_JOINTS = 256;

typedef int pose_v4i __attribute__((ext_vector_type(4)));
constexpr int POSE_RSRC_FLAGS = 0x00020000;                      // raw buffer, 32-bit data format
constexpr uint32_t POSE_CLIPPED = 0x7ffffff0u;                   // an offset past every descriptor's range: the store is dropped

// byte_off: the lane's offset; uniform_off: a wave-uniform part that travels as the instruction's scalar offset (one lane
// offset register then serves every 1 KiB piece of a row)
__device__ __forceinline__ void buffer_store4(const float4 v, __amdgpu_buffer_rsrc_t r, uint32_t byte_off, bool stream, uint32_t uniform_off = 0)
{
    const pose_v4i d = { __float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w) };
    if (stream) __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)byte_off, (int)uniform_off, 2);      // nt
    else __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)byte_off, (int)uniform_off, 0);
}

struct PoseKeys { float4 ta, tb, ra, rb, sa, sb; uint4 rc; float f0, f1, f2; uint32_t has; };

// The three paths' brackets for the key-major pools: key k of (path, joint j) sits at row k, column j.
//   * times [path][row][LPC]: lane j's read of ANY row lands in bank j % 32 of its half-wave -- the searches' ds_read_b32
//     are conflict-free whatever rows the lanes are at;
//   * rows past a channel's last key hold +INF, so the search needs no bounds test: lo = #{k : t[k] < time} by
//     log2(kp) (add, read, compare-select) steps, the three paths' reads independent of each other;
//   * model.c:1266-1288 for strictly increasing key times (glTF): the bracket does not depend on the reference's search
//     cursor; time before the first / past the last key wraps to (nr - 1, 0).  lo == nr <=> time > t[nr - 1], and with
//     lo == 0 the bracket's own first key IS t[0].
//   * values [path][row][LPC] float4, the rotation interval's constants [row = prev][LPC].
// has: bit p set where path p has a channel (MISSING: a path without one keeps its stored value, model.c:1301).
template <int LPC, bool MISSING, bool TIMES_LDS>
__device__ __forceinline__ PoseKeys pose_gather_keys(const float *tl, const float4 *vals, const uint4 *rc, const int kp, const int kk,
                                                     const float time, int n0, int n1, int n2, const int col)
{
    constexpr int COLS = LPC;
    PoseKeys k;
    k.has = (n0 > 0 ? 1u : 0u) | (n1 > 0 ? 2u : 0u) | (n2 > 0 ? 4u : 0u);
    if (MISSING) { n0 = n0 > 0 ? n0 : 1; n1 = n1 > 0 ? n1 : 1; n2 = n2 > 0 ? n2 : 1; }      // row 0 of an absent channel: +INF, value 0
    const float *t0 = tl + col, *t1 = t0 + kp * COLS, *t2 = t1 + kp * COLS;
    int l0 = 0, l1 = 0, l2 = 0;
    auto probe = [&](const int step) {
        const float a0 = t0[(l0 + step - 1) * COLS], a1 = t1[(l1 + step - 1) * COLS], a2 = t2[(l2 + step - 1) * COLS];
        l0 += a0 < time ? step : 0;
        l1 += a1 < time ? step : 0;
        l2 += a2 < time ? step : 0;
    };
    if (TIMES_LDS) {                                             // rows that fit in LDS: kp <= 32, the steps straight-line
#pragma unroll
        for (int sb = 4; sb >= 0; sb--)
            if ((1 << sb) < kp) probe(1 << sb);                  // uniform
    } else {
        for (int step = kp >> 1; step > 0; step >>= 1) probe(step);
    }
    // model.c:1266-1288 + 1312-1317 without a branch and without reading t[0] / t[nr - 1] again:
    //   * inside the keys: prev = lo - 1 (0 at lo == 0: time == t[0]), next = min(prev + 1, nr - 1), fac the quotient
    //     (0 where prev == next: one key, or time on the last key);
    //   * wrapped -- lo == nr (time past the last key) or lo == 0 with time < t[0] (t[prev] IS t[0] then): the pair is
    //     (nr - 1, 0), p_time > n_time for nr > 1, so fac = time < t[0] ? 1 : 0 -- which is "lo == 0" -- and 0 for nr == 1.
    auto finish = [&](const float *t, int nr, int lo, int &prev, int &next, float &fac) {
        lo = lo < nr ? lo : nr;                                      // MISSING's stand-in count
        const int pn = lo > 0 ? lo - 1 : 0;
        const int nn = pn + 1 < nr - 1 ? pn + 1 : nr - 1;
        const float tp = t[pn * COLS], tn = t[nn * COLS];
        const bool wrap = lo == nr || (lo == 0 && time < tp);
        const float q = (time - tp) / (tn - tp);                     // the IEEE quotient; tp == tn: a NaN nobody reads
        const float inside = tp < tn ? q : 0.f;
        const float outside = (lo == 0 && nr > 1) ? 1.f : 0.f;
        prev = wrap ? nr - 1 : pn;
        next = wrap ? 0 : nn;
        fac = wrap ? outside : inside;
    };
    int p0, q0, p1, q1, p2, q2;
    finish(t0, n0, l0, p0, q0, k.f0); finish(t1, n1, l1, p1, q1, k.f1); finish(t2, n2, l2, p2, q2, k.f2);
    // uniform base + unsigned 32-bit byte offset per lane: the loads take their base from SGPRs (no 64-bit vector adds)
    const float4 *v1 = vals + kk * COLS, *v2 = v1 + kk * COLS;
    auto row16 = [&](const void *base, int row) {
        const uint32_t off = (uint32_t)(row * COLS + col) * 16u;
        return *reinterpret_cast<const float4 *>(static_cast<const char *>(base) + off);
    };
    k.ta = row16(vals, p0); k.tb = row16(vals, q0);
    k.ra = row16(v1, p1);   k.rb = row16(v1, q1);
    k.sa = row16(v2, p2);   k.sb = row16(v2, q2);
    const float4 rcv = row16(rc, p1);
    k.rc = make_uint4(__float_as_uint(rcv.x), __float_as_uint(rcv.y), __float_as_uint(rcv.z), __float_as_uint(rcv.w));
    return k;
}

// XOR swizzle of a joint slot's four 16-byte columns: the stores of the eight lanes ds_write_b128 services at a time and
// the reads of the sixteen lanes ds_read_b128 services at a time spread over the banks
__device__ __forceinline__ int slot_swz(int j) { return (j >> 2) & 3; }

// LDS hand-over between the wavefronts of one character (skeletons of more than 64 joints: 2-4 wavefronts each): the LDS
// counter only -- __syncthreads() would also wait for the wavefront's global stores, the very wait this loop exists to avoid
// One wavefront per character: its DS operations execute in issue order, so a read issued after a write sees it -- only
// the COMPILER has to be kept from moving them across each other (no s_waitcnt: the level passes would otherwise wait
// for each pass's store to be acknowledged before issuing the next pass's reads).
template <int LPC>
__device__ __forceinline__ void pose_lds_sync()
{
    if (LPC == WAVE) { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The level passes' per-lane program: for pass p the lane with role (entry q, column c) reads its joint's local column
// -- at the same place of the character's LOC half as the global column it writes has in the G half -- and the parent's
// four columns at pa ^ 16 k: one word (own column | pa << 16, byte offsets inside the character's G half) that depends on
// the skeleton alone, so that a pass spends 8 instead of 15 instructions on addresses.  Idle lanes work on the scratch
// slot.
template <int LPC>
__device__ __forceinline__ uint32_t pose_prog_word(const uint2 *passes, const uint32_t *order, int p, int q, int c)
{
    const uint2 pr = passes[p];
    const uint32_t e = (uint32_t)q < pr.y ? order[pr.x + q] : ((uint32_t)(LPC + 1) | ((uint32_t)LPC << 16));
    const uint32_t jj = e & 0xffffu, pp = e >> 16;
    return 16u * (4u * jj + ((uint32_t)c ^ (uint32_t)slot_swz((int)jj))) | (16u * (4u * pp + (uint32_t)slot_swz((int)pp))) << 16;
}

// ---- the loop: one wavefront per 64 joints, no wavefront ever waits for its own stores -------------------------------
// gfx950 retires vector loads and stores through ONE counter, in issue order: waiting for a load means waiting for
// every store issued before it, and the compiler can only wait for "all but the last N operations" when it can count
// the operations behind the load on every path.  So:
//   * every store is a buffer store issued by all 64 lanes under no branch; rows shorter than 64 joints, characters
//     past the end, masked outputs and joints outside joint 0's tree are clipped by the descriptor's size (range
//     check per dword), not by exec;
//   * per-character scalars (animation id, frame time, root pose) come through the scalar cache, a character ahead;
//   * the order per character g is: interpolate g's keys -> search + gather the keys of g+1 -> hierarchy and palette
//     of g -> stores of g.  Every wait is then for an operation with a known number of younger ones behind it, and
//     the gathers of g+1 fly under g's level passes.
// LPC = lanes per character (64, 128, 192, 256); BLOCK threads = CPB characters per block.  MISSING: some (joint, path)
// has no channel (its stored T / R / S is read back).  TIMES_LDS: the model's key times are staged in LDS once per
// (persistent) block; otherwise the searches read them through L2.
template <int LPC, int BLOCK, bool MISSING, bool TIMES_LDS>
__global__ __launch_bounds__(BLOCK, (LPC == WAVE) ? POSE_WAVES : 1)
void k_pose(PoseArgs a)
{
    constexpr int CPB = BLOCK / LPC;
    constexpr int SPP = LPC / 4;                                 // joints per level pass: four lanes each
    // per character, the G half: the joints' globals, four 16-byte columns each (XOR-swizzled), + the root pose's slot
    // [LPC] + the idle lanes' scratch slot [LPC + 1]; each wavefront's own 4 KiB of it is afterwards the staging tile of
    // its stores.  The LOC half, the same slots HALF bytes on: the joints' local columns (R column c, scale c) for c < 3
    // and (translation, 1).  A character's halves start on a 64-byte boundary (pa ^ 16 k addresses the parent's columns).
    constexpr int HALF_SLOTS = (LPC + 2) * 4;
    constexpr uint32_t HALF = HALF_SLOTS * 16u;
    static_assert(HALF % 64u == 0 && 2u * HALF < 65536u, "a character's halves: 64-byte aligned, LOC within a DS offset of G");
    __shared__ __attribute__((aligned(64))) float4 gl_lds[CPB][2 * HALF_SLOTS];
    __shared__ float times_lds[TIMES_LDS ? POSE_TIMES_LDS_MAX * (LPC / WAVE) : 4];
    __shared__ float4 jconst_lds[5 * LPC];                       // per joint: the four columns of invmx, column 3 of bind
    // level passes: order[] = the joints reachable from joint 0 in pass order as joint | parent slot << 16,
    // passes[p] = (first entry, entries).  A pass holds up to SPP joints whose parents were done in EARLIER passes: the
    // joints of a level, topped up with joints of the next level whose parents are already done (joints WITH children
    // are taken first, so that what a level leaves over are leaves) -- configs[2]'s skeleton: 8 passes, not 9
    __shared__ uint32_t order_lds[POSE_MAX_JOINTS];
    __shared__ uint2 passes_lds[POSE_MAX_JOINTS];
    __shared__ int16_t depth_lds[POSE_MAX_JOINTS];
    __shared__ int16_t done_pass[POSE_MAX_JOINTS + 1];          // the pass a joint is computed in (-1: not yet); [J..]: -1
    __shared__ uint32_t has_child[POSE_MAX_JOINTS];
    __shared__ uint32_t n_passes_s;
    extern __shared__ uint32_t prog_lds[];                       // [prog_passes][LPC] program words (dynamic: sized by the host from n_levels)

    const int tid = threadIdx.x;
    const int cib = tid / LPC, j = tid % LPC;                    // character in block, joint (BLOCK is a multiple of LPC)
#ifdef CLAPGPU_POSE_PROF                                         // tools/build_variant.sh prof pose.hip -DCLAPGPU_POSE_PROF: block 0 prints its phases
    unsigned long long pt[8]; int pn = 0;                        // (10 ns units): setup, program, first gather, characters 1-3, the rest
#define PT() do { pt[pn++] = wall_clock64(); } while (0)
    PT();
#else
#define PT() do {} while (0)
#endif
#ifdef CLAPGPU_POSE_PROF_ITER                                    // the phases of ONE character (the block's fourth) by s_memtime (core clocks)
    unsigned long long qt[12]; int qn = 0; int iter_no = 0;
#define QT() do { if (iter_no == 3 && qn < 12) qt[qn++] = __builtin_readcyclecounter(); } while (0)
#else
#define QT() do {} while (0)
#endif
    const int lane = lane_id();
    const uint32_t J = a.J;
    const int kp = (int)a.pk_kp, kk = (int)a.pk_k;

    // ---- once per (persistent) block: the first wavefront schedules the level passes while the others fill the tables ----
    if (j == 0) {
        float4 *root = &gl_lds[cib][4 * LPC];                    // slot_swz(LPC) == 0: columns unswizzled
#pragma unroll
        for (int q = 0; q < 4; q++)
            root[q] = make_float4(a.root_pose[4 * q], a.root_pose[4 * q + 1], a.root_pose[4 * q + 2], a.root_pose[4 * q + 3]);
    }
    if (tid < WAVE) {
        // up to four joints per lane (tid, tid + 64, ...), their state in registers; what a pass reads from LDS is only
        // "when was my parent done".  DS operations of one wavefront execute in issue order: no barrier inside.
        constexpr int ROWS = POSE_MAX_JOINTS / WAVE;
        const int rows = (int)((J + WAVE - 1) / WAVE);           // uniform
        int par_r[ROWS];
        bool live_r[ROWS], hc_r[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const int jj = tid + WAVE * r;
            const uint32_t jq = (uint32_t)jj < J ? (uint32_t)jj : J - 1;   // loads under no lane test: all in flight together
            const int d_ld = a.depth[jq];
            const int32_t par_ld = a.parent[jq];
            const int d = (uint32_t)jj < J ? d_ld : -1;
            const int32_t par = (uint32_t)jj < J ? par_ld : -1;
            par_r[r] = (par < 0 || par >= (int32_t)J) ? -1 : par;
            live_r[r] = d >= 0;                                    // reachable and not yet scheduled
            depth_lds[jj] = (int16_t)d;
            has_child[jj] = 0;
            done_pass[jj] = -1;
        }
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < ROWS; r++)
            if (live_r[r] && par_r[r] >= 0) atomicOr(&has_child[par_r[r]], 1u);
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < ROWS; r++) hc_r[r] = has_child[tid + WAVE * r] != 0;
        const uint64_t below = (1ull << tid) - 1ull;
        uint32_t at = 0, np = 0;
        for (;;) {
            bool ready_r[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                ready_r[r] = false;
                if (r < rows) {                                  // uniform
                    const int dp = par_r[r] >= 0 ? done_pass[par_r[r]] : 0;
                    ready_r[r] = live_r[r] && (par_r[r] < 0 || (dp >= 0 && dp < (int)np));
                }
            }
            uint32_t taken = 0;
#pragma unroll
            for (int cls = 1; cls >= 0; cls--)                   // joints with children first, then leaves; rows in order
#pragma unroll
                for (int r = 0; r < ROWS; r++)
                    if (r < rows) {                              // uniform
                        const bool mine = ready_r[r] && (int)hc_r[r] == cls;
                        const uint64_t m = __ballot(mine);
                        const uint32_t rank = taken + (uint32_t)__popcll(m & below);
                        if (mine && rank < (uint32_t)SPP) {
                            const int jj = tid + WAVE * r;
                            order_lds[at + rank] = (uint32_t)jj | ((uint32_t)(par_r[r] < 0 ? LPC : par_r[r]) << 16);
                            done_pass[jj] = (int16_t)np;
                            live_r[r] = false;
                        }
                        taken += (uint32_t)__popcll(m);
                    }
            const uint32_t cnt = taken < (uint32_t)SPP ? taken : (uint32_t)SPP;
            if (!cnt) break;                                     // everything reachable is scheduled
            if (tid == 0) passes_lds[np] = make_uint2(at, cnt);
            at += cnt;
            np++;
            wave_lds_fence();
        }
        if (tid == 0) n_passes_s = np;
    } else {
        const int t1 = tid - WAVE, nt1 = BLOCK - WAVE;
        if (TIMES_LDS) {                                         // key-major times of every animation, then the key counts
            const uint32_t nt = a.n_anims * 3u * a.pk_kp * LPC + a.n_anims * 3u * LPC;
            for (uint32_t q = t1; q < nt; q += nt1)
                times_lds[q] = a.pk_times[q];
        }
        for (int q = t1; q < LPC; q += nt1) {
            const uint32_t jq = (uint32_t)q < J ? (uint32_t)q : J - 1;
#pragma unroll
            for (int k = 0; k < 4; k++) jconst_lds[k * LPC + q] = a.invmx[4 * jq + k];
            jconst_lds[4 * LPC + q] = a.bind[4 * jq + 3];
        }
    }
    __syncthreads();
    PT();
    const int n_passes = (int)n_passes_s;
    const int prog_passes = n_passes < (int)a.prog_passes ? n_passes : (int)a.prog_passes;
    for (int q = tid; q < prog_passes * LPC; q += BLOCK)
        prog_lds[q] = pose_prog_word<LPC>(passes_lds, order_lds, q / LPC, (q % LPC) >> 2, q & 3);
    __syncthreads();
    PT();

    // ---- per lane ------------------------------------------------------------------------------------------------------
    float4 *G = gl_lds[cib];
    float4 *LOC = G + HALF_SLOTS;
    const uint32_t char_off = (uint32_t)cib * 2u * HALF;       // this character's G half, in bytes from gl_lds
    const float *times = TIMES_LDS ? times_lds : a.pk_times;
    const uint32_t *nr_tab = reinterpret_cast<const uint32_t *>(times + (size_t)a.n_anims * 3 * kp * LPC);
    const bool reachable = (uint32_t)j < J && depth_lds[j < POSE_MAX_JOINTS ? j : 0] >= 0;
    const uint64_t reach_row = __ballot(reachable);              // this wavefront's 64-joint row
    const int row_j0 = j - lane;                                 // first joint of this wavefront's row
    // Character of (round it, block b, character slot cib): it * (blocks * CPB) + cib * blocks + b -- the characters of the
    // last, partial round are spread over ALL blocks, a few wavefronts each, instead of filling some blocks and leaving the
    // others idle (a round with 4 of 12 wavefronts busy is through sooner than a full one)
    const uint32_t per_round = gridDim.x * (uint32_t)CPB;
    const uint32_t n_rounds = (a.n_chars + per_round - 1) / per_round;
    const bool with_trs = !(a.skip & CLAPGPU_POSE_SKIP_TRS), with_pos = !(a.skip & CLAPGPU_POSE_SKIP_JOINT_POS);
    const bool pos_world = with_pos && !(a.skip & CLAPGPU_POSE_JOINT_POS_MODEL);   // e->mx * mpos here, or later (clapgpu_joint_pos_world)
    const uint32_t cib_u = (uint32_t)__builtin_amdgcn_readfirstlane(cib);
    const uint32_t jc = (uint32_t)j < J ? (uint32_t)j : J - 1;
    // role in the level passes: joint entry q of the pass, column c of its global
    const int pq = j >> 2, pc = j & 3;
    const float v3 = pc == 3 ? 1.f : 0.f;

    // Per-character scalars (animation, frame time, entity): lane L of a wavefront holds those of the character the
    // wavefront works on L iterations after `it0` -- three vector loads per 64 iterations -- and an iteration takes its
    // own with v_readlane.  (Scalar loads would share lgkmcnt with the LDS operations: every wait for an LDS result in
    // the loop would also wait for the scalar cache's miss to L2 -- 4 us of the 94 of a palette-only launch.)
    uint32_t v_an = 0, v_ent = 0;
    float v_tm = 0.f;
    auto load_scalars = [&](uint32_t it0) {
        uint64_t c64 = (uint64_t)(it0 + (uint32_t)lane) * per_round + (uint64_t)cib_u * gridDim.x + blockIdx.x;
        const uint32_t cL = c64 < a.n_chars ? (uint32_t)c64 : a.n_chars - 1;     // past the end: a valid character whose stores are clipped
        const uint32_t an = a.anim[cL];
        v_an = an < a.n_anims ? an : 0u;
        v_tm = a.frame_time[cL];
        v_ent = a.entity ? a.entity[cL] : cL;
    };
    auto lane_u32 = [&](uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); };
    auto lane_f32 = [&](float v, uint32_t l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)l)); };
    auto gather = [&](uint32_t an, float tm) {
        const uint32_t *nr = nr_tab + an * 3 * LPC + j;
        return pose_gather_keys<LPC, MISSING, TIMES_LDS>(times + (size_t)an * 3 * kp * LPC, a.pk_vals + (size_t)an * 3 * kk * LPC,
                                              a.pk_rc + (size_t)an * kk * LPC, kp, kk, tm, (int)nr[0], (int)nr[LPC], (int)nr[2 * LPC], j);
    };

    uint32_t it = 0;
    load_scalars(0);
    // the character's entity matrix: element (lane & 15) per lane, one vector load a character ahead, read back with
    // v_readlane where joint positions are formed
    const uint32_t em_lane_off = (uint32_t)(lane & 15) * 4u;
    auto load_em = [&](uint32_t ei) {                             // (a buffer load: the 64-byte matrix is the descriptor's range, its base scalar)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.entity_mx) + 16 * (size_t)ei, 0, 64, POSE_RSRC_FLAGS);
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)em_lane_off, 0, 0));
    };
    float em_v = pos_world ? load_em(lane_u32(v_ent, 0)) : 0.f;
    // the first character's keys -- waited for HERE, so that no wait for them is left pending into the loop, where it
    // would stand for "all but a few operations" on the way round
    PoseKeys kv = gather(lane_u32(v_an, 0), lane_f32(v_tm, 0));
    asm volatile("" : : "v"(kv.ta.x), "v"(kv.tb.x), "v"(kv.ra.x), "v"(kv.rb.x), "v"(kv.sa.x), "v"(kv.sb.x), "v"(kv.rc.x), "v"(em_v));
    PT();

    for (; it < n_rounds; it++) {
        const uint64_t c_raw = (uint64_t)it * per_round + (uint64_t)cib_u * gridDim.x + blockIdx.x;
        const bool c_ok = c_raw < a.n_chars;                     // wave-uniform
        const uint32_t c = c_ok ? (uint32_t)c_raw : a.n_chars - 1;
        // the scalars of the character after this one
        const uint32_t nx = (it + 1) & (uint32_t)(WAVE - 1);
        if (nx == 0) load_scalars(it + 1);                       // uniform: once per 64 iterations
        const uint32_t an_next = lane_u32(v_an, nx), ei_next = lane_u32(v_ent, nx);
        const float tm_next = lane_f32(v_tm, nx);

        QT();
#ifdef CLAPGPU_POSE_PROF_ITER                                    // how long the keys gathered a character ago are still waited for
        asm volatile("" : : "v"(kv.ta.x), "v"(kv.tb.x), "v"(kv.ra.x), "v"(kv.rb.x), "v"(kv.sa.x), "v"(kv.sb.x), "v"(kv.rc.x));
        QT();
#endif
        // ---- 1. channels_transform: this character's T, R, S from its keys (model.c:1290-1350)
        float T[3], R[4], S[3];
        {
            const double g0 = 1.0 - (double)kv.f0, g2 = 1.0 - (double)kv.f2;
            T[0] = lerp_ref(kv.ta.x, kv.tb.x, kv.f0, g0); T[1] = lerp_ref(kv.ta.y, kv.tb.y, kv.f0, g0); T[2] = lerp_ref(kv.ta.z, kv.tb.z, kv.f0, g0);
#ifdef CLAPGPU_POSE_PROF_ITER
            asm volatile("" : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]) : : "memory");
            QT();
#endif
            const float qa[4] = { kv.ra.x, kv.ra.y, kv.ra.z, kv.ra.w };
            const float qb[4] = { kv.rb.x, kv.rb.y, kv.rb.z, kv.rb.w };
            slerp_ref(R, qa, qb, kv.f1, kv.rc);
#ifdef CLAPGPU_POSE_PROF_ITER
            asm volatile("" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]) : : "memory");
            QT();
#endif
            S[0] = lerp_ref(kv.sa.x, kv.sb.x, kv.f2, g2); S[1] = lerp_ref(kv.sa.y, kv.sb.y, kv.f2, g2); S[2] = lerp_ref(kv.sa.z, kv.sb.z, kv.f2, g2);
        }
        if (MISSING) {                                           // a path without a channel keeps its value (model.c:1301)
            const float *st = a.trs + 10 * ((size_t)c * J + jc);
            if (!(kv.has & 1u)) { T[0] = st[0]; T[1] = st[1]; T[2] = st[2]; }
            if (!(kv.has & 2u)) { R[0] = st[3]; R[1] = st[4]; R[2] = st[5]; R[3] = st[6]; }
            if (!(kv.has & 4u)) { S[0] = st[7]; S[1] = st[8]; S[2] = st[9]; }
        }

#ifdef CLAPGPU_POSE_PROF_ITER
        asm volatile("" : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(S[0]), "+v"(S[1]), "+v"(S[2]) : : "memory");
#endif
        QT();
        // ---- 2. the next character's key search (LDS) and key gathers, in flight under the level passes below
        kv = gather(an_next, tm_next);

        QT();
        // ---- 3. one_joint_transform (model.c:1352-1404).  The joint's local columns: R = mat4x4_from_quat(rotation)
        // (linmath.h:959-987), column c with scale[c] beside it; the translation with 1 beside it.
        {
            float Rm[16];
            lmd::from_quat(Rm, R[0], R[1], R[2], R[3]);
            const int sw = slot_swz(j);
            LOC[4 * j + (0 ^ sw)] = make_float4(E_(Rm, 0, 0), E_(Rm, 0, 1), E_(Rm, 0, 2), S[0]);
            LOC[4 * j + (1 ^ sw)] = make_float4(E_(Rm, 1, 0), E_(Rm, 1, 1), E_(Rm, 1, 2), S[1]);
            LOC[4 * j + (2 ^ sw)] = make_float4(E_(Rm, 2, 0), E_(Rm, 2, 1), E_(Rm, 2, 2), S[2]);
            LOC[4 * j + (3 ^ sw)] = make_float4(T[0], T[1], T[2], 1.0f);
        }
        pose_lds_sync<LPC>();                                    // (also: every wavefront is past the previous character's staging reads)
        QT();
        // Level passes.  global = ((parent * T) * R) * S column by column: with P the parent's global,
        //   column c < 3:  ((P0 R[c][0] + P1 R[c][1]) + P2 R[c][2]) + P3' * 0, then * scale[c]   (mat4x4_mul by R, mat4x4_scale_aniso)
        //   column 3:      ((P0 tx + P1 ty) + P2 tz) + P3 * 1                                  (mat4x4_mul by T; R and S leave it alone)
        // -- the products with the 0s and 1s of T and R that are left out are exact, the sums with their zeros too.
        auto level_pass = [&](const uint32_t w) {
            char *const base = reinterpret_cast<char *>(&gl_lds[0][0]);
            const uint32_t own = char_off + (w & 0xffffu), pa = char_off + (w >> 16);
            const float4 v = *reinterpret_cast<const float4 *>(base + own + HALF);
            const Col P0 = col_of(*reinterpret_cast<const float4 *>(base + pa)), P1 = col_of(*reinterpret_cast<const float4 *>(base + (pa ^ 16u)));
            const Col P2 = col_of(*reinterpret_cast<const float4 *>(base + (pa ^ 32u))), P3 = col_of(*reinterpret_cast<const float4 *>(base + (pa ^ 48u)));
            Col o = comb4<true>(P0, P1, P2, P3, v.x, v.y, v.z, v3);
            o.lo = o.lo * v.w; o.hi = o.hi * v.w;
            *reinterpret_cast<float4 *>(base + own) = f4_of(o);
            pose_lds_sync<LPC>();
        };
        {
            uint32_t w = prog_lds[j];
            for (int p = 0; p < prog_passes; p++) {
                const uint32_t w_now = w;
                w = prog_lds[(p + 1 < prog_passes ? p + 1 : p) * LPC + j];       // the next pass's word: in flight under this pass's arithmetic
                level_pass(w_now);
            }
            for (int p = prog_passes; p < n_passes; p++)                        // a skeleton deeper than the host was told
                level_pass(pose_prog_word<LPC>(passes_lds, order_lds, p, pq, pc));
        }
        QT();
        // joint_transforms = global * invmx (model.c:1389); mpos = column 3 of joint_transforms * bind (model.c:1392-1397);
        // pos = e->mx * mpos (model.c:1400)
        Col JT0, JT1, JT2, JT3, POS;
        {
            const int sw = slot_swz(j);
            const Col G0 = col_of(G[4 * j + (0 ^ sw)]), G1 = col_of(G[4 * j + (1 ^ sw)]);
            const Col G2 = col_of(G[4 * j + (2 ^ sw)]), G3 = col_of(G[4 * j + (3 ^ sw)]);
            const float4 i0 = jconst_lds[0 * LPC + j], i1 = jconst_lds[1 * LPC + j], i2 = jconst_lds[2 * LPC + j], i3 = jconst_lds[3 * LPC + j];
            JT0 = comb4<true>(G0, G1, G2, G3, i0.x, i0.y, i0.z, i0.w);
            JT1 = comb4<true>(G0, G1, G2, G3, i1.x, i1.y, i1.z, i1.w);
            JT2 = comb4<true>(G0, G1, G2, G3, i2.x, i2.y, i2.z, i2.w);
            JT3 = comb4<true>(G0, G1, G2, G3, i3.x, i3.y, i3.z, i3.w);
            POS.lo = v2f{ 0.f, 0.f }; POS.hi = v2f{ 0.f, 0.f };
            if (with_pos) {                                      // uniform
                const float4 b3 = jconst_lds[4 * LPC + j];
                const Col mp = comb4<true>(JT0, JT1, JT2, JT3, b3.x, b3.y, b3.z, b3.w);    // column 3 of mat4x4_mul(joint_transforms, bind)
                POS = mp;
            }
            if (pos_world) {                                     // uniform
                const Col mp = POS;
                float em[16];
#pragma unroll
                for (int k = 0; k < 16; k++) em[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(em_v), k));
                Col E0, E1, E2, E3;
                E0.lo = v2f{ em[0], em[1] };   E0.hi = v2f{ em[2], em[3] };
                E1.lo = v2f{ em[4], em[5] };   E1.hi = v2f{ em[6], em[7] };
                E2.lo = v2f{ em[8], em[9] };   E2.hi = v2f{ em[10], em[11] };
                E3.lo = v2f{ em[12], em[13] }; E3.hi = v2f{ em[14], em[15] };
                POS = comb4<false>(E0, E1, E2, E3, mp.lo.x, mp.lo.y, mp.hi.x, mp.hi.y);
            }
        }
        pose_lds_sync<LPC>();                                    // every lane has its joint's global: the slots become staging tiles

        QT();
        // ---- 4. the next character's entity matrix
        if (pos_world) em_v = load_em(ei_next);

        // ---- 5. stores: all lanes, no branch; the descriptors clip (each wavefront's own: its 64-joint row of the character)
        const uint32_t rj0 = (uint32_t)__builtin_amdgcn_readfirstlane(row_j0);
        const uint32_t nrow = c_ok && J > rj0 ? (J - rj0 < (uint32_t)WAVE ? J - rj0 : (uint32_t)WAVE) : 0u;
        const size_t row0 = (size_t)c * J + rj0;
        const __amdgpu_buffer_rsrc_t rs_trs = __builtin_amdgcn_make_buffer_rsrc(a.trs + 10 * row0, 0, with_trs ? (int)(nrow * 40u) : 0, POSE_RSRC_FLAGS);
        const __amdgpu_buffer_rsrc_t rs_jt = __builtin_amdgcn_make_buffer_rsrc(a.joint_transforms + 16 * row0, 0, (int)(nrow * 64u), POSE_RSRC_FLAGS);
        const __amdgpu_buffer_rsrc_t rs_pos = __builtin_amdgcn_make_buffer_rsrc(a.joint_pos ? a.joint_pos + 4 * row0 : a.joint_transforms, 0, with_pos ? (int)(nrow * 16u) : 0, POSE_RSRC_FLAGS);
        // this wavefront's 64-joint row of the character: its own 4 KiB of the character's slots as the staging tile
        float4 *tile = G + 4 * row_j0;
        float *tile_f = reinterpret_cast<float *>(tile);
        {
            const float trs_row[10] = { T[0], T[1], T[2], R[0], R[1], R[2], R[3], S[0], S[1], S[2] };
            stage_rows<10>(tile_f, trs_row, lane);
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < 3; k++)                           // 160 16-byte pieces; the third round's upper half lies past the row
                buffer_store4(tile[k * WAVE + lane], rs_trs, (uint32_t)lane * 16u, true, (uint32_t)k * WAVE * 16u);
            wave_lds_fence();
        }
        {
            float JT[16];
            const float4 c0v = f4_of(JT0), c1v = f4_of(JT1), c2v = f4_of(JT2), c3v = f4_of(JT3);
            JT[0] = c0v.x; JT[1] = c0v.y; JT[2] = c0v.z; JT[3] = c0v.w;
            JT[4] = c1v.x; JT[5] = c1v.y; JT[6] = c1v.z; JT[7] = c1v.w;
            JT[8] = c2v.x; JT[9] = c2v.y; JT[10] = c2v.z; JT[11] = c2v.w;
            JT[12] = c3v.x; JT[13] = c3v.y; JT[14] = c3v.z; JT[15] = c3v.w;
            float4 v[4];
            stage_mat4(tile, JT, lane);
            wave_lds_fence();
            unstage_mat4(tile, v, lane);
            // joints that joint 0's tree does not hold are never written (the reference's recursion starts at joint 0)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t piece = (uint32_t)(k * WAVE + lane);
                const bool wr = (reach_row >> (piece >> 2)) & 1ull;
                buffer_store4(v[k], rs_jt, wr ? (uint32_t)lane * 16u : POSE_CLIPPED, false, (uint32_t)k * WAVE * 16u);
            }
            buffer_store4(f4_of(POS), rs_pos, reachable ? (uint32_t)lane * 16u : POSE_CLIPPED, false);
            wave_lds_fence();
        }
        QT();
#ifdef CLAPGPU_POSE_PROF_ITER
        iter_no++;
#endif
#ifdef CLAPGPU_POSE_PROF
        if (pn < 7) PT();
#endif
    }
#ifdef CLAPGPU_POSE_PROF_ITER
    if (tid == 0 && blockIdx.x == 0) {
        printf("pose iter (cycles): wait for keys, T, R, S, gather issue, loc, passes, tail, stores:");
        for (int q = 1; q < qn; q++) printf(" %llu", qt[q] - qt[q - 1]);
        printf("\n");
    }
#endif
#ifdef CLAPGPU_POSE_PROF
    asm volatile("s_waitcnt vmcnt(0)");
    PT();
    if (tid == 0 && blockIdx.x == 0) {
        printf("pose prof (x10 ns):");
        for (int q = 1; q < pn; q++) printf(" %llu", pt[q] - pt[q - 1]);
        printf("\n");
    }
#endif
}

// model.c:1400 behind a pose that stopped at the model-space position (CLAPGPU_POSE_JOINT_POS_MODEL): one lane per joint
__global__ __launch_bounds__(256)
void k_joint_pos_world(uint32_t n_joints_total, uint32_t J, const int32_t *depth, const uint32_t *entity, const float4 *entity_mx,
                       float4 *joint_pos)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n_joints_total) return;
    const uint32_t c = q / J;
    if (depth[q - c * J] < 0) return;                            // outside joint 0's tree: never written (model.c:1583)
    const float4 *em = entity_mx + 4 * (size_t)(entity ? entity[c] : c);
    const float4 mp = joint_pos[q];
    joint_pos[q] = f4_of(comb4<false>(col_of(em[0]), col_of(em[1]), col_of(em[2]), col_of(em[3]), mp.x, mp.y, mp.z, mp.w));
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

// ---- key-major pools (clapgpu_animations_pack): once per model, on the HOST ------------------------------------------
// layout of `packed`, L = the joints rounded up to whole wavefronts (64, 128, 192, 256):
//   times  [n_anims][3][kp][L] f32 (+INF past a channel's last key) | key counts [n_anims][3][L] u32 |
//   (16-byte aligned) values [n_anims][3][k][L] float4 | rotation interval constants [n_anims][k][L] RotConst
// Columns past the last joint repeat the last joint's channels, as the loop's clamped joint index does.
extern "C" int clapgpu_joint_pos_world(void *stream, const clapgpu_skeleton *sk, const clapgpu_pose_batch *pb)
{
    if (!pb || !sk || !sk->nr_joints || !sk->depth)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const uint32_t nr_joints = sk->nr_joints;
    if (pb->n_chars == 0 || !pb->joint_pos)
        return CLAPGPU_OK;
    if (!pb->entity_mx || (uint64_t)pb->n_chars * nr_joints > 0xffffffffull)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const uint32_t total = pb->n_chars * nr_joints;
    hipLaunchKernelGGL(k_joint_pos_world, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), total, nr_joints, sk->depth, pb->entity,
                       reinterpret_cast<const float4 *>(pb->entity_mx), reinterpret_cast<float4 *>(pb->joint_pos));
    CLAPGPU_LAUNCH_CHECK("k_joint_pos_world");
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

static size_t pack_rc_offset(uint32_t n_anims, uint32_t kp, uint32_t kk, uint32_t lanes)
{
    return pack_vals_offset(n_anims, kp, lanes) + (size_t)n_anims * 3 * kk * lanes * 16;
}

#define POSE_LAYOUT_TAG      0x100u
#define POSE_LAYOUT_MISSING  0x010u

extern "C" size_t clapgpu_animations_packed_bytes(uint32_t n_anims, uint32_t max_keys, uint32_t nr_joints)
{
    if (!n_anims || !max_keys || !nr_joints || nr_joints > POSE_MAX_JOINTS) return 0;
    const uint32_t lanes = (nr_joints + 63) / 64 * 64;
    return pack_rc_offset(n_anims, pack_kp(max_keys), max_keys, lanes) + (size_t)n_anims * max_keys * lanes * sizeof(RotConst);
}

// interp.h:91-118 up to the point where the frame's blend factor enters, for the key pair (a, b): the host's float and
// double arithmetic and the host's libm, as the reference runs it
static RotConst rot_const(const float *a, const float *b)
{
    RotConst rc;
    float dot = 0.f;                                             // quat_inner_product (linmath.h:915-922)
    for (int i = 0; i < 4; i++)
        dot += b[i] * a[i];
    bool flip = false;
    if (dot < 0.0) {
        dot = -dot;
        flip = true;
    }
    if (dot > 0.9995) {                                          // quat_interp: nothing to precompute
        rc.theta0 = -1.0f;
        rc.inv_sin0 = 0.0;
    } else {
        const float theta_0 = (float)acos((double)dot);         // C's acos(float) is the double function (in C++ it would be acosf)
        const float sin_theta_0 = (float)sin((double)theta_0);
        rc.theta0 = theta_0;
        rc.inv_sin0 = 1.0 / (double)sin_theta_0;
    }
    uint32_t bits;
    memcpy(&bits, &dot, 4);
    bits = (bits & 0x7fffffffu) | (flip ? 0x80000000u : 0u);
    memcpy(&rc.dot_flip, &bits, 4);
    return rc;
}

extern "C" int clapgpu_animations_pack(void *stream, const clapgpu_animations *an, uint32_t nr_joints, uint32_t max_keys,
                                       void *packed, uint32_t *packed_layout)
{
    if (!an || !packed || !packed_layout || !an->chan_table || !an->times || !an->data || !an->n_anims || !max_keys)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (nr_joints == 0 || nr_joints > POSE_MAX_JOINTS || an->n_anims > 0xffffu)   // JOINTS_MAX is 200 (shader_constants.h:6)
        return CLAPGPU_ERR_TOO_LARGE;
    if ((reinterpret_cast<uintptr_t>(packed) & 15u) != 0)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const uint32_t A = an->n_anims, J = nr_joints, kk = max_keys;
    const uint32_t kp = pack_kp(kk), L = (J + 63) / 64 * 64;
    hipStream_t s = as_stream(stream);

    // the model's channel records, then as much of the pools as they address
    std::vector<uint32_t> tab((size_t)A * J * 3 * 4);
    CLAPGPU_HIP(hipMemcpyAsync(tab.data(), an->chan_table, tab.size() * 4, hipMemcpyDeviceToHost, s));
    CLAPGPU_HIP(hipStreamSynchronize(s));
    size_t n_times = 0, n_data = 0;
    bool missing = false;
    for (size_t q = 0; q < (size_t)A * J * 3; q++) {
        const uint32_t t_off = tab[4 * q], d_off = tab[4 * q + 1], nr = tab[4 * q + 2];
        if ((int32_t)nr <= 0) { missing = true; continue; }
        if (nr > kk)
            return CLAPGPU_ERR_INVALID_ARGUMENTS;                 // max_keys is not the longest channel
        const uint32_t stride = (q % 3) == 1 ? 4u : 3u;
        if ((size_t)t_off + nr > n_times) n_times = (size_t)t_off + nr;
        if ((size_t)d_off + (size_t)nr * stride > n_data) n_data = (size_t)d_off + (size_t)nr * stride;
    }
    if ((an->n_times && n_times > an->n_times) || (an->n_data && n_data > an->n_data))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;                     // a channel record that reads past its pool
    std::vector<float> times(n_times ? n_times : 1), data(n_data ? n_data : 1);
    if (n_times) CLAPGPU_HIP(hipMemcpyAsync(times.data(), an->times, n_times * 4, hipMemcpyDeviceToHost, s));
    if (n_data) CLAPGPU_HIP(hipMemcpyAsync(data.data(), an->data, n_data * 4, hipMemcpyDeviceToHost, s));
    CLAPGPU_HIP(hipStreamSynchronize(s));

    // The kernel's bracket search counts the keys below the time (lo = #{t[k] < time}); that is channel_time_to_idx
    // (model.c:1266-1288) for strictly increasing key times and for nothing else: the reference's cursor-dependent scan
    // gives other pairs on equal or descending times.  Such an asset is refused here, once, not mis-posed every frame.
    for (size_t q = 0; q < (size_t)A * J * 3; q++) {
        const uint32_t nr = tab[4 * q + 2];
        if ((int32_t)nr <= 1) continue;
        const float *t = times.data() + tab[4 * q];
        for (uint32_t k = 0; k + 1 < nr; k++)
            if (!(t[k] < t[k + 1])) {
                char msg[160];
                snprintf(msg, sizeof(msg), "clapgpu_animations_pack: key times of animation %zu joint %zu path %zu are not strictly increasing at key %u",
                         q / (3 * (size_t)J), (q / 3) % J, q % 3, k);
                set_last_error(msg);
                return CLAPGPU_ERR_INVALID_ARGUMENTS;
            }
    }

    const size_t total = clapgpu_animations_packed_bytes(A, kk, J);
    std::vector<unsigned char> img(total, 0);
    float *o_times = reinterpret_cast<float *>(img.data());
    uint32_t *o_nr = reinterpret_cast<uint32_t *>(o_times + (size_t)A * 3 * kp * L);
    float *o_vals = reinterpret_cast<float *>(img.data() + pack_vals_offset(A, kp, L));
    RotConst *o_rc = reinterpret_cast<RotConst *>(img.data() + pack_rc_offset(A, kp, kk, L));
    for (uint32_t a = 0; a < A; a++)
        for (uint32_t p = 0; p < 3; p++)
            for (uint32_t lane = 0; lane < L; lane++) {
                const uint32_t j = lane < J ? lane : J - 1;
                const uint32_t *e = &tab[(((size_t)a * J + j) * 3 + p) * 4];
                const uint32_t nr = (int32_t)e[2] > 0 ? e[2] : 0u;
                const uint32_t stride = p == 1 ? 4u : 3u;
                const float *t = times.data() + e[0], *d = data.data() + e[1];
                const size_t ap = (size_t)a * 3 + p;
                o_nr[ap * L + lane] = nr;
                for (uint32_t k = 0; k < kp; k++)
                    o_times[(ap * kp + k) * L + lane] = k < nr ? t[k] : INFINITY;
                for (uint32_t k = 0; k < nr; k++) {
                    float *v = o_vals + ((ap * kk + k) * L + lane) * 4;
                    v[0] = d[stride * k]; v[1] = d[stride * k + 1]; v[2] = d[stride * k + 2];
                    v[3] = p == 1 ? d[stride * k + 3] : 0.f;
                }
                if (p == 1)                                       // interval k = the key pair (k, k + 1), the last one wraps to key 0
                    for (uint32_t k = 0; k < nr; k++)
                        o_rc[((size_t)a * kk + k) * L + lane] = rot_const(d + 4 * k, d + 4 * (k + 1 < nr ? k + 1 : 0));
            }
    CLAPGPU_HIP(hipMemcpyAsync(packed, img.data(), total, hipMemcpyHostToDevice, s));
    CLAPGPU_HIP(hipStreamSynchronize(s));
    *packed_layout = (L / 64) | (missing ? POSE_LAYOUT_MISSING : 0u) | POSE_LAYOUT_TAG | (A << 16);
    return CLAPGPU_OK;
}

template <int LPC, int BLOCK>
static int pose_launch(hipStream_t s, PoseArgs &a, bool missing, bool times_lds, int n_cus, uint32_t n_levels)
{
    constexpr uint32_t cpb = BLOCK / LPC;
    const uint32_t n_groups = (a.n_chars + cpb - 1) / cpb;
    const void *fn = missing ? (times_lds ? (const void *)k_pose<LPC, BLOCK, true, true> : (const void *)k_pose<LPC, BLOCK, true, false>)
                             : (times_lds ? (const void *)k_pose<LPC, BLOCK, false, true> : (const void *)k_pose<LPC, BLOCK, false, false>);
    // the level passes' program words: a level of w joints is ceil(w / (LPC / 4)) passes, so n_levels + J / (LPC / 4)
    // bounds them; what the LDS left beside the kernel's static arrays cannot hold is computed on the fly
    static thread_local struct { int dev; uint32_t res[4], dyn[4]; size_t stat[4]; } cache = { -1, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
    int dev = 0;
    CLAPGPU_HIP(hipGetDevice(&dev));
    if (cache.dev != dev) { cache.dev = dev; memset(cache.res, 0, sizeof(cache.res)); memset(cache.stat, 0, sizeof(cache.stat)); }
    const int slot = (missing ? 2 : 0) + (times_lds ? 1 : 0);
    if (!cache.stat[slot]) {
        hipFuncAttributes fa;
        CLAPGPU_HIP(hipFuncGetAttributes(&fa, fn));
        cache.stat[slot] = fa.sharedSizeBytes ? fa.sharedSizeBytes : 1;
        if (cache.stat[slot] < 160u * 1024u)
            CLAPGPU_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160u * 1024u - cache.stat[slot])));
    }
    const size_t lds_total = 160u * 1024u;
    uint32_t want = (n_levels + (a.J + LPC / 4 - 1) / (LPC / 4) + 1);
    const size_t room = lds_total > cache.stat[slot] ? (lds_total - cache.stat[slot]) / ((size_t)LPC * 4) : 0;
    if (want > room && times_lds)                               // the passes' program does not fit beside the key times in LDS:
        return pose_launch<LPC, BLOCK>(s, a, missing, false, n_cus, n_levels);   // the times through L2 instead (passes computed on the fly cost more)
    if (want > room) want = (uint32_t)room;
    if (want > 2 * POSE_MAX_JOINTS) want = 2 * POSE_MAX_JOINTS;
    a.prog_passes = want;
    const uint32_t dyn = want * LPC * 4;
    // persistent blocks (their LDS tables and key times are built once): exactly as many as are resident at once
    uint32_t &res = cache.res[slot];
    if (!res || cache.dyn[slot] != dyn) {
        int per_cu = 0;
        CLAPGPU_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, BLOCK, dyn));
        res = (uint32_t)(per_cu > 0 ? per_cu : 1) * (uint32_t)n_cus;
        cache.dyn[slot] = dyn;
    }
    uint32_t blocks = n_groups < res ? n_groups : res;
    if (const char *cap = getenv("CLAPGPU_POSE_BLOCKS")) {      // tests: a few persistent blocks take many characters each
        const long v = strtol(cap, nullptr, 10);
        if (v > 0 && (uint32_t)v < blocks) blocks = (uint32_t)v;
    }
    const dim3 grid(blocks), block(BLOCK);
    if (getenv("CLAPGPU_POSE_DEBUG"))
        fprintf(stderr, "k_pose<%d, %d>: %u blocks resident (%u per CU), static LDS %zu + dynamic %u, %u program passes\n", LPC, BLOCK,
                res, res / (uint32_t)n_cus, cache.stat[slot], dyn, want);
    if (missing) {
        if (times_lds) hipLaunchKernelGGL((k_pose<LPC, BLOCK, true, true>), grid, block, dyn, s, a);
        else hipLaunchKernelGGL((k_pose<LPC, BLOCK, true, false>), grid, block, dyn, s, a);
    } else {
        if (times_lds) hipLaunchKernelGGL((k_pose<LPC, BLOCK, false, true>), grid, block, dyn, s, a);
        else hipLaunchKernelGGL((k_pose<LPC, BLOCK, false, false>), grid, block, dyn, s, a);
    }
    CLAPGPU_LAUNCH_CHECK("k_pose");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_pose_update(void *stream, const clapgpu_skeleton *sk, const clapgpu_animations *an,
                                   const clapgpu_pose_batch *pb)
{
    if (!sk || !an || !pb)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!sk->parent || !sk->depth || !sk->root_pose || !sk->invmx || !sk->bind)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (pb->n_chars == 0)
        return CLAPGPU_OK;
    if (pb->skip & ~(uint32_t)(CLAPGPU_POSE_SKIP_TRS | CLAPGPU_POSE_SKIP_JOINT_POS | CLAPGPU_POSE_JOINT_POS_MODEL))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    // A (joint, path) without a channel keeps the joint's last interpolated value (model.c:1301): the kernel re-reads it
    // from trs[], so a model with such paths cannot run with trs[] left unwritten -- a character that switches from an
    // animation with the channel to one without would pose from the values of its first frame.
    if ((pb->skip & CLAPGPU_POSE_SKIP_TRS) && (an->packed_layout & POSE_LAYOUT_TAG) && (an->packed_layout & POSE_LAYOUT_MISSING)) {
        set_last_error("clapgpu_pose_update: CLAPGPU_POSE_SKIP_TRS with a model some of whose (joint, path) pairs have no channel");
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    }
    if (!pb->anim || !pb->frame_time || !pb->trs || !pb->joint_transforms ||
        (pb->joint_pos && !pb->entity_mx && !(pb->skip & (CLAPGPU_POSE_SKIP_JOINT_POS | CLAPGPU_POSE_JOINT_POS_MODEL))))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (sk->nr_joints == 0 || sk->nr_joints > POSE_MAX_JOINTS)  // JOINTS_MAX is 200 (shader_constants.h:6)
        return CLAPGPU_ERR_TOO_LARGE;
    // the pools of clapgpu_animations_pack() for THIS skeleton class and animation count (they carry what the reference's
    // slerp derives from each key pair with the host's libm: the kernel has no other source for it)
    const uint32_t lpc = (sk->nr_joints + 63) / 64 * 64;
    const uint32_t n_anims = an->n_anims ? an->n_anims : 1;
    if (!an->packed || !an->packed_keys || !(an->packed_layout & POSE_LAYOUT_TAG) || (an->packed_layout & 0xfu) * 64u != lpc ||
        (an->packed_layout >> 16) != n_anims)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;

    PoseArgs a;
    a.J = sk->nr_joints;
    a.parent = sk->parent;
    a.depth = sk->depth;
    a.root_pose = sk->root_pose;
    a.invmx = reinterpret_cast<const float4 *>(sk->invmx);
    a.bind = reinterpret_cast<const float4 *>(sk->bind);
    a.n_anims = n_anims;
    const uint32_t kk = an->packed_keys, kp = pack_kp(kk);
    const char *base = static_cast<const char *>(an->packed);
    a.pk_times = reinterpret_cast<const float *>(base);
    a.pk_vals = reinterpret_cast<const float4 *>(base + pack_vals_offset(n_anims, kp, lpc));
    a.pk_rc = reinterpret_cast<const uint4 *>(base + pack_rc_offset(n_anims, kp, kk, lpc));
    a.pk_k = kk; a.pk_kp = kp;
    a.n_chars = pb->n_chars;
    a.anim = pb->anim;
    a.frame_time = pb->frame_time;
    a.entity = pb->entity;
    a.entity_mx = pb->entity_mx;
    a.trs = pb->trs;
    a.joint_transforms = pb->joint_transforms;
    a.joint_pos = pb->joint_pos;
    a.skip = pb->skip | (pb->joint_pos ? 0u : (uint32_t)CLAPGPU_POSE_SKIP_JOINT_POS);

    const bool missing = (an->packed_layout & POSE_LAYOUT_MISSING) != 0;
    // key times in LDS whenever the model's rows fit (through L2 instead: 132 -> 147 us at 64 joints, 170 -> 202 at 128)
    const bool times_lds = (uint64_t)n_anims * (3u * kp + 3u) * lpc <= (uint64_t)POSE_TIMES_LDS_MAX * (lpc / 64);
    hipStream_t s = as_stream(stream);
    static thread_local struct { int dev, n_cus; } cus = { -1, 0 };
    {
        int dev = 0;
        CLAPGPU_HIP(hipGetDevice(&dev));
        if (cus.dev != dev) {
            hipDeviceProp_t prop;
            CLAPGPU_HIP(hipGetDeviceProperties(&prop, dev));
            cus.dev = dev;
            cus.n_cus = prop.multiProcessorCount;
        }
    }
    // One block per CU, as many wavefronts as its LDS and the registers allow: the model's key times (25 KiB per 64 lanes)
    // are staged once per block, each character in flight needs 8.2 KiB (globals + local columns) per 64 joints, and at
    // ~160 VGPRs three wavefronts fit a SIMD -- 12 characters of <= 64 joints per block (136 + 3 KiB of the CU's 160).
    // (Two blocks of 320 threads measured as ONE resident block per CU although 2 x 80.8 KB fit 160 KiB on paper; two of
    // 256: 121 / 139 us against 112 / 136 for this form.)
    switch (lpc) {
    case 64:  return pose_launch<64, POSE_BLOCK64>(s, a, missing, times_lds, cus.n_cus, sk->n_levels);
    case 128: return pose_launch<128, 512>(s, a, missing, times_lds, cus.n_cus, sk->n_levels);
    case 192: return pose_launch<192, 384>(s, a, missing, times_lds, cus.n_cus, sk->n_levels);
    default:  return pose_launch<256, 256>(s, a, missing, times_lds, cus.n_cus, sk->n_levels);
    }
}
