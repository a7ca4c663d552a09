// lm_dev.h -- fp32 matrix arithmetic for the gfx950 kernels, in the operation
// ORDER of the reference's x86-64 scalar path (core/linmath.h), so that world
// matrices, AABBs and therefore cull decisions come out bit-identical.
//
// Build rule: -ffp-contract=off (hipcc defaults to fast contraction).  The
// reference's x86 path has no FMA; a fused a*b+c changes the last bit of mx,
// which moves AABBs, which flips cull results at frustum edges.
//
// mat4 = float[16] column-major, (col c,row r) at [4c+r].  Everything is
// fully unrolled with compile-time indices so matrices live in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

#define LMD __host__ __device__ __forceinline__
#define E_(m, c, r) ((m)[4 * (c) + (r)])

namespace lmd {

// linmath.h:506-516: out[c][r] = ((0 + a[0][r] b[c][0]) + a[1][r] b[c][1]) + ...
LMD void mul(float (&out)[16], const float (&a)[16], const float (&b)[16])
{
    float t[16];
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++)
                s += E_(a, k, r) * E_(b, c, k);
            E_(t, c, r) = s;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = t[i];
}

// linmath.h:959-987 with a=w b=x c=y d=z
LMD void from_quat(float (&m)[16], float qx, float qy, float qz, float qw)
{
    float a = qw, b = qx, c = qy, d = qz;
    float a2 = a * a, b2 = b * b, c2 = c * c, d2 = d * d;

    E_(m, 0, 0) = a2 + b2 - c2 - d2;
    E_(m, 0, 1) = 2.f * (b * c + a * d);
    E_(m, 0, 2) = 2.f * (b * d - a * c);
    E_(m, 0, 3) = 0.f;
    E_(m, 1, 0) = 2.f * (b * c - a * d);
    E_(m, 1, 1) = a2 - b2 + c2 - d2;
    E_(m, 1, 2) = 2.f * (c * d + a * b);
    E_(m, 1, 3) = 0.f;
    E_(m, 2, 0) = 2.f * (b * d + a * c);
    E_(m, 2, 1) = 2.f * (c * d - a * b);
    E_(m, 2, 2) = a2 - b2 - c2 + d2;
    E_(m, 2, 3) = 0.f;
    E_(m, 3, 0) = 0.f;
    E_(m, 3, 1) = 0.f;
    E_(m, 3, 2) = 0.f;
    E_(m, 3, 3) = 1.f;
}

LMD void identity(float (&m)[16])
{
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.f : 0.f;
}

// linmath.h:525-534: M[3][i] += dot4(row_i(M), (x,y,z,0)), dot4 = (((0 + r0 x) + r1 y) + r2 z) + r3*0
LMD void translate_in_place(float (&m)[16], float x, float y, float z)
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float p = 0.f;
        p += x * E_(m, 0, i);
        p += y * E_(m, 1, i);
        p += z * E_(m, 2, i);
        p += 0.f * E_(m, 3, i);
        E_(m, 3, i) += p;
    }
}

// linmath.h:448-457
LMD void scale_aniso(float (&m)[16], float x, float y, float z)
{
#pragma unroll
    for (int r = 0; r < 4; r++) {
        E_(m, 0, r) = E_(m, 0, r) * x;
        E_(m, 1, r) = E_(m, 1, r) * y;
        E_(m, 2, r) = E_(m, 2, r) * z;
    }
}

// model.c:1618-1622 / 1670-1675: I -> translate_in_place(pos) -> (* R(quat)) -> scale_aniso(s,s,s)
LMD void trs(float (&m)[16], float px, float py, float pz, float s,
             float qx, float qy, float qz, float qw)
{
    float r[16];
    identity(m);
    translate_in_place(m, px, py, pz);
    from_quat(r, qx, qy, qz, qw);
    mul(m, m, r);
    scale_aniso(m, s, s, s);
}

// linmath.h:611-651
LMD void invert(float (&t)[16], const float (&m)[16])
{
    float s0 = E_(m,0,0)*E_(m,1,1) - E_(m,1,0)*E_(m,0,1);
    float s1 = E_(m,0,0)*E_(m,1,2) - E_(m,1,0)*E_(m,0,2);
    float s2 = E_(m,0,0)*E_(m,1,3) - E_(m,1,0)*E_(m,0,3);
    float s3 = E_(m,0,1)*E_(m,1,2) - E_(m,1,1)*E_(m,0,2);
    float s4 = E_(m,0,1)*E_(m,1,3) - E_(m,1,1)*E_(m,0,3);
    float s5 = E_(m,0,2)*E_(m,1,3) - E_(m,1,2)*E_(m,0,3);

    float c0 = E_(m,2,0)*E_(m,3,1) - E_(m,3,0)*E_(m,2,1);
    float c1 = E_(m,2,0)*E_(m,3,2) - E_(m,3,0)*E_(m,2,2);
    float c2 = E_(m,2,0)*E_(m,3,3) - E_(m,3,0)*E_(m,2,3);
    float c3 = E_(m,2,1)*E_(m,3,2) - E_(m,3,1)*E_(m,2,2);
    float c4 = E_(m,2,1)*E_(m,3,3) - E_(m,3,1)*E_(m,2,3);
    float c5 = E_(m,2,2)*E_(m,3,3) - E_(m,3,2)*E_(m,2,3);

    float idet = 1.0f / (s0*c5 - s1*c4 + s2*c3 + s3*c2 - s4*c1 + s5*c0);

    E_(t,0,0) = ( E_(m,1,1)*c5 - E_(m,1,2)*c4 + E_(m,1,3)*c3) * idet;
    E_(t,0,1) = (-E_(m,0,1)*c5 + E_(m,0,2)*c4 - E_(m,0,3)*c3) * idet;
    E_(t,0,2) = ( E_(m,3,1)*s5 - E_(m,3,2)*s4 + E_(m,3,3)*s3) * idet;
    E_(t,0,3) = (-E_(m,2,1)*s5 + E_(m,2,2)*s4 - E_(m,2,3)*s3) * idet;

    E_(t,1,0) = (-E_(m,1,0)*c5 + E_(m,1,2)*c2 - E_(m,1,3)*c1) * idet;
    E_(t,1,1) = ( E_(m,0,0)*c5 - E_(m,0,2)*c2 + E_(m,0,3)*c1) * idet;
    E_(t,1,2) = (-E_(m,3,0)*s5 + E_(m,3,2)*s2 - E_(m,3,3)*s1) * idet;
    E_(t,1,3) = ( E_(m,2,0)*s5 - E_(m,2,2)*s2 + E_(m,2,3)*s1) * idet;

    E_(t,2,0) = ( E_(m,1,0)*c4 - E_(m,1,1)*c2 + E_(m,1,3)*c0) * idet;
    E_(t,2,1) = (-E_(m,0,0)*c4 + E_(m,0,1)*c2 - E_(m,0,3)*c0) * idet;
    E_(t,2,2) = ( E_(m,3,0)*s4 - E_(m,3,1)*s2 + E_(m,3,3)*s0) * idet;
    E_(t,2,3) = (-E_(m,2,0)*s4 + E_(m,2,1)*s2 - E_(m,2,3)*s0) * idet;

    E_(t,3,0) = (-E_(m,1,0)*c3 + E_(m,1,1)*c1 - E_(m,1,2)*c0) * idet;
    E_(t,3,1) = ( E_(m,0,0)*c3 - E_(m,0,1)*c1 + E_(m,0,2)*c0) * idet;
    E_(t,3,2) = (-E_(m,3,0)*s3 + E_(m,3,1)*s1 - E_(m,3,2)*s0) * idet;
    E_(t,3,3) = ( E_(m,2,0)*s3 - E_(m,2,1)*s1 + E_(m,2,2)*s0) * idet;
}

// linmath.h:297-305 with v = (x,y,z,w): ((m0 x + m1 y) + m2 z) + m3 w, rows 0..2 only
LMD void mul_point3(float &ox, float &oy, float &oz, const float (&m)[16],
                    float x, float y, float z, float w)
{
    float t0 = E_(m,0,0) * x + E_(m,1,0) * y + E_(m,2,0) * z;  t0 += E_(m,3,0) * w;
    float t1 = E_(m,0,1) * x + E_(m,1,1) * y + E_(m,2,1) * z;  t1 += E_(m,3,1) * w;
    float t2 = E_(m,0,2) * x + E_(m,1,2) * y + E_(m,2,2) * z;  t2 += E_(m,3,2) * w;
    ox = t0; oy = t1; oz = t2;
}

LMD void mul_vec4(float (&o)[4], const float (&m)[16], const float (&v)[4])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float t = E_(m,0,i) * v[0] + E_(m,1,i) * v[1] + E_(m,2,i) * v[2];
        t += E_(m,3,i) * v[3];
        o[i] = t;
    }
}

// util.h:188-201: ternary min / max (kept literal for NaN ordering)
LMD float tmin(float a, float b) { return a < b ? a : b; }
LMD float tmax(float a, float b) { return a > b ? a : b; }

// model.c:1200-1234 (corner order 1207-1216) + util.h:104-109 aabb_center
LMD void world_aabb(float (&bb)[6], float (&ctr)[3], const float (&m)[16],
                    float lx, float ly, float lz, float hx, float hy, float hz)
{
    bb[0] = bb[1] = bb[2] = INFINITY;
    bb[3] = bb[4] = bb[5] = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float x = (i & 4) ? hx : lx;
        float y = (i & 1) ? hy : ly;
        float z = (i & 2) ? hz : lz;
        float vx, vy, vz;
        mul_point3(vx, vy, vz, m, x, y, z, 1.0f);
        bb[0] = tmin(vx, bb[0]);  bb[3] = tmax(vx, bb[3]);
        bb[1] = tmin(vy, bb[1]);  bb[4] = tmax(vy, bb[4]);
        bb[2] = tmin(vz, bb[2]);  bb[5] = tmax(vz, bb[5]);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float d = bb[3 + k] - bb[k];
        d = d * 0.5f;
        ctr[k] = d + bb[k];
    }
}

struct Frustum {
    float planes[6][4];
    float corners[8][4];
};

// view.c:296-337.  dot4 order: (((0 + x px) + y py) + z pz) + 1 pw, compared `< 0.0`.
LMD bool aabb_in_frustum(const Frustum &f, const float (&bb)[6])
{
#pragma unroll
    for (int i = 0; i < 6; i++) {
        int r = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float x = (k & 1) ? bb[3] : bb[0];
            float y = (k & 2) ? bb[4] : bb[1];
            float z = (k & 4) ? bb[5] : bb[2];
            float p = 0.f;
            p += x * f.planes[i][0];
            p += y * f.planes[i][1];
            p += z * f.planes[i][2];
            p += 1.0f * f.planes[i][3];
            r += (p < 0.0f) ? 1 : 0;
        }
        if (r == 8)
            return false;
    }
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
        int r = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) r += (f.corners[i][ax] > bb[3 + ax]) ? 1 : 0;
        if (r == 8) return false;
        r = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) r += (f.corners[i][ax] < bb[ax]) ? 1 : 0;
        if (r == 8) return false;
    }
    return true;
}

// Kernel-side frustum: the reference's planes/corners plus per-axis extremes of the 8
// frustum corners, precomputed on the host (make_frustum_k in entities.hip).
struct FrustumK {
    Frustum  f;
    float    cmin[3];      // min_i corners[i][ax], NaN if any is NaN
    float    cmax[3];      // max_i corners[i][ax], NaN if any is NaN
    uint32_t finite;       // every plane component is finite
};

// Same decisions as aabb_in_frustum with ~1/8 of the arithmetic:
//  * plane i rejects iff all 8 corner dots are < 0 iff the LARGEST dot is < 0.  Each dot is the
//    fp32 chain (((0 + x px) + y py) + z pz) + pw; rounding is monotone, so the chain is
//    non-decreasing in every product and the largest value is reached at the corner that takes
//    max[a] where the plane component is >= 0 and min[a] otherwise -- the same fp32 operations
//    on the same operands, hence the same bits.  With a non-finite box or plane (0 * inf = NaN
//    breaks monotonicity) the literal 8-corner test is used.
//  * "all 8 frustum corners beyond the box on axis a" compares the per-axis extreme only.
LMD bool aabb_in_frustum_fast(const FrustumK &k, const float (&bb)[6])
{
    bool box_finite = true;
#pragma unroll
    for (int a = 0; a < 6; a++) box_finite = box_finite && (fabsf(bb[a]) <= 3.402823466e+38f);
    if (!(k.finite && box_finite))
        return aabb_in_frustum(k.f, bb);
#pragma unroll
    for (int i = 0; i < 6; i++) {
        // (fmin/fmax: a stored box with max < min is still handled exactly)
        const float x = (k.f.planes[i][0] >= 0.f) ? fmaxf(bb[0], bb[3]) : fminf(bb[0], bb[3]);
        const float y = (k.f.planes[i][1] >= 0.f) ? fmaxf(bb[1], bb[4]) : fminf(bb[1], bb[4]);
        const float z = (k.f.planes[i][2] >= 0.f) ? fmaxf(bb[2], bb[5]) : fminf(bb[2], bb[5]);
        float p = 0.f;
        p += x * k.f.planes[i][0];
        p += y * k.f.planes[i][1];
        p += z * k.f.planes[i][2];
        p += 1.0f * k.f.planes[i][3];
        if (p < 0.0f)
            return false;
    }
#pragma unroll
    for (int ax = 0; ax < 3; ax++) {
        if (k.cmin[ax] > bb[3 + ax]) return false;
        if (k.cmax[ax] < bb[ax]) return false;
    }
    return true;
}

} // namespace lmd
