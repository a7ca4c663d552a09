/*
 * oracle/lm.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Scalar restatement of the x86-64 (non-NEON) arithmetic of the reference's
 * core/linmath.h, core/interp.h and core/util.h, flattened: a mat4 is
 * float[16], column-major, element (col c, row r) at [4*c + r]
 * (reference: `mat4x4 M`, `M[c][r]`, linmath.h:281-296); a quat is
 * (x, y, z, w) (linmath.h:835-840).
 *
 * Every function names the reference lines whose operation ORDER it
 * reproduces.  Build with -ffp-contract=off: the reference's x86 path has no
 * fused multiply-add, and bit-exact AABBs / cull results depend on that.
 */
#ifndef CLAP_ORACLE_LM_H
#define CLAP_ORACLE_LM_H

#include <math.h>
#include <string.h>

#define LM_E(m, c, r) ((m)[4 * (c) + (r)])

/* linmath.h:283-289 */
static inline void lm_m4_identity(float *m)
{
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++)
            LM_E(m, c, r) = (c == r) ? 1.f : 0.f;
}

/* linmath.h:40-47: p = 0; p += b[i]*a[i], i ascending */
static inline float lm_dot4(const float *a, const float *b)
{
    float p = 0.f;
    for (int i = 0; i < 4; i++)
        p += b[i] * a[i];
    return p;
}

static inline float lm_dot3(const float *a, const float *b)
{
    float p = 0.f;
    for (int i = 0; i < 3; i++)
        p += b[i] * a[i];
    return p;
}

/*
 * linmath.h:506-516 (scalar mat4x4_mul): out[c][r] = 0 + sum_k a[k][r]*b[c][k],
 * k ascending, through a temporary so out may alias a or b.
 */
static inline void lm_m4_mul(float *out, const float *a, const float *b)
{
    float t[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            float s = 0.f;
            for (int k = 0; k < 4; k++)
                s += LM_E(a, k, r) * LM_E(b, c, k);
            LM_E(t, c, r) = s;
        }
    memcpy(out, t, sizeof(t));
}

/*
 * linmath.h:525-534 (mat4x4_translate_in_place): for each row i,
 * M[3][i] += dot4(row_i(M), (x, y, z, 0)).
 */
static inline void lm_m4_translate_in_place(float *m, float x, float y, float z)
{
    const float t[4] = { x, y, z, 0.f };
    for (int i = 0; i < 4; i++) {
        float row[4] = { LM_E(m, 0, i), LM_E(m, 1, i), LM_E(m, 2, i), LM_E(m, 3, i) };
        LM_E(m, 3, i) += lm_dot4(row, t);
    }
}

/* linmath.h:518-524 (mat4x4_translate): identity with column 3 = (x,y,z,1) */
static inline void lm_m4_translate(float *m, float x, float y, float z)
{
    lm_m4_identity(m);
    LM_E(m, 3, 0) = x;
    LM_E(m, 3, 1) = y;
    LM_E(m, 3, 2) = z;
}

/* linmath.h:959-987 (mat4x4_from_quat); a=w b=x c=y d=z */
static inline void lm_m4_from_quat(float *m, const float *q)
{
    float a = q[3], b = q[0], c = q[1], d = q[2];
    float a2 = a * a, b2 = b * b, c2 = c * c, d2 = d * d;

    LM_E(m, 0, 0) = a2 + b2 - c2 - d2;
    LM_E(m, 0, 1) = 2.f * (b * c + a * d);
    LM_E(m, 0, 2) = 2.f * (b * d - a * c);
    LM_E(m, 0, 3) = 0.f;

    LM_E(m, 1, 0) = 2 * (b * c - a * d);
    LM_E(m, 1, 1) = a2 - b2 + c2 - d2;
    LM_E(m, 1, 2) = 2.f * (c * d + a * b);
    LM_E(m, 1, 3) = 0.f;

    LM_E(m, 2, 0) = 2.f * (b * d + a * c);
    LM_E(m, 2, 1) = 2.f * (c * d - a * b);
    LM_E(m, 2, 2) = a2 - b2 - c2 + d2;
    LM_E(m, 2, 3) = 0.f;

    LM_E(m, 3, 0) = LM_E(m, 3, 1) = LM_E(m, 3, 2) = 0.f;
    LM_E(m, 3, 3) = 1.f;
}

/* linmath.h:448-457 (mat4x4_scale_aniso): columns 0..2 scaled, column 3 copied */
static inline void lm_m4_scale_aniso(float *out, const float *a, float x, float y, float z)
{
    for (int r = 0; r < 4; r++) LM_E(out, 0, r) = LM_E(a, 0, r) * x;
    for (int r = 0; r < 4; r++) LM_E(out, 1, r) = LM_E(a, 1, r) * y;
    for (int r = 0; r < 4; r++) LM_E(out, 2, r) = LM_E(a, 2, r) * z;
    for (int r = 0; r < 4; r++) LM_E(out, 3, r) = LM_E(a, 3, r);
}

/* linmath.h:611-651 (mat4x4_invert): 2x2 sub-determinant cofactor form */
static inline void lm_m4_invert(float *t, const float *m)
{
#define M_(c, r) LM_E(m, c, r)
    float s[6], c[6];
    s[0] = M_(0,0)*M_(1,1) - M_(1,0)*M_(0,1);
    s[1] = M_(0,0)*M_(1,2) - M_(1,0)*M_(0,2);
    s[2] = M_(0,0)*M_(1,3) - M_(1,0)*M_(0,3);
    s[3] = M_(0,1)*M_(1,2) - M_(1,1)*M_(0,2);
    s[4] = M_(0,1)*M_(1,3) - M_(1,1)*M_(0,3);
    s[5] = M_(0,2)*M_(1,3) - M_(1,2)*M_(0,3);

    c[0] = M_(2,0)*M_(3,1) - M_(3,0)*M_(2,1);
    c[1] = M_(2,0)*M_(3,2) - M_(3,0)*M_(2,2);
    c[2] = M_(2,0)*M_(3,3) - M_(3,0)*M_(2,3);
    c[3] = M_(2,1)*M_(3,2) - M_(3,1)*M_(2,2);
    c[4] = M_(2,1)*M_(3,3) - M_(3,1)*M_(2,3);
    c[5] = M_(2,2)*M_(3,3) - M_(3,2)*M_(2,3);

    float idet = 1.0f / (s[0]*c[5] - s[1]*c[4] + s[2]*c[3] + s[3]*c[2] - s[4]*c[1] + s[5]*c[0]);
    float o[16];

    LM_E(o,0,0) = ( M_(1,1)*c[5] - M_(1,2)*c[4] + M_(1,3)*c[3]) * idet;
    LM_E(o,0,1) = (-M_(0,1)*c[5] + M_(0,2)*c[4] - M_(0,3)*c[3]) * idet;
    LM_E(o,0,2) = ( M_(3,1)*s[5] - M_(3,2)*s[4] + M_(3,3)*s[3]) * idet;
    LM_E(o,0,3) = (-M_(2,1)*s[5] + M_(2,2)*s[4] - M_(2,3)*s[3]) * idet;

    LM_E(o,1,0) = (-M_(1,0)*c[5] + M_(1,2)*c[2] - M_(1,3)*c[1]) * idet;
    LM_E(o,1,1) = ( M_(0,0)*c[5] - M_(0,2)*c[2] + M_(0,3)*c[1]) * idet;
    LM_E(o,1,2) = (-M_(3,0)*s[5] + M_(3,2)*s[2] - M_(3,3)*s[1]) * idet;
    LM_E(o,1,3) = ( M_(2,0)*s[5] - M_(2,2)*s[2] + M_(2,3)*s[1]) * idet;

    LM_E(o,2,0) = ( M_(1,0)*c[4] - M_(1,1)*c[2] + M_(1,3)*c[0]) * idet;
    LM_E(o,2,1) = (-M_(0,0)*c[4] + M_(0,1)*c[2] - M_(0,3)*c[0]) * idet;
    LM_E(o,2,2) = ( M_(3,0)*s[4] - M_(3,1)*s[2] + M_(3,3)*s[0]) * idet;
    LM_E(o,2,3) = (-M_(2,0)*s[4] + M_(2,1)*s[2] - M_(2,3)*s[0]) * idet;

    LM_E(o,3,0) = (-M_(1,0)*c[3] + M_(1,1)*c[1] - M_(1,2)*c[0]) * idet;
    LM_E(o,3,1) = ( M_(0,0)*c[3] - M_(0,1)*c[1] + M_(0,2)*c[0]) * idet;
    LM_E(o,3,2) = (-M_(3,0)*s[3] + M_(3,1)*s[1] - M_(3,2)*s[0]) * idet;
    LM_E(o,3,3) = ( M_(2,0)*s[3] - M_(2,1)*s[1] + M_(2,2)*s[0]) * idet;
#undef M_
    memcpy(t, o, sizeof(o));
}

/*
 * linmath.h:297-305 (mat4x4_mul_vec4_post):
 * t[i] = M[0][i]*v0 + M[1][i]*v1 + M[2][i]*v2;  t[i] += M[3][i]*v3
 */
static inline void lm_m4_mul_v4_post(float *r, const float *m, const float *v)
{
    float t[4];
    for (int i = 0; i < 4; i++) {
        t[i] = LM_E(m, 0, i) * v[0] + LM_E(m, 1, i) * v[1] + LM_E(m, 2, i) * v[2];
        t[i] += LM_E(m, 3, i) * v[3];
    }
    memcpy(r, t, sizeof(t));
}

/* linmath.h:317-325 (scalar mat4x4_transpose) */
static inline void lm_m4_transpose(float *out, const float *n)
{
    float t[16];
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            LM_E(t, i, j) = LM_E(n, j, i);
    memcpy(out, t, sizeof(t));
}

/* linmath.h:408-416 (scalar mat4x4_transpose_mat3x3): transpose the upper-left 3x3 */
static inline void lm_m4_transpose_3x3(float *m)
{
    float t[16];
    memcpy(t, m, sizeof(t));
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            LM_E(t, c, r) = LM_E(m, r, c);
    memcpy(m, t, sizeof(t));
}

/* linmath.h:857-870 (quat_from_euler_xyz) */
static inline void lm_quat_from_euler_xyz(float *q, float x, float y, float z)
{
    float cx = cosf(x * 0.5f), sx = sinf(x * 0.5f);
    float cy = cosf(y * 0.5f), sy = sinf(y * 0.5f);
    float cz = cosf(z * 0.5f), sz = sinf(z * 0.5f);

    q[0] = sx * cy * cz - cx * sy * sz;
    q[1] = cx * sy * cz + sx * cy * sz;
    q[2] = cx * cy * sz - sx * sy * cz;
    q[3] = cx * cy * cz + sx * sy * sz;
}

/* linmath.h:709-734 / 753-776: `a` is float(1.f / tan((double)(fov/2.f))) */
static inline void lm_m4_perspective(float *m, float y_fov, float aspect, float n, float f,
                                     int ndc_z_zero_one)
{
    float const a = 1.f / tan(y_fov / 2.f);

    memset(m, 0, 16 * sizeof(float));
    LM_E(m, 0, 0) = a / aspect;
    LM_E(m, 1, 1) = a;
    LM_E(m, 2, 3) = -1.f;
    if (ndc_z_zero_one) {
        LM_E(m, 2, 2) = -((f) / (f - n));
        LM_E(m, 3, 2) = -((f * n) / (f - n));
    } else {
        LM_E(m, 2, 2) = -((f + n) / (f - n));
        LM_E(m, 3, 2) = -((2.f * f * n) / (f - n));
    }
}

/*
 * interp.h:25-29,59-64: linf_interp is `a * (1.0 - blend) + b * blend` on floats: the `1.0`
 * literal makes the FIRST product and the sum double, but `b * blend` is a float * float
 * product (rounded to fp32) that is only then promoted for the addition.
 */
static inline float lm_lerp(float a, float b, float fac)
{
    float bf = b * fac;
    return (float)((double)a * (1.0 - (double)fac) + (double)bf);
}

/* linmath.h:58-62 (vec4_norm): k = float(1.0 / (double)len) */
static inline void lm_v4_norm(float *r, const float *v)
{
    float len = sqrtf(lm_dot4(v, v));
    float k = 1.0 / len;
    for (int i = 0; i < 4; i++)
        r[i] = v[i] * k;
}

/* interp.h:67-85 (quat_interp) */
static inline void lm_quat_nlerp(float *res, const float *a, const float *b, float fac)
{
    float dot = lm_dot4(a, b);
    float rfac = 1.f - fac;
    float t[4];

    if (dot < 0) {
        t[3] = rfac * a[3] - fac * b[3];
        t[0] = rfac * a[0] - fac * b[0];
        t[1] = rfac * a[1] - fac * b[1];
        t[2] = rfac * a[2] - fac * b[2];
    } else {
        t[3] = rfac * a[3] + fac * b[3];
        t[0] = rfac * a[0] + fac * b[0];
        t[1] = rfac * a[1] + fac * b[1];
        t[2] = rfac * a[2] + fac * b[2];
    }
    lm_v4_norm(res, t);
}

/*
 * interp.h:91-118 (quat_slerp).  Mixed precision is reproduced literally:
 * `dot < 0.0` / `dot > 0.9995` compare a float with a double literal;
 * acos/sin/cos are the double libm functions applied to float arguments and
 * stored back to float; `dot * sin_theta / sin_theta_0` is a float expression
 * promoted only for the subtraction from cos(theta).
 */
static inline void lm_quat_slerp(float *res, const float *a, const float *b, float fac)
{
    float dot = lm_dot4(a, b);
    float nb[4] = { b[0], b[1], b[2], b[3] };

    if (dot < 0.0) {
        dot = -dot;
        for (int i = 0; i < 4; i++) nb[i] = -b[i];
    }
    if (dot > 0.9995) {
        lm_quat_nlerp(res, a, nb, fac);
        return;
    }

    float theta_0 = acos(dot);
    float theta = fac * theta_0;
    float sin_theta = sin(theta);
    float sin_theta_0 = sin(theta_0);

    float rf = cos(theta) - dot * sin_theta / sin_theta_0;
    float f = sin_theta / sin_theta_0;
    for (int i = 0; i < 4; i++)
        res[i] = a[i] * rf + nb[i] * f;
}

#endif /* CLAP_ORACLE_LM_H */
