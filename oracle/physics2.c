/*
 * oracle/physics2.c -- TEST INFRASTRUCTURE ONLY (see clap_oracle.h).
 *
 * PARITY UNPINNED, like physics.c: everything below sits on ODE (submodule deps/ode, absent from
 * /root/reference, commit not recorded).  Restated from the reference's call sites and from ODE's
 * published sources (0.16 line) as far as they can be restated without the tree:
 *
 *   from the reference (core/physics.c):
 *     - phys_geom_capsule_new: capsule radius / length / yoffset / direction from the entity's
 *       AABB extents, sphere when length == 0, dMassSetCapsuleTotal / dMassSetSphereTotal   814-873
 *     - the capsule geoms' offset rotation dRFromAxisAndAngle(1,1,1, -2 pi / 3)            974-978
 *     - auto-disable parameters, gravity, damping (physics.c)                               1039-1042, 1125-1129
 *     - near_callback's use of dCollide + phys_contact_surface                              399-449, 291-330
 *     - phys_body_sweep_capsule: the marching probe and its arithmetic                      559-670
 *   from ODE (ode/src/mass.cpp, rotation.cpp, capsule.cpp, sphere.cpp, collision_util.cpp,
 *   quickstep.cpp, util.cpp):
 *     - dMassSetSphere / dMassSetCapsule + dMassAdjust
 *     - dQFromAxisAndAngle, dQtoR, dMultiply0_333 for the geom's final rotation (body R * offset R)
 *     - dxCapsule::computeAABB, dxSphere::computeAABB
 *     - dxQuickStepIsland stage 0 for a body without joints: world inertia R I R^T, the implicit
 *       gyroscopic torque (Lacoursiere 2006) of bodies with dxBodyGyroscopic, gravity; then
 *       lvel += h invM facc, avel += invI (h tacc); dxStepBody; damping
 *     - dInternalHandleAutoDisabling with its sample-averaging ring, and its rule that a body
 *       without joints is never put to sleep ("don't freeze objects mid-air")
 *     - dCollideSpheres, dCollideCapsuleSphere, dCollideCapsuleCapsule (with the two-contact
 *       parallel case), dClosestLineSegmentPoints, dCollideCapsuleBox + dClosestLineBoxPoints for
 *       an axis-aligned box; dCollide's reversal when only the swapped collider exists
 *   NOT restated: dBoxBox, which dCollideCapsuleBox falls into when the capsule's axis touches the
 *   box (closest points coincide): such pairs are flagged CLAPO_CONTACT_DEEP with nc = 0.  Contact
 *   joints and the SOR-LCP solve stay in ODE (its row order is randomised).
 *
 * Candidate pairs are the canonical ascending set; a pair's g1 is its first index.
 */
#include <stdlib.h>
#include <string.h>
#include "clap_oracle.h"
#include "lm.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- ODE rotation helpers (rotation.cpp) ---- */
static void q_from_axis_and_angle(double q[4], double ax, double ay, double az, double angle)
{
    double l = ax * ax + ay * ay + az * az;
    if (l > 0.0) {
        angle *= 0.5;
        q[0] = cos(angle);
        l = sin(angle) * (1.0 / sqrt(l));
        q[1] = ax * l; q[2] = ay * l; q[3] = az * l;
    } else {
        q[0] = 1; q[1] = q[2] = q[3] = 0;
    }
}

/* dQtoR: R is ODE's dMatrix3 (3 rows of 4, element (i,j) at [4 i + j]) */
static void q_to_R(const double q[4], double R[12])
{
    const double qq1 = 2 * q[1] * q[1], qq2 = 2 * q[2] * q[2], qq3 = 2 * q[3] * q[3];
    R[0] = 1 - qq2 - qq3;
    R[1] = 2 * (q[1] * q[2] - q[0] * q[3]);
    R[2] = 2 * (q[1] * q[3] + q[0] * q[2]);
    R[3] = 0;
    R[4] = 2 * (q[1] * q[2] + q[0] * q[3]);
    R[5] = 1 - qq1 - qq3;
    R[6] = 2 * (q[2] * q[3] - q[0] * q[1]);
    R[7] = 0;
    R[8] = 2 * (q[1] * q[3] - q[0] * q[2]);
    R[9] = 2 * (q[2] * q[3] + q[0] * q[1]);
    R[10] = 1 - qq1 - qq2;
    R[11] = 0;
}

/* physics.c:974-978: dRFromAxisAndAngle(R, 1.0, 1.0, 1.0, -M_PI * 2.0 / 3.0) */
void clapo_geom_offset_rotation(double R[12])
{
    double q[4];
    q_from_axis_and_angle(q, 1.0, 1.0, 1.0, -M_PI * 2.0 / 3.0);
    q_to_R(q, R);
}

/* dMassSetSphereTotal = dMassSetSphere(m, 1.0, r); dMassAdjust(m, total) */
void clapo_mass_sphere_total(double total_mass, double radius, double I[3])
{
    const double density = 1.0;
    const double m1 = (4.0 / 3.0) * M_PI * radius * radius * radius * density;
    const double II = 0.4 * m1 * radius * radius;
    const double scale = total_mass / m1;
    I[0] = I[1] = I[2] = II * scale;
}

/* dMassSetCapsuleTotal = dMassSetCapsule(m, 1.0, direction, a, b); dMassAdjust(m, total) */
void clapo_mass_capsule_total(double total_mass, int direction, double a, double b, double I[3])
{
    const double density = 1.0;
    const double M1 = M_PI * a * a * b * density;                       /* cylinder */
    const double M2 = (4.0 / 3.0) * M_PI * a * a * a * density;         /* the two caps */
    const double m = M1 + M2;
    const double Ia = M1 * (0.25 * a * a + (1.0 / 12.0) * b * b) + M2 * (0.4 * a * a + 0.375 * a * b + 0.25 * b * b);
    const double Ib = (M1 * 0.5 + M2 * 0.4) * a * a;
    const double scale = total_mass / m;
    I[0] = I[1] = I[2] = Ia;
    I[direction - 1] = Ib;
    I[0] *= scale; I[1] *= scale; I[2] *= scale;
}

/* physics.c:814-873.  min3/max/xmax3 as the reference's util.h macros; `direction` keeps the value the
 * reference hands to dMassSetCapsuleTotal (case 1 falls through to case 2 without changing it). */
void clapo_capsule_geom(float X, float Y, float Z, double geom_radius, double geom_offset,
                        float *pr, float *plength, float *pyoffset, int *pdirection, float *pray_off)
{
    float r = 0.0f, length = 0.0f, off = 0.0f, ray_off = 0.0f;
    float mx3 = Y > Z ? Y : Z;                                          /* xmax3 (util.h:204-209): max3, then == Y wins over == Z */
    if (X > mx3) mx3 = X;
    int xm = 0;
    if (mx3 == Y) xm = 1; else if (mx3 == Z) xm = 2;
    int direction = xm + 1;
    switch (direction) {
    case 1:
    case 2: {
        float mn = Y < Z ? Y : Z;                                       /* min3 = min(a, min(b, c)) */
        if (X < mn) mn = X;
        r = geom_radius ? (float)geom_radius : mn / 2;
        float l = Y / 2 - r * 2;
        length = l > 0 ? l : 0;
        off = geom_offset ? (float)geom_offset : Y / 2;
        ray_off = r + length / 2;
        break;
    }
    case 3:
        r = geom_radius ? (float)geom_radius : X / 2;
        length = Z - r * 2;
        off = geom_offset ? (float)geom_offset : (Y - r * 2) / 2;
        ray_off = r;
        break;
    }
    *pr = r; *plength = length; *pyoffset = off; *pdirection = direction; *pray_off = ray_off;
}

/* column 2 of (R_body * R_offset): dMultiply0_333 = row . column, summed left to right */
static void capsule_axis(const double q[4], const double Roff[12], double axis[3])
{
    double R[12];
    q_to_R(q, R);
    for (int i = 0; i < 3; i++)
        axis[i] = R[4 * i] * Roff[2] + R[4 * i + 1] * Roff[6] + R[4 * i + 2] * Roff[10];
}

/* dxSphere::computeAABB / dxCapsule::computeAABB */
static void geom_aabb(const double *p, double radius, double lz, const double axis[3], double *bb)
{
    if (lz == 0.0) {
        for (int a = 0; a < 3; a++) { bb[2 * a] = p[a] - radius; bb[2 * a + 1] = p[a] + radius; }
        return;
    }
    for (int a = 0; a < 3; a++) {
        const double range = fabs(axis[a] * lz) * 0.5 + radius;
        bb[2 * a] = p[a] - range;
        bb[2 * a + 1] = p[a] + range;
    }
}

void clapo_bodies_aabb(const clapo_bodies *b)
{
    for (uint32_t i = 0; i < b->n; i++) {
        double axis[3] = { 0, 0, 1 };
        const double lz = b->length ? b->length[i] : 0.0;
        capsule_axis(b->quat + 4 * (size_t)i, b->geom_offset_R, axis);
        if (b->axis) memcpy(b->axis + 3 * (size_t)i, axis, sizeof(axis));
        if (b->aabb) geom_aabb(b->pos + 3 * (size_t)i, b->radius[i], lz, axis, b->aabb + 6 * (size_t)i);
    }
}

/* ---- 3x3 helpers on dMatrix3 ---- */
static double det3(const double *m)
{
    return m[0] * (m[5] * m[10] - m[9] * m[6]) - m[1] * (m[4] * m[10] - m[8] * m[6]) + m[2] * (m[4] * m[9] - m[8] * m[5]);
}

static double invert3(double *dst, const double *ma)                    /* dInvertMatrix3 */
{
    const double det = det3(ma);
    if (det == 0) return 0;
    const double r = 1.0 / det;
    dst[0] = (ma[5] * ma[10] - ma[6] * ma[9]) * r;
    dst[1] = (ma[9] * ma[2] - ma[1] * ma[10]) * r;
    dst[2] = (ma[1] * ma[6] - ma[5] * ma[2]) * r;
    dst[4] = (ma[6] * ma[8] - ma[4] * ma[10]) * r;
    dst[5] = (ma[0] * ma[10] - ma[8] * ma[2]) * r;
    dst[6] = (ma[4] * ma[2] - ma[0] * ma[6]) * r;
    dst[8] = (ma[4] * ma[9] - ma[8] * ma[5]) * r;
    dst[9] = (ma[8] * ma[1] - ma[0] * ma[9]) * r;
    dst[10] = (ma[0] * ma[5] - ma[1] * ma[4]) * r;
    dst[3] = dst[7] = dst[11] = 0;
    return det;
}

/* W = R diag(d) R^T as ODE does it: tmp = D R^T (dMultiply2_333), W = R tmp (dMultiply0_333) */
static void world_tensor(const double R[12], const double d[3], double W[12])
{
    double tmp[12];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            tmp[4 * i + j] = d[i] * R[4 * j + i];                        /* row i of D has one non-zero */
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            W[4 * i + j] = R[4 * i] * tmp[j] + R[4 * i + 1] * tmp[4 + j] + R[4 * i + 2] * tmp[8 + j];
        W[4 * i + 3] = 0;
    }
}

static void mul331(double out[3], const double *M, const double v[3])
{
    for (int i = 0; i < 3; i++)
        out[i] = M[4 * i] * v[0] + M[4 * i + 1] * v[1] + M[4 * i + 2] * v[2];
}

/* one dWorldQuickStep(world, h) for bodies without constraint rows */
void clapo_bodies_step2(const clapo_bodies *b, const clapo_world *w, double h)
{
    const uint32_t S = b->adis_average_samples > 1 ? b->adis_average_samples : 1;
    for (uint32_t i = 0; i < b->n; i++) {
        double *p = b->pos + 3 * (size_t)i, *q = b->quat + 4 * (size_t)i;
        double *v = b->lvel + 3 * (size_t)i, *om = b->avel + 3 * (size_t)i;
        uint32_t fl = b->bflags[i];
        const double lz = b->length ? b->length[i] : 0.0;

        if (fl & CLAPO_BODY_DISABLED)
            continue;
        /* dInternalHandleAutoDisabling: only bodies that hold a joint, enabled, with the flag */
        if ((fl & CLAPO_BODY_AUTO_DISABLE) && (fl & CLAPO_BODY_HAS_JOINT)) {
            int idle = 0;
            double al[3], aa[3];
            if (S == 1) {                                                /* counter wraps at once: the sample itself */
                memcpy(al, v, sizeof(al));
                memcpy(aa, om, sizeof(aa));
                idle = 1;
            } else {
                double *ring = b->adis_samples + (size_t)i * S * 6;
                uint32_t c = b->adis_counter[i] & 0x7fffffffu, ready = b->adis_counter[i] >> 31;
                memcpy(ring + 6 * (size_t)c, v, 3 * sizeof(double));
                memcpy(ring + 6 * (size_t)c + 3, om, 3 * sizeof(double));
                if (++c >= S) { c = 0; ready = 1; }
                b->adis_counter[i] = c | ready << 31;
                if (ready) {
                    idle = 1;
                    memcpy(al, ring, sizeof(al));
                    memcpy(aa, ring + 3, sizeof(aa));
                    for (uint32_t s = 1; s < S; s++)
                        for (int a = 0; a < 3; a++) { al[a] += ring[6 * (size_t)s + a]; aa[a] += ring[6 * (size_t)s + 3 + a]; }
                    const double r1 = 1.0 / (double)S;
                    for (int a = 0; a < 3; a++) { al[a] *= r1; aa[a] *= r1; }
                }
            }
            if (idle) {
                if (al[0] * al[0] + al[1] * al[1] + al[2] * al[2] > w->adis_linear_threshold_sq)
                    idle = 0;
                else if (aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2] > w->adis_angular_threshold_sq)
                    idle = 0;
            }
            if (idle) {
                b->adis_steps_left[i]--;
                b->adis_time_left[i] -= h;
            } else {
                b->adis_steps_left[i] = w->adis_steps;
                b->adis_time_left[i] = w->adis_time;
            }
            if (b->adis_steps_left[i] <= 0 && b->adis_time_left[i] <= 0) {
                b->bflags[i] = (fl | CLAPO_BODY_DISABLED) & ~CLAPO_BODY_HAS_JOINT;
                v[0] = v[1] = v[2] = 0;
                om[0] = om[1] = om[2] = 0;
                continue;
            }
        }
        b->bflags[i] = fl & ~CLAPO_BODY_HAS_JOINT;                        /* dJointGroupEmpty after the step */

        /* stage 0: torque accumulator from the gyroscopic term, force accumulator from gravity */
        double tacc[3] = { 0, 0, 0 };
        double invIw[12];
        int have_inertia = b->inertia != NULL;
        if (have_inertia) {
            const double *Ib = b->inertia + 3 * (size_t)i;
            const double invIb[3] = { 1.0 / Ib[0], 1.0 / Ib[1], 1.0 / Ib[2] };
            double R[12], Iw[12];
            q_to_R(q, R);
            world_tensor(R, invIb, invIw);
            if (fl & CLAPO_BODY_GYROSCOPIC) {
                double L[3], Itild[12] = { 0 }, itInv[12];
                world_tensor(R, Ib, Iw);
                mul331(L, Iw, om);
                /* dSetCrossMatrixMinus(Itild, L, 4) */
                Itild[1] = L[2]; Itild[2] = -L[1];
                Itild[4] = -L[2]; Itild[6] = L[0];
                Itild[8] = L[1]; Itild[9] = -L[0];
                for (int k = 0; k < 12; k++)
                    Itild[k] = Itild[k] * h + Iw[k];
                const double rh = 1.0 / h;
                L[0] *= rh; L[1] *= rh; L[2] *= rh;
                if (invert3(itInv, Itild) != 0) {
                    double T[12];
                    for (int r = 0; r < 3; r++)
                        for (int c = 0; c < 3; c++)
                            T[4 * r + c] = Iw[4 * r] * itInv[c] + Iw[4 * r + 1] * itInv[4 + c] + Iw[4 * r + 2] * itInv[8 + c];
                    T[0] -= 1; T[5] -= 1; T[10] -= 1;
                    double tau0[3];
                    mul331(tau0, T, L);
                    tacc[0] += tau0[0]; tacc[1] += tau0[1]; tacc[2] += tau0[2];
                }
            }
        }
        const double m = b->mass[i];
        const double k = h * (1.0 / m);
        for (int j = 0; j < 3; j++) {
            const double f = (fl & CLAPO_BODY_NO_GRAVITY) ? 0.0 : m * w->gravity[j];
            v[j] += k * f;
        }
        if (have_inertia) {
            double d[3];
            tacc[0] *= h; tacc[1] *= h; tacc[2] *= h;
            mul331(d, invIw, tacc);
            om[0] += d[0]; om[1] += d[1]; om[2] += d[2];
        }
        /* dxStepBody */
        for (int j = 0; j < 3; j++)
            p[j] += h * v[j];
        double dq[4];
        dq[0] = 0.5 * (-om[0] * q[1] - om[1] * q[2] - om[2] * q[3]);
        dq[1] = 0.5 * ( om[0] * q[0] + om[1] * q[3] - om[2] * q[2]);
        dq[2] = 0.5 * (-om[0] * q[3] + om[1] * q[0] + om[2] * q[1]);
        dq[3] = 0.5 * ( om[0] * q[2] - om[1] * q[1] + om[2] * q[0]);
        for (int j = 0; j < 4; j++)
            q[j] += h * dq[j];
        double l = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
        if (l > 0) {
            l = 1.0 / sqrt(l);
            for (int j = 0; j < 4; j++) q[j] *= l;
        } else {
            q[0] = 1; q[1] = q[2] = q[3] = 0;
        }
        if (w->linear_damping != 0.0) {
            const double speed2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
            if (speed2 > w->linear_damping_threshold_sq) {
                const double s = 1 - w->linear_damping;
                v[0] *= s; v[1] *= s; v[2] *= s;
            }
        }
        /* the geom moved with its body: new axis and AABB (ODE recomputes them lazily before the next collide) */
        double axis[3];
        capsule_axis(q, b->geom_offset_R, axis);
        if (b->axis) memcpy(b->axis + 3 * (size_t)i, axis, sizeof(axis));
        if (b->aabb) geom_aabb(p, b->radius[i], lz, axis, b->aabb + 6 * (size_t)i);
    }
}

/* ---- broadphase over explicit AABBs (sweep and prune on x: deliberately not the GPU's grid) ---- */
static int aabb_overlap(const double *a, const double *b)
{
    return !(a[0] > b[1] || a[1] < b[0] || a[2] > b[3] || a[3] < b[2] || a[4] > b[5] || a[5] < b[4]);
}

struct sweep_ent { double lo; uint32_t id; };
static int sweep_cmp(const void *a, const void *b)
{
    const struct sweep_ent *x = a, *y = b;
    return x->lo < y->lo ? -1 : x->lo > y->lo ? 1 : (x->id < y->id ? -1 : x->id > y->id);
}
static int pair_cmp(const void *a, const void *b)
{
    const uint32_t *x = a, *y = b;
    return x[0] != y[0] ? (x[0] < y[0] ? -1 : 1) : (x[1] < y[1] ? -1 : x[1] > y[1]);
}

uint64_t clapo_broadphase_aabb_pairs(uint32_t n, const double *bb, uint32_t *pairs, uint64_t max_pairs)
{
    struct sweep_ent *s = malloc(sizeof(*s) * (n ? n : 1));
    uint64_t count = 0;
    for (uint32_t i = 0; i < n; i++) { s[i].lo = bb[6 * (size_t)i]; s[i].id = i; }
    qsort(s, n, sizeof(*s), sweep_cmp);
    for (uint32_t a = 0; a < n; a++) {
        const double *ba = bb + 6 * (size_t)s[a].id;
        for (uint32_t c = a + 1; c < n && !(s[c].lo > ba[1]); c++) {
            if (!aabb_overlap(ba, bb + 6 * (size_t)s[c].id))
                continue;
            if (count < max_pairs) {
                const uint32_t i = s[a].id, j = s[c].id;
                pairs[2 * count] = i < j ? i : j;
                pairs[2 * count + 1] = i < j ? j : i;
            }
            count++;
        }
    }
    qsort(pairs, count < max_pairs ? count : max_pairs, 2 * sizeof(uint32_t), pair_cmp);
    free(s);
    return count;
}

/* pairs (body, static), ascending by (body, static): sweep over the union on x */
uint64_t clapo_broadphase_aabb_static_pairs(uint32_t n_static, const double *sbb, uint32_t n, const double *bb,
                                            uint32_t *pairs, uint64_t max_pairs)
{
    const uint32_t tot = n + n_static;
    struct sweep_ent *s = malloc(sizeof(*s) * (tot ? tot : 1));
    uint64_t count = 0;
    for (uint32_t i = 0; i < n; i++) { s[i].lo = bb[6 * (size_t)i]; s[i].id = i; }
    for (uint32_t i = 0; i < n_static; i++) { s[n + i].lo = sbb[6 * (size_t)i]; s[n + i].id = n + i; }
    qsort(s, tot, sizeof(*s), sweep_cmp);
    for (uint32_t a = 0; a < tot; a++) {
        const int a_static = s[a].id >= n;
        const double *ba = a_static ? sbb + 6 * (size_t)(s[a].id - n) : bb + 6 * (size_t)s[a].id;
        for (uint32_t c = a + 1; c < tot && !(s[c].lo > ba[1]); c++) {
            const int c_static = s[c].id >= n;
            if (a_static == c_static)
                continue;
            const double *bc = c_static ? sbb + 6 * (size_t)(s[c].id - n) : bb + 6 * (size_t)s[c].id;
            if (!aabb_overlap(ba, bc))
                continue;
            if (count < max_pairs) {
                pairs[2 * count] = a_static ? s[c].id : s[a].id;
                pairs[2 * count + 1] = (a_static ? s[a].id : s[c].id) - n;
            }
            count++;
        }
    }
    qsort(pairs, count < max_pairs ? count : max_pairs, 2 * sizeof(uint32_t), pair_cmp);
    free(s);
    return count;
}

/* ---- narrowphase ---- */
typedef struct { double pos[3], normal[3], depth; } cgeom;

/* dCollideSpheres (sphere.cpp) */
static int collide_spheres(const double *p1, double r1, const double *p2, double r2, cgeom *c)
{
    const double dx = p1[0] - p2[0], dy = p1[1] - p2[1], dz = p1[2] - p2[2];
    const double d = sqrt(dx * dx + dy * dy + dz * dz);
    if (d > r1 + r2) return 0;
    if (d <= 0) {
        c->pos[0] = p1[0]; c->pos[1] = p1[1]; c->pos[2] = p1[2];
        c->normal[0] = 1; c->normal[1] = 0; c->normal[2] = 0;
        c->depth = r1 + r2;
    } else {
        const double d1 = 1.0 / d;
        c->normal[0] = dx * d1; c->normal[1] = dy * d1; c->normal[2] = dz * d1;
        const double k = 0.5 * (r2 - r1 - d);
        c->pos[0] = p1[0] + c->normal[0] * k;
        c->pos[1] = p1[1] + c->normal[1] * k;
        c->pos[2] = p1[2] + c->normal[2] * k;
        c->depth = r1 + r2 - d;
    }
    return 1;
}

/* dCollideCapsuleSphere (capsule.cpp): o1 capsule, o2 sphere */
static int collide_capsule_sphere(const double *cp, const double *ax, double cr, double lz, const double *sp, double sr, cgeom *c)
{
    double alpha = ax[0] * (sp[0] - cp[0]) + ax[1] * (sp[1] - cp[1]) + ax[2] * (sp[2] - cp[2]);
    const double lz2 = lz * 0.5;
    if (alpha > lz2) alpha = lz2;
    if (alpha < -lz2) alpha = -lz2;
    const double p[3] = { cp[0] + alpha * ax[0], cp[1] + alpha * ax[1], cp[2] + alpha * ax[2] };
    return collide_spheres(p, cr, sp, sr, c);
}

#define DOT3(a, b) ((a)[0] * (b)[0] + (a)[1] * (b)[1] + (a)[2] * (b)[2])

/* dClosestLineSegmentPoints (collision_util.cpp) */
static void closest_segment_points(const double *a1, const double *a2, const double *b1, const double *b2, double *cp1, double *cp2)
{
    double a1a2[3], b1b2[3], a1b1[3], a1b2[3], a2b1[3], a2b2[3], n[3];
    double la, lb, k, da1, da2, da3, da4, db1, db2, db3, db4, det;
#define SET2(a, b) do { (a)[0] = (b)[0]; (a)[1] = (b)[1]; (a)[2] = (b)[2]; } while (0)
#define SUB3(a, b, c) do { (a)[0] = (b)[0] - (c)[0]; (a)[1] = (b)[1] - (c)[1]; (a)[2] = (b)[2] - (c)[2]; } while (0)
    SUB3(a1a2, a2, a1);
    SUB3(b1b2, b2, b1);
    SUB3(a1b1, b1, a1);
    da1 = DOT3(a1a2, a1b1);
    db1 = DOT3(b1b2, a1b1);
    if (da1 <= 0 && db1 >= 0) { SET2(cp1, a1); SET2(cp2, b1); return; }
    SUB3(a1b2, b2, a1);
    da2 = DOT3(a1a2, a1b2);
    db2 = DOT3(b1b2, a1b2);
    if (da2 <= 0 && db2 <= 0) { SET2(cp1, a1); SET2(cp2, b2); return; }
    SUB3(a2b1, b1, a2);
    da3 = DOT3(a1a2, a2b1);
    db3 = DOT3(b1b2, a2b1);
    if (da3 >= 0 && db3 >= 0) { SET2(cp1, a2); SET2(cp2, b1); return; }
    SUB3(a2b2, b2, a2);
    da4 = DOT3(a1a2, a2b2);
    db4 = DOT3(b1b2, a2b2);
    if (da4 >= 0 && db4 <= 0) { SET2(cp1, a2); SET2(cp2, b2); return; }
    la = DOT3(a1a2, a1a2);
    if (da1 >= 0 && da3 <= 0) {
        k = da1 / la;
        for (int i = 0; i < 3; i++) n[i] = a1b1[i] - k * a1a2[i];
        if (DOT3(b1b2, n) >= 0) {
            for (int i = 0; i < 3; i++) cp1[i] = a1[i] + k * a1a2[i];
            SET2(cp2, b1);
            return;
        }
    }
    if (da2 >= 0 && da4 <= 0) {
        k = da2 / la;
        for (int i = 0; i < 3; i++) n[i] = a1b2[i] - k * a1a2[i];
        if (DOT3(b1b2, n) <= 0) {
            for (int i = 0; i < 3; i++) cp1[i] = a1[i] + k * a1a2[i];
            SET2(cp2, b2);
            return;
        }
    }
    lb = DOT3(b1b2, b1b2);
    if (db1 <= 0 && db2 >= 0) {
        k = -db1 / lb;
        for (int i = 0; i < 3; i++) n[i] = -a1b1[i] - k * b1b2[i];
        if (DOT3(a1a2, n) >= 0) {
            SET2(cp1, a1);
            for (int i = 0; i < 3; i++) cp2[i] = b1[i] + k * b1b2[i];
            return;
        }
    }
    if (db3 <= 0 && db4 >= 0) {
        k = -db3 / lb;
        for (int i = 0; i < 3; i++) n[i] = -a2b1[i] - k * b1b2[i];
        if (DOT3(a1a2, n) <= 0) {
            SET2(cp1, a2);
            for (int i = 0; i < 3; i++) cp2[i] = b1[i] + k * b1b2[i];
            return;
        }
    }
    k = DOT3(a1a2, b1b2);
    det = la * lb - k * k;
    if (det <= 0) { SET2(cp1, a1); SET2(cp2, b1); return; }
    det = 1.0 / det;
    const double alpha = (lb * da1 - k * db1) * det;
    const double beta = (k * da1 - la * db1) * det;
    for (int i = 0; i < 3; i++) cp1[i] = a1[i] + alpha * a1a2[i];
    for (int i = 0; i < 3; i++) cp2[i] = b1[i] + beta * b1b2[i];
}

/* dCollideCapsuleCapsule (capsule.cpp); up to two contacts */
static int collide_capsule_capsule(const double *pos1, const double *ax1, double r1, double l1,
                                   const double *pos2, const double *ax2in, double r2, double l2, cgeom *c)
{
    const double tolerance = 1e-5;
    const double lz1 = l1 * 0.5, lz2 = l2 * 0.5;
    double axis2[3] = { ax2in[0], ax2in[1], ax2in[2] };
    double sphere1[3], sphere2[3];
    const double a1a2 = DOT3(ax1, axis2);
    const double det = 1.0 - a1a2 * a1a2;
    if (det < tolerance) {
        if (a1a2 < 0) { axis2[0] = -axis2[0]; axis2[1] = -axis2[1]; axis2[2] = -axis2[2]; }
        const double q[3] = { pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2] };
        const double k = DOT3(ax1, q);
        const double a1lo = -lz1, a1hi = lz1, a2lo = -lz2 - k, a2hi = lz2 - k;
        const double lo = a1lo > a2lo ? a1lo : a2lo;
        const double hi = a1hi < a2hi ? a1hi : a2hi;
        if (lo <= hi) {
            if (lo < hi) {                                               /* MAX_CONTACTS = 16 >= 2 */
                for (int i = 0; i < 3; i++) sphere1[i] = pos1[i] + lo * ax1[i];
                for (int i = 0; i < 3; i++) sphere2[i] = pos2[i] + (lo + k) * axis2[i];
                const int n1 = collide_spheres(sphere1, r1, sphere2, r2, c);
                if (n1) {
                    for (int i = 0; i < 3; i++) sphere1[i] = pos1[i] + hi * ax1[i];
                    for (int i = 0; i < 3; i++) sphere2[i] = pos2[i] + (hi + k) * axis2[i];
                    const int n2 = collide_spheres(sphere1, r1, sphere2, r2, c + 1);
                    if (n2) return 2;
                }
            }
            const double alpha1 = (lo + hi) * 0.5, alpha2 = alpha1 + k;
            for (int i = 0; i < 3; i++) sphere1[i] = pos1[i] + alpha1 * ax1[i];
            for (int i = 0; i < 3; i++) sphere2[i] = pos2[i] + alpha2 * axis2[i];
            return collide_spheres(sphere1, r1, sphere2, r2, c);
        }
    }
    double a1[3], a2[3], b1[3], b2[3];
    for (int i = 0; i < 3; i++) {
        a1[i] = pos1[i] + ax1[i] * lz1;
        a2[i] = pos1[i] - ax1[i] * lz1;
        b1[i] = pos2[i] + axis2[i] * lz2;
        b2[i] = pos2[i] - axis2[i] * lz2;
    }
    closest_segment_points(a1, a2, b1, b2, sphere1, sphere2);
    return collide_spheres(sphere1, r1, sphere2, r2, c);
}

/* dClosestLineBoxPoints (collision_util.cpp) for a box with R = identity */
static void closest_line_box_points(const double *p1, const double *p2, const double *c, const double *side, double *lret, double *bret)
{
    double tmp[3], s[3], v[3], sign[3], v2[3], h[3], tanchor[3];
    int region[3];
    const double tanchor_eps = 1e-307;
    for (int i = 0; i < 3; i++) { s[i] = p1[i] - c[i]; tmp[i] = p2[i] - p1[i]; v[i] = tmp[i]; }
    for (int i = 0; i < 3; i++) {
        if (v[i] < 0) { s[i] = -s[i]; v[i] = -v[i]; sign[i] = -1; }
        else sign[i] = 1;
    }
    for (int i = 0; i < 3; i++) { v2[i] = v[i] * v[i]; h[i] = 0.5 * side[i]; }
    for (int i = 0; i < 3; i++) {
        if (v[i] > tanchor_eps) {
            if (s[i] < -h[i]) { region[i] = -1; tanchor[i] = (-h[i] - s[i]) / v[i]; }
            else { region[i] = (s[i] > h[i]); tanchor[i] = (h[i] - s[i]) / v[i]; }
        } else { region[i] = 0; tanchor[i] = 2; }
    }
    double t = 0, dd2dt = 0;
    for (int i = 0; i < 3; i++) dd2dt -= (region[i] ? v2[i] : 0) * tanchor[i];
    if (dd2dt >= 0) goto got_answer;
    do {
        double next_t = 1;
        for (int i = 0; i < 3; i++)
            if (tanchor[i] > t && tanchor[i] < 1 && tanchor[i] < next_t) next_t = tanchor[i];
        double next_dd2dt = 0;
        for (int i = 0; i < 3; i++) next_dd2dt += (region[i] ? v2[i] : 0) * (next_t - tanchor[i]);
        if (next_dd2dt >= 0) {
            const double m = (next_dd2dt - dd2dt) / (next_t - t);
            t -= dd2dt / m;
            goto got_answer;
        }
        for (int i = 0; i < 3; i++)
            if (tanchor[i] == next_t) { tanchor[i] = (h[i] - s[i]) / v[i]; region[i]++; }
        t = next_t;
        dd2dt = next_dd2dt;
    } while (t < 1);
    t = 1;
got_answer:
    for (int i = 0; i < 3; i++) lret[i] = p1[i] + t * tmp[i];
    for (int i = 0; i < 3; i++) {
        double x = sign[i] * (s[i] + t * v[i]);
        if (x < -h[i]) x = -h[i];
        else if (x > h[i]) x = h[i];
        bret[i] = x + c[i];                                              /* R = identity: dMultiply0_331 is the copy */
    }
}

/* dCollideCapsuleBox (capsule.cpp), box axis-aligned; returns -1 for the dBoxBox branch */
static int collide_capsule_box(const double *cp, const double *ax, double radius, double lz, const double *bb, cgeom *c)
{
    const double clen = lz * 0.5;
    double p1[3], p2[3], bc[3], side[3], pl[3], pb[3];
    for (int i = 0; i < 3; i++) {
        p1[i] = cp[i] + clen * ax[i];
        p2[i] = cp[i] - clen * ax[i];
        bc[i] = (bb[2 * i] + bb[2 * i + 1]) * 0.5;
        side[i] = bb[2 * i + 1] - bb[2 * i];
    }
    closest_line_box_points(p1, p2, bc, side, pl, pb);
    const double dx = pl[0] - pb[0], dy = pl[1] - pb[1], dz = pl[2] - pb[2];
    if (sqrt(dx * dx + dy * dy + dz * dz) < 1e-15)
        return -1;
    return collide_spheres(pl, radius, pb, 0, c);
}

/* dCollideSphereBox for an axis-aligned box (physics.c of round 1 restates it; same arithmetic) */
static void safe_normalize3(double a[3])
{
    const double aa[3] = { fabs(a[0]), fabs(a[1]), fabs(a[2]) };
    int idx;
    if (aa[1] > aa[0]) idx = aa[2] > aa[1] ? 2 : 1;
    else if (aa[2] > aa[0]) idx = 2;
    else {
        if (aa[0] <= 0) { a[0] = 1; a[1] = 0; a[2] = 0; return; }
        idx = 0;
    }
    a[0] /= aa[idx]; a[1] /= aa[idx]; a[2] /= aa[idx];
    const double l = 1.0 / sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    a[0] *= l; a[1] *= l; a[2] *= l;
}

static int collide_sphere_box(const double *c0, double rad, const double *bb, cgeom *c)
{
    double bp[3], l[3], p[3], t[3];
    int onborder = 0;
    memset(c, 0, sizeof(*c));
    for (int a = 0; a < 3; a++) {
        bp[a] = (bb[2 * a] + bb[2 * a + 1]) * 0.5;
        l[a] = (bb[2 * a + 1] - bb[2 * a]) * 0.5;
        p[a] = c0[a] - bp[a];
        t[a] = p[a];
        if (t[a] < -l[a]) { t[a] = -l[a]; onborder = 1; }
        if (t[a] > l[a]) { t[a] = l[a]; onborder = 1; }
    }
    if (!onborder) {
        double min_distance = l[0] - fabs(t[0]);
        int mini = 0;
        for (int a = 1; a < 3; a++) {
            const double fd = l[a] - fabs(t[a]);
            if (fd < min_distance) { min_distance = fd; mini = a; }
        }
        c->pos[0] = c0[0]; c->pos[1] = c0[1]; c->pos[2] = c0[2];
        c->normal[mini] = t[mini] > 0 ? 1.0 : -1.0;
        c->depth = min_distance + rad;
        return 1;
    }
    double r[3] = { p[0] - t[0], p[1] - t[1], p[2] - t[2] };
    const double depth = rad - sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (depth < 0) return 0;
    c->pos[0] = t[0] + bp[0]; c->pos[1] = t[1] + bp[1]; c->pos[2] = t[2] + bp[2];
    safe_normalize3(r);
    c->normal[0] = r[0]; c->normal[1] = r[1]; c->normal[2] = r[2];
    c->depth = depth;
    return 1;
}

static int geom_kind(const clapo_geoms *g, uint32_t i)
{
    if (g->kind) return g->kind[i];
    return (g->length && g->length[i] != 0.0) ? CLAPO_GEOM_CAPSULE : CLAPO_GEOM_SPHERE;
}

static const double zero3[3] = { 0, 0, 0 };

/* dCollide(o1 = A[ia], o2 = B[ib]): the class pair's collider, swapped and reversed (normals negated)
 * when only the swapped one exists (collision_kernel.cpp).  Returns nc, -1 for the unrestated dBoxBox
 * branch, 0 for class pairs without a narrowphase here. */
static int collide(const clapo_geoms *A, uint32_t ia, const clapo_geoms *B, uint32_t ib, cgeom *c)
{
    const int ka = geom_kind(A, ia), kb = geom_kind(B, ib);
    const double *pa = A->pos ? A->pos + 3 * (size_t)ia : zero3, *pb = B->pos ? B->pos + 3 * (size_t)ib : zero3;
    const double *xa = A->axis ? A->axis + 3 * (size_t)ia : zero3, *xb = B->axis ? B->axis + 3 * (size_t)ib : zero3;
    const double ra = A->radius ? A->radius[ia] : 0, rb = B->radius ? B->radius[ib] : 0;
    const double la = A->length ? A->length[ia] : 0, lb = B->length ? B->length[ib] : 0;
    int nc = 0, reverse = 0;

    if (ka == CLAPO_GEOM_SPHERE && kb == CLAPO_GEOM_SPHERE) nc = collide_spheres(pa, ra, pb, rb, c);
    else if (ka == CLAPO_GEOM_CAPSULE && kb == CLAPO_GEOM_SPHERE) nc = collide_capsule_sphere(pa, xa, ra, la, pb, rb, c);
    else if (ka == CLAPO_GEOM_SPHERE && kb == CLAPO_GEOM_CAPSULE) { nc = collide_capsule_sphere(pb, xb, rb, lb, pa, ra, c); reverse = 1; }
    else if (ka == CLAPO_GEOM_CAPSULE && kb == CLAPO_GEOM_CAPSULE) nc = collide_capsule_capsule(pa, xa, ra, la, pb, xb, rb, lb, c);
    else if (ka == CLAPO_GEOM_SPHERE && kb == CLAPO_GEOM_BOX) nc = collide_sphere_box(pa, ra, B->aabb + 6 * (size_t)ib, c);
    else if (ka == CLAPO_GEOM_BOX && kb == CLAPO_GEOM_SPHERE) { nc = collide_sphere_box(pb, rb, A->aabb + 6 * (size_t)ia, c); reverse = 1; }
    else if (ka == CLAPO_GEOM_CAPSULE && kb == CLAPO_GEOM_BOX) nc = collide_capsule_box(pa, xa, ra, la, B->aabb + 6 * (size_t)ib, c);
    else if (ka == CLAPO_GEOM_BOX && kb == CLAPO_GEOM_CAPSULE) { nc = collide_capsule_box(pb, xb, rb, lb, A->aabb + 6 * (size_t)ia, c); reverse = 1; }
    if (reverse)
        for (int k = 0; k < nc; k++)
            for (int a = 0; a < 3; a++) c[k].normal[a] = -c[k].normal[a];
    return nc;
}

#define CLAPO_CONTACT_BOUNCE   0x004
#define CLAPO_CONTACT_SOFT_ERP 0x008
#define CLAPO_CONTACT_SOFT_CFM 0x010

uint32_t clapo_contacts_geoms(uint32_t n_pairs, const uint32_t *pairs, const clapo_geoms *A, const clapo_geoms *B,
                              clapo_contact2 *out)
{
    uint32_t total = 0;
    for (uint32_t k = 0; k < n_pairs; k++) {
        const uint32_t ia = pairs[2 * k], ib = pairs[2 * k + 1];
        clapo_contact2 *c = out + k;
        cgeom g[2];
        memset(c, 0, sizeof(*c));
        memset(g, 0, sizeof(g));
        const int nc = collide(A, ia, B, ib, g);
        if (nc < 0) { c->nc = CLAPO_CONTACT_DEEP; total++; }
        if (nc <= 0) continue;
        memcpy(c->pos, g[0].pos, sizeof(c->pos)); memcpy(c->normal, g[0].normal, sizeof(c->normal)); c->depth = g[0].depth;
        if (nc > 1) { memcpy(c->pos2, g[1].pos, sizeof(c->pos2)); memcpy(c->normal2, g[1].normal, sizeof(c->normal2)); c->depth2 = g[1].depth; }
        /* phys_contact_surface (physics.c:291-330) */
        double bounce = 0, bounce_vel = 0, mu = 0, soft_erp = 0.05, soft_cfm = 0.01;
        if (A->material && B->material) {
            const double *m1 = A->material + 5 * (size_t)ia, *m2 = B->material + 5 * (size_t)ib;
            bounce = fmax(m1[0], m2[0]);
            bounce_vel = (m1[1] + m2[1]) * 0.5;
            mu = sqrt(m1[2] * m2[2]);
            if (m1[3] > 0 && m2[3] > 0) soft_erp = fmin(m1[3], m2[3]);
            else if (m1[3] > 0) soft_erp = m1[3];
            else if (m2[3] > 0) soft_erp = m2[3];
            if (m1[4] > 0 && m2[4] > 0) soft_cfm = fmax(m1[4], m2[4]);
            else if (m1[4] > 0) soft_cfm = m1[4];
            else if (m2[4] > 0) soft_cfm = m2[4];
        }
        c->mode = CLAPO_CONTACT_SOFT_CFM | CLAPO_CONTACT_SOFT_ERP | (bounce > 0 ? CLAPO_CONTACT_BOUNCE : 0);
        c->mu = mu; c->bounce = bounce; c->bounce_vel = bounce_vel; c->soft_erp = soft_erp; c->soft_cfm = soft_cfm;
        c->nc = (uint32_t)nc;
        total++;
    }
    return total;
}

/*
 * phys_body_sweep_capsule (physics.c:559-670).  The probe is a copy of the body's geom (same radius,
 * length and rotation) marched along delta in nsteps = max(2, ceil(|delta| / (radius / 2))) steps; at each
 * step it is collided (probe = g1) against every candidate geom in candidate order (the reference collides
 * it against the whole space, in ODE's traversal order, and keeps the first MAX_CONTACTS = 16 contacts of a
 * step: with more than 16 the survivors depend on that order -- the cap is applied here in candidate order).
 * float arithmetic where the reference uses float (vec3 / float locals), double where it reads dReal.
 */
float clapo_sweep_capsule(const clapo_geoms *A, uint32_t self, const float delta[3], const clapo_geoms *B,
                          uint32_t n_cand, const uint32_t *cand, float normal[3], int32_t *hit)
{
    float delta_len = sqrtf(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);   /* vec3_len */
    normal[0] = 0; normal[1] = 1; normal[2] = 0;
    if (hit) *hit = -1;
    if (delta_len < 1e-6f)
        return 1.0f;
    const double body_radius = A->radius[self];                         /* body->radius is a dReal */
    float dir[3];
    {                                                                   /* vec3_norm: scale by 1 / len */
        const float k = 1.0f / delta_len;
        dir[0] = delta[0] * k; dir[1] = delta[1] * k; dir[2] = delta[2] * k;
    }
    int nsteps = (int)ceilf((float)(delta_len / (body_radius * 0.5f)));
    if (nsteps < 2) nsteps = 2;
    float best_frac = 1.0f, best_normal[3] = { 0, 1, 0 };
    int32_t best_hit = -1;
    const double *gp = A->pos + 3 * (size_t)self;

    /* the probe as a one-geom set */
    double ppos[3];
    clapo_geoms P = *A;
    P.n = 1; P.pos = ppos; P.axis = A->axis ? A->axis + 3 * (size_t)self : NULL;
    P.radius = A->radius + self; P.length = A->length ? A->length + self : NULL;
    P.kind = A->kind ? A->kind + self : NULL; P.aabb = NULL; P.material = NULL;

    for (int s = 1; s <= nsteps; s++) {
        const float t = (float)s / nsteps;
        ppos[0] = gp[0] + delta[0] * t;                                 /* dReal + float * float */
        ppos[1] = gp[1] + delta[1] * t;
        ppos[2] = gp[2] + delta[2] * t;
        int nc_step = 0;
        for (uint32_t k = 0; k < n_cand && nc_step < 16; k++) {
            const int is_body = (cand[k] >> 31) != 0;
            const uint32_t id = cand[k] & 0x7fffffffu;
            if (is_body && id == self)
                continue;
            cgeom g[2];
            int nc = collide(&P, 0, is_body ? A : B, id, g);
            if (nc < 0) nc = 0;
            for (int i = 0; i < nc && nc_step < 16; i++, nc_step++) {
                /* the obstacle is g2 (other == 2): the normal is used as ODE gives it */
                const float cnorm[3] = { (float)g[i].normal[0], (float)g[i].normal[1], (float)g[i].normal[2] };
                const float ndot = dir[0] * cnorm[0] + dir[1] * cnorm[1] + dir[2] * cnorm[2];
                if (ndot > -0.1f)
                    continue;
                const float backup = (float)(g[i].depth / -ndot);        /* dReal / float -> float */
                const float step_dist = t * delta_len;
                float safe_dist = step_dist - backup;
                if (safe_dist < 0) safe_dist = 0;
                const float frac = safe_dist / delta_len;
                if (frac < best_frac) {
                    best_frac = frac;
                    best_normal[0] = cnorm[0]; best_normal[1] = cnorm[1]; best_normal[2] = cnorm[2];
                    best_hit = is_body ? (int32_t)id : -2 - (int32_t)id;
                }
            }
        }
        if (best_frac < t)
            break;
    }
    normal[0] = best_normal[0]; normal[1] = best_normal[1]; normal[2] = best_normal[2];
    if (hit) *hit = best_hit;
    return best_frac;
}
