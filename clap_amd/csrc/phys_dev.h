// phys_dev.h -- fp64 rigid-body geometry shared by the physics kernels and their host-side set-up code:
// ODE's rotation / mass / AABB / narrowphase arithmetic for spheres, capsules and axis-aligned boxes, in
// ODE's operation order (the reference builds ODE with dDOUBLE, physics.h:5-9; -ffp-contract=off).
// ODE (deps/ode) is an absent submodule of the reference: what follows is restated from its published
// sources (0.16 line: rotation.cpp, mass.cpp, capsule.cpp, sphere.cpp, collision_util.cpp) and from the
// call sites in core/physics.c -- PARITY UNPINNED, see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define PHD __host__ __device__ __forceinline__

namespace phd {

struct CGeom { double pos[3], normal[3], depth; };       // dContactGeom's numbers

// dQtoR: dMatrix3 = 3 rows of 4
PHD void q_to_R(const double (&q)[4], double (&R)[12])
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

// column 2 of R_body * R_offset (dMultiply0_333): the capsule's axis in world space
PHD void capsule_axis(const double (&R)[12], const double (&Roff)[12], double (&axis)[3])
{
#pragma unroll
    for (int i = 0; i < 3; i++)
        axis[i] = R[4 * i] * Roff[2] + R[4 * i + 1] * Roff[6] + R[4 * i + 2] * Roff[10];
}

// dxSphere::computeAABB / dxCapsule::computeAABB -> (minx,maxx,miny,maxy,minz,maxz)
PHD void geom_aabb(const double (&p)[3], double radius, double lz, const double (&axis)[3], double (&bb)[6])
{
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double range = lz == 0.0 ? radius : fabs(axis[a] * lz) * 0.5 + radius;
        bb[2 * a] = p[a] - range;
        bb[2 * a + 1] = p[a] + range;
    }
}

PHD double det3(const double (&m)[12])
{
    return m[0] * (m[5] * m[10] - m[9] * m[6]) - m[1] * (m[4] * m[10] - m[8] * m[6]) + m[2] * (m[4] * m[9] - m[8] * m[5]);
}

// dInvertMatrix3
PHD bool invert3(double (&dst)[12], const double (&ma)[12])
{
    const double det = det3(ma);
    if (det == 0) return false;
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
    return true;
}

// W = R diag(d) R^T the way quickstep builds the world-frame tensors: tmp = D R^T, W = R tmp
PHD void world_tensor(const double (&R)[12], const double (&d)[3], double (&W)[12])
{
    double tmp[12];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            tmp[4 * i + j] = d[i] * R[4 * j + i];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++)
            W[4 * i + j] = R[4 * i] * tmp[j] + R[4 * i + 1] * tmp[4 + j] + R[4 * i + 2] * tmp[8 + j];
        W[4 * i + 3] = 0;
    }
}

PHD void mul331(double (&out)[3], const double (&M)[12], const double (&v)[3])
{
#pragma unroll
    for (int i = 0; i < 3; i++)
        out[i] = M[4 * i] * v[0] + M[4 * i + 1] * v[1] + M[4 * i + 2] * v[2];
}

PHD double dot3(const double (&a)[3], const double (&b)[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// dCollideSpheres
PHD int collide_spheres(const double (&p1)[3], double r1, const double (&p2)[3], double r2, CGeom &c)
{
    const double dx = p1[0] - p2[0], dy = p1[1] - p2[1], dz = p1[2] - p2[2];
    const double d = sqrt(dx * dx + dy * dy + dz * dz);
    if (d > r1 + r2) return 0;
    if (d <= 0) {
        c.pos[0] = p1[0]; c.pos[1] = p1[1]; c.pos[2] = p1[2];
        c.normal[0] = 1; c.normal[1] = 0; c.normal[2] = 0;
        c.depth = r1 + r2;
    } else {
        const double d1 = 1.0 / d;
        c.normal[0] = dx * d1; c.normal[1] = dy * d1; c.normal[2] = dz * d1;
        const double k = 0.5 * (r2 - r1 - d);
        c.pos[0] = p1[0] + c.normal[0] * k;
        c.pos[1] = p1[1] + c.normal[1] * k;
        c.pos[2] = p1[2] + c.normal[2] * k;
        c.depth = r1 + r2 - d;
    }
    return 1;
}

// dCollideCapsuleSphere
PHD int collide_capsule_sphere(const double (&cp)[3], const double (&ax)[3], double cr, double lz,
                               const double (&sp)[3], double sr, CGeom &c)
{
    double alpha = ax[0] * (sp[0] - cp[0]) + ax[1] * (sp[1] - cp[1]) + ax[2] * (sp[2] - cp[2]);
    const double lz2 = lz * 0.5;
    if (alpha > lz2) alpha = lz2;
    if (alpha < -lz2) alpha = -lz2;
    const double p[3] = { cp[0] + alpha * ax[0], cp[1] + alpha * ax[1], cp[2] + alpha * ax[2] };
    return collide_spheres(p, cr, sp, sr, c);
}

// dClosestLineSegmentPoints
PHD void closest_segment_points(const double (&a1)[3], const double (&a2)[3], const double (&b1)[3], const double (&b2)[3],
                                double (&cp1)[3], double (&cp2)[3])
{
    double a1a2[3], b1b2[3], a1b1[3], a1b2[3], a2b1[3], a2b2[3], n[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { a1a2[i] = a2[i] - a1[i]; b1b2[i] = b2[i] - b1[i]; a1b1[i] = b1[i] - a1[i]; }
    const double da1 = dot3(a1a2, a1b1), db1 = dot3(b1b2, a1b1);
    if (da1 <= 0 && db1 >= 0) {
        for (int i = 0; i < 3; i++) { cp1[i] = a1[i]; cp2[i] = b1[i]; }
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) a1b2[i] = b2[i] - a1[i];
    const double da2 = dot3(a1a2, a1b2), db2 = dot3(b1b2, a1b2);
    if (da2 <= 0 && db2 <= 0) {
        for (int i = 0; i < 3; i++) { cp1[i] = a1[i]; cp2[i] = b2[i]; }
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) a2b1[i] = b1[i] - a2[i];
    const double da3 = dot3(a1a2, a2b1), db3 = dot3(b1b2, a2b1);
    if (da3 >= 0 && db3 >= 0) {
        for (int i = 0; i < 3; i++) { cp1[i] = a2[i]; cp2[i] = b1[i]; }
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) a2b2[i] = b2[i] - a2[i];
    const double da4 = dot3(a1a2, a2b2), db4 = dot3(b1b2, a2b2);
    if (da4 >= 0 && db4 <= 0) {
        for (int i = 0; i < 3; i++) { cp1[i] = a2[i]; cp2[i] = b2[i]; }
        return;
    }
    const double la = dot3(a1a2, a1a2);
    if (da1 >= 0 && da3 <= 0) {
        const double k = da1 / la;
        for (int i = 0; i < 3; i++) n[i] = a1b1[i] - k * a1a2[i];
        if (dot3(b1b2, n) >= 0) {
            for (int i = 0; i < 3; i++) { cp1[i] = a1[i] + k * a1a2[i]; cp2[i] = b1[i]; }
            return;
        }
    }
    if (da2 >= 0 && da4 <= 0) {
        const double k = da2 / la;
        for (int i = 0; i < 3; i++) n[i] = a1b2[i] - k * a1a2[i];
        if (dot3(b1b2, n) <= 0) {
            for (int i = 0; i < 3; i++) { cp1[i] = a1[i] + k * a1a2[i]; cp2[i] = b2[i]; }
            return;
        }
    }
    const double lb = dot3(b1b2, b1b2);
    if (db1 <= 0 && db2 >= 0) {
        const double k = -db1 / lb;
        for (int i = 0; i < 3; i++) n[i] = -a1b1[i] - k * b1b2[i];
        if (dot3(a1a2, n) >= 0) {
            for (int i = 0; i < 3; i++) { cp1[i] = a1[i]; cp2[i] = b1[i] + k * b1b2[i]; }
            return;
        }
    }
    if (db3 <= 0 && db4 >= 0) {
        const double k = -db3 / lb;
        for (int i = 0; i < 3; i++) n[i] = -a2b1[i] - k * b1b2[i];
        if (dot3(a1a2, n) <= 0) {
            for (int i = 0; i < 3; i++) { cp1[i] = a2[i]; cp2[i] = b1[i] + k * b1b2[i]; }
            return;
        }
    }
    const double k = dot3(a1a2, b1b2);
    double det = la * lb - k * k;
    if (det <= 0) {
        for (int i = 0; i < 3; i++) { cp1[i] = a1[i]; cp2[i] = b1[i]; }
        return;
    }
    det = 1.0 / det;
    const double alpha = (lb * da1 - k * db1) * det;
    const double beta = (k * da1 - la * db1) * det;
    for (int i = 0; i < 3; i++) { cp1[i] = a1[i] + alpha * a1a2[i]; cp2[i] = b1[i] + beta * b1b2[i]; }
}

// dCollideCapsuleCapsule: up to two contacts (c0, c1)
PHD int collide_capsule_capsule(const double (&pos1)[3], const double (&ax1)[3], double r1, double l1,
                                const double (&pos2)[3], const double (&ax2in)[3], double r2, double l2, CGeom &c0, CGeom &c1)
{
    const double tolerance = 1e-5;
    const double lz1 = l1 * 0.5, lz2 = l2 * 0.5;
    double axis2[3] = { ax2in[0], ax2in[1], ax2in[2] };
    double sphere1[3], sphere2[3];
    const double a1a2 = dot3(ax1, axis2);
    const double det = 1.0 - a1a2 * a1a2;
    if (det < tolerance) {
        if (a1a2 < 0) { axis2[0] = -axis2[0]; axis2[1] = -axis2[1]; axis2[2] = -axis2[2]; }
        const double q[3] = { pos1[0] - pos2[0], pos1[1] - pos2[1], pos1[2] - pos2[2] };
        const double k = dot3(ax1, q);
        const double a1lo = -lz1, a1hi = lz1, a2lo = -lz2 - k, a2hi = lz2 - k;
        const double lo = a1lo > a2lo ? a1lo : a2lo;
        const double hi = a1hi < a2hi ? a1hi : a2hi;
        if (lo <= hi) {
            if (lo < hi) {
                for (int i = 0; i < 3; i++) { sphere1[i] = pos1[i] + lo * ax1[i]; sphere2[i] = pos2[i] + (lo + k) * axis2[i]; }
                if (collide_spheres(sphere1, r1, sphere2, r2, c0)) {
                    for (int i = 0; i < 3; i++) { sphere1[i] = pos1[i] + hi * ax1[i]; sphere2[i] = pos2[i] + (hi + k) * axis2[i]; }
                    if (collide_spheres(sphere1, r1, sphere2, r2, c1))
                        return 2;
                }
            }
            const double alpha1 = (lo + hi) * 0.5, alpha2 = alpha1 + k;
            for (int i = 0; i < 3; i++) { sphere1[i] = pos1[i] + alpha1 * ax1[i]; sphere2[i] = pos2[i] + alpha2 * axis2[i]; }
            return collide_spheres(sphere1, r1, sphere2, r2, c0);
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
    return collide_spheres(sphere1, r1, sphere2, r2, c0);
}

// dClosestLineBoxPoints for a box with R = identity
PHD void closest_line_box_points(const double (&p1)[3], const double (&p2)[3], const double (&c)[3], const double (&side)[3],
                                 double (&lret)[3], double (&bret)[3])
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
    if (!(dd2dt >= 0)) {
        bool done = false;
        do {
            double next_t = 1;
            for (int i = 0; i < 3; i++)
                if (tanchor[i] > t && tanchor[i] < 1 && tanchor[i] < next_t) next_t = tanchor[i];
            double next_dd2dt = 0;
            for (int i = 0; i < 3; i++) next_dd2dt += (region[i] ? v2[i] : 0) * (next_t - tanchor[i]);
            if (next_dd2dt >= 0) {
                const double m = (next_dd2dt - dd2dt) / (next_t - t);
                t -= dd2dt / m;
                done = true;
                break;
            }
            for (int i = 0; i < 3; i++)
                if (tanchor[i] == next_t) { tanchor[i] = (h[i] - s[i]) / v[i]; region[i]++; }
            t = next_t;
            dd2dt = next_dd2dt;
        } while (t < 1);
        if (!done) t = 1;
    }
    for (int i = 0; i < 3; i++) lret[i] = p1[i] + t * tmp[i];
    for (int i = 0; i < 3; i++) {
        double x = sign[i] * (s[i] + t * v[i]);
        if (x < -h[i]) x = -h[i];
        else if (x > h[i]) x = h[i];
        bret[i] = x + c[i];
    }
}

// dCollideCapsuleBox against an axis-aligned box (aabb[6]); -1 = the dBoxBox branch (capsule axis touches the box)
PHD int collide_capsule_box(const double (&cp)[3], const double (&ax)[3], double radius, double lz, const double (&bb)[6], CGeom &c)
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

// dSafeNormalize3
PHD void safe_normalize3(double (&a)[3])
{
    const double aa[3] = { fabs(a[0]), fabs(a[1]), fabs(a[2]) };
    int idx;
    if (aa[1] > aa[0]) idx = aa[2] > aa[1] ? 2 : 1;
    else if (aa[2] > aa[0]) idx = 2;
    else {
        if (aa[0] <= 0) { a[0] = 1; a[1] = 0; a[2] = 0; return; }
        idx = 0;
    }
    const double s = idx == 0 ? aa[0] : idx == 1 ? aa[1] : aa[2];
    a[0] /= s; a[1] /= s; a[2] /= s;
    const double l = 1.0 / sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    a[0] *= l; a[1] *= l; a[2] *= l;
}

// dCollideSphereBox against an axis-aligned box
PHD int collide_sphere_box(const double (&c0)[3], double rad, const double (&bb)[6], CGeom &c)
{
    double bp[3], l[3], p[3], t[3];
    bool onborder = false;
    for (int a = 0; a < 3; a++) { c.pos[a] = 0; c.normal[a] = 0; }
    c.depth = 0;
    for (int a = 0; a < 3; a++) {
        bp[a] = (bb[2 * a] + bb[2 * a + 1]) * 0.5;
        l[a] = (bb[2 * a + 1] - bb[2 * a]) * 0.5;
        p[a] = c0[a] - bp[a];
        t[a] = p[a];
        if (t[a] < -l[a]) { t[a] = -l[a]; onborder = true; }
        if (t[a] > l[a]) { t[a] = l[a]; onborder = true; }
    }
    if (!onborder) {
        double min_distance = l[0] - fabs(t[0]);
        int mini = 0;
        for (int a = 1; a < 3; a++) {
            const double fd = l[a] - fabs(t[a]);
            if (fd < min_distance) { min_distance = fd; mini = a; }
        }
        c.pos[0] = c0[0]; c.pos[1] = c0[1]; c.pos[2] = c0[2];
        const double sgn = (mini == 0 ? t[0] : mini == 1 ? t[1] : t[2]) > 0 ? 1.0 : -1.0;
        c.normal[0] = mini == 0 ? sgn : 0.0;
        c.normal[1] = mini == 1 ? sgn : 0.0;
        c.normal[2] = mini == 2 ? sgn : 0.0;
        c.depth = min_distance + rad;
        return 1;
    }
    double r[3] = { p[0] - t[0], p[1] - t[1], p[2] - t[2] };
    const double depth = rad - sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (depth < 0) return 0;
    c.pos[0] = t[0] + bp[0]; c.pos[1] = t[1] + bp[1]; c.pos[2] = t[2] + bp[2];
    safe_normalize3(r);
    c.normal[0] = r[0]; c.normal[1] = r[1]; c.normal[2] = r[2];
    c.depth = depth;
    return 1;
}

// One geom as the narrowphase sees it
struct Geom {
    int    kind;                         // CLAPGPU_GEOM_*
    double pos[3], axis[3], radius, length, aabb[6];
};

// dCollide(o1 = a, o2 = b): the class pair's collider; swapped and reversed (normals negated) when only the
// swapped one exists (collision_kernel.cpp).  Returns nc; -1 for the dBoxBox branch; 0 without a collider here.
PHD int collide(const Geom &a, const Geom &b, CGeom &c0, CGeom &c1)
{
    int nc = 0;
    bool reverse = false;
    if (a.kind == 0 && b.kind == 0) nc = collide_spheres(a.pos, a.radius, b.pos, b.radius, c0);
    else if (a.kind == 1 && b.kind == 0) nc = collide_capsule_sphere(a.pos, a.axis, a.radius, a.length, b.pos, b.radius, c0);
    else if (a.kind == 0 && b.kind == 1) { nc = collide_capsule_sphere(b.pos, b.axis, b.radius, b.length, a.pos, a.radius, c0); reverse = true; }
    else if (a.kind == 1 && b.kind == 1) nc = collide_capsule_capsule(a.pos, a.axis, a.radius, a.length, b.pos, b.axis, b.radius, b.length, c0, c1);
    else if (a.kind == 0 && b.kind == 2) nc = collide_sphere_box(a.pos, a.radius, b.aabb, c0);
    else if (a.kind == 2 && b.kind == 0) { nc = collide_sphere_box(b.pos, b.radius, a.aabb, c0); reverse = true; }
    else if (a.kind == 1 && b.kind == 2) nc = collide_capsule_box(a.pos, a.axis, a.radius, a.length, b.aabb, c0);
    else if (a.kind == 2 && b.kind == 1) { nc = collide_capsule_box(b.pos, b.axis, b.radius, b.length, a.aabb, c0); reverse = true; }
    if (reverse && nc > 0)
        for (int k = 0; k < 3; k++) { c0.normal[k] = -c0.normal[k]; c1.normal[k] = -c1.normal[k]; }
    return nc;
}

} // namespace phd
