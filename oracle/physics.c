/*
 * oracle/physics.c -- TEST INFRASTRUCTURE ONLY (see clap_oracle.h).
 *
 * PARITY UNPINNED.  The reference's rigid-body arithmetic lives in ODE (submodule deps/ode ->
 * git@github.com:virtuoso/ode.git, commit not recorded, built dDOUBLE with libccd + OPCODE,
 * CMakeLists.txt:344-390), which is ABSENT from /root/reference, and the reference has no test
 * at this boundary.  What is restated here:
 *
 *   from the reference itself (core/physics.c):
 *     - the fixed-step schedule of phys_step()                         physics.c:773-787
 *     - world parameters: gravity (0,-9.8,0), linear damping 0.001      physics.c:1125-1129
 *     - auto-disable thresholds 0.05 / 0.05, 30 steps                   physics.c:1039-1042
 *     - body -> entity read-back phys_body_update()                     physics.c:789-812, 96-109
 *     - the two broadphase calls dSpaceCollide2(ground, bodies) and
 *       dSpaceCollide(bodies)                                           physics.c:751-753
 *   from ODE's published algorithm (ode/src/quickstep.cpp, util.cpp dxStepBody /
 *   dInternalHandleAutoDisabling, collision_space.cpp collideAABBs; ODE 0.16 line), for a
 *   world step with no joints or contacts:
 *     - gravity into the force accumulator, lvel += (h * invMass) * facc
 *     - pos += h * lvel; q += h * 0.5 * (0,w) (x) q; renormalise
 *     - linear damping: lvel *= (1 - scale) when |lvel|^2 > threshold^2 (default 0.01^2)
 *     - auto-disable: see physics2.c (sample window, joint-holding bodies only)
 *     - AABB overlap: two boxes collide unless separated on an axis (touching counts)
 *   Capsule bodies, anisotropic inertia, general AABBs, capsule contacts and the capsule sweep: physics2.c.
 *
 * Candidate pairs are reported as the canonical ascending set, not in ODE's hash-space
 * callback order (implementation-defined).
 */
#include <stdlib.h>
#include "clap_oracle.h"
#include "lm.h"

/* physics.c:773-787 */
int clapo_phys_step_schedule(double *time_acc, double dt)
{
    const double fixed_dt = 1.0 / 120.0;
    int steps, max_steps;

    *time_acc += dt;
    for (steps = 0, max_steps = 5; *time_acc >= fixed_dt && steps < max_steps; *time_acc -= fixed_dt, steps++)
        ;
    if (steps == max_steps)
        *time_acc = 0.0;
    return steps;
}

void clapo_world_defaults(clapo_world *w)
{
    w->gravity[0] = 0; w->gravity[1] = -9.8; w->gravity[2] = 0;      /* physics.c:1125 */
    w->linear_damping = 0.001;                                         /* physics.c:1129 */
    w->linear_damping_threshold_sq = 0.01 * 0.01;                      /* ODE default threshold 0.01 */
    w->adis_linear_threshold_sq = 0.05 * 0.05;                         /* physics.c:1040 */
    w->adis_angular_threshold_sq = 0.05 * 0.05;                        /* physics.c:1041 */
    w->adis_steps = 30;                                                /* physics.c:1042 */
    w->adis_time = 0.0;
}

/* the body stage of dWorldQuickStep: clapo_bodies_step2 (physics2.c) */

/* phys_body_update (physics.c:789-812) + phys_body_rotation (96-109): body -> entity TRS */
void clapo_phys_body_update(uint32_t n, const double *pos, const double *quat, const double *lvel,
                            const double *yoffset, const int32_t *body_entity,
                            float *pos_scale, float *rot, uint32_t *entity_flags, uint8_t *moving)
{
    for (uint32_t i = 0; i < n; i++) {
        const int32_t e = body_entity[i];
        const double *p = pos + 3 * (size_t)i, *q = quat + 4 * (size_t)i, *v = lvel + 3 * (size_t)i;
        if (e >= 0) {
            pos_scale[4 * (size_t)e + 0] = p[0];
            pos_scale[4 * (size_t)e + 1] = p[1] - yoffset[i];
            pos_scale[4 * (size_t)e + 2] = p[2];
            rot[4 * (size_t)e + 0] = q[1];                          /* ODE (w,x,y,z) -> (x,y,z,w) */
            rot[4 * (size_t)e + 1] = q[2];
            rot[4 * (size_t)e + 2] = q[3];
            rot[4 * (size_t)e + 3] = q[0];
            entity_flags[e] |= CLAPO_E_DIRTY;                          /* transform_set_pos/_quat */
        }
        if (moving)
            moving[i] = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) > 1e-3 ? 1 : 0;
    }
}

/* sphere AABB: centre -/+ radius */
static void sphere_aabb(const double *p, double r, double *bb)
{
    bb[0] = p[0] - r; bb[1] = p[0] + r;
    bb[2] = p[1] - r; bb[3] = p[1] + r;
    bb[4] = p[2] - r; bb[5] = p[2] + r;
}

/* collideAABBs: disjoint iff separated on some axis (touching boxes do collide) */
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

/*
 * dSpaceCollide(bodies): every unordered pair of sphere geoms with overlapping AABBs, as the
 * ascending list of (i, j), i < j.  Sweep-and-prune on x (deliberately not the grid the GPU uses).
 * Returns the number of pairs found; at most max_pairs are written.
 */
uint64_t clapo_broadphase_pairs(uint32_t n, const double *pos, const double *radius,
                                uint32_t *pairs, uint64_t max_pairs)
{
    struct sweep_ent *s = malloc(sizeof(*s) * (n ? n : 1));
    double *bb = malloc(sizeof(double) * 6 * (n ? n : 1));
    uint64_t count = 0;

    for (uint32_t i = 0; i < n; i++) {
        sphere_aabb(pos + 3 * (size_t)i, radius[i], bb + 6 * (size_t)i);
        s[i].lo = bb[6 * (size_t)i];
        s[i].id = i;
    }
    qsort(s, n, sizeof(*s), sweep_cmp);
    for (uint32_t a = 0; a < n; a++) {
        const double *ba = bb + 6 * (size_t)s[a].id;
        for (uint32_t b = a + 1; b < n && !(s[b].lo > ba[1]); b++) {
            if (!aabb_overlap(ba, bb + 6 * (size_t)s[b].id))
                continue;
            if (count < max_pairs) {
                uint32_t i = s[a].id, j = s[b].id;
                pairs[2 * count] = i < j ? i : j;
                pairs[2 * count + 1] = i < j ? j : i;
            }
            count++;
        }
    }
    qsort(pairs, count < max_pairs ? count : max_pairs, 2 * sizeof(uint32_t), pair_cmp);
    free(s);
    free(bb);
    return count;
}

/*
 * dSpaceCollide2(ground_space, bodies): pairs (body b, static geom s) with overlapping AABBs,
 * ascending by (b, s).  static_aabb[s] = (minx,maxx,miny,maxy,minz,maxz) like ODE's dReal aabb[6].
 */
uint64_t clapo_broadphase_static_pairs(uint32_t n_static, const double *static_aabb,
                                       uint32_t n, const double *pos, const double *radius,
                                       uint32_t *pairs, uint64_t max_pairs)
{
    uint64_t count = 0;
    for (uint32_t b = 0; b < n; b++) {
        double bb[6];
        sphere_aabb(pos + 3 * (size_t)b, radius[b], bb);
        for (uint32_t s = 0; s < n_static; s++)
            if (aabb_overlap(bb, static_aabb + 6 * (size_t)s)) {
                if (count < max_pairs) {
                    pairs[2 * count] = b;
                    pairs[2 * count + 1] = s;
                }
                count++;
            }
    }
    return count;
}


/*
 * near_callback for sphere bodies (physics.c:399-449), SURVEY 8f rank 3 -- PARITY UNPINNED like the
 * rest of this file: dCollide lives in ODE (absent).  Geometry follows ODE's published
 * dCollideSpheres (ode/src/sphere.cpp: d = |p1 - p2|; none if d > r1 + r2; coincident centres give
 * normal (1,0,0), depth r1 + r2; else normal = (p1 - p2) / d, pos = p1 + normal * 0.5 (r2 - r1 - d),
 * depth = r1 + r2 - d), the surface parameters are the reference's own phys_contact_surface
 * (physics.c:291-330).  material[b] = (bounce, bounce_vel, mu, soft_erp, soft_cfm) per body, may be
 * NULL (= a body without phys_body parameters: all zero).  One clapo_contact per candidate pair.
 */
#define CLAPO_CONTACT_BOUNCE   0x004        /* dContactBounce, ode/contact.h */
#define CLAPO_CONTACT_SOFT_ERP 0x008        /* dContactSoftERP */
#define CLAPO_CONTACT_SOFT_CFM 0x010        /* dContactSoftCFM */

uint32_t clapo_contacts_spheres(uint32_t n_pairs, const uint32_t *pairs, const double *pos, const double *radius,
                                const double *material, clapo_contact *out)
{
    uint32_t total = 0;
    for (uint32_t k = 0; k < n_pairs; k++) {
        const uint32_t i = pairs[2 * k], j = pairs[2 * k + 1];
        const double *p1 = pos + 3 * (size_t)i, *p2 = pos + 3 * (size_t)j;
        const double r1 = radius[i], r2 = radius[j];
        clapo_contact *c = out + k;
        memset(c, 0, sizeof(*c));
        const double dx = p1[0] - p2[0], dy = p1[1] - p2[1], dz = p1[2] - p2[2];
        const double d = sqrt(dx * dx + dy * dy + dz * dz);
        if (d > r1 + r2) continue;
        if (d <= 0) {
            c->pos[0] = p1[0]; c->pos[1] = p1[1]; c->pos[2] = p1[2];
            c->normal[0] = 1; c->normal[1] = 0; c->normal[2] = 0;
            c->depth = r1 + r2;
        } else {
            const double d1 = 1.0 / d;
            c->normal[0] = dx * d1; c->normal[1] = dy * d1; c->normal[2] = dz * d1;
            const double kk = 0.5 * (r2 - r1 - d);
            c->pos[0] = p1[0] + c->normal[0] * kk;
            c->pos[1] = p1[1] + c->normal[1] * kk;
            c->pos[2] = p1[2] + c->normal[2] * kk;
            c->depth = r1 + r2 - d;
        }
        /* phys_contact_surface (physics.c:291-330) */
        double bounce = 0, bounce_vel = 0, mu = 0, soft_erp = 0.05, soft_cfm = 0.01;
        if (material) {
            const double *m1 = material + 5 * (size_t)i, *m2 = material + 5 * (size_t)j;
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
        c->nc = 1;
        total++;
    }
    return total;
}

/*
 * near_callback for (sphere body, static box) candidate pairs (physics.c:399-449 after
 * dSpaceCollide2(ground, bodies), physics.c:751): ODE's dCollideSphereBox (ode/src/sphere.cpp) for an
 * axis-aligned box given as ODE's aabb[6] = (minx,maxx,miny,maxy,minz,maxz) -- box position = its centre,
 * side = max - min, R = identity -- followed by phys_contact_surface (physics.c:291-330) with the parameters
 * of the body and of the static collider.  g1 = sphere, g2 = box, so the normal points from the box to the
 * sphere.  PARITY UNPINNED: ODE is not part of the reference tree; the algorithm is restated from ODE 0.16:
 *   p = c - boxpos;  t = clamp(p, -l, l) per axis, l = side / 2, onborder = any clamp happened;
 *   centre inside (not onborder): the face with the smallest l_i - |t_i| (first on ties): pos = c,
 *       normal = +-e_i (sign of t_i, +1 only if t_i > 0), depth = that distance + radius;
 *   else r = p - t, depth = radius - |r|, no contact if depth < 0; pos = t + boxpos, normal =
 *       dSafeNormalize3(r): scaled by its largest |component| first, then by 1/sqrt(sum of squares); (1,0,0) if r = 0.
 */
static void clapo_safe_normalize3(double a[3])
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

uint32_t clapo_contacts_sphere_box(uint32_t n_pairs, const uint32_t *pairs, const double *pos, const double *radius,
                                   const double *static_aabb, const double *material, const double *static_material,
                                   clapo_contact *out)
{
    uint32_t total = 0;
    for (uint32_t k = 0; k < n_pairs; k++) {
        const uint32_t i = pairs[2 * k], s = pairs[2 * k + 1];
        const double *c0 = pos + 3 * (size_t)i, *bb = static_aabb + 6 * (size_t)s;
        const double rad = radius[i];
        clapo_contact *c = out + k;
        memset(c, 0, sizeof(*c));
        double bp[3], l[3], p[3], t[3];
        int onborder = 0;
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
                const double face_distance = l[a] - fabs(t[a]);
                if (face_distance < min_distance) { min_distance = face_distance; mini = a; }
            }
            c->pos[0] = c0[0]; c->pos[1] = c0[1]; c->pos[2] = c0[2];
            c->normal[mini] = t[mini] > 0 ? 1.0 : -1.0;
            c->depth = min_distance + rad;
        } else {
            double r[3] = { p[0] - t[0], p[1] - t[1], p[2] - t[2] };
            const double depth = rad - sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
            if (depth < 0) continue;
            c->pos[0] = t[0] + bp[0]; c->pos[1] = t[1] + bp[1]; c->pos[2] = t[2] + bp[2];
            clapo_safe_normalize3(r);
            c->normal[0] = r[0]; c->normal[1] = r[1]; c->normal[2] = r[2];
            c->depth = depth;
        }
        double bounce = 0, bounce_vel = 0, mu = 0, soft_erp = 0.05, soft_cfm = 0.01;
        if (material && static_material) {
            const double *m1 = material + 5 * (size_t)i, *m2 = static_material + 5 * (size_t)s;
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
        c->nc = 1;
        total++;
    }
    return total;
}


/*
 * default_update -> phys_body_rotate_xform (model.c:1680-1687, physics.c:136-145): entity rotation
 * (x,y,z,w floats) -> body quaternion (w,x,y,z doubles), normalised as ODE's dBodySetQuaternion
 * does (dNormalize4: scale by 1 / sqrt(sum of squares)) -- PARITY UNPINNED (ODE).  dirty[e] =
 * transform_is_updated; entities with a parent take parent_transform_apply's branch and do not push.
 */
void clapo_bodies_rotate_from_entities(uint32_t n_links, const uint32_t *link_body, const uint32_t *link_entity,
                                       const float *rot, const int32_t *parent, const uint8_t *dirty, double *quat)
{
    for (uint32_t k = 0; k < n_links; k++) {
        const uint32_t b = link_body[k], e = link_entity[k];
        if (parent[e] >= 0 || !dirty[e]) continue;
        const float *r = rot + 4 * (size_t)e;
        double q[4] = { r[3], r[0], r[1], r[2] };
        const double l = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int a = 0; a < 4; a++) quat[4 * (size_t)b + a] = q[a] * l;
    }
}
