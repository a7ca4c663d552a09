// physics.hip -- the reference-side glue of phys_step() for gfx950: the fixed-step schedule (physics.c:773-787),
// world defaults, phys_body_update() (physics.c:789-812), default_update's rotation push
// (physics.c:136-145) and the round-1 sphere / sphere-box contact records.  The integrate, the AABBs, the
// broadphase, the general narrowphase and the capsule sweep are in physics2.hip.
// ODE is an absent submodule of the reference: PARITY UNPINNED (oracle/physics.c).
#include <string.h>
#include "common.h"

namespace clapgpu {

constexpr int PHYS_BLOCK = 256;

// phys_body_update (physics.c:789-812): scatter body pose into the entity SoA, mark it dirty
__global__ __launch_bounds__(PHYS_BLOCK)
void k_phys_body_update(uint32_t n, const double *pos, const double *quat, const double *lvel,
                        const double *yoffset, const int32_t *body_entity, uint32_t n_entities,
                        float *pos_scale, float *rot, uint32_t *entity_flags, uint8_t *moving)
{
    const uint32_t i = blockIdx.x * PHYS_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    const int32_t e = body_entity[i];
    const double *p = pos + 3 * (size_t)i, *q = quat + 4 * (size_t)i, *v = lvel + 3 * (size_t)i;
    if (e >= 0 && (uint32_t)e < n_entities) {
        pos_scale[4 * (size_t)e + 0] = (float)p[0];
        pos_scale[4 * (size_t)e + 1] = (float)(p[1] - yoffset[i]);
        pos_scale[4 * (size_t)e + 2] = (float)p[2];
        reinterpret_cast<float4 *>(rot)[e] = make_float4((float)q[1], (float)q[2], (float)q[3], (float)q[0]);
        atomicOr(&entity_flags[e], CLAPGPU_E_DIRTY);
    }
    if (moving)
        moving[i] = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) > 1e-3 ? 1 : 0;
}

// phys_body_rotate_xform (physics.c:136-145) for the linked entities that default_update rebuilt
__global__ __launch_bounds__(PHYS_BLOCK)
void k_bodies_rotate_from_entities(uint32_t n_links, const uint32_t *link_body, const uint32_t *link_entity,
                                   uint32_t n_bodies, uint32_t n_entities, uint32_t mode, const float4 *rot,
                                   const int32_t *parent, const uint32_t *flags, double *quat)
{
    const uint32_t k = blockIdx.x * PHYS_BLOCK + threadIdx.x;
    if (k >= n_links) return;
    const uint32_t b = link_body[k], e = link_entity[k];
    if (b >= n_bodies || e >= n_entities || parent[e] >= 0) return;
    if (!(mode & CLAPGPU_UPDATE_ALL_DIRTY) && !(flags[e] & CLAPGPU_E_DIRTY)) return;
    const float4 r = rot[e];
    double q[4] = { (double)r.w, (double)r.x, (double)r.y, (double)r.z };
    const double l = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);   // dNormalize4
    for (int a = 0; a < 4; a++) quat[4 * (size_t)b + a] = q[a] * l;
}

// phys_contact_surface (physics.c:291-330) for the two colliders' parameter rows (NULL: defaults)
__device__ __forceinline__ void contact_surface(clapgpu_contact &c, const double *m1, const double *m2)
{
    double bounce = 0, bounce_vel = 0, mu = 0, soft_erp = 0.05, soft_cfm = 0.01;   // physics.c:293-294
    if (m1 && m2) {
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
    c.mode = CLAPGPU_CONTACT_SOFT_CFM | CLAPGPU_CONTACT_SOFT_ERP | (bounce > 0 ? CLAPGPU_CONTACT_BOUNCE : 0);
    c.mu = mu; c.bounce = bounce; c.bounce_vel = bounce_vel; c.soft_erp = soft_erp; c.soft_cfm = soft_cfm;
    c.nc = 1;
}

// ODE's dCollideSpheres
__device__ __forceinline__ bool contact_sphere_sphere(clapgpu_contact &c, const double *p1, double r1, const double *p2,
                                                      double r2)
{
    const double dx = p1[0] - p2[0], dy = p1[1] - p2[1], dz = p1[2] - p2[2];
    const double d = sqrt(dx * dx + dy * dy + dz * dz);
    if (d > r1 + r2) return false;
    if (d <= 0) {
        c.pos[0] = p1[0]; c.pos[1] = p1[1]; c.pos[2] = p1[2];
        c.normal[0] = 1; c.normal[1] = 0; c.normal[2] = 0;
        c.depth = r1 + r2;
    } else {
        const double d1 = 1.0 / d;
        c.normal[0] = dx * d1; c.normal[1] = dy * d1; c.normal[2] = dz * d1;
        const double kk = 0.5 * (r2 - r1 - d);
        c.pos[0] = p1[0] + c.normal[0] * kk;
        c.pos[1] = p1[1] + c.normal[1] * kk;
        c.pos[2] = p1[2] + c.normal[2] * kk;
        c.depth = r1 + r2 - d;
    }
    return true;
}

// ODE's dCollideSphereBox for an axis-aligned box given as aabb[6] = (minx,maxx,miny,maxy,minz,maxz):
// box position = centre, side = max - min, R = identity (oracle/physics.c has the restated algorithm)
__device__ __forceinline__ bool contact_sphere_box(clapgpu_contact &c, const double *c0, double rad, const double *bb)
{
    double bp[3], l[3], p[3], t[3];
    bool onborder = false;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        bp[a] = (bb[2 * a] + bb[2 * a + 1]) * 0.5;
        l[a] = (bb[2 * a + 1] - bb[2 * a]) * 0.5;
        p[a] = c0[a] - bp[a];
        t[a] = p[a];
        if (t[a] < -l[a]) { t[a] = -l[a]; onborder = true; }
        if (t[a] > l[a]) { t[a] = l[a]; onborder = true; }
    }
    if (!onborder) {                                                // centre inside: push out through the closest face
        double min_distance = l[0] - fabs(t[0]);
        int mini = 0;
#pragma unroll
        for (int a = 1; a < 3; a++) {
            const double face_distance = l[a] - fabs(t[a]);
            if (face_distance < min_distance) { min_distance = face_distance; mini = a; }
        }
        c.pos[0] = c0[0]; c.pos[1] = c0[1]; c.pos[2] = c0[2];
        const double sgn = t[mini] > 0 ? 1.0 : -1.0;
        c.normal[0] = mini == 0 ? sgn : 0.0; c.normal[1] = mini == 1 ? sgn : 0.0; c.normal[2] = mini == 2 ? sgn : 0.0;
        c.depth = min_distance + rad;
        return true;
    }
    double r[3] = { p[0] - t[0], p[1] - t[1], p[2] - t[2] };
    const double depth = rad - sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (depth < 0) return false;
    c.pos[0] = t[0] + bp[0]; c.pos[1] = t[1] + bp[1]; c.pos[2] = t[2] + bp[2];
    {                                                               // dSafeNormalize3
        const double aa[3] = { fabs(r[0]), fabs(r[1]), fabs(r[2]) };
        int idx = 0;
        bool zero = false;
        if (aa[1] > aa[0]) idx = aa[2] > aa[1] ? 2 : 1;
        else if (aa[2] > aa[0]) idx = 2;
        else zero = aa[0] <= 0;
        if (zero) { r[0] = 1; r[1] = 0; r[2] = 0; }
        else {
            const double m = idx == 0 ? aa[0] : idx == 1 ? aa[1] : aa[2];
            r[0] /= m; r[1] /= m; r[2] /= m;
            const double k = 1.0 / sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
            r[0] *= k; r[1] *= k; r[2] *= k;
        }
    }
    c.normal[0] = r[0]; c.normal[1] = r[1]; c.normal[2] = r[2];
    c.depth = depth;
    return true;
}

// near_callback's dCollide + phys_contact_surface (see include/clapgpu.h): one lane per candidate pair,
// IEEE fp64 (sqrt, divide), no contraction.  BOX = false: (body, body) sphere pairs; BOX = true: (body,
// static box) pairs, `other` = static_aabb, `other_material` = the static colliders' parameter rows.
template <bool BOX>
__global__ __launch_bounds__(PHYS_BLOCK)
void k_contacts(const double *pos, const double *radius, uint32_t n_bodies, const double *other, uint32_t n_other,
                const uint2 *pairs, const uint32_t *pair_total, uint32_t capacity, const double *material,
                const double *other_material, clapgpu_contact *out, uint32_t *contact_total)
{
    __shared__ __attribute__((aligned(16))) double recs[PHYS_BLOCK / WAVE][WAVE * 13];
    static_assert(sizeof(clapgpu_contact) == 13 * sizeof(double), "contact record layout");
    const uint32_t n_pairs = *pair_total < capacity ? *pair_total : capacity;
    const int lane = lane_id();
    uint32_t found = 0;
    // the pair count is only known on the device: a fixed grid strides over the pairs (a grid sized for
    // the capacity spends 50 us launching empty workgroups)
    for (uint32_t k = blockIdx.x * PHYS_BLOCK + threadIdx.x; k - lane < n_pairs; k += gridDim.x * PHYS_BLOCK) {
    double *rec = recs[threadIdx.x / WAVE];
    bool touch = false;
    if (k < n_pairs) {
        const uint2 pr = pairs[k];
        clapgpu_contact c;
        memset(&c, 0, sizeof(c));
        if (pr.x < n_bodies && pr.y < (BOX ? n_other : n_bodies)) {
            if (BOX) {
                touch = contact_sphere_box(c, pos + 3 * (size_t)pr.x, radius[pr.x], other + 6 * (size_t)pr.y);
                if (touch)
                    contact_surface(c, material && other_material ? material + 5 * (size_t)pr.x : nullptr,
                                    material && other_material ? other_material + 5 * (size_t)pr.y : nullptr);
            } else {
                touch = contact_sphere_sphere(c, pos + 3 * (size_t)pr.x, radius[pr.x], pos + 3 * (size_t)pr.y, radius[pr.y]);
                if (touch)
                    contact_surface(c, material ? material + 5 * (size_t)pr.x : nullptr,
                                    material ? material + 5 * (size_t)pr.y : nullptr);
            }
        }
        // the 104-byte records of a wave are contiguous in memory: stage them in LDS and write the run as
        // 16-byte pieces (a record per lane straight to memory is 13 scattered 8-byte stores per lane)
        memcpy(rec + (size_t)lane * 13, &c, sizeof(c));
    }
    wave_lds_fence();
    {
        const uint32_t wave_first = k - lane;                       // first pair of this wave
        const uint32_t n_here = wave_first < n_pairs ? (n_pairs - wave_first < WAVE ? n_pairs - wave_first : WAVE) : 0;
        const uint32_t n16 = n_here * (uint32_t)(sizeof(clapgpu_contact) / 8) / 2;      // 16-byte pieces (104 * 64 % 16 == 0 only for even counts)
        const double2 *src = reinterpret_cast<const double2 *>(rec);
        double2 *dst = reinterpret_cast<double2 *>(out + wave_first);
        if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
            for (uint32_t q = lane; q < n16; q += WAVE) dst[q] = src[q];
            if ((n_here & 1) && lane == 0)                          // odd count: the last 8 bytes
                reinterpret_cast<double *>(out + wave_first)[n_here * 13 - 1] = rec[n_here * 13 - 1];
        } else {
            for (uint32_t q = lane; q < n_here * 13; q += WAVE)
                reinterpret_cast<double *>(out + wave_first)[q] = rec[q];
        }
    }
    found += (uint32_t)__popcll(__ballot(touch));
    wave_lds_fence();                                               // the staging tile is reused by the next trip
    }
    // one global atomic per workgroup: same-address atomics serialise at ~12 ns each (4096 of them were
    // 50 us of this kernel)
    __shared__ uint32_t block_found;
    if (threadIdx.x == 0) block_found = 0;
    __syncthreads();
    if (lane == 0 && found) atomicAdd(&block_found, found);
    __syncthreads();
    if (contact_total && threadIdx.x == 0 && block_found)
        atomicAdd(contact_total, block_found);
}

} // namespace clapgpu

using namespace clapgpu;

// physics.c:773-787 (host)
extern "C" int clapgpu_phys_step_schedule(double *time_acc, double dt)
{
    const double fixed_dt = 1.0 / 120.0;
    int steps = 0;
    const int max_steps = 5;
    *time_acc += dt;
    for (; *time_acc >= fixed_dt && steps < max_steps; *time_acc -= fixed_dt, steps++)
        ;
    if (steps == max_steps)
        *time_acc = 0.0;
    return steps;
}

extern "C" void clapgpu_world_defaults(clapgpu_world *w)
{
    memset(w, 0, sizeof(*w));
    w->gravity[1] = -9.8;                          // physics.c:1125
    w->linear_damping = 0.001;                     // physics.c:1129
    w->linear_damping_threshold_sq = 0.01 * 0.01;  // ODE default damping threshold
    w->adis_linear_threshold_sq = 0.05 * 0.05;     // physics.c:1040
    w->adis_angular_threshold_sq = 0.05 * 0.05;    // physics.c:1041
    w->adis_steps = 30;                            // physics.c:1042
    w->adis_time = 0.0;
}

static int check_bodies(const clapgpu_bodies *b)
{
    if (!b || !b->pos || !b->quat || !b->lvel || !b->avel || !b->mass || !b->radius || !b->bflags ||
        !b->adis_steps_left || !b->adis_time_left)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    return CLAPGPU_OK;
}

extern "C" int clapgpu_phys_body_update(void *stream, const clapgpu_bodies *b, uint32_t n_entities, float *pos_scale,
                                        float *rot, uint32_t *entity_flags, uint8_t *moving)
{
    int rc = check_bodies(b);
    if (rc) return rc;
    if (!b->yoffset || !b->body_entity || !pos_scale || !rot || !entity_flags)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (b->n == 0) return CLAPGPU_OK;
    hipLaunchKernelGGL(k_phys_body_update, dim3((b->n + PHYS_BLOCK - 1) / PHYS_BLOCK), dim3(PHYS_BLOCK), 0,
                       as_stream(stream), b->n, b->pos, b->quat, b->lvel, b->yoffset, b->body_entity, n_entities,
                       pos_scale, rot, entity_flags, moving);
    CLAPGPU_LAUNCH_CHECK("k_phys_body_update");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_contacts_spheres(void *stream, const clapgpu_bodies *b, const uint32_t *pairs,
                                        const uint32_t *pair_total, uint32_t capacity, const double *material,
                                        clapgpu_contact *contacts, uint32_t *contact_total)
{
    int rc = check_bodies(b);
    if (rc) return rc;
    if (!pair_total || (capacity && (!pairs || !contacts)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    hipStream_t s = as_stream(stream);
    if (contact_total)
        CLAPGPU_HIP(hipMemsetAsync(contact_total, 0, sizeof(uint32_t), s));
    if (capacity == 0 || b->n == 0)
        return CLAPGPU_OK;
    // the pair count lives on the device: launch for the capacity, lanes past the count retire at once
    const uint32_t blocks = (capacity + PHYS_BLOCK - 1) / PHYS_BLOCK;
    hipLaunchKernelGGL(k_contacts<false>, dim3(blocks < 512 ? blocks : 512), dim3(PHYS_BLOCK), 0, s,
                       b->pos, b->radius, b->n, nullptr, 0u, reinterpret_cast<const uint2 *>(pairs), pair_total, capacity,
                       material, nullptr, contacts, contact_total);
    CLAPGPU_LAUNCH_CHECK("k_contacts<spheres>");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_contacts_sphere_box(void *stream, const clapgpu_bodies *b, uint32_t n_static,
                                           const double *static_aabb, const uint32_t *pairs, const uint32_t *pair_total,
                                           uint32_t capacity, const double *material, const double *static_material,
                                           clapgpu_contact *contacts, uint32_t *contact_total)
{
    int rc = check_bodies(b);
    if (rc) return rc;
    if (!pair_total || (n_static && !static_aabb) || (capacity && (!pairs || !contacts)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    hipStream_t s = as_stream(stream);
    if (contact_total)
        CLAPGPU_HIP(hipMemsetAsync(contact_total, 0, sizeof(uint32_t), s));
    if (capacity == 0 || b->n == 0 || n_static == 0)
        return CLAPGPU_OK;
    const uint32_t blocks = (capacity + PHYS_BLOCK - 1) / PHYS_BLOCK;
    hipLaunchKernelGGL(k_contacts<true>, dim3(blocks < 512 ? blocks : 512), dim3(PHYS_BLOCK), 0, s,
                       b->pos, b->radius, b->n, static_aabb, n_static, reinterpret_cast<const uint2 *>(pairs), pair_total,
                       capacity, material, static_material, contacts, contact_total);
    CLAPGPU_LAUNCH_CHECK("k_contacts<sphere_box>");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_bodies_rotate_from_entities(void *stream, const clapgpu_bodies *b, const clapgpu_entities *e,
                                                   uint32_t mode, uint32_t n_links, const uint32_t *link_body,
                                                   const uint32_t *link_entity)
{
    int rc = check_bodies(b);
    if (rc) return rc;
    if (!e || !e->rot || !e->parent || !e->flags || (n_links && (!link_body || !link_entity)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (n_links == 0 || b->n == 0)
        return CLAPGPU_OK;
    hipLaunchKernelGGL(k_bodies_rotate_from_entities, dim3((n_links + PHYS_BLOCK - 1) / PHYS_BLOCK), dim3(PHYS_BLOCK), 0,
                       as_stream(stream), n_links, link_body, link_entity, b->n, e->n, mode,
                       reinterpret_cast<const float4 *>(e->rot), e->parent, e->flags, b->quat);
    CLAPGPU_LAUNCH_CHECK("k_bodies_rotate_from_entities");
    return CLAPGPU_OK;
}
