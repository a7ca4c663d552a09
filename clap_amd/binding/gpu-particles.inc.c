/*
 * gpu-particles.inc.c -- CLAP-side binding of libclapgpu for particle systems.
 *
 * `struct particle` and `struct particle_system` are private to core/particle.c, so this file is
 * meant to be #include'd at the end of that translation unit (under the same CONFIG_GPU_SCENE switch
 * as gpu-scene.c); the drop-in checker oracle/ref/dropin.c includes both the same way.  It replaces the
 * N per-entity particles_update() hooks of a frame (particle.c:89-120) by ONE call:
 *
 *   gpu_particles_update(gp, mq, scene, scatter)
 *       every ALIVE particle-system entity of the queue whose hook is particles_update, in list order
 *       (= the order the reference consumes its single libc drand48 stream in), on the device:
 *       respawn test, random_point_sphere + particle_set_velocity with the exact drand48 sequence,
 *       advect, billboard matrix.  On return ps->pos_array (what particle_system_upload hands to the
 *       shader, particle.c:122-125), e->mx and libc's drand48 state are what the reference would have
 *       left; with `scatter` the per-particle structs (p->pos, p->velocity) too.
 *   gpu_particle_system_position(gp, ps, center)
 *       particle_system_position() (particle.c:132-157) for a mirrored system: an attached system
 *       carries its device-resident particles along.
 *
 * The device copy is authoritative for positions / velocities between frames; it is rebuilt from the
 * host lists whenever the set of systems (or a count) changes.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "clapgpu.h"

struct gpu_particles {
    uint32_t                n_sys, cap_sys, n;       /* n = padded particle slots (systems start at multiples of 64) */
    particle_system         **ps, **seen;    /* mirrored systems; this frame's walk */
    clapgpu_particle_system *sys_host;
    clapgpu_particles       d;                       /* device descriptor */
    void                    *d_sys;
    float                   *h_pos, *h_vel, *h_mx;   /* page-locked staging */
    uint32_t                cap_n;
    uint64_t                h_rng[2];
    bool                    host_stale;              /* p->pos / p->velocity older than the device copy */
};

#define GP_CK(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

static uint64_t gp_libc_state_get(void)
{
    unsigned short zero[3] = { 0, 0, 0 }, *old = seed48(zero), keep[3] = { old[0], old[1], old[2] };
    seed48(keep);
    return (uint64_t)keep[0] | ((uint64_t)keep[1] << 16) | ((uint64_t)keep[2] << 32);
}

static void gp_libc_state_set(uint64_t st)
{
    unsigned short s16[3] = { st & 0xffff, (st >> 16) & 0xffff, (st >> 32) & 0xffff };
    seed48(s16);
}

int gpu_particles_init(struct gpu_particles **out, int device)
{
    if (!out) return _CERR_INVALID_ARGUMENTS;
    int rc = clapgpu_init(device);
    if (rc) return rc;
    *out = calloc(1, sizeof(**out));
    return *out ? 0 : _CERR_NOMEM;
}

static void gp_free_device(struct gpu_particles *gp)
{
    void *dev[] = { gp->d_sys, (void *)gp->d.row_sys, gp->d.pos, gp->d.vel, gp->d.rng_state, gp->d.billboard_mx,
                    gp->d.respawn_mask, gp->d.respawn_row_pop, gp->d.respawn_list, gp->d.respawn_count, gp->d.scratch,
                    gp->d.respawn_groups };
    for (unsigned i = 0; i < sizeof(dev) / sizeof(dev[0]); i++)
        if (dev[i]) clapgpu_free(dev[i]);
    void *host[] = { gp->h_pos, gp->h_vel, gp->h_mx };
    for (unsigned i = 0; i < sizeof(host) / sizeof(host[0]); i++)
        if (host[i]) clapgpu_host_free(host[i]);
    memset(&gp->d, 0, sizeof(gp->d));
    gp->d_sys = NULL;
    gp->h_pos = gp->h_vel = gp->h_mx = NULL;
}

void gpu_particles_done(struct gpu_particles *gp)
{
    if (!gp) return;
    gp_free_device(gp);
    free(gp->ps); free(gp->seen); free(gp->sys_host);
    free(gp);
}

static void gp_sys_record(clapgpu_particle_system *r, const particle_system *ps, uint32_t first)
{
    memset(r, 0, sizeof(*r));
    transform_pos(&ps->e->xform, r->center);
    r->dist = ps->dist == PART_DIST_SQRT ? CLAPGPU_PART_DIST_SQRT : ps->dist == PART_DIST_CBRT ? CLAPGPU_PART_DIST_CBRT :
              ps->dist == PART_DIST_POW075 ? CLAPGPU_PART_DIST_POW075 : CLAPGPU_PART_DIST_LIN;
    r->radius = ps->radius; r->min_radius = ps->min_radius;
    r->radius_squared = ps->radius_squared; r->velocity = ps->velocity;
    r->first = first; r->count = ps->count;
}

/* (Re)build the device batch from the host lists: layout, particle state, work space. */
static int gp_rebuild(struct gpu_particles *gp)
{
    uint32_t n = 0;
    for (uint32_t s = 0; s < gp->n_sys; s++) {
        gp_sys_record(&gp->sys_host[s], gp->ps[s], n);
        n += (gp->ps[s]->count + 63u) & ~63u;
    }
    if (n == 0) n = 64;
    gp_free_device(gp);
    gp->n = gp->cap_n = n;
    const uint32_t rows = n / 64;
    uint32_t *row_sys = malloc(rows * sizeof(*row_sys));
    if (!row_sys) return _CERR_NOMEM;
    GP_CK(clapgpu_host_malloc((void **)&gp->h_pos, (size_t)n * 12));
    GP_CK(clapgpu_host_malloc((void **)&gp->h_vel, (size_t)n * 12));
    GP_CK(clapgpu_host_malloc((void **)&gp->h_mx, (size_t)(gp->n_sys ? gp->n_sys : 1) * 64));
    memset(gp->h_pos, 0, (size_t)n * 12);
    memset(gp->h_vel, 0, (size_t)n * 12);
    for (uint32_t r = 0; r < rows; r++) row_sys[r] = 0;
    for (uint32_t s = 0; s < gp->n_sys; s++) {
        const clapgpu_particle_system *r = &gp->sys_host[s];
        for (uint32_t k = r->first / 64; k < (r->first + ((r->count + 63u) & ~63u)) / 64; k++) row_sys[k] = s;
        particle *p;
        uint32_t i = r->first;
        list_for_each_entry(p, &gp->ps[s]->particles, entry) {
            memcpy(gp->h_pos + 3 * (size_t)i, p->pos, 12);
            memcpy(gp->h_vel + 3 * (size_t)i, p->velocity, 12);
            i++;
        }
    }
    const size_t sys_bytes = (size_t)(gp->n_sys ? gp->n_sys : 1) * sizeof(clapgpu_particle_system);
    GP_CK(clapgpu_malloc(&gp->d_sys, sys_bytes));
    GP_CK(clapgpu_malloc((void **)&gp->d.row_sys, (size_t)rows * 4));
    GP_CK(clapgpu_malloc((void **)&gp->d.pos, (size_t)n * 12));
    GP_CK(clapgpu_malloc((void **)&gp->d.vel, (size_t)n * 12));
    GP_CK(clapgpu_malloc((void **)&gp->d.rng_state, 16));
    GP_CK(clapgpu_malloc((void **)&gp->d.billboard_mx, (size_t)(gp->n_sys ? gp->n_sys : 1) * 64));
    GP_CK(clapgpu_malloc((void **)&gp->d.respawn_mask, (size_t)rows * 8));
    GP_CK(clapgpu_malloc((void **)&gp->d.respawn_row_pop, ((size_t)rows + 15) / 16 * 16));
    GP_CK(clapgpu_malloc((void **)&gp->d.respawn_list, (size_t)n * 4));
    GP_CK(clapgpu_malloc((void **)&gp->d.respawn_count, 4));
    GP_CK(clapgpu_malloc(&gp->d.scratch, clapgpu_visible_scratch_bytes(n) + 16));
    GP_CK(clapgpu_malloc((void **)&gp->d.respawn_groups, CLAPGPU_RESPAWN_GROUP_WORDS * 4));
    GP_CK(clapgpu_memset(gp->d.respawn_groups, 0, CLAPGPU_RESPAWN_GROUP_WORDS * 4, NULL));
    GP_CK(clapgpu_memset(gp->d.respawn_mask, 0, (size_t)rows * 8, NULL));
    GP_CK(clapgpu_memset(gp->d.respawn_row_pop, 0, ((size_t)rows + 15) / 16 * 16, NULL));
    GP_CK(clapgpu_memset(gp->d.respawn_count, 0, 4, NULL));
    GP_CK(clapgpu_memcpy_h2d((void *)gp->d.row_sys, row_sys, (size_t)rows * 4, NULL));
    GP_CK(clapgpu_memcpy_h2d(gp->d.pos, gp->h_pos, (size_t)n * 12, NULL));
    GP_CK(clapgpu_memcpy_h2d(gp->d.vel, gp->h_vel, (size_t)n * 12, NULL));
    GP_CK(clapgpu_stream_sync(NULL));
    free(row_sys);
    gp->d.n = n;
    gp->d.n_sys = gp->n_sys;
    gp->d.sys = gp->d_sys;
    gp->host_stale = false;
    return 0;
}

/* Bring p->pos / p->velocity of every mirrored system up to date with the device copy. */
int gpu_particles_sync_host(struct gpu_particles *gp)
{
    if (!gp || !gp->host_stale) return 0;
    GP_CK(clapgpu_memcpy_d2h(gp->h_pos, gp->d.pos, (size_t)gp->n * 12, NULL));
    GP_CK(clapgpu_memcpy_d2h(gp->h_vel, gp->d.vel, (size_t)gp->n * 12, NULL));
    GP_CK(clapgpu_stream_sync(NULL));
    for (uint32_t s = 0; s < gp->n_sys; s++) {
        if (!gp->ps[s]) continue;
        particle *p;
        size_t i = gp->sys_host[s].first;
        list_for_each_entry(p, &gp->ps[s]->particles, entry) {
            memcpy(p->pos, gp->h_pos + 3 * i, 12);
            memcpy(p->velocity, gp->h_vel + 3 * i, 12);
            i++;
        }
    }
    gp->host_stale = false;
    return 0;
}

int gpu_particles_update(struct gpu_particles *gp, struct mq *mq, struct scene *scene, bool scatter)
{
    if (!gp || !mq || !scene) return _CERR_INVALID_ARGUMENTS;

    /* the queue's particle systems, list order */
    uint32_t k = 0;
    bool changed = false;
    model3dtx *txm;
    entity3d *e, *it;
    list_for_each_entry(txm, &mq->txmodels, entry) {
        list_for_each_entry_iter(e, it, &txm->entities, entry) {
            if (!entity3d_matches(e, ENTITY3D_ALIVE) || !entity3d_matches(e, ENTITY3D_IS_PARTICLE) ||
                e->update != particles_update)
                continue;
            particle_system *ps = e->priv;
            if (k == gp->cap_sys) {
                gp->cap_sys = gp->cap_sys ? 2 * gp->cap_sys : 64;
                gp->ps = realloc(gp->ps, gp->cap_sys * sizeof(*gp->ps));
                gp->seen = realloc(gp->seen, gp->cap_sys * sizeof(*gp->seen));
                gp->sys_host = realloc(gp->sys_host, gp->cap_sys * sizeof(*gp->sys_host));
                if (!gp->ps || !gp->seen || !gp->sys_host) return _CERR_NOMEM;
            }
            if (k >= gp->n_sys || gp->ps[k] != ps || gp->sys_host[k].count != ps->count) changed = true;
            gp->seen[k++] = ps;
        }
    }
    if (k != gp->n_sys) changed = true;
    if (changed) {
        /* keep what the device has simulated so far (a system that is gone was freed by its owner: only
         * systems that are still in the queue are written back) */
        if (gp->n_sys && gp->d.pos && gp->host_stale) {
            for (uint32_t s = 0; s < gp->n_sys; s++) {
                bool still = false;
                for (uint32_t q = 0; q < k && !still; q++) still = gp->seen[q] == gp->ps[s];
                if (!still) gp->sys_host[s].count = 0, gp->ps[s] = NULL;
            }
            GP_CK(gpu_particles_sync_host(gp));
        }
        memcpy(gp->ps, gp->seen, k * sizeof(*gp->ps));
        gp->n_sys = k;
        GP_CK(gp_rebuild(gp));
    }
    if (gp->n_sys == 0) return 0;

    /* per-frame inputs: emitter centres / parameters, the libc stream position */
    for (uint32_t s = 0; s < gp->n_sys; s++)
        gp_sys_record(&gp->sys_host[s], gp->ps[s], gp->sys_host[s].first);
    gp->h_rng[0] = gp->h_rng[1] = gp_libc_state_get();
    GP_CK(clapgpu_memcpy_h2d(gp->d_sys, gp->sys_host, (size_t)gp->n_sys * sizeof(clapgpu_particle_system), NULL));
    GP_CK(clapgpu_memcpy_h2d(gp->d.rng_state, gp->h_rng, 16, NULL));

    GP_CK(clapgpu_particles_update(NULL, &gp->d, (const float *)scene->camera->view.main.view_mx));

    GP_CK(clapgpu_memcpy_d2h(gp->h_pos, gp->d.pos, (size_t)gp->n * 12, NULL));
    GP_CK(clapgpu_memcpy_d2h(gp->h_mx, gp->d.billboard_mx, (size_t)gp->n_sys * 64, NULL));
    GP_CK(clapgpu_memcpy_d2h(gp->h_rng, gp->d.rng_state, 16, NULL));
    if (scatter) GP_CK(clapgpu_memcpy_d2h(gp->h_vel, gp->d.vel, (size_t)gp->n * 12, NULL));
    GP_CK(clapgpu_stream_sync(NULL));

    gp_libc_state_set(gp->h_rng[1] & 0xffffffffffffull);                    /* other drand48() users go on from here */
    for (uint32_t s = 0; s < gp->n_sys; s++) {
        particle_system *ps = gp->ps[s];
        const size_t first = gp->sys_host[s].first;
        memcpy(ps->pos_array, gp->h_pos + 3 * first, (size_t)ps->count * sizeof(vec3));   /* particle.c:116 */
        memcpy(ps->e->mx, gp->h_mx + 16 * (size_t)s, sizeof(mat4x4));                     /* particle.c:93-100 */
        if (scatter) {
            particle *p;
            size_t i = first;
            list_for_each_entry(p, &ps->particles, entry) {
                memcpy(p->pos, gp->h_pos + 3 * i, 12);
                memcpy(p->velocity, gp->h_vel + 3 * i, 12);
                i++;
            }
        }
    }
    gp->host_stale = !scatter;
    return 0;
}

int gpu_particle_system_position(struct gpu_particles *gp, particle_system *ps, const vec3 center)
{
    uint32_t s;
    for (s = 0; gp && s < gp->n_sys; s++)
        if (gp->ps[s] == ps) break;
    if (!gp || s == gp->n_sys || !ps->attached) {           /* not mirrored yet, or nothing to carry along */
        particle_system_position(ps, center);
        return 0;
    }
    vec3 delta;
    transform_pos(&ps->e->xform, delta);
    vec3_sub(delta, center, delta);
    if (vec3_mul_inner(delta, delta) == 0.0)
        return 0;
    transform_set_pos(&ps->e->xform, center);
    /* the system's slice of the device copy: down, the reference's own vec3_add, up */
    const size_t first = gp->sys_host[s].first, bytes = (size_t)ps->count * 12;
    GP_CK(clapgpu_memcpy_d2h(gp->h_pos + 3 * first, gp->d.pos + 3 * first, bytes, NULL));
    GP_CK(clapgpu_stream_sync(NULL));
    for (size_t i = 0; i < ps->count; i++)
        vec3_add(gp->h_pos + 3 * (first + i), gp->h_pos + 3 * (first + i), delta);
    GP_CK(clapgpu_memcpy_h2d(gp->d.pos + 3 * first, gp->h_pos + 3 * first, bytes, NULL));
    GP_CK(clapgpu_stream_sync(NULL));
    gp->host_stale = true;
    return 0;
}

/* CONFIG_GPU_SCENE: the engine's own name (particle.h:47), served by the binding for the bound particle mirror.  This
 * file is included while `particle_system_position` still names the reference's body (ref_particle_system_position:
 * the includer renames it around particle.c, like gpu-exports.inc.c's functions), so the calls above reach that body. */
#ifdef CONFIG_GPU_SCENE
static struct gpu_particles *g_bound_particles;
void gpu_particles_bind(struct gpu_particles *gp) { g_bound_particles = gp; }
#undef particle_system_position
void ref_particle_system_position(particle_system *ps, const vec3 center);
void particle_system_position(particle_system *ps, const vec3 center)
{
    if (g_bound_particles && !gpu_particle_system_position(g_bound_particles, ps, center))
        return;
    ref_particle_system_position(ps, center);
}
#endif
