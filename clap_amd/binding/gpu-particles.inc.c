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
#include <stdio.h>
#include <time.h>
#include "clapgpu.h"
#include "gpu-scene.h"                               /* gpu_scene_par_for: the bindings' worker threads */

struct gpu_particles {
    uint32_t                n_sys, cap_sys, n;       /* n = padded particle slots (systems start at multiples of 64) */
    particle_system         **ps, **seen;    /* mirrored systems; this frame's walk */
    clapgpu_particle_system *sys_host;
    clapgpu_particles       d;                       /* device descriptor */
    void                    *d_sys;
    float                   *h_pos, *h_vel, *h_mx;   /* page-locked staging */
    uint32_t                cap_n, cap_sys_dev;      /* particle slots / systems the buffers hold */
    uint64_t                h_rng[2];
    bool                    host_stale;              /* p->pos / p->velocity older than the device copy */
    /* Up to GP_MAPPED_MAX particle slots the batch LIVES in page-locked, device-mapped host memory: positions, velocities,
     * the system table, billboard matrices and the libc stream state are read and written by the kernels through their
     * device aliases -- a frame is the three launches and one wait, none of the two copies up and three or four down that
     * cost a testbed-sized frame (20 systems, 7 k particles) 0.25 ms against the reference's 0.10. */
    bool                    mapped;
    clapgpu_particle_system *m_sys;                  /* mapped copy of sys_host */
    uint64_t                *m_rng;                  /* mapped libc stream state (2 words) */
};
#define GP_MAPPED_MAX 65536u

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
    if (!*out) return _CERR_NOMEM;
    gpu_scene_pool_ref();                            /* gpu_scene_par_for: the struct particle write-back */
    return 0;
}

static void gp_free_device(struct gpu_particles *gp)
{
    void *dev[] = { gp->d_sys, (void *)gp->d.row_sys, gp->d.pos, gp->d.vel, gp->d.rng_state, gp->d.billboard_mx,
                    gp->d.respawn_mask, gp->d.respawn_row_pop, gp->d.respawn_list, gp->d.respawn_count, gp->d.scratch,
                    gp->d.respawn_groups };
    if (gp->mapped)                                  /* these five are device aliases of the host buffers freed below */
        dev[0] = dev[2] = dev[3] = dev[4] = dev[5] = NULL;
    for (unsigned i = 0; i < sizeof(dev) / sizeof(dev[0]); i++)
        if (dev[i]) clapgpu_free(dev[i]);
    void *host[] = { gp->h_pos, gp->h_vel, gp->h_mx, gp->m_sys, gp->m_rng };
    for (unsigned i = 0; i < sizeof(host) / sizeof(host[0]); i++)
        if (host[i]) clapgpu_host_free(host[i]);
    memset(&gp->d, 0, sizeof(gp->d));
    gp->d_sys = NULL;
    gp->h_pos = gp->h_vel = gp->h_mx = NULL;
    gp->m_sys = NULL; gp->m_rng = NULL;
    gp->mapped = false;
    gp->cap_n = gp->cap_sys_dev = 0;
}

void gpu_particles_done(struct gpu_particles *gp)
{
    if (!gp) return;
    gp_free_device(gp);
    free(gp->ps); free(gp->seen); free(gp->sys_host);
    free(gp);
    gpu_scene_pool_unref();
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

/* (Re)build the device batch from the host lists: layout, particle state, work space.  Buffers are kept while they are
 * large enough (a system appearing or dying re-lays the batch out, it should not cost a dozen allocations). */
static uint32_t gp_pow2(uint32_t x, uint32_t lo) { uint32_t p = lo; while (p < x) p *= 2; return p; }

static int gp_rebuild(struct gpu_particles *gp)
{
    uint32_t n = 0;
    for (uint32_t s = 0; s < gp->n_sys; s++) {
        gp_sys_record(&gp->sys_host[s], gp->ps[s], n);
        n += (gp->ps[s]->count + 63u) & ~63u;
    }
    if (n == 0) n = 64;
    const uint32_t rows = n / 64;
    uint32_t *row_sys = malloc(rows * sizeof(*row_sys));
    if (!row_sys) return _CERR_NOMEM;
    if (!gp->d.pos || n > gp->cap_n || gp->n_sys > gp->cap_sys_dev) {
        gp_free_device(gp);
        const uint32_t cap = gp_pow2(n, 4096), cap_rows = cap / 64, cap_sys = gp_pow2(gp->n_sys ? gp->n_sys : 1, 16);
        const bool mapped = cap <= GP_MAPPED_MAX && !getenv("GPU_PARTICLES_STAGED");
        const size_t sys_bytes = (size_t)cap_sys * sizeof(clapgpu_particle_system);
        if (mapped) {
            void *a_pos = NULL, *a_vel = NULL, *a_mx = NULL, *a_sys = NULL, *a_rng = NULL;
            GP_CK(clapgpu_host_malloc_mapped((void **)&gp->h_pos, &a_pos, (size_t)cap * 12));
            GP_CK(clapgpu_host_malloc_mapped((void **)&gp->h_vel, &a_vel, (size_t)cap * 12));
            GP_CK(clapgpu_host_malloc_mapped((void **)&gp->h_mx, &a_mx, (size_t)cap_sys * 64));
            GP_CK(clapgpu_host_malloc_mapped((void **)&gp->m_sys, &a_sys, sys_bytes));
            GP_CK(clapgpu_host_malloc_mapped((void **)&gp->m_rng, &a_rng, 16));
            gp->mapped = true;
            gp->d_sys = a_sys; gp->d.pos = a_pos; gp->d.vel = a_vel; gp->d.rng_state = a_rng; gp->d.billboard_mx = a_mx;
        } else {
            GP_CK(clapgpu_host_malloc((void **)&gp->h_pos, (size_t)cap * 12));
            GP_CK(clapgpu_host_malloc((void **)&gp->h_vel, (size_t)cap * 12));
            GP_CK(clapgpu_host_malloc((void **)&gp->h_mx, (size_t)cap_sys * 64));
            GP_CK(clapgpu_malloc(&gp->d_sys, sys_bytes));
            GP_CK(clapgpu_malloc((void **)&gp->d.pos, (size_t)cap * 12));
            GP_CK(clapgpu_malloc((void **)&gp->d.vel, (size_t)cap * 12));
            GP_CK(clapgpu_malloc((void **)&gp->d.rng_state, 16));
            GP_CK(clapgpu_malloc((void **)&gp->d.billboard_mx, (size_t)cap_sys * 64));
        }
        GP_CK(clapgpu_malloc((void **)&gp->d.row_sys, (size_t)cap_rows * 4));
        GP_CK(clapgpu_malloc((void **)&gp->d.respawn_mask, (size_t)cap_rows * 8));
        GP_CK(clapgpu_malloc((void **)&gp->d.respawn_row_pop, ((size_t)cap_rows + 15) / 16 * 16));
        GP_CK(clapgpu_malloc((void **)&gp->d.respawn_list, (size_t)cap * 4));
        GP_CK(clapgpu_malloc((void **)&gp->d.respawn_count, 4));
        GP_CK(clapgpu_malloc(&gp->d.scratch, clapgpu_visible_scratch_bytes(cap) + 16));
        GP_CK(clapgpu_malloc((void **)&gp->d.respawn_groups, CLAPGPU_RESPAWN_GROUP_WORDS * 4));
        gp->cap_n = cap;
        gp->cap_sys_dev = cap_sys;
    }
    gp->n = n;
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
    GP_CK(clapgpu_memset(gp->d.respawn_groups, 0, CLAPGPU_RESPAWN_GROUP_WORDS * 4, NULL));
    GP_CK(clapgpu_memset(gp->d.respawn_mask, 0, (size_t)rows * 8, NULL));
    GP_CK(clapgpu_memset(gp->d.respawn_row_pop, 0, ((size_t)rows + 15) / 16 * 16, NULL));
    GP_CK(clapgpu_memset(gp->d.respawn_count, 0, 4, NULL));
    GP_CK(clapgpu_memcpy_h2d((void *)gp->d.row_sys, row_sys, (size_t)rows * 4, NULL));
    if (!gp->mapped) {
        GP_CK(clapgpu_memcpy_h2d(gp->d.pos, gp->h_pos, (size_t)n * 12, NULL));
        GP_CK(clapgpu_memcpy_h2d(gp->d.vel, gp->h_vel, (size_t)n * 12, NULL));
    }
    GP_CK(clapgpu_stream_sync(NULL));
    free(row_sys);
    gp->d.n = n;
    gp->d.n_sys = gp->n_sys;
    gp->d.sys = gp->d_sys;
    gp->host_stale = false;
    return 0;
}

/* systems [lo, hi): every struct particle's pos / velocity from the staging arrays (a linked list per system: memory
 * latency, split over the bindings' workers above GP_PAR_MIN particles -- 1 M particles: 10 ms on one core) */
#define GP_PAR_MIN 65536u
static void gp_structs_back(void *ctx, uint32_t lo, uint32_t hi)
{
    struct gpu_particles *gp = ctx;
    for (uint32_t s = lo; s < hi; s++) {
        if (!gp->ps[s]) continue;
        particle *p;
        size_t i = gp->sys_host[s].first;
        list_for_each_entry(p, &gp->ps[s]->particles, entry) {
            memcpy(p->pos, gp->h_pos + 3 * i, 12);
            memcpy(p->velocity, gp->h_vel + 3 * i, 12);
            i++;
        }
    }
}

static void gp_structs_back_all(struct gpu_particles *gp)
{
    static uint32_t par_min;                             /* (GPU_PARTICLES_PAR_MIN: the threshold from the environment, for the sanitizer runs) */
    if (!par_min) { const char *env = getenv("GPU_PARTICLES_PAR_MIN"); par_min = env && atol(env) > 0 ? (uint32_t)atol(env) : GP_PAR_MIN; }
    gpu_scene_par_for(gp_structs_back, gp, gp->n_sys, gp->n >= par_min && gp->n_sys >= 16 ? 8 : 1);
}

/* Bring p->pos / p->velocity of every mirrored system up to date with the device copy. */
int gpu_particles_sync_host(struct gpu_particles *gp)
{
    if (!gp || !gp->host_stale) return 0;
    if (!gp->mapped) {                                /* mapped: h_pos / h_vel ARE the batch, and every update has waited for its kernels */
        GP_CK(clapgpu_memcpy_d2h(gp->h_pos, gp->d.pos, (size_t)gp->n * 12, NULL));
        GP_CK(clapgpu_memcpy_d2h(gp->h_vel, gp->d.vel, (size_t)gp->n * 12, NULL));
        GP_CK(clapgpu_stream_sync(NULL));
    }
    gp_structs_back_all(gp);
    gp->host_stale = false;
    return 0;
}

int gpu_particles_update(struct gpu_particles *gp, struct mq *mq, struct scene *scene, bool scatter)
{
    if (!gp || !mq || !scene) return _CERR_INVALID_ARGUMENTS;
    const bool timing = getenv("GPU_PARTICLES_TIMING") != NULL;
    struct timespec tp_[4];
    if (timing) clock_gettime(CLOCK_MONOTONIC, &tp_[0]);

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
    if (timing) clock_gettime(CLOCK_MONOTONIC, &tp_[1]);
    if (gp->mapped) {
        memcpy(gp->m_sys, gp->sys_host, (size_t)gp->n_sys * sizeof(clapgpu_particle_system));
        gp->m_rng[0] = gp->m_rng[1] = gp->h_rng[0];
        GP_CK(clapgpu_particles_update(NULL, &gp->d, (const float *)scene->camera->view.main.view_mx));
        GP_CK(clapgpu_stream_sync(NULL));
        gp->h_rng[0] = gp->m_rng[0]; gp->h_rng[1] = gp->m_rng[1];
    } else {
        GP_CK(clapgpu_memcpy_h2d(gp->d_sys, gp->sys_host, (size_t)gp->n_sys * sizeof(clapgpu_particle_system), NULL));
        GP_CK(clapgpu_memcpy_h2d(gp->d.rng_state, gp->h_rng, 16, NULL));

        GP_CK(clapgpu_particles_update(NULL, &gp->d, (const float *)scene->camera->view.main.view_mx));

        GP_CK(clapgpu_memcpy_d2h(gp->h_pos, gp->d.pos, (size_t)gp->n * 12, NULL));
        GP_CK(clapgpu_memcpy_d2h(gp->h_mx, gp->d.billboard_mx, (size_t)gp->n_sys * 64, NULL));
        GP_CK(clapgpu_memcpy_d2h(gp->h_rng, gp->d.rng_state, 16, NULL));
        if (scatter) GP_CK(clapgpu_memcpy_d2h(gp->h_vel, gp->d.vel, (size_t)gp->n * 12, NULL));
        GP_CK(clapgpu_stream_sync(NULL));
    }

    if (timing) clock_gettime(CLOCK_MONOTONIC, &tp_[2]);
    gp_libc_state_set(gp->h_rng[1] & 0xffffffffffffull);                    /* other drand48() users go on from here */
    for (uint32_t s = 0; s < gp->n_sys; s++) {
        particle_system *ps = gp->ps[s];
        const size_t first = gp->sys_host[s].first;
        memcpy(ps->pos_array, gp->h_pos + 3 * first, (size_t)ps->count * sizeof(vec3));   /* particle.c:116 */
        memcpy(ps->e->mx, gp->h_mx + 16 * (size_t)s, sizeof(mat4x4));                     /* particle.c:93-100 */
    }
    if (scatter) gp_structs_back_all(gp);
    gp->host_stale = !scatter;
    if (timing) {
        clock_gettime(CLOCK_MONOTONIC, &tp_[3]);
        double d[3];
        for (int i = 0; i < 3; i++) d[i] = (tp_[i + 1].tv_sec - tp_[i].tv_sec) * 1e3 + (tp_[i + 1].tv_nsec - tp_[i].tv_nsec) * 1e-6;
        fprintf(stderr, "gpu_particles_update: walk + records %.3f  device %.3f  back into the systems %.3f ms (%s)\n", d[0], d[1], d[2],
                gp->mapped ? "mapped" : "staged");
    }
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
    if (!gp->mapped) {
        GP_CK(clapgpu_memcpy_d2h(gp->h_pos + 3 * first, gp->d.pos + 3 * first, bytes, NULL));
        GP_CK(clapgpu_stream_sync(NULL));
    }
    for (size_t i = 0; i < ps->count; i++)
        vec3_add(gp->h_pos + 3 * (first + i), gp->h_pos + 3 * (first + i), delta);
    if (!gp->mapped) {
        GP_CK(clapgpu_memcpy_h2d(gp->d.pos + 3 * first, gp->h_pos + 3 * first, bytes, NULL));
        GP_CK(clapgpu_stream_sync(NULL));
    }
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
