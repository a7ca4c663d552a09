/*
 * gpu-anim.inc.c -- CLAP-side binding of libclapgpu for skeletal animation.
 *
 * `struct channel`, animation_next(), ani_current() and animated_update() are private to core/model.c,
 * so this file is meant to be #include'd at the end of that translation unit (the drop-in checker
 * oracle/ref/dropin.c includes it the same way).  It replaces the animated_update() tail of
 * default_update (model.c:1563-1592, 1715-1716) for every animated entity of a queue by ONE batched
 * call per model:
 *
 *   gpu_anim_update(ga, gs, mq, scene)      (gs: the queue's gpu_scene, or NULL when mq_update stays on the host)
 *       for every ALIVE entity whose model has animations: the clock and queue bookkeeping of
 *       animated_update on the host (frame time in double, sfx_state reset, animation_next when the
 *       queue is empty or the animation has ended, frame_cb / frame_sfx callbacks -- the reference's own
 *       functions, in list order), channels_transform + one_joint_transform on the device
 *       (clapgpu_pose_update), and the results scattered back to where the draw path and the game read
 *       them: e->joint_transforms[j] (UNIFORM_JOINT_TRANSFORMS), e->joints[j].translation / rotation /
 *       scale / pos.
 *
 * It runs after the entities' transforms are current (mq_update / gpu_mq_update), because a joint's
 * world position uses e->mx (model.c:1400).  gpu-scene.c batches animated entities' transforms when
 * told that the pose is taken care of here (gpu_scene_animation_elsewhere).
 *
 * Not mirrored: joint.off[] (the search cursor, irrelevant for strictly increasing key times),
 * joint.global (scratch).  Parity bar of this row: 1e-5 relative; since round 4 the kernel performs the
 * reference's own arithmetic and every float comes back with the reference's bits (clap_dropin anim: tolerance 0, -0 is not +0).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <stdio.h>
#include "clapgpu.h"
#include "gpu-scene.h"

struct ga_model {
    model3d             *model;
    uint32_t            J, n_anims;
    int32_t             *depth_host;                      /* < 0: not under joint 0, never written (model.c:1583) */
    void                *d_parent, *d_depth, *d_root_pose, *d_invmx, *d_bind, *d_chan_table, *d_times, *d_data, *d_packed;
    clapgpu_skeleton    sk;
    clapgpu_animations  an;
    /* the model's animated entities this frame, list order */
    uint32_t            n, n_prev, cap;
    entity3d            **ents, **prev;
    double              *frame_time;
    int                 *anim_of;
    void                *d_anim, *d_ftime, *d_emx, *d_trs, *d_jt, *d_jpos;
    uint32_t            *h_anim;                          /* page-locked staging, device-mapped: a_* are the device's aliases */
    float               *h_ftime, *h_emx, *h_trs, *h_jt, *h_jpos;
    void                *a_anim, *a_ftime, *a_emx, *a_trs, *a_jt, *a_jpos;
    bool                mapped_prev;                      /* the last frame ran on the aliases: T/R/S live in h_trs */
};

struct gpu_anim {
    struct ga_model *models;
    uint32_t        n_models, cap_models;
    uint32_t        walk_seen; bool walk_seen_valid;     /* gpu_scene_walk_generation() at the last update */
};

#define GA_CK(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

int gpu_anim_init(struct gpu_anim **out, int device)
{
    if (!out) return _CERR_INVALID_ARGUMENTS;
    int rc = clapgpu_init(device);
    if (rc) return rc;
    *out = calloc(1, sizeof(**out));
    if (!*out) return _CERR_NOMEM;
    gpu_scene_pool_ref();                                /* gpu_scene_par_for: the write-back of many characters */
    return 0;
}

static void ga_free_batch(struct ga_model *m)
{
    void *dev[] = { m->d_anim, m->d_ftime, m->d_emx, m->d_trs, m->d_jt, m->d_jpos };
    for (unsigned i = 0; i < sizeof(dev) / sizeof(dev[0]); i++)
        if (dev[i]) clapgpu_free(dev[i]);
    void *host[] = { m->h_anim, m->h_ftime, m->h_emx, m->h_trs, m->h_jt, m->h_jpos };
    for (unsigned i = 0; i < sizeof(host) / sizeof(host[0]); i++)
        if (host[i]) clapgpu_host_free(host[i]);
    m->d_anim = m->d_ftime = m->d_emx = m->d_trs = m->d_jt = m->d_jpos = NULL;
    m->h_anim = NULL; m->h_ftime = m->h_emx = m->h_trs = m->h_jt = m->h_jpos = NULL;
    m->a_anim = m->a_ftime = m->a_emx = m->a_trs = m->a_jt = m->a_jpos = NULL;
}

void gpu_anim_done(struct gpu_anim *ga)
{
    if (!ga) return;
    for (uint32_t k = 0; k < ga->n_models; k++) {
        struct ga_model *m = &ga->models[k];
        void *dev[] = { m->d_parent, m->d_depth, m->d_root_pose, m->d_invmx, m->d_bind, m->d_chan_table, m->d_times, m->d_data, m->d_packed };
        for (unsigned i = 0; i < sizeof(dev) / sizeof(dev[0]); i++)
            if (dev[i]) clapgpu_free(dev[i]);
        ga_free_batch(m);
        free(m->ents); free(m->prev); free(m->frame_time); free(m->anim_of); free(m->depth_host);
    }
    free(ga->models);
    free(ga);
    gpu_scene_pool_unref();
}

/* Skeleton constants and the keyframe pools of one model3d, uploaded once. */
static int ga_model_build(struct ga_model *m, model3d *model)
{
    const uint32_t J = model->nr_joints, A = model->anis.da.nr_el;
    memset(m, 0, sizeof(*m));
    m->model = model; m->J = J; m->n_anims = A;

    int32_t *parent = malloc(J * 4), *depth = malloc(J * 4);
    float *invmx = malloc((size_t)J * 64), *bind = malloc((size_t)J * 64);
    if (!parent || !depth || !invmx || !bind) { free(parent); free(depth); free(invmx); free(bind); return _CERR_NOMEM; }
    for (uint32_t j = 0; j < J; j++) { parent[j] = -1; depth[j] = -1; }
    for (uint32_t j = 0; j < J; j++) {                        /* model_joint.children, inverted */
        int *c;
        darray_for_each(c, model->joints[j].children)
            if (*c >= 0 && (uint32_t)*c < J) parent[*c] = (int32_t)j;
        memcpy(invmx + 16 * j, model->joints[j].invmx, 64);
        memcpy(bind + 16 * j, model->joints[j].bind, 64);
    }
    /* levels under joint 0: one_joint_transform(e, 0, -1) recurses from there (model.c:1583) */
    uint32_t n_levels = 1;
    depth[0] = 0;
    for (bool grew = true; grew;) {
        grew = false;
        for (uint32_t j = 1; j < J; j++)
            if (depth[j] < 0 && parent[j] >= 0 && depth[parent[j]] >= 0) {
                depth[j] = depth[parent[j]] + 1;
                if ((uint32_t)depth[j] + 1 > n_levels) n_levels = depth[j] + 1;
                grew = true;
            }
    }

    /* chan_table[a][j][path] = (time_off, data_off, nr, 0); a later channel of the same (joint, path) wins */
    size_t n_times = 0, n_data = 0;
    for (uint32_t a = 0; a < A; a++)
        for (unsigned int ch = 0; ch < model->anis.x[a].nr_channels; ch++) {
            const struct channel *c = &model->anis.x[a].channels[ch];
            n_times += c->nr;
            n_data += (size_t)c->nr * (c->path == PATH_ROTATION ? 4 : 3);
        }
    uint32_t *table = calloc((size_t)(A ? A : 1) * J * 3 * 4, 4);
    float *times = malloc((n_times ? n_times : 1) * 4), *data = malloc((n_data ? n_data : 1) * 4);
    if (!table || !times || !data) {
        free(parent); free(depth); free(invmx); free(bind); free(table); free(times); free(data);
        return _CERR_NOMEM;
    }
    size_t t_at = 0, d_at = 0;
    for (uint32_t a = 0; a < A; a++)
        for (unsigned int ch = 0; ch < model->anis.x[a].nr_channels; ch++) {
            const struct channel *c = &model->anis.x[a].channels[ch];
            if (!c->nr || !c->data || !c->time || c->target >= J || c->path >= PATH_NONE) continue;   /* model.c:1301 */
            const unsigned int w = c->path == PATH_ROTATION ? 4 : 3;
            uint32_t *row = table + (((size_t)a * J + c->target) * 3 + c->path) * 4;
            row[0] = (uint32_t)t_at; row[1] = (uint32_t)d_at; row[2] = c->nr; row[3] = 0;
            memcpy(times + t_at, c->time, (size_t)c->nr * 4);
            for (unsigned int k = 0; k < c->nr; k++)          /* keys are `stride` bytes apart in the reference */
                memcpy(data + d_at + (size_t)k * w, (const char *)c->data + (size_t)k * c->stride, w * 4);
            t_at += c->nr;
            d_at += (size_t)c->nr * w;
        }

    GA_CK(clapgpu_malloc(&m->d_parent, J * 4));          GA_CK(clapgpu_memcpy_h2d(m->d_parent, parent, J * 4, NULL));
    GA_CK(clapgpu_malloc(&m->d_depth, J * 4));           GA_CK(clapgpu_memcpy_h2d(m->d_depth, depth, J * 4, NULL));
    GA_CK(clapgpu_malloc(&m->d_root_pose, 64));          GA_CK(clapgpu_memcpy_h2d(m->d_root_pose, model->root_pose, 64, NULL));
    GA_CK(clapgpu_malloc(&m->d_invmx, (size_t)J * 64));  GA_CK(clapgpu_memcpy_h2d(m->d_invmx, invmx, (size_t)J * 64, NULL));
    GA_CK(clapgpu_malloc(&m->d_bind, (size_t)J * 64));   GA_CK(clapgpu_memcpy_h2d(m->d_bind, bind, (size_t)J * 64, NULL));
    const size_t tb = (size_t)(A ? A : 1) * J * 3 * 16;
    GA_CK(clapgpu_malloc(&m->d_chan_table, tb));         GA_CK(clapgpu_memcpy_h2d(m->d_chan_table, table, tb, NULL));
    GA_CK(clapgpu_malloc(&m->d_times, (t_at ? t_at : 1) * 4)); GA_CK(clapgpu_memcpy_h2d(m->d_times, times, t_at * 4, NULL));
    GA_CK(clapgpu_malloc(&m->d_data, (d_at ? d_at : 1) * 4));  GA_CK(clapgpu_memcpy_h2d(m->d_data, data, d_at * 4, NULL));
    uint32_t max_keys = 0;
    for (size_t q = 0; q < (size_t)(A ? A : 1) * J * 3; q++)
        if (table[4 * q + 2] > max_keys) max_keys = table[4 * q + 2];
    GA_CK(clapgpu_stream_sync(NULL));
    m->depth_host = depth;
    free(parent); free(invmx); free(bind); free(table); free(times); free(data);

    m->sk = (clapgpu_skeleton){ .nr_joints = J, .n_levels = n_levels, .parent = m->d_parent, .depth = m->d_depth,
                                .root_pose = m->d_root_pose, .invmx = m->d_invmx, .bind = m->d_bind };
    m->an = (clapgpu_animations){ .n_anims = A, .n_times = (uint32_t)t_at, .chan_table = m->d_chan_table,
                                  .times = m->d_times, .data = m->d_data, .n_data = (uint32_t)d_at };
    if (!max_keys) max_keys = 1;                          /* animations without a single key: every path keeps its value */
    if (J <= 256 && A) {
        /* the key-major copy of the pools, once per model, with what quat_slerp derives from each rotation key pair alone
         * (acos of the inner product and its sine, by this host's libm as in interp.h:107-110): the pose kernel's searches
         * then run without LDS bank conflicts, its key gathers are contiguous rows and its slerp is the reference's
         * (clapgpu.h: clapgpu_animations_pack) */
        uint32_t layout = 0;
        GA_CK(clapgpu_malloc(&m->d_packed, clapgpu_animations_packed_bytes(A, max_keys, J)));
        GA_CK(clapgpu_animations_pack(NULL, &m->an, J, max_keys, m->d_packed, &layout));
        m->an.packed = m->d_packed;
        m->an.packed_keys = max_keys;
        m->an.packed_layout = layout;
    }
    return 0;
}

static struct ga_model *ga_model_of(struct gpu_anim *ga, model3d *model, int *rc)
{
    for (uint32_t k = 0; k < ga->n_models; k++)
        if (ga->models[k].model == model) return &ga->models[k];
    if (ga->n_models == ga->cap_models) {
        ga->cap_models = ga->cap_models ? 2 * ga->cap_models : 8;
        ga->models = realloc(ga->models, ga->cap_models * sizeof(*ga->models));
        if (!ga->models) { *rc = _CERR_NOMEM; return NULL; }
    }
    struct ga_model *m = &ga->models[ga->n_models];
    *rc = ga_model_build(m, model);
    if (*rc) return NULL;
    ga->n_models++;
    return m;
}

static struct ga_model *ga_model_peek(struct gpu_anim *ga, const model3d *model)
{
    for (uint32_t k = 0; k < ga->n_models; k++)
        if (ga->models[k].model == model) return &ga->models[k];
    return NULL;
}

static int ga_reserve(struct ga_model *m, uint32_t n)
{
    if (n <= m->cap) return 0;
    uint32_t cap = m->cap ? m->cap : 64;
    while (cap < n) cap *= 2;
    m->ents = realloc(m->ents, cap * sizeof(*m->ents));
    m->prev = realloc(m->prev, cap * sizeof(*m->prev));
    m->frame_time = realloc(m->frame_time, cap * sizeof(*m->frame_time));
    m->anim_of = realloc(m->anim_of, cap * sizeof(*m->anim_of));
    if (!m->ents || !m->prev || !m->frame_time || !m->anim_of) return _CERR_NOMEM;
    ga_free_batch(m);
    const size_t cj = (size_t)cap * m->J;
    GA_CK(clapgpu_malloc(&m->d_anim, (size_t)cap * 4));   GA_CK(clapgpu_host_malloc_mapped((void **)&m->h_anim, &m->a_anim, (size_t)cap * 4));
    GA_CK(clapgpu_malloc(&m->d_ftime, (size_t)cap * 4));  GA_CK(clapgpu_host_malloc_mapped((void **)&m->h_ftime, &m->a_ftime, (size_t)cap * 4));
    GA_CK(clapgpu_malloc(&m->d_emx, (size_t)cap * 64));   GA_CK(clapgpu_host_malloc_mapped((void **)&m->h_emx, &m->a_emx, (size_t)cap * 64));
    GA_CK(clapgpu_malloc(&m->d_trs, cj * 40));            GA_CK(clapgpu_host_malloc_mapped((void **)&m->h_trs, &m->a_trs, cj * 40));
    GA_CK(clapgpu_malloc(&m->d_jt, cj * 64));             GA_CK(clapgpu_host_malloc_mapped((void **)&m->h_jt, &m->a_jt, cj * 64));
    GA_CK(clapgpu_malloc(&m->d_jpos, cj * 16));           GA_CK(clapgpu_host_malloc_mapped((void **)&m->h_jpos, &m->a_jpos, cj * 16));
    GA_CK(clapgpu_memset(m->d_jt, 0, cj * 64, NULL));
    GA_CK(clapgpu_memset(m->d_jpos, 0, cj * 16, NULL));
    memset(m->h_jt, 0, cj * 64);
    memset(m->h_jpos, 0, cj * 16);
    m->cap = cap;
    m->n_prev = 0;                                         /* the device T/R/S are gone: upload again */
    return 0;
}

#define GA_PAR_MIN 65536u
/* (GPU_ANIM_PAR_MIN: the same threshold from the environment -- the sanitizer runs force the worker path on small scenes) */
static size_t ga_par_min(void)
{
    static size_t cached;
    if (!cached) {
        const char *env = getenv("GPU_ANIM_PAR_MIN");
        cached = env && atol(env) > 0 ? (size_t)atol(env) : GA_PAR_MIN;
    }
    return cached;
}
#define GA_MAPPED_MAX 65536u        /* joints of a model's batch up to which the pose works on the mapped staging arrays */
/* characters [lo, hi) of a model: T / R / S, palette and world position of every joint from the downloaded arrays */
static void ga_joints_back(struct ga_model *m, uint32_t lo, uint32_t hi)
{
    const uint32_t J = m->J;
    for (uint32_t c = lo; c < hi; c++) {
        entity3d *e = m->ents[c];
        for (uint32_t j = 0; j < J; j++) {
            const size_t q = (size_t)c * J + j;
            const float *t = m->h_trs + 10 * q;
            memcpy(e->joints[j].translation, t, 12);
            memcpy(e->joints[j].rotation, t + 3, 16);
            memcpy(e->joints[j].scale, t + 7, 12);
        }
        /* joints outside joint 0's tree are written neither by the device nor by the reference */
        for (uint32_t j = 0; j < J; j++) {
            const size_t q = (size_t)c * J + j;
            if (m->depth_host[j] < 0) continue;
            memcpy(e->joint_transforms[j], m->h_jt + 16 * q, 64);
            memcpy(e->joints[j].pos, m->h_jpos + 4 * q, 16);
        }
    }
}

static void ga_joints_back_range(void *m, uint32_t lo, uint32_t hi) { ga_joints_back(m, lo, hi); }

static int ga_threads(void)
{
    static int cached;
    if (!cached) {
        long n = sysconf(_SC_NPROCESSORS_ONLN);
        const char *env = getenv("GPU_ANIM_THREADS");            /* tuning knob (gpu_scene_par_for clamps to the pool's size) */
        if (env && atoi(env) > 0) n = atoi(env);
        else if (n > 24) n = 24;
        cached = n < 1 ? 1 : (int)n;
    }
    return cached;
}

int gpu_anim_update(struct gpu_anim *ga, struct gpu_scene *gs, struct mq *mq, struct scene *s)
{
    if (!ga || !mq || !s) return _CERR_INVALID_ARGUMENTS;
    const double time = clap_get_current_time(s->clap_ctx);
    const bool timing = getenv("GPU_ANIM_TIMING") != NULL;
    struct timespec ts_[5];
    if (timing) clock_gettime(CLOCK_MONOTONIC, &ts_[0]);

    for (uint32_t k = 0; k < ga->n_models; k++) {
        struct ga_model *m = &ga->models[k];
        entity3d **t = m->prev; m->prev = m->ents; m->ents = t;
        m->n_prev = m->n;
        m->n = 0;
    }
    const uint32_t walk_now = gs ? gpu_scene_walk_generation(gs) : 0;
    const bool same_walk = gs && ga->walk_seen == walk_now && ga->walk_seen_valid;
    ga->walk_seen = walk_now; ga->walk_seen_valid = gs != NULL;
    /* animated_update's host part, list order (model.c:1563-1581) */
    model3dtx *txm;
    entity3d *e, *it;
    list_for_each_entry(txm, &mq->txmodels, entry) {
        model3d *model = txm->model;
        if (!model->anis.da.nr_el || !model->nr_joints || model->nr_joints > 256) continue;
        struct ga_model *m = NULL;
        list_for_each_entry_iter(e, it, &txm->entities, entry) {
            if (!entity3d_matches(e, ENTITY3D_ALIVE)) continue;
            /* whoever ran the entity's own hook ran its animated_update too; what gpu_mq_update batched (default_update
             * entities, and body-less characters, whose hook ends in default_update) is posed here */
            /* (the same entity at the same place of its model's batch as last frame, and no walk of the queue since: it
             * is still batched -- no look-up; 50 000 characters: 3 of the 5 ms this loop took were the hash) */
            if (gs) {
                struct ga_model *pm = m ? m : ga_model_peek(ga, model);
                const bool known = same_walk && pm && pm->n < pm->n_prev && pm->prev[pm->n] == e;
                if (!known && !gpu_scene_entity_is_batched(gs, e)) continue;
            } else if (e->update != default_update) continue;
            if (e->animation < 0)
                animation_next(e, s);
            struct queued_animation *qa = ani_current(e);
            if (!qa) continue;
            if (!m) {
                int rc = 0;
                m = ga_model_of(ga, model, &rc);
                if (!m) return rc;
            }
            GA_CK(ga_reserve(m, m->n + 1));
            const double frame_time = (time - e->ani_time) * qa->speed;
            if (frame_time == 0.0)
                qa->sfx_state = 0;
            m->ents[m->n] = e;
            m->frame_time[m->n] = frame_time;
            m->anim_of[m->n] = qa->animation;
            m->n++;
        }
    }

    if (timing) clock_gettime(CLOCK_MONOTONIC, &ts_[1]);
    for (uint32_t k = 0; k < ga->n_models; k++) {
        struct ga_model *m = &ga->models[k];
        if (!m->n) continue;
        const uint32_t J = m->J;
        const size_t cj = (size_t)m->n * J;
        /* Up to GA_MAPPED_MAX joints the pose kernel reads its per-character inputs from, and writes T/R/S, palettes and joint
         * positions straight to, the page-locked staging arrays (their device aliases): one launch and one wait instead of
         * three or four copies up, the launch, three copies down and the wait -- 10 characters x 64 joints 0.112 -> 0.081 ms a
         * frame (the reference's host pose: 0.061), 200 characters 0.29 -> 0.21, 500 characters 0.48 -> 0.42; at 5 000
         * characters the copies' link rate wins (3.6 ms staged, 4.2 mapped). */
        const bool mapped = cj <= GA_MAPPED_MAX;
        bool same = m->n == m->n_prev && mapped == m->mapped_prev;   /* the T/R/S state lives where the last frame left it */
        m->mapped_prev = mapped;
        for (uint32_t c = 0; same && c < m->n; c++) same = m->ents[c] == m->prev[c];
        for (uint32_t c = 0; c < m->n; c++) {
            e = m->ents[c];
            m->h_anim[c] = (uint32_t)m->anim_of[c];
            m->h_ftime[c] = (float)m->frame_time[c];            /* channels_transform takes a float (model.c:1345) */
            memcpy(m->h_emx + 16 * (size_t)c, e->mx, 64);
            if (!same)                                           /* membership changed: the host joints are the state */
                for (uint32_t j = 0; j < J; j++) {
                    float *d = m->h_trs + 10 * ((size_t)c * J + j);
                    memcpy(d, e->joints[j].translation, 12);
                    memcpy(d + 3, e->joints[j].rotation, 16);
                    memcpy(d + 7, e->joints[j].scale, 12);
                }
        }
        if (mapped) {
            const clapgpu_pose_batch pb = { .n_chars = m->n, .anim = m->a_anim, .frame_time = m->a_ftime, .entity = NULL,
                                            .entity_mx = m->a_emx, .trs = m->a_trs, .joint_transforms = m->a_jt,
                                            .joint_pos = m->a_jpos };
            GA_CK(clapgpu_pose_update(NULL, &m->sk, &m->an, &pb));
            continue;
        }
        GA_CK(clapgpu_memcpy_h2d(m->d_anim, m->h_anim, (size_t)m->n * 4, NULL));
        GA_CK(clapgpu_memcpy_h2d(m->d_ftime, m->h_ftime, (size_t)m->n * 4, NULL));
        GA_CK(clapgpu_memcpy_h2d(m->d_emx, m->h_emx, (size_t)m->n * 64, NULL));
        if (!same) GA_CK(clapgpu_memcpy_h2d(m->d_trs, m->h_trs, cj * 40, NULL));
        const clapgpu_pose_batch pb = { .n_chars = m->n, .anim = m->d_anim, .frame_time = m->d_ftime, .entity = NULL,
                                        .entity_mx = m->d_emx, .trs = m->d_trs, .joint_transforms = m->d_jt,
                                        .joint_pos = m->d_jpos };
        GA_CK(clapgpu_pose_update(NULL, &m->sk, &m->an, &pb));
        GA_CK(clapgpu_memcpy_d2h(m->h_trs, m->d_trs, cj * 40, NULL));
        GA_CK(clapgpu_memcpy_d2h(m->h_jt, m->d_jt, cj * 64, NULL));
        GA_CK(clapgpu_memcpy_d2h(m->h_jpos, m->d_jpos, cj * 16, NULL));
    }
    GA_CK(clapgpu_stream_sync(NULL));
    if (timing) clock_gettime(CLOCK_MONOTONIC, &ts_[2]);

    /* Joints back into the entities.  Above GA_PAR_MIN joints the copies are split over a few worker threads (memory
     * latency on one core: 11 ns a joint, 3.6 of a 6 ms frame at 5 000 characters x 64 joints) and the per-entity
     * callbacks follow in list order; a frame callback of entity k then finds this frame's joints in the entities behind
     * it as well, where the reference (and the one-thread path below that size) still shows it last frame's. */
    size_t joints_total = 0;
    for (uint32_t k = 0; k < ga->n_models; k++) joints_total += (size_t)ga->models[k].n * ga->models[k].J;
    const int nt = joints_total >= ga_par_min() ? ga_threads() : 1;
    for (uint32_t k = 0; k < ga->n_models; k++) {
        struct ga_model *m = &ga->models[k];
        if (nt > 1 && m->n >= (uint32_t)nt)
            gpu_scene_par_for(ga_joints_back_range, m, m->n, nt);
        for (uint32_t c = 0; c < m->n; c++) {
            e = m->ents[c];
            if (!(nt > 1 && m->n >= (uint32_t)nt)) ga_joints_back(m, c, c + 1);
            /* model.c:1585-1591 */
            struct queued_animation *qa = ani_current(e);
            struct animation *an = &m->model->anis.x[m->anim_of[c]];
            const double frame_time = m->frame_time[c];
            if (qa && qa->frame_cb)
                qa->frame_cb(qa, e, s, frame_time / an->time_end);
            if (qa && an->frame_sfx)
                an->frame_sfx(qa, e, s, frame_time / an->time_end);
            if (frame_time >= an->time_end)
                animation_next(e, s);
        }
    }
    if (timing) clock_gettime(CLOCK_MONOTONIC, &ts_[3]);
    if (gs) gpu_scene_run_deferred(gs, mq);                      /* joint-attached subtrees: the palettes of this frame are in place */
    if (timing) {
        clock_gettime(CLOCK_MONOTONIC, &ts_[4]);
        double d[4];
        for (int i = 0; i < 4; i++) d[i] = (ts_[i + 1].tv_sec - ts_[i].tv_sec) * 1e3 + (ts_[i + 1].tv_nsec - ts_[i].tv_nsec) * 1e-6;
        fprintf(stderr, "gpu_anim_update: clocks %.3f  up+pose+down %.3f  joints back %.3f  deferred %.3f ms\n", d[0], d[1], d[2], d[3]);
    }
    return 0;
}
