/*
 * gpu-scene.c -- CLAP-side binding of libclapgpu (see gpu-scene.h).  C23 like the engine,
 * compiled with the engine's flags against the engine's headers.
 *
 * Per gpu_mq_update():
 *   1. walk mq->txmodels -> txm->entities in list order (what mq_for_each_matching does,
 *      model.c:1911-1922); look every ALIVE entity up in a pointer -> record table;
 *   2. decide which entities are batched: hook == default_update, no skeleton animation
 *      (animated_update, model.c:1715-1716), no physics body (phys_body_update /
 *      phys_body_rotate_xform, model.c:1659-1687), no light (light_set_pos, model.c:1689-1694),
 *      no joint attachment, and a batched (or no) parent that comes EARLIER in the list;
 *   3. mirror creations, deletions, e->parent, e->flags and -- where xform.updated is set --
 *      position / rotation / scale into libclapgpu_scene, clearing xform.updated as
 *      default_update does (model.c:1615, 1668);
 *   4. clapgpu_scene_mq_update(): tile, upload, ONE kernel launch (update + cull), download;
 *   5. second walk in list order: a batched entity that the reference would have rebuilt this
 *      frame (root: xform.updated; child: xform.updated or parent_seq != parent->seq,
 *      model.c:1609-1616) takes mx / inverse_mx / aabb / aabb_center from the download and has
 *      seq / parent_seq advanced the same way; the camera bounding-volume pick
 *      (model.c:1697-1713) is replayed per entity; every other entity runs its own hook here, so
 *      host entities see their device parents' fresh matrices and the list order of side effects
 *      is the reference's.
 *
 * A child that precedes its parent in list order lags one frame in the reference (model.c:1911-1922
 * walks creation order).  The device computes converged, parents-first results, so such a child -- and
 * its subtree -- is left on the host, where the lag is reproduced exactly: every entity, batched or not,
 * ends the frame with the reference's bits.
 */
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>

#include "gpu-scene.h"
#include "scene.h"
#include "clapgpu_scene.h"
#include "clapgpu_snapshot.h"

#define NO_REC 0xffffffffu

struct gs_rec {
    entity3d    *e;             /* key; NULL = free record */
    model3d     *model;
    entity3d    *parent_e;      /* e->parent when parent_rec was resolved */
    uint32_t    parent_rec;
    uint32_t    next;           /* hash chain / free list */
    uint32_t    handle;         /* libclapgpu_scene handle, CLAPGPU_NO_ENTITY while on the host */
    uint32_t    slot;           /* its row in the result arrays; refreshed when the layout is rebuilt */
    uint32_t    parent_handle;
    uint32_t    flags;
    uint32_t    gen;            /* last frame this entity was met in the queue */
    uint32_t    order_pos;      /* its position in that frame's walk */
    uint8_t     cls;            /* 0 unknown, 1 batched, 2 host */
    uint8_t     self_ok;
    uint8_t     xform_dirty;    /* xform.updated as seen in step 3 (cleared in step 5, like default_update) */
};

struct gs_model { model3d *model; uint32_t handle; };

struct gpu_scene {
    clapgpu_scene   *scene;
    int             (*default_hook)(entity3d *, void *);
    /* records: dense array + chained pointer hash.  A steady queue never hashes: the k-th entity of
     * this walk is checked against the k-th record of the previous walk first. */
    struct gs_rec   *rec;   uint32_t n_rec, cap_rec, free_rec, n_live;
    uint32_t        *bucket; uint32_t n_bucket;
    uint32_t        *order, *prev_order; uint32_t n_order, n_prev, cap_order;
    clapgpu_scene_arrays res;
    struct gs_model *models; uint32_t n_models, cap_models;
    uint32_t        gen, vis_cursor;
    bool            anim_elsewhere;
    struct view     *culled_view;
    vec4            culled_planes[6];
    struct gpu_scene_stats stats;
};

static inline uint32_t ptr_hash(const void *p)
{
    uint64_t x = (uint64_t)(uintptr_t)p;
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33;
    return (uint32_t)x;
}

static uint32_t rec_find(const struct gpu_scene *gs, const entity3d *e)
{
    if (!gs->n_bucket) return NO_REC;
    for (uint32_t i = gs->bucket[ptr_hash(e) & (gs->n_bucket - 1)]; i != NO_REC; i = gs->rec[i].next)
        if (gs->rec[i].e == e) return i;
    return NO_REC;
}

static int rehash(struct gpu_scene *gs, uint32_t n_bucket)
{
    uint32_t *nb = malloc((size_t)n_bucket * sizeof(*nb));
    if (!nb) return _CERR_NOMEM;
    memset(nb, 0xff, (size_t)n_bucket * sizeof(*nb));
    for (uint32_t i = 0; i < gs->n_rec; i++) {
        struct gs_rec *r = &gs->rec[i];
        if (!r->e) continue;
        uint32_t *b = &nb[ptr_hash(r->e) & (n_bucket - 1)];
        r->next = *b;
        *b = i;
    }
    free(gs->bucket);
    gs->bucket = nb; gs->n_bucket = n_bucket;
    return 0;
}

static uint32_t rec_add(struct gpu_scene *gs, entity3d *e)
{
    uint32_t i;
    if (gs->free_rec != NO_REC) {
        i = gs->free_rec;
        gs->free_rec = gs->rec[i].next;
    } else {
        if (gs->n_rec == gs->cap_rec) {
            const uint32_t cap = gs->cap_rec ? 2 * gs->cap_rec : 4096;
            struct gs_rec *nr = realloc(gs->rec, (size_t)cap * sizeof(*nr));
            if (!nr) return NO_REC;
            gs->rec = nr; gs->cap_rec = cap;
        }
        i = gs->n_rec++;
    }
    if (gs->n_live + 1 > gs->n_bucket) {
        gs->rec[i].e = NULL;                                  /* not yet hashable */
        if (rehash(gs, gs->n_bucket ? 2 * gs->n_bucket : 8192)) return NO_REC;
    }
    gs->rec[i] = (struct gs_rec){ .e = e, .parent_rec = NO_REC, .handle = CLAPGPU_NO_ENTITY, .slot = CLAPGPU_NO_ENTITY,
                                  .parent_handle = CLAPGPU_NO_ENTITY };
    uint32_t *b = &gs->bucket[ptr_hash(e) & (gs->n_bucket - 1)];
    gs->rec[i].next = *b;
    *b = i;
    gs->n_live++;
    return i;
}

static void rec_del(struct gpu_scene *gs, uint32_t i)
{
    uint32_t *link = &gs->bucket[ptr_hash(gs->rec[i].e) & (gs->n_bucket - 1)];
    while (*link != i) link = &gs->rec[*link].next;
    *link = gs->rec[i].next;
    gs->rec[i].e = NULL;
    gs->rec[i].next = gs->free_rec;
    gs->free_rec = i;
    gs->n_live--;
}

static int model_handle(struct gpu_scene *gs, model3d *m, uint32_t *out)
{
    for (uint32_t i = 0; i < gs->n_models; i++)
        if (gs->models[i].model == m) { *out = gs->models[i].handle; return 0; }
    if (gs->n_models == gs->cap_models) {
        gs->cap_models = gs->cap_models ? 2 * gs->cap_models : 16;
        gs->models = realloc(gs->models, gs->cap_models * sizeof(*gs->models));
        if (!gs->models) return _CERR_NOMEM;
    }
    const float aabb[6] = { m->aabb[0][0], m->aabb[0][1], m->aabb[0][2], m->aabb[1][0], m->aabb[1][1], m->aabb[1][2] };
    int rc = clapgpu_scene_model_new(gs->scene, aabb, m->skip_aabb, out);
    if (rc) return rc;
    gs->models[gs->n_models++] = (struct gs_model){ m, *out };
    return 0;
}

int gpu_scene_init(struct gpu_scene **out, int device, int (*default_hook)(entity3d *, void *))
{
    if (!out || !default_hook) return _CERR_INVALID_ARGUMENTS;
    struct gpu_scene *gs = calloc(1, sizeof(*gs));
    if (!gs) return _CERR_NOMEM;
    int rc = clapgpu_scene_create(&gs->scene, device);
    if (rc) { free(gs); return rc; }
    gs->default_hook = default_hook;
    gs->free_rec = NO_REC;
    *out = gs;
    return 0;
}

void gpu_scene_done(struct gpu_scene *gs)
{
    if (!gs) return;
    clapgpu_scene_destroy(gs->scene);
    free(gs->rec); free(gs->bucket); free(gs->order); free(gs->prev_order); free(gs->models);
    free(gs);
}

const struct gpu_scene_stats *gpu_scene_last_stats(const struct gpu_scene *gs) { return &gs->stats; }

void gpu_scene_animation_elsewhere(struct gpu_scene *gs, bool elsewhere) { gs->anim_elsewhere = elsewhere; }

bool gpu_scene_entity_is_batched(struct gpu_scene *gs, entity3d *e)
{
    const uint32_t i = rec_find(gs, e);
    return i != NO_REC && gs->rec[i].gen == gs->gen && gs->rec[i].cls == 1;
}

/* Criteria an entity meets on its own (step 2); the parent's class is folded in during the walk. */
static bool self_batchable(const struct gpu_scene *gs, entity3d *e)
{
    return e->update == gs->default_hook &&
           (gs->anim_elsewhere || !entity_animated(e)) &&
           !(e->flags & (ENTITY3D_HAS_PHYSICS | ENTITY3D_IS_CHARACTER | ENTITY3D_IS_UI | ENTITY3D_IS_PARTICLE)) &&
           e->light_idx < 0 &&
           e->parent_joint == JOINT_TYPE_MAX;
}

/* The record of r's parent, or NO_REC if the parent is not an ALIVE member of this queue. */
static uint32_t parent_rec(struct gpu_scene *gs, struct gs_rec *r)
{
    entity3d *p = r->e->parent;
    if (r->parent_e != p || r->parent_rec == NO_REC || gs->rec[r->parent_rec].e != p) {   /* a miss is retried: the parent may be met later in the walk */
        r->parent_e = p;
        r->parent_rec = rec_find(gs, p);
    }
    return (r->parent_rec != NO_REC && gs->rec[r->parent_rec].gen == gs->gen) ? r->parent_rec : NO_REC;
}

static int frustum_of(const struct view *view, clapgpu_frustum *fr)
{
    memcpy(fr->planes, view->main.frustum_planes, sizeof(fr->planes));      /* view.h:16 */
    memcpy(fr->corners, view->main.frustum_corners, sizeof(fr->corners));   /* view.h:17 */
    return 0;
}

/* model.c:1697-1713, for an entity whose aabb is current */
static void bv_pick(struct scene *scene, entity3d *e)
{
    struct camera *cam = scene->camera;
    if ((aabb_point_is_inside(e->aabb, transform_pos(&cam->xform, NULL)) ||
         (scene->control && aabb_point_is_inside(e->aabb, transform_pos(&scene->control->xform, NULL)))) &&
         e != scene->control) {
        float volume = entity3d_aabb_X(e) * entity3d_aabb_Y(e) * entity3d_aabb_Z(e);

        if (!cam->bv || volume > cam->bv_volume) {
            cam->bv = e;
            cam->bv_volume = volume;
        }
    }
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

#define CK(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

static inline void prefetch_entity(const entity3d *e)
{
    /* sizeof(entity3d) is seven cache lines and both passes touch most of them; the record array
     * tells us which entity comes eight steps later without chasing the list */
    const char *p = (const char *)e;
    for (unsigned o = 0; o < sizeof(entity3d); o += 64)
        __builtin_prefetch(p + o, 1, 1);
}

/* Step 3 for one batched entity: creation, flags, transform.  The parent link follows in link_parent(). */
static int mirror_one(struct gpu_scene *gs, struct gs_rec *r)
{
    struct gpu_scene_stats *st = &gs->stats;
    entity3d *e = r->e;
    model3d *model = e->txmodel->model;

    if (r->handle != CLAPGPU_NO_ENTITY && r->model != model) {   /* same address, another entity */
        CK(clapgpu_scene_entity_delete(gs->scene, r->handle));
        r->handle = r->parent_handle = CLAPGPU_NO_ENTITY;
        st->deleted++;
    }
    const bool fresh = r->handle == CLAPGPU_NO_ENTITY;
    if (fresh) {
        uint32_t mh;
        CK(model_handle(gs, model, &mh));
        CK(clapgpu_scene_entity_new(gs->scene, mh, e, &r->handle));
        r->model = model;
        r->flags = ENTITY3D_ALIVE | ENTITY3D_VISIBLE;            /* what entity_new starts with */
        st->registered++;
    }
    const uint32_t flags = e->flags & (ENTITY3D_ALIVE | 0xffffu);
    if (flags != r->flags) {
        CK(clapgpu_scene_entity_flags(gs->scene, r->handle, flags & ~r->flags, r->flags & ~flags));
        r->flags = flags;
    }
    r->xform_dirty = transform_is_updated(&e->xform);
    if (r->xform_dirty || fresh) {
        CK(clapgpu_scene_entity_transform(gs->scene, r->handle, transform_pos(&e->xform, NULL),
                                          transform_rotation_quat(&e->xform), e->scale));
        st->uploaded++;
    }
    return 0;
}

static int link_parent(struct gpu_scene *gs, struct gs_rec *r)
{
    const uint32_t ph = r->e->parent ? gs->rec[parent_rec(gs, r)].handle : CLAPGPU_NO_ENTITY;
    if (ph != r->parent_handle) {
        CK(clapgpu_scene_entity_set_parent(gs->scene, r->handle, ph));
        r->parent_handle = ph;
    }
    return 0;
}

static int unbatch(struct gpu_scene *gs, struct gs_rec *r)        /* left the batch (gained a body, a hook, ...) */
{
    if (r->handle != CLAPGPU_NO_ENTITY) {
        CK(clapgpu_scene_entity_delete(gs->scene, r->handle));
        r->handle = r->parent_handle = CLAPGPU_NO_ENTITY;
        gs->stats.deleted++;
    }
    return 0;
}

int gpu_mq_update(struct gpu_scene *gs, struct mq *mq, struct view *view)
{
    if (!gs || !mq) return _CERR_INVALID_ARGUMENTS;
    struct gpu_scene_stats *st = &gs->stats;
    struct scene *scene = mq->priv;
    memset(st, 0, sizeof(*st));
    gs->gen++;

    const double t0 = now_ms();
    /*
     * 1-3 in ONE walk of the queue (the entity structs are far larger than the caches, so every
     * extra pass over them costs as much as the reference's whole update).  prev_order[] is last
     * frame's walk: an unchanged queue is matched without hashing, and its entities are prefetched
     * ahead of the list chase.
     */
    { uint32_t *t = gs->prev_order; gs->prev_order = gs->order; gs->order = t; }
    gs->n_prev = gs->n_order;
    gs->n_order = 0;
    uint32_t cursor = 0;
    model3dtx *txm;
    entity3d *e, *it;
    list_for_each_entry(txm, &mq->txmodels, entry) {
        list_for_each_entry_iter(e, it, &txm->entities, entry) {
            if (!entity3d_matches(e, ENTITY3D_ALIVE)) continue;
            uint32_t i;
            if (cursor < gs->n_prev && gs->rec[gs->prev_order[cursor]].e == e) {
                i = gs->prev_order[cursor++];
                if (cursor + 8 < gs->n_prev)
                    prefetch_entity(gs->rec[gs->prev_order[cursor + 8]].e);
            } else {
                i = rec_find(gs, e);
                if (i == NO_REC) {
                    i = rec_add(gs, e);
                    if (i == NO_REC) return _CERR_NOMEM;
                } else if (gs->rec[i].gen + 1 == gs->gen) {
                    cursor = gs->rec[i].order_pos + 1;            /* resynchronise after a deletion */
                }
            }
            if (gs->n_order == gs->cap_order) {
                const uint32_t cap = gs->cap_order ? 2 * gs->cap_order : 4096;
                uint32_t *o = realloc(gs->order, (size_t)cap * sizeof(*o));
                if (o) gs->order = o;
                uint32_t *po = realloc(gs->prev_order, (size_t)cap * sizeof(*po));
                if (po) gs->prev_order = po;
                if (!o || !po) return _CERR_NOMEM;
                gs->cap_order = cap;
            }
            struct gs_rec *r = &gs->rec[i];
            r->gen = gs->gen;
            r->order_pos = gs->n_order;
            gs->order[gs->n_order++] = i;
            r->self_ok = self_batchable(gs, e);
            if (!r->self_ok) {
                r->cls = 2;
            } else if (!e->parent) {
                r->cls = 1;
            } else {
                /* NO_REC unless already met in THIS walk.  A child that precedes its parent in list order sees the
                 * parent's matrix of the previous frame in the reference (model.c:1911-1922 walks creation order):
                 * it stays on the host, where that lag is reproduced exactly, and so does everything below it. */
                const uint32_t p = parent_rec(gs, r);
                r->cls = p != NO_REC ? gs->rec[p].cls : 2;
            }
            if (r->cls == 1) {
                CK(mirror_one(gs, r));
                CK(link_parent(gs, r));
            } else {
                CK(unbatch(gs, r));
            }
        }
    }
    const double t1 = now_ms();
    /* entities that left the queue (entity3d_delete, model.c:1787): met last frame, not this one */
    if (gs->n_live != gs->n_order) {
        for (uint32_t k = 0; k < gs->n_prev; k++) {
            const uint32_t i = gs->prev_order[k];
            struct gs_rec *r = &gs->rec[i];
            if (!r->e || r->gen == gs->gen) continue;
            if (r->handle != CLAPGPU_NO_ENTITY) {
                CK(clapgpu_scene_entity_delete(gs->scene, r->handle));
                st->deleted++;
            }
            rec_del(gs, i);
        }
    }

    const double t2 = now_ms();
    /* 4: the device */
    const uint32_t layout_before = clapgpu_scene_layout_generation(gs->scene);
    clapgpu_frustum fr;
    if (view) frustum_of(view, &fr);
    CK(clapgpu_scene_mq_update(gs->scene, view ? &fr : NULL));
    st->retiled = layout_before != clapgpu_scene_layout_generation(gs->scene);
    gs->culled_view = view;
    gs->vis_cursor = 0;
    if (view) memcpy(gs->culled_planes, view->main.frustum_planes, sizeof(gs->culled_planes));
    clapgpu_scene_arrays res = { 0 };
    if (clapgpu_scene_results(gs->scene, &res))                  /* an empty batch has none */
        memset(&res, 0, sizeof(res));
    gs->res = res;

    const double t3 = now_ms();
    /* 5: results and host hooks, list order */
    for (uint32_t k = 0; k < gs->n_order; k++) {
        struct gs_rec *r = &gs->rec[gs->order[k]];
        e = r->e;
        if (k + 8 < gs->n_order) {
            /* the entity eight steps ahead, and its rows of the download (DMA left them out of the caches) */
            const struct gs_rec *a = &gs->rec[gs->order[k + 8]];
            prefetch_entity(a->e);
            if (a->cls == 1 && a->slot < res.n_slots && !st->retiled) {
                __builtin_prefetch(res.mx + 16 * (size_t)a->slot, 0, 0);
                __builtin_prefetch(res.inverse_mx + 16 * (size_t)a->slot, 0, 0);
                __builtin_prefetch(res.aabb + 6 * (size_t)a->slot, 0, 0);
                __builtin_prefetch(res.aabb_center + 3 * (size_t)a->slot, 0, 0);
            }
        }
        if (r->cls != 1) {
            entity3d_update(e, mq->priv);
            st->host++;
            continue;
        }
        st->batched++;
        if (st->retiled || r->slot == CLAPGPU_NO_ENTITY)
            r->slot = clapgpu_scene_entity_slot(gs->scene, r->handle);
        entity3d *parent = e->parent;
        const bool rebuilt = parent ? (r->xform_dirty || e->parent_seq != parent->seq) : r->xform_dirty;
        if (rebuilt) {
            const size_t slot = r->slot;
            if (parent) e->parent_seq = parent->seq;             /* model.c:1613 */
            if (r->xform_dirty) transform_clear_updated(&e->xform);
            e->seq++;                                            /* model.c:1616, 1669 */
            memcpy(e->mx, res.mx + 16 * slot, sizeof(mat4x4));
            memcpy(e->inverse_mx, res.inverse_mx + 16 * slot, sizeof(mat4x4));
            if (!r->model->skip_aabb) {                          /* entity3d_aabb_update, model.c:1204-1205 */
                memcpy(e->aabb, res.aabb + 6 * slot, sizeof(e->aabb));
                memcpy(e->aabb_center, res.aabb_center + 3 * slot, sizeof(vec3));
            }
            st->written_back++;
        }
        if (scene)
            bv_pick(scene, e);
    }
    st->ms_walk = t1 - t0; st->ms_mirror = t2 - t1; st->ms_device = t3 - t2; st->ms_scatter = now_ms() - t3;
    return 0;
}

bool gpu_view_entity_in_frustum(struct gpu_scene *gs, struct view *view, entity3d *e)
{
    if (gs && view == gs->culled_view &&
        !memcmp(gs->culled_planes, view->main.frustum_planes, sizeof(gs->culled_planes)) &&
        (e->flags & (ENTITY3D_ALIVE | ENTITY3D_VISIBLE | ENTITY3D_SKIP_CULLING)) == (ENTITY3D_ALIVE | ENTITY3D_VISIBLE)) {
        /* _models_render asks in list order (model.c:958-973): try the next record of the walk first */
        uint32_t i;
        if (gs->vis_cursor < gs->n_order && gs->rec[gs->order[gs->vis_cursor]].e == e)
            i = gs->order[gs->vis_cursor];
        else
            i = rec_find(gs, e);
        const struct gs_rec *r = i != NO_REC ? &gs->rec[i] : NULL;
        if (r) gs->vis_cursor = r->order_pos + 1 < gs->n_order ? r->order_pos + 1 : 0;
        /* the mask bit is the draw predicate ALIVE && VISIBLE && (SKIP_CULLING || in frustum):
         * for an alive, visible, culled entity it is the frustum test itself */
        if (r && r->gen == gs->gen && r->cls == 1 && r->flags == (e->flags & (ENTITY3D_ALIVE | 0xffffu)))
            return (gs->res.vis_mask[r->slot >> 6] >> (r->slot & 63)) & 1;
    }
    return view_entity_in_frustum(view, e);
}

int gpu_scene_snapshot_begin(struct gpu_scene *gs, const char *path, struct clapgpu_snapshot_writer **out)
{
    if (!gs || !path || !out) return _CERR_INVALID_ARGUMENTS;
    uint32_t n = 0;
    for (uint32_t k = 0; k < gs->n_order; k++) n += gs->rec[gs->order[k]].cls == 1;
    const uint32_t nm = gs->n_models ? gs->n_models : 1;
    uint32_t *index_of = malloc((size_t)(gs->n_rec ? gs->n_rec : 1) * 4);       /* record -> row of the dump */
    float *pos_scale = calloc((size_t)(n ? n : 1) * 4, 4), *rot = calloc((size_t)(n ? n : 1) * 4, 4);
    int32_t *parent = calloc(n ? n : 1, 4), *model = calloc(n ? n : 1, 4);
    uint32_t *flags = calloc(n ? n : 1, 4), *seqs = calloc(n ? n : 1, 4);
    float *maabb = calloc((size_t)nm * 6, 4);
    uint8_t *mskip = calloc(nm, 1);
    int rc = _CERR_NOMEM;
    clapgpu_snapshot_writer *w = NULL;
    if (!index_of || !pos_scale || !rot || !parent || !model || !flags || !seqs || !maabb || !mskip) goto done;
    uint32_t row = 0;
    for (uint32_t k = 0; k < gs->n_order; k++)
        if (gs->rec[gs->order[k]].cls == 1) index_of[gs->order[k]] = row++;
    for (uint32_t k = 0; k < gs->n_order; k++) {
        struct gs_rec *r = &gs->rec[gs->order[k]];
        if (r->cls != 1) continue;
        entity3d *e = r->e;
        const uint32_t i = index_of[gs->order[k]];
        memcpy(pos_scale + 4 * (size_t)i, transform_pos(&e->xform, NULL), 12);
        pos_scale[4 * (size_t)i + 3] = e->scale;
        memcpy(rot + 4 * (size_t)i, transform_rotation_quat(&e->xform), 16);
        parent[i] = e->parent ? (int32_t)index_of[parent_rec(gs, r)] : -1;
        flags[i] = (e->flags & (ENTITY3D_ALIVE | 0xffffu)) | CLAPGPU_E_DIRTY;     /* a replay rebuilds everything */
        for (uint32_t m = 0; m < gs->n_models; m++)
            if (gs->models[m].model == r->model) model[i] = (int32_t)m;
    }
    for (uint32_t m = 0; m < gs->n_models; m++) {
        const model3d *md = gs->models[m].model;
        const float a[6] = { md->aabb[0][0], md->aabb[0][1], md->aabb[0][2], md->aabb[1][0], md->aabb[1][1], md->aabb[1][2] };
        memcpy(maabb + 6 * (size_t)m, a, 24);
        mskip[m] = md->skip_aabb;
    }
    rc = clapgpu_snapshot_create(&w, path);
    if (rc) goto done;
    const int64_t n64 = n;
#define ADD(name, dt, nd, d0, d1, ptr) do { const uint64_t dims__[2] = { d0, d1 }; \
        if ((rc = clapgpu_snapshot_add(w, name, dt, nd, dims__, ptr))) { clapgpu_snapshot_abort(w); w = NULL; goto done; } } while (0)
    ADD("entities.n", CLAPGPU_DT_I64, 1, 1, 0, &n64);
    ADD("entities.pos_scale", CLAPGPU_DT_F32, 2, n, 4, pos_scale);
    ADD("entities.rot", CLAPGPU_DT_F32, 2, n, 4, rot);
    ADD("entities.parent", CLAPGPU_DT_I32, 1, n, 0, parent);
    ADD("entities.model", CLAPGPU_DT_I32, 1, n, 0, model);
    ADD("entities.flags", CLAPGPU_DT_U32, 1, n, 0, flags);
    ADD("entities.seqs", CLAPGPU_DT_U32, 1, n, 0, seqs);
    ADD("entities.model_aabb", CLAPGPU_DT_F32, 2, nm, 6, maabb);
    ADD("entities.model_skip", CLAPGPU_DT_U8, 1, nm, 0, mskip);
    if (gs->culled_view) {
        ADD("frustum.planes", CLAPGPU_DT_F32, 2, 6, 4, gs->culled_view->main.frustum_planes);
        ADD("frustum.corners", CLAPGPU_DT_F32, 2, 8, 4, gs->culled_view->main.frustum_corners);
    }
#undef ADD
    *out = w;
    rc = 0;
done:
    free(index_of); free(pos_scale); free(rot); free(parent); free(model); free(flags); free(seqs); free(maabb); free(mskip);
    return rc;
}
