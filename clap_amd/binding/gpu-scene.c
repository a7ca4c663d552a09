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
 *      no joint attachment, and a batched (or no) parent;
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
 * Deliberate difference: a child that precedes its parent in list order lags one frame in the
 * reference (model.c:1911-1922 walks creation order); the device computes the converged
 * parents-first result, and step 5 hands it over when the reference's own rule fires.
 */
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "gpu-scene.h"
#include "scene.h"
#include "clapgpu_scene.h"

struct gs_rec {
    entity3d    *e;             /* key; NULL = empty bucket */
    model3d     *model;
    uint32_t    handle;         /* libclapgpu_scene handle, CLAPGPU_NO_ENTITY while on the host */
    uint32_t    parent_handle;
    uint32_t    flags;
    uint32_t    gen;
    uint8_t     cls;            /* 0 unknown, 1 batched, 2 host */
    uint8_t     self_ok;
    uint8_t     xform_dirty;    /* xform.updated as seen (and cleared) in step 3 */
};

struct gs_model { model3d *model; uint32_t handle; };

struct gpu_scene {
    clapgpu_scene   *scene;
    int             (*default_hook)(entity3d *, void *);
    struct gs_rec   *tab;   uint32_t tab_cap, tab_used;
    entity3d        **order; uint32_t n_order, cap_order;     /* ALIVE entities in list order */
    struct gs_model *models; uint32_t n_models, cap_models;
    uint32_t        gen;
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

static struct gs_rec *tab_find(struct gpu_scene *gs, entity3d *e)
{
    if (!gs->tab_cap) return NULL;
    for (uint32_t i = ptr_hash(e) & (gs->tab_cap - 1);; i = (i + 1) & (gs->tab_cap - 1)) {
        if (!gs->tab[i].e) return NULL;
        if (gs->tab[i].e == e) return &gs->tab[i];
    }
}

static struct gs_rec *tab_insert_raw(struct gs_rec *tab, uint32_t cap, const struct gs_rec *r)
{
    uint32_t i = ptr_hash(r->e) & (cap - 1);
    while (tab[i].e) i = (i + 1) & (cap - 1);
    tab[i] = *r;
    return &tab[i];
}

/* Rebuild without the records of entities that are gone; also grows. */
static int tab_rebuild(struct gpu_scene *gs, uint32_t min_live, bool drop_stale)
{
    uint32_t cap = 1024;
    while (cap < 2 * min_live + 2) cap <<= 1;
    struct gs_rec *nt = calloc(cap, sizeof(*nt));
    if (!nt) return _CERR_NOMEM;
    uint32_t used = 0;
    for (uint32_t i = 0; i < gs->tab_cap; i++) {
        struct gs_rec *r = &gs->tab[i];
        if (!r->e || (drop_stale && r->gen != gs->gen)) continue;
        tab_insert_raw(nt, cap, r);
        used++;
    }
    free(gs->tab);
    gs->tab = nt; gs->tab_cap = cap; gs->tab_used = used;
    return 0;
}

static struct gs_rec *tab_get_or_add(struct gpu_scene *gs, entity3d *e, bool *is_new)
{
    struct gs_rec *r = tab_find(gs, e);
    *is_new = !r;
    if (r) return r;
    if (2 * (gs->tab_used + 1) > gs->tab_cap && tab_rebuild(gs, gs->tab_used + 1, false))
        return NULL;
    struct gs_rec nr = { .e = e, .handle = CLAPGPU_NO_ENTITY, .parent_handle = CLAPGPU_NO_ENTITY };
    gs->tab_used++;
    return tab_insert_raw(gs->tab, gs->tab_cap, &nr);
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
    *out = gs;
    return 0;
}

void gpu_scene_done(struct gpu_scene *gs)
{
    if (!gs) return;
    clapgpu_scene_destroy(gs->scene);
    free(gs->tab); free(gs->order); free(gs->models);
    free(gs);
}

const struct gpu_scene_stats *gpu_scene_last_stats(const struct gpu_scene *gs) { return &gs->stats; }

/* Criteria an entity meets on its own (step 2); the parent's class is folded in by classify(). */
static bool self_batchable(const struct gpu_scene *gs, entity3d *e)
{
    return e->update == gs->default_hook &&
           !entity_animated(e) &&
           !(e->flags & (ENTITY3D_HAS_PHYSICS | ENTITY3D_IS_CHARACTER | ENTITY3D_IS_UI | ENTITY3D_IS_PARTICLE)) &&
           e->light_idx < 0 &&
           e->parent_joint == JOINT_TYPE_MAX;
}

static uint8_t classify(struct gpu_scene *gs, struct gs_rec *r, int depth)
{
    if (r->cls) return r->cls;
    if (!r->self_ok || depth > 64) return r->cls = 2;
    if (!r->e->parent) return r->cls = 1;
    struct gs_rec *p = tab_find(gs, r->e->parent);
    if (!p || p->gen != gs->gen) return r->cls = 2;      /* parent not alive in this queue */
    return r->cls = classify(gs, p, depth + 1);
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

#define CK(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

int gpu_mq_update(struct gpu_scene *gs, struct mq *mq, struct view *view)
{
    if (!gs || !mq) return _CERR_INVALID_ARGUMENTS;
    struct gpu_scene_stats *st = &gs->stats;
    struct scene *scene = mq->priv;
    memset(st, 0, sizeof(*st));
    gs->gen++;
    gs->n_order = 0;

    /* 1: list walk (order[] keeps the entities; the table may move while it grows) */
    model3dtx *txm;
    entity3d *e, *it;
    list_for_each_entry(txm, &mq->txmodels, entry) {
        list_for_each_entry_iter(e, it, &txm->entities, entry) {
            if (!entity3d_matches(e, ENTITY3D_ALIVE)) continue;
            bool is_new;
            struct gs_rec *r = tab_get_or_add(gs, e, &is_new);
            if (!r) return _CERR_NOMEM;
            if (gs->n_order == gs->cap_order) {
                gs->cap_order = gs->cap_order ? 2 * gs->cap_order : 4096;
                gs->order = realloc(gs->order, gs->cap_order * sizeof(*gs->order));
                if (!gs->order) return _CERR_NOMEM;
            }
            r->gen = gs->gen;
            r->cls = 0;
            r->self_ok = self_batchable(gs, e);
            gs->order[gs->n_order++] = e;
        }
    }

    /* 2 + 3: classify, mirror */
    for (uint32_t k = 0; k < gs->n_order; k++) {
        e = gs->order[k];
        struct gs_rec *r = tab_find(gs, e);
        model3d *model = e->txmodel->model;

        if (classify(gs, r, 0) != 1) {
            if (r->handle != CLAPGPU_NO_ENTITY) {               /* left the batch (gained a body, a hook, ...) */
                CK(clapgpu_scene_entity_delete(gs->scene, r->handle));
                r->handle = r->parent_handle = CLAPGPU_NO_ENTITY;
                st->deleted++;
            }
            continue;
        }
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
            r->flags = ENTITY3D_ALIVE | ENTITY3D_VISIBLE;        /* what entity_new starts with */
            st->registered++;
        }
        const uint32_t flags = e->flags & (ENTITY3D_ALIVE | 0xffffu);
        if (flags != r->flags) {
            CK(clapgpu_scene_entity_flags(gs->scene, r->handle, flags & ~r->flags, r->flags & ~flags));
            r->flags = flags;
        }
        r->xform_dirty = transform_is_updated(&e->xform);
        if (r->xform_dirty || fresh) {
            const float *q = transform_rotation_quat(&e->xform);
            CK(clapgpu_scene_entity_position(gs->scene, r->handle, transform_pos(&e->xform, NULL)));
            CK(clapgpu_scene_entity_rotation(gs->scene, r->handle, q));
            CK(clapgpu_scene_entity_scale(gs->scene, r->handle, e->scale));
            st->uploaded++;
        }
    }
    /* parents after every batched entity has its handle */
    for (uint32_t k = 0; k < gs->n_order; k++) {
        e = gs->order[k];
        struct gs_rec *r = tab_find(gs, e);
        if (r->cls != 1) continue;
        const uint32_t ph = e->parent ? tab_find(gs, e->parent)->handle : CLAPGPU_NO_ENTITY;
        if (ph != r->parent_handle) {
            CK(clapgpu_scene_entity_set_parent(gs->scene, r->handle, ph));
            r->parent_handle = ph;
        }
    }
    /* entities that left the queue (entity3d_delete, model.c:1787) */
    if (gs->tab_used != gs->n_order) {
        for (uint32_t i = 0; i < gs->tab_cap; i++) {
            struct gs_rec *r = &gs->tab[i];
            if (r->e && r->gen != gs->gen && r->handle != CLAPGPU_NO_ENTITY) {
                CK(clapgpu_scene_entity_delete(gs->scene, r->handle));
                st->deleted++;
            }
        }
        CK(tab_rebuild(gs, gs->n_order, true));
    }

    /* 4: the device */
    const uint32_t slots_before = clapgpu_scene_slot_count(gs->scene);
    clapgpu_frustum fr;
    if (view) frustum_of(view, &fr);
    CK(clapgpu_scene_mq_update(gs->scene, view ? &fr : NULL));
    st->retiled = st->registered || st->deleted || slots_before != clapgpu_scene_slot_count(gs->scene);
    gs->culled_view = view;
    if (view) memcpy(gs->culled_planes, view->main.frustum_planes, sizeof(gs->culled_planes));

    /* 5: results and host hooks, list order */
    for (uint32_t k = 0; k < gs->n_order; k++) {
        e = gs->order[k];
        struct gs_rec *r = tab_find(gs, e);
        if (r->cls != 1) {
            entity3d_update(e, mq->priv);
            st->host++;
            continue;
        }
        st->batched++;
        entity3d *parent = e->parent;
        const bool rebuilt = parent ? (r->xform_dirty || e->parent_seq != parent->seq) : r->xform_dirty;
        if (rebuilt) {
            if (parent) e->parent_seq = parent->seq;             /* model.c:1613 */
            if (r->xform_dirty) transform_clear_updated(&e->xform);
            e->seq++;                                            /* model.c:1616, 1669 */
            memcpy(e->mx, clapgpu_scene_entity_mx(gs->scene, r->handle), sizeof(mat4x4));
            memcpy(e->inverse_mx, clapgpu_scene_entity_inverse_mx(gs->scene, r->handle), sizeof(mat4x4));
            if (!r->model->skip_aabb) {                          /* entity3d_aabb_update, model.c:1204-1205 */
                memcpy(e->aabb, clapgpu_scene_entity_aabb(gs->scene, r->handle), sizeof(e->aabb));
                memcpy(e->aabb_center, clapgpu_scene_entity_aabb_center(gs->scene, r->handle), sizeof(vec3));
            }
            st->written_back++;
        }
        if (scene)
            bv_pick(scene, e);
    }
    return 0;
}

bool gpu_view_entity_in_frustum(struct gpu_scene *gs, struct view *view, entity3d *e)
{
    if (gs && view == gs->culled_view &&
        !memcmp(gs->culled_planes, view->main.frustum_planes, sizeof(gs->culled_planes)) &&
        (e->flags & (ENTITY3D_ALIVE | ENTITY3D_VISIBLE | ENTITY3D_SKIP_CULLING)) == (ENTITY3D_ALIVE | ENTITY3D_VISIBLE)) {
        const struct gs_rec *r = tab_find(gs, e);
        /* the mask bit is the draw predicate ALIVE && VISIBLE && (SKIP_CULLING || in frustum):
         * for an alive, visible, culled entity it is the frustum test itself */
        if (r && r->gen == gs->gen && r->cls == 1 && r->flags == (e->flags & (ENTITY3D_ALIVE | 0xffffu)))
            return clapgpu_scene_entity_in_frustum(gs->scene, r->handle);
    }
    return view_entity_in_frustum(view, e);
}
