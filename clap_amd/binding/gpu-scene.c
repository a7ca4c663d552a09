/*
 * gpu-scene.c -- CLAP-side binding of libclapgpu (see gpu-scene.h).  C23 like the engine,
 * compiled with the engine's flags against the engine's headers.
 *
 * What one gpu_mq_update() has to achieve (the steps of a WALKED frame; most frames skip most of them, below):
 *   1. walk mq->txmodels -> txm->entities in list order (what mq_for_each_matching does,
 *      model.c:1911-1922); look every ALIVE entity up in a pointer -> record table;
 *   2. decide which entities are batched: hook == default_update, no skeleton animation
 *      (animated_update, model.c:1715-1716), no physics body (phys_body_update /
 *      phys_body_rotate_xform, model.c:1659-1687 -- ODE is an absent submodule of the reference, so nothing that
 *      reads a dBody can be built into the checker), no joint attachment (it rides the palette of the SAME frame,
 *      which gpu-anim.inc.c computes after this update), and a batched (or no) parent that comes EARLIER in the
 *      list.  An entity that carries a light IS batched: step 5 hands its position on (light_set_pos,
 *      model.c:1689-1694);
 *   3. mirror creations, deletions, e->parent, e->flags and -- where xform.updated is set --
 *      position / rotation / scale into libclapgpu_scene, clearing xform.updated as
 *      default_update does (model.c:1615, 1668);
 *   4. clapgpu_scene_mq_update(): ONE kernel launch (update + cull of the frame's views), results in mapped memory;
 *   5. a batched entity that the reference would have rebuilt this frame (root: xform.updated; child: xform.updated or
 *      parent_seq != parent->seq, model.c:1609-1616) takes mx / inverse_mx / aabb / aabb_center from the results and has
 *      seq / parent_seq advanced the same way; the camera bounding-volume pick (model.c:1697-1713) is replayed for the
 *      entities whose box holds a query point; every other entity runs its own hook, in list order, so host entities
 *      see their device parents' fresh matrices and the order of side effects is the reference's.
 *
 * Where to find what (in file order): records and the pointer table; mirror_one / link_parent (step 3 for one entity);
 * the draw-list arrays and the address table a walk leaves behind; notifications (gpu_scene_touch*, _entity_created /
 * _deleting: entities placed into / taken out of the standing layout); scatter_one / scatter_fetched (step 5 for one
 * entity, the GPU_SCATTER_DRAWN counters); the worker pool and gpu_scene_par_for; fast_frame (a NOTIFIED frame:
 * O(touched + rebuilt)) and frame_results (the second half of every frame: write-back by mask on the workers, hooks
 * and bounding-volume candidates merged in list order); queue_unchanged (frames WITHOUT notifications go by the
 * records); by_host_fields (after a re-tile the mask comes from the host fields); the walked frame as named parts --
 * walk_begin, walk_queue (the list chase on this thread, criteria / classes / pushes on the workers), walk_settle,
 * walk_device, second_half_serial, walk_tail -- and mq_update_frame, which picks among them; then views, verdicts,
 * LOD pick and draw list.
 *
 * A child that precedes its parent in list order lags one frame in the reference (model.c:1911-1922
 * walks creation order).  The device computes converged, parents-first results, so such a child -- and
 * its subtree -- is left on the host, where the lag is reproduced exactly: every entity, batched or not,
 * ends the frame with the reference's bits.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE                 /* qsort_r */
#endif
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include <pthread.h>
#include <unistd.h>
#include <stdio.h>

#include "gpu-scene.h"
#include "scene.h"

#ifdef CONFIG_GPU_SCENE
/* the engine's view_entity_in_frustum IS the binding then (gpu-exports.inc.c): fall back to the reference's body */
bool ref_view_entity_in_frustum(struct view *view, entity3d *e);
#define view_entity_in_frustum ref_view_entity_in_frustum
/* and so is entity3d_update: the hooks this file runs itself are the reference's dispatch, not a notification */
void ref_entity3d_update(entity3d *e, void *data);
#define entity3d_update ref_entity3d_update
/* and entity3d_set_lod: the pick this file makes for a host-class entity is the reference's own, not a notification */
void ref_entity3d_set_lod(entity3d *e, int lod, bool force);
#define entity3d_set_lod ref_entity3d_set_lod
#endif
#include "clapgpu_scene.h"
#include "clapgpu_snapshot.h"

#define NO_REC 0xffffffffu
#define CK(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

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
    uint8_t     cls;            /* 0 unknown, 1 batched, 2 host, 3 host but deferred behind the frame's pose, 4 batched in the frame's
                                   SECOND entity launch, behind the pose: subtrees riding a batched character's joint */
    uint8_t     att;            /* the mirror has this entity marked as joint-attached */
    uint8_t     self_ok;
    uint8_t     xform_dirty;    /* xform.updated as seen in step 3 (cleared in step 5, like default_update) */
    uint8_t     pending;        /* on the touched list (notification mode) */
    uint8_t     host_done;      /* entity3d_update() / entity3d_reset() ran this entity's update on the host between frames: the device
                                   still has to rebuild it (its children follow its seq), the host fields are already final */
    uint8_t     gone;           /* gpu_scene_entity_deleting() named this entity and it was not taken out in place: whatever the next walk
                                   meets at this address is ANOTHER entity (malloc hands a freed entity3d's memory to the next one) */
    uint8_t     rides, animated; /* e->parent_joint names a joint / entity_animated(e), as the last walk saw them (inputs of its class) */
    uint8_t     keep_auto;      /* a standing host reader the walk can see on the entity itself (light carrier, hook half of its own, animated, joint rider) */
    uint8_t     keep, user_keep, host_child;   /* GPU_SCATTER_DRAWN: written back whenever rebuilt (as the mirror holds it) / asked for by
                                   gpu_scene_keep() / a host-class child reads this entity's mx and seq (last walk) */
    uint32_t    lag;            /* host-class entity listed BEFORE its batched parent: index + 1 into gs->lag_*[], else 0 */
    uint64_t    order_key;      /* its place in the queue: txmodel's rank << 32 | position in that txmodel's list (order_pos is the
                                   place in order[], where entities taken in without a walk stand at the end) */
    int32_t     lod_force, lod_cur; /* e->force_lod / e->cur_lod as the mirror holds them (gpu_scene_select_lod) */
};

struct gs_model { model3d *model; uint32_t handle; unsigned int lod_min, lod_max; };

static struct gpu_scene *g_bound;     /* the scene the engine-named entry points (gpu-exports.inc.c) serve */

struct gs_wq;
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
    /* body-less characters (gpu-character.inc.c): is this entity's hook character_update over default_update, and the
     * host half of that hook, run before the entity is mirrored */
    bool            (*char_plain)(entity3d *, int (*)(entity3d *, void *));
    int             (*char_half)(entity3d *, void *);
    uint32_t        *char_list; uint32_t n_char, cap_char;         /* batched characters in list order (last walk) */
    /* notification mode: the engine's mutators report what they touch (gpu_scene_touch / gpu_scene_topology) and
     * a frame costs O(touched + rebuilt + host-class entities) instead of two walks over every entity3d */
    bool            notify, topology_pending, walked, last_fast, verify;
    /* verdict table by queue position: entity, slot, 'the mask bit is the answer' -- 13 bytes per entity read in order
     * by _models_render's loop instead of a 64-byte record and the 448-byte entity */
    entity3d        **vq_e; uint32_t *vq_slot; uint8_t *vq_ok; uint32_t cap_vq;
    bool            cull_checked, cull_ok;                         /* the culled view's planes were compared since they last changed */
    uint32_t        *touched; uint32_t n_touched, cap_touched;
    /* transform-only notifications (gpu_scene_touch_xform): the entity's address is all a mutator leaves behind -- no
     * look-up, no cache miss beside the entity it has just written; the frame's mirror pass resolves the addresses
     * through a flat table (address -> record, mirror handle) rebuilt by every walk, on all worker threads */
    entity3d        **xptr; uint32_t n_xptr, cap_xptr;
    uint64_t        *claim; uint32_t cap_claim;                    /* one bit per slot: taken by a worker of this frame's address-list pass */
    struct gs_fast { uint64_t key; uint32_t handle, slot; } *ftab; uint32_t ftab_mask, ftab_cap;
    uint32_t        *host_list; uint32_t n_host, cap_host;         /* host-class records in list order (last walk) */
    uint32_t        *deferred; uint32_t n_deferred, cap_deferred;  /* class 3 records in list order (last walk) */
    uint32_t        *att_list; uint32_t n_att, cap_att;            /* class 4 records in list order (last walk) */
    uint32_t        *att_handles; float *att_jt, *att_bind; uint32_t cap_att_roots;   /* scratch of the second launch */
    /* host-class entities whose BATCHED parent comes later in the list (last walk): the reference runs such a child
     * before its parent, i.e. against the parent's mx / seq of the PREVIOUS frame (model.c:1911-1922); a fast frame
     * writes all batched results back first, so it keeps each such parent's old mx / seq aside for the child's hook */
    uint32_t        *lag_parent; uint32_t n_lag, cap_lag;
    struct lag_keep { mat4x4 mx; uint16_t seq; } *lag_keep;
    uint64_t        *posmap; uint32_t cap_posmap;                  /* scratch: bounding-volume candidates by queue position */
    uint32_t        *slots; uint32_t cap_slots;                    /* scratch: rebuilt slots of the frame */
    uint32_t        n_batched;
    struct mq       *bound_mq; struct view *bound_view;
    void            *hook_data;                                    /* mq->priv of the running gpu_mq_update(): what the hooks get as `data` */
    struct view     *culled_view;
    vec4            culled_planes[6];
    /* the frame's other views (gpu_scene_add_view): xview[k] registered; xslot[k] = its plane among the mirror's extra views
     * in the last update (-1: it was the main view, or no view was culled), the planes that were culled, and whether a
     * verdict has compared them since */
    struct view     *xview[GPU_SCENE_EXTRA_VIEWS]; uint32_t n_xview;
    int             xslot[GPU_SCENE_EXTRA_VIEWS];
    vec4            xplanes[GPU_SCENE_EXTRA_VIEWS][6];
    bool            xchecked[GPU_SCENE_EXTRA_VIEWS], xok[GPU_SCENE_EXTRA_VIEWS];
    entity3d        **draw; int32_t *draw_lod; uint32_t n_draw, cap_draw;   /* gpu_scene_select_lod's draw list */
    uint16_t        *draw_txm;                                     /* ... and each entry's txmodel, as an index into txms[] */
    /* by device slot, laid out by every walk: the entity, its txmodel's index and the cur_lod its entity3d holds -- a pass's
     * draw list is built from these three streams without touching an entity3d (or a record) unless its LOD changed */
    entity3d        **slot_ent; uint16_t *slot_txm; int8_t *slot_lod; uint32_t cap_slot_arrays;
    const model3dtx **txms; uint32_t n_txms, cap_txms;
    /* the same list grouped by txmodel, in the order the txmodels first appear on it (gpu_scene_visible_of) */
    entity3d        **draw_g; int32_t *draw_g_lod; uint32_t cap_draw_g;
    struct gs_draw_group { const model3dtx *txm; uint32_t start, n; } *groups; uint32_t n_groups, cap_groups;
    bool            groups_valid;
    /* GPU_SCATTER_DRAWN: rebuilds of a slot the host has not been shown yet (e->seq lags by this much, uint16 like seq) */
    bool            scatter_drawn, drawn_now;                      /* the policy; it is in force for the frame being run (a fast frame) */
    bool            shown_stale;                                   /* the policy was switched on: the next walk lays shown[] out anew */
    bool            shown_live;                                    /* shown[] describes the CURRENT slots: laid out by the last walk (a walk under
                                                                      GPU_SCATTER_ALL re-tiles without it) and kept by every write-back since */
    uint16_t        *pend; uint32_t cap_pend; bool any_pend;
    /* ... and the seq each batched entity's entity3d was last GIVEN by a frame (walk, write-back or fetch; a host update in
     * between -- entity3d_update / _reset -- does not count: the device catches up with one rebuild in the next frame and
     * the children follow only then).  shown[p] + pend[p] is what a child of p copied into parent_seq when it was last
     * rebuilt on the device (model.c:1613), whatever has happened to e->parent or to p's entity3d on the host since */
    uint16_t        *shown;
    entity3d        *last_control;
    uint32_t        fetch_seen;                                    /* clapgpu_scene_arrays.fetch_serial already copied out */
    uint64_t        *walk_fetch; uint32_t cap_walk_fetch; bool walk_fetch_on;   /* rows fetched for a walk, applied as the walk meets each entity */
    /* creation / deletion without a walk (gpu_scene_entity_created / _deleting) */
    entity3d        **created; uint32_t n_created, cap_created;    /* reported since the last update, in creation order */
    uint32_t        *dead_recs; uint32_t n_dead_recs, cap_dead_recs;   /* records of entities taken out in place: tombstones in order[] until the next walk */
    struct gs_wtxm { const model3dtx *txm; uint32_t next, first; } *wtxm; uint32_t n_wtxm, cap_wtxm;   /* the queue's txmodels in list order (last walk); the next list position in each */
    bool            in_frame;                                      /* gpu_mq_update() is running (its hooks may call back into the notifications) */
    bool            replay, replaying;                             /* frames without notifications may go by the records (queue_unchanged); this frame does */
    bool            incremental, roomy;                            /* allowed; the mirror's re-tiles leave room (from the first entity that came or went between frames) */
    bool            appended;                                      /* order[] is no longer in list order: entities were taken in since the last walk */
    uint32_t        ftab_count;
    struct gs_cand { uint64_t key; uint32_t rec; } *cands; uint32_t cap_cands;
    uint32_t        inc_placed, inc_removed;
    /* a walked frame that re-tiled: what the REFERENCE would have rebuilt, decided from the host fields on the workers (by_host_fields) */
    struct gs_hf { uint32_t ppos; uint16_t seq0, pseq; uint8_t dirty, state; } *hf; uint32_t cap_hf;
    uint64_t        *hf_mask; uint32_t cap_hf_mask;
    uint32_t        *keep_changes; uint32_t cap_keep_changes;      /* scratch of a walk's last pass */
    struct gs_wq    *wq; uint32_t cap_wq;                          /* a big queue's walk: per queue position, its steps 2 and 3 on the workers */
    struct gpu_scene_stats stats;
};

static int par_threads(void);

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
    rc = clapgpu_scene_model_lods(gs->scene, *out, m->lod_min, m->lod_max);
    if (rc) return rc;
    gs->models[gs->n_models++] = (struct gs_model){ m, *out, m->lod_min, m->lod_max };
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
    clapgpu_scene_set_lod_sync(gs->scene, 1);                    /* gpu_scene_select_lod reports the LODs a pick changed itself */
    gs->free_rec = NO_REC;
    gs->verify = getenv("GPU_SCENE_VERIFY") != NULL;
    const char *sp = getenv("GPU_SCENE_SCATTER");
    gs->scatter_drawn = sp && !strcmp(sp, "drawn");
    const char *ip = getenv("GPU_SCENE_INCREMENTAL");
    gs->incremental = !(ip && !strcmp(ip, "0"));
    const char *rp = getenv("GPU_SCENE_REPLAY");
    gs->replay = !(rp && !strcmp(rp, "0"));
    gpu_scene_pool_ref();
    clapgpu_scene_set_parallel_for(gs->scene, gpu_scene_par_for, par_threads());   /* the mirror's re-tile borrows the pool */
    *out = gs;
    return 0;
}

void gpu_scene_done(struct gpu_scene *gs)
{
    if (!gs) return;
    gpu_scene_pool_unref();
    clapgpu_scene_destroy(gs->scene);
    free(gs->rec); free(gs->bucket); free(gs->order); free(gs->prev_order); free(gs->models);
    free(gs->char_list);
    free(gs->lag_parent); free(gs->lag_keep); free(gs->att_list); free(gs->att_handles); free(gs->att_jt); free(gs->att_bind);
    free(gs->draw); free(gs->draw_lod); free(gs->draw_g); free(gs->draw_g_lod); free(gs->groups); free(gs->pend); free(gs->shown); free(gs->xptr); free(gs->ftab);
    free(gs->claim); free(gs->created); free(gs->dead_recs); free(gs->wtxm); free(gs->cands); free(gs->hf); free(gs->hf_mask); free(gs->keep_changes); free(gs->wq);
    free(gs->walk_fetch); free(gs->draw_txm); free(gs->slot_ent); free(gs->slot_txm); free(gs->slot_lod); free(gs->txms);
    free(gs->touched); free(gs->host_list); free(gs->deferred); free(gs->posmap); free(gs->slots); free(gs->vq_e); free(gs->vq_slot); free(gs->vq_ok);
    if (g_bound == gs) g_bound = NULL;
    free(gs);
}

static unsigned g_device_errors;

const struct gpu_scene_stats *gpu_scene_last_stats(const struct gpu_scene *gs)
{
    ((struct gpu_scene *)gs)->stats.device_errors = g_device_errors;
    return &gs->stats;
}

void gpu_scene_device_error(const char *what, int rc)
{
    static const char *seen[8];
    g_device_errors++;
    for (unsigned k = 0; k < 8; k++) {
        if (seen[k] == what) return;                                 /* this call site has been reported */
        if (!seen[k]) { seen[k] = what; break; }
    }
    fprintf(stderr, "clap gpu binding: %s failed (%d): %s -- served by the engine's host path; further failures of this call are "
                    "only counted (gpu_scene_device_errors())\n", what, rc, clapgpu_last_error());
}

unsigned gpu_scene_device_errors(void) { return g_device_errors; }

void gpu_scene_animation_elsewhere(struct gpu_scene *gs, bool elsewhere) { gs->anim_elsewhere = elsewhere; }

void gpu_scene_characters(struct gpu_scene *gs, bool (*is_plain)(entity3d *, int (*)(entity3d *, void *)),
                          int (*host_half)(entity3d *, void *))
{
    if (!gs) return;
    gs->char_plain = is_plain; gs->char_half = host_half;
    gs->topology_pending = true;
}

/* The joint-attached subtrees this frame's gpu_mq_update() held back (class 3), in list order, now that the parents'
 * joint transforms of the frame exist.  Called by gpu_anim_update(); a frame driver without it calls this itself. */
static void bv_pick(struct scene *scene, entity3d *e);
static void scatter_one(struct gpu_scene *gs, struct gs_rec *r, const clapgpu_scene_arrays *res, size_t slot, bool parent_seq);
static void consume_fetched(struct gpu_scene *gs);
static void fetch_met_in_queue(struct gpu_scene *gs, struct mq *mq);

static int frustum_of(const struct view *view, clapgpu_frustum *fr);

/* the registered views ride the update's launch: their frusta to the mirror (those that are not the main view itself) */
static int views_before_update(struct gpu_scene *gs, struct view *view)
{
    clapgpu_frustum xfr[GPU_SCENE_EXTRA_VIEWS];
    uint32_t n = 0;
    for (uint32_t k = 0; k < gs->n_xview; k++) {
        gs->xslot[k] = -1;
        gs->xchecked[k] = gs->xok[k] = false;
        if (!view || gs->xview[k] == view) continue;
        frustum_of(gs->xview[k], &xfr[n]);
        memcpy(gs->xplanes[k], gs->xview[k]->main.frustum_planes, sizeof(gs->xplanes[k]));
        gs->xslot[k] = (int)n++;
    }
    gs->stats.views_culled = (view != NULL) + n;
    return clapgpu_scene_set_views(gs->scene, n, n ? xfr : NULL);
}

/* which registered view is `view` (and has a mask from the last update)?  -1: none */
static int xview_of(const struct gpu_scene *gs, const struct view *view)
{
    for (uint32_t k = 0; k < gs->n_xview; k++)
        if (gs->xview[k] == view) return gs->xslot[k] >= 0 ? (int)k : -1;
    return -1;
}

int gpu_scene_add_view(struct gpu_scene *gs, struct view *view)
{
    if (!gs || !view) return _CERR_INVALID_ARGUMENTS;
    for (uint32_t k = 0; k < gs->n_xview; k++) if (gs->xview[k] == view) return 0;
    if (gs->n_xview == GPU_SCENE_EXTRA_VIEWS) return _CERR_TOO_LARGE;
    gs->xview[gs->n_xview] = view;
    gs->xslot[gs->n_xview++] = -1;                               /* culled from the next update on */
    return 0;
}

void gpu_scene_remove_view(struct gpu_scene *gs, struct view *view)
{
    if (!gs) return;
    for (uint32_t k = 0; k < gs->n_xview; k++) {
        if (gs->xview[k] != view) continue;
        /* the mirror's planes keep their order until the next update: the others' slots stand */
        for (uint32_t j = k; j + 1 < gs->n_xview; j++) {
            gs->xview[j] = gs->xview[j + 1]; gs->xslot[j] = gs->xslot[j + 1];
            memcpy(gs->xplanes[j], gs->xplanes[j + 1], sizeof(gs->xplanes[j]));
            gs->xchecked[j] = gs->xchecked[j + 1]; gs->xok[j] = gs->xok[j + 1];
        }
        gs->n_xview--;
        return;
    }
}

/* The frame's second entity launch: the subtrees riding a batched character's joint (class 4), now that the palettes of
 * the frame are in the entities (e->parent->joint_transforms[e->parent_joint], model.c:1633-1640). */
static int attached_pass(struct gpu_scene *gs, struct mq *mq)
{
    struct gpu_scene_stats *st = &gs->stats;
    uint32_t n_roots = 0;
    for (uint32_t k = 0; k < gs->n_att; k++) {
        const struct gs_rec *r = &gs->rec[gs->att_list[k]];
        n_roots += r->e && r->att;
    }
    if (!n_roots) return 0;
    if (n_roots > gs->cap_att_roots) {
        uint32_t cap = gs->cap_att_roots ? gs->cap_att_roots : 64;
        while (cap < n_roots) cap *= 2;
        uint32_t *h = realloc(gs->att_handles, (size_t)cap * 4);
        if (h) gs->att_handles = h;
        float *a = realloc(gs->att_jt, (size_t)cap * 64);
        if (a) gs->att_jt = a;
        float *b = realloc(gs->att_bind, (size_t)cap * 64);
        if (b) gs->att_bind = b;
        if (!h || !a || !b) return _CERR_NOMEM;
        gs->cap_att_roots = cap;
    }
    uint32_t q = 0;
    for (uint32_t k = 0; k < gs->n_att; k++) {
        const struct gs_rec *r = &gs->rec[gs->att_list[k]];
        if (!r->e || !r->att) continue;
        entity3d *e = r->e, *parent = e->parent;
        if (!parent || !parent->joint_transforms || e->parent_joint < 0 ||
            e->parent_joint >= (int)parent->txmodel->model->nr_joints) return _CERR_INVALID_ARGUMENTS;
        gs->att_handles[q] = r->handle;
        memcpy(gs->att_jt + 16 * (size_t)q, parent->joint_transforms[e->parent_joint], 64);
        memcpy(gs->att_bind + 16 * (size_t)q, parent->txmodel->model->joints[e->parent_joint].bind, 64);
        q++;
    }
    CK(clapgpu_scene_attached_update(gs->scene, q, gs->att_handles, gs->att_jt, gs->att_bind));
    clapgpu_scene_arrays res = { 0 };
    CK(clapgpu_scene_results(gs->scene, &res));
    gs->res = res;
    struct scene *scene = mq->priv;
    for (uint32_t k = 0; k < gs->n_att; k++) {                   /* list order, parents first */
        struct gs_rec *r = &gs->rec[gs->att_list[k]];
        if (!r->e || r->slot >= res.n_slots) continue;
        if ((res.rebuilt_mask[r->slot >> 6] >> (r->slot & 63)) & 1) {
            scatter_one(gs, r, &res, r->slot, true);
            st->written_back++;
        }
        if (scene) bv_pick(scene, r->e);                         /* default_update's pick, with this frame's box (model.c:1697-1713) */
        st->attached++;
    }
    return 0;
}

void gpu_scene_run_deferred(struct gpu_scene *gs, struct mq *mq)
{
    if (!gs || !mq) return;
    if (gs->n_att) {
        const int rc = attached_pass(gs, mq);
        if (rc) {
            /* the device pass could not run: the entities' own hooks keep the frame whole, and the failure is loud */
            fprintf(stderr, "gpu_scene: joint-attached pass failed (%d, %s): running %u hooks on the host\n", rc, clapgpu_last_error(), gs->n_att);
            gs->stats.attach_failures++;
            for (uint32_t k = 0; k < gs->n_att; k++) {
                struct gs_rec *r = &gs->rec[gs->att_list[k]];
                if (r->e && entity3d_matches(r->e, ENTITY3D_ALIVE)) entity3d_update(r->e, mq->priv);
            }
        }
    }
    for (uint32_t k = 0; k < gs->n_deferred; k++) {
        struct gs_rec *r = &gs->rec[gs->deferred[k]];
        if (r->e && !r->gone && entity3d_matches(r->e, ENTITY3D_ALIVE))
            entity3d_update(r->e, mq->priv);
    }
}

/* advances with every walk of the queue: between two equal values no entity changed its class */
uint32_t gpu_scene_walk_generation(const struct gpu_scene *gs) { return gs ? gs->gen : 0; }

bool gpu_scene_entity_is_batched(struct gpu_scene *gs, entity3d *e)
{
    const uint32_t i = rec_find(gs, e);
    return i != NO_REC && gs->rec[i].gen == gs->gen && (gs->rec[i].cls == 1 || gs->rec[i].cls == 4);
}

/* Criteria an entity meets on its own (step 2); the parent's class is folded in during the walk. */
static bool self_batchable(const struct gpu_scene *gs, entity3d *e)
{
    /* light carriers are batched: scatter_one() hands the position on.  An entity riding a joint (e->parent_joint) is
     * batchable too -- in the frame's second launch, if its parent's palette is computed on the device this frame: the walk
     * decides (class 4), since that depends on the parent */
    const bool plain_char = gs->char_plain && gs->hook_data && (e->flags & ENTITY3D_IS_CHARACTER) && gs->char_plain(e, gs->default_hook);
    return (e->update == gs->default_hook || plain_char) &&
           (gs->anim_elsewhere || !entity_animated(e)) &&
           !(e->flags & (ENTITY3D_HAS_PHYSICS | (plain_char ? 0 : ENTITY3D_IS_CHARACTER) | ENTITY3D_IS_UI | ENTITY3D_IS_PARTICLE));
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


static inline void prefetch_entity(const entity3d *e)
{
    /* sizeof(entity3d) is seven cache lines and both passes touch most of them; the record array
     * tells us which entity comes eight steps later without chasing the list */
    const char *p = (const char *)e;
    if (!p) return;                                              /* a tombstone of order[] (gpu_scene_entity_deleting) */
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
        CK(clapgpu_scene_entity_new(gs->scene, mh, (void *)(uintptr_t)((uint32_t)(r - gs->rec) + 1u), &r->handle));
        r->model = model;
        r->flags = ENTITY3D_ALIVE | ENTITY3D_VISIBLE;            /* what entity_new starts with */
        r->lod_force = -1; r->lod_cur = 0;                        /* likewise (entity3d_make, model.c:1741) */
        r->keep = 0;
        st->registered++;
    }
    if (e->force_lod != r->lod_force || e->cur_lod != r->lod_cur) {   /* entity3d_set_lod since (model.c:593-609) */
        CK(clapgpu_scene_entity_lod(gs->scene, r->handle, e->force_lod, e->cur_lod));
        r->lod_force = e->force_lod; r->lod_cur = e->cur_lod;
    }
    const uint32_t flags = e->flags & (ENTITY3D_ALIVE | 0xffffu);
    if (flags != r->flags) {
        CK(clapgpu_scene_entity_flags(gs->scene, r->handle, flags & ~r->flags, r->flags & ~flags));
        r->flags = flags;
    }
    r->xform_dirty = transform_is_updated(&e->xform);
    if (r->host_done && r->xform_dirty) r->host_done = 2;        /* written again since the host updated it */
    if (gs->drawn_now && r->xform_dirty) transform_clear_updated(&e->xform);   /* its write-back may not come: default_update's clear (model.c:1615, 1668) here */
    if (r->xform_dirty || fresh || r->host_done) {       /* host_done: the flag is already cleared, the device copy is not yet current */
        CK(clapgpu_scene_entity_transform(gs->scene, r->handle, transform_pos(&e->xform, NULL),
                                          transform_rotation_quat(&e->xform), e->scale));
        st->uploaded++;
    }
    const uint8_t att = r->cls == 4 && e->parent_joint != JOINT_TYPE_MAX;
    if (fresh) r->att = 0;
    if (att != r->att) {
        CK(clapgpu_scene_entity_set_attach(gs->scene, r->handle, att));
        r->att = att;
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

static inline uint8_t verdict_ok(const struct gs_rec *r)
{
    return (r->cls == 1 || r->cls == 4) &&
           (r->flags & (ENTITY3D_ALIVE | ENTITY3D_VISIBLE | ENTITY3D_SKIP_CULLING)) == (ENTITY3D_ALIVE | ENTITY3D_VISIBLE);
}

static int push_u32(uint32_t **arr, uint32_t *n, uint32_t *cap, uint32_t v)
{
    if (*n == *cap) {
        const uint32_t c = *cap ? 2 * *cap : 1024;
        uint32_t *p = realloc(*arr, (size_t)c * sizeof(*p));
        if (!p) return _CERR_NOMEM;
        *arr = p; *cap = c;
    }
    (*arr)[(*n)++] = v;
    return 0;
}

bool gpu_scene_last_was_fast(const struct gpu_scene *gs) { return gs->last_fast; }

void gpu_scene_set_notify(struct gpu_scene *gs, bool on)
{
    gs->notify = on; gs->topology_pending = true;
    if (!on) { gs->roomy = false; clapgpu_scene_set_incremental(gs->scene, 0); }
}

/* The first entity that comes or goes between two frames says what kind of queue this is: that frame is walked and
 * re-tiled as it always was, and from that re-tile on the mirror leaves room for such edits (an eighth of every row's lanes,
 * a spare row per tile of a hierarchy).  A queue whose make-up never changes stays packed tight and pays nothing (1 M
 * entities, all moving: the room costs ~10 % of a frame). */
static void want_room(struct gpu_scene *gs)
{
    if (gs->roomy) return;
    gs->roomy = true;
    clapgpu_scene_set_incremental(gs->scene, 1);
    gs->topology_pending = true;
}

void gpu_scene_set_incremental(struct gpu_scene *gs, bool on)
{
    if (!gs) return;
    gs->incremental = on;
    if (!on && (gs->n_created || gs->n_dead_recs)) gs->topology_pending = true;
    if (!on) { gs->roomy = false; clapgpu_scene_set_incremental(gs->scene, 0); }
}
void gpu_scene_set_verify(struct gpu_scene *gs, bool on) { if (gs) gs->verify = on; }

/* verification mode: batched entities whose transform was written past the mutators; they join the touched list */
static unsigned int verify_untouched(struct gpu_scene *gs)
{
    unsigned int found = 0;
    for (uint32_t k = 0; k < gs->n_order; k++) {
        struct gs_rec *r = &gs->rec[gs->order[k]];
        if ((r->cls != 1 && r->cls != 4) || !r->e || r->pending || !transform_is_updated(&r->e->xform)) continue;
        if (found++ < 4)
            fprintf(stderr, "gpu_scene: entity %p (queue position %u) has xform.updated set but was not reported: a transform_* "
                            "write without gpu_scene_touch()\n", (void *)r->e, k);
        gpu_scene_touch(gs, r->e);
    }
    return found;
}

void gpu_scene_bind(struct gpu_scene *gs, struct mq *mq, struct view *view)
{
    g_bound = gs;
    if (!gs) return;
    if (gs->bound_mq && mq != gs->bound_mq) {
        /* the object serves another queue from now on: what GPU_SCATTER_DRAWN left on the device for the old one's entities
         * comes over first (the new queue's walk meets none of them), and the next update walks.  With a topology report
         * pending a record may name freed memory: then the rows go to the entities the OLD queue's lists still hold, as a
         * walk would hand them out (found by `clap_dropin fuzz 305`: topology report, then another queue's frame) */
        if (gs->any_pend && gpu_scene_fetch_all(gs) == _CERR_NOT_SUPPORTED) fetch_met_in_queue(gs, gs->bound_mq);
        gs->topology_pending = true;
    }
    gs->bound_mq = mq; gs->bound_view = view;
}

struct gpu_scene *gpu_scene_bound(void) { return g_bound; }
struct mq *gpu_scene_bound_mq(void) { return g_bound ? g_bound->bound_mq : NULL; }
struct view *gpu_scene_bound_view(void) { return g_bound ? g_bound->bound_view : NULL; }

/* an engine mutator (entity3d_position / _move / _rotate / _scale / _visible, model.c:1810-1842) changed e */
void gpu_scene_touch(struct gpu_scene *gs, entity3d *e)
{
    if (!gs || !gs->notify) return;
    const uint32_t i = rec_find(gs, e);
    if (i == NO_REC) return;                                         /* not ours (another queue), or new: its creator reports it */
    struct gs_rec *r = &gs->rec[i];
    if (r->pending) return;
    r->pending = 1;
    if (r->order_pos < gs->n_order && gs->vq_ok) gs->vq_ok[r->order_pos] = 0;      /* until the next update has mirrored it */
    if (push_u32(&gs->touched, &gs->n_touched, &gs->cap_touched, i)) gs->topology_pending = true;
}

/* entity3d_position / _move / _rotate / _scale changed e's transform and nothing else (transform_set_updated is set,
 * model.c:1810-1842): remembered by address.  Whoever changes e->flags, e->parent or e->update says so through
 * gpu_scene_touch() / gpu_scene_topology() as before -- this path does not look for it (the verification aid does: with
 * it on, every notification takes the checked path). */
void gpu_scene_touch_xform(struct gpu_scene *gs, entity3d *e)
{
    if (!gs || !gs->notify || gs->topology_pending) return;      /* a pending walk re-reads every transform anyway */
    if (gs->verify || !gs->ftab) { gpu_scene_touch(gs, e); return; }
    if (gs->n_xptr == gs->cap_xptr) {
        /* an entity may be written several times a frame; a list four times the queue says the game re-writes everything
         * all the time: the walk is the cheaper frame then */
        const uint64_t limit = 4ull * gs->n_order + 65536;
        if (gs->cap_xptr >= limit) { gs->topology_pending = true; return; }
        uint64_t cap = gs->cap_xptr ? 2ull * gs->cap_xptr : 4096;
        if (cap > limit) cap = limit;
        entity3d **q = realloc(gs->xptr, cap * sizeof(*q));
        if (!q) { gs->topology_pending = true; return; }
        gs->xptr = q; gs->cap_xptr = (uint32_t)cap;
    }
    gs->xptr[gs->n_xptr++] = e;
}

/* a scene has tens of txmodels: the last hit first, then a scan */
static uint32_t txm_index(struct gpu_scene *gs, const model3dtx *txm)
{
    static uint32_t last;
    if (last < gs->n_txms && gs->txms[last] == txm) return last;
    for (uint32_t g = 0; g < gs->n_txms; g++)
        if (gs->txms[g] == txm) return last = g;
    if (gs->n_txms == gs->cap_txms) {
        const uint32_t cap = gs->cap_txms ? 2 * gs->cap_txms : 32;
        const model3dtx **q = realloc(gs->txms, (size_t)cap * sizeof(*q));
        if (!q || cap > 65535) return 0xffffffffu;
        gs->txms = q; gs->cap_txms = cap;
    }
    gs->txms[gs->n_txms] = txm;
    return last = gs->n_txms++;
}

/* the tables a walk leaves behind are filled from the records on the workers (1 M entities: ~40 ms of a walked frame on one) */
#define GS_TABLES_PAR_MIN 16384u
struct walk_tables_ctx { struct gpu_scene *gs; uint32_t n_slots; int bad; uint32_t count; };
static void slot_arrays_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct walk_tables_ctx *wc = ctx;
    struct gpu_scene *gs = wc->gs;
    for (uint32_t k = lo; k < hi; k++) {
        const struct gs_rec *r = &gs->rec[gs->order[k]];
        if ((r->cls != 1 && r->cls != 4) || r->slot >= wc->n_slots) continue;
        if (r->lod_cur < -128 || r->lod_cur > 127) { __atomic_store_n(&wc->bad, 1, __ATOMIC_RELAXED); return; }
        gs->slot_ent[r->slot] = r->e;                            /* (a slot has one record) */
        gs->slot_txm[r->slot] = (uint16_t)(r->order_key >> 32);
        gs->slot_lod[r->slot] = (int8_t)r->lod_cur;
    }
}

static int slot_arrays_build(struct gpu_scene *gs)
{
    const uint32_t n = clapgpu_scene_slot_count(gs->scene);
    if (n > gs->cap_slot_arrays) {
        entity3d **a = realloc(gs->slot_ent, (size_t)n * sizeof(*a));
        if (a) gs->slot_ent = a;
        uint16_t *b = realloc(gs->slot_txm, (size_t)n * sizeof(*b));
        if (b) gs->slot_txm = b;
        int8_t *c = realloc(gs->slot_lod, n);
        if (c) gs->slot_lod = c;
        if (!a || !b || !c) { gs->cap_slot_arrays = 0; return _CERR_NOMEM; }
        gs->cap_slot_arrays = n;
    }
    if (n) memset(gs->slot_ent, 0, (size_t)n * sizeof(*gs->slot_ent));
    /* from the records alone (an entity3d is 448 bytes somewhere else): the txmodel is the walk's rank of it (order_key), the
     * LOD what mirror_one() last saw in e->cur_lod */
    gs->n_txms = 0;
    if (gs->n_wtxm > 65535) { gs->cap_slot_arrays = 0; return _CERR_NOMEM; }
    for (uint32_t t = 0; t < gs->n_wtxm; t++) {
        gs->n_txms = t;                                          /* (txm_index appends at n_txms) */
        if (gs->n_txms == gs->cap_txms) {
            const uint32_t cap = gs->cap_txms ? 2 * gs->cap_txms : 32;
            const model3dtx **q = realloc(gs->txms, (size_t)cap * sizeof(*q));
            if (!q) { gs->cap_slot_arrays = 0; return _CERR_NOMEM; }
            gs->txms = q; gs->cap_txms = cap;
        }
        gs->txms[t] = gs->wtxm[t].txm;
    }
    gs->n_txms = gs->n_wtxm;
    struct walk_tables_ctx wc = { gs, n, 0, 0 };
    gpu_scene_par_for(slot_arrays_range, &wc, gs->n_order, gs->n_order >= GS_TABLES_PAR_MIN ? par_threads() : 1);
    if (wc.bad) { gs->cap_slot_arrays = 0; return _CERR_NOMEM; }   /* a LOD outside int8: the record path then */
    return 0;
}

static inline uint32_t ftab_home(const struct gpu_scene *gs, const void *e) { return ptr_hash(e) & gs->ftab_mask; }

/* (every key is distinct and nobody looks anything up before the join: a home is claimed by its key, the rest follows) */
static void ftab_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct walk_tables_ctx *wc = ctx;
    struct gpu_scene *gs = wc->gs;
    uint32_t count = 0;
    for (uint32_t k = lo; k < hi; k++) {
        const struct gs_rec *r = &gs->rec[gs->order[k]];
        if (!r->e) continue;                                      /* taken out in place since the walk */
        count++;
        uint32_t h = ftab_home(gs, r->e);
        for (;;) {
            uint64_t none = 0;
            if (__atomic_compare_exchange_n(&gs->ftab[h].key, &none, (uint64_t)(uintptr_t)r->e, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
            h = (h + 1) & gs->ftab_mask;
        }
        gs->ftab[h].handle = (r->cls == 1 || r->cls == 4) ? r->handle : CLAPGPU_NO_ENTITY;
        gs->ftab[h].slot = r->slot;
    }
    __atomic_fetch_add(&wc->count, count, __ATOMIC_RELAXED);
}

static int ftab_build(struct gpu_scene *gs)
{
    uint32_t cap = 1024;
    while (cap < 2 * gs->n_order) cap *= 2;
    if (cap != gs->ftab_cap) {
        struct gs_fast *t = realloc(gs->ftab, (size_t)cap * sizeof(*t));
        if (!t) { free(gs->ftab); gs->ftab = NULL; gs->ftab_cap = 0; return _CERR_NOMEM; }
        gs->ftab = t; gs->ftab_cap = cap;
    }
    gs->ftab_mask = cap - 1;
    memset(gs->ftab, 0, (size_t)cap * sizeof(*gs->ftab));
    struct walk_tables_ctx wc = { gs, 0, 0, 0 };
    gpu_scene_par_for(ftab_range, &wc, gs->n_order, gs->n_order >= GS_TABLES_PAR_MIN ? par_threads() : 1);
    gs->ftab_count = wc.count;
    return 0;
}

/* an entity taken in (or out: handle CLAPGPU_NO_ENTITY -- the key stays, the probe chains run through it) without a walk */
static int ftab_set(struct gpu_scene *gs, const entity3d *e, uint32_t handle, uint32_t slot)
{
    if (!gs->ftab) return 0;
    uint32_t h = ftab_home(gs, e);
    while (gs->ftab[h].key && gs->ftab[h].key != (uint64_t)(uintptr_t)e) h = (h + 1) & gs->ftab_mask;
    if (!gs->ftab[h].key) {
        if (handle == CLAPGPU_NO_ENTITY) return 0;
        if (10ull * (gs->ftab_count + 1) > 7ull * gs->ftab_cap) return ftab_build(gs);   /* (order[] holds the new record already) */
        gs->ftab_count++;
    }
    gs->ftab[h] = (struct gs_fast){ (uint64_t)(uintptr_t)e, handle, slot };
    return 0;
}

/* the mirror pass over [lo, hi) of the address list: table line asked for sixteen entries ahead, entity eight ahead */
struct xptr_ctx { struct gpu_scene *gs; int mt, rc; uint32_t pushed; };
static void xptr_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct xptr_ctx *xc = ctx;
    struct gpu_scene *gs = xc->gs;
    uint32_t ring[8];                                            /* table positions of entries k .. k + 7 */
    uint32_t pushed = 0;
    for (uint32_t k = lo; k < hi + 8; k++) {
        if (k + 8 < hi) __builtin_prefetch(&gs->ftab[ftab_home(gs, gs->xptr[k + 8])], 0, 1);
        if (k >= lo + 8) {                                       /* entry k - 8: resolved eight steps ago, its entity asked for then */
            const uint32_t h = ring[(k - 8) & 7];
            if (h != NO_REC) {
                const struct gs_fast *f = &gs->ftab[h];
                entity3d *e = (entity3d *)(uintptr_t)f->key;
                /* an entity moved twice this frame is on the list twice: between workers, whoever claims its slot first
                 * takes it (every entry would push the same, final, transform; two workers on one entity3d race) */
                const bool taken = xc->mt && f->slot < gs->cap_claim &&
                    ((__atomic_fetch_or(&gs->claim[f->slot >> 6], 1ull << (f->slot & 63), __ATOMIC_RELAXED) >> (f->slot & 63)) & 1);
                if (!taken) {
                    const bool upd = transform_is_updated(&e->xform);
                    const int rc = xc->mt ? clapgpu_scene_entity_xform_mt(gs->scene, f->handle, transform_pos(&e->xform, NULL),
                                                                          transform_rotation_quat(&e->xform), e->scale, upd)
                                          : (upd ? clapgpu_scene_entity_transform(gs->scene, f->handle, transform_pos(&e->xform, NULL),
                                                                                  transform_rotation_quat(&e->xform), e->scale) : 0);
                    if (rc) xc->rc = rc;
                    if (gs->drawn_now && upd) transform_clear_updated(&e->xform);   /* see mirror_one */
                    pushed++;
                }
            }
        }
        if (k < hi) {
            const entity3d *e = gs->xptr[k];
            uint32_t h = ftab_home(gs, e);
            while (gs->ftab[h].key && gs->ftab[h].key != (uint64_t)(uintptr_t)e) h = (h + 1) & gs->ftab_mask;
            if (gs->ftab[h].key && gs->ftab[h].handle != CLAPGPU_NO_ENTITY) {   /* ours, and on the device */
                ring[k & 7] = h;
                __builtin_prefetch(&e->xform, 1, 1);
                __builtin_prefetch((const char *)&e->xform + 32, 1, 1);  /* (transform_t + scale may straddle a line) */
                clapgpu_scene_entity_xform_prefetch(gs->scene, gs->ftab[h].handle, gs->ftab[h].slot);
            } else
                ring[k & 7] = NO_REC;                            /* another queue's entity, or a host-class one: its own hook reads the transform */
        }
    }
    __atomic_fetch_add(&xc->pushed, pushed, __ATOMIC_RELAXED);
}

/* entity3d_update(e, data) / entity3d_reset(e) is about to run e's update on the host: under GPU_SCATTER_DRAWN the
 * reference's body reads e->parent->mx / ->seq and advances e's own counters (model.c:1609-1616), so both entities are
 * shown what the device has first; e->seq is noted so that gpu_scene_host_updated() knows how far the host moved it */
void gpu_scene_host_update_begin(struct gpu_scene *gs, entity3d *e)
{
    if (!gs || !e) return;
    if (e->parent) gpu_scene_fetch(gs, e->parent);
    gpu_scene_fetch(gs, e);
}

/* entity3d_update(e, data) / entity3d_reset(e) (model.c:1793, 1726; callers outside the frame loop: instantiate_entity
 * model.c:1872, terrain.c:551) ran e's update on the host just now: mx, inverse_mx, aabb, seq, xform.updated are final, as
 * the reference leaves them.  The device's copy of a batched entity is not: it gets the transform with the next update and
 * rebuilds the entity there (its children follow its seq counter), and the write-back skips the fields the host owns. */
void gpu_scene_host_updated(struct gpu_scene *gs, entity3d *e)
{
    if (!gs) return;
    const uint32_t i = rec_find(gs, e);
    if (i == NO_REC) return;
    struct gs_rec *r = &gs->rec[i];
    if (r->cls != 1 && r->cls != 4) return;                      /* host-class: nothing is mirrored */
    r->host_done = 1;
    /* GPU_SCATTER_DRAWN: the host counted this rebuild itself; the device's must come back to be reconciled with it */
    if (gs->scatter_drawn && !r->keep && r->handle != CLAPGPU_NO_ENTITY && !clapgpu_scene_entity_keep(gs->scene, r->handle, 1)) r->keep = 1;
    if (!gs->notify || r->pending) return;
    r->pending = 1;
    if (r->order_pos < gs->n_order && gs->vq_ok) gs->vq_ok[r->order_pos] = 0;
    if (push_u32(&gs->touched, &gs->n_touched, &gs->cap_touched, i)) gs->topology_pending = true;
}

/* entity3d_make / entity3d_delete, e->parent = ..., e->update = ..., a body / light / joint attached: the next
 * gpu_mq_update() walks the queue once */
void gpu_scene_topology(struct gpu_scene *gs) { if (gs) gs->topology_pending = true; }

/* ---- creation / deletion without a walk (gpu-scene.h) ------------------------------------------------------------------ */
void gpu_scene_entity_created(struct gpu_scene *gs, entity3d *e)
{
    if (!gs || !e) return;
    if (gs->notify && gs->incremental && gs->walked) want_room(gs);
    if (!gs->notify || !gs->incremental || !gs->walked || gs->topology_pending) { gs->topology_pending = true; return; }
    if (gs->n_created == gs->cap_created) {
        const uint32_t cap = gs->cap_created ? 2 * gs->cap_created : 64;
        entity3d **q = cap > (1u << 20) ? NULL : realloc(gs->created, (size_t)cap * sizeof(*q));   /* a level load: the walk is the cheaper frame */
        if (!q) { gs->topology_pending = true; return; }
        gs->created = q; gs->cap_created = cap;
    }
    gs->created[gs->n_created++] = e;
}

void gpu_scene_entity_deleting(struct gpu_scene *gs, entity3d *e)
{
    if (!gs || !e) return;
    if (gs->notify && gs->incremental && gs->walked) want_room(gs);
    if (!gs->notify || !gs->incremental || !gs->walked || gs->topology_pending || gs->in_frame) {
        /* the next update walks the queue.  If the allocator hands this entity3d's memory to a new entity before then, the walk
         * meets a familiar address: the record says that it is not the entity it knew (same model, xform.updated cleared by an
         * instantiate_entity-style default_update: nothing else would tell, and the device would keep the old transform) */
        const uint32_t g = rec_find(gs, e);
        if (g != NO_REC) gs->rec[g].gone = 1;
        gs->topology_pending = true;
        return;
    }
    for (uint32_t k = gs->n_created; k-- > 0;)                   /* made and gone between two frames: never seen */
        if (gs->created[k] == e) {
            memmove(gs->created + k, gs->created + k + 1, (size_t)(gs->n_created - k - 1) * sizeof(*gs->created));
            gs->n_created--;
            return;
        }
    const uint32_t i = rec_find(gs, e);
    if (i == NO_REC) return;                                     /* another queue's entity */
    struct gs_rec *r = &gs->rec[i];
    const struct scene *scene = gs->hook_data;
    if (r->cls == 2 && !r->lag && e != gs->last_control && !(scene && e == scene->control)) {
        /* a host-class entity (its own hook, or below such a one): nothing of it is on the device; it leaves the list of
         * hooks a fast frame runs -- unless another host-class entity hangs below it (the walk sorts that out) */
        uint32_t at = NO_REC;
        for (uint32_t k = 0; k < gs->n_host; k++) {
            const struct gs_rec *h = &gs->rec[gs->host_list[k]];
            if (gs->host_list[k] == i) at = k;
            else if (h->e && h->parent_e == e) { r->gone = 1; gs->topology_pending = true; return; }
        }
        if (at == NO_REC || push_u32(&gs->dead_recs, &gs->n_dead_recs, &gs->cap_dead_recs, i)) { r->gone = 1; gs->topology_pending = true; return; }
        memmove(gs->host_list + at, gs->host_list + at + 1, (size_t)(gs->n_host - at - 1) * sizeof(*gs->host_list));
        gs->n_host--;
        if (gs->vq_e && r->order_pos < gs->cap_vq) { gs->vq_e[r->order_pos] = NULL; gs->vq_ok[r->order_pos] = 0; }
        uint32_t *hl = &gs->bucket[ptr_hash(e) & (gs->n_bucket - 1)];
        while (*hl != i) hl = &gs->rec[*hl].next;
        *hl = r->next;
        r->e = NULL; r->cls = 0; r->self_ok = 0;
        r->parent_e = NULL; r->parent_rec = NO_REC;
        gs->n_live--;
        gs->inc_removed++;
        return;
    }
    /* in place: a batched leaf nobody depends on -- no batched child (the mirror knows), no host-class child reading its
     * matrix, not the control entity, no hook half of its own, no joint */
    if (r->cls != 1 || r->host_child || r->att || r->handle == CLAPGPU_NO_ENTITY || e->update != gs->default_hook ||
        e == gs->last_control || (scene && e == scene->control) ||
        push_u32(&gs->dead_recs, &gs->n_dead_recs, &gs->cap_dead_recs, i)) {
        r->gone = 1;
        gs->topology_pending = true;
        return;
    }
    if (clapgpu_scene_entity_delete_placed(gs->scene, r->handle)) {
        gs->n_dead_recs--;
        r->gone = 1;
        gs->topology_pending = true;
        return;
    }
    const uint32_t slot = r->slot;
    if (gs->pend && slot < gs->cap_pend) { gs->pend[slot] = 0; if (gs->shown) gs->shown[slot] = 0; }
    if (slot < gs->cap_slot_arrays) gs->slot_ent[slot] = NULL;
    if (gs->vq_e && r->order_pos < gs->cap_vq) { gs->vq_e[r->order_pos] = NULL; gs->vq_ok[r->order_pos] = 0; }
    ftab_set(gs, e, CLAPGPU_NO_ENTITY, 0);
    /* the record stays where order[] names it, as a tombstone, until the next walk: out of the hash, no entity, no class */
    uint32_t *link = &gs->bucket[ptr_hash(e) & (gs->n_bucket - 1)];
    while (*link != i) link = &gs->rec[*link].next;
    *link = r->next;
    r->e = NULL; r->cls = 0; r->self_ok = 0; r->keep = r->user_keep = 0;
    r->handle = r->parent_handle = CLAPGPU_NO_ENTITY;
    r->parent_e = NULL; r->parent_rec = NO_REC;
    gs->n_live--;
    if (gs->n_batched) gs->n_batched--;
    gs->inc_removed++;
}

static int ensure_order(struct gpu_scene *gs, uint32_t n)
{
    if (n > gs->cap_order) {
        uint32_t cap = gs->cap_order ? gs->cap_order : 4096;
        while (cap < n) cap *= 2;
        uint32_t *o = realloc(gs->order, (size_t)cap * sizeof(*o));
        if (o) gs->order = o;
        uint32_t *po = realloc(gs->prev_order, (size_t)cap * sizeof(*po));
        if (po) gs->prev_order = po;
        if (!o || !po) return _CERR_NOMEM;
        gs->cap_order = cap;
    }
    if (gs->cap_order > gs->cap_vq) {
        const uint32_t cap = gs->cap_order;
        entity3d **ve = realloc(gs->vq_e, (size_t)cap * sizeof(*ve));
        if (ve) gs->vq_e = ve;
        uint32_t *vs = realloc(gs->vq_slot, (size_t)cap * 4);
        if (vs) gs->vq_slot = vs;
        uint8_t *vo = realloc(gs->vq_ok, cap);
        if (vo) gs->vq_ok = vo;
        if (!ve || !vs || !vo) return _CERR_NOMEM;
        gs->cap_vq = cap;
    }
    return 0;
}

/* the per-slot state of a fast frame, for a layout that grew at its end (a growth tile) */
static int ensure_slot_state(struct gpu_scene *gs, uint32_t n_slots)
{
    if (gs->pend && gs->shown && n_slots > gs->cap_pend) {
        uint16_t *pn = realloc(gs->pend, (size_t)n_slots * sizeof(*pn));
        if (pn) gs->pend = pn;
        uint16_t *sn = realloc(gs->shown, (size_t)n_slots * sizeof(*sn));
        if (sn) gs->shown = sn;
        if (!pn || !sn) return _CERR_NOMEM;
        memset(gs->pend + gs->cap_pend, 0, (size_t)(n_slots - gs->cap_pend) * sizeof(*pn));
        memset(gs->shown + gs->cap_pend, 0, (size_t)(n_slots - gs->cap_pend) * sizeof(*sn));
        gs->cap_pend = n_slots;
    }
    if (gs->cap_slot_arrays && n_slots > gs->cap_slot_arrays) {
        const uint32_t old = gs->cap_slot_arrays;
        entity3d **a = realloc(gs->slot_ent, (size_t)n_slots * sizeof(*a));
        if (a) gs->slot_ent = a;
        uint16_t *b = realloc(gs->slot_txm, (size_t)n_slots * sizeof(*b));
        if (b) gs->slot_txm = b;
        int8_t *c = realloc(gs->slot_lod, n_slots);
        if (c) gs->slot_lod = c;
        if (!a || !b || !c) { gs->cap_slot_arrays = 0; return 0; }   /* the draw list goes through the records then */
        memset(gs->slot_ent + old, 0, (size_t)(n_slots - old) * sizeof(*a));
        gs->cap_slot_arrays = n_slots;
    }
    return 0;
}

/*
 * The entities reported by gpu_scene_entity_created() since the last update, in creation order (= list order inside a
 * txmodel: entity3d_make appends, model.c:1759): each gets a record, a place in the standing device layout and a seat at
 * the end of order[]; its transform and flags travel with this frame's touched entities.  Returns 1 when one of them has
 * to be met by a walk instead (then the frame is a walk: records made so far are found by it like any other).
 */
static int take_created(struct gpu_scene *gs, struct mq *mq)
{
    const struct scene *scene = mq->priv;
    for (uint32_t k = 0; k < gs->n_created; k++) {
        entity3d *e = gs->created[k];
        if (!entity3d_matches(e, ENTITY3D_ALIVE)) continue;      /* the walk would not meet it either */
        uint32_t rank = 0xffffffffu;
        for (uint32_t t = gs->n_wtxm; t-- > 0;)
            if (gs->wtxm[t].txm == e->txmodel) { rank = t; break; }
        if (rank == 0xffffffffu) {
            model3dtx *txm;
            bool ours = false;
            list_for_each_entry(txm, &mq->txmodels, entry) if (txm == e->txmodel) { ours = true; break; }
            if (ours) return 1;                                  /* a txmodel the last walk has not seen */
            continue;                                            /* another queue's entity */
        }
        if ((e->parent && e->parent_joint != JOINT_TYPE_MAX) || rec_find(gs, e) != NO_REC) return 1;
        /* plain: what the device can hold without anything else being set up.  Batchable in another way (a body-less
         * character, an animated entity whose pose runs elsewhere): the walk registers those.  Everything else is
         * host-class -- its own hook runs it, at its place in the list */
        const bool selfb = self_batchable(gs, e);
        const bool plain = selfb && e->update == gs->default_hook && !entity_animated(e);
        if (selfb && !plain) return 1;
        const uint64_t key = ((uint64_t)rank << 32) | gs->wtxm[rank].next;
        uint32_t pi = NO_REC;
        if (e->parent) {
            /* below a parent the last walk met, listed earlier (one listed later is read a frame late: the walk's lag
             * machinery), batched or host-class (not one of the frame's second launch) */
            pi = rec_find(gs, e->parent);
            if (pi == NO_REC || (gs->rec[pi].cls != 1 && gs->rec[pi].cls != 2) || gs->rec[pi].order_key > key ||
                (gs->rec[pi].cls == 1 && gs->rec[pi].handle == CLAPGPU_NO_ENTITY))
                return 1;
        }
        if (!plain || (pi != NO_REC && gs->rec[pi].cls == 2)) {
            /* host-class: a record, a seat in order[] and, by its place in the queue, in the list of hooks */
            CK(ensure_order(gs, gs->n_order + 1));
            if (gs->n_host == gs->cap_host) {
                if (push_u32(&gs->host_list, &gs->n_host, &gs->cap_host, 0)) return _CERR_NOMEM;
                gs->n_host--;
            }
            const uint32_t i = rec_add(gs, e);
            if (i == NO_REC) return _CERR_NOMEM;
            struct gs_rec *r = &gs->rec[i];
            gs->wtxm[rank].next++;
            r->model = e->txmodel->model;
            r->parent_e = e->parent; r->parent_rec = pi;
            r->gen = gs->gen;
            r->cls = 2; r->self_ok = selfb; r->animated = entity_animated(e);
            r->order_key = key;
            r->order_pos = gs->n_order;
            gs->order[gs->n_order++] = i;
            gs->appended = true;
            gs->vq_e[r->order_pos] = e; gs->vq_slot[r->order_pos] = CLAPGPU_NO_ENTITY; gs->vq_ok[r->order_pos] = 0;
            uint32_t at = gs->n_host;
            while (at && gs->rec[gs->host_list[at - 1]].order_key > key) at--;
            memmove(gs->host_list + at + 1, gs->host_list + at, (size_t)(gs->n_host - at) * sizeof(*gs->host_list));
            gs->host_list[at] = i;
            gs->n_host++;
            if (pi != NO_REC && gs->rec[pi].cls == 1) {          /* its hook reads that parent's mx / seq every frame: a standing reader */
                struct gs_rec *pr = &gs->rec[pi];
                pr->host_child = 1;
                if (gs->scatter_drawn && !pr->keep && !clapgpu_scene_entity_keep(gs->scene, pr->handle, 1)) {
                    pr->keep = 1;
                    gpu_scene_fetch(gs, pr->e);
                }
            }
            gs->inc_placed++;
            continue;
        }
        uint32_t mh;
        CK(model_handle(gs, e->txmodel->model, &mh));
        CK(ensure_order(gs, gs->n_order + 1));
        const uint32_t i = rec_add(gs, e);
        if (i == NO_REC) return _CERR_NOMEM;
        struct gs_rec *r = &gs->rec[i];
        uint32_t handle, slot;
        const int rc = clapgpu_scene_entity_new_placed(gs->scene, mh, (void *)(uintptr_t)(i + 1u),
                                                       pi == NO_REC ? CLAPGPU_NO_ENTITY : gs->rec[pi].handle, &handle, &slot);
        if (rc) {
            rec_del(gs, i);
            if (rc == CLAPGPU_ERR_NOT_SUPPORTED) return 1;       /* no room where it would have to go: the walk re-tiles */
            return rc;
        }
        gs->wtxm[rank].next++;
        r->model = e->txmodel->model;
        r->handle = handle; r->slot = slot;
        r->parent_e = e->parent; r->parent_rec = pi;
        r->parent_handle = pi == NO_REC ? CLAPGPU_NO_ENTITY : gs->rec[pi].handle;
        r->flags = ENTITY3D_ALIVE | ENTITY3D_VISIBLE;            /* what the mirror's entity starts with; the touched pass brings e->flags */
        r->lod_force = -1; r->lod_cur = 0;
        if (e->force_lod != -1 || e->cur_lod != 0) {
            CK(clapgpu_scene_entity_lod(gs->scene, handle, e->force_lod, e->cur_lod));
            r->lod_force = e->force_lod; r->lod_cur = e->cur_lod;
        }
        r->gen = gs->gen;
        r->cls = 1; r->self_ok = 1;
        r->order_key = key;
        r->order_pos = gs->n_order;
        gs->order[gs->n_order++] = i;
        gs->appended = true;
        CK(ensure_slot_state(gs, clapgpu_scene_slot_count(gs->scene)));
        if (gs->pend && slot < gs->cap_pend) { gs->pend[slot] = 0; gs->shown[slot] = e->seq; }
        if (slot < gs->cap_slot_arrays) {
            const uint32_t g = txm_index(gs, e->txmodel);
            if (g == 0xffffffffu || e->cur_lod < -128 || e->cur_lod > 127) gs->cap_slot_arrays = 0;
            else { gs->slot_ent[slot] = e; gs->slot_txm[slot] = (uint16_t)g; gs->slot_lod[slot] = (int8_t)e->cur_lod; }
        }
        gs->vq_e[r->order_pos] = e; gs->vq_slot[r->order_pos] = slot; gs->vq_ok[r->order_pos] = 0;   /* until the touched pass has its flags */
        CK(ftab_set(gs, e, handle, slot));
        if (gs->scatter_drawn && (e->light_idx >= 0 || (scene && e == scene->control)) &&
            !clapgpu_scene_entity_keep(gs->scene, handle, 1))
            r->keep = 1;
        if (!transform_is_updated(&e->xform)) {
            /* never positioned, or updated on the spot already (entity3d_update / _reset before its first frame): the host
             * fields are final as they are; the device builds its copy, the write-back leaves the entity3d alone unless its
             * parent moved on (gpu_scene_host_updated) */
            r->host_done = 1;
            if (gs->scatter_drawn && !r->keep && !clapgpu_scene_entity_keep(gs->scene, handle, 1)) r->keep = 1;
        }
        r->pending = 1;
        if (push_u32(&gs->touched, &gs->n_touched, &gs->cap_touched, i)) return _CERR_NOMEM;
        gs->n_batched++;
        gs->inc_placed++;
    }
    gs->n_created = 0;
    return 0;
}

/* A rebuilt entity WITHOUT a parent hands its position to the light it carries (model.c:1687-1692).  At most LIGHTS_MAX
 * entities do, each to its own slot, so this is safe from the scatter workers. */
static inline void light_hand_off(struct gpu_scene *gs, entity3d *e)
{
    if (e->parent || e->light_idx < 0 || !gs->hook_data) return;
    struct scene *scene = gs->hook_data;
    vec3 pos;
    transform_pos(&e->xform, pos);
    vec3_add(pos, pos, e->light_off);
    light_set_pos(&scene->light, e->light_idx, pos);
}

/* GPU_SCATTER_DRAWN: rebuilds of `slot` the entity3d has not been shown (0 under GPU_SCATTER_ALL) */
static inline uint16_t pend_of(const struct gpu_scene *gs, uint32_t slot)
{
    return (gs->any_pend && slot < gs->cap_pend) ? gs->pend[slot] : 0;
}

/* what r's parent's seq counter read when the DEVICE last rebuilt r's entity (model.c:1613 copies it into parent_seq): by
 * the parent's record as the last walk linked it -- not by e->parent, which the game may have cleared or the engine freed
 * since -- and without the steps a host update took since the last frame.  GPU_SCATTER_ALL: the parent's own counter. */
static inline uint16_t parent_seq_now(const struct gpu_scene *gs, const struct gs_rec *r, const entity3d *parent)
{
    /* (only while shown[] is kept: a walk under GPU_SCATTER_ALL does not lay it out, and a re-tile moves the slots under it --
     * `clap_dropin fuzz 77`: drawn, back to all, a re-tile, then a child rebuilt in a frame that is not walked.  The policy
     * alone does not say: rows left stale before a switch to GPU_SCATTER_ALL are still owed their counters -- fuzz 5016) */
    if (gs->shown_live && gs->shown && r->parent_rec != NO_REC) {
        const struct gs_rec *pr = &gs->rec[r->parent_rec];
        if ((pr->cls == 1 || pr->cls == 4) && pr->slot < gs->cap_pend)
            return (uint16_t)(gs->shown[pr->slot] + gs->pend[pr->slot]);
    }
    return parent ? parent->seq : 0;
}

static inline void seq_shown(struct gpu_scene *gs, size_t slot, uint16_t seq)
{
    if (gs->shown && slot < gs->cap_pend) gs->shown[slot] = seq;
}

static void copy_rows(struct gs_rec *r, const clapgpu_scene_arrays *res, size_t slot)
{
    entity3d *e = r->e;
    memcpy(e->mx, res->mx + 16 * slot, sizeof(mat4x4));
    memcpy(e->inverse_mx, res->inverse_mx + 16 * slot, sizeof(mat4x4));
    if (!r->model->skip_aabb) {                                  /* entity3d_aabb_update, model.c:1204-1205 */
        memcpy(e->aabb, res->aabb + 6 * slot, sizeof(e->aabb));
        memcpy(e->aabb_center, res->aabb_center + 3 * slot, sizeof(vec3));
    }
}

static void scatter_one(struct gpu_scene *gs, struct gs_rec *r, const clapgpu_scene_arrays *res, size_t slot, bool parent_seq)
{
    entity3d *e = r->e, *parent = e->parent;
    if (r->host_done) {                                          /* gpu_scene_host_updated(): the host wrote these fields itself */
        const uint8_t hd = r->host_done;                         /* 2: its transform was written again since (the mirror pass saw it) */
        r->host_done = 0;
        if (hd == 1 && !transform_is_updated(&e->xform) && !(parent && e->parent_seq != parent_seq_now(gs, r, parent))) {
            seq_shown(gs, slot, e->seq);                         /* the device has caught up with what the host did */
            return;
        }
        /* ... but it was touched again since (or its parent moved): an ordinary rebuild */
    }
    if (parent && parent_seq) e->parent_seq = parent_seq_now(gs, r, parent);   /* model.c:1613 (parents sit in lower slots: already advanced) */
    if (transform_is_updated(&e->xform)) transform_clear_updated(&e->xform);
    e->seq = (uint16_t)(e->seq + 1 + pend_of(gs, (uint32_t)slot));  /* model.c:1616, 1669 (+ the rebuilds it was not shown) */
    if (gs->any_pend && slot < gs->cap_pend) gs->pend[slot] = 0;
    seq_shown(gs, slot, e->seq);
    copy_rows(r, res, slot);
    light_hand_off(gs, e);
}

/* GPU_SCATTER_DRAWN: an entity the device rebuilt in earlier frames without telling the host, fetched now (it came into
 * view, or somebody asked): the rows, and the counters as the reference would have left them -- seq advanced once per
 * rebuild, parent_seq equal to the parent's (model.c:1613-1616: a child is rebuilt whenever its parent was). */
static void scatter_fetched(struct gpu_scene *gs, struct gs_rec *r, const clapgpu_scene_arrays *res, size_t slot)
{
    entity3d *e = r->e, *parent = e->parent;
    const uint16_t k = pend_of(gs, (uint32_t)slot);
    if (k) {
        e->seq = (uint16_t)(e->seq + k);
        gs->pend[slot] = 0;
        if (r->parent_e) e->parent_seq = parent_seq_now(gs, r, parent);   /* the parent it had when those rebuilds ran */
    }
    seq_shown(gs, slot, e->seq);
    copy_rows(r, res, slot);
}

/* the rows the mirror's last call fetched (clapgpu_scene_arrays.fetched_mask) into their entity3d */
static void consume_fetched(struct gpu_scene *gs)
{
    clapgpu_scene_arrays res;
    if (clapgpu_scene_results(gs->scene, &res)) return;
    gs->res = res;
    if (!res.n_fetched || res.fetch_serial == gs->fetch_seen) return;   /* nothing new: an earlier fetch's rows may be older than the host's by now */
    gs->fetch_seen = res.fetch_serial;
    const uint32_t words = res.n_slots / 64;
    for (uint32_t w = 0; w < words; w++) {
        uint64_t m = res.fetched_mask[w];
        while (m) {
            const uint32_t slot = w * 64 + (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            const uintptr_t u = (uintptr_t)res.slot_user[slot];
            if (!u) continue;
            struct gs_rec *r = &gs->rec[u - 1];
            if (!r->e || (r->cls != 1 && r->cls != 4)) continue;
            scatter_fetched(gs, r, &res, slot);
            gs->stats.fetched++;
        }
    }
}

/* every stale row to the entity3d the queue's own lists still hold (not by the records: some may name freed memory) */
static void fetch_met_in_queue(struct gpu_scene *gs, struct mq *mq)
{
    uint32_t n_rows = 0;
    clapgpu_scene_arrays fr;
    if (clapgpu_scene_fetch(gs->scene, NULL, &n_rows) || !n_rows || clapgpu_scene_results(gs->scene, &fr)) return;
    gs->res = fr;
    gs->fetch_seen = fr.fetch_serial;
    model3dtx *txm;
    entity3d *e, *it;
    list_for_each_entry(txm, &mq->txmodels, entry) list_for_each_entry_iter(e, it, &txm->entities, entry) {
        if (!entity3d_matches(e, ENTITY3D_ALIVE)) continue;
        const uint32_t i = rec_find(gs, e);
        if (i == NO_REC) continue;
        struct gs_rec *r = &gs->rec[i];
        if (r->gone || r->e != e || (r->cls != 1 && r->cls != 4) || r->slot >= fr.n_slots) continue;
        if (!((fr.fetched_mask[r->slot >> 6] >> (r->slot & 63)) & 1)) continue;
        scatter_fetched(gs, r, &fr, r->slot);
        gs->stats.fetched++;
    }
    if (gs->pend) memset(gs->pend, 0, (size_t)gs->cap_pend * sizeof(*gs->pend));
    gs->any_pend = false;
}

void gpu_scene_set_scatter(struct gpu_scene *gs, int policy)
{
    if (!gs) return;
    const bool drawn = policy == GPU_SCATTER_DRAWN;
    if (gs->scatter_drawn && !drawn) gpu_scene_fetch_all(gs);    /* back to "everything is always current" */
    if (drawn && !gs->scatter_drawn) { gs->topology_pending = true; gs->shown_stale = true; }   /* its per-slot counters are laid out by a walk: the next frame is one */
    gs->scatter_drawn = drawn;
}

static inline bool slot_is_stale(const struct gpu_scene *gs, uint32_t slot)
{
    return gs->res.n_stale_words && gs->res.stale_mask && slot < gs->res.n_slots && ((gs->res.stale_mask[slot >> 6] >> (slot & 63)) & 1);
}

bool gpu_scene_entity_is_stale(struct gpu_scene *gs, entity3d *e)
{
    if (!gs || !gs->any_pend) return false;
    const uint32_t i = rec_find(gs, e);
    if (i == NO_REC || (gs->rec[i].cls != 1 && gs->rec[i].cls != 4)) return false;
    clapgpu_scene_arrays res;
    if (clapgpu_scene_results(gs->scene, &res)) return false;
    gs->res = res;
    return slot_is_stale(gs, gs->rec[i].slot);
}

int gpu_scene_fetch(struct gpu_scene *gs, entity3d *e)
{
    if (!gs || !e) return _CERR_INVALID_ARGUMENTS;
    if (!gs->any_pend) return 0;
    const uint32_t i = rec_find(gs, e);
    if (i == NO_REC) return 0;
    struct gs_rec *r = &gs->rec[i];
    if ((r->cls != 1 && r->cls != 4) || r->handle == CLAPGPU_NO_ENTITY) return 0;
    CK(clapgpu_scene_fetch_entity(gs->scene, r->handle));
    consume_fetched(gs);
    return 0;
}

int gpu_scene_fetch_all(struct gpu_scene *gs)
{
    if (!gs) return _CERR_INVALID_ARGUMENTS;
    if (!gs->any_pend) return 0;
    /* entities were created or DELETED since the last update (gpu_scene_topology): a record may name freed memory, and only
     * the walk of the next gpu_mq_update() finds out which -- it fetches everything itself, as it meets the entities */
    if (gs->topology_pending) return _CERR_NOT_SUPPORTED;
    uint32_t n = 0;
    CK(clapgpu_scene_fetch(gs->scene, NULL, &n));
    consume_fetched(gs);
    gs->any_pend = false;                                        /* every counter was consumed with its row */
    if (gs->verify)                                              /* the aid's own check: nothing is owed after a full fetch */
        for (uint32_t i = 0; i < gs->cap_pend; i++)
            if (gs->pend[i]) { fprintf(stderr, "gpu_scene: slot %u still owes %u seq steps after gpu_scene_fetch_all\n", i, gs->pend[i]); gs->pend[i] = 0; }
    return 0;
}

/* one line about e's record, for a checker's mismatch report */
void gpu_scene_describe(struct gpu_scene *gs, entity3d *e, char *buf, size_t len)
{
    const uint32_t i = gs ? rec_find(gs, e) : NO_REC;
    if (i == NO_REC) { snprintf(buf, len, "no record"); return; }
    const struct gs_rec *r = &gs->rec[i];
    clapgpu_scene_arrays res;
    const bool have = !clapgpu_scene_results(gs->scene, &res);
    if (have) gs->res = res;
    int n = snprintf(buf, len, "class %u slot %u keep %u user_keep %u host_child %u host_done %u pend %u stale %d parent_rec %d order_pos %u",
             r->cls, r->slot, r->keep, r->user_keep, r->host_child, r->host_done, pend_of(gs, r->slot),
             have ? (int)slot_is_stale(gs, r->slot) : -1, r->parent_rec == NO_REC ? -1 : (int)r->parent_rec, r->order_pos);
    if (have && n > 0 && (size_t)n < len && r->slot < res.n_slots && (r->cls == 1 || r->cls == 4)) {   /* the row the mirror holds, beside the entity3d's */
        const float *b = res.aabb + 6 * (size_t)r->slot;
        snprintf(buf + n, len - (size_t)n, "; mirror box %.9g %.9g %.9g %.9g %.9g %.9g vis %d rebuilt %d, entity3d box %.9g %.9g %.9g %.9g %.9g %.9g flags %x/%x",
                 b[0], b[1], b[2], b[3], b[4], b[5], (int)((res.vis_mask[r->slot >> 6] >> (r->slot & 63)) & 1),
                 (int)((res.rebuilt_mask[r->slot >> 6] >> (r->slot & 63)) & 1),
                 ((const float *)e->aabb)[0], ((const float *)e->aabb)[1], ((const float *)e->aabb)[2], ((const float *)e->aabb)[3],
                 ((const float *)e->aabb)[4], ((const float *)e->aabb)[5], (unsigned)r->flags, (unsigned)(e->flags & (ENTITY3D_ALIVE | 0xffffu)));
    }
}

void gpu_scene_keep(struct gpu_scene *gs, entity3d *e, bool keep)
{
    if (!gs || !e) return;
    const uint32_t i = rec_find(gs, e);
    if (i == NO_REC) return;
    struct gs_rec *r = &gs->rec[i];
    r->user_keep = keep;
    if (keep && !r->keep && r->handle != CLAPGPU_NO_ENTITY && !clapgpu_scene_entity_keep(gs->scene, r->handle, 1)) {
        r->keep = 1;
        gpu_scene_fetch(gs, e);                                  /* from now on it is always current: starting now */
    }
}

/*
 * Frames that touch or rebuild hundreds of thousands of entities: the two passes over the 448-byte entity3d structs
 * are memory latency on one core, so they are split over a few worker threads (the engine's frame is single-threaded;
 * the binding may use workers as long as every call is synchronous, SURVEY 8b "Threading").
 */
#define GS_PAR_MIN 65536u
/* A frame without notifications goes by the records only where that is done on the workers: on one thread the two passes it
 * takes (queue check, mirror pass, both through records in list order over entities that lie in creation order) LOSE to
 * the plain list walk -- 20 k entities 0.95 vs 0.78 ms, 64 k 6.1 vs 3.9 --, split over the workers they win from ~16 k
 * entities on (two wake-ups of the pool, ~0.1 ms, against a walk of 0.35 ms and up). */
#define GS_REPLAY_MIN 16384u
#define GS_REPLAY_MIN_DEFAULT 16384u
/* the smallest queue whose frames without notifications go by the records (GPU_SCENE_REPLAY_MIN; below GS_REPLAY_MIN the
 * check and the passes run on the calling thread) */
static uint32_t replay_min(void)
{
    static uint32_t cached = 0xffffffffu;
    if (cached == 0xffffffffu) {
        const char *env = getenv("GPU_SCENE_REPLAY_MIN");
        cached = env ? (uint32_t)strtoul(env, NULL, 0) : GS_REPLAY_MIN_DEFAULT;
    }
    return cached;
}
/* rebuilt rows from which the write-back is split over the workers (a row is ~60 ns on one thread -- a 448-byte entity3d
 * and its 164 bytes of results, both cold --, a wake-up of the pool ~0.05 ms): 70 k entities, 13 k rebuilt: 0.87 ms serial */
#define GS_SCATTER_PAR_MIN 12288u
#define GS_SCATTER_SPARSE 8              /* ... off the mask words when GS_SCATTER_SPARSE * rebuilt <= entities in the queue, else in list order */
/* touched entities (reported one by one, or by address) from which the mirror pass is split over the workers */
#define GS_MIRROR_PAR_MIN 16384u
static inline void prefetch_entity(const entity3d *e);

struct par_job {
    struct gpu_scene *gs;
    const clapgpu_scene_arrays *res;
    const uint64_t *scat;       /* which slots' rows came back this frame */
    uint32_t lo, hi;            /* range of touched[] or order[] */
    int phase;
    uint32_t count;             /* out: uploaded / written back */
    int need_walk, rc;
    uint32_t *deferred; uint32_t n_deferred, cap_deferred;       /* children whose parent lies in an earlier chunk */
    uint32_t *whole; uint32_t n_whole, cap_whole;                /* entities updated on the host since the last frame (host_done): left out */
    void (*range_fn)(void *, uint32_t, uint32_t); void *ctx;     /* gpu_scene_par_for */
    uint32_t *cursor; uint32_t total, grain;                     /* ... its ranges handed out piece by piece (see there) */
};

#define GS_MAX_THREADS 32
static int par_threads(void)
{
    static int cached;
    if (!cached) {
        long n = sysconf(_SC_NPROCESSORS_ONLN);
        const char *env = getenv("GPU_SCENE_THREADS");           /* the passes are memory latency: they scale with the cores until DRAM says no */
        if (env && atoi(env) > 0) n = atoi(env);
        else if (n > 24) n = 24;                                 /* measured on a 128-core host: 8 -> 16 -> 24 threads 26 -> 15 -> 12 ms, 32: 14 */
        if (n > GS_MAX_THREADS) n = GS_MAX_THREADS;
        cached = n < 1 ? 1 : (int)n;
    }
    return cached;
}

/*
 * The workers are kept: created with the first frame that wants them, parked on a condition variable between passes,
 * joined by gpu_scene_done().  Created per pass (round 2) every frame of a million entities paid for fourteen thread
 * creations with cold stacks (1 M all moving: walk 18 -> 15 ms, write-back 23 -> 15 ms with the workers kept).  Waking a
 * parked worker still costs tens to hundreds of microseconds (the core has to leave its idle state), so the passes are
 * split only from GS_PAR_MIN entities up: at 10 000 entities a split pass measured four times SLOWER than one thread.
 * One pool per process: the passes of one frame follow each other, and every call of the binding is synchronous on the
 * engine's one thread.  Every binding object that may split a pass (a gpu_scene, gpu_anim, gpu_particles) holds a
 * reference (gpu_scene_pool_ref / _unref); the last one to go joins the workers.
 *
 * A pass is identified by its generation.  A worker serves exactly the generations that began after it was created:
 * it starts with `seen` = the generation current at its creation (threads are created under the pool's mutex, so no
 * pass can begin in between), and `pending` is set, under the same mutex, to the number of workers alive when the
 * generation is raised -- a thread created later never decrements a count it was not part of, and never sees the
 * function or the (stack-allocated) job array of a pass that has returned.
 */
static struct {
    pthread_t th[GS_MAX_THREADS - 1];
    int n;                                                       /* workers running */
    pthread_mutex_t mu;
    pthread_cond_t work;
    void *(*fn)(void *);
    struct par_job *jobs;
    int nt;                                                      /* jobs of the current pass (job 0 is the caller's) */
    unsigned gen;
    int pending;                                                 /* workers still busy with the current pass */
    int users;                                                   /* binding objects holding the pool */
    bool quit;
} g_pool = { .mu = PTHREAD_MUTEX_INITIALIZER, .work = PTHREAD_COND_INITIALIZER };

struct pool_arg { int me; unsigned seen; };

static void *pool_worker(void *arg)
{
    const struct pool_arg pa = *(struct pool_arg *)arg;           /* serves job me + 1 */
    free(arg);
    const int me = pa.me;
    unsigned seen = pa.seen;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (g_pool.gen == seen && !g_pool.quit) pthread_cond_wait(&g_pool.work, &g_pool.mu);
        if (g_pool.quit) break;
        seen = g_pool.gen;
        void *(*fn)(void *) = g_pool.fn;
        struct par_job *job = me + 1 < g_pool.nt ? &g_pool.jobs[me + 1] : NULL;
        pthread_mutex_unlock(&g_pool.mu);
        if (job) fn(job);
        __atomic_fetch_sub(&g_pool.pending, 1, __ATOMIC_RELEASE);
        pthread_mutex_lock(&g_pool.mu);
    }
    pthread_mutex_unlock(&g_pool.mu);
    return NULL;
}

/* called with the pool's mutex held */
static void pool_grow(int workers)
{
    while (g_pool.n < workers && g_pool.n < GS_MAX_THREADS - 1) {
        struct pool_arg *pa = malloc(sizeof(*pa));
        if (!pa) break;
        *pa = (struct pool_arg){ .me = g_pool.n, .seen = g_pool.gen };
        if (pthread_create(&g_pool.th[g_pool.n], NULL, pool_worker, pa)) { free(pa); break; }
        g_pool.n++;
    }
}

static void pool_stop(void)
{
    pthread_mutex_lock(&g_pool.mu);
    const int n = g_pool.n;
    g_pool.quit = true;
    pthread_cond_broadcast(&g_pool.work);
    pthread_mutex_unlock(&g_pool.mu);
    for (int t = 0; t < n; t++) pthread_join(g_pool.th[t], NULL);
    pthread_mutex_lock(&g_pool.mu);
    g_pool.n = 0;
    g_pool.quit = false;
    g_pool.fn = NULL; g_pool.jobs = NULL; g_pool.nt = 0;          /* nothing of a finished pass survives the workers */
    g_pool.pending = 0;
    pthread_mutex_unlock(&g_pool.mu);
}

void gpu_scene_pool_ref(void)
{
    pthread_mutex_lock(&g_pool.mu);
    g_pool.users++;
    pthread_mutex_unlock(&g_pool.mu);
}

void gpu_scene_pool_unref(void)
{
    pthread_mutex_lock(&g_pool.mu);
    const bool last = g_pool.users > 0 && --g_pool.users == 0;
    pthread_mutex_unlock(&g_pool.mu);
    if (last) pool_stop();                                       /* every call of the binding is on the engine's one thread: no pass is running */
}

static void par_run(void *(*fn)(void *), struct par_job *jobs, int nt)
{
    pthread_mutex_lock(&g_pool.mu);
    pool_grow(nt - 1);
    const int workers = g_pool.n;                                /* fewer than asked for if thread creation failed */
    if (workers > 0) {
        g_pool.fn = fn; g_pool.jobs = jobs; g_pool.nt = nt < workers + 1 ? nt : workers + 1;
        __atomic_store_n(&g_pool.pending, workers, __ATOMIC_RELAXED);     /* exactly the workers that will see this generation */
        g_pool.gen++;
        pthread_cond_broadcast(&g_pool.work);
    }
    pthread_mutex_unlock(&g_pool.mu);
    fn(&jobs[0]);
    for (int t = workers + 1; t < nt; t++) fn(&jobs[t]);        /* jobs no worker exists for */
    while (__atomic_load_n(&g_pool.pending, __ATOMIC_ACQUIRE) > 0)   /* the caller has nothing else to do: spin */
        __builtin_ia32_pause();
}

static void *par_range(void *arg)
{
    struct par_job *j = arg;
    if (!j->cursor) { j->range_fn(j->ctx, j->lo, j->hi); return NULL; }
    for (;;) {                                                   /* the next piece nobody has taken yet */
        const uint32_t k = __atomic_fetch_add(j->cursor, j->grain, __ATOMIC_RELAXED);
        if (k >= j->total) break;
        j->range_fn(j->ctx, k, j->total - k < j->grain ? j->total : k + j->grain);
    }
    return NULL;
}

/* fn(ctx, lo, hi) over a partition of [0, n) on the binding's workers and the caller.  The ranges are handed out piece by
 * piece from a shared cursor (about eight pieces a thread), not cut into one range per thread: the hosts this runs on are
 * shared, a worker that loses its core for a millisecond would otherwise hold the whole pass for it (measured: the same
 * pass 2x slower on a busy box than on a quiet one with one range a thread).  Nothing may depend on the cut: every range
 * function here writes what its indices own. */
void gpu_scene_par_for(void (*fn)(void *, uint32_t, uint32_t), void *ctx, uint32_t n, int threads)
{
    if (threads > par_threads()) threads = par_threads();
    if (threads < 2 || n < (uint32_t)threads) { fn(ctx, 0, n); return; }
    static int pieces = -1;
    if (pieces < 0) { const char *e = getenv("GPU_SCENE_PAR_PIECES"); pieces = e ? atoi(e) : 8; }   /* tuning knob: 0 = one range a thread */
    struct par_job jobs[GS_MAX_THREADS] = { 0 };
    uint32_t cursor = 0;
    uint32_t grain = pieces > 0 ? n / ((uint32_t)threads * (uint32_t)pieces) : 0;
    if (grain && grain < 64) grain = 64;
    for (int t = 0; t < threads; t++)
        jobs[t] = (struct par_job){ .lo = (uint32_t)((uint64_t)n * t / threads), .hi = (uint32_t)((uint64_t)n * (t + 1) / threads),
                                    .range_fn = fn, .ctx = ctx, .cursor = grain ? &cursor : NULL, .total = n, .grain = grain };
    par_run(par_range, jobs, threads);
}

static bool self_batchable(const struct gpu_scene *gs, entity3d *e);

/* What a walk would decide an entity's class from, against what the last walk saw: its own criteria (hook, flags, animation:
 * self_ok; a plain entity that is host-class only because of where its parent stands in the list -- cls 2, self_ok 1 -- may be
 * touched without forcing a walk), its parent, whether it rides a joint, its model; a batched one must still be on the device */
static inline bool class_inputs_changed(const struct gpu_scene *gs, const struct gs_rec *r, entity3d *e)
{
    return !entity3d_matches(e, ENTITY3D_ALIVE) || self_batchable(gs, e) != (bool)r->self_ok || e->parent != r->parent_e ||
           (e->parent && e->parent_joint != JOINT_TYPE_MAX) != (bool)r->rides || entity_animated(e) != (bool)r->animated ||
           ((r->cls == 1 || r->cls == 4) && (r->model != e->txmodel->model || r->handle == CLAPGPU_NO_ENTITY));
}

static void *par_mirror(void *arg)
{
    struct par_job *j = arg;
    struct gpu_scene *gs = j->gs;
    for (uint32_t k = j->lo; k < j->hi; k++) {
        struct gs_rec *r = &gs->rec[gs->touched[k]];
        if (k + 8 < j->hi) {
            const struct gs_rec *a = &gs->rec[gs->touched[k + 8]];
            if (a->e) { __builtin_prefetch(&a->e->xform, 0, 1); __builtin_prefetch(&a->e->flags, 0, 1); }
        }
        r->pending = 0;
        r->xform_dirty = 0;
        if (!r->e) continue;
        entity3d *e = r->e;
        if (class_inputs_changed(gs, r, e)) {
            j->need_walk = 1;
            continue;
        }
        if (r->cls != 1 && r->cls != 4) continue;
        if (e->force_lod != r->lod_force || e->cur_lod != r->lod_cur)   /* entity3d_set_lod since (model.c:593-609): after the join, on one thread */
            if (push_u32(&j->deferred, &j->n_deferred, &j->cap_deferred, gs->touched[k])) j->rc = _CERR_NOMEM;
        const uint32_t flags = e->flags & (ENTITY3D_ALIVE | 0xffffu);
        const bool same_flags = flags == r->flags;
        r->flags = flags;
        if (gs->vq_ok && r->order_pos < gs->n_order) gs->vq_ok[r->order_pos] = verdict_ok(r);
        r->xform_dirty = transform_is_updated(&e->xform);
        if (r->host_done && r->xform_dirty) r->host_done = 2;
        if (!r->xform_dirty && !r->host_done && same_flags) continue;   /* (a frame that looks at EVERY record: most have nothing to say) */
        if (gs->drawn_now && r->xform_dirty) transform_clear_updated(&e->xform);
        const int rc = clapgpu_scene_entity_transform_mt(gs->scene, r->handle, transform_pos(&e->xform, NULL),
                                                         transform_rotation_quat(&e->xform), e->scale, flags, r->xform_dirty || r->host_done);
        if (rc) j->rc = rc;
        j->count++;
    }
    return NULL;
}

static void scatter_one(struct gpu_scene *gs, struct gs_rec *r, const clapgpu_scene_arrays *res, size_t slot, bool parent_seq);

/* A list-order chunk of the rebuilt entities.  A batched entity's parent precedes it in the list, so inside a chunk
 * parent_seq can be taken at once; a child whose parent lies in an EARLIER chunk (another thread) is noted and
 * finished after the join. */
static void *par_scatter(void *arg)
{
    struct par_job *j = arg;
    struct gpu_scene *gs = j->gs;
    const clapgpu_scene_arrays *res = j->res;
    for (uint32_t k = j->lo; k < j->hi; k++) {
        struct gs_rec *r = &gs->rec[gs->order[k]];
        if (k + 8 < j->hi) {
            const struct gs_rec *a = &gs->rec[gs->order[k + 8]];
            if (a->cls == 1 && a->slot < res->n_slots && ((j->scat[a->slot >> 6] >> (a->slot & 63)) & 1)) {
                prefetch_entity(a->e);
                __builtin_prefetch(res->mx + 16 * (size_t)a->slot, 0, 0);
                __builtin_prefetch(res->inverse_mx + 16 * (size_t)a->slot, 0, 0);
                __builtin_prefetch(res->aabb + 6 * (size_t)a->slot, 0, 0);
            }
        }
        if (r->cls != 1 || r->slot >= res->n_slots || !((j->scat[r->slot >> 6] >> (r->slot & 63)) & 1)) continue;
        if (r->host_done) {                                      /* see frame_results: after the join, on one thread */
            if (push_u32(&j->whole, &j->n_whole, &j->cap_whole, gs->order[k])) j->rc = _CERR_NOMEM;
            continue;
        }
        bool here = true;
        if (r->e->parent) {
            const uint32_t pr = r->parent_rec;
            here = pr != NO_REC && gs->rec[pr].e == r->e->parent && gs->rec[pr].order_pos >= j->lo && !gs->rec[pr].host_done;
            if (!here && push_u32(&j->deferred, &j->n_deferred, &j->cap_deferred, gs->order[k])) j->rc = _CERR_NOMEM;
        }
        scatter_one(gs, r, res, r->slot, here);
        j->count++;
    }
    return NULL;
}

/* The same over a range of MASK WORDS (slots in ascending order: parents first): for a rebuilt set that is large enough for the
 * workers but a small part of the queue, where a pass over every record to find it costs more than the rows themselves
 * (1 M entities, 46 k rows to write: 80 MB of records read for 7 MB of rows). */
static void *par_scatter_mask(void *arg)
{
    struct par_job *j = arg;
    struct gpu_scene *gs = j->gs;
    const clapgpu_scene_arrays *res = j->res;
    for (uint32_t w = j->lo; w < j->hi; w++) {
        uint64_t m = j->scat[w];
        if (w + 1 < j->hi && j->scat[w + 1]) {                   /* the next word's first entity: its record's line */
            const uint32_t ns = (w + 1) * 64 + (uint32_t)__builtin_ctzll(j->scat[w + 1]);
            const uintptr_t nu = (uintptr_t)res->slot_user[ns];
            if (nu) __builtin_prefetch(&gs->rec[nu - 1], 0, 1);
        }
        while (m) {
            const uint32_t slot = w * 64 + (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            const uintptr_t u = (uintptr_t)res->slot_user[slot];
            if (!u) continue;
            struct gs_rec *r = &gs->rec[u - 1];
            if (r->cls != 1 || !r->e) continue;                  /* class 4: after the pose, from the second launch */
            if (r->host_done) {
                if (push_u32(&j->whole, &j->n_whole, &j->cap_whole, (uint32_t)(u - 1))) j->rc = _CERR_NOMEM;
                continue;
            }
            bool here = true;
            if (r->e->parent) {
                const uint32_t pr = r->parent_rec;
                here = pr != NO_REC && gs->rec[pr].e == r->e->parent && gs->rec[pr].slot >= j->lo * 64u && gs->rec[pr].slot < slot &&
                       !gs->rec[pr].host_done;
                if (!here && push_u32(&j->deferred, &j->n_deferred, &j->cap_deferred, (uint32_t)(u - 1))) j->rc = _CERR_NOMEM;
            }
            scatter_one(gs, r, res, slot, here);
            j->count++;
        }
    }
    return NULL;
}

static int rec_slot_cmp(const void *a, const void *b, void *ctx)
{
    const struct gpu_scene *gs = ctx;
    const uint32_t x = gs->rec[*(const uint32_t *)a].slot, y = gs->rec[*(const uint32_t *)b].slot;
    return x < y ? -1 : x > y;
}

static void *par_deferred(void *arg)
{
    struct par_job *j = arg;
    struct gpu_scene *gs = j->gs;
    for (uint32_t d = 0; d < j->n_deferred; d++) {
        if (d + 8 < j->n_deferred) {
            const entity3d *a = gs->rec[j->deferred[d + 8]].e;
            __builtin_prefetch(&a->parent_seq, 1, 1);
            __builtin_prefetch(&a->parent->seq, 0, 1);
        }
        const struct gs_rec *cr = &gs->rec[j->deferred[d]];
        entity3d *c = cr->e;
        c->parent_seq = parent_seq_now(gs, cr, c->parent);       /* model.c:1613: every parent is final by now */
    }
    return NULL;
}

/* GPU_SCATTER_DRAWN, after a fast frame's launch: every slot the device rebuilt without writing it back */
struct pend_ctx { struct gpu_scene *gs; const clapgpu_scene_arrays *res; uint32_t left; };
static void pend_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct pend_ctx *pc = ctx;
    struct gpu_scene *gs = pc->gs;
    const clapgpu_scene_arrays *res = pc->res;
    uint32_t left = 0;
    for (uint32_t w = lo; w < hi; w++) {
        uint64_t m = res->rebuilt_mask[w] & ~res->exported_mask[w];
        left += (uint32_t)__builtin_popcountll(m);
        while (m) {
            const uint32_t slot = w * 64 + (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            gs->pend[slot]++;
            if (gs->verify) {                                    /* a read nobody announced must show: poison what went stale */
                const uintptr_t u = (uintptr_t)res->slot_user[slot];
                if (u && gs->rec[u - 1].e) gs->rec[u - 1].e->mx[0][0] = __builtin_nanf("");
            }
        }
    }
    __atomic_fetch_add(&pc->left, left, __ATOMIC_RELAXED);
}

/* a host-class entity's own hook in a fast frame */
static void host_hook(struct gpu_scene *gs, struct mq *mq, struct gs_rec *hr)
{
    if (hr->gone || !hr->e) return;                              /* deleted by a hook that ran earlier in this frame */
    if (hr->lag) {
        /* listed before its batched parent: the reference has not updated that parent yet when this hook runs */
        struct lag_keep *kp = &gs->lag_keep[hr->lag - 1], now;
        entity3d *p = gs->rec[gs->lag_parent[hr->lag - 1]].e;
        memcpy(now.mx, p->mx, sizeof(mat4x4)); now.seq = p->seq;
        memcpy(p->mx, kp->mx, sizeof(mat4x4)); p->seq = kp->seq;
        entity3d_update(hr->e, mq->priv);
        memcpy(p->mx, now.mx, sizeof(mat4x4)); p->seq = now.seq;
    } else
        entity3d_update(hr->e, mq->priv);
}

static int cand_cmp(const void *a, const void *b)
{
    const struct gs_cand *x = a, *y = b;
    return x->key < y->key ? -1 : x->key > y->key;
}

/*
 * One frame in notification mode, nothing re-parented or re-hooked since the last walk (entities made or deleted since are
 * taken in / out in place where that is possible, gpu_scene_entity_created / _deleting):
 *   touched batched entities -> flags + transform to the mirror; the device; the slots the kernel reports as rebuilt
 *   -> back into their entity3d (ascending slot = parents first); host-class entities' own hooks and the camera
 *   bounding-volume pick of the few entities whose box contains a query point, merged in list order.
 * Returns 1 if the frame has to be done by the full walk after all (a touched entity changed class or parent).
 */
static int frame_results(struct gpu_scene *gs, struct mq *mq, const clapgpu_scene_arrays *resp, double t0, double t1, double t2);

static int fast_frame(struct gpu_scene *gs, struct mq *mq, struct view *view)
{
    struct gpu_scene_stats *st = &gs->stats;
    struct scene *scene = mq->priv;
    const double t0 = now_ms();
    if (gs->n_created) {                                         /* entities made since the last frame: into the standing layout, or a walk */
        const int rc = take_created(gs, mq);
        if (rc) return rc;
    }
    st->placed = gs->inc_placed; st->removed = gs->inc_removed;
    st->registered += gs->inc_placed; st->deleted += gs->inc_removed;
    gs->inc_placed = gs->inc_removed = 0;
    clapgpu_scene_set_export(gs->scene, gs->scatter_drawn && gs->notify ? CLAPGPU_SCENE_EXPORT_DRAWN : CLAPGPU_SCENE_EXPORT_ALL);   /* (the policy lives on notifications: gpu-scene.h) */
    gs->drawn_now = clapgpu_scene_export_is_drawn(gs->scene);
    if (gs->drawn_now && scene && scene->control != gs->last_control) {
        /* the control entity is read every frame (camera target, camera.c:191-205; the bounding-volume pick): a standing reader */
        gs->last_control = scene->control;
        if (scene->control) gpu_scene_keep(gs, scene->control, true);
    }
    /* batched characters: the host half of character_update (limbo teleport, motion reset), which may touch them */
    for (uint32_t k = 0; k < gs->n_char; k++) {
        struct gs_rec *r = &gs->rec[gs->char_list[k]];
        if (r->e && entity3d_matches(r->e, ENTITY3D_ALIVE)) gs->char_half(r->e, mq->priv);
    }
    static uint32_t mirror_par_min;
    if (!mirror_par_min) {
        const char *mp = getenv("GPU_SCENE_MIRROR_PAR_MIN");     /* tuning knob */
        mirror_par_min = mp && atoi(mp) > 0 ? (uint32_t)atoi(mp) : GS_MIRROR_PAR_MIN;
    }
    if (gs->n_touched >= mirror_par_min || (gs->replaying && par_threads() > 1)) {
        struct par_job jobs[GS_MAX_THREADS] = { 0 };
        const int nt = par_threads();
        for (int t = 0; t < nt; t++)
            jobs[t] = (struct par_job){ .gs = gs, .lo = (uint32_t)((uint64_t)gs->n_touched * t / nt),
                                        .hi = (uint32_t)((uint64_t)gs->n_touched * (t + 1) / nt) };
        par_run(par_mirror, jobs, nt);
        int need_walk = 0, prc = 0;
        for (int t = 0; t < nt; t++) {
            need_walk |= jobs[t].need_walk; st->uploaded += jobs[t].count;
            if (jobs[t].rc) prc = jobs[t].rc;
            for (uint32_t d = 0; d < jobs[t].n_deferred && !prc; d++) {          /* LODs set since: the mirror's copy follows */
                struct gs_rec *r = &gs->rec[jobs[t].deferred[d]];
                if (!r->e || r->handle == CLAPGPU_NO_ENTITY) continue;
                prc = clapgpu_scene_entity_lod(gs->scene, r->handle, r->e->force_lod, r->e->cur_lod);
                r->lod_force = r->e->force_lod; r->lod_cur = r->e->cur_lod;
                if (r->slot < gs->cap_slot_arrays) {
                    if (r->lod_cur >= -128 && r->lod_cur <= 127) gs->slot_lod[r->slot] = (int8_t)r->lod_cur;
                    else gs->cap_slot_arrays = 0;
                }
            }
            free(jobs[t].deferred); jobs[t].deferred = NULL; jobs[t].n_deferred = jobs[t].cap_deferred = 0;
        }
        if (prc) return prc;
        clapgpu_scene_mark_all_dirty(gs->scene);
        if (need_walk) {
            if (gs->drawn_now)                                   /* the walk decides by xform.updated: give back what this pass cleared */
                for (uint32_t k = 0; k < gs->n_touched; k++) {
                    struct gs_rec *r = &gs->rec[gs->touched[k]];
                    if (r->e && r->xform_dirty) transform_set_updated(&r->e->xform);
                }
            gs->n_touched = 0;
            return 1;
        }
    } else
    for (uint32_t k = 0; k < gs->n_touched; k++) {
        struct gs_rec *r = &gs->rec[gs->touched[k]];
        r->pending = 0;
        r->xform_dirty = 0;
        if (!r->e) continue;
        entity3d *e = r->e;
        if (class_inputs_changed(gs, r, e)) {
            if (gs->drawn_now)                                   /* the walk decides by xform.updated: give back what this pass cleared */
                for (uint32_t j = 0; j < k; j++) {
                    struct gs_rec *q = &gs->rec[gs->touched[j]];
                    if (q->e && q->xform_dirty) transform_set_updated(&q->e->xform);
                }
            for (k++; k < gs->n_touched; k++) gs->rec[gs->touched[k]].pending = 0;
            gs->n_touched = 0;
            return 1;
        }
        if (r->cls == 1 || r->cls == 4) CK(mirror_one(gs, r));
        if (gs->vq_ok && r->order_pos < gs->n_order) gs->vq_ok[r->order_pos] = verdict_ok(r);
    }
    gs->n_touched = 0;
    if (gs->n_xptr) {
        struct xptr_ctx xc = { .gs = gs, .mt = gs->n_xptr >= mirror_par_min && par_threads() > 1 };
        if (xc.mt) {
            const uint32_t need = clapgpu_scene_slot_count(gs->scene);
            if (need > gs->cap_claim) {
                uint64_t *q = realloc(gs->claim, ((size_t)need / 64 + 1) * 8);
                if (!q) return _CERR_NOMEM;
                gs->claim = q; gs->cap_claim = need;
            }
            memset(gs->claim, 0, ((size_t)gs->cap_claim / 64 + 1) * 8);
            gpu_scene_par_for(xptr_range, &xc, gs->n_xptr, par_threads());
            clapgpu_scene_mark_all_dirty(gs->scene);
        } else
            xptr_range(&xc, 0, gs->n_xptr);
        gs->n_xptr = 0;
        if (xc.rc) return xc.rc;
        st->uploaded += xc.pushed;
    }
    if (scene && scene->camera)
        clapgpu_scene_set_bv_points(gs->scene, transform_pos(&scene->camera->xform, NULL),
                                    scene->control ? transform_pos(&scene->control->xform, NULL) : NULL, CLAPGPU_NO_ENTITY);
    else
        clapgpu_scene_set_bv_points(gs->scene, NULL, NULL, CLAPGPU_NO_ENTITY);
    const double t1 = now_ms();
    clapgpu_frustum fr;
    if (view) frustum_of(view, &fr);
    CK(views_before_update(gs, view));
    CK(clapgpu_scene_mq_update(gs->scene, view ? &fr : NULL));
    gs->culled_view = view;
    gs->vis_cursor = 0;
    if (view) memcpy(gs->culled_planes, view->main.frustum_planes, sizeof(gs->culled_planes));
    gs->cull_checked = false;
    clapgpu_scene_arrays res = { 0 };
    if (clapgpu_scene_results(gs->scene, &res)) memset(&res, 0, sizeof(res));
    gs->res = res;
    return frame_results(gs, mq, &res, t0, t1, now_ms());
}

/*
 * The second half of a frame whose device step did not re-tile: what the kernel rebuilt goes back into the entity3d structs
 * (by the device's masks, on the workers when there is much of it), then the host-class entities' own hooks and the camera
 * bounding-volume pick, merged in list order.  A notified frame ends here, and so does a WALKED one whose layout stood (a
 * frame without notifications, or one that only had to look at the queue again): the device rebuilds exactly what the
 * reference's own tests would (model.c:1609-1616, 1667: xform.updated, or a parent that was rebuilt), so its mask is the
 * walk's answer too -- instead of a second serial pass over every entity3d (1 M entities: 41-53 ms of a walked frame).
 */
static int frame_results(struct gpu_scene *gs, struct mq *mq, const clapgpu_scene_arrays *resp, double t0, double t1, double t2)
{
    struct gpu_scene_stats *st = &gs->stats;
    struct scene *scene = mq->priv;
    const clapgpu_scene_arrays res = *resp;
    for (uint32_t k = 0; k < gs->n_lag; k++) {                  /* last frame's bits of the parents some host child still has to see */
        const entity3d *p = gs->rec[gs->lag_parent[k]].e;
        memcpy(gs->lag_keep[k].mx, p->mx, sizeof(mat4x4));
        gs->lag_keep[k].seq = p->seq;
    }

    /* results: only what the kernel rebuilt.  Few of them: straight off the mask, in slot order (parents first), each
     * entity and its rows prefetched a few steps ahead.  Many: in LIST order -- the entity3d structs lie in memory in
     * creation order, a slot-order pass over most of them would miss the caches on every one. */
    const uint32_t words = res.n_slots / 64;
    /* GPU_SCATTER_DRAWN: the rows that came back are the ones somebody reads (exported_mask); a slot rebuilt without
     * coming back is owed one more seq step when its entity3d is next written */
    const uint64_t *scat = res.exported_mask ? res.exported_mask : res.rebuilt_mask;
    if (gs->drawn_now && res.rebuilt_mask && scat != res.rebuilt_mask) {
        if (res.n_slots > gs->cap_pend || !gs->shown) return _CERR_INVALID_ARGUMENTS;   /* laid out by the walk that made this layout */
        struct pend_ctx pc = { gs, &res };
        gpu_scene_par_for(pend_range, &pc, words, words >= 2048 ? par_threads() : 1);
        st->left_stale = pc.left;
        if (pc.left) gs->any_pend = true;
    }
    const bool timing = getenv("GPU_SCENE_TIMING") != NULL;
    const double ts0 = timing ? now_ms() : 0;
    uint64_t n_rebuilt = 0;
    if (scat)
        for (uint32_t w = 0; w < words; w++) n_rebuilt += (uint64_t)__builtin_popcountll(scat[w]);
    const double ts1 = timing ? now_ms() : 0;
    static uint64_t scatter_par_min;
    if (!scatter_par_min) {
        const char *sp = getenv("GPU_SCENE_SCATTER_PAR_MIN");    /* tuning knob */
        scatter_par_min = sp && atoll(sp) > 0 ? (uint64_t)atoll(sp) : GS_SCATTER_PAR_MIN;
    }
    static int by_mask = -1;
    if (by_mask < 0) { const char *bm = getenv("GPU_SCENE_SCATTER_BY_MASK"); by_mask = bm ? atoi(bm) : GS_SCATTER_SPARSE; }   /* tuning knob: 0 = never, k = when k * rebuilt <= queue */
    if (n_rebuilt >= scatter_par_min && par_threads() > 1) {
        const int nt = par_threads();
        struct par_job jobs[GS_MAX_THREADS] = { 0 };
        /* most of the queue: in LIST order (the entity3d structs lie in creation order); a small part of it: off the mask */
        const bool sparse = by_mask > 0 && (uint64_t)by_mask * n_rebuilt <= gs->n_order;
        const uint32_t span = sparse ? words : gs->n_order;
        for (int t = 0; t < nt; t++)
            jobs[t] = (struct par_job){ .gs = gs, .res = &res, .scat = scat, .lo = (uint32_t)((uint64_t)span * t / nt),
                                        .hi = (uint32_t)((uint64_t)span * (t + 1) / nt) };
        par_run(sparse ? par_scatter_mask : par_scatter, jobs, nt);
        /* Entities updated on the host since the last frame (entity3d_update / _reset, instantiate_entity: few).  For them
         * scatter_one DECIDES by the parent's counters -- did the parent move on since, or has the device merely caught up? --
         * and that must not be read while another worker is half-way through writing them (found on the GPU box: the sum read
         * between the two stores said "not moved", and a rebuild was dropped).  So the workers leave them out (and mark their
         * children for the parent_seq pass below); here, with every other entity final, they follow on this thread, parents
         * first (ascending slot); then the children's parent_seq, which reads final counters only. */
        int rc = 0;
        uint32_t n_whole = 0;
        for (int t = 0; t < nt; t++) { if (jobs[t].rc) rc = jobs[t].rc; n_whole += jobs[t].n_whole; }
        if (n_whole && !rc) {
            uint32_t *all = malloc((size_t)n_whole * sizeof(*all)), at = 0;
            if (!all) rc = _CERR_NOMEM;
            for (int t = 0; t < nt && all; t++) {
                if (jobs[t].n_whole) memcpy(all + at, jobs[t].whole, (size_t)jobs[t].n_whole * sizeof(*all));
                at += jobs[t].n_whole;
            }
            if (all) {
                qsort_r(all, n_whole, sizeof(*all), rec_slot_cmp, gs);
                for (uint32_t k = 0; k < n_whole; k++) {
                    struct gs_rec *r = &gs->rec[all[k]];
                    scatter_one(gs, r, &res, r->slot, true);
                    st->written_back++;
                }
                free(all);
            }
        }
        if (!rc) par_run(par_deferred, jobs, nt);
        for (int t = 0; t < nt; t++) {
            st->written_back += jobs[t].count;
            free(jobs[t].deferred); free(jobs[t].whole);
            if (jobs[t].rc) rc = jobs[t].rc;
        }
        if (rc) return rc;
    } else if (4 * n_rebuilt > gs->n_order) {
        /* a batched entity's parent precedes it in the list (else it would be host-class): one pass, parents first */
        for (uint32_t k = 0; k < gs->n_order; k++) {
            struct gs_rec *r = &gs->rec[gs->order[k]];
            if (k + 8 < gs->n_order) {
                const struct gs_rec *a = &gs->rec[gs->order[k + 8]];
                if (a->cls == 1 && a->slot < res.n_slots && ((scat[a->slot >> 6] >> (a->slot & 63)) & 1)) {
                    prefetch_entity(a->e);
                    __builtin_prefetch(res.mx + 16 * (size_t)a->slot, 0, 0);
                    __builtin_prefetch(res.inverse_mx + 16 * (size_t)a->slot, 0, 0);
                    __builtin_prefetch(res.aabb + 6 * (size_t)a->slot, 0, 0);
                }
            }
            if (r->cls != 1 || r->slot >= res.n_slots || !((scat[r->slot >> 6] >> (r->slot & 63)) & 1)) continue;
            scatter_one(gs, r, &res, r->slot, true);
            st->written_back++;
        }
    } else {
        /* the rebuilt slots off the mask (ascending = parents first), then a plain loop that asks for the record eight
         * steps ahead and, once that has arrived, for the entity four steps ahead */
        uint32_t R = 0;
        for (uint32_t w = 0; w < words; w++) {
            uint64_t m = scat ? scat[w] : 0;
            while (m) {
                const uint32_t slot = w * 64 + (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                if (res.slot_user[slot] && push_u32(&gs->slots, &R, &gs->cap_slots, slot)) return _CERR_NOMEM;
            }
        }
        for (uint32_t k = 0; k < R; k++) {
            if (k + 8 < R) {
                const uint32_t sl = gs->slots[k + 8];
                __builtin_prefetch(&gs->rec[(uintptr_t)res.slot_user[sl] - 1], 0, 1);
                __builtin_prefetch(res.mx + 16 * (size_t)sl, 0, 0);
                __builtin_prefetch(res.inverse_mx + 16 * (size_t)sl, 0, 0);
            }
            if (k + 4 < R)
                prefetch_entity(gs->rec[(uintptr_t)res.slot_user[gs->slots[k + 4]] - 1].e);
            const uint32_t slot = gs->slots[k];
            struct gs_rec *rr = &gs->rec[(uintptr_t)res.slot_user[slot] - 1];
            if (rr->cls != 1) continue;                          /* class 4: after the pose, from the second launch */
            scatter_one(gs, rr, &res, slot, true);
            st->written_back++;
        }
    }
    const double ts2 = timing ? now_ms() : 0;
    consume_fetched(gs);                                         /* came into view (or contain the camera) after frames of being left out */
    const double t3 = now_ms();
    /* host hooks + bounding-volume pick, merged in list order */
    /* candidates come off the mask in slot order; list order is restored through a bitmap over the walk's positions
     * (one bit per queue position: 125 KB per million entities), which the merge below scans upwards */
    uint32_t n_cand = 0;
    const uint32_t pos_words = (gs->n_order + 63) / 64;
    if (gs->appended) {
        /* order[] is not the list any more (entities taken in since the walk stand at its end): the candidates -- few --
         * are sorted by their place in the queue instead, and merged with the host-class entities by that */
        if (scene && res.inside_mask)
            for (uint32_t w = 0; w < words; w++) {
                uint64_t m = res.inside_mask[w];
                while (m) {
                    const size_t slot = (size_t)w * 64 + (size_t)__builtin_ctzll(m);
                    m &= m - 1;
                    const uintptr_t u = (uintptr_t)res.slot_user[slot];
                    if (!u) continue;
                    if (n_cand == gs->cap_cands) {
                        const uint32_t cap = gs->cap_cands ? 2 * gs->cap_cands : 256;
                        struct gs_cand *q = realloc(gs->cands, (size_t)cap * sizeof(*q));
                        if (!q) return _CERR_NOMEM;
                        gs->cands = q; gs->cap_cands = cap;
                    }
                    gs->cands[n_cand++] = (struct gs_cand){ gs->rec[u - 1].order_key, (uint32_t)(u - 1) };
                }
            }
        if (n_cand > 1) qsort(gs->cands, n_cand, sizeof(*gs->cands), cand_cmp);
        uint32_t hc = 0, ci = 0;
        for (;;) {
            const uint64_t ck = ci < n_cand ? gs->cands[ci].key : UINT64_MAX;
            const uint64_t hk = hc < gs->n_host ? gs->rec[gs->host_list[hc]].order_key : UINT64_MAX;
            if (ck == UINT64_MAX && hk == UINT64_MAX) break;
            if (hk < ck)
                host_hook(gs, mq, &gs->rec[gs->host_list[hc++]]);
            else {
                const struct gs_rec *cr = &gs->rec[gs->cands[ci++].rec];
                if (cr->cls == 1 && cr->e && !cr->gone) bv_pick(scene, cr->e);
            }
        }
    } else {
    if (scene && res.inside_mask) {
        if (pos_words > gs->cap_posmap) {
            uint64_t *pm = realloc(gs->posmap, (size_t)pos_words * 8);
            if (!pm) return _CERR_NOMEM;
            gs->posmap = pm; gs->cap_posmap = pos_words;
        }
        bool cleared = false;
        for (uint32_t w = 0; w < words; w++) {
            uint64_t m = res.inside_mask[w];
            while (m) {
                const size_t slot = (size_t)w * 64 + (size_t)__builtin_ctzll(m);
                m &= m - 1;
                const uintptr_t u = (uintptr_t)res.slot_user[slot];
                if (!u) continue;
                if (!cleared) { memset(gs->posmap, 0, (size_t)pos_words * 8); cleared = true; }
                const uint32_t op = gs->rec[u - 1].order_pos;
                gs->posmap[op >> 6] |= 1ull << (op & 63);
                n_cand++;
            }
        }
    }
    uint32_t hc = 0, cw = 0;
    uint64_t cm = n_cand ? gs->posmap[0] : 0;
    for (;;) {
        while (n_cand && !cm && cw + 1 < pos_words) cm = gs->posmap[++cw];
        const uint32_t co = cm ? cw * 64 + (uint32_t)__builtin_ctzll(cm) : 0xffffffffu;
        const uint32_t ho = hc < gs->n_host ? gs->rec[gs->host_list[hc]].order_pos : 0xffffffffu;
        if (co == 0xffffffffu && ho == 0xffffffffu) break;
        if (ho < co) {
            host_hook(gs, mq, &gs->rec[gs->host_list[hc++]]);
        } else {
            cm &= cm - 1;
            if (gs->rec[gs->order[co]].cls == 1 && !gs->rec[gs->order[co]].gone)   /* class 4 boxes are last frame's until the second launch */
                bv_pick(scene, gs->rec[gs->order[co]].e);
        }
    }
    }
    st->batched = gs->n_batched; st->host = gs->n_host + gs->n_deferred;
    if (timing) fprintf(stderr, "fast_frame: mirror %.3f device %.3f scatter %.3f = lag+pend %.3f count %.3f rows %.3f fetched %.3f (rebuilt %llu) hooks+bv %.3f (cand %u host %u)\n", t1 - t0, t2 - t1, t3 - t2, ts0 - t2, ts1 - ts0, ts2 - ts1, t3 - ts2, (unsigned long long)n_rebuilt, now_ms() - t3, n_cand, gs->n_host);
    st->ms_walk = t1 - t0; st->ms_device = t2 - t1; st->ms_scatter = now_ms() - t2;
    return 0;
}

/*
 * Frames WITHOUT notifications.  Nothing tells the binding what changed, so the reference's way is to look at every entity --
 * but not necessarily by chasing the lists on one core: if the queue is still the one the last walk met (every entity's list
 * successor is the next record's entity, every txmodel's list starts and ends where it did: checked on the workers, one
 * list node per entity) the frame goes by the records -- every record "touched", the mirror pass on the workers re-reading
 * what a walk would read (flags, xform.updated, the inputs of the entity's class, its LODs) -- and falls back to the walk
 * the moment anything a walk would have classified differently shows up.  1 M entities: 72 ms of list walk -> a few ms.
 */
struct quc_ctx { struct gpu_scene *gs; int changed; };
static void queue_unchanged_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct quc_ctx *qc = ctx;
    struct gpu_scene *gs = qc->gs;
    for (uint32_t k = lo; k < hi; k++) {
        const struct gs_rec *r = &gs->rec[gs->order[k]];
        if (k + 8 < hi) __builtin_prefetch(&gs->rec[gs->order[k + 8]].e->entry, 0, 1);
        const uint32_t rank = (uint32_t)(r->order_key >> 32);
        const struct list *head = &gs->wtxm[rank].txm->entities;
        const struct list *n = r->e->entry.next;                 /* the next ALIVE entity behind it in its txmodel's list */
        while (n != head && !entity3d_matches(list_entry((struct list *)n, entity3d, entry), ENTITY3D_ALIVE)) n = n->next;
        const entity3d *want = (k + 1 < gs->n_order && (uint32_t)(gs->rec[gs->order[k + 1]].order_key >> 32) == rank)
                               ? gs->rec[gs->order[k + 1]].e : NULL;
        const entity3d *got = n == head ? NULL : list_entry((struct list *)n, entity3d, entry);
        if (got != want) { __atomic_store_n(&qc->changed, 1, __ATOMIC_RELAXED); return; }
    }
}

static bool queue_unchanged(struct gpu_scene *gs, struct mq *mq)
{
    uint32_t t = 0;
    model3dtx *txm;
    list_for_each_entry(txm, &mq->txmodels, entry) {             /* the txmodels, and where each one's list starts */
        if (t >= gs->n_wtxm || gs->wtxm[t].txm != txm) return false;
        const struct list *head = &txm->entities, *n = head->next;
        while (n != head && !entity3d_matches(list_entry((struct list *)n, entity3d, entry), ENTITY3D_ALIVE)) n = n->next;
        const entity3d *first = n == head ? NULL : list_entry((struct list *)n, entity3d, entry);
        const entity3d *want = gs->wtxm[t].next ? gs->rec[gs->order[gs->wtxm[t].first]].e : NULL;
        if (first != want) return false;
        t++;
    }
    if (t != gs->n_wtxm) return false;
    struct quc_ctx qc = { gs, 0 };
    const double q0 = getenv("GPU_SCENE_TIMING") ? now_ms() : 0;
    /* a small queue on the calling thread: waking the workers costs more than looking at a few thousand list nodes */
    gpu_scene_par_for(queue_unchanged_range, &qc, gs->n_order, gs->n_order >= GS_REPLAY_MIN ? par_threads() : 1);
    if (q0 != 0) fprintf(stderr, "queue_unchanged: %u entities in %.3f ms (%s)\n", gs->n_order, now_ms() - q0, qc.changed ? "changed" : "the same");
    return !qc.changed;
}

/*
 * A walked frame that RE-TILED.  The device has rebuilt every row of the new layout, so its mask cannot say what the reference
 * would have rebuilt -- the host fields do (model.c:1609-1616, 1667): an entity is rebuilt if its transform was written, or if
 * its parent_seq is not its parent's seq AS THE PARENT LEAVES THIS FRAME (the parent comes earlier in the list).  That is a
 * recurrence up the ancestor chain -- rebuilt(e) = dirty(e) || parent_seq(e) != seq(parent) + rebuilt(parent) --, which one
 * thread used to evaluate in list order over every entity3d (1 M entities: 45-50 ms).  Here: pass A copies the four values it
 * needs out of every batched entity (on the workers), pass B walks each entity's chain over that compact array until it
 * meets a decided ancestor (states are written once with the same value by whoever gets there first), and the result is a
 * mask by slot of the NEW layout -- which frame_results() takes in place of the device's, write-back on the workers, hooks
 * and bounding-volume pick merged in list order, exactly as after a frame whose layout stood.
 */
#define HF_NONE 0xffffffffu
struct hf_ctx { struct gpu_scene *gs; uint32_t n_slots; };
static void hf_collect_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct hf_ctx *hc = ctx;
    struct gpu_scene *gs = hc->gs;
    for (uint32_t k = lo; k < hi; k++) {
        struct gs_rec *r = &gs->rec[gs->order[k]];
        struct gs_hf *h = &gs->hf[k];
        if (k + 24 < hi) __builtin_prefetch(&gs->rec[gs->order[k + 24]], 0, 1);
        if (k + 8 < hi) prefetch_entity(gs->rec[gs->order[k + 8]].e);
        h->state = 1; h->dirty = 0; h->ppos = HF_NONE; h->seq0 = h->pseq = 0;
        if (r->gone || (r->cls != 1 && r->cls != 4)) continue;
        const entity3d *e = r->e;
        r->slot = clapgpu_scene_entity_slot(gs->scene, r->handle);
        if (r->slot != CLAPGPU_NO_ENTITY) seq_shown(gs, r->slot, e->seq);    /* (rebuilt ones are shown their new seq by the write-back) */
        if (r->cls == 4) continue;                               /* after the pose, from the second launch: gpu_scene_run_deferred() */
        r->host_done = 0;                                        /* the host fields decide here: a host-updated entity is simply not dirty */
        h->seq0 = e->seq; h->pseq = e->parent_seq; h->dirty = r->xform_dirty;
        h->state = 0;
        if (r->slot == CLAPGPU_NO_ENTITY || r->slot >= hc->n_slots) { h->state = 1; continue; }   /* (cannot be: the mirror holds every batched entity) */
        if (h->dirty) h->state = 2;
        else if (!e->parent) h->state = 1;
        else if (r->parent_rec != NO_REC && gs->rec[r->parent_rec].e == e->parent && gs->rec[r->parent_rec].gen == gs->gen)
            h->ppos = gs->rec[r->parent_rec].order_pos;          /* a batched entity's parent is batched and comes earlier (the class rules) */
        else
            h->state = e->parent_seq != e->parent->seq ? 2 : 1;  /* (cannot be either; by the parent as it stands) */
    }
}

static uint8_t hf_decide(struct gs_hf *hf, uint32_t k)
{
    uint32_t chain[64], n = 0, cur = k;
    uint8_t s;
    for (;;) {
        s = __atomic_load_n(&hf[cur].state, __ATOMIC_RELAXED);
        if (s) break;
        if (n == 64) { s = hf_decide(hf, cur); break; }           /* a chain deeper than the stack here: in pieces */
        chain[n++] = cur;
        cur = hf[cur].ppos;
    }
    while (n) {
        const uint32_t c = chain[--n];
        s = hf[c].pseq != (uint16_t)(hf[hf[c].ppos].seq0 + (s == 2)) ? 2 : 1;
        __atomic_store_n(&hf[c].state, s, __ATOMIC_RELAXED);
    }
    return s;
}

static void hf_decide_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct hf_ctx *hc = ctx;
    struct gpu_scene *gs = hc->gs;
    for (uint32_t k = lo; k < hi; k++) {
        if (hf_decide(gs->hf, k) != 2) continue;
        const uint32_t slot = gs->rec[gs->order[k]].slot;
        __atomic_fetch_or(&gs->hf_mask[slot >> 6], 1ull << (slot & 63), __ATOMIC_RELAXED);
    }
}

static bool retile_by_mask(void)
{
    static int on = -1;
    if (on < 0) { const char *v = getenv("GPU_SCENE_RETILE_BY_MASK"); on = v ? atoi(v) != 0 : 1; }   /* A/B switch: 0 = the serial pass */
    return on;
}

static int by_host_fields(struct gpu_scene *gs, clapgpu_scene_arrays *res)
{
    if (gs->n_order > gs->cap_hf) {
        struct gs_hf *q = realloc(gs->hf, (size_t)gs->cap_order * sizeof(*q));
        if (!q) return _CERR_NOMEM;
        gs->hf = q; gs->cap_hf = gs->cap_order;
    }
    const uint32_t words = res->n_slots / 64;
    if (words > gs->cap_hf_mask) {
        uint64_t *q = realloc(gs->hf_mask, (size_t)words * 8);
        if (!q) return _CERR_NOMEM;
        gs->hf_mask = q; gs->cap_hf_mask = words;
    }
    memset(gs->hf_mask, 0, (size_t)words * 8);
    struct hf_ctx hc = { gs, res->n_slots };
    const int nt = gs->n_order >= 8192 ? par_threads() : 1;
    gpu_scene_par_for(hf_collect_range, &hc, gs->n_order, nt);
    gpu_scene_par_for(hf_decide_range, &hc, gs->n_order, nt);
    res->rebuilt_mask = gs->hf_mask;
    res->exported_mask = NULL;                                   /* (a walked frame exports everything) */
    return 0;
}

/* What a walk leaves behind for the frames that are not walked, from the records, on the workers: the verdict table in list
 * order, which batched parents a host-class child reads, which entities are standing readers under GPU_SCATTER_DRAWN. */
struct walk_tail_ctx { struct gpu_scene *gs; struct scene *scene; uint32_t n_changes; };
static void tail_verdicts_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct gpu_scene *gs = ((struct walk_tail_ctx *)ctx)->gs;
    for (uint32_t k = lo; k < hi; k++) {
        struct gs_rec *r = &gs->rec[gs->order[k]];
        gs->vq_e[k] = r->e; gs->vq_slot[k] = r->slot; gs->vq_ok[k] = verdict_ok(r) && r->slot != CLAPGPU_NO_ENTITY;
        r->host_child = 0;
    }
}

static void tail_host_child_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct gpu_scene *gs = ((struct walk_tail_ctx *)ctx)->gs;
    for (uint32_t k = lo; k < hi; k++) {
        const struct gs_rec *r = &gs->rec[gs->order[k]];
        if ((r->cls != 2 && r->cls != 3) || !r->e->parent) continue;
        const uint32_t pr = rec_find(gs, r->e->parent);
        if (pr != NO_REC) __atomic_store_n(&gs->rec[pr].host_child, 1, __ATOMIC_RELAXED);   /* (several children, one value) */
    }
}

static void tail_keep_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct walk_tail_ctx *tc = ctx;
    struct gpu_scene *gs = tc->gs;
    for (uint32_t k = lo; k < hi; k++) {
        const struct gs_rec *r = &gs->rec[gs->order[k]];
        if ((r->cls != 1 && r->cls != 4) || r->handle == CLAPGPU_NO_ENTITY) continue;
        const uint8_t keep = r->user_keep || r->host_child || r->keep_auto || (tc->scene && r->e == tc->scene->control);   /* (the records alone: keep_auto was taken while the entity was at hand) */
        if (keep != r->keep) gs->keep_changes[__atomic_fetch_add(&tc->n_changes, 1, __ATOMIC_RELAXED)] = gs->order[k];
    }
}

static int mq_update_frame(struct gpu_scene *gs, struct mq *mq, struct view *view);

int gpu_mq_update(struct gpu_scene *gs, struct mq *mq, struct view *view)
{
    if (!gs || !mq) return _CERR_INVALID_ARGUMENTS;
    /* a hook that runs inside the frame may delete entities (the reference's list walk takes that in its stride): nothing is
     * taken out of the lists the frame is iterating -- gpu_scene_entity_deleting() marks the record, the rest of the frame
     * skips it, the next frame walks */
    gs->in_frame = true;
    const int rc = mq_update_frame(gs, mq, view);
    gs->in_frame = false;
    return rc;
}

/* ---- a WALKED frame, in the order mq_update_frame() runs its parts -------------------------------------------------- */

/* A walked frame writes everything back, and it may re-tile: whatever GPU_SCATTER_DRAWN left on the device comes over
 * first, so that the host fields the walk decides by (xform.updated, seq / parent_seq) are the reference's.  The rows
 * only: an entity3d is written when the walk MEETS it -- what was deleted since the last frame (the reason for many a
 * walk) is freed memory, and nothing but the queue's own lists says which entities those are.  Then the frame's lists and
 * counters start empty. */
static int walk_begin(struct gpu_scene *gs, struct mq *mq)
{
    struct gpu_scene_stats *st = &gs->stats;
    struct scene *scene = mq->priv;
    gs->walk_fetch_on = false;
    if (gs->any_pend) {
        uint32_t n_rows = 0;
        CK(clapgpu_scene_fetch(gs->scene, NULL, &n_rows));
        clapgpu_scene_arrays fr;
        if (n_rows && !clapgpu_scene_results(gs->scene, &fr)) {
            const uint32_t words = fr.n_slots / 64;
            if (words > gs->cap_walk_fetch) {
                uint64_t *q = realloc(gs->walk_fetch, (size_t)words * 8);
                if (!q) return _CERR_NOMEM;
                gs->walk_fetch = q; gs->cap_walk_fetch = words;
            }
            memcpy(gs->walk_fetch, fr.fetched_mask, (size_t)words * 8);
            gs->res = fr;
            gs->fetch_seen = fr.fetch_serial;
            gs->walk_fetch_on = true;
        }
    }
    clapgpu_scene_set_export(gs->scene, CLAPGPU_SCENE_EXPORT_ALL);
    gs->drawn_now = false;
    for (uint32_t k = 0; k < gs->n_touched; k++) gs->rec[gs->touched[k]].pending = 0;
    gs->n_touched = 0;
    gs->n_xptr = 0;                                              /* the walk reads every transform itself */
    gs->n_created = 0;                                           /* ... and meets every entity made since the last one */
    gs->appended = false;
    gs->n_wtxm = 0;
    st->placed = gs->inc_placed; st->removed = gs->inc_removed;  /* (taken in / out in place before something else asked for the walk) */
    st->registered += gs->inc_placed; st->deleted += gs->inc_removed;
    gs->inc_placed = gs->inc_removed = 0;
    gs->topology_pending = false;
    gs->last_fast = false;
    gs->n_host = 0; gs->n_batched = 0; gs->n_deferred = 0; gs->n_att = 0; gs->n_char = 0;
    if (scene && scene->camera)                                  /* the device's containment mask: what the second half goes by when the layout stands */
        clapgpu_scene_set_bv_points(gs->scene, transform_pos(&scene->camera->xform, NULL),
                                    scene->control ? transform_pos(&scene->control->xform, NULL) : NULL, CLAPGPU_NO_ENTITY);
    else
        clapgpu_scene_set_bv_points(gs->scene, NULL, NULL, CLAPGPU_NO_ENTITY);

    return 0;
}

/* steps 2 and 3 for ONE entity the walk has met and given its place in order[]: its class (from its own criteria and its
 * parent's class, which is settled: the parent comes earlier or does not count), then what the class asks of the mirror */
static int walk_act(struct gpu_scene *gs, struct mq *mq, uint32_t i);

static int walk_classify(struct gpu_scene *gs, struct mq *mq, uint32_t i)
{
    struct gs_rec *r = &gs->rec[i];
    entity3d *e = r->e;
    r->self_ok = self_batchable(gs, e);
    const bool rides_joint = e->parent && e->parent_joint != JOINT_TYPE_MAX;
    r->rides = rides_joint; r->animated = entity_animated(e);
    if (!r->self_ok) {
        r->cls = 2;
        r->parent_e = e->parent; r->parent_rec = NO_REC;
        /* With the pose computed after this update (gpu_anim_update), an entity riding a parent's joint
         * (model.c:1626-1641) must wait for it: the reference gives it the joint transforms of THIS frame,
         * written by the parent's animated_update earlier in the list.  It -- and everything below it -- is
         * run by gpu_scene_run_deferred(), which gpu_anim_update calls when the palettes are back. */
        if (gs->anim_elsewhere && e->parent) {
            /* only behind a parent that comes EARLIER in the list: one that comes later is read one frame late
             * by the reference, joint transforms included, which running the hook right here reproduces */
            const uint32_t p = rec_find(gs, e->parent);
            if (p != NO_REC && gs->rec[p].gen == gs->gen &&
                (rides_joint || gs->rec[p].cls == 3 || gs->rec[p].cls == 4))
                r->cls = 3;
        }
    } else if (!e->parent) {
        r->cls = 1;
        r->parent_e = NULL; r->parent_rec = NO_REC;       /* (a detached child: else every later touch reads as "re-parented") */
    } else {
        /* NO_REC unless already met in THIS walk.  A child that precedes its parent in list order sees the
         * parent's matrix of the previous frame in the reference (model.c:1911-1922 walks creation order):
         * it stays on the host, where that lag is reproduced exactly, and so does everything below it. */
        const uint32_t p = parent_rec(gs, r);
        const uint8_t pc = p != NO_REC ? gs->rec[p].cls : 2;
        if (!rides_joint) {
            r->cls = pc;                                  /* 1, 4 (below a joint rider), or the parent's host class */
            if (pc == 4 && entity_animated(e)) r->cls = 3;   /* its own pose would need its matrix before the second launch */
        } else if (gs->anim_elsewhere && pc == 1 && !entity_animated(e)) {
            /* rides a joint of a character whose palette the device computes this frame: the frame's second
             * entity launch, behind the pose (gpu_scene_run_deferred) */
            r->cls = 4;
        } else {
            /* the parent's hook runs on the host (its palette is fresh when it returns), or the rider is nested
             * below another rider / animated itself: its own hook, deferred behind the pose when that runs elsewhere */
            r->cls = (gs->anim_elsewhere && p != NO_REC) ? 3 : 2;
        }
    }
    return walk_act(gs, mq, i);
}

static int walk_act(struct gpu_scene *gs, struct mq *mq, uint32_t i)
{
    struct gs_rec *r = &gs->rec[i];
    entity3d *e = r->e;
    if (r->cls == 1 || r->cls == 4) {
        r->keep_auto = r->cls == 4 || e->light_idx >= 0 || e->update != gs->default_hook || entity_animated(e);
        if (e->update != gs->default_hook) {             /* a body-less character: its hook's host half, at its place in the list */
            gs->char_half(e, mq->priv);
            if (push_u32(&gs->char_list, &gs->n_char, &gs->cap_char, i)) return _CERR_NOMEM;
            r = &gs->rec[i];
        }
        CK(mirror_one(gs, r));
        CK(link_parent(gs, r));
    } else {
        CK(unbatch(gs, r));
    }
    return 0;
}

/*
 * The same two steps for a big queue, on the workers.  The list chase is serial by nature; what the walk does per entity
 * besides it is not, and at a million entities that was most of its 70 ms (four or five cache lines of every 448-byte
 * entity3d, its record, the mirror's record, three rows of the upload image).  So the chase only matches records and fills
 * order[], and then, over order[]:
 *   A  every entity's own criteria and its parent's place in the list (own record only; the parents' records are read),
 *   B  the classes: an entity's class is a function of its criteria and of its parent's class when that parent comes EARLIER
 *      in the list -- a recurrence up the ancestor chain, walked per entity until it meets a decided ancestor (states are
 *      written once, the same value by whoever gets there first),
 *   C  what the class asks of the mirror, where that is a push of flags and transform (clapgpu_scene_entity_transform_mt:
 *      nothing shared is touched); anything that changes the mirror's make-up -- a new handle, another model, another parent,
 *      a joint attachment, a LOD, an entity that leaves the batch, a character's host half -- is noted and
 *   D  done afterwards on this thread in list order by walk_act(), the serial walk's own code; so is the parent link of the
 *      children of such entities.
 */
#define WQ_NONE 0xffffffffu
struct gs_wq { uint32_t ppos; uint8_t state, todo; };             /* todo: 1 = walk_act on this thread, 2 = its parent link only */
struct wq_ctx { struct gpu_scene *gs; int rc; uint32_t pushed; };

static void wq_inputs_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct gpu_scene *gs = ((struct wq_ctx *)ctx)->gs;
    for (uint32_t k = lo; k < hi; k++) {
        struct gs_rec *r = &gs->rec[gs->order[k]];
        struct gs_wq *w = &gs->wq[k];
        if (k + 24 < hi) __builtin_prefetch(&gs->rec[gs->order[k + 24]], 0, 1);
        if (k + 8 < hi) prefetch_entity(gs->rec[gs->order[k + 8]].e);
        entity3d *e = r->e, *p = e->parent;
        w->ppos = WQ_NONE; w->todo = 0;
        r->self_ok = self_batchable(gs, e);
        r->rides = p && e->parent_joint != JOINT_TYPE_MAX;
        r->animated = entity_animated(e);
        uint32_t pi = NO_REC;
        if (!r->self_ok) {
            r->parent_e = p; r->parent_rec = NO_REC;
            if (p && gs->anim_elsewhere) pi = rec_find(gs, p);
        } else if (!p) {
            r->parent_e = NULL; r->parent_rec = NO_REC;
        } else {
            if (r->parent_e != p || r->parent_rec == NO_REC || gs->rec[r->parent_rec].e != p) {   /* (parent_rec()) */
                r->parent_e = p;
                r->parent_rec = rec_find(gs, p);
            }
            pi = r->parent_rec;
        }
        if (pi != NO_REC && gs->rec[pi].gen == gs->gen && gs->rec[pi].order_pos < k) w->ppos = gs->rec[pi].order_pos;   /* met EARLIER in this walk */
        if (!r->self_ok) w->state = (w->ppos == WQ_NONE) ? 2 : r->rides ? 3 : 0;
        else if (!p) w->state = 1;
        else w->state = (w->ppos == WQ_NONE) ? 2 : 0;
    }
}

static uint8_t wq_class(const struct gpu_scene *gs, uint32_t k)
{
    struct gs_wq *wq = gs->wq;
    uint32_t chain[64], n = 0, cur = k;
    uint8_t s;
    for (;;) {
        s = __atomic_load_n(&wq[cur].state, __ATOMIC_RELAXED);
        if (s) break;
        if (n == 64) { s = wq_class(gs, cur); break; }
        chain[n++] = cur;
        cur = wq[cur].ppos;
    }
    while (n) {
        const uint32_t c = chain[--n];
        const struct gs_rec *r = &gs->rec[gs->order[c]];
        const uint8_t pc = s;                                    /* the class of c's parent, which comes earlier in the list */
        if (!r->self_ok) s = (pc == 3 || pc == 4) ? 3 : 2;
        else if (!r->rides) s = (pc == 4 && r->animated) ? 3 : pc;
        else if (gs->anim_elsewhere && pc == 1 && !r->animated) s = 4;
        else s = gs->anim_elsewhere ? 3 : 2;
        __atomic_store_n(&wq[c].state, s, __ATOMIC_RELAXED);
    }
    return s;
}

static void wq_class_range(void *ctx, uint32_t lo, uint32_t hi)
{
    const struct gpu_scene *gs = ((struct wq_ctx *)ctx)->gs;
    for (uint32_t k = lo; k < hi; k++) wq_class(gs, k);
}

static void wq_act_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct wq_ctx *wc = ctx;
    struct gpu_scene *gs = wc->gs;
    uint32_t pushed = 0;
    for (uint32_t k = lo; k < hi; k++) {
        struct gs_rec *r = &gs->rec[gs->order[k]];
        struct gs_wq *w = &gs->wq[k];
        entity3d *e = r->e;
        r->cls = w->state;
        if (r->cls != 1 && r->cls != 4) { w->todo = r->handle != CLAPGPU_NO_ENTITY; continue; }   /* leaves the batch: unbatch() */
        r->keep_auto = r->cls == 4 || e->light_idx >= 0 || e->update != gs->default_hook || r->animated;
        const uint8_t att = r->cls == 4 && e->parent_joint != JOINT_TYPE_MAX;
        const uint32_t ph = e->parent ? gs->rec[r->parent_rec].handle : CLAPGPU_NO_ENTITY;
        if (e->update != gs->default_hook || r->handle == CLAPGPU_NO_ENTITY || r->model != e->txmodel->model ||
            e->force_lod != r->lod_force || e->cur_lod != r->lod_cur || att != r->att || ph != r->parent_handle ||
            (e->parent && ph == CLAPGPU_NO_ENTITY)) {
            w->todo = 1;
            continue;
        }
        /* mirror_one(), the part that changes nothing but this entity's own inputs */
        const uint32_t flags = e->flags & (ENTITY3D_ALIVE | 0xffffu);
        const bool same_flags = flags == r->flags;
        r->flags = flags;
        r->xform_dirty = transform_is_updated(&e->xform);
        if (r->host_done && r->xform_dirty) r->host_done = 2;
        if (!r->xform_dirty && !r->host_done && same_flags) continue;
        const int rc = clapgpu_scene_entity_transform_mt(gs->scene, r->handle, transform_pos(&e->xform, NULL),
                                                         transform_rotation_quat(&e->xform), e->scale, flags, r->xform_dirty || r->host_done);
        if (rc) __atomic_store_n(&wc->rc, rc, __ATOMIC_RELAXED);
        if (r->xform_dirty || r->host_done) pushed++;
    }
    __atomic_fetch_add(&wc->pushed, pushed, __ATOMIC_RELAXED);
}

/* children of entities whose handle is about to change (a new handle, another model): their parent link follows it */
static void wq_links_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct gpu_scene *gs = ((struct wq_ctx *)ctx)->gs;
    for (uint32_t k = lo; k < hi; k++) {
        const struct gs_rec *r = &gs->rec[gs->order[k]];
        struct gs_wq *w = &gs->wq[k];
        if (__atomic_load_n(&w->todo, __ATOMIC_RELAXED) || (r->cls != 1 && r->cls != 4) || !r->e->parent) continue;
        const struct gs_rec *pr = &gs->rec[r->parent_rec];
        if (pr->order_pos < gs->n_order && __atomic_load_n(&gs->wq[pr->order_pos].todo, __ATOMIC_RELAXED) == 1)
            __atomic_store_n(&w->todo, 2, __ATOMIC_RELAXED);     /* (a neighbour may be reading this one as ITS parent's: 0 or 2, never 1) */
    }
}

static uint32_t walk_par_min(void)
{
    static uint32_t v;
    if (!v) { const char *e = getenv("GPU_SCENE_WALK_PAR_MIN"); v = e && atoi(e) > 0 ? (uint32_t)atoi(e) : 16384u; }   /* tuning knob; the tests set 1 */
    return v;
}

static int walk_queue(struct gpu_scene *gs, struct mq *mq)
{
    struct gpu_scene_stats *st = &gs->stats;
    /*
     * 1-3 in ONE walk of the queue (the entity structs are far larger than the caches, so every
     * extra pass over them costs as much as the reference's whole update).  prev_order[] is last
     * frame's walk: an unchanged queue is matched without hashing, and its entities are prefetched
     * ahead of the list chase.  A big queue's steps 2 and 3 follow on the workers (above).
     */
    { uint32_t *t = gs->prev_order; gs->prev_order = gs->order; gs->order = t; }
    gs->n_prev = gs->n_order;
    gs->n_order = 0;
    const bool later = gs->n_prev >= walk_par_min() && par_threads() > 1;   /* (by last walk's size: a first walk goes one by one) */
    const double t_chase = now_ms();
    static uint32_t ahead_by;
    static uint32_t rec_ahead;
    if (!ahead_by) {                                             /* tuning knobs */
        const char *a = getenv("GPU_SCENE_CHASE_AHEAD"), *b = getenv("GPU_SCENE_CHASE_REC_AHEAD");
        ahead_by = a && atoi(a) > 0 ? (uint32_t)atoi(a) : 12u;   /* the entity's list node 12 steps ahead, through a record asked for 32 ahead: */
        rec_ahead = b ? (uint32_t)atoi(b) : 32u;                /* 15.4-16.9 -> 11.6-12.8 ms at 1 M entities (8 / none before; 16 / 40 and 32 / none: slower) */
    }
    uint32_t cursor = 0;
    model3dtx *txm;
    entity3d *e, *it;
    list_for_each_entry(txm, &mq->txmodels, entry) {
        if (gs->n_wtxm == gs->cap_wtxm) {
            const uint32_t cap = gs->cap_wtxm ? 2 * gs->cap_wtxm : 32;
            struct gs_wtxm *q = realloc(gs->wtxm, (size_t)cap * sizeof(*q));
            if (!q) return _CERR_NOMEM;
            gs->wtxm = q; gs->cap_wtxm = cap;
        }
        const uint32_t rank = gs->n_wtxm++;
        gs->wtxm[rank] = (struct gs_wtxm){ txm, 0, gs->n_order };
        list_for_each_entry_iter(e, it, &txm->entities, entry) {
            if (!entity3d_matches(e, ENTITY3D_ALIVE)) continue;
            uint32_t i;
            if (cursor < gs->n_prev && gs->rec[gs->prev_order[cursor]].e == e) {
                i = gs->prev_order[cursor++];
                if (rec_ahead && cursor + rec_ahead < gs->n_prev) __builtin_prefetch(&gs->rec[gs->prev_order[cursor + rec_ahead]], 0, 1);
                if (cursor + ahead_by < gs->n_prev) {
                    const entity3d *ahead = gs->rec[gs->prev_order[cursor + ahead_by]].e;   /* (NULL: a tombstone of order[]) */
                    if (!later) prefetch_entity(ahead);
                    else if (ahead) __builtin_prefetch(&ahead->entry, 0, 1);   /* the chase reads the list node and the flags */
                }
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
            if (r->gone) {                                       /* the entity this record knew was deleted: e is a new one at its address */
                CK(unbatch(gs, r));
                *r = (struct gs_rec){ .e = e, .next = r->next, .parent_rec = NO_REC, .handle = CLAPGPU_NO_ENTITY, .slot = CLAPGPU_NO_ENTITY,
                                      .parent_handle = CLAPGPU_NO_ENTITY };
            }
            if (gs->walk_fetch_on && (r->cls == 1 || r->cls == 4) && r->slot < gs->res.n_slots &&
                ((gs->walk_fetch[r->slot >> 6] >> (r->slot & 63)) & 1)) {
                scatter_fetched(gs, r, &gs->res, r->slot);       /* (its class and slot are still last walk's) */
                st->fetched++;
            }
            r->gen = gs->gen;
            r->order_pos = gs->n_order;
            r->order_key = ((uint64_t)rank << 32) | gs->wtxm[rank].next++;
            gs->order[gs->n_order++] = i;
            if (!later) CK(walk_classify(gs, mq, i));
        }
    }
    if (later && gs->n_order) {
        if (gs->n_order > gs->cap_wq) {
            struct gs_wq *q = realloc(gs->wq, (size_t)gs->cap_order * sizeof(*q));
            if (!q) return _CERR_NOMEM;
            gs->wq = q; gs->cap_wq = gs->cap_order;
        }
        struct wq_ctx wc = { gs, 0, 0 };
        const bool timing = getenv("GPU_SCENE_TIMING") != NULL;
        double tw[6] = { 0 };
        if (timing) tw[0] = now_ms();
        gpu_scene_par_for(wq_inputs_range, &wc, gs->n_order, par_threads());
        if (timing) tw[1] = now_ms();
        gpu_scene_par_for(wq_class_range, &wc, gs->n_order, par_threads());
        if (timing) tw[2] = now_ms();
        gpu_scene_par_for(wq_act_range, &wc, gs->n_order, par_threads());
        if (timing) tw[3] = now_ms();
        gpu_scene_par_for(wq_links_range, &wc, gs->n_order, par_threads());
        if (timing) tw[4] = now_ms();
        if (wc.rc) return wc.rc;
        st->uploaded += wc.pushed;
        clapgpu_scene_mark_all_dirty(gs->scene);
        uint32_t n_todo = 0;
        for (uint32_t k = 0; k < gs->n_order; k++) {             /* D: what changes the mirror's make-up, in list order */
            const uint8_t todo = gs->wq[k].todo;
            if (todo == 1) CK(walk_act(gs, mq, gs->order[k]));
            else if (todo == 2) CK(link_parent(gs, &gs->rec[gs->order[k]]));
            n_todo += todo != 0;
        }
        if (timing)
            fprintf(stderr, "walk: chase %.3f ms, criteria %.3f, classes %.3f, pushes %.3f, links %.3f, %u entities one by one %.3f\n",
                    tw[0] - t_chase, tw[1] - tw[0], tw[2] - tw[1], tw[3] - tw[2], tw[4] - tw[3], n_todo, now_ms() - tw[4]);
    }
    if (gs->any_pend) {                                          /* what the walk did not meet is gone, and its counters with it */
        if (gs->pend) memset(gs->pend, 0, (size_t)gs->cap_pend * sizeof(*gs->pend));
        gs->any_pend = false;
        gs->walk_fetch_on = false;
    }
    return 0;
}

/* records of entities that left the queue since the last walk; the lists the second half of the frame goes by */
static int walk_settle(struct gpu_scene *gs)
{
    struct gpu_scene_stats *st = &gs->stats;
    for (uint32_t k = 0; k < gs->n_dead_recs; k++) {             /* taken out in place since the last walk: order[] no longer names them */
        struct gs_rec *r = &gs->rec[gs->dead_recs[k]];
        if (r->e) continue;                                      /* (cannot be: nothing hands a tombstone out before this) */
        r->next = gs->free_rec;
        gs->free_rec = gs->dead_recs[k];
    }
    gs->n_dead_recs = 0;
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

    /* the lists the second half of the frame goes by, in list order -- from the records: the classes are settled */
    for (uint32_t k = 0; k < gs->n_order; k++) {
        const struct gs_rec *r = &gs->rec[gs->order[k]];
        if (r->cls == 1) { gs->n_batched++; continue; }
        if (r->cls == 4) { gs->n_batched++; if (push_u32(&gs->att_list, &gs->n_att, &gs->cap_att, gs->order[k])) return _CERR_NOMEM; }
        else if (r->cls == 3) { if (push_u32(&gs->deferred, &gs->n_deferred, &gs->cap_deferred, gs->order[k])) return _CERR_NOMEM; }
        else if (push_u32(&gs->host_list, &gs->n_host, &gs->cap_host, gs->order[k])) return _CERR_NOMEM;
    }
    /* host-class children that precede their BATCHED parent in the list (see lag_parent above) */
    gs->n_lag = 0;
    for (uint32_t k = 0; k < gs->n_host; k++) {
        struct gs_rec *r = &gs->rec[gs->host_list[k]];
        r->lag = 0;
        if (!r->e->parent) continue;
        const uint32_t pr = rec_find(gs, r->e->parent);
        if (pr == NO_REC || gs->rec[pr].gen != gs->gen || gs->rec[pr].cls != 1 || gs->rec[pr].order_pos < r->order_pos) continue;
        if (push_u32(&gs->lag_parent, &gs->n_lag, &gs->cap_lag, pr)) return _CERR_NOMEM;
        r->lag = gs->n_lag;
    }
    if (gs->n_lag) {
        struct lag_keep *lk = realloc(gs->lag_keep, (size_t)gs->cap_lag * sizeof(*lk));
        if (!lk) return _CERR_NOMEM;
        gs->lag_keep = lk;
    }

    return 0;
}

/* 4: the device -- and, under GPU_SCATTER_DRAWN, the per-slot counters laid out for the layout it left */
static int walk_device(struct gpu_scene *gs, struct view *view, clapgpu_scene_arrays *out, bool *shown_stands_out)
{
    struct gpu_scene_stats *st = &gs->stats;
    const uint32_t layout_before = clapgpu_scene_layout_generation(gs->scene);
    clapgpu_frustum fr;
    if (view) frustum_of(view, &fr);
    CK(views_before_update(gs, view));
    CK(clapgpu_scene_mq_update(gs->scene, view ? &fr : NULL));
    st->retiled = layout_before != clapgpu_scene_layout_generation(gs->scene);
    gs->culled_view = view;
    gs->vis_cursor = 0;
    if (view) memcpy(gs->culled_planes, view->main.frustum_planes, sizeof(gs->culled_planes));
    gs->cull_checked = false;
    clapgpu_scene_arrays res = { 0 };
    if (clapgpu_scene_results(gs->scene, &res))                  /* an empty batch has none */
        memset(&res, 0, sizeof(res));
    gs->res = res;

    bool shown_stands = true;                                    /* shown[] of the last frames still describes this layout's slots */
    if (gs->scatter_drawn && gs->notify && res.n_slots) {        /* the counters GPU_SCATTER_DRAWN keeps per slot, for this layout */
        shown_stands = !st->retiled && gs->shown && gs->cap_pend >= res.n_slots && !gs->shown_stale;
        gs->shown_stale = false;
        if (res.n_slots > gs->cap_pend) {
            uint16_t *pn = realloc(gs->pend, (size_t)res.n_slots * sizeof(*pn));
            if (pn) gs->pend = pn;
            uint16_t *sn = realloc(gs->shown, (size_t)res.n_slots * sizeof(*sn));
            if (sn) gs->shown = sn;
            if (!pn || !sn) return _CERR_NOMEM;
            gs->cap_pend = res.n_slots;
        } else if (!gs->shown) {
            gs->shown = malloc((size_t)gs->cap_pend * sizeof(*gs->shown));
            if (!gs->shown) return _CERR_NOMEM;
        }
        memset(gs->pend, 0, (size_t)gs->cap_pend * sizeof(*gs->pend));   /* (every counter was consumed with the walk's fetch) */
        if (!shown_stands) memset(gs->shown, 0, (size_t)gs->cap_pend * sizeof(*gs->shown));
    }
    gs->shown_live = gs->scatter_drawn && gs->notify && res.n_slots && gs->shown;
    *out = res;
    *shown_stands_out = shown_stands;
    return 0;
}

/* 5 on one thread in list order (GPU_SCENE_RETILE_BY_MASK=0, and a queue with nothing batched): results and host hooks */
static void second_half_serial(struct gpu_scene *gs, struct mq *mq, const clapgpu_scene_arrays *resp)
{
    struct gpu_scene_stats *st = &gs->stats;
    struct scene *scene = mq->priv;
    const clapgpu_scene_arrays res = *resp;
    entity3d *e;
    for (uint32_t k = 0; k < gs->n_order; k++) {
        struct gs_rec *r = &gs->rec[gs->order[k]];
        e = r->e;
        if (r->gone) continue;                                   /* deleted by a hook that ran earlier in this very pass: freed memory */
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
        if (r->cls == 4) {                                       /* after the pose, from the second launch: gpu_scene_run_deferred() */
            st->batched++;
            if (st->retiled || r->slot == CLAPGPU_NO_ENTITY)
                r->slot = clapgpu_scene_entity_slot(gs->scene, r->handle);
            seq_shown(gs, r->slot, e->seq);
            continue;
        }
        if (r->cls != 1) {
            st->host++;
            if (r->cls == 3) continue;                           /* after the pose: gpu_scene_run_deferred() */
            if (!r->gone) entity3d_update(e, mq->priv);          /* (gone: deleted by a hook that ran earlier in this very pass) */
            continue;
        }
        st->batched++;
        if (st->retiled || r->slot == CLAPGPU_NO_ENTITY)
            r->slot = clapgpu_scene_entity_slot(gs->scene, r->handle);
        entity3d *parent = e->parent;
        const bool rebuilt = parent ? (r->xform_dirty || e->parent_seq != parent->seq) : r->xform_dirty;
        r->host_done = 0;                                        /* the host fields decide here: a host-updated entity is simply not dirty */
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
            light_hand_off(gs, e);
            st->written_back++;
        }
        seq_shown(gs, r->slot, e->seq);
        if (scene)
            bv_pick(scene, e);
    }
}

/* what a walk leaves behind for the frames that are not walked: verdict table, address table, slot arrays, standing readers */
static int walk_tail(struct gpu_scene *gs, struct scene *scene)
{
    if (gs->n_order > gs->cap_vq) {
        const uint32_t cap = gs->cap_order;
        entity3d **ve = realloc(gs->vq_e, (size_t)cap * sizeof(*ve));
        if (ve) gs->vq_e = ve;
        uint32_t *vs = realloc(gs->vq_slot, (size_t)cap * 4);
        if (vs) gs->vq_slot = vs;
        uint8_t *vo = realloc(gs->vq_ok, cap);
        if (vo) gs->vq_ok = vo;
        if (!ve || !vs || !vo) return _CERR_NOMEM;
        gs->cap_vq = cap;
    }
    struct walk_tail_ctx tc = { gs, scene, 0 };
    const int tail_threads = gs->n_order >= GS_TABLES_PAR_MIN ? par_threads() : 1;
    gpu_scene_par_for(tail_verdicts_range, &tc, gs->n_order, tail_threads);
    if (gs->notify && ftab_build(gs)) gs->n_xptr = 0;            /* without the table gpu_scene_touch_xform takes the checked path */
    slot_arrays_build(gs);                                       /* on failure the draw list goes through the records */
    /* GPU_SCATTER_DRAWN: the standing host readers (gpu-scene.h).  A host-class entity's hook reads its parent's mx / seq
     * (parent_transform_apply, model.c:1609-1641) -- also when that parent comes later in the list (lag_parent) */
    if (gs->notify)                                              /* (also: such a parent cannot be taken out of the layout in place) */
        gpu_scene_par_for(tail_host_child_range, &tc, gs->n_order, tail_threads);
    if (gs->scatter_drawn && gs->notify) {
        gs->last_control = scene ? scene->control : NULL;
        if (gs->n_order > gs->cap_keep_changes) {
            uint32_t *q = realloc(gs->keep_changes, (size_t)gs->cap_order * sizeof(*q));
            if (!q) return _CERR_NOMEM;
            gs->keep_changes = q; gs->cap_keep_changes = gs->cap_order;
        }
        gpu_scene_par_for(tail_keep_range, &tc, gs->n_order, tail_threads);
        for (uint32_t c = 0; c < tc.n_changes; c++) {            /* the mirror's own bookkeeping: on this thread */
            struct gs_rec *r = &gs->rec[gs->keep_changes[c]];
            const uint8_t keep = !r->keep;
            if (!clapgpu_scene_entity_keep(gs->scene, r->handle, keep)) r->keep = keep;
        }
    }
    return 0;
}

static int mq_update_frame(struct gpu_scene *gs, struct mq *mq, struct view *view)
{
    struct gpu_scene_stats *st = &gs->stats;
    struct scene *scene = mq->priv;
    gs->hook_data = mq->priv;
    memset(st, 0, sizeof(*st));
    gs->gen++;
    if (gs->notify && gs->walked && !gs->topology_pending) {
        gs->gen--;                                                /* nothing entered or left the queue: the records' generation stands */
        const unsigned int untouched = gs->verify ? verify_untouched(gs) : 0;
        const int rc = fast_frame(gs, mq, view);
        st->untouched_writes = untouched;
        gs->last_fast = rc == 0;
        if (rc <= 0) return rc;
        gs->gen++;
        memset(st, 0, sizeof(*st));                               /* a touched entity changed class: walk */
    } else if (!gs->notify && gs->replay && gs->walked && !gs->topology_pending && gs->n_order >= replay_min() &&
               (par_threads() > 1 || gs->n_order < GS_REPLAY_MIN) && !gs->n_touched && queue_unchanged(gs, mq)) {
        /* no notifications, and the queue is the one the last walk met: the frame by the records (see queue_unchanged) */
        if (gs->n_order > gs->cap_touched) {
            uint32_t *q = realloc(gs->touched, (size_t)gs->cap_order * sizeof(*q));
            if (!q) return _CERR_NOMEM;
            gs->touched = q; gs->cap_touched = gs->cap_order;
        }
        memcpy(gs->touched, gs->order, (size_t)gs->n_order * sizeof(*gs->touched));
        gs->n_touched = gs->n_order;
        gs->gen--;
        gs->replaying = true;
        const int rc = fast_frame(gs, mq, view);
        gs->replaying = false;
        gs->last_fast = false;                                    /* (the word is kept for frames that looked at what was reported only) */
        if (rc <= 0) { st->replayed = rc == 0; return rc; }
        gs->gen++;
        memset(st, 0, sizeof(*st));                               /* an entity would be classified differently now: walk */
    }
    CK(walk_begin(gs, mq));
    const double t0 = now_ms();
    CK(walk_queue(gs, mq));                                      /* 1-3: the one serial pass over the lists */
    const double t1 = now_ms();
    CK(walk_settle(gs));
    const double t2 = now_ms();
    clapgpu_scene_arrays res = { 0 };
    bool shown_stands = true;
    CK(walk_device(gs, view, &res, &shown_stands));
    const double t3 = now_ms();
    if (!st->retiled && gs->walked && shown_stands && res.n_slots) {
        /* 5, the layout stood: the device's masks say what was rebuilt and which boxes hold the camera -- the second half of
         * a notified frame (frame_results), on the workers where there is much to write back */
        const int rc = frame_results(gs, mq, &res, t0, t2, t3);
        if (rc) return rc;
        st->ms_walk = t1 - t0; st->ms_mirror = t2 - t1;
    } else if (res.n_slots && retile_by_mask()) {
        /* 5, after a re-tile (the device rebuilt EVERYTHING; the host fields say what the reference would have): the same second
         * half, over a mask made from the host fields on the workers */
        clapgpu_scene_arrays hres = res;
        CK(by_host_fields(gs, &hres));
        const int rc = frame_results(gs, mq, &hres, t0, t2, t3);
        if (rc) return rc;
        st->ms_walk = t1 - t0; st->ms_mirror = t2 - t1;
    } else {
        second_half_serial(gs, mq, &res);
        st->ms_walk = t1 - t0; st->ms_mirror = t2 - t1; st->ms_device = t3 - t2; st->ms_scatter = now_ms() - t3;
    }
    CK(walk_tail(gs, scene));
    gs->walked = true;
    return 0;
}

/* view_calc_frustum() ran for `view` (view.c:291): the next verdict for it re-culls on the device if the planes changed */
void gpu_scene_view_changed(struct gpu_scene *gs, struct view *view)
{
    if (!gs) return;
    if (view == gs->culled_view) gs->cull_checked = false;
    for (uint32_t k = 0; k < gs->n_xview; k++)
        if (gs->xview[k] == view) gs->xchecked[k] = false;
}

/* The mask that answers for `view`, current for the planes the view holds NOW: the main view's (the one the last update
 * was given) or a registered view's own plane -- compared once per frustum, not per entity; planes that moved since the
 * launch that culled them cost one cull launch (every view of the frame for the main one, the one view alone otherwise).
 * NULL: the device has no answer for this view (not known to the last update, or the re-cull failed). */
static const uint64_t *mask_for_view(struct gpu_scene *gs, struct view *view)
{
    if (view == gs->culled_view) {
        if (!gs->cull_checked) {
            gs->cull_checked = true;
            gs->cull_ok = !memcmp(gs->culled_planes, view->main.frustum_planes, sizeof(gs->culled_planes));
            if (!gs->cull_ok) {
                clapgpu_frustum fr;
                frustum_of(view, &fr);
                gs->stats.cull_launches_after_update++;
                if (!clapgpu_scene_cull(gs->scene, &fr)) {
                    memcpy(gs->culled_planes, view->main.frustum_planes, sizeof(gs->culled_planes));
                    gs->cull_ok = true;
                    consume_fetched(gs);                         /* GPU_SCATTER_DRAWN: what the new planes bring into view */
                }
            }
        }
        return gs->cull_ok ? gs->res.vis_mask : NULL;
    }
    const int k = xview_of(gs, view);
    if (k < 0) return NULL;
    if (!gs->xchecked[k]) {
        gs->xchecked[k] = true;
        gs->xok[k] = !memcmp(gs->xplanes[k], view->main.frustum_planes, sizeof(gs->xplanes[k]));
        if (!gs->xok[k]) {
            clapgpu_frustum fr;
            frustum_of(view, &fr);
            gs->stats.cull_launches_after_update++;
            if (!clapgpu_scene_cull_view(gs->scene, (uint32_t)gs->xslot[k], &fr)) {
                memcpy(gs->xplanes[k], view->main.frustum_planes, sizeof(gs->xplanes[k]));
                gs->xok[k] = true;
                consume_fetched(gs);
            }
        }
    }
    return (gs->xok[k] && (uint32_t)gs->xslot[k] < gs->res.n_views) ? gs->res.view_mask[gs->xslot[k]] : NULL;
}

bool gpu_view_entity_in_frustum(struct gpu_scene *gs, struct view *view, entity3d *e)
{
    const uint64_t *mask = gs ? mask_for_view(gs, view) : NULL;
    if (mask) {
        if (gs->notify && gs->vis_cursor < gs->n_order && gs->vq_e[gs->vis_cursor] == e) {
            /* notification mode, asked in list order (model.c:958-973): the table answers */
            const uint32_t c = gs->vis_cursor;
            gs->vis_cursor = c + 1 < gs->n_order ? c + 1 : 0;
            if (gs->vq_ok[c])
                return (mask[gs->vq_slot[c] >> 6] >> (gs->vq_slot[c] & 63)) & 1;
        } else {
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
            if (r && r->gen == gs->gen && r->cls == 1) {
                /* with notifications an untouched record's flags ARE the entity's: the 448-byte struct is not read at all */
                const uint32_t fl = (gs->notify && !r->pending) ? r->flags : (e->flags & (ENTITY3D_ALIVE | 0xffffu));
                if (fl == r->flags &&
                    (fl & (ENTITY3D_ALIVE | ENTITY3D_VISIBLE | ENTITY3D_SKIP_CULLING)) == (ENTITY3D_ALIVE | ENTITY3D_VISIBLE))
                    return (mask[r->slot >> 6] >> (r->slot & 63)) & 1;
            }
        }
    }
    /* the reference's test reads e->aabb: under GPU_SCATTER_DRAWN an entity nobody draws (hidden, or asked about out of
     * turn) may not have been shown its latest box yet */
    if (gs && gs->any_pend) gpu_scene_fetch(gs, e);
    return view_entity_in_frustum(view, e);
}

/*
 * _models_render's per-entity block (model.c:959-992) for one entity on the host -- the engine's own predicates,
 * entity3d_aabb_avg_edge and entity3d_set_lod around the five lines of glue between them -- for the entities the
 * device does not hold (foreign hooks, physics bodies, ...).  Returns whether the pass draws the entity.
 */
static bool lod_pick_host(struct view *view, entity3d *e, const float *cam_pos)
{
    if (!entity3d_matches(e, ENTITY3D_ALIVE) || !entity3d_matches(e, ENTITY3D_VISIBLE))
        return false;
    if (!entity3d_matches(e, ENTITY3D_SKIP_CULLING) && view && !view_entity_in_frustum(view, e))
        return false;
    if (cam_pos) {
        if (e->force_lod >= 0) {
            e->cur_lod = e->force_lod;
        } else if (!aabb_point_is_inside(e->aabb, cam_pos)) {       /* only when the camera is outside the box */
            vec3 dist;
            vec3_sub(dist, e->aabb_center, cam_pos);
            const float side = entity3d_aabb_avg_edge(e);
            const float scale = fabsf(vec3_mul_inner(dist, dist) - side * side) / 3600.0;
            entity3d_set_lod(e, (int)scale, false);
        }
    }
    return true;
}

static int draw_push(struct gpu_scene *gs, entity3d *e, int lod, uint32_t txm)
{
    if (gs->n_draw == gs->cap_draw) {
        const uint32_t cap = gs->cap_draw ? 2 * gs->cap_draw : 4096;
        entity3d **d = realloc(gs->draw, (size_t)cap * sizeof(*d));
        if (!d) return _CERR_NOMEM;
        gs->draw = d;
        int32_t *l = realloc(gs->draw_lod, (size_t)cap * sizeof(*l));
        if (!l) return _CERR_NOMEM;
        gs->draw_lod = l;
        uint16_t *t = realloc(gs->draw_txm, (size_t)cap * sizeof(*t));
        if (!t) return _CERR_NOMEM;
        gs->draw_txm = t;
        gs->cap_draw = cap;
    }
    if (txm == 0xffffffffu && (txm = txm_index(gs, e->txmodel)) == 0xffffffffu) return _CERR_NOMEM;
    gs->draw[gs->n_draw] = e;
    gs->draw_txm[gs->n_draw] = (uint16_t)txm;
    gs->draw_lod[gs->n_draw++] = lod;
    return 0;
}

static int draw_reserve(struct gpu_scene *gs, uint32_t n)
{
    if (n <= gs->cap_draw) return 0;
    uint32_t cap = gs->cap_draw ? gs->cap_draw : 4096;
    while (cap < n) cap *= 2;
    entity3d **d = realloc(gs->draw, (size_t)cap * sizeof(*d));
    if (d) gs->draw = d;
    int32_t *l = realloc(gs->draw_lod, (size_t)cap * sizeof(*l));
    if (l) gs->draw_lod = l;
    uint16_t *t = realloc(gs->draw_txm, (size_t)cap * sizeof(*t));
    if (t) gs->draw_txm = t;
    if (!d || !l || !t) return _CERR_NOMEM;
    gs->cap_draw = cap;
    return 0;
}

/* entries [lo, hi) of the device's draw list into the binding's (gpu_scene_select_lod, a list too long for one thread) */
struct draw_ctx { struct gpu_scene *gs; const clapgpu_scene_arrays *res; const uint32_t *slots; const int32_t *lods; uint32_t holes; };
static void draw_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct draw_ctx *dc = ctx;
    struct gpu_scene *gs = dc->gs;
    uint32_t holes = 0;
    for (uint32_t k = lo; k < hi; k++) {
        const uint32_t slot = dc->slots[k];
        const int32_t lod = dc->lods[k];
        entity3d *e = gs->slot_ent[slot];
        if (e && lod != gs->slot_lod[slot] && lod >= -128 && lod <= 127) {
            e->cur_lod = lod;                                    /* as model.c:977 / entity3d_set_lod leave it */
            gs->slot_lod[slot] = (int8_t)lod;
            const uint32_t tag = (uint32_t)(uintptr_t)dc->res->slot_user[slot];
            if (tag) gs->rec[tag - 1].lod_cur = lod;
            clapgpu_scene_lod_picked(gs->scene, slot, lod);
        }
        gs->draw[k] = e; gs->draw_lod[k] = lod; gs->draw_txm[k] = e ? gs->slot_txm[slot] : 0;
        holes += !e;
    }
    if (holes) __atomic_fetch_add(&dc->holes, holes, __ATOMIC_RELAXED);
}

void gpu_scene_lod_changed(struct gpu_scene *gs, entity3d *e)
{
    if (!gs || !e) return;
    const uint32_t i = rec_find(gs, e);
    if (i == NO_REC) return;
    struct gs_rec *r = &gs->rec[i];
    if (r->handle == CLAPGPU_NO_ENTITY || (e->force_lod == r->lod_force && e->cur_lod == r->lod_cur)) return;
    if (!clapgpu_scene_entity_lod(gs->scene, r->handle, e->force_lod, e->cur_lod)) {
        r->lod_force = e->force_lod; r->lod_cur = e->cur_lod;
        if (r->slot < gs->cap_slot_arrays) {
            if (e->cur_lod >= -128 && e->cur_lod <= 127) gs->slot_lod[r->slot] = (int8_t)e->cur_lod;
            else gs->cap_slot_arrays = 0;                        /* out of the byte's range: the record path */
        }
    }
}

int gpu_scene_select_lod(struct gpu_scene *gs, struct view *view, const float *cam_pos)
{
    if (!gs) return _CERR_INVALID_ARGUMENTS;
    gs->n_draw = 0;
    gs->groups_valid = false;
    /* entities came or went since the frame's update (notification mode knows): the list would miss what the reference's
     * walk of the txmodels draws -- this pass is the reference's */
    if (gs->notify && (gs->topology_pending || gs->n_created)) return _CERR_NOT_SUPPORTED;
    /* the mask that lists what the pass draws: the main view's or a registered view's own, re-culled if its planes moved; a
     * view the last update did not know takes the main view's place (one cull launch, and the main mask is its from now on) */
    uint32_t of_view = CLAPGPU_SCENE_MAIN_VIEW;
    if (view) {
        const int xk = view != gs->culled_view ? xview_of(gs, view) : -1;
        if (xk >= 0) {
            if (!mask_for_view(gs, view)) return _CERR_NOT_SUPPORTED;
            of_view = (uint32_t)gs->xslot[xk];
        } else if (view != gs->culled_view) {
            clapgpu_frustum fr;
            frustum_of(view, &fr);
            gs->stats.cull_launches_after_update++;
            CK(clapgpu_scene_cull(gs->scene, &fr));
            consume_fetched(gs);                                 /* GPU_SCATTER_DRAWN: what the new planes bring into view */
            memcpy(gs->culled_planes, view->main.frustum_planes, sizeof(gs->culled_planes));
            gs->culled_view = view;
            gs->cull_checked = gs->cull_ok = true;
        } else if (!mask_for_view(gs, view)) {
            return _CERR_NOT_SUPPORTED;
        }
    }
    /* models whose LOD range moved since they were registered (model3d's mesh LODs are added at load time) */
    for (uint32_t k = 0; k < gs->n_models; k++) {
        struct gs_model *gm = &gs->models[k];
        if (gm->lod_min != gm->model->lod_min || gm->lod_max != gm->model->lod_max) {
            CK(clapgpu_scene_model_lods(gs->scene, gm->handle, gm->model->lod_min, gm->model->lod_max));
            gm->lod_min = gm->model->lod_min; gm->lod_max = gm->model->lod_max;
        }
    }
    /* a frame that is walked re-reads every batched entity's force_lod / cur_lod anyway (mirror()); in notification
     * mode the engine's entity3d_set_lod reports them (gpu-exports.inc.c -> gpu_scene_lod_changed) */
    uint32_t n = 0;
    bool device_ok = true;                                       /* false: no update has run on the device yet -- everything below by the host block */
    if (!view) {
        /* A pass without a view draws every ALIVE and VISIBLE entity (model.c:969-970: `view && !view_entity_in_frustum`);
         * the device's mask answers for the frustum of the last update, not for "no frustum": the reference's own block
         * for every entity, batched ones included (their mirrored LODs follow below), on current host fields */
        CK(gpu_scene_fetch_all(gs));
        device_ok = false;
    } else
    if (gs->n_batched) {
        const int rc = clapgpu_scene_select_lod_view(gs->scene, of_view, cam_pos, &n);
        if (rc && rc != CLAPGPU_ERR_NOT_SUPPORTED) return rc;
        device_ok = !rc;
    }
    clapgpu_scene_arrays res;
    const uint32_t *slots = NULL; const int32_t *lods = NULL;
    if (n && !clapgpu_scene_results(gs->scene, &res) && clapgpu_scene_draw_list(gs->scene, &slots, &lods) == n) {
        const bool by_slot = gs->cap_slot_arrays >= res.n_slots;
        if (by_slot && cam_pos && n >= GS_MIRROR_PAR_MIN && par_threads() > 1 && !draw_reserve(gs, n)) {
            /* a long list: the gather on the workers (entry k -> draw[k]: nothing shared but the arrays) */
            struct draw_ctx dc = { gs, &res, slots, lods, 0 };
            gpu_scene_par_for(draw_range, &dc, n, par_threads());
            gs->n_draw = n;
            if (dc.holes) {                                      /* lanes vacated since the update (gpu_scene_entity_deleting): out */
                uint32_t w = 0;
                for (uint32_t k = 0; k < n; k++) {
                    if (!gs->draw[k]) continue;
                    gs->draw[w] = gs->draw[k]; gs->draw_lod[w] = gs->draw_lod[k]; gs->draw_txm[w] = gs->draw_txm[k];
                    w++;
                }
                gs->n_draw = w;
            }
        } else
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t slot = slots[k];
            if (by_slot) {
                /* three arrays read in ascending slot order; an entity3d (and its record) only when the pick changed its LOD */
                entity3d *e = gs->slot_ent[slot];
                if (!e) continue;
                if (cam_pos && lods[k] != gs->slot_lod[slot] && lods[k] >= -128 && lods[k] <= 127) {
                    e->cur_lod = lods[k];                           /* as model.c:977 / entity3d_set_lod leave it */
                    gs->slot_lod[slot] = (int8_t)lods[k];
                    const uint32_t tag = (uint32_t)(uintptr_t)res.slot_user[slot];
                    if (tag) gs->rec[tag - 1].lod_cur = lods[k];
                    clapgpu_scene_lod_picked(gs->scene, slot, lods[k]);
                }
                CK(draw_push(gs, e, lods[k], gs->slot_txm[slot]));
                continue;
            }
            const uint32_t tag = (uint32_t)(uintptr_t)res.slot_user[slot];
            if (!tag) continue;
            struct gs_rec *r = &gs->rec[tag - 1];
            if (!r->e || r->gen != gs->gen || (r->cls != 1 && r->cls != 4)) continue;
            if (cam_pos && r->lod_cur != lods[k]) clapgpu_scene_lod_picked(gs->scene, slot, lods[k]);
            r->e->cur_lod = lods[k];                                /* as model.c:977 / entity3d_set_lod leave it */
            r->lod_cur = lods[k];
            CK(draw_push(gs, r->e, lods[k], 0xffffffffu));
        }
    }
    /* the entities the device does not hold, in list order, by the reference's own block: the host-class ones -- the two
     * lists the walk keeps of them (own hook now / behind the pose), merged by their place in the queue; NOT a scan of every
     * record for the few that are not batched (1 M records: 3-4 ms of a 5 ms call) */
    if (device_ok) {
        uint32_t a = 0, b = 0;
        while (a < gs->n_host || b < gs->n_deferred) {
            const uint64_t ka = a < gs->n_host ? gs->rec[gs->host_list[a]].order_key : UINT64_MAX;
            const uint64_t kb = b < gs->n_deferred ? gs->rec[gs->deferred[b]].order_key : UINT64_MAX;
            struct gs_rec *r = ka <= kb ? &gs->rec[gs->host_list[a++]] : &gs->rec[gs->deferred[b++]];
            if (!r->e || r->cls == 1 || r->cls == 4) continue;
            if (lod_pick_host(view, r->e, cam_pos))
                CK(draw_push(gs, r->e, r->e->cur_lod, 0xffffffffu));
        }
        return 0;
    }
    for (uint32_t k = 0; k < gs->n_order; k++) {
        struct gs_rec *r = &gs->rec[gs->order[k]];
        if (!r->e) continue;
        if (lod_pick_host(view, r->e, cam_pos))
            CK(draw_push(gs, r->e, r->e->cur_lod, 0xffffffffu));
        if ((r->cls == 1 || r->cls == 4) && r->handle != CLAPGPU_NO_ENTITY && r->e->cur_lod != r->lod_cur &&
            !clapgpu_scene_entity_lod(gs->scene, r->handle, r->e->force_lod, r->e->cur_lod)) {
            r->lod_force = r->e->force_lod; r->lod_cur = r->e->cur_lod;   /* the host block picked for a batched entity: the mirror follows */
            if (r->slot < gs->cap_slot_arrays && r->e->cur_lod >= -128 && r->e->cur_lod <= 127) gs->slot_lod[r->slot] = (int8_t)r->e->cur_lod;
        }
    }
    return 0;
}

uint32_t gpu_scene_visible(struct gpu_scene *gs, entity3d ***ents, const int32_t **lods)
{
    if (!gs) return 0;
    if (ents) *ents = gs->draw;
    if (lods) *lods = gs->draw_lod;
    return gs->n_draw;
}

/* the draw list grouped by txmodel (a stable counting sort, once per gpu_scene_select_lod and only when asked for) */
static int draw_group(struct gpu_scene *gs)
{
    if (gs->groups_valid) return 0;
    gs->n_groups = 0;
    if (gs->n_draw > gs->cap_draw_g) {
        entity3d **d = realloc(gs->draw_g, (size_t)gs->cap_draw * sizeof(*d));
        if (!d) return _CERR_NOMEM;
        gs->draw_g = d;
        int32_t *l = realloc(gs->draw_g_lod, (size_t)gs->cap_draw * sizeof(*l));
        if (!l) return _CERR_NOMEM;
        gs->draw_g_lod = l;
        gs->cap_draw_g = gs->cap_draw;
    }
    /* a stable counting sort over the entries' txmodel indices: the entities themselves are not read */
    if (gs->n_txms > gs->cap_groups) {
        struct gs_draw_group *q = realloc(gs->groups, (size_t)gs->n_txms * sizeof(*q));
        if (!q) return _CERR_NOMEM;
        gs->groups = q; gs->cap_groups = gs->n_txms;
    }
    gs->n_groups = gs->n_txms;
    for (uint32_t g = 0; g < gs->n_groups; g++) gs->groups[g] = (struct gs_draw_group){ .txm = gs->txms[g], .start = 0, .n = 0 };
    for (uint32_t k = 0; k < gs->n_draw; k++) gs->groups[gs->draw_txm[k]].n++;
    uint32_t at = 0;
    for (uint32_t g = 0; g < gs->n_groups; g++) { gs->groups[g].start = at; at += gs->groups[g].n; gs->groups[g].n = 0; }
    for (uint32_t k = 0; k < gs->n_draw; k++) {
        struct gs_draw_group *grp = &gs->groups[gs->draw_txm[k]];
        const uint32_t pos = grp->start + grp->n++;
        gs->draw_g[pos] = gs->draw[k];
        gs->draw_g_lod[pos] = gs->draw_lod[k];
    }
    gs->groups_valid = true;
    return 0;
}

uint32_t gpu_scene_visible_of(struct gpu_scene *gs, const model3dtx *txm, entity3d ***ents, const int32_t **lods)
{
    if (!gs || !txm || draw_group(gs)) return 0;
    for (uint32_t g = 0; g < gs->n_groups; g++)
        if (gs->groups[g].txm == txm) {
            if (ents) *ents = gs->draw_g + gs->groups[g].start;
            if (lods) *lods = gs->draw_g_lod + gs->groups[g].start;
            return gs->groups[g].n;
        }
    return 0;
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
