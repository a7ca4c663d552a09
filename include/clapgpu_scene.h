/*
 * clapgpu_scene.h -- C host mirror of a CLAP model queue over libclapgpu (libclapgpu_scene.so).
 *
 * This is the host side a CLAP maintainer links instead of walking `mq->txmodels` ->
 * `txm->entities` (model.h:334,222,377): entities are registered once, mutated through the
 * same verbs the engine uses (entity3d_position / _rotate / _scale / _visible, model.c:1810-1842;
 * e->parent, model.h:402), and clapgpu_scene_mq_update() is `mq_update(mq)` (model.c:1953) +
 * the per-entity view_entity_in_frustum() of _models_render (model.c:969-970) in one call:
 * re-tile if the topology changed, upload dirty transforms, run the HIP kernel, download
 * mx / inverse_mx / aabb / aabb_center / visibility, all visible on return (the reference's
 * calls are synchronous, single-threaded).
 *
 * Plain C.  Returns cerr_enum-compatible ints (error.h:12-49).  Handles are stable until
 * clapgpu_scene_entity_delete(); slots (device indices) are not.
 */
#ifndef CLAPGPU_SCENE_H
#define CLAPGPU_SCENE_H

#include "clapgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct clapgpu_scene clapgpu_scene;
#define CLAPGPU_NO_ENTITY 0xffffffffu

/* Scenes of at most this many slots (64-entity rows, padding included) run without copy calls: the mirror's upload
 * image and result arrays are page-locked, device-mapped memory; a frame's touched inputs are read from the image by the
 * update kernel itself, what it rebuilds is written to the result arrays beside the device's own, and mq_update waits
 * on a polled completion word -- one launch, no upload copy, no download copy, no blocking runtime wait
 * (clapgpu_entities_update_tiles_hostio).  Larger scenes stage through device slabs (one copy up, the whole result slab
 * down).  Measured at 10 % of the entities moving, device step staged / mapped: 30 k entities 0.32 / 0.08 ms, 100 k 0.98 /
 * 0.31 ms, 1 M 8.8 / 3.0 ms; all moving, 1 M: 9.3 / 8.8 ms -- the mapped form moves only what changed and is never the
 * slower one, so by default every scene takes it; the staged form stays for hosts where mapped memory is not wanted. */
#define CLAPGPU_SCENE_ZERO_COPY_SLOTS 0xffffffffu
/* up to this many slots clapgpu_scene_select_lod's draw list is written by the device straight into mapped host memory */
#define CLAPGPU_SCENE_LOD_MAPPED_SLOTS 131072u
int  clapgpu_scene_create(clapgpu_scene **out, int device);
void clapgpu_scene_destroy(clapgpu_scene *s);
/* 0 = always stage through device slabs; takes effect at the next layout rebuild (the call forces one).  The
 * environment variable CLAPGPU_SCENE_ZERO_COPY_SLOTS overrides the default at create time. */
void clapgpu_scene_set_zero_copy_slots(clapgpu_scene *s, uint32_t max_slots);
int  clapgpu_scene_is_zero_copy(const clapgpu_scene *s);

/* model3d: local-space AABB (model3d.aabb, model.h:55) and skip_aabb (model.h:64) */
int  clapgpu_scene_model_new(clapgpu_scene *s, const float aabb[6], int skip_aabb, uint32_t *model);

/* entity3d_make (model.c:1735-1744): identity transform, scale 1, ALIVE | VISIBLE, dirty */
int  clapgpu_scene_entity_new(clapgpu_scene *s, uint32_t model, void *user, uint32_t *handle);
/* entity3d_delete (model.c:1787): the slot becomes a tombstone until the next re-tile */
int  clapgpu_scene_entity_delete(clapgpu_scene *s, uint32_t handle);
/* The two verbs above and set_parent change the queue's make-up: the next mq_update lays the entities out anew (every slot
 * moves).  A queue that gains and loses a few entities a frame uses these instead: the STANDING tile layout is edited --
 * a new root takes a free first-row lane (of a growth tile appended behind the others if need be), a new child a free lane
 * of the row below its parent in the parent's tile, a deleted leaf's lane stops being ALIVE -- and nothing else moves:
 * slots, masks and a caller's per-slot state stand, layout_generation does not advance, slot counts may grow (at the end).
 * CLAPGPU_ERR_NOT_SUPPORTED = it does not fit (no free lane or row, capacity, an entity with children or riding a joint, not
 * the one-launch tile form, nothing on the device yet) and NOTHING was changed: use the plain verbs then.
 * clapgpu_scene_set_incremental(s, 1) makes re-tiles leave room for such edits (an eighth of every row's lanes, one spare row
 * per tile of a hierarchy).  The edits reach the device with the next mq_update. */
void clapgpu_scene_set_incremental(clapgpu_scene *s, int on);

/*
 * A re-tile (any frame after clapgpu_scene_entity_new / _delete / _set_parent that could not be placed in the standing
 * layout) walks every handle and every slot several times: depths, tree widths, the upload image.  The mirror owns no
 * threads; a caller that has a pool lends it here -- `fn(range, ctx, n, threads)` must call range(ctx, lo, hi) over a
 * partition of [0, n) and return when all of them have (gpu-scene.c passes gpu_scene_par_for).  NULL / threads < 2: on the
 * calling thread.  The layout does not depend on it (same slots either way).
 */
typedef void (*clapgpu_scene_parallel_for)(void (*range)(void *ctx, uint32_t lo, uint32_t hi), void *ctx, uint32_t n, int threads);
void clapgpu_scene_set_parallel_for(clapgpu_scene *s, clapgpu_scene_parallel_for fn, int threads);
int  clapgpu_scene_entity_new_placed(clapgpu_scene *s, uint32_t model, void *user, uint32_t parent, uint32_t *handle, uint32_t *slot);
int  clapgpu_scene_entity_delete_placed(clapgpu_scene *s, uint32_t handle);
/* e->parent = p (jointless attachment, model.h:386-402); CLAPGPU_NO_ENTITY detaches */
int  clapgpu_scene_entity_set_parent(clapgpu_scene *s, uint32_t handle, uint32_t parent);

/* entity3d_position / transform_set_quat / entity3d_scale / entity3d_visible: set xform.updated */
int  clapgpu_scene_entity_position(clapgpu_scene *s, uint32_t handle, const float pos[3]);
int  clapgpu_scene_entity_rotation(clapgpu_scene *s, uint32_t handle, const float quat_xyzw[4]);
int  clapgpu_scene_entity_scale(clapgpu_scene *s, uint32_t handle, float scale);
/* the three above in one call: what a binding pushes when it finds xform.updated set */
int  clapgpu_scene_entity_transform(clapgpu_scene *s, uint32_t handle, const float pos[3],
                                    const float quat_xyzw[4], float scale);
/* the same + the entity's flags, for a binding that mirrors many DIFFERENT entities from several threads at once (no
 * creation / deletion / re-parenting meanwhile); finish with clapgpu_scene_mark_all_dirty() */
int  clapgpu_scene_entity_transform_mt(clapgpu_scene *s, uint32_t handle, const float pos[3], const float quat_xyzw[4],
                                       float scale, uint32_t flags, int xform_updated);
int  clapgpu_scene_entity_xform_mt(clapgpu_scene *s, uint32_t handle, const float pos[3], const float quat_xyzw[4],
                                   float scale, int xform_updated);   /* the transform alone; xform_updated is OR-ed in */
void clapgpu_scene_entity_xform_prefetch(const clapgpu_scene *s, uint32_t handle, uint32_t slot);   /* the lines the call above writes */
void clapgpu_scene_mark_all_dirty(clapgpu_scene *s);
/* entity3d_move / entity3d_rotate (radians) / entity3d_visible (model.c:1810-1842) */
int  clapgpu_scene_entity_move(clapgpu_scene *s, uint32_t handle, const float off[3]);
int  clapgpu_scene_entity_rotate(clapgpu_scene *s, uint32_t handle, float rx, float ry, float rz);
int  clapgpu_scene_entity_visible(clapgpu_scene *s, uint32_t handle, unsigned int visible);
/* transform_set_angles' quaternion (transform.c:62-73): angle clamp, optional degrees, euler xyz -> (x,y,z,w) */
void clapgpu_quat_from_angles(const float angles[3], int degrees, float quat_xyzw[4]);
int  clapgpu_scene_entity_flags(clapgpu_scene *s, uint32_t handle, uint32_t set, uint32_t clear);

/* mq_update + cull against `frustum` (NULL: no cull) */
int  clapgpu_scene_mq_update(clapgpu_scene *s, const clapgpu_frustum *frustum);
/*
 * Joint attachments (e->parent_joint != JOINT_TYPE_MAX: parent_transform_apply's second flavour, model.c:1626-1641).
 * clapgpu_scene_entity_set_attach() marks an entity (which has a parent) as riding one of its parent's joints;
 * clapgpu_scene_mq_update() then computes everything but the attached subtrees' final matrices, the caller runs the
 * frame's pose, and clapgpu_scene_attached_update() -- the frame's SECOND entity launch -- rebuilds each attached entity
 * as parent.mx * ((jt[k] * bind[k]) * local) with jt[k] = its parent's joint_transforms[parent_joint] of this frame and
 * bind[k] = that joint's bind matrix (16 floats each, column-major), and everything below it.  Culls against the
 * frustum of the preceding mq_update; afterwards the result arrays hold the rebuilt rows, rebuilt_mask names them.
 */
int  clapgpu_scene_entity_set_attach(clapgpu_scene *s, uint32_t handle, int attached);
int  clapgpu_scene_attached_update(clapgpu_scene *s, uint32_t n, const uint32_t *handles, const float *jt, const float *bind);

/* the cull alone, against another frustum (scene_cameras_calc recomputes the frusta after mq_update, clap.c:614-616):
 * refreshes the visibility results below; CLAPGPU_ERR_NOT_SUPPORTED before the first mq_update */
int  clapgpu_scene_cull(clapgpu_scene *s, const clapgpu_frustum *frustum);
/*
 * The frame's OTHER views.  pipeline_render() runs the shadow passes -- one per cascade, view = &light->view[0], no camera
 * (pipeline-builder.c:34-46, 246-272; model.c:752-760) -- before the model pass with the camera's view, and each pass asks
 * view_entity_in_frustum(view, e) for every entity (model.c:966-973): two frusta a frame.  The n_extra frusta given here
 * (up to CLAPGPU_EXTRA_VIEWS_MAX; 0 takes them off) are culled by the SAME launch as the main frustum of every following
 * clapgpu_scene_mq_update(frustum != NULL) and clapgpu_scene_cull(): clapgpu_scene_arrays.view_mask[v] answers for view v
 * without a launch of its own, clapgpu_scene_select_lod_view(v) lists what it draws, and under CLAPGPU_SCENE_EXPORT_DRAWN
 * an entity drawn by ANY of the views counts as read.  clapgpu_scene_cull_view(v) re-tests one extra view alone (its planes
 * moved after the update).
 */
int  clapgpu_scene_set_views(clapgpu_scene *s, uint32_t n_extra, const clapgpu_frustum *extra);
int  clapgpu_scene_cull_view(clapgpu_scene *s, uint32_t view, const clapgpu_frustum *frustum);

/* results of the last mq_update; pointers stay valid until the next mq_update */
const float *clapgpu_scene_entity_mx(const clapgpu_scene *s, uint32_t handle);          /* e->mx */
const float *clapgpu_scene_entity_inverse_mx(const clapgpu_scene *s, uint32_t handle);  /* e->inverse_mx */
const float *clapgpu_scene_entity_aabb(const clapgpu_scene *s, uint32_t handle);        /* e->aabb (6) */
const float *clapgpu_scene_entity_aabb_center(const clapgpu_scene *s, uint32_t handle); /* e->aabb_center */
int          clapgpu_scene_entity_in_frustum(const clapgpu_scene *s, uint32_t handle);  /* view_entity_in_frustum */
void        *clapgpu_scene_entity_user(const clapgpu_scene *s, uint32_t handle);
/* number of entities that pass the draw predicate; handles[] (ascending slot order) if non-NULL */
uint32_t     clapgpu_scene_visible(const clapgpu_scene *s, uint32_t *handles, uint32_t capacity);
/*
 * Bulk form of the accessors above for a binding that scatters a whole frame back into entity3d
 * structs: the page-locked result arrays of the last mq_update, indexed by SLOT, and an entity's
 * slot.  Slots change only when the layout is rebuilt (creation, deletion, re-parenting); the
 * pointers stay valid until the next mq_update.
 */
typedef struct clapgpu_scene_arrays {
    uint32_t        n_slots;
    const float    *mx;             /* [n_slots][16] */
    const float    *inverse_mx;     /* [n_slots][16] */
    const float    *aabb;           /* [n_slots][6]  */
    const float    *aabb_center;    /* [n_slots][3]  */
    const uint64_t *vis_mask;       /* bit (slot & 63) of word slot >> 6 */
    const uint64_t *rebuilt_mask;   /* same indexing: the last mq_update rebuilt this slot (mx, inverse_mx, aabb, seq changed):
                                       all a binding has to copy back.  Parents sit in lower slots than their children. */
    const uint64_t *inside_mask;    /* same indexing: the slot's box contains a bounding-volume point (clapgpu_scene_set_bv_points) */
    void *const    *slot_user;      /* [n_slots] the `user` pointer given to clapgpu_scene_entity_new, NULL for padding slots */
    /* export policy (clapgpu_scene_set_export), same indexing; without EXPORT_DRAWN exported == rebuilt, nothing is stale */
    const uint64_t *exported_mask;  /* the rows of the arrays above the last mq_update / attached_update wrote */
    const uint64_t *stale_mask;     /* rows rebuilt on the device since they were last written here: the arrays hold older values */
    const uint64_t *fetched_mask;   /* rows the last mq_update / cull / fetch brought over because somebody reads them now */
    uint32_t        n_stale_words, n_fetched;   /* non-zero words of stale_mask; rows in fetched_mask */
    uint32_t        fetch_serial;   /* advances with every call that fetched something: fetched_mask is to be copied out once per value */
    /* the frame's other views (clapgpu_scene_set_views): view_mask[v] as vis_mask, for extra view v of the last mq_update / cull */
    uint32_t        n_views;
    const uint64_t *view_mask[CLAPGPU_EXTRA_VIEWS_MAX];
} clapgpu_scene_arrays;
int          clapgpu_scene_results(const clapgpu_scene *s, clapgpu_scene_arrays *out);
uint32_t     clapgpu_scene_entity_slot(const clapgpu_scene *s, uint32_t handle);

/*
 * Export policy: which rebuilt rows a frame writes back to the host.  The draw path reads e->mx / e->inverse_mx of what it
 * DRAWS (model.c:1022-1028) and the LOD block e->aabb / e->aabb_center of the same entities (model.c:975-992); at a million
 * entities writing back every rebuilt row -- 164 bytes over PCIe and a scatter into a 448-byte entity3d each -- costs two
 * hundred times the kernel.  CLAPGPU_SCENE_EXPORT_DRAWN (one-launch frames only: zero-copy + tile layout; otherwise the
 * call is remembered and has no effect) writes back the rebuilt rows of
 *   - entities that pass the draw predicate against the frustum of this mq_update (all of them when it has none),
 *   - entities whose box contains a bounding-volume point (clapgpu_scene_set_bv_points),
 *   - entities with a standing host reader (clapgpu_scene_entity_keep),
 * marks the other rebuilt rows STALE (stale_mask) and, in the same call, fetches every row that is read now and was left
 * stale earlier (an entity that came into view): after clapgpu_scene_mq_update() and after clapgpu_scene_cull() every
 * drawn, containing or kept entity's rows are current; exported_mask | fetched_mask names the rows that changed.
 * clapgpu_scene_fetch(want) brings over the stale rows flagged in `want` (n_slots / 64 words; NULL: all of them) and
 * leaves them in fetched_mask; clapgpu_scene_fetch_entity() one entity's; the per-entity accessors above fetch by
 * themselves.  The default is CLAPGPU_SCENE_EXPORT_ALL: nothing changes unless asked for.
 */
#define CLAPGPU_SCENE_EXPORT_ALL   0
#define CLAPGPU_SCENE_EXPORT_DRAWN 1
void         clapgpu_scene_set_export(clapgpu_scene *s, int policy);
int          clapgpu_scene_export_is_drawn(const clapgpu_scene *s);     /* the policy is DRAWN and the layout supports it */
int          clapgpu_scene_entity_keep(clapgpu_scene *s, uint32_t handle, int keep);
int          clapgpu_scene_fetch(clapgpu_scene *s, const uint64_t *want, uint32_t *n_rows);
int          clapgpu_scene_fetch_entity(clapgpu_scene *s, uint32_t handle);

/*
 * default_update's camera bounding-volume pick (model.c:1703-1713): the points to test every entity's box against in
 * the next mq_update -- camera position, and the control entity's position (NULL: none; ctl_handle is that entity,
 * which never counts).  cam_pos == NULL switches the test off.  Result: clapgpu_scene_arrays.inside_mask.
 */
void         clapgpu_scene_set_bv_points(clapgpu_scene *s, const float cam_pos[3], const float *ctl_pos, uint32_t ctl_handle);

/* 1 = tiles (one launch), 0 = level-major (a tree wider than 64 at some level); after mq_update */
/*
 * The render passes' per-entity block of _models_render (model.c:959-992) as ONE call per pass: the entities that pass
 * the draw predicate against the view of the last clapgpu_scene_mq_update / _cull, in ascending slot order (the
 * reference's order is list order; the draw path binds per txmodel either way), and the LOD each is drawn with --
 *   force_lod >= 0: that LOD (model.c:976-977); else, unless the camera is inside the entity's box (model.c:982), 
 *   clamp((int)(| |aabb_center - cam|^2 - avg_edge^2 | / 3600), lod_min, lod_max) (entity3d_aabb_avg_edge model.c:1261-1264,
 *   entity3d_set_lod / model3d_validate_lod model.c:593-609, 63-66); inside the box the entity keeps its cur_lod
 * -- written to the entity's cur_lod as the reference writes e->cur_lod.  Entities that fail the predicate keep theirs.
 *   clapgpu_scene_model_lods    model3d.lod_min / .lod_max (model.h:55-56), 0 / 0 until set
 *   clapgpu_scene_entity_lod    the entity's force_lod (-1: none) and cur_lod as the engine holds them: after
 *                               entity3d_make (force_lod -1, cur_lod 0: the defaults here) only entity3d_set_lod
 *                               (model.c:593-609) changes them
 *   clapgpu_scene_select_lod    cam_pos NULL = a pass without a camera (model.c:974): the list alone, cur_lod untouched
 *   clapgpu_scene_draw_list     (slot, lod) of the last pick, n entries; slot_user[slot] (clapgpu_scene_results) is the entity
 */
int          clapgpu_scene_model_lods(clapgpu_scene *s, uint32_t model, unsigned int lod_min, unsigned int lod_max);
int          clapgpu_scene_entity_lod(clapgpu_scene *s, uint32_t handle, int force_lod, int cur_lod);
int          clapgpu_scene_entity_cur_lod(const clapgpu_scene *s, uint32_t handle);
int          clapgpu_scene_select_lod(clapgpu_scene *s, const float cam_pos[3], uint32_t *n_draw);
/* ... over the mask of extra view `view` (clapgpu_scene_set_views); CLAPGPU_SCENE_MAIN_VIEW = the call above */
#define CLAPGPU_SCENE_MAIN_VIEW 0xffffffffu
int          clapgpu_scene_select_lod_view(clapgpu_scene *s, uint32_t view, const float cam_pos[3], uint32_t *n_draw);
/* A caller that walks the draw list anyway and knows every entity's last LOD can take over the bookkeeping the call above does
 * per entry (is this entity's cur_lod still what the mirror holds?): clapgpu_scene_set_lod_sync(s, 1), then
 * clapgpu_scene_lod_picked(s, slot, lod) for each entry of a list picked WITH a camera whose LOD changed (distinct slots may
 * be reported from several threads at once). */
void         clapgpu_scene_set_lod_sync(clapgpu_scene *s, int by_caller);
void         clapgpu_scene_lod_picked(clapgpu_scene *s, uint32_t slot, int lod);
uint32_t     clapgpu_scene_draw_list(const clapgpu_scene *s, const uint32_t **slots, const int32_t **lods);

int          clapgpu_scene_layout_is_tiled(const clapgpu_scene *s);
uint32_t     clapgpu_scene_slot_count(const clapgpu_scene *s);
/* incremented every time mq_update rebuilds the layout: cached slots are stale when it changes */
uint32_t     clapgpu_scene_layout_generation(const clapgpu_scene *s);

#ifdef __cplusplus
}
#endif
#endif /* CLAPGPU_SCENE_H */
