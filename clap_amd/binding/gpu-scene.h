/*
 * gpu-scene.h -- the CLAP-side binding of libclapgpu: the translation unit a CLAP maintainer adds
 * to core/ (INTEGRATION.md section 2).  Unlike the rest of clap_amd/ it is written AGAINST THE
 * ENGINE'S OWN HEADERS (model.h, view.h, scene.h) and works on the engine's own objects:
 * `struct mq`, `model3dtx`, `entity3d`, `struct view`.  It needs the reference tree to compile
 * (-I <clap>/core) and is therefore built only where that tree exists, by oracle/ref/Makefile,
 * into the drop-in checker oracle/_ref/clap_dropin; libclapgpu_scene.so underneath has no such
 * dependency.
 *
 * What it replaces:
 *   mq_update(mq)                     model.h:342, model.c:1953   -> gpu_mq_update(gs, mq, view)
 *   view_entity_in_frustum(view, e)   view.h:37,   view.c:296-337 -> gpu_view_entity_in_frustum(gs, view, e)
 * Everything the draw path reads stays where it is: e->mx, e->inverse_mx, e->aabb, e->aabb_center,
 * e->seq / e->parent_seq, xform.updated, scene->camera->bv / bv_volume are written exactly as
 * default_update (model.c:1649-1723) writes them.
 */
#ifndef CLAP_GPU_SCENE_H
#define CLAP_GPU_SCENE_H

#include "model.h"
#include "view.h"

struct gpu_scene;

struct gpu_scene_stats {
    unsigned int batched;       /* entities whose update ran on the device this frame */
    unsigned int host;          /* entities whose own hook ran on the host (foreign hook, animated, physics, light, joint-attached) */
    unsigned int uploaded;      /* entities whose transform was pushed this frame */
    unsigned int written_back;  /* entities whose mx / inverse_mx / aabb were rebuilt this frame */
    unsigned int attached;      /* of `batched`: entities of joint-attached subtrees, updated by the frame's second launch
                                   (gpu_scene_run_deferred, behind the pose) */
    unsigned int attach_failures;
    unsigned int untouched_writes;  /* verification mode: batched entities found with xform.updated set that nobody had
                                       reported (a direct transform_* write without gpu_scene_touch) -- taken in this frame */
    unsigned int registered, deleted;
    int          replayed;      /* without notifications: the queue was found to be the one the last walk met (same entities, same
                                   order, same classes), so the frame went by the records -- every pass on the worker threads --
                                   instead of walking the lists */
    unsigned int placed, removed;   /* of those: entities created / deleted since the last frame that were put into / taken out of the
                                   standing device layout, without a walk of the queue (gpu_scene_entity_created / _deleting) */
    int          retiled;       /* the device layout was rebuilt (creation, deletion, re-parenting) */
    double       ms_walk, ms_mirror, ms_device, ms_scatter;   /* steps 1, 2+3, 4, 5 of gpu_mq_update() */
    unsigned int fetched;       /* GPU_SCATTER_DRAWN: entities brought over after the fact this frame (came into view, asked for) */
    unsigned int left_stale;    /* GPU_SCATTER_DRAWN: entities the device rebuilt this frame whose entity3d was not written */
    unsigned int device_errors; /* CUMULATIVE, process-wide: calls of the binding that failed on the device and were served by
                                   the engine's host path instead (gpu_scene_device_errors()) */
    unsigned int views_culled;  /* views the update's own launch culled: the one it was given + the registered ones (gpu_scene_add_view) */
    unsigned int cull_launches_after_update;   /* cull launches since (a view whose planes moved, a view that is not registered):
                                   0 in a frame whose passes use the views the update knew */
};

/*
 * `default_hook` is model.c's default_update (static there: the maintainer's patch passes it from
 * mq_init or drops the `static`).  Entities with any other hook stay on the host.
 * Returns a cerr_enum value (error.h:12-49): 0 or negative.
 */
int  gpu_scene_init(struct gpu_scene **out, int device, int (*default_hook)(entity3d *, void *));
void gpu_scene_done(struct gpu_scene *gs);

/*
 * One mq_update(): on return every ALIVE entity of `mq` has been updated -- batched ones by the HIP
 * kernel (results scattered back into the entity3d structs), the others by their own hook, both in
 * the queue's list order.  `view` (may be NULL) is culled against in the same launch; its result
 * serves gpu_view_entity_in_frustum() until the next call.  Synchronous, like the reference.
 */
int  gpu_mq_update(struct gpu_scene *gs, struct mq *mq, struct view *view);

/* view_entity_in_frustum(): bit lookup for batched entities tested against the view of the last
 * gpu_mq_update(); the engine's own function for anything else. */
bool gpu_view_entity_in_frustum(struct gpu_scene *gs, struct view *view, entity3d *e);

/*
 * The frame's OTHER views.  pipeline_render() runs one shadow pass per cascade with view = &light->view[0] and no camera
 * (pipeline-builder.c:34-46, 246-272; model.c:752-760) before the model pass with the camera's view, and every pass asks
 * view_entity_in_frustum(view, e) for every entity (model.c:966-973; the test reads view->main alone, view.c:296-337).
 * A view registered here (up to GPU_SCENE_EXTRA_VIEWS; the light's: gpu_scene_add_view(gs, &light->view[0])) is culled by
 * the SAME launch as the view gpu_mq_update() is given, into a mask of its own: gpu_view_entity_in_frustum() and
 * gpu_scene_select_lod() then answer for either view without a launch and without touching the other view's mask;
 * under GPU_SCATTER_DRAWN an entity drawn by ANY of the views is written back.  A registered view whose planes moved
 * since the update (light_update runs behind mq_update, scene.c:1166-1171) costs one cull launch of that view alone.
 * _CERR_TOO_LARGE beyond the maximum; registering the same view twice is harmless.
 */
#define GPU_SCENE_EXTRA_VIEWS 4
int  gpu_scene_add_view(struct gpu_scene *gs, struct view *view);
void gpu_scene_remove_view(struct gpu_scene *gs, struct view *view);

/* view_calc_frustum() ran for `view` (the engine recomputes its frusta in scene_cameras_calc, after mq_update,
 * clap.c:614-616): the first verdict asked for it afterwards compares the planes once and, if they differ from the ones
 * the update culled against, runs the cull kernel alone for the new ones.  Code that writes frustum_planes by any other
 * route has to call this too. */
void gpu_scene_view_changed(struct gpu_scene *gs, struct view *view);

/*
 * One render pass of _models_render over the bound queue (model.c:958-992) without its per-entity host loop: which
 * entities the pass draws -- ALIVE, VISIBLE and SKIP_CULLING or inside `view`'s frustum -- and the LOD each is drawn
 * with, written to e->cur_lod exactly as the reference's block writes it: force_lod wins (model.c:976-977); inside the
 * entity's box the LOD stays (model.c:982); otherwise entity3d_set_lod(e, (int)(|dist^2 - avg_edge^2| / 3600), false)
 * (model.c:983-990, entity3d_aabb_avg_edge model.c:1261-1264, the clamp of model.c:63-66, 593-609).  Batched entities:
 * two launches over the device's boxes (ordered visible list, LOD pick) and a write-back of the DRAWN entities only;
 * host-class entities: the reference's own functions, on the host.  cam_pos = transform_pos(&camera->xform, NULL), NULL
 * for a pass without a camera (model.c:974): the list alone.  `view` other than the one the last update culled
 * against, or with planes that moved since (scene_cameras_calc runs after mq_update), costs one more cull launch.
 * Call after gpu_mq_update() of the frame; entities created or destroyed between that update and the pass are not
 * known to the list: in notification mode the call then returns _CERR_NOT_SUPPORTED with an empty list and the pass
 * takes the reference's loop (a queue that is walked every frame has no way to notice: do not create entities between
 * update and render there).  gpu_scene_visible(): the draw list of the last call -- batched entities in
 * device order, then host-class entities in list order -- for _models_render to iterate instead of every entity3d of
 * every txmodel (it binds per txmodel: e->txmodel of each entry says which).
 * gpu_scene_lod_changed(): entity3d_set_lod() wrote e->force_lod / e->cur_lod outside a walked frame (the engine-side
 * export reports it by itself, gpu-exports.inc.c).
 */
int      gpu_scene_select_lod(struct gpu_scene *gs, struct view *view, const float *cam_pos);
uint32_t gpu_scene_visible(struct gpu_scene *gs, entity3d ***ents, const int32_t **lods);
/* ... and the part of it that belongs to one txmodel: what _models_render's loop body iterates, `txmodel` by `txmodel`
 * (model.c:899-958 bind program, material and buffers per txmodel; the entity loop follows at 958).  Grouped lazily, once
 * per gpu_scene_select_lod(); a txmodel with nothing drawn returns 0. */
uint32_t gpu_scene_visible_of(struct gpu_scene *gs, const model3dtx *txm, entity3d ***ents, const int32_t **lods);
void     gpu_scene_lod_changed(struct gpu_scene *gs, entity3d *e);

const struct gpu_scene_stats *gpu_scene_last_stats(const struct gpu_scene *gs);

/*
 * Write-back policy (notification mode; a frame that walks the queue writes everything back whatever the policy).
 *
 * GPU_SCATTER_ALL (default): after gpu_mq_update() every entity3d the reference would have rebuilt holds the reference's
 * mx / inverse_mx / aabb / aabb_center / seq / parent_seq, whether anybody reads them or not.
 *
 * GPU_SCATTER_DRAWN: a fast frame writes back what is READ -- the draw path reads e->mx / e->inverse_mx of the entities a
 * pass draws (model.c:1022-1028) and their aabb / aabb_center for the LOD (model.c:975-992); default_update's side
 * effects read a few more -- and leaves the rest on the device.  Written back when rebuilt:
 *   - entities that pass the draw predicate of the view the update culls against (all of them without a view),
 *   - entities whose box contains the camera / control position (the bounding-volume pick reads e->aabb),
 *   - entities with a standing host reader, found by the walk: batched parents of host-class children (their hooks read
 *     parent->mx / ->seq), light carriers, the control entity, characters, animated entities, joint riders,
 *     entities updated on the spot by entity3d_update / _reset, and whatever gpu_scene_keep() names.
 * An entity left out is STALE: its entity3d keeps older mx / inverse_mx / aabb / aabb_center / seq / parent_seq until
 *   - it comes into view: the update, gpu_view_entity_in_frustum()'s and gpu_scene_select_lod()'s re-cull fetch it
 *     before they return (everything a pass draws is current), or
 *   - somebody asks: gpu_scene_fetch(gs, e) / gpu_scene_fetch_all(gs) (code that reads e->mx of an entity it does not
 *     draw -- a gameplay query, a debugger -- calls one of them first; the engine-side exports do so for
 *     entity3d_update / entity3d_reset), or
 *   - the queue is walked (topology change): everything is fetched first.
 * After a fetch the entity3d equals the reference's bit for bit, seq counters included.  With the verification aid on
 * (gpu_scene_set_verify / GPU_SCENE_VERIFY) a stale entity's mx[0][0] is poisoned with a NaN so that a read nobody
 * announced shows on the screen instead of lagging a frame.  Also: environment GPU_SCENE_SCATTER=drawn at gpu_scene_init.
 */
enum { GPU_SCATTER_ALL = 0, GPU_SCATTER_DRAWN = 1 };
void gpu_scene_set_scatter(struct gpu_scene *gs, int policy);
int  gpu_scene_fetch(struct gpu_scene *gs, entity3d *e);       /* 0, or a negative cerr_enum value */
/* _CERR_NOT_SUPPORTED between a gpu_scene_topology() report and the next gpu_mq_update(): entities may have been deleted,
 * a record may name freed memory, and only the walk of that update finds out which (it fetches everything as it goes) */
int  gpu_scene_fetch_all(struct gpu_scene *gs);
void gpu_scene_keep(struct gpu_scene *gs, entity3d *e, bool keep);   /* a standing host reader of e exists (takes effect with the next walk or at once) */
bool gpu_scene_entity_is_stale(struct gpu_scene *gs, entity3d *e);
void gpu_scene_describe(struct gpu_scene *gs, entity3d *e, char *buf, size_t len);   /* e's record in one line (checkers' reports) */

/* A call of the binding failed (rc != 0) and the caller is about to take the engine's host path instead: counted for the
 * life of the process; the first failure of each `what` is reported on stderr with clapgpu_last_error() (the engine-side
 * exports, gpu-exports.inc.c, also put it through the engine's err()).  A dead device must not look like a slow frame. */
void     gpu_scene_device_error(const char *what, int rc);
unsigned gpu_scene_device_errors(void);

/*
 * Notification mode.  The reference's mutators already mark what they change (transform_set_updated behind
 * entity3d_position / _move / _rotate / _scale, model.c:1810-1842); with CONFIG_GPU_SCENE they also tell the binding
 * (gpu-exports.inc.c), and gpu_mq_update() stops walking every entity3d twice a frame: it costs
 * O(touched + rebuilt + host-class entities).  Whatever changes the queue's make-up -- entity3d creation / deletion,
 * a write to e->parent or e->update, a body / light / joint attachment -- is reported with gpu_scene_topology(); the
 * next update then walks the queue once, as without notifications.
 */
void gpu_scene_set_notify(struct gpu_scene *gs, bool on);
/* Without notifications a frame still need not chase the lists on one core: when the queue is found to be the one the last
 * walk met -- every entity's list successor is the next record's entity: one list node per entity, read on the worker
 * threads -- the frame goes by the records (every entity's flags, xform.updated, class inputs and LODs re-read on the
 * workers) and falls back to the walk when anything would be classified differently.  The check reads the entities the
 * last walk met: one that was FREED since must have been reported (the exported entity3d_delete / the line in entity3d_drop
 * do that; gpu_scene_topology() is enough).  Environment GPU_SCENE_REPLAY=0 keeps the serial walk for every frame. */
bool gpu_scene_last_was_fast(const struct gpu_scene *gs);      /* the last gpu_mq_update() did not walk the queue */
void gpu_scene_touch(struct gpu_scene *gs, entity3d *e);
/* ... its transform alone (what entity3d_position / _move / _rotate / _scale change): O(1), no look-up -- the address is
 * queued and resolved by the frame's mirror pass.  A write to e->flags is reported with gpu_scene_touch(). */
void gpu_scene_touch_xform(struct gpu_scene *gs, entity3d *e);
/* entity3d_update(e, data) / entity3d_reset(e) run e's update on the host, outside the frame loop (gpu-exports.inc.c):
 * _begin before the reference's body (shows e and its parent what the device has, notes e->seq), _updated after it */
void gpu_scene_host_update_begin(struct gpu_scene *gs, entity3d *e);
void gpu_scene_host_updated(struct gpu_scene *gs, entity3d *e);
void gpu_scene_topology(struct gpu_scene *gs);
/*
 * Creation and deletion without a walk.  A queue whose make-up changes by a few entities a frame (pickups, projectiles,
 * effects) would otherwise walk every entity3d and lay the device's tiles out anew each time -- at a million entities
 * more than the reference's whole update.  entity3d_make reports the new entity with gpu_scene_entity_created() (the last
 * line of model.c:1730-1762; whatever the game sets afterwards -- position, e->parent, a light -- is read when the next
 * gpu_mq_update() takes the entity in), entity3d_delete / entity3d_drop report gpu_scene_entity_deleting() BEFORE the
 * entity goes (model.c:1765-1791; gpu-exports.inc.c does it for entity3d_delete).  A plain entity -- default_update, no
 * skeleton, body or joint, without a parent or below a batched one that comes earlier in the list -- is then placed into
 * the standing layout (a free lane of its parent's tile, or a growth tile), a batched leaf nobody depends on is taken out
 * of it, and the frame stays O(touched).  Anything else -- a custom hook, a txmodel the queue has not seen, a parent with
 * no room below it, an entity with children -- falls back to gpu_scene_topology() by itself.  A queue is packed tight until
 * its first entity comes or goes: THAT frame is walked and re-tiled as before, and from then on the device layout keeps room
 * for such edits (an eighth of every row, a spare row per tile: ~10 % of a million-entity frame, nothing at ten thousand).  Both are no-ops for entities
 * of other queues.  Without notification mode they equal gpu_scene_topology().
 */
void gpu_scene_entity_created(struct gpu_scene *gs, entity3d *e);
void gpu_scene_entity_deleting(struct gpu_scene *gs, entity3d *e);
/* on by default; off (also: environment GPU_SCENE_INCREMENTAL=0 at gpu_scene_init): both calls above equal gpu_scene_topology() */
void gpu_scene_set_incremental(struct gpu_scene *gs, bool on);
/*
 * Verification aid for notification mode (also: environment GPU_SCENE_VERIFY=1 at gpu_scene_init): before every fast
 * frame, look at each batched entity once for a transform somebody wrote WITHOUT telling the binding -- the engine moves
 * entities past its entity3d_* mutators too (transform_set_angles in the inspector, scene.c:871,934) -- report the first
 * few on stderr, count them in stats.untouched_writes, and take them in this frame as if they had been touched.  One pass
 * over the records and a flag read per entity3d: meant for debug builds and for finding the call sites, not for release.
 */
void gpu_scene_set_verify(struct gpu_scene *gs, bool on);

/* The binding's worker threads (kept between frames, at most 31 beside the caller, 23 by default; GPU_SCENE_THREADS sets the total) for its other translation units:
 * fn(ctx, lo, hi) over [0, n) in `threads` contiguous ranges, the caller taking the first; returns when all are done. */
void gpu_scene_par_for(void (*fn)(void *, uint32_t, uint32_t), void *ctx, uint32_t n, int threads);
/* The pool behind gpu_scene_par_for is shared by every binding object of the process: whoever may call it holds a
 * reference from its _init to its _done (a gpu_scene does by itself); the last reference to go joins the workers. */
void gpu_scene_pool_ref(void);
void gpu_scene_pool_unref(void);
/* the scene, queue and view the engine-named entry points (mq_update, view_entity_in_frustum, ...) serve */
void gpu_scene_bind(struct gpu_scene *gs, struct mq *mq, struct view *view);
struct gpu_scene *gpu_scene_bound(void);
struct mq *gpu_scene_bound_mq(void);
struct view *gpu_scene_bound_view(void);

/*
 * Skeletal animation.  By default an entity whose model has animations stays on the host, because
 * default_update ends in animated_update (model.c:1715-1716).  When the pose is computed elsewhere
 * (gpu-anim.inc.c: gpu_anim_update() after gpu_mq_update()), say so: such entities' transforms are then
 * batched like any other and nothing runs animated_update for them here.
 */
void gpu_scene_animation_elsewhere(struct gpu_scene *gs, bool elsewhere);
/* With the pose elsewhere, an entity riding a parent's JOINT (e->parent_joint, model.c:1626-1641) needs the joint
 * transforms of the same frame, which exist only after gpu_anim_update(): gpu_mq_update() holds such entities and
 * everything below them back, and this call finishes them (gpu_anim_update() ends with it): subtrees riding a BATCHED
 * character's joint go through the frame's second entity launch (clapgpu_scene_attached_update: TRS -> (joint * bind) *
 * local -> parent.mx * ... -> inverse, box, cull on the device), the rest -- riders of host-class parents, nested riders,
 * riders that are animated themselves -- run their own hooks, in list order. */
void gpu_scene_run_deferred(struct gpu_scene *gs, struct mq *mq);
/*
 * Characters without a physics body (gpu-character.inc.c, #include'd at the end of character.c, calls this through
 * gpu_scene_bind_characters): is_plain(e, default_hook) says whether e's hook is character_update chained to
 * default_update with no body behind it; host_half(e, data) runs everything character_update does before the chained call
 * (character.c:583-609).  Such entities are batched: host half first, at their place in the list, then the device.
 */
void gpu_scene_characters(struct gpu_scene *gs, bool (*is_plain)(entity3d *, int (*)(entity3d *, void *)),
                          int (*host_half)(entity3d *, void *));
void gpu_scene_bind_characters(struct gpu_scene *gs);          /* gpu-character.inc.c */
/* true if `e` was updated on the device by the last gpu_mq_update() */
/* advances with every walk of the queue: between two equal values no entity has changed its class (batched or not) */
uint32_t gpu_scene_walk_generation(const struct gpu_scene *gs);
bool gpu_scene_entity_is_batched(struct gpu_scene *gs, entity3d *e);

/*
 * Dump the batched part of the queue, as mirrored by the last gpu_mq_update(), into a scene snapshot
 * (include/clapgpu_snapshot.h): entities.pos_scale / rot / parent / model / flags / seqs / n in list order
 * (parents by index, -1 for none), entities.model_aabb / model_skip, and -- if a view was culled against --
 * frustum.planes / frustum.corners.  `python bench.py --snapshot file` and clap_amd.snapshot.load_scene()
 * replay it through the kernels without the engine.  The writer is returned open so that the caller can
 * add arrays of its own (clapgpu_snapshot_add) before clapgpu_snapshot_finish().
 */
struct clapgpu_snapshot_writer;
int gpu_scene_snapshot_begin(struct gpu_scene *gs, const char *path, struct clapgpu_snapshot_writer **out);

#endif /* CLAP_GPU_SCENE_H */
