/*
 * clapgpu.h -- C ABI of libclapgpu.so: the MI355X (gfx950) batch replacement for
 * the per-frame scene-update hot path of virtuoso/clap.
 *
 * Plain C, plain pointers and sizes.  All `dev` pointers are HIP device pointers
 * (from clapgpu_malloc, hipMalloc, or any allocator sharing the HIP context, e.g.
 * a torch tensor's data_ptr()); `stream` is a hipStream_t passed as void* (NULL =
 * the default stream).  Every function returns a cerr_enum-compatible int
 * (reference core/error.h:12-49): 0 = OK, negative = error; launches are
 * asynchronous on `stream` unless stated otherwise.
 *
 * Each entry point names the reference interface it replaces (file:line relative
 * to the reference tree).  INTEGRATION.md shows the binding a CLAP maintainer adds.
 *
 * Conventions (identical to oracle/clap_oracle.h):
 *   mat4  = float[16], column-major, (col c,row r) at [4c+r]  (linmath.h `mat4x4 M`, M[c][r])
 *   quat  = (x,y,z,w)                                          (linmath.h:835-840)
 */
#ifndef CLAPGPU_H
#define CLAPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: the subset of cerr_enum (error.h:12-49) this library returns ---- */
#define CLAPGPU_OK                      0
#define CLAPGPU_ERR_NOMEM              -1   /* _CERR_NOMEM */
#define CLAPGPU_ERR_INVALID_ARGUMENTS  -2   /* _CERR_INVALID_ARGUMENTS */
#define CLAPGPU_ERR_NOT_SUPPORTED      -3   /* _CERR_NOT_SUPPORTED */
#define CLAPGPU_ERR_TOO_LARGE         -11   /* _CERR_TOO_LARGE */
#define CLAPGPU_ERR_INIT_FAILED       -14   /* _CERR_INITIALIZATION_FAILED */
#define CLAPGPU_ERR_OUT_OF_BOUNDS     -26   /* _CERR_OUT_OF_BOUNDS */
#define CLAPGPU_ERR_UNKNOWN           -32   /* _CERR_UNKNOWN_ERROR (a HIP runtime error; see clapgpu_last_error) */

/* ---- entity3d_flags bits the path reads (model.h:293-312) + the dirty mirror ---- */
#define CLAPGPU_E_VISIBLE       (1u << 0)    /* ENTITY3D_VISIBLE */
#define CLAPGPU_E_SKIP_CULLING  (1u << 14)   /* ENTITY3D_SKIP_CULLING */
#define CLAPGPU_E_DIRTY         (1u << 16)   /* mirror of transform_t.updated (transform.h:11) */
#define CLAPGPU_E_JOINT_ATTACHED (1u << 17) /* e->parent_joint != JOINT_TYPE_MAX (model.h:386-403); see clapgpu_attach */
#define CLAPGPU_E_ALIVE         (1u << 31)   /* ENTITY3D_ALIVE */

/* ---- runtime ---- */
/* Number of HIP devices visible; <0 on error.  Does not create a context. */
int clapgpu_device_count(void);
/* Bind the calling thread to `device` (hipSetDevice) and verify it is a gfx950 part. */
int clapgpu_init(int device);
/* Text of the last HIP error seen by this library on the calling thread ("" if none). */
const char *clapgpu_last_error(void);
/* For the CALLERS' failure-path tests (oracle/ref/dropin.c `fail`, tests/test_dropin.py): after `launches` more kernel
 * launches have been checked, every later launch of this process reports a launch failure (the kernel itself runs;
 * the entry point returns CLAPGPU_ERR_UNKNOWN with clapgpu_last_error() set).  < 0 turns it off (the default).  Armed only
 * in a process started with CLAPGPU_TEST_HOOKS in its environment: anywhere else the call does nothing. */
void clapgpu_test_fail_after(int launches);
/* ABI version: bumped whenever a signature or struct below changes. */
uint32_t clapgpu_abi_version(void);

int clapgpu_malloc(void **dev, size_t bytes);
int clapgpu_free(void *dev);
/* Page-locked host memory for the staging arrays of a binding: copies to and from it run at the
 * link rate and truly asynchronously (pageable memory is bounced through a driver buffer). */
int clapgpu_host_malloc(void **host, size_t bytes);
/* Page-locked host memory that kernels address directly (zero-copy over PCIe): *dev_alias is what goes into the
 * device-pointer fields of the structs below; free with clapgpu_host_free(*host).  Coherent: a kernel's stores are
 * visible to the host once the kernel has completed (or, inside a kernel, after __threadfence_system()).  For batches
 * small enough that the copies' fixed latencies (one each way + a blocking wait) exceed the transfer itself:
 * libclapgpu_scene uses it below ~128 k entities (clapgpu_entities_apply_inputs / _export_rebuilt). */
int clapgpu_host_malloc_mapped(void **host, void **dev_alias, size_t bytes);
int clapgpu_host_free(void *host);
int clapgpu_memcpy_h2d(void *dev, const void *host, size_t bytes, void *stream);
int clapgpu_memcpy_d2h(void *host, const void *dev, size_t bytes, void *stream);
int clapgpu_memset(void *dev, int value, size_t bytes, void *stream);
int clapgpu_stream_sync(void *stream);

/* ======================================================================== */
/* Entities: transform hierarchy -> inverse -> world AABB -> frustum cull    */
/* ======================================================================== */

/*
 * View frustum of the main subview: view.h:16-17 (`frustum_planes[6]`,
 * `frustum_corners[8]`), produced on the host by clapgpu_frustum_calc().
 */
typedef struct clapgpu_frustum {
    float planes[6][4];
    float corners[8][4];
} clapgpu_frustum;

/* transform.c:132-138 transform_view_mat4x4 (host, O(1) per frame) */
void clapgpu_view_matrix(const float pos[3], const float quat[4], float view_mx[16]);
/* linmath.h:611-651 mat4x4_invert, 959-987 mat4x4_from_quat (host; the arithmetic the kernels use) */
void clapgpu_mat4_invert(const float m[16], float out[16]);
void clapgpu_mat4_from_quat(const float quat_xyzw[4], float out[16]);
/* linmath.h:709-776 mat4x4_perspective_ndc_z_{2,1} via render-common.c:77-82 (host) */
void clapgpu_perspective(float fov, float aspect, float near_plane, float far_plane,
                         int ndc_z_zero_one, float proj_mx[16]);
/* view.c:248-289 subview_calc_frustum (host) */
void clapgpu_frustum_calc(const float view_mx[16], const float proj_mx[16],
                          int ndc_z_zero_one, clapgpu_frustum *out);

/*
 * SoA mirror of the entity3d fields on the path (model.h:372-429), all device
 * pointers, `n` entities.  Entities are stored so that parent[i] < i; a
 * "level" is a maximal index range whose parents all lie in earlier levels.
 * Level starts must be multiples of 64 (pad with flags == 0 entities) so that
 * each 64-entity wavefront owns exactly one vis_mask word.
 *
 *   inputs   pos_scale[n][4]  (pos.xyz, scale)       transform_t.pos, entity3d.scale
 *            rot[n][4]        quat xyzw              transform_t.rotation
 *            parent[n]        index or -1            entity3d.parent (jointless attach)
 *            model[n]         index into model_table entity3d.txmodel->model
 *            model_table[m][8]  (min.xyz, skip_aabb as uint32 bits, max.xyz,
 *                                lod_min | lod_max << 8 as uint32 bits)
 *                                                    model3d.aabb, .skip_aabb, .lod_min, .lod_max
 *   in/out   flags[n]         CLAPGPU_E_* bits       entity3d.flags + xform.updated
 *            seqs[n]          seq | parent_seq<<16   entity3d.seq / .parent_seq (uint16 wrap)
 *   outputs  mx[n][16], inv_mx[n][16]                entity3d.mx, .inverse_mx
 *            aabb[n][6]  (min.xyz, max.xyz)          entity3d.aabb
 *            center[n][3]                            entity3d.aabb_center
 *            vis_mask[ceil(n/64)]  bit i%64 of word i/64 = entity i passes the
 *                                  draw predicate of _models_render (model.c:959-973)
 *            vis_row_pop[ceil(n/64)] popcount of each vis_mask word (uint8); allocate the
 *                                  array rounded up to a multiple of 16 bytes, 16-B aligned
 */
/*
 * Joint attachment (parent_transform_apply's second flavour, model.c:1626-1641): entity `entity`
 * (flag CLAPGPU_E_JOINT_ATTACHED) rides mx = parent.mx * ((jt_pool[jt] * bind_pool[bind]) * local)
 * where jt_pool[jt] is the parent's joint_transforms[parent_joint] of THIS frame and
 * bind_pool[bind] that joint's bind matrix; it is rebuilt every frame (never skipped).  Launch
 * order is the caller's: parents' entity update -> clapgpu_pose_update -> the update of the
 * tiles holding attached subtrees.  The table is sorted by `entity`.  (In the reference a
 * resolved joint index equal to JOINT_TYPE_MAX (6) is indistinguishable from "no joint",
 * model.c:1609,1624: do not flag such an entity.)
 */
typedef struct clapgpu_attach {
    uint32_t entity, jt, bind, pad;
} clapgpu_attach;

/*
 * default_update's camera bounding-volume pick (model.c:1703-1713, consumed by
 * scene_camera_calc, scene.c:1018-1038): among ALIVE entities whose world AABB contains the
 * camera position or the control entity's position (except the control entity itself) the one of
 * largest volume (|model dx| scale)(|model dy| scale)(|model dz| scale).
 * *result (device uint64, zeroed by the update call) = float bits of the volume << 32 |
 * (0xFFFFFFFF - entity index), 0 if none: the largest key is the first entity of largest volume.
 * result may be NULL when inside_mask is given (no maximum is formed then, and the update call queues no fill for it).
 * inside_mask (device, n / 64 words, may be NULL): bit i = entity i passed the containment test.  A caller whose
 * entity order differs from the device's index order (the reference breaks volume ties by LIST order) replays
 * the pick over those few entities itself.
 */
typedef struct clapgpu_bv_query {
    float     cam_pos[3];
    uint32_t  has_ctl;
    float     ctl_pos[3];
    uint32_t  ctl_entity;
    uint64_t *result;
    uint64_t *inside_mask;
} clapgpu_bv_query;

/*
 * Extra views culled by the SAME launch as the main frustum: the frame's other render passes.  pipeline_render() runs one
 * shadow pass per cascade with view = &light->view[0] and no camera (pipeline-builder.c:34-46, 246-272; model.c:752-760)
 * before the model pass with the camera's view, and every pass asks view_entity_in_frustum(view, e) for every entity
 * (model.c:966-973; the test reads view->main only, view.c:296-337): two frusta a frame, more with more shadowed lights.
 * The rows' boxes are in registers when the main view is tested: a further view costs ~150 flops and 1/64 word per entity.
 * Every view v gets its own vis_mask / vis_row_pop plane (same shapes as clapgpu_entities'), written exactly as the main
 * plane is: bit i = entity i passes _models_render's draw predicate for that view.  Only read when the call has a main
 * frustum.  host_vis_mask: clapgpu_entities_update_tiles_hostio only -- device aliases of mapped host words the launch
 * writes as well (NULL: not wanted); such a view's drawn entities count as read for the export policy (keep_mask).
 */
#define CLAPGPU_EXTRA_VIEWS_MAX 4                          /* CASCADES_MAX, shader_constants.h:9 */
typedef struct clapgpu_views {
    uint32_t        n, pad;                                /* extra views: 0 .. CLAPGPU_EXTRA_VIEWS_MAX */
    clapgpu_frustum frustum[CLAPGPU_EXTRA_VIEWS_MAX];
    uint64_t       *vis_mask[CLAPGPU_EXTRA_VIEWS_MAX];     /* device */
    uint8_t        *vis_row_pop[CLAPGPU_EXTRA_VIEWS_MAX];  /* device, 16-byte aligned */
    uint64_t       *host_vis_mask[CLAPGPU_EXTRA_VIEWS_MAX];
} clapgpu_views;

typedef struct clapgpu_entities {
    uint32_t        n;
    uint32_t        n_models;
    const float    *pos_scale;
    const float    *rot;
    const int32_t  *parent;
    const int32_t  *model;
    const float    *model_table;
    uint32_t       *flags;
    uint32_t       *seqs;
    float          *mx;
    float          *inv_mx;
    float          *aabb;
    float          *center;
    uint64_t       *vis_mask;
    uint8_t        *vis_row_pop;
    /* optional (zero / NULL when unused) */
    uint32_t        n_attach;
    uint32_t        pad;
    const clapgpu_attach *attach;        /* device, sorted by entity */
    const float    *jt_pool;             /* device mat4[]: joint_transforms of the characters */
    const float    *bind_pool;           /* device mat4[]: model_joint.bind */
    float          *attach_local;        /* device work space, n_attach mat4 */
    const clapgpu_bv_query *bv;          /* HOST pointer */
    uint64_t       *rebuilt_mask;        /* device, n / 64 words, may be NULL: bit i = this update rebuilt entity i
                                            (mx / inverse_mx / aabb / seq changed): what a host mirror has to copy back */
    const clapgpu_views *views;          /* HOST pointer or NULL: the frame's other frusta, culled by the same launch */
} clapgpu_entities;

/* mode bits for clapgpu_entities_update */
#define CLAPGPU_UPDATE_ALL_DIRTY  (1u << 0)   /* treat every ALIVE entity as xform.updated; flags[] is not written */

/*
 * Replaces mq_update() over default_update entities (model.c:1953 -> 1649-1695,
 * parent_transform_apply 1594-1647, mat4x4_invert, entity3d_aabb_update 1200-1234)
 * and, when `frustum` is non-NULL, the per-entity view_entity_in_frustum() test of
 * _models_render (view.c:296-337, model.c:959-973) fused into the same pass.
 *
 * level_start is a HOST array of n_levels+1 ascending offsets (level_start[0] == 0,
 * level_start[n_levels] == e->n, every start a multiple of 64).  One launch per level.
 * With frustum == NULL vis_mask is left untouched.
 */
int clapgpu_entities_update(void *stream, const clapgpu_entities *e,
                            const uint32_t *level_start, uint32_t n_levels,
                            uint32_t mode, const clapgpu_frustum *frustum);

/*
 * One hierarchy level of the above: entities [first, first+count), whose parents
 * were all updated by earlier calls on the same stream.  `first` must be a multiple
 * of 64.  (Used by callers that interleave their own work or timing between levels.)
 */
int clapgpu_entities_update_level(void *stream, const clapgpu_entities *e,
                                  uint32_t first, uint32_t count,
                                  uint32_t mode, const clapgpu_frustum *frustum);

/*
 * Single-launch form of clapgpu_entities_update for forests laid out in TILES.
 * A tile is a run of consecutive 64-entity rows; row r = entities [64r, 64r+64).
 * Tile t = rows [tile_row_start[t], tile_row_start[t+1]) and holds whole subtrees,
 * one hierarchy level per row: a child in row r has its parent in row r-1 of the
 * same tile (taken from registers, never re-read from HBM).  Unused lanes of a row
 * are padding entities (flags == 0).  A parent outside the tile's previous row is
 * legal only if it is not rebuilt by this call (it is read from mx[] / seqs[]).
 * tile_row_start is a DEVICE array of n_tiles+1 ascending row indices.
 * One wavefront walks one tile; tiles are independent, so one launch covers every level.
 */
int clapgpu_entities_update_tiles(void *stream, const clapgpu_entities *e,
                                  const uint32_t *tile_row_start, uint32_t n_tiles,
                                  uint32_t mode, const clapgpu_frustum *frustum);

/*
 * Cull only: view_entity_in_frustum() over all entities from the stored aabb[]
 * (one call per render pass in the reference, model.c:969-970).  Writes vis_mask, and the planes of e->views with it.
 */
int clapgpu_entities_cull(void *stream, const clapgpu_entities *e, const clapgpu_frustum *frustum);

/*
 * Frames of a HOST mirror without copy calls (what libclapgpu_scene does for the level layout): the frame's touched
 * entities travel as one list of 40-byte records in device-mapped host memory (clapgpu_host_malloc_mapped) and are
 * scattered into pos_scale / rot / flags by clapgpu_entities_apply_inputs(); after the update,
 * clapgpu_entities_export_rebuilt() copies the rows of mx / inv_mx / aabb / center the update rebuilt (e->rebuilt_mask)
 * and the three bit masks into the mirror's mapped result arrays and then stores done_value into *done, which the host
 * polls (clapgpu_wait_word) -- no copy calls, no blocking wait.  counter: one zeroed device uint32 of scratch.
 */
typedef struct clapgpu_entity_input {
    uint32_t slot, flags;               /* flags as in clapgpu_entities.flags (CLAPGPU_E_DIRTY where xform.updated) */
    float    pos_scale[4], rot[4];
} clapgpu_entity_input;
typedef struct clapgpu_entities_export {
    float    *mx, *inv_mx, *aabb, *center;                 /* device-mapped host arrays, same shapes as clapgpu_entities' */
    uint64_t *vis_mask, *rebuilt_mask, *inside_mask;       /* inside_mask may be NULL */
    uint32_t *counter;                                     /* device scratch */
    uint32_t *done;                                        /* device-mapped host word */
    uint32_t  done_value, pad;
    uint64_t *stale_mask;                                  /* clapgpu_entities_export_rows: clapgpu_entities_hostio.stale_mask, or NULL */
} clapgpu_entities_export;
int clapgpu_entities_apply_inputs(void *stream, const clapgpu_entities *e, const clapgpu_entity_input *list, uint32_t n_list);
int clapgpu_entities_export_rebuilt(void *stream, const clapgpu_entities *e, const clapgpu_entities_export *x);
/*
 * A standing layout edited in place (entity3d_make / entity3d_delete between two frames, model.c:1730-1791; libclapgpu_scene's
 * clapgpu_scene_entity_new_placed): the lanes that got a new tenant, as one list in device-mapped host memory -- parent[slot]
 * and model[slot] are set; CLAPGPU_PLACE_ZERO_BOX also clears the lane's aabb / center rows (a model with skip_aabb never writes
 * them, and a fresh entity3d's are zeros, not the last tenant's).  One small launch instead of two copies and two fills per edit.
 */
typedef struct clapgpu_entity_place { uint32_t slot; int32_t parent, model; uint32_t flags; } clapgpu_entity_place;
#define CLAPGPU_PLACE_ZERO_BOX    1u    /* clear the lane's aabb / center rows */
#define CLAPGPU_PLACE_CLEAR_STALE 2u    /* clear the lane's bit in stale_mask (clapgpu_entities_hostio.stale_mask; the lane's last tenant's) */
int clapgpu_entities_place(void *stream, const clapgpu_entities *e, const clapgpu_entity_place *list, uint32_t n_list,
                           uint64_t *stale_mask);
/*
 * The rows a mirror asks for after the fact (a mirror that takes back only what is drawn, clapgpu_entities_hostio.keep_mask,
 * fetches the rest on demand: an entity that comes into view, entity3d_update() on a hidden one, ...): copies the rows of
 * mx / inv_mx / aabb / center flagged in select_mask (e->n / 64 words, device-readable: device memory or a mapped host
 * alias) into x's mapped arrays and raises *x->done = x->done_value.  x's three mask pointers are not used.
 */
int clapgpu_entities_export_rows(void *stream, const clapgpu_entities *e, const clapgpu_entities_export *x,
                                 const uint64_t *select_mask);

/*
 * The same small frame as ONE launch, for the tile layout: clapgpu_entities_update_tiles() that takes the inputs of the
 * slots flagged in `touched` (one bit per slot, set by the host for this frame) from the mirror's device-mapped upload
 * image (pos_scale / rot / flags in slot order, the bytes a copy would have carried) and stores them into the device
 * arrays as it goes, writes every row it rebuilds into the mapped result arrays as well, the three masks with them, and
 * raises *done = done_value when the last workgroup has finished (clapgpu_wait_word).  touched == NULL: no inputs this
 * launch (a second launch behind a pose: clapgpu_scene_attached_update).  Three dependent launches of a few microseconds
 * each cost a 10 k-entity frame 50 of its 68 us in launch-to-launch latency; this is one.
 */
#define CLAPGPU_HOSTIO_EXPORT_STALE_READ 1u
typedef struct clapgpu_entities_hostio {
    const float    *pos_scale, *rot;                       /* device aliases of the mapped upload image */
    const uint32_t *flags;
    const uint64_t *touched;                               /* device alias of the mapped touched-slot bits, or NULL */
    float    *mx, *inv_mx, *aabb, *center;                 /* device aliases of the mapped result arrays */
    uint64_t *vis_mask, *rebuilt_mask, *inside_mask;       /* vis_mask needed with a frustum; inside_mask may be NULL */
    uint32_t *counter;                                     /* device scratch: one zeroed uint32 */
    uint32_t *done;                                        /* device-mapped host word */
    uint32_t  done_value, options;                         /* CLAPGPU_HOSTIO_* bits */
    /* Export policy.  keep_mask == NULL: every rebuilt row is written to the mapped result arrays (what _models_render and
     * every other reader of e->mx / e->inverse_mx / e->aabb finds in the reference after mq_update, model.c:1022-1028,
     * 975-998).  keep_mask != NULL (device-readable, e->n / 64 words): only the rebuilt rows a reader exists for THIS
     * frame -- entities that pass the draw predicate (vis_mask; all of them without a frustum), entities whose box contains
     * a bounding-volume point (inside_mask), entities flagged in keep_mask (the mirror's standing readers: parents of
     * host-updated children, light carriers, the control entity ...).  The device arrays hold every row either way;
     * clapgpu_entities_export_rows() brings any of them over later.  exported_mask (mapped, may be NULL): bit i = row i was
     * written by this launch. */
    const uint64_t *keep_mask;
    uint64_t       *exported_mask;
    /* stale_mask (DEVICE memory, e->n / 64 words, zeroed by the caller once; may be NULL): bit i = the mirror's copy of row i is
     * older than the device's.  The launch keeps it -- a row it rebuilds without exporting becomes stale, one it exports stops
     * being so -- and, with CLAPGPU_HOSTIO_EXPORT_STALE_READ in `options`, also writes every stale row that has a reader NOW
     * (drawn by this launch's frustum, holding a bounding-volume point, flagged in keep_mask; every one without keep_mask)
     * although it did not rebuild it: what came into view is current when the launch is, without a second launch asking for
     * it.  Such rows are flagged in exported_mask and NOT in rebuilt_mask.  clapgpu_entities_export_rows() clears the bits
     * of what it hands over. */
    uint64_t       *stale_mask;
} clapgpu_entities_hostio;
int clapgpu_entities_update_tiles_hostio(void *stream, const clapgpu_entities *e, const uint32_t *tile_row_start,
                                         uint32_t n_tiles, uint32_t mode, const clapgpu_frustum *frustum,
                                         const clapgpu_entities_hostio *io);
/* Busy-wait until *word == value (a word a kernel stores into mapped host memory).  Checks `stream` every so often: if
 * it has drained and the word still differs, the signalling launch failed -> CLAPGPU_ERR_UNKNOWN instead of a hang. */
int clapgpu_wait_word(const volatile uint32_t *word, uint32_t value, void *stream);

/*
 * Ordered compaction of vis_mask into the ascending entity-index list the draw
 * loop iterates: visible[0..*count) = index_base + i for every set bit i.
 * `visible` needs room for n entries, `count` is one device uint32; index_base is the
 * global id of this shard's entity 0 (0 on a single GPU).  With vis_row_pop (as left by
 * the update / cull kernels) and n <= 4M the list is built in ONE launch; otherwise
 * (vis_row_pop == NULL or larger n) in two, using `scratch` =
 * clapgpu_visible_scratch_bytes(n) bytes of device memory.  (Build-defined: the
 * reference walks its entity list and tests each entity in place, model.c:958-973.)
 */
size_t clapgpu_visible_scratch_bytes(uint32_t n);
int clapgpu_visible_compact(void *stream, const uint64_t *vis_mask, const uint8_t *vis_row_pop,
                            uint32_t n, uint32_t index_base, uint32_t *visible, uint32_t *count,
                            void *scratch);

/*
 * Per-pass LOD pick of _models_render for the entities on the visible list (model.c:975-992,
 * SURVEY 8f rank 1): a forced LOD wins; otherwise, unless the camera is inside the entity's box,
 * lod = clamp((int)(| |aabb_center - cam|^2 - avg_edge^2 | / 3600), lod_min, lod_max) with
 * avg_edge = cbrtf(X Y Z) (entity3d_aabb_avg_edge, model.c:1261-1264).  visible / count as
 * produced by clapgpu_visible_compact (ids minus index_base index this shard's arrays);
 * force_lod[n] may be NULL (= -1 everywhere); cur_lod[n] is entity3d.cur_lod (in/out);
 * draw_lod[k] is the LOD visible[k] is drawn with: (visible, draw_lod) is the draw list.
 */
int clapgpu_entities_lod(void *stream, const clapgpu_entities *e, const uint32_t *visible,
                         const uint32_t *count, uint32_t index_base, const float cam_pos[3],
                         const int32_t *force_lod, int32_t *cur_lod, int32_t *draw_lod);
/* clapgpu_visible_compact + clapgpu_entities_lod over this batch's own vis_mask by one call: the ordered visible list of a
 * render pass and the LOD each entry is drawn with.  (Two launches: the single-kernel form was measured slower.) */
int clapgpu_visible_compact_lod(void *stream, const clapgpu_entities *e, uint32_t index_base, const float cam_pos[3],
                                const int32_t *force_lod, int32_t *cur_lod, uint32_t *visible, uint32_t *count,
                                int32_t *draw_lod, void *scratch);

/* ======================================================================== */
/* Multi-GPU: range sharding + the one exchange                  */
/* ======================================================================== */

/*
 * The path shards by entity range with whole subtrees on one rank: contiguous TILE ranges of (nearly) equal row
 * count.  tile_row_start: HOST array of n_tiles + 1 row offsets (clapgpu_entities_update_tiles).  Rank r of `world`
 * owns tiles [*first_tile, *end_tile).  Pure function: every rank computes the same cuts.
 */
int clapgpu_shard_tile_range(const uint32_t *tile_row_start, uint32_t n_tiles, uint32_t rank, uint32_t world,
                             uint32_t *first_tile, uint32_t *end_tile);

/*
 * The only exchange of a frame: the RCCL allgather (over xGMI) of every rank's compacted visible set, sent as its
 * 1-bit-per-entity mask -- one fixed-size collective, no counts, no padding, no host sync -- and its local expansion
 * into the identical ascending GLOBAL id list on every rank (shards are equal padded ranges: rank r's entity i has
 * global id r * n_pad + i).  RCCL is opened at run time (clapgpu_exchange_set_library: a path, e.g. the copy the
 * process already has; default: librccl.so of the process / the system); CLAPGPU_ERR_NOT_SUPPORTED without one.
 * Bootstrap: rank 0 calls clapgpu_exchange_unique_id(), the launcher carries the 128 bytes to every rank.
 *   vis_mask       this rank's mask, device, n_pad / 64 words
 *   gathered_mask  device, world * n_pad / 64 words (out)
 *   visible, visible_count, scratch   as clapgpu_visible_compact over world * n_pad entities; visible NULL = mask only
 * Issued on `stream`: put it on a side stream to overlap the next frame's update (two mask buffers).
 */
#define CLAPGPU_EXCHANGE_ID_BYTES 128
typedef struct clapgpu_exchange clapgpu_exchange;
void clapgpu_exchange_set_library(const char *path);
/* 1 if RCCL can be opened in this process (no communicator is made).  clapgpu_exchange_create() is COLLECTIVE -- it
 * returns when every rank has called it -- so ranks agree on this answer first (MIN over ranks) and only create the
 * exchange when all of them can; otherwise all take the launcher's fallback together. */
int  clapgpu_exchange_available(void);
int  clapgpu_exchange_unique_id(uint8_t id[CLAPGPU_EXCHANGE_ID_BYTES]);
int  clapgpu_exchange_create(clapgpu_exchange **out, const uint8_t id[CLAPGPU_EXCHANGE_ID_BYTES], int rank, int world);
/* ncclCommCount / ncclCommUserRank of the communicator and the PCI bus id of the device this rank drives
 * ("0000:c1:00.0"): gathered over all ranks they prove that N ranks on N DISTINCT GPUs took part (bench.py refuses to
 * print a line otherwise). */
int  clapgpu_exchange_info(const clapgpu_exchange *x, int *comm_ranks, int *comm_rank, char pci_bus_id[32]);
void clapgpu_exchange_destroy(clapgpu_exchange *x);
int  clapgpu_exchange_visible(void *stream, clapgpu_exchange *x, const uint64_t *vis_mask, uint32_t n_pad,
                              uint64_t *gathered_mask, uint32_t *visible, uint32_t *visible_count, void *scratch);
/* ... for shards of DIFFERENT sizes (the ranges clapgpu_shard_tile_range cuts from one scene are only nearly equal; a rank
 * may even be empty): every rank sends cap_pad / 64 words -- its own mask, zero beyond its n_pad[rank] -- and rank r's slot i
 * gets the scene-global id base[r] + i.  base / n_pad: HOST arrays of `world` entries, identical on every rank, ascending and
 * disjoint (clapgpu_shard_bases makes them); gathered_mask: world * cap_pad / 64 words; visible / scratch sized for
 * world * cap_pad entities (clapgpu_visible_scratch_bytes).  clapgpu_visible_compact_ranges is the expansion alone (the
 * exchange's second half: testable on one GPU), clapgpu_visible_expand_ranges_host the same on the host. */
int  clapgpu_exchange_visible_ranges(void *stream, clapgpu_exchange *x, const uint64_t *vis_mask, uint32_t cap_pad,
                                     const uint32_t *base, const uint32_t *n_pad, uint64_t *gathered_mask,
                                     uint32_t *visible, uint32_t *visible_count, void *scratch);
int  clapgpu_visible_compact_ranges(void *stream, const uint64_t *gathered_mask, uint32_t n_ranges, uint32_t cap_pad,
                                    const uint32_t *base, const uint32_t *n_pad, uint32_t *visible, uint32_t *count, void *scratch);
uint32_t clapgpu_visible_expand_ranges_host(const uint64_t *gathered_mask, uint32_t n_ranges, uint32_t cap_pad,
                                            const uint32_t *base, const uint32_t *n_pad, uint32_t *visible, uint32_t capacity);
/* first global id, padded size of every rank's shard and the largest of them, for clapgpu_shard_tile_range's cut (host) */
int  clapgpu_shard_bases(const uint32_t *tile_row_start, uint32_t n_tiles, uint32_t world, uint32_t *base, uint32_t *n_pad,
                         uint32_t *cap_pad);

/* ======================================================================== */
/* Particle systems: advect / respawn / billboard (core/particle.c)          */
/* ======================================================================== */

#define CLAPGPU_PART_DIST_LIN     0   /* particle_dist, particle.h:13-18 */
#define CLAPGPU_PART_DIST_SQRT    1
#define CLAPGPU_PART_DIST_CBRT    2
#define CLAPGPU_PART_DIST_POW075  3

/* struct particle_system's simulation fields (particle.c:17-29), 64 bytes */
typedef struct clapgpu_particle_system {
    float    center[3];          /* transform_pos(&ps->e->xform) */
    uint32_t dist;               /* ps->dist */
    double   radius, min_radius, radius_squared, velocity;
    uint32_t first, count;       /* particles [first, first+count); first is a multiple of 64 */
    uint32_t pad[2];
} clapgpu_particle_system;

/*
 * All particle systems of a model queue, particles stored system after system, every
 * system starting at a multiple of 64 (row_sys[r] = system of particles [64r, 64r+64)).
 *   pos[n][3]   particle.pos; after the call it IS ps->pos_array (particle.c:116,124):
 *               system s uploads pos + 3*first, count vec3s
 *   vel[n][3]   particle.velocity
 *   rng_state[2] the libc drand48 state as 48-bit integers: [1] = position of the stream
 *               before this call (input) and after it (output); [0] is internal.
 *               Initialise both to the same value (glibc default 0x1234ABCD330E).
 *   billboard_mx[n_sys][16]  entity3d.mx of each system's entity (particle.c:93-100), or NULL
 *   respawn_mask[n/64], respawn_row_pop[n/64 rounded up to 16], respawn_list[n],
 *   respawn_count[1], scratch (clapgpu_visible_scratch_bytes(n)): work space
 *   respawn_groups[CLAPGPU_RESPAWN_GROUP_WORDS]: persistent work space, ZEROED ONCE by the caller
 *               when the batch is created and then left alone (per-frame respawn counts of groups
 *               of rows, double-banked by frame); may be NULL, which costs ~10 us per call at 4 M
 *               particles (the respawn ranks then come from a scan of every row's count)
 */
#define CLAPGPU_RESPAWN_GROUP_WORDS 132
typedef struct clapgpu_particles {
    uint32_t  n;
    uint32_t  n_sys;
    const clapgpu_particle_system *sys;
    const uint32_t *row_sys;
    float    *pos;
    float    *vel;
    uint64_t *rng_state;
    float    *billboard_mx;
    uint64_t *respawn_mask;
    uint8_t  *respawn_row_pop;
    uint32_t *respawn_list;
    uint32_t *respawn_count;
    void     *scratch;
    uint32_t *respawn_groups;
} clapgpu_particles;

/*
 * Replaces particles_update() (particle.c:89-120) for every system, in system order, with
 * the reference's single drand48 stream reproduced exactly (7 draws per respawn, in particle
 * order).  view_mx = scene camera's view.main.view_mx (HOST pointer, 16 floats).
 */
int clapgpu_particles_update(void *stream, const clapgpu_particles *p, const float view_mx[16]);

/* ======================================================================== */
/* Skeletal pose blend + joint palette (core/model.c:1266-1404, interp.h)    */
/* ======================================================================== */

/*
 * Skinning constants of one model3d (model.h:104-110,59; set once by model3d_add_skinning,
 * model.c:524-538, and gltf_instantiate_one), device pointers.
 *   parent[j]   parent joint or -1 (child of root_pose)        model_joint.children, inverted
 *   depth[j]    level of joint j under joint 0, or -1 if j is NOT reachable from joint 0:
 *               the reference starts its recursion at joint 0 (model.c:1583) and never
 *               touches the others
 *   n_levels    1 + max depth
 *   invmx[j], bind[j] = invert(invmx[j])                       model_joint.invmx, .bind
 */
typedef struct clapgpu_skeleton {
    uint32_t        nr_joints;          /* <= 256 (JOINTS_MAX = 200, shader_constants.h:6) */
    uint32_t        n_levels;
    const int32_t  *parent;
    const int32_t  *depth;
    const float    *root_pose;          /* mat4 */
    const float    *invmx;              /* [nr_joints] mat4 */
    const float    *bind;               /* [nr_joints] mat4 */
} clapgpu_skeleton;

/*
 * Every animation of the model (struct animation / struct channel, model.h:133-142,
 * model.c:678-685), keyframe times strictly increasing per channel (glTF).
 *   chan_table[a][j][path] = (time_off, data_off, nr, 0) uint32x4: the channel of animation a
 *       that drives (joint j, path) -- key times at times[time_off .. +nr), key values at
 *       data[data_off ..] (floats, 3 per key for T/S, 4 for R); nr == 0: no such channel.
 *       path 0 translation, 1 rotation, 2 scale (enum chan_path); when the reference lists
 *       several channels for one (joint, path) the last one wins (model.c:1348-1349).
 *       16-byte aligned.
 */
typedef struct clapgpu_animations {
    uint32_t        n_anims;
    uint32_t        n_times;        /* floats in times[] (0: unknown) */
    const uint32_t *chan_table;
    const float    *times;
    const float    *data;
    /* the key-major copy of the pools made by clapgpu_animations_pack() for this model, the max_keys it was made with and
     * the layout word pack() returned (lanes, animation count, whether some (joint, path) has no channel).  REQUIRED by
     * clapgpu_pose_update(): besides the re-layout the copy holds, per rotation key pair, what quat_slerp (interp.h:91-118)
     * derives from the pair alone -- acos of the inner product and its sine -- evaluated by the host's libm, the very calls
     * the reference makes; the kernel has no other source for them.  Skeletons whose animations' key rows fit in LDS (one
     * animation of <= 31 keys per channel, two of <= 15, ...) search them there; one wavefront per character up to
     * 64 joints, two to four above. */
    const void     *packed;
    uint32_t        packed_keys, packed_layout;
    uint32_t        n_data, pad;    /* floats in data[] (0: unknown): with it clapgpu_animations_pack() refuses a channel that reads past the pool */
} clapgpu_animations;

/* Key-major pools, once per model: a HOST-side re-layout of chan_table / times / data (three synchronous copies on
 * `stream`) plus the rotation intervals' constants.  Checks what the kernel's exactness rests on: every channel's key times
 * strictly increasing (glTF requires it; the kernel's branch-free bracket equals channel_time_to_idx, model.c:1266-1288, only
 * then -- a channel with equal or descending times returns CLAPGPU_ERR_INVALID_ARGUMENTS and the caller keeps such a model on
 * the host path), no channel longer than max_keys or reaching past n_times / n_data.  max_keys = the largest nr of any channel; packed:
 * clapgpu_animations_packed_bytes() of device memory, 16-byte aligned; *packed_layout goes into
 * clapgpu_animations.packed_layout (clapgpu_pose_update rejects pools made for another skeleton class or animation count). */
size_t clapgpu_animations_packed_bytes(uint32_t n_anims, uint32_t max_keys, uint32_t nr_joints);
int    clapgpu_animations_pack(void *stream, const clapgpu_animations *an, uint32_t nr_joints, uint32_t max_keys, void *packed,
                               uint32_t *packed_layout);

/*
 * The animated entities of that model.
 *   anim[c]        queued_animation.animation of the current queue entry (model.c:1572-1576)
 *   frame_time[c]  (float)((now - ani_time) * speed), model.c:1578-1582
 *   entity[c]      index of the character's entity in entity_mx (NULL: c)
 *   entity_mx      the entity world matrices (clapgpu_entities.mx)
 *   trs[c][j][10]  joint translation(3) rotation(4) scale(3): struct joint (model.h:363-366),
 *                  in/out -- a path no channel drives keeps its value
 *   joint_transforms[c][j][16]  entity3d.joint_transforms, the UNIFORM_JOINT_TRANSFORMS payload
 *                  (model.c:1020-1022); joints not reachable from joint 0 are not written
 *   joint_pos[c][j][4]          struct joint.pos (camera.c:195-196); NULL = not computed
 *   skip           CLAPGPU_POSE_SKIP_*: outputs nothing reads this frame.  struct joint's translation / rotation / scale
 *                  and pos are host-visible state whose only per-frame reader is one_joint_transform itself
 *                  (model.c:1352-1404) and camera_target (camera.c:191-205); the draw path consumes joint_transforms
 *                  alone (model.c:1020-1022).  With SKIP_TRS the blended T/R/S stay in registers; that is 40 of the 120
 *                  bytes a joint writes, joint_pos another 16.  SKIP_TRS is refused (CLAPGPU_ERR_INVALID_ARGUMENTS) for a
 *                  model some of whose (joint, path) pairs have no channel in some animation: such a path keeps the
 *                  joint's LAST interpolated value (model.c:1301), which lives in trs[].
 */
#define CLAPGPU_POSE_SKIP_TRS        (1u << 0)
#define CLAPGPU_POSE_SKIP_JOINT_POS  (1u << 1)
/* joint_pos[] receives the MODEL-space position (model.c:1392-1397: column 3 of joint_transforms * bind) instead of
 * e->mx times it, and entity_mx is not read: the pose no longer waits for the frame's entity update.
 * clapgpu_joint_pos_world() finishes model.c:1400 afterwards -- the same mat4x4_mul_vec4_post, the same bits. */
#define CLAPGPU_POSE_JOINT_POS_MODEL (1u << 2)
typedef struct clapgpu_pose_batch {
    uint32_t        n_chars;
    uint32_t        skip;
    const uint32_t *anim;
    const float    *frame_time;
    const uint32_t *entity;
    const float    *entity_mx;
    float          *trs;
    float          *joint_transforms;
    float          *joint_pos;
} clapgpu_pose_batch;

/*
 * The time base of animated_update() (model.c:1563-1592) for the batch, on the device: per character
 *   frame_time[c] = (float)((now - ani_time[c]) * speed[c])        (double arithmetic, model.c:1578-1582)
 *   ended[c]      = (now - ani_time[c]) * speed[c] >= time_end[anim[c]]      (model.c:1590)
 * and for a character whose current queue entry repeats (restart[c] != 0) the animation_next() ->
 * animation_start() that follows: ani_time[c] = now (model.c:1406-1424, 1455-1483).  Entries that do
 * not repeat are left to the host, which reads ended[] to run its queue logic and callbacks
 * (animation_end, frame_cb, ani_cleared).  Run before clapgpu_pose_update with the same frame_time
 * array.  time_end[a] = animation.time_end of the model's animation a; now = clap_get_current_time().
 */
typedef struct clapgpu_anim_clock {
    uint32_t        n_chars;
    uint32_t        n_anims;
    const uint32_t *anim;
    const float    *time_end;
    double         *ani_time;
    const float    *speed;
    const uint8_t  *restart;
    float          *frame_time;
    uint8_t        *ended;
} clapgpu_anim_clock;
int clapgpu_animation_time(void *stream, const clapgpu_anim_clock *clk, double now);
/* The same with `now` read from device memory (one double): the launch then carries no per-frame value
 * and can sit in a captured HIP graph; the caller writes *now_dev before replaying it. */
int clapgpu_animation_time_dev(void *stream, const clapgpu_anim_clock *clk, const double *now_dev);

/* Replaces channels_transform() + one_joint_transform(e, 0, -1) of animated_update()
 * (model.c:1582-1583) for every character of the batch. */
int clapgpu_pose_update(void *stream, const clapgpu_skeleton *sk, const clapgpu_animations *an,
                        const clapgpu_pose_batch *pb);
/* model.c:1400 for a batch posed with CLAPGPU_POSE_JOINT_POS_MODEL: joint_pos[c][j] = entity_mx[entity[c]] * joint_pos[c][j]
 * (mat4x4_mul_vec4_post), in place, for every joint under joint 0 (the others are never written, as in the fused form). */
int clapgpu_joint_pos_world(void *stream, const clapgpu_skeleton *sk, const clapgpu_pose_batch *pb);

/* ======================================================================== */
/* Vertex skinning (shaders/model.vert:32-48)                                */
/* ======================================================================== */

/*
 * Character c skins mesh vertices [vert_first[c], vert_first[c] + vert_count[c]) of the
 * vertex pool with its palette joint_transforms[c][nr_joints] and writes them at
 * out_first[c].  Vertex attributes in the reference's formats (mesh.h:125-131, gltf.c:387-388):
 * position f32x3, normal f32x3, joints u8x4, weights f32x4 (16-B aligned).  Instances of one
 * model share vert_first.  Outputs: position f32x3 + normal f32x3 in the joint-space-blended
 * model space the shader would feed to `trs` (model.vert:44-45).
 *
 * The w component.  The shader multiplies proj * view * trs by the vec4 total_local_pos (model.vert:32-45), whose
 * w = sum_i weights[i] * (row 3 of joint_transforms[joints[i]] . (position, 1)) -- for affine palettes the SUM OF THE
 * WEIGHTS, which the reference never renormalises.  A draw that feeds vec4(out_position, 1) therefore equals the
 * shader only for vertices whose weights sum to 1.  Two ways to stay exact:
 *   - out_w != NULL: the kernel also writes that w (f32 per vertex, 28 B / vertex out instead of 24) and the draw
 *     feeds vec4(out_position, out_w);
 *   - out_w == NULL: the caller vouches that |sum(weights) - 1| is negligible for every vertex of the pool;
 *     clapgpu_load_gltf() reports the worst deviation per mesh (clapgpu_load.h: weight_sum_max_dev) so that a
 *     caller can decide.
 */
typedef struct clapgpu_skin_batch {
    uint32_t        n_chars;
    uint32_t        nr_joints;
    const uint32_t *vert_first;
    const uint32_t *vert_count;
    const uint32_t *out_first;
    const float    *position;
    const float    *normal;
    const uint8_t  *joints;
    const float    *weights;
    const float    *joint_transforms;
    float          *out_position;
    float          *out_normal;
    float          *out_w;              /* optional: total_local_pos.w per output vertex (see above) */
} clapgpu_skin_batch;

int clapgpu_skin(void *stream, const clapgpu_skin_batch *b);

/* ======================================================================== */
/* Rigid bodies: phys_step schedule, integrate, read-back, AABB broadphase   */
/* (core/physics.c over ODE).  ODE is an absent submodule of the reference:  */
/* the arithmetic is restated from ODE's published quickstep -- see          */
/* oracle/physics.c and DESIGN.md; parity for this block is UNPINNED.        */
/* ======================================================================== */

#define CLAPGPU_BODY_DISABLED      (1u << 0)   /* dxBodyDisabled */
#define CLAPGPU_BODY_AUTO_DISABLE  (1u << 1)   /* dBodySetAutoDisableFlag(body, 1), physics.c:1039 */
#define CLAPGPU_BODY_NO_GRAVITY    (1u << 2)   /* dBodySetGravityMode(body, 0) */

/* world parameters phys_init() / phys_body_new() set (physics.c:1125-1129, 1039-1042) */
typedef struct clapgpu_world {
    double  gravity[3];
    double  linear_damping;
    double  linear_damping_threshold_sq;
    double  adis_linear_threshold_sq;
    double  adis_angular_threshold_sq;
    double  adis_time;
    int32_t adis_steps;
    int32_t pad;
} clapgpu_world;

#define CLAPGPU_BODY_GYROSCOPIC    (1u << 3)   /* dxBodyGyroscopic: dBodyCreate sets it, dBodySetGyroscopicMode(b, 0) clears it */
#define CLAPGPU_BODY_HAS_JOINT     (1u << 4)   /* the body holds a (contact) joint this step: ODE never auto-disables a jointless
                                                  body; set by clapgpu_contacts_geoms, cleared by clapgpu_bodies_step */
#define CLAPGPU_GEOM_SPHERE  0
#define CLAPGPU_GEOM_CAPSULE 1
#define CLAPGPU_GEOM_BOX     2                 /* an axis-aligned box given by its AABB (stand-in for any static geom) */
#define CLAPGPU_GEOM_OTHER   3                 /* trimesh etc.: broadphase only, no narrowphase here */

/*
 * Bodies of the character_space, fp64 like the reference's dDOUBLE ODE (physics.h:5-9): capsules
 * (phys_geom_capsule_new -> dCreateCapsule + dMassSetCapsuleTotal, physics.c:814-873) and, when the
 * capsule's length comes out 0, spheres (dCreateSphere + dMassSetSphereTotal, physics.c:866-873).
 *   pos[n][3], quat[n][4] (w,x,y,z = ODE order), lvel[n][3], avel[n][3]  in/out
 *   mass[n], radius[n]                                                     dMass.mass, geom radius
 *   yoffset[n]      phys_body.yoffset (physics.c:797-799)
 *   bflags[n]       CLAPGPU_BODY_* ; adis_steps_left / adis_time_left: ODE's auto-disable counters
 *   body_entity[n]  index of the entity3d the geom's data points at, or -1
 *   length[n]       capsule cylinder length, 0 = sphere; NULL = all spheres
 *   inertia[n][3]   diagonal of dMass.I in the body frame (clapgpu_mass_capsule_total / _sphere_total);
 *                   NULL = no rotational dynamics beyond the free spin (as if dBodySetGyroscopicMode(b, 0))
 *   geom_offset_R   dGeomSetOffsetRotation of the capsule geoms (physics.c:974-978), ODE dMatrix3
 *                   (3 rows of 4); clapgpu_geom_offset_rotation() fills it
 *   aabb[n][6]      out: the geom's AABB (minx,maxx,miny,maxy,minz,maxz), ODE's dReal aabb[6]; written by
 *                   clapgpu_bodies_aabb and by clapgpu_bodies_step for the bodies it moves; may be NULL
 *   axis[n][3]      out: the capsule's axis in world space (column 2 of body R * offset R); may be NULL
 *   adis_average_samples  dBodySetAutoDisableAverageSamplesCount (ODE's default: 1 = the instantaneous
 *                   velocity); > 1 needs adis_samples[n][samples][6] (lvel, avel ring) and adis_counter[n]
 */
typedef struct clapgpu_bodies {
    uint32_t        n;
    uint32_t        adis_average_samples;
    double         *pos;
    double         *quat;
    double         *lvel;
    double         *avel;
    const double   *mass;
    const double   *radius;
    const double   *yoffset;
    uint32_t       *bflags;
    int32_t        *adis_steps_left;
    double         *adis_time_left;
    const int32_t  *body_entity;
    const double   *length;
    const double   *inertia;
    double          geom_offset_R[12];
    double         *aabb;
    double         *axis;
    double         *adis_samples;
    uint32_t       *adis_counter;
    /* optional (NULL: none): [n][8] doubles -- position, capsule axis, radius, length of every body's geom in ONE
     * 64-byte record, rewritten by clapgpu_bodies_step / clapgpu_bodies_aabb beside pos / axis whenever they move a geom.
     * Hand the same pointer to the narrowphase as clapgpu_geoms.records: a candidate pair's PARTNER is then one 64-byte
     * sector instead of four scattered ones (position, axis, radius, length) -- near_callback's gathers were 158 of the
     * contact kernel's 231 MB at configs[3].  16-byte aligned. */
    double         *geom_records;
} clapgpu_bodies;

/* host helpers for the set-up the reference does once per body (no device work) */
void clapgpu_geom_offset_rotation(double R[12]);                        /* dRFromAxisAndAngle(1,1,1,-2pi/3), physics.c:974-978 */
void clapgpu_mass_sphere_total(double total_mass, double radius, double I[3]);                 /* dMassSetSphereTotal */
void clapgpu_mass_capsule_total(double total_mass, int direction, double radius, double length, double I[3]);
/* phys_geom_capsule_new (physics.c:814-873): entity AABB extents -> capsule radius, length, yoffset, direction, ray_off */
void clapgpu_capsule_geom(float X, float Y, float Z, double geom_radius, double geom_offset,
                          float *radius, float *length, float *yoffset, int *direction, float *ray_off);
/* axis + AABB of every body geom from the current pose (dxCapsule::computeAABB / dxSphere::computeAABB) */
int clapgpu_bodies_aabb(void *stream, const clapgpu_bodies *b);

/* physics.c:773-787: adds dt to *time_acc and returns how many 1/120 s substeps to run (0..5) */
int  clapgpu_phys_step_schedule(double *time_acc, double dt);
void clapgpu_world_defaults(clapgpu_world *w);

/*
 * dWorldQuickStep(world, h) for bodies without constraint rows (physics.c:769): auto-disable bookkeeping
 * (bodies holding a joint only, averaged over adis_average_samples), gravity, the implicit gyroscopic torque
 * of CLAPGPU_BODY_GYROSCOPIC bodies with an inertia tensor, velocity and pose update, linear damping, then
 * the moved geom's axis and AABB.
 */
int clapgpu_bodies_step(void *stream, const clapgpu_bodies *b, const clapgpu_world *w, double h);
/* The same step, which ALSO does the first launch of the next clapgpu_bp_collide(bp, b->n, b->aabb) -- the bin pass reads
 * nothing but the box the step has just computed: one launch and one pass over the boxes less per substep.  The collide
 * call recognises the pre-binned array by (pointer, count) and skips its bin launch; if b->aabb is changed by anything
 * else in between (clapgpu_bodies_aabb, an upload) call clapgpu_bp_invalidate(bp) first.  Results are the same pairs
 * (the list is canonical whatever order the bin atomics took).  b->aabb is required. */
struct clapgpu_bp;
int clapgpu_bodies_step_prebin(void *stream, const clapgpu_bodies *b, const clapgpu_world *w, double h, struct clapgpu_bp *bp);
int clapgpu_bp_invalidate(void *stream, struct clapgpu_bp *bp);

/*
 * phys_body_update() for every body (physics.c:789-812, 96-109): writes entity pos
 * (y - yoffset, double -> float) and rotation (wxyz -> xyzw) into the entity SoA
 * (clapgpu_entities.pos_scale / .rot, n_entities slots: a body_entity outside them writes nothing),
 * sets CLAPGPU_E_DIRTY, and moving[i] = |lvel| > 1e-3 (may be NULL).
 */
int clapgpu_phys_body_update(void *stream, const clapgpu_bodies *b, uint32_t n_entities, float *pos_scale,
                             float *rot, uint32_t *entity_flags, uint8_t *moving);

/*
 * default_update's push of the entity rotation to the physics body of characters and static
 * colliders (model.c:1680-1687 -> phys_body_rotate_xform, physics.c:136-145): link k = (body
 * link_body[k], entity link_entity[k]).  For a linked entity without a parent whose CLAPGPU_E_DIRTY
 * is set (or any, under CLAPGPU_UPDATE_ALL_DIRTY) the body quaternion becomes the entity's
 * (x,y,z,w) -> (w,x,y,z) in double, normalised as dBodySetQuaternion does.  Call BEFORE
 * clapgpu_entities_update, which clears the dirty flags.
 */
int clapgpu_bodies_rotate_from_entities(void *stream, const clapgpu_bodies *b, const clapgpu_entities *e,
                                        uint32_t mode, uint32_t n_links, const uint32_t *link_body,
                                        const uint32_t *link_entity);

/*
 * near_callback() for the sphere bodies (physics.c:399-449): dCollide on every
 * candidate pair of clapgpu_bp_collide, plus the surface parameters phys_contact_surface
 * (physics.c:291-330) gives each contact.  One record per pair, in pair order (the reference
 * creates its contact joints in callback order; here the canonical pair order): nc = dCollide's
 * return value (0 or 1 for spheres).  ODE's dContactGeom / dSurfaceParameters fields, doubles.
 * material[b] = (bounce, bounce_vel, mu, soft_erp, soft_cfm) of body b's phys_body
 * (physics.c:77-81), or NULL for bodies without parameters.  *contact_total (device, may be NULL)
 * receives the number of touching pairs.  n_pairs is read from the device counter pair_total,
 * clamped to capacity.  Contact joints and the LCP solve stay in ODE.
 */
#define CLAPGPU_CONTACT_BOUNCE   0x004   /* dContactBounce  (ode/contact.h) */
#define CLAPGPU_CONTACT_SOFT_ERP 0x008   /* dContactSoftERP */
#define CLAPGPU_CONTACT_SOFT_CFM 0x010   /* dContactSoftCFM */
typedef struct clapgpu_contact {
    double   pos[3], normal[3], depth;
    double   mu, bounce, bounce_vel, soft_erp, soft_cfm;
    uint32_t mode;
    uint32_t nc;
} clapgpu_contact;
int clapgpu_contacts_spheres(void *stream, const clapgpu_bodies *b, const uint32_t *pairs,
                             const uint32_t *pair_total, uint32_t capacity, const double *material,
                             clapgpu_contact *contacts, uint32_t *contact_total);
/*
 * The same for the (body, static geom) candidate pairs of clapgpu_bp_collide: dCollide of a
 * sphere (g1) against an axis-aligned box (g2) -- ODE's dCollideSphereBox with the box given as static_aabb[s]
 * (position = centre, side = max - min, no rotation): the sphere centre clamped to the box, contact at the
 * clamped point with the normal from the box to the sphere, or, for a centre inside the box, at the
 * centre with the normal through the closest face and depth = distance to that face + radius.
 * static_material[s] = the static collider's phys_body parameters (same 5 doubles); surface defaults
 * unless both material arrays are given (phys_contact_surface needs both phys bodies, physics.c:296).
 */
int clapgpu_contacts_sphere_box(void *stream, const clapgpu_bodies *b, uint32_t n_static,
                                const double *static_aabb, const uint32_t *pairs, const uint32_t *pair_total,
                                uint32_t capacity, const double *material, const double *static_material,
                                clapgpu_contact *contacts, uint32_t *contact_total);

/*
 * Broadphase over explicit fp64 AABBs (round 2): both calls of __phys_step (physics.c:751-753) in one pass of
 * four launches.  Bodies are binned by their AABB centre into a hash grid of 4x4x4-cell blocks (cell >= the
 * largest body AABB edge); one wavefront per block tests the block's own bodies against the block + its halo
 * staged in LDS, and against the static geoms registered for that block.
 *   clapgpu_bp_create   n_max bodies; statics: n_static AABBs in HOST memory (copied; binned once: statics do
 *                       not move; those spanning more than 64 blocks are kept in a list every block tests)
 *   clapgpu_bp_collide  aabb: device [n][6] (clapgpu_bodies.aabb).  pairs / static_pairs are ascending (i, j), i < j resp. (body, static);
 *                       *_total = number found (device uint32), at most `capacity` written (when a total
 *                       exceeds its capacity the written part is incomplete).  static outputs may be NULL.
 *   clapgpu_bp_status   host sync; bit 0: a body AABB edge exceeded `cell` (pairs may be missing)
 */
typedef struct clapgpu_bp clapgpu_bp;
int  clapgpu_bp_create(clapgpu_bp **out, uint32_t n_max, double cell, uint32_t n_static, const double *static_aabb_host);
void clapgpu_bp_destroy(clapgpu_bp *bp);
int  clapgpu_bp_collide(void *stream, clapgpu_bp *bp, uint32_t n, const double *aabb,
                        uint32_t *pairs, uint32_t capacity, uint32_t *pair_total,
                        uint32_t *static_pairs, uint32_t static_capacity, uint32_t *static_pair_total);
int  clapgpu_bp_status(void *stream, clapgpu_bp *bp, uint32_t *status);
const double *clapgpu_bp_static_aabb(const clapgpu_bp *bp);            /* the device copy of the static AABBs */

/*
 * near_callback (physics.c:399-449) on candidate pairs (ia in A, ib in B; g1 = A's geom): dCollide for spheres,
 * capsules and axis-aligned boxes (dCollideSpheres, dCollideCapsuleSphere, dCollideCapsuleCapsule with its
 * two-contact parallel case, dCollideSphereBox, dCollideCapsuleBox; reversed like dCollide when only the
 * swapped collider exists) + phys_contact_surface (physics.c:291-330).  One 160-byte record per pair; the record
 * arrays must be 16-byte aligned (CLAPGPU_ERR_INVALID_ARGUMENTS otherwise: they are written as 16-byte pieces).
 * A capsule whose axis touches a box is where ODE switches to dBoxBox: flagged CLAPGPU_CONTACT_DEEP, nc bits 0.
 * body_flags_a / _b (may be NULL): bflags of the body sets behind A / B; touching pairs set CLAPGPU_BODY_HAS_JOINT.
 */
typedef struct clapgpu_geoms {
    uint32_t        n, pad;
    const double   *pos;            /* [n][3] geom position (boxes: the centre of aabb is used) */
    const double   *axis;           /* [n][3] capsule axis (clapgpu_bodies.axis) */
    const double   *radius, *length;/* [n]; length NULL = no capsules */
    const uint8_t  *kind;           /* [n] CLAPGPU_GEOM_*; NULL = sphere when length is 0, else capsule */
    const double   *aabb;           /* [n][6] boxes */
    const double   *material;       /* [n][5] bounce, bounce_vel, mu, soft_erp, soft_cfm (physics.c:77-81); may be NULL */
    const double   *records;        /* optional: [n][8] = (pos, axis, radius, length) per geom, the SAME values as the arrays
                                       above in one 64-byte record (clapgpu_bodies.geom_records); only read for geom sets
                                       without kind / aabb (spheres and capsules: what bodies are).  16-byte aligned */
} clapgpu_geoms;
#define CLAPGPU_CONTACT_DEEP 0x80000000u
typedef struct clapgpu_contact2 {
    double   pos[3], normal[3], depth;
    double   mu, bounce, bounce_vel, soft_erp, soft_cfm;
    uint32_t mode;
    uint32_t nc;                    /* 0, 1, 2, or CLAPGPU_CONTACT_DEEP */
    double   pos2[3], normal2[3], depth2;
} clapgpu_contact2;
int clapgpu_contacts_geoms(void *stream, const clapgpu_geoms *A, const clapgpu_geoms *B, const uint32_t *pairs,
                           const uint32_t *pair_total, uint32_t capacity, clapgpu_contact2 *contacts,
                           uint32_t *contact_total, uint32_t *body_flags_a, uint32_t *body_flags_b);
/*
 * near_callback over BOTH candidate lists of a step -- bodies x bodies and bodies x statics (physics.c:751-753) -- in ONE
 * launch, with no counter fill in front of it: the same records and totals as clapgpu_contacts_geoms(bodies, bodies, ...)
 * followed by clapgpu_contacts_geoms(bodies, statics, ...), body_flags as body_flags_a of both.  `bp` lends eight bytes of
 * scratch (a ticket-and-counts word the kernel leaves at zero); capacities below 2^24 pairs each, else
 * CLAPGPU_ERR_TOO_LARGE (use the two calls).  Two launches and two fills were 62 us of a frame for 47 us of work.
 */
int clapgpu_contacts_geoms_both(void *stream, clapgpu_bp *bp, const clapgpu_geoms *bodies, const clapgpu_geoms *statics,
                                const uint32_t *pairs, const uint32_t *pair_total, uint32_t capacity,
                                clapgpu_contact2 *contacts, uint32_t *contact_total,
                                const uint32_t *static_pairs, const uint32_t *static_pair_total, uint32_t static_capacity,
                                clapgpu_contact2 *static_contacts, uint32_t *static_contact_total, uint32_t *body_flags);

/*
 * phys_body_sweep_capsule (physics.c:559-670) for a batch of sweeps: sweep k marches a probe copy of body
 * sweep_body[k] (a geom of A) along delta[k] in max(2, ceil(|delta| / (radius / 2))) steps and collides it at
 * every step with its candidate geoms cand[cand_first[k] .. cand_first[k + 1]): an index into B (statics) or,
 * with bit 31 set, into A (other bodies; the body itself is skipped).  Out per sweep: frac (best_frac),
 * normal[3], hit (body index, -2 - static index, or -1).  Contacts are taken in candidate order, 16 per
 * step at most (MAX_CONTACTS).  One wavefront per sweep.
 */
int clapgpu_sweep_capsules(void *stream, const clapgpu_geoms *A, const clapgpu_geoms *B, uint32_t n_sweeps,
                           const uint32_t *sweep_body, const float *delta, const uint32_t *cand_first,
                           const uint32_t *cand, float *frac, float *normal, int32_t *hit);

/* ======================================================================== */
/* Characters: the feeder in front of default_update (core/character.c)      */
/* ======================================================================== */

#define CLAPGPU_POS_HISTORY_MAX 8      /* POS_HISTORY_MAX, character.h:21 */

/*
 * struct character's per-frame feeder state (character.h:25-50), SoA over the characters of the
 * scene in list order (scene->characters, character.c:627):
 *   entity[c]         slot of character.entity in the entity SoA
 *   body[c]           index into clapgpu_bodies of its phys body, or -1 (no ENTITY3D_HAS_PHYSICS);
 *                     NULL = no character has one.  Give such bodies body_entity = -1 in
 *                     clapgpu_phys_body_update: characters sync themselves here and keep their rotation
 *   hist_pos[c][8][3], hist_head[c], hist_wrapped[c]   character.history (in/out)
 *   airborne[c]       character.airborne (written by the host's character_move)
 *   moved[c]          out: phys_body_update() reported motion -> host calls character_set_moved()
 *   limbo_height      scene.limbo_height (scene.h:52)
 */
typedef struct clapgpu_characters {
    uint32_t        n;
    float           limbo_height;
    const uint32_t *entity;
    const int32_t  *body;
    float          *hist_pos;
    uint32_t       *hist_head;
    uint8_t        *hist_wrapped;
    const uint8_t  *airborne;
    uint8_t        *moved;
} clapgpu_characters;

/*
 * character_update() (character.c:583-611) for every character, without its tail call: limbo
 * teleport out of the position history, body read-back (position only) + history_push when the
 * body moves.  Writes the entity SoA (pos, CLAPGPU_E_DIRTY) and, on a teleport, the body position
 * (y + yoffset, phys_body_set_position).  Run before clapgpu_entities_update (= the chained
 * orig_update).  b may be NULL when no character has a body.
 */
int clapgpu_characters_update(void *stream, const clapgpu_characters *c, const clapgpu_entities *e,
                              const clapgpu_bodies *b);
/* ... and animated_update's clock (clapgpu_animation_time; now_dev != NULL: clapgpu_animation_time_dev) in the same launch:
 * two per-character passes over different state, both in front of kernels that wait for them.  Same results. */
int clapgpu_characters_update_clock(void *stream, const clapgpu_characters *c, const clapgpu_entities *e, const clapgpu_bodies *b,
                                    const clapgpu_anim_clock *clk, double now, const double *now_dev);

/* ======================================================================== */
/* Clustered lighting: lights x screen tiles bitmask (core/light.c)           */
/* ======================================================================== */

#define CLAPGPU_LIGHTS_MAX 128     /* LIGHTS_MAX, shader_constants.h:8: one bit per slot in an RGBA32UI texel */

/*
 * The slots of struct light (light.h:19-27) that light_grid_compute reads, as device arrays of
 * CLAPGPU_LIGHTS_MAX entries: pos/color/attenuation are float[3] per slot, is_dir is the
 * reference's int flag, active[i] != 0 <=> bit i of light->active is set.
 */
typedef struct clapgpu_lights {
    uint32_t        nr_lights;     /* highest allocated slot + 1 (light.h:46) */
    uint32_t        pad;
    float          *pos;
    const float    *color;
    const float    *attenuation;
    const int32_t  *is_dir;
    const uint32_t *active;
} clapgpu_lights;

/* Tile counts light_grid_update gives the grid (light.c:51-52): ceilf((float)extent / cell).  Host. */
void clapgpu_light_grid_dims(uint32_t width, uint32_t height, uint32_t cell, uint32_t *twidth, uint32_t *theight);

/*
 * light_grid_compute() (light.c:88-154): tiles[gy * twidth + gx][4] (device,
 * the RGBA32UI image the reference uploads with texture_load, light.c:150-153) gets bit idx set
 * iff light idx is active and directional, or is a point light in front of the far plane whose
 * screen-space disc (radius from light_get_radius, light.c:301-309) reaches one of the tile's four
 * corners.  view_mx / proj_mx: the main subview's matrices (host, column-major).
 */
int clapgpu_light_grid_compute(void *stream, const clapgpu_lights *lights, const float view_mx[16],
                               const float proj_mx[16], uint32_t width, uint32_t height, uint32_t cell,
                               uint32_t *tiles);

/*
 * default_update's light hand-off (model.c:1689-1694): carrier k = entity carrier_entity[k] holding
 * light slot carrier_light[k] at offset carrier_off[k] (e->light_idx, e->light_off).  A carrier
 * without a parent whose CLAPGPU_E_DIRTY is set (or any, under CLAPGPU_UPDATE_ALL_DIRTY) writes
 * pos + offset into lights->pos[slot] if the slot is active (light_set_pos, light.c:473-480);
 * carriers apply in list order.  Call BEFORE clapgpu_entities_update, which clears the dirty flags.
 */
int clapgpu_lights_from_entities(void *stream, const clapgpu_entities *e, uint32_t mode, uint32_t n_carriers,
                                 const uint32_t *carrier_entity, const int32_t *carrier_light,
                                 const float *carrier_off, const clapgpu_lights *lights);

/* ======================================================================== */
/* One frame (core/clap.c:551-665)                                           */
/* ======================================================================== */

/*
 * Everything clap_frame() does on the batched path, as one C call that issues the launches in the reference's
 * order on `stream` -- phys_step's substeps (broadphase x2, contact records, world step), character hooks, body
 * read-back + rotation push + light hand-off, the entity update with the main view's cull, animation clock + pose +
 * skinning, particles, the light grid, the ordered visible list + LOD pick -- without reading anything back.
 * Every pointer except `entities` may be NULL: that part of the frame is skipped.  The descriptor holds no state: a
 * caller builds it once and reuses it; capturing the call in a HIP graph (now_dev for the clock) replays the frame
 * with one launch.
 */
typedef struct clapgpu_frame {
    /* entities: tile layout (tile_row_start, device) or level-major (level_start, host) */
    const clapgpu_entities   *entities;
    const uint32_t           *tile_row_start;  uint32_t n_tiles;
    const uint32_t           *level_start;     uint32_t n_levels;
    const clapgpu_frustum    *frustum;         /* main view: fused cull; NULL = no cull, no visible list */
    /* physics (physics.c:746-812) */
    const clapgpu_bodies     *bodies;
    const clapgpu_world      *world;
    clapgpu_bp               *bp;
    uint32_t *pairs, pair_capacity, *pair_total;
    uint32_t *static_pairs, static_pair_capacity, *static_pair_total;
    const clapgpu_geoms      *body_geoms, *static_geoms;
    clapgpu_contact2 *contacts, *static_contacts;
    uint32_t *contact_total, *static_contact_total;
    uint32_t n_body_links; const uint32_t *link_body, *link_entity;
    /* character feeder (character.c:583-611) */
    const clapgpu_characters *characters;
    /* lights: hand-off from their carrier entities, then the tile masks */
    const clapgpu_lights     *lights;
    uint32_t n_light_carriers; const uint32_t *carrier_entity; const int32_t *carrier_light; const float *carrier_offset;
    uint32_t light_width, light_height, light_cell; uint32_t *light_tiles;
    const float              *view_mx, *proj_mx;     /* HOST, 16 floats each */
    /* skeletal animation */
    const clapgpu_anim_clock *anim_clock;      const double *now_dev;   /* device clock for graph replay, or NULL: `now` */
    const clapgpu_skeleton   *skeleton;
    const clapgpu_animations *animations;
    const clapgpu_pose_batch *pose;
    const clapgpu_skin_batch *skin;
    /* particles */
    const clapgpu_particles  *particles;
    /* render-pass glue */
    uint32_t index_base; uint32_t *visible, *visible_count; void *visible_scratch;
    float cam_pos[3]; const int32_t *force_lod; int32_t *cur_lod, *draw_lod;
    uint32_t flags;                            /* CLAPGPU_FRAME_* */
} clapgpu_frame;

/* Default (0): everything on the caller's stream in the reference's order.  CLAPGPU_FRAME_OVERLAP: the frame's three
 * independent chains -- physics -> entity update; animation clock -> pose -> skinning; particles -- run on the caller's
 * stream and two helper streams forked from and joined into it by events (graph capture records the same edges); joint
 * world positions, light grid, visible list and LOD pick follow the join.  Results are identical either way; on one
 * MI355X the overlapped frame is NOT shorter (0.64 against 0.62 ms at BASELINE sizes: csrc/frame.hip). */
#define CLAPGPU_FRAME_OVERLAP 1u
/* CLAPGPU_FRAME_PREBIN: every substep's body step also bins the boxes it writes for the NEXT broadphase pass
 * (clapgpu_bodies_step_prebin): one launch less per substep.  The caller promises that nothing but this frame call writes
 * bodies->aabb between frames (or calls clapgpu_bp_invalidate after doing so). */
#define CLAPGPU_FRAME_PREBIN  2u

int clapgpu_frame_issue(void *stream, const clapgpu_frame *f, double now, uint32_t physics_substeps);

#ifdef __cplusplus
}
#endif
#endif /* CLAPGPU_H */
