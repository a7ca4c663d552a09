/*
 * clapgpu_scene.c -- C host mirror of a CLAP model queue over libclapgpu (see
 * include/clapgpu_scene.h).  Plain C11; owns the handle table, the tile layout
 * (C counterpart of clap_amd/tiler.py), the host staging arrays and the device SoA.
 *
 * Reference structures mirrored: struct mq / model3dtx / entity3d lists (model.h:334,222,377),
 * transform_t (transform.h:8-12), entity3d.parent / seq / parent_seq (model.h:402-405),
 * entity3d_flags (model.h:293-312).
 */
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <stdint.h>
#include <stdio.h>
#include <time.h>
#include "clapgpu_scene.h"

#define WAVE 64u

struct ent {
    float    pos_scale[4];
    float    rot[4];
    uint32_t flags;          /* entity3d_flags bits, no DIRTY */
    uint32_t parent;         /* handle or CLAPGPU_NO_ENTITY */
    uint32_t model;
    uint32_t slot;
    void    *user;
    uint8_t  live, dirty, attached;   /* attached: rides a joint of its parent (e->parent_joint, model.c:1626-1641) */
    uint8_t  keep;                    /* clapgpu_scene_entity_keep: a standing host reader, exported whenever rebuilt */
    uint32_t n_children;              /* live entities whose parent this is (an entity with children cannot be deleted in place) */
    int32_t  force_lod, cur_lod;      /* entity3d.force_lod / .cur_lod (model.h:415-416; entity3d_set_lod, model.c:593-609) */
};

/* the flags word of the upload image: the entity3d bits + what only the device knows */
static inline uint32_t img_flags(const struct ent *e, int xform_updated)
{
    return e->flags | (e->attached ? CLAPGPU_E_JOINT_ATTACHED : 0) | (xform_updated ? CLAPGPU_E_DIRTY : 0);
}

struct clapgpu_scene {
    struct ent *e;  uint32_t n_handles, cap_handles;
    uint32_t   *free_list;  uint32_t n_free, cap_free;
    uint32_t   *dead_list;  uint32_t n_dead, cap_dead;           /* deleted since the last re-tile: handles not reusable yet */
    uint32_t   *dirty_list; uint32_t n_dirty, cap_dirty;
    float      *models;     uint32_t n_models, cap_models;       /* [m][8] model_table rows */
    int         topology_dirty, models_dirty, tiled, bulk_dirty;

    /* layout */
    uint32_t    n_slots, n_rows, n_tiles, n_levels;
    uint32_t   *slot_handle;                                     /* slot -> handle or NO_ENTITY */
    uint32_t   *tile_row_start_host, *level_start_host;

    /* host staging (slot order) */
    float      *h_pos_scale, *h_rot, *h_mx, *h_inv, *h_aabb, *h_center;
    int32_t    *h_parent, *h_model;
    uint32_t   *h_flags;
    uint64_t   *h_mask, *h_rebuilt, *h_inside;
    void      **slot_user;                                       /* slot -> the entity's user pointer (NULL: padding) */
    uint32_t    cap_slots;
    uint32_t    up_lo, up_hi, n_staged;                           /* slots whose upload image was written since the last frame */
    /* camera bounding-volume points (default_update's pick, model.c:1703-1713) */
    int         bv_on, bv_has_ctl; float bv_cam[3], bv_ctl[3]; uint32_t bv_ctl_handle;
    clapgpu_bv_query bvq; uint64_t *d_bv_result;

    /* the arrays above that cross PCIe every frame are carved out of two page-locked slabs that
     * mirror two device slabs: one copy up (pos_scale | rot | flags), one copy down
     * (mx | inv_mx | aabb | center | vis_mask) */
    void       *h_in, *h_out, *d_in, *d_out;
    size_t      in_bytes, out_bytes;
    /* small scenes (zero_copy): no copy calls and no blocking wait in a frame, and with the tile layout ONE launch:
     * the upload image and the result slab are device-mapped, the frame's touched slots are flagged in h_touched, and
     * clapgpu_entities_update_tiles_hostio reads the flagged inputs from the image, writes what it rebuilds (and the
     * masks) into h_out as well and raises *h_done, which mq_update polls.  With the level layout (a tree wider than a
     * wavefront) the touched records travel as a mapped list scattered by clapgpu_entities_apply_inputs and the results
     * come back through clapgpu_entities_export_rebuilt.  At a testbed-sized scene (10 k entities) the three copies'
     * fixed latencies and the blocking wait were 0.13 of a 0.15 ms device step around a 15-30 us kernel. */
    int         zero_copy;
    uint32_t    zero_copy_max_slots;
    clapgpu_entity_input *h_list; void *d_list; uint32_t cap_list;    /* mapped: host pointer / device alias */
    void       *d_out_host;                                            /* device alias of h_out */
    void       *d_in_host;                                             /* device alias of h_in (zero_copy: the image is mapped) */
    uint64_t   *h_touched;                                             /* behind the image: one bit per slot written since the last frame */
    uint32_t   *h_done, *d_done, *d_counter, frame_id;
    /* joint attachments (clapgpu_scene_attached_update): table + the two matrix pools + the kernel's work space */
    void       *h_att, *d_att; size_t att_bytes; uint32_t cap_att; int att_mapped;
    float      *d_att_local;
    clapgpu_frustum last_frustum; int have_frustum;
    /* export policy (clapgpu_scene_set_export): with EXPORT_DRAWN a one-launch frame writes back only the rebuilt rows
     * somebody reads (drawn, containing a bounding-volume point, kept); the others go stale in h_out -- the device arrays
     * hold them -- and are fetched when they come into view or when asked for (clapgpu_scene_fetch) */
    int         export_drawn;
    uint64_t   *h_keep, *d_keep; int keep_dirty;                       /* slot order; the device copy follows before a launch */
    uint64_t   *h_exported;                                            /* mapped, behind the three masks of h_out */
    uint64_t   *h_stale, *h_fetched; uint32_t n_stale_words, n_fetched, fetch_serial; /* plain host memory, cap_slots / 64 + 2 words */
    uint64_t   *h_select; void *d_select;                              /* mapped: the rows a fetch asks for */
    int         fetch_accumulate;                                      /* fetch_rows adds to the rows this mq_update's launch already brought over */
    uint64_t   *d_stale;                                               /* device twin of h_stale, kept by the launches themselves (clapgpu_entities_hostio.stale_mask) */
    /* the layout edited in place (clapgpu_scene_entity_new_placed / _delete_placed): a queue whose make-up changes by a few
     * entities a frame keeps its tiles; a re-tile is the fall-back */
    uint32_t    max_depth;                                             /* rows of the deepest tree at the last re-tile */
    uint32_t    grow_tile;                                             /* the tile new roots go into (NO_ENTITY: none yet) */
    uint32_t    cap_tiles;                                             /* entries tile_row_start_host can hold, minus one */
    int         incremental;                                           /* clapgpu_scene_set_incremental: re-tiles leave room for edits */
    uint32_t   *free_roots; uint32_t n_free_roots, cap_free_roots;     /* first-row slots freed by deletions */
    uint32_t   *raw_words; uint32_t n_raw, cap_raw, raw_lo, raw_hi;    /* words of h_touched set outside the dirty list (tombstones); their slot range */
    uint32_t   *edits; uint32_t n_edits, cap_edits, edit_lo, edit_hi;  /* slots whose parent / model the device has not been given yet */
    clapgpu_entity_place *h_place; void *d_place; uint32_t cap_place;  /* ... as the mapped list clapgpu_entities_place takes */
    uint32_t    grown_from, tiles_from;                                /* first slot / tile appended since the device last saw the layout (NO_ENTITY: none) */
    uint32_t   *limbo; uint32_t n_limbo, cap_limbo;                    /* handles deleted in place: reusable once the frame's dirty list is spent */

    /* device */
    clapgpu_entities d;
    uint32_t   *d_tile_row_start;
    float      *d_models; uint32_t d_models_cap;
    int         have_results;
    uint32_t    layout_gen;

    /* the render passes' LOD pick and draw list (clapgpu_scene_select_lod): force_lod / cur_lod in slot order on both
     * sides (the host copy follows every pick, so a range of it can be pushed at any time), the ordered visible list
     * and the LOD each entry is drawn with */
    int32_t    *h_force_lod, *h_cur_lod;  int32_t *d_force_lod, *d_cur_lod;
    uint32_t   *d_visible, *d_visible_count; int32_t *d_draw_lod; void *d_vis_scratch;
    uint32_t   *h_draw_slot; int32_t *h_draw_lod; uint32_t *h_visible_count;     /* page-locked */
    uint32_t    lod_cap, lod_lo, lod_hi, n_draw;                                  /* [lod_lo, lod_hi): host values not on the device yet */
    uint32_t    lod_layout_gen;
    int         lod_sync_by_caller;                                               /* clapgpu_scene_set_lod_sync */
    /* a small scene's draw list lands in device-mapped host memory: the two launches write it (and its length) where the host
     * reads it, one wait -- no length copy, wait, list copies, wait (two round trips of ~35 us around two ~8 us launches) */
    int         lod_mapped; void *a_draw_slot, *a_draw_lod, *a_visible_count;

    /* the frame's other views (clapgpu_scene_set_views): frusta + device planes in xv (what clapgpu_entities.views points at),
     * the host copies of the masks (device-mapped when the scene is zero-copy: the one-launch frame writes them itself) and
     * the union of every view's mask for the export policy's fetches */
    clapgpu_views xv; uint32_t xv_want, xv_cap_slots; int xv_mapped;
    uint64_t   *h_xv_mask[CLAPGPU_EXTRA_VIEWS_MAX]; void *a_xv_mask[CLAPGPU_EXTRA_VIEWS_MAX];
    uint64_t   *h_xv_union;

    /* a caller's thread pool for the re-tile's passes over every handle / slot (clapgpu_scene_set_parallel_for) */
    clapgpu_scene_parallel_for par_for; int par_threads;
};

#define CK(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

static double scene_now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

static struct ent *get(const clapgpu_scene *s, uint32_t h)
{
    return (s && h < s->n_handles && s->e[h].live) ? &s->e[h] : NULL;
}

/* dirty bit 0: queued for upload; bit 1: xform.updated (transform_set_updated, transform.c:21-24) */
static void mark_dirty(clapgpu_scene *s, uint32_t h, int xform_updated)
{
    if (!s->e[h].dirty) {
        if (s->n_dirty == s->cap_dirty) {
            const uint32_t cap = s->cap_dirty ? 2 * s->cap_dirty : 1024;
            uint32_t *q = realloc(s->dirty_list, cap * sizeof(uint32_t));
            if (!q) {                                            /* out of memory: the next frame uploads everything instead */
                s->e[h].dirty |= xform_updated ? 3 : 1;
                s->topology_dirty = 1;
                return;
            }
            s->dirty_list = q; s->cap_dirty = cap;
        }
        s->dirty_list[s->n_dirty++] = h;
    }
    s->e[h].dirty |= xform_updated ? 3 : 1;
    /* the layout stands: write the upload image now, while the caller's data is hot, instead of in a second pass */
    if (!s->topology_dirty && s->h_in && s->e[h].slot < s->n_slots) {
        const struct ent *e = &s->e[h];
        const uint32_t slot = e->slot;
        memcpy(s->h_pos_scale + 4 * (size_t)slot, e->pos_scale, 16);
        memcpy(s->h_rot + 4 * (size_t)slot, e->rot, 16);
        s->h_flags[slot] = img_flags(e, e->dirty & 2);
        if (slot < s->up_lo) s->up_lo = slot;
        if (slot >= s->up_hi) s->up_hi = slot + 1;
    }
}

/* the mapped side of a one-launch small frame: the image (and its touched bits) in, the result slab and the word out */
static void scene_hostio(clapgpu_scene *s, clapgpu_entities_hostio *io, int with_inputs, int filtered)
{
    const size_t cn = s->cap_slots;
    const char *mi = s->d_in_host;
    char *mo = s->d_out_host;
    memset(io, 0, sizeof(*io));
    io->pos_scale = (const float *)mi; io->rot = (const float *)(mi + cn * 16); io->flags = (const uint32_t *)(mi + cn * 32);
    io->touched = with_inputs ? (const uint64_t *)(mi + cn * 36) : NULL;
    io->mx = (float *)mo; io->inv_mx = (float *)(mo + cn * 64); io->aabb = (float *)(mo + cn * 128);
    io->center = (float *)(mo + cn * 152); io->vis_mask = (uint64_t *)(mo + cn * 164);
    io->rebuilt_mask = io->vis_mask + (cn / 64 + 2);
    io->inside_mask = s->bv_on ? io->rebuilt_mask + (cn / 64 + 2) : NULL;
    io->exported_mask = io->rebuilt_mask + 2 * (cn / 64 + 2);
    io->keep_mask = filtered ? s->d_keep : NULL;
    io->stale_mask = s->d_stale;
    io->counter = s->d_counter; io->done = s->d_done; io->done_value = ++s->frame_id;
}

/* After a one-launch frame: rows the launch rebuilt but did not write back are stale in h_out, rows it wrote are fresh. */
static void stale_after_launch(clapgpu_scene *s)
{
    const size_t words = s->n_slots / 64;
    uint32_t nz = 0, late_rows = 0;
    /* fetched_mask names the rows of THIS call only (fetch_rows): here the ones the launch itself brought over although it
     * did not rebuild them -- stale rows that have a reader now (exported, not rebuilt); exported_mask goes back to "rebuilt
     * and written", which is what a caller scatters as this frame's rebuilds */
    if (s->n_fetched) { memset(s->h_fetched, 0, words * 8); s->n_fetched = 0; }
    for (size_t w = 0; w < words; w++) {
        const uint64_t ex = s->h_exported[w], rb = s->h_rebuilt[w], late = ex & ~rb;
        const uint64_t st = (s->h_stale[w] | rb) & ~ex;
        s->h_stale[w] = st;
        nz += st != 0;
        if (late) {
            s->h_fetched[w] = late;
            s->h_exported[w] = ex & rb;
            late_rows += (uint32_t)__builtin_popcountll(late);
        }
    }
    s->n_stale_words = nz;
    if (late_rows) { s->n_fetched = late_rows; s->fetch_serial++; }
}

static int apply_edits(clapgpu_scene *s);

/* rows = stale & want (NULL: every stale row): over from the device arrays into h_out, named in h_fetched */
static int fetch_rows(clapgpu_scene *s, const uint64_t *w0, const uint64_t *w1, const uint64_t *w2)
{
    const size_t words = s->n_slots / 64;
    /* fetched_mask names the rows of THIS call only: a caller copies them out once (fetch_serial says whether there is
     * anything new); rows of an earlier call may since have been superseded on the host */
    if (s->n_fetched && !s->fetch_accumulate) { memset(s->h_fetched, 0, words * 8); s->n_fetched = 0; }
    if (!s->n_stale_words || !s->h_select) return CLAPGPU_OK;
    uint32_t cnt = 0;
    for (size_t w = 0; w < words; w++) {
        uint64_t sel = s->h_stale[w];
        if (sel && (w0 || w1 || w2)) sel &= (w0 ? w0[w] : 0) | (w1 ? w1[w] : 0) | (w2 ? w2[w] : 0);
        s->h_select[w] = sel;
        cnt += (uint32_t)__builtin_popcountll(sel);
    }
    if (!cnt) return CLAPGPU_OK;
    CK(apply_edits(s));
    char *mo = s->d_out_host;
    const size_t cn = s->cap_slots;
    clapgpu_entities_export x = { .mx = (float *)mo, .inv_mx = (float *)(mo + cn * 64), .aabb = (float *)(mo + cn * 128),
                                  .center = (float *)(mo + cn * 152) };
    x.counter = s->d_counter; x.done = s->d_done; x.done_value = ++s->frame_id;
    x.stale_mask = s->d_stale;
    CK(clapgpu_entities_export_rows(NULL, &s->d, &x, s->d_select));
    CK(clapgpu_wait_word(s->h_done, s->frame_id, NULL));
    uint32_t nz = 0;
    for (size_t w = 0; w < words; w++) {
        s->h_fetched[w] = s->fetch_accumulate ? (s->h_fetched[w] | s->h_select[w]) : s->h_select[w];
        s->h_stale[w] &= ~s->h_select[w];
        nz += s->h_stale[w] != 0;
    }
    s->n_stale_words = nz;
    s->n_fetched = s->fetch_accumulate ? s->n_fetched + cnt : cnt;
    s->fetch_serial++;
    return CLAPGPU_OK;
}

static void free_views(clapgpu_scene *s)
{
    for (int v = 0; v < CLAPGPU_EXTRA_VIEWS_MAX; v++) {
        if (s->xv.vis_mask[v]) clapgpu_free(s->xv.vis_mask[v]);
        if (s->xv.vis_row_pop[v]) clapgpu_free(s->xv.vis_row_pop[v]);
        if (s->h_xv_mask[v]) clapgpu_host_free(s->h_xv_mask[v]);
        s->xv.vis_mask[v] = NULL; s->xv.vis_row_pop[v] = NULL; s->xv.host_vis_mask[v] = NULL;
        s->h_xv_mask[v] = NULL; s->a_xv_mask[v] = NULL;
    }
    free(s->h_xv_union); s->h_xv_union = NULL;
    s->xv_cap_slots = 0;
}

/* the planes of the extra views, for the current capacity */
static int ensure_views(clapgpu_scene *s)
{
    if (!s->xv_want) { s->xv.n = 0; return CLAPGPU_OK; }
    if (s->xv_cap_slots != s->cap_slots || s->xv_mapped != s->zero_copy) {
        free_views(s);
        const size_t mw = (size_t)s->cap_slots / 64 + 2;
        s->xv_mapped = s->zero_copy;
        for (uint32_t v = 0; v < CLAPGPU_EXTRA_VIEWS_MAX; v++) {
            CK(clapgpu_malloc((void **)&s->xv.vis_mask[v], mw * 8));
            CK(clapgpu_malloc((void **)&s->xv.vis_row_pop[v], ((size_t)s->cap_slots / 64 + 16) / 16 * 16));
            CK(clapgpu_memset(s->xv.vis_mask[v], 0, mw * 8, NULL));
            if (s->xv_mapped) CK(clapgpu_host_malloc_mapped((void **)&s->h_xv_mask[v], &s->a_xv_mask[v], mw * 8));
            else CK(clapgpu_host_malloc((void **)&s->h_xv_mask[v], mw * 8));
            memset(s->h_xv_mask[v], 0, mw * 8);
        }
        s->h_xv_union = calloc(mw, 8);
        if (!s->h_xv_union) return CLAPGPU_ERR_NOMEM;
        s->xv_cap_slots = s->cap_slots;
    }
    s->xv.n = s->xv_want;
    return CLAPGPU_OK;
}

/* what ANY view of the last launch draws: the main mask alone without extra views */
static const uint64_t *views_union(clapgpu_scene *s)
{
    if (!s->xv.n || !s->h_xv_union) return s->h_mask;
    const size_t words = s->n_slots / 64;
    for (size_t w = 0; w < words; w++) {
        uint64_t m = s->h_mask[w];
        for (uint32_t v = 0; v < s->xv.n; v++) m |= s->h_xv_mask[v][w];
        s->h_xv_union[w] = m;
    }
    return s->h_xv_union;
}

/* the extra views' masks of a launch that did not write them to the host itself */
static int download_views(clapgpu_scene *s)
{
    for (uint32_t v = 0; v < s->xv.n; v++)
        CK(clapgpu_memcpy_d2h(s->h_xv_mask[v], s->xv.vis_mask[v], ((size_t)s->n_slots / 64) * 8, NULL));
    return CLAPGPU_OK;
}

int clapgpu_scene_set_views(clapgpu_scene *s, uint32_t n_extra, const clapgpu_frustum *extra)
{
    if (!s || n_extra > CLAPGPU_EXTRA_VIEWS_MAX || (n_extra && !extra)) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    s->xv_want = n_extra;
    for (uint32_t v = 0; v < n_extra; v++) s->xv.frustum[v] = extra[v];
    if (!n_extra) s->xv.n = 0;
    return CLAPGPU_OK;
}

int clapgpu_scene_create(clapgpu_scene **out, int device)
{
    if (!out) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    CK(clapgpu_init(device));
    clapgpu_scene *s = calloc(1, sizeof(*s));
    if (!s) return CLAPGPU_ERR_NOMEM;
    s->topology_dirty = 1;
    s->grow_tile = s->grown_from = s->tiles_from = CLAPGPU_NO_ENTITY;
    s->zero_copy_max_slots = CLAPGPU_SCENE_ZERO_COPY_SLOTS;
    const char *zc = getenv("CLAPGPU_SCENE_ZERO_COPY_SLOTS");   /* tuning knob: 0 = always copy */
    if (zc) s->zero_copy_max_slots = (uint32_t)strtoul(zc, NULL, 0);
    *out = s;
    return CLAPGPU_OK;
}

void clapgpu_scene_set_zero_copy_slots(clapgpu_scene *s, uint32_t max_slots)
{
    if (!s || s->zero_copy_max_slots == max_slots) return;
    s->zero_copy_max_slots = max_slots;
    s->cap_slots = 0;                                   /* the slabs are re-made in the new mode by the next re-tile */
    s->topology_dirty = 1;
}

int clapgpu_scene_is_zero_copy(const clapgpu_scene *s) { return s ? s->zero_copy : 0; }

static void free_lod(clapgpu_scene *s)
{
    void *dev[] = { s->d_force_lod, s->d_cur_lod, s->d_visible, s->d_visible_count, s->d_draw_lod, s->d_vis_scratch };
    for (unsigned i = 0; i < sizeof(dev) / sizeof(dev[0]); i++)
        if (dev[i]) clapgpu_free(dev[i]);
    void *host[] = { s->h_draw_slot, s->h_draw_lod, s->h_visible_count };
    for (unsigned i = 0; i < sizeof(host) / sizeof(host[0]); i++)
        if (host[i]) clapgpu_host_free(host[i]);
    free(s->h_force_lod); free(s->h_cur_lod);
    s->d_force_lod = s->d_cur_lod = s->d_draw_lod = NULL; s->d_visible = s->d_visible_count = NULL; s->d_vis_scratch = NULL;
    s->h_draw_slot = NULL; s->h_draw_lod = NULL; s->h_visible_count = NULL; s->h_force_lod = s->h_cur_lod = NULL;
    s->lod_cap = 0; s->n_draw = 0;
}

static void free_device(clapgpu_scene *s)
{
    free_lod(s);
    void *p[] = { s->d_in, s->d_out, (void *)s->d.parent, (void *)s->d.model, s->d.seqs, s->d.vis_row_pop,
                  s->d_tile_row_start };
    for (unsigned i = 0; i < sizeof(p) / sizeof(p[0]); i++)
        if (p[i]) clapgpu_free(p[i]);
    memset(&s->d, 0, sizeof(s->d));
    s->d_in = s->d_out = NULL;
    s->d_tile_row_start = NULL;
}

void clapgpu_scene_destroy(clapgpu_scene *s)
{
    if (s) free_views(s);
    if (!s) return;
    free_device(s);
    if (s->d_models) clapgpu_free(s->d_models);
    free(s->e); free(s->free_list); free(s->dead_list); free(s->dirty_list); free(s->models); free(s->slot_handle);
    free(s->tile_row_start_host); free(s->level_start_host);
    if (s->h_in) clapgpu_host_free(s->h_in);
    if (s->h_out) clapgpu_host_free(s->h_out);
    if (s->h_parent) clapgpu_host_free(s->h_parent);
    if (s->h_model) clapgpu_host_free(s->h_model);
    free(s->slot_user);
    if (s->d_bv_result) clapgpu_free(s->d_bv_result);
    if (s->h_list) clapgpu_host_free(s->h_list);
    if (s->h_done) clapgpu_host_free(s->h_done);
    if (s->d_counter) clapgpu_free(s->d_counter);
    if (s->h_att) clapgpu_host_free(s->h_att);
    if (s->d_att && !s->att_mapped) clapgpu_free(s->d_att);
    if (s->d_att_local) clapgpu_free(s->d_att_local);
    if (s->d_keep) clapgpu_free(s->d_keep);
    if (s->h_select) clapgpu_host_free(s->h_select);
    free(s->h_keep); free(s->h_stale); free(s->h_fetched); free(s->free_roots); free(s->raw_words); free(s->edits); free(s->limbo);
    if (s->h_place) clapgpu_host_free(s->h_place);
    if (s->d_stale) clapgpu_free(s->d_stale);
    free(s);
}

int clapgpu_scene_model_new(clapgpu_scene *s, const float aabb[6], int skip_aabb, uint32_t *model)
{
    if (!s || !aabb || !model) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (s->n_models == s->cap_models) {
        const uint32_t cap = s->cap_models ? 2 * s->cap_models : 16;
        float *q = realloc(s->models, (size_t)cap * 8 * sizeof(float));
        if (!q) return CLAPGPU_ERR_NOMEM;
        s->models = q; s->cap_models = cap;
    }
    float *row = s->models + 8 * (size_t)s->n_models;
    uint32_t skip = skip_aabb ? 1u : 0u;
    row[0] = aabb[0]; row[1] = aabb[1]; row[2] = aabb[2];
    memcpy(&row[3], &skip, 4);
    row[4] = aabb[3]; row[5] = aabb[4]; row[6] = aabb[5]; row[7] = 0.f;     /* lod_min = lod_max = 0 until clapgpu_scene_model_lods */
    *model = s->n_models++;
    s->models_dirty = 1;
    return CLAPGPU_OK;
}

int clapgpu_scene_model_lods(clapgpu_scene *s, uint32_t model, unsigned int lod_min, unsigned int lod_max)
{
    if (!s || model >= s->n_models || lod_min > 255 || lod_max > 255) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const uint32_t bits = lod_min | (lod_max << 8);
    memcpy(&s->models[8 * (size_t)model + 7], &bits, 4);
    s->models_dirty = 1;
    return CLAPGPU_OK;
}

static int new_handle(clapgpu_scene *s, uint32_t model, void *user, uint32_t *handle)
{
    uint32_t h;
    if (s->n_free) {
        h = s->free_list[--s->n_free];
    } else {
        if (s->n_handles == s->cap_handles) {
            const uint32_t cap = s->cap_handles ? 2 * s->cap_handles : 1024;
            struct ent *q = realloc(s->e, (size_t)cap * sizeof(struct ent));
            if (!q) return CLAPGPU_ERR_NOMEM;
            s->e = q; s->cap_handles = cap;
        }
        h = s->n_handles++;
    }
    struct ent *e = &s->e[h];
    memset(e, 0, sizeof(*e));
    e->pos_scale[3] = 1.f;                              /* model.c:1738 scale = 1 */
    e->rot[3] = 1.f;                                    /* transform_init: identity quat */
    e->flags = CLAPGPU_E_ALIVE | CLAPGPU_E_VISIBLE;
    e->parent = CLAPGPU_NO_ENTITY;
    e->model = model;
    e->user = user;
    e->live = 1;
    e->force_lod = -1;                                  /* entity3d_make, model.c:1741; cur_lod 0 */
    *handle = h;
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_new(clapgpu_scene *s, uint32_t model, void *user, uint32_t *handle)
{
    if (!s || !handle || model >= s->n_models) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    CK(new_handle(s, model, user, handle));
    s->topology_dirty = 1;
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_delete(clapgpu_scene *s, uint32_t handle)
{
    struct ent *e = get(s, handle);
    if (!e) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    /* Its children become roots (like a NULL e->parent) -- found by ONE pass for all of a frame's deletions when the layout
     * is rebuilt (release_dead), not by a pass over every entity per deletion; until then the handle is not handed out again,
     * so a child's parent field cannot come to name a stranger. */
    if (s->n_dead == s->cap_dead) {
        const uint32_t cap = s->cap_dead ? 2 * s->cap_dead : 256;
        uint32_t *q = realloc(s->dead_list, (size_t)cap * sizeof(uint32_t));
        if (!q) return CLAPGPU_ERR_NOMEM;
        s->dead_list = q; s->cap_dead = cap;
    }
    e->live = 0;
    if (e->parent != CLAPGPU_NO_ENTITY && e->parent < s->n_handles && s->e[e->parent].live && s->e[e->parent].n_children)
        s->e[e->parent].n_children--;
    s->dead_list[s->n_dead++] = handle;
    s->topology_dirty = 1;
    return CLAPGPU_OK;
}

/* before a re-tile: orphans of the entities deleted since the last one become roots, their handles reusable */
static int release_dead(clapgpu_scene *s)
{
    if (!s->n_dead) return CLAPGPU_OK;
    for (uint32_t h = 0; h < s->n_handles; h++) {
        struct ent *c = &s->e[h];
        if (c->live && c->parent != CLAPGPU_NO_ENTITY && !(c->parent < s->n_handles && s->e[c->parent].live))
            c->parent = CLAPGPU_NO_ENTITY;
    }
    if (s->n_free + s->n_dead > s->cap_free) {
        uint32_t cap = s->cap_free ? s->cap_free : 256;
        while (cap < s->n_free + s->n_dead) cap *= 2;
        uint32_t *q = realloc(s->free_list, (size_t)cap * sizeof(uint32_t));
        if (!q) return CLAPGPU_ERR_NOMEM;
        s->free_list = q; s->cap_free = cap;
    }
    memcpy(s->free_list + s->n_free, s->dead_list, (size_t)s->n_dead * sizeof(uint32_t));
    s->n_free += s->n_dead;
    s->n_dead = 0;
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_set_parent(clapgpu_scene *s, uint32_t handle, uint32_t parent)
{
    struct ent *e = get(s, handle);
    if (!e || (parent != CLAPGPU_NO_ENTITY && (!get(s, parent) || parent == handle)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (e->parent != parent) {
        if (e->parent != CLAPGPU_NO_ENTITY && e->parent < s->n_handles && s->e[e->parent].live && s->e[e->parent].n_children)
            s->e[e->parent].n_children--;
        if (parent != CLAPGPU_NO_ENTITY) s->e[parent].n_children++;
        e->parent = parent;
        s->topology_dirty = 1;
    }
    return CLAPGPU_OK;
}

/* ---- the standing layout edited in place ---------------------------------------------------------------------------------
 * A queue that gains and loses a few entities a frame (pickups, projectiles, effects) would pay for a re-tile -- every
 * entity's depth, a new packing, the whole upload image, every slot moved under the caller -- each time.  These two verbs
 * edit the tile layout where it stands instead: a new root takes a free first-row lane (one a deleted root left, or one of
 * a growth tile appended behind the others), a new child a free lane of the row below its parent in the parent's own tile
 * (the kernel hands a parent's matrix to the next row through registers: that is the only place a child can be), a deleted
 * leaf becomes a lane that is not ALIVE.  No other entity moves: slots, masks and the caller's per-slot state stand.
 * Either verb returns CLAPGPU_ERR_NOT_SUPPORTED, having changed nothing, when the edit does not fit (no free lane, no row
 * below, out of capacity, a layout that is not the one-launch tile form): the caller then uses the plain verbs and the next
 * mq_update re-tiles.  The device is told with the next mq_update (the new lanes' inputs through the touched bits like any
 * moved entity's, parent / model indices by a small copy): until then results for such an entity are not defined. */
void clapgpu_scene_set_parallel_for(clapgpu_scene *s, clapgpu_scene_parallel_for fn, int threads)
{
    if (!s) return;
    s->par_for = threads > 1 ? fn : NULL;
    s->par_threads = threads;
}

/* a pass over [0, n) on the caller's pool, or right here */
static void run_ranges(const clapgpu_scene *s, void (*fn)(void *, uint32_t, uint32_t), void *ctx, uint32_t n)
{
    static uint32_t par_min;
    if (!par_min) {
        const char *v = getenv("CLAPGPU_SCENE_PAR_MIN");         /* tuning knob; the tests set 1 */
        par_min = v && atoi(v) > 0 ? (uint32_t)atoi(v) : 16384u;
    }
    if (s->par_for && n >= par_min) s->par_for(fn, ctx, n, s->par_threads);
    else fn(ctx, 0, n);
}

void clapgpu_scene_set_incremental(clapgpu_scene *s, int on)
{
    if (s) s->incremental = on != 0;                     /* from the next re-tile on */
}

static int push_list(uint32_t **arr, uint32_t *n, uint32_t *cap, uint32_t v)
{
    if (*n == *cap) {
        const uint32_t c = *cap ? 2 * *cap : 64;
        uint32_t *q = realloc(*arr, (size_t)c * sizeof(uint32_t));
        if (!q) return CLAPGPU_ERR_NOMEM;
        *arr = q; *cap = c;
    }
    (*arr)[(*n)++] = v;
    return CLAPGPU_OK;
}

static int layout_editable(const clapgpu_scene *s)
{
    return !s->topology_dirty && s->tiled && s->zero_copy && s->have_results && s->h_in && s->n_tiles && s->n_models;
}

static uint32_t tile_of_row(const clapgpu_scene *s, uint32_t row)
{
    uint32_t lo = 0, hi = s->n_tiles;                    /* tile_row_start_host[lo] <= row < tile_row_start_host[hi] */
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s->tile_row_start_host[mid] <= row) lo = mid; else hi = mid;
    }
    return lo;
}

static int free_lane(const clapgpu_scene *s, uint32_t row)
{
    const uint32_t *sh = s->slot_handle + (size_t)row * WAVE;
    for (int l = 0; l < (int)WAVE; l++)
        if (sh[l] == CLAPGPU_NO_ENTITY) return l;
    return -1;
}

static void touch_raw(clapgpu_scene *s, uint32_t slot)
{
    s->h_touched[slot >> 6] |= 1ull << (slot & 63);
    if (!s->n_raw || slot < s->raw_lo) s->raw_lo = slot;
    if (!s->n_raw || slot + 1 > s->raw_hi) s->raw_hi = slot + 1;
    if (push_list(&s->raw_words, &s->n_raw, &s->cap_raw, slot >> 6))
        s->bulk_dirty = 1;                               /* cannot remember the word: the next frame clears them all */
}

/* a tile of max_depth (+ spare) empty rows behind the others; the device hears of it in apply_edits() */
static int append_tile(clapgpu_scene *s)
{
    const uint32_t rows = s->max_depth + ((s->incremental && s->max_depth > 1) ? 1 : 0);
    if (!rows || (uint64_t)(s->n_rows + rows) * WAVE > s->cap_slots) return CLAPGPU_ERR_NOT_SUPPORTED;
    if (s->n_tiles + 1 > s->cap_tiles) {
        const uint32_t cap = 2 * s->cap_tiles + 16;
        uint32_t *q = realloc(s->tile_row_start_host, ((size_t)cap + 1) * 4);
        if (!q) return CLAPGPU_ERR_NOMEM;
        s->tile_row_start_host = q; s->cap_tiles = cap;
    }
    const uint32_t first = s->n_slots, end = first + rows * WAVE;
    static const float id[4] = { 0, 0, 0, 1 };
    for (uint32_t i = first; i < end; i++) {
        s->slot_handle[i] = CLAPGPU_NO_ENTITY; s->slot_user[i] = NULL;
        memcpy(s->h_pos_scale + 4 * (size_t)i, id, 16);
        memcpy(s->h_rot + 4 * (size_t)i, id, 16);
        s->h_parent[i] = -1; s->h_model[i] = 0; s->h_flags[i] = 0;
    }
    if (s->lod_cap >= end && s->lod_layout_gen == s->layout_gen) {
        for (uint32_t i = first; i < end; i++) { s->h_force_lod[i] = -1; s->h_cur_lod[i] = 0; }
        if (first < s->lod_lo) s->lod_lo = first;
        if (end > s->lod_hi) s->lod_hi = end;
    }
    if (s->grown_from == CLAPGPU_NO_ENTITY) { s->grown_from = first; s->tiles_from = s->n_tiles; }
    s->grow_tile = s->n_tiles;
    s->tile_row_start_host[s->n_tiles] = s->n_rows;      /* (it was the end of the last tile already) */
    s->n_tiles++;
    s->n_rows += rows;
    s->tile_row_start_host[s->n_tiles] = s->n_rows;
    s->n_slots = s->n_rows * WAVE;
    return CLAPGPU_OK;
}

/* parent / model indices of the edited slots, and appended tiles, to the device: before anything is launched on the layout */
static int apply_edits(clapgpu_scene *s)
{
    if (s->grown_from == CLAPGPU_NO_ENTITY && !s->n_edits) return CLAPGPU_OK;
    if (s->grown_from != CLAPGPU_NO_ENTITY) {
        const size_t a = s->grown_from, cnt = s->n_slots - a;
        CK(clapgpu_memcpy_h2d((int32_t *)s->d.parent + a, s->h_parent + a, cnt * 4, NULL));
        CK(clapgpu_memcpy_h2d((int32_t *)s->d.model + a, s->h_model + a, cnt * 4, NULL));
        CK(clapgpu_memset(s->d.flags + a, 0, cnt * 4, NULL));           /* nothing ALIVE there until the image says so */
        CK(clapgpu_memset(s->d.seqs + a, 0, cnt * 4, NULL));
        CK(clapgpu_memcpy_h2d(s->d_tile_row_start + s->tiles_from, s->tile_row_start_host + s->tiles_from,
                              ((size_t)s->n_tiles + 1 - s->tiles_from) * 4, NULL));
        s->d.n = s->n_slots;
        s->grown_from = s->tiles_from = CLAPGPU_NO_ENTITY;
    }
    if (s->n_edits) {
        /* the frame's edited lanes as one mapped list, one small launch (two copies and two fills each, queued one behind
         * the other in front of the update, cost a 10 k-entity frame 35 us).  zero_box: a model without a box (skip_aabb)
         * never writes one, so the lane's last tenant's must not stay (a fresh entity3d's is all zeros, and so is every
         * row after a re-tile) */
        if (s->n_edits > s->cap_place) {
            uint32_t cap = s->cap_place ? s->cap_place : 64;
            while (cap < s->n_edits) cap *= 2;
            if (s->h_place) clapgpu_host_free(s->h_place);
            s->h_place = NULL; s->cap_place = 0;
            CK(clapgpu_host_malloc_mapped((void **)&s->h_place, &s->d_place, (size_t)cap * sizeof(*s->h_place)));
            s->cap_place = cap;
        }
        for (uint32_t k = 0; k < s->n_edits; k++) {
            const uint32_t i = s->edits[k] & 0x3fffffffu;
            s->h_place[k] = (clapgpu_entity_place){ .slot = i, .parent = s->h_parent[i], .model = s->h_model[i],
                                                    .flags = ((s->edits[k] >> 31) ? CLAPGPU_PLACE_ZERO_BOX : 0) |
                                                             ((s->edits[k] & 0x40000000u) ? CLAPGPU_PLACE_CLEAR_STALE : 0) };
        }
        CK(clapgpu_entities_place(NULL, &s->d, (const clapgpu_entity_place *)s->d_place, s->n_edits, s->d_stale));
        s->n_edits = 0;
    }
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_new_placed(clapgpu_scene *s, uint32_t model, void *user, uint32_t parent, uint32_t *handle, uint32_t *slot_out)
{
    if (!s || !handle || model >= s->n_models || (parent != CLAPGPU_NO_ENTITY && !get(s, parent))) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!layout_editable(s)) return CLAPGPU_ERR_NOT_SUPPORTED;
    uint32_t slot = CLAPGPU_NO_ENTITY;
    if (parent == CLAPGPU_NO_ENTITY) {
        while (s->n_free_roots && slot == CLAPGPU_NO_ENTITY) {
            const uint32_t c = s->free_roots[--s->n_free_roots];
            if (c < s->n_slots && s->slot_handle[c] == CLAPGPU_NO_ENTITY) slot = c;
        }
        if (slot == CLAPGPU_NO_ENTITY && s->grow_tile != CLAPGPU_NO_ENTITY) {
            const uint32_t row = s->tile_row_start_host[s->grow_tile];
            const int l = free_lane(s, row);
            if (l >= 0) slot = row * WAVE + (uint32_t)l;
        }
        if (slot == CLAPGPU_NO_ENTITY) {
            CK(append_tile(s));
            slot = s->tile_row_start_host[s->grow_tile] * WAVE;
        }
    } else {
        const struct ent *pe = &s->e[parent];
        if (pe->slot >= s->n_slots || pe->attached) return CLAPGPU_ERR_NOT_SUPPORTED;
        const uint32_t row = pe->slot / WAVE + 1, t = tile_of_row(s, row - 1);
        if (row >= s->tile_row_start_host[t + 1]) return CLAPGPU_ERR_NOT_SUPPORTED;     /* the parent sits in its tile's last row */
        const int l = free_lane(s, row);
        if (l < 0) return CLAPGPU_ERR_NOT_SUPPORTED;
        slot = row * WAVE + (uint32_t)l;
    }
    if (s->n_edits == s->cap_edits) {                    /* before anything is changed: the list must be able to take the slot */
        const uint32_t c = s->cap_edits ? 2 * s->cap_edits : 64;
        uint32_t *q = realloc(s->edits, (size_t)c * 4);
        if (!q) return CLAPGPU_ERR_NOMEM;
        s->edits = q; s->cap_edits = c;
    }
    const int32_t parent_slot = parent == CLAPGPU_NO_ENTITY ? -1 : (int32_t)s->e[parent].slot;
    CK(new_handle(s, model, user, handle));              /* (may move s->e) */
    struct ent *e = &s->e[*handle];
    e->slot = slot;
    e->parent = parent;
    if (parent != CLAPGPU_NO_ENTITY) s->e[parent].n_children++;
    s->slot_handle[slot] = *handle;
    s->slot_user[slot] = user;
    s->h_parent[slot] = parent_slot;
    s->h_model[slot] = (int32_t)model;
    if (!s->n_edits || slot < s->edit_lo) s->edit_lo = slot;
    if (!s->n_edits || slot + 1 > s->edit_hi) s->edit_hi = slot + 1;
    uint32_t skip_bits;
    memcpy(&skip_bits, &s->models[8 * (size_t)model + 3], 4);
    const uint64_t bit = 1ull << (slot & 63);
    int was_stale = 0;
    if (s->h_stale[slot >> 6] & bit) {
        s->h_stale[slot >> 6] &= ~bit;
        if (!s->h_stale[slot >> 6] && s->n_stale_words) s->n_stale_words--;
        was_stale = 1;                                   /* the device's twin follows with the frame's place list */
    }
    s->edits[s->n_edits++] = slot | (skip_bits ? 0x80000000u : 0) | (was_stale ? 0x40000000u : 0);
    if (s->h_keep[slot >> 6] & bit) { s->h_keep[slot >> 6] &= ~bit; s->keep_dirty = 1; }
    s->h_fetched[slot >> 6] &= ~bit;
    if (s->lod_cap > slot && s->lod_layout_gen == s->layout_gen) {
        s->h_force_lod[slot] = -1; s->h_cur_lod[slot] = 0;
        if (slot < s->lod_lo) s->lod_lo = slot;
        if (slot + 1 > s->lod_hi) s->lod_hi = slot + 1;
    }
    mark_dirty(s, *handle, 1);                           /* its inputs into the image; the launch takes them by the touched bit */
    if (slot_out) *slot_out = slot;
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_delete_placed(clapgpu_scene *s, uint32_t handle)
{
    struct ent *e = get(s, handle);
    if (!e) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!layout_editable(s) || e->n_children || e->attached || e->slot >= s->n_slots) return CLAPGPU_ERR_NOT_SUPPORTED;
    if (s->n_limbo == s->cap_limbo) {
        const uint32_t c = s->cap_limbo ? 2 * s->cap_limbo : 64;
        uint32_t *q = realloc(s->limbo, (size_t)c * 4);
        if (!q) return CLAPGPU_ERR_NOMEM;
        s->limbo = q; s->cap_limbo = c;
    }
    const uint32_t slot = e->slot;
    s->h_flags[slot] = 0;                                /* not ALIVE: never rebuilt, drawn or picked again */
    touch_raw(s, slot);
    s->slot_handle[slot] = CLAPGPU_NO_ENTITY;
    s->slot_user[slot] = NULL;
    const uint64_t bit = 1ull << (slot & 63);
    if (s->h_stale[slot >> 6] & bit) {
        s->h_stale[slot >> 6] &= ~bit;
        if (!s->h_stale[slot >> 6] && s->n_stale_words) s->n_stale_words--;
        /* the device's twin follows with the frame's place list (parent / model as they are) */
        if (push_list(&s->edits, &s->n_edits, &s->cap_edits, slot | 0x40000000u)) s->topology_dirty = 1;   /* (a re-tile clears both) */
    }
    if (s->h_keep[slot >> 6] & bit) { s->h_keep[slot >> 6] &= ~bit; s->keep_dirty = 1; }
    s->h_fetched[slot >> 6] &= ~bit;
    if (e->parent != CLAPGPU_NO_ENTITY && e->parent < s->n_handles && s->e[e->parent].live && s->e[e->parent].n_children)
        s->e[e->parent].n_children--;
    const uint32_t row = slot / WAVE;
    if (s->tile_row_start_host[tile_of_row(s, row)] == row)
        push_list(&s->free_roots, &s->n_free_roots, &s->cap_free_roots, slot);   /* (a failure only loses the lane until the next re-tile) */
    e->live = 0;
    s->limbo[s->n_limbo++] = handle;
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_position(clapgpu_scene *s, uint32_t handle, const float pos[3])
{
    struct ent *e = get(s, handle);
    if (!e || !pos) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    memcpy(e->pos_scale, pos, 12);
    mark_dirty(s, handle, 1);
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_transform(clapgpu_scene *s, uint32_t handle, const float pos[3], const float q[4], float scale)
{
    struct ent *e = get(s, handle);
    if (!e || !pos || !q) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    memcpy(e->pos_scale, pos, 12);
    e->pos_scale[3] = scale;
    memcpy(e->rot, q, 16);
    mark_dirty(s, handle, 1);
    return CLAPGPU_OK;
}

/* entity_transform + entity_flags for callers that update many DIFFERENT handles from several threads at once (no
 * topology verb may run meanwhile): nothing shared is touched, so the caller has to finish with
 * clapgpu_scene_mark_all_dirty(), which makes the next mq_update upload the whole image. */
int clapgpu_scene_entity_transform_mt(clapgpu_scene *s, uint32_t handle, const float pos[3], const float q[4], float scale,
                                      uint32_t flags, int xform_updated)
{
    struct ent *e = get(s, handle);
    if (!e || !pos || !q) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    memcpy(e->pos_scale, pos, 12);
    e->pos_scale[3] = scale;
    memcpy(e->rot, q, 16);
    e->flags = flags & ~CLAPGPU_E_DIRTY;
    if (!s->topology_dirty && s->h_in && e->slot < s->n_slots) {
        memcpy(s->h_pos_scale + 4 * (size_t)e->slot, e->pos_scale, 16);
        memcpy(s->h_rot + 4 * (size_t)e->slot, e->rot, 16);
        s->h_flags[e->slot] = img_flags(e, xform_updated);
        if (s->zero_copy && s->tiled)                    /* one-launch frames read the flagged slots from the image: no bulk copy */
            __atomic_fetch_or(&s->h_touched[e->slot >> 6], 1ull << (e->slot & 63), __ATOMIC_RELAXED);
    } else {
        e->dirty |= xform_updated ? 3 : 1;               /* picked up by the re-tile's full image */
    }
    return CLAPGPU_OK;
}

/* The transform alone (entity3d_position / _move / _rotate / _scale leave the flags as they are), same threading rules;
 * xform_updated is OR-ed in, so pushing an entity twice in a frame -- the second time with its flag already taken -- is
 * harmless.  Finish with clapgpu_scene_mark_all_dirty(). */
int clapgpu_scene_entity_xform_mt(clapgpu_scene *s, uint32_t handle, const float pos[3], const float q[4], float scale, int xform_updated)
{
    struct ent *e = get(s, handle);
    if (!e || !pos || !q) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    memcpy(e->pos_scale, pos, 12);
    e->pos_scale[3] = scale;
    memcpy(e->rot, q, 16);
    if (!s->topology_dirty && s->h_in && e->slot < s->n_slots) {
        memcpy(s->h_pos_scale + 4 * (size_t)e->slot, e->pos_scale, 16);
        memcpy(s->h_rot + 4 * (size_t)e->slot, e->rot, 16);
        if (xform_updated) s->h_flags[e->slot] |= CLAPGPU_E_DIRTY;
        if (s->zero_copy && s->tiled)
            __atomic_fetch_or(&s->h_touched[e->slot >> 6], 1ull << (e->slot & 63), __ATOMIC_RELAXED);
    } else if (xform_updated) {
        e->dirty |= 3;                                   /* picked up by the re-tile's full image */
    }
    return CLAPGPU_OK;
}

/* what the call above will touch for (handle, slot), asked for ahead of time: the mirror's own record and the three rows of
 * the upload image -- four cache lines nothing else would bring in before the call stalls on each in turn */
void clapgpu_scene_entity_xform_prefetch(const clapgpu_scene *s, uint32_t handle, uint32_t slot)
{
    if (!s || handle >= s->n_handles) return;
    __builtin_prefetch(&s->e[handle], 1, 1);
    if (s->topology_dirty || !s->h_in || slot >= s->n_slots) return;
    __builtin_prefetch(s->h_pos_scale + 4 * (size_t)slot, 1, 1);
    __builtin_prefetch(s->h_rot + 4 * (size_t)slot, 1, 1);
    __builtin_prefetch(s->h_flags + slot, 1, 1);
}

void clapgpu_scene_mark_all_dirty(clapgpu_scene *s) { if (s) s->bulk_dirty = 1; }

int clapgpu_scene_entity_rotation(clapgpu_scene *s, uint32_t handle, const float q[4])
{
    struct ent *e = get(s, handle);
    if (!e || !q) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    memcpy(e->rot, q, 16);
    mark_dirty(s, handle, 1);
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_move(clapgpu_scene *s, uint32_t handle, const float off[3])
{
    struct ent *e = get(s, handle);
    if (!e || !off) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    for (int i = 0; i < 3; i++)                             /* transform_move: vec3_add(pos, pos, off) */
        e->pos_scale[i] = e->pos_scale[i] + off[i];
    mark_dirty(s, handle, 1);
    return CLAPGPU_OK;
}

/* transform_set_angles (transform.c:62-73): clamp_radians / clamp_degrees + to_radians (util.h:77-95),
 * then quat_from_euler_xyz (linmath.h:857-870) with the host's sinf / cosf, like the reference */
void clapgpu_quat_from_angles(const float angles[3], int degrees, float q[4])
{
    float r[3];
    for (int i = 0; i < 3; i++) {
        float a = angles[i];
        if (degrees) {
            a = fabsf(a) <= 180.0 ? a : (a - copysignf(360.0, a));
            a = a * M_PI / 180.0;
        } else {
            a = fabsf(a) <= M_PI ? a : (a - copysignf(M_PI * 2.0, a));
        }
        r[i] = a;
    }
    float cx = cosf(r[0] * 0.5f), sx = sinf(r[0] * 0.5f);
    float cy = cosf(r[1] * 0.5f), sy = sinf(r[1] * 0.5f);
    float cz = cosf(r[2] * 0.5f), sz = sinf(r[2] * 0.5f);
    q[0] = sx * cy * cz - cx * sy * sz;
    q[1] = cx * sy * cz + sx * cy * sz;
    q[2] = cx * cy * sz - sx * sy * cz;
    q[3] = cx * cy * cz + sx * sy * sz;
}

int clapgpu_scene_entity_rotate(clapgpu_scene *s, uint32_t handle, float rx, float ry, float rz)
{
    struct ent *e = get(s, handle);
    if (!e) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const float angles[3] = { rx, ry, rz };
    clapgpu_quat_from_angles(angles, 0, e->rot);            /* entity3d_rotate: radians (model.c:1818-1821) */
    mark_dirty(s, handle, 1);
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_visible(clapgpu_scene *s, uint32_t handle, unsigned int visible)
{
    return clapgpu_scene_entity_flags(s, handle, visible ? CLAPGPU_E_VISIBLE : 0, visible ? 0 : CLAPGPU_E_VISIBLE);
}

int clapgpu_scene_entity_scale(clapgpu_scene *s, uint32_t handle, float scale)
{
    struct ent *e = get(s, handle);
    if (!e) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    e->pos_scale[3] = scale;
    mark_dirty(s, handle, 1);
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_flags(clapgpu_scene *s, uint32_t handle, uint32_t set, uint32_t clear)
{
    struct ent *e = get(s, handle);
    if (!e) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    e->flags = ((e->flags | set) & ~clear) & ~CLAPGPU_E_DIRTY;
    mark_dirty(s, handle, 0);                           /* flags upload only: entity3d_visible() does not touch xform */
    return CLAPGPU_OK;
}

/* ---------------------------------------------------------------- layout */
static int ensure_slots(clapgpu_scene *s, uint32_t n_slots)
{
    if (n_slots <= s->cap_slots) return CLAPGPU_OK;
    /* an eighth of head room, in 4096-slot steps: the slabs cross PCIe whole, so capacity is traffic */
    uint32_t cap = (n_slots + n_slots / 8 + 4095u) & ~4095u;
    size_t n = cap;                                     /* a multiple of 64: every sub-array below starts 16-B aligned */
#define RE(p, bytes) do { void *q__ = realloc(p, bytes); if (!q__) return CLAPGPU_ERR_NOMEM; p = q__; } while (0)
    RE(s->slot_handle, n * 4); RE(s->slot_user, n * sizeof(void *));
#undef RE
    /* retile() rewrites the upload image in full and downloads are overwritten by the next frame, so
     * nothing has to survive the growth */
    free_device(s);
    if (s->h_in) clapgpu_host_free(s->h_in);
    if (s->h_out) clapgpu_host_free(s->h_out);
    s->h_in = s->h_out = NULL;
    /* page-locked: the small copies that carry an in-place edit's parent / model index are then queued, not staged and waited for */
    if (s->h_parent) clapgpu_host_free(s->h_parent);
    if (s->h_model) clapgpu_host_free(s->h_model);
    s->h_parent = s->h_model = NULL;
    CK(clapgpu_host_malloc((void **)&s->h_parent, n * 4));
    CK(clapgpu_host_malloc((void **)&s->h_model, n * 4));
    s->models_dirty = 1;                                /* free_device() dropped d.model_table */
    s->have_results = 0;
    s->in_bytes = n * 36 + (n / 64 + 2) * 8;                 /* + the touched-slot bits */
    s->out_bytes = n * 164 + 4 * (n / 64 + 2) * 8;           /* + visibility, rebuilt, bounding-volume and exported masks */
    s->zero_copy = cap <= s->zero_copy_max_slots;
    if (s->zero_copy) CK(clapgpu_host_malloc_mapped(&s->h_in, &s->d_in_host, s->in_bytes));
    else CK(clapgpu_host_malloc(&s->h_in, s->in_bytes));
    if (s->zero_copy) {
        CK(clapgpu_host_malloc_mapped(&s->h_out, &s->d_out_host, s->out_bytes));
        memset(s->h_out, 0, s->out_bytes);
        if (!s->h_done) {
            void *dd = NULL;
            CK(clapgpu_host_malloc_mapped((void **)&s->h_done, &dd, 64));
            s->d_done = dd;
            *s->h_done = 0;
            CK(clapgpu_malloc((void **)&s->d_counter, 4));
            CK(clapgpu_memset(s->d_counter, 0, 4, NULL));
        }
    } else {
        CK(clapgpu_host_malloc(&s->h_out, s->out_bytes));
    }
    CK(clapgpu_malloc(&s->d_in, s->in_bytes));
    CK(clapgpu_malloc(&s->d_out, s->out_bytes));
    char *hi = s->h_in, *ho = s->h_out, *di = s->d_in, *dq = s->d_out;
    s->h_pos_scale = (float *)hi;              s->d.pos_scale = (const float *)di;
    s->h_rot = (float *)(hi + n * 16);         s->d.rot = (const float *)(di + n * 16);
    s->h_flags = (uint32_t *)(hi + n * 32);    s->d.flags = (uint32_t *)(di + n * 32);
    s->h_touched = (uint64_t *)(hi + n * 36);
    memset(s->h_touched, 0, (n / 64 + 2) * 8);
    s->h_mx = (float *)ho;                     s->d.mx = (float *)dq;
    s->h_inv = (float *)(ho + n * 64);         s->d.inv_mx = (float *)(dq + n * 64);
    s->h_aabb = (float *)(ho + n * 128);       s->d.aabb = (float *)(dq + n * 128);
    s->h_center = (float *)(ho + n * 152);     s->d.center = (float *)(dq + n * 152);
    s->h_mask = (uint64_t *)(ho + n * 164);    s->d.vis_mask = (uint64_t *)(dq + n * 164);
    s->h_rebuilt = s->h_mask + (n / 64 + 2);   s->d.rebuilt_mask = s->d.vis_mask + (n / 64 + 2);
    s->h_inside = s->h_rebuilt + (n / 64 + 2); s->bvq.inside_mask = s->d.rebuilt_mask + (n / 64 + 2);
    s->h_exported = s->h_inside + (n / 64 + 2);
    {
        const size_t mw = n / 64 + 2;
        if (s->d_keep) clapgpu_free(s->d_keep);
        if (s->h_select) clapgpu_host_free(s->h_select);
        s->d_keep = NULL; s->h_select = NULL; s->d_select = NULL;
        free(s->h_keep); free(s->h_stale); free(s->h_fetched);
        s->h_keep = calloc(mw, 8); s->h_stale = calloc(mw, 8); s->h_fetched = calloc(mw, 8);
        if (!s->h_keep || !s->h_stale || !s->h_fetched) return CLAPGPU_ERR_NOMEM;
        CK(clapgpu_malloc((void **)&s->d_keep, mw * 8));
        if (s->d_stale) clapgpu_free(s->d_stale);
        s->d_stale = NULL;
        CK(clapgpu_malloc((void **)&s->d_stale, mw * 8));
        CK(clapgpu_memset(s->d_stale, 0, mw * 8, NULL));
        if (s->zero_copy) CK(clapgpu_host_malloc_mapped((void **)&s->h_select, &s->d_select, mw * 8));
        s->n_stale_words = 0; s->n_fetched = 0; s->keep_dirty = 1;
    }
    void **dp[] = { (void **)&s->d.parent, (void **)&s->d.model, (void **)&s->d.seqs, (void **)&s->d.vis_row_pop,
                    (void **)&s->d_tile_row_start };
    size_t sz[] = { n * 4, n * 4, n * 4, (n / 64 + 16) / 16 * 16, (n / 64 + 2) * 4 };
    for (unsigned i = 0; i < sizeof(dp) / sizeof(dp[0]); i++)
        CK(clapgpu_malloc(dp[i], sz[i]));
    s->cap_slots = cap;
    return CLAPGPU_OK;
}

/* depth of every live entity under its root; returns max depth + 1, or 0 on a parent cycle.  Each range walks up from its
 * handles to the first ancestor whose depth is known and assigns the chain; ranges that meet on a chain write the same values
 * (root before depth, depth with release: whoever reads a depth finds its root). */
#define DEPTH_UNK 0xffffffffu
struct depth_ctx { clapgpu_scene *s; uint32_t *depth, *root; uint32_t maxd; int cycle; };
static void depths_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct depth_ctx *dc = ctx;
    clapgpu_scene *s = dc->s;
    uint32_t *depth = dc->depth, *root = dc->root, maxd = 0;
    for (uint32_t h = lo; h < hi; h++) {
        if (!s->e[h].live || __atomic_load_n(&depth[h], __ATOMIC_ACQUIRE) != DEPTH_UNK) continue;
        uint32_t cur = h, len = 0;                       /* walk up to a known ancestor (or the root) */
        while (s->e[cur].parent != CLAPGPU_NO_ENTITY && __atomic_load_n(&depth[s->e[cur].parent], __ATOMIC_ACQUIRE) == DEPTH_UNK) {
            cur = s->e[cur].parent;
            if (++len > s->n_handles) { __atomic_store_n(&dc->cycle, 1, __ATOMIC_RELAXED); return; }
        }
        uint32_t base_d, base_r;
        if (s->e[cur].parent == CLAPGPU_NO_ENTITY) { base_d = 0; base_r = cur; }
        else { base_d = __atomic_load_n(&depth[s->e[cur].parent], __ATOMIC_ACQUIRE) + 1; base_r = __atomic_load_n(&root[s->e[cur].parent], __ATOMIC_RELAXED); }
        uint32_t x = h;
        for (uint32_t k = 0; k <= len; k++) {
            const uint32_t d = base_d + (len - k);
            __atomic_store_n(&root[x], base_r, __ATOMIC_RELAXED);
            __atomic_store_n(&depth[x], d, __ATOMIC_RELEASE);
            if (d + 1 > maxd) maxd = d + 1;
            x = s->e[x].parent;
        }
    }
    uint32_t seen = __atomic_load_n(&dc->maxd, __ATOMIC_RELAXED);
    while (maxd > seen && !__atomic_compare_exchange_n(&dc->maxd, &seen, maxd, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
}

static uint32_t compute_depths(clapgpu_scene *s, uint32_t *depth, uint32_t *root)
{
    memset(depth, 0xff, (size_t)s->n_handles * 4);
    struct depth_ctx dc = { s, depth, root, 0, 0 };
    run_ranges(s, depths_range, &dc, s->n_handles);
    if (dc.cycle) return 0;
    return dc.maxd ? dc.maxd : 1;
}

/* the re-tile's other passes over every handle / slot */
struct retile_ctx {
    clapgpu_scene *s; const uint32_t *depth, *root, *tree_of; uint32_t *width; uint32_t maxd; int wide;
};
static void widths_range(void *ctx, uint32_t lo, uint32_t hi)
{
    struct retile_ctx *rc = ctx;
    const clapgpu_scene *s = rc->s;
    for (uint32_t h = lo; h < hi; h++)
        if (s->e[h].live) {
            const uint32_t t = rc->tree_of[rc->root[h]];
            if (__atomic_add_fetch(&rc->width[(size_t)t * rc->maxd + rc->depth[h]], 1, __ATOMIC_RELAXED) > WAVE)
                __atomic_store_n(&rc->wide, 1, __ATOMIC_RELAXED);
        }
}

static void image_range(void *ctx, uint32_t lo, uint32_t hi)      /* in units of 64 slots: a range owns its words of h_keep */
{
    struct retile_ctx *rc = ctx;
    clapgpu_scene *s = rc->s;
    for (uint32_t i = lo * WAVE; i < hi * WAVE; i++) {
        const uint32_t h = s->slot_handle[i];
        s->slot_user[i] = h == CLAPGPU_NO_ENTITY ? NULL : s->e[h].user;
        if (h != CLAPGPU_NO_ENTITY && s->e[h].keep) s->h_keep[i >> 6] |= 1ull << (i & 63);
        if (h == CLAPGPU_NO_ENTITY) {
            const float id[4] = { 0, 0, 0, 1 };
            memcpy(s->h_pos_scale + 4 * (size_t)i, id, 16);
            memcpy(s->h_rot + 4 * (size_t)i, id, 16);
            s->h_parent[i] = -1; s->h_model[i] = 0; s->h_flags[i] = 0;
            continue;
        }
        const struct ent *e = &s->e[h];
        memcpy(s->h_pos_scale + 4 * (size_t)i, e->pos_scale, 16);
        memcpy(s->h_rot + 4 * (size_t)i, e->rot, 16);
        s->h_parent[i] = e->parent == CLAPGPU_NO_ENTITY ? -1 : (int32_t)s->e[e->parent].slot;
        s->h_model[i] = (int32_t)e->model;
        s->h_flags[i] = img_flags(e, 1);                             /* everything is rebuilt after a re-tile */
    }
}

/* Slots in HANDLE order inside each row, as one thread would give them, from passes that have no order in them: the handles
 * are cut into chunks; every chunk counts its entities per row (or per level), a pass over the rows turns the counts into
 * each chunk's first lane, and every chunk then hands out its lanes in handle order. */
static uint32_t rt_chunk(void)                                    /* handles per chunk (CLAPGPU_SCENE_RT_CHUNK: the tests set a few hundred) */
{
    static uint32_t v;
    if (!v) { const char *e = getenv("CLAPGPU_SCENE_RT_CHUNK"); v = e && atoi(e) > 0 ? (uint32_t)atoi(e) : 16384u; }
    return v;
}
struct slots_ctx {
    clapgpu_scene *s; const uint32_t *depth, *root, *tree_of, *row_of_tree; uint32_t *cnt; uint32_t n_cells, n_chunks, H, chunk; int tiled;
};
static inline uint32_t slots_cell(const struct slots_ctx *sc, uint32_t h)
{
    return sc->tiled ? sc->row_of_tree[sc->tree_of[sc->root[h]]] + sc->depth[h] : sc->s->level_start_host[sc->depth[h]] / WAVE;
}

static void slots_count_range(void *ctx, uint32_t lo, uint32_t hi)      /* in chunks */
{
    struct slots_ctx *sc = ctx;
    for (uint32_t c = lo; c < hi; c++) {
        uint32_t *cnt = sc->cnt + (size_t)c * sc->n_cells;
        const uint32_t h1 = (c + 1) * sc->chunk < sc->H ? (c + 1) * sc->chunk : sc->H;
        for (uint32_t h = c * sc->chunk; h < h1; h++)
            if (sc->s->e[h].live) cnt[slots_cell(sc, h)]++;
    }
}

static void slots_first_range(void *ctx, uint32_t lo, uint32_t hi)      /* in cells */
{
    struct slots_ctx *sc = ctx;
    for (uint32_t cell = lo; cell < hi; cell++) {
        uint32_t run = 0;
        for (uint32_t c = 0; c < sc->n_chunks; c++) {
            uint32_t *p = sc->cnt + (size_t)c * sc->n_cells + cell;
            const uint32_t t = *p;
            *p = run; run += t;
        }
    }
}

static void slots_assign_range(void *ctx, uint32_t lo, uint32_t hi)     /* in chunks */
{
    struct slots_ctx *sc = ctx;
    clapgpu_scene *s = sc->s;
    for (uint32_t c = lo; c < hi; c++) {
        uint32_t *cnt = sc->cnt + (size_t)c * sc->n_cells;
        const uint32_t h1 = (c + 1) * sc->chunk < sc->H ? (c + 1) * sc->chunk : sc->H;
        for (uint32_t h = c * sc->chunk; h < h1; h++) {
            if (!s->e[h].live) continue;
            const uint32_t cell = slots_cell(sc, h);
            const uint32_t k = cnt[cell]++;
            s->e[h].slot = (sc->tiled ? cell * WAVE : s->level_start_host[sc->depth[h]]) + k;
            s->slot_handle[s->e[h].slot] = h;
        }
    }
}

static void undirty_range(void *ctx, uint32_t lo, uint32_t hi)
{
    clapgpu_scene *s = ctx;
    for (uint32_t h = lo; h < hi; h++) s->e[h].dirty = 0;
}

static int retile(clapgpu_scene *s)
{
    const int timing = getenv("CLAPGPU_SCENE_TIMING") != NULL;
    double tp[8] = { 0 };
    tp[0] = timing ? scene_now_us() : 0;
    CK(release_dead(s));
    if (s->h_in)                                         /* tombstones of in-place deletions: the whole image follows anyway */
        for (uint32_t k = 0; k < s->n_raw; k++) s->h_touched[s->raw_words[k]] = 0;
    const uint32_t H = s->n_handles;
    uint32_t *depth = malloc(((size_t)H + 1) * 4), *root = malloc(((size_t)H + 1) * 4);
    uint32_t *tree_of = malloc(((size_t)H + 1) * 4);
    if (!depth || !root || !tree_of) return CLAPGPU_ERR_NOMEM;
    uint32_t maxd = compute_depths(s, depth, root);
    if (!maxd) { free(depth); free(root); free(tree_of); return CLAPGPU_ERR_INVALID_ARGUMENTS; }
    if (timing) tp[1] = scene_now_us();

    uint32_t n_trees = 0, n_live = 0;
    for (uint32_t h = 0; h < H; h++)
        if (s->e[h].live) { n_live++; if (s->e[h].parent == CLAPGPU_NO_ENTITY) tree_of[h] = n_trees++; }
    uint32_t *width = calloc((size_t)(n_trees ? n_trees : 1) * maxd, 4);
    if (!width) return CLAPGPU_ERR_NOMEM;
    struct retile_ctx rtc = { s, depth, root, tree_of, width, maxd, 0 };
    run_ranges(s, widths_range, &rtc, H);
    int tiled = !rtc.wide;

    uint32_t n_rows = 0;
    uint32_t *row_of_tree = malloc(((size_t)n_trees + 1) * 4);       /* first row of the tree's tile */
    uint32_t *row_fill = NULL;
    free(s->tile_row_start_host);
    free(s->level_start_host);
    s->tile_row_start_host = malloc(((size_t)n_trees + 2) * 4);      /* at most one tile per tree */
    s->cap_tiles = n_trees + 1;
    s->max_depth = maxd; s->grow_tile = CLAPGPU_NO_ENTITY; s->n_free_roots = 0;
    s->n_raw = 0; s->n_edits = 0; s->grown_from = s->tiles_from = CLAPGPU_NO_ENTITY;
    /* a mirror that is edited in place (clapgpu_scene_set_incremental) leaves every row an eighth of its lanes and every tile
     * of a hierarchy one row: room for the children that come before the next re-tile */
    const uint32_t row_limit = s->incremental ? WAVE - WAVE / 8 : WAVE;
    const uint32_t spare_rows = (s->incremental && maxd > 1) ? 1 : 0;
    s->level_start_host = malloc(((size_t)maxd + 2) * 4);
    if (!row_of_tree || !s->tile_row_start_host || !s->level_start_host) return CLAPGPU_ERR_NOMEM;
    if (tiled) {
        /* next-fit packing of whole trees: every level of a tile holds <= 64 entities */
        uint32_t *fill = calloc(maxd, 4);
        uint32_t tile_first_row = 0, tile_rows = 0;
        s->n_tiles = 0;
        for (uint32_t t = 0; t < n_trees; t++) {
            const uint32_t *w = width + (size_t)t * maxd;
            int fits = 1;
            uint32_t rows = 0;
            for (uint32_t d = 0; d < maxd; d++) { if (fill[d] && fill[d] + w[d] > row_limit) fits = 0; if (w[d]) rows = d + 1; }
            if (!fits) {                                            /* close the tile */
                s->tile_row_start_host[s->n_tiles++] = tile_first_row;
                tile_first_row += tile_rows + spare_rows;
                tile_rows = 0;
                memset(fill, 0, maxd * 4);
            }
            for (uint32_t d = 0; d < maxd; d++) fill[d] += w[d];
            if (rows > tile_rows) tile_rows = rows;
            row_of_tree[t] = tile_first_row;
        }
        if (n_trees) { s->tile_row_start_host[s->n_tiles++] = tile_first_row; tile_first_row += tile_rows + spare_rows; }
        s->tile_row_start_host[s->n_tiles] = tile_first_row;
        n_rows = tile_first_row;
        free(fill);
    } else {
        /* level-major: level d = rows [level_row[d], level_row[d+1]) */
        uint32_t *cnt = calloc(maxd, 4);
        for (uint32_t h = 0; h < H; h++) if (s->e[h].live) cnt[depth[h]]++;
        s->n_levels = maxd;
        uint32_t r = 0;
        for (uint32_t d = 0; d < maxd; d++) { s->level_start_host[d] = r * WAVE; r += (cnt[d] + WAVE - 1) / WAVE; }
        s->level_start_host[maxd] = r * WAVE;
        n_rows = r;
        free(cnt);
    }
    if (n_rows == 0) n_rows = 1;
    if (timing) tp[2] = scene_now_us();
    int rc = ensure_slots(s, n_rows * WAVE);
    if (rc) return rc;
    s->n_rows = n_rows;
    s->n_slots = n_rows * WAVE;
    s->tiled = tiled;
    if (!tiled) s->level_start_host[s->n_levels] = s->n_slots;      /* the kernel wants the last start == n */

    /* slots: handle order inside each row */
    if (timing) tp[3] = scene_now_us();
    memset(s->slot_handle, 0xff, (size_t)s->n_slots * 4);          /* CLAPGPU_NO_ENTITY */
    const uint32_t chunk = rt_chunk(), n_chunks = (H + chunk - 1) / chunk;
    uint32_t *chunk_cnt = (s->par_for && n_chunks > 1 && (uint64_t)n_chunks * n_rows <= (64u << 20)) ? calloc((size_t)n_chunks * n_rows, 4) : NULL;
    if (chunk_cnt) {
        struct slots_ctx sc = { s, depth, root, tree_of, row_of_tree, chunk_cnt, n_rows, n_chunks, H, chunk, tiled };
        s->par_for(slots_count_range, &sc, n_chunks, s->par_threads);
        run_ranges(s, slots_first_range, &sc, n_rows);
        s->par_for(slots_assign_range, &sc, n_chunks, s->par_threads);
        free(chunk_cnt);
    } else {
    row_fill = calloc(n_rows, 4);
    for (uint32_t h = 0; h < H; h++) {
        if (!s->e[h].live) continue;
        uint32_t row;
        if (tiled) {
            row = row_of_tree[tree_of[root[h]]] + depth[h];
            s->e[h].slot = row * WAVE + row_fill[row]++;
        } else {
            uint32_t base = s->level_start_host[depth[h]] / WAVE;
            uint32_t k = row_fill[base]++;                           /* counter kept in the level's first row */
            s->e[h].slot = s->level_start_host[depth[h]] + k;
        }
        s->slot_handle[s->e[h].slot] = h;
    }
    }
    /* the slots moved: what was stale under the old layout is rebuilt (and exported or marked stale again) by the launch
     * that follows; the standing readers' bits are laid out anew */
    if (timing) tp[4] = scene_now_us();
    memset(s->h_stale, 0, ((size_t)s->cap_slots / 64 + 2) * 8);
    CK(clapgpu_memset(s->d_stale, 0, ((size_t)s->cap_slots / 64 + 2) * 8, NULL));
    memset(s->h_fetched, 0, ((size_t)s->cap_slots / 64 + 2) * 8);
    memset(s->h_keep, 0, ((size_t)s->cap_slots / 64 + 2) * 8);
    s->n_stale_words = 0; s->n_fetched = 0; s->keep_dirty = 1;
    /* full staging image */
    run_ranges(s, image_range, &rtc, s->n_slots / WAVE);
    free(depth); free(root); free(tree_of); free(width); free(row_of_tree); free(row_fill);
    if (timing) tp[5] = scene_now_us();

    s->d.n = s->n_slots;
    const size_t n = s->n_slots;
    CK(clapgpu_memcpy_h2d((void *)s->d.parent, s->h_parent, n * 4, NULL));
    CK(clapgpu_memcpy_h2d((void *)s->d.model, s->h_model, n * 4, NULL));
    CK(clapgpu_memset(s->d.seqs, 0, n * 4, NULL));
    CK(clapgpu_memset(s->d_out, 0, s->out_bytes, NULL));
    /* the host's result slab is NOT cleared here (at a million entities that alone was 15 ms of a re-tile): the launch that
     * follows rebuilds and exports every live row and writes every mask word of the layout; padding rows are never read
     * (no slot_user), and a box-less model's rows are never copied out.  It is zeroed once, where it is allocated. */
    if (tiled)
        CK(clapgpu_memcpy_h2d(s->d_tile_row_start, s->tile_row_start_host, ((size_t)s->n_tiles + 1) * 4, NULL));
    /* every live handle, not only the listed ones: an entity marked dirty while the list could not grow (mark_dirty's
     * out-of-memory path) would otherwise stay "queued" for ever and never be listed again */
    run_ranges(s, undirty_range, s, H);
    s->n_dirty = 0;
    s->topology_dirty = 0;
    s->up_lo = 0xffffffffu; s->up_hi = 0;
    s->layout_gen++;
    if (timing) {
        const double t_end = scene_now_us();
        uint64_t fnv = 1469598103934665603ull;                   /* the layout, for comparing runs (serial / on a pool) */
        for (uint32_t i = 0; i < s->n_slots; i++) fnv = (fnv ^ s->slot_handle[i]) * 1099511628211ull;
        fprintf(stderr, "retile: %u handles -> %u slots: depths %.0f us, trees + packing %.0f, slabs %.0f, slots %.0f, image %.0f, uploads %.0f; layout %016llx\n",
                H, s->n_slots, tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], tp[4] - tp[3], tp[5] - tp[4], t_end - tp[5], (unsigned long long)fnv);
    }
    return CLAPGPU_OK;
}

int clapgpu_scene_mq_update(clapgpu_scene *s, const clapgpu_frustum *frustum)
{
    if (!s) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    int upload = 0, full = 0;
    uint32_t lo = 0xffffffffu, hi = 0, n_touched = 0, n_bits = 0;
    if (s->topology_dirty) {
        CK(retile(s));
        upload = full = 1;
    } else
        CK(apply_edits(s));
    if (!full && s->n_dirty) {
        /* the upload image was written as the verbs came in (mark_dirty); here only the bookkeeping */
        const int bits = s->zero_copy && s->tiled;       /* one launch: the kernel reads the flagged slots from the image */
        if (s->zero_copy && !bits && s->n_dirty > s->cap_list) {  /* the mapped record list grows with the busiest frame seen */
            uint32_t cap = s->cap_list ? s->cap_list : 1024;
            while (cap < s->n_dirty) cap *= 2;
            if (s->h_list) clapgpu_host_free(s->h_list);
            s->h_list = NULL; s->cap_list = 0;
            CK(clapgpu_host_malloc_mapped((void **)&s->h_list, &s->d_list, (size_t)cap * sizeof(*s->h_list)));
            s->cap_list = cap;
        }
        for (uint32_t k = 0; k < s->n_dirty; k++) {
            struct ent *e = &s->e[s->dirty_list[k]];
            const uint8_t was = e->dirty;
            e->dirty = 0;
            if (!e->live) continue;
            if (bits) {
                s->h_touched[e->slot >> 6] |= 1ull << (e->slot & 63);
            } else if (s->zero_copy) {
                clapgpu_entity_input *r = &s->h_list[n_touched];
                r->slot = e->slot;
                r->flags = img_flags(e, was & 2);
                memcpy(r->pos_scale, e->pos_scale, 16);
                memcpy(r->rot, e->rot, 16);
            }
            s->dirty_list[n_touched++] = e->slot;        /* the list is reused for the slots touched */
        }
        lo = s->up_lo; hi = s->up_hi;
        s->n_dirty = 0;
        upload = n_touched != 0 && hi > lo;
        if (bits) n_bits = n_touched;
    }
    if (s->n_limbo) {                                    /* deleted in place: nothing lists these handles any more */
        if (s->n_free + s->n_limbo > s->cap_free) {
            uint32_t cap = s->cap_free ? s->cap_free : 256;
            while (cap < s->n_free + s->n_limbo) cap *= 2;
            uint32_t *q = realloc(s->free_list, (size_t)cap * sizeof(uint32_t));
            if (!q) return CLAPGPU_ERR_NOMEM;
            s->free_list = q; s->cap_free = cap;
        }
        memcpy(s->free_list + s->n_free, s->limbo, (size_t)s->n_limbo * sizeof(uint32_t));
        s->n_free += s->n_limbo;
        s->n_limbo = 0;
    }
    const uint32_t n_raw = full ? 0 : s->n_raw;          /* tombstones: flags words flagged outside the dirty list */
    if (n_raw) {
        if (!upload || s->raw_lo < lo) lo = s->raw_lo;
        if (!upload || s->raw_hi > hi) hi = s->raw_hi;
        upload = 1;
    }
    const int bulk_any = s->bulk_dirty;
    const int bulk = s->bulk_dirty && !full;
    if (bulk) {                                          /* clapgpu_scene_entity_transform_mt wrote the image directly */
        upload = 1; lo = 0; hi = s->n_slots; n_touched = s->n_slots;   /* whole image up, flags cleared linearly */
    }
    s->bulk_dirty = 0;
    s->up_lo = 0xffffffffu; s->up_hi = 0;
    if (s->models_dirty) {
        if (s->n_models > s->d_models_cap) {
            if (s->d_models) clapgpu_free(s->d_models);
            s->d_models_cap = s->n_models * 2;
            CK(clapgpu_malloc((void **)&s->d_models, (size_t)s->d_models_cap * 32));
        }
        CK(clapgpu_memcpy_h2d(s->d_models, s->models, (size_t)s->n_models * 32, NULL));
        s->d.model_table = s->d_models;
        s->d.n_models = s->n_models;
        s->models_dirty = 0;
        full = 1;
    }
    if (s->n_models == 0) {
        for (uint32_t k = 0; k < n_bits; k++) s->h_touched[s->dirty_list[k] >> 6] = 0;
        return CLAPGPU_OK;
    }
    const size_t n = s->n_slots;
    const size_t cap = s->cap_slots;
    const int fused = s->zero_copy && s->tiled;          /* update + export (+ the touched inputs) as one launch */
    const int by_bits = fused && upload && !full;        /* bulk: clapgpu_scene_entity_transform_mt flagged its slots itself */
    const int by_list = s->zero_copy && !fused && upload && !full && !bulk && n_touched <= s->cap_list;   /* same bytes as the image, no copy call */
    if (by_bits) {
        /* nothing to issue: h_touched says which slots of the mapped image the kernel has to take */
    } else if (by_list) {
        CK(clapgpu_entities_apply_inputs(NULL, &s->d, (const clapgpu_entity_input *)s->d_list, n_touched));
    } else if (upload) {
        /* one copy of the whole input slab after a re-tile or when most of it changed; else the slot range */
        const size_t a = full ? 0 : lo, cnt = full ? n : (size_t)hi - lo;
        if (full || 2 * cnt > n) {
            CK(clapgpu_memcpy_h2d(s->d_in, s->h_in, cap * 32 + n * 4, NULL));
        } else {
            CK(clapgpu_memcpy_h2d((float *)s->d.pos_scale + 4 * a, s->h_pos_scale + 4 * a, cnt * 16, NULL));
            CK(clapgpu_memcpy_h2d((float *)s->d.rot + 4 * a, s->h_rot + 4 * a, cnt * 16, NULL));
            CK(clapgpu_memcpy_h2d(s->d.flags + a, s->h_flags + a, cnt * 4, NULL));
        }
    }
    if (s->bv_on) {
        memcpy(s->bvq.cam_pos, s->bv_cam, 12); memcpy(s->bvq.ctl_pos, s->bv_ctl, 12);
        const struct ent *ce = s->bv_has_ctl ? get(s, s->bv_ctl_handle) : NULL;
        s->bvq.has_ctl = s->bv_has_ctl; s->bvq.ctl_entity = ce ? ce->slot : 0xffffffffu;
        s->bvq.result = NULL;                            /* the containment mask is what the callers replay: no result word, no fill launch */
        s->d.bv = &s->bvq;
    } else {
        s->d.bv = NULL;
    }
    s->have_frustum = frustum != NULL;
    if (frustum) s->last_frustum = *frustum;
    if (frustum) CK(ensure_views(s)); else s->xv.n = 0;  /* the frame's other views ride the main one's launch */
    s->d.views = s->xv.n ? &s->xv : NULL;
    s->d.n_attach = 0;                                   /* joint attachments ride the palettes of THIS frame: clapgpu_scene_attached_update */
    const double tt0 = scene_now_us();
    const size_t mask_words = n / 64, mask_stride = cap / 64 + 2;
    if (fused) {
        clapgpu_entities_hostio io;
        if (s->keep_dirty && s->export_drawn) {
            CK(clapgpu_memcpy_h2d(s->d_keep, s->h_keep, (n / 64) * 8, NULL));
            CK(clapgpu_stream_sync(NULL));                 /* h_keep is pageable and may change right after this call */
            s->keep_dirty = 0;
        }
        scene_hostio(s, &io, by_bits, s->export_drawn);
        io.options |= CLAPGPU_HOSTIO_EXPORT_STALE_READ;    /* what an earlier frame left stale and this one reads comes over in the same launch */
        for (uint32_t v = 0; v < s->xv.n; v++) s->xv.host_vis_mask[v] = s->a_xv_mask[v];   /* the launch writes the views' masks home itself */
        CK(clapgpu_entities_update_tiles_hostio(NULL, &s->d, s->d_tile_row_start, s->n_tiles, 0, frustum, &io));
        const double tt2 = scene_now_us();
        CK(clapgpu_wait_word(s->h_done, s->frame_id, NULL));
        if (s->export_drawn || s->n_stale_words) stale_after_launch(s);
        else if (s->n_fetched) { memset(s->h_fetched, 0, (n / 64) * 8); s->n_fetched = 0; }
        if (getenv("CLAPGPU_SCENE_TIMING"))
            fprintf(stderr, "scene small frame: %u inputs by %s, one launch %.1f us, wait %.1f us\n", n_touched,
                    by_bits ? "touched bits" : upload ? "copy" : "none", tt2 - tt0, scene_now_us() - tt2);
    } else if (s->tiled)
        CK(clapgpu_entities_update_tiles(NULL, &s->d, s->d_tile_row_start, s->n_tiles, 0, frustum));
    else
        CK(clapgpu_entities_update(NULL, &s->d, s->level_start_host, s->n_levels, 0, frustum));
    if (fused) {
        /* results and masks are in h_out already */
    } else if (s->zero_copy) {
        /* what the update rebuilt, and the masks, straight into the mapped result slab; then the completion word */
        char *mo = s->d_out_host;
        const size_t cn = cap;
        clapgpu_entities_export x = { .mx = (float *)mo, .inv_mx = (float *)(mo + cn * 64), .aabb = (float *)(mo + cn * 128),
                                      .center = (float *)(mo + cn * 152), .vis_mask = (uint64_t *)(mo + cn * 164) };
        x.rebuilt_mask = x.vis_mask + (cn / 64 + 2);
        x.inside_mask = s->bv_on ? x.rebuilt_mask + (cn / 64 + 2) : NULL;
        x.counter = s->d_counter; x.done = s->d_done; x.done_value = ++s->frame_id;
        CK(clapgpu_entities_export_rebuilt(NULL, &s->d, &x));
        const double tt2 = scene_now_us();
        CK(clapgpu_wait_word(s->h_done, s->frame_id, NULL));
        if (s->xv.n) { CK(download_views(s)); CK(clapgpu_stream_sync(NULL)); }
        if (getenv("CLAPGPU_SCENE_TIMING"))
            fprintf(stderr, "scene small frame: %u inputs by %s, launches %.1f us, wait %.1f us\n", n_touched,
                    by_list ? "list" : upload ? "copy" : "none", tt2 - tt0, scene_now_us() - tt2);
    } else
    if (upload || full || !s->have_results) {            /* otherwise the kernel rebuilt nothing: the last download stands */
        if (1) {                                         /* one copy of the output slab (the three masks included): cap <= 9/8 n + 4096 */
            CK(clapgpu_memcpy_d2h(s->h_out, s->d_out, cap * 164 + (2 * mask_stride + mask_words) * 8, NULL));
        } else {
            CK(clapgpu_memcpy_d2h(s->h_mx, s->d.mx, n * 64, NULL));
            CK(clapgpu_memcpy_d2h(s->h_inv, s->d.inv_mx, n * 64, NULL));
            CK(clapgpu_memcpy_d2h(s->h_aabb, s->d.aabb, n * 24, NULL));
            CK(clapgpu_memcpy_d2h(s->h_center, s->d.center, n * 12, NULL));
            CK(clapgpu_memcpy_d2h(s->h_mask, s->d.vis_mask, (2 * mask_stride + mask_words) * 8, NULL));
        }
    } else {                                             /* masks only: visibility of this view, nothing rebuilt */
        CK(clapgpu_memcpy_d2h(s->h_mask, s->d.vis_mask, (2 * mask_stride + mask_words) * 8, NULL));
    }
    if (!s->zero_copy) { CK(download_views(s)); CK(clapgpu_stream_sync(NULL)); }
    for (uint32_t k = 0; k < n_bits; k++) s->h_touched[s->dirty_list[k] >> 6] = 0;   /* taken by this frame's launch or by its copy */
    for (uint32_t k = 0; k < s->n_raw; k++) s->h_touched[s->raw_words[k]] = 0;
    s->n_raw = 0;
    if (bulk_any && s->zero_copy) memset(s->h_touched, 0, (cap / 64 + 2) * 8);   /* what clapgpu_scene_entity_transform_mt flagged */
    if (!frustum)
        memset(s->h_mask, 0, mask_words * 8);                /* (and no extra view has a mask: xv.n is 0 for this frame) */
    if (!s->bv_on) memset(s->h_inside, 0, mask_words * 8);
    /* EXPORT_DRAWN: whoever is read this frame and was left stale by an earlier one -- an entity that came into view, a
     * box that now contains the camera, a reader registered since -- comes over now */
    if (fused) {
        /* (the launch has brought over what it found stale and read -- stale_after_launch; this catches what it could not
         * know: nothing, unless a caller's masks changed behind it) */
        s->fetch_accumulate = 1;
        const int frc = frustum ? fetch_rows(s, views_union(s), s->bv_on ? s->h_inside : NULL, s->h_keep)
                                : fetch_rows(s, NULL, NULL, NULL);      /* a pass without a camera draws everything (model.c:969) */
        s->fetch_accumulate = 0;
        CK(frc);
    }
    if (full || 4 * (size_t)n_touched > n)
        for (size_t i = 0; i < n; i++) s->h_flags[i] &= ~CLAPGPU_E_DIRTY;   /* the kernel cleared its copy too */
    else
        for (uint32_t k = 0; k < n_touched; k++) s->h_flags[s->dirty_list[k]] &= ~CLAPGPU_E_DIRTY;
    s->have_results = 1;
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_set_attach(clapgpu_scene *s, uint32_t handle, int attached)
{
    struct ent *e = get(s, handle);
    if (!e) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (e->attached != (attached ? 1 : 0)) {
        e->attached = attached ? 1 : 0;
        s->topology_dirty = 1;                           /* the flag travels with the upload image */
    }
    return CLAPGPU_OK;
}

struct att_key { uint32_t slot, k; };
static int att_cmp(const void *a, const void *b)
{
    const struct att_key *x = a, *y = b;
    return x->slot < y->slot ? -1 : x->slot > y->slot;
}

/*
 * The second launch of a frame with joint attachments (model.c:1626-1641): entity handles[k] rides
 * parent.mx * ((jt[k] * bind[k]) * local), jt[k] = its parent's joint_transforms[parent_joint] of THIS frame -- which
 * exist only after the pose that followed clapgpu_scene_mq_update() -- and bind[k] that joint's bind matrix.  Such
 * entities are rebuilt every frame, everything below them follows through the seq counters; nothing else is touched.
 * On return the result arrays hold the rebuilt rows and rebuilt_mask says which they are.
 */
int clapgpu_scene_attached_update(clapgpu_scene *s, uint32_t n, const uint32_t *handles, const float *jt, const float *bind)
{
    if (!s || (n && (!handles || !jt || !bind))) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!n) return CLAPGPU_OK;
    if (s->topology_dirty || !s->have_results || !s->n_models) return CLAPGPU_ERR_NOT_SUPPORTED;   /* mq_update first */
    CK(apply_edits(s));
    const size_t need = (size_t)n * (sizeof(clapgpu_attach) + 128);
    if (n > s->cap_att || s->att_mapped != s->zero_copy) {
        uint32_t cap = s->cap_att ? s->cap_att : 64;
        while (cap < n) cap *= 2;
        if (s->h_att) clapgpu_host_free(s->h_att);
        if (s->d_att && !s->att_mapped) clapgpu_free(s->d_att);
        if (s->d_att_local) clapgpu_free(s->d_att_local);
        s->h_att = s->d_att = NULL; s->d_att_local = NULL; s->cap_att = 0;
        const size_t bytes = (size_t)cap * (sizeof(clapgpu_attach) + 128);
        s->att_mapped = s->zero_copy;
        if (s->att_mapped) CK(clapgpu_host_malloc_mapped(&s->h_att, &s->d_att, bytes));
        else { CK(clapgpu_host_malloc(&s->h_att, bytes)); CK(clapgpu_malloc(&s->d_att, bytes)); }
        CK(clapgpu_malloc((void **)&s->d_att_local, (size_t)cap * 64));
        s->cap_att = cap;
    }
    struct att_key *key = malloc((size_t)n * sizeof(*key));
    if (!key) return CLAPGPU_ERR_NOMEM;
    for (uint32_t k = 0; k < n; k++) {
        const struct ent *e = get(s, handles[k]);
        if (!e || !e->attached || e->parent == CLAPGPU_NO_ENTITY || e->slot >= s->n_slots) { free(key); return CLAPGPU_ERR_INVALID_ARGUMENTS; }
        key[k].slot = e->slot; key[k].k = k;
    }
    qsort(key, n, sizeof(*key), att_cmp);                /* the kernel looks an entity up by binary search */
    clapgpu_attach *tab = s->h_att;
    float *pj = (float *)((char *)s->h_att + (size_t)n * sizeof(clapgpu_attach)), *pb = pj + 16 * (size_t)n;
    for (uint32_t i = 0; i < n; i++) {
        if (i && key[i].slot == key[i - 1].slot) { free(key); return CLAPGPU_ERR_INVALID_ARGUMENTS; }
        tab[i] = (clapgpu_attach){ .entity = key[i].slot, .jt = i, .bind = i };
        memcpy(pj + 16 * (size_t)i, jt + 16 * (size_t)key[i].k, 64);
        memcpy(pb + 16 * (size_t)i, bind + 16 * (size_t)key[i].k, 64);
    }
    free(key);
    if (!s->att_mapped) CK(clapgpu_memcpy_h2d(s->d_att, s->h_att, need, NULL));
    s->d.n_attach = n;
    s->d.attach = s->d_att;
    s->d.jt_pool = (const float *)((const char *)s->d_att + (size_t)n * sizeof(clapgpu_attach));
    s->d.bind_pool = s->d.jt_pool + 16 * (size_t)n;
    s->d.attach_local = s->d_att_local;
    const clapgpu_frustum *fr = s->have_frustum ? &s->last_frustum : NULL;
    const int fused = s->zero_copy && s->tiled;
    int rc;
    if (fused) {
        clapgpu_entities_hostio io;
        scene_hostio(s, &io, 0, 0);                   /* the few attached subtrees: every rebuilt row comes back */
        rc = clapgpu_entities_update_tiles_hostio(NULL, &s->d, s->d_tile_row_start, s->n_tiles, 0, fr, &io);
    } else
        rc = s->tiled ? clapgpu_entities_update_tiles(NULL, &s->d, s->d_tile_row_start, s->n_tiles, 0, fr)
                      : clapgpu_entities_update(NULL, &s->d, s->level_start_host, s->n_levels, 0, fr);
    s->d.n_attach = 0;
    if (rc) return rc;
    const size_t nn = s->n_slots, cap = s->cap_slots, mask_words = nn / 64, mask_stride = cap / 64 + 2;
    if (fused) {
        CK(clapgpu_wait_word(s->h_done, s->frame_id, NULL));
        if (s->n_stale_words) stale_after_launch(s);
    } else if (s->zero_copy) {
        char *mo = s->d_out_host;
        clapgpu_entities_export x = { .mx = (float *)mo, .inv_mx = (float *)(mo + cap * 64), .aabb = (float *)(mo + cap * 128),
                                      .center = (float *)(mo + cap * 152), .vis_mask = (uint64_t *)(mo + cap * 164) };
        x.rebuilt_mask = x.vis_mask + mask_stride;
        x.inside_mask = s->bv_on ? x.rebuilt_mask + mask_stride : NULL;
        x.counter = s->d_counter; x.done = s->d_done; x.done_value = ++s->frame_id;
        CK(clapgpu_entities_export_rebuilt(NULL, &s->d, &x));
        CK(clapgpu_wait_word(s->h_done, s->frame_id, NULL));
        if (fr && s->xv.n) { CK(download_views(s)); CK(clapgpu_stream_sync(NULL)); }
    } else {
        /* staged: the masks first, then only the span of rows this launch rebuilt */
        CK(clapgpu_memcpy_d2h(s->h_mask, s->d.vis_mask, (2 * mask_stride + mask_words) * 8, NULL));
        if (fr) CK(download_views(s));
        CK(clapgpu_stream_sync(NULL));
        size_t lo = mask_words, hi = 0;
        for (size_t w = 0; w < mask_words; w++)
            if (s->h_rebuilt[w]) { if (w < lo) lo = w; hi = w + 1; }
        if (hi > lo) {
            const size_t a = lo * 64, cnt = (hi - lo) * 64;
            CK(clapgpu_memcpy_d2h(s->h_mx + 16 * a, s->d.mx + 16 * a, cnt * 64, NULL));
            CK(clapgpu_memcpy_d2h(s->h_inv + 16 * a, s->d.inv_mx + 16 * a, cnt * 64, NULL));
            CK(clapgpu_memcpy_d2h(s->h_aabb + 6 * a, s->d.aabb + 6 * a, cnt * 24, NULL));
            CK(clapgpu_memcpy_d2h(s->h_center + 3 * a, s->d.center + 3 * a, cnt * 12, NULL));
            CK(clapgpu_stream_sync(NULL));
        }
    }
    if (!fr) memset(s->h_mask, 0, mask_words * 8);
    if (!s->bv_on) memset(s->h_inside, 0, mask_words * 8);
    return CLAPGPU_OK;
}

/* ---------------------------------------------------------------- results */
/* under EXPORT_DRAWN a row nobody has read since it was rebuilt is fetched by the first accessor that asks for it */
#define RESULT(field, stride)                                                        \
    const struct ent *e = get(s, handle);                                            \
    if (e && s->n_stale_words && clapgpu_scene_fetch_entity((clapgpu_scene *)s, handle)) return NULL; \
    return (e && s->have_results && e->slot < s->n_slots) ? s->field + (stride) * (size_t)e->slot : NULL

const float *clapgpu_scene_entity_mx(const clapgpu_scene *s, uint32_t handle)          { RESULT(h_mx, 16); }
const float *clapgpu_scene_entity_inverse_mx(const clapgpu_scene *s, uint32_t handle)  { RESULT(h_inv, 16); }
const float *clapgpu_scene_entity_aabb(const clapgpu_scene *s, uint32_t handle)        { RESULT(h_aabb, 6); }
const float *clapgpu_scene_entity_aabb_center(const clapgpu_scene *s, uint32_t handle) { RESULT(h_center, 3); }

int clapgpu_scene_entity_in_frustum(const clapgpu_scene *s, uint32_t handle)
{
    const struct ent *e = get(s, handle);
    if (!e || !s->have_results || e->slot >= s->n_slots) return 0;
    return (int)((s->h_mask[e->slot >> 6] >> (e->slot & 63)) & 1);
}

void *clapgpu_scene_entity_user(const clapgpu_scene *s, uint32_t handle)
{
    const struct ent *e = get(s, handle);
    return e ? e->user : NULL;
}

uint32_t clapgpu_scene_visible(const clapgpu_scene *s, uint32_t *handles, uint32_t capacity)
{
    uint32_t cnt = 0;
    if (!s || !s->have_results) return 0;
    for (uint32_t w = 0; w < s->n_slots / 64; w++) {
        uint64_t m = s->h_mask[w];
        while (m) {
            const uint32_t bit = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            if (handles && cnt < capacity) handles[cnt] = s->slot_handle[w * 64 + bit];
            cnt++;
        }
    }
    return cnt;
}

uint32_t clapgpu_scene_entity_slot(const clapgpu_scene *s, uint32_t handle)
{
    const struct ent *e = get(s, handle);
    return (e && e->slot < s->n_slots) ? e->slot : CLAPGPU_NO_ENTITY;
}

int clapgpu_scene_results(const clapgpu_scene *s, clapgpu_scene_arrays *out)
{
    if (!s || !out || !s->have_results) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    out->n_slots = s->n_slots;
    out->mx = s->h_mx; out->inverse_mx = s->h_inv; out->aabb = s->h_aabb; out->aabb_center = s->h_center;
    out->vis_mask = s->h_mask; out->rebuilt_mask = s->h_rebuilt; out->inside_mask = s->h_inside;
    out->slot_user = (void *const *)s->slot_user;
    out->exported_mask = (s->zero_copy && s->tiled) ? s->h_exported : s->h_rebuilt;
    out->stale_mask = s->h_stale; out->fetched_mask = s->h_fetched;
    out->n_stale_words = s->n_stale_words; out->n_fetched = s->n_fetched; out->fetch_serial = s->fetch_serial;
    out->n_views = s->xv.n;
    for (uint32_t v = 0; v < CLAPGPU_EXTRA_VIEWS_MAX; v++) out->view_mask[v] = v < s->xv.n ? s->h_xv_mask[v] : NULL;
    return CLAPGPU_OK;
}

void clapgpu_scene_set_export(clapgpu_scene *s, int policy)
{
    if (s) s->export_drawn = policy == CLAPGPU_SCENE_EXPORT_DRAWN;
}

int clapgpu_scene_export_is_drawn(const clapgpu_scene *s) { return s && s->export_drawn && s->zero_copy && s->tiled; }

int clapgpu_scene_entity_keep(clapgpu_scene *s, uint32_t handle, int keep)
{
    struct ent *e = get(s, handle);
    if (!e) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (e->keep == (keep ? 1 : 0)) return CLAPGPU_OK;
    e->keep = keep ? 1 : 0;
    if (!s->topology_dirty && s->h_keep && e->slot < s->n_slots) {   /* else the re-tile lays the bits out */
        if (keep) s->h_keep[e->slot >> 6] |= 1ull << (e->slot & 63);
        else s->h_keep[e->slot >> 6] &= ~(1ull << (e->slot & 63));
        s->keep_dirty = 1;
    }
    return CLAPGPU_OK;
}

int clapgpu_scene_fetch(clapgpu_scene *s, const uint64_t *want, uint32_t *n_rows)
{
    if (!s) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (n_rows) *n_rows = 0;
    if (!s->have_results || s->topology_dirty) return s->n_stale_words ? CLAPGPU_ERR_NOT_SUPPORTED : CLAPGPU_OK;
    CK(fetch_rows(s, want, NULL, NULL));
    if (n_rows) *n_rows = s->n_fetched;
    return CLAPGPU_OK;
}

int clapgpu_scene_fetch_entity(clapgpu_scene *s, uint32_t handle)
{
    const struct ent *e = get(s, handle);
    if (!e) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!s->n_stale_words || e->slot >= s->n_slots || !((s->h_stale[e->slot >> 6] >> (e->slot & 63)) & 1)) return CLAPGPU_OK;
    if (!s->have_results || s->topology_dirty) return CLAPGPU_ERR_NOT_SUPPORTED;
    /* one row: a one-bit `want` (n / 64 words, zeroed) beside a device round trip */
    const size_t words = s->n_slots / 64;
    uint64_t *want = calloc(words ? words : 1, 8);
    if (!want) return CLAPGPU_ERR_NOMEM;
    want[e->slot >> 6] = 1ull << (e->slot & 63);
    const int rc = fetch_rows(s, want, NULL, NULL);
    free(want);
    return rc;
}

/* view_entity_in_frustum for a frustum other than the one of the last mq_update (the engine recomputes its frusta in
 * scene_cameras_calc, AFTER mq_update: clap.c:614-616): re-tests every entity's stored box, refreshes vis_mask */
int clapgpu_scene_cull(clapgpu_scene *s, const clapgpu_frustum *frustum)
{
    if (!s || !frustum) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!s->have_results || s->topology_dirty) return CLAPGPU_ERR_NOT_SUPPORTED;   /* nothing on the device yet */
    CK(apply_edits(s));
    CK(ensure_views(s));
    s->d.views = s->xv.n ? &s->xv : NULL;
    for (uint32_t v = 0; v < s->xv.n; v++) s->xv.host_vis_mask[v] = NULL;
    CK(clapgpu_entities_cull(NULL, &s->d, frustum));   /* every view of the frame from one read of the boxes */
    CK(clapgpu_memcpy_d2h(s->h_mask, s->d.vis_mask, ((size_t)s->n_slots / 64) * 8, NULL));
    CK(download_views(s));
    CK(clapgpu_stream_sync(NULL));
    s->have_frustum = 1; s->last_frustum = *frustum;
    CK(fetch_rows(s, views_union(s), NULL, NULL));     /* EXPORT_DRAWN: what the views draw and an earlier frame left stale */
    return CLAPGPU_OK;
}

/* one extra view alone: its planes moved since the launch that culled it (light_update runs after mq_update, scene.c:1166-1171) */
int clapgpu_scene_cull_view(clapgpu_scene *s, uint32_t view, const clapgpu_frustum *frustum)
{
    if (!s || !frustum || view >= s->xv_want) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!s->have_results || s->topology_dirty) return CLAPGPU_ERR_NOT_SUPPORTED;
    CK(apply_edits(s));
    CK(ensure_views(s));
    s->xv.frustum[view] = *frustum;
    clapgpu_entities one = s->d;
    one.vis_mask = s->xv.vis_mask[view]; one.vis_row_pop = s->xv.vis_row_pop[view]; one.views = NULL;
    CK(clapgpu_entities_cull(NULL, &one, frustum));
    CK(clapgpu_memcpy_d2h(s->h_xv_mask[view], s->xv.vis_mask[view], ((size_t)s->n_slots / 64) * 8, NULL));
    CK(clapgpu_stream_sync(NULL));
    CK(fetch_rows(s, s->h_xv_mask[view], NULL, NULL));
    return CLAPGPU_OK;
}

void clapgpu_scene_set_bv_points(clapgpu_scene *s, const float cam_pos[3], const float *ctl_pos, uint32_t ctl_handle)
{
    if (!s) return;
    s->bv_on = cam_pos != NULL;
    if (cam_pos) memcpy(s->bv_cam, cam_pos, 12);
    s->bv_has_ctl = ctl_pos != NULL;
    if (ctl_pos) memcpy(s->bv_ctl, ctl_pos, 12);
    s->bv_ctl_handle = ctl_handle;
}

/* ---- the render passes' LOD pick and draw list (model.c:959-992) ------------------------------------------------------ */
int clapgpu_scene_entity_lod(clapgpu_scene *s, uint32_t handle, int force_lod, int cur_lod)
{
    struct ent *e = get(s, handle);
    if (!e) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    e->force_lod = force_lod;
    e->cur_lod = cur_lod;
    if (s->lod_cap && s->lod_layout_gen == s->layout_gen && !s->topology_dirty && e->slot < s->n_slots) {
        s->h_force_lod[e->slot] = force_lod;
        s->h_cur_lod[e->slot] = cur_lod;
        if (e->slot < s->lod_lo) s->lod_lo = e->slot;
        if (e->slot + 1 > s->lod_hi) s->lod_hi = e->slot + 1;
    }
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_cur_lod(const clapgpu_scene *s, uint32_t handle)
{
    const struct ent *e = get(s, handle);
    return e ? e->cur_lod : -1;
}

static int ensure_lod(clapgpu_scene *s)
{
    if (s->lod_cap < s->cap_slots) {
        free_lod(s);
        const size_t n = s->cap_slots;
        s->h_force_lod = malloc(n * 4); s->h_cur_lod = malloc(n * 4);
        if (!s->h_force_lod || !s->h_cur_lod) return CLAPGPU_ERR_NOMEM;
        CK(clapgpu_malloc((void **)&s->d_force_lod, n * 4)); CK(clapgpu_malloc((void **)&s->d_cur_lod, n * 4));
        CK(clapgpu_malloc((void **)&s->d_visible, n * 4));   CK(clapgpu_malloc((void **)&s->d_draw_lod, n * 4));
        CK(clapgpu_malloc((void **)&s->d_visible_count, 16));
        CK(clapgpu_malloc(&s->d_vis_scratch, clapgpu_visible_scratch_bytes((uint32_t)n)));
        s->lod_mapped = s->zero_copy && n <= CLAPGPU_SCENE_LOD_MAPPED_SLOTS;
        if (s->lod_mapped) {
            CK(clapgpu_host_malloc_mapped((void **)&s->h_draw_slot, &s->a_draw_slot, n * 4));
            CK(clapgpu_host_malloc_mapped((void **)&s->h_draw_lod, &s->a_draw_lod, n * 4));
            CK(clapgpu_host_malloc_mapped((void **)&s->h_visible_count, &s->a_visible_count, 16));
        } else {
            CK(clapgpu_host_malloc((void **)&s->h_draw_slot, n * 4)); CK(clapgpu_host_malloc((void **)&s->h_draw_lod, n * 4));
            CK(clapgpu_host_malloc((void **)&s->h_visible_count, 16));
        }
        s->lod_cap = s->cap_slots;
        s->lod_layout_gen = s->layout_gen - 1;                       /* force the fill below */
    }
    if (s->lod_layout_gen != s->layout_gen) {                        /* a re-tile moved the entities: slot order anew */
        for (uint32_t i = 0; i < s->n_slots; i++) {
            const uint32_t h = s->slot_handle[i];
            s->h_force_lod[i] = h == CLAPGPU_NO_ENTITY ? -1 : s->e[h].force_lod;
            s->h_cur_lod[i] = h == CLAPGPU_NO_ENTITY ? 0 : s->e[h].cur_lod;
        }
        s->lod_lo = 0; s->lod_hi = s->n_slots;
        s->lod_layout_gen = s->layout_gen;
    }
    if (s->lod_lo < s->lod_hi) {
        const size_t off = s->lod_lo, cnt = s->lod_hi - s->lod_lo;
        CK(clapgpu_memcpy_h2d(s->d_force_lod + off, s->h_force_lod + off, cnt * 4, NULL));
        CK(clapgpu_memcpy_h2d(s->d_cur_lod + off, s->h_cur_lod + off, cnt * 4, NULL));
    }
    s->lod_lo = 0xffffffffu; s->lod_hi = 0;
    return CLAPGPU_OK;
}

int clapgpu_scene_select_lod(clapgpu_scene *s, const float cam_pos[3], uint32_t *n_draw)
{
    return clapgpu_scene_select_lod_view(s, CLAPGPU_SCENE_MAIN_VIEW, cam_pos, n_draw);
}

int clapgpu_scene_select_lod_view(clapgpu_scene *s, uint32_t view, const float cam_pos[3], uint32_t *n_draw)
{
    if (!s || !n_draw || (view != CLAPGPU_SCENE_MAIN_VIEW && view >= s->xv.n)) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    *n_draw = 0;
    if (!s->have_results || s->topology_dirty) return CLAPGPU_ERR_NOT_SUPPORTED;   /* nothing on the device yet */
    if (s->n_slots == 0) { s->n_draw = 0; return CLAPGPU_OK; }
    CK(apply_edits(s));
    CK(ensure_lod(s));
    /* the ordered visible list from the mask the last update / cull left on the device, then -- with a camera -- the LOD
     * pick over it (one launch each); without one the pass keeps every cur_lod (model.c:974: `if (camera)`) */
    uint32_t *out_slot = s->lod_mapped ? s->a_draw_slot : s->d_visible, *out_count = s->lod_mapped ? s->a_visible_count : s->d_visible_count;
    int32_t *out_lod = s->lod_mapped ? s->a_draw_lod : s->d_draw_lod;
    clapgpu_entities of_view = s->d;                     /* the plane the list is made from */
    if (view != CLAPGPU_SCENE_MAIN_VIEW) { of_view.vis_mask = s->xv.vis_mask[view]; of_view.vis_row_pop = s->xv.vis_row_pop[view]; }
    if (cam_pos)
        CK(clapgpu_visible_compact_lod(NULL, &of_view, 0, cam_pos, s->d_force_lod, s->d_cur_lod, out_slot, out_count, out_lod, s->d_vis_scratch));
    else
        CK(clapgpu_visible_compact(NULL, of_view.vis_mask, of_view.vis_row_pop, s->n_slots, 0, out_slot, out_count, s->d_vis_scratch));
    if (!s->lod_mapped) CK(clapgpu_memcpy_d2h(s->h_visible_count, s->d_visible_count, 4, NULL));
    CK(clapgpu_stream_sync(NULL));
    const uint32_t n = *s->h_visible_count;
    if (n > s->n_slots) return CLAPGPU_ERR_UNKNOWN;
    if (n && !s->lod_mapped) {
        CK(clapgpu_memcpy_d2h(s->h_draw_slot, s->d_visible, (size_t)n * 4, NULL));
        if (cam_pos) CK(clapgpu_memcpy_d2h(s->h_draw_lod, s->d_draw_lod, (size_t)n * 4, NULL));
        CK(clapgpu_stream_sync(NULL));
    }
    for (uint32_t k = 0; k < n && !(cam_pos && s->lod_sync_by_caller); k++) {   /* the host copies follow the pick: where it changed something */
        const uint32_t slot = s->h_draw_slot[k];
        if (!cam_pos) { s->h_draw_lod[k] = s->h_cur_lod[slot]; continue; }
        if (s->h_cur_lod[slot] == s->h_draw_lod[k]) continue;
        s->h_cur_lod[slot] = s->h_draw_lod[k];
        const uint32_t h = s->slot_handle[slot];
        if (h != CLAPGPU_NO_ENTITY) s->e[h].cur_lod = s->h_draw_lod[k];
    }
    s->n_draw = n;
    *n_draw = n;
    return CLAPGPU_OK;
}

/* A caller that walks the draw list anyway (and knows every entity's last LOD) tells the mirror where the pick changed one,
 * instead of the mirror comparing every entry itself: clapgpu_scene_set_lod_sync(s, 1), then clapgpu_scene_lod_picked() for
 * each changed entry of every list picked with a camera -- distinct slots may be reported from several threads at once. */
void clapgpu_scene_set_lod_sync(clapgpu_scene *s, int by_caller) { if (s) s->lod_sync_by_caller = by_caller != 0; }

void clapgpu_scene_lod_picked(clapgpu_scene *s, uint32_t slot, int lod)
{
    if (!s || !s->lod_cap || slot >= s->n_slots) return;
    s->h_cur_lod[slot] = lod;
    const uint32_t h = s->slot_handle[slot];
    if (h != CLAPGPU_NO_ENTITY) s->e[h].cur_lod = lod;
}

uint32_t clapgpu_scene_draw_list(const clapgpu_scene *s, const uint32_t **slots, const int32_t **lods)
{
    if (!s || !s->lod_cap) return 0;
    if (slots) *slots = s->h_draw_slot;
    if (lods) *lods = s->h_draw_lod;
    return s->n_draw;
}

int clapgpu_scene_layout_is_tiled(const clapgpu_scene *s) { return s ? s->tiled : 0; }
uint32_t clapgpu_scene_slot_count(const clapgpu_scene *s) { return s ? s->n_slots : 0; }
uint32_t clapgpu_scene_layout_generation(const clapgpu_scene *s) { return s ? s->layout_gen : 0; }
