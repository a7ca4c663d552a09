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
    uint8_t  live, dirty;
};

struct clapgpu_scene {
    struct ent *e;  uint32_t n_handles, cap_handles;
    uint32_t   *free_list;  uint32_t n_free;
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

    /* device */
    clapgpu_entities d;
    uint32_t   *d_tile_row_start;
    float      *d_models; uint32_t d_models_cap;
    int         have_results;
    uint32_t    layout_gen;
};

#define CK(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

static struct ent *get(const clapgpu_scene *s, uint32_t h)
{
    return (s && h < s->n_handles && s->e[h].live) ? &s->e[h] : NULL;
}

/* dirty bit 0: queued for upload; bit 1: xform.updated (transform_set_updated, transform.c:21-24) */
static void mark_dirty(clapgpu_scene *s, uint32_t h, int xform_updated)
{
    if (!s->e[h].dirty) {
        if (s->n_dirty == s->cap_dirty) {
            s->cap_dirty = s->cap_dirty ? 2 * s->cap_dirty : 1024;
            s->dirty_list = realloc(s->dirty_list, s->cap_dirty * sizeof(uint32_t));
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
        s->h_flags[slot] = e->flags | ((e->dirty & 2) ? CLAPGPU_E_DIRTY : 0);
        if (slot < s->up_lo) s->up_lo = slot;
        if (slot >= s->up_hi) s->up_hi = slot + 1;
    }
}

int clapgpu_scene_create(clapgpu_scene **out, int device)
{
    if (!out) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    CK(clapgpu_init(device));
    clapgpu_scene *s = calloc(1, sizeof(*s));
    if (!s) return CLAPGPU_ERR_NOMEM;
    s->topology_dirty = 1;
    *out = s;
    return CLAPGPU_OK;
}

static void free_device(clapgpu_scene *s)
{
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
    if (!s) return;
    free_device(s);
    if (s->d_models) clapgpu_free(s->d_models);
    free(s->e); free(s->free_list); free(s->dirty_list); free(s->models); free(s->slot_handle);
    free(s->tile_row_start_host); free(s->level_start_host);
    if (s->h_in) clapgpu_host_free(s->h_in);
    if (s->h_out) clapgpu_host_free(s->h_out);
    free(s->h_parent); free(s->h_model); free(s->slot_user);
    if (s->d_bv_result) clapgpu_free(s->d_bv_result);
    free(s);
}

int clapgpu_scene_model_new(clapgpu_scene *s, const float aabb[6], int skip_aabb, uint32_t *model)
{
    if (!s || !aabb || !model) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (s->n_models == s->cap_models) {
        s->cap_models = s->cap_models ? 2 * s->cap_models : 16;
        s->models = realloc(s->models, (size_t)s->cap_models * 8 * sizeof(float));
        if (!s->models) return CLAPGPU_ERR_NOMEM;
    }
    float *row = s->models + 8 * (size_t)s->n_models;
    uint32_t skip = skip_aabb ? 1u : 0u;
    row[0] = aabb[0]; row[1] = aabb[1]; row[2] = aabb[2];
    memcpy(&row[3], &skip, 4);
    row[4] = aabb[3]; row[5] = aabb[4]; row[6] = aabb[5]; row[7] = 0.f;
    *model = s->n_models++;
    s->models_dirty = 1;
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_new(clapgpu_scene *s, uint32_t model, void *user, uint32_t *handle)
{
    if (!s || !handle || model >= s->n_models) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    uint32_t h;
    if (s->n_free) {
        h = s->free_list[--s->n_free];
    } else {
        if (s->n_handles == s->cap_handles) {
            s->cap_handles = s->cap_handles ? 2 * s->cap_handles : 1024;
            s->e = realloc(s->e, (size_t)s->cap_handles * sizeof(struct ent));
            if (!s->e) return CLAPGPU_ERR_NOMEM;
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
    s->topology_dirty = 1;
    *handle = h;
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_delete(clapgpu_scene *s, uint32_t handle)
{
    struct ent *e = get(s, handle);
    if (!e) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    for (uint32_t h = 0; h < s->n_handles; h++)         /* orphans become roots, like a NULL e->parent */
        if (s->e[h].live && s->e[h].parent == handle)
            s->e[h].parent = CLAPGPU_NO_ENTITY;
    e->live = 0;
    s->free_list = realloc(s->free_list, ((size_t)s->n_free + 1) * sizeof(uint32_t));
    s->free_list[s->n_free++] = handle;
    s->topology_dirty = 1;
    return CLAPGPU_OK;
}

int clapgpu_scene_entity_set_parent(clapgpu_scene *s, uint32_t handle, uint32_t parent)
{
    struct ent *e = get(s, handle);
    if (!e || (parent != CLAPGPU_NO_ENTITY && (!get(s, parent) || parent == handle)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (e->parent != parent) {
        e->parent = parent;
        s->topology_dirty = 1;
    }
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
        s->h_flags[e->slot] = e->flags | (xform_updated ? CLAPGPU_E_DIRTY : 0);
    } else {
        e->dirty |= xform_updated ? 3 : 1;               /* picked up by the re-tile's full image */
    }
    return CLAPGPU_OK;
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
    RE(s->h_parent, n * 4); RE(s->h_model, n * 4); RE(s->slot_handle, n * 4); RE(s->slot_user, n * sizeof(void *));
#undef RE
    /* retile() rewrites the upload image in full and downloads are overwritten by the next frame, so
     * nothing has to survive the growth */
    if (s->h_in) clapgpu_host_free(s->h_in);
    if (s->h_out) clapgpu_host_free(s->h_out);
    s->h_in = s->h_out = NULL;
    free_device(s);
    s->models_dirty = 1;                                /* free_device() dropped d.model_table */
    s->have_results = 0;
    s->in_bytes = n * 36;
    s->out_bytes = n * 164 + 3 * (n / 64 + 2) * 8;           /* + visibility, rebuilt and bounding-volume masks */
    CK(clapgpu_host_malloc(&s->h_in, s->in_bytes));
    CK(clapgpu_host_malloc(&s->h_out, s->out_bytes));
    CK(clapgpu_malloc(&s->d_in, s->in_bytes));
    CK(clapgpu_malloc(&s->d_out, s->out_bytes));
    char *hi = s->h_in, *ho = s->h_out, *di = s->d_in, *dq = s->d_out;
    s->h_pos_scale = (float *)hi;              s->d.pos_scale = (const float *)di;
    s->h_rot = (float *)(hi + n * 16);         s->d.rot = (const float *)(di + n * 16);
    s->h_flags = (uint32_t *)(hi + n * 32);    s->d.flags = (uint32_t *)(di + n * 32);
    s->h_mx = (float *)ho;                     s->d.mx = (float *)dq;
    s->h_inv = (float *)(ho + n * 64);         s->d.inv_mx = (float *)(dq + n * 64);
    s->h_aabb = (float *)(ho + n * 128);       s->d.aabb = (float *)(dq + n * 128);
    s->h_center = (float *)(ho + n * 152);     s->d.center = (float *)(dq + n * 152);
    s->h_mask = (uint64_t *)(ho + n * 164);    s->d.vis_mask = (uint64_t *)(dq + n * 164);
    s->h_rebuilt = s->h_mask + (n / 64 + 2);   s->d.rebuilt_mask = s->d.vis_mask + (n / 64 + 2);
    s->h_inside = s->h_rebuilt + (n / 64 + 2); s->bvq.inside_mask = s->d.rebuilt_mask + (n / 64 + 2);
    void **dp[] = { (void **)&s->d.parent, (void **)&s->d.model, (void **)&s->d.seqs, (void **)&s->d.vis_row_pop,
                    (void **)&s->d_tile_row_start };
    size_t sz[] = { n * 4, n * 4, n * 4, (n / 64 + 16) / 16 * 16, (n / 64 + 2) * 4 };
    for (unsigned i = 0; i < sizeof(dp) / sizeof(dp[0]); i++)
        CK(clapgpu_malloc(dp[i], sz[i]));
    s->cap_slots = cap;
    return CLAPGPU_OK;
}

/* depth of every live entity under its root; returns max depth + 1, or 0 on a parent cycle */
static uint32_t compute_depths(clapgpu_scene *s, uint32_t *depth, uint32_t *root)
{
    const uint32_t UNK = 0xffffffffu;
    uint32_t maxd = 0;
    for (uint32_t h = 0; h < s->n_handles; h++) depth[h] = UNK;
    for (uint32_t h = 0; h < s->n_handles; h++) {
        if (!s->e[h].live || depth[h] != UNK) continue;
        uint32_t cur = h, steps = 0;                    /* walk up to a known ancestor */
        while (s->e[cur].parent != CLAPGPU_NO_ENTITY && depth[s->e[cur].parent] == UNK) {
            cur = s->e[cur].parent;
            if (++steps > s->n_handles) return 0;
        }
        uint32_t base_d, base_r;
        if (s->e[cur].parent == CLAPGPU_NO_ENTITY) { base_d = 0; base_r = cur; }
        else { base_d = depth[s->e[cur].parent] + 1; base_r = root[s->e[cur].parent]; }
        /* second walk: assign from h upward needs distances; count chain length first */
        uint32_t len = 0;
        for (uint32_t x = h; x != cur; x = s->e[x].parent) len++;
        uint32_t x = h;
        for (uint32_t k = 0; k <= len; k++) {
            depth[x] = base_d + (len - k);
            root[x] = base_r;
            if (depth[x] + 1 > maxd) maxd = depth[x] + 1;
            x = s->e[x].parent;
        }
    }
    return maxd ? maxd : 1;
}

static int retile(clapgpu_scene *s)
{
    const uint32_t H = s->n_handles;
    uint32_t *depth = malloc(((size_t)H + 1) * 4), *root = malloc(((size_t)H + 1) * 4);
    uint32_t *tree_of = malloc(((size_t)H + 1) * 4);
    if (!depth || !root || !tree_of) return CLAPGPU_ERR_NOMEM;
    uint32_t maxd = compute_depths(s, depth, root);
    if (!maxd) { free(depth); free(root); free(tree_of); return CLAPGPU_ERR_INVALID_ARGUMENTS; }

    uint32_t n_trees = 0, n_live = 0;
    for (uint32_t h = 0; h < H; h++)
        if (s->e[h].live) { n_live++; if (s->e[h].parent == CLAPGPU_NO_ENTITY) tree_of[h] = n_trees++; }
    uint32_t *width = calloc((size_t)(n_trees ? n_trees : 1) * maxd, 4);
    if (!width) return CLAPGPU_ERR_NOMEM;
    int tiled = 1;
    for (uint32_t h = 0; h < H; h++)
        if (s->e[h].live) {
            uint32_t t = tree_of[root[h]];
            if (++width[(size_t)t * maxd + depth[h]] > WAVE) tiled = 0;
        }

    uint32_t n_rows = 0;
    uint32_t *row_of_tree = malloc(((size_t)n_trees + 1) * 4);       /* first row of the tree's tile */
    uint32_t *row_fill = NULL;
    free(s->tile_row_start_host);
    free(s->level_start_host);
    s->tile_row_start_host = malloc(((size_t)n_trees + 2) * 4);      /* at most one tile per tree */
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
            for (uint32_t d = 0; d < maxd; d++) { if (fill[d] + w[d] > WAVE) fits = 0; if (w[d]) rows = d + 1; }
            if (!fits) {                                            /* close the tile */
                s->tile_row_start_host[s->n_tiles++] = tile_first_row;
                tile_first_row += tile_rows;
                tile_rows = 0;
                memset(fill, 0, maxd * 4);
            }
            for (uint32_t d = 0; d < maxd; d++) fill[d] += w[d];
            if (rows > tile_rows) tile_rows = rows;
            row_of_tree[t] = tile_first_row;
        }
        if (n_trees) { s->tile_row_start_host[s->n_tiles++] = tile_first_row; tile_first_row += tile_rows; }
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
    int rc = ensure_slots(s, n_rows * WAVE);
    if (rc) return rc;
    s->n_rows = n_rows;
    s->n_slots = n_rows * WAVE;
    s->tiled = tiled;
    if (!tiled) s->level_start_host[s->n_levels] = s->n_slots;      /* the kernel wants the last start == n */

    /* slots: handle order inside each row */
    row_fill = calloc(n_rows, 4);
    for (uint32_t i = 0; i < s->n_slots; i++) s->slot_handle[i] = CLAPGPU_NO_ENTITY;
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
    /* full staging image */
    for (uint32_t i = 0; i < s->n_slots; i++) {
        const uint32_t h = s->slot_handle[i];
        s->slot_user[i] = h == CLAPGPU_NO_ENTITY ? NULL : s->e[h].user;
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
        s->h_flags[i] = e->flags | CLAPGPU_E_DIRTY;                  /* everything is rebuilt after a re-tile */
    }
    free(depth); free(root); free(tree_of); free(width); free(row_of_tree); free(row_fill);

    s->d.n = s->n_slots;
    const size_t n = s->n_slots;
    CK(clapgpu_memcpy_h2d((void *)s->d.parent, s->h_parent, n * 4, NULL));
    CK(clapgpu_memcpy_h2d((void *)s->d.model, s->h_model, n * 4, NULL));
    CK(clapgpu_memset(s->d.seqs, 0, n * 4, NULL));
    CK(clapgpu_memset(s->d_out, 0, s->out_bytes, NULL));
    if (tiled)
        CK(clapgpu_memcpy_h2d(s->d_tile_row_start, s->tile_row_start_host, ((size_t)s->n_tiles + 1) * 4, NULL));
    for (uint32_t k = 0; k < s->n_dirty; k++) s->e[s->dirty_list[k]].dirty = 0;
    s->n_dirty = 0;
    s->topology_dirty = 0;
    s->up_lo = 0xffffffffu; s->up_hi = 0;
    s->layout_gen++;
    return CLAPGPU_OK;
}

int clapgpu_scene_mq_update(clapgpu_scene *s, const clapgpu_frustum *frustum)
{
    if (!s) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    int upload = 0, full = 0;
    uint32_t lo = 0xffffffffu, hi = 0, n_touched = 0;
    if (s->topology_dirty) {
        CK(retile(s));
        upload = full = 1;
    } else if (s->n_dirty) {
        /* the upload image was written as the verbs came in (mark_dirty); here only the bookkeeping */
        for (uint32_t k = 0; k < s->n_dirty; k++) {
            struct ent *e = &s->e[s->dirty_list[k]];
            e->dirty = 0;
            if (!e->live) continue;
            s->dirty_list[n_touched++] = e->slot;        /* the list is reused for the slots touched */
        }
        lo = s->up_lo; hi = s->up_hi;
        s->n_dirty = 0;
        upload = n_touched != 0 && hi > lo;
    }
    if (s->bulk_dirty && !full) {                        /* clapgpu_scene_entity_transform_mt wrote the image directly */
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
    if (s->n_models == 0) return CLAPGPU_OK;
    const size_t n = s->n_slots;
    const size_t cap = s->cap_slots;
    if (upload) {
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
        if (!s->d_bv_result) CK(clapgpu_malloc((void **)&s->d_bv_result, 8));
        memcpy(s->bvq.cam_pos, s->bv_cam, 12); memcpy(s->bvq.ctl_pos, s->bv_ctl, 12);
        const struct ent *ce = s->bv_has_ctl ? get(s, s->bv_ctl_handle) : NULL;
        s->bvq.has_ctl = s->bv_has_ctl; s->bvq.ctl_entity = ce ? ce->slot : 0xffffffffu;
        s->bvq.result = s->d_bv_result;
        s->d.bv = &s->bvq;
    } else {
        s->d.bv = NULL;
    }
    if (s->tiled)
        CK(clapgpu_entities_update_tiles(NULL, &s->d, s->d_tile_row_start, s->n_tiles, 0, frustum));
    else
        CK(clapgpu_entities_update(NULL, &s->d, s->level_start_host, s->n_levels, 0, frustum));
    const size_t mask_words = n / 64, mask_stride = cap / 64 + 2;
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
    if (!frustum)
        memset(s->h_mask, 0, mask_words * 8);
    CK(clapgpu_stream_sync(NULL));
    if (!s->bv_on) memset(s->h_inside, 0, mask_words * 8);
    if (full || 4 * (size_t)n_touched > n)
        for (size_t i = 0; i < n; i++) s->h_flags[i] &= ~CLAPGPU_E_DIRTY;   /* the kernel cleared its copy too */
    else
        for (uint32_t k = 0; k < n_touched; k++) s->h_flags[s->dirty_list[k]] &= ~CLAPGPU_E_DIRTY;
    s->have_results = 1;
    return CLAPGPU_OK;
}

/* ---------------------------------------------------------------- results */
#define RESULT(field, stride)                                                        \
    const struct ent *e = get(s, handle);                                            \
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
    return CLAPGPU_OK;
}

/* view_entity_in_frustum for a frustum other than the one of the last mq_update (the engine recomputes its frusta in
 * scene_cameras_calc, AFTER mq_update: clap.c:614-616): re-tests every entity's stored box, refreshes vis_mask */
int clapgpu_scene_cull(clapgpu_scene *s, const clapgpu_frustum *frustum)
{
    if (!s || !frustum) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!s->have_results || s->topology_dirty) return CLAPGPU_ERR_NOT_SUPPORTED;   /* nothing on the device yet */
    CK(clapgpu_entities_cull(NULL, &s->d, frustum));
    CK(clapgpu_memcpy_d2h(s->h_mask, s->d.vis_mask, ((size_t)s->n_slots / 64) * 8, NULL));
    CK(clapgpu_stream_sync(NULL));
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

int clapgpu_scene_layout_is_tiled(const clapgpu_scene *s) { return s ? s->tiled : 0; }
uint32_t clapgpu_scene_slot_count(const clapgpu_scene *s) { return s ? s->n_slots : 0; }
uint32_t clapgpu_scene_layout_generation(const clapgpu_scene *s) { return s ? s->layout_gen : 0; }
