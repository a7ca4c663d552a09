/*
 * tests/c/fake_clapgpu.c -- TEST INFRASTRUCTURE ONLY.  Never shipped, never linked into clap_amd/, earns no parity credit.
 *
 * A CPU stand-in for the entry points of libclapgpu.so (include/clapgpu.h) that the C host mirror
 * (clap_amd/host/clapgpu_scene.c) and the CLAP-side binding (clap_amd/binding/gpu-scene.c) call, so that those two --
 * the largest and most pointer-heavy host C of the product: handle tables, tombstones, re-tiling, a hand-rolled worker
 * pool, the write-back policies -- can run in a container WITHOUT a GPU under -fsanitize=address,undefined and
 * -fsanitize=thread (tests/test_sanitize_host.py builds oracle/ref/dropin.c against it).  The reference's debug preset
 * runs ASan + UBSan over its own host code (/root/reference/CMakeLists.txt:17-18, 33-39); this is the same for ours.
 *
 * "Device" memory is malloc; a mapped allocation's device alias is its host pointer; copies are memcpy; streams are
 * immediate.  The entity entry points are served by the oracle (oracle/entity.c, oracle/lod.c) in one ascending pass
 * over the slots -- the mirror's layouts put every parent in a lower slot than its children -- with the device's
 * contract around it: touched inputs, rebuilt / visibility / containment / exported masks, row popcounts, export policy,
 * completion word.  What is under test is the HOST code's memory and thread behaviour, not these results.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "clapgpu.h"
#include "clap_oracle.h"

static const char *g_err = "";
static int g_fail_after = -1;

int clapgpu_device_count(void) { return 1; }
int clapgpu_init(int device) { return device == 0 ? CLAPGPU_OK : CLAPGPU_ERR_INVALID_ARGUMENTS; }
const char *clapgpu_last_error(void) { return g_err; }
void clapgpu_test_fail_after(int launches) { g_fail_after = launches; }
uint32_t clapgpu_abi_version(void) { return 0x7fffffffu; }

static int launch_ok(const char *what)
{
    if (g_fail_after < 0) return 1;
    if (g_fail_after == 0) { g_err = what; return 0; }
    g_fail_after--;
    return 1;
}

int clapgpu_malloc(void **dev, size_t bytes)
{
    if (!dev) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    *dev = malloc(bytes ? bytes : 1);               /* uninitialised like device memory: MSan-style surprises show as ASan-clean garbage */
    return *dev ? CLAPGPU_OK : CLAPGPU_ERR_NOMEM;
}
int clapgpu_free(void *dev) { free(dev); return CLAPGPU_OK; }
int clapgpu_host_malloc(void **host, size_t bytes) { return clapgpu_malloc(host, bytes); }
int clapgpu_host_malloc_mapped(void **host, void **dev_alias, size_t bytes)
{
    if (!host || !dev_alias) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    *host = calloc(1, bytes ? bytes : 1);
    *dev_alias = *host;
    return *host ? CLAPGPU_OK : CLAPGPU_ERR_NOMEM;
}
int clapgpu_host_free(void *host) { free(host); return CLAPGPU_OK; }
int clapgpu_memcpy_h2d(void *dev, const void *host, size_t bytes, void *stream) { (void)stream; if (bytes) memcpy(dev, host, bytes); return CLAPGPU_OK; }
int clapgpu_memcpy_d2h(void *host, const void *dev, size_t bytes, void *stream) { (void)stream; if (bytes) memcpy(host, dev, bytes); return CLAPGPU_OK; }
int clapgpu_memset(void *dev, int value, size_t bytes, void *stream) { (void)stream; if (bytes) memset(dev, value, bytes); return CLAPGPU_OK; }
int clapgpu_stream_sync(void *stream) { (void)stream; return CLAPGPU_OK; }
int clapgpu_wait_word(const volatile uint32_t *word, uint32_t value, void *stream)
{
    (void)stream;
    if (*word == value) return CLAPGPU_OK;
    g_err = "clapgpu_wait_word: the signalling launch did not run";
    return CLAPGPU_ERR_UNKNOWN;
}

/* ---- entities ---------------------------------------------------------------------------------------------------------- */
static int model_arrays(const clapgpu_entities *e, float **aabb, uint8_t **skip, uint8_t **lod)
{
    const uint32_t nm = e->n_models ? e->n_models : 1;
    *aabb = malloc((size_t)nm * 6 * sizeof(float));
    *skip = malloc(nm);
    *lod = malloc((size_t)nm * 2);
    if (!*aabb || !*skip || !*lod) { free(*aabb); free(*skip); free(*lod); return CLAPGPU_ERR_NOMEM; }
    for (uint32_t m = 0; m < nm; m++) {
        const float *row = e->model_table + 8 * (size_t)m;
        uint32_t s, l;
        memcpy(&s, &row[3], 4); memcpy(&l, &row[7], 4);
        (*aabb)[6 * m + 0] = row[0]; (*aabb)[6 * m + 1] = row[1]; (*aabb)[6 * m + 2] = row[2];
        (*aabb)[6 * m + 3] = row[4]; (*aabb)[6 * m + 4] = row[5]; (*aabb)[6 * m + 5] = row[6];
        (*skip)[m] = s != 0;
        (*lod)[2 * m] = (uint8_t)(l & 0xff); (*lod)[2 * m + 1] = (uint8_t)((l >> 8) & 0xff);
    }
    return CLAPGPU_OK;
}

static inline int in_box(const float p[3], const float *b)
{
    return p[0] >= b[0] && p[0] <= b[3] && p[1] >= b[1] && p[1] <= b[4] && p[2] >= b[2] && p[2] <= b[5];
}

/* one update launch: the oracle's pass, then the masks the kernels leave behind */
/* clapgpu_entities.views: every further view's plane from the same boxes (and the mirror's mapped words, for a hostio launch) */
static int cull_views(const clapgpu_entities *e, int hostio)
{
    const clapgpu_views *v = e->views;
    if (!v || !v->n) return CLAPGPU_OK;
    if (v->n > CLAPGPU_EXTRA_VIEWS_MAX) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const uint32_t words = (e->n + 63) / 64;
    for (uint32_t k = 0; k < v->n; k++) {
        if (!v->vis_mask[k] || !v->vis_row_pop[k]) return CLAPGPU_ERR_INVALID_ARGUMENTS;
        clapo_entities_cull(e->n, e->flags, e->aabb, (const clapo_frustum *)&v->frustum[k], NULL, v->vis_mask[k]);
        for (uint32_t w = 0; w < words; w++) v->vis_row_pop[k][w] = (uint8_t)__builtin_popcountll(v->vis_mask[k][w]);
        if (hostio && v->host_vis_mask[k]) memcpy(v->host_vis_mask[k], v->vis_mask[k], (size_t)(e->n / 64) * 8);
    }
    return CLAPGPU_OK;
}

static int run_update(const clapgpu_entities *e, uint32_t mode, const clapgpu_frustum *fr, uint64_t *rebuilt_out)
{
    if (mode) return CLAPGPU_ERR_NOT_SUPPORTED;
    if (!launch_ok("k_entities (fake)")) return CLAPGPU_ERR_UNKNOWN;
    const uint32_t n = e->n, words = (n + 63) / 64;
    float *maabb; uint8_t *mskip, *mlod;
    int rc = model_arrays(e, &maabb, &mskip, &mlod);
    if (rc) return rc;
    uint32_t *before = malloc((size_t)(n ? n : 1) * 4);
    int32_t *mi = malloc((size_t)(n ? n : 1) * 4);
    if (!before || !mi) { free(before); free(mi); free(maabb); free(mskip); free(mlod); return CLAPGPU_ERR_NOMEM; }
    memcpy(before, e->seqs, (size_t)n * 4);
    for (uint32_t i = 0; i < n; i++) mi[i] = (uint32_t)e->model[i] < (e->n_models ? e->n_models : 1) ? e->model[i] : 0;
    const uint32_t na = (e->attach && e->jt_pool && e->bind_pool && e->attach_local) ? e->n_attach : 0;
    clapo_entities_update_range(0, n, e->pos_scale, e->rot, e->parent, mi, maabb, mskip, e->flags, e->seqs, e->mx, e->inv_mx,
                                e->aabb, e->center, na, (const clapo_attach *)e->attach, e->jt_pool, e->bind_pool);
    uint64_t *rb = rebuilt_out ? rebuilt_out : e->rebuilt_mask;
    if (rb) {
        memset(rb, 0, (size_t)words * 8);
        for (uint32_t i = 0; i < n; i++)
            if (before[i] != e->seqs[i]) rb[i >> 6] |= 1ull << (i & 63);
        if (rebuilt_out && e->rebuilt_mask) memcpy(e->rebuilt_mask, rb, (size_t)words * 8);
    }
    if (e->bv && (e->bv->result || e->bv->inside_mask)) {
        const clapgpu_bv_query *q = e->bv;
        if (q->inside_mask) memset(q->inside_mask, 0, (size_t)words * 8);
        for (uint32_t i = 0; i < n; i++) {
            if (!(e->flags[i] & CLAPGPU_E_ALIVE)) continue;
            const float *b = e->aabb + 6 * (size_t)i;
            int inside = in_box(q->cam_pos, b) || (q->has_ctl && in_box(q->ctl_pos, b));
            if (inside && q->has_ctl && i == q->ctl_entity) inside = 0;
            if (inside && q->inside_mask) q->inside_mask[i >> 6] |= 1ull << (i & 63);
        }
    }
    if (fr) {
        clapo_entities_cull(n, e->flags, e->aabb, (const clapo_frustum *)fr, NULL, e->vis_mask);
        for (uint32_t w = 0; w < words; w++) e->vis_row_pop[w] = (uint8_t)__builtin_popcountll(e->vis_mask[w]);
        rc = cull_views(e, rebuilt_out != NULL);
    }
    free(before); free(mi); free(maabb); free(mskip); free(mlod);
    return rc;
}

static int check_entities(const clapgpu_entities *e, int need_mask)
{
    if (!e || !e->pos_scale || !e->rot || !e->parent || !e->model || !e->model_table || !e->flags || !e->seqs || !e->mx ||
        !e->inv_mx || !e->aabb || !e->center || (need_mask && (!e->vis_mask || !e->vis_row_pop)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    return CLAPGPU_OK;
}

int clapgpu_entities_update(void *stream, const clapgpu_entities *e, const uint32_t *level_start, uint32_t n_levels, uint32_t mode,
                            const clapgpu_frustum *frustum)
{
    (void)stream;
    int rc = check_entities(e, frustum != NULL);
    if (rc) return rc;
    if (!level_start || (e->n && !n_levels)) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!e->n) return CLAPGPU_OK;
    if (level_start[0] != 0 || level_start[n_levels] != e->n) return CLAPGPU_ERR_OUT_OF_BOUNDS;
    return run_update(e, mode, frustum, NULL);
}

int clapgpu_entities_update_tiles(void *stream, const clapgpu_entities *e, const uint32_t *tile_row_start, uint32_t n_tiles,
                                  uint32_t mode, const clapgpu_frustum *frustum)
{
    (void)stream;
    int rc = check_entities(e, frustum != NULL);
    if (rc) return rc;
    if (!e->n || !n_tiles) return CLAPGPU_OK;
    if (!tile_row_start) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (tile_row_start[n_tiles] * 64u > e->n) return CLAPGPU_ERR_OUT_OF_BOUNDS;       /* (the device clips; a mirror never asks) */
    return run_update(e, mode, frustum, NULL);
}

int clapgpu_entities_cull(void *stream, const clapgpu_entities *e, const clapgpu_frustum *frustum)
{
    (void)stream;
    if (!e || !frustum || !e->flags || !e->aabb || !e->vis_mask || !e->vis_row_pop) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!launch_ok("k_entities_cull (fake)")) return CLAPGPU_ERR_UNKNOWN;
    clapo_entities_cull(e->n, e->flags, e->aabb, (const clapo_frustum *)frustum, NULL, e->vis_mask);
    for (uint32_t w = 0; w < (e->n + 63) / 64; w++) e->vis_row_pop[w] = (uint8_t)__builtin_popcountll(e->vis_mask[w]);
    return cull_views(e, 0);
}

int clapgpu_entities_apply_inputs(void *stream, const clapgpu_entities *e, const clapgpu_entity_input *list, uint32_t n_list)
{
    (void)stream;
    if (!e || !e->pos_scale || !e->rot || !e->flags || (n_list && !list)) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!launch_ok("k_entities_apply_inputs (fake)")) return CLAPGPU_ERR_UNKNOWN;
    for (uint32_t k = 0; k < n_list; k++) {
        const uint32_t slot = list[k].slot;
        if (slot >= e->n) continue;
        memcpy((float *)e->pos_scale + 4 * (size_t)slot, list[k].pos_scale, 16);
        memcpy((float *)e->rot + 4 * (size_t)slot, list[k].rot, 16);
        e->flags[slot] = list[k].flags;
    }
    return CLAPGPU_OK;
}

int clapgpu_entities_place(void *stream, const clapgpu_entities *e, const clapgpu_entity_place *list, uint32_t n_list, uint64_t *stale_mask)
{
    (void)stream;
    if (!e || !e->parent || !e->model || !e->aabb || !e->center || (n_list && !list)) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!launch_ok("k_entities_place (fake)")) return CLAPGPU_ERR_UNKNOWN;
    for (uint32_t k = 0; k < n_list; k++) {
        const uint32_t slot = list[k].slot;
        if (slot >= e->n) continue;
        ((int32_t *)e->parent)[slot] = list[k].parent;
        ((int32_t *)e->model)[slot] = list[k].model;
        if ((list[k].flags & CLAPGPU_PLACE_CLEAR_STALE) && stale_mask) stale_mask[slot >> 6] &= ~(1ull << (slot & 63));
        if (list[k].flags & CLAPGPU_PLACE_ZERO_BOX) { memset(e->aabb + 6 * (size_t)slot, 0, 24); memset(e->center + 3 * (size_t)slot, 0, 12); }
    }
    return CLAPGPU_OK;
}

static void copy_rows(const clapgpu_entities *e, float *o_mx, float *o_inv, float *o_aabb, float *o_center, const uint64_t *mask)
{
    for (uint32_t i = 0; i < e->n; i++) {
        if (!((mask[i >> 6] >> (i & 63)) & 1)) continue;
        memcpy(o_mx + 16 * (size_t)i, e->mx + 16 * (size_t)i, 64);
        memcpy(o_inv + 16 * (size_t)i, e->inv_mx + 16 * (size_t)i, 64);
        memcpy(o_aabb + 6 * (size_t)i, e->aabb + 6 * (size_t)i, 24);
        memcpy(o_center + 3 * (size_t)i, e->center + 3 * (size_t)i, 12);
    }
}

int clapgpu_entities_export_rebuilt(void *stream, const clapgpu_entities *e, const clapgpu_entities_export *x)
{
    (void)stream;
    if (!e || !x || !e->rebuilt_mask || !x->mx || !x->inv_mx || !x->aabb || !x->center || !x->rebuilt_mask || !x->counter || !x->done ||
        (e->vis_mask && !x->vis_mask) || (e->n & 63u))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!launch_ok("k_entities_export_rebuilt (fake)")) return CLAPGPU_ERR_UNKNOWN;
    const size_t words = e->n / 64;
    copy_rows(e, x->mx, x->inv_mx, x->aabb, x->center, e->rebuilt_mask);
    memcpy(x->rebuilt_mask, e->rebuilt_mask, words * 8);
    if (e->vis_mask) memcpy(x->vis_mask, e->vis_mask, words * 8);
    if (x->inside_mask) {
        if (e->bv && e->bv->inside_mask) memcpy(x->inside_mask, e->bv->inside_mask, words * 8);
        else memset(x->inside_mask, 0, words * 8);
    }
    *x->done = x->done_value;
    return CLAPGPU_OK;
}

int clapgpu_entities_export_rows(void *stream, const clapgpu_entities *e, const clapgpu_entities_export *x, const uint64_t *select_mask)
{
    (void)stream;
    if (!e || !x || !select_mask || !x->mx || !x->inv_mx || !x->aabb || !x->center || !x->counter || !x->done || (e->n & 63u))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!launch_ok("k_entities_export_rows (fake)")) return CLAPGPU_ERR_UNKNOWN;
    copy_rows(e, x->mx, x->inv_mx, x->aabb, x->center, select_mask);
    if (x->stale_mask)
        for (uint32_t w = 0; w < e->n / 64; w++) x->stale_mask[w] &= ~select_mask[w];
    *x->done = x->done_value;
    return CLAPGPU_OK;
}

int clapgpu_entities_update_tiles_hostio(void *stream, const clapgpu_entities *e, const uint32_t *tile_row_start, uint32_t n_tiles,
                                         uint32_t mode, const clapgpu_frustum *frustum, const clapgpu_entities_hostio *io)
{
    (void)stream;
    int rc = check_entities(e, frustum != NULL);
    if (rc) return rc;
    if (!io || !io->mx || !io->inv_mx || !io->aabb || !io->center || !io->rebuilt_mask || !io->counter || !io->done ||
        (frustum && !io->vis_mask) || (io->touched && (!io->pos_scale || !io->rot || !io->flags)) || (n_tiles && !tile_row_start))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const uint32_t n = e->n, words = n / 64;
    if (io->touched)
        for (uint32_t i = 0; i < n; i++)
            if ((io->touched[i >> 6] >> (i & 63)) & 1) {
                memcpy((float *)e->pos_scale + 4 * (size_t)i, io->pos_scale + 4 * (size_t)i, 16);
                memcpy((float *)e->rot + 4 * (size_t)i, io->rot + 4 * (size_t)i, 16);
                e->flags[i] = io->flags[i];
            }
    uint64_t *rb = calloc(words ? words : 1, 8);
    if (!rb) return CLAPGPU_ERR_NOMEM;
    rc = (n && n_tiles) ? run_update(e, mode, frustum, rb) : CLAPGPU_OK;
    if (rc) { free(rb); return rc; }
    const uint64_t *inside = (e->bv && e->bv->inside_mask) ? e->bv->inside_mask : NULL;
    for (uint32_t w = 0; w < words; w++) {
        uint64_t want = ~0ull;
        if (io->keep_mask) {
            want = io->keep_mask[w] | (inside ? inside[w] : 0);
            want = frustum ? (want | e->vis_mask[w]) : ~0ull;
            for (uint32_t k = 0; frustum && e->views && k < e->views->n; k++) want |= e->views->vis_mask[k][w];   /* drawn by any view */
        }
        uint64_t ex = rb[w] & want, alive = 0;
        for (uint32_t l = 0; l < 64; l++) alive |= (uint64_t)((e->flags[w * 64 + l] & CLAPGPU_E_ALIVE) != 0) << l;
        if (io->stale_mask) {                                        /* the device's twin of the mirror's stale mask, and the rows read now */
            const uint64_t st = io->stale_mask[w];
            const uint64_t late = (io->options & CLAPGPU_HOSTIO_EXPORT_STALE_READ) ? st & want & ~rb[w] & alive : 0;
            io->stale_mask[w] = (st | rb[w]) & ~(ex | late);
            ex |= late;
        }
        io->rebuilt_mask[w] = rb[w];
        if (io->exported_mask) io->exported_mask[w] = ex;
        if (frustum) io->vis_mask[w] = e->vis_mask[w];
        if (io->inside_mask) io->inside_mask[w] = inside ? inside[w] : 0;
        rb[w] = ex;
    }
    copy_rows(e, io->mx, io->inv_mx, io->aabb, io->center, rb);
    free(rb);
    *io->done = io->done_value;
    return CLAPGPU_OK;
}

size_t clapgpu_visible_scratch_bytes(uint32_t n) { return ((size_t)n / 4096 + 1) * 4; }

int clapgpu_visible_compact(void *stream, const uint64_t *vis_mask, const uint8_t *vis_row_pop, uint32_t n, uint32_t index_base,
                            uint32_t *visible, uint32_t *count, void *scratch)
{
    (void)stream; (void)vis_row_pop; (void)scratch;
    if (!count || (n && (!vis_mask || !visible))) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!launch_ok("k_visible_expand (fake)")) return CLAPGPU_ERR_UNKNOWN;
    uint32_t c = 0;
    for (uint32_t i = 0; i < n; i++)
        if ((vis_mask[i >> 6] >> (i & 63)) & 1) visible[c++] = index_base + i;
    *count = c;
    return CLAPGPU_OK;
}

int clapgpu_entities_lod(void *stream, const clapgpu_entities *e, const uint32_t *visible, const uint32_t *count, uint32_t index_base,
                         const float cam_pos[3], const int32_t *force_lod, int32_t *cur_lod, int32_t *draw_lod)
{
    (void)stream;
    if (!e || !visible || !count || !cam_pos || !cur_lod || !draw_lod || !e->aabb || !e->center || !e->pos_scale || !e->model ||
        !e->model_table || index_base)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!launch_ok("k_entities_lod (fake)")) return CLAPGPU_ERR_UNKNOWN;
    float *maabb; uint8_t *mskip, *mlod;
    int rc = model_arrays(e, &maabb, &mskip, &mlod);
    if (rc) return rc;
    int32_t *none = NULL;
    if (!force_lod) {
        none = malloc((size_t)(e->n ? e->n : 1) * 4);
        if (!none) { free(maabb); free(mskip); free(mlod); return CLAPGPU_ERR_NOMEM; }
        for (uint32_t i = 0; i < e->n; i++) none[i] = -1;
    }
    clapo_entities_lod(*count < e->n ? *count : e->n, visible, cam_pos, e->aabb, e->center, e->pos_scale, e->model, maabb, mlod,
                       force_lod ? force_lod : none, cur_lod, draw_lod);
    free(none); free(maabb); free(mskip); free(mlod);
    return CLAPGPU_OK;
}

int clapgpu_visible_compact_lod(void *stream, const clapgpu_entities *e, uint32_t index_base, const float cam_pos[3],
                                const int32_t *force_lod, int32_t *cur_lod, uint32_t *visible, uint32_t *count, int32_t *draw_lod,
                                void *scratch)
{
    if (!e) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    int rc = clapgpu_visible_compact(stream, e->vis_mask, e->vis_row_pop, e->n, index_base, visible, count, scratch);
    if (rc) return rc;
    return clapgpu_entities_lod(stream, e, visible, count, index_base, cam_pos, force_lod, cur_lod, draw_lod);
}

/* the host-side view helpers (tests/c/test_scene.c builds its frustum with them): the oracle's */
void clapgpu_view_matrix(const float pos[3], const float quat[4], float view_mx[16]) { clapo_view_matrix(pos, quat, view_mx); }
void clapgpu_perspective(float fov, float aspect, float near_plane, float far_plane, int ndc_z_zero_one, float proj_mx[16])
{
    clapo_perspective(fov, aspect, near_plane, far_plane, ndc_z_zero_one, proj_mx);
}
void clapgpu_frustum_calc(const float view_mx[16], const float proj_mx[16], int ndc_z_zero_one, clapgpu_frustum *out)
{
    clapo_frustum_calc(view_mx, proj_mx, ndc_z_zero_one, (clapo_frustum *)out);
}
