/*
 * tests/c/test_scene.c -- drives the C host mirror (include/clapgpu_scene.h) the way CLAP's frame
 * loop would and checks every result against the oracle (oracle/clap_oracle.h), bit for bit.
 * Built and run by tests/test_scene_c.py (GPU).  Exit code 0 = pass.
 *
 * Scenario: entities are created in random order (children may precede their parents, as on the
 * engine's creation-ordered lists), then across frames some move, some are re-parented, some are
 * deleted and new ones appear.  After each clapgpu_scene_mq_update the oracle is run on a
 * parents-first ordering of the same scene; mx / inverse_mx / aabb / aabb_center / in-frustum
 * must match exactly.  argv[1] = "wide" makes one tree wider than 64 (level-major fallback).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "clapgpu_scene.h"
#include "clap_oracle.h"

#define MAXE 60000
static uint64_t rng_s = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) { uint64_t z = (rng_s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static float frand(float a, float b) { return a + (b - a) * (float)((rnd() >> 11) * (1.0 / 9007199254740992.0)); }

struct host_ent { uint32_t handle, parent /* index into ents[] or -1u */, model; float ps[4], rot[4]; uint32_t flags; int live;
                  int32_t force_lod, cur_lod; /* entity3d.force_lod / .cur_lod as the engine would hold them */ };
static struct host_ent ents[MAXE];
static uint32_t n_ents;
/* joint attachments (clapgpu_scene_entity_set_attach / clapgpu_scene_attached_update): entity i rides joint * bind */
static uint8_t att_on[MAXE];
static float att_jt[MAXE][16], att_bind[MAXE][16];
static uint32_t n_att_on;

static void rand_trs(struct host_ent *e, int child)
{
    float q[4] = { frand(-1, 1), frand(-1, 1), frand(-1, 1), frand(-1, 1) };
    float l = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) e->rot[i] = q[i] / l;
    float r = child ? 3.f : 200.f;
    e->ps[0] = frand(-r, r); e->ps[1] = frand(-r / 4, r / 4); e->ps[2] = frand(-r, r); e->ps[3] = frand(0.6f, 1.4f);
}

static int fail(const char *what, uint32_t i) { fprintf(stderr, "FAIL: %s (entity %u)\n", what, i); return 1; }

/* the frame's second view (the light's, which the shadow passes render with: pipeline-builder.c:34-46, model.c:752-760):
 * registered once with clapgpu_scene_set_views, culled by every mq_update's own launch */
static clapgpu_frustum light_fr;
static int light_on;
static uint32_t light_checks, light_drawn;

/* the oracle's view of the last checked frame (parents-first order), kept for check_lod() */
static uint32_t order[MAXE], o_n, o_vis[MAXE], o_nvis;
static float ps[MAXE * 4], aabb[MAXE * 6], ctr[MAXE * 3];
static int32_t model[MAXE];
static const float model_aabb[2][6] = { { -1, -2, -3, 1, 2, 3 }, { -0.5f, -0.5f, -0.5f, 2, 1, 0.5f } };
static const uint8_t model_lod[2][2] = { { 0, 3 }, { 1, 2 } };       /* model3d.lod_min / lod_max */

/* oracle on a parents-first ordering; compare with the scene mirror */
static int check_frame(clapgpu_scene *s, const clapgpu_frustum *fr, float *o_mx_keep /* [MAXE][16] persistent oracle state */,
                       float *o_inv, float *o_aabb, float *o_ctr, uint32_t *o_seqs, uint32_t *o_flags_dirty)
{
    static uint32_t pos_in_order[MAXE], depth[MAXE];
    uint32_t n = 0, maxd = 0;
    for (uint32_t i = 0; i < n_ents; i++) {
        if (!ents[i].live) continue;
        uint32_t d = 0, x = i;
        while (ents[x].parent != UINT32_MAX) { x = ents[x].parent; d++; }
        depth[i] = d; if (d > maxd) maxd = d;
    }
    for (uint32_t d = 0; d <= maxd; d++)
        for (uint32_t i = 0; i < n_ents; i++)
            if (ents[i].live && depth[i] == d) { pos_in_order[i] = n; order[n++] = i; }
    static float rot[MAXE * 4], mx[MAXE * 16], inv[MAXE * 16];
    static int32_t parent[MAXE];
    static uint32_t flags[MAXE], seqs[MAXE];
    static const uint8_t model_skip[2] = { 0, 0 };
    for (uint32_t k = 0; k < n; k++) {
        const struct host_ent *e = &ents[order[k]];
        memcpy(ps + 4 * k, e->ps, 16); memcpy(rot + 4 * k, e->rot, 16);
        parent[k] = e->parent == UINT32_MAX ? -1 : (int32_t)pos_in_order[e->parent];
        model[k] = (int32_t)e->model;
        flags[k] = e->flags | CLAPO_E_DIRTY;         /* converged result == everything rebuilt from current TRS */
        seqs[k] = 0;
    }
    if (n_att_on) {                                      /* model.c:1626-1641: mx = parent.mx * ((joint * bind) * local) */
        static clapo_attach tab[MAXE];
        static float jt_pool[MAXE * 16], bind_pool[MAXE * 16];
        uint32_t na = 0;
        for (uint32_t k = 0; k < n; k++) {               /* ascending k: the table is sorted by entity */
            const uint32_t i = order[k];
            if (!att_on[i] || ents[i].parent == UINT32_MAX) continue;
            flags[k] |= CLAPO_E_JOINT_ATTACHED;
            tab[na] = (clapo_attach){ .entity = k, .jt = na, .bind = na };
            memcpy(jt_pool + 16 * na, att_jt[i], 64); memcpy(bind_pool + 16 * na, att_bind[i], 64);
            na++;
        }
        clapo_entities_update_range(0, n, ps, rot, parent, model, &model_aabb[0][0], model_skip, flags, seqs, mx, inv, aabb, ctr,
                                    na, tab, jt_pool, bind_pool);
    } else
    clapo_entities_update(n, ps, rot, parent, model, &model_aabb[0][0], model_skip, flags, seqs, mx, inv, aabb, ctr);
    (void)o_mx_keep; (void)o_inv; (void)o_aabb; (void)o_ctr; (void)o_seqs; (void)o_flags_dirty;
    uint32_t vis_exp = 0;
    clapgpu_scene_arrays res;                            /* the bulk form a binding scatters from */
    static uint32_t last_layout = 0xffffffffu;
    if (clapgpu_scene_results(s, &res) || res.n_slots != clapgpu_scene_slot_count(s)) return fail("clapgpu_scene_results", 0);
    const uint32_t layout = clapgpu_scene_layout_generation(s);
    if (last_layout != 0xffffffffu && layout < last_layout) return fail("layout generation went backwards", layout);
    last_layout = layout;
    for (uint32_t k = 0; k < n; k++) {
        const struct host_ent *e = &ents[order[k]];
        const float *g;
        const uint32_t slot = clapgpu_scene_entity_slot(s, e->handle);
        if (slot >= res.n_slots) return fail("entity slot", order[k]);
        if (memcmp(res.mx + 16 * (size_t)slot, mx + 16 * k, 64) || memcmp(res.inverse_mx + 16 * (size_t)slot, inv + 16 * k, 64) ||
            memcmp(res.aabb + 6 * (size_t)slot, aabb + 6 * k, 24) || memcmp(res.aabb_center + 3 * (size_t)slot, ctr + 3 * k, 12))
            return fail("bulk result arrays", order[k]);
        if (!(g = clapgpu_scene_entity_mx(s, e->handle)) || memcmp(g, mx + 16 * k, 64)) return fail("mx", order[k]);
        if (!(g = clapgpu_scene_entity_inverse_mx(s, e->handle)) || memcmp(g, inv + 16 * k, 64)) return fail("inverse_mx", order[k]);
        if (!(g = clapgpu_scene_entity_aabb(s, e->handle)) || memcmp(g, aabb + 6 * k, 24)) return fail("aabb", order[k]);
        if (!(g = clapgpu_scene_entity_aabb_center(s, e->handle)) || memcmp(g, ctr + 3 * k, 12)) return fail("aabb_center", order[k]);
        int exp = (e->flags & CLAPO_E_VISIBLE) &&
                  ((e->flags & CLAPO_E_SKIP_CULLING) || clapo_aabb_in_frustum((const clapo_frustum *)fr, aabb + 6 * k));
        if (clapgpu_scene_entity_in_frustum(s, e->handle) != exp) return fail("view_entity_in_frustum", order[k]);
        if ((int)((res.vis_mask[slot >> 6] >> (slot & 63)) & 1) != exp) return fail("bulk visibility mask", order[k]);
        if (clapgpu_scene_entity_user(s, e->handle) != (void *)e) return fail("user pointer", order[k]);
        if (exp) o_vis[vis_exp] = k;
        vis_exp += exp;
    }
    o_n = n; o_nvis = vis_exp;
    if (clapgpu_scene_visible(s, NULL, 0) != vis_exp) return fail("visible count", vis_exp);
    if (light_on) {
        /* the light's view: its mask from the same launch, then the shadow pass's list (no camera: no LOD pick, model.c:974) */
        if (res.n_views != 1 || !res.view_mask[0]) return fail("extra view missing from the results", res.n_views);
        static uint8_t exp_l[MAXE];
        uint32_t n_l = 0;
        for (uint32_t k = 0; k < n; k++) {
            const struct host_ent *e = &ents[order[k]];
            const uint32_t slot = clapgpu_scene_entity_slot(s, e->handle);
            exp_l[k] = (e->flags & CLAPO_E_VISIBLE) &&
                       ((e->flags & CLAPO_E_SKIP_CULLING) || clapo_aabb_in_frustum((const clapo_frustum *)&light_fr, aabb + 6 * k));
            if ((int)((res.view_mask[0][slot >> 6] >> (slot & 63)) & 1) != exp_l[k]) return fail("light view mask", order[k]);
            n_l += exp_l[k];
        }
        uint32_t n_draw = 0;
        if (clapgpu_scene_select_lod_view(s, 0, NULL, &n_draw)) { fprintf(stderr, "select_lod_view: %s\n", clapgpu_last_error()); return 1; }
        if (n_draw != n_l) return fail("shadow pass list length", n_draw);
        const uint32_t *slots; const int32_t *lods;
        if (clapgpu_scene_draw_list(s, &slots, &lods) != n_draw) return fail("shadow pass list", 0);
        for (uint32_t k = 0; k < n_draw; k++) {
            if (k && slots[k] <= slots[k - 1]) return fail("shadow pass list not ascending", k);
            const struct host_ent *e = res.slot_user[slots[k]];
            if (!e || !e->live) return fail("shadow pass entry without an entity", k);
            uint32_t ko = 0;
            while (order[ko] != (uint32_t)(e - ents)) ko++;
            if (!exp_l[ko]) return fail("shadow pass draws what the light does not see", (uint32_t)(e - ents));
            if (lods[k] != clapgpu_scene_entity_cur_lod(s, e->handle)) return fail("shadow pass changed a LOD", (uint32_t)(e - ents));
        }
        light_checks++; light_drawn += n_l;
    }
    printf("  frame ok: %u live entities, %u visible, layout %s, %u slots\n", n, vis_exp,
           clapgpu_scene_layout_is_tiled(s) ? "tiles" : "levels", clapgpu_scene_slot_count(s));
    return 0;
}

/* One render pass of _models_render (model.c:959-992) through the mirror: clapgpu_scene_select_lod over the frame check_frame
 * just verified, against oracle/lod.c on the oracle's boxes -- the draw list as a set with its LODs, and every live
 * entity's cur_lod (kept inside the box, kept when culled, forced when forced). */
static int check_lod(clapgpu_scene *s, const float cam[3])
{
    static int32_t force[MAXE], cur[MAXE], draw[MAXE];
    static uint8_t seen[MAXE];
    for (uint32_t k = 0; k < o_n; k++) { force[k] = ents[order[k]].force_lod; cur[k] = ents[order[k]].cur_lod; }
    clapo_entities_lod(o_nvis, o_vis, cam, aabb, ctr, ps, model, &model_aabb[0][0], &model_lod[0][0], force, cur, draw);
    uint32_t n_draw = 0;
    if (clapgpu_scene_select_lod(s, cam, &n_draw)) { fprintf(stderr, "select_lod: %s\n", clapgpu_last_error()); return 1; }
    if (n_draw != o_nvis) return fail("draw list length", n_draw);
    const uint32_t *slots; const int32_t *lods;
    clapgpu_scene_arrays res;
    if (clapgpu_scene_draw_list(s, &slots, &lods) != n_draw || clapgpu_scene_results(s, &res)) return fail("draw list", 0);
    memset(seen, 0, sizeof(seen));
    for (uint32_t k = 0; k < n_draw; k++) {
        if (k && slots[k] <= slots[k - 1]) return fail("draw list not ascending", k);
        const struct host_ent *e = res.slot_user[slots[k]];
        if (!e || !e->live) return fail("draw list entry without an entity", k);
        seen[e - ents] = 1;
    }
    uint32_t distinct[8] = { 0 }, inside = 0;
    for (uint32_t v = 0; v < o_nvis; v++) {
        const uint32_t k = o_vis[v], i = order[k];
        if (!seen[i]) return fail("an entity the reference draws is not on the list", i);
        distinct[draw[v] & 7]++;
        const float *b = aabb + 6 * k;
        inside += cam[0] >= b[0] && cam[0] <= b[3] && cam[1] >= b[1] && cam[1] <= b[4] && cam[2] >= b[2] && cam[2] <= b[5];
    }
    for (uint32_t k = 0; k < n_draw; k++) {
        const struct host_ent *e = res.slot_user[slots[k]];
        uint32_t ko = 0;
        while (order[ko] != (uint32_t)(e - ents)) ko++;
        if (lods[k] != cur[ko]) return fail("draw LOD", (uint32_t)(e - ents));
    }
    for (uint32_t k = 0; k < o_n; k++) {
        struct host_ent *e = &ents[order[k]];
        if (clapgpu_scene_entity_cur_lod(s, e->handle) != cur[k]) return fail("cur_lod", order[k]);
        e->cur_lod = cur[k];                                 /* what the engine's entity3d would now hold */
    }
    uint32_t levels = 0;
    for (int q = 0; q < 8; q++) levels += distinct[q] > 0;
    printf("  lod pass ok: %u drawn, %u LOD levels, camera inside %u boxes\n", n_draw, levels, inside);
    return 0;
}

static int add_entity(clapgpu_scene *s, uint32_t parent_idx)
{
    struct host_ent *e = &ents[n_ents];
    memset(e, 0, sizeof(*e));
    e->model = (uint32_t)(rnd() & 1);
    e->parent = parent_idx;
    e->flags = CLAPO_E_ALIVE | CLAPO_E_VISIBLE;
    e->live = 1;
    e->force_lod = -1;                                   /* entity3d_make, model.c:1741 */
    rand_trs(e, parent_idx != UINT32_MAX);
    if (clapgpu_scene_entity_new(s, e->model, e, &e->handle)) return 1;
    if (n_ents & 1) {                                    /* the one-call form of the three pushes below */
        if (clapgpu_scene_entity_transform(s, e->handle, e->ps, e->rot, e->ps[3])) return 1;
    } else if (clapgpu_scene_entity_position(s, e->handle, e->ps) || clapgpu_scene_entity_rotation(s, e->handle, e->rot) ||
               clapgpu_scene_entity_scale(s, e->handle, e->ps[3])) return 1;
    n_ents++;
    return 0;
}

/* TEST_SCENE_PAR_FOR: the re-tile's passes as seven ranges taken BACKWARDS (clapgpu_scene_set_parallel_for): the layout, and
 * with it every frame below, may not depend on who takes which range in what order */
static void ranges_backwards(void (*range)(void *, uint32_t, uint32_t), void *ctx, uint32_t n, int threads)
{
    for (int t = threads - 1; t >= 0; t--)
        range(ctx, (uint32_t)((uint64_t)n * (uint32_t)t / (uint32_t)threads), (uint32_t)((uint64_t)n * ((uint32_t)t + 1) / (uint32_t)threads));
}

int main(int argc, char **argv)
{
    const int wide = argc > 1 && !strcmp(argv[1], "wide");
    /* optional: test_scene [wide|tiles] <seed> <frames of in-place edits>  (a soak: tests/test_scene_c.py runs the defaults) */
    if (argc > 2) rng_s ^= strtoull(argv[2], NULL, 0) * 0x9E3779B97F4A7C15ull;
    const int inplace_frames = argc > 3 ? atoi(argv[3]) : 4;
    clapgpu_scene *s;
    if (clapgpu_scene_create(&s, 0)) { fprintf(stderr, "create: %s\n", clapgpu_last_error()); return 2; }
    if (getenv("TEST_SCENE_PAR_FOR")) { clapgpu_scene_set_parallel_for(s, ranges_backwards, 7); printf("re-tile passes as 7 ranges, backwards\n"); }
    const float a0[6] = { -1, -2, -3, 1, 2, 3 }, a1[6] = { -0.5f, -0.5f, -0.5f, 2, 1, 0.5f };
    uint32_t m0, m1;
    if (clapgpu_scene_model_new(s, a0, 0, &m0) || clapgpu_scene_model_new(s, a1, 0, &m1) || m0 != 0 || m1 != 1) return 2;
    if (clapgpu_scene_model_lods(s, m0, model_lod[0][0], model_lod[0][1]) || clapgpu_scene_model_lods(s, m1, model_lod[1][0], model_lod[1][1])) return 2;

    /* camera: scene.c:74-76 defaults */
    float view[16], proj[16];
    const float cpos[3] = { 0, 10, 120 }, cq[4] = { 0, 0, 0, 1 };
    clapgpu_frustum fr;
    clapgpu_view_matrix(cpos, cq, view);
    clapgpu_perspective(70.f * 3.14159265f / 180.f, 16.f / 9.f, 0.1f, 500.f, 0, proj);
    clapgpu_frustum_calc(view, proj, 0, &fr);

    {   /* the light: above the scene, looking down and along +x, a narrower and shorter frustum than the camera's */
        float lview[16], lproj[16];
        const float lpos[3] = { -60, 90, 20 }, ang[3] = { -0.9f, -0.7f, 0.f };
        float lq[4];
        clapgpu_quat_from_angles(ang, 0, lq);
        clapgpu_view_matrix(lpos, lq, lview);
        clapgpu_perspective(50.f * 3.14159265f / 180.f, 1.f, 1.f, 260.f, 0, lproj);
        clapgpu_frustum_calc(lview, lproj, 0, &light_fr);
        if (clapgpu_scene_set_views(s, 1, &light_fr)) return 2;
        if (clapgpu_scene_set_views(s, CLAPGPU_EXTRA_VIEWS_MAX + 1, &light_fr) != CLAPGPU_ERR_INVALID_ARGUMENTS) return fail("too many views", 0);
        light_on = 1;
    }

    /* creation order: children first for a third of the scene (parents are attached afterwards) */
    for (int i = 0; i < 1500; i++) if (add_entity(s, UINT32_MAX)) return 2;
    for (int i = 0; i < 1500; i++) {
        uint32_t c = (uint32_t)(rnd() % n_ents), p = (uint32_t)(rnd() % n_ents);
        /* attach c under p if that keeps a forest of depth <= 7 and (unless "wide") narrow trees */
        uint32_t d = 0, x = p; int cyc = 0;
        while (ents[x].parent != UINT32_MAX) { x = ents[x].parent; if (x == c) cyc = 1; if (++d > 6) break; }
        if (cyc || x == c || d > 5 || ents[c].parent != UINT32_MAX || c == p) continue;
        int has_child = 0;
        for (uint32_t k = 0; k < n_ents; k++) if (ents[k].parent == c) has_child = 1;
        if (has_child && d > 2) continue;
        ents[c].parent = p;
        rand_trs(&ents[c], 1);
        clapgpu_scene_entity_position(s, ents[c].handle, ents[c].ps);
        clapgpu_scene_entity_rotation(s, ents[c].handle, ents[c].rot);
        clapgpu_scene_entity_scale(s, ents[c].handle, ents[c].ps[3]);
        if (clapgpu_scene_entity_set_parent(s, ents[c].handle, ents[p].handle)) return 2;
    }
    if (wide)                                           /* 200 children under one root: a level wider than a wavefront */
        for (int i = 0; i < 200; i++) if (add_entity(s, 0)) return 2; else
            if (clapgpu_scene_entity_set_parent(s, ents[n_ents - 1].handle, ents[0].handle)) return 2;
    if (clapgpu_scene_mq_update(s, &fr)) { fprintf(stderr, "mq_update: %s\n", clapgpu_last_error()); return 2; }
    if (clapgpu_scene_layout_is_tiled(s) == wide) return fail("layout choice", 0);
    if (check_frame(s, &fr, 0, 0, 0, 0, 0, 0)) return 1;

    for (int frame = 0; frame < 4; frame++) {
        for (int k = 0; k < 300; k++) {                  /* entity3d_position / rotate on a random subset */
            struct host_ent *e = &ents[rnd() % n_ents];
            if (!e->live) continue;
            rand_trs(e, e->parent != UINT32_MAX);
            clapgpu_scene_entity_position(s, e->handle, e->ps);
            clapgpu_scene_entity_rotation(s, e->handle, e->rot);
            clapgpu_scene_entity_scale(s, e->handle, e->ps[3]);
        }
        for (int k = 0; k < 80; k++) {                   /* entity3d_move / entity3d_rotate (euler, radians) */
            struct host_ent *e = &ents[rnd() % n_ents];
            if (!e->live) continue;
            const float off[3] = { (float)(rnd() % 200) * 0.01f - 1.f, 0.25f, (float)(rnd() % 200) * -0.01f };
            const float ang[3] = { (float)(rnd() % 1400) * 0.01f - 7.f, (float)(rnd() % 1400) * 0.01f - 7.f, 0.5f };
            for (int a = 0; a < 3; a++) e->ps[a] = e->ps[a] + off[a];
            clapgpu_quat_from_angles(ang, 0, e->rot);    /* the host copy follows the same (reference-pinned) helper */
            clapgpu_scene_entity_move(s, e->handle, off);
            clapgpu_scene_entity_rotate(s, e->handle, ang[0], ang[1], ang[2]);
        }
        for (int k = 0; k < 40; k++) {                   /* entity3d_visible(e, false) / SKIP_CULLING */
            struct host_ent *e = &ents[rnd() % n_ents];
            if (!e->live) continue;
            if (k & 1) { e->flags &= ~CLAPO_E_VISIBLE; clapgpu_scene_entity_visible(s, e->handle, 0); }
            else { e->flags |= CLAPO_E_SKIP_CULLING; clapgpu_scene_entity_flags(s, e->handle, CLAPGPU_E_SKIP_CULLING, 0); }
        }
        if (frame & 1) {                                 /* topology change: delete leaves, add new children */
            for (int k = 0; k < 60; k++) {
                uint32_t i = (uint32_t)(rnd() % n_ents);
                int has_child = 0;
                for (uint32_t c = 0; c < n_ents; c++) if (ents[c].live && ents[c].parent == i) has_child = 1;
                if (!ents[i].live || has_child || i == 0) continue;
                clapgpu_scene_entity_delete(s, ents[i].handle);
                ents[i].live = 0;
            }
            for (int k = 0; k < 80 && n_ents < MAXE; k++) {
                uint32_t p = (uint32_t)(rnd() % n_ents);
                uint32_t d = 0, x = p;
                while (ents[x].parent != UINT32_MAX) { x = ents[x].parent; d++; }
                if (!ents[p].live || d > 5) continue;
                if (add_entity(s, p) || clapgpu_scene_entity_set_parent(s, ents[n_ents - 1].handle, ents[p].handle)) return 2;
            }
        }
        for (int k = 0; k < 50; k++) {                   /* entity3d_set_lod(e, lod, true / false): forced, released, set */
            struct host_ent *e = &ents[rnd() % n_ents];
            if (!e->live) continue;
            if (k % 3 == 0) e->force_lod = -1; else if (k % 3 == 1) e->force_lod = (int32_t)(rnd() % 4);
            else e->cur_lod = model_lod[e->model][0];
            if (e->force_lod >= 0) e->cur_lod = e->force_lod;
            clapgpu_scene_entity_lod(s, e->handle, e->force_lod, e->cur_lod);
        }
        if (clapgpu_scene_mq_update(s, &fr)) { fprintf(stderr, "mq_update: %s\n", clapgpu_last_error()); return 2; }
        if (check_frame(s, &fr, 0, 0, 0, 0, 0, 0)) return 1;
        /* render passes: the frame's camera, then a camera sitting inside some drawn entity's box */
        if (check_lod(s, cpos)) return 1;
        if (o_nvis) {
            const float *c = ctr + 3 * o_vis[(frame * 7) % o_nvis];
            const float inside[3] = { c[0], c[1], c[2] };
            if (check_lod(s, inside)) return 1;
        }
    }
    /* ---- the standing layout edited in place (clapgpu_scene_entity_new_placed / _delete_placed): leaves go, roots and
     * children come, nothing else moves -- no re-tile in a frame whose edits all fitted; where one does not (or the layout is
     * not the one-launch tile form) the plain verbs take over and the frame re-tiles as before. */
    {
        clapgpu_scene_set_incremental(s, 1);
        uint32_t placed = 0, removed = 0, fell_back = 0, quiet_frames = 0;
        for (int frame = 0; frame < inplace_frames; frame++) {
            if (frame == 0) {                                /* one plain edit: the re-tile that leaves room */
                if (add_entity(s, UINT32_MAX)) return 2;
                if (clapgpu_scene_mq_update(s, &fr)) { fprintf(stderr, "mq_update: %s\n", clapgpu_last_error()); return 2; }
            }
            const uint32_t gen0 = clapgpu_scene_layout_generation(s);
            uint32_t fb = 0;
            for (int k = 0; k < 50; k++) {
                uint32_t i = (uint32_t)(rnd() % n_ents);
                int has_child = 0;
                for (uint32_t c = 0; c < n_ents; c++) if (ents[c].live && ents[c].parent == i) has_child = 1;
                if (!ents[i].live || has_child || i == 0) continue;
                const int rc = clapgpu_scene_entity_delete_placed(s, ents[i].handle);
                if (rc == CLAPGPU_ERR_NOT_SUPPORTED) { fb++; if (clapgpu_scene_entity_delete(s, ents[i].handle)) return 2; }
                else if (rc) return fail("delete_placed", i);
                else removed++;
                ents[i].live = 0;
            }
            uint32_t n_live_now = 0;
            for (uint32_t c = 0; c < n_ents; c++) n_live_now += ents[c].live;
            for (int k = 0; k < 90 && n_ents < MAXE && n_live_now + (uint32_t)k < 2600; k++) {   /* (a long run stays around a steady population) */
                uint32_t p = (k % 3 == 0) ? UINT32_MAX : (uint32_t)(rnd() % n_ents);
                if (p != UINT32_MAX) {
                    uint32_t d = 0, x = p;
                    while (ents[x].parent != UINT32_MAX) { x = ents[x].parent; d++; }
                    if (!ents[p].live || d > 5) continue;
                }
                struct host_ent *e = &ents[n_ents];
                memset(e, 0, sizeof(*e));
                e->model = (uint32_t)(rnd() & 1);
                e->parent = p;
                e->flags = CLAPO_E_ALIVE | CLAPO_E_VISIBLE;
                e->live = 1;
                e->force_lod = -1;
                rand_trs(e, p != UINT32_MAX);
                uint32_t slot = 0;
                const int rc = clapgpu_scene_entity_new_placed(s, e->model, e, p == UINT32_MAX ? CLAPGPU_NO_ENTITY : ents[p].handle, &e->handle, &slot);
                if (rc == CLAPGPU_ERR_NOT_SUPPORTED) {
                    fb++;
                    if (clapgpu_scene_entity_new(s, e->model, e, &e->handle)) return 2;
                    if (p != UINT32_MAX && clapgpu_scene_entity_set_parent(s, e->handle, ents[p].handle)) return 2;
                } else if (rc) return fail("new_placed", n_ents);
                else placed++;
                if (clapgpu_scene_entity_transform(s, e->handle, e->ps, e->rot, e->ps[3])) return 2;
                if (k % 7 == 0) { e->flags &= ~CLAPO_E_VISIBLE; clapgpu_scene_entity_visible(s, e->handle, 0); }
                n_ents++;
            }
            for (int k = 0; k < 200; k++) {
                struct host_ent *e = &ents[rnd() % n_ents];
                if (!e->live) continue;
                rand_trs(e, e->parent != UINT32_MAX);
                clapgpu_scene_entity_transform(s, e->handle, e->ps, e->rot, e->ps[3]);
            }
            if (clapgpu_scene_mq_update(s, &fr)) { fprintf(stderr, "mq_update: %s\n", clapgpu_last_error()); return 2; }
            if (!fb && clapgpu_scene_layout_generation(s) != gen0) return fail("a frame whose edits all fitted re-tiled", (uint32_t)frame);
            quiet_frames += !fb;
            fell_back += fb;
            if (check_frame(s, &fr, 0, 0, 0, 0, 0, 0)) return 1;
            if (check_lod(s, cpos)) return 1;
        }
        const int editable = clapgpu_scene_is_zero_copy(s) && clapgpu_scene_layout_is_tiled(s);
        if (editable && (placed < 100 || removed < 60 || !quiet_frames)) return fail("too few in-place edits in the scenario", placed);
        if (!editable && (placed || removed)) return fail("in-place edits on a layout that does not take them", placed);
        printf("  layout edited in place: %u entities placed, %u removed, %u edits fell back to a re-tile, %u of %d frames without one\n",
               placed, removed, fell_back, quiet_frames, inplace_frames);
    }
    /* ---- joint attachments: the frame's SECOND launch.  Some children ride "a joint of their parent": mq_update computes
     * everything else, attached_update (with the joints' matrices of the frame) the riders and everything below them. */
    {
        static uint32_t handles[MAXE];
        static float jt[MAXE * 16], bind[MAXE * 16];
        for (int frame = 0; frame < 3; frame++) {
            uint32_t na = 0;
            if (frame == 0) {
                for (uint32_t i = 0; i < n_ents && n_att_on < 60; i++) {
                    if (!ents[i].live || ents[i].parent == UINT32_MAX || att_on[ents[i].parent] || (rnd() % 5)) continue;
                    uint32_t x = ents[i].parent; int nested = 0;
                    while (x != UINT32_MAX) { if (att_on[x]) nested = 1; x = ents[x].parent; }
                    if (nested) continue;
                    att_on[i] = 1; n_att_on++;
                    if (clapgpu_scene_entity_set_attach(s, ents[i].handle, 1)) return fail("set_attach", i);
                }
            }
            for (uint32_t i = 0; i < n_ents; i++) {      /* this frame's joint matrices: a rigid-ish transform each */
                if (!att_on[i]) continue;
                for (int m = 0; m < 2; m++) {
                    float *M = m ? att_bind[i] : att_jt[i];
                    struct host_ent t; rand_trs(&t, 1);
                    const float x = t.rot[0], y = t.rot[1], z = t.rot[2], w = t.rot[3];
                    const float R[16] = { 1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w), 0,
                                          2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w), 0,
                                          2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y), 0,
                                          t.ps[0], t.ps[1], t.ps[2], 1 };
                    memcpy(M, R, 64);
                }
                handles[na] = ents[i].handle;
                memcpy(jt + 16 * na, att_jt[i], 64); memcpy(bind + 16 * na, att_bind[i], 64);
                na++;
            }
            for (int k = 0; k < 100; k++) {              /* and some entities move, riders and their parents among them */
                struct host_ent *e = &ents[rnd() % n_ents];
                if (!e->live) continue;
                rand_trs(e, e->parent != UINT32_MAX);
                clapgpu_scene_entity_transform(s, e->handle, e->ps, e->rot, e->ps[3]);
            }
            if (clapgpu_scene_mq_update(s, &fr)) { fprintf(stderr, "mq_update: %s\n", clapgpu_last_error()); return 2; }
            if (clapgpu_scene_attached_update(s, na, handles, jt, bind)) { fprintf(stderr, "attached_update: %s\n", clapgpu_last_error()); return 2; }
            if (check_frame(s, &fr, 0, 0, 0, 0, 0, 0)) return 1;
        }
        if (n_att_on < 20) return fail("too few riders in the scenario", n_att_on);
        printf("  %u joint riders through the second launch (%s)\n", n_att_on, clapgpu_scene_is_zero_copy(s) ? "small-frame path" : "staged path");
    }
    {   /* the light moved after the update (light_update runs behind mq_update, scene.c:1166-1171): its view alone again */
        clapgpu_frustum moved;
        float lview[16], lproj[16];
        const float lpos[3] = { 40, 70, -30 }, lq[4] = { 0, 0, 0, 1 };
        clapgpu_view_matrix(lpos, lq, lview);
        clapgpu_perspective(60.f * 3.14159265f / 180.f, 1.f, 1.f, 300.f, 0, lproj);
        clapgpu_frustum_calc(lview, lproj, 0, &moved);
        if (clapgpu_scene_cull_view(s, 0, &moved)) { fprintf(stderr, "cull_view: %s\n", clapgpu_last_error()); return 2; }
        if (clapgpu_scene_cull_view(s, 1, &moved) != CLAPGPU_ERR_INVALID_ARGUMENTS) return fail("cull_view of a view that is not registered", 1);
        clapgpu_scene_arrays res;
        if (clapgpu_scene_results(s, &res)) return fail("results after cull_view", 0);
        uint32_t differs = 0;
        for (uint32_t k = 0; k < o_n; k++) {
            const struct host_ent *e = &ents[order[k]];
            const uint32_t slot = clapgpu_scene_entity_slot(s, e->handle);
            const int exp = (e->flags & CLAPO_E_VISIBLE) &&
                            ((e->flags & CLAPO_E_SKIP_CULLING) || clapo_aabb_in_frustum((const clapo_frustum *)&moved, aabb + 6 * k));
            if ((int)((res.view_mask[0][slot >> 6] >> (slot & 63)) & 1) != exp) return fail("light view mask after cull_view", order[k]);
            differs += exp != (int)((res.vis_mask[slot >> 6] >> (slot & 63)) & 1);
        }
        if (!differs) return fail("the light sees exactly what the camera sees: the test shows nothing", 0);
        if (!light_checks || !light_drawn) return fail("no light view checked", light_checks);
        printf("  light view ok: %u frames, %u entities drawn by the shadow passes in all\n", light_checks, light_drawn);
        if (clapgpu_scene_set_views(s, 0, NULL)) return 2;       /* off again */
        if (clapgpu_scene_mq_update(s, &fr) || clapgpu_scene_results(s, &res) || res.n_views != 0) return fail("views not taken off", 0);
    }
    if (clapgpu_scene_entity_position(s, 0xdeadbeef, cpos) != CLAPGPU_ERR_INVALID_ARGUMENTS) return fail("bad handle", 0);
    clapgpu_scene_destroy(s);
    printf("PASS\n");
    return 0;
}
