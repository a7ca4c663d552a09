/*
 * tests/c/test_load.c -- the scene loader (include/clapgpu_load.h) from C.
 *
 *   test_load <fixture dir> <scratch dir>
 *
 * 1. loads tests/golden/scene_fixture/scene.json (+ hero.glb, crate.gltf, lamp.gltf) into a snapshot and reads it back;
 * 2. damaged inputs: every truncation of the GLB at 97-byte steps and 400 bit-flipped copies must be refused or
 *    loaded, never crash (the sanitizer build of tests/test_load_scene.py runs exactly this, -DTEST_LOAD_NO_GPU);
 * 3. (GPU build) replays the loaded scene like the engine's frame would: entities through the C host mirror
 *    (clapgpu_scene_*) against the oracle's entity update, bit for bit; the two characters' pose and skinning through
 *    the flat ABI (clapgpu_pose_update, clapgpu_skin) against the oracle, value for value.
 * Built and run by tests/test_load_scene.py.  Exit code 0 and "PASS" = pass.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "clapgpu.h"
#include "clapgpu_load.h"
#include "clapgpu_scene.h"
#include "clapgpu_snapshot.h"
#ifndef TEST_LOAD_NO_GPU
#include "clap_oracle.h"
#else
/* The sanitizer build checks memory safety of the parser, not values, and must not pull the HIP runtime into an
 * AddressSanitizer process: the three arithmetic helpers the loader takes from the libraries are given here. */
void clapgpu_mat4_invert(const float m[16], float out[16]) { memcpy(out, m, 64); }
void clapgpu_mat4_from_quat(const float q[4], float out[16]) { memset(out, 0, 64); out[0] = out[5] = out[10] = out[15] = q[3]; }
void clapgpu_quat_from_angles(const float a[3], int degrees, float q[4]) { q[0] = a[0]; q[1] = a[1]; q[2] = a[2]; q[3] = (float)degrees; }
#endif

static int fails;
#define CHECK(c, ...) do { if (!(c)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); fails++; } } while (0)

static uint8_t *slurp(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    *n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *b = malloc(*n ? *n : 1);
    if (fread(b, 1, *n, f) != *n) { fclose(f); free(b); return NULL; }
    fclose(f);
    return b;
}

static void spit(const char *path, const void *b, size_t n)
{
    FILE *f = fopen(path, "wb");
    fwrite(b, 1, n, f);
    fclose(f);
}

static const void *arr(clapgpu_snapshot *s, const char *name, uint32_t dtype, uint64_t count)
{
    clapgpu_snapshot_array a;
    if (clapgpu_snapshot_find(s, name, &a)) { CHECK(0, "snapshot has no '%s'", name); return NULL; }
    CHECK(a.dtype == dtype, "'%s' dtype %u != %u", name, a.dtype, dtype);
    if (count != UINT64_MAX) CHECK(a.count == count, "'%s' holds %llu elements, expected %llu", name, (unsigned long long)a.count, (unsigned long long)count);
    return a.data;
}

static int64_t scalar(clapgpu_snapshot *s, const char *name)
{
    const int64_t *p = arr(s, name, CLAPGPU_DT_I64, 1);
    return p ? *p : -1;
}

#ifndef TEST_LOAD_NO_GPU
static int replay(clapgpu_snapshot *s)
{
    const uint32_t n = (uint32_t)scalar(s, "entities.n"), nm = (uint32_t)scalar(s, "scene.n_models");
    const float *ps = arr(s, "entities.pos_scale", CLAPGPU_DT_F32, 4ull * n), *rot = arr(s, "entities.rot", CLAPGPU_DT_F32, 4ull * n);
    const int32_t *parent = arr(s, "entities.parent", CLAPGPU_DT_I32, n), *model = arr(s, "entities.model", CLAPGPU_DT_I32, n);
    const int32_t *pj = arr(s, "entities.parent_joint", CLAPGPU_DT_I32, n);
    const uint32_t *flags = arr(s, "entities.flags", CLAPGPU_DT_U32, n);
    const float *maabb = arr(s, "entities.model_aabb", CLAPGPU_DT_F32, 6ull * nm);
    const uint8_t *mskip = arr(s, "entities.model_skip", CLAPGPU_DT_U8, nm);
    if (fails) return 1;

    /* ---- entities through the host mirror, in list order (parents were created before their children) ---- */
    clapgpu_scene *sc = NULL;
    CHECK(!clapgpu_scene_create(&sc, 0), "clapgpu_scene_create: %s", clapgpu_last_error());
    if (!sc) return 1;
    uint32_t *mh = calloc(nm, 4), *eh = calloc(n, 4);
    for (uint32_t k = 0; k < nm; k++) CHECK(!clapgpu_scene_model_new(sc, maabb + 6 * k, mskip[k], &mh[k]), "model_new");
    for (uint32_t i = 0; i < n; i++) {
        CHECK(!clapgpu_scene_entity_new(sc, mh[model[i]], NULL, &eh[i]), "entity_new");
        /* a joint attachment rides the palette, which this replay computes afterwards: such entities are checked as plain children */
        if (parent[i] >= 0) CHECK(!clapgpu_scene_entity_set_parent(sc, eh[i], eh[parent[i]]), "set_parent");
        CHECK(!clapgpu_scene_entity_transform(sc, eh[i], ps + 4 * i, rot + 4 * i, ps[4 * i + 3]), "transform");
        const uint32_t want = flags[i] & 0xffffu, have = 1u;            /* entity_new: VISIBLE */
        CHECK(!clapgpu_scene_entity_flags(sc, eh[i], want & ~have, have & ~want), "flags");
    }
    clapgpu_frustum fr;
    {
        const float cam[3] = { 0, 5, 40 }, q[4] = { 0, 0, 0, 1 };
        float view[16], proj[16];
        clapgpu_view_matrix(cam, q, view);
        clapgpu_perspective(1.0f, 16.0f / 9.0f, 0.1f, 500.0f, 0, proj);
        clapgpu_frustum_calc(view, proj, 0, &fr);
    }
    CHECK(!clapgpu_scene_mq_update(sc, &fr), "mq_update: %s", clapgpu_last_error());

    float *o_mx = calloc(n, 64), *o_inv = calloc(n, 64), *o_aabb = calloc(n, 24), *o_ctr = calloc(n, 12);
    uint32_t *o_flags = calloc(n, 4), *o_seqs = calloc(n, 4);
    for (uint32_t i = 0; i < n; i++) o_flags[i] = (flags[i] & (0xffffu | CLAPGPU_E_ALIVE)) | CLAPGPU_E_DIRTY;
    clapo_entities_update(n, ps, rot, parent, model, maabb, mskip, o_flags, o_seqs, o_mx, o_inv, o_aabb, o_ctr);
    uint32_t bad = 0;
    for (uint32_t i = 0; i < n; i++) {
        bad += memcmp(clapgpu_scene_entity_mx(sc, eh[i]), o_mx + 16 * i, 64) != 0;
        bad += memcmp(clapgpu_scene_entity_inverse_mx(sc, eh[i]), o_inv + 16 * i, 64) != 0;
        bad += memcmp(clapgpu_scene_entity_aabb(sc, eh[i]), o_aabb + 6 * i, 24) != 0;
        bad += memcmp(clapgpu_scene_entity_aabb_center(sc, eh[i]), o_ctr + 3 * i, 12) != 0;
    }
    CHECK(!bad, "%u entity results differ from the oracle's bits", bad);
    (void)pj;

    /* ---- the characters of model 0: pose (both animations) and skinning through the flat ABI ---- */
    const uint32_t J = (uint32_t)scalar(s, "model0.nr_joints"), V = (uint32_t)scalar(s, "model0.n_verts");
    const uint32_t n_anims = (uint32_t)scalar(s, "model0.n_anims");
    const int32_t *jparent = arr(s, "model0.joint_parent", CLAPGPU_DT_I32, J);
    const float *invmx = arr(s, "model0.invmx", CLAPGPU_DT_F32, 16ull * J), *bind = arr(s, "model0.bind", CLAPGPU_DT_F32, 16ull * J);
    const float *root_pose = arr(s, "model0.root_pose", CLAPGPU_DT_F32, 16);
    const float *vpos = arr(s, "model0.position", CLAPGPU_DT_F32, 3ull * V), *vnor = arr(s, "model0.normal", CLAPGPU_DT_F32, 3ull * V);
    const uint8_t *vj = arr(s, "model0.joints", CLAPGPU_DT_U8, 4ull * V);
    const float *vw = arr(s, "model0.weights", CLAPGPU_DT_F32, 4ull * V);
    clapgpu_snapshot_array ce;
    CHECK(!clapgpu_snapshot_find(s, "characters.entity", &ce), "characters.entity");
    const uint32_t nc = (uint32_t)ce.count, *cent = ce.data;
    if (fails) return 1;
    /* joint depths / parents-first order of the joints reachable from joint 0 */
    int32_t *depth = malloc(4 * J), *order = malloc(4 * J);
    uint32_t n_order = 0, levels = 0;
    for (uint32_t j = 0; j < J; j++) depth[j] = j ? -1 : 0;
    for (int again = 1; again;) {
        again = 0;
        for (uint32_t j = 1; j < J; j++)
            if (depth[j] < 0 && jparent[j] >= 0 && depth[jparent[j]] >= 0) { depth[j] = depth[jparent[j]] + 1; again = 1; }
    }
    for (uint32_t d = 0; d <= J; d++) for (uint32_t j = 0; j < J; j++) if (depth[j] == (int32_t)d) { order[n_order++] = (int32_t)j; levels = d + 1; }
    /* pooled channel table of all animations (clapgpu_animations) */
    uint32_t *table = calloc((size_t)n_anims * J * 3 * 4, 4), t_total = 0, d_total = 0;
    clapo_animation *oan = calloc(n_anims, sizeof(*oan));
    float *times = NULL, *data = NULL;
    for (uint32_t a = 0; a < n_anims; a++) {
        char nm_[48];
        clapgpu_snapshot_array A[7];
        static const char *keys[7] = { "ch_target", "ch_path", "ch_nr", "ch_time_off", "ch_data_off", "times", "data" };
        for (int k = 0; k < 7; k++) { snprintf(nm_, sizeof(nm_), "model0.a%u_%s", a, keys[k]); CHECK(!clapgpu_snapshot_find(s, nm_, &A[k]), "%s", nm_); }
        if (fails) return 1;
        oan[a] = (clapo_animation){ (uint32_t)A[0].count, A[0].data, A[1].data, A[2].data, A[3].data, A[4].data, A[5].data, A[6].data };
        for (uint32_t c = 0; c < oan[a].n_channels; c++) {
            uint32_t *t = table + (((size_t)a * J + oan[a].ch_target[c]) * 3 + oan[a].ch_path[c]) * 4;
            t[0] = t_total + oan[a].ch_time_off[c]; t[1] = d_total + oan[a].ch_data_off[c]; t[2] = oan[a].ch_nr[c]; t[3] = 0;
        }
        times = realloc(times, 4 * (t_total + A[5].count + 1)); memcpy(times + t_total, A[5].data, 4 * A[5].count);
        data = realloc(data, 4 * (d_total + A[6].count + 1)); memcpy(data + d_total, A[6].data, 4 * A[6].count);
        t_total += (uint32_t)A[5].count; d_total += (uint32_t)A[6].count;
    }
    float *char_mx = malloc(64 * nc);
    for (uint32_t c = 0; c < nc; c++) memcpy(char_mx + 16 * c, o_mx + 16 * cent[c], 64);

    void *d_parent, *d_depth, *d_root, *d_inv, *d_bind, *d_table, *d_times, *d_data, *d_anim, *d_ft, *d_cmx, *d_trs, *d_jt, *d_jp;
    void *d_vpos, *d_vnor, *d_vj, *d_vw, *d_vf, *d_vc, *d_op, *d_on;
    uint32_t *anim_h = calloc(nc, 4), *vf = calloc(nc, 4), *vc = malloc(4 * nc);
    float *ft = calloc(nc, 4);
    for (uint32_t c = 0; c < nc; c++) { vc[c] = V; vf[c] = 0; }
#define UP(dev, host, bytes) do { const size_t nb__ = (size_t)(bytes) + 4; CHECK(!clapgpu_malloc(&dev, nb__), "malloc"); \
        if (host) CHECK(!clapgpu_memcpy_h2d(dev, host, (size_t)(bytes), NULL), "h2d"); else CHECK(!clapgpu_memset(dev, 0, nb__, NULL), "memset"); } while (0)
    UP(d_parent, jparent, 4 * J); UP(d_depth, depth, 4 * J); UP(d_root, root_pose, 64); UP(d_inv, invmx, 64 * J); UP(d_bind, bind, 64 * J);
    UP(d_table, table, (size_t)n_anims * J * 48); UP(d_times, times, 4 * t_total); UP(d_data, data, 4 * d_total);
    UP(d_anim, anim_h, 4 * nc); UP(d_ft, ft, 4 * nc); UP(d_cmx, char_mx, 64 * nc);
    UP(d_trs, NULL, (size_t)40 * J * nc); UP(d_jt, NULL, (size_t)64 * J * nc); UP(d_jp, NULL, (size_t)16 * J * nc);
    UP(d_vpos, vpos, 12 * V); UP(d_vnor, vnor, 12 * V); UP(d_vj, vj, 4 * V); UP(d_vw, vw, 16 * V); UP(d_vf, vf, 4 * nc); UP(d_vc, vc, 4 * nc);
    UP(d_op, NULL, (size_t)12 * V * nc); UP(d_on, NULL, (size_t)12 * V * nc);
    const clapgpu_skeleton sk = { J, levels, d_parent, d_depth, d_root, d_inv, d_bind };
    clapgpu_animations an = { n_anims, t_total, d_table, d_times, d_data, NULL, 0, 0, d_total, 0 };
    {                                                                   /* the key-major pools (required): once per model */
        uint32_t max_keys = 0, layout = 0;
        void *d_packed;
        for (size_t q = 0; q < (size_t)n_anims * J * 3; q++) if (table[4 * q + 2] > max_keys) max_keys = table[4 * q + 2];
        CHECK(!clapgpu_malloc(&d_packed, clapgpu_animations_packed_bytes(n_anims, max_keys, J)), "malloc packed");
        CHECK(!clapgpu_animations_pack(NULL, &an, J, max_keys, d_packed, &layout), "animations_pack: %s", clapgpu_last_error());
        an.packed = d_packed; an.packed_keys = max_keys; an.packed_layout = layout;
    }
    clapgpu_pose_batch pb = { nc, 0, d_anim, d_ft, NULL, d_cmx, d_trs, d_jt, d_jp };
    uint32_t *of = malloc(4 * nc);
    void *d_of;
    for (uint32_t c = 0; c < nc; c++) of[c] = c * V;
    UP(d_of, of, 4 * nc);
    const clapgpu_skin_batch sb2 = { nc, J, d_vf, d_vc, d_of, d_vpos, d_vnor, d_vj, d_vw, d_jt, d_op, d_on };

    const clapo_skeleton osk = { J, n_order, jparent, order, root_pose, invmx, bind };
    float *o_trs = calloc((size_t)nc * J * 10, 4), *o_gl = calloc((size_t)nc * J * 16, 4), *o_jt = calloc((size_t)nc * J * 16, 4), *o_jp = calloc((size_t)nc * J * 4, 4);
    int32_t *cursor = calloc((size_t)nc * J * 3, 4);
    float *g_jt = malloc((size_t)64 * J * nc), *g_op = malloc((size_t)12 * V * nc), *g_on = malloc((size_t)12 * V * nc);
    float *o_op = malloc((size_t)12 * V), *o_on = malloc((size_t)12 * V);
    static const float when[6] = { 0.0f, 0.4f, 1.25f, 0.1f, 1.9f, 2.6f };
    for (int f = 0; f < 6 && !fails; f++) {
        const uint32_t a = (uint32_t)f % n_anims;
        for (uint32_t c = 0; c < nc; c++) { anim_h[c] = a; ft[c] = when[f] * (c ? 0.5f : 1.0f) + 0.05f * c; }
        memset(cursor, 0, (size_t)nc * J * 3 * 4);                      /* every frame starts another animation: animation_start zeroes joint->off (model.c:1421) */
        CHECK(!clapgpu_memcpy_h2d(d_anim, anim_h, 4 * nc, NULL) && !clapgpu_memcpy_h2d(d_ft, ft, 4 * nc, NULL), "h2d");
        CHECK(!clapgpu_pose_update(NULL, &sk, &an, &pb), "pose_update: %s", clapgpu_last_error());
        CHECK(!clapgpu_skin(NULL, &sb2), "skin: %s", clapgpu_last_error());
        CHECK(!clapgpu_memcpy_d2h(g_jt, d_jt, (size_t)64 * J * nc, NULL) && !clapgpu_memcpy_d2h(g_op, d_op, (size_t)12 * V * nc, NULL) &&
              !clapgpu_memcpy_d2h(g_on, d_on, (size_t)12 * V * nc, NULL) && !clapgpu_stream_sync(NULL), "d2h");
        for (uint32_t c = 0; c < nc; c++) {
            clapo_pose_channels(&oan[a], ft[c], o_trs + (size_t)c * J * 10, cursor + (size_t)c * J * 3);
            clapo_pose_palette(&osk, o_trs + (size_t)c * J * 10, char_mx + 16 * c, o_gl + (size_t)c * J * 16, o_jt + (size_t)c * J * 16, o_jp + (size_t)c * J * 4);
            uint32_t differ = 0;                                        /* the same bits (the kernel's arithmetic is the reference's) */
            for (uint32_t k = 0; k < n_order; k++) for (int x = 0; x < 16; x++) {
                const size_t at = ((size_t)c * J + (size_t)order[k]) * 16 + x;
                differ += !!memcmp(&g_jt[at], &o_jt[at], 4);
            }
            CHECK(!differ, "frame %d character %u: %u palette floats differ from the oracle's", f, c, differ);
            clapo_skin(V, vpos, vnor, vj, vw, o_jt + (size_t)c * J * 16, o_op, o_on);
            uint32_t pd = 0, nd = 0;
            for (uint32_t x = 0; x < 3 * V; x++) {
                pd += !!memcmp(&g_op[(size_t)c * 3 * V + x], &o_op[x], 4);
                nd += !!memcmp(&g_on[(size_t)c * 3 * V + x], &o_on[x], 4);
            }
            CHECK(!pd && !nd, "frame %d character %u: %u skinned position / %u normal floats differ from the oracle's", f, c, pd, nd);
        }
    }
    clapgpu_scene_destroy(sc);
    return fails != 0;
}
#endif

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <fixture dir> <scratch dir>\n", argv[0]); return 2; }
    char scene[1024], out[1024], err[256], tmp[1024], tmp_scene[1024];
    snprintf(scene, sizeof(scene), "%s/scene.json", argv[1]);
    snprintf(out, sizeof(out), "%s/fixture.clps", argv[2]);
    int rc = clapgpu_load_scene(scene, NULL, out, err, sizeof(err));
    CHECK(!rc, "clapgpu_load_scene: %d %s", rc, err);
    clapgpu_snapshot *s = NULL;
    CHECK(!clapgpu_snapshot_open(&s, out), "reading the snapshot back");
    if (fails) return 1;
    CHECK(scalar(s, "entities.n") == 51 && scalar(s, "scene.n_models") == 3, "entity / model counts");
    CHECK(scalar(s, "model0.nr_joints") == 12 && scalar(s, "model0.n_anims") == 2 && scalar(s, "lights.nr_lights") == 5, "model0 / lights");

    /* one asset alone */
    snprintf(tmp, sizeof(tmp), "%s/hero.glb", argv[1]);
    snprintf(tmp_scene, sizeof(tmp_scene), "%s/hero.clps", argv[2]);
    CHECK(!clapgpu_load_gltf(tmp, 0, tmp_scene, err, sizeof(err)), "clapgpu_load_gltf: %s", err);

    /* damaged copies of the GLB under a one-model scene */
    size_t n = 0;
    uint8_t *glb = slurp(tmp, &n);
    CHECK(glb != NULL, "reading %s", tmp);
    snprintf(tmp_scene, sizeof(tmp_scene), "%s/one.json", argv[2]);
    static const char one[] = "{\"model\": [{\"name\": \"h\", \"gltf\": \"h.glb\", \"character\": [{\"position\": [0, 0, 0, 1, 30]}]}]}";
    spit(tmp_scene, one, sizeof(one) - 1);
    snprintf(tmp, sizeof(tmp), "%s/h.glb", argv[2]);
    snprintf(out, sizeof(out), "%s/damaged.clps", argv[2]);
    unsigned refused = 0, loaded = 0;
    for (size_t cut = 0; glb && cut < n; cut += 97) {
        spit(tmp, glb, cut);
        if (clapgpu_load_scene(tmp_scene, NULL, out, err, sizeof(err))) refused++; else loaded++;
    }
    CHECK(loaded == 0, "%u truncated files were accepted", loaded);
    uint64_t r = 88172645463325252ull;
    for (int k = 0; glb && k < 400; k++) {
        uint8_t *b = malloc(n);
        memcpy(b, glb, n);
        for (int j = 0; j < 3; j++) { r ^= r << 13; r ^= r >> 7; r ^= r << 17; b[r % n] ^= (uint8_t)(1u << ((r >> 40) & 7)); }
        spit(tmp, b, n);
        if (clapgpu_load_scene(tmp_scene, NULL, out, err, sizeof(err))) refused++; else loaded++;
        free(b);
    }
    spit(tmp, glb, n);
    CHECK(!clapgpu_load_scene(tmp_scene, NULL, out, err, sizeof(err)), "the intact copy: %s", err);
    free(glb);
    /* the same for the scene file itself and for the .gltf with its base64 buffer: truncations and byte flips of the
     * JSON text (the fixture's assets stay in place next to the damaged copy) */
    {
        char src[1024], dst[1024], cmd[4096];
        snprintf(cmd, sizeof(cmd), "cp %s/hero.glb %s/crate.gltf %s/lamp.gltf %s/ 2>/dev/null", argv[1], argv[1], argv[1], argv[2]);
        CHECK(system(cmd) == 0, "copying the assets");
        static const char *victims[2] = { "scene.json", "crate.gltf" };
        for (int v = 0; v < 2; v++) {
            size_t jn = 0;
            snprintf(src, sizeof(src), "%s/%s", argv[1], victims[v]);
            snprintf(dst, sizeof(dst), "%s/%s", argv[2], victims[v]);
            uint8_t *js = slurp(src, &jn);
            CHECK(js != NULL, "reading %s", src);
            snprintf(tmp_scene, sizeof(tmp_scene), "%s/scene.json", argv[2]);
            if (v == 1) { size_t sn = 0; uint8_t *sj = slurp(scene, &sn); spit(tmp_scene, sj, sn); free(sj); }
            for (size_t cut = 0; js && cut < jn; cut += 53) {
                spit(dst, js, cut);
                if (clapgpu_load_scene(tmp_scene, NULL, out, err, sizeof(err))) refused++; else loaded++;
            }
            for (int k = 0; js && k < 600; k++) {
                uint8_t *b = malloc(jn);
                memcpy(b, js, jn);
                for (int j = 0; j < 2; j++) { r ^= r << 13; r ^= r >> 7; r ^= r << 17; b[r % jn] = (uint8_t)(r >> 33); }
                spit(dst, b, jn);
                if (clapgpu_load_scene(tmp_scene, NULL, out, err, sizeof(err))) refused++; else loaded++;
                free(b);
            }
            spit(dst, js, jn);
            free(js);
        }
        snprintf(tmp_scene, sizeof(tmp_scene), "%s/scene.json", argv[2]);
        CHECK(!clapgpu_load_scene(tmp_scene, NULL, out, err, sizeof(err)), "the intact scene copy: %s", err);
    }
    printf("damaged inputs: %u refused, %u loaded\n", refused, loaded);
#ifndef TEST_LOAD_NO_GPU
    replay(s);
#endif
    clapgpu_snapshot_close(s);
    if (fails) return 1;
    printf("PASS\n");
    return 0;
}
