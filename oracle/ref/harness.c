/*
 * oracle/ref/harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Drives the REFERENCE's own hot-path functions (compiled from the sources
 * where they lie under /root/reference; see Makefile) over flat arrays, so
 * tests can (1) pin oracle/'s restatement against the real thing and
 * (2) regenerate tests/golden/.  It is never linked into the product.
 *
 * The reference sources are #include'd by absolute path so that their
 * `static` hot-path functions (default_update, parent_transform_apply,
 * channels_transform, one_joint_transform, particles_update,
 * subview_calc_frustum) are callable.  Nothing is copied.
 *
 * Four test doubles are defined here, all for functions whose real definitions live in
 * translation units that cannot be built in this image:
 *   clap_get_current_time()    (clap.c, the frame driver): the engine's frame clock; the double
 *                              returns the time the job file gives for the frame being run
 *                              (animated_update / animation_start read it, model.c:1423,1565);
 *   texture_load()             (render-gl.c, needs GL headers): the sink light_grid_compute hands
 *                              its finished tile masks to (light.c:150-153).  The double records
 *                              format, size and the bytes handed over -- i.e. exactly what the
 *                              reference would upload -- and reports success;
 *   renderer_get_caps()        (render-common.c:65-68, needs GL headers): on this path exactly
 *                              one field of its result is read (ndc_z_zero_one, view.c:267),
 *                              which the job file supplies;
 *   clap_get_render_options()  (clap.c, the frame driver): default_update reads
 *                              overlay_draws_enabled (model.c:1719) when given a scene; the
 *                              double returns zeroed options (no debug overlay), which is what
 *                              a CONFIG_FINAL build runs with.  Only used when a job asks for
 *                              the camera bounding-volume pick (model.c:1703-1713).
 *
 * Usage: clap_ref <command> <in.clpio> <out.clpio>
 *   entities   default_update x frames, then view_entity_in_frustum
 *   pose       channels_transform + one_joint_transform per character
 *   particles  particles_update x frames (drand48 stream from a given state)
 *   bench_entities   time default_update + view_entity_in_frustum
 *   lod        entity3d_aabb_avg_edge + entity3d_set_lod
 *   lightgrid  light_grid_compute: lights x screen tiles -> RGBA32UI masks
 *   characters character_update (limbo teleport + history) for body-less characters
 *   transform  transform_set_angles / transform_move
 */
/* resolved through -I $(REF)/core (Makefile): /root/reference/core/{model,view,particle}.c */
#include "model.c"
#include "view.c"
#include "particle.c"
#include "light.c"
#include "character.c"

#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <time.h>

const char *build_date = "oracle";
const char *clap_version = "oracle";

static render_options harness_ropts;
render_options *clap_get_render_options(struct clap_context *ctx)
{
    return &harness_ropts;
}

static double harness_now;
double clap_get_current_time(struct clap_context *ctx)
{
    return harness_now;
}

static renderer_caps harness_caps;
const renderer_caps *renderer_get_caps(renderer_t *r)
{
    return &harness_caps;
}

static struct { texture_format format; unsigned int width, height; void *buf; int calls; } harness_tex;
cerr texture_load(texture_t *tex, texture_format format, unsigned int width, unsigned int height, void *buf)
{
    harness_tex.format = format;
    harness_tex.width = width;
    harness_tex.height = height;
    harness_tex.buf = buf;
    harness_tex.calls++;
    return CERR_OK;
}

/* ------------------------------------------------------------------ */
/* clpio: named flat arrays.  file = "CLPIO1\0\0" u64 count, records  */
/* record = char name[24], u64 nbytes, payload padded to 8 bytes      */
/* ------------------------------------------------------------------ */
#define MAX_ARR 64
struct arr { char name[24]; uint64_t nbytes; void *data; };
struct arrset { struct arr a[MAX_ARR]; int n; };

static void die(const char *msg, const char *arg)
{
    fprintf(stderr, "clap_ref: %s %s\n", msg, arg ? arg : "");
    exit(2);
}

static void clpio_read(const char *path, struct arrset *s)
{
    FILE *f = fopen(path, "rb");
    char magic[8];
    uint64_t count;

    if (!f) die("cannot open", path);
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "CLPIO1\0\0", 8)) die("bad magic", path);
    if (fread(&count, 8, 1, f) != 1 || count > MAX_ARR) die("bad count", path);
    s->n = (int)count;
    for (int i = 0; i < s->n; i++) {
        struct arr *a = &s->a[i];
        if (fread(a->name, 1, 24, f) != 24 || fread(&a->nbytes, 8, 1, f) != 1) die("short read", path);
        uint64_t padded = (a->nbytes + 7) & ~7ull;
        a->data = malloc(padded ? padded : 8);
        if (padded && fread(a->data, 1, padded, f) != padded) die("short payload", a->name);
    }
    fclose(f);
}

static void *arr_get(struct arrset *s, const char *name, uint64_t *nbytes)
{
    for (int i = 0; i < s->n; i++)
        if (!strncmp(s->a[i].name, name, 24)) {
            if (nbytes) *nbytes = s->a[i].nbytes;
            return s->a[i].data;
        }
    die("missing array", name);
    return NULL;
}

static int arr_has(struct arrset *s, const char *name)
{
    for (int i = 0; i < s->n; i++)
        if (!strncmp(s->a[i].name, name, 24))
            return 1;
    return 0;
}

static void *arr_add(struct arrset *s, const char *name, uint64_t nbytes)
{
    struct arr *a = &s->a[s->n++];
    if (s->n > MAX_ARR) die("too many outputs", name);
    memset(a->name, 0, 24);
    strncpy(a->name, name, 23);
    a->nbytes = nbytes;
    a->data = calloc(1, ((nbytes + 7) & ~7ull) + 8);
    return a->data;
}

static void clpio_write(const char *path, struct arrset *s)
{
    FILE *f = fopen(path, "wb");
    uint64_t count = s->n;

    if (!f) die("cannot write", path);
    fwrite("CLPIO1\0\0", 1, 8, f);
    fwrite(&count, 8, 1, f);
    for (int i = 0; i < s->n; i++) {
        struct arr *a = &s->a[i];
        fwrite(a->name, 1, 24, f);
        fwrite(&a->nbytes, 8, 1, f);
        fwrite(a->data, 1, (a->nbytes + 7) & ~7ull, f);
    }
    fclose(f);
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

/* ------------------------------------------------------------------ */
/* entity fixtures                                                     */
/* ------------------------------------------------------------------ */
struct ent_world {
    uint32_t    n, n_models;
    entity3d    *e;
    model3d     *m;
    model3dtx   *txm;
};

/* Mirrors what entity3d_make sets (model.c:1735-1744) without a renderer. */
static void world_make(struct ent_world *w, uint32_t n, uint32_t n_models,
                       const float *model_aabb, const uint8_t *model_skip,
                       const int32_t *model_of, const int32_t *parent, const uint32_t *flags)
{
    w->n = n;
    w->n_models = n_models;
    w->e = calloc(n, sizeof(entity3d));
    w->m = calloc(n_models, sizeof(model3d));
    w->txm = calloc(n_models, sizeof(model3dtx));
    for (uint32_t k = 0; k < n_models; k++) {
        memcpy(w->m[k].aabb, model_aabb + 6 * k, 6 * sizeof(float));
        w->m[k].skip_aabb = model_skip[k];
        w->txm[k].model = &w->m[k];
    }
    for (uint32_t i = 0; i < n; i++) {
        entity3d *e = &w->e[i];
        e->txmodel = &w->txm[model_of[i]];
        transform_init(&e->xform);
        e->scale = 1.0;
        e->parent_joint = JOINT_TYPE_MAX;
        e->light_idx = -1;
        e->flags = flags[i] & ~(1u << 16);   /* bit 16 is the shim's dirty mirror, not a reference flag */
        e->parent = parent[i] >= 0 ? &w->e[parent[i]] : NULL;
        e->update = default_update;
    }
}

static void view_setup(struct view *view, struct arrset *in, struct arrset *out)
{
    float *cam_pos = arr_get(in, "cam_pos", NULL);
    float *cam_quat = arr_get(in, "cam_quat", NULL);
    float *persp = arr_get(in, "persp", NULL);          /* fov, aspect, near, far */
    uint32_t ndcz01 = *(uint32_t *)arr_get(in, "ndc_z_zero_one", NULL);
    transform_t cam;

    memset(view, 0, sizeof(*view));
    harness_caps.ndc_z_zero_one = ndcz01;
    transform_init(&cam);
    transform_set_pos(&cam, cam_pos);
    transform_set_quat(&cam, cam_quat);

    transform_view_mat4x4(&cam, view->main.view_mx);                 /* view.c:165-169 */
    mat4x4_invert(view->main.inv_view_mx, view->main.view_mx);
    /* render-common.c:77-82 dispatch */
    if (ndcz01)
        mat4x4_perspective_ndc_z_1(view->main.proj_mx, persp[0], persp[1], persp[2], persp[3]);
    else
        mat4x4_perspective_ndc_z_2(view->main.proj_mx, persp[0], persp[1], persp[2], persp[3]);
    subview_calc_frustum(&view->main, NULL);                          /* view.c:248 */

    memcpy(arr_add(out, "view_mx", 64), view->main.view_mx, 64);
    memcpy(arr_add(out, "proj_mx", 64), view->main.proj_mx, 64);
    memcpy(arr_add(out, "planes", 96), view->main.frustum_planes, 96);
    memcpy(arr_add(out, "corners", 128), view->main.frustum_corners, 128);
}

static int cmd_entities(struct arrset *in, struct arrset *out)
{
    uint32_t n = *(uint32_t *)arr_get(in, "n", NULL);
    uint32_t frames = *(uint32_t *)arr_get(in, "frames", NULL);
    uint64_t mbytes;
    float *model_aabb = arr_get(in, "model_aabb", &mbytes);
    uint32_t n_models = mbytes / 24;
    uint8_t *model_skip = arr_get(in, "model_skip", NULL);
    int32_t *model_of = arr_get(in, "model", NULL);
    int32_t *parent = arr_get(in, "parent", NULL);
    uint32_t *flags = arr_get(in, "flags", NULL);
    float *pos_scale = arr_get(in, "pos_scale", NULL);     /* [frames][n][4] */
    float *rot = arr_get(in, "rot", NULL);                 /* [frames][n][4] */
    uint8_t *dirty = arr_get(in, "dirty", NULL);           /* [frames][n]    */
    struct ent_world w;
    struct view view;

    world_make(&w, n, n_models, model_aabb, model_skip, model_of, parent, flags);
    view_setup(&view, in, out);

    /* optional: joint attachments (model.c:1626-1641).  The parent of attach_entity[k] gets its own
     * model3d whose joint attach_joint[k] has bind = attach_bind[k], and joint_transforms[attach_joint[k]]
     * = attach_jt[k] -- exactly the two matrices parent_transform_apply reads. */
    if (arr_has(in, "attach_entity")) {
        uint64_t nb;
        uint32_t *a_ent = arr_get(in, "attach_entity", &nb);
        uint32_t na = nb / 4;
        int32_t *a_joint = arr_get(in, "attach_joint", NULL);
        float *a_jt = arr_get(in, "attach_jt", NULL), *a_bind = arr_get(in, "attach_bind", NULL);
        for (uint32_t k = 0; k < na; k++) {
            entity3d *e = &w.e[a_ent[k]], *par = e->parent;
            if (!par->joint_transforms) {
                model3d *pm = calloc(1, sizeof(*pm));
                model3dtx *ptx = calloc(1, sizeof(*ptx));
                *pm = *par->txmodel->model;
                pm->nr_joints = JOINTS_MAX;
                pm->joints = calloc(JOINTS_MAX, sizeof(struct model_joint));
                ptx->model = pm;
                par->txmodel = ptx;
                par->joint_transforms = calloc(JOINTS_MAX, sizeof(mat4x4));
            }
            memcpy(par->joint_transforms[a_joint[k]], a_jt + 16 * k, 64);
            memcpy(par->txmodel->model->joints[a_joint[k]].bind, a_bind + 16 * k, 64);
            e->parent_joint = a_joint[k];
        }
    }
    /* optional: camera bounding-volume pick needs a scene (model.c:1697-1713) */
    struct scene *scene = NULL;
    int32_t *o_bv = NULL;
    float *o_bv_vol = NULL;
    if (arr_has(in, "bv_cam_pos")) {
        scene = calloc(1, sizeof(*scene));
        scene->camera = &scene->cameras[0];
        transform_init(&scene->camera->xform);
        transform_set_pos(&scene->camera->xform, arr_get(in, "bv_cam_pos", NULL));
        int32_t ctl = *(int32_t *)arr_get(in, "bv_ctl", NULL);
        scene->control = ctl >= 0 ? &w.e[ctl] : NULL;
        o_bv = arr_add(out, "bv", (uint64_t)frames * 4);
        o_bv_vol = arr_add(out, "bv_volume", (uint64_t)frames * 4);
    }

    float *o_mx = arr_add(out, "mx", (uint64_t)frames * n * 64);
    float *o_inv = arr_add(out, "inv_mx", (uint64_t)frames * n * 64);
    float *o_aabb = arr_add(out, "aabb", (uint64_t)frames * n * 24);
    float *o_ctr = arr_add(out, "center", (uint64_t)frames * n * 12);
    uint32_t *o_seq = arr_add(out, "seqs", (uint64_t)frames * n * 4);
    uint8_t *o_vis = arr_add(out, "visible", (uint64_t)frames * n);

    for (uint32_t f = 0; f < frames; f++) {
        for (uint32_t i = 0; i < n; i++) {
            size_t k = (size_t)f * n + i;
            entity3d *e = &w.e[i];
            if (!dirty[k])
                continue;
            /* the public mutators a game would use (model.c:1810-1842, transform.c:35-79) */
            transform_set_pos(&e->xform, &pos_scale[4 * k]);
            transform_set_quat(&e->xform, &rot[4 * k]);
            e->scale = pos_scale[4 * k + 3];
        }
        /* mq_update (model.c:1953): ALIVE entities, list order == index order */
        if (scene)
            scene->camera->bv = NULL;                                  /* scene_camera_calc, scene.c:1018-1019 */
        for (uint32_t i = 0; i < n; i++)
            if (entity3d_matches(&w.e[i], ENTITY3D_ALIVE))
                entity3d_update(&w.e[i], scene);
        if (scene) {
            o_bv[f] = scene->camera->bv ? (int32_t)(scene->camera->bv - w.e) : -1;
            o_bv_vol[f] = scene->camera->bv ? scene->camera->bv_volume : 0.f;
        }
        for (uint32_t i = 0; i < n; i++) {
            size_t k = (size_t)f * n + i;
            entity3d *e = &w.e[i];
            memcpy(o_mx + 16 * k, e->mx, 64);
            memcpy(o_inv + 16 * k, e->inverse_mx, 64);
            memcpy(o_aabb + 6 * k, e->aabb, 24);
            memcpy(o_ctr + 3 * k, e->aabb_center, 12);
            o_seq[k] = (uint32_t)e->seq | ((uint32_t)e->parent_seq << 16);
            /* draw predicate of _models_render (model.c:959-973) */
            o_vis[k] = entity3d_matches(e, ENTITY3D_ALIVE) && entity3d_matches(e, ENTITY3D_VISIBLE) &&
                       (entity3d_matches(e, ENTITY3D_SKIP_CULLING) || view_entity_in_frustum(&view, e));
        }
    }
    return 0;
}

static int cmd_bench_entities(struct arrset *in, struct arrset *out)
{
    uint32_t n = *(uint32_t *)arr_get(in, "n", NULL);
    uint32_t reps = *(uint32_t *)arr_get(in, "reps", NULL);
    uint64_t mbytes;
    float *model_aabb = arr_get(in, "model_aabb", &mbytes);
    uint32_t n_models = mbytes / 24;
    struct ent_world w;
    struct view view;
    float *pos_scale = arr_get(in, "pos_scale", NULL);
    float *rot = arr_get(in, "rot", NULL);

    world_make(&w, n, n_models, model_aabb, arr_get(in, "model_skip", NULL), arr_get(in, "model", NULL),
               arr_get(in, "parent", NULL), arr_get(in, "flags", NULL));
    view_setup(&view, in, out);

    double best = 1e30, total = 0;
    uint64_t visible = 0;
    for (uint32_t r = 0; r < reps; r++) {
        for (uint32_t i = 0; i < n; i++) {       /* untimed: mark everything dirty */
            transform_set_pos(&w.e[i].xform, &pos_scale[4 * (size_t)i]);
            transform_set_quat(&w.e[i].xform, &rot[4 * (size_t)i]);
            w.e[i].scale = pos_scale[4 * (size_t)i + 3];
        }
        double t0 = now_s();
        for (uint32_t i = 0; i < n; i++)
            entity3d_update(&w.e[i], NULL);
        visible = 0;
        for (uint32_t i = 0; i < n; i++)
            visible += view_entity_in_frustum(&view, &w.e[i]);
        double dt = now_s() - t0;
        total += dt;
        if (dt < best) best = dt;
    }
    double *o = arr_add(out, "seconds", 24);
    o[0] = best; o[1] = total / reps; o[2] = (double)visible;
    return 0;
}

/* ------------------------------------------------------------------ */
/* particles: particle_spawn x count, then particles_update x frames   */
/* ------------------------------------------------------------------ */
static int cmd_particles(struct arrset *in, struct arrset *out)
{
    uint32_t n_sys = *(uint32_t *)arr_get(in, "n_sys", NULL);
    uint32_t frames = *(uint32_t *)arr_get(in, "frames", NULL);
    float *center = arr_get(in, "center", NULL);          /* [n_sys][3] */
    uint32_t *dist = arr_get(in, "dist", NULL);
    double *radius = arr_get(in, "radius", NULL);
    double *min_radius = arr_get(in, "min_radius", NULL);
    double *velocity = arr_get(in, "velocity", NULL);
    uint32_t *count = arr_get(in, "count", NULL);
    float *view_mx = arr_get(in, "view_mx", NULL);
    uint64_t state0 = *(uint64_t *)arr_get(in, "rng_state", NULL);
    struct scene *scene = calloc(1, sizeof(*scene));
    particle_system *ps = calloc(n_sys, sizeof(*ps));
    entity3d *ents = calloc(n_sys, sizeof(*ents));
    uint64_t total = 0;

    unsigned short seed16[3] = { state0 & 0xffff, (state0 >> 16) & 0xffff, (state0 >> 32) & 0xffff };
    seed48(seed16);                                        /* the libc stream drand48() draws from */

    scene->camera = &scene->cameras[0];
    memcpy(scene->camera->view.main.view_mx, view_mx, 64);

    for (uint32_t s = 0; s < n_sys; s++) {
        entity3d *e = &ents[s];
        transform_init(&e->xform);
        transform_set_pos(&e->xform, &center[3 * s]);
        e->flags = ENTITY3D_ALIVE | ENTITY3D_IS_PARTICLE | ENTITY3D_SKIP_CULLING;   /* particle.c:212-216 */
        e->priv = &ps[s];
        e->update = particles_update;
        ps[s].e = e;
        list_init(&ps[s].particles);
        ps[s].count = count[s];                            /* particle.c:219-225 */
        ps[s].radius = radius[s];
        ps[s].min_radius = min_radius[s];
        ps[s].radius_squared = radius[s] * radius[s];
        ps[s].velocity = velocity[s];
        ps[s].dist = dist[s];
        ps[s].pos_array = calloc(count[s] ? count[s] : 1, sizeof(vec3));
        total += count[s];
    }
    float *o_pos0 = arr_add(out, "pos0", total * 12);
    float *o_vel0 = arr_add(out, "vel0", total * 12);
    float *o_pos = arr_add(out, "pos", (uint64_t)frames * total * 12);
    float *o_vel = arr_add(out, "vel", (uint64_t)frames * total * 12);
    float *o_mx = arr_add(out, "mx", (uint64_t)frames * n_sys * 64);
    uint64_t *o_state = arr_add(out, "rng_state", (uint64_t)(frames + 1) * 8);

    uint64_t off = 0;
    for (uint32_t s = 0; s < n_sys; s++) {                 /* particle.c:229-234 */
        for (uint32_t i = 0; i < ps[s].count; i++) {
            particle_spawn(&ps[s]);
            particle *p = list_last_entry(&ps[s].particles, particle, entry);
            vec3_dup(ps[s].pos_array[i], p->pos);
            memcpy(o_pos0 + 3 * (off + i), p->pos, 12);
            memcpy(o_vel0 + 3 * (off + i), p->velocity, 12);
        }
        off += ps[s].count;
    }
    {
        unsigned short z[3] = { 0, 0, 0 }, *old = seed48(z);
        o_state[0] = (uint64_t)old[0] | ((uint64_t)old[1] << 16) | ((uint64_t)old[2] << 32);
        unsigned short back[3] = { old[0], old[1], old[2] };
        seed48(back);
    }
    for (uint32_t f = 0; f < frames; f++) {
        off = 0;
        for (uint32_t s = 0; s < n_sys; s++) {
            entity3d_update(&ents[s], scene);              /* mq_update order = system order */
            memcpy(o_pos + 3 * ((uint64_t)f * total + off), ps[s].pos_array, (uint64_t)ps[s].count * 12);
            memcpy(o_mx + 16 * ((uint64_t)f * n_sys + s), ents[s].mx, 64);
            particle *p;
            uint64_t i = 0;
            list_for_each_entry(p, &ps[s].particles, entry)
                memcpy(o_vel + 3 * ((uint64_t)f * total + off + i++), p->velocity, 12);
            off += ps[s].count;
        }
        unsigned short z[3] = { 0, 0, 0 }, *old = seed48(z);
        o_state[f + 1] = (uint64_t)old[0] | ((uint64_t)old[1] << 16) | ((uint64_t)old[2] << 32);
        unsigned short back[3] = { old[0], old[1], old[2] };
        seed48(back);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* pose: channels_transform + one_joint_transform per character        */
/* ------------------------------------------------------------------ */
static int cmd_pose(struct arrset *in, struct arrset *out)
{
    uint32_t J = *(uint32_t *)arr_get(in, "nr_joints", NULL);
    uint32_t n_chars = *(uint32_t *)arr_get(in, "n_chars", NULL);
    uint32_t frames = *(uint32_t *)arr_get(in, "frames", NULL);
    uint32_t n_ch = *(uint32_t *)arr_get(in, "n_channels", NULL);
    int32_t *parent = arr_get(in, "parent", NULL);
    float *invmx = arr_get(in, "invmx", NULL);
    float *root_pose = arr_get(in, "root_pose", NULL);
    uint32_t *ch_target = arr_get(in, "ch_target", NULL), *ch_path = arr_get(in, "ch_path", NULL);
    uint32_t *ch_nr = arr_get(in, "ch_nr", NULL);
    uint32_t *ch_time_off = arr_get(in, "ch_time_off", NULL), *ch_data_off = arr_get(in, "ch_data_off", NULL);
    float *times = arr_get(in, "times", NULL), *data = arr_get(in, "data", NULL);
    float *char_time = arr_get(in, "char_time", NULL);      /* [frames][n_chars] */
    float *char_mx = arr_get(in, "char_mx", NULL);          /* [n_chars][16] */
    float *trs0 = arr_get(in, "trs0", NULL);                /* [J][10] initial joint T,R,S */

    model3d *m = calloc(1, sizeof(*m));
    model3dtx *txm = calloc(1, sizeof(*txm));
    txm->model = m;
    darray_init(m->anis);
    model3d_add_skinning(m, J, (mat4x4 *)invmx);            /* model.c:524-538: copies invmx, bind = invert */
    memcpy(m->root_pose, root_pose, 64);
    for (uint32_t j = 0; j < J; j++)
        if (parent[j] >= 0) {
            int *c = darray_add(m->joints[parent[j]].children);
            *c = j;
        }
    struct animation *an = animation_new(m, "a", n_ch);
    for (uint32_t c = 0; c < n_ch; c++)
        animation_add_channel(an, ch_nr[c], times + ch_time_off[c], data + ch_data_off[c],
                              (ch_path[c] == PATH_ROTATION ? 4 : 3) * sizeof(float), ch_target[c], ch_path[c]);

    entity3d *ents = calloc(n_chars, sizeof(*ents));
    for (uint32_t i = 0; i < n_chars; i++) {
        entity3d *e = &ents[i];
        e->txmodel = txm;
        e->joints = calloc(J, sizeof(struct joint));        /* model.c:1751-1754 */
        e->joint_transforms = calloc(J, sizeof(mat4x4));
        memcpy(e->mx, char_mx + 16 * i, 64);
        for (uint32_t j = 0; j < J; j++) {
            memcpy(e->joints[j].translation, trs0 + 10 * j, 12);
            memcpy(e->joints[j].rotation, trs0 + 10 * j + 3, 16);
            memcpy(e->joints[j].scale, trs0 + 10 * j + 7, 12);
        }
    }
    uint64_t per = (uint64_t)n_chars * J;
    float *o_trs = arr_add(out, "trs", frames * per * 40);
    float *o_jt = arr_add(out, "joint_transforms", frames * per * 64);
    float *o_gl = arr_add(out, "global", frames * per * 64);
    float *o_pos = arr_add(out, "joint_pos", frames * per * 16);
    float *o_bind = arr_add(out, "bind", (uint64_t)J * 64);
    float *o_time_end = arr_add(out, "time_end", 4);
    for (uint32_t j = 0; j < J; j++)
        memcpy(o_bind + 16 * j, m->joints[j].bind, 64);
    *o_time_end = an->time_end;

    /* optional: drive the characters through animated_update (model.c:1563-1592) with a frame clock:
     * "now"[frames] doubles, per character a start time, a speed and whether the queued entry repeats */
    double *now = arr_has(in, "now") ? arr_get(in, "now", NULL) : NULL;
    struct scene *scene = NULL;
    double *o_ani = NULL;
    if (now) {
        double *start = arr_get(in, "start", NULL);
        float *speed = arr_get(in, "speed", NULL);
        uint8_t *repeat = arr_get(in, "repeat", NULL);
        scene = calloc(1, sizeof(*scene));
        scene->clap_ctx = (struct clap_context *)scene;     /* only handed to the clock double */
        for (uint32_t i = 0; i < n_chars; i++) {
            entity3d *e = &ents[i];
            darray_init(e->aniq);
            harness_now = start[i];
            if (!animation_push_by_name(e, scene, "a", true, repeat[i])) die("animation_push_by_name", NULL);
            ani_current(e)->speed = speed[i];               /* animation_set_speed's store (model.c:1517) */
        }
        o_ani = arr_add(out, "ani_time", (uint64_t)frames * n_chars * 8);
    }

    for (uint32_t f = 0; f < frames; f++)
        for (uint32_t i = 0; i < n_chars; i++) {
            entity3d *e = &ents[i];
            if (now) {
                harness_now = now[f];
                animated_update(e, scene);
                o_ani[(uint64_t)f * n_chars + i] = e->ani_time;
            } else {
                channels_transform(e, an, char_time[(uint64_t)f * n_chars + i]);   /* model.c:1582 */
                one_joint_transform(e, 0, -1);                                      /* model.c:1583 */
            }
            for (uint32_t j = 0; j < J; j++) {
                uint64_t k = (uint64_t)f * per + (uint64_t)i * J + j;
                memcpy(o_trs + 10 * k, e->joints[j].translation, 12);
                memcpy(o_trs + 10 * k + 3, e->joints[j].rotation, 16);
                memcpy(o_trs + 10 * k + 7, e->joints[j].scale, 12);
                memcpy(o_jt + 16 * k, e->joint_transforms[j], 64);
                memcpy(o_gl + 16 * k, e->joints[j].global, 64);
                memcpy(o_pos + 4 * k, e->joints[j].pos, 16);
            }
        }
    return 0;
}

/* ------------------------------------------------------------------ */
/* lod: the reference's building blocks of the per-pass LOD pick       */
/* ------------------------------------------------------------------ */
static int cmd_lod(struct arrset *in, struct arrset *out)
{
    uint32_t n = *(uint32_t *)arr_get(in, "n", NULL);
    float *model_aabb = arr_get(in, "model_aabb", NULL);   /* [n][6]: one model per probe */
    float *scale = arr_get(in, "scale", NULL);
    int32_t *lod_min = arr_get(in, "lod_min", NULL), *lod_max = arr_get(in, "lod_max", NULL);
    int32_t *request = arr_get(in, "request", NULL);
    float *o_edge = arr_add(out, "avg_edge", (uint64_t)n * 4);
    int32_t *o_lod = arr_add(out, "cur_lod", (uint64_t)n * 4);

    for (uint32_t i = 0; i < n; i++) {
        model3d m = {};
        model3dtx txm = { .model = &m };
        entity3d e = { .txmodel = &txm };
        memcpy(m.aabb, model_aabb + 6 * i, 24);
        m.lod_min = lod_min[i];
        m.lod_max = lod_max[i];
        m.nr_lods = lod_max[i] + 1;
        e.scale = scale[i];
        e.force_lod = -1;
        o_edge[i] = entity3d_aabb_avg_edge(&e);            /* model.c:1261-1264 */
        entity3d_set_lod(&e, request[i], false);           /* model.c:593-609 */
        o_lod[i] = e.cur_lod;
    }
    return 0;
}

/* transform_set_angles (transform.c:62-73) and transform_move (transform.c:41-47) on given inputs */
static int cmd_transform(struct arrset *in, struct arrset *out)
{
    uint32_t n = *(uint32_t *)arr_get(in, "n", NULL);
    float *angles = arr_get(in, "angles", NULL);           /* [n][3] */
    uint8_t *degrees = arr_get(in, "degrees", NULL);       /* [n] */
    float *pos = arr_get(in, "pos", NULL), *off = arr_get(in, "off", NULL);
    float *o_q = arr_add(out, "quat", (uint64_t)n * 16);
    float *o_p = arr_add(out, "pos", (uint64_t)n * 12);
    for (uint32_t i = 0; i < n; i++) {
        transform_t t;
        transform_init(&t);
        transform_set_pos(&t, pos + 3 * i);
        transform_move(&t, off + 3 * i);
        transform_set_angles(&t, angles + 3 * i, degrees[i]);
        memcpy(o_q + 4 * i, transform_rotation_quat(&t), 16);
        transform_pos(&t, o_p + 3 * i);
    }
    return 0;
}

/*
 * character_update (character.c:583-611) for characters WITHOUT a physics body: the limbo teleport
 * out of the position history (history_newest / history_fetch, character.c:558-581), then the
 * chained default_update.  (With a body the function also calls into ODE: not buildable here.)
 * Per frame the entity positions are set with entity3d_position, like game code moving a character.
 */
static int cmd_characters(struct arrset *in, struct arrset *out)
{
    uint32_t n = *(uint32_t *)arr_get(in, "n", NULL);
    uint32_t frames = *(uint32_t *)arr_get(in, "frames", NULL);
    float limbo = *(float *)arr_get(in, "limbo_height", NULL);
    float *pos = arr_get(in, "pos", NULL);                 /* [frames][n][3] */
    float *rot = arr_get(in, "rot", NULL);                 /* [n][4] */
    float *scale = arr_get(in, "scale", NULL);             /* [n] */
    float *hpos = arr_get(in, "hist_pos", NULL);           /* [n][POS_HISTORY_MAX][3] */
    uint32_t *hhead = arr_get(in, "hist_head", NULL);
    uint8_t *hwrapped = arr_get(in, "hist_wrapped", NULL);
    float aabb[6] = { -1, -2, -3, 1, 2, 3 };
    uint8_t skip = 0;
    int32_t *zero_i = calloc(n, 4), *no_parent = malloc(n * 4);
    uint32_t *flags = malloc(n * 4);
    struct ent_world w;
    struct character *ch = calloc(n, sizeof(*ch));
    struct scene *scene = calloc(1, sizeof(*scene));

    for (uint32_t i = 0; i < n; i++) { no_parent[i] = -1; flags[i] = ENTITY3D_ALIVE | ENTITY3D_VISIBLE; }
    world_make(&w, n, 1, aabb, &skip, zero_i, no_parent, flags);
    scene->camera = &scene->cameras[0];
    transform_init(&scene->camera->xform);
    transform_set_pos(&scene->camera->xform, (vec3){ 1e6f, 1e6f, 1e6f });
    scene->limbo_height = limbo;
    for (uint32_t i = 0; i < n; i++) {
        entity3d *e = &w.e[i];
        struct character *c = &ch[i];
        c->entity = e;
        entity3d_set(e, ENTITY3D_IS_CHARACTER, c);          /* character_make, character.c:621-625 */
        c->orig_update = e->update;
        e->update = character_update;
        memcpy(c->history.pos, hpos + (size_t)i * POS_HISTORY_MAX * 3, sizeof(c->history.pos));
        c->history.head = hhead[i];
        c->history.wrapped = hwrapped[i];
        e->scale = scale[i];
        transform_set_quat(&e->xform, rot + 4 * i);
    }
    float *o_pos = arr_add(out, "pos", (uint64_t)frames * n * 12);
    float *o_mx = arr_add(out, "mx", (uint64_t)frames * n * 64);
    uint32_t *o_head = arr_add(out, "hist_head", (uint64_t)frames * n * 4);
    uint8_t *o_wr = arr_add(out, "hist_wrapped", (uint64_t)frames * n);
    for (uint32_t f = 0; f < frames; f++)
        for (uint32_t i = 0; i < n; i++) {
            size_t k = (size_t)f * n + i;
            entity3d *e = &w.e[i];
            entity3d_position(e, pos + 3 * k);
            e->update(e, scene);                            /* character_update -> default_update */
            transform_pos(&e->xform, o_pos + 3 * k);
            memcpy(o_mx + 16 * k, e->mx, 64);
            o_head[k] = ch[i].history.head;
            o_wr[k] = ch[i].history.wrapped;
        }
    return 0;
}

/* light_grid_compute (light.c:88-154) over given light slots; the grid's tile array is allocated here
 * at the size light_grid_update (light.c:46-74) would give it, so that function returns early instead
 * of touching the (unbuildable) texture object. */
static int cmd_lightgrid(struct arrset *in, struct arrset *out)
{
    uint32_t nr_lights = *(uint32_t *)arr_get(in, "nr_lights", NULL);
    uint32_t *gridp = arr_get(in, "grid", NULL);           /* width, height, cell */
    uint32_t *active = arr_get(in, "active", NULL);        /* [nr_lights] 0/1 */
    int32_t *is_dir = arr_get(in, "is_dir", NULL);
    float *pos = arr_get(in, "pos", NULL), *color = arr_get(in, "color", NULL);
    float *att = arr_get(in, "attenuation", NULL);
    static struct light light;                              /* large (per-light views) */
    struct view view;
    struct arrset tmp = {};

    view_setup(&view, in, &tmp);
    memset(&light, 0, sizeof(light));
    bitmap_init(&light.active, LIGHTS_MAX);
    light.nr_lights = nr_lights;
    for (uint32_t i = 0; i < nr_lights; i++) {
        if (active[i]) bitmap_set(&light.active, i);
        light.is_dir[i] = is_dir[i];
        memcpy(&light.pos[3 * i], pos + 3 * i, 12);
        memcpy(&light.color[3 * i], color + 3 * i, 12);
        memcpy(&light.attenuation[3 * i], att + 3 * i, 12);
    }
    light.grid.width = gridp[0];
    light.grid.height = gridp[1];
    light.grid.cell = gridp[2];
    light.grid.twidth = (unsigned int)ceilf((float)gridp[0] / gridp[2]);     /* light.c:51-52 */
    light.grid.theight = (unsigned int)ceilf((float)gridp[1] / gridp[2]);
    uint64_t ntiles = (uint64_t)light.grid.twidth * light.grid.theight;
    light.grid.tiles = malloc(ntiles * sizeof(ui32vec4));
    memset(light.grid.tiles, 0xa5, ntiles * sizeof(ui32vec4));

    light_grid_compute(&light, &view);

    if (harness_tex.calls != 1 || harness_tex.format != TEX_FMT_RGBA32UI)
        die("light_grid_compute did not upload one RGBA32UI texture", NULL);
    uint32_t dims[2] = { harness_tex.width, harness_tex.height };
    memcpy(arr_add(out, "tile_dims", 8), dims, 8);
    memcpy(arr_add(out, "tiles", ntiles * 16), harness_tex.buf, ntiles * 16);
    float *o_rad = arr_add(out, "radius", (uint64_t)nr_lights * 4);
    for (uint32_t i = 0; i < nr_lights; i++)
        o_rad[i] = light_get_radius(&light, i);             /* light.c:301-309 */
    memcpy(arr_add(out, "view_mx", 64), view.main.view_mx, 64);
    memcpy(arr_add(out, "proj_mx", 64), view.main.proj_mx, 64);
    return 0;
}

int main(int argc, char **argv)
{
    struct arrset in = {}, out = {};
    int rc;

    if (argc != 4) die("usage: clap_ref <entities|bench_entities> <in> <out>", NULL);
    clpio_read(argv[2], &in);
    if (!strcmp(argv[1], "entities"))             rc = cmd_entities(&in, &out);
    else if (!strcmp(argv[1], "bench_entities"))  rc = cmd_bench_entities(&in, &out);
    else if (!strcmp(argv[1], "particles"))       rc = cmd_particles(&in, &out);
    else if (!strcmp(argv[1], "pose"))            rc = cmd_pose(&in, &out);
    else if (!strcmp(argv[1], "lod"))             rc = cmd_lod(&in, &out);
    else if (!strcmp(argv[1], "lightgrid"))       rc = cmd_lightgrid(&in, &out);
    else if (!strcmp(argv[1], "characters"))      rc = cmd_characters(&in, &out);
    else if (!strcmp(argv[1], "transform"))       rc = cmd_transform(&in, &out);
    else die("unknown command", argv[1]);
    clpio_write(argv[3], &out);
    return rc;
}
