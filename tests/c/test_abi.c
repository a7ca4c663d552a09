/*
 * tests/c/test_abi.c -- a plain C caller of the flat C ABI (include/clapgpu.h), the way a binding inside
 * CLAP (C23) would use it: device buffers through clapgpu_malloc / clapgpu_memcpy_*, descriptors filled
 * by hand, results compared with the oracle library (oracle/clap_oracle.h) bit for bit.
 * Built and run by tests/test_scene_c.py (GPU).  Exit code 0 = pass.
 *
 * Covers: particles (advect + respawn with the libc drand48 stream), rigid bodies (integrate, both
 * broadphase passes, sphere contacts, body -> entity read-back), the light grid, the character feeder.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "clapgpu.h"
#include "clap_oracle.h"

static uint64_t rng_s = 0x243F6A8885A308D3ull;
static uint64_t rnd(void) { uint64_t z = (rng_s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static double drand(double a, double b) { return a + (b - a) * ((rnd() >> 11) * (1.0 / 9007199254740992.0)); }

#define CK(call) do { int rc__ = (call); if (rc__) { fprintf(stderr, "FAIL: %s -> %d (%s)\n", #call, rc__, clapgpu_last_error()); return 1; } } while (0)
static int fail(const char *what) { fprintf(stderr, "FAIL: %s\n", what); return 1; }

static void *dev_upload(const void *host, size_t bytes)
{
    void *d = NULL;
    if (clapgpu_malloc(&d, bytes ? bytes : 16)) return NULL;
    if (bytes && host) clapgpu_memcpy_h2d(d, host, bytes, NULL); else clapgpu_memset(d, 0, bytes ? bytes : 16, NULL);
    return d;
}
#define UP(ptr, count) dev_upload((ptr), sizeof(*(ptr)) * (size_t)(count))
#define ZERO(type, count) ((type *)dev_upload(NULL, sizeof(type) * (size_t)(count)))
#define DOWN(host, dev, count) clapgpu_memcpy_d2h((host), (dev), sizeof(*(host)) * (size_t)(count), NULL)

/* ------------------------------------------------------------------ particles */
static int test_particles(void)
{
    enum { NSYS = 5, COUNT = 300, PER = 320 /* systems start at multiples of 64 */, N = NSYS * PER, FRAMES = 6 };
    static clapo_particle_system ps[NSYS];
    static clapgpu_particle_system gps[NSYS];
    static float pos[N * 3], vel[N * 3], gpos[N * 3], gvel[N * 3], view[16];
    static uint32_t row_sys[N / 64];
    uint64_t state = 0x1234ABCD330Eull;
    for (int s = 0; s < NSYS; s++) {
        memset(&ps[s], 0, sizeof(ps[s]));
        ps[s].center[0] = (float)drand(-5, 5); ps[s].center[1] = (float)drand(-5, 5); ps[s].center[2] = (float)drand(-5, 5);
        ps[s].dist = (uint32_t)s % 4;
        ps[s].radius = 2.0 + s; ps[s].min_radius = s == 3 ? 1.0 : 0.0;
        ps[s].radius_squared = ps[s].radius * ps[s].radius;
        ps[s].velocity = 0.6;
        ps[s].first = (uint32_t)(s * PER); ps[s].count = COUNT;
        memcpy(&gps[s], &ps[s], sizeof(gps[s]));            /* same 64-byte layout */
        for (int r = 0; r < PER / 64; r++) row_sys[(s * PER) / 64 + r] = (uint32_t)s;
    }
    if (sizeof(clapo_particle_system) != sizeof(clapgpu_particle_system)) return fail("particle system layouts differ");
    memset(pos, 0, sizeof(pos)); memset(vel, 0, sizeof(vel));
    clapo_particles_spawn(ps, NSYS, pos, vel, &state);       /* particle_spawn for every system, in order */
    uint64_t st2[2];
    st2[0] = st2[1] = state;
    for (int i = 0; i < 16; i++) view[i] = (i % 5 == 0) ? 1.f : 0.1f * (float)i;

    clapgpu_particles gp;
    memset(&gp, 0, sizeof(gp));
    gp.n = N; gp.n_sys = NSYS;
    gp.sys = UP(gps, NSYS); gp.row_sys = UP(row_sys, N / 64);
    gp.pos = UP(pos, N * 3); gp.vel = UP(vel, N * 3);
    gp.rng_state = UP(st2, 2);
    gp.billboard_mx = ZERO(float, NSYS * 16);
    gp.respawn_mask = ZERO(uint64_t, N / 64);
    gp.respawn_row_pop = ZERO(uint8_t, 64);
    gp.respawn_list = ZERO(uint32_t, N);
    gp.respawn_count = ZERO(uint32_t, 1);
    gp.scratch = dev_upload(NULL, clapgpu_visible_scratch_bytes(N));
    gp.respawn_groups = ZERO(uint32_t, CLAPGPU_RESPAWN_GROUP_WORDS);
    uint32_t respawned_total = 0;
    for (int f = 0; f < FRAMES; f++) {
        uint32_t k = clapo_particles_update(ps, NSYS, pos, vel, &state);
        respawned_total += k;
        CK(clapgpu_particles_update(NULL, &gp, view));
        uint32_t gk = 0;
        DOWN(gpos, gp.pos, N * 3); DOWN(gvel, gp.vel, N * 3); DOWN(&gk, gp.respawn_count, 1); DOWN(st2, gp.rng_state, 2);
        CK(clapgpu_stream_sync(NULL));
        for (int s = 0; s < NSYS; s++) {
            size_t o = 3 * (size_t)ps[s].first, b = 12 * (size_t)COUNT;
            if (memcmp(gpos + o, pos + o, b) || memcmp(gvel + o, vel + o, b)) return fail("particle pos_array / velocity");
        }
        if (gk != k || st2[1] != state) return fail("respawn count / drand48 state");
    }
    if (!respawned_total) return fail("particle fixture never respawns");
    printf("particles ok (%u respawns)\n", respawned_total);
    return 0;
}

/* ------------------------------------------------------------------ bodies, broadphase, contacts */
static int test_bodies(void)
{
    enum { N = 3000, NS = 9, CAP = 64 * N, STEPS = 3 };
    static double pos[N * 3], quat[N * 4], lvel[N * 3], avel[N * 3], mass[N], radius[N], yoff[N], adt[N], statics[NS * 6];
    static uint32_t bflags[N], eflags[N], geflags[N];
    static int32_t adis[N], body_entity[N];
    static float ps[N * 4], rot[N * 4], gps[N * 4], grot[N * 4];
    static uint32_t pairs[2 * CAP], gpairs[2 * CAP], spairs[2 * CAP], gspairs[2 * CAP];
    static clapo_contact contacts[CAP];
    static clapgpu_contact gcontacts[CAP];
    for (int i = 0; i < N; i++) {
        for (int a = 0; a < 3; a++) { pos[3 * i + a] = drand(0, 14); lvel[3 * i + a] = drand(-1, 1); avel[3 * i + a] = drand(-.5, .5); }
        double q[4] = { drand(-1, 1), drand(-1, 1), drand(-1, 1), drand(-1, 1) }, l = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
        for (int a = 0; a < 4; a++) quat[4 * i + a] = q[a] / l;
        radius[i] = drand(0.1, 0.5); mass[i] = 4.0 / 3.0 * M_PI * radius[i] * radius[i] * radius[i]; yoff[i] = radius[i];
        bflags[i] = CLAPGPU_BODY_AUTO_DISABLE; adis[i] = 30; adt[i] = 0; body_entity[i] = i;
        ps[4 * i + 3] = 1.f; rot[4 * i + 3] = 1.f;
    }
    for (int s = 0; s < NS; s++)
        for (int a = 0; a < 3; a++) { double lo = drand(0, 12); statics[6 * s + 2 * a] = lo; statics[6 * s + 2 * a + 1] = lo + drand(0.5, 4); }
    if (sizeof(clapo_contact) != sizeof(clapgpu_contact)) return fail("contact layouts differ");

    clapgpu_bodies gb;
    memset(&gb, 0, sizeof(gb));
    gb.n = N;
    gb.pos = UP(pos, N * 3); gb.quat = UP(quat, N * 4); gb.lvel = UP(lvel, N * 3); gb.avel = UP(avel, N * 3);
    gb.mass = UP(mass, N); gb.radius = UP(radius, N); gb.yoffset = UP(yoff, N); gb.bflags = UP(bflags, N);
    gb.adis_steps_left = UP(adis, N); gb.adis_time_left = UP(adt, N); gb.body_entity = UP(body_entity, N);
    static double aabb[N * 6], axis[N * 3];
    gb.aabb = (double *)dev_upload(NULL, sizeof(aabb)); gb.axis = (double *)dev_upload(NULL, sizeof(axis));
    clapgpu_geom_offset_rotation(gb.geom_offset_R);
    clapo_bodies ob_;                                       /* the oracle's view of the same bodies (host arrays) */
    memset(&ob_, 0, sizeof(ob_));
    ob_.n = N; ob_.pos = pos; ob_.quat = quat; ob_.lvel = lvel; ob_.avel = avel; ob_.mass = mass; ob_.radius = radius;
    ob_.yoffset = yoff; ob_.bflags = bflags; ob_.adis_steps_left = adis; ob_.adis_time_left = adt; ob_.body_entity = body_entity;
    ob_.aabb = aabb; ob_.axis = axis;
    clapo_geom_offset_rotation(ob_.geom_offset_R);
    /* the oracle's struct is clapgpu_bodies without its last member, geom_records (a device-side cache the oracle has no use for) */
    if (sizeof(clapo_bodies) != offsetof(clapgpu_bodies, geom_records) || memcmp(ob_.geom_offset_R, gb.geom_offset_R, 96)) return fail("bodies layout / offset rotation");
    clapo_bodies_aabb(&ob_);
    CK(clapgpu_bodies_aabb(NULL, &gb));
    clapgpu_bp *bp = NULL;
    CK(clapgpu_bp_create(&bp, N, 1.0, NS, statics));
    uint32_t *d_pairs = ZERO(uint32_t, 2 * CAP), *d_spairs = ZERO(uint32_t, 2 * CAP), *d_tot = ZERO(uint32_t, 4);
    clapgpu_contact *d_contacts = (clapgpu_contact *)dev_upload(NULL, sizeof(clapgpu_contact) * CAP);
    float *d_ps = UP(ps, N * 4), *d_rot = UP(rot, N * 4);
    uint32_t *d_eflags = ZERO(uint32_t, N);
    clapgpu_world gw; clapo_world ow;
    clapgpu_world_defaults(&gw); clapo_world_defaults(&ow);
    if (sizeof(gw) != sizeof(ow) || memcmp(gw.gravity, ow.gravity, 24) || gw.linear_damping != ow.linear_damping ||
        gw.adis_linear_threshold_sq != ow.adis_linear_threshold_sq || gw.adis_angular_threshold_sq != ow.adis_angular_threshold_sq ||
        gw.adis_time != ow.adis_time || gw.adis_steps != ow.adis_steps)
        return fail("world defaults");

    double acc_g = 0, acc_o = 0;
    for (int f = 0; f < STEPS; f++) {
        int sg = clapgpu_phys_step_schedule(&acc_g, 1.0 / 60.0), so = clapo_phys_step_schedule(&acc_o, 1.0 / 60.0);
        if (sg != so || acc_g != acc_o) return fail("phys_step schedule");
        for (int s = 0; s < sg; s++) {
            uint64_t ns = clapo_broadphase_aabb_static_pairs(NS, statics, N, aabb, spairs, CAP);
            uint64_t np = clapo_broadphase_aabb_pairs(N, aabb, pairs, CAP);
            uint32_t nc = clapo_contacts_spheres((uint32_t)np, pairs, pos, radius, NULL, contacts);
            CK(clapgpu_bp_collide(NULL, bp, N, gb.aabb, d_pairs, CAP, d_tot, d_spairs, CAP, d_tot + 1));
            CK(clapgpu_contacts_spheres(NULL, &gb, d_pairs, d_tot, CAP, NULL, d_contacts, d_tot + 2));
            uint32_t tot[4];
            DOWN(tot, d_tot, 4); DOWN(gpairs, d_pairs, 2 * CAP); DOWN(gspairs, d_spairs, 2 * CAP); DOWN(gcontacts, d_contacts, CAP);
            CK(clapgpu_stream_sync(NULL));
            if (tot[0] != np || tot[1] != ns || tot[2] != nc) return fail("pair / contact totals");
            if (memcmp(gpairs, pairs, 8 * np) || memcmp(gspairs, spairs, 8 * ns)) return fail("pair lists");
            if (memcmp(gcontacts, contacts, sizeof(clapo_contact) * np)) return fail("contact records");
            if (np < N / 10 || nc == 0 || nc == np) return fail("broadphase fixture too sparse or too dense");
            clapo_bodies_step2(&ob_, &ow, 1.0 / 120.0);
            CK(clapgpu_bodies_step(NULL, &gb, &gw, 1.0 / 120.0));
        }
        clapo_phys_body_update(N, pos, quat, lvel, yoff, body_entity, ps, rot, eflags, NULL);
        CK(clapgpu_phys_body_update(NULL, &gb, N, d_ps, d_rot, d_eflags, NULL));
        static double gpos[N * 3], gquat[N * 4];
        DOWN(gpos, gb.pos, N * 3); DOWN(gquat, gb.quat, N * 4); DOWN(gps, d_ps, N * 4); DOWN(grot, d_rot, N * 4); DOWN(geflags, d_eflags, N);
        CK(clapgpu_stream_sync(NULL));
        if (memcmp(gpos, pos, sizeof(pos)) || memcmp(gquat, quat, sizeof(quat))) return fail("body state after the substeps");
        if (memcmp(gps, ps, sizeof(ps)) || memcmp(grot, rot, sizeof(rot)) || memcmp(geflags, eflags, sizeof(eflags)))
            return fail("entity TRS from phys_body_update");
    }
    uint32_t st = 99;
    CK(clapgpu_bp_status(NULL, bp, &st));
    if (st != 0) return fail("broadphase status: a body larger than the cell");
    clapgpu_bp_destroy(bp);
    printf("bodies ok\n");
    return 0;
}

/* ------------------------------------------------------------------ light grid */
static int test_lights(void)
{
    enum { NL = 40, W = 1600, H = 900, CELL = 32 };
    static float pos[CLAPGPU_LIGHTS_MAX * 3], color[CLAPGPU_LIGHTS_MAX * 3], att[CLAPGPU_LIGHTS_MAX * 3];
    static int32_t is_dir[CLAPGPU_LIGHTS_MAX];
    static uint32_t active[CLAPGPU_LIGHTS_MAX];
    float view[16], proj[16];
    const float cam_pos[3] = { 1, 2, 3 }, cam_quat[4] = { 0.05f, 0.1f, 0.f, 0.99373f };
    clapgpu_view_matrix(cam_pos, cam_quat, view);
    clapgpu_perspective(1.2217305f, 16.f / 9.f, 0.1f, 500.f, 0, proj);
    for (int i = 0; i < NL; i++) {
        pos[3 * i] = (float)drand(-40, 40); pos[3 * i + 1] = (float)drand(-10, 10); pos[3 * i + 2] = (float)drand(-90, 20);
        for (int a = 0; a < 3; a++) color[3 * i + a] = (float)drand(0.2, 3.0);
        att[3 * i] = 1.f; att[3 * i + 1] = (float)drand(0.02, 0.8); att[3 * i + 2] = (float)exp(drand(log(0.02), log(40.0)));
        is_dir[i] = i < 2; active[i] = (i % 7) != 3;
        if (is_dir[i]) { att[3 * i + 1] = att[3 * i + 2] = 0.f; }
    }
    uint32_t tw, th, otw, oth;
    clapgpu_light_grid_dims(W, H, CELL, &tw, &th);
    clapo_light_grid_dims(W, H, CELL, &otw, &oth);
    if (tw != otw || th != oth || tw != 50 || th != 29) return fail("light grid tile counts");
    uint32_t *tiles = calloc((size_t)tw * th * 4, 4), *gtiles = calloc((size_t)tw * th * 4, 4);
    clapo_light_grid_compute(NL, active, is_dir, pos, color, att, view, proj, W, H, CELL, tiles);
    clapgpu_lights gl;
    memset(&gl, 0, sizeof(gl));
    gl.nr_lights = NL;
    gl.pos = UP(pos, CLAPGPU_LIGHTS_MAX * 3); gl.color = UP(color, CLAPGPU_LIGHTS_MAX * 3);
    gl.attenuation = UP(att, CLAPGPU_LIGHTS_MAX * 3); gl.is_dir = UP(is_dir, CLAPGPU_LIGHTS_MAX); gl.active = UP(active, CLAPGPU_LIGHTS_MAX);
    uint32_t *d_tiles = ZERO(uint32_t, tw * th * 4);
    CK(clapgpu_light_grid_compute(NULL, &gl, view, proj, W, H, CELL, d_tiles));
    DOWN(gtiles, d_tiles, tw * th * 4);
    CK(clapgpu_stream_sync(NULL));
    if (memcmp(tiles, gtiles, (size_t)tw * th * 16)) return fail("light grid masks");
    uint32_t bits = 0;
    for (uint32_t i = 0; i < tw * th * 4; i++) bits += (uint32_t)__builtin_popcount(tiles[i]);
    if (bits < tw * th * 2 || bits > tw * th * 30) return fail("light grid fixture degenerate");
    printf("light grid ok (%u bits)\n", bits);
    return 0;
}

int main(void)
{
    if (clapgpu_init(0)) { fprintf(stderr, "FAIL: clapgpu_init: %s\n", clapgpu_last_error()); return 1; }
    if (test_particles() || test_bodies() || test_lights()) return 1;
    printf("PASS\n");
    return 0;
}
