/*
 * oracle/particles.c -- TEST INFRASTRUCTURE ONLY (see clap_oracle.h).
 *
 * Particle systems: spawn, per-frame advect / respawn, billboard matrix, restated
 * from the reference's core/particle.c.  drand48 is glibc's 48-bit LCG, threaded
 * through every system in update order exactly as the single global stream of the
 * reference is (particle.c:36-74 consume it from a list walk).
 */
#include "clap_oracle.h"
#include "lm.h"

/* glibc drand48: X' = (0x5DEECE66D X + 0xB) mod 2^48, value = X' / 2^48 (exact in double) */
#define R48_A 0x5DEECE66DULL
#define R48_C 0xBULL
#define R48_MASK ((1ULL << 48) - 1)

uint64_t clapo_srand48(int64_t seed)
{
    return (((uint64_t)seed << 16) | 0x330EULL) & R48_MASK;     /* srand48(3) */
}

double clapo_drand48(uint64_t *state)
{
    *state = (R48_A * *state + R48_C) & R48_MASK;
    return (double)*state * (1.0 / 281474976710656.0);          /* 2^-48: exact scaling */
}

/* particle.c:36-67 random_point_sphere + linmath.h:63-69 vec3_norm_safe */
static void random_point_sphere(float pos[3], const float center[3], double radius, double min_radius,
                                uint32_t dist, uint64_t *rng)
{
    float dir[3];
    dir[0] = clapo_drand48(rng) * 2.0 - 1.0;
    dir[1] = clapo_drand48(rng) * 2.0 - 1.0;
    dir[2] = clapo_drand48(rng) * 2.0 - 1.0;
    if (sqrtf(lm_dot3(dir, dir))) {
        float k = 1.0 / sqrtf(lm_dot3(dir, dir));               /* vec3_norm: double divide, float store */
        for (int i = 0; i < 3; i++) dir[i] = dir[i] * k;
    }

    double u;
    switch (dist) {
    case CLAPO_PART_DIST_POW075: u = pow(clapo_drand48(rng), 0.75); break;
    case CLAPO_PART_DIST_CBRT:   u = cbrt(clapo_drand48(rng)); break;
    case CLAPO_PART_DIST_SQRT:   u = sqrt(clapo_drand48(rng)); break;
    default:                     u = clapo_drand48(rng); break;
    }
    float r = (float)(min_radius + (radius - min_radius) * u);
    for (int i = 0; i < 3; i++)                                 /* vec3_add_scaled(pos, center, dir, 1.0, r) */
        pos[i] = center[i] * 1.0f + dir[i] * r;
}

/* particle.c:69-74 */
static void set_velocity(float vel[3], double velocity, uint64_t *rng)
{
    vel[0] = (clapo_drand48(rng) * 2.0 - 1.0) * velocity;
    vel[1] = (clapo_drand48(rng) * 2.0 - 1.0) * velocity;
    vel[2] = (clapo_drand48(rng) * 2.0 - 1.0) * velocity;
}

/* particle_system_make's spawn loop (particle.c:229-234 -> particle_spawn 76-87) */
void clapo_particles_spawn(const clapo_particle_system *sys, uint32_t n_sys,
                           float *pos, float *vel, uint64_t *rng)
{
    for (uint32_t s = 0; s < n_sys; s++)
        for (uint32_t k = 0; k < sys[s].count; k++) {
            size_t i = (size_t)sys[s].first + k;
            random_point_sphere(pos + 3 * i, sys[s].center, sys[s].radius, sys[s].min_radius, sys[s].dist, rng);
            set_velocity(vel + 3 * i, sys[s].velocity, rng);
        }
}

/* particles_update's list walk (particle.c:105-117); pos doubles as pos_array (particle.c:116) */
uint32_t clapo_particles_update(const clapo_particle_system *sys, uint32_t n_sys,
                                float *pos, float *vel, uint64_t *rng)
{
    uint32_t respawned = 0;
    for (uint32_t s = 0; s < n_sys; s++)
        for (uint32_t k = 0; k < sys[s].count; k++) {
            float *p = pos + 3 * ((size_t)sys[s].first + k);
            float *v = vel + 3 * ((size_t)sys[s].first + k);
            float d[3] = { p[0] - sys[s].center[0], p[1] - sys[s].center[1], p[2] - sys[s].center[2] };
            if (lm_dot3(d, d) > sys[s].radius_squared) {        /* float promoted, double compare */
                random_point_sphere(p, sys[s].center, sys[s].radius, sys[s].min_radius, sys[s].dist, rng);
                set_velocity(v, sys[s].velocity, rng);
                respawned++;
            }
            for (int i = 0; i < 3; i++) p[i] = p[i] + v[i];
        }
    return respawned;
}

/* particle.c:93-100: mx = view_mx, transpose the 3x3, column 3 xyz = system position */
void clapo_particles_billboard(const float view_mx[16], const float center[3], float mx[16])
{
    memcpy(mx, view_mx, 16 * sizeof(float));
    lm_m4_transpose_3x3(mx);
    mx[12] = center[0];
    mx[13] = center[1];
    mx[14] = center[2];
}
