// particles.hip -- particle systems for gfx950: per-frame advect / respawn + billboard matrix.
//
// Replaces particles_update() (particle.c:89-120) for every particle system of the model
// queue in one pass.  The reference draws its random numbers from the single libc drand48
// stream while it walks systems and particles in list order, 7 draws per respawned particle
// (random_point_sphere 4 + particle_set_velocity 3, particle.c:36-74).  To reproduce that
// stream bit for bit on a parallel machine:
//
//   1. k_particles_advect   one lane per particle: respawn test, advect the survivors,
//                           ballot the respawn flags into a bitmask (+ per-row popcounts)
//   2. k_particles_respawn_rp  a lane whose flag is set ranks itself among all respawns (byte
//                           prefix over the per-row popcounts + bits below it in its row), jumps
//                           the LCG ahead by 7 * rank draws (affine-map power, mod 2^48) and
//                           regenerates position and velocity.
//      (above 4M particles: ordered compaction (entities.hip) + k_particles_respawn over the list)
//
// HBM: 36 B / particle (pos 12 + vel 12 read, pos 12 written; pos doubles as the pos_array
// the renderer uploads, particle.c:116,124).  Respawns are rare, passes 2-3 are tiny.
#include <string.h>
#include "common.h"
#include "lm_dev.h"

namespace clapgpu {

constexpr int PART_BLOCK = 256;
constexpr uint64_t R48_A = 0x5DEECE66DULL, R48_C = 0xBULL, R48_MASK = (1ULL << 48) - 1;

struct PartK {
    const clapgpu_particle_system *sys;
    const uint32_t *row_sys;
    float    *pos;
    float    *vel;
    uint64_t *rng_state;          // [2]: [0] stream position of this frame, [1] of the next
    uint64_t *respawn_mask;
    uint8_t  *respawn_row_pop;
    float    *billboard_mx;
    uint32_t *groups;             // respawn_groups (see include/clapgpu.h), or NULL
    uint32_t  n, n_sys;
};

// respawn_groups layout: [0] frames completed (written by the respawn pass), [1] this frame's copy of
// it (written by the advect pass), then two banks of per-group respawn counts, one group =
// PART_GROUP_ROWS rows.  A frame accumulates into bank (frame & 1) while clearing the other one for
// the next frame, so no separate clearing launch is needed.
constexpr uint32_t PART_RESPAWN_LIST = 256;      // respawns of one wave's 64 rows handled in one pass
constexpr uint32_t PART_GROUP_ROWS = 1024, PART_GROUPS = 64, PART_GROUP_BANK0 = 4;

struct Mat4Arg { float m[16]; };

__global__ __launch_bounds__(PART_BLOCK)
void k_particles_advect(PartK k, Mat4Arg view)
{
    const uint32_t i = blockIdx.x * PART_BLOCK + threadIdx.x;
    const int lane = lane_id();
    const uint32_t row = (i - lane) >> 6;
    if (i - lane >= k.n)
        return;
    uint32_t *bank = nullptr;
    if (k.groups) {
        const uint32_t frame = __builtin_amdgcn_readfirstlane(k.groups[0]);
        bank = k.groups + PART_GROUP_BANK0 + PART_GROUPS * (frame & 1u);
        if (i < PART_GROUPS)
            k.groups[PART_GROUP_BANK0 + PART_GROUPS * ((frame & 1u) ^ 1u) + i] = 0;     // next frame's bank
        if (i == 0)
            k.groups[1] = frame;
    }
    if (i == 0)
        k.rng_state[0] = k.rng_state[1];                  // last frame's respawn pass has finished

    // position and velocity are requested first: they do not depend on the system record, whose two
    // dependent scalar loads would otherwise sit in front of them
    float px = 0.f, py = 0.f, pz = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
    if (i < k.n) {
        const float *p = k.pos + 3 * (size_t)i, *v = k.vel + 3 * (size_t)i;
        px = p[0]; py = p[1]; pz = p[2];
        vx = v[0]; vy = v[1]; vz = v[2];
    }

    // one system per 64-particle row: wave-uniform -> scalar loads
    uint32_t s = __builtin_amdgcn_readfirstlane(k.row_sys[row]);
    if (s >= k.n_sys) s = 0;                              // a row that names no system would be a wild read
    const clapgpu_particle_system &ps = k.sys[s];
    const float cx = ps.center[0], cy = ps.center[1], cz = ps.center[2];
    const double r2 = ps.radius_squared;
    const uint32_t in_sys = i - ps.first;                 // index inside the system
    const bool live = i < k.n && in_sys < ps.count;

    bool respawn = false;
    if (live) {
        float *p = k.pos + 3 * (size_t)i;
        const float dx = px - cx, dy = py - cy, dz = pz - cz;       // particle.c:109
        float dd = 0.f;
        dd += dx * dx;
        dd += dy * dy;
        dd += dz * dz;
        respawn = (double)dd > r2;                                   // particle.c:110
        if (!respawn) {
            p[0] = px + vx;                                          // particle.c:115-116
            p[1] = py + vy;
            p[2] = pz + vz;
        }
    }
    const uint64_t m = __ballot(respawn);
    if (lane == 0) {
        k.respawn_mask[row] = m;
        k.respawn_row_pop[row] = (uint8_t)__popcll(m);
        if (bank && m)                                    // respawns are rare: a few hundred atomics per frame
            atomicAdd(&bank[row / PART_GROUP_ROWS], (uint32_t)__popcll(m));
    }

    // billboard matrix of the system (particle.c:93-100), once per system
    if (live && in_sys == 0 && k.billboard_mx) {
        float *o = k.billboard_mx + 16 * (size_t)s;
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                o[4 * c + r] = (c < 3 && r < 3) ? view.m[4 * r + c] : view.m[4 * c + r];
        o[12] = cx; o[13] = cy; o[14] = cz;
    }
}

// X_{n+m} = A^m X_n + C_m (mod 2^48): power of the affine map by squaring
__device__ __forceinline__ uint64_t lcg_skip(uint64_t x, uint64_t m)
{
    uint64_t acc_a = 1, acc_c = 0, cur_a = R48_A, cur_c = R48_C;
    while (m) {
        if (m & 1) {
            acc_a = acc_a * cur_a;
            acc_c = acc_c * cur_a + cur_c;
        }
        cur_c = cur_c * (cur_a + 1);
        cur_a = cur_a * cur_a;
        m >>= 1;
    }
    return (acc_a * x + acc_c) & R48_MASK;
}

__device__ __forceinline__ double drand48_next(uint64_t &x)
{
    x = (R48_A * x + R48_C) & R48_MASK;
    return (double)x * (1.0 / 281474976710656.0);
}

// glibc 2.35 sysdeps/ieee754/dbl-64/s_cbrt.c restated (the reference calls libm cbrt(),
// particle.c:50); agrees with it bit for bit on all but ~5e-8 of inputs (1 ulp of double,
// invisible after the cast to float).
__device__ __forceinline__ double cbrt_glibc(double x)
{
    int xe;
    const double xm = frexp(fabs(x), &xe);
    if (xe == 0 && (x == 0.0 || x != x || isinf(x)))
        return x + x;
    const double u = (0.354895765043919860 + ((1.50819193781584896 - ((2.11499494167371287
                   - ((2.44693122563534430 - ((1.83469277483613086 - (0.784932344976639262
                   - 0.145263899385486377 * xm) * xm) * xm)) * xm)) * xm)) * xm));
    const double t2 = u * u * u;
    const int r = xe % 3;
    const double f = r == -2 ? 1.0 / 1.5874010519681994748 : r == -1 ? 1.0 / 1.2599210498948731648
                   : r == 0 ? 1.0 : r == 1 ? 1.2599210498948731648 : 1.5874010519681994748;
    const double ym = u * (t2 + 2.0 * xm) / (2.0 * t2 + xm) * f;
    return ldexp(x > 0.0 ? ym : -ym, xe / 3);
}

// the rank-th respawn of the frame, for particle i (random_point_sphere + particle_set_velocity)
__device__ __forceinline__ void respawn_one(const PartK &k, uint64_t state0, uint32_t rank, uint32_t i)
{
    const clapgpu_particle_system &ps = k.sys[k.row_sys[i >> 6]];
    uint64_t x = lcg_skip(state0, 7ull * rank);

    // random_point_sphere (particle.c:36-67)
    float dx = (float)(drand48_next(x) * 2.0 - 1.0);
    float dy = (float)(drand48_next(x) * 2.0 - 1.0);
    float dz = (float)(drand48_next(x) * 2.0 - 1.0);
    float dd = 0.f;
    dd += dx * dx;
    dd += dy * dy;
    dd += dz * dz;
    const float len = sqrtf(dd);
    if (len) {                                                   // vec3_norm_safe
        const float kk = (float)(1.0 / (double)len);
        dx = dx * kk; dy = dy * kk; dz = dz * kk;
    }
    const double d3 = drand48_next(x);
    double u;
    switch (ps.dist) {
    case CLAPGPU_PART_DIST_POW075: u = pow(d3, 0.75); break;
    case CLAPGPU_PART_DIST_CBRT:   u = cbrt_glibc(d3); break;
    case CLAPGPU_PART_DIST_SQRT:   u = sqrt(d3); break;
    default:                       u = d3; break;
    }
    const float r = (float)(ps.min_radius + (ps.radius - ps.min_radius) * u);
    const float px = ps.center[0] * 1.0f + dx * r;               // vec3_add_scaled(.., 1.0, r)
    const float py = ps.center[1] * 1.0f + dy * r;
    const float pz = ps.center[2] * 1.0f + dz * r;
    // particle_set_velocity (particle.c:69-74)
    const float vx = (float)((drand48_next(x) * 2.0 - 1.0) * ps.velocity);
    const float vy = (float)((drand48_next(x) * 2.0 - 1.0) * ps.velocity);
    const float vz = (float)((drand48_next(x) * 2.0 - 1.0) * ps.velocity);
    float *p = k.pos + 3 * (size_t)i, *v = k.vel + 3 * (size_t)i;
    v[0] = vx; v[1] = vy; v[2] = vz;
    p[0] = px + vx;                                              // particle.c:115
    p[1] = py + vy;
    p[2] = pz + vz;
}

// large-n path: respawns listed by the ordered compaction
__global__ __launch_bounds__(PART_BLOCK)
void k_particles_respawn(PartK k, const uint32_t *list, const uint32_t *count)
{
    const uint32_t total = *count;
    const uint64_t state0 = k.rng_state[0];
    const uint32_t stride = gridDim.x * PART_BLOCK;
    for (uint32_t j = blockIdx.x * PART_BLOCK + threadIdx.x; j < total; j += stride)
        respawn_one(k, state0, j, list[j]);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        k.rng_state[1] = lcg_skip(state0, 7ull * total);         // where the libc stream now stands
}

// n <= 4M: a wave owns 64 rows (one mask word per lane).  Rows without respawns -- almost all of
// them -- cost one load; a wave with work ranks its rows from the per-row popcount bytes.
__global__ __launch_bounds__(PART_BLOCK)
void k_particles_respawn_rp(PartK k, uint32_t *respawn_count)
{
    const int lane = lane_id();
    const uint32_t w = blockIdx.x * (PART_BLOCK / WAVE) + threadIdx.x / WAVE;
    const uint32_t n_rows = k.n / WAVE;
    const uint32_t row0 = w * WAVE;
    if (row0 >= n_rows)
        return;
    const uint32_t my_row = row0 + lane;
    const uint64_t m = my_row < n_rows ? k.respawn_mask[my_row] : 0ull;
    const uint64_t busy = __ballot(m != 0);
    const bool last = row0 + WAVE >= n_rows;
    if (busy == 0 && !last)
        return;                                                  // the common case: nothing to do

    // respawns in rows [0, row0); row0 % 64 == 0.  With the group counts of the advect pass this is
    // one load of <= 64 counts plus <= 1 KiB of popcount bytes instead of up to 64 KiB of bytes.
    uint32_t pre, frame = 0;
    if (k.groups) {
        frame = __builtin_amdgcn_readfirstlane(k.groups[1]);
        const uint32_t *bank = k.groups + PART_GROUP_BANK0 + PART_GROUPS * (frame & 1u);
        const uint32_t g0 = row0 / PART_GROUP_ROWS;
        uint32_t before = (uint32_t)lane < g0 ? bank[lane] : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            before += __shfl_xor(before, off);
        pre = before + wave_byte_sum(k.respawn_row_pop + g0 * PART_GROUP_ROWS, row0 - g0 * PART_GROUP_ROWS, lane);
    } else {
        pre = wave_byte_sum(k.respawn_row_pop, row0, lane);
    }

    const uint32_t cnt = __popcll(m);
    uint32_t incl = cnt;                                         // inclusive scan of the 64 row counts
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    const uint32_t excl = incl - cnt;
    const uint64_t state0 = k.rng_state[0];
    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);

    // All respawns of the wave's 64 rows at once, one per lane: each row lists its respawning particles
    // at its rank offset in LDS (rank order = particle order), then lane t takes the wave's t-th one.
    // (Walking the rows one after the other would leave a single lane busy per row for the whole
    // jump-ahead + 7-draw chain.)
    __shared__ uint32_t lists[PART_BLOCK / WAVE][PART_RESPAWN_LIST];
    uint32_t *list = lists[threadIdx.x / WAVE];
    const uint32_t wave_total = (uint32_t)__shfl((int)incl, WAVE - 1);
    if (wave_total <= PART_RESPAWN_LIST) {
        uint64_t mm = m;
        uint32_t o = excl;
        while (mm) {
            list[o++] = my_row * WAVE + (uint32_t)__builtin_ctzll(mm);
            mm &= mm - 1;
        }
        wave_lds_fence();
        for (uint32_t t = lane; t < wave_total; t += WAVE)
            respawn_one(k, state0, pre + t, list[t]);
    } else {
        uint64_t todo = busy;
        while (todo) {                                           // crowded wave: row by row
            const int r = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint64_t mr = (uint64_t)(uint32_t)__shfl((int)lo, r) | ((uint64_t)(uint32_t)__shfl((int)hi, r) << 32);
            const uint32_t base = pre + (uint32_t)__shfl((int)excl, r);
            if ((mr >> lane) & 1ull)
                respawn_one(k, state0, base + __popcll(mr & ((1ull << lane) - 1ull)), (row0 + r) * WAVE + lane);
        }
    }
    if (last && lane == WAVE - 1) {
        const uint32_t total = pre + incl;
        *respawn_count = total;
        k.rng_state[1] = lcg_skip(state0, 7ull * total);         // where the libc stream now stands
        if (k.groups) k.groups[0] = frame + 1u;
    }
}

} // namespace clapgpu

using namespace clapgpu;

static_assert(sizeof(clapgpu_particle_system) == 64, "clapgpu_particle_system layout");

extern "C" int clapgpu_particles_update(void *stream, const clapgpu_particles *p, const float view_mx[16])
{
    if (!p || !p->sys || !p->row_sys || !p->pos || !p->vel || !p->rng_state || !p->respawn_mask ||
        !p->respawn_row_pop || !p->respawn_list || !p->respawn_count || !view_mx)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (p->n & 63u)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (p->n == 0)
        return CLAPGPU_OK;

    PartK k;
    k.sys = p->sys;
    k.row_sys = p->row_sys;
    k.pos = p->pos;
    k.vel = p->vel;
    k.rng_state = p->rng_state;
    k.respawn_mask = p->respawn_mask;
    k.respawn_row_pop = p->respawn_row_pop;
    k.billboard_mx = p->billboard_mx;
    k.n = p->n;
    k.n_sys = p->n_sys ? p->n_sys : 1;
    const bool row_path = p->n / WAVE <= (1u << 16) && (reinterpret_cast<uintptr_t>(p->respawn_row_pop) & 15u) == 0;
    k.groups = row_path ? p->respawn_groups : nullptr;
    Mat4Arg view;
    memcpy(view.m, view_mx, sizeof(view.m));

    hipLaunchKernelGGL(k_particles_advect, dim3((p->n + PART_BLOCK - 1) / PART_BLOCK), dim3(PART_BLOCK), 0,
                       as_stream(stream), k, view);
    CLAPGPU_LAUNCH_CHECK("k_particles_advect");
    if (row_path) {
        const uint32_t waves = (p->n / WAVE + WAVE - 1) / WAVE, per_block = PART_BLOCK / WAVE;
        hipLaunchKernelGGL(k_particles_respawn_rp, dim3((waves + per_block - 1) / per_block), dim3(PART_BLOCK), 0,
                           as_stream(stream), k, p->respawn_count);
        CLAPGPU_LAUNCH_CHECK("k_particles_respawn_rp");
        return CLAPGPU_OK;
    }
    int rc = clapgpu_visible_compact(stream, p->respawn_mask, nullptr, p->n, 0, p->respawn_list,
                                     p->respawn_count, p->scratch);
    if (rc) return rc;
    const uint32_t blocks = p->n / PART_BLOCK < 256 ? (p->n / PART_BLOCK ? p->n / PART_BLOCK : 1) : 256;
    hipLaunchKernelGGL(k_particles_respawn, dim3(blocks), dim3(PART_BLOCK), 0, as_stream(stream), k,
                       p->respawn_list, p->respawn_count);
    CLAPGPU_LAUNCH_CHECK("k_particles_respawn");
    return CLAPGPU_OK;
}
