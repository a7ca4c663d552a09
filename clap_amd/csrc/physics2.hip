// physics2.hip -- capsule / sphere bodies: integrate, geom AABBs, broadphase over explicit AABBs, narrowphase
// contact records and the capsule sweep, for gfx950.
//
// What phys_step() (physics.c:773-787) does per fixed substep through ODE, for bodies without constraint rows:
//   k_bodies_step     dWorldQuickStep's body stage (quickstep.cpp stage 0 + dxStepBody + auto-disable), fused with
//                     the moved geom's axis / AABB (dxCapsule::computeAABB)                      HBM-bound, 1 lane / body
//   k_bp_*            dSpaceCollide2(ground, bodies) + dSpaceCollide(bodies) (physics.c:751-753) as ascending
//                     candidate-pair lists, four launches for both passes: bodies binned by AABB centre into a hash
//                     grid of 4x4x4-cell blocks (own block + the neighbour blocks its cell touches), one wavefront
//                     per block with the block's candidates staged through LDS and read back as broadcasts
//   k_contacts_geoms  near_callback's dCollide + phys_contact_surface (physics.c:399-449, 291-330)
//   k_sweep_capsules  phys_body_sweep_capsule (physics.c:559-670), one wavefront per sweep
// fp64 throughout (the reference builds ODE with dDOUBLE, physics.h:5-9), no FMA contraction.
// ODE is an absent submodule of the reference: PARITY UNPINNED (oracle/physics2.c states what is restated).
#include <string.h>
#include <stdlib.h>
#include <vector>
#include "common.h"
#include "phys_dev.h"

namespace clapgpu {

constexpr int PB = 256;

struct WorldK2 {
    double  gravity[3];
    double  linear_damping;
    double  linear_damping_threshold_sq;
    double  adis_linear_threshold_sq;
    double  adis_angular_threshold_sq;
    double  adis_time;
    int32_t adis_steps;
    int32_t pad;
};
static_assert(sizeof(WorldK2) == sizeof(clapgpu_world), "world layout");

struct BodiesK {
    uint32_t n, samples;
    double *pos, *quat, *lvel, *avel;
    const double *mass, *radius;
    uint32_t *bflags;
    int32_t *adis_steps_left;
    double *adis_time_left;
    const double *length, *inertia;
    double Roff[12];
    double *aabb, *axis, *adis_samples;
    uint32_t *adis_counter;
};

__device__ __forceinline__ void write_geom(const BodiesK &b, uint32_t i, const double (&p)[3], const double (&q)[4])
{
    if (!b.aabb && !b.axis) return;
    double R[12], axis[3], bb[6];
    phd::q_to_R(q, R);
    phd::capsule_axis(R, b.Roff, axis);
    const double lz = b.length ? b.length[i] : 0.0;
    phd::geom_aabb(p, b.radius[i], lz, axis, bb);
    if (b.axis) { double *a = b.axis + 3 * (size_t)i; a[0] = axis[0]; a[1] = axis[1]; a[2] = axis[2]; }
    if (b.aabb) {
        double2 *o = reinterpret_cast<double2 *>(b.aabb + 6 * (size_t)i);
        o[0] = make_double2(bb[0], bb[1]); o[1] = make_double2(bb[2], bb[3]); o[2] = make_double2(bb[4], bb[5]);
    }
}

__global__ __launch_bounds__(PB)
void k_bodies_aabb(BodiesK b)
{
    const uint32_t i = blockIdx.x * PB + threadIdx.x;
    if (i >= b.n) return;
    const double p[3] = { b.pos[3 * (size_t)i], b.pos[3 * (size_t)i + 1], b.pos[3 * (size_t)i + 2] };
    const double q[4] = { b.quat[4 * (size_t)i], b.quat[4 * (size_t)i + 1], b.quat[4 * (size_t)i + 2], b.quat[4 * (size_t)i + 3] };
    write_geom(b, i, p, q);
}

__global__ __launch_bounds__(PB)
void k_bodies_step(BodiesK b, WorldK2 w, double h)
{
    const uint32_t i = blockIdx.x * PB + threadIdx.x;
    if (i >= b.n) return;
    uint32_t fl = b.bflags[i];
    if (fl & CLAPGPU_BODY_DISABLED) return;
    double *pp = b.pos + 3 * (size_t)i, *qp = b.quat + 4 * (size_t)i, *vp = b.lvel + 3 * (size_t)i, *op = b.avel + 3 * (size_t)i;
    double v[3] = { vp[0], vp[1], vp[2] }, om[3] = { op[0], op[1], op[2] };

    // dInternalHandleAutoDisabling: enabled bodies with the flag that hold a joint
    if ((fl & CLAPGPU_BODY_AUTO_DISABLE) && (fl & CLAPGPU_BODY_HAS_JOINT)) {
        bool idle = false;
        double al[3], aa[3];
        const uint32_t S = b.samples > 1 ? b.samples : 1;
        if (S == 1) {
            for (int a = 0; a < 3; a++) { al[a] = v[a]; aa[a] = om[a]; }
            idle = true;
        } else {
            double *ring = b.adis_samples + (size_t)i * S * 6;
            uint32_t c = b.adis_counter[i] & 0x7fffffffu, ready = b.adis_counter[i] >> 31;
            for (int a = 0; a < 3; a++) { ring[6 * (size_t)c + a] = v[a]; ring[6 * (size_t)c + 3 + a] = om[a]; }
            if (++c >= S) { c = 0; ready = 1; }
            b.adis_counter[i] = c | ready << 31;
            if (ready) {
                idle = true;
                for (int a = 0; a < 3; a++) { al[a] = ring[a]; aa[a] = ring[3 + a]; }
                for (uint32_t s = 1; s < S; s++)
                    for (int a = 0; a < 3; a++) { al[a] += ring[6 * (size_t)s + a]; aa[a] += ring[6 * (size_t)s + 3 + a]; }
                const double r1 = 1.0 / (double)S;
                for (int a = 0; a < 3; a++) { al[a] *= r1; aa[a] *= r1; }
            }
        }
        if (idle) {
            if (al[0] * al[0] + al[1] * al[1] + al[2] * al[2] > w.adis_linear_threshold_sq) idle = false;
            else if (aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2] > w.adis_angular_threshold_sq) idle = false;
        }
        int32_t sl = b.adis_steps_left[i];
        double tl = b.adis_time_left[i];
        if (idle) { sl--; tl -= h; } else { sl = w.adis_steps; tl = w.adis_time; }
        b.adis_steps_left[i] = sl;
        b.adis_time_left[i] = tl;
        if (sl <= 0 && tl <= 0) {
            b.bflags[i] = (fl | CLAPGPU_BODY_DISABLED) & ~CLAPGPU_BODY_HAS_JOINT;
            vp[0] = vp[1] = vp[2] = 0;
            op[0] = op[1] = op[2] = 0;
            return;
        }
    }
    if (fl & CLAPGPU_BODY_HAS_JOINT)
        b.bflags[i] = fl & ~CLAPGPU_BODY_HAS_JOINT;                   // dJointGroupEmpty after the step

    double q[4] = { qp[0], qp[1], qp[2], qp[3] };
    double tacc[3] = { 0, 0, 0 }, invIw[12];
    const bool have_inertia = b.inertia != nullptr;
    if (have_inertia) {
        const double Ib[3] = { b.inertia[3 * (size_t)i], b.inertia[3 * (size_t)i + 1], b.inertia[3 * (size_t)i + 2] };
        const double invIb[3] = { 1.0 / Ib[0], 1.0 / Ib[1], 1.0 / Ib[2] };
        double R[12];
        phd::q_to_R(q, R);
        phd::world_tensor(R, invIb, invIw);
        if (fl & CLAPGPU_BODY_GYROSCOPIC) {                             // implicit gyroscopic torque (quickstep.cpp stage 0)
            double Iw[12], L[3], Itild[12], itInv[12];
            phd::world_tensor(R, Ib, Iw);
            phd::mul331(L, Iw, om);
            for (int k = 0; k < 12; k++) Itild[k] = 0;
            Itild[1] = L[2]; Itild[2] = -L[1];                          // dSetCrossMatrixMinus
            Itild[4] = -L[2]; Itild[6] = L[0];
            Itild[8] = L[1]; Itild[9] = -L[0];
            for (int k = 0; k < 12; k++) Itild[k] = Itild[k] * h + Iw[k];
            const double rh = 1.0 / h;
            L[0] *= rh; L[1] *= rh; L[2] *= rh;
            if (phd::invert3(itInv, Itild)) {
                double T[12], tau0[3];
                for (int r = 0; r < 3; r++) {
                    for (int c = 0; c < 3; c++)
                        T[4 * r + c] = Iw[4 * r] * itInv[c] + Iw[4 * r + 1] * itInv[4 + c] + Iw[4 * r + 2] * itInv[8 + c];
                    T[4 * r + 3] = 0;
                }
                T[0] -= 1; T[5] -= 1; T[10] -= 1;
                phd::mul331(tau0, T, L);
                tacc[0] += tau0[0]; tacc[1] += tau0[1]; tacc[2] += tau0[2];
            }
        }
    }
    const double m = b.mass[i];
    const double k = h * (1.0 / m);
    const bool grav = !(fl & CLAPGPU_BODY_NO_GRAVITY);
    for (int j = 0; j < 3; j++)
        v[j] += k * (grav ? m * w.gravity[j] : 0.0);
    if (have_inertia) {
        double d[3];
        tacc[0] *= h; tacc[1] *= h; tacc[2] *= h;
        phd::mul331(d, invIw, tacc);
        om[0] += d[0]; om[1] += d[1]; om[2] += d[2];
        op[0] = om[0]; op[1] = om[1]; op[2] = om[2];
    }
    double p[3] = { pp[0], pp[1], pp[2] };
    for (int j = 0; j < 3; j++) p[j] += h * v[j];                      // dxStepBody
    pp[0] = p[0]; pp[1] = p[1]; pp[2] = p[2];
    const double d0 = 0.5 * (-om[0] * q[1] - om[1] * q[2] - om[2] * q[3]);   // dWtoDQ
    const double d1 = 0.5 * ( om[0] * q[0] + om[1] * q[3] - om[2] * q[2]);
    const double d2 = 0.5 * (-om[0] * q[3] + om[1] * q[0] + om[2] * q[1]);
    const double d3 = 0.5 * ( om[0] * q[2] - om[1] * q[1] + om[2] * q[0]);
    q[0] += h * d0; q[1] += h * d1; q[2] += h * d2; q[3] += h * d3;
    double l = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];   // dNormalize4
    if (l > 0) {
        l = 1.0 / sqrt(l);
        q[0] *= l; q[1] *= l; q[2] *= l; q[3] *= l;
    } else {
        q[0] = 1; q[1] = q[2] = q[3] = 0;
    }
    qp[0] = q[0]; qp[1] = q[1]; qp[2] = q[2]; qp[3] = q[3];
    if (w.linear_damping != 0.0) {
        const double speed2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        if (speed2 > w.linear_damping_threshold_sq) {
            const double s = 1 - w.linear_damping;
            v[0] *= s; v[1] *= s; v[2] *= s;
        }
    }
    vp[0] = v[0]; vp[1] = v[1]; vp[2] = v[2];
    write_geom(b, i, p, q);
}

// ================================================================================== broadphase
constexpr int BP_LIST = 16;            // partners kept per body in its fixed slot; longer lists go to the arena
constexpr int BP_TILE = 64;            // candidates staged per round (one per lane)
constexpr int BP_EMIT_TILE = 256;      // bodies per tile of the pair-offset scan (= emit block)
constexpr int BP_MAX_TARGETS = 8;      // own block + at most 7 neighbour blocks a cell can touch
constexpr int CTRL_TICKET_BIN = 0, CTRL_TICKET_SEARCH = 1, CTRL_ARENA = 2, CTRL_SARENA = 3, CTRL_STATUS = 4;

__host__ __device__ __forceinline__ uint32_t block_hash(int32_t bx, int32_t by, int32_t bz, uint32_t mask)
{
    const uint32_t h = ((uint32_t)bx * 73856093u) ^ ((uint32_t)by * 19349663u) ^ ((uint32_t)bz * 83492791u);
    return (h ^ (h >> 15)) & mask;
}

__host__ __device__ __forceinline__ int32_t cell_coord(double x, double cell)
{
    double c = floor(x / cell);
    if (!(c > -5.0e8)) c = -5.0e8;                                       // also catches NaN
    if (c > 5.0e8) c = 5.0e8;
    return (int32_t)c;
}

struct BpK {
    uint32_t n;
    double cell;
    uint32_t mask;                       // buckets - 1
    const double *aabb;
    unsigned long long *bucket_cnt;      // [buckets] own << 32 | total, zero between frames
    uint32_t *bucket_start;              // [buckets + 1]
    uint32_t *bucket_own;                // [buckets]
    uint32_t *ranks;                     // [n][8]
    uint32_t *entries;                   // [8 n] body indices in bucket order: own ... | ... halo
    uint32_t *cnt, *scnt;                // [n] partners (larger index) / statics per body
    uint32_t *ref, *sref;                // [n] arena offsets of lists longer than BP_LIST
    uint32_t *partners, *spartners;      // [n][BP_LIST]
    uint32_t *arena, *sarena;            // [capacity], [static capacity]
    uint32_t *tile_sum, *stile_sum;      // [tiles] zero between frames
    uint32_t *tile_off, *stile_off;      // [tiles]
    uint32_t *ctrl;
    uint32_t n_tiles;
    // statics (binned on the host at create time)
    const uint32_t *s_start;             // [buckets + 1]
    const uint32_t *s_entries;
    const double *s_aabb;
    const uint32_t *s_large;
    uint32_t n_large, n_static;
    // outputs
    uint32_t *pairs, capacity, *pair_total;
    uint32_t *spairs, scapacity, *spair_total;
};

// the buckets a body is entered into: its own block first, then the neighbour blocks its cell borders
// (distinct bucket ids only: blocks that share a hash slot get one entry)
__device__ __forceinline__ int bp_targets(const double (&bb)[6], double cell, uint32_t mask, uint32_t (&t)[BP_MAX_TARGETS])
{
    const int32_t cx = cell_coord((bb[0] + bb[1]) * 0.5, cell), cy = cell_coord((bb[2] + bb[3]) * 0.5, cell),
                  cz = cell_coord((bb[4] + bb[5]) * 0.5, cell);
    const int32_t bx = cx >> 2, by = cy >> 2, bz = cz >> 2;
    const int dx = (cx & 3) == 0 ? -1 : (cx & 3) == 3 ? 1 : 0;
    const int dy = (cy & 3) == 0 ? -1 : (cy & 3) == 3 ? 1 : 0;
    const int dz = (cz & 3) == 0 ? -1 : (cz & 3) == 3 ? 1 : 0;
    int nt = 0;
    t[nt++] = block_hash(bx, by, bz, mask);
#pragma unroll
    for (int m = 1; m < 8; m++) {
        const int ox = (m & 1) ? dx : 0, oy = (m & 2) ? dy : 0, oz = (m & 4) ? dz : 0;
        if (((m & 1) && !dx) || ((m & 2) && !dy) || ((m & 4) && !dz)) continue;
        const uint32_t h = block_hash(bx + ox, by + oy, bz + oz, mask);
        bool dup = false;
        for (int e = 0; e < nt; e++) dup |= t[e] == h;
        if (!dup) t[nt++] = h;
    }
    return nt;
}

__device__ __forceinline__ void load_box(const double *aabb, uint32_t i, double (&bb)[6])
{
    const double2 *p = reinterpret_cast<const double2 *>(aabb + 6 * (size_t)i);
    const double2 a = p[0], b = p[1], c = p[2];
    bb[0] = a.x; bb[1] = a.y; bb[2] = b.x; bb[3] = b.y; bb[4] = c.x; bb[5] = c.y;
}

constexpr int BIN_BLOCK = 1024;

// Launch 1: count the entries per bucket (the atomic's return value is the entry's rank); the block that
// finishes last turns the counts into bucket starts and clears them for the next frame.
__global__ __launch_bounds__(BIN_BLOCK)
void k_bp_bin(BpK k)
{
    __shared__ uint32_t lds[BIN_BLOCK / WAVE];
    __shared__ bool is_last;
    const uint32_t i = blockIdx.x * BIN_BLOCK + threadIdx.x;
    if (i < k.n) {
        double bb[6];
        load_box(k.aabb, i, bb);
        if (bb[1] - bb[0] > k.cell || bb[3] - bb[2] > k.cell || bb[5] - bb[4] > k.cell)
            atomicOr(&k.ctrl[CTRL_STATUS], 1u);
        uint32_t t[BP_MAX_TARGETS];
        const int nt = bp_targets(bb, k.cell, k.mask, t);
        uint4 r0 = make_uint4(0, 0, 0, 0), r1 = make_uint4(0, 0, 0, 0);
        uint32_t r[BP_MAX_TARGETS] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
        for (int e = 0; e < BP_MAX_TARGETS; e++) {
            if (e < nt) {
                const unsigned long long old = atomicAdd(&k.bucket_cnt[t[e]], e == 0 ? ((1ull << 32) | 1ull) : 1ull);
                r[e] = e == 0 ? (uint32_t)(old >> 32) : (uint32_t)old - (uint32_t)(old >> 32);
            }
        }
        r0 = make_uint4(r[0], r[1], r[2], r[3]);
        r1 = make_uint4(r[4], r[5], r[6], r[7]);
        uint4 *rp = reinterpret_cast<uint4 *>(k.ranks + 8 * (size_t)i);
        rp[0] = r0;
        rp[1] = r1;
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
        is_last = atomicAdd(&k.ctrl[CTRL_TICKET_BIN], 1u) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    // exclusive scan of the bucket totals by this one block
    const uint32_t nb = k.mask + 1;
    const int lane = lane_id(), wave = threadIdx.x / WAVE;
    uint32_t carry = 0;
    constexpr int ITEMS = 8;
    for (uint32_t base = 0; base < nb; base += BIN_BLOCK * ITEMS) {
        const uint32_t first = base + threadIdx.x * ITEMS;
        uint32_t tot[ITEMS], own[ITEMS], s = 0;
#pragma unroll
        for (int e = 0; e < ITEMS; e++) {
            unsigned long long c = 0;
            if (first + e < nb) {
                c = __hip_atomic_load(&k.bucket_cnt[first + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                k.bucket_cnt[first + e] = 0;
            }
            tot[e] = (uint32_t)c;
            own[e] = (uint32_t)(c >> 32);
            s += tot[e];
        }
        uint32_t incl = s;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off);
            if (lane >= off) incl += u;
        }
        if (lane == WAVE - 1) lds[wave] = incl;
        __syncthreads();
        uint32_t wave_off = 0, chunk_total = 0;
        for (int q = 0; q < BIN_BLOCK / WAVE; q++) {
            const uint32_t v = lds[q];
            if (q < wave) wave_off += v;
            chunk_total += v;
        }
        __syncthreads();
        uint32_t run = carry + wave_off + incl - s;
#pragma unroll
        for (int e = 0; e < ITEMS; e++) {
            if (first + e < nb) {
                k.bucket_start[first + e] = run;
                k.bucket_own[first + e] = own[e];
            }
            run += tot[e];
        }
        carry += chunk_total;
    }
    if (threadIdx.x == 0) {
        k.bucket_start[nb] = carry;
        k.ctrl[CTRL_TICKET_BIN] = 0;
        k.ctrl[CTRL_ARENA] = 0;
        k.ctrl[CTRL_SARENA] = 0;
    }
}

// Launch 2: body indices into bucket order, the owners of a bucket in front, its halo behind
__global__ __launch_bounds__(PB)
void k_bp_scatter(BpK k)
{
    const uint32_t i = blockIdx.x * PB + threadIdx.x;
    if (i >= k.n) return;
    double bb[6];
    load_box(k.aabb, i, bb);
    uint32_t t[BP_MAX_TARGETS];
    const int nt = bp_targets(bb, k.cell, k.mask, t);
    const uint4 *rp = reinterpret_cast<const uint4 *>(k.ranks + 8 * (size_t)i);
    const uint4 r0 = rp[0], r1 = rp[1];
    const uint32_t r[BP_MAX_TARGETS] = { r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w };
#pragma unroll
    for (int e = 0; e < BP_MAX_TARGETS; e++) {
        if (e < nt) {
            const uint32_t slot = e == 0 ? k.bucket_start[t[e]] + r[e] : k.bucket_start[t[e] + 1] - 1 - r[e];
            k.entries[slot] = i;
        }
    }
}

struct Tile {
    double lo[3][BP_TILE], hi[3][BP_TILE];
    uint32_t idx[BP_TILE];
};

// one round of tests of the lane's own box against the nc staged boxes; all lanes read the same box (LDS broadcast)
template <bool ORDERED, typename F>
__device__ __forceinline__ void test_tile(const Tile &t, int nc, bool has, uint32_t ia, const double (&a)[6], F &&on_hit)
{
    for (int c = 0; c < nc; c++) {
        const uint32_t jc = t.idx[c];
        const bool hit = has && (!ORDERED || jc > ia) &&
                         !(a[0] > t.hi[0][c] || a[1] < t.lo[0][c] || a[2] > t.hi[1][c] || a[3] < t.lo[1][c] ||
                           a[4] > t.hi[2][c] || a[5] < t.lo[2][c]);
        if (hit) on_hit(jc);
    }
}

__device__ __forceinline__ void stage(Tile &t, int lane, int nc, uint32_t j, const double *aabb)
{
    if (lane < nc) {
        double bb[6];
        load_box(aabb, j, bb);
        t.idx[lane] = j;
#pragma unroll
        for (int a = 0; a < 3; a++) { t.lo[a][lane] = bb[2 * a]; t.hi[a][lane] = bb[2 * a + 1]; }
    }
}

// every candidate of one own chunk: the bucket's bodies (ordered: partner index > own index), then the statics
// registered for the bucket and the large statics
template <typename FB, typename FS>
__device__ __forceinline__ void sweep_candidates(const BpK &k, Tile &t, int lane, uint32_t start, uint32_t T, uint32_t s0, uint32_t s1,
                                                 bool has, uint32_t ia, const double (&a)[6], bool bodies, bool statics,
                                                 FB &&on_body, FS &&on_static)
{
    if (bodies) {
        for (uint32_t c0 = 0; c0 < T; c0 += BP_TILE) {
            const int nc = T - c0 < BP_TILE ? (int)(T - c0) : BP_TILE;
            stage(t, lane, nc, lane < nc ? k.entries[start + c0 + lane] : 0, k.aabb);
            wave_lds_fence();
            test_tile<true>(t, nc, has, ia, a, on_body);
            wave_lds_fence();
        }
    }
    if (statics) {
        for (uint32_t c0 = s0; c0 < s1; c0 += BP_TILE) {
            const int nc = s1 - c0 < BP_TILE ? (int)(s1 - c0) : BP_TILE;
            stage(t, lane, nc, lane < nc ? k.s_entries[c0 + lane] : 0, k.s_aabb);
            wave_lds_fence();
            test_tile<false>(t, nc, has, ia, a, on_static);
            wave_lds_fence();
        }
        for (uint32_t c0 = 0; c0 < k.n_large; c0 += BP_TILE) {
            const int nc = k.n_large - c0 < BP_TILE ? (int)(k.n_large - c0) : BP_TILE;
            stage(t, lane, nc, lane < nc ? k.s_large[c0 + lane] : 0, k.s_aabb);
            wave_lds_fence();
            test_tile<false>(t, nc, has, ia, a, on_static);
            wave_lds_fence();
        }
    }
}

// Launch 3: one wavefront per bucket.  Lane = one of the bucket's own bodies (its box in registers); the bucket's
// entries are staged 64 at a time in the wave's LDS tile and every lane walks the tile.  A pair (i, j), i < j, is
// recorded by the lane that owns i: all partners of a body are found by one lane, so its list needs no atomics.
// Lists go to the body's fixed slot (unsorted; the emit pass ranks them), longer ones to the arena.  Per-tile
// partner sums are accumulated for the emit pass; the last block turns them into tile offsets and totals.
__global__ __launch_bounds__(PB)
void k_bp_search(BpK k)
{
    __shared__ Tile tiles[PB / WAVE];
    __shared__ uint32_t lds[PB / WAVE];
    __shared__ bool is_last;
    const int lane = lane_id(), wave = threadIdx.x / WAVE;
    const uint32_t b = blockIdx.x * (PB / WAVE) + wave;
    Tile &t = tiles[wave];
    if (b <= k.mask) {
        const uint32_t start = k.bucket_start[b], T = k.bucket_start[b + 1] - start, M = k.bucket_own[b];
        const bool with_statics = k.n_static != 0;
        const uint32_t s0 = with_statics ? k.s_start[b] : 0, s1 = with_statics ? k.s_start[b + 1] : 0;
        for (uint32_t oc = 0; oc < M; oc += WAVE) {
            const bool has = oc + lane < M;
            const uint32_t ia = has ? k.entries[start + oc + lane] : 0xffffffffu;
            double a[6] = { 0, 0, 0, 0, 0, 0 };
            if (has) load_box(k.aabb, ia, a);
            uint32_t cnt = 0, scnt = 0;
            uint32_t *mine = k.partners + (size_t)BP_LIST * (has ? ia : 0), *smine = k.spartners + (size_t)BP_LIST * (has ? ia : 0);
            sweep_candidates(k, t, lane, start, T, s0, s1, has, ia, a, true, with_statics,
                             [&](uint32_t j) { if (cnt < BP_LIST) mine[cnt] = j; cnt++; },
                             [&](uint32_t s) { if (scnt < BP_LIST) smine[scnt] = s; scnt++; });
            // long lists: a second walk writes them whole into the arena
            const bool ovf = has && cnt > BP_LIST, sovf = has && scnt > BP_LIST;
            if (__any(ovf || sovf)) {
                uint32_t base = 0, sbase = 0;
                if (ovf) base = atomicAdd(&k.ctrl[CTRL_ARENA], cnt);
                if (sovf) sbase = atomicAdd(&k.ctrl[CTRL_SARENA], scnt);
                const bool w_ok = ovf && (unsigned long long)base + cnt <= k.capacity;
                const bool s_ok = sovf && (unsigned long long)sbase + scnt <= k.scapacity;
                uint32_t e = 0, se = 0;
                sweep_candidates(k, t, lane, start, T, s0, s1, has, ia, a, __any(w_ok), __any(s_ok),
                                 [&](uint32_t j) { if (w_ok) k.arena[base + e] = j; e++; },
                                 [&](uint32_t s) { if (s_ok) k.sarena[sbase + se] = s; se++; });
                if (ovf) k.ref[ia] = w_ok ? base : 0xffffffffu;
                if (sovf) k.sref[ia] = s_ok ? sbase : 0xffffffffu;
            }
            if (has) {
                k.cnt[ia] = cnt;
                if (cnt) atomicAdd(&k.tile_sum[ia / BP_EMIT_TILE], cnt);
                if (with_statics) {
                    k.scnt[ia] = scnt;
                    if (scnt) atomicAdd(&k.stile_sum[ia / BP_EMIT_TILE], scnt);
                }
            }
        }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
        is_last = atomicAdd(&k.ctrl[CTRL_TICKET_SEARCH], 1u) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    // exclusive scan of the tile sums (both lists) by this block
    for (int which = 0; which < 2; which++) {
        uint32_t *sum = which ? k.stile_sum : k.tile_sum, *off = which ? k.stile_off : k.tile_off;
        uint32_t *total = which ? k.spair_total : k.pair_total;
        if (which && !k.n_static) {
            if (threadIdx.x == 0 && total) *total = 0;
            continue;
        }
        uint32_t carry = 0;
        for (uint32_t base = 0; base < k.n_tiles; base += PB) {
            const uint32_t i = base + threadIdx.x;
            uint32_t v = 0;
            if (i < k.n_tiles) {
                v = __hip_atomic_load(&sum[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sum[i] = 0;
            }
            uint32_t incl = v;
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) {
                const uint32_t u = __shfl_up(incl, o);
                if (lane >= o) incl += u;
            }
            if (lane == WAVE - 1) lds[wave] = incl;
            __syncthreads();
            uint32_t wave_off = 0, chunk = 0;
            for (int q = 0; q < PB / WAVE; q++) {
                const uint32_t x = lds[q];
                if (q < wave) wave_off += x;
                chunk += x;
            }
            __syncthreads();
            if (i < k.n_tiles) off[i] = carry + wave_off + incl - v;
            carry += chunk;
        }
        if (threadIdx.x == 0 && total) *total = carry;
    }
    if (threadIdx.x == 0) k.ctrl[CTRL_TICKET_SEARCH] = 0;
}

// Launch 4: one thread per body in index order: its offset = tile offset + scan inside the tile; its list is
// written in ascending partner order (rank = number of smaller entries: the lists are short).
__global__ __launch_bounds__(BP_EMIT_TILE)
void k_bp_emit(BpK k)
{
    __shared__ uint32_t lds[2][BP_EMIT_TILE / WAVE];
    const uint32_t i = blockIdx.x * BP_EMIT_TILE + threadIdx.x;
    const int lane = lane_id(), wave = threadIdx.x / WAVE;
    const bool with_statics = k.n_static != 0;
    const uint32_t c[2] = { i < k.n ? k.cnt[i] : 0, (with_statics && i < k.n) ? k.scnt[i] : 0 };
    uint32_t incl[2] = { c[0], c[1] };
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t u0 = __shfl_up(incl[0], o), u1 = __shfl_up(incl[1], o);
        if (lane >= o) { incl[0] += u0; incl[1] += u1; }
    }
    if (lane == WAVE - 1) { lds[0][wave] = incl[0]; lds[1][wave] = incl[1]; }
    __syncthreads();
    uint32_t woff[2] = { 0, 0 };
    for (int q = 0; q < wave; q++) { woff[0] += lds[0][q]; woff[1] += lds[1][q]; }
    if (i >= k.n) return;
    for (int which = 0; which < (with_statics ? 2 : 1); which++) {
        const uint32_t n_mine = c[which];
        if (!n_mine) continue;
        uint32_t *out = which ? k.spairs : k.pairs;
        const uint32_t cap = which ? k.scapacity : k.capacity;
        const uint32_t off = (which ? k.stile_off : k.tile_off)[blockIdx.x] + woff[which] + incl[which] - n_mine;
        const uint32_t *src;
        if (n_mine <= BP_LIST) {
            src = (which ? k.spartners : k.partners) + (size_t)BP_LIST * i;
        } else {
            const uint32_t r = (which ? k.sref : k.ref)[i];
            if (r == 0xffffffffu) continue;                                 // arena full: total > capacity anyway
            src = (which ? k.sarena : k.arena) + r;
        }
        for (uint32_t e = 0; e < n_mine; e++) {
            const uint32_t v = src[e];
            uint32_t rank = 0;
            for (uint32_t f = 0; f < n_mine; f++) rank += src[f] < v;
            if (off + rank < cap)
                reinterpret_cast<uint2 *>(out)[off + rank] = make_uint2(i, v);
        }
    }
}

// ================================================================================== narrowphase
struct GeomsK {
    uint32_t n;
    const double *pos, *axis, *radius, *length, *aabb, *material;
    const uint8_t *kind;
};

__device__ __forceinline__ void load_geom(const GeomsK &g, uint32_t i, phd::Geom &o)
{
    o.kind = g.kind ? g.kind[i] : ((g.length && g.length[i] != 0.0) ? CLAPGPU_GEOM_CAPSULE : CLAPGPU_GEOM_SPHERE);
    for (int a = 0; a < 3; a++) {
        o.pos[a] = g.pos ? g.pos[3 * (size_t)i + a] : 0.0;
        o.axis[a] = g.axis ? g.axis[3 * (size_t)i + a] : 0.0;
    }
    o.radius = g.radius ? g.radius[i] : 0.0;
    o.length = g.length ? g.length[i] : 0.0;
    for (int a = 0; a < 6; a++) o.aabb[a] = (g.aabb && o.kind == CLAPGPU_GEOM_BOX) ? g.aabb[6 * (size_t)i + a] : 0.0;
}

__device__ __forceinline__ void contact_surface2(clapgpu_contact2 &c, const double *m1, const double *m2)
{
    double bounce = 0, bounce_vel = 0, mu = 0, soft_erp = 0.05, soft_cfm = 0.01;   // physics.c:293-294
    if (m1 && m2) {
        bounce = fmax(m1[0], m2[0]);
        bounce_vel = (m1[1] + m2[1]) * 0.5;
        mu = sqrt(m1[2] * m2[2]);
        if (m1[3] > 0 && m2[3] > 0) soft_erp = fmin(m1[3], m2[3]);
        else if (m1[3] > 0) soft_erp = m1[3];
        else if (m2[3] > 0) soft_erp = m2[3];
        if (m1[4] > 0 && m2[4] > 0) soft_cfm = fmax(m1[4], m2[4]);
        else if (m1[4] > 0) soft_cfm = m1[4];
        else if (m2[4] > 0) soft_cfm = m2[4];
    }
    c.mode = CLAPGPU_CONTACT_SOFT_CFM | CLAPGPU_CONTACT_SOFT_ERP | (bounce > 0 ? CLAPGPU_CONTACT_BOUNCE : 0);
    c.mu = mu; c.bounce = bounce; c.bounce_vel = bounce_vel; c.soft_erp = soft_erp; c.soft_cfm = soft_cfm;
}

__global__ __launch_bounds__(PB)
void k_contacts_geoms(GeomsK A, GeomsK B, const uint2 *pairs, const uint32_t *pair_total, uint32_t capacity,
                      clapgpu_contact2 *out, uint32_t *contact_total, uint32_t *flags_a, uint32_t *flags_b)
{
    __shared__ uint32_t block_hits;
    if (threadIdx.x == 0) block_hits = 0;
    __syncthreads();
    uint32_t np = *pair_total;
    if (np > capacity) np = capacity;
    uint32_t mine = 0;
    for (uint32_t p = blockIdx.x * PB + threadIdx.x; p < np; p += gridDim.x * PB) {
        const uint2 pr = pairs[p];
        clapgpu_contact2 c;
        memset(&c, 0, sizeof(c));
        if (pr.x < A.n && pr.y < B.n) {
            phd::Geom ga, gb;
            load_geom(A, pr.x, ga);
            load_geom(B, pr.y, gb);
            phd::CGeom c0, c1;
            memset(&c0, 0, sizeof(c0));
            memset(&c1, 0, sizeof(c1));
            const int nc = phd::collide(ga, gb, c0, c1);
            if (nc < 0) {
                c.nc = CLAPGPU_CONTACT_DEEP;
                mine++;
            } else if (nc > 0) {
                for (int a = 0; a < 3; a++) { c.pos[a] = c0.pos[a]; c.normal[a] = c0.normal[a]; }
                c.depth = c0.depth;
                if (nc > 1) {
                    for (int a = 0; a < 3; a++) { c.pos2[a] = c1.pos[a]; c.normal2[a] = c1.normal[a]; }
                    c.depth2 = c1.depth;
                }
                contact_surface2(c, (A.material && B.material) ? A.material + 5 * (size_t)pr.x : nullptr,
                                 (A.material && B.material) ? B.material + 5 * (size_t)pr.y : nullptr);
                c.nc = (uint32_t)nc;
                mine++;
                if (flags_a) atomicOr(&flags_a[pr.x], CLAPGPU_BODY_HAS_JOINT);
                if (flags_b) atomicOr(&flags_b[pr.y], CLAPGPU_BODY_HAS_JOINT);
            }
        }
        out[p] = c;
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane_id() == 0 && mine) atomicAdd(&block_hits, mine);
    __syncthreads();
    if (threadIdx.x == 0 && block_hits && contact_total) atomicAdd(contact_total, block_hits);
}

// phys_body_sweep_capsule: one wavefront per sweep, the candidates of a step spread over the lanes
__global__ __launch_bounds__(PB)
void k_sweep_capsules(GeomsK A, GeomsK B, uint32_t n_sweeps, const uint32_t *sweep_body, const float *delta_in,
                      const uint32_t *cand_first, const uint32_t *cand, float *frac_out, float *normal_out, int32_t *hit_out)
{
    const int lane = lane_id();
    const uint32_t sw = blockIdx.x * (PB / WAVE) + threadIdx.x / WAVE;
    if (sw >= n_sweeps) return;
    const uint32_t self = sweep_body[sw];
    const float delta[3] = { delta_in[3 * (size_t)sw], delta_in[3 * (size_t)sw + 1], delta_in[3 * (size_t)sw + 2] };
    const float delta_len = sqrtf(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
    float best_frac = 1.0f, best_normal[3] = { 0.f, 1.f, 0.f };
    int32_t best_hit = -1;
    if (!(delta_len < 1e-6f) && self < A.n) {
        phd::Geom probe;
        load_geom(A, self, probe);
        const double gp[3] = { probe.pos[0], probe.pos[1], probe.pos[2] };
        const float k = 1.0f / delta_len;
        const float dir[3] = { delta[0] * k, delta[1] * k, delta[2] * k };
        int nsteps = (int)ceilf((float)(delta_len / (probe.radius * 0.5f)));
        if (nsteps < 2) nsteps = 2;
        const uint32_t c0 = cand_first[sw], c1 = cand_first[sw + 1];
        for (int s = 1; s <= nsteps; s++) {
            const float t = (float)s / nsteps;
            probe.pos[0] = gp[0] + delta[0] * t;
            probe.pos[1] = gp[1] + delta[1] * t;
            probe.pos[2] = gp[2] + delta[2] * t;
            uint32_t taken = 0;                                              // contacts of this step so far (cap 16)
            // (frac, order) of the wave's best contact this step; order = position in the candidate sequence
            float step_frac = best_frac;
            uint32_t step_order = 0xffffffffu;
            float step_normal[3] = { 0, 0, 0 };
            int32_t step_hit = -1;
            for (uint32_t base = c0; base < c1 && taken < 16; base += WAVE) {
                const uint32_t kk = base + lane;
                int nc = 0;
                phd::CGeom cg[2];
                memset(cg, 0, sizeof(cg));
                bool is_body = false;
                uint32_t id = 0;
                if (kk < c1) {
                    const uint32_t cv = cand[kk];
                    is_body = (cv >> 31) != 0;
                    id = cv & 0x7fffffffu;
                    if (!(is_body && id == self) && id < (is_body ? A.n : B.n)) {
                        phd::Geom other;
                        load_geom(is_body ? A : B, id, other);
                        nc = phd::collide(probe, other, cg[0], cg[1]);
                        if (nc < 0) nc = 0;
                    }
                }
                // ordinal of this lane's first contact among the step's contacts
                uint32_t incl = (uint32_t)nc;
#pragma unroll
                for (int o = 1; o < WAVE; o <<= 1) {
                    const uint32_t u = __shfl_up(incl, o);
                    if (lane >= o) incl += u;
                }
                const uint32_t first = taken + incl - (uint32_t)nc;
                for (int i = 0; i < nc; i++) {
                    if (first + i >= 16) break;
                    const float cn[3] = { (float)cg[i].normal[0], (float)cg[i].normal[1], (float)cg[i].normal[2] };
                    const float ndot = dir[0] * cn[0] + dir[1] * cn[1] + dir[2] * cn[2];
                    if (ndot > -0.1f) continue;
                    const float backup = (float)(cg[i].depth / -ndot);
                    const float step_dist = t * delta_len;
                    float safe_dist = step_dist - backup;
                    if (safe_dist < 0) safe_dist = 0;
                    const float frac = safe_dist / delta_len;
                    const uint32_t order = first + i;
                    if (frac < step_frac) {                                   // within a lane: contacts in order, strict <
                        step_frac = frac; step_order = order;
                        step_normal[0] = cn[0]; step_normal[1] = cn[1]; step_normal[2] = cn[2];
                        step_hit = is_body ? (int32_t)id : -2 - (int32_t)id;
                    }
                }
                taken += __shfl(incl, WAVE - 1);
            }
            // the sequential loop keeps the FIRST contact (in order) among those with the smallest frac below best_frac
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float of = __shfl_xor(step_frac, o);
                const uint32_t oo = __shfl_xor(step_order, o);
                const float n0 = __shfl_xor(step_normal[0], o), n1 = __shfl_xor(step_normal[1], o), n2 = __shfl_xor(step_normal[2], o);
                const int32_t oh = __shfl_xor(step_hit, o);
                if (of < step_frac || (of == step_frac && oo < step_order)) {
                    step_frac = of; step_order = oo; step_normal[0] = n0; step_normal[1] = n1; step_normal[2] = n2; step_hit = oh;
                }
            }
            if (step_order != 0xffffffffu) {
                best_frac = step_frac;
                best_normal[0] = step_normal[0]; best_normal[1] = step_normal[1]; best_normal[2] = step_normal[2];
                best_hit = step_hit;
            }
            if (best_frac < t) break;
        }
    }
    if (lane == 0) {
        frac_out[sw] = best_frac;
        normal_out[3 * (size_t)sw] = best_normal[0];
        normal_out[3 * (size_t)sw + 1] = best_normal[1];
        normal_out[3 * (size_t)sw + 2] = best_normal[2];
        hit_out[sw] = best_hit;
    }
}

} // namespace clapgpu

using namespace clapgpu;

// ---------------------------------------------------------------------------------- host helpers
static void h_q_from_axis_and_angle(double (&q)[4], double ax, double ay, double az, double angle)
{
    double l = ax * ax + ay * ay + az * az;
    if (l > 0.0) {
        angle *= 0.5;
        q[0] = cos(angle);
        l = sin(angle) * (1.0 / sqrt(l));
        q[1] = ax * l; q[2] = ay * l; q[3] = az * l;
    } else {
        q[0] = 1; q[1] = q[2] = q[3] = 0;
    }
}

extern "C" void clapgpu_geom_offset_rotation(double R[12])
{
    double q[4], M[12];
    h_q_from_axis_and_angle(q, 1.0, 1.0, 1.0, -M_PI * 2.0 / 3.0);
    phd::q_to_R(q, M);
    memcpy(R, M, sizeof(M));
}

extern "C" void clapgpu_mass_sphere_total(double total_mass, double radius, double I[3])
{
    const double m1 = (4.0 / 3.0) * M_PI * radius * radius * radius * 1.0;      // dMassSetSphere(m, 1.0, r)
    const double II = 0.4 * m1 * radius * radius;
    const double scale = total_mass / m1;                                        // dMassAdjust
    I[0] = I[1] = I[2] = II * scale;
}

extern "C" void clapgpu_mass_capsule_total(double total_mass, int direction, double a, double b, double I[3])
{
    if (direction < 1 || direction > 3) direction = 3;
    const double M1 = M_PI * a * a * b * 1.0;
    const double M2 = (4.0 / 3.0) * M_PI * a * a * a * 1.0;
    const double m = M1 + M2;
    const double Ia = M1 * (0.25 * a * a + (1.0 / 12.0) * b * b) + M2 * (0.4 * a * a + 0.375 * a * b + 0.25 * b * b);
    const double Ib = (M1 * 0.5 + M2 * 0.4) * a * a;
    const double scale = total_mass / m;
    I[0] = I[1] = I[2] = Ia;
    I[direction - 1] = Ib;
    I[0] *= scale; I[1] *= scale; I[2] *= scale;
}

// physics.c:814-873
extern "C" void clapgpu_capsule_geom(float X, float Y, float Z, double geom_radius, double geom_offset,
                                     float *radius, float *length, float *yoffset, int *direction, float *ray_off)
{
    float r = 0.f, len = 0.f, off = 0.f, ro = 0.f;
    float mx = Y > Z ? Y : Z;                                                   // max3 / xmax3 (util.h:203-209)
    if (X > mx) mx = X;
    int w = 0;
    if (mx == Y) w = 1; else if (mx == Z) w = 2;
    const int dir = w + 1;
    if (dir == 3) {
        r = geom_radius ? (float)geom_radius : X / 2;
        len = Z - r * 2;
        off = geom_offset ? (float)geom_offset : (Y - r * 2) / 2;
        ro = r;
    } else {
        float mn = Y < Z ? Y : Z;
        if (X < mn) mn = X;
        r = geom_radius ? (float)geom_radius : mn / 2;
        const float l = Y / 2 - r * 2;
        len = l > 0 ? l : 0;
        off = geom_offset ? (float)geom_offset : Y / 2;
        ro = r + len / 2;
    }
    *radius = r; *length = len; *yoffset = off; *direction = dir; *ray_off = ro;
}

static int check_bodies2(const clapgpu_bodies *b)
{
    if (!b || !b->pos || !b->quat || !b->lvel || !b->avel || !b->mass || !b->radius || !b->bflags ||
        !b->adis_steps_left || !b->adis_time_left)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (b->adis_average_samples > 1 && (!b->adis_samples || !b->adis_counter))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    return CLAPGPU_OK;
}

static BodiesK bodies_k(const clapgpu_bodies *b)
{
    BodiesK k;
    k.n = b->n; k.samples = b->adis_average_samples;
    k.pos = b->pos; k.quat = b->quat; k.lvel = b->lvel; k.avel = b->avel;
    k.mass = b->mass; k.radius = b->radius; k.bflags = b->bflags;
    k.adis_steps_left = b->adis_steps_left; k.adis_time_left = b->adis_time_left;
    k.length = b->length; k.inertia = b->inertia;
    memcpy(k.Roff, b->geom_offset_R, sizeof(k.Roff));
    bool zero = true;
    for (int i = 0; i < 12; i++) zero &= k.Roff[i] == 0.0;
    if (zero) k.Roff[0] = k.Roff[5] = k.Roff[10] = 1.0;                          // unset = no offset rotation
    k.aabb = b->aabb; k.axis = b->axis; k.adis_samples = b->adis_samples; k.adis_counter = b->adis_counter;
    return k;
}

extern "C" int clapgpu_bodies_aabb(void *stream, const clapgpu_bodies *b)
{
    int rc = check_bodies2(b);
    if (rc) return rc;
    if (b->n == 0) return CLAPGPU_OK;
    hipLaunchKernelGGL(k_bodies_aabb, dim3((b->n + PB - 1) / PB), dim3(PB), 0, as_stream(stream), bodies_k(b));
    CLAPGPU_LAUNCH_CHECK("k_bodies_aabb");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_bodies_step(void *stream, const clapgpu_bodies *b, const clapgpu_world *w, double h)
{
    int rc = check_bodies2(b);
    if (rc) return rc;
    if (!w) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (b->n == 0) return CLAPGPU_OK;
    WorldK2 wk;
    memcpy(&wk, w, sizeof(wk));
    hipLaunchKernelGGL(k_bodies_step, dim3((b->n + PB - 1) / PB), dim3(PB), 0, as_stream(stream), bodies_k(b), wk, h);
    CLAPGPU_LAUNCH_CHECK("k_bodies_step");
    return CLAPGPU_OK;
}

// ---------------------------------------------------------------------------------- broadphase object
struct clapgpu_bp {
    uint32_t n_max, buckets, n_static, n_large, n_tiles;
    double cell;
    void *dev;                     // one allocation
    BpK k;                         // device pointers filled in
    uint32_t arena_cap, sarena_cap;
};

static uint32_t buckets_for(uint32_t n)
{
    uint32_t b = 1024;
    while (b < n / 16 && b < (1u << 22)) b <<= 1;
    return b;
}

static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int clapgpu_bp_create(clapgpu_bp **out, uint32_t n_max, double cell, uint32_t n_static, const double *static_aabb)
{
    if (!out || !(cell > 0.0) || (n_static && !static_aabb) || n_max > (1u << 27))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    clapgpu_bp *bp = static_cast<clapgpu_bp *>(calloc(1, sizeof(*bp)));
    if (!bp) return CLAPGPU_ERR_NOMEM;
    const uint32_t n = n_max ? n_max : 1, nb = buckets_for(n);
    bp->n_max = n_max; bp->buckets = nb; bp->cell = cell; bp->n_static = n_static;
    bp->n_tiles = (n + BP_EMIT_TILE - 1) / BP_EMIT_TILE;

    // statics: every block whose own bodies could touch the static (its AABB grown by half a cell), as a CSR over
    // the same buckets; statics that would enter more than 64 blocks go to the large list
    std::vector<uint32_t> s_count(nb + 1, 0), s_entries, s_large;
    std::vector<std::pair<uint32_t, uint32_t>> ins;                               // (bucket, static)
    const double grow = cell * 0.5 * (1.0 + 1e-9);
    for (uint32_t s = 0; s < n_static; s++) {
        const double *bb = static_aabb + 6 * (size_t)s;
        int32_t lo[3], hi[3];
        bool large = false;
        unsigned long long blocks = 1;
        for (int a = 0; a < 3; a++) {
            lo[a] = cell_coord(bb[2 * a] - grow, cell) >> 2;
            hi[a] = cell_coord(bb[2 * a + 1] + grow, cell) >> 2;
            if (!(bb[2 * a] <= bb[2 * a + 1])) large = true;                      // NaN / inverted: keep it in the tested-by-all list
            blocks *= (unsigned long long)(hi[a] - lo[a] + 1);
            if (blocks > 64) large = true;
        }
        if (large) { s_large.push_back(s); continue; }
        const size_t first = ins.size();
        for (int32_t z = lo[2]; z <= hi[2]; z++)
            for (int32_t y = lo[1]; y <= hi[1]; y++)
                for (int32_t x = lo[0]; x <= hi[0]; x++) {
                    const uint32_t h = block_hash(x, y, z, nb - 1);
                    bool dup = false;
                    for (size_t e = first; e < ins.size(); e++) dup |= ins[e].first == h;
                    if (!dup) ins.push_back({ h, s });
                }
    }
    for (auto &e : ins) s_count[e.first + 1]++;
    for (uint32_t b = 0; b < nb; b++) s_count[b + 1] += s_count[b];
    s_entries.resize(ins.size() ? ins.size() : 1);
    {
        std::vector<uint32_t> cur(s_count.begin(), s_count.end() - 1);
        for (auto &e : ins) s_entries[cur[e.first]++] = e.second;                 // ascending static index inside a bucket
    }
    bp->n_large = (uint32_t)s_large.size();
    if (s_large.empty()) s_large.push_back(0);

    // one device allocation, carved
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t o_bcnt = take(sizeof(unsigned long long) * nb), o_bstart = take(4 * ((size_t)nb + 1)), o_bown = take(4 * (size_t)nb);
    const size_t o_ranks = take(32 * (size_t)n), o_entries = take(4 * 8 * (size_t)n);
    const size_t o_cnt = take(4 * (size_t)n), o_scnt = take(4 * (size_t)n), o_ref = take(4 * (size_t)n), o_sref = take(4 * (size_t)n);
    const size_t o_part = take(4 * (size_t)BP_LIST * n), o_spart = take(4 * (size_t)BP_LIST * n);
    const size_t o_tsum = take(4 * (size_t)bp->n_tiles), o_stsum = take(4 * (size_t)bp->n_tiles);
    const size_t o_toff = take(4 * (size_t)bp->n_tiles), o_stoff = take(4 * (size_t)bp->n_tiles);
    const size_t o_ctrl = take(64);
    const size_t o_sstart = take(4 * ((size_t)nb + 1)), o_sent = take(4 * s_entries.size()), o_slarge = take(4 * s_large.size());
    const size_t o_saabb = take(48 * (size_t)(n_static ? n_static : 1));
    const size_t fixed = off;
    if (hipMalloc(&bp->dev, fixed) != hipSuccess) {
        (void)hipGetLastError();
        free(bp);
        return CLAPGPU_ERR_NOMEM;
    }
    char *d = static_cast<char *>(bp->dev);
    if (hipMemset(d, 0, fixed) != hipSuccess ||
        hipMemcpy(d + o_sstart, s_count.data(), 4 * ((size_t)nb + 1), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + o_sent, s_entries.data(), 4 * s_entries.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + o_slarge, s_large.data(), 4 * s_large.size(), hipMemcpyHostToDevice) != hipSuccess ||
        (n_static && hipMemcpy(d + o_saabb, static_aabb, 48 * (size_t)n_static, hipMemcpyHostToDevice) != hipSuccess)) {
        (void)hipGetLastError();
        (void)hipFree(bp->dev);
        free(bp);
        return CLAPGPU_ERR_UNKNOWN;
    }
    BpK &k = bp->k;
    memset(&k, 0, sizeof(k));
    k.cell = cell; k.mask = nb - 1;
    k.bucket_cnt = reinterpret_cast<unsigned long long *>(d + o_bcnt);
    k.bucket_start = reinterpret_cast<uint32_t *>(d + o_bstart); k.bucket_own = reinterpret_cast<uint32_t *>(d + o_bown);
    k.ranks = reinterpret_cast<uint32_t *>(d + o_ranks); k.entries = reinterpret_cast<uint32_t *>(d + o_entries);
    k.cnt = reinterpret_cast<uint32_t *>(d + o_cnt); k.scnt = reinterpret_cast<uint32_t *>(d + o_scnt);
    k.ref = reinterpret_cast<uint32_t *>(d + o_ref); k.sref = reinterpret_cast<uint32_t *>(d + o_sref);
    k.partners = reinterpret_cast<uint32_t *>(d + o_part); k.spartners = reinterpret_cast<uint32_t *>(d + o_spart);
    k.tile_sum = reinterpret_cast<uint32_t *>(d + o_tsum); k.stile_sum = reinterpret_cast<uint32_t *>(d + o_stsum);
    k.tile_off = reinterpret_cast<uint32_t *>(d + o_toff); k.stile_off = reinterpret_cast<uint32_t *>(d + o_stoff);
    k.ctrl = reinterpret_cast<uint32_t *>(d + o_ctrl);
    k.s_start = reinterpret_cast<const uint32_t *>(d + o_sstart); k.s_entries = reinterpret_cast<const uint32_t *>(d + o_sent);
    k.s_large = reinterpret_cast<const uint32_t *>(d + o_slarge); k.s_aabb = reinterpret_cast<const double *>(d + o_saabb);
    k.n_large = bp->n_large; k.n_static = n_static;
    *out = bp;
    return CLAPGPU_OK;
}

extern "C" void clapgpu_bp_destroy(clapgpu_bp *bp)
{
    if (!bp) return;
    if (bp->dev) (void)hipFree(bp->dev);
    if (bp->k.arena) (void)hipFree(bp->k.arena);
    if (bp->k.sarena) (void)hipFree(bp->k.sarena);
    free(bp);
}

extern "C" const double *clapgpu_bp_static_aabb(const clapgpu_bp *bp) { return bp ? bp->k.s_aabb : nullptr; }

extern "C" int clapgpu_bp_collide(void *stream, clapgpu_bp *bp, uint32_t n, const double *aabb,
                                  uint32_t *pairs, uint32_t capacity, uint32_t *pair_total,
                                  uint32_t *static_pairs, uint32_t static_capacity, uint32_t *static_pair_total)
{
    if (!bp || !pair_total || (n && !aabb) || (capacity && !pairs) || (static_capacity && !static_pairs))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (n > bp->n_max) return CLAPGPU_ERR_TOO_LARGE;
    hipStream_t s = as_stream(stream);
    const bool statics = bp->n_static && static_pair_total;
    if (n == 0) {
        CLAPGPU_HIP(hipMemsetAsync(pair_total, 0, sizeof(uint32_t), s));
        if (static_pair_total) CLAPGPU_HIP(hipMemsetAsync(static_pair_total, 0, sizeof(uint32_t), s));
        return CLAPGPU_OK;
    }
    // the arenas hold the lists longer than BP_LIST: never more entries than the caller's capacity
    if (capacity > bp->arena_cap) {
        if (bp->k.arena) CLAPGPU_HIP(hipFree(bp->k.arena));
        bp->k.arena = nullptr;
        CLAPGPU_HIP(hipMalloc(reinterpret_cast<void **>(&bp->k.arena), 4 * (size_t)capacity));
        bp->arena_cap = capacity;
    }
    if (statics && static_capacity > bp->sarena_cap) {
        if (bp->k.sarena) CLAPGPU_HIP(hipFree(bp->k.sarena));
        bp->k.sarena = nullptr;
        CLAPGPU_HIP(hipMalloc(reinterpret_cast<void **>(&bp->k.sarena), 4 * (size_t)static_capacity));
        bp->sarena_cap = static_capacity;
    }
    BpK k = bp->k;
    k.n = n; k.aabb = aabb; k.n_tiles = (n + BP_EMIT_TILE - 1) / BP_EMIT_TILE;
    k.pairs = pairs; k.capacity = capacity; k.pair_total = pair_total;
    k.spairs = static_pairs; k.scapacity = static_capacity; k.spair_total = static_pair_total;
    if (!statics) { k.n_static = 0; k.n_large = 0; if (static_pair_total) CLAPGPU_HIP(hipMemsetAsync(static_pair_total, 0, 4, s)); }
    hipLaunchKernelGGL(k_bp_bin, dim3((n + BIN_BLOCK - 1) / BIN_BLOCK), dim3(BIN_BLOCK), 0, s, k);
    CLAPGPU_LAUNCH_CHECK("k_bp_bin");
    hipLaunchKernelGGL(k_bp_scatter, dim3((n + PB - 1) / PB), dim3(PB), 0, s, k);
    CLAPGPU_LAUNCH_CHECK("k_bp_scatter");
    hipLaunchKernelGGL(k_bp_search, dim3((bp->buckets + PB / WAVE - 1) / (PB / WAVE)), dim3(PB), 0, s, k);
    CLAPGPU_LAUNCH_CHECK("k_bp_search");
    hipLaunchKernelGGL(k_bp_emit, dim3(k.n_tiles), dim3(BP_EMIT_TILE), 0, s, k);
    CLAPGPU_LAUNCH_CHECK("k_bp_emit");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_bp_status(void *stream, clapgpu_bp *bp, uint32_t *status)
{
    if (!bp || !status) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    CLAPGPU_HIP(hipMemcpyAsync(status, bp->k.ctrl + CTRL_STATUS, sizeof(uint32_t), hipMemcpyDeviceToHost, as_stream(stream)));
    CLAPGPU_HIP(hipStreamSynchronize(as_stream(stream)));
    return CLAPGPU_OK;
}

static GeomsK geoms_k(const clapgpu_geoms *g)
{
    GeomsK k;
    k.n = g->n; k.pos = g->pos; k.axis = g->axis; k.radius = g->radius; k.length = g->length; k.aabb = g->aabb;
    k.material = g->material; k.kind = g->kind;
    return k;
}

extern "C" int clapgpu_contacts_geoms(void *stream, const clapgpu_geoms *A, const clapgpu_geoms *B, const uint32_t *pairs,
                                      const uint32_t *pair_total, uint32_t capacity, clapgpu_contact2 *contacts,
                                      uint32_t *contact_total, uint32_t *body_flags_a, uint32_t *body_flags_b)
{
    if (!A || !B || !pair_total || (capacity && (!pairs || !contacts)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    hipStream_t s = as_stream(stream);
    if (contact_total)
        CLAPGPU_HIP(hipMemsetAsync(contact_total, 0, sizeof(uint32_t), s));
    if (capacity == 0 || A->n == 0 || B->n == 0)
        return CLAPGPU_OK;
    const uint32_t blocks = (capacity + PB - 1) / PB;
    hipLaunchKernelGGL(k_contacts_geoms, dim3(blocks < 2048 ? blocks : 2048), dim3(PB), 0, s, geoms_k(A), geoms_k(B),
                       reinterpret_cast<const uint2 *>(pairs), pair_total, capacity, contacts, contact_total, body_flags_a,
                       body_flags_b);
    CLAPGPU_LAUNCH_CHECK("k_contacts_geoms");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_sweep_capsules(void *stream, const clapgpu_geoms *A, const clapgpu_geoms *B, uint32_t n_sweeps,
                                      const uint32_t *sweep_body, const float *delta, const uint32_t *cand_first,
                                      const uint32_t *cand, float *frac, float *normal, int32_t *hit)
{
    if (!A || !B || (n_sweeps && (!sweep_body || !delta || !cand_first || !frac || !normal || !hit)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (n_sweeps == 0) return CLAPGPU_OK;
    hipLaunchKernelGGL(k_sweep_capsules, dim3((n_sweeps + PB / WAVE - 1) / (PB / WAVE)), dim3(PB), 0, as_stream(stream),
                       geoms_k(A), geoms_k(B), n_sweeps, sweep_body, delta, cand_first, cand, frac, normal, hit);
    CLAPGPU_LAUNCH_CHECK("k_sweep_capsules");
    return CLAPGPU_OK;
}
