// physics.hip -- rigid-body integrate, body->entity read-back and AABB broadphase for gfx950.
//
// Replaces, for free sphere bodies, what phys_step() (physics.c:773-787) does per fixed substep
// through ODE: the two broadphase calls of __phys_step() (dSpaceCollide2(ground, bodies),
// dSpaceCollide(bodies), physics.c:751-753) as candidate-pair lists, dWorldQuickStep()'s
// integrate for bodies without joints (physics.c:769), and phys_body_update() (physics.c:789-812).
// ODE itself is an absent submodule: the arithmetic follows ODE's published quickstep / dxStepBody
// (see oracle/physics.c for the statement and its limits) -- PARITY UNPINNED.
//
// fp64 throughout (the reference builds ODE with dDOUBLE, physics.h:5-9).  HBM-bound:
// 232 B / body for the integrate.  Broadphase: spatial hash of cell >= the largest
// AABB edge, bucket lists built by counting (histogram -> scan -> scatter), one lane per body
// probing its 27 neighbour cells; pairs come out as the canonical ascending (i, j) list, whatever
// order the atomics filled the buckets in.  Static geoms are streamed through LDS tiles.
#include <string.h>
#include "common.h"

namespace clapgpu {

constexpr int PHYS_BLOCK = 256;

struct WorldK {
    double  gravity[3];
    double  linear_damping;
    double  linear_damping_threshold_sq;
    double  adis_linear_threshold_sq;
    double  adis_angular_threshold_sq;
    double  adis_time;
    int32_t adis_steps;
    int32_t pad;
};
static_assert(sizeof(WorldK) == sizeof(clapgpu_world), "world layout");

__global__ __launch_bounds__(PHYS_BLOCK)
void k_bodies_step(uint32_t n, double h, WorldK w, double *pos, double *quat, double *lvel, double *avel,
                   const double *mass, uint32_t *bflags, int32_t *adis_steps_left, double *adis_time_left)
{
    const uint32_t i = blockIdx.x * PHYS_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    uint32_t fl = bflags[i];
    if (fl & CLAPGPU_BODY_DISABLED)
        return;
    double *p = pos + 3 * (size_t)i, *q = quat + 4 * (size_t)i, *v = lvel + 3 * (size_t)i, *om = avel + 3 * (size_t)i;
    double vx = v[0], vy = v[1], vz = v[2];
    const double ox = om[0], oy = om[1], oz = om[2];

    if (fl & CLAPGPU_BODY_AUTO_DISABLE) {                          // dInternalHandleAutoDisabling
        bool idle = true;
        if (vx * vx + vy * vy + vz * vz > w.adis_linear_threshold_sq)
            idle = false;
        else if (ox * ox + oy * oy + oz * oz > w.adis_angular_threshold_sq)
            idle = false;
        int32_t sl = adis_steps_left[i];
        double tl = adis_time_left[i];
        if (idle) { sl--; tl -= h; } else { sl = w.adis_steps; tl = w.adis_time; }
        adis_steps_left[i] = sl;
        adis_time_left[i] = tl;
        if (sl <= 0 && tl <= 0) {
            bflags[i] = fl | CLAPGPU_BODY_DISABLED;
            v[0] = v[1] = v[2] = 0;
            om[0] = om[1] = om[2] = 0;
            return;
        }
    }
    const double m = mass[i];
    const double k = h * (1.0 / m);
    const bool grav = !(fl & CLAPGPU_BODY_NO_GRAVITY);
    vx += k * (grav ? m * w.gravity[0] : 0.0);
    vy += k * (grav ? m * w.gravity[1] : 0.0);
    vz += k * (grav ? m * w.gravity[2] : 0.0);
    p[0] += h * vx;                                                 // dxStepBody
    p[1] += h * vy;
    p[2] += h * vz;
    double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const double d0 = 0.5 * (-ox * q1 - oy * q2 - oz * q3);         // dWtoDQ
    const double d1 = 0.5 * ( ox * q0 + oy * q3 - oz * q2);
    const double d2 = 0.5 * (-ox * q3 + oy * q0 + oz * q1);
    const double d3 = 0.5 * ( ox * q2 - oy * q1 + oz * q0);
    q0 += h * d0; q1 += h * d1; q2 += h * d2; q3 += h * d3;
    double l = q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;               // dNormalize4
    if (l > 0) {
        l = 1.0 / sqrt(l);
        q0 *= l; q1 *= l; q2 *= l; q3 *= l;
    } else {
        q0 = 1; q1 = q2 = q3 = 0;
    }
    q[0] = q0; q[1] = q1; q[2] = q2; q[3] = q3;
    if (w.linear_damping != 0.0) {
        const double speed2 = vx * vx + vy * vy + vz * vz;
        if (speed2 > w.linear_damping_threshold_sq) {
            const double s = 1 - w.linear_damping;
            vx *= s; vy *= s; vz *= s;
        }
    }
    v[0] = vx; v[1] = vy; v[2] = vz;
}

// phys_body_update (physics.c:789-812): scatter body pose into the entity SoA, mark it dirty
__global__ __launch_bounds__(PHYS_BLOCK)
void k_phys_body_update(uint32_t n, const double *pos, const double *quat, const double *lvel,
                        const double *yoffset, const int32_t *body_entity,
                        float *pos_scale, float *rot, uint32_t *entity_flags, uint8_t *moving)
{
    const uint32_t i = blockIdx.x * PHYS_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    const int32_t e = body_entity[i];
    const double *p = pos + 3 * (size_t)i, *q = quat + 4 * (size_t)i, *v = lvel + 3 * (size_t)i;
    if (e >= 0) {
        pos_scale[4 * (size_t)e + 0] = (float)p[0];
        pos_scale[4 * (size_t)e + 1] = (float)(p[1] - yoffset[i]);
        pos_scale[4 * (size_t)e + 2] = (float)p[2];
        reinterpret_cast<float4 *>(rot)[e] = make_float4((float)q[1], (float)q[2], (float)q[3], (float)q[0]);
        atomicOr(&entity_flags[e], CLAPGPU_E_DIRTY);
    }
    if (moving)
        moving[i] = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) > 1e-3 ? 1 : 0;
}

// ---------------------------------------------------------------- exclusive scan (uint32)
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = PHYS_BLOCK * SCAN_ITEMS;                  // 2048 values per block

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t &block_total, uint32_t *lds)
{
    const int lane = lane_id(), wave = threadIdx.x / WAVE;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == WAVE - 1) lds[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < PHYS_BLOCK / WAVE; k++) {
        const uint32_t s = lds[k];
        if (k < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    block_total = tot;
    return wave_off + incl - v;
}

__global__ __launch_bounds__(PHYS_BLOCK)
void k_scan_block_sums(const uint32_t *in, uint32_t n, uint32_t *block_sums)
{
    __shared__ uint32_t lds[PHYS_BLOCK / WAVE];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < n) s += in[base + k];
    uint32_t tot;
    block_exclusive_scan(s, tot, lds);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// one block: exclusive scan of the block sums in place, grand total to *total
__global__ __launch_bounds__(PHYS_BLOCK)
void k_scan_sums(uint32_t *block_sums, uint32_t n_blocks, uint32_t *total)
{
    __shared__ uint32_t lds[PHYS_BLOCK / WAVE];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_blocks; base += PHYS_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sums[i] : 0;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan(v, tot, lds);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(PHYS_BLOCK)
void k_scan_apply(const uint32_t *in, uint32_t n, const uint32_t *block_sums, uint32_t *out)
{
    __shared__ uint32_t lds[PHYS_BLOCK / WAVE];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = base + k < n ? in[base + k] : 0;
        s += v[k];
    }
    uint32_t tot;
    uint32_t run = block_sums[blockIdx.x] + block_exclusive_scan(s, tot, lds);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
}

// out[i] = sum of in[0..i); *total = sum of all.  scratch: ceil(n / 2048) uint32.  in may equal out.
static int exclusive_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total,
                              uint32_t *scratch)
{
    const uint32_t blocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_block_sums, dim3(blocks), dim3(PHYS_BLOCK), 0, s, in, n, scratch);
    CLAPGPU_LAUNCH_CHECK("k_scan_block_sums");
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(PHYS_BLOCK), 0, s, scratch, blocks, total);
    CLAPGPU_LAUNCH_CHECK("k_scan_sums");
    hipLaunchKernelGGL(k_scan_apply, dim3(blocks), dim3(PHYS_BLOCK), 0, s, in, n, scratch, out);
    CLAPGPU_LAUNCH_CHECK("k_scan_apply");
    return CLAPGPU_OK;
}

// ---------------------------------------------------------------- broadphase
struct BpK {
    uint32_t      n;
    const double *pos;
    const double *radius;
    double        cell;
    uint32_t      hash_mask;          // buckets - 1 (power of two)
    uint32_t     *bucket_count;       // [buckets]   -> becomes bucket_start after the scan
    uint32_t     *bucket_cursor;      // [buckets]
    uint32_t     *bucket_items;       // [n]
    uint32_t     *pair_count;         // [n]         -> becomes pair_start after the scan
    uint32_t     *pairs;              // [2 * capacity]
    uint32_t      capacity;
};

__device__ __forceinline__ void cell_of(const double *p, double cell, int32_t &cx, int32_t &cy, int32_t &cz)
{
    cx = (int32_t)floor(p[0] / cell);
    cy = (int32_t)floor(p[1] / cell);
    cz = (int32_t)floor(p[2] / cell);
}

__device__ __forceinline__ uint32_t cell_hash(int32_t cx, int32_t cy, int32_t cz, uint32_t mask)
{
    return (((uint32_t)cx * 73856093u) ^ ((uint32_t)cy * 19349663u) ^ ((uint32_t)cz * 83492791u)) & mask;
}

__global__ __launch_bounds__(PHYS_BLOCK)
void k_bp_histogram(BpK k)
{
    const uint32_t i = blockIdx.x * PHYS_BLOCK + threadIdx.x;
    if (i >= k.n) return;
    int32_t cx, cy, cz;
    cell_of(k.pos + 3 * (size_t)i, k.cell, cx, cy, cz);
    atomicAdd(&k.bucket_count[cell_hash(cx, cy, cz, k.hash_mask)], 1u);
}

__global__ __launch_bounds__(PHYS_BLOCK)
void k_bp_scatter(BpK k)
{
    const uint32_t i = blockIdx.x * PHYS_BLOCK + threadIdx.x;
    if (i >= k.n) return;
    int32_t cx, cy, cz;
    cell_of(k.pos + 3 * (size_t)i, k.cell, cx, cy, cz);
    const uint32_t h = cell_hash(cx, cy, cz, k.hash_mask);
    const uint32_t slot = k.bucket_count[h] + atomicAdd(&k.bucket_cursor[h], 1u);   // bucket_count holds starts now
    if (slot < k.n)
        k.bucket_items[slot] = i;
}

// collideAABBs: disjoint iff separated on an axis (touching boxes collide)
__device__ __forceinline__ bool spheres_aabb_overlap(const double *pa, double ra, const double *pb, double rb)
{
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double alo = pa[a] - ra, ahi = pa[a] + ra, blo = pb[a] - rb, bhi = pb[a] + rb;
        if (alo > bhi || ahi < blo) return false;
    }
    return true;
}

// EMIT = false: count the partners j > i of body i; EMIT = true: write them, ascending, at pair_start[i]
template <bool EMIT>
__global__ __launch_bounds__(PHYS_BLOCK)
void k_bp_pairs(BpK k, const uint32_t *bucket_start_end /* [buckets + 1] */)
{
    const uint32_t i = blockIdx.x * PHYS_BLOCK + threadIdx.x;
    if (i >= k.n) return;
    const double *pi = k.pos + 3 * (size_t)i;
    const double ri = k.radius[i];
    int32_t cx, cy, cz;
    cell_of(pi, k.cell, cx, cy, cz);
    const uint32_t start = EMIT ? k.pair_count[i] : 0;              // pair_count holds starts when emitting
    uint32_t cnt = 0;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int32_t nx = cx + dx, ny = cy + dy, nz = cz + dz;
                const uint32_t h = cell_hash(nx, ny, nz, k.hash_mask);
                const uint32_t b0 = bucket_start_end[h], b1 = bucket_start_end[h + 1];
                for (uint32_t s = b0; s < b1 && s < k.n; s++) {
                    const uint32_t j = k.bucket_items[s];
                    if (j <= i || j >= k.n) continue;
                    const double *pj = k.pos + 3 * (size_t)j;
                    int32_t jx, jy, jz;
                    cell_of(pj, k.cell, jx, jy, jz);
                    if (jx != nx || jy != ny || jz != nz) continue;  // another cell sharing the bucket
                    if (!spheres_aabb_overlap(pi, ri, pj, k.radius[j])) continue;
                    if (EMIT) {
                        const uint32_t o = start + cnt;
                        if (o < k.capacity) { k.pairs[2 * (size_t)o] = i; k.pairs[2 * (size_t)o + 1] = j; }
                    }
                    cnt++;
                }
            }
    if (!EMIT) {
        k.pair_count[i] = cnt;
    } else {
        // ascending j: insertion sort of this body's own short run (cnt is a handful)
        const uint32_t lim = start + cnt <= k.capacity ? cnt : (start < k.capacity ? k.capacity - start : 0);
        for (uint32_t a = 1; a < lim; a++) {
            const uint32_t key = k.pairs[2 * (size_t)(start + a) + 1];
            uint32_t b = a;
            while (b > 0 && k.pairs[2 * (size_t)(start + b - 1) + 1] > key) {
                k.pairs[2 * (size_t)(start + b) + 1] = k.pairs[2 * (size_t)(start + b - 1) + 1];
                b--;
            }
            k.pairs[2 * (size_t)(start + b) + 1] = key;
        }
    }
}

// bodies x static geoms: statics streamed through LDS tiles; pairs (body, static) ascending
constexpr int STATIC_TILE = 256;
template <bool EMIT>
__global__ __launch_bounds__(PHYS_BLOCK)
void k_bp_static(uint32_t n, const double *pos, const double *radius, uint32_t n_static, const double *static_aabb,
                 uint32_t *pair_count, uint32_t *pairs, uint32_t capacity)
{
    __shared__ double tile[STATIC_TILE * 6];
    const uint32_t i = blockIdx.x * PHYS_BLOCK + threadIdx.x;
    const bool live = i < n;
    double bb[6] = { 0, 0, 0, 0, 0, 0 };
    if (live) {
        const double r = radius[i];
#pragma unroll
        for (int a = 0; a < 3; a++) { bb[2 * a] = pos[3 * (size_t)i + a] - r; bb[2 * a + 1] = pos[3 * (size_t)i + a] + r; }
    }
    const uint32_t start = (EMIT && live) ? pair_count[i] : 0;
    uint32_t cnt = 0;
    for (uint32_t base = 0; base < n_static; base += STATIC_TILE) {
        const uint32_t m = n_static - base < STATIC_TILE ? n_static - base : STATIC_TILE;
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < m * 6; q += PHYS_BLOCK)
            tile[q] = static_aabb[(size_t)base * 6 + q];
        __syncthreads();
        if (live)
            for (uint32_t s = 0; s < m; s++) {
                const double *sb = tile + 6 * s;
                if (bb[0] > sb[1] || bb[1] < sb[0] || bb[2] > sb[3] || bb[3] < sb[2] || bb[4] > sb[5] || bb[5] < sb[4])
                    continue;
                if (EMIT) {
                    const uint32_t o = start + cnt;
                    if (o < capacity) { pairs[2 * (size_t)o] = i; pairs[2 * (size_t)o + 1] = base + s; }
                }
                cnt++;
            }
    }
    if (!EMIT && live)
        pair_count[i] = cnt;
}

} // namespace clapgpu

using namespace clapgpu;

// physics.c:773-787 (host)
extern "C" int clapgpu_phys_step_schedule(double *time_acc, double dt)
{
    const double fixed_dt = 1.0 / 120.0;
    int steps = 0;
    const int max_steps = 5;
    *time_acc += dt;
    for (; *time_acc >= fixed_dt && steps < max_steps; *time_acc -= fixed_dt, steps++)
        ;
    if (steps == max_steps)
        *time_acc = 0.0;
    return steps;
}

extern "C" void clapgpu_world_defaults(clapgpu_world *w)
{
    memset(w, 0, sizeof(*w));
    w->gravity[1] = -9.8;                          // physics.c:1125
    w->linear_damping = 0.001;                     // physics.c:1129
    w->linear_damping_threshold_sq = 0.01 * 0.01;  // ODE default damping threshold
    w->adis_linear_threshold_sq = 0.05 * 0.05;     // physics.c:1040
    w->adis_angular_threshold_sq = 0.05 * 0.05;    // physics.c:1041
    w->adis_steps = 30;                            // physics.c:1042
    w->adis_time = 0.0;
}

static int check_bodies(const clapgpu_bodies *b)
{
    if (!b || !b->pos || !b->quat || !b->lvel || !b->avel || !b->mass || !b->radius || !b->bflags ||
        !b->adis_steps_left || !b->adis_time_left)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    return CLAPGPU_OK;
}

extern "C" int clapgpu_bodies_step(void *stream, const clapgpu_bodies *b, const clapgpu_world *w, double h)
{
    int rc = check_bodies(b);
    if (rc) return rc;
    if (!w) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (b->n == 0) return CLAPGPU_OK;
    WorldK wk;
    memcpy(&wk, w, sizeof(wk));
    hipLaunchKernelGGL(k_bodies_step, dim3((b->n + PHYS_BLOCK - 1) / PHYS_BLOCK), dim3(PHYS_BLOCK), 0,
                       as_stream(stream), b->n, h, wk, b->pos, b->quat, b->lvel, b->avel, b->mass, b->bflags,
                       b->adis_steps_left, b->adis_time_left);
    CLAPGPU_LAUNCH_CHECK("k_bodies_step");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_phys_body_update(void *stream, const clapgpu_bodies *b, float *pos_scale, float *rot,
                                        uint32_t *entity_flags, uint8_t *moving)
{
    int rc = check_bodies(b);
    if (rc) return rc;
    if (!b->yoffset || !b->body_entity || !pos_scale || !rot || !entity_flags)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (b->n == 0) return CLAPGPU_OK;
    hipLaunchKernelGGL(k_phys_body_update, dim3((b->n + PHYS_BLOCK - 1) / PHYS_BLOCK), dim3(PHYS_BLOCK), 0,
                       as_stream(stream), b->n, b->pos, b->quat, b->lvel, b->yoffset, b->body_entity, pos_scale,
                       rot, entity_flags, moving);
    CLAPGPU_LAUNCH_CHECK("k_phys_body_update");
    return CLAPGPU_OK;
}

static uint32_t bucket_count_for(uint32_t n)
{
    uint32_t b = 1024;
    while (b < 2 * n && b < (1u << 26)) b <<= 1;
    return b;
}

// layout of the broadphase work space (uint32 units)
struct BpScratch { uint32_t *bucket, *cursor, *items, *pcount, *scan, *total; };

static size_t bp_scratch_words(uint32_t n)
{
    const size_t buckets = bucket_count_for(n);
    const size_t scan = (buckets + 1 + SCAN_TILE - 1) / SCAN_TILE + ((size_t)n + SCAN_TILE) / SCAN_TILE + 8;
    return (buckets + 1) + buckets + n + (n + 1) + scan + 8;
}

extern "C" size_t clapgpu_broadphase_scratch_bytes(uint32_t n)
{
    return bp_scratch_words(n) * sizeof(uint32_t);
}

static BpScratch carve(void *scratch, uint32_t n)
{
    const size_t buckets = bucket_count_for(n);
    BpScratch s;
    uint32_t *p = static_cast<uint32_t *>(scratch);
    s.bucket = p;  p += buckets + 1;
    s.cursor = p;  p += buckets;
    s.items = p;   p += n;
    s.pcount = p;  p += n + 1;
    s.total = p;   p += 8;
    s.scan = p;
    return s;
}

extern "C" int clapgpu_broadphase_pairs(void *stream, const clapgpu_bodies *b, double cell,
                                        uint32_t *pairs, uint32_t capacity, uint32_t *pair_total, void *scratch)
{
    int rc = check_bodies(b);
    if (rc) return rc;
    if (!pair_total || !scratch || (capacity && !pairs) || !(cell > 0.0))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    hipStream_t s = as_stream(stream);
    if (b->n == 0) {
        CLAPGPU_HIP(hipMemsetAsync(pair_total, 0, sizeof(uint32_t), s));
        return CLAPGPU_OK;
    }
    const uint32_t n = b->n, buckets = bucket_count_for(n);
    BpScratch sc = carve(scratch, n);
    CLAPGPU_HIP(hipMemsetAsync(sc.bucket, 0, ((size_t)buckets + 1 + buckets) * sizeof(uint32_t), s));   // counts + cursors

    BpK k;
    k.n = n; k.pos = b->pos; k.radius = b->radius; k.cell = cell; k.hash_mask = buckets - 1;
    k.bucket_count = sc.bucket; k.bucket_cursor = sc.cursor; k.bucket_items = sc.items;
    k.pair_count = sc.pcount; k.pairs = pairs; k.capacity = capacity;
    const dim3 grid((n + PHYS_BLOCK - 1) / PHYS_BLOCK), block(PHYS_BLOCK);

    hipLaunchKernelGGL(k_bp_histogram, grid, block, 0, s, k);
    CLAPGPU_LAUNCH_CHECK("k_bp_histogram");
    rc = exclusive_scan_u32(s, sc.bucket, sc.bucket, buckets + 1, sc.total, sc.scan);   // starts; [buckets] = n
    if (rc) return rc;
    hipLaunchKernelGGL(k_bp_scatter, grid, block, 0, s, k);
    CLAPGPU_LAUNCH_CHECK("k_bp_scatter");
    hipLaunchKernelGGL(k_bp_pairs<false>, grid, block, 0, s, k, sc.bucket);
    CLAPGPU_LAUNCH_CHECK("k_bp_pairs<count>");
    rc = exclusive_scan_u32(s, sc.pcount, sc.pcount, n, pair_total, sc.scan);
    if (rc) return rc;
    hipLaunchKernelGGL(k_bp_pairs<true>, grid, block, 0, s, k, sc.bucket);
    CLAPGPU_LAUNCH_CHECK("k_bp_pairs<emit>");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_broadphase_static_pairs(void *stream, const clapgpu_bodies *b, uint32_t n_static,
                                               const double *static_aabb, uint32_t *pairs, uint32_t capacity,
                                               uint32_t *pair_total, void *scratch)
{
    int rc = check_bodies(b);
    if (rc) return rc;
    if (!pair_total || !scratch || (capacity && !pairs) || (n_static && !static_aabb))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    hipStream_t s = as_stream(stream);
    if (b->n == 0 || n_static == 0) {
        CLAPGPU_HIP(hipMemsetAsync(pair_total, 0, sizeof(uint32_t), s));
        return CLAPGPU_OK;
    }
    BpScratch sc = carve(scratch, b->n);
    const dim3 grid((b->n + PHYS_BLOCK - 1) / PHYS_BLOCK), block(PHYS_BLOCK);
    hipLaunchKernelGGL(k_bp_static<false>, grid, block, 0, s, b->n, b->pos, b->radius, n_static, static_aabb,
                       sc.pcount, pairs, capacity);
    CLAPGPU_LAUNCH_CHECK("k_bp_static<count>");
    rc = exclusive_scan_u32(s, sc.pcount, sc.pcount, b->n, pair_total, sc.scan);
    if (rc) return rc;
    hipLaunchKernelGGL(k_bp_static<true>, grid, block, 0, s, b->n, b->pos, b->radius, n_static, static_aabb,
                       sc.pcount, pairs, capacity);
    CLAPGPU_LAUNCH_CHECK("k_bp_static<emit>");
    return CLAPGPU_OK;
}
