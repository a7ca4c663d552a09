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
                        const double *yoffset, const int32_t *body_entity, uint32_t n_entities,
                        float *pos_scale, float *rot, uint32_t *entity_flags, uint8_t *moving)
{
    const uint32_t i = blockIdx.x * PHYS_BLOCK + threadIdx.x;
    if (i >= n)
        return;
    const int32_t e = body_entity[i];
    const double *p = pos + 3 * (size_t)i, *q = quat + 4 * (size_t)i, *v = lvel + 3 * (size_t)i;
    if (e >= 0 && (uint32_t)e < n_entities) {
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

// Block sums of `in`; the block that finishes last (a ticket counter, zeroed by the caller) also turns
// the sums into their exclusive scan and writes the grand total -- the middle launch of a classic
// three-launch scan folded into the first.
__global__ __launch_bounds__(PHYS_BLOCK)
void k_scan_block_sums(const uint32_t *in, uint32_t n, uint32_t *block_sums, uint32_t n_blocks, uint32_t *ticket,
                       uint32_t *total)
{
    __shared__ uint32_t lds[PHYS_BLOCK / WAVE];
    __shared__ bool is_last;
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < n) s += in[base + k];
    uint32_t tot;
    block_exclusive_scan(s, tot, lds);
    if (threadIdx.x == 0) {
        __hip_atomic_store(&block_sums[blockIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();                                            // the sum is visible before the ticket is taken
        is_last = atomicAdd(ticket, 1u) == n_blocks - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += PHYS_BLOCK) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < n_blocks ? __hip_atomic_load(&block_sums[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        uint32_t t2;
        const uint32_t ex = block_exclusive_scan(v, t2, lds);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += t2;
    }
    if (threadIdx.x == 0) {
        *total = carry;
        *ticket = 0;                                                // ready for the next scan on this stream
    }
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
// ticket: one device word that is zero when the scan starts (it is left zero again).
static int exclusive_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total,
                              uint32_t *scratch, uint32_t *ticket)
{
    const uint32_t blocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_block_sums, dim3(blocks), dim3(PHYS_BLOCK), 0, s, in, n, scratch, blocks, ticket, total);
    CLAPGPU_LAUNCH_CHECK("k_scan_block_sums");
    hipLaunchKernelGGL(k_scan_apply, dim3(blocks), dim3(PHYS_BLOCK), 0, s, in, n, scratch, out);
    CLAPGPU_LAUNCH_CHECK("k_scan_apply");
    return CLAPGPU_OK;
}

// ---------------------------------------------------------------- broadphase
// Uniform hash grid over the sphere centres, cell edge >= the largest diameter, so every partner of a
// body lies in the 27 cells around it.  HBM layout of the work space: the bodies are copied into
// bucket order as 48-byte records (centre, radius, cell, index) so a cell's members are one contiguous
// run, and a body's cell walk is spread over the lanes of a lane group (a cell or two per lane) instead
// of being a long chain of dependent gathers in one lane.  Each unordered pair is tested once (own
// cell + the 13 cells after it), see k_bp_search.
struct BpRec {
    double   p[3], r;                 // centre, radius: the AABB is p -+ r as dGeomSphere computes it
    int32_t  cx, cy, cz;
    uint32_t idx;
};
static_assert(sizeof(BpRec) == 48, "record layout");

constexpr int BP_GROUP = 8;               // lanes per body in the pair search (16: 42 us, 8: 39 us at one body per cell)
constexpr int BP_WORK = 64;               // candidate records listed per body and round

constexpr int BP_LIST = 16;               // partners kept per body between the search and the emit pass

struct BpK {
    uint32_t      n;
    const double *pos;
    const double *radius;
    double        cell;
    uint32_t      hash_mask;          // buckets - 1 (power of two)
    uint32_t     *bucket_count;       // [buckets + 1] -> bucket starts after the scan
    int4         *cells;              // [n]  (cx, cy, cz, rank of the body inside its bucket)
    BpRec        *recs;               // [n]  bodies in bucket order
    uint32_t     *pair_count;         // [n + 1] -> pair starts after the scan
    uint32_t     *partners;           // [n][BP_LIST] partners (larger index) of bodies with <= BP_LIST of them
    uint32_t     *pairs;              // [2 * capacity]
    uint32_t      capacity;
};

__device__ __forceinline__ void cell_of(const double *p, double cell, int32_t &cx, int32_t &cy, int32_t &cz)
{
    cx = (int32_t)floor(p[0] / cell);
    cy = (int32_t)floor(p[1] / cell);
    cz = (int32_t)floor(p[2] / cell);
}

// Bucket of a cell: the 4x4x4 block of cells it belongs to is hashed, the position inside the block
// is kept in the low 6 bits.  Buckets (and with them the bucket-ordered records) of neighbouring
// cells are therefore neighbours in memory, and one body's 27-cell walk touches a few cache lines
// instead of 27 scattered ones.  Blocks that share a hash slot are told apart by the cell test.
__device__ __forceinline__ uint32_t cell_hash(int32_t cx, int32_t cy, int32_t cz, uint32_t mask)
{
    const uint32_t hb = ((uint32_t)(cx >> 2) * 73856093u) ^ ((uint32_t)(cy >> 2) * 19349663u) ^
                        ((uint32_t)(cz >> 2) * 83492791u);
    const uint32_t fine = (uint32_t)(cx & 3) | ((uint32_t)(cy & 3) << 2) | ((uint32_t)(cz & 3) << 4);
    return ((hb << 6) | fine) & mask;
}

__global__ __launch_bounds__(PHYS_BLOCK)
void k_bp_histogram(BpK k)
{
    const uint32_t i = blockIdx.x * PHYS_BLOCK + threadIdx.x;
    if (i == 0) k.pair_count[k.n] = 0;
    if (i >= k.n) return;
    k.pair_count[i] = 0;                                            // the search pass counts with atomics
    int32_t cx, cy, cz;
    cell_of(k.pos + 3 * (size_t)i, k.cell, cx, cy, cz);
    const uint32_t rank = atomicAdd(&k.bucket_count[cell_hash(cx, cy, cz, k.hash_mask)], 1u);
    k.cells[i] = make_int4(cx, cy, cz, (int)rank);
}

__global__ __launch_bounds__(PHYS_BLOCK)
void k_bp_scatter(BpK k)
{
    const uint32_t i = blockIdx.x * PHYS_BLOCK + threadIdx.x;
    if (i >= k.n) return;
    const int4 c = k.cells[i];
    const uint32_t slot = k.bucket_count[cell_hash(c.x, c.y, c.z, k.hash_mask)] + (uint32_t)c.w;   // starts now
    if (slot >= k.n) return;
    BpRec rec;
#pragma unroll
    for (int a = 0; a < 3; a++) rec.p[a] = k.pos[3 * (size_t)i + a];
    rec.r = k.radius[i];
    rec.cx = c.x; rec.cy = c.y; rec.cz = c.z; rec.idx = i;
    k.recs[slot] = rec;
}

// collideAABBs: disjoint iff separated on an axis (touching boxes collide)
__device__ __forceinline__ bool aabb_overlap(const double (&alo)[3], const double (&ahi)[3], const BpRec &b)
{
#pragma unroll
    for (int a = 0; a < 3; a++)
        if (alo[a] > b.p[a] + b.r || ahi[a] < b.p[a] - b.r) return false;
    return true;
}

// Search pass: BP_GROUP lanes per body, bodies taken in bucket order.  Every unordered pair is tested
// ONCE: a body scans its own cell (partners with a larger index) and the 13 neighbour cells that come
// after its own in (z, y, x) order -- for two bodies in different cells exactly one of them has the
// other's cell among those 13.  The lanes look up the 14 record runs and spread them into an LDS work
// list (one entry per record, tagged with the cell it was listed for), then test the listed records
// one per lane, so the lanes stay busy whatever the individual runs' lengths are.  A hit (i, j) is
// appended to the partner list of min(i, j) with an atomic on that body's count; the emit pass puts
// each list in ascending order.  Bodies with more than BP_LIST partners only get the count here and
// are searched again by the emit pass.
__global__ __launch_bounds__(PHYS_BLOCK)
void k_bp_search(BpK k)
{
    constexpr int G = BP_GROUP, GROUPS = PHYS_BLOCK / G, NCELL = 14, CPL = (NCELL + G - 1) / G;
    constexpr uint32_t WL = BP_WORK;
    __shared__ uint32_t work[GROUPS][WL];
    const int grp = threadIdx.x / G, q = threadIdx.x % G;
    const uint32_t slot = blockIdx.x * GROUPS + grp;                // neighbouring groups walk neighbouring cells
    const bool body = slot < k.n;

    double alo[3] = { 0, 0, 0 }, ahi[3] = { 0, 0, 0 };
    uint32_t b0[CPL], len[CPL], i = 0, mylen = 0;
    int32_t mx = 0, my = 0, mz = 0;
#pragma unroll
    for (int c = 0; c < CPL; c++) { b0[c] = 0; len[c] = 0; }
    if (body) {
        const BpRec me = k.recs[slot];
        i = me.idx;
#pragma unroll
        for (int a = 0; a < 3; a++) { alo[a] = me.p[a] - me.r; ahi[a] = me.p[a] + me.r; }
        mx = me.cx - 1; my = me.cy - 1; mz = me.cz - 1;             // corner cell of the 3x3x3 neighbourhood
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const int cq = 13 + q + c * G;                          // 13 = own cell, 14..26 = the cells after it
            if (cq < 27) {
                const uint32_t h = cell_hash(mx + cq % 3, my + (cq / 3) % 3, mz + cq / 9, k.hash_mask);
                uint32_t se[2];                                     // start and end of the run: one 8-byte load
                __builtin_memcpy(se, k.bucket_count + h, sizeof(se));
                const uint32_t s0 = se[0];
                uint32_t s1 = se[1];
                if (s1 > k.n) s1 = k.n;
                b0[c] = s0;
                len[c] = s1 > s0 ? s1 - s0 : 0;
                mylen += len[c];
            }
        }
    }
    // this lane's first entry in the group's candidate sequence, and the sequence's length
    uint32_t incl = mylen;
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, G);
        if (q >= d) incl += up;
    }
    const uint32_t total = __shfl(incl, G - 1, G);
    const uint32_t first = incl - mylen;

    for (uint32_t base = 0; __any(base < total); base += WL) {      // one round unless > WL candidates
        uint32_t off = first;
#pragma unroll
        for (int c = 0; c < CPL; c++) {
            const int cq = 13 + q + c * G;                          // entry = record slot | cell offset (2+2+2 bits)
            const uint32_t tag = (uint32_t)(cq % 3 | ((cq / 3) % 3) << 2 | (cq / 9) << 4) << 26;
            for (uint32_t t = 0; __any(t < len[c]); t++) {
                if (t < len[c]) {
                    const uint32_t o = off + t - base;              // wraps below base: then >= WL
                    if (o < WL) work[grp][o] = (b0[c] + t) | tag;
                }
            }
            off += len[c];
        }
        wave_lds_fence();
        const uint32_t todo = total > base ? (total - base < WL ? total - base : WL) : 0;
        for (uint32_t t = q; __any(t < todo); t += G) {
            if (t < todo) {
                const uint32_t e = work[grp][t];
                const BpRec r = k.recs[e & 0x3ffffffu];
                const uint32_t j = r.idx;
                const bool own = (e >> 26) == (1u | 1u << 2 | 1u << 4);         // listed for the body's own cell
                if (j != i && (!own || j > i) && r.cx - mx == (int)((e >> 26) & 3) &&
                    r.cy - my == (int)((e >> 28) & 3) && r.cz - mz == (int)(e >> 30) && aabb_overlap(alo, ahi, r)) {
                    const uint32_t lo = i < j ? i : j, hi = i < j ? j : i;
                    const uint32_t at = atomicAdd(&k.pair_count[lo], 1u);
                    if (at < BP_LIST) k.partners[(size_t)lo * BP_LIST + at] = hi;
                }
            }
        }
        wave_lds_fence();
    }
}

// Emit pass: 16 lanes per body move its partner list to pairs[] at pair_start[i], each entry at its
// rank (the partners are distinct, so rank = number of smaller ones): ascending whatever order the
// atomics of the search pass took.  A body with more partners than the list holds is searched again
// by one lane (all 27 cells, ascending by insertion).
__global__ __launch_bounds__(PHYS_BLOCK)
void k_bp_emit(BpK k)
{
    const uint32_t t = blockIdx.x * PHYS_BLOCK + threadIdx.x;
    const uint32_t i = t / BP_LIST, q = t % BP_LIST;
    if (i >= k.n) return;
    const uint32_t start = k.pair_count[i], cnt = k.pair_count[i + 1] - start;
    uint2 *out = reinterpret_cast<uint2 *>(k.pairs);
    if (cnt <= BP_LIST) {
        if (q < cnt) {
            const uint32_t *mine = k.partners + (size_t)i * BP_LIST;
            const uint32_t v = mine[q];
            uint32_t rank = 0;
            for (uint32_t e = 0; e < cnt; e++) rank += mine[e] < v;
            if (start + rank < k.capacity)
                out[start + rank] = make_uint2(i, v);
        }
        return;
    }
    if (q != 0) return;
    const int4 c = k.cells[i];
    const double ri = k.radius[i];
    double alo[3], ahi[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double p = k.pos[3 * (size_t)i + a];
        alo[a] = p - ri;
        ahi[a] = p + ri;
    }
    uint32_t w = 0;
    for (int cq = 0; cq < 27; cq++) {
        const int32_t nx = c.x + cq % 3 - 1, ny = c.y + (cq / 3) % 3 - 1, nz = c.z + cq / 9 - 1;
        const uint32_t h = cell_hash(nx, ny, nz, k.hash_mask);
        uint32_t b0 = k.bucket_count[h], b1 = k.bucket_count[h + 1];
        if (b1 > k.n) b1 = k.n;
        for (uint32_t s = b0; s < b1; s++) {
            const BpRec r = k.recs[s];
            const uint32_t j = r.idx;
            if (!(j > i && r.cx == nx && r.cy == ny && r.cz == nz && aabb_overlap(alo, ahi, r)))
                continue;
            if (start + w < k.capacity) {
                uint32_t b = w;                                     // insertion keeps the run ascending
                while (b > 0 && out[start + b - 1].y > j) {
                    out[start + b] = out[start + b - 1];
                    b--;
                }
                out[start + b] = make_uint2(i, j);
            }
            w++;
        }
    }
}

// bodies x static geoms (dSpaceCollide2(ground, bodies)): the statics are streamed through LDS tiles,
// every body tests all of them; pairs (body, static) ascending.  The search pass keeps the first
// BP_LIST hits of a body (already ascending) for the emit pass, which only re-tests the few bodies
// with more (k_bp_static_emit).
constexpr int STATIC_TILE = 256;
__global__ __launch_bounds__(PHYS_BLOCK)
void k_bp_static_search(uint32_t n, const double *pos, const double *radius, uint32_t n_static, const double *static_aabb,
                        uint32_t *pair_count, uint32_t *partners)
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
                if (cnt < BP_LIST)
                    partners[(size_t)i * BP_LIST + cnt] = base + s;
                cnt++;
            }
    }
    if (live)
        pair_count[i] = cnt;
    if (i == 0) {                                                   // the scan's end marker and its ticket
        pair_count[n] = 0;
        pair_count[n + 1] = 0;
    }
}

// Emit pass of the statics: 16 lanes per body copy its listed hits to pairs[] at pair_start[i]; a body
// with more hits than the list holds is re-tested by one lane against every static box (ascending).
__global__ __launch_bounds__(PHYS_BLOCK)
void k_bp_static_emit(uint32_t n, const double *pos, const double *radius, uint32_t n_static, const double *static_aabb,
                      const uint32_t *pair_start, const uint32_t *partners, uint32_t *pairs, uint32_t capacity)
{
    const uint32_t t = blockIdx.x * PHYS_BLOCK + threadIdx.x;
    const uint32_t i = t / BP_LIST, q = t % BP_LIST;
    if (i >= n) return;
    const uint32_t start = pair_start[i], cnt = pair_start[i + 1] - start;
    uint2 *out = reinterpret_cast<uint2 *>(pairs);
    if (cnt <= BP_LIST) {
        if (q < cnt && start + q < capacity)
            out[start + q] = make_uint2(i, partners[(size_t)i * BP_LIST + q]);
        return;
    }
    if (q != 0) return;
    const double r = radius[i];
    double bb[6];
#pragma unroll
    for (int a = 0; a < 3; a++) { bb[2 * a] = pos[3 * (size_t)i + a] - r; bb[2 * a + 1] = pos[3 * (size_t)i + a] + r; }
    uint32_t w = 0;
    for (uint32_t s = 0; s < n_static; s++) {
        const double *sb = static_aabb + 6 * (size_t)s;
        if (bb[0] > sb[1] || bb[1] < sb[0] || bb[2] > sb[3] || bb[3] < sb[2] || bb[4] > sb[5] || bb[5] < sb[4])
            continue;
        if (start + w < capacity) out[start + w] = make_uint2(i, s);
        w++;
    }
}

// phys_body_rotate_xform (physics.c:136-145) for the linked entities that default_update rebuilt
__global__ __launch_bounds__(PHYS_BLOCK)
void k_bodies_rotate_from_entities(uint32_t n_links, const uint32_t *link_body, const uint32_t *link_entity,
                                   uint32_t n_bodies, uint32_t n_entities, uint32_t mode, const float4 *rot,
                                   const int32_t *parent, const uint32_t *flags, double *quat)
{
    const uint32_t k = blockIdx.x * PHYS_BLOCK + threadIdx.x;
    if (k >= n_links) return;
    const uint32_t b = link_body[k], e = link_entity[k];
    if (b >= n_bodies || e >= n_entities || parent[e] >= 0) return;
    if (!(mode & CLAPGPU_UPDATE_ALL_DIRTY) && !(flags[e] & CLAPGPU_E_DIRTY)) return;
    const float4 r = rot[e];
    double q[4] = { (double)r.w, (double)r.x, (double)r.y, (double)r.z };
    const double l = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);   // dNormalize4
    for (int a = 0; a < 4; a++) quat[4 * (size_t)b + a] = q[a] * l;
}

// phys_contact_surface (physics.c:291-330) for the two colliders' parameter rows (NULL: defaults)
__device__ __forceinline__ void contact_surface(clapgpu_contact &c, const double *m1, const double *m2)
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
    c.nc = 1;
}

// ODE's dCollideSpheres
__device__ __forceinline__ bool contact_sphere_sphere(clapgpu_contact &c, const double *p1, double r1, const double *p2,
                                                      double r2)
{
    const double dx = p1[0] - p2[0], dy = p1[1] - p2[1], dz = p1[2] - p2[2];
    const double d = sqrt(dx * dx + dy * dy + dz * dz);
    if (d > r1 + r2) return false;
    if (d <= 0) {
        c.pos[0] = p1[0]; c.pos[1] = p1[1]; c.pos[2] = p1[2];
        c.normal[0] = 1; c.normal[1] = 0; c.normal[2] = 0;
        c.depth = r1 + r2;
    } else {
        const double d1 = 1.0 / d;
        c.normal[0] = dx * d1; c.normal[1] = dy * d1; c.normal[2] = dz * d1;
        const double kk = 0.5 * (r2 - r1 - d);
        c.pos[0] = p1[0] + c.normal[0] * kk;
        c.pos[1] = p1[1] + c.normal[1] * kk;
        c.pos[2] = p1[2] + c.normal[2] * kk;
        c.depth = r1 + r2 - d;
    }
    return true;
}

// ODE's dCollideSphereBox for an axis-aligned box given as aabb[6] = (minx,maxx,miny,maxy,minz,maxz):
// box position = centre, side = max - min, R = identity (oracle/physics.c has the restated algorithm)
__device__ __forceinline__ bool contact_sphere_box(clapgpu_contact &c, const double *c0, double rad, const double *bb)
{
    double bp[3], l[3], p[3], t[3];
    bool onborder = false;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        bp[a] = (bb[2 * a] + bb[2 * a + 1]) * 0.5;
        l[a] = (bb[2 * a + 1] - bb[2 * a]) * 0.5;
        p[a] = c0[a] - bp[a];
        t[a] = p[a];
        if (t[a] < -l[a]) { t[a] = -l[a]; onborder = true; }
        if (t[a] > l[a]) { t[a] = l[a]; onborder = true; }
    }
    if (!onborder) {                                                // centre inside: push out through the closest face
        double min_distance = l[0] - fabs(t[0]);
        int mini = 0;
#pragma unroll
        for (int a = 1; a < 3; a++) {
            const double face_distance = l[a] - fabs(t[a]);
            if (face_distance < min_distance) { min_distance = face_distance; mini = a; }
        }
        c.pos[0] = c0[0]; c.pos[1] = c0[1]; c.pos[2] = c0[2];
        const double sgn = t[mini] > 0 ? 1.0 : -1.0;
        c.normal[0] = mini == 0 ? sgn : 0.0; c.normal[1] = mini == 1 ? sgn : 0.0; c.normal[2] = mini == 2 ? sgn : 0.0;
        c.depth = min_distance + rad;
        return true;
    }
    double r[3] = { p[0] - t[0], p[1] - t[1], p[2] - t[2] };
    const double depth = rad - sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (depth < 0) return false;
    c.pos[0] = t[0] + bp[0]; c.pos[1] = t[1] + bp[1]; c.pos[2] = t[2] + bp[2];
    {                                                               // dSafeNormalize3
        const double aa[3] = { fabs(r[0]), fabs(r[1]), fabs(r[2]) };
        int idx = 0;
        bool zero = false;
        if (aa[1] > aa[0]) idx = aa[2] > aa[1] ? 2 : 1;
        else if (aa[2] > aa[0]) idx = 2;
        else zero = aa[0] <= 0;
        if (zero) { r[0] = 1; r[1] = 0; r[2] = 0; }
        else {
            const double m = idx == 0 ? aa[0] : idx == 1 ? aa[1] : aa[2];
            r[0] /= m; r[1] /= m; r[2] /= m;
            const double k = 1.0 / sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
            r[0] *= k; r[1] *= k; r[2] *= k;
        }
    }
    c.normal[0] = r[0]; c.normal[1] = r[1]; c.normal[2] = r[2];
    c.depth = depth;
    return true;
}

// near_callback's dCollide + phys_contact_surface (see include/clapgpu.h): one lane per candidate pair,
// IEEE fp64 (sqrt, divide), no contraction.  BOX = false: (body, body) sphere pairs; BOX = true: (body,
// static box) pairs, `other` = static_aabb, `other_material` = the static colliders' parameter rows.
template <bool BOX>
__global__ __launch_bounds__(PHYS_BLOCK)
void k_contacts(const double *pos, const double *radius, uint32_t n_bodies, const double *other, uint32_t n_other,
                const uint2 *pairs, const uint32_t *pair_total, uint32_t capacity, const double *material,
                const double *other_material, clapgpu_contact *out, uint32_t *contact_total)
{
    __shared__ __attribute__((aligned(16))) double recs[PHYS_BLOCK / WAVE][WAVE * 13];
    static_assert(sizeof(clapgpu_contact) == 13 * sizeof(double), "contact record layout");
    const uint32_t n_pairs = *pair_total < capacity ? *pair_total : capacity;
    const int lane = lane_id();
    uint32_t found = 0;
    // the pair count is only known on the device: a fixed grid strides over the pairs (a grid sized for
    // the capacity spends 50 us launching empty workgroups)
    for (uint32_t k = blockIdx.x * PHYS_BLOCK + threadIdx.x; k - lane < n_pairs; k += gridDim.x * PHYS_BLOCK) {
    double *rec = recs[threadIdx.x / WAVE];
    bool touch = false;
    if (k < n_pairs) {
        const uint2 pr = pairs[k];
        clapgpu_contact c;
        memset(&c, 0, sizeof(c));
        if (pr.x < n_bodies && pr.y < (BOX ? n_other : n_bodies)) {
            if (BOX) {
                touch = contact_sphere_box(c, pos + 3 * (size_t)pr.x, radius[pr.x], other + 6 * (size_t)pr.y);
                if (touch)
                    contact_surface(c, material && other_material ? material + 5 * (size_t)pr.x : nullptr,
                                    material && other_material ? other_material + 5 * (size_t)pr.y : nullptr);
            } else {
                touch = contact_sphere_sphere(c, pos + 3 * (size_t)pr.x, radius[pr.x], pos + 3 * (size_t)pr.y, radius[pr.y]);
                if (touch)
                    contact_surface(c, material ? material + 5 * (size_t)pr.x : nullptr,
                                    material ? material + 5 * (size_t)pr.y : nullptr);
            }
        }
        // the 104-byte records of a wave are contiguous in memory: stage them in LDS and write the run as
        // 16-byte pieces (a record per lane straight to memory is 13 scattered 8-byte stores per lane)
        memcpy(rec + (size_t)lane * 13, &c, sizeof(c));
    }
    wave_lds_fence();
    {
        const uint32_t wave_first = k - lane;                       // first pair of this wave
        const uint32_t n_here = wave_first < n_pairs ? (n_pairs - wave_first < WAVE ? n_pairs - wave_first : WAVE) : 0;
        const uint32_t n16 = n_here * (uint32_t)(sizeof(clapgpu_contact) / 8) / 2;      // 16-byte pieces (104 * 64 % 16 == 0 only for even counts)
        const double2 *src = reinterpret_cast<const double2 *>(rec);
        double2 *dst = reinterpret_cast<double2 *>(out + wave_first);
        if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
            for (uint32_t q = lane; q < n16; q += WAVE) dst[q] = src[q];
            if ((n_here & 1) && lane == 0)                          // odd count: the last 8 bytes
                reinterpret_cast<double *>(out + wave_first)[n_here * 13 - 1] = rec[n_here * 13 - 1];
        } else {
            for (uint32_t q = lane; q < n_here * 13; q += WAVE)
                reinterpret_cast<double *>(out + wave_first)[q] = rec[q];
        }
    }
    found += (uint32_t)__popcll(__ballot(touch));
    wave_lds_fence();                                               // the staging tile is reused by the next trip
    }
    // one global atomic per workgroup: same-address atomics serialise at ~12 ns each (4096 of them were
    // 50 us of this kernel)
    __shared__ uint32_t block_found;
    if (threadIdx.x == 0) block_found = 0;
    __syncthreads();
    if (lane == 0 && found) atomicAdd(&block_found, found);
    __syncthreads();
    if (contact_total && threadIdx.x == 0 && block_found)
        atomicAdd(contact_total, block_found);
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

extern "C" int clapgpu_phys_body_update(void *stream, const clapgpu_bodies *b, uint32_t n_entities, float *pos_scale,
                                        float *rot, uint32_t *entity_flags, uint8_t *moving)
{
    int rc = check_bodies(b);
    if (rc) return rc;
    if (!b->yoffset || !b->body_entity || !pos_scale || !rot || !entity_flags)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (b->n == 0) return CLAPGPU_OK;
    hipLaunchKernelGGL(k_phys_body_update, dim3((b->n + PHYS_BLOCK - 1) / PHYS_BLOCK), dim3(PHYS_BLOCK), 0,
                       as_stream(stream), b->n, b->pos, b->quat, b->lvel, b->yoffset, b->body_entity, n_entities,
                       pos_scale, rot, entity_flags, moving);
    CLAPGPU_LAUNCH_CHECK("k_phys_body_update");
    return CLAPGPU_OK;
}

static uint32_t bucket_count_for(uint32_t n)
{
    uint32_t b = 1024;
    while (b < 2 * n && b < (1u << 26)) b <<= 1;
    return b;
}

// layout of the broadphase work space (uint32 units; cells and records are 16-byte aligned)
struct BpScratch { uint32_t *bucket, *cells, *recs, *pcount, *partners, *scan, *total; };

static size_t align4(size_t w) { return (w + 3) & ~(size_t)3; }

static size_t bp_scratch_words(uint32_t n)
{
    const size_t buckets = bucket_count_for(n);
    const size_t scan = (buckets + 1 + SCAN_TILE - 1) / SCAN_TILE + ((size_t)n + 1 + SCAN_TILE) / SCAN_TILE + 8;
    return align4(buckets + 1) + 4 * (size_t)n + 12 * (size_t)n + align4((size_t)n + 2) +
           (size_t)BP_LIST * n + 8 + scan + 8 + 16;
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
    s.bucket = p;   p += align4(buckets + 1);
    s.cells = p;    p += 4 * (size_t)n;
    s.recs = p;     p += 12 * (size_t)n;
    s.pcount = p;   p += align4((size_t)n + 2);          // [n + 1]: ticket of the statics' scan
    s.partners = p; p += (size_t)BP_LIST * n;
    s.total = p;    p += 8;
    s.scan = p;
    return s;
}

extern "C" int clapgpu_broadphase_pairs(void *stream, const clapgpu_bodies *b, double cell,
                                        uint32_t *pairs, uint32_t capacity, uint32_t *pair_total, void *scratch)
{
    int rc = check_bodies(b);
    if (rc) return rc;
    if (!pair_total || !scratch || (capacity && !pairs) || !(cell > 0.0) || ((uintptr_t)scratch & 15))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (b->n > (1u << 26))                                         // work-list entries carry the cell offset in 6 bits
        return CLAPGPU_ERR_TOO_LARGE;
    hipStream_t s = as_stream(stream);
    if (b->n == 0) {
        CLAPGPU_HIP(hipMemsetAsync(pair_total, 0, sizeof(uint32_t), s));
        return CLAPGPU_OK;
    }
    const uint32_t n = b->n, buckets = bucket_count_for(n);
    BpScratch sc = carve(scratch, n);
    // bucket counts and, in the padding word behind them, the scans' ticket
    CLAPGPU_HIP(hipMemsetAsync(sc.bucket, 0, ((size_t)buckets + 2) * sizeof(uint32_t), s));

    BpK k;
    k.n = n; k.pos = b->pos; k.radius = b->radius; k.cell = cell; k.hash_mask = buckets - 1;
    k.bucket_count = sc.bucket;
    k.cells = reinterpret_cast<int4 *>(sc.cells); k.recs = reinterpret_cast<BpRec *>(sc.recs);
    k.pair_count = sc.pcount; k.partners = sc.partners; k.pairs = pairs; k.capacity = capacity;
    const dim3 grid((n + PHYS_BLOCK - 1) / PHYS_BLOCK), block(PHYS_BLOCK);

    hipLaunchKernelGGL(k_bp_histogram, grid, block, 0, s, k);
    CLAPGPU_LAUNCH_CHECK("k_bp_histogram");
    rc = exclusive_scan_u32(s, sc.bucket, sc.bucket, buckets + 1, sc.total, sc.scan, sc.bucket + buckets + 1);   // starts; [buckets] = n
    if (rc) return rc;
    hipLaunchKernelGGL(k_bp_scatter, grid, block, 0, s, k);
    CLAPGPU_LAUNCH_CHECK("k_bp_scatter");
    constexpr uint32_t per_block = PHYS_BLOCK / BP_GROUP;
    hipLaunchKernelGGL(k_bp_search, dim3((n + per_block - 1) / per_block), block, 0, s, k);
    CLAPGPU_LAUNCH_CHECK("k_bp_search");
    rc = exclusive_scan_u32(s, sc.pcount, sc.pcount, n + 1, pair_total, sc.scan, sc.bucket + buckets + 1);   // starts; [n] = total
    if (rc) return rc;
    const uint64_t emit_threads = (uint64_t)n * BP_LIST;
    hipLaunchKernelGGL(k_bp_emit, dim3((uint32_t)((emit_threads + PHYS_BLOCK - 1) / PHYS_BLOCK)), block, 0, s, k);
    CLAPGPU_LAUNCH_CHECK("k_bp_emit");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_broadphase_static_pairs(void *stream, const clapgpu_bodies *b, uint32_t n_static,
                                               const double *static_aabb, uint32_t *pairs, uint32_t capacity,
                                               uint32_t *pair_total, void *scratch)
{
    int rc = check_bodies(b);
    if (rc) return rc;
    if (!pair_total || !scratch || (capacity && !pairs) || (n_static && !static_aabb) || ((uintptr_t)scratch & 15))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    hipStream_t s = as_stream(stream);
    if (b->n == 0 || n_static == 0) {
        CLAPGPU_HIP(hipMemsetAsync(pair_total, 0, sizeof(uint32_t), s));
        return CLAPGPU_OK;
    }
    const uint32_t n = b->n;
    BpScratch sc = carve(scratch, n);
    const dim3 grid((n + PHYS_BLOCK - 1) / PHYS_BLOCK), block(PHYS_BLOCK);
    hipLaunchKernelGGL(k_bp_static_search, grid, block, 0, s, n, b->pos, b->radius, n_static, static_aabb,
                       sc.pcount, sc.partners);
    CLAPGPU_LAUNCH_CHECK("k_bp_static_search");
    rc = exclusive_scan_u32(s, sc.pcount, sc.pcount, n + 1, pair_total, sc.scan, sc.pcount + n + 1);   // starts; [n] = total
    if (rc) return rc;
    const uint64_t copy_threads = (uint64_t)n * BP_LIST;
    hipLaunchKernelGGL(k_bp_static_emit, dim3((uint32_t)((copy_threads + PHYS_BLOCK - 1) / PHYS_BLOCK)), block, 0, s,
                       n, b->pos, b->radius, n_static, static_aabb, sc.pcount, sc.partners, pairs, capacity);
    CLAPGPU_LAUNCH_CHECK("k_bp_static_emit");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_contacts_spheres(void *stream, const clapgpu_bodies *b, const uint32_t *pairs,
                                        const uint32_t *pair_total, uint32_t capacity, const double *material,
                                        clapgpu_contact *contacts, uint32_t *contact_total)
{
    int rc = check_bodies(b);
    if (rc) return rc;
    if (!pair_total || (capacity && (!pairs || !contacts)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    hipStream_t s = as_stream(stream);
    if (contact_total)
        CLAPGPU_HIP(hipMemsetAsync(contact_total, 0, sizeof(uint32_t), s));
    if (capacity == 0 || b->n == 0)
        return CLAPGPU_OK;
    // the pair count lives on the device: launch for the capacity, lanes past the count retire at once
    const uint32_t blocks = (capacity + PHYS_BLOCK - 1) / PHYS_BLOCK;
    hipLaunchKernelGGL(k_contacts<false>, dim3(blocks < 512 ? blocks : 512), dim3(PHYS_BLOCK), 0, s,
                       b->pos, b->radius, b->n, nullptr, 0u, reinterpret_cast<const uint2 *>(pairs), pair_total, capacity,
                       material, nullptr, contacts, contact_total);
    CLAPGPU_LAUNCH_CHECK("k_contacts<spheres>");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_contacts_sphere_box(void *stream, const clapgpu_bodies *b, uint32_t n_static,
                                           const double *static_aabb, const uint32_t *pairs, const uint32_t *pair_total,
                                           uint32_t capacity, const double *material, const double *static_material,
                                           clapgpu_contact *contacts, uint32_t *contact_total)
{
    int rc = check_bodies(b);
    if (rc) return rc;
    if (!pair_total || (n_static && !static_aabb) || (capacity && (!pairs || !contacts)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    hipStream_t s = as_stream(stream);
    if (contact_total)
        CLAPGPU_HIP(hipMemsetAsync(contact_total, 0, sizeof(uint32_t), s));
    if (capacity == 0 || b->n == 0 || n_static == 0)
        return CLAPGPU_OK;
    const uint32_t blocks = (capacity + PHYS_BLOCK - 1) / PHYS_BLOCK;
    hipLaunchKernelGGL(k_contacts<true>, dim3(blocks < 512 ? blocks : 512), dim3(PHYS_BLOCK), 0, s,
                       b->pos, b->radius, b->n, static_aabb, n_static, reinterpret_cast<const uint2 *>(pairs), pair_total,
                       capacity, material, static_material, contacts, contact_total);
    CLAPGPU_LAUNCH_CHECK("k_contacts<sphere_box>");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_bodies_rotate_from_entities(void *stream, const clapgpu_bodies *b, const clapgpu_entities *e,
                                                   uint32_t mode, uint32_t n_links, const uint32_t *link_body,
                                                   const uint32_t *link_entity)
{
    int rc = check_bodies(b);
    if (rc) return rc;
    if (!e || !e->rot || !e->parent || !e->flags || (n_links && (!link_body || !link_entity)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (n_links == 0 || b->n == 0)
        return CLAPGPU_OK;
    hipLaunchKernelGGL(k_bodies_rotate_from_entities, dim3((n_links + PHYS_BLOCK - 1) / PHYS_BLOCK), dim3(PHYS_BLOCK), 0,
                       as_stream(stream), n_links, link_body, link_entity, b->n, e->n, mode,
                       reinterpret_cast<const float4 *>(e->rot), e->parent, e->flags, b->quat);
    CLAPGPU_LAUNCH_CHECK("k_bodies_rotate_from_entities");
    return CLAPGPU_OK;
}
