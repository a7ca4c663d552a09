// physics2.hip -- capsule / sphere bodies: integrate, geom AABBs, broadphase over explicit AABBs, for gfx950
// (narrowphase contact records and the capsule sweep: contacts.hip).
//
// What phys_step() (physics.c:773-787) does per fixed substep through ODE, for bodies without constraint rows:
//   k_bodies_step     dWorldQuickStep's body stage (quickstep.cpp stage 0 + dxStepBody + auto-disable), fused with
//                     the moved geom's axis / AABB (dxCapsule::computeAABB)                      HBM-bound, 1 lane / body
//   k_bp_*            dSpaceCollide2(ground, bodies) + dSpaceCollide(bodies) (physics.c:751-753) as ascending
//                     candidate-pair lists, five launches for both passes: bodies binned by AABB centre into a hash
//                     grid of 4x4x4-cell blocks, copied into cell order, searched one wavefront per 16-body tile with
//                     the candidates of each distinct cell listed once in LDS (section comment below)
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
static_assert(true, "");

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
    double *geom_records;
};

// The next broadphase's first launch (k_bp_bin: one atomic per body on its cell's counter) done by the step that writes the
// box it would read: clapgpu_bodies_step_prebin.  key == nullptr: off.
struct BinK { double cell; uint32_t mask; uint32_t *key, *rank, *cell_cnt, *ctrl; };
__device__ __forceinline__ void bin_body(const BinK &bin, uint32_t i, const double (&bb)[6]);

__device__ __forceinline__ void write_geom(const BodiesK &b, uint32_t i, const double (&p)[3], const double (&q)[4],
                                           double (*bb_out)[6] = nullptr)
{
    if (!b.aabb && !b.axis && !b.geom_records) return;
    double R[12], axis[3], bb[6];
    phd::q_to_R(q, R);
    phd::capsule_axis(R, b.Roff, axis);
    const double lz = b.length ? b.length[i] : 0.0;
    phd::geom_aabb(p, b.radius[i], lz, axis, bb);
    if (b.geom_records) {                                        // the narrowphase's view of this geom, one 64-byte sector
        double2 *r = reinterpret_cast<double2 *>(b.geom_records + 8 * (size_t)i);
        r[0] = make_double2(p[0], p[1]); r[1] = make_double2(p[2], axis[0]);
        r[2] = make_double2(axis[1], axis[2]); r[3] = make_double2(b.radius[i], lz);
    }
    if (b.axis) { double *a = b.axis + 3 * (size_t)i; a[0] = axis[0]; a[1] = axis[1]; a[2] = axis[2]; }
    if (b.aabb) {
        double2 *o = reinterpret_cast<double2 *>(b.aabb + 6 * (size_t)i);
        o[0] = make_double2(bb[0], bb[1]); o[1] = make_double2(bb[2], bb[3]); o[2] = make_double2(bb[4], bb[5]);
    }
    if (bb_out)
#pragma unroll
        for (int a = 0; a < 6; a++) (*bb_out)[a] = bb[a];
}

// a body the step leaves alone keeps its stored box: binned from there
__device__ __forceinline__ void bin_stored(const BinK &bin, const BodiesK &b, uint32_t i)
{
    const double2 *p = reinterpret_cast<const double2 *>(b.aabb + 6 * (size_t)i);
    const double2 x = p[0], y = p[1], z = p[2];
    const double bb[6] = { x.x, x.y, y.x, y.y, z.x, z.y };
    bin_body(bin, i, bb);
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

template <bool BIN>
__global__ __launch_bounds__(PB)
void k_bodies_step(BodiesK b, WorldK2 w, double h, BinK bin)
{
    const uint32_t i = blockIdx.x * PB + threadIdx.x;
    if (BIN && i == 0) bin.ctrl[3] = bin.ctrl[3] + 1;                   // CTRL_EPOCH: what k_bp_bin's first thread does
    if (i >= b.n) return;
    uint32_t fl = b.bflags[i];
    if (fl & CLAPGPU_BODY_DISABLED) { if (BIN) bin_stored(bin, b, i); return; }
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
            if (BIN) bin_stored(bin, b, i);
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
    if (BIN) {
        double bb[6];
        write_geom(b, i, p, q, &bb);
        bin_body(bin, i, bb);
    } else
        write_geom(b, i, p, q);
}

// ================================================================================== broadphase
// Hash grid over the AABB centres, cell >= the largest body AABB edge, so a body's partners have their centres in
// the 27 cells around its own.  Cells are grouped in 4x4x4 blocks: a cell's slot = (hash of its block) * 64 + its
// position inside the block, so the 64 cells of a block are neighbours in memory and the per-frame prefix work
// splits into a wave-sized piece per block (k_bp_cells) and a scan over the block totals.
// Five launches for BOTH passes of __phys_step (bodies x bodies and statics x bodies):
//   k_bp_bin      one atomic per body on its cell's counter (the return value is its rank in the cell)
//   k_bp_cells    one wavefront per block: exclusive prefix of its 64 cell counts; block starts by a single-pass scan
//                 with decoupled look-back over the workgroups' totals
//   k_bp_scatter  64-byte records (box, index, cell coordinates) into cell order
//   k_bp_search   one wavefront per tile of 16 bodies in cell order; candidates (own cell: partners with a larger
//                 index; the 13 cells after it; the statics registered for the block) listed once per DISTINCT cell /
//                 block bucket of the tile in an LDS work list and tested against the tile's boxes; hits go to the
//                 partner list of min(i, j); then the large statics from LDS
//   k_bp_emit     one thread per body in index order: offset = tile offset (look-back scan over the 256-body tiles) +
//                 scan inside the tile, its list written in ascending partner order, so the output is the canonical
//                 ascending list whatever order the atomics took
// Two different cells of one 3x3x3 neighbourhood never share a slot (same position inside a block means at least four
// cells apart), and a body from a far block that shares a slot cannot overlap (cell >= every edge), so candidates need
// no cell check beyond the box test.
constexpr int BP_LIST = 16;            // partners kept per body in its fixed slot
constexpr int BP_TILE = 16;            // bodies per wavefront of the search
constexpr int BP_WORK = 512;           // candidate entries listed per tile and round (256: spheres -2 us, capsules +5 us)
constexpr int BP_EMIT_TILE = 1024;     // bodies per tile of the pair-offset scan (= emit block; 256: +3 us, four times the look-back words)
#ifndef BP_SEARCH_IN_FLIGHT
#define BP_SEARCH_IN_FLIGHT 1            // candidate records gathered per lane and round
#endif
constexpr int CTRL_STATUS = 2, CTRL_EPOCH = 3, CTRL_CONTACT_WORD = 8;   // [8..9]: clapgpu_contacts_geoms_both's ticket + counts, zero between launches     // the frame counter lives on the device: a captured graph replays the same arguments

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

__host__ __device__ __forceinline__ uint32_t cell_slot(int32_t cx, int32_t cy, int32_t cz, uint32_t mask)
{
    return block_hash(cx >> 2, cy >> 2, cz >> 2, mask) << 6 | (uint32_t)(cx & 3) | (uint32_t)(cy & 3) << 2 | (uint32_t)(cz & 3) << 4;
}

struct BpRec { double bb[6]; uint32_t idx; int32_t cell[3]; };    // 64 bytes; cell = the box centre's cell (dynamic records)

struct BpK {
    uint32_t n;
    double cell;
    uint32_t mask;                       // block buckets - 1
    const double *aabb;
    uint32_t *cell_cnt;                  // [buckets * 64] the bin pass's counters, zero between frames
    uint2    *cell_range;                // [buckets * 64] (first position in cell order, bodies) of every cell: one load per lookup
    uint32_t *key, *rank;                // [n] cell slot and rank inside the cell
    uint32_t *entries;                   // [n] body indices in cell order
    struct BpRec *recs;                  // [n] the same with the boxes: what the search reads
    uint32_t *cnt, *scnt;                // [n] partners (larger index) / statics per body: atomics in the search
    uint32_t *partners, *spartners;      // [n][BP_LIST]
    uint64_t *lb_body, *lb_static;       // [tiles] look-back words of the pair-offset scan (k_bp_emit)
    uint64_t *lb_cells;                  // [buckets / 4] look-back words of the block-start scan (k_bp_cells)
    uint32_t *ctrl;
    uint32_t n_tiles;
    // statics (binned on the host at create time)
    const uint32_t *s_start;             // [buckets + 1]
    const uint32_t *s_entries;
    const double *s_aabb;
    const uint32_t *s_large;
    const struct BpRec *s_recs;          // s_entries with their boxes (what the search gathers)
    const struct BpRec *s_lrecs;         // the large statics with their boxes
    uint32_t n_large, n_static;
    // outputs
    uint32_t *pairs, capacity, *pair_total;
    uint32_t *spairs, scapacity, *spair_total;
};

__device__ __forceinline__ void load_box(const double *aabb, uint32_t i, double (&bb)[6])
{
    const double2 *p = reinterpret_cast<const double2 *>(aabb + 6 * (size_t)i);
    const double2 a = p[0], b = p[1], c = p[2];
    bb[0] = a.x; bb[1] = a.y; bb[2] = b.x; bb[3] = b.y; bb[4] = c.x; bb[5] = c.y;
}

__device__ __forceinline__ void box_cell(const double (&bb)[6], double cell, int32_t &cx, int32_t &cy, int32_t &cz)
{
    cx = cell_coord((bb[0] + bb[1]) * 0.5, cell);
    cy = cell_coord((bb[2] + bb[3]) * 0.5, cell);
    cz = cell_coord((bb[4] + bb[5]) * 0.5, cell);
}

__device__ __forceinline__ bool boxes_overlap(const double (&a)[6], const double (&b)[6])
{
    return !(a[0] > b[1] || a[1] < b[0] || a[2] > b[3] || a[3] < b[2] || a[4] > b[5] || a[5] < b[4]);
}

__device__ __forceinline__ void bin_body(const BinK &bin, uint32_t i, const double (&bb)[6])
{
    if (bb[1] - bb[0] > bin.cell || bb[3] - bb[2] > bin.cell || bb[5] - bb[4] > bin.cell)
        atomicOr(&bin.ctrl[CTRL_STATUS], 1u);
    int32_t cx, cy, cz;
    box_cell(bb, bin.cell, cx, cy, cz);
    const uint32_t slot = cell_slot(cx, cy, cz, bin.mask);
    bin.key[i] = slot;
    bin.rank[i] = atomicAdd(&bin.cell_cnt[slot], 1u);
}

// Launch 1 (skipped when the step before it has binned the boxes it wrote: clapgpu_bodies_step_prebin)
__global__ __launch_bounds__(PB)
void k_bp_bin(BpK k)
{
    const uint32_t i = blockIdx.x * PB + threadIdx.x;
    if (i == 0) k.ctrl[CTRL_EPOCH] = k.ctrl[CTRL_EPOCH] + 1;          // first launch of the frame; read by the later ones
    if (i >= k.n) return;
    double bb[6];
    load_box(k.aabb, i, bb);
    if (bb[1] - bb[0] > k.cell || bb[3] - bb[2] > k.cell || bb[5] - bb[4] > k.cell)
        atomicOr(&k.ctrl[CTRL_STATUS], 1u);
    int32_t cx, cy, cz;
    box_cell(bb, k.cell, cx, cy, cz);
    const uint32_t slot = cell_slot(cx, cy, cz, k.mask);
    k.key[i] = slot;
    k.rank[i] = atomicAdd(&k.cell_cnt[slot], 1u);
}

// Single-pass scans with decoupled look-back (the block starts in k_bp_cells, the pair offsets of the 256-body emit
// tiles in k_bp_emit): a tile's offset = the sum of everything before it.  Tile b publishes (flag, epoch, value) as ONE 64-bit word -- its own sum first
// (AGGREGATE), its inclusive prefix once known (PREFIX) -- and a wavefront walks back over its predecessors' words, 64 at
// a time, until it meets a PREFIX.  Workgroups are dispatched in index order and wait only on lower indices, so the
// walk always terminates; the frame's epoch in the word makes last frame's entries read as empty (no clearing pass).
constexpr uint64_t LB_AGG = 1ull << 62, LB_PREFIX = 2ull << 62, LB_FLAGS = 3ull << 62;
__device__ __forceinline__ uint64_t lb_word(uint64_t flag, uint32_t epoch, uint32_t value)
{
    return flag | ((uint64_t)(epoch & 0x3fffffffu) << 32) | value;
}

// exclusive prefix of tile `b` (called by one whole wavefront); publishes the tile's own words
__device__ __forceinline__ uint32_t lb_exclusive(uint64_t *state, uint32_t b, uint32_t sum, uint32_t epoch, uint32_t *status)
{
    const int lane = lane_id();
    if (b == 0) {
        if (lane == 0) __hip_atomic_store(&state[0], lb_word(LB_PREFIX, epoch, sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0;
    }
    if (lane == 0) __hip_atomic_store(&state[b], lb_word(LB_AGG, epoch, sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t excl = 0;
    for (int64_t top = (int64_t)b - 1; top >= 0; top -= WAVE) {         // window: tiles top, top-1, ..., top-63
        const int64_t t = top - lane;
        uint64_t w = 0;
        if (t >= 0) {
            uint32_t spins = 0;
            do {
                w = __hip_atomic_load(&state[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((w & LB_FLAGS) && (uint32_t)((w >> 32) & 0x3fffffffu) == (epoch & 0x3fffffffu)) break;
                w = 0;
                __builtin_amdgcn_s_sleep(1);
            } while (++spins < (1u << 22));                               // a bound, not an expectation: see above
            if (!w) atomicOr(status, 4u);
        }
        const uint64_t is_prefix = __ballot(t >= 0 && (w & LB_FLAGS) == LB_PREFIX);
        const int stop = is_prefix ? __builtin_ctzll(is_prefix) : WAVE - 1;   // nearest predecessor that knows its prefix
        uint32_t v = (t >= 0 && lane <= stop) ? (uint32_t)w : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        excl += v;
        if (is_prefix) break;
    }
    if (lane == 0) __hip_atomic_store(&state[b], lb_word(LB_PREFIX, epoch, excl + sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// Launch 2: wave w = block bucket w; a workgroup's BP_CELLS_BLOCK / 64 block totals enter the look-back scan as one tile
constexpr int BP_CELLS_BLOCK = 1024;   // 16 buckets a workgroup: 512 look-back words at 262 144 bodies (256: 2 048 words, +2 us)
__global__ __launch_bounds__(BP_CELLS_BLOCK)
void k_bp_cells(BpK k)
{
    __shared__ uint32_t tot[BP_CELLS_BLOCK / WAVE];
    __shared__ uint32_t excl_s;
    const int lane = lane_id(), wave = threadIdx.x / WAVE;
    const uint32_t b = blockIdx.x * (BP_CELLS_BLOCK / WAVE) + wave;
    uint32_t block_total = 0, c = 0, before = 0;                        // this lane's cell: bodies, bodies of the block's cells before it
    if (b <= k.mask) {
        c = k.cell_cnt[(size_t)b * 64 + lane];
        k.cell_cnt[(size_t)b * 64 + lane] = 0;                          // ready for the next frame
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const uint32_t u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        before = incl - c;
        block_total = __shfl(incl, WAVE - 1);
    }
    if (lane == 0) tot[wave] = block_total;
    __syncthreads();
    if (wave == 0) {
        uint32_t sum = 0;
#pragma unroll
        for (int q = 0; q < BP_CELLS_BLOCK / WAVE; q++) sum += tot[q];
        const uint32_t excl = lb_exclusive(k.lb_cells, blockIdx.x, sum, k.ctrl[CTRL_EPOCH], k.ctrl + CTRL_STATUS);
        if (lane == 0) excl_s = excl;
    }
    __syncthreads();
    if (b <= k.mask) {
        uint32_t start = excl_s;
        for (int q = 0; q < wave; q++) start += tot[q];
        k.cell_range[(size_t)b * 64 + lane] = make_uint2(start + before, c);
    }
}

// Launch 3
__global__ __launch_bounds__(PB)
void k_bp_scatter(BpK k)
{
    const uint32_t i = blockIdx.x * PB + threadIdx.x;
    if (i >= k.n) return;
    const uint32_t slot = k.key[i];
    const uint32_t at = k.cell_range[slot].x + k.rank[i];
    k.entries[at] = i;
    const double2 *p = reinterpret_cast<const double2 *>(k.aabb + 6 * (size_t)i);
    double2 *o = reinterpret_cast<double2 *>(k.recs + at);
    const double2 b0 = p[0], b1 = p[1], b2 = p[2];
    o[0] = b0; o[1] = b1; o[2] = b2;
    // the cell coordinates travel with the record: the search would otherwise redo three fp64 divisions per body
    const int32_t cx = cell_coord((b0.x + b0.y) * 0.5, k.cell), cy = cell_coord((b1.x + b1.y) * 0.5, k.cell),
                  cz = cell_coord((b2.x + b2.y) * 0.5, k.cell);
    reinterpret_cast<int4 *>(o)[3] = make_int4((int)i, cx, cy, cz);
}

// Launch 4.  One wavefront per tile of BP_TILE bodies that are neighbours in cell order.  Bodies of one cell have the
// same 14 candidate cells, so the tile's candidates are listed per DISTINCT cell ("leader": the first body of each run
// of equal cell coordinates), each entry with the range of tile bodies it has to be tested against: a candidate record is
// gathered once per cell, not once per body, and the tile's own boxes are read back from LDS as broadcasts.  The
// statics registered for a block are listed the same way, once per distinct block bucket.  Steps that depend on memory:
// tile records (+ the large statics' records) -> cell / static ranges (up to four per lane, loaded together) -> the work
// list in LDS -> candidate records, two per lane and round -> hit atomics.  Hits on bodies go to the partner list of
// min(i, j) (global atomics); static hits count in LDS, because only this wavefront writes its bodies' static lists.
__global__ __launch_bounds__(PB)
void k_bp_search(BpK k)
{
    constexpr int WAVES = PB / WAVE, T = BP_TILE, LOOKUPS = 4;          // lookups per lane: T * 14 + T <= 64 * LOOKUPS
    constexpr uint32_t WL = BP_WORK, OWN = 0x80000000u, STAT = 0x40000000u, IDX = 0x3fffffffu;
    constexpr int LARGE_TILE = 32;                                      // large statics staged per round (LDS <= 26 KB: six workgroups per CU)
    constexpr uint32_t HITS = 64;                                       // body x body hits parked per round
    static_assert(T * 15 <= WAVE * LOOKUPS && T <= 16, "tile lookups");
    __shared__ __attribute__((aligned(8))) uint32_t work[WAVES][WL][2];  // (record | flags, a_lo | a_hi << 8)
    __shared__ double abox[WAVES][T][6];
    __shared__ uint32_t aidx[WAVES][T];
    __shared__ int32_t lead[WAVES][2 * T][4];                           // cell leaders: (cx, cy, cz, range); then block leaders: (bucket, -, -, range)
    __shared__ uint32_t shits[WAVES][T];
    __shared__ uint32_t hits[WAVES][HITS][2], nhits[WAVES];
    __shared__ BpRec large[LARGE_TILE];
    const int lane = lane_id(), wave = threadIdx.x / WAVE;
    const uint32_t t0 = (blockIdx.x * WAVES + wave) * T;
    const uint32_t nA = t0 < k.n ? (k.n - t0 < (uint32_t)T ? k.n - t0 : (uint32_t)T) : 0;
    const bool statics = k.n_static != 0;

    const uint32_t m0 = statics ? (k.n_large < LARGE_TILE ? k.n_large : LARGE_TILE) : 0;
    if (threadIdx.x < m0) large[threadIdx.x] = k.s_lrecs[threadIdx.x];  // issued with the tile's own records: no extra step
    // ---- the tile's bodies, cell leaders and block leaders
    int32_t cx = 0, cy = 0, cz = 0;
    uint32_t ob = 0;
    const bool isA = (uint32_t)lane < nA;
    if (isA) {
        const BpRec me = k.recs[t0 + lane];
#pragma unroll
        for (int x = 0; x < 6; x++) abox[wave][lane][x] = me.bb[x];
        aidx[wave][lane] = me.idx;
        cx = me.cell[0]; cy = me.cell[1]; cz = me.cell[2];
        ob = block_hash(cx >> 2, cy >> 2, cz >> 2, k.mask);
    }
    if (lane < T) shits[wave][lane] = 0;
    if (lane == 0) nhits[wave] = 0;
    const int32_t px = __shfl_up(cx, 1), py = __shfl_up(cy, 1), pz = __shfl_up(cz, 1);
    const uint32_t pob = __shfl_up(ob, 1);
    const bool cell_leader = isA && (lane == 0 || px != cx || py != cy || pz != cz);
    const bool block_leader = isA && statics && (lane == 0 || pob != ob);
    const uint32_t cmask = (uint32_t)__ballot(cell_leader), bmask = (uint32_t)__ballot(block_leader);
    const uint32_t n_lead = __popc(cmask), n_blead = __popc(bmask);
    if (isA) {
        const uint32_t below = (1u << lane) - 1u;
        if (cell_leader) {
            const uint32_t above = cmask >> (lane + 1);
            const uint32_t hi = above ? lane + 1 + __builtin_ctz(above) : nA;
            int32_t *d = lead[wave][__popc(cmask & below)];
            d[0] = cx; d[1] = cy; d[2] = cz; d[3] = (int32_t)((uint32_t)lane | hi << 8);
        }
        if (block_leader) {
            const uint32_t above = bmask >> (lane + 1);
            const uint32_t hi = above ? lane + 1 + __builtin_ctz(above) : nA;
            int32_t *d = lead[wave][T + __popc(bmask & below)];
            d[0] = (int32_t)ob; d[3] = (int32_t)((uint32_t)lane | hi << 8);
        }
    }
    wave_lds_fence();
    // ---- candidate runs: (leader, cell) and (block leader) lookups, up to four per lane, loads in flight together
    const uint32_t n_cell_runs = n_lead * 14u, n_runs = n_cell_runs + n_blead;
    // All of a lane's lookups are addressed first and loaded together, under no lane test (a lookup that does not exist
    // reads entry 0 and is given length 0): written as "if cell run ... else if block run ..." per lookup, each of the
    // four became its own branch with its own waits -- eight dependent trips to L2 before the first candidate.
    uint32_t b0[LOOKUPS], len[LOOKUPS], rng[LOOKUPS], mylen = 0;
    uint32_t slot[LOOKUPS], sidx[LOOKUPS];
    bool is_cell[LOOKUPS], is_stat[LOOKUPS], own[LOOKUPS];
#pragma unroll
    for (int r = 0; r < LOOKUPS; r++) {
        const uint32_t u = lane + WAVE * r;
        is_cell[r] = u < n_cell_runs;
        is_stat[r] = !is_cell[r] && u < n_runs;
        const uint32_t l = is_cell[r] ? u / 14u : 0u, cq = 13u + u % 14u;           // 13 = own cell, 14..26 = the cells after it
        const int32_t *d = lead[wave][is_stat[r] ? T + (u - n_cell_runs) : l];
        const int32_t d0 = d[0], d1 = d[1], d2 = d[2];
        rng[r] = (is_cell[r] || is_stat[r]) ? (uint32_t)d[3] : 0u;
        own[r] = is_cell[r] && cq == 13u;
        slot[r] = is_cell[r] ? cell_slot(d0 - 1 + (int32_t)(cq % 3), d1 - 1 + (int32_t)((cq / 3) % 3), d2 - 1 + (int32_t)(cq / 9), k.mask) : 0u;
        sidx[r] = is_stat[r] ? (uint32_t)d0 : 0u;
    }
    uint2 v_cr[LOOKUPS];
    uint32_t v_s0[LOOKUPS], v_s1[LOOKUPS];
#pragma unroll
    for (int r = 0; r < LOOKUPS; r++) v_cr[r] = k.cell_range[slot[r]];
    if (statics) {                                                       // uniform
#pragma unroll
        for (int r = 0; r < LOOKUPS; r++) { v_s0[r] = k.s_start[sidx[r]]; v_s1[r] = k.s_start[sidx[r] + 1]; }
    } else {
#pragma unroll
        for (int r = 0; r < LOOKUPS; r++) { v_s0[r] = 0; v_s1[r] = 0; }
    }
#pragma unroll
    for (int r = 0; r < LOOKUPS; r++) {
        b0[r] = is_cell[r] ? (v_cr[r].x | (own[r] ? OWN : 0u)) : is_stat[r] ? (v_s0[r] | STAT) : 0u;
        len[r] = is_cell[r] ? v_cr[r].y : is_stat[r] ? v_s1[r] - v_s0[r] : 0u;
        mylen += len[r];
    }
    uint32_t incl = mylen;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const uint32_t total = __shfl(incl, WAVE - 1);
    const uint32_t first = incl - mylen;

    // A hit on another body needs one returning atomic on the partner count of min(i, j), ~2 us under load: hits are
    // parked in LDS while the candidates are tested and their atomics issued together once per round.
    auto flush_hits = [&]() {
        wave_lds_fence();
        const uint32_t nh = nhits[wave] < HITS ? nhits[wave] : HITS;
        for (uint32_t h = lane; h < nh; h += WAVE) {
            const uint32_t lo = hits[wave][h][0], hi = hits[wave][h][1];
            const uint32_t at = atomicAdd(&k.cnt[lo], 1u);
            if (at < BP_LIST) k.partners[(size_t)lo * BP_LIST + at] = hi;
        }
        wave_lds_fence();
        if (lane == 0) nhits[wave] = 0;
        wave_lds_fence();
    };
    auto test = [&](uint32_t w, uint32_t range, const BpRec &r) {
        const uint32_t j = r.idx, a_hi = range >> 8;
        for (uint32_t x = range & 0xffu; x < a_hi; x++) {
            const double2 *ab = reinterpret_cast<const double2 *>(abox[wave][x]);
            const double2 a01 = ab[0], a23 = ab[1], a45 = ab[2];         // all six before any compare: one LDS wait per body
            const bool apart = (a01.x > r.bb[1]) | (a01.y < r.bb[0]) | (a23.x > r.bb[3]) | (a23.y < r.bb[2]) |
                               (a45.x > r.bb[5]) | (a45.y < r.bb[4]);
            if (apart) continue;
            const uint32_t i = aidx[wave][x];
            if (w & STAT) {
                const uint32_t at = atomicAdd(&shits[wave][x], 1u);
                if (at < BP_LIST) k.spartners[(size_t)i * BP_LIST + at] = j;
            } else if (j != i && (!(w & OWN) || j > i)) {
                const uint32_t lo = i < j ? i : j, hi = i < j ? j : i;
                const uint32_t h = atomicAdd(&nhits[wave], 1u);
                if (h < HITS) { hits[wave][h][0] = lo; hits[wave][h][1] = hi; }
                else {                                                   // list full (a pile-up): straight to memory
                    const uint32_t at = atomicAdd(&k.cnt[lo], 1u);
                    if (at < BP_LIST) k.partners[(size_t)lo * BP_LIST + at] = hi;
                }
            }
        }
    };

    for (uint32_t base = 0; base < total; base += WL) {                 // one round unless > WL candidates (total is wave-uniform)
        uint32_t off = first;
#pragma unroll
        for (int r = 0; r < LOOKUPS; r++) {
            for (uint32_t e = 0; __any(e < len[r]); e++) {
                if (e < len[r]) {
                    const uint32_t o = off + e - base;                  // wraps below base: then >= WL
                    if (o < WL) *reinterpret_cast<uint2 *>(work[wave][o]) = make_uint2(b0[r] + e, rng[r]);
                }
            }
            off += len[r];
        }
        wave_lds_fence();
        const uint32_t todo = total - base < WL ? total - base : WL;
#if BP_SEARCH_IN_FLIGHT == 2
        for (uint32_t e = lane; e < todo + lane; e += 2 * WAVE) {       // wave-uniform trip count; two records in flight per lane
            const bool v0 = e < todo, v1 = e + WAVE < todo;
            const uint2 e0 = v0 ? *reinterpret_cast<const uint2 *>(work[wave][e]) : make_uint2(0u, 0u);
            const uint2 e1 = v1 ? *reinterpret_cast<const uint2 *>(work[wave][e + WAVE]) : make_uint2(0u, 0u);
            const uint32_t w0 = e0.x, g0 = e0.y, w1 = e1.x, g1 = e1.y;
            BpRec r0, r1;
            if (v0) r0 = ((w0 & STAT) ? k.s_recs : k.recs)[w0 & IDX];
            if (v1) r1 = ((w1 & STAT) ? k.s_recs : k.recs)[w1 & IDX];
            if (v0) test(w0, g0, r0);
            if (v1) test(w1, g1, r1);
        }
#else
        for (uint32_t e = lane; e < todo; e += WAVE) {
            const uint2 e0 = *reinterpret_cast<const uint2 *>(work[wave][e]);
            const BpRec r0 = ((e0.x & STAT) ? k.s_recs : k.recs)[e0.x & IDX];
            test(e0.x, e0.y, r0);
        }
#endif
        flush_hits();
    }
    if (!statics) return;
    // ---- the large statics (tested by every body), staged through LDS for the whole workgroup
    for (uint32_t base = 0; base < k.n_large; base += LARGE_TILE) {
        const uint32_t m = k.n_large - base < LARGE_TILE ? k.n_large - base : LARGE_TILE;
        if (base) {
            __syncthreads();
            if (threadIdx.x < m) large[threadIdx.x] = k.s_lrecs[base + threadIdx.x];
        }
        __syncthreads();
        const uint32_t x = lane % T;                                    // tile body of this lane; the large list is strided by WAVE / T
        if (x < nA) {
            double a[6];
#pragma unroll
            for (int y = 0; y < 6; y++) a[y] = abox[wave][x][y];
            for (uint32_t e = lane / T; e < m; e += WAVE / T) {
                double bs[6];
#pragma unroll
                for (int y = 0; y < 6; y++) bs[y] = large[e].bb[y];
                if (boxes_overlap(a, bs)) {
                    const uint32_t at = atomicAdd(&shits[wave][x], 1u);
                    if (at < BP_LIST) k.spartners[(size_t)aidx[wave][x] * BP_LIST + at] = large[e].idx;
                }
            }
        }
    }
    wave_lds_fence();
    if (isA) k.scnt[aidx[wave][lane]] = shits[wave][lane];
}

// all partners of body i (larger index) in ascending order, for a body whose list did not fit its slot: one lane
// walks its 27 cells
template <typename F>
__device__ __forceinline__ void research_body(const BpK &k, uint32_t i, F &&emit_sorted)
{
    double a[6];
    load_box(k.aabb, i, a);
    int32_t cx, cy, cz;
    box_cell(a, k.cell, cx, cy, cz);
    uint32_t last = i;                                                   // partners > i, ascending: repeated minimum search
    for (;;) {
        uint32_t best = 0xffffffffu;
        for (int cq = 0; cq < 27; cq++) {
            const uint32_t slot = cell_slot(cx - 1 + cq % 3, cy - 1 + (cq / 3) % 3, cz - 1 + cq / 9, k.mask);
            const uint2 cr = k.cell_range[slot];
            const uint32_t s0 = cr.x, s1 = cr.x + cr.y;
            for (uint32_t s = s0; s < s1; s++) {
                const uint32_t j = k.entries[s];
                if (j <= last || j >= best) continue;
                double bj[6];
                load_box(k.aabb, j, bj);
                if (boxes_overlap(a, bj)) best = j;
            }
        }
        if (best == 0xffffffffu) break;
        emit_sorted(best);
        last = best;
    }
}

// Launch 5
__global__ __launch_bounds__(BP_EMIT_TILE)
void k_bp_emit(BpK k)
{
    __shared__ uint32_t lds[2][BP_EMIT_TILE / WAVE];
    __shared__ uint32_t tile_excl[2];
    const uint32_t i = blockIdx.x * BP_EMIT_TILE + threadIdx.x;
    const int lane = lane_id(), wave = threadIdx.x / WAVE;
    const bool with_statics = k.n_static != 0;
    uint32_t c[2] = { 0, 0 };
    if (i < k.n) {
        c[0] = k.cnt[i];
        k.cnt[i] = 0;                                                    // ready for the next frame's atomics
        if (with_statics) { c[1] = k.scnt[i]; k.scnt[i] = 0; }
    }
    // the lists travel while the offsets are scanned: their loads do not depend on the look-back
    uint4 pl[2][BP_LIST / 4];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int q = 0; q < BP_LIST / 4; q++) pl[t][q] = make_uint4(0, 0, 0, 0);
    if (c[0] && c[0] <= (uint32_t)BP_LIST) {
        const uint4 *src = reinterpret_cast<const uint4 *>(k.partners + (size_t)BP_LIST * i);
#pragma unroll
        for (int q = 0; q < BP_LIST / 4; q++) if ((uint32_t)(4 * q) < c[0]) pl[0][q] = src[q];
    }
    if (c[1] && c[1] <= (uint32_t)BP_LIST) {
        const uint4 *src = reinterpret_cast<const uint4 *>(k.spartners + (size_t)BP_LIST * i);
#pragma unroll
        for (int q = 0; q < BP_LIST / 4; q++) if ((uint32_t)(4 * q) < c[1]) pl[1][q] = src[q];
    }
    uint32_t incl[2] = { c[0], c[1] };
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        const uint32_t u0 = __shfl_up(incl[0], o), u1 = __shfl_up(incl[1], o);
        if (lane >= o) { incl[0] += u0; incl[1] += u1; }
    }
    if (lane == WAVE - 1) { lds[0][wave] = incl[0]; lds[1][wave] = incl[1]; }
    __syncthreads();
    if (wave < 2) {                                                      // wavefront 0: the body list's offsets; wavefront 1: the statics'
        uint32_t sum = 0;
        for (int qq = 0; qq < BP_EMIT_TILE / WAVE; qq++) sum += lds[wave][qq];
        uint32_t excl = 0;
        if (wave == 0 || with_statics)
            excl = lb_exclusive(wave ? k.lb_static : k.lb_body, blockIdx.x, sum, k.ctrl[CTRL_EPOCH], k.ctrl + CTRL_STATUS);
        if (lane == 0) {
            tile_excl[wave] = excl;
            if (blockIdx.x == gridDim.x - 1) {                           // the last tile's inclusive prefix is the total
                uint32_t *tot = wave ? k.spair_total : k.pair_total;
                if (tot) *tot = (wave == 0 || with_statics) ? excl + sum : 0u;
            }
        }
    }
    __syncthreads();
    uint32_t woff[2] = { 0, 0 };
    for (int qq = 0; qq < wave; qq++) { woff[0] += lds[0][qq]; woff[1] += lds[1][qq]; }
    if (i >= k.n) return;
    // a list of n <= BP_LIST entries, out in ascending order: the rank of an entry = the entries below it
    auto ranked = [&](const uint4 (&l4)[BP_LIST / 4], uint32_t n, uint32_t off, uint2 *out, uint32_t cap) {
        uint32_t v[BP_LIST];
#pragma unroll
        for (int q = 0; q < BP_LIST / 4; q++) { v[4 * q] = l4[q].x; v[4 * q + 1] = l4[q].y; v[4 * q + 2] = l4[q].z; v[4 * q + 3] = l4[q].w; }
#pragma unroll
        for (int e = 0; e < BP_LIST; e++) {
            if ((uint32_t)e < n) {
                uint32_t rank = 0;
#pragma unroll
                for (int f = 0; f < BP_LIST; f++) rank += ((uint32_t)f < n) & (v[f] < v[e]);
                if (off + rank < cap) out[off + rank] = make_uint2(i, v[e]);
            }
        }
    };
    if (c[0]) {
        const uint32_t off = tile_excl[0] + woff[0] + incl[0] - c[0];
        uint2 *out = reinterpret_cast<uint2 *>(k.pairs);
        if (c[0] <= (uint32_t)BP_LIST) ranked(pl[0], c[0], off, out, k.capacity);
        else {
            uint32_t w = 0;
            research_body(k, i, [&](uint32_t j) { if (off + w < k.capacity) out[off + w] = make_uint2(i, j); w++; });
        }
    }
    if (c[1]) {
        const uint32_t off = tile_excl[1] + woff[1] + incl[1] - c[1];
        uint2 *out = reinterpret_cast<uint2 *>(k.spairs);
        if (c[1] <= (uint32_t)BP_LIST) ranked(pl[1], c[1], off, out, k.scapacity);
        else {                                                           // every static of the block + the large ones, ascending
            double a[6];
            load_box(k.aabb, i, a);
            int32_t cx, cy, cz;
            box_cell(a, k.cell, cx, cy, cz);
            const uint32_t ob = block_hash(cx >> 2, cy >> 2, cz >> 2, k.mask);
            const uint32_t s0 = k.s_start[ob], nloc = k.s_start[ob + 1] - s0;
            uint32_t w = 0;
            int64_t last = -1;
            for (;;) {
                uint32_t best = 0xffffffffu;
                for (uint32_t e = 0; e < nloc + k.n_large; e++) {
                    const uint32_t sidx = e < nloc ? k.s_entries[s0 + e] : k.s_large[e - nloc];
                    if ((int64_t)sidx <= last || sidx >= best) continue;
                    double bs[6];
                    load_box(k.s_aabb, sidx, bs);
                    if (boxes_overlap(a, bs)) best = sidx;
                }
                if (best == 0xffffffffu) break;
                if (off + w < k.scapacity) out[off + w] = make_uint2(i, best);
                w++;
                last = best;
            }
        }
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
    k.geom_records = (reinterpret_cast<uintptr_t>(b->geom_records) & 15u) ? nullptr : b->geom_records;
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
    hipLaunchKernelGGL(k_bodies_step<false>, dim3((b->n + PB - 1) / PB), dim3(PB), 0, as_stream(stream), bodies_k(b), wk, h, BinK{});
    CLAPGPU_LAUNCH_CHECK("k_bodies_step");
    return CLAPGPU_OK;
}

// ---------------------------------------------------------------------------------- broadphase object
struct clapgpu_bp {
    uint32_t n_max, buckets, n_static, n_large, n_tiles;
    double cell;
    void *dev;                     // one allocation
    BpK k;                         // device pointers filled in
    // clapgpu_bodies_step_prebin: the step that wrote these boxes has also binned them (key / rank / cell counters / epoch):
    // the next clapgpu_bp_collide over the same array skips its first launch
    const double *prebinned_aabb;
    uint32_t prebinned_n;
};

// The step + the NEXT broadphase's bin pass in one launch (the bin pass reads nothing but the box the step has in
// registers, and its one atomic per body hides under the step's fp64 traffic): -1 launch and the boxes' re-read per substep.
extern "C" int clapgpu_bodies_step_prebin(void *stream, const clapgpu_bodies *b, const clapgpu_world *w, double h, clapgpu_bp *bp)
{
    int rc = check_bodies2(b);
    if (rc) return rc;
    if (!w || !bp) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!b->aabb || b->n > bp->n_max) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (b->n == 0) return CLAPGPU_OK;
    if (bp->prebinned_aabb) {                                    // a step binned already and no collide consumed it: start over
        rc = clapgpu_bp_invalidate(stream, bp);
        if (rc) return rc;
    }
    WorldK2 wk;
    memcpy(&wk, w, sizeof(wk));
    BinK bin = { bp->cell, bp->k.mask, bp->k.key, bp->k.rank, bp->k.cell_cnt, bp->k.ctrl };
    hipLaunchKernelGGL(k_bodies_step<true>, dim3((b->n + PB - 1) / PB), dim3(PB), 0, as_stream(stream), bodies_k(b), wk, h, bin);
    CLAPGPU_LAUNCH_CHECK("k_bodies_step<prebin>");
    bp->prebinned_aabb = b->aabb;
    bp->prebinned_n = b->n;
    return CLAPGPU_OK;
}

// The boxes a step pre-binned were changed by somebody else (clapgpu_bodies_aabb, an upload, another body count): the cell
// counters go back to zero -- what k_bp_cells leaves between frames -- and the next collide bins for itself.
extern "C" int clapgpu_bp_invalidate(void *stream, clapgpu_bp *bp)
{
    if (!bp) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!bp->prebinned_aabb) return CLAPGPU_OK;
    bp->prebinned_aabb = nullptr; bp->prebinned_n = 0;
    CLAPGPU_HIP(hipMemsetAsync(bp->k.cell_cnt, 0, (size_t)bp->buckets * 64 * sizeof(uint32_t), as_stream(stream)));
    return CLAPGPU_OK;
}

static uint32_t buckets_for(uint32_t n)
{
    uint32_t b = 1024;                                                  // block buckets: 64 cell slots each, about two slots per body
    while (b < n / 32 && b < (1u << 22)) b <<= 1;
    return b;
}

static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" int clapgpu_bp_create(clapgpu_bp **out, uint32_t n_max, double cell, uint32_t n_static, const double *static_aabb)
{
    if (!out || !(cell > 0.0) || (n_static && !static_aabb) || n_max > (1u << 30))
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
    std::vector<BpRec> s_recs(s_entries.size()), s_lrecs(s_large.size() ? s_large.size() : 1);
    auto fill_rec = [&](BpRec &r, uint32_t sidx) {
        memset(&r, 0, sizeof(r));
        if (n_static) memcpy(r.bb, static_aabb + 6 * (size_t)sidx, sizeof(r.bb));
        r.idx = sidx;
    };
    for (size_t e = 0; e < ins.size(); e++) fill_rec(s_recs[e], s_entries[e]);
    for (size_t e = 0; e < s_large.size(); e++) fill_rec(s_lrecs[e], s_large[e]);
    if (ins.empty()) memset(&s_recs[0], 0, sizeof(BpRec));
    if (s_large.empty()) { memset(&s_lrecs[0], 0, sizeof(BpRec)); s_large.push_back(0); }

    // one device allocation, carved
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes); return o; };
    const size_t o_ccnt = take(4 * (size_t)nb * 64), o_crange = take(8 * (size_t)nb * 64);
    const size_t o_key = take(4 * (size_t)n), o_ranks = take(4 * (size_t)n), o_entries = take(4 * (size_t)n), o_recs = take(64 * (size_t)n);
    const size_t o_cnt = take(4 * (size_t)n), o_scnt = take(4 * (size_t)n);
    const size_t o_part = take(4 * (size_t)BP_LIST * n), o_spart = take(4 * (size_t)BP_LIST * n);
    const size_t o_lbb = take(8 * (size_t)bp->n_tiles), o_lbs = take(8 * (size_t)bp->n_tiles), o_lbc = take(8 * ((size_t)nb / 4 + 1));
    const size_t o_ctrl = take(4 * 160);
    const size_t o_sstart = take(4 * ((size_t)nb + 1)), o_sent = take(4 * s_entries.size()), o_slarge = take(4 * s_large.size());
    const size_t o_saabb = take(48 * (size_t)(n_static ? n_static : 1));
    const size_t o_srecs = take(sizeof(BpRec) * s_recs.size()), o_slrecs = take(sizeof(BpRec) * s_lrecs.size());
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
        hipMemcpy(d + o_srecs, s_recs.data(), sizeof(BpRec) * s_recs.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + o_slrecs, s_lrecs.data(), sizeof(BpRec) * s_lrecs.size(), hipMemcpyHostToDevice) != hipSuccess ||
        (n_static && hipMemcpy(d + o_saabb, static_aabb, 48 * (size_t)n_static, hipMemcpyHostToDevice) != hipSuccess)) {
        (void)hipGetLastError();
        (void)hipFree(bp->dev);
        free(bp);
        return CLAPGPU_ERR_UNKNOWN;
    }
    BpK &k = bp->k;
    memset(&k, 0, sizeof(k));
    k.cell = cell; k.mask = nb - 1;
    k.cell_cnt = reinterpret_cast<uint32_t *>(d + o_ccnt); k.cell_range = reinterpret_cast<uint2 *>(d + o_crange);
    k.key = reinterpret_cast<uint32_t *>(d + o_key); k.rank = reinterpret_cast<uint32_t *>(d + o_ranks);
    k.entries = reinterpret_cast<uint32_t *>(d + o_entries); k.recs = reinterpret_cast<BpRec *>(d + o_recs);
    k.cnt = reinterpret_cast<uint32_t *>(d + o_cnt); k.scnt = reinterpret_cast<uint32_t *>(d + o_scnt);
    k.partners = reinterpret_cast<uint32_t *>(d + o_part); k.spartners = reinterpret_cast<uint32_t *>(d + o_spart);
    k.lb_body = reinterpret_cast<uint64_t *>(d + o_lbb); k.lb_static = reinterpret_cast<uint64_t *>(d + o_lbs);
    k.lb_cells = reinterpret_cast<uint64_t *>(d + o_lbc);
    k.ctrl = reinterpret_cast<uint32_t *>(d + o_ctrl);
    k.s_start = reinterpret_cast<const uint32_t *>(d + o_sstart); k.s_entries = reinterpret_cast<const uint32_t *>(d + o_sent);
    k.s_large = reinterpret_cast<const uint32_t *>(d + o_slarge); k.s_aabb = reinterpret_cast<const double *>(d + o_saabb);
    k.s_recs = reinterpret_cast<const BpRec *>(d + o_srecs); k.s_lrecs = reinterpret_cast<const BpRec *>(d + o_slrecs);
    k.n_large = bp->n_large; k.n_static = n_static;
    *out = bp;
    return CLAPGPU_OK;
}

extern "C" void clapgpu_bp_destroy(clapgpu_bp *bp)
{
    if (!bp) return;
    if (bp->dev) (void)hipFree(bp->dev);
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
        return clapgpu_bp_invalidate(stream, bp);
    }
    BpK k = bp->k;
    k.n = n; k.aabb = aabb; k.n_tiles = (n + BP_EMIT_TILE - 1) / BP_EMIT_TILE;
    k.pairs = pairs; k.capacity = capacity; k.pair_total = pair_total;
    k.spairs = static_pairs; k.scapacity = static_capacity; k.spair_total = static_pair_total;
    if (!statics) { k.n_static = 0; k.n_large = 0; if (static_pair_total) CLAPGPU_HIP(hipMemsetAsync(static_pair_total, 0, 4, s)); }
    if (bp->prebinned_aabb == aabb && bp->prebinned_n == n) {
        bp->prebinned_aabb = nullptr; bp->prebinned_n = 0;       // the step that wrote these boxes binned them: launch 1 is done
    } else {
        if (bp->prebinned_aabb) {                                // binned for other boxes: undo
            int rc = clapgpu_bp_invalidate(stream, bp);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_bp_bin, dim3((n + PB - 1) / PB), dim3(PB), 0, s, k);
        CLAPGPU_LAUNCH_CHECK("k_bp_bin");
    }
    hipLaunchKernelGGL(k_bp_cells, dim3((bp->buckets + BP_CELLS_BLOCK / WAVE - 1) / (BP_CELLS_BLOCK / WAVE)), dim3(BP_CELLS_BLOCK), 0, s, k);
    CLAPGPU_LAUNCH_CHECK("k_bp_cells");
    hipLaunchKernelGGL(k_bp_scatter, dim3((n + PB - 1) / PB), dim3(PB), 0, s, k);
    CLAPGPU_LAUNCH_CHECK("k_bp_scatter");
    hipLaunchKernelGGL(k_bp_search, dim3((n + (PB / WAVE) * BP_TILE - 1) / ((PB / WAVE) * BP_TILE)), dim3(PB), 0, s, k);
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

// contacts.hip's one-launch form keeps its ticket + counts in this object's control words
__attribute__((visibility("hidden"))) unsigned long long *clapgpu_bp_contact_ticket(clapgpu_bp *bp)
{
    static_assert((CTRL_CONTACT_WORD * sizeof(uint32_t)) % 8 == 0, "the ticket word is a 64-bit atomic");
    return reinterpret_cast<unsigned long long *>(bp->k.ctrl + CTRL_CONTACT_WORD);
}
