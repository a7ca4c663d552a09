// contacts.hip -- narrowphase contact records and the capsule sweep, for gfx950 (split from physics2.hip so that this
// translation unit alone is built with -mllvm -simplifycfg-sink-common=false, see the Makefile).
//
//   k_contacts_geoms[_both]  near_callback's dCollide + phys_contact_surface (physics.c:399-449, 291-330)
//   k_sweep_capsules         phys_body_sweep_capsule (physics.c:559-670), one wavefront per sweep
//
// Why the flag: phd::collide() writes its (up to two) contacts through CGeom references.  After inlining, LLVM's
// SimplifyCFG sinks the "same" stores of different call sites into one block that stores through a SELECTED pointer
// (c0 or c1), which keeps both contacts addressable: 56 bytes (contacts) / 128 bytes (sweep) of scratch per lane, and
// scratch is HBM traffic on gfx950.  Without the sinking SROA turns them into registers (private segment 0, +6 VGPRs).
// fp64 throughout, no FMA contraction.  ODE is an absent submodule of the reference: PARITY UNPINNED.
#include <string.h>
#include <stdlib.h>
#include "common.h"
#include "phys_dev.h"

struct clapgpu_bp;
unsigned long long *clapgpu_bp_contact_ticket(clapgpu_bp *bp);       // physics2.hip

namespace clapgpu {

constexpr int PB = 256;

// ================================================================================== narrowphase
struct GeomsK {
    uint32_t n;
    const double *pos, *axis, *radius, *length, *aabb, *material;
    const uint8_t *kind;
    const double *rec;               // [n][8] (pos, axis, radius, length): only for sets without kind / aabb
};

__device__ __forceinline__ void load_geom(const GeomsK &g, uint32_t i, phd::Geom &o)
{
    if (g.rec) {                                                 // spheres and capsules: the whole geom in one 64-byte record
        const double2 *r = reinterpret_cast<const double2 *>(g.rec + 8 * (size_t)i);
        const double2 a = r[0], b = r[1], c = r[2], d = r[3];
        o.pos[0] = a.x; o.pos[1] = a.y; o.pos[2] = b.x;
        o.axis[0] = b.y; o.axis[1] = c.x; o.axis[2] = c.y;
        o.radius = d.x; o.length = d.y;
        o.kind = d.y != 0.0 ? CLAPGPU_GEOM_CAPSULE : CLAPGPU_GEOM_SPHERE;
        for (int k = 0; k < 6; k++) o.aabb[k] = 0.0;
        return;
    }
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

// one candidate pair -> its record; true if the pair produced contacts (or is flagged deep)
__device__ __forceinline__ bool contact_of_pair(const GeomsK &A, const GeomsK &B, const uint2 pr, clapgpu_contact2 &c,
                                                uint32_t *flags_a, uint32_t *flags_b)
{
    bool counted = false;
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
            counted = true;
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
            counted = true;
            // plain read-modify-write: every writer of this launch sets the same bit and nothing else changes the word
            if (flags_a && !(flags_a[pr.x] & CLAPGPU_BODY_HAS_JOINT)) flags_a[pr.x] |= CLAPGPU_BODY_HAS_JOINT;
            if (flags_b && !(flags_b[pr.y] & CLAPGPU_BODY_HAS_JOINT)) flags_b[pr.y] |= CLAPGPU_BODY_HAS_JOINT;
        }
    }
    return counted;
}

// 64 consecutive pairs of one list on one wavefront: each lane's 160-byte record goes through a wave-private LDS tile
// (rows padded to 176 bytes: the 16-byte writes of eight neighbouring lanes then fall on all 32 banks) and leaves as ten
// 1 KiB stores -- written per lane, ten 16-byte pieces at a 160-byte stride touched 64 cache lines per instruction.
constexpr int CONTACT_ROW = 11;                                          // uint4 per staged record (10 used)
__device__ __forceinline__ uint32_t contacts_chunk(const GeomsK &A, const GeomsK &B, const uint2 *pairs, uint32_t p0, uint32_t np,
                                                    clapgpu_contact2 *out, uint32_t *flags_a, uint32_t *flags_b, uint4 *tile)
{
    static_assert(sizeof(clapgpu_contact2) == 160, "ten 16-byte pieces");
    const int lane = lane_id();
    const uint32_t p = p0 + lane;
    clapgpu_contact2 c;
    memset(&c, 0, sizeof(c));
    uint32_t counted = 0;
    if (p < np) counted = contact_of_pair(A, B, pairs[p], c, flags_a, flags_b);
    uint4 v[10];
    memcpy(v, &c, sizeof(c));
#pragma unroll
    for (int k = 0; k < 10; k++) tile[lane * CONTACT_ROW + k] = v[k];
    wave_lds_fence();
    const uint32_t pieces = (np - p0 < (uint32_t)WAVE ? np - p0 : (uint32_t)WAVE) * 10u;
    uint4 *o = reinterpret_cast<uint4 *>(out + p0);
#pragma unroll
    for (int k = 0; k < 10; k++) {
        const uint32_t idx = (uint32_t)(k * WAVE + lane);
        if (idx < pieces) o[idx] = tile[(idx / 10u) * CONTACT_ROW + idx % 10u];
    }
    wave_lds_fence();
    return counted;
}

// ---- the one-launch form's loop: a wavefront takes several chunks, and a chunk's inputs are asked for while the chunk
// before it is still being worked on (the chain per chunk is pair -> two geoms + two flag words -> arithmetic -> record;
// measured: a wavefront of the one-chunk-per-wavefront kernel lives ~17 us, two thirds of it waiting, and the kernel is
// two such rounds).  The next chunk's PAIR is requested before this chunk's arithmetic; the flag words come with the geoms.
// (The next chunk's geoms as well, under this chunk's stores: 224 VGPRs and 152 bytes of scratch -- not kept.)
struct PairInputs { uint2 pr; bool live; phd::Geom ga, gb; uint32_t fa, fb; };

__device__ __forceinline__ void load_pair_inputs(const GeomsK &A, const GeomsK &B, uint2 pr, bool in_range, uint32_t *flags_a,
                                                 uint32_t *flags_b, PairInputs &in)
{
    in.pr = pr;
    in.live = in_range && pr.x < A.n && pr.y < B.n;
    in.fa = in.fb = CLAPGPU_BODY_HAS_JOINT;
    if (in.live) {
        load_geom(A, pr.x, in.ga);
        load_geom(B, pr.y, in.gb);
        // the flag words early: every writer of this launch sets the same bit and nothing else changes the words
        if (flags_a) in.fa = flags_a[pr.x];
        if (flags_b) in.fb = flags_b[pr.y];
    }
}

__device__ __forceinline__ uint32_t contact_from_inputs(const GeomsK &A, const GeomsK &B, const PairInputs &in, clapgpu_contact2 &c,
                                                        uint32_t *flags_a, uint32_t *flags_b)
{
    if (!in.live) return 0;
    phd::CGeom c0, c1;
    memset(&c0, 0, sizeof(c0));
    memset(&c1, 0, sizeof(c1));
    const int nc = phd::collide(in.ga, in.gb, c0, c1);
    if (nc < 0) { c.nc = CLAPGPU_CONTACT_DEEP; return 1; }
    if (nc == 0) return 0;
    for (int a = 0; a < 3; a++) { c.pos[a] = c0.pos[a]; c.normal[a] = c0.normal[a]; }
    c.depth = c0.depth;
    if (nc > 1) {
        for (int a = 0; a < 3; a++) { c.pos2[a] = c1.pos[a]; c.normal2[a] = c1.normal[a]; }
        c.depth2 = c1.depth;
    }
    contact_surface2(c, (A.material && B.material) ? A.material + 5 * (size_t)in.pr.x : nullptr,
                     (A.material && B.material) ? B.material + 5 * (size_t)in.pr.y : nullptr);
    c.nc = (uint32_t)nc;
    if (flags_a && !(in.fa & CLAPGPU_BODY_HAS_JOINT)) flags_a[in.pr.x] = in.fa | CLAPGPU_BODY_HAS_JOINT;
    if (flags_b && !(in.fb & CLAPGPU_BODY_HAS_JOINT)) flags_b[in.pr.y] = in.fb | CLAPGPU_BODY_HAS_JOINT;
    return 1;
}

// a chunk's 64 records through the wave-private tile and out as ten 1 KiB stores (see contacts_chunk)
__device__ __forceinline__ void store_chunk(const clapgpu_contact2 &c, clapgpu_contact2 *out, uint32_t p0, uint32_t np, uint4 *tile)
{
    const int lane = lane_id();
    uint4 v[10];
    memcpy(v, &c, sizeof(c));
#pragma unroll
    for (int k = 0; k < 10; k++) tile[lane * CONTACT_ROW + k] = v[k];
    wave_lds_fence();
    const uint32_t pieces = (np - p0 < (uint32_t)WAVE ? np - p0 : (uint32_t)WAVE) * 10u;
    uint4 *o = reinterpret_cast<uint4 *>(out + p0);
#pragma unroll
    for (int k = 0; k < 10; k++) {
        const uint32_t idx = (uint32_t)(k * WAVE + lane);
        if (idx < pieces) o[idx] = tile[(idx / 10u) * CONTACT_ROW + idx % 10u];
    }
    wave_lds_fence();
}

__device__ __forceinline__ uint32_t chunk_from_pair(const GeomsK &A, const GeomsK &B, uint2 pr, bool in_range, uint32_t p0, uint32_t np,
                                                     clapgpu_contact2 *out, uint32_t *flags_a, uint32_t *flags_b, uint4 *tile)
{
    PairInputs in;
    load_pair_inputs(A, B, pr, in_range, flags_a, flags_b, in);
    clapgpu_contact2 c;
    memset(&c, 0, sizeof(c));
    const uint32_t counted = contact_from_inputs(A, B, in, c, flags_a, flags_b);
    store_chunk(c, out, p0, np, tile);
    return counted;
}

__global__ __launch_bounds__(PB)
void k_contacts_geoms(GeomsK A, GeomsK B, const uint2 *pairs, const uint32_t *pair_total, uint32_t capacity,
                      clapgpu_contact2 *out, uint32_t *contact_total, uint32_t *flags_a, uint32_t *flags_b)
{
    __shared__ uint32_t block_hits;
    __shared__ uint4 tile[PB / WAVE][WAVE * CONTACT_ROW];
    if (threadIdx.x == 0) block_hits = 0;
    __syncthreads();
    uint32_t np = *pair_total;
    if (np > capacity) np = capacity;
    uint32_t mine = 0;
    const uint32_t wave = threadIdx.x / WAVE;
    for (uint32_t p0 = blockIdx.x * PB + wave * WAVE; p0 < np; p0 += gridDim.x * PB)      // wave-uniform
        mine += contacts_chunk(A, B, pairs, p0, np, out, flags_a, flags_b, tile[wave]);
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane_id() == 0 && mine) atomicAdd(&block_hits, mine);
    __syncthreads();
    if (threadIdx.x == 0 && block_hits && contact_total) atomicAdd(contact_total, block_hits);
}

// near_callback over BOTH lists of a step (bodies x bodies, bodies x statics: physics.c:751-753) in one launch, and
// without a cleared counter in front of it: a workgroup adds (1, its static count, its body count) to ONE 64-bit word with one
// atomic; the workgroup that finds every other ticket already taken holds the totals in what came back, stores them and
// leaves the word at zero for the next launch.  Two launches and two counter fills were 62 us of a frame for 47 us of work.
#ifndef CONTACTS_BOTH_WAVES
#define CONTACTS_BOTH_WAVES 2                                            // wavefronts a SIMD (174 VGPRs, nothing spilled; 3 = 168 VGPRs + 24 B of scratch: 0.3 us faster)
#endif
__global__ __launch_bounds__(PB) __attribute__((amdgpu_waves_per_eu(CONTACTS_BOTH_WAVES, CONTACTS_BOTH_WAVES)))
void k_contacts_geoms_both(GeomsK A, GeomsK B, const uint2 *pairs, const uint32_t *pair_total, uint32_t capacity,
                           clapgpu_contact2 *out, uint32_t *contact_total, const uint2 *spairs, const uint32_t *spair_total,
                           uint32_t scapacity, clapgpu_contact2 *sout, uint32_t *scontact_total, uint32_t *flags,
                           unsigned long long *word)
{
    __shared__ uint32_t block_hits[2];
    __shared__ uint4 tile[PB / WAVE][WAVE * CONTACT_ROW];
    if (threadIdx.x < 2) block_hits[threadIdx.x] = 0;
    __syncthreads();
    uint32_t nb = *pair_total, ns = spair_total ? *spair_total : 0u;
    if (nb > capacity) nb = capacity;
    if (ns > scapacity) ns = scapacity;
    uint32_t mine_b = 0, mine_s = 0;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));   // in an SGPR: what is selected by chunk is scalar
    const uint32_t cb = (nb + WAVE - 1) / WAVE, cs = (ns + WAVE - 1) / WAVE;   // 64-pair chunks: the bodies' list, then the statics'
    const int lane = lane_id();
    const uint32_t stride = gridDim.x * (PB / WAVE);
    // the pair a lane takes from chunk `c` (wave-uniform choice of list: the bodies', then the statics')
    auto pair_of = [&](uint32_t c, uint2 &pr) {
        const uint2 *list = c < cb ? pairs : spairs;
        const uint32_t p = (c < cb ? c : c - cb) * WAVE + lane, n = c < cb ? nb : ns;
        pr = make_uint2(0, 0);
        if (p < n) pr = list[p];
        return p < n;
    };
    uint32_t ch = blockIdx.x * (PB / WAVE) + wave;
    uint2 pr = make_uint2(0, 0);
    bool pin = ch < cb + cs && pair_of(ch, pr);
    while (ch < cb + cs) {
        const uint2 cur_pr = pr;
        const bool cur_in = pin;
        const uint32_t next = ch + stride;
        if (next < cb + cs) pin = pair_of(next, pr);                     // the next chunk's pair: under this chunk's geoms and arithmetic
        if (ch < cb) mine_b += chunk_from_pair(A, A, cur_pr, cur_in, ch * WAVE, nb, out, flags, flags, tile[wave]);
        else mine_s += chunk_from_pair(A, B, cur_pr, cur_in, (ch - cb) * WAVE, ns, sout, flags, nullptr, tile[wave]);
        ch = next;
    }
    for (int o = 32; o > 0; o >>= 1) { mine_b += __shfl_xor(mine_b, o); mine_s += __shfl_xor(mine_s, o); }
    if (lane_id() == 0) {
        if (mine_b) atomicAdd(&block_hits[0], mine_b);
        if (mine_s) atomicAdd(&block_hits[1], mine_s);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long add = (1ull << 48) | ((unsigned long long)block_hits[1] << 24) | block_hits[0];
        const unsigned long long old = atomicAdd(word, add);
        if ((uint32_t)(old >> 48) == gridDim.x - 1) {                   // the last ticket: `old` holds everybody else's counts
            if (contact_total) *contact_total = (uint32_t)(old & 0xffffffu) + block_hits[0];
            if (scontact_total) *scontact_total = (uint32_t)((old >> 24) & 0xffffffu) + block_hits[1];
            *word = 0;                                                   // ready for the next launch (stream order)
        }
    }
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
                phd::CGeom cg0, cg1;                                          // (two locals, not an array: nothing indexes them)
                memset(&cg0, 0, sizeof(cg0));
                memset(&cg1, 0, sizeof(cg1));
                bool is_body = false;
                uint32_t id = 0;
                if (kk < c1) {
                    const uint32_t cv = cand[kk];
                    is_body = (cv >> 31) != 0;
                    id = cv & 0x7fffffffu;
                    if (!(is_body && id == self) && id < (is_body ? A.n : B.n)) {
                        phd::Geom other;
                        load_geom(is_body ? A : B, id, other);
                        nc = phd::collide(probe, other, cg0, cg1);
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
                auto take = [&](const phd::CGeom &g, uint32_t i) {
                    if (first + i >= 16) return;
                    const float cn[3] = { (float)g.normal[0], (float)g.normal[1], (float)g.normal[2] };
                    const float ndot = dir[0] * cn[0] + dir[1] * cn[1] + dir[2] * cn[2];
                    if (ndot > -0.1f) return;
                    const float backup = (float)(g.depth / -ndot);
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
                };
                if (nc > 0) take(cg0, 0);
                if (nc > 1) take(cg1, 1);
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

static GeomsK geoms_k(const clapgpu_geoms *g)
{
    GeomsK k;
    k.n = g->n; k.pos = g->pos; k.axis = g->axis; k.radius = g->radius; k.length = g->length; k.aabb = g->aabb;
    k.material = g->material; k.kind = g->kind;
    // the one-sector records stand in for (pos, axis, radius, length) of sphere / capsule sets only
    k.rec = (g->records && !g->kind && !g->aabb && !(reinterpret_cast<uintptr_t>(g->records) & 15u)) ? g->records : nullptr;
    return k;
}

// workgroups of PB threads of `kernel` that fit the device at once (`cached`: per kernel, asked once).
// CLAPGPU_CONTACTS_GRID, read at every call, overrides it: the A/B knob, and how the tests make every wavefront walk many chunks
static uint32_t resident_workgroups(const void *kernel, uint32_t *cached)
{
    const char *g = getenv("CLAPGPU_CONTACTS_GRID");
    if (g && atoi(g) > 0) return (uint32_t)atoi(g);
    if (!*cached) {
        int per_cu = 0, cus = 0, dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, PB, 0) != hipSuccess)
            *cached = 2048;
        else
            *cached = (uint32_t)(per_cu > 0 ? per_cu : 1) * (uint32_t)(cus > 0 ? cus : 1);
    }
    return *cached;
}

extern "C" int clapgpu_contacts_geoms(void *stream, const clapgpu_geoms *A, const clapgpu_geoms *B, const uint32_t *pairs,
                                      const uint32_t *pair_total, uint32_t capacity, clapgpu_contact2 *contacts,
                                      uint32_t *contact_total, uint32_t *body_flags_a, uint32_t *body_flags_b)
{
    if (!A || !B || !pair_total || (capacity && (!pairs || !contacts)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (reinterpret_cast<uintptr_t>(contacts) & 15u)                    // the records leave as 16-byte pieces (contacts_chunk)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    hipStream_t s = as_stream(stream);
    if (contact_total)
        CLAPGPU_HIP(hipMemsetAsync(contact_total, 0, sizeof(uint32_t), s));
    if (capacity == 0 || A->n == 0 || B->n == 0)
        return CLAPGPU_OK;
    const uint32_t blocks = (capacity + PB - 1) / PB;
    static uint32_t cached;                                              // (see clapgpu_contacts_geoms_both)
    const uint32_t resident = resident_workgroups(reinterpret_cast<const void *>(k_contacts_geoms), &cached);
    hipLaunchKernelGGL(k_contacts_geoms, dim3(blocks < resident ? blocks : resident), dim3(PB), 0, s, geoms_k(A), geoms_k(B),
                       reinterpret_cast<const uint2 *>(pairs), pair_total, capacity, contacts, contact_total, body_flags_a,
                       body_flags_b);
    CLAPGPU_LAUNCH_CHECK("k_contacts_geoms");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_contacts_geoms_both(void *stream, clapgpu_bp *bp, const clapgpu_geoms *bodies, const clapgpu_geoms *statics,
                                           const uint32_t *pairs, const uint32_t *pair_total, uint32_t capacity,
                                           clapgpu_contact2 *contacts, uint32_t *contact_total,
                                           const uint32_t *static_pairs, const uint32_t *static_pair_total, uint32_t static_capacity,
                                           clapgpu_contact2 *static_contacts, uint32_t *static_contact_total, uint32_t *body_flags)
{
    if (!bp || !bodies || !statics || !pair_total || !static_pair_total || (capacity && (!pairs || !contacts)) ||
        (static_capacity && (!static_pairs || !static_contacts)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if ((reinterpret_cast<uintptr_t>(contacts) | reinterpret_cast<uintptr_t>(static_contacts)) & 15u)   // 16-byte pieces
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (capacity >= (1u << 24) || static_capacity >= (1u << 24))         // the counts travel as 24-bit fields of one word
        return CLAPGPU_ERR_TOO_LARGE;
    hipStream_t s = as_stream(stream);
    if (bodies->n == 0 || (capacity == 0 && static_capacity == 0)) {
        if (contact_total) CLAPGPU_HIP(hipMemsetAsync(contact_total, 0, sizeof(uint32_t), s));
        if (static_contact_total) CLAPGPU_HIP(hipMemsetAsync(static_contact_total, 0, sizeof(uint32_t), s));
        return CLAPGPU_OK;
    }
    const uint32_t blocks = (capacity + static_capacity + PB - 1) / PB;
    // as many workgroups as are resident at once: a wavefront then walks its chunks with the next one's inputs in flight
    static uint32_t cached;
    const uint32_t resident = resident_workgroups(reinterpret_cast<const void *>(k_contacts_geoms_both), &cached);
    hipLaunchKernelGGL(k_contacts_geoms_both, dim3(blocks < resident ? blocks : resident), dim3(PB), 0, s, geoms_k(bodies), geoms_k(statics),
                       reinterpret_cast<const uint2 *>(pairs), pair_total, capacity, contacts, contact_total,
                       reinterpret_cast<const uint2 *>(static_pairs), static_pair_total, statics->n ? static_capacity : 0u,
                       static_contacts, static_contact_total, body_flags,
                       clapgpu_bp_contact_ticket(bp));
    CLAPGPU_LAUNCH_CHECK("k_contacts_geoms_both");
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
