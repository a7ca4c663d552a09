// entities_row.h -- what the entity kernels share: the kernel-argument structs, the per-row input loads and
// process_row (one 64-entity row on one wavefront).  Included by entities.hip (the level / tile kernels) and by
// entities_host.hip (the one-launch small frame of a host mirror); two translation units on purpose: with the small-frame
// kernel in the same module the compiler scheduled k_entities_tiles differently (8 more instructions, registers
// renumbered), and that kernel's store / wait structure is measured work that no other feature may move.
#pragma once
#include <string.h>
#include <math.h>
#include "common.h"
#include "lm_dev.h"

namespace clapgpu {

struct EntK {                    // kernel-argument copy of clapgpu_entities
    const float4   *pos_scale;
    const float4   *rot;
    const int32_t  *parent;
    const int32_t  *model;
    const float4   *model_table;
    uint32_t       *flags;
    uint32_t       *seqs;
    float          *mx;
    float          *inv_mx;
    float          *aabb;
    float          *center;
    uint64_t       *vis_mask;
    uint8_t        *vis_row_pop;
    uint32_t        n_attach;
    const clapgpu_attach *attach;
    const float    *jt_pool;
    const float    *bind_pool;
    float          *attach_local;    // [n_attach] mat4 = (jt * bind) * local, written by k_attach_prepare
    // camera bounding-volume query (bv_on == 0: off); bv_result may be NULL when only the containment mask is wanted
    float           bv_cam[3], bv_ctl[3];
    uint32_t        bv_has_ctl, bv_ctl_entity, bv_on;
    unsigned long long *bv_result;
    uint64_t       *bv_inside;       // optional: one bit per entity, set where the query's boxes contain the point(s)
    uint64_t       *rebuilt_mask;    // optional: one bit per entity, set where this launch rebuilt the entity
    uint32_t        n, n_models;     // bounds of the two indices the caller supplies per entity
};

// clapgpu_views as the kernels take it (by value, in the kernarg segment: a view's planes come through the scalar cache)
struct XViewsK {
    uint32_t n, pad;
    uint64_t *mask[CLAPGPU_EXTRA_VIEWS_MAX];
    uint8_t  *pop[CLAPGPU_EXTRA_VIEWS_MAX];
    uint64_t *o_mask[CLAPGPU_EXTRA_VIEWS_MAX];       // HOST kernels: the mirror's mapped words, or NULL
    lmd::FrustumK fr[CLAPGPU_EXTRA_VIEWS_MAX];
};

// The further views of a row whose main view has just been tested: `base` = the lanes that are ALIVE and VISIBLE (the
// draw predicate without its frustum term), bb = the boxes the main view was tested against.  Returns the union of the
// views' masks (what a host mirror counts as drawn).
// aabb_in_frustum_fast() for a further view, without its early exits: the six plane chains and the six extreme compares are
// all evaluated and AND-ed (the same fp32 operations on the same operands, so the same verdict; twelve fewer branch points a
// view and row), on the per-axis extremes of the box computed once per row.  Non-finite boxes or planes: the literal form.
__device__ __forceinline__ bool extra_view_test(const lmd::FrustumK &k, const float (&bb)[6], const float (&lo)[3], const float (&hi)[3],
                                                const bool box_finite)
{
    if (!(k.finite && box_finite))
        return lmd::aabb_in_frustum(k.f, bb);
    bool in = true;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const float x = (k.f.planes[i][0] >= 0.f) ? hi[0] : lo[0];
        const float y = (k.f.planes[i][1] >= 0.f) ? hi[1] : lo[1];
        const float z = (k.f.planes[i][2] >= 0.f) ? hi[2] : lo[2];
        float p = 0.f;
        p += x * k.f.planes[i][0];
        p += y * k.f.planes[i][1];
        p += z * k.f.planes[i][2];
        p += 1.0f * k.f.planes[i][3];
        in = in & !(p < 0.0f);
    }
#pragma unroll
    for (int ax = 0; ax < 3; ax++)
        in = in & !(k.cmin[ax] > bb[3 + ax]) & !(k.cmax[ax] < bb[ax]);
    return in;
}

__device__ __forceinline__ uint64_t cull_extra_views(const XViewsK &xv, const bool base, const uint32_t fl, const float (&bb)[6],
                                                     const uint32_t word, const int lane)
{
    uint64_t any = 0;
    // (fmin / fmax: a stored box with max < min is still handled exactly -- lm_dev.h)
    const float lo[3] = { fminf(bb[0], bb[3]), fminf(bb[1], bb[4]), fminf(bb[2], bb[5]) };
    const float hi[3] = { fmaxf(bb[0], bb[3]), fmaxf(bb[1], bb[4]), fmaxf(bb[2], bb[5]) };
    bool box_finite = true;
#pragma unroll
    for (int a = 0; a < 6; a++) box_finite = box_finite && (fabsf(bb[a]) <= 3.402823466e+38f);
    const bool tested = base && !(fl & CLAPGPU_E_SKIP_CULLING);
    for (uint32_t v = 0; v < xv.n; v++) {                        // uniform: a view's planes and extremes are scalar loads (NOT unrolled: 3x the code, 15 us slower)
        const bool vis = tested ? extra_view_test(xv.fr[v], bb, lo, hi, box_finite) : base;
        const uint64_t m = __ballot(vis);
        if (lane == 0) {
            xv.mask[v][word] = m;
            xv.pop[v][word] = (uint8_t)__popcll(m);
            if (xv.o_mask[v]) xv.o_mask[v][word] = m;
        }
        any |= m;
    }
    return any;
}

constexpr int ENT_BLOCK = 256;
constexpr int LDS_F4_PER_WAVE = 512;                             // 8 KiB of wave-private LDS: the rows' store staging

__device__ __forceinline__ void load_mat4(float (&m)[16], const float *src)
{
    const float4 *p = reinterpret_cast<const float4 *>(src);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float4 v = p[c];
        m[4 * c] = v.x; m[4 * c + 1] = v.y; m[4 * c + 2] = v.z; m[4 * c + 3] = v.w;
    }
}

// Per-lane inputs of one row, all loaded unconditionally so that a tile-walking wave
// can issue the NEXT row's loads before it computes the current one.
struct RowIn {
    uint32_t fl, sq;
    int32_t  p, mi;
    float4   ps, q;
};

__device__ __forceinline__ RowIn load_row(const EntK &e, const int lane, const uint32_t row_first,
                                          const uint32_t row_count)
{
    const uint32_t i = row_first + ((uint32_t)lane < row_count ? lane : 0);   // idle lanes re-read lane 0's entity
    RowIn r;
    r.fl = e.flags[i];
    r.p = e.parent[i];
    r.sq = e.seqs[i];
    r.mi = e.model[i];
    // a parent or model index outside the arrays would be a wild read: such an entity is treated as a
    // root / as model 0 instead (the reference holds pointers here, which cannot be out of range)
    if ((uint32_t)r.p >= e.n) r.p = -1;
    if ((uint32_t)r.mi >= e.n_models) r.mi = 0;
    r.ps = e.pos_scale[i];
    r.q = e.rot[i];
    return r;
}

// A host mirror's small frames in ONE launch (k_entities_tiles_host): the touched entities' inputs are read from the
// mirror's device-mapped upload image, what the row rebuilt is also written into its mapped result arrays.
struct HostIO {
    const float4   *pos_scale, *rot;                     // device-mapped host image of the inputs (slot order)
    const uint32_t *flags;
    const uint64_t *touched;                             // mapped: one bit per slot the host rewrote for this frame; NULL: none
    uint64_t *stale;                                     // device: rows the host holds an older copy of (kept here); NULL: not tracked
    uint32_t late_ok;                                    // CLAPGPU_HOSTIO_EXPORT_STALE_READ: stale rows with a reader now are exported too
    float          *o_mx, *o_inv, *o_aabb, *o_center;    // device-mapped host result arrays
    uint64_t       *o_vis, *o_rebuilt, *o_inside;
    uint32_t       *counter, *done, done_value;
    // Export policy (clapgpu_entities_hostio.keep_mask): NULL = every rebuilt row goes to the host; else only the rebuilt
    // rows somebody reads this frame -- drawn (vis_mask), containing a bounding-volume point, or flagged in keep[] --
    // and o_exported says which those were.
    const uint64_t *keep;
    uint64_t       *o_exported;
};

// load_row with the touched lanes' (flags, TRS) taken from the host image instead
__device__ __forceinline__ RowIn load_row_host(const EntK &e, const HostIO &h, const int lane, const uint32_t row_first,
                                               const uint32_t row_count, const uint64_t touched)
{
    RowIn r = load_row(e, lane, row_first, row_count);
    if ((uint32_t)lane < row_count && ((touched >> lane) & 1ull)) {
        const uint32_t i = row_first + lane;
        r.fl = h.flags[i];
        r.ps = h.pos_scale[i];
        r.q = h.rot[i];
    }
    return r;
}

// What process_row would load WHILE it works on a row (the stored box of a lane it does not rebuild, the model's box,
// the stored matrix and seq of a parent this launch does not rebuild), asked for rows ahead instead.  A small scene is a
// few dozen wavefronts walking half a dozen rows each: nothing hides a dependent load there, and each was 1-2 us of a
// 4-5 us row.  Safe to read early: a lane uses the stored box only if it is not rebuilt, a stored parent matrix / seq
// only if the parent is not rebuilt by this launch -- neither is written by it then.
struct RowPre {
    float2   bb0, bb1, bb2;          // aabb[i]                      (needs only the row's position)
    float4   lo, hi;                 // model_table[2 mi], [2 mi + 1] (needs RowIn)
    float4   pm[4];                  // mx[p]                         (needs RowIn)
    uint32_t pseq;                   // seqs[p]
};

__device__ __forceinline__ void load_row_box(const EntK &e, RowPre &r, const int lane, const uint32_t row_first, const uint32_t row_count)
{
    const uint32_t i = row_first + ((uint32_t)lane < row_count ? lane : 0);
    const float2 *b = reinterpret_cast<const float2 *>(e.aabb + 6 * (size_t)i);
    r.bb0 = b[0]; r.bb1 = b[1]; r.bb2 = b[2];
}

__device__ __forceinline__ void load_row_pre(const EntK &e, RowPre &r, const RowIn &in)
{
    r.lo = e.model_table[2 * in.mi];
    r.hi = e.model_table[2 * in.mi + 1];
    const size_t p = in.p >= 0 ? (size_t)in.p : 0;               // roots re-read entity 0: never used
    const float4 *m = reinterpret_cast<const float4 *>(e.mx + 16 * p);
    r.pm[0] = m[0]; r.pm[1] = m[1]; r.pm[2] = m[2]; r.pm[3] = m[3];
    r.pseq = e.seqs[p];
}

// joint attachment of entity i: binary search of the (short, sorted) table; index or -1
__device__ __forceinline__ int find_attach(const EntK &e, uint32_t i)
{
    uint32_t lo = 0, hi = e.n_attach;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (e.attach[mid].entity < i) lo = mid + 1; else hi = mid;
    }
    return (lo < e.n_attach && e.attach[lo].entity == i) ? (int)lo : -1;
}

__device__ __forceinline__ bool point_in_box(const float (&p)[3], const float (&bb)[6])
{
    return p[0] >= bb[0] && p[0] <= bb[3] && p[1] >= bb[1] && p[1] <= bb[4] && p[2] >= bb[2] && p[2] <= bb[5];
}

// One 64-entity row (= one vis_mask word) processed by one wave.
//   row_first  first entity of the row (multiple of 64), row_count valid lanes (1..64)
//   tile       8 KiB of wave-private LDS
//   TILE       the wave walks a tile of consecutive rows (= hierarchy levels of a group of
//              whole subtrees); a parent that sits in the previous row is taken from the
//              registers of the lane that just computed it (16 cross-lane reads) instead of
//              being re-read from HBM.  carry_* hold the previous row's results.
//   XV         the launch culls further views (clapgpu_entities.views) with the main one
template <bool CULL, bool TILE, bool HOST = false, bool XV = false>
__device__ __forceinline__ void process_row(const EntK &e, const RowIn &in, float4 *tile, const int lane,
                                            const uint32_t row_first, const uint32_t row_count,
                                            const uint32_t mode, const lmd::FrustumK &fr,
                                            const bool have_prev, const uint32_t prev_first,
                                            float (&carry_mx)[16], uint32_t &carry_seq, bool &carry_valid,
                                            const HostIO *hio = nullptr, const RowPre *pre = nullptr,
                                            const XViewsK *xv = nullptr)
{
    const bool in_range = (uint32_t)lane < row_count;
    const uint32_t i = row_first + (in_range ? lane : 0);
    // HOST: which of this row's rebuilt lanes the mirror wants back (asked for now, used after the cull)
    uint64_t host_want = ~0ull;
    bool host_filter = false;
    uint64_t stale_w = 0;
    if constexpr (HOST) {
        host_filter = hio->keep != nullptr;
        host_want = host_filter ? hio->keep[row_first >> 6] : ~0ull;
        if (hio->stale) stale_w = hio->stale[row_first >> 6];   // (asked for now, used after the cull)
    }

    const uint32_t fl = in.fl;
    const bool alive = in_range && (fl & CLAPGPU_E_ALIVE);
    const bool dirty = (mode & CLAPGPU_UPDATE_ALL_DIRTY) ? true : (fl & CLAPGPU_E_DIRTY) != 0;
    const int32_t p = in.p;
    uint32_t seq = in.sq & 0xffffu, pseq = in.sq >> 16;

    // ---- where does the parent live? ----
    bool in_prev = false;
    float pm[16];
    uint32_t parent_seq_now = 0;
    bool parent_in_regs = false;
    if (TILE) {
        in_prev = have_prev && p >= 0 && (uint32_t)p >= prev_first && (uint32_t)p < prev_first + WAVE;
        const int src = in_prev ? (int)((uint32_t)p - prev_first) : lane;
#pragma unroll
        for (int k = 0; k < 16; k++) pm[k] = __shfl(carry_mx[k], src);
        parent_seq_now = __shfl(carry_seq, src);
        if constexpr (HOST) {
            // The cross-lane read must not sit behind the `in_prev &&`: the compiler then runs it with only the in_prev
            // lanes active, and a parent whose lane number belongs, in THIS row, to a padding lane reads as "not rebuilt"
            // (ds_bpermute returns 0 from an inactive lane).  The plain kernels fall back to the stored matrix then --
            // already written by this wavefront, so merely a slower lane; with the parent matrix asked for rows ahead
            // (RowPre) it would be last frame's.  Their code is left as it is: k_entities_tiles' ISA is not to move.
            const int valid = __shfl((int)carry_valid, src);
            parent_in_regs = in_prev && valid != 0;
        } else
            parent_in_regs = in_prev && (__shfl((int)carry_valid, src) != 0);
    }
    if (p >= 0 && !in_prev)
        parent_seq_now = (HOST ? pre->pseq : e.seqs[p]) & 0xffffu;   // parent updated by an earlier launch

    // parent_transform_apply's skip test (model.c:1609-1611) / default_update's dirty test (1667)
    bool rebuild = alive;
    int at = -1;
    if (alive && p >= 0 && (fl & CLAPGPU_E_JOINT_ATTACHED) && e.n_attach)
        at = find_attach(e, i);
    const bool attached = at >= 0;
    if (p >= 0) {
        if (!attached && pseq == parent_seq_now && !dirty)     // joint attachments are rebuilt every frame
            rebuild = false;
        else
            pseq = parent_seq_now;
    } else if (!dirty) {
        rebuild = false;
    }

    float mx[16], inv[16], bb[6], ctr[3];
    bool has_aabb = false;

    if (rebuild) {
        const float4 lo = HOST ? pre->lo : e.model_table[2 * in.mi];       // min.xyz, skip_aabb bits
        const float4 hi = HOST ? pre->hi : e.model_table[2 * in.mi + 1];   // max.xyz, 0
        float local_mx[16];
        if (attached)                                            // (jt * bind) * local from k_attach_prepare
            load_mat4(local_mx, e.attach_local + 16 * (size_t)at);
        else
            lmd::trs(local_mx, in.ps.x, in.ps.y, in.ps.z, in.ps.w, in.q.x, in.q.y, in.q.z, in.q.w);
        if (p >= 0) {
            if (!parent_in_regs) {                               // stored matrix of a parent not rebuilt here
                if constexpr (HOST) {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        pm[4 * c] = pre->pm[c].x; pm[4 * c + 1] = pre->pm[c].y; pm[4 * c + 2] = pre->pm[c].z; pm[4 * c + 3] = pre->pm[c].w;
                    }
                } else
                    load_mat4(pm, e.mx + 16 * (size_t)p);
            }
            lmd::mul(mx, pm, local_mx);                          // model.c:1625 / 1640
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) mx[k] = local_mx[k];
        }
#ifdef CLAPGPU_EXP_NO_INVERT                                     // sensitivity experiments only (tools/entities_sensitivity.sh): wrong results
#pragma unroll
        for (int k = 0; k < 16; k++) inv[k] = mx[k];
#else
        lmd::invert(inv, mx);
#endif

        has_aabb = __float_as_uint(lo.w) == 0u;                  // model.c:1204
#ifdef CLAPGPU_EXP_NO_AABB
        if (has_aabb) {
#pragma unroll
            for (int k = 0; k < 6; k++) bb[k] = mx[12 + k % 3] + (k < 3 ? lo.x : hi.x);
#pragma unroll
            for (int k = 0; k < 3; k++) ctr[k] = mx[12 + k];
        }
#else
        if (has_aabb)
            lmd::world_aabb(bb, ctr, mx, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z);
#endif

        seq = (seq + 1) & 0xffffu;                               // uint16 wrap (model.h:404)
        e.seqs[i] = seq | (pseq << 16);
        if (!(mode & CLAPGPU_UPDATE_ALL_DIRTY) && (fl & CLAPGPU_E_DIRTY))
            e.flags[i] = fl & ~CLAPGPU_E_DIRTY;                  // transform_clear_updated
    }

    if (TILE) {                                                  // hand this row to the next one
#pragma unroll
        for (int k = 0; k < 16; k++) carry_mx[k] = mx[k];
        carry_seq = seq;
        carry_valid = rebuild;
    }

    // ---- stores: whole-wave fast path when every valid lane rebuilt (the common case) ----
    const uint64_t rebuilt_mask = __ballot(rebuild);
    if (e.rebuilt_mask && lane == 0) e.rebuilt_mask[row_first >> 6] = rebuilt_mask;
    const uint64_t aabb_mask = __ballot(rebuild && has_aabb);
    const uint64_t full = row_count == WAVE ? ~0ull : ((1ull << row_count) - 1ull);
    const size_t e0 = row_first;
    float4 *tile_a = tile, *tile_b = tile + 256;                 // 2 x 4 KiB
    float *tile_f = reinterpret_cast<float *>(tile);

    if (rebuilt_mask == full) {
        float4 va[4], vb[4];
        stage_mat4(tile_a, mx, lane);
        stage_mat4(tile_b, inv, lane);
        wave_lds_fence();
        unstage_mat4(tile_a, va, lane);
        unstage_mat4(tile_b, vb, lane);
        store_mat4_rows(e.mx + 16 * e0, va, lane, (int)row_count);
        store_mat4_rows(e.inv_mx + 16 * e0, vb, lane, (int)row_count);
        wave_lds_fence();
    } else if (rebuild) {
        float4 *dm = reinterpret_cast<float4 *>(e.mx + 16 * (size_t)i);
        float4 *di = reinterpret_cast<float4 *>(e.inv_mx + 16 * (size_t)i);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            dm[c] = make_float4(mx[4 * c], mx[4 * c + 1], mx[4 * c + 2], mx[4 * c + 3]);
            di[c] = make_float4(inv[4 * c], inv[4 * c + 1], inv[4 * c + 2], inv[4 * c + 3]);
        }
    }
    if (aabb_mask == full) {
        stage_rows<6>(tile_f, bb, lane);                         // 1536 B
        stage_rows<3>(tile_f + 6 * WAVE, ctr, lane);             //  768 B
        wave_lds_fence();
        store_rows<6>(tile_f, e.aabb + 6 * e0, lane, (int)row_count);
        store_rows<3>(tile_f + 6 * WAVE, e.center + 3 * e0, lane, (int)row_count);
        wave_lds_fence();
    } else if (rebuild && has_aabb) {
#pragma unroll
        for (int k = 0; k < 6; k++) e.aabb[6 * (size_t)i + k] = bb[k];
#pragma unroll
        for (int k = 0; k < 3; k++) e.center[3 * (size_t)i + k] = ctr[k];
    }

    const bool want_bv = e.bv_on != 0;
    if (CULL || want_bv) {
        // Entities that were not rebuilt (or whose model skips AABBs) use their stored box.
        if (in_range && !(rebuild && has_aabb)) {
            if constexpr (HOST) {
                bb[0] = pre->bb0.x; bb[1] = pre->bb0.y; bb[2] = pre->bb1.x; bb[3] = pre->bb1.y; bb[4] = pre->bb2.x; bb[5] = pre->bb2.y;
            } else {
#pragma unroll
                for (int k = 0; k < 6; k++) bb[k] = e.aabb[6 * (size_t)i + k];
            }
        }
    }
    if (want_bv) {                                               // model.c:1703-1713
        bool inside = in_range && (fl & CLAPGPU_E_ALIVE) && point_in_box(e.bv_cam, bb);
        if (!inside && e.bv_has_ctl)
            inside = in_range && (fl & CLAPGPU_E_ALIVE) && point_in_box(e.bv_ctl, bb);
        if (inside && e.bv_has_ctl && i == e.bv_ctl_entity)
            inside = false;
        const uint64_t inside_mask = __ballot(inside);
        if (e.bv_inside && lane == 0) e.bv_inside[e0 >> 6] = inside_mask;
        if constexpr (HOST) {
            if (hio->o_inside && lane == 0) hio->o_inside[e0 >> 6] = inside_mask;
            host_want |= inside_mask;                            // default_update's pick reads the box on the host
        }
        if (inside_mask) {                                       // rare: almost no box contains the camera
            unsigned long long key = 0;
            if (inside) {
                const float4 lo = e.model_table[2 * in.mi], hi = e.model_table[2 * in.mi + 1];
                const float X = fabsf(hi.x - lo.x) * in.ps.w, Y = fabsf(hi.y - lo.y) * in.ps.w,
                            Z = fabsf(hi.z - lo.z) * in.ps.w;  // entity3d_aabb_X/Y/Z (model.c:1185-1198)
                const float vol = X * Y * Z;
                key = ((unsigned long long)__float_as_uint(vol) << 32) | (0xFFFFFFFFu - i);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(key, off);
                key = o > key ? o : key;
            }
            if (lane == 0 && key && e.bv_result)
                atomicMax(e.bv_result, key);
        }
    }
    if (CULL) {
        bool vis = in_range && (fl & CLAPGPU_E_ALIVE) && (fl & CLAPGPU_E_VISIBLE);   // model.c:959-965
        if (vis && !(fl & CLAPGPU_E_SKIP_CULLING))
            vis = lmd::aabb_in_frustum_fast(fr, bb);                                  // model.c:967-971
        const uint64_t m = __ballot(vis);
        if (lane == 0) {
            e.vis_mask[e0 >> 6] = m;
            e.vis_row_pop[e0 >> 6] = (uint8_t)__popcll(m);       // feeds the single-launch compaction
            if constexpr (HOST) hio->o_vis[e0 >> 6] = m;
        }
        if constexpr (HOST) host_want |= m;                      // what the render passes draw
        if constexpr (XV) {
            const uint64_t mx_any = cull_extra_views(*xv, in_range && (fl & CLAPGPU_E_ALIVE) && (fl & CLAPGPU_E_VISIBLE), fl, bb,
                                                     (uint32_t)(e0 >> 6), lane);
            if constexpr (HOST) host_want |= mx_any;             // ... in any of the frame's passes
        }
    } else {
        if constexpr (HOST) host_want = ~0ull;                   // no frustum: a pass without a camera draws everything (model.c:969)
    }

    if constexpr (HOST) {                                        // the mirror's copy of what this row rebuilt, and which lanes those are
        const uint64_t exported = host_filter ? (rebuilt_mask & host_want) : rebuilt_mask;
        if (rebuild && ((exported >> lane) & 1ull)) {
            float4 *hm = reinterpret_cast<float4 *>(hio->o_mx + 16 * (size_t)i);
            float4 *hi = reinterpret_cast<float4 *>(hio->o_inv + 16 * (size_t)i);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                hm[c] = make_float4(mx[4 * c], mx[4 * c + 1], mx[4 * c + 2], mx[4 * c + 3]);
                hi[c] = make_float4(inv[4 * c], inv[4 * c + 1], inv[4 * c + 2], inv[4 * c + 3]);
            }
            if (has_aabb) {
                float2 *hb = reinterpret_cast<float2 *>(hio->o_aabb + 6 * (size_t)i);
                hb[0] = make_float2(bb[0], bb[1]); hb[1] = make_float2(bb[2], bb[3]); hb[2] = make_float2(bb[4], bb[5]);
                float *hc = hio->o_center + 3 * (size_t)i;
                hc[0] = ctr[0]; hc[1] = ctr[1]; hc[2] = ctr[2];
            }
        }
        // Rows the host was left an older copy of by earlier launches and that have a reader NOW (came into view, contain
        // the camera, were flagged since): over from the device arrays in this same pass -- rare lanes, plain loads
        const uint64_t late = hio->late_ok ? stale_w & host_want & ~rebuilt_mask & __ballot(alive) : 0ull;
        if ((late >> lane) & 1ull) {
            const float4 *sm = reinterpret_cast<const float4 *>(e.mx + 16 * (size_t)i);
            const float4 *si = reinterpret_cast<const float4 *>(e.inv_mx + 16 * (size_t)i);
            float4 *hm = reinterpret_cast<float4 *>(hio->o_mx + 16 * (size_t)i);
            float4 *hi = reinterpret_cast<float4 *>(hio->o_inv + 16 * (size_t)i);
#pragma unroll
            for (int c = 0; c < 4; c++) { hm[c] = sm[c]; hi[c] = si[c]; }
            const float2 *sb = reinterpret_cast<const float2 *>(e.aabb + 6 * (size_t)i);
            float2 *hb = reinterpret_cast<float2 *>(hio->o_aabb + 6 * (size_t)i);
            hb[0] = sb[0]; hb[1] = sb[1]; hb[2] = sb[2];
            const float *sc = e.center + 3 * (size_t)i;
            float *hc = hio->o_center + 3 * (size_t)i;
            hc[0] = sc[0]; hc[1] = sc[1]; hc[2] = sc[2];
        }
        if (lane == 0) {
            hio->o_rebuilt[row_first >> 6] = rebuilt_mask;
            if (hio->o_exported) hio->o_exported[row_first >> 6] = exported | late;
            if (hio->stale) {
                const uint64_t ns = (stale_w | rebuilt_mask) & ~(exported | late);
                if (ns != stale_w) hio->stale[row_first >> 6] = ns;
            }
        }
    }
}


// entities_host.hip
int launch_entities_tiles_host(hipStream_t stream, bool cull, const lmd::FrustumK &fr, const EntK &e, const HostIO &h,
                               const uint32_t *tile_row_start, uint32_t n_tiles, uint32_t n, uint32_t mode, const XViewsK &xv);

} // namespace clapgpu
