// entities.hip -- entity transform hierarchy -> inverse -> world AABB -> frustum cull
// for gfx950 (MI355X).  One lane per entity, one launch per hierarchy level.
//
// Replaces, per entity, the reference's default_update() transform branch
// (model.c:1649-1695), parent_transform_apply() jointless attachment
// (model.c:1594-1647), mat4x4_invert (linmath.h:611-651), entity3d_aabb_update
// (model.c:1200-1234) and the draw predicate of _models_render with
// view_entity_in_frustum (model.c:959-973, view.c:296-337).
//
// HBM-bound: ~276 algorithmic bytes / entity.  Inputs are read as
// coalesced float4 / dword streams, the parent matrix as 4 x 16 B per lane, and
// every output row block (64 entities x 64 / 64 / 24 / 12 B) is transposed through
// a wave-private LDS tile so each store instruction writes 1 KiB contiguous.
#include <string.h>
#include <math.h>
#include <type_traits>
#include "common.h"
#include "lm_dev.h"
#include "entities_row.h"

namespace clapgpu {

// The common row of a tile walk, as straight-line code: all 64 lanes alive and rebuilt, every parent in the
// previous row's registers (or none), no joint attachment, every model with a box (taken from the LDS copy `mt` of
// the model table), no camera query.  Same arithmetic as process_row; what differs is what the wavefront waits for.
// gfx950 has one counter for vector loads AND stores, retired in issue order: a wait for the newest load is a wait
// for every store before it.  process_row loads the model table (and, on its other paths, parents) between the
// previous row's stores and its own arithmetic, and its stores sit in branches, so the compiler can only wait for
// "everything" there and again before the row hand-over -- each row's 10 KB of stores was drained twice while the
// SIMD idled.  Here no vector load is issued between the prefetch of the next row and the hand-over, and the big
// stores are unconditional, so the hand-over waits for "all but the last N operations" and the stores stay in
// flight under the next row's arithmetic.
template <bool CULL, bool XV = false>
__device__ __forceinline__ void process_row_fast(const EntK &e, const RowIn &in, float4 *tile, const float4 *mt,
                                                 const int lane, const uint32_t row_first, const uint32_t mode,
                                                 const lmd::FrustumK &fr, const int src, const uint32_t parent_seq_now,
                                                 float (&carry_mx)[16], uint32_t &carry_seq, const XViewsK *xv = nullptr)
{
    const uint32_t i = row_first + lane;
    const uint32_t fl = in.fl;
    const int32_t p = in.p;
    uint32_t seq = in.sq & 0xffffu, pseq = in.sq >> 16;
    float pm[16];
#pragma unroll
    for (int k = 0; k < 16; k++) pm[k] = __shfl(carry_mx[k], src);
    if (p >= 0) pseq = parent_seq_now;

    const float4 lo = mt[2 * in.mi], hi = mt[2 * in.mi + 1];
    float mx[16], inv[16], bb[6], ctr[3], local_mx[16];
    lmd::trs(local_mx, in.ps.x, in.ps.y, in.ps.z, in.ps.w, in.q.x, in.q.y, in.q.z, in.q.w);
    if (p >= 0) {
        lmd::mul(mx, pm, local_mx);                              // model.c:1625
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
#ifdef CLAPGPU_EXP_NO_AABB
#pragma unroll
    for (int k = 0; k < 6; k++) bb[k] = mx[12 + k % 3] + (k < 3 ? lo.x : hi.x);
#pragma unroll
    for (int k = 0; k < 3; k++) ctr[k] = mx[12 + k];
#else
    lmd::world_aabb(bb, ctr, mx, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z);
#endif
    seq = (seq + 1) & 0xffffu;
#pragma unroll
    for (int k = 0; k < 16; k++) carry_mx[k] = mx[k];
    carry_seq = seq;

    const size_t e0 = row_first;
    float4 *tile_a = tile, *tile_b = tile + 256;
    float *tile_f = reinterpret_cast<float *>(tile);
    float4 va[4], vb[4];
    stage_mat4(tile_a, mx, lane);
    stage_mat4(tile_b, inv, lane);
    wave_lds_fence();
    unstage_mat4(tile_a, va, lane);
    unstage_mat4(tile_b, vb, lane);
    store_mat4_rows(e.mx + 16 * e0, va, lane, WAVE);
    store_mat4_rows(e.inv_mx + 16 * e0, vb, lane, WAVE);
    e.seqs[i] = seq | (pseq << 16);
    wave_lds_fence();
    // every store below is issued by all 64 lanes, none under a lane test: a store in a branch is one the compiler
    // cannot count.  The row's 1536 B of boxes = one 16-byte and one 8-byte piece per lane; the centres are the
    // lanes' own 12 bytes in lane order already.
    stage_rows<6>(tile_f, bb, lane);
    wave_lds_fence();
    {
        float *ab = e.aabb + 6 * e0;
        store_stream(&reinterpret_cast<float4 *>(ab)[lane], reinterpret_cast<const float4 *>(tile_f)[lane]);
        const float2 t2 = reinterpret_cast<const float2 *>(tile_f + 4 * WAVE)[lane];
        const clapgpu_f2 v2 = { t2.x, t2.y };
        const clapgpu_f3 v3 = { ctr[0], ctr[1], ctr[2] };
#ifdef CLAPGPU_PLAIN_STORES          // A/B builds only (tools/profile_entities_scale.sh)
        reinterpret_cast<clapgpu_f2 *>(ab + 4 * WAVE)[lane] = v2;
        *reinterpret_cast<clapgpu_f3 *>(e.center + 3 * (e0 + lane)) = v3;
#else
        __builtin_nontemporal_store(v2, reinterpret_cast<clapgpu_f2 *>(ab + 4 * WAVE) + lane);
        __builtin_nontemporal_store(v3, reinterpret_cast<clapgpu_f3 *>(e.center + 3 * (e0 + lane)));   // sizeof(f3) is 16: index in floats
#endif
    }
    wave_lds_fence();
    if (!(mode & CLAPGPU_UPDATE_ALL_DIRTY) && (fl & CLAPGPU_E_DIRTY))
        e.flags[i] = fl & ~CLAPGPU_E_DIRTY;                      // transform_clear_updated
    if (e.rebuilt_mask && lane == 0) e.rebuilt_mask[row_first >> 6] = ~0ull;
    if (CULL) {
        bool vis = (fl & CLAPGPU_E_VISIBLE) != 0;                                      // model.c:959-965
        if (vis && !(fl & CLAPGPU_E_SKIP_CULLING))
            vis = lmd::aabb_in_frustum_fast(fr, bb);                                  // model.c:967-971
        const uint64_t m = __ballot(vis);
        e.vis_mask[e0 >> 6] = m;                                 // the same word from all 64 lanes: one request
        e.vis_row_pop[e0 >> 6] = (uint8_t)__popcll(m);
        if constexpr (XV) cull_extra_views(*xv, (fl & CLAPGPU_E_VISIBLE) != 0, fl, bb, (uint32_t)(e0 >> 6), lane);
    }
}

// model.c:1618-1622 + 1633-1639: local = TRS of the attached entity, joint_mx = joint_transforms[j] * bind[j],
// attach_local = joint_mx * local.  One lane per attachment (there are few).
__global__ __launch_bounds__(ENT_BLOCK)
void k_attach_prepare(EntK e)
{
    const uint32_t k = blockIdx.x * ENT_BLOCK + threadIdx.x;
    if (k >= e.n_attach) return;
    const clapgpu_attach at = e.attach[k];
    const float4 ps = e.pos_scale[at.entity], q = e.rot[at.entity];
    float local_mx[16], jt[16], bd[16], joint_mx[16], out[16];
    lmd::trs(local_mx, ps.x, ps.y, ps.z, ps.w, q.x, q.y, q.z, q.w);
    load_mat4(jt, e.jt_pool + 16 * (size_t)at.jt);
    load_mat4(bd, e.bind_pool + 16 * (size_t)at.bind);
    lmd::mul(joint_mx, jt, bd);
    lmd::mul(out, joint_mx, local_mx);
    float4 *d = reinterpret_cast<float4 *>(e.attach_local + 16 * (size_t)k);
#pragma unroll
    for (int c = 0; c < 4; c++)
        d[c] = make_float4(out[4 * c], out[4 * c + 1], out[4 * c + 2], out[4 * c + 3]);
}


// One launch per hierarchy level: every parent was written by an earlier launch.
// `first` is a multiple of 64, so each wave owns exactly one vis_mask word.
template <bool CULL>
__global__ __launch_bounds__(ENT_BLOCK)
void k_entities_level(EntK e, uint32_t first, uint32_t count, uint32_t mode, lmd::FrustumK fr)
{
    __shared__ float4 lds_tiles[ENT_BLOCK / WAVE][LDS_F4_PER_WAVE];
    const int lane = lane_id();
    const int wave = threadIdx.x / WAVE;
    const uint32_t wave_local0 = blockIdx.x * ENT_BLOCK + wave * WAVE;
    if (wave_local0 >= count)
        return;                                                  // whole wave; no block-level sync below
    const uint32_t row_count = count - wave_local0 < WAVE ? count - wave_local0 : WAVE;
    float carry_mx[16];
    uint32_t carry_seq = 0;
    bool carry_valid = false;
    const RowIn in = load_row(e, lane, first + wave_local0, row_count);
    process_row<CULL, false>(e, in, lds_tiles[wave], lane, first + wave_local0, row_count, mode, fr,
                             false, 0, carry_mx, carry_seq, carry_valid);
}

// One launch for the whole forest: wave t walks tile t = rows [tile_row_start[t], tile_row_start[t+1]),
// row r = entities [64r, 64r+64) = one hierarchy level of the subtrees packed into the tile.
// The next row's inputs are in flight while the current row is computed.
constexpr int ENT_MT_CAP = 256;       // models whose table the tile kernel keeps in LDS (8 KiB); more: general loop only
constexpr int ENT_TILE_WAVES = 1;       // occupancy hint; 4 measured the same 44 us
// The frustum (63 dwords) is the FIRST kernel argument and is read where it is used, through the kernarg segment
// pointer, with the pointer laundered once per row: held in SGPRs across the row loop next to ~30 array pointers it
// cost 194 spilled SGPRs -- some 300 v_readlane / s_nop per row of a kernel that issues 850 vector instructions per row.
template <bool CULL>
__global__ __launch_bounds__(ENT_BLOCK, ENT_TILE_WAVES)
void k_entities_tiles(lmd::FrustumK fr_arg, EntK e, const uint32_t *tile_row_start, uint32_t n_tiles, uint32_t n,
                      uint32_t mode)
{
#define TILES_XV 0
#define TILES_XV_PTR nullptr
#include "entities_tiles_body.inc"
#undef TILES_XV
#undef TILES_XV_PTR
}

// ... with the frame's further views (clapgpu_entities.views): one more mask word per view and row
__global__ __launch_bounds__(ENT_BLOCK, ENT_TILE_WAVES)
void k_entities_tiles_xv(lmd::FrustumK fr_arg, EntK e, const uint32_t *tile_row_start, uint32_t n_tiles, uint32_t n,
                         uint32_t mode, XViewsK xv)
{
    constexpr bool CULL = true;
#define TILES_XV 1
#define TILES_XV_PTR (&xv)
#include "entities_tiles_body.inc"
#undef TILES_XV
#undef TILES_XV_PTR
}

__global__ __launch_bounds__(ENT_BLOCK)
void k_entities_level_xv(EntK e, uint32_t first, uint32_t count, uint32_t mode, lmd::FrustumK fr, XViewsK xv)
{
    __shared__ float4 lds_tiles[ENT_BLOCK / WAVE][LDS_F4_PER_WAVE];
    const int lane = lane_id();
    const int wave = threadIdx.x / WAVE;
    const uint32_t wave_local0 = blockIdx.x * ENT_BLOCK + wave * WAVE;
    if (wave_local0 >= count)
        return;
    const uint32_t row_count = count - wave_local0 < WAVE ? count - wave_local0 : WAVE;
    float carry_mx[16];
    uint32_t carry_seq = 0;
    bool carry_valid = false;
    const RowIn in = load_row(e, lane, first + wave_local0, row_count);
    process_row<true, false, false, true>(e, in, lds_tiles[wave], lane, first + wave_local0, row_count, mode, fr,
                                          false, 0, carry_mx, carry_seq, carry_valid, nullptr, nullptr, &xv);
}

// Cull-only pass over stored AABBs (one per render pass in the reference).
// ... every view of the frame from one read of the boxes
__global__ __launch_bounds__(ENT_BLOCK)
void k_entities_cull_xv(const uint32_t *flags, const float *aabb, uint64_t *vis_mask, uint8_t *vis_row_pop,
                        uint32_t n, lmd::FrustumK fr, XViewsK xv)
{
    const uint32_t i = blockIdx.x * ENT_BLOCK + threadIdx.x;
    const int lane = lane_id();
    bool base = false, vis = false;
    uint32_t fl = 0;
    float bb[6] = { 0, 0, 0, 0, 0, 0 };
    if (i < n) {
        fl = flags[i];
        base = (fl & CLAPGPU_E_ALIVE) && (fl & CLAPGPU_E_VISIBLE);
        vis = base;
        if (base && !(fl & CLAPGPU_E_SKIP_CULLING)) {
#pragma unroll
            for (int k = 0; k < 6; k++) bb[k] = aabb[6 * (size_t)i + k];
            vis = lmd::aabb_in_frustum_fast(fr, bb);
        }
    }
    if ((i - lane) >= n) return;                                 // the whole wavefront is past the end
    const uint64_t m = __ballot(vis);
    if (lane == 0) {
        vis_mask[i >> 6] = m;
        vis_row_pop[i >> 6] = (uint8_t)__popcll(m);
    }
    cull_extra_views(xv, base, fl, bb, i >> 6, lane);
}

__global__ __launch_bounds__(ENT_BLOCK)
void k_entities_cull(const uint32_t *flags, const float *aabb, uint64_t *vis_mask, uint8_t *vis_row_pop,
                     uint32_t n, lmd::FrustumK fr)
{
    const uint32_t i = blockIdx.x * ENT_BLOCK + threadIdx.x;
    bool vis = false;
    if (i < n) {
        const uint32_t fl = flags[i];
        vis = (fl & CLAPGPU_E_ALIVE) && (fl & CLAPGPU_E_VISIBLE);
        if (vis && !(fl & CLAPGPU_E_SKIP_CULLING)) {
            float bb[6];
#pragma unroll
            for (int k = 0; k < 6; k++) bb[k] = aabb[6 * (size_t)i + k];
            vis = lmd::aabb_in_frustum_fast(fr, bb);
        }
    }
    const uint64_t m = __ballot(vis);
    if (lane_id() == 0 && (i - lane_id()) < n) {
        vis_mask[i >> 6] = m;
        vis_row_pop[i >> 6] = (uint8_t)__popcll(m);
    }
}

// ---- ordered compaction of the visibility bitmask ----
// A group = 64 mask words = 4096 entities = one wave.
constexpr int GROUP_WORDS = 64;

__device__ __forceinline__ uint64_t load_mask_word(const uint64_t *vis_mask, uint32_t w, uint32_t n)
{
    const uint32_t nwords = (n + 63) / 64;
    if (w >= nwords)
        return 0;
    uint64_t v = vis_mask[w];
    const uint32_t rem = n - w * 64;                 // entities covered by this word
    if (rem < 64)
        v &= (1ull << rem) - 1ull;                   // bits past n are padding
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(WAVE)
void k_mask_group_count(const uint64_t *vis_mask, uint32_t n, uint32_t *group_count)
{
    const uint32_t w = blockIdx.x * GROUP_WORDS + threadIdx.x;
    const uint32_t c = wave_sum(__popcll(load_mask_word(vis_mask, w, n)));
    if (threadIdx.x == 0)
        group_count[blockIdx.x] = c;
}

__global__ __launch_bounds__(WAVE)
void k_visible_expand(const uint64_t *vis_mask, uint32_t n, const uint32_t *group_count,
                      uint32_t n_groups, uint32_t index_base, uint32_t *visible, uint32_t *count)
{
    const int lane = threadIdx.x;
    const uint32_t g = blockIdx.x;

    uint32_t pre = 0;                                 // visible entities in groups before g
    for (uint32_t j = lane; j < g; j += WAVE)
        pre += group_count[j];
    pre = wave_sum(pre);

    const uint64_t word = load_mask_word(vis_mask, g * GROUP_WORDS + lane, n);
    const uint32_t cnt = __popcll(word);
    uint32_t incl = cnt;                              // inclusive scan of per-word counts over lanes
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        uint32_t t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    const uint32_t excl = incl - cnt;

    // one mask word per iteration: lane l owns bit l, ranks come from the bits below it,
    // so the 4-byte stores of an iteration are contiguous and ascending.
    for (int k = 0; k < GROUP_WORDS; k++) {
        const uint64_t wk = __shfl(word, k);
        if (wk == 0) continue;                        // wave-uniform
        const uint32_t base = pre + __shfl(excl, k);
        if ((wk >> lane) & 1ull) {
            const uint32_t rank = __popcll(wk & ((1ull << lane) - 1ull));
            visible[base + rank] = index_base + (g * GROUP_WORDS + k) * 64u + lane;
        }
    }
    if (g == n_groups - 1 && lane == WAVE - 1)
        *count = pre + incl;
}

// ---- the same over a GATHERED mask: one segment of cap_words words per rank, rank r's slot i has global id base[r] + i ----
// (clapgpu_visible_compact_ranges: shards cut from one scene by clapgpu_shard_tile_range are uneven; every rank sends a
// mask of the common capacity, only its first n_words[r] words count)
constexpr int MAX_SEGMENTS = 64;
struct SegK { uint32_t cap_words, n_seg; uint32_t base[MAX_SEGMENTS], n_words[MAX_SEGMENTS]; };

__device__ __forceinline__ uint64_t load_seg_word(const uint64_t *mask, const SegK &sg, uint32_t w, uint32_t *id_base)
{
    const uint32_t r = w / sg.cap_words, lw = w - r * sg.cap_words;
    if (r >= sg.n_seg || lw >= sg.n_words[r]) { *id_base = 0; return 0ull; }
    *id_base = sg.base[r] + lw * 64u;
    return mask[w];
}

__global__ __launch_bounds__(WAVE)
void k_mask_group_count_seg(const uint64_t *mask, SegK sg, uint32_t *group_count)
{
    uint32_t idb;
    const uint32_t c = wave_sum(__popcll(load_seg_word(mask, sg, blockIdx.x * GROUP_WORDS + threadIdx.x, &idb)));
    if (threadIdx.x == 0)
        group_count[blockIdx.x] = c;
}

__global__ __launch_bounds__(WAVE)
void k_visible_expand_seg(const uint64_t *mask, SegK sg, const uint32_t *group_count, uint32_t n_groups, uint32_t *visible,
                          uint32_t *count)
{
    const int lane = threadIdx.x;
    const uint32_t g = blockIdx.x;
    uint32_t pre = 0;
    for (uint32_t j = lane; j < g; j += WAVE)
        pre += group_count[j];
    pre = wave_sum(pre);
    uint32_t idb;
    const uint64_t word = load_seg_word(mask, sg, g * GROUP_WORDS + lane, &idb);
    const uint32_t cnt = __popcll(word);
    uint32_t incl = cnt;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        uint32_t t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    const uint32_t excl = incl - cnt;
    for (int k = 0; k < GROUP_WORDS; k++) {
        const uint64_t wk = __shfl(word, k);
        if (wk == 0) continue;                        // wave-uniform
        const uint32_t base = pre + __shfl(excl, k), ids = __shfl(idb, k);
        if ((wk >> lane) & 1ull) {
            const uint32_t rank = __popcll(wk & ((1ull << lane) - 1ull));
            visible[base + rank] = ids + lane;
        }
    }
    if (g == n_groups - 1 && lane == WAVE - 1)
        *count = pre + incl;
}

// Single-launch compaction for up to RP_MAX_ROWS rows: the update / cull kernels leave one
// popcount byte per 64-entity row, so a wave gets the number of visible entities before its
// first row from at most RP_MAX_ROWS / 1024 16-byte loads per lane -- no separate count pass.
constexpr int RP_ROWS = 16;                 // rows (mask words) per wave; 16 keeps the byte prefix 16-B aligned
constexpr uint32_t RP_MAX_ROWS = 1u << 16;  // 4M entities; beyond that the two-pass path scales better

__device__ __forceinline__ uint32_t sum_bytes(uint32_t v, uint32_t acc)
{
    return __builtin_amdgcn_sad_u8(v, 0u, acc);      // v_sad_u8: acc + sum of the 4 bytes
}

__global__ __launch_bounds__(ENT_BLOCK)
void k_visible_expand_rp(const uint64_t *vis_mask, const uint8_t *row_pop, uint32_t n,
                         uint32_t index_base, uint32_t *visible, uint32_t *count)
{
    const int lane = lane_id();
    const uint32_t g = blockIdx.x * (ENT_BLOCK / WAVE) + threadIdx.x / WAVE;
    const uint32_t n_rows = (n + 63) / 64;
    const uint32_t row0 = g * RP_ROWS;
    if (row0 >= n_rows)
        return;

    // visible entities in rows [0, row0); row0 % 16 == 0
    const uint32_t pre = wave_byte_sum(row_pop, row0, lane);

    const uint64_t word = lane < RP_ROWS ? load_mask_word(vis_mask, row0 + lane, n) : 0ull;
    const uint32_t cnt = __popcll(word);
    uint32_t incl = cnt;
#pragma unroll
    for (int off = 1; off < RP_ROWS; off <<= 1) {
        uint32_t t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    const uint32_t excl = incl - cnt;
    const uint32_t lo = (uint32_t)word, hi = (uint32_t)(word >> 32);

#pragma unroll
    for (int k = 0; k < RP_ROWS; k++) {
        // readlane returns int: go through uint32_t or bit 31 sign-extends into the high half
        const uint64_t wk = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(lo, k) |
                            ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(hi, k) << 32);
        if (wk == 0) continue;                        // scalar branch
        const uint32_t base = pre + (uint32_t)__builtin_amdgcn_readlane(excl, k);
        if ((wk >> lane) & 1ull) {
            const uint32_t rank = __popcll(wk & ((1ull << lane) - 1ull));
            visible[base + rank] = index_base + (row0 + k) * 64u + lane;
        }
    }
    if (row0 + RP_ROWS >= n_rows && lane == RP_ROWS - 1)
        *count = pre + incl;
}

// ---- per-pass LOD pick for the entities on the visible list (model.c:975-992) ----
// glibc 2.35 sysdeps/ieee754/flt-32/s_cbrtf.c restated (the reference calls libm cbrtf(),
// model.c:1263): the float is rescaled by frexpf, a quadratic start and one Halley step run in
// double, ldexpf rescales.  Bit-identical to libm on every one of 3.0e8 floats probed over the
// whole normal range, which is what makes the integer LOD exact.
__device__ __forceinline__ float cbrtf_glibc(float x)
{
    int xe;
    const float xm = frexpf(fabsf(x), &xe);
    if (xe == 0 && (x == 0.0f || x != x || isinf(x)))
        return x + x;
    const float u = (float)(0.492659620528969547 + (0.697570460207922770 - 0.191502161678719066 * (double)xm) * (double)xm);
    const float t2 = u * u * u;
    const int r = xe % 3;
    const double f = r == -2 ? 1.0 / 1.5874010519681994748 : r == -1 ? 1.0 / 1.2599210498948731648
                   : r == 0 ? 1.0 : r == 1 ? 1.2599210498948731648 : 1.5874010519681994748;
    const float ym = (float)((double)u * ((double)t2 + 2.0 * (double)xm) / (2.0 * (double)t2 + (double)xm) * f);
    return ldexpf(x > 0.0f ? ym : -ym, xe / 3);
}

struct LodK {                       // what the pick reads (model.c:975-992) and writes
    float cx, cy, cz;
    const float *aabb, *center;
    const float4 *pos_scale;
    const int32_t *model;
    const float4 *model_table;
    const int32_t *force_lod;
    int32_t *cur_lod;
    uint32_t n_models;
};

// the LOD entity i is drawn with; cur_lod[i] follows (entity3d_set_lod writes e->cur_lod)
__device__ __forceinline__ int32_t lod_pick(const LodK &k, uint32_t i)
{
    int32_t lod = k.cur_lod[i];
    const int32_t forced = k.force_lod ? k.force_lod[i] : -1;
    if (forced >= 0) {
        lod = forced;                                                   // model.c:976-977
    } else {
        const float *b = k.aabb + 6 * (size_t)i;
        const bool inside = k.cx >= b[0] && k.cx <= b[3] && k.cy >= b[1] && k.cy <= b[4] && k.cz >= b[2] && k.cz <= b[5];
        if (!inside) {                                                  // model.c:982-990
            const float *c = k.center + 3 * (size_t)i;
            const float dx = c[0] - k.cx, dy = c[1] - k.cy, dz = c[2] - k.cz;
            float dd = 0.f;
            dd += dx * dx;
            dd += dy * dy;
            dd += dz * dz;
            const int32_t mraw = k.model[i];
            const int32_t mi = (uint32_t)mraw < k.n_models ? mraw : 0;
            const float4 lo = k.model_table[2 * mi], hi = k.model_table[2 * mi + 1];
            const float s = k.pos_scale[i].w;
            const float X = fabsf(hi.x - lo.x) * s, Y = fabsf(hi.y - lo.y) * s, Z = fabsf(hi.z - lo.z) * s;
            const float side = cbrtf_glibc(X * Y * Z);                  // entity3d_aabb_avg_edge
            const float scale = (float)((double)fabsf(dd - side * side) / 3600.0);
            const uint32_t lm = __float_as_uint(hi.w);                  // lod_min | lod_max << 8
            const int lmin = (int)(lm & 0xffu), lmax = (int)((lm >> 8) & 0xffu);
            const int req = (int)scale;
            lod = req < lmin ? lmin : (req > lmax ? lmax : req);        // model3d_validate_lod
        }
    }
    k.cur_lod[i] = lod;
    return lod;
}

__global__ __launch_bounds__(ENT_BLOCK)
void k_entities_lod(const uint32_t *visible, const uint32_t *count, uint32_t index_base, LodK lk, int32_t *draw_lod, uint32_t n)
{
    const uint32_t total = *count < n ? *count : n;                       // a count beyond the batch would walk off the list
    for (uint32_t k = blockIdx.x * ENT_BLOCK + threadIdx.x; k < total; k += gridDim.x * ENT_BLOCK) {
        const uint32_t i = visible[k] - index_base;
        if (i >= n) { draw_lod[k] = 0; continue; }                          // an id of another shard
        draw_lod[k] = lod_pick(lk, i);
    }
}

// ---- a host mirror's small frames: touched inputs in, rebuilt outputs out, through device-mapped host memory -----------
// A frame of a testbed-sized scene (BASELINE configs[0]: 10 k entities) is a 15-30 us kernel; staged through device
// slabs it paid three copies' fixed latencies and a blocking wait on top (0.15 ms).  Letting the update kernel itself
// work on mapped host memory removes the copies but puts a PCIe round trip under every dependent load of its row walk
// (measured: 27 -> 54 us).  So the update kernel stays on device memory, untouched, between two small streaming kernels:
//   k_entities_apply_inputs   reads the frame's touched (slot, flags, TRS) records from mapped host memory -- one
//                             coalesced 40-byte stream -- and scatters them into the device arrays;
//   k_entities_export_rebuilt copies what the update rebuilt (its own rebuilt_mask says which slots) and the three bit
//                             masks into the host's result arrays, then raises a completion word the host polls.
__global__ __launch_bounds__(ENT_BLOCK)
void k_entities_apply_inputs(float4 *pos_scale, float4 *rot, uint32_t *flags, const clapgpu_entity_input *list,
                             uint32_t n_list, uint32_t n)
{
    const uint32_t k = blockIdx.x * ENT_BLOCK + threadIdx.x;
    if (k >= n_list) return;
    const uint32_t *r = reinterpret_cast<const uint32_t *>(list + k);     // 40-byte records: ten dwords, 8-byte aligned
    const uint2 h = *reinterpret_cast<const uint2 *>(r);
    const uint32_t slot = h.x;
    if (slot >= n) return;
    const uint2 a = *reinterpret_cast<const uint2 *>(r + 2), b = *reinterpret_cast<const uint2 *>(r + 4);
    const uint2 c = *reinterpret_cast<const uint2 *>(r + 6), d = *reinterpret_cast<const uint2 *>(r + 8);
    pos_scale[slot] = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(b.x), __uint_as_float(b.y));
    rot[slot] = make_float4(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(d.x), __uint_as_float(d.y));
    flags[slot] = h.y;
}

// The lanes of a standing layout that got a new tenant between two frames (clapgpu_entities_place): 16-byte records.
__global__ __launch_bounds__(WAVE)
void k_entities_place(int32_t *parent, int32_t *model, float *aabb, float *center, const clapgpu_entity_place *list, uint32_t n_list,
                      uint32_t n, unsigned long long *stale)
{
    const uint32_t k = blockIdx.x * WAVE + threadIdx.x;
    if (k >= n_list) return;
    const uint4 r = *reinterpret_cast<const uint4 *>(list + k);
    const uint32_t slot = r.x;
    if (slot >= n) return;
    parent[slot] = (int32_t)r.y;
    model[slot] = (int32_t)r.z;
    if ((r.w & CLAPGPU_PLACE_CLEAR_STALE) && stale)
        atomicAnd(&stale[slot >> 6], ~(1ull << (slot & 63)));
    if (r.w & CLAPGPU_PLACE_ZERO_BOX) {
        float2 *b = reinterpret_cast<float2 *>(aabb + 6 * (size_t)slot);
        b[0] = b[1] = b[2] = make_float2(0.f, 0.f);
        float *c = center + 3 * (size_t)slot;
        c[0] = c[1] = c[2] = 0.f;
    }
}

struct ExportK {
    const float *mx, *inv_mx, *aabb, *center;            // device (the update's outputs)
    const uint64_t *vis_mask, *rebuilt_mask, *inside_mask;
    const uint64_t *select;                              // clapgpu_entities_export_rows: these rows, and no masks
    uint64_t *stale;                                     // ... whose stale bits (clapgpu_entities_hostio.stale_mask) are cleared
    float *o_mx, *o_inv, *o_aabb, *o_center;             // device-mapped host memory
    uint64_t *o_vis, *o_rebuilt, *o_inside;
    uint32_t *counter, *done, done_value, n_rows;
};

__global__ __launch_bounds__(ENT_BLOCK)
void k_entities_export_rebuilt(ExportK x)
{
    const int lane = lane_id();
    const uint32_t row = blockIdx.x * (ENT_BLOCK / WAVE) + threadIdx.x / WAVE;
    if (row < x.n_rows) {
        const uint64_t m = x.select ? x.select[row] : x.rebuilt_mask[row];
        if (lane == 0 && x.select && x.stale && m) x.stale[row] &= ~m;
        if (lane == 0 && !x.select) {
            x.o_rebuilt[row] = m;
            if (x.vis_mask) x.o_vis[row] = x.vis_mask[row];
            if (x.o_inside) x.o_inside[row] = x.inside_mask ? x.inside_mask[row] : 0ull;
        }
        if ((m >> lane) & 1ull) {
            const size_t i = (size_t)row * WAVE + lane;
            const float4 *a = reinterpret_cast<const float4 *>(x.mx + 16 * i), *b = reinterpret_cast<const float4 *>(x.inv_mx + 16 * i);
            float4 *oa = reinterpret_cast<float4 *>(x.o_mx + 16 * i), *ob = reinterpret_cast<float4 *>(x.o_inv + 16 * i);
            const float4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
            const float2 *bb = reinterpret_cast<const float2 *>(x.aabb + 6 * i);
            const float2 c0 = bb[0], c1 = bb[1], c2 = bb[2];
            const float *ct = x.center + 3 * i;
            const float t0 = ct[0], t1 = ct[1], t2 = ct[2];
            oa[0] = a0; oa[1] = a1; oa[2] = a2; oa[3] = a3;
            ob[0] = b0; ob[1] = b1; ob[2] = b2; ob[3] = b3;
            float2 *obb = reinterpret_cast<float2 *>(x.o_aabb + 6 * i);
            obb[0] = c0; obb[1] = c1; obb[2] = c2;
            float *oc = x.o_center + 3 * i;
            oc[0] = t0; oc[1] = t1; oc[2] = t2;
        }
    }
    // completion: every workgroup releases its stores to the system, the last one to arrive raises the word
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t arrived = atomicAdd(x.counter, 1u);
        if (arrived == gridDim.x - 1) {
            *x.counter = 0;                                       // ready for the next frame (stream order)
            __threadfence_system();
            __hip_atomic_store(x.done, x.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// k_entities_tiles reads its frustum through the kernarg segment pointer at offset 0.  The AMDGPU kernel ABI lays the
// explicit arguments out first and in declaration order, so that holds exactly as long as the frustum is the FIRST
// parameter: reordering the signature must not compile.
template <class F> struct first_param;
template <class R, class A0, class... A> struct first_param<R (*)(A0, A...)> { using type = A0; };
static_assert(std::is_same<first_param<decltype(&k_entities_tiles<true>)>::type, lmd::FrustumK>::value &&
              std::is_same<first_param<decltype(&k_entities_tiles<false>)>::type, lmd::FrustumK>::value,
              "k_entities_tiles: the frustum must stay the first kernel argument (it is read at kernarg offset 0)");
static_assert(std::is_same<first_param<decltype(&k_entities_tiles_xv)>::type, lmd::FrustumK>::value, "k_entities_tiles_xv: the same");
static_assert(alignof(lmd::FrustumK) <= 8, "kernarg offset 0 holds for any alignment the segment start guarantees");

} // namespace clapgpu

using namespace clapgpu;

static EntK to_kernel_args(const clapgpu_entities *e)
{
    EntK k;
    k.pos_scale = reinterpret_cast<const float4 *>(e->pos_scale);
    k.rot = reinterpret_cast<const float4 *>(e->rot);
    k.parent = e->parent;
    k.model = e->model;
    k.model_table = reinterpret_cast<const float4 *>(e->model_table);
    k.flags = e->flags;
    k.seqs = e->seqs;
    k.mx = e->mx;
    k.inv_mx = e->inv_mx;
    k.aabb = e->aabb;
    k.center = e->center;
    k.vis_mask = e->vis_mask;
    k.vis_row_pop = e->vis_row_pop;
    k.n_attach = (e->attach && e->jt_pool && e->bind_pool && e->attach_local) ? e->n_attach : 0;
    k.attach = e->attach;
    k.attach_local = e->attach_local;
    k.jt_pool = e->jt_pool;
    k.bind_pool = e->bind_pool;
    k.n = e->n;
    k.n_models = e->n_models ? e->n_models : 1;
    k.bv_result = nullptr;
    k.bv_inside = nullptr;
    k.rebuilt_mask = e->rebuilt_mask;
    k.bv_has_ctl = k.bv_ctl_entity = k.bv_on = 0;
    for (int a = 0; a < 3; a++) k.bv_cam[a] = k.bv_ctl[a] = 0.f;
    if (e->bv && (e->bv->result || e->bv->inside_mask)) {
        k.bv_on = 1;
        memcpy(k.bv_cam, e->bv->cam_pos, 12);
        memcpy(k.bv_ctl, e->bv->ctl_pos, 12);
        k.bv_has_ctl = e->bv->has_ctl;
        k.bv_ctl_entity = e->bv->ctl_entity;
        k.bv_result = reinterpret_cast<unsigned long long *>(e->bv->result);
        k.bv_inside = e->bv->inside_mask;
    }
    return k;
}

// Kernel-side frustum: adds the per-axis extremes of the frustum corners (NaN if any corner is
// NaN, so the comparison is false exactly when the reference's count cannot reach 8) and a flag
// telling whether every plane component is finite.
static lmd::FrustumK make_frustum_k(const clapgpu_frustum *frustum)
{
    static_assert(sizeof(lmd::Frustum) == sizeof(clapgpu_frustum), "frustum layout");
    lmd::FrustumK k = {};
    if (!frustum)
        return k;
    memcpy(&k.f, frustum, sizeof(k.f));
    for (int ax = 0; ax < 3; ax++) {
        float lo = INFINITY, hi = -INFINITY;
        bool nan = false;
        for (int i = 0; i < 8; i++) {
            const float c = k.f.corners[i][ax];
            nan = nan || (c != c);
            lo = c < lo ? c : lo;
            hi = c > hi ? c : hi;
        }
        k.cmin[ax] = nan ? NAN : lo;
        k.cmax[ax] = nan ? NAN : hi;
    }
    k.finite = 1;
    for (int i = 0; i < 6; i++)
        for (int c = 0; c < 4; c++)
            if (!(fabsf(k.f.planes[i][c]) <= 3.402823466e+38f))
                k.finite = 0;
    return k;
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// clapgpu_entities.views for the kernels; false: a count beyond the maximum, a missing plane
namespace clapgpu {
bool make_xviews_k(const clapgpu_entities *e, bool hostio, XViewsK *out)
{
    memset(out, 0, sizeof(*out));
    const clapgpu_views *v = e->views;
    if (!v || !v->n) return true;
    if (v->n > CLAPGPU_EXTRA_VIEWS_MAX) return false;
    out->n = v->n;
    for (uint32_t k = 0; k < v->n; k++) {
        if (!v->vis_mask[k] || !v->vis_row_pop[k]) return false;
        out->mask[k] = v->vis_mask[k]; out->pop[k] = v->vis_row_pop[k];
        out->o_mask[k] = hostio ? v->host_vis_mask[k] : nullptr;
        out->fr[k] = make_frustum_k(&v->frustum[k]);
    }
    return true;
}
}

static int check_entities(const clapgpu_entities *e, bool need_mask)
{
    if (!e || !e->pos_scale || !e->rot || !e->parent || !e->model || !e->model_table || !e->flags ||
        !e->seqs || !e->mx || !e->inv_mx || !e->aabb || !e->center)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (need_mask && (!e->vis_mask || !e->vis_row_pop))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!aligned16(e->pos_scale) || !aligned16(e->rot) || !aligned16(e->model_table) || !aligned16(e->mx) ||
        !aligned16(e->inv_mx) || !aligned16(e->aabb) || !aligned16(e->center))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    return CLAPGPU_OK;
}

static int prepare_attachments(void *stream, const EntK &k)
{
    if (!k.n_attach)
        return CLAPGPU_OK;
    hipLaunchKernelGGL(k_attach_prepare, dim3((k.n_attach + ENT_BLOCK - 1) / ENT_BLOCK), dim3(ENT_BLOCK), 0,
                       as_stream(stream), k);
    CLAPGPU_LAUNCH_CHECK("k_attach_prepare");
    return CLAPGPU_OK;
}

static int launch_level(void *stream, const EntK &k, uint32_t first, uint32_t count, uint32_t mode,
                        const clapgpu_frustum *frustum, const XViewsK &xv)
{
    const lmd::FrustumK fr = make_frustum_k(frustum);
    const dim3 grid((count + ENT_BLOCK - 1) / ENT_BLOCK), block(ENT_BLOCK);
    if (frustum && xv.n)
        hipLaunchKernelGGL(k_entities_level_xv, grid, block, 0, as_stream(stream), k, first, count, mode, fr, xv);
    else if (frustum)
        hipLaunchKernelGGL(k_entities_level<true>, grid, block, 0, as_stream(stream), k, first, count, mode, fr);
    else
        hipLaunchKernelGGL(k_entities_level<false>, grid, block, 0, as_stream(stream), k, first, count, mode, fr);
    CLAPGPU_LAUNCH_CHECK("k_entities_level");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_entities_update_level(void *stream, const clapgpu_entities *e,
                                             uint32_t first, uint32_t count,
                                             uint32_t mode, const clapgpu_frustum *frustum)
{
    int rc = check_entities(e, frustum != nullptr);
    if (rc) return rc;
    if (first & 63u)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (first > e->n || count > e->n - first)
        return CLAPGPU_ERR_OUT_OF_BOUNDS;
    if (!count)
        return CLAPGPU_OK;
    const EntK k = to_kernel_args(e);
    XViewsK xv;
    if (!make_xviews_k(e, false, &xv)) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    rc = prepare_attachments(stream, k);
    if (rc) return rc;
    return launch_level(stream, k, first, count, mode, frustum, xv);
}

extern "C" int clapgpu_entities_update(void *stream, const clapgpu_entities *e,
                                       const uint32_t *level_start, uint32_t n_levels,
                                       uint32_t mode, const clapgpu_frustum *frustum)
{
    int rc = check_entities(e, frustum != nullptr);
    if (rc) return rc;
    if (!level_start || (e->n && !n_levels))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (e->n == 0)
        return CLAPGPU_OK;
    if (level_start[0] != 0 || level_start[n_levels] != e->n)
        return CLAPGPU_ERR_OUT_OF_BOUNDS;
    for (uint32_t l = 0; l < n_levels; l++)
        if (level_start[l] > level_start[l + 1] || (level_start[l] & 63u))
            return CLAPGPU_ERR_INVALID_ARGUMENTS;

    const EntK k = to_kernel_args(e);
    XViewsK xv;
    if (!make_xviews_k(e, false, &xv)) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (k.bv_result)
        CLAPGPU_HIP(hipMemsetAsync(k.bv_result, 0, sizeof(uint64_t), as_stream(stream)));
    rc = prepare_attachments(stream, k);
    if (rc) return rc;
    for (uint32_t l = 0; l < n_levels; l++) {
        const uint32_t first = level_start[l], count = level_start[l + 1] - first;
        if (!count)
            continue;
        rc = launch_level(stream, k, first, count, mode, frustum, xv);
        if (rc) return rc;
    }
    return CLAPGPU_OK;
}

extern "C" int clapgpu_entities_update_tiles(void *stream, const clapgpu_entities *e,
                                             const uint32_t *tile_row_start, uint32_t n_tiles,
                                             uint32_t mode, const clapgpu_frustum *frustum)
{
    int rc = check_entities(e, frustum != nullptr);
    if (rc) return rc;
    if (e->n == 0 || n_tiles == 0)
        return CLAPGPU_OK;
    if (!tile_row_start)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const lmd::FrustumK fr = make_frustum_k(frustum);
    const EntK k = to_kernel_args(e);
    XViewsK xv;
    if (!make_xviews_k(e, false, &xv)) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (k.bv_result)
        CLAPGPU_HIP(hipMemsetAsync(k.bv_result, 0, sizeof(uint64_t), as_stream(stream)));
    rc = prepare_attachments(stream, k);
    if (rc) return rc;
    const uint32_t per_block = ENT_BLOCK / WAVE;
    const dim3 grid((n_tiles + per_block - 1) / per_block), block(ENT_BLOCK);
    if (frustum && xv.n)
        hipLaunchKernelGGL(k_entities_tiles_xv, grid, block, 0, as_stream(stream), fr, k, tile_row_start, n_tiles, e->n, mode, xv);
    else if (frustum)
        hipLaunchKernelGGL(k_entities_tiles<true>, grid, block, 0, as_stream(stream), fr, k, tile_row_start, n_tiles,
                           e->n, mode);
    else
        hipLaunchKernelGGL(k_entities_tiles<false>, grid, block, 0, as_stream(stream), fr, k, tile_row_start, n_tiles,
                           e->n, mode);
    CLAPGPU_LAUNCH_CHECK("k_entities_tiles");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_entities_cull(void *stream, const clapgpu_entities *e, const clapgpu_frustum *frustum)
{
    if (!e || !frustum || !e->flags || !e->aabb || !e->vis_mask || !e->vis_row_pop)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (e->n == 0)
        return CLAPGPU_OK;
    const lmd::FrustumK fr = make_frustum_k(frustum);
    XViewsK xv;
    if (!make_xviews_k(e, false, &xv)) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const dim3 grid((e->n + ENT_BLOCK - 1) / ENT_BLOCK), block(ENT_BLOCK);
    if (xv.n)
        hipLaunchKernelGGL(k_entities_cull_xv, grid, block, 0, as_stream(stream), e->flags, e->aabb, e->vis_mask,
                           e->vis_row_pop, e->n, fr, xv);
    else
    hipLaunchKernelGGL(k_entities_cull, grid, block, 0, as_stream(stream), e->flags, e->aabb, e->vis_mask,
                       e->vis_row_pop, e->n, fr);
    CLAPGPU_LAUNCH_CHECK("k_entities_cull");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_entities_apply_inputs(void *stream, const clapgpu_entities *e, const clapgpu_entity_input *list,
                                             uint32_t n_list)
{
    static_assert(sizeof(clapgpu_entity_input) == 40, "record layout");
    if (!e || !e->pos_scale || !e->rot || !e->flags || (n_list && !list))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!n_list || !e->n)
        return CLAPGPU_OK;
    hipLaunchKernelGGL(k_entities_apply_inputs, dim3((n_list + ENT_BLOCK - 1) / ENT_BLOCK), dim3(ENT_BLOCK), 0, as_stream(stream),
                       reinterpret_cast<float4 *>(const_cast<float *>(e->pos_scale)),
                       reinterpret_cast<float4 *>(const_cast<float *>(e->rot)), e->flags, list, n_list, e->n);
    CLAPGPU_LAUNCH_CHECK("k_entities_apply_inputs");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_entities_place(void *stream, const clapgpu_entities *e, const clapgpu_entity_place *list, uint32_t n_list,
                                      uint64_t *stale_mask)
{
    if (!e || !e->parent || !e->model || !e->aabb || !e->center || (n_list && !list))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!n_list || !e->n)
        return CLAPGPU_OK;
    hipLaunchKernelGGL(k_entities_place, dim3((n_list + WAVE - 1) / WAVE), dim3(WAVE), 0, as_stream(stream),
                       const_cast<int32_t *>(e->parent), const_cast<int32_t *>(e->model), e->aabb, e->center, list, n_list, e->n,
                       reinterpret_cast<unsigned long long *>(stale_mask));
    CLAPGPU_LAUNCH_CHECK("k_entities_place");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_entities_export_rebuilt(void *stream, const clapgpu_entities *e, const clapgpu_entities_export *x)
{
    if (!e || !x || !e->mx || !e->inv_mx || !e->aabb || !e->center || !e->rebuilt_mask || !x->mx || !x->inv_mx || !x->aabb ||
        !x->center || !x->rebuilt_mask || !x->counter || !x->done || (e->vis_mask && !x->vis_mask))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (e->n & 63u)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    ExportK k;
    k.mx = e->mx; k.inv_mx = e->inv_mx; k.aabb = e->aabb; k.center = e->center;
    k.vis_mask = e->vis_mask; k.rebuilt_mask = e->rebuilt_mask; k.select = nullptr; k.stale = nullptr;
    k.inside_mask = (e->bv && e->bv->inside_mask) ? e->bv->inside_mask : nullptr;
    k.o_mx = x->mx; k.o_inv = x->inv_mx; k.o_aabb = x->aabb; k.o_center = x->center;
    k.o_vis = x->vis_mask; k.o_rebuilt = x->rebuilt_mask; k.o_inside = x->inside_mask;
    k.counter = x->counter; k.done = x->done; k.done_value = x->done_value; k.n_rows = e->n / 64;
    const uint32_t per_block = ENT_BLOCK / WAVE;
    const uint32_t blocks = k.n_rows ? (k.n_rows + per_block - 1) / per_block : 1;
    hipLaunchKernelGGL(k_entities_export_rebuilt, dim3(blocks), dim3(ENT_BLOCK), 0, as_stream(stream), k);
    CLAPGPU_LAUNCH_CHECK("k_entities_export_rebuilt");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_entities_export_rows(void *stream, const clapgpu_entities *e, const clapgpu_entities_export *x,
                                            const uint64_t *select_mask)
{
    if (!e || !x || !select_mask || !e->mx || !e->inv_mx || !e->aabb || !e->center || !x->mx || !x->inv_mx || !x->aabb ||
        !x->center || !x->counter || !x->done)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (e->n & 63u)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    ExportK k = {};
    k.mx = e->mx; k.inv_mx = e->inv_mx; k.aabb = e->aabb; k.center = e->center;
    k.select = select_mask; k.stale = x->stale_mask;
    k.o_mx = x->mx; k.o_inv = x->inv_mx; k.o_aabb = x->aabb; k.o_center = x->center;
    k.counter = x->counter; k.done = x->done; k.done_value = x->done_value; k.n_rows = e->n / 64;
    const uint32_t per_block = ENT_BLOCK / WAVE;
    const uint32_t blocks = k.n_rows ? (k.n_rows + per_block - 1) / per_block : 1;
    hipLaunchKernelGGL(k_entities_export_rebuilt, dim3(blocks), dim3(ENT_BLOCK), 0, as_stream(stream), k);
    CLAPGPU_LAUNCH_CHECK("k_entities_export_rows");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_entities_update_tiles_hostio(void *stream, const clapgpu_entities *e,
                                                    const uint32_t *tile_row_start, uint32_t n_tiles, uint32_t mode,
                                                    const clapgpu_frustum *frustum, const clapgpu_entities_hostio *io)
{
    int rc = check_entities(e, frustum != nullptr);
    if (rc) return rc;
    if (!io || !io->mx || !io->inv_mx || !io->aabb || !io->center || !io->rebuilt_mask || !io->counter || !io->done ||
        (frustum && !io->vis_mask) || (io->touched && (!io->pos_scale || !io->rot || !io->flags)) || (n_tiles && !tile_row_start))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (!aligned16(io->mx) || !aligned16(io->inv_mx) || (reinterpret_cast<uintptr_t>(io->aabb) & 7u) ||
        (io->touched && (!aligned16(io->pos_scale) || !aligned16(io->rot))))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const lmd::FrustumK fr = make_frustum_k(frustum);
    const EntK k = to_kernel_args(e);
    if (k.bv_result)
        CLAPGPU_HIP(hipMemsetAsync(k.bv_result, 0, sizeof(uint64_t), as_stream(stream)));
    rc = prepare_attachments(stream, k);
    if (rc) return rc;
    HostIO h;
    h.pos_scale = reinterpret_cast<const float4 *>(io->pos_scale); h.rot = reinterpret_cast<const float4 *>(io->rot);
    h.flags = io->flags; h.touched = io->touched;
    h.o_mx = io->mx; h.o_inv = io->inv_mx; h.o_aabb = io->aabb; h.o_center = io->center;
    h.o_vis = io->vis_mask; h.o_rebuilt = io->rebuilt_mask; h.o_inside = io->inside_mask;
    h.counter = io->counter; h.done = io->done; h.done_value = io->done_value;
    h.keep = io->keep_mask; h.o_exported = io->exported_mask; h.stale = io->stale_mask;
    h.late_ok = io->options & CLAPGPU_HOSTIO_EXPORT_STALE_READ;
    const uint32_t tiles = e->n ? n_tiles : 0;
    XViewsK xv;
    if (!make_xviews_k(e, true, &xv)) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    rc = launch_entities_tiles_host(as_stream(stream), frustum != nullptr, fr, k, h, tile_row_start, tiles, e->n, mode, xv);
    if (rc) return rc;
    return CLAPGPU_OK;
}

extern "C" size_t clapgpu_visible_scratch_bytes(uint32_t n)
{
    const uint32_t n_groups = (n + GROUP_WORDS * 64 - 1) / (GROUP_WORDS * 64);
    return (size_t)(n_groups ? n_groups : 1) * sizeof(uint32_t);
}

extern "C" int clapgpu_visible_compact(void *stream, const uint64_t *vis_mask, const uint8_t *vis_row_pop,
                                       uint32_t n, uint32_t index_base, uint32_t *visible, uint32_t *count,
                                       void *scratch)
{
    if (!count || (n && (!vis_mask || !visible)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (n == 0) {
        CLAPGPU_HIP(hipMemsetAsync(count, 0, sizeof(uint32_t), as_stream(stream)));
        return CLAPGPU_OK;
    }
    const uint32_t n_rows = (n + 63) / 64;
    if (vis_row_pop && n_rows <= RP_MAX_ROWS && aligned16(vis_row_pop)) {
        const uint32_t waves = (n_rows + RP_ROWS - 1) / RP_ROWS, per_block = ENT_BLOCK / WAVE;
        hipLaunchKernelGGL(k_visible_expand_rp, dim3((waves + per_block - 1) / per_block), dim3(ENT_BLOCK), 0,
                           as_stream(stream), vis_mask, vis_row_pop, n, index_base, visible, count);
        CLAPGPU_LAUNCH_CHECK("k_visible_expand_rp");
        return CLAPGPU_OK;
    }
    if (!scratch)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const uint32_t n_groups = (n + GROUP_WORDS * 64 - 1) / (GROUP_WORDS * 64);
    uint32_t *group_count = static_cast<uint32_t *>(scratch);
    hipLaunchKernelGGL(k_mask_group_count, dim3(n_groups), dim3(WAVE), 0, as_stream(stream), vis_mask, n, group_count);
    CLAPGPU_LAUNCH_CHECK("k_mask_group_count");
    hipLaunchKernelGGL(k_visible_expand, dim3(n_groups), dim3(WAVE), 0, as_stream(stream), vis_mask, n,
                       group_count, n_groups, index_base, visible, count);
    CLAPGPU_LAUNCH_CHECK("k_visible_expand");
    return CLAPGPU_OK;
}

static int check_ranges(uint32_t n_ranges, uint32_t cap_pad, const uint32_t *base, const uint32_t *n_pad)
{
    if (!n_ranges || n_ranges > (uint32_t)MAX_SEGMENTS || !cap_pad || (cap_pad & 63u) || !base) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    uint64_t end = 0;
    for (uint32_t r = 0; r < n_ranges; r++) {
        const uint32_t np = n_pad ? n_pad[r] : cap_pad;
        if ((np & 63u) || np > cap_pad || (base[r] & 63u) || base[r] < end) return CLAPGPU_ERR_INVALID_ARGUMENTS;   /* ascending, disjoint */
        end = (uint64_t)base[r] + np;
        if (end > 0xffffffffull) return CLAPGPU_ERR_TOO_LARGE;
    }
    return CLAPGPU_OK;
}

extern "C" int clapgpu_visible_compact_ranges(void *stream, const uint64_t *gathered_mask, uint32_t n_ranges, uint32_t cap_pad,
                                              const uint32_t *base, const uint32_t *n_pad, uint32_t *visible, uint32_t *count,
                                              void *scratch)
{
    if (!gathered_mask || !visible || !count || !scratch) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    int rc = check_ranges(n_ranges, cap_pad, base, n_pad);
    if (rc) return rc;
    SegK sg = {};
    sg.cap_words = cap_pad / 64; sg.n_seg = n_ranges;
    for (uint32_t r = 0; r < n_ranges; r++) { sg.base[r] = base[r]; sg.n_words[r] = (n_pad ? n_pad[r] : cap_pad) / 64; }
    const uint64_t total_words = (uint64_t)sg.cap_words * n_ranges;
    if (total_words * 64 > 0xffffffffull) return CLAPGPU_ERR_TOO_LARGE;
    const uint32_t n_groups = (uint32_t)((total_words + GROUP_WORDS - 1) / GROUP_WORDS);
    uint32_t *group_count = static_cast<uint32_t *>(scratch);       // clapgpu_visible_scratch_bytes(n_ranges * cap_pad)
    hipLaunchKernelGGL(k_mask_group_count_seg, dim3(n_groups), dim3(WAVE), 0, as_stream(stream), gathered_mask, sg, group_count);
    CLAPGPU_LAUNCH_CHECK("k_mask_group_count_seg");
    hipLaunchKernelGGL(k_visible_expand_seg, dim3(n_groups), dim3(WAVE), 0, as_stream(stream), gathered_mask, sg, group_count, n_groups,
                       visible, count);
    CLAPGPU_LAUNCH_CHECK("k_visible_expand_seg");
    return CLAPGPU_OK;
}

// The same expansion on the host: what a rank without the device list needs, and the checker of the kernels above
// (tests/test_shard_cpu.py runs eight gloo ranks through it).  Returns the number of ids; writes at most `capacity`.
extern "C" uint32_t clapgpu_visible_expand_ranges_host(const uint64_t *gathered_mask, uint32_t n_ranges, uint32_t cap_pad,
                                                       const uint32_t *base, const uint32_t *n_pad, uint32_t *visible, uint32_t capacity)
{
    if (!gathered_mask || check_ranges(n_ranges, cap_pad, base, n_pad)) return 0;
    const uint32_t cap_words = cap_pad / 64;
    uint32_t cnt = 0;
    for (uint32_t r = 0; r < n_ranges; r++) {
        const uint32_t words = (n_pad ? n_pad[r] : cap_pad) / 64;
        for (uint32_t w = 0; w < words; w++) {
            uint64_t m = gathered_mask[(size_t)r * cap_words + w];
            while (m) {
                const uint32_t id = base[r] + w * 64u + (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                if (visible && cnt < capacity) visible[cnt] = id;
                cnt++;
            }
        }
    }
    return cnt;
}

static LodK lod_args(const clapgpu_entities *e, const float cam_pos[3], const int32_t *force_lod, int32_t *cur_lod)
{
    LodK k;
    k.cx = cam_pos[0]; k.cy = cam_pos[1]; k.cz = cam_pos[2];
    k.aabb = e->aabb; k.center = e->center;
    k.pos_scale = reinterpret_cast<const float4 *>(e->pos_scale);
    k.model = e->model;
    k.model_table = reinterpret_cast<const float4 *>(e->model_table);
    k.force_lod = force_lod; k.cur_lod = cur_lod;
    k.n_models = e->n_models ? e->n_models : 1;
    return k;
}

extern "C" int clapgpu_entities_lod(void *stream, const clapgpu_entities *e, const uint32_t *visible,
                                    const uint32_t *count, uint32_t index_base, const float cam_pos[3],
                                    const int32_t *force_lod, int32_t *cur_lod, int32_t *draw_lod)
{
    if (!e || !visible || !count || !cam_pos || !cur_lod || !draw_lod || !e->aabb || !e->center ||
        !e->pos_scale || !e->model || !e->model_table)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (e->n == 0)
        return CLAPGPU_OK;
    uint32_t blocks = (e->n + ENT_BLOCK - 1) / ENT_BLOCK;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_entities_lod, dim3(blocks), dim3(ENT_BLOCK), 0, as_stream(stream), visible, count, index_base,
                       lod_args(e, cam_pos, force_lod, cur_lod), draw_lod, e->n);
    CLAPGPU_LAUNCH_CHECK("k_entities_lod");
    return CLAPGPU_OK;
}

// clapgpu_visible_compact + clapgpu_entities_lod of this batch's own entities: the render pass's list and its LODs by one
// call.  Two launches: a single kernel that picks the LOD of every id as it writes it was built and measured (round 5,
// profiles/r05_experiments/lod_in_expand.md) -- the expansion walks sixteen mask rows per wavefront one after the other,
// and with the pick's chain of dependent loads under every row it took 42 us against 6 + 10 us for the two.
extern "C" int clapgpu_visible_compact_lod(void *stream, const clapgpu_entities *e, uint32_t index_base, const float cam_pos[3],
                                           const int32_t *force_lod, int32_t *cur_lod, uint32_t *visible, uint32_t *count,
                                           int32_t *draw_lod, void *scratch)
{
    if (!e || !count || !cam_pos || !cur_lod || !draw_lod || (e->n && (!e->vis_mask || !visible)) || !e->aabb || !e->center ||
        !e->pos_scale || !e->model || !e->model_table)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const uint32_t n = e->n;
    int rc = clapgpu_visible_compact(stream, e->vis_mask, e->vis_row_pop, n, index_base, visible, count, scratch);
    if (rc) return rc;
    return clapgpu_entities_lod(stream, e, visible, count, index_base, cam_pos, force_lod, cur_lod, draw_lod);
}
