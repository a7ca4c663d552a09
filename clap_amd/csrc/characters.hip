// characters.hip -- the per-character feeder in front of default_update, for gfx950.
//
// Replaces character_update() (character.c:583-611) for every character of the scene: the limbo
// teleport out of the 8-entry position history (history_newest / _fetch / _push, character.c:546-581),
// the body read-back the function does itself (phys_body_update / phys_body_set_position,
// physics.c:789-812, 208-225; characters keep their own rotation) and the `moved` result the host
// turns into character_set_moved().  character_motion_reset() acts on the one controlled character
// only and stays host code, like character_move()'s ODE sweeps.  The chained orig_update =
// default_update is the entity kernel: run this first, it sets CLAPGPU_E_DIRTY on what it moves.
//
// One lane per character; state is SoA (history 96 B + 5 B per character).  C3 has 50 k characters:
// ~10 MB of traffic, launch-latency bound.
#include <string.h>
#include "common.h"

namespace clapgpu {

constexpr int CHAR_BLOCK = 256;
constexpr int HIST = CLAPGPU_POS_HISTORY_MAX;

struct CharK {
    uint32_t        n;
    float           limbo_height;
    const uint32_t *entity;
    const int32_t  *body;
    float          *hist_pos;
    uint32_t       *hist_head;
    uint8_t        *hist_wrapped;
    const uint8_t  *airborne;
    uint8_t        *moved;
    float4         *pos_scale;
    uint32_t       *entity_flags;
    uint32_t        n_entities, n_bodies;
    double         *body_pos;
    double         *body_geom_records;  // clapgpu_bodies.geom_records (or NULL): the narrowphase's copy of the position
    const double   *body_lvel, *body_yoffset;
};

__device__ __forceinline__ void character_update(const CharK &k, uint32_t c);

__global__ __launch_bounds__(CHAR_BLOCK)
void k_characters_update(CharK k)
{
    const uint32_t c = blockIdx.x * CHAR_BLOCK + threadIdx.x;
    if (c >= k.n) return;
    character_update(k, c);
}

// The character hooks and animated_update's clock (model.c:1563-1592; k_animation_time in pose.hip: the same four lines)
// as ONE launch: two per-character passes over different state, neither reads what the other writes, and both sit in
// front of kernels that need them (entity update / pose) -- as two launches the second cost a dependent launch's latency
// for 7 us of work.  Blocks [0, char_blocks) run the hooks, the rest the clock.
__global__ __launch_bounds__(CHAR_BLOCK)
void k_characters_update_clock(CharK k, uint32_t char_blocks, clapgpu_anim_clock clk, double now, const double *now_dev)
{
    if (blockIdx.x < char_blocks) {
        const uint32_t c = blockIdx.x * CHAR_BLOCK + threadIdx.x;
        if (c < k.n) character_update(k, c);
        return;
    }
    const uint32_t c = (blockIdx.x - char_blocks) * CHAR_BLOCK + threadIdx.x;
    if (c >= clk.n_chars) return;
    if (now_dev) now = *now_dev;
    const double ft = (now - clk.ani_time[c]) * (double)clk.speed[c];
    clk.frame_time[c] = (float)ft;
    const uint32_t an = clk.anim[c];
    const bool ended = an < clk.n_anims && ft >= (double)clk.time_end[an];
    clk.ended[c] = ended ? 1 : 0;
    if (ended && clk.restart[c])
        clk.ani_time[c] = now;                                      // animation_next -> animation_start
}

__device__ __forceinline__ void character_update(const CharK &k, uint32_t c)
{
    const uint32_t e = k.entity[c];
    if (e >= k.n_entities) return;
    int32_t b = k.body ? k.body[c] : -1;
    if (!k.body_pos || (uint32_t)b >= k.n_bodies) b = -1;
    float *hp = k.hist_pos + (size_t)c * HIST * 3;
    uint32_t head = k.hist_head[c];
    bool wrapped = k.hist_wrapped[c] != 0;
    float4 ps = k.pos_scale[e];
    bool dirty = false;

    float last[3] = { 0.f, 0.f, 0.f };                              // history_newest
    if (head) { last[0] = hp[3 * (head - 1)]; last[1] = hp[3 * (head - 1) + 1]; last[2] = hp[3 * (head - 1) + 2]; }
    else if (wrapped) { last[0] = hp[3 * (HIST - 1)]; last[1] = hp[3 * (HIST - 1) + 1]; last[2] = hp[3 * (HIST - 1) + 2]; }

    float dot = 0.f;                                                // vec3_mul_inner
    dot += last[0] * last[0];
    dot += last[1] * last[1];
    dot += last[2] * last[2];
    if ((double)dot > 0.0 && fabsf(ps.y - last[1]) >= k.limbo_height) {        // character.c:595
        const float *src = wrapped ? hp + 3 * head : hp;           // history_fetch
        wrapped = false;
        head = 0;
        ps.x = src[0]; ps.y = src[1]; ps.z = src[2];                // entity3d_position
        dirty = true;
        if (b >= 0) {                                               // phys_body_set_position: y + yoffset
            k.body_pos[3 * (size_t)b + 0] = (double)ps.x;
            k.body_pos[3 * (size_t)b + 1] = (double)ps.y + k.body_yoffset[b];
            k.body_pos[3 * (size_t)b + 2] = (double)ps.z;
            if (k.body_geom_records) {                              // the geom follows its body (dGeomSetBody): so does its record
                double *r = k.body_geom_records + 8 * (size_t)b;
                r[0] = (double)ps.x; r[1] = (double)ps.y + k.body_yoffset[b]; r[2] = (double)ps.z;
            }
        }
    }

    uint8_t moved = 0;
    if (b >= 0) {                                                   // phys_body_update, position only
        const double *bp = k.body_pos + 3 * (size_t)b, *v = k.body_lvel + 3 * (size_t)b;
        ps.x = (float)bp[0];
        ps.y = (float)(bp[1] - k.body_yoffset[b]);
        ps.z = (float)bp[2];
        dirty = true;
        if (sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) > 1e-3) {
            if (!k.airborne[c]) {                                   // history_push
                hp[3 * head] = ps.x; hp[3 * head + 1] = ps.y; hp[3 * head + 2] = ps.z;
                head = (head + 1) % HIST;
                if (!wrapped && !head) wrapped = true;
            }
            moved = 1;
        }
    }
    k.hist_head[c] = head;
    k.hist_wrapped[c] = wrapped ? 1 : 0;
    k.moved[c] = moved;
    if (dirty) {
        k.pos_scale[e] = ps;
        k.entity_flags[e] |= CLAPGPU_E_DIRTY;                       // one character per entity: no other writer
    }
}

} // namespace clapgpu

using namespace clapgpu;

static int char_args(const clapgpu_characters *c, const clapgpu_entities *e, const clapgpu_bodies *b, CharK &k);

extern "C" int clapgpu_characters_update(void *stream, const clapgpu_characters *c, const clapgpu_entities *e,
                                         const clapgpu_bodies *b)
{
    if (!c || !e)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (c->n == 0)
        return CLAPGPU_OK;
    CharK k;
    int rc = char_args(c, e, b, k);
    if (rc) return rc;
    hipLaunchKernelGGL(k_characters_update, dim3((c->n + CHAR_BLOCK - 1) / CHAR_BLOCK), dim3(CHAR_BLOCK), 0,
                       as_stream(stream), k);
    CLAPGPU_LAUNCH_CHECK("k_characters_update");
    return CLAPGPU_OK;
}

// clapgpu_characters_update + clapgpu_animation_time(_dev) in one launch (now_dev != NULL: the clock's `now` from a device
// double, as clapgpu_animation_time_dev).  Either part may be empty.
extern "C" int clapgpu_characters_update_clock(void *stream, const clapgpu_characters *c, const clapgpu_entities *e,
                                               const clapgpu_bodies *b, const clapgpu_anim_clock *clk, double now, const double *now_dev)
{
    if (!c || !e || !clk)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (clk->n_chars && (!clk->anim || !clk->time_end || !clk->ani_time || !clk->speed || !clk->restart || !clk->frame_time || !clk->ended))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    CharK k = {};
    if (c->n) {
        int rc = char_args(c, e, b, k);
        if (rc) return rc;
    }
    const uint32_t cb = (c->n + CHAR_BLOCK - 1) / CHAR_BLOCK, tb = (clk->n_chars + CHAR_BLOCK - 1) / CHAR_BLOCK;
    if (cb + tb == 0) return CLAPGPU_OK;
    hipLaunchKernelGGL(k_characters_update_clock, dim3(cb + tb), dim3(CHAR_BLOCK), 0, as_stream(stream), k, cb, *clk, now, now_dev);
    CLAPGPU_LAUNCH_CHECK("k_characters_update_clock");
    return CLAPGPU_OK;
}

static int char_args(const clapgpu_characters *c, const clapgpu_entities *e, const clapgpu_bodies *b, CharK &k)
{
    if (!c->entity || !c->hist_pos || !c->hist_head || !c->hist_wrapped || !c->airborne || !c->moved ||
        !e->pos_scale || !e->flags)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (b && b->n && (!b->pos || !b->lvel || !b->yoffset))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    k.n = c->n; k.limbo_height = c->limbo_height;
    k.entity = c->entity; k.body = c->body;
    k.hist_pos = c->hist_pos; k.hist_head = c->hist_head; k.hist_wrapped = c->hist_wrapped;
    k.airborne = c->airborne; k.moved = c->moved;
    k.pos_scale = reinterpret_cast<float4 *>(const_cast<float *>(e->pos_scale));
    k.entity_flags = e->flags;
    k.n_entities = e->n;
    k.n_bodies = b ? b->n : 0;
    k.body_pos = b ? b->pos : nullptr;
    k.body_geom_records = b ? b->geom_records : nullptr;
    k.body_lvel = b ? b->lvel : nullptr;
    k.body_yoffset = b ? b->yoffset : nullptr;
    return CLAPGPU_OK;
}
