// lights.hip -- clustered-lighting tile masks for gfx950.
//
// Replaces light_grid_compute() (light.c:88-154): for every screen tile, the 128-bit mask of the
// light slots whose screen-space disc reaches one of the tile's corners (directional slots always).
// The reference loops lights x tiles on the host and uploads the masks as an RGBA32UI texture;
// here the masks are produced in device memory in the same layout (tiles[gy * twidth + gx] =
// 4 x u32, bit idx % 32 of word idx / 32).
//
// Mapping: one lane per light, twice (lane l holds slots l and l + 64 in registers), one wavefront
// per run of 16 tiles: a tile's mask is two 64-lane ballots of the disc test, so the 128-light loop
// of the reference becomes two vector compares and the per-light constants never leave registers.
// Lane k of the wave keeps tile k's mask and the run is stored as one 256-byte row.
//
// Numerics: the reference's fp32 operation order (no contraction, IEEE divide and sqrt); the two
// double-typed comparisons (light.c:116,118) are made in double.  The masks are bit-exact.
#include <string.h>
#include <math.h>
#include "common.h"
#include "lm_dev.h"

namespace clapgpu {

constexpr int LIGHT_BLOCK = 256;

struct LightGridArgs {
    uint32_t        nr_lights;
    const float    *pos, *color, *attenuation;
    const int32_t  *is_dir;
    const uint32_t *active;
    float           view[16], mvp[16];
    float           fx;
    uint32_t        width, height, cell, twidth, theight;
    uint4          *tiles;
};

// light.c:301-309; max3 = max(a, max(b, c)), max(a, b) = a > b ? a : b (util.h:200-203)
__device__ __forceinline__ float light_radius(const float *color, const float *att)
{
    const float bc = color[1] > color[2] ? color[1] : color[2];
    const float comp_max = color[0] > bc ? color[0] : bc;
    const float cutoff = 1.0f / 256.0f;                              // LIGHT_CUTOFF, shader_constants.h:15
    return (-att[1] + sqrtf(att[1] * att[1] - 4.0f * att[2] * (att[0] - comp_max / cutoff))) / (2.0f * att[2]);
}

constexpr int LIGHT_TILES_PER_WAVE = 16;

// Per-slot constants of the disc test (light.c:100-127).  A slot that contributes nothing gets
// rsq = -1 (no squared distance is below it), a directional slot rsq = +inf at (0, 0): every
// corner distance is finite, so its test always passes -- the `goto grid` / `continue` of the reference.
__device__ __forceinline__ void light_constants(const LightGridArgs &a, uint32_t t, float &x, float &y, float &r2)
{
    x = 0.f; y = 0.f; r2 = -1.0f;
    if (t >= a.nr_lights || !a.active[t]) return;
    if (a.is_dir[t]) { r2 = __builtin_inff(); return; }
    const float lp[4] = { a.pos[3 * t], a.pos[3 * t + 1], a.pos[3 * t + 2], 1.0f };
    float vp[4], ndc[4];
    lmd::mul_vec4(vp, a.view, lp);                                   // light.c:108-109
    lmd::mul_vec4(ndc, a.mvp, lp);
    const float s = 1.0f / ndc[3];                                   // vec3_scale: w itself is kept
    ndc[0] = ndc[0] * s; ndc[1] = ndc[1] * s; ndc[2] = ndc[2] * s;
    if ((double)fabsf(ndc[3]) < 1e-3 || (double)ndc[2] > 1.0) return;  // light.c:112-114
    const float radius = light_radius(a.color + 3 * t, a.attenuation + 3 * t) * a.fx / -vp[2] * ((float)a.width / 2.0f);
    r2 = radius * radius;
    x = (ndc[0] + 1.0f) / 2.0f * (float)a.width;
    y = (1.0f - ndc[1]) / 2.0f * (float)a.height;
}

// any of the tile's four corners inside the light's disc (light.c:138-147); vec2_mul_inner: p = 0; p += d0*d0; p += d1*d1
__device__ __forceinline__ bool disc_reaches(float sx, float sy, float r2, float x0, float x1, float y0, float y1)
{
    const float dx0 = sx - x0, dx1 = sx - x1, dy0 = sy - y0, dy1 = sy - y1;
    float d00 = 0.f; d00 += dx0 * dx0; d00 += dy0 * dy0;
    float d10 = 0.f; d10 += dx1 * dx1; d10 += dy0 * dy0;
    float d01 = 0.f; d01 += dx0 * dx0; d01 += dy1 * dy1;
    float d11 = 0.f; d11 += dx1 * dx1; d11 += dy1 * dy1;
    return d00 < r2 || d10 < r2 || d01 < r2 || d11 < r2;
}

__global__ __launch_bounds__(LIGHT_BLOCK)
void k_light_grid(LightGridArgs a)
{
    const int lane = lane_id();
    float ax, ay, ar, bx, by, br;
    light_constants(a, (uint32_t)lane, ax, ay, ar);
    light_constants(a, (uint32_t)lane + 64u, bx, by, br);

    const uint32_t n_tiles = a.twidth * a.theight;
    const uint32_t wave = blockIdx.x * (LIGHT_BLOCK / WAVE) + threadIdx.x / WAVE;
    const uint32_t first = wave * LIGHT_TILES_PER_WAVE;
    uint4 mine = make_uint4(0, 0, 0, 0);
    for (int k = 0; k < LIGHT_TILES_PER_WAVE; k++) {
        const uint32_t tile = first + k;                             // wave-uniform
        if (tile >= n_tiles) break;
        const uint32_t gx = tile % a.twidth, gy = tile / a.twidth;
        // the tile's corner coordinates: unsigned arithmetic, then float (light.c:140-143)
        const float x0 = (float)(gx * a.cell), x1 = (float)(gx * a.cell + a.cell);
        const float y0 = (float)(gy * a.cell), y1 = (float)(gy * a.cell + a.cell);
        const uint64_t lo = __ballot(disc_reaches(ax, ay, ar, x0, x1, y0, y1));      // slots 0..63
        const uint64_t hi = __ballot(disc_reaches(bx, by, br, x0, x1, y0, y1));      // slots 64..127
        if (lane == k)
            mine = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
    }
    if (lane < LIGHT_TILES_PER_WAVE && first + lane < n_tiles)
        a.tiles[first + lane] = mine;
}

struct LightCarrierArgs {
    uint32_t        n_carriers, mode, n_entities, nr_lights;
    const uint32_t *carrier_entity;
    const int32_t  *carrier_light;
    const float    *carrier_off;
    const float4   *pos_scale;
    const int32_t  *parent;
    const uint32_t *flags;
    const uint32_t *active;
    float          *light_pos;
};

// Carriers apply in list order and the last applicable carrier of a slot wins (the reference's
// entity-order loop).  One lane per carrier decides whether it applies and records its index with an
// LDS atomicMax on its slot; then one lane per slot writes the winner's position.  (A lane per slot
// walking the whole list would be a chain of dependent loads: 16 us for 64 carriers.)
__global__ __launch_bounds__(CLAPGPU_LIGHTS_MAX)
void k_lights_from_entities(LightCarrierArgs a)
{
    __shared__ uint32_t last[CLAPGPU_LIGHTS_MAX];                    // 1 + index of the winning carrier, 0 = none
    const uint32_t t = threadIdx.x;
    last[t] = 0;
    __syncthreads();
    for (uint32_t k = t; k < a.n_carriers; k += CLAPGPU_LIGHTS_MAX) {
        const int32_t l = a.carrier_light[k];
        const uint32_t e = a.carrier_entity[k];
        if (l < 0 || (uint32_t)l >= a.nr_lights || e >= a.n_entities) continue;
        if (!a.active[l] || a.parent[e] >= 0) continue;
        if (!(a.mode & CLAPGPU_UPDATE_ALL_DIRTY) && !(a.flags[e] & CLAPGPU_E_DIRTY)) continue;
        atomicMax(&last[l], k + 1);
    }
    __syncthreads();
    if (t < a.nr_lights && last[t]) {
        const uint32_t k = last[t] - 1;
        const float4 ps = a.pos_scale[a.carrier_entity[k]];
        a.light_pos[3 * t] = ps.x + a.carrier_off[3 * k];
        a.light_pos[3 * t + 1] = ps.y + a.carrier_off[3 * k + 1];
        a.light_pos[3 * t + 2] = ps.z + a.carrier_off[3 * k + 2];
    }
}

} // namespace clapgpu

using namespace clapgpu;

// light.c:51-52
extern "C" void clapgpu_light_grid_dims(uint32_t width, uint32_t height, uint32_t cell, uint32_t *twidth,
                                        uint32_t *theight)
{
    if (twidth) *twidth = cell ? (uint32_t)ceilf((float)width / (float)cell) : 0;
    if (theight) *theight = cell ? (uint32_t)ceilf((float)height / (float)cell) : 0;
}

// host mat4x4_mul in linmath.h order (linmath.h:506-516); compiled without contraction like the device code
static void host_mat4_mul(float *out, const float *a, const float *b)
{
    float t[16];
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            float s = 0.f;
            for (int k = 0; k < 4; k++)
                s += a[4 * k + r] * b[4 * c + k];
            t[4 * c + r] = s;
        }
    memcpy(out, t, sizeof(t));
}

static int check_lights(const clapgpu_lights *l)
{
    if (!l || !l->pos || !l->color || !l->attenuation || !l->is_dir || !l->active)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (l->nr_lights > CLAPGPU_LIGHTS_MAX)
        return CLAPGPU_ERR_TOO_LARGE;
    return CLAPGPU_OK;
}

extern "C" int clapgpu_light_grid_compute(void *stream, const clapgpu_lights *lights, const float view_mx[16],
                                          const float proj_mx[16], uint32_t width, uint32_t height, uint32_t cell,
                                          uint32_t *tiles)
{
    int rc = check_lights(lights);
    if (rc) return rc;
    if (!view_mx || !proj_mx)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    uint32_t tw, th;
    clapgpu_light_grid_dims(width, height, cell, &tw, &th);
    if (!width || !height || !cell || !tw || !th)                   // light.c:46, 55, 93: nothing to do
        return CLAPGPU_OK;
    if (!tiles || ((uintptr_t)tiles & 15))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if ((uint64_t)tw * th > 0x7fffffffull || (uint64_t)tw * cell + cell > 0xffffffffull ||
        (uint64_t)th * cell + cell > 0xffffffffull)
        return CLAPGPU_ERR_TOO_LARGE;

    LightGridArgs a;
    a.nr_lights = lights->nr_lights;
    a.pos = lights->pos; a.color = lights->color; a.attenuation = lights->attenuation;
    a.is_dir = lights->is_dir; a.active = lights->active;
    memcpy(a.view, view_mx, sizeof(a.view));
    host_mat4_mul(a.mvp, proj_mx, view_mx);                         // light.c:98
    a.fx = proj_mx[0];
    a.width = width; a.height = height; a.cell = cell; a.twidth = tw; a.theight = th;
    a.tiles = reinterpret_cast<uint4 *>(tiles);
    const uint32_t n_tiles = tw * th;
    constexpr uint32_t per_block = LIGHT_TILES_PER_WAVE * (LIGHT_BLOCK / WAVE);
    hipLaunchKernelGGL(k_light_grid, dim3((n_tiles + per_block - 1) / per_block), dim3(LIGHT_BLOCK), 0,
                       as_stream(stream), a);
    CLAPGPU_LAUNCH_CHECK("k_light_grid");
    return CLAPGPU_OK;
}

extern "C" int clapgpu_lights_from_entities(void *stream, const clapgpu_entities *e, uint32_t mode,
                                            uint32_t n_carriers, const uint32_t *carrier_entity,
                                            const int32_t *carrier_light, const float *carrier_off,
                                            const clapgpu_lights *lights)
{
    int rc = check_lights(lights);
    if (rc) return rc;
    if (!e || !e->pos_scale || !e->parent || !e->flags)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if (n_carriers == 0 || lights->nr_lights == 0)
        return CLAPGPU_OK;
    if (!carrier_entity || !carrier_light || !carrier_off)
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    LightCarrierArgs a;
    a.n_carriers = n_carriers; a.mode = mode; a.n_entities = e->n; a.nr_lights = lights->nr_lights;
    a.carrier_entity = carrier_entity; a.carrier_light = carrier_light; a.carrier_off = carrier_off;
    a.pos_scale = reinterpret_cast<const float4 *>(e->pos_scale);
    a.parent = e->parent; a.flags = e->flags; a.active = lights->active; a.light_pos = lights->pos;
    hipLaunchKernelGGL(k_lights_from_entities, dim3(1), dim3(CLAPGPU_LIGHTS_MAX), 0, as_stream(stream), a);
    CLAPGPU_LAUNCH_CHECK("k_lights_from_entities");
    return CLAPGPU_OK;
}
