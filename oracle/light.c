/*
 * oracle/light.c -- TEST INFRASTRUCTURE ONLY (see clap_oracle.h).
 *
 * Clustered-lighting tile masks: light_grid_compute (light.c:88-154) with light_grid_update's
 * tile counts (light.c:51-52), ui32vec4_set (light.c:82-89) and light_get_radius
 * (light.c:301-309), plus the entity -> light position hand-off inside default_update
 * (model.c:1689-1694).  Pinned against the reference: oracle/ref harness "lightgrid" runs the real
 * light_grid_compute and captures the RGBA32UI buffer it uploads.
 */
#include "clap_oracle.h"
#include "lm.h"

#define CLAPO_LIGHT_CUTOFF (1.0f / 256.0f)      /* shader_constants.h:15 */

/* light.c:301-309; max3 = max(a, max(b, c)) with max(a,b) = a > b ? a : b (util.h:200-203) */
float clapo_light_radius(const float color[3], const float att[3], int is_dir)
{
    if (is_dir) return 0.0f;
    float bc = color[1] > color[2] ? color[1] : color[2];
    float comp_max = color[0] > bc ? color[0] : bc;
    return (-att[1] + sqrtf(att[1] * att[1] - 4.0f * att[2] * (att[0] - comp_max / CLAPO_LIGHT_CUTOFF))) / (2.0f * att[2]);
}

/* light.c:51-52 */
void clapo_light_grid_dims(uint32_t width, uint32_t height, uint32_t cell, uint32_t *twidth, uint32_t *theight)
{
    *twidth = cell ? (uint32_t)ceilf((float)width / cell) : 0;
    *theight = cell ? (uint32_t)ceilf((float)height / cell) : 0;
}

/*
 * active[i] != 0 <=> slot i is set in light->active.  tiles: [theight][twidth][4] u32, the
 * RGBA32UI image handed to texture_load (light.c:150-153).
 */
void clapo_light_grid_compute(uint32_t nr_lights, const uint32_t *active, const int32_t *is_dir,
                              const float *pos, const float *color, const float *attenuation,
                              const float view_mx[16], const float proj_mx[16],
                              uint32_t width, uint32_t height, uint32_t cell, uint32_t *tiles)
{
    uint32_t twidth, theight;
    clapo_light_grid_dims(width, height, cell, &twidth, &theight);
    if (!width || !height || !cell || !twidth || !theight) return;

    memset(tiles, 0, (size_t)twidth * theight * 16);
    float mvp[16];
    lm_m4_mul(mvp, proj_mx, view_mx);

    for (uint32_t idx = 0; idx < nr_lights; idx++) {
        if (!active[idx]) continue;
        float radius = 0.f, rsq = 0.f, screen[2] = { 0.f, 0.f };
        if (!is_dir[idx]) {
            float lp[4] = { pos[3 * idx], pos[3 * idx + 1], pos[3 * idx + 2], 1.0f };
            float ndc[4], vp[4];
            lm_m4_mul_v4_post(vp, view_mx, lp);
            lm_m4_mul_v4_post(ndc, mvp, lp);
            float s = 1.0f / ndc[3];                                   /* vec3_scale: w itself stays */
            ndc[0] = ndc[0] * s; ndc[1] = ndc[1] * s; ndc[2] = ndc[2] * s;
            if (fabsf(ndc[3]) < 1e-3) continue;                        /* double compare */
            if (ndc[2] > 1.0) continue;
            float fx = proj_mx[0];
            radius = clapo_light_radius(color + 3 * idx, attenuation + 3 * idx, 0) * fx / -vp[2] * (width / 2.0f);
            rsq = radius * radius;
            screen[0] = (ndc[0] + 1.0f) / 2.0f * width;
            screen[1] = (1.0f - ndc[1]) / 2.0f * height;
        }
        for (uint32_t gy = 0; gy < theight; gy++)
            for (uint32_t gx = 0; gx < twidth; gx++) {
                uint32_t *v = tiles + 4 * ((size_t)gy * twidth + gx);
                int set = is_dir[idx] != 0;
                for (uint32_t corner = 0; corner < 4 && !set; corner++) {
                    float cx = (float)(gx * cell + cell * !!(corner & 1));   /* unsigned arithmetic, then float */
                    float cy = (float)(gy * cell + cell * !!(corner & 2));
                    float d0 = screen[0] - cx, d1 = screen[1] - cy;
                    float distsq = 0.f;
                    distsq += d0 * d0;
                    distsq += d1 * d1;
                    set = distsq < rsq;
                }
                if (set) v[idx / 32] |= 1u << (idx % 32);                   /* ui32vec4_set */
            }
    }
}

/*
 * default_update's light hand-off (model.c:1667-1694): an entity without a parent whose xform was
 * updated pushes pos + light_off to its light slot, if that slot is active (light_set_pos,
 * light.c:473-480).  Carriers are visited in entity-list order; `dirty` = transform_is_updated.
 */
void clapo_lights_from_entities(uint32_t n_carriers, const uint32_t *carrier_entity, const int32_t *carrier_light,
                                const float *carrier_off, const float *pos_scale, const int32_t *parent,
                                const uint8_t *dirty, uint32_t nr_lights, const uint32_t *active, float *light_pos)
{
    for (uint32_t k = 0; k < n_carriers; k++) {
        const uint32_t e = carrier_entity[k];
        const int32_t l = carrier_light[k];
        if (parent[e] >= 0 || !dirty[e]) continue;
        if (l < 0 || (uint32_t)l >= nr_lights || !active[l]) continue;
        for (int a = 0; a < 3; a++)
            light_pos[3 * l + a] = pos_scale[4 * (size_t)e + a] + carrier_off[3 * k + a];
    }
}
