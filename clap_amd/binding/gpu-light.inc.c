/*
 * gpu-light.inc.c -- CLAP-side binding of libclapgpu for the clustered-lighting tile masks.
 *
 * light_grid_update() is private to core/light.c, so this file is meant to be #include'd at the end of
 * that translation unit (the drop-in checker oracle/ref/dropin.c includes it the same way).
 *
 *   gpu_light_grid_compute(gl, light, view)    the body of light_grid_compute() (light.c:88-154):
 *       the reference's own light_grid_update() (tile counts, host array, texture resize), the touched
 *       slot arrays of `struct light` (5 KB) up, lights x tiles x corners on the device, the RGBA32UI masks
 *       down into light->grid.tiles, and the same texture_load() the reference ends with.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "clapgpu.h"

struct gpu_lights {
    clapgpu_lights  d;
    void            *d_slots;               /* pos | color | attenuation | is_dir | active, one allocation */
    char            *h_slots;               /* page-locked image of the same */
    uint32_t        *d_tiles;
    ui32vec4        *h_tiles;               /* page-locked */
    size_t          cap_tiles;
};

#define GL_SLOT_BYTES (CLAPGPU_LIGHTS_MAX * (3 * 12 + 4 + 4))
#define GL_CK(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

int gpu_lights_init(struct gpu_lights **out, int device)
{
    _Static_assert(CLAPGPU_LIGHTS_MAX == LIGHTS_MAX, "slot count");
    if (!out) return _CERR_INVALID_ARGUMENTS;
    int rc = clapgpu_init(device);
    if (rc) return rc;
    struct gpu_lights *gl = calloc(1, sizeof(*gl));
    if (!gl) return _CERR_NOMEM;
    GL_CK(clapgpu_malloc(&gl->d_slots, GL_SLOT_BYTES));
    GL_CK(clapgpu_host_malloc((void **)&gl->h_slots, GL_SLOT_BYTES));
    char *d = gl->d_slots;
    gl->d.pos = (float *)d;
    gl->d.color = (const float *)(d + CLAPGPU_LIGHTS_MAX * 12);
    gl->d.attenuation = (const float *)(d + CLAPGPU_LIGHTS_MAX * 24);
    gl->d.is_dir = (const int32_t *)(d + CLAPGPU_LIGHTS_MAX * 36);
    gl->d.active = (const uint32_t *)(d + CLAPGPU_LIGHTS_MAX * 40);
    *out = gl;
    return 0;
}

void gpu_lights_done(struct gpu_lights *gl)
{
    if (!gl) return;
    if (gl->d_slots) clapgpu_free(gl->d_slots);
    if (gl->h_slots) clapgpu_host_free(gl->h_slots);
    if (gl->d_tiles) clapgpu_free(gl->d_tiles);
    if (gl->h_tiles) clapgpu_host_free(gl->h_tiles);
    free(gl);
}

int gpu_light_grid_compute(struct gpu_lights *gl, struct light *light, struct view *view)
{
    if (!gl || !light || !view) return _CERR_INVALID_ARGUMENTS;
    light_grid_update(light);                                        /* light.c:90, unchanged */

    auto grid = &light->grid;
    if (!grid->twidth || !grid->theight || !grid->tiles)    return 0;
    const size_t ntiles = (size_t)grid->twidth * grid->theight;
    if (ntiles > gl->cap_tiles) {
        if (gl->d_tiles) clapgpu_free(gl->d_tiles);
        if (gl->h_tiles) clapgpu_host_free(gl->h_tiles);
        gl->d_tiles = NULL; gl->h_tiles = NULL;
        GL_CK(clapgpu_malloc((void **)&gl->d_tiles, ntiles * sizeof(ui32vec4)));
        GL_CK(clapgpu_host_malloc((void **)&gl->h_tiles, ntiles * sizeof(ui32vec4)));
        gl->cap_tiles = ntiles;
    }

    /* the slots light_grid_compute reads (light.h:19-27, 35, 49) */
    char *h = gl->h_slots;
    memcpy(h, light->pos, CLAPGPU_LIGHTS_MAX * 12);
    memcpy(h + CLAPGPU_LIGHTS_MAX * 12, light->color, CLAPGPU_LIGHTS_MAX * 12);
    memcpy(h + CLAPGPU_LIGHTS_MAX * 24, light->attenuation, CLAPGPU_LIGHTS_MAX * 12);
    memcpy(h + CLAPGPU_LIGHTS_MAX * 36, light->is_dir, CLAPGPU_LIGHTS_MAX * 4);
    uint32_t *act = (uint32_t *)(h + CLAPGPU_LIGHTS_MAX * 40);
    for (int i = 0; i < CLAPGPU_LIGHTS_MAX; i++)
        act[i] = i < light->nr_lights && bitmap_is_set(&light->active, i);
    gl->d.nr_lights = (uint32_t)light->nr_lights;
    GL_CK(clapgpu_memcpy_h2d(gl->d_slots, h, GL_SLOT_BYTES, NULL));

    GL_CK(clapgpu_light_grid_compute(NULL, &gl->d, (const float *)view->main.view_mx, (const float *)view->main.proj_mx,
                                     grid->width, grid->height, grid->cell, gl->d_tiles));
    GL_CK(clapgpu_memcpy_d2h(gl->h_tiles, gl->d_tiles, ntiles * sizeof(ui32vec4), NULL));
    GL_CK(clapgpu_stream_sync(NULL));
    memcpy(grid->tiles, gl->h_tiles, ntiles * sizeof(ui32vec4));

    CERR_RET(                                                        /* light.c:150-153, unchanged */
        texture_load(&grid->tex, TEX_FMT_RGBA32UI, grid->twidth, grid->theight, grid->tiles),
        err_cerr(__cerr, "grid texture (%u x %u) load failed\n", grid->twidth, grid->theight);
    );
    return 0;
}
