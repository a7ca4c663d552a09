/*
 * oracle/lod.c -- TEST INFRASTRUCTURE ONLY (see clap_oracle.h).
 *
 * Per-pass LOD selection of _models_render (model.c:975-992) for the entities that passed the
 * cull, restated with entity3d_aabb_avg_edge (model.c:1261-1264), entity3d_set_lod /
 * model3d_validate_lod (model.c:593-609, 63-66) and aabb_point_is_inside (util.h:157-165).
 * The building blocks are pinned against the reference (oracle/ref harness "lod"); the five lines
 * of loop glue around them (dist, |dist|^2 - side^2, / 3600.0) have no callable reference form.
 */
#include "clap_oracle.h"
#include "lm.h"

/* entity3d_aabb_avg_edge: cbrtf(X * Y * Z), X = fabs(model dx) * scale (model.c:433-447,1185-1198) */
float clapo_aabb_avg_edge(const float model_aabb[6], float scale)
{
    float X = (float)fabs(model_aabb[3] - model_aabb[0]) * scale;
    float Y = (float)fabs(model_aabb[4] - model_aabb[1]) * scale;
    float Z = (float)fabs(model_aabb[5] - model_aabb[2]) * scale;
    return cbrtf(X * Y * Z);
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/*
 * visible[k] -> entity i.  model_lod[m] = (lod_min, lod_max).  cur_lod is entity3d.cur_lod
 * (in/out: unchanged while the camera is inside the box); draw_lod[k] = the LOD entity
 * visible[k] is drawn with.
 */
void clapo_entities_lod(uint32_t n_visible, const uint32_t *visible, const float cam_pos[3],
                        const float *aabb, const float *center, const float *pos_scale,
                        const int32_t *model, const float *model_aabb, const uint8_t *model_lod,
                        const int32_t *force_lod, int32_t *cur_lod, int32_t *draw_lod)
{
    for (uint32_t k = 0; k < n_visible; k++) {
        const uint32_t i = visible[k];
        const float *b = aabb + 6 * (size_t)i;
        if (force_lod[i] >= 0) {
            cur_lod[i] = force_lod[i];                                        /* model.c:976-977 */
        } else if (!(cam_pos[0] >= b[0] && cam_pos[0] <= b[3] && cam_pos[1] >= b[1] && cam_pos[1] <= b[4] &&
                     cam_pos[2] >= b[2] && cam_pos[2] <= b[5])) {             /* model.c:982 */
            const float *c = center + 3 * (size_t)i;
            float dist[3] = { c[0] - cam_pos[0], c[1] - cam_pos[1], c[2] - cam_pos[2] };
            float side = clapo_aabb_avg_edge(model_aabb + 6 * (size_t)model[i], pos_scale[4 * (size_t)i + 3]);
            float scale = fabsf(lm_dot3(dist, dist) - side * side) / 3600.0;  /* double divide, float store */
            const uint8_t *ml = model_lod + 2 * (size_t)model[i];
            cur_lod[i] = clampi((int)scale, ml[0], ml[1]);                    /* entity3d_set_lod(e, lod, false) */
        }
        draw_lod[k] = cur_lod[i];
    }
}
