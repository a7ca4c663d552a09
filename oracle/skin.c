/*
 * oracle/skin.c -- TEST INFRASTRUCTURE ONLY (see clap_oracle.h).
 *
 * Vertex skinning.  PARITY UNPINNED: the reference has no CPU implementation and no test of
 * this step -- it exists only as GLSL, shaders/model.vert:32-48 (same loop in shadow.vert:19-23
 * and shadow_vsm.vert:19-23).  This file follows that shader literally:
 *
 *     total_local_pos += (joint_transforms[joints[i]] * vec4(position, 1.0)) * weights[i];
 *     total_normal    += (joint_transforms[joints[i]] * vec4(normal,   0.0)) * weights[i];
 *
 * for i = 0..3 in order, fp32, no weight renormalisation.  GLSL leaves the evaluation order of
 * mat4 * vec4 to the implementation; it is fixed here as ((M0 x + M1 y) + M2 z) + M3 w per
 * component, unfused, and the HIP kernel is held to 1e-5 relative of that.
 */
#include "clap_oracle.h"
#include "lm.h"

void clapo_skin(uint32_t n_verts, const float *position, const float *normal,
                const uint8_t *joints, const float *weights,
                const float *joint_transforms, float *out_pos, float *out_nor)
{
    clapo_skin_w(n_verts, position, normal, joints, weights, joint_transforms, out_pos, out_nor, 0);
}

/* The same with total_local_pos.w (model.vert:36-38,44: the shader carries the vec4 on into proj * view * trs), the
 * fourth component of the very loop above; out_w may be NULL. */
void clapo_skin_w(uint32_t n_verts, const float *position, const float *normal,
                  const uint8_t *joints, const float *weights,
                  const float *joint_transforms, float *out_pos, float *out_nor, float *out_w)
{
    for (uint32_t v = 0; v < n_verts; v++) {
        const float p[4] = { position[3 * (size_t)v], position[3 * (size_t)v + 1], position[3 * (size_t)v + 2], 1.0f };
        const float n[4] = { normal[3 * (size_t)v], normal[3 * (size_t)v + 1], normal[3 * (size_t)v + 2], 0.0f };
        float tp[4] = { 0, 0, 0, 0 }, tn[4] = { 0, 0, 0, 0 };

        for (int i = 0; i < 4; i++) {
            const float *J = joint_transforms + 16 * (size_t)joints[4 * (size_t)v + i];
            const float w = weights[4 * (size_t)v + i];
            for (int r = 0; r < 4; r++) {
                float lp = ((LM_E(J, 0, r) * p[0] + LM_E(J, 1, r) * p[1]) + LM_E(J, 2, r) * p[2]) + LM_E(J, 3, r) * p[3];
                float ln = ((LM_E(J, 0, r) * n[0] + LM_E(J, 1, r) * n[1]) + LM_E(J, 2, r) * n[2]) + LM_E(J, 3, r) * n[3];
                tp[r] += lp * w;
                tn[r] += ln * w;
            }
        }
        for (int r = 0; r < 3; r++) {
            out_pos[3 * (size_t)v + r] = tp[r];
            out_nor[3 * (size_t)v + r] = tn[r];
        }
        if (out_w)
            out_w[v] = tp[3];
    }
}
