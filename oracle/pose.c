/*
 * oracle/pose.c -- TEST INFRASTRUCTURE ONLY (see clap_oracle.h).
 *
 * Skeletal pose: keyframe bracket + lerp/slerp per channel, then the joint palette
 * (global chain, joint_transforms, joint world position), restated from the
 * reference's core/model.c:1266-1404 and core/interp.h.
 */
#include "clap_oracle.h"
#include "lm.h"

#define P_MAX(a, b) ((a) > (b) ? (a) : (b))
#define P_MIN(a, b) ((a) < (b) ? (a) : (b))

/* model.c:1266-1288 channel_time_to_idx */
static void time_to_idx(const float *t, int nr, float time, int start, int *prev, int *next)
{
    int i;

    if (time < t[0])
        goto tail;
    if (time < t[start])
        start = 0;
    for (i = start; i < nr && time > t[i]; i++)
        ;
    if (i == nr)
        goto tail;
    *prev = P_MAX(i - 1, 0);
    *next = P_MIN(*prev + 1, nr - 1);
    return;
tail:
    *prev = nr - 1;
    *next = 0;
}

/* model.c:1290-1350 channel_transform / channels_transform */
void clapo_pose_channels(const clapo_animation *an, float time, float *trs, int32_t *cursor)
{
    for (uint32_t c = 0; c < an->n_channels; c++) {
        const int nr = (int)an->ch_nr[c];
        const uint32_t path = an->ch_path[c], joint = an->ch_target[c];
        if (!nr || path > 2)
            continue;                                        /* model.c:1301, 1340 */
        const float *t = an->times + an->ch_time_off[c];
        const uint32_t stride = path == 1 ? 4 : 3;           /* floats per keyframe */
        const float *data = an->data + an->ch_data_off[c];
        float *j = trs + 10 * (size_t)joint;
        int32_t *off = cursor + 3 * (size_t)joint + path;
        int prev, next;
        float fac = 0;

        time_to_idx(t, nr, time, *off, &prev, &next);
        *off = P_MIN(prev, next);                            /* model.c:1310 */
        const float p_time = t[prev], n_time = t[next];
        if (p_time > n_time)                                 /* model.c:1314-1317 */
            fac = time < n_time ? 1 : 0;
        else if (p_time < n_time)
            fac = (time - p_time) / (n_time - p_time);
        const float *p = data + (size_t)prev * stride, *n = data + (size_t)next * stride;
        switch (path) {
        case 0: for (int k = 0; k < 3; k++) j[k] = lm_lerp(p[k], n[k], fac); break;       /* vec3_interp */
        case 1: lm_quat_slerp(j + 3, p, n, fac); break;
        case 2: for (int k = 0; k < 3; k++) j[7 + k] = lm_lerp(p[k], n[k], fac); break;
        }
    }
}

/* model.c:1352-1404 one_joint_transform, iterated in a parents-first order of the joints
 * reachable from joint 0 (the reference recurses depth-first from joint 0) */
void clapo_pose_palette(const clapo_skeleton *sk, const float *trs, const float *entity_mx,
                        float *global, float *joint_transforms, float *joint_pos)
{
    for (uint32_t o = 0; o < sk->n_order; o++) {
        const uint32_t j = (uint32_t)sk->order[o];
        const int32_t parent = sk->parent[j];
        const float *tr = trs + 10 * (size_t)j;
        float *jt = global + 16 * (size_t)j;
        float T[16], R[16], trsm[16], mpos[4];
        const float origin[4] = { 0.0f, 0.0f, 0.0f, 1.0f };

        lm_m4_identity(jt);
        lm_m4_mul(jt, parent >= 0 ? global + 16 * (size_t)parent : sk->root_pose, jt);    /* model.c:1366-1371 */
        lm_m4_translate(T, tr[0], tr[1], tr[2]);
        lm_m4_mul(jt, jt, T);
        lm_m4_from_quat(R, tr + 3);
        lm_m4_mul(jt, jt, R);
        lm_m4_scale_aniso(jt, jt, tr[7], tr[8], tr[9]);

        float *JT = joint_transforms + 16 * (size_t)j;
        lm_m4_mul(JT, jt, sk->invmx + 16 * (size_t)j);                                     /* model.c:1389 */
        lm_m4_mul(trsm, JT, sk->bind + 16 * (size_t)j);
        lm_m4_mul_v4_post(mpos, trsm, origin);
        lm_m4_mul_v4_post(joint_pos + 4 * (size_t)j, entity_mx, mpos);                     /* model.c:1400 */
    }
}

/* model3d_add_skinning: bind = invert(invmx) (model.c:531-532) */
void clapo_skeleton_bind(uint32_t nr_joints, const float *invmx, float *bind)
{
    for (uint32_t j = 0; j < nr_joints; j++)
        lm_m4_invert(bind + 16 * (size_t)j, invmx + 16 * (size_t)j);
}


/*
 * The clock of animated_update (model.c:1563-1592) for a batch: frame_time in double from
 * now / ani_time / speed, handed on as float (channels_transform takes a float); ended = the test of
 * model.c:1590; a repeating entry restarts through animation_next -> animation_start
 * (model.c:1455-1483, 1406-1424: ani_time = now).
 */
void clapo_animation_time(uint32_t n_chars, uint32_t n_anims, const uint32_t *anim, const float *time_end,
                          double *ani_time, const float *speed, const uint8_t *restart, double now,
                          float *frame_time, uint8_t *ended)
{
    for (uint32_t c = 0; c < n_chars; c++) {
        double ft = (now - ani_time[c]) * speed[c];
        frame_time[c] = ft;
        ended[c] = anim[c] < n_anims && ft >= time_end[anim[c]];
        if (ended[c] && restart[c])
            ani_time[c] = now;
    }
}
