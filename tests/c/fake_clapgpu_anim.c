/*
 * fake_clapgpu_anim.c -- the rest of the CPU stand-in for libclapgpu.so (see fake_clapgpu.c): the entry points the
 * animation, particle and light bindings call (clap_amd/binding/gpu-anim.inc.c, gpu-particles.inc.c, gpu-light.inc.c),
 * computed by the oracle (oracle/pose.c, particles.c, light.c), so that `clap_dropin anim | particles | characters |
 * lights` link and run in a container without a GPU under -fsanitize=address,undefined and -fsanitize=thread.
 * TEST INFRASTRUCTURE ONLY: never shipped, proves nothing about the kernels (the -m gpu tests do that); what is under
 * test is the bindings' host code -- packing, staging, write-back over the worker pool.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "clapgpu.h"
#include "clap_oracle.h"

/* ---- skeletal animation ---------------------------------------------------------------------------------------------- */
size_t clapgpu_animations_packed_bytes(uint32_t n_anims, uint32_t max_keys, uint32_t nr_joints)
{
    (void)max_keys;
    return 64 + (size_t)n_anims * nr_joints * 16;              /* the fake keeps nothing in it but a tag */
}

int clapgpu_animations_pack(void *stream, const clapgpu_animations *an, uint32_t nr_joints, uint32_t max_keys, void *packed,
                            uint32_t *packed_layout)
{
    (void)stream;
    if (!an || !packed || !packed_layout || !an->chan_table || !an->times || !an->data || !nr_joints) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    /* what the real pack() checks on the host: key times strictly increasing, no channel longer than max_keys or past the pools */
    for (uint32_t a = 0; a < an->n_anims; a++)
        for (uint32_t j = 0; j < nr_joints; j++)
            for (uint32_t p = 0; p < 3; p++) {
                const uint32_t *c = an->chan_table + 4 * ((size_t)(a * nr_joints + j) * 3 + p);
                const uint32_t nr = c[2];
                if (!nr) continue;
                if (nr > max_keys || (an->n_times && c[0] + nr > an->n_times)) return CLAPGPU_ERR_INVALID_ARGUMENTS;
                if (an->n_data && c[1] + nr * (p == 1 ? 4u : 3u) > an->n_data) return CLAPGPU_ERR_INVALID_ARGUMENTS;
                for (uint32_t k = 1; k < nr; k++)
                    if (!(an->times[c[0] + k - 1] < an->times[c[0] + k])) return CLAPGPU_ERR_INVALID_ARGUMENTS;
            }
    memset(packed, 0, 64);
    *packed_layout = 0x7a000000u | (an->n_anims & 0xffffu);
    return CLAPGPU_OK;
}

int clapgpu_pose_update(void *stream, const clapgpu_skeleton *sk, const clapgpu_animations *an, const clapgpu_pose_batch *pb)
{
    (void)stream;
    if (!sk || !an || !pb || !sk->parent || !sk->depth || !sk->root_pose || !sk->invmx || !sk->bind || !an->chan_table ||
        !an->packed || (pb->n_chars && (!pb->anim || !pb->frame_time || !pb->joint_transforms)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    if ((an->packed_layout & 0xffff0000u) != 0x7a000000u || (an->packed_layout & 0xffffu) != (an->n_anims & 0xffffu))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const uint32_t J = sk->nr_joints;
    /* joints under joint 0, parents first (by depth, then index) */
    int32_t *order = malloc((size_t)(J ? J : 1) * sizeof(*order));
    if (!order) return CLAPGPU_ERR_NOMEM;
    uint32_t n_order = 0;
    for (uint32_t d = 0; d < sk->n_levels; d++)
        for (uint32_t j = 0; j < J; j++)
            if (sk->depth[j] == (int32_t)d) order[n_order++] = (int32_t)j;
    const clapo_skeleton os = { J, n_order, sk->parent, order, sk->root_pose, sk->invmx, sk->bind };
    /* one animation as the oracle takes it: the channels of chan_table[a] that exist */
    uint32_t *ct = malloc((size_t)J * 3 * 5 * sizeof(*ct) + 4);
    float *global = malloc((size_t)(J ? J : 1) * 16 * 4), *trs_tmp = malloc((size_t)(J ? J : 1) * 10 * 4);
    int32_t *cursor = malloc((size_t)(J ? J : 1) * 3 * 4);
    float *jp_tmp = malloc((size_t)(J ? J : 1) * 4 * 4);
    int rc = CLAPGPU_OK;
    if (!ct || !global || !trs_tmp || !cursor || !jp_tmp) rc = CLAPGPU_ERR_NOMEM;
    for (uint32_t c = 0; !rc && c < pb->n_chars; c++) {
        const uint32_t a = pb->anim[c];
        if (a >= an->n_anims) { rc = CLAPGPU_ERR_INVALID_ARGUMENTS; break; }
        uint32_t n_ch = 0, *tgt = ct, *path = ct + (size_t)J * 3, *nr = ct + (size_t)J * 6, *toff = ct + (size_t)J * 9, *doff = ct + (size_t)J * 12;
        for (uint32_t j = 0; j < J; j++)
            for (uint32_t p = 0; p < 3; p++) {
                const uint32_t *e = an->chan_table + 4 * ((size_t)(a * J + j) * 3 + p);
                if (!e[2]) continue;
                tgt[n_ch] = j; path[n_ch] = p; nr[n_ch] = e[2]; toff[n_ch] = e[0]; doff[n_ch] = e[1];
                n_ch++;
            }
        const clapo_animation oa = { n_ch, tgt, path, nr, toff, doff, an->times, an->data };
        float *trs = pb->trs ? pb->trs + (size_t)c * J * 10 : NULL;
        if (!trs) { rc = CLAPGPU_ERR_INVALID_ARGUMENTS; break; }
        float *use = trs;
        if (pb->skip & CLAPGPU_POSE_SKIP_TRS) { memcpy(trs_tmp, trs, (size_t)J * 40); use = trs_tmp; }   /* the blend stays "in registers" */
        memset(cursor, 0, (size_t)J * 12);                         /* channel_time_to_idx restarts from 0 when time < t[start]: hint only */
        clapo_pose_channels(&oa, pb->frame_time[c], use, cursor);
        const uint32_t ent = pb->entity ? pb->entity[c] : c;
        static const float ident[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1 };
        const float *emx = (pb->skip & CLAPGPU_POSE_JOINT_POS_MODEL) || !pb->entity_mx ? ident : pb->entity_mx + 16 * (size_t)ent;
        float *jp = (pb->joint_pos && !(pb->skip & CLAPGPU_POSE_SKIP_JOINT_POS)) ? pb->joint_pos + (size_t)c * J * 4 : jp_tmp;
        /* joints not under joint 0 are never written (the reference's recursion does not reach them): work on copies of their rows */
        clapo_pose_palette(&os, use, emx, global, pb->joint_transforms + (size_t)c * J * 16, jp);
    }
    free(order); free(ct); free(global); free(trs_tmp); free(cursor); free(jp_tmp);
    return rc;
}

/* ---- particles ---------------------------------------------------------------------------------------------------------- */
int clapgpu_particles_update(void *stream, const clapgpu_particles *p, const float view_mx[16])
{
    (void)stream;
    if (!p || !view_mx || (p->n_sys && (!p->sys || !p->pos || !p->vel || !p->rng_state))) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    uint64_t st = p->rng_state[1];                                 /* [1]: the stream's position before this call, and after it */
    clapo_particles_update((const clapo_particle_system *)p->sys, p->n_sys, p->pos, p->vel, &st);
    p->rng_state[0] = p->rng_state[1] = st;
    if (p->billboard_mx)
        for (uint32_t s = 0; s < p->n_sys; s++)
            clapo_particles_billboard(view_mx, p->sys[s].center, p->billboard_mx + 16 * (size_t)s);
    return CLAPGPU_OK;
}

/* ---- clustered-lighting tile masks ----------------------------------------------------------------------------------------- */
void clapgpu_light_grid_dims(uint32_t width, uint32_t height, uint32_t cell, uint32_t *twidth, uint32_t *theight)
{
    clapo_light_grid_dims(width, height, cell, twidth, theight);
}

int clapgpu_light_grid_compute(void *stream, const clapgpu_lights *l, const float view_mx[16], const float proj_mx[16],
                               uint32_t width, uint32_t height, uint32_t cell, uint32_t *tiles)
{
    (void)stream;
    if (!l || !view_mx || !proj_mx || !tiles || !cell || (l->nr_lights && (!l->pos || !l->color || !l->attenuation || !l->is_dir || !l->active)))
        return CLAPGPU_ERR_INVALID_ARGUMENTS;
    clapo_light_grid_compute(l->nr_lights, l->active, l->is_dir, l->pos, l->color, l->attenuation, view_mx, proj_mx, width, height, cell, tiles);
    return CLAPGPU_OK;
}
