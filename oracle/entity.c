/*
 * oracle/entity.c -- TEST INFRASTRUCTURE ONLY (see clap_oracle.h).
 *
 * Entity transform hierarchy -> inverse -> world AABB -> frustum cull,
 * restated from the reference's core/model.c, core/view.c, core/transform.c.
 */
#ifdef _OPENMP
#include <omp.h>
#endif
#include "clap_oracle.h"
#include "lm.h"

/* transform.c:132-138: I, *= R(quat), transpose 3x3, translate_in_place(-pos) */
void clapo_view_matrix(const float pos[3], const float quat[4], float view_mx[16])
{
    float r[16];

    lm_m4_identity(view_mx);
    lm_m4_from_quat(r, quat);                 /* transform.c:125-130 */
    lm_m4_mul(view_mx, view_mx, r);
    lm_m4_transpose_3x3(view_mx);
    lm_m4_translate_in_place(view_mx, -pos[0], -pos[1], -pos[2]);
}

void clapo_perspective(float fov, float aspect, float near_plane, float far_plane,
                       int ndc_z_zero_one, float proj_mx[16])
{
    lm_m4_perspective(proj_mx, fov, aspect, near_plane, far_plane, ndc_z_zero_one);
}

/* view.c:248-289 */
void clapo_frustum_calc(const float view_mx[16], const float proj_mx[16],
                        int ndc_z_zero_one, clapo_frustum *out)
{
    float mvp[16], trans[16], invmvp[16];
    const float zn = ndc_z_zero_one ? 0.f : -1.f;      /* view.c:252-265 corner tables */
    const float ndc[8][4] = {
        { -1, -1, zn, 1 }, { 1, -1, zn, 1 }, { 1, 1, zn, 1 }, { -1, 1, zn, 1 },
        { -1, -1,  1, 1 }, { 1, -1,  1, 1 }, { 1, 1,  1, 1 }, { -1, 1,  1, 1 },
    };

    lm_m4_mul(mvp, proj_mx, view_mx);                   /* view.c:270 */
    lm_m4_transpose(trans, mvp);
    lm_m4_invert(invmvp, mvp);

    /* view.c:275-280: trans[3] +/- trans[0|1|2]; trans[c] is column c of the transpose */
    for (int k = 0; k < 4; k++) {
        out->planes[0][k] = trans[12 + k] + trans[0 + k];
        out->planes[1][k] = trans[12 + k] - trans[0 + k];
        out->planes[2][k] = trans[12 + k] + trans[4 + k];
        out->planes[3][k] = trans[12 + k] - trans[4 + k];
        out->planes[4][k] = trans[12 + k] + trans[8 + k];
        out->planes[5][k] = trans[12 + k] - trans[8 + k];
    }

    /* view.c:283-288 */
    for (int i = 0; i < 8; i++) {
        float q[4];
        lm_m4_mul_v4_post(q, invmvp, ndc[i]);
        float s = 1.f / q[3];
        for (int k = 0; k < 4; k++)
            out->corners[i][k] = q[k] * s;
    }
}

void clapo_trs_matrix(const float pos_scale[4], const float rot[4], float mx[16])
{
    float r[16];

    lm_m4_identity(mx);                                            /* model.c:1619/1670 */
    lm_m4_translate_in_place(mx, pos_scale[0], pos_scale[1], pos_scale[2]);   /* transform.c:57-60 */
    lm_m4_from_quat(r, rot);                                       /* transform.c:125-130 */
    lm_m4_mul(mx, mx, r);
    lm_m4_scale_aniso(mx, mx, pos_scale[3], pos_scale[3], pos_scale[3]);      /* model.c:1622/1675 */
}

void clapo_mat4_mul(float out[16], const float a[16], const float b[16])
{
    lm_m4_mul(out, a, b);
}

void clapo_mat4_invert(float out[16], const float m[16])
{
    lm_m4_invert(out, m);
}

/* util.h:188-201: min/max are plain ternaries (NaN-order sensitive, kept literal) */
#define O_MIN(a, b) ((a) < (b) ? (a) : (b))
#define O_MAX(a, b) ((a) > (b) ? (a) : (b))

/* model.c:1200-1234 + util.h:104-109 */
void clapo_aabb_update(const float mx[16], const float a[6], float aabb[6], float center[3])
{
    /* corner order of model.c:1207-1216: x-major, then z, then y toggling fastest */
    const float corners[8][4] = {
        { a[0], a[1], a[2], 1.0f }, { a[0], a[4], a[2], 1.0f },
        { a[0], a[1], a[5], 1.0f }, { a[0], a[4], a[5], 1.0f },
        { a[3], a[1], a[2], 1.0f }, { a[3], a[4], a[2], 1.0f },
        { a[3], a[1], a[5], 1.0f }, { a[3], a[4], a[5], 1.0f },
    };

    aabb[0] = aabb[1] = aabb[2] = INFINITY;
    aabb[3] = aabb[4] = aabb[5] = -INFINITY;
    for (int i = 0; i < 8; i++) {
        float v[4];
        lm_m4_mul_v4_post(v, mx, corners[i]);
        for (int k = 0; k < 3; k++) {
            aabb[k]     = O_MIN(v[k], aabb[k]);
            aabb[3 + k] = O_MAX(v[k], aabb[3 + k]);
        }
    }
    /* aabb_center: (max - min) * 0.5 + min */
    for (int k = 0; k < 3; k++) {
        float d = aabb[3 + k] - aabb[k];
        d = d * 0.5f;
        center[k] = d + aabb[k];
    }
}

/* view.c:296-337 */
int clapo_aabb_in_frustum(const clapo_frustum *f, const float aabb[6])
{
    const float *mn = aabb, *mx = aabb + 3;

    for (int i = 0; i < 6; i++) {
        int r = 0;
        /* corner order of view.c:308-323: x toggles fastest, then y, then z */
        for (int k = 0; k < 8; k++) {
            float v[4] = { (k & 1) ? mx[0] : mn[0], (k & 2) ? mx[1] : mn[1],
                           (k & 4) ? mx[2] : mn[2], 1.0f };
            r += (lm_dot4(f->planes[i], v) < 0.0) ? 1 : 0;
        }
        if (r == 8)
            return 0;
    }

    for (int ax = 0; ax < 3; ax++) {
        int r = 0;
        for (int i = 0; i < 8; i++) r += f->corners[i][ax] > mx[ax] ? 1 : 0;
        if (r == 8) return 0;
        r = 0;
        for (int i = 0; i < 8; i++) r += f->corners[i][ax] < mn[ax] ? 1 : 0;
        if (r == 8) return 0;
    }
    return 1;
}

static const clapo_attach *find_attach(uint32_t n_attach, const clapo_attach *attach, uint32_t entity)
{
    for (uint32_t k = 0; k < n_attach; k++)
        if (attach[k].entity == entity)
            return &attach[k];
    return NULL;
}

uint32_t clapo_entities_update_range(uint32_t first, uint32_t count,
                                     const float *pos_scale, const float *rot,
                                     const int32_t *parent, const int32_t *model,
                                     const float *model_aabb, const uint8_t *model_skip_aabb,
                                     uint32_t *flags, uint32_t *seqs,
                                     float *mx, float *inv_mx, float *aabb, float *center,
                                     uint32_t n_attach, const clapo_attach *attach,
                                     const float *jt_pool, const float *bind_pool)
{
    uint32_t rebuilt = 0;

    for (uint32_t i = first; i < first + count; i++) {
        if (!(flags[i] & CLAPO_E_ALIVE))          /* mq_update: model.c:1955 */
            continue;

        int dirty = !!(flags[i] & CLAPO_E_DIRTY);
        uint16_t seq = (uint16_t)(seqs[i] & 0xffff);
        uint16_t pseq = (uint16_t)(seqs[i] >> 16);
        int32_t p = parent[i];
        float *m = mx + 16 * (size_t)i;
        const clapo_attach *at = (p >= 0 && (flags[i] & CLAPO_E_JOINT_ATTACHED))
                                 ? find_attach(n_attach, attach, i) : NULL;

        if (p >= 0) {
            /* parent_transform_apply: model.c:1609-1641; joint attachments never skip */
            uint16_t parent_seq_now = (uint16_t)(seqs[p] & 0xffff);
            if (!at && pseq == parent_seq_now && !dirty)
                continue;
            pseq = parent_seq_now;
            seq++;
            float local[16];
            clapo_trs_matrix(pos_scale + 4 * (size_t)i, rot + 4 * (size_t)i, local);
            if (at) {
                float joint_mx[16];                                   /* model.c:1636-1640 */
                lm_m4_mul(joint_mx, jt_pool + 16 * (size_t)at->jt, bind_pool + 16 * (size_t)at->bind);
                lm_m4_mul(m, joint_mx, local);
                lm_m4_mul(m, mx + 16 * (size_t)p, m);
            } else {
                lm_m4_mul(m, mx + 16 * (size_t)p, local);             /* model.c:1625 */
            }
        } else {
            if (!dirty)                             /* model.c:1667 */
                continue;
            seq++;
            clapo_trs_matrix(pos_scale + 4 * (size_t)i, rot + 4 * (size_t)i, m);
        }
        flags[i] &= ~CLAPO_E_DIRTY;
        seqs[i] = (uint32_t)seq | ((uint32_t)pseq << 16);

        lm_m4_invert(inv_mx + 16 * (size_t)i, m);  /* model.c:1643/1676 */
        if (!model_skip_aabb[model[i]])             /* model.c:1204 */
            clapo_aabb_update(m, model_aabb + 6 * (size_t)model[i],
                              aabb + 6 * (size_t)i, center + 3 * (size_t)i);
        rebuilt++;
    }
    return rebuilt;
}

uint32_t clapo_entities_update(uint32_t n,
                               const float *pos_scale, const float *rot,
                               const int32_t *parent, const int32_t *model,
                               const float *model_aabb, const uint8_t *model_skip_aabb,
                               uint32_t *flags, uint32_t *seqs,
                               float *mx, float *inv_mx, float *aabb, float *center)
{
    return clapo_entities_update_range(0, n, pos_scale, rot, parent, model, model_aabb, model_skip_aabb,
                                       flags, seqs, mx, inv_mx, aabb, center, 0, NULL, NULL, NULL);
}

/* model.c:1703-1713 + util.h:157-165 + model.c:433-447,1185-1198 */
int32_t clapo_camera_bv(uint32_t n, const uint32_t *flags, const float *aabb, const float *pos_scale,
                        const int32_t *model, const float *model_aabb,
                        const float cam_pos[3], const float *ctl_pos, int32_t ctl_entity, float *volume)
{
    int32_t bv = -1;
    float bv_volume = 0.f;
    for (uint32_t i = 0; i < n; i++) {
        if (!(flags[i] & CLAPO_E_ALIVE))
            continue;
        const float *b = aabb + 6 * (size_t)i;
        int inside = cam_pos[0] >= b[0] && cam_pos[0] <= b[3] && cam_pos[1] >= b[1] && cam_pos[1] <= b[4] &&
                     cam_pos[2] >= b[2] && cam_pos[2] <= b[5];
        if (!inside && ctl_pos)
            inside = ctl_pos[0] >= b[0] && ctl_pos[0] <= b[3] && ctl_pos[1] >= b[1] && ctl_pos[1] <= b[4] &&
                     ctl_pos[2] >= b[2] && ctl_pos[2] <= b[5];
        if (!inside || (int32_t)i == ctl_entity)
            continue;
        const float *ma = model_aabb + 6 * (size_t)model[i];
        const float s = pos_scale[4 * (size_t)i + 3];
        float X = (float)fabs(ma[3] - ma[0]) * s, Y = (float)fabs(ma[4] - ma[1]) * s, Z = (float)fabs(ma[5] - ma[2]) * s;
        float vol = X * Y * Z;
        if (bv < 0 || vol > bv_volume) {
            bv = (int32_t)i;
            bv_volume = vol;
        }
    }
    if (volume) *volume = bv_volume;
    return bv;
}

uint32_t clapo_entities_cull(uint32_t n, const uint32_t *flags, const float *aabb,
                             const clapo_frustum *f, uint32_t *visible, uint64_t *vis_mask)
{
    uint32_t count = 0;

    if (vis_mask)
        memset(vis_mask, 0, ((size_t)n + 63) / 64 * sizeof(uint64_t));
    for (uint32_t i = 0; i < n; i++) {
        uint32_t fl = flags[i];
        if (!(fl & CLAPO_E_ALIVE) || !(fl & CLAPO_E_VISIBLE))       /* model.c:959-965 */
            continue;
        if (!(fl & CLAPO_E_SKIP_CULLING) &&
            !clapo_aabb_in_frustum(f, aabb + 6 * (size_t)i))        /* model.c:967-971 */
            continue;
        if (visible)
            visible[count] = i;
        if (vis_mask)
            vis_mask[i >> 6] |= 1ull << (i & 63);
        count++;
    }
    return count;
}


/*
 * The same frame on all host cores, for context next to the single-threaded figure: the
 * reference's frame loop is single-threaded, but whole subtrees are independent, so the tile layout
 * (tile = contiguous range holding whole subtrees, parents first) parallelises over tiles, and the
 * frustum test over 64-entity words.  OpenMP; same arithmetic as clapo_entities_update / _cull.
 */
/* bench.py caps the team at the container's CPU quota: libgomp read OMP_NUM_THREADS long before (torch loaded it) */
void clapo_omp_set_threads(uint32_t n)
{
#ifdef _OPENMP
    if (n) omp_set_num_threads((int)n);
#else
    (void)n;
#endif
}

/* threads the OpenMP figure below really runs on (bench.py reports it beside the figure) */
uint32_t clapo_omp_max_threads(void)
{
#ifdef _OPENMP
    return (uint32_t)omp_get_max_threads();
#else
    return 1u;
#endif
}

uint32_t clapo_entities_frame_tiles_mt(uint32_t n_tiles, const uint32_t *tile_row_start, uint32_t n,
                                       const float *pos_scale, const float *rot,
                                       const int32_t *parent, const int32_t *model,
                                       const float *model_aabb, const uint8_t *model_skip_aabb,
                                       uint32_t *flags, uint32_t *seqs,
                                       float *mx, float *inv_mx, float *aabb, float *center,
                                       const clapo_frustum *f, uint64_t *vis_mask)
{
    const uint32_t n_rows = (n + 63) / 64;
#pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t t = 0; t < n_tiles; t++) {
        uint32_t r0 = tile_row_start[t], r1 = tile_row_start[t + 1];
        if (r1 > n_rows) r1 = n_rows;
        if (r0 >= r1) continue;
        uint32_t first = r0 * 64, last = r1 * 64 < n ? r1 * 64 : n;
        clapo_entities_update_range(first, last - first, pos_scale, rot, parent, model, model_aabb, model_skip_aabb,
                                    flags, seqs, mx, inv_mx, aabb, center, 0, NULL, NULL, NULL);
    }
    uint32_t count = 0;
#pragma omp parallel for schedule(static) reduction(+ : count)
    for (uint32_t w = 0; w < n_rows; w++) {
        uint64_t m = 0;
        uint32_t end = (w + 1) * 64 < n ? (w + 1) * 64 : n;
        for (uint32_t i = w * 64; i < end; i++) {
            uint32_t fl = flags[i];
            if (!(fl & CLAPO_E_ALIVE) || !(fl & CLAPO_E_VISIBLE)) continue;
            if (!(fl & CLAPO_E_SKIP_CULLING) && !clapo_aabb_in_frustum(f, aabb + 6 * (size_t)i)) continue;
            m |= 1ull << (i & 63);
            count++;
        }
        vis_mask[w] = m;
    }
    return count;
}
