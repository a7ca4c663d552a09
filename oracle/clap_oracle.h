/*
 * oracle/clap_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference (virtuoso/clap) per-frame scene-update hot
 * path over the same SoA arrays the HIP kernels use.  It exists to CHECK the
 * GPU path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * Nothing under clap_amd/ may include, link or call it.
 *
 * Pinning status (details: oracle/README.md, DESIGN.md "Oracle"):
 *   entity transform / inverse / AABB / frustum / cull ... pinned bit-exact against
 *       the reference's own core/model.c, core/view.c, core/transform.c compiled
 *       from /root/reference (oracle/ref -> oracle/_ref) and against the golden
 *       vectors in tests/golden produced by that build.
 *   pose (channels, slerp), joint palette ................. pinned <=1e-5 (same way)
 *   particles ............................................. pinned bit-exact (same way)
 *   vertex skinning ....................................... PARITY UNPINNED: the reference has
 *       only GLSL (shaders/model.vert:32-48), no CPU implementation and no test.
 *   rigid-body integrate + broadphase ..................... PARITY UNPINNED: ODE is an absent
 *       submodule (deps/ode, .gitmodules:1-3); restated from ODE's published algorithm.
 *
 * Layout conventions (identical to include/clapgpu.h):
 *   mat4   = float[16], column-major, (col c,row r) at [4c+r]   (linmath.h `M[c][r]`)
 *   quat   = (x, y, z, w)                                       (linmath.h:835-840)
 *   pos_scale[i] = (pos.x, pos.y, pos.z, entity scale)          (transform.h:8-12, model.h:412)
 *   aabb[i]      = (min.x,min.y,min.z, max.x,max.y,max.z)       (model.h `vec3 aabb[2]`)
 */
#ifndef CLAP_ORACLE_H
#define CLAP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* entity3d_flags bits the path reads (model.h:293-312) + one shim bit */
#define CLAPO_E_VISIBLE       (1u << 0)
#define CLAPO_E_SKIP_CULLING  (1u << 14)
#define CLAPO_E_DIRTY         (1u << 16)   /* mirror of transform_t.updated (transform.h:11) */
#define CLAPO_E_JOINT_ATTACHED (1u << 17) /* e->parent_joint != JOINT_TYPE_MAX (model.h:386-403) */
#define CLAPO_E_ALIVE         (1u << 31)

typedef struct clapo_frustum {
    float planes[6][4];     /* view.h:16  frustum_planes  */
    float corners[8][4];    /* view.h:17  frustum_corners */
} clapo_frustum;

/* ---- view / frustum (host-side O(1) per frame) ---- */
/* transform.c:132-138 transform_view_mat4x4 */
void clapo_view_matrix(const float pos[3], const float quat[4], float view_mx[16]);
/* linmath.h:709-776 via render-common.c:70-82 */
void clapo_perspective(float fov, float aspect, float near_plane, float far_plane,
                       int ndc_z_zero_one, float proj_mx[16]);
/* view.c:248-289 subview_calc_frustum */
void clapo_frustum_calc(const float view_mx[16], const float proj_mx[16],
                        int ndc_z_zero_one, clapo_frustum *out);

/* ---- single-entity pieces (used by unit tests against golden vectors) ---- */
/* model.c:1618-1622 / 1670-1675: I -> translate_in_place -> *R -> scale_aniso */
void clapo_trs_matrix(const float pos_scale[4], const float rot[4], float mx[16]);
void clapo_mat4_mul(float out[16], const float a[16], const float b[16]);
void clapo_mat4_invert(float out[16], const float m[16]);
/* model.c:1200-1234 entity3d_aabb_update */
void clapo_aabb_update(const float mx[16], const float model_aabb[6], float aabb[6], float center[3]);
/* view.c:296-337 view_entity_in_frustum */
int clapo_aabb_in_frustum(const clapo_frustum *f, const float aabb[6]);

/*
 * mq_update over default_update entities, converged level order
 * (model.c:1953,1649-1695,1594-1647).  Requires parent[i] < i or parent[i] < 0.
 * seqs[i] = seq | parent_seq << 16 (model.h:404-405, uint16 wrap).
 * Returns the number of entities whose matrices were rebuilt.
 */
uint32_t clapo_entities_update(uint32_t n,
                               const float *pos_scale, const float *rot,
                               const int32_t *parent, const int32_t *model,
                               const float *model_aabb, const uint8_t *model_skip_aabb,
                               uint32_t *flags, uint32_t *seqs,
                               float *mx, float *inv_mx, float *aabb, float *center);

/* joint attachment of entity `entity` (flag CLAPO_E_JOINT_ATTACHED): it rides joint matrix
 * jt_pool[jt] * bind_pool[bind] of its parent (model.c:1626-1641) and is rebuilt every frame */
typedef struct clapo_attach {
    uint32_t entity, jt, bind, pad;
} clapo_attach;

/* clapo_entities_update restricted to [first, first+count) and aware of joint attachments
 * (attach sorted by entity; jt_pool / bind_pool are arrays of mat4) */
uint32_t clapo_entities_update_range(uint32_t first, uint32_t count,
                                     const float *pos_scale, const float *rot,
                                     const int32_t *parent, const int32_t *model,
                                     const float *model_aabb, const uint8_t *model_skip_aabb,
                                     uint32_t *flags, uint32_t *seqs,
                                     float *mx, float *inv_mx, float *aabb, float *center,
                                     uint32_t n_attach, const clapo_attach *attach,
                                     const float *jt_pool, const float *bind_pool);

/*
 * default_update's camera bounding-volume pick (model.c:1703-1713): among ALIVE entities whose
 * world AABB contains the camera position or (if ctl_pos) the control entity's position, other
 * than the control entity itself, the first one of largest volume
 * (|model dx| * scale) * (|model dy| * scale) * (|model dz| * scale).  Returns its index or -1.
 */
int32_t clapo_camera_bv(uint32_t n, const uint32_t *flags, const float *aabb, const float *pos_scale,
                        const int32_t *model, const float *model_aabb,
                        const float cam_pos[3], const float *ctl_pos, int32_t ctl_entity, float *volume);

/*
 * _models_render's per-entity draw predicate (model.c:959-973):
 * ALIVE && VISIBLE && (SKIP_CULLING || view_entity_in_frustum).
 * Writes ascending entity indices to visible[] (may be NULL) and one bit per
 * entity to vis_mask[(n+63)/64] (may be NULL).  Returns the count.
 */
uint32_t clapo_entities_cull(uint32_t n, const uint32_t *flags, const float *aabb,
                             const clapo_frustum *f, uint32_t *visible, uint64_t *vis_mask);

/* ---- per-pass LOD selection (model.c:975-992; lod.c) ---- */
float clapo_aabb_avg_edge(const float model_aabb[6], float scale);
void clapo_entities_lod(uint32_t n_visible, const uint32_t *visible, const float cam_pos[3],
                        const float *aabb, const float *center, const float *pos_scale,
                        const int32_t *model, const float *model_aabb, const uint8_t *model_lod,
                        const int32_t *force_lod, int32_t *cur_lod, int32_t *draw_lod);

/* ---- particles (core/particle.c) ---- */
#define CLAPO_PART_DIST_LIN     0   /* particle.h:13-18 */
#define CLAPO_PART_DIST_SQRT    1
#define CLAPO_PART_DIST_CBRT    2
#define CLAPO_PART_DIST_POW075  3

/* one particle system; identical to clapgpu_particle_system (64 bytes) */
typedef struct clapo_particle_system {
    float    center[3];          /* transform_pos(&ps->e->xform) */
    uint32_t dist;               /* particle_dist */
    double   radius, min_radius, radius_squared, velocity;   /* particle.c:21-24 */
    uint32_t first, count;       /* particles [first, first+count) of pos/vel */
    uint32_t pad[2];
} clapo_particle_system;

uint64_t clapo_srand48(int64_t seed);            /* drand48 state after srand48(seed) */
double   clapo_drand48(uint64_t *state);
void     clapo_particles_spawn(const clapo_particle_system *sys, uint32_t n_sys,
                               float *pos, float *vel, uint64_t *rng);
/* returns the number of respawned particles; pos is also the pos_array the renderer uploads */
uint32_t clapo_particles_update(const clapo_particle_system *sys, uint32_t n_sys,
                                float *pos, float *vel, uint64_t *rng);
void     clapo_particles_billboard(const float view_mx[16], const float center[3], float mx[16]);

/* ---- skeletal pose + joint palette (core/model.c:1266-1404, core/interp.h) ---- */
/* struct model_joint[] + model3d.root_pose (model.h:104-110,59), flattened */
typedef struct clapo_skeleton {
    uint32_t       nr_joints;
    uint32_t       n_order;
    const int32_t *parent;       /* [nr_joints], -1 = child of root_pose */
    const int32_t *order;        /* joints reachable from joint 0, parents first */
    const float   *root_pose;    /* mat4 */
    const float   *invmx;        /* [nr_joints] mat4 (inverse bind) */
    const float   *bind;         /* [nr_joints] mat4 = invert(invmx) */
} clapo_skeleton;

/* struct animation / struct channel (model.c:678-685), flattened */
typedef struct clapo_animation {
    uint32_t        n_channels;
    const uint32_t *ch_target;   /* joint */
    const uint32_t *ch_path;     /* 0 translation, 1 rotation, 2 scale (enum chan_path) */
    const uint32_t *ch_nr;       /* keyframes */
    const uint32_t *ch_time_off; /* offset into times[] */
    const uint32_t *ch_data_off; /* offset into data[] (floats; 3 or 4 per keyframe) */
    const float    *times;
    const float    *data;
} clapo_animation;

/* trs[j] = (T.xyz, R.xyzw, S.xyz): struct joint's translation/rotation/scale (model.h:363-370);
 * cursor[j][path] = joint->off[path] */
void clapo_pose_channels(const clapo_animation *an, float time, float *trs, int32_t *cursor);
void clapo_pose_palette(const clapo_skeleton *sk, const float *trs, const float *entity_mx,
                        float *global, float *joint_transforms, float *joint_pos);
void clapo_skeleton_bind(uint32_t nr_joints, const float *invmx, float *bind);

/* ---- vertex skinning (shaders/model.vert:32-48; PARITY UNPINNED, see skin.c) ---- */
/* one character: n_verts vertices in the reference's attribute formats (mesh.h:125-131):
 * position f32x3, normal f32x3, joints u8x4, weights f32x4; palette = its joint_transforms */
void clapo_skin(uint32_t n_verts, const float *position, const float *normal,
                const uint8_t *joints, const float *weights,
                const float *joint_transforms, float *out_pos, float *out_nor);
void clapo_skin_w(uint32_t n_verts, const float *position, const float *normal,
                  const uint8_t *joints, const float *weights,
                  const float *joint_transforms, float *out_pos, float *out_nor, float *out_w);

/* ---- rigid bodies: schedule, integrate, broadphase (core/physics.c over ODE; PARITY UNPINNED,
 *      see physics.c) ---- */
#define CLAPO_BODY_DISABLED      (1u << 0)   /* dxBodyDisabled */
#define CLAPO_BODY_AUTO_DISABLE  (1u << 1)   /* dBodySetAutoDisableFlag(body, 1), physics.c:1039 */
#define CLAPO_BODY_NO_GRAVITY    (1u << 2)   /* dBodySetGravityMode(body, 0) */

typedef struct clapo_world {
    double  gravity[3];
    double  linear_damping;
    double  linear_damping_threshold_sq;
    double  adis_linear_threshold_sq;
    double  adis_angular_threshold_sq;
    double  adis_time;
    int32_t adis_steps;
    int32_t pad;
} clapo_world;

int      clapo_phys_step_schedule(double *time_acc, double dt);
void     clapo_world_defaults(clapo_world *w);
void     clapo_phys_body_update(uint32_t n, const double *pos, const double *quat, const double *lvel,
                                const double *yoffset, const int32_t *body_entity,
                                float *pos_scale, float *rot, uint32_t *entity_flags, uint8_t *moving);
uint64_t clapo_broadphase_pairs(uint32_t n, const double *pos, const double *radius,
                                uint32_t *pairs, uint64_t max_pairs);
uint64_t clapo_broadphase_static_pairs(uint32_t n_static, const double *static_aabb,
                                       uint32_t n, const double *pos, const double *radius,
                                       uint32_t *pairs, uint64_t max_pairs);

/* the frame on all host cores (OpenMP over tiles / mask words): context for the 1-thread baseline */
uint32_t clapo_omp_max_threads(void);
void clapo_omp_set_threads(uint32_t n);
uint32_t clapo_entities_frame_tiles_mt(uint32_t n_tiles, const uint32_t *tile_row_start, uint32_t n,
                                       const float *pos_scale, const float *rot,
                                       const int32_t *parent, const int32_t *model,
                                       const float *model_aabb, const uint8_t *model_skip_aabb,
                                       uint32_t *flags, uint32_t *seqs,
                                       float *mx, float *inv_mx, float *aabb, float *center,
                                       const clapo_frustum *f, uint64_t *vis_mask);

/* ---- animated_update's clock (model.c:1563-1592; pose.c) ---- */
void clapo_animation_time(uint32_t n_chars, uint32_t n_anims, const uint32_t *anim, const float *time_end,
                          double *ani_time, const float *speed, const uint8_t *restart, double now,
                          float *frame_time, uint8_t *ended);

/* ---- character feeder in front of default_update (character.c:546-611; character.c) ---- */
void clapo_characters_update(uint32_t n_chars, const uint32_t *char_entity, const int32_t *char_body,
                             float limbo_height, float *hist_pos, uint32_t *hist_head, uint8_t *hist_wrapped,
                             const uint8_t *airborne, float *pos_scale, uint32_t *entity_flags,
                             double *body_pos, const double *body_lvel, const double *body_yoffset,
                             uint8_t *moved);

void clapo_bodies_rotate_from_entities(uint32_t n_links, const uint32_t *link_body, const uint32_t *link_entity,
                                       const float *rot, const int32_t *parent, const uint8_t *dirty, double *quat);

/* ---- sphere contacts after the broadphase (physics.c:291-330, 399-449; physics.c) ---- */
typedef struct clapo_contact {
    double   pos[3], normal[3], depth;                  /* dContactGeom */
    double   mu, bounce, bounce_vel, soft_erp, soft_cfm;/* dSurfaceParameters as phys_contact_surface fills them */
    uint32_t mode;                                      /* dContactSoftCFM | dContactSoftERP [| dContactBounce] */
    uint32_t nc;                                        /* dCollide's return value for the pair: 0 or 1 */
} clapo_contact;
uint32_t clapo_contacts_spheres(uint32_t n_pairs, const uint32_t *pairs, const double *pos, const double *radius,
                                const double *material, clapo_contact *out);
/* the same for (sphere body, static axis-aligned box) pairs: ODE's dCollideSphereBox, restated (PARITY UNPINNED) */
uint32_t clapo_contacts_sphere_box(uint32_t n_pairs, const uint32_t *pairs, const double *pos, const double *radius,
                                   const double *static_aabb, const double *material, const double *static_material,
                                   clapo_contact *out);


/* ---- round 2: capsule bodies, general AABBs, capsule contacts, capsule sweep (physics2.c; PARITY UNPINNED: ODE) ---- */
#define CLAPO_BODY_GYROSCOPIC    (1u << 3)   /* dxBodyGyroscopic: set by dBodyCreate, cleared by dBodySetGyroscopicMode(b, 0) */
#define CLAPO_BODY_HAS_JOINT     (1u << 4)   /* the body has a (contact) joint this step: dInternalHandleAutoDisabling skips jointless bodies */
#define CLAPO_GEOM_SPHERE  0
#define CLAPO_GEOM_CAPSULE 1
#define CLAPO_GEOM_BOX     2                 /* an axis-aligned box given by its AABB (stand-in for any static geom) */
#define CLAPO_GEOM_OTHER   3                 /* trimesh / anything without a narrowphase here: broadphase only */

/* same layout as clapgpu_bodies (include/clapgpu.h) */
typedef struct clapo_bodies {
    uint32_t        n;
    uint32_t        adis_average_samples;   /* dBodySetAutoDisableAverageSamplesCount; 0 and 1 = ODE's default single sample */
    double         *pos, *quat, *lvel, *avel;
    const double   *mass, *radius, *yoffset;
    uint32_t       *bflags;
    int32_t        *adis_steps_left;
    double         *adis_time_left;
    const int32_t  *body_entity;
    const double   *length;                 /* [n] capsule cylinder length, 0 = sphere; NULL = all spheres */
    const double   *inertia;                /* [n][3] diagonal of dMass.I (body frame); NULL = no gyroscopic torque */
    double          geom_offset_R[12];      /* dGeomSetOffsetRotation of the capsule geoms (physics.c:974-978), dMatrix3 */
    double         *aabb;                   /* [n][6] out: geom AABB (minx,maxx,miny,maxy,minz,maxz) */
    double         *axis;                   /* [n][3] out: capsule axis = column 2 of the geom's final rotation */
    double         *adis_samples;           /* [n][samples][6] lvel, avel ring (samples > 1 only) */
    uint32_t       *adis_counter;           /* [n] average_counter | average_ready << 31 */
} clapo_bodies;

void clapo_geom_offset_rotation(double R[12]);                                         /* physics.c:974-978 */
void clapo_mass_sphere_total(double total_mass, double radius, double I[3]);           /* dMassSetSphereTotal */
void clapo_mass_capsule_total(double total_mass, int direction, double radius, double length, double I[3]);
/* phys_geom_capsule_new (physics.c:814-873): entity AABB extents X, Y, Z -> radius, length, yoffset, direction, ray_off */
void clapo_capsule_geom(float X, float Y, float Z, double geom_radius, double geom_offset,
                        float *r, float *length, float *yoffset, int *direction, float *ray_off);
void clapo_bodies_aabb(const clapo_bodies *b);
void clapo_bodies_step2(const clapo_bodies *b, const clapo_world *w, double h);
uint64_t clapo_broadphase_aabb_pairs(uint32_t n, const double *aabb, uint32_t *pairs, uint64_t max_pairs);
uint64_t clapo_broadphase_aabb_static_pairs(uint32_t n_static, const double *static_aabb, uint32_t n, const double *aabb,
                                            uint32_t *pairs, uint64_t max_pairs);

/* narrowphase view of a geom set (bodies after clapo_bodies_aabb, or statics) */
typedef struct clapo_geoms {
    uint32_t        n, pad;
    const double   *pos;                    /* [n][3] geom position (box: unused, the centre of aabb is taken) */
    const double   *axis;                   /* [n][3] capsule axis (unit) */
    const double   *radius, *length;        /* [n] (length NULL = all spheres / boxes) */
    const uint8_t  *kind;                   /* [n] CLAPO_GEOM_*; NULL = sphere when length is 0, else capsule */
    const double   *aabb;                   /* [n][6] boxes: position = centre, side = extent */
    const double   *material;               /* [n][5] bounce, bounce_vel, mu, soft_erp, soft_cfm; may be NULL */
} clapo_geoms;

#define CLAPO_CONTACT_DEEP 0x80000000u      /* capsule axis inside a box: ODE switches to dBoxBox; left to the host */
typedef struct clapo_contact2 {
    double   pos[3], normal[3], depth;                  /* first dContactGeom */
    double   mu, bounce, bounce_vel, soft_erp, soft_cfm;
    uint32_t mode;
    uint32_t nc;                                        /* 0, 1 or 2 (| CLAPO_CONTACT_DEEP) */
    double   pos2[3], normal2[3], depth2;               /* second dContactGeom (parallel capsules) */
} clapo_contact2;
/* near_callback on candidate pairs (ia in A, ib in B): g1 = A's geom, g2 = B's geom.  Returns the number of touching pairs. */
uint32_t clapo_contacts_geoms(uint32_t n_pairs, const uint32_t *pairs, const clapo_geoms *A, const clapo_geoms *B,
                              clapo_contact2 *out);
/* phys_body_sweep_capsule (physics.c:559-670) of body `self` of A along delta against candidate geoms:
 * cand[k] = index into B (statics) or, with bit 31 set, into A (other bodies).  Returns best_frac. */
float clapo_sweep_capsule(const clapo_geoms *A, uint32_t self, const float delta[3], const clapo_geoms *B,
                          uint32_t n_cand, const uint32_t *cand, float normal[3], int32_t *hit);

/* ---- clustered-lighting tile masks (light.c:88-154, 301-309; light.c) ---- */
float clapo_light_radius(const float color[3], const float att[3], int is_dir);
void clapo_light_grid_dims(uint32_t width, uint32_t height, uint32_t cell, uint32_t *twidth, uint32_t *theight);
void clapo_light_grid_compute(uint32_t nr_lights, const uint32_t *active, const int32_t *is_dir,
                              const float *pos, const float *color, const float *attenuation,
                              const float view_mx[16], const float proj_mx[16],
                              uint32_t width, uint32_t height, uint32_t cell, uint32_t *tiles);
void clapo_lights_from_entities(uint32_t n_carriers, const uint32_t *carrier_entity, const int32_t *carrier_light,
                                const float *carrier_off, const float *pos_scale, const int32_t *parent,
                                const uint8_t *dirty, uint32_t nr_lights, const uint32_t *active, float *light_pos);

#ifdef __cplusplus
}
#endif

#endif /* CLAP_ORACLE_H */
