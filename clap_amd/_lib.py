"""ctypes binding of clap_amd/lib/libclapgpu.so (the C ABI of include/clapgpu.h).

There is no CPU fallback: if the HIP library is missing or a call fails the
binding raises.  Nothing here imports ``oracle``.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLAPGPU_LIB") or os.path.join(_HERE, "lib", "libclapgpu.so")   # override: A/B builds
CSRC = os.path.join(_HERE, "csrc")
ABI_VERSION = 32

OK = 0
ERR_NOMEM = -1
ERR_INVALID_ARGUMENTS = -2
ERR_NOT_SUPPORTED = -3
ERR_TOO_LARGE = -11
ERR_INIT_FAILED = -14
ERR_OUT_OF_BOUNDS = -26
ERR_UNKNOWN = -32

E_VISIBLE = 1 << 0
E_SKIP_CULLING = 1 << 14
E_DIRTY = 1 << 16
E_JOINT_ATTACHED = 1 << 17
E_ALIVE = 1 << 31
UPDATE_ALL_DIRTY = 1 << 0

_ERR_NAMES = {ERR_NOMEM: "NOMEM", ERR_INVALID_ARGUMENTS: "INVALID_ARGUMENTS", ERR_NOT_SUPPORTED: "NOT_SUPPORTED",
              ERR_TOO_LARGE: "TOO_LARGE", ERR_INIT_FAILED: "INITIALIZATION_FAILED",
              ERR_OUT_OF_BOUNDS: "OUT_OF_BOUNDS", ERR_UNKNOWN: "UNKNOWN_ERROR"}


class ClapGpuError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: CERR_{_ERR_NAMES.get(code, code)} ({code}) {detail}".rstrip())


class Frustum(C.Structure):
    """clapgpu_frustum: planes[6][4], corners[8][4] (view.h:16-17)."""
    _fields_ = [("planes", C.c_float * 24), ("corners", C.c_float * 32)]


class BvQuery(C.Structure):
    """clapgpu_bv_query (include/clapgpu.h)."""
    _fields_ = [("cam_pos", C.c_float * 3), ("has_ctl", C.c_uint32), ("ctl_pos", C.c_float * 3),
                ("ctl_entity", C.c_uint32), ("result", C.c_void_p), ("inside_mask", C.c_void_p)]


EXTRA_VIEWS_MAX = 4


class Views(C.Structure):
    """clapgpu_views (include/clapgpu.h): the frame's other frusta, culled by the same launch as the main one."""
    _fields_ = [("n", C.c_uint32), ("pad", C.c_uint32), ("frustum", Frustum * EXTRA_VIEWS_MAX),
                ("vis_mask", C.c_void_p * EXTRA_VIEWS_MAX), ("vis_row_pop", C.c_void_p * EXTRA_VIEWS_MAX),
                ("host_vis_mask", C.c_void_p * EXTRA_VIEWS_MAX)]


class Entities(C.Structure):
    """clapgpu_entities (include/clapgpu.h)."""
    _fields_ = [("n", C.c_uint32), ("n_models", C.c_uint32),
                ("pos_scale", C.c_void_p), ("rot", C.c_void_p), ("parent", C.c_void_p),
                ("model", C.c_void_p), ("model_table", C.c_void_p), ("flags", C.c_void_p),
                ("seqs", C.c_void_p), ("mx", C.c_void_p), ("inv_mx", C.c_void_p),
                ("aabb", C.c_void_p), ("center", C.c_void_p), ("vis_mask", C.c_void_p),
                ("vis_row_pop", C.c_void_p), ("n_attach", C.c_uint32), ("pad", C.c_uint32),
                ("attach", C.c_void_p), ("jt_pool", C.c_void_p), ("bind_pool", C.c_void_p),
                ("attach_local", C.c_void_p), ("bv", C.POINTER(BvQuery)), ("rebuilt_mask", C.c_void_p),
                ("views", C.POINTER(Views))]


class Particles(C.Structure):
    """clapgpu_particles (include/clapgpu.h)."""
    _fields_ = [("n", C.c_uint32), ("n_sys", C.c_uint32), ("sys", C.c_void_p), ("row_sys", C.c_void_p),
                ("pos", C.c_void_p), ("vel", C.c_void_p), ("rng_state", C.c_void_p),
                ("billboard_mx", C.c_void_p), ("respawn_mask", C.c_void_p), ("respawn_row_pop", C.c_void_p),
                ("respawn_list", C.c_void_p), ("respawn_count", C.c_void_p), ("scratch", C.c_void_p),
                ("respawn_groups", C.c_void_p)]


class AnimClock(C.Structure):
    """clapgpu_anim_clock (include/clapgpu.h)."""
    _fields_ = [("n_chars", C.c_uint32), ("n_anims", C.c_uint32), ("anim", C.c_void_p), ("time_end", C.c_void_p),
                ("ani_time", C.c_void_p), ("speed", C.c_void_p), ("restart", C.c_void_p), ("frame_time", C.c_void_p),
                ("ended", C.c_void_p)]


class Skeleton(C.Structure):
    _fields_ = [("nr_joints", C.c_uint32), ("n_levels", C.c_uint32), ("parent", C.c_void_p), ("depth", C.c_void_p),
                ("root_pose", C.c_void_p), ("invmx", C.c_void_p), ("bind", C.c_void_p)]


class Animations(C.Structure):
    _fields_ = [("n_anims", C.c_uint32), ("n_times", C.c_uint32), ("chan_table", C.c_void_p),
                ("times", C.c_void_p), ("data", C.c_void_p), ("packed", C.c_void_p), ("packed_keys", C.c_uint32),
                ("packed_layout", C.c_uint32), ("n_data", C.c_uint32), ("pad", C.c_uint32)]


class PoseBatch(C.Structure):
    _fields_ = [("n_chars", C.c_uint32), ("skip", C.c_uint32), ("anim", C.c_void_p), ("frame_time", C.c_void_p), ("entity", C.c_void_p),
                ("entity_mx", C.c_void_p), ("trs", C.c_void_p), ("joint_transforms", C.c_void_p),
                ("joint_pos", C.c_void_p)]


class SkinBatch(C.Structure):
    _fields_ = [("n_chars", C.c_uint32), ("nr_joints", C.c_uint32), ("vert_first", C.c_void_p),
                ("vert_count", C.c_void_p), ("out_first", C.c_void_p), ("position", C.c_void_p),
                ("normal", C.c_void_p), ("joints", C.c_void_p), ("weights", C.c_void_p),
                ("joint_transforms", C.c_void_p), ("out_position", C.c_void_p), ("out_normal", C.c_void_p),
                ("out_w", C.c_void_p)]


class World(C.Structure):
    _fields_ = [("gravity", C.c_double * 3), ("linear_damping", C.c_double),
                ("linear_damping_threshold_sq", C.c_double), ("adis_linear_threshold_sq", C.c_double),
                ("adis_angular_threshold_sq", C.c_double), ("adis_time", C.c_double),
                ("adis_steps", C.c_int32), ("pad", C.c_int32)]


class Bodies(C.Structure):
    """clapgpu_bodies (include/clapgpu.h)."""
    _fields_ = [("n", C.c_uint32), ("adis_average_samples", C.c_uint32), ("pos", C.c_void_p), ("quat", C.c_void_p),
                ("lvel", C.c_void_p), ("avel", C.c_void_p), ("mass", C.c_void_p), ("radius", C.c_void_p),
                ("yoffset", C.c_void_p), ("bflags", C.c_void_p), ("adis_steps_left", C.c_void_p),
                ("adis_time_left", C.c_void_p), ("body_entity", C.c_void_p),
                ("length", C.c_void_p), ("inertia", C.c_void_p), ("geom_offset_R", C.c_double * 12),
                ("aabb", C.c_void_p), ("axis", C.c_void_p), ("adis_samples", C.c_void_p), ("adis_counter", C.c_void_p),
                ("geom_records", C.c_void_p)]


class Geoms(C.Structure):
    """clapgpu_geoms (include/clapgpu.h)."""
    _fields_ = [("n", C.c_uint32), ("pad", C.c_uint32), ("pos", C.c_void_p), ("axis", C.c_void_p),
                ("radius", C.c_void_p), ("length", C.c_void_p), ("kind", C.c_void_p), ("aabb", C.c_void_p),
                ("material", C.c_void_p), ("records", C.c_void_p)]


POSE_SKIP_TRS, POSE_SKIP_JOINT_POS, POSE_JOINT_POS_MODEL = 1, 2, 4
BODY_DISABLED, BODY_AUTO_DISABLE, BODY_NO_GRAVITY, BODY_GYROSCOPIC, BODY_HAS_JOINT = 1, 2, 4, 8, 16
GEOM_SPHERE, GEOM_CAPSULE, GEOM_BOX, GEOM_OTHER = 0, 1, 2, 3
CONTACT_DEEP = 0x80000000


class Characters(C.Structure):
    """clapgpu_characters (include/clapgpu.h)."""
    _fields_ = [("n", C.c_uint32), ("limbo_height", C.c_float), ("entity", C.c_void_p), ("body", C.c_void_p),
                ("hist_pos", C.c_void_p), ("hist_head", C.c_void_p), ("hist_wrapped", C.c_void_p),
                ("airborne", C.c_void_p), ("moved", C.c_void_p)]


class Lights(C.Structure):
    """clapgpu_lights (include/clapgpu.h)."""
    _fields_ = [("nr_lights", C.c_uint32), ("pad", C.c_uint32), ("pos", C.c_void_p), ("color", C.c_void_p),
                ("attenuation", C.c_void_p), ("is_dir", C.c_void_p), ("active", C.c_void_p)]


class Frame(C.Structure):
    """clapgpu_frame (include/clapgpu.h)."""
    _fields_ = [("entities", C.POINTER(Entities)), ("tile_row_start", C.c_void_p), ("n_tiles", C.c_uint32),
                ("level_start", C.c_void_p), ("n_levels", C.c_uint32), ("frustum", C.POINTER(Frustum)),
                ("bodies", C.POINTER(Bodies)), ("world", C.POINTER(World)), ("bp", C.c_void_p),
                ("pairs", C.c_void_p), ("pair_capacity", C.c_uint32), ("pair_total", C.c_void_p),
                ("static_pairs", C.c_void_p), ("static_pair_capacity", C.c_uint32), ("static_pair_total", C.c_void_p),
                ("body_geoms", C.POINTER(Geoms)), ("static_geoms", C.POINTER(Geoms)),
                ("contacts", C.c_void_p), ("static_contacts", C.c_void_p),
                ("contact_total", C.c_void_p), ("static_contact_total", C.c_void_p),
                ("n_body_links", C.c_uint32), ("link_body", C.c_void_p), ("link_entity", C.c_void_p),
                ("characters", C.POINTER(Characters)),
                ("lights", C.POINTER(Lights)),
                ("n_light_carriers", C.c_uint32), ("carrier_entity", C.c_void_p), ("carrier_light", C.c_void_p),
                ("carrier_offset", C.c_void_p),
                ("light_width", C.c_uint32), ("light_height", C.c_uint32), ("light_cell", C.c_uint32), ("light_tiles", C.c_void_p),
                ("view_mx", C.POINTER(C.c_float)), ("proj_mx", C.POINTER(C.c_float)),
                ("anim_clock", C.POINTER(AnimClock)), ("now_dev", C.c_void_p),
                ("skeleton", C.POINTER(Skeleton)), ("animations", C.POINTER(Animations)), ("pose", C.POINTER(PoseBatch)),
                ("skin", C.POINTER(SkinBatch)),
                ("particles", C.POINTER(Particles)),
                ("index_base", C.c_uint32), ("visible", C.c_void_p), ("visible_count", C.c_void_p), ("visible_scratch", C.c_void_p),
                ("cam_pos", C.c_float * 3), ("force_lod", C.c_void_p), ("cur_lod", C.c_void_p), ("draw_lod", C.c_void_p),
                ("flags", C.c_uint32)]


LIGHTS_MAX = 128


# every symbol include/clapgpu.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "clapgpu_device_count": (C.c_int, []),
    "clapgpu_init": (C.c_int, [C.c_int]),
    "clapgpu_last_error": (C.c_char_p, []),
    "clapgpu_test_fail_after": (None, [C.c_int]),
    "clapgpu_abi_version": (C.c_uint32, []),
    "clapgpu_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "clapgpu_free": (C.c_int, [C.c_void_p]),
    "clapgpu_host_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "clapgpu_host_free": (C.c_int, [C.c_void_p]),
    "clapgpu_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "clapgpu_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "clapgpu_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "clapgpu_stream_sync": (C.c_int, [C.c_void_p]),
    "clapgpu_view_matrix": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "clapgpu_perspective": (None, [C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_float)]),
    "clapgpu_frustum_calc": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(Frustum)]),
    "clapgpu_entities_update": (C.c_int, [C.c_void_p, C.POINTER(Entities), C.POINTER(C.c_uint32), C.c_uint32,
                                          C.c_uint32, C.POINTER(Frustum)]),
    "clapgpu_entities_update_level": (C.c_int, [C.c_void_p, C.POINTER(Entities), C.c_uint32, C.c_uint32,
                                                C.c_uint32, C.POINTER(Frustum)]),
    "clapgpu_entities_update_tiles": (C.c_int, [C.c_void_p, C.POINTER(Entities), C.c_void_p, C.c_uint32,
                                                C.c_uint32, C.POINTER(Frustum)]),
    "clapgpu_entities_lod": (C.c_int, [C.c_void_p, C.POINTER(Entities), C.c_void_p, C.c_void_p, C.c_uint32,
                                       C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p]),
    "clapgpu_entities_cull": (C.c_int, [C.c_void_p, C.POINTER(Entities), C.POINTER(Frustum)]),
    "clapgpu_visible_scratch_bytes": (C.c_size_t, [C.c_uint32]),
    "clapgpu_visible_compact": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "clapgpu_animation_time": (C.c_int, [C.c_void_p, C.POINTER(AnimClock), C.c_double]),
    "clapgpu_animation_time_dev": (C.c_int, [C.c_void_p, C.POINTER(AnimClock), C.c_void_p]),
    "clapgpu_animations_packed_bytes": (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "clapgpu_animations_pack": (C.c_int, [C.c_void_p, C.POINTER(Animations), C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]),
    "clapgpu_joint_pos_world": (C.c_int, [C.c_void_p, C.POINTER(Skeleton), C.POINTER(PoseBatch)]),
    "clapgpu_pose_update": (C.c_int, [C.c_void_p, C.POINTER(Skeleton), C.POINTER(Animations), C.POINTER(PoseBatch)]),
    "clapgpu_skin": (C.c_int, [C.c_void_p, C.POINTER(SkinBatch)]),
    "clapgpu_phys_step_schedule": (C.c_int, [C.POINTER(C.c_double), C.c_double]),
    "clapgpu_world_defaults": (None, [C.POINTER(World)]),
    "clapgpu_bodies_step": (C.c_int, [C.c_void_p, C.POINTER(Bodies), C.POINTER(World), C.c_double]),
    "clapgpu_phys_body_update": (C.c_int, [C.c_void_p, C.POINTER(Bodies), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    "clapgpu_geom_offset_rotation": (None, [C.POINTER(C.c_double)]),
    "clapgpu_mass_sphere_total": (None, [C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "clapgpu_mass_capsule_total": (None, [C.c_double, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "clapgpu_capsule_geom": (None, [C.c_float, C.c_float, C.c_float, C.c_double, C.c_double, C.POINTER(C.c_float),
                                    C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "clapgpu_bodies_aabb": (C.c_int, [C.c_void_p, C.POINTER(Bodies)]),
    "clapgpu_bp_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_double, C.c_uint32, C.c_void_p]),
    "clapgpu_bp_destroy": (None, [C.c_void_p]),
    "clapgpu_bp_collide": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                     C.c_void_p, C.c_uint32, C.c_void_p]),
    "clapgpu_bp_status": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "clapgpu_bp_static_aabb": (C.c_void_p, [C.c_void_p]),
    "clapgpu_contacts_geoms": (C.c_int, [C.c_void_p, C.POINTER(Geoms), C.POINTER(Geoms), C.c_void_p, C.c_void_p, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "clapgpu_contacts_geoms_both": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Geoms), C.POINTER(Geoms), C.c_void_p, C.c_void_p,
                                              C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    "clapgpu_sweep_capsules": (C.c_int, [C.c_void_p, C.POINTER(Geoms), C.POINTER(Geoms), C.c_uint32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "clapgpu_bodies_rotate_from_entities": (C.c_int, [C.c_void_p, C.POINTER(Bodies), C.POINTER(Entities), C.c_uint32,
                                                      C.c_uint32, C.c_void_p, C.c_void_p]),
    "clapgpu_contacts_spheres": (C.c_int, [C.c_void_p, C.POINTER(Bodies), C.c_void_p, C.c_void_p, C.c_uint32,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "clapgpu_contacts_sphere_box": (C.c_int, [C.c_void_p, C.POINTER(Bodies), C.c_uint32, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "clapgpu_shard_tile_range": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_uint32)]),
    "clapgpu_bodies_step_prebin": (C.c_int, [C.c_void_p, C.POINTER(Bodies), C.POINTER(World), C.c_double, C.c_void_p]),
    "clapgpu_bp_invalidate": (C.c_int, [C.c_void_p, C.c_void_p]),
    "clapgpu_visible_compact_lod": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "clapgpu_characters_update_clock": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "clapgpu_host_malloc_mapped": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t]),
    "clapgpu_entities_apply_inputs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "clapgpu_entities_place": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "clapgpu_entities_export_rebuilt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "clapgpu_entities_export_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "clapgpu_entities_update_tiles_hostio": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                                       C.c_void_p]),
    "clapgpu_wait_word": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "clapgpu_exchange_set_library": (None, [C.c_char_p]),
    "clapgpu_exchange_available": (C.c_int, []),
    "clapgpu_exchange_unique_id": (C.c_int, [C.c_void_p]),
    "clapgpu_exchange_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int]),
    "clapgpu_exchange_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p]),
    "clapgpu_exchange_destroy": (None, [C.c_void_p]),
    "clapgpu_exchange_visible": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    "clapgpu_exchange_visible_ranges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "clapgpu_visible_compact_ranges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]),
    "clapgpu_visible_expand_ranges_host": (C.c_uint32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "clapgpu_shard_bases": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    "clapgpu_mat4_invert": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "clapgpu_mat4_from_quat": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "clapgpu_frame_issue": (C.c_int, [C.c_void_p, C.POINTER(Frame), C.c_double, C.c_uint32]),
    "clapgpu_particles_update": (C.c_int, [C.c_void_p, C.POINTER(Particles), C.POINTER(C.c_float)]),
    "clapgpu_characters_update": (C.c_int, [C.c_void_p, C.POINTER(Characters), C.POINTER(Entities),
                                            C.POINTER(Bodies)]),
    "clapgpu_light_grid_dims": (None, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_uint32)]),
    "clapgpu_light_grid_compute": (C.c_int, [C.c_void_p, C.POINTER(Lights), C.POINTER(C.c_float),
                                             C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "clapgpu_lights_from_entities": (C.c_int, [C.c_void_p, C.POINTER(Entities), C.c_uint32, C.c_uint32, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.POINTER(Lights)]),
}

_lib = None


def build(verbose=False):
    """Compile every HIP source for gfx950 into clap_amd/lib/libclapgpu.so (needs hipcc, no GPU)."""
    subprocess.run(["make", "-C", CSRC] + ([] if verbose else ["-s"]), check=True)


def lib():
    """Load the library; raise loudly if it is missing or its ABI differs."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ClapGpuError(ERR_INIT_FAILED, "clap_amd",
                               f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)")
        # This harness shares device memory and streams with torch, and torch ships its own copy of the HIP / HSA runtime:
        # whichever copy is loaded first serves the whole process, and a second HSA runtime finds no device ("no HIP
        # device" from clapgpu_init).  Let torch load its copy before the library resolves libamdhip64.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)           # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        got = L.clapgpu_abi_version()
        if got & 0x80000000 and os.environ.get("CLAPGPU_ALLOW_EXPERIMENT") == "1":
            got &= 0x7FFFFFFF                # an experiment build (csrc/common.h), asked for by name: A/B tools only
        if got != ABI_VERSION:
            raise ClapGpuError(ERR_INIT_FAILED, "clap_amd",
                               "libclapgpu ABI version mismatch; rebuild" if not got & 0x80000000 else
                               "libclapgpu is an EXPERIMENT build (-DCLAPGPU_EXPERIMENT: kernels compiled with parts "
                               "switched off); rebuild without EXTRA=, or set CLAPGPU_ALLOW_EXPERIMENT=1 for an A/B run")
        _lib = L
    return _lib


def check(rc, where):
    if rc != OK:
        detail = lib().clapgpu_last_error()
        raise ClapGpuError(rc, where, detail.decode() if detail else "")
