/*
 * clapgpu_load.h -- CLAP's scene files -> SoA scene snapshot.
 * Part of libclapgpu_scene.so, plain C, no GPU.
 *
 * The engine builds its scene by walking scene.json (scene.c:1318-1724 model_new_from_json,
 * scene.c:1726-1813 scene_add_light_from_json, scene.c:1816-1884 scene_onload) and, per model, one
 * glTF 2.0 asset (.glb or .gltf with base64 buffers: gltf.c:666-1124 parse, gltf.c:1158-1331
 * gltf_instantiate_one -- mesh attributes, skin, animations).  clapgpu_load_scene() walks the same
 * files in the same order with the same rules (which keys are read, their defaults, which entries are
 * skipped) and writes what the hot path consumes as snapshot arrays (include/clapgpu_snapshot.h):
 *
 *   entities.*    n, pos_scale[n][4], rot[n][4] (x,y,z,w), parent[n] (list index or -1), parent_joint[n]
 *                 (joint index or -1), model[n], flags[n] (ENTITY3D_* bits, model.h:294-310, + CLAPGPU_E_DIRTY),
 *                 seqs[n] = 0, model_aabb[m][6], model_skip[m]: list order = creation order, as on mq's lists
 *   model<k>.*    per skinned model k: nr_joints, joint_parent[J] (from the joints' children lists, -1 = root),
 *                 invmx[J][16], bind[J][16] (mat4x4_invert, model.c:524-537), root_pose[16], joint_types[6],
 *                 n_verts, position / normal / joints (u8x4) / weights; n_anims, and per animation a:
 *                 a<a>_ch_target / _ch_path / _ch_nr / _ch_time_off / _ch_data_off / _times / _data / _time_end
 *                 (channel order and time_end as animation_add_channel leaves them, model.c:725-742);
 *                 weight_sum_max_dev = max over vertices of |sum(weights) - 1| (the shader's total_local_pos.w is that
 *                 sum, model.vert:36-38: clapgpu_skin_batch.out_w); a<a>_ch_nonstrict[c] = keys of channel c whose time
 *                 does not exceed the previous key's, key_times_nonstrict = their total over the model (see below)
 *   characters.*  entity[], model[], speed[], can_jump[], can_dash[]      ("character" arrays, scene.c:1508-1515)
 *   lights.*      nr_lights, pos / color / attenuation / dir [128][3], cutoff[128], is_dir[128], active[128],
 *                 ambient[3], shadow_tint[3]                               (light.c:311-340, 473-530)
 *   carriers.*    entity[], light[], offset[][3]                           (light_color / light_offset, scene.c:1587-1632)
 *   attach.*      entity[], parent[], joint[]                              ("attach" / "attach_joint", scene.c:1529-1541)
 *   bodies.*      entity[], geom_class[], phys_type[], mass[], radius[], length[], yoffset[], bounce[], bounce_vel[]
 *                                                                          ("physics", scene.c:1436-1466, 1653-1661)
 *
 * Channels are kept as listed.  Should two channels of one animation drive the same (joint, path), the engine shares one
 * keyframe cursor between them (joint->off[path], model.c:1305-1311) and its result depends on that cursor's history;
 * the device path evaluates the last listed one statelessly.  Assets the engine plays correctly have no such pair.
 * Key times that do not strictly increase (equal or descending neighbours) are kept and FLAGGED (a<a>_ch_nonstrict,
 * key_times_nonstrict): channel_time_to_idx scans from the cursor joint->off[path] of the previous frame (model.c:1266-1288,
 * 1310), so the bracket it finds among equal times depends on that history; the device search is stateless and
 * takes the first key with time <= t[i] as if the cursor were 0.  glTF requires strictly increasing input accessors.
 * Not restated: mesh_optimize()'s vertex reordering (meshoptimizer, an absent third-party dependency:
 * vertices stay in file order, which permutes the skinned output, not its values), textures, materials,
 * sfx, the editor's instantiators (models without an "entity" / "character" array create no entities here).
 * Returns cerr_enum-compatible ints (error.h:12-49): 0, or negative (-4 CERR_PARSE_FAILED-like, -2 not found).
 */
#ifndef CLAPGPU_LOAD_H
#define CLAPGPU_LOAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* scene_json: path of the scene file; asset_dir: where its "gltf" names are looked up (NULL = the scene
 * file's directory); snapshot_path: output.  err (may be NULL) receives a one-line reason on failure. */
int clapgpu_load_scene(const char *scene_json, const char *asset_dir, const char *snapshot_path,
                       char *err, size_t err_len);

/* One glTF asset alone -> snapshot with entities.model_aabb / model_skip of its instantiated mesh and, if the
 * mesh is skinned, component model0.* (mesh choice as model_new_from_json makes it: the scene's root mesh, or
 * the first mesh that is not named "collision"). */
int clapgpu_load_gltf(const char *gltf_path, int fix_origin, const char *snapshot_path, char *err, size_t err_len);

#ifdef __cplusplus
}
#endif
#endif /* CLAPGPU_LOAD_H */
