// frame.hip -- one display frame of the batched path as ONE C call: the sequence clap_frame() runs
// (core/clap.c:551-665) -- phys_step (clap.c:604: per fixed substep the two broadphase passes, near_callback's
// contact records, the world step) -> scene_update -> mq_update with every entity's hook in list order
// (character_update in front of default_update: body read-back, rotation push to colliders, light hand-off, TRS
// rebuild, animated_update; particles_update) -> light grid -> render-pass glue (visible list, LOD pick) -- issued
// as a fixed sequence of launches on one stream.  Nothing is read back.  Every part is optional (NULL).
// The caller keeps the time base (clapgpu_phys_step_schedule) and passes the number of substeps.
#include <string.h>
#include "common.h"

using namespace clapgpu;

#define FR(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

extern "C" int clapgpu_frame_issue(void *stream, const clapgpu_frame *f, double now, uint32_t substeps)
{
    if (!f || !f->entities) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const clapgpu_entities *e = f->entities;

    // ---- phys_step: per substep broadphase x2, contacts, dWorldQuickStep's body stage (physics.c:746-771) ----
    if (f->bodies && f->world) {
        for (uint32_t s = 0; s < substeps; s++) {
            if (f->bp) {
                FR(clapgpu_bp_collide(stream, f->bp, f->bodies->n, f->bodies->aabb, f->pairs, f->pair_capacity, f->pair_total,
                                      f->static_pairs, f->static_pair_capacity, f->static_pair_total));
                if (f->body_geoms && f->contacts && f->static_geoms && f->static_contacts && f->static_pair_total &&
                    f->pair_capacity < (1u << 24) && f->static_pair_capacity < (1u << 24)) {
                    FR(clapgpu_contacts_geoms_both(stream, f->bp, f->body_geoms, f->static_geoms, f->pairs, f->pair_total,
                                                   f->pair_capacity, f->contacts, f->contact_total, f->static_pairs,
                                                   f->static_pair_total, f->static_pair_capacity, f->static_contacts,
                                                   f->static_contact_total, f->bodies->bflags));
                } else if (f->body_geoms && f->contacts) {
                    FR(clapgpu_contacts_geoms(stream, f->body_geoms, f->body_geoms, f->pairs, f->pair_total, f->pair_capacity,
                                              f->contacts, f->contact_total, f->bodies->bflags, f->bodies->bflags));
                    if (f->static_geoms && f->static_contacts && f->static_pair_total)
                        FR(clapgpu_contacts_geoms(stream, f->body_geoms, f->static_geoms, f->static_pairs, f->static_pair_total,
                                                  f->static_pair_capacity, f->static_contacts, f->static_contact_total,
                                                  f->bodies->bflags, nullptr));
                }
            }
            FR(clapgpu_bodies_step(stream, f->bodies, f->world, 1.0 / 120.0));          // fixed_dt, physics.c:775
        }
    }
    // ---- character_update hooks (character.c:583-611) ----
    if (f->characters)
        FR(clapgpu_characters_update(stream, f->characters, e, f->bodies));
    // ---- default_update: phys_body_update of dynamic bodies (model.c:1659-1665), rotation push (1680-1687),
    //      light hand-off (1689-1694) ----
    if (f->bodies) {
        FR(clapgpu_phys_body_update(stream, f->bodies, e->n, const_cast<float *>(e->pos_scale), const_cast<float *>(e->rot),
                                    e->flags, nullptr));
        if (f->n_body_links)
            FR(clapgpu_bodies_rotate_from_entities(stream, f->bodies, e, 0, f->n_body_links, f->link_body, f->link_entity));
    }
    if (f->lights && f->n_light_carriers)
        FR(clapgpu_lights_from_entities(stream, e, 0, f->n_light_carriers, f->carrier_entity, f->carrier_light, f->carrier_offset,
                                        f->lights));
    // ---- TRS -> mx -> inverse -> AABB (+ the main view's cull) ----
    if (f->tile_row_start)
        FR(clapgpu_entities_update_tiles(stream, e, f->tile_row_start, f->n_tiles, 0, f->frustum));
    else
        FR(clapgpu_entities_update(stream, e, f->level_start, f->n_levels, 0, f->frustum));
    // ---- animated_update: clock, pose, palette; the vertex shader's skinning loop once per frame ----
    if (f->skeleton && f->animations && f->pose) {
        if (f->anim_clock) {
            if (f->now_dev) FR(clapgpu_animation_time_dev(stream, f->anim_clock, f->now_dev));
            else FR(clapgpu_animation_time(stream, f->anim_clock, now));
        }
        FR(clapgpu_pose_update(stream, f->skeleton, f->animations, f->pose));
        if (f->skin)
            FR(clapgpu_skin(stream, f->skin));
    }
    // ---- particles_update hooks ----
    if (f->particles && f->view_mx)
        FR(clapgpu_particles_update(stream, f->particles, f->view_mx));
    // ---- scene_update: light_grid_compute ----
    if (f->lights && f->light_tiles && f->view_mx && f->proj_mx)
        FR(clapgpu_light_grid_compute(stream, f->lights, f->view_mx, f->proj_mx, f->light_width, f->light_height, f->light_cell,
                                      f->light_tiles));
    // ---- render pass glue: ordered visible list + LOD pick ----
    if (f->frustum && f->visible && f->visible_count && f->visible_scratch) {
        FR(clapgpu_visible_compact(stream, e->vis_mask, e->vis_row_pop, e->n, f->index_base, f->visible, f->visible_count,
                                   f->visible_scratch));
        if (f->cur_lod && f->draw_lod)
            FR(clapgpu_entities_lod(stream, e, f->visible, f->visible_count, f->index_base, f->cam_pos, f->force_lod, f->cur_lod,
                                    f->draw_lod));
    }
    return CLAPGPU_OK;
}
