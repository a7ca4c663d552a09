/*
 * gpu-exports.inc.c -- CONFIG_GPU_SCENE: the engine's OWN entry points for the batched path, served by the binding.
 *
 * "batch entry points a C-ABI replacement must export (same names / signatures so clap_frame /
 * _models_render link unchanged)".  This file defines them under the reference's names; the reference's bodies
 * stay available as ref_<name> (in the engine: the same functions behind `#ifndef CONFIG_GPU_SCENE`, or renamed by
 * a two-line patch each; in the drop-in checker oracle/ref/dropin.c the renaming is done
 * by the preprocessor around its #include of model.c / view.c / light.c, so not a line of the reference changes).
 *
 *   void mq_update(struct mq *mq)                                   model.h:342   model.c:1953
 *   bool view_entity_in_frustum(struct view *view, entity3d *e)     view.h:37     view.c:296-337
 *   void view_calc_frustum(clap_context *ctx, struct view *view)    view.h:36     view.c:291
 *   void light_grid_compute(struct light *light, struct view *view) light.h:61    light.c:88-154
 *   void entity3d_position / _move / _rotate / _scale / _visible    model.h:712-770  model.c:1810-1842
 *       the reference's bodies + gpu_scene_touch(): the dirty notification that lets gpu_mq_update() run in
 *       O(touched) instead of walking every entity3d
 *   void entity3d_update(entity3d *e, void *data)                   model.h:647   model.c:1793
 *   void entity3d_reset(entity3d *e)                                model.h:532   model.c:1726
 *       one entity's update outside the frame loop (instantiate_entity model.c:1872, terrain.c:551): the reference's
 *       body -- a single entity is host work -- + gpu_scene_host_updated(), which brings the device's copy of a
 *       batched entity (and, through its seq, its children) up to date with the next mq_update
 *   void entity3d_set_lod(entity3d *e, int lod, bool force)         model.h:692   model.c:593-609
 *       the reference's body + gpu_scene_lod_changed(): the per-pass LOD pick (gpu_scene_select_lod) runs on the device
 *   void entity3d_delete(entity3d *e)                               model.h       model.c:1787-1791
 *       gpu_scene_entity_deleting() + the reference's body: a queue that loses (and, through the one line in
 *       entity3d_make, gains) a few entities a frame is not walked for it
 *   void particle_system_position(particle_system *ps, const vec3 c) particle.h:47  particle.c:132-157
 *       in gpu-particles.inc.c (the struct is private to particle.c): an attached, mirrored system carries its
 *       device-resident particles along
 *
 * A queue the binding is not bound to (the UI queue, ui.c:188) takes the reference's path unchanged.
 * Included at the end of the translation unit that holds model.c (after gpu-anim.inc.c), view.c and light.c.
 */
#ifdef CONFIG_GPU_SCENE

void ref_mq_update(struct mq *mq);
bool ref_view_entity_in_frustum(struct view *view, entity3d *e);
void ref_light_grid_compute(struct light *light, struct view *view);
void ref_view_calc_frustum(clap_context *ctx, struct view *view);

static struct gpu_lights *g_bound_lights;
void gpu_lights_bind(struct gpu_lights *gl) { g_bound_lights = gl; }

void mq_update(struct mq *mq)
{
    struct gpu_scene *gs = gpu_scene_bound();
    if (gs && mq == gpu_scene_bound_mq()) {
        const int rc = gpu_mq_update(gs, mq, gpu_scene_bound_view());
        if (!rc)
            return;
        /* a device error is not fatal to the frame -- the reference's loop still works on the same objects -- but it is
         * reported (once through the engine's log, every time in gpu_scene_last_stats()->device_errors) */
        if (!gpu_scene_device_errors())
            err("gpu_mq_update failed (%d): %s; mq_update falls back to the host loop\n", rc, clapgpu_last_error());
        gpu_scene_device_error("gpu_mq_update", rc);
    }
    ref_mq_update(mq);
}

bool view_entity_in_frustum(struct view *view, entity3d *e)
{
    struct gpu_scene *gs = gpu_scene_bound();
    return gs ? gpu_view_entity_in_frustum(gs, view, e) : ref_view_entity_in_frustum(view, e);
}

/* view.h:36, view.c:291: the reference's body, then the binding learns that the planes moved */
void view_calc_frustum(clap_context *ctx, struct view *view)
{
    ref_view_calc_frustum(ctx, view);
    gpu_scene_view_changed(gpu_scene_bound(), view);
}

void light_grid_compute(struct light *light, struct view *view)
{
    if (g_bound_lights) {
        const int rc = gpu_light_grid_compute(g_bound_lights, light, view);
        if (!rc)
            return;
        if (!gpu_scene_device_errors())
            err("gpu_light_grid_compute failed (%d): %s; light_grid_compute falls back to the host loop\n", rc, clapgpu_last_error());
        gpu_scene_device_error("gpu_light_grid_compute", rc);
    }
    ref_light_grid_compute(light, view);
}

#define GPU_EXPORT_MUTATOR(name, params, args, notify)           \
    void ref_##name params;                                     \
    void name params                                            \
    {                                                           \
        ref_##name args;                                        \
        notify(gpu_scene_bound(), e);                           \
    }

/* the four that write the transform alone leave the entity's address; entity3d_visible writes e->flags */
GPU_EXPORT_MUTATOR(entity3d_position, (entity3d *e, vec3 pos), (e, pos), gpu_scene_touch_xform)
GPU_EXPORT_MUTATOR(entity3d_move, (entity3d *e, vec3 off), (e, off), gpu_scene_touch_xform)
GPU_EXPORT_MUTATOR(entity3d_rotate, (entity3d *e, float rx, float ry, float rz), (e, rx, ry, rz), gpu_scene_touch_xform)
GPU_EXPORT_MUTATOR(entity3d_scale, (entity3d *e, float scale), (e, scale), gpu_scene_touch_xform)
GPU_EXPORT_MUTATOR(entity3d_visible, (entity3d *e, unsigned int visible), (e, visible), gpu_scene_touch)

static void gpu_before_host_update(entity3d *e) { gpu_scene_host_update_begin(gpu_scene_bound(), e); }

void ref_entity3d_update(entity3d *e, void *data);
void entity3d_update(entity3d *e, void *data)
{
    gpu_before_host_update(e);
    ref_entity3d_update(e, data);
    gpu_scene_host_updated(gpu_scene_bound(), e);
}

void ref_entity3d_reset(entity3d *e);
void entity3d_reset(entity3d *e)
{
    gpu_before_host_update(e);
    ref_entity3d_reset(e);
    gpu_scene_host_updated(gpu_scene_bound(), e);
}

/* model.h:676-692, model.c:593-609: the reference's body, then the mirror of a batched entity learns its force_lod /
 * cur_lod (gpu_scene_select_lod picks on the device from there on) */
void ref_entity3d_set_lod(entity3d *e, int lod, bool force);
void entity3d_set_lod(entity3d *e, int lod, bool force)
{
    ref_entity3d_set_lod(e, lod, force);
    gpu_scene_lod_changed(gpu_scene_bound(), e);
}

/* model.h, model.c:1787-1791: the binding hears of it BEFORE the entity goes (a batched leaf is taken out of the standing
 * device layout without a walk of the queue, gpu_scene_entity_deleting), then the reference's body.  entity3d_make and
 * entity3d_drop are static in model.c: the maintainer's patch gives each its one line. */
void ref_entity3d_delete(entity3d *e);
void entity3d_delete(entity3d *e)
{
    gpu_scene_entity_deleting(gpu_scene_bound(), e);
    ref_entity3d_delete(e);
}

#endif /* CONFIG_GPU_SCENE */
