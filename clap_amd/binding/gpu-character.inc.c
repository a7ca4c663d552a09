/*
 * gpu-character.inc.c -- CLAP-side binding of libclapgpu for the character feeder.
 *
 * character_update() (character.c:583-611) is the per-entity hook of every character: the limbo teleport out of the
 * position history (history_newest / history_fetch), the body -> entity sync of characters WITH a physics body
 * (phys_body_update: ODE, not part of the reference tree), the controlled character's motion reset, and then the chained
 * c->orig_update == default_update.  For a character WITHOUT a body everything before the chained call is a handful of
 * host instructions on the character's own state, and the chained call is exactly what gpu_mq_update() batches.  So:
 *
 *   gpu_scene_bind_characters(gs)    body-less characters whose chained hook is default_update become batchable:
 *       their host half runs first (the reference's OWN character_update, with the chained hook parked -- not a
 *       restatement), in list order, then their transform goes through the device like any other entity's; skeletal
 *       animation follows in gpu_anim_update().
 *
 * character_update and struct character's layout are private to character.c / character.h, so this file is meant to be
 * #include'd at the end of that translation unit (oracle/ref/dropin.c does).  Characters with a body keep their own hook
 * on the host: nothing that reads a dBody can be built without ODE.
 */
#include "gpu-scene.h"

static int gpu_character_parked_tail(entity3d *e, void *data)
{
    (void)e; (void)data;
    return 0;
}

/* everything character_update does before `return c->orig_update(e, data)`, by character_update itself */
static int gpu_character_host_half(entity3d *e, void *data)
{
    cresp(character) cres = entity3d_character(e);
    if (IS_CERR(cres)) return -1;
    struct character *c = cres.val;
    int (*tail)(entity3d *, void *) = c->orig_update;
    c->orig_update = gpu_character_parked_tail;
    const int rc = character_update(e, data);
    c->orig_update = tail;
    return rc;
}

/* a character the batch can take: no body (ENTITY3D_HAS_PHYSICS: phys_body_update reads ODE), default_update behind it */
static bool gpu_character_is_plain(entity3d *e, int (*default_hook)(entity3d *, void *))
{
    if (e->update != character_update || entity3d_matches(e, ENTITY3D_HAS_PHYSICS)) return false;
    cresp(character) cres = entity3d_character(e);
    return !IS_CERR(cres) && cres.val->orig_update == default_hook;
}

void gpu_scene_bind_characters(struct gpu_scene *gs)
{
    gpu_scene_characters(gs, gpu_character_is_plain, gpu_character_host_half);
}
