// frame.hip -- one display frame of the batched path as ONE C call: the sequence clap_frame() runs
// (core/clap.c:551-665) -- phys_step (clap.c:604: per fixed substep the two broadphase passes, near_callback's
// contact records, the world step) -> scene_update -> mq_update with every entity's hook in list order
// (character_update in front of default_update: body read-back, rotation push to colliders, light hand-off, TRS
// rebuild, animated_update; particles_update) -> light grid -> render-pass glue (visible list, LOD pick).
// Nothing is read back.  Every part is optional (NULL).  The caller keeps the time base
// (clapgpu_phys_step_schedule) and passes the number of substeps.
//
// Round 4: the frame as THREE CHAINS (opt-in: CLAPGPU_FRAME_OVERLAP).  What the reference runs in list order has only these
// data dependences:
//   A  physics substeps -> character hooks -> body read-back / rotation push / light hand-off -> entity update
//      (a chain of ~12 launches at the launch floor and two dozen MB each: latency, HBM idle)
//   B  animation clock -> keyframes + palette (k_pose) -> skinning (k_skin): 1.3 GB of HBM traffic, no input from A but
//      ONE -- joint world positions are e->mx * mpos (model.c:1400) and e->mx is A's last product.  k_pose therefore
//      leaves the model-space mpos (model.c:1392-1397) in joint_pos[] and a 16-byte-per-joint pass behind the join
//      multiplies it by e->mx: the same two mat4x4_mul_vec4_post, bit for bit (clapgpu_joint_pos_world)
//   C  particles (needs the view matrix alone)
// A runs on the caller's stream, B and C on two helper streams forked from it by an event and joined by two; behind
// the join: joint positions, light grid (needs A's light hand-off), visible list + LOD (need A's boxes).  Fork / join
// by events is what stream capture records as graph edges, so FrameLoop.capture() captures the overlap as it is.
// MEASURED (profiles/r04_a/frame_overlap.md): the chains do run at the same time -- 368 of a frame's 596 us have two or
// more kernels in flight -- and the frame is no shorter: 0.64-0.65 ms against 0.62-0.63 ms on one stream.  The physics
// chain is not idle time waiting to be filled: its kernels are bound by atomics, LDS and dependent loads on the SAME
// CUs, and next to k_pose (whose persistent workgroups hold every CU's registers) or k_skin they run 2-10x longer
// (k_bp_scatter 10 -> 110 us, k_bp_emit 20 -> 142 us).  Lower stream priority for B / C, a k_pose of 8 instead of 12
// wavefronts per CU, and CU-masked helper streams (slower still) changed nothing.  The default therefore stays ONE
// stream in the reference's order; the overlapped form is kept behind the flag, bit-identical (tests/test_frame_gpu.py).
#include <stdlib.h>
#include <string.h>
#include "common.h"

using namespace clapgpu;

#define FR(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

namespace {
struct FrameStreams { int dev; hipStream_t b, c; hipEvent_t fork, join_b, join_c; };
thread_local FrameStreams g_fs = { -1, nullptr, nullptr, nullptr, nullptr, nullptr };

int frame_streams(FrameStreams **out)
{
    int dev = 0;
    CLAPGPU_HIP(hipGetDevice(&dev));
    if (g_fs.dev != dev) {                                       // per thread and device, for the life of the process
        FrameStreams n = { dev, nullptr, nullptr, nullptr, nullptr, nullptr };
        hipError_t err = hipStreamCreateWithFlags(&n.b, hipStreamNonBlocking);
        if (err == hipSuccess) err = hipStreamCreateWithFlags(&n.c, hipStreamNonBlocking);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&n.fork, hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&n.join_b, hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&n.join_c, hipEventDisableTiming);
        if (err != hipSuccess) {                                 // nothing half-made is kept (or leaked)
            if (n.b) (void)hipStreamDestroy(n.b);
            if (n.c) (void)hipStreamDestroy(n.c);
            if (n.fork) (void)hipEventDestroy(n.fork);
            if (n.join_b) (void)hipEventDestroy(n.join_b);
            if (n.join_c) (void)hipEventDestroy(n.join_c);
            return hip_fail(err, "clapgpu_frame_issue: helper streams");
        }
        g_fs = n;
    }
    *out = &g_fs;
    return CLAPGPU_OK;
}
} // namespace

// Everything between the fork and the join of an overlapped frame.  Whatever it returns, the caller joins the helper
// streams before it returns itself: a failed launch in chain A must not leave the caller's stream unordered behind pose,
// skinning and particles work that is still writing (and, under stream capture, must not leave forked streams unjoined:
// EndCapture would fail and invalidate the capture).
static int frame_body(void *stream, const clapgpu_frame *f, double now, uint32_t substeps, bool overlap, FrameStreams *fs,
                      void *sb, void *sc, bool *joined);

extern "C" int clapgpu_frame_issue(void *stream, const clapgpu_frame *f, double now, uint32_t substeps)
{
    if (!f || !f->entities) return CLAPGPU_ERR_INVALID_ARGUMENTS;
    const bool animated = f->skeleton && f->animations && f->pose;
    const bool particles = f->particles && f->view_mx;
    const bool overlap = (f->flags & CLAPGPU_FRAME_OVERLAP) && (animated || particles);
    void *sb = stream, *sc = stream;                             // chains B and C: the caller's stream unless they overlap
    FrameStreams *fs = nullptr;
    if (overlap) {
        FR(frame_streams(&fs));
        CLAPGPU_HIP(hipEventRecord(fs->fork, as_stream(stream)));
        if (animated) {
            const hipError_t err = hipStreamWaitEvent(fs->b, fs->fork, 0);
            if (err != hipSuccess) return hip_fail(err, "clapgpu_frame_issue: fork");      // nothing forked yet
            sb = fs->b;
        }
        if (particles) {
            const hipError_t err = hipStreamWaitEvent(fs->c, fs->fork, 0);
            if (err != hipSuccess) {
                if (animated) {                                  // b is forked already: join it before giving up
                    (void)hipEventRecord(fs->join_b, fs->b);
                    (void)hipStreamWaitEvent(as_stream(stream), fs->join_b, 0);
                }
                return hip_fail(err, "clapgpu_frame_issue: fork");
            }
            sc = fs->c;
        }
    }
    bool joined = false;
    const int rc = frame_body(stream, f, now, substeps, overlap, fs, sb, sc, &joined);
    if (overlap && !joined) {                                    // an early exit: the one join every path goes through
        if (animated) { (void)hipEventRecord(fs->join_b, fs->b); (void)hipStreamWaitEvent(as_stream(stream), fs->join_b, 0); }
        if (particles) { (void)hipEventRecord(fs->join_c, fs->c); (void)hipStreamWaitEvent(as_stream(stream), fs->join_c, 0); }
    }
    return rc;
}

static int frame_body(void *stream, const clapgpu_frame *f, double now, uint32_t substeps, bool overlap, FrameStreams *fs,
                      void *sb, void *sc, bool *joined)
{
    const clapgpu_entities *e = f->entities;
    const bool animated = f->skeleton && f->animations && f->pose;
    const bool particles = f->particles && f->view_mx;
    // joint positions need e->mx: with the chains apart k_pose stops at the model-space position
    clapgpu_pose_batch pose_b;
    bool finish_pos = false;
    if (animated) {
        pose_b = *f->pose;
        finish_pos = overlap && pose_b.joint_pos && !(pose_b.skip & CLAPGPU_POSE_SKIP_JOINT_POS);
        if (finish_pos) pose_b.skip |= CLAPGPU_POSE_JOINT_POS_MODEL;
    }

    // ---- chain B: animated_update -- clock, pose, palette; the vertex shader's skinning loop once per frame ----
    // one stream: the clock rides the character hooks' launch (two per-character passes over different state)
    const bool clock_with_hooks = !overlap && animated && f->anim_clock && f->characters;
    auto chain_b = [&]() -> int {
        if (!animated) return CLAPGPU_OK;
        if (f->anim_clock && !clock_with_hooks) {
            if (f->now_dev) FR(clapgpu_animation_time_dev(sb, f->anim_clock, f->now_dev));
            else FR(clapgpu_animation_time(sb, f->anim_clock, now));
        }
        FR(clapgpu_pose_update(sb, f->skeleton, f->animations, &pose_b));
        if (f->skin)
            FR(clapgpu_skin(sb, f->skin));
        return CLAPGPU_OK;
    };
    // ---- chain C: particles_update hooks ----
    auto chain_c = [&]() -> int {
        if (particles) FR(clapgpu_particles_update(sc, f->particles, f->view_mx));
        return CLAPGPU_OK;
    };
    if (overlap) {                                               // issued first: their launches are in the queues while A's are made
        FR(chain_b());
        FR(chain_c());
    }

    // ---- chain A.  phys_step: per substep broadphase x2, contacts, dWorldQuickStep's body stage (physics.c:746-771) ----
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
            if ((f->flags & CLAPGPU_FRAME_PREBIN) && f->bp && f->bodies->aabb)
                FR(clapgpu_bodies_step_prebin(stream, f->bodies, f->world, 1.0 / 120.0, f->bp));
            else
                FR(clapgpu_bodies_step(stream, f->bodies, f->world, 1.0 / 120.0));      // fixed_dt, physics.c:775
        }
    }
    // ---- character_update hooks (character.c:583-611) ----
    if (clock_with_hooks)
        FR(clapgpu_characters_update_clock(stream, f->characters, e, f->bodies, f->anim_clock, now, f->now_dev));
    else if (f->characters)
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

    if (overlap) {                                               // join
        *joined = true;                                          // (a failure inside these four calls cannot be joined any better by the caller)
        if (animated) { CLAPGPU_HIP(hipEventRecord(fs->join_b, fs->b)); CLAPGPU_HIP(hipStreamWaitEvent(as_stream(stream), fs->join_b, 0)); }
        if (particles) { CLAPGPU_HIP(hipEventRecord(fs->join_c, fs->c)); CLAPGPU_HIP(hipStreamWaitEvent(as_stream(stream), fs->join_c, 0)); }
        if (finish_pos) FR(clapgpu_joint_pos_world(stream, f->skeleton, f->pose));
    } else {
        FR(chain_b());
        FR(chain_c());
    }
    // ---- scene_update: light_grid_compute ----
    if (f->lights && f->light_tiles && f->view_mx && f->proj_mx)
        FR(clapgpu_light_grid_compute(stream, f->lights, f->view_mx, f->proj_mx, f->light_width, f->light_height, f->light_cell,
                                      f->light_tiles));
    // ---- render pass glue: ordered visible list + LOD pick ----
    if (f->frustum && f->visible && f->visible_count && f->visible_scratch) {
        if (f->cur_lod && f->draw_lod)                           // list + LODs in one launch
            FR(clapgpu_visible_compact_lod(stream, e, f->index_base, f->cam_pos, f->force_lod, f->cur_lod, f->visible,
                                           f->visible_count, f->draw_lod, f->visible_scratch));
        else
            FR(clapgpu_visible_compact(stream, e->vis_mask, e->vis_row_pop, e->n, f->index_base, f->visible, f->visible_count,
                                       f->visible_scratch));
    }
    return CLAPGPU_OK;
}
